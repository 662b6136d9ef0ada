#!/bin/bash
# usage: tools/_full_ab.sh "label=ENV=val,..[@so]" ...   full-size C4 / C5 throughput per variant (same box)
cp lumillyrender_amd/liblumilly_hip.so /tmp/cur.so
for rep in 1 2; do
for spec in "$@"; do
  label=${spec%%=*}; rest=${spec#*=}; so=""
  if [[ "$rest" == *@* ]]; then so=${rest##*@}; rest=${rest%@*}; fi
  IFS=',' read -ra ENVV <<< "$rest"
  for kv in "${ENVV[@]}"; do [ -n "$kv" ] && export "$kv"; done
  if [ -n "$so" ]; then cp build/$so lumillyrender_amd/liblumilly_hip.so; else cp /tmp/cur.so lumillyrender_amd/liblumilly_hip.so; fi
  for c in "mesh-box.toml 1920 1370 ${SPP4:-2048}" "ibl-lens.toml 2048 2048 ${SPP5:-1024}"; do
    echo "$label | $c: $(python3 tools/quick_perf.py $c 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print(d['Msamples_s'], {n: round(k[n]['ms']/max(k[n]['timed'],1),3) for n in ('trace','shade','shadow')})")"
  done
  for kv in "${ENVV[@]}"; do [ -n "$kv" ] && unset "${kv%%=*}"; done
done
done
cp /tmp/cur.so lumillyrender_amd/liblumilly_hip.so
