#!/bin/bash
# usage (on the GPU box): tools/sweep.sh scene W H spp  -- times every build/v_*.so variant
cp lumillyrender_amd/liblumilly_hip.so /tmp/orig.so
for f in build/v_*.so; do
  cp $f lumillyrender_amd/liblumilly_hip.so
  r=$(python tools/quick_perf.py $1 $2 $3 $4 0 ${5:-1} ${6:-} | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['Msamples_s'])")
  echo "$f $r"
done
cp /tmp/orig.so lumillyrender_amd/liblumilly_hip.so
