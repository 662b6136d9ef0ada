#!/bin/bash
# usage (GPU box): tools/exp_ab.sh <tag> "<variant label>=<ENV=val,...>[@so]" ...   A/B of library variants / env switches
# each variant: throughput on the two mesh configs (reduced spp) and one PMC pass (lanes per VALU instruction, L1/L2 traffic)
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cp lumillyrender_amd/liblumilly_hip.so /tmp/cur.so
for spec in "$@"; do
  label=${spec%%=*}; rest=${spec#*=}; so=""
  if [[ "$rest" == *@* ]]; then so=${rest##*@}; rest=${rest%@*}; fi
  IFS=',' read -ra ENVV <<< "$rest"
  for kv in "${ENVV[@]}"; do [ -n "$kv" ] && export "$kv"; done
  if [ -n "$so" ]; then cp build/$so lumillyrender_amd/liblumilly_hip.so; else cp /tmp/cur.so lumillyrender_amd/liblumilly_hip.so; fi
  for cfg in "mesh-box.toml 1920 1370 256" "ibl-lens.toml 2048 2048 128"; do
    set -- $cfg
    r=$(python3 tools/quick_perf.py $1 $2 $3 $4 2>&1 | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']
f=lambda n: k[n]['ms']/max(k[n]['timed'],1)
print('%8.1f Msamples/s  iters %4d  trace %.3f shade %.3f shadow %.3f ms/launch'%(d['Msamples_s'], d['iterations'], f('trace'), f('shade'), f('shadow')))")
    echo "$label | $1 | $r" | tee -a $OUT/results.txt
  done
  if [ -z "${NO_PMC:-}" ]; then
    rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES -d $OUT/p1 -o p -- python3 tools/quick_perf.py mesh-box.toml 1920 1370 64 > /dev/null 2>&1
    rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCC_REQ TCC_HIT -d $OUT/p2 -o p -- python3 tools/quick_perf.py mesh-box.toml 1920 1370 64 > /dev/null 2>&1
    rocprofv3 --pmc FETCH_SIZE -d $OUT/p3 -o p -- python3 tools/quick_perf.py mesh-box.toml 1920 1370 64 > /dev/null 2>&1
    python3 tools/rocpd_export.py pmc $OUT/pmc_$label.json $OUT/p1/p_results.db $OUT/p2/p_results.db $OUT/p3/p_results.db variant=$label > /dev/null 2>&1
    python3 - <<PY | tee -a $OUT/results.txt
import json
d=json.load(open("$OUT/pmc_$label.json"))["kernels"]
for k,v in d.items():
    if not (k.startswith("k_trace") or k.startswith("k_shade") or k.startswith("k_shadow")): continue
    g=lambda n: v.get(n,0.0)
    print("   $label %-22s us %8.1f  VALU %.2e lanes/instr %4.1f  wait %.2f  VMEM_RD %.2e  TCP acc %.2e -> TCC rd %.2e  TCC req %.2e hit %.2f  FETCH %.1f MB"%(k[:22], g("avg_us_in_pmc_pass"), g("SQ_INSTS_VALU"), g("SQ_THREAD_CYCLES_VALU")/max(g("SQ_INSTS_VALU"),1), g("SQ_WAIT_ANY")/max(g("SQ_WAVE_CYCLES"),1), g("SQ_INSTS_VMEM_RD"), g("TCP_TOTAL_CACHE_ACCESSES"), g("TCP_TCC_READ_REQ"), g("TCC_REQ"), g("TCC_HIT")/max(g("TCC_REQ"),1), g("FETCH_SIZE")*2*1024/1e6))
PY
    rm -rf $OUT/p1 $OUT/p2 $OUT/p3
  fi
  for kv in "${ENVV[@]}"; do [ -n "$kv" ] && unset "${kv%%=*}"; done
done
cp /tmp/cur.so lumillyrender_amd/liblumilly_hip.so
