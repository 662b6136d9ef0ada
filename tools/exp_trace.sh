#!/bin/bash
# usage (GPU box): tools/exp_trace.sh <tag>  -- A/B of the streaming traversal variants on the mesh configs (reduced spp)
TAG=${1:-exp}
OUT=gpurun_out/$TAG; mkdir -p $OUT
run() {  # label env...
  local label=$1; shift
  for cfg in "mesh-box.toml 1920 1370 256" "ibl-lens.toml 2048 2048 128"; do
    set -- $cfg
    r=$(env "${ENVV[@]}" python3 tools/quick_perf.py $1 $2 $3 $4 | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']
f=lambda n: k[n]['ms']/max(k[n]['timed'],1)
print('%8.1f Msamples/s  iters %4d  trace %.3f shade %.3f shadow %.3f ms/launch  slots %d'%(d['Msamples_s'], d['iterations'], f('trace'), f('shade'), f('shadow'), d['path_slots']))")
    echo "$label | $1 | $r" | tee -a $OUT/results.txt
  done
}
ENVV=(LR_SORT=0); run "nosort          "
ENVV=(LR_SORT=1); run "sort 16K window "
ENVV=(LR_SORT=1 LR_MAXGROUP=8); run "sort 4K window  "
ENVV=(LR_SORT=1 LR_MAXGROUP=4); run "sort 2K window  "
ENVV=(LR_SORT=0 LR_MAXGROUP=8); run "nosort 4K pass  "
