#!/bin/bash
# usage (GPU box): tools/ranks_one_gpu.sh [tag]   -> profiles/<tag>_bench_c2_{1,2,8}ranks_one_gpu.json
# the coordination overhead of the N-rank bench path with all ranks on ONE GPU (round 4: 1.4 ms per frame at 8 ranks with a gloo barrier per frame;
# round 5: 0.7 ms with the shared-memory barrier); the lines carry the per-rank lists (`ranks`) the driver's SCALE run will carry
set -u
TAG=${1:-r06}
O=gpurun_out/${TAG}_ranks; mkdir -p $O profiles
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
for n in 1 2 8; do
  if [ $n -eq 1 ]; then python3 bench.py --gpus 1 --config c2 --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs > $O/bench_c2_${n}ranks_one_gpu.json 2> $O/err_$n.txt
  else python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2950$n bench.py --gpus $n --same-device --backend gloo --config c2 --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs > $O/bench_c2_${n}ranks_one_gpu.json 2> $O/err_$n.txt; fi
  python3 -c "
import json,sys
j=json.loads([l for l in open('$O/bench_c2_${n}ranks_one_gpu.json') if l.startswith('{')][-1]); print($n, 'ranks on one GPU:', j['value'], 'Msamples/s', j['ms_per_step'], 'ms per frame; per-rank render_ms', j['ranks']['render_ms'], 'readback_ms', j['ranks']['readback_ms'], 'barrier_wait_ms', j['ranks']['barrier_wait_ms'])"
  cp $O/bench_c2_${n}ranks_one_gpu.json profiles/${TAG}_bench_c2_${n}ranks_one_gpu.json
done
