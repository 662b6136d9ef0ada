#!/bin/bash
# chunk-schedule sweep: taper repeat R and body length L, on a 1/8 share and on the whole frame
set -u
O=gpurun_out/r05e; mkdir -p $O
run() { # label env... -- args
  echo "== $*" >> $O/sweep.txt
  "$@" 2>&1 | tail -1 >> $O/sweep.txt
}
for cfg in "brdf-row.toml 960 540 4096" "cbox-spheres.toml 1024 1024 1024" "ibl-lens.toml 2048 2048 2048" "mesh-box.toml 1920 1370 2048"; do
  for world in "8 3" "1 0"; do
    for knobs in "LR_TAPER=8" "LR_TAPER=16" "LR_TAPER=32" "LR_TAPER=64" "LR_TAPER=16 LR_CHUNK_LEN=8" "LR_TAPER=32 LR_CHUNK_LEN=8" "LR_TAPER=0"; do
      run env $knobs python3 tools/timeline_probe.py $cfg $world
    done
  done
done
cat $O/sweep.txt
