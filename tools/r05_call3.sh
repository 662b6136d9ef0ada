#!/bin/bash
# round 5, GPU call 3: GPU suite on the tapered chunk schedule + diagonal deal; timelines and all-ranks probe after
set -u
O=gpurun_out/r05c; mkdir -p $O
python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -15 $O/gpu_tests.txt
export LR_HIP_LIB=$PWD/build/v_timeline.so
for a in "cbox-spheres.toml 1024 1024 1024" "brdf-row.toml 960 540 4096" "mesh-box.toml 1920 1370 2048" "ibl-lens.toml 2048 2048 2048"; do
  echo "== $a world 1" >> $O/timeline.txt; python3 tools/timeline_probe.py $a 1 0 2>&1 | tail -2 >> $O/timeline.txt
  echo "== $a world 8 rank 3" >> $O/timeline.txt; python3 tools/timeline_probe.py $a 8 3 2>&1 | tail -2 >> $O/timeline.txt
done
unset LR_HIP_LIB
cat $O/timeline.txt
python3 tools/strong_rank_probe.py cbox-spheres.toml 1024 1024 1024 0 $O/r05_strong_rank_c2.json 3 2>&1 | tail -4
python3 tools/strong_rank_probe.py brdf-row.toml 960 540 4096 0 $O/r05_strong_rank_c3.json 3 2>&1 | tail -4
python3 tools/strong_rank_probe.py mesh-box.toml 1920 1370 2048 0 $O/r05_strong_rank_c4.json 2 2>&1 | tail -4
python3 tools/strong_rank_probe.py ibl-lens.toml 2048 2048 2048 0 $O/r05_strong_rank_c5.json 2 2>&1 | tail -4
