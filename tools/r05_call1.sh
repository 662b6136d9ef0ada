#!/bin/bash
# round 5, GPU call 1: all-ranks strong-scaling probe BEFORE the tile split changes + the EXEC-half microbenchmark
set -u
O=gpurun_out/r05a; mkdir -p $O
./build/exec_half > $O/exec_half.txt 2>&1; cat $O/exec_half.txt
python3 tools/strong_rank_probe.py cbox-spheres.toml 1024 1024 1024 64 $O/r05_strong_rank_before_c2.json 3 2>&1 | tail -4
python3 tools/strong_rank_probe.py brdf-row.toml 960 540 4096 64 $O/r05_strong_rank_before_c3.json 3 2>&1 | tail -4
python3 tools/strong_rank_probe.py mesh-box.toml 1920 1370 2048 64 $O/r05_strong_rank_before_c4.json 2 2>&1 | tail -4
python3 tools/strong_rank_probe.py ibl-lens.toml 2048 2048 2048 64 $O/r05_strong_rank_before_c5.json 2 2>&1 | tail -4
