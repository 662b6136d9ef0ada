#!/bin/bash
# round 5, final measurement pass on one box: GPU suite, the profile passes of the four configs at their stated sizes, the all-ranks
# strong-scaling probe, the default bench line
set -u
O=gpurun_out/r05z; mkdir -p $O
python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -4 $O/gpu_tests.txt
bash tools/profile_round.sh r05 c2 c3 c4 c5 > $O/profile_round.log 2>&1; tail -30 $O/profile_round.log
python3 tools/strong_rank_probe.py cbox-spheres.toml 1024 1024 1024 0 $O/r05_strong_rank_c2.json 3 2>&1 | tail -4
python3 tools/strong_rank_probe.py brdf-row.toml 960 540 4096 0 $O/r05_strong_rank_c3.json 3 2>&1 | tail -4
python3 tools/strong_rank_probe.py mesh-box.toml 1920 1370 2048 0 $O/r05_strong_rank_c4.json 2 2>&1 | tail -4
python3 tools/strong_rank_probe.py ibl-lens.toml 2048 2048 8192 0 $O/r05_strong_rank_c5.json 1 2>&1 | tail -4
python3 bench.py > $O/r05_bench_default.json 2> $O/bench_default.err; cut -c1-600 $O/r05_bench_default.json
