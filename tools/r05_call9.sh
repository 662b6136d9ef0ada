#!/bin/bash
set -u
O=gpurun_out/r05i; mkdir -p $O
python3 -m pytest tests -m gpu -x -q -k "residual or brute or culling_slack or random_transforms or tree or mesh" > $O/gpu_tests_trav.txt 2>&1; tail -8 $O/gpu_tests_trav.txt
python3 tools/ab4.py "mesh-box.toml 1920 1370 1024;ibl-lens.toml 2048 2048 512" 4 build/v_base.so product 2>&1 | tee $O/ab_slack.txt
python3 tools/fuzz_traversal.py 1000 40 > $O/fuzz_trav.txt 2>&1; tail -3 $O/fuzz_trav.txt
