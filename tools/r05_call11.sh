#!/bin/bash
set -u
O=gpurun_out/r05k; mkdir -p $O
python3 -m pytest tests -m gpu -x -q -k "residual or brute or culling_slack or random_transforms" > $O/gpu_tests_trav.txt 2>&1; tail -4 $O/gpu_tests_trav.txt
python3 tools/fuzz_traversal.py 2000 120 > $O/fuzz_trav.txt 2>&1; grep -v "outside own box 0$" $O/fuzz_trav.txt | tail -12
