#!/bin/bash
set -u
O=gpurun_out/r05h; mkdir -p $O
python3 -m pytest tests/test_gpu_parity_r5.py -x -q -k "bands" > $O/gpu_tests_bands.txt 2>&1; tail -5 $O/gpu_tests_bands.txt
for a in "ibl-lens.toml 2048 2048 2048" "mesh-box.toml 1920 1370 2048" "brdf-row.toml 960 540 4096" "cbox-spheres.toml 1024 1024 1024"; do
  for w in "1 0" "8 3"; do
    for sh in 17 0 16 18; do echo "== $a world $w sub_shift $sh"; LR_SUB_SHIFT=$sh python3 tools/timeline_probe.py $a $w 2>&1 | tail -1; done
  done
done
python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -5 $O/gpu_tests.txt
