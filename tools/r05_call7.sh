#!/bin/bash
set -u
O=gpurun_out/r05g; mkdir -p $O
python3 -m pytest tests/test_gpu_parity_r5.py -x -q -k "bands or three_gigabytes or stated" > $O/gpu_tests_bands.txt 2>&1; tail -5 $O/gpu_tests_bands.txt
for a in "ibl-lens.toml 2048 2048 2048" "mesh-box.toml 1920 1370 2048" "brdf-row.toml 960 540 4096" "cbox-spheres.toml 1024 1024 1024"; do
  echo "== $a bands (default)"; python3 tools/timeline_probe.py $a 1 0 2>&1 | tail -2
  echo "== $a one band"; LR_BAND_PIX=100000000 python3 tools/timeline_probe.py $a 1 0 2>&1 | tail -2
done
echo "== c5 band sweep"
for b in 65536 131072 262144 524288 1048576; do echo "band $b"; LR_BAND_PIX=$b python3 tools/timeline_probe.py ibl-lens.toml 2048 2048 2048 1 0 2>&1 | tail -1; done
echo "== c4 band sweep"
for b in 131072 262144 524288 1048576; do echo "band $b"; LR_BAND_PIX=$b python3 tools/timeline_probe.py mesh-box.toml 1920 1370 2048 1 0 2>&1 | tail -1; done
echo "== c5 share of 8, band sweep"
for b in 65536 131072 262144 100000000; do echo "band $b"; LR_BAND_PIX=$b python3 tools/timeline_probe.py ibl-lens.toml 2048 2048 2048 8 3 2>&1 | tail -1; done
python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -5 $O/gpu_tests.txt
