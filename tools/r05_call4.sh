#!/bin/bash
set -u
O=gpurun_out/r05d; mkdir -p $O
export LR_HIP_LIB=$PWD/build/v_timeline.so
python3 tools/timeline_probe.py brdf-row.toml 960 540 4096 8 3 2>&1 | tail -3
python3 tools/timeline_probe.py brdf-row.toml 960 540 4096 1 0 2>&1 | tail -3
