#!/bin/bash
set -u
O=gpurun_out/r05j; mkdir -p $O
python3 tools/ab4.py "mesh-box.toml 1920 1370 1024;ibl-lens.toml 2048 2048 512" 3 build/v_base.so product,LR_CULL_SLACK=0 product,LR_CULL_SLACK=24 product,LR_CULL_SLACK=64 product,LR_CULL_SLACK=200 2>&1 | tee $O/ab_slack_k.txt
