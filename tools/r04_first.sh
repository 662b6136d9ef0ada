#!/bin/bash
# round-4 opening call: GPU suite, baseline rates, LR_DIAG phase shares of the two tree configs, tree-size probe on the fused kernel
mkdir -p gpurun_out/r04a; O=gpurun_out/r04a
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
python tools/quick_perf.py mesh-box.toml 1920 1370 1024 > $O/c4_base.log 2>&1
python tools/quick_perf.py ibl-lens.toml 2048 2048 512 > $O/c5_base.log 2>&1
LR_HIP_LIB=$PWD/build/v_diag.so python tools/quick_perf.py mesh-box.toml 1920 1370 256 > $O/c4_diag.log 2>&1
LR_HIP_LIB=$PWD/build/v_diag.so python tools/quick_perf.py ibl-lens.toml 2048 2048 128 > $O/c5_diag.log 2>&1
timeout 600 python tools/tree_size_probe.py > $O/tree_size.log 2>&1
timeout 600 python tools/c5_breakdown.py > $O/c5_breakdown.log 2>&1
tail -3 $O/tests.log; grep -h "LR_DIAG" $O/c4_diag.log $O/c5_diag.log; cat $O/tree_size.log $O/c5_breakdown.log | tail -12
