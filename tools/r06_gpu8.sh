cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06h
timeout 900 python -m pytest tests/test_gpu_own_box.py -x -q 2>&1 | tail -15 > gpurun_out/r06h/own_box.log
timeout 1700 python tools/ab4.py "mesh-box.toml 1920 1370 512;ibl-lens.toml 2048 2048 512" 3 product build/v_parkt.so build/v_nosettle.so build/v_r05.so > gpurun_out/r06h/ab.log 2>&1
timeout 1700 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_own_box.py --deselect tests/test_gpu_parity_r5.py::test_lateral_residual_is_pinned 2>&1 | tail -25 > gpurun_out/r06h/suite.log
