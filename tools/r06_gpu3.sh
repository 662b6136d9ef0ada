set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06c
timeout 900 python -m pytest tests/test_gpu_own_box.py -x -q -k "edge_rays or live" 2>&1 | tail -15 > gpurun_out/r06c/own_box.log
timeout 1500 python tools/ab4.py "mesh-box.toml 1920 1370 512;ibl-lens.toml 2048 2048 512" 3 product build/v_preload.so build/v_nee5.so build/v_nosettle.so build/v_r05.so > gpurun_out/r06c/ab.log 2>&1
