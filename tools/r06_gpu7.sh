cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06g
timeout 900 python -m pytest tests/test_gpu_own_box.py -x -q 2>&1 | tail -15 > gpurun_out/r06g/own_box.log
timeout 1700 python tools/ab4.py "cbox-spheres.toml 1024 1024 1024;mesh-box.toml 1920 1370 512;ibl-lens.toml 2048 2048 512;brdf-row.toml 960 540 2048" 3 product build/v_nosettle.so build/v_r05.so > gpurun_out/r06g/ab.log 2>&1
