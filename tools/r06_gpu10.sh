cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06l
timeout 900 python -m pytest tests/test_gpu_traversal.py tests/test_gpu_film.py -x -q -k "own_box or stated_size" 2>&1 | tail -8 > gpurun_out/r06l/own_box.log
timeout 2000 python tools/ab4.py "cbox-spheres.toml 1024 1024 1024;mesh-box.toml 1920 1370 512;ibl-lens.toml 2048 2048 512;brdf-row.toml 960 540 2048" 3 product build/v_rows0.so build/v_rows2.so build/v_r05.so > gpurun_out/r06l/ab.log 2>&1
