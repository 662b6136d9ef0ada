mkdir -p gpurun_out/r04z; O=gpurun_out/r04z
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_r3.py -x -q > $O/t.log 2>&1; tail -2 $O/t.log
timeout 2400 python tools/ab4.py "mesh-box.toml 1920 1370 1024;ibl-lens.toml 2048 2048 512;cbox-spheres.toml 1024 1024 1024" 3 product build/v_rngmad.so > $O/ab_rng2.log 2>&1
cat $O/ab_rng2.log
