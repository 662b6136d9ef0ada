mkdir -p gpurun_out/r04d; O=gpurun_out/r04d
timeout 1500 python -m pytest tests/test_gpu_parity_r3.py tests/test_gpu_parity_r4.py -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -3 $O/tests.log
timeout 2000 python tools/ab4.py "mesh-box.toml 1920 1370 512;ibl-lens.toml 2048 2048 256" 3 product build/v_nopairs.so build/v_pairs_w6.so > $O/ab.log 2>&1
cat $O/ab.log
