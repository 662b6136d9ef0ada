mkdir -p gpurun_out/r04m; O=gpurun_out/r04m
timeout 1500 python -m pytest tests/test_gpu_parity_r2.py tests/test_gpu_parity_r3.py tests/test_gpu_parity.py -x -q -k "not two_rank" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -3 $O/tests.log
timeout 2000 python tools/ab4.py "mesh-box.toml 1920 1370 1024;ibl-lens.toml 2048 2048 512" 3 product build/v_base.so > $O/ab.log 2>&1
cat $O/ab.log
