mkdir -p gpurun_out/r04g; O=gpurun_out/r04g
timeout 1500 python -m pytest tests/test_gpu_parity_r4.py -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -15 $O/tests.log
timeout 1200 python tools/residual_probe.py 22 > $O/residual.log 2>&1
tail -70 $O/residual.log
