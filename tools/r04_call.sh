mkdir -p gpurun_out/r04i; O=gpurun_out/r04i
LR_DEBUG=1 python tools/absurd_probe.py 2>&1 | grep -v "^\[lr\] wide\|radix sort" | tail -12
timeout 1500 python -m pytest tests/test_gpu_parity_r3.py tests/test_gpu_parity_r2.py -x -q -k "not two_rank" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -4 $O/tests.log
timeout 2000 python tools/ab4.py "ibl-lens.toml 2048 2048 256" 3 product build/v_noemrec.so > $O/ab.log 2>&1
cat $O/ab.log
