mkdir -p gpurun_out/r04o; O=gpurun_out/r04o
timeout 2400 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -3 $O/tests.log
timeout 2500 python tools/ab4.py "cbox-spheres.toml 1024 1024 1024;brdf-row.toml 960 540 4096;mesh-box.toml 1920 1370 1024" 4 product build/v_noswitch.so > $O/ab.log 2>&1
cat $O/ab.log
