mkdir -p gpurun_out/r04z; O=gpurun_out/r04z
timeout 3000 python tools/ab4.py "mesh-box.toml 1920 1370 1024;ibl-lens.toml 2048 2048 512" 3 product build/v_bpt3.so build/v_bpt1.so build/v_bn2.so build/v_bn4.so build/v_re3.so build/v_re5.so > $O/ab.log 2>&1
cat $O/ab.log
