mkdir -p gpurun_out/r04l; O=gpurun_out/r04l
timeout 2000 python tools/ab4.py "mesh-box.toml 1920 1370 2048;ibl-lens.toml 2048 2048 1024;cbox-spheres.toml 1024 1024 1024" 3 product product,LR_CHUNK_MIN=32 product,LR_CHUNK_MIN=64 product,LR_CHUNK_MIN=128 > $O/ab.log 2>&1
cat $O/ab.log
