cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06e
for v in diag6 diag6ns diag_r05; do
  for wl in "mesh-box.toml 1920 1370 128 0 1" "ibl-lens.toml 2048 2048 128 0 1"; do
    echo "== $v $wl" >> gpurun_out/r06e/diag.log
    LR_HIP_LIB=$PWD/build/v_$v.so timeout 300 python tools/quick_perf.py $wl >> gpurun_out/r06e/diag.log 2>&1
  done
done
