cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06f
tools/pmc_one.sh r06f c4_product mesh-box.toml 1920 1370 256 1 > gpurun_out/r06f/pmc.log 2>&1
export LR_HIP_LIB=$PWD/build/v_nosettle.so
tools/pmc_one.sh r06f c4_nosettle mesh-box.toml 1920 1370 256 1 >> gpurun_out/r06f/pmc.log 2>&1
unset LR_HIP_LIB
tools/pmc_one.sh r06f c5_product ibl-lens.toml 2048 2048 128 1 >> gpurun_out/r06f/pmc.log 2>&1
export LR_HIP_LIB=$PWD/build/v_nosettle.so
tools/pmc_one.sh r06f c5_nosettle ibl-lens.toml 2048 2048 128 1 >> gpurun_out/r06f/pmc.log 2>&1
