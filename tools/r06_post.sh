# usage (GPU box): tools/r06_post.sh   -- the N-rank evidence of a round: ranks-on-one-GPU bench lines and every rank's strong-scaling share
cd $GRAFT_REPO_ROOT
TAG=r06
P=gpurun_out/$TAG/profiles; mkdir -p $P
tools/ranks_one_gpu.sh $TAG > gpurun_out/$TAG/ranks_one_gpu.log 2>&1
cp gpurun_out/${TAG}_ranks/bench_c2_*ranks_one_gpu.json $P/ 2>/dev/null
for f in $P/bench_c2_*ranks_one_gpu.json; do mv $f $P/${TAG}_$(basename $f); done
python3 tools/strong_rank_probe.py cbox-spheres.toml 1024 1024 1024 0 $P/${TAG}_strong_rank_c2.json > gpurun_out/$TAG/strong_c2.log 2>&1
python3 tools/strong_rank_probe.py brdf-row.toml 960 540 4096 0 $P/${TAG}_strong_rank_c3.json > gpurun_out/$TAG/strong_c3.log 2>&1
python3 tools/strong_rank_probe.py mesh-box.toml 1920 1370 2048 0 $P/${TAG}_strong_rank_c4.json > gpurun_out/$TAG/strong_c4.log 2>&1
python3 tools/strong_rank_probe.py ibl-lens.toml 2048 2048 8192 0 $P/${TAG}_strong_rank_c5.json > gpurun_out/$TAG/strong_c5.log 2>&1
tail -2 gpurun_out/$TAG/strong_c*.log
