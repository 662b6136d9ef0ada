cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -15 > gpurun_out/r06/suite.log
tools/profile_round.sh r06 c2 c3 c3p c3b c4 c5 > gpurun_out/r06_profile.log 2>&1
python3 tools/resource_usage.py > gpurun_out/r06/profiles/r06_resource_usage.txt 2>&1
python3 bench.py > gpurun_out/r06/profiles/r06_bench_default.json 2> gpurun_out/r06/bench_default.err
bash tools/r06_post.sh > gpurun_out/r06/post.log 2>&1
