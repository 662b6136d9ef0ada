cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r04 c2 c3 > gpurun_out/r04_profile2.log 2>&1
tail -6 gpurun_out/r04_profile2.log | cut -c1-200
(time python bench.py) > gpurun_out/r04/profiles/r04_bench_default.json 2> gpurun_out/r04/bench_default.err
tail -4 gpurun_out/r04/bench_default.err
cut -c1-300 gpurun_out/r04/profiles/r04_bench_default.json
