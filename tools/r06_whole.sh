# usage (GPU box): tools/r06_whole.sh "<configs>"   -- every pixel of the stated frames against the literal oracle (tests/test_gpu_film.py, opt-in test)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06w
export LUMILLY_WHOLE_FRAMES=$(echo $1 | tr ' ' ',') LUMILLY_RECORD=gpurun_out/r06w
timeout 5000 python -m pytest tests/test_gpu_film.py -q -m gpu -k whole_stated_frame 2>&1 | tail -25 > gpurun_out/r06w/whole_$(echo $1 | tr ' ' '_').log
for c in c3p c3b; do [ -s gpurun_out/r06w/r06_bench_$c.json ] || python3 bench.py --config $c --no-cpu-baseline > gpurun_out/r06w/r06_bench_$c.json 2> gpurun_out/r06w/bench_$c.err; done
