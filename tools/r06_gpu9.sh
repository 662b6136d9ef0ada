cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06i
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -40 > gpurun_out/r06i/suite.log
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/r06i/bench.json 2> gpurun_out/r06i/bench.err
