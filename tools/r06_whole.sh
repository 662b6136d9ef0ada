# usage (GPU box): tools/r06_whole.sh "<configs>" [k/n]   -- every pixel (of band k of n) of the stated frames against the literal oracle
# (tests/test_gpu_film.py::test_whole_stated_frame_against_the_live_oracle, opt-in); records under gpurun_out/r06w/
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06w
export LUMILLY_WHOLE_FRAMES=$(echo $1 | tr ' ' ',') LUMILLY_RECORD=gpurun_out/r06w LUMILLY_WHOLE_BAND=${2:-0/1}
timeout 6500 python -m pytest tests/test_gpu_film.py -q -m gpu -k whole_stated_frame 2>&1 | tail -25 > gpurun_out/r06w/whole_$(echo $1 | tr ' ' '_')_$(echo ${2:-0/1} | sed 's:/:of:').log
