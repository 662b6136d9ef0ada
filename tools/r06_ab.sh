# usage (GPU box): tools/r06_ab.sh <tag> "<workloads>" rounds variant...   -> gpurun_out/<tag>/ab.log
cd $GRAFT_REPO_ROOT
TAG=$1; WL=$2; R=$3; shift 3
mkdir -p gpurun_out/$TAG
timeout 2400 python tools/ab4.py "$WL" $R "$@" > gpurun_out/$TAG/ab.log 2>&1
