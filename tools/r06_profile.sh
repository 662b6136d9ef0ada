cd $GRAFT_REPO_ROOT
tools/profile_round.sh r06 c2 c3 c3p c3b c4 c5 > gpurun_out/r06_profile.log 2>&1
python3 tools/resource_usage.py > gpurun_out/r06/profiles/r06_resource_usage.txt 2>&1
