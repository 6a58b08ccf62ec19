cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4o
bash scripts/prof_round.sh r04 > gpurun_out/r4o/prof.log 2>&1
tail -3 gpurun_out/r4o/prof.log | cut -c1-300
