cd $GRAFT_REPO_ROOT
EXP_OUT=r4u bash scripts/exp_variants.sh > gpurun_out/r4u_exp.log 2>&1
cat gpurun_out/r4u/summary.txt
