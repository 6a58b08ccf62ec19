cd $GRAFT_REPO_ROOT
EXP_NO_PARITY=1 EXP_OUT=r4s bash scripts/exp_variants.sh > gpurun_out/r4s_exp.log 2>&1
cat gpurun_out/r4s/summary.txt
