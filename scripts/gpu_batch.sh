cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
( time timeout 1500 python -m pytest tests/test_comm_driver.py -x -q -m gpu -k "not whole_box" 2>&1 | tail -5 ) > gpurun_out/r4f/tests_comm.log 2>&1
( PROBE_LINK_GBPS="0 60 40" RGPU_ARITH=contracted timeout 1200 python scripts/slab_probe.py 2>&1 | grep "nz=\|rror"
  PROBE_LINK_GBPS="0 60 40" RGPU_ARITH=exact timeout 1200 python scripts/slab_probe.py 2>&1 | grep "nz=\|rror" ) > gpurun_out/r4f/slab_probe.log 2>&1
cat gpurun_out/r4f/tests_comm.log gpurun_out/r4f/slab_probe.log
