cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4m
( time timeout 1800 python -m pytest tests/test_comm_driver.py -x -q -m gpu -k "not whole_box" 2>&1 | tail -6 ) > gpurun_out/r4m/tests.log 2>&1
export RGPU_ARITH=contracted
for PK in 1 0; do
  echo "== RGPU_COMM_PACK=$PK" >> gpurun_out/r4m/probe.log
  ( RGPU_COMM_PACK=$PK PROBE_LINK_GBPS="0 60" timeout 900 python scripts/slab_probe.py 2>&1 | grep "nz=\|rror" ) >> gpurun_out/r4m/probe.log 2>&1
done
RGPU_COMM_PACK=1 bash scripts/slab_timeline.sh 1 60 > gpurun_out/r4m/tl1_pack.log 2>&1
RGPU_COMM_PACK=1 bash scripts/slab_timeline.sh 2 60 > gpurun_out/r4m/tl2_pack.log 2>&1
cat gpurun_out/r4m/tests.log gpurun_out/r4m/probe.log; tail -40 gpurun_out/r4m/tl1_pack.log | cut -c1-120
