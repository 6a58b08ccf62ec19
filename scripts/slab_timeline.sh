#!/bin/bash
# Kernel timeline of one rank of the z-slab schedules (a 512 x 512 x 64 slab as a ring of one that really exchanges: the halo planes
# travel through RCCL send / recv on the halo stream, followed by the link-time hold of RGPU_COMM_EMULATE_GBPS): which kernels run
# together, and when the transfer starts and ends inside a step.  usage: slab_timeline.sh [schedule 1|2] [GB/s]
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
SCHED=${1:-1}; RATE=${2:-60}
OUT=$R/gpurun_out/slab_timeline; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/slab_tl.py <<PY
import os, sys
sys.path.insert(0, "$R")
if float("$RATE") > 0: os.environ["RGPU_COMM_EMULATE_GBPS"] = "$RATE"; os.environ["RGPU_COMM_EMULATE_PEERS"] = "2"
from ramsesgpu_amd import comm as rcomm
from ramsesgpu_amd.solver import load_library
L = load_library(); CL = rcomm.load_comm_library(rcomm.comm_lib_path(L.arithmetic))
ini = os.path.join("$R", "configs", "mhd_mri_3d.ini")
run = rcomm.CommRun(ini, "mesh.nx=512;mesh.ny=512;mesh.nz=64", 0, 1, rcomm.unique_id(CL), library=L, comm_library=CL, overlap=$SCHED, self_ring=True)
assert run.halo_bytes() > 0
run.init_simulation()
for _ in range(2): run.oneStepIntegration()
if os.environ.get("PROBE_HOST_LOOP"):
    for _ in range(10): run.oneStepIntegration()
else:
    assert run.run_steps(10) == 10
run.solver.synchronize(); run.close()
PY
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr$SCHED -o b -- python /tmp/slab_tl.py > /dev/null 2> $OUT/err$SCHED.txt
( echo "== schedule $SCHED, emulated link $RATE GB/s, $RGPU_ARITH arithmetic, host loop: ${PROBE_HOST_LOOP:-no}"
  python $R/scripts/timeline.py $(find $OUT/tr$SCHED -name "*kernel_trace.csv" | head -1) 0.5 --gantt ) | tee $OUT/timeline_s${SCHED}.txt
rm -rf $OUT/tr$SCHED
