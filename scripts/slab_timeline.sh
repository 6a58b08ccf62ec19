#!/bin/bash
# Kernel timeline of one rank of the z-slab schedule (a 512 x 512 x 64 slab, its own z neighbour: the halo planes travel through
# RCCL send / recv on the halo stream): which kernels run together -- the exchange next to the update of the inner planes.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
OUT=$R/gpurun_out/slab_timeline; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/slab_tl.py <<PY
import os, sys
sys.path.insert(0, "$R")
from ramsesgpu_amd import comm as rcomm
from ramsesgpu_amd.solver import load_library
L = load_library(); CL = rcomm.load_comm_library()
ini = os.path.join("$R", "configs", "mhd_mri_3d.ini")
run = rcomm.CommRun(ini, "mesh.nx=512;mesh.ny=512;mesh.nz=64", 0, 1, rcomm.unique_id(CL), library=L, comm_library=CL, overlap=True)
run.init_simulation()
for _ in range(12): run.oneStepIntegration()
run.solver.synchronize(); run.close()
PY
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o b -- python /tmp/slab_tl.py > /dev/null 2> $OUT/err.txt
python $R/scripts/timeline.py $(find $OUT/tr -name "*kernel_trace.csv" | head -1) 0.5 | tee $OUT/timeline.txt
rm -rf $OUT/tr
