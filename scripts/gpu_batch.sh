cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4d
( RGPU_ARITH=contracted PROBE_NZ=64 PROBE_LINK_GBPS="0 1" timeout 300 python scripts/slab_probe.py 2>&1 | grep "nz=" ) > gpurun_out/r4d/knob.log 2>&1
( PROBE_LINK_GBPS="0 60 40" RGPU_ARITH=contracted timeout 900 python scripts/slab_probe.py 2>&1 | grep "nz="
  PROBE_LINK_GBPS="0 60 40" RGPU_ARITH=exact timeout 900 python scripts/slab_probe.py 2>&1 | grep "nz=" ) > gpurun_out/r4d/slab_probe.log 2>&1
( for so in librgpu.so librgpu_exp_noalf.so; do echo "== $so"; RGPU_LIB=$PWD/ramsesgpu_amd/$so timeout 600 python scripts/probe_2d.py 2>&1 | grep orszag; done ) > gpurun_out/r4d/ot2d.log 2>&1
cat gpurun_out/r4d/knob.log gpurun_out/r4d/slab_probe.log gpurun_out/r4d/ot2d.log
