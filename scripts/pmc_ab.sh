#!/bin/bash
# A/B of hardware counters of the MHD sweep between builds of the library: pmc_ab.sh <lib.so> ...  (512^3 MRI box, 2 steps each)
# -> gpurun_out/$JOB_OUT/pmc_<lib>.txt: per-launch sums of the main sweep kernel.  Counters in their own passes (no tracing next to --pmc).
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
OUT=$R/gpurun_out/${JOB_OUT:-job}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for so in "$@"; do
  tag=$(basename $so .so)
  # PMC_GROUPS="A B C|D E": other counter groups, one pass each (default: the two SQ groups below)
  DEFAULT_GROUPS="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES|SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
  IFS='|' read -ra GROUPS_ <<< "${PMC_GROUPS:-$DEFAULT_GROUPS}"
  for grp in "${GROUPS_[@]}"; do
    d=$OUT/pmc_${tag}_$(echo $grp | md5sum | cut -c1-8)
    RGPU_LIB=$R/ramsesgpu_amd/$tag.so rocprofv3 --pmc $grp --output-format csv -d $d -o pmc -- python $R/scripts/probe_sweep.py mhd_mri_3d ${PMC_N:-512} 2 > /dev/null 2> $d.err
  done
  python - "$OUT" "$tag" <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("%s/pmc_%s_*/**/*counter_collection.csv" % (out, tag), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "mhd3d_sweep_kernel" in k and "MhTile<16" in k: name = "sweep"
        elif "K_mhd_update3d" in k: name = "update"
        else: continue
        acc[(name, row["Counter_Name"])][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
with open("%s/pmc_%s.txt" % (out, tag), "w") as o:
    for (name, c), per in sorted(acc.items()):
        o.write("%-8s %-28s %14.5g  (%d launches)\n" % (name, c, sum(per.values()) / len(per), len(per)))
print(open("%s/pmc_%s.txt" % (out, tag)).read())
PY
  rm -rf $OUT/pmc_${tag}_*/
done
