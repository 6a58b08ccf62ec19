R=$PWD; OUT=$R/gpurun_out/ovl; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
RGPU_CHUNKS=8 rocprofv3 --kernel-trace --output-format csv -d $OUT -o ovl -- python $R/bench.py --size 256 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/err.txt
python - <<'PY'
import csv,glob
f=glob.glob('/root/repo/gpurun_out/ovl/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# take the last 60 launches of the timed region (before timers run)
sel=[r for r in rows if 'range' in r['Kernel_Name']]
t0=int(sel[200]['Start_Timestamp'])
for r in sel[200:260]:
    n=r['Kernel_Name']; short=n[n.find('K_'):n.find('>(')][:22]
    print('%-24s q=%s start=%8.3f end=%8.3f ms' % (short, r.get('Queue_Id','?'), (int(r['Start_Timestamp'])-t0)/1e6, (int(r['End_Timestamp'])-t0)/1e6))
PY
