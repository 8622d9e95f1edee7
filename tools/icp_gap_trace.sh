R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/icp_lp; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o lp -- python $R/tools/prof_icp.py replica 40 > $O/log.txt 2>&1
tail -2 $O/log.txt
python - <<'PY'
import csv,glob,os,collections
O=os.environ.get('GRAFT_REPO_ROOT')+'/gpurun_out/icp_lp'
f=glob.glob(O+'/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# take the last full track: find sequence of icp_reduce
names=[r['Kernel_Name'] for r in rows]
idx=[i for i,n in enumerate(names) if 'icp_reduce' in n]
last=idx[-15:]
t0=int(rows[last[0]]['Start_Timestamp'])
prev_end=None
for i in range(last[0]-3, min(len(rows), last[-1]+3)):
    r=rows[i]; s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    gap = (s-prev_end)/1e3 if prev_end else 0
    print('%-40s start %8.1f dur %6.1f gap %5.1f grid %s' % (r['Kernel_Name'][:40], (s-t0)/1e3, (e-s)/1e3, gap, r.get('Grid_Size','?')))
    prev_end=e
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
