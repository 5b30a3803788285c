#!/bin/bash
# single frame: kernel durations and the gaps between them from a rocprofv3 kernel trace (no per-kernel events in the stream)
export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/lat_tr
rocprofv3 --kernel-trace --output-format csv -d /tmp/lat_tr -o t -- python tools/latency_kernels.py $1 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,collections,statistics
f=glob.glob('/tmp/lat_tr/**/t_kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'k_' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
def short(n): return n.split('::')[-1].split('(')[0][:22]
# take the un-profiled phase: sequences starting with k_clear
seqs=[];cur=[]
for r in rows:
    if 'k_clear' in r['Kernel_Name'] or ('k_czm_bin' in r['Kernel_Name'] and cur and 'k_clear' not in cur[-1]['Kernel_Name']):
        if cur: seqs.append(cur)
        cur=[]
    cur.append(r)
seqs=[s for s in seqs if len(s)==len(seqs[20])][10:60]
names=[short(r['Kernel_Name']) for r in seqs[0]]
dur=collections.defaultdict(list);gap=collections.defaultdict(list);tot=[]
for s in seqs:
    for i,r in enumerate(s):
        dur[i].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
        if i: gap[i].append((int(r['Start_Timestamp'])-int(s[i-1]['End_Timestamp']))/1e3)
    tot.append((int(s[-1]['End_Timestamp'])-int(s[0]['Start_Timestamp']))/1e3)
print('first kernel start -> last kernel end: median %.1f us'%statistics.median(tot))
for i,n in enumerate(names):
    print('%-24s %6.1f us   gap before %5.1f us'%(n,statistics.median(dur[i]),statistics.median(gap[i]) if i else 0))
PY
