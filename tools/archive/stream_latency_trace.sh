#!/bin/bash
# one stateful stream in steady state: kernel durations and gaps from a rocprofv3 kernel trace (median over the last 60 steps of kitti 0)
# usage: tools/stream_latency_trace.sh   (PWPP_ONE_PASS_MIN_FRAMES=1 in the environment: one-pass binning for the single stream)
export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/slat_tr
rocprofv3 --kernel-trace --output-format csv -d /tmp/slat_tr -o t -- python tools/stream_trace.py > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,collections,statistics
f=glob.glob('/tmp/slat_tr/**/t_kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'k_' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
def short(n): return n.split('::')[-1].split('(')[0][:24]
seqs=[];cur=[]
for r in rows:
    if 'k_czm_bin' in r['Kernel_Name'] and cur:
        seqs.append(cur); cur=[]
    cur.append(r)
seqs.append(cur)
seqs=[s for s in seqs[-120:]]
# steps cycle through 6 source frames: take every 6th (same source as the last one)
sel=seqs[::-1][::6]
L=len(sel[0]); sel=[s for s in sel if len(s)==L]
tot=[(int(s[-1]['End_Timestamp'])-int(s[0]['Start_Timestamp']))/1e3 for s in sel]
print('first kernel start -> last kernel end: median %.1f us over %d steps'%(statistics.median(tot),len(sel)))
for i in range(L):
    d=[(int(s[i]['End_Timestamp'])-int(s[i]['Start_Timestamp']))/1e3 for s in sel]
    g=[(int(s[i]['Start_Timestamp'])-int(s[i-1]['End_Timestamp']))/1e3 for s in sel] if i else [0]
    print('%-26s %6.1f us   gap before %5.1f us'%(short(sel[0][i]['Kernel_Name']),statistics.median(d),statistics.median(g)))
PY
