#!/bin/bash
# registers / spills / LDS / occupancy of every kernel of a .hip file (compile-time report, no GPU needed)
# usage: tools/kernel_resources.sh patchwork-plusplus_amd/csrc/pwpp_fit.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -c -o /dev/null "$1" -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys,re,subprocess
cur=None; rows=[]
for l in sys.stdin:
    m=re.search(r"remark:\s+(.*?) \[-Rpass",l)
    if not m: continue
    t=m.group(1).strip()
    if t.startswith("Function Name:"):
        cur={"name":t.split(":",1)[1].strip()}; rows.append(cur)
    elif cur is not None and ":" in t:
        k,v=t.rsplit(":",1); cur[k.strip()]=v.strip()
for r in rows:
    n=subprocess.run(["c++filt",r["name"]],capture_output=True,text=True).stdout.strip()
    n=re.sub(r"\(anonymous namespace\)::","",n).split("(")[0].replace("void ","")
    print("%-28s VGPR %3s AGPR %3s spillV %3s spillS %3s scratch %4s occ %s LDS %s"%(n,r.get("VGPRs"),r.get("AGPRs"),r.get("VGPRs Spill"),r.get("SGPRs Spill"),r.get("ScratchSize [bytes/lane]"),r.get("Occupancy [waves/SIMD]"),r.get("LDS Size [bytes/block]")))
'
