#!/bin/bash
# register / spill table of one HIP source: tools/regs.sh csrc/file.hip [filter]
cd "$(dirname "$0")/../multiposenet/pytorch_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../../include $EXTRA -c "$1" -o /tmp/regs_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys,re,subprocess
cur=None; rows=[]
for line in sys.stdin:
    m=re.search(r"Function Name: (\S+)", line)
    if m: cur={"name":m.group(1)}; rows.append(cur); continue
    for key in ("TotalSGPRs","VGPRs","AGPRs","Occupancy \[waves/SIMD\]","SGPRs Spill","VGPRs Spill","LDS Size \[bytes/block\]"):
        m=re.search(r"remark:\s+"+key+r": (\d+)", line)
        if m and cur is not None: cur[key]=m.group(1)
for r in rows:
    n=subprocess.run(["c++filt",r["name"]],capture_output=True,text=True).stdout.strip()
    n=n.replace("(anonymous namespace)::","")[:90]
    print("%-92s sgpr %3s vgpr %3s agpr %3s occ %s spill s%s v%s lds %s"%(n,r.get("TotalSGPRs"),r.get("VGPRs"),r.get("AGPRs"),r.get("Occupancy \\[waves/SIMD\\]"),r.get("SGPRs Spill"),r.get("VGPRs Spill"),r.get("LDS Size \\[bytes/block\\]")))
' | grep -E "${2:-.}"
rm -f /tmp/regs_$$.o
