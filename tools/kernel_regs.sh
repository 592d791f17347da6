#!/bin/bash
# usage: tools/kernel_regs.sh <file.hip> [name pattern]   -> VGPRs / spills / LDS / occupancy per kernel (compiler remarks)
F=$1; PAT=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -c $F -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
python3 -c "
import sys,re
cur=None; rows={}
for ln in sys.stdin:
    m=re.search(r'Function Name: (\S+)',ln)
    if m: cur=m.group(1); rows[cur]={}; continue
    m=re.search(r'remark:\s+([A-Za-z ]+(?:\[[A-Za-z/]*\])?): (\d+)',ln)
    if m and cur: rows[cur][m.group(1).strip()]=m.group(2)
import subprocess
for k,v in rows.items():
    d=subprocess.run(['c++filt',k],capture_output=True,text=True).stdout.strip()
    if re.search(r'$PAT',d): print(d[:100],'|',' '.join(f'{a}={b}' for a,b in v.items() if a in ('VGPRs','AGPRs','ScratchSize [bytes/lane]','Occupancy [waves/SIMD]','LDS Size [bytes/block]','TotalSGPRs','VGPRs Spill')))
"
