#!/bin/bash
# tools/resource_usage.sh [file.hip ...]: VGPRs / spills / scratch / LDS / occupancy of every kernel, from hipcc's
# -Rpass-analysis=kernel-resource-usage (no GPU needed).  Default: the four kernel files.
cd "$(dirname "$0")/../dna-sequences-pg-extension_amd/csrc"
FILES=${@:-"superkmer_kernels.hip count_kernels.hip extract_kernels.hip filter_kernels.hip"}
for f in $FILES; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c $f -o /tmp/ru_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import re,sys
cur=None; rows=[]
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur={'name':m.group(1)}; rows.append(cur); continue
    m=re.search(r'remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)',l)
    if m and cur is not None: cur[m.group(1).strip()]=int(m.group(2))
import subprocess
for r in rows:
    nm=subprocess.run(['c++filt',r['name']],capture_output=True,text=True).stdout.strip().split('(')[0]
    print('%-60s vgpr %3d  spill v%-3d s%-3d scratch %4d B  lds %6d  occ %d' % (nm[:60], r.get('VGPRs',0), r.get('VGPRs Spill',0), r.get('SGPRs Spill',0), r.get('ScratchSize',0), r.get('LDS Size',0), r.get('Occupancy',0)))
"
done
rm -f /tmp/ru_$$.o
