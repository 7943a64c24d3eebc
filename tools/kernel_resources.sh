#!/bin/bash
# Register / LDS / occupancy of every kernel as the compiler reports it (no GPU needed):
#   tools/kernel_resources.sh [-DKNOB=..]   ->  name, VGPRs, SGPRs, scratch bytes, occupancy (waves per SIMD)
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -c ring_zk_amd/csrc/rzk_kernels.hip -o /dev/null \
  -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c "
import re,sys,subprocess
txt=sys.stdin.read()
cur=None;rows=[]
for line in txt.splitlines():
    m=re.search(r'Function Name: (\S+)',line)
    if m: cur={'name':m.group(1)}; rows.append(cur); continue
    for key,pat in (('vgpr',r' VGPRs: (\d+)'),('agpr',r'AGPRs: (\d+)'),('sgpr',r'SGPRs: (\d+)'),('scratch',r'ScratchSize \[bytes/lane\]: (\d+)'),('occ',r'Occupancy \[waves/SIMD\]: (\d+)'),('lds',r'LDS Size \[bytes/block\]: (\d+)')):
        m=re.search(pat,line)
        if m and cur is not None: cur[key]=int(m.group(1))
names=subprocess.run(['c++filt']+[r['name'] for r in rows],capture_output=True,text=True).stdout.splitlines()
for r,n in zip(rows,names):
    n=re.sub(r'\(.*','',n).replace('void rzk::','')
    print('%-48s vgpr %3d sgpr %3d scratch %4d occ %d'%(n,r.get('vgpr',-1),r.get('sgpr',-1),r.get('scratch',-1),r.get('occ',-1)))
"
