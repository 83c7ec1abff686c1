#!/bin/bash
# resource usage (registers, spills, scratch, occupancy) of the kernels of one source file matching a pattern: tools/dev/r05/resusage.sh conv_gemm.hip conv_halo [extra flags]
f=$1; pat=$2; shift 2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fno-slp-vectorize -fno-strict-aliasing -Wno-unused-result -I yolosharp_amd/csrc -c yolosharp_amd/csrc/$f -o /tmp/ru.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c "
import sys,re
cur=None; rows={}
for l in sys.stdin:
    m=re.search(r'remark:\s+(.*?) \[-Rpass', l)
    if not m: 
        if 'error' in l: print(l.strip())
        continue
    t=m.group(1).strip()
    if t.startswith('Function Name:') or t.startswith('Name:'): cur=t.split(':',1)[1].strip(); rows[cur]={}
    elif cur and ':' in t: k,v=t.split(':',1); rows[cur][k.strip()]=v.strip()
for k,v in rows.items():
    if '$pat' in k: print(k[:70].ljust(70), 'VGPR',v.get('VGPRs'),'AGPR',v.get('AGPRs'),'SGPR',v.get('TotalSGPRs'),'spillV',v.get('VGPRs Spill'),'spillS',v.get('SGPRs Spill'),'scratch',v.get('ScratchSize [bytes/lane]'),'occ',v.get('Occupancy [waves/SIMD]'))
"
