#!/bin/bash
# prints kernel name, VGPRs, scratch, occupancy for every kernel in the library
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -shared -fPIC -Rpass-analysis=kernel-resource-usage -o /tmp/_ru.so $(dirname "$0")/../sdfs_via_autodiff_amd/csrc/sdfs_api.hip 2>&1 | python3 -c "
import sys,re
name=None
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: name=m.group(1); d={}
    for k in ('VGPRs','ScratchSize \[bytes/lane\]','Occupancy \[waves/SIMD\]','SGPRs Spill','VGPRs Spill'):
        m=re.search(k+r': (\d+)',l)
        if m: d[k]=m.group(1)
    if 'LDS Size' in l and name: print(name[:60], d); name=None
    if ' error' in l: print(l)
"
