#!/usr/bin/env python
"""Scan the gfx950 ISA of every kernel for the two patterns that cost this repo the most ("guarded memory
operations"): stores that are each preceded by `s_waitcnt vmcnt(0)` (a tile's stores complete one after the other) and
`s_waitcnt vmcnt(0)` inside loops (a software prefetch drained at its first use).
    mkdir -p /tmp/isa/scan && cd /tmp/isa/scan && for f in $REPO/mlagg-unet_amd/csrc/*.hip; do
        hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -save-temps -c $f -o $(basename $f .hip).o -I$REPO/include -I$REPO/mlagg-unet_amd/csrc; done
    python tools/isa_scan.py [/tmp/isa/scan]
"""
import sys
DIR = sys.argv[1] if len(sys.argv) > 1 else "/tmp/isa/scan"
import re, glob, collections
for f in sorted(glob.glob(DIR + '/*gfx950.s')):
    S=open(f).read().split('\n')
    i=0
    while i < len(S):
        l=S[i]
        m=re.match(r'^(_Z\w+):',l)
        if m and i+1 < len(S):
            name=m.group(1)
            j=i+1
            while j < len(S) and not S[j].startswith('.Lfunc_end'): j+=1
            body=S[i:j]
            st=[k for k,x in enumerate(body) if 'global_store' in x or 'buffer_store' in x]
            w0=[k for k,x in enumerate(body) if 's_waitcnt vmcnt(0)' in x]
            ld=[k for k,x in enumerate(body) if 'global_load' in x or 'buffer_load' in x]
            # serialized stores: a vmcnt(0) with a store within the previous 14 lines AND a store within next 6 lines
            ser=0
            for k in w0:
                prev=any(k-14 <= s < k for s in st); nxt=any(k < s <= k+8 for s in st)
                if prev and nxt: ser+=1
            # waits in loops: vmcnt(0) located between a loop header label and its back branch: approximate by "in Loop" annotation on the nearest preceding label
            inloop=0
            lab=''
            for k,x in enumerate(body):
                if x.startswith('.LBB'): lab=x
                if 's_waitcnt vmcnt(0)' in x and ('Loop' in lab): inloop+=1
            import subprocess
            short=subprocess.run(['c++filt',name],capture_output=True,text=True).stdout.strip()[:100]
            if ser>0 or inloop>0:
                print(f"{f.split('/')[-1].split('-hip')[0]:18s} st={len(st):3d} ld={len(ld):3d} vmcnt0={len(w0):3d} serialized_stores={ser:3d} vmcnt0_in_loops={inloop:3d}  {short}")
            i=j
        else:
            i+=1
