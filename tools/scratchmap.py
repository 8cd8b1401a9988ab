#!/usr/bin/env python3
"""Where do the register spills of residual_stack_(h2|q)_kernel sit?  Prints, per kernel instantiation and basic block of its ISA, the runs of
matrix instructions (M<n>), scratch loads (L<n>), scratch stores (S<n>) and barriers (|) with the branch targets, so that one can see
whether a spill lands inside a matrix loop (a block that branches to itself with M48) or at a phase boundary.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/h2.s bisinger_amd/csrc/diffnet_h2.hip
    python tools/scratchmap.py /tmp/h2.s [ILb1ELb1ELi2E ...]      (template arguments as mangled: FAIR, TAIL, NCT)"""
import re,sys
lines=open(sys.argv[1]).read().split('\n')
starts=[(i,l.split(':')[0]) for i,l in enumerate(lines) if re.match(r'^_ZN3bsg.*residual_stack_(h2|q)_kernel.*:',l)]
want=sys.argv[2:]
for k,(i,name) in enumerate(starts):
    tag=re.search(r'kernel(I\w+?)EEvNS',name).group(1)
    if want and tag not in want: continue
    end=starts[k+1][0] if k+1<len(starts) else len(lines)
    body=lines[i:end]
    out=[];prev=None;cnt=0
    def flush():
        global prev,cnt
        if prev: out.append(f'{prev}{cnt}')
        prev=None;cnt=0
    for l in body:
        e=None
        if 'v_mfma' in l: e='M'
        elif 'scratch_load' in l: e='L'
        elif 'scratch_store' in l: e='S'
        elif 's_barrier' in l: e='|'
        elif re.match(r'^\.LBB\d+_\d+:',l): flush(); out.append('\n'+l.split(':')[0]); continue
        elif re.search(r's_cbranch|s_branch',l): flush(); out.append('->'+l.split()[-1]); continue
        if e is None: continue
        if e==prev: cnt+=1
        else: flush(); prev=e;cnt=1
    flush()
    print('=====',tag,len(body)); print(' '.join(out))
