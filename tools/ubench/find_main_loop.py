#!/usr/bin/env python3
"""usage: find_main_loop.py <kernel.s> <mangled kernel name> -> first label, label behind the loop, static VALU count of the depth-1 loop
with the most vector instructions (its blocks by the compiler's loop annotations): the range tools/ubench/stream_replay_gen.py replays."""
import re,sys,collections
asm,kern=sys.argv[1],sys.argv[2]
L=open(asm).read().split('\n')
k0=next(i for i,l in enumerate(L) if l.startswith(kern+':'))
k1=next(i for i in range(k0,len(L)) if 's_endpgm' in L[i])
body=L[k0:k1+1]
labs=[(i,re.match(r'^(\.LBB\d+_\d+):(.*)',l)) for i,l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:',l)]
# depth-1 loops: header name -> member label indices
mem=collections.defaultdict(list)
for idx,(i,m) in enumerate(labs):
    c=m.group(2)
    h=re.search(r'Loop Header: Depth=1',c)
    if h: mem[m.group(1)[1:].replace('L','',1) if False else m.group(1)].append(idx)
    g=re.search(r'Header=(BB\d+_\d+) Depth=1',c)
    if g: mem['.L'+g.group(1)].append(idx)
    g2=re.search(r'Parent Loop (BB\d+_\d+) Depth=1',c)
    if g2: mem['.L'+g2.group(1)].append(idx)
best=None
for h,ids in mem.items():
    lo,hi=min(ids),max(ids)
    i0=labs[lo][0]; i1=labs[hi+1][0] if hi+1<len(labs) else len(body)
    nv=sum(1 for l in body[i0:i1] if l.strip().startswith('v_'))
    if best is None or nv>best[2]: best=(labs[lo][1].group(1), labs[hi+1][1].group(1) if hi+1<len(labs) else 'end', nv)
print(*best)
