import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import numpy as np
import oracle as om
from rattle_amd.api import Context
orc=om.Oracle(); ctx=Context(0)
rng=np.random.default_rng(1)
ACGT=np.frombuffer(b"ACGT",np.uint8)
def mutate(base, rate):
    out=[]
    for c in base:
        r=rng.random()
        if r<rate/3: continue
        if r<2*rate/3: out.append(int(ACGT[rng.integers(0,4)])); continue
        out.append(int(c))
        if rng.random()<rate/3: out.append(int(ACGT[rng.integers(0,4)]))
    return bytes(out)
bad=None
for trial in range(300):
    L=int(rng.integers(20,90)); n=int(rng.integers(2,7))
    base=ACGT[rng.integers(0,4,L)]
    pack=[mutate(base,0.15) for _ in range(n)]
    pack=[p for p in pack if len(p)>0]
    rows,width,c=ctx.poa_msa([pack]); want,_=orc.poa_msa(pack)
    if rows[0]!=want:
        if bad is None or sum(map(len,pack))<sum(map(len,bad[0])): bad=(pack,rows[0],want)
print('first/smallest failing:' , None if bad is None else (len(bad[0]), [len(p) for p in bad[0]]))
if bad:
    for p in bad[0]: print(p.decode())
    print('got'); [print(r.decode()) for r in bad[1]]
    print('want'); [print(r.decode()) for r in bad[2]]
