import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np
from rattle_amd.api import Context
# chain-graph microbenchmark: identical sequences -> every row's predecessor is the previous row
rng=np.random.default_rng(0)
packs=[]
for p in range(16):
    s=np.frombuffer(b"ACGT",np.uint8)[rng.integers(0,4,1000)].tobytes()
    packs.append([s]*30)
ctx=Context(0)
t=time.time(); rows,width,c=ctx.poa_msa(packs); dt=time.time()-t
tick=1e-8
print('chain: time %.2f GCUPS %.2f | per-row us: topo %.2f dp %.2f | per-aln ms: tb %.2f add %.2f'%(dt, c[0]/dt/1e9, c[4]*tick/c[3]*1e6, c[5]*tick/c[3]*1e6, c[6]*tick/c[1]*1e3, c[7]*tick/c[1]*1e3))
