import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np
from rattle_amd import synth
from rattle_amd.api import Context
seqs, quals, tid, _ = synth.reads(3000, 15, 1, False, seed=4, exon=(50,210))
packs=[]
for g in range(15):
    mem=[seqs[i] for i in range(len(seqs)) if tid[i]==g]; mem.sort(key=lambda s:-len(s))
    if len(mem)>=2: packs.append(mem[:200])
print([len(p) for p in packs])
ctx=Context(0)
t=time.time(); rows,width,c=ctx.poa_msa(packs); dt=time.time()-t
print('time',dt,'cells',c[0],'GCUPS',c[0]/dt/1e9,'rows',c[3])
tick=1e-8
print('ties %d needing sort %d of %d alignments | per-row us: dp %.2f  | per-aln ms: tb %.2f add %.2f'%(int(c[4])>>32, int(c[4])&0xffffffff, c[1], c[5]*tick/c[3]*1e6, c[6]*tick/c[1]*1e3, c[7]*tick/c[1]*1e3))
print('totals s: dp %.2f tb %.2f add %.2f'%(c[5]*tick,c[6]*tick,c[7]*tick))
