import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np
from rattle_amd import synth
from rattle_amd.api import Context
seqs, quals, tid, _ = synth.reads(3000, 15, 1, False, seed=4, exon=(50,210))
packs=[]
for g in range(15):
    mem=[seqs[i] for i in range(len(seqs)) if tid[i]==g]; mem.sort(key=lambda s:-len(s))
    if len(mem)>=2: packs.append(mem[:200])
ctx=Context(0)
rows,width,c=ctx.poa_msa(packs)
c=[int(x) for x in c]
print('pred distance histogram (counts are polluted by the normal counters at [0..3]):')
print(dict(zip(['d1','d2','d3-4','d5-8','d9-16','d17-64','d>64','extra_preds'],c)))
