import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
from rattle_amd.api import Context
ctx=Context(0)
pack=[b"TACCCGGGTTAGCTGACCCTT", b"TACCTGAGAGTTAGCTCACCCTTT"]
rows,width,c=ctx.poa_msa([pack])
for r in rows[0]: print(r.decode())
