"""The N>1 path of bench.py on CPU: 2 ranks (gloo) shard the reads, each produces its assignment
array, all_gather reassembles it, timing is the max over ranks.  No compute calls (no GPU here):
the per-rank work is replaced by a deterministic stub with the same tensor shapes."""
import os
import subprocess
import sys

from conftest import ROOT

SCRIPT = r"""
import os, sys, time
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ['RATTLE_ROOT'])
from rattle_amd import synth
dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
seqs, quals, tid, _ = synth.reads(200, 4, 1, True, seed=20260929 + rank, exon=(50, 210))
assign = torch.from_numpy(tid.astype(np.int32))           # stand-in for the per-read cluster id
parts = [torch.empty_like(assign) for _ in range(world)]
dist.all_gather(parts, assign)
t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
if rank == 0:
    # every rank's shard is different (seed + rank) and arrives in rank order
    other, _, tid1, _ = synth.reads(200, 4, 1, True, seed=20260929 + 1, exon=(50, 210))
    assert np.array_equal(parts[1].numpy(), tid1.astype(np.int32))
    assert abs(float(t) - 0.2) < 1e-12
    print('DIST_OK', len(parts), int(parts[0].numel()))
dist.destroy_process_group()
"""


def test_two_rank_gloo_allgather(tmp_path):
    p = tmp_path / "w.py"
    p.write_text(SCRIPT)
    env = dict(os.environ, RATTLE_ROOT=ROOT)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29531", str(p)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert "DIST_OK 2 200" in out.stdout, out.stdout + out.stderr
