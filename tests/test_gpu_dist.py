"""ONE job over several ranks with the real kernels: two, three and eight processes share the box's single GPU, the
exchange goes through host buffers (gloo) or through exchange.hip's RCCL call path over a file-backed double of
librccl.so (RCCL wants one GPU per rank, which only the driver's 8-GPU run provides), and every sharded result must equal the unsharded one bit for bit: gene-level clusters (candidate axis),
`--iso` transcript clusters (gene axis), and the three outputs of `correct` (pack axis, reassembled on rank 0).
The RCCL transport itself is exercised with a world of one rank."""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    sys.path.insert(0, os.environ["RATTLE_ROOT"])
    import torch.distributed as dist
    from rattle_amd import synth
    from rattle_amd.api import Context
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    cat, qcat, off, tid, _ = synth.reads_packed(4000, 14, 3, True, seed=21, exon=(50, 210))
    ref = None
    if rank == 0:                                  # unsharded reference, same process, own context
        c0 = Context(0)
        cl0 = c0.cluster_unsorted_packed(cat, off)
        r0 = c0.correct_packed(cat, qcat, off, cl0, split=40, digest=True)
        iso0, gid0, ng0 = c0.cluster_iso_unsorted_packed(cat, off)
        ref = (cl0.as_list(), r0, iso0.as_list(), list(gid0), ng0, [int(x) for x in cl0.counters[:3]])
        c0.close()
    dist.barrier()
    ctx = Context(0)
    if os.environ.get("RATTLE_RCCL_LIB"):
        ctx.comm_init_rccl()                       # the device-buffer transport (RCCL call pattern) over the tests' double
    else:
        ctx.set_exchange_gloo()
    ctx.comm_probe()
    cl = ctx.cluster_unsorted_packed(cat, off)
    res = ctx.correct_packed(cat, qcat, off, cl, split=40, digest=True, gather_root=0)
    iso, gid, ng = ctx.cluster_iso_unsorted_packed(cat, off)
    calls, nbytes = ctx.comm_stats()
    assert calls > 0 and nbytes > 0
    if rank == 0:
        assert cl.as_list() == ref[0], "sharded gene-level clusters differ"
        assert [int(x) for x in cl.counters[:3]] == ref[5], "work counters differ"          # pair tests, comparisons, matches: summed over ranks
        assert res[:3] == ref[1][:3] and res[4] == ref[1][4], ("sharded correct differs", res, ref[1])
        assert int(res[3][0]) == int(ref[1][3][0]) and int(res[3][1]) == int(ref[1][3][1])     # DP cells and alignments add up
        assert iso.as_list() == ref[2] and list(gid) == ref[3] and ng == ref[4], "sharded --iso clusters differ"
        assert any(len(m) > 80 for _, m in ref[0])                                            # multi-pack clusters (POA #3 over exchanged consensi)
        print("GPU_DIST_OK", world, len(ref[0]), len(ref[2]), res[0])
    else:
        assert res[0] == 0 and res[4] is None
        assert cl.as_list() is not None
    ctx.close()
    dist.destroy_process_group()
''')


def fake_rccl(tmp_path):
    """tests/stubs/fake_rccl.cpp built into the test's directory: librccl's entry points over files, several ranks per GPU."""
    so = tmp_path / "libfake_rccl.so"
    r = subprocess.run(["g++", "-O1", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "stubs", "fake_rccl.cpp"),
                        "-o", str(so), "-L/opt/rocm/lib", "-lamdhip64"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    box = tmp_path / "mailbox"
    box.mkdir()
    return {"RATTLE_RCCL_LIB": str(so), "FAKE_RCCL_DIR": str(box)}


@pytest.mark.parametrize("world,transport,big", [(2, "host", 0), (3, "host", 1), (2, "device", 1), (3, "device", 0), (8, "host", 1), (8, "device", 0)])
def test_sharded_job_equals_single_gpu(tmp_path, world, transport, big):
    """transport "host": the caller's all-gather-v on host buffers (gloo); "device": the RCCL code path of exchange.hip
    (sizes all-gather, grouped broadcasts, grouped send / recv to the root) with the file-backed double standing in
    for librccl.so, since RCCL itself refuses two ranks on one device."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, RATTLE_ROOT=ROOT, MASTER_ADDR="127.0.0.1", RATTLE_HOST_THREADS="8")
    env.pop("RATTLE_RCCL_LIB", None)
    if transport == "device":
        env.update(fake_rccl(tmp_path))
    if big:
        # the big-cluster flow of `correct` on every rank: POA #2 of the many-pack clusters first (2a), all-gather, their POA #3 beside
        # everybody else's POA #2 (correct_driver.hip); the unsharded reference takes the same path
        env.update(RATTLE_BIG_CLUSTER_PACKS="3", RATTLE_BIG_MIN_PACKS="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                        "--master-port", str(29700 + world + (10 if transport == "device" else 0)), str(script)], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert f"GPU_DIST_OK {world}" in r.stdout


def test_rccl_transport_world_of_one(gpu_ctx):
    """rattle_hip_comm_init with nranks = 1: librccl.so loads, the communicator comes up on this device and goes away."""
    import ctypes as C
    import numpy as np
    uid = np.zeros(128, np.uint8)
    lib = gpu_ctx.lib
    from rattle_amd._lib import check
    check(lib.rattle_hip_comm_unique_id(uid.ctypes.data_as(C.POINTER(C.c_uint8))))
    assert uid.any()
    check(lib.rattle_hip_comm_init(gpu_ctx.h, 0, 1, uid.ctypes.data_as(C.POINTER(C.c_uint8))))
    try:
        assert lib.rattle_hip_comm_init(gpu_ctx.h, 0, 1, uid.ctypes.data_as(C.POINTER(C.c_uint8))) != 0      # already attached
    finally:
        check(lib.rattle_hip_comm_destroy(gpu_ctx.h))


def test_bench_sharded_on_one_device(tmp_path):
    """bench.py's N > 1 path end to end (strong scaling, reference digest, gather on rank 0) with two ranks on the one GPU."""
    import json
    env = dict(os.environ, RATTLE_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29711",
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--reads", "20000", "--no-cpu-baseline", "--transport", "gloo"],
                       capture_output=True, text=True, env=env, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["checks"]["digest_equal_to_single_gpu"] and line["checks"]["digest_equal_across_steps"]
    assert line["checks"]["n_corrected"] + line["checks"]["n_uncorrected"] == 20000


def test_bench_gpus_flag_launches_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it (the driver's command shape) must become two ranks:
    n_gpus == 2 in the line, the exchange reports two ranks, digest == the single-GPU digest."""
    import json
    env = dict(os.environ, RATTLE_BENCH_ONE_DEVICE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--reads", "20000", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["exchange"]["ranks"] == 2 and line["exchange"]["collectives"] > 0
    assert line["checks"]["digest_equal_to_single_gpu"] and line["checks"]["n_corrected"] + line["checks"]["n_uncorrected"] == 20000


def test_bench_eight_ranks_on_one_device():
    """The driver's 8-GPU command shape (`python bench.py --gpus 8`, BASELINE configs[3]) run once with the eight ranks sharing the
    box's one GPU: the LPT partition with eight bins, pack_owner, the all-gathers of every greedy round and consensus stage and
    the gather order all execute, and the sharded result must carry the single-GPU digest.  (No scaling figure: the ranks
    compete for one device.)"""
    import json
    env = dict(os.environ, RATTLE_BENCH_ONE_DEVICE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1", "--reads", "100000", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=1800, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["exchange"]["ranks"] == 8 and line["exchange"]["collectives"] > 0
    assert line["checks"]["digest_equal_to_single_gpu"] and line["checks"]["digest_equal_across_steps"]
    assert line["checks"]["n_corrected"] + line["checks"]["n_uncorrected"] == 100000


FAIL_WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    sys.path.insert(0, os.environ["RATTLE_ROOT"]); sys.path.insert(0, os.path.join(os.environ["RATTLE_ROOT"], "tests"))
    import torch.distributed as dist
    from rattle_amd import synth
    from rattle_amd.api import Context
    from test_dist_cpu import _plan
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    cat, qcat, off, tid, _ = synth.reads_packed(1500, 6, 1, True, seed=5, exon=(50, 210))
    ctx = Context(0)
    ctx.set_exchange_gloo()
    cl = ctx.cluster_unsorted_packed(cat, off)
    plan = _plan(off, cl.offsets, cl.member_id, cl.member_rev, world, split=40)
    victim = int(np.nonzero(plan["owner"] == world - 1)[0][0])          # a pack of the LAST rank only
    bad = cat.copy()
    bad[int(off[plan["member"][plan["first"][victim]]]) + 3] = ord("N")
    try:
        ctx.correct_packed(bad, qcat, off, cl, split=40, gather_root=0)
        print("NO_ERROR", rank)
    except RuntimeError as e:
        msg = str(e)
        assert ("base other than" in msg) == (rank == world - 1), msg
        assert rank == world - 1 or "failed on rank %d" % (world - 1) in msg, msg
        print("FAIL_OK", rank)
    # the ranks are still in step: a good job right after the failed one
    res = ctx.correct_packed(cat, qcat, off, cl, split=40, gather_root=0)
    if rank == 0:
        assert res[0] + res[1] == 1500
        print("THEN_OK")
    ctx.close()
    dist.destroy_process_group()
''')


@pytest.mark.parametrize("world", [2, 3, 8])
def test_local_failure_reaches_every_rank_instead_of_hanging(tmp_path, world):
    """A bad base in a read that only the last rank's packs hold: that rank must keep joining the exchanges
    (correct_driver.hip: failure record) so that every rank returns an error -- none is left in an all-gather."""
    script = tmp_path / "worker.py"
    script.write_text(FAIL_WORKER)
    env = dict(os.environ, RATTLE_ROOT=ROOT, MASTER_ADDR="127.0.0.1", RATTLE_HOST_THREADS="8")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                        "--master-port", str(29730 + world), str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("FAIL_OK") == world and "THEN_OK" in r.stdout and "NO_ERROR" not in r.stdout


def test_rank_replay_reproduces_the_single_rank_result(tmp_path):
    """tools/rank_replay.py (the multi-GPU curve measured on one GPU): the single-rank job records what it holds at every exchange
    point, then every rank of a world of three runs its own share ALONE with its peers' payloads from that record (replay transport
    of exchange.hip) -- the root's merged result must be the single-rank result, byte for byte (digest over every output array)."""
    out = tmp_path / "replay.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rank_replay.py"), "--reads", "4000", "--world", "3", "--record", str(tmp_path / "rec.bin"),
                        "--out", str(out)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    import json
    res = json.loads(out.read_text())
    assert res["root_result_equals_single_rank"] and len(res["ranks"]) == 3
