"""ONE job over several ranks with the real kernels: two (and three) processes share the box's single GPU, the
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


@pytest.mark.parametrize("world,transport", [(2, "host"), (3, "host"), (2, "device"), (3, "device")])
def test_sharded_job_equals_single_gpu(tmp_path, world, transport):
    """transport "host": the caller's all-gather-v on host buffers (gloo); "device": the RCCL code path of exchange.hip
    (sizes all-gather, grouped broadcasts, grouped send / recv to the root) with the file-backed double standing in
    for librccl.so, since RCCL itself refuses two ranks on one device."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, RATTLE_ROOT=ROOT, MASTER_ADDR="127.0.0.1", RATTLE_HOST_THREADS="8")
    env.pop("RATTLE_RCCL_LIB", None)
    if transport == "device":
        env.update(fake_rccl(tmp_path))
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
