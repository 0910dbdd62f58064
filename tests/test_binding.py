"""the reference-side binding (integration/rattle_binding.hpp = bodies for cluster_reads,
cluster.hpp:44, and correct_reads, correct.hpp:44) compiles as plain C++14 against restated type declarations,
links librattle_hip.so, and -- on a GPU -- produces exactly what the Python mirror produces."""
import os
import subprocess

import pytest

from conftest import ROOT
from rattle_amd import synth

CSRC = os.path.join(ROOT, "rattle_amd", "csrc")


def build_driver(tmp_path):
    if not os.path.exists(os.path.join(CSRC, "librattle_hip.so")):
        subprocess.check_call(["make", "-s", "-j4", "-C", CSRC])
    exe = tmp_path / "binding_driver"
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "binding"),
                           os.path.join(ROOT, "tests", "binding", "driver.cpp"), "-o", str(exe), "-L", CSRC, "-lrattle_hip", f"-Wl,-rpath,{CSRC}"])
    return exe


def test_binding_compiles_and_links(tmp_path):
    exe = build_driver(tmp_path)
    assert subprocess.run([str(exe)], capture_output=True).returncode == 2          # usage exit: the binary loads and runs without a device


@pytest.mark.gpu
@pytest.mark.parametrize("is_rna", [False, True])
def test_binding_matches_python_mirror(gpu_ctx, tmp_path, is_rna):
    from rattle_amd.api import cluster_command, correct_command
    exe = build_driver(tmp_path)
    seqs, quals, _, _ = synth.reads(600, 6, 1, not is_rna, seed=41)
    fq = tmp_path / "in.fastq"
    fq.write_bytes(synth.fastq_text(seqs, quals))
    r = subprocess.run([str(exe), str(fq), "1" if is_rna else "0", "40", str(tmp_path / "b")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    headers = [b"@r%d" % i for i in range(len(seqs))]
    clusters, _ = cluster_command(gpu_ctx, seqs, list(range(len(seqs))), is_rna=is_rna)
    want_txt = "".join("%d:%d |%s\n" % (m[0], m[1], "".join(" %d:%d" % (s[0], s[1]) for s in mem)) for m, mem in clusters)
    assert (tmp_path / "b.clusters.txt").read_text() == want_txt
    want = correct_command(gpu_ctx, headers, seqs, quals, clusters, split=40, ann=[b"+"] * len(seqs))
    assert (tmp_path / "b.corrected.fq").read_bytes() == want[0]
    assert (tmp_path / "b.uncorrected.fq").read_bytes() == want[1]
    assert (tmp_path / "b.consensi.fq").read_bytes() == want[2]
    assert any(len(mem) > 40 for _, mem in clusters)
