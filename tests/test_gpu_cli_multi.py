"""GPU suite (-m gpu): `cnF2freq --gpus N` -- the executable forks N ranks (one GPU each; --single-device puts them all on
GPU 0, which is what a one-GPU box can run), the ranks exchange through shared memory (csrc/host/cnf2_shm_transport.h) what
their windows share, and rank 0 writes ONE output in the order of a single-GPU run (the other ranks' rows reach it through
files in --tmppath).  Model: the reference's dead MPI code (cnF2freq.cpp:5297-5299, 6245-6254); the output format is main()'s
(cnF2freq.cpp:6183-6188, 8157-8192)."""
import os
import subprocess

import pytest

from cnf2freq_amd import synth
from conftest import ROOT

pytestmark = pytest.mark.gpu

EXE = os.path.join(ROOT, "cnf2freq_amd", "cnF2freq")


def write_plantimpute(ped, d):
    """map / ped / gen files of a synth.make_outbred3 pedigree (readers: cnF2freq.cpp:6495-6685)."""
    with open(d / "x.map", "w") as f:
        f.write("\n".join("%.10g" % p for p in ped.pos) + "\n")
    with open(d / "x.ped", "w") as f:
        for r in range(ped.n_rec):
            p0, p1 = (ped.names[q] if q >= 0 else "0" for q in ped.par[r])
            f.write("%s %s %s %d\n" % (ped.names[r], p0, p1, ped.gen[r]))
    tok = {(1, 1): "0", (1, 2): "1", (2, 2): "2", (0, 0): "9"}
    with open(d / "x.gen", "w") as f:
        for r in range(ped.n_rec):
            a = ped.allele[ped.row_of[r]]
            f.write(ped.names[r] + " " + " ".join(tok[(int(x[0]), int(x[1]))] for x in a) + "\n")
    return ["--mapfile", str(d / "x.map"), "--pedfile", str(d / "x.ped"), "--genfile", str(d / "x.gen")]


def tokens(path):
    return [line.replace("\t", " ").split() for line in open(path).read().split("\n")]


def row_heads(lines):
    return [l[0] for l in lines if len(l) == 1 and ":" in l[0]]


@pytest.mark.parametrize("shape", ["straddling_family", "private_families"])
def test_two_ranks_write_the_output_of_one(tmp_path, shape):
    import __graft_entry__ as g
    g.build()
    if shape == "straddling_family":
        ped = synth.make_outbred3(3, 6, 13, 2, seed=41, missing=0.15)      # the split falls inside family 1: 6 shared records
    else:
        ped = synth.make_outbred3(6, 3, 13, 2, seed=42, missing=0.15)      # boundaries between families: nothing shared
    files = write_plantimpute(ped, tmp_path)
    spool = tmp_path / "spool"
    spool.mkdir()
    outs = {}
    for name, extra in (("one", []), ("two", ["--gpus", "2", "--single-device", "--tmppath", str(spool)])):
        out = tmp_path / (name + ".txt")
        r = subprocess.run([EXE] + files + ["--output", str(out), "--count", "3", "--quiet"] + extra, capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        outs[name] = (tokens(out), r.stdout, r.stderr)
    assert os.listdir(spool) == [], "rank 0 should have consumed and removed the spooled rows"
    shared = "6 shared records" if shape == "straddling_family" else "0 shared records"
    assert "2 ranks: blocks" in outs["two"][2] and shared in outs["two"][2], outs["two"][2][-600:]
    a, b = outs["one"][0], outs["two"][0]
    assert len(a) == len(b) and len(a) > 1000
    n_num = 0
    for la, lb in zip(a, b):
        assert len(la) == len(lb), (la, lb)
        for x, y in zip(la, lb):
            try:
                fx, fy = float(x), float(y)
            except ValueError:
                assert x == y, (la, lb)                      # headers ("name:chrom", "n name", pass lines): to the character
                continue
            assert abs(fx - fy) <= 2e-5, (la, lb)            # printed with 5 - 6 decimals; the sums differ by rounding
            n_num += 1
    assert n_num > 5000
    # the rows of all 18 analysed individuals on both chromosomes are there, in the order of the single-GPU run
    assert row_heads(b) == row_heads(a) and len(row_heads(a)) == 18 * 2
    # what goes to stdout (the rows of the non-final iteration, the step-size lines) is rank 0's and complete
    sa = [l for l in outs["one"][1].split("\n") if l.startswith("Scale factor")]
    sb = [l for l in outs["two"][1].split("\n") if l.startswith("Scale factor")]
    assert sa == sb and len(sa) == 4
    assert outs["one"][1].count(":1\n") == outs["two"][1].count(":1\n") == 18


def test_two_ranks_print_every_likelihood_line(tmp_path):
    """Without --quiet (the default) a run prints the two likelihood values of every analysed individual and chromosome
    (cnF2freq.cpp:5399-5401) on stdout: with two ranks the lines of rank 1's block reach rank 0 like its rows do, and stdout is
    that of the single-rank run -- same lines, same order, numbers to the printed precision."""
    import __graft_entry__ as g
    g.build()
    ped = synth.make_outbred3(6, 3, 13, 2, seed=44, missing=0.15)
    files = write_plantimpute(ped, tmp_path)
    spool = tmp_path / "spool"
    spool.mkdir()
    outs = {}
    for name, extra in (("one", []), ("two", ["--gpus", "2", "--single-device", "--tmppath", str(spool)])):
        r = subprocess.run([EXE] + files + ["--output", str(tmp_path / (name + ".txt")), "--count", "3"] + extra, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        outs[name] = [l.replace("\t", " ").split() for l in r.stdout.split("\n")]
    assert os.listdir(spool) == []
    a, b = outs["one"], outs["two"]
    lik = lambda lines: [l for l in lines if len(l) == 4 and l[0].endswith(":") and l[0].count(",") == 2]
    assert len(lik(a)) == 18 * 2 * 2                      # 18 analysed individuals x 2 chromosomes x the 2 iterations that sweep
    assert [l[0] for l in lik(b)] == [l[0] for l in lik(a)]
    assert len(a) == len(b)
    for la, lb in zip(a, b):
        assert len(la) == len(lb), (la, lb)
        for x, y in zip(la, lb):
            try:
                fx, fy = float(x.rstrip(":")), float(y.rstrip(":"))
            except ValueError:
                assert x == y, (la, lb)
                continue
            assert abs(fx - fy) <= 2e-5 * max(1.0, abs(fx)), (la, lb)


def test_a_failing_rank_ends_the_run(tmp_path):
    """More ranks than GPUs without --single-device: the rank that finds no device ends, the parent stops the others (which
    would wait at a barrier for ever) and the run aborts like every failure of the reference (cnF2freq.cpp:21-25)."""
    import torch
    if torch.cuda.device_count() > 3:
        pytest.skip("needs a box with fewer than 4 GPUs")
    ped = synth.make_outbred3(4, 2, 7, 1, seed=43, missing=0.1)
    files = write_plantimpute(ped, tmp_path)
    r = subprocess.run([EXE] + files + ["--output", str(tmp_path / "o.txt"), "--count", "2", "--quiet", "--gpus", "4", "--tmppath",
                        str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "a rank ended abnormally" in r.stderr and "needs N GPUs" in r.stderr


def test_rccl_transport_of_the_executable_with_a_world_of_one():
    """`cnF2freq --gpus N` exchanges through RCCL, in place on the context's exchange buffer, when every rank has a GPU of its own
    (csrc/host/cnf2_rccl_transport.h; the reference's reduce per individual, cnF2freq.cpp:6245-6254).  A one-GPU box can run
    its collectives with a world of one: the communicator's set-up through the shared region, ncclReduceScatter and ncclAllGather
    on the device buffer through the entry the engine calls, the host-side operations behind them."""
    import __graft_entry__ as g
    g.build()
    r = subprocess.run([EXE, "--rccl-selftest"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rccl selftest: ok" in r.stdout, r.stdout[-1000:] + r.stderr[-2000:]
