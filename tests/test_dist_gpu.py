"""GPU suite: the N > 1 path with the HIP sweep as the per-rank worker (BASELINE config 4's partition,
cnF2freq.cpp:5294-5298: analysed individuals are split across ranks, nothing else is exchanged but the
results).  Two ranks share the one GPU of the test box, so the transport is gloo (RCCL refuses two ranks on
one device); the sweep, the sharding and the marker-tiled gather are the code bench.py runs under RCCL."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("tile", [7, 38])
def test_two_rank_hip_sweep_equals_single_rank(tmp_path, tile):
    import __graft_entry__ as g
    g.build()
    from cnf2freq_amd import capi
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_worker
    out = str(tmp_path / "two_rank.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), out, "gloo",
           str(tile)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    z = np.load(out)
    ped = dist_worker.make_ped()
    ctx = capi.Context(0)
    ctx.upload(ped)
    one = ctx.sweep()
    acc = ctx.sweep_accumulate(ctx.descendants())
    ctx.close()
    # bit-equality with the single-rank sweep: sharding and tiling must not change a single value
    assert np.array_equal(z["dosage"], one["dosage"])
    assert np.array_equal(z["loglik"], one["loglik"])
    assert np.array_equal(z["factors"], one["factors"])
    # the accumulators of a haplotyping sweep: per-rank partial sums + one all-reduce = the single-rank sums (the F2
    # founders are ancestors of every individual on both ranks); equal to rounding, the order of the additions differs
    for k in ("infprobs", "haplobase", "haplocount"):
        np.testing.assert_allclose(z["acc_" + k], acc[k], rtol=1e-11, atol=1e-13, equal_nan=True, err_msg=k)
    assert np.abs(acc["infprobs"][0]).max() > 0, "founder A should collect evidence from every F2"
    # the tiles cover the marker axis exactly once, in order
    tiles = z["tiles"]
    assert tiles[0, 0] == 0 and tiles[-1, 1] == ped.n_markers and np.all(tiles[1:, 0] == tiles[:-1, 1])
    # rank 0 never held more than depth (2) tiles x world for receiving
    assert int(z["root_bytes"]) <= 2 * int(z["world"]) * int(z["tile_bytes"])


def _row_blocks(path):
    """{header: [row lines]} of a rows file (cnF2freq.cpp:6183-6188: "name:chrom", the rows, a blank line)."""
    blocks, cur = {}, None
    for line in open(path).read().split("\n"):
        if cur is None:
            if ":" in line and not line.startswith(("FIRST PASS", "SKEWNESS PASS")):
                cur = line
                blocks[cur] = []
        elif line == "":
            cur = None
        else:
            blocks[cur].append(line)
    return blocks


def test_two_rank_iterations_equal_single_rank(tmp_path):
    """BASELINE config 5's multi-GPU leg, correct before it is fast: 2 ranks x 3 haplotyping iterations on an outbred pedigree
    that mixes families private to a rank with one family that straddles the split (sweep of the rank's block, ONE
    reduce-scatter of the SHARED records' accumulators, every rank updating the records it owns, one sum of the hit counters
    per pass, ONE all-gather of the shared records' new rows) = the single-rank run: genotypes and hit counters identical,
    certainties / haplotype weights / haplobase / haplocount to 1e-9, the gathered state bit-identical on both ranks, and the
    rows of the first sweep (same parameters on both sides) equal to the character.  What is exchanged is the six shared
    records, not the slabs.  The runs add their accumulators in a fixed order (CNF2_DETERMINISTIC), so the families that are
    private to a rank -- nothing of them is exchanged, their sums are formed in the single-rank order -- must come out
    BIT-IDENTICAL to the single-rank run; the straddling family's sums are two partial sums added, equal to rounding."""
    import __graft_entry__ as g
    g.build()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_iter_worker
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    worker = os.path.join(ROOT, "tests", "dist_iter_worker.py")
    two, one = str(tmp_path / "two"), str(tmp_path / "one")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), worker, two, "gloo", "3", "1"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    r = subprocess.run([sys.executable, worker, one, "gloo", "3", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    z0, z1, z = np.load(two + "_rank0.npz"), np.load(two + "_rank1.npz"), np.load(one + "_rank0.npz")
    ped = dist_iter_worker.make_ped()
    M, R = ped.n_markers, ped.n_rec
    b0, b1 = z0["block"], z1["block"]
    assert b0[0] == 0 and b0[1] == b1[0] and b1[1] == 18 and 6 < b0[1] < 12, "the split should fall inside family 1"
    # the plan: family 1's two parents and four grandparents are shared, everything else the windows touch is private
    assert z0["n_shared"] == 6 and z1["n_shared"] == 6
    own0, own1 = set(z0["owned"].tolist()), set(z1["owned"].tolist())
    assert not own0 & own1 and len(own0 | own1) == R, "every record of this pedigree is touched and owned exactly once"
    # what travelled: 3 iterations x (6 shared records x (48 + 25) B x M in one reduce-scatter and one all-gather, 2 hit counters)
    assert z0["bytes_payload"] == 6 * (M * 48 + ((M * 25 + 7) // 8) * 8) + 2 * 4
    # 3 reduce-scatters, 3 x 2 passes' hit sums, 3 all-gathers + the state's gather; postmarkerdata ran on rank 0 only and
    # its result came by 2 broadcasts (the rows in one slab, the descendant counts and lock positions)
    assert list(z0["calls"]) == [3, 6, 4, 2]
    assert z0["bytes_buffers"] < 0.5 * R * M * 48, "6 of 36 records are shared here: the buffers are a fraction of the slabs"
    for k in ("allele", "sure", "hw", "hits", "scalefactor"):
        assert np.array_equal(z0[k], z1[k], equal_nan=True), "ranks differ in " + k
    assert np.array_equal(z0["hits"], z["hits"]) and z["hits"].sum() > 0
    assert z0["scalefactor"] == z["scalefactor"]
    # families 0 and 2 (12 records each, private to rank 0 and rank 1): to the bit
    fam = np.arange(R) // 12
    private = fam != 1
    for k in ("allele", "sure", "hw"):
        assert np.array_equal(z0[k][private], z[k][private], equal_nan=True), "private families differ in " + k
    # family 1: a side whose two allele values are tied (certainty 1/2 within the bisection's tolerance) is called by the
    # last bits of its sums in any implementation ("ill-conditioned elements"): left out
    tie = np.abs(z["sure"] - 0.5) < 2e-3
    assert tie[~private].mean() < 0.2
    ok = ~tie
    assert np.array_equal(z0["allele"][ok], z["allele"][ok])
    np.testing.assert_allclose(z0["sure"][ok], z["sure"][ok], rtol=1e-9, atol=1e-12, err_msg="sure")
    np.testing.assert_allclose(z0["hw"], z["hw"], rtol=1e-9, atol=1e-12, err_msg="hw")
    # haplobase / haplocount as left behind, on the rank that owns the record: a slot that is homozygous with EQUAL
    # certainties takes no part in the HAPLOS update (cnF2freq.cpp:1224-1239 compares the two certainties for equality).
    # Where the two sides of an individual collect the same evidence their certainties are equal up to the order of the
    # additions, and one ulp decides whether the slot counts: such elements are left out (their haplotype weights agree
    # all the same, see above).
    knife = (z["allele"][..., 0] == z["allele"][..., 1]) & np.isclose(z["sure"][..., 0], z["sure"][..., 1], rtol=1e-9, atol=0)
    knife |= tie.any(axis=2)
    assert knife.sum() < 0.5 * knife.size
    for zk in (z0, z1):
        own = zk["owned"]
        for k in ("haplobase", "haplocount"):
            np.testing.assert_allclose(zk[k][own][~knife[own]], z[k][own][~knife[own]], rtol=1e-9, atol=1e-12, err_msg=k)
    want = _row_blocks(one + "_rows_it1_rank0.txt")
    got = dict(_row_blocks(two + "_rows_it1_rank0.txt"))
    got1 = _row_blocks(two + "_rows_it1_rank1.txt")
    assert not set(got) & set(got1)
    got.update(got1)
    assert got == want and len(want) == 18 * 2


def test_rccl_exchange_on_one_rank():
    """The nccl (= RCCL) backend with a world of one on GPU 0: the all-reduce of an iteration's exchange on tensors that alias
    the engine's device slabs, and the tiled gather, through the library that carries them on a multi-GPU node (two ranks
    cannot share a GPU under RCCL, so this is what a one-GPU box can run of it)."""
    import subprocess
    import sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "rccl_one_rank_worker.py"), "29581"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "OK " in r.stdout
