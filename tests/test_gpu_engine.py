"""GPU suite: what surrounds the sweep in a cnF2freq run, through libcnf2host.so
(include/cnf2host.h, the engine the `cnF2freq` executable links) and the update entry points of libcnf2hip.so:
  * postmarkerdata (fixkid / fixparents inference, descendants, variances, lockhaplos) against the reference's own
    postmarkerdata run on the same inputs (goldens G12, tests/golden/make_golden.py);
  * the per-iteration updates on the device against the oracle's literal restatement of processinfprobs /
    updatehaploweights (parity unpinned there: Boost's quadrature is absent from the image);
  * whole iterations: parameters move, the dump round-trips through --deserialize, the demo files of the reference."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from cnf2freq_amd import synth
from conftest import GOLDEN_CASES, ROOT, TRAJ_CASES, TRAJ_CASES_LONG, TrajectoryChecker, load_golden, load_trajectory, oracle_ped

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def libs():
    import __graft_entry__ as g
    g.build()
    from cnf2freq_amd import capi, host
    assert capi.load().cnf2_device_count() >= 1
    host.load()
    return capi, host


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_postmarkerdata_matches_reference(libs, case):
    capi, host = libs
    ped, z = load_golden(case)
    run = host.Run(ped)
    run.postmarkerdata()
    st = run.state()
    assert np.array_equal(st["descendants"], z["pm_descendants"])
    assert np.array_equal(st["children"], z["pm_children"])
    assert np.array_equal(st["allele"], z["pm_allele"])
    np.testing.assert_allclose(st["sure"], z["pm_sure"], rtol=1e-12, atol=0)
    # a variance is a squared difference of two nearly equal sums: rounding is amplified by the cancellation
    np.testing.assert_allclose(st["variances"], z["pm_variances"], rtol=1e-8, atol=1e-16)
    # lockhaplos locks the first marker of strictly largest variance per chromosome (cnF2freq.cpp:3058-3065).  The closed-form
    # variances agree with the reference's to 1e-9, not to the bit, and mirror-image configurations tie but for the
    # reference's own rounding: the markers that can win are compared on cnf2_variances_exact's values -- the reference's
    # bits -- so every record locks the reference's marker
    h0 = ped.dense()[2]
    for r in range(ped.n_rec):
        got_m, want_m = np.flatnonzero(st["hw"][r] != h0[r]), np.flatnonzero(z["pm_hw"][r] != h0[r])
        assert np.array_equal(got_m, want_m), r
        assert np.array_equal(st["hw"][r][got_m], np.where(h0[r][got_m] <= 0.5, 0.0, 1.0))
    assert (st["hw"] != h0).any(), "the fixture should lock some haplotype weights"
    run.close()


def test_fixparents_scan_against_oracle_emission(libs):
    """cnf2_fixparents_scan: 'is any (state, path of parity b) possible' under shift mode 0 with
    CORRECTIONINFERENCE set and no founder flags, against the oracle's path-resolved emission."""
    capi, _ = libs
    from oracle.pyoracle import OraclePed
    ped = synth.make_random_windows(12, 4, seed=5)
    ped.sure = ped.sure.copy()
    ped.sure[ped.sure < 0.015] = 0.0                       # exact zeros make some parities impossible
    ctx = capi.Context(0)
    ctx.upload(ped)
    recs = np.arange(ped.n_rec, dtype=np.int32)
    got = ctx.fixparents_scan(recs)
    a, s, h = ped.dense()
    o = OraclePed(a, s, h, ped.par, ped.empty, ped.pos, founder=np.zeros(ped.n_rec, np.uint8), correction_inference=1,
                  apply_founder_flags=False)
    seen = set()
    for r in range(ped.n_rec):
        for m in range(ped.n_markers):
            for b in range(2):
                want = any(o.emission(r, m, i, f2, 0) != 0.0 for i in range(64) for f2 in range(b, 128, 2))
                assert bool(got[r, m, b]) == want
                seen.add(want)
    assert seen == {True, False}
    ctx.close()


def test_closed_form_variances_equal_the_brute_force_kernel(libs):
    """cnf2_variances (closed form, every record) against the same entry point in brute-force mode (the reference's
    65 536 emission calls per marker) and the per-individual hook, both founder-flag conventions."""
    capi, _ = libs
    for ped in (synth.make_random_windows(20, 5, seed=17), synth.make_outbred3(2, 2, 9, 1, seed=3, missing=0.3, random_sure=True)):
        ctx = capi.Context(0)
        ctx.upload(ped)
        recs = np.arange(ped.n_rec, dtype=np.int32)
        for ordered in (True, False):
            a = ctx.variances(recs, ordered=ordered)
            b = ctx.variances(recs, ordered=ordered, brute=True)
            assert np.array_equal(np.isnan(a), np.isnan(b))
            ok = ~np.isnan(b)
            np.testing.assert_allclose(a[ok], b[ok], rtol=1e-8, atol=1e-18)
        full = ctx.variances(ped.dous, ordered=False)
        for j in range(len(ped.dous)):
            hook = ctx.addvariance(j, 0)
            ok = ~np.isnan(hook)
            np.testing.assert_allclose(full[j][ok], hook[ok], rtol=1e-8, atol=1e-18)
        ctx.close()


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_variances_exact_has_the_references_bits(libs, case):
    """cnf2_variances_exact -- addvariance's additions in the reference's order, no operation contracted -- for EVERY record and
    marker of goldens G10 against the reference's own variances[] (oracle/_ref): equal to the bit, where the closed form
    agrees to 1e-9.  It is what lockhaplos' comparison is made on where more than one configuration can win."""
    capi, _ = libs
    ped, z = load_golden(case)
    ctx = capi.Context(0)
    ctx.upload(ped)
    recs = np.repeat(np.arange(ped.n_rec, dtype=np.int32), ped.n_markers)
    markers = np.tile(np.arange(ped.n_markers, dtype=np.int32), ped.n_rec)
    got = ctx.variances_exact(recs, markers, ordered=False).reshape(ped.n_rec, ped.n_markers)
    want = z["variances"]
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want)
    assert ok.sum() > 100 and np.array_equal(got[ok], want[ok]), "%d of %d entries differ from the reference's bits" % ((got[ok] != want[ok]).sum(), ok.sum())
    closed = ctx.variances(np.arange(ped.n_rec, dtype=np.int32), ordered=False)
    assert (closed[ok] != want[ok]).any(), "the closed form has the reference's bits everywhere: the case proves nothing"
    np.testing.assert_allclose(closed[ok], want[ok], rtol=1e-8, atol=1e-18)
    with pytest.raises(RuntimeError):
        ctx.variances_exact(recs[:1], np.array([ped.n_markers], np.int32))
    ctx.close()


def _oracle_update(ped, acc, children, desc, chrom, scalefactor, allele, sure, hw, prior_allele, prior_sure, has_prior):
    """processinfprobs for the markers of `chrom`, updatehaploweights for chromosomes <= chrom, per record, through
    the oracle's literal restatement (cnf2_oracle_iter.c).  Arrays are per record and modified in place."""
    from oracle import pyiter, pyoracle
    return pyiter.update_pass(pyoracle.lib(), ped.chromstarts, ped.empty, acc, children, desc, chrom, scalefactor, allele, sure,
                              hw, prior_allele, prior_sure, has_prior)


@pytest.mark.parametrize("maker", [
    lambda: synth.make_outbred3(2, 3, 9, 2, seed=3, missing=0.2),
    lambda: synth.make_f2(6, 8, 2, seed=9, chrom_cm=30.0, missing=0.15),
    lambda: synth.make_ail(4, 6, 3, 7, 1, seed=5, chrom_cm=20.0, missing=0.05),
])
def test_update_pass_matches_oracle(libs, maker):
    """One full haplotyping sweep on the GPU, then the update passes of every chromosome on the GPU against the
    oracle's processinfprobs / updatehaploweights fed with the same accumulators: new genotypes, certainties,
    haplotype weights, haplobase / haplocount as left behind, and the hit counter."""
    capi, _ = libs
    ped = maker()
    # one row per record (updates write rows in place)
    a, s, h = ped.dense()
    ped.allele, ped.sure, ped.hw = np.concatenate([a[:1] * 0, a]).astype(np.uint8), np.concatenate([s[:1] * 0, s]), \
        np.concatenate([h[:1] * 0 + 0.5, h])
    ped.row_of = np.arange(1, ped.n_rec + 1, dtype=np.int32)
    rs = np.random.RandomState(1)
    ped.hw[1:] = np.where(rs.rand(*ped.hw[1:].shape) < 0.2, 0.5, 0.1 + 0.8 * rs.rand(*ped.hw[1:].shape))
    ped.hw[1:][ped.empty == 1] = 0.5
    ctx = capi.Context(0)
    ctx.upload(ped)
    has_prior = (1 - ped.empty).astype(np.uint8)
    ctx.snapshot_priors(has_prior)
    desc = ctx.descendants()
    children = np.zeros(ped.n_rec, np.int32)
    for r in ped.dous:
        for k in range(2):
            if ped.par[r, k] >= 0:
                children[ped.par[r, k]] += 1
    acc = ctx.sweep_accumulate(desc)
    want_acc = {k: acc[k].copy() for k in ("infprobs", "haplobase", "haplocount")}
    allele = ped.allele[1:].astype(np.int32).copy()
    sure, hw = ped.sure[1:].copy(), ped.hw[1:].copy()
    prior_allele, prior_sure = allele.copy(), sure.copy()
    moved = 0
    for c in range(len(ped.chromstarts) - 1):
        hits = ctx.update_pass(c, children, desc, 0.013, 1.0, acc)
        want_hits = _oracle_update(ped, want_acc, children, desc, c, 0.013, allele, sure, hw, prior_allele, prior_sure,
                                   has_prior)
        assert hits == want_hits
        ga, gs, gh = ctx.download_rows(1, ped.n_rec)
        assert np.array_equal(ga, allele)
        np.testing.assert_allclose(gs, sure, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(gh, hw, rtol=1e-9, atol=1e-12)
        for k in ("haplobase", "haplocount"):
            np.testing.assert_allclose(acc[k], want_acc[k], rtol=1e-9, atol=1e-12)
        assert not acc["infprobs"][:, ped.chromstarts[c]:ped.chromstarts[c + 1]].any()
        moved += int(np.abs(gh - ped.hw[1:]).max() > 1e-6)
    assert moved > 0
    ctx.close()


@pytest.mark.parametrize("case", TRAJ_CASES + TRAJ_CASES_LONG)
def test_iterations_follow_the_reference_trajectory(libs, case, tmp_path):
    """G13: main()'s sequence readers -> postmarkerdata -> 3 x doit (cnF2freq.cpp:8083-8136) through cnf2h_postmarkerdata /
    cnf2h_iteration against the reference's own replay of it (oracle/_ref: cnF2freq.cpp:4004-4734 verbatim around
    ref_driver.inc's ref_iteration): genotypes identical, certainties / haplotype weights / haplobase / haplocount to 1e-9,
    the hit counter of every chromosome's pass and the scale factor identical, after each of the three iterations (ten for
    outbred3_long, whose scale factor goes through the "good" and the "bad" branch of the step-size control,
    cnF2freq.cpp:6373-6392).  Elements whose result is rounding noise in the reference itself are excused with their
    pedigree component (tests/conftest.py: TrajectoryChecker); how many records were compared after every iteration goes
    to gpurun_out/trajectory_compared.txt."""
    capi, host = libs
    ped, z, n_iter = load_trajectory(case)
    run = host.Run(ped)
    run.postmarkerdata()
    st = run.state()
    assert np.array_equal(st["allele"], z["pm_allele"])
    np.testing.assert_allclose(st["sure"], z["pm_sure"], rtol=1e-12, atol=0)
    chk = TrajectoryChecker(ped, z)
    # lockhaplos locks the first marker of strictly largest variance (cnF2freq.cpp:3058-3065); where mirror-image
    # configurations tie, the reference's own last bits decide, and so do they here (cnf2_variances_exact, see
    # test_postmarkerdata_matches_reference): EVERY record starts on the reference's marker.  (Round 4 had 22 of
    # outbred3_long's 100 records start apart and continued from the reference's deserialised state instead.)
    assert np.array_equal(st["hw"], z["pm_hw"]), "%d records lock another marker than the reference" % (st["hw"] != z["pm_hw"]).any(axis=1).sum()
    if case in TRAJ_CASES_LONG:
        sf = [float(z["it%d_scalefactor" % k]) for k in range(1, n_iter + 1)]
        moves = np.diff(np.log([0.013] + sf))
        assert n_iter == 10 and (moves > 0.05).any() and (moves < -0.05).any(), "the golden should grow and shrink the step size"
    compared = 0
    for k in range(1, n_iter + 1):
        run.iteration()
        st = run.state()
        ps = run.passes()
        st.update(hits=ps["hits"], haplobase=ps["haplobase"], haplocount=ps["haplocount"])
        # certainties are 1 - p with p up to 0.9996: an error of 5e-12 in p is 1e-8 of a certainty of 4e-4, so the bound on
        # probabilities is absolute (1e-10) next to the relative 1e-9
        if case in TRAJ_CASES_LONG:
            # two runs that differ by rounding stay together element by element for four iterations (measured on this golden,
            # tools/diag_trajectory.py: 2e-15, 1e-10, 8e-9, 4e-8 relative; 1.5e-6 after the fifth), then a handful of haplotype
            # weights drifts (16 - 18 of 20 200 beyond 1e-6 from iteration 8 on, after a knife-edge decision has moved one
            # haplobase by 1.5); the step-size control, EVERY hit counter and every genotype stay those of the reference
            # through all ten iterations, and so do all certainties to 2e-7
            compared = chk.check(k, st, rtol=(1e-9, 1e-9, 1e-8, 1e-6)[min(k, 4) - 1], atol=1e-10, hits_slack=0, values=k <= 4)
            np.testing.assert_allclose(st["sure"], z["it%d_sure" % k], rtol=0, atol=1e-6)
            assert (np.abs(st["hw"] - z["it%d_hw" % k]) > 1e-6).sum() <= 40
        else:
            compared = chk.check(k, st, rtol=1e-9, atol=1e-10)
    print(chk.report(case, os.path.join(ROOT, "gpurun_out", "trajectory_compared.txt")))
    assert compared == ped.n_rec or case == "random_windows"
    assert compared >= ped.n_rec // 2
    run.close()


def test_flow_kernels_equal_one_thread_per_element(libs):
    """cnf2_update_pass: the scout / finish kernels (root memo, bounds that spare quadratures, flows replayed from their
    decisions) against CNF2_UPDATE_PLAIN (one thread per record x marker, the literal bisection with every quadrature):
    the same decisions per flow, so the same bits -- over four rounds of sweep + update passes at a large scale factor,
    through which the state moves from "everything far from equilibrium" towards the steady state of a run."""
    capi, _ = libs
    ped = synth.make_outbred3(5, 4, 37, 2, seed=8, missing=0.2)
    a, s, h = ped.dense()
    ped.allele, ped.sure, ped.hw = np.concatenate([a[:1] * 0, a]).astype(np.uint8), np.concatenate([s[:1] * 0, s]), \
        np.concatenate([h[:1] * 0 + 0.5, h])
    ped.row_of = np.arange(1, ped.n_rec + 1, dtype=np.int32)
    rs = np.random.RandomState(2)
    ped.hw[1:] = np.where(rs.rand(*ped.hw[1:].shape) < 0.2, 0.5, 0.05 + 0.9 * rs.rand(*ped.hw[1:].shape))
    # a few alleles carry the sentinel 9 (neither 1 nor 2): on such a side the two values' flows are not mirror images
    # (both start at the same certainty) and the mirrored form has to run them both
    nine = rs.rand(*ped.allele[1:].shape) < 0.01
    ped.allele[1:][nine & (ped.allele[1:] != 0)] = 9
    ctxs = {}
    for name in ("mirror", "flow", "flow1", "flowL", "plain"):
        ctx = capi.Context(0)
        ctx.upload(ped)
        ctx.snapshot_priors((1 - ped.empty).astype(np.uint8))
        ctxs[name] = ctx
    desc = ctxs["flow"].descendants()
    children = np.zeros(ped.n_rec, np.int32)
    for r in ped.dous:
        for k in range(2):
            if ped.par[r, k] >= 0:
                children[ped.par[r, k]] += 1
    total_hits = 0
    for rnd in range(4):
        out = {}
        # mirror: the product's kernels (one flow per side where both values have evidence, the other its mirror image);
        # flow: both values' flows as the reference runs them (CNF2_UPDATE_BOTH_FLOWS), the certainties' scout in two passes;
        # flow1: that scout in one pass (CNF2_UPDATE_ONE_SCOUT).  The forms are flags of the call: nothing reads the environment
        for name, flags in (("mirror", 0), ("flow", capi.UPDATE_BOTH_FLOWS),
                            ("flow1", capi.UPDATE_BOTH_FLOWS | capi.UPDATE_ONE_SCOUT),
                            ("flowL", capi.UPDATE_BOTH_FLOWS | capi.UPDATE_LITERAL_FINISH), ("plain", capi.UPDATE_PLAIN)):
            ctx = ctxs[name]
            acc = ctx.sweep_accumulate(desc, deterministic=True)
            hits = [ctx.update_pass(c, children, desc, 0.19, 1.0, acc, flags=flags) for c in range(2)]
            out[name] = (hits, ctx.download_rows(1, ped.n_rec), {k: acc[k].copy() for k in ("infprobs", "haplobase", "haplocount")})
        total_hits += sum(out["flow"][0])
        # flowL: the set-aside flows one literal step per round (rounds 3 / 4) instead of the guided bisection (the default)
        for name in ("flow", "flow1", "flowL"):
            assert out[name][0] == out["plain"][0], (rnd, name)
            for x, y in zip(out[name][1], out["plain"][1]):
                assert np.array_equal(x, y), (rnd, name)
            for k in ("infprobs", "haplobase", "haplocount"):
                assert np.array_equal(out[name][2][k], out["plain"][2][k], equal_nan=True), (rnd, name, k)
        # the mirrored form: from the same state the same capped moves and values within rounding of the literal ones.  (Its
        # context is put back on the literal rows after every round: a flow's result is a bisection midpoint, so an input
        # that differs in its last bit can end a bisection a step earlier or later and move the result by the width of the
        # tolerance band -- states that differ by rounding drift apart over the rounds, as two runs of the reference do.)
        assert out["mirror"][0] == out["plain"][0], rnd
        for x, y in zip(out["mirror"][1], out["plain"][1]):
            if x.dtype.kind == "f":
                np.testing.assert_allclose(x, y, rtol=1e-12, atol=1e-13)
            else:
                assert np.array_equal(x, y), rnd
        ctxs["mirror"].update_rows(1, *out["plain"][1])
    assert total_hits > 0
    for ctx in ctxs.values():
        ctx.close()


def test_deterministic_iterations_are_byte_identical(libs, tmp_path):
    """CNF2_DETERMINISTIC (cnf2h_set_deterministic): two runs of 5 iterations leave byte-identical dumps; the ordered sums
    agree with the atomic ones to rounding."""
    capi, host = libs
    ped = synth.make_outbred3(6, 4, 21, 2, seed=31, missing=0.2)
    dumps = []
    for rep in range(2):
        run = host.Run(ped)
        run.set_deterministic(True)
        run.postmarkerdata()
        for _ in range(5):
            run.iteration()
        path = tmp_path / ("dump%d.txt" % rep)
        run.dump(path)
        dumps.append(open(path, "rb").read())
        st = run.state()
        run.close()
    assert dumps[0] == dumps[1] and len(dumps[0]) > 1000
    ctx = capi.Context(0)
    ctx.upload(ped)
    desc = ctx.descendants()
    a = ctx.sweep_accumulate(desc)
    b = ctx.sweep_accumulate(desc, deterministic=True)
    c = ctx.sweep_accumulate(desc, deterministic=True)
    ctx.close()
    for k in ("infprobs", "haplobase", "haplocount"):
        assert np.array_equal(b[k], c[k], equal_nan=True), k
        np.testing.assert_allclose(a[k], b[k], rtol=1e-12, atol=1e-14, equal_nan=True, err_msg=k)
    assert np.array_equal(a["homozyg"], b["homozyg"], equal_nan=True)


def test_fixparents_scan_beyond_one_grid(libs):
    """More records than one launch's grid.y holds (65 535): the scan runs in slabs (config 4 has ~300 000 records)."""
    capi, _ = libs
    base = synth.make_f2(3, 2, 1, seed=4, chrom_cm=5.0, missing=0.3)
    ctx = capi.Context(0)
    ctx.upload(base)
    recs = np.tile(np.arange(base.n_rec, dtype=np.int32), 70000 // base.n_rec + 1)[:70000]
    got = ctx.fixparents_scan(recs)
    want = ctx.fixparents_scan(np.arange(base.n_rec, dtype=np.int32))
    assert np.array_equal(got, want[recs])
    ctx.close()


def _children_of(ped):
    children = np.zeros(ped.n_rec, np.int32)
    for r in ped.dous:
        for k in range(2):
            if ped.par[r, k] >= 0:
                children[ped.par[r, k]] += 1
    return children


@pytest.mark.parametrize("flags", ["mirror", "both"])
def test_update_pass_by_record_lists_equals_the_whole_pass(libs, flags):
    """cnf2_update_pass_records: a rank of a multi-process run updates the records it owns (cnF2freq.cpp:6344-6368 loops over
    individuals; an individual's update reads only its own accumulators and rows).  Three disjoint lists -- one of them
    empty -- over two chromosomes leave the rows, haplobase / haplocount and the hit counters of ONE pass over all records,
    to the bit."""
    capi, _ = libs
    uf = capi.UPDATE_BOTH_FLOWS if flags == "both" else 0
    ped = synth.make_outbred3(5, 4, 23, 2, seed=18, missing=0.2)
    children = _children_of(ped)
    rs = np.random.RandomState(5)
    label = rs.randint(0, 2, ped.n_rec)
    lists = [np.flatnonzero(label == 0).astype(np.int32), np.zeros(0, np.int32), np.flatnonzero(label == 1).astype(np.int32)]
    out = []
    for split in (False, True):
        run_rows = []
        ctx = capi.Context(0)
        ctx.upload_for_updates(ped)
        desc = ctx.descendants()
        total = []
        for rnd in range(2):
            ctx.sweep_accumulate_keep(desc, deterministic=True)
            for c in range(2):
                if split:
                    total.append(sum(ctx.update_pass_records(c, l, children, desc, 0.19, 1.0, flags=uf) for l in lists))
                else:
                    total.append(ctx.update_pass(c, children, desc, 0.19, 1.0, None, flags=uf))
        acc = ctx.download_accumulators()
        out.append((total, ctx.download_rows(1, ped.n_rec), acc))
        ctx.close()
    assert out[0][0] == out[1][0] and sum(out[0][0]) > 0
    for x, y in zip(out[0][1], out[1][1]):
        assert np.array_equal(x, y)
    for k in ("infprobs", "haplobase", "haplocount"):
        assert np.array_equal(out[0][2][k], out[1][2][k], equal_nan=True), k


def test_pack_and_unpack_of_the_exchange(libs):
    """The packers of a multi-process run's exchange (cnf2_pack_accumulators / cnf2_pack_rows and their reverse): listed
    records' accumulators and rows to a packed device buffer and back, any order of records, against the plain downloads."""
    import torch
    capi, _ = libs
    from cnf2freq_amd import dist as cdist
    ped = synth.make_outbred3(4, 3, 19, 2, seed=14, missing=0.2, random_hw=True, random_sure=True)
    ctx = capi.Context(0)
    ctx.upload_for_updates(ped)
    desc = ctx.descendants()
    ctx.sweep_accumulate_keep(desc)
    acc = ctx.download_accumulators()
    rows = ctx.download_rows(1, ped.n_rec)
    M = ped.n_markers
    recs = np.array([7, 2, 11, 3, ped.n_rec - 1], np.int32)
    S, B = M * 6, ((M * 25 + 7) // 8) * 8
    buf = ctx.exchange_buffer(len(recs) * max(S * 8, B))
    ctx.pack_accumulators(recs, buf)
    t = cdist.device_view(buf, len(recs) * S, torch.float64, torch.device("cuda", 0)).cpu().numpy().reshape(len(recs), S)
    for i, r in enumerate(recs):
        assert np.array_equal(t[i, :M * 4], acc["infprobs"][r].ravel())
        assert np.array_equal(t[i, M * 4:M * 5], acc["haplobase"][r]) and np.array_equal(t[i, M * 5:], acc["haplocount"][r])
    assert np.abs(t).sum() > 0
    # scatter them onto other records: the accumulators of recs[i] land on dst[i]
    dst = np.array([0, 1, 4, 5, 6], np.int32)
    ctx.unpack_accumulators(dst, buf)
    acc2 = ctx.download_accumulators()
    for k in acc:
        assert np.array_equal(acc2[k][dst], acc[k][recs]), k
        keep = np.setdiff1d(np.arange(ped.n_rec), dst)
        assert np.array_equal(acc2[k][keep], acc[k][keep]), k
    # rows
    ctx.pack_rows(recs, buf)
    b = cdist.device_view(buf, len(recs) * B, torch.uint8, torch.device("cuda", 0)).cpu().numpy().reshape(len(recs), B)
    for i, r in enumerate(recs):
        assert np.array_equal(b[i, :M * 16].view(np.float64).reshape(M, 2), rows[1][r])
        assert np.array_equal(b[i, M * 16:M * 24].view(np.float64), rows[2][r])
        a8 = b[i, M * 24:M * 25]
        assert np.array_equal(np.stack([a8 & 15, a8 >> 4], axis=1), rows[0][r])
    ctx.unpack_rows(dst, buf)
    rows2 = ctx.download_rows(1, ped.n_rec)
    for x, y in zip(rows2, rows):
        assert np.array_equal(x[dst], y[recs]) and np.array_equal(np.delete(x, dst, axis=0), np.delete(y, dst, axis=0))
    # the window tables are derived again after rows changed (homozygous-everywhere flags): a sweep still runs
    ctx.sweep()
    ctx.close()


def test_iterations_do_not_lose_the_withheld_genotypes(libs):
    """BASELINE config 5 scaled down (125 families = 500 analysed individuals x 2 x 1 000 markers, 20 % of the genotypes
    withheld, 10 haplotyping iterations): the run's picture of the withheld genotypes -- expected allele dosage against the
    generator's truth -- must not get worse over the iterations, and what postmarkerdata had already inferred from relatives
    must still be called."""
    capi, host = libs
    ped = synth.make_outbred3(125, 4, 1000, 2, seed=2, missing=0.2)
    run = host.Run(ped)
    run.postmarkerdata()
    s0 = run.state()
    before = synth.dosage_accuracy(ped, s0)
    withheld = (ped.dense()[0] == 0).all(axis=2) & (np.asarray(ped.empty)[:, None] == 0)
    inferred = withheld & (np.asarray(s0["allele"]) != 0).all(axis=2)
    for _ in range(10):
        run.iteration()
    s1 = run.state()
    after = synth.dosage_accuracy(ped, s1)
    kept = synth.dosage_accuracy(ped, s1, mask=inferred)
    run.close()
    assert before["n"] > 100000 and inferred.sum() > 10000
    assert after["mae"] <= before["mae"] + 1e-3, (before, after)
    assert after["called"] >= before["called"] - 1e-3, (before, after)
    # (inferred genotypes carry no prior, so the updates may move them: measured 0.943 of the hard calls stay right after 10
    # iterations -- the reference's own iteration, see the trajectory goldens -- while every confident call is right)
    assert kept["called"] >= 1 - 1e-3 and kept["concordance"] >= 0.9, kept
    assert after["concordance_confident"] >= 0.99, after
    record = os.path.join(ROOT, "gpurun_out", "withheld_genotypes_500x2000.txt")
    os.makedirs(os.path.dirname(record), exist_ok=True)
    with open(record, "w") as f:
        f.write("before %s\nafter  %s\nkept   %s\n" % (before, after, kept))


def test_iterations_move_parameters_and_round_trip_through_deserialize(libs, tmp_path):
    """--count 3 semantics through the engine: two haplotyping iterations change haplotype weights and certainties,
    the dump of the state re-loads into a fresh run (deserialize, cnF2freq.cpp:7757-7832) to the printed precision, and
    a further iteration from the re-loaded state equals one from the original state to that precision."""
    capi, host = libs
    ped = synth.make_outbred3(3, 3, 12, 2, seed=12, missing=0.2)
    run = host.Run(ped)
    run.postmarkerdata()
    s0 = run.state()
    run.iteration(tmp_path / "rows1.txt")
    s1 = run.state()
    run.iteration(tmp_path / "rows2.txt")
    s2 = run.state()
    free = (s0["hw"] > 0) & (s0["hw"] < 1) & (ped.empty[:, None] == 0)
    assert np.abs(s1["hw"] - s0["hw"])[free].max() > 1e-4, "haplotype weights did not move"
    assert np.abs(s2["hw"] - s1["hw"])[free].max() > 1e-6
    assert np.abs(s1["sure"] - s0["sure"]).max() > 1e-6, "certainties did not move"
    assert np.array_equal(s1["hw"][~free & (ped.empty[:, None] == 0)], s0["hw"][~free & (ped.empty[:, None] == 0)]), \
        "locked weights must stay"
    assert s1["scalefactor"] != 0.013
    text = open(tmp_path / "rows1.txt").read()
    assert "FIRST PASS: 1\n" in text and "SKEWNESS PASS: 1\n" in text
    run.dump(tmp_path / "dump.txt")
    again = host.Run(ped)
    again.postmarkerdata()
    again.deserialize(tmp_path / "dump.txt")
    s3 = again.state()
    assert np.array_equal(s3["allele"], s2["allele"])
    np.testing.assert_allclose(s3["hw"], s2["hw"], atol=5.1e-7)          # "%f"
    np.testing.assert_allclose(s3["sure"], s2["sure"], atol=5.1e-7)      # "%lf"
    run.close()
    again.close()


def test_demo_files_of_the_reference(libs, tmp_path):
    """The reference's only fixture (demo.sh:37 on demoplantimpute.{map,ped,gen}; CRLF files, a read-count token, an
    implicit F1 generation, a gen-1 individual as parent): block headers C:1, D:1, F:1 with 18 rows each in --output.
    `demooutput` is stale (older fork: 4 columns, C and D only, normalised rows): its differences from our
    normalised rows are recorded, not asserted."""
    demo = os.path.join(ROOT, "tests", "golden", "demo")
    exe = os.path.join(ROOT, "cnf2freq_amd", "cnF2freq")
    out = tmp_path / "demo_out.txt"
    r = subprocess.run([exe, "--mapfile", os.path.join(demo, "demoplantimpute.map"), "--pedfile",
                        os.path.join(demo, "demoplantimpute.ped"), "--genfile", os.path.join(demo, "demoplantimpute.gen"),
                        "--output", str(out), "--count", "10", "--normalise"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = open(out).read().split("\n")
    blocks = {}
    i = 0
    while i < len(lines):
        if lines[i] in ("C:1", "D:1", "F:1"):
            rows = []
            j = i + 1
            while lines[j] != "":
                rows.append([float(x) for x in lines[j].split("\t")])
                j += 1
            blocks[lines[i]] = np.array(rows)
            i = j
        i += 1
    assert sorted(blocks) == ["C:1", "D:1", "F:1"]
    for k, b in blocks.items():
        assert b.shape == (18, 3), k
        assert np.allclose(b.sum(axis=1), 1.0, atol=2e-5)
    # 10 dumps of 11 individuals (A B C C_aux_realf C_aux_realm D D_aux_realf D_aux_realm E F H + haplo = 12) x 18 markers
    assert sum(1 for x in lines if x == "1 A") == 10
    assert sum(1 for x in lines if x.startswith("FIRST PASS: ")) > 0
    # record the differences from the stale demooutput (first three columns of blocks C and D)
    stale = open(os.path.join(demo, "demooutput")).read().replace("\r", "").split("\n")
    diffs = {}
    for name in ("C:1", "D:1"):
        k = stale.index(name)
        ref = np.array([[float(x) for x in stale[k + 1 + t].split("\t")[:3]] for t in range(18)])
        diffs[name] = float(np.abs(ref - blocks[name]).max())
    record = os.path.join(ROOT, "gpurun_out", "demo_vs_stale_demooutput.txt")
    os.makedirs(os.path.dirname(record), exist_ok=True)
    with open(record, "w") as f:
        for name, d in diffs.items():
            f.write("%s max |row - demooutput row| = %.5f\n" % (name, d))
        f.write("argmax class agrees on %d of 36 rows\n" % sum(
            int(np.argmax(blocks[n][t]) == np.argmax([float(x) for x in stale[stale.index(n) + 1 + t].split("\t")[:3]]))
            for n in ("C:1", "D:1") for t in range(18)))


def test_reserve_accumulate_and_sweep_clock(libs):
    """cnf2_reserve_accumulate makes the allocations of a later cnf2_sweep_accumulate (the same call is then a no-op for the
    batch buffer: the workspace does not grow), and cnf2_sweep_clock reports the shader clock of the last plain sweep from
    the kernel's own stamps (between 0.5 and 3 GHz on an MI355X)."""
    capi, host = libs
    ped = synth.make_outbred3(6, 4, 40, 2, seed=5, missing=0.2)
    ctx = capi.Context(0)
    ctx.upload(ped)
    assert ctx.sweep_clock() == 0.0
    ctx._chk(ctx.L.cnf2_reserve_accumulate(ctx.h, 0, len(ped.dous), 0), "cnf2_reserve_accumulate")
    w0 = ctx.L.cnf2_workspace_bytes(ctx.h)
    desc = ctx.descendants()
    r1 = ctx.sweep_accumulate(desc)
    assert ctx.L.cnf2_workspace_bytes(ctx.h) == w0
    r2 = ctx.sweep()
    mhz = ctx.sweep_clock()
    assert 500.0 < mhz < 3000.0, mhz
    assert np.allclose(r1["loglik"], r2["loglik"], rtol=1e-12, atol=0)
    ctx.close()
