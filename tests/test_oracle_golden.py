"""CPU suite: the C restatement (oracle/) against the golden vectors produced by the
reference's own code (tests/golden/make_golden.py).  Bit-exact where the arithmetic
order is the same (everything except log/exp library rounding, which is shared too)."""
import numpy as np

from conftest import oracle_ped


def test_founder_and_fixtrees(golden):
    ped, z = golden
    o = oracle_ped(ped)
    assert (ped.founder == z["founder"]).all()
    assert (o.founder == z["founder"]).all()
    for j, ind in enumerate(ped.dous):
        t = o.fixtrees(int(ind))
        assert (t.shiftignore, t.flag2ignore) == tuple(z["fixtrees"][j])
        rel = sorted(zip(list(t.rel_rec)[:t.n_rel], list(t.rel_map)[:t.n_rel], list(t.rel_mapshift)[:t.n_rel]))
        want = [tuple(r) for r in z["rel"][j] if r[0] >= 0]
        assert rel == want
        assert list(t.ordered) == list(z["ordered"][j])


def test_emission_and_mapval(golden):
    ped, z = golden
    o = oracle_ped(ped)
    for (ind, m, g, f2, s), want in zip(z["em_idx"], z["em_val"]):
        assert o.emission(int(ind), int(m), int(g), int(f2), int(s)) == want
    for (ind, m, g, f2, s), want in zip(z["mv_idx"], z["mv_val"]):
        assert o.mapval(int(ind), int(m), int(g), int(f2), int(s))[0] == want


def test_ignoreflag2(golden):
    ped, z = golden
    o = oracle_ped(ped)
    trees = {j: o.fixtrees(int(ind)) for j, ind in enumerate(ped.dous)}
    for (j, m, g, f2, s), want in zip(z["ig_idx"], z["ig_val"]):
        assert o.ignoreflag2(trees[int(j)], int(f2), int(g), int(s), int(m)) == want


def test_sweep_store_factors_dosage(golden):
    ped, z = golden
    o = oracle_ped(ped)
    for j, ind in enumerate(ped.dous):
        r = o.sweep_ind(int(ind), int(ped.gen[ind]), mode=0, keep_store=True)
        assert r["ok"] == bool(z["ok"][j])
        live = z["factors"][j] > -1e29
        # the terminal query reads markerposes[endmark+1], one past the vector for the last
        # chromosome (cnF2freq.cpp:1984,2193-2195): a sum-preserving step over a garbage
        # distance, so the reference's own factors carry ~1 ulp of noise there.
        np.testing.assert_allclose(r["factors"][live], z["factors"][j][live], rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(r["factor"], z["factor"][j], rtol=1e-13, atol=1e-13)
        assert np.array_equal(r["fwbw"][live], z["fwbw"][j][live])
        assert np.array_equal(r["fwbwfactors"][live], z["fwbwfactors"][j][live])
        if r["ok"]:
            np.testing.assert_allclose(r["dosage"], z["dosage"][j], rtol=1e-12, atol=1e-15)


def test_closed_form_dosage_matches_reference_fanout(golden):
    """mode 2 (rank-2 class-split tables + tie rule) is what the HIP kernels implement."""
    ped, z = golden
    o = oracle_ped(ped)
    for j, ind in enumerate(ped.dous):
        if not z["ok"][j]:
            continue
        r = o.sweep_ind(int(ind), int(ped.gen[ind]), mode=2)
        np.testing.assert_allclose(r["dosage"], z["dosage"][j], rtol=1e-11, atol=1e-14)


def test_turn_scan(golden):
    ped, z = golden
    o = oracle_ped(ped)
    for j, ind in enumerate(ped.dous):
        if not z["ok"][j]:
            continue
        for ti, m in enumerate(z["turn_markers"]):
            got = o.turn_scan(int(ind), int(m), int(ped.gen[ind]))
            want = z["rawervals"][j, ti]
            assert np.array_equal(np.isnan(got), np.isnan(want))
            live = ~np.isnan(want)
            np.testing.assert_allclose(got[live], want[live], rtol=1e-12, atol=1e-12)


def test_rank2_tables_reproduce_emission():
    from cnf2freq_amd import synth
    ped = synth.make_random_windows(20, 4, seed=5)
    o = oracle_ped(ped)
    for ind in ped.dous:
        for m in range(ped.n_markers):
            for s in range(8):
                T = o.emission_tables(int(ind), m, s, 0)
                for g in range(0, 64, 5):
                    e = o.emission(int(ind), m, g, -1, s)
                    e2 = sum(T["c"][f] * T["A"][f][g & 7] * T["B"][f][g >> 3] for f in range(2))
                    assert abs(e - e2) <= 1e-14 * max(abs(e), 1e-300) + 1e-300


def test_haplos_accumulators(golden):
    """HAPLOS update mode (updatehaplo) of HOT LOOP 2 against the reference's thread-private haplos."""
    ped, z = golden
    o = oracle_ped(ped)
    for j, ind in enumerate(ped.dous):
        if not z["ok"][j]:
            continue
        for ti, m in enumerate(z["turn_markers"]):
            got = o.haplos_row(int(ind), int(m), int(ped.gen[ind]))
            np.testing.assert_allclose(got, z["haplos"][j, ti], rtol=1e-12, atol=1e-15)


def test_infprobs_homozyg_accumulators(golden):
    """GENOSPROBE / HOMOZYGOUS / GENOS update modes of HOT LOOP 2 (cnF2freq.cpp:5513-5577) against the
    reference's thread-private infprobs and its homozyg increments."""
    ped, z = golden
    o = oracle_ped(ped)
    for j, ind in enumerate(ped.dous):
        if not z["ok"][j]:
            continue
        for ti, m in enumerate(z["turn_markers"]):
            inf, hz = o.infprobs_row(int(ind), int(m), int(ped.gen[ind]))
            np.testing.assert_allclose(inf, z["infprobs"][j, ti], rtol=1e-12, atol=1e-15)
            np.testing.assert_allclose(hz, z["homozyg"][j, ti], rtol=1e-12, atol=1e-15)


def test_addvariance(golden):
    """individ::addvariance (cnF2freq.cpp:1489-1558; trackpossible with zeropropagate = NO_EQUIVALENCE) for
    every record and marker, with the record's own flag2ignore as postmarkerdata passes it."""
    ped, z = golden
    o = oracle_ped(ped)
    for rec in range(ped.n_rec):
        f2i = int(z["variances_flag2ignore"][rec])
        assert o.fixtrees(rec).flag2ignore == f2i
        for m in range(ped.n_markers):
            got = o.addvariance(rec, m, f2i)
            want = z["variances"][rec, m]
            if np.isnan(want):
                assert got is None
            else:
                assert got is not None
                np.testing.assert_allclose(got, want, rtol=1e-13, atol=0)


def test_hot_loop_2_reductions(golden):
    """HOT LOOP 2 with moveinfprobs / movehaplos (cnF2freq.cpp:5876-5902, 3577-3616) over all analysed
    individuals in order: per-record infprobs, haplobase, haplocount, per-individual homozyg."""
    ped, z = golden
    o = oracle_ped(ped)
    got = o.accumulate(ped.dous, ped.gen[ped.dous], z["acc_desc"])
    for k in ("infprobs", "haplobase", "haplocount", "homozyg"):
        np.testing.assert_allclose(got[k], z["acc_" + k], rtol=1e-12, atol=1e-15, equal_nan=True)


def test_the_1e300_rule_of_adjustprobs_where_it_decides_a_result():
    """Golden G15 (flush_impossible.npz: the reference's own likelihoods on data built so that adjustprobs' 1e-300 rule,
    cnF2freq.cpp:1607-1611, makes four shift modes of a child impossible): the restatement applies the rule and gives the
    reference's factors, MINFACTOR pattern included."""
    from conftest import load_golden
    ped, z = load_golden("flush_impossible")
    o = oracle_ped(ped)
    got = o.sweep_batch(ped.dous, ped.gen[ped.dous], mode=2)
    gone = z["factors"] < -1e14
    assert gone.sum() == 4
    assert np.array_equal(got["factors"] < -1e14, gone)
    np.testing.assert_allclose(got["factors"][~gone], z["factors"][~gone], rtol=1e-12)
