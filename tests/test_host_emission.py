"""CPU suite: the product's emission/window code (cnf2freq_amd/csrc/cnf2_emission.h,
cnf2_lane.h, cnf2_window.cpp -- the same headers the HIP kernels include) compiled for the
host and compared with the oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from cnf2freq_amd import synth
from conftest import ROOT, oracle_ped

SHIM_DIR = os.path.join(ROOT, "tests", "shim")
CSRC = os.path.join(ROOT, "cnf2freq_amd", "csrc")


@pytest.fixture(scope="module")
def shim():
    from conftest import build_host_shim
    return build_host_shim()


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _ped_args(ped):
    return [C.c_int(ped.n_rec), _p(ped.par), _p(ped.empty), _p(ped.gen), _p(ped.row_of)]


def lane_index(P, f, sp, k):
    return (P << 5) | (f << 4) | (sp << 3) | k


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_window_matches_fixtrees(shim, seed):
    ped = synth.make_random_windows(40, 3, seed=seed)
    o = oracle_ped(ped)
    f = np.zeros(ped.n_rec, np.uint8)
    shim.shim_founders(*_ped_args(ped), _p(f))
    assert (f == o.founder).all()
    for ind in ped.dous:
        out = np.zeros(17, np.int32)
        ng = shim.shim_window(*_ped_args(ped), int(ind), _p(out))
        t = o.fixtrees(int(ind))
        assert (out[0], out[1]) == (t.shiftignore, t.flag2ignore)
        # tie groups == relmap entries with more than one bit
        want = sorted(m for m in list(t.rel_map)[:t.n_rel] if bin(m).count("1") > 1)
        got = sorted(sum(1 << i for i in range(7) if out[10 + i] == g) for g in range(ng))
        assert got == want


@pytest.mark.parametrize("seed", [11, 12])
def test_table_entries_match_oracle(shim, seed):
    ped = synth.make_random_windows(30, 4, seed=seed)
    o = oracle_ped(ped)
    tot, rtot, two, c4 = np.zeros(64), np.zeros(64), np.zeros(64), np.zeros(4)
    for ind in ped.dous:
        ind = int(ind)
        t = o.fixtrees(ind)
        for m in range(ped.n_markers):
            shim.shim_emtab(*_ped_args(ped), _p(ped.allele), _p(ped.sure), _p(ped.hw), ped.n_markers,
                            ind, m, 0, _p(tot), _p(rtot), _p(two), _p(c4))
            for s in range(8):
                s0, s1, s2 = s & 1, (s >> 1) & 1, (s >> 2) & 1
                T = o.emission_tables(ind, m, s, t.flag2ignore)
                for f in range(2):
                    np.testing.assert_allclose(c4[f * 2 + s0], T["c"][f], rtol=1e-15, atol=0)
                    if T["c"][f] == 0:
                        continue  # the oracle skips the recursion when the root term is zero
                    for k in range(8):
                        np.testing.assert_allclose(tot[lane_index(0, f, s1, k)], T["A"][f][k], rtol=4e-16, atol=0)
                        np.testing.assert_allclose(tot[lane_index(1, f, s2, k)], T["B"][f][k], rtol=4e-16, atol=0)
                # full emission e(g) from the lane table
                for g in range(0, 64, 7):
                    e = sum(c4[f * 2 + s0] * tot[lane_index(0, f, s1, g & 7)] * tot[lane_index(1, f, s2, g >> 3)]
                            for f in range(2))
                    np.testing.assert_allclose(e, o.emission(ind, m, g, -1, s), rtol=1e-14, atol=1e-300)


@pytest.mark.parametrize("seed", [21, 22, 23])
def test_restricted_class_split_reproduces_reference_rows(shim, seed):
    """Row = sum over tie combos of the class-split restricted tables weighted by
    alpha-minus * beta; must equal the reference-semantics fan-out (oracle mode 0)."""
    ped = synth.make_random_windows(24, 3, seed=seed)
    o = oracle_ped(ped)
    tot, rtot, two, c4 = np.zeros(64), np.zeros(64), np.zeros(64), np.zeros(4)
    for ind in ped.dous:
        ind = int(ind)
        r = o.sweep_ind(ind, int(ped.gen[ind]), mode=0, keep_store=True)
        if not r["ok"]:
            continue
        t = o.fixtrees(ind)
        out = np.zeros(17, np.int32)
        ng = shim.shim_window(*_ped_args(ped), ind, _p(out))
        for m in range(ped.n_markers):
            row = np.zeros(3)
            for s in range(8):
                if s & t.shiftignore or r["factor"] - r["factors"][s] > 40:
                    continue
                s0, s1, s2 = s & 1, (s >> 1) & 1, (s >> 2) & 1
                w = (r["fwbw"][s, m, 0] * r["fwbw"][s, m, 1]
                     * np.exp(r["fwbwfactors"][s, m, 0] + r["fwbwfactors"][s, m, 1] - r["factor"]))
                for combo in range(1 << ng):
                    shim.shim_emtab(*_ped_args(ped), _p(ped.allele), _p(ped.sure), _p(ped.hw), ped.n_markers,
                                    ind, m, combo, _p(tot), _p(rtot), _p(two), _p(c4))
                    for g in range(64):
                        for f in range(2):
                            la, lb = lane_index(0, f, s1, g & 7), lane_index(1, f, s2, g >> 3)
                            a, a1, b, b1 = rtot[la], two[la], rtot[lb], two[lb]
                            c = c4[f * 2 + s0]
                            row[2] += w[g] * c * a1 * b1
                            row[1] += w[g] * c * (a1 * (b - b1) + (a - a1) * b1)
                            row[0] += w[g] * c * (a - a1) * (b - b1)
            np.testing.assert_allclose(row, r["dosage"][m], rtol=1e-10, atol=1e-14)


@pytest.mark.parametrize("maker", [
    lambda: synth.make_random_windows(40, 4, seed=21),
    lambda: synth.make_random_windows(40, 3, seed=22),
    lambda: synth.make_random_windows(40, 4, seed=23),
    lambda: synth.make_outbred3(2, 2, 7, 1, seed=8, missing=0.25, random_hw=True, random_sure=True),
    lambda: synth.make_f2(3, 6, 1, seed=5, chrom_cm=20.0, missing=0.2),
    lambda: synth.make_ail(4, 6, 3, 5, 1, seed=5, chrom_cm=20.0),
])
@pytest.mark.parametrize("form", [0, 2])
def test_path_form_of_all_accumulators_matches_fanout(shim, maker, form):
    """cnf2_accpath.h (the accumulators as sums over the allele paths of a line, with the state bits entering through
    butterfly transforms of the entry weights: what the fast accumulate kernel evaluates, emulated here lane by lane)
    against the oracle's brute-force fan-outs.  form 0: one lane per path, butterflies across lanes; form 2: one lane
    per (P, f, traced grandparent) with its 8 paths in registers (the kernel's tile form)."""
    ped = maker()
    o = oracle_ped(ped)
    checked = tied = 0
    for ind in ped.dous:
        slots = np.zeros(17, np.int32)
        shim.shim_window(*_ped_args(ped), int(ind), _p(slots))
        for m in (0, ped.n_markers - 1):
            wg = _mode_weights(o, ped, ind, m)
            if wg is None:
                continue
            inf, hz, hap = np.zeros((7, 2, 2)), np.zeros(2), np.zeros((7, 2))
            ng = shim.shim_acc_contract_paths(*_ped_args(ped), _p(ped.allele), _p(ped.sure), _p(ped.hw), ped.n_markers,
                                              int(ind), m, _p(np.ascontiguousarray(wg)), form, _p(inf), _p(hz), _p(hap))
            if ng < 0:
                continue
            want, want_hz = o.infprobs_row(int(ind), m, int(ped.gen[ind]))
            want_hap = o.haplos_row(int(ind), m, int(ped.gen[ind]))
            got, got_hap = np.zeros_like(want), np.zeros_like(want_hap)
            for k in range(7):
                if slots[3 + k] >= 0:
                    got[slots[3 + k]] += inf[k]
                    got_hap[slots[3 + k]] += hap[k]
            np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-13)
            np.testing.assert_allclose(hz, want_hz, rtol=1e-9, atol=1e-13)
            np.testing.assert_allclose(got_hap, want_hap, rtol=1e-9, atol=1e-13)
            checked += 1
            tied += int(ng > 0)
    assert checked > 0


@pytest.mark.parametrize("seed", [31, 32, 33])
def test_fast_tile_producer_matches_general_producer(shim, seed):
    """cnf2_emtab.h (division-free tile producer of the fast kernel) against cnf2_emission.h:
    same emission e_s(g) and same class-split products for every state and shift mode."""
    ped = synth.make_random_windows(40, 4, seed=seed)
    tot, rtot, two, c4 = np.zeros(64), np.zeros(64), np.zeros(64), np.zeros(4)
    ftot, frtot, ftwo, fc4 = np.zeros(64), np.zeros(64), np.zeros(64), np.zeros(4)
    g = np.arange(64)
    for ind in ped.dous:
        ind = int(ind)
        for m in range(ped.n_markers):
            args = _ped_args(ped) + [_p(ped.allele), _p(ped.sure), _p(ped.hw), ped.n_markers, ind, m]
            ng = shim.shim_emtab(*args, 0, _p(tot), _p(rtot), _p(two), _p(c4))
            shim.shim_emtab_fast(*args, _p(ftot), _p(frtot), _p(ftwo), _p(fc4))
            kinds = [((tot, tot), (ftot, ftot))]
            if ng == 0:  # the fast producer has no tie rule; tied windows take the general kernel
                kinds += [((rtot, rtot), (frtot, frtot)), ((two, rtot), (ftwo, frtot)),
                          ((rtot, two), (frtot, ftwo)), ((two, two), (ftwo, ftwo))]
            for s in range(8):
                s0, s1, s2 = s & 1, (s >> 1) & 1, (s >> 2) & 1
                for (ta, tb), (fa, fb) in kinds:
                    want = sum(c4[f * 2 + s0] * ta[lane_index(0, f, s1, g & 7)] * tb[lane_index(1, f, s2, g >> 3)]
                               for f in range(2))
                    got = sum(fc4[f * 2 + s0] * fa[lane_index(0, f, s1, g & 7)] * fb[lane_index(1, f, s2, g >> 3)]
                              for f in range(2))
                    np.testing.assert_allclose(got, want, rtol=1e-13, atol=1e-300)


@pytest.mark.parametrize("seed", [21, 22, 23])
def test_tile_producer_with_the_tie_rule_matches_general_producer(shim, seed):
    """The tile producer's tie form (restricted tables of one tie combination: admissibility masks per allele and state
    bit, cnf2_emtab.h TIES) against line_restricted() with the forces of the combination, for every combination of
    every tied window: the tables the tied windows' instantiation of the fast sweep kernel reads."""
    ped = synth.make_random_windows(40, 4, seed=seed)
    tot, rtot, two, c4 = np.zeros(64), np.zeros(64), np.zeros(64), np.zeros(4)
    ftot, frtot, ftwo, fc4 = np.zeros(64), np.zeros(64), np.zeros(64), np.zeros(4)
    g = np.arange(64)
    tied = 0
    for ind in ped.dous:
        ind = int(ind)
        for m in (0, ped.n_markers - 1):
            args = _ped_args(ped) + [_p(ped.allele), _p(ped.sure), _p(ped.hw), ped.n_markers, ind, m]
            ng = shim.shim_emtab(*args, 0, _p(tot), _p(rtot), _p(two), _p(c4))
            for combo in range(1 << ng):
                shim.shim_emtab(*args, combo, _p(tot), _p(rtot), _p(two), _p(c4))
                shim.shim_emtab_fast_ties(*args, combo, _p(ftot), _p(frtot), _p(ftwo), _p(fc4))
                # the two producers split the root's factors differently between the tables and c: compare what the
                # kernels form, c_f(s0) * A * B, for the emission and the four class products
                kinds = [((tot, tot), (ftot, ftot)), ((rtot, rtot), (frtot, frtot)), ((two, rtot), (ftwo, frtot)),
                         ((rtot, two), (frtot, ftwo)), ((two, two), (ftwo, ftwo))]
                for s in range(8):
                    s0, s1, s2 = s & 1, (s >> 1) & 1, (s >> 2) & 1
                    for (ta, tb), (fa, fb) in kinds:
                        want = sum(c4[f * 2 + s0] * ta[lane_index(0, f, s1, g & 7)] * tb[lane_index(1, f, s2, g >> 3)]
                                   for f in range(2))
                        got = sum(fc4[f * 2 + s0] * fa[lane_index(0, f, s1, g & 7)] * fb[lane_index(1, f, s2, g >> 3)]
                                  for f in range(2))
                        np.testing.assert_allclose(got, want, rtol=1e-13, atol=1e-300)
            tied += int(ng > 0)
    assert tied > 0


@pytest.mark.parametrize("seed", [31, 32])
def test_table_form_of_the_tile_producer_is_identical(shim, seed):
    """The two-phase tile producer (7 per-slot match records shared by the 8 parts, SlotTable) must give
    exactly the tables of the form that redoes the match logic in every part (SlotDirect)."""
    for ped in (synth.make_random_windows(40, 4, seed=seed),
                synth.make_outbred3(2, 3, 9, 1, seed=seed, missing=0.3, random_hw=True, random_sure=True)):
        a = [np.zeros(64), np.zeros(64), np.zeros(64), np.zeros(4)]
        b = [np.zeros(64), np.zeros(64), np.zeros(64), np.zeros(4)]
        for ind in ped.dous:
            for m in range(ped.n_markers):
                args = _ped_args(ped) + [_p(ped.allele), _p(ped.sure), _p(ped.hw), ped.n_markers, int(ind), m]
                shim.shim_emtab_fast(*args, *[_p(x) for x in a])
                shim.shim_emtab_tables(*args, *[_p(x) for x in b])
                for x, y in zip(a, b):
                    assert np.array_equal(x, y)


def test_single_parent_allele_shortcut_is_identical(shim):
    """HOMPAR / HOMLEAF: where the parent (and both its parents) are homozygous with equal sure, evaluating one
    of the allele indices gives exactly the same tables (F2 with empty F1 parents and inbred founders; random
    windows with missing data)."""
    used = 0
    for ped in (synth.make_f2(4, 9, 1, seed=3, chrom_cm=30.0, missing=0.2), synth.make_random_windows(40, 4, seed=77),
                synth.make_outbred3(3, 3, 9, 1, seed=5, missing=0.2, random_hw=True, random_sure=True)):
        a = [np.zeros(64), np.zeros(64), np.zeros(64), np.zeros(4)]
        b = [np.zeros(64), np.zeros(64), np.zeros(64), np.zeros(4)]
        for ind in ped.dous:
            for m in range(ped.n_markers):
                args = _ped_args(ped) + [_p(ped.allele), _p(ped.sure), _p(ped.hw), ped.n_markers, int(ind), m]
                shim.shim_emtab_fast(*args, *[_p(x) for x in a])
                used += shim.shim_emtab_hompar(*args, *[_p(x) for x in b])
                for x, y in zip(a, b):
                    assert np.array_equal(x, y)
    # all three shortcuts were exercised (NORESTR: complete windows, flag2ignore == 0)
    assert used % 1000 > 100 and (used // 1000) % 1000 > 100 and used // 1000000 > 100


def _mode_weights(o, ped, ind, m):
    """wg[s][g] = exp(scales - factor) * alphaminus_s(g) * beta_s(g) from the oracle's store, 0 for the
    modes HOT LOOP 2 skips (cnF2freq.cpp:5420-5421)."""
    gen = int(ped.gen[ind])
    r = o.sweep_ind(int(ind), gen, mode=2, keep_store=True)
    if not r["ok"]:
        return None
    t = o.fixtrees(int(ind))
    wg = np.zeros((8, 64))
    for s in range(8 if gen >= 2 else 2):
        if (s & t.shiftignore) or r["factor"] - r["factors"][s] > 40:
            continue
        wg[s] = (np.exp(r["fwbwfactors"][s, m, 0] + r["fwbwfactors"][s, m, 1] - r["factor"])
                 * r["fwbw"][s, m, 0] * r["fwbw"][s, m, 1])
    return wg


@pytest.mark.parametrize("maker", [
    lambda: synth.make_random_windows(30, 4, seed=21),
    lambda: synth.make_outbred3(2, 2, 7, 1, seed=8, missing=0.25, random_hw=True, random_sure=True),
    lambda: synth.make_f2(3, 6, 1, seed=5, chrom_cm=20.0, missing=0.2),
    lambda: synth.make_ail(4, 6, 3, 5, 1, seed=5, chrom_cm=20.0),
])
def test_closed_form_infprobs_matches_fanout(shim, maker):
    """cnf2_accum.h (closed form of the infprobs / homozyg accumulators of HOT LOOP 2) against the oracle's
    brute-force fan-out over (state, shift mode, path), which is pinned on the reference."""
    ped = maker()
    o = oracle_ped(ped)
    checked = 0
    for ind in ped.dous:
        slots = np.zeros(17, np.int32)
        shim.shim_window(*_ped_args(ped), int(ind), _p(slots))
        for m in (0, ped.n_markers - 1):
            wg = _mode_weights(o, ped, ind, m)
            if wg is None:
                continue
            inf, hz = np.zeros((7, 2, 2)), np.zeros(2)
            shim.shim_accum_infprobs(*_ped_args(ped), _p(ped.allele), _p(ped.sure), _p(ped.hw), ped.n_markers,
                                     int(ind), m, _p(np.ascontiguousarray(wg)), 0, _p(inf), _p(hz))
            want, want_hz = o.infprobs_row(int(ind), m, int(ped.gen[ind]))
            got = np.zeros_like(want)
            for k in range(7):
                if slots[3 + k] >= 0:
                    got[slots[3 + k]] += inf[k]
            np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-13)
            np.testing.assert_allclose(hz, want_hz, rtol=1e-9, atol=1e-13)
            checked += 1
    assert checked > 0


@pytest.mark.parametrize("maker", [
    lambda: synth.make_random_windows(40, 4, seed=21),
    lambda: synth.make_random_windows(40, 3, seed=22),
    lambda: synth.make_outbred3(2, 2, 7, 1, seed=8, missing=0.25, random_hw=True, random_sure=True),
    lambda: synth.make_f2(3, 6, 1, seed=5, chrom_cm=20.0, missing=0.2),
    lambda: synth.make_ail(4, 6, 3, 5, 1, seed=5, chrom_cm=20.0),
])
def test_table_form_of_all_accumulators_matches_fanout(shim, maker):
    """cnf2_acctab.h (every HOT LOOP 2 accumulator of one (individual, marker) as 16-term dot products of per-line
    tables with partial contractions of wg: what the batched accumulate kernel evaluates) against the oracle's
    brute-force fan-outs: infprobs / homozyg (GENOSPROBE, GENOS, HOMOZYGOUS) and haplos (updatehaplo)."""
    ped = maker()
    o = oracle_ped(ped)
    checked = tied = 0
    for ind in ped.dous:
        slots = np.zeros(17, np.int32)
        ng = shim.shim_window(*_ped_args(ped), int(ind), _p(slots))
        for m in (0, ped.n_markers - 1):
            wg = _mode_weights(o, ped, ind, m)
            if wg is None:
                continue
            inf, hz, hap = np.zeros((7, 2, 2)), np.zeros(2), np.zeros((7, 2))
            shim.shim_acc_contract(*_ped_args(ped), _p(ped.allele), _p(ped.sure), _p(ped.hw), ped.n_markers,
                                   int(ind), m, _p(np.ascontiguousarray(wg)), 0, _p(inf), _p(hz), _p(hap))
            want, want_hz = o.infprobs_row(int(ind), m, int(ped.gen[ind]))
            want_hap = o.haplos_row(int(ind), m, int(ped.gen[ind]))
            got, got_hap = np.zeros_like(want), np.zeros_like(want_hap)
            for k in range(7):
                if slots[3 + k] >= 0:
                    got[slots[3 + k]] += inf[k]
                    got_hap[slots[3 + k]] += hap[k]
            np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-13)
            np.testing.assert_allclose(hz, want_hz, rtol=1e-9, atol=1e-13)
            np.testing.assert_allclose(got_hap, want_hap, rtol=1e-9, atol=1e-13)
            checked += 1
            tied += int(ng > 0)
    assert checked > 0


@pytest.mark.parametrize("seed", [31, 32, 33])
def test_factored_table_entries_equal_the_path_walk(shim, seed):
    """acc_entry (sums over paths factored into 2 x 2 sums; what the kernel evaluates) against acc_entry_paths (the
    8-path walk pinned on the oracle above), every entry and tie combination, random windows with exact zeros
    (impossible paths, 0 / 0 weights: NaN must appear in the same places)."""
    ped = synth.make_random_windows(40, 3, seed=seed)
    ped.sure = ped.sure.copy()
    ped.sure[ped.sure < 0.02] = 0.0
    # the sex-marker sentinel (cnF2freq.cpp:226) with sure 0 matches neither probe value: 0 / 0 weights
    rs = np.random.RandomState(seed)
    hit = rs.rand(*ped.allele.shape[:2]) < 0.1
    hit[0] = False
    ped.allele = ped.allele.copy()
    ped.allele[hit, 1] = 9
    ped.sure[hit, 1] = 0.0
    nan_seen = 0
    for ind in ped.dous:
        for m in range(ped.n_markers):
            a, b = np.zeros((64, 23)), np.zeros((64, 23))
            ng = shim.shim_acc_entries(*_ped_args(ped), _p(ped.allele), _p(ped.sure), _p(ped.hw), ped.n_markers, int(ind), m,
                                       0, _p(a), _p(b))
            for combo in range(1 << ng):
                shim.shim_acc_entries(*_ped_args(ped), _p(ped.allele), _p(ped.sure), _p(ped.hw), ped.n_markers, int(ind), m,
                                      combo, _p(a), _p(b))
                assert np.array_equal(np.isnan(a), np.isnan(b))
                np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-300, equal_nan=True)
                nan_seen += int(np.isnan(b).any())
    assert nan_seen > 0 or seed != 31


@pytest.mark.parametrize("maker", [
    lambda: synth.make_random_windows(40, 4, seed=61),
    lambda: synth.make_outbred3(2, 2, 7, 1, seed=8, missing=0.25, random_hw=True, random_sure=True),
    lambda: synth.make_f2(3, 6, 1, seed=5, chrom_cm=20.0, missing=0.2),
])
def test_closed_form_variance_matches_addvariance(shim, maker):
    """cnf2_variance.h (addvariance as a product of two per-line sums) against the oracle's restatement of the
    reference's 65 536-call loops (cnF2freq.cpp:1489-1558), for EVERY record of the pedigree (postmarkerdata calls it
    for everybody), incl. founders, empty individuals, missing parents and the sex-marker sentinel."""
    ped = maker()
    rs = np.random.RandomState(7)
    hit = rs.rand(*ped.allele.shape[:2]) < 0.05
    hit[0] = False
    ped.allele = ped.allele.copy()
    ped.allele[hit, 1] = 9
    o = oracle_ped(ped)
    seen_none = seen_val = differs = 0
    for rec in range(ped.n_rec):
        f2i = o.fixtrees(rec).flag2ignore
        for m in range(ped.n_markers):
            want = o.addvariance(rec, m, f2i)
            got = np.zeros(1)
            ok = shim.shim_variance(*_ped_args(ped), _p(ped.allele), _p(ped.sure), _p(ped.hw), ped.n_markers, rec, m, _p(got))
            if want is None:
                assert ok == 0
                seen_none += 1
            else:
                assert ok == 1
                # the value is a squared difference of two nearly equal sums: cancellation amplifies rounding
                assert abs(got[0] - want) <= 1e-8 * abs(want) + 1e-18, (rec, m, got[0], want)
                seen_val += 1
            # variance_exact: the reference's additions in the reference's order -- the same bits (the oracle's are the
            # reference's on goldens G10 / G12, tests/test_oracle_golden.py)
            exact = np.zeros(1)
            ok = shim.shim_variance_exact(*_ped_args(ped), _p(ped.allele), _p(ped.sure), _p(ped.hw), ped.n_markers, rec, m, _p(exact))
            assert ok == (0 if want is None else 1)
            if want is not None:
                assert exact[0] == want, (rec, m, exact[0], want)
                differs += int(exact[0] != got[0])
    assert seen_val > 50
    assert differs > 0, "the closed form has the reference's bits everywhere: the case proves nothing"


def test_variance_exact_has_the_references_bits_on_the_goldens(shim):
    """variance_exact (cnf2_variance.h: addvariance's additions in the reference's order) against the reference's OWN
    variances[] on goldens G10 (oracle/_ref), every record and marker: equal to the bit."""
    from conftest import load_golden, GOLDEN_CASES
    total = 0
    for case in GOLDEN_CASES:
        ped, z = load_golden(case)
        for rec in range(ped.n_rec):
            for m in range(ped.n_markers):
                want = z["variances"][rec, m]
                got = np.zeros(1)
                ok = shim.shim_variance_exact(*_ped_args(ped), _p(ped.allele), _p(ped.sure), _p(ped.hw), ped.n_markers, rec, m, _p(got))
                if np.isnan(want):
                    assert ok == 0
                else:
                    assert ok == 1 and got[0] == want, (case, rec, m, got[0], want)
                    total += 1
    assert total > 1500
