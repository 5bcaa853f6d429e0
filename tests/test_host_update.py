"""CPU suite: the per-iteration parameter updates.  The product's form
(cnf2freq_amd/csrc/cnf2_update.h, the header the update kernels include, compiled for the host) against the
oracle's literal restatement of processinfprobs / updatehaploweights / cappedgd / relskewhmm
(oracle/cnf2_oracle_iter.c, cnF2freq.cpp:4004-4734).  Tolerance 1e-9; parity of
the restatement itself: bit-exact on goldens G14 (tests/test_oracle_iter_golden.py).  The last two tests put the product's
form directly against G14, i.e. against the outputs of the reference's own functions."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN_DIR, ROOT
from oracle import pyoracle

SHIM_DIR = os.path.join(ROOT, "tests", "shim")
CSRC = os.path.join(ROOT, "cnf2freq_amd", "csrc")
D = C.c_double


@pytest.fixture(scope="module")
def shim():
    from conftest import build_host_shim
    L = build_host_shim()
    IP = C.POINTER(C.c_int)
    L.shim_cap_step.argtypes = [D, D, D, IP, C.c_int]
    L.shim_cap_step.restype = D
    L.shim_gauss15_reciprocal_linear.argtypes = [D, D, D, D]
    L.shim_gauss15_reciprocal_linear.restype = D
    L.shim_evidence_slope.argtypes = [D, D, D, D]
    L.shim_evidence_slope.restype = D
    L.shim_update_certainty.argtypes = [C.c_void_p, C.c_int, C.c_int, D, C.c_int, C.c_int, D, C.c_int, C.c_int, D, D,
                                        IP, IP, C.POINTER(D)]
    L.shim_phase_ratio.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.shim_update_haploweights.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_int, C.c_int, D, D, IP]
    L.shim_flow.argtypes = [C.c_int, D, D, D, D, D, D, D, D, D, C.c_int, C.c_void_p]
    L.shim_time_bound.argtypes = [C.c_int, D, D, D, D, D, D, D, D, D, C.c_int, C.c_void_p]
    L.shim_adapt_scalefactor.argtypes = [D, C.c_int, C.c_void_p, C.c_int]
    L.shim_adapt_scalefactor.restype = D
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_gauss_legendre_15_nodes_are_the_published_rule(shim):
    """Both tables against numpy's own 15-point Gauss-Legendre rule, through an integral with a known value:
    int_a^b dx / (s x + c) = log((s b + c) / (s a + c)) / s (smooth on the interval: the rule is exact to ~1e-15)."""
    O = pyoracle.lib()
    x, w = np.polynomial.legendre.leggauss(15)
    for (s, c, a, b) in [(1.0, 1.0, 0.0, 1.0), (-0.7, 2.0, 0.2, 0.9), (3.0, 0.5, 0.1, 0.11)]:
        exact = np.log((s * b + c) / (s * a + c)) / s
        ref = 0.5 * (b - a) * np.sum(w / (s * (0.5 * (a + b) + 0.5 * (b - a) * x) + c))
        for got in (shim.shim_gauss15_reciprocal_linear(s, c, a, b), O.cnf2o_gauss15_reciprocal_linear(s, c, a, b)):
            assert abs(got - ref) < 2e-15 * max(1.0, abs(ref))
            assert abs(got - exact) < 1e-12 * max(1.0, abs(exact))


def test_cap_step_equals_caplogitchange(shim):
    O = pyoracle.lib()
    rs = np.random.RandomState(1)
    for _ in range(4000):
        orig = rs.choice([rs.rand(), 0.5, 1e-7, 1 - 1e-7, 0.02, 0.98])
        intended = rs.choice([rs.rand(), orig, 0.0, 1.0, orig + 1e-9])
        eps = rs.choice([5e-6, 2.5e-6, 1e-3])
        bah = int(rs.rand() < 0.3)
        h1, h2 = C.c_int(0), C.c_int(0)
        a = shim.shim_cap_step(intended, orig, eps, C.byref(h1), bah)
        b = O.cnf2o_caplogitchange(intended, orig, eps, C.byref(h2), bah)
        assert a == b and h1.value == h2.value


def test_evidence_slope_is_the_reference_polynomial(shim):
    """The division-free form against the reference's long polynomial (as restated in the oracle) through
    processinfprobs with a zero entropy factor is covered below; here the closed form against a numerical
    derivative of val(x) = (H (1-x) log(1-x) + G x log x) / (H (1-x) + G x)."""
    rs = np.random.RandomState(2)
    for _ in range(200):
        y, x = rs.uniform(0.02, 0.98, 2)
        h = rs.uniform(0.1, 50)
        g = h * rs.uniform(0.01, 0.99)
        G, H = g / y, (h - g) / (1 - y)
        val = lambda t: (H * (1 - t) * np.log(1 - t) + G * t * np.log(t)) / (H * (1 - t) + G * t)
        e = 1e-6
        num = (val(x + e) - val(x - e)) / (2 * e)
        assert abs(shim.shim_evidence_slope(y, g, h, x) - num) < 1e-6 * max(1.0, abs(num))


def test_certainty_update_matches_processinfprobs(shim):
    O = pyoracle.lib()
    rs = np.random.RandomState(3)
    n_assigned = n_two = 0
    for it in range(1500):
        inf = np.where(rs.rand(2) < 0.8, rs.gamma(1.0, 2.0, 2), 0.0)
        if it % 7 == 0:
            inf *= 1e-3
        present = (inf > 0).astype(np.int32)
        side = int(rs.randint(2))
        allele = int(rs.choice([0, 1, 2]))
        sure = float(rs.choice([0.02, 0.0, rs.uniform(0, 0.5), 5e-6]))
        has_prior = int(rs.rand() < 0.7)
        prior_allele = int(rs.choice([0, 1, 2]))
        prior_sure = float(rs.choice([0.02, 0.0, 1.0, rs.uniform(0, 0.5)]))
        empty = int(rs.rand() < 0.1)
        children = int(rs.randint(0, 4))
        sf = float(rs.choice([0.013, 0.05, 0.0, 0.3]))
        ef = float(rs.choice([1.0, 0.5]))
        h1, h2 = C.c_int(0), C.c_int(0)
        na, ns = C.c_int(-1), D(-1.0)
        r1 = shim.shim_update_certainty(_p(inf), side, allele, sure, has_prior, prior_allele, prior_sure, empty, children,
                                        sf, ef, C.byref(h1), C.byref(na), C.byref(ns))
        out = np.zeros(2)
        oa, os_ = C.c_int(-1), D(-1.0)
        r2 = O.cnf2o_processinfprobs(_p(inf), _p(present), side, allele, sure, has_prior, prior_allele, prior_sure, empty,
                                     children, sf, ef, C.byref(h2), _p(out), C.byref(oa), C.byref(os_))
        assert r1 == r2 and h1.value == h2.value
        if r1:
            assert na.value == oa.value
            assert abs(ns.value - os_.value) < 1e-9
            n_assigned += 1
            n_two += int(present.sum() == 2)
    assert n_assigned > 300 and n_two > 100


def test_phase_ratio_matches_relskewhmm(shim):
    O = pyoracle.lib()
    rs = np.random.RandomState(4)
    for M in (1, 2, 9, 60):
        hw = rs.choice([0.5, 0.0, 1.0, 0.3], size=M, p=[0.3, 0.1, 0.1, 0.5]) * rs.choice([1.0, rs.rand()], size=M)
        hw = np.clip(hw, 0, 1)
        rel = rs.choice([0.5, 0.9, 0.99, 0.6], size=M)
        a, b = np.zeros(M), np.zeros(M)
        shim.shim_phase_ratio(_p(hw), _p(rel), 0, M, _p(a))
        O.cnf2o_relskew_ratio(_p(hw), _p(rel), 0, M, _p(b))
        np.testing.assert_allclose(a, b, rtol=1e-12, atol=0, equal_nan=True)
    # with relhaplo = 1/2 everywhere (the only value the PlantImpute path ever holds, cnF2freq.cpp:2496) the ratio
    # is the weight itself
    hw = rs.uniform(0.05, 0.95, 12)
    out = np.zeros(12)
    shim.shim_phase_ratio(_p(hw), _p(np.full(12, 0.5)), 0, 12, _p(out))
    np.testing.assert_allclose(out, hw, rtol=1e-12)


def test_haploweight_update_matches_updatehaploweights(shim):
    O = pyoracle.lib()
    rs = np.random.RandomState(5)
    moved = 0
    for it in range(60):
        cs = np.array([0, 7, 8, 20], np.int32)
        M = 20
        hw = rs.choice([0.5, 0.0, 1.0, 0.2, 0.8], size=M, p=[0.4, 0.1, 0.1, 0.2, 0.2]).astype(np.float64)
        hw = np.where((hw > 0) & (hw < 1), np.clip(hw + rs.uniform(-0.1, 0.1, M), 0.01, 0.99), hw)
        hc = rs.choice([0.0, 1.0, 3.0, 7.5], size=M, p=[0.3, 0.3, 0.2, 0.2])
        if it % 5 == 0:
            hc[7:8] = 0                                     # a chromosome without information is skipped
        hb = hc * rs.uniform(0, 1, M)
        allele = rs.choice([0, 1, 2], size=(M, 2)).astype(np.int32)
        sure = rs.choice([0.02, 0.0, 0.5, 0.3], size=(M, 2))
        rel = np.full(M, 0.5) if it % 2 else rs.choice([0.5, 0.95], size=M)
        children, desc = int(rs.randint(0, 5)), int(rs.randint(1, 9))
        sf = float(rs.choice([0.013, 0.1]))
        A = [x.copy() for x in (hw, hb, hc)]
        B = [x.copy() for x in (hw, hb, hc)]
        h1, h2 = C.c_int(0), C.c_int(0)
        shim.shim_update_haploweights(3, _p(cs), _p(A[0]), _p(A[1]), _p(A[2]), _p(allele), _p(sure), _p(rel), children,
                                      desc, sf, 1.0, C.byref(h1))
        O.cnf2o_updatehaploweights(3, _p(cs), _p(B[0]), _p(B[1]), _p(B[2]), _p(allele), _p(sure), _p(rel), children,
                                   desc, sf, 1.0, C.byref(h2))
        assert h1.value == h2.value
        for x, y in zip(A, B):
            np.testing.assert_allclose(x, y, rtol=1e-9, atol=1e-12)
        locked = (hw == 0) | (hw == 1)
        assert np.array_equal(A[0][locked], hw[locked])
        moved += int(np.abs(A[0] - hw).max() > 1e-6)
    assert moved > 30


def test_scalefactor_control(shim):
    O = pyoracle.lib()
    rs = np.random.RandomState(6)
    old1, old2 = np.zeros(2, np.int32), np.zeros(2, np.int32)
    s1 = s2 = 0.013
    for _ in range(50):
        hits = int(rs.randint(0, 40))
        s1 = shim.shim_adapt_scalefactor(s1, hits, _p(old1), 100)
        s2 = O.cnf2o_scalefactor_step(s2, hits, _p(old2), 100)
        assert s1 == s2 and np.array_equal(old1, old2)


def test_certainty_update_matches_the_reference_goldens(shim):
    """update_certainty against G14: the reference's own processinfprobs on random inputs (cnF2freq.cpp:4179-4323).
    Cases holding a key with value 0 are left out: the product takes an entry as present when it is > 0."""
    z = np.load(os.path.join(GOLDEN_DIR, "update_units.npz"))
    n = 0
    for row, (want_a, want_s, want_hits) in zip(z["pip_in"], z["pip_out"]):
        inf = np.array(row[0:2])
        present = row[2:4].astype(bool)
        if (present & ~(inf > 0)).any():
            continue
        inf = np.where(present, inf, 0.0)
        side, cur, cursure, has_prior, priorval, priorsure, empty, children, sf = row[4:]
        hits = C.c_int(0)
        na, ns = C.c_int(int(cur)), D(cursure)
        shim.shim_update_certainty(_p(inf), int(side), int(cur), float(cursure), int(has_prior),
                                   int(priorval) if has_prior else 0, float(priorsure), int(empty), int(children), float(sf),
                                   1.0, C.byref(hits), C.byref(na), C.byref(ns))
        assert na.value == int(want_a)
        assert abs(ns.value - want_s) < 1e-9
        assert hits.value == int(want_hits)
        n += 1
    assert n > 400


def test_haploweight_update_matches_the_reference_goldens(shim):
    """update_haploweight / phase_ratio against G14: the reference's own updatehaploweights (cnF2freq.cpp:4533-4734)."""
    z = np.load(os.path.join(GOLDEN_DIR, "update_units.npz"))
    cs = np.ascontiguousarray(z["chromstarts"], np.int32)
    rel = np.full(int(cs[-1]), 0.5)
    for x, (children, desc, sf), want, want_hits in zip(z["uhw_in"], z["uhw_meta"], z["uhw_out"], z["uhw_hits"]):
        hw, hb, hc = (np.ascontiguousarray(x[:, k]) for k in range(3))
        a32 = np.ascontiguousarray(x[:, 3:5], np.int32)
        sure = np.ascontiguousarray(x[:, 5:7])
        hits = C.c_int(0)
        shim.shim_update_haploweights(len(cs) - 1, _p(cs), _p(hw), _p(hb), _p(hc), _p(a32), _p(sure), _p(rel), int(children),
                                      int(desc), float(sf), 1.0, C.byref(hits))
        np.testing.assert_allclose(hw, want[:, 0], rtol=1e-9, atol=1e-12, equal_nan=True)
        np.testing.assert_allclose(hb, want[:, 1], rtol=1e-9, atol=1e-12, equal_nan=True)
        np.testing.assert_allclose(hc, want[:, 2], rtol=1e-9, atol=1e-12, equal_nan=True)
        assert hits.value == int(want_hits)


def _random_flow(rs):
    """Parameters of a flow in the regimes a run visits: beliefs near 0 / 1 / anywhere, evidence near or far from the belief."""
    kind = int(rs.randint(2))
    y = float(rs.choice([rs.uniform(0.02, 0.98), 10 ** rs.uniform(-5, -1.5), 1 - 10 ** rs.uniform(-5, -1.5)]))
    h = float(10 ** rs.uniform(-2, 2))
    share = float(np.clip(rs.choice([y * 10 ** rs.uniform(-0.5, 0.5), rs.uniform(0, 1), y]), 1e-9, 1 - 1e-9))
    g = h * share
    if kind == 0:
        e, c0, d, pr = float(rs.choice([1.0, 0.5])), float(rs.choice([0.0, rs.normal() * 3, np.log(0.02 / 0.98), np.log(0.98 / 0.02)])), 0.0, 0.0
    else:
        e, c0 = float(rs.choice([5.0068e-6, 0.04, 0.5, 1.0])), 0.0
        d, pr = float(rs.randint(1, 9)), float(np.clip(rs.choice([y, rs.uniform(0, 1), y + rs.normal() * 1e-3]), 0, 1))
    return kind, y, g, h, e, c0, d, pr


def test_time_bound_is_a_bound(shim):
    """flow_time_bound: whenever it reports a finite value for an interval, (a) the gradient is monotone there with at least
    the slope the bound assumes (checked by differences at 200 points) and keeps its sign, and (b) the 15-point rule over
    the interval does not exceed it."""
    rs = np.random.RandomState(11)
    out = np.zeros(6)
    finite = tight = 0
    for _ in range(30000):
        kind, y, g, h, e, c0, d, pr = _random_flow(rs)
        width = 10 ** rs.uniform(-12, -0.3) * min(y, 1 - y)
        lo = int(rs.randint(2))
        xa, xb = (y - width, y) if lo else (y, y + width)
        if xa <= 1e-6 or xb >= 1 - 1e-6:
            continue
        shim.shim_time_bound(kind, y, g, h, e, c0, d, pr, xa, xb, 0 if lo else 1, _p(out))
        # flow_advance asks only when both ends have the same sign, and a flow moves down exactly when its gradient is negative
        if not np.isfinite(out[0]) or out[4] != 1.0 or (out[5] < 0) != bool(lo):
            continue
        finite += 1
        assert out[3] == 1.0, "the gradient changes sign inside an interval the bound accepted"
        assert out[2] > 0, "the gradient is not monotone on an interval the bound accepted"
        assert out[1] <= out[0] * (1 + 1e-9), (kind, y, g, h, e, c0, d, pr, xa, xb, out)
        tight += int(out[1] > 0.2 * out[0])
    assert finite > 3000 and tight > 300


def test_flows_with_the_bound_equal_flows_without(shim):
    """The bound only spares quadratures whose verdict it knows: results and hit counts are identical to the bit."""
    rs = np.random.RandomState(12)
    a, b, m = np.zeros(8), np.zeros(8), np.zeros(8)
    spared = quads = scout_evals = scout_done = literal_evals = 0
    guided_evals = {4: 0, 5: 0}
    for _ in range(20000):
        kind, y, g, h, e, c0, d, pr = _random_flow(rs)
        eps = 5e-6 / rs.randint(1, 4)
        y = float(np.clip(y, eps, 1 - eps))
        sf = float(rs.choice([0.013, 0.05, 0.2, 0.4]))
        shim.shim_flow(kind, y, g, h, e, c0, d, pr, eps, sf, 1, _p(a))
        shim.shim_flow(kind, y, g, h, e, c0, d, pr, eps, sf, 0, _p(b))
        assert a[0] == b[0] and a[3] == b[3], (kind, y, g, h, e, c0, d, pr, eps, sf, a, b)
        assert a[1] + a[2] == b[1]
        # the same steps one gradient evaluation at a time (FlowRun)
        shim.shim_flow(kind, y, g, h, e, c0, d, pr, eps, sf, 2, _p(m))
        assert np.array_equal(m[:5], a[:5]), (kind, y, g, h, e, c0, d, pr, eps, sf, a, m)
        # the kernels' form: the scout (root memo, one certificate for the spared steps), then replay + literal steps
        shim.shim_flow(kind, y, g, h, e, c0, d, pr, eps, sf, 3, _p(m))
        assert m[0] == b[0] and m[3] == b[3] and m[4] == b[4], (kind, y, g, h, e, c0, d, pr, eps, sf, b, m)
        scout_evals += m[5]
        scout_done += int(m[6] == 0)
        literal_evals += b[4] + 15 * b[1]
        spared += a[2]
        quads += b[1]
        # the guided bisection (facts from literal evaluations, ordered by the monotone gradient): from the start, and after the scout
        for form in (4, 5):
            shim.shim_flow(kind, y, g, h, e, c0, d, pr, eps, sf, form, _p(m))
            assert m[0] == b[0] and m[3] == b[3] and m[4] == b[4], (form, kind, y, g, h, e, c0, d, pr, eps, sf, b, m)
            guided_evals[form] += m[5]
    assert spared > 0.2 * quads
    assert scout_done > 4000, scout_done
    # ... and it is what it is for: under half the gradient evaluations of the literal bisection
    assert guided_evals[5] < 0.5 * literal_evals, (guided_evals, literal_evals)
