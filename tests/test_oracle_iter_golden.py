"""CPU suite: the restatement of the per-iteration updates (oracle/cnf2_oracle_iter.c, strung into iterations by
oracle/pyiter.py) against goldens produced by the reference's own update functions (cnF2freq.cpp:4004-4734 compiled
verbatim into oracle/_ref; the one stand-in is oracle/ref_extract/boost_gauss_shim.h):
  G14 update_units.npz   caplogitchange / processinfprobs / updatehaploweights on random inputs -- BIT-EXACT
  G13 traj_<case>.npz    readers -> postmarkerdata -> 3 iterations -- hit counters and scale factors identical, values to
                         1e-9 (bit-exact wherever the sweep restatement's accumulators are)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, TRAJ_CASES, TrajectoryChecker, load_trajectory
from oracle import pyoracle
from oracle.pyiter import OracleRun


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope="module")
def units():
    return np.load(os.path.join(GOLDEN_DIR, "update_units.npz"))


def test_caplogitchange_bit_exact(units):
    O = pyoracle.lib()
    for intended, orig, eps, bh, want, want_hits in units["cap"]:
        h = C.c_int(0)
        got = O.cnf2o_caplogitchange(intended, orig, eps, C.byref(h), int(bh))
        assert got == want or (np.isnan(got) and np.isnan(want))
        assert h.value == int(want_hits)


def test_processinfprobs_bit_exact(units):
    O = pyoracle.lib()
    seen_hits = 0
    for row, (want_a, want_s, want_hits) in zip(units["pip_in"], units["pip_out"]):
        inf = np.array(row[0:2])
        present = np.array(row[2:4], np.int32)
        side, cur, cursure, has_prior, priorval, priorsure, empty, children, sf = row[4:]
        out = np.zeros(2)
        oa, os_ = C.c_int(int(cur)), C.c_double(cursure)
        hits = C.c_int(0)
        O.cnf2o_processinfprobs(_p(inf), _p(present), int(side), int(cur), float(cursure), int(has_prior), int(priorval),
                                float(priorsure), int(empty), int(children), float(sf), 1.0, C.byref(hits), _p(out),
                                C.byref(oa), C.byref(os_))
        assert oa.value == int(want_a)
        assert os_.value == want_s, (row, os_.value, want_s)
        assert hits.value == int(want_hits)
        seen_hits += hits.value
    assert seen_hits > 0


def test_updatehaploweights_bit_exact(units):
    O = pyoracle.lib()
    cs = np.ascontiguousarray(units["chromstarts"], np.int32)
    M = int(cs[-1])
    relhaplo = np.full(M, 0.5)
    moved = 0
    for x, (children, desc, sf), want, want_hits in zip(units["uhw_in"], units["uhw_meta"], units["uhw_out"],
                                                        units["uhw_hits"]):
        hw, hb, hc = (np.ascontiguousarray(x[:, k]) for k in range(3))
        a32 = np.ascontiguousarray(x[:, 3:5], np.int32)
        sure = np.ascontiguousarray(x[:, 5:7])
        hits = C.c_int(0)
        O.cnf2o_updatehaploweights(len(cs) - 1, _p(cs), _p(hw), _p(hb), _p(hc), _p(a32), _p(sure), _p(relhaplo),
                                   int(children), int(desc), float(sf), 1.0, C.byref(hits))
        assert np.array_equal(hw, want[:, 0], equal_nan=True)
        assert np.array_equal(hb, want[:, 1], equal_nan=True)
        assert np.array_equal(hc, want[:, 2], equal_nan=True)
        assert hits.value == int(want_hits)
        moved += int(np.nanmax(np.abs(hw - x[:, 0])) > 1e-6)
    assert moved > len(units["uhw_in"]) // 2


@pytest.mark.parametrize("case", TRAJ_CASES)
def test_trajectory_matches_reference(case):
    ped, z, n_iter = load_trajectory(case)
    a, s, _ = ped.dense()
    run = OracleRun(ped, z["pm_allele"], z["pm_sure"], z["pm_hw"], a, s, 1 - ped.empty, z["pm_descendants"])
    chk = TrajectoryChecker(ped, z)
    compared = 0
    for k in range(1, n_iter + 1):
        compared = chk.check(k, run.iteration())
    # random_windows holds ill-conditioned elements (see oracle/pyiter.py); every other case is compared in full
    assert compared == ped.n_rec or case == "random_windows"
    assert compared >= ped.n_rec // 2


def test_quadrature_tables_are_leggauss15():
    """The rule both the restatement and the shim of oracle/_ref tabulate is numpy's 15-point Gauss-Legendre rule."""
    O = pyoracle.lib()
    x, w = np.polynomial.legendre.leggauss(15)
    for slope, icpt, a, b in ((2.0, 1.0, 0.1, 0.7), (-3.0, 5.0, 0.2, 0.9), (40.0, 0.5, 0.0, 1.0)):
        t = 0.5 * (a + b) + 0.5 * (b - a) * x
        want = 0.5 * (b - a) * np.sum(w / (slope * t + icpt))
        got = O.cnf2o_gauss15_reciprocal_linear(slope, icpt, a, b)
        assert abs(got - want) <= 1e-14 * abs(want)
        try:
            from oracle.ref_extract import pyref
        except Exception:
            continue
        if pyref.available():
            # same nodes, same summation order: the restatement and the stand-in agree to the bit
            assert pyref.lib().ref_gauss15_reciprocal_linear(slope, icpt, a, b) == got
