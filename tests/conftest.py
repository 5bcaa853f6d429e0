import os
import sys

import numpy as np
import pytest
import torch  # noqa: F401  -- before libcnf2hip.so is loaded: one HIP runtime per process (cnf2freq_amd.capi.hip_runtimes)

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(os.path.dirname(__file__), "golden")
GOLDEN_CASES = ["f2_implicit_f1", "outbred3_missing", "random_windows", "f2_ungenotyped"]


# G13: trajectories (tests/golden/traj_<case>.npz): the four fixtures, a two-chromosome pedigree, tied windows
TRAJ_CASES = GOLDEN_CASES + ["outbred3_two_chrom", "ail_ties"]
# ... and the 10-iteration run (40 analysed x 2 x 100 markers) that takes the step-size control through its "good" and "bad"
# branches (cnF2freq.cpp:6373-6392): compared with the product on the GPU only -- the CPU restatement of HOT LOOP 2 is
# quadratic in the chromosome length (it re-sweeps per marker) and would need the better part of an hour for it
TRAJ_CASES_LONG = ["outbred3_long"]


def build_host_shim():
    """tests/shim/libcnf2hostshim.so: the product's host-side headers (emission tables, window derivation, accumulator
    algebra, update math, the partition plan, the executable's shared-memory transport) behind a C interface for the CPU
    tests.  Linked against libcnf2hip.so for the two C-ABI calls the transport's device path names (never made here)."""
    import ctypes
    import subprocess
    import __graft_entry__ as g
    g.build()
    shim_dir = os.path.join(ROOT, "tests", "shim")
    csrc = os.path.join(ROOT, "cnf2freq_amd", "csrc")
    pkg = os.path.join(ROOT, "cnf2freq_amd")
    so = os.path.join(shim_dir, "libcnf2hostshim.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", "-I" + csrc, "-I" + os.path.join(ROOT, "include"),
                           "-o", so, os.path.join(shim_dir, "host_shim.cpp"), os.path.join(csrc, "cnf2_window.cpp"),
                           "-L" + pkg, "-lcnf2hip", "-Wl,-rpath," + pkg, "-lpthread"])
    return ctypes.CDLL(so)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """(Pedigree rebuilt from the fixture's inputs, dict of reference outputs)."""
    from cnf2freq_amd.synth import Pedigree
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    R = len(z["in_par"])
    ped = Pedigree(["r%d" % i for i in range(R)], z["in_par"], z["in_gen"], z["in_empty"], z["in_row_of"],
                   z["in_allele"], z["in_sure"], z["in_hw"], z["in_pos"], z["in_chromstarts"], z["in_dous"])
    ped.founder_flags()
    return ped, z


def load_trajectory(name):
    """(Pedigree as the readers leave it, dict with the state after postmarkerdata (pm_*) and after every iteration
    (it<k>_*), number of iterations) of the reference's replayed run (tests/golden/make_golden.py, G13)."""
    from cnf2freq_amd.synth import Pedigree
    z = np.load(os.path.join(GOLDEN_DIR, "traj_" + name + ".npz"))
    R = len(z["in_par"])
    ped = Pedigree(["r%d" % i for i in range(R)], z["in_par"], z["in_gen"], z["in_empty"], z["in_row_of"],
                   z["in_allele"], z["in_sure"], z["in_hw"], z["in_pos"], z["in_chromstarts"], z["in_dous"])
    ped.founder_flags()
    n_iter = sum(1 for k in z.files if k.endswith("_scalefactor"))
    return ped, z, n_iter


def pedigree_components(ped):
    """label[R]: records connected through parent links share a label."""
    lab = list(range(ped.n_rec))

    def find(x):
        while lab[x] != x:
            lab[x] = lab[lab[x]]
            x = lab[x]
        return x
    for r in range(ped.n_rec):
        for k in range(2):
            if ped.par[r, k] >= 0:
                lab[find(r)] = find(int(ped.par[r, k]))
    return np.array([find(r) for r in range(ped.n_rec)])


class TrajectoryChecker:
    """Compares a run's state after iteration k with G13.  An element the goldens mark `unstable` (its result is
    rounding noise in the reference itself unless the evidence sums are "round", oracle/pyiter.py) is excused when it
    differs, and then taints its whole pedigree component from that iteration on: tainted records are left out, and
    the hit counters / scale factor -- sums over all records -- are only compared while nothing is tainted.  A
    difference anywhere else fails."""

    def __init__(self, ped, z):
        self.ped, self.z = ped, z
        self.comp = pedigree_components(ped)
        self.tainted = np.zeros(ped.n_rec, bool)
        self.log = []          # per iteration: (k, records compared, records in all, hit counters and scale factor compared)

    def check(self, k, st, rtol=1e-9, atol=1e-12, exact_hits=True, hits_slack=None, values=True):
        """hits_slack: compare the hit counters (within that many) and the scale factor (exactly) even when components are
        tainted -- for long runs, where an excused element moves a counter by a hit or two but must not move the step size.
        values=False: past the horizon up to which two runs that differ by rounding can be compared element by element
        (a knife-edge decision -- a slot whose two certainties are equal but for their last bits takes part in the HAPLOS
        update or does not, cnF2freq.cpp:1224-1239 -- flips and moves its haplobase by O(1)): genotypes must still be
        identical on the untainted records and all but a per cent of their certainties within 1e-6 (measured on the
        10-iteration golden: 99.4 % after the tenth)."""
        z = self.z
        flagged = z["it%d_unstable" % k].astype(bool)
        differs = ~np.isclose(np.asarray(st["sure"]), z["it%d_sure" % k], rtol=rtol, atol=atol) | \
            (np.asarray(st["allele"]) != z["it%d_allele" % k])
        bad = (flagged & differs).any(axis=(1, 2))
        self.tainted |= np.isin(self.comp, np.unique(self.comp[bad]))
        ok = ~self.tainted
        if hits_slack is not None:
            assert np.abs(np.asarray(st["hits"]) - z["it%d_hits" % k]).max() <= hits_slack, (k, st["hits"], z["it%d_hits" % k])
            np.testing.assert_allclose(st["scalefactor"], float(z["it%d_scalefactor" % k]), rtol=1e-12)
        elif not self.tainted.any():
            if exact_hits:
                assert np.array_equal(st["hits"], z["it%d_hits" % k]), (k, st["hits"], z["it%d_hits" % k])
            np.testing.assert_allclose(st["scalefactor"], float(z["it%d_scalefactor" % k]), rtol=1e-15 if exact_hits else 0.25)
        assert np.array_equal(np.asarray(st["allele"])[ok], z["it%d_allele" % k][ok]), k
        if not values:
            close = np.isclose(np.asarray(st["sure"])[ok], z["it%d_sure" % k][ok], rtol=1e-6, atol=1e-9)
            assert close.mean() > 0.99, (k, close.mean())
            self.log.append((k, int(ok.sum()), int(len(ok)), hits_slack is not None))
            return int(ok.sum())
        # haplobase / haplocount as left behind: a slot that is homozygous with EQUAL certainties takes no part in the HAPLOS
        # update (cnF2freq.cpp:1224-1239 compares the two certainties for equality), and certainties that are equal by
        # symmetry are equal to the bit or not depending on the order of the additions behind them: left out
        za, zs = z["it%d_allele" % k], z["it%d_sure" % k]
        knife = (za[..., 0] == za[..., 1]) & np.isclose(zs[..., 0], zs[..., 1], rtol=1e-9, atol=0)
        for key in ("sure", "hw", "haplobase", "haplocount"):
            if key in st:
                keep = ok[:, None] & ~knife if key in ("haplobase", "haplocount") else ok
                # haplobase is rewritten as (hb - c w + c (1 - s) w) / (1 - s) with 1 - s down to 5e-6 (cnF2freq.cpp:4667-4675):
                # rounding of the weight w is amplified by up to 1 / (1 - s)
                np.testing.assert_allclose(np.asarray(st[key])[keep], z["it%d_%s" % (k, key)][keep],
                                           rtol=max(10 * rtol, 1e-8) if key == "haplobase" else rtol, atol=atol,
                                           err_msg="%s after iteration %d" % (key, k))
        self.log.append((k, int(ok.sum()), int(len(ok)), hits_slack is not None or not self.tainted.any()))
        return int(ok.sum())

    def report(self, case, path=None):
        """One line per case: how much of the pedigree was really compared after every iteration."""
        line = "%s: " % case + ", ".join("it%d %d/%d%s" % (k, c, n, "" if full else " (hits not compared)") for k, c, n, full in self.log)
        if path:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "a") as f:
                f.write(line + "\n")
        return line


def oracle_ped(ped):
    from oracle.pyoracle import OraclePed
    a, s, h = ped.dense()
    return OraclePed(a, s, h, ped.par, ped.empty, ped.pos)


def oracle_accumulate_threaded(o, ped, desc, first, last, threads=None):
    """o.accumulate(ped.dous, ...) with the individuals spread over threads: what an individual adds to the per-record
    accumulators does not depend on the others (cnF2freq.cpp:5876-5902 runs per individual), so the slabs of disjoint groups
    add up to the whole list's; homozyg is per individual.  The oracle's C code is reentrant and ctypes drops the GIL."""
    from concurrent.futures import ThreadPoolExecutor
    n = len(ped.dous)
    threads = threads or max(1, min(16, len(os.sched_getaffinity(0)), n))
    chunks = [np.arange(k, n, threads) for k in range(threads)]
    chunks = [c for c in chunks if len(c)]
    gens = ped.gen[ped.dous]

    def work(idx):
        return idx, o.accumulate(ped.dous[idx], gens[idx], desc, first=first, last=last)
    with ThreadPoolExecutor(len(chunks)) as ex:
        parts = list(ex.map(work, chunks))
    out = {k: np.zeros_like(parts[0][1][k]) for k in ("infprobs", "haplobase", "haplocount")}
    nm = last - first + 1
    out["homozyg"] = np.zeros((n, nm, 2))
    for idx, r in parts:
        for k in ("infprobs", "haplobase", "haplocount"):
            out[k] += r[k]
        out["homozyg"][idx] = r["homozyg"]
    return out


@pytest.fixture(params=GOLDEN_CASES)
def golden(request):
    return load_golden(request.param)
