"""ctypes binding of oracle/_ref/libcnf2ref*.so (the reference's own hot-path code
behind our driver, see build_ref.sh).  TEST INFRASTRUCTURE ONLY; exists only where
the .so has been built (the build container, or a GPU box that received the prebuilt
file).  Used to pin oracle/cnf2_oracle.c and to generate tests/golden/."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(_HERE, "..", "_ref")


def available(ieee=True):
    return os.path.exists(os.path.join(REF_DIR, "libcnf2ref_ieee.so" if ieee else "libcnf2ref.so"))


_libs = {}


def lib(ieee=True):
    key = bool(ieee)
    if key not in _libs:
        os.environ.setdefault("OMP_STACKSIZE", "128M")  # demo.sh:36
        L = C.CDLL(os.path.join(REF_DIR, "libcnf2ref_ieee.so" if ieee else "libcnf2ref.so"))
        L.ref_reset.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.ref_add_ind.argtypes = [C.c_int] * 5
        L.ref_set_marker.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double]
        L.ref_set_founder.argtypes = [C.c_int, C.c_int]
        L.ref_get_founder.argtypes = [C.c_int]
        L.ref_fixtrees.argtypes = [C.c_int, C.c_void_p]
        L.ref_trackpossible.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_uint, C.c_int, C.c_int,
                                        C.c_uint, C.c_int, C.POINTER(C.c_int)]
        L.ref_trackpossible.restype = C.c_double
        L.ref_emission.argtypes = [C.c_int] * 5
        L.ref_emission.restype = C.c_double
        L.ref_mapval.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_double)]
        L.ref_ignoreflag2.argtypes = [C.c_int] * 4
        L.ref_sweep.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_query.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double]
        L.ref_query.restype = C.c_double
        L.ref_turn_query.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double]
        L.ref_turn_query.restype = C.c_double
        L.ref_dosage_rows.argtypes = [C.c_void_p]
        L.ref_haplos_row.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.ref_infprobs_row.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ref_addvariance.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.ref_accumulate.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_sweep_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.ref_postmarkerdata.argtypes = [C.c_int]
        L.ref_get_marker.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.ref_get_counts.argtypes = [C.c_int, C.c_void_p]
        L.ref_get_variance.argtypes = [C.c_int, C.c_int]
        L.ref_get_variance.restype = C.c_double
        L.ref_set_prior.argtypes = [C.c_int]
        L.ref_set_dous.argtypes = [C.c_void_p, C.c_int]
        L.ref_set_stepsize.argtypes = [C.c_double, C.c_int, C.c_int]
        L.ref_get_stepsize.argtypes = [C.c_void_p]
        L.ref_get_haplo_accumulators.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.ref_set_haplo_accumulators.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double]
        L.ref_set_counts.argtypes = [C.c_int, C.c_int, C.c_int]
        L.ref_set_infprobs.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_caplogitchange.argtypes = [C.c_double, C.c_double, C.c_double, C.POINTER(C.c_int), C.c_int]
        L.ref_caplogitchange.restype = C.c_double
        L.ref_processinfprobs.argtypes = [C.c_int, C.c_int, C.c_int]
        L.ref_updatehaploweights.argtypes = [C.c_int]
        L.ref_gauss15_reciprocal_linear.argtypes = [C.c_double] * 4
        L.ref_gauss15_reciprocal_linear.restype = C.c_double
        L.ref_iteration.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.ref_copy_prior.argtypes = [C.c_int] * 4
        L.ref_set_empty.argtypes = [C.c_int, C.c_int]
        _libs[key] = L
    return _libs[key]


class RefPed:
    """Loads a cnf2freq_amd.synth.Pedigree into the reference's `individ` graph.
    Record r maps to the reference's individual number r+1."""

    def __init__(self, ped, ieee=True, fixtrees_all=True):
        self.L = lib(ieee)
        self.ped = ped
        pos = np.ascontiguousarray(ped.pos, np.float64)
        cs = np.ascontiguousarray(ped.chromstarts, np.int32)
        self.L.ref_reset(len(pos), pos.ctypes.data, cs.ctypes.data, len(cs))
        a, s, h = ped.dense()
        for r in range(ped.n_rec):
            self.L.ref_add_ind(r + 1, int(ped.par[r, 0]) + 1, int(ped.par[r, 1]) + 1, int(ped.gen[r]),
                               int(ped.empty[r]))
        for r in range(ped.n_rec):
            for m in range(ped.n_markers):
                self.L.ref_set_marker(r + 1, m, int(a[r, m, 0]), int(a[r, m, 1]), float(s[r, m, 0]),
                                      float(s[r, m, 1]), float(h[r, m]))
        if fixtrees_all:
            self.L.ref_fixtrees_all()
        self.M = ped.n_markers

    def founder(self):
        return np.array([self.L.ref_get_founder(r + 1) for r in range(self.ped.n_rec)], np.uint8)

    def fixtrees(self, rec):
        out = np.zeros(32, np.int32)
        self.L.ref_fixtrees(rec + 1, out.ctypes.data)
        n = int(out[2])
        rel = out[3:3 + 3 * n].reshape(n, 3).copy()
        rel[:, 0] -= 1
        return dict(shiftignore=int(out[0]), flag2ignore=int(out[1]), rel=rel, ordered=out[24:31] - 1)

    def emission(self, rec, marker, g, flag2, shift):
        return self.L.ref_emission(rec + 1, marker, g, flag2, shift)

    def mapval(self, rec, marker, g, flag2, shift):
        v = C.c_double()
        mv = self.L.ref_mapval(rec + 1, marker, g, flag2, shift, C.byref(v))
        return mv, v.value

    def sweep(self, rec, gen=2, first=0, last=None, store=True):
        last = self.M - 1 if last is None else last
        factors = np.zeros(8)
        factor = np.zeros(1)
        fw = np.zeros((8, self.M, 3, 64)) if store else None
        ff = np.zeros((8, self.M, 3)) if store else None
        ok = self.L.ref_sweep(rec + 1, gen, first, last, factors.ctypes.data, factor.ctypes.data,
                              fw.ctypes.data if store else None, ff.ctypes.data if store else None)
        return dict(ok=bool(ok), factors=factors, factor=float(factor[0]), fwbw=fw, fwbwfactors=ff,
                    first=first, last=last)

    def query(self, marker, g, flag2, shift, minfactor):
        return self.L.ref_query(marker, g, flag2, shift, minfactor)

    def turn_query(self, marker, turn, shift, minfactor):
        return self.L.ref_turn_query(marker, turn, shift, minfactor)

    def dosage_rows(self, n_rows):
        rows = np.zeros((n_rows, 3))
        self.L.ref_dosage_rows(rows.ctypes.data)
        return rows

    def haplos_row(self, marker):
        """HAPLOS accumulators at one marker for the individual of the last sweep(): [n_rec][2]."""
        h = np.zeros((self.ped.n_rec, 2))
        self.L.ref_haplos_row(marker, self.ped.n_rec, h.ctypes.data)
        return h

    def infprobs_row(self, marker):
        """infprobs [n_rec][allele index][markerval - 1] and homozyg[2] left by HOT LOOP 2 at one marker for
        the individual of the last sweep()."""
        inf = np.zeros((self.ped.n_rec, 2, 2))
        hz = np.zeros(2)
        self.L.ref_infprobs_row(marker, self.ped.n_rec, inf.ctypes.data, hz.ctypes.data)
        return inf, hz

    def accumulate(self, recs, gens, desc, first=0, last=None):
        last = self.M - 1 if last is None else last
        nm = last - first + 1
        ns = np.ascontiguousarray(np.asarray(recs) + 1, np.int32)
        gens = np.ascontiguousarray(gens, np.int32)
        desc = np.ascontiguousarray(desc, np.int32)
        R = self.ped.n_rec
        inf = np.zeros((R, nm, 2, 2)); hb = np.zeros((R, nm)); hc = np.zeros((R, nm)); hz = np.zeros((len(ns), nm, 2))
        self.L.ref_accumulate(ns.ctypes.data, gens.ctypes.data, len(ns), first, last, desc.ctypes.data, R,
                              inf.ctypes.data, hb.ctypes.data, hc.ctypes.data, hz.ctypes.data)
        return dict(infprobs=inf, haplobase=hb, haplocount=hc, homozyg=hz)

    def postmarkerdata(self):
        """The reference's own postmarkerdata (cnF2freq.cpp:3190-3412) as main() calls it (8083-8085), on the loaded
        pedigree (load it with fixtrees_all=False: main runs it straight after the readers).  Returns the state it
        leaves: allele [R][M][2], sure [R][M][2], hw [R][M], descendants / children / founder [R], variances [R][M]."""
        R, M = self.ped.n_rec, self.M
        self.L.ref_postmarkerdata(R + 1)
        allele = np.zeros((R, M, 2), np.int32)
        sure = np.zeros((R, M, 2))
        hw = np.zeros((R, M))
        var = np.zeros((R, M))
        cnt = np.zeros((R, 3), np.int32)
        buf = np.zeros(5)
        for r in range(R):
            self.L.ref_get_counts(r + 1, cnt[r].ctypes.data)
            for m in range(M):
                self.L.ref_get_marker(r + 1, m, buf.ctypes.data)
                allele[r, m] = buf[:2]
                sure[r, m] = buf[2:4]
                hw[r, m] = buf[4]
                var[r, m] = self.L.ref_get_variance(r + 1, m)
        return dict(allele=allele, sure=sure, hw=hw, descendants=cnt[:, 0].copy(), children=cnt[:, 1].copy(),
                    founder=cnt[:, 2].copy(), variances=var)

    def addvariance(self, rec, marker, flag2ignore):
        v = C.c_double(0.0)
        ok = self.L.ref_addvariance(rec + 1, marker, flag2ignore, C.byref(v))
        return v.value if ok else None

    def sweep_batch(self, recs, first=0, last=None, threads=0):
        last = self.M - 1 if last is None else last
        ns = np.ascontiguousarray(np.asarray(recs) + 1, np.int32)
        f = np.zeros((len(ns), 8))
        used = self.L.ref_sweep_batch(ns.ctypes.data, len(ns), first, last, threads, f.ctypes.data)
        return f, used

    # ---- the per-iteration updates (the reference's own cnF2freq.cpp:4004-4734 behind ref_driver.inc) ----
    def set_priors(self, has_prior=None):
        """readalphadata's copy of the rows as read (cnF2freq.cpp:6664-6665) for the genotyped records."""
        hp = (1 - np.asarray(self.ped.empty)) if has_prior is None else np.asarray(has_prior)
        for r in range(self.ped.n_rec):
            if hp[r]:
                self.L.ref_set_prior(r + 1)

    def set_dous(self, recs=None):
        ns = np.ascontiguousarray(np.asarray(self.ped.dous if recs is None else recs) + 1, np.int32)
        self.L.ref_set_dous(ns.ctypes.data, len(ns))

    def rows(self):
        """allele [R][M][2], sure [R][M][2], hw [R][M], haplobase [R][M], haplocount [R][M] as they stand."""
        R, M = self.ped.n_rec, self.M
        allele = np.zeros((R, M, 2), np.int32)
        sure = np.zeros((R, M, 2))
        hw = np.zeros((R, M))
        hb = np.zeros((R, M))
        hc = np.zeros((R, M))
        buf = np.zeros(5)
        acc = np.zeros(2)
        for r in range(R):
            for m in range(M):
                self.L.ref_get_marker(r + 1, m, buf.ctypes.data)
                allele[r, m] = buf[:2]
                sure[r, m] = buf[2:4]
                hw[r, m] = buf[4]
                self.L.ref_get_haplo_accumulators(r + 1, m, acc.ctypes.data)
                hb[r, m], hc[r, m] = acc
        return dict(allele=allele, sure=sure, hw=hw, haplobase=hb, haplocount=hc)

    def stepsize(self):
        out = np.zeros(3)
        self.L.ref_get_stepsize(out.ctypes.data)
        return float(out[0]), int(out[1]), int(out[2])

    def iteration(self):
        """One doit<false, genotypereporter> (cnF2freq.cpp:8132): returns the state it leaves plus hits[C] and the
        scale factor, and the mask of elements whose result is rounding noise in the reference itself (see ref_iteration)."""
        hits = np.zeros(len(self.ped.chromstarts) - 1, np.int32)
        unstable = np.zeros((self.ped.n_rec, self.M, 2), np.uint8)
        self.L.ref_iteration(hits.ctypes.data, unstable.ctypes.data, self.ped.n_rec)
        st = self.rows()
        st["hits"] = hits
        st["unstable"] = unstable
        st["scalefactor"] = self.stepsize()[0]
        return st
