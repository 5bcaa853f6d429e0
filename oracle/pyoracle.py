"""ctypes binding of the CPU oracle (oracle/libcnf2oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (cnf2freq_amd) never imports
this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libcnf2oracle.so")

NUMTYPES, NUMSHIFTS, NUMPATHS = 64, 8, 128
MINFACTOR = float(np.float32(-1e15))


class _Ped(C.Structure):
    _fields_ = [
        ("n_rec", C.c_int), ("n_markers", C.c_int),
        ("allele", C.c_void_p), ("sure", C.c_void_p), ("hw", C.c_void_p),
        ("par", C.c_void_p), ("founder", C.c_void_p), ("empty", C.c_void_p),
        ("pos", C.c_void_p), ("genrec", C.c_double * 3),
        ("correction_inference", C.c_int),
    ]


class _Tree(C.Structure):
    _fields_ = [
        ("shiftignore", C.c_int), ("flag2ignore", C.c_int), ("founder", C.c_int),
        ("n_rel", C.c_int), ("rel_rec", C.c_int * 7), ("rel_map", C.c_int * 7),
        ("rel_mapshift", C.c_int * 7), ("ordered", C.c_int * 7),
    ]


class _EmTab(C.Structure):
    _fields_ = [
        ("c", C.c_double * 2), ("A", (C.c_double * 8) * 2), ("B", (C.c_double * 8) * 2),
        ("cr", C.c_double * 2), ("Ar", (C.c_double * 8) * 2), ("Br", (C.c_double * 8) * 2),
        ("A1", (C.c_double * 8) * 2), ("B1", (C.c_double * 8) * 2),
        ("rootclass", C.c_int * 2),
    ]


class _Fwbw(C.Structure):
    _fields_ = [("n_markers", C.c_int), ("fwbw", C.POINTER(C.c_double)),
                ("factors", C.POINTER(C.c_double))]


def build(force=False):
    """Compile the oracle with gcc (building the checker is not using it)."""
    srcs = [os.path.join(_HERE, f) for f in ("cnf2_oracle.c", "cnf2_oracle_iter.c", "cnf2_oracle_pre.c")]
    srcs = [f for f in srcs if os.path.exists(f)]
    if (not force and os.path.exists(_LIB)
            and os.path.getmtime(_LIB) >= max([os.path.getmtime(f) for f in srcs] +
                                              [os.path.getmtime(os.path.join(_HERE, "cnf2_oracle.h"))])):
        return _LIB
    subprocess.check_call(["gcc", "-O2", "-std=c99", "-fopenmp", "-fPIC", "-shared",
                           "-o", _LIB] + srcs + ["-lm"], cwd=_HERE)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        PP = C.POINTER(_Ped)
        L.cnf2o_fixtrees.argtypes = [PP, C.c_int, C.POINTER(_Tree)]
        L.cnf2o_founder_flags.argtypes = [PP, C.c_void_p]
        L.cnf2o_ignoreflag2.argtypes = [PP, C.POINTER(_Tree), C.c_int, C.c_int, C.c_int, C.c_int]
        L.cnf2o_ignoreflag2.restype = C.c_int
        L.cnf2o_trackpossible.argtypes = [PP, C.c_int, C.c_int, C.c_double, C.c_int, C.c_uint,
                                          C.c_int, C.c_int, C.c_uint, C.c_int, C.POINTER(C.c_int)]
        L.cnf2o_trackpossible.restype = C.c_double
        L.cnf2o_emission.argtypes = [PP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.cnf2o_emission.restype = C.c_double
        L.cnf2o_mapval.argtypes = [PP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.POINTER(C.c_double)]
        L.cnf2o_mapval.restype = C.c_int
        L.cnf2o_fwbw_new.argtypes = [C.c_int]
        L.cnf2o_fwbw_new.restype = C.POINTER(_Fwbw)
        L.cnf2o_fwbw_free.argtypes = [C.POINTER(_Fwbw)]
        L.cnf2o_initfwbw.argtypes = [PP, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_Fwbw)]
        L.cnf2o_total.argtypes = [PP, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_Fwbw), C.c_double]
        L.cnf2o_total.restype = C.c_double
        L.cnf2o_query.argtypes = [PP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.POINTER(_Fwbw), C.c_double]
        L.cnf2o_query.restype = C.c_double
        L.cnf2o_turn_query.argtypes = [PP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.POINTER(_Fwbw), C.c_double]
        L.cnf2o_turn_query.restype = C.c_double
        L.cnf2o_sweep_ind.argtypes = [PP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_int, C.POINTER(_Fwbw)]
        L.cnf2o_sweep_ind.restype = C.c_int
        L.cnf2o_emission_tables.argtypes = [PP, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_EmTab)]
        L.cnf2o_val_table.argtypes = [PP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.cnf2o_haplos_row.argtypes = [PP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.cnf2o_infprobs_row.argtypes = [PP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.cnf2o_addvariance.argtypes = [PP, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.cnf2o_descendants.argtypes = [PP, C.c_void_p]
        L.cnf2o_accumulate.argtypes = [PP, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cnf2o_sweep_batch.argtypes = [PP, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.cnf2o_sweep_batch.restype = C.c_int
        D = C.c_double
        IP = C.POINTER(C.c_int)
        L.cnf2o_caplogitchange.argtypes = [D, D, D, IP, C.c_int]
        L.cnf2o_caplogitchange.restype = D
        L.cnf2o_processinfprobs.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, D, C.c_int, C.c_int, D, C.c_int,
                                            C.c_int, D, D, IP, C.c_void_p, IP, C.POINTER(D)]
        L.cnf2o_processinfprobs.restype = C.c_int
        L.cnf2o_relskew_ratio.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.cnf2o_updatehaploweights.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_int, C.c_int, D, D, IP]
        L.cnf2o_scalefactor_step.argtypes = [D, C.c_int, C.c_void_p, C.c_int]
        L.cnf2o_scalefactor_step.restype = D
        L.cnf2o_gauss15_reciprocal_linear.argtypes = [D, D, D, D]
        L.cnf2o_gauss15_reciprocal_linear.restype = D
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class OraclePed:
    """Holds numpy arrays alive and exposes the oracle calls on them.

    allele int32 [R,M,2]; sure f64 [R,M,2]; hw f64 [R,M]; par int32 [R,2];
    founder/empty uint8 [R]; pos f64 [M].
    """

    def __init__(self, allele, sure, hw, par, empty, pos, founder=None,
                 genrec=(-0.02, -0.02, -0.02), correction_inference=False,
                 apply_founder_flags=True):
        self.allele = np.ascontiguousarray(allele, dtype=np.int32)
        self.sure = np.ascontiguousarray(sure, dtype=np.float64)
        self.hw = np.ascontiguousarray(hw, dtype=np.float64)
        self.par = np.ascontiguousarray(par, dtype=np.int32)
        self.empty = np.ascontiguousarray(empty, dtype=np.uint8)
        self.pos = np.ascontiguousarray(pos, dtype=np.float64)
        R, M = self.hw.shape
        assert self.allele.shape == (R, M, 2) and self.sure.shape == (R, M, 2)
        assert self.par.shape == (R, 2) and self.pos.shape == (M,)
        self.founder = (np.zeros(R, np.uint8) if founder is None
                        else np.ascontiguousarray(founder, dtype=np.uint8))
        self.R, self.M = R, M
        self.c = _Ped(R, M, _ptr(self.allele), _ptr(self.sure), _ptr(self.hw), _ptr(self.par),
                      _ptr(self.founder), _ptr(self.empty), _ptr(self.pos),
                      (C.c_double * 3)(*genrec), int(correction_inference))
        if apply_founder_flags:
            out = np.zeros(R, np.uint8)
            lib().cnf2o_founder_flags(C.byref(self.c), _ptr(out))
            self.founder[:] = out

    # -- window topology
    def fixtrees(self, ind):
        t = _Tree()
        lib().cnf2o_fixtrees(C.byref(self.c), ind, C.byref(t))
        return t

    def ignoreflag2(self, tree, flag2, g, shift, marker):
        return lib().cnf2o_ignoreflag2(C.byref(self.c), C.byref(tree), flag2, g, shift, marker)

    # -- emission
    def emission(self, ind, marker, g, flag2, shift):
        return lib().cnf2o_emission(C.byref(self.c), ind, marker, g, flag2, shift)

    def mapval(self, ind, marker, g, flag2, shift):
        v = C.c_double()
        mv = lib().cnf2o_mapval(C.byref(self.c), ind, marker, g, flag2, shift, C.byref(v))
        return mv, v.value

    def emission_tables(self, ind, marker, shift, flag2ignore=0):
        t = _EmTab()
        lib().cnf2o_emission_tables(C.byref(self.c), ind, marker, shift, flag2ignore, C.byref(t))
        conv = lambda x: np.array([list(r) for r in x])
        return dict(c=np.array(list(t.c)), A=conv(t.A), B=conv(t.B), cr=np.array(list(t.cr)),
                    Ar=conv(t.Ar), Br=conv(t.Br), A1=conv(t.A1), B1=conv(t.B1),
                    rootclass=list(t.rootclass))

    # -- per-individual body
    def sweep_ind(self, ind, gen=2, first=0, last=None, mode=0, dosage=True, keep_store=False):
        last = self.M - 1 if last is None else last
        factors = np.zeros(NUMSHIFTS)
        factor = np.zeros(1)
        nm = last - first + 1
        dos = np.zeros((nm, 3)) if dosage else None
        W = lib().cnf2o_fwbw_new(self.M) if keep_store else None
        ok = lib().cnf2o_sweep_ind(C.byref(self.c), ind, gen, first, last, _ptr(factors), _ptr(factor),
                                   _ptr(dos) if dosage else None, mode, W)
        res = dict(ok=bool(ok), factors=factors, factor=float(factor[0]), dosage=dos)
        if keep_store:
            fw = np.ctypeslib.as_array(W.contents.fwbw, shape=(NUMSHIFTS, self.M, 3, NUMTYPES)).copy()
            ff = np.ctypeslib.as_array(W.contents.factors, shape=(NUMSHIFTS, self.M, 3)).copy()
            lib().cnf2o_fwbw_free(W)
            res.update(fwbw=fw, fwbwfactors=ff)
        return res

    def turn_scan(self, ind, marker, gen=2, first=0, last=None):
        """rawervals[turn][s] of cpp:5686-5752 for one marker (NaN where skipped)."""
        last = self.M - 1 if last is None else last
        W = lib().cnf2o_fwbw_new(self.M)
        factors = np.zeros(NUMSHIFTS)
        factor = np.zeros(1)
        lib().cnf2o_sweep_ind(C.byref(self.c), ind, gen, first, last, _ptr(factors), _ptr(factor),
                              None, 0, W)
        t = self.fixtrees(ind)
        shiftend = NUMSHIFTS if gen >= 2 else 2
        out = np.full((128, NUMSHIFTS), np.nan)
        for turn in range(128):
            if turn & (t.flag2ignore >> 1):
                continue
            for s in range(shiftend):
                if s & t.shiftignore:
                    continue
                out[turn, s] = lib().cnf2o_turn_query(C.byref(self.c), ind, s, first, last, marker,
                                                      turn, W, -50000 + factor[0]) - factor[0]
        lib().cnf2o_fwbw_free(W)
        return out

    def val_table(self, ind, marker, gen=2, first=0, last=None):
        """(val[8][64][128], mapval[8][64][128]) of HOT LOOP 2 at one marker; -1 = term skipped."""
        last = self.M - 1 if last is None else last
        v = np.zeros((NUMSHIFTS, NUMTYPES, NUMPATHS))
        mv = np.zeros((NUMSHIFTS, NUMTYPES, NUMPATHS), np.int32)
        lib().cnf2o_val_table(C.byref(self.c), ind, gen, first, last, marker, _ptr(v), _ptr(mv))
        return v, mv

    def haplos_row(self, ind, marker, gen=2, first=0, last=None):
        """haplos[n_rec][2] accumulated by HOT LOOP 2 at one marker (before movehaplos)."""
        last = self.M - 1 if last is None else last
        h = np.zeros((self.R, 2))
        lib().cnf2o_haplos_row(C.byref(self.c), ind, gen, first, last, marker, _ptr(h))
        return h

    def infprobs_row(self, ind, marker, gen=2, first=0, last=None):
        """infprobs [n_rec][allele index][markerval - 1] and the homozyg[2] increments of HOT LOOP 2 at one
        marker (before moveinfprobs)."""
        last = self.M - 1 if last is None else last
        inf = np.zeros((self.R, 2, 2))
        hz = np.zeros(2)
        lib().cnf2o_infprobs_row(C.byref(self.c), ind, gen, first, last, marker, _ptr(inf), _ptr(hz))
        return inf, hz

    def descendants(self):
        d = np.zeros(self.R, np.int32)
        lib().cnf2o_descendants(C.byref(self.c), _ptr(d))
        return d

    def accumulate(self, inds, gens, desc, first=0, last=None):
        """HOT LOOP 2 + moveinfprobs/movehaplos for a list of individuals: dict(infprobs[R][nm][2][2],
        haplobase[R][nm], haplocount[R][nm], homozyg[n][nm][2])."""
        last = self.M - 1 if last is None else last
        nm = last - first + 1
        inds = np.ascontiguousarray(inds, np.int32)
        gens = np.ascontiguousarray(gens, np.int32)
        desc = np.ascontiguousarray(desc, np.int32)
        inf = np.zeros((self.R, nm, 2, 2)); hb = np.zeros((self.R, nm)); hc = np.zeros((self.R, nm))
        hz = np.zeros((len(inds), nm, 2))
        lib().cnf2o_accumulate(C.byref(self.c), _ptr(inds), _ptr(gens), len(inds), first, last, _ptr(desc),
                               _ptr(inf), _ptr(hb), _ptr(hc), _ptr(hz))
        return dict(infprobs=inf, haplobase=hb, haplocount=hc, homozyg=hz)

    def addvariance(self, rec, marker, flag2ignore):
        """variances[marker] as individ::addvariance sets it (cpp:1489-1558); None when it is left alone."""
        v = C.c_double(0.0)
        ok = lib().cnf2o_addvariance(C.byref(self.c), rec, marker, flag2ignore, C.byref(v))
        return v.value if ok else None

    def sweep_batch(self, inds, gens=None, first=0, last=None, mode=2, dosage=True, n_threads=0):
        last = self.M - 1 if last is None else last
        inds = np.ascontiguousarray(inds, dtype=np.int32)
        gens = (np.full(len(inds), 2, np.int32) if gens is None
                else np.ascontiguousarray(gens, dtype=np.int32))
        n = len(inds)
        nm = last - first + 1
        factors = np.zeros((n, NUMSHIFTS))
        factor = np.zeros(n)
        dos = np.zeros((n, nm, 3)) if dosage else None
        used = lib().cnf2o_sweep_batch(C.byref(self.c), _ptr(inds), _ptr(gens), n, first, last,
                                       _ptr(factors), _ptr(factor), _ptr(dos) if dosage else None,
                                       mode, n_threads)
        return dict(factors=factors, factor=factor, dosage=dos, threads=used)
