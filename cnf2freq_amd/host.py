"""ctypes binding of libcnf2host.so (include/cnf2host.h): the host side of a cnF2freq run -- postmarkerdata, the
haplotyping iteration with its device-side updates, dump / deserialize -- driven from arrays.  Same code as the
`cnF2freq` executable; everything numeric goes on to libcnf2hip.so (no CPU compute path here either)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CNF2HOST_LIB: a copy of the host library next to another build of libcnf2hip.so (kernel A/B timing: it binds the libcnf2hip.so of its own directory)
LIB_PATH = os.environ.get("CNF2HOST_LIB") or os.path.join(_HERE, "libcnf2host.so")

SYMBOLS = ["cnf2h_create", "cnf2h_create_on", "cnf2h_destroy", "cnf2h_last_error", "cnf2h_postmarkerdata", "cnf2h_iteration",
           "cnf2h_dump", "cnf2h_deserialize", "cnf2h_get_state", "cnf2h_set_block", "cnf2h_balanced_block", "cnf2h_set_partition", "cnf2h_get_partition", "cnf2h_set_update_flags", "cnf2h_reserve", "cnf2h_get_timing",
           "cnf2h_set_deterministic", "cnf2h_context", "cnf2h_get_passes"]

# int fn(void *user, int op, void *buf, size_t count, size_t seg) -- the transport of a multi-process run (cnf2host.h)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t)
X_SUM_SEGMENTS, X_SUM_HITS, X_GATHER_SEGMENTS, X_BARRIER, X_BCAST_HOST = 0, 1, 2, 3, 4

_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libcnf2host.so is not built: run `make -C cnf2freq_amd/csrc`")
        L = C.CDLL(LIB_PATH)
        vp, i32 = C.c_void_p, C.c_int
        L.cnf2h_create.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, vp, i32, i32]
        L.cnf2h_create.restype = vp
        L.cnf2h_create_on.argtypes = [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, vp, i32, i32]
        L.cnf2h_create_on.restype = vp
        L.cnf2h_set_block.argtypes = [vp, i32, i32]
        L.cnf2h_balanced_block.argtypes = [vp, i32, i32, vp, vp]
        L.cnf2h_set_partition.argtypes = [vp, i32, i32, EXCHANGE_FN, vp]
        L.cnf2h_get_partition.argtypes = [vp, vp, vp]
        L.cnf2h_set_update_flags.argtypes = [vp, C.c_uint32]
        L.cnf2h_reserve.argtypes = [vp]
        L.cnf2h_get_timing.argtypes = [vp, vp]
        L.cnf2h_set_deterministic.argtypes = [vp, i32]
        L.cnf2h_context.argtypes = [vp]
        L.cnf2h_context.restype = vp
        L.cnf2h_get_passes.argtypes = [vp, vp, vp, vp]
        L.cnf2h_destroy.argtypes = [vp]
        L.cnf2h_destroy.restype = None
        L.cnf2h_last_error.restype = C.c_char_p
        L.cnf2h_postmarkerdata.argtypes = [vp, i32]
        L.cnf2h_iteration.argtypes = [vp, C.c_char_p, i32]
        L.cnf2h_dump.argtypes = [vp, C.c_char_p, i32]
        L.cnf2h_deserialize.argtypes = [vp, C.c_char_p]
        L.cnf2h_get_state.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Run:
    """One run over a cnf2freq_amd.synth.Pedigree (records that are not `empty` count as genotyped, i.e. they have
    priors, unless has_prior is given)."""

    def __init__(self, ped, has_prior=None, quiet=True, device=0):
        self.L = load()
        if np.array_equal(ped.row_of, np.arange(1, ped.n_rec + 1)):
            a, s, h = ped.allele[1:], ped.sure[1:], ped.hw[1:]       # one row per record already: no copies (bench-scale inputs)
        else:
            a, s, h = ped.dense()
        self.n_rec, self.M = ped.n_rec, ped.n_markers
        hp = (1 - np.asarray(ped.empty)).astype(np.uint8) if has_prior is None else np.ascontiguousarray(has_prior, np.uint8)
        args = [np.ascontiguousarray(ped.par, np.int32), np.ascontiguousarray(ped.empty, np.uint8),
                np.ascontiguousarray(ped.gen, np.int32), hp, np.ascontiguousarray(a, np.uint8),
                np.ascontiguousarray(s, np.float64), np.ascontiguousarray(h, np.float64),
                np.ascontiguousarray(ped.pos, np.float64)]
        cs = np.ascontiguousarray(ped.chromstarts, np.int32)
        dous = np.ascontiguousarray(ped.dous, np.int32)
        self.n_chrom = len(cs) - 1
        self.h = self.L.cnf2h_create_on(device, ped.n_rec, *[_p(x) for x in args[:8]], self.M, _p(cs), len(cs) - 1, _p(dous),
                                        len(dous), 1 if quiet else 0)
        if not self.h:
            raise RuntimeError("cnf2h_create: %s" % self.L.cnf2h_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.cnf2h_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s: %s" % (what, self.L.cnf2h_last_error().decode()))

    def postmarkerdata(self, indcount=None):
        self._chk(self.L.cnf2h_postmarkerdata(self.h, self.n_rec + 1 if indcount is None else indcount), "cnf2h_postmarkerdata")

    def iteration(self, rows_path=None, update=True):
        self._chk(self.L.cnf2h_iteration(self.h, None if rows_path is None else str(rows_path).encode(), 1 if update else 0),
                  "cnf2h_iteration")

    def set_block(self, begin, end):
        self._chk(self.L.cnf2h_set_block(self.h, begin, end), "cnf2h_set_block")

    def balanced_block(self, rank, world):
        b, e = C.c_int32(0), C.c_int32(0)
        self._chk(self.L.cnf2h_balanced_block(self.h, rank, world, C.byref(b), C.byref(e)), "cnf2h_balanced_block")
        return b.value, e.value

    def set_partition(self, rank, world, fn=None):
        """Plans the multi-process run and sets this rank's block (cnf2host.h: cnf2h_set_partition).
        fn(op, buf, count, seg) -> 0 is the transport: op one of X_SUM_SEGMENTS (buf = device address, count doubles),
        X_SUM_HITS (buf = host address of int32[count]), X_GATHER_SEGMENTS (buf = device address, count bytes)."""
        def tramp(_user, op, buf, count, seg):
            try:
                return int(fn(op, buf, count, seg) or 0)
            except Exception:                 # an exception must not unwind through the C frames
                import traceback
                traceback.print_exc()
                return -1
        self._exchange = EXCHANGE_FN(tramp if fn is not None else 0)   # keep the trampoline alive
        self._chk(self.L.cnf2h_set_partition(self.h, rank, world, self._exchange, None), "cnf2h_set_partition")
        return self.partition()

    def partition(self):
        info = np.zeros(10, np.int64)
        self._chk(self.L.cnf2h_get_partition(self.h, _p(info), None), "cnf2h_get_partition")
        owned = np.zeros(int(info[2]), np.int32)
        self._chk(self.L.cnf2h_get_partition(self.h, _p(info), _p(owned) if len(owned) else None), "cnf2h_get_partition")
        return dict(block=(int(info[0]), int(info[1])), owned=owned, n_shared=int(info[3]), segment_records=int(info[4]),
                    bytes_accumulators=int(info[5]), bytes_rows=int(info[6]), bytes_hits=int(info[7]), bytes_payload=int(info[8]),
                    n_private=int(info[9]))

    def reserve(self):
        """The device buffers of the iterations now, not inside the first one (optional)."""
        self._chk(self.L.cnf2h_reserve(self.h), "cnf2h_reserve")

    def timing(self):
        """Wall time of the last iteration by where it went (seconds)."""
        t = np.zeros(5)
        self._chk(self.L.cnf2h_get_timing(self.h, _p(t)), "cnf2h_get_timing")
        return dict(sweep_accumulate_s=float(t[0]), exchange_s=float(t[1]), update_s=float(t[2]), host_s=float(t[3]), total_s=float(t[4]))

    def set_update_flags(self, flags):
        """capi.UPDATE_BOTH_FLOWS (bit-exact fast form), capi.UPDATE_PLAIN (literal kernels), capi.UPDATE_ONE_SCOUT; 0 = default."""
        self._chk(self.L.cnf2h_set_update_flags(self.h, flags), "cnf2h_set_update_flags")

    def set_deterministic(self, on=True):
        self._chk(self.L.cnf2h_set_deterministic(self.h, 1 if on else 0), "cnf2h_set_deterministic")

    def context(self):
        return self.L.cnf2h_context(self.h)

    def passes(self, accumulators=True):
        """hits[C] of the last iteration's update passes and (optionally) haplobase / haplocount [R][M] as left behind."""
        hits = np.zeros(self.n_chrom, np.int32)
        hb = np.zeros((self.n_rec, self.M)) if accumulators else None
        hc = np.zeros((self.n_rec, self.M)) if accumulators else None
        self._chk(self.L.cnf2h_get_passes(self.h, _p(hits), None if hb is None else _p(hb), None if hc is None else _p(hc)),
                  "cnf2h_get_passes")
        return dict(hits=hits, haplobase=hb, haplocount=hc)

    def dump(self, path, limit=1000000):
        assert self.L.cnf2h_dump(self.h, str(path).encode(), limit) == 0

    def deserialize(self, path):
        assert self.L.cnf2h_deserialize(self.h, str(path).encode()) == 0

    def state(self):
        R, M = self.n_rec, self.M
        allele = np.zeros((R, M, 2), np.uint8)
        sure = np.zeros((R, M, 2))
        hw = np.zeros((R, M))
        desc = np.zeros(R, np.int32)
        ch = np.zeros(R, np.int32)
        var = np.zeros((R, M))
        sf = C.c_double(0)
        hits = C.c_int(0)
        rc = self.L.cnf2h_get_state(self.h, _p(allele), _p(sure), _p(hw), _p(desc), _p(ch), _p(var), C.byref(sf),
                                    C.byref(hits))
        assert rc == 0
        return dict(allele=allele, sure=sure, hw=hw, descendants=desc, children=ch, variances=var,
                    scalefactor=sf.value, hits=hits.value)
