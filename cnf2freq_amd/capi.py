"""ctypes binding of libcnf2hip.so (include/cnf2hip.h).

This is the Python host mirror used by tests and bench.py; the reference's own host side
is compiled C++, whose mirror is cnf2freq_amd/csrc/host (readers + CLI).  Everything that
computes goes through the C ABI; there is no CPU fallback: if the shared library is
missing or no HIP device is usable, calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CNF2HIP_LIB: another build of the same library (kernel A/B timing); still the HIP path, never a fallback
LIB_PATH = os.environ.get("CNF2HIP_LIB") or os.path.join(_HERE, "libcnf2hip.so")

OUT_DEVICE, NO_DOSAGE, RAW_DOSAGE, NO_TIES, FULL_SPILL, MERGE_MODES, ACC_DEVICE, ACC_KEEP, XPOSE, LOG_PATHS = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512
ACC_TABLE = 1024
ACC_LANES = 2048
TIES_GENERAL = 4096
UPDATE_PLAIN = 8192
UPDATE_BOTH_FLOWS = 1 << 16       # both allele values' certainty flows (the bit-exact form of the fast update kernels)
UPDATE_ONE_SCOUT = 1 << 17        # the scouts (certainties', weights') in one pass instead of two (A/B)
UPDATE_LITERAL_FINISH = 1 << 19   # the set-aside flows one literal bisection step per round instead of the guided bisection (A/B)
DETERMINISTIC = 16384
TURN_VALU = 32768
FLUSH_TINY = 1 << 20              # cnf2_sweep: general kernel with adjustprobs' 1e-300 rule (the reference's behaviour to the letter)
STATIC_JOBS = 1 << 18             # wave w sweeps jobs w, w + waves, ... instead of taking jobs from the launch's counter (A/B)
MINFACTOR = float(np.float32(-1e15))
IGNORED = -1e30

# every symbol include/cnf2hip.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "cnf2_device_count", "cnf2_ctx_create", "cnf2_ctx_destroy", "cnf2_last_error", "cnf2_version",
    "cnf2_upload_map", "cnf2_upload_rows", "cnf2_update_rows", "cnf2_update_rows_device",
    "cnf2_upload_pedigree",
    "cnf2_window_info", "cnf2_sweep", "cnf2_sync", "cnf2_fwbw_store", "cnf2_locked_query",
    "cnf2_turn_scan", "cnf2_turn_scan_rows", "cnf2_state_posterior", "cnf2_haplos", "cnf2_infprobs", "cnf2_infprobs_rows", "cnf2_descendants", "cnf2_accumulate", "cnf2_sweep_accumulate", "cnf2_sweep_turn_scan", "cnf2_fixparents_scan", "cnf2_variances", "cnf2_variances_exact",
    "cnf2_snapshot_priors", "cnf2_update_pass", "cnf2_download_rows", "cnf2_download_accumulators", "cnf2_upload_accumulators", "cnf2_accumulator_ptrs", "cnf2_update_stats", "cnf2_update_stats_guided", "cnf2_addvariance", "cnf2_emission", "cnf2_emission_paths",
    "cnf2_selftest_lane_xor", "cnf2_last_kernel_ms", "cnf2_last_paths", "cnf2_workspace_bytes", "cnf2_reserve_accumulate", "cnf2_clock_probe", "cnf2_sweep_clock", "cnf2_stream",
    "cnf2_set_grid_reserve", "cnf2_set_batch_jobs", "cnf2_window_table", "cnf2_update_pass_records", "cnf2_exchange_buffer", "cnf2_exchange_download", "cnf2_exchange_upload", "cnf2_exchange_read", "cnf2_exchange_write",
    "cnf2_packed_accumulator_doubles", "cnf2_packed_row_bytes", "cnf2_pack_accumulators", "cnf2_unpack_accumulators",
    "cnf2_pack_rows", "cnf2_unpack_rows",
]


class Cnf2Error(RuntimeError):
    pass


_lib = None


def hip_runtimes():
    """Paths of the HIP runtime libraries mapped into this process.  PyTorch's ROCm wheel carries its own libamdhip64.so
    (SONAME libamdhip64.so.7); libcnf2hip.so asks for libamdhip64.so.7.  If torch is imported FIRST the loader hands its
    copy to this library too (one runtime: device pointers, streams and RCCL buffers are interchangeable); the other way
    round torch loads a second copy next to /opt/rocm's and the two do not know each other's allocations.  Processes
    that use both import torch before the first cnf2freq_amd.capi.load()."""
    paths = set()
    try:
        for line in open("/proc/self/maps"):
            if "libamdhip64" in line:
                paths.add(line.split()[-1])
    except OSError:
        pass
    return sorted(paths)


def require_single_hip_runtime():
    r = hip_runtimes()
    if len(r) > 1:
        raise Cnf2Error("two HIP runtimes are loaded (%s): import torch before cnf2freq_amd loads libcnf2hip.so" % ", ".join(r))


def load():
    """Load libcnf2hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Cnf2Error("libcnf2hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "or `make -C cnf2freq_amd/csrc`")
        L = C.CDLL(LIB_PATH)
        vp, i32 = C.c_void_p, C.c_int
        L.cnf2_device_count.restype = i32
        L.cnf2_ctx_create.argtypes = [i32, C.POINTER(vp)]
        L.cnf2_ctx_destroy.argtypes = [vp]
        L.cnf2_ctx_destroy.restype = None
        L.cnf2_last_error.argtypes = [vp]
        L.cnf2_last_error.restype = C.c_char_p
        L.cnf2_version.restype = C.c_char_p
        L.cnf2_upload_map.argtypes = [vp, vp, i32, vp, i32, vp]
        L.cnf2_upload_rows.argtypes = [vp, i32, vp, vp, vp]
        L.cnf2_update_rows.argtypes = [vp, i32, i32, vp, vp, vp]
        L.cnf2_update_rows_device.argtypes = [vp, i32, i32, vp, vp, vp]
        L.cnf2_upload_pedigree.argtypes = [vp, i32, vp, vp, vp, vp, vp, i32]
        L.cnf2_window_info.argtypes = [vp, i32, vp]
        L.cnf2_sweep.argtypes = [vp, i32, i32, vp, vp, vp, C.c_uint32]
        L.cnf2_sync.argtypes = [vp]
        L.cnf2_fwbw_store.argtypes = [vp, i32, i32, vp, vp]
        L.cnf2_locked_query.argtypes = [vp, i32, i32, i32, vp]
        L.cnf2_turn_scan.argtypes = [vp, i32, i32, i32, vp]
        L.cnf2_turn_scan_rows.argtypes = [vp, i32, i32, vp]
        L.cnf2_state_posterior.argtypes = [vp, i32, i32, vp, C.c_uint32]
        L.cnf2_haplos.argtypes = [vp, i32, i32, vp, C.c_uint32]
        L.cnf2_infprobs.argtypes = [vp, i32, i32, i32, vp, vp, C.c_uint32]
        L.cnf2_infprobs_rows.argtypes = [vp, i32, i32, vp, C.c_uint32]
        L.cnf2_addvariance.argtypes = [vp, i32, i32, vp]
        L.cnf2_descendants.argtypes = [vp, vp]
        L.cnf2_accumulate.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, C.c_uint32]
        L.cnf2_sweep_accumulate.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, C.c_uint32]
        L.cnf2_reserve_accumulate.argtypes = [vp, i32, i32, C.c_uint32]
        L.cnf2_sweep_turn_scan.argtypes = [vp, i32, i32, vp, vp, C.c_uint32]
        L.cnf2_fixparents_scan.argtypes = [vp, vp, i32, vp]
        L.cnf2_variances.argtypes = [vp, vp, i32, i32, vp]
        L.cnf2_variances_exact.argtypes = [vp, vp, vp, i32, i32, vp]
        L.cnf2_snapshot_priors.argtypes = [vp, vp]
        L.cnf2_update_pass.argtypes = [vp, i32, vp, vp, vp, vp, vp, C.c_double, C.c_double, vp, C.c_uint32]
        L.cnf2_download_rows.argtypes = [vp, i32, i32, vp, vp, vp]
        L.cnf2_download_accumulators.argtypes = [vp, vp, vp, vp]
        L.cnf2_upload_accumulators.argtypes = [vp, vp, vp, vp]
        L.cnf2_accumulator_ptrs.argtypes = [vp, vp, vp, vp]
        L.cnf2_update_stats.argtypes = [vp, vp]
        L.cnf2_update_stats_guided.argtypes = [vp, vp]
        L.cnf2_emission.argtypes = [vp, i32, i32, vp]
        L.cnf2_emission_paths.argtypes = [vp, i32, i32, vp]
        L.cnf2_selftest_lane_xor.argtypes = [vp, vp]
        L.cnf2_last_kernel_ms.argtypes = [vp, vp, i32]
        L.cnf2_last_paths.argtypes = [vp, vp, i32]
        L.cnf2_workspace_bytes.argtypes = [vp]
        L.cnf2_workspace_bytes.restype = C.c_size_t
        L.cnf2_clock_probe.argtypes = [vp, vp]
        L.cnf2_sweep_clock.argtypes = [vp, vp]
        L.cnf2_set_grid_reserve.argtypes = [vp, i32]
        L.cnf2_set_batch_jobs.argtypes = [vp, i32]
        L.cnf2_window_table.argtypes = [vp, vp]
        L.cnf2_update_pass_records.argtypes = [vp, i32, vp, i32, vp, vp, C.c_double, C.c_double, vp, C.c_uint32]
        L.cnf2_exchange_buffer.argtypes = [vp, C.c_size_t, vp]
        L.cnf2_exchange_download.argtypes = [vp, vp, C.c_size_t]
        L.cnf2_exchange_upload.argtypes = [vp, vp, C.c_size_t]
        L.cnf2_exchange_read.argtypes = [vp, C.c_size_t, vp, C.c_size_t]
        L.cnf2_exchange_write.argtypes = [vp, C.c_size_t, vp, C.c_size_t]
        L.cnf2_packed_accumulator_doubles.argtypes = [vp]
        L.cnf2_packed_accumulator_doubles.restype = C.c_size_t
        L.cnf2_packed_row_bytes.argtypes = [vp]
        L.cnf2_packed_row_bytes.restype = C.c_size_t
        for f in (L.cnf2_pack_accumulators, L.cnf2_unpack_accumulators, L.cnf2_pack_rows, L.cnf2_unpack_rows):
            f.argtypes = [vp, vp, i32, vp]
        L.cnf2_stream.argtypes = [vp]
        L.cnf2_stream.restype = vp
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """One GPU context (one per process/rank)."""

    def __init__(self, device=0):
        self.L = load()
        h = C.c_void_p()
        rc = self.L.cnf2_ctx_create(device, C.byref(h))
        if rc != 0:
            raise Cnf2Error("cnf2_ctx_create: %s" % self.L.cnf2_last_error(None).decode())
        self.h = h
        self.n_markers = self.n_chrom = self.n_ind = self.n_rec = 0
        self.chromstarts = None

    def close(self):
        if getattr(self, "h", None):
            self.L.cnf2_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise Cnf2Error("%s failed (%d): %s" % (what, rc, self.L.cnf2_last_error(self.h).decode()))

    # -- uploads ---------------------------------------------------------------
    def upload_map(self, pos, chromstarts, genrec=None):
        pos = np.ascontiguousarray(pos, np.float64)
        cs = np.ascontiguousarray(chromstarts, np.int32)
        g = None if genrec is None else np.ascontiguousarray(genrec, np.float64)
        self._chk(self.L.cnf2_upload_map(self.h, _p(pos), len(pos), _p(cs), len(cs) - 1,
                                         None if g is None else _p(g)), "cnf2_upload_map")
        self.n_markers, self.n_chrom, self.chromstarts = len(pos), len(cs) - 1, cs

    def upload_rows(self, allele, sure, hw):
        allele = np.ascontiguousarray(allele, np.uint8)
        sure = np.ascontiguousarray(sure, np.float64)
        hw = np.ascontiguousarray(hw, np.float64)
        assert allele.shape == sure.shape == hw.shape + (2,) and hw.shape[1] == self.n_markers
        self._chk(self.L.cnf2_upload_rows(self.h, hw.shape[0], _p(allele), _p(sure), _p(hw)), "cnf2_upload_rows")

    def alloc_blank_rows(self, n_rows):
        self._chk(self.L.cnf2_upload_rows(self.h, n_rows, None, None, None), "cnf2_upload_rows(blank)")

    def update_rows_device(self, row0, n, d_allele8, d_sure, d_hw):
        """Device pointers (ints): packed allele bytes [n][M], sure [n][M][2], hw [n][M]."""
        self._chk(self.L.cnf2_update_rows_device(self.h, row0, n, C.c_void_p(d_allele8), C.c_void_p(d_sure),
                                                 C.c_void_p(d_hw)), "cnf2_update_rows_device")

    def update_rows(self, row0, allele, sure, hw):
        allele = np.ascontiguousarray(allele, np.uint8)
        sure = np.ascontiguousarray(sure, np.float64)
        hw = np.ascontiguousarray(hw, np.float64)
        self._chk(self.L.cnf2_update_rows(self.h, row0, hw.shape[0], _p(allele), _p(sure), _p(hw)), "cnf2_update_rows")

    def upload_pedigree(self, par, empty, gen, row_of, dous):
        par = np.ascontiguousarray(par, np.int32)
        empty = np.ascontiguousarray(empty, np.uint8)
        gen = np.ascontiguousarray(gen, np.int32)
        row_of = np.ascontiguousarray(row_of, np.int32)
        dous = np.ascontiguousarray(dous, np.int32)
        self._chk(self.L.cnf2_upload_pedigree(self.h, len(par), _p(par), _p(empty), _p(gen), _p(row_of), _p(dous),
                                              len(dous)), "cnf2_upload_pedigree")
        self.n_ind = len(dous)
        self.n_rec = len(par)

    def upload(self, ped, dous=None):
        """Convenience: everything from a cnf2freq_amd.synth.Pedigree."""
        self.upload_map(ped.pos, ped.chromstarts)
        self.upload_rows(ped.allele, ped.sure, ped.hw)
        self.upload_pedigree(ped.par, ped.empty, ped.gen, ped.row_of, ped.dous if dous is None else dous)

    def upload_for_updates(self, ped, has_prior=None):
        """upload() in the form the update passes need: one genotype row per record (row 0 stays the blank row; updates
        write rows in place), and the rows remembered as priors (records that are not `empty` count as genotyped)."""
        a, s, h = ped.dense()
        R = ped.n_rec
        self.upload_map(ped.pos, ped.chromstarts)
        self.upload_rows(np.concatenate([a[:1] * 0, a]).astype(np.uint8), np.concatenate([s[:1] * 0, s]),
                         np.concatenate([h[:1] * 0 + 0.5, h]))
        self.upload_pedigree(ped.par, ped.empty, ped.gen, np.arange(1, R + 1, dtype=np.int32), ped.dous)
        self.snapshot_priors((1 - np.asarray(ped.empty)).astype(np.uint8) if has_prior is None else has_prior)

    def sweep_accumulate_keep(self, desc, ind_begin=0, ind_end=None, deterministic=False):
        """One haplotyping sweep whose accumulators stay in the context (for update_pass(..., acc=None),
        update_pass_records, pack_accumulators, download_accumulators)."""
        ind_end = self.n_ind if ind_end is None else ind_end
        desc = np.ascontiguousarray(desc, np.int32)
        self._chk(self.L.cnf2_sweep_accumulate(self.h, ind_begin, ind_end, _p(desc), None, None, None, None, None, None, None,
                                               DETERMINISTIC if deterministic else 0), "cnf2_sweep_accumulate")

    # -- the sweep -------------------------------------------------------------
    def sweep(self, ind_begin=0, ind_end=None, dosage=True, raw=False, ties=True, full_spill=False,
              merge_modes=False, xpose=False, log_paths=False, ties_general=False, static_jobs=False, flush_tiny=False):
        ind_end = self.n_ind if ind_end is None else ind_end
        n = ind_end - ind_begin
        factors = np.zeros((n, self.n_chrom, 8))
        loglik = np.zeros((n, self.n_chrom))
        dos = np.zeros((n, self.n_markers, 3)) if dosage else None
        flags = ((0 if dosage else NO_DOSAGE) | (RAW_DOSAGE if raw else 0) | (0 if ties else NO_TIES)
                 | (FULL_SPILL if full_spill else 0) | (MERGE_MODES if merge_modes else 0) | (XPOSE if xpose else 0)
                 | (LOG_PATHS if log_paths else 0) | (TIES_GENERAL if ties_general else 0) | (STATIC_JOBS if static_jobs else 0)
                 | (FLUSH_TINY if flush_tiny else 0))
        self._chk(self.L.cnf2_sweep(self.h, ind_begin, ind_end, _p(factors), _p(loglik),
                                    _p(dos) if dosage else None, flags), "cnf2_sweep")
        out = dict(factors=factors, loglik=loglik, dosage=dos)
        if log_paths:
            paths = np.zeros((n, self.n_chrom), np.int32)
            self._chk(self.L.cnf2_last_paths(self.h, _p(paths), paths.size), "cnf2_last_paths")
            out["paths"] = paths
        return out

    def sweep_device(self, ind_begin, ind_end, d_factors, d_loglik, d_dosage, flags=0):
        """Device-pointer form (ints or None); only enqueues on the context's stream."""
        self._chk(self.L.cnf2_sweep(self.h, ind_begin, ind_end, C.c_void_p(d_factors), C.c_void_p(d_loglik),
                                    C.c_void_p(d_dosage) if d_dosage else None, flags | OUT_DEVICE), "cnf2_sweep")

    def sync(self):
        self._chk(self.L.cnf2_sync(self.h), "cnf2_sync")

    def last_kernel_ms(self):
        ms = np.zeros(4, np.float32)
        self._chk(self.L.cnf2_last_kernel_ms(self.h, _p(ms), 4), "cnf2_last_kernel_ms")
        return float(ms[0])

    def set_grid_reserve(self, blocks):
        self._chk(self.L.cnf2_set_grid_reserve(self.h, blocks), "cnf2_set_grid_reserve")

    def set_batch_jobs(self, jobs):
        """Cap on the jobs per batch of sweep_accumulate / sweep_turn_scan (0 = what memory allows)."""
        self._chk(self.L.cnf2_set_batch_jobs(self.h, jobs), "cnf2_set_batch_jobs")

    def workspace_bytes(self):
        return int(self.L.cnf2_workspace_bytes(self.h))

    # -- parity hooks ------------------------------------------------------------
    def window_info(self, ind):
        out = np.zeros(17, np.int32)
        self._chk(self.L.cnf2_window_info(self.h, ind, _p(out)), "cnf2_window_info")
        return dict(shiftignore=int(out[0]), flag2ignore=int(out[1]), founder=int(out[2]),
                    slots=out[3:10].copy(), tie=out[10:17].copy())

    def fwbw_store(self, ind, chrom=0):
        mc = int(self.chromstarts[chrom + 1] - self.chromstarts[chrom])
        fw = np.zeros((8, mc, 3, 64))
        ff = np.zeros((8, mc, 3))
        self._chk(self.L.cnf2_fwbw_store(self.h, ind, chrom, _p(fw), _p(ff)), "cnf2_fwbw_store")
        return fw, ff

    def locked_query(self, ind, chrom, marker):
        v = np.zeros((8, 64, 128))
        self._chk(self.L.cnf2_locked_query(self.h, ind, chrom, marker, _p(v)), "cnf2_locked_query")
        return v

    def turn_scan(self, ind, chrom, marker):
        v = np.zeros((128, 8))
        self._chk(self.L.cnf2_turn_scan(self.h, ind, chrom, marker, _p(v)), "cnf2_turn_scan")
        return v

    def state_posterior(self, ind, chrom=0, ties=True):
        mc = int(self.chromstarts[chrom + 1] - self.chromstarts[chrom])
        v = np.zeros((mc, 64))
        self._chk(self.L.cnf2_state_posterior(self.h, ind, chrom, _p(v), 0 if ties else NO_TIES), "cnf2_state_posterior")
        return v

    def turn_scan_rows(self, ind, chrom=0):
        mc = int(self.chromstarts[chrom + 1] - self.chromstarts[chrom])
        v = np.zeros((mc, 128, 8))
        self._chk(self.L.cnf2_turn_scan_rows(self.h, ind, chrom, _p(v)), "cnf2_turn_scan_rows")
        return v

    def haplos(self, ind, chrom=0, ties=True):
        mc = int(self.chromstarts[chrom + 1] - self.chromstarts[chrom])
        v = np.zeros((mc, 7, 2))
        self._chk(self.L.cnf2_haplos(self.h, ind, chrom, _p(v), 0 if ties else NO_TIES), "cnf2_haplos")
        return v

    def infprobs(self, ind, marker, chrom=0, ties=True):
        inf = np.zeros((7, 2, 2))
        hz = np.zeros(2)
        self._chk(self.L.cnf2_infprobs(self.h, ind, chrom, marker, _p(inf), _p(hz), 0 if ties else NO_TIES),
                  "cnf2_infprobs")
        return inf, hz

    def infprobs_rows(self, ind, chrom=0, ties=True):
        """Closed-form rows for every marker of the chromosome: (inf[mc][7][2][2], hz[mc][2])."""
        mc = int(self.chromstarts[chrom + 1] - self.chromstarts[chrom])
        v = np.zeros((mc, 30))
        self._chk(self.L.cnf2_infprobs_rows(self.h, ind, chrom, _p(v), 0 if ties else NO_TIES), "cnf2_infprobs_rows")
        return v[:, :28].reshape(mc, 7, 2, 2).copy(), v[:, 28:].copy()

    def descendants(self):
        d = np.zeros(self.n_rec, np.int32)
        self._chk(self.L.cnf2_descendants(self.h, _p(d)), "cnf2_descendants")
        return d

    def accumulate(self, desc, ind_begin=0, ind_end=None, ties=True):
        ind_end = self.n_ind if ind_end is None else ind_end
        desc = np.ascontiguousarray(desc, np.int32)
        inf = np.zeros((self.n_rec, self.n_markers, 2, 2))
        hb = np.zeros((self.n_rec, self.n_markers))
        hc = np.zeros((self.n_rec, self.n_markers))
        hz = np.zeros((ind_end - ind_begin, self.n_markers, 2))
        self._chk(self.L.cnf2_accumulate(self.h, ind_begin, ind_end, _p(desc), _p(inf), _p(hb), _p(hc), _p(hz),
                                         0 if ties else NO_TIES), "cnf2_accumulate")
        return dict(infprobs=inf, haplobase=hb, haplocount=hc, homozyg=hz)

    def sweep_accumulate(self, desc, ind_begin=0, ind_end=None, ties=True, raw=False, table_form=False, lane_form=False,
                         ties_general=False, deterministic=False, static_jobs=False, rows=True):
        """One haplotyping sweep: the outputs of sweep() and the per-record accumulators, batched on the device.
        rows=False: no dosage pointer is passed (what an iteration that prints no rows does): "dosage" comes back as zeros."""
        ind_end = self.n_ind if ind_end is None else ind_end
        n = ind_end - ind_begin
        desc = np.ascontiguousarray(desc, np.int32)
        factors = np.zeros((n, self.n_chrom, 8))
        loglik = np.zeros((n, self.n_chrom))
        dos = np.zeros((n, self.n_markers, 3))
        inf = np.zeros((self.n_rec, self.n_markers, 2, 2))
        hb = np.zeros((self.n_rec, self.n_markers))
        hc = np.zeros((self.n_rec, self.n_markers))
        hz = np.zeros((n, self.n_markers, 2))
        self._chk(self.L.cnf2_sweep_accumulate(self.h, ind_begin, ind_end, _p(desc), _p(factors), _p(loglik), _p(dos) if rows else None,
                                               _p(inf), _p(hb), _p(hc), _p(hz),
                                               (0 if ties else NO_TIES) | (RAW_DOSAGE if raw else 0)
                                               | (ACC_TABLE if table_form else 0) | (ACC_LANES if lane_form else 0)
                                               | (TIES_GENERAL if ties_general else 0)
                                               | (DETERMINISTIC if deterministic else 0) | (STATIC_JOBS if static_jobs else 0)),
                  "cnf2_sweep_accumulate")
        return dict(factors=factors, loglik=loglik, dosage=dos, infprobs=inf, haplobase=hb, haplocount=hc, homozyg=hz)

    def sweep_accumulate_device(self, desc, ind_begin, ind_end, d_factors, d_loglik, d_dosage, d_inf, d_hb, d_hc, d_hz,
                                flags=0):
        """Device-pointer form (ints): outputs and accumulators stay on the GPU; only enqueues (use sync())."""
        desc = np.ascontiguousarray(desc, np.int32)
        self._chk(self.L.cnf2_sweep_accumulate(self.h, ind_begin, ind_end, _p(desc), C.c_void_p(d_factors),
                                               C.c_void_p(d_loglik), C.c_void_p(d_dosage), C.c_void_p(d_inf),
                                               C.c_void_p(d_hb), C.c_void_p(d_hc), C.c_void_p(d_hz),
                                               flags | OUT_DEVICE | ACC_DEVICE), "cnf2_sweep_accumulate")

    def sweep_turn_scan(self, ind_begin=0, ind_end=None, full=True, lse=True, ties=True, ties_general=False, valu=False,
                        static_jobs=False):
        """Batched turn scan: rawervals [n][M][128][8] and / or their log-sum-exp over the admissible modes [n][M][128]."""
        ind_end = self.n_ind if ind_end is None else ind_end
        n = ind_end - ind_begin
        raw = np.zeros((n, self.n_markers, 128, 8)) if full else None
        ls = np.zeros((n, self.n_markers, 128)) if lse else None
        self._chk(self.L.cnf2_sweep_turn_scan(self.h, ind_begin, ind_end, _p(raw) if full else None, _p(ls) if lse else None,
                                              (0 if ties else NO_TIES) | (TIES_GENERAL if ties_general else 0)
                                              | (TURN_VALU if valu else 0) | (STATIC_JOBS if static_jobs else 0)),
                  "cnf2_sweep_turn_scan")
        return raw, ls

    def fixparents_scan(self, recs):
        recs = np.ascontiguousarray(recs, np.int32)
        ok = np.zeros((len(recs), self.n_markers, 2), np.uint8)
        self._chk(self.L.cnf2_fixparents_scan(self.h, _p(recs), len(recs), _p(ok)), "cnf2_fixparents_scan")
        return ok

    def variances(self, recs, ordered=True, brute=False):
        recs = np.ascontiguousarray(recs, np.int32)
        v = np.zeros((len(recs), self.n_markers))
        self._chk(self.L.cnf2_variances(self.h, _p(recs), len(recs), (1 if ordered else 0) | (2 if brute else 0), _p(v)),
                  "cnf2_variances")
        return v

    def variances_exact(self, recs, markers, ordered=True):
        """addvariance of (recs[q], markers[q]) with the reference's own rounding (cnf2_variances_exact)"""
        recs, markers = np.ascontiguousarray(recs, np.int32), np.ascontiguousarray(markers, np.int32)
        assert recs.shape == markers.shape
        v = np.zeros(len(recs))
        self._chk(self.L.cnf2_variances_exact(self.h, _p(recs), _p(markers), len(recs), 1 if ordered else 0, _p(v)), "cnf2_variances_exact")
        return v

    def snapshot_priors(self, has_prior):
        hp = np.ascontiguousarray(has_prior, np.uint8)
        assert len(hp) == self.n_rec
        self._chk(self.L.cnf2_snapshot_priors(self.h, _p(hp)), "cnf2_snapshot_priors")

    def update_pass(self, chrom, children, descendants, scalefactor, entropyfactor=1.0, acc=None, flags=0):
        """acc: dict with host arrays infprobs / haplobase / haplocount (updated in place), or None for the
        accumulators the last sweep_accumulate(..., keep=True) left in the context.  Returns hitnnn."""
        ch = np.ascontiguousarray(children, np.int32)
        de = np.ascontiguousarray(descendants, np.int32)
        hits = np.zeros(1, np.int32)
        a = (None, None, None) if acc is None else (_p(acc["infprobs"]), _p(acc["haplobase"]), _p(acc["haplocount"]))
        self._chk(self.L.cnf2_update_pass(self.h, chrom, _p(ch), _p(de), a[0], a[1], a[2], scalefactor, entropyfactor,
                                          _p(hits), flags), "cnf2_update_pass")
        return int(hits[0])

    def update_pass_records(self, chrom, recs, children, descendants, scalefactor, entropyfactor=1.0, flags=0):
        """update_pass restricted to the listed records (ascending), on the accumulators the context holds."""
        rc_ = np.ascontiguousarray(recs, np.int32)
        ch = np.ascontiguousarray(children, np.int32)
        de = np.ascontiguousarray(descendants, np.int32)
        hits = np.zeros(1, np.int32)
        self._chk(self.L.cnf2_update_pass_records(self.h, chrom, _p(rc_), len(rc_), _p(ch), _p(de), scalefactor, entropyfactor,
                                                  _p(hits), flags), "cnf2_update_pass_records")
        return int(hits[0])

    def window_table(self):
        """window_info() of every analysed individual in one call: int32 [n_ind][17]."""
        out = np.zeros((self.n_ind, 17), np.int32)
        self._chk(self.L.cnf2_window_table(self.h, _p(out)), "cnf2_window_table")
        return out

    def exchange_buffer(self, nbytes):
        p = C.c_void_p(0)
        self._chk(self.L.cnf2_exchange_buffer(self.h, nbytes, C.byref(p)), "cnf2_exchange_buffer")
        return p.value

    def pack_accumulators(self, recs, d_packed):
        r = np.ascontiguousarray(recs, np.int32)
        self._chk(self.L.cnf2_pack_accumulators(self.h, _p(r), len(r), C.c_void_p(d_packed)), "cnf2_pack_accumulators")

    def unpack_accumulators(self, recs, d_packed):
        r = np.ascontiguousarray(recs, np.int32)
        self._chk(self.L.cnf2_unpack_accumulators(self.h, _p(r), len(r), C.c_void_p(d_packed)), "cnf2_unpack_accumulators")

    def pack_rows(self, recs, d_packed):
        r = np.ascontiguousarray(recs, np.int32)
        self._chk(self.L.cnf2_pack_rows(self.h, _p(r), len(r), C.c_void_p(d_packed)), "cnf2_pack_rows")

    def unpack_rows(self, recs, d_packed):
        r = np.ascontiguousarray(recs, np.int32)
        self._chk(self.L.cnf2_unpack_rows(self.h, _p(r), len(r), C.c_void_p(d_packed)), "cnf2_unpack_rows")

    def clock_probe(self):
        """Shader clock under a double-precision vector load, MHz."""
        v = C.c_double(0)
        self._chk(self.L.cnf2_clock_probe(self.h, C.byref(v)), "cnf2_clock_probe")
        return v.value

    def sweep_clock(self):
        """Shader clock of the last sweep's untied fast-kernel launch (its own counters), MHz; 0 if none ran."""
        v = C.c_double(0)
        self._chk(self.L.cnf2_sweep_clock(self.h, C.byref(v)), "cnf2_sweep_clock")
        return v.value

    def update_stats(self):
        """Diagnostics of the last update pass: dict of (flows, steps, slots, refills) for certainty / haploweight."""
        out = np.zeros(16, np.uint64)
        self._chk(self.L.cnf2_update_stats(self.h, _p(out)), "cnf2_update_stats")
        names = ("flows", "scout_evaluations", "ended_in_scout", "pinned", "finish_steps", "finish_slots", "quadratures", "ended_by_tolerance")
        return dict(certainty=dict(zip(names, (int(x) for x in np.r_[out[0:4], out[8:12]]))),
                    haploweight=dict(zip(names, (int(x) for x in np.r_[out[4:8], out[12:16]]))))

    def download_accumulators(self):
        """The accumulators the context holds: dict(infprobs [R][M][2][2], haplobase [R][M], haplocount [R][M])."""
        R, M = self.n_rec, self.n_markers
        inf, hb, hc = np.zeros((R, M, 2, 2)), np.zeros((R, M)), np.zeros((R, M))
        self._chk(self.L.cnf2_download_accumulators(self.h, _p(inf), _p(hb), _p(hc)), "cnf2_download_accumulators")
        return dict(infprobs=inf, haplobase=hb, haplocount=hc)

    @staticmethod
    def accumulators_of(ctx_handle, n_rec, n_markers):
        """download_accumulators() for a raw cnf2_ctx handle (e.g. cnf2h_context of a host run)."""
        L = load()
        inf, hb, hc = np.zeros((n_rec, n_markers, 2, 2)), np.zeros((n_rec, n_markers)), np.zeros((n_rec, n_markers))
        if L.cnf2_download_accumulators(ctx_handle, _p(inf), _p(hb), _p(hc)) != 0:
            raise Cnf2Error("cnf2_download_accumulators: " + L.cnf2_last_error(ctx_handle).decode())
        return dict(infprobs=inf, haplobase=hb, haplocount=hc)

    def upload_accumulators(self, acc):
        a = [np.ascontiguousarray(acc[k], np.float64) for k in ("infprobs", "haplobase", "haplocount")]
        self._chk(self.L.cnf2_upload_accumulators(self.h, _p(a[0]), _p(a[1]), _p(a[2])), "cnf2_upload_accumulators")

    def download_rows(self, row0, n):
        allele = np.zeros((n, self.n_markers, 2), np.uint8)
        sure = np.zeros((n, self.n_markers, 2))
        hw = np.zeros((n, self.n_markers))
        self._chk(self.L.cnf2_download_rows(self.h, row0, n, _p(allele), _p(sure), _p(hw)), "cnf2_download_rows")
        return allele, sure, hw

    def addvariance(self, ind, chrom=0):
        mc = int(self.chromstarts[chrom + 1] - self.chromstarts[chrom])
        v = np.zeros(mc)
        self._chk(self.L.cnf2_addvariance(self.h, ind, chrom, _p(v)), "cnf2_addvariance")
        return v

    def emission(self, ind, marker):
        e = np.zeros((8, 64))
        self._chk(self.L.cnf2_emission(self.h, ind, marker, _p(e)), "cnf2_emission")
        return e

    def emission_paths(self, ind, marker):
        e = np.zeros((8, 64, 128))
        self._chk(self.L.cnf2_emission_paths(self.h, ind, marker, _p(e)), "cnf2_emission_paths")
        return e

    def selftest_lane_xor(self):
        out = np.zeros((6, 64))
        self._chk(self.L.cnf2_selftest_lane_xor(self.h, _p(out)), "cnf2_selftest_lane_xor")
        return out
