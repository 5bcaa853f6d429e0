"""Haplotyping iterations through the CPU oracle -- TEST INFRASTRUCTURE ONLY (see pyoracle.py).

`OracleRun` strings the restatements together the way doit<false, genotypereporter> does (cnF2freq.cpp:5189-6392
without the toulbar2 bridge): per iteration haplobase / haplocount cleared and the children of every record counted
(cpp:5217-5264); per chromosome HOT LOOP 1 + 2 with the reductions for every analysed individual
(cnf2_oracle.c: cnf2o_accumulate), then processinfprobs for every record, marker and side and updatehaploweights for
every record (cnf2_oracle_iter.c), then the step-size control (cpp:6373-6392).  Pinned to the reference extract's own
replay of the same loop (oracle/ref_extract/ref_driver.inc: ref_iteration) by the trajectory goldens G13
(tests/golden/make_golden.py, tests/test_oracle_iter_golden.py); the update functions themselves are pinned bit for bit
by G14.

Ill-conditioned elements.  Where nothing was learnt about an allele -- no prior, and the evidence g of a value out of
the total h is in proportion to the current belief y in it, g (1 - y) = (h - g) y -- the data term of the gradient
(cpp:4275) is logit(x) and the entropy term (cpp:4280-4283) -logit(x): the gradient vanishes identically, the flow
should stand still, and what the reference's long polynomial leaves is rounding noise whose sign and size decide a move
of up to several 1e-2 (an exact 0 only for "round" sums such as 1/2 : 1/2).  The reference's own result is noise there,
so G13 carries a mask of these elements (it<k>_unstable) and the tests leave the pedigree components in which one of
them differs out of the comparison from then on (tests/conftest.py: TrajectoryChecker).
"""
import ctypes as C

import numpy as np

from . import pyoracle


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def update_pass(O, chromstarts, empty, acc, children, desc, chrom, scalefactor, allele, sure, hw, prior_allele, prior_sure,
                has_prior, entropyfactor=1.0):
    """processinfprobs (cpp:4179-4323) for the markers of `chrom` and updatehaploweights (cpp:4533-4734) for the
    chromosomes that hold any haplocount (<= chrom inside an iteration), record by record, on per-record arrays that are
    modified in place: allele int32 [R][M][2], sure [R][M][2], hw [R][M], acc = dict(infprobs [R][M][2][2],
    haplobase [R][M], haplocount [R][M]).  An infprobs entry counts as present when it is > 0.  Returns hitnnn."""
    cs = np.ascontiguousarray(chromstarts, np.int32)
    hits = C.c_int(0)
    R, M = hw.shape
    relhaplo = np.full(M, 0.5)                                            # cpp:2496
    for r in range(R):
        for m in range(int(cs[chrom]), int(cs[chrom + 1])):
            for side in range(2):
                inf = np.ascontiguousarray(acc["infprobs"][r, m, side])
                present = (inf > 0).astype(np.int32)
                if not present.any():
                    continue
                out = np.zeros(2)
                oa, os_ = C.c_int(0), C.c_double(0)
                if O.cnf2o_processinfprobs(_p(inf), _p(present), side, int(allele[r, m, side]), float(sure[r, m, side]),
                                           int(has_prior[r]), int(prior_allele[r, m, side]), float(prior_sure[r, m, side]),
                                           int(empty[r]), int(children[r]), scalefactor, entropyfactor, C.byref(hits),
                                           _p(out), C.byref(oa), C.byref(os_)):
                    allele[r, m, side] = oa.value
                    sure[r, m, side] = os_.value
            acc["infprobs"][r, m] = 0
        sub = np.ascontiguousarray(cs[:chrom + 2])
        a32 = np.ascontiguousarray(allele[r], np.int32)
        s_r = np.ascontiguousarray(sure[r])
        O.cnf2o_updatehaploweights(chrom + 1, _p(sub), _p(hw[r]), _p(acc["haplobase"][r]), _p(acc["haplocount"][r]), _p(a32),
                                   _p(s_r), _p(relhaplo), int(children[r]), int(desc[r]), scalefactor, entropyfactor,
                                   C.byref(hits))
    return hits.value


class OracleRun:
    """State of a run after the readers and postmarkerdata: per-record rows (allele int [R][M][2], sure [R][M][2],
    hw [R][M]), the rows as read (priors of the records with has_prior), descendants [R]."""

    def __init__(self, ped, allele, sure, hw, prior_allele, prior_sure, has_prior, descendants):
        self.ped = ped
        self.allele = np.array(allele, np.int32)
        self.sure = np.array(sure, np.float64)
        self.hw = np.array(hw, np.float64)
        self.prior_allele = np.array(prior_allele, np.int32)
        self.prior_sure = np.array(prior_sure, np.float64)
        self.has_prior = np.asarray(has_prior, np.uint8)
        self.desc = np.asarray(descendants, np.int32)
        self.scalefactor = 0.013                                          # cpp:3573
        self.old = np.zeros(2, np.int32)                                  # oldhitnnn, oldhitnnn2 (cpp:4004-4005)
        R, M = self.hw.shape
        self.haplobase = np.zeros((R, M))
        self.haplocount = np.zeros((R, M))

    def iteration(self):
        ped = self.ped
        O = pyoracle.lib()
        R, M = self.hw.shape
        cs = [int(x) for x in ped.chromstarts]
        children = np.zeros(R, np.int32)                                  # cpp:5250-5263
        for r in ped.dous:
            for k in range(2):
                if ped.par[r, k] >= 0:
                    children[ped.par[r, k]] += 1
        self.haplobase[:] = 0                                             # cpp:5230-5234
        self.haplocount[:] = 0
        acc = dict(infprobs=np.zeros((R, M, 2, 2)), haplobase=self.haplobase, haplocount=self.haplocount)
        hits = np.zeros(len(cs) - 1, np.int32)
        for c in range(len(cs) - 1):
            o = pyoracle.OraclePed(self.allele, self.sure, self.hw, ped.par, ped.empty, ped.pos)
            part = o.accumulate(ped.dous, ped.gen[ped.dous], self.desc, first=cs[c], last=cs[c + 1] - 1)
            acc["infprobs"][:, cs[c]:cs[c + 1]] += part["infprobs"]
            self.haplobase[:, cs[c]:cs[c + 1]] += part["haplobase"]
            self.haplocount[:, cs[c]:cs[c + 1]] += part["haplocount"]
            hits[c] = update_pass(O, ped.chromstarts, ped.empty, acc, children, self.desc, c, self.scalefactor,
                                  self.allele, self.sure, self.hw, self.prior_allele, self.prior_sure, self.has_prior)
            self.scalefactor = O.cnf2o_scalefactor_step(self.scalefactor, int(hits[c]), _p(self.old), len(ped.dous))
        return dict(allele=self.allele.copy(), sure=self.sure.copy(), hw=self.hw.copy(), haplobase=self.haplobase.copy(),
                    haplocount=self.haplocount.copy(), hits=hits, scalefactor=self.scalefactor)
