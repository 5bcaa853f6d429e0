"""Deterministic synthetic pedigrees for tests and bench.py.

Own generator (counter-based splitmix64, double = (x >> 11) * 2**-53); nothing
here comes from the reference.  Records are laid out in the order
`readalphaped` would number individuals (order of first mention,
/root/reference/cnF2freq.cpp:6480-6540): founders, then for every F2 the
individual followed by its two private empty F1 parents.

Genotype rows are de-duplicated: `row_of[rec]` indexes into the per-marker
tables so that the (many) empty records share the single blank row 0.
"""
from dataclasses import dataclass, field

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed, idx):
    """Counter-based splitmix64: value number `idx` (array ok) of stream `seed`."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) + (np.asarray(idx, dtype=np.uint64) + np.uint64(1))
             * np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform(seed, idx):
    return (splitmix64(seed, idx) >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)


@dataclass
class Pedigree:
    """Host-side mirror of the `individ` graph restricted to what the sweep reads."""
    names: list
    par: np.ndarray          # int32 [R,2], -1 = missing
    gen: np.ndarray          # int32 [R]
    empty: np.ndarray        # uint8 [R]
    row_of: np.ndarray       # int32 [R] -> row in the tables below
    allele: np.ndarray       # uint8 [rows,M,2] values 0,1,2,9
    sure: np.ndarray         # f64   [rows,M,2]
    hw: np.ndarray           # f64   [rows,M]
    pos: np.ndarray          # f64   [M]
    chromstarts: np.ndarray  # int32 [C+1]
    dous: np.ndarray         # int32 [N] analysed records (gen >= 2)
    founder: np.ndarray = field(default=None)  # uint8 [R], filled by founder_flags()
    truth: np.ndarray = field(default=None)    # uint8 [R,M] true allele-2 dosage where the generator knows it (make_outbred3)

    @property
    def n_rec(self):
        return len(self.par)

    @property
    def n_markers(self):
        return len(self.pos)

    def founder_flags(self):
        """individ::founder as fixtrees leaves it for every individual
        (cnF2freq.cpp:3119-3177, applied to all by postmarkerdata 3373-3389):
        true when no parent is non-empty or has a non-empty parent."""
        R = self.n_rec
        informative = np.zeros(R, dtype=bool)
        for lev1 in range(2):
            p = self.par[:, lev1]
            has = p >= 0
            pe = np.where(has, self.empty[np.maximum(p, 0)] == 0, False)
            anypars = np.zeros(R, dtype=bool)
            for lev2 in range(2):
                gp = np.where(has, self.par[np.maximum(p, 0), lev2], -1)
                anypars |= np.where(gp >= 0, self.empty[np.maximum(gp, 0)] == 0, False)
            informative |= has & (pe | anypars)
        self.founder = (~informative).astype(np.uint8)
        return self.founder

    def dense(self):
        """Dense per-record arrays (small cases only): allele int32, sure, hw."""
        return (self.allele[self.row_of].astype(np.int32), self.sure[self.row_of],
                self.hw[self.row_of])


def make_map(n_chrom, markers_per_chrom, chrom_cm=100.0, dummy=True):
    """Top-down cM positions; each chromosome restarts at 0 and gets the trailing
    dummy marker the reference asks for (demo.sh:21-22)."""
    step = chrom_cm / markers_per_chrom
    one = np.arange(markers_per_chrom, dtype=np.float64) * step
    if dummy:
        one = np.concatenate([one, [one[-1] + step]])
    pos = np.tile(one, n_chrom)
    starts = np.arange(n_chrom + 1, dtype=np.int32) * len(one)
    return pos, starts


def _meiosis(seed, stream, n, pos, chromstarts, genrec=-0.02, first=0, fast=False):
    """n gametes (numbers first..first+n-1 of `stream`) over the map: which grand-parental
    strand (0/1) is transmitted per marker.  Haldane crossover, rho = 0.5(1-exp(genrec*delta))
    between adjacent markers.  fast=True draws from numpy's PCG64 instead of the counter-based
    splitmix64 (bench-scale inputs; still deterministic in (seed, stream, first, n))."""
    M = len(pos)
    if fast:
        d = np.diff(pos, prepend=pos[0])
        rho = (0.5 * (1.0 - np.exp(genrec * np.maximum(d, 0.0)))).astype(np.float32)
        rho[np.asarray(chromstarts[:-1])] = 0.5
        out = np.empty((n, M), dtype=np.uint8)
        chunk = max(1, (1 << 25) // max(M, 1))
        for a in range(0, n, chunk):
            b = min(n, a + chunk)
            g = np.random.Generator(np.random.PCG64([seed, stream, first + a]))
            rec = (g.random((b - a, M), dtype=np.float32) < rho[None, :]).astype(np.uint8)
            out[a:b] = np.bitwise_xor.accumulate(rec, axis=1)
        return out
    d = np.diff(pos, prepend=pos[0])
    rho = 0.5 * (1.0 - np.exp(genrec * np.maximum(d, 0.0)))
    rho[np.asarray(chromstarts[:-1])] = 0.5  # free recombination between chromosomes
    out = np.empty((n, M), dtype=np.uint8)
    chunk = max(1, (1 << 24) // max(M, 1))
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        idx = (np.uint64(stream) << np.uint64(40)) + (
            np.arange(first + a, first + b, dtype=np.uint64)[:, None] * np.uint64(M)
            + np.arange(M, dtype=np.uint64)[None, :])
        u = uniform(seed, idx)
        rec = (u < rho[None, :]).astype(np.uint8)
        out[a:b] = np.bitwise_xor.accumulate(rec, axis=1)
    return out


def f2_genotype_rows(i0, i1, pos, starts, seed=12345, sure=0.02, missing=0.0, fast=False):
    """Genotype rows (allele uint8 [n,M,2], sure f64 [n,M,2], hw f64 [n,M]) of F2 individuals
    i0..i1-1: unphased allele-2 dosage of two F1 gametes, stored as the .gen reader does
    (cnF2freq.cpp:6568-6587)."""
    n, M = i1 - i0, len(pos)
    g0 = _meiosis(seed, 1, n, pos, starts, first=i0, fast=fast)
    g1 = _meiosis(seed, 2, n, pos, starts, first=i0, fast=fast)
    dosage = g0 + g1
    allele = np.empty((n, M, 2), np.uint8)
    allele[:, :, 0] = np.where(dosage == 2, 2, 1)
    allele[:, :, 1] = np.where(dosage == 0, 1, 2)
    if missing > 0:
        idx = (np.uint64(3) << np.uint64(40)) + (np.arange(i0, i1, dtype=np.uint64)[:, None] * np.uint64(M)
                                                 + np.arange(M, dtype=np.uint64)[None, :])
        allele[uniform(seed, idx) < missing] = 0
    sr = np.where(allele != 0, sure, 0.0)
    hw = np.full((n, M), 0.5)
    return allele, sr, hw


def f2_pedigree_tables(n_ind):
    """par/gen/empty/row_of/dous of an F2 with private empty F1 parents, in readalphaped's
    numbering (rows: 0 blank, 1 = A, 2 = B, 3+i = F2 i)."""
    R = 2 + 3 * n_ind
    par = np.full((R, 2), -1, np.int32)
    gen = np.zeros(R, np.int32)
    empty = np.ones(R, np.uint8)
    row_of = np.zeros(R, np.int32)
    row_of[0], row_of[1] = 1, 2
    empty[0] = empty[1] = 0
    r = 2 + 3 * np.arange(n_ind, dtype=np.int32)
    par[r, 0], par[r, 1] = r + 1, r + 2
    par[r + 1] = (0, 1)
    par[r + 2] = (0, 1)
    gen[r], gen[r + 1], gen[r + 2] = 2, 1, 1
    empty[r] = 0
    row_of[r] = 3 + np.arange(n_ind, dtype=np.int32)
    return par, gen, empty, row_of, r.copy()


def make_f2(n_ind, markers_per_chrom, n_chrom=1, seed=12345, chrom_cm=100.0, sure=0.02,
            missing=0.0):
    """F2 intercross of two inbred founders A=(1,1), B=(2,2) (configs C1/C2/C4)."""
    pos, starts = make_map(n_chrom, markers_per_chrom, chrom_cm)
    M = len(pos)
    R = 2 + 3 * n_ind
    names = ["A", "B"]
    par = np.full((R, 2), -1, np.int32)
    gen = np.zeros(R, np.int32)
    empty = np.ones(R, np.uint8)
    row_of = np.zeros(R, np.int32)
    n_rows = 3 + n_ind
    allele = np.zeros((n_rows, M, 2), np.uint8)
    sr = np.zeros((n_rows, M, 2), np.float64)
    hw = np.full((n_rows, M), 0.5, np.float64)
    # rows: 0 blank, 1 = A, 2 = B, 3.. = F2s
    allele[1] = 1
    allele[2] = 2
    sr[1:3] = sure
    row_of[0], row_of[1] = 1, 2
    empty[0] = empty[1] = 0
    # each F1 gamete: F1 is (A-strand, B-strand); transmitted allele = 1 + strand choice
    g0 = _meiosis(seed, 1, n_ind, pos, starts)
    g1 = _meiosis(seed, 2, n_ind, pos, starts)
    dosage = (g0 + g1).astype(np.uint8)  # number of '2' alleles
    a0 = np.where(dosage == 2, 2, 1).astype(np.uint8)
    a1 = np.where(dosage == 0, 1, 2).astype(np.uint8)
    if missing > 0:
        idx = (np.uint64(3) << np.uint64(40)) + np.arange(n_ind * M, dtype=np.uint64).reshape(n_ind, M)
        miss = uniform(seed, idx) < missing
        a0[miss] = 0
        a1[miss] = 0
    allele[3:, :, 0] = a0
    allele[3:, :, 1] = a1
    sr[3:] = np.where(allele[3:] != 0, sure, 0.0)
    dous = np.empty(n_ind, np.int32)
    for i in range(n_ind):
        r = 2 + 3 * i
        nm = "F2_%d" % i
        names += [nm, nm + "_aux_realf", nm + "_aux_realm"]
        dous[i] = r
        par[r] = (r + 1, r + 2)
        par[r + 1] = (0, 1)
        par[r + 2] = (0, 1)
        gen[r], gen[r + 1], gen[r + 2] = 2, 1, 1
        empty[r] = 0
        row_of[r] = 3 + i
    ped = Pedigree(names, par, gen, empty, row_of, allele, sr, hw, pos, starts, dous)
    ped.founder_flags()
    return ped


def make_outbred3(n_fam, kids_per_fam, markers_per_chrom, n_chrom=1, seed=777, missing=0.2,
                  sure=0.02, chrom_cm=100.0, random_hw=False, random_sure=False, fast=False):
    """3-generation outbred pedigree (config C5): per family 4 genotyped grandparents,
    2 genotyped parents, `kids_per_fam` analysed children; `missing` of all genotype
    tokens unknown; founder allele-2 frequency U(0.1,0.9) per SNP.  fast=True draws from numpy's
    PCG64 instead of the counter-based splitmix64 (bench-scale inputs: ~8 x quicker; another
    pedigree of the same law, still deterministic in the arguments)."""
    if fast:
        return _make_outbred3_fast(n_fam, kids_per_fam, markers_per_chrom, n_chrom, seed, missing, sure, chrom_cm)
    pos, starts = make_map(n_chrom, markers_per_chrom, chrom_cm)
    M = len(pos)
    per = 6 + kids_per_fam
    R = n_fam * per
    names, par = [], np.full((R, 2), -1, np.int32)
    gen = np.zeros(R, np.int32)
    empty = np.zeros(R, np.uint8)
    row_of = np.arange(1, R + 1, dtype=np.int32)
    hap = np.zeros((R, M, 2), np.uint8)  # true phased alleles (0 = paternal strand)
    freq = 0.1 + 0.8 * uniform(seed, (np.uint64(9) << np.uint64(40)) + np.arange(M, dtype=np.uint64))
    dous = []
    stream = 20

    def gamete(parent_rows, stream_id):
        g = _meiosis(seed, stream_id, len(parent_rows), pos, starts)
        return np.where(g == 0, hap[parent_rows, :, 0], hap[parent_rows, :, 1])

    for f in range(n_fam):
        base = f * per
        for k in range(4):
            r = base + k
            names.append("G%d_%d" % (f, k))
            u = uniform(seed, (np.uint64(10) << np.uint64(40)) + np.uint64(r) * np.uint64(2 * M)
                        + np.arange(2 * M, dtype=np.uint64)).reshape(M, 2)
            hap[r] = 1 + (u < freq[:, None])
        for k in range(2):
            r = base + 4 + k
            names.append("P%d_%d" % (f, k))
            par[r] = (base + 2 * k, base + 2 * k + 1)
            gen[r] = 1
        for k in range(kids_per_fam):
            r = base + 6 + k
            names.append("K%d_%d" % (f, k))
            par[r] = (base + 4, base + 5)
            gen[r] = 2
            dous.append(r)
    prows = np.array([f * per + 4 + k for f in range(n_fam) for k in range(2)])
    hap[prows, :, 0] = gamete(par[prows, 0], stream)
    hap[prows, :, 1] = gamete(par[prows, 1], stream + 1)
    krows = np.array(dous)
    hap[krows, :, 0] = gamete(par[krows, 0], stream + 2)
    hap[krows, :, 1] = gamete(par[krows, 1], stream + 3)

    # unphased genotype as the .gen reader stores it: (1,1),(1,2),(2,2) (cpp:6568-6582)
    d = (hap == 2).sum(axis=2)
    allele = np.zeros((R + 1, M, 2), np.uint8)
    allele[1:, :, 0] = np.where(d == 2, 2, 1)
    allele[1:, :, 1] = np.where(d == 0, 1, 2)
    miss = uniform(seed, (np.uint64(11) << np.uint64(40)) + np.arange(R * M, dtype=np.uint64)).reshape(R, M) < missing
    allele[1:][miss] = 0
    sr = np.zeros((R + 1, M, 2))
    sr[1:] = np.where(allele[1:] != 0, sure, 0.0)
    if random_sure:
        us = uniform(seed, (np.uint64(12) << np.uint64(40)) + np.arange(R * M * 2, dtype=np.uint64)).reshape(R, M, 2)
        sr[1:] = np.where(allele[1:] != 0, 0.001 + 0.1 * us, 0.0)
    hw = np.full((R + 1, M), 0.5)
    if random_hw:
        hw[1:] = 0.05 + 0.9 * uniform(seed, (np.uint64(13) << np.uint64(40)) + np.arange(R * M, dtype=np.uint64)).reshape(R, M)
    ped = Pedigree(names, par, gen, empty, row_of, allele, sr, hw, pos, starts,
                   np.array(dous, np.int32))
    ped.founder_flags()
    ped.truth = d.astype(np.uint8)
    return ped


def _make_outbred3_fast(n_fam, kids_per_fam, markers_per_chrom, n_chrom, seed, missing, sure, chrom_cm):
    """make_outbred3 with vectorised draws (same pedigree layout, same law)."""
    pos, starts = make_map(n_chrom, markers_per_chrom, chrom_cm)
    M = len(pos)
    per = 6 + kids_per_fam
    R = n_fam * per
    g = np.random.Generator(np.random.PCG64([seed, 77]))
    fam = np.arange(R) // per
    k = np.arange(R) % per
    base = fam * per
    par = np.full((R, 2), -1, np.int32)
    isp = (k == 4) | (k == 5)
    par[isp, 0] = base[isp] + 2 * (k[isp] - 4)
    par[isp, 1] = base[isp] + 2 * (k[isp] - 4) + 1
    isk = k >= 6
    par[isk, 0] = base[isk] + 4
    par[isk, 1] = base[isk] + 5
    gen = np.where(isk, 2, np.where(isp, 1, 0)).astype(np.int32)
    names = [("G%d_%d" % (f, j)) if j < 4 else ("P%d_%d" % (f, j - 4)) if j < 6 else ("K%d_%d" % (f, j - 6))
             for f, j in zip(fam.tolist(), k.tolist())]
    freq = (0.1 + 0.8 * g.random(M, dtype=np.float32))
    hap = np.zeros((R, M, 2), np.uint8)
    gp = np.flatnonzero(k < 4)
    chunk = max(1, (1 << 24) // M)
    for a in range(0, len(gp), chunk):
        rows = gp[a:a + chunk]
        hap[rows] = 1 + (g.random((len(rows), M, 2), dtype=np.float32) < freq[None, :, None])
    for rows, stream in ((np.flatnonzero(isp), 20), (np.flatnonzero(isk), 22)):
        for side in range(2):
            gm = _meiosis(seed, stream + side, len(rows), pos, starts, fast=True)
            src = par[rows, side]
            for a in range(0, len(rows), chunk):
                sl = slice(a, a + chunk)
                hap[rows[sl], :, side] = np.where(gm[sl] == 0, hap[src[sl], :, 0], hap[src[sl], :, 1])
    d = (hap == 2).sum(axis=2).astype(np.uint8)
    allele = np.zeros((R + 1, M, 2), np.uint8)
    allele[1:, :, 0] = np.where(d == 2, 2, 1)
    allele[1:, :, 1] = np.where(d == 0, 1, 2)
    for a in range(0, R, chunk):
        miss = g.random((min(chunk, R - a), M), dtype=np.float32) < missing
        allele[1 + a:1 + a + chunk][miss] = 0
    sr = np.zeros((R + 1, M, 2))
    sr[1:] = np.where(allele[1:] != 0, sure, 0.0)
    hw = np.full((R + 1, M), 0.5)
    ped = Pedigree(names, par, gen, np.zeros(R, np.uint8), np.arange(1, R + 1, dtype=np.int32), allele, sr, hw, pos, starts,
                   np.flatnonzero(isk).astype(np.int32))
    ped.founder_flags()
    ped.truth = d
    return ped


def dosage_accuracy(ped, state, mask=None):
    """How well a run's state (cnf2freq_amd.host.Run.state(): allele [R][M][2], sure [R][M][2]) recovers the generator's true
    allele-2 dosage on the genotypes that were withheld (read as missing), or on `mask` [R][M].  Expected dosage of a side =
    P(allele 2) = 1 - sure for a called 2, sure for a called 1, 1/2 for an unknown allele.  Returns dict(n, called = fraction
    with both alleles called, concordance = of those the fraction whose called dosage is the truth, mae = mean absolute
    error of the expected dosage)."""
    a, s = np.asarray(state["allele"]), np.asarray(state["sure"])
    if mask is None:
        mask = (ped.dense()[0] == 0).all(axis=2) & (np.asarray(ped.empty)[:, None] == 0)
    p2 = np.where(a == 2, 1.0 - s, np.where(a == 1, s, 0.5))
    expect = p2.sum(axis=2)
    both = (a != 0).all(axis=2)
    hard = (a == 2).sum(axis=2)
    t = ped.truth.astype(np.float64)
    n = int(mask.sum())
    called = mask & both
    conf = called & (s.max(axis=2) < 0.1)
    return dict(n=n, called=float(called.sum()) / max(n, 1),
                concordance=float((hard[called] == ped.truth[called]).mean()) if called.any() else float("nan"),
                confident=float(conf.sum()) / max(n, 1),
                concordance_confident=float((hard[conf] == ped.truth[conf]).mean()) if conf.any() else float("nan"),
                mae=float(np.abs(expect - t)[mask].mean()) if n else float("nan"))


def make_random_windows(n_windows, n_markers, seed=4242):
    """Small adversarial pedigrees for emission/topology tests: random 3-generation
    windows with missing parents/grandparents, empty members, ancestors shared between
    slots, unknown alleles, the sex-marker sentinel excluded, random sure in [0,0.1]
    (with exact ties), random haplotype weights including the locked values 0 and 1."""
    rs = np.random.RandomState(seed)
    recs, par, gen, empty, names, dous = [], [], [], [], [], []

    def new(g, e, p=(-1, -1)):
        par.append(list(p)); gen.append(g); empty.append(e)
        names.append("r%d" % len(names))
        return len(names) - 1

    for w in range(n_windows):
        kind = rs.randint(0, 8)
        gps = [new(0, int(rs.rand() < 0.25)) for _ in range(4)]
        if kind == 1:
            gps[2], gps[3] = gps[0], gps[1]            # full-sib parents (F2-like sharing)
        if kind == 2:
            gps[2] = gps[1]                            # one shared grandparent
        gp = [g if rs.rand() > 0.2 else -1 for g in gps]
        ps = [new(1, int(rs.rand() < 0.3), (gp[0], gp[1])), new(1, int(rs.rand() < 0.3), (gp[2], gp[3]))]
        if kind == 3:
            ps[1] = -1
        if kind == 4:
            ps[0] = -1
        if kind == 5:
            ps = [-1, -1]
        if kind == 6:
            ps[1] = ps[0]                              # selfed: same parent twice
        kid = new(2, 0, (ps[0], ps[1]))
        dous.append(kid)
    R, M = len(names), n_markers
    allele = np.zeros((R + 1, M, 2), np.uint8)
    sr = np.zeros((R + 1, M, 2))
    hw = np.full((R + 1, M), 0.5)
    a = rs.choice([0, 1, 2], size=(R, M, 2), p=[0.2, 0.4, 0.4]).astype(np.uint8)
    both_missing = rs.rand(R, M) < 0.15
    a[both_missing] = 0
    s = np.round(rs.rand(R, M, 2) * 0.1, 2)            # rounding creates exact sure ties
    s[rs.rand(R, M) < 0.3] = 0.02
    s = np.where(a == 0, np.where(rs.rand(R, M, 2) < 0.5, 0.0, s), s)
    h = rs.rand(R, M)
    h[rs.rand(R, M) < 0.1] = 0.0
    h[rs.rand(R, M) < 0.1] = 1.0
    h[rs.rand(R, M) < 0.3] = 0.5
    em = np.array(empty, bool)
    a[em] = 0
    s[em] = 0.0
    h[em] = 0.5
    allele[1:], sr[1:], hw[1:] = a, s, h
    pos = np.cumsum(rs.choice([0.0, 0.05, 0.5, 3.0], size=M, p=[0.1, 0.3, 0.4, 0.2]))
    ped = Pedigree(names, np.array(par, np.int32), np.array(gen, np.int32),
                   np.array(empty, np.uint8), np.arange(1, R + 1, dtype=np.int32), allele, sr, hw,
                   pos, np.array([0, M], np.int32), np.array(dous, np.int32))
    ped.founder_flags()
    return ped


def make_ail(n_f1, n_per_gen, n_gen, markers_per_chrom, n_chrom=1, seed=2024, chrom_cm=100.0,
             sure=0.02, missing=0.0):
    """Advanced intercross (config C3): 2 inbred founders, `n_f1` genotyped F1 (ped generation 1),
    then `n_gen` random-mating generations of `n_per_gen` analysed individuals (ped generation
    >= 2, explicit parents: readalphaped's direct-parent branch, cnF2freq.cpp:6528-6533).
    Everyone genotyped.  Small populations make full/half-sib matings common, i.e. windows where
    one heterozygous ancestor occupies several slots (the all-or-none rule of ignoreflag2)."""
    pos, starts = make_map(n_chrom, markers_per_chrom, chrom_cm)
    M = len(pos)
    rs = np.random.RandomState(seed)
    names, par, gen = ["A", "B"], [[-1, -1], [-1, -1]], [0, 0]
    hap = [np.ones((M, 2), np.uint8), np.full((M, 2), 2, np.uint8)]
    prev = []
    stream = 100

    def child(p0, p1, g, nm):
        nonlocal stream
        idx = len(names)
        names.append(nm)
        par.append([p0, p1])
        gen.append(g)
        h = np.empty((M, 2), np.uint8)
        for k, p in enumerate((p0, p1)):
            s = _meiosis(seed, stream, 1, pos, starts)[0]
            stream += 1
            h[:, k] = np.where(s == 0, hap[p][:, 0], hap[p][:, 1])
        hap.append(h)
        return idx

    for i in range(n_f1):
        prev.append(child(0, 1, 1, "F1_%d" % i))
    dous = []
    for g in range(n_gen):
        cur = []
        for i in range(n_per_gen):
            p0, p1 = rs.choice(prev, 2, replace=False)
            cur.append(child(int(p0), int(p1), 2 + g, "G%d_%d" % (g + 2, i)))
        dous += cur
        prev = cur
    R = len(names)
    hap = np.stack(hap)
    d = (hap == 2).sum(axis=2)
    allele = np.zeros((R + 1, M, 2), np.uint8)
    allele[1:, :, 0] = np.where(d == 2, 2, 1)
    allele[1:, :, 1] = np.where(d == 0, 1, 2)
    if missing > 0:
        miss = uniform(seed, (np.uint64(14) << np.uint64(40)) + np.arange(R * M, dtype=np.uint64)).reshape(R, M) < missing
        allele[1:][miss] = 0
    sr = np.zeros((R + 1, M, 2))
    sr[1:] = np.where(allele[1:] != 0, sure, 0.0)
    hw = np.full((R + 1, M), 0.5)
    ped = Pedigree(names, np.array(par, np.int32), np.array(gen, np.int32), np.zeros(R, np.uint8),
                   np.arange(1, R + 1, dtype=np.int32), allele, sr, hw, pos, starts, np.array(dous, np.int32))
    ped.founder_flags()
    return ped
