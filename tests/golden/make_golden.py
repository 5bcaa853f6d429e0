"""Generate tests/golden/*.npz from the reference's own hot-path code.

Run in the build container only (needs oracle/_ref/libcnf2ref_ieee.so, built by
oracle/ref_extract/build_ref.sh from /root/reference):

    python tests/golden/make_golden.py

Each fixture holds the inputs (pedigree arrays from cnf2freq_amd.synth with fixed
seeds) and the reference's outputs for them -- data only, no reference text:
  G1 emission  e(m,g,flag2,s) for sampled (g,flag2) incl. flag2=-1      (cpp:1380-1385)
  G2 fwbw[s][m][0..2][64], fwbwfactors                                   (cpp:2074-2418)
  G3 factors[8], factor                                                  (cpp:5375-5403)
  G4/G7 dosage rows = sum of val by mapval (full fan-out)                (cpp:5406-5553)
  G5 rawervals[turn][s] for sampled markers                              (cpp:5686-5752)
  G6 fixtrees outputs, founder flags, ignoreflag2 samples                (cpp:3099-3187,3462-3496)
  G7 mapval(g,flag2,s) samples                                           (cpp:5511-5512)
  G8 haplos[n][2] left by HOT LOOP 2 (updatehaplo, HAPLOS mode) at the turn_markers  (cpp:1561-1575,5556)
  G9 infprobs[n][allele index][markerval 1,2] and the homozyg[2] increments of HOT LOOP 2 (GENOSPROBE /
     HOMOZYGOUS / GENOS modes) at the turn_markers                                       (cpp:5513-5577)
  G11 HOT LOOP 2 with its reductions over all analysed individuals in order (moveinfprobs / movehaplos,
     cpp:5876-5902, 3577-3616): per-record infprobs, haplobase, haplocount and per-individual homozyg, for the
     descendant counts stored next to them (acc_desc)
  G12 the reference's own postmarkerdata (cpp:3190-3412 with fixkid / fixparents 1392-1487, lockhaplos 3045-3081) run on
     the inputs as main() does: pm_allele / pm_sure / pm_hw / pm_descendants / pm_children / pm_variances
  G13 (traj_<case>.npz) trajectories: main()'s sequence readers -> postmarkerdata -> 3 x doit<false, genotypereporter>
     (cpp:8083-8136) replayed by ref_driver.inc's ref_iteration around the reference's own update functions
     (cpp:4004-4734; the one stand-in is oracle/ref_extract/boost_gauss_shim.h): after each iteration allele / sure / hw
     of every record, haplobase / haplocount as the last pass left them, hitnnn of every chromosome's pass, scalefactor,
     and the mask `unstable` of elements whose result is rounding noise in the reference itself (oracle/pyiter.py)
  G14 (update_units.npz) the reference's caplogitchange / processinfprobs / updatehaploweights on random inputs
  G10 variances[record][marker] of individ::addvariance with the record's own flag2ignore (NaN where the
     function leaves the entry alone)                                                    (cpp:1489-1558, 3373-3389)
"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_STACKSIZE", "128M")
# postmarkerdata's loops are OpenMP loops over individuals whose results depend on the visiting order (founder flags
# appear as fixtrees reaches each individual, cnF2freq.cpp:3373-3389): one thread = ascending order, reproducible
os.environ["OMP_NUM_THREADS"] = "1"

from cnf2freq_amd import synth  # noqa: E402
from oracle.ref_extract.pyref import RefPed  # noqa: E402

CASES = {
    # name: (constructor, kwargs)
    "f2_implicit_f1": (synth.make_f2, dict(n_ind=4, markers_per_chrom=23, n_chrom=1, seed=12345,
                                           chrom_cm=40.0, missing=0.1)),
    "outbred3_missing": (synth.make_outbred3, dict(n_fam=2, kids_per_fam=2, markers_per_chrom=15,
                                                   seed=777, missing=0.2, random_hw=True,
                                                   random_sure=True)),
    "random_windows": (synth.make_random_windows, dict(n_windows=24, n_markers=6, seed=4242)),
    # an analysed individual that is in the .ped but has no genotype line (empty, all unknown): the window root is
    # itself empty (reltreeordered[0] is set unconditionally, cnF2freq.cpp:3111)
    "f2_ungenotyped": (None, dict()),
}


def make_f2_ungenotyped():
    ped = synth.make_f2(5, 11, 1, seed=321, chrom_cm=25.0, missing=0.1)
    r = int(ped.dous[2])
    ped.empty = ped.empty.copy()
    ped.row_of = ped.row_of.copy()
    ped.empty[r] = 1
    ped.row_of[r] = 0
    ped.founder_flags()
    return ped


def ped_inputs(ped):
    return dict(par=ped.par, gen=ped.gen, empty=ped.empty, row_of=ped.row_of, allele=ped.allele,
                sure=ped.sure, hw=ped.hw, pos=ped.pos, chromstarts=ped.chromstarts, dous=ped.dous)


def generate(name):
    ctor, kw = CASES[name]
    ped = make_f2_ungenotyped() if ctor is None else ctor(**kw)
    # G12 first, on its own load of the pedigree: the reference's postmarkerdata exactly as main() runs it (straight
    # after the readers, no fixtrees before; cnF2freq.cpp:8083-8085) -- genotypes after fixkid / fixparents inference,
    # certainties, haplotype weights after lockhaplos, descendant and children counts, variances
    pm = RefPed(ped, ieee=True, fixtrees_all=False).postmarkerdata()
    R = RefPed(ped, ieee=True)
    rs = np.random.RandomState(99)
    out = {"in_" + k: v for k, v in ped_inputs(ped).items()}
    out["founder"] = R.founder()
    M = ped.n_markers
    n = len(ped.dous)
    fix = np.zeros((n, 2), np.int32)
    rel = np.full((n, 7, 3), -1, np.int32)
    ordered = np.zeros((n, 7), np.int32)
    factors = np.zeros((n, 8))
    factor = np.zeros(n)
    ok = np.zeros(n, np.uint8)
    fwbw = np.zeros((n, 8, M, 3, 64))
    fwbwf = np.zeros((n, 8, M, 3))
    dosage = np.zeros((n, M, 3))
    turn_markers = np.array(sorted(set([0, M // 2, M - 1])), np.int32)
    rawer = np.full((n, len(turn_markers), 128, 8), np.nan)
    haplos = np.zeros((n, len(turn_markers), ped.n_rec, 2))
    infprobs = np.zeros((n, len(turn_markers), ped.n_rec, 2, 2))
    homozyg = np.zeros((n, len(turn_markers), 2))
    em_idx, em_val, mv_idx, mv_val, ig_idx, ig_val = [], [], [], [], [], []
    for j, ind in enumerate(ped.dous):
        ind = int(ind)
        gen = int(ped.gen[ind])
        t = R.fixtrees(ind)
        fix[j] = (t["shiftignore"], t["flag2ignore"])
        rel[j, :len(t["rel"])] = t["rel"]
        ordered[j] = t["ordered"]
        r = R.sweep(ind, gen)
        factors[j], factor[j], ok[j] = r["factors"], r["factor"], r["ok"]
        fwbw[j], fwbwf[j] = r["fwbw"], r["fwbwfactors"]
        shiftend = 8 if gen >= 2 else 2
        if r["ok"]:
            dosage[j] = R.dosage_rows(M)
            for ti, m in enumerate(turn_markers):
                haplos[j, ti] = R.haplos_row(int(m))
                infprobs[j, ti], homozyg[j, ti] = R.infprobs_row(int(m))
                for turn in range(128):
                    if turn & (t["flag2ignore"] >> 1):
                        continue
                    for s in range(shiftend):
                        if s & t["shiftignore"]:
                            continue
                        rawer[j, ti, turn, s] = R.turn_query(int(m), turn, s, -50000 + r["factor"]) - r["factor"]
        for m in range(M):
            for s in range(8):
                for g in rs.choice(64, 3, replace=False):
                    for f2 in [-1] + list(rs.choice(128, 3)):
                        em_idx.append((ind, m, int(g), int(f2), s))
                        em_val.append(R.emission(ind, m, int(g), int(f2), s))
                        if f2 >= 0:
                            mv_idx.append((ind, m, int(g), int(f2), s))
                            mv_val.append(R.mapval(ind, m, int(g), int(f2), s)[0])
                            ig_idx.append((j, m, int(g), int(f2), s))
                            # relmap of the last fixtrees belongs to this individual after sweep()
                            ig_val.append(0)
        # ignoreflag2 uses the thread-private relmap left by the last fixtrees: re-run it
        R.sweep(ind, gen, store=False)
        for k in range(len(ig_idx) - M * 8 * 9, len(ig_idx)):
            _, m, g, f2, s = ig_idx[k]
            ig_val[k] = R.L.ref_ignoreflag2(f2, g, s, m)
    variances = np.full((ped.n_rec, M), np.nan)
    var_f2i = np.zeros(ped.n_rec, np.int32)
    for rec in range(ped.n_rec):
        var_f2i[rec] = R.fixtrees(rec)["flag2ignore"]
        for m in range(M):
            v = R.addvariance(rec, m, int(var_f2i[rec]))
            if v is not None:
                variances[rec, m] = v
    out.update(variances=variances, variances_flag2ignore=var_f2i)
    # descendant counts: the reference's own (individ::descendants after postmarkerdata, cnF2freq.cpp:3224-3255)
    desc = pm["descendants"].astype(np.int32)
    acc = R.accumulate(ped.dous, ped.gen[ped.dous], desc)
    out.update(pm_allele=pm["allele"].astype(np.uint8), pm_sure=pm["sure"], pm_hw=pm["hw"],
               pm_descendants=pm["descendants"], pm_children=pm["children"], pm_variances=pm["variances"])
    out.update(acc_desc=desc, acc_infprobs=acc["infprobs"], acc_haplobase=acc["haplobase"],
               acc_haplocount=acc["haplocount"], acc_homozyg=acc["homozyg"])
    out.update(fixtrees=fix, rel=rel, ordered=ordered, factors=factors, factor=factor, ok=ok,
               fwbw=fwbw, fwbwfactors=fwbwf, dosage=dosage, turn_markers=turn_markers, rawervals=rawer,
               haplos=haplos, infprobs=infprobs, homozyg=homozyg,
               em_idx=np.array(em_idx, np.int32), em_val=np.array(em_val),
               mv_idx=np.array(mv_idx, np.int32), mv_val=np.array(mv_val, np.int32),
               ig_idx=np.array(ig_idx, np.int32), ig_val=np.array(ig_val, np.uint8))
    path = os.path.join(os.path.dirname(__file__), name + ".npz")
    np.savez_compressed(path, **out)
    print(name, "->", path, "%.1f KB" % (os.path.getsize(path) / 1024))


TRAJ_CASES = dict(CASES)
# two chromosomes: the update pass of chromosome c runs again after every later chromosome of the iteration.  Everybody
# is genotyped: an allele nothing is known about sits on an unstable fixed point of the reference's flow (oracle/pyiter.py),
# and with 20 % missing genotypes every family of so small a pedigree holds one
TRAJ_CASES["outbred3_two_chrom"] = (synth.make_outbred3, dict(n_fam=2, kids_per_fam=3, markers_per_chrom=9, n_chrom=2,
                                                             seed=3, missing=0.0, random_sure=True, random_hw=True))
# windows with tie groups (an ancestor in two slots), 3 analysed generations
TRAJ_CASES["ail_ties"] = (synth.make_ail, dict(n_f1=4, n_per_gen=6, n_gen=3, markers_per_chrom=7, n_chrom=1, seed=5,
                                               chrom_cm=20.0, missing=0.05))
# the step-size control (cnF2freq.cpp:6373-6392) over a longer run: 10 families x 4 analysed children, two chromosomes of
# 100 markers, 10 iterations -- the scale factor grows ("good": fewer capped moves than the floor of N / 7) and shrinks
# ("bad": more than in both previous passes) along the way.  Takes ~20 minutes of the reference's own update code.
# Not part of the default run (python make_golden.py regenerates everything else in a few minutes): python make_golden.py traj_long
# (everybody genotyped, as in outbred3_two_chrom: no allele nothing is known about, so no element of the run is rounding noise in
# the reference itself and every record can be compared)
TRAJ_CASES_LONG = {"outbred3_long": (synth.make_outbred3, dict(n_fam=10, kids_per_fam=4, markers_per_chrom=100, n_chrom=2, seed=17,
                                                              missing=0.0))}
TRAJ_ITERATIONS = 3
TRAJ_ITERATIONS_OF = {"outbred3_long": 10}


def traj_ped(name):
    ctor, kw = TRAJ_CASES[name] if name in TRAJ_CASES else TRAJ_CASES_LONG[name]
    return make_f2_ungenotyped() if ctor is None else ctor(**kw)


def generate_trajectory(name):
    ped = traj_ped(name)
    R = RefPed(ped, ieee=True, fixtrees_all=False)
    R.set_priors()                      # the readers' last step (cnF2freq.cpp:6664-6665)
    R.set_dous()
    pm = R.postmarkerdata()             # cnF2freq.cpp:8083-8085
    out = {"in_" + k: v for k, v in ped_inputs(ped).items()}
    out.update(pm_allele=pm["allele"].astype(np.uint8), pm_sure=pm["sure"], pm_hw=pm["hw"],
               pm_descendants=pm["descendants"])
    n_iter = TRAJ_ITERATIONS_OF.get(name, TRAJ_ITERATIONS)
    for k in range(1, n_iter + 1):
        st = R.iteration()
        out["it%d_allele" % k] = st["allele"].astype(np.uint8)
        for key in ("sure", "hw", "haplobase", "haplocount", "hits", "unstable"):
            out["it%d_%s" % (k, key)] = st[key]
        out["it%d_scalefactor" % k] = np.float64(st["scalefactor"])
    path = os.path.join(os.path.dirname(__file__), "traj_" + name + ".npz")
    np.savez_compressed(path, **out)
    print(name, "->", path, "%.1f KB" % (os.path.getsize(path) / 1024), "hits",
          [list(out["it%d_hits" % k]) for k in range(1, n_iter + 1)], "scale factors",
          [float(out["it%d_scalefactor" % k]) for k in range(1, n_iter + 1)])


def generate_update_units(seed=20261004):
    """G14: the reference's own update functions on random inputs, one call per case."""
    import ctypes as C
    rs = np.random.RandomState(seed)
    ped = synth.make_outbred3(1, 1, 12, 2, seed=1)     # any loaded pedigree: record 0 (individual 1) is the scratch individual
    R = RefPed(ped, ieee=True, fixtrees_all=False)
    L = R.L
    M = ped.n_markers
    out = {"chromstarts": np.asarray(ped.chromstarts, np.int32)}
    # caplogitchange
    n = 400
    cap = np.zeros((n, 6))
    for i in range(n):
        eps = 5e-6 / rs.randint(1, 5)
        orig = rs.choice([rs.rand(), eps, 1 - eps, 0.5, rs.rand() * 1e-3, 1 - rs.rand() * 1e-3])
        intended = rs.choice([rs.rand(), eps, 1 - eps, orig, 0.0, 1.0])
        bh = int(rs.rand() < 0.3)
        h = C.c_int(0)
        cap[i] = (intended, orig, eps, bh, L.ref_caplogitchange(intended, orig, eps, C.byref(h), bh), h.value)
    out["cap"] = cap
    # processinfprobs: (inputs) inf[2], present[2], side, curmarker, cursure, has_prior, priorval, priorsure, empty,
    # children, scalefactor -> allele, sure after the call and the hits
    n = 600
    pin = np.zeros((n, 13))
    pout = np.zeros((n, 3))
    for i in range(n):
        side = rs.randint(2)
        present = [int(rs.rand() < 0.8), int(rs.rand() < 0.8)]
        if not any(present):
            present[rs.randint(2)] = 1
        scale = 10.0 ** rs.uniform(-3, 2)
        inf = rs.rand(2) * scale
        if rs.rand() < 0.1:
            inf[rs.randint(2)] = 0.0                       # a key that exists with value 0
        if rs.rand() < 0.15:
            inf[rs.randint(2)] *= 1e-6
        cur = rs.randint(3)
        cursure = rs.choice([rs.rand() * 0.5, 0.02, 0.0, rs.rand(), 1e-7])
        has_prior = int(rs.rand() < 0.8)
        priorval = rs.randint(3)
        priorsure = rs.choice([0.02, rs.rand() * 0.3, 0.0, 1.0, 0.5])
        empty = int(rs.rand() < 0.1)
        children = rs.randint(4)
        sf = rs.choice([0.013, 0.05, 0.19, 1e-3, 0.0])
        m = rs.randint(M)
        other = rs.randint(3)
        pa = [0, 0]
        ps = [0.0, 0.0]
        pa[side], ps[side] = priorval, priorsure
        L.ref_set_stepsize(float(sf), 0, 0)
        L.ref_set_marker(1, m, cur if side == 0 else other, cur if side == 1 else other, cursure if side == 0 else 0.1,
                         cursure if side == 1 else 0.1, 0.5)
        # priors: exist for the whole individual or not at all
        ind_has = has_prior
        L.ref_set_marker(2, m, pa[0], pa[1], ps[0], ps[1], 0.5)
        L.ref_copy_prior(1, 2, m, ind_has)
        L.ref_set_counts(1, children, 1)
        L.ref_set_empty(1, empty)
        pr = np.array(present, np.int32)
        L.ref_set_infprobs(1, m, side, pr.ctypes.data, inf.ctypes.data, None)
        hits = L.ref_processinfprobs(1, m, side)
        buf = np.zeros(5)
        L.ref_get_marker(1, m, buf.ctypes.data)
        pin[i] = (inf[0], inf[1], present[0], present[1], side, cur, cursure, has_prior, priorval, priorsure, empty,
                  children, sf)
        pout[i] = (buf[side], buf[2 + side], hits)
    out["pip_in"], out["pip_out"] = pin, pout
    L.ref_set_empty(1, 0)
    # updatehaploweights: per case the whole individual (two chromosomes)
    n = 60
    uin = np.zeros((n, M, 7))      # hw, haplobase, haplocount, a0, a1, s0, s1
    umeta = np.zeros((n, 3))       # children, descendants, scalefactor
    uout = np.zeros((n, M, 3))     # hw, haplobase, haplocount after
    uhits = np.zeros(n, np.int32)
    for i in range(n):
        children, desc = rs.randint(4), rs.randint(1, 9)
        sf = rs.choice([0.013, 0.05, 0.19])
        hw = np.where(rs.rand(M) < 0.15, 0.5, rs.rand(M))
        hw[rs.rand(M) < 0.1] = rs.choice([0.0, 1.0])
        hw[rs.rand(M) < 0.1] = rs.choice([1e-6, 1 - 1e-6, 5e-6])
        hc = np.where(rs.rand(M) < 0.3, 0.0, rs.rand(M) * desc * 3)
        if i % 5 == 0:
            hc[:ped.chromstarts[1]] = 0                     # a chromosome without any information
        hb = hc * np.clip(rs.rand(M) * 1.2 - 0.1, 0, 1)
        a = rs.randint(0, 3, size=(M, 2))
        s = np.where(rs.rand(M, 2) < 0.5, 0.02, rs.rand(M, 2) * 0.5)
        s[rs.rand(M, 2) < 0.05] = 0.0
        L.ref_set_stepsize(float(sf), 0, 0)
        L.ref_set_counts(1, children, desc)
        for m in range(M):
            L.ref_set_marker(1, m, int(a[m, 0]), int(a[m, 1]), float(s[m, 0]), float(s[m, 1]), float(hw[m]))
            L.ref_set_haplo_accumulators(1, m, float(hb[m]), float(hc[m]))
        uhits[i] = L.ref_updatehaploweights(1)
        buf = np.zeros(5)
        acc = np.zeros(2)
        for m in range(M):
            L.ref_get_marker(1, m, buf.ctypes.data)
            L.ref_get_haplo_accumulators(1, m, acc.ctypes.data)
            uout[i, m] = (buf[4], acc[0], acc[1])
        uin[i] = np.concatenate([hw[:, None], hb[:, None], hc[:, None], a, s], axis=1)
        umeta[i] = (children, desc, sf)
    out.update(uhw_in=uin, uhw_meta=umeta, uhw_out=uout, uhw_hits=uhits)
    path = os.path.join(os.path.dirname(__file__), "update_units.npz")
    np.savez_compressed(path, **out)
    print("update units ->", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def make_flush_impossible():
    """Adversarial but valid data on which adjustprobs' 1e-300 rule (cnF2freq.cpp:1607-1611) decides a result: 152 markers at ONE
    map position at which a nearly phase-locked heterozygous parent makes half of a child's states lose ~1e-2 per marker (they
    end below 1e-300 of their vector and the reference sets them to exactly 0), then one more marker at the same position at
    which every genotype is certain and the parent's phase is locked the OTHER way round: the states that survived are now
    exactly impossible, the states that would fit are the ones the rule removed -- the reference declares the shift modes that
    lock onto that strand impossible (MINFACTOR)."""
    ped = synth.make_outbred3(1, 2, 190, 1, seed=9, missing=0.0, chrom_cm=30.0)
    ped.allele, ped.sure, ped.hw, ped.pos = ped.allele.copy(), ped.sure.copy(), ped.hw.copy(), ped.pos.copy()
    kid = int(ped.dous[0])
    p0, p1 = int(ped.par[kid, 0]), int(ped.par[kid, 1])
    lo, hi = 10, 162
    ped.pos[lo:hi + 1] = ped.pos[lo]
    for r, al in ((kid, (1, 1)), (p0, (1, 2)), (p1, (1, 1))):
        ped.allele[ped.row_of[r], lo:hi] = al
        ped.sure[ped.row_of[r], lo:hi] = 1e-3
        ped.allele[ped.row_of[r], hi] = al
        ped.sure[ped.row_of[r], hi] = 0.0
    ped.hw[ped.row_of[p0], lo:hi] = 0.999
    ped.hw[ped.row_of[p0], hi] = 0.0
    return ped


def generate_flush_impossible():
    """G15: the reference's own likelihoods (oracle/_ref, HOT LOOP 1) on make_flush_impossible()."""
    ped = make_flush_impossible()
    R = RefPed(ped, ieee=True)
    factors, _ = R.sweep_batch(ped.dous, threads=1)
    out = {"in_" + k: v for k, v in ped_inputs(ped).items()}
    out["factors"] = factors
    path = os.path.join(os.path.dirname(__file__), "flush_impossible.npz")
    np.savez_compressed(path, **out)
    print("flush case ->", path, "%.1f KB" % (os.path.getsize(path) / 1024), "impossible modes:", int((factors < -1e14).sum()))


if __name__ == "__main__":
    what = sys.argv[1:] or ["fixtures", "traj", "units", "flush"]
    if "flush" in what:
        generate_flush_impossible()
    if "fixtures" in what:
        for name in CASES:
            generate(name)
    if "traj" in what:
        for name in TRAJ_CASES:
            generate_trajectory(name)
    if "traj_long" in what:                # ~40 minutes of the reference's own update code on one core
        for name in TRAJ_CASES_LONG:
            generate_trajectory(name)
    for name in what:                      # single trajectories by name: python make_golden.py traj:outbred3_two_chrom
        if name.startswith("traj:"):
            generate_trajectory(name[5:])
    if "units" in what:
        generate_update_units()
