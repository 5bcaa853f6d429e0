"""GPU suite (-m gpu): the HIP path through the C ABI against (a) the golden vectors the
reference's own code produced and (b) the oracle on seeded synthetic pedigrees.
Tolerances: north_star asks for 1e-6 relative on posteriors; the kernels agree to ~1e-12,
the asserts use 1e-9 so that a real regression cannot hide."""
import numpy as np
import pytest

from cnf2freq_amd import synth
from conftest import GOLDEN_CASES, load_golden, oracle_accumulate_threaded, oracle_ped

pytestmark = pytest.mark.gpu

RTOL = 1e-9
ATOL = 1e-12


@pytest.fixture(scope="module")
def capi():
    import __graft_entry__ as g
    g.build()
    from cnf2freq_amd import capi as c
    assert c.load().cnf2_device_count() >= 1, "no HIP device: the product path has no fallback"
    return c


def norm_rows(d):
    s = d.sum(axis=-1, keepdims=True)
    return d / np.where(s > 0, s, 1.0)


def test_lane_exchange_primitives(capi):
    ctx = capi.Context(0)
    out = ctx.selftest_lane_xor()
    lanes = np.arange(64)
    for k in range(6):
        assert np.array_equal(out[k], 1000.0 + (lanes ^ (1 << k))), "xor distance %d" % (1 << k)
    ctx.close()


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_window_topology_matches_reference(capi, case):
    ped, z = load_golden(case)
    ctx = capi.Context(0)
    ctx.upload(ped)
    for j in range(len(ped.dous)):
        w = ctx.window_info(j)
        assert (w["shiftignore"], w["flag2ignore"]) == tuple(z["fixtrees"][j])
        assert w["founder"] == z["founder"][ped.dous[j]]
        # reltreeordered (cnF2freq.cpp:3111-3152): the individual itself, then the NON-EMPTY ancestors by slot;
        # the library's slot table also lists empty ancestors (they are members of reltree)
        for k, r in enumerate(z["ordered"][j]):
            if r >= 0:
                assert w["slots"][k] == r
            else:
                assert k != 0 and (w["slots"][k] < 0 or ped.empty[w["slots"][k]])
    ctx.close()


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_emission_matches_reference(capi, case):
    ped, z = load_golden(case)
    ctx = capi.Context(0)
    ctx.upload(ped)
    index = {int(r): j for j, r in enumerate(ped.dous)}
    cache, paths = {}, {}
    n_paths = 0
    for (ind, m, g, f2, s), want in zip(z["em_idx"], z["em_val"]):
        key = (int(ind), int(m))
        if f2 != -1:
            # path-resolved emission (flag2 >= 0) through the stage-2 line terms
            if key not in paths:
                paths[key] = ctx.emission_paths(index[int(ind)], int(m))
            np.testing.assert_allclose(paths[key][s, g, f2], want, rtol=1e-12, atol=1e-300)
            n_paths += 1
            continue
        if key not in cache:
            cache[key] = ctx.emission(index[int(ind)], int(m))
        np.testing.assert_allclose(cache[key][s, g], want, rtol=1e-13, atol=1e-300)
    assert n_paths > 0
    ctx.close()


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_alpha_beta_store_matches_reference(capi, case):
    ped, z = load_golden(case)
    ctx = capi.Context(0)
    ctx.upload(ped)
    for j in range(len(ped.dous)):
        fw, ff = ctx.fwbw_store(j, 0)
        live = z["factors"][j] > -1e29
        np.testing.assert_allclose(fw[live], z["fwbw"][j][live], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(ff[live], z["fwbwfactors"][j][live], rtol=RTOL, atol=1e-9)
    ctx.close()


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_sweep_matches_reference(capi, case):
    ped, z = load_golden(case)
    ctx = capi.Context(0)
    ctx.upload(ped)
    got = ctx.sweep()
    raw = ctx.sweep(raw=True)
    for j in range(len(ped.dous)):
        live = z["factors"][j] > -1e29
        np.testing.assert_allclose(got["factors"][j, 0][live], z["factors"][j][live], rtol=RTOL, atol=1e-9)
        assert np.all(got["factors"][j, 0][~live] == -1e30)
        np.testing.assert_allclose(got["loglik"][j, 0], z["factor"][j], rtol=RTOL, atol=1e-9)
        if z["ok"][j]:
            np.testing.assert_allclose(raw["dosage"][j], z["dosage"][j], rtol=1e-8, atol=1e-12)
            np.testing.assert_allclose(got["dosage"][j], norm_rows(z["dosage"][j]), rtol=1e-8, atol=1e-12)
    ctx.close()


def test_f2_config1_against_oracle(capi):
    """BASELINE config 1 shape (F2, implicit empty F1s) at a size the oracle finishes quickly."""
    ped = synth.make_f2(48, 125, 2, seed=12345, chrom_cm=100.0, missing=0.05)
    ctx = capi.Context(0)
    ctx.upload(ped)
    got = ctx.sweep()
    o = oracle_ped(ped)
    for c in range(2):
        first, last = int(ped.chromstarts[c]), int(ped.chromstarts[c + 1]) - 1
        want = o.sweep_batch(ped.dous, ped.gen[ped.dous], first=first, last=last, mode=2)
        np.testing.assert_allclose(got["factors"][:, c], want["factors"], rtol=RTOL, atol=1e-8)
        np.testing.assert_allclose(got["loglik"][:, c], want["factor"], rtol=RTOL, atol=1e-8)
        np.testing.assert_allclose(got["dosage"][:, first:last + 1], want["dosage"], rtol=1e-7, atol=1e-11)
    ctx.close()


def test_outbred_missing_against_oracle(capi):
    """BASELINE config 5 shape: 3-generation outbred, 20% missing, random weights."""
    ped = synth.make_outbred3(6, 4, 60, 2, seed=31, missing=0.2, random_hw=True, random_sure=True)
    ctx = capi.Context(0)
    ctx.upload(ped)
    o = oracle_ped(ped)
    # tied windows through the tile-producer kernel (a backward pass per tie combination: the default) and through the
    # general kernel (per-marker producer, loop over the combinations inside a marker)
    for general in (False, True):
        got = ctx.sweep(ties_general=general)
        for c in range(2):
            first, last = int(ped.chromstarts[c]), int(ped.chromstarts[c + 1]) - 1
            want = o.sweep_batch(ped.dous, ped.gen[ped.dous], first=first, last=last, mode=2)
            np.testing.assert_allclose(got["factors"][:, c], want["factors"], rtol=RTOL, atol=1e-8)
            np.testing.assert_allclose(got["dosage"][:, first:last + 1], want["dosage"], rtol=1e-7, atol=1e-11)
    ctx.close()


def test_tied_and_degenerate_windows_against_oracle(capi):
    """Adversarial windows: missing parents, founders, one ancestor in several slots
    (the all-or-none rule of ignoreflag2), locked haplotype weights."""
    ped = synth.make_random_windows(64, 12, seed=77)
    ctx = capi.Context(0)
    ctx.upload(ped)
    got = ctx.sweep(raw=True)
    noties = ctx.sweep(raw=True, ties=False)
    o = oracle_ped(ped)
    for j, ind in enumerate(ped.dous):
        r2 = o.sweep_ind(int(ind), int(ped.gen[ind]), mode=2)
        r3 = o.sweep_ind(int(ind), int(ped.gen[ind]), mode=3)
        live = r2["factors"] > -1e29
        np.testing.assert_allclose(got["factors"][j, 0][live], r2["factors"][live], rtol=RTOL, atol=1e-8)
        if r2["ok"]:
            np.testing.assert_allclose(got["dosage"][j], r2["dosage"], rtol=1e-7, atol=1e-12)
            np.testing.assert_allclose(noties["dosage"][j], r3["dosage"], rtol=1e-7, atol=1e-12)
    ctx.close()


def test_range_and_ragged_inputs(capi):
    """Sub-ranges of individuals, a one-marker chromosome, and an empty range."""
    ped = synth.make_f2(9, 17, 1, seed=3, chrom_cm=20.0)
    # make the map ragged: chromosomes of 1, 5 and 12 markers
    ped.chromstarts = np.array([0, 1, 6, 18], np.int32)
    ped.pos = np.concatenate([[0.0], np.arange(5) * 0.7, np.arange(12) * 1.3])
    ctx = capi.Context(0)
    ctx.upload(ped)
    full = ctx.sweep()
    part = ctx.sweep(3, 7)
    assert np.array_equal(part["dosage"], full["dosage"][3:7])
    assert np.array_equal(part["factors"], full["factors"][3:7])
    empty = ctx.sweep(4, 4)
    assert empty["factors"].shape[0] == 0
    o = oracle_ped(ped)
    for c in range(3):
        first, last = int(ped.chromstarts[c]), int(ped.chromstarts[c + 1]) - 1
        want = o.sweep_batch(ped.dous, ped.gen[ped.dous], first=first, last=last, mode=2)
        np.testing.assert_allclose(full["factors"][:, c], want["factors"], rtol=RTOL, atol=1e-8)
        np.testing.assert_allclose(full["dosage"][:, first:last + 1], want["dosage"], rtol=1e-7, atol=1e-11)
    ctx.close()


def test_drop_in_command_line_matches_oracle(capi, tmp_path):
    """The `cnF2freq` executable on PlantImpute-format files.  --output receives the dump of every iteration
    (iteration 0 only dumps, cnF2freq.cpp:8128-8136, 8166-8186) and the rows of the last one; rows are the raw class
    sums the reporter accumulates (cnF2freq.cpp:3523), here against the oracle to the 5 printed decimals."""
    import os
    import subprocess
    from conftest import ROOT
    ped = synth.make_f2(5, 30, 2, seed=99, chrom_cm=50.0, missing=0.1)
    names = ped.names
    with open(tmp_path / "x.map", "w") as f:
        f.write("\n".join("%.10g" % p for p in ped.pos) + "\n")
    with open(tmp_path / "x.ped", "w") as f:
        f.write("A 0 0\nB 0 0\n")
        for r in ped.dous:
            f.write("%s A B 2\n" % names[r])
    tok = {(1, 1): "0", (1, 2): "1", (2, 2): "2", (0, 0): "9"}
    with open(tmp_path / "x.gen", "w") as f:
        for r in [0, 1] + list(ped.dous):
            a = ped.allele[ped.row_of[r]]
            f.write(names[r] + " " + " ".join(tok[(int(x[0]), int(x[1]))] for x in a) + "\n")
    exe = os.path.join(ROOT, "cnf2freq_amd", "cnF2freq")
    subprocess.run([exe, "--mapfile", str(tmp_path / "x.map"), "--pedfile", str(tmp_path / "x.ped"), "--genfile",
                    str(tmp_path / "x.gen"), "--output", str(tmp_path / "out.txt"), "--count", "2", "--quiet",
                    "--no-preprocess", "--no-update"],
                   check=True, capture_output=True)
    text = open(tmp_path / "out.txt").read().split("\n")
    M = ped.n_markers
    n_named = 2 + 3 * len(ped.dous) + 1           # A, B, every F2 with its two private F1 parents, "haplo"
    # iteration 0: the dump alone
    assert text[0] == "1 A"
    assert text[1].startswith("0.500000\t1\t1\t\t0.000000\t0.020000 0.020000 0.500000\t1\t1\t")
    pos = n_named * (1 + M)
    o = oracle_ped(ped)
    for c in range(2):
        first, last = int(ped.chromstarts[c]), int(ped.chromstarts[c + 1]) - 1
        raw = [o.sweep_ind(int(r), 2, first=first, last=last, mode=2)["dosage"] for r in ped.dous]
        for j, r in enumerate(ped.dous):
            assert text[pos] == "%s:%d" % (names[r], c + 1)
            pos += 1
            for m in range(first, last + 1):
                got = [float(x) for x in text[pos].split("\t")]
                assert len(got) == 3
                assert np.allclose(got, raw[j][m - first], atol=6e-6)
                pos += 1
            assert text[pos] == ""
            pos += 1
    # then the dump of the last iteration: "n name" followed by one line per marker (cnF2freq.cpp:8168-8186)
    assert text[pos] == "1 A"


def test_advanced_intercross_with_active_ties_against_oracle(capi):
    """BASELINE config 3 shape: deeper pedigree, explicit parents, sib matings => heterozygous
    ancestors in several window slots (general kernel) next to untied windows (fast kernel)."""
    ped = synth.make_ail(6, 10, 4, 40, 2, seed=11, chrom_cm=60.0, missing=0.05)
    ctx = capi.Context(0)
    ctx.upload(ped)
    tied = sum(1 for j in range(len(ped.dous)) if (ctx.window_info(j)["tie"] >= 0).any())
    assert 0 < tied < len(ped.dous), "the fixture should mix tied and untied windows"
    o = oracle_ped(ped)
    # tied windows through the tile-producer kernel (a backward pass per tie combination: the default) and through the
    # general kernel (per-marker producer, loop over the combinations inside a marker)
    for general in (False, True):
        got = ctx.sweep(ties_general=general)
        for c in range(2):
            first, last = int(ped.chromstarts[c]), int(ped.chromstarts[c + 1]) - 1
            want = o.sweep_batch(ped.dous, ped.gen[ped.dous], first=first, last=last, mode=2)
            np.testing.assert_allclose(got["factors"][:, c], want["factors"], rtol=RTOL, atol=1e-8)
            np.testing.assert_allclose(got["dosage"][:, first:last + 1], want["dosage"], rtol=1e-7, atol=1e-11)
    ctx.close()


def test_reference_fanout_semantics_on_tied_windows(capi):
    """Same as above against oracle mode 0 (the reference's own (g, s, path) fan-out) on a few
    individuals: ties are where the closed form is least obvious."""
    ped = synth.make_ail(4, 6, 3, 6, 1, seed=5, chrom_cm=20.0)
    ctx = capi.Context(0)
    ctx.upload(ped)
    raw = ctx.sweep(raw=True)
    o = oracle_ped(ped)
    checked = 0
    for j, ind in enumerate(ped.dous):
        if not (ctx.window_info(j)["tie"] >= 0).any():
            continue
        r0 = o.sweep_ind(int(ind), int(ped.gen[ind]), mode=0)
        if r0["ok"]:
            np.testing.assert_allclose(raw["dosage"][j], r0["dosage"], rtol=1e-7, atol=1e-12)
            checked += 1
    assert checked > 0
    ctx.close()


def test_full_length_chromosome_properties(capi):
    """Size-independent checks at BASELINE config 2's chromosome length (2 501 markers), where the
    oracle would be slow: rows are distributions, results do not depend on how individuals are
    batched or on the job's position in the grid, likelihoods are finite and negative."""
    ped = synth.make_f2(300, 2500, 1, seed=2, chrom_cm=100.0)
    ctx = capi.Context(0)
    ctx.upload(ped)
    full = ctx.sweep()
    s = full["dosage"].sum(axis=2)
    assert np.allclose(s, 1.0, atol=1e-12)
    assert np.all(full["dosage"] >= 0)
    assert np.all(np.isfinite(full["loglik"])) and np.all(full["loglik"] < 0)
    a = ctx.sweep(0, 7)
    b = ctx.sweep(7, 300)
    assert np.array_equal(np.concatenate([a["dosage"], b["dosage"]]), full["dosage"])
    assert np.array_equal(np.concatenate([a["loglik"], b["loglik"]]), full["loglik"])
    # logsumexp identity between the per-mode and the total likelihoods
    f = full["factors"][:, 0]
    mx = f.max(axis=1, keepdims=True)
    assert np.allclose(mx[:, 0] + np.log(np.exp(f - mx).sum(axis=1)), full["loglik"][:, 0], rtol=1e-12)
    # spot check 3 individuals against the oracle at full length
    o = oracle_ped(ped)
    want = o.sweep_batch(ped.dous[:3], ped.gen[ped.dous[:3]], mode=2)
    np.testing.assert_allclose(full["dosage"][:3], want["dosage"], rtol=1e-6, atol=1e-10)
    np.testing.assert_allclose(full["loglik"][:3, 0], want["factor"], rtol=1e-10)
    ctx.close()


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_turn_scan_matches_reference(capi, case):
    """rawervals[turn][s] (HOT LOOP 3, aroundturner) against the reference's own values."""
    ped, z = load_golden(case)
    ctx = capi.Context(0)
    ctx.upload(ped)
    for j in range(len(ped.dous)):
        if not z["ok"][j]:
            continue
        rows = ctx.turn_scan_rows(j, 0)                   # every marker of the chromosome in one launch
        for ti, m in enumerate(z["turn_markers"]):
            got = ctx.turn_scan(j, 0, int(m))
            want = z["rawervals"][j, ti]
            live = ~np.isnan(want)
            np.testing.assert_allclose(got[live], want[live], rtol=1e-9, atol=1e-8)
            assert np.array_equal(rows[int(m)], got)
    ctx.close()


def test_locked_queries_and_state_rows_against_oracle(capi):
    """Every term of HOT LOOP 2 at a marker (val for each shift mode, state and path) against the
    oracle's doanalyze(classicstop(q, g), flag2); their sums by class must give the dosage row and
    their sums by state the statereporter row."""
    ped = synth.make_random_windows(10, 5, seed=123)
    ctx = capi.Context(0)
    ctx.upload(ped)
    o = oracle_ped(ped)
    raw = ctx.sweep(raw=True)
    checked = 0
    for j, ind in enumerate(ped.dous):
        gen = int(ped.gen[ind])
        if not o.sweep_ind(int(ind), gen, mode=2)["ok"]:
            continue
        state_rows = ctx.state_posterior(j, 0)
        for m in (0, 3):
            want, mapval = o.val_table(int(ind), m, gen)
            got = ctx.locked_query(j, 0, m)
            live = want >= 0
            np.testing.assert_allclose(got[live], want[live], rtol=1e-8, atol=1e-13)
            row = np.array([got[live & (mapval == d)].sum() for d in range(3)])
            np.testing.assert_allclose(row, raw["dosage"][j, m], rtol=1e-8, atol=1e-12)
            by_state = np.where(live, got, 0.0).sum(axis=(0, 2))
            np.testing.assert_allclose(state_rows[m], by_state, rtol=1e-8, atol=1e-12)
            checked += 1
    assert checked >= 6
    ctx.close()


def _oracle_with(ped, **kw):
    from oracle.pyoracle import OraclePed
    a, s, h = ped.dense()
    return OraclePed(a, s, h, ped.par, ped.empty, ped.pos, **kw)


def test_baseline_config1_f2_200x500(capi):
    """BASELINE configs[0]: F2, 200 individuals x 1 chromosome x 500 SNPs (+ dummy marker), every
    individual against the oracle."""
    ped = synth.make_f2(200, 500, 1, seed=12345, chrom_cm=100.0)
    ctx = capi.Context(0)
    ctx.upload(ped)
    got = ctx.sweep()
    want = oracle_ped(ped).sweep_batch(ped.dous, ped.gen[ped.dous], mode=2)
    np.testing.assert_allclose(got["factors"][:, 0], want["factors"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(got["loglik"][:, 0], want["factor"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(got["dosage"], want["dosage"], rtol=1e-7, atol=1e-11)
    ctx.close()


def test_generation_below_two_uses_two_shift_modes(capi):
    """shiftend = 2 for gen < 2 (cnF2freq.cpp:5359): only modes 0 and 1 take part."""
    ped = synth.make_outbred3(2, 3, 20, 1, seed=8, missing=0.1, random_hw=True)
    ped.gen = ped.gen.copy()
    ped.gen[ped.dous[::2]] = 1
    ctx = capi.Context(0)
    ctx.upload(ped)
    got = ctx.sweep(raw=True)
    o = oracle_ped(ped)
    for j, ind in enumerate(ped.dous):
        r = o.sweep_ind(int(ind), int(ped.gen[ind]), mode=2)
        live = r["factors"] > -1e29
        assert live.sum() == (2 if ped.gen[ind] < 2 else 8)
        np.testing.assert_allclose(got["factors"][j, 0][live], r["factors"][live], rtol=RTOL, atol=1e-8)
        assert np.all(got["factors"][j, 0][~live] == -1e30)
        np.testing.assert_allclose(got["loglik"][j, 0], r["factor"], rtol=RTOL, atol=1e-8)
        np.testing.assert_allclose(got["dosage"][j], r["dosage"], rtol=1e-7, atol=1e-12)
    ctx.close()


def test_unequal_recombination_rates_per_generation(capi):
    """genrec[0] != genrec[1]: state bits 0 and 3 (TYPEGENS == 1) use a different rho than the
    grandparental bits (settings.h:23, cnF2freq.cpp:2329-2340)."""
    ped = synth.make_outbred3(2, 2, 25, 1, seed=4, missing=0.1)
    genrec = (-0.035, -0.011, -0.02)
    ctx = capi.Context(0)
    ctx.upload_map(ped.pos, ped.chromstarts, genrec)
    ctx.upload_rows(ped.allele, ped.sure, ped.hw)
    ctx.upload_pedigree(ped.par, ped.empty, ped.gen, ped.row_of, ped.dous)
    got = ctx.sweep()
    want = _oracle_with(ped, genrec=genrec).sweep_batch(ped.dous, ped.gen[ped.dous], mode=2)
    np.testing.assert_allclose(got["factors"][:, 0], want["factors"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(got["dosage"], want["dosage"], rtol=1e-7, atol=1e-11)
    base = oracle_ped(ped).sweep_batch(ped.dous, ped.gen[ped.dous], mode=2)
    assert np.abs(base["factor"] - want["factor"]).max() > 1e-3   # the rates really matter
    ctx.close()


def test_sex_marker_sentinel_alleles(capi):
    """Allele value 9 (sexmarkerval, cnF2freq.cpp:226,313): never matched by an unknown allele."""
    ped = synth.make_random_windows(32, 8, seed=909)
    rs = np.random.RandomState(1)
    a = ped.allele.copy()
    hit = rs.rand(*a.shape[:2]) < 0.08
    hit[0] = False
    a[hit, 1] = 9
    ped.allele = a
    ctx = capi.Context(0)
    ctx.upload(ped)
    got = ctx.sweep(raw=True)
    o = oracle_ped(ped)
    for j, ind in enumerate(ped.dous):
        r = o.sweep_ind(int(ind), int(ped.gen[ind]), mode=2)
        live = r["factors"] > -1e29
        np.testing.assert_allclose(got["factors"][j, 0][live], r["factors"][live], rtol=RTOL, atol=1e-8)
        if r["factor"] < -1e14:
            # every mode impossible: factor = MINFACTOR + log(8) is not < MINFACTOR, so the reference goes
            # on and adds up noise (cnF2freq.cpp:5403); the build reports zero rows
            assert np.all(got["dosage"][j] == 0)
        elif r["ok"]:
            np.testing.assert_allclose(got["dosage"][j], r["dosage"], rtol=1e-7, atol=1e-12)
    ctx.close()


def test_impossible_individual_is_reported_in_band(capi):
    """Contradictory genotypes with sure = 0 make every path impossible: the likelihood collapses
    to MINFACTOR (cnF2freq.cpp:1656-1660), no exception, rows of that individual are zero, the
    neighbours are untouched."""
    ped = synth.make_outbred3(1, 3, 12, 1, seed=2, missing=0.0)
    kid, p0 = int(ped.dous[1]), int(ped.par[ped.dous[1], 0])
    p1 = int(ped.par[ped.dous[1], 1])
    for r, al in ((kid, (2, 2)), (p0, (1, 1)), (p1, (1, 1))):
        ped.allele[ped.row_of[r], 5] = al
        ped.sure[ped.row_of[r], 5] = 0.0
    # the other children must not see the edited parents at marker 5 as impossible
    for other in (int(ped.dous[0]), int(ped.dous[2])):
        ped.allele[ped.row_of[other], 5] = (1, 1)
    ctx = capi.Context(0)
    ctx.upload(ped)
    got = ctx.sweep()
    o = oracle_ped(ped)
    r = o.sweep_ind(kid, 2, mode=2)
    assert not r["ok"] or r["factor"] <= capi.MINFACTOR + 16
    assert got["loglik"][1, 0] <= capi.MINFACTOR + 16
    assert np.all(got["dosage"][1] == 0)
    for j in (0, 2):
        rr = o.sweep_ind(int(ped.dous[j]), 2, mode=2)
        assert rr["ok"]
        np.testing.assert_allclose(got["loglik"][j, 0], rr["factor"], rtol=RTOL, atol=1e-8)
        d = rr["dosage"] / rr["dosage"].sum(axis=1, keepdims=True)
        np.testing.assert_allclose(got["dosage"][j], d, rtol=1e-7, atol=1e-11)
    ctx.close()


def test_update_rows_between_sweeps_and_tie_activation(capi):
    """cnf2_update_rows (the per-iteration parameter updates stay on the host): new weights are
    used by the next sweep, and an F2 founder that stops being homozygous re-activates the
    all-or-none rule (the window moves from the fast to the general kernel)."""
    ped = synth.make_f2(6, 18, 1, seed=77, chrom_cm=30.0, missing=0.1)
    ctx = capi.Context(0)
    ctx.upload(ped)
    before = ctx.sweep(raw=True)
    rs = np.random.RandomState(3)
    ped.hw[3:] = 0.1 + 0.8 * rs.rand(*ped.hw[3:].shape)             # F2 rows: new haplotype weights
    ped.sure[3:] = np.where(ped.allele[3:] != 0, 0.01 + 0.05 * rs.rand(*ped.sure[3:].shape), 0.0)
    ped.allele[1, 4] = (1, 2)                                        # founder A heterozygous at marker 4
    ped.hw[1, 4] = 0.3
    ctx.update_rows(1, ped.allele[1:], ped.sure[1:], ped.hw[1:])
    after = ctx.sweep(raw=True)
    assert np.abs(after["loglik"] - before["loglik"]).max() > 1e-3
    o = oracle_ped(ped)
    for j, ind in enumerate(ped.dous):
        r = o.sweep_ind(int(ind), 2, mode=0 if j < 2 else 2)         # two of them through the full fan-out
        np.testing.assert_allclose(after["loglik"][j, 0], r["factor"], rtol=RTOL, atol=1e-8)
        np.testing.assert_allclose(after["dosage"][j], r["dosage"], rtol=1e-7, atol=1e-12)
    ctx.close()


def test_haplos_accumulators_against_oracle(capi):
    """What HOT LOOP 2 leaves in `haplos` per window member (the input of movehaplos): GPU rows per
    slot, summed per individual, against the oracle's update-mode (HAPLOS) fan-out, which itself is
    pinned bit-exact on the reference's updatehaplo."""
    for ped in (synth.make_random_windows(14, 4, seed=41), synth.make_ail(4, 6, 3, 5, 1, seed=5, chrom_cm=20.0),
                synth.make_f2(3, 6, 1, seed=5, chrom_cm=20.0, missing=0.2)):
        ctx = capi.Context(0)
        ctx.upload(ped)
        o = oracle_ped(ped)
        checked = 0
        for j, ind in enumerate(ped.dous):
            gen = int(ped.gen[ind])
            if not o.sweep_ind(int(ind), gen, mode=2)["ok"]:
                continue
            rows = ctx.haplos(j, 0)
            slots = ctx.window_info(j)["slots"]
            for m in (0, ped.n_markers - 1):
                want = o.haplos_row(int(ind), m, gen)
                got = np.zeros_like(want)
                for k, r in enumerate(slots):
                    if r >= 0:
                        got[r] += rows[m, k]
                np.testing.assert_allclose(got, want, rtol=1e-8, atol=1e-12)
                checked += 1
        assert checked > 0
        ctx.close()


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_haplos_accumulators_match_reference(capi, case):
    ped, z = load_golden(case)
    ctx = capi.Context(0)
    ctx.upload(ped)
    for j in range(len(ped.dous)):
        if not z["ok"][j]:
            continue
        rows = ctx.haplos(j, 0)
        slots = ctx.window_info(j)["slots"]
        for ti, m in enumerate(z["turn_markers"]):
            got = np.zeros((ped.n_rec, 2))
            for k, r in enumerate(slots):
                if r >= 0:
                    got[r] += rows[int(m), k]
            np.testing.assert_allclose(got, z["haplos"][j, ti], rtol=1e-8, atol=1e-12)
    ctx.close()


def _infprobs_by_record(ctx, j, m, n_rec):
    inf, hz = ctx.infprobs(j, int(m))
    got = np.zeros((n_rec, 2, 2))
    for k, r in enumerate(ctx.window_info(j)["slots"]):
        if r >= 0:
            got[r] += inf[k]
    return got, hz


def test_infprobs_homozyg_accumulators_against_oracle(capi):
    """infprobs (GENOSPROBE weights, GENOS updates) and homozyg (HOMOZYGOUS) accumulators of HOT LOOP 2 per
    window slot, summed per individual, against the oracle's update-mode fan-out (pinned on the reference)."""
    for ped in (synth.make_random_windows(14, 4, seed=41), synth.make_ail(4, 6, 3, 5, 1, seed=5, chrom_cm=20.0),
                synth.make_f2(3, 6, 1, seed=5, chrom_cm=20.0, missing=0.2)):
        ctx = capi.Context(0)
        ctx.upload(ped)
        o = oracle_ped(ped)
        checked = 0
        for j, ind in enumerate(ped.dous):
            gen = int(ped.gen[ind])
            if not o.sweep_ind(int(ind), gen, mode=2)["ok"]:
                continue
            for m in (0, ped.n_markers - 1):
                want, want_hz = o.infprobs_row(int(ind), m, gen)
                got, hz = _infprobs_by_record(ctx, j, m, ped.n_rec)
                np.testing.assert_allclose(got, want, rtol=1e-8, atol=1e-12)
                np.testing.assert_allclose(hz, want_hz, rtol=1e-8, atol=1e-12)
                checked += 1
        assert checked > 0
        ctx.close()


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_infprobs_homozyg_accumulators_match_reference(capi, case):
    ped, z = load_golden(case)
    ctx = capi.Context(0)
    ctx.upload(ped)
    for j in range(len(ped.dous)):
        if not z["ok"][j]:
            continue
        for ti, m in enumerate(z["turn_markers"]):
            got, hz = _infprobs_by_record(ctx, j, m, ped.n_rec)
            np.testing.assert_allclose(got, z["infprobs"][j, ti], rtol=1e-8, atol=1e-12)
            np.testing.assert_allclose(hz, z["homozyg"][j, ti], rtol=1e-8, atol=1e-12)
    ctx.close()


def test_infprobs_rows_closed_form_against_fanout_and_oracle(capi):
    """cnf2_infprobs_rows (closed form, every marker of the chromosome) against the brute-force hook on the
    same device and against the oracle; tied (AIL), outbred with missing data, random windows."""
    for ped in (synth.make_random_windows(10, 4, seed=43), synth.make_ail(4, 6, 3, 5, 1, seed=5, chrom_cm=20.0),
                synth.make_outbred3(2, 2, 9, 1, seed=3, missing=0.2, random_hw=True, random_sure=True)):
        ctx = capi.Context(0)
        ctx.upload(ped)
        o = oracle_ped(ped)
        checked = 0
        for j, ind in enumerate(ped.dous):
            gen = int(ped.gen[ind])
            if not o.sweep_ind(int(ind), gen, mode=2)["ok"]:
                continue
            inf, hz = ctx.infprobs_rows(j, 0)
            slots = ctx.window_info(j)["slots"]
            for m in range(ped.n_markers):
                bi, bh = ctx.infprobs(j, m)
                np.testing.assert_allclose(inf[m], bi, rtol=1e-8, atol=1e-12)
                np.testing.assert_allclose(hz[m], bh, rtol=1e-8, atol=1e-12)
            for m in (0, ped.n_markers - 1):
                want, want_hz = o.infprobs_row(int(ind), m, gen)
                got = np.zeros_like(want)
                for k, r in enumerate(slots):
                    if r >= 0:
                        got[r] += inf[m, k]
                np.testing.assert_allclose(got, want, rtol=1e-8, atol=1e-12)
                np.testing.assert_allclose(hz[m], want_hz, rtol=1e-8, atol=1e-12)
                checked += 1
        assert checked > 0
        ctx.close()


def test_merge_modes_equals_plain_sweep(capi):
    """CNF2_MERGE_MODES: F2 individuals with private empty F1 parents go four to a wavefront and only the
    modes s0 = 0, 1 are swept; likelihoods must be bit-identical to the ordinary sweep for all 8 modes, rows
    equal to rounding (normalised and raw).  10 individuals = 2 groups of four + 2 left to the ordinary
    kernel; chromosomes of length 1, even and odd; missing genotypes."""
    ped = synth.make_f2(10, 21, 1, seed=9, chrom_cm=30.0, missing=0.15)
    ped.chromstarts = np.array([0, 1, 9, 22], np.int32)
    ped.pos = np.concatenate([[0.0], np.arange(8) * 0.9, np.arange(13) * 1.7])
    ctx = capi.Context(0)
    ctx.upload(ped)
    for raw in (False, True):
        a = ctx.sweep(raw=raw)
        b = ctx.sweep(raw=raw, merge_modes=True)
        # the two kernels run the same arithmetic on the same numbers; whether the results agree to the last bit is up to
        # the compiler's choice of fused multiply-adds in each, so the bar is rounding level
        np.testing.assert_allclose(b["factors"], a["factors"], rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(b["loglik"], a["loglik"], rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(b["dosage"], a["dosage"], rtol=1e-9, atol=1e-14)
    # a pedigree without such parents: the flag changes nothing
    ped2 = synth.make_outbred3(3, 3, 11, 1, seed=12, missing=0.1)
    ctx2 = capi.Context(0)
    ctx2.upload(ped2)
    a, b = ctx2.sweep(), ctx2.sweep(merge_modes=True)
    assert np.array_equal(a["dosage"], b["dosage"]) and np.array_equal(a["factors"], b["factors"])
    ctx.close()
    ctx2.close()


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_addvariance_matches_reference(capi, case):
    """cnf2_addvariance (pre-processing user of the emission, cnF2freq.cpp:1489-1558) for the analysed
    individuals against the reference's variances[] (goldens G10)."""
    ped, z = load_golden(case)
    ctx = capi.Context(0)
    ctx.upload(ped)
    for j, ind in enumerate(ped.dous):
        got = ctx.addvariance(j, 0)
        want = z["variances"][int(ind)]
        assert np.array_equal(np.isnan(got), np.isnan(want))
        ok = ~np.isnan(want)
        np.testing.assert_allclose(got[ok], want[ok], rtol=1e-9, atol=1e-14)
    ctx.close()


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_hot_loop_2_reductions_match_reference(capi, case):
    """cnf2_accumulate (accumulator rows on the GPU, moveinfprobs / movehaplos reductions on the host) against
    the reference's per-individual infprobs, haplobase, haplocount and homozyg (goldens G11); descendant
    counts from cnf2_descendants."""
    ped, z = load_golden(case)
    ctx = capi.Context(0)
    ctx.upload(ped)
    # descendant counts as the reference's postmarkerdata left them (individ::descendants; every round of its outer
    # inference loop propagates all counts again, cnF2freq.cpp:3224-3255): an input here.  cnf2_descendants is one round.
    desc = z["acc_desc"]
    assert np.all(ctx.descendants() <= desc)
    got = ctx.accumulate(desc)
    for k in ("infprobs", "haplobase", "haplocount", "homozyg"):
        np.testing.assert_allclose(got[k], z["acc_" + k], rtol=1e-8, atol=1e-12, equal_nan=True)
    ctx.close()


def test_error_behaviour_of_the_c_abi(capi):
    """Errors are status codes with a message, never exceptions or aborts on the library side (the reference
    aborts, cnF2freq.cpp:21-25): calls before the uploads, out-of-range individuals / chromosomes / markers."""
    ctx = capi.Context(0)
    with pytest.raises(capi.Cnf2Error, match="uploaded first"):
        ctx.sweep(0, 0)
    ped = synth.make_f2(2, 5, 1, seed=1, chrom_cm=10.0)
    ctx.upload(ped)
    with pytest.raises(capi.Cnf2Error, match="out of bounds"):
        ctx.sweep(0, 3)
    with pytest.raises(capi.Cnf2Error, match="out of range"):
        ctx.locked_query(0, 1, 0)
    with pytest.raises(capi.Cnf2Error, match="out of range"):
        ctx.fwbw_store(5, 0)
    with pytest.raises(capi.Cnf2Error, match="marker"):
        ctx.infprobs(0, 99)
    with pytest.raises(capi.Cnf2Error, match="marker"):
        ctx.turn_scan(0, 0, 99)
    ok = ctx.sweep()                      # the context stays usable after errors
    assert np.all(np.isfinite(ok["loglik"]))
    ctx.close()


def test_half_spill_recompute_equals_full_spill(capi):
    """Default: alpha-minus stored at every second marker, the odd ones rebuilt in the backward
    pass by one forward step.  Must give exactly what storing every marker gives (even and odd
    chromosome lengths, one-marker chromosome)."""
    ped = synth.make_outbred3(3, 3, 21, 1, seed=12, missing=0.15, random_hw=True, random_sure=True)
    ped.chromstarts = np.array([0, 1, 9, 22], np.int32)              # lengths 1, 8 (even), 13 (odd)
    ped.pos = np.concatenate([[0.0], np.arange(8) * 0.9, np.arange(13) * 1.7])
    ctx = capi.Context(0)
    ctx.upload(ped)
    half = ctx.sweep(raw=True)
    full = ctx.sweep(raw=True, full_spill=True)
    np.testing.assert_allclose(half["factors"], full["factors"], rtol=1e-13, atol=1e-12)
    np.testing.assert_allclose(half["dosage"], full["dosage"], rtol=1e-11, atol=1e-15)
    o = oracle_ped(ped)
    for c in range(3):
        first, last = int(ped.chromstarts[c]), int(ped.chromstarts[c + 1]) - 1
        want = o.sweep_batch(ped.dous, ped.gen[ped.dous], first=first, last=last, mode=2)
        d = ctx.sweep()["dosage"][:, first:last + 1]
        np.testing.assert_allclose(d, want["dosage"], rtol=1e-7, atol=1e-11)
    ctx.close()


def test_batched_accumulate_against_oracle_on_tied_and_ragged_pedigrees(capi):
    """cnf2_sweep_accumulate (the product form of HOT LOOP 2: every individual and chromosome in batched launches,
    path form of the accumulators in its tile layout, reductions with atomics on the device) against the oracle's accumulate
    (pinned bit-exact on the reference's own moveinfprobs / movehaplos): advanced intercross with active ties
    (general kernel + tie combinations), outbred with missing data, and an F2 over three ragged chromosomes.
    The sweep outputs that come with it must be those of cnf2_sweep."""
    peds = [synth.make_ail(4, 6, 3, 9, 1, seed=5, chrom_cm=20.0, missing=0.05),
            synth.make_outbred3(2, 3, 11, 1, seed=3, missing=0.2, random_hw=True, random_sure=True),
            synth.make_random_windows(24, 5, seed=43)]
    f2 = synth.make_f2(5, 17, 1, seed=3, chrom_cm=20.0, missing=0.1)
    f2.chromstarts = np.array([0, 1, 6, 18], np.int32)
    f2.pos = np.concatenate([[0.0], np.arange(5) * 0.7, np.arange(12) * 1.3])
    peds.append(f2)
    for ped in peds:
        ctx = capi.Context(0)
        ctx.upload(ped)
        desc = ctx.descendants()
        got = ctx.sweep_accumulate(desc)
        plain = ctx.sweep()
        # (not bit-equal: tied windows take the tile-producer kernel in cnf2_sweep and the general kernel here)
        np.testing.assert_allclose(got["factors"], plain["factors"], rtol=1e-12, atol=1e-9)
        np.testing.assert_allclose(got["loglik"], plain["loglik"], rtol=1e-12, atol=1e-9)
        np.testing.assert_allclose(got["dosage"], plain["dosage"], rtol=1e-12, atol=1e-15)
        o = oracle_ped(ped)
        for c in range(len(ped.chromstarts) - 1):
            first, last = int(ped.chromstarts[c]), int(ped.chromstarts[c + 1]) - 1
            want = o.accumulate(ped.dous, ped.gen[ped.dous], desc, first=first, last=last)
            for k, sl in (("infprobs", np.s_[:, first:last + 1]), ("haplobase", np.s_[:, first:last + 1]),
                          ("haplocount", np.s_[:, first:last + 1]), ("homozyg", np.s_[:, first:last + 1])):
                np.testing.assert_allclose(got[k][sl], want[k], rtol=1e-8, atol=1e-12, equal_nan=True, err_msg=k)
        # the table form (one lane per emission-table entry; the kernel of the windows whose root is the top of its
        # lines) on every window: the same sums as the path form, against the oracle too
        tab = ctx.sweep_accumulate(desc, table_form=True)
        for k in ("infprobs", "haplobase", "haplocount", "homozyg"):
            np.testing.assert_allclose(tab[k], got[k], rtol=1e-8, atol=1e-12, equal_nan=True, err_msg="table form " + k)
        # ... the sweep of the tied windows through the general kernel instead of the tile-producer kernel
        gen = ctx.sweep_accumulate(desc, ties_general=True)
        for k in ("infprobs", "haplobase", "haplocount", "homozyg", "dosage"):
            np.testing.assert_allclose(gen[k], got[k], rtol=1e-8, atol=1e-12, equal_nan=True, err_msg="general sweep " + k)
        # ... and the path form with one lane per path (the kernel of the windows with tie groups) on every window
        lanes = ctx.sweep_accumulate(desc, lane_form=True)
        for k in ("infprobs", "haplobase", "haplocount", "homozyg"):
            np.testing.assert_allclose(lanes[k], got[k], rtol=1e-8, atol=1e-12, equal_nan=True, err_msg="lane form " + k)
        # a sub-range of individuals gives that range's homozyg and (for disjoint windows) its share of the rest
        part = ctx.sweep_accumulate(desc, 1, 3)
        np.testing.assert_allclose(part["homozyg"], got["homozyg"][1:3], rtol=1e-12, atol=1e-15, equal_nan=True)
        ctx.close()


@pytest.mark.parametrize("seed", [101, 102, 103, 104, 105, 106])
def test_random_windows_sweep_and_accumulate_against_oracle(capi, seed):
    """More random pedigrees (missing and founder slots, restricted slots, tie groups, empty individuals) through the
    product routes of this round -- tied windows on the tile-producer kernel, accumulators in the tile form -- against
    the oracle: likelihoods, rows and every accumulator."""
    ped = synth.make_random_windows(24, 6, seed=seed)
    ctx = capi.Context(0)
    ctx.upload(ped)
    desc = ctx.descendants()
    got = ctx.sweep_accumulate(desc)
    plain = ctx.sweep()
    o = oracle_ped(ped)
    want = o.sweep_batch(ped.dous, ped.gen[ped.dous], mode=2)
    np.testing.assert_allclose(plain["factors"][:, 0], want["factors"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(plain["dosage"], want["dosage"], rtol=1e-7, atol=1e-11)
    np.testing.assert_allclose(got["dosage"], want["dosage"], rtol=1e-7, atol=1e-11)
    acc = o.accumulate(ped.dous, ped.gen[ped.dous], desc, first=0, last=ped.n_markers - 1)
    for k in ("infprobs", "haplobase", "haplocount", "homozyg"):
        np.testing.assert_allclose(got[k], acc[k], rtol=1e-8, atol=1e-12, equal_nan=True, err_msg=k)
    ctx.close()


def test_batched_accumulate_rows_equal_the_per_individual_hooks(capi):
    """The table form inside the batched kernel against the per-individual closed-form hooks (cnf2_haplos,
    cnf2_infprobs_rows), which read the reference-layout store: same accumulators before the reductions.  Checked
    through an individual without relatives in the batch, where the reductions are the identity up to known factors."""
    ped = synth.make_outbred3(1, 1, 13, 1, seed=6, missing=0.15, random_hw=True, random_sure=True)
    ctx = capi.Context(0)
    ctx.upload(ped)
    desc = np.ones(ped.n_rec, np.int32)
    got = ctx.sweep_accumulate(desc)
    inf, hz = ctx.infprobs_rows(0, 0)
    slots = ctx.window_info(0)["slots"]
    self0 = inf[:, 0, 0, :].sum(axis=1)                       # the individual's own allele-index-0 mass
    for k, r in enumerate(slots):
        # every member occupies one slot: norm = sum * 2 / 2 * descendants = 1 / self0
        np.testing.assert_allclose(got["infprobs"][r], inf[:, k] / self0[:, None, None], rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(got["homozyg"][0], hz / self0[:, None], rtol=1e-9, atol=1e-13)
    ctx.close()


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_batched_turn_scan_matches_reference(capi, case):
    """cnf2_sweep_turn_scan (every individual and marker in batched launches, alpha / beta straight from the sweep's
    registers) against the reference's rawervals (goldens G5) and, at every marker, against the per-individual hook
    that reads the reference-layout store."""
    ped, z = load_golden(case)
    ctx = capi.Context(0)
    ctx.upload(ped)
    raw, lse = ctx.sweep_turn_scan()                       # the dot products on the matrix cores (default)
    raw_v, lse_v = ctx.sweep_turn_scan(valu=True)          # ... on the vector ALU: the same sums in another order
    both = np.isfinite(raw) & np.isfinite(raw_v) & (np.abs(raw) < 1e14)
    assert np.array_equal(np.abs(raw) < 1e14, np.abs(raw_v) < 1e14)
    np.testing.assert_allclose(raw[both], raw_v[both], rtol=1e-11, atol=1e-10)
    np.testing.assert_allclose(lse, lse_v, rtol=1e-11, atol=1e-10)
    for j in range(len(ped.dous)):
        if not z["ok"][j]:
            continue
        for ti, m in enumerate(z["turn_markers"]):
            want = z["rawervals"][j, ti]
            live = ~np.isnan(want)
            np.testing.assert_allclose(raw[j, int(m)][live], want[live], rtol=1e-9, atol=1e-8)
        rows = ctx.turn_scan_rows(j, 0)
        fin = np.abs(rows) < 1e14
        np.testing.assert_allclose(raw[j][fin], rows[fin], rtol=1e-9, atol=1e-8)
        # the reduced form: log-sum-exp over the admissible modes, as computew forms it (cnF2freq.cpp:5800-5812)
        w = ctx.window_info(j)
        s_ok = np.array([not (s & w["shiftignore"]) and s < (8 if ped.gen[ped.dous[j]] >= 2 else 2) for s in range(8)])
        r = np.where(s_ok[None, None, :], raw[j], -np.inf)
        mx = np.maximum(r.max(axis=2), capi.MINFACTOR)
        want_lse = mx + np.log(np.exp(r - mx[:, :, None]).sum(axis=2))
        np.testing.assert_allclose(lse[j], want_lse, rtol=1e-10, atol=1e-9)
    ctx.close()


def test_batched_turn_scan_on_tied_windows_and_chromosomes(capi):
    """Same on an advanced intercross (tied windows) over two chromosomes, against the oracle: alpha and beta do not see
    the tie rule, so the tied windows take the tile-producer kernel too; the general kernel's route must agree."""
    ped = synth.make_ail(4, 6, 3, 7, 2, seed=5, chrom_cm=20.0, missing=0.05)
    ctx = capi.Context(0)
    ctx.upload(ped)
    raw, _ = ctx.sweep_turn_scan(lse=False)
    raw_general, _ = ctx.sweep_turn_scan(lse=False, ties_general=True)
    both = ~np.isnan(raw) & ~np.isnan(raw_general)
    assert np.array_equal(np.isnan(raw), np.isnan(raw_general))
    np.testing.assert_allclose(raw[both], raw_general[both], rtol=1e-9, atol=1e-8)
    o = oracle_ped(ped)
    checked = 0
    for j, ind in enumerate(ped.dous):
        gen = int(ped.gen[ind])
        for c in range(2):
            first, last = int(ped.chromstarts[c]), int(ped.chromstarts[c + 1]) - 1
            if not o.sweep_ind(int(ind), gen, first=first, last=last, mode=2)["ok"]:
                continue
            for m in (first, last):
                want = o.turn_scan(int(ind), m, gen, first=first, last=last)
                live = ~np.isnan(want)
                np.testing.assert_allclose(raw[j, m][live], want[live], rtol=1e-9, atol=1e-8)
                checked += 1
    assert checked > 10
    ctx.close()


def test_transposing_sweep_variant_equals_the_dpp_variant(capi):
    """CNF2_XPOSE: the fast kernel with its three lane-held state bits brought into registers by a transpose through
    LDS instead of DPP exchanges (the layout alternates from marker to marker).  Same arithmetic in another order of
    lanes: results equal to rounding.  F2 (single-allele producer shortcuts), outbred with missing data and random
    weights, ragged chromosomes (lengths 1, even, odd), and against the oracle."""
    f2 = synth.make_f2(9, 21, 1, seed=9, chrom_cm=30.0, missing=0.15)
    f2.chromstarts = np.array([0, 1, 9, 22], np.int32)
    f2.pos = np.concatenate([[0.0], np.arange(8) * 0.9, np.arange(13) * 1.7])
    out = synth.make_outbred3(3, 3, 33, 2, seed=12, missing=0.2, random_hw=True, random_sure=True)
    for ped in (f2, out, synth.make_random_windows(40, 7, seed=91)):
        ctx = capi.Context(0)
        ctx.upload(ped)
        for raw in (False, True):
            a = ctx.sweep(raw=raw)
            b = ctx.sweep(raw=raw, xpose=True)
            np.testing.assert_allclose(b["factors"], a["factors"], rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(b["loglik"], a["loglik"], rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(b["dosage"], a["dosage"], rtol=1e-9, atol=1e-13)
        ctx.close()
    ctx = capi.Context(0)
    ctx.upload(out)
    got = ctx.sweep(xpose=True)
    o = oracle_ped(out)
    for c in range(2):
        first, last = int(out.chromstarts[c]), int(out.chromstarts[c + 1]) - 1
        want = o.sweep_batch(out.dous, out.gen[out.dous], first=first, last=last, mode=2)
        np.testing.assert_allclose(got["factors"][:, c], want["factors"], rtol=RTOL, atol=1e-8)
        np.testing.assert_allclose(got["dosage"][:, first:last + 1], want["dosage"], rtol=1e-7, atol=1e-11)
    ctx.close()


def test_sparse_rescaling_guard_on_data_that_loses_many_decades(capi):
    """The half-spill sweep rescales its vectors once per tile of 8 markers; a stretch of data that loses more than
    150 decades between two rescalings switches the wave to dense rescaling.  Genotypes that contradict the pedigree
    with tiny certainties (every marker costs ~1e-24) against the oracle, which rescales at every marker like the
    reference (cnF2freq.cpp:1664-1668)."""
    ped = synth.make_outbred3(2, 2, 60, 1, seed=5, missing=0.0)
    ped.allele = ped.allele.copy()
    ped.sure = ped.sure.copy()
    kid = int(ped.dous[0])
    p0, p1 = int(ped.par[kid, 0]), int(ped.par[kid, 1])
    for r, al in ((kid, (2, 2)), (p0, (1, 1)), (p1, (1, 1))):
        ped.allele[ped.row_of[r], 10:50] = al
        ped.sure[ped.row_of[r], 10:50] = 1e-6
    ctx = capi.Context(0)
    ctx.upload(ped)
    got = ctx.sweep()
    o = oracle_ped(ped)
    want = o.sweep_batch(ped.dous, ped.gen[ped.dous], mode=2)
    assert want["factor"][0] < -400, "the fixture should lose hundreds of log units"
    np.testing.assert_allclose(got["loglik"][:, 0], want["factor"], rtol=1e-9, atol=1e-8)
    np.testing.assert_allclose(got["dosage"], want["dosage"], rtol=1e-6, atol=1e-10)
    ctx.close()


def test_states_the_reference_flushes_below_1e300_are_carried_here_without_effect(capi):
    """Known deviation, bounded: adjustprobs sets a state whose normalised probability has fallen below 1e-300 to exactly 0
    before it evaluates the emission (cnF2freq.cpp:1607-1611); the kernels carry such a value on (it is 1e-300 of the
    vector: nothing of it reaches a double's 16 digits).  Adversarial but valid data that puts states into that band: a
    stretch of markers at ONE map position (no transition between them, cnF2freq.cpp:2273, so nothing leaks back into a
    state) at which a nearly phase-locked heterozygous parent and a homozygous child make the states that inherit the
    parent's other strand lose a factor ~1e-2 per marker -- 2e-230 after 112 markers, measured; 1e-311 after 152 -- followed by
    ordinary markers.  Likelihoods
    and rows must still equal the oracle's (which applies the rule, bit-exact on the reference extract).  What is NOT
    reproduced: data under which every OTHER state then becomes exactly impossible -- the reference
    declares the individual impossible (MINFACTOR), the kernels return the likelihood of the 1e-303 state."""
    ped = synth.make_outbred3(1, 2, 190, 1, seed=9, missing=0.0, chrom_cm=30.0)
    ped.allele, ped.sure, ped.hw, ped.pos = ped.allele.copy(), ped.sure.copy(), ped.hw.copy(), ped.pos.copy()
    kid = int(ped.dous[0])
    p0, p1 = int(ped.par[kid, 0]), int(ped.par[kid, 1])
    lo, hi = 10, 162                                        # 152 markers on one position
    ped.pos[lo:hi] = ped.pos[lo]
    for r, al in ((kid, (1, 1)), (p0, (1, 2)), (p1, (1, 1))):
        ped.allele[ped.row_of[r], lo:hi] = al
        ped.sure[ped.row_of[r], lo:hi] = 1e-3
    ped.hw[ped.row_of[p0], lo:hi] = 0.999                   # strand 0 of the parent carries allele 1, almost surely
    ctx = capi.Context(0)
    ctx.upload(ped)
    got = ctx.sweep()
    o = oracle_ped(ped)
    want = o.sweep_batch(ped.dous, ped.gen[ped.dous], mode=2)
    # the fixture does what it is built for: in the reference-layout store the smallest non-zero forward value of the
    # stretch's end is far below anything a sum of 64 states resolves, and some states are below 1e-300
    fw, _ = ctx.fwbw_store(0, 0)
    end = fw[:, hi - 1, 0, :]
    live = end[end > 0]
    assert live.min() < 1e-300 or (end == 0).any(), "no state fell below the reference's 1e-300 threshold: %g" % live.min()
    np.testing.assert_allclose(got["factors"][:, 0], want["factors"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(got["loglik"][:, 0], want["factor"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(got["dosage"], want["dosage"], rtol=1e-6, atol=1e-10)
    # ... and through the accumulate mode (HOT LOOP 2 sees the same vectors)
    desc = ctx.descendants()
    acc = ctx.sweep_accumulate(desc)
    ref = oracle_accumulate_threaded(o, ped, desc, 0, ped.n_markers - 1)
    for k in ("infprobs", "haplobase", "haplocount", "homozyg"):
        np.testing.assert_allclose(acc[k], ref[k], rtol=1e-6, atol=1e-10, equal_nan=True, err_msg=k)
    ctx.close()


def test_the_1e300_rule_to_the_letter_where_it_decides_a_result(capi):
    """Golden G15 (tests/golden/flush_impossible.npz: the reference's own likelihoods, oracle/_ref): after 152 markers on one map
    position half of a child's states are below 1e-300 of their vector and adjustprobs sets them to exactly 0
    (cnF2freq.cpp:1607-1611); one more marker there -- every genotype certain, the parent's phase locked the other way round --
    makes the surviving states exactly impossible, so the reference declares the four shift modes that sit on that strand
    impossible (MINFACTOR).  cnf2_sweep with CNF2_FLUSH_TINY (the general kernel, vectors normalised at every marker as the
    reference normalises them, the rule applied) gives the reference's factors to the letter; the default route is
    compared on the modes the rule does not touch."""
    ped, z = load_golden("flush_impossible")
    want = z["factors"]
    gone = want < -1e14
    assert gone.sum() == 4 and gone[0, [0, 1, 4, 5]].all()
    ctx = capi.Context(0)
    ctx.upload(ped)
    got = ctx.sweep(flush_tiny=True)
    f = got["factors"][:, 0]
    assert np.array_equal(f < -1e14, gone)
    np.testing.assert_allclose(f[~gone], want[~gone], rtol=RTOL, atol=1e-8)
    # the total is the log-sum-exp over the modes that are left (cnF2freq.cpp:5384-5400)
    for j in range(len(ped.dous)):
        live = want[j][~gone[j]]
        np.testing.assert_allclose(got["loglik"][j, 0], live.max() + np.log(np.exp(live - live.max()).sum()), rtol=1e-9)
    assert np.allclose(got["dosage"].sum(axis=2), 1.0, atol=1e-9)
    # the restatement applies the rule as well
    o = oracle_ped(ped).sweep_batch(ped.dous, ped.gen[ped.dous], mode=2)
    assert np.array_equal(o["factors"] < -1e14, gone)
    np.testing.assert_allclose(got["dosage"], o["dosage"], rtol=1e-6, atol=1e-10)
    # default route: the fast kernel does not apply the rule -- a state it carries at 1e-303 of its vector would give those
    # four modes a (tiny) likelihood.  On this fixture it gives the reference's answer all the same (measured: the four modes
    # come out impossible: between two of its rescalings, eight markers apart, the carried values leave the double range by
    # themselves); what the test pins is that the modes the rule does not touch are the reference's either way
    plain = ctx.sweep()["factors"][:, 0]
    assert ((plain[gone] < -1e14) | (plain[gone] < want[0][~gone[0]].max() + 50)).all()
    # (the child's other four modes lose the same states to the rule at the last marker of the stretch, where they are not
    # yet impossible without them: carried, they add 1.5e-3 to a log-likelihood of -1682 -- 9e-7 relative, the size of
    # the deviation where it shows at all; north_star's bar is 1e-6)
    np.testing.assert_allclose(plain[~gone], want[~gone], rtol=2e-6)
    ctx.close()


def _four_founder_pedigree(n_kids, M, seed, empty_f1=True, empty_gp=False):
    """n_kids analysed children of two F1 parents with four distinct, heterozygous grandparents (no ancestor in two
    slots: no tie group).  empty_f1: the parents have no data (homozygous-everywhere blank rows); empty_gp: one
    grandparent has no data either (its slot is restricted: flag2ignore != 0)."""
    rs = np.random.RandomState(seed)
    R = 6 + n_kids
    par = np.full((R, 2), -1, np.int32)
    gen = np.zeros(R, np.int32)
    empty = np.zeros(R, np.uint8)
    par[4], par[5] = (0, 1), (2, 3)
    gen[4] = gen[5] = 1
    for k in range(n_kids):
        par[6 + k] = (4, 5)
        gen[6 + k] = 2
    allele = np.zeros((R + 1, M, 2), np.uint8)
    allele[1:] = rs.choice([1, 2], size=(R, M, 2))
    allele[1:5][:, :, 0], allele[1:5][:, :, 1] = 1, 2            # grandparents heterozygous everywhere
    sure = np.where(allele != 0, 0.02, 0.0)
    hw = np.full((R + 1, M), 0.5)
    hw[1:] = 0.2 + 0.6 * rs.rand(R, M)
    row_of = np.arange(1, R + 1, dtype=np.int32)
    if empty_f1:
        empty[4] = empty[5] = 1
        row_of[4] = row_of[5] = 0
    if empty_gp:
        empty[3] = 1
        row_of[3] = 0
    pos = np.arange(M) * 1.5
    ped = synth.Pedigree(["r%d" % i for i in range(R)], par, gen, empty, row_of, allele, sure, hw, pos,
                         np.array([0, M], np.int32), np.arange(6, R, dtype=np.int32))
    ped.founder_flags()
    return ped


def test_every_producer_specialisation_runs_and_is_exact(capi):
    """The sweep picks, per job, a kernel and a table-producer specialisation from the data (exact shortcuts): each one
    must be reached by some fixture (cnf2_last_paths says which code swept a job) and agree with the oracle."""
    cases = [
        ("fast kernel, general producer (hom 0): a restricted slot, parents with data",
         _four_founder_pedigree(3, 9, 1, empty_f1=False, empty_gp=True), False, {0}),
        ("fast kernel, parents homozygous everywhere (hom 1)", _four_founder_pedigree(3, 9, 2), False, {1}),
        ("fast kernel, parents and grandparents homozygous everywhere (hom 2): the F2",
         synth.make_f2(5, 9, 1, seed=3, chrom_cm=20.0, missing=0.1), False, {2}),
        ("fast kernel, complete window (hom 3)", synth.make_outbred3(1, 3, 9, 1, seed=4, missing=0.0, random_hw=True), False, {3}),
        ("merged-modes kernel, grandparents not homozygous (homleaf 0)", _four_founder_pedigree(5, 9, 5), True, {32, 1}),
        ("merged-modes kernel, grandparents homozygous (homleaf 1)",
         synth.make_f2(6, 9, 1, seed=6, chrom_cm=20.0, missing=0.1), True, {33, 2}),
        ("fast kernel, tied windows: a backward pass per tie combination (16)",
         synth.make_ail(4, 6, 3, 9, 1, seed=5, chrom_cm=20.0), False, {16, 3, 0}),
        ("general kernel: tied windows", synth.make_ail(4, 6, 3, 9, 1, seed=5, chrom_cm=20.0), "general", {64, 3, 0}),
    ]
    for what, ped, merge, allowed in cases:
        ctx = capi.Context(0)
        ctx.upload(ped)
        got = ctx.sweep(merge_modes=merge is True, log_paths=True, ties_general=merge == "general")
        paths = set(int(x) for x in got["paths"].ravel())
        assert paths <= allowed and min(allowed) in paths or max(allowed) in paths, (what, paths)
        if merge is True:
            assert any(p >= 32 and p < 64 for p in paths), (what, paths)
        elif 64 in allowed or 16 in allowed:
            assert max(allowed) in paths, (what, paths)
        else:
            assert paths == allowed, (what, paths)
        want = oracle_ped(ped).sweep_batch(ped.dous, ped.gen[ped.dous], mode=2)
        np.testing.assert_allclose(got["factors"][:, 0], want["factors"], rtol=RTOL, atol=1e-8, err_msg=what)
        np.testing.assert_allclose(got["dosage"], want["dosage"], rtol=1e-7, atol=1e-11, err_msg=what)
        ctx.close()
