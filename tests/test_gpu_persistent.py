"""GPU suite (-m gpu): the two code paths that test-size inputs never reached on their own.

(a) The persistent job loop.  The sweep kernels keep one wave per job in flight up to what is resident (256 CUs x 2
    blocks x 4 waves) and a wave that finishes a job takes the next one: LDS tables, spill slot, the mantissa / exponent
    likelihood state, the lane id -- everything a wave carries from job k to job k + 1 -- is only exercised when there are
    more jobs than resident waves.  cnf2_set_grid_reserve leaves all but ONE block free here, so 4 waves sweep every job
    of the fixture (50 or more each); the results must equal the unconstrained launch to the bit and the oracle at the
    usual tolerances -- plain sweep (cnF2freq.cpp:5294-5403), accumulate mode (5406-5583) and turn-scan mode (5668-5752).
(b) The multi-batch path of the batched consumers: cnf2_set_batch_jobs caps a batch at 3 jobs, so cnf2_sweep_accumulate
    and cnf2_sweep_turn_scan walk many batches on a two-chromosome pedigree; against the oracle and the one-batch run."""
import numpy as np
import pytest

from cnf2freq_amd import synth
from conftest import oracle_accumulate_threaded, oracle_ped

pytestmark = pytest.mark.gpu

RTOL = 1e-9
ONE_BLOCK = 1 << 20          # more slots than the GPU has: the grid is clamped to one block of 4 waves


@pytest.fixture(scope="module")
def capi():
    import __graft_entry__ as g
    g.build()
    from cnf2freq_amd import capi as c
    assert c.load().cnf2_device_count() >= 1, "no HIP device: the product path has no fallback"
    return c


def test_one_block_sweeps_config1_fifty_jobs_per_wave(capi):
    """BASELINE configs[0] (F2 200 x 501): 200 jobs on 4 waves."""
    ped = synth.make_f2(200, 500, 1, seed=12345, chrom_cm=100.0)
    ctx = capi.Context(0)
    ctx.upload(ped)
    free = ctx.sweep()
    ctx.set_grid_reserve(ONE_BLOCK)
    one = ctx.sweep()
    ctx.set_grid_reserve(0)
    for k in ("factors", "loglik", "dosage"):
        assert np.array_equal(one[k], free[k]), "one block differs from the full grid in " + k
    want = oracle_ped(ped).sweep_batch(ped.dous, ped.gen[ped.dous], mode=2)
    np.testing.assert_allclose(one["factors"][:, 0], want["factors"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(one["loglik"][:, 0], want["factor"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(one["dosage"], want["dosage"], rtol=1e-7, atol=1e-11)
    ctx.close()


def test_one_block_sweeps_the_tied_advanced_intercross(capi):
    """Advanced intercross with tied and untied windows over two chromosomes: both the plain and the tied instantiation
    run their jobs on one block each (the tied kernel's job loop also restores beta and the scales per tie combination)."""
    ped = synth.make_ail(6, 30, 4, 30, 2, seed=11, chrom_cm=60.0, missing=0.05)
    ctx = capi.Context(0)
    ctx.upload(ped)
    n = len(ped.dous)
    tied = sum(1 for j in range(n) if (ctx.window_info(j)["tie"] >= 0).any())
    assert tied * 2 >= 40 and (n - tied) * 2 >= 100, (tied, n)
    free = ctx.sweep()
    ctx.set_grid_reserve(ONE_BLOCK)
    one = ctx.sweep()
    ctx.set_grid_reserve(0)
    for k in ("factors", "loglik", "dosage"):
        assert np.array_equal(one[k], free[k]), "one block differs from the full grid in " + k
    o = oracle_ped(ped)
    for c in range(2):
        first, last = int(ped.chromstarts[c]), int(ped.chromstarts[c + 1]) - 1
        want = o.sweep_batch(ped.dous, ped.gen[ped.dous], first=first, last=last, mode=2)
        np.testing.assert_allclose(one["factors"][:, c], want["factors"], rtol=RTOL, atol=1e-8)
        np.testing.assert_allclose(one["dosage"][:, first:last + 1], want["dosage"], rtol=1e-7, atol=1e-11)
    ctx.close()


def _check_accumulators(got, ped, o, desc, rtol=1e-8):
    for c in range(len(ped.chromstarts) - 1):
        first, last = int(ped.chromstarts[c]), int(ped.chromstarts[c + 1]) - 1
        want = oracle_accumulate_threaded(o, ped, desc, first, last)
        for k in ("infprobs", "haplobase", "haplocount", "homozyg"):
            np.testing.assert_allclose(got[k][:, first:last + 1], want[k], rtol=rtol, atol=1e-12, equal_nan=True, err_msg=k)


@pytest.mark.parametrize("kind", ["outbred", "ail"])
def test_one_block_accumulate_mode(capi, kind):
    """The accumulate instantiation (sweep + posterior weights + HOT LOOP 2 kernels) with every job on one block."""
    if kind == "outbred":
        ped = synth.make_outbred3(25, 4, 6, 2, seed=21, missing=0.2, random_hw=True, random_sure=True)
    else:
        ped = synth.make_ail(6, 25, 4, 6, 2, seed=13, chrom_cm=40.0, missing=0.05)
    ctx = capi.Context(0)
    ctx.upload(ped)
    assert len(ped.dous) * 2 >= 200          # 50 jobs per wave of the one block
    desc = ctx.descendants()
    free = ctx.sweep_accumulate(desc, deterministic=True)
    ctx.set_grid_reserve(ONE_BLOCK)
    one = ctx.sweep_accumulate(desc, deterministic=True)
    one_atomic = ctx.sweep_accumulate(desc)
    ctx.set_grid_reserve(0)
    # CNF2_DETERMINISTIC: every individual's contributions in a row of its own, added in ascending order -- the grid
    # must not show in a single bit
    for k in ("factors", "loglik", "dosage", "infprobs", "haplobase", "haplocount", "homozyg"):
        assert np.array_equal(one[k], free[k], equal_nan=True), "one block differs from the full grid in " + k
    o = oracle_ped(ped)
    _check_accumulators(one, ped, o, desc)
    _check_accumulators(one_atomic, ped, o, desc)
    ctx.close()


def test_one_block_turn_scan_mode(capi):
    """The turn-scan instantiation (alpha after emission, beta and scales of every marker into the batch buffer) with
    every job on one block: equal to the full grid to the bit, and to the oracle's aroundturner queries."""
    ped = synth.make_outbred3(25, 4, 9, 2, seed=22, missing=0.2, random_hw=True, random_sure=True)
    ctx = capi.Context(0)
    ctx.upload(ped)
    raw_free, lse_free = ctx.sweep_turn_scan()
    ctx.set_grid_reserve(ONE_BLOCK)
    raw, lse = ctx.sweep_turn_scan()
    ctx.set_grid_reserve(0)
    assert np.array_equal(raw, raw_free, equal_nan=True) and np.array_equal(lse, lse_free, equal_nan=True)
    o = oracle_ped(ped)
    checked = 0
    for j in range(0, len(ped.dous), 5):
        ind = int(ped.dous[j])
        gen = int(ped.gen[ind])
        for c in range(2):
            first, last = int(ped.chromstarts[c]), int(ped.chromstarts[c + 1]) - 1
            if not o.sweep_ind(ind, gen, first=first, last=last, mode=2)["ok"]:
                continue
            for m in (first, (first + last) // 2, last):
                want = o.turn_scan(ind, m, gen, first=first, last=last)
                live = ~np.isnan(want)
                np.testing.assert_allclose(raw[j, m][live], want[live], rtol=1e-9, atol=1e-8)
                checked += 1
    assert checked > 20
    ctx.close()


@pytest.mark.parametrize("deterministic", [False, True])
def test_accumulate_in_batches_of_three_jobs(capi, deterministic):
    """cnf2_sweep_accumulate with the batch capped at 3 jobs (cnf2_set_batch_jobs) on a two-chromosome outbred pedigree
    with 20 % missing genotypes: 24 jobs = 8 batches per kernel, against the oracle and against the one-batch run."""
    ped = synth.make_outbred3(4, 3, 17, 2, seed=31, missing=0.2, random_hw=True, random_sure=True)
    ctx = capi.Context(0)
    ctx.upload(ped)
    desc = ctx.descendants()
    whole = ctx.sweep_accumulate(desc, deterministic=deterministic)
    ctx.set_batch_jobs(3)
    got = ctx.sweep_accumulate(desc, deterministic=deterministic)
    ctx.set_batch_jobs(0)
    for k in ("factors", "loglik", "dosage", "homozyg"):
        assert np.array_equal(got[k], whole[k], equal_nan=True), k
    for k in ("infprobs", "haplobase", "haplocount"):
        if deterministic:
            assert np.array_equal(got[k], whole[k], equal_nan=True), k
        else:
            np.testing.assert_allclose(got[k], whole[k], rtol=1e-12, atol=1e-15, equal_nan=True, err_msg=k)
    _check_accumulators(got, ped, oracle_ped(ped), desc)
    ctx.close()


def test_tied_accumulate_and_turn_scan_in_batches(capi):
    """The same cap on an advanced intercross (tied windows take their own kernels and their own batches) and on the
    batched turn scan."""
    ped = synth.make_ail(4, 6, 3, 9, 2, seed=5, chrom_cm=20.0, missing=0.05)
    ctx = capi.Context(0)
    ctx.upload(ped)
    desc = ctx.descendants()
    whole = ctx.sweep_accumulate(desc, deterministic=True)
    raw_whole, lse_whole = ctx.sweep_turn_scan()
    ctx.set_batch_jobs(3)
    got = ctx.sweep_accumulate(desc, deterministic=True)
    raw, lse = ctx.sweep_turn_scan()
    ctx.set_batch_jobs(0)
    for k in ("factors", "loglik", "dosage", "infprobs", "haplobase", "haplocount", "homozyg"):
        assert np.array_equal(got[k], whole[k], equal_nan=True), k
    assert np.array_equal(raw, raw_whole, equal_nan=True) and np.array_equal(lse, lse_whole, equal_nan=True)
    _check_accumulators(got, ped, oracle_ped(ped), desc)
    ctx.close()


def test_accumulator_invariants_at_full_chromosome_length(capi):
    """Size-independent properties of HOT LOOP 2's accumulators (cnF2freq.cpp:5416-5577 with moveinfprobs / movehaplos
    3577-3616) at BASELINE config 5's chromosome length (2 501 markers, 20 % missing genotypes), where the oracle would need
    hours: every path of an analysed child carries one allele value per side of the child and of each parent and passes
    through ONE grandparent of each parent, and a child's contributions are normalised to its own total, so with every
    descendant count 1
      * a child's infprobs sum to 1 on either side at every marker,
      * a parent's sum to its number of analysed children over both sides,
      * the two grandparents of a parent together collect that number as well,
      * haplocount is a whole number of children, 0 <= haplobase <= haplocount,
    whatever the batching (batches of 64 jobs here) and the grid (one block for half of the comparisons).  Verified on the
    oracle at small size first (this test's derivation)."""
    fams, kids, per = 40, 4, 10
    ped = synth.make_outbred3(fams, kids, 2500, 2, seed=23, missing=0.2)
    ctx = capi.Context(0)
    ctx.upload(ped)
    desc = np.ones(ped.n_rec, np.int32)
    ctx.set_batch_jobs(64)
    got = ctx.sweep_accumulate(desc, deterministic=True)
    ctx.set_grid_reserve(ONE_BLOCK)
    ctx.set_batch_jobs(0)
    one = ctx.sweep_accumulate(desc, deterministic=True)
    ctx.set_grid_reserve(0)
    for k in ("infprobs", "haplobase", "haplocount", "homozyg", "dosage", "loglik"):
        assert np.array_equal(got[k], one[k], equal_nan=True), "batches of 64 jobs and one block of one batch differ in " + k
    inf = got["infprobs"]                                     # [R][M][2][2]
    assert np.all(np.isfinite(inf)) and np.all(inf >= 0)
    base = np.arange(fams) * per
    kid = (base[:, None] + 6 + np.arange(kids)[None, :]).ravel()
    np.testing.assert_allclose(inf[kid].sum(axis=3), 1.0, rtol=0, atol=1e-9)
    for p in range(2):
        np.testing.assert_allclose(inf[base + 4 + p].sum(axis=(2, 3)), float(kids), rtol=0, atol=1e-8)
        pair = inf[base + 2 * p].sum(axis=(2, 3)) + inf[base + 2 * p + 1].sum(axis=(2, 3))
        np.testing.assert_allclose(pair, float(kids), rtol=0, atol=1e-8)
    hc, hb = got["haplocount"], got["haplobase"]
    assert np.array_equal(hc, np.round(hc)) and hc.min() >= 0 and hc.max() <= kids
    assert np.all(hb >= 0) and np.all(hb <= hc + 1e-12)
    assert (hc > 0).mean() > 0.2
    # the rows that come with it are distributions
    assert np.allclose(got["dosage"].sum(axis=2), 1.0, atol=1e-12)
    ctx.close()


def test_jobs_from_the_counter_equal_strided_jobs_on_unequal_chromosomes(capi):
    """The waves of a sweep launch take their jobs one at a time from the launch's counter (KernelParams::job_next); with
    CNF2_STATIC_JOBS wave w sweeps jobs w, w + waves, ...  A job's arithmetic does not depend on the wave that runs it: both
    forms must agree to the bit, on one block (every wave takes many jobs) and on the full grid, for tied and untied
    windows over chromosomes of very different lengths -- and with the oracle."""
    ped = synth.make_ail(6, 24, 3, 10, 3, seed=17, chrom_cm=60.0, missing=0.05)
    # chromosomes of 11, 11, 11 markers -> cut the map into 3 + 7 + 23
    cs = np.array([0, 3, 10, 33], np.int32)
    assert ped.n_markers == 33
    ped.chromstarts = cs
    pos = np.asarray(ped.pos, float).copy()
    for c in range(3):
        pos[cs[c]:cs[c + 1]] = np.arange(cs[c + 1] - cs[c]) * 4.0
    ped.pos = pos
    ctx = capi.Context(0)
    ctx.upload(ped)
    n = len(ped.dous)
    tied = sum(1 for j in range(n) if (ctx.window_info(j)["tie"] >= 0).any())
    assert tied >= 5 and n - tied >= 20, (tied, n)
    runs = {}
    for reserve in (0, ONE_BLOCK):
        ctx.set_grid_reserve(reserve)
        for static in (False, True):
            runs[reserve, static] = ctx.sweep(static_jobs=static)
    ctx.set_grid_reserve(0)
    ref = runs[0, True]
    for key, r in runs.items():
        for k in ("factors", "loglik", "dosage"):
            assert np.array_equal(r[k], ref[k]), (key, k)
    o = oracle_ped(ped)
    for c in range(3):
        first, last = int(cs[c]), int(cs[c + 1]) - 1
        want = o.sweep_batch(ped.dous, ped.gen[ped.dous], first=first, last=last, mode=2)
        np.testing.assert_allclose(ref["factors"][:, c], want["factors"], rtol=RTOL, atol=1e-8)
        np.testing.assert_allclose(ref["dosage"][:, first:last + 1], want["dosage"], rtol=1e-7, atol=1e-11)
    ctx.close()


def test_accumulate_and_turn_scan_jobs_from_the_counter_equal_strided_jobs(capi):
    """The same for the two batched consumers (their sweeps are the accumulate and the turn-scan instantiation of the kernel):
    CNF2_DETERMINISTIC accumulators (fixed order of addition) and the turn scan's values must not depend on which wave swept
    which job -- one block, batches of 5 jobs, chromosomes of unequal length, tied and untied windows."""
    ped = synth.make_ail(6, 24, 3, 10, 3, seed=19, chrom_cm=60.0, missing=0.05)
    cs = np.array([0, 5, 12, 33], np.int32)
    ped.chromstarts = cs
    pos = np.asarray(ped.pos, float).copy()
    for c in range(3):
        pos[cs[c]:cs[c + 1]] = np.arange(cs[c + 1] - cs[c]) * 3.0
    ped.pos = pos
    ctx = capi.Context(0)
    ctx.upload(ped)
    desc = ctx.descendants()
    ctx.set_batch_jobs(5)
    ctx.set_grid_reserve(ONE_BLOCK)
    acc = {s: ctx.sweep_accumulate(desc, deterministic=True, static_jobs=s) for s in (False, True)}
    turn = {s: ctx.sweep_turn_scan(full=False, lse=True, static_jobs=s)[1] for s in (False, True)}
    ctx.set_grid_reserve(0)
    ctx.set_batch_jobs(0)
    for k in ("factors", "loglik", "dosage", "infprobs", "haplobase", "haplocount", "homozyg"):
        assert np.array_equal(acc[False][k], acc[True][k], equal_nan=True), k
    assert np.array_equal(turn[False], turn[True], equal_nan=True)
    o = oracle_ped(ped)
    _check_accumulators(acc[False], ped, o, desc)
    ctx.close()


def test_tied_windows_on_chromosomes_of_one_two_eight_and_nine_markers(capi):
    """The tied instantiation runs a tile once per PAIR of tie combinations and restores its state per pass: chromosomes of
    1, 2, 8, 9 and 13 markers put the tile boundary, the even top marker and the one-marker job under it -- sweep rows and
    likelihoods and the accumulators of tied and untied windows against the oracle."""
    ped = synth.make_ail(6, 24, 3, 10, 3, seed=23, chrom_cm=60.0, missing=0.05)
    cs = np.array([0, 1, 3, 11, 20, 33], np.int32)
    ped.chromstarts = cs
    pos = np.asarray(ped.pos, float).copy()
    for c in range(5):
        pos[cs[c]:cs[c + 1]] = np.arange(cs[c + 1] - cs[c]) * 5.0
    ped.pos = pos
    ctx = capi.Context(0)
    ctx.upload(ped)
    n = len(ped.dous)
    groups = [int((ctx.window_info(j)["tie"].max()) + 1) for j in range(n)]
    assert sum(1 for g in groups if g == 1) >= 3 and sum(1 for g in groups if g >= 2) >= 3, groups   # one pass and several
    got = ctx.sweep()
    o = oracle_ped(ped)
    for c in range(5):
        first, last = int(cs[c]), int(cs[c + 1]) - 1
        want = o.sweep_batch(ped.dous, ped.gen[ped.dous], first=first, last=last, mode=2)
        np.testing.assert_allclose(got["factors"][:, c], want["factors"], rtol=RTOL, atol=1e-8)
        np.testing.assert_allclose(got["dosage"][:, first:last + 1], want["dosage"], rtol=1e-7, atol=1e-11)
    desc = ctx.descendants()
    acc = ctx.sweep_accumulate(desc, deterministic=True)
    assert np.array_equal(acc["dosage"], got["dosage"]) and np.array_equal(acc["factors"], got["factors"])
    _check_accumulators(acc, ped, o, desc)
    ctx.close()


def test_more_jobs_than_resident_waves_on_the_full_grid(capi):
    """4 800 jobs on the full grid (2 048 resident waves: every wave takes a second and a third job while others are still
    on their first): jobs from the counter and strided jobs agree to the bit, a sample of individuals equals the oracle."""
    ped = synth.make_f2(600, 40, 8, seed=77, chrom_cm=50.0)
    ctx = capi.Context(0)
    ctx.upload(ped)
    a = ctx.sweep()
    b = ctx.sweep(static_jobs=True)
    for k in ("factors", "loglik", "dosage"):
        assert np.array_equal(a[k], b[k]), k
    assert np.isfinite(a["loglik"]).all() and (np.abs(a["dosage"].sum(axis=2) - 1.0) < 1e-9).all()
    o = oracle_ped(ped)
    pick = np.arange(0, 600, 37)
    for c in (0, 7):
        first, last = int(ped.chromstarts[c]), int(ped.chromstarts[c + 1]) - 1
        want = o.sweep_batch(ped.dous[pick], ped.gen[ped.dous[pick]], first=first, last=last, mode=2)
        np.testing.assert_allclose(a["factors"][pick, c], want["factors"], rtol=RTOL, atol=1e-8)
        np.testing.assert_allclose(a["dosage"][pick][:, first:last + 1], want["dosage"], rtol=1e-7, atol=1e-11)
    ctx.close()


def test_accumulate_without_rows_gives_the_same_accumulators(capi):
    """A call that passes no dosage pointer (an iteration that prints no rows) sweeps in the instantiation that forms none,
    windows with tie groups included (their posterior weights do not see the tie rule; their accumulators, which do, keep
    their own pass).  Likelihoods equal the with-rows call to the bit; so do the CNF2_DETERMINISTIC accumulators wherever
    the same sweep kernel is behind them (all windows with the tie rule off, the untied windows with it on); the tied
    windows' weights come from another instantiation of the kernel (its multiply-adds are fused differently) and agree to
    rounding.  One block and the full grid, chromosomes of unequal length; the with-rows call against the oracle."""
    ped = synth.make_ail(6, 24, 3, 10, 3, seed=29, chrom_cm=60.0, missing=0.05)
    cs = np.array([0, 4, 13, 33], np.int32)
    ped.chromstarts = cs
    pos = np.asarray(ped.pos, float).copy()
    for c in range(3):
        pos[cs[c]:cs[c + 1]] = np.arange(cs[c + 1] - cs[c]) * 3.0
    ped.pos = pos
    ctx = capi.Context(0)
    ctx.upload(ped)
    n = len(ped.dous)
    tied = np.array([(ctx.window_info(j)["tie"] >= 0).any() for j in range(n)])
    assert tied.sum() >= 5 and (~tied).sum() >= 20, (tied.sum(), n)
    desc = ctx.descendants()
    keys = ("infprobs", "haplobase", "haplocount", "homozyg")
    for ties in (False, True):
        with_rows = ctx.sweep_accumulate(desc, deterministic=True, ties=ties)
        for reserve in (0, ONE_BLOCK):
            ctx.set_grid_reserve(reserve)
            got = ctx.sweep_accumulate(desc, deterministic=True, ties=ties, rows=False)
            ctx.set_grid_reserve(0)
            assert not got["dosage"].any()
            for k in ("factors", "loglik"):
                assert np.array_equal(got[k], with_rows[k]), (ties, reserve, k)
            assert np.array_equal(got["homozyg"][~tied], with_rows["homozyg"][~tied], equal_nan=True), (ties, reserve)
            for k in keys:
                if not ties:
                    assert np.array_equal(got[k], with_rows[k], equal_nan=True), (reserve, k)
                else:
                    np.testing.assert_allclose(got[k], with_rows[k], rtol=1e-13, atol=1e-300, equal_nan=True, err_msg=k)
        if ties:
            _check_accumulators(with_rows, ped, oracle_ped(ped), desc)
            atomic = ctx.sweep_accumulate(desc, rows=False)             # the default (atomic) accumulators: equal to rounding
            for k in keys:
                np.testing.assert_allclose(atomic[k], with_rows[k], rtol=1e-12, atol=1e-14, equal_nan=True, err_msg=k)
    ctx.close()
