"""BASELINE.json's configurations at their full sizes on one GPU (configs 2, 3 and 5: the single-GPU ones), where the
oracle cannot follow: the size-independent properties bench.py checks on its timed step (every row a distribution, likelihoods
finite and negative, the log-sum-exp identity, every output rewritten after a NaN poisoning), jobs taken from the launch's
counter against strided jobs on a 2 000-job subset (bit equality: catches a job counter, batch buffer or 32-bit index that
only breaks at size), and the oracle on three individuals x one full chromosome each.  Each test is a few GPU-seconds
plus the generation of its input."""
import time
from types import SimpleNamespace

import numpy as np
import pytest

from cnf2freq_amd import synth
from conftest import oracle_ped

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    import torch  # noqa: F401  (one HIP runtime per process: before libcnf2hip.so)
    from cnf2freq_amd import capi as c
    c.load()
    return c


def _device_outputs(n, chroms, M):
    import torch
    dev = torch.device("cuda", 0)
    return (torch.empty((n, chroms, 8), dtype=torch.float64, device=dev), torch.empty((n, chroms), dtype=torch.float64, device=dev),
            torch.empty((n, M, 3), dtype=torch.float64, device=dev))


def _property_checks(capi, ctx, n, factors, loglik, dosage):
    """bench.py's five checks on the outputs of a sweep already in the tensors (the sweep is run once more here, after the
    outputs have been overwritten with NaN)."""
    import torch
    ll_first = loglik.clone()
    for t in (loglik, dosage, factors):
        t.fill_(float("nan"))
    ctx.sweep_device(0, n, factors.data_ptr(), loglik.data_ptr(), dosage.data_ptr())
    ctx.sync()
    assert torch.equal(loglik, ll_first), "two sweeps of the same state differ"
    assert not torch.isnan(dosage).any().item() and not torch.isnan(factors).any().item(), "an output was not rewritten: a job was left out"
    rs = dosage.sum(dim=2)
    assert ((rs - 1.0).abs() < 1e-9).all().item(), "a row is not a distribution"
    assert (dosage >= 0).all().item()
    assert (torch.isfinite(loglik) & (loglik < 0)).all().item()
    fmx = factors.max(dim=2, keepdim=True).values
    lse = fmx.squeeze(2) + torch.log(torch.exp(factors - fmx).sum(dim=2))
    assert ((lse - loglik).abs() < 1e-9 * loglik.abs().clamp(min=1.0)).all().item(), "log-sum-exp identity"


def _counter_against_strided(capi, ctx, k0, k, chroms, M, factors, loglik, dosage):
    """individuals [k0, k0 + k) swept on their own with strided jobs (CNF2_STATIC_JOBS): the bits of the full launch, whose
    waves took their jobs from the counter"""
    import torch
    f, l, d = _device_outputs(k, chroms, M)
    ctx.sweep_device(k0, k0 + k, f.data_ptr(), l.data_ptr(), d.data_ptr(), capi.STATIC_JOBS)
    ctx.sync()
    assert torch.equal(l, loglik[k0:k0 + k]) and torch.equal(f, factors[k0:k0 + k]) and torch.equal(d, dosage[k0:k0 + k])


def _chromosome_slice(ped, c):
    first, end = int(ped.chromstarts[c]), int(ped.chromstarts[c + 1])
    sub = synth.Pedigree(ped.names, ped.par, ped.gen, ped.empty, ped.row_of, np.ascontiguousarray(ped.allele[:, first:end]),
                         np.ascontiguousarray(ped.sure[:, first:end]), np.ascontiguousarray(ped.hw[:, first:end]),
                         ped.pos[first:end].copy(), np.array([0, end - first], np.int32), ped.dous)
    sub.founder_flags()
    return sub, first, end


def test_config2_f2_10000_individuals_x_50020_markers(capi):
    """BASELINE config 2 (the headline's workload): synthetic F2, 10 000 individuals x 20 chromosomes x 2 501 markers,
    generated on the GPU as bench.py generates it."""
    import torch
    import bench
    args = SimpleNamespace(inds=10000, chroms=20, snps_per_chrom=2500, seed=12345)
    dev = torch.device("cuda", 0)
    pos, starts = synth.make_map(args.chroms, args.snps_per_chrom)
    M, n = len(pos), args.inds
    ctx = capi.Context(0)
    ctx.upload_map(pos, starts)
    sample = bench.generate_on_gpu(ctx, args, 0, dev, pos, starts)
    par, gen, empty, row_of, dous = synth.f2_pedigree_tables(n)
    ctx.upload_pedigree(par, empty, gen, row_of, dous)
    factors, loglik, dosage = _device_outputs(n, args.chroms, M)
    t0 = time.perf_counter()
    ctx.sweep_device(0, n, factors.data_ptr(), loglik.data_ptr(), dosage.data_ptr())
    ctx.sync()
    first_sweep_s = time.perf_counter() - t0
    _property_checks(capi, ctx, n, factors, loglik, dosage)
    _counter_against_strided(capi, ctx, 4321, 100, args.chroms, M, factors, loglik, dosage)      # 100 individuals x 20 chromosomes = 2 000 jobs
    # the oracle on 3 individuals x chromosome 1 (the generator's first individuals come back as the host sample)
    k, mc = 3, int(starts[1] - starts[0])
    par, gen, empty, row_of, dous = synth.f2_pedigree_tables(k)
    allele = np.zeros((3 + k, mc, 2), np.uint8)
    allele[1], allele[2] = 1, 2
    allele[3:, :, 0] = sample[:k, :mc] & 15
    allele[3:, :, 1] = sample[:k, :mc] >> 4
    small = synth.Pedigree(["r%d" % i for i in range(len(par))], par, gen, empty, row_of, allele, np.where(allele != 0, 0.02, 0.0),
                           np.full((3 + k, mc), 0.5), pos[:mc].copy(), np.array([0, mc], np.int32), dous)
    small.founder_flags()
    want = oracle_ped(small).sweep_batch(small.dous, small.gen[small.dous], mode=2)
    np.testing.assert_allclose(dosage[:k, :mc].cpu().numpy(), want["dosage"], rtol=1e-6, atol=1e-10)
    np.testing.assert_allclose(loglik[:k, 0].cpu().numpy(), want["factor"], rtol=1e-10)
    print("config 2 at size: first sweep (with the spill's allocation) %.2f s" % first_sweep_s)
    ctx.close()


def test_config3_advanced_intercross_5000_individuals_x_20008_markers(capi):
    """BASELINE config 3: 2 inbred founders, 64 genotyped F1, 8 random-mating generations x 625 analysed individuals,
    8 chromosomes x 2 501 markers; windows with tie groups beside untied ones."""
    ped = synth.make_ail(64, 625, 8, 2500, 8, seed=3)
    n, M, chroms = len(ped.dous), ped.n_markers, len(ped.chromstarts) - 1
    assert n == 5000 and M == 8 * 2501
    ctx = capi.Context(0)
    ctx.upload(ped)
    tied = [j for j in range(n) if (ctx.window_info(j)["tie"] >= 0).any()]
    assert 0 < len(tied) < n
    factors, loglik, dosage = _device_outputs(n, chroms, M)
    ctx.sweep_device(0, n, factors.data_ptr(), loglik.data_ptr(), dosage.data_ptr())
    ctx.sync()
    _property_checks(capi, ctx, n, factors, loglik, dosage)
    _counter_against_strided(capi, ctx, max(0, tied[0] - 100), 250, chroms, M, factors, loglik, dosage)     # 250 x 8 = 2 000 jobs, a tied window among them
    # the oracle on 3 individuals x one full chromosome each, one of them a window with a tie group
    for j, c in ((0, 0), (tied[0], 1), (n - 1, 7)):
        sub, first, end = _chromosome_slice(ped, c)
        ind = int(ped.dous[j])
        want = oracle_ped(sub).sweep_batch(np.array([ind], np.int32), ped.gen[[ind]], mode=2)
        np.testing.assert_allclose(dosage[j, first:end].cpu().numpy(), want["dosage"][0], rtol=1e-6, atol=1e-10)
        np.testing.assert_allclose(float(loglik[j, c].item()), want["factor"][0], rtol=1e-10)
    ctx.close()


def test_config5_shape_2500_families_x_10004_markers_sweep_accumulate_update(capi):
    """BASELINE config 5's pedigree at its size (2 500 three-generation families: 25 000 individuals, 10 000 analysed,
    4 x 2 501 markers, 20 % of the genotypes missing): the sweep's properties, then -- through the engine, as a run does it --
    postmarkerdata and one haplotyping iteration (cnf2_sweep_accumulate over all chromosomes, the update passes of the four
    chromosomes), whose state must be a state: certainties and weights in [0, 1], the alleles called, the step size moved by the rule."""
    import torch
    from cnf2freq_amd import host
    ped = synth.make_outbred3(2500, 4, 2500, 4, seed=2, missing=0.2)
    n, M, chroms = len(ped.dous), ped.n_markers, len(ped.chromstarts) - 1
    assert n == 10000 and M == 4 * 2501 and ped.n_rec == 25000
    ctx = capi.Context(0)
    ctx.upload(ped)
    factors, loglik, dosage = _device_outputs(n, chroms, M)
    ctx.sweep_device(0, n, factors.data_ptr(), loglik.data_ptr(), dosage.data_ptr())
    ctx.sync()
    _property_checks(capi, ctx, n, factors, loglik, dosage)
    _counter_against_strided(capi, ctx, 7000, 500, chroms, M, factors, loglik, dosage)       # 500 x 4 = 2 000 jobs
    for j, c in ((0, 0), (n // 2, 2), (n - 1, 3)):
        sub, first, end = _chromosome_slice(ped, c)
        ind = int(ped.dous[j])
        want = oracle_ped(sub).sweep_batch(np.array([ind], np.int32), ped.gen[[ind]], mode=2)
        np.testing.assert_allclose(dosage[j, first:end].cpu().numpy(), want["dosage"][0], rtol=1e-6, atol=1e-10)
        np.testing.assert_allclose(float(loglik[j, c].item()), want["factor"][0], rtol=1e-10)
    ctx.close()
    del factors, loglik, dosage
    torch.cuda.empty_cache()
    # one iteration of a run
    run = host.Run(ped)
    run.postmarkerdata()
    before = run.state()
    run.iteration(None)
    st = run.state()
    run.close()
    assert st["hits"] > 0
    assert np.isfinite(st["sure"]).all() and (st["sure"] >= 0).all() and (st["sure"] <= 1).all()
    assert np.isfinite(st["hw"]).all() and (st["hw"] >= 0).all() and (st["hw"] <= 1).all()
    assert np.isin(st["allele"], (0, 1, 2)).all()
    # cnF2freq.cpp:6373-6392 after each of the four passes: x 0.997, / 1.1 when the pass had more capped moves than the two before
    # it, x 1.21 when it had few
    ratio = st["scalefactor"] / before["scalefactor"]
    assert any(abs(ratio - 0.997 ** 4 * 1.21 ** a / 1.1 ** b) < 1e-9 for a in range(5) for b in range(5)), ratio
    moved = np.abs(st["hw"] - before["hw"]) > 0
    assert moved.mean() > 0.05, "the haplotype weights did not move"
