#!/usr/bin/env python3
"""bench.py -- headline benchmark of the forward-backward sweep (BASELINE.json metric:
individual x marker fwd-bwd steps/sec on the synthetic 10k-individual x 50k-SNP F2).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full sweep (cnf2_sweep through the C ABI: forward + backward over all 8
shift modes, likelihoods and the per-locus dosage rows) over the rank's individuals, inputs
resident in HBM.  Individuals shard across ranks with no data-path collective (weak scaling:
every rank owns --inds individuals); after the sweep the posteriors are gathered on rank 0
with one RCCL gather inside the timed region.  Rank 0 prints ONE JSON line.

`--gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): this process starts the N
ranks itself -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py <same
arguments>` as a CHILD process, before anything here has imported torch or loaded the HIP
library (a process that has touched the GPU must not start or become another program's
launcher) -- relays rank 0's JSON line on stdout and exits with the child's code.  The
partition the ranks take is the reference's own (cnF2freq.cpp:5297-5299).
"""
import os

os.environ.setdefault("OMP_STACKSIZE", "128M")  # demo.sh:36 (reference-extract CPU baseline)

import argparse
import json
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def host_cpu_share():
    """CPUs this process may really use: affinity mask and cgroup quota, not the host's count."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    return n


def requested_gpus(argv):
    """--gpus N of the command line, read without argparse's exit paths (the full parser runs in the ranks)."""
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            return int(argv[i + 1])
        if a.startswith("--gpus="):
            return int(a.split("=", 1)[1])
    return 1


def launch_ranks(argv):
    """Start `--gpus N` ranks as a child process when no launcher did (see the module docstring).  Returns None when
    this process is itself a rank (or N = 1), else the exit code to leave with.  Nothing that initialises HIP may
    have been imported when this runs: CNF2_BENCH_PARENT_TRACE=<file> records sys.modules at the moment of the
    spawn for the test that checks it."""
    n = requested_gpus(argv)
    if n <= 1 or "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return None
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cpu_share() // n)))      # the ranks share this process's CPU quota
    # --standalone: torchrun hosts the rendezvous itself on a port IT finds free (a port probed here could be taken by another
    # process before the child binds it: several benches started at the same moment)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(n), os.path.abspath(__file__)] + list(argv)
    trace = os.environ.get("CNF2_BENCH_PARENT_TRACE")
    if trace:
        with open(trace, "w") as f:
            json.dump({"modules": sorted(sys.modules), "cmd": cmd}, f)
    print("bench.py: starting %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = 0
    for line in child.stdout:                 # rank 0's JSON line goes to stdout, anything else a rank printed to stderr
        if line.startswith("{") and '"metric"' in line:
            sys.stdout.write(line)
            sys.stdout.flush()
            lines += 1
        else:
            sys.stderr.write(line)
    rc = child.wait()
    if rc == 0 and lines != 1:
        print("bench.py: the ranks printed %d result lines, expected 1" % lines, file=sys.stderr)
        rc = 1
    return rc


if __name__ == "__main__":
    _rc = launch_ranks(sys.argv[1:])
    if _rc is not None:
        sys.exit(_rc)

import numpy as np
import torch
import torch.distributed as dist

B_UNIT = 8248.0        # algorithmic bytes per individual x marker (S_act = 8, f64)
HBM_PEAK = 8.0e12      # MI355X_MICROARCH.md: 8 TB/s


def kernel_source_sha():
    """Identity of the kernel sources the library was built from: PMC traffic measured for another
    version of the kernels must not be reported next to this one's timing."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "cnf2freq_amd", "csrc")
    for f in ("cnf2_kernels.hip", "cnf2_emtab.h", "cnf2_emission.h", "cnf2_lane.h", "cnf2_device.h"):
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--inds", type=int, default=10000, help="analysed F2 individuals per GPU")
    ap.add_argument("--chroms", type=int, default=20)
    ap.add_argument("--snps-per-chrom", type=int, default=2500)
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = this process's CPU share)")
    ap.add_argument("--full-spill", action="store_true", help="A/B: store alpha-minus at every marker (CNF2_FULL_SPILL)")
    ap.add_argument("--extra-flags", type=int, default=0,
                    help="tuning aid: OR into the sweep flags (2 = CNF2_NO_DOSAGE, forward pass only); not the metric")
    ap.add_argument("--no-merge-probe", action="store_true",
                    help="skip the extra CNF2_MERGE_MODES sweeps reported next to the headline (N = 1 only)")
    ap.add_argument("--no-iteration-probe", action="store_true",
                    help="skip the scaled config-5 haplotyping iterations reported next to the headline (rank 0, outside the timed region)")
    ap.add_argument("--no-gather", action="store_true", help="leave the posteriors on their GPUs")
    ap.add_argument("--no-overlap", action="store_true",
                    help="wait for each step's gather before the next sweep (default: the gather of step k runs "
                         "beside the sweep of step k+1, double-buffered)")
    ap.add_argument("--reserve-blocks", type=int, default=16,
                    help="workgroup slots the sweep leaves free for the RCCL kernels when gathers overlap")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the multi-rank control flow on one GPU (posteriors staged through the host)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--traffic-file", default=os.path.join(ROOT, "profiles", "hbm_traffic.json"))
    ap.add_argument("--gather-tile-markers", type=int, default=0,
                    help="markers per tile of the streaming gather to rank 0 (0 = one chromosome); rank 0 holds "
                         "2 tiles x world_size, never the whole posterior")
    ap.add_argument("--workload", default="f2", choices=["f2", "ail", "outbred"],
                    help="f2 = BASELINE configs 2/4 (headline); ail = config 3 (advanced intercross, 2 founders + 64 F1 + "
                         "8 generations, tied windows); outbred = config 5's shape (3-generation outbred, 20 %% missing)")
    ap.add_argument("--iterations", type=int, default=0,
                    help="with --workload outbred: time that many HAPLOTYPING ITERATIONS (BASELINE config 5: sweep + HOT LOOP 2 "
                         "accumulators on the rank's block of individuals, one all-reduce of the accumulators, the update passes) "
                         "instead of plain sweeps; --inds = analysed individuals in all (not per GPU: iterations do not shard, "
                         "individuals do)")
    return ap.parse_args()


def generate_on_gpu(ctx, args, rank, device, pos, starts):
    """Synthetic F2 genotypes generated on the GPU (torch) and handed to the library as
    device rows: founders A=(1,1), B=(2,2); every F2 = two F1 gametes with Haldane crossovers;
    unphased dosage, sure = 0.02, hw = 0.5."""
    M = len(pos)
    n = args.inds
    ctx.alloc_blank_rows(3 + n)
    d = np.diff(pos, prepend=pos[0])
    rho = 0.5 * (1.0 - np.exp(-0.02 * np.maximum(d, 0.0)))
    rho[np.asarray(starts[:-1])] = 0.5
    rho_t = torch.tensor(rho, dtype=torch.float32, device=device)
    gen = torch.Generator(device=device)
    gen.manual_seed(args.seed + 7919 * rank)

    def put(row0, packed):
        k = packed.shape[0]
        sure = torch.full((k, M, 2), 0.02, dtype=torch.float64, device=device)
        hw = torch.full((k, M), 0.5, dtype=torch.float64, device=device)
        torch.cuda.synchronize()
        ctx.update_rows_device(row0, k, packed.data_ptr(), sure.data_ptr(), hw.data_ptr())

    founders = torch.empty((2, M), dtype=torch.uint8, device=device)
    founders[0] = 1 | (1 << 4)
    founders[1] = 2 | (2 << 4)
    put(1, founders)
    chunk = 500
    keep = None
    for i0 in range(0, n, chunk):
        k = min(chunk, n - i0)
        strands = []
        for _ in range(2):
            rec = (torch.rand((k, M), generator=gen, device=device) < rho_t).to(torch.int32)
            strands.append(torch.cumsum(rec, dim=1) & 1)
        dosage = strands[0] + strands[1]
        a0 = torch.where(dosage == 2, 2, 1)
        a1 = torch.where(dosage == 0, 1, 2)
        packed = (a0 | (a1 << 4)).to(torch.uint8).contiguous()
        put(3 + i0, packed)
        if i0 == 0:
            keep = packed.cpu().numpy()   # sample for the CPU baseline leg
    return keep


def generate_outbred_on_gpu(n_fam, kids, snps_per_chrom, n_chrom, seed, missing, device):
    """cnf2freq_amd.synth.make_outbred3's pedigree (BASELINE config 5's shape: per family 4 genotyped grandparents, 2 genotyped
    parents, `kids` analysed children; founder allele-2 frequency U(0.1, 0.9) per SNP; Haldane crossovers; unphased genotypes,
    `missing` of them withheld) drawn with torch on the GPU -- the numpy generator needs ~4 ms per record x 1 000 markers --
    and handed back as host arrays in a synth.Pedigree (with .truth, the generator's allele-2 dosages)."""
    from cnf2freq_amd import synth
    pos, starts = synth.make_map(n_chrom, snps_per_chrom)
    M = len(pos)
    per = 6 + kids
    R = n_fam * per
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    idx = torch.arange(R, device=device)
    fam, k = idx // per, idx % per
    base = fam * per
    isp, isk = (k == 4) | (k == 5), k >= 6
    par = torch.full((R, 2), -1, dtype=torch.int64, device=device)
    par[isp, 0] = (base + 2 * (k - 4))[isp]
    par[isp, 1] = (base + 2 * (k - 4) + 1)[isp]
    par[isk, 0] = (base + 4)[isk]
    par[isk, 1] = (base + 5)[isk]
    d = np.diff(pos, prepend=pos[0])
    rho = 0.5 * (1.0 - np.exp(-0.02 * np.maximum(d, 0.0)))
    rho[np.asarray(starts[:-1])] = 0.5
    rho_t = torch.tensor(rho, dtype=torch.float32, device=device)
    freq = 0.1 + 0.8 * torch.rand(M, generator=g, device=device)
    hap = torch.zeros((R, M, 2), dtype=torch.uint8, device=device)
    gp = torch.nonzero(k < 4).squeeze(1)
    hap[gp] = (1 + (torch.rand((len(gp), M, 2), generator=g, device=device) < freq[None, :, None])).to(torch.uint8)
    for rows in (torch.nonzero(isp).squeeze(1), torch.nonzero(isk).squeeze(1)):
        for side in range(2):
            rec = (torch.rand((len(rows), M), generator=g, device=device) < rho_t).to(torch.int32)
            strand = (torch.cumsum(rec, dim=1) & 1).to(torch.int64)
            src = hap[par[rows, side]]                                   # [n][M][2]
            hap[rows, :, side] = torch.gather(src, 2, strand.unsqueeze(2)).squeeze(2)
    dosage = (hap == 2).sum(dim=2).to(torch.uint8)
    allele = torch.zeros((R + 1, M, 2), dtype=torch.uint8, device=device)
    allele[1:, :, 0] = torch.where(dosage == 2, 2, 1).to(torch.uint8)
    allele[1:, :, 1] = torch.where(dosage == 0, 1, 2).to(torch.uint8)
    miss = torch.rand((R, M), generator=g, device=device) < missing
    allele[1:][miss] = 0
    sure = torch.where(allele != 0, 0.02, 0.0).to(torch.float64)
    fam_l, k_l = fam.tolist(), k.tolist()
    names = [("G%d_%d" % (f, j)) if j < 4 else ("P%d_%d" % (f, j - 4)) if j < 6 else ("K%d_%d" % (f, j - 6)) for f, j in zip(fam_l, k_l)]
    gen = torch.where(isk, 2, torch.where(isp, 1, 0)).to(torch.int32)
    ped = synth.Pedigree(names, par.to(torch.int32).cpu().numpy(), gen.cpu().numpy(), np.zeros(R, np.uint8),
                         np.arange(1, R + 1, dtype=np.int32), allele.cpu().numpy(), sure.cpu().numpy(), np.full((R + 1, M), 0.5),
                         pos, starts, torch.nonzero(isk).squeeze(1).to(torch.int32).cpu().numpy())
    ped.founder_flags()
    ped.truth = dosage.cpu().numpy()
    return ped


def iteration_probe(local, device, fams=500, snps_per_chrom=2500, chroms=4, warmup=2, timed=3, update_flags=None, with_stats=True):
    """BASELINE config 5's unit of work at a fifth of its size, beside the headline (outside every timed region): haplotyping
    iterations (sweep + HOT LOOP 2 accumulators, update passes, step-size control) of a 3-generation outbred pedigree with 20 %
    missing genotypes -- `fams` families x 4 analysed children x `chroms` x `snps_per_chrom` markers -- through libcnf2host.so,
    the accumulate sweep in 4 batches as at full size.  Reports the iteration's rate and where its wall time goes, and the
    kernel time of sweep + accumulators against the plain sweep of the same windows (rows f1 / f4 of SURVEY.md section 8)."""
    import ctypes as C
    from cnf2freq_amd import capi, host
    t_all = time.perf_counter()
    ped = generate_outbred_on_gpu(fams, 4, snps_per_chrom, chroms, 2, 0.2, device)
    n, M, R = len(ped.dous), ped.n_markers, ped.n_rec
    sys.stdout.flush()
    saved_stdout = os.dup(1)          # the engine prints the reference's progress lines on stdout: to stderr while it runs
    os.dup2(2, 1)
    try:
        run = host.Run(ped, device=local)
        if update_flags is not None:
            run.set_update_flags(update_flags)
        L = capi.load()
        ctx = run.context()
        L.cnf2_set_batch_jobs(ctx, max(1, n * chroms // 4))
        t0 = time.perf_counter()
        run.postmarkerdata()
        t_pm = time.perf_counter() - t0
        run.reserve()
        for _ in range(warmup):
            run.iteration()
        laps, acc_ms, wall = [], [], []
        ms = np.zeros(4, np.float32)
        for _ in range(timed):
            t0 = time.perf_counter()
            run.iteration()
            wall.append(time.perf_counter() - t0)
            laps.append(run.timing())
            L.cnf2_last_kernel_ms(ctx, ms.ctypes.data_as(C.c_void_p), 4)
            acc_ms.append(float(ms[0]))
        # the plain sweep of the same windows on the same context (device outputs)
        f = torch.empty((n, chroms, 8), dtype=torch.float64, device=device)
        ll = torch.empty((n, chroms), dtype=torch.float64, device=device)
        dos = torch.empty((n, M, 3), dtype=torch.float64, device=device)
        sweep_ms = []
        for _ in range(2):
            rc = L.cnf2_sweep(ctx, 0, n, C.c_void_p(f.data_ptr()), C.c_void_p(ll.data_ptr()), C.c_void_p(dos.data_ptr()), capi.OUT_DEVICE)
            if rc != 0:
                raise RuntimeError(L.cnf2_last_error(ctx).decode())
            L.cnf2_sync(ctx)
            L.cnf2_last_kernel_ms(ctx, ms.ctypes.data_as(C.c_void_p), 4)
            sweep_ms.append(float(ms[0]))
        # one more iteration with the update kernels' statistics on (a few atomics per wavefront; not among the timed ones):
        # flows and gradient evaluations of an iteration, for the update pass's own roofline below
        upd = None
        try:
            if not with_stats:
                raise KeyError("no statistics asked for")
            os.environ["CNF2_UPDATE_STATS"] = "1"
            run.iteration()
            s16, s8 = np.zeros(16, np.uint64), np.zeros(8, np.uint64)
            if L.cnf2_update_stats(ctx, s16.ctypes.data_as(C.c_void_p)) == 0 and L.cnf2_update_stats_guided(ctx, s8.ctypes.data_as(C.c_void_p)) == 0:
                upd = (s16.astype(float), s8.astype(float))
        except KeyError:
            pass
        finally:
            os.environ.pop("CNF2_UPDATE_STATS", None)
        st = run.state()
        run.close()
    finally:
        sys.stdout.flush()
        C.CDLL(None).fflush(None)
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    mean = lambda k: float(np.mean([l[k] for l in laps]))
    it_s = float(np.mean(wall))
    update_roofline = None
    if upd is not None:
        s16, s8 = upd
        # gradient evaluations of one iteration: the scouts' (one per evaluation), the guided kernels' (16 per literal point, the
        # seeds' single ones counted), the step-per-round tail's (a step's midpoint + 15 per quadrature)
        flows = s16[0] + s16[4]
        evals = s16[1] + s16[5] + s8[2] + s8[6] + (s16[8] + 15.0 * s16[10]) + (s16[12] + 15.0 * s16[14])
        points = s8[0] + s8[4]
        upd_s = mean("update_s")
        # a gradient evaluation: ~62 double-precision vector instructions (logit: two divisions by reciprocal + Newton, an
        # 11-term series; the rational data term; one more division), of which ~45 are fused multiply-adds: ~100 flops
        FLOPS_PER_EVAL, FP64_VECTOR_PEAK = 100.0, 78.6e12
        update_roofline = {
            "bound": "fp64_valu", "flows_per_iteration": flows, "certainty_flows": s16[0], "haploweight_flows": s16[4],
            "gradient_evaluations_per_iteration": evals, "evaluations_per_flow": evals / max(flows, 1.0),
            "literal_points_per_iteration": points, "evaluations_per_s": evals / upd_s,
            "flops_per_evaluation": FLOPS_PER_EVAL, "achieved_tflops": evals * FLOPS_PER_EVAL / upd_s / 1e12,
            "peak_tflops": FP64_VECTOR_PEAK / 1e12, "frac": evals * FLOPS_PER_EVAL / upd_s / FP64_VECTOR_PEAK,
            "note": "evaluations the kernels really made (the literal bisection of the reference would make ~3x as many: "
                    "cnf2_update.h); update_s is the wall time of the four passes' calls, launches and syncs included",
        }
    return {"workload": "3-generation outbred pedigree, %d families = %d individuals, %d analysed, 20%% of genotypes missing, "
                        "%d chromosomes x %d SNPs (+1 dummy each): BASELINE config 5 at %d/2500 of its families"
                        % (fams, R, n, chroms, snps_per_chrom, fams),
            "iterations_timed": timed, "warmup": warmup, "iteration_s": it_s, "units_per_s": float(n) * M / it_s,
            "sweep_accumulate_s": mean("sweep_accumulate_s"), "update_s": mean("update_s"), "host_s": mean("host_s"),
            "sweep_accumulate_kernel_ms": float(np.mean(acc_ms)), "plain_sweep_kernel_ms": float(min(sweep_ms)),
            "accumulate_over_sweep": float(np.mean(acc_ms)) / float(min(sweep_ms)),
            "update_roofline": update_roofline,
            "postmarkerdata_s": t_pm, "scalefactor": st["scalefactor"], "last_hits": st["hits"],
            "probe_wall_s": time.perf_counter() - t_all}


def cpu_baseline(sample_packed, pos, starts, args):
    """CPU path timed beside the GPU on a bounded sample of the same workload: the first
    individuals of rank 0 on chromosome 1.  kind "reference" = the reference's own code
    (oracle/_ref, HOT LOOP 1 as in cnF2freq.cpp:5294-5403); kind "port" = the C restatement
    (oracle/, sweep + closed-form dosage rows).  The oracle is the thing timed here, never
    the thing shipped."""
    from cnf2freq_amd import synth
    first, last = int(starts[0]), int(starts[1]) - 1
    mc = last - first + 1
    threads = args.cpu_threads or host_cpu_share()

    def build_ped(k):
        par, gen, empty, row_of, dous = synth.f2_pedigree_tables(k)
        allele = np.zeros((3 + k, mc, 2), np.uint8)
        allele[1], allele[2] = 1, 2
        pk = sample_packed[:k, first:last + 1]
        allele[3:, :, 0] = pk & 15
        allele[3:, :, 1] = pk >> 4
        sure = np.where(allele != 0, 0.02, 0.0)
        hw = np.full((3 + k, mc), 0.5)
        ped = synth.Pedigree(["r%d" % i for i in range(len(par))], par, gen, empty, row_of, allele, sure, hw,
                             pos[first:last + 1].copy(), np.array([0, mc], np.int32), dous)
        ped.founder_flags()
        return ped

    kind = "port"
    try:
        from oracle.ref_extract import pyref
        if pyref.available(ieee=False):
            kind = "reference"
    except Exception:
        kind = "port"

    def run(k):
        ped = build_ped(k)
        if kind == "reference":
            R = pyref.RefPed(ped, ieee=False)
            R.sweep_batch(ped.dous[:min(k, threads)], threads=threads)   # warm the thread-private stores
            t0 = time.perf_counter()
            _, used = R.sweep_batch(ped.dous, threads=threads)
            return time.perf_counter() - t0, used
        from oracle.pyoracle import OraclePed
        a, s, h = ped.dense()
        o = OraclePed(a, s, h, ped.par, ped.empty, ped.pos)
        t0 = time.perf_counter()
        r = o.sweep_batch(ped.dous, ped.gen[ped.dous], mode=2, n_threads=threads)
        return time.perf_counter() - t0, r["threads"]

    k_max = sample_packed.shape[0]
    k = min(k_max, max(threads, 8))
    dt, used = run(k)
    rate = k * mc / dt
    want = int(min(k_max, max(k, rate * args.cpu_seconds / mc)))
    if want > k * 1.5:
        k = want
        dt, used = run(k)
        rate = k * mc / dt
    what = ("HOT LOOP 1 (8 shift modes fwd+bwd, cnF2freq.cpp:5375-5382)" if kind == "reference"
            else "fwd+bwd over 8 shift modes + closed-form dosage rows")
    out = {"value": rate, "unit": "individual*marker/s", "cores": int(used), "kind": kind,
           "sample": "%d F2 individuals x %d markers (chromosome 1 of the GPU workload), %.1f s, %s"
                     % (k, mc, dt, what)}
    if kind == "reference":
        # the reference's shipped build forms no per-locus rows, so its own code times HOT LOOP 1 only;
        # the metric's unit includes the rows: the C restatement with them, on a smaller sample of the same data, beside it
        try:
            from oracle.pyoracle import OraclePed
            kp = min(k_max, max(threads, 8) * 2)
            ped = build_ped(kp)
            a, s_, h = ped.dense()
            o = OraclePed(a, s_, h, ped.par, ped.empty, ped.pos)
            t0 = time.perf_counter()
            r = o.sweep_batch(ped.dous, ped.gen[ped.dous], mode=2, n_threads=threads)
            dtp = time.perf_counter() - t0
            out["port_with_rows"] = {"value": kp * mc / dtp, "unit": "individual*marker/s", "cores": int(r["threads"]), "kind": "port",
                                     "sample": "%d F2 individuals x %d markers, %.1f s, fwd+bwd over 8 shift modes + closed-form dosage rows" % (kp, mc, dtp)}
        except Exception as e:
            out["port_with_rows"] = {"value": None, "sample": "failed: %r" % (e,)}
    return out


def cpu_baseline_iterations(args):
    """CPU path of a haplotyping iteration timed beside the GPU on a bounded sample of the same workload: a few families of
    the same generator (same seed, shape and missing rate), chromosome 1 only.  kind "reference" = the reference's own code
    (oracle/_ref: its sweep, HOT LOOP 2 with reductions and the update functions cnF2freq.cpp:4004-4734, driven by
    ref_iteration -- a serial replay of doit's loops, so 1 core); the oracle is the thing timed here, never the thing shipped."""
    from cnf2freq_amd import synth
    from oracle.ref_extract import pyref
    if not pyref.available(ieee=False):
        return {"value": None, "unit": "individual*marker/s per iteration", "cores": 0, "kind": "reference",
                "sample": "oracle/_ref is not built on this host"}

    def run(fams, snps):
        ped = synth.make_outbred3(fams, 4, snps, 1, seed=2, missing=0.2)
        R = pyref.RefPed(ped, ieee=False, fixtrees_all=False)
        R.set_priors()
        R.set_dous()
        R.L.ref_postmarkerdata(ped.n_rec + 1)
        hits = np.zeros(1, np.int32)
        t0 = time.perf_counter()
        R.L.ref_iteration(hits.ctypes.data, None, ped.n_rec)
        return time.perf_counter() - t0, len(ped.dous) * ped.n_markers

    # the reference spends ~14 ms per individual x marker of an iteration (its updates bisect with a 15-point quadrature of a
    # long polynomial per step): one family on a stretch of chromosome 1, sized from a 40-marker probe to the time budget
    snps = min(40, args.snps_per_chrom)
    dt, units = run(1, snps)
    want = int(min(args.snps_per_chrom, max(snps, snps * args.cpu_seconds / max(dt, 1e-3))))
    if want > snps * 1.5:
        snps = want
        dt, units = run(1, snps)
    return {"value": units / dt, "unit": "individual*marker/s per iteration", "cores": 1, "kind": "reference",
            "sample": "1 family (4 analysed individuals, 10 records) x %d markers of the same generator, one iteration after "
                      "postmarkerdata, %.1f s" % (units // 4, dt)}


def main_iterations(args):
    """BASELINE config 5's unit of work: haplotyping iterations of one pedigree, the analysed individuals split over the ranks
    in work-balanced blocks cut between families, every rank updating the records it owns; what ranks exchange per iteration
    is the records their windows share (one reduce-scatter of their accumulators, one all-gather of their new rows -- nothing
    when no family straddles a boundary) and the hit counters (cnf2freq_amd.dist.Transport over RCCL).  Strong scaling (the
    pedigree is fixed).  One JSON line from rank 0 with the time of every iteration."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if args.single_device else int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend="gloo")
        world = dist.get_world_size()        # n_gpus of the line = the world the process group really has
    if args.gpus != world and rank == 0:
        print("bench.py: --gpus %d but the process group has %d rank(s): reporting n_gpus = %d" % (args.gpus, world, world),
              file=sys.stderr, flush=True)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: there is no CPU fallback")
    torch.cuda.set_device(local)
    import __graft_entry__ as g
    if rank == 0:
        g.build()
    if world > 1:
        dist.barrier()
    from cnf2freq_amd import synth
    from cnf2freq_amd import dist as cdist
    chroms = 4 if args.chroms == 20 else args.chroms
    fams = 2500 if args.inds == 10000 else max(1, args.inds // 4)
    ped = synth.make_outbred3(fams, 4, args.snps_per_chrom, chroms, seed=2, missing=0.2)      # the same pedigree on every rank
    n, M, R = len(ped.dous), ped.n_markers, ped.n_rec
    # the engine prints the reference's progress lines ("Scale factor now ...") on stdout: they go to stderr here, so that
    # stdout carries the one JSON line
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    t0 = time.perf_counter()
    run = cdist.start_iterations(ped, device=local)
    t_setup = time.perf_counter() - t0
    st0 = run.state()                 # (a collective once iterations have run: every rank calls it)
    before = synth.dosage_accuracy(ped, st0) if rank == 0 else None
    del st0
    for _ in range(args.warmup):
        run.iteration()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    per_it = []
    t0 = time.perf_counter()
    for _ in range(args.iterations):
        t1 = time.perf_counter()
        run.iteration()
        per_it.append(time.perf_counter() - t1)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    staged = world > 1 and args.backend == "gloo"
    tmax = torch.tensor([dt], dtype=torch.float64, device=torch.device("cpu") if staged or world == 1 else torch.device("cuda", local))
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    sys.stdout.flush()
    import ctypes
    ctypes.CDLL(None).fflush(None)          # the C library's own buffer of the engine's lines, before stdout comes back
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    st = run.state()                  # gathers every rank's private records: a collective, outside the timed region
    if rank == 0:
        plan = run.plan
        out = {
            "metric": "haplotyping iterations/s (BASELINE config 5: sweep + accumulators + update passes per iteration)",
            "value": args.iterations / dt, "unit": "iterations/s", "n_gpus": world, "steps": args.iterations, "warmup": args.warmup,
            "ms_per_step": dt / args.iterations * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "3-generation outbred pedigree (BASELINE config 5): %d families = %d individuals, %d analysed, "
                                   "20%% of genotypes missing, %d chromosomes x %d SNPs (+1 dummy each)"
                                   % (fams, R, n, chroms, args.snps_per_chrom),
                       "analysed_individuals": n, "markers": M,
                       "parallelism": "analysed individuals in %d work-balanced block(s) cut between families; per iteration one "
                                      "reduce-scatter of the shared records' accumulators, update passes of the records a rank "
                                      "owns, one sum of the hit counters per pass, one all-gather of the shared records' rows" % world},
            # what the collectives carry per iteration: the buffers' bytes and the payload in them (shared records x (48 + 25) B
            # x markers + 4 B per chromosome pass); the full slabs an all-reduce of everything would move, for scale
            "exchange_bytes_per_iteration": plan["bytes_accumulators"] + plan["bytes_rows"] + plan["bytes_hits"],
            "exchange": {"shared_records": plan["n_shared"], "records": R, "payload_bytes_per_iteration": plan["bytes_payload"],
                         "reduce_scatter_buffer_bytes": plan["bytes_accumulators"], "all_gather_buffer_bytes": plan["bytes_rows"],
                         "hit_counter_bytes": plan["bytes_hits"], "full_slab_bytes": R * M * 48,
                         "bytes_moved_by_rank0_in_all": run.transport.bytes_moved,
                         "records_owned_by_rank0": int(len(plan["owned"]))},
            "units_per_s": float(n) * M * args.iterations / dt,
            "iteration_s": per_it, "block": list(run.block), "setup_s": t_setup,
            "scalefactor": st["scalefactor"], "last_hits": st["hits"],
            "withheld_genotypes_before": before, "withheld_genotypes_after": synth.dosage_accuracy(ped, st),
        }
        if args.cpu_seconds > 0:          # CPU baseline leg: rank 0, after the timed region, at every N
            try:
                out["cpu_baseline"] = cpu_baseline_iterations(args)
                if out["cpu_baseline"]["value"]:
                    out["gpu_over_cpu"] = out["units_per_s"] / out["cpu_baseline"]["value"]
            except Exception as e:  # the baseline must never take the GPU line down
                out["cpu_baseline"] = {"value": None, "unit": "individual*marker/s per iteration", "cores": 0, "kind": "reference",
                                       "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)
    run.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if args.iterations > 0:
        if args.workload != "outbred":
            raise SystemExit("--iterations goes with --workload outbred (BASELINE config 5)")
        return main_iterations(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.single_device:
        local = 0
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend="gloo")
        world = dist.get_world_size()        # n_gpus of the line = the world the process group really has
    if args.gpus != world and rank == 0:
        print("bench.py: --gpus %d but the process group has %d rank(s): reporting n_gpus = %d" % (args.gpus, world, world),
              file=sys.stderr, flush=True)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the sweep has no CPU fallback")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    import __graft_entry__ as g
    if rank == 0:
        g.build()
    if world > 1:
        dist.barrier()
    from cnf2freq_amd import capi, synth
    from cnf2freq_amd import dist as cdist

    sample = None
    if args.workload == "f2":
        pos, starts = synth.make_map(args.chroms, args.snps_per_chrom)
        M = len(pos)
        n = args.inds
        ctx = capi.Context(local)
        ctx.upload_map(pos, starts)
        sample = generate_on_gpu(ctx, args, rank, device, pos, starts)
        par, gen, empty, row_of, dous = synth.f2_pedigree_tables(n)
        ctx.upload_pedigree(par, empty, gen, row_of, dous)
        workload = ("synthetic F2 intercross, %d individuals x %d SNPs per GPU (%d chromosomes x %d + 1 dummy marker "
                    "each = %d markers swept), private empty F1 parents, 2 inbred founders"
                    % (n, args.chroms * args.snps_per_chrom, args.chroms, args.snps_per_chrom, M))
    else:
        # BASELINE configs 3 and 5 (shape): generated on the host by cnf2freq_amd.synth, uploaded through the C ABI
        if args.workload == "ail":
            chroms = 8 if args.chroms == 20 else args.chroms
            per_gen = 625 if args.inds == 10000 else max(1, args.inds // 8)
            ped = synth.make_ail(64, per_gen, 8, args.snps_per_chrom, chroms, seed=3 + rank)
            workload = ("advanced intercross (BASELINE config 3): 2 inbred founders, 64 genotyped F1, 8 random-mating "
                        "generations x %d analysed individuals, %d chromosomes x %d SNPs (+1 dummy each)"
                        % (per_gen, chroms, args.snps_per_chrom))
        else:
            chroms = 4 if args.chroms == 20 else args.chroms
            fams = 2500 if args.inds == 10000 else max(1, args.inds // 4)
            ped = synth.make_outbred3(fams, 4, args.snps_per_chrom, chroms, seed=2 + rank, missing=0.2)
            workload = ("3-generation outbred pedigree (BASELINE config 5 shape): %d families of 4 grandparents, 2 parents, "
                        "4 analysed children, 20%% of genotypes missing, %d chromosomes x %d SNPs (+1 dummy each)"
                        % (fams, chroms, args.snps_per_chrom))
        pos, starts = ped.pos, ped.chromstarts
        M = len(pos)
        n = len(ped.dous)
        args.chroms = len(starts) - 1
        ctx = capi.Context(local)
        ctx.upload(ped)
        del ped
    # the kernels a step launches: windows without tie groups on the plain instantiation, the others beside it
    n_tied = sum(1 for j in range(n) if (ctx.window_info(j)["tie"] >= 0).any()) if args.workload == "ail" else 0
    kernels = ["cnf2::fb_fast_kernel<true, 0, false, false>"] + (["cnf2::fb_fast_kernel<true, 0, false, true> (%d windows with tie groups)" % n_tied] if n_tied else [])

    factors = torch.empty((n, args.chroms, 8), dtype=torch.float64, device=device)
    do_gather = world > 1 and not args.no_gather
    overlap = do_gather and not args.no_overlap
    nbuf = 2 if overlap else 1
    logliks = [torch.empty((n, args.chroms), dtype=torch.float64, device=device) for _ in range(nbuf)]
    dosages = [torch.empty((n, M, 3), dtype=torch.float64, device=device) for _ in range(nbuf)]
    loglik, dosage = logliks[0], dosages[0]
    if overlap:
        ctx.set_grid_reserve(args.reserve_blocks)
    staged = world > 1 and args.backend == "gloo"
    gdev = torch.device("cpu") if staged else device
    # The one collective of the path: posteriors to rank 0 (RCCL over xGMI), streamed in marker tiles so that the
    # root holds 2 tiles x world_size and never the whole posterior (config 4: 60 GB per GPU, 480 GB in all).  The
    # root "consumes" a tile by folding it into a checksum (a real driver would write it out).
    tile_markers = args.gather_tile_markers or int(starts[1] - starts[0])
    tiler = gather_ll = None
    gsum = torch.zeros(1, dtype=torch.float64, device=gdev)
    if do_gather:
        tiler = cdist.TiledGather(n, M, 3, tile_markers, torch.float64, gdev, dst=0, depth=2)
        if rank == 0:
            gather_ll = [torch.empty(loglik.shape, dtype=loglik.dtype, device=gdev) for _ in range(world)]

    def consume(m0, m1, parts):
        for t in parts:
            gsum.add_(t.sum())

    kernel_ms = []
    gather_ms = []
    pending = [None] * nbuf     # event after the enqueued gathers that read buffer i
    state = {"k": 0}

    def drain(i):
        if pending[i] is not None:
            for w in pending[i][0]:
                w.wait()
            pending[i][1].synchronize()
            pending[i] = None

    def step():
        i = state["k"] % nbuf
        state["k"] += 1
        drain(i)                             # the gathers that last read this buffer must be done
        ll_i, dos_i = logliks[i], dosages[i]
        ctx.sweep_device(0, n, factors.data_ptr(), ll_i.data_ptr(), dos_i.data_ptr(),
                         (capi.FULL_SPILL if args.full_spill else 0) | args.extra_flags)
        ctx.sync()
        kernel_ms.append(ctx.last_kernel_ms())
        if do_gather:
            # with overlap the calls below only enqueue: the tile gathers run beside the next sweep (own
            # streams; the sweep leaves --reserve-blocks workgroup slots to the RCCL kernels) and are awaited
            # before their buffer is reused and at the end of the timed region
            tg = time.perf_counter()
            src_ll = ll_i.cpu() if staged else ll_i
            src_d = dos_i.cpu() if staged else dos_i
            works = [dist.gather(src_ll, gather_ll if rank == 0 else None, dst=0, async_op=True)]
            tiler.run(src_d, consume)
            ev = torch.cuda.Event()
            ev.record()
            pending[i] = (works, ev, src_ll, src_d)
            if not overlap:
                drain(i)
            gather_ms.append((time.perf_counter() - tg) * 1e3)

    def finish():
        for i in range(nbuf):
            drain(i)

    if do_gather:
        # set up the point-to-point connections of the gather before anything is timed
        tiny = torch.zeros(8, dtype=torch.float64, device=gdev)
        cdist.gather_to_root(tiny, 0)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    finish()
    kernel_ms.clear()
    gather_ms.clear()
    gsum.zero_()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    finish()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    loglik, dosage = logliks[(state["k"] - 1) % nbuf], dosages[(state["k"] - 1) % nbuf]
    tmax = torch.tensor([dt], dtype=torch.float64, device=torch.device("cpu") if staged else device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    # every rank's own launch time and kernel clock (roofline per rank: the ranks run the same launch on their own GPU)
    sweep_clock_mhz = ctx.sweep_clock()          # the last timed launch's own shader / wall clock stamps
    k_ms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
    mine = torch.tensor([k_ms, sweep_clock_mhz], dtype=torch.float64, device=tmax.device)
    per_rank = [mine]
    if world > 1:
        per_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(per_rank, mine)
    per_rank = [[float(x) for x in t.cpu()] for t in per_rank]

    if rank == 0:
        units_per_step = float(n) * M * world
        value = units_per_step * args.steps / dt
        achieved = (float(n) * M * B_UNIT) / (k_ms * 1e-3) / 1e9   # GB/s, this rank's launch
        # HBM bytes per launch from the PMC counters (profiles/hbm_traffic.json, written by
        # tools/profile_round.sh): reported only if it was measured for THIS workload and THESE kernel sources
        traffic = None
        src_sha = kernel_source_sha()
        try:
            tj = json.load(open(args.traffic_file))
            if (tj.get("inds") == n and tj.get("markers") == M and tj.get("kernel_src_sha") == src_sha
                    and args.workload == "f2" and not args.extra_flags and not args.full_spill):
                traffic = tj.get("bytes_per_launch")
        except Exception:
            traffic = None
        # outside the timed region: the outputs are overwritten with NaN and the sweep is run once more -- the waves of a
        # launch take their jobs from a counter, and a job nobody took would leave the previous step's (equal) values in
        # place unnoticed.  Every output must be written again and the likelihoods must equal the timed step's to the bit.
        ll_timed = loglik.clone()
        for t in (loglik, dosage, factors):
            t.fill_(float("nan"))
        ctx.sweep_device(0, n, factors.data_ptr(), loglik.data_ptr(), dosage.data_ptr(),
                         (capi.FULL_SPILL if args.full_spill else 0) | args.extra_flags)
        ctx.sync()
        every_output_rewritten = bool(torch.equal(loglik, ll_timed) and not torch.isnan(dosage).any().item()
                                      and not torch.isnan(factors).any().item())
        del ll_timed
        # size-independent checks on the full output of that step:
        # every row is a distribution, every likelihood is finite and negative, and the logsumexp
        # identity between per-mode and total likelihoods holds
        rs = dosage.sum(dim=2)
        fmx = factors.max(dim=2, keepdim=True).values
        lse = (fmx.squeeze(2) + torch.log(torch.exp(factors - fmx).sum(dim=2)))
        checks = {
            "rows_sum_to_one": bool(((rs - 1.0).abs() < 1e-9).all().item()),
            "rows_nonnegative": bool((dosage >= 0).all().item()),
            "loglik_finite_negative": bool((torch.isfinite(loglik) & (loglik < 0)).all().item()),
            "logsumexp_identity": bool(((lse - loglik).abs() < 1e-9 * loglik.abs().clamp(min=1.0)).all().item()),
            "every_output_rewritten": every_output_rewritten,
        }
        del rs, fmx, lse
        ll = loglik.cpu().numpy()
        # outside the timed region and after the checks, rank 0 at N = 1: the same sweep with CNF2_MERGE_MODES
        # (exact; the F2's empty F1 parents make the four shift modes that share s0 bit-identical, see DESIGN.md
        # 5b) -- reported next to the headline, never as the headline
        merge_info = None
        if world == 1 and not args.extra_flags and not args.full_spill and not args.no_merge_probe and args.workload == "f2":
            ns = min(n, 512)
            rows_h = dosage[:ns].clone()
            ll_h = loglik.clone()
            ms_m = []
            for _ in range(2):
                ctx.sweep_device(0, n, factors.data_ptr(), loglik.data_ptr(), dosage.data_ptr(), capi.MERGE_MODES)
                ctx.sync()
                ms_m.append(ctx.last_kernel_ms())
            merge_info = {"value": float(n) * M / (ms_m[-1] * 1e-3), "unit": "individual*marker/s",
                          "kernel_ms": ms_m[-1], "kernel": "cnf2::fb_packed_kernel",
                          "loglik_bit_identical": bool(torch.equal(loglik, ll_h)),
                          "loglik_max_rel_diff": float(((loglik - ll_h).abs() / ll_h.abs()).max().item()),
                          "rows_max_abs_diff": float((dosage[:ns] - rows_h).abs().max().item()),
                          "note": "CNF2_MERGE_MODES: modes differing only in the shift bits of parents that are "
                                  "homozygous everywhere are swept once; all 8 modes are output"}
            del rows_h, ll_h

        # the shader clock this device runs at under a vector load (boxes differ by several per cent and the sweep is
        # issue-bound, so its time tracks the clock); VALU instructions per unit from the SQ counters where they were
        # measured for these kernel sources (profiles/hbm_traffic.json)
        probe_mhz = ctx.clock_probe()
        # the kernel's own stamps when its shader-clock counter really counts shader cycles (a constant-rate counter would
        # read as the wall clock's 100 MHz), else the FMA-loop probe
        clock_mhz = sweep_clock_mhz if sweep_clock_mhz > 500.0 else probe_mhz
        valu_per_unit = None
        try:
            tj = json.load(open(args.traffic_file))
            if tj.get("kernel_src_sha") == src_sha and args.workload == "f2":
                valu_per_unit = tj.get("valu_per_unit")
        except Exception:
            pass
        n_simd = 256 * 4
        valu_issue_frac = (valu_per_unit * float(n) * M / n_simd * 4.0 / (k_ms * 1e-3 * clock_mhz * 1e6)) if valu_per_unit else None
        out = {
            "metric": "individual*marker fwd-bwd steps/sec (all 8 shift modes, forward+backward, dosage rows)",
            "value": value, "unit": "individual*marker/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload,
                       "individuals_per_gpu": n, "markers": M, "shift_modes": 8, "states": 64,
                       "parallelism": "individuals sharded over %d GPU(s)%s" % (
                           world, (", posteriors gathered to rank 0 per step in tiles of %d markers (%d tiles, rank 0 "
                                   "holds 2 tiles x %d ranks = %.2f GB)%s"
                                   % (tile_markers, tiler.n_tiles(), world, tiler.root_bytes() / 1e9,
                                      ", overlapped with the next sweep" if overlap else "")) if do_gather else "")},
            # achieved / frac: ALGORITHMIC bytes (8 248 B per unit, textbook forward-backward) over the
            # kernel's time.  traffic / frac_physical: what the HBM counters saw for the same launch (the half
            # spill moves about half the textbook bytes), null when not measured for these kernel sources.
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved * 1e9 / HBM_PEAK, "traffic": traffic,
                         "frac_physical": (traffic / (k_ms * 1e-3) / HBM_PEAK) if traffic else None,
                         "kernel": " + ".join(kernels), "kernel_ms": k_ms,
                         "algorithmic_bytes_per_unit": B_UNIT, "kernel_src_sha": src_sha,
                         "effective_clock_mhz": clock_mhz, "sweep_kernel_clock_mhz": sweep_clock_mhz,
                         "fma_probe_clock_mhz": probe_mhz,
                         "valu_per_unit": valu_per_unit, "valu_issue_frac": valu_issue_frac,
                         # what binds: the vector ALU's issue slots.  `frac` prices SURVEY 8(d)'s textbook bytes (alpha stored
                         # and reloaded at every marker); the kernel spills every second marker and rebuilds the others, so
                         # it moves about half of them (frac_physical) and `frac` can pass 1 without HBM being the limit
                         "binding": "valu_issue",
                         # ... so, beside the contract's `frac`: the same launch against the other ceilings.  frac_flops = SURVEY
                         # 8(d)'s algorithmic flops (2.7e4 per unit: the butterflies, the emission combine, the normalisation;
                         # 8 modes, forward and backward) over the kernel's time over the FP64 vector peak; the physical bytes
                         # against what HBM delivers to a streaming kernel (6.3 TB/s, MI355X_MICROARCH.md) rather than its 8 TB/s
                         "bound_actual": "fp64_valu_issue",
                         "hbm_ceiling_of_survey_8d_reached": bool(achieved * 1e9 / HBM_PEAK >= 0.97),
                         "algorithmic_flops_per_unit": 2.7e4,
                         "frac_flops": 2.7e4 * float(n) * M / (k_ms * 1e-3) / 78.6e12,
                         "frac_physical_achievable": (traffic / (k_ms * 1e-3) / 6.3e12) if traffic else None,
                         "valu_per_unit_source": "profiles/hbm_traffic.json: rocprofv3 --pmc of the same kernel sources (kernel_src_sha), not of this run",
                         # achieved / frac above are rank 0's launch; every rank's own launch beside it
                         "per_rank": [{"rank": r, "kernel_ms": km, "achieved": float(n) * M * B_UNIT / (km * 1e-3) / 1e9,
                                       "frac": float(n) * M * B_UNIT / (km * 1e-3) / HBM_PEAK,
                                       "sweep_kernel_clock_mhz": ck} for r, (km, ck) in enumerate(per_rank)]},
            "loglik_checksum": float(np.sum(ll[np.isfinite(ll)])),
            "checks": checks,
            "gather_ms_per_step": float(np.mean(gather_ms)) if gather_ms else 0.0,
            "gather_checksum": float(gsum.item()) if do_gather else None,
        }
        if merge_info:
            out["merge_modes"] = merge_info
        if args.cpu_seconds > 0 and sample is not None:      # CPU baseline leg: rank 0, after the timed region, at every N
            try:
                out["cpu_baseline"] = cpu_baseline(sample, pos, starts, args)
                out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]      # whole job over the CPU path
            except Exception as e:  # the baseline must never take the GPU line down
                out["cpu_baseline"] = {"value": None, "unit": "individual*marker/s", "cores": 0, "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        if not args.no_iteration_probe:
            # the other unit of work of the path (BASELINE config 5: whole haplotyping iterations) into the driver-observed
            # line; the headline's context is closed first so that the probe sees the memory a run of its own would
            ctx.close()
            del dosages, logliks, dosage, loglik, factors
            torch.cuda.empty_cache()
            try:
                out["iteration_probe"] = iteration_probe(local, device)
            except Exception as e:
                out["iteration_probe"] = {"failed": repr(e)}
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
