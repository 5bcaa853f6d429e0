"""bench.py --gpus N starts N ranks by itself (no launcher around it): the parent must not have imported anything that
initialises HIP when it spawns `python -m torch.distributed.run ...` as a child, and it leaves with the child's exit
code.  The partition the ranks take is the reference's own (cnF2freq.cpp:5297-5299).  On a box without a GPU the ranks
stop with "bench.py needs a GPU" -- which is exactly what shows that N of them were started."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def test_requested_gpus_parsing():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.requested_gpus(["--steps", "3"]) == 1
    assert bench.requested_gpus(["--gpus", "4", "--steps", "3"]) == 4
    assert bench.requested_gpus(["--steps", "3", "--gpus=8"]) == 8


def test_a_rank_or_one_gpu_never_spawns(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    assert bench.launch_ranks(["--gpus", "1"]) is None
    assert bench.launch_ranks([]) is None
    monkeypatch.setenv("WORLD_SIZE", "2")
    assert bench.launch_ranks(["--gpus", "2"]) is None      # already a rank of a launcher's group


def test_parent_spawns_ranks_before_touching_hip(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("the CPU form of this test: the GPU form is test_bench_two_ranks_without_a_launcher")
    trace = tmp_path / "parent.json"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["CNF2_BENCH_PARENT_TRACE"] = str(trace)
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--single-device", "--inds", "16",
                        "--chroms", "1", "--snps-per-chrom", "50", "--steps", "1", "--warmup", "0", "--cpu-seconds", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    t = json.load(open(trace))
    mods = set(t["modules"])
    for bad in ("torch", "numpy", "ctypes", "cnf2freq_amd", "cnf2freq_amd.capi", "__graft_entry__"):
        assert bad not in mods, "the parent had imported %s before the spawn" % bad
    assert t["cmd"][1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in t["cmd"]
    assert t["cmd"][t["cmd"].index("--nproc-per-node") + 1] == "2"
    assert "--standalone" in t["cmd"] and "127.0.0.1" in t["cmd"]       # torchrun picks the rendezvous port itself
    # no GPU here: the ranks refuse to run (no CPU fallback), and the parent relays the failure
    assert r.returncode != 0
    assert "needs a GPU" in r.stderr, r.stderr[-3000:]
    assert r.stdout.strip() == ""


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["sweep", "iterations"])
def test_bench_two_ranks_without_a_launcher(mode):
    """`python bench.py --gpus 2` on a one-GPU box (gloo transport, both ranks on GPU 0): one JSON line, n_gpus = 2, a
    roofline entry per rank and the CPU baseline beside it."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--single-device", "--steps", "2", "--warmup", "1"]
    if mode == "sweep":
        cmd += ["--inds", "300", "--chroms", "2", "--snps-per-chrom", "400", "--cpu-seconds", "1"]
    else:
        cmd += ["--workload", "outbred", "--iterations", "2", "--inds", "240", "--chroms", "2", "--snps-per-chrom", "300",
                "--cpu-seconds", "1"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2
    assert out["value"] > 0
    if mode == "sweep":
        assert out["scaling"] == "weak"
        assert len(out["roofline"]["per_rank"]) == 2
        assert all(p["kernel_ms"] > 0 for p in out["roofline"]["per_rank"])
        assert out["cpu_baseline"]["value"] > 0
        assert all(out["checks"].values())
    else:
        assert out["scaling"] == "strong"
        assert "exchange_bytes_per_iteration" in out
