"""GPU suite: the N > 1 path with the HIP sweep as the per-rank worker (BASELINE config 4's partition,
cnF2freq.cpp:5294-5298: analysed individuals are split across ranks, nothing else is exchanged but the
results).  Two ranks share the one GPU of the test box, so the transport is gloo (RCCL refuses two ranks on
one device); the sweep, the sharding and the marker-tiled gather are the code bench.py runs under RCCL."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("tile", [7, 38])
def test_two_rank_hip_sweep_equals_single_rank(tmp_path, tile):
    import __graft_entry__ as g
    g.build()
    from cnf2freq_amd import capi
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_worker
    out = str(tmp_path / "two_rank.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), out, "gloo",
           str(tile)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    z = np.load(out)
    ped = dist_worker.make_ped()
    ctx = capi.Context(0)
    ctx.upload(ped)
    one = ctx.sweep()
    acc = ctx.sweep_accumulate(ctx.descendants())
    ctx.close()
    # bit-equality with the single-rank sweep: sharding and tiling must not change a single value
    assert np.array_equal(z["dosage"], one["dosage"])
    assert np.array_equal(z["loglik"], one["loglik"])
    assert np.array_equal(z["factors"], one["factors"])
    # the accumulators of a haplotyping sweep: per-rank partial sums + one all-reduce = the single-rank sums (the F2
    # founders are ancestors of every individual on both ranks); equal to rounding, the order of the additions differs
    for k in ("infprobs", "haplobase", "haplocount"):
        np.testing.assert_allclose(z["acc_" + k], acc[k], rtol=1e-11, atol=1e-13, equal_nan=True, err_msg=k)
    assert np.abs(acc["infprobs"][0]).max() > 0, "founder A should collect evidence from every F2"
    # the tiles cover the marker axis exactly once, in order
    tiles = z["tiles"]
    assert tiles[0, 0] == 0 and tiles[-1, 1] == ped.n_markers and np.all(tiles[1:, 0] == tiles[:-1, 1])
    # rank 0 never held more than depth (2) tiles x world for receiving
    assert int(z["root_bytes"]) <= 2 * int(z["world"]) * int(z["tile_bytes"])
