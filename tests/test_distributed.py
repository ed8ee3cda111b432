"""N>1 path on CPU: world_size 2, gloo.  (The 8-GPU RCCL run is the driver's; bench.py --gpus N
uses device-resident shards and no collective in the timed region.)"""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_lane_sharding_world2_gloo():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(ROOT, "tests", "dist_worker.py")], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "DIST_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_xgmi_block_at_bench_shapes_world8_gloo():
    """bench.py's xGMI block (scatter / gather of 128 MiB slices, (1024 N)^2 all-to-all, sharded fft2) with the real run's argument shapes at N = 8."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", DIST_WORKER_MODE="xgmi_bench_shapes")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(ROOT, "tests", "dist_worker.py")], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "DIST_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_shard_helpers():
    from ndrustfft_amd import distributed as d
    assert d.shard_dim((4096, 4096), 1) == 0
    assert d.shard_dim((8192, 8192), 0) == 1
    assert d.shard_dim((1, 5, 7), 2) == 1
    assert d.shard_bounds(65536, 8) == [(i * 8192, (i + 1) * 8192) for i in range(8)]
    assert d.local_shape((65536, 4096), 1, 3, 8) == ((8192, 4096), 0, (24576, 32768))


def test_bench_spawns_n_ranks_dry_gloo():
    """`python bench.py --gpus 2` with no launcher environment must itself start two ranks (--gpus was
    parsed and ignored).  --dry: rendezvous + an all-reduce that counts the ranks, no GPU work."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                      # rank 0 prints ONE JSON line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2


@pytest.mark.parametrize("nranks", [2, 8])
def test_bench_under_launcher_dry_gloo(nranks):
    """The driver's N > 1 shape: torch.distributed.run provides the rank environment; bench.py must not spawn again.
    nranks = 8 is the exact command line of the driver's SCALE run (8 ranks of one node), so that spawn path has run once."""
    import json
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks),
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(ROOT, "bench.py"), "--gpus", str(nranks), "--steps", "20", "--warmup", "5", "--backend", "gloo", "--dry"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["ranks_seen"] == nranks and out["n_gpus"] == nranks, r.stdout


def test_synth_torch_generator_matches_numpy():
    import numpy as np
    import synth
    a = synth.complex_array((37, 64), offset=12345)
    b = synth.complex_array_torch((37, 64), "cpu", offset=12345, chunk_rows=5).numpy()
    assert np.array_equal(a, b)
