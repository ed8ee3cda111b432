"""N>1 path on CPU: world_size 2, gloo.  (The 8-GPU RCCL run is the driver's; bench.py --gpus N
uses device-resident shards and no collective in the timed region.)"""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_lane_sharding_world2_gloo():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(ROOT, "tests", "dist_worker.py")], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "DIST_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_shard_helpers():
    from ndrustfft_amd import distributed as d
    assert d.shard_dim((4096, 4096), 1) == 0
    assert d.shard_dim((8192, 8192), 0) == 1
    assert d.shard_dim((1, 5, 7), 2) == 1
    assert d.shard_bounds(65536, 8) == [(i * 8192, (i + 1) * 8192) for i in range(8)]
    assert d.local_shape((65536, 4096), 1, 3, 8) == ((8192, 4096), 0, (24576, 32768))
