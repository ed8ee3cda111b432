"""On one MI355X: the torch.distributed (backend "nccl" = RCCL) code path of bench.py and of the
lane-sharding layer at world size 1 -- the 2/4/8-GPU runs are the driver's."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_distributed_path_world1():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                        "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20",
                        "--warmup", "3", "--dist", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["value"] > 50 and out["roofline"]["frac"] > 0.3
    for k in ("metric", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in out


def test_scatter_exec_gather_rccl_world1():
    code = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import synth
from ndrustfft_amd import FftHandler, ndfft, distributed as nd
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
x = synth.complex_array((64, 256))
full = torch.from_numpy(x).cuda()
out = nd.transform_sharded(ndfft, full, x.shape, full.dtype, x.shape, full.dtype, FftHandler(256), 1, device=torch.device("cuda", 0))
torch.cuda.synchronize()
err = np.abs(out.cpu().numpy() - np.fft.fft(x, axis=1)).max()
assert err < 1e-10, err
dist.destroy_process_group()
print("RCCL_OK")
''' % (ROOT, ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29534", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_sharded_fft2_rccl_world1():
    """transform_axes_sharded (slab -> re-shard -> slab) with the HIP kernels as the local executor."""
    code = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import synth
from ndrustfft_amd import FftHandler, R2cFftHandler, ndfft, ndfft_r2c, distributed as nd
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
x = synth.real_array((96, 128))
loc = torch.from_numpy(x).cuda()
steps = [(ndfft_r2c, R2cFftHandler(128), 1, 65, torch.complex128), (ndfft, FftHandler(96), 0, 96, torch.complex128)]
y, gshape, d = nd.transform_axes_sharded(steps, loc, x.shape, 0)
torch.cuda.synchronize()
ref = np.fft.fft(np.fft.rfft(x, axis=1), axis=0)
err = np.abs(y.cpu().numpy() - ref).max() / np.abs(ref).max()
assert gshape == (96, 65) and d == 1 and err < 1e-10, (gshape, d, err)
back = nd.reshard(y, gshape, d, 0)
assert torch.equal(back, y)
dist.destroy_process_group()
print("RCCL_OK")
''' % (ROOT, ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29535", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
