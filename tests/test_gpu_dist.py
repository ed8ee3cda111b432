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


_TWO_RANK_WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import synth
from ndrustfft_amd import FftHandler, R2cFftHandler, ndfft, ndfft_r2c, distributed as nd
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("gloo", rank=rank, world_size=world)
# every rank derives the same global array; it keeps only its slab of dimension 0 on the device
x = synth.real_array((90, 128))                 # 90 rows over 2 ranks, 65 columns over 2 ranks: ragged slabs both times
lo, hi = nd.shard_bounds(90, world)[rank]
loc = torch.from_numpy(x[lo:hi].copy()).to(dev)
steps = [(ndfft_r2c, R2cFftHandler(128), 1, 65, torch.complex128), (ndfft, FftHandler(90), 0, 90, torch.complex128)]
y, gshape, d = nd.transform_axes_sharded(steps, loc, x.shape, 0)
torch.cuda.synchronize()
assert y.is_cuda and gshape == (90, 65) and d == 1, (y.device, gshape, d)
ref = np.fft.fft(np.fft.rfft(x, axis=1), axis=0)
clo, chi = nd.shard_bounds(65, world)[rank]
err = np.abs(y.cpu().numpy() - ref[:, clo:chi]).max() / np.abs(ref).max()
assert err < 1e-10, err
# back to row slabs: the opposite exchange
back = nd.reshard(y, gshape, 1, 0)
assert back.is_cuda and tuple(back.shape) == (hi - lo, 65) and np.abs(back.cpu().numpy() - ref[lo:hi]).max() / np.abs(ref).max() < 1e-10
# scatter from the root / transform / gather on the root, device tensors all the way
z = synth.complex_array((33, 256))
full = torch.from_numpy(z).to(dev) if rank == 0 else None
out = nd.transform_sharded(ndfft, full, z.shape, torch.complex128, z.shape, torch.complex128, FftHandler(256), 1, device=dev)
torch.cuda.synchronize()
if rank == 0:
    e2 = np.abs(out.cpu().numpy() - np.fft.fft(z, axis=1)).max()
    assert out.is_cuda and e2 < 1e-10, e2
else:
    assert out is None
dist.barrier()
dist.destroy_process_group()
print("GLOO2_OK", rank)
'''


def test_sharded_transforms_two_ranks_one_gpu_gloo():
    """World size 2 with device-resident shards and the HIP kernels as the local executor: the two ranks share this box's one GPU
    (RCCL refuses two ranks on one device), so the exchange travels over gloo through host staging (distributed._exchange) -- what runs
    here that the world-1 tests above cannot is the pairwise send / receive logic of reshard / scatter_lanes / gather_lanes between
    two processes whose slabs live in HBM."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    code = _TWO_RANK_WORKER % (ROOT, ROOT)
    procs = []
    for rank in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, so, se))
    for rank, (rc, so, se) in enumerate(outs):
        assert rc == 0 and "GLOO2_OK %d" % rank in so, (rank, so[-2000:], se[-3000:])
