"""Round-3 extended randomized parity run (several seeds, three regimes): default policy, COLD hint (streaming-load forms), and large arrays (2^21 points);
then the sharded device-resident pipeline on random shapes / axes / views with NDFFT_SHARD_FORCE_REMOTE=1."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import parity_suite as ps
from ndrustfft_amd import _lib
L = _lib.default()
LENS = (1, 2, 3, 5, 8, 11, 15, 16, 20, 27, 36, 49, 60, 64, 81, 100, 121, 125, 128, 144, 169, 180, 210, 240, 256, 289, 320, 360, 385, 420, 511, 512, 540, 625, 720, 768, 840, 1000,
        1023, 1024, 1155, 1331, 1536, 2000, 2047, 2048, 2310, 2520, 3125, 4095, 4096, 4098, 6000, 6561, 8191, 8192, 8193, 10007, 12000, 16384, 16385, 20000, 32768, 40000, 65536)
t0 = time.time()
tot = {}
for seed in (501, 502, 503):
    for hint in (_lib.INPUT_AUTO, _lib.INPUT_COLD):
        L.check(L.c.ndfft_set_input_hint(hint))
        p = ps.fuzz(L, seed=seed + 10 * hint, count=300, lengths=LENS)
        for k, v in p.items(): tot[k] = tot.get(k, 0) + v
        print("seed", seed, "hint", hint, "ok", round(time.time() - t0), "s", flush=True)
L.check(L.c.ndfft_set_input_hint(_lib.INPUT_COLD))
for seed in (21, 22):
    p = ps.fuzz(L, seed=seed, count=120, max_points=1 << 21, lengths=(60, 96, 100, 127, 128, 210, 256, 264, 385, 500, 511, 512, 840, 1000, 1009, 1024, 2000, 2048, 2520, 3003, 4096, 4099, 6000, 8192, 10000, 16384))
    for k, v in p.items(): tot[k] = tot.get(k, 0) + v
    print("big seed", seed, "ok", round(time.time() - t0), "s", flush=True)
L.check(L.c.ndfft_set_input_hint(_lib.INPUT_AUTO))
print(sorted(tot.items(), key=lambda kv: -kv[1]))
# sharded device-resident pipeline, random cases
os.environ["NDFFT_SHARD_FORCE_REMOTE"] = "1"; os.environ["NDFFT_SHARD_CHUNK_KB"] = "64"
rng = np.random.default_rng(99)
names = ["ndfft", "ndifft", "ndfft_r2c", "ndifft_r2c", "nddct1", "nddct2", "nddct3", "nddct4"]
for it in range(60):
    name = names[rng.integers(len(names))]
    ndim = int(rng.integers(2, 4))
    shape = tuple(int(rng.integers(2, 40)) for _ in range(ndim))
    axis = int(rng.integers(ndim))
    n = int(rng.choice([6, 16, 30, 64, 100, 128, 257]))
    shape = shape[:axis] + (n,) + shape[axis + 1:]
    rdt = np.float64 if rng.integers(2) else np.float32
    ids = [0] * int(rng.integers(1, 4))
    kw = {}
    if rng.integers(3) == 0:      # output view with holes: every second element of the last (or first) dimension of a larger allocation
        sin, sout = ps.shapes_for(name, shape, axis)
        d = ndim - 1 if axis != ndim - 1 else 0
        alloc = list(sout); alloc[d] *= 2
        idx = [slice(None)] * ndim; idx[d] = slice(None, None, 2)
        kw["out_view"] = (tuple(alloc), tuple(idx))
    ps.dev_sharded_case(L, name, shape, axis, root=0, ids=ids, rdt=rdt, repeats=1, **kw)
print("sharded pipeline fuzz ok", round(time.time() - t0), "s")
