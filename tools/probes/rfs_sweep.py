"""real four-step vs the packed complex route over lane lengths 2^16 .. 2^21 (2^24 points per array), all four ops, both dtypes.
One process per route (the switch is read per call, the split per plan): python rfs_sweep.py"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np, torch
    from ndrustfft_amd import DctHandler, R2cFftHandler, _lib, nddct2, nddct3, nddct4, ndfft_r2c, ndifft_r2c
    dev = torch.device("cuda:0")
    def t(fn, *a, steps=10):
        for _ in range(3): fn(*a)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps): fn(*a)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / steps
    out = {}
    for rdt, cdt in ((np.float64, np.complex128), (np.float32, np.complex64)):
        tr = torch.from_numpy(np.zeros(1, rdt)).dtype; tc = torch.from_numpy(np.zeros(1, cdt)).dtype
        if os.environ.get("SWEEP_DTYPE") and os.environ["SWEEP_DTYPE"] != np.dtype(rdt).name: continue
        for e in [int(v) for v in os.environ.get("SWEEP_E", "16,17,18,19,20,21").split(",")]:
            n = 1 << e; L = (1 << 24) // n
            x = torch.randn((L, n), dtype=tr, device=dev); y = torch.empty_like(x)
            xh = torch.randn((L, n // 2 + 1), dtype=tc, device=dev)
            hd = DctHandler(n, rdt); hr = R2cFftHandler(n, rdt)
            for name, fn, a, b, h in (("nddct2", nddct2, x, y, hd), ("nddct3", nddct3, x, y, hd), ("nddct4", nddct4, x, y, hd), ("ndfft_r2c", ndfft_r2c, x, xh, hr), ("ndifft_r2c", ndifft_r2c, xh, y, hr)):
                if os.environ.get("SWEEP_OPS") and name not in os.environ["SWEEP_OPS"].split(","): continue
                us = t(fn, a, b, h, 1)
                out[f"{name} {np.dtype(rdt).name} {L}x2^{e}"] = (round(us, 1), _lib.default().last_path())
    print("RESULT " + json.dumps(out))
    sys.exit(0)
res = {}
routes = [("packed", {"NDFFT_REAL_FOURSTEP": "0"}), ("real", {"NDFFT_REAL_FOURSTEP": os.environ.get("SWEEP_REAL_MODE", "1")})] + [(f"N1=2^{a}", {"NDFFT_RFS_LOGN1": a}) for a in os.environ.get("SWEEP_LOGN1", "").split(",") if a]
for label, env in routes:
    p = subprocess.run([sys.executable, __file__, "child"], env={**os.environ, **env}, capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
    if not line: print(p.stdout[-2000:], p.stderr[-2000:]); sys.exit(1)
    res[label] = json.loads(line[0][7:])
print(f"{'case':40s} " + " ".join(f"{l:>10s}" for l, _ in routes) + "  packed/real  path")
for k in res["packed"]:
    a, b = res["packed"][k][0], res["real"][k][0]
    print(f"{k:40s} " + " ".join(f"{res[l][k][0]:10.1f}" for l, _ in routes) + f"  {a / b:5.2f}  {res['real'][k][1]}")
