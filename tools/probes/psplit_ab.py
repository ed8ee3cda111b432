"""A-B of the position-split exchange (NDFFT_PSPLIT=1 off / 2 on) on n = 8192 and 16384 C2C rows, 2^24 points, warm and cold (6 rotating pairs)"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np, torch
    from ndrustfft_amd import FftHandler, ndfft
    dev = torch.device("cuda:0")
    out = {}
    for n in (8192, 16384):
        for rdt, cdt in ((np.float64, torch.complex128), (np.float32, torch.complex64)):
            rows = (1 << 24) // n
            pairs = [(torch.randn((rows, n), dtype=cdt, device=dev), torch.empty((rows, n), dtype=cdt, device=dev)) for _ in range(6)]
            h = FftHandler(n, rdt)
            def run(np_, steps=120):
                for i in range(30): ndfft(*pairs[i % np_], h, 1)
                torch.cuda.synchronize()
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(steps): ndfft(*pairs[i % np_], h, 1)
                e1.record(); torch.cuda.synchronize()
                return e0.elapsed_time(e1) * 1e3 / steps
            nbytes = 2 * rows * n * (16 if rdt == np.float64 else 8)
            w, c = run(1), run(6)
            out[f"{n} {'c128' if rdt == np.float64 else 'c64'}"] = [round(w, 1), round(nbytes / w / 8e6, 3), round(c, 1), round(nbytes / c / 8e6, 3)]
    print(json.dumps(out))
else:
    for rep in range(3):
        for ps in ("1", "2"):
            env = dict(os.environ, NDFFT_PSPLIT=ps)
            r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            print(f"PSPLIT={ps}:", line[-1] if line else r.stderr[-300:], flush=True)
