#!/usr/bin/env python3
"""Probe (round 6, late): f32 C2C rows of the lengths whose default recipe cannot use 16-byte accesses, with and without jit.hip's jit_c2c_row_vec (developer build,
knob NDFFT_JIT_ROW_VEC = 0 / 1, one process per setting, three alternating rounds).  HBM-sourced (3 rotating pairs of 2^24 points), every output checked against torch.fft."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NS = (432, 500, 576, 648, 864, 1000, 1296, 2000, 2500, 3456, 5000, 5184)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from ndrustfft_amd import FftHandler, ndfft, _lib
    import numpy as np
    dev = torch.device("cuda:0")
    for n in NS:
        rows = (1 << 24) // n
        xs = [torch.randn(rows, n, dtype=torch.complex64, device=dev) for _ in range(3)]
        ys = [torch.empty_like(x) for x in xs]
        h = FftHandler(n, np.float32)
        for k in range(3):
            ndfft(xs[k], ys[k], h, 1)
        torch.cuda.synchronize()
        ref = torch.fft.fft(xs[0].to(torch.complex128), dim=1)
        err = float((ys[0].to(torch.complex128) - ref).abs().max() / ref.abs().max())
        ts = []
        for _ in range(3):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(30):
                ndfft(xs[i % 3], ys[i % 3], h, 1)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 30)
        print(json.dumps({"knob": os.environ.get("NDFFT_JIT_ROW_VEC"), "n": n, "us": round(sorted(ts)[1], 2), "rel_err": err, "path": _lib.default().last_path()}), flush=True)
        assert err < 1e-5, (n, err)
        del xs, ys
    sys.exit(0)
for rnd in range(3):
    for knob in ("0", "1"):
        env = dict(os.environ, NDFFT_MI355X_LIB=os.path.join(ROOT, "ndrustfft_amd", "csrc", "libndfft_mi355x_dev.so"), NDFFT_JIT_ROW_VEC=knob)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True, timeout=600)
        sys.stdout.write(r.stdout)
        if r.returncode:
            sys.stdout.write("CHILD FAILED knob=%s: %s\n" % (knob, r.stderr[-1500:]))
