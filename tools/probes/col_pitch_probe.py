#!/usr/bin/env python3
"""Probe (round 6): does the row pitch of a small column call decide its time?  ndfft axis 0 of (1024, W) c128 and nddct1 axis 0 of (1025, W) f64 for several W,
each replayed from a HIP graph of 20 launches.  A power-of-two pitch (W = 1024 c128: 16384 B) against nearby pitches shows channel camping if there is any."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ndrustfft_amd import DctHandler, FftHandler, nddct1, ndfft, _lib
dev = torch.device("cuda:0")


def graph_us(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(20):
                fn()
    torch.cuda.synchronize()
    def replay(k=20):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / (20 * k)
    replay(10)
    return sorted(replay() for _ in range(7))[3]


ONLY = sys.argv[1] if len(sys.argv) > 1 else ""      # e.g. "nddct1:1025"
for op, n, Ws in (("ndfft", 1024, (512, 768, 1000, 1024, 1028, 1032, 1040, 1056, 1088, 1536, 2048)), ("nddct1", 1025, (512, 768, 1000, 1024, 1025, 1032, 1040, 1056, 1088, 1536, 2048)),
                  ("ndfft", 512, (512, 520, 1024)), ("nddct1", 513, (513, 520, 1026))):
    for W in Ws:
        if ONLY and ONLY != "%s:%d" % (op, n):
            continue
        if op == "ndfft":
            x = torch.randn(n, W, dtype=torch.complex128, device=dev); h = FftHandler(n); f = ndfft
        else:
            x = torch.randn(n, W, dtype=torch.float64, device=dev); h = DctHandler(n); f = nddct1
        y = torch.empty_like(x)
        us = graph_us(lambda: f(x, y, h, 0))
        print(json.dumps({"op": op, "n": n, "W": W, "pitch_bytes": W * x.element_size(), "graph_us": round(us, 2), "ns_per_lane": round(us * 1e3 / W, 2), "path": _lib.default().last_path()}), flush=True)
