"""nddct1 n = 512 (F = 511 = 7 x 73) on the symmetric Rader kernel (rader_kernel.h: SYM): sweep of FFT_72 recipes and lanes per workgroup.
Developer build only (NDFFT_MI355X_LIB=ndrustfft_amd/csrc/libndfft_mi355x_dev.so): NDFFT_RADER_CFG / NDFFT_RADER_LPB / NDFFT_RADER_SYM are read per plan."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import synth
from ndrustfft_amd import DctHandler, nddct1, _lib
from bench_configs import timeit
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
x = torch.from_numpy(synth.real_array((65536 * 512 // n, n))).to(dev); y = torch.empty_like(x)
ref = None
cfgs = [("1", None, 0)]     # the planner's own choice
for sym in ("1", "0"):
    for cfg in ("5:9.8", "8:9.8", "8:8.9", "9:9.8", "9:8.9", "16:9.8", "12:12.6", "12:6.12", "6:12.6", "18:4.6.3", "18:6.4.3", "24:3.4.6", "12:6.4.3", "4:9.8", "3:9.8", "9:8.3.3", "8:9.4.2"):
        for lpb in (0, 1, 2, 3, 4, 6, 8):
            cfgs.append((sym, cfg, lpb))
for sym, cfg, lpb in cfgs:
    os.environ["NDFFT_RADER_SYM"] = sym
    if cfg: os.environ["NDFFT_RADER_CFG"] = cfg
    else: os.environ.pop("NDFFT_RADER_CFG", None)
    if lpb: os.environ["NDFFT_RADER_LPB"] = str(lpb)
    else: os.environ.pop("NDFFT_RADER_LPB", None)
    try:
        h = DctHandler(n)
        nddct1(x, y, h, 1)
        torch.cuda.synchronize()
        path = _lib.default().last_path()
        if path != "rader_reg":
            print(f"sym={sym} cfg={cfg} lpb={lpb}: path {path}", flush=True); continue
        if ref is None:
            ref = y.clone()
        err = float((y - ref).abs().max() / ref.abs().max())
        t = timeit(lambda: nddct1(x, y, h, 1), 20, ramp_ms=60)
        print(f"sym={sym} cfg={cfg} lpb={lpb}: {t*1e6:8.1f} us  err_vs_first={err:.1e}", flush=True)
    except Exception as ex:
        print(f"sym={sym} cfg={cfg} lpb={lpb}: {str(ex)[:80]}", flush=True)
