#!/usr/bin/env python3
"""The reference's own bench matrix (/root/reference/benches/ndrustfft.rs:6-7, 9-67; ndrustfft_par.rs:9-11) on this machine, CPU beside GPU:

  fft2d  : ndfft     axis 0 of n x n Complex<f64>, n = 128, 264, 512, 1024   (fill re = im = flat index, benches/ndrustfft.rs:15-18)
  rfft2d : ndfft_r2c axis 0 of n x n f64 -> (n/2+1) x n
  dct2d  : nddct1    axis 0 of n x n f64, n = 129, 265, 513, 1025

columns (microseconds per call, median of several batches):
  cpu_serial   the CPU restatement (oracle/, scalar C) of the reference's serial form     -- NOT rustfft: a lower bound on its speed
  cpu_par      the same, `_par` form (OpenMP over lanes), threads = the CPUs the cgroup grants
  gpu_eager    device-resident arrays, one ndfft_exec_device per call from Python (ctypes): host-launch-bound below ~10 us
  gpu_graph    the same calls replayed from a HIP graph of 20 launches: what the GPU itself needs per call
  gpu_host     ndfft_exec on pageable numpy arrays (upload + transform + download)
The oracle is used here as the CPU baseline (test infrastructure, never the product).  Writes JSON lines; --md prints a markdown table.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def usable_cpus():
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, (int(q) + int(p) - 1) // int(p)))
    except Exception:
        pass
    return n


def med(f, batches=7, reps=None, budget=0.15):
    for _ in range(3):
        f()
    if reps is None:
        t0 = time.perf_counter(); f(); one = time.perf_counter() - t0
        reps = max(1, min(200, int(budget / max(one, 1e-7))))
    ts = []
    for _ in range(batches):
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        ts.append((time.perf_counter() - t0) / reps)
    return sorted(ts)[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--md", action="store_true"); ap.add_argument("--no-gpu", action="store_true")
    ap.add_argument("--host-only", action="store_true", help="only the gpu_host column (A/B runs of the host path)")
    ap.add_argument("--gpu-only", action="store_true", help="only the device-resident columns (eager, graph replay): A/B runs of kernel variants")
    a = ap.parse_args()
    os.environ.setdefault("OMP_NUM_THREADS", str(usable_cpus()))
    # libgomp's spinning waiters and a container CPU quota do not mix: with the default policy every small `_par` call here took 32 / 64 ms
    # (whole scheduler quanta) instead of ~0.4 ms
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    from oracle import oracle_ctypes as orc
    gpu = not a.no_gpu
    if gpu:
        import torch
        from ndrustfft_amd import DctHandler, FftHandler, R2cFftHandler, nddct1, ndfft, ndfft_r2c
        dev = torch.device("cuda:0")
    rows = []
    groups = [("fft2d", "ndfft", [128, 264, 512, 1024]), ("rfft2d", "ndfft_r2c", [128, 264, 512, 1024]), ("dct2d", "nddct1", [129, 265, 513, 1025])]
    for grp, op, sizes in groups:
        for n in sizes:
            i = np.arange(n * n, dtype=np.float64).reshape(n, n)
            if op == "ndfft":
                x = (i + 1j * i); y = np.zeros((n, n), np.complex128); oh = orc.FftHandler(n)
                alg = 2 * n * n * 16
            elif op == "ndfft_r2c":
                x = i.copy(); y = np.zeros((n // 2 + 1, n), np.complex128); oh = orc.R2cFftHandler(n)
                alg = n * n * 8 + (n // 2 + 1) * n * 16
            else:
                x = i.copy(); y = np.zeros((n, n)); oh = orc.DctHandler(n)
                alg = 2 * n * n * 8
            r = {"group": grp, "op": op, "n": n, "algorithmic_bytes": alg, "cpu_threads": orc.num_threads()}
            if a.host_only:
                getattr(orc, op + "_par")(x, y, oh, 0); yo = y.copy()
                h = {"ndfft": FftHandler, "ndfft_r2c": R2cFftHandler, "nddct1": DctHandler}[op](n)
                fn = {"ndfft": ndfft, "ndfft_r2c": ndfft_r2c, "nddct1": nddct1}[op]
                yh = np.zeros_like(y)
                r["gpu_host_us"] = med(lambda: fn(x, yh, h, 0), budget=0.1) * 1e6
                assert np.abs(yh - yo).max() / max(np.abs(yo).max(), 1e-300) < 1e-10
                print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in r.items()}), flush=True)
                continue
            if a.gpu_only:
                getattr(orc, op + "_par")(x, y, oh, 0); yo = y.copy()
            else:
                r["cpu_serial_us"] = med(lambda: getattr(orc, op)(x, y, oh, 0)) * 1e6
                yo = y.copy()
                r["cpu_par_us"] = med(lambda: getattr(orc, op + "_par")(x, y, oh, 0)) * 1e6
            if gpu:
                h = {"ndfft": FftHandler, "ndfft_r2c": R2cFftHandler, "nddct1": DctHandler}[op](n)
                fn = {"ndfft": ndfft, "ndfft_r2c": ndfft_r2c, "nddct1": nddct1}[op]
                xd = torch.from_numpy(x).to(dev); yd = torch.zeros(y.shape, dtype=torch.from_numpy(y).dtype, device=dev)
                fn(xd, yd, h, 0); torch.cuda.synchronize()
                from ndrustfft_amd import _lib
                r["path"] = _lib.default().last_path()
                err = np.abs(yd.cpu().numpy() - yo).max() / max(np.abs(yo).max(), 1e-300)
                assert err < 1e-10, (grp, n, err)
                def eager(k=200):
                    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(k):
                        fn(xd, yd, h, 0)
                    e1.record(); torch.cuda.synchronize()
                    return e0.elapsed_time(e1) * 1e3 / k
                eager(50)
                r["gpu_eager_us"] = sorted(eager() for _ in range(5))[2]
                # 20 calls captured in a HIP graph on a side stream, replayed
                try:
                    s = torch.cuda.Stream()
                    with torch.cuda.stream(s):
                        fn(xd, yd, h, 0)
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g, stream=s):
                            for _ in range(20):
                                fn(xd, yd, h, 0)
                    torch.cuda.synchronize()
                    def replay(k=20):
                        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(k):
                            g.replay()
                        e1.record(); torch.cuda.synchronize()
                        return e0.elapsed_time(e1) * 1e3 / (20 * k)
                    replay(5)
                    r["gpu_graph_us"] = sorted(replay() for _ in range(5))[2]
                    r["gpu_graph_frac_of_8TBs"] = alg / (r["gpu_graph_us"] * 1e-6) / 8e12
                    del g
                except Exception as ex:                       # pragma: no cover
                    r["gpu_graph_error"] = repr(ex)[:200]
                if not a.gpu_only:
                    yh = np.zeros_like(y)
                    r["gpu_host_us"] = med(lambda: fn(x, yh, h, 0), budget=0.1) * 1e6
                    assert np.abs(yh - yo).max() / max(np.abs(yo).max(), 1e-300) < 1e-10
            rows.append(r)
            print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in r.items()}), flush=True)
    if a.md and rows:
        print("\n| bench (reference) | n | CPU serial | CPU `_par` (%d threads) | GPU eager (Python) | GPU graph replay | of 8 TB/s | GPU host arrays | kernel |" % rows[0]["cpu_threads"])
        print("|---|---|---|---|---|---|---|---|---|")
        for r in rows:
            print("| %s `%s` axis 0 | %d | %.0f us | %.0f us | %.1f us | %.1f us | %.2f | %.0f us | `%s` |" % (
                r["group"], r["op"], r["n"], r["cpu_serial_us"], r["cpu_par_us"], r.get("gpu_eager_us", float("nan")),
                r.get("gpu_graph_us", float("nan")), r.get("gpu_graph_frac_of_8TBs", float("nan")), r.get("gpu_host_us", float("nan")), r.get("path", "")))


if __name__ == "__main__":
    main()
