#!/usr/bin/env python3
"""Compact per-kernel register / scratch / occupancy table for the gfx950 build (developer tool).

  python tools/kernel_resources.py [file.hip ...]      (default: every kernels_*.hip, transpose.hip, big.hip)

Uses clang's -Rpass-analysis=kernel-resource-usage remarks; nothing is linked or run.
"""
import glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "ndrustfft_amd", "csrc")
files = [os.path.abspath(a) for a in sys.argv[1:]] or sorted(glob.glob(os.path.join(SRC, "kernels_*.hip"))) + [os.path.join(SRC, f) for f in ("transpose.hip", "big.hip")]
for f in files:
    out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast", "--cuda-device-only",
                          "-c", f, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, cwd=SRC).stderr
    cur = None
    rows = []
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            name = name.replace("ndfft::", "").replace("void ", "")
            cur = {"name": name}; rows.append(cur); continue
        for key, pat in (("vgpr", r"\bVGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    print(f"== {os.path.basename(f)}")
    for r in rows:
        print(f"  vgpr {r.get('vgpr', 0):3d} agpr {r.get('agpr', 0):3d} scratch {r.get('scratch', 0):4d} occ {r.get('occ', 0):2d}  {r['name'][:150]}")
