"""Markdown tables of DESIGN.md section 6 from tools/bench_configs.py output: mk_tables.py <configs.jsonl> <configs_cold.jsonl> <landscape.jsonl>"""
import json
import re
import sys


def load(p):
    return [json.loads(l) for l in open(p) if l.startswith("{")]


warm, cold, land = load(sys.argv[1]), load(sys.argv[2]), load(sys.argv[3])
coldmap = {d["workload"].split(" [cold")[0]: d for d in cold}
print("| workload | path | µs | GFFT-pts/s | of 8 TB/s (same buffers every launch) | cache-cold: 6 rotating (in, out) pairs |")
print("|---|---|---|---|---|---|")
for d in warm:
    w = d["workload"]
    if w.startswith(("primes[", "small ", "refbench", "hostpath", "pow2 ")):
        continue
    c = coldmap.get(w)
    cs = f"{c['us']:.1f} µs, {100 * c['frac_of_8TBs']:.1f} %" if c else ""
    fr = d.get("frac_of_8TBs")
    print(f"| {w} | `{d.get('path', '')}` | {d['us']:.1f} | {d.get('GFFT-points/s', 0):.1f} | {100 * fr:.1f} % | {cs} |" if fr else f"| {w} | | {d['us']:.1f} | | | |")
print()
cols = ("ndfft c128", "ndfft c64", "nddct2 f64", "ndfft_r2c f32")
tab = {}
for d in land:
    m = re.match(r"landscape (.*) n=(\d+)", d["workload"])
    tab.setdefault(int(m.group(2)), {})[m.group(1)] = d
print("| n | `ndfft` c128 | `ndfft` c64 | `nddct2` f64 | `ndfft_r2c` f32 |")
print("|---|---|---|---|---|")
for n in sorted(tab):
    print(f"| {n} | " + " | ".join(f"{100 * tab[n][c]['frac_of_8TBs']:.0f} % `{tab[n][c]['path']}`" for c in cols) + " |")
