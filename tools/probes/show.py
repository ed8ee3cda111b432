"""Prints the JSON lines of tools/bench_configs.py as a table: workload, us, fraction of 8 TB/s, path."""
import json
import sys

for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print(f"{d['workload']:66s} {d['us']:8.1f} {d.get('frac_of_8TBs', float('nan')):.3f} {d.get('path', '')}")
