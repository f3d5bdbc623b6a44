"""Prints value (Gbit/s), ms per step, per-kernel times and the parity note of a bench.py JSON line: python tools/show_bench.py FILE"""
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(j["value"] / 1e9, j["ms_per_step"], j["kernel_ms"], j.get("parity_check"))
