#!/usr/bin/env python3
"""Print the section-0 table of DESIGN.md from profiles/<round>_*: kernel, workload, ms (rocprofv3 average), fraction of
8 TB/s, PMC traffic / algorithmic bytes.   tools/state_table.py r04"""
import csv, json, os, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r04"
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
traffic = {(e["workload"], e["n"], e["s"]): e for e in json.load(open(os.path.join(P, "pmc_traffic.json")))}
ROWS = [("quant", "quant", 2_000_000, 500, "ps_tile_v3", lambda n, s: 8.0 * n * s, "quant 2 M x 500"),
        ("quantc2", "quant", 1_000_000, 100, "ps_tile_v3", lambda n, s: 8.0 * n * s, "quant 1 M x 100"),
        ("compare", "compare", 1_000_000, 100, "ranksum_pairq", lambda n, s: (4.0 * s + 28) * n, "compare 1 M x (50 v 50)"),
        ("e2e", "e2e", 625_000, 1000, "ranksum_count", lambda n, s: (4.0 * s + 28) * n, "e2e 625 k x (500 v 500)"),
        ("pairwise", "pairwise", 25_000, 200, "fisher_pairs_kernel<16, false>", lambda n, s: 8.0 * n * s * (s - 1) / 2 + 12.0 * n * s, "pairwise 25 k x 200")]
for tag, wl, n, s, kern, alg, label in ROWS:
    rows = list(csv.DictReader(open(os.path.join(P, f"{R}_{tag}_kernel_stats.csv"))))
    k = next(r for r in rows if r["kernel"].startswith(kern))
    ms = float(k["avg_ns"]) / 1e6
    a = alg(n, s)
    t = traffic.get((wl, n, s))
    tr = f"{t['hbm_bytes_per_launch'] / a:.2f} x ({t['hbm_bytes_per_launch'] / 1e9:.2f} GB)" if t else "-"
    print(f"| `{k['kernel']}` | {label} | {ms:.4g} (min {float(k['min_ns']) / 1e6:.4g}) | {a / ms / 1e6 / 8000:.3f} | {tr} | `profiles/{R}_{tag}_kernel_stats.csv`, `_pmc.csv` |")
    others = [(r["kernel"], float(r["avg_ns"]) / 1e3, int(r["calls"])) for r in rows[:14] if r is not k]
    print("   others:", ", ".join(f"{nm.split('<')[0]} {us:.1f}us x{c}" for nm, us, c in others))
