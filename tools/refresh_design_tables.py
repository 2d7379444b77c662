#!/usr/bin/env python3
"""Rewrite the whole-step table of DESIGN.md section 0 (between the STEP_TABLE markers) from profiles/<round>_bench_default.json
and, in brackets, a second run.  The per-kernel table above it is edited by hand from `tools/state_table.py <round>`.

    tools/refresh_design_tables.py r03 [second_bench.json]"""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r04"
first = os.path.join(REPO, "profiles", f"{R}_bench_default.json")
second = sys.argv[2] if len(sys.argv) > 2 else os.path.join(REPO, "profiles", f"{R}_bench_default_second_box.json")
d, e = json.load(open(first)), json.load(open(second))
NAMES = {"quant_c2": "quant 1 M × 100 (config 2)", "compare_c3": "compare 1 M × (50 v 50) (config 3)",
         "pairwise_c4_shard": "pairwise 25 k × 200 (config-4 shard: sums + Fisher + BH per column)",
         "e2e_c5_shard": "quant → compare 625 k × 1000 (config-5 shard)",
         "pairwise_c4_full": "pairwise 200 k × 200, FULL config 4 on one GPU", "e2e_c5_full": "quant → compare 5 M × 1000, FULL config 5 on one GPU"}
R2 = {"headline": "4.9–5.25 × 10¹¹, 1.90–2.04 ms", "quant_c2": "3.05–3.12 × 10¹¹, 0.32–0.33 ms", "compare_c3": "2.7 × 10⁹, 0.37 ms",
      "pairwise_c4_shard": "1.35–1.37 × 10¹⁰, 36–37 ms", "e2e_c5_shard": "2.07–2.2 × 10¹¹, 2.84–3.02 ms",
      "pairwise_c4_full": "— (never run)", "e2e_c5_full": "— (never run)"}
rows = [("headline", d, e)] + [(k, v, e["also"][k]) for k, v in d["also"].items()]
out = ["| record | value (a second run on another box) | ms per step | dominant kernel: ms, frac of 8 TB/s | CPU port, 1 core | round 3 |",
       "|---|---|---|---|---|---|"]
for k, v, w in rows:
    ro, cb = v["roofline"], v.get("cpu_baseline") or {}
    out.append(f"| {NAMES.get(k, 'quant 2 M × 500 (headline)')} | **{v['value']:.3g}** ({w['value']:.3g}) {v['unit']} | {v['ms_per_step']:.4g} ({w['ms_per_step']:.4g}) | "
               f"`{ro['kernel']}` {ro['avg_kernel_ms']:.4g} ms, {ro['frac']:.3f} ({w['roofline']['avg_kernel_ms']:.4g} ms, {w['roofline']['frac']:.3f}) | "
               f"{cb.get('value', float('nan')):.2g} | {R2[k]} |".replace("| nan |", "| (as the shard) |"))
p = os.path.join(REPO, "DESIGN.md")
s = open(p).read()
a, b = s.index("<!-- STEP_TABLE -->"), s.index("<!-- /STEP_TABLE -->")
s = s[:a] + "<!-- STEP_TABLE -->\n" + "\n".join(out) + "\n" + s[b:]
open(p, "w").write(s)
print("\n".join(out))
