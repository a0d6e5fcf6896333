#!/usr/bin/env python3
"""Rewrite the "Other BASELINE configs" table of DESIGN.md from profiles/r01_configs.jsonl (output of tools/bench_configs.py)."""
import json
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
rows = [json.loads(l) for l in (ROOT / "profiles" / "r01_configs.jsonl").read_text().splitlines() if l.strip()]
lines = ["| config | GPU (device-resident entry) | per-kernel avg launch (ms) | CPU port, 1 core |", "|---|---|---|---|"]
for r in rows:
    unit = r["unit"]
    cb = r["cpu_baseline"]
    cu = cb["unit"]
    extra = cu[len(unit):] if cu.startswith(unit) else " " + cu
    lines.append(f"| {r['config']} | {r['value']:,.0f} {unit} ({r['batch']}) | " + ", ".join(f"{k} {v}" for k, v in r["kernels_ms"].items())
                 + f" | {cb['value']:,.0f} {unit}{extra} |")
p = ROOT / "DESIGN.md"
s = p.read_text()
s2 = re.sub(r"\| config \| GPU \(device-resident entry\).*?\n\n", "\n".join(lines) + "\n\n", s, count=1, flags=re.S)
p.write_text(s2)
print("rows:", len(rows), "changed:", s != s2)
