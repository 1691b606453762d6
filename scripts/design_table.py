#!/usr/bin/env python3
"""Regenerates the per-kernel table of DESIGN.md section 4 (between the KERNEL_TABLE markers) from the round's evidence:
profiles/<tag>_bench.json (bench.py's line: algorithmic bytes of this run, the committed counters echoed per kernel, what binds) and
profiles/<tag>_bench_kernel_stats.csv (rocprofv3 --kernel-trace --stats).   python scripts/design_table.py r06"""
import csv
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
line = json.loads([ln for ln in (ROOT / "profiles" / f"{tag}_bench.json").read_text().splitlines() if ln.startswith("{")][-1])
stats = {}
for r in csv.DictReader(open(ROOT / "profiles" / f"{tag}_bench_kernel_stats.csv")):
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    stats[name] = float(r["AverageNs"]) / 1e3
files = {"sh_": "sh.hip", "vis_color": "viscolor.hip", "front_": "front.hip", "project_bwd": "project_bwd.hip", "bin3_": "bin3.hip", "blend_": "blend.hip"}
rows = []
for k in line["roofline"]["kernels"]:
    name = k["kernel"]
    us = next((v for n, v in stats.items() if n == name or n.startswith(name + "<")), k.get("avg_us_committed"))
    f = next((v for p, v in files.items() if name.startswith(p)), "")
    cpi, ceil_ = k.get("cycles_per_valu_inst_committed"), k.get("valu_ceiling_cycles_per_inst")
    rows.append((us or 0.0, f"| `{name}` (`{f}`) | {k['algorithmic_bytes'] / 1e6:.0f} MB | {us:.1f} | "
                 f"{k.get('frac_of_hbm_peak_counter', 0) or 0:.2f} ({k['counter_bytes_committed'] / 1e6:.0f} MB) | "
                 + (f"{cpi:.2f} ({(ceil_ / cpi):.2f} of {ceil_:.2f})" if cpi and ceil_ else "-") + f" | {k.get('bound') or '-'} |"))
split = next((e.get("algorithmic_bytes_split") for e in line["roofline"]["entry_points"] if e["entry_point"] == "mtgs_blend_fwd_packed"), None)
if split:      # the zeros that ride on the compositing forward are not compositing bytes (SURVEY 8(d)'s unit)
    rows = [(u, r.replace(" MB | ", f" MB ({split['compositing'] / 1e6:.0f} compositing + {split['riding_zeros'] / 1e6:.0f} riding zeros) | ", 1)
             if "`blend_fwd_kernel" in r else r) for u, r in rows]
rows.sort(key=lambda r: -r[0])
head = ("| kernel (file) | algorithmic HBM bytes | avg µs | of HBM peak (counter bytes) | cycles / VALU inst (of its ceiling) | bound |\n"
        "|---|---|---|---|---|---|\n")
ws = line["roofline"]["whole_step"]
foot = (f"\nWhole step: **{line['value']:.0f} Mpix/s, {line['ms_per_step']:.3f} ms** ({line['config']['launch'].split(':')[0]}); "
        f"{ws['algorithmic_bytes'] / 1e6:.0f} MB algorithmic (SURVEY §8(d) B_F + B_B) = {ws['achieved'] / 1e3:.2f} TB/s = {ws['frac']:.3f} of the HBM peak; "
        f"`cpu_baseline` {line.get('cpu_baseline', {}).get('value', '-')} Mpix/s on {line.get('cpu_baseline', {}).get('cores', '-')} host threads "
        f"(`profiles/{tag}_bench.json`).  Entry points live (HIP events): "
        + ", ".join(f"`{e['entry_point']}` {e['avg_us_live']:.0f}" for e in line["roofline"]["entry_points"]) + " µs.\n")
p = ROOT / "DESIGN.md"
t = p.read_text()
t = re.sub(r"<!-- KERNEL_TABLE_BEGIN -->.*?<!-- KERNEL_TABLE_END -->",
           "<!-- KERNEL_TABLE_BEGIN -->\n" + head + "\n".join(r[1] for r in rows) + "\n" + foot + "<!-- KERNEL_TABLE_END -->", t, flags=re.S)
p.write_text(t)
print(head + "\n".join(r[1] for r in rows) + foot)
