#!/usr/bin/env python3
"""Copies the rocprofv3 outputs merged into gpurun_out/ (scripts/prof_bench.sh, pmc_traffic.sh,
pmc_blend.sh, bench.py) into profiles/ and derives profiles/pmc_blend_bwd.json.  Run in the build
container after the gpurun call."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def newest(pattern):
    files = glob.glob(os.path.join(G, pattern), recursive=True)
    return max(files, key=os.path.getmtime) if files else None


def avg_counters(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = "blend_fwd" if "blend_fwd" in r["Kernel_Name"] else "blend_bwd"
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


shutil.copy(newest("bench_prof/**/*kernel_stats.csv"), os.path.join(P, f"{tag}_bench_kernel_stats.csv"))
shutil.copy(os.path.join(G, "bench_r01.json"), os.path.join(P, f"{tag}_bench.json"))
out = {}
for name in ("fetch", "write", "req"):
    p = newest(f"pmc_traffic/{name}/**/*counter_collection.csv")
    shutil.copy(p, os.path.join(P, f"{tag}_pmc_{name}_counter_collection.csv"))
    for k, d in avg_counters(p).items():
        out.setdefault(k, {}).update(d)
sq = {}
for name in ("p1", "p2", "p3"):
    p = newest(f"pmc/{name}/**/*counter_collection.csv")
    shutil.copy(p, os.path.join(P, f"{tag}_pmc_sq_{name}_counter_collection.csv"))
    for k, d in avg_counters(p).items():
        sq.setdefault(k, {}).update(d)
bw = out["blend_bwd"]


def valu_busy(d):
    # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE is summed over 8 XCDs
    return round(d["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (d["GRBM_GUI_ACTIVE"] / 8), 3)


summary = {
    "mtgs": {
        "kernel": "blend_bwd_kernel<4,4>", "workload": "bench.py default (2M Gaussians, 1920x1080, variant mtgs)",
        "FETCH_SIZE_KB": round(bw["FETCH_SIZE"], 1), "WRITE_SIZE_KB": round(bw["WRITE_SIZE"], 1),
        "TCC_EA0_RDREQ_sum": round(bw["TCC_EA0_RDREQ_sum"]), "TCC_EA0_WRREQ_sum": round(bw["TCC_EA0_WRREQ_sum"]),
        "TCC_HIT_sum": round(bw["TCC_HIT_sum"]), "TCC_MISS_sum": round(bw["TCC_MISS_sum"]),
        "hbm_bytes_per_launch": int((bw["FETCH_SIZE"] + bw["WRITE_SIZE"]) * 1024),
        "SQ_INSTS_VALU": round(sq["blend_bwd"]["SQ_INSTS_VALU"]), "valu_busy_frac": valu_busy(sq["blend_bwd"]),
        "note": "FETCH_SIZE and WRITE_SIZE collected in separate --pmc passes (scripts/pmc_traffic.sh), averaged over the "
                "launches of the run, KB -> bytes x1024.  The guide's gfx950 x2 FETCH_SIZE correction is calibrated for wide "
                "coalesced streams only; this kernel's reads are 4-16 B gathers and its writes are fp32 atomics, so the "
                "value is reported UNCORRECTED.  Infinity-Cache hits are counted by these counters "
                "(MI355X_MICROARCH.md, HBM section).  valu_busy_frac = SQ_ACTIVE_INST_VALU*4/1024 / (GRBM_GUI_ACTIVE/8).",
    },
    "blend_fwd_mtgs": {"FETCH_SIZE_KB": round(out["blend_fwd"]["FETCH_SIZE"], 1),
                       "WRITE_SIZE_KB": round(out["blend_fwd"]["WRITE_SIZE"], 1),
                       "SQ_INSTS_VALU": round(sq["blend_fwd"]["SQ_INSTS_VALU"]), "valu_busy_frac": valu_busy(sq["blend_fwd"])},
}
json.dump(summary, open(os.path.join(P, "pmc_blend_bwd.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
