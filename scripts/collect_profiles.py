#!/usr/bin/env python3
"""Copies the rocprofv3 outputs merged into gpurun_out/ (scripts/prof_bench.sh, scripts/pmc_step.sh) into profiles/ and
derives profiles/<round>_pmc_step.json: per kernel of the benchmark step the counter-based HBM traffic (FETCH_SIZE and
WRITE_SIZE from SEPARATE --pmc passes, corrected by the factors measured on kernels with a known byte count), the
cycles per VALU instruction per SIMD with the resident waves and wait fractions, and the average duration from the kernel trace
of the same command.  Run in the build container after
the gpurun calls:   python scripts/collect_profiles.py r02"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def newest(pattern):
    files = glob.glob(os.path.join(G, pattern), recursive=True)
    return max(files, key=os.path.getmtime) if files else None


def per_kernel(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}, {k: len(next(iter(d.values()))) for k, d in agg.items()}


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


stats = newest("prof_bench/**/*kernel_stats.csv")
shutil.copy(stats, os.path.join(P, f"{tag}_bench_kernel_stats.csv"))
avg_us = {r["Name"]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(stats))}
sys.path.insert(0, ROOT)
from mtgs_amd import _lib  # noqa: E402  (only for the ABI version the profiles were taken with)
out = {"abi_version": _lib.ABI_VERSION, "hot_abi_version": _lib.HOT_ABI_VERSION,
       "lists": "tight" if os.environ.get("MTGS_TIGHT_LISTS", "0") == "1" else "gsplat", "_about": "scripts/pmc_step.sh on the MI355X box: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ counters in SEPARATE passes "
                 "over `bench.py --steps 6 --warmup 2 --cpu-steps 0` (headline workload), per-kernel averages over the launches of "
                 "the run; avg_us from the rocprofv3 --kernel-trace --stats run of scripts/prof_bench.sh.  Counters are KB: bytes = "
                 "value x 1024 x correction.",
       "calibration": {}}
# ---- calibration on kernels that move exactly 384 MiB (scripts/dev/read_bench.hip --calibrate)
true_bytes = 384 << 20
for c, kernels in (("FETCH_SIZE", ("rd<HIP_vector_type<float, 4u>, 16>", "rd<float, 16>")), ("WRITE_SIZE", ("wr(",))):
    p = newest(f"pmc_step/cal_{c}/**/*counter_collection.csv")
    shutil.copy(p, os.path.join(P, f"{tag}_pmc_cal_{c}_counter_collection.csv"))
    vals, _ = per_kernel(p)
    for k, d in vals.items():
        if any(x in k for x in kernels):
            out["calibration"][f"{c} {short(k)}"] = {"counter_KB": d[c], "true_bytes": true_bytes,
                                                     "true_over_counter": round(true_bytes / (d[c] * 1024), 4)}
fetch_corr = round(sum(v["true_over_counter"] for k, v in out["calibration"].items() if k.startswith("FETCH")) /
                   max(sum(1 for k in out["calibration"] if k.startswith("FETCH")), 1), 3)
write_corr = round(sum(v["true_over_counter"] for k, v in out["calibration"].items() if k.startswith("WRITE")) /
                   max(sum(1 for k in out["calibration"] if k.startswith("WRITE")), 1), 3)
out["corrections"] = {"FETCH_SIZE": fetch_corr, "WRITE_SIZE": write_corr,
                      "note": "the guide's gfx950 note (FETCH_SIZE reports half of a coalesced streaming read) verified here for 16-byte and "
                              "4-byte per-lane loads; WRITE_SIZE is exact for 16-byte streaming stores.  Gather-dominated kernels (compositing) "
                              "are reported with the same factors: an upper bound on their fetch traffic if partial-line requests are "
                              "tallied differently."}
F, nF = per_kernel(newest("pmc_step/FETCH_SIZE/**/*counter_collection.csv"))
W, _ = per_kernel(newest("pmc_step/WRITE_SIZE/**/*counter_collection.csv"))
S, _ = per_kernel(newest("pmc_step/sq/**/*counter_collection.csv"))
for name in ("FETCH_SIZE", "WRITE_SIZE", "sq"):
    shutil.copy(newest(f"pmc_step/{name}/**/*counter_collection.csv"), os.path.join(P, f"{tag}_pmc_step_{name}_counter_collection.csv"))
kernels = {}
for k in F:
    if "at::native" in k or "rocclr" in k or "rocsolver" in k or "elementwise_kernel_with_index" in k:
        continue   # the library's kernels only (PyTorch's glue kernels stay in the CSVs)
    fb = F[k]["FETCH_SIZE"] * 1024 * fetch_corr
    wb = W.get(k, {}).get("WRITE_SIZE", 0.0) * 1024 * write_corr
    rec = {"launches_in_pmc_run": nF[k], "fetch_bytes": int(fb), "write_bytes": int(wb), "hbm_bytes": int(fb + wb),
           "avg_us": round(avg_us.get(k, 0.0), 2)}
    if rec["avg_us"] > 0:
        rec["counter_GBs"] = round((fb + wb) / rec["avg_us"] / 1e3, 1)
        rec["frac_of_8TBs"] = round((fb + wb) / rec["avg_us"] / 1e3 / 8000.0, 4)
    s = S.get(k)
    if s and s.get("GRBM_GUI_ACTIVE") and s.get("SQ_INSTS_VALU"):
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_* over the chip's 1024 SIMDs.  SQ_ACTIVE_INST_VALU is an instruction count
        # weighted per opcode class (1: plain / DPP, 2: transcendentals, permlane swaps), NOT busy cycles (profiles/r06_valu_ceiling.md)
        cyc = s["GRBM_GUI_ACTIVE"] / 8
        rec["SQ_INSTS_VALU"] = int(s["SQ_INSTS_VALU"])
        rec["gpu_cycles"] = int(cyc)
        rec["cycles_per_valu_inst_per_simd"] = round(cyc * 1024 / s["SQ_INSTS_VALU"], 3)
        rec["active_over_insts_valu"] = round(s["SQ_ACTIVE_INST_VALU"] / s["SQ_INSTS_VALU"], 3)
        if s.get("SQ_WAVE_CYCLES"):
            rec["mean_resident_waves_per_simd"] = round(s["SQ_WAVE_CYCLES"] * 4 / cyc / 1024, 2)
            rec["wait_any_frac"] = round(s.get("SQ_WAIT_ANY", 0.0) / s["SQ_WAVE_CYCLES"], 3)
            rec["wait_inst_any_frac"] = round(s.get("SQ_WAIT_INST_ANY", 0.0) / s["SQ_WAVE_CYCLES"], 3)
        if s.get("SQ_INSTS_SALU") is not None:
            rec["salu_per_valu"] = round(s["SQ_INSTS_SALU"] / s["SQ_INSTS_VALU"], 3)
    kernels[short(k)] = rec
out["kernels"] = dict(sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes"]))
json.dump(out, open(os.path.join(P, f"{tag}_pmc_step.json"), "w"), indent=1)
b = newest("prof_bench/bench.log")
if b:
    lines = [ln for ln in open(b) if ln.startswith("{")]
    if lines:
        open(os.path.join(P, f"{tag}_bench_under_rocprof.json"), "w").write(lines[-1])
print(json.dumps({"corrections": out["corrections"], "kernels": {k: (v["hbm_bytes"], v.get("avg_us"), v.get("frac_of_8TBs"))
                                                                  for k, v in list(out["kernels"].items())[:14]}}, indent=1))
