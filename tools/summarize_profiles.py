#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of tools/profile_round.sh (gpurun_out/prof) into the tracked evidence under
profiles/: the kernel-stats CSV as collected, and a per-kernel table of launches, average duration and HBM
traffic per launch.  HBM bytes follow MI355X_MICROARCH.md section HBM: WRITE_SIZE is exact, FETCH_SIZE reports
exactly half of a coalesced streaming read on gfx950 (re-verified here on kernels with a known byte count:
sgd_nesterov, bn_add_relu_fwd, affine2), both counters are in KiB:   bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# usage: summarize_profiles.py [tag [subdir-of-gpurun_out [fp32|bf16|pathB]]]
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
P = os.path.join(ROOT, "gpurun_out", sys.argv[2] if len(sys.argv) > 2 else "prof")
MODE = sys.argv[3] if len(sys.argv) > 3 else "fp32"
EXTRA = (" " + " ".join(sys.argv[4:])) if len(sys.argv) > 4 else ""     # the extra bench.py arguments of the profiled command, if not the mode's default
BF16 = MODE == "bf16"
OUT = os.path.join(ROOT, "profiles")
os.makedirs(OUT, exist_ok=True)


def short(n):
    return n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]


def newest(pattern):
    """the most recent run only (gpurun_out accumulates the files of earlier calls)"""
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:]


def pmc(sub, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in newest(os.path.join(P, sub, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                a = agg[short(r["Kernel_Name"])]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    return agg


def held_clock():
    """{kernel: median GHz} from the GRBM_GUI_ACTIVE pass (tools/profile_round.sh step 3): counter / 8 XCDs / dispatch duration, over the
    dispatches of at least 20 us (shorter ones are dominated by the counter's start / stop granularity)"""
    cc = newest(os.path.join(P, "clock", "*", "*counter_collection.csv"))
    kt = newest(os.path.join(P, "clock", "*", "*kernel_trace.csv"))
    if not cc or not kt:
        return {}
    span = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(kt[0]))}
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(cc[0])):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
            continue
        ns = span.get(r["Dispatch_Id"])
        if ns and ns >= 20000:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]) / 8.0 / ns)
    return {k: round(sorted(v)[len(v) // 2], 3) for k, v in agg.items()}


stats = newest(os.path.join(P, "trace", "*", "*kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(OUT, "%s_bench_kernel_stats.csv" % tag))
dur = {}
for r in csv.DictReader(open(stats)):
    dur[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]), float(r["Percentage"]))
fetch, write = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
CLOCK = held_clock()
rows = []
for k, (calls, avg_ns, pct) in sorted(dur.items(), key=lambda kv: -kv[1][2]):
    f = fetch[k][1] / fetch[k][0] if fetch[k][0] else None
    w = write[k][1] / write[k][0] if write[k][0] else None
    hbm = (2 * f + w) * 1024 if f is not None and w is not None else None
    rows.append({"kernel": k, "calls": calls, "avg_us": round(avg_ns / 1e3, 2), "pct_time": pct, "clock_ghz": CLOCK.get(k),
                 "fetch_kib_raw": None if f is None else round(f, 1), "write_kib": None if w is None else round(w, 1),
                 "hbm_bytes_per_launch": None if hbm is None else int(hbm),
                 "hbm_gbps": None if hbm is None else round(hbm / avg_ns, 1)})
if BF16:   # conv_gemm_cn8_kernel<TR, TAPS = 9, ...> / the deep-prefetch variant (bf16 CN8 activations)
    fam = [r for r in rows if any(r["kernel"].startswith("conv_gemm_cn8_kernel<%d, 9" % tr) for tr in (0, 1, 2, 3))
           or r["kernel"].startswith(("conv_gemm_cn8_db_kernel<", "conv_gemm_cn8_dma_kernel<"))]
elif MODE == "pathB":
    fam = [r for r in rows if r["kernel"].startswith("conv2d_gemm_kernel")]
elif MODE == "pathB_f32_split":   # the 3x3 / stride-1 forward / data-gradient launches of conv2d_split_kernel<arith, NS, DEEP>
    fam = [r for r in rows if r["kernel"].startswith("conv2d_split_kernel<")]
elif MODE.startswith("f32_split"):   # the 9-tap temporal forward / data-gradient launches of conv_gemm_split_kernel<TR, arith, WIDE>
    fam = [r for r in rows if r["kernel"].startswith("conv_gemm_split_kernel<")]
else:
    fam = [r for r in rows if any(r["kernel"].startswith("conv_gemm_kernel<1, %d, 9" % tr) for tr in (0, 1, 2, 3))]
calls = sum(r["calls"] for r in fam)
# train steps the traced process ran: one optimizer launch per step (the fp32 headline leg appends its 100 sustained steps
# unless --sustained-steps 0 is passed)
nsteps = next((dur[k][0] for k in ("sgd_nesterov_kernel", "adam_kernel") if k in dur), 7)
import subprocess
try:
    if os.environ.get("SAR_PROFILE_COMMIT"):      # re-summarising an older trace: name the tree it was measured on
        raise KeyError
    commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    if subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "skeleton-action-recognition_amd", "bench.py"], text=True).strip():
        commit += "+uncommitted"
except KeyError:
    commit = os.environ["SAR_PROFILE_COMMIT"]
except Exception:
    commit = "unknown"
# a kernel whose launch count is not a multiple of the step count belongs to the set-up (parameter initialisation copies, table
# builds), not to the step: it stays in the list, flagged, and out of the per-step totals
for r in rows:
    r["per_step"] = (r["calls"] % nsteps == 0)
step_rows = [r for r in rows if r["per_step"]]
summary = {
    "commit": commit,          # the tree the profiled build was made from (bench.py quotes it next to roofline.traffic)
    "command": ("rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-isolated-pass --no-secondary --warm-seconds 0%s ; "
                "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 1 "
                "--no-cpu-baseline --no-isolated-pass --no-secondary --warm-seconds 0%s") % ((EXTRA or {"bf16": " --mfma bf16", "pathB": " --workload spectrogram", "f32_split": " --mfma f32_split", "f32_split_bf16x6": " --mfma f32_split_bf16x6", "pathB_f32_split": " --workload spectrogram --mfma f32_split (SAR_PATHB_GRAPH=0)"}.get(MODE, ""),) * 2),
    "hbm_rule": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts half of a coalesced stream; calibrated on "
                "sgd_nesterov / bn_add_relu_fwd / affine2 whose byte counts are known)",
    "dominant_family": {"bf16": "conv_gemm_cn8_kernel<9 taps> + conv_gemm_cn8_dma_kernel (9-tap data gradients with LDS-DMA staging)", "pathB": "conv2d_gemm_kernel (3x3 / 1x1)", "pathB_f32_split": "conv2d_split_kernel<f16x3a> (3x3 / stride 1",
                        "f32_split": "conv_gemm_split_kernel<TR, f16x3a> (9-tap temporal", "f32_split_bf16x6": "conv_gemm_split_kernel<TR, bf16x6> (9-tap temporal"}.get(
        MODE, "conv_gemm_kernel<TEMPORAL, 9 taps>") + " (forward + data-gradient instantiations)",
    "dominant_family_launches": calls,
    "dominant_family_avg_us": round(sum(r["avg_us"] * r["calls"] for r in fam) / calls, 2),
    "dominant_family_hbm_bytes_per_launch": int(sum(r["hbm_bytes_per_launch"] * r["calls"] for r in fam) / calls),
    # the clock the chip held under the dominant family (launch-weighted mean of the per-kernel medians; None without the clock pass)
    "dominant_family_clock_ghz": (round(sum(r["clock_ghz"] * r["calls"] for r in fam if r["clock_ghz"]) / max(1, sum(r["calls"] for r in fam if r["clock_ghz"])), 3)
                                  if any(r["clock_ghz"] for r in fam) else None),
    "steps_in_trace": nsteps,
    "total_kernel_ms_per_step": round(sum(r["avg_us"] * r["calls"] for r in step_rows) / (1e3 * nsteps), 3),
    "setup_kernel_ms_excluded": round(sum(r["avg_us"] * r["calls"] for r in rows if not r["per_step"]) / 1e3, 3),
    "total_hbm_bytes_per_step": int(sum((r["hbm_bytes_per_launch"] or 0) * r["calls"] for r in step_rows) / nsteps),
    "kernels": rows,
}
json.dump(summary, open(os.path.join(OUT, "%s_kernel_summary.json" % tag), "w"), indent=1)
for name in ("bench_plain.log", "bench_under_trace.log"):
    src = os.path.join(P, name)
    if os.path.exists(src):
        shutil.copy(src, os.path.join(OUT, "%s_%s" % (tag, name)))
print("dominant family: %d launches, avg %.1f us, %.1f MB HBM per launch" %
      (calls, summary["dominant_family_avg_us"], summary["dominant_family_hbm_bytes_per_launch"] / 1e6))
for r in rows[:12]:
    print("%-62s %4d x %9.1f us  %5.1f%%  %s" % (r["kernel"][:62], r["calls"], r["avg_us"], r["pct_time"],
          "-" if r["hbm_bytes_per_launch"] is None else "%.0f MB/launch %.0f GB/s" % (r["hbm_bytes_per_launch"] / 1e6, r["hbm_gbps"])))
