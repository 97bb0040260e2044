#!/bin/bash
# Per-kernel durations of a train step with the weight-gradient stream OFF (SAR_WGRAD_STREAM=0: every kernel runs alone) from a
# rocprofv3 kernel trace of bench.py: launches per step, average duration, ms per step, and their sum -- next to the two-stream step time.
#   gpurun -- tools/isolated_step.sh bf16 > profiles/r06_bf16_isolated_kernels.txt        (modes: fp32 | bf16 | f32_split)
R=${GRAFT_REPO_ROOT:-/root/repo}
MODE=${1:-bf16}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/iso_$MODE
export SAR_WGRAD_STREAM=0
rocprofv3 --kernel-trace --output-format csv -d /tmp/iso_$MODE -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-isolated-pass --no-secondary --warm-seconds 0 --sustained-steps 0 --mfma $MODE > /tmp/iso_$MODE.log 2>&1
python3 - $MODE <<'PY'
import collections, csv, glob, json, sys
mode = sys.argv[1]
f = glob.glob("/tmp/iso_%s/**/*kernel_trace.csv" % mode, recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
opt = [i for i, r in enumerate(rows) if "sgd_nesterov" in r["Kernel_Name"]]
lo, hi = opt[1], opt[-1]          # whole steps between the second and the last optimizer launch
steps = len(opt) - 2
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows[lo + 1:hi + 1]:
    n = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
    a = agg[n]
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = 0.0
for n, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-60s %5.1f/step avg %8.1f us %8.3f ms/step" % (n[:60], c / steps, us / c, us / steps / 1e3))
    tot += us / steps / 1e3
js = [l for l in open("/tmp/iso_%s.log" % mode).read().splitlines() if l.startswith("{")]
ms = json.loads(js[-1])["ms_per_step"] if js else float("nan")
print("TOTAL kernel ms/step, every kernel alone (one stream, under the tracer): %.3f   (step time of this run, one stream, traced: %.3f ms; %d steps)" % (tot, ms, steps))
PY
