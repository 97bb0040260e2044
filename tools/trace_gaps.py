"""GPU busy / idle time per train step from a rocprofv3 kernel trace (gpurun_out/<sub>/trace): the union of all kernel intervals between
two optimizer launches, per stream (Queue_Id) and overall.  Usage: python tools/trace_gaps.py <subdir-of-gpurun_out> [optimizer-kernel-substring]"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sub = sys.argv[1]
opt = sys.argv[2] if len(sys.argv) > 2 else "sgd_nesterov_kernel"
f = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", sub, "trace", "*", "*kernel_trace.csv")), key=os.path.getmtime)[-1]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")) for r in csv.DictReader(open(f)))
marks = [i for i, r in enumerate(rows) if opt in r[2]]


def union(seg):
    busy, cs, ce = 0, seg[0][0], seg[0][1]
    for s, e, *_ in seg[1:]:
        if s > ce:
            busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return busy + ce - cs


for a, b in list(zip(marks[:-1], marks[1:]))[-3:]:
    seg = rows[a + 1:b + 1]
    t0, t1 = seg[0][0], seg[-1][1]
    per_q = {}
    for r in seg:
        per_q.setdefault(r[3], []).append(r)
    print("step %.3f ms: busy(any) %.3f, idle %.3f, kernels %d | per queue: %s" % (
        (t1 - t0) / 1e6, union(seg) / 1e6, (t1 - t0 - union(seg)) / 1e6, len(seg),
        ", ".join("q%s %.2f ms (%d)" % (q, union(v) / 1e6, len(v)) for q, v in sorted(per_q.items()))))
