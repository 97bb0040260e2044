#!/usr/bin/env python3
"""Counts, per HIP translation unit, the packed-fp32 instructions whose destination register pair is also a source pair
AND whose high half reads the low register of that pair (op_sel_hi = 0): the pattern behind the intermittent ReLU-mask
flips of conv_gemm_cn8's MASK epilogue on MI355X (csrc/Makefile).  Usage: tools/check_pk_hazard.py [file.hip ...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "skeleton-action-recognition_amd", "csrc")
# every translation unit of the library by default: the conv2d_* shims compile conv2d.hip part by part
files = sys.argv[1:] or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip") and f != "conv2d.hip")
pat = re.compile(r"\s*v_pk_(fma|mul|add)_f32 (v\[\d+:\d+\]), ([^,]+), ([^,\s]+)(?:, ([^,\s]+))?(.*)")
for f in files:
    extra = ["-fno-slp-vectorize"] if f == "conv_gemm_cn8.hip" else (["-ffp-contract=off"] if f == "radar.hip" else [])
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-I" + CSRC, *extra,
                        os.path.join(CSRC, f), "-o", tmp.name], check=True, stderr=subprocess.DEVNULL)
        total = hazard = 0
        for line in open(tmp.name):
            m = pat.match(line)
            if not m:
                continue
            total += 1
            dst, srcs, rest = m.group(2), [m.group(3).strip(), m.group(4).strip(), (m.group(5) or "").strip()], m.group(6)
            oh = re.search(r"op_sel_hi:\[([\d,]+)\]", rest)
            sel_hi = [int(x) for x in oh.group(1).split(",")] if oh else [1, 1, 1]
            if any(sv == dst and i < len(sel_hi) and sel_hi[i] == 0 for i, sv in enumerate(srcs)):
                hazard += 1
        print("%-24s packed fp32 ops %5d, high half reads the low register of its own destination: %d" % (f, total, hazard))
