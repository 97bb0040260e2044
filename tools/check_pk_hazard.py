#!/usr/bin/env python3
"""Counts, per HIP translation unit, the packed-fp32 instructions that read ONE VGPR pair through TWO source operands with
DIFFERENT half selections (op_sel / op_sel_hi), e.g.
    v_pk_fma_f32 v[68:69], v[108:109], v[104:105], v[104:105] op_sel:[0,0,1] op_sel_hi:[1,0,1]
(hipcc's SLP vectoriser emits it for fma(aux, scale, shift) of two columns when (scale, shift) sit in one pair).
On MI355X that form returns a wrong result in ~5e-5 (beside v_mfma_f32_32x32x16_bf16) to 5e-3 (beside
v_mfma_f32_16x16x32_bf16) of its executions while ANOTHER wave of the same SIMD issues bf16 matrix instructions, and is
exact otherwise (alone, beside fp32 MFMA, with distinct pairs, or without modifiers): tools/pk_hazard_forms.hip is the
minimal reproducer, profiles/r04_pk_hazard_forms.txt its output.  (Round 2 had blamed the destination / source overlap of
the same instruction; the overlap is irrelevant.)  Usage: tools/check_pk_hazard.py [file.hip ...]; a second column counts
the round-2 pattern for comparison."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "skeleton-action-recognition_amd", "csrc")


def find_objdump():
    """llvm-objdump of the ROCm install that built the library: $ROCM_PATH, the directory of hipcc, /opt/rocm, then $PATH; None when
    there is none (the gate then WARNS instead of failing a successful build -- ADVICE r05)"""
    import shutil
    cands = []
    for root in (os.environ.get("ROCM_PATH"), os.environ.get("HIP_PATH")):
        if root:
            cands.append(os.path.join(root, "lib", "llvm", "bin", "llvm-objdump"))
    hipcc = shutil.which("hipcc")
    if hipcc:
        cands.append(os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))), "lib", "llvm", "bin", "llvm-objdump"))
    cands.append("/opt/rocm/lib/llvm/bin/llvm-objdump")
    for c in cands:
        if os.path.isfile(c) and os.access(c, os.X_OK):
            return c
    return shutil.which("llvm-objdump")


OBJDUMP = find_objdump()
PAT = re.compile(r"\s*v_pk_(fma|mul|add)_f32 (v\[\d+:\d+\]), ([^,]+), ([^,\s]+)(?:, ([^,\s]+))?(.*)")
# Units whose kernels can be resident on a SIMD while another wave issues bf16 / fp16 matrix instructions: the bf16 (CN8) engine's and
# the split engines' own units, the round-1 bf16-operand units, and every fp32 unit those engines launch beside them (element-wise
# passes, the fp32 conv kernels that keep the shapes without a split form, on the main or the weight-gradient stream).  Path B
# (conv2d / radar), ST-GIN and the dense-adjacency kernels only ever run beside fp32 MFMAs (measured exact): not gated.
# NOT gated (allow-list): units that only ever run beside fp32 MFMAs (measured exact) or hold no device arithmetic of that kind
UNGATED_UNITS = {"radar", "gin", "graph_dense", "host_io", "box_probe"}


def gated_units(csrc=CSRC):
    """every translation unit the Makefile links (SRCS) minus the allow-list: a new unit is gated by default"""
    mk = open(os.path.join(csrc, "Makefile")).read()
    m = re.search(r"^SRCS := (.*)$", mk, re.M)
    units = [u[:-4] for u in m.group(1).split() if u.endswith(".hip")]
    return [u for u in units if u not in UNGATED_UNITS]


GATED_UNITS = gated_units()

def scan_text(lines):
    """(packed fp32 ops, ONE pair through two operands with different half selects, of them v_pk_fma_f32 = the failing form,
    round-2 pattern) over disassembly / assembly lines"""
    total = hazard = same_pair = same_pair_fma = 0
    for line in lines:
        line = re.sub(r"^\s*(//|;).*$", "", line)
        m = PAT.match(line)
        if not m:
            continue
        total += 1
        dst, srcs, rest = m.group(2), [m.group(3).strip(), m.group(4).strip(), (m.group(5) or "").strip()], m.group(6)
        oh = re.search(r"op_sel_hi:\[([\d,]+)\]", rest)
        sel_hi = [int(x) for x in oh.group(1).split(",")] if oh else [1, 1, 1]
        if any(sv == dst and i < len(sel_hi) and sel_hi[i] == 0 for i, sv in enumerate(srcs)):
            hazard += 1
        ol = re.search(r"op_sel:\[([\d,]+)\]", rest)
        sel_lo = [int(x) for x in ol.group(1).split(",")] if ol else [0, 0, 0]
        sel_lo += [0] * (3 - len(sel_lo))
        sel_hi += [1] * (3 - len(sel_hi))
        n = 3 if m.group(1) == "fma" else 2
        for i in range(n):
            for j in range(i + 1, n):
                if srcs[i] and srcs[i] == srcs[j] and srcs[i].startswith("v[") and (sel_lo[i], sel_hi[i]) != (sel_lo[j], sel_hi[j]):
                    same_pair += 1
                    same_pair_fma += m.group(1) == "fma"
    return total, same_pair, same_pair_fma, hazard


def scan_object(obj):
    """the same counts over the gfx950 code object embedded in a BUILT host object (the unit exactly as the library links it):
    llvm-objdump --offloading extracts the bundle, llvm-objdump -d disassembles it; seconds per unit"""
    import shutil
    with tempfile.TemporaryDirectory() as tmp:
        o = os.path.join(tmp, os.path.basename(obj))
        shutil.copy(obj, o)
        subprocess.run([OBJDUMP, "--offloading", o], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
        co = [f for f in os.listdir(tmp) if "amdgcn" in f]
        if not co:
            raise RuntimeError("no gfx950 code object in %s" % obj)
        dis = subprocess.run([OBJDUMP, "-d", os.path.join(tmp, co[0])], check=True, capture_output=True, text=True).stdout
    return scan_text(l.split("//")[0] for l in dis.splitlines())


def gate(units=None, csrc=CSRC):
    """{unit: failing-form count} over the gated units as built (missing objects raise): the build and the CPU suite assert all zeros"""
    if OBJDUMP is None:
        print("check_pk_hazard: WARNING -- no llvm-objdump found (ROCM_PATH / hipcc / /opt/rocm / PATH): the packed-fp32 hazard gate did not run")
        return {}
    out = {}
    for u in (units or GATED_UNITS):
        out[u] = scan_object(os.path.join(csrc, u + ".o"))[2]
    return out


def main():
    # every translation unit of the library by default: the conv2d_* shims compile conv2d.hip part by part
    if sys.argv[1:] == ["--gate"]:
        res = gate()
        for u, n in res.items():
            print("%-22s failing v_pk_fma_f32 forms in the built object: %d" % (u + ".o", n))
        sys.exit(1 if any(res.values()) else 0)
    files = sys.argv[1:] or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip") and f != "conv2d.hip")
    pat = PAT  # (r"\s*v_pk_(fma|mul|add)_f32 (v\[\d+:\d+\]), ([^,]+), ([^,\s]+)(?:, ([^,\s]+))?(.*)")
    for f in files:
        noslp = os.environ.get("PK_NOSLP", "")      # comma list of TUs to scan as built with -fno-slp-vectorize ("Makefile" = as the Makefile builds them)
        mk = open(os.path.join(CSRC, "Makefile")).read()
        as_makefile = f.replace(".hip", ".o") in " ".join(l for l in mk.splitlines() if "-fno-slp-vectorize" in l)
        extra = ["-fno-slp-vectorize"] if (f in noslp.split(",") or (noslp in ("", "Makefile") and as_makefile)) else []
        if f == "radar.hip":
            extra.append("-ffp-contract=off")
        with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
            subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-I" + CSRC, "-I" + os.path.join(ROOT, "include"), *extra,
                            os.path.join(CSRC, f), "-o", tmp.name], check=True, stderr=subprocess.DEVNULL)
            total = hazard = same_pair = same_pair_fma = 0
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
                ol = re.search(r"op_sel:\[([\d,]+)\]", rest)
                sel_lo = [int(x) for x in ol.group(1).split(",")] if ol else [0, 0, 0]
                sel_lo += [0] * (3 - len(sel_lo))
                sel_hi += [1] * (3 - len(sel_hi))
                n = 3 if m.group(1) == "fma" else 2
                for i in range(n):
                    for j in range(i + 1, n):
                        if srcs[i] and srcs[i] == srcs[j] and srcs[i].startswith("v[") and (sel_lo[i], sel_hi[i]) != (sel_lo[j], sel_hi[j]):
                            same_pair += 1
                            same_pair_fma += m.group(1) == "fma"
            print("%-24s packed fp32 ops %5d | ONE pair through two operands with different half selects: %4d, of them v_pk_fma_f32 (the form "
                  "that fails beside bf16 MFMAs; the mul / add forms measured exact): %4d | (round-2 pattern: %d)" % (f, total, same_pair, same_pair_fma, hazard))


if __name__ == "__main__":
    main()
