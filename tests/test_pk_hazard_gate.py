"""The packed-fp32 hazard as a GATE.  `v_pk_fma_f32` reading ONE register pair through TWO source operands
with different half selects returns wrong results on MI355X while another wave of the SIMD issues bf16 / fp16 matrix instructions
(tools/pk_hazard_forms.hip, profiles/r04_pk_hazard_forms.txt).  Every unit that can run beside such instructions is
disassembled AS BUILT (the objects the library links) and must hold none; the scanner itself is checked against an object that
contains the form."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_pk_hazard as H  # noqa: E402

BAD = "v_pk_fma_f32 v[68:69], v[108:109], v[104:105], v[104:105] op_sel:[0,0,1] op_sel_hi:[1,0,1]"


def test_scanner_flags_the_failing_form_and_only_it():
    assert H.scan_text(["\t" + BAD])[2] == 1
    # the forms measured exact: distinct pairs with the same modifiers, the same pair without modifiers, mul / add with crossed selects
    assert H.scan_text(["\tv_pk_fma_f32 v[68:69], v[108:109], v[104:105], v[106:107] op_sel:[0,0,1] op_sel_hi:[1,0,1]"])[2] == 0
    assert H.scan_text(["\tv_pk_fma_f32 v[68:69], v[108:109], v[104:105], v[104:105]"])[2] == 0
    assert H.scan_text(["\tv_pk_mul_f32 v[68:69], v[104:105], v[104:105] op_sel:[0,1] op_sel_hi:[1,0]"])[1:3] == (1, 0)


def test_gate_fails_on_an_object_that_contains_the_form(tmp_path):
    """a deliberately re-introduced instance (inline asm) in a BUILT object must be counted"""
    src = tmp_path / "bad.hip"
    src.write_text('#include <hip/hip_runtime.h>\n'
                   'typedef float f2 __attribute__((ext_vector_type(2)));\n'
                   '__global__ void k(f2* p) { f2 a = p[0], b = p[1], d;\n'
                   '  asm volatile("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b));\n'
                   '  p[2] = d; }\n')
    obj = tmp_path / "bad.o"
    subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-c", str(src), "-o", str(obj)], check=True)
    total, same_pair, failing, _ = H.scan_object(str(obj))
    assert failing == 1 and same_pair == 1 and total >= 1


def test_no_unit_that_runs_beside_bf16_or_fp16_mfma_holds_the_failing_form():
    missing = [u for u in H.GATED_UNITS if not os.path.exists(os.path.join(H.CSRC, u + ".o"))]
    if missing:
        pytest.skip("library objects not built here (python __graft_entry__.py build): %s" % missing)
    res = H.gate()
    assert not any(res.values()), "failing v_pk_fma_f32 forms in built objects: %s" % {u: n for u, n in res.items() if n}
    # every conv / element-wise unit of the ST-GCN engines is in the list
    for u in ("conv_gemm_split", "conv_wgrad_split", "conv_wgrad_cn8", "conv_gemm_bf16", "conv_wgrad_bf16", "elementwise"):
        assert u in res
