"""The host code only RUNS on the GPU box; here (no GPU) it is at least checked for names that are bound nowhere -- a missing
`import os` in sar_amd/ops.py once cost a GPU call (round 6).  tools/undefined_names.py is the checker."""
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_undefined_names_in_the_host_code():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import undefined_names
    files = [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    pkg = os.path.join(ROOT, "skeleton-action-recognition_amd")
    for pat in ("*.py", "sar_amd/*.py", "models/*.py", "layers/*.py", "graph/*.py"):
        files += sorted(glob.glob(os.path.join(pkg, pat)))
    files += sorted(glob.glob(os.path.join(ROOT, "tests", "*.py"))) + sorted(glob.glob(os.path.join(ROOT, "oracle", "*.py")))
    assert len(files) > 40
    assert sum(undefined_names.check(f) for f in files) == 0
