"""Per-kernel-family table of one engine's train step from bench.py --detail (HIP events around every ops region):
    python tools/step_table.py [bench.py args ...]      e.g.  SAR_WGRAD_STREAM=0 python tools/step_table.py --mfma f32_split
prints ms/step, TFLOP/s and GB/s (algorithmic) per family, sorted, and the sum -- with the weight-gradient stream off the sum is the
step's GPU time and each row is that family's own time (nothing overlaps)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:] or ["--mfma", "f32_split"]
r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-secondary", "--no-cpu-baseline",
                    "--no-isolated-pass", "--detail", "--warm-seconds", "2", "--sustained-steps", "0"] + args, capture_output=True, text=True)
line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
if not line:
    sys.exit(r.stderr[-3000:])
d = json.loads(line[-1])
k = d["detail"]["headline"]["kernel_ms_per_step"]
tf = d["detail"]["headline"]["kernel_tflops"]
print("%s: %.2f clips/s, %.3f ms/step (SAR_WGRAD_STREAM=%s)" % (" ".join(args), d["value"], d["ms_per_step"], os.environ.get("SAR_WGRAD_STREAM", "1")))
tot = 0.0
for name, ms in sorted(k.items(), key=lambda kv: -kv[1]):
    print("  %-34s %7.3f ms  %8.1f TF" % (name, ms, tf.get(name, 0.0)))
    tot += ms
print("  %-34s %7.3f ms" % ("sum of bracketed regions", tot))
