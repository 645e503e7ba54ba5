"""Manual probe (not a test): per-kernel eager times, host-driven (2) vs device-side (1) re-bin decision."""
import json, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wl = sys.argv[1] if len(sys.argv) > 1 else "C5"
steps = sys.argv[2] if len(sys.argv) > 2 else "40"
for d in ("2", "1"):
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", wl, "--dynamic", d, "--steps", steps, "--warmup", "40",
                          "--no-cpu-baseline", "--no-aux", "--profile-steps", "20"], capture_output=True, text=True).stdout
    r = json.loads(out)
    print("dynamic", d, "us/step %.1f" % (r["ms_per_step"] * 1e3), {k: round(v * 1e3, 1) for k, v in r["kernels_ms"].items()}, flush=True)
