"""Manual probe (not a test): one workload under several SPHX_DEBUG_SWITCHES settings, alternating, each in a fresh process.
    python tools/probes/probe_ab_switches.py C5 100 40 1000 300 3 "" no_drift_top2 no_sched_redirect ...
args: workload, timed steps, warm-up, steps skipped before the sustained window (0: none), its length, repetitions, then the
switch sets ("" = none; "switches@tools/_exp/libsphx_r3.so" runs an older build of the library).  Prints the window's and the sustained us/step, the drift-triggered re-binnings and (last repetition)
per-kernel times."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wl, steps, warm, skip, nsus, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
variants = sys.argv[7:] or [""]
code = r'''
import importlib, json, sys
sys.path.insert(0, %r)
import bench
pkg = importlib.import_module(bench.PKG)
name, kw = bench.parse_workload(%r)
r = bench.run_case(pkg.capi, pkg.config, pkg.geometry, name, kw, %d, %d, %d, sustained=%s)[0]
print(json.dumps({k: r[k] for k in ("ms_per_step", "kernels_ms", "sustained", "tuning")}))
'''
for rep in range(reps):
    for v in variants:
        prof = 16 if rep == reps - 1 else 0
        src = code % (root, wl, steps, warm, prof, repr((skip, nsus)) if skip > 0 else "None")
        sw, _, lib = v.partition("@")  # "switches@path/to/libsphx.so": an older build (tools/build_baseline_lib.sh) in the same alternation
        env = dict(os.environ, SPHX_DEBUG_SWITCHES=sw)
        if lib:
            env["SPHX_LIB"] = os.path.join(root, lib)
        p = subprocess.run([sys.executable, "-c", src], env=env, capture_output=True, text=True)
        try:
            r = json.loads(p.stdout.strip().splitlines()[-1])
        except Exception:
            print(f"[{v or 'default'}] FAILED", p.stdout[-500:], p.stderr[-1500:], flush=True)
            continue
        sus = r["sustained"]
        print(f"[{v or 'default':40s}] {wl} window {1e3 * r['ms_per_step']:8.1f} us/step" +
              (f"  sustained {1e3 * sus['ms_per_step']:8.1f} us/step, drift-triggered re-binnings {sus['forced_rebuilds']}" if sus else "") +
              (f"  forced {r['tuning'].get('forced_rebuilds')}" if not sus else "") +
              ("  " + str({a: round(1e3 * b, 1) for a, b in r["kernels_ms"].items()}) if prof else ""), flush=True)
