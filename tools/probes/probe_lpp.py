"""Manual probe (not a test): lanes per particle at one workload, each setting in a fresh process (bench.run_case).
    python tools/probes/probe_lpp.py "dp=0.008,DL=8" 1000 100 0 2 4 8      (0 = the context's own choice)"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wl, steps, warm = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
code = r'''
import importlib, json, sys
sys.path.insert(0, %r)
import bench
pkg = importlib.import_module(bench.PKG)
name, kw = bench.parse_workload(%r)
r = bench.run_case(pkg.capi, pkg.config, pkg.geometry, name, kw, %d, %d, 0, lpp=%d)[0]
print(json.dumps({k: r[k] for k in ("ms_per_step", "tuning")}))
'''
for rep in range(2):
    for lpp in sys.argv[4:]:
        p = subprocess.run([sys.executable, "-c", code % (root, wl, steps, warm, int(lpp))], capture_output=True, text=True)
        try:
            r = json.loads(p.stdout.strip().splitlines()[-1])
            print(f"[lpp {lpp:>2} -> {r['tuning']['lanes_per_particle']:>2}] {wl} {1e3 * r['ms_per_step']:8.1f} us/step  K {r['tuning']['rebuild_every']} forced {r['tuning']['forced_rebuilds']}", flush=True)
        except Exception:
            print(f"[lpp {lpp}] FAILED", p.stderr[-400:], flush=True)
