"""Manual probe (not a test): re-binning interval K and cell skin (in units of h) of one workload, each setting in a fresh
process: the window right after bench.py's start, the sustained rate 1 000 steps in, drift-triggered re-binnings.
    python tools/probes/probe_k_skin.py C5 100 40 1000 300 5:0 5:0.22 6:0.28 8:0.35 ...   (K:skin[:dynamic], skin 0 = the default for K)"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wl, steps, warm, skip, nsus = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
code = r'''
import importlib, json, sys
sys.path.insert(0, %r)
import bench
pkg = importlib.import_module(bench.PKG)
name, kw = bench.parse_workload(%r)
r = bench.run_case(pkg.capi, pkg.config, pkg.geometry, name, kw, %d, %d, 0, rebuild_every=%d, skin_h=%g, dynamic=%d, sustained=%s)[0]
print(json.dumps({k: r[k] for k in ("ms_per_step", "sustained", "tuning")}))
'''
for spec in sys.argv[6:]:
    K, skin, dyn = (spec.split(":") + ["0"])[:3]  # dyn: 0 = by size, 1 = device-decided re-binning, 2 = static schedule
    src = code % (root, wl, steps, warm, int(K), float(skin), int(dyn), repr((skip, nsus)) if skip > 0 else "None")
    p = subprocess.run([sys.executable, "-c", src], capture_output=True, text=True)
    try:
        r = json.loads(p.stdout.strip().splitlines()[-1])
    except Exception:
        print(f"[K={K} skin={skin}] FAILED", p.stdout[-300:], p.stderr[-800:], flush=True)
        continue
    sus = r["sustained"]
    print(f"[K={K:>2} skin={skin:>5} h -> {r['tuning']['skin']:.3e} dyn={dyn}] {wl} window {1e3 * r['ms_per_step']:8.1f} us/step" +
          (f"  sustained {1e3 * sus['ms_per_step']:8.1f} us/step, drift-triggered re-binnings {sus['forced_rebuilds']}" if sus else
           f"  forced {r['tuning'].get('forced_rebuilds')}"), flush=True)
