"""Manual probe (not a test): sphx_ctx_time_kernel of one pass (back-to-back replays on a state that is not fed back) under
several SPHX_DEBUG_SWITCHES / library variants, alternating, each in a fresh process.
    python tools/probes/probe_time_kernel.py C5 k_forces 20 3 "" forces_tile_320 forces_tile_320@tools/_exp/libsphx_pretend320.so
args: workload, kernel, replays, repetitions, variants ("switches@library")."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wl, kernel, reps_k, reps = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
variants = sys.argv[5:] or [""]
pre = int(os.environ.get("PROBE_PRE_STEPS", "30"))
code = r'''
import importlib, sys
sys.path.insert(0, %r)
import bench
pkg = importlib.import_module(bench.PKG)
name, kw = bench.parse_workload(%r)
prm = pkg.config.params_from_values(end_time=1e9, **kw)
parts = pkg.geometry.init_particles(prm)
nf, nt = parts["n_fluid"], parts["n_total"]
pos, vel = pkg.geometry.developed_state(prm, parts, jitter=0.05, seed=12345)
with pkg.capi.Context(prm, nf, nt, pos, vel, parts["drho_dt"], parts["mass"], parts["wall_vel"], t_end=1e9) as ctx:
    if %d > 0:  # (a measurement variant whose kernels compute wrong results is timed on the start state)
        ctx.advance(1e9, max_steps=%d)
        ctx.sync()
    print("RESULT", " ".join("%%s=%%.1f" %% (k, 1e3 * ctx.time_kernel(k, %d)) for k in %r.split(",")))
'''
for rep in range(reps):
    for v in variants:
        sw, _, lib = v.partition("@")
        env = dict(os.environ, SPHX_DEBUG_SWITCHES=sw)
        if lib:
            env["SPHX_LIB"] = os.path.join(root, lib)
        p = subprocess.run([sys.executable, "-c", code % (root, wl, pre, pre, reps_k, kernel)], env=env, capture_output=True, text=True)
        out = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
        print(f"[{v or 'default':60s}] {wl} {out[0][7:] if out else 'FAILED ' + p.stderr[-800:]} us per launch", flush=True)
