"""L2 of the full dp = 0.025, 20 s run for several re-binning intervals (manual probe: the flow is chaotic at round-off,
every K is another realisation).  python tools/probes/probe_longrun_k.py 8 12 16"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("sph-poiseuille-flow_amd")
for K in [int(a) for a in sys.argv[1:]] or [8]:
    prm = pkg.config.params_from_values(dp=0.025, DL=3.0, end_time=20.0, output_interval=1.0)
    r = pkg.driver.run(prm, rebuild_every=K)
    print(f"K={K}: steps {r.steps}, wall {r.wall_seconds:.2f} s, L2(t=20) {100*r.L2_error:.3f} %, L2(mean 16..20) {100*r.L2_time_mean(5):.3f} %, "
          f"tau {r.tau_bottom:.4f}/{r.tau_top:.4f}, policy {r.grid_policy}")
