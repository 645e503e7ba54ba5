"""Manual probe (not a test): the full C4 run (378 700 steps) with the device-side re-bin decision forced on."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("sph-poiseuille-flow_amd")
from types import SimpleNamespace
prm = pkg.config.params_from_values(dp=0.005, DL=12.0, end_time=20.0, output_interval=5.0)
import functools
capi = pkg.capi
orig = capi.Context.__init__
def patched(self, *a, **k):
    k["dynamic_rebin"] = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    return orig(self, *a, **k)
capi.Context.__init__ = patched
res = pkg.driver.run(prm)
print(dict(steps=res.steps, wall=round(res.wall_seconds, 1), rate="%.3e" % res.particle_steps_per_s, L2=round(res.L2_error, 5),
           tau=(round(res.tau_bottom, 4), round(res.tau_top, 4)), policy=res.grid_policy))
