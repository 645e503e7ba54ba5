import importlib, os, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
from helpers import make_case
pkg = importlib.import_module("sph-poiseuille-flow_amd")
cfg, geo, capi = pkg.config, pkg.geometry, pkg.capi
prm, parts = make_case(cfg, geo, dp=0.04, DL=3.0, jitter=0.3, seed=11, developed=True)
def mk(**kw):
    return capi.Context(prm, parts["n_fluid"], parts["n_total"], parts["pos"], parts["vel"], parts["drho_dt"], parts["mass"], parts["wall_vel"], t_end=1e9, **kw)
for use_prepare in (True, False):
    with mk(rebuild_every=8, skin_h=0.8) as ctx:
        ctx.enqueue_steps(5); print(ctx.sync(), ctx.grid_policy(), ctx.graph_stats())
        if use_prepare: ctx.prepare_steps(20)
        print("after prepare", ctx.graph_stats())
        ctx.enqueue_steps(20); print(ctx.sync(), ctx.grid_policy(), ctx.graph_stats())
