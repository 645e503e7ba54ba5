"""In-process ring of slabs vs the single-GPU context on one device (manual probe): us/step and the ratio.
python tools/probes/probe_slab_ring.py C4 2 40 [one-stream]
one-stream: every slab of the ring runs on the SAME HIP stream, so the ring executes serially -- kernel durations in a
rocprofv3 trace are then those of one slab having the chip to itself (what a rank of a real multi-GPU run sees)."""
import ctypes, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("sph-poiseuille-flow_amd")
slab = importlib.import_module("sph-poiseuille-flow_amd.slab")
cfg, geo, capi = pkg.config, pkg.geometry, pkg.capi
W = {"C2": dict(dp=0.025, DL=3.0), "C2x2": dict(dp=0.025, DL=6.0), "C3": dict(dp=0.01, DL=6.0), "C4": dict(dp=0.005, DL=12.0),
     "C5": dict(dp=0.002, DL=24.0), "C4x2": dict(dp=0.005, DL=24.0)}
name, world, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
one_stream = "one-stream" in sys.argv[4:]
graph = "graph" in sys.argv[4:]  # the steps replayed as one hipGraph (sphx_slab_graph_prepare)
K = next((int(a[2:]) for a in sys.argv[4:] if a.startswith("K=")), 0)  # re-binning interval of the slabs (0 = the default)
prm = cfg.params_from_values(end_time=1e9, **W[name])
parts = geo.init_particles(prm)
pos, vel = geo.developed_state(prm, parts, jitter=0.05, seed=12345)
stream = None
if one_stream:
    hip = ctypes.CDLL("libamdhip64.so")
    capi.set_device(0)
    h = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(h), ctypes.c_uint(1)) == 0  # hipStreamNonBlocking
    stream = h.value
engines = [slab.HipSlabEngine(prm, parts, r, world, 0, t_end=1e9, pos=pos, vel=vel, native=True, hip_stream=stream, rebuild_every=K) for r in range(world)]
slab.HipSlabEngine.group_run(engines, 8); [e.sync() for e in engines]
if graph:
    slab.HipSlabEngine.graph_prepare(engines)
    slab.HipSlabEngine.group_run(engines, 10); [e.sync() for e in engines]
t0 = time.perf_counter(); slab.HipSlabEngine.group_run(engines, steps); [e.sync() for e in engines]; ring = (time.perf_counter() - t0) / steps
lay = engines[0].layout()
for e in engines: e.close()
ctx = capi.Context(prm, parts["n_fluid"], parts["n_total"], pos, vel, parts["drho_dt"], parts["mass"], parts["wall_vel"], t_end=1e9)
ctx.enqueue_steps(40); ctx.sync()
t0 = time.perf_counter(); ctx.enqueue_steps(steps); ctx.sync(); one = (time.perf_counter() - t0) / steps
print(f"{name}: ring of {world} slabs on one device{' (ONE stream)' if one_stream else ''}{' (step graph)' if graph else ''}{f' K={K}' if K else ''} {1e6*ring:.1f} us/step, single context "
      f"{1e6*one:.1f} us/step, ratio {ring/one:.2f}; slab 0: {lay['n_local']} particles held, capacity {lay['capacity']}")
