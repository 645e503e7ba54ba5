"""Manual probe (not a test): throughput of the unmodified six-MEX-calls-per-step loop through the stateless C ABI
(host buffers in and out on every call: the PCIe-inclusive rate) next to the device-resident loop."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("sph-poiseuille-flow_amd")
for name, kw, t_end in (("C2", dict(dp=0.025, DL=3.0), 0.1), ("C3", dict(dp=0.01, DL=6.0), 0.01)):
    prm = pkg.config.params_from_values(end_time=t_end, output_interval=t_end, **kw)
    a = pkg.driver.run(prm, engine="mex")
    b = pkg.driver.run(prm, engine="resident")
    print(name, "mex engine: %d steps, %.3e particle-steps/s (%.0f us/step)" % (a.steps, a.particle_steps_per_s, 1e6 * a.wall_seconds / a.steps),
          "| resident: %.3e" % b.particle_steps_per_s, flush=True)
