#!/bin/bash
# re-binning interval K at C4 / C5 with the round-3 kernels (window right after the developed start + sustained 1000 steps in)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_ksweep; mkdir -p $O
for wl in C5 C4; do
for K in 5 6 8 10; do
  python - $wl $K <<'PY'
import importlib, sys, time
sys.path.insert(0, '.')
import bench
pkg = importlib.import_module(bench.PKG)
wl, K = sys.argv[1], int(sys.argv[2])
steps, sus = (100, (1000, 300)) if wl == "C5" else (300, (2000, 1000))
r = bench.run_case(pkg.capi, pkg.config, pkg.geometry, wl, dict(bench.WORKLOADS[wl]), steps, 40, 0, rebuild_every=K, sustained=sus)[0]
print(wl, "K", K, f"window {1e3*r['ms_per_step']:.1f} us/step  sustained {1e3*r['sustained']['ms_per_step']:.1f} us/step  forced/drift re-binnings {r['sustained']['forced_rebuilds']}", flush=True)
PY
done; done 2>&1 | tee $O/ksweep.txt
