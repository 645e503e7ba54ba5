"""Multi-rank x-slab path.  CPU (gloo, world 2 and 3): the decomposition logic with an oracle-backed
emulation engine.  GPU (-m gpu): two ranks share the one GPU of the box (gloo, host-staged messages) and
drive the real libsphx slab contexts; the result must match the single-GPU context."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "slab_worker.py")


def _launch(world, *extra, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), WORKER, *extra]
    return subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)


@pytest.mark.parametrize("world,DL", [(2, 3.0), (3, 3.0)])
def test_slab_decomposition_oracle_engine_gloo(world, DL, oracle):
    r = _launch(world, "--engine", "oracle", "--steps", "4", "--dp", "0.05", "--DL", str(DL), port=29511 + world)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "OK" in r.stdout


@pytest.mark.parametrize("fail_rank,stage", [(-1, "none"), (1, "available"), (0, "engine"), (0, "id"), (1, "init")])
def test_ranks_agree_before_the_collective_init(fail_rank, stage):
    """slab.join_native_ring over gloo with a stand-in C API: a rank that cannot load librccl, build its slab, make the id or
    initialise its communicator must take every rank out with it (RuntimeError everywhere, nobody left inside the collective
    ncclCommInitRank) -- and a healthy ring joins.  CPU only."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(29571 + max(fail_rank, 0) + 3 * ["none", "available", "engine", "id", "init"].index(stage)),
           os.path.join(ROOT, "tests", "join_worker.py"), "--fail-rank", str(fail_rank), "--stage", stage]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("OK") == 2, r.stdout


def test_watchdog_ends_a_phase_that_does_not_return():
    """bench.py --gpus N runs every potentially blocking phase under slab.Watchdog: a phase that overruns ends the process
    with exit code 3 and says which phase it was; one that returns in time leaves no trace."""
    code = ("import importlib, sys, time; sys.path.insert(0, %r); slab = importlib.import_module('sph-poiseuille-flow_amd.slab')\n"
            "with slab.Watchdog('quick phase', 5.0, 0): pass\n"
            "with slab.Watchdog('stuck collective', 0.3, 7): time.sleep(30)\n"
            "print('not reached')" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 3, (r.returncode, r.stdout, r.stderr)
    assert "rank 7" in r.stderr and "stuck collective" in r.stderr and "not reached" not in r.stdout


def test_partition_rules():
    import importlib
    sys.path.insert(0, ROOT)
    slab = importlib.import_module("sph-poiseuille-flow_amd.slab")
    cols = slab.partition(46, 2)
    assert cols == [(0, 23), (23, 46)]
    assert slab.partition(369, 8)[-1][1] == 369
    with pytest.raises(ValueError):
        slab.partition(46, 16)  # 2-3 columns per slab < halo+1


@pytest.mark.gpu
@pytest.mark.parametrize("lpp", [0, 1])
def test_slab_hip_two_ranks_one_gpu(lpp):
    r = _launch(2, "--engine", "hip", "--steps", "7", "--dp", "0.05", "--DL", "3.0", "--lpp", str(lpp), port=29531 + lpp)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "OK" in r.stdout


@pytest.mark.gpu
def test_slab_hip_two_ranks_half_million_particles():
    """C4 (dp 0.005, DL 12: 499 200 particles) cut into two slabs sharing the GPU: the multi-block cell scan
    (> 8 192 cells per slab window), automatic lanes per particle and message buffers of thousands of
    particles -- against the single-GPU context, which test_gpu_large_configs.py pins to the oracle."""
    r = _launch(2, "--engine", "hip", "--steps", "4", "--dp", "0.005", "--DL", "12.0", port=29541)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "OK" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("world,dp,DL,steps,kw", [
    (2, 0.05, 3.0, 7, dict(rebuild_every=1)),                # re-binning every step: the protocol of compute / finish
    (2, 0.05, 3.0, 27, dict()),                              # default: every 24th step or when the drift bound says so, frozen layouts in between
    (2, 0.05, 3.0, 49, dict(calls=[5, 18, 1, 1, 23, 1])),    # ... in six calls, some of them ending on a scheduled re-binning step
    (2, 0.05, 3.0, 23, dict(calls=[5, 5, 1, 9, 3], rebuild_every=5)),
    (3, 0.05, 4.5, 23, dict(rebuild_every=4)),
    # the two-stream step (maxima + all-reduce beside pass E, the exchange beside the interior of pass A) is the default from
    # 150 k particles per slab only: here it is asked for on small rings (SPHX_SLAB_OVERLAP), and refused on a large one below
    (2, 0.05, 3.0, 49, dict(calls=[5, 18, 1, 1, 23, 1], overlap="always")),
    (3, 0.05, 4.5, 23, dict(rebuild_every=4, overlap="always")),
    (2, 0.04, 3.0, 17, dict(rebuild_every=8, skin_h=0.05, overlap="always")),
    (4, 0.01, 6.0, 26, dict(overlap="always")),
    (2, 0.005, 12.0, 7, dict(overlap="never")),
    (2, 0.04, 3.0, 17, dict(rebuild_every=8, skin_h=0.05)),  # skin far too thin: the drift bound triggers the re-binnings
    (4, 0.01, 6.0, 26, dict()),
    (2, 0.005, 12.0, 7, dict()),                             # 0.25 M particles per slab: multi-block scan, 2 lanes per particle
    (2, 0.004, 40.0, 7, dict()),                             # 1.25 M per slab: LDS tiles in every pass, slot-coded lists, stored tile layouts
    # four slabs of 0.25 M particles, each on a stream of its own and competing for the chip: k_slab_pack3's grid is many times
    # the workgroups a frozen step needs, so some are dispatched after the kernel's epilogue has advanced the clock -- they
    # must still take the decision the others took (Clock::pos_q); 14 steps at K = 12: the scheduled re-binning, whatever
    # the drift bound asks for on the way, and the steps before them
    (4, 0.005, 24.0, 14, dict(rebuild_every=12, skin_h=0.28)),
    # the steps as ONE replayed hipGraph (sphx_slab_graph_prepare: ten steps per replay, kernels + copies + cross-stream
    # dependencies captured): prepared after the first call; 25 = two replays + five eager steps, and the last call finds
    # the other state parity -> eager again
    (2, 0.05, 3.0, 47, dict(calls=[3, 25, 19], graph_after=0)),
    (3, 0.05, 4.5, 36, dict(calls=[4, 32], graph_after=0, rebuild_every=4)),
    (2, 0.04, 3.0, 42, dict(calls=[2, 40], graph_after=0, rebuild_every=8, skin_h=0.05)),  # drift-triggered re-binnings inside replays
])
def test_slab_native_ring_in_one_process(world, dp, DL, steps, kw):
    """The library's own step loop (sphx_slab_group_run: every slab of the ring in this process, device-to-device
    copies as the transport, events for the ordering -- the loop sphx_slab_run runs over RCCL) against the single-GPU
    context: ownership hand-over at the re-binnings, fixed exchange lists in between, device-side re-binning decision."""
    import importlib
    import numpy as np
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import assert_close, make_case
    pkg = importlib.import_module("sph-poiseuille-flow_amd")
    slab = importlib.import_module("sph-poiseuille-flow_amd.slab")
    prm, parts = make_case(pkg.config, pkg.geometry, dp=dp, DL=DL, jitter=0.2, seed=11, developed=True, end_time=1e9)
    nf, nt = parts["n_fluid"], parts["n_total"]
    kw = dict(kw)
    calls = kw.pop("calls", [steps])  # the run in several calls: the ids of a re-binning in a call's last step travel with the next call
    graph_after = kw.pop("graph_after", None)
    overlap = kw.pop("overlap", None)  # which form of the skinned step (read by the library when a slab's buffers are made)
    assert sum(calls) == steps
    env_before = os.environ.pop("SPHX_SLAB_OVERLAP", None)
    if overlap:
        os.environ["SPHX_SLAB_OVERLAP"] = overlap
    engines = []
    try:
        engines = [slab.HipSlabEngine(prm, parts, r, world, 0, t_end=1e9, native=True, **kw) for r in range(world)]
        for k_call, n_call in enumerate(calls):
            slab.HipSlabEngine.group_run(engines, n_call)
            if graph_after == k_call:
                slab.HipSlabEngine.graph_prepare(engines)
        sts = [e.sync() for e in engines]
        snaps = [e.snapshot() for e in engines]
    finally:
        for e in engines:
            e.close()
        os.environ.pop("SPHX_SLAB_OVERLAP", None)
        if env_before is not None:
            os.environ["SPHX_SLAB_OVERLAP"] = env_before
    pos, vel, drho = np.full((nf, 2), np.nan), np.full((nf, 2), np.nan), np.full(nf, np.nan)
    seen = np.zeros(nf, dtype=int)
    for sn in snaps:
        o = sn["owned"]
        i = sn["id"][o]
        pos[i, 0], pos[i, 1], vel[i, 0], vel[i, 1], drho[i] = sn["x"][o], sn["y"][o], sn["vx"][o], sn["vy"][o], sn["drho"][o]
        np.add.at(seen, i, 1)
    assert np.all(seen == 1), (int((seen == 0).sum()), int((seen > 1).sum()))
    with pkg.capi.Context(prm, nf, nt, parts["pos"], parts["vel"], parts["drho_dt"], parts["mass"], parts["wall_vel"],
                          t_end=1e9) as ctx:
        rs = ctx.advance(1e9, max_steps=steps)
        ref = ctx.download(fields=("pos", "vel", "drho_dt"))
    for st in sts:
        assert st["step"] == steps and abs(st["t"] - rs["t"]) <= 1e-12 * rs["t"]
    tol = dict(rtol=1e-9, atol_scale=1e-10)
    # a slab keeps x in the frame of its window: an owned particle that has drifted across x = DL (or 0) since the last
    # re-binning is wrapped only then -- compare x modulo the period
    dx = pos[:, 0] - ref["pos"][:nf, 0]
    pos[:, 0] -= np.round(dx / prm.DL) * prm.DL
    assert_close(pos, ref["pos"][:nf], name="pos", **tol)
    assert_close(vel, ref["vel"][:nf], name="vel", **tol)
    assert_close(drho, ref["drho_dt"][:nf], name="drho_dt", **tol)


def _n_gpus():
    import importlib
    sys.path.insert(0, ROOT)
    try:
        return importlib.import_module("sph-poiseuille-flow_amd").capi.device_count()
    except Exception:  # noqa: BLE001 -- no library / no driver: nothing to run on
        return 0


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [(), ("--graph",)], ids=["eager", "step-graph"])
def test_slab_native_rccl_two_ranks(extra):
    """sphx_slab_run over a real two-rank RCCL communicator (one GPU per rank) against the single-GPU context: both messages
    of a step go to the same peer, every rank must take the same re-binning decisions, the ids of a re-binning travel with
    the next step.  Needs two GPUs: skipped on the one-GPU test box (there the loop runs as an in-process ring,
    test_slab_native_ring_in_one_process, and the RCCL calls on a one-rank communicator, below)."""
    if _n_gpus() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    r = _launch(2, "--engine", "hip", "--native", "--steps", "37", "--dp", "0.05", "--DL", "3.0", *extra, port=29551)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "OK" in r.stdout


@pytest.mark.gpu
def test_rccl_exchange_pattern_on_one_rank():
    """No second GPU on the test box (RCCL refuses two ranks on one device): the next best check of sphx_slab_run's RCCL
    calls is the library's self-test -- the dlopen'ed entry points, the enum values, the grouped two sends / two receives
    to the same peer (served in order, which the two-rank ring relies on) and the 16-byte all-reduce(max), on a
    communicator of one rank."""
    import importlib
    sys.path.insert(0, ROOT)
    pkg = importlib.import_module("sph-poiseuille-flow_amd")
    pkg.capi.check(pkg.capi.lib().sphx_comm_selftest())
    pkg.capi.check(pkg.capi.lib().sphx_comm_selftest())  # (communicators come and go cleanly)


@pytest.mark.gpu
def test_rccl_calls_can_be_captured_into_a_graph():
    """What sphx_slab_graph_prepare stakes the multi-GPU loop on: the grouped sends / receives and the all-reduce captured
    into a hipGraph (two exchanges per graph) and replayed twice with fresh payloads, on a communicator of one rank."""
    import importlib
    sys.path.insert(0, ROOT)
    pkg = importlib.import_module("sph-poiseuille-flow_amd")
    pkg.capi.check(pkg.capi.lib().sphx_comm_selftest_graph())


def _bench(*argv, env=None, timeout=600):
    e = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TORCHELASTIC_RUN_ID", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=e, cwd=ROOT, capture_output=True,
                          text=True, timeout=timeout)


def _json_line(stdout):
    import json
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_bare_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it must start two ranks itself (children of a parent that makes no
    GPU call) and report n_gpus = 2 -- not measure one GPU and say so quietly.  --check-launch stops after the rendezvous, so
    this runs without a GPU."""
    r = _bench("--gpus", "2", "--check-launch", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["check_launch"] is True
    assert "starting 2 ranks" in r.stderr


@pytest.mark.parametrize("world,gpus", [(1, 2), (1, 8), (2, 1)])
def test_bench_rejects_a_world_size_that_is_not_gpus(world, gpus):
    """Under a launcher (RANK / WORLD_SIZE set) a mismatch is an error with a non-zero exit -- including WORLD_SIZE = 1."""
    r = _bench("--gpus", str(gpus), "--check-launch", env=dict(RANK="0", WORLD_SIZE=str(world), LOCAL_RANK="0",
                                                               MASTER_ADDR="127.0.0.1", MASTER_PORT="29591"), timeout=120)
    assert r.returncode == 2, (r.returncode, r.stdout, r.stderr)
    assert f"--gpus {gpus}" in r.stderr and "{" not in r.stdout


@pytest.mark.gpu
def test_bare_bench_two_ranks_share_the_gpu():
    """The bare two-rank command for real: SPHX_DIST_BACKEND=gloo lets the ranks share the box's one GPU (messages staged
    through host memory), everything else is the path `bench.py --gpus 2` takes on a two-GPU node."""
    r = _bench("--gpus", "2", "--steps", "6", "--warmup", "2", "--no-aux", env=dict(SPHX_DIST_BACKEND="gloo"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["steps"] == 6 and line["value"] > 0 and line["scaling"] == "weak"
