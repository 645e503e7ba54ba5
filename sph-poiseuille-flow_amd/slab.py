"""x-slab domain decomposition: one process per GPU, ring halo exchange over torch.distributed.

The channel is periodic and long in x, so it is cut into `world` slabs of whole cell columns (cells are
column-major on the device, so a slab is one contiguous index range).  Each rank keeps its columns plus
HALO_COLS columns of copies on either side and runs the ordinary step kernels on that open window; per
step it sends the end-of-step state of its boundary columns (and of particles that crossed the boundary)
to its two ring neighbours and all-reduces max|v| for the global dt (SPH_Poiseuille.m:521).  That is the
only communication: two point-to-point messages per neighbour pair (direct xGMI hops on a ring of 8) and
one 8-byte all-reduce per step.  The reference has no counterpart (single process).

Layers:
  partition()      -- which cell columns a rank owns (same arithmetic as sphx_slab_create)
  RingExchange     -- the per-step message choreography (nccl == RCCL on GPU tensors; gloo staged
                      through host memory for CPU tests and single-GPU rehearsals)
  HipSlabEngine    -- the per-rank compute engine (libsphx slab context)
  SlabDriver       -- step loop = engine.compute -> exchange -> engine.finish
  bench_main()     -- bench.py's multi-GPU leg
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os
import time

import numpy as np

HALO_COLS = 4  # four dependent neighbour passes per step, each reaching one >=2h column further


def n_cell_columns(prm) -> int:
    return int(math.floor(prm.DL / (2.0 * prm.h)))


def partition(ncx: int, world: int, halo_cols: int = HALO_COLS):
    """[(col0, col1)] per rank; every slab needs halo_cols+1 columns so that halos come from the two
    ring neighbours only."""
    cols = [((r * ncx) // world, ((r + 1) * ncx) // world) for r in range(world)]
    for c0, c1 in cols:
        if c1 - c0 < halo_cols + 1:
            raise ValueError(f"slab of {c1 - c0} columns < halo_cols+1 = {halo_cols + 1}: use fewer ranks or a longer channel")
    if world > 1 and max(c1 - c0 for c0, c1 in cols) + 2 * halo_cols >= ncx:
        raise ValueError("slab window would cover the whole period")
    return cols


class RingExchange:
    """Per-step communication of one rank: send_l -> left neighbour, send_r -> right neighbour,
    recv_r <- right neighbour's send_l, recv_l <- left neighbour's send_r, then max-all-reduce of vmax.
    With world == 2 both neighbours are the same peer; the tags and the posting order (receives in the
    order the peer sends: its left message first) keep the two messages apart."""

    def __init__(self, rank: int, world: int, group=None, stage_through_host: bool | None = None):
        import torch.distributed as dist
        self.dist = dist
        self.rank, self.world, self.group = rank, world, group
        self.left, self.right = (rank - 1) % world, (rank + 1) % world
        backend = dist.get_backend(group)
        self.stage = (backend != "nccl") if stage_through_host is None else stage_through_host
        # RCCL: the 8-byte max all-reduce gets a communicator (and therefore a stream) of its own, so it runs beside
        # the halo messages instead of behind them -- both only wait for the pack kernel
        self.reduce_group = None
        if backend == "nccl" and world > 1 and os.environ.get("SPHX_SLAB_SERIAL_COLLECTIVES") != "1":
            ranks = list(range(world)) if group is None else dist.get_process_group_ranks(group)
            self.reduce_group = dist.new_group(ranks=ranks, backend="nccl")  # collective call: every rank gets here

    def __call__(self, send_l, send_r, recv_l, recv_r, vmax):
        dist = self.dist
        if self.stage and send_l.is_cuda:
            import torch
            h = [t.cpu() for t in (send_l, send_r)]
            hr_l, hr_r = torch.empty_like(h[0]), torch.empty_like(h[1])
            hv = vmax.cpu()
            self._p2p(h[0], h[1], hr_l, hr_r)
            dist.all_reduce(hv, op=dist.ReduceOp.MAX, group=self.group)
            recv_l.copy_(hr_l)
            recv_r.copy_(hr_r)
            vmax.copy_(hv)
        elif self.reduce_group is not None:
            work = dist.all_reduce(vmax, op=dist.ReduceOp.MAX, group=self.reduce_group, async_op=True)
            self._p2p(send_l, send_r, recv_l, recv_r)
            work.wait()  # the current stream waits for the reduce stream; no host block
        else:
            self._p2p(send_l, send_r, recv_l, recv_r)
            dist.all_reduce(vmax, op=dist.ReduceOp.MAX, group=self.group)

    def _p2p(self, send_l, send_r, recv_l, recv_r):
        dist = self.dist
        key = (send_l.data_ptr(), send_r.data_ptr(), recv_l.data_ptr(), recv_r.data_ptr())
        if getattr(self, "_ops_key", None) != key:  # the four buffers are fixed: build the op list once, not per step
            self._ops = [dist.P2POp(dist.isend, send_l, self.left, group=self.group, tag=0),
                         dist.P2POp(dist.isend, send_r, self.right, group=self.group, tag=1),
                         dist.P2POp(dist.irecv, recv_r, self.right, group=self.group, tag=0),
                         dist.P2POp(dist.irecv, recv_l, self.left, group=self.group, tag=1)]
            self._ops_key = key
        for req in dist.batch_isend_irecv(self._ops):
            req.wait()

    def reduce_max(self, vmax):
        dist = self.dist
        if self.stage and vmax.is_cuda:
            hv = vmax.cpu()
            dist.all_reduce(hv, op=dist.ReduceOp.MAX, group=self.group)
            vmax.copy_(hv)
        else:
            dist.all_reduce(vmax, op=dist.ReduceOp.MAX, group=self.group)


class HipSlabEngine:
    """libsphx slab context of one rank; device buffers are torch tensors so RCCL can move them."""

    def __init__(self, prm, parts, rank, world, device, lanes_per_particle=0, halo_cols=HALO_COLS, t_end=None,
                 pos=None, vel=None, drho_dt=None, native=False, rebuild_every=0, skin_h=0.0, hip_stream=None):
        """native=True: the context runs on a stream of its own and keeps its message buffers inside the library (the
        native loop: run() over RCCL, or group_run() for a ring living in one process); no torch tensors involved.
        hip_stream (native only): a hipStream_t handle the context runs on instead (profiling: a ring on ONE stream)."""
        from . import capi
        self.capi = capi
        self.rank, self.world, self.native = rank, world, native
        capi.set_device(device)
        if not native:
            import torch
            self.torch = torch
            self.device = torch.device("cuda", device)
            torch.cuda.set_device(self.device)
            self.stream = torch.cuda.Stream(device=self.device)
        # the caller-driven protocol (compute / exchange / finish) re-bins every step; the native loops re-bin every
        # rebuild_every-th step (0 = auto) with a cell skin and fixed exchange lists in between
        self.params = capi.make_params(prm, t_end, None, lanes_per_particle, 0, rebuild_every if native else 1, skin_h)
        nf, nt = parts["n_fluid"], parts["n_total"]
        f = capi.f64
        pos = f(parts["pos"] if pos is None else pos)
        vel = f(parts["vel"] if vel is None else vel)
        drho = f(parts["drho_dt"] if drho_dt is None else drho_dt)
        mass, wv = f(parts["mass"]), f(parts["wall_vel"])
        self._h = C.c_void_p()
        capi.check(capi.lib().sphx_slab_create(C.byref(self._h), C.byref(self.params), C.c_int(nf), C.c_int(nt),
                                               capi.ptr(pos), capi.ptr(vel), capi.ptr(drho), capi.ptr(mass), capi.ptr(wv),
                                               C.c_double(0.0), C.c_int64(0), C.c_int(rank), C.c_int(world),
                                               C.c_int(halo_cols),
                                               C.c_void_p(hip_stream if native else self.stream.cuda_stream)))
        self.n_total_global = nt
        if native:
            return
        lay = self.layout()
        n = lay["msg_doubles"]
        with torch.cuda.stream(self.stream):
            mk = lambda: torch.zeros(n, dtype=torch.float64, device=self.device)
            self.send_l, self.send_r, self.recv_l, self.recv_r = mk(), mk(), mk(), mk()
            self.vmax = torch.zeros(1, dtype=torch.float64, device=self.device)

    def _p(self, t):
        return C.cast(C.c_void_p(t.data_ptr()), C.POINTER(C.c_double))

    def layout(self):
        m, a, b, n, cap = C.c_int64(0), C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        self.capi.check(self.capi.lib().sphx_slab_layout(self._h, C.byref(m), C.byref(a), C.byref(b), C.byref(n), C.byref(cap)))
        return dict(msg_doubles=m.value, col0=a.value, col1=b.value, n_local=n.value, capacity=cap.value)

    def local_vmax(self):
        self.capi.check(self.capi.lib().sphx_slab_local_vmax(self._h, self._p(self.vmax)))

    def prepare(self, t_target, max_steps):
        self.capi.check(self.capi.lib().sphx_slab_prepare(self._h, C.c_double(t_target), C.c_int64(max_steps), self._p(self.vmax)))

    def compute(self):
        self.capi.check(self.capi.lib().sphx_slab_compute(self._h, self._p(self.send_l), self._p(self.send_r), self._p(self.vmax)))

    def finish(self):
        self.capi.check(self.capi.lib().sphx_slab_finish(self._h, self._p(self.recv_l), self._p(self.recv_r), self._p(self.vmax)))

    def sync(self) -> dict:
        st = self.capi.SphxStatus()
        self.capi.check(self.capi.lib().sphx_slab_sync(self._h, C.byref(st)))
        return st.as_dict()

    def snapshot(self) -> dict:
        cap = self.layout()["capacity"]
        n = C.c_int(0)
        d = {k: np.zeros(cap) for k in ("x", "y", "vx", "vy", "drho")}
        ids = np.zeros(cap, dtype=np.int32)
        owned = np.zeros(cap, dtype=np.int32)
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
        self.capi.check(self.capi.lib().sphx_slab_snapshot(self._h, C.c_int(cap), C.byref(n), *[self.capi.ptr(d[k]) for k in
                                                           ("x", "y", "vx", "vy", "drho")], ip(ids), ip(owned)))
        m = n.value
        out = {k: v[:m] for k, v in d.items()}
        out["id"], out["owned"] = ids[:m], owned[:m].astype(bool)
        return out

    def stream_ctx(self):
        return self.torch.cuda.stream(self.stream)

    def rebuild_every(self) -> int:
        k = C.c_int(0)
        self.capi.check(self.capi.lib().sphx_ctx_grid_policy(self._h, C.byref(k), None, None, None))
        return k.value

    # ---- native loop --------------------------------------------------------------------------------
    @staticmethod
    def unique_id(capi) -> bytes:
        buf = C.create_string_buffer(128)
        capi.check(capi.lib().sphx_comm_unique_id(buf, C.c_int(128)))
        return buf.raw

    def comm_init(self, id_bytes: bytes):
        self.capi.check(self.capi.lib().sphx_slab_comm_init(self._h, C.c_char_p(id_bytes)))

    def run(self, n_steps, t_target=1e300):
        """n_steps whole steps over RCCL, enqueued natively; returns at once (sync() waits)."""
        self.capi.check(self.capi.lib().sphx_slab_run(self._h, C.c_double(t_target), C.c_int64(n_steps)))

    @staticmethod
    def group_run(engines, n_steps, t_target=1e300):
        """The ring `engines` (all in this process, one device) takes n_steps steps with device-to-device copies."""
        capi = engines[0].capi
        arr = (C.c_void_p * len(engines))(*[e._h for e in engines])
        capi.check(capi.lib().sphx_slab_group_run(arr, C.c_int(len(engines)), C.c_double(t_target), C.c_int64(n_steps)))

    @staticmethod
    def graph_prepare(engines):
        """Capture ten whole steps (kernels + RCCL calls, or the copies of an in-process ring) into one hipGraph that run()
        / group_run() replay from now on.  One engine: this rank's (collective: every rank calls it at the same point)."""
        capi = engines[0].capi
        arr = (C.c_void_p * len(engines))(*[e._h for e in engines])
        capi.check(capi.lib().sphx_slab_graph_prepare(arr, C.c_int(len(engines))))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.capi.lib().sphx_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: module globals may already be gone
            pass


class SlabDriver:
    """The distributed step loop.  `engine` needs: send_l/send_r/recv_l/recv_r/vmax tensors, local_vmax(),
    prepare(t_target,max_steps), compute(), finish(), sync(), snapshot(), stream_ctx()."""

    def __init__(self, engine, exchange: RingExchange):
        self.e, self.x = engine, exchange
        self.steps_done = 0

    def arm(self, t_target: float, max_steps: int):
        e = self.e
        with e.stream_ctx():
            e.local_vmax()
            self.x.reduce_max(e.vmax)
            e.prepare(t_target, max_steps)

    def step(self):
        e = self.e
        with e.stream_ctx():
            e.compute()
            self.x(e.send_l, e.send_r, e.recv_l, e.recv_r, e.vmax)
            e.finish()
        self.steps_done += 1

    def run_steps(self, n: int, t_target: float = 1e300) -> dict:
        """Exactly n steps (the caller guarantees the loop does not stop earlier)."""
        self.arm(t_target, n)
        e, x = self.e, self.x
        with e.stream_ctx():  # entered once, not per step: the host loop is the critical path of small slabs
            for _ in range(n):
                e.compute()
                x(e.send_l, e.send_r, e.recv_l, e.recv_r, e.vmax)
                e.finish()
        self.steps_done += n
        return self.e.sync()

    def advance(self, t_target: float, dt_floor_hint: float) -> dict:
        """Run to t_target: whole batches while at least that many steps certainly remain, then singly."""
        st = self.e.sync()
        while st["t"] < t_target - 1e-12:
            remaining = t_target - st["t"]
            n_safe = int(remaining / dt_floor_hint) - 1  # dt never exceeds dt_floor_hint -> these all execute
            n = max(1, min(n_safe, 256))
            self.arm(t_target, n)
            for _ in range(n):
                self.step()
            st = self.e.sync()
        return st

    def gather_owned(self, n_fluid: int):
        """Rank 0 gets (pos[n_fluid,2], vel[n_fluid,2], drho[n_fluid]) of the fluid, assembled by particle id."""
        dist = self.x.dist
        s = self.e.snapshot()
        own = s["owned"]
        mine = {k: s[k][own] for k in ("x", "y", "vx", "vy", "drho", "id")}
        bucket = [None] * self.x.world if self.x.rank == 0 else None
        dist.gather_object(mine, bucket, dst=0, group=self.x.group)
        if self.x.rank != 0:
            return None
        pos, vel, drho = np.full((n_fluid, 2), np.nan), np.full((n_fluid, 2), np.nan), np.full(n_fluid, np.nan)
        seen = np.zeros(n_fluid, dtype=np.int64)
        for b in bucket:
            i = b["id"]
            pos[i, 0], pos[i, 1], vel[i, 0], vel[i, 1], drho[i] = b["x"], b["y"], b["vx"], b["vy"], b["drho"]
            np.add.at(seen, i, 1)
        if not np.all(seen == 1):
            raise RuntimeError(f"slab ownership broken: {int((seen == 0).sum())} particles lost, "
                               f"{int((seen > 1).sum())} duplicated")
        return pos, vel, drho


def dt_upper_bound(prm) -> float:
    """dt can never exceed this (vmax = 0): used to batch steps safely in SlabDriver.advance."""
    return min(0.25 * prm.h / max(prm.c_f, 1e-12), 0.125 * prm.h ** 2 / max(prm.nu, 1e-12),
               0.25 * math.sqrt(prm.h / max(abs(prm.gravity_g), 1e-12)))


class Watchdog:
    """A collective that never returns (a rank that died inside ncclCommInitRank, a wedged first exchange) would hold a
    whole node until somebody notices: every potentially blocking phase of the multi-GPU bench runs under this timer,
    which ends THIS process with exit code 3 and a message when the phase overruns."""

    def __init__(self, what: str, seconds: float, rank: int):
        import threading
        self.t = threading.Timer(seconds, self._fire)
        self.t.daemon = True
        self.what, self.seconds, self.rank = what, seconds, rank

    def _fire(self):
        import sys
        sys.stderr.write(f"[sphx slab] rank {self.rank}: '{self.what}' did not finish within {self.seconds:.0f} s -- giving up "
                         f"(exit 3) instead of hanging the node\n")
        sys.stderr.flush()
        os._exit(3)

    def __enter__(self):
        self.t.start()
        return self

    def __exit__(self, *exc):
        self.t.cancel()


def all_agree(dist, ok: bool, group=None) -> bool:
    """True on every rank iff `ok` on every rank (control plane: CPU tensor, gloo)."""
    import torch
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(int(flag.item()))


def join_native_ring(make_engine, capi, rank, dist, group=None, timeout_s=240.0):
    """Collective over `group` (a gloo control plane): every rank returns an engine whose RCCL communicator is up, or every
    rank raises RuntimeError.  ncclCommInitRank is itself collective -- a rank that cannot load librccl or cannot build its
    slab must not leave the others waiting inside it, so the ranks first agree that everything local succeeded."""
    eng, why = None, None
    try:
        capi.check(capi.lib().sphx_comm_available())
        eng = make_engine()
    except Exception as e:  # noqa: BLE001 -- whatever went wrong locally, the other ranks must hear about it
        why = f"rank {rank}: {e}"
    if not all_agree(dist, why is None, group):
        if eng is not None:
            eng.close()
        raise RuntimeError("native RCCL slab loop unavailable: " + (why or "another rank could not load librccl / build its slab"))
    ident = [None]
    if rank == 0:
        try:
            ident = [HipSlabEngine.unique_id(capi)]
        except Exception as e:  # noqa: BLE001
            why = f"rank 0: {e}"
    dist.broadcast_object_list(ident, src=0, group=group)
    if ident[0] is None:
        eng.close()
        raise RuntimeError("native RCCL slab loop unavailable: " + (why or "rank 0 could not make a communicator id"))
    try:
        with Watchdog("ncclCommInitRank", timeout_s, rank):
            eng.comm_init(ident[0])
    except Exception as e:  # noqa: BLE001
        why = f"rank {rank}: {e}"
    if not all_agree(dist, why is None, group):
        eng.close()
        raise RuntimeError("native RCCL slab loop unavailable: " + (why or "ncclCommInitRank failed on another rank"))
    return eng


def bench_main(args, rank, world, local_rank):
    """bench.py --gpus N (N > 1).  Headline: weak scaling of the headline configuration -- every GPU holds one
    dp = 0.025, DL = 3 channel section (5 760 particles), the N-GPU channel is DL = 3 N long.  `aux`: the
    6.1 M-particle channel (C5) cut into N slabs -- the strong-scaling case the north star quotes.

    Control plane (id broadcast, agreement votes, barriers, the timing reduction) = torch.distributed on gloo; data plane =
    the library's own loop over its own dlopen'ed RCCL communicator (SPHX_SLAB_LOOP=native, the default with a GPU per
    rank) -- the only RCCL instance in the process.  When that loop cannot be brought up the run FAILS (non-zero exit,
    reason on stderr); it never measures something else silently.  SPHX_SLAB_LOOP=python asks for the caller-driven
    protocol over torch.distributed's nccl backend instead (opt-in; ten times slower at 5 760 particles per slab);
    SPHX_DIST_BACKEND=gloo is the rehearsal with ranks sharing one GPU (messages staged through host memory)."""
    import importlib
    import sys
    import torch
    import torch.distributed as dist
    pkg = importlib.import_module(__package__)
    cfg, geo, capi = pkg.config, pkg.geometry, pkg.capi
    rehearsal = os.environ.get("SPHX_DIST_BACKEND", "nccl") == "gloo"  # ranks may share one GPU
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    loop = os.environ.get("SPHX_SLAB_LOOP", "native")
    if loop not in ("native", "python"):
        raise SystemExit(f"SPHX_SLAB_LOOP={loop}: expected 'native' or 'python'")
    native = loop == "native" and not rehearsal
    if native and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} with {torch.cuda.device_count()} visible device(s): RCCL needs one GPU per rank "
                         f"(SPHX_DIST_BACKEND=gloo rehearses with ranks sharing a GPU)")
    dist.init_process_group("gloo")  # control plane only
    data_group = None
    if not native and not rehearsal:
        data_group = dist.new_group(ranks=list(range(world)), backend="nccl")  # the Python loop's transport (opt-in)
    workloads = {"C1": dict(dp=0.04, DL=3.0), "C2": dict(dp=0.025, DL=3.0), "C3": dict(dp=0.01, DL=6.0),
                 "C4": dict(dp=0.005, DL=12.0), "C5": dict(dp=0.002, DL=24.0)}

    def run_case(name, steps, warmup):
        strong = name.endswith(":strong")
        base = dict(workloads[name.split(":")[0]])
        kw = dict(base) if strong else dict(base, DL=base["DL"] * world)
        prm = cfg.params_from_values(end_time=1e9, **kw)
        parts = geo.init_particles(prm)
        if args.lattice:
            pos, vel, start = parts["pos"], parts["vel"], "lattice at rest"
        else:
            pos, vel = geo.developed_state(prm, parts, jitter=0.05, seed=12345)
            start = "developed (analytic parabola + 0.05dp jitter, seed 12345)"
        if native:
            eng = join_native_ring(lambda: HipSlabEngine(prm, parts, rank, world, local_rank, lanes_per_particle=args.lpp,
                                                         t_end=1e9, pos=pos, vel=vel, native=True), capi, rank, dist)
            run = lambda n: (eng.run(n), eng.sync())[1]
            graph = os.environ.get("SPHX_SLAB_GRAPH", "0") == "1"  # opt-in: ten steps per replayed hipGraph, RCCL calls inside
        else:
            graph = False
            eng = HipSlabEngine(prm, parts, rank, world, local_rank, lanes_per_particle=args.lpp, t_end=1e9, pos=pos, vel=vel)
            drv = SlabDriver(eng, RingExchange(rank, world, group=data_group))
            run = drv.run_steps
        with Watchdog(f"{name}: warm-up ({warmup} steps)", 600.0, rank):
            if warmup > 0:
                run(warmup)
            if graph:
                if warmup < 2:
                    run(2)
                HipSlabEngine.graph_prepare([eng])
            dist.barrier()
            torch.cuda.synchronize()
        with Watchdog(f"{name}: timed region ({steps} steps)", 900.0, rank):
            t0 = time.perf_counter()
            st = run(steps)
            torch.cuda.synchronize()
            dist.barrier()
            seconds = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
            dist.all_reduce(seconds, op=dist.ReduceOp.MAX)
        seconds = float(seconds.item())
        nt, lay = parts["n_total"], eng.layout()
        K = eng.rebuild_every()
        eng.close()
        alg = (528 * parts["n_fluid"] + 120 * parts["n_wall"]) * steps / seconds / 1e9
        how = (f"native step loop in libsphx over its own RCCL communicator (control plane: gloo): re-binning every {K}th step or "
               "when the all-reduced drift hits the bound, per step one group of ncclSend/ncclRecv with both ring neighbours "
               "(state + list ids) and one 16-byte ncclAllReduce(max)" + ("; ten steps per replayed hipGraph" if graph else "; launched step by step")
               if native else "Python step loop (compute -> exchange -> finish), re-binning every step, over torch.distributed["
                              + ("gloo, messages staged through host memory" if rehearsal else "nccl") + "]")
        return dict(value=nt * steps / seconds, ms_per_step=1e3 * seconds / steps, steps=steps, warmup=warmup,
                    scaling="strong" if strong else "weak",
                    workload=f"{name} x{world if not strong else 1}: dp={prm.dp}, DL={prm.DL}, DH={prm.DH}, "
                             f"n_fluid={parts['n_fluid']}, n_wall={parts['n_wall']}, n_total={nt}; start={start}",
                    parallelism=f"{world} x-slabs (one rank per GPU), {HALO_COLS}-column halo, {how}", slab0=lay,
                    native_loop=bool(native),
                    roofline={"bound": "hbm", "achieved": alg, "peak": 8000.0 * world, "unit": "GB/s",
                              "frac": alg / (8000.0 * world), "traffic": None, "kernel": "whole step, all ranks"},
                    sim={"t": st["t"], "dt": st["dt_last"], "vmax": st["vmax"]})

    try:
        head = run_case(args.workload or "C2", args.steps, args.warmup)
    except RuntimeError as e:  # join_native_ring: raised on every rank
        sys.stderr.write(f"[sphx slab] rank {rank}: {e}\n")
        dist.destroy_process_group()
        raise SystemExit(4)
    aux = {}
    if args.workload is None and not args.no_aux:
        try:  # every rank takes the same path: partition() raises on all ranks or on none
            partition(n_cell_columns(cfg.params_from_values(end_time=1e9, **workloads["C5"])), world)
            aux["C5:strong"] = run_case("C5:strong", 40, 8)
        except (ValueError, RuntimeError) as e:
            aux["C5:strong"] = {"error": repr(e)}
    if rank == 0:
        out = {
            "metric": "particle-steps/s", "value": head["value"], "unit": "particle-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": head["scaling"], "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": head["workload"], "parallelism": head["parallelism"], "slab0": head["slab0"],
                       "native_loop": head["native_loop"]},
            "roofline": head["roofline"], "sim": head["sim"],
        }
        if aux:
            out["aux"] = aux
        print(json.dumps(out))
    dist.destroy_process_group()
