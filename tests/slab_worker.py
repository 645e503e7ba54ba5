"""Worker for the multi-rank slab tests (launched once per rank by torch.distributed.run).

  --engine hip     : every rank drives a libsphx slab context; ranks may share one GPU (gloo backend,
                     messages staged through host memory) -- the rehearsal possible on a 1-GPU box.
                     Rank 0 also runs the same case on a single-GPU context and compares.
  --engine oracle  : CPU only.  The per-rank engine is a numpy + oracle emulation of the slab step
                     (tests only); checks the decomposition logic (ownership, 4-column halo sufficiency,
                     ring choreography incl. the world==2 same-peer case) against the global oracle loop.
Exit code 0 = pass.
"""
import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
PKG = "sph-poiseuille-flow_amd"


class OracleSlabEngine:
    """Emulates one rank of the slab decomposition on the CPU with the oracle's operators (TEST ONLY)."""

    def __init__(self, prm, parts, rank, world, halo_cols, msg_cap=4096):
        import torch
        import oracle
        self.torch, self.orc, self.prm = torch, oracle, prm
        self.rank, self.world, self.H = rank, world, halo_cols
        slab = importlib.import_module(PKG + ".slab")
        ncx = slab.n_cell_columns(prm)
        self.csx = prm.DL / ncx
        c0, c1 = slab.partition(ncx, world, halo_cols)[rank]
        self.own_lo, self.own_hi = c0 * self.csx, (prm.DL if rank == world - 1 else c1 * self.csx)
        self.win_lo, self.win_hi = (c0 - halo_cols) * self.csx, (c1 + halo_cols) * self.csx
        self.halo_w = halo_cols * self.csx
        self.shift_l = prm.DL if rank == 0 else 0.0
        self.shift_r = -prm.DL if rank == world - 1 else 0.0
        nf, nt = parts["n_fluid"], parts["n_total"]
        xw = parts["pos"][:, 0] - np.floor(parts["pos"][:, 0] / prm.DL) * prm.DL
        rows = {k: [] for k in ("x", "y", "vx", "vy", "drho", "mass", "id")}
        wall = {k: [] for k in ("x", "y", "mass", "wvx", "wvy")}
        for im in (-1, 0, 1):
            xs = xw + im * prm.DL
            sel = (xs >= self.win_lo) & (xs < self.win_hi)
            f = sel & (np.arange(nt) < nf)
            w = sel & (np.arange(nt) >= nf)
            rows["x"].append(xs[f]); rows["y"].append(parts["pos"][f, 1]); rows["vx"].append(parts["vel"][f, 0])
            rows["vy"].append(parts["vel"][f, 1]); rows["drho"].append(parts["drho_dt"][f]); rows["mass"].append(parts["mass"][f])
            rows["id"].append(np.nonzero(f)[0])
            wall["x"].append(xs[w]); wall["y"].append(parts["pos"][w, 1]); wall["mass"].append(parts["mass"][w])
            wall["wvx"].append(parts["wall_vel"][w, 0]); wall["wvy"].append(parts["wall_vel"][w, 1])
        self.f = {k: np.concatenate(v) for k, v in rows.items()}
        self.w = {k: np.concatenate(v) for k, v in wall.items()}
        self.msg_cap = msg_cap
        mk = lambda: torch.zeros(1 + 7 * msg_cap, dtype=torch.float64)
        self.send_l, self.send_r, self.recv_l, self.recv_r = mk(), mk(), mk(), mk()
        self.vmax = torch.zeros(1, dtype=torch.float64)
        self.t, self.step, self.dt, self.t_target, self.steps_left = 0.0, 0, 0.0, 0.0, 0

    def stream_ctx(self):
        import contextlib
        return contextlib.nullcontext()

    def _owned(self, x):
        return (x >= self.own_lo) & (x < self.own_hi)

    def local_vmax(self):
        o = self._owned(self.f["x"])
        v = np.sqrt(self.f["vx"][o] ** 2 + self.f["vy"][o] ** 2)
        self.vmax[0] = float(v.max()) if v.size else 0.0

    def _next_dt(self, vmax):
        p = self.prm
        remain = min(self.t_target - self.t, p.t_end - self.t)
        dt = min(0.25 * p.h / max(p.c_f + vmax, 1e-12), 0.125 * p.h ** 2 / max(p.nu, 1e-12),
                 0.25 * np.sqrt(p.h / max(abs(p.gravity_g), 1e-12)), remain)
        return max(dt, 1e-12)

    def prepare(self, t_target, max_steps):
        self.t_target = min(t_target, self.prm.t_end)
        self.dt = self._next_dt(float(self.vmax[0]))

    def compute(self):
        p, f, w, orc = self.prm, self.f, self.w, self.orc
        nfl, nwl = len(f["x"]), len(w["x"])
        ntl = nfl + nwl
        big, off = 1.0e6, 10.0 - self.win_lo  # open window: no periodic images inside the oracle's search
        pos = np.zeros((ntl, 2), order="F")
        pos[:nfl, 0], pos[:nfl, 1] = f["x"] + off, f["y"]
        pos[nfl:, 0], pos[nfl:, 1] = w["x"] + off, w["y"]
        vel = np.zeros((ntl, 2), order="F"); vel[:nfl, 0], vel[:nfl, 1] = f["vx"], f["vy"]
        wv = np.zeros((ntl, 2), order="F"); wv[nfl:, 0], wv[nfl:, 1] = w["wvx"], w["wvy"]
        mass = np.concatenate([f["mass"], w["mass"]])
        drho = np.concatenate([f["drho"], np.zeros(nwl)])
        nb = orc.neighbor_search(pos, nfl, ntl, p.h, big)
        rho, Vol, B = orc.density_correction(nb, mass, nfl, ntl, p.rho0, p.h, p.inv_sigma0)
        fp = orc.viscous_force(nb, vel, Vol, B, p.mu, p.h, nfl, ntl, mass, wv)
        fp[:nfl, 0] += mass[:nfl] * p.gravity_g
        pos_t = orc.transport_correction(nb, Vol, B, pos, p.h, nfl, ntl, p.transport_coeff)
        _, _, pos2, vel2, drho2, _ = orc.integration_verlet(nb, Vol, B, rho, mass, pos_t, vel, drho, fp, self.dt, nfl, ntl,
                                                            p.rho0, p.p0, p.c_f, wv)
        own = self._owned(f["x"])
        xn, yn = pos2[:nfl, 0] - off, pos2[:nfl, 1]
        new = dict(x=xn, y=yn, vx=vel2[:nfl, 0], vy=vel2[:nfl, 1], drho=drho2[:nfl], mass=f["mass"], id=f["id"].astype(np.float64))
        v = np.sqrt(new["vx"][own] ** 2 + new["vy"][own] ** 2)
        self.vmax[0] = float(v.max()) if v.size else 0.0
        keep = own & (xn >= self.win_lo) & (xn < self.win_hi)  # migrants stay as halo copies
        sl = own & (xn < self.own_lo + self.halo_w)
        sr = own & (xn >= self.own_hi - self.halo_w)
        self.keep = {k: a[keep] for k, a in new.items()}
        for msg, sel, shift in ((self.send_l, sl, self.shift_l), (self.send_r, sr, self.shift_r)):
            n = int(sel.sum())
            assert n <= self.msg_cap
            msg.zero_()
            msg[0] = n
            for b, k in enumerate(("x", "y", "vx", "vy", "drho", "mass", "id")):
                vals = new[k][sel] + (shift if k == "x" else 0.0)
                msg[1 + b * self.msg_cap: 1 + b * self.msg_cap + n] = self.torch.from_numpy(np.ascontiguousarray(vals))

    def finish(self):
        parts = [self.keep]
        for msg in (self.recv_l, self.recv_r):
            n = int(msg[0].item())
            m = msg.numpy()
            parts.append({k: m[1 + b * self.msg_cap: 1 + b * self.msg_cap + n].copy()
                          for b, k in enumerate(("x", "y", "vx", "vy", "drho", "mass", "id"))})
        self.f = {k: np.concatenate([q[k] for q in parts]) for k in ("x", "y", "vx", "vy", "drho", "mass", "id")}
        self.f["id"] = self.f["id"].astype(np.int64)
        self.t += self.dt
        self.step += 1
        self.dt = self._next_dt(float(self.vmax[0]))

    def sync(self):
        return dict(t=self.t, step=self.step, dt_last=0.0, dt_next=self.dt, vmax=float(self.vmax[0]), done=0, device_status=0)

    def snapshot(self):
        f = self.f
        return dict(x=f["x"], y=f["y"], vx=f["vx"], vy=f["vy"], drho=f["drho"], id=f["id"].astype(np.int64), owned=self._owned(f["x"]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--engine", choices=("hip", "oracle"), required=True)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--dp", type=float, default=0.05)
    ap.add_argument("--DL", type=float, default=3.0)
    ap.add_argument("--lpp", type=int, default=0)
    ap.add_argument("--native", action="store_true", help="hip engine: the library's own loop over RCCL (one GPU per rank)")
    ap.add_argument("--graph", action="store_true", help="--native: replay the steps as a captured hipGraph after the first four")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    from helpers import assert_close, make_case
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    pkg = importlib.import_module(PKG)
    slab = importlib.import_module(PKG + ".slab")
    prm, parts = make_case(pkg.config, pkg.geometry, dp=args.dp, DL=args.DL, jitter=0.2, seed=11, developed=True, end_time=1e9)
    nf = parts["n_fluid"]
    if args.native:
        # the library's own loop over its own RCCL communicator (one GPU per rank); the ranks agree over gloo that
        # everybody can join before anybody enters the collective ncclCommInitRank (slab.join_native_ring)
        ndev = pkg.capi.device_count()
        assert ndev >= world, f"--native needs a GPU per rank ({ndev} visible, {world} ranks)"
        eng = slab.join_native_ring(lambda: slab.HipSlabEngine(prm, parts, rank, world, rank, lanes_per_particle=args.lpp,
                                                               t_end=1e9, native=True), pkg.capi, rank, dist)
        with slab.Watchdog("native slab steps", 300.0, rank):
            first = min(args.steps, 4)
            eng.run(first)
            if args.graph and args.steps - first >= 10:
                eng.sync()
                slab.HipSlabEngine.graph_prepare([eng])  # ten steps per replay, RCCL calls captured with the kernels
            if args.steps > first:
                eng.run(args.steps - first)
            st = eng.sync()
        drv = slab.SlabDriver.__new__(slab.SlabDriver)
        drv.e, drv.x, drv.steps_done = eng, slab.RingExchange(rank, world), args.steps
    elif args.engine == "hip":
        ndev = max(pkg.capi.device_count(), 1)
        eng = slab.HipSlabEngine(prm, parts, rank, world, rank % ndev, lanes_per_particle=args.lpp, t_end=1e9)
    else:
        eng = OracleSlabEngine(prm, parts, rank, world, slab.HALO_COLS)
    if args.native:
        pass
    else:
        drv = slab.SlabDriver(eng, slab.RingExchange(rank, world))
        st = drv.run_steps(args.steps)
    got = drv.gather_owned(nf)
    ok = True
    if rank == 0:
        pos, vel, drho = got
        if args.engine == "hip":
            with pkg.capi.Context(prm, nf, parts["n_total"], parts["pos"], parts["vel"], parts["drho_dt"], parts["mass"],
                                  parts["wall_vel"], t_end=1e9, lanes_per_particle=args.lpp) as ctx:
                rs = ctx.advance(1e9, max_steps=args.steps)
                ref = ctx.download(fields=("pos", "vel", "drho_dt"))
            t_ref = rs["t"]
        else:
            import oracle
            ref = oracle.run(prm, parts, t_end=1e9, output_interval=1e9, max_steps=args.steps, enable_sort=False)
            t_ref = ref["stats"]["t"]
        try:
            assert abs(st["t"] - t_ref) <= 1e-12 * t_ref, (st["t"], t_ref)
            assert st["step"] == args.steps
            tol = dict(rtol=1e-9, atol_scale=1e-10)
            # (a skinned slab keeps x in the frame of its window until the next re-binning: compare x modulo the period)
            pos[:, 0] -= np.round((pos[:, 0] - ref["pos"][:nf, 0]) / prm.DL) * prm.DL
            assert_close(pos, ref["pos"][:nf], name="pos", **tol)
            assert_close(vel, ref["vel"][:nf], name="vel", **tol)
            assert_close(drho, ref["drho_dt"][:nf], name="drho_dt", **tol)
            print(f"slab {args.engine} world={world} steps={args.steps}: OK t={st['t']:.6g}")
        except AssertionError as e:
            ok = False
            print("SLAB TEST FAILED:", e)
            if os.environ.get("SLAB_DEBUG"):
                err = np.abs(vel - ref["vel"][:nf]).max(axis=1)
                bad = err > 1e-9
                xs = ref["pos"][:nf, 0][bad]
                print("bad count", bad.sum(), "x range of bad:", np.sort(np.unique(np.round(xs, 2)))[:60])
    flag = torch.tensor([1 if ok else 0])
    dist.broadcast(flag, src=0)
    if args.engine == "hip":
        eng.close()
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
