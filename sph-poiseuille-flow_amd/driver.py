"""Host driver -- the time loop of SPH_Poiseuille.m S5-S7 on top of libsphx.

Two engines with the same observable behaviour:
  * engine="resident" (default): state lives in HBM in a sphx context; every inner-while of
    SPH_Poiseuille.m:250-292 is one sphx_ctx_advance call, the host only acts at output points
    (:294-301) and for the every-20-steps log line (:285-291, opt-in).
  * engine="mex": the reference's own call sequence, six MEX-surface calls per step through
    mex_surface.py (:254-281) -- kept to show that the per-call surface is a drop-in.
Restart / post-process files (:127-163,:295,:305-306) go through restart.py; the plots (S7) are out of scope.
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field

import numpy as np

from . import capi
from .geometry import init_particles
from .mex_surface import sph_neighbor_search_mex, sph_physics_shell_mex
from . import restart
from .profile import compute_mid_channel_profile, final_profile, l2_error, n_profile_bins


@dataclass
class RunResult:
    prm: object
    n_fluid: int
    n_total: int
    t: float
    steps: int
    wall_seconds: float
    pos: np.ndarray
    vel: np.ndarray
    y_mid: np.ndarray
    u_mean: np.ndarray
    u_exact: np.ndarray
    L2_error: float
    profile_times: list = field(default_factory=list)
    mid_profile_u: list = field(default_factory=list)
    tau_bottom: float = 0.0
    tau_top: float = 0.0
    tau_target: float = 0.0
    grid_policy: dict = field(default_factory=dict)  # resident engine: rebuild interval, skin, forced rebuilds
    full_profile_u: list = field(default_factory=list)  # whole-channel binned u_x(y) at every output point
    n_inner: int = 1  # inner sub-steps per counted step (> 1 only with the opt-in dual-rate loop)

    def L2_time_mean(self, last=5):
        """L2 of the whole-channel profile averaged over the last `last` output points: the instantaneous profile of
        one chaotic realisation wanders by ~0.1 pp around its mean, the average does not."""
        u = np.nanmean(np.stack(self.full_profile_u[-last:]), axis=0)
        return l2_error(u, self.u_exact)

    @property
    def particle_steps_per_s(self):
        return self.n_total * self.steps / max(self.wall_seconds, 1e-30)


def verlet_time_step(vel_fluid, c_max, h, nu, gravity_g, remain):
    """SPH_Poiseuille.m:519-527 (used by the 'mex' engine; the resident engine does this on device)."""
    v_max = float(np.max(np.sqrt(vel_fluid[:, 0] ** 2 + vel_fluid[:, 1] ** 2))) if len(vel_fluid) else 0.0
    dt_acoustic = 0.25 * h / max(c_max + v_max, 1e-12)
    dt_viscous = 0.125 * h * h / max(nu, 1e-12)
    dt_body = 0.25 * np.sqrt(h / max(abs(gravity_g), 1e-12))
    return max(min(dt_acoustic, dt_viscous, dt_body, remain), 1e-12)


def periodic_bounding(pos, n_fluid, DL):
    """SPH_Poiseuille.m:570-577."""
    x = pos[:n_fluid, 0]
    pos[:n_fluid, 0] = x - np.floor(x / DL) * DL
    return pos


def run(prm, engine="resident", log=None, log_every=0, parts=None, lanes_per_particle=0, steps_per_graph=0,
        rebuild_every=0, restart_path=None, postprocess_path=None, dual_rate=0, mat_format="auto"):
    """Run to prm.t_end and return the final profile and L2 (SPH_Poiseuille.m:246-307 + postprocess :42).

    restart_path (resident engine): the reference's restart.mat protocol -- resume from it when
    prm.restart_from_file is set and its signature / sizes match (:132-163), rewrite it at every output point
    (:295).  postprocess_path: write SPH_Poiseuille_postprocess.mat at the end (:305-306).  mat_format: "7.3" (HDF5, what
    the reference writes), "5", or "auto" = 7.3 where a libhdf5 can be loaded (restart.py); either is read back.
    dual_rate (resident engine, opt-in, NOT the reference's loop): up to that many acoustic sub-steps per outer step,
    see sphx_params.dual_rate in include/sphx.h; the result then carries n_inner and steps counts outer steps."""
    parts = init_particles(prm) if parts is None else parts
    nf, nt = parts["n_fluid"], parts["n_total"]
    t_start, step_start, n_inner = 0.0, 0, 1
    if restart_path and engine != "resident":
        raise ValueError("restart files are handled by the resident engine")
    if restart_path and prm.restart_from_file:
        st, why = restart.load_restart(restart_path, nt, prm.config_signature)
        if st is not None:
            parts = dict(parts, pos=st["pos"], vel=st["vel"], drho_dt=st["drho_dt"])
            t_start, step_start = st["t"], st["step"]
            if log:
                log(f"Restart: resuming from t={t_start:.6f}, step={step_start}")
        elif log:
            log(f"Restart file not used ({why}); starting from scratch")
    n_bins = n_profile_bins(prm.DH, prm.dp)
    mid_x, mid_hw = 0.5 * prm.DL, max(prm.dp, prm.h)
    profile_times, mid_profiles, full_profiles = [0.0], [], []
    _, u0 = compute_mid_channel_profile(parts["pos"][:nf], parts["vel"][:nf, 0], prm.DL, prm.DH, mid_x, mid_hw, n_bins)
    mid_profiles.append(u0)
    tau_b = tau_t = 0.0
    policy = {}
    t0 = time.perf_counter()
    if engine == "resident":
        ctx = capi.Context(prm, nf, nt, parts["pos"], parts["vel"], parts["drho_dt"], parts["mass"],
                           parts["wall_vel"], t0=t_start, step0=step_start, lanes_per_particle=lanes_per_particle,
                           steps_per_graph=steps_per_graph, rebuild_every=rebuild_every, dual_rate=dual_rate)
        n_inner = ctx.substeps()
        t, step = t_start, step_start
        try:
            while t < prm.t_end - 1e-12:
                target = min(t + prm.output_interval, prm.t_end)
                while t < target - 1e-12:
                    st = ctx.advance(target, max_steps=log_every if log_every else 0)
                    t, step = st["t"], st["step"]
                    if log and log_every:
                        tb, tt, npairs = ctx.monitor(tau=True, pairs=True)
                        log(f"step={step}, t={t:.6f}/{prm.t_end:.6f}, dt={st['dt_last']:.4e}, pairs={int(npairs)}, "
                            f"vmax={st['vmax']:.4f}\n  [thick-wall-noslip] tau_bot={tb:.4f}, tau_top={tt:.4f}, "
                            f"tau_target={prm.gravity_g * prm.rho0 * prm.DH / 2:.4f}")
                d = ctx.download() if restart_path else ctx.download(fields=("pos", "vel"))
                _, u = compute_mid_channel_profile(d["pos"][:nf], d["vel"][:nf, 0], prm.DL, prm.DH, mid_x, mid_hw, n_bins)
                profile_times.append(t)
                mid_profiles.append(u)
                full_profiles.append(final_profile(np.column_stack([np.mod(d["pos"][:nf, 0], prm.DL), d["pos"][:nf, 1]]),
                                                   d["vel"][:nf, 0], prm)[1])
                if restart_path:
                    restart.save_restart(restart_path, prm.config_signature, dict(d, t=t, step=step), fmt=mat_format)
                if log:
                    log(f"output point: t={t:.6f}, step={step}")
            wall = time.perf_counter() - t0
            tau_b, tau_t, _ = ctx.monitor(tau=True)
            d = ctx.download(fields=("pos", "vel"))
            pos, vel = d["pos"], d["vel"]
            policy = ctx.grid_policy()
        finally:
            ctx.close()
    elif engine == "mex":
        S = {k: np.array(parts[k], order="F", copy=True) for k in ("pos", "vel", "drho_dt", "mass", "wall_vel")}
        nb = sph_neighbor_search_mex(S["pos"], nf, nt, prm.h, prm.DL)
        t, step = 0.0, 0
        while t < prm.t_end - 1e-12:
            target = min(t + prm.output_interval, prm.t_end)
            while t < target - 1e-12:
                step += 1
                remain = min(target - t, prm.t_end - t)
                pi, pj, dx, dy, r, W, dW = nb
                rho, Vol, B = sph_physics_shell_mex("density_correction", pi, pj, dx, dy, r, W, dW, S["mass"], nf, nt,
                                                    prm.rho0, prm.h, prm.inv_sigma0)
                fp = sph_physics_shell_mex("viscous_force", pi, pj, dx, dy, r, dW, S["vel"], Vol, B, prm.mu, prm.h,
                                           nf, nt, S["mass"], S["wall_vel"])
                fp[:nf, 0] += S["mass"][:nf] * prm.gravity_g
                fp[nf:, :] = 0.0
                S["pos"] = sph_physics_shell_mex("transport_correction", pi, pj, dx, dy, r, dW, Vol, B, S["pos"],
                                                 prm.h, nf, nt, prm.transport_coeff)
                dt = verlet_time_step(S["vel"][:nf], prm.c_f, prm.h, prm.nu, prm.gravity_g, remain)
                if dt < 1e-14:
                    raise RuntimeError(f"unified Verlet dt collapsed (dt={dt:.2e}) at t={t:.6f} step={step}")
                rho, p, S["pos"], S["vel"], S["drho_dt"], force = sph_physics_shell_mex(
                    "integration_verlet", pi, pj, dx, dy, r, dW, Vol, B, rho, S["mass"], S["pos"], S["vel"],
                    S["drho_dt"], fp, dt, nf, nt, prm.rho0, prm.p0, prm.c_f, S["wall_vel"])
                t += dt
                periodic_bounding(S["pos"], nf, prm.DL)
                S["vel"][nf:, :] = 0.0
                nb = sph_neighbor_search_mex(S["pos"], nf, nt, prm.h, prm.DL)
                pi, pj, dx, dy, r, W, dW = nb
                tau_b, tau_t = sph_physics_shell_mex("wall_shear_monitor", pi, pj, dx, dy, r, dW, S["pos"], S["vel"],
                                                     S["wall_vel"], Vol, B, nf, prm.DL, prm.DH, prm.mu, prm.h)
                if log and log_every and step % log_every == 0:
                    log(f"step={step}, t={t:.6f}/{prm.t_end:.6f}, dt={dt:.4e}, pairs={len(pi)}")
            _, u = compute_mid_channel_profile(S["pos"][:nf], S["vel"][:nf, 0], prm.DL, prm.DH, mid_x, mid_hw, n_bins)
            profile_times.append(t)
            mid_profiles.append(u)
            full_profiles.append(final_profile(S["pos"][:nf], S["vel"][:nf, 0], prm)[1])
        wall = time.perf_counter() - t0
        pos, vel = S["pos"], S["vel"]
    else:
        raise ValueError("engine must be 'resident' or 'mex'")
    fluid_pos = pos[:nf].copy()
    fluid_pos[:, 0] = np.mod(fluid_pos[:, 0], prm.DL)
    y_mid, u_mean, u_exact = final_profile(fluid_pos, vel[:nf, 0], prm)
    if postprocess_path:
        restart.save_postprocess_data(postprocess_path, restart.make_postprocess_data(
            prm, nf, pos, vel, n_bins, profile_times, np.column_stack([np.nan_to_num(u, nan=np.nan) for u in mid_profiles])),
                                      fmt=mat_format)
    return RunResult(prm=prm, n_fluid=nf, n_total=nt, t=t, steps=int(step), wall_seconds=wall, pos=pos, vel=vel,
                     y_mid=y_mid, u_mean=u_mean, u_exact=u_exact, L2_error=l2_error(u_mean, u_exact),
                     profile_times=profile_times, mid_profile_u=mid_profiles, tau_bottom=tau_b, tau_top=tau_t,
                     tau_target=prm.gravity_g * prm.rho0 * prm.DH / 2, grid_policy=policy, full_profile_u=full_profiles, n_inner=n_inner)
