"""ctypes front-end of the CPU oracle (oracle/sph_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path never does.  Array conventions are MATLAB's: float64, column-major, [n x 2] / [n x 4].
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)


class OrcPairs(C.Structure):
    _fields_ = [("count", C.c_size_t), ("capacity", C.c_size_t), ("pair_i", _dp), ("pair_j", _dp),
                ("dx", _dp), ("dy", _dp), ("r", _dp), ("W", _dp), ("dW", _dp)]


class OrcRunConfig(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("DL", "DH", "rho0", "mu", "c_f", "h", "p0", "inv_sigma0",
                                           "gravity_g", "transport_coeff", "t_end", "output_interval")] + \
               [("sort_interval", C.c_int), ("enable_sort", C.c_int), ("max_steps", C.c_long),
                ("log_every", C.c_int)]


class OrcRunStats(C.Structure):
    _fields_ = [("steps", C.c_long), ("t", C.c_double), ("seconds_neighbor", C.c_double),
                ("seconds_physics", C.c_double), ("seconds_total", C.c_double),
                ("tau_bottom", C.c_double), ("tau_top", C.c_double), ("vmax", C.c_double),
                ("dt_last", C.c_double), ("n_pairs_last", C.c_double)]


def build(force: bool = False) -> None:
    """Compile both oracle flavours with gcc (see oracle/Makefile)."""
    targets = [os.path.join(_HERE, n) for n in ("libsph_oracle.so", "libsph_oracle_omp.so")]
    src = os.path.join(_HERE, "sph_oracle.c")
    if not force and all(os.path.exists(t) and os.path.getmtime(t) >= os.path.getmtime(src) for t in targets):
        return
    subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])


_LIBS: dict = {}


def lib(omp: bool = False) -> C.CDLL:
    key = bool(omp)
    if key not in _LIBS:
        path = os.path.join(_HERE, "libsph_oracle_omp.so" if omp else "libsph_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_neighbor_search.restype = C.c_int
        L.orc_run.restype = C.c_int
        L.orc_verlet_time_step.restype = C.c_double
        L.orc_num_threads.restype = C.c_int
        _LIBS[key] = L
    return _LIBS[key]


def _f(a, ncol=None):
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 2:
        a = np.asfortranarray(a)
    else:
        a = np.ascontiguousarray(a)
    return a


def _p(a):
    return a.ctypes.data_as(_dp)


def neighbor_search(pos, n_fluid, n_total, h, DL, omp=False):
    """-> (pair_i, pair_j, dx, dy, r, W, dW) as float64 vectors (1-based indices stored as doubles)."""
    pos = _f(pos)
    assert pos.shape == (n_total, 2)
    out = OrcPairs()
    rc = lib(omp).orc_neighbor_search(_p(pos), C.c_int(n_fluid), C.c_int(n_total), C.c_double(h),
                                      C.c_double(DL), C.byref(out))
    if rc != 0:
        raise RuntimeError(f"orc_neighbor_search failed rc={rc}")
    n = out.count
    cols = tuple(np.ctypeslib.as_array(getattr(out, k), shape=(max(n, 1),))[:n].copy()
                 for k in ("pair_i", "pair_j", "dx", "dy", "r", "W", "dW"))
    lib(omp).orc_pairs_free(C.byref(out))
    return cols


def _pairs6(nb):
    pi, pj, dx, dy, r, W, dW = [_f(a) for a in nb]
    return pi, pj, dx, dy, r, W, dW


def density_correction(nb, mass, n_fluid, n_total, rho0, h, inv_sigma0, omp=False):
    pi, pj, dx, dy, r, W, dW = _pairs6(nb)
    mass = _f(mass)
    rho = np.zeros(n_total); Vol = np.zeros(n_total); B = np.zeros((n_total, 4), order="F")
    lib(omp).orc_density_correction(C.c_size_t(len(pi)), _p(pi), _p(pj), _p(dx), _p(dy), _p(r), _p(W),
                                    _p(dW), _p(mass), C.c_int(n_fluid), C.c_int(n_total),
                                    C.c_double(rho0), C.c_double(h), C.c_double(inv_sigma0), _p(rho),
                                    _p(Vol), _p(B))
    return rho, Vol, B


def viscous_force(nb, vel, Vol, B, mu, h, n_fluid, n_total, mass, wall_vel, omp=False):
    pi, pj, dx, dy, r, W, dW = _pairs6(nb)
    vel, Vol, B, mass, wall_vel = _f(vel), _f(Vol), _f(B), _f(mass), _f(wall_vel)
    force = np.zeros((n_total, 2), order="F")
    lib(omp).orc_viscous_force(C.c_size_t(len(pi)), _p(pi), _p(pj), _p(dx), _p(dy), _p(r), _p(dW),
                               _p(vel), _p(Vol), _p(B), C.c_double(mu), C.c_double(h), C.c_int(n_fluid),
                               C.c_int(n_total), _p(mass), _p(wall_vel), _p(force))
    return force


def transport_correction(nb, Vol, B, pos, h, n_fluid, n_total, transport_coeff=0.2, omp=False):
    pi, pj, dx, dy, r, W, dW = _pairs6(nb)
    Vol, B, pos = _f(Vol), _f(B), _f(pos)
    out = np.zeros((n_total, 2), order="F")
    lib(omp).orc_transport_correction(C.c_size_t(len(pi)), _p(pi), _p(pj), _p(dx), _p(dy), _p(r), _p(dW),
                                      _p(Vol), _p(B), _p(pos), C.c_double(h), C.c_int(n_fluid),
                                      C.c_int(n_total), C.c_double(transport_coeff), _p(out))
    return out


def integration_1st(nb, Vol, B, rho, mass, pos, vel, drho_dt, force_prior, dt, n_fluid, n_total,
                    rho0, p0, c_f, wall_vel, omp=False):
    pi, pj, dx, dy, r, W, dW = _pairs6(nb)
    Vol, B, rho, mass, pos, vel, drho_dt, force_prior, wall_vel = map(
        _f, (Vol, B, rho, mass, pos, vel, drho_dt, force_prior, wall_vel))
    rho_o = np.zeros(n_total); p_o = np.zeros(n_total); pos_o = np.zeros((n_total, 2), order="F")
    f_o = np.zeros((n_total, 2), order="F"); d_o = np.zeros(n_total)
    lib(omp).orc_integration_1st(C.c_size_t(len(pi)), _p(pi), _p(pj), _p(dx), _p(dy), _p(r), _p(dW),
                                 _p(Vol), _p(B), _p(rho), _p(mass), _p(pos), _p(vel), _p(drho_dt),
                                 _p(force_prior), C.c_double(dt), C.c_int(n_fluid), C.c_int(n_total),
                                 C.c_double(rho0), C.c_double(p0), C.c_double(c_f), _p(wall_vel),
                                 _p(rho_o), _p(p_o), _p(pos_o), _p(f_o), _p(d_o))
    return rho_o, p_o, pos_o, f_o, d_o


def integration_2nd(nb, Vol, rho, pos, vel, dt, n_fluid, n_total, wall_vel, omp=False):
    pi, pj, dx, dy, r, W, dW = _pairs6(nb)
    Vol, rho, pos, vel, wall_vel = map(_f, (Vol, rho, pos, vel, wall_vel))
    pos_o = np.zeros((n_total, 2), order="F"); d_o = np.zeros(n_total)
    z_o = np.zeros((n_total, 2), order="F")
    lib(omp).orc_integration_2nd(C.c_size_t(len(pi)), _p(pi), _p(pj), _p(dx), _p(dy), _p(r), _p(dW),
                                 _p(Vol), _p(rho), _p(pos), _p(vel), C.c_double(dt), C.c_int(n_fluid),
                                 C.c_int(n_total), _p(wall_vel), _p(pos_o), _p(d_o), _p(z_o))
    return pos_o, d_o, z_o


def integration_verlet(nb, Vol, B, rho, mass, pos, vel, drho_dt, force_prior, dt, n_fluid, n_total,
                       rho0, p0, c_f, wall_vel, omp=False):
    pi, pj, dx, dy, r, W, dW = _pairs6(nb)
    Vol, B, rho, mass, pos, vel, drho_dt, force_prior, wall_vel = map(
        _f, (Vol, B, rho, mass, pos, vel, drho_dt, force_prior, wall_vel))
    rho_o = np.zeros(n_total); p_o = np.zeros(n_total); pos_o = np.zeros((n_total, 2), order="F")
    vel_o = np.zeros((n_total, 2), order="F"); d_o = np.zeros(n_total)
    f_o = np.zeros((n_total, 2), order="F")
    lib(omp).orc_integration_verlet(C.c_size_t(len(pi)), _p(pi), _p(pj), _p(dx), _p(dy), _p(r), _p(dW),
                                    _p(Vol), _p(B), _p(rho), _p(mass), _p(pos), _p(vel), _p(drho_dt),
                                    _p(force_prior), C.c_double(dt), C.c_int(n_fluid), C.c_int(n_total),
                                    C.c_double(rho0), C.c_double(p0), C.c_double(c_f), _p(wall_vel),
                                    _p(rho_o), _p(p_o), _p(pos_o), _p(vel_o), _p(d_o), _p(f_o))
    return rho_o, p_o, pos_o, vel_o, d_o, f_o


def advance_shell_step(nb, mass, pos, vel, wall_vel, rho, drho_dt, dt, n_fluid, n_total, rho0, p0,
                       c_f, mu, h, inv_sigma0, gravity_g, omp=False):
    pi, pj, dx, dy, r, W, dW = _pairs6(nb)
    mass, pos, vel, wall_vel, rho, drho_dt = map(_f, (mass, pos, vel, wall_vel, rho, drho_dt))
    z1 = lambda: np.zeros(n_total)
    z2 = lambda: np.zeros((n_total, 2), order="F")
    rho_o, p_o, pos_o, vel_o, d_o, f_o, fp_o, Vol_o = z1(), z1(), z2(), z2(), z1(), z2(), z2(), z1()
    B_o = np.zeros((n_total, 4), order="F")
    lib(omp).orc_advance_shell_step(C.c_size_t(len(pi)), _p(pi), _p(pj), _p(dx), _p(dy), _p(r), _p(W),
                                    _p(dW), _p(mass), _p(pos), _p(vel), _p(wall_vel), _p(rho),
                                    _p(drho_dt), C.c_double(dt), C.c_int(n_fluid), C.c_int(n_total),
                                    C.c_double(rho0), C.c_double(p0), C.c_double(c_f), C.c_double(mu),
                                    C.c_double(h), C.c_double(inv_sigma0), C.c_double(gravity_g),
                                    _p(rho_o), _p(p_o), _p(pos_o), _p(vel_o), _p(d_o), _p(f_o),
                                    _p(fp_o), _p(Vol_o), _p(B_o))
    return rho_o, p_o, pos_o, vel_o, d_o, f_o, fp_o, Vol_o, B_o


def wall_shear_monitor(nb, pos, vel, wall_vel, Vol, B, n_fluid, DL, DH, mu, h, omp=False):
    pi, pj, dx, dy, r, W, dW = _pairs6(nb)
    pos, vel, wall_vel, Vol, B = map(_f, (pos, vel, wall_vel, Vol, B))
    n_total = len(Vol)
    tb, tt = C.c_double(0.0), C.c_double(0.0)
    lib(omp).orc_wall_shear_monitor(C.c_size_t(len(pi)), _p(pi), _p(pj), _p(dx), _p(dy), _p(r), _p(dW),
                                    _p(pos), _p(vel), _p(wall_vel), _p(Vol), _p(B), C.c_int(n_fluid),
                                    C.c_int(n_total), C.c_double(DL), C.c_double(DH), C.c_double(mu),
                                    C.c_double(h), C.byref(tb), C.byref(tt))
    return tb.value, tt.value


def verlet_time_step(vel, n_fluid, c_max, h, nu, gravity_g, remain):
    vel = _f(vel)
    return lib().orc_verlet_time_step(_p(vel), C.c_int(n_fluid), C.c_int(vel.shape[0]),
                                      C.c_double(c_max), C.c_double(h), C.c_double(nu),
                                      C.c_double(gravity_g), C.c_double(remain))


def run(prm, parts, t_end=None, output_interval=None, max_steps=0, enable_sort=True, omp=False,
        pos=None, vel=None, drho_dt=None, t0=0.0, step0=0, log_every=0):
    """Run the reference time loop (SPH_Poiseuille.m:246-302) on the oracle.

    Returns a dict with the final state (rows in the oracle's current order), `order` (row -> initial
    row index), and the stats struct."""
    nf, nt = parts["n_fluid"], parts["n_total"]
    cfg = OrcRunConfig(DL=prm.DL, DH=prm.DH, rho0=prm.rho0, mu=prm.mu, c_f=prm.c_f, h=prm.h, p0=prm.p0,
                       inv_sigma0=prm.inv_sigma0, gravity_g=prm.gravity_g,
                       transport_coeff=prm.transport_coeff,
                       t_end=prm.t_end if t_end is None else t_end,
                       output_interval=prm.output_interval if output_interval is None else output_interval,
                       sort_interval=prm.sort_interval, enable_sort=int(bool(enable_sort)),
                       max_steps=int(max_steps), log_every=int(log_every))
    st = dict(
        pos=_f(parts["pos"] if pos is None else pos).copy(order="F"),
        vel=_f(parts["vel"] if vel is None else vel).copy(order="F"),
        drho_dt=_f(parts["drho_dt"] if drho_dt is None else drho_dt).copy(),
        mass=_f(parts["mass"]).copy(), wall_vel=_f(parts["wall_vel"]).copy(order="F"),
        rho=np.zeros(nt), p=np.zeros(nt), force=np.zeros((nt, 2), order="F"),
        force_prior=np.zeros((nt, 2), order="F"), Vol=np.zeros(nt), B=np.zeros((nt, 4), order="F"))
    order = np.arange(nt, dtype=np.int32)
    stats = OrcRunStats()
    rc = lib(omp).orc_run(C.byref(cfg), C.c_int(nf), C.c_int(nt), _p(st["pos"]), _p(st["vel"]),
                          _p(st["drho_dt"]), _p(st["mass"]), _p(st["wall_vel"]), _p(st["rho"]),
                          _p(st["p"]), _p(st["force"]), _p(st["force_prior"]), _p(st["Vol"]), _p(st["B"]),
                          order.ctypes.data_as(C.POINTER(C.c_int)), C.c_double(t0), C.c_long(step0),
                          C.byref(stats))
    if rc != 0:
        raise RuntimeError(f"orc_run failed rc={rc}")
    st["order"] = order
    st["stats"] = {k: getattr(stats, k) for k, _ in OrcRunStats._fields_}
    return st


def set_num_threads(n: int, omp=True) -> None:
    lib(omp).orc_set_num_threads(C.c_int(int(n)))


def num_threads(omp=True) -> int:
    return lib(omp).orc_num_threads()
