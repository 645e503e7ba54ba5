"""ctypes binding of libsphx.so (include/sphx.h).  Plumbing only: numpy float64 column-major arrays
in, numpy arrays out.  There is no CPU fallback: if the shared library is missing or no HIP device is
present the calls raise."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPHX_LIB") or os.path.join(HERE, "csrc", "libsphx.so")  # SPHX_LIB: experiment builds
_dp = C.POINTER(C.c_double)

SPHX_OK = 0
SPHX_ERR_ARG, SPHX_ERR_DEVICE, SPHX_ERR_STATE, SPHX_ERR_DIVERGED, SPHX_ERR_GRID = -1, -2, -3, -4, -5


class SphxError(RuntimeError):
    """Raised for any non-zero status; .identifier is the MEX-style error id."""

    def __init__(self, code, identifier, message):
        super().__init__(f"[{identifier}] {message} (status {code})")
        self.code = code
        self.identifier = identifier
        self.message = message


class SphxParams(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("DL", "DH", "dp", "h", "rho0", "mu", "c_f", "p0", "inv_sigma0",
                                           "gravity_g", "transport_coeff", "t_end")] + \
               [("sort_interval", C.c_int32), ("lanes_per_particle", C.c_int32),
                ("steps_per_graph", C.c_int32), ("dual_rate", C.c_int32), ("rebuild_every", C.c_int32),
                ("dynamic_rebin", C.c_int32), ("skin_h", C.c_double)]


class SphxStatus(C.Structure):
    _fields_ = [("t", C.c_double), ("dt_last", C.c_double), ("dt_next", C.c_double), ("vmax", C.c_double),
                ("step", C.c_int64), ("done", C.c_int32), ("device_status", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


EXPORTS = [
    "sphx_version", "sphx_last_error", "sphx_last_error_id", "sphx_device_count", "sphx_set_device",
    "sphx_neighbor_search", "sphx_neighbor_fetch", "sphx_density_correction", "sphx_viscous_force",
    "sphx_transport_correction", "sphx_integration_1st", "sphx_integration_2nd",
    "sphx_integration_verlet", "sphx_advance_shell_step", "sphx_wall_shear_monitor",
    "sphx_ctx_create", "sphx_ctx_destroy", "sphx_ctx_advance", "sphx_ctx_enqueue_steps", "sphx_ctx_sync",
    "sphx_ctx_prepare_steps", "sphx_ctx_graph_stats",
    "sphx_ctx_download", "sphx_ctx_monitor", "sphx_ctx_neighbor_list", "sphx_ctx_profile_enable",
    "sphx_ctx_profile_read", "sphx_ctx_info", "sphx_ctx_tuning", "sphx_ctx_substeps", "sphx_ctx_schedule", "sphx_ctx_kernel_forms", "sphx_ctx_grid_policy", "sphx_ctx_time_kernel",
    "sphx_slab_create", "sphx_slab_layout", "sphx_slab_local_vmax", "sphx_slab_prepare", "sphx_slab_compute",
    "sphx_slab_finish", "sphx_slab_sync", "sphx_slab_snapshot", "sphx_comm_available", "sphx_comm_unique_id", "sphx_comm_selftest", "sphx_comm_selftest_graph", "sphx_slab_comm_init",
    "sphx_slab_comm_destroy", "sphx_slab_run", "sphx_slab_group_run", "sphx_slab_graph_prepare",
]

_LIB = None


def lib() -> C.CDLL:
    """Load libsphx.so; fail loudly when it has not been built (see __graft_entry__.build)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` "
                              f"(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name in ("sphx_version", "sphx_last_error", "sphx_last_error_id"):
            getattr(L, name).restype = C.c_char_p
        L.sphx_ctx_destroy.restype = None
        _LIB = L
    return _LIB


def check(rc: int) -> None:
    if rc != SPHX_OK:
        L = lib()
        raise SphxError(rc, (L.sphx_last_error_id() or b"").decode(), (L.sphx_last_error() or b"").decode())


def f64(a, fortran=True):
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 2:
        return np.asfortranarray(a) if fortran else np.ascontiguousarray(a)
    return np.ascontiguousarray(a)


def ptr(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def device_count() -> int:
    return int(lib().sphx_device_count())


def set_device(dev: int) -> None:
    check(lib().sphx_set_device(C.c_int(dev)))


def make_params(prm, t_end=None, transport_coeff=None, lanes_per_particle=0, steps_per_graph=0, rebuild_every=0,
                skin_h=0.0, dynamic_rebin=0, dual_rate=0) -> SphxParams:
    return SphxParams(DL=prm.DL, DH=prm.DH, dp=prm.dp, h=prm.h, rho0=prm.rho0, mu=prm.mu, c_f=prm.c_f,
                      p0=prm.p0, inv_sigma0=prm.inv_sigma0, gravity_g=prm.gravity_g,
                      transport_coeff=prm.transport_coeff if transport_coeff is None else transport_coeff,
                      t_end=prm.t_end if t_end is None else t_end, sort_interval=int(prm.sort_interval),
                      lanes_per_particle=int(lanes_per_particle), steps_per_graph=int(steps_per_graph),
                      dual_rate=int(dual_rate), rebuild_every=int(rebuild_every), dynamic_rebin=int(dynamic_rebin), skin_h=float(skin_h))


class Context:
    """Device-resident simulation state (sphx_ctx)."""

    def __init__(self, prm, n_fluid, n_total, pos, vel, drho_dt, mass, wall_vel, t0=0.0, step0=0,
                 t_end=None, transport_coeff=None, lanes_per_particle=0, steps_per_graph=0, rebuild_every=0,
                 skin_h=0.0, dynamic_rebin=0, dual_rate=0):
        self._h = C.c_void_p()
        self.n_fluid, self.n_total = int(n_fluid), int(n_total)
        self.params = make_params(prm, t_end, transport_coeff, lanes_per_particle, steps_per_graph, rebuild_every,
                                  skin_h, dynamic_rebin, dual_rate)
        pos, vel, wall_vel = f64(pos), f64(vel), f64(wall_vel)
        drho_dt, mass = f64(drho_dt), f64(mass)
        assert pos.shape == (n_total, 2) and vel.shape == (n_total, 2) and wall_vel.shape == (n_total, 2)
        assert drho_dt.shape == (n_total,) and mass.shape == (n_total,)
        check(lib().sphx_ctx_create(C.byref(self._h), C.byref(self.params), C.c_int(n_fluid), C.c_int(n_total),
                                    ptr(pos), ptr(vel), ptr(drho_dt), ptr(mass), ptr(wall_vel),
                                    C.c_double(t0), C.c_int64(step0)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().sphx_ctx_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def advance(self, t_target, max_steps=0) -> dict:
        st = SphxStatus()
        check(lib().sphx_ctx_advance(self._h, C.c_double(t_target), C.c_int64(max_steps), C.byref(st)))
        return st.as_dict()

    def enqueue_steps(self, n_steps):
        check(lib().sphx_ctx_enqueue_steps(self._h, C.c_int64(n_steps)))

    def sync(self) -> dict:
        st = SphxStatus()
        check(lib().sphx_ctx_sync(self._h, C.byref(st)))
        return st.as_dict()

    def prepare_steps(self, n_steps):
        """Capture (without running) the graphs the next enqueue_steps(n_steps) / advance(max_steps=n_steps) replays."""
        check(lib().sphx_ctx_prepare_steps(self._h, C.c_int64(n_steps)))

    def graph_stats(self) -> dict:
        a, b, g = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        check(lib().sphx_ctx_graph_stats(self._h, C.byref(a), C.byref(b), C.byref(g)))
        return dict(slots_replayed=a.value, slots_eager=b.value, graphs_captured=g.value)

    def download(self, fields=("pos", "vel", "rho", "p", "drho_dt", "force", "force_prior", "Vol", "B")) -> dict:
        nt = self.n_total
        shapes = dict(pos=(nt, 2), vel=(nt, 2), rho=(nt,), p=(nt,), drho_dt=(nt,), force=(nt, 2),
                      force_prior=(nt, 2), Vol=(nt,), B=(nt, 4))
        order = ("pos", "vel", "rho", "p", "drho_dt", "force", "force_prior", "Vol", "B")
        out = {k: np.zeros(shapes[k], order="F") for k in fields}
        args = [ptr(out[k]) if k in out else None for k in order]
        check(lib().sphx_ctx_download(self._h, *args))
        return out

    def monitor(self, tau=True, pairs=False):
        tb, tt, npairs = C.c_double(0), C.c_double(0), C.c_double(0)
        check(lib().sphx_ctx_monitor(self._h, C.byref(tb) if tau else None, C.byref(tt) if tau else None,
                                     C.byref(npairs) if pairs else None))
        return tb.value, tt.value, npairs.value

    def neighbor_list(self):
        n = C.c_size_t(0)
        check(lib().sphx_ctx_neighbor_list(self._h, C.byref(n)))
        return _fetch_pairs(n.value)

    def info(self):
        a, b, c, d = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        check(lib().sphx_ctx_info(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return dict(n_fluid=a.value, n_wall=b.value, n_cell_x=c.value, n_cell_y=d.value)

    def grid_policy(self):
        a, b, c, d = C.c_int(0), C.c_double(0.0), C.c_int64(0), C.c_double(0.0)
        check(lib().sphx_ctx_grid_policy(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return dict(rebuild_every=a.value, skin=b.value, forced_rebuilds=c.value, drift=d.value)

    def tuning(self):
        a, b = C.c_int(0), C.c_int(0)
        check(lib().sphx_ctx_tuning(self._h, C.byref(a), C.byref(b)))
        return dict(lanes_per_particle=a.value, steps_per_graph=b.value)

    def schedule(self):
        """Launch schedule in force and the re-binnings carried out by step slots so far (sphx_ctx_schedule)."""
        a, b, d, r = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int64(0)
        check(lib().sphx_ctx_schedule(self._h, C.byref(a), C.byref(b), C.byref(d), C.byref(r)))
        return dict(fuse_ea=a.value, tail_clock=b.value, dynamic=d.value, rebins=r.value)

    def kernel_forms(self):
        """Which forms of the passes the context runs (sphx_ctx_kernel_forms)."""
        v = [C.c_int(0) for _ in range(4)]
        check(lib().sphx_ctx_kernel_forms(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("walk_kernels", "lds_tiles", "tiles_abe", "coded_lists"), (bool(x.value) for x in v)))

    def substeps(self) -> int:
        """Inner sub-steps per step slot (1 = the reference's single-rate loop, see sphx_params.dual_rate)."""
        n = C.c_int(0)
        check(lib().sphx_ctx_substeps(self._h, C.byref(n)))
        return n.value

    def time_kernel(self, name, reps=200) -> float:
        ms = C.c_double(0.0)
        check(lib().sphx_ctx_time_kernel(self._h, name.encode(), C.c_int(reps), C.byref(ms)))
        return ms.value

    def profile_enable(self, on=True):
        check(lib().sphx_ctx_profile_enable(self._h, C.c_int(1 if on else 0)))

    def profile_read(self) -> dict:
        cap = 32
        names = (C.c_char_p * cap)()
        avg = (C.c_double * cap)()
        cnt = (C.c_int64 * cap)()
        n = C.c_int(0)
        check(lib().sphx_ctx_profile_read(self._h, C.c_int(cap), names, avg, cnt, C.byref(n)))
        return {names[k].decode(): dict(avg_ms=avg[k], launches=cnt[k]) for k in range(min(n.value, cap))}


def _fetch_pairs(n):
    cols = [np.zeros(max(n, 1)) for _ in range(7)]
    check(lib().sphx_neighbor_fetch(*[ptr(c) for c in cols], C.c_size_t(max(n, 1))))
    return tuple(c[:n] for c in cols)


def neighbor_search(pos, n_fluid, n_total, h, DL):
    pos = f64(pos)
    n = C.c_size_t(0)
    check(lib().sphx_neighbor_search(ptr(pos), C.c_int(n_fluid), C.c_int(n_total), C.c_double(h), C.c_double(DL),
                                     C.byref(n)))
    return _fetch_pairs(n.value)
