"""Host-side mirror of the reference's MEX call surface, backed by libsphx.so (HIP, gfx950).

    [pair_i,pair_j,dx,dy,r,W,dW] = sph_neighbor_search_mex(pos, n_fluid, n_total, h, DL)
    [...] = sph_physics_shell_mex(mode, ...)          % 8 modes

Same names, argument order, arity and error identifiers as the gateways in
/root/reference/mex/sph_neighbor_search_mex.c:185-242 and /root/reference/mex/sph_physics_mex.c
(dispatcher :1745-1772, per-mode checks cited below), so the parity tests read like calls from
SPH_Poiseuille.m:167,169,366-432.  MATLAB's `nlhs` is passed as the keyword `nargout` (defaults to the
mode's full output count).  Errors surface as MexError(identifier, message) -- the analogue of
mexErrMsgIdAndTxt.  Arrays are float64; [n x 2]/[n x 4] matrices are column-major.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi


class MexError(RuntimeError):
    def __init__(self, identifier, message):
        super().__init__(f"{identifier}: {message}")
        self.identifier = identifier
        self.message = message


def _require(cond, identifier, message):
    if not cond:
        raise MexError(identifier, message)


def _call(fn, *args):
    rc = fn(*args)
    if rc != capi.SPHX_OK:
        L = capi.lib()
        raise MexError((L.sphx_last_error_id() or b"").decode(), (L.sphx_last_error() or b"").decode())


def _is_double_matrix(a):
    return isinstance(a, np.ndarray) and a.dtype == np.float64


def _numel(a):
    return int(np.asarray(a).size)


def _shape2(a):
    a = np.asarray(a)
    if a.ndim == 1:
        return (a.shape[0], 1)
    return tuple(a.shape[:2])


def _scalar(a):
    return float(np.asarray(a, dtype=np.float64).reshape(-1)[0])


def _vec(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))


def _mat(a):
    return capi.f64(np.asarray(a, dtype=np.float64))


P = capi.ptr


def sph_neighbor_search_mex(*args, nargout=7):
    """sph_neighbor_search_mex.c:185-421 -- 5 inputs, 7 outputs (column vectors)."""
    _require(len(args) == 5, "SPH:Neighbor:nrhs", "Expected 5 inputs.")
    _require(nargout == 7, "SPH:Neighbor:nlhs", "Expected 7 outputs.")
    pos, n_fluid, n_total, h, DL = args
    _require(_is_double_matrix(np.asarray(pos)) and np.asarray(pos).ndim == 2 and np.asarray(pos).shape[1] == 2,
             "SPH:Neighbor:pos", "pos must be a double matrix of size [n_total x 2].")
    n_fluid, n_total, h, DL = int(_scalar(n_fluid)), int(_scalar(n_total)), _scalar(h), _scalar(DL)
    n_rows = np.asarray(pos).shape[0]
    _require(not (n_total <= 0 or n_fluid <= 0 or n_fluid > n_total or n_total != n_rows),
             "SPH:Neighbor:count", "Invalid n_fluid/n_total or inconsistent pos size.")
    _require(not (h <= 0.0 or DL <= 0.0), "SPH:Neighbor:param", "h and DL must be positive.")
    try:
        return capi.neighbor_search(_mat(pos), n_fluid, n_total, h, DL)
    except capi.SphxError as e:
        raise MexError(e.identifier, e.message) from None


def _pairs(args, lo, with_W, ident="SPH:Physics:pairs"):
    """get_pair_data (sph_physics_mex.c:49-74): equal lengths, returned as contiguous vectors."""
    n = 7 if with_W else 6
    cols = [_vec(a) for a in args[lo:lo + n]]
    _require(all(len(c) == len(cols[0]) for c in cols), ident,
             "Pair arrays must have same length." if ident.endswith("pairs") and "density" not in ident
             else "Pair arrays mismatch.")
    return cols


def _density_correction(args, nargout):
    _require(len(args) == 14, "SPH:Physics:density:nrhs", "density_correction expects 13 inputs after mode.")
    _require(nargout == 3, "SPH:Physics:density:nlhs", "density_correction expects 3 outputs.")
    pi, pj, dx, dy, r, W, dW = _pairs(args, 1, True, "SPH:Physics:density:pairs")
    mass = _vec(args[8])
    _require(len(pi) <= 2147483647, "SPH:Physics:density:pairsize", "Pair count exceeds INT_MAX.")
    nf, nt = int(_scalar(args[9])), int(_scalar(args[10]))
    rho0, h, inv_sigma0 = _scalar(args[11]), _scalar(args[12]), _scalar(args[13])
    _require(nf > 0 and nt >= nf, "SPH:Physics:density:count", "Invalid n_fluid/n_total.")
    _require(len(mass) == nt, "SPH:Physics:density:mass", "mass size mismatch.")
    _require(rho0 > 0.0 and h > 0.0, "SPH:Physics:density:param", "rho0 and h must be positive.")
    rho, Vol, B = np.zeros(nt), np.zeros(nt), np.zeros((nt, 4), order="F")
    _call(capi.lib().sphx_density_correction, C.c_size_t(len(pi)), P(pi), P(pj), P(dx), P(dy), P(r), P(W), P(dW),
          P(mass), C.c_int(nf), C.c_int(nt), C.c_double(rho0), C.c_double(h), C.c_double(inv_sigma0),
          P(rho), P(Vol), P(B))
    return rho, Vol, B


def _viscous_force(args, nargout):
    _require(len(args) in (16, 17), "SPH:Physics:viscous:nrhs", "viscous_force expects 15 inputs after mode.")
    _require(nargout == 1, "SPH:Physics:viscous:nlhs", "viscous_force expects 1 output.")
    pi, pj, dx, dy, r, dW = _pairs(args, 1, False)
    vel, Vol, B = _mat(args[7]), _vec(args[8]), _mat(args[9])
    mu, h = _scalar(args[10]), _scalar(args[11])
    nf, nt = int(_scalar(args[12])), int(_scalar(args[13]))
    mass, wall_vel = _vec(args[14]), _mat(args[15])
    _require(_shape2(args[7]) == (nt, 2), "SPH:Physics:viscous:vel", "vel size mismatch.")
    _require(_numel(args[8]) == nt, "SPH:Physics:viscous:Vol", "Vol size mismatch.")
    _require(_shape2(args[9]) == (nt, 4), "SPH:Physics:viscous:B", "B size mismatch.")
    _require(_numel(args[14]) == nt, "SPH:Physics:viscous:mass", "mass size mismatch.")
    _require(_shape2(args[15]) == (nt, 2), "SPH:Physics:viscous:wallvel", "wall_vel size mismatch.")
    force = np.zeros((nt, 2), order="F")
    _call(capi.lib().sphx_viscous_force, C.c_size_t(len(pi)), P(pi), P(pj), P(dx), P(dy), P(r), P(dW), P(vel),
          P(Vol), P(B), C.c_double(mu), C.c_double(h), C.c_int(nf), C.c_int(nt), P(mass), P(wall_vel), P(force))
    return (force,)


def _transport_correction(args, nargout):
    _require(len(args) in (13, 14), "SPH:Physics:transport:nrhs",
             "transport_correction expects 12 or 13 inputs after mode.")
    _require(nargout == 1, "SPH:Physics:transport:nlhs", "transport_correction expects 1 output.")
    pi, pj, dx, dy, r, dW = _pairs(args, 1, False)
    Vol, B, pos = _vec(args[7]), _mat(args[8]), _mat(args[9])
    h, nf, nt = _scalar(args[10]), int(_scalar(args[11])), int(_scalar(args[12]))
    coeff = _scalar(args[13]) if len(args) == 14 else 0.2  # default, sph_physics_mex.c:584
    _require(coeff >= 0.0, "SPH:Physics:transport:coeff", "transport_coeff must be non-negative.")
    _require(_numel(args[7]) == nt, "SPH:Physics:transport:Vol", "Vol size mismatch.")
    _require(_shape2(args[8]) == (nt, 4), "SPH:Physics:transport:B", "B size mismatch.")
    _require(_shape2(args[9]) == (nt, 2), "SPH:Physics:transport:pos", "pos size mismatch.")
    out = np.zeros((nt, 2), order="F")
    _call(capi.lib().sphx_transport_correction, C.c_size_t(len(pi)), P(pi), P(pj), P(dx), P(dy), P(r), P(dW),
          P(Vol), P(B), P(pos), C.c_double(h), C.c_int(nf), C.c_int(nt), C.c_double(coeff), P(out))
    return (out,)


def _int1_like(args, tag):
    """shared unpacking/validation of integration_1st (:786-821) and integration_verlet (:1339-1361)."""
    pi, pj, dx, dy, r, dW = _pairs(args, 1, False)
    nt = int(_scalar(args[17]))
    nf = int(_scalar(args[16]))
    pre = "SPH:Physics:%s:" % tag
    if tag == "int1":
        _require(_shape2(args[8]) == (nt, 4), pre + "B", "B size mismatch.")
        _require(_numel(args[7]) == nt, pre + "Vol", "Vol size mismatch.")
    else:
        _require(_numel(args[7]) == nt, pre + "Vol", "Vol size mismatch.")
        _require(_shape2(args[8]) == (nt, 4), pre + "B", "B size mismatch.")
    _require(_numel(args[9]) == nt, pre + "rho", "rho size mismatch.")
    _require(_numel(args[10]) == nt, pre + "mass", "mass size mismatch.")
    _require(_shape2(args[11]) == (nt, 2), pre + "pos", "pos size mismatch.")
    _require(_shape2(args[12]) == (nt, 2), pre + "vel", "vel size mismatch.")
    _require(_numel(args[13]) == nt, pre + "drho", "drho size mismatch.")
    _require(_shape2(args[14]) == (nt, 2), pre + "force_prior", "force_prior size mismatch.")
    _require(_shape2(args[21]) == (nt, 2), pre + "wall_vel", "wall_vel size mismatch.")
    arrs = dict(Vol=_vec(args[7]), B=_mat(args[8]), rho=_vec(args[9]), mass=_vec(args[10]), pos=_mat(args[11]),
                vel=_mat(args[12]), drho=_vec(args[13]), fp=_mat(args[14]), wall_vel=_mat(args[21]))
    sc = dict(dt=_scalar(args[15]), nf=nf, nt=nt, rho0=_scalar(args[18]), p0=_scalar(args[19]), c_f=_scalar(args[20]))
    return (pi, pj, dx, dy, r, dW), arrs, sc


def _integration_1st(args, nargout):
    _require(len(args) == 22, "SPH:Physics:int1:nrhs", "integration_1st expects 21 inputs after mode.")
    _require(nargout == 5, "SPH:Physics:int1:nlhs", "integration_1st expects 5 outputs.")
    (pi, pj, dx, dy, r, dW), a, s = _int1_like(args, "int1")
    nt = s["nt"]
    rho, p, pos = np.zeros(nt), np.zeros(nt), np.zeros((nt, 2), order="F")
    force, drho = np.zeros((nt, 2), order="F"), np.zeros(nt)
    _call(capi.lib().sphx_integration_1st, C.c_size_t(len(pi)), P(pi), P(pj), P(dx), P(dy), P(r), P(dW), P(a["Vol"]),
          P(a["B"]), P(a["rho"]), P(a["mass"]), P(a["pos"]), P(a["vel"]), P(a["drho"]), P(a["fp"]),
          C.c_double(s["dt"]), C.c_int(s["nf"]), C.c_int(nt), C.c_double(s["rho0"]), C.c_double(s["p0"]),
          C.c_double(s["c_f"]), P(a["wall_vel"]), P(rho), P(p), P(pos), P(force), P(drho))
    return rho, p, pos, force, drho


def _integration_2nd(args, nargout):
    _require(len(args) == 15, "SPH:Physics:int2:nrhs", "integration_2nd expects 15 inputs after mode.")
    _require(nargout == 3, "SPH:Physics:int2:nlhs", "integration_2nd expects 3 outputs.")
    pi, pj, dx, dy, r, dW = _pairs(args, 1, False)
    dt, nf, nt = _scalar(args[11]), int(_scalar(args[12])), int(_scalar(args[13]))
    _require(_numel(args[7]) == nt, "SPH:Physics:int2:Vol", "Vol size mismatch.")
    _require(_numel(args[8]) == nt, "SPH:Physics:int2:rho", "rho size mismatch.")
    _require(_shape2(args[9]) == (nt, 2), "SPH:Physics:int2:pos", "pos size mismatch.")
    _require(_shape2(args[10]) == (nt, 2), "SPH:Physics:int2:vel", "vel size mismatch.")
    _require(_shape2(args[14]) == (nt, 2), "SPH:Physics:int2:wall_vel", "wall_vel size mismatch.")
    Vol, rho, pos, vel, wv = _vec(args[7]), _vec(args[8]), _mat(args[9]), _mat(args[10]), _mat(args[14])
    pos_o, drho_o, zeros_o = np.zeros((nt, 2), order="F"), np.zeros(nt), np.zeros((nt, 2), order="F")
    _call(capi.lib().sphx_integration_2nd, C.c_size_t(len(pi)), P(pi), P(pj), P(dx), P(dy), P(r), P(dW), P(Vol),
          P(rho), P(pos), P(vel), C.c_double(dt), C.c_int(nf), C.c_int(nt), P(wv), P(pos_o), P(drho_o), P(zeros_o))
    return pos_o, drho_o, zeros_o


def _integration_verlet(args, nargout):
    _require(len(args) == 22, "SPH:Physics:verlet:nrhs", "integration_verlet expects 21 inputs after mode.")
    _require(nargout == 6, "SPH:Physics:verlet:nlhs", "integration_verlet expects 6 outputs.")
    (pi, pj, dx, dy, r, dW), a, s = _int1_like(args, "verlet")
    nt = s["nt"]
    rho, p, pos, vel = np.zeros(nt), np.zeros(nt), np.zeros((nt, 2), order="F"), np.zeros((nt, 2), order="F")
    drho, force = np.zeros(nt), np.zeros((nt, 2), order="F")
    _call(capi.lib().sphx_integration_verlet, C.c_size_t(len(pi)), P(pi), P(pj), P(dx), P(dy), P(r), P(dW),
          P(a["Vol"]), P(a["B"]), P(a["rho"]), P(a["mass"]), P(a["pos"]), P(a["vel"]), P(a["drho"]), P(a["fp"]),
          C.c_double(s["dt"]), C.c_int(s["nf"]), C.c_int(nt), C.c_double(s["rho0"]), C.c_double(s["p0"]),
          C.c_double(s["c_f"]), P(a["wall_vel"]), P(rho), P(p), P(pos), P(vel), P(drho), P(force))
    return rho, p, pos, vel, drho, force


def _advance_shell_step(args, nargout):
    _require(len(args) == 24, "SPH:Physics:advance:nrhs", "advance_shell_step expects 23 inputs after mode.")
    _require(nargout == 9, "SPH:Physics:advance:nlhs", "advance_shell_step expects 9 outputs.")
    pi, pj, dx, dy, r, W, dW = _pairs(args, 1, True, "SPH:Physics:density:pairs")
    dt, nf, nt = _scalar(args[14]), int(_scalar(args[15])), int(_scalar(args[16]))
    _require(_numel(args[8]) == nt, "SPH:Physics:advance:mass", "mass size mismatch.")
    _require(_shape2(args[9]) == (nt, 2), "SPH:Physics:advance:pos", "pos size mismatch.")
    _require(_shape2(args[10]) == (nt, 2), "SPH:Physics:advance:vel", "vel size mismatch.")
    _require(_shape2(args[11]) == (nt, 2), "SPH:Physics:advance:wall_vel", "wall_vel size mismatch.")
    _require(_numel(args[12]) == nt, "SPH:Physics:advance:rho", "rho size mismatch.")
    _require(_numel(args[13]) == nt, "SPH:Physics:advance:drho_dt", "drho_dt size mismatch.")
    _require(nf > 0 and nt >= nf, "SPH:Physics:advance:count", "Invalid n_fluid/n_total.")
    mass, pos, vel, wv = _vec(args[8]), _mat(args[9]), _mat(args[10]), _mat(args[11])
    rho, drho = _vec(args[12]), _vec(args[13])
    rho0, p0, c_f, mu, h, inv_sigma0, g = (_scalar(args[k]) for k in range(17, 24))
    z1 = lambda: np.zeros(nt)
    z2 = lambda: np.zeros((nt, 2), order="F")
    o_rho, o_p, o_pos, o_vel, o_d, o_f, o_fp, o_Vol = z1(), z1(), z2(), z2(), z1(), z2(), z2(), z1()
    o_B = np.zeros((nt, 4), order="F")
    _call(capi.lib().sphx_advance_shell_step, C.c_size_t(len(pi)), P(pi), P(pj), P(dx), P(dy), P(r), P(W), P(dW),
          P(mass), P(pos), P(vel), P(wv), P(rho), P(drho), C.c_double(dt), C.c_int(nf), C.c_int(nt),
          C.c_double(rho0), C.c_double(p0), C.c_double(c_f), C.c_double(mu), C.c_double(h), C.c_double(inv_sigma0),
          C.c_double(g), P(o_rho), P(o_p), P(o_pos), P(o_vel), P(o_d), P(o_f), P(o_fp), P(o_Vol), P(o_B))
    return o_rho, o_p, o_pos, o_vel, o_d, o_f, o_fp, o_Vol, o_B


def _wall_shear_monitor(args, nargout):
    _require(len(args) == 17, "SPH:Physics:wallshear:nrhs", "wall_shear_monitor expects 16 inputs after mode.")
    _require(nargout == 2, "SPH:Physics:wallshear:nlhs", "wall_shear_monitor expects 2 outputs.")
    pi, pj, dx, dy, r, dW = _pairs(args, 1, False)
    nf = int(_scalar(args[12]))
    nt = _numel(args[10])
    DL, DH, mu, h = _scalar(args[13]), _scalar(args[14]), _scalar(args[15]), _scalar(args[16])
    _require(DL > 0.0 and h > 0.0, "SPH:Physics:wallshear:param", "DL and h must be positive.")
    _require(_shape2(args[7]) == (nt, 2), "SPH:Physics:wallshear:pos", "pos size mismatch.")
    _require(_shape2(args[8]) == (nt, 2), "SPH:Physics:wallshear:vel", "vel size mismatch.")
    _require(_shape2(args[9]) == (nt, 2), "SPH:Physics:wallshear:wall_vel", "wall_vel size mismatch.")
    _require(_shape2(args[11]) == (nt, 4), "SPH:Physics:wallshear:B", "B size mismatch.")
    pos, vel, wv, Vol, B = _mat(args[7]), _mat(args[8]), _mat(args[9]), _vec(args[10]), _mat(args[11])
    tb, tt = C.c_double(0.0), C.c_double(0.0)
    _call(capi.lib().sphx_wall_shear_monitor, C.c_size_t(len(pi)), P(pi), P(pj), P(dx), P(dy), P(r), P(dW), P(pos),
          P(vel), P(wv), P(Vol), P(B), C.c_int(nf), C.c_int(nt), C.c_double(DL), C.c_double(DH), C.c_double(mu),
          C.c_double(h), C.byref(tb), C.byref(tt))
    return tb.value, tt.value


_MODES = {  # mode -> (handler, default nargout); dispatcher sph_physics_mex.c:1753-1771
    "density_correction": (_density_correction, 3),
    "viscous_force": (_viscous_force, 1),
    "transport_correction": (_transport_correction, 1),
    "integration_1st": (_integration_1st, 5),
    "integration_2nd": (_integration_2nd, 3),
    "integration_verlet": (_integration_verlet, 6),
    "advance_shell_step": (_advance_shell_step, 9),
    "wall_shear_monitor": (_wall_shear_monitor, 2),
}


def sph_physics_shell_mex(*args, nargout=None):
    """sph_physics_mex.c:1745-1772.  args[0] is the mode string; returns the mode's outputs as a tuple
    (a bare array for single-output modes, like MATLAB)."""
    _require(len(args) >= 1, "SPH:Physics:nrhs", "At least mode input is required.")
    _require(isinstance(args[0], str), "SPH:Physics:mode", "First input must be mode string.")
    mode = args[0][:63]
    if mode not in _MODES:
        raise MexError("SPH:Physics:mode", "Unsupported mode.")
    handler, n_default = _MODES[mode]
    out = handler(args, n_default if nargout is None else nargout)
    return out[0] if len(out) == 1 else out
