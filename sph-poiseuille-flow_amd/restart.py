"""Restart and post-process files of the reference driver.

The reference writes two MAT-files at every output point / at the end of a run:
  restart.mat                      variables `state` (struct) and `config_signature` (char)
                                   (SPH_Poiseuille.m:295, 434-445, 607-610)
  SPH_Poiseuille_postprocess.mat   variable `postprocess_data` (struct of structs), consumed by
                                   SPH_Poiseuille_postprocess.m (SPH_Poiseuille.m:305-306, 612-639)
and resumes from restart.mat when the signature string and every array size match (:132-163).

The reference saves with '-v7.3' (HDF5).  This image has no HDF5 library, so the files are written here in MAT
level 5 format (scipy.io.savemat), which MATLAB's `load` -- the call the reference uses (:133) -- reads just the
same; files written BY the reference (v7.3) cannot be read here and `load_restart` says so instead of guessing.
Variable names, field names, shapes ([n x 2], [n x 1] columns, scalars) and the signature string are the
reference's, so a run of this package can be resumed or plotted by the unmodified MATLAB scripts.
"""
from __future__ import annotations

import os

import numpy as np

RESTART_FIELDS = ("pos", "vel", "rho", "p", "drho_dt", "force", "force_prior")  # make_restart_state, :434-445


class RestartError(RuntimeError):
    pass


def _col(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1, 1))


def _mat2(a):
    a = np.asarray(a, dtype=np.float64)
    if a.ndim != 2 or a.shape[1] != 2:
        raise RestartError(f"expected an [n x 2] array, got shape {a.shape}")
    return np.ascontiguousarray(a)


def make_restart_state(state: dict) -> dict:
    """SPH_Poiseuille.m:434-445: the nine fields, in the reference's shapes."""
    out = {"pos": _mat2(state["pos"]), "vel": _mat2(state["vel"]), "rho": _col(state["rho"]), "p": _col(state["p"]),
           "drho_dt": _col(state["drho_dt"]), "force": _mat2(state["force"]), "force_prior": _mat2(state["force_prior"]),
           "t": float(state["t"]), "step": float(int(state["step"]))}  # MATLAB numbers are doubles
    n = out["pos"].shape[0]
    for k in RESTART_FIELDS:
        if out[k].shape[0] != n:
            raise RestartError(f"state.{k} has {out[k].shape[0]} rows, pos has {n}")
    return out


def save_restart(path: str, config_signature: str, state: dict) -> None:
    """SPH_Poiseuille.m:607-610 (variables `state`, `config_signature`)."""
    from scipy.io import savemat
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:  # savemat appends ".mat" to bare names; a file object keeps the name exact
        savemat(f, {"state": make_restart_state(state), "config_signature": str(config_signature)}, format="5",
                oned_as="column")
    os.replace(tmp, path)  # an interrupted run never leaves half a restart file behind


def _is_hdf5(path: str) -> bool:
    with open(path, "rb") as f:
        head = f.read(520)
    return head[:8] == b"\x89HDF\r\n\x1a\n" or head[512:520] == b"\x89HDF\r\n\x1a\n"


def load_restart(path: str, n_total: int, config_signature: str):
    """The resume test of SPH_Poiseuille.m:132-163.  Returns (state dict, None) when the file can be resumed from,
    otherwise (None, reason) -- the reference prints the reason and starts from scratch."""
    if not os.path.exists(path):
        return None, "no restart file"
    if _is_hdf5(path):
        raise RestartError(f"{path} is a MAT v7.3 (HDF5) file as written by the reference; no HDF5 reader in this "
                           "environment -- re-save it in MATLAB with save(..., '-v7')")
    from scipy.io import loadmat
    data = loadmat(path, squeeze_me=False, struct_as_record=False)
    if "state" not in data or "config_signature" not in data:
        return None, "signature mismatch"  # can_resume is false, :134-135,161
    sig = data["config_signature"]
    sig = "".join(np.asarray(sig).ravel().tolist()) if not isinstance(sig, str) else sig
    if sig != config_signature:
        return None, "signature mismatch"
    st = data["state"][0, 0]
    want = {"pos": (n_total, 2), "vel": (n_total, 2), "rho": (n_total, 1), "p": (n_total, 1), "drho_dt": (n_total, 1),
            "force": (n_total, 2), "force_prior": (n_total, 2)}
    out = {}
    for k, shape in want.items():  # valid_state, :138-146
        if not hasattr(st, k) or tuple(np.asarray(getattr(st, k)).shape) != shape:
            return None, "incompatible state"
        out[k] = np.array(getattr(st, k), dtype=np.float64, order="F")
    if not hasattr(st, "t") or not hasattr(st, "step"):
        return None, "incompatible state"
    out["t"] = float(np.asarray(st.t).ravel()[0])
    out["step"] = int(round(float(np.asarray(st.step).ravel()[0])))
    for k in ("rho", "p", "drho_dt"):
        out[k] = out[k].ravel()
    return out, None


def make_postprocess_data(prm, n_fluid: int, pos, vel, n_bins: int, profile_times, mid_profile_u, result_png: str = "",
                          profile_evolution_png: str = "") -> dict:
    """SPH_Poiseuille.m:617-639."""
    from .profile import compute_binned_profile_mean
    pos, vel = _mat2(pos), _mat2(vel)
    fx = np.mod(pos[:n_fluid, 0], prm.DL)
    y_mid, u_mean = compute_binned_profile_mean(pos[:n_fluid, 1], vel[:n_fluid, 0], 0.0, prm.DH, n_bins)
    u_exact = prm.gravity_g / (2.0 * prm.nu) * y_mid * (prm.DH - y_mid)
    del fx  # the wrap only matters for the contour plot of the MATLAB script; the profile bins in y
    cfg = {k: (float(v) if not isinstance(v, str) else v) for k, v in prm.as_dict().items()}
    mid = np.asarray(mid_profile_u, dtype=np.float64)
    if mid.ndim == 2 and mid.shape[0] != n_bins:  # list of profiles -> [n_bins x n_times] columns, :265-266 of the driver
        mid = mid.T
    return {"cfg": cfg, "geom": {"n_fluid": float(n_fluid)}, "state": {"pos": pos, "vel": vel},
            "monitor": {"n_bins": float(n_bins), "profile_times": np.asarray(profile_times, dtype=np.float64).reshape(1, -1),
                        "mid_profile_u": np.ascontiguousarray(mid)},
            "final_profile": {"y_mid": _col(y_mid), "u_mean": _col(u_mean), "u_exact": _col(u_exact)},
            "output": {"result_png": result_png, "profile_evolution_png": profile_evolution_png}}


def save_postprocess_data(path: str, postprocess_data: dict) -> None:
    """SPH_Poiseuille.m:612-615 (variable `postprocess_data`)."""
    from scipy.io import savemat
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        savemat(f, {"postprocess_data": postprocess_data}, format="5", oned_as="column")
