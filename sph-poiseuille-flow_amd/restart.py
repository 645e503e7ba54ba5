"""Restart and post-process files of the reference driver.

The reference writes two MAT-files at every output point / at the end of a run:
  restart.mat                      variables `state` (struct) and `config_signature` (char)
                                   (SPH_Poiseuille.m:295, 434-445, 607-610)
  SPH_Poiseuille_postprocess.mat   variable `postprocess_data` (struct of structs), consumed by
                                   SPH_Poiseuille_postprocess.m (SPH_Poiseuille.m:305-306, 612-639)
and resumes from restart.mat when the signature string and every array size match (:132-163).

The reference saves with '-v7.3' (HDF5).  So does this module wherever a libhdf5 can be loaded (mat73.py: ctypes over
the system library -- there is no h5py in the image, but HDF5 1.10 ships in /opt/conda/lib), and it reads what the
reference wrote: a run started in MATLAB resumes here and the other way round.  Without the library the files are MAT
level 5 (scipy.io.savemat), which MATLAB's `load` -- the call the reference uses (:133) -- reads just the same, and a
v7.3 file is refused with a message that says why.  `fmt` = "7.3" | "5" | "auto" (7.3 when possible) on the writers;
the readers look at the file.  Variable names, field names, shapes ([n x 2], [n x 1] columns, scalars) and the
signature string are the reference's in both formats.
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


def _resolve_format(fmt: str) -> str:
    from . import mat73
    if fmt not in ("auto", "7.3", "5"):
        raise RestartError(f"fmt must be 'auto', '7.3' or '5', got {fmt!r}")
    if fmt == "auto":
        return "7.3" if mat73.available() else "5"
    if fmt == "7.3":
        mat73.lib()  # raises Mat73Unavailable (a RuntimeError that names the remedy) when there is no libhdf5
    return fmt


def _save_mat(path: str, variables: dict, fmt: str) -> None:
    fmt = _resolve_format(fmt)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    tmp = path + ".tmp"
    if fmt == "7.3":
        from . import mat73
        mat73.save(tmp, variables)
    else:
        from scipy.io import savemat
        with open(tmp, "wb") as f:  # savemat appends ".mat" to bare names; a file object keeps the name exact
            savemat(f, variables, format="5", oned_as="column")
    os.replace(tmp, path)  # an interrupted run never leaves half a file behind


def save_restart(path: str, config_signature: str, state: dict, fmt: str = "auto") -> None:
    """SPH_Poiseuille.m:607-610 (variables `state`, `config_signature`)."""
    _save_mat(path, {"state": make_restart_state(state), "config_signature": str(config_signature)}, fmt)


def _is_hdf5(path: str) -> bool:
    from . import mat73
    return mat73.is_mat73(path)


def _load_mat(path: str, names) -> dict:
    """{variable: value} with structs as dicts of arrays in MATLAB's shapes, whichever of the two formats the file has."""
    if _is_hdf5(path):
        from . import mat73
        try:
            return mat73.load(path, names)
        except mat73.Mat73Unavailable as e:
            raise RestartError(f"{path} is a MAT v7.3 (HDF5) file as the reference writes it, and {e}") from e
        except mat73.Mat73Error as e:
            raise RestartError(f"{path}: {e}") from e
    from scipy.io import loadmat
    raw = loadmat(path, squeeze_me=False, struct_as_record=False, variable_names=names)

    def plain(v):
        if isinstance(v, np.ndarray) and v.dtype == object and v.shape == (1, 1) and hasattr(v[0, 0], "_fieldnames"):
            return {k: plain(getattr(v[0, 0], k)) for k in v[0, 0]._fieldnames}
        if isinstance(v, np.ndarray) and v.dtype.kind in "US":
            return "".join(np.asarray(v).ravel().tolist())
        return v

    return {k: plain(v) for k, v in raw.items() if not k.startswith("__")}


def load_restart(path: str, n_total: int, config_signature: str):
    """The resume test of SPH_Poiseuille.m:132-163.  Returns (state dict, None) when the file can be resumed from,
    otherwise (None, reason) -- the reference prints the reason and starts from scratch."""
    if not os.path.exists(path):
        return None, "no restart file"
    data = _load_mat(path, ["state", "config_signature"])
    if "state" not in data or "config_signature" not in data or not isinstance(data["state"], dict):
        return None, "signature mismatch"  # can_resume is false, :134-135,161
    if data["config_signature"] != config_signature:
        return None, "signature mismatch"
    st = data["state"]
    want = {"pos": (n_total, 2), "vel": (n_total, 2), "rho": (n_total, 1), "p": (n_total, 1), "drho_dt": (n_total, 1),
            "force": (n_total, 2), "force_prior": (n_total, 2)}
    out = {}
    for k, shape in want.items():  # valid_state, :138-146
        if k not in st or tuple(np.asarray(st[k]).shape) != shape:
            return None, "incompatible state"
        out[k] = np.array(st[k], dtype=np.float64, order="F")
    if "t" not in st or "step" not in st:
        return None, "incompatible state"
    out["t"] = float(np.asarray(st["t"]).ravel()[0])
    out["step"] = int(round(float(np.asarray(st["step"]).ravel()[0])))
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


def save_postprocess_data(path: str, postprocess_data: dict, fmt: str = "auto") -> None:
    """SPH_Poiseuille.m:612-615 (variable `postprocess_data`)."""
    _save_mat(path, {"postprocess_data": postprocess_data}, fmt)


def load_postprocess_data(path: str) -> dict:
    """What SPH_Poiseuille_postprocess.m loads: the `postprocess_data` struct as nested dicts (either format)."""
    data = _load_mat(path, ["postprocess_data"])
    if not isinstance(data.get("postprocess_data"), dict):
        raise RestartError(f"{path} holds no postprocess_data struct")
    return data["postprocess_data"]
