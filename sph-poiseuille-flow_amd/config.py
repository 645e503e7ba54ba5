"""config.ini reader and derived parameters -- host-side mirror of SPH_Poiseuille.m S2.

Follows /root/reference/SPH_Poiseuille.m:43-91 (parameter derivation) and :447-512 (INI parser):
sections in [..], '#'/';' comment lines, trailing comments stripped after the value, numeric values
through str2double semantics, required keys fetched with get_ini_numeric (error when missing / not
finite).  Environment override SPH_CONFIG_OVERRIDE is honoured like :19.
"""
from __future__ import annotations

import math
import os
import re
from dataclasses import dataclass, asdict


class ConfigError(ValueError):
    pass


def _is_varname(s: str) -> bool:
    return re.fullmatch(r"[A-Za-z][A-Za-z0-9_]*", s) is not None and len(s) <= 63


def _str2double(s: str):
    """MATLAB str2double for the plain decimal/exponent forms config files use; NaN -> None."""
    t = s.strip()
    if re.fullmatch(r"[+-]?(\d+\.?\d*([eEdD][+-]?\d+)?|\.\d+([eEdD][+-]?\d+)?|[iI]nf|NaN|nan)", t) is None:
        return None
    t = t.replace("d", "e").replace("D", "e")
    try:
        v = float(t)
    except ValueError:
        return None
    return None if math.isnan(v) else v


def parse_ini(filename: str) -> dict:
    """SPH_Poiseuille.m:447-499."""
    if not os.path.isfile(filename):
        raise ConfigError(f"config file does not exist: {filename}")
    with open(filename, "r", encoding="utf-8") as f:
        text = f.read()
    cfg: dict = {}
    section = ""
    for raw in re.split(r"\r\n|\n|\r", text):
        line = raw.strip()
        if not line or line.startswith(";") or line.startswith("#"):
            continue
        if line.startswith("[") and line.endswith("]"):
            section = line[1:-1].strip()
            if not _is_varname(section):
                raise ConfigError(f"illegal INI section name: {section}")
            cfg.setdefault(section, {})
            continue
        eq = line.find("=")
        if eq < 0:
            continue
        key = line[:eq].strip()
        val = re.sub(r"[;#].*$", "", line[eq + 1:].strip()).strip()
        if not section:
            raise ConfigError(f"INI key outside any section: {line}")
        if not _is_varname(key):
            raise ConfigError(f"illegal INI key: {key}")
        num = _str2double(val)
        cfg[section][key] = num if num is not None else val
    return cfg


def get_ini_numeric(cfg: dict, section: str, key: str) -> float:
    """SPH_Poiseuille.m:501-512."""
    if section not in cfg:
        raise ConfigError(f"missing section: [{section}]")
    if key not in cfg[section]:
        raise ConfigError(f"missing key: [{section}].{key}")
    v = cfg[section][key]
    if not isinstance(v, float) or not math.isfinite(v):
        raise ConfigError(f"key [{section}].{key} is not a valid number")
    return v


def _matlab_round(x: float) -> float:
    return math.floor(abs(x) + 0.5) * (1.0 if x >= 0 else -1.0)


@dataclass
class SimParams:
    """Everything S2 derives (SPH_Poiseuille.m:46-91,175-196)."""
    DL: float
    DH: float
    dp: float
    rho0: float
    mu: float
    U_bulk: float
    c_f: float
    t_end: float
    output_interval: float
    sort_interval: int
    restart_from_file: int
    gravity_g: float
    U_max: float
    h: float
    wall_thickness: float
    periodic_buffer: float
    transport_coeff: float
    p0: float
    inv_sigma0: float
    nu: float
    config_signature: str

    def as_dict(self):
        return asdict(self)


def config_signature(DL, DH, dp, rho0, mu, U_bulk, c_f, t_end, output_interval, sort_interval) -> str:
    """SPH_Poiseuille.m:514-517."""
    g = lambda v: "%.12g" % v
    return ("DL=%s|DH=%s|dp=%s|rho0=%s|mu=%s|Ub=%s|cf=%s|t=%s|oi=%s|si=%d|wall=thick-wall-noslip-dual-dt"
            % (g(DL), g(DH), g(dp), g(rho0), g(mu), g(U_bulk), g(c_f), g(t_end), g(output_interval),
               int(sort_interval)))


def derive_params(cfg: dict, transport_coeff: float = 0.30) -> SimParams:
    """SPH_Poiseuille.m:46-91.  transport_coeff is hard-coded 0.30 in the reference (:77)."""
    DL = get_ini_numeric(cfg, "physical", "DL")
    DH = get_ini_numeric(cfg, "physical", "DH")
    dp = get_ini_numeric(cfg, "physical", "dp")
    rho0 = get_ini_numeric(cfg, "physical", "rho0")
    mu = get_ini_numeric(cfg, "physical", "mu")
    U_bulk = get_ini_numeric(cfg, "physical", "U_bulk")
    c_f = get_ini_numeric(cfg, "physical", "c_f")
    t_end = get_ini_numeric(cfg, "simulation", "end_time")
    output_interval = get_ini_numeric(cfg, "simulation", "output_interval")
    sort_interval = int(_matlab_round(get_ini_numeric(cfg, "simulation", "sort_interval")))
    restart_from_file = int(_matlab_round(get_ini_numeric(cfg, "simulation", "restart_from_file")))
    DL = _matlab_round(DL / dp) * dp  # :64-65 geometry snapped to whole particle spacings
    DH = _matlab_round(DH / dp) * dp
    if sort_interval <= 0:
        raise ConfigError("sort_interval must be a positive integer")  # :89-91
    gravity_g = 12.0 * mu * U_bulk / (rho0 * DH ** 2)
    h = 1.3 * dp
    cutoff_depth = math.ceil((2.0 * h) / dp) * dp
    wall_thickness = max(4.0 * dp, cutoff_depth)
    return SimParams(
        DL=DL, DH=DH, dp=dp, rho0=rho0, mu=mu, U_bulk=U_bulk, c_f=c_f, t_end=t_end,
        output_interval=output_interval, sort_interval=sort_interval,
        restart_from_file=restart_from_file, gravity_g=gravity_g, U_max=1.5 * U_bulk, h=h,
        wall_thickness=wall_thickness, periodic_buffer=0.0, transport_coeff=transport_coeff,
        p0=rho0 * c_f ** 2, inv_sigma0=dp ** 2, nu=mu / rho0,
        config_signature=config_signature(DL, DH, dp, rho0, mu, U_bulk, c_f, t_end, output_interval,
                                          sort_interval))


def load_config(path: str | None = None, **overrides) -> SimParams:
    """Read config.ini (SPH_CONFIG_OVERRIDE wins, SPH_Poiseuille.m:19) and apply keyword overrides
    to the raw [physical]/[simulation] keys before derivation (test hook)."""
    path = os.environ.get("SPH_CONFIG_OVERRIDE") or path
    if path is None:
        raise ConfigError("no config path given")
    cfg = parse_ini(path)
    tc = overrides.pop("transport_coeff", 0.30)
    for k, v in overrides.items():
        sec = "simulation" if k in ("end_time", "output_interval", "sort_interval", "restart_from_file") else "physical"
        cfg.setdefault(sec, {})[k] = float(v)
    return derive_params(cfg, transport_coeff=tc)


def params_from_values(DL=3.0, DH=1.0, dp=0.05, rho0=1.0, mu=0.1, U_bulk=0.666667, c_f=15.0,
                       end_time=20.0, output_interval=1.0, sort_interval=100, restart_from_file=0,
                       transport_coeff=0.30) -> SimParams:
    """Same derivation from explicit values (defaults = config.ini as shipped, config.ini:6-19)."""
    cfg = {"physical": dict(DL=float(DL), DH=float(DH), dp=float(dp), rho0=float(rho0), mu=float(mu),
                            U_bulk=float(U_bulk), c_f=float(c_f)),
           "simulation": dict(end_time=float(end_time), output_interval=float(output_interval),
                              sort_interval=float(sort_interval),
                              restart_from_file=float(restart_from_file))}
    return derive_params(cfg, transport_coeff=transport_coeff)
