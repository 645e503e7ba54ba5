"""sph-poiseuille-flow_amd -- MI355X (gfx950) HIP implementation of the SPH Poiseuille hot path.

The directory name carries a hyphen (it mirrors the reference repo's name), so import it with
    importlib.import_module("sph-poiseuille-flow_amd")
Sub-modules: config (config.ini + derived parameters), geometry (lattice + thick walls),
mex_surface (sph_neighbor_search_mex / sph_physics_shell_mex mirrors), capi (ctypes binding of
libsphx.so), driver (time loop), profile (u(y) binning + L2), build (hipcc driver).
Nothing here falls back to the CPU: compute entry points need csrc/libsphx.so and a HIP device.
"""
from . import config, geometry, profile  # noqa: F401  (pure host logic, importable without the .so)

__all__ = ["config", "geometry", "profile", "restart", "capi", "mex_surface", "driver", "build"]


def __getattr__(name):
    if name in ("capi", "mex_surface", "driver", "build", "slab", "restart"):
        import importlib
        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
