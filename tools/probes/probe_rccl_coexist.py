"""Manual probe: libsphx's dlopen'ed RCCL beside torch.distributed's own RCCL in one process (single rank)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.ones(4, dtype=torch.float64, device="cuda")
dist.all_reduce(x, op=dist.ReduceOp.MAX); torch.cuda.synchronize()
pkg = importlib.import_module("sph-poiseuille-flow_amd")
pkg.capi.check(pkg.capi.lib().sphx_comm_selftest())
dist.all_reduce(x, op=dist.ReduceOp.MAX); torch.cuda.synchronize()
pkg.capi.check(pkg.capi.lib().sphx_comm_selftest())
print("ok: torch RCCL and libsphx RCCL side by side;", [l.split()[-1] for l in open("/proc/self/maps") if "rccl" in l and "r-xp" in l])
dist.destroy_process_group()
