"""Manual probe (not a test): host cost of one torch.distributed collective call on the RCCL backend (single rank)."""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.zeros(1, dtype=torch.float64, device="cuda")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(50): dist.all_reduce(x, op=dist.ReduceOp.MAX)
    torch.cuda.synchronize()
    for n in (2000,):
        t0 = time.perf_counter()
        for _ in range(n): dist.all_reduce(x, op=dist.ReduceOp.MAX)
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        ta = time.perf_counter() - t0
        print(f"all_reduce (8 bytes, 1 rank): host {th/n*1e6:.1f} us/call, until idle {ta/n*1e6:.1f} us/call")
        t0 = time.perf_counter()
        for _ in range(n):
            w = dist.all_reduce(x, op=dist.ReduceOp.MAX, async_op=True); w.wait()
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        print(f"async + wait: host {th/n*1e6:.1f} us/call")
dist.destroy_process_group()
