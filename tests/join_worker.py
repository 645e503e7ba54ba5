"""Worker for test_slab.py::test_ranks_agree_before_the_collective_init (CPU, gloo): slab.join_native_ring with a stand-in
C API.  Scenario `--fail-rank R --stage S`: rank R fails at stage S (available | engine | id | init); every rank must come
out with RuntimeError (nobody may be left inside the collective comm_init), or, with no failure, with an engine whose
comm_init ran exactly once.  Prints OK on every rank that behaved."""
import argparse
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class FakeError(RuntimeError):
    pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fail-rank", type=int, default=-1)
    ap.add_argument("--stage", default="none")
    args = ap.parse_args()
    import torch.distributed as dist
    rank = int(os.environ["RANK"])
    dist.init_process_group("gloo")
    slab = importlib.import_module("sph-poiseuille-flow_amd.slab")
    fail = rank == args.fail_rank
    log = []

    class Lib:
        def sphx_comm_available(self):
            return -2 if (fail and args.stage == "available") else 0

    class Capi:
        SphxError = FakeError

        def lib(self):
            return Lib()

        def check(self, rc):
            if rc != 0:
                raise FakeError(f"status {rc}")

    class Engine:
        def comm_init(self, ident):
            log.append(("init", ident))
            if fail and args.stage == "init":
                raise FakeError("ncclCommInitRank refused")

        def close(self):
            log.append(("close",))

    def make_engine():
        if fail and args.stage == "engine":
            raise FakeError("slab does not fit")
        return Engine()

    real_unique_id = slab.HipSlabEngine.unique_id

    def unique_id(capi):
        if fail and args.stage == "id":
            raise FakeError("ncclGetUniqueId failed")
        return b"x" * 128

    slab.HipSlabEngine.unique_id = staticmethod(unique_id)
    try:
        eng = slab.join_native_ring(make_engine, Capi(), rank, dist, timeout_s=30.0)
        outcome = "joined"
        assert [e[0] for e in log] == ["init"], log
        assert isinstance(eng, Engine)
    except RuntimeError as e:
        outcome = "refused"
        assert "native RCCL slab loop unavailable" in str(e), e
        # nobody entered the collective init unless every rank got that far
        if args.stage in ("available", "engine", "id"):
            assert not any(e[0] == "init" for e in log), log
    finally:
        slab.HipSlabEngine.unique_id = real_unique_id
    want = "joined" if args.fail_rank < 0 else "refused"
    ok = outcome == want
    print(f"rank {rank}: {outcome} ({'OK' if ok else 'WRONG'})", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
