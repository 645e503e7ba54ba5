#!/usr/bin/env python3
"""Static instruction mix of the loops of one kernel in `hipcc -S --cuda-device-only` output.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude --cuda-device-only -S -o /tmp/resident.s <file>.hip
    python3 tools/isa_loops.py /tmp/resident.s 'k_forces_wILi2ELi320'

For every backward branch (a loop) prints the line range and how many instructions of each class lie between the label
and the branch: f64 arithmetic, f64 transcendental (v_rcp/v_rsq/v_sqrt_f64, quarter rate), other VALU, SALU, LDS, VMEM, waits.
Nested loops are counted inside their parents too."""
import collections
import re
import sys


def classify(op):
    if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_div_")):
        return "f64_trans"
    if op.startswith("v_") and "_f64" in op:
        return "f64"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    detail = len(sys.argv) > 3 and sys.argv[3] == "ops"
    lines = open(path).read().split("\n")
    start = next(k for k, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(pat) + r"\S*:", l))
    end = next(k for k in range(start, len(lines)) if lines[k].startswith("\t.end_amdhsa_kernel") or lines[k].startswith(".Lfunc_end"))
    labels, body = {}, []
    for k in range(start, end):
        l = lines[k]
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = len(body)
            continue
        m = re.match(r"^\t([a-z_0-9]+)\s*(.*)", l)
        if m and not m.group(1).startswith("."):
            body.append((k + 1, m.group(1), m.group(2)))
    total = collections.Counter(classify(op) for _, op, _ in body)
    print("kernel lines %d-%d  instructions %d  %s" % (start + 1, end, len(body), dict(total)))
    for idx, (ln, op, args) in enumerate(body):
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = args.split()[-1] if args else ""
            if tgt in labels and labels[tgt] <= idx:
                seg = body[labels[tgt]: idx + 1]
                c = collections.Counter(classify(o) for _, o, _ in seg)
                print("loop %-10s lines %d-%d  n=%d  %s" % (tgt, seg[0][0], ln, len(seg), dict(sorted(c.items()))))
                if detail:
                    ops = collections.Counter(o for _, o, _ in seg)
                    print("    " + "  ".join("%s:%d" % kv for kv in ops.most_common()))


if __name__ == "__main__":
    main()
