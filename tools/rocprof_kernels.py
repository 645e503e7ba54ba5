"""Per-kernel totals of a rocprofv3 --kernel-trace run kept as SQLite (rocprofv3's default output): python tools/rocprof_kernels.py x.db [top]
Optional third argument: only dispatches whose index lies in the last FRACTION of the trace (skip warm-up), e.g. 0.5."""
import collections, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
rows = db.execute("select name, start, end from kernels order by start").fetchall()
rows = rows[int(len(rows) * (1.0 - frac)):]
agg = collections.defaultdict(lambda: [0, 0])
for n, s, e in rows:
    k = re.sub(r"\(.*", "", n).replace("sphx::", "").replace("void ", "").replace("(anonymous namespace)::", "")
    agg[k][0] += 1
    agg[k][1] += e - s
tot = sum(v[1] for v in agg.values())
print(f"{'kernel':60s} {'calls':>7s} {'total us':>12s} {'avg us':>9s} {'%':>6s}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{k[:60]:60s} {v[0]:7d} {v[1]/1e3:12.1f} {v[1]/v[0]/1e3:9.2f} {100*v[1]/tot:6.1f}")
print(f"sum of kernel durations {tot/1e3:.1f} us over a span of {(rows[-1][2]-rows[0][1])/1e3:.1f} us ({len(rows)} dispatches)")
