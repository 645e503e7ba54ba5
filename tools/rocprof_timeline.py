"""Dispatch timeline of the last N kernel dispatches of a rocprofv3 SQLite trace: python tools/rocprof_timeline.py x.db [N]
name, duration, gap to the previous dispatch's end (us)."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
rows = db.execute("select name, start, end from kernels order by start").fetchall()[-n:]
prev = None
for name, s, e in rows:
    k = re.sub(r"\(.*", "", name).replace("sphx::", "").replace("void ", "").replace("(anonymous namespace)::", "")
    print(f"{k[:50]:50s} dur {(e-s)/1e3:8.2f}  gap {((s-prev)/1e3 if prev else 0):8.2f}")
    prev = e
