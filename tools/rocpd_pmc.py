"""Per-kernel PMC counter averages from a rocprofv3 rocpd database.  usage: rocpd_pmc.py <db> [kernel-substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in cur.execute("pragma table_info('counters_collection')")]
print("# columns:", cols)
name_col = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else cols[0])
rows = cur.execute(f"select {name_col}, counter_name, count(*), avg(value), min(value), max(value) from counters_collection "
                   f"group by {name_col}, counter_name").fetchall()
for r in rows:
    if flt in str(r[0]):
        print(f"{str(r[0])[:80]:<82} {r[1]:<14} n={r[2]:<4} avg={r[3]:.1f} min={r[4]:.1f} max={r[5]:.1f}")
