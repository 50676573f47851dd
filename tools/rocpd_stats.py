"""Summarise a rocprofv3 (ROCm 7.2, rocpd SQLite output) kernel trace: per-kernel calls / total / avg / min / max.

    python tools/rocpd_stats.py <results.db> [--last-steps K --kernels-per-step N] > profiles/<name>.txt
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    where = ""
    if "--tail-ms" in sys.argv:   # only kernels that start in the last X ms of the trace (the timed steps)
        x = float(sys.argv[sys.argv.index("--tail-ms") + 1])
        tmax = cur.execute("select max(end) from kernels").fetchone()[0]
        where = f" where start >= {tmax - int(x * 1e6)}"
        span = cur.execute(f"select min(start), max(end), sum(duration) from kernels{where}").fetchone()
        print(f"# window: last {x} ms of the trace; wall span {(span[1]-span[0])/1e6:.3f} ms, kernel-busy {span[2]/1e6:.3f} ms")
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                       f"from kernels{where} group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace --stats summary of {sys.argv[1]}")
    print(f"# {'kernel':<90} {'calls':>7} {'total_us':>12} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}")
    for name, n, tot, avg, mn, mx in rows:
        short = name if len(name) <= 90 else name[:87] + "..."
        print(f"{short:<92} {n:>7} {tot/1e3:>12.1f} {avg/1e3:>10.2f} {mn/1e3:>10.2f} {mx/1e3:>10.2f} {100*tot/total:>6.2f}")


if __name__ == "__main__":
    main()
