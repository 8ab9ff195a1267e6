"""Per-kernel summary (calls, total ms, average us, share) of a rocprofv3 `--kernel-trace` run stored as a rocpd sqlite database.

    python tools/prof_db_summary.py gpurun_out/prof_x/x_results.db [steps] > profiles/rNN_kernel_stats_x.txt
"""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(ConvArgs\)|\(WgArgs\)", "", n)
    n = re.sub(r"^void ", "", n)
    return n[:84]


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(cur.execute(f"select s.display_name, count(*), sum(d.end - d.start), avg(d.end - d.start) from {kd} d join {ks} s on d.kernel_id = s.id "
                            "group by s.display_name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    print(f"total GPU kernel time {tot / 1e6:.1f} ms over {sum(r[1] for r in rows)} launches" + (f" ({steps} timed + warm-up steps in the trace)" if steps else ""))
    print(f"{'kernel':84s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'pct':>6s}")
    for name, calls, t, avg in rows[:28]:
        print(f"{short(name):84s} {calls:7d} {t / 1e6:10.2f} {avg / 1e3:9.1f} {100.0 * t / tot:6.2f}")


if __name__ == "__main__":
    main()
