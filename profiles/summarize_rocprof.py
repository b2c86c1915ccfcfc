"""Dump the per-kernel summary (rocprofv3 --kernel-trace --stats) of a rocpd .db to CSV.

    python profiles/summarize_rocprof.py gpurun_out/prof_r1/bench_results.db profiles/r1_bench_kernel_stats.csv
"""
import csv
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    cur = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels")
    rows = cur.fetchall()
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
        for r in rows:
            w.writerow([r[0], r[1], "%.3f" % r[2], "%.3f" % r[3], "%.3f" % r[4]])
    print("wrote", out, len(rows), "kernels")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
