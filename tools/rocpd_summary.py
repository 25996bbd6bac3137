#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (ROCm 7.2 default output) as the per-kernel
--stats table: name, calls, total us, average us, percentage.  Usage: rocpd_summary.py x.db"""
import sqlite3, sys

def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    print(f"{'kernel':90s} {'calls':>7s} {'total_us':>14s} {'avg_us':>12s} {'pct':>7s}")
    for name, calls, tot, avg, pct in rows:
        print(f"{name[:90]:90s} {calls:7d} {tot:14.3f} {avg:12.3f} {pct:7.2f}")
    try:
        rows = c.execute("select counter_name, kernel_name, sum(value), count(*) from counters_collection "
                         "group by counter_name, kernel_name").fetchall()
        if rows:
            print("\ncounter, kernel, sum(value), dispatches")
            for r in rows:
                print(r)
    except Exception as e:  # no counters in this run
        pass

if __name__ == "__main__":
    main(sys.argv[1])
