#!/usr/bin/env python
"""Turn a rocprofv3 rocpd SQLite result (``*_results.db`` from ``rocprofv3 --kernel-trace --stats``) into the
per-kernel summary CSV committed under profiles/ (name, calls, total_us, avg_us, percent)."""
import csv
import sqlite3
import sys


def main(db_path, out_csv):
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
        for name, calls, tot, avg, pct in rows:
            w.writerow([name, calls, "%.3f" % tot, "%.3f" % avg, "%.3f" % pct])
    print("wrote %d kernels to %s" % (len(rows), out_csv))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
