#!/usr/bin/env python
"""One training step out of a rocprofv3 rocpd result (``*_results.db`` of ``rocprofv3 --kernel-trace``): kernels in start order with
their queue, gap to the previous kernel on the same queue, duration; per-queue and union busy time.
usage: python tools/rocpd_timeline.py <results.db> [marker substring, default whiten_stats]"""
import collections
import sqlite3
import sys


def main(db_path, marker="whiten_stats"):
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select name, start, end, queue_id from kernels order by start"))
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    i0, i1 = marks[-3], marks[-2]
    step = rows[i0:i1]
    t0 = step[0][1]
    wall = rows[i1][1] - t0
    print("step wall %.1f us, %d kernels" % (wall / 1e3, len(step)))
    byq = collections.defaultdict(list)
    for r in step:
        byq[r[3]].append(r)
    qn = {q: i for i, q in enumerate(sorted(byq, key=lambda q: byq[q][0][1]))}
    for q, rs in byq.items():
        print("queue %d: %d kernels, busy %.1f us" % (qn[q], len(rs), sum(r[2] - r[1] for r in rs) / 1e3))
    iv = sorted((r[1], r[2]) for r in step)
    u, cs, ce = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > ce:
            u += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    u += ce - cs
    print("union busy %.1f us -> idle %.1f us; sum of durations %.1f us" % (u / 1e3, (wall - u) / 1e3, sum(e - s for s, e in iv) / 1e3))
    last = {}
    for r in step:
        q = r[3]
        gap = (r[1] - last[q]) / 1e3 if q in last else 0.0
        last[q] = r[2]
        print("%9.1f q%d gap %7.1f dur %7.1f  %s" % ((r[1] - t0) / 1e3, qn[q], gap, (r[2] - r[1]) / 1e3, r[0][:64]))


if __name__ == "__main__":
    main(*sys.argv[1:])
