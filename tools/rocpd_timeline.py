#!/usr/bin/env python
"""Dev tool: the kernel TIMELINE of a rocprofv3 (ROCm 7.2) rocpd SQLite result -- per dispatch its start relative to the previous
dispatch's end (the gap) and its duration, for a window of the steady state, plus per kernel name the mean gap in front of it.
Answers "where does a step spend the time its kernels do not account for".
Usage: rocpd_timeline.py results.db [skip_fraction=0.6] [rows=60]"""
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sys.argv[1]
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
    rows = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    cur = sqlite3.connect(db).cursor()
    ks = cur.execute("select name, start, end, grid_x from kernels order by start").fetchall()
    lo = int(len(ks) * skip)
    gaps = defaultdict(list)
    durs = defaultdict(list)
    for i in range(max(lo, 1), len(ks)):
        name = ks[i][0].split("(")[0][-70:]
        gaps[name].append((ks[i][1] - ks[i - 1][2]) / 1e3)
        durs[name].append((ks[i][2] - ks[i][1]) / 1e3)
    print("window of %d dispatches from #%d:" % (rows, lo))
    for i in range(max(lo, 1), min(lo + rows, len(ks))):
        print("  gap %7.2f us  dur %8.2f us  grid %8d  %s" % ((ks[i][1] - ks[i - 1][2]) / 1e3, (ks[i][2] - ks[i][1]) / 1e3, ks[i][3], ks[i][0].split("(")[0][-80:]))
    print("per kernel (steady state): calls, mean duration, mean gap in front")
    for name in sorted(durs, key=lambda n: -sum(durs[n])):
        g = sorted(gaps[name])
        print("  %6d  dur %8.2f  gap mean %7.2f median %7.2f  %s" % (len(durs[name]), sum(durs[name]) / len(durs[name]), sum(g) / len(g), g[len(g) // 2], name))


if __name__ == "__main__":
    main()
