#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2) rocpd SQLite result: per-kernel calls / total / avg / min / max duration and
register counts -- the same content as `--stats` kernel_stats.csv.  Usage: rocpd_summary.py results.db [out.md]"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    cur = con.cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
        "max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total us | avg us | min us | max us | % | vgpr | agpr | sgpr | lds B | grid.x | wg.x |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        name = r[0] if len(r[0]) < 110 else r[0][:107] + "..."
        lines.append("| `%s` | %d | %.1f | %.2f | %.2f | %.2f | %.1f | %s | %s | %s | %s | %s | %s |" % (
            name, r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot, r[6], r[7], r[8], r[9], r[10], r[11]))
    text = "\n".join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
