#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2) rocpd SQLite result: one row per (kernel, grid bucket) -- calls / total / avg / min / max
duration and register counts, the content of `--stats` kernel_stats.csv split by launch geometry (the same kernel name is
launched at several batch sizes inside one bench run; a per-name average would mix them).  A bucket is 16 384 threads of
grid.x wide: the owner-computes step's grid follows the item count of each batch's incidence index (677 376 ... 679 680
threads over the 14 batches of the B = 32 768 epoch), which is one workload, not fourteen.
Usage: rocpd_summary.py results.db [out.md]"""
import sqlite3
import sys


def rows_of(db):
    cur = sqlite3.connect(db).cursor()
    return cur.execute(
        "select name, case when min(grid_x) = max(grid_x) then cast(min(grid_x) as text) else min(grid_x) || '-' || max(grid_x) end, "
        "grid_y, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
        "max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(workgroup_x) "
        "from kernels group by name, grid_x / 16384, grid_y order by sum(duration) desc").fetchall()


def table(rows):
    tot = sum(r[4] for r in rows) or 1
    lines = ["| kernel | grid.x (threads) | grid.y | calls | total us | avg us | min us | max us | % | vgpr | agpr | sgpr | lds B | wg.x |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        name = r[0] if len(r[0]) < 110 else r[0][:107] + "..."
        lines.append("| `%s` | %s | %s | %d | %.1f | %.2f | %.2f | %.2f | %.1f | %s | %s | %s | %s | %s |" % (
            name, r[1], r[2], r[3], r[4] / 1e3, r[5] / 1e3, r[6] / 1e3, r[7] / 1e3, 100.0 * r[4] / tot, r[8], r[9], r[10], r[11], r[12]))
    return "\n".join(lines)


def main():
    text = table(rows_of(sys.argv[1]))
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
