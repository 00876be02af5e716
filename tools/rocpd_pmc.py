#!/usr/bin/env python
"""Merge rocprofv3 --pmc passes (ROCm 7.2 rocpd SQLite results, one counter set per pass) into one JSON: per
(kernel, grid) and counter the number of dispatches and avg / min / max of the per-dispatch value.  Keys are
"<kernel> @grid=<threads in x>[x<y>]": one kernel name launched at several batch sizes gets one entry per geometry.
Usage: rocpd_pmc.py out.json "<source note>" pass1.db [pass2.db ...]"""
import json
import sqlite3
import sys


def short_name(name):
    short = name[5:] if name.startswith("void ") else name
    return short.split("(")[0]


def main():
    out, note, dbs = sys.argv[1], sys.argv[2], sys.argv[3:]
    kernels = {}
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        rows = cur.execute("select kernel_name, grid_size_x, grid_size_y, counter_name, count(*), avg(value), min(value), max(value), "
                           "avg(duration) from counters_collection group by kernel_name, grid_size_x, grid_size_y, counter_name").fetchall()
        for name, gx, gy, ctr, n, avg, lo, hi, dur in rows:
            if not name.startswith("kge::") and not name.startswith("void kge::"):
                continue
            key = "%s @grid=%d%s" % (short_name(name), gx, "x%d" % gy if gy and gy > 1 else "")
            suffix = "_KB" if ctr in ("FETCH_SIZE", "WRITE_SIZE") else ""
            kernels.setdefault(key, {})[ctr] = {"dispatches": n, "avg" + suffix: avg, "min" + suffix: lo, "max" + suffix: hi,
                                                "avg_duration_us_in_this_pass": dur / 1e3}
    json.dump({"source": note, "key": "<kernel> @grid=<grid.x threads>[x<grid.y>]", "kernels": kernels}, open(out, "w"), indent=1)
    print("wrote", out, len(kernels), "kernel/grid entries")


if __name__ == "__main__":
    main()
