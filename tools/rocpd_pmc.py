#!/usr/bin/env python
"""Merge rocprofv3 --pmc passes (ROCm 7.2 rocpd SQLite results, one counter set per pass) into one JSON:
per kernel and counter the number of dispatches and avg / min / max of the per-dispatch value.
Usage: rocpd_pmc.py out.json "<source note>" pass1.db [pass2.db ...]"""
import json
import sqlite3
import sys


def main():
    out, note, dbs = sys.argv[1], sys.argv[2], sys.argv[3:]
    kernels = {}
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        rows = cur.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) "
                           "from counters_collection group by kernel_name, counter_name").fetchall()
        for name, ctr, n, avg, lo, hi in rows:
            if not name.startswith("kge::") and not name.startswith("void kge::"):
                continue
            short = name[5:] if name.startswith("void ") else name
            short = short.split("(")[0]
            suffix = "_KB" if ctr in ("FETCH_SIZE", "WRITE_SIZE") else ""
            kernels.setdefault(short, {})[ctr] = {"dispatches": n, "avg" + suffix: avg, "min" + suffix: lo, "max" + suffix: hi}
    json.dump({"source": note, "kernels": kernels}, open(out, "w"), indent=1)
    print("wrote", out, len(kernels), "kernels")


if __name__ == "__main__":
    main()
