#!/usr/bin/env python
"""rocprofv3 --pmc passes (ROCm 7.2 rocpd SQLite results, one counter set per pass) -> JSON.

per_kernel(dbs): per (kernel, grid) and counter the number of dispatches and avg / min / max of the per-dispatch value.  Keys are
"<kernel> @grid=<threads in x>[x<y>]": one kernel name launched at several batch sizes gets one entry per geometry.

segments(db): the dispatch sequence of ONE process cut at the marker launches of kge_debug_marker ("kge::k_marker", grid.x = 64 x
tag): per segment (= the dispatches between marker `tag` and the next marker) the sum of every counter over all kge:: kernels, and
the same per kernel.  bench.py's counter child brackets each configuration's timed steps with markers, so "HBM bytes per step"
of a configuration is the segment sum / steps -- nothing is attributed by kernel name.

Usage: rocpd_pmc.py out.json "<source note>" pass1.db [pass2.db ...]"""
import json
import sqlite3
import sys

ORDER_COLUMNS = ("dispatch_id", "start", "start_timestamp", "timestamp", "id")


def short_name(name):
    short = name[5:] if name.startswith("void ") else name
    return short.split("(")[0]


def is_ours(name):
    return name.startswith("kge::") or name.startswith("void kge::")


def per_kernel(dbs):
    kernels = {}
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        rows = cur.execute("select kernel_name, grid_size_x, grid_size_y, counter_name, count(*), avg(value), min(value), max(value), "
                           "avg(duration) from counters_collection group by kernel_name, grid_size_x, grid_size_y, counter_name").fetchall()
        for name, gx, gy, ctr, n, avg, lo, hi, dur in rows:
            if not is_ours(name) or "k_marker" in name:
                continue
            key = "%s @grid=%d%s" % (short_name(name), gx, "x%d" % gy if gy and gy > 1 else "")
            suffix = "_KB" if ctr in ("FETCH_SIZE", "WRITE_SIZE") else ""
            kernels.setdefault(key, {})[ctr] = {"dispatches": n, "avg" + suffix: avg, "min" + suffix: lo, "max" + suffix: hi,
                                                "avg_duration_us_in_this_pass": dur / 1e3}
    return kernels


def columns(db):
    cur = sqlite3.connect(db).cursor()
    cur.execute("select * from counters_collection limit 1")
    return [d[0] for d in cur.description]


def segments(db):
    """{"order_by": column, "segments": {tag: {"dispatches": n, "counters": {name: sum}, "kernels": {short name: {counter: sum, "dispatches": n,
    "duration_us": sum}}}}} -- or {"error": ...} when the view has no column to order dispatches by."""
    cols = columns(db)
    order = next((c for c in ORDER_COLUMNS if c in cols), None)
    if order is None:
        return {"error": "no ordering column in counters_collection", "columns": cols}
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, grid_size_x, counter_name, value, duration, %s from counters_collection order by %s" % (order, order))
    segs, tag = {}, None
    for name, gx, ctr, value, dur, _ in rows:
        if "k_marker" in name:
            tag = int(gx) // 64
            continue
        if tag is None or not is_ours(name):
            continue
        seg = segs.setdefault(tag, {"dispatch_rows": 0, "counters": {}, "kernels": {}})
        seg["dispatch_rows"] += 1
        seg["counters"][ctr] = seg["counters"].get(ctr, 0.0) + value
        k = seg["kernels"].setdefault(short_name(name), {"rows": 0, "duration_us": 0.0})
        k[ctr] = k.get(ctr, 0.0) + value
        k["rows"] += 1
        k["duration_us"] += (dur or 0) / 1e3
    return {"order_by": order, "segments": segs}


def main():
    out, note, dbs = sys.argv[1], sys.argv[2], sys.argv[3:]
    kernels = per_kernel(dbs)
    json.dump({"source": note, "key": "<kernel> @grid=<grid.x threads>[x<grid.y>]", "kernels": kernels}, open(out, "w"), indent=1)
    print("wrote", out, len(kernels), "kernel/grid entries")


if __name__ == "__main__":
    main()
