#!/usr/bin/env python
"""One of bench.py's BASELINE configs (C2 / C3 / C4) on its own: prints step_us / eval numbers as one JSON line.  For same-box A/B
runs through tools/gpu_ab.sh and for rocprofv3 kernel tables of a single config.  Usage: extra_config.py C4 [--eval]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tools import bench_extra

key = sys.argv[1]
for a in sys.argv[2:]:
    if a.startswith("--optimizer="):      # A/B of the optimiser kinds on a config's shape (the bench itself keeps the preset's)
        bench_extra.EXTRA_CONFIGS[key]["optimizer"] = a.split("=", 1)[1]
out = bench_extra.run_extra_config(key, "cuda:0")
keep = ("mode", "step_us", "scored_triples_per_s") + (("eval_ms_per_pass", "eval_test_triples_per_s", "eval_setup_ms") if "--eval" in sys.argv else ())
print(json.dumps({k: out[k] for k in keep}))
