#!/usr/bin/env python
"""BUILD CONTAINER ONLY: time the unmodified reference's CPU-PyTorch path on the bench.py workload (oracle/ref_cpu_baseline.py)
and write profiles/r04_reference_cpu_baseline.json.  bench.py makes the same measurement in-run wherever the reference tree can
be imported; on the GPU box (no reference tree) it falls back to the C/OpenMP port and quotes this file next to it."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import ref_cpu_baseline  # noqa: E402

train, valid, test = bench.synthetic_split(bench.E, bench.R, (bench.N_TRAIN, bench.N_VALID, bench.N_TEST))
n_eval = 200
hr_t, tr_h = bench.build_filters(np.concatenate([train, valid, test]), test[:n_eval], bench.R)
doc = ref_cpu_baseline.measure(bench.E, bench.R, bench.DIM, train, valid, test, hr_t, tr_h, n_eval=n_eval, train_budget_s=60.0, min_timed=30)
json.dump(doc, open(os.path.join(ROOT, "profiles", "r04_reference_cpu_baseline.json"), "w"), indent=1)
print(json.dumps(doc, indent=1))
