#!/usr/bin/env python
"""GPU-side half of the TRAINED rank fixtures (tests/golden_util.TRAINED): train the seeded full-size tables with the drop-in
Trainer's default (bit-reproducible) step path, twice, check both runs give byte-identical tables, and write them under gpurun_out/
for oracle/make_golden_trained.py (which runs the LIVE reference over them in the build container).  gpurun merges at most 64 MiB
per call: PART=i/n writes the i-th of n slices of the table list.

Usage (through gpurun):  python tools/make_trained_tables.py c1_transe_l1 | c2_complex | c3_rotate   [PART=1/2]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import golden_util as gu  # noqa: E402
import hip_util  # noqa: E402

name = sys.argv[1]
part, nparts = (int(x) for x in os.environ.get("PART", "1/1").split("/"))
tables, m, spec, splits, info = hip_util.train_fullsize(name)
digest = gu.tables_sha256(tables)
again = gu.tables_sha256(hip_util.train_fullsize(name)[0])
assert again == digest, "the default step path of %s is not bit-reproducible: %s vs %s" % (name, digest, again)
keys = sorted(tables)
mine = keys[(part - 1) * len(keys) // nparts: part * len(keys) // nparts]
out = os.path.join(ROOT, "gpurun_out")
os.makedirs(out, exist_ok=True)
np.savez(os.path.join(out, "trained_%s_part%dof%d.npz" % (name, part, nparts)), **{k: tables[k] for k in mine})
meta = dict(name=name, digest=digest, path=info["path"], first_loss=info["losses"][0], last_loss=info["losses"][-1],
            epochs=len(info["losses"]), tables={k: list(tables[k].shape) for k in keys}, part=[part, nparts], keys=mine)
json.dump(meta, open(os.path.join(out, "trained_%s_meta.json" % name), "w"), indent=1)
print(json.dumps(meta))
