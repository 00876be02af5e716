"""Freeze the LIVE reference's outputs for tests/test_aten_restatement.py -> tests/golden/ref_aten_step.npz (container only:
imports /root/reference through oracle/ref_shim.py).  Re-run: python oracle/make_golden_aten.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.join(os.path.dirname(HERE), "tests")]
import test_aten_restatement as T  # noqa: E402

out = {"torch_version": np.array(torch.__version__)}
for l1, opt_name in T.CASES:
    ref = T.run_reference(l1, opt_name)
    for k, v in ref.items():
        out["%s_%s_%s" % ("l1" if l1 else "l2", opt_name, k)] = v
path = os.path.join(os.path.dirname(HERE), "tests", "golden", "ref_aten_step.npz")
np.savez_compressed(path, **out)
print(path, os.path.getsize(path), "bytes")
