"""Dev tool: where a (chunk, slab) workgroup of the RESCAL slab step spends its time.  Needs the experiment build of
csrc/kge_rescal_slab.hip (-DKGE_SLAB_TS, see tools/_run_slab.sh) selected through KGE_HIP_LIB.  Prints, per kernel, the mean / p90 of
every phase (wall_clock64 of thread 0 behind a full s_waitcnt) over the live workgroups of the last launch at the C4 shape."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from tools import bench_extra
from pykg2vec_amd import _lib as L

c, cfg, model, tr, q, steps = bench_extra.build_extra_config("C4", "cuda:0", steps_cap=8)
tr.train_model_epoch(0)
torch.cuda.synchronize()
lib = L.load()
fn = lib.kge_ts_dump_slab
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_int, ctypes.c_void_p]
names = {0: ["entry->tdesc", "->gids+M loaded", "->rows loaded", "->MFMA done", "->stores drained"],
         1: ["entry->tdesc", "->gids/wsP/M/oldG loaded", "->hinge+syncs", "->rows/wsV/G0 loaded", "->U MFMA", "->atomics drained", "->G MFMA", "->G stored"]}
for which in (0, 1):
    buf = np.zeros(1024 * 12, dtype=np.uint64)
    assert fn(which, buf.ctypes.data_as(ctypes.c_void_p)) == 0
    t = buf.reshape(1024, 12).astype(np.int64)
    last = 5 if which == 0 else 8
    live = t[(t[:, 0] > 0) & (t[:, last] > t[:, 0])]
    base = live[:, 0].min()
    print("kernel %s: %d live workgroups; launch span %.2f us; entry p50 %.2f max %.2f; exit p50 %.2f max %.2f" % (
        "fwd" if which == 0 else "bwd", len(live), (live[:, last].max() - base) / 100.0, np.median(live[:, 0] - base) / 100.0,
        (live[:, 0].max() - base) / 100.0, np.median(live[:, last] - base) / 100.0, (live[:, last].max() - base) / 100.0))
    for i, nm in enumerate(names[which]):
        d = (live[:, i + 1] - live[:, i]) / 100.0
        print("   %-28s mean %.2f  p50 %.2f  p90 %.2f  max %.2f us" % (nm, d.mean(), np.median(d), np.percentile(d, 90), d.max()))
    # the slowest workgroups: block id -> (tile, slab), per-phase times
    idx = np.where((t[:, 0] > 0) & (t[:, last] > t[:, 0]))[0]
    order = idx[np.argsort(-(t[idx, last] - base))][:10]
    for b in order:
        q, x = b >> 3, b & 7
        print("   slow block %4d tile %3d slab %d: start %.2f end %.2f | phases %s" % (
            b, x + 8 * (q // 7), q % 7, (t[b, 0] - base) / 100.0, (t[b, last] - base) / 100.0,
            " ".join("%.2f" % ((t[b, i + 1] - t[b, i]) / 100.0) for i in range(last))))
    ends = (t[idx, last] - base) / 100.0
    print("   exit quantiles: p50 %.2f p75 %.2f p90 %.2f p95 %.2f p99 %.2f max %.2f" % tuple(np.percentile(ends, [50, 75, 90, 95, 99, 100])))
