"""Dev tool: per-workgroup timeline of an instrumented kernel (csrc/kge_ts_debug.h).  Needs the experiment build
(make -C pykg2vec_amd/csrc ts -> tools/_libs/ts.so) selected through KGE_HIP_LIB, an MI355X, and a workload that ends with the kernel
of interest (the records are those of the LAST launch of a slot).

  KGE_HIP_LIB=tools/_libs/ts.so python tools/wg_timeline.py c1       # k_pull_step (slot 0) and k_pull_eval (slot 1) of the C1 step
  KGE_HIP_LIB=tools/_libs/ts.so python tools/wg_timeline.py transr   # k_transr_g2 of the TransR FB15k 100 / 100 B = 32 768 step

Prints: kernel span, workgroup lifetimes (mean / quantiles), start- and end-time quantiles, the maximum number of concurrently
resident workgroups per CU (from HW_REG_HW_ID / HW_REG_XCC_ID), live workgroups in flight over time.  Tags: 1 = worker, 2 = the
sampler blocks riding in k_pull_step; k_transr_g2: 1 + slabs of the run, 0 = a block that exited at once."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np, torch
import hip_util
from pykg2vec_amd.trainer import Trainer
from pykg2vec_amd import _lib as L

MAXB = 16384


def dump(unit, slot):
    lib = L.load()
    try:
        fn = getattr(lib, "kge_ts_dump_" + unit)
    except AttributeError:
        sys.exit("this library has no kge_ts_dump_%s: build tools/_libs/ts.so (make -C pykg2vec_amd/csrc ts) and set KGE_HIP_LIB" % unit)
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong]
    buf = np.zeros(MAXB * 4, dtype=np.uint64)
    rc = fn(slot, buf.ctypes.data_as(ctypes.c_void_p), MAXB)
    assert rc == 0, rc
    t = buf.reshape(-1, 4)
    return t[t[:, 0] > 0]


def analyse(name, t, live_tags=None):
    t0 = t[:, 0].astype(np.int64); t1 = t[:, 1].astype(np.int64); tag = t[:, 3].astype(np.int64)
    xcc = t[:, 2].astype(np.int64) & 15; hw = t[:, 2].astype(np.int64) >> 8
    # a record of an older launch geometry can survive at a block id the last launch did not have: keep the last cluster of entry times
    # a block id the last launch did not have keeps the record of an older launch: a handful of entry times, separated from the
    # launch's own by a gap of more than 5 us (a later residency round is a gap too, but behind at least 1 % of the records)
    order = np.argsort(t0)
    gaps = np.diff(t0[order])
    keep = np.ones(len(t0), dtype=bool)
    for gi in np.where(gaps > 500)[0]:
        if gi + 1 < 0.01 * len(t0):
            keep[order[:gi + 1]] = False
    t0, t1, tag, xcc, hw = t0[keep], t1[keep], tag[keep], xcc[keep], hw[keep]
    base = t0.min()
    us = lambda x: x / 100.0
    print("== %s: %d workgroups, span %.2f us" % (name, len(t0), us(t1.max() - base)))
    for tg in sorted(set(tag.tolist())):
        sel = tag == tg
        d = us((t1 - t0)[sel]); st = us((t0 - base)[sel]); en = us((t1 - base)[sel])
        print("  tag %d: n %d | lifetime mean %.2f p10 %.2f p50 %.2f p90 %.2f max %.2f | start p50 %.2f p90 %.2f p99 %.2f max %.2f | end p50 %.2f p90 %.2f max %.2f"
              % (tg, sel.sum(), d.mean(), np.percentile(d, 10), np.percentile(d, 50), np.percentile(d, 90), d.max(),
                 np.percentile(st, 50), np.percentile(st, 90), np.percentile(st, 99), st.max(), np.percentile(en, 50), np.percentile(en, 90), en.max()))
    live = tag > 0 if live_tags is None else np.isin(tag, live_tags)
    key = xcc * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 8) & 15)
    mx = []
    for k in set(key[live].tolist()):
        sel = (key == k) & live
        e = np.concatenate([np.stack([t0[sel], np.ones(sel.sum(), dtype=np.int64)], 1), np.stack([t1[sel], -np.ones(sel.sum(), dtype=np.int64)], 1)])
        e = e[np.lexsort((e[:, 1], e[:, 0]))]
        mx.append(int(np.cumsum(e[:, 1]).max()))
    print("  CUs seen %d; max concurrently resident workgroups (tag > 0) per CU: min %d median %d max %d" % (len(mx), min(mx), int(np.median(mx)), max(mx)))
    span = us(t1.max() - base)
    pts = [q for q in (1, 2, 4, 6, 8, 10, 15, 20, 30, 40, 60, 80) if q < span]
    print("  in flight (tag > 0):", ", ".join("t=%dus %d" % (q, int((((t0 - base) <= q * 100) & ((t1 - base) > q * 100) & live).sum())) for q in pts))


def run_c1():
    E, R, NTR, B = 14951, 1345, 483142, 32768
    rng = np.random.default_rng(1234)
    hp = dict(hidden_size=100, l1_flag=True, margin=1.0)
    train = np.stack([rng.integers(E, size=NTR), rng.integers(R, size=NTR), rng.integers(E, size=NTR)], 1)
    cfg = hip_util.make_config(E, R, dict(hp, neg_rate=1), train, train[:16], train[:16], optimizer="adam", lr=0.01, batch_size=B)
    torch.manual_seed(0)
    m = hip_util.model_from_params("transe", {}, hp, E, R, train=train)
    tr = Trainer(m, cfg); tr.build_model()
    tr.generator = tr._new_generator()
    tr.train_model_epoch(0); tr.train_model_epoch(1)
    torch.cuda.synchronize()
    analyse("k_pull_step (C1, B = 32768)", dump("pull", 0))
    analyse("k_pull_eval (C1, B = 32768)", dump("pull", 1))


def run_transr():
    E, R, B = 14951, 1345, 32768
    rng = np.random.default_rng(0)
    hp = dict(ent_hidden_size=100, rel_hidden_size=100, l1_flag=True, margin=1.0)
    train = np.stack([rng.integers(E, size=B), rng.integers(R, size=B), rng.integers(E, size=B)], 1)
    cfg = hip_util.make_config(E, R, dict(hp, neg_rate=1), train[:64], train[:4], train[:4], optimizer="sgd", lr=0.01, batch_size=B)
    torch.manual_seed(0)
    m = hip_util.model_from_params("transr", {}, hp, E, R, train=train)
    tr = Trainer(m, cfg, use_graph=False); tr.build_model()
    ph, pr, pt = [hip_util.dev(train[:, k]) for k in range(3)]
    nh = hip_util.dev(rng.integers(E, size=B)); nt = pt.clone()
    for _ in range(5):
        tr._accumulate_pairwise(ph, pr, pt, nh, pr, nt)
    torch.cuda.synchronize()
    analyse("k_transr_g2 (TransR FB15k 100/100, B = 32768; tag = 1 + slabs)", dump("transr", 0))


if __name__ == "__main__":
    {"c1": run_c1, "transr": run_transr}[sys.argv[1] if len(sys.argv) > 1 else "c1"]()
