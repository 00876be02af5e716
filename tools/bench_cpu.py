"""bench.py's `cpu_baseline` leg (moved out of bench.py in round 6; no behaviour change): the reference's algorithm for the bench
workload timed on the GPU box's host cores in the same run -- the ATen restatement of the reference step (oracle/aten_step.py, bit-equal
to the live reference per tests/test_aten_restatement.py), the multi-threaded C port (oracle/kge_oracle_c.c), and the numbers the live
reference measured in the build container.  TEST / MEASUREMENT INFRASTRUCTURE: the only place outside tests/ and smoke() that touches
oracle/ (the product path never does)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from bench import DIM, E, N_TRAIN, R, build_filters  # noqa: E402

def cpu_baseline_train(train, budget_s=1.2, batch=32768):
    """C/OpenMP restatement of one reference train step (utils/trainer.py:147-157,298-299 + criterion.py:25-29 + dense
    Adam) on all host cores -- oracle/kge_oracle_c.c, a *port* held to the numpy oracle by tests/test_oracle_c.py."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import kge_oracle as ko
    import kge_oracle_c as kc
    rng = np.random.default_rng(0)
    P = ko.init_params("transe", rng, tot_entity=E, tot_relation=R, hidden_size=DIM)
    st = kc.TransEAdam(P["ent_embeddings"], P["rel_embeddings"], True, 1.0, 0.01)
    batches = []
    for k in range(N_TRAIN // batch):  # one epoch of distinct batches, like the GPU leg walks the permutation
        pos = train[k * batch:(k + 1) * batch]
        neg = pos.copy()
        flip = rng.random(batch) > 0.5
        rnd = rng.integers(E, size=batch)
        neg[:, 2] = np.where(flip, rnd, neg[:, 2])
        neg[:, 0] = np.where(flip, neg[:, 0], rnd)
        batches.append([np.ascontiguousarray(a) for a in (pos[:, 0], pos[:, 1], pos[:, 2], neg[:, 0], neg[:, 1], neg[:, 2])])
    st.train_step(*batches[0])  # warm
    # thread count: the container may expose more logical cores than it can run; probe a few counts briefly, keep the best
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else kc.threads()
    best, best_rate = avail, 0.0
    for nt in sorted({min(avail, c) for c in (8, 16, 32, avail)}):
        kc.set_threads(nt)
        t0, k = time.perf_counter(), 0
        while time.perf_counter() - t0 < 0.3:
            st.train_step(*batches[k % len(batches)])
            k += 1
        rate = k / (time.perf_counter() - t0)
        if rate > best_rate:
            best, best_rate = nt, rate
    kc.set_threads(best)
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < budget_s:
        st.train_step(*batches[n % len(batches)])
        n += 1
    dt = time.perf_counter() - t0
    return (2 * batch * n / dt, kc.threads(),
            "%d dense-Adam steps of B=%d positives + %d negatives (FB15k-shape TransE d=100 L1), C/OpenMP fp32" % (n, batch, batch))


def cpu_baseline_eval(P_np, test, csr, budget_s=1.2):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import kge_oracle_c as kc
    t_off, t_ids, h_off, h_ids = csr
    t0, n, chunk = time.perf_counter(), 0, 4 * kc.threads()
    while time.perf_counter() - t0 < budget_s and n < len(test):
        m = min(chunk, len(test) - n)
        to = t_off[n:n + m + 1] - t_off[n]
        ho = h_off[n:n + m + 1] - h_off[n]
        kc.transe_eval(P_np["ent_embeddings"], P_np["rel_embeddings"], True, test[n:n + m], to,
                       t_ids[t_off[n]:t_off[n + m]], ho, h_ids[h_off[n]:h_off[n + m]])
        n += m
    return n / (time.perf_counter() - t0), n


def reference_cpu_numbers():
    """The UNMODIFIED reference's CPU-PyTorch throughput on this workload as measured in the build container
    (profiles/r04_reference_cpu_baseline.json, tools/ref_cpu_baseline.py; the round-2 file when that is absent).  Quoted next to
    the in-run port wherever the reference tree cannot be imported (the GPU box); never used as `value`."""
    for name in ("r04_reference_cpu_baseline.json", "r02_reference_cpu_baseline.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            doc = json.load(open(path))
            return {"train_scored_triples_per_s": doc["train"]["value"], "eval_test_triples_per_s": doc["eval"]["value"],
                    "cores": doc["cores"], "host": doc["host"], "source": "profiles/" + name, "same_run": False, "same_host": False}
    return None


def cpu_baseline(H):
    """`cpu_baseline` of the JSON line, measured on THIS host in THIS run.  First choice: the reference itself (SURVEY 8(d):
    Trainer.train_step_pairwise + backward + optimizer.step, utils/trainer.py:147-157,298-299; Evaluator.test on 200 triples,
    utils/evaluator.py:309-334) -- possible wherever its tree is importable (PYKG2VEC_REFERENCE, default /root/reference; NOT on the GPU
    box).  Otherwise oracle/aten_step.py: the same ATen call sequence on the same torch CPU build, proven bit-equal to the live reference
    in the build container (tests/test_aten_restatement.py) -> `kind: "aten-restatement"`.  The C/OpenMP port of the algorithm (what a
    tuned CPU implementation reaches, ~24x the reference) rides beside either as `port`."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from pykg2vec_amd.evaluator import build_filter_csr
    n_ref = 100
    out, tried = None, None
    try:
        import ref_cpu_baseline
        if ref_cpu_baseline.available():
            hr_t, tr_h = build_filters(np.concatenate([H.train, H.valid, H.test]), H.test[:n_ref], R)
            doc = ref_cpu_baseline.measure(E, R, DIM, H.train, H.valid, H.test, hr_t, tr_h, batch=H.cfg.batch_size, n_eval=n_ref,
                                           train_budget_s=4.0, max_timed=20)
            out = {"value": doc["train"]["value"], "unit": "scored triples/s", "cores": doc["cores"], "kind": "reference",
                   "sample": doc["train"]["sample"], "what": doc["what"], "host": doc["host"], "same_run": True, "same_host": True,
                   "eval": {"value": doc["eval"]["value"], "unit": "test triples ranked/s", "sample": doc["eval"]["sample"]}}
        else:
            tried = "reference tree not present at %s" % ref_cpu_baseline.ref_shim.REFERENCE_ROOT
    except Exception as e:   # the baseline leg must never take the line down
        tried = "reference import / run failed: %s: %s" % (type(e).__name__, e)
    if out is None:
        try:
            import aten_step
            hr_t, tr_h = build_filters(np.concatenate([H.train, H.valid, H.test]), H.test[:n_ref], R)
            doc = aten_step.measure(E, R, DIM, H.train, H.test, hr_t, tr_h, batch=H.cfg.batch_size, n_eval=n_ref, margin=H.cfg.margin,
                                    lr=H.cfg.learning_rate, train_budget_s=3.0, eval_budget_s=3.0, max_timed=20)
            out = {"value": doc["train"]["value"], "unit": "scored triples/s", "cores": doc["cores"], "kind": "aten-restatement",
                   "sample": doc["train"]["sample"], "what": doc["what"], "host": doc["host"], "same_run": True, "same_host": True,
                   "kind_note": "the reference's exact ATen op sequence on this host's torch CPU build (%s)" % tried,
                   "torch_default_threads": doc.get("torch_default_threads"),
                   "value_at_torch_default_threads": doc["train"].get("value_at_torch_default_threads"),
                   "eval": {"value": doc["eval"]["value"], "unit": "test triples ranked/s", "sample": doc["eval"]["sample"]}}
        except Exception as e:
            tried = "%s; ATen restatement failed: %s: %s" % (tried, type(e).__name__, e)
    v, cores, sample = cpu_baseline_train(H.train)
    P_np = {"ent_embeddings": H.model.ent_embeddings.weight.detach().cpu().numpy(),
            "rel_embeddings": H.model.rel_embeddings.weight.detach().cpu().numpy()}
    ve, ne = cpu_baseline_eval(P_np, H.my_test, build_filter_csr(H.my_test, H.hr_t, H.tr_h))
    port = {"value": v, "unit": "scored triples/s", "cores": cores, "kind": "port", "sample": sample,
            "kind_note": "C/OpenMP restatement of the reference ALGORITHM (oracle/kge_oracle_c.c), all host cores: an upper estimate of what "
                         "a tuned CPU implementation reaches, not the reference's CPU-PyTorch path",
            "eval": {"value": ve, "unit": "test triples ranked/s", "sample": "%d test triples, two full-entity sweeps each, C/OpenMP fp32" % ne}}
    if out is None:   # neither the reference nor its ATen restatement ran: the port is all there is
        out = dict(port, same_run=True, same_host=True, kind_note=port["kind_note"] + " (%s)" % tried)
    else:
        out["port"] = port
    ref = reference_cpu_numbers()
    if ref is not None and out["kind"] != "reference":
        out["reference_in_build_container"] = ref
    return out
