#!/usr/bin/env python
"""bench.py -- BASELINE.json metric on MI355X: scored triples/sec (train) + test triples ranked/sec,
FB15k-shape TransE d=100 (configs[1]).

A *step* is one pass of the training hot path over one batch of the HBM-resident train set: ONE kernel does the
negative corruption, score(+), score(-), hinge and the backward scatter (kge_train_pairwise_hinge_sampled)
-> [N>1: RCCL reduce-scatter of the flat dense gradient] -> fused dense Adam sweep over the rank's 1/N shard
(kge_optimizer_step) -> [N>1: RCCL all-gather of the updated tables].  Nothing is skipped or cached inside the timed
region.  After the timed training steps the same process times the filtered-rank evaluation sweep (kge_eval_ranks),
the other BASELINE configs (C2 ComplEx-WN18RR, C3 RotatE-FB15k-237, C4 RESCAL-YAGO3-10; N=1 only, `extra`) and, on
rank 0 at N=1, the CPU baseline: the UNMODIFIED reference's CPU-PyTorch path when its tree can be imported
(oracle/ref_cpu_baseline.py, `kind: "reference"`; the tree does not exist on the GPU box), else a multi-threaded C *port* of the
reference algorithm (oracle/kge_oracle_c.c, pinned to the numpy oracle and the reference's golden vectors, `kind: "port"`) --
on a bounded sample of the same workload, all host cores.

HBM traffic (`roofline.traffic`, `eval.roofline.traffic`, `extra.C*.traffic`) is OBSERVED IN THIS RUN when rocprofv3 is on the
box: rank 0 at N=1 re-runs a short version of every leg as a child process under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and
again under `--pmc WRITE_SIZE` (separate passes, MI355X_MICROARCH.md), each leg's timed steps bracketed by marker launches
(kge_debug_marker), and sums the counters of all kernels between the markers.  `--no-live-pmc` (or a missing / failing
rocprofv3) falls back to the committed passes under profiles/ and says so in `traffic_source`.

Launch:  python bench.py [--gpus N --steps K --warmup W]
         N>1 without a torch.distributed environment: bench.py re-executes itself under torch.distributed.run
         (one rank per GPU); an existing RANK/WORLD_SIZE environment (torchrun) is used as is.
Rank 0 writes the complete record to gpurun_out/bench_detail.json and prints ONE compact JSON line (< 4 KB: the driver parses the
last line of an 8 KB stdout tail) as its LAST stdout line; --full-line also prints the complete record as an earlier line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# FB15k shape: entities, relations, train / valid / test triples
E, R, N_TRAIN, N_VALID, N_TEST = 14951, 1345, 483142, 50000, 59071
DIM = 100
TRAIN_BYTES_PER_SCORED_TRIPLE = 3 * DIM * 4 * 3 + 28   # 3 628 B: fwd gather + grad read-modify-write + ids
EVAL_BYTES_PER_CANDIDATE = DIM * 4                      # 400 B: one candidate row read once
HBM_PEAK_GBS = 8000.0                                   # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3                            # MI355X_MICROARCH.md: f32-input MFMA = fp32 vector peak
# VALU issue roof of the L1 rank sweep (k_eval_sweep<L1>): per (query, candidate, k) ELEMENT the kernel issues two plain
# VALU instructions (v_subrev_f32 with the query element as SGPR operand + v_add_f32 |d|).  A wave64 VALU instruction
# occupies a SIMD-32 for 2 cycles (MI355X_MICROARCH.md), so the roof is 1024 SIMDs x 64 lanes / 4 cycles x 2.4 GHz =
# 39.3 T elements/s; tools/valu_bench.hip measures 32.9 T elements/s for exactly this instruction mix with all operands
# in registers (profiles/r02_valu_bench.txt: the chip does not hold 2.4 GHz under full VALU load).
VALU_SIMDS = 256 * 4
VALU_PEAK_CLOCK_HZ = 2.4e9
L1_SWEEP_CYCLES_PER_ELEMENT_PER_WAVE = 4.0              # 2 plain VALU issues x 2 cycles each per element per wave64
L1_SWEEP_ISSUES_PER_ELEMENT = 2.0
L1_SWEEP_MICROBENCH_TELEMS = 32.9                       # register-resident ceiling of the same mix (tools/valu_bench.hip)
REPEATS = 7                                             # timed regions per run (train) / timed passes (eval): the MEDIAN is reported
MIN_WARM_SECONDS = 0.05                                 # warm until >= 50 ms of GPU work has run, whatever --warmup says
# marker tags of the counter child (kge_debug_marker: grid.x = 64 x tag); a segment runs from its tag to the next marker
PMC_TAGS = {"C1_train": 101, "C1_eval": 102, "C1_small": 103, "C2_train": 111, "C2_eval": 112, "C3_train": 121, "C3_eval": 122,
            "C4_train": 131, "C4_eval": 132, "end": 99}
PMC_C1_STEPS, PMC_EVAL_REPS, PMC_EXTRA_STEPS, PMC_SMALL_STEPS = 28, 1, 20, 400
# third (optional) pass of the counter child: what the SQ sees -- VALU / VMEM / SALU wave-instructions, and where a wave's time goes
# (parked on s_waitcnt / issue-stalled / issuing).  Eight SQ counters fit one pass (MI355X_MICROARCH.md, counter table).
SQ_PASS = "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
# dependent-load latency under load, from the committed random-row microbenchmark (profiles/r03_gather_bench.txt, 16 296-row table,
# "chain G=32 8 hops": 21.53 us at 32 768 groups, 9.23 us at 8 192 groups -> us per hop)
HOP_US_AT_32K_GROUPS, HOP_US_AT_8K_GROUPS = 21.53 / 8, 9.23 / 8


class _KG:
    def __init__(self, cache, name="fb15k-shape-synthetic"):
        self.cache = cache
        self.dataset_name = name

    def read_cache_data(self, key):
        return self.cache[key]


def synthetic_split(E_, R_, sizes, seed=1234):
    rng = np.random.default_rng(seed)

    def draw(n):
        return np.stack([rng.integers(E_, size=n), rng.integers(R_, size=n), rng.integers(E_, size=n)], 1).astype(np.int64)

    return tuple(draw(n) for n in sizes)


def make_config(E_, R_, n_train, batch_size, device, **hp):
    cfg = types.SimpleNamespace(
        tot_entity=E_, tot_relation=R_, device=device, optimizer="adam", learning_rate=0.01, neg_rate=1, alpha=0.1,
        margin=1.0, batch_size=batch_size, epochs=1, test_num=0, test_step=1, debug=False, hits=[1, 3, 5, 10],
        patience=3, dataset_name="synthetic", sampling="uniform", tot_train_triples=n_train, seed=0,
        knowledge_graph=None)
    for k, v in hp.items():
        setattr(cfg, k, v)
    return cfg


def build_filters(all_triples, queries, R_):
    """hr_t / tr_h restricted to the keys the evaluated queries use (same sets the reference would look up)."""
    want_hr = {(int(h), int(r)) for h, r, t in queries}
    want_tr = {(int(t), int(r)) for h, r, t in queries}
    hr_t, tr_h = {k: set() for k in want_hr}, {k: set() for k in want_tr}
    key_hr = all_triples[:, 0] * R_ + all_triples[:, 1]
    key_tr = all_triples[:, 2] * R_ + all_triples[:, 1]
    q_hr = np.fromiter((h * R_ + r for h, r in want_hr), dtype=np.int64)
    q_tr = np.fromiter((t * R_ + r for t, r in want_tr), dtype=np.int64)
    for row in all_triples[np.isin(key_hr, q_hr)]:
        hr_t[(int(row[0]), int(row[1]))].add(int(row[2]))
    for row in all_triples[np.isin(key_tr, q_tr)]:
        tr_h[(int(row[2]), int(row[1]))].add(int(row[0]))
    return hr_t, tr_h


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


def pmc_traffic(kernel_prefix, batch, fetch_scale=1.0):
    """HBM bytes per launch of the dominant train kernel from the committed rocprofv3 PMC passes
    (profiles/*pmc_traffic.json: FETCH_SIZE + WRITE_SIZE, separate passes, same bench command and batch size).
    The same kernel name is launched at several geometries inside one bench run (B=32768 headline steps, the B=128
    reference-default-batch leg), so the entry is selected by GRID: tools/rocpd_pmc.py keys its rows "<kernel> @grid=<threads>"
    and the headline launches are the largest grid of that kernel.  (Files written before the per-grid keys carry one mixed
    row per kernel: its max_KB -- the big launches -- is used, never the mixed average.)
    fetch_scale: the gfx950 correction of MI355X_MICROARCH.md (HBM section) -- FETCH_SIZE reports half the bytes of wide
    (16 B per lane) coalesced reads, which is how the owner-computes kernel fetches every row; the round-1 push kernel
    reads one dword per lane (uncalibrated width: left raw).  Returns (bytes or None, source)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
    if not files or batch != 32768:
        return None, None
    for f in reversed(files):   # newest round first
        doc = json.load(open(f))
        best = None
        for name, ctr in doc["kernels"].items():
            if not (name.startswith(kernel_prefix) and "FETCH_SIZE" in ctr and "WRITE_SIZE" in ctr):
                continue
            if "@grid=" in name:
                grid = int(name.split("@grid=")[1].split("x")[0])
                cand = (grid, fetch_scale * ctr["FETCH_SIZE"]["avg_KB"] + ctr["WRITE_SIZE"]["avg_KB"], name)
            else:
                cand = (0, fetch_scale * ctr["FETCH_SIZE"]["max_KB"] + ctr["WRITE_SIZE"]["max_KB"], name + " (max rows)")
            if best is None or cand[0] > best[0]:
                best = cand
        if best is not None:
            return best[1] * 1024.0, "%s :: %s" % (os.path.basename(f), best[2])
    return None, None


def self_launch(args):
    """`python bench.py --gpus N` with no torch.distributed environment: re-execute under torch.distributed.run, one
    rank per GPU, rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


def timed_epochs(tr, steps_per_epoch, n_epochs=1):
    """One warm-up epoch (captures the hipGraph when the step is launch-bound), then n_epochs timed ones."""
    import torch
    tr.train_model_epoch(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for e in range(n_epochs):
        tr.train_model_epoch(1 + e)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (n_epochs * steps_per_epoch)


# the other BASELINE.json configs: shapes, presets, algorithmic bytes / flops per unit
EXTRA_CONFIGS = {
    "C2": dict(name="ComplEx WN18RR-shape d=200, pointwise logistic + F2 reg, Adagrad, B=5000 (+5000 negatives)",
               model="complex", E=40943, R=11, splits=(86835, 3034, 3134), hp=dict(hidden_size=200, lmbda=1e-4),
               optimizer="adagrad", batch=5000, neg=1, n_eval=3134, train_bytes=14428, eval_bytes=1600),
    "C3": dict(name="RotatE FB15k-237-shape d=1000, self-adversarial neg 16, Adam, B=1024",
               model="rotate", E=14541, R=237, splits=(272115, 17535, 20466),
               hp=dict(hidden_size=1000, margin=24.0, alpha=1.0), optimizer="adam", batch=1024, neg=16, n_eval=2048,
               train_bytes=60028, eval_bytes=8000),
    "C4": dict(name="RESCAL YAGO3-10-shape k=200, hinge, Adam, B=1024 (f32 MFMA path)",
               model="rescal", E=123182, R=37, splits=(1079040, 5000, 5000), hp=dict(hidden_size=200, margin=1.0),
               optimizer="adam", batch=1024, neg=1, n_eval=1024, train_bytes=4828, eval_bytes=800, train_flops=80400),
}


def build_extra_config(key, device, steps_cap=200):
    import torch
    import pykg2vec_amd as pa
    from pykg2vec_amd.trainer import Trainer
    c = EXTRA_CONFIGS[key]
    E_, R_ = c["E"], c["R"]
    train, valid, test = synthetic_split(E_, R_, c["splits"], seed=1234)
    q = test[:c["n_eval"]]
    hr_t, tr_h = build_filters(np.concatenate([train, valid, test]), q, R_)
    hp = dict(c["hp"])
    cfg = make_config(E_, R_, len(train), c["batch"], device, optimizer=c["optimizer"], neg_rate=c["neg"], **hp)
    cfg.knowledge_graph = _KG({"triplets_train": train, "triplets_valid": valid, "triplets_test": test, "hr_t": hr_t,
                               "tr_h": tr_h}, key)
    torch.manual_seed(0)
    model = pa.import_model(c["model"])(**cfg.__dict__)
    tr = Trainer(model, cfg)
    tr.build_model()
    tr.generator = tr._new_generator()
    steps = min(steps_cap, len(train) // c["batch"])
    cfg.tot_train_triples = steps * c["batch"]
    return c, cfg, model, tr, q, steps


def run_extra_config(key, device):
    import torch
    from pykg2vec_amd.evaluator import Evaluator
    c, cfg, model, tr, q, steps = build_extra_config(key, device)
    E_ = c["E"]
    dt = timed_epochs(tr, steps)
    rows = c["batch"] * (1 + c["neg"])
    ev = Evaluator(model, cfg)
    t0 = time.perf_counter()
    ev.rank_all(q, len(q))   # first pass: builds the per-query filter CSR (host) and uploads it
    torch.cuda.synchronize()
    first_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        ev.rank_all(q, len(q))
    torch.cuda.synchronize()
    edt = (time.perf_counter() - t0) / reps
    out = {"workload": c["name"],
           "mode": step_mode(tr),
           "step_us": dt * 1e6, "scored_triples_per_s": rows / dt,
           "train_algorithmic_GBps_whole_step": rows * c["train_bytes"] / dt / 1e9,
           "train_nominal_hbm_frac_whole_step": rows * c["train_bytes"] / dt / 1e9 / HBM_PEAK_GBS,
           "eval_test_triples_per_s": len(q) / edt, "eval_ms_per_pass": edt * 1e3, "eval_test_triples": len(q),
           "eval_setup_ms": max(0.0, first_ms - edt * 1e3),
           "eval_algorithmic_GBps": 2.0 * len(q) * E_ * c["eval_bytes"] / edt / 1e9,
           "eval_sweep": ("matrix cores (k_eval_gemm, f32 MFMA)" if c["model"] in ("complex", "rotate", "rescal") and 2 * len(q) >= 512
                          else "VALU (k_eval_sweep)"),
           "eval_TFLOPs": 2.0 * 2 * len(q) * E_ * (c["eval_bytes"] / 4) / edt / 1e12}
    if "train_flops" in c:
        out["train_TFLOPs_whole_step"] = rows * c["train_flops"] / dt / 1e12
        out["train_mfma_frac_whole_step"] = out["train_TFLOPs_whole_step"] / MFMA_F32_PEAK_TFLOPS
    del tr, ev, model
    torch.cuda.empty_cache()
    return out


def step_mode(tr):
    return ("hipGraph replay" if tr._graph is not None else
            "eager, staged gradients (no atomics, kge_optimizer_step_staged)" if getattr(tr, "_staged", None) is not None else
            "owner-computes, staged (kge_own_run: k_own_eval + k_own_step per step, no atomics, one native call per epoch)"
            if getattr(tr, "_own", None) is not None else
            "owner-computes (kge_pull_run)" if getattr(tr, "_pull", None) is not None else "eager")


# ---------------------------------------------------------------------------- HBM counters observed in this run
def pmc_child(args):
    """The process rocprofv3 wraps (one pass per counter): a short version of every leg, each timed part bracketed by
    kge_debug_marker launches so that the parent can cut the dispatch sequence into per-leg segments.  Prints nothing."""
    import torch
    from pykg2vec_amd import kernels as K
    from pykg2vec_amd.evaluator import Evaluator
    torch.cuda.set_device(0)
    device = "cuda:0"
    mark = lambda name: K.debug_marker(PMC_TAGS[name])
    H = setup_headline(args.batch, args.eval_triples, device)
    run_headline_steps(H, 2 * H.steps_per_epoch)        # warm: index build, code objects, list sets
    reset_headline(H)
    torch.cuda.synchronize()
    mark("C1_train")
    run_headline_steps(H, PMC_C1_STEPS)
    mark("end")
    H.tr.sync_model()
    ev = Evaluator(H.model, H.cfg)
    ev.rank_all(H.my_test, H.n_eval)
    mark("C1_eval")
    for _ in range(PMC_EVAL_REPS):
        ev.rank_all(H.my_test, H.n_eval)
    mark("end")
    torch.cuda.synchronize()
    for key in EXTRA_CONFIGS:
        c, cfg, model, tr, q, steps = build_extra_config(key, device, steps_cap=PMC_EXTRA_STEPS)
        tr.train_model_epoch(0)                           # warm (captures the hipGraph where the step is launch-bound)
        torch.cuda.synchronize()
        mark(key + "_train")
        tr.train_model_epoch(1)
        mark("end")
        ev = Evaluator(model, cfg)
        ev.rank_all(q, len(q))
        mark(key + "_eval")
        for _ in range(PMC_EVAL_REPS):
            ev.rank_all(q, len(q))
        mark("end")
        torch.cuda.synchronize()
        del tr, ev, model
        torch.cuda.empty_cache()


def pmc_child_units(batch, eval_triples):
    """Units (train steps / eval passes) the child runs inside each marker segment -- what a segment's counter sum is divided by."""
    units = {PMC_TAGS["C1_train"]: PMC_C1_STEPS, PMC_TAGS["C1_eval"]: PMC_EVAL_REPS}
    for key, c in EXTRA_CONFIGS.items():
        units[PMC_TAGS[key + "_train"]] = min(PMC_EXTRA_STEPS, c["splits"][0] // c["batch"])
        units[PMC_TAGS[key + "_eval"]] = PMC_EVAL_REPS
    return units


def live_pmc(args, timeout_s=150):
    """Run the counter child under rocprofv3 once per counter (FETCH_SIZE and WRITE_SIZE do not fit one pass) and return
    {leg: {"fetch_raw_bytes", "write_bytes", "bytes" (2 x fetch + write), "kernels": {...}}} per unit (step / pass), or
    (None, reason).  Everything is best effort: a missing rocprofv3, a timeout or an unreadable result only costs the live figure."""
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import rocpd_pmc
    units = pmc_child_units(args.batch, args.eval_triples)
    name_of = {v: k for k, v in PMC_TAGS.items()}
    legs, meta = {}, {}
    tmp = tempfile.mkdtemp(prefix="kge_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        # third pass: VALU issue counters (the owner kernel of the train leg is VALU-bound, profiles/r04_experiments.md section 7)
        for ctr in ("FETCH_SIZE", "WRITE_SIZE", SQ_PASS):
            out_dir = os.path.join(tmp, ctr.split()[0])
            cmd = [exe, "--pmc"] + ctr.split() + ["--kernel-trace", "-d", out_dir, "-o", "b", "--", sys.executable, os.path.abspath(__file__),
                   "--pmc-child", "--batch", str(args.batch), "--eval-triples", str(args.eval_triples)]
            t0 = time.perf_counter()
            optional = ctr.startswith("SQ_")     # the issue-counter pass is extra evidence: its failure must not cost the traffic figure
            try:
                res = subprocess.run(cmd, env=env, cwd="/tmp", timeout=timeout_s, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
            except subprocess.TimeoutExpired:
                if optional:
                    meta["sq_pass_error"] = "exceeded %d s" % timeout_s
                    continue
                return None, "rocprofv3 --pmc %s pass exceeded %d s" % (ctr, timeout_s)
            meta[ctr.split()[0] + "_pass_s"] = time.perf_counter() - t0
            dbs = glob.glob(os.path.join(out_dir, "**", "*.db"), recursive=True)
            if res.returncode != 0 or not dbs:
                msg = "rocprofv3 --pmc %s pass failed (rc %d, %d result files): %s" % (
                    ctr, res.returncode, len(dbs), res.stdout.decode(errors="replace")[-300:])
                if optional:
                    meta["sq_pass_error"] = msg
                    continue
                return None, msg
            seg = rocpd_pmc.segments(dbs[0])
            if "error" in seg:
                if optional:
                    meta["sq_pass_error"] = seg["error"]
                    continue
                return None, "%s (columns: %s)" % (seg["error"], seg.get("columns"))
            meta["order_by"] = seg["order_by"]
            for tag, rec in seg["segments"].items():
                if tag not in units:
                    continue
                leg = legs.setdefault(name_of[tag], {"units": units[tag], "kernels": {}})
                if ctr.startswith("SQ_"):    # several counters in one pass: one row per (dispatch, counter)
                    ncs = len(ctr.split())
                    for kname, k in rec["kernels"].items():
                        kk = leg["kernels"].setdefault(kname, {})
                        for c in ctr.split():
                            kk[c + "_per_unit"] = k.get(c, 0.0) / units[tag]
                        kk["us_per_unit_in_sq_pass"] = k["duration_us"] / ncs / units[tag]
                    continue
                total_kb = rec["counters"].get(ctr, 0.0)
                leg["fetch_raw_bytes" if ctr == "FETCH_SIZE" else "write_bytes"] = total_kb * 1024.0 / units[tag]
                for kname, k in rec["kernels"].items():
                    kk = leg["kernels"].setdefault(kname, {})
                    kk[ctr + "_KB_per_unit"] = k.get(ctr, 0.0) / units[tag]
                    kk["launches_per_unit"] = k["rows"] / units[tag]
                    kk["us_per_unit_in_counter_pass"] = k["duration_us"] / units[tag]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    for leg in legs.values():
        if "fetch_raw_bytes" in leg and "write_bytes" in leg:
            # gfx950: FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 B (MI355X_MICROARCH.md, HBM section):
            # doubled, as the guide prescribes for 16-byte-per-lane reads, which is how these kernels read rows and streams
            leg["bytes"] = 2.0 * leg["fetch_raw_bytes"] + leg["write_bytes"]
    return legs, meta


def valu_record(kernels):
    """VALU issue load of the train leg's kernels from the third live counter pass, in the ONE convention both legs of the line use
    (MI355X_MICROARCH.md, "Wave scheduling" + the per-instruction table): a wave64 VALU instruction occupies its SIMD-32 for 2 cycles, the
    roof is 1 024 SIMDs x 2.4 GHz / 2 = 1.2288 T wave-instructions/s.  `issue_frac` = SQ_INSTS_VALU / duration / that roof.
    (SQ_ACTIVE_INST_VALU, in quad-cycles, is kept raw: rounds 3-4 divided it by a busy-cycle clock and read 0.73 "VALU-bound" off it;
    by the guide's own issue rate the same launch sits near 0.3 -- see DESIGN.md section 4 for what does bound it.)"""
    if not kernels:
        return None
    out = {}
    roof = VALU_SIMDS * VALU_PEAK_CLOCK_HZ / 2.0
    for name, k in kernels.items():
        if "SQ_INSTS_VALU_per_unit" not in k or not k.get("us_per_unit_in_sq_pass"):
            continue
        dur = k["us_per_unit_in_sq_pass"] * 1e-6
        rec = {"valu_wave_instructions_per_step": k.get("SQ_INSTS_VALU_per_unit"), "waves_per_step": k.get("SQ_WAVES_per_unit"),
               "active_quad_cycles_per_step": k.get("SQ_ACTIVE_INST_VALU_per_unit"), "us_per_step_in_this_pass": dur * 1e6,
               "issue_frac": k["SQ_INSTS_VALU_per_unit"] / dur / roof}
        for c in ("SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU", "SQ_WAIT_INST_ANY", "SQ_INST_CYCLES_VMEM", "SQ_WAVE_CYCLES",
                  "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"):
            if c + "_per_unit" in k:
                rec[c] = k[c + "_per_unit"]
        if rec.get("SQ_WAVE_CYCLES"):
            wc = rec["SQ_WAVE_CYCLES"]
            rec["wave_time_split"] = {n: rec[c] / wc for n, c in (("parked_waitcnt", "SQ_WAIT_ANY"), ("issue_stall", "SQ_WAIT_INST_ANY"),
                                                                   ("issuing", "SQ_ACTIVE_INST_ANY")) if rec.get(c) is not None}
        out[name] = rec
    if not out:
        return None
    out["convention"] = "issue_frac = wave64 VALU instructions / s over 1024 SIMDs x 2.4 GHz / 2 cycles per instruction"
    return out


def setup_headline(batch, eval_triples, device, world=1, rank=0):
    """The headline workload (configs[1]): FB15k-shape TransE d=100 L1, synthetic ids, model + Trainer + generator on `device`."""
    import torch
    import pykg2vec_amd.pairwise as pw
    from pykg2vec_amd.trainer import Trainer
    H = types.SimpleNamespace()
    H.train, H.valid, H.test = synthetic_split(E, R, (N_TRAIN, N_VALID, N_TEST))
    H.n_eval = min(eval_triples, N_TEST // world)
    H.my_test = H.test[rank * H.n_eval:(rank + 1) * H.n_eval]  # queries sharded over ranks, tables replicated: no collective
    H.hr_t, H.tr_h = build_filters(np.concatenate([H.train, H.valid, H.test]), H.my_test, R)
    H.cfg = make_config(E, R, N_TRAIN, batch * world, device, hidden_size=DIM, l1_flag=True)
    # the cache carries the three splits as arrays (the Evaluator builds its filter lists from them on the device) and, for the
    # CPU baselines, the reference-format dicts of sets restricted to the evaluated queries
    H.cfg.knowledge_graph = _KG({"triplets_train": H.train, "triplets_valid": H.valid, "triplets_test": H.test,
                                 "hr_t": H.hr_t, "tr_h": H.tr_h})
    torch.manual_seed(0)
    H.model = pw.TransE(**H.cfg.__dict__)
    H.tr = Trainer(H.model, H.cfg)
    H.tr.build_model()
    H.gen = H.tr._new_generator()
    H.tr.generator = H.gen
    H.steps_per_epoch = N_TRAIN // H.cfg.batch_size
    H.init_param = H.tr.flat.param.clone()
    H.pull = H.tr._pull_ok()   # single GPU, big batch: the atomic-free owner-computes step (csrc/kge_pull.hip)
    H.two_phase = bool(H.pull and H.tr._pull_two_phase())   # ... in two launches: every pair evaluated once, owners sum the records
    H.pull_dp = H.tr._pull_dp_ok()   # N > 1: the same kernel writes the rank's dense gradient (no atomics), then the sharded step
    return H


def reset_headline(H):
    """Back to the freshly initialised tables and optimiser state.  The hinge kernel skips the backward of pairs whose
    margin is already satisfied, so a step gets cheaper as training progresses: every measurement starts
    from the same (initial, all-margins-violated) state, whatever the warm-up length."""
    tr = H.tr
    ps = getattr(tr, "_pull", None)
    if ps is not None:
        ps.cur = 0
    tr.flat.param.copy_(H.init_param)
    tr.flat.grad.zero_()
    for st in (tr.flat.state1, tr.flat.state2):
        if st is not None:
            st.zero_()
    tr.flat.step = 0
    if ps is not None:
        ps.sync_in()   # row norms of the restored tables


def run_headline_steps(H, n, events=None):
    """n training steps through the product's step path.  Owner-computes path: the steps of an epoch are enqueued by one
    native call (kge_pull_run), so there are no per-step events.  Push path: ONE launch does corruption + score(+) + score(-) +
    hinge + backward scatter, then the optimiser."""
    tr, gen = H.tr, H.gen
    if not (H.pull or H.pull_dp):
        for k in range(n):
            if gen._pending <= 0:
                gen.start_one_epoch(H.steps_per_epoch)
            if events is not None:
                events[k][0].record()
            tr._accumulate_next_batch()
            if events is not None:
                events[k][1].record()
            tr._reduce_and_step()
        return
    while n > 0:
        if gen._pending <= 0:
            gen.start_one_epoch(H.steps_per_epoch)
        k = min(n, gen._pending)
        tr.step_next_batches(k)
        n -= k


def rccl_setup_summary():
    """What RCCL reported when it built this process's communicator (the NCCL_DEBUG=INFO / INIT file main() asked for): channel
    count, transports, connected topologies, and -- only with KGE_BENCH_RCCL_TUNING=1 -- the algorithm / protocol it chose per
    collective size.  Best effort: None when there is no log (gloo, user-set NCCL_DEBUG) or nothing recognisable in it."""
    import re
    try:
        path = "/tmp/kge_rccl_%d.log" % os.getpid()
        if not os.path.exists(path):
            return None
        txt = open(path, errors="replace").read()
        chans = [int(m) for m in re.findall(r"Channel (\d+)/\d+ :", txt)]
        nchan = re.findall(r"(\d+) coll channels", txt)
        out = {"version": (re.findall(r"(?:RCCL|NCCL) version ([^\s]+)", txt) or [None])[0],
               "channels": (int(nchan[-1]) if nchan else (max(chans) + 1 if chans else None)),
               "transports": sorted(set(re.findall(r"via (P2P/[A-Za-z/]+|SHM[A-Za-z/]*|NET/[A-Za-z]+|direct)", txt))) or None,
               "connected": sorted(set(m.lower() for m in re.findall(r"Connected all (rings|trees)", txt))) or None}
        choice = re.findall(r"(\w+): (\d+) Bytes -> Algo (\d+) proto (\d+)", txt)
        if choice:
            algo = {"0": "tree", "1": "ring", "2": "collnet_direct", "3": "collnet_chain", "4": "nvls", "5": "nvls_tree"}
            proto = {"0": "LL", "1": "LL128", "2": "simple"}
            seen = {}
            for coll, nbytes, a, pr in choice:
                seen["%s %s B" % (coll, nbytes)] = "%s/%s" % (algo.get(a, a), proto.get(pr, pr))
            out["chosen"] = dict(list(seen.items())[:8])
        return out if any(v is not None for v in out.values()) else None
    except Exception as e:   # never let log parsing take the line down
        return {"error": "%s: %s" % (type(e).__name__, e)}


def predicted_step_us(world, allreduce):
    """DESIGN.md section 5b/5d's arithmetic for the C1 step at N ranks (nothing measured: 61 GB/s per xGMI link and direction, one link per
    peer, collective latency 10 / 15 / 25 us at N = 2 / 4 / 8), so that the first real `phases_us` is judged against a stated model."""
    S = (E + R) * DIM * 4.0
    alpha = {2: 10.0, 4: 15.0, 8: 25.0}.get(world, 25.0)
    wire = (world - 1.0) / world * S / 61e3 if world == 2 else S / world / 61e3      # us; N = 2 has ONE link
    compute, norms = 33.0, 5.0
    if allreduce:
        return {"compute": compute, "all_reduce": alpha + 2 * wire, "optimiser": 8.0, "row_norms": norms,
                "step": compute + alpha + 2 * wire + 8.0 + norms}
    return {"compute": compute, "reduce_scatter": alpha + wire, "optimiser": 5.0, "all_gather": alpha + wire, "row_norms": norms,
            "step": compute + 2 * (alpha + wire) + 5.0 + norms}


# ---------------------------------------------------------------------------- the line the driver reads
COMPACT_LIMIT = 4096     # bytes; the driver keeps an 8 KB stdout tail and parses its last line


def _r(x, sig=5):
    """Round a number to `sig` significant digits (keeps the compact line short); passes None / non-numbers through."""
    if isinstance(x, bool) or not isinstance(x, (int, float)):
        return x
    if x == 0 or x != x or x in (float("inf"), float("-inf")):
        return x
    return float("%.*g" % (sig, x))


def _short(sv, n):
    return sv if sv is None or len(sv) <= n else sv[:n - 1] + "~"


def compact_line(out, detail_path=None):
    """The LAST stdout line: every field of the bench contract + roofline + cpu_baseline + one record per other config, in
    < COMPACT_LIMIT bytes.  `out` is the full record (written to gpurun_out/bench_detail.json); nothing is recomputed here."""
    ro, ev, cb = out.get("roofline") or {}, out.get("eval") or {}, out.get("cpu_baseline")
    cfgd = out.get("config") or {}
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    line["value"], line["ms_per_step"] = _r(line["value"], 7), _r(line["ms_per_step"], 6)
    line["repeats"] = out.get("repeats")
    line["ms_per_step_min"], line["ms_per_step_max"] = _r(out.get("ms_per_step_min"), 5), _r(out.get("ms_per_step_max"), 5)
    line["timed_region_s"] = None if out.get("timed_region_s") is None else round(out["timed_region_s"], 7)
    line["config"] = {"workload": _short(cfgd.get("workload"), 150), "batch_per_gpu": cfgd.get("batch_per_gpu"),
                      "global_batch": cfgd.get("global_batch"), "parallelism": cfgd.get("parallelism"),
                      "step_path": _short(cfgd.get("step_path_short") or cfgd.get("step_path"), 110)}
    line["roofline"] = {"kernel": _short(ro.get("kernel_short") or ro.get("kernel"), 90), "bound": ro.get("bound"),
                        "achieved": _r(ro.get("achieved")), "peak": ro.get("peak"), "unit": ro.get("unit"), "frac": _r(ro.get("frac"), 4),
                        "traffic": _r(ro.get("traffic"), 6), "traffic_src": ro.get("traffic_src_short"),
                        "frac_roof": _short(ro.get("frac_roof"), 60), "algorithmic_frac": _r(ro.get("algorithmic_frac"), 4),
                        "avg_launch_ms": _r(ro.get("avg_launch_ms")), "nominal_frac": _r(ro.get("nominal_frac"), 4),
                        "hbm_frac": _r(ro.get("hbm_frac"), 4), "valu_frac": _r(ro.get("valu_frac"), 4),
                        "bound_note": _short(ro.get("bound_note"), 160)}
    er = ev.get("roofline") or {}
    line["eval"] = {"value": _r(ev.get("value"), 7), "unit": ev.get("unit"), "ms_per_pass": _r(ev.get("ms_per_pass")),
                    "ms_per_pass_min": _r(ev.get("ms_per_pass_min"), 4), "ms_per_pass_max": _r(ev.get("ms_per_pass_max"), 4),
                    "test_triples": ev.get("test_triples_per_gpu"), "setup_ms": _r(ev.get("setup_ms"), 4),
                    "roofline": {"kernel": _short(er.get("kernel_short") or er.get("kernel"), 60), "bound": er.get("bound"),
                                 "achieved": _r(er.get("achieved")), "peak": _r(er.get("peak")), "unit": _short(er.get("unit"), 40),
                                 "frac": _r(er.get("frac"), 4), "traffic": _r(er.get("traffic"), 6),
                                 "hbm_frac": _r(er.get("traffic_hbm_frac"), 4)}}
    if cb:
        c = {"value": _r(cb.get("value"), 6), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
             "same_run": cb.get("same_run", True), "same_host": cb.get("same_host", True), "sample": _short(cb.get("sample"), 120)}
        if cb.get("eval"):
            c["eval"] = {"value": _r(cb["eval"].get("value"), 5), "unit": cb["eval"].get("unit")}
        if cb.get("value_at_torch_default_threads"):
            c["at_torch_default_threads"] = {"value": _r(cb["value_at_torch_default_threads"], 5), "threads": cb.get("torch_default_threads")}
        if cb.get("port"):
            c["port"] = {"value": _r(cb["port"].get("value"), 5), "cores": cb["port"].get("cores"),
                         "eval": _r((cb["port"].get("eval") or {}).get("value"), 5)}
        if cb.get("reference_in_build_container"):
            rb = cb["reference_in_build_container"]
            c["ref_build_container"] = {"value": _r(rb.get("train_scored_triples_per_s"), 5), "eval": _r(rb.get("eval_test_triples_per_s"), 4),
                                        "cores": rb.get("cores"), "same_run": False}
        c["gpu_over_cpu"] = _r(out["value"] / cb["value"], 4) if cb.get("value") else None
        if cb.get("eval") and cb["eval"].get("value") and ev.get("value"):
            c["gpu_over_cpu_eval"] = _r(ev["value"] / cb["eval"]["value"], 4)
        line["cpu_baseline"] = c
    if out.get("extra"):
        ex = {}
        for key, rec in out["extra"].items():
            if "error" in rec:
                ex[key] = {"error": _short(rec["error"], 80)}
                continue
            e = {"train": _r(rec.get("scored_triples_per_s")), "step_us": _r(rec.get("step_us"), 4),
                 "eval": _r(rec.get("eval_test_triples_per_s")), "eval_ms": _r(rec.get("eval_ms_per_pass"), 4)}
            if rec.get("train_traffic"):
                e["train_hbm_frac"] = _r(rec["train_traffic"].get("hbm_frac"), 3)
            if rec.get("eval_TFLOPs") is not None and "matrix" in (rec.get("eval_sweep") or ""):
                e["eval_mfma_frac"] = _r(rec["eval_TFLOPs"] / MFMA_F32_PEAK_TFLOPS, 3)
            for k in ("dominant_kernel", "dominant_kernel_us", "dominant_kernel_mfma_frac"):
                if rec.get(k) is not None:
                    e[k] = _r(rec[k], 4) if not isinstance(rec[k], str) else _short(rec[k], 40)
            ex[key] = e
        line["extra"] = ex
        line["extra_units"] = "train: scored triples/s; eval: test triples ranked/s"
    if out.get("train_reference_default_batch"):
        sm = out["train_reference_default_batch"]
        line["default_batch_128"] = {"value": _r(sm.get("value")), "ms_per_step": _r(sm.get("ms_per_step"), 4)}
    if out.get("setup_ms") is not None:
        line["setup_ms"] = _r(out["setup_ms"], 4)
    for k in ("phases_us", "predicted_step_us", "replicas_identical"):
        if out.get(k) is not None:
            line[k] = {a: _r(b, 4) for a, b in out[k].items()} if isinstance(out[k], dict) else out[k]
    if out.get("collectives"):
        co = out["collectives"]
        line["collectives"] = {"backend": co.get("backend"), "world_size": co.get("world_size"), "per_step": _short(co.get("per_step"), 120),
                               "captured": co.get("captured"), "NCCL_ALGO": co.get("NCCL_ALGO"), "NCCL_PROTO": co.get("NCCL_PROTO"),
                               "rccl": co.get("rccl")}
    line["detail"] = detail_path
    txt = json.dumps(line, separators=(",", ":"))
    # never exceed the limit: shed optional parts, most expendable first
    for drop in (("roofline", "bound_note"), ("cpu_baseline", "sample"), ("default_batch_128",), ("extra_units",), ("phases_us",),
                 ("cpu_baseline", "ref_build_container"), ("extra",), ("config", "step_path"), ("roofline", "kernel")):
        if len(txt) < COMPACT_LIMIT:
            break
        tgt = line
        for k in drop[:-1]:
            tgt = tgt.get(k) or {}
        tgt.pop(drop[-1], None)
        txt = json.dumps(line, separators=(",", ":"))
    return txt


def write_detail(out):
    """Full record (per-kernel counter dumps, notes, models) -> gpurun_out/bench_detail.json; returns the path or None."""
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, "bench_detail.json" if out.get("n_gpus", 1) == 1 else "bench_detail_n%d.json" % out["n_gpus"])
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
        return os.path.relpath(path, ROOT)
    except OSError:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32768, help="positives per GPU per step (weak scaling)")
    ap.add_argument("--eval-triples", type=int, default=N_TEST, help="test triples ranked per GPU in the eval leg (default: the whole FB15k-shape test split, 59 071 = a full_test(); capped at N_TEST // world)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the C2 / C3 / C4 `extra` records (N=1 only)")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not run the rocprofv3 counter passes (use the committed ones)")
    ap.add_argument("--full-line", action="store_true", help="also print the complete record (tens of KB) as an earlier stdout line; it is always written to gpurun_out/bench_detail.json")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.pmc_child:
        return pmc_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    # one rank per GPU; KGE_BENCH_SHARE_GPU=1 (tests only) lets the ranks of a 1-GPU box share device 0 over gloo, which
    # exercises the whole N>1 path (sharded sampler stream, gradient exchange, sharded optimiser, replica consistency)
    share = os.environ.get("KGE_BENCH_SHARE_GPU") == "1"
    dev_index = local_rank % torch.cuda.device_count() if share else local_rank
    torch.cuda.set_device(dev_index)
    device = "cuda:%d" % dev_index
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            # what RCCL set up (channels, transports, rings / trees) is logged once at communicator creation: INIT-only debug
            # output into a per-process file costs nothing inside the timed region (KGE_BENCH_RCCL_TUNING=1 adds the per-call
            # algorithm / protocol choice -- a line per collective, which DOES perturb the timing: diagnosis only)
            if "NCCL_DEBUG" not in os.environ:
                os.environ["NCCL_DEBUG"] = "INFO"
                os.environ["NCCL_DEBUG_SUBSYS"] = "INIT,TUNING" if os.environ.get("KGE_BENCH_RCCL_TUNING") == "1" else "INIT"
                os.environ["NCCL_DEBUG_FILE"] = "/tmp/kge_rccl_%p.log"
            dist.init_process_group("nccl", device_id=torch.device(device))

    import pykg2vec_amd.pairwise as pw
    from pykg2vec_amd import kernels as K
    from pykg2vec_amd.evaluator import Evaluator
    from pykg2vec_amd.trainer import Trainer

    H = setup_headline(args.batch, args.eval_triples, device, world, rank)
    train, valid, test, my_test, n_eval, hr_t, tr_h = H.train, H.valid, H.test, H.my_test, H.n_eval, H.hr_t, H.tr_h
    cfg, model, tr, gen, steps_per_epoch = H.cfg, H.model, H.tr, H.gen, H.steps_per_epoch
    pull, two_phase, pull_dp = H.pull, H.two_phase, H.pull_dp

    # ---- per-run set-up of the owner-computes path: the incidence index of every batch of the epoch order, built on the device
    # (csrc/kge_index.hip).  Timed twice: cold (first call: includes loading the code objects) and warm (a rebuild), host wall
    # clock around the call incl. its one device->host read; plus HIP events around the warm build.
    setup = None
    if pull or pull_dp:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gen.pull_index()
        torch.cuda.synchronize()
        cold_ms = (time.perf_counter() - t0) * 1e3
        gen._pull_index = None
        es0, es1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        es0.record()
        idx0 = gen.pull_index()
        es1.record()
        torch.cuda.synchronize()
        warm_ms = (time.perf_counter() - t0) * 1e3
        setup = {"what": "incidence index of %d batches of %d pairs (kge_pull_index_build: key build, three batched bitonic sorts, "
                         "row list, placement)" % (idx0.n_batches, idx0.batch_size), "built_on": idx0.built_on,
                 "host_wall_ms_cold": cold_ms, "host_wall_ms_warm": warm_ms, "device_ms_warm": es0.elapsed_time(es1),
                 "gpu_step_equivalents": None}

    reset_model = lambda: reset_headline(H)
    run_steps = lambda n, events=None: run_headline_steps(H, n, events)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warm-up: the W steps asked for, then more until >= MIN_WARM_SECONDS of GPU work has run (clocks, caches,
    # code objects, RCCL channels), so that a short timed region measures the steady state
    run_steps(args.warmup)
    torch.cuda.synchronize()
    warm_extra, t_warm = 0, time.perf_counter()
    while True:
        flag = torch.tensor([float(time.perf_counter() - t_warm < MIN_WARM_SECONDS)], device=device)
        if world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)   # every rank runs the same number of collective steps
        if flag.item() == 0.0:
            break
        run_steps(16)
        warm_extra += 16
        torch.cuda.synchronize()

    # ---- the timed region: EXACTLY --steps steps between barrier + synchronize on both sides.  A region of 20 steps is 0.6 ms: one
    # sample of it moves by 5-9 % with the box's clock state, so the region is run REPEATS times (tables reset in between, each one
    # bracketed and MAX-reduced over ranks on its own) and the MEDIAN region is what the line reports; min / max go next to it
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    regions = []
    for _rep in range(REPEATS):
        reset_model()
        barrier()
        ev_t0, ev_t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev_t0.record()
        run_steps(args.steps, events)
        ev_t1.record()
        barrier()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        regions.append((float(t.item()), ev_t0.elapsed_time(ev_t1) / args.steps))
    by_wall = sorted(regions)
    dt, region_ms_per_step = by_wall[len(by_wall) // 2]   # the median region: its wall clock and its HIP-event time per step
    region_spread = {"repeats": REPEATS, "ms_per_step_min": by_wall[0][0] / args.steps * 1e3, "ms_per_step_max": by_wall[-1][0] / args.steps * 1e3,
                     "ms_per_step_all": [r[0] / args.steps * 1e3 for r in regions]}
    # ---- N > 1: where a step's time goes.  A separate, untimed pass of 16 steps with an event at every phase boundary of the
    # data-parallel step (Trainer._mark): compute (the owner-computes kernel writing this rank's dense gradient rows, next batch's
    # sampler riding along) / reduce-scatter / optimiser on the rank's shard / all-gather of the updated tables / row norms of the
    # gathered tables.  Mean per phase, MAX over ranks.  (Outside the timed region: the events serialise the async all-gather.)
    phases_us = None
    if world > 1:
        reset_model()
        tr.phase_marks = []
        run_steps(16)
        torch.cuda.synchronize()
        marks, tr.phase_marks = tr.phase_marks, None
        acc = {}
        for (n0, e0), (n1, e1) in zip(marks, marks[1:]):
            if n1 != "begin":
                acc.setdefault(n1, []).append(e0.elapsed_time(e1) * 1e3)
        names = ["compute", "reduce_scatter", "optimiser", "all_gather", "row_norms"]
        vec = torch.tensor([float(np.mean(acc[k])) if k in acc else 0.0 for k in names], dtype=torch.float64, device=device)
        dist.all_reduce(vec, op=dist.ReduceOp.MAX)
        phases_us = {k: float(v) for k, v in zip(names, vec.tolist())}
        phases_us["steps"] = 16
    per_rank_batch = args.batch
    scored_per_step = 2 * per_rank_batch * world
    value = scored_per_step * args.steps / dt
    # HIP events around each launch of the timed region: they bracket [dispatch gap after the previous kernel + the
    # kernel], i.e. an upper bound of the kernel's duration ...
    event_ms = None if (pull or pull_dp) else float(np.mean([a.elapsed_time(b) for a, b in events]))
    # ... the kernel's own duration (what rocprofv3 --kernel-trace reports, profiles/) is measured right after the
    # timed region by a burst of back-to-back launches of the SAME kernel on the same stream between two events: no
    # host gap, no optimiser in between (gradients just keep accumulating; they are cleared afterwards)
    burst = 32
    reset_model()
    eb0, eb1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if pull_dp:
        tr.generator.start_one_epoch(steps_per_epoch)
        tr.step_next_batches(1)   # (allocates the gradient-mode state if the timed loop did not)
        ps, idx = tr._pull, gen.pull_index()
        pairs_b, inc_b, items_b, multi_b = idx.batch(0)
        for ls in ps.lists:
            ls.clear()
        ps.ready, ps.cur_list = None, 0
        K.pull_sample(pairs_b, idx.inv(0), E, gen.bern, gen.slots, gen.seed, 0, ps.lists[0])
        torch.cuda.synchronize()
        eb0.record()
        for _ in range(burst):
            K.pull_step(tr._desc, ps.tables[1], ps.hats[0], None, ps.norms[0], None, None, None, pairs_b, ps.lists[0], items_b, inc_b,
                        ps.partials, multi_b, cfg.margin, "gradient", 0.0, 1, tr.loss_buf, reset_lists=False, run_finish=False,
                        dense_skip=idx.skip(0))
        eb1.record()
        torch.cuda.synchronize()
        ps.lists[0].clear()
    elif pull:
        ps, idx = tr._pull_state()
        pairs_b, inc_b, items_b, multi_b = idx.batch(0)
        for ls in ps.lists:   # a sampler riding in the last timed step may have filled a set for a batch that never ran
            ls.clear()
        ps.ready, ps.cur_list = None, 0
        K.pull_sample(pairs_b, idx.inv(0), E, gen.bern, gen.slots, gen.seed, 0, ps.lists[0])
        desc_b = K.make_desc("transe", ps.tables[0], None, tot_entity=E, tot_relation=R, **model.desc_kwargs())
        torch.cuda.synchronize()
        eb0.record()
        for _ in range(burst):   # same inputs every time (lists kept, no buffer swap): the kernel's own duration
            K.pull_step(desc_b, ps.tables[1], ps.hats[0], ps.hats[1], ps.norms[0], ps.norms[1], ps.state1, ps.state2, pairs_b, ps.lists[0],
                        items_b, inc_b, ps.partials, multi_b, cfg.margin, cfg.optimizer, cfg.learning_rate, 1, tr.loss_buf,
                        reset_lists=False, run_finish=False, dense_skip=idx.skip(0),   # the small finishing launch of multi-segment rows is not in the burst
                        direction=ps.direction)
        eb1.record()
        torch.cuda.synchronize()
        ps.lists[0].clear()
    else:
        torch.cuda.synchronize()
        eb0.record()
        for _ in range(burst):
            if gen._pending <= 0:
                gen.start_one_epoch(steps_per_epoch)
            tr._accumulate_next_batch()
        eb1.record()
        torch.cuda.synchronize()
    burst_ms = eb0.elapsed_time(eb1) / burst
    # Owner-computes path: a timed step IS one k_pull_step launch (kge_pull_run enqueues them back to back), walking the epoch's
    # batches with evolving tables, so the kernel's average duration is taken from the events around the timed region itself
    # (it includes the ~1 us dispatch gap between consecutive launches and agrees with the rocprofv3 average of the same
    # command, profiles/r03_kernel_stats.md).  The burst replays ONE batch on the initial tables with its index slice hot in
    # L2 -- a lower bound, reported alongside.  Push path: several launches per step, the burst is the kernel's own duration.
    kern_ms = region_ms_per_step if pull else burst_ms
    reset_model()
    alg_bytes = 2 * per_rank_batch * TRAIN_BYTES_PER_SCORED_TRIPLE

    tr.sync_model()
    # ---- eval leg: filtered ranks of n_eval test triples per rank
    ev = Evaluator(model, cfg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev.rank_all(my_test, n_eval)  # first pass: builds the per-query filter CSR, uploads it, loads the code objects
    torch.cuda.synchronize()
    eval_first_ms = (time.perf_counter() - t0) * 1e3
    eval_setup = dict(getattr(ev, "setup_stats", {}) or {})
    eval_setup["csr_ms_cold"] = eval_setup.get("csr_ms")
    # the same set-up again with the code objects loaded (a second Evaluator: nothing cached), for this leg's queries and for the
    # whole FB15k-shape test split (59 071 queries): what a full_test() pays once
    for label, qs in (("csr_ms", my_test), ("csr_ms_full_test_split", test)):
        ev_w = Evaluator(model, cfg)
        ev_w._known_dev = ev._known_dev          # (the 14 MB upload of train + valid + test is per run, not per split)
        ev_w._device_inputs(qs, len(qs))
        eval_setup[label] = ev_w.setup_stats["csr_ms"]
        eval_setup[label + "_queries"] = len(qs)
    del ev_w
    passes = []
    for _rep in range(REPEATS):      # one pass per timed region, as for the train leg: median over REPEATS
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        ranks = ev.rank_all(my_test, n_eval)
        e1.record()
        barrier()
        te = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
        passes.append((float(te.item()), e0.elapsed_time(e1)))
    by_wall_e = sorted(passes)
    edt, eval_kern_ms = by_wall_e[len(by_wall_e) // 2]
    eval_spread = {"repeats": REPEATS, "ms_per_pass_min": by_wall_e[0][0] * 1e3, "ms_per_pass_max": by_wall_e[-1][0] * 1e3}
    eval_value = n_eval * world / edt
    eval_elements = 2.0 * n_eval * E * DIM                       # (query, candidate, k) elements per pass
    eval_elem_rate = eval_elements / (eval_kern_ms * 1e-3)
    valu_peak_elems = VALU_SIMDS * 64.0 / L1_SWEEP_CYCLES_PER_ELEMENT_PER_WAVE * VALU_PEAK_CLOCK_HZ
    eval_alg = 2.0 * n_eval * E * EVAL_BYTES_PER_CANDIDATE
    mean_rank = float(ranks[:2].float().mean().item()) + 1.0

    # ---- the reference's DEFAULT batch (B=128, common.py:48) through Trainer.train_model_epoch: the launch-bound
    # regime (N=1 only; informational, not `value`)
    small = None
    if world == 1:
        cfg_s = make_config(E, R, 128 * 400, 128, device, hidden_size=DIM, l1_flag=True)
        cfg_s.knowledge_graph = cfg.knowledge_graph
        torch.manual_seed(0)
        tr_s = Trainer(pw.TransE(**cfg_s.__dict__), cfg_s)
        tr_s.build_model()
        tr_s.generator = tr_s._new_generator()
        dts = timed_epochs(tr_s, 400)
        small = {"batch": 128, "value": 256 / dts, "unit": "scored triples/s", "ms_per_step": dts * 1e3,
                 "mode": ("hipGraph replay, 8 steps per graph, of fused step + Adam (next step's state derived inside the Adam launch)" if tr_s._graph is not None
                          else "owner-computes step, one launch per step, the epoch enqueued by one native call (kge_pull_run); compact incidence index" if getattr(tr_s, "_pull", None) is not None
                          else "eager")}
        del tr_s

    # ---- HBM traffic of every leg, observed in THIS run by two rocprofv3 counter passes over a short child run (N=1, rank 0)
    live, live_meta = (None, "disabled (--no-live-pmc)") if (args.no_live_pmc or world > 1) else live_pmc(args)

    out = None
    kernel_label = ("k_pull_eval<L1,G=32> + k_pull_step<Adam,L1,G=32,two-phase> (owner-computes step in two launches: every pair evaluated "
                    "once -- 4 row gathers, hinge, 2-bit direction codes --, then one owner per row sums the records of its incidences, "
                    "normalisation backward, dense Adam; no atomics.  Durations and traffic are the SUM of the two kernels)" if two_phase else
                    "k_pull_step<Adam,G=32,NCH=4> (owner-computes step: per-row re-evaluation of incident pairs, hinge, backward, "
                    "normalisation backward, dense Adam; no atomics)" if pull else
                    "k_pull_step<gradient,G=32,NCH=4> (owner-computes gradient of the rank's share of the batch: per-row re-evaluation of "
                    "incident pairs, hinge, backward, normalisation backward; dense gradient rows written once, no atomics)" if pull_dp else
                    "k_transe_pair_sampled<G=32,NCH=4,CH=4> (sampler + score(+) + score(-) + hinge + backward)")
    traffic_kernels = None
    if live is not None and "bytes" in live.get("C1_train", {}):
        leg = live["C1_train"]
        traffic, traffic_src = leg["bytes"], ("observed in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) "
                                             "over %d steps of the same step path in a child process, all kernels between the marker launches" % leg["units"])
        traffic_kernels = leg["kernels"]
    elif two_phase:
        t1, s1 = pmc_traffic("kge::k_pull_step<1, true, 32, 1, true>", per_rank_batch, fetch_scale=2.0)
        t2, s2 = pmc_traffic("kge::k_pull_eval<true, 32", per_rank_batch, fetch_scale=2.0)
        traffic, traffic_src = (t1 + t2, "committed passes (not observed in this run: %s): %s + %s" % (live_meta, s1, s2)) if (t1 is not None and t2 is not None) else (None, None)
    else:
        traffic, traffic_src = (None, None) if pull_dp else pmc_traffic(
            "kge::k_pull_step<1, true, 32, 1, false>" if pull else "kge::k_transe_pair_sampled<32, 4, 4", per_rank_batch, fetch_scale=2.0 if pull else 1.0)
        if traffic is not None:
            traffic_src = "committed passes (not observed in this run: %s): %s" % (live_meta, traffic_src)
    nominal = alg_bytes / (kern_ms * 1e-3) / 1e9
    achieved = traffic / (kern_ms * 1e-3) / 1e9 if traffic else nominal
    valu = valu_record(traffic_kernels)
    valu_frac = None
    if valu:   # all VALU wave-instructions of a step over the step's launch time, same convention as the eval leg's roof
        insts = sum(v["valu_wave_instructions_per_step"] for v in valu.values() if isinstance(v, dict) and v.get("valu_wave_instructions_per_step"))
        valu_frac = insts / (kern_ms * 1e-3) / (VALU_SIMDS * VALU_PEAK_CLOCK_HZ / 2.0)
    # what bounds the owner kernel when it is not bandwidth: one residency round of owner groups, each a chain of dependent loads
    # (item -> row + optimiser state + visit lists -> records / direction codes -> stores).  Hop latency under load from the committed
    # random-row microbenchmark, interpolated in the number of concurrently resident groups.
    latency_model = None
    if pull and not pull_dp:
        ps_, idx_ = tr._pull_state()
        n_groups = int(idx_.batch(0)[2].shape[0])       # work items = owner groups of one launch (32 lanes each)
        resident = 256 * 4 * 8 * 2     # CUs x SIMDs x 8 waves (32-62 VGPRs) x two 32-lane owner groups per wave
        load = min(1.0, max(0.0, (min(n_groups, resident) - 8192) / (32768 - 8192.0)))
        hop_us = HOP_US_AT_8K_GROUPS + load * (HOP_US_AT_32K_GROUPS - HOP_US_AT_8K_GROUPS)
        hops = 4 if two_phase else 5    # item -> {row, state, lists} -> {records + codes | three hat rows per visit -> ...} -> store drain
        rounds = max(1.0, n_groups / float(resident))
        stride = K.pull_hat_stride(DIM)
        row_bytes = (E + R) * (6 * DIM * 4 + stride * 4 + 4)        # p, m, v read and written; normalised copy + norm written
        stream_us = row_bytes / 5.0e6                                # at the ~5 TB/s an L2 / Infinity-Cache resident sweep streams (k_opt over the same tables: 7-8 us)
        visits_us = 4.3 if two_phase else None                       # measured with the visits compiled out (profiles/r03_experiments.md section 11)
        latency_model = {"owner_groups": n_groups, "resident_groups": resident, "residency_rounds": rounds,
                         "dependent_hops_per_owner": hops, "hop_latency_us": hop_us,
                         "hop_latency_source": "profiles/r03_gather_bench.txt (chain G=32: 21.53 us / 8 hops at 32768 groups, 9.23 us / 8 at 8192)",
                         "hop_chain_us": rounds * hops * hop_us, "row_io_bytes": row_bytes, "row_io_stream_us": stream_us,
                         "visits_us": visits_us,
                         "predicted_owner_kernel_us": rounds * hops * hop_us + stream_us + (visits_us or 0.0),
                         "note": "the round-3 model of the owner launch (a chain of dependent loads per owner group + the rows' read-modify-write "
                                 "stream + the visits): kept for comparison -- it matched the 20.0 us of the round-3 kernel, but the round-4 "
                                 "measurements below say the kernel is bound by VALU issue, not by this sum",
                         "superseded_by": {
                             "source": "profiles/r04_experiments.md section 7 (per-workgroup wall_clock64 timestamps + HW_REG_HW_ID of an experiment "
                                       "build; SQ counter passes of this command in profiles/r04_pmc_traffic.json)",
                             "resident_workgroups_per_cu": 7, "resident_limit": "SGPR file (81 SGPRs per wave)",
                             "workgroups_per_launch": 2173, "resident_slots": 1792,
                             "workgroup_lifetime_us": {"mean": 11.8, "p10": 6.9, "p90": 15.5, "max": 17.5},
                             "valu_busy_frac_round3_kernel": 0.75, "valu_wave_instructions_per_wave_round3_kernel": 1005,
                             "valu_wave_instructions_per_wave_after_first_cut": 804,
                             "what_helped": "fewer VALU instructions per visit (2-bit two's-complement codes, coefficient tables, integer half-unit "
                                            "sums): 29.8 -> 27.7 us per step; residency, prefetch depth and layering changes did not"}}
    if rank == 0:
        out = {
            "metric": "scored triples/sec (train) + test-triples ranked/sec, FB15k TransE d=100",
            "value": value, "unit": "scored triples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "repeats": REPEATS, "ms_per_step_min": region_spread["ms_per_step_min"],
            "ms_per_step_max": region_spread["ms_per_step_max"], "timed_region_s": dt, "ms_per_step_all": region_spread["ms_per_step_all"],
            "timing_note": "the timed region (--steps steps between barrier + synchronize) is run %d times, tables reset in between; "
                           "value / ms_per_step / timed_region_s are the MEDIAN region's" % REPEATS,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "FB15k-shape TransE d=100 L1 margin=1.0 hinge, neg_rate=1, dense Adam lr=0.01, "
                                   "on-device uniform corruption; E=14951 R=1345 train=483142",
                       "batch_per_gpu": per_rank_batch, "global_batch": per_rank_batch * world,
                       "scored_triples_per_step": scored_per_step, "parallelism": "dp%d" % world,
                       "warmup_steps_run": args.warmup + warm_extra,
                       "model_state": "timed steps start from the freshly initialised tables (reset after warm-up)",
                       "step_path_short": ("owner-computes two-phase: k_pull_eval + k_pull_step per step (kge_pull_run), no atomics" if two_phase else
                                           "owner-computes: one k_pull_step per step (kge_pull_run), no atomics" if pull else
                                           "owner-computes gradient (k_pull_step, no atomics) + exchange + kge_optimizer_step + kge_row_norms" if pull_dp else
                                           "push: kge_train_pairwise_hinge_sampled (atomic scatter) + kge_optimizer_step"),
                       "step_path": "owner-computes (pull), two-phase: kge_pull_run, k_pull_eval + k_pull_step<two-phase> per step (the next batch's sampler rides in the second launch)" if two_phase else
                                    "owner-computes (pull): kge_pull_run, one k_pull_step launch per step (the next batch's sampler rides in its leading blocks)" if pull else
                                    "owner-computes gradient (k_pull_step, KGE_OPT_GRADIENT: no atomics) + gradient exchange (see `collectives`) + kge_optimizer_step + kge_row_norms" if pull_dp else
                                    "push: kge_train_pairwise_hinge_sampled (atomic scatter) + kge_optimizer_step"},
            "roofline": {"kernel": kernel_label,
                         "kernel_short": ("k_pull_eval<L1,32> + k_pull_step<Adam,L1,32,two-phase> (sum of both launches)" if two_phase else
                                          "k_pull_step<Adam,G=32>" if pull else "k_pull_step<gradient,G=32>" if pull_dp else "k_transe_pair_sampled<32,4,4>"),
                         "traffic_src_short": (None if not traffic else "live rocprofv3 --pmc passes in this run" if traffic_kernels is not None
                                               else "committed profiles/ passes"),
                         "bound": "hbm", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "frac_basis": ("bytes that crossed the L2 <-> fabric boundary per step (PMC: 2 x FETCH_SIZE + WRITE_SIZE, `traffic`) / "
                                        "avg_launch_ms / peak: a physical fraction, <= 1 by construction" if traffic else
                                        "NO counter figure available (%s): algorithmic bytes / avg_launch_ms / peak -- nominal, see nominal_note" % (live_meta,)),
                         "traffic": traffic,
                         "traffic_source": traffic_src,
                         "traffic_note": ("2 x FETCH_SIZE (gfx950: 16-byte-per-lane reads are tallied at half, MI355X_MICROARCH.md HBM section) + WRITE_SIZE; "
                                          "the counters sit on the fabric side of the per-XCD L2s and include Infinity-Cache hits" if pull else "FETCH_SIZE + WRITE_SIZE, raw"),
                         "traffic_kernels": traffic_kernels,
                         "valu": valu,
                         "hbm_frac": achieved / HBM_PEAK_GBS if traffic else None,
                         "valu_frac": valu_frac,
                         "bound_note": ("neither roof is reached: counter bytes / time = hbm_frac of 8 TB/s, VALU issue = valu_frac of 1024 SIMDs x 1.2 G "
                                        "wave-instr/s; the launch is bound by the dependent-load chains of its owner groups (DESIGN.md section 4)"),
                         "frac_roof": ("hbm-counter: bytes at the L2 <-> fabric boundary per launch / avg_launch_ms / 8 TB/s" if traffic
                                       else "hbm-algorithmic (no counter pass in this run)"),
                         "algorithmic_frac": nominal / HBM_PEAK_GBS,
                         "algorithmic_roof": "SURVEY 8(d): 3628 B per scored triple x scored triples per launch / avg_launch_ms / 8 TB/s (can exceed 1: "
                                             "the owner-computes step does no gradient read-modify-write and its 6.5 MB of tables stay in L2 / Infinity Cache)",
                         "nominal_achieved": nominal, "nominal_frac": nominal / HBM_PEAK_GBS,
                         "nominal_note": ("ALGORITHMIC bytes of SURVEY section 8(d) (forward gathers + gradient read-modify-write + ids per scored "
                                          "triple: 3628 B) / avg_launch_ms.  The owner-computes step performs no gradient read-modify-write and "
                                          "gathers from 6.5 MB tables that stay in L2 / Infinity Cache, so this figure can exceed 1: it is kept "
                                          "for continuity with SURVEY 8(d), it is not a fraction of a roof the kernel can hit"),
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "latency_model": latency_model,
                         "avg_launch_ms": kern_ms,
                         "avg_launch_ms_method": ("HIP events on the launch stream around the timed region / steps (%s per step; "
                                                  "includes the dispatch gaps between consecutive launches)" % ("two launches" if two_phase else "one launch") if pull else
                                                  "HIP events around a burst of %d back-to-back launches of the kernel right after "
                                                  "the timed region (= rocprofv3 kernel duration)" % burst),
                         "burst_launch_ms": burst_ms,
                         "burst_launch_ms_note": "%d back-to-back launches of ONE batch on the initial tables (index slice hot in L2): lower bound" % burst,
                         "timed_region_event_ms": event_ms,
                         "timed_region_event_ms_note": "HIP events around each launch inside the timed region: includes "
                                                       "the dispatch gap in front of the kernel"},
            "eval": {"value": eval_value, "unit": "test triples ranked/s", "test_triples_per_gpu": n_eval,
                     "ms_per_pass": edt * 1e3, "repeats": REPEATS, "ms_per_pass_min": eval_spread["ms_per_pass_min"],
                     "ms_per_pass_max": eval_spread["ms_per_pass_max"], "mean_rank_check": mean_rank,
                     "setup_ms": eval_setup.get("csr_ms", max(0.0, eval_first_ms - edt * 1e3)),
                     "setup": dict(eval_setup, first_pass_ms=eval_first_ms,
                                   what="per-query filter lists (hr_t / tr_h of train + valid + test, data/kgcontroller.py:410-428) as CSR on the "
                                        "device, built once per evaluated split; the reference looks the sets up per query inside its rank loop"),
                     "roofline": {"kernel": "kge_eval_ranks pipeline (k_eval_sweep<L1,QT=16> dominant)", "kernel_short": "k_eval_sweep<L1,QT=16>", "bound": "valu",
                                  "achieved": eval_elem_rate / 1e12, "peak": valu_peak_elems / 1e12,
                                  "unit": "T (query,candidate,k) elements/s", "frac": eval_elem_rate / valu_peak_elems,
                                  "valu_issues_per_element": L1_SWEEP_ISSUES_PER_ELEMENT,
                                  "achieved_lane_issues_per_s": eval_elem_rate * L1_SWEEP_ISSUES_PER_ELEMENT,
                                  "peak_note": "1024 SIMDs x 64 lanes / (2 plain VALU issues x 2 cycles) per element x 2.4 GHz; "
                                               "the same instruction mix with operands in registers sustains "
                                               "%.1f T elements/s (tools/valu_bench.hip, profiles/r02_valu_bench.txt)"
                                               % L1_SWEEP_MICROBENCH_TELEMS,
                                  "frac_of_microbench_ceiling": eval_elem_rate / 1e12 / L1_SWEEP_MICROBENCH_TELEMS,
                                  "algorithmic_GBps": eval_alg / (eval_kern_ms * 1e-3) / 1e9,
                                  "algorithmic_note": "400 B per scored candidate (SURVEY 8d); each candidate tile is "
                                                      "reused by 16 queries from registers, so this exceeds the HBM "
                                                      "peak and is not the bound",
                                  "traffic": None, "traffic_source": None}},
        }
        er = out["eval"]["roofline"]
        if live is not None and "bytes" in live.get("C1_eval", {}):
            leg = live["C1_eval"]
            er["traffic"], er["traffic_source"] = leg["bytes"], "observed in this run (rocprofv3 counter passes, %d passes, all kernels of a pass)" % leg["units"]
            er["traffic_kernels"] = leg["kernels"]
        else:
            tsw, ssw = pmc_traffic("kge::k_eval_sweep<0, 0, 16", 32768, fetch_scale=2.0)
            if tsw is not None:
                er["traffic"], er["traffic_source"] = tsw, "committed passes, k_eval_sweep only (not observed in this run: %s): %s" % (live_meta, ssw)
        if er["traffic"]:
            er["traffic_GBps"] = er["traffic"] / (eval_kern_ms * 1e-3) / 1e9
            er["traffic_hbm_frac"] = er["traffic_GBps"] / HBM_PEAK_GBS
        if setup is not None:
            setup["gpu_step_equivalents"] = setup["host_wall_ms_warm"] / (dt / args.steps * 1e3)
            out["setup_ms"] = setup["host_wall_ms_warm"]
            out["setup"] = setup
        if small is not None:
            out["train_reference_default_batch"] = small
        out["live_pmc"] = live_meta if not isinstance(live_meta, str) else {"unavailable": live_meta}
    if world == 1 and not args.no_extra_configs:
        extra = {}
        for key in EXTRA_CONFIGS:
            try:
                extra[key] = run_extra_config(key, device)
                for legname, field in ((key + "_train", "train"), (key + "_eval", "eval")):
                    leg = (live or {}).get(legname)
                    if leg and "bytes" in leg:
                        secs = extra[key]["step_us"] * 1e-6 if field == "train" else extra[key]["eval_ms_per_pass"] * 1e-3
                        extra[key][field + "_traffic"] = {
                            "bytes_per_" + ("step" if field == "train" else "pass"): leg["bytes"], "fetch_raw_bytes": leg["fetch_raw_bytes"],
                            "write_bytes": leg["write_bytes"], "GBps": leg["bytes"] / secs / 1e9, "hbm_frac": leg["bytes"] / secs / 1e9 / HBM_PEAK_GBS,
                            "source": "observed in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over %d %s" % (
                                leg["units"], "steps" if field == "train" else "passes"),
                            "kernels": leg["kernels"]}
            except Exception as e:  # an `extra` record must never take the headline line down
                extra[key] = {"error": "%s: %s" % (type(e).__name__, e)}
        out["extra"] = extra
    replicas_identical = None
    if world > 1:   # every replica must hold the same tables after the run (data parallel with a deterministic exchange): always checked
        ref_p = tr.flat.param.clone()
        dist.broadcast(ref_p, src=0)
        same = torch.tensor([float(torch.equal(ref_p, tr.flat.param))], device=device)
        dist.all_reduce(same, op=dist.ReduceOp.MIN)
        replicas_identical = bool(same.item())
        del ref_p
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(H)
        if world > 1:
            out["phases_us"] = phases_us
            allred = bool(getattr(tr, "_dp_allreduce", False))
            out["collectives"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                                  "NCCL_ALGO": os.environ.get("NCCL_ALGO", "(unset: RCCL picks)"),
                                  "NCCL_PROTO": os.environ.get("NCCL_PROTO", "(unset)"),
                                  "rccl": rccl_setup_summary(),
                                  "per_step": ("all_reduce(flat grad, %d B); every rank steps every row (tables <= 32 MB)" % (tr.flat.numel * 4)
                                               if allred else
                                               "reduce_scatter(flat grad, %d B) + all_gather(flat param)" % (tr.flat.numel * 4)),
                                  "captured": tr._graph is not None,
                                  "optimizer_shard_floats": tr.flat.shard_numel}
            out["predicted_step_us"] = predicted_step_us(world, allred)
            out["replicas_identical"] = replicas_identical
        if args.full_line:
            print(json.dumps(out), flush=True)   # the complete record as an EARLIER line (opt-in; it is tens of KB)
        if world > 1 and os.environ.get("KGE_BENCH_CHECK_REPLICAS") == "1":  # tests grep this line
            print("REPLICAS_IDENTICAL %d" % int(replicas_identical), flush=True)
        print(compact_line(out, write_detail(out)), flush=True)   # ALWAYS the last stdout line, < COMPACT_LIMIT bytes
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
