#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on MI355X: scored triples/sec (train) + test triples ranked/sec, FB15k-shape TransE d=100
(configs[1]), synthetic data of that shape, random-init tables, fp32.

A *step* is one pass of the training hot path over one batch of the HBM-resident train split.  N = 1: the owner-computes step
(csrc/kge_pull.hip) -- k_pull_eval (every pair once: four row gathers, hinge, direction codes) + k_pull_step (one owner per table
row: sums its incidences, normalisation backward, dense Adam; the NEXT batch's negative sampler rides in its leading blocks), all
steps of the region enqueued by one native call (kge_pull_run).  N > 1: each rank computes the gradient of its share of the batch
with the same kernel, one RCCL exchange of the flat gradient, the optimiser, row norms.  Nothing is skipped or cached inside the
timed region, which is run REPEATS times (median reported).  Then, untimed for `value`: the filtered-rank sweep of the whole test
split (`eval`), the other BASELINE configs (`extra`, tools/bench_extra.py), the reference's CPU path on the same host (`cpu_baseline`,
tools/bench_cpu.py) and the hardware-counter passes that give `roofline.traffic` (tools/bench_pmc.py: a child run under rocprofv3
--pmc FETCH_SIZE / WRITE_SIZE, separate passes, segmented by marker launches; `--no-live-pmc` uses the committed passes).

Launch:  python bench.py [--gpus N --steps K --warmup W]      (N > 1 without a torch.distributed environment re-executes itself under
         torch.distributed.run, one rank per GPU; an existing torchrun environment is used as is)
Rank 0 writes the complete record to gpurun_out/bench_detail.json and prints ONE compact JSON line (< 4 KB) as its LAST stdout line.
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
        line["default_batch_128"] = {"value": _r(sm.get("value")), "ms_per_step": _r(sm.get("ms_per_step"), 4),
                                     "adam_floor_frac": _r((sm.get("floor") or {}).get("frac_of_hbm_peak"), 3)}
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
        from tools import bench_pmc
        return bench_pmc.pmc_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))

    import torch
    import torch.distributed as dist
    from tools import bench_extra, bench_pmc   # the C2-C4 `extra` records; the rocprofv3 counter-pass harness
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
        dts = bench_extra.timed_epochs(tr_s, 400)
        adam_bytes = (E + R) * DIM * 4 * 7     # dense Adam moves every row every step: p, m, v read and written + the gradient
        small = {"batch": 128, "value": 256 / dts, "unit": "scored triples/s", "ms_per_step": dts * 1e3,
                 "floor": {"what": "the B-independent dense Adam sweep: %.1f MB per step" % (adam_bytes / 1e6), "us_at_hbm_peak": adam_bytes / HBM_PEAK_GBS / 1e3,
                           "achieved_GBps": adam_bytes / dts / 1e9, "frac_of_hbm_peak": adam_bytes / dts / 1e9 / HBM_PEAK_GBS},
                 "mode": ("hipGraph replay of fused step + Adam" if tr_s._graph is not None else
                          "owner-computes step, one launch per step (kge_pull_run)" if getattr(tr_s, "_pull", None) is not None else "eager")}
        del tr_s
    # ---- HBM traffic of every leg, observed in THIS run by two rocprofv3 counter passes over a short child run (N=1, rank 0)
    live, live_meta = (None, "disabled (--no-live-pmc)") if (args.no_live_pmc or world > 1) else bench_pmc.live_pmc(args)

    out = None
    traffic_kernels = None
    if live is not None and "bytes" in live.get("C1_train", {}):
        leg = live["C1_train"]
        traffic, traffic_src = leg["bytes"], ("observed in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) "
                                             "over %d steps of the same step path in a child process, all kernels between the marker launches" % leg["units"])
        traffic_kernels = leg["kernels"]
    elif two_phase:
        t1, s1 = bench_pmc.pmc_traffic("kge::k_pull_step<1, true, 32, 1, true>", per_rank_batch, fetch_scale=2.0)
        t2, s2 = bench_pmc.pmc_traffic("kge::k_pull_eval<true, 32", per_rank_batch, fetch_scale=2.0)
        traffic, traffic_src = (t1 + t2, "committed passes (not observed in this run: %s): %s + %s" % (live_meta, s1, s2)) if (t1 is not None and t2 is not None) else (None, None)
    else:
        traffic, traffic_src = (None, None) if pull_dp else bench_pmc.pmc_traffic(
            "kge::k_pull_step<1, true, 32, 1, false>" if pull else "kge::k_transe_pair_sampled<32, 4, 4", per_rank_batch, fetch_scale=2.0 if pull else 1.0)
        if traffic is not None:
            traffic_src = "committed passes (not observed in this run: %s): %s" % (live_meta, traffic_src)
    nominal = alg_bytes / (kern_ms * 1e-3) / 1e9
    achieved = traffic / (kern_ms * 1e-3) / 1e9 if traffic else nominal
    valu = bench_pmc.valu_record(traffic_kernels)
    valu_frac = None
    if valu:   # all VALU wave-instructions of a step over the step's launch time, same convention as the eval leg's roof
        insts = sum(v["valu_wave_instructions_per_step"] for v in valu.values() if isinstance(v, dict) and v.get("valu_wave_instructions_per_step"))
        valu_frac = insts / (kern_ms * 1e-3) / (VALU_SIMDS * VALU_PEAK_CLOCK_HZ / 2.0)
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
                                           "push: kge_train_pairwise_hinge_sampled (atomic scatter) + kge_optimizer_step")},
            "roofline": {"kernel": ("k_pull_eval<L1,32> + k_pull_step<Adam,L1,32,two-phase> (sum of both launches)" if two_phase else
                                          "k_pull_step<Adam,G=32>" if pull else "k_pull_step<gradient,G=32>" if pull_dp else "k_transe_pair_sampled<32,4,4>"),
                         "traffic_src_short": (None if not traffic else "live rocprofv3 --pmc passes in this run" if traffic_kernels is not None
                                               else "committed profiles/ passes"),
                         "bound": "hbm", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
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
                         "nominal_frac": nominal / HBM_PEAK_GBS,   # (the same figure under its pre-round-6 name)
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "avg_launch_ms": kern_ms,
                         "avg_launch_ms_method": ("HIP events on the launch stream around the timed region / steps (%s per step; "
                                                  "includes the dispatch gaps between consecutive launches)" % ("two launches" if two_phase else "one launch") if pull else
                                                  "HIP events around a burst of %d back-to-back launches of the kernel right after "
                                                  "the timed region (= rocprofv3 kernel duration)" % burst),
                         "burst_launch_ms": burst_ms,
                         "burst_launch_ms_note": "%d back-to-back launches of ONE batch on the initial tables (index slice hot in L2): lower bound" % burst,
                         "timed_region_event_ms": event_ms},   # (HIP events around each launch inside the timed region, dispatch gap included)
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
            tsw, ssw = bench_pmc.pmc_traffic("kge::k_eval_sweep<0, 0, 16", 32768, fetch_scale=2.0)
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
        for key in bench_extra.EXTRA_CONFIGS:
            try:
                extra[key] = bench_extra.run_extra_config(key, device)
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
            from tools import bench_cpu
            out["cpu_baseline"] = bench_cpu.cpu_baseline(H)
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
