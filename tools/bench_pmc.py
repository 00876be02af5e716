"""bench.py's hardware-counter harness (moved out of bench.py in round 6; no behaviour change): a child run of the same workloads under
two or three `rocprofv3 --pmc` passes (FETCH_SIZE / WRITE_SIZE, optionally the SQ pass), segmented by marker launches, read back from
the rocpd databases, so that the bench line's `roofline.traffic` is observed in THE SAME run.  Entry points bench.py uses:
live_pmc(args), pmc_child(args) (the `--pmc-child` mode), pmc_traffic(...) (committed passes as a fallback), valu_record(...)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from bench import VALU_PEAK_CLOCK_HZ, VALU_SIMDS, reset_headline, run_headline_steps, setup_headline  # noqa: E402
from tools.bench_extra import EXTRA_CONFIGS, build_extra_config  # noqa: E402

# marker tags of the counter child (kge_debug_marker: grid.x = 64 x tag); a segment runs from its tag to the next marker
PMC_TAGS = {"C1_train": 101, "C1_eval": 102, "C1_small": 103, "C2_train": 111, "C2_eval": 112, "C3_train": 121, "C3_eval": 122,
            "C4_train": 131, "C4_eval": 132, "end": 99}
PMC_C1_STEPS, PMC_EVAL_REPS, PMC_EXTRA_STEPS, PMC_SMALL_STEPS = 28, 1, 20, 400
# third (optional) pass of the counter child: what the SQ sees -- VALU / VMEM / SALU wave-instructions, and where a wave's time goes
# (parked on s_waitcnt / issue-stalled / issuing).  Eight SQ counters fit one pass (MI355X_MICROARCH.md, counter table).
SQ_PASS = "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"


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
