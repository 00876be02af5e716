#!/usr/bin/env python
"""Host emulation (numpy, CPU only) of summation orders for the dot-product rank sweep: which order of the k chain would move the HIP
ranks closest to the reference's / to float64 on the 512-triple C2 fixture?

The HIP sweep's energy of (query, candidate) is ONE in-order fmaf chain over k (the order `v_mfma_f32_16x16x4_f32` accumulates in,
reproduced on the VALU by k_eval_target_filter_chain).  On tests/golden/ref_full_ranks_c2_complex.npz float64 sided with the reference
on every one of the 10 rank entries the HIP path differed in.  Emulated here, bit-faithfully (every fmaf as a correctly rounded
float32 of the exact double product-sum):

  chain          the shipped order: acc = fmaf(q_k, c_k, acc), k = 0 .. K-1
  split2/4/8     S contiguous K segments, each its own chain, summed in fixed order ((p0 + p1) + p2) + ...
  chunk16/32/64  a chain per chunk of C consecutive k, folded into a running total after every chunk (two accumulator sets on the
                 matrix cores whatever the chunk count: `total += acc; acc = 0`)

Only candidates whose float64 energy lies within BAND of the true candidate's can change a rank, so the emulation runs on those pairs
only (the rest keep their float64 ordering under every variant: BAND is 400x the largest fp32 deviation observed).

Writes profiles/r06_rank_chain_study.json.  Usage: python tools/rank_chain_study.py [c2_complex]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import golden_util as gu  # noqa: E402

f32 = np.float32


def fma32(a, b, c):
    """float32 fmaf(a, b, c) for float32 arrays: the double product of two floats is exact, the double sum is rounded once to 53
    bits and once more to 24 -- a double rounding differs from a true fmaf only on exact half-way cases of the 53-bit sum."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def chain(q, c, lo, hi):
    acc = np.zeros(q.shape[0], f32)
    for k in range(lo, hi):
        acc = fma32(q[:, k], c[:, k], acc)
    return acc


def variants(q, c):
    K = q.shape[1]
    out = {"chain": chain(q, c, 0, K)}
    for S in (2, 4, 8):
        seg = -(-K // S)
        parts = [chain(q, c, lo, min(K, lo + seg)) for lo in range(0, K, seg)]
        tot = parts[0]
        for p in parts[1:]:
            tot = (tot + p).astype(f32)
        out["split%d" % S] = tot
    for C in (16, 32, 64):
        tot = np.zeros(q.shape[0], f32)
        for lo in range(0, K, C):
            tot = (tot + chain(q, c, lo, min(K, lo + C))).astype(f32)
        out["chunk%d" % C] = tot
    return out


def complex_queries(P, trip):
    """The float32 query vectors k_eval_queries<ComplEx> builds (separately rounded products, -ffp-contract=off):
    tail sweep <h o r, conj-free form>, head sweep <t o conj(r)>; candidates are [re | im]."""
    er, ei, rr, ri = (P[k].astype(f32) for k in ("ent_embeddings_real", "ent_embeddings_img", "rel_embeddings_real", "rel_embeddings_img"))
    h, r, t = trip[:, 0], trip[:, 1], trip[:, 2]
    qt = np.concatenate([er[h] * rr[r] - ei[h] * ri[r], ei[h] * rr[r] + er[h] * ri[r]], 1)
    qh = np.concatenate([er[t] * rr[r] + ei[t] * ri[r], ei[t] * rr[r] - er[t] * ri[r]], 1)
    return qt, qh, np.concatenate([er, ei], 1)


def main(name="c2_complex"):
    spec, P, train, valid, test, ids, batch = gu.fullsize_inputs(name)
    z = np.load(os.path.join(ROOT, "tests", "golden", "ref_full_ranks_%s.npz" % name))
    n = int(z["n"])
    q = test[:n]
    ref, r64 = z["ranks"], z["ranks64"]
    qt, qh, cand = complex_queries(P, q)
    cand64 = cand.astype(np.float64)
    names = None
    delta = {}       # variant -> [2, n] rank correction relative to float64 (raw ranks; the filtered ranks move by the same pairs)
    dev = {}         # variant -> largest |fp32 - float64| energy seen
    n_pairs = 0
    for side, (qv, truth) in enumerate(((qt, q[:, 2]), (qh, q[:, 0]))):
        for lo in range(0, n, 64):
            qs = qv[lo:lo + 64]
            s64 = -(qs.astype(np.float64) @ cand64.T)                         # [64, E]
            st = s64[np.arange(len(qs)), truth[lo:lo + 64]]
            scale = np.abs(qs).astype(np.float64).sum(1) * np.abs(cand64).max() / np.sqrt(qs.shape[1])
            band = 400 * 2.0 ** -24 * scale
            qi, ei = np.nonzero(np.abs(s64 - st[:, None]) <= band[:, None])
            n_pairs += len(qi)
            v_e = variants(qs[qi], cand[ei])
            v_t = variants(qs, cand[truth[lo:lo + 64]])
            names = list(v_e)
            for v in names:
                lt32 = (-v_e[v] < -v_t[v][qi])
                lt64 = s64[qi, ei] < st[qi]
                d = np.zeros(len(qs), np.int64)
                np.add.at(d, qi, lt32.astype(np.int64) - lt64.astype(np.int64))
                delta.setdefault(v, np.zeros((2, n), np.int64))[side, lo:lo + 64] = d
                dev[v] = max(dev.get(v, 0.0), float(np.abs(-v_e[v].astype(np.float64) - s64[qi, ei]).max()))
    # raw ranks: row 0 = head sweep, row 1 = tail sweep (side 0 above = tail sweep)
    report = {"case": name, "test_triples": n, "rank_entries": 2 * n, "pairs_emulated": int(n_pairs),
              "reference_vs_float64": int((ref[:2] != r64[:2]).sum()), "variants": {}}
    for v in names:
        mine = np.stack([r64[0] + delta[v][1], r64[1] + delta[v][0]])
        differ = mine != ref[:2]
        report["variants"][v] = {
            "raw_ranks_differing_from_reference": int(differ.sum()),
            "of_those_float64_sides_with_variant": int((differ & (mine == r64[:2])).sum()),
            "of_those_float64_sides_with_reference": int((differ & (ref[:2] == r64[:2])).sum()),
            "raw_ranks_differing_from_float64": int((mine != r64[:2]).sum()),
            "max_abs_energy_deviation_from_float64": dev[v]}
        print(v, report["variants"][v], flush=True)
    out = os.path.join(ROOT, "profiles", "r06_rank_chain_study.json")
    doc = json.load(open(out)) if os.path.exists(out) else {}
    doc[name] = report
    json.dump(doc, open(out, "w"), indent=1)
    print("reference vs float64:", report["reference_vs_float64"], "of", 2 * n, "raw rank entries; wrote", out)


if __name__ == "__main__":
    main(*(sys.argv[1:] or ["c2_complex"]))
