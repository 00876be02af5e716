"""CPU oracle for the KGE scoring / training / ranking hot path.

TEST INFRASTRUCTURE, NOT PRODUCT.  Only `tests/`, `__graft_entry__.smoke()` and
the `cpu_baseline` leg of `bench.py` may import this file.  The shipped path
(`pykg2vec_amd/`) never does; it runs HIP kernels or fails loudly.

This is a from-scratch numpy restatement (explicit formulas + hand-derived
gradients, no autograd, no torch) of the algorithms of Sujit-O/pykg2vec v0.0.52
for the path BASELINE.json names.  Every function cites the reference lines it
follows (paths relative to /root/reference).  The arithmetic itself lives in a
third-party dependency of the reference (PyTorch ATen; the reference pins
`torch<1.7.0`, requirements.txt:1; this container has 2.10.0) and numpy's
MT19937 RNG for the sampler, so:

PARITY PINNING.  The reference ships no golden vectors / known-answer tests for
this path.  The oracle is pinned instead against outputs
of the live reference executed in the build container: `oracle/make_golden.py`
imports /root/reference (torch 2.10 CPU, fp32), records scores, losses, dense
gradients, post-optimizer weights, candidate orderings and ranks into
`tests/golden/*.npz`, and `tests/test_oracle_golden.py` holds this file to
those vectors (fp32: atol 1e-5 + rtol 1e-5; integer ranks exact).

`dtype` arguments let tests evaluate in float64 to bound fp32 ordering noise.
"""
import numpy as np

EPS_NORMALIZE = 1e-12  # torch.nn.functional.normalize default eps
PI = 3.14159265358979323846  # literal used by pairwise.py:776

# parameter (state_dict) names per model, in `parameter_list` order
# pairwise.py:44-52,128-139,213-227,750-761,817-825,889-906 ; pointwise.py:42-66,149-161,417-423
PARAM_NAMES = {
    "transe": ["ent_embeddings", "rel_embeddings"],
    "transh": ["ent_embeddings", "rel_embeddings", "w"],
    "transd": ["ent_embeddings", "rel_embeddings", "ent_mappings", "rel_mappings"],
    "rotate": ["ent_embeddings", "ent_embeddings_imag", "rel_embeddings"],
    "rescal": ["ent_embeddings", "rel_matrices"],
    "ntn": ["ent_embeddings", "rel_embeddings", "mr1", "mr2", "br", "mr"],
    "distmult": ["ent_embeddings", "rel_embeddings"],
    "complex": ["ent_embeddings_real", "ent_embeddings_img", "rel_embeddings_real", "rel_embeddings_img"],
    "complexn3": ["ent_embeddings_real", "ent_embeddings_img", "rel_embeddings_real", "rel_embeddings_img"],
    "analogy": ["ent_embeddings", "rel_embeddings", "ent_embeddings_real", "ent_embeddings_img",
                "rel_embeddings_real", "rel_embeddings_img"],
    # pairwise.py:299-320 ; pointwise.py:339-356,481-500,620-676
    "transm": ["ent_embeddings", "rel_embeddings"],
    "transr": ["ent_embeddings", "rel_embeddings", "rel_matrix"],  # pairwise.py:389-402
    "cp": ["sub_embeddings", "rel_embeddings", "obj_embeddings"],
    "simple": ["ent_head_embeddings", "ent_tail_embeddings", "rel_embeddings", "rel_inv_embeddings"],
    "simple_ignr": ["ent_head_embeddings", "ent_tail_embeddings", "rel_embeddings", "rel_inv_embeddings"],
    "quate": ["ent_s_embedding", "ent_x_embedding", "ent_y_embedding", "ent_z_embedding",
              "rel_s_embedding", "rel_x_embedding", "rel_y_embedding", "rel_z_embedding", "rel_w_embedding"],
}
PAIRWISE = ("transe", "transh", "transd", "rotate", "rescal", "ntn", "transm", "transr")
POINTWISE = ("distmult", "complex", "complexn3", "analogy", "cp", "simple", "simple_ignr", "quate")


def param_shapes(model, tot_entity, tot_relation, hidden_size=None, ent_hidden_size=None,
                 rel_hidden_size=None):
    """Table shapes as the reference constructors allocate them."""
    E, R, k = tot_entity, tot_relation, hidden_size
    if model in ("transe", "distmult", "transm"):
        return {"ent_embeddings": (E, k), "rel_embeddings": (R, k)}
    if model == "transr":
        return {"ent_embeddings": (E, ent_hidden_size), "rel_embeddings": (R, rel_hidden_size),
                "rel_matrix": (R, ent_hidden_size * rel_hidden_size)}
    if model == "cp":
        return {"sub_embeddings": (E, k), "rel_embeddings": (R, k), "obj_embeddings": (E, k)}
    if model in ("simple", "simple_ignr"):
        return {"ent_head_embeddings": (E, k), "ent_tail_embeddings": (E, k), "rel_embeddings": (R, k),
                "rel_inv_embeddings": (R, k)}
    if model == "quate":
        # the reference re-assigns the four rel_{s,x,y,z} tables from _quaternion_init(tot_entity, k), so they
        # carry tot_entity rows (pointwise.py:653-657); rel_w keeps [R, k] and is not used by forward
        out = {"ent_%s_embedding" % c: (E, k) for c in "sxyz"}
        out.update({"rel_%s_embedding" % c: (E, k) for c in "sxyz"})
        out["rel_w_embedding"] = (R, k)
        return out
    if model == "transh":
        return {"ent_embeddings": (E, k), "rel_embeddings": (R, k), "w": (R, k)}
    if model == "transd":
        return {"ent_embeddings": (E, ent_hidden_size), "rel_embeddings": (R, rel_hidden_size),
                "ent_mappings": (E, ent_hidden_size), "rel_mappings": (R, rel_hidden_size)}
    if model == "rotate":
        return {"ent_embeddings": (E, k), "ent_embeddings_imag": (E, k), "rel_embeddings": (R, k)}
    if model == "rescal":
        return {"ent_embeddings": (E, k), "rel_matrices": (R, k * k)}
    if model == "ntn":
        d, kr = ent_hidden_size, rel_hidden_size
        return {"ent_embeddings": (E, d), "rel_embeddings": (R, kr), "mr1": (d, kr), "mr2": (d, kr),
                "br": (1, kr), "mr": (kr, d * d)}
    if model in ("complex", "complexn3"):
        return {"ent_embeddings_real": (E, k), "ent_embeddings_img": (E, k),
                "rel_embeddings_real": (R, k), "rel_embeddings_img": (R, k)}
    if model == "analogy":
        return {"ent_embeddings": (E, k), "rel_embeddings": (R, k),
                "ent_embeddings_real": (E, k // 2), "ent_embeddings_img": (E, k // 2),
                "rel_embeddings_real": (R, k // 2), "rel_embeddings_img": (R, k // 2)}
    raise KeyError(model)


def init_params(model, rng, **shape_kw):
    """xavier_uniform_ (bound sqrt(6/(fan_in+fan_out)), pairwise.py:46-47) or, for RotatE,
    uniform(+-(margin+2)/hidden) (pairwise.py:748-755).  Deterministic given `rng`."""
    margin = shape_kw.pop("margin", None)
    shapes = param_shapes(model, **shape_kw)
    out = {}
    for name in PARAM_NAMES[model]:
        n, d = shapes[name]
        if model == "rotate":
            bound = (margin + 2.0) / shape_kw["hidden_size"]
        else:
            bound = np.sqrt(6.0 / (n + d))
        out[name] = rng.uniform(-bound, bound, size=(n, d)).astype(np.float32)
    return out


# --------------------------------------------------------------------------- helpers
def _norm_rows(x):
    return np.sqrt(np.sum(x * x, axis=-1, keepdims=True))


def _normalize(x):
    """F.normalize(x, p=2, dim=-1): x / max(||x||, eps).  Returns (x_hat, denom)."""
    den = np.maximum(_norm_rows(x), x.dtype.type(EPS_NORMALIZE))
    return x / den, den


def _normalize_bwd(x_hat, den, raw_norm_gt_eps, g):
    """Gradient of F.normalize wrt x given gradient g wrt x_hat."""
    inner = np.sum(x_hat * g, axis=-1, keepdims=True)
    return np.where(raw_norm_gt_eps, (g - x_hat * inner) / den, g / den)


def _cast(params, dtype):
    return {k: np.asarray(v, dtype=dtype) for k, v in params.items()}


def _idx(a):
    return np.asarray(a, dtype=np.int64)


# --------------------------------------------------------------------------- distance tail shared by TransE/H/D
def _trans_tail(a, b, c, l1_flag):
    """|| a^ + b^ - c^ ||_1 or _2 with a^ = F.normalize(a)  (pairwise.py:69-76,166-174,270-278)."""
    ah, na = _normalize(a)
    bh, nb = _normalize(b)
    ch, nc = _normalize(c)
    u = ah + bh - ch
    if l1_flag:
        s = np.sum(np.abs(u), axis=-1)
    else:
        s = np.sqrt(np.sum(u * u, axis=-1))
    return s, (ah, na, bh, nb, ch, nc, u)


def _trans_tail_bwd(a, b, c, saved, s, ds, l1_flag):
    ah, na, bh, nb, ch, nc, u = saved
    eps = a.dtype.type(EPS_NORMALIZE)
    if l1_flag:
        g = np.sign(u)
    else:
        safe = np.where(s > 0, s, 1).astype(a.dtype)
        g = np.where((s > 0)[:, None], u / safe[:, None], 0).astype(a.dtype)
    g = g * ds[:, None]
    ga = _normalize_bwd(ah, na, _norm_rows(a) > eps, g)
    gb = _normalize_bwd(bh, nb, _norm_rows(b) > eps, g)
    gc = -_normalize_bwd(ch, nc, _norm_rows(c) > eps, g)
    return ga, gb, gc


# --------------------------------------------------------------------------- forward scores
def score(model, params, h, r, t, dtype=np.float32, **hp):
    """Energy (lower = more plausible) of each (h[i], r[i], t[i]); mirrors `Model.forward`.

    hp: l1_flag (TransE/H/D), margin + hidden_size (RotatE).  RESCAL's in-place table
    renormalisation (pairwise.py:843-844) is NOT applied here -- call `rescal_normalize_tables`
    first, as the reference's forward does.
    """
    P = _cast(params, dtype)
    h, r, t = _idx(h), _idx(r), _idx(t)
    if model == "transe":  # pairwise.py:56-93
        s, _ = _trans_tail(P["ent_embeddings"][h], P["rel_embeddings"][r], P["ent_embeddings"][t], hp["l1_flag"])
        return s
    if model == "transh":  # pairwise.py:143-182
        what, _ = _normalize(P["w"][r])
        eh, et = P["ent_embeddings"][h], P["ent_embeddings"][t]
        a = eh - np.sum(eh * what, axis=-1, keepdims=True) * what
        c = et - np.sum(et * what, axis=-1, keepdims=True) * what
        s, _ = _trans_tail(a, P["rel_embeddings"][r], c, hp["l1_flag"])
        return s
    if model == "transd":  # pairwise.py:229-278
        eh, et = P["ent_embeddings"][h], P["ent_embeddings"][t]
        hm, tm, rm = P["ent_mappings"][h], P["ent_mappings"][t], P["rel_mappings"][r]
        a = eh + np.sum(eh * hm, axis=-1, keepdims=True) * rm
        c = et + np.sum(et * tm, axis=-1, keepdims=True) * rm
        s, _ = _trans_tail(a, P["rel_embeddings"][r], c, hp["l1_flag"])
        return s
    if model == "rotate":  # pairwise.py:765-791
        rng_ = dtype((hp["margin"] + 2.0) / hp["hidden_size"])
        phase = P["rel_embeddings"][r] / dtype(rng_ / dtype(PI))
        rr, ri = np.cos(phase), np.sin(phase)
        hr_, hi_ = P["ent_embeddings"][h], P["ent_embeddings_imag"][h]
        tr_, ti_ = P["ent_embeddings"][t], P["ent_embeddings_imag"][t]
        sr = hr_ * rr - hi_ * ri - tr_
        si = hr_ * ri + hi_ * rr - ti_
        return -(dtype(hp["margin"]) - np.sum(sr * sr + si * si, axis=-1))
    if model == "rescal":  # pairwise.py:829-860
        k = P["ent_embeddings"].shape[1]
        M = P["rel_matrices"][r].reshape(-1, k, k)
        Mt = np.einsum("nij,nj->ni", M, P["ent_embeddings"][t])
        return -np.sum(P["ent_embeddings"][h] * Mt, axis=-1)
    if model == "ntn":  # pairwise.py:919-960
        hh, _ = _normalize(P["ent_embeddings"][h])
        rh, _ = _normalize(P["rel_embeddings"][r])
        th, _ = _normalize(P["ent_embeddings"][t])
        return -np.sum(rh * _ntn_layer(P, hh, th), axis=-1)
    if model == "distmult":  # pointwise.py:444-446
        return -np.sum(P["ent_embeddings"][h] * P["rel_embeddings"][r] * P["ent_embeddings"][t], axis=-1)
    if model in ("complex", "complexn3"):  # pointwise.py:185-188
        return _complex_score(P["ent_embeddings_real"][h], P["ent_embeddings_img"][h],
                              P["rel_embeddings_real"][r], P["rel_embeddings_img"][r],
                              P["ent_embeddings_real"][t], P["ent_embeddings_img"][t])
    if model == "analogy":  # pointwise.py:97-104
        cs = _complex_score(P["ent_embeddings_real"][h], P["ent_embeddings_img"][h],
                            P["rel_embeddings_real"][r], P["rel_embeddings_img"][r],
                            P["ent_embeddings_real"][t], P["ent_embeddings_img"][t])
        dm = -np.sum(P["ent_embeddings"][h] * P["rel_embeddings"][r] * P["ent_embeddings"][t], axis=-1)
        return cs + dm
    if model == "transm":  # pairwise.py:325-347: theta_r * TransE distance; hp["theta"] = transm_theta(train, R)
        s, _ = _trans_tail(P["ent_embeddings"][h], P["rel_embeddings"][r], P["ent_embeddings"][t], hp["l1_flag"])
        return np.asarray(hp["theta"], dtype=dtype)[r] * s
    if model == "transr":  # pairwise.py:404-470: normalise, project by M_r = rel_matrix[r].view(d_e, d_r), TransE tail
        hh, _ = _normalize(P["ent_embeddings"][h])
        rh, _ = _normalize(P["rel_embeddings"][r])
        th, _ = _normalize(P["ent_embeddings"][t])
        M = P["rel_matrix"][r].reshape(len(r), hh.shape[1], rh.shape[1])
        s, _ = _trans_tail(np.einsum("na,nab->nb", hh, M), rh, np.einsum("na,nab->nb", th, M), hp["l1_flag"])
        return s
    if model == "cp":  # pointwise.py:374-376
        return -np.sum(P["sub_embeddings"][h] * P["rel_embeddings"][r] * P["obj_embeddings"][t], axis=-1)
    if model in ("simple", "simple_ignr"):  # pointwise.py:522-526, 581-585
        return -np.clip(_simple_init(model, P, h, r, t, dtype), dtype(-20), dtype(20))
    if model == "quate":  # pointwise.py:683-700
        H, T, Rn = _quate_rows(P, h, r, t)
        a, b, c, d = _hamilton(H, Rn)
        return -np.sum(a * T[0] + b * T[1] + c * T[2] + d * T[3], axis=-1)
    raise KeyError(model)


def transm_theta(train_triples, tot_relation):
    """TransM's fixed per-relation weight (pairwise.py:303-315).  rel_head / rel_tail are LISTS there (one entry per
    train triple, duplicates kept), so both lengths equal the relation's triple count c:
    theta_r = 1 / log(2 + c/(1+c) + c/(1+c)), computed in float64 then cast to float32."""
    cnt = np.bincount(np.asarray(train_triples)[:, 1], minlength=tot_relation).astype(np.float64)
    return (1.0 / np.log(2.0 + cnt / (1.0 + cnt) + cnt / (1.0 + cnt))).astype(np.float32)


def _simple_init(model, P, h, r, t, dtype):
    a = np.sum(P["ent_head_embeddings"][h] * P["rel_embeddings"][r] * P["ent_tail_embeddings"][t], axis=-1)
    b = np.sum(P["ent_head_embeddings"][t] * P["rel_inv_embeddings"][r] * P["ent_tail_embeddings"][h], axis=-1)
    # SimplE: operator precedence halves only the inverse term (pointwise.py:525); SimplE_ignr sums the
    # concatenated [k | k] vectors, i.e. no halving (pointwise.py:584)
    return a + b / dtype(2.0) if model == "simple" else a + b


def _quate_rows(P, h, r, t):
    H = [P["ent_%s_embedding" % c][h] for c in "sxyz"]
    T = [P["ent_%s_embedding" % c][t] for c in "sxyz"]
    Rr = [P["rel_%s_embedding" % c][r] for c in "sxyz"]
    den = np.sqrt(Rr[0] ** 2 + Rr[1] ** 2 + Rr[2] ** 2 + Rr[3] ** 2)
    return H, T, [x / den for x in Rr]


def _hamilton(H, Rn):
    hs, hx, hy, hz = H
    ps, px, py, pz = Rn
    return (hs * ps - hx * px - hy * py - hz * pz, hs * px + ps * hx + hy * pz - py * hz,
            hs * py + ps * hy + hz * px - pz * hx, hs * pz + ps * hz + hx * py - px * hy)


def _complex_score(hr_, hi_, rr, ri, tr_, ti_):
    return -np.sum(hr_ * tr_ * rr + hi_ * ti_ * rr + hr_ * ti_ * ri - hi_ * tr_ * ri, axis=-1)


def _ntn_layer(P, hh, th):
    """NTN.train_layer (pairwise.py:919-936): tanh(h^T W_k t + h M1 + t M2 + b), per slice k."""
    d = hh.shape[1]
    kr = P["mr1"].shape[1]
    W = P["mr"].reshape(kr, d, d)
    bil = np.einsum("ni,kij,nj->nk", hh, W, th)
    return np.tanh(bil + hh @ P["mr1"] + th @ P["mr2"] + P["br"])


def rescal_normalize_tables(params, dtype=np.float32):
    """Rescal.embed side effect (pairwise.py:843-844,862-865): both tables are overwritten by their
    row-L2-normalised versions (plain division, no eps) on EVERY forward, training and eval."""
    out = dict(params)
    for name in ("ent_embeddings", "rel_matrices"):
        w = np.asarray(params[name], dtype=dtype)
        out[name] = (w / _norm_rows(w)).astype(dtype)
    return out


# --------------------------------------------------------------------------- dense gradients of sum(ds * score)
def score_grad(model, params, h, r, t, ds, dtype=np.float32, **hp):
    """Dense [num, dim] gradients (nn.Embedding sparse=False semantics, Domain.py:8-13) of
    sum_i ds[i]*score_i wrt every table.  Hand-derived; equals the reference's autograd result."""
    P = _cast(params, dtype)
    h, r, t = _idx(h), _idx(r), _idx(t)
    ds = np.asarray(ds, dtype=dtype)
    G = {k: np.zeros_like(v) for k, v in P.items()}

    def add(name, idx, val):
        np.add.at(G[name], idx, val.astype(dtype))

    if model == "transe":
        a, b, c = P["ent_embeddings"][h], P["rel_embeddings"][r], P["ent_embeddings"][t]
        s, saved = _trans_tail(a, b, c, hp["l1_flag"])
        ga, gb, gc = _trans_tail_bwd(a, b, c, saved, s, ds, hp["l1_flag"])
        add("ent_embeddings", h, ga); add("rel_embeddings", r, gb); add("ent_embeddings", t, gc)
    elif model == "transh":
        w = P["w"][r]
        what, nw = _normalize(w)
        eh, et, b = P["ent_embeddings"][h], P["ent_embeddings"][t], P["rel_embeddings"][r]
        ph = np.sum(eh * what, axis=-1, keepdims=True)
        pt = np.sum(et * what, axis=-1, keepdims=True)
        a, c = eh - ph * what, et - pt * what
        s, saved = _trans_tail(a, b, c, hp["l1_flag"])
        ga, gb, gc = _trans_tail_bwd(a, b, c, saved, s, ds, hp["l1_flag"])
        gaw = np.sum(ga * what, axis=-1, keepdims=True)
        gcw = np.sum(gc * what, axis=-1, keepdims=True)
        add("ent_embeddings", h, ga - gaw * what)
        add("ent_embeddings", t, gc - gcw * what)
        add("rel_embeddings", r, gb)
        gwhat = -(ph * ga + gaw * eh) - (pt * gc + gcw * et)
        add("w", r, _normalize_bwd(what, nw, _norm_rows(w) > dtype(EPS_NORMALIZE), gwhat))
    elif model == "transd":
        eh, et, b = P["ent_embeddings"][h], P["ent_embeddings"][t], P["rel_embeddings"][r]
        hm, tm, rm = P["ent_mappings"][h], P["ent_mappings"][t], P["rel_mappings"][r]
        ph = np.sum(eh * hm, axis=-1, keepdims=True)
        pt = np.sum(et * tm, axis=-1, keepdims=True)
        a, c = eh + ph * rm, et + pt * rm
        s, saved = _trans_tail(a, b, c, hp["l1_flag"])
        ga, gb, gc = _trans_tail_bwd(a, b, c, saved, s, ds, hp["l1_flag"])
        gar = np.sum(ga * rm, axis=-1, keepdims=True)
        gcr = np.sum(gc * rm, axis=-1, keepdims=True)
        add("ent_embeddings", h, ga + gar * hm); add("ent_mappings", h, gar * eh)
        add("ent_embeddings", t, gc + gcr * tm); add("ent_mappings", t, gcr * et)
        add("rel_embeddings", r, gb); add("rel_mappings", r, ph * ga + pt * gc)
    elif model == "rotate":
        rng_ = dtype((hp["margin"] + 2.0) / hp["hidden_size"])
        scale = dtype(rng_ / dtype(PI))
        phase = P["rel_embeddings"][r] / scale
        rr, ri = np.cos(phase), np.sin(phase)
        hr_, hi_ = P["ent_embeddings"][h], P["ent_embeddings_imag"][h]
        tr_, ti_ = P["ent_embeddings"][t], P["ent_embeddings_imag"][t]
        sr = (hr_ * rr - hi_ * ri - tr_) * (2 * ds[:, None])
        si = (hr_ * ri + hi_ * rr - ti_) * (2 * ds[:, None])
        add("ent_embeddings", h, sr * rr + si * ri)
        add("ent_embeddings_imag", h, -sr * ri + si * rr)
        add("ent_embeddings", t, -sr); add("ent_embeddings_imag", t, -si)
        gphase = sr * (-hr_ * ri - hi_ * rr) + si * (hr_ * rr - hi_ * ri)
        add("rel_embeddings", r, gphase / scale)
    elif model == "rescal":
        k = P["ent_embeddings"].shape[1]
        eh, et = P["ent_embeddings"][h], P["ent_embeddings"][t]
        M = P["rel_matrices"][r].reshape(-1, k, k)
        nds = -ds[:, None]
        add("ent_embeddings", h, np.einsum("nij,nj->ni", M, et) * nds)
        add("ent_embeddings", t, np.einsum("nij,ni->nj", M, eh) * nds)
        add("rel_matrices", r, (eh[:, :, None] * et[:, None, :] * nds[:, :, None]).reshape(-1, k * k))
    elif model == "ntn":
        eh, er, et = P["ent_embeddings"][h], P["rel_embeddings"][r], P["ent_embeddings"][t]
        hh, nh = _normalize(eh); rh, nr = _normalize(er); th, nt = _normalize(et)
        d, kr = hh.shape[1], P["mr1"].shape[1]
        W = P["mr"].reshape(kr, d, d)
        z = _ntn_layer(P, hh, th)
        gz = -(rh * ds[:, None]) * (1 - z * z)          # grad wrt pre-tanh
        grh = -(z * ds[:, None])
        ghh = np.einsum("nk,kij,nj->ni", gz, W, th) + gz @ P["mr1"].T
        gth = np.einsum("nk,kij,ni->nj", gz, W, hh) + gz @ P["mr2"].T
        G["mr"] += np.einsum("nk,ni,nj->kij", gz, hh, th).reshape(kr, d * d).astype(dtype)
        G["mr1"] += (hh.T @ gz).astype(dtype); G["mr2"] += (th.T @ gz).astype(dtype)
        G["br"] += np.sum(gz, axis=0, keepdims=True).astype(dtype)
        eps = dtype(EPS_NORMALIZE)
        add("ent_embeddings", h, _normalize_bwd(hh, nh, _norm_rows(eh) > eps, ghh))
        add("ent_embeddings", t, _normalize_bwd(th, nt, _norm_rows(et) > eps, gth))
        add("rel_embeddings", r, _normalize_bwd(rh, nr, _norm_rows(er) > eps, grh))
    elif model == "distmult":
        eh, er, et = P["ent_embeddings"][h], P["rel_embeddings"][r], P["ent_embeddings"][t]
        nds = -ds[:, None]
        add("ent_embeddings", h, er * et * nds); add("rel_embeddings", r, eh * et * nds)
        add("ent_embeddings", t, eh * er * nds)
    elif model in ("complex", "complexn3", "analogy"):
        hr_, hi_ = P["ent_embeddings_real"][h], P["ent_embeddings_img"][h]
        rr, ri = P["rel_embeddings_real"][r], P["rel_embeddings_img"][r]
        tr_, ti_ = P["ent_embeddings_real"][t], P["ent_embeddings_img"][t]
        nds = -ds[:, None]
        add("ent_embeddings_real", h, (tr_ * rr + ti_ * ri) * nds)
        add("ent_embeddings_img", h, (ti_ * rr - tr_ * ri) * nds)
        add("rel_embeddings_real", r, (hr_ * tr_ + hi_ * ti_) * nds)
        add("rel_embeddings_img", r, (hr_ * ti_ - hi_ * tr_) * nds)
        add("ent_embeddings_real", t, (hr_ * rr - hi_ * ri) * nds)
        add("ent_embeddings_img", t, (hi_ * rr + hr_ * ri) * nds)
        if model == "analogy":
            eh, er, et = P["ent_embeddings"][h], P["rel_embeddings"][r], P["ent_embeddings"][t]
            add("ent_embeddings", h, er * et * nds); add("rel_embeddings", r, eh * et * nds)
            add("ent_embeddings", t, eh * er * nds)
    elif model == "transm":
        a, b, c = P["ent_embeddings"][h], P["rel_embeddings"][r], P["ent_embeddings"][t]
        s, saved = _trans_tail(a, b, c, hp["l1_flag"])
        ga, gb, gc = _trans_tail_bwd(a, b, c, saved, s, ds * np.asarray(hp["theta"], dtype=dtype)[r], hp["l1_flag"])
        add("ent_embeddings", h, ga); add("rel_embeddings", r, gb); add("ent_embeddings", t, gc)
    elif model == "transr":
        eh, er, et = P["ent_embeddings"][h], P["rel_embeddings"][r], P["ent_embeddings"][t]
        hh, nh = _normalize(eh); rh, nr = _normalize(er); th, nt = _normalize(et)
        de, dr = hh.shape[1], rh.shape[1]
        M = P["rel_matrix"][r].reshape(len(r), de, dr)
        a, c = np.einsum("na,nab->nb", hh, M), np.einsum("na,nab->nb", th, M)
        s, saved = _trans_tail(a, rh, c, hp["l1_flag"])
        ga, gb, gc = _trans_tail_bwd(a, rh, c, saved, s, ds, hp["l1_flag"])
        gM = np.einsum("na,nb->nab", hh, ga) + np.einsum("na,nb->nab", th, gc)
        add("rel_matrix", r, gM.reshape(len(r), de * dr))
        eps = dtype(EPS_NORMALIZE)
        add("ent_embeddings", h, _normalize_bwd(hh, nh, _norm_rows(eh) > eps, np.einsum("nab,nb->na", M, ga)))
        add("ent_embeddings", t, _normalize_bwd(th, nt, _norm_rows(et) > eps, np.einsum("nab,nb->na", M, gc)))
        add("rel_embeddings", r, _normalize_bwd(rh, nr, _norm_rows(er) > eps, gb))
    elif model == "cp":
        eh, er, et = P["sub_embeddings"][h], P["rel_embeddings"][r], P["obj_embeddings"][t]
        nds = -ds[:, None]
        add("sub_embeddings", h, er * et * nds); add("rel_embeddings", r, eh * et * nds)
        add("obj_embeddings", t, eh * er * nds)
    elif model in ("simple", "simple_ignr"):
        init = _simple_init(model, P, h, r, t, dtype)
        inside = (init >= -20) & (init <= 20)  # torch.clamp passes the gradient on the closed interval
        g1 = np.where(inside, -ds, 0).astype(dtype)[:, None]
        g2 = g1 / dtype(2.0) if model == "simple" else g1
        h1, h2 = P["ent_head_embeddings"][h], P["ent_head_embeddings"][t]
        r1, r2 = P["rel_embeddings"][r], P["rel_inv_embeddings"][r]
        t1, t2 = P["ent_tail_embeddings"][t], P["ent_tail_embeddings"][h]
        add("ent_head_embeddings", h, r1 * t1 * g1); add("rel_embeddings", r, h1 * t1 * g1)
        add("ent_tail_embeddings", t, h1 * r1 * g1)
        add("ent_head_embeddings", t, r2 * t2 * g2); add("rel_inv_embeddings", r, h2 * t2 * g2)
        add("ent_tail_embeddings", h, h2 * r2 * g2)
    elif model == "quate":
        H, T, Rn = _quate_rows(P, h, r, t)
        hs, hx, hy, hz = H
        ts, tx, ty, tz = T
        ps, px, py, pz = Rn
        g = -ds[:, None]
        for c, v in zip("sxyz", _hamilton(H, Rn)):
            add("ent_%s_embedding" % c, t, v * g)
        add("ent_s_embedding", h, (ps * ts + px * tx + py * ty + pz * tz) * g)
        add("ent_x_embedding", h, (-px * ts + ps * tx - pz * ty + py * tz) * g)
        add("ent_y_embedding", h, (-py * ts + pz * tx + ps * ty - px * tz) * g)
        add("ent_z_embedding", h, (-pz * ts - py * tx + px * ty + ps * tz) * g)
        gp = [(hs * ts + hx * tx + hy * ty + hz * tz) * g, (-hx * ts + hs * tx + hz * ty - hy * tz) * g,
              (-hy * ts - hz * tx + hs * ty + hx * tz) * g, (-hz * ts + hy * tx - hx * ty + hs * tz) * g]
        raw = [P["rel_%s_embedding" % c][r] for c in "sxyz"]
        den = np.sqrt(raw[0] ** 2 + raw[1] ** 2 + raw[2] ** 2 + raw[3] ** 2)
        dot = ps * gp[0] + px * gp[1] + py * gp[2] + pz * gp[3]
        for c, pc, gc_ in zip("sxyz", Rn, gp):
            add("rel_%s_embedding" % c, r, (gc_ - pc * dot) / den)
    else:
        raise KeyError(model)
    return G


# --------------------------------------------------------------------------- losses (utils/criterion.py)
def _logsigmoid(x):
    return -np.logaddexp(0, -x).astype(x.dtype)


def _sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x))).astype(x.dtype)


def pairwise_hinge(pos, neg, margin):
    """criterion.py:25-29: sum(max(pos + margin - neg, 0)).  Returns (loss, dL/dpos, dL/dneg).
    torch.max(a, zeros) sends the whole gradient to `a` only where a > 0 (at a == 0 it is split;
    we take 0.5 there like ATen's max backward)."""
    v = pos + pos.dtype.type(margin) - neg
    loss = np.sum(np.maximum(v, 0), dtype=pos.dtype)
    m = np.where(v > 0, 1.0, np.where(v == 0, 0.5, 0.0)).astype(pos.dtype)
    return loss, m, -m


def pairwise_logistic_selfadv(pos, neg, neg_rate, alpha):
    """criterion.py:13-23 (`pariwise_logistic`, RotatE): self-adversarial negative sampling loss with
    DETACHED softmax weights.  neg is [B*neg_rate] with the negatives of positive i contiguous.
    Returns (loss, dL/dpos, dL/dneg)."""
    dt = pos.dtype
    B = pos.shape[0]
    p = -pos
    n = (-neg).reshape(-1, neg_rate)
    z = n * dt.type(alpha)
    z = z - np.max(z, axis=1, keepdims=True)
    w = np.exp(z); w = (w / np.sum(w, axis=1, keepdims=True)).astype(dt)
    neg_term = np.sum(w * _logsigmoid(-n), axis=-1)
    loss = -np.mean(neg_term, dtype=dt) - np.mean(_logsigmoid(p), dtype=dt)
    dpos = (_sigmoid(-p) / dt.type(B)).astype(dt)                  # d/dpos of -mean(logsig(-pos))
    dneg = (-(w * _sigmoid(n)) / dt.type(B)).reshape(-1).astype(dt)  # d/dneg of -mean(sum w*logsig(neg))
    return dt.type(loss), dpos, dneg


def pointwise_logistic(preds, target):
    """criterion.py:31-34: mean(softplus(target*preds)).  Returns (loss, dL/dpreds)."""
    dt = preds.dtype
    y = target.astype(dt)
    x = y * preds
    loss = np.mean(np.logaddexp(0, x), dtype=dt)
    return dt.type(loss), (y * _sigmoid(x) / dt.type(preds.shape[0])).astype(dt)


def pointwise_reg(model, params, h, r, t, lmbda, reg_type=None, dtype=np.float32):
    """get_reg of DistMult (pointwise.py:448-458), Complex (:190-202), ComplexN3 (:224-238),
    ANALOGY (:106-119): lmbda * mean_i(sum of squares (F2) or cubes (N3; |x|^3 for ComplexN3))
    over the rows gathered for the batch.  Returns (reg, dense grads)."""
    P = _cast(params, dtype)
    h, r, t = _idx(h), _idx(r), _idx(t)
    if reg_type is None:
        reg_type = "N3" if model in ("complexn3", "cp", "quate") else "F2"
    reg_type = reg_type.lower()
    use_abs = model in ("complexn3", "quate")
    N = dtype(h.shape[0])
    G = {k: np.zeros_like(v) for k, v in P.items()}
    total = np.zeros(h.shape[0], dtype=dtype)
    if model in ("simple", "simple_ignr"):
        # SimplE.get_reg (pointwise.py:528-536) is handed the ID tensors and regularises THEM:
        # lmbda * (sum(h^2) + sum(r^2) + sum(t^2)) in float32 ("mean" of a 0-d tensor) -- a constant, no gradient
        p = 2 if reg_type == "f2" else 3
        tot = sum(np.sum(np.asarray(x, dtype=np.float32) ** p, dtype=np.float32) for x in (h, r, t))
        return dtype(dtype(lmbda) * dtype(tot)), G
    if model == "quate":
        # QuatE.get_reg (pointwise.py:702-736): lmbda * sum over the 12 gathered rows-sets of mean over ALL B*k
        # elements of |x|^p
        p = 2 if reg_type == "f2" else 3
        lam, cnt, reg = dtype(lmbda), dtype(h.shape[0] * P["ent_s_embedding"].shape[1]), dtype(0)
        for names, idx in (((("ent_%s_embedding" % c) for c in "sxyz"), h), ((("ent_%s_embedding" % c) for c in "sxyz"), t),
                           ((("rel_%s_embedding" % c) for c in "sxyz"), r)):
            for name in names:
                x = P[name][idx]
                reg += np.mean(np.abs(x) ** p, dtype=dtype)
                np.add.at(G[name], idx, (p * lam / cnt) * np.sign(x) * np.abs(x) ** (p - 1))
        return dtype(lam * reg), G
    if model == "cp":
        rows = [("sub_embeddings", h), ("rel_embeddings", r), ("obj_embeddings", t)]
    elif model == "distmult":
        rows = [("ent_embeddings", h), ("rel_embeddings", r), ("ent_embeddings", t)]
    elif model in ("complex", "complexn3"):
        rows = [("ent_embeddings_real", h), ("ent_embeddings_img", h), ("rel_embeddings_real", r),
                ("rel_embeddings_img", r), ("ent_embeddings_real", t), ("ent_embeddings_img", t)]
    elif model == "analogy":
        rows = [("ent_embeddings_real", h), ("ent_embeddings_img", h), ("rel_embeddings_real", r),
                ("rel_embeddings_img", r), ("ent_embeddings_real", t), ("ent_embeddings_img", t),
                ("ent_embeddings", h), ("rel_embeddings", r), ("ent_embeddings", t)]
    else:
        raise KeyError(model)
    lam = dtype(lmbda)
    for name, idx in rows:
        x = P[name][idx]
        if reg_type == "f2":
            total += np.sum(x * x, axis=-1)
            np.add.at(G[name], idx, (2 * lam / N) * x)
        elif reg_type == "n3":
            if use_abs:
                total += np.sum(np.abs(x) ** 3, axis=-1)
                np.add.at(G[name], idx, (3 * lam / N) * x * np.abs(x))
            else:
                total += np.sum(x ** 3, axis=-1)
                np.add.at(G[name], idx, (3 * lam / N) * x * x)
        else:
            raise NotImplementedError(reg_type)
    if model == "analogy":
        # reference sums two separate means (pointwise.py:110-111); identical to one mean of the total
        pass
    return dtype(lam * np.mean(total, dtype=dtype)), G


def ntn_reg(params, lmbda, dtype=np.float32):
    """NTN.get_reg (pairwise.py:962-963): lmbda*sqrt(sum over all tables of sum(w^2)).  (reg, grads)."""
    P = _cast(params, dtype)
    tot = dtype(0)
    for v in P.values():
        tot += np.sum(v * v, dtype=dtype)
    root = np.sqrt(tot)
    return dtype(lmbda) * root, {k: (dtype(lmbda) * v / root).astype(dtype) for k, v in P.items()}


# --------------------------------------------------------------------------- 1-N scoring head (projection models)
def head_1n_forward(x, ent, bias=None, dtype=np.float32):
    """Last lines of ConvE / TuckER / InteractE / HypER / AcrE forward (projection.py:100-102, 335-336, 444-447, 606-609,
    734-737): sigmoid(x @ ent.T + bias), [B, E]."""
    z = np.asarray(x, dtype) @ np.asarray(ent, dtype).T
    if bias is not None:
        z = z + np.asarray(bias, dtype).reshape(1, -1)
    return _sigmoid(z)


def bf16_round(a):
    """fp32 -> bfloat16 (round to nearest even) -> fp32, as the bf16 option of the head rounds its operands."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return r.view(np.float32)


def head_1n_forward_bf16(x, ent, bias=None):
    """The bf16 option of the head (kge_head_1n_forward_bf16): operands rounded to bfloat16, exact products, wide accumulation
    (float64 here: the kernel accumulates in fp32 in the matrix core's own order), bias and sigmoid in fp32."""
    z = (bf16_round(x).astype(np.float64) @ bf16_round(ent).astype(np.float64).T).astype(np.float32)
    if bias is not None:
        z = z + np.asarray(bias, np.float32).reshape(1, -1)
    return _sigmoid(z)


def multi_class_bce_dir(preds, labels, label_smoothing, tot_entity, dtype=np.float32):
    """One direction of Criterion.multi_class_bce (criterion.py:41-49): optional label smoothing
    y <- y (1 - ls) + 1/E, then torch.nn.BCEWithLogitsLoss (mean over all B*E elements) applied to the sigmoid OUTPUTS
    as if they were logits.  Returns (loss, d loss / d preds)."""
    p = np.asarray(preds, dtype)
    y = np.asarray(labels, dtype)
    if label_smoothing is not None and tot_entity is not None:
        y = y * dtype(1.0 - label_smoothing) + dtype(1.0 / tot_entity)
    elem = np.maximum(p, 0) - p * y + np.log1p(np.exp(-np.abs(p)))
    return dtype(np.mean(elem, dtype=dtype)), ((_sigmoid(p) - y) / dtype(p.size)).astype(dtype)


def head_1n_backward(x, ent, preds, dpreds, dtype=np.float32):
    """(dx, d ent, d bias) of head_1n_forward given d loss / d preds."""
    p = np.asarray(preds, dtype)
    dz = np.asarray(dpreds, dtype) * p * (1 - p)
    return dz @ np.asarray(ent, dtype), dz.T @ np.asarray(x, dtype), dz.sum(axis=0)


# --------------------------------------------------------------------------- one training step
def train_step_grads(model, params, batch, dtype=np.float32, **hp):
    """Loss and dense gradients of one reference train step.

    pairwise  (trainer.py:147-157): batch = (ph, pr, pt, nh, nr, nt); hinge(margin) or, when
              model == 'rotate', the self-adversarial logistic loss(neg_rate, alpha); + get_reg(None..).
    pointwise (trainer.py:176-180): batch = (h, r, t, y); pointwise_logistic + get_reg(h, r, t).
    """
    P = params
    if model in PAIRWISE:
        ph, pr, pt, nh, nr, nt = batch
        if model == "rescal":
            P = rescal_normalize_tables(P, dtype)
        pos = score(model, P, ph, pr, pt, dtype=dtype, **hp)
        if model == "rescal":
            P = rescal_normalize_tables(P, dtype)  # second forward renormalises again (idempotent up to rounding)
        neg = score(model, P, nh, nr, nt, dtype=dtype, **hp)
        if model == "rotate":
            loss, dpos, dneg = pairwise_logistic_selfadv(pos, neg, hp["neg_rate"], hp["alpha"])
        else:
            loss, dpos, dneg = pairwise_hinge(pos, neg, hp["margin"])
        Gp = score_grad(model, P, ph, pr, pt, dpos, dtype=dtype, **hp)
        Gn = score_grad(model, P, nh, nr, nt, dneg, dtype=dtype, **hp)
        G = {k: Gp[k] + Gn[k] for k in Gp}
        if model == "ntn":
            reg, Gr = ntn_reg(P, hp["lmbda"], dtype)
            loss = loss + reg
            G = {k: G[k] + Gr[k] for k in G}
        return dtype(loss), G, (pos, neg), P
    h, r, t, y = batch
    preds = score(model, P, h, r, t, dtype=dtype, **hp)
    loss, dpred = pointwise_logistic(preds, np.asarray(y))
    G = score_grad(model, P, h, r, t, dpred, dtype=dtype, **hp)
    reg, Gr = pointwise_reg(model, P, h, r, t, hp["lmbda"], hp.get("reg_type"), dtype)
    G = {k: G[k] + Gr[k] for k in G}
    return dtype(loss + reg), G, (preds,), P


# --------------------------------------------------------------------------- dense optimisers (trainer.py:112-131)
def optimizer_init(name, params):
    st = {"step": 0}
    if name == "adam":
        st["m"] = {k: np.zeros_like(v) for k, v in params.items()}
        st["v"] = {k: np.zeros_like(v) for k, v in params.items()}
    elif name in ("adagrad", "rms"):
        st["sq"] = {k: np.zeros_like(v) for k, v in params.items()}
    elif name != "sgd":
        raise NotImplementedError(name)
    return st


def optimizer_step(name, params, grads, state, lr):
    """torch.optim.{SGD,Adam,Adagrad,RMSprop}(params, lr=lr) with every other argument at its
    PyTorch default, dense (all rows every step).  Third-party arithmetic (PyTorch, not vendored):
    SGD p -= lr*g; Adam betas (0.9,0.999) eps 1e-8 bias-corrected; Adagrad eps 1e-10, lr_decay 0;
    RMSprop alpha 0.99 eps 1e-8, momentum 0, not centered.  In place on `params` and `state`."""
    state["step"] += 1
    n = state["step"]
    for k, p in params.items():
        g = grads[k]
        f = p.dtype.type
        if name == "sgd":
            p -= f(lr) * g
        elif name == "adam":
            b1, b2, eps = 0.9, 0.999, 1e-8
            m, v = state["m"][k], state["v"][k]
            m += f(1 - b1) * (g - m)
            v *= f(b2); v += f(1 - b2) * g * g
            step_size = lr / (1 - b1 ** n)
            bc2_sqrt = np.sqrt(1 - b2 ** n)
            denom = np.sqrt(v) / f(bc2_sqrt) + f(eps)
            p += f(-step_size) * m / denom
        elif name == "adagrad":
            sq = state["sq"][k]
            sq += g * g
            p -= f(lr) * g / (np.sqrt(sq) + f(1e-10))
        elif name == "rms":
            sq = state["sq"][k]
            sq *= f(0.99); sq += f(1 - 0.99) * g * g
            p -= f(lr) * g / (np.sqrt(sq) + f(1e-8))
        else:
            raise NotImplementedError(name)


# --------------------------------------------------------------------------- negative corruption (data/generator.py)
def bern_probability(train_triples, tot_relation):
    """KnowledgeGraph.read_relation_property (data/kgcontroller.py:466-492):
    prob[r] = |unique tails of r| / (|unique heads of r| + |unique tails of r|) over train."""
    tr = np.asarray(train_triples, dtype=np.int64)
    prob = np.zeros(tot_relation, dtype=np.float64)
    for rel in range(tot_relation):
        m = tr[:, 1] == rel
        if not m.any():
            continue
        nh_, nt_ = len(np.unique(tr[m, 0])), len(np.unique(tr[m, 2]))
        prob[rel] = nt_ / (nh_ + nt_)
    return prob


def corrupt_batch(pos_triples, train_set, tot_entity, neg_rate, prob_of_rel, rng):
    """process_function_pairwise (data/generator.py:71-95): for every positive and every one of its
    neg_rate slots draw u~U[0,1): u > prob -> replace tail else replace head, redrawing the entity
    while the corrupted triple is a TRAIN triple.  prob = relation_property[r] (bern) or 0.5.
    Returns int64 arrays nh, nr, nt of length B*neg_rate, negatives of positive i contiguous."""
    nh, nr, nt = [], [], []
    for (h, r, t) in np.asarray(pos_triples, dtype=np.int64):
        prob = prob_of_rel[r] if prob_of_rel is not None else 0.5
        for _ in range(neg_rate):
            if rng.random() > prob:
                e = int(rng.integers(tot_entity))
                while (h, r, e) in train_set:
                    e = int(rng.integers(tot_entity))
                nh.append(h); nr.append(r); nt.append(e)
            else:
                e = int(rng.integers(tot_entity))
                while (e, r, t) in train_set:
                    e = int(rng.integers(tot_entity))
                nh.append(e); nr.append(r); nt.append(t)
    return np.asarray(nh, np.int64), np.asarray(nr, np.int64), np.asarray(nt, np.int64)


def pointwise_layout(pos_triples, nh, nr, nt, neg_rate):
    """process_function_pointwise (data/generator.py:125-156): rows [pos_i, its neg_rate negatives]
    per positive with labels +1 / -1."""
    pos = np.asarray(pos_triples, dtype=np.int64)
    B = pos.shape[0]
    H = np.concatenate([pos[:, 0:1], nh.reshape(B, neg_rate)], axis=1).reshape(-1)
    R = np.concatenate([pos[:, 1:2], nr.reshape(B, neg_rate)], axis=1).reshape(-1)
    T = np.concatenate([pos[:, 2:3], nt.reshape(B, neg_rate)], axis=1).reshape(-1)
    Y = np.tile(np.array([1] + [-1] * neg_rate, dtype=np.int64), B)
    return H, R, T, Y


# --------------------------------------------------------------------------- evaluation (utils/evaluator.py)
def sweep_scores(model, params, h, r, t, side, dtype=np.float32, **hp):
    """Evaluator.test_tail_rank / test_head_rank (evaluator.py:249-273): scores of (h, r, e) for every
    entity e (side='tail') or (e, r, t) (side='head')."""
    E = next(iter(params.values())).shape[0] if model != "ntn" else params["ent_embeddings"].shape[0]
    E = params[PARAM_NAMES[model][0]].shape[0]
    ents = np.arange(E, dtype=np.int64)
    if side == "tail":
        return score(model, params, np.full(E, h), np.full(E, r), ents, dtype=dtype, **hp)
    return score(model, params, ents, np.full(E, r), np.full(E, t), dtype=dtype, **hp)


def rank_from_ordering(ordering_desc, true_id, known):
    """MetricCalculator.get_tail_rank / get_head_rank (evaluator.py:70-123).  `ordering_desc` is
    torch.topk(preds, k=E)'s index output (descending energy); the scan runs from its END (lowest
    energy first) to the true id.  rank = candidates met before it; filtered rank additionally
    skips those in `known` (hr_t[(h,r)] or tr_h[(t,r)], = train+valid+test, kgcontroller.py:410-428)."""
    rank = frank = 0
    for j in range(len(ordering_desc)):
        val = int(ordering_desc[-j - 1])
        if val == true_id:
            break
        rank += 1
        frank += 1
        if val in known:
            frank -= 1
    return rank, frank


def rank_from_scores(scores, true_id, known):
    """Sort-free statement of the same ranks: rank = #{e : s_e < s_true};
    filtered = rank - #{e in known, e != true : s_e < s_true}.  Equal to `rank_from_ordering`
    whenever no other candidate ties the true one exactly (topk's tie order is unspecified)."""
    st = scores[true_id]
    rank = int(np.sum(scores < st))
    kn = np.fromiter((e for e in known if e != true_id), dtype=np.int64)
    frank = rank - (int(np.sum(scores[kn] < st)) if kn.size else 0)
    return rank, frank


def settle(rank_head, rank_tail, f_rank_head, f_rank_tail, hits=(1, 3, 5, 10)):
    """MetricCalculator.settle (evaluator.py:125-141): ranks+1 as float32; head and tail concatenated."""
    ranks = np.concatenate((np.asarray(rank_head, np.float32) + 1, np.asarray(rank_tail, np.float32) + 1))
    franks = np.concatenate((np.asarray(f_rank_head, np.float32) + 1, np.asarray(f_rank_tail, np.float32) + 1))
    out = {"mr": np.mean(ranks), "mrr": np.mean(np.reciprocal(ranks)),
           "fmr": np.mean(franks), "fmrr": np.mean(np.reciprocal(franks))}
    for k in hits:
        out["hit%d" % k] = np.mean(ranks <= k, dtype=np.float32)
        out["fhit%d" % k] = np.mean(franks <= k, dtype=np.float32)
    return out


def build_filters(all_triples):
    """hr_t / tr_h dict-of-sets over train+valid+test (kgcontroller.py:410-428)."""
    hr_t, tr_h = {}, {}
    for h, r, t in np.asarray(all_triples, dtype=np.int64):
        hr_t.setdefault((int(h), int(r)), set()).add(int(t))
        tr_h.setdefault((int(t), int(r)), set()).add(int(h))
    return hr_t, tr_h


def evaluate(model, params, test_triples, hr_t, tr_h, dtype=np.float32, hits=(1, 3, 5, 10), **hp):
    """Evaluator.test loop (evaluator.py:309-334) with count-based ranks.  Returns (metrics, ranks)."""
    if model == "rescal":
        params = rescal_normalize_tables(params, dtype)
    rh, rt, frh, frt = [], [], [], []
    for h, r, t in np.asarray(test_triples, dtype=np.int64):
        h, r, t = int(h), int(r), int(t)
        sh = sweep_scores(model, params, h, r, t, "head", dtype=dtype, **hp)
        st = sweep_scores(model, params, h, r, t, "tail", dtype=dtype, **hp)
        a, b = rank_from_scores(sh, h, tr_h.get((t, r), ()))
        c, d = rank_from_scores(st, t, hr_t.get((h, r), ()))
        rh.append(a); frh.append(b); rt.append(c); frt.append(d)
    ranks = {"head": np.asarray(rh, np.int64), "tail": np.asarray(rt, np.int64),
             "fhead": np.asarray(frh, np.int64), "ftail": np.asarray(frt, np.int64)}
    return settle(rh, rt, frh, frt, hits), ranks
