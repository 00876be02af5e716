"""Dev tool: randomised shapes through the 1-N head (both tile shapes forced in turn) against the numpy oracle.
ITERS (default 40), SEED."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import kge_oracle as ko
from pykg2vec_amd import kernels as K

rng = np.random.default_rng(int(os.environ.get("SEED", 1)))
bad = 0
for it in range(int(os.environ.get("ITERS", 40))):
    B = int(rng.choice([1, 3, 64, 127, 128, 129, 300, 513, 700]))
    E = int(rng.choice([1, 2, 3, 4, 5, 63, 64, 65, 127, 129, 257, 1000, 1001, 1002, 1003, 2500]))
    d = int(rng.choice([4, 8, 12, 64, 100, 128, 132, 200, 260])) if rng.random() < 0.85 else int(rng.choice([1, 7, 33]))
    if it < 4:      # the corner the wide backward products must refuse: fewer than four elements in the whole [B, E] array
        B, E, d = 1, it % 3 + 1, 4
    with_bias = bool(rng.random() < 0.7)
    dens = float(rng.choice([0.0, 0.01, 0.2]))
    ls = None if rng.random() < 0.3 else 0.1
    x = rng.normal(size=(B, d)).astype(np.float32)
    ent = (rng.normal(size=(E, d)) * 0.3).astype(np.float32)
    bias = (rng.normal(size=E) * 0.1).astype(np.float32) if with_bias else None
    lab = (rng.random((B, E)) < dens).astype(np.float32)
    p_ref = ko.head_1n_forward(x, ent, bias)
    loss_ref, dp = ko.multi_class_bce_dir(p_ref, lab, ls, E)
    dx_ref, ge_ref, gb_ref = ko.head_1n_backward(x, ent, p_ref, dp)
    xd, ed = torch.from_numpy(x).cuda(), torch.from_numpy(ent).cuda()
    bd = torch.from_numpy(bias).cuda() if with_bias else None
    off = np.concatenate([[0], np.cumsum(lab.sum(1).astype(np.int64))]).astype(np.int64)
    ids = np.nonzero(lab)[1].astype(np.int32)
    scale = 1.0 / (B * E)
    for tile in (0, 1):
        K.set_switch("HEAD_TILE", tile)
        try:
            p = K.head_1n_forward(xd, ed, bd)
            dx, ge, gb = K.head_1n_backward(xd, ed, p, torch.from_numpy(dp).cuda(), need_bias=with_bias)
            loss_buf = K.new_loss_buffer("cuda")
            g_ent = torch.zeros_like(ed); g_bias = torch.zeros(E, device="cuda") if with_bias else None
            dx2 = K.head_1n_bce(xd, ed, bd, torch.from_numpy(off).cuda(), torch.from_numpy(ids).cuda(), ls, loss_buf, g_ent, g_bias)
            loss = K.read_loss(loss_buf).item()
        finally:
            K.set_switch("HEAD_TILE", None)
        ok = np.allclose(p.cpu().numpy(), p_ref, atol=2e-6)
        for got, ref in ((dx, dx_ref), (ge, ge_ref), (dx2, dx_ref), (g_ent, ge_ref)):
            ok &= np.allclose(got.cpu().numpy(), ref, atol=1e-3 * scale, rtol=1e-3)
        if with_bias:
            ok &= np.allclose(gb.cpu().numpy(), gb_ref, atol=1e-3 * scale, rtol=1e-3) and np.allclose(g_bias.cpu().numpy(), gb_ref, atol=1e-3 * scale, rtol=1e-3)
        ok &= bool(np.isclose(loss, loss_ref, rtol=3e-5, atol=1e-7))
        if not ok:
            bad += 1
            print("BAD", dict(B=B, E=E, d=d, bias=with_bias, dens=dens, ls=ls, tile=tile, loss=(loss, float(loss_ref))), flush=True)
print("fuzz_head: %d bad" % bad)
