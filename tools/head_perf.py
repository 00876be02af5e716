"""Dev tool: 1-N scoring head (kge_head_1n_*) vs the stock ATen chain the reference issues
(matmul + add + sigmoid, BCEWithLogitsLoss on dense multi-hot labels, autograd backward) on one MI355X."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pykg2vec_amd import kernels as K


def bench(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


SHAPES = ((128, 14951, 200), (1000, 14951, 200), (128, 40943, 200), (4096, 14951, 200), (16384, 14951, 200))
if os.environ.get("HEAD_B"):   # one batch size only, no ATen legs: the form to run under rocprofv3
    SHAPES = ((int(os.environ["HEAD_B"]), 14951, 200),)
for B, E, d in SHAPES:
    rng = np.random.default_rng(0)
    x = torch.randn(B, d, device="cuda"); ent = torch.randn(E, d, device="cuda") * 0.2; bias = torch.randn(E, device="cuda") * 0.1
    lab = (torch.rand(B, E, device="cuda") < 0.002).float()
    off = torch.cat([torch.zeros(1, dtype=torch.int64, device="cuda"), lab.sum(1).long().cumsum(0)])
    ids = lab.nonzero()[:, 1].int().contiguous()
    loss_buf = K.new_loss_buffer("cuda"); g_ent = torch.zeros_like(ent); g_bias = torch.zeros(E, device="cuda")
    t_fwd = bench(lambda: K.head_1n_forward(x, ent, bias))
    t_bf16 = bench(lambda: K.head_1n_forward(x, ent, bias, precision="bf16"))
    t_fused = bench(lambda: K.head_1n_bce(x, ent, bias, off, ids, 0.1, loss_buf, g_ent, g_bias))
    preds = K.head_1n_forward(x, ent, bias)
    dpr = torch.randn_like(preds)
    t_bwd = bench(lambda: K.head_1n_backward(x, ent, preds, dpr))      # autograd form: (dx, g_ent, g_bias) from d loss / d preds
    if os.environ.get("HEAD_B"):
        print(f"B={B} E={E} d={d}: forward {t_fwd:.1f} us, bf16 {t_bf16:.1f} us, backward (autograd form) {t_bwd:.1f} us, "
              f"fused head+bce+backward {t_fused:.1f} us", flush=True)
        continue
    xr, er, br = x.clone().requires_grad_(), ent.clone().requires_grad_(), bias.clone().requires_grad_()
    bce = torch.nn.BCEWithLogitsLoss()

    def aten():
        p = torch.sigmoid(torch.matmul(xr, er.T) + br)
        y = lab * 0.9 + 1.0 / E
        loss = bce(p, y)
        loss.backward()
        xr.grad = er.grad = br.grad = None
    t_aten = bench(aten)
    t_aten_fwd = bench(lambda: torch.sigmoid(torch.matmul(x, ent.T) + bias))
    flops = 2.0 * B * E * d
    print(f"B={B} E={E} d={d}: forward {t_fwd:.1f} us ({flops/t_fwd/1e6:.1f} TFLOP/s; ATen {t_aten_fwd:.1f} us; bf16 option {t_bf16:.1f} us = "
          f"{flops/t_bf16/1e6:.1f} TFLOP/s, output write {B*E*4/t_bf16/1e6:.2f} TB/s) | "
          f"fused head+bce+backward {t_fused:.1f} us ({3*flops/t_fused/1e6:.1f} TFLOP/s; ATen chain {t_aten:.1f} us) | "
          f"backward of the autograd form {t_bwd:.1f} us ({2*flops/t_bwd/1e6:.1f} TFLOP/s)", flush=True)
