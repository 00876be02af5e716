"""The scorer and the 1-N head as dispatcher-registered custom ops (pykg2vec_amd/ops.py; BASELINE north_star: "a PyTorch-ROCm custom op
that keeps each model's forward()/embed() signature"): schema / fake-tensor / autograd registration checked by torch.library.opcheck,
and `model(h, r, t)` traced by torch.compile without a graph break."""
import numpy as np
import pytest
import torch

from golden_util import Case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import hip_util
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return hip_util


@pytest.mark.parametrize("name", ["transe_l1", "transh_l2", "rotate", "complex", "rescal", "ntn", "quate"])
def test_opcheck_score(hip, name):
    from pykg2vec_amd import ops
    c = Case(name)
    m = hip.model_from_case(c)
    h, r, t = (hip.dev(c.test[:16, i]) for i in range(3))
    weights = [p.weight for p in m.parameter_list]
    key = ops.register_model(m)
    assert ops.register_model(m) == key                         # idempotent handle
    # (test_autograd_registration / test_aot_dispatch_dynamic run the op for real: float atomics in the backward of some models
    # make two runs differ in the last bits, which opcheck's exact comparison would flag -- the static checks are what is asked here)
    torch.library.opcheck(torch.ops.kge.score.default, (key, h, r, t, weights),
                          test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
    want = m(h, r, t)                                           # (Rescal.forward renormalises its tables first, pairwise.py:843-844)
    got = torch.ops.kge.score(key, h, r, t, weights)
    assert got.shape == (16,) and got.dtype == torch.float32 and got.requires_grad
    assert torch.equal(got, want)                               # Model.forward IS this op


def test_opcheck_one_to_n_head(hip):
    rng = np.random.default_rng(0)
    x = torch.tensor(rng.normal(size=(9, 40)), dtype=torch.float32, device="cuda", requires_grad=True)
    ent = torch.tensor(rng.normal(size=(130, 40)) * 0.3, dtype=torch.float32, device="cuda", requires_grad=True)
    bias = torch.tensor(rng.normal(size=(1, 130)) * 0.1, dtype=torch.float32, device="cuda", requires_grad=True)
    for b in (bias, None):
        torch.library.opcheck(torch.ops.kge.one_to_n_scores.default, (x, ent, b, False),
                              test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
    from pykg2vec_amd.head import one_to_n_scores
    want = torch.sigmoid(x @ ent.T + bias)
    got = one_to_n_scores(x, ent, bias)
    assert torch.allclose(got, want, atol=1e-6, rtol=1e-5)
    g = torch.autograd.grad(got.square().sum(), (x, ent, bias))
    w = torch.autograd.grad(want.square().sum(), (x, ent, bias))
    for a, b in zip(g, w):
        assert a.shape == b.shape and torch.allclose(a, b, atol=1e-5, rtol=1e-4)


@pytest.mark.parametrize("name", ["transe_l1", "complex"])
def test_compiled_model_traces_the_scorer_without_a_graph_break(hip, name):
    """torch.compile(fullgraph=True) raises on any graph break; the aot_eager backend traces forward AND backward through the op's fake
    kernel and autograd formula and then runs the captured graphs eagerly (no code generation involved)."""
    c = Case(name)
    m = hip.model_from_case(c)
    h, r, t = (hip.dev(c.test[:32, i]) for i in range(3))
    eager = m(h, r, t)
    eager.sum().backward()
    want_g = [p.weight.grad.clone() for p in m.parameter_list]
    for p in m.parameter_list:
        p.weight.grad = None
    compiled = torch.compile(m, backend="aot_eager", fullgraph=True)
    got = compiled(h, r, t)
    assert torch.equal(got, eager)
    got.sum().backward()
    for p, w in zip(m.parameter_list, want_g):
        assert torch.allclose(p.weight.grad, w, atol=1e-6, rtol=1e-5)
