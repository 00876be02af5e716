"""The staged (atomic-free) RotatE training step (csrc/kge_staged.hip + the STAGED form of k_rotate_bundle_sampled):
gradient rows are stored once per bundle slot and summed per parameter row inside the optimiser sweep.  Held to (a) the
push path (atomic scatter + dense sweep) on the batches the fused sampler draws, (b) the numpy oracle's dense gradient
of the same sampled batch (through an SGD step with lr = 1), (c) itself, bit for bit, across runs."""
import numpy as np
import pytest
import torch

import kge_oracle as ko

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import hip_util
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return hip_util


def _world(E, R, D, n_train, seed=7):
    rng = np.random.default_rng(seed)
    train = np.stack([rng.integers(E, size=n_train), rng.integers(R, size=n_train), rng.integers(E, size=n_train)], 1)
    test = np.stack([rng.integers(E, size=8), rng.integers(R, size=8), rng.integers(E, size=8)], 1)
    P = ko.init_params("rotate", rng, tot_entity=E, tot_relation=R, hidden_size=D, margin=6.0)
    return train, test, P


def _trainer(hip, world, E, R, D, B, neg, opt, staged, monkeypatch, lr=0.01):
    from pykg2vec_amd.trainer import Trainer
    train, test, P = world
    hp = dict(hidden_size=D, margin=6.0, neg_rate=neg, alpha=0.5)
    cfg = hip.make_config(E, R, hp, train, test, test, optimizer=opt, lr=lr, batch_size=B)
    m = hip.model_from_params("rotate", P, hp, E, R)
    monkeypatch.setenv("KGE_STAGED", "1" if staged else "0")
    tr = Trainer(m, cfg, use_graph=False)
    tr.build_model()
    tr.generator = tr._new_generator()
    assert tr._staged_ok() == staged
    return tr, m


def _run(tr, m, hip, n_steps):
    from pykg2vec_amd import kernels as K
    tr.generator.start_one_epoch(n_steps)
    tr.loss_buf.zero_()
    tr.step_next_batches(n_steps)
    tr.sync_model()
    return K.read_loss(tr.loss_buf).item(), {k: p.detach().clone() for k, p in hip.table_parameters(m)}


@pytest.mark.parametrize("E,R,D,B,neg,steps", [(53, 7, 40, 32, 4, 3),        # toy: short last batch included (n_train 80)
                                               (12, 2, 8, 32, 8, 2),         # every entity drawn > 16 times: overflow chains (E large enough for the
                                                                             # train-set rejection to find negatives)
                                               (2000, 37, 1000, 256, 16, 2)])  # C3 row length and negative rate
@pytest.mark.parametrize("opt", ["sgd", "adam", "adagrad", "rms"])
def test_staged_steps_equal_push_steps(hip, monkeypatch, E, R, D, B, neg, steps, opt):
    world = _world(E, R, D, (80 if E > 12 else 40) if B == 32 else 3 * B)
    res = {}
    for staged in (False, True):
        tr, m = _trainer(hip, world, E, R, D, B, neg, opt, staged, monkeypatch)
        res[staged] = _run(tr, m, hip, steps)
    assert np.isclose(res[True][0], res[False][0], rtol=2e-5), (res[True][0], res[False][0])
    for k in res[True][1]:
        a, b = res[True][1][k].cpu().numpy(), res[False][1][k].cpu().numpy()
        bad = ~np.isclose(a, b, atol=2e-5, rtol=1e-4)
        # Adam / Adagrad / RMSprop divide by sqrt(accumulated g^2): a gradient entry that is pure rounding residue (sums that
        # cancel in one order and not in the other) moves the parameter by O(lr) either way; such entries are rare
        lim = 0.0 if opt == "sgd" else 2e-3
        assert bad.mean() <= lim, (opt, k, bad.mean(), np.abs(a - b).max())


def test_staged_training_is_bit_reproducible(hip, monkeypatch):
    E, R, D, B, neg = 2000, 37, 1000, 256, 16
    world = _world(E, R, D, 3 * B)
    out = []
    for _ in range(2):
        tr, m = _trainer(hip, world, E, R, D, B, neg, "adam", True, monkeypatch)
        out.append(_run(tr, m, hip, 3))
    # (the reported loss is a float-atomic sum: equal up to the order of those additions; tables involve no atomics)
    assert np.allclose(out[0][0], out[1][0], rtol=1e-6), (out[0][0], out[1][0])
    for k in out[0][1]:
        assert torch.equal(out[0][1][k], out[1][1][k]), k


@pytest.mark.parametrize("D,staged", [(64, True), (1200, True), (2048, True), (1028, True),
                                      (1200, False)])   # rows beyond 1 024 floats: the staged form over four waves per bundle; without the
                                                        # stage sink the epoch takes the stand-alone sampler + the explicit-id bundle kernel
def test_staged_gradient_matches_oracle(hip, monkeypatch, D, staged):
    """One SGD step with lr = 1 turns the epoch's step path into its own gradient: p_before - p_after must be the oracle's dense
    gradient of the batch the sampler drew (kge_sample_batch with the same counters).  D > 1 024 is the round-5 advisor case:
    Trainer.step_next_batches (what train_model_epoch runs) on the sampled and the staged path, every element of a row covered."""
    from pykg2vec_amd import kernels as K
    E, R, B, neg = 300, 11, 128, 8
    world = _world(E, R, D, B)
    tr, m = _trainer(hip, world, E, R, D, B, neg, "sgd", staged, monkeypatch, lr=1.0)
    assert tr._fused_rotate_ok() == (D <= 1024)
    before = {k[:-len(".weight")]: p.detach().cpu().numpy().copy() for k, p in hip.table_parameters(m)}
    gen = tr.generator
    ph, pr, pt, nh, nr, nt = [x.cpu().numpy() for x in
                              K.sample_batch(gen.triples, gen.perm, 0, B, neg, E, gen.bern, gen.slots, gen.seed, 0)]
    loss, after = _run(tr, m, hip, 1)
    want_loss, grads, _, _ = ko.train_step_grads("rotate", before, (ph, pr, pt, nh, nr, nt), hidden_size=D, margin=6.0,
                                                 neg_rate=neg, alpha=0.5)
    assert np.isclose(loss, want_loss, rtol=2e-5)
    for k in before:
        got = before[k] - after[k + ".weight"].cpu().numpy()
        assert np.allclose(got, grads[k], atol=2e-6, rtol=2e-4), (k, np.abs(got - grads[k]).max())
        if D > 1024:     # the columns past 1 024 carry gradient too (they were silently dropped before round 6)
            assert np.abs(grads[k][:, 1024:]).max() > 0 and np.abs(got[:, 1024:]).max() > 0, k


def test_sampled_rotate_refuses_wide_rows_without_a_sink(hip, monkeypatch):
    """The fused-sampler entry point without a stage sink has no kernel for rows of more than 1 024 floats: loud, not partial."""
    from pykg2vec_amd import kernels as K
    from pykg2vec_amd._lib import KgeHipError
    E, R, D, B, neg = 300, 11, 1200, 64, 4
    tr, m = _trainer(hip, _world(E, R, D, B), E, R, D, B, neg, "sgd", False, monkeypatch)
    gen = tr.generator
    with pytest.raises(KgeHipError, match="1024"):
        K.train_pairwise_selfadv_sampled(tr._desc, gen.triples, gen.perm, 0, B, neg, 0.5, gen.bern, gen.slots, gen.seed, 0,
                                         tr.loss_buf)


def _pw_trainer(hip, model, world, E, R, D, B, neg, opt, staged, monkeypatch, lr=0.01):
    from pykg2vec_amd.trainer import Trainer
    train, test, P = world
    hp = dict(hidden_size=D, lmbda=1e-3, neg_rate=neg)
    cfg = hip.make_config(E, R, hp, train, test, test, optimizer=opt, lr=lr, batch_size=B)
    m = hip.model_from_params(model, P, hp, E, R)
    monkeypatch.setenv("KGE_STAGED", "1" if staged else "0")
    tr = Trainer(m, cfg, use_graph=False)
    tr.build_model()
    tr.generator = tr._new_generator()
    assert tr._staged_ok() == staged
    return tr, m


@pytest.mark.parametrize("model,E,R,D,B,neg,steps", [("distmult", 53, 7, 40, 32, 3, 3), ("complex", 53, 7, 40, 32, 2, 3),
                                                     ("complexn3", 53, 7, 40, 32, 1, 3),
                                                     ("complex", 12, 2, 8, 32, 8, 2),          # overflow chains
                                                     ("complex", 4000, 11, 200, 512, 1, 2),    # C2 row length, few relations
                                                     ("complex", 4000, 3, 200, 512, 1, 2),     # relation lists of ~170 slots: chunked pre-reduction
                                                     ("distmult", 3000, 400, 100, 1024, 1, 2)])
@pytest.mark.parametrize("opt", ["sgd", "adagrad", "adam"])
def test_staged_pointwise_steps_equal_push_steps(hip, monkeypatch, model, E, R, D, B, neg, steps, opt):
    rng = np.random.default_rng(11)
    n_train = (80 if E > 12 else 40) if B == 32 else 2 * B + 17
    train = np.stack([rng.integers(E, size=n_train), rng.integers(R, size=n_train), rng.integers(E, size=n_train)], 1)
    test = train[:8]
    P = ko.init_params("complex" if model.startswith("complex") else model, rng, tot_entity=E, tot_relation=R, hidden_size=D)
    res = {}
    for staged in (False, True):
        tr, m = _pw_trainer(hip, model, (train, test, P), E, R, D, B, neg, opt, staged, monkeypatch)
        res[staged] = _run(tr, m, hip, steps)
    assert np.isclose(res[True][0], res[False][0], rtol=2e-5), (res[True][0], res[False][0])
    for k in res[True][1]:
        a, b = res[True][1][k].cpu().numpy(), res[False][1][k].cpu().numpy()
        bad = ~np.isclose(a, b, atol=2e-5, rtol=1e-4)
        lim = 0.0 if opt == "sgd" else 2e-3
        assert bad.mean() <= lim, (opt, k, bad.mean(), np.abs(a - b).max())
