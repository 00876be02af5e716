"""The multi-threaded C restatement used by bench.py's cpu_baseline must agree with the numpy oracle (itself pinned
to the live reference's golden vectors) and with those golden vectors directly."""
import numpy as np
import pytest

import kge_oracle as ko
import kge_oracle_c as kc
from golden_util import Case, close


@pytest.mark.parametrize("name", ["transe_l1", "transe_l2"])
def test_c_train_steps_match_reference_adam_weights(name):
    c = Case(name)
    P = c.params()
    st = kc.TransEAdam(P["ent_embeddings"], P["rel_embeddings"], c.hp["l1_flag"], c.hp["margin"], 0.05)
    losses = [st.train_step(*c.batch(s)) for s in range(3)]
    assert close(np.asarray(losses, np.float32), c.z["adam.losses"], atol=3e-5, rtol=3e-5)
    assert np.allclose(st.ent, c.z["adam.final.ent_embeddings.weight"], atol=1e-4, rtol=1e-4)
    assert np.allclose(st.rel, c.z["adam.final.rel_embeddings.weight"], atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("name", ["transe_l1", "transe_l2"])
def test_c_eval_matches_reference_ranks(name):
    from pykg2vec_amd.evaluator import build_filter_csr
    c = Case(name)
    P = c.params("adam.final.")
    hr_t, tr_h = c.filters()
    n = len(c.z["eval.rank_head"])
    csr = build_filter_csr(c.test[:n], hr_t, tr_h)
    ranks = kc.transe_eval(P["ent_embeddings"], P["rel_embeddings"], c.hp["l1_flag"], c.test[:n], *csr)
    ref = np.stack([c.z["eval.rank_head"], c.z["eval.rank_tail"], c.z["eval.frank_head"], c.z["eval.frank_tail"]])
    assert (ranks != ref).sum() <= 1 and np.abs(ranks - ref).max() <= 1
    _, rk = ko.evaluate("transe", P, c.test[:n], hr_t, tr_h, **c.hp)
    assert (ranks[1] != rk["tail"]).sum() == 0  # tail sweep: same operation order as the numpy oracle


def test_c_uses_all_cores():
    assert kc.threads() >= 1
