"""Pin the sampler restatement (oracle/sampler_oracle.py) to the Random123 known-answer vectors, and the DEVICE sampler
to the restatement bit-exactly (integer / index work: no tolerance)."""
import numpy as np
import pytest

import sampler_oracle as so
from golden_util import Case


def test_philox4x32_10_known_answers():
    for ctr, key, want in so.PHILOX_KAT:
        assert so.philox4x32_10(ctr, key) == want


def test_oracle_sampler_follows_the_reference_rule():
    c = Case("transe_l1")
    train_set = {tuple(map(int, x)) for x in c.train}
    nh, nr, nt = so.corrupt(c.train[:200, 0], c.train[:200, 1], c.train[:200, 2], 3, c.E, None, train_set, 5, 0)
    for j in range(len(nh)):
        h, r, t = map(int, c.train[j // 3])
        assert nr[j] == r and (nh[j] == h) != (nt[j] == t) or (nh[j] == h and nt[j] == t) is False
        assert (int(nh[j]), int(nr[j]), int(nt[j])) not in train_set
    assert 0.4 < np.mean(nh == np.repeat(c.train[:200, 0], 3)) < 0.6


@pytest.mark.gpu
@pytest.mark.parametrize("neg_rate,bern,seed,offset", [(1, False, 0, 0), (4, False, 7, 12345), (3, True, 2 ** 40 + 17, 2 ** 33 + 5)])
def test_device_sampler_is_bit_exact(neg_rate, bern, seed, offset):
    import torch
    import hip_util
    from pykg2vec_amd import kernels as K
    c = Case("transe_l1")
    train = hip_util.dev(c.train)
    slots = K.triple_set_build(train)
    train_set = {tuple(map(int, x)) for x in c.train}
    prob = np.linspace(0.1, 0.9, c.R).astype(np.float32) if bern else None
    bp = torch.from_numpy(prob).cuda() if bern else None
    n = 300
    ph, pr, pt = (train[:n, i].contiguous() for i in range(3))
    nh, nr, nt = K.corrupt(ph, pr, pt, neg_rate, c.E, bp, slots, seed, offset)
    rh, rr, rt = so.corrupt(c.train[:n, 0], c.train[:n, 1], c.train[:n, 2], neg_rate, c.E, prob, train_set, seed, offset)
    assert np.array_equal(nh.cpu().numpy(), rh) and np.array_equal(nr.cpu().numpy(), rr) and np.array_equal(nt.cpu().numpy(), rt)
    # the fused batch sampler draws the same stream through the permutation
    perm = torch.arange(len(c.train), device="cuda").flip(0).contiguous()
    b = K.sample_batch(train, perm, 10, 64, neg_rate, c.E, bp, slots, seed, offset)
    rows = c.train[::-1][10:74]
    sh, sr, st = so.corrupt(rows[:, 0], rows[:, 1], rows[:, 2], neg_rate, c.E, prob, train_set, seed, offset)
    assert np.array_equal(b[0].cpu().numpy(), rows[:, 0]) and np.array_equal(b[3].cpu().numpy(), sh) and np.array_equal(b[5].cpu().numpy(), st)


# ---- rows a16 / a17 / a19 pinned to the reference itself: tests/golden/ref_sampler.npz holds outputs of the reference's
# process_function_pairwise / _pointwise and read_relation_property (oracle/make_golden.py::golden_sampler)
SAMPLER_KEYS = ["%s.%s.%s.n%d" % (g, k, s, n) for g in ("sparse", "dense") for s in ("uniform", "bern") for n in (1, 3)
                for k in ("pairwise", "pointwise")]


def _sampler_case(key):
    import os
    from golden_util import GOLDEN
    z = np.load(os.path.join(GOLDEN, "ref_sampler.npz"))
    g, kind, sampling, n = key.split(".")
    prob64 = z[g + ".relation_property"] if sampling == "bern" else None
    return z, g, kind, int(n[1:]), prob64


@pytest.mark.parametrize("graph", ["sparse", "dense"])
def test_relation_property_equals_reference_output(graph):
    """KnowledgeGraph.read_relation_property (data/kgcontroller.py:466-492) frozen by make_golden: product host code
    and oracle restatement must reproduce the reference's doubles exactly."""
    import os
    import kge_oracle as ko
    from golden_util import GOLDEN
    from pykg2vec_amd.generator import bern_table, relation_property
    z = np.load(os.path.join(GOLDEN, "ref_sampler.npz"))
    train, R, ref = z[graph + ".train"], int(z[graph + ".R"]), z[graph + ".relation_property"]
    assert np.array_equal(relation_property(train, R), ref)
    assert np.array_equal(ko.bern_probability(train, R), ref)
    t32 = bern_table(ref)
    assert t32.dtype == np.float32 and np.array_equal(t32, so.bern_table_f32(ref))
    # u > table[r] must equal the reference's double comparison for EVERY 24-bit uniform next to the threshold
    for p64, p32 in zip(ref, t32):
        k = int(np.floor(p64 * 16777216.0))
        for u in ((k - 1) / 16777216.0, k / 16777216.0, (k + 1) / 16777216.0):
            assert (np.float32(u) > p32) == (u > p64)


@pytest.mark.parametrize("key", SAMPLER_KEYS)
def test_oracle_sampler_equals_reference_function_output(key):
    z, g, kind, neg_rate, prob64 = _sampler_case(key)
    pos, E = z[key + ".pos"], int(z[g + ".E"])
    train_set = {tuple(map(int, x)) for x in z[g + ".train"]}
    table = None if prob64 is None else so.bern_table_f32(prob64)
    nh, nr, nt = so.corrupt(pos[:, 0], pos[:, 1], pos[:, 2], neg_rate, E, table, train_set, int(z[key + ".seed"]),
                            int(z[key + ".offset"]))
    if kind == "pairwise":   # [ph, pr, pt, nh, nr, nt] (data/generator.py:97)
        want = [pos[:, 0], pos[:, 1], pos[:, 2], nh, nr, nt]
    else:                    # [h, r, t, y], every positive followed by its negatives (data/generator.py:125-158)
        import kge_oracle as ko
        want = list(ko.pointwise_layout(pos, nh, nr, nt, neg_rate))
    for i, a in enumerate(want):
        assert np.array_equal(a, z[key + ".out%d" % i]), (key, i)


@pytest.mark.gpu
@pytest.mark.parametrize("key", SAMPLER_KEYS)
def test_device_sampler_equals_reference_function_output(key):
    """kge_sample_batch against the frozen outputs of the reference's process_function_* (same Philox bits)."""
    import torch
    import hip_util
    from pykg2vec_amd import kernels as K
    from pykg2vec_amd.generator import bern_table
    z, g, kind, neg_rate, prob64 = _sampler_case(key)
    train_np, pos, E = z[g + ".train"], z[key + ".pos"], int(z[g + ".E"])
    train = hip_util.dev(train_np)
    slots = K.triple_set_build(train)
    # the batch's positives as a permutation slice of the train array
    index = {tuple(map(int, x)): i for i, x in enumerate(train_np)}
    perm = hip_util.dev(np.asarray([index[tuple(map(int, x))] for x in pos], dtype=np.int64))
    bp = None if prob64 is None else torch.from_numpy(bern_table(prob64)).cuda()
    out = K.sample_batch(train, perm, 0, len(pos), neg_rate, E, bp, slots, int(z[key + ".seed"]), int(z[key + ".offset"]),
                         pointwise=(kind == "pointwise"))
    assert len(out) == (4 if kind == "pointwise" else 6)
    for i, a in enumerate(out):
        assert np.array_equal(a.cpu().numpy().astype(np.int64), z[key + ".out%d" % i]), (key, i)
