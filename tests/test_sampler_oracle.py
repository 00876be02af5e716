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
