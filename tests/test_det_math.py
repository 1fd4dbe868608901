"""CPU suite: the injected-noise spec (oracle/gaz_det.h) — Philox known answers, det_log/det_exp accuracy,
numpy pairwise-sum twin, sampler sanity."""
import ctypes as C
import math

import numpy as np


def test_philox_known_answer(oracle):
    # Random123 kat_vectors: philox4x32-10, counter = key = 0 and the all-ones / pi vectors
    L = oracle.lib()

    def ph(ctr, key):
        c = (C.c_uint32 * 4)(*ctr); k = (C.c_uint32 * 2)(*key); o = (C.c_uint32 * 4)()
        L.gaz_api_philox(c, k, o)
        return [int(x) for x in o]
    assert ph([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert ph([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert ph([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_det_log_exp_accuracy(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.random(20000), rng.random(20000) * 1e6, 2.0 ** rng.uniform(-1000, 1000, 5000),
                         [1.0, 2.0, 0.5, 1e-300, 5e-324, 1.0000000000000002]])
    for x in xs:
        ref = math.log(x); got = L.gaz_api_log(float(x))
        assert abs(got - ref) <= 1.0 * abs(math.ulp(ref)) + 0.0, (x, got, ref)
    for x in np.concatenate([rng.uniform(-700, 700, 20000), rng.uniform(-1, 1, 20000), [0.0, -745.2, 709.5]]):
        ref = math.exp(min(x, 709.0)); got = L.gaz_api_exp(float(x))
        assert abs(got - ref) <= 1.0 * math.ulp(ref) + 5e-324, (x, got, ref)


def test_numpy_pairwise_sum_twin(oracle):
    rng = np.random.default_rng(0)
    for n in list(range(1, 300)) + [1000, 4097]:
        a = rng.random(n).astype(np.float32)
        assert np.sum(a) == oracle.np_sum_f32(a)
        b = rng.random(n)
        assert np.sum(b) == oracle.np_sum_f64(b)


def test_dirichlet_sampler_is_a_dirichlet(oracle):
    # moments of Dirichlet(alpha 1_n): mean 1/n, var (n-1)/(n^2 (n alpha + 1))
    for alpha, n in ((0.5, 7), (0.05, 225), (1.0, 9)):
        draws = np.stack([oracle.dirichlet(42, 0, 0, 0, e, alpha, n) for e in range(3000)])
        assert np.allclose(draws.sum(1), 1.0, atol=1e-12)
        assert (draws >= 0).all()
        assert abs(draws.mean() - 1.0 / n) < 1e-12
        var = (n - 1) / (n * n * (n * alpha + 1))
        assert abs(draws.var(0).mean() - var) < 0.08 * var
    # streams are independent of each other and reproducible
    a = oracle.dirichlet(42, 1, 0, 0, 0, 0.5, 7); b = oracle.dirichlet(42, 1, 0, 0, 0, 0.5, 7)
    c = oracle.dirichlet(42, 1, 0, 1, 0, 0.5, 7)
    assert np.array_equal(a, b) and not np.array_equal(a, c)


def test_pick_and_uniform_ranges(oracle):
    ks = [oracle.pick(7, 0, 0, 0, e, 5) for e in range(5000)]
    assert min(ks) == 0 and max(ks) == 4
    assert abs(np.bincount(ks).min() / 1000.0 - 1.0) < 0.15
    us = [oracle.uniform(7, 0, 0, 0, e, 2) for e in range(2000)]
    assert 0.0 <= min(us) and max(us) < 1.0 and abs(np.mean(us) - 0.5) < 0.03


def test_hash_evaluator_numpy_twin(oracle):
    rng = np.random.default_rng(3)
    for shape, A in (((3, 3, 2), 9), ((6, 7, 4), 7), ((15, 15, 2), 225)):
        s = rng.integers(-1, 2, size=shape).astype(np.int8)
        p1, v1 = oracle.hash_eval(s, A, 11)
        if A <= 9:
            p2, v2 = oracle.hash_eval_np(s, A, 11)
            assert np.array_equal(p1, p2) and v1 == v2
        assert (p1 > 0).all() and (p1 <= 1).all() and -1 <= v1 < 1
