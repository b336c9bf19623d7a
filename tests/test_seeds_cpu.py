"""CPU: include/npb_seeds.h (host code of libnpb.so) -- the scenario seeds' random streams for whole arrays of seeds, against
the interpreter's own generators: CPython's random.Random(seed) and numpy's legacy RandomState(seed), which are what the
reference's randomisers draw from (randomization_utils.py:32-33,812-813,857,876-886)."""
import random

import numpy as np
import pytest

from nuclear_sim_amd import scenarios

SEEDS = list(range(0, 80)) + [255, 256, 65535, 65536, 2**31 - 1, 2**31, 2**32 - 1, 123456789, 987654321]
BIG = [2**32, 2**32 + 5, 2**40 + 123, 2**62 + 7]     # random.seed takes any int (two 32-bit digits here); numpy's legacy seeding stops at 2**32 - 1


def test_python_stream_is_random_Random():
    k = 700                                            # past one 624-word generation: 1 400 words
    seeds = SEEDS + BIG
    got = scenarios._seed_streams("py", np.array(seeds), k)
    for s, row in zip(seeds, got):
        r = random.Random(s)
        assert np.array_equal(row, [r.random() for _ in range(k)]), s
    r = random.Random(7); r.random()
    assert 59.2 + (59.6 - 59.2) * got[7, 1] == r.uniform(59.2, 59.6)       # Random.uniform is a + (b - a) * random()


def test_numpy_streams_are_the_legacy_RandomState():
    k = 700
    u = scenarios._seed_streams("np", np.array(SEEDS), k)
    g = scenarios._seed_streams("gauss", np.array(SEEDS), k)
    for s, ru, rg in zip(SEEDS, u, g):
        assert np.array_equal(ru, np.random.RandomState(s).random_sample(k)), s
        assert np.array_equal(rg, np.random.RandomState(s).standard_normal(k)), s
    rs = np.random.RandomState(11)
    assert rs.uniform(-0.18, 0.18) == -0.18 + (0.18 - -0.18) * u[11, 0]
    assert np.random.RandomState(11).normal(58.5, 1.75) == 58.5 + 1.75 * g[11, 0]
    # an odd count leaves the cached second value of a pair unused, as RandomState does
    assert np.array_equal(scenarios._seed_streams("gauss", np.array([3]), 5)[0], np.random.RandomState(3).standard_normal(5))


def test_ragged_batches_and_rejected_seeds():
    for n in (1, 15, 16, 17, 1000):
        seeds = np.arange(100, 100 + n)
        got = scenarios._seed_streams("py", seeds, 3)
        for j in (0, n // 2, n - 1):
            r = random.Random(int(seeds[j]))
            assert list(got[j]) == [r.random() for _ in range(3)]
    assert scenarios._seed_streams("py", np.array([], dtype=np.int64), 3).shape[0] == 0
    with pytest.raises(ValueError):
        scenarios._seed_streams("py", np.array([-1]), 2)
    with pytest.raises(ValueError):
        scenarios._seed_streams("np", np.array([2**32]), 2)


def test_thread_count_does_not_change_the_streams():
    from nuclear_sim_amd import _lib
    L = _lib.load()
    seeds = np.arange(5000)
    try:
        L.npb_seed_set_threads(1); a = scenarios._seed_streams("gauss", seeds, 6).copy()
        L.npb_seed_set_threads(7); b = scenarios._seed_streams("gauss", seeds, 6).copy()
    finally:
        L.npb_seed_set_threads(0)
    assert np.array_equal(a, b)
