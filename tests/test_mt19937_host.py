"""Host-side mathematics of the MT19937 replay (randomfield_amd/mt19937.py): seeding, the recurrence, the
characteristic polynomial and the jump polynomials the GPU library is fed with.  No GPU needed."""
import numpy as np
import pytest

from randomfield_amd import mt19937 as mt


import os
# the round-1 binary jump tree t^(L 2^k): an independent table the radix-16 one is checked against (a fixture, not product data)
BINARY_TREE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mt19937_jump_L638976.npz")


def _same_state(a, b):
    """MT19937 states are equal when all words agree except the unused low 31 bits of word 0."""
    a, b = np.asarray(a, np.uint32), np.asarray(b, np.uint32)
    return np.array_equal(a[1:], b[1:]) and ((int(a[0]) ^ int(b[0])) & 0x80000000) == 0


@pytest.mark.parametrize("seed", [0, 1, 123, 2 ** 32 - 1])
def test_seeding_matches_numpy(seed):
    assert np.array_equal(mt.init_genrand(seed), np.random.RandomState(seed).get_state()[1])


@pytest.mark.parametrize("key", [[0, 0], [1, 2, 3], [2 ** 32 - 1] * 7, list(range(700)), np.arange(5, dtype=np.int64) * 123456789])
def test_array_seeding_matches_numpy(key):
    """init_by_array: the state numpy's legacy RandomState(array) starts from (random.py:24 accepts any such seed)."""
    want = np.random.RandomState(key).get_state()[1]
    assert np.array_equal(mt.init_by_array(key), want)
    assert np.array_equal(mt.seed_state(np.asarray(key)), want)
    assert np.array_equal(mt.seed_state([5]), np.random.RandomState([5]).get_state()[1])     # a list: init_by_array
    assert np.array_equal(mt.seed_state(np.array([5])), np.random.RandomState(np.array([5])).get_state()[1])   # squeezed to 5


def test_seed_state_rules():
    assert np.array_equal(mt.seed_state(7), mt.init_genrand(7))
    assert mt.seed_state(None).shape == (624,)
    with pytest.raises(ValueError):
        mt.seed_state(2 ** 32)
    with pytest.raises(ValueError):
        mt.seed_state(-1)
    with pytest.raises(ValueError):
        mt.seed_state([[1, 2], [3, 4]])
    with pytest.raises(ValueError):
        mt.seed_state([1, 2 ** 32])


def test_sequence_and_tempering_match_numpy_raw_words():
    n = 5 * mt.N
    raw = np.frombuffer(np.random.RandomState(42).bytes(4 * n), dtype="<u4")
    seq = mt.sequence(mt.init_genrand(42), n + mt.N)
    assert np.array_equal(mt.temper(seq[mt.N:]), raw)         # outputs are the tempered words of the NEXT blocks


@pytest.fixture(scope="module")
def phi():
    return mt.characteristic_polynomial()


def test_characteristic_polynomial(phi):
    assert phi.bit_length() - 1 == mt.DEGREE == 19937
    assert bin(phi).count("1") == 135                          # Matsumoto & Nishimura 1998, table II
    # phi annihilates the sequence of any bit of the state words
    seq = mt.sequence(mt.init_genrand(99), mt.DEGREE + 700)
    pos = np.nonzero(np.array([(phi >> i) & 1 for i in range(mt.DEGREE + 1)]))[0]
    for n in (1, 17, 600):
        assert np.bitwise_xor.reduce(seq[n + pos]) == 0


def test_cached_jump_polynomials_are_powers_of_t(phi):
    polys = mt.jump_polynomials(16, path=BINARY_TREE)
    g = mt.power_of_t(mt.SEGMENT_WORDS, phi)
    for k in range(16):
        assert np.array_equal(np.frombuffer(g.to_bytes(mt.N * 4, "little"), dtype="<u4"), polys[k]), k
        g = mt._mod(mt._square(g), phi, mt.DEGREE)


def test_jump_equals_stepping():
    polys = mt.jump_polynomials(3, path=BINARY_TREE)
    st = mt.init_genrand(123)
    seq = mt.sequence(st, 3 * mt.SEGMENT_WORDS + mt.N)
    one = mt.jump_state(st, polys[0])
    assert _same_state(one, seq[mt.SEGMENT_WORDS:mt.SEGMENT_WORDS + mt.N])
    two = mt.jump_state(st, polys[1])
    assert _same_state(two, seq[2 * mt.SEGMENT_WORDS:2 * mt.SEGMENT_WORDS + mt.N])
    assert _same_state(mt.jump_state(one, polys[0]), two)
    assert _same_state(mt.jump_state(one, polys[1]), seq[3 * mt.SEGMENT_WORDS:])


def test_attempts_margin():
    for ncells in (1, 4096, 10 ** 6, 2 ** 29 + 2 ** 20):
        need = mt.attempts_needed(ncells)
        p = np.pi / 4
        assert (need * p - ncells) / np.sqrt(need * p * (1 - p)) > 9.0     # > 9 sigma of head-room


def test_tree_polynomials_are_the_right_powers(phi):
    """rows of the radix-16 tree table: t^(m 16^t L); where m 16^t is a power of two they equal the binary-tree rows"""
    tree = mt.tree_polynomials(3)
    binary = mt.jump_polynomials(10, path=BINARY_TREE)
    assert tree.shape == (45, mt.N)
    for (t, m, k) in ((0, 1, 0), (0, 2, 1), (0, 4, 2), (0, 8, 3), (1, 1, 4), (1, 2, 5), (2, 1, 8), (2, 2, 9)):
        assert np.array_equal(tree[t * 15 + m - 1], binary[k])
    g = mt.power_of_t(3 * 16 * mt.SEGMENT_WORDS, phi)
    assert np.array_equal(tree[15 + 2], np.frombuffer(g.to_bytes(mt.N * 4, "little"), dtype="<u4"))
    # a jump by 5 segments lands on the sequence 5 L words ahead
    st = mt.init_genrand(4)
    seq = mt.sequence(st, 5 * mt.SEGMENT_WORDS + mt.N)
    assert _same_state(mt.jump_state(st, tree[4]), seq[5 * mt.SEGMENT_WORDS:5 * mt.SEGMENT_WORDS + mt.N])


def test_digitwise_jump_to_a_rank_first_segment(tmp_path, monkeypatch):
    """The shared replay (rf_mt_share_begin) reaches segment s from the seed's state with one jump per non-zero radix-16 digit
    of s: digit d of weight 16^t uses row t*15 + d - 1 of the tree table.  Short segments (16 blocks) so that stepping the
    generator to the target is cheap: s = 17 (two digits), 32 (one digit of weight 16), 9 (one digit)."""
    monkeypatch.setenv("RANDOMFIELD_CACHE_DIR", str(tmp_path))
    blocks = 16
    tree = mt.tree_polynomials(2, cache=False, segment_blocks=blocks)
    L = blocks * mt.N
    st = mt.init_genrand(2026)
    seq = mt.sequence(st, 33 * L + mt.N)
    for s in (17, 32, 9):
        cur, rest, t = st, s, 0
        while rest:
            d = rest % 16
            if d:
                cur = mt.jump_state(cur, tree[t * 15 + d - 1])
            rest //= 16
            t += 1
        assert _same_state(cur, seq[s * L:s * L + mt.N])


def test_tables_computed_at_run_time_go_to_the_user_cache_atomically(tmp_path, monkeypatch):
    """A segment length without a shipped table is computed once, written to cache_dir() through a temporary file, found
    there by the next process -- and a truncated file (another rank caught mid-write under the old scheme) is recomputed."""
    monkeypatch.setenv("RANDOMFIELD_CACHE_DIR", str(tmp_path))
    mt._TREE_MEMO.pop(7, None)
    a = mt.tree_polynomials(1, segment_blocks=7)
    path = mt._tree_cache(7, shipped=False)
    assert path.startswith(str(tmp_path)) and os.path.exists(path) and not [f for f in os.listdir(str(tmp_path)) if ".tmp" in f]
    assert not os.path.exists(mt._tree_cache(7, shipped=True))          # nothing is written into the package
    mt._TREE_MEMO.pop(7, None)
    assert np.array_equal(mt.tree_polynomials(1, segment_blocks=7), a)
    with open(path, "r+b") as f:
        f.truncate(100)
    mt._TREE_MEMO.pop(7, None)
    assert np.array_equal(mt.tree_polynomials(1, segment_blocks=7), a)
    mt._TREE_MEMO.pop(7, None)
