"""
Jump-ahead support for replaying ``numpy.random.RandomState(seed).normal`` (MT19937 + the legacy polar
method, the stream ``randomfield.random.randomize`` draws -- random.py:24-28) on the GPU.

MT19937 is a linear recurrence over GF(2): the sequence of state words x_n satisfies
x_{n+J} = XOR_{j : g_j = 1} x_{n+j} for every n, where g(t) = t^J mod phi(t) and phi is the
characteristic (minimal) polynomial of the generator, of degree 19937.  So the state J words ahead is an
XOR-combination of 19937 + 623 consecutive sequence words -- embarrassingly parallel on a GPU, no
sequential Horner scheme.  This module computes, on the host and once,

* phi (Berlekamp-Massey on one output bit of the generator), and
* the jump polynomials g_k = t^(L * 2^k) mod phi for a fixed segment length L (in words),

as Python integers (bit i = coefficient of t^i); the GPU library receives the positions of their set bits.
A binary tree of such jumps turns the seed state into the start states of 2^K consecutive segments.

Also here: the legacy seeding ``init_genrand`` (numpy's ``RandomState(int)``) and small numpy reference
implementations used by the tests.
"""
from __future__ import annotations

import os

import numpy as np

N, M_ = 624, 397
DEGREE = 19937
SEGMENT_BLOCKS = 1024                 # blocks of 624 words per segment
SEGMENT_WORDS = SEGMENT_BLOCKS * N    # L: 638 976 words = 159 744 polar attempts per segment
_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")      # shipped tables: read only


def cache_dir():
    """Where tables computed at run time are kept: $RANDOMFIELD_CACHE_DIR, else $XDG_CACHE_HOME/randomfield_amd, else
    ~/.cache/randomfield_amd (never the installed package)."""
    d = os.environ.get("RANDOMFIELD_CACHE_DIR") or os.path.join(
        os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache"), "randomfield_amd")
    return d


def _load_polys(path, rows):
    """a cached table, or None when it is missing, short or unreadable (e.g. half written by another rank: recomputed then)"""
    try:
        arr = np.load(path)["polys"]
        return arr if arr.shape[0] >= rows and arr.shape[1] == N else None
    except Exception:
        return None


def _save_polys(path, arr):
    """atomic: temporary file in the same directory, then os.replace -- every rank of a job may get here at once"""
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        tmp = "%s.tmp%d.npz" % (path, os.getpid())
        np.savez_compressed(tmp, polys=arr)
        os.replace(tmp, path)
    except OSError:
        pass

_UPPER, _LOWER, _MATRIX_A = 0x80000000, 0x7FFFFFFF, 0x9908B0DF


def init_genrand(seed):
    """Initial 624-word state of numpy's legacy ``RandomState(seed)`` for an integer seed
    (Knuth's multiplier 1812433253, as in the original mt19937ar.c)."""
    mt = np.empty(N, np.uint32)
    s = int(seed) & 0xFFFFFFFF
    for i in range(N):
        mt[i] = s
        s = (1812433253 * (s ^ (s >> 30)) + i + 1) & 0xFFFFFFFF
    return mt


def init_by_array(key):
    """Initial state of ``RandomState(key)`` for a 1-D array seed (mt19937ar.c ``init_by_array``, the routine
    numpy's legacy seeding calls for anything that is not a scalar integer)."""
    key = [int(v) & 0xFFFFFFFF for v in np.asarray(key).ravel()]
    if not key:
        raise ValueError("Seed must be non-empty")
    mt = [int(v) for v in init_genrand(19650218)]
    i, j = 1, 0
    for _ in range(max(N, len(key))):
        mt[i] = ((mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525)) + key[j] + j) & 0xFFFFFFFF
        i, j = i + 1, j + 1
        if i >= N:
            mt[0], i = mt[N - 1], 1
        if j >= len(key):
            j = 0
    for _ in range(N - 1):
        mt[i] = ((mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941)) - i) & 0xFFFFFFFF
        i += 1
        if i >= N:
            mt[0], i = mt[N - 1], 1
    mt[0] = 0x80000000
    return np.array(mt, np.uint32)


def seed_state(seed):
    """The 624-word state ``RandomState(seed)`` starts from (random.py:24), for every kind of seed numpy's legacy
    seeding accepts: an integer in [0, 2**32) (ValueError outside, as numpy), or an array-like of integers; None
    draws a 32-bit seed from the OS (numpy reads its entropy there too, so such a run is not reproducible either)."""
    if seed is None:
        return init_genrand(int.from_bytes(os.urandom(4), "little"))
    # numpy squeezes ndarray seeds first (a one-element ARRAY seeds like its scalar; a one-element list does not)
    arr = seed.squeeze() if hasattr(seed, "squeeze") else np.asarray(seed)
    arr = np.asarray(arr)
    if arr.ndim == 0 and not hasattr(seed, "squeeze") and not np.isscalar(seed):
        arr = arr.reshape(1)
    if arr.ndim == 0:
        if not np.issubdtype(arr.dtype, np.integer):
            raise TypeError("Cannot cast scalar seed to an integer")
        if not 0 <= int(arr) < 2 ** 32:
            raise ValueError("Seed must be between 0 and 2**32 - 1")
        return init_genrand(int(arr))
    if arr.ndim != 1 or not np.issubdtype(arr.dtype, np.integer):
        raise ValueError("Seed array must be 1-d (integers)")
    if arr.size == 0:
        raise ValueError("Seed must be non-empty")
    if (arr.astype(object) < 0).any() or (arr.astype(object) >= 2 ** 32).any():
        raise ValueError("Seed values must be between 0 and 2**32 - 1")
    return init_by_array(arr)


def next_block(mt):
    """Regenerate the state: words x_{p+624 .. p+1247} from x_{p .. p+623} (vectorised in the three
    dependency-free chunks of the recurrence).  Returns a new array."""
    mt = mt.astype(np.uint64)
    new = np.empty(N, np.uint64)

    def f(a, b, c):
        y = (a & _UPPER) | (b & _LOWER)
        return c ^ (y >> np.uint64(1)) ^ np.where(y & np.uint64(1), np.uint64(_MATRIX_A), np.uint64(0))

    new[:227] = f(mt[:227], mt[1:228], mt[397:624])
    new[227:454] = f(mt[227:454], mt[228:455], new[:227])
    new[454:623] = f(mt[454:623], mt[455:624], new[227:396])
    new[623] = f(mt[623], new[0], new[396])
    return new.astype(np.uint32)


def sequence(state, nwords):
    """x_p .. x_{p+nwords-1} given the state (x_p .. x_{p+623})."""
    out = [np.asarray(state, np.uint32)]
    have = N
    while have < nwords:
        out.append(next_block(out[-1]))
        have += N
    return np.concatenate(out)[:nwords]


def temper(y):
    y = np.asarray(y, np.uint32).copy()
    y ^= y >> np.uint32(11)
    y ^= (y << np.uint32(7)) & np.uint32(0x9D2C5680)
    y ^= (y << np.uint32(15)) & np.uint32(0xEFC60000)
    y ^= y >> np.uint32(18)
    return y


# ----------------------------------------------------------------- GF(2)[t] on python integers

def _berlekamp_massey(bits):
    """Minimal polynomial of a binary sequence (list of 0/1), as an int: bit i = coefficient of t^i of the
    *connection* polynomial C (s_n = XOR_{i>=1} C_i s_{n-i})."""
    n = len(bits)
    s = 0
    for i, b in enumerate(bits):
        if b:
            s |= 1 << i
    C, B, L, m = 1, 1, 0, 1
    for i in range(n):
        # discrepancy d = s_i + sum_{j=1..L} C_j s_{i-j}  ==  parity of (C reversed against the window)
        window = (s >> (i - L)) & ((1 << (L + 1)) - 1) if i >= L else None
        # compute with bit tricks: sum_j C_j s_{i-j}, j = 0..L
        d = 0
        if window is not None:
            # reverse C over L+1 bits and AND with window
            rc = int(format(C, "0%db" % (L + 1))[::-1], 2) if C.bit_length() <= L + 1 else None
            d = bin(rc & window).count("1") & 1
        if d == 0:
            m += 1
        elif 2 * L <= i:
            T = C
            C ^= B << m
            L = i + 1 - L
            B = T
            m = 1
        else:
            C ^= B << m
            m += 1
    return C, L


def characteristic_polynomial():
    """phi(t) of MT19937 as an int (degree 19937): the reversed connection polynomial of the sequence
    formed by one bit of the state words."""
    seq = sequence(init_genrand(5489), 2 * DEGREE + 64)
    bits = [int(v) & 1 for v in seq >> np.uint32(1)]        # bit 1 of every state word
    C, L = _berlekamp_massey_fast(bits)
    if L != DEGREE:
        raise RuntimeError("unexpected linear complexity %d" % L)
    # connection polynomial C(t) = sum C_i t^i with s_n = XOR_{i=1..L} C_i s_{n-i}; the characteristic
    # polynomial is phi(t) = t^L C(1/t)
    phi = 0
    for i in range(L + 1):
        if (C >> i) & 1:
            phi |= 1 << (L - i)
    return phi


def _berlekamp_massey_fast(bits):
    """Berlekamp-Massey with the sequence kept bit-REVERSED so that the discrepancy is one AND + popcount."""
    n = len(bits)
    # R holds s_{i}, s_{i-1}, ... in ascending bit positions: bit j of R = s_{i-j}
    R = 0
    C, B, L, m = 1, 1, 0, 1
    for i in range(n):
        R = (R << 1) | bits[i]
        d = bin(C & R).count("1") & 1                      # sum_{j=0..L} C_j s_{i-j}
        if d == 0:
            m += 1
        elif 2 * L <= i:
            T = C
            C ^= B << m
            L = i + 1 - L
            B = T
            m = 1
        else:
            C ^= B << m
            m += 1
    return C, L


_SPREAD = None


def _square(p):
    """p(t)^2 over GF(2): spread the bits apart (bytewise table)."""
    global _SPREAD
    if _SPREAD is None:
        _SPREAD = []
        for b in range(256):
            v = 0
            for k in range(8):
                if (b >> k) & 1:
                    v |= 1 << (2 * k)
            _SPREAD.append(v)
    raw = p.to_bytes((p.bit_length() + 7) // 8 or 1, "little")
    out = bytearray(2 * len(raw))
    for i, b in enumerate(raw):
        v = _SPREAD[b]
        out[2 * i] = v & 0xFF
        out[2 * i + 1] = v >> 8
    return int.from_bytes(out, "little")


def _mod(p, phi, deg):
    """p mod phi (phi of degree deg)."""
    while p.bit_length() > deg:
        p ^= phi << (p.bit_length() - 1 - deg)
    return p


def power_of_t(J, phi, deg=DEGREE):
    """t^J mod phi by left-to-right square-and-multiply."""
    g = 1
    for bit in bin(J)[2:]:
        g = _mod(_square(g), phi, deg)
        if bit == "1":
            g = _mod(g << 1, phi, deg)
    return g


def jump_polynomials(nlevels=20, cache=True, path=None):
    """g_k = t^(SEGMENT_WORDS * 2^k) mod phi for k < nlevels, as a (nlevels, 624) uint32 array of
    coefficient bits (word w, bit b = coefficient of t^(32 w + b)): the binary jump tree of round 1, kept as an independent
    check of the radix-16 tables (tests/test_mt19937_host.py, which passes its fixture as ``path``)."""
    path = path or os.path.join(cache_dir(), "mt19937_jump_L%d.npz" % SEGMENT_WORDS)
    if cache:
        arr = _load_polys(path, nlevels)
        if arr is not None:
            return arr[:nlevels]
    phi = characteristic_polynomial()
    g = power_of_t(SEGMENT_WORDS, phi)
    rows = []
    for _ in range(nlevels):
        rows.append(np.frombuffer(g.to_bytes(N * 4, "little"), dtype="<u4").astype(np.uint32))
        g = _mod(_square(g), phi, DEGREE)
    arr = np.stack(rows)
    if cache:
        _save_polys(path, arr)
    return arr


TREE_RADIX = 16
WAVE_SLOTS = 4096                     # segments are sized for this many concurrent waves (MI355X: 256 CUs x 16 waves)
_TREE_MEMO = {}


def _tree_cache(segment_blocks, shipped=True):
    return os.path.join(_DATA if shipped else cache_dir(), "mt19937_tree_R%d_L%d.npz" % (TREE_RADIX, segment_blocks * N))


def segment_blocks_for(ncells, slots=WAVE_SLOTS):
    """Blocks of 624 words per segment for a stream of ``ncells`` accepted pairs.  One wave replays one segment and all
    waves are resident at once, so the replay takes as long as the most loaded SIMD: 4300 segments over 4096 wave slots
    put five waves on some SIMDs and four on the others (1024^3 with 1024-block segments; +20 %).  Large streams
    therefore get a whole multiple of ``slots`` segments, the multiple chosen so that a segment stays close to
    SEGMENT_BLOCKS; small ones keep SEGMENT_BLOCKS."""
    blocks = -(-4 * attempts_needed(ncells) // N)
    rounds = int(round(blocks / float(slots * SEGMENT_BLOCKS)))
    if rounds < 1:
        return SEGMENT_BLOCKS
    return -(-blocks // (slots * rounds))


def tree_polynomials(nstages=4, cache=True, segment_blocks=SEGMENT_BLOCKS):
    """The jump polynomials of the radix-16 tree over segments of ``segment_blocks`` blocks: row t*15 + (m-1) =
    t^(m * 16^t * L) mod phi, L = 624 * segment_blocks, m = 1 .. 15, as (nstages*15, 624) uint32 coefficient words.
    4 stages reach 65 536 segments (1.0e10 polar attempts: a 2048^3 grid needs 35 000).  Shipped in the package data for
    the segment lengths of 1024^3 and 2048^3 grids; any other length (non-cubic grids, other large sizes) is built once in
    ~0.2 s per polynomial (45-60 of them, ~10 s), memoised per process and kept in cache_dir() (written atomically: all the
    ranks of a job may build it at once)."""
    rows = nstages * (TREE_RADIX - 1)
    memo = _TREE_MEMO.get(segment_blocks)
    if memo is not None and memo.shape[0] >= rows:
        return memo[:rows]
    if cache:              # shipped with the package (the segment lengths of 1024^3 and 2048^3 grids), else the user's cache
        for path in (_tree_cache(segment_blocks, True), _tree_cache(segment_blocks, False)):
            arr = _load_polys(path, rows)
            if arr is not None:
                _TREE_MEMO[segment_blocks] = arr
                return arr[:rows]
    phi = characteristic_polynomial()
    out = []
    for t in range(nstages):
        for m in range(1, TREE_RADIX):
            g = power_of_t(m * TREE_RADIX ** t * segment_blocks * N, phi)
            out.append(np.frombuffer(g.to_bytes(N * 4, "little"), dtype="<u4").astype(np.uint32))
    arr = np.stack(out)
    _TREE_MEMO[segment_blocks] = arr
    if cache:
        _save_polys(_tree_cache(segment_blocks, False), arr)
    return arr


def set_bit_positions(poly_words):
    """Positions j of the set coefficient bits of one jump polynomial (ascending uint16 array)."""
    bits = np.unpackbits(np.asarray(poly_words, "<u4").view(np.uint8), bitorder="little")
    return np.nonzero(bits)[0].astype(np.uint16)


def jump_state(state, poly_words):
    """numpy reference of the jump: state J words ahead = XOR-combination of sequence words."""
    pos = set_bit_positions(poly_words).astype(np.int64)
    seq = sequence(state, DEGREE + N + 1)
    out = np.zeros(N, np.uint32)
    idx = np.arange(N)
    for j in pos:
        out ^= seq[idx + j]
    return out


def attempts_needed(ncells):
    """Polar attempts to generate so that at least ncells are accepted (acceptance pi/4), with a margin
    of 10 standard deviations plus a constant."""
    p = np.pi / 4
    return int(np.ceil(ncells / p + 10.0 * np.sqrt(ncells * (1 - p)) / p + 1024))
