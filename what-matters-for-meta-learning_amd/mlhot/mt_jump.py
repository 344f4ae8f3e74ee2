"""Jump-ahead polynomials of MT19937 (host side of the parallel device stream, csrc/mt_normal.h).

mlhot_mt19937_normal continues torch's CPU MT19937 stream on the GPU.  The recurrence is sequential from block to block (624
words), so the first version ran ONE workgroup for ~1 ms per c5 step.  MT19937's state transition is linear over GF(2): the raw
word sequence x_0, x_1, ... of the generator satisfies, bit by bit, a linear recurrence whose characteristic polynomial phi has
degree 19937, hence for every J

    x_{n + J} = XOR over { i : g_i = 1 } of x_{n + i},     g(t) = t^J mod phi(t)   (19937 coefficients)

for every n >= 1 (the low 31 bits of the window's first word x_0 are not part of the state; the regeneration reads only its top
bit, and no jumped window's first word is ever output).  So K sub-streams can start at once: one workgroup produces the next
19937 + 624 raw words (33 blocks, ~25 us), K - 1 workgroup groups XOR them together under their polynomials g^(k) = t^(624 S k) mod
phi into the block the stream will hold S k blocks later, and every sub-stream then regenerates its own S blocks.  Uniforms and the
final engine state stay bit-identical to the sequential stream's (tests/test_host_loops.py, tests/test_gpu_parity.py).

This module computes phi once (Berlekamp-Massey on 2 x 19937 output bits) and the polynomials for a block stride S (Haramoto,
Matsumoto, Nishimura, Panneton, L'Ecuyer: "Efficient jump ahead for F2-linear random number generators", 2008 - the published
technique, evaluated here on the output sequence instead of by Horner steps of the state).  Polynomials are Python integers (bit i =
coefficient of t^i); products go through an exact float64 FFT convolution, reductions through shifted XORs of phi.
"""
import numpy as np

N, M, DEG = 624, 397, 19937
WORDS = N                                   # a polynomial travels as 624 uint32 words (19968 bits >= 19937)
_phi = None
_cache = {}


def next_block(st):
    """One regeneration of the 624-word block (ATen MT19937RNGEngine::next_state) as three vector runs + the wrap-around word."""
    def twist(u, v):
        y = (u & np.uint32(0x80000000)) | (v & np.uint32(0x7fffffff))
        return (y >> np.uint32(1)) ^ np.where(v & np.uint32(1), np.uint32(0x9908b0df), np.uint32(0))
    s = st.copy()
    s[0:227] = s[397:624] ^ twist(s[0:227], s[1:228])
    s[227:454] = s[0:227] ^ twist(s[227:454], s[228:455])
    s[454:623] = s[227:396] ^ twist(s[454:623], s[455:624])
    s[623] = s[396] ^ twist(s[623:624], s[0:1])[0]
    return s


def raw_words(st, blocks):
    """`blocks` regenerations of `st`: the raw (untempered) words that follow the block `st`, [blocks * 624]."""
    out = np.empty(blocks * N, dtype=np.uint32)
    for b in range(blocks):
        st = next_block(st)
        out[b * N:(b + 1) * N] = st
    return out


def _berlekamp_massey(bits):
    """Minimal connection polynomial C (C_0 = 1: sum_i C_i s_{n-i} = 0) of a GF(2) sequence given as a list of 0 / 1; returns
    (C as int, linear complexity L).  Python integers as bit vectors: the window s_{n-L..n} is a shift + mask of the whole sequence."""
    n_bits = len(bits)
    # seq_rev bit (n_bits - 1 - j) = s_j, so that bits j = n, n-1, ..., n-L line up with C's bits 0..L after a shift
    seq_rev = 0
    for j, b in enumerate(bits):
        if b:
            seq_rev |= 1 << (n_bits - 1 - j)
    C, B, L, m = 1, 1, 0, 1
    for n in range(n_bits):
        window = (seq_rev >> (n_bits - 1 - n)) & ((1 << (L + 1)) - 1)        # bit i = s_{n-i}
        d = (C & window).bit_count() & 1
        if d:
            T = C
            C ^= B << m
            if 2 * L <= n:
                L, B, m = n + 1 - L, T, 1
            else:
                m += 1
        else:
            m += 1
    return C, L


def char_poly():
    """phi(t), degree 19937, as an int (bit i = coefficient of t^i): the reciprocal of the Berlekamp-Massey connection polynomial of
    one output bit's sequence."""
    global _phi
    if _phi is None:
        st = (np.arange(N, dtype=np.uint64) * np.uint64(1812433253) + np.uint64(5489)).astype(np.uint32)   # any non-degenerate block
        st[0] |= np.uint32(0x80000000)
        words = raw_words(st, (2 * DEG + 64) // N + 2)
        bits = ((words[:2 * DEG + 32] >> np.uint32(0)) & np.uint32(1)).astype(np.uint8).tolist()
        C, L = _berlekamp_massey(bits)
        if L != DEG:
            raise RuntimeError(f"mt_jump: linear complexity {L}, expected {DEG}")
        # s_n = sum_{i>=1} C_i s_{n-i}  <=>  the shift operator satisfies t^L + C_1 t^(L-1) + ... + C_L = 0: phi_j = C_{L-j}
        phi = 0
        for i in range(L + 1):
            if (C >> i) & 1:
                phi |= 1 << (L - i)
        _phi = phi
    return _phi


def _to_bits(p, n):
    return np.unpackbits(np.frombuffer(p.to_bytes((n + 7) // 8, "little"), dtype=np.uint8), bitorder="little")[:n]


def _from_bits(b):
    return int.from_bytes(np.packbits(b.astype(np.uint8), bitorder="little").tobytes(), "little")


def _reduce(p, phi):
    top = p.bit_length() - 1
    while top >= DEG:
        p ^= phi << (top - DEG)
        top = p.bit_length() - 1
    return p


def mul_mod(a, b, phi):
    """a b mod phi over GF(2): exact float64 FFT convolution of the 0 / 1 coefficient vectors (counts <= 19937), parity, reduction."""
    from scipy.signal import fftconvolve
    prod = np.rint(fftconvolve(_to_bits(a, DEG).astype(np.float64), _to_bits(b, DEG).astype(np.float64))).astype(np.int64) & 1
    return _reduce(_from_bits(prod), phi)


def pow_t(e, phi):
    """t^e mod phi."""
    result, base = 1, 2          # 1, t
    while e:
        if e & 1:
            result = mul_mod(result, base, phi) if result != 1 else base
        e >>= 1
        if e:
            base = mul_mod(base, base, phi)
    return _reduce(result, phi)


def jump_polys(stride_blocks, count):
    """[count, 624] uint32: row k - 1 holds g^(k) = t^(624 * stride_blocks * k) mod phi, k = 1 .. count, bit i of the polynomial =
    bit (i % 32) of word (i // 32).  Cached per stride (a longer table extends a shorter one)."""
    phi = char_poly()
    have = _cache.setdefault(stride_blocks, [])
    if len(have) < count:
        step = have[0] if have else pow_t(N * stride_blocks, phi)
        if not have:
            have.append(step)
        while len(have) < count:
            have.append(mul_mod(have[-1], step, phi))
    out = np.zeros((count, WORDS), dtype=np.uint32)
    for k in range(count):
        out[k] = np.frombuffer(have[k].to_bytes(4 * WORDS, "little"), dtype=np.uint32)
    return out


def apply_poly(poly_words, window):
    """Host restatement of the device's jump (tests): window = the 19937 + 624 raw words from x_0 on -> the 624 words x_{J} .. x_{J+623}
    (word 0 valid in its top bit only)."""
    out = np.zeros(N, dtype=np.uint32)
    bits = np.unpackbits(poly_words.view(np.uint8), bitorder="little")[:DEG]
    for i in np.nonzero(bits)[0]:
        out ^= window[i:i + N]
    return out
