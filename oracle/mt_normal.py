"""CPU restatement of torch's CPU `normal_()` random stream.  TEST INFRASTRUCTURE ONLY (the checker of mlhot_mt19937_normal).

The reference draws every Bayes-by-backprop eps with `torch.empty(size).normal_(0, 1)` on the default CPU generator
(/root/reference/networks/bbb/BBBConv.py:86-95, BBBLinear.py:79-88).  The arithmetic behind that call lives in the pinned
third-party dependency torch (ATen), not in the reference's tree; its published algorithm, restated here with numpy:

  * engine: MT19937 (ATen/core/MT19937RNGEngine.h): 624-word state, `if (--left == 0) next_state(); y = state[next++]`,
    standard twist (M = 397, matrix 0x9908b0df) and tempering (11 / 7 & 0x9d2c5680 / 15 & 0xefc60000 / 18);
  * float uniform in [0, 1): (y & (2^24 - 1)) * 2^-24 (ATen/core/TransformationHelper.h uniform_real, 24 digits);
  * normal_() of a contiguous float tensor with >= 16 elements (ATen/native/cpu/DistributionTemplates.h normal_fill): fill the
    whole tensor with uniforms in order, then for every group of 16: for j < 8: u1 = 1 - x[j], u2 = x[j + 8],
    r = sqrt(-2 log u1), th = 2 pi u2, x[j] = r cos th, x[j + 8] = r sin th; a size that is not a multiple of 16 redraws
    16 fresh uniforms for the LAST 16 elements and transforms those again.  (Tensors below 16 elements take a scalar
    double-precision path; every eps tensor of the in-scope models has >= 32 elements, the restatement refuses smaller ones.)

Pinned by tests/test_oracle_golden.py::test_mt_normal_*: the uniforms and the generator state after a draw are compared with
torch bit for bit, the normals within a few ulp (numpy's log / sin / cos vs Sleef's inside ATen).
"""
import numpy as np
import torch

N, M = 624, 397
_STATE_OFF, _LEFT_OFF, _NEXT_OFF = 24, 8, 16          # byte offsets inside torch.get_rng_state() (THGeneratorState layout)


def unpack_state(rng_state):
    """torch.get_rng_state() (uint8[5056]) -> (state uint32[624], left, next)."""
    b = rng_state.numpy().tobytes()
    left = int(np.frombuffer(b, dtype=np.int32, count=1, offset=_LEFT_OFF)[0])
    nxt = int(np.frombuffer(b, dtype=np.uint64, count=1, offset=_NEXT_OFF)[0])
    st = np.frombuffer(b, dtype=np.uint64, count=N, offset=_STATE_OFF).astype(np.uint32)
    return st.copy(), left, nxt


def pack_state(rng_state, st, left, nxt):
    """The same byte tensor with (state, left, next) replaced; the cached-normal fields are cleared like torch does not need them."""
    b = bytearray(rng_state.numpy().tobytes())
    b[_LEFT_OFF:_LEFT_OFF + 4] = np.int32(left).tobytes()
    b[_NEXT_OFF:_NEXT_OFF + 8] = np.uint64(nxt).tobytes()
    b[_STATE_OFF:_STATE_OFF + 8 * N] = st.astype(np.uint64).tobytes()
    return torch.from_numpy(np.frombuffer(bytes(b), dtype=np.uint8).copy())


def next_state(st):
    """One regeneration of the 624-word block (in place semantics of MT19937RNGEngine::next_state, restated sequentially in
    three vectorisable runs: 227 + 227 + 169 words and the wrap-around word)."""
    def twist(u, v):
        y = (u & np.uint32(0x80000000)) | (v & np.uint32(0x7fffffff))
        return (y >> np.uint32(1)) ^ np.where(v & np.uint32(1), np.uint32(0x9908b0df), np.uint32(0))
    s = st.copy()
    s[0:227] = s[397:624] ^ twist(s[0:227], s[1:228])                 # j < N - M: partner p[M] is still the old block
    s[227:454] = s[0:227] ^ twist(s[227:454], s[228:455])            # partner p[M - N]: the words just made
    s[454:623] = s[227:396] ^ twist(s[454:623], s[455:624])
    s[623] = s[396] ^ twist(s[623:624], s[0:1])[0]
    return s


def temper(y):
    y = y ^ (y >> np.uint32(11))
    y = y ^ ((y << np.uint32(7)) & np.uint32(0x9d2c5680))
    y = y ^ ((y << np.uint32(15)) & np.uint32(0xefc60000))
    return y ^ (y >> np.uint32(18))


def raw_outputs(st, left, nxt, count):
    """`count` consecutive 32-bit outputs of the engine and the (state, left, next) it is left in."""
    out = np.empty(count, dtype=np.uint32)
    done = 0
    while done < count:
        if left <= 1:                                  # `--left == 0` on the next call: regenerate first
            st, left, nxt = next_state(st), N + 1, 0
        take = min(count - done, left - 1)
        out[done:done + take] = temper(st[nxt:nxt + take])
        done, left, nxt = done + take, left - take, nxt + take
    return out, st, left, nxt


def uniforms(raw):
    return ((raw & np.uint32((1 << 24) - 1)).astype(np.float32) * np.float32(1.0 / (1 << 24))).astype(np.float32)


def box_muller_16(x):
    """x: float32 [groups, 16] of uniforms -> normals, group by group like normal_fill_16."""
    u1 = (np.float32(1.0) - x[:, :8]).astype(np.float32)
    u2 = x[:, 8:]
    r = np.sqrt((np.float32(-2.0) * np.log(u1)).astype(np.float32)).astype(np.float32)
    th = (np.float32(2.0 * np.pi) * u2).astype(np.float32)
    return np.concatenate([(r * np.cos(th)).astype(np.float32), (r * np.sin(th)).astype(np.float32)], axis=1)


def normal_(size, st, left, nxt):
    """torch.empty(size).normal_() drawn from engine (st, left, nxt) -> (float32 [size], new engine)."""
    if size < 16:
        raise ValueError("tensors below 16 elements take torch's scalar double-precision path (not restated)")
    raw, st, left, nxt = raw_outputs(st, left, nxt, size)
    x = uniforms(raw)
    body = size - size % 16
    x[:body] = box_muller_16(x[:body].reshape(-1, 16)).reshape(-1)
    if size % 16:
        raw, st, left, nxt = raw_outputs(st, left, nxt, 16)
        x[size - 16:] = box_muller_16(uniforms(raw).reshape(1, 16)).reshape(-1)
    return x, st, left, nxt
