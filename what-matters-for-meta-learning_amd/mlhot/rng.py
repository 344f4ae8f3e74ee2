"""torch's CPU random stream, handed over to the device (mlhot_mt19937_normal; csrc/mt_normal.h).

The reference draws its Bayes-by-backprop eps with `torch.empty(size).normal_(0, 1)` on the default CPU generator
(bbb/BBBConv.py:86-95).  `DeviceNormal` continues exactly that stream on the GPU:

    dn = DeviceNormal(device, sizes)     # sizes: the element counts of the draws of one step, in call order (each >= 16)
    dn.take_over()                       # the CPU generator's MT19937 engine moves to the device
    for step in ...: eps = dn.draw()     # one flat device tensor; slices at dn.offsets are the step's eps tensors
    dn.hand_back()                       # the CPU generator continues where the device stopped

Between take_over() and hand_back() the torch CPU generator must not be used: the device owns the stream.  The uniforms and
the engine state are bit-identical to what the CPU calls would have produced; the normals agree up to the last ulps of the
device's logf / sincosf against ATen's Sleef (tests: <= 8 ulp over 1 M draws).
"""
import os

import numpy as np
import torch

from . import lib

_N = 624
_LEFT_OFF, _NEXT_OFF, _STATE_OFF = 8, 16, 24        # byte offsets inside torch.get_rng_state() of the CPU generator


def _unpack(rng_state):
    b = rng_state.numpy().tobytes()
    if len(b) < _STATE_OFF + 8 * _N:
        raise ValueError("unexpected CPU generator state layout")
    left = np.frombuffer(b, dtype=np.int32, count=1, offset=_LEFT_OFF)[0]
    nxt = np.frombuffer(b, dtype=np.uint64, count=1, offset=_NEXT_OFF)[0]
    st = np.frombuffer(b, dtype=np.uint64, count=_N, offset=_STATE_OFF).astype(np.uint32)
    return np.concatenate([st, np.array([left, nxt], dtype=np.uint32)])


def _pack(rng_state, engine):
    b = bytearray(rng_state.numpy().tobytes())
    b[_LEFT_OFF:_LEFT_OFF + 4] = np.int32(engine[_N]).tobytes()
    b[_NEXT_OFF:_NEXT_OFF + 8] = np.uint64(engine[_N + 1]).tobytes()
    b[_STATE_OFF:_STATE_OFF + 8 * _N] = engine[:_N].astype(np.uint64).tobytes()
    return torch.from_numpy(np.frombuffer(bytes(b), dtype=np.uint8).copy())


class DeviceNormal:
    """`sub_streams`: K >= 2 draws from K parallel MT19937 sub-streams positioned by jump-ahead polynomials (mlhot/mt_jump.py;
    identical uniforms / normals / final state; ~0.5 s of host arithmetic per table, cached per process); 1 = ONE workgroup
    walking the recurrence.  Default: MLHOT_MT_SUBSTREAMS or 4.  The draw of step k + 1 runs beside step k's kernels and has to be
    finished when step k ends.  Measured on c5 (896 k outputs per draw, MI355X): ONE workgroup takes 0.99 ms alone but ~1.45 ms
    beside a step that fills the chip (its CU is shared with the trunk kernels' workgroups) - as long as the step itself, i.e. the
    draw is the step's floor: 1.527 ms per step; 2 / 3 / 4 sub-streams (0.4-0.6 ms of draw on 2-4 + 8-24 workgroups) 1.474 ms;
    8 sub-streams 1.52 and 64 sub-streams 1.60 - their 56 / 504 jump workgroups take more from the trunk kernels than the shorter
    draw gives back (rocprofv3 timelines: scripts/dev/step_timeline.py)."""

    def __init__(self, device, sizes, sub_streams=None):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("DeviceNormal needs a ROCm device (the CPU route is torch's own normal_())")
        n_sub = int(os.environ.get("MLHOT_MT_SUBSTREAMS", "4")) if sub_streams is None else int(sub_streams)
        sizes = [int(n) for n in sizes]
        if not sizes or min(sizes) < 16:
            raise ValueError("every draw needs >= 16 elements (smaller tensors take torch's scalar double-precision path)")
        self.sizes, self.offsets, segs = sizes, [], []
        dst = src = groups = 0
        for n in sizes:
            self.offsets.append(dst)
            segs.append((dst, n, src, groups))
            dst += (n + 3) // 4 * 4                         # every slice stays 16-byte aligned
            src += n + (16 if n % 16 else 0)
            groups += n // 16 + (1 if n % 16 else 0)
        self.total, self.total_outputs, self.total_groups = dst, src, groups
        self._segs = torch.tensor(segs, dtype=torch.int64).to(self.device)
        self._uniform = torch.empty(src, device=self.device)
        self._engine = None
        # jump-ahead plan: K sub-streams of `stride` blocks each cover the draw whatever the engine's position inside its block
        self._polys = self._jump_ws = None
        if n_sub >= 2 and src >= n_sub * _N * 4:
            from . import mt_jump
            self._n_sub = n_sub
            self._stride = -(-(src // _N + 1) // self._n_sub)            # ceil((blocks + 1) / K)
            polys = mt_jump.jump_polys(self._stride, self._n_sub - 1)      # ~3 s of host arithmetic, once per (stride) and process
            self._polys = torch.from_numpy(polys.view(np.int32).copy()).to(self.device)
            self._jump_ws = torch.empty(lib().mt19937_jump_ws_words(self._n_sub), dtype=torch.int32, device=self.device)

    def take_over(self, generator=None):
        """The (default) CPU generator's engine moves to the device."""
        g = generator if generator is not None else torch.default_generator
        self._generator, self._cpu_state = g, g.get_state()
        self._engine = torch.from_numpy(_unpack(self._cpu_state).view(np.int32).copy()).to(self.device)

    def draw(self, out=None):
        """The next step's draws: one flat float32 device tensor of `total` elements (enqueued on the current stream)."""
        if self._engine is None:
            raise RuntimeError("DeviceNormal.draw(): call take_over() first")
        out = out if out is not None else torch.empty(self.total, device=self.device)
        if self._polys is not None:
            lib().mt19937_normal_par(self._engine, self._uniform, out, self._segs, len(self.sizes), self.total_outputs, self.total_groups,
                                     self._polys, self._n_sub, self._stride, self._jump_ws)
        else:
            lib().mt19937_normal(self._engine, self._uniform, out, self._segs, len(self.sizes), self.total_outputs, self.total_groups)
        return out

    def views(self, flat, shapes):
        return [flat[o:o + n].view(s) for o, n, s in zip(self.offsets, self.sizes, shapes)]

    def hand_back(self):
        """The CPU generator continues where the device stopped (one small device -> host copy; synchronises the stream)."""
        if self._engine is None:
            return
        engine = self._engine.cpu().numpy().view(np.uint32)
        self._generator.set_state(_pack(self._cpu_state, engine))
        self._engine = None
