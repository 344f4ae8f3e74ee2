"""Where the Bayes-by-backprop layers get their eps from.

The reference draws eps per layer on the torch CPU generator inside every forward and copies it to the device
(bbb/BBBConv.py:88-95: `torch.empty(size).normal_(0, 1).to(device)`), weight first, then bias.  `draw()` is that route
by default.  A `StagedEps` pre-draws the SAME sequence (same generator, same shapes, same order, hence the same numbers)
before the forward starts, ships it in one pinned host -> device copy and hands the layers slices of a fixed device
buffer - which is what makes a Bayes-by-backprop training step capturable in a hipGraph:

    eps = StagedEps(device)
    with eps.recording():  step()         # one eager step with lazy draws; notes the shapes in call order
    eps.stage()
    with eps.active():     graph = capture(step)
    for it in ...:         eps.stage(); eps.prefetch(); graph.replay()

`prefetch()` takes the draws off the critical path: step k+1's eps are drawn on a worker thread (the torch CPU generator, same
order, so the numbers are exactly those of the lazy route) into the alternate pinned buffer while step k runs on the GPU; the
next `stage()` only waits for that thread and starts the copy.  A step then costs max(GPU time, draw time), not their sum.  The
drawer thread also sends its numbers across PCIe itself, on a copy stream, into one of two device-side staging buffers (3.5 MB per
c5 step: ~65 us the replaying stream would otherwise wait for); `stage()` is then an event wait and a device -> device copy.
Nothing else may use the torch CPU generator between prefetch() and the stage() that collects it.

The host draw itself runs on `threads` host threads (default MLHOT_EPS_THREADS or 4): the recorded sequence is cut - at multiples
of 16 outputs, where ATen's normal_fill has no state beyond the engine - into that many pieces of equal length, the engine state
at the start of every piece comes from mlhot_mt19937_advance (the MT19937 recurrence without the outputs, ~0.3 ms per million on
one core), every piece is drawn by `normal_()` itself on a torch.Generator of its own, and the CPU generator is left where the
sequential draw would leave it.  Same numbers, bit for bit (checked once per plan against the one-thread draw); c5's 896 k normals
per step: 1.83 ms on one thread of the GPU box - longer than the 1.43 ms GPU step, i.e. the step's floor - against ~0.3 + 1.83 / K.

`StagedEps(device, source="device")` lifts the host out of the loop altogether: the CPU generator's MT19937 engine is handed to
the device (mlhot.rng.DeviceNormal -> mlhot_mt19937_normal) and every stage() receives the SAME random stream from there -
identical uniforms, normals equal to torch's up to <= 4 ulp of logf / sincosf (so a model output moves by ~1e-7, far inside the
1e-4 parity bar, but it is not bit-identical: the host route stays the default).  Step k+1's numbers are produced on a side
stream while step k runs (one CU for ~0.3 ms); stage() is then an event wait and one 3.5 MB device copy.  release() hands the
engine back: the CPU generator continues exactly where a host-only run would be.
"""
import contextlib
import os
import threading
from concurrent.futures import ThreadPoolExecutor

import torch

from mlhot.binding import MlhotError

_active = None        # the StagedEps the layers read from (None: the reference's lazy route); callers are single-threaded
_recorder = None


def draw(size, device):
    """One eps tensor for a Bayes-by-backprop layer, on `device`."""
    size = tuple(size)
    if _active is not None:
        return _active._next(size, device)
    if _recorder is not None:
        _recorder.shapes.append(size)
    return torch.empty(size).normal_(0, 1).to(device)


def usable_cores():
    try:
        return len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return os.cpu_count() or 1


def default_threads():
    """Host threads of one rank's eps draw: MLHOT_EPS_THREADS, else min(4, usable cores // (2 x ranks on this node)), at least 1 - a
    real loader's workers (imgaug on the host) need the other half, and 8 ranks x (4 draw threads + the drawer + the batch prefetch)
    must not oversubscribe the node.  An explicit MLHOT_EPS_THREADS is capped by usable cores // ranks as well."""
    ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE") or os.environ.get("WORLD_SIZE") or 1))
    cores = usable_cores()
    env = os.environ.get("MLHOT_EPS_THREADS")
    if env is not None:
        return max(1, min(int(env), max(1, cores // ranks)))
    return max(1, min(4, cores // (2 * ranks)))


class StagedEps:
    def __init__(self, device, source="host", threads=None):
        if source not in ("host", "device"):
            raise ValueError("StagedEps: source is 'host' (torch CPU generator, bit-exact) or 'device' (the same stream on the GPU)")
        self.device = torch.device(device)
        self.source = source
        self.threads = default_threads() if threads is None else max(1, int(threads))
        self._pieces, self._pool, self._gens = None, None, None
        self._dn, self._dn_buf, self._dn_stream, self._dn_ready, self._dn_free = None, None, None, None, None
        self.shapes = []
        self._offsets, self._total, self._cursor = None, 0, 0
        self._host, self._dev, self._done, self._turn = None, None, None, 0
        self._worker, self._worker_err = None, None
        self._thread, self._go, self._ready, self._job = None, None, None, 0
        self._runs = None

    @contextlib.contextmanager
    def recording(self):
        global _recorder
        self.shapes, self._offsets, self._runs, self._pieces = [], None, None, None
        if self._pool is not None:
            self._pool.shutdown(wait=False)          # a new recording re-plans the pieces: the old plan's threads are not needed
            self._pool = None
        _recorder = self
        try:
            yield self
        finally:
            _recorder = None

    @contextlib.contextmanager
    def active(self):
        global _active
        _active = self
        try:
            yield self
        finally:
            _active = None

    def _plan(self):
        offs, n = [], 0
        for s in self.shapes:
            offs.append(n)
            n += (int(torch.Size(s).numel()) + 3) // 4 * 4          # keep every slice 16-byte aligned
        self._offsets, self._total = offs, n
        pin = self.device.type == "cuda"
        self._host = [torch.empty(n).pin_memory() if pin else torch.empty(n) for _ in range(2)]
        # prefetched draws cross PCIe on a stream of their own, under the step that is running (3.5 MB per c5 step = ~65 us that the
        # replaying stream would otherwise wait for); stage() then only copies device -> device
        self._near = [torch.empty(n, device=self.device) for _ in range(2)] if pin else None
        self._copy_stream = torch.cuda.Stream(self.device) if pin else None
        self._arrived = [None, None]
        self._dev = torch.empty(n, device=self.device)
        self._done = [torch.cuda.Event() if pin else None for _ in range(2)]

    def draw_host(self, out, generator=None):
        """The recorded sequence of draws, in order, into the flat host tensor `out` (the numbers the lazy route would see).
        Neighbouring tensors whose sizes are multiples of 16 are drawn with ONE `normal_()` over their common slice: ATen fills a
        float tensor with uniforms in order and turns them into normals 16 at a time, so the concatenation of such tensors gets
        exactly the numbers the separate calls would (checked once per plan against the separate calls, `_plan_runs`); 52 calls
        per step become 2, ~10 % of the draw time."""
        if self._runs is None:
            self._plan_runs()
        if self._pieces:
            return self._draw_pieces(out, generator)
        for lo, hi in self._runs:
            out[lo:hi].normal_(0, 1, generator=generator)
        return out

    # ---- the same draw on several host threads ------------------------------------------------------------------------------
    def _plan_pieces(self):
        """Cut the runs into `threads` pieces of (nearly) equal engine consumption.  A run of n elements consumes n outputs (+ 16 when
        n % 16 != 0: ATen redoes the last 16 elements with fresh uniforms); a run may be cut at any multiple of 16 as long as what
        follows the cut keeps >= 16 elements (below that normal_() takes the scalar double-precision path)."""
        self._pieces = None
        if self.threads < 2 or any(hi - lo < 16 for lo, hi in self._runs):
            return
        cons = [(hi - lo) + (16 if (hi - lo) % 16 else 0) for lo, hi in self._runs]
        total = sum(cons)
        if total < 64 * 1024:                       # a small draw is faster on one thread than through the pool
            return
        target = -(-total // self.threads)
        pieces, cur, room = [], [], target          # piece = [(lo, hi, consumption)]
        for lo, hi in self._runs:
            while lo < hi:
                n = hi - lo
                whole = n + (16 if n % 16 else 0)
                if whole <= room or len(pieces) == self.threads - 1:      # fits, or this is the last piece: it takes what is left
                    cur.append((lo, hi, whole)); room -= whole; lo = hi
                    continue
                cut = min(room // 16 * 16, (n - 16) // 16 * 16)           # the head that still fits; >= 16 elements stay behind
                if cut >= 16:
                    cur.append((lo, lo + cut, cut)); lo += cut
                elif not cur:
                    cur.append((lo, hi, whole)); lo = hi                  # cannot be cut: goes whole
                pieces.append(cur); cur, room = [], target
        if cur:
            pieces.append(cur)
        pieces = [p for p in pieces if p]
        if len(pieces) < 2:
            return
        self._pieces = [([(lo, hi) for lo, hi, _ in p], sum(c for _, _, c in p)) for p in pieces]
        self._pool = ThreadPoolExecutor(max_workers=len(self._pieces) - 1, thread_name_prefix="mlhot-eps-piece")
        self._gens = [torch.Generator() for _ in self._pieces]
        # once per plan: the pieces give the numbers AND the final generator state of the one-thread draw, or they are not used.
        # Checked where training runs them: from an engine in the MIDDLE of a 624-word block (1000 normals = 1016 outputs drawn
        # first), over two consecutive steps (the second starts wherever the first ended; several block regenerations per piece)
        g = torch.Generator()
        g.manual_seed(0x5eed + 1)
        torch.empty(1000).normal_(generator=g)
        keep, ok = self._pieces, True
        try:
            for _ in range(2):
                state = g.get_state()
                self._pieces = None
                ref = self.draw_host(torch.zeros(self._total), g)
                after = g.get_state()
                g.set_state(state)
                self._pieces = keep
                ok = ok and torch.equal(ref, self.draw_host(torch.zeros(self._total), g)) and torch.equal(after, g.get_state())
        except (ImportError, OSError, MlhotError):   # no library to position the generators with: the one-thread draw needs nothing
            ok = False
        self._pieces = keep if ok else None

    def _draw_pieces(self, out, generator=None):
        from mlhot import lib
        from mlhot.rng import _pack, _unpack
        g = generator if generator is not None else torch.default_generator
        s0 = g.get_state()
        engine = _unpack(s0).copy()
        L = lib()
        for (_, cons), gk in zip(self._pieces, self._gens):
            gk.set_state(_pack(s0, engine))
            L.mt19937_advance(engine, cons)

        def one(k):
            runs, _ = self._pieces[k]
            for lo, hi in runs:
                out[lo:hi].normal_(0, 1, generator=self._gens[k])

        jobs = [self._pool.submit(one, k) for k in range(1, len(self._pieces))]
        one(0)
        for j in jobs:
            j.result()
        g.set_state(_pack(s0, engine))
        return out

    def _plan_runs(self):
        sizes = [int(torch.Size(s).numel()) for s in self.shapes]
        single = [(o, o + n) for o, n in zip(self._offsets, sizes)]
        merged = []
        for (lo, hi), n in zip(single, sizes):
            if merged and merged[-1][1] == lo and n % 16 == 0 and (merged[-1][1] - merged[-1][0]) % 16 == 0:
                merged[-1] = (merged[-1][0], hi)
            else:
                merged.append((lo, hi))
        self._runs = single
        if len(merged) < len(single):
            g = torch.Generator()
            g.manual_seed(0x5eed)
            state = g.get_state()
            ref = self.draw_host(torch.zeros(self._total), g)
            g.set_state(state)
            self._runs = merged
            if not torch.equal(ref, self.draw_host(torch.zeros(self._total), g)):
                self._runs = single                 # this torch build fills differently: keep one call per tensor
        self._plan_pieces()

    def prefetch(self):
        """Start drawing the NEXT stage()'s eps on a worker thread (into the pinned buffer that stage() will ship)."""
        if not self.shapes:
            raise RuntimeError("StagedEps.prefetch(): nothing recorded; run one step under recording() first")
        if self._worker is not None:
            raise RuntimeError("StagedEps.prefetch(): the previous prefetch has not been collected by stage() yet")
        if self._offsets is None:
            self._plan()
        k = self._turn ^ 1
        if self._thread is None:                   # one long-lived drawer: starting a thread per step costs ~0.1 ms of the step
            self._go, self._ready = threading.Event(), threading.Event()
            self._thread = threading.Thread(target=self._draw_loop, name="mlhot-eps-draw", daemon=True)
            self._thread.start()
        self._job = k
        self._ready.clear()
        self._worker = True
        self._go.set()

    def _draw_loop(self):
        while True:
            self._go.wait()
            self._go.clear()
            k = self._job
            try:
                if self._done[k] is not None:
                    with torch.cuda.device(self.device):
                        self._done[k].synchronize()    # the copies out of this pair of buffers two steps ago
                        if self._arrived[k] is not None:
                            self._arrived[k].synchronize()
                self.draw_host(self._host[k])
                if self._near is not None:
                    with torch.cuda.device(self.device), torch.cuda.stream(self._copy_stream):
                        self._near[k].copy_(self._host[k], non_blocking=True)
                        self._arrived[k] = self._copy_stream.record_event()
                self._state_after = torch.get_rng_state()      # stage() checks that nobody else drew in the meantime
            except BaseException as e:             # noqa: BLE001 - surfaced by the collecting stage()
                self._worker_err = e
            self._ready.set()

    # ---- source = "device" ------------------------------------------------------------------------------------------
    def _device_issue(self):
        """Produce the NEXT stage()'s numbers on the side stream (behind the copy that emptied the buffer)."""
        with torch.cuda.stream(self._dn_stream):
            if self._dn_free is not None:
                self._dn_stream.wait_event(self._dn_free)
            self._dn_before = self._dn._engine.clone()      # the engine as of the numbers handed out so far (release())
            self._dn.draw(out=self._dn_buf)
            self._dn_ready = self._dn_stream.record_event()

    def _device_stage(self):
        if self._dn is None:
            from mlhot.rng import DeviceNormal
            sizes = [int(torch.Size(s).numel()) for s in self.shapes]
            self._dn = DeviceNormal(self.device, sizes)
            if self._dn.offsets != self._offsets or self._dn.total != self._total:
                raise RuntimeError("StagedEps: device and host plans disagree")
            with torch.cuda.device(self.device):
                self._dn_stream = torch.cuda.Stream(self.device)
                self._dn_buf = torch.empty(self._total, device=self.device)
            self._dn.take_over()                       # from here on the device owns the CPU generator's stream
            self._dn_stream.wait_stream(torch.cuda.current_stream(self.device))
            self._device_issue()
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(self._dn_ready)
        self._dev.copy_(self._dn_buf)
        self._dn_free = cur.record_event()
        self._device_issue()                           # step k+1's numbers, under step k
        self._cursor = 0

    def release(self):
        """source = "device": hand the engine back to the torch CPU generator: it continues exactly where a host-only run would be
        after the stage() calls made so far (the one look-ahead draw that is already on the device is given up - the engine is
        handed back as it was before it)."""
        if self._dn is None:
            return
        self._dn_stream.synchronize()
        self._dn._engine = self._dn_before
        self._dn.hand_back()
        self._dn = None

    def stage(self):
        """The next forward's eps on the device: draws them on the CPU generator (or collects the draws a prefetch() made
        meanwhile) and starts the copy (current stream)."""
        if not self.shapes:
            raise RuntimeError("StagedEps.stage(): nothing recorded; run one step under recording() first")
        if self._offsets is None:
            self._plan()
        if self.source == "device":
            return self._device_stage()
        k = self._turn = self._turn ^ 1
        prefetched = self._worker is not None
        if self._worker is not None:
            self._ready.wait()
            self._worker = None
            if self._worker_err is not None:
                err, self._worker_err = self._worker_err, None
                raise err
            if not torch.equal(self._state_after, torch.get_rng_state()):
                raise RuntimeError("StagedEps: the torch CPU generator was used between prefetch() and stage() (a loader, an init, CPU "
                                   "dropout ...): the eps draw order of the reference is broken - draw without prefetch(), or keep other "
                                   "consumers off the default generator while a prefetch is pending")
        else:
            if self._done[k] is not None:
                self._done[k].synchronize()        # the copy out of this pinned buffer two steps ago
            self.draw_host(self._host[k])
        if prefetched and self._near is not None and self._arrived[k] is not None:
            torch.cuda.current_stream(self.device).wait_event(self._arrived[k])
            self._dev.copy_(self._near[k])             # the drawer thread already sent the numbers across
        else:
            self._dev.copy_(self._host[k], non_blocking=True)
        if self._done[k] is not None:
            self._done[k].record()
        self._cursor = 0

    def rewind(self):
        """Start handing out the staged buffer from its first draw again (a step function that runs more than once per
        stage(), e.g. warm-up iterations before a capture, calls this first)."""
        self._cursor = 0

    def _next(self, size, device):
        i = self._cursor
        if i >= len(self.shapes) or self.shapes[i] != size:
            raise RuntimeError(f"StagedEps: draw {i} asks for {size}, recorded {self.shapes[i] if i < len(self.shapes) else None}")
        self._cursor += 1
        o = self._offsets[i]
        return self._dev[o:o + int(torch.Size(size).numel())].view(size)
