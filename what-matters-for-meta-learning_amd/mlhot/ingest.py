"""Batch ingest: host uint8 channel-last images -> device fp32 channel-first tensors.

The reference's loaders convert on the host (dataset/shapenet_1d.py:189-196: `astype(float32) / 255.0`, then
utils/utils.py:26-30 permute) and the trainer copies pageable fp32 tensors with `.to(device)`
(trainer/model_trainer.py:67-70): 31.5 MB per 16-task ShapeNet1D batch, more than twice the GPU step time on PCIe.
Here a batch crosses the bus as uint8 (7.9 MB) from pinned staging on a copy stream while the previous step computes,
and `mlhot_ingest_u8_nhwc` does the divide + permute on the device (bit-identical to the host arithmetic).

    ing = BatchIngest(device)
    ing.stage(xs_u8, xq_u8, ys, yq)          # host arrays of batch k+1: returns at once (async H2D)
    ... run step k ...
    ctx_x, qry_x, ctx_y, qry_y = ing.take()   # fp32 [T,N,C,H,W] on the device, ordered on the current stream

`take()` writes into the same device tensors for every batch of the same shape, so a captured hipGraph of the step keeps
reading valid addresses.  There is no CPU fallback: the device must be a ROCm GPU.
"""
import collections
import threading

import numpy as np
import torch

from . import lib
from .binding import MlhotError


def _host(a, dtype):
    t = torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a.contiguous()
    if t.dtype != dtype:
        raise MlhotError(f"BatchIngest: expected {dtype}, got {t.dtype}")
    return t


def _layout(key):
    """Byte layout of one packed staging buffer: [ctx images | qry images | pad to 16 | ctx labels | qry labels]."""
    n_img = [int(np.prod(key[0])), int(np.prod(key[1]))]
    n_lab = [int(np.prod(key[2])), int(np.prod(key[3]))]
    lab_off = (n_img[0] + n_img[1] + 15) // 16 * 16
    return n_img, n_lab, lab_off, lab_off + 4 * (n_lab[0] + n_lab[1])


class _Slot:
    """One staging slot: ONE pinned host buffer + its device twin holding a whole batch (images as bytes, labels as fp32),
    so a batch is one H2D copy."""

    def __init__(self, key, device):
        n_img, n_lab, lab_off, total = _layout(key)
        self.host = torch.empty(max(total, 16), dtype=torch.uint8).pin_memory()
        self.dev = torch.empty(max(total, 16), dtype=torch.uint8, device=device)
        hn = self.host.numpy()
        self.host_np = [hn[:n_img[0]].reshape(key[0]), hn[n_img[0]:n_img[0] + n_img[1]].reshape(key[1]),
                        hn[lab_off:lab_off + 4 * n_lab[0]].view(np.float32).reshape(key[2]),
                        hn[lab_off + 4 * n_lab[0]:total].view(np.float32).reshape(key[3])]
        self.dev_img = self.dev[:n_img[0] + n_img[1]]
        self.dev_lab = self.dev[lab_off:total].view(torch.float32)
        self.copied = torch.cuda.Event()       # H2D of this slot finished (host buffer reusable, device buffer readable)
        self.consumed = torch.cuda.Event()     # the kernels that read this slot's device buffer finished
        self.busy = False


class _Out:
    """The fixed fp32 tensors batches of one shape are delivered in: images of both sets in one flat buffer (one ingest
    launch when the image geometry is shared), labels in another (one device copy)."""

    def __init__(self, key, device):
        (T, Nc, H, W, Cc), (_, Nq, H2, W2, C2) = key[0], key[1]
        n_img, n_lab, _, _ = _layout(key)
        self.same_geometry = (H, W, Cc) == (H2, W2, C2)
        self.img = torch.empty(n_img[0] + n_img[1], device=device)
        self.lab = torch.empty(n_lab[0] + n_lab[1], device=device)
        self.n_img = n_img
        self.tensors = (self.img[:n_img[0]].view(T, Nc, Cc, H, W), self.img[n_img[0]:].view(T, Nq, C2, H2, W2),
                        self.lab[:n_lab[0]].view(key[2]), self.lab[n_lab[0]:].view(key[3]))


class BatchIngest:
    def __init__(self, device, slots=2, div=255.0):
        device = torch.device(device)
        if device.type != "cuda":
            raise MlhotError("BatchIngest: the ingest path needs a ROCm device; there is no CPU fallback")
        self.device, self.n_slots, self.div = device, slots, div
        self.copy_stream = torch.cuda.Stream(device)
        self._slots = {}                        # shapes -> [slot, ...]
        self._out = {}                          # shapes -> _Out (fixed fp32 outputs)
        self._queue = collections.deque()
        # stage*() may run on a worker thread while the owner take()s an earlier batch (trainer._HostPrefetch, two batches drawn ahead):
        # slot choice and the queue are guarded; the fill and the copy of a reserved slot are not (they touch only that slot).
        self._lock = threading.Lock()

    def _free_slot(self, key):
        with self._lock:
            ring = self._slots.setdefault(key, [])
            slot = next((sl for sl in ring if not sl.busy), None)
            if slot is None:
                if len(ring) >= self.n_slots:
                    raise MlhotError("BatchIngest: more batches staged than slots; call take() first")
                slot = _Slot(key, self.device)
                ring.append(slot)
            slot.busy = True                    # reserved from here on (given back by take(), or by a fill that refuses the batch)
        slot.copied.synchronize()               # the previous H2D out of this pinned buffer is done (no-op when fresh)
        return slot

    def stage(self, xs_u8, xq_u8, ys, yq):
        """Queue one host batch: images uint8 [T,N,H,W,C] (channel-last), labels fp32 [T,N,L].  Returns a ticket."""
        src = [_host(xs_u8, torch.uint8), _host(xq_u8, torch.uint8), _host(ys, torch.float32), _host(yq, torch.float32)]
        if src[0].dim() != 5 or src[1].dim() != 5:
            raise MlhotError("BatchIngest: images must be [T, N, H, W, C]")
        key = tuple(tuple(t.shape) for t in src)
        slot = self._free_slot(key)
        for h, t in zip(slot.host_np, src):
            np.copyto(h, t.numpy())             # one thread on purpose: torch's copy_ wakes the whole OpenMP pool, whose
                                                # spinning workers then starve the HIP runtime's helper threads
        return self._ship(key, slot)

    def stage_filled(self, key, fill):
        """Queue a batch whose bytes the CALLER writes into the pinned staging buffers: `fill(host_np)` gets the slot's four numpy views
        ([ctx images u8 | qry images u8 | ctx labels f32 | qry labels f32], shaped like `key`) and returns True to ship the batch or
        False to give the slot back (nothing is queued; returns None)."""
        slot = self._free_slot(key)
        try:
            ok = fill(slot.host_np)
        except BaseException:
            slot.busy = False
            raise
        if not ok:
            slot.busy = False
            return None
        return self._ship(key, slot)

    def _ship(self, key, slot):
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(slot.consumed)      # do not overwrite bytes an ingest kernel still reads
            slot.dev.copy_(slot.host, non_blocking=True)
            slot.copied.record(self.copy_stream)
        with self._lock:
            self._queue.append((key, slot))
        return slot

    def device_views(self, ticket=None):
        """(uint8 [n, H, W, C] staged images on the device, fp32 [n, C, H, W] destination) of a staged batch whose two
        image sets share their geometry - what take() hands to mlhot_ingest_u8_nhwc; for profiling that kernel alone."""
        key, slot = self._queue[0] if ticket is None else next(e for e in self._queue if e[1] is ticket)
        out = self._out.setdefault(key, _Out(key, self.device))
        if not out.same_geometry:
            raise MlhotError("device_views: context and target images differ in geometry")
        _, _, H, W, Cc = key[0]
        return slot.dev_img.view(-1, H, W, Cc), out.img.view(-1, Cc, H, W)

    def take(self, ticket=None):
        """A staged batch (the oldest, or the one `ticket` names) as (ctx_x, qry_x, ctx_y, qry_y): fp32, channel-first,
        valid on the current stream.  Batches of one shape share their output tensors: use a batch before taking the next."""
        with self._lock:
            if not self._queue:
                raise MlhotError("BatchIngest.take() without a staged batch")
            if ticket is None:
                key, slot = self._queue.popleft()
            else:
                hit = [e for e in self._queue if e[1] is ticket]
                if not hit:
                    raise MlhotError("BatchIngest.take(): unknown or already taken ticket")
                key, slot = hit[0]
                self._queue.remove(hit[0])
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(slot.copied)
        out = self._out.get(key)
        if out is None:
            out = self._out[key] = _Out(key, self.device)
        L = lib()
        with torch.cuda.device(self.device):
            if out.same_geometry:               # both image sets are one packed run of (H, W, C) images
                _, _, H, W, Cc = key[0]
                L.ingest_u8_nhwc(slot.dev_img.view(-1, H, W, Cc), out=out.img.view(-1, Cc, H, W), div=self.div)
            else:
                n0 = out.n_img[0]
                L.ingest_u8_nhwc(slot.dev_img[:n0].view(key[0]), out=out.tensors[0], div=self.div)
                L.ingest_u8_nhwc(slot.dev_img[n0:].view(key[1]), out=out.tensors[1], div=self.div)
            out.lab.copy_(slot.dev_lab)
        slot.consumed.record(cur)
        slot.busy = False
        return out.tensors


class ExactU8Feed:
    """fp32 host batches of a reference-style loader (dataset/shapenet_1d.py:189-196 -> utils/utils.py:26-30: `img.astype(float32) /
    255.0`, channel-first) across PCIe as BYTES when - and only when - every image element is exactly k / 255 for a byte k
    (mlhot_host_f32_to_u8_exact checks all of them while it converts; K native host threads, one pass): a quarter of the traffic, and the ingest
    kernel's `(float)k / 255` on the device gives the loader's fp32 values back bit for bit.  A batch with ANY other value (an
    augmentation that blends pixels, a loader that normalises differently) is refused - stage() returns None and the caller ships the
    fp32 tensors as before; after `give_up` refusals in a row the check is not attempted any more.

        feed = ExactU8Feed(device)
        ticket = feed.stage((ctx_x, qry_x, ctx_y, qry_y))     # fp32 host tensors [T, N, C, H, W] / [T, N, L]; None = not byte images
        ctx_x, qry_x, ctx_y, qry_y = feed.take(ticket)        # fp32 device tensors (fixed addresses per batch shape)
    """

    def __init__(self, device, threads=None, div=255.0, give_up=3, slots=3):
        self.ing = BatchIngest(device, slots=slots, div=div)      # three: two batches drawn ahead (trainer, host_prefetch_depth 2) + one on the spot
        self.div, self.give_up = float(div), int(give_up)
        self.threads = default_feed_threads() if threads is None else max(1, min(64, int(threads)))
        self.ok, self.refused_in_a_row = True, 0
        self.shipped, self.refused = 0, 0

    def stage(self, host_batch):
        if not self.ok:
            return None
        xs, xq, ys, yq = host_batch
        for t in (xs, xq, ys, yq):
            if not (torch.is_tensor(t) and t.device.type == "cpu" and t.dtype == torch.float32 and t.is_contiguous()):
                return self._refuse()
        if xs.dim() != 5 or xq.dim() != 5:
            return self._refuse()
        # channel-first bytes are ingested as one-channel images: [T, N * C, H, W, 1] -> [T, N * C, 1, H, W] = the fp32 layout itself
        (T, Nc, C, H, W), (_, Nq, C2, H2, W2) = xs.shape, xq.shape
        key = ((T, Nc * C, H, W, 1), (T, Nq * C2, H2, W2, 1), tuple(ys.shape), tuple(yq.shape))
        L = lib()

        def fill(host_np):
            for src, dst in ((xs, host_np[0]), (xq, host_np[1])):
                if L.host_f32_to_u8_exact(src.data_ptr(), dst.ctypes.data, src.numel(), self.div, threads=self.threads):
                    return False
            np.copyto(host_np[2], ys.numpy())
            np.copyto(host_np[3], yq.numpy())
            return True

        slot = self.ing.stage_filled(key, fill)
        if slot is None:
            return self._refuse()
        self.refused_in_a_row = 0
        self.shipped += 1
        return (slot, tuple(xs.shape), tuple(xq.shape))

    def _refuse(self):
        self.refused += 1
        self.refused_in_a_row += 1
        if self.refused_in_a_row >= self.give_up:
            self.ok = False                    # this loader does not hand out byte images: stop paying for the check
        return None

    def take(self, ticket):
        slot, sc, sq = ticket
        cx, qx, cy, qy = self.ing.take(slot)
        return cx.view(sc), qx.view(sq), cy, qy


def default_feed_threads():
    """Host threads of the byte conversion: MLHOT_FEED_THREADS, else min(4, usable cores // (2 x ranks on this node)), at least 1.
    Measured on the GPU box (EPYC 9575F, scripts/dev/u8_feed_probe.py, c3's 7.9 M floats): 1.23 / 0.62 / 0.34 / 0.30 / 0.47 / 0.86 ms on
    1 / 2 / 4 / 8 / 16 / 32 threads (the call starts its threads itself: beyond 8 the starts cost more than the pieces save); stage()
    as the trainer calls it - two image tensors, labels, the H2D issue - 0.43 ms with 4 threads, 0.54 with 8."""
    import os
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = os.cpu_count() or 1
    ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE") or os.environ.get("WORLD_SIZE") or 1))
    env = os.environ.get("MLHOT_FEED_THREADS")
    if env is not None:
        return max(1, min(int(env), max(1, cores // ranks)))
    return max(1, min(4, cores // (2 * ranks)))
