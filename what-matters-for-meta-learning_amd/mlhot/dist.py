"""Task-parallel training across the GPUs of one node.

Tasks of a meta-batch are independent, so each rank (one process per GPU) owns a contiguous
slice of the tasks and builds its model with the LOCAL tasks_per_batch.  The only exchange is
one sum all-reduce of a single flat fp32 gradient bucket per step (RCCL over xGMI on the GPU
box, gloo in the CPU tests); with equal shards mean_r(grad_r) equals the full-batch gradient
because every loss is a mean over (task, target).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """torchrun-style rendezvous (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend or ("nccl" if torch.cuda.is_available() else "gloo"), rank=rank, world_size=world)
    return rank, local, world


def task_slice(n_tasks, rank, world):
    """Contiguous, equal task shard of rank `rank`."""
    if n_tasks % world:
        raise ValueError(f"{n_tasks} tasks do not split evenly over {world} ranks")
    per = n_tasks // world
    return slice(rank * per, (rank + 1) * per)


class GradBucket:
    """One flat fp32 bucket for every parameter that receives a gradient.

    sync(): [pack ->] all_reduce(SUM) [-> unpack], i.e. exactly one collective per step (1.96 MB for ANPShapeNet1D), and the
    1/world average - either applied here, or (`defer_scale=True`) left to the caller, who folds the returned factor into the
    optimizer's gradient scale (mlhot.optim.FlatAdam.step(grad_scale=...)): one elementwise pass over the bucket less.
    Parameters whose grad is None on this step (e.g. the latent path with an empty context) must be None on every rank; they
    are skipped.

    `side_stream=True` issues the collective on a communication stream of its own (ordered behind the backward by an event, the
    consumer waits on another): the RCCL kernels do not queue in front of whatever the caller enqueues next on the compute
    stream (e.g. the next batch's ingest kernel or a validation forward)."""

    def __init__(self, params, group=None, side_stream=False):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.flat = None
        self.side_stream = side_stream
        self._comm = None

    @staticmethod
    def _shared_flat(live):
        """The one fp32 tensor all live gradients are views of, or None."""
        g0 = live[0].grad
        st = g0.untyped_storage()
        lo, hi = None, None
        for p in live:
            g = p.grad
            if g.dtype != torch.float32 or not g.is_contiguous() or g.untyped_storage().data_ptr() != st.data_ptr():
                return None
            a, b = g.storage_offset(), g.storage_offset() + g.numel()
            lo, hi = (a if lo is None else min(lo, a)), (b if hi is None else max(hi, b))
        if len(live) < 2:
            return None
        return torch.empty(0, dtype=torch.float32, device=g0.device).set_(st, lo, (hi - lo,))

    def world_size(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def _all_reduce(self, flat):
        """SUM over the ranks, in place (the one exchange of the data path; tests override it to run without a process group)."""
        if not (self.side_stream and flat.is_cuda):
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            return
        if self._comm is None:
            self._comm = torch.cuda.Stream(flat.device)
        cur = torch.cuda.current_stream(flat.device)
        self._comm.wait_stream(cur)                      # behind the backward that filled the bucket
        with torch.cuda.stream(self._comm):
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        cur.wait_stream(self._comm)                      # whoever reads the gradients next on the compute stream
        flat.record_stream(self._comm)

    def sync(self, defer_scale=False):
        """Returns the factor the caller still has to apply to the gradients (1.0 unless defer_scale)."""
        world = self.world_size()
        live = [p for p in self.params if p.grad is not None]
        if world == 1 or not live:
            return 1.0
        scale = 1.0 / world
        shared = self._shared_flat(live)
        if shared is not None:
            # the library already wrote every gradient into ONE flat buffer (mlhot_np_grads_flat_layout): reduce it in
            # place, no pack / unpack.  Alignment padding and regions of unused parameters ride along harmlessly.
            self._all_reduce(shared)
            if defer_scale:
                return scale
            shared.mul_(scale)
            return 1.0
        n = sum(p.grad.numel() for p in live)
        if self.flat is None or self.flat.numel() != n or self.flat.device != live[0].grad.device:
            self.flat = torch.empty(n, dtype=torch.float32, device=live[0].grad.device)
        views = list(self.flat.split([p.grad.numel() for p in live]))
        torch._foreach_copy_(views, [p.grad.reshape(-1) for p in live])
        self._all_reduce(self.flat)
        if not defer_scale:
            self.flat.mul_(scale)
        torch._foreach_copy_([p.grad.view(-1) for p in live], views)
        return scale if defer_scale else 1.0


def rank():
    return dist.get_rank() if dist.is_initialized() else 0
