"""Task-parallel training across the GPUs of one node.

Tasks of a meta-batch are independent, so each rank (one process per GPU) owns a contiguous
slice of the tasks and builds its model with the LOCAL tasks_per_batch.  The only exchange is
one sum all-reduce of a single flat fp32 gradient bucket per step (RCCL over xGMI on the GPU
box, gloo in the CPU tests); with equal shards mean_r(grad_r) equals the full-batch gradient
because every loss is a mean over (task, target).
"""
import collections
import logging
import os

import torch
import torch.distributed as dist


def force_collectives():
    """MLHOT_FORCE_COLLECTIVES=1: a world of ONE still initialises the process group and runs every collective of the data path
    (GradBucket's all-reduces incl. the early bucket and the communication stream, StabiliserExchange's exchanges) instead of
    taking the world == 1 shortcuts.  A test hook: it lets a 1-GPU box drive backend "nccl" - i.e. load librccl and order its
    kernels against the compute / communication streams - before an 8-GPU node ever does (tests/test_gpu_parity.py::
    test_rccl_world_of_one_*); results are unchanged (a sum over one rank)."""
    return os.environ.get("MLHOT_FORCE_COLLECTIVES") == "1"


def init_from_env(backend=None):
    """torchrun-style rendezvous (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force_collectives()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl" and not os.environ.get("MLHOT_ONE_DEVICE"):
            kw["device_id"] = torch.device("cuda", local)          # binds the communicator to this rank's GPU up front
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local, world


def task_slice(n_tasks, rank, world):
    """Contiguous, equal task shard of rank `rank`."""
    if n_tasks % world:
        raise ValueError(f"{n_tasks} tasks do not split evenly over {world} ranks")
    per = n_tasks // world
    return slice(rank * per, (rank + 1) * per)


class StabiliserExchange:
    """Strict sharded parity of the FAVOR+ key stabiliser (fast_attention.py:96-97: torch.max over the keys of the WHOLE batch).

    With the meta-batch sharded over ranks every rank's FAVOR+ pass sees only its own tasks' keys.  The staged entry points of
    include/mlhot.h stop where the rank-local scalar exists; this object is the collective in between, on the 4-float device
    block `x` (csrc/stab_xchg.h):
      forward(x):  x[0] <- MAX over ranks of x[0]; x[1] <- 1 on the lowest rank whose maximum equals it (it holds the arg-max key),
                   0 elsewhere - ONE collective on `world` floats, the rest is element arithmetic on the device (no host sync)
      backward(x): x[2] <- SUM over ranks of x[2] (the stabiliser's gradient; the owner routes it to its arg-max key).  The ranks'
                   losses are means over their own tasks and GradBucket averages the gradients, so the owner's contribution is
                   the plain sum: mean_r(sum_r' g_r' [r == owner]) = (1 / world) sum_r' g_r', the un-sharded gradient.
    Install with mlhot.ops.set_stabiliser_exchange(StabiliserExchange()) (ModelTrainer does, for config.strict_sharded_parity);
    without it each rank uses its own maximum (a <= 1e-6 effect on the loss)."""

    def __init__(self, group=None, dedicated_group=False):
        """`dedicated_group`: create a process group of its own for the scalar exchanges (every rank must construct the object
        at the same point).  ProcessGroupNCCL runs the collectives of one group on one internal stream, so on the default group
        the backward's 4-byte exchange would queue behind an early gradient bucket of several MB that is still in flight."""
        if dedicated_group and group is None and dist.is_initialized() and dist.get_world_size() > 1:
            group = dist.new_group()
        self.group = group
        self.calls = collections.deque(maxlen=64)     # the most recent "fwd" / "bwd", in issue order (tests, diagnostics)

    def _world(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def forward(self, x):
        self.calls.append("fwd")
        world = self._world()
        if world == 1 and not (force_collectives() and dist.is_initialized()):
            x[1] = 1.0
            return
        rank = dist.get_rank(self.group)
        # an all-gather spelled as a SUM all-reduce of a one-hot vector (exact: every other rank adds zeros; and the one form
        # every backend takes for device tensors)
        vals = torch.zeros(world, device=x.device, dtype=x.dtype)
        vals[rank] = x[0]
        dist.all_reduce(vals, op=dist.ReduceOp.SUM, group=self.group)
        g = vals.max()
        idx = torch.arange(world, device=x.device, dtype=torch.float32)
        owner = torch.where(vals == g, idx, torch.full_like(idx, float(world))).min()
        x[0] = g
        x[1] = (owner == rank).to(x.dtype)

    def backward(self, x):
        self.calls.append("bwd")
        if self._world() > 1 or (force_collectives() and dist.is_initialized()):
            s = x[2:3].clone()
            dist.all_reduce(s, op=dist.ReduceOp.SUM, group=self.group)
            x[2:3] = s


class GradBucket:
    """One flat fp32 bucket for every parameter that receives a gradient.

    sync(): [pack ->] all_reduce(SUM) [-> unpack], i.e. exactly one collective per step (1.96 MB for ANPShapeNet1D), and the
    1/world average - either applied here, or (`defer_scale=True`) left to the caller, who folds the returned factor into the
    optimizer's gradient scale (mlhot.optim.FlatAdam.step(grad_scale=...)): one elementwise pass over the bucket less.
    Parameters whose grad is None on this step (e.g. the latent path with an empty context) must be None on every rank; they
    are skipped.

    `side_stream=True` issues the collectives on a communication stream of their own, ordered behind the producing kernels by an
    event.  The compute stream does NOT wait inside sync(wait=False): the caller enqueues whatever does not read the gradients
    (the next batch's ingest kernel, a validation forward) and calls wait() right before the first reader (the optimizer
    step) - only then do the RCCL kernels stop being in anybody's way.  sync() with the default wait=True is the plain,
    fully ordered form.

    `early=[params]`: a second bucket for parameters whose gradients are complete long before the backward ends - in the
    ResNet-family models everything except the image trunks (MLPs, attention, decoder head: 11.6 of ANPMRShapeNet3D's 15.1 MB),
    because the trunks' backward (one C call, ~1 ms) is the LAST node of the autograd graph.  arm() before backward() installs a
    one-shot countdown over those parameters' post-accumulate hooks; the hook of the last of them packs the early bucket and
    issues ITS all-reduce at once (asynchronously: comm stream on the GPU, async work handle on gloo), so it runs under the
    trunks' backward; sync() then reduces only the rest and joins both.  Without arm() (e.g. a hipGraph replay, where no
    autograd runs) sync() reduces everything in one collective as before."""

    def __init__(self, params, group=None, side_stream=False, early=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.flat = None
        self.side_stream = side_stream
        self._comm = None
        self._pending = []                   # async work handles / the comm stream's event still to be joined
        self.early = [p for p in (early or []) if p.requires_grad]
        self._early_ids = {id(p) for p in self.early}
        self._hooks, self._armed, self._left, self._early_state = None, False, 0, None
        self._warned_prune = False
        self.issue_log = []                  # ("early" | "rest" | "all", n_elements) per collective of the last step (tests, diagnostics)

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

    def _single(self):
        """A world of one without the force_collectives() test hook: nothing to exchange."""
        return self.world_size() == 1 and not (force_collectives() and dist.is_initialized())

    # ---- the collective -------------------------------------------------------------------------------------------------
    def _all_reduce(self, flat):
        """SUM over the ranks, in place (the one exchange of the data path; tests override it to run without a process group)."""
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)

    def _issue(self, flat, asynchronous=False):
        """The collective on `flat`: on the communication stream (side_stream, GPU), as an async work handle (asynchronous, gloo)
        or in line; wait() joins whatever is still pending."""
        if self.side_stream and flat.is_cuda:
            if self._comm is None:
                self._comm = torch.cuda.Stream(flat.device)
            cur = torch.cuda.current_stream(flat.device)
            self._comm.wait_stream(cur)                      # behind the kernels that filled the bucket
            with torch.cuda.stream(self._comm):
                self._all_reduce(flat)
            flat.record_stream(self._comm)
            self._pending.append((self._comm.record_event(), flat.device))
        elif asynchronous and type(self)._all_reduce is GradBucket._all_reduce and not flat.is_cuda:
            self._pending.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self._all_reduce(flat)

    def wait(self):
        """Join every collective issued since the last wait(): the current stream (GPU) / the host (gloo) may read the buckets."""
        pending, self._pending = self._pending, []
        for h in pending:
            if isinstance(h, tuple):                          # (event of the communication stream, the bucket's device)
                torch.cuda.current_stream(h[1]).wait_event(h[0])
            elif h is not None:
                h.wait()

    # ---- early bucket ---------------------------------------------------------------------------------------------------
    def arm(self):
        """Call before backward() of a step whose gradients sync() will reduce: the early bucket goes out from inside the backward."""
        self.issue_log = []
        if not self.early or self._single():
            return
        if self._hooks is None:
            self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.early]
        self._armed, self._left, self._early_state = True, len(self.early), None

    def _on_grad(self, p):
        if not self._armed:
            return
        self._left -= 1
        if self._left == 0:
            self._armed = False
            self._issue_early_now()

    def _issue_early_now(self):
        live = [q for q in self.early if q.grad is not None]
        if live:
            flat, views = self._shared_flat(live), None      # a gradient arena (mlhot/arena.py): reduce the range in place
            if flat is not None and flat.numel() > sum(q.grad.numel() for q in live) + 4 * len(live) + sum(
                    q.numel() for q in self.early if q.grad is None):
                flat = None                                  # the range holds more than the early gradients (+ padding, unused ones): not theirs alone
            if flat is None:
                flat = torch.empty(sum(q.grad.numel() for q in live), dtype=torch.float32, device=live[0].grad.device)
                views = list(flat.split([q.grad.numel() for q in live]))
                torch._foreach_copy_(views, [q.grad.reshape(-1) for q in live])
            self.issue_log.append(("early", flat.numel()))
            self._issue(flat, asynchronous=True)
            self._early_state = (live, flat, views)

    def issue_early(self):
        """The early bucket's all-reduce, issued by the CALLER at the point where those gradients are complete - for steps that run
        no autograd hooks: a step replayed as two hipGraphs (everything down to the image trunks' inputs | the trunks' backward,
        backward_in_two below) calls this between the two replays, so the collective runs under the second graph; sync() then
        reduces the rest and joins both, exactly as after an armed eager backward."""
        self.issue_log = []
        self._armed = False
        if not self.early or self._single():
            return
        self._issue_early_now()

    # ---- per step -------------------------------------------------------------------------------------------------------
    def sync(self, defer_scale=False, wait=True):
        """Returns the factor the caller still has to apply to the gradients (1.0 unless defer_scale).  wait=False: the
        collectives are only issued (side stream / async handles); call wait() before the gradients are read."""
        if not wait and not defer_scale:      # before anything is issued: a rank that raises must not leave the others inside a collective
            raise ValueError("GradBucket.sync(wait=False) needs defer_scale=True (the average would read the bucket)")
        world = self.world_size()
        live = [p for p in self.params if p.grad is not None]
        if not self.issue_log or self.issue_log[0][0] != "early":
            self.issue_log = []
        if self._armed:
            # The backward never reached every early parameter (modules a flavour / step does not use: transform_y, the latent
            # heads, FCL-only paths): nothing went out early, everything is reduced below.  From the next step on the early
            # bucket counts only parameters that DID receive a gradient - grad is None on every rank alike (class docstring), so
            # the ranks prune identically - and the silent fallback is logged once.
            self._armed = False
            missing = [p for p in self.early if p.grad is None]
            if missing and len(missing) < len(self.early):
                if not self._warned_prune:
                    self._warned_prune = True
                    logging.getLogger("mlhot.dist").warning(
                        "GradBucket: %d of %d early parameters received no gradient; the early bucket did not fire this step and "
                        "is re-planned without them", len(missing), len(self.early))
                gone = {id(p) for p in missing}
                self.early = [p for p in self.early if id(p) not in gone]
                self._early_ids = {id(p) for p in self.early}
                if self._hooks is not None:
                    for h in self._hooks:
                        h.remove()
                    self._hooks = None
        if self._single() or not live:
            return 1.0
        scale = 1.0 / world
        early, self._early_state = self._early_state, None
        if early is not None:
            done = {id(q) for q in early[0]}
            live = [p for p in live if id(p) not in done]
        rest_flat, rest_views = None, None
        if live:
            # Gradients that are views of ONE flat buffer (the vanilla models' single backward call writes them so,
            # mlhot_np_grads_flat_layout; the ResNet / BBB family through mlhot.arena.GradArena) are reduced in place: no pack /
            # unpack.  Alignment padding and regions of unused parameters ride along harmlessly.
            rest_flat = self._shared_flat(live)
            if rest_flat is not None and early is not None:
                # the early bucket has been (or is being) reduced already: an in-place range for the rest must not reach into it
                # (it cannot when the arena was built with first=<the early parameters>; any other layout packs the rest)
                a0, a1 = rest_flat.data_ptr(), rest_flat.data_ptr() + 4 * rest_flat.numel()
                for q in early[0]:
                    if q.grad.data_ptr() < a1 and q.grad.data_ptr() + 4 * q.grad.numel() > a0:
                        rest_flat = None
                        break
            if rest_flat is None:
                n = sum(p.grad.numel() for p in live)
                if self.flat is None or self.flat.numel() != n or self.flat.device != live[0].grad.device:
                    self.flat = torch.empty(n, dtype=torch.float32, device=live[0].grad.device)
                rest_flat, rest_views = self.flat, list(self.flat.split([p.grad.numel() for p in live]))
                torch._foreach_copy_(rest_views, [p.grad.reshape(-1) for p in live])
            self.issue_log.append(("rest" if early is not None else "all", rest_flat.numel()))
            self._issue(rest_flat, asynchronous=not wait)
        self._unpack = (live, rest_flat, rest_views, early)
        if wait:
            self.wait()
        return self._finish(scale, defer_scale) if wait else scale

    def _finish(self, scale, defer_scale):
        live, rest_flat, rest_views, early = self._unpack
        self._unpack = None
        if not defer_scale:
            if rest_flat is not None:
                rest_flat.mul_(scale)
            if early is not None:
                early[1].mul_(scale)
        if rest_views is not None:
            torch._foreach_copy_([p.grad.view(-1) for p in live], rest_views)
        if early is not None and early[2] is not None:
            torch._foreach_copy_([q.grad.view(-1) for q in early[0]], early[2])
        return scale if defer_scale else 1.0

    def finish(self):
        """After sync(defer_scale=True, wait=False): join the collectives and copy the packed buckets back into the .grad tensors."""
        self.wait()
        if getattr(self, "_unpack", None) is not None:
            self._finish(1.0, True)


def backward_in_two(loss, model, gradient=None, between=None):
    """loss.backward() of a model whose forward cut its autograd graph in front of the image trunks (ResNetNP.enable_split_backward):
    part 1 runs every node above the cut - all of GradBucket's early parameters have their gradients then -, `between()` is called
    (e.g. bucket.issue_early()), part 2 runs the trunks' backward (and what hangs below it: the Bayes-by-backprop sampling) from
    the gradients part 1 left at the cut.  Same gradients as the one-piece backward (tests/test_gpu_parity.py).  The two parts can
    be captured as two hipGraphs (bench.py): the all-reduce issued in between overlaps the second."""
    pairs = model.__dict__.get("_cut_pairs")
    if not pairs:
        raise ValueError("backward_in_two: the model's last forward recorded no cut (enable_split_backward(True) before the forward)")
    loss.backward(gradient=gradient)
    if between is not None:
        between()
    run_second_part(model)


def run_second_part(model):
    pairs = model.__dict__.get("_cut_pairs") or []
    live = [(o, leaf.grad) for o, leaf in pairs if leaf.grad is not None]
    model.__dict__["_cut_pairs"] = []
    if live:
        torch.autograd.backward([o for o, _ in live], [g for _, g in live])


def rank():
    return dist.get_rank() if dist.is_initialized() else 0
