"""CNP / ANP trainer with the reference's loop (trainer/model_trainer.py:33-143): per iteration one
meta-batch -> forward -> loss (+ beta * kl) -> backward -> optimizer step; periodic validation with
best-model saving; intermediate / final checkpoints (`state_dict` only, like the reference).

Differences that matter on MI355X: the loss value is fetched from the device once per iteration (one host
sync instead of three), with more than one rank the gradients are averaged by ONE flat-bucket
all-reduce (mlhot.dist.GradBucket) before the optimizer step, and a data source that can hand out its
batches before the host-side conversion (`get_batch_u8`: uint8 channel-last images) is read through
mlhot.ingest.BatchIngest: the next training batch crosses PCIe as uint8 while the current step computes.

The reference's own calling sequence is the fast one (round 5): `ModelTrainer(model, loss, torch.optim.Adam(model.parameters(),
lr=...), config, data).train()` as train.py:52-90 writes it, with a loader that hands out fp32 host batches, runs as
  * mlhot.optim.FlatAdam.from_torch_adam(...): a plain torch.optim.Adam over the model's parameters is continued by the
    one-launch flat update (same hyper-parameters, moments and step count; `config.promote_optimizer = False` keeps torch's),
  * `config.graph_steps` defaulting to True whenever the optimizer is capture-safe and the model has a flat parameter layout (the
    vanilla CNP / ANP plugins: the library's; the ResNet / Bayes-by-backprop family: ResNetNP.flat_layout, gradients through the
    mirror arena, the Bayes-by-backprop eps staged per step by networks/bbb/eps.py - drawn on host threads under the previous step),
  * the next host batch copied to the device on a copy stream while the current step computes (`_HostPrefetch`; the reference's
    pageable `.to(device)` - the fastest route for fp32 host tensors on this box, 50 GB/s - just not in front of the step).
`config.graph_steps` (needs an optimizer whose step is capture-safe, e.g.
mlhot.optim.FlatAdam(capturable=True)) replays every training iteration from a hipGraph: the eager host path of one
iteration (autograd bookkeeping, ~20 launches, the optimizer) costs about twice the GPU time of the step, the replay a few
tens of microseconds.  One graph per batch shape (the context size is drawn per iteration, dataset/shapenet_1d.py:120); the
first iteration of a shape runs eagerly and doubles as the warm-up, the second captures.  The loss is then fetched every
`config.log_every` iterations only (default 1 = the reference's per-iteration log and finiteness check).

`config.strict_sharded_parity = True` (off by default; attention models on more than one rank): the FAVOR+ key stabiliser is the
maximum over the keys of the WHOLE meta-batch as in the reference's single-process batch (fast_attention.py:96-97), not of the
rank's shard - one scalar all-gather in the forward and one scalar all-reduce in the backward
(mlhot.dist.StabiliserExchange, include/mlhot.h "strict sharded parity").  Eager iterations only: the exchange runs between
two C calls, so it cannot sit inside a replayed hipGraph.
"""
import collections
import contextlib
import math
import os
import sys

import torch

from mlhot.dist import GradBucket, rank as dist_rank
from mlhot.ops import add_scaled, loss_value_aside
from trainer.base_trainer import BaseTrainer


from mlhot.graphs import CAPTURE_MODE, capture as capture_graph      # thread_local error mode, collector paused: see mlhot/graphs.py

class _HostPrefetch:
    """fp32 host batches (the reference's loaders: dataset/shapenet_1d.py:189-196 -> utils/utils.py:26-30) to the device on a copy
    stream.  stage() copies - `.to(device)` from pageable memory blocks the HOST for the transfer (0.62 ms for c3's 31.5 MB), which
    is why the trainer calls it behind the step's launch - and take() orders the batch on the current stream.

    Round 6 (`config.host_u8`, default on): those loaders' images ARE bytes divided by 255, so a batch first goes through
    mlhot.ingest.ExactU8Feed - every element is checked to be exactly k / 255 while it is converted back to its byte (host threads, one
    pass) and the batch then crosses PCIe as 7.9 instead of 31.5 MB, expanded by the ingest kernel to the same fp32 bits.  A batch that
    holds anything else takes the fp32 route below, unchanged; the loader's contract is untouched either way."""

    def __init__(self, device, u8=True, background=True):
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(self.device)
        self.u8 = None
        self.last_fixed = False         # did the last take() hand out the byte route's fixed per-shape device tensors?
        if u8:
            from mlhot.ingest import ExactU8Feed
            self.u8 = ExactU8Feed(self.device)
        # Round 6: the copy itself runs on ONE worker thread.  Whichever route a batch takes, putting it on its way blocks the calling
        # thread for ~0.5 ms (the pageable fp32 copy: 0.60 ms; the byte conversion + its issue: 0.58 ms - measured inside this loop,
        # scripts/dev/trainer_iter_probe.py), and with the reference's `loss.item()` every iteration that time is SERIAL with the
        # iteration's other host work (~0.25 ms of Python around the replay): 0.87 ms per iteration for a 0.6 ms GPU step.  The worker
        # makes the hand-over a queue push; the loader itself (get_batch, possibly on a shared generator) stays on the caller's thread.
        self._pool = None
        if background:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="mlhot-host-batch")

    def _stage_now(self, host_batch):
        with torch.cuda.device(self.device):
            if self.u8 is not None:
                ticket = self.u8.stage(host_batch)
                if ticket is not None:
                    return ("u8", ticket)
            with torch.cuda.stream(self.stream):
                dev = tuple(t.to(self.device, non_blocking=True) for t in host_batch)
                ev = torch.cuda.Event()
                ev.record(self.stream)
            return ("f32", dev, ev)

    def stage(self, host_batch):
        if self._pool is not None:
            return ("later", self._pool.submit(self._stage_now, host_batch))
        return self._stage_now(host_batch)

    def take(self, ticket):
        if ticket[0] == "later":
            ticket = ticket[1].result()         # normally long done: the worker had a whole GPU step for it
        self.last_fixed = ticket[0] == "u8"
        if ticket[0] == "u8":
            return self.u8.take(ticket[1])      # fixed device tensors per batch shape, ordered on the current stream
        _, dev, ev = ticket
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        for t in dev:
            t.record_stream(cur)
        return dev


class ModelTrainer(BaseTrainer):
    def __init__(self, model, loss, optimizer, config, data):
        super().__init__(model=model, loss=loss, optimizer=optimizer, config=config)
        self.data = data
        # side stream: the RCCL kernels never sit in the compute queue; early bucket: models that know which gradients are complete
        # before the backward ends (ResNet family: everything but the image trunks) all-reduce them under the rest of the backward
        early = model.early_grad_parameters() if hasattr(model, "early_grad_parameters") else None
        self.bucket = GradBucket(model.parameters(), side_stream=torch.device(config.device).type == "cuda", early=early)
        if not self.bucket._single() and hasattr(model, "enable_flat_grads") and torch.device(config.device).type == "cuda":
            model.enable_flat_grads()      # more than one rank: the ResNet / BBB family's gradients in one flat buffer, reduced in place
        if getattr(config, "strict_sharded_parity", False):
            if getattr(config, "graph_steps", False):
                raise ValueError("config.strict_sharded_parity runs a collective between two C calls of the forward: not with config.graph_steps")
            from mlhot import ops
            from mlhot.dist import StabiliserExchange
            ops.set_stabiliser_exchange(StabiliserExchange(dedicated_group=True))
        cuda = torch.device(config.device).type == "cuda"
        # train.py:52-56's optimizer continued by the flat one-launch update
        if cuda and getattr(config, "promote_optimizer", True):
            from mlhot.optim import FlatAdam
            shot = int(getattr(config, "max_ctx_num", 15) or 15)
            flat = FlatAdam.from_torch_adam(self.optimizer, model, ctx_num=min(shot, 15), test_num=min(shot, 15))
            if flat is not None:
                # From here on `trainer.optimizer` IS the optimizer: checkpoint its state_dict() (torch.optim.Adam's layout), point an LR
                # scheduler at its param_groups[0]["lr"].  The caller's torch object is left as it was - it no longer steps anything.
                self.replaced_optimizer, self.optimizer = self.optimizer, flat
                if hasattr(model, "enable_flat_grads") and model.__dict__.get("_arena") is None:
                    model.enable_flat_grads()      # ResNet / BBB family: the gradients as the mirror of the flat parameter buffer
                    self._installed_arena = True
        if not hasattr(config, "graph_steps"):
            # default: replay whenever it is possible - a capture-safe optimizer, a model whose step is one static launch sequence
            # (the vanilla plugins), no collective between two C calls of the forward
            self._graph_default = (cuda and getattr(self.optimizer, "capturable", False) and hasattr(model, "flat_layout")
                                   and not getattr(config, "strict_sharded_parity", False) and not getattr(config, "contrastive", False))
        else:
            self._graph_default = bool(config.graph_steps)
        self.ingest, self._staged = None, None
        self._staged_q = collections.deque()      # host-batch route: tickets of the training batches drawn ahead, oldest first
        self._host_prefetch, self._fixed_batch = None, False
        self._prefetch = False          # set per iteration by train(): may the NEXT training batch be drawn right away?  (an int: how many)
        self.rank0 = dist_rank() == 0   # files / logs / TensorBoard are rank 0's business (every rank holds the same weights)
        self._graphs, self._static_in, self._side = {}, {}, None       # graph_steps: per batch shape
        self._eps = None                # graph_steps of a Bayes-by-backprop model: its eps draws staged per step (networks/bbb/eps.py)
        if hasattr(data, "get_batch_u8") and cuda and getattr(config, "ingest_u8", True):
            from mlhot.ingest import BatchIngest
            self.ingest = BatchIngest(config.device)
        elif cuda and getattr(config, "host_prefetch", True):
            self._host_prefetch = _HostPrefetch(config.device, u8=bool(getattr(config, "host_u8", True)),
                                                background=bool(getattr(config, "host_copy_thread", True)))

    def _announce(self):
        """Once, at the start of train(): what the constructor promoted (nothing here is silent)."""
        from mlhot.optim import FlatAdam
        if isinstance(self.optimizer, FlatAdam) and getattr(self, "replaced_optimizer", None) is not None:
            self._log("mlhot: torch.optim.Adam continued by mlhot.optim.FlatAdam (one launch over the flat parameter buffer; same hyper-parameters, "
                      "moments and step count).  Checkpoint / schedule `trainer.optimizer`; the optimizer object passed in no longer steps.")
        if self._graph_default:
            self._log("mlhot: training iterations are replayed from hipGraphs (one per batch shape); a change of trainer.optimizer.param_groups[0] "
                      "(lr, betas, eps, weight_decay) is picked up by re-capturing.")
        if self._host_prefetch is not None:
            self._log("mlhot: host batches are copied on a copy stream behind the step" +
                      (" - as bytes when every image element is exactly k / 255 (checked per batch), as fp32 otherwise" if self._host_prefetch.u8 is not None else "")
                      + f"; up to {max(1, int(getattr(self.config, 'host_prefetch_depth', 2)))} batches are drawn ahead where no validation round / background regeneration lies between")
        if self._graph_default and self._lagged():
            self._log("mlhot: every iteration's loss is logged and checked one iteration late (read behind the NEXT iteration's launch; flushed before "
                      "validation rounds, checkpoints and the end of training).  config.lagged_loss_log = False reads it right behind the step.")

    def close(self):
        """Undo the process-wide installs of the constructor (the gradient arena in mlhot.binding, the stabiliser exchange in mlhot.ops);
        train() calls it when it is done, a caller that only uses _train_iter calls it itself."""
        if getattr(self, "_installed_arena", False):
            from mlhot import binding
            if binding.get_grad_arena() is self.model.__dict__.get("_arena"):
                binding.set_grad_arena(None)
            self._installed_arena = False
        if getattr(self.config, "strict_sharded_parity", False):
            from mlhot import ops
            ops.set_stabiliser_exchange(None)
        hp = self._host_prefetch
        if hp is not None and hp._pool is not None:
            while self._staged_q:
                staged = self._staged_q.popleft()
                if staged[0] == "later":
                    staged[1].result()                  # nothing left in flight when the worker goes
            hp._pool.shutdown(wait=True)
            hp._pool = None                             # a later stage() copies on the caller's thread

    def _hyper(self):
        g = self.optimizer.param_groups[0]
        return (float(g["lr"]), tuple(g["betas"]), float(g["eps"]), float(g.get("weight_decay", 0.0)))

    def _log(self, msg):
        logger = getattr(self.config, "logger", None)
        if logger is not None and self.rank0:
            logger.info(msg)

    def _save(self, name):
        if self.rank0:
            torch.save(self.model.state_dict(), f"{self.config.save_path}/models/{name}")

    def train(self):
        self._log("\n================== Start training ===================")
        self._announce()
        it = self.start_iter
        for it in range(self.start_iter, self.iterations + 1):
            if it % self.config.bg_gen_freq == 0 and self.config.gen_bg:
                self.data.gen_bg(self.config, data="train")
            # The reference draws train_k, [validation / test batches of k], [gen_bg(k+1)], train_k+1 - and its loaders may share
            # one global generator.  Batch k+1 is therefore prefetched (drawn while step k computes) only when nothing else
            # draws or regenerates between the two; otherwise it is drawn at the top of iteration k+1, in the reference's place.
            # `config.host_prefetch_depth` (default 2, host-batch route): the conversion + copy of a batch takes about as long as a
            # step, so drawn ONE ahead the iteration waited for the worker (scripts/dev/trainer_iter_probe.py lagged); the same rule,
            # applied to every iteration in between, lets batch k+2 be drawn under step k.
            self._prefetch = self._clear_ahead(it)
            self._train_iter(it)
            if getattr(self, "_loss_pending", None) is not None and (it % self.config.val_freq == 0 or it == self.iterations or it % 1000 == 0):
                pending, self._loss_pending = self._loss_pending, None
                self._flush_loss(pending)                               # lagged log: nothing stays behind a validation round, a checkpoint or the end
            if it % self.config.val_freq == 0:
                self._validate_iter(it, source="validation")
                if self.config.task != "pascal_1d":
                    self._validate_iter(it, source="test")
            if it % 1000 == 0:
                self.save_intermediate_model(it)
        self._save(f"model_end_{it}.pt")
        if getattr(self.config, "close_after_train", True):
            self.close()
        self._log(f"models have been saved to {self.config.save_path}")
        self._log("================= Training finished =================\n")

    def _clear_ahead(self, it):
        """How many of the training batches behind iteration `it`'s may be drawn now: batch it+j only if no validation round, end of
        training or background regeneration lies between iteration it and it+j (the reference's order of draws, train.py /
        model_trainer.py:59-70)."""
        depth = max(1, int(getattr(self.config, "host_prefetch_depth", 2)))
        n = 0
        for i in range(it, it + depth):
            if i < self.iterations and i % self.config.val_freq != 0 and not ((i + 1) % self.config.bg_gen_freq == 0 and self.config.gen_bg):
                n += 1
            else:
                break
        return n

    def _batch(self, source):
        """One device batch of `source`.  With the ingest path the NEXT training batch starts its host -> device copy as
        soon as the current one is handed out (when train() allows it: see `_prefetch`), so it overlaps with the step the
        caller is about to run; validation / test batches are staged and taken on the spot."""
        if self.ingest is None:
            def draw(src):
                return self.data.get_batch(source=src, tasks_per_batch=self.config.tasks_per_batch, shot=self.config.max_ctx_num)
            hp = self._host_prefetch
            if hp is None:
                dev = self.config.device
                return tuple(t.to(dev) for t in draw(source))
            if source != "train":
                return hp.take(hp.stage(draw(source)))
            ticket = self._staged_q.popleft() if self._staged_q else hp.stage(draw("train"))
            self._stage_later = int(self._prefetch)     # the next batches' (host-blocking) copies go out BEHIND this step's launch: _stage_next()
            batch = hp.take(ticket)
            self._fixed_batch = hp.last_fixed           # the byte route delivers every batch of a shape in the same device tensors
            return batch

        def stage(src):
            return self.ingest.stage(*self.data.get_batch_u8(source=src, tasks_per_batch=self.config.tasks_per_batch,
                                                             shot=self.config.max_ctx_num))
        if source != "train":
            return self.ingest.take(stage(source))
        ticket, self._staged = (self._staged or stage("train")), None
        batch = self.ingest.take(ticket)
        if self._prefetch:
            self._staged = stage("train")
        return batch

    def _stage_next(self):
        """Host-batch route: draw the next training batch and start its copy to the device - called right after the current
        step has been enqueued, so the (pageable, host-blocking) copy runs beside the step instead of in front of it."""
        if self._host_prefetch is not None and getattr(self, "_stage_later", 0):
            ahead, self._stage_later = self._stage_later, 0
            if self._eps is not None and self._eps is not False:
                ahead = 1       # Bayes-by-backprop models: batch k+2 would be drawn in front of step k+1's eps - one ahead keeps the reference's order on a shared generator
            while len(self._staged_q) < ahead:
                self._staged_q.append(self._host_prefetch.stage(self.data.get_batch(source="train", tasks_per_batch=self.config.tasks_per_batch,
                                                                                    shot=self.config.max_ctx_num)))

    def _seed(self, loss):
        """d loss / d loss = 1, allocated once: autograd's implicit seed is a fill kernel per iteration."""
        s = getattr(self, "_one", None)
        if s is None or s.device != loss.device or s.dtype != loss.dtype:
            s = self._one = torch.ones((), device=loss.device, dtype=loss.dtype)
        return s

    # ---- graph-replayed training iterations -------------------------------------------------------------------
    def _step_body(self, ctx_x, qry_x, ctx_y, qry_y, with_optimizer):
        self.optimizer.zero_grad()
        if getattr(self.config, "contrastive", False):
            pr_mu, pr_var, kl, contra_loss = self.model(ctx_x, ctx_y, qry_x, qry_y)
        else:
            pr_mu, pr_var, kl = self.model(ctx_x, ctx_y, qry_x)
            contra_loss = None
        # config.loss_aside (default on): with the bare loss as the objective (no KL / contrastive term computes with its value) the value
        # is left to the model's first backward kernel (mlhot.ops.loss_value_aside) - it is read after the backward, below
        with loss_value_aside(enabled=self._bare_loss(kl, contra_loss)):
            losses = self._objective(pr_mu, pr_var, qry_y, kl)
            if contra_loss is not None:
                losses = losses + contra_loss * self.config.contrastive_rate
            losses.backward(gradient=self._seed(losses))
        if with_optimizer:
            self.optimizer.step()
        return losses.detach()

    def _objective(self, pr_mu, pr_var, qry_y, kl):
        """loss + kl * beta: inside the loss's launches when the loss object offers that (trainer.losses.LossFunc.calc_objective), as the
        reference's two operators behind `calc_loss` for any other loss object."""
        fused = getattr(self.loss, "calc_objective", None)
        if fused is not None:
            out = fused(pr_mu, pr_var, qry_y, kl, self.config.beta)
            if out is not None:
                return out
        return add_scaled(self.loss.calc_loss(pr_mu, pr_var, qry_y), kl, self.config.beta)

    def _bare_loss(self, kl, contra_loss):
        return (bool(getattr(self.config, "loss_aside", True)) and contra_loss is None and not isinstance(kl, torch.Tensor)
                and (not kl or not self.config.beta))

    def _graph_train_iter(self, it):
        """One training iteration replayed from a hipGraph (see the module docstring).  Returns the device loss tensor."""
        if not getattr(self.optimizer, "capturable", False):
            raise ValueError("config.graph_steps needs a capture-safe optimizer (e.g. mlhot.optim.FlatAdam(capturable=True))")
        self.model.train()
        batch = self._batch("train")
        key = tuple(tuple(t.shape) for t in batch)
        single = self.bucket.world_size() == 1                       # the all-reduce (and the step behind it) stays outside the graph
        if self._side is None:
            self._side = torch.cuda.Stream(self.config.device)
        static = self._static_in.get(key)
        if static is None:                                           # fixed input addresses for this shape
            static = self._static_in[key] = batch if (self.ingest is not None or self._fixed_batch) else tuple(t.clone() for t in batch)
        if static[0].data_ptr() != batch[0].data_ptr():
            for d, t in zip(static, batch):
                d.copy_(t)
        # lr / betas / eps / weight decay are scalar kernel arguments, frozen inside a captured graph: when a scheduler (or the caller)
        # changed them since the capture, every shape's graph is dropped - this iteration runs eagerly with the new values, the next
        # one captures again
        hyper = self._hyper()
        if getattr(self, "_captured_hyper", hyper) != hyper:
            self._graphs.clear()
            self.recaptures = getattr(self, "recaptures", 0) + 1
        self._captured_hyper = hyper
        entry = self._graphs.get(key)
        cur = torch.cuda.current_stream(self.config.device)
        eps = self._eps_stager()
        if eps is not None and eps.shapes:
            eps.stage()                                              # this iteration's draws (collected from the prefetch, or drawn now): the
        staged = eps.active() if eps is not None and eps.shapes else contextlib.nullcontext()     # reference's order - batch, then eps
        if entry is None:                                            # first time: a real, eager iteration on the capture stream
            self._side.wait_stream(cur)
            with torch.cuda.stream(self._side), (eps.recording() if eps is not None and not eps.shapes else staged):
                loss = self._step_body(*static, with_optimizer=single)
            cur.wait_stream(self._side)
            self._graphs[key] = "warm"
        else:
            from mlhot import ops
            if entry == "warm":                                      # second time: capture (nothing executes), then replay below
                graph = torch.cuda.CUDAGraph()
                self._side.wait_stream(cur)
                taps, ops.saved_taps = ops.saved_taps, []            # the captured forward's saved buffers (test / diagnostic hook, see below)
                loggers = [m for m in self.model.modules() if hasattr(m, "tap_log")]      # the ResNet family's form of the same hook
                listening = [m.tap_log for m in loggers]
                for m in loggers:
                    m.tap_log = []
                try:
                    with staged, capture_graph(graph, self._side):
                        static_loss = self._step_body(*static, with_optimizer=single)
                finally:
                    captured_taps, ops.saved_taps = ops.saved_taps, taps
                    captured_logs = [m.tap_log for m in loggers]
                    for m, log in zip(loggers, listening):
                        m.tap_log = log
                # the gradient tensors THIS graph writes (its private pool): a replay does not rebind p.grad, and another
                # shape's graph or eager warm-up may have re-pointed it since
                entry = self._graphs[key] = (graph, static_loss, [p.grad for p in self.bucket.params], captured_taps, (loggers, captured_logs))
            entry[0].replay()
            loss = entry[1]
            for p, g in zip(self.bucket.params, entry[2]):
                p.grad = g
            if ops.saved_taps is not None:                          # a replay runs no Python forward: hand a listener the graph's own saved
                ops.saved_taps.extend(entry[3])                      # buffers, which now hold THIS iteration's routing
            for m, log in zip(*entry[4]):
                if m.tap_log is not None:
                    m.tap_log.extend(log)
        if not single:
            self._sync_and_step()
        self._stage_next()
        if eps is not None and eps.shapes and self._prefetch:
            eps.prefetch()      # the next iteration's draws on host threads under this step - behind the next batch's draw, and only
        return loss             # when nothing else (a validation forward) touches the CPU generator in between

    def _eps_stager(self):
        """A StagedEps for models with Bayes-by-backprop layers (their forward draws eps on the torch CPU generator, bbb/BBBConv.py:86-95:
        not capturable as it stands), None otherwise.  Always the host source: the same numbers as the lazy draws, bit for bit, and the
        validation forwards in between keep drawing from the same generator."""
        if self._eps is None:
            from networks.bbb.BBBConv import BBBConv2d
            from networks.bbb.BBBLinear import BBBLinear
            if any(isinstance(m, (BBBConv2d, BBBLinear)) for m in self.model.modules()):
                from networks.bbb.eps import StagedEps
                self._eps = StagedEps(self.config.device)
            else:
                self._eps = False
        return self._eps or None

    def _sync_and_step(self):
        """Gradient all-reduce + optimizer step of a multi-rank iteration; the 1/world average rides in the optimizer's
        gradient scale when it has one (mlhot.optim.FlatAdam), instead of a separate pass over the bucket."""
        from mlhot.optim import FlatAdam
        fused = isinstance(self.optimizer, FlatAdam)                # its step(grad_scale=...) folds the average into the update
        if fused:
            scale = self.bucket.sync(defer_scale=True, wait=False)  # issued on the communication stream ...
            self.bucket.finish()                                    # ... joined right before the first reader of the gradients
            self.optimizer.step(grad_scale=scale)
        else:
            self.bucket.sync()
            self.optimizer.step()

    def _train_iter(self, it):
        if self._graph_default:
            loss = self._graph_train_iter(it)
            every = max(1, int(getattr(self.config, "log_every", 1)))
            if self._lagged() and every == 1:
                return self._lagged_log(it, loss)
            if it % every and it != self.iterations:
                return None                                          # no host sync on this iteration
            value = loss.item()
            return self._report(it, value)
        self.model.train()
        self.optimizer.zero_grad()
        ctx_x, qry_x, ctx_y, qry_y = self._batch("train")
        contrastive = getattr(self.config, "contrastive", False)
        if contrastive:                                   # FCL* models take the target labels and return the NT-Xent term
            pr_mu, pr_var, kl, contra_loss = self.model(ctx_x, ctx_y, qry_x, qry_y)
        else:
            pr_mu, pr_var, kl = self.model(ctx_x, ctx_y, qry_x)
        with loss_value_aside(enabled=self._bare_loss(kl, contra_loss if contrastive else None)):
            losses = self._objective(pr_mu, pr_var, qry_y, kl)        # loss + kl * beta (model_trainer.py:77-78)
            if contrastive:
                losses = losses + contra_loss * self.config.contrastive_rate
            self.bucket.arm()                                         # world > 1: the early bucket's all-reduce goes out from inside backward()
            losses.backward(gradient=self._seed(losses))
        self._sync_and_step()
        self._stage_next()
        value = losses.item()                                     # the iteration's only host sync
        return self._report(it, value)

    def _report(self, it, value):
        if self.writer is not None and self.rank0:
            self.writer.add_scalar("Loss/train", value, it)
        self._log(f"Train Iteration {it} loss: {value:.4f}\n")
        if not math.isfinite(value):
            self._log(f"Loss is {value}, stopping training")
            sys.exit(1)
        return value

    def _lagged(self):
        return bool(getattr(self.config, "lagged_loss_log", True))

    def _lagged_log(self, it, loss):
        """`config.lagged_loss_log` (default on for replayed iterations; False = the read right behind the step): every iteration's loss
        is still fetched, logged and checked - one iteration LATE.  What an observer of the reference's loop sees is unchanged: the same
        log lines and TensorBoard points in the same order, the same exit code on a non-finite loss with the same files on disk (the
        pending loss is flushed before everything that writes - validation rounds, the it % 1000 checkpoints, the final save); only the
        process's in-memory weights have taken one more step when it exits.  The
        reference reads `losses.item()` right behind the step (model_trainer.py:87-91); behind a replayed step that read is a host
        sync, so iteration k + 1 cannot be launched before k has finished and the GPU idles through the host's turn-around (~0.2 ms of
        a 0.8 ms iteration at c3's shape: scripts/dev/trainer_iter_probe.py).  Here iteration k's loss leaves the device by an
        asynchronous copy into pinned memory queued behind its graph; it is read, logged and checked when iteration k + 1 has been
        launched (a non-finite loss stops training one optimizer step later than the reference would), the last one when train()
        ends.  Returns the PREVIOUS iteration's loss (None for the first)."""
        dev = torch.device(self.config.device)
        ring = getattr(self, "_loss_ring", None)
        if ring is None:
            ring = self._loss_ring = [(torch.zeros((), dtype=torch.float32).pin_memory(), torch.cuda.Event()) for _ in range(2)]
            self._loss_pending = None
        buf, ev = ring[it & 1]
        buf.copy_(loss, non_blocking=True)
        ev.record(torch.cuda.current_stream(dev))
        prev, self._loss_pending = self._loss_pending, (it, buf, ev)
        if prev is None:
            return None
        return self._flush_loss(prev)

    def _flush_loss(self, pending):
        it, buf, ev = pending
        ev.synchronize()
        return self._report(it, float(buf))

    def _validate_iter(self, it, source):
        self.model.eval()
        with torch.no_grad():
            self.data.test_counter = 0
            rng = getattr(self.data, "test_rng" if source == "test" else "val_rng", None)
            if rng is not None:
                rng.seed(42)
            vals = []
            for _ in range(self.config.val_iters):
                ctx_x, qry_x, ctx_y, qry_y = self._batch(source)
                if getattr(self.config, "contrastive", False):
                    pr_mu, pr_var, _, _ = self.model(ctx_x, ctx_y, qry_x, qry_y, test=True)
                else:
                    pr_mu, pr_var, _ = self.model(ctx_x, ctx_y, qry_x, test=True)
                vals.append(self.loss.calc_loss(pr_mu, pr_var, qry_y, test=True).view(1))
            vals = torch.cat(vals)
            # torch.std of a single value is nan (the reference writes that nan when val_iters == 1)
            loss, std = vals.mean(), (vals.std() if vals.numel() > 1 else vals.new_full((), float("nan")))
            if self.writer is not None and self.rank0:
                self.writer.add_scalar(f"Loss/{source}", loss, it)
            self._log(f"{source} {it} loss: {loss.item():.4f}")
            if loss < self.best_loss[source]:
                self.best_loss[source] = loss
                self._log(f"save best {source} model epoch : {it}\n")
                self._save(f"best_{source}_model.pt")
                if self.rank0:       # the reference's three lines (trainer/model_trainer.py:135-138)
                    with open(os.path.join(self.config.save_path, f"best_{source}_error.txt"), "a") as f:
                        f.write(f"Best Step: {it} \n")
                        f.write(f"Best {source} Loss: \n{str(loss)}\n")
                        f.write(f"Best {source} Loss std: \n{str(std)}\n")
        return loss.item()

    def save_intermediate_model(self, it):
        self._save("model_intermediate.pt")
        self._log(f"save intermediate model iter: {it}")
