"""Common implementation of the ResNet-encoder neural processes for ShapeNet3D / Distractor
(reference: networks/CondNeuralProcess.py, networks/ANP.py).

Same construction order / state_dict keys as the reference; the forward composes the mlhot HIP
operators (run-time-shaped convolutions, linears, shot-axis aggregators, FAVOR+ attention).
"""
import torch
from torch import nn

from mlhot.ops import AggFunction, FavorFunction, HeadStacksFunction, LinearFunction, StackedLinearFunction
from networks.fast_attention import FastAttention
from networks.models import AttnLinear, ImageEncoder, NPDecoder, _mlp3, run_trunks


class HeadStack:
    """The weights / biases of the N per-head AttnLinear layers as views of ONE [N*h, h] / [N*h] buffer, so that the heads run as
    one linear without a torch.cat per forward (3 stacks x (weight + bias) = 6 concatenations of 6 MB per step, and as many
    gradient splits in the backward).  The modules keep their own Parameters and state_dict keys (the reference's layout:
    `_W_k.3.linear.weight`); only their storage is shared.  `.to(device)` / `load_state_dict(assign=True)` give every
    parameter a storage of its own again - tensors() notices (data pointers) and re-stacks."""

    def __init__(self, mods):
        self.mods, self.w, self.b = list(mods), None, None

    def _aliased(self):
        if self.w is None:
            return False
        h = self.mods[0].linear.weight.shape[0]
        for i, m in enumerate(self.mods):
            wt, bs = m.linear.weight, m.linear.bias
            if (wt.device != self.w.device or wt.data_ptr() != self.w.data_ptr() + 4 * i * h * wt.shape[1] or not wt.is_contiguous()
                    or bs.data_ptr() != self.b.data_ptr() + 4 * i * h):
                return False
        return True

    def _adopt(self):
        """The heads' Parameters already lie behind one another in ONE storage (mlhot.optim.FlatAdam re-points every parameter at
        a view of its flat buffer, laid out by ResNetNP.flat_layout with the stacks contiguous): the stack IS that range - take
        views of it instead of concatenating (a cat would pull the parameters out of the optimizer's buffer again)."""
        ws, bs = [m.linear.weight for m in self.mods], [m.linear.bias for m in self.mods]
        for ts in (ws, bs):
            t0, n = ts[0], ts[0].numel()
            for i, t in enumerate(ts):
                if (not t.is_contiguous() or t.device != t0.device or t.dtype != t0.dtype or t.shape != t0.shape
                        or t.untyped_storage().data_ptr() != t0.untyped_storage().data_ptr() or t.storage_offset() != t0.storage_offset() + i * n):
                    return False
        N, (h, k) = len(ws), ws[0].shape
        with torch.no_grad():
            self.w = torch.empty(0, dtype=ws[0].dtype, device=ws[0].device).set_(ws[0].untyped_storage(), ws[0].storage_offset(), (N * h, k))
            self.b = torch.empty(0, dtype=bs[0].dtype, device=bs[0].device).set_(bs[0].untyped_storage(), bs[0].storage_offset(), (N * h,))
        return True

    def tensors(self):
        if not self._aliased() and not self._adopt():
            ws, bs = [m.linear.weight for m in self.mods], [m.linear.bias for m in self.mods]
            h = ws[0].shape[0]
            if any(t.untyped_storage().nbytes() > 4 * t.numel() + 64 for t in ws + bs):
                # views of a larger buffer (mlhot.optim.FlatAdam's) that do not lie behind one another: concatenating would pull them out
                # of the optimizer's buffer and training of these heads would stop without a word
                raise RuntimeError("HeadStack: the heads' parameters are views of a flat buffer but not one contiguous block "
                                   "(ResNetNP.flat_layout lays a stack's weights, then its biases, back to back)")
            with torch.no_grad():
                self.w, self.b = torch.cat([t.detach() for t in ws], dim=0), torch.cat([t.detach() for t in bs], dim=0)
                for i, (wt, bt) in enumerate(zip(ws, bs)):
                    wt.data = self.w[i * h:(i + 1) * h]
                    bt.data = self.b[i * h:(i + 1) * h]
        return self.w, self.b

    def params(self):
        return [m.linear.weight for m in self.mods] + [m.linear.bias for m in self.mods]


class ResNetNP(nn.Module):
    ATTENTION = False
    TRANSFORM_Y = False   # *Distractor classes: labels go through Linear(label_dim -> dim_w) first (CNPDistractor.py:43,89)
    CONTRASTIVE = False   # FCL* classes: forward takes the target labels too and returns a 4th value, the NT-Xent term
    N_HEADS = 8

    def __init__(self, config):
        super().__init__()
        self.device = config.device
        self.img_size = config.img_size
        self.img_channels = self.img_size[2] - 1 if config.task == "shapenet_3d" else self.img_size[2]
        self.task_num = config.tasks_per_batch
        self.label_dim = config.input_dim
        self.agg_mode = config.agg_mode
        self.img_agg = config.img_agg
        self.y_dim = config.output_dim
        if self.ATTENTION:
            self.temperature = getattr(config, "temperature", 0.07)
        if self.TRANSFORM_Y:
            self.dim_w = config.dim_w
        torch.manual_seed(config.seed)

        self.img_encoder = ImageEncoder(aggregate=self.img_agg, task_num=self.task_num, img_channels=self.img_channels)
        label_width = self.label_dim
        if self.TRANSFORM_Y:
            self.transform_y = nn.Linear(self.label_dim, self.dim_w)
            label_width = self.dim_w
        self.task_encoder = nn.Sequential(nn.Linear(256 + label_width, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(),
                                          nn.Linear(256, 256), nn.ReLU())
        if not self.ATTENTION and self.agg_mode == "baco":
            self.latent_mu = nn.Linear(256, 256)
            self.latent_var = nn.Linear(256, 256)
        self.mu = nn.Linear(256, 256)
        self.decoder = NPDecoder(aggregate=self.img_agg, output_dim=self.y_dim, task_num=self.task_num,
                                 img_channels=self.img_channels, img_size=self.img_size)
        if self.ATTENTION:
            h = 256
            self._W_k = nn.ModuleList([AttnLinear(h, h) for _ in range(self.N_HEADS)])
            self._W_v = nn.ModuleList([AttnLinear(h, h) for _ in range(self.N_HEADS)])
            self._W_q = nn.ModuleList([AttnLinear(h, h) for _ in range(self.N_HEADS)])
            self._W = AttnLinear(self.N_HEADS * h, h)
            self.attn = FastAttention(dim_heads=256, causal=False)
            self.n_heads = self.N_HEADS

    def early_grad_parameters(self):
        """Parameters whose gradients are complete before the image trunks' backward starts (that backward - ONE C call over
        the context / target / decoder passes - is the last node of the autograd graph): the MLPs, the attention and the decoder
        head, i.e. everything outside `img_encoder`, `decoder.resnet` and `decoder.conv1`.  mlhot.dist.GradBucket(early=...) all-reduces them
        from inside the backward, under the trunks' ~1 ms."""
        return [p for k, p in self.named_parameters() if not self._trunk_parameter(k)]

    @staticmethod
    def _trunk_parameter(name):
        """Gradient produced by the image trunks' backward: the encoder, the decoder's ResNet and the decoder's stem convolution
        (`decoder.conv1` is the first layer of the decoder's trunk pass, models.py:120-192 - until round 5 it was counted among the
        early parameters, so an armed eager backward issued the early bucket only after the trunks: correct, but nothing overlapped)."""
        return name.startswith("img_encoder.") or name.startswith("decoder.resnet.") or name.startswith("decoder.conv1.")

    def flat_layout(self, ctx_num=None, test_num=None):
        """(total floats, {parameter name: offset}, floats that are ever stepped) for mlhot.optim.FlatAdam: ONE flat parameter buffer
        (and, through enable_flat_grads(), its mirror for the gradients) - the early-bucket parameters first (the range
        mlhot.dist.GradBucket all-reduces from inside the backward stays contiguous), each per-head AttnLinear stack as one block
        (its 8 weights, then its 8 biases: HeadStack adopts the range), the image trunks, and at the very end the parameters no
        forward ever uses (`resnet.fc.*`: part of the reference's state_dict and of its optimizer, never given a gradient - torch's
        Adam skips them, the flat update stops in front of them).  Every slice 16-byte aligned.  The batch shape does not matter."""
        named = dict(self.named_parameters())
        order, seen = [], set()

        def take(names):
            for n in names:
                if n in named and n not in seen:
                    seen.add(n)
                    order.append(n)
        if self.ATTENTION:
            for stack in ("_W_q", "_W_k", "_W_v"):
                take([f"{stack}.{i}.linear.weight" for i in range(self.N_HEADS)])
                take([f"{stack}.{i}.linear.bias" for i in range(self.N_HEADS)])
        dead = [n for n in named if ".resnet.fc." in n]
        late = [n for n in named if self._trunk_parameter(n) and n not in dead]
        take([n for n in named if n not in late and n not in dead])
        take(late)
        active_names = len(order)
        take(dead)
        offs, total, active = {}, 0, 0
        stacked = set()        # members of a head stack's block except the last: laid back to back, the 16-byte padding behind the block
        if self.ATTENTION:
            for stack in ("_W_q", "_W_k", "_W_v"):
                for kind in ("weight", "bias"):
                    stacked.update(f"{stack}.{i}.linear.{kind}" for i in range(self.N_HEADS - 1))
        for i, n in enumerate(order):
            if i == active_names:
                active = total
            offs[n] = total
            total += named[n].numel() if n in stacked else (named[n].numel() + 3) // 4 * 4
        if active_names == len(order):
            active = total
        return total, offs, active

    def enable_flat_grads(self, on=True):
        """Every gradient of this model in ONE flat buffer (mlhot/arena.py): the kernels write weight / bias gradients straight
        into their slots, mlhot.dist.GradBucket all-reduces the buffer in place (the early bucket = its first range) instead of
        packing / unpacking <= 15 MB per step.  Process-global while on (one model trains at a time)."""
        from mlhot import binding
        from mlhot.arena import GradArena
        arena = GradArena(self.parameters(), first=self.early_grad_parameters()) if on else None
        self.__dict__["_arena"] = arena
        binding.set_grad_arena(arena)
        return arena

    def enable_split_backward(self, on=True):
        """Cut the autograd graph in front of the image trunks: the forward hands everything downstream DETACHED copies of the
        trunks' output maps (and of the Bayes-by-backprop KL) and remembers the (output, leaf) pairs, so that the backward can run
        in two parts - mlhot.dist.backward_in_two: all early-bucket gradients, then the trunks - with the first bucket's
        all-reduce in between.  Values and gradients are unchanged; a plain loss.backward() on a cut forward would stop at the
        cut, so only callers that run the second part switch this on."""
        self.__dict__["_split_backward"] = bool(on)
        self.__dict__["_cut_pairs"] = []

    def _cut(self, tensors):
        """Identity unless enable_split_backward(True): then each tensor that carries a graph is replaced by a detached leaf."""
        if not self.__dict__.get("_split_backward") or not torch.is_grad_enabled():
            return tensors
        out = []
        for t in tensors:
            if torch.is_tensor(t) and t.requires_grad:
                leaf = t.detach().requires_grad_(True)
                self.__dict__["_cut_pairs"].append((t, leaf))
                out.append(leaf)
            else:
                out.append(t)
        return out

    def _refresh_arena(self):
        arena = self.__dict__.get("_arena")
        if arena is not None:
            arena.refresh()          # a no-op unless a parameter's storage changed (first step: the head stacks were just built)

    # the 8 per-head AttnLinear layers run as ONE linear over the stacked weights; rows come out
    # token-major / head-minor, which is the layout the FAVOR+ kernels take
    def _heads(self, x, mods):
        stacks = self.__dict__.setdefault("_head_stacks", {})
        st = stacks.get(id(mods))
        if st is None:
            st = stacks[id(mods)] = HeadStack(mods)
        w, b = st.tensors()
        T, N, _ = x.shape
        return StackedLinearFunction.apply(x, w, b, self.N_HEADS, *st.params()).view(T, N, self.N_HEADS, -1)

    def _stack(self, mods):
        stacks = self.__dict__.setdefault("_head_stacks", {})
        st = stacks.get(id(mods))
        if st is None:
            st = stacks[id(mods)] = HeadStack(mods)
        return st

    def _multihead_attention(self, k, v, q):
        if q.is_cuda and max(q.shape[0] * q.shape[1], k.shape[0] * k.shape[1]) <= 512 and q.shape[-1] % 4 == 0:
            # the three head stacks in ONE launch per direction (few rows: mlhot_linear_multi_*)
            sts = [self._stack(m) for m in (self._W_q, self._W_k, self._W_v)]
            args, params = [], []
            for x, st in zip((q, k, v), sts):
                w, b = st.tensors()
                args += [x, w, b]
                params += st.params()
            qh, kh, vh = HeadStacksFunction.apply(self.N_HEADS, 3, *args, *params)
        else:
            qh, kh, vh = self._heads(q, self._W_q), self._heads(k, self._W_k), self._heads(v, self._W_v)
        merged = FavorFunction.apply(qh, kh, vh, self.attn.projection_matrix)
        return self._W(merged)

    def _aggregate(self, feats, quirk=False):
        """Shot-axis aggregation of the task-encoder features + `mu` -> task embedding [T, 256].  `quirk`: the target-set path
        of FCLCNPDistractor feeds latent_var with latent_mu's OUTPUT (FCLCNPDistractor.py:133-134); reproduced as is."""
        if self.agg_mode in ("mean", "max"):
            r, _ = AggFunction.apply(self.agg_mode, feats, None)
        elif self.agg_mode == "baco":
            mu_l = LinearFunction.apply(feats, self.latent_mu.weight, self.latent_mu.bias, "none")
            lv = LinearFunction.apply(mu_l if quirk else feats, self.latent_var.weight, self.latent_var.bias, "none")
            r, _ = AggFunction.apply("baco", mu_l, lv)
        else:
            raise TypeError("agg_mode is not applicable for CNP, choose from ['mean', 'max', 'baco']")
        return LinearFunction.apply(r, self.mu.weight, self.mu.bias, "none")

    def forward(self, batch_train_images, label_train, batch_test_images, *rest, test=False):
        """(ctx images, ctx labels, target images[, test]) -> (mu, var, 0); the FCL classes take the target labels as 4th
        positional argument and return (mu, var, 0, contrastive term) like the reference (FCLANP.py:108, FCLCNPDistractor.py:82)."""
        self._refresh_arena()
        label_test = None
        if self.CONTRASTIVE:
            if not rest:
                raise TypeError("forward() missing the target labels (label_test)")
            label_test, rest = rest[0], rest[1:]
        if rest:
            test = rest[0]
        from trainer.losses import LossFunc
        self.test_num = batch_test_images.shape[1]
        self.ctx_num = batch_train_images.shape[1]
        C, H, W = self.img_channels, self.img_size[0], self.img_size[1]
        tgt_imgs = batch_test_images.reshape(-1, C, H, W)
        ctx_imgs = batch_train_images.reshape(-1, C, H, W)
        contra_cnp = self.CONTRASTIVE and not test and not self.ATTENTION
        # Every ResNet pass of the step goes out together (one launch sequence instead of one per pass): context images and -
        # for the attention / contrastive models - target images through the encoder, target images through the decoder.  Call
        # order of the reference (and of the tap logs): context, [target,] decoder.
        jobs, roles = [], []
        if self.ctx_num:
            jobs.append(self.img_encoder.trunk_job(ctx_imgs)); roles.append("ctx")
            if self.ATTENTION or contra_cnp:
                jobs.append(self.img_encoder.trunk_job(tgt_imgs)); roles.append("tgt")
        jobs.append(self.decoder.trunk_job(tgt_imgs)); roles.append("dec")
        if self.__dict__.get("_split_backward"):
            self.__dict__["_cut_pairs"] = []
        maps = run_trunks(jobs)
        if maps is not None:
            maps = self._cut(maps)
        fm = dict(zip(roles, maps)) if maps is not None else {}

        def encode(imgs, role):
            return self.img_encoder.features(fm[role]) if role in fm else self.img_encoder(imgs)

        z_0 = None
        if self.ctx_num:
            if self.TRANSFORM_Y:
                label_train = LinearFunction.apply(label_train, self.transform_y.weight, self.transform_y.bias, "none")
            x_ctx = encode(ctx_imgs, "ctx")
            feats = _mlp3(x_ctx, self.task_encoder, last_relu=True, side=label_train)      # cat([x_ctx, labels]), ANP.py:113
            pre = None
            if self.ATTENTION:
                x_tgt = encode(tgt_imgs, "tgt")
                sample = self._multihead_attention(x_ctx, feats, x_tgt)
                if self.CONTRASTIVE and not test:                 # the contrastive term reads mu's output itself
                    sample = LinearFunction.apply(sample, self.mu.weight, self.mu.bias, "none")
                else:
                    pre = self.mu                                 # mu runs inside the decoder head's launch (models._mlp3)
            else:
                z_0 = self._aggregate(feats)
                sample = z_0[:, None, :].expand(-1, self.test_num, -1)
        else:
            sample, pre = torch.zeros(self.task_num, self.test_num, 256, device=batch_test_images.device), None
        contra = 0
        if self.CONTRASTIVE and not test:
            if self.ATTENTION:
                contra = LossFunc.contrastive_loss_ANP(sample, t=self.temperature)
            else:
                if z_0 is None:
                    raise ValueError("the contrastive term needs a non-empty context set (the reference fails here as well: z_0 is unbound)")
                x_qry = encode(tgt_imgs, "tgt")
                if self.TRANSFORM_Y:
                    label_test = LinearFunction.apply(label_test, self.transform_y.weight, self.transform_y.bias, "none")
                z_q = self._aggregate(_mlp3(torch.cat([x_qry, label_test], dim=2), self.task_encoder, last_relu=True), quirk=True)
                contra = LossFunc.contrastive_loss(z_0, z_q)
        out, var = self.decoder(batch_test_images, sample, fmap=fm.get("dec"), pre=pre)
        if self.CONTRASTIVE:
            return out, var, 0, contra
        return out, var, 0
