"""torch.autograd bridges onto the C ABI (one Function per hot-path row of SURVEY.md §8a).

The Functions are glue only: they hand raw device pointers to libmlhot.so and keep the
opaque `saved` workspace alive between forward and backward.  There is NO eager/CPU
fallback - a tensor that is not on a HIP device raises.
"""
import torch

from . import lib
from .binding import MlhotError


# Test / diagnostic hook: when set to a list, the whole-model and encoder bridges append (kind, dims | n_images, saved) for every
# forward, so that a parity test can read the kernels' own ReLU / pool routing decisions (binding.enc_routes, np_saved_views).
saved_taps = None


# Strict sharded parity of the FAVOR+ key stabiliser (mlhot.dist.StabiliserExchange): when set, the attention passes run as two
# staged calls around exchange.forward(x) / exchange.backward(x) on a 4-float device block (include/mlhot.h, "strict sharded parity").
_stab_exchange = None


def set_stabiliser_exchange(exchange):
    """exchange: an object with forward(x) / backward(x) (x: 4 floats on the device), or None for the rank-local stabiliser."""
    global _stab_exchange
    _stab_exchange = exchange


# The loss's gradient inside the model's backward.  LossFunc.calc_loss(mu, ., y).backward() is two dependent launches in front of
# the model's first backward kernel (the loss reduction, then d loss / d mu); the gradient needs no reduction, so when mu comes
# straight out of VanillaNPFunction the loss hands the model's backward a DESCRIPTOR (kind, labels, upstream scalar) instead of a
# gradient tensor, and the first backward kernel derives d loss / d mu itself (mlhot_np_vanilla_bwd_loss).  What autograd carries
# between the two nodes is a cached all-zero tensor of mu's shape: whatever else may flow into mu's gradient is ADDED to it by
# autograd as usual and the kernel adds the loss's share on top, so the result equals the unfused graph's for any consumer set.
#
# OPT-IN (round 6; it was a global default): the hand-over assumes that the backward pass which runs LossFunction.backward also runs
# the producer's node.  `loss.backward()` of a training step does; `torch.autograd.grad(loss, mu)` or `loss.backward(inputs=[mu])`
# stop at mu and would be handed the placeholder.  So only a caller that owns the whole step switches it on - `with
# loss_grad_in_backward():` around calc_loss + backward (trainer.ModelTrainer, bench.py's step) - and even there every hand-over is
# checked: LossFunction.backward queues an end-of-pass callback on the autograd engine, and a descriptor that is still parked on
# the producer when the pass ends (the pass stopped at mu, or raised in between) is removed and reported as an error instead of
# leaving zeros in somebody's gradient and a stale descriptor for the next pass.
defer_loss_grad = False
_zero_grads = {}


class loss_grad_in_backward:
    """Scope in which calc_loss(...).backward() leaves d loss / d mu to the model's first backward kernel (see above)."""

    def __init__(self, enabled=True):
        self.enabled = enabled

    def __enter__(self):
        global defer_loss_grad
        self.prev, defer_loss_grad = defer_loss_grad, bool(self.enabled)
        return self

    def __exit__(self, *exc):
        global defer_loss_grad
        defer_loss_grad = self.prev
        return False


def _zero_like(mu):
    key = (tuple(mu.shape), mu.device)
    z = _zero_grads.get(key)
    if z is None:
        z = _zero_grads[key] = torch.zeros_like(mu)
    return z


def _check_handed_over(node, desc):
    """End of the backward pass that parked `desc` on `node`: the producer's backward must have taken it."""
    def check():
        if node.loss is desc:
            node.loss = None
            raise RuntimeError("mlhot: the loss's gradient was left to the model's backward (loss_grad_in_backward), but this "
                               "backward pass ended without running it (autograd.grad(loss, mu) / backward(inputs=[mu])?): the "
                               "gradient handed out for mu is a placeholder of zeros.  Run such passes outside the scope.")
    return check


# The loss VALUE off the critical path.  Nothing in the backward reads the value, but as a launch of its own the reduction sits
# between the forward's last kernel and the backward's first.  Inside `with loss_value_aside():` a loss whose gradient is deferred to
# the model's backward (above) defers its value too: LossFunction.forward returns an UNWRITTEN scalar and the model's first backward
# kernel fills it from one extra workgroup (mlhot_loss_desc.value; the bits of mlhot_loss_fwd's result).  The caller promises to run
# the backward before it reads the loss and not to compute with the value in between (loss + kl * beta does: trainer.ModelTrainer
# switches this on only for a bare loss) - bench.py's step and the trainer's step are such callers.  Round 5: the first form of this
# switch ran the reduction on a forked stream (a parallel branch of the captured graph): 0.585 -> 0.608 ms per c3 step - a two-branch
# graph costs the step's other kernels more than the 4.7 us it takes out of the chain; this form has no branch.
_loss_aside = False


class loss_value_aside:
    """`enabled`: leave the loss VALUE to the backward as well; the scope always implies loss_grad_in_backward (the value rides on
    the gradient's descriptor)."""

    def __init__(self, enabled=True):
        self.enabled = enabled

    def __enter__(self):
        global _loss_aside, defer_loss_grad
        self.prev, _loss_aside = (_loss_aside, defer_loss_grad), bool(self.enabled)
        defer_loss_grad = True
        return self

    def __exit__(self, *exc):
        global _loss_aside, defer_loss_grad
        _loss_aside, defer_loss_grad = self.prev
        return False


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise MlhotError("mlhot: the hand-written HIP path needs tensors on a ROCm device "
                             "(got a CPU tensor); there is no CPU fallback")


def _c(t):
    return t if t is None or t.is_contiguous() else t.contiguous()


class VanillaNPFunction(torch.autograd.Function):
    """Whole CNP/ANP vanilla forward+backward: mlhot_np_vanilla_fwd / _bwd."""

    @staticmethod
    def forward(ctx, dims, keys, proj, ctx_x, ctx_y, qry_x, *params):
        _need_gpu(ctx_x, ctx_y, qry_x, *params)
        L = lib()
        ctx_x, ctx_y, qry_x = _c(ctx_x.float()), _c(ctx_y.float()), _c(qry_x.float())
        pd = {k: _c(p.detach()) for k, p in zip(keys, params)}
        # the exchange only concerns the attention models with a context (the empty-context branch has no keys)
        ctx.xchg = (_stab_exchange, torch.zeros(4, device=qry_x.device)) if _stab_exchange is not None and proj is not None and dims.Nc > 0 else None
        mu, saved, scratch = L.np_vanilla_fwd(dims, pd, ctx_x, ctx_y, qry_x, proj, exchange=ctx.xchg)
        if saved_taps is not None:
            saved_taps.append(("np", dims, saved))
        ctx.dims, ctx.keys, ctx.proj = dims, keys, proj
        ctx.takes_loss = ctx.xchg is None       # LossFunction may leave its gradient to this node's backward (a descriptor in ctx.loss)
        ctx.loss = None
        ctx.scratch = scratch
        ctx.save_for_backward(ctx_x, ctx_y, qry_x, mu, saved, *[pd[k] for k in keys])
        return mu

    @staticmethod
    def backward(ctx, dmu):
        ctx_x, ctx_y, qry_x, mu, saved, *params = ctx.saved_tensors
        pd = dict(zip(ctx.keys, params))
        loss, ctx.loss = ctx.loss, None
        if loss is not None and dmu.data_ptr() == _zero_like(mu).data_ptr():
            dmu = None                          # nothing but the loss flowed into mu: the kernel does not even read an addend
        grads = lib().np_vanilla_bwd(ctx.dims, pd, ctx_x, ctx_y, qry_x, mu, _c(dmu) if dmu is not None else None, saved, ctx.scratch, ctx.proj,
                                     exchange=ctx.xchg, loss=loss)
        ctx.scratch = None
        used = used_param_keys(ctx.keys, ctx.dims.Nc)
        return (None, None, None, None, None, None) + tuple(grads[k] if k in used else None for k in ctx.keys)


def used_param_keys(keys, Nc):
    """Parameters the forward touches: with an empty context only the image encoder and the
    decoder run (the zero-latent branch, ANPShapeNet1D.py:148-149), the rest get grad=None."""
    if Nc > 0:
        return set(keys)
    return {k for k in keys if k.startswith("encoder_w0.") or k.startswith("decoder0.")}


class EncVanillaFunction(torch.autograd.Function):
    """E1 alone: mlhot_enc_vanilla_fwd / _bwd on one image batch."""

    @staticmethod
    def forward(ctx, img, *params):
        _need_gpu(img, *params)
        img = _c(img.float())
        ps = [_c(p.detach()) for p in params]
        dim_w = ps[6].shape[0]
        feat, _, saved = lib().enc_vanilla_fwd(img, None, ps, dim_w)
        if saved_taps is not None:
            saved_taps.append(("enc", img.shape[0], saved))
        ctx.dim_w = dim_w
        ctx.save_for_backward(img, saved, *ps)
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        img, saved, *ps = ctx.saved_tensors
        empty = torch.empty(0, ctx.dim_w, device=img.device)
        grads = lib().enc_vanilla_bwd(img, None, ps, ctx.dim_w, _c(dfeat), empty, saved)
        return (None,) + tuple(grads)


class LinearFunction(torch.autograd.Function):
    """y = act(x W^T + b): mlhot_linear_fwd / _bwd (rows = all leading dims)."""

    @staticmethod
    def forward(ctx, x, w, b, act):
        _need_gpu(x, w, b)
        shp = x.shape
        x2 = _c(x.reshape(-1, shp[-1]).float())
        y = lib().linear_fwd(x2, _c(w.detach()), _c(b.detach()) if b is not None else None, act)
        ctx.act, ctx.shp, ctx.has_b = act, shp, b is not None
        ctx.save_for_backward(x2, _c(w.detach()), y, *([b.detach()] if b is not None else []))
        return y.view(*shp[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w, y, *bias = ctx.saved_tensors
        dx, dw, db = lib().linear_bwd(x2, w, y, _c(dy.reshape(-1, w.shape[0])), ctx.act, need_dx=ctx.needs_input_grad[0],
                                      b=bias[0] if bias else None)
        return (dx.view(ctx.shp) if dx is not None else None), dw, (db if ctx.has_b else None), None


class StackedLinearFunction(torch.autograd.Function):
    """y = x [W_0; W_1; ...]^T + [b_0; b_1; ...] for the per-head Linear layers of the attention (networks/ANP.py:75-93 runs
    them one by one and stacks the outputs): ONE linear over the stacked weight.  `stack_w` / `stack_b` are the buffers the
    heads' parameters are views of (networks/_resnet_np.py::HeadStack), so nothing is concatenated per step; the backward's
    dW / db are handed to the heads as views of one gradient tensor."""

    @staticmethod
    def forward(ctx, x, stack_w, stack_b, n_heads, *head_params):
        _need_gpu(x, stack_w, stack_b)
        shp = x.shape
        x2 = _c(x.reshape(-1, shp[-1]).float())
        y = lib().linear_fwd(x2, stack_w, stack_b, "none")
        ctx.shp, ctx.n_heads = shp, n_heads
        ctx.save_for_backward(x2, stack_w, y, stack_b)
        return y.view(*shp[:-1], stack_w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w, y, b = ctx.saved_tensors
        dx, dw, db = lib().linear_bwd(x2, w, y, _c(dy.reshape(-1, w.shape[0])), "none", need_dx=ctx.needs_input_grad[0], b=b)
        h = ctx.n_heads
        return ((dx.view(ctx.shp) if dx is not None else None), None, None, None,
                *dw.view(h, w.shape[0] // h, w.shape[1]).unbind(0), *db.view(h, -1).unbind(0))


class Linear2Function(torch.autograd.Function):
    """y = act(cat([xa, xb], -1) W^T + b) on few rows without the concatenation: the two inputs are the two k ranges of ONE launch
    (mlhot_linear_multi_* with a two-source job; the reference's torch.cat([x_ctx, labels]) -> Linear, ANP.py:113, and
    torch.cat([x, sample_features]) -> fc_mu, models.py:182-184).  Backward: both input gradients, dW and db in one launch."""

    @staticmethod
    def forward(ctx, xa, xb, w, b, act):
        _need_gpu(xa, xb, w, b)
        shp_a, shp_b = xa.shape, xb.shape
        a2, b2 = _c(xa.reshape(-1, shp_a[-1]).float()), _c(xb.reshape(-1, shp_b[-1]).float())
        wd, bd = _c(w.detach()), (_c(b.detach()) if b is not None else None)
        (y,) = lib().linear_multi_fwd([(a2, wd, bd, act, b2)])
        ctx.act, ctx.shp_a, ctx.shp_b, ctx.has_b = act, shp_a, shp_b, b is not None
        ctx.save_for_backward(a2, b2, wd, y, *([bd] if bd is not None else []))
        return y.view(*shp_a[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        a2, b2, w, y, *bias = ctx.saved_tensors
        need_a, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        ((dxa, dw, db, dxb),) = lib().linear_multi_bwd([(a2, w, y, _c(dy.reshape(-1, w.shape[0]).float()), ctx.act, bias[0] if bias else None,
                                                         b2, need_a, need_b)])
        return (dxa.view(ctx.shp_a) if dxa is not None else None, dxb.view(ctx.shp_b) if dxb is not None else None, dw,
                db if ctx.has_b else None, None)


def linear2_ok(xa, xb, w):
    """Shapes the two-source few-row Linear takes (else: torch.cat + LinearFunction)."""
    rows = xa.numel() // xa.shape[-1]
    return (xa.is_cuda and rows <= 512 and xa.shape[-1] % 4 == 0 and xb.shape[-1] % 4 == 0 and xb.shape[-1] >= 4 and w.shape[0] % 4 == 0
            and xa.shape[-1] + xb.shape[-1] == w.shape[1])


class ScaledAddFunction(torch.autograd.Function):
    """a + alpha * x for same-shaped fp32 device tensors (the trainer's `loss + kl * beta`, trainer/model_trainer.py:77-78): one launch
    per direction instead of torch's mul + add (+ mul in the backward); the same two roundings."""

    @staticmethod
    def forward(ctx, a, x, alpha):
        _need_gpu(a, x)
        ctx.alpha = float(alpha)
        return lib().axpy(_c(a.float()), _c(x.float()), ctx.alpha)

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy.float())
        return (dy if ctx.needs_input_grad[0] else None), (lib().axpy(None, dy, ctx.alpha) if ctx.needs_input_grad[1] else None), None


def add_scaled(a, x, alpha):
    """a + x * alpha the way the reference writes it; fused into one launch when both are fp32 device tensors of one shape."""
    if torch.is_tensor(a) and torch.is_tensor(x) and a.is_cuda and x.is_cuda and a.shape == x.shape and a.dtype == x.dtype == torch.float32:
        return ScaledAddFunction.apply(a, x, alpha)
    if not torch.is_tensor(x) and x * alpha == 0:
        return a                           # the models without a KL term return the int 0: loss + 0 is loss, no launch
    return a + x * alpha


class MlpChainFunction(torch.autograd.Function):
    """Up to four Linear(+ReLU / tanh) layers on few rows in ONE launch (backward: two): mlhot_mlp_chain_fwd / _bwd
    (csrc/mlp_chain.h).  Layer k's input is the previous layer's output, optionally concatenated with a second tensor
    (`side`, before or behind it) - the reference's torch.cat([x, sample_features]) / cat([x_ctx, labels]) folded into the
    kernel.  Call through mlp_chain()."""

    @staticmethod
    def forward(ctx, spec, x0, *tensors):
        # spec: per layer (act, has_bias, has_side, side_first); tensors: per layer weight[, bias][, side]
        _need_gpu(x0, *tensors)
        shp = x0.shape
        x2 = _c(x0.reshape(-1, shp[-1]).float())
        layers, it, where = [], iter(range(len(tensors))), []
        for act, has_b, has_side, side_first in spec:
            iw = next(it)
            ib = next(it) if has_b else None
            isd = next(it) if has_side else None
            side = tensors[isd] if isd is not None else None
            layers.append((_c(tensors[iw].detach()), _c(tensors[ib].detach()) if ib is not None else None, act,
                           _c(side.detach().reshape(-1, side.shape[-1]).float()) if side is not None else None, side_first))
            where.append((iw, ib, isd))
        ys = lib().mlp_chain_fwd(x2, layers)
        ctx.spec, ctx.where, ctx.shp, ctx.n_t = spec, where, shp, len(tensors)
        ctx.side_shapes = [tensors[isd].shape if isd is not None else None for _, _, isd in where]
        flat = [x2]
        for (w, b, _, side, _), y in zip(layers, ys):
            flat += [w, y] + ([b] if b is not None else []) + ([side] if side is not None else [])
        ctx.save_for_backward(*flat)
        return ys[-1].view(*shp[:-1], ys[-1].shape[-1])

    @staticmethod
    def backward(ctx, dy):
        saved = list(ctx.saved_tensors)
        x2, pos = saved[0], 1
        layers, ys = [], []
        for act, has_b, has_side, side_first in ctx.spec:
            w, y = saved[pos], saved[pos + 1]
            pos += 2
            b = side = None
            if has_b:
                b, pos = saved[pos], pos + 1
            if has_side:
                side, pos = saved[pos], pos + 1
            layers.append((w, b, act, side, side_first))
            ys.append(y)
        need_side = [isd is not None and ctx.needs_input_grad[2 + isd] for _, _, isd in ctx.where]
        dx0, gl = lib().mlp_chain_bwd(x2, layers, ys, _c(dy.reshape(-1, dy.shape[-1]).float()), need_dx0=ctx.needs_input_grad[1],
                                      need_dside=need_side)
        grads = [None] * ctx.n_t
        for (iw, ib, isd), (dw, db, ds), shp in zip(ctx.where, gl, ctx.side_shapes):
            grads[iw] = dw
            if ib is not None:
                grads[ib] = db
            if isd is not None and ds is not None:
                grads[isd] = ds.view(shp)
        return (None, dx0.view(ctx.shp) if dx0 is not None else None) + tuple(grads)


def mlp_chain(x0, layers):
    """layers: [(weight, bias | None, act, side | None, side_first)] -> the last layer's output, or None when the shapes are
    outside the chain kernels' limits (the caller then runs the layers one by one)."""
    rows = x0.reshape(-1, x0.shape[-1])
    probe = [(w, b, act, side.reshape(-1, side.shape[-1]) if side is not None else None, sf) for w, b, act, side, sf in layers]
    if not x0.is_cuda or not lib().chain_ok(rows, probe):
        return None
    spec, tensors = [], []
    for w, b, act, side, side_first in layers:
        spec.append((act, b is not None, side is not None, bool(side_first)))
        tensors += [w] + ([b] if b is not None else []) + ([side] if side is not None else [])
    return MlpChainFunction.apply(tuple(spec), x0, *tensors)


class HeadStacksFunction(torch.autograd.Function):
    """The per-head Linear stacks of SEVERAL attention inputs (query, key, value: ANP.py:80-93) in one launch per direction:
    mlhot_linear_multi_fwd / _bwd over the stacked head weights (networks/_resnet_np.py::HeadStack).  Inputs: n stacks of
    (x [T, N, h], stacked weight [H h, h], stacked bias [H h]) followed by every head's own Parameters (their gradients are views of
    the stacks' gradient tensors).  Returns n tensors [T, N, H, h]."""

    @staticmethod
    def forward(ctx, n_heads, n_stacks, *args):
        xs, ws, bs = args[0:3 * n_stacks:3], args[1:3 * n_stacks:3], args[2:3 * n_stacks:3]
        _need_gpu(*xs, *ws, *bs)
        x2 = [_c(x.reshape(-1, x.shape[-1]).float()) for x in xs]
        ys = lib().linear_multi_fwd([(x, w, b, "none") for x, w, b in zip(x2, ws, bs)])
        ctx.n_heads, ctx.n_stacks, ctx.shapes = n_heads, n_stacks, [x.shape for x in xs]
        ctx.save_for_backward(*x2, *ws, *ys, *bs)
        return tuple(y.view(*x.shape[:-1], n_heads, -1) for y, x in zip(ys, xs))

    @staticmethod
    def backward(ctx, *dys):
        n, h = ctx.n_stacks, ctx.n_heads
        sv = ctx.saved_tensors
        x2, ws, ys, bs = sv[:n], sv[n:2 * n], sv[2 * n:3 * n], sv[3 * n:]
        outs = lib().linear_multi_bwd([(x, w, y, _c(dy.reshape(-1, w.shape[0]).float()), "none", b)
                                       for x, w, y, dy, b in zip(x2, ws, ys, dys, bs)])
        lead = []
        heads_w, heads_b = [], []
        for (dx, dw, db), shp, w in zip(outs, ctx.shapes, ws):
            lead += [dx.view(shp), None, None]
            heads_w.append(dw.view(h, w.shape[0] // h, w.shape[1]).unbind(0))
            heads_b.append(db.view(h, -1).unbind(0))
        per_stack = []
        for hw, hb in zip(heads_w, heads_b):          # the order HeadStack.params() hands the Parameters over: weights, then biases
            per_stack += list(hw) + list(hb)
        return (None, None) + tuple(lead) + tuple(per_stack)


class NTXentFunction(torch.autograd.Function):
    """NT-Xent of [N, d] embeddings whose labels are arange blocks, label(i) = (i // div) % mod: mlhot_nt_xent_fwd / _bwd
    (trainer/losses.py:82-99).  No host-side index tensors, no host -> device copies, capturable."""

    @staticmethod
    def forward(ctx, z, div, mod, t):
        _need_gpu(z)
        z2 = _c(z.float())
        loss, ws = lib().nt_xent_fwd(z2, div, mod, float(t))
        ctx.meta = (div, mod, float(t))
        ctx.save_for_backward(z2, ws)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        z2, ws = ctx.saved_tensors
        div, mod, t = ctx.meta
        return lib().nt_xent_bwd(z2, div, mod, t, ws, _c(dloss.float())), None, None, None


class AggFunction(torch.autograd.Function):
    """mean / max / baco over dim 1 of rs[T,Nc,R]: mlhot_agg_fwd / _bwd.
    For baco, `rs` is mu and `lv` the pre-softplus variance logits; returns (r, sigma_z)."""

    @staticmethod
    def forward(ctx, mode, rs, lv):
        _need_gpu(rs, lv)
        rs, lv = _c(rs), _c(lv)
        r, sigma, amax = lib().agg_fwd(mode, rs, lv)
        ctx.mode = mode
        ctx.save_for_backward(rs, lv, r, sigma, amax)
        ctx.mark_non_differentiable(sigma)
        return r, sigma

    @staticmethod
    def backward(ctx, dr, _dsigma):
        rs, lv, r, sigma, amax = ctx.saved_tensors
        drs, dlv = lib().agg_bwd(ctx.mode, rs, lv, r, sigma, amax, _c(dr))
        return None, drs, dlv


class FavorFunction(torch.autograd.Function):
    """FAVOR+ attention on token-major rows q[T,Nq,H,d], k/v[T,Nc,H,d] -> merged [T,Nq,d*H]."""

    @staticmethod
    def forward(ctx, q, k, v, proj):
        _need_gpu(q, k, v, proj)
        q, k, v, proj = _c(q), _c(k), _c(v), _c(proj)
        ctx.xchg = (_stab_exchange, torch.zeros(4, device=q.device)) if _stab_exchange is not None else None
        out, ws = lib().favor_fwd(q, k, v, proj, exchange=ctx.xchg)
        ctx.save_for_backward(q, k, v, proj, out, ws)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, proj, out, ws = ctx.saved_tensors
        dq, dk, dv = lib().favor_bwd(q, k, v, proj, out, _c(dout), ws, exchange=ctx.xchg)
        return dq, dk, dv, None


class LossFunction(torch.autograd.Function):
    """LossFunc.calc_loss kinds: mlhot_loss_fwd / _bwd."""

    @staticmethod
    def forward(ctx, kind, mu, gt):
        _need_gpu(mu, gt)
        node = mu.grad_fn
        mu_c, gt = _c(mu.float()), _c(gt.float())
        ctx.kind = kind
        # the producer of mu takes the loss's gradient itself when it says so and mu reaches the loss as it left the producer
        # (not when the caller wants d loss / d mu itself - retain_grad() or a tensor hook on mu - which the placeholder would hide)
        ctx.node = node if (defer_loss_grad and kind != "degree" and mu_c is mu and getattr(node, "takes_loss", False)
                            and not mu.retains_grad and not mu._backward_hooks) else None
        ctx.save_for_backward(mu_c, gt)
        ctx.value = None
        if ctx.node is not None and _loss_aside:          # (node: mu carries a graph, i.e. a backward can follow)
            out = torch.empty((), device=mu_c.device)             # written by the model's backward (loss_value_aside above)
            ctx.value = out.detach()                               # (the same storage without the output's grad_fn: no reference cycle)
            return out
        return lib().loss_fwd(kind, mu_c, gt)

    @staticmethod
    def backward(ctx, dloss):
        mu, gt = ctx.saved_tensors
        dloss = _c(dloss.float())
        if ctx.node is not None and ctx.node.loss is None:
            # VanillaNPFunction.backward runs next (it is this gradient's only consumer node)
            desc = ctx.node.loss = (ctx.kind, gt, dloss) if ctx.value is None else (ctx.kind, gt, dloss, ctx.value)
            torch.autograd.Variable._execution_engine.queue_callback(_check_handed_over(ctx.node, desc))
            return None, _zero_like(mu), None
        if ctx.value is not None:                         # the producer's slot was taken (a second loss on the same mu): the value now
            ctx.value.copy_(lib().loss_fwd(ctx.kind, mu, gt))
        return None, lib().loss_bwd(ctx.kind, mu, gt, dloss), None


class LossPlusFunction(torch.autograd.Function):
    """loss(kind; mu, gt) + alpha * x for a device scalar x - the trainer's `losses = loss + kl * beta` (trainer/model_trainer.py:77-78)
    inside the loss's own launches: mlhot_loss_plus_fwd / _bwd (ABI 7).  Bit-identical to LossFunction followed by ScaledAddFunction
    (same reduction, product and sum rounded separately); in a replayed step it is two dependent launches instead of four, and every
    dependent launch of a graph costs ~4.7 us on this GPU whatever it computes.  Call through loss_plus()."""

    @staticmethod
    def forward(ctx, kind, mu, gt, x, alpha):
        _need_gpu(mu, gt, x)
        mu_c, gt, x = _c(mu.float()), _c(gt.float()), _c(x.float())
        ctx.kind, ctx.alpha = kind, float(alpha)
        ctx.save_for_backward(mu_c, gt)
        return lib().loss_plus_fwd(kind, mu_c, gt, x, ctx.alpha)

    @staticmethod
    def backward(ctx, dtotal):
        mu, gt = ctx.saved_tensors
        dmu, dx = lib().loss_plus_bwd(ctx.kind, mu, gt, _c(dtotal.float()), ctx.alpha, need_dx=ctx.needs_input_grad[3])
        return None, (dmu if ctx.needs_input_grad[1] else None), None, dx, None


def loss_plus(kind, mu, gt, x, alpha):
    """`LossFunction.apply(kind, mu, gt) + x * alpha` the way the reference's trainer writes its objective; one launch per direction
    when x is a one-element fp32 device tensor (a KL term), add_scaled() behind the plain loss otherwise (x = 0: the bare loss)."""
    if torch.is_tensor(x) and x.is_cuda and x.numel() == 1 and x.dtype == torch.float32 and mu.is_cuda and kind != "degree":
        return LossPlusFunction.apply(kind, mu, gt, x.reshape(()), alpha)
    return add_scaled(LossFunction.apply(kind, mu, gt), x, alpha)


class Conv2dFunction(torch.autograd.Function):
    """nn.Conv2d (+ optional fused ReLU): mlhot_conv2d_fwd / _bwd (generic run-time-shaped conv)."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, relu):
        _need_gpu(x, w, b)
        x, w = _c(x.float()), _c(w)
        bb = _c(b) if b is not None else None
        y = lib().conv2d_fwd(x, w.detach(), bb.detach() if bb is not None else None, stride, pad, relu)
        ctx.cfg = (stride, pad, relu, b is not None)
        ctx.save_for_backward(x, w.detach(), y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        stride, pad, relu, has_b = ctx.cfg
        dx, dw, db = lib().conv2d_bwd(x, w, y, _c(dy), stride, pad, relu, need_dx=ctx.needs_input_grad[0], has_bias=has_b)
        return dx, dw, db, None, None, None


class AddReluFunction(torch.autograd.Function):
    """Residual join relu(a + b) of a BasicBlock."""

    @staticmethod
    def forward(ctx, a, b):
        _need_gpu(a, b)
        y = lib().add_relu_fwd(_c(a), _c(b))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        g = lib().add_relu_bwd(y, _c(dy))
        return g, g


class MaxPool2Function(torch.autograd.Function):
    """2x2 max-pool (AdaptiveMaxPool2d((2,2)) on a 4x4 map)."""

    @staticmethod
    def forward(ctx, x):
        _need_gpu(x)
        x = _c(x)
        y, amax = lib().pool2_fwd(x)
        ctx.hw = x.shape[2:]
        ctx.save_for_backward(amax)
        return y

    @staticmethod
    def backward(ctx, dy):
        (amax,) = ctx.saved_tensors
        return lib().pool2_bwd(_c(dy), amax, *ctx.hw)


class BBBSampleFunction(torch.autograd.Function):
    """(mu, rho, eps) -> (w = mu + eps*softplus(rho), kl): mlhot_bbb_sample_fwd / _bwd."""

    @staticmethod
    def forward(ctx, mu, rho, eps):
        _need_gpu(mu, rho, eps)
        mu, rho, eps = _c(mu.detach()), _c(rho.detach()), _c(eps)
        w, kl = lib().bbb_sample_fwd(mu, rho, eps)
        ctx.save_for_backward(mu, rho, eps)
        return w, kl

    @staticmethod
    def backward(ctx, dw, dkl):
        mu, rho, eps = ctx.saved_tensors
        if dw is None:
            dw = torch.zeros_like(mu)
        if dkl is None:
            dkl = torch.zeros((), device=mu.device)
        dmu, drho = lib().bbb_sample_bwd(mu, rho, eps, _c(dw), _c(dkl.float()))
        return dmu, drho, None


class BBBSampleMultiFunction(torch.autograd.Function):
    """All weight / bias samples of a Bayes-by-backprop encoder in one launch pair (mlhot_bbb_sample_multi_fwd / _bwd).
    apply(eps_list, mu_0, rho_0, mu_1, rho_1, ...) -> (w_0, w_1, ..., kl); eps_list[i] is the device tensor of draw i.
    With 2k draws for k (mu, rho) pairs, the second k draws make a SECOND independent sample of every tensor in the same launch:
    -> (w_0 .. w_{k-1}, w2_0 .. w2_{k-1}, kl); the KL does not depend on eps and is computed once, and the backward folds both
    samples' gradients into d mu / d rho in one pass (no per-tensor accumulation kernels)."""

    @staticmethod
    def forward(ctx, eps_list, *mu_rho):
        _need_gpu(*mu_rho, *eps_list)
        mus = [_c(t.detach()) for t in mu_rho[0::2]]
        rhos = [_c(t.detach()) for t in mu_rho[1::2]]
        k = len(mus)
        if len(eps_list) not in (k, 2 * k):
            raise MlhotError(f"BBBSampleMultiFunction: {k} tensors need {k} or {2 * k} eps draws, got {len(eps_list)}")
        epss = [_c(e) for e in eps_list]
        ctx.k, ctx.two = k, len(epss) == 2 * k
        ctx.save_for_backward(*mus, *rhos, *epss)
        if ctx.two:
            ws, ws2, kl = lib().bbb_sample_multi_fwd(mus, rhos, epss[:k], epss[k:])
            return (*ws, *ws2, kl)
        ws, kl = lib().bbb_sample_multi_fwd(mus, rhos, epss)
        return (*ws, kl)

    @staticmethod
    def backward(ctx, *grads):
        k = ctx.k
        saved = ctx.saved_tensors
        mus, rhos, epss = saved[:k], saved[k:2 * k], saved[2 * k:]
        dws = [(_c(g) if g is not None else None) for g in grads[:k]]
        dws2 = [(_c(g) if g is not None else None) for g in grads[k:2 * k]] if ctx.two else None
        dkl = grads[-1]
        if dkl is None:
            dkl = torch.zeros((), device=mus[0].device)
        if ctx.two:
            dmus, drhos = lib().bbb_sample_multi_bwd(mus, rhos, epss[:k], dws, _c(dkl.float()), epss[k:], dws2)
        else:
            dmus, drhos = lib().bbb_sample_multi_bwd(mus, rhos, epss, dws, _c(dkl.float()))
        out = [None]
        for a, b in zip(dmus, drhos):
            out += [a, b]
        return tuple(out)


class ResNetTrunkFunction(torch.autograd.Function):
    """Every ResNet-trunk pass of a model step in ONE call per direction (mlhot_trunk_fwd / _bwd; csrc/resnet_trunk.h).
    apply(spec, *tensors): spec = (passes, skip_ks, n_imgs, taps) with passes = [(image tensor index, weight-set index)],
    skip_ks[w] = 1 | 3 the skip convolution of weight set w, n_imgs the number of leading image tensors, taps = None or a list
    that receives, per pass, the nine post-ReLU activations; tensors = the image batches [n, C, H, H], then 26 tensors per weight
    set (w, b of the stem and of (conv1, conv2, skip) of the four blocks).  Returns one output map [n, 64, H/32, H/32] per pass.
    Gradients flow to the weights only (images are leaves of the models)."""

    @staticmethod
    def forward(ctx, spec, *tensors):
        passes, skip_ks, n_imgs, taps = spec
        _need_gpu(*tensors)
        imgs = [_c(t.float()) for t in tensors[:n_imgs]]
        flat_w = [_c(t.detach()) for t in tensors[n_imgs:]]
        wsets = [(flat_w[26 * i:26 * i + 26], skip_ks[i]) for i in range(len(skip_ks))]
        acts = lib().trunk_fwd([(imgs[i], w) for i, w in passes], wsets)
        ctx.spec = (tuple(passes), tuple(skip_ks), n_imgs)
        ctx.n_acts = [len(a) for a in acts]
        ctx.save_for_backward(*imgs, *flat_w, *[a for ac in acts for a in ac])
        if taps is not None:
            taps.extend([list(ac) for ac in acts])
        return tuple(ac[8] for ac in acts)

    @staticmethod
    def backward(ctx, *dfeats):
        passes, skip_ks, n_imgs = ctx.spec
        saved = ctx.saved_tensors
        imgs, nw = saved[:n_imgs], 26 * len(skip_ks)
        flat_w, flat_a = saved[n_imgs:n_imgs + nw], saved[n_imgs + nw:]
        wsets = [(list(flat_w[26 * i:26 * i + 26]), skip_ks[i]) for i in range(len(skip_ks))]
        full, dfs = [], []
        for pi, (i, w) in enumerate(passes):
            acts = list(flat_a[9 * pi:9 * pi + 9])
            full.append((imgs[i], w, acts))
            df = dfeats[pi]
            dfs.append(_c(df) if df is not None else torch.zeros_like(acts[8]))
        grads = lib().trunk_bwd(full, wsets, dfs)
        return (None,) + (None,) * n_imgs + tuple(g for gs in grads for g in gs)


class BatchNormReluFunction(torch.autograd.Function):
    """Train-mode batch norm over dim 0 (+ fused ReLU), updating the running buffers in place like
    F.batch_norm(training=True): mlhot_bn_relu_fwd / _bwd."""

    @staticmethod
    def forward(ctx, x, gamma, beta, run_mean, run_var, momentum, eps):
        _need_gpu(x, gamma, beta, run_mean, run_var)
        x = _c(x)
        y, mean, var = lib().bn_relu_fwd(x, _c(gamma.detach()), _c(beta.detach()), run_mean, run_var, momentum, eps)
        ctx.eps = eps
        ctx.save_for_backward(x, y, _c(gamma.detach()), mean, var)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, mean, var = ctx.saved_tensors
        dx, dgamma, dbeta = lib().bn_relu_bwd(x, y, _c(dy), gamma, mean, var, ctx.eps)
        return dx, dgamma, dbeta, None, None, None, None


class SpatialMeanFunction(torch.autograd.Function):
    """[n, C, H, W] -> [n, C] mean over the spatial axes."""

    @staticmethod
    def forward(ctx, x):
        _need_gpu(x)
        x = _c(x)
        ctx.shape = tuple(x.shape)
        return lib().spatial_mean_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        return lib().spatial_mean_bwd(_c(dy), ctx.shape)
