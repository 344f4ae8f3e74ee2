"""One flat gradient buffer for models whose gradients come out of several C calls (the ResNet / Bayes-by-backprop family).

The reference has ONE optimizer over all parameters (train.py:52-56); data-parallel training all-reduces all their gradients once
per step.  The vanilla models' single backward call already writes every gradient into one flat buffer
(mlhot_np_grads_flat_layout).  The ResNet-family models' gradients are produced by ~15 calls (trunk backward, linears, head stacks,
Bayes-by-backprop sampling): without an arena each call allocates its own tensors and mlhot.dist.GradBucket packs / unpacks them
with two _foreach_copy_ passes over <= 15 MB per step.

GradArena mirrors the STORAGES of a model's parameters in one fp32 buffer: the gradient slot of parameter p sits at the same
offset inside its storage's segment as p inside its storage - so tensors that several parameters are views of (the stacked
per-head weights, networks/_resnet_np.py::HeadStack) get ONE contiguous gradient the kernels write as a whole.  While an arena is
installed (mlhot.binding.set_grad_arena), every binding call that is about to allocate a weight / bias gradient asks it for the
slot of the tensor the gradient belongs to (looked up by storage pointer + offset: detached views of a parameter hit, temporaries
such as sampled Bayes-by-backprop weights miss and get a fresh tensor as before).  autograd then hands the kernels' output views to
the parameters' .grad as they are, GradBucket finds every live gradient inside one storage and all-reduces the range in place.
`first`: parameters whose segments come first (GradBucket's early bucket: one contiguous range of its own)."""
import torch


class GradArena:
    def __init__(self, params, first=None):
        params = [p for p in params if p.requires_grad]
        head = [p for p in (first or []) if p.requires_grad]
        ids = {id(p) for p in head}
        self.params = head + [p for p in params if id(p) not in ids]
        self.n_first = len(head)
        self.flat, self._base, self._sig, self.first_numel = None, {}, None, 0

    def _signature(self):
        return tuple((p.untyped_storage().data_ptr(), p.storage_offset(), p.numel(), str(p.device)) for p in self.params)

    def refresh(self):
        """(Re)build the layout when a parameter's storage changed (`.to(device)`, load_state_dict(assign=True), a re-stacked
        HeadStack).  Cheap when nothing changed: one tuple comparison per step."""
        sig = self._signature()
        if sig == self._sig:
            return self
        order, rng, first_keys = [], {}, set()
        for i, p in enumerate(self.params):
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise ValueError("GradArena: parameters must be contiguous fp32 tensors")
            key = p.untyped_storage().data_ptr()
            lo, hi = p.storage_offset(), p.storage_offset() + p.numel()
            if key not in rng:
                rng[key] = [lo, hi]
                order.append(key)
                if i < self.n_first:
                    first_keys.add(key)
            else:
                rng[key][0], rng[key][1] = min(rng[key][0], lo), max(rng[key][1], hi)
        total, base, self._range = 0, {}, {}
        for part in (True, False):                       # the `first` parameters' storages, then the rest
            for key in order:
                if (key in first_keys) != part:
                    continue
                lo, hi = rng[key]
                base[key] = total - lo                   # flat index of the storage's element 0
                self._range[key] = (lo, hi)
                total += (hi - lo + 3) // 4 * 4          # every segment 16-byte aligned
            if part:
                self.first_numel = total
        self.flat = torch.zeros(total, dtype=torch.float32, device=self.params[0].device)
        self._base, self._sig = base, sig
        return self

    def slot(self, t):
        """The gradient slot of `t` (a parameter, a detached view of one, or a tensor several parameters are views of), or None."""
        if self.flat is None or t is None or t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.flat.device:
            return None
        key = t.untyped_storage().data_ptr()
        b = self._base.get(key)
        if b is None:
            return None
        lo, hi = self._range[key]
        a, e = t.storage_offset(), t.storage_offset() + t.numel()
        if a < lo or e > hi:
            return None
        return self.flat[b + a:b + e].view(t.shape)
