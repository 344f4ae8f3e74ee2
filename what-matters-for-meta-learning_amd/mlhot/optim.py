"""Optimizer step on the flat buffers (reference: train.py:52-56 builds torch.optim.Adam
over ~70 small tensors and steps them one by one).

FlatAdam re-points every parameter of a vanilla CNP/ANP plugin at a view of ONE flat fp32 tensor laid out like
the library's flat gradient buffer (mlhot_np_grads_flat_layout), so that zero_grad / backward / all-reduce /
step touch two flat tensors and the update is ONE mlhot_adam_step launch.  Same arithmetic as torch.optim.Adam
(amsgrad off).  state_dict()/load_state_dict() of the module keep working: the views ARE its parameters.
"""
import torch

from . import lib


class FlatAdam:
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, ctx_num=15, test_num=15, capturable=False):
        """capturable: keep the step count on the device (mlhot_adam_step_counter) so that step() may sit inside a captured
        hipGraph and still advance the bias correction on every replay (torch.optim.Adam's `capturable` for the same reason)."""
        self.model, self.lr, self.betas, self.eps, self.weight_decay = model, lr, betas, eps, weight_decay
        self.t = 0
        self.capturable = capturable
        total, offs = model.flat_layout(ctx_num, test_num)
        params = dict(model.named_parameters())
        if set(offs) != set(params):
            raise ValueError("FlatAdam: the model's parameters and the library's gradient layout differ")
        dev = next(iter(params.values())).device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.offsets = offs
        with torch.no_grad():
            for k, p in params.items():
                view = self.flat[offs[k]:offs[k] + p.numel()].view_as(p)
                view.copy_(p)
                p.data = view
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev) if capturable else None
        self._gather = None

    def zero_grad(self, set_to_none=True):
        self.model.zero_grad(set_to_none=set_to_none)

    def _flat_grad(self):
        """The gradients as one tensor aligned with self.flat: the library's own buffer when backward produced the same
        layout (the normal case), otherwise a gathered copy (e.g. a step with an empty context uses another layout)."""
        params = dict(self.model.named_parameters())
        g0 = next((p.grad for p in params.values() if p.grad is not None), None)
        if g0 is None:
            return None
        st, base, ok = g0.untyped_storage(), None, True
        for k, p in params.items():
            g = p.grad
            if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.untyped_storage().data_ptr() != st.data_ptr():
                ok = False
                break
            b = g.storage_offset() - self.offsets[k]
            if base is None:
                base = b
            elif b != base:
                ok = False
                break
        if ok and base is not None and base >= 0 and (base + self.flat.numel()) * 4 <= st.nbytes():
            return torch.empty(0, dtype=torch.float32, device=self.flat.device).set_(st, base, (self.flat.numel(),))
        if self._gather is None:
            self._gather = torch.zeros_like(self.flat)
        live = [(k, p) for k, p in params.items() if p.grad is not None]
        self._gather.zero_()
        torch._foreach_copy_([self._gather[self.offsets[k]:self.offsets[k] + p.numel()].view_as(p) for k, p in live],
                             [p.grad for _, p in live])
        return self._gather

    def step(self, grad_scale=1.0):
        g = self._flat_grad()
        if g is None:
            return
        if self.capturable:
            lib().adam_step_counter(self.flat, g, self.exp_avg, self.exp_avg_sq, self.lr, self.betas[0], self.betas[1], self.eps,
                                    self.weight_decay, grad_scale, self.step_dev)
            return
        self.t += 1
        lib().adam_step(self.flat, g, self.exp_avg, self.exp_avg_sq, self.lr, self.betas[0], self.betas[1], self.eps,
                        self.weight_decay, grad_scale, self.t)
