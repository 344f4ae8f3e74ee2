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
        self.model = model
        # torch.optim's surface: one parameter group whose hyper-parameters step() reads (an LR scheduler writes param_groups[0]["lr"];
        # inside a replayed hipGraph the values are the ones of the capture)
        self.param_groups = [{"params": [p for _, p in model.named_parameters()], "lr": float(lr), "betas": tuple(betas), "eps": float(eps),
                              "weight_decay": float(weight_decay), "amsgrad": False, "maximize": False, "foreach": None,
                              "capturable": bool(capturable), "differentiable": False, "fused": None}]
        self.t = 0
        self.capturable = capturable
        total, offs, *rest = model.flat_layout(ctx_num, test_num)
        # floats the update covers: a layout may park parameters that never receive a gradient behind it (the ResNet family's
        # `resnet.fc.*`): torch.optim.Adam skips a parameter whose .grad is None - no moments, no weight decay - and so does this
        self.active = int(rest[0]) if rest else total
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

    lr = property(lambda self: self.param_groups[0]["lr"])
    betas = property(lambda self: self.param_groups[0]["betas"])
    eps = property(lambda self: self.param_groups[0]["eps"])
    weight_decay = property(lambda self: self.param_groups[0]["weight_decay"])

    def zero_grad(self, set_to_none=True):
        self.model.zero_grad(set_to_none=set_to_none)

    # ---- torch.optim.Adam's checkpoint layout --------------------------------------------------------------------------------
    def _steps_taken(self):
        return int(self.step_dev.item()) if self.capturable else self.t

    def state_dict(self):
        """The dictionary torch.optim.Adam(model.parameters()).state_dict() would hold after the same steps: per parameter index
        `step`, `exp_avg`, `exp_avg_sq` (copies of the flat buffers' slices), one param group."""
        names = [k for k, _ in self.model.named_parameters()]
        t = self._steps_taken()
        state = {}
        if t > 0:
            for i, (k, p) in enumerate(self.model.named_parameters()):
                if self.offsets[k] >= self.active:
                    continue                       # never stepped: torch's Adam holds no state for a parameter without gradients
                sl = slice(self.offsets[k], self.offsets[k] + p.numel())
                state[i] = {"step": torch.tensor(float(t)), "exp_avg": self.exp_avg[sl].view_as(p).clone(), "exp_avg_sq": self.exp_avg_sq[sl].view_as(p).clone()}
        group = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        group["params"] = list(range(len(names)))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        group = sd["param_groups"][0]
        for k in ("lr", "betas", "eps", "weight_decay"):
            self.param_groups[0][k] = tuple(group[k]) if k == "betas" else float(group[k])
        steps = set()
        self.exp_avg.zero_(); self.exp_avg_sq.zero_()
        for i, (k, p) in enumerate(self.model.named_parameters()):
            st = sd["state"].get(i)
            if st is None:
                continue
            sl = slice(self.offsets[k], self.offsets[k] + p.numel())
            self.exp_avg[sl].view_as(p).copy_(st["exp_avg"]); self.exp_avg_sq[sl].view_as(p).copy_(st["exp_avg_sq"])
            steps.add(int(st["step"]))
        if len(steps) > 1:
            raise ValueError("FlatAdam.load_state_dict: the parameters' step counts differ (one flat update has one count)")
        t = steps.pop() if steps else 0
        if self.capturable:
            self.step_dev.fill_(t)
        else:
            self.t = t

    @classmethod
    def from_torch_adam(cls, opt, model, ctx_num=15, test_num=15, capturable=True):
        """The FlatAdam that continues a plain torch.optim.Adam over exactly `model`'s parameters (train.py:52-56 builds that one:
        `torch.optim.Adam(model.parameters(), lr=config.lr[, weight_decay=config.beta])`) - same hyper-parameters, same moments and step
        count if it has already stepped - or None when the optimizer is anything else (a subclass, several groups, amsgrad / maximize,
        a tensor learning rate, parameters that are not the model's, a model without a flat gradient layout)."""
        if type(opt) is not torch.optim.Adam or len(opt.param_groups) != 1 or not hasattr(model, "flat_layout"):
            return None
        g = opt.param_groups[0]
        if g.get("amsgrad") or g.get("maximize") or g.get("differentiable") or torch.is_tensor(g["lr"]):
            return None
        mine = [p for _, p in model.named_parameters()]
        if len(mine) != len(g["params"]) or any(a is not b for a, b in zip(mine, g["params"])):
            return None
        if any(not p.is_cuda or p.dtype != torch.float32 for p in mine):
            return None
        old = opt.state_dict() if len(opt.state) else None
        new = cls(model, lr=g["lr"], betas=g["betas"], eps=g["eps"], weight_decay=g["weight_decay"], ctx_num=ctx_num, test_num=test_num,
                  capturable=capturable)
        if old is not None:
            new.load_state_dict(old)
        return new

    def _flat_grad(self):
        """The gradients as one tensor aligned with self.flat: the library's own buffer when backward produced the same
        layout (the normal case), otherwise a gathered copy (e.g. a step with an empty context uses another layout).

        Sets self._skip: the (offset, numel) ranges of stepped parameters WITHOUT a gradient this step.  torch.optim.Adam skips
        such a parameter - no moment decay, no weight decay, no move - and step() restores those ranges after the flat update, so
        that whatever their gradient slots hold (a mirrored arena is zeroed once, at refresh(): a parameter that had a gradient
        last step and has none now still shows last step's values there) never reaches the parameter.  The one thing a single
        flat update cannot mirror is torch's PER-PARAMETER step count: a skipped parameter's bias correction runs one step ahead
        of torch's from then on."""
        params = dict(self.model.named_parameters())
        self._skip = [(self.offsets[k], p.numel()) for k, p in params.items() if p.grad is None and self.offsets[k] < self.active]
        g0 = next((p.grad for p in params.values() if p.grad is not None), None)
        if g0 is None:
            return None
        st, base, ok = g0.untyped_storage(), None, True
        arena = getattr(self.model, "__dict__", {}).get("_arena")
        mirrored = arena is not None and arena.flat is not None and arena.flat.untyped_storage().data_ptr() == st.data_ptr()
        for k, p in params.items():
            g = p.grad
            if g is None and mirrored:
                continue                # its slot of the mirror holds zeros or a stale gradient: in self._skip, restored by step()
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

    def _assert_owned(self):
        """Every parameter must still be a view of self.flat: something that re-points `.data` (a HeadStack falling back to
        torch.cat, load_state_dict(assign=True), `.to()`) would leave the flat update training a buffer nobody reads."""
        lo, hi = self.flat.data_ptr(), self.flat.data_ptr() + 4 * self.flat.numel()
        for k, p in self.model.named_parameters():
            if not lo <= p.data_ptr() < hi:
                raise RuntimeError(f"FlatAdam: parameter {k} no longer lives in the optimizer's flat buffer (its .data was re-pointed)")

    def step(self, grad_scale=1.0):
        g = self._flat_grad()
        if g is None:
            return
        self._assert_owned()
        n = self.active
        flat, g, m1, m2 = (self.flat, g, self.exp_avg, self.exp_avg_sq) if n == self.flat.numel() else \
            (self.flat[:n], g[:n], self.exp_avg[:n], self.exp_avg_sq[:n])
        kept = None
        if self._skip:
            kept = [buf[a:a + k] for buf in (flat, m1, m2) for a, k in self._skip]
            kept = (kept, [v.clone() for v in kept])
        if self.capturable:
            lib().adam_step_counter(flat, g, m1, m2, self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, grad_scale, self.step_dev)
        else:
            self.t += 1
            lib().adam_step(flat, g, m1, m2, self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, grad_scale, self.t)
        if kept is not None:
            torch._foreach_copy_(*kept)
