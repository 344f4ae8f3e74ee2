"""Host logic of the trainer / evaluator mirrors (trainer/model_trainer.py:33-143, evaluator/model_evaluator.py:95-179 of the
reference) with a stand-in model on the CPU: iteration / validation cadence, files written, draw order of the data source,
context-size sweep.  The numerics of the real models are the GPU parity tests' business."""
import os
import types

import numpy as np
import pytest
import torch

from mlhot import synth


class LossFunc:
    """Stand-in with trainer.losses.LossFunc's call contract (the real one is a HIP kernel and refuses CPU tensors)."""

    def __init__(self, *a):
        pass

    def calc_loss(self, mu, var, gt, test=False):
        d = ((gt[..., :2] - mu) ** 2).sum(dim=-1)
        return d.sqrt().mean() if test else d.mean()


class TinyModel(torch.nn.Module):
    """Same call contract as the plugins: (ctx_x, ctx_y, qry_x, test=False) -> (mu [T, Nq, 2], None, 0)."""

    def __init__(self):
        super().__init__()
        self.lin = torch.nn.Linear(4, 2)
        self.calls = []

    def forward(self, ctx_x, ctx_y, qry_x, test=False):
        self.calls.append((ctx_x.shape[1], qry_x.shape[1], bool(test), self.training))
        feat = torch.stack([qry_x.mean(dim=(2, 3, 4)), qry_x.amax(dim=(2, 3, 4)), ctx_x.mean().expand(qry_x.shape[:2]),
                            ctx_y[..., 0].mean().expand(qry_x.shape[:2])], dim=-1)
        return torch.tanh(self.lin(feat)), None, 0


class CountingData(synth.SyntheticData):
    def __init__(self):
        super().__init__()
        self.log = []

    def get_batch(self, source, tasks_per_batch, shot):
        self.log.append((source, shot))
        xs, xq, ys, yq = super().get_batch(source, tasks_per_batch, shot)
        return xs[..., ::16, ::16], xq[..., ::16, ::16], ys, yq        # small images: this is a host-logic test


def _cfg(tmp_path, **kw):
    base = dict(device=torch.device("cpu"), tasks_per_batch=2, task="shapenet_1d", iterations=4, val_freq=2, val_iters=2, bg_gen_freq=1000,
                gen_bg=False, max_ctx_num=5, beta=0, contrastive=False, save_path=str(tmp_path / "run"), logger=None)
    base.update(kw)
    return types.SimpleNamespace(**base)


def test_trainer_cadence_and_files(tmp_path):
    from trainer.model_trainer import ModelTrainer
    model, data = TinyModel(), CountingData()
    cfg = _cfg(tmp_path)
    tr = ModelTrainer(model=model, loss=LossFunc("mse", "shapenet_1d"), optimizer=torch.optim.Adam(model.parameters(), lr=1e-2), config=cfg,
                      data=data)
    assert tr.ingest is None                                    # CPU device: the reference's host route
    w0 = model.lin.weight.detach().clone()
    tr.train()
    # 4 training iterations, validation + test after iterations 2 and 4, val_iters batches each
    assert [s for s, _ in data.log] == ["train", "train"] + ["validation"] * 2 + ["test"] * 2 + ["train", "train"] + ["validation"] * 2 + ["test"] * 2
    train_calls = [c for c in model.calls if c[3]]
    assert len(train_calls) == 4 and all(3 <= c[0] <= 5 and c[1] == 5 and not c[2] for c in train_calls)     # random context size, fixed targets
    eval_calls = [c for c in model.calls if not c[3]]
    assert len(eval_calls) == 8 and all(c[0] == 5 and c[2] for c in eval_calls)
    for f in ("models/model_end_4.pt", "models/best_validation_model.pt", "models/best_test_model.pt", "best_validation_error.txt"):
        assert os.path.exists(os.path.join(cfg.save_path, f)), f
    assert not torch.equal(w0, model.lin.weight)
    # best_<source>_error.txt: the reference's three writes per improvement (trainer/model_trainer.py:135-138), real newlines
    lines = open(os.path.join(cfg.save_path, "best_validation_error.txt")).read().split("\n")
    assert lines[0] == "Best Step: 2 " and lines[1] == "Best validation Loss: " and lines[2].startswith("tensor(")
    assert lines[3] == "Best validation Loss std: " and lines[4].startswith("tensor(") and len(lines) % 5 == 1 and lines[-1] == ""
    assert tr.best_loss["validation"] < 50000 and tr.best_loss["test"] < 20000        # the reference's initial values were beaten


def test_evaluator_sweep_and_files(tmp_path):
    from evaluator.model_evaluator import ModelEvaluator
    model, data = TinyModel(), CountingData()
    cfg = _cfg(tmp_path, iterations=0, val_iters=3, max_ctx_num=4)
    ev = ModelEvaluator(model=model, loss=LossFunc("mse", "shapenet_1d"), config=cfg, data=data)
    val, test = ev.evaluate()
    # context sizes 1..4, validation then test per size (the reference's interleaving), val_iters batches each, Nc = Nq = size
    assert data.log == [(src, n) for n in range(1, 5) for src in ("validation", "test") for _ in range(3)]
    assert all(c == (n, n, True, False) for c, n in zip(model.calls, [n for n in range(1, 5) for _ in range(6)]))
    for name, res in (("val_losses.txt", val), ("test_losses.txt", test)):
        table = np.loadtxt(os.path.join(cfg.save_path, name))
        assert table.shape == (4, 3) and list(table[:, 0]) == [1, 2, 3, 4]
        assert np.allclose(table[:, 1], res[0], atol=1e-4) and np.allclose(table[:, 2], res[1], atol=1e-4)
    assert os.path.exists(os.path.join(cfg.save_path, "models", "model.pt"))
    # the sweep re-seeds the split generators per call (model_evaluator.py:152-159): a second sweep sees the same batches
    val2, _ = ModelEvaluator(model=model, loss=LossFunc("mse", "shapenet_1d"), config=cfg, data=data).evaluate()
    assert val2 == val
    # pascal_1d has no test split
    cfg_p = _cfg(tmp_path, iterations=0, val_iters=1, max_ctx_num=2, task="shapenet_1d", save_path=str(tmp_path / "p"))
    cfg_p.task = "pascal_1d"
    data_p = CountingData()
    ModelEvaluator(model=TinyModel(), loss=LossFunc("mse", "shapenet_1d"), config=cfg_p, data=data_p).evaluate()
    assert [s for s, _ in data_p.log] == ["validation", "validation"] and not os.path.exists(tmp_path / "p" / "test_losses.txt")


def test_contrastive_models_get_the_target_labels(tmp_path):
    """config.contrastive (model_trainer.py:72-81, 118-119 of the reference): 4-argument call, 4-tuple return, the contrastive term
    scaled by contrastive_rate joins the loss; validation passes test=True."""
    from trainer.model_trainer import ModelTrainer
    seen = []

    class FclModel(TinyModel):
        def forward(self, ctx_x, ctx_y, qry_x, qry_y, test=False):
            seen.append((tuple(qry_y.shape), bool(test)))
            mu, var, kl = super().forward(ctx_x, ctx_y, qry_x, test=test)
            return mu, var, kl, (0 if test else (mu ** 2).mean())

    model = FclModel()
    cfg = _cfg(tmp_path, iterations=2, val_freq=2, val_iters=1, contrastive=True, contrastive_rate=0.25)
    ModelTrainer(model=model, loss=LossFunc("mse", "shapenet_1d"), optimizer=torch.optim.SGD(model.parameters(), lr=1e-2), config=cfg,
                 data=CountingData()).train()
    assert [t for _, t in seen] == [False, False, True, True] and all(s == (2, 5, 3) for s, _ in seen)


def test_single_validation_batch_writes_nan_std_like_the_reference(tmp_path):
    """val_iters = 1: torch.std of one value is nan, which is what the reference writes (trainer/model_trainer.py:124,138)."""
    from trainer.model_trainer import ModelTrainer
    model = TinyModel()
    cfg = _cfg(tmp_path, iterations=1, val_freq=1, val_iters=1)
    ModelTrainer(model=model, loss=LossFunc("mse", "shapenet_1d"), optimizer=torch.optim.SGD(model.parameters(), lr=1e-2), config=cfg,
                 data=CountingData()).train()
    text = open(os.path.join(cfg.save_path, "best_test_error.txt")).read()
    assert "Best test Loss std: \ntensor(nan)\n" in text and "\\n" not in text


def test_grad_bucket_deferred_scale_and_overridable_collective():
    """GradBucket.sync(defer_scale=True) leaves the summed gradients in place and returns 1/world for the optimizer's gradient
    scale; the collective itself is one overridable method (here: a stand-in for two identical ranks)."""
    from mlhot.dist import GradBucket

    class TwoRanks(GradBucket):
        def world_size(self):
            return 2

        def _all_reduce(self, flat):
            flat.mul_(2.0)

    lin = torch.nn.Linear(3, 2)
    for defer in (False, True):
        lin.weight.grad, lin.bias.grad = torch.ones(2, 3), torch.full((2,), 3.0)
        b = TwoRanks(lin.parameters())
        scale = b.sync(defer_scale=defer)
        assert scale == (0.5 if defer else 1.0)
        assert torch.equal(lin.weight.grad * scale, torch.ones(2, 3)) and torch.equal(lin.bias.grad * scale, torch.full((2,), 3.0))


def test_strict_sharded_parity_option_installs_the_exchange_and_refuses_graph_steps(tmp_path):
    """config.strict_sharded_parity: ModelTrainer installs mlhot.dist.StabiliserExchange as the attention passes' exchange
    (mlhot.ops.set_stabiliser_exchange); together with config.graph_steps it is refused (the collective sits between two C calls)."""
    from mlhot import ops
    from mlhot.dist import StabiliserExchange
    from trainer.model_trainer import ModelTrainer
    model = TinyModel()
    mk = lambda **kw: ModelTrainer(model=model, loss=LossFunc("mse", "shapenet_1d"), optimizer=torch.optim.SGD(model.parameters(), lr=1e-2),
                                   config=_cfg(tmp_path, **kw), data=CountingData())
    try:
        assert ops._stab_exchange is None
        mk()
        assert ops._stab_exchange is None
        mk(strict_sharded_parity=True)
        assert isinstance(ops._stab_exchange, StabiliserExchange)
        with pytest.raises(ValueError):
            mk(strict_sharded_parity=True, graph_steps=True)
    finally:
        ops.set_stabiliser_exchange(None)


def test_mt19937_jump_ahead_polynomials_against_brute_force():
    """mlhot/mt_jump.py (host side of the parallel device eps stream): Berlekamp-Massey recovers MT19937's characteristic polynomial
    (degree 19937, 135 terms), and the jump polynomials t^(624 S k) mod phi, applied to the raw output window as the device kernel
    applies them, reproduce the block the generator holds S k regenerations later - against plain regeneration (oracle/mt_normal.py,
    itself pinned to torch's generator)."""
    import numpy as np
    from mlhot import mt_jump as J
    from oracle import mt_normal as MT
    phi = J.char_poly()
    assert phi.bit_length() - 1 == 19937 and bin(phi).count("1") == 135
    S = 3
    polys = J.jump_polys(S, 5)
    rng = np.random.RandomState(3)
    st = rng.randint(0, 2 ** 32, size=624, dtype=np.uint64).astype(np.uint32)
    blocks = [st]
    for _ in range(34):
        blocks.append(MT.next_state(blocks[-1]))
    window = np.concatenate(blocks[:34])[:J.DEG + J.N]
    for k in range(1, 6):
        got, want = J.apply_poly(polys[k - 1], window), blocks[S * k]
        assert np.array_equal(got[1:], want[1:]) and (got[0] >> 31) == (want[0] >> 31), k
    # the regeneration of a jumped block does not depend on the undefined low bits of its word 0
    a = blocks[S * 2].copy()
    a[0] ^= np.uint32(0x7fffffff)
    assert np.array_equal(MT.next_state(a), blocks[S * 2 + 1])


def test_flat_adam_promotion_refuses_what_it_cannot_continue():
    """mlhot.optim.FlatAdam.from_torch_adam only continues a plain torch.optim.Adam over exactly the model's parameters (train.py:52-56's
    optimizer); anything else - and CPU parameters, where the flat update kernel cannot run - is left alone (None), without touching
    the library."""
    import torch
    from mlhot.optim import FlatAdam

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.b = torch.nn.Linear(3, 3), torch.nn.Linear(3, 2)

        def flat_layout(self, *a):
            raise AssertionError("must not be reached for a refused optimizer")

    m = M()
    assert FlatAdam.from_torch_adam(torch.optim.AdamW(m.parameters(), lr=1e-3), m) is None
    assert FlatAdam.from_torch_adam(torch.optim.SGD(m.parameters(), lr=1e-3), m) is None
    assert FlatAdam.from_torch_adam(torch.optim.Adam([{"params": m.a.parameters()}, {"params": m.b.parameters()}], lr=1e-3), m) is None
    assert FlatAdam.from_torch_adam(torch.optim.Adam(m.parameters(), lr=1e-3, amsgrad=True), m) is None
    assert FlatAdam.from_torch_adam(torch.optim.Adam(m.a.parameters(), lr=1e-3), m) is None          # not all of the model's parameters
    assert FlatAdam.from_torch_adam(torch.optim.Adam(m.parameters(), lr=torch.tensor(1e-3)), m) is None
    assert FlatAdam.from_torch_adam(torch.optim.Adam(m.parameters(), lr=1e-3), m) is None             # CPU parameters
    assert FlatAdam.from_torch_adam(torch.optim.Adam(m.parameters(), lr=1e-3), torch.nn.Linear(2, 2)) is None   # no flat gradient layout


@pytest.mark.parametrize("method", ["ANPMRShapeNet3D", "ANP", "CondNeuralProcess"])
def test_resnet_family_flat_layout_keeps_buckets_and_head_stacks_contiguous(method):
    """ResNetNP.flat_layout (what FlatAdam lays the ResNet / Bayes-by-backprop models' parameters out by, train.py:52-56's one optimizer
    over all parameters): early-bucket parameters first, each per-head stack one block that HeadStack adopts without a copy, the
    trunks behind, `resnet.fc.*` (never given a gradient) parked behind the stepped range; GradArena mirrors it offset for offset."""
    import importlib
    from mlhot.arena import GradArena
    cfg = types.SimpleNamespace(device=torch.device("cpu"), seed=2578, img_size=[64, 64, 4], tasks_per_batch=2, input_dim=4, output_dim=4,
                                agg_mode="attention" if method != "CondNeuralProcess" else "max", img_agg="reshape" if method == "ANPMRShapeNet3D" else "max",
                                task="shapenet_3d", temperature=0.07)
    model = getattr(importlib.import_module("networks." + method), method)(cfg)
    named = dict(model.named_parameters())
    total, offs, active = model.flat_layout()
    assert set(offs) == set(named) and all(o % 4 == 0 for o in offs.values()) and total % 4 == 0
    spans = sorted((o, o + named[n].numel(), n) for n, o in offs.items())
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])) and spans[-1][1] <= total              # no overlap
    early = {n for n, p in named.items() if any(p is q for q in model.early_grad_parameters())}
    dead = {n for n in named if ".resnet.fc." in n}
    assert dead and "decoder.conv1.weight" not in early and "decoder.fc_mu.0.weight" in early
    assert max(offs[n] + named[n].numel() for n in early) <= min(offs[n] for n in named if n not in early)
    assert min(offs[n] for n in dead) == active and max(offs[n] + named[n].numel() for n in named if n not in dead) <= active
    # what FlatAdam.__init__ does, on the CPU: every parameter a view of one flat tensor
    before = {n: p.detach().clone() for n, p in named.items()}
    flat = torch.zeros(total)
    with torch.no_grad():
        for n, p in named.items():
            v = flat[offs[n]:offs[n] + p.numel()].view_as(p)
            v.copy_(p)
            p.data = v
    if model.ATTENTION:
        for prefix, mods in (("_W_q", model._W_q), ("_W_k", model._W_k), ("_W_v", model._W_v)):
            w, b = model._stack(mods).tensors()
            assert w.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() and w.shape == (8 * 256, 256) and b.shape == (8 * 256,)
            assert all(m.linear.weight.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() for m in mods)   # nothing was pulled out again
            assert torch.equal(w, torch.cat([before[f"{prefix}.{i}.linear.weight"] for i in range(8)]))
            assert torch.equal(b, torch.cat([before[f"{prefix}.{i}.linear.bias"] for i in range(8)]))
    assert all(torch.equal(p, before[n]) for n, p in named.items())
    arena = GradArena(model.parameters(), first=model.early_grad_parameters()).refresh()
    assert arena.flat.numel() == total
    for n, p in named.items():
        g = arena.slot(p)
        assert g is not None and g.shape == p.shape and g.storage_offset() == offs[n]


def test_batches_drawn_ahead_never_cross_a_validation_round_or_a_background_regeneration():
    """trainer.ModelTrainer._clear_ahead: how many training batches may be drawn under step `it` (host_prefetch_depth, default 2).
    The reference draws train_k, [validation / test of k], [gen_bg(k + 1)], train_k+1 (model_trainer.py:59-70, train loop): batch
    it + j may be drawn early only if no iteration it .. it + j - 1 ends in a validation round, is the last one, or is followed by a
    background regeneration.  Replayed here against a literal walk of the reference's order of events."""
    from trainer.model_trainer import ModelTrainer
    for depth, val_freq, bg, gen_bg, iters in ((2, 3, 1000, False, 10), (2, 4, 5, True, 23), (1, 2, 3, True, 9), (3, 5, 7, True, 40), (2, 1, 1, True, 6)):
        stub = types.SimpleNamespace(config=types.SimpleNamespace(host_prefetch_depth=depth, val_freq=val_freq, bg_gen_freq=bg, gen_bg=gen_bg),
                                     iterations=iters)
        for it in range(1, iters + 1):
            n = ModelTrainer._clear_ahead(stub, it)
            assert 0 <= n <= depth
            # literal walk: the events between "train_it has been drawn" and "train_it+j is drawn"
            ok = 0
            for j in range(1, depth + 1):
                last = it + j - 1                               # the iteration whose tail lies in front of draw it + j
                blocked = last >= iters or last % val_freq == 0 or (gen_bg and (last + 1) % bg == 0)
                if blocked:
                    break
                ok = j
            assert n == ok, (depth, val_freq, bg, gen_bg, it, n, ok)
