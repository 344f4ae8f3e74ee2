"""N>1 path on CPU: two gloo ranks, tasks sharded contiguously, ONE flat-bucket gradient
all-reduce (mlhot.dist.GradBucket) - the averaged shard gradients must equal the gradient of the
un-sharded batch.  The per-rank compute here is the CPU oracle (checker), since the
product's kernels only run on a GPU; what is under test is the sharding + bucket + collective."""
import json
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import util as U


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, name, out, shared=False):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [os.path.join(root, "what-matters-for-meta-learning_amd"), root]
    from oracle import ref_cpu as O
    from mlhot import dist as mdist
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    r, _, w = mdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    fx, meta = U.load_case(name)
    model = U.build_model(meta, fx=fx)          # every rank: same seed -> same weights
    cx, qx, cy, qy = U.case_inputs(meta)
    task = meta["cfg"]["task"]

    def grads_of(sl):
        p = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "projection" not in k) for k, v in model.state_dict().items()}
        mu = O.vanilla_np_forward(p, cx[sl], cy[sl], qx[sl], meta["cfg"]["agg_mode"], tanh=model.OUT_TANH)
        O.calc_loss(task, mu, qy[sl]).backward()
        return {k: p[k].grad for k, _ in model.named_parameters()}

    T = cx.shape[0]
    full = grads_of(slice(0, T))
    local = grads_of(mdist.task_slice(T, rank, world))
    if shared:
        # the product hands out every gradient as a view of ONE flat buffer (mlhot_np_grads_flat_layout, with alignment
        # padding between tensors): the bucket must reduce that buffer in place, without packing
        sizes = [(k, prm.numel()) for k, prm in model.named_parameters()]
        flat = torch.full((sum((n + 3) // 4 * 4 for _, n in sizes) + 8,), float("nan"))
        off = 4
        for (k, n), (_, prm) in zip(sizes, model.named_parameters()):
            view = flat[off:off + n].view_as(prm)
            view.copy_(local[k])
            prm.grad = view
            off += (n + 3) // 4 * 4
    else:
        for k, prm in model.named_parameters():
            prm.grad = local[k].clone()
    bucket = mdist.GradBucket(model.parameters())
    bucket.sync()
    if shared:
        assert bucket.flat is None and all(prm.grad.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() for prm in model.parameters())
    else:
        assert bucket.flat.numel() == sum(p.numel() for p in model.parameters())
    floor = U.GRAD_FLOOR * max(g.abs().max().item() for g in full.values())   # same floor as the parity tests
    worst = max(U.rel_err(prm.grad, full[k], floor=floor) for k, prm in model.named_parameters())
    out[rank] = worst
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,shared", [("s_cnp_shapenet1d_baco", False), ("s_anp_shapenet1d_ragged", False), ("s_anp_shapenet1d_ragged", True)])
def test_two_rank_task_sharding_matches_full_batch(name, shared):
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, name, out, shared)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            assert p.exitcode == 0
        # sharded-vs-full differs only through the FAVOR+ batch-global key stabiliser
        assert len(out) == 2 and max(out.values()) <= 1e-4, dict(out)


def test_task_slice():
    from mlhot.dist import task_slice
    assert [task_slice(128, r, 8) for r in (0, 7)] == [slice(0, 16), slice(112, 128)]
    with pytest.raises(ValueError):
        task_slice(10, 0, 4)


# ---- the Bayes-by-backprop model under task sharding ------------------------------------------------------------
def _bbb_worker(rank, world, port, out):
    """Every rank seeds the torch CPU generator alike, so all ranks draw the SAME eps (hence sample the same weights), and the KL
    term - identical on every rank - is averaged by the flat all-reduce, never summed: the sharded gradient of mean-loss + beta * kl
    equals the full-batch gradient.  (Per-rank compute = the CPU oracle; under test: seeding rule + sharding + bucket.)"""
    import hashlib
    import sys
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [os.path.join(root, "what-matters-for-meta-learning_amd"), root]
    from oracle import ref_cpu as O
    from mlhot import dist as mdist
    from networks.ANPMRShapeNet3D import ANPMRShapeNet3D
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    mdist.init_from_env("gloo")
    T, Nc, Nq, beta = 2, 2, 2, 1e-5
    cfg = types.SimpleNamespace(device=torch.device("cpu"), seed=2578, img_size=[64, 64, 4], tasks_per_batch=T // world, input_dim=4, output_dim=4,
                                agg_mode="attention", img_agg="reshape", task="shapenet_3d", temperature=0.07)
    model = ANPMRShapeNet3D(cfg)                       # same seed -> same weights on every rank
    g = torch.Generator().manual_seed(5)
    cx, qx = torch.rand(T, Nc, 3, 64, 64, generator=g), torch.rand(T, Nq, 3, 64, 64, generator=g)
    cy = torch.nn.functional.normalize(torch.randn(T, Nc, 4, generator=g), dim=-1)
    qy = torch.nn.functional.normalize(torch.randn(T, Nq, 4, generator=g), dim=-1)

    def grads_of(sl):
        p = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "projection" not in k) for k, v in model.state_dict().items()}
        torch.manual_seed(1234)                        # the rule under test: the same CPU-generator seed on every rank
        first = torch.empty(5).normal_(0, 1)
        torch.manual_seed(1234)
        mu, kl = O.anpmr3d_forward(p, cx[sl], cy[sl], qx[sl])
        (O.calc_loss("shapenet_3d", mu, qy[sl]) + beta * kl).backward()
        return {k: p[k].grad for k, _ in model.named_parameters()}, kl.item(), hashlib.sha256(first.numpy().tobytes()).hexdigest()

    full, kl_full, _ = grads_of(slice(0, T))
    local, kl_local, eps_sha = grads_of(mdist.task_slice(T, rank, world))
    assert abs(kl_local - kl_full) <= 1e-6 * kl_full          # the KL does not depend on the shard
    for k, prm in model.named_parameters():
        prm.grad = local[k].clone() if local[k] is not None else None
    mdist.GradBucket(model.parameters()).sync()
    live = [k for k, prm in model.named_parameters() if prm.grad is not None]
    floor = U.GRAD_FLOOR * max(full[k].abs().max().item() for k in live)
    # sharded-vs-full differs only through the FAVOR+ batch-global key stabiliser
    worst = max(U.rel_err(dict(model.named_parameters())[k].grad, full[k], floor=floor) for k in live)
    # had the KL been SUMMED over the ranks, W_rho's gradient (dominated by beta * dkl at init) would be off by ~2x
    rho = "img_encoder.net.layer2.conv1.W_rho"
    out[rank] = (worst, eps_sha, U.rel_err(dict(model.named_parameters())[rho].grad, full[rho]))
    dist.barrier()
    dist.destroy_process_group()


def test_bbb_model_sharding_same_eps_on_every_rank_and_kl_not_summed():
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_bbb_worker, args=(r, 2, port, out)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(600)
            assert p.exitcode == 0
        assert len(out) == 2
        assert out[0][1] == out[1][1]                              # identical eps draws
        assert max(v[0] for v in out.values()) <= 1e-4 and max(v[2] for v in out.values()) <= 1e-4, dict(out)


# ---- bench.py's multi-rank control flow on 8 CPU ranks (the 8-GPU RCCL launch is the driver's; its protocol is checked here) ----
def _bench_flow_worker(rank, world, port, out):
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [os.path.join(root, "what-matters-for-meta-learning_amd"), root]
    import bench
    from mlhot import dist as mdist
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    r, local, w = mdist.init_from_env("gloo")
    assert (r, local, w) == (rank, rank, world)
    calls = []
    grad = torch.full((8,), float(rank + 1))

    class Bucket(mdist.GradBucket):                  # the bench's collective on a stand-in gradient
        pass

    lin = torch.nn.Linear(7, 1)

    def run():                                         # a step: "compute" (the slowest rank sets the pace), then the flat all-reduce
        time.sleep(0.002 * (1 + (rank == 3)))
        lin.weight.grad, lin.bias.grad = grad[:7].clone().view(1, 7), grad[7:].clone()
        scale = Bucket(lin.parameters()).sync(defer_scale=True)
        calls.append(scale)
        return lin.weight.grad

    elapsed, enq, last = bench.timed_region(run, steps=5, warmup=2, world=world, device=torch.device("cpu"))
    assert len(calls) == 7 and all(abs(c - 1.0 / world) < 1e-12 for c in calls)
    assert torch.allclose(last, torch.full((1, 7), float(sum(range(1, world + 1)))))          # summed; the 1/world scale is deferred
    out[rank] = elapsed
    dist.barrier()
    dist.destroy_process_group()


def test_bench_timing_protocol_on_eight_ranks():
    """bench.timed_region under 8 gloo ranks: W warm-up + exactly K timed steps between fences, the reported time is the MAX over
    the ranks (every rank returns the same number, at least the slowest rank's compute), one deferred-scale all-reduce per step."""
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_bench_flow_worker, args=(r, 8, port, out)) for r in range(8)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(600)
            assert p.exitcode == 0
        vals = [out[r] for r in range(8)]
        assert max(vals) - min(vals) < 1e-9 and vals[0] >= 5 * 0.004


# ---- bench.py --gpus N without a launcher: spawn_ranks (no GPU needed: the children are tiny python programs) ---------------------
_SPAWN_CHILD = r"""
import json, os, sys
import torch, torch.distributed as dist
r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert int(os.environ["LOCAL_RANK"]) == r and os.environ["MASTER_ADDR"] == "127.0.0.1" and os.environ["NCCL_DEBUG"] == "VERSION"
dist.init_process_group("gloo", rank=r, world_size=w)          # MASTER_PORT: one free port, the same for every rank
t = torch.tensor([float(r + 1)])
dist.all_reduce(t)
print("rank %d says hello" % r)                                 # the other ranks' stdout must NOT reach the parent's stdout
if r == 0:
    print(json.dumps({"n_gpus": dist.get_world_size(), "sum": t.item(), "backend": dist.get_backend()}))
dist.barrier()
dist.destroy_process_group()
sys.exit(int(os.environ.get("FAIL_RANK", "-1")) == r and 7 or 0)
"""


def test_bench_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` with no WORLD_SIZE: bench.spawn_ranks starts N processes with the launcher's environment, relays
    rank 0's stdout only, returns the worst exit code, and does not hang on a rank that dies."""
    import io
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root]
    import bench
    child = tmp_path / "child.py"
    child.write_text(_SPAWN_CHILD)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "NCCL_DEBUG")}
    out, err = io.StringIO(), io.StringIO()
    rc = bench.spawn_ranks(3, [sys.executable, str(child)], env=env, out=out, err=err)
    assert rc == 0, err.getvalue()
    lines = out.getvalue().splitlines()
    assert len(lines) == 1 and json.loads(lines[0]) == {"n_gpus": 3, "sum": 6.0, "backend": "gloo"}, lines
    assert all(f"[rank {r}] rank {r} says hello" in err.getvalue() for r in range(3))
    # the worst exit code wins
    out, err = io.StringIO(), io.StringIO()
    assert bench.spawn_ranks(2, [sys.executable, str(child)], env=dict(env, FAIL_RANK="1"), out=out, err=err) == 7
    # a rank that dies before the rendezvous: the survivor is terminated after the grace period instead of waiting out gloo's timeout
    dies = tmp_path / "dies.py"
    dies.write_text("import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(3)\ntime.sleep(600)\n")
    t0 = time.monotonic()
    rc = bench.spawn_ranks(2, [sys.executable, str(dies)], env=env, grace_s=1.0, out=io.StringIO(), err=io.StringIO())
    assert rc == 128 + 15 and time.monotonic() - t0 < 30          # SIGTERM of the sleeping rank outranks the exit code 3


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    """Under a launcher whose WORLD_SIZE differs from --gpus bench.py exits with a message and code 2 (it used to be an assert)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE is 1" in r.stderr and "Traceback" not in r.stderr


# ---- the early bucket: its all-reduce leaves from inside backward(), before the late gradients exist -----------------------------
def _early_worker(rank, world, port, out):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [os.path.join(root, "what-matters-for-meta-learning_amd"), root]
    from mlhot import dist as mdist
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    mdist.init_from_env("gloo")
    torch.manual_seed(0)
    late = torch.nn.Linear(8, 8)            # first layer of the forward = LAST gradient of the backward (the image trunks' role)
    head = torch.nn.Sequential(torch.nn.Linear(8, 6), torch.nn.Tanh(), torch.nn.Linear(6, 3))
    params = list(late.parameters()) + list(head.parameters())
    log = []

    class Logged(mdist.GradBucket):
        def _issue(self, flat, asynchronous=False):
            log.append(("issue", flat.numel(), asynchronous))
            super()._issue(flat, asynchronous)

    late.weight.register_post_accumulate_grad_hook(lambda p: log.append(("late gradient ready",)))
    x = torch.randn(5, 8, generator=torch.Generator().manual_seed(10 + rank))
    results = {}
    for mode in ("one bucket", "early bucket"):
        for p in params:
            p.grad = None
        del log[:]
        bucket = Logged(params, early=list(head.parameters()) if mode == "early bucket" else None)
        loss = head(torch.relu(late(x))).pow(2).mean()
        bucket.arm()
        loss.backward()
        bucket.sync()
        results[mode] = ([p.grad.clone() for p in params], list(log), list(bucket.issue_log))
    n_head, n_late = sum(p.numel() for p in head.parameters()), sum(p.numel() for p in late.parameters())
    # one bucket: a single collective after the backward
    assert results["one bucket"][1] == [("late gradient ready",), ("issue", n_head + n_late, False)]
    assert results["one bucket"][2] == [("all", n_head + n_late)]
    # early bucket: its collective is issued (asynchronously) BEFORE the late gradient exists, the rest goes after the backward
    assert results["early bucket"][1] == [("issue", n_head, True), ("late gradient ready",), ("issue", n_late, False)]
    assert results["early bucket"][2] == [("early", n_head), ("rest", n_late)]
    same = all(torch.equal(a, b) for a, b in zip(results["one bucket"][0], results["early bucket"][0]))
    # and both are the mean over the ranks of the per-rank gradients
    ref = []
    for r in range(world):
        xr = torch.randn(5, 8, generator=torch.Generator().manual_seed(10 + r))
        for p in params:
            p.grad = None
        head(torch.relu(late(xr))).pow(2).mean().backward()
        ref.append([p.grad.clone() for p in params])
    mean = [sum(g) / world for g in zip(*ref)]
    # An early list that names a module the forward never uses (transform_y, latent heads, FCL-only paths of a ResNetNP flavour):
    # step 1 cannot count down - everything goes out in one late collective and the list is re-planned; step 2 fires early.
    unused = torch.nn.Linear(3, 3)
    bucket = Logged(params + list(unused.parameters()), early=list(head.parameters()) + list(unused.parameters()))
    pruned = []
    for step in range(2):
        for p in params:
            p.grad = None
        del log[:]
        loss = head(torch.relu(late(x))).pow(2).mean()
        bucket.arm()
        loss.backward()
        bucket.sync()
        pruned.append((list(bucket.issue_log), max(float((p.grad - m).abs().max()) for p, m in zip(params, mean))))
    assert pruned[0][0] == [("all", n_head + n_late)] and pruned[1][0] == [("early", n_head), ("rest", n_late)], pruned
    assert len(bucket.early) == len(list(head.parameters())) and max(e for _, e in pruned) <= 1e-7
    # sync(wait=False) without defer_scale is refused BEFORE anything is issued (no rank is left inside a collective)
    del log[:]
    try:
        bucket.sync(wait=False)
        raise AssertionError("sync(wait=False, defer_scale=False) must raise")
    except ValueError:
        assert log == []
    out[rank] = (same, max(float((a - b).abs().max()) for a, b in zip(results["early bucket"][0], mean)))
    dist.barrier()
    dist.destroy_process_group()


def test_early_bucket_is_issued_inside_backward_and_equals_the_single_bucket():
    """GradBucket(early=...): DESIGN.md §6 says the gradients that are complete early in the backward are all-reduced under the
    rest of it - here the issue order is recorded on two gloo ranks (early collective, THEN the late layer's gradient becomes
    ready, THEN the rest), and the result is bit-identical to the single-bucket path and equal to the mean over the ranks."""
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_early_worker, args=(r, 2, port, out)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            assert p.exitcode == 0
        assert len(out) == 2 and all(v[0] for v in out.values()) and max(v[1] for v in out.values()) <= 1e-7, dict(out)


def _stab_worker(rank, world, port, out):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [os.path.join(root, "what-matters-for-meta-learning_amd"), root]
    from mlhot import dist as mdist
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    mdist.init_from_env("gloo")
    ex = mdist.StabiliserExchange()
    res = []
    # (rank-local key maxima, rank-local stabiliser-gradient sums) -> every rank: the batch maximum, ONE owner, the batch sum
    for maxima, sums in (([1.5, 3.0], [2.0, -0.5]), ([3.0, 3.0], [0.25, 0.5]), ([-2.0, -7.0], [1.0, 1.0])):
        x = torch.tensor([maxima[rank], 7.0, sums[rank], 0.0])        # x[1] holds garbage on entry
        ex.forward(x)
        ex.backward(x)
        res.append(x.tolist())
    out[rank] = (res, list(ex.calls))
    dist.barrier()
    dist.destroy_process_group()


def _two_part_worker(rank, world, port, out):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [os.path.join(root, "what-matters-for-meta-learning_amd"), root]
    from mlhot import dist as mdist
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    mdist.init_from_env("gloo")
    torch.manual_seed(0)

    class Cut(torch.nn.Module):                         # the ResNet family's shape: a "trunk" whose output the forward detaches
        def __init__(self):
            super().__init__()
            self.trunk = torch.nn.Linear(8, 8)
            self.head = torch.nn.Sequential(torch.nn.Linear(8, 6), torch.nn.Tanh(), torch.nn.Linear(6, 3))

        def forward(self, x, cut):
            f = torch.relu(self.trunk(x))
            if cut:
                leaf = f.detach().requires_grad_(True)
                self.__dict__["_cut_pairs"] = [(f, leaf)]
                f = leaf
            return self.head(f).pow(2).mean()

    m = Cut()
    params = list(m.parameters())
    x = torch.randn(5, 8, generator=torch.Generator().manual_seed(10 + rank))
    grads = {}
    for cut in (False, True):
        for p in params:
            p.grad = None
        bucket = mdist.GradBucket(params, early=list(m.head.parameters()))
        loss = m(x, cut)
        if cut:
            seen = {}

            def between():
                seen["trunk_has_no_gradient_yet"] = m.trunk.weight.grad is None
                bucket.issue_early()                     # what bench.py does between its two graph replays
            mdist.backward_in_two(loss, m, between=between)
        else:
            loss.backward()
        bucket.sync()
        grads[cut] = ([p.grad.clone() for p in params], [k for k, _ in bucket.issue_log])
    assert grads[False][1] == ["all"] and grads[True][1] == ["early", "rest"] and seen["trunk_has_no_gradient_yet"]
    out[rank] = all(torch.equal(a, b) for a, b in zip(grads[False][0], grads[True][0]))
    dist.barrier()
    dist.destroy_process_group()


def test_backward_in_two_with_the_early_bucket_issued_in_between():
    """mlhot.dist.backward_in_two + GradBucket.issue_early() on two gloo ranks: the early bucket leaves between the two parts of
    the backward (no autograd hook involved - the form a step replayed as two hipGraphs uses), the rest behind the second part,
    and the averaged gradients are bit-identical to one backward + one collective."""
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_two_part_worker, args=(r, 2, port, out)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            assert p.exitcode == 0
        assert out[0] and out[1]


def test_stabiliser_exchange_on_two_ranks():
    """mlhot.dist.StabiliserExchange (the collective between the staged halves of the attention passes, include/mlhot.h "strict
    sharded parity"): the batch maximum reaches every rank, exactly one rank owns the arg-max (the lowest on ties), the
    stabiliser's gradient is summed."""
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_stab_worker, args=(r, 2, port, out)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            assert p.exitcode == 0
        (r0, c0), (r1, c1) = out[0], out[1]
    assert c0 == c1 == ["fwd", "bwd"] * 3
    assert r0 == [[3.0, 0.0, 1.5, 0.0], [3.0, 1.0, 0.75, 0.0], [-2.0, 1.0, 2.0, 0.0]]
    assert r1 == [[3.0, 1.0, 1.5, 0.0], [3.0, 0.0, 0.75, 0.0], [-2.0, 0.0, 2.0, 0.0]]


def test_stabiliser_exchange_without_a_process_group_owns_the_maximum():
    from mlhot import dist as mdist
    x = torch.tensor([0.5, 0.0, -3.0, 0.0])
    ex = mdist.StabiliserExchange()
    ex.forward(x)
    ex.backward(x)
    assert x.tolist() == [0.5, 1.0, -3.0, 0.0]
