#!/usr/bin/env python3
"""Headline benchmark: meta-tasks/s, forward+backward, ANPShapeNet1D 15+15-shot, 16 tasks per GPU.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = zero_grad + forward + loss + backward (+ one flat gradient all-reduce when N>1) of the
hand-written HIP path on a synthetic meta-batch that is already resident in HBM.  Rank 0 prints ONE
JSON line; it also carries `roofline` (the dominant kernel, timed live with HIP events on its launch
stream) and `cpu_baseline` (the CPU oracle - a port of the reference arithmetic - on the host cores).
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "what-matters-for-meta-learning_amd"), ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

WORKLOADS = {   # BASELINE.json configs[1] / configs[2]
    "c3": dict(method="ANPShapeNet1D", agg_mode="attention", dim_r=64,
               name="ANPShapeNet1D 128x128x1 15-shot context + 15 target, 16 tasks/GPU (BASELINE configs[2])"),
    "c2": dict(method="CNPShapeNet1D", agg_mode="mean", dim_r=100,
               name="CNPShapeNet1D mean-agg 128x128x1 15+15-shot, 16 tasks/GPU (BASELINE configs[1])"),
}
T_LOCAL, NC, NQ = 16, 15, 15
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 = fp32 vector rate


def alg_flops(label, n_img):
    """Algorithmic FLOPs of ONE launch carrying `label` (2 x MACs of the layer)."""
    pos2 = n_img * 1024
    conv1 = 2.0 * n_img * 4096 * 32 * 9
    table = {
        # conv1-fused kernels: conv2's work plus conv1's own (the halo recompute is not algorithmic work)
        "enc.conv12": 2.0 * pos2 * 48 * 288 + conv1,
        "enc.bwd.conv12.wgrad": 2.0 * pos2 * 48 * 288,
        "enc.bwd.conv12.dgrad": 2.0 * pos2 * 32 * 48 * 9 + conv1,   # conv2 data gradient + conv1 weight gradient
        "enc.conv2": 2.0 * pos2 * 48 * 288,
        "enc.bwd.conv2.wgrad": 2.0 * pos2 * 48 * 288,
        "enc.bwd.conv2.dgrad": 2.0 * pos2 * 32 * 48 * 9,          # all 4 parity classes in one launch
        "enc.conv3": 2.0 * n_img * 64 * 64 * 432,
        "enc.bwd.conv3.wgrad": 2.0 * n_img * 64 * 64 * 432,
        "enc.bwd.conv3.dgrad": 2.0 * n_img * 64 * 48 * 64 * 9,          # all 4 parity classes in one launch
        "enc.conv1": 2.0 * n_img * 4096 * 32 * 9,
        "enc.bwd.conv1.wgrad": 2.0 * n_img * 4096 * 32 * 9,
        "enc.linear": 2.0 * n_img * 4096 * 64,
        "enc.bwd.linear": 2.0 * 2.0 * n_img * 4096 * 64,           # data + weight gradient in one launch
        "enc.bwd.linear.dgrad": 2.0 * n_img * 4096 * 64,
        "enc.bwd.linear.wgrad": 2.0 * n_img * 4096 * 64,
    }
    return table.get(label)


def make_cfg(w, device):
    return types.SimpleNamespace(device=device, seed=2578, img_size=[128, 128, 1], tasks_per_batch=T_LOCAL, input_dim=3,
                                 output_dim=2, agg_mode=w["agg_mode"], img_agg="", dim_w=64, n_hidden_units_r=[100, 100],
                                 dim_r=w["dim_r"], dim_z=64, task="shapenet_1d", method=w["method"])


def cpu_baseline(w, steps=3):
    """The CPU oracle (torch-CPU restatement of the reference forward, autograd backward) on the
    same 16-task batch, all host cores; a bounded sample: 1 warm-up + `steps` timed steps."""
    import importlib
    from oracle import ref_cpu as O
    from mlhot import synth
    cfg = make_cfg(w, torch.device("cpu"))
    model = getattr(importlib.import_module("networks." + w["method"]), w["method"])(cfg)
    p = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "projection" not in k)
         for k, v in model.state_dict().items()}
    cx, qx, cy, qy = synth.get_batch("shapenet_1d", T_LOCAL, NC, NQ, seed=1234)
    times = []
    for i in range(steps + 1):
        for t in p.values():
            t.grad = None
        t0 = time.perf_counter()
        mu = O.vanilla_np_forward(p, cx, cy, qx, w["agg_mode"], tanh=True)
        O.calc_loss("shapenet_1d", mu, qy).backward()
        dt = time.perf_counter() - t0
        if i:
            times.append(dt)
    med = sorted(times)[len(times) // 2]
    return {"value": T_LOCAL / med, "unit": "meta-tasks/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} fwd+bwd steps of the same 16-task 15+15 batch after 1 warm-up, median {med * 1e3:.1f} ms/step"}


def _time_graph(fn, iters):
    """Capture fn() once into a hipGraph, replay it `iters` times, return ms per replay."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):      # the warm-up's stream: the parameters' grad-accumulation nodes live there
        fn()
    for _ in range(5):
        graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        graph.replay()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / iters


def _capture(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):      # the warm-up's stream: the parameters' grad-accumulation nodes live there
        fn()
    return graph


def measure_host_fed(w, device, loss_fn, model, iters):
    """PCIe-inclusive view of the same step when every batch starts in HOST memory (the reference's loaders hand over host
    batches, trainer/model_trainer.py:63-70); never the headline `value`.
      reference route: fp32 channel-first host tensors (converted on the host), `.copy_` from pageable memory, then the step;
      ingest route   : uint8 channel-last host arrays -> pinned staging -> H2D on a copy stream while the previous step
                       computes -> mlhot_ingest_u8_nhwc -> the step (mlhot.ingest.BatchIngest)."""
    from mlhot import synth
    from mlhot.ingest import BatchIngest
    hb = synth.get_batch_u8("shapenet_1d", T_LOCAL, NC, NQ, seed=1234)
    ing = BatchIngest(device)
    ing.stage(*hb)
    cx, qx, cy, qy = ing.take()                                    # the fixed device tensors every later take() refills

    def step():
        model.zero_grad(set_to_none=True)
        loss_fn.calc_loss(model(cx, cy, qx)[0], None, qy).backward()

    graph = _capture(step)
    out = {}
    # ingest route, steady state
    ing.stage(*hb)
    for timed in (False, True):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            ing.take()
            graph.replay()
            ing.stage(*hb)                                          # host memcpy into pinned staging + async H2D of the next batch
        torch.cuda.synchronize()
        if timed:
            out["ingest_route_ms_per_step"] = 1e3 * (time.perf_counter() - t0) / iters
    ing.take()
    # the ingest kernel alone (context + target images are one packed run: one launch)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ing.stage(*hb)
    torch.cuda.synchronize()
    import mlhot
    L = mlhot.lib()
    src_u8, dst_f32 = ing.device_views()
    ev[0].record()
    for _ in range(iters):
        L.ingest_u8_nhwc(src_u8, out=dst_f32)
    ev[1].record()
    torch.cuda.synchronize()
    ing.take()
    ingest_ms = ev[0].elapsed_time(ev[1]) / iters
    nbytes = 5 * (hb[0].size + hb[1].size)                          # 1 byte read + 4 written per pixel
    out["ingest_kernel_us_per_batch"] = 1e3 * ingest_ms
    out["ingest_kernel_GBps"] = nbytes / (ingest_ms * 1e-3) / 1e9
    # reference route: host conversion done (not timed), pageable fp32 tensors copied synchronously, then the step
    host = [synth.host_convert(hb[0]), synth.host_convert(hb[1]), hb[2], hb[3]]
    for timed in (False, True):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(max(3, iters // 4)):
            for d_, h in zip((cx, qx, cy, qy), host):
                d_.copy_(h)
            graph.replay()
        torch.cuda.synchronize()
        if timed:
            out["reference_route_ms_per_step"] = 1e3 * (time.perf_counter() - t0) / max(3, iters // 4)
    out["host_bytes_per_batch"] = {"ingest_route": int(hb[0].size + hb[1].size + 4 * (hb[2].numel() + hb[3].numel())),
                                   "reference_route": int(sum(4 * t.numel() for t in host))}
    out["tasks_per_s_host_fed"] = {"ingest_route": 1e3 * T_LOCAL / out["ingest_route_ms_per_step"],
                                   "reference_route": 1e3 * T_LOCAL / out["reference_route_ms_per_step"]}
    return out


def measure_variable_nc(w, device, loss_fn, model, batch, iters):
    """The reference's TRAINING batches draw the context size per iteration (dataset/shapenet_1d.py:120: 3..shot); the
    target count stays `shot`.  One captured hipGraph per context size, replayed in a seeded random order."""
    import numpy as np
    cx, qx, cy, qy = batch
    graphs, ins = {}, {}
    for nc in range(3, NC + 1):
        ins[nc] = (cx[:, :nc].contiguous(), cy[:, :nc].contiguous())

        def step(nc=nc):
            model.zero_grad(set_to_none=True)
            loss_fn.calc_loss(model(ins[nc][0], ins[nc][1], qx)[0], None, qy).backward()

        graphs[nc] = _capture(step)
    order = np.random.RandomState(0).randint(3, NC + 1, size=4 * iters)
    for nc in order[:8]:
        graphs[int(nc)].replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for nc in order:
        graphs[int(nc)].replay()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / len(order)
    return {"ms_per_step": ms, "tasks_per_s": 1e3 * T_LOCAL / ms, "mean_context": float(order.mean()), "steps": int(len(order)),
            "note": "context size ~ U{3..15} per step (the reference's training draw), 15 targets, one hipGraph per size"}


def measure_train_loop(w, device, loss_fn, iters):
    """A whole TRAINING iteration from host batch to updated weights, two ways (informational; `value` is the resident fwd+bwd step):
      reference-style loop: fp32 host batch `.to(device)`, eager zero_grad / forward / loss / backward on the HIP kernels,
                            torch.optim.Adam over the ~70 parameter tensors, loss.item() every iteration (trainer/model_trainer.py:59-93);
      replayed loop       : uint8 batch through mlhot.ingest.BatchIngest, ONE hipGraph holding forward, loss, backward and the
                            capturable flat Adam step (mlhot.optim.FlatAdam, device-side step count), loss fetched every 50 iterations
                            (trainer.ModelTrainer with config.graph_steps)."""
    import importlib
    from mlhot import synth
    from mlhot.ingest import BatchIngest
    from mlhot.optim import FlatAdam
    cls = getattr(importlib.import_module("networks." + w["method"]), w["method"])
    hb = synth.get_batch_u8("shapenet_1d", T_LOCAL, NC, NQ, seed=1234)
    host = [synth.host_convert(hb[0]), synth.host_convert(hb[1]), hb[2], hb[3]]
    out = {}
    # reference-style loop
    model = cls(make_cfg(w, device)).to(device)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    n_ref = max(5, iters // 2)
    for timed in (False, True):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_ref):
            cx, qx, cy, qy = (t.to(device) for t in host)
            opt.zero_grad()
            loss = loss_fn.calc_loss(model(cx, cy, qx)[0], None, qy)
            loss.backward()
            opt.step()
            loss.item()
        torch.cuda.synchronize()
        if timed:
            out["reference_style_ms_per_iter"] = 1e3 * (time.perf_counter() - t0) / n_ref
    # replayed loop
    model = cls(make_cfg(w, device)).to(device)
    opt = FlatAdam(model, lr=1e-4, ctx_num=NC, test_num=NQ, capturable=True)
    ing = BatchIngest(device)
    ing.stage(*hb)
    cx, qx, cy, qy = ing.take()

    def it():
        opt.zero_grad()
        loss = loss_fn.calc_loss(model(cx, cy, qx)[0], None, qy)
        loss.backward()
        opt.step()
        return loss.detach()

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            it()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        static_loss = it()
    ing.stage(*hb)
    n_rep = 4 * iters
    for timed in (False, True):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n_rep):
            ing.take()
            graph.replay()
            ing.stage(*hb)
            if i % 50 == 49:
                static_loss.item()
        torch.cuda.synchronize()
        if timed:
            out["replayed_ms_per_iter"] = 1e3 * (time.perf_counter() - t0) / n_rep
    ing.take()
    out["tasks_per_s"] = {"reference_style": 1e3 * T_LOCAL / out["reference_style_ms_per_iter"],
                          "replayed": 1e3 * T_LOCAL / out["replayed_ms_per_iter"]}
    out["adam_steps_taken"] = int(opt.step_dev.item())
    return out


def measure_extras(w, device, loss_fn, batch, iters):
    """Not the headline metric: (a) the training forward alone (activations saved, no backward), (b) the full step
    followed by the fused flat Adam update (mlhot.optim.FlatAdam: one launch over the flat parameter buffer).  The Adam
    leg is timing only: inside a replayed graph the bias-correction step count is frozen at capture time."""
    import importlib
    from mlhot.optim import FlatAdam
    cx, qx, cy, qy = batch
    try:
        model = getattr(importlib.import_module("networks." + w["method"]), w["method"])(make_cfg(w, device)).to(device)
        opt = FlatAdam(model, lr=1e-4, ctx_num=NC, test_num=NQ)

        def fwd():
            return loss_fn.calc_loss(model(cx, cy, qx)[0], None, qy)

        def step_adam():
            model.zero_grad(set_to_none=True)
            loss_fn.calc_loss(model(cx, cy, qx)[0], None, qy).backward()
            opt.step()

        out = {"fwd_only_ms": _time_graph(fwd, iters), "step_with_flat_adam_ms": _time_graph(step_adam, iters),
               "note": "hipGraph replays; informational, not the headline metric"}
        # PCIe-inclusive view (the reference hands over HOST batches, model_trainer.py:63-70): pinned host -> device copy of
        # one batch's images + labels, NOT overlapped with compute; `value` above never includes it
        host = [t.detach().cpu().pin_memory() for t in (cx, qx, cy, qy)]
        dst = [torch.empty_like(t) for t in (cx, qx, cy, qy)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            for h, d_ in zip(host, dst):
                d_.copy_(h, non_blocking=True)
        torch.cuda.synchronize()
        out["h2d_ms_per_batch"] = 1e3 * (time.perf_counter() - t0) / iters
        out["h2d_mbytes_per_batch"] = sum(t.numel() * 4 for t in host) / 1e6
        del opt                                                      # FlatAdam re-pointed the parameters; build a fresh model
        model = getattr(importlib.import_module("networks." + w["method"]), w["method"])(make_cfg(w, device)).to(device)
        out["host_fed"] = measure_host_fed(w, device, loss_fn, model, iters)
        out["variable_context"] = measure_variable_nc(w, device, loss_fn, model, batch, iters)
        out["train_loop"] = measure_train_loop(w, device, loss_fn, iters)
        return out
    except Exception as e:  # noqa: BLE001 - extras must never break the bench line
        return {"error": f"{type(e).__name__}: {e}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prof-steps", type=int, default=5)
    ap.add_argument("--no-graph", action="store_true", help="run the step eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--no-extras", action="store_true", help="skip the fwd-only / +Adam timing legs")
    ap.add_argument("--dbg", type=int, default=0, help="kernel timing experiments (results become WRONG; never a bench line)")
    args = ap.parse_args()
    w = WORKLOADS[args.workload]

    import importlib
    import torch.distributed as dist
    import mlhot
    from mlhot import dist as mdist, synth
    from trainer.losses import LossFunc
    # MLHOT_DIST_BACKEND / MLHOT_ONE_DEVICE: test hooks (e.g. two gloo ranks sharing the only GPU of a 1-GPU box, to
    # exercise the N>1 control flow); the driver's launch uses neither (nccl = RCCL, one rank per GPU)
    rank, local, world = mdist.init_from_env(os.environ.get("MLHOT_DIST_BACKEND"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if os.environ.get("MLHOT_ONE_DEVICE"):
        local = 0
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    mlhot.build_product()
    if args.dbg:
        mlhot.lib().set_option("dbg", args.dbg)

    model = getattr(importlib.import_module("networks." + w["method"]), w["method"])(make_cfg(w, device)).to(device)
    loss_fn = LossFunc("mse", "shapenet_1d")
    cx, qx, cy, qy = synth.get_batch("shapenet_1d", T_LOCAL, NC, NQ, seed=1234 + rank, device=device)
    bucket = mdist.GradBucket(model.parameters())

    def step():
        model.zero_grad(set_to_none=True)
        mu, var, kl = model(cx, cy, qx)
        loss = loss_fn.calc_loss(mu, var, qy)
        loss.backward()
        bucket.sync()
        return loss

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # The step has static shapes and no host sync, so it is captured once into a hipGraph and replayed
    # (the RCCL all-reduce stays outside the graph).  --no-graph runs it eagerly.
    run = step
    graphed = False
    if not args.no_graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    model.zero_grad(set_to_none=True)
                    loss_fn.calc_loss(model(cx, cy, qx)[0], None, qy).backward()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            model.zero_grad(set_to_none=True)
            with torch.cuda.graph(graph, stream=side):      # the warm-up's stream: the parameters' grad-accumulation nodes live there
                static_loss = loss_fn.calc_loss(model(cx, cy, qx)[0], None, qy)
                static_loss.backward()

            def run():
                graph.replay()
                bucket.sync()
                return static_loss
            graphed = True
        except Exception as e:  # noqa: BLE001 - fall back to eager, say so
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            run = step

    for _ in range(args.warmup):
        run()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = run()
    t_enqueue = time.perf_counter() - t0
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    final_loss = loss.item()

    # ---- roofline leg: per-launch HIP-event timing of every kernel over a few more steps ----------
    roof, kernels = None, None
    if args.prof_steps > 0:
        L = mlhot.lib()
        L.prof_begin(8192)
        for _ in range(args.prof_steps):
            step()
        torch.cuda.synchronize()
        recs = L.prof_end()
        agg = {}
        for label, ms in recs:
            a = agg.setdefault(label, [0, 0.0])
            a[0] += 1
            a[1] += ms
        n_img = T_LOCAL * (NC + NQ)
        kernels = {k: {"launches_per_step": v[0] / args.prof_steps, "avg_us": 1e3 * v[1] / v[0],
                       "us_per_step": 1e3 * v[1] / args.prof_steps} for k, v in agg.items()}
        dom = max((k for k in agg if alg_flops(k, n_img)), key=lambda k: agg[k][1])
        avg_s = agg[dom][1] / agg[dom][0] * 1e-3
        ach = alg_flops(dom, n_img) / avg_s / 1e12
        traffic = None   # HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/)
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                traffic = json.load(f).get(dom, {}).get("hbm_bytes")
        except OSError:
            pass
        roof = {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic, "avg_launch_us": avg_s * 1e6,
                "alg_flops_per_launch": alg_flops(dom, n_img)}
    # ---- informational extras (SURVEY §8d "also report fwd-only and +Adam"), rank 0, single-GPU runs only -----------
    extras = None
    if world == 1 and not args.no_extras and not args.no_graph:
        extras = measure_extras(w, device, loss_fn, (cx, qx, cy, qy), max(10, args.steps // 2))
    if world > 1:
        dist.barrier()

    if rank == 0:
        out = {"metric": "meta-tasks/sec (fwd+bwd), ANP ShapeNet1D 15+15-shot 16-task batch",
               "value": world * T_LOCAL * args.steps / elapsed, "unit": "meta-tasks/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": w["name"], "tasks_per_gpu": T_LOCAL, "global_tasks": world * T_LOCAL,
                          "context_shots": NC, "target_shots": NQ, "image": "128x128x1",
                          "parallelism": f"task-sharded x{world}, one flat grad all-reduce" if world > 1 else "single GPU"},
               "final_loss": final_loss, "hipgraph": graphed, "host_enqueue_ms_per_step": 1e3 * t_enqueue / args.steps,
               "roofline": roof}
        if extras:
            out["extras"] = extras
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(w)
        else:
            out["cpu_baseline"] = None
        if kernels:
            top = sorted(kernels.items(), key=lambda kv: -kv[1]["us_per_step"])
            out["kernel_us_per_step"] = {k: round(v["us_per_step"], 1) for k, v in top[:12]}
            out["gpu_busy_us_per_step"] = round(sum(v["us_per_step"] for v in kernels.values()), 1)
            out["launches_per_step"] = round(sum(v["launches_per_step"] for v in kernels.values()), 1)
            if os.environ.get("MLHOT_BENCH_KERNELS"):
                with open(os.environ["MLHOT_BENCH_KERNELS"], "w") as f:
                    json.dump({k: v for k, v in top}, f, indent=1)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
