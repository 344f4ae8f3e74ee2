#!/usr/bin/env python3
"""Headline benchmark: meta-tasks/s, forward+backward, ANPShapeNet1D 15+15-shot, 16 tasks per GPU.

    python bench.py --gpus N --steps K --warmup W
    N>1 works both ways: under a launcher (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...:
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment), or as the plain command above - with no WORLD_SIZE in the
    environment this process touches no GPU, starts N rank processes of itself (one per GPU, rendezvous on 127.0.0.1), relays rank
    0's JSON line and exits with the worst rank's code (`spawn_ranks`).

A step = zero_grad + forward + loss + backward (+ one flat gradient all-reduce when N>1) of the
hand-written HIP path on a synthetic meta-batch that is already resident in HBM.  Rank 0 prints ONE
JSON line; it also carries `roofline` (the dominant kernel, timed live with HIP events on its launch
stream; `roofline.forward`: the training forward against the MFMA and the HBM roof) and `cpu_baseline`
(the CPU oracle - a port of the reference arithmetic - on the host's physical cores, plus a 1-thread figure).

    --workload c3 (default, BASELINE configs[2] = the headline metric) | c2 (configs[1]) | c5 (the per-GPU share of configs[4]:
    ANPMRShapeNet3D, Bayes-by-backprop ResNet encoder; eps drawn on the CPU generator by host threads, one step ahead)

The step is replayed as a hipGraph; with one rank and a vanilla workload --steps-per-graph (default 5) consecutive steps share a
graph (the 8.8 us between two graph launches is paid once per five steps; K and W must be multiples, the timed region stays EXACTLY
K steps).  `value` is exact fp32 arithmetic; `extras.split_precision_conv2` reports the same step with conv2 of the vanilla encoder
on the bf16 matrix pipe over 3-piece splits of the fp32 operands (opt-in, csrc/conv_split.h) next to its own roof - never `value`.
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

WORKLOADS = {   # BASELINE.json configs[1] / configs[2] (the headline metric) and the per-GPU share of configs[4]
    "c3": dict(kind="vanilla", method="ANPShapeNet1D", agg_mode="attention", dim_r=64, T=16, image="128x128x1", task="shapenet_1d",
               cfg="bench_c3_ANP_ShapeNet1D.yaml", fwd_gflop_per_task=1.062, name="ANPShapeNet1D 128x128x1 15-shot context + 15 target, 16 tasks/GPU (BASELINE configs[2])"),
    "c2": dict(kind="vanilla", method="CNPShapeNet1D", agg_mode="mean", dim_r=100, T=16, image="128x128x1", task="shapenet_1d",
               cfg="bench_c2_CNP_ShapeNet1D.yaml", fwd_gflop_per_task=1.044, name="CNPShapeNet1D mean-agg 128x128x1 15+15-shot, 16 tasks/GPU (BASELINE configs[1])"),
    "c5": dict(kind="resnet3d", method="ANPMRShapeNet3D", agg_mode="attention", T=8, image="64x64x3", task="shapenet_3d", beta=1e-7,
               cfg="bench_c5_ANPMR_ShapeNet3D.yaml", fwd_gflop_per_task=3.92,
               name="ANPMRShapeNet3D (Bayes-by-backprop ResNet encoder) 64x64x3 15+15-shot + task augmentation of the labels, 8 tasks/GPU "
                    "(the per-GPU share of BASELINE configs[4]: 64 tasks over 8 GPUs)"),
    # SURVEY §8f rank 4 (not a BASELINE config; kernel-time evidence for the 128 x 128 x 1 trunk geometries): cfg/train/ANP_Distractor.yaml
    "distractor": dict(kind="resnet_dis", method="ANPDistractor", agg_mode="attention", T=20, image="128x128x1", task="distractor",
                       fwd_gflop_per_task=11.1,
                       name="ANPDistractor (ResNet encoder + decoder, 1x1 skips) 128x128x1 15+15-shot, 20 tasks/GPU (cfg/train/ANP_Distractor.yaml)"),
}
METRIC = {   # BASELINE.json's headline metric string is c3's; every other workload says what it ran
    "c3": "meta-tasks/sec (fwd+bwd), ANP ShapeNet1D 15+15-shot 16-task batch",
    "c2": "meta-tasks/sec (fwd+bwd), CNP (mean-agg) ShapeNet1D 15+15-shot 16-task batch",
    "c5": "meta-tasks/sec (fwd+bwd), ANPMR ShapeNet3D 15+15-shot, 8 tasks per GPU",
    "distractor": "meta-tasks/sec (fwd+bwd), ANP Distractor 15+15-shot, 20 tasks per GPU",
}
# hipGraph capture mode: "global" (torch's default) makes ANY thread's event query / allocation during a capture an error - and
# with a process group alive ProcessGroupNCCL's watchdog thread polls its work events all the time (seen on the GPU box: the
# bench aborting inside the capture, rc -6, once in two runs at a forced world of one).  "thread_local" confines the check to the
# capturing thread, which is the one that matters here.  mlhot.graphs.capture also pauses Python's cyclic collector for the capture.
CAPTURE_MODE = "thread_local"


def capture_graph(graph, stream, pool=None):
    from mlhot.graphs import capture
    return capture(graph, stream, pool=pool)


NC, NQ = 15, 15
T_LOCAL = 16                    # vanilla workloads; WORKLOADS[...]["T"] is authoritative
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 = fp32 vector rate
PEAK_HBM_BYTES_S = 8.0e12       # MI355X_MICROARCH.md: HBM3E spec


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
        "enc.bwd.conv3": 2.0 * n_img * 64 * 64 * 432 + 2.0 * n_img * 64 * 48 * 64 * 9,      # weight + data gradient in one launch (csrc/conv3_tc.h conv3_bwd_kernel)
        "enc.conv1": 2.0 * n_img * 4096 * 32 * 9,
        "enc.bwd.conv1.wgrad": 2.0 * n_img * 4096 * 32 * 9,
        "enc.linear": 2.0 * n_img * 4096 * 64,
        "enc.bwd.linear": 2.0 * 2.0 * n_img * 4096 * 64,           # data + weight gradient in one launch
        "enc.bwd.linear.dgrad": 2.0 * n_img * 4096 * 64,
        "enc.bwd.linear.wgrad": 2.0 * n_img * 4096 * 64,
    }
    return table.get(label)


def trunk_flops_per_step(T, kind="resnet3d"):
    """Algorithmic FLOPs (2 x MACs) one step spends under every ResNet-trunk launch label.  c5: two Bayes-by-backprop encoder
    passes (context, target: 3x3 skip) and the decoder pass (1x1 skip) over 15 3x64x64 images per task each; distractor: three
    1x1-skip passes over 1x128x128 images.  L = output map sizes."""
    n, C, L = (15 * T, 3, [32, 16, 8, 4, 2]) if kind == "resnet3d" else (15 * T, 1, [64, 32, 16, 8, 4])
    f = {"trunk.stem": 0.0, "trunk.bwd.stem.wgrad": 0.0, "trunk.bwd.skip1.wgrad": 0.0}
    for skip_k in ((3, 3, 1) if kind == "resnet3d" else (1, 1, 1)):      # the three passes
        stem = 2.0 * n * L[0] ** 2 * 64 * 25 * C
        f["trunk.stem"] += stem
        f["trunk.bwd.stem.wgrad"] += stem
        for b in range(1, 5):                              # labels carry the block: one label = one kernel geometry
            c33 = 2.0 * n * L[b] ** 2 * 64 * 576           # one 3x3 64 -> 64 convolution on the block's output grid
            c11 = 2.0 * n * L[b] ** 2 * 64 * 64

            def add(name, v):
                f[f"{name}.b{b}"] = f.get(f"{name}.b{b}", 0.0) + v
            add("trunk.conv1", c33 + (c33 if skip_k == 3 else c11))
            add("trunk.conv2", c33)
            add("trunk.bwd.conv2.dgrad", c33)
            add("trunk.bwd.conv2.wgrad", c33)
            # blocks 1-2 (blocks 3-4 of a 64 x 64 trunk are one fused launch, no label of their own): the 1x1-skip passes' conv1^T + fused
            # skip under "dgrad"; the 3x3-skip passes' two sources (skip^T, conv1^T) in ONE launch under "dgrad2" (rw::dgrad2_dual_kernel)
            add("trunk.bwd.conv1.dgrad", (c33 + c11) if skip_k == 1 else 0.0)
            add("trunk.bwd.conv1.dgrad2", 2.0 * c33 if skip_k == 3 else 0.0)
            add("trunk.bwd.conv1.wgrad", c33 + (c33 if skip_k == 3 else 0.0))
            f["trunk.bwd.skip1.wgrad"] += c11 if skip_k == 1 else 0.0
    return f


def make_cfg(w, device):
    """The model's config.  The BASELINE workloads come from the cfg/ YAMLs the package ships (cfg/train/bench_c*.yaml: the reference's
    key surface, parsed by configs.config.Config as train.py would) with tasks_per_batch replaced by this rank's share."""
    if w.get("cfg"):
        import yaml
        from configs.config import Config
        with open(os.path.join(ROOT, "what-matters-for-meta-learning_amd", "cfg", "train", w["cfg"]), "rb") as f:
            raw = yaml.safe_load(f)
        raw["device"] = str(device)
        cfg = Config()
        cfg.set_init_values(raw, side_effects=False)          # no results/ directory, no log file
        assert cfg.method == w["method"] and cfg.task == w["task"] and (w["kind"] != "vanilla" or cfg.agg_mode == w["agg_mode"])
        cfg.tasks_per_batch = w["T"]
        return cfg
    return types.SimpleNamespace(device=device, seed=2578, img_size=[128, 128, 1], tasks_per_batch=w["T"], input_dim=2, output_dim=2,
                                 agg_mode="attention", img_agg="max", dim_w=16, task="distractor", temperature=0.07, method=w["method"])


def make_batch(w, seed, device="cpu"):
    from mlhot import synth
    if w["kind"] == "resnet3d":
        return synth.get_batch_3d(w["T"], NC, NQ, seed=seed, device=device, task_aug=True)
    if w["kind"] == "resnet_dis":                          # images in [0, 1), labels = object positions in [0, 1)^2 (dataset/shapenet_distractor.py)
        g = torch.Generator().manual_seed(seed)
        xs, xq = torch.rand(w["T"], NC, 1, 128, 128, generator=g), torch.rand(w["T"], NQ, 1, 128, 128, generator=g)
        ys, yq = torch.rand(w["T"], NC, 2, generator=g), torch.rand(w["T"], NQ, 2, generator=g)
        return tuple(t.to(device) for t in (xs, xq, ys, yq))
    return synth.get_batch("shapenet_1d", w["T"], NC, NQ, seed=seed, device=device)


def host_cpu():
    """(physical cores, model string) of the host the baseline runs on."""
    model, phys = "unknown", None
    try:
        cores = set()
        with open("/proc/cpuinfo") as f:
            pid = None
            for ln in f:
                if ln.startswith("model name") and model == "unknown":
                    model = ln.split(":", 1)[1].strip()
                elif ln.startswith("physical id"):
                    pid = ln.split(":", 1)[1].strip()
                elif ln.startswith("core id"):
                    cores.add((pid, ln.split(":", 1)[1].strip()))
        phys = len(cores) or None
    except OSError:
        pass
    if not phys:
        try:
            import psutil
            phys = psutil.cpu_count(logical=False)
        except Exception:  # noqa: BLE001
            phys = None
    return phys or os.cpu_count() or 1, model


def cpu_baseline(w, steps=5, warmups=2):
    """The CPU oracle (torch-CPU fp32 restatement of the reference forward, autograd backward; kind "port") on the same synthetic
    batch, at 1 / 8 / 32 / all physical cores: the SAME bounded sample at every thread count - median of `steps` (5) timed steps
    after `warmups` (2) untimed ones, ~30 s of CPU work in all (SURVEY §8d's median of 10 after 3 would be ~60 s; the default
    bench run has to finish within minutes).  `value` is the best thread count's figure, `cores` that count."""
    import importlib
    from oracle import ref_cpu as O
    cfg = make_cfg(w, torch.device("cpu"))
    model = getattr(importlib.import_module("networks." + w["method"]), w["method"])(cfg)
    p = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "projection" not in k)
         for k, v in model.state_dict().items()}
    cx, qx, cy, qy = make_batch(w, 1234)
    cores, cpu_model = host_cpu()
    prev = torch.get_num_threads()

    def one():
        for t in p.values():
            t.grad = None
        t0 = time.perf_counter()
        if w["kind"] == "resnet3d":
            torch.manual_seed(99)
            mu, kl = O.anpmr3d_forward(p, cx, cy, qx)
            (O.calc_loss("shapenet_3d", mu, qy) + w["beta"] * kl).backward()
        elif w["kind"] == "resnet_dis":
            O.calc_loss("distractor", O.resnet_np_forward(p, cx, cy, qx, "attention", "max"), qy).backward()
        else:
            mu = O.vanilla_np_forward(p, cx, cy, qx, w["agg_mode"], tanh=True)
            O.calc_loss("shapenet_1d", mu, qy).backward()
        return time.perf_counter() - t0

    # torch's CPU convolutions do not scale to 128 threads on a 16-task batch (measured: 1 thread beats 128 on the 2 x 64-core
    # host), so the baseline is taken at the BEST of a few thread counts, each a bounded sample
    by_threads = {}
    try:
        for nthr in sorted({1, 8, 32, cores} & set(range(1, cores + 1))):
            torch.set_num_threads(nthr)
            for _ in range(warmups):
                one()
            times = sorted(one() for _ in range(steps))
            by_threads[nthr] = times[len(times) // 2]
    finally:
        torch.set_num_threads(prev)
    best = min(by_threads, key=by_threads.get)
    return {"value": w["T"] / by_threads[best], "unit": "meta-tasks/s", "cores": best, "kind": "port", "cpu_model": cpu_model,
            "physical_cores": cores, "single_thread_value": w["T"] / by_threads[1],
            "tasks_per_s_by_threads": {str(k): round(w["T"] / v, 2) for k, v in by_threads.items()},
            "sample": f"fwd+bwd steps of the same {w['T']}-task 15+15 batch, torch threads in {sorted(by_threads)}: at EVERY count {warmups} "
                      f"warm-ups, then the median of {steps} timed steps; value = the best thread count ({best}): "
                      f"{by_threads[best] * 1e3:.0f} ms/step; all {cores} physical cores: {by_threads[cores] * 1e3:.0f} ms/step"}


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
    with capture_graph(graph, side):      # the warm-up's stream: the parameters' grad-accumulation nodes live there
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
    with capture_graph(graph, side):      # the warm-up's stream: the parameters' grad-accumulation nodes live there
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
    from mlhot.ops import loss_grad_in_backward
    hb = synth.get_batch_u8("shapenet_1d", T_LOCAL, NC, NQ, seed=1234)
    ing = BatchIngest(device)
    ing.stage(*hb)
    cx, qx, cy, qy = ing.take()                                    # the fixed device tensors every later take() refills

    def step():
        model.zero_grad(set_to_none=True)
        with loss_grad_in_backward():
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
    from mlhot.ops import loss_grad_in_backward
    cx, qx, cy, qy = batch
    graphs, ins = {}, {}
    for nc in range(3, NC + 1):
        ins[nc] = (cx[:, :nc].contiguous(), cy[:, :nc].contiguous())

        def step(nc=nc):
            model.zero_grad(set_to_none=True)
            with loss_grad_in_backward():
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


def measure_variable_nc_3d(w, device, loss_fn, model, eps, fwd_bwd_on, iters):
    """c5: the reference's ShapeNet3D TRAINING draw (dataset/shapenet_3d.py:110, 200-204): context size ~ U{1..15} per batch,
    the REST of the object's 30 views are the targets (Nq = 30 - Nc, up to 29).  One captured hipGraph per context size,
    replayed in a seeded random order with the eps staged per step like the headline loop."""
    import numpy as np
    from mlhot import synth
    cx, qx, cy, qy = synth.get_batch_3d(w["T"], 15, 15, seed=4321, device=device, task_aug=True)
    pool_x, pool_y = torch.cat([cx, qx], dim=1), torch.cat([cy, qy], dim=1)          # 30 views per task
    graphs, keep = {}, []
    for nc in range(1, 16):
        b = (pool_x[:, :nc].contiguous(), pool_x[:, nc:].contiguous(), pool_y[:, :nc].contiguous(), pool_y[:, nc:].contiguous())
        keep.append(b)                                   # a graph's inputs have to outlive its capture

        def step(b=b):
            eps.rewind()
            fwd_bwd_on(*b)
        eps.stage()
        with eps.active():
            graphs[nc] = _capture(step)
    order = np.random.RandomState(0).randint(1, 16, size=2 * iters)

    def run(seq):
        for nc in seq:
            eps.stage()
            if eps.source == "host":
                eps.prefetch()
            graphs[int(nc)].replay()
    run(order[:6])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(order)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / len(order)
    if eps._worker is not None:
        eps.stage()
    return {"ms_per_step": ms, "tasks_per_s": 1e3 * w["T"] / ms, "mean_context": float(order.mean()), "steps": int(len(order)),
            "note": "context size ~ U{1..15} per step, targets = the other 30 - Nc views (the reference's ShapeNet3D training draw), "
                    "one hipGraph per size"}


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
    from mlhot.ops import loss_grad_in_backward
    cls = getattr(importlib.import_module("networks." + w["method"]), w["method"])
    hb = synth.get_batch_u8("shapenet_1d", T_LOCAL, NC, NQ, seed=1234)
    host = [synth.host_convert(hb[0]), synth.host_convert(hb[1]), hb[2], hb[3]]
    out = {}
    # the reference's calling sequence, unchanged (train.py:52-90): torch.optim.Adam(model.parameters(), lr), a loader handing out fp32
    # host batches, ModelTrainer(model, loss, optimizer, config, data) and its per-iteration loop with loss.item() every iteration.
    # trainer.ModelTrainer continues that optimizer as the flat one-launch update, replays the iteration from a hipGraph and copies the
    # next host batch while the step computes (round 5; `promoted` below says what it did).
    import tempfile
    from trainer.model_trainer import ModelTrainer

    class HostLoader:                                   # the reference's data interface: get_batch(source, tasks_per_batch, shot) -> fp32 host tensors
        def get_batch(self, source, tasks_per_batch, shot):
            return tuple(host)

        def gen_bg(self, *a, **k):
            pass

    n_ref = max(20, 2 * iters)
    with tempfile.TemporaryDirectory() as tmp:
        cfg = make_cfg(w, device)
        cfg.iterations, cfg.val_freq, cfg.val_iters, cfg.bg_gen_freq, cfg.gen_bg = 4, 10 ** 9, 1, 10 ** 9, False
        cfg.save_path, cfg.logger, cfg.contrastive, cfg.max_ctx_num, cfg.beta = tmp, None, False, NC, 0
        cfg.close_after_train = False
        model = cls(cfg).to(device)
        tr = ModelTrainer(model=model, loss=loss_fn, optimizer=torch.optim.Adam(model.parameters(), lr=1e-4), config=cfg, data=HostLoader())
        tr.train()                                      # 4 iterations: eager warm-up of the batch shape, capture, two replays (+ the final checkpoint)
        tr.iterations = 10 ** 9                         # the timed region below is train()'s loop body, iteration by iteration
        for timed in (False, True):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for it in range(5, 5 + n_ref):
                tr._prefetch = 2      # what train() sets far from a validation round: two batches may be drawn ahead
                tr._train_iter(it)
            torch.cuda.synchronize()
            if timed:
                out["reference_style_ms_per_iter"] = 1e3 * (time.perf_counter() - t0) / n_ref
        feed = tr._host_prefetch.u8 if tr._host_prefetch is not None else None
        out["reference_style_promoted"] = {"optimizer": type(tr.optimizer).__name__, "graph_replay": bool(tr._graph_default),
                                           "host_batch_prefetch": tr._host_prefetch is not None, "batches_drawn_ahead": 2,
                                           "loss_logged_one_iteration_late": bool(tr._lagged()),
                                           "host_batches_as_bytes": ({"shipped": feed.shipped, "refused": feed.refused, "host_threads": feed.threads,
                                                                      "bytes_per_batch": int(host[0].numel() + host[1].numel() + 4 * (host[2].numel() + host[3].numel()))}
                                                                     if feed is not None else None)}
        # the same loop with the byte route off (config.host_u8 = False): the round-5 form, fp32 over PCIe behind the step
        cfg2 = make_cfg(w, device)
        cfg2.iterations, cfg2.val_freq, cfg2.val_iters, cfg2.bg_gen_freq, cfg2.gen_bg = 4, 10 ** 9, 1, 10 ** 9, False
        cfg2.save_path, cfg2.logger, cfg2.contrastive, cfg2.max_ctx_num, cfg2.beta = tmp, None, False, NC, 0
        cfg2.close_after_train, cfg2.host_u8 = False, False
        model2 = cls(cfg2).to(device)
        tr2 = ModelTrainer(model=model2, loss=loss_fn, optimizer=torch.optim.Adam(model2.parameters(), lr=1e-4), config=cfg2, data=HostLoader())
        tr2.train()
        tr2.iterations = 10 ** 9
        for timed in (False, True):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for it in range(5, 5 + n_ref):
                tr2._prefetch = 2      # what train() sets far from a validation round: two batches may be drawn ahead
                tr2._train_iter(it)
            torch.cuda.synchronize()
            if timed:
                out["reference_style_fp32_over_pcie_ms_per_iter"] = 1e3 * (time.perf_counter() - t0) / n_ref
        # the promoted loop with `losses.item()` read right BEHIND every step, as the reference does (config.lagged_loss_log = False): a
        # host sync that keeps iteration k + 1 from being launched before k is done.  The trainer's default reads, logs and checks every
        # iteration's loss one iteration late instead (same log, same exit on a non-finite loss, same files: model_trainer._lagged_log).
        cfg3 = make_cfg(w, device)
        cfg3.iterations, cfg3.val_freq, cfg3.val_iters, cfg3.bg_gen_freq, cfg3.gen_bg = 4, 10 ** 9, 1, 10 ** 9, False
        cfg3.save_path, cfg3.logger, cfg3.contrastive, cfg3.max_ctx_num, cfg3.beta = tmp, None, False, NC, 0
        cfg3.close_after_train, cfg3.lagged_loss_log = False, False
        model3 = cls(cfg3).to(device)
        tr3 = ModelTrainer(model=model3, loss=loss_fn, optimizer=torch.optim.Adam(model3.parameters(), lr=1e-4), config=cfg3, data=HostLoader())
        tr3.train()
        tr3.iterations = 10 ** 9
        for timed in (False, True):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for it in range(5, 5 + n_ref):
                tr3._prefetch = 2      # what train() sets far from a validation round: two batches may be drawn ahead
                tr3._train_iter(it)
            torch.cuda.synchronize()
            if timed:
                out["reference_style_sync_every_iter_ms_per_iter"] = 1e3 * (time.perf_counter() - t0) / n_ref
    # the same sequence with nothing promoted: eager autograd, torch.optim.Adam over ~70 tensors, the copy in front of the step
    model = cls(make_cfg(w, device)).to(device)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    n_eager = max(5, iters // 2)
    for timed in (False, True):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_eager):
            cx, qx, cy, qy = (t.to(device) for t in host)
            opt.zero_grad()
            loss = loss_fn.calc_loss(model(cx, cy, qx)[0], None, qy)
            loss.backward()
            opt.step()
            loss.item()
        torch.cuda.synchronize()
        if timed:
            out["reference_style_unpromoted_ms_per_iter"] = 1e3 * (time.perf_counter() - t0) / n_eager
    # replayed loop
    model = cls(make_cfg(w, device)).to(device)
    opt = FlatAdam(model, lr=1e-4, ctx_num=NC, test_num=NQ, capturable=True)
    ing = BatchIngest(device)
    ing.stage(*hb)
    cx, qx, cy, qy = ing.take()

    def it():
        opt.zero_grad()
        with loss_grad_in_backward():
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
    with capture_graph(graph, side):
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
                          "reference_style_sync_every_iter": 1e3 * T_LOCAL / out["reference_style_sync_every_iter_ms_per_iter"],
                          "reference_style_unpromoted": 1e3 * T_LOCAL / out["reference_style_unpromoted_ms_per_iter"],
                          "replayed": 1e3 * T_LOCAL / out["replayed_ms_per_iter"]}
    out["adam_steps_taken"] = int(opt.step_dev.item())
    return out


def measure_train_loop_3d(w, device, loss_fn, iters):
    """c5's model through the reference's own training sequence (train.py:52-90: torch.optim.Adam(model.parameters(), lr), a loader
    handing out fp32 host batches, ModelTrainer(...).train() with loss.item() every iteration), twice: as trainer.ModelTrainer runs it
    by default (round 5: FlatAdam over ResNetNP.flat_layout, gradients in the mirror arena, hipGraph replay, the Bayes-by-backprop eps
    staged per step and drawn on host threads under the previous step, host batches on a copy stream) and with every promotion off
    (eager autograd, torch's optimizer over 144 tensors, 52 lazy `normal_()` + `.to(device)` draws per forward)."""
    import importlib
    import tempfile
    from mlhot import binding, synth
    from trainer.model_trainer import ModelTrainer
    cls = getattr(importlib.import_module("networks." + w["method"]), w["method"])
    host = tuple(synth.get_batch_3d(w["T"], NC, NQ, seed=1234))

    class HostLoader:
        def get_batch(self, source, tasks_per_batch, shot):
            return host

        def gen_bg(self, *a, **k):
            pass

    out = {}
    for name, promoted in (("reference_style_ms_per_iter", True), ("reference_style_unpromoted_ms_per_iter", False)):
        with tempfile.TemporaryDirectory() as tmp:
            cfg = make_cfg(w, device)
            cfg.iterations, cfg.val_freq, cfg.val_iters, cfg.bg_gen_freq, cfg.gen_bg = 4, 10 ** 9, 1, 10 ** 9, False
            cfg.save_path, cfg.logger, cfg.contrastive, cfg.max_ctx_num = tmp, None, False, NC
            cfg.close_after_train = False                  # the timed loop below keeps calling _train_iter: tr.close() in the finally
            if not promoted:
                cfg.promote_optimizer, cfg.graph_steps, cfg.host_prefetch = False, False, False
            model = cls(cfg).to(device)
            try:
                tr = ModelTrainer(model=model, loss=loss_fn, optimizer=torch.optim.Adam(model.parameters(), lr=1e-4), config=cfg, data=HostLoader())
                tr.train()                                  # eager warm-up of the batch shape, capture, two replays (+ the final checkpoint)
                tr.iterations = 10 ** 9
                n = max(20, 2 * iters) if promoted else max(5, iters // 2)
                for timed in (False, True):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for it in range(5, 5 + n):
                        tr._prefetch = 2      # what train() sets far from a validation round: two batches may be drawn ahead
                        tr._train_iter(it)
                    torch.cuda.synchronize()
                    if timed:
                        out[name] = 1e3 * (time.perf_counter() - t0) / n
                if promoted:
                    out["reference_style_promoted"] = {"optimizer": type(tr.optimizer).__name__, "graph_replay": bool(tr._graph_default),
                                                       "host_batch_prefetch": tr._host_prefetch is not None,
                                                       "eps_host_threads": len(tr._eps._pieces) if tr._eps and tr._eps._pieces else 1}
                    if tr._eps and tr._eps._worker is not None:
                        tr._eps.stage()                     # collect the last prefetch: the CPU generator is free again
            finally:
                if "tr" in locals():
                    tr.close()
                binding.set_grad_arena(None)
    out["tasks_per_s"] = {"reference_style": 1e3 * w["T"] / out["reference_style_ms_per_iter"],
                          "reference_style_unpromoted": 1e3 * w["T"] / out["reference_style_unpromoted_ms_per_iter"]}
    return out


SHIPPED = {"c3": ("shipped_ANP_ShapeNet1D.yaml", "cfg/train/ANP_ShapeNet1D.yaml:10-11; dataset/shapenet_1d.py:120,139-141"),
           "c5": ("shipped_ANPMR_ShapeNet3D.yaml", "cfg/train/ANPMR_ShapeNet3D.yaml:9-10; dataset/shapenet_3d.py:110,200-204")}


def measure_shipped_cfg(w, device, loss_fn, iters):
    """extras.shipped_cfg: tasks/s of trainer.ModelTrainer's promoted loop (the reference's calling sequence, train.py:52-90: a
    torch.optim.Adam over the model's parameters, a loader of fp32 host batches, loss.item() every iteration) at the reference's
    SHIPPED training shape - not the 16- / 8-task BASELINE configs: ANPShapeNet1D 10 tasks, context ~ U{3..15}, 15 targets; ANPMRShapeNet3D
    20 tasks, context ~ U{1..15}, targets = the other 30 - Nc views.  Image counts per step are then not 480 / 360: the persistent band
    decompositions see 180 .. 300 (1D) resp. 2 x 600 (3D) images.  One hipGraph per context size (13 / 15), a pool of one host batch
    per size handed out in a seeded random order (synthesising a batch on the host costs 100 x the step)."""
    import importlib
    import tempfile
    import numpy as np
    import yaml
    from configs.config import Config
    from mlhot import binding, synth
    from trainer.model_trainer import ModelTrainer
    name, cite = SHIPPED[w["key"]]
    with open(os.path.join(ROOT, "what-matters-for-meta-learning_amd", "cfg", "train", name), "rb") as f:
        raw = yaml.safe_load(f)
    raw["device"] = str(device)
    cfg = Config()
    cfg.set_init_values(raw, side_effects=False)
    assert cfg.method == w["method"]
    T, shot = int(cfg.tasks_per_batch), int(cfg.max_ctx_num)
    three_d = w["kind"] == "resnet3d"
    sizes = list(range(1 if three_d else 3, shot + 1))
    pool = {}
    for nc in sizes:
        if three_d:
            cx, qx, cy, qy = synth.get_batch_3d(T, 15, 15, seed=100 + nc, task_aug=False)
            px, py = torch.cat([cx, qx], dim=1), torch.cat([cy, qy], dim=1)                  # the object's 30 views
            pool[nc] = (px[:, :nc].contiguous(), px[:, nc:].contiguous(), py[:, :nc].contiguous(), py[:, nc:].contiguous())
        else:
            hb = synth.get_batch_u8("shapenet_1d", T, nc, shot, seed=100 + nc)
            pool[nc] = (synth.host_convert(hb[0]), synth.host_convert(hb[1]), hb[2], hb[3])     # bytes / 255, channel-first: what the loader hands out

    class Loader:
        def __init__(self):
            self.rng, self.drawn = np.random.RandomState(0), []

        def get_batch(self, source, tasks_per_batch, shot):
            nc = int(self.rng.randint(sizes[0], shot + 1))
            self.drawn.append(nc)
            return pool[nc]

        def gen_bg(self, *a, **k):
            pass

    out = {"yaml": "cfg/train/" + name, "reference": cite, "tasks_per_batch": T, "context": f"U{{{sizes[0]}..{shot}}} per batch",
           "targets": "30 - Nc" if three_d else str(shot)}
    with tempfile.TemporaryDirectory() as tmp:
        cfg.iterations, cfg.val_freq, cfg.val_iters, cfg.bg_gen_freq, cfg.gen_bg = 3 * len(sizes) + 8, 10 ** 9, 1, 10 ** 9, False
        cfg.save_path, cfg.logger, cfg.contrastive, cfg.close_after_train = tmp, None, False, False
        model = getattr(importlib.import_module("networks." + w["method"]), w["method"])(cfg).to(device)
        data = Loader()
        tr = None
        try:
            tr = ModelTrainer(model=model, loss=loss_fn, optimizer=torch.optim.Adam(model.parameters(), lr=cfg.lr), config=cfg, data=data)
            tr.train()                                   # every size's eager warm-up + capture happens in here or in the untimed round below
            tr.iterations = 10 ** 9
            n = max(150 if three_d else 400, 4 * iters)      # ~0.2 / 0.5 s: a 60-iteration window moved by 15 % between boxes (one first-use allocation = hundreds of iterations' worth)
            for timed in (False, True):
                torch.cuda.synchronize()
                d0 = len(data.drawn)
                t0 = time.perf_counter()
                for it in range(1000, 1000 + n):
                    tr._prefetch = 2      # what train() sets far from a validation round: two batches may be drawn ahead
                    tr._train_iter(it)
                torch.cuda.synchronize()
                if timed:
                    ms = 1e3 * (time.perf_counter() - t0) / n
                    out.update(ms_per_iter=ms, tasks_per_s=1e3 * T / ms, iterations=n, mean_context=float(np.mean(data.drawn[d0:])))
            out["graphs"] = len([v for v in tr._graphs.values() if isinstance(v, tuple)])
            out["promoted"] = {"optimizer": type(tr.optimizer).__name__, "graph_replay": bool(tr._graph_default),
                               "host_batches_as_bytes": bool(tr._host_prefetch is not None and tr._host_prefetch.u8 is not None and tr._host_prefetch.u8.shipped)}
            if tr._eps and tr._eps._worker is not None:
                tr._eps.stage()
        finally:
            if tr is not None:
                tr.close()
            binding.set_grad_arena(None)
    return out


def measure_extras(w, device, loss_fn, batch, iters):
    """Not the headline metric: (a) the training forward alone (activations saved, no backward), (b) the full step
    followed by the fused flat Adam update (mlhot.optim.FlatAdam: one launch over the flat parameter buffer).  The Adam
    leg is timing only: inside a replayed graph the bias-correction step count is frozen at capture time."""
    import importlib
    from mlhot.optim import FlatAdam
    from mlhot.ops import loss_grad_in_backward
    cx, qx, cy, qy = batch
    try:
        model = getattr(importlib.import_module("networks." + w["method"]), w["method"])(make_cfg(w, device)).to(device)
        opt = FlatAdam(model, lr=1e-4, ctx_num=NC, test_num=NQ)

        def fwd():
            return loss_fn.calc_loss(model(cx, cy, qx)[0], None, qy)

        def step_adam():
            model.zero_grad(set_to_none=True)
            with loss_grad_in_backward():
                loss_fn.calc_loss(model(cx, cy, qx)[0], None, qy).backward()
            opt.step()

        out = {"fwd_only_ms": _time_graph(fwd, iters), "step_with_flat_adam_ms": _time_graph(step_adam, iters),
               "note": "hipGraph replays; informational, not the headline metric"}
        # opt-in arithmetic, never the headline: conv2's forward on the bf16 matrix pipe over hi / mid / lo split fp32 operands
        # (csrc/conv_split.h; the option is read when the graph is captured)
        from mlhot import lib

        def step():
            model.zero_grad(set_to_none=True)
            with loss_grad_in_backward():
                loss_fn.calc_loss(model(cx, cy, qx)[0], None, qy).backward()

        # bits: 1 forward, 2 data gradient, 4 weight gradient.  Its own roof next to the fp32 one: the bf16 matrix pipe's dense 2.5 PFLOP/s
        # (MI355X_MICROARCH.md) / 6 piece products = 417 fp32-equivalent TFLOP/s (fp32 MFMA: 157.3)
        def split_run(bits):
            lib().set_option("conv2_split", bits)
            try:
                r = {"fwd_only_ms": _time_graph(fwd, iters), "step_ms": _time_graph(step, iters)}
            finally:
                lib().set_option("conv2_split", 0)
            r["tasks_per_s"] = round(w["T"] / (r["step_ms"] * 1e-3), 1)
            return r
        out["split_precision_conv2"] = {
            "forward": split_run(1), "forward_and_gradients": split_run(7),
            "dtype": "bf16x3-split fp32 (conv2 of the vanilla encoder only; fp32 everywhere else)",
            "arithmetic": "v_mfma_f32_16x16x32_bf16 over 3-piece nearest-even bf16 splits of both fp32 operands (x = hi + mid + lo exactly), "
                          "6 of 9 piece products (dropped: below 2^-26 of a product), fp32 accumulation; csrc/conv_split.h",
            "roof": {"peak_tflops_fp32_equivalent": round(2500.0 / 6, 1), "from": "2.5 PFLOP/s dense bf16 / 6 products", "fp32_mfma_peak": PEAK_FP32_MFMA_TFLOPS},
            "error_vs_float64": "tests/test_gpu_parity.py::test_split_precision_error_vs_fp32_mfma (forward 0.2-0.8 x the fp32 kernels' error, gradients 0.2-2.5 x)",
            "note": "opt-in (mlhot_set_option conv2_split); never in `value`"}
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
        if w["key"] in SHIPPED:
            out["shipped_cfg"] = {w["method"]: measure_shipped_cfg(w, device, loss_fn, iters)}
        return out
    except Exception as e:  # noqa: BLE001 - extras must never break the bench line
        return {"error": f"{type(e).__name__}: {e}"}


def measure_other_configs(keys, steps, warmup, timeout_s=600):
    """extras.configs: the other BASELINE configs that fit one GPU - c2 (configs[1]) and c5 (the per-GPU share of configs[4]) - each
    timed by a CHILD process running this file with `--workload <key>` under the same protocol (pre-warm, W warm-up steps, EXACTLY K
    timed steps between fences, hipGraph replay, the per-launch HIP-event roofline leg), after this process's own timed region is
    over and its GPU work drained.  A child is a fresh process (started, never exec'ed over this one); `value` stays c3's."""
    import subprocess
    out = {}
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                            "MLHOT_BENCH_SEQ", "MLHOT_BENCH_KERNELS", "MLHOT_BENCH_LAUNCHER", "MLHOT_FORCE_COLLECTIVES")}
    for key in keys:
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--workload", key, "--steps", str(steps), "--warmup", str(warmup),
               "--no-extras", "--no-cpu-baseline", "--no-configs"] + (["--shipped-cfg"] if key == "c5" else [])
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                out[key] = {"error": f"rc {r.returncode}: {(r.stderr or '').strip()[-300:]}"}
                continue
            j = json.loads(lines[-1])
            roof = j.get("roofline") or {}
            out[key] = {"workload": j["config"]["workload"], "ms_per_step": j["ms_per_step"], "ms_per_step_event_median": j.get("ms_per_step_event_median"),
                        "tasks_per_s": j["value"], "steps": j["steps"], "warmup": j["warmup"], "hipgraph": j.get("hipgraph"),
                        "steps_per_graph": j.get("steps_per_graph"), "final_loss": j.get("final_loss"),
                        "roofline": {k: roof.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us",
                                                              "launches_per_step", "alg_flops_per_launch")},
                        "fwd_only_ms": (roof.get("forward") or {}).get("fwd_ms"),
                        "launches_per_step": j.get("launches_per_step"), "gpu_busy_us_per_step": j.get("gpu_busy_us_per_step"),
                        "eps_source": j["config"].get("eps_source"), "eps": j.get("eps"), "library_sha256": (j.get("library") or {}).get("sha256"),
                        "shipped_cfg": (j.get("extras") or {}).get("shipped_cfg"),
                        "wall_s": round(time.perf_counter() - t0, 1)}
        except Exception as e:  # noqa: BLE001 - extras must never break the bench line
            out[key] = {"error": f"{type(e).__name__}: {e}"}
    return out


def forward_roofline(w, fwd_ms):
    """the training forward against both roofs.  FLOPs: algorithmic (SURVEY's per-task figure).  HBM bytes, twice:
    compulsory (inputs + parameters once: what a perfectly fused forward has to move) and measured (rocprofv3 FETCH_SIZE x 2 +
    WRITE_SIZE of the forward's kernels, profiles/pmc_traffic.json, when those passes were collected for this workload)."""
    t = fwd_ms * 1e-3
    flops = w["fwd_gflop_per_task"] * 1e9 * w["T"]
    img_bytes = {"128x128x1": 65536, "64x64x3": 49152}[w["image"]]
    n_dec = NQ if w["kind"] != "vanilla" else 0                      # the decoder ResNet re-reads the target images
    compulsory = w["T"] * (NC + NQ + n_dec) * img_bytes + {"c3": 1.96e6, "c2": 1.45e6, "c5": 17.2e6}.get(w.get("key", ""), 0.0)
    measured = None
    for name in ("pmc_traffic.json", f"pmc_traffic_{w.get('key')}.json"):       # the per-workload file, when present, is the one read
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                tr = json.load(f)
            if tr.get("_workload", "c3") == w.get("key"):
                fw = [v["hbm_bytes"] for k, v in tr.items() if isinstance(v, dict) and ".bwd." not in k and "hbm_bytes" in v]
                measured = float(sum(fw)) if fw else measured
        except (OSError, ValueError):
            pass
    return {"fwd_ms": fwd_ms, "alg_gflop": flops / 1e9, "achieved_tflops": flops / t / 1e12, "achieved_flops_frac": flops / t / (PEAK_FP32_MFMA_TFLOPS * 1e12),
            "hbm_compulsory_bytes": compulsory, "achieved_hbm_compulsory_frac": compulsory / t / PEAK_HBM_BYTES_S,
            "hbm_measured_bytes": measured, "achieved_hbm_measured_frac": (measured / t / PEAK_HBM_BYTES_S) if measured else None,
            "note": "the fused forward is bound by the fp32 matrix pipe, not by HBM (DESIGN.md §5): north_star's '>= 30 % of the HBM roofline' "
                    "corresponds to >= 63 % of the fp32 MFMA peak over the whole forward"}


def timed_region(run, steps, warmup, world, device, sync=None, record=None):
    """The contract's timing protocol: `warmup` untimed steps, then EXACTLY `steps` steps bracketed by fences (synchronize +
    barrier + synchronize) on both sides; returns (seconds of the slowest rank, host enqueue seconds, last step's result).
    `sync`: device synchronisation (torch.cuda.synchronize; a no-op for the CPU control-flow test); `record(i, 0 | 1)`: optional
    per-step event hooks."""
    import torch.distributed as dist
    sync = sync or (lambda: None)

    multi = world > 1 or (dist.is_available() and dist.is_initialized())     # a forced world of one (mlhot.dist.force_collectives) runs the protocol too

    def fence():
        sync()
        if multi:
            dist.barrier()
            sync()

    for _ in range(warmup):
        run()
    fence()
    t0 = time.perf_counter()
    out = None
    for i in range(steps):
        if record:
            record(i, 0)
        out = run()
        if record:
            record(i, 1)
    t_enqueue = time.perf_counter() - t0
    fence()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    return elapsed, t_enqueue, out


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, child_cmd, env=None, port=None, poll_s=0.05, grace_s=20.0, out=None, err=None):
    """`python bench.py --gpus N` without a launcher: start N rank processes of `child_cmd` (a list; each gets RANK = LOCAL_RANK = r,
    WORLD_SIZE = n, MASTER_ADDR = 127.0.0.1 and one free MASTER_PORT), wait for all of them, relay rank 0's stdout - the ONE JSON
    line(s) starting with '{' - to `out` and everything else (library chatter, the other ranks' stdout; prefixed) to `err`; return the WORST exit code.  When a rank dies
    the others would sit in a collective until its timeout, so they are given `grace_s` and then terminated - by PID, these are
    this process's own children.  The caller must not have touched the GPU: the children are fresh processes (no fork of a HIP
    context, no exec of a process that initialised one)."""
    import subprocess
    import tempfile
    out, err = out or sys.stdout, err or sys.stderr
    env = dict(os.environ if env is None else env)
    env.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port or _free_port()))
    env.setdefault("NCCL_DEBUG", "VERSION")              # RCCL prints its version line on stderr: evidence of the backend that ran
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC (the only kind this host driver supports) for RCCL's P2P setup
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    procs, files = [], []
    try:
        for r in range(n):
            f = tempfile.TemporaryFile(mode="w+")
            files.append(f)
            procs.append(subprocess.Popen(list(child_cmd), env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=f, stderr=None))
        codes, first_fail = [None] * n, None
        while any(c is None for c in codes):
            for r, p in enumerate(procs):
                if codes[r] is None:
                    codes[r] = p.poll()
                    if codes[r] not in (None, 0) and first_fail is None:
                        first_fail = time.monotonic()
                        print(f"[bench] rank {r} exited with code {codes[r]}; the other ranks get {grace_s:.0f} s", file=err)
            if first_fail is not None and time.monotonic() - first_fail > grace_s:
                for r, p in enumerate(procs):
                    if codes[r] is None:
                        p.terminate()
                first_fail = float("inf")
            time.sleep(poll_s)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, f in enumerate(files):
        f.seek(0)
        for ln in f.read().splitlines():                     # rank 0's JSON line(s) are the product; library chatter on stdout ("[Gloo] Rank 0 is
            if r == 0 and ln.startswith("{"):                # connected to ...") and the other ranks' stdout go to stderr
                print(ln, file=out)
            else:
                print(f"[rank {r}] {ln}", file=err)
        f.close()
    out.flush()
    worst = max(codes, key=lambda c: (c != 0, abs(c)))
    return worst if worst >= 0 else 128 - worst          # a rank killed by signal s: 128 + s, as a shell reports it


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prof-steps", type=int, default=5)
    ap.add_argument("--steps-per-graph", type=int, default=5,
                    help="single-GPU vanilla workloads: this many consecutive steps are captured into ONE hipGraph (every step is still "
                         "zero_grad + forward + loss + backward on the resident batch; --steps of them are timed in all).  Between two graph "
                         "launches the GPU idles ~9 us (rocprofv3 kernel trace: 0.0 us between the kernels of a graph, 8.8 us between graphs) - "
                         "1.5 %% of a 0.6 ms step that no kernel owns.  1 = one step per graph.  Ignored (1) with more than one rank (the "
                         "all-reduce sits between steps) and for the Bayes-by-backprop workload (fresh eps per step)")
    ap.add_argument("--no-graph", action="store_true", help="run the step eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--one-graph", action="store_true",
                    help="N > 1, ResNet family: ONE graph per step and one gradient all-reduce behind it (A/B; default: two graphs with the "
                         "early bucket's all-reduce between them)")
    ap.add_argument("--strict", action="store_true",
                    help="N > 1: FAVOR+'s key stabiliser is the maximum over the WHOLE meta-batch, as in the reference's single-process batch "
                         "(fast_attention.py:96-97), through mlhot.dist.StabiliserExchange - two scalar collectives per attention pass between "
                         "the staged C calls, so the step runs eagerly (implies --no-graph).  Default: each rank's own maximum "
                         "(negligible for ANPShapeNet1D, up to 15 %% of a gradient's scale for the d = 256 models, DESIGN.md section 6d)")
    ap.add_argument("--no-extras", action="store_true", help="skip the fwd-only / +Adam timing legs")
    ap.add_argument("--shipped-cfg", action="store_true", help="with --no-extras: still run the extras.shipped_cfg leg (the reference's shipped training shape)")
    ap.add_argument("--no-configs", action="store_true",
                    help="default workload, one GPU: skip extras.configs (c2 and c5 timed by child processes of this file after the headline's timed region)")
    ap.add_argument("--no-prewarm", action="store_true", help="skip the ~0.1 s of untimed steps in front of the W warm-up steps")
    ap.add_argument("--no-loss-aside", action="store_true",
                    help="A/B: the loss VALUE as a launch of its own between forward and backward (mlhot_loss_fwd) instead of one extra workgroup of "
                         "the model's first backward kernel (mlhot.ops.loss_value_aside; vanilla workloads)")
    ap.add_argument("--no-loss-plus", action="store_true",
                    help="A/B (c5): `loss + kl * beta` as two launches per direction (mlhot_loss_fwd + mlhot_axpy) instead of inside the loss's own (mlhot_loss_plus_*)")
    ap.add_argument("--dbg", type=int, default=0, help="kernel timing experiments (results become WRONG; never a bench line)")
    ap.add_argument("--eps", choices=("host", "device"), default="host",
                    help="c5: where the Bayes-by-backprop eps stream is produced - the torch CPU generator (the reference's route, bit-exact; "
                         "default) or the same MT19937 stream continued on the GPU (mlhot_mt19937_normal)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="mlhot_set_option switches for A/B runs, e.g. c12_split=0")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: become one.  Nothing above touched the GPU (torch.cuda.device_count() does not initialise it on this image)
        have = torch.cuda.device_count()
        if have < args.gpus and not os.environ.get("MLHOT_ONE_DEVICE"):
            print(f"[bench] --gpus {args.gpus} but this node shows {have} GPU(s)", file=sys.stderr)
            sys.exit(2)
        print(f"[bench] no WORLD_SIZE in the environment: starting {args.gpus} rank processes (one per GPU, rendezvous on 127.0.0.1)", file=sys.stderr)
        sys.exit(spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=dict(os.environ, MLHOT_BENCH_LAUNCHER="self")))
    w = dict(WORKLOADS[args.workload], key=args.workload)
    T = w["T"]
    c5 = w["kind"] == "resnet3d"

    import importlib
    import torch.distributed as dist
    import mlhot
    from mlhot import dist as mdist
    from mlhot.ops import add_scaled, loss_value_aside
    from trainer.losses import LossFunc
    # MLHOT_DIST_BACKEND / MLHOT_ONE_DEVICE: test hooks (e.g. two gloo ranks sharing the only GPU of a 1-GPU box, to
    # exercise the N>1 control flow); the driver's launch uses neither (nccl = RCCL, one rank per GPU)
    rank, local, world = mdist.init_from_env(os.environ.get("MLHOT_DIST_BACKEND"))
    if world != args.gpus:
        print(f"[bench] --gpus {args.gpus} but the launcher's WORLD_SIZE is {world}: start me with --gpus {world}, or without a launcher", file=sys.stderr)
        sys.exit(2)
    if os.environ.get("MLHOT_ONE_DEVICE"):
        local = 0
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    mlhot.build_product()
    if args.dbg:
        mlhot.lib().set_option("dbg", args.dbg)
    for kv in args.opt:
        name, _, val = kv.partition("=")
        mlhot.lib().set_option(name, int(val))
    stabiliser = "single process: the batch maximum"
    if world > 1 or mdist.force_collectives():
        stabiliser = "rank-local maximum (graph replay; --strict for the reference's batch-global form)"
        if args.strict:
            from mlhot import ops as mops
            mops.set_stabiliser_exchange(mdist.StabiliserExchange(dedicated_group=True))
            args.no_graph = True
            stabiliser = "batch-global maximum over all ranks (StabiliserExchange, eager steps)"
    if (world > 1 or mdist.force_collectives()) and rank == 0:
        print(f"[bench] key stabiliser: {stabiliser}", file=sys.stderr)
    if world > 1 and rank == 0:
        print(f"[bench] backend {dist.get_backend()} (RCCL when 'nccl') reports {dist.get_world_size()} ranks; rank 0 on "
              f"{torch.cuda.get_device_name(device)}; NCCL_DEBUG={os.environ.get('NCCL_DEBUG', '-')}", file=sys.stderr)

    model = getattr(importlib.import_module("networks." + w["method"]), w["method"])(make_cfg(w, device)).to(device)
    loss_fn = LossFunc("mse", w["task"])
    cx, qx, cy, qy = make_batch(w, 1234 + rank, device)
    # the trainer's own setting (trainer/model_trainer.py): collectives on a communication stream, and - for the models that
    # name them - the gradients that are complete early in the backward as a bucket of their own
    if (world > 1 or mdist.force_collectives()) and hasattr(model, "enable_flat_grads"):
        model.enable_flat_grads()            # ResNet / BBB family: gradients in one flat buffer, all-reduced in place (mlhot/arena.py)
    bucket = mdist.GradBucket(model.parameters(), side_stream=True,
                              early=model.early_grad_parameters() if hasattr(model, "early_grad_parameters") else None)
    beta = w.get("beta", 0.0)
    seed = torch.ones((), device=device)     # d loss / d loss: torch.ones_like(loss) allocated once instead of per step

    # More than one rank, ResNet family: the step runs as TWO graphs - everything down to the image trunks' inputs, then the trunks'
    # backward - with the early bucket's all-reduce (11.6 of c5's 15.1 MB) issued between the two replays, under the second
    # (mlhot.dist.backward_in_two; the model cuts its autograd graph in front of the trunks).  --one-graph: the single graph + one
    # collective behind it, for the A/B.
    split = ((world > 1 or mdist.force_collectives()) and hasattr(model, "enable_split_backward") and not args.one_graph
             and not args.no_graph and not args.strict)
    if split:
        model.enable_split_backward(True)

    def fwd_bwd(arm=False, batch=None, part=None):
        """part: None = the whole step; 1 = forward + the backward above the cut; 2 = the trunks' backward (split only)."""
        if part == 2:
            mdist.run_second_part(model)
            return None
        bx, by, tx, ty = batch if batch is not None else (cx, cy, qx, qy)
        model.zero_grad(set_to_none=True)
        mu, var, kl = model(bx, by, tx)
        # the loss VALUE comes out of the model's first backward kernel (one extra workgroup; mlhot.ops.loss_value_aside) - the step reads
        # it only after the backward, as trainer.ModelTrainer does.  Not for c5: its objective adds kl * beta to the value right here.
        with loss_value_aside(enabled=not args.no_loss_aside and not c5):
            if c5:      # loss + kl * beta as trainer.ModelTrainer writes it (LossFunc.calc_objective: the sum inside the loss's launches, the bits of
                loss = loss_fn.calc_objective(mu, var, ty, kl, beta) if not args.no_loss_plus else add_scaled(loss_fn.calc_loss(mu, var, ty), kl, beta)
                # loss + kl * beta); identical on every rank (same weights, kl does not depend on the batch): averaged, never summed
            else:
                loss = loss_fn.calc_loss(mu, var, ty)
            if arm:
                bucket.arm()                     # eager steps only: the early bucket's all-reduce is issued from inside backward()
            loss.backward(gradient=seed)         # the constant 1.0 autograd would otherwise make with a fill kernel every step
            if split and part is None:
                mdist.run_second_part(model)     # (the armed early bucket went out when part 1's last gradient landed)
        return loss.detach()

    # Bayes-by-backprop eps: the reference draws them on the torch CPU generator inside the forward (bbb/BBBConv.py:88).  Every rank
    # seeds that generator alike, so all ranks sample the same weights; the draws of step k+1 run on a host
    # thread while step k is on the GPU (networks/bbb/eps.py).
    eps = None
    if c5:
        from networks.bbb.eps import StagedEps
        torch.manual_seed(1234)
        eps = StagedEps(device, source=args.eps)
        with eps.recording():
            fwd_bwd()
        torch.cuda.synchronize()

    def step():                              # eager form (also the profiled one)
        if eps is not None:
            eps.stage()
            with eps.active():
                loss = fwd_bwd(arm=True)
        else:
            loss = fwd_bwd(arm=True)
        bucket.sync(defer_scale=True)        # the 1/world average rides in the optimizer's gradient scale (FlatAdam.step(grad_scale=...))
        return loss

    # The step has static shapes and no host sync, so it is captured once into a hipGraph and replayed
    # (the RCCL all-reduce stays outside the graph).  --no-graph runs it eagerly.
    run = step
    graphed = False
    spg = 1
    if not args.no_graph:
        try:
            spg = args.steps_per_graph if (world == 1 and not mdist.force_collectives() and w["kind"] == "vanilla") else 1
            if spg < 1 or args.steps % spg or args.warmup % spg:
                spg = 1                      # K and W must be whole graphs: the timed region is EXACTLY --steps steps

            def body(part=None):
                if eps is not None and part != 2:
                    eps.rewind()
                out = None
                for _ in range(spg):
                    out = fwd_bwd(part=part)
                return out
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            if eps is not None:
                eps.stage()
            with torch.cuda.stream(side), (eps.active() if eps is not None else contextlib_null()):
                for _ in range(3):
                    body()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            model.zero_grad(set_to_none=True)
            graph2 = None
            with (eps.active() if eps is not None else contextlib_null()), capture_graph(graph, side):   # the warm-up's stream: the parameters' grad-accumulation nodes live there
                static_loss = body(part=1 if split else None)
            if split:                        # the trunks' backward: a second graph in the first one's memory pool (always replayed in this order)
                graph2 = torch.cuda.CUDAGraph()
                with (eps.active() if eps is not None else contextlib_null()), capture_graph(graph2, side, pool=graph.pool()):
                    body(part=2)

            def run():
                if eps is not None:
                    eps.stage()              # collect the prefetched draws (or draw now) + one pinned H2D copy
                    if eps.source == "host":
                        eps.prefetch()       # next step's draws, on a host thread, while this step runs (the device source looks ahead by itself)
                graph.replay()
                if graph2 is not None:
                    bucket.issue_early()     # on the communication stream, behind graph 1 and beside graph 2
                    graph2.replay()
                bucket.sync(defer_scale=True)
                return static_loss
            graphed = True
        except Exception as e:  # noqa: BLE001 - fall back to eager, say so
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            run = step

    spg = spg if graphed else 1
    n_run = args.steps // spg                                # graph launches in the timed region: n_run x spg = exactly --steps steps
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_run)]
    # ~0.1 s of the same step before the protocol's W warm-up steps: a default run is 6 ms of warm-up and 30 ms of timed region, short
    # enough for the GPU's clock / power state to still be settling inside it (one run in six showed a 10 % slower wall clock at
    # an unchanged per-step event median).  Untimed; the W warm-up steps and the EXACTLY K timed steps follow as the contract says.
    prewarm = 0 if args.no_prewarm else max(1, int(0.1 / (5e-4 * spg if w["kind"] == "vanilla" else 1.5e-3)))
    for _ in range(prewarm):
        run()
    torch.cuda.synchronize()
    elapsed, t_enqueue, loss = timed_region(run, n_run, args.warmup // spg, world, device, sync=torch.cuda.synchronize,
                                            record=lambda i, k: ev[i][k].record())
    final_loss = loss.item()
    replay_collectives = [list(e) for e in bucket.issue_log]       # of the timed region's last step (the roofline leg below runs eager steps)
    step_ms = sorted(a.elapsed_time(b) / spg for a, b in ev)       # device time per step (events on the replaying stream, per graph launch / steps per graph)
    if eps is not None and eps._worker is not None:
        eps.stage()                                          # collect the last prefetch: the CPU generator is free again

    # ---- roofline leg: per-launch HIP-event timing of every kernel over a few more (eager) steps ----------
    roof, kernels = None, None
    if args.prof_steps > 0:
        L = mlhot.lib()
        L.prof_begin(16384)
        for _ in range(args.prof_steps):
            if graphed and not args.no_prewarm:
                for _ in range(max(1, 4 // spg)):         # the GPU busy in front of every profiled eager step: the eager kernels then
                    run()                                 # run at the clocks of the replayed region (idle in front, they read ~3 % longer)
            step()
        torch.cuda.synchronize()
        recs = L.prof_end()
        if os.environ.get("MLHOT_BENCH_SEQ") and rank == 0:      # launch labels of ONE step, in launch order (scripts/pmc_by_label.py
            per = len(recs) // args.prof_steps                   # folds a rocprofv3 --pmc dispatch list of eager steps onto them)
            with open(os.environ["MLHOT_BENCH_SEQ"], "w") as f:
                json.dump([lb for lb, _ in recs[:per]], f)
        agg = {}
        for label, ms in recs:
            a = agg.setdefault(label, [0, 0.0])
            a[0] += 1
            a[1] += ms
        n_img = T * (NC + NQ)
        kernels = {k: {"launches_per_step": v[0] / args.prof_steps, "avg_us": 1e3 * v[1] / v[0],
                       "us_per_step": 1e3 * v[1] / args.prof_steps} for k, v in agg.items()}
        # algorithmic FLOPs of ALL launches of a label in one step (vanilla: one launch per label)
        per_step = trunk_flops_per_step(T, w["kind"]) if w["kind"] != "vanilla" else {k: alg_flops(k, n_img) * kernels[k]["launches_per_step"] for k in agg if alg_flops(k, n_img)}
        per_step = {k: v for k, v in per_step.items() if k in agg and v}
        if per_step:
            # the kernel with the largest total time; labels within 3 % of it count as tied (the three conv12 kernels are 1 - 2 us apart
            # and would swap places from run to run) and the tie goes to the first in launch order - the forward
            top = max(agg[k][1] for k in per_step)
            order = [lb for lb, _ in recs]
            dom = min((k for k in per_step if agg[k][1] >= 0.97 * top), key=order.index)
            sec_per_step = agg[dom][1] / args.prof_steps * 1e-3
            ach = per_step[dom] / sec_per_step / 1e12
            traffic = None   # HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/)
            for name in ("pmc_traffic.json", f"pmc_traffic_{w['key']}.json"):
                try:
                    with open(os.path.join(ROOT, "profiles", name)) as f:
                        tr = json.load(f)
                    if tr.get("_workload", "c3") == w["key"] and tr.get(dom, {}).get("hbm_bytes"):
                        traffic = tr[dom]["hbm_bytes"] / max(tr[dom].get("launches_per_step", 1), 1) if "launches_per_step" in tr[dom] else tr[dom]["hbm_bytes"]
                except (OSError, ValueError):
                    pass
            lps = kernels[dom]["launches_per_step"]
            roof = {"bound": "mfma", "kernel": dom,
                    "measured_in": "eager steps, two HIP events around every launch, each step behind a few replays of the step graph so that "
                                   "the GPU is in the state of the timed region (events cannot be read out of a replayed graph in this process: "
                                   "scripts/micro/graph_events.hip; with the GPU idle in front, --no-prewarm, the same kernels read ~5 % longer)",
                    "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic, "avg_launch_us": kernels[dom]["avg_us"],
                    "launches_per_step": lps, "alg_flops_per_launch": per_step[dom] / lps,
                    "all_labels_tflops": {k: round(per_step[k] / (agg[k][1] / args.prof_steps * 1e-3) / 1e12, 1) for k in per_step}}
    # ---- the training forward alone, against both roofs ------------------------------------------------------
    fwd = None
    if world == 1 and not args.no_graph:
        try:
            def fwd_only():
                if eps is not None:
                    eps.rewind()
                mu, var, kl = model(cx, cy, qx)
                return loss_fn.calc_loss(mu, var, qy)
            if eps is not None:
                eps.stage()
            with (eps.active() if eps is not None else contextlib_null()):
                fwd = forward_roofline(w, _time_graph(fwd_only, max(10, args.steps // 2)))
        except Exception as e:  # noqa: BLE001
            fwd = {"error": f"{type(e).__name__}: {e}"}
    if roof is not None:
        roof["forward"] = fwd
    # ---- informational extras (SURVEY §8d "also report fwd-only and +Adam"), rank 0, single-GPU runs only -----------
    extras = None
    if world == 1 and not args.no_extras and not args.no_graph and w["kind"] == "vanilla":
        extras = measure_extras(w, device, loss_fn, (cx, qx, cy, qy), max(10, args.steps // 2))
    if world == 1 and not args.no_extras and not args.no_graph and c5:
        try:
            extras = {"variable_context": measure_variable_nc_3d(w, device, loss_fn, model, eps,
                                                                 lambda a, b_, c_, d_: fwd_bwd(batch=(a, c_, b_, d_)), max(10, args.steps // 2)),
                      "note": "informational, not the headline metric"}
            if eps is not None and eps._worker is not None:
                eps.stage()                                      # the trainer below draws from the same CPU generator
            extras["train_loop"] = measure_train_loop_3d(w, device, loss_fn, max(10, args.steps // 2))
        except Exception as e:  # noqa: BLE001 - extras must never break the bench line
            extras = dict(extras or {}, error=f"{type(e).__name__}: {e}")
    if world == 1 and args.shipped_cfg and (args.no_extras or c5) and args.workload in SHIPPED and not args.no_graph:
        try:
            if eps is not None and eps._worker is not None:
                eps.stage()
            extras = dict(extras or {})
            extras["shipped_cfg"] = {w["method"]: measure_shipped_cfg(w, device, loss_fn, max(10, args.steps // 2))}
        except Exception as e:  # noqa: BLE001 - extras must never break the bench line
            extras = dict(extras or {}, shipped_cfg={"error": f"{type(e).__name__}: {e}"})
    if world == 1 and not mdist.force_collectives() and args.workload == "c3" and not args.no_extras and not args.no_configs and not args.no_graph:
        torch.cuda.synchronize()
        extras = dict(extras or {})
        extras["configs"] = measure_other_configs(("c2", "c5"), steps=20, warmup=5)
        moved = (extras["configs"].get("c5") or {}).pop("shipped_cfg", None)            # one place for both shipped shapes
        if moved:
            extras.setdefault("shipped_cfg", {}).update(moved)
    per_rank = None
    if dist.is_initialized():
        # every rank's loss of the last timed step (rank-local: the ranks hold different tasks) and a checksum of its gradients after
        # the last all-reduce (the SUM over ranks, still unscaled: the same bits on every rank, or the collective is broken)
        torch.cuda.synchronize()
        live = [p.grad for p in bucket.params if p.grad is not None]
        mine = torch.stack([torch.as_tensor(final_loss, dtype=torch.float64, device=device),
                            torch.stack([g.double().sum() for g in live]).sum() if live else torch.zeros((), dtype=torch.float64, device=device),
                            torch.stack([g.double().abs().sum() for g in live]).sum() if live else torch.zeros((), dtype=torch.float64, device=device)])
        table = torch.zeros(world, 3, dtype=torch.float64, device=device)
        table[rank] = mine
        dist.all_reduce(table, op=dist.ReduceOp.SUM)            # an all-gather spelled as a sum of one-hot rows (every backend takes it)
        per_rank = {"final_loss": table[:, 0].tolist(), "grad_sum": table[:, 1].tolist(), "grad_abs_sum": table[:, 2].tolist()}
        dist.barrier()

    if rank == 0:
        out = {"metric": METRIC[w["key"]],
               "value": world * T * args.steps / elapsed, "unit": "meta-tasks/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": w["name"], "tasks_per_gpu": T, "global_tasks": world * T,
                          "context_shots": NC, "target_shots": NQ, "image": w["image"],
                          "parallelism": f"task-sharded x{world}, one flat grad all-reduce" if world > 1 else "single GPU"},
               "dist": ({"backend": dist.get_backend(), "ranks_reported_by_backend": dist.get_world_size(),
                         "collectives_per_step": replay_collectives, "per_rank": per_rank,
                         "step_graphs": (2 if split and graphed else 1) if graphed else 0,
                         "launcher": "bench.py itself (spawn_ranks)" if os.environ.get("MLHOT_BENCH_LAUNCHER") == "self" else "external (WORLD_SIZE in the environment)"}
                        if dist.is_initialized() else None),
               "library": _library_id(),
               "final_loss": final_loss, "hipgraph": graphed, "steps_per_graph": spg,
               "steps_per_graph_note": (f"`value` replays {spg} consecutive steps per hipGraph launch (a bench arrangement: ~9 us of launch gap paid once per {spg} "
                                        "steps); a training loop replays ONE iteration per graph - extras.train_loop.replayed_ms_per_iter") if spg > 1 else None, "key_stabiliser": stabiliser, "host_enqueue_ms_per_step": 1e3 * t_enqueue / args.steps,
               "ms_per_step_event_median": step_ms[len(step_ms) // 2], "ms_per_step_event_min": step_ms[0], "ms_per_step_event_max": step_ms[-1],
               "timing": f"value = wall clock over {args.steps} steps ({n_run} graph launches of {spg} step(s)) between two barrier + synchronize fences (max over ranks), "
                         f"after {prewarm * spg} untimed pre-warm steps and the {args.warmup} warm-up steps; "
                         "event_median / _min / _max = per-step HIP-event durations on the replaying stream",
               "roofline": roof}
        if args.opt:
            out["options"] = args.opt
            if any(kv.partition("=")[0] == "conv2_split" and int(kv.partition("=")[2]) for kv in args.opt):
                # an A/B line, not the headline: conv2 ran on the bf16 pipe over split operands (extras.split_precision_conv2)
                out["dtype"] = "f32, conv2 of the vanilla encoder on bf16x3-split operands (opt-in arithmetic: NOT the headline line)"
        if eps is not None:
            # which random stream this line's Bayes-by-backprop weights were sampled from (VERDICT r4 item 4 iv): "host" = the torch CPU
            # generator in the reference's draw order, bit-identical samples, the step then costs max(GPU, host draw); "device" = the same
            # MT19937 stream continued on the GPU (uniforms bit-identical, normals within 4 ulp)
            out["config"]["eps_source"] = eps.source
        if eps is not None and eps.source == "host":
            d0 = time.perf_counter()
            eps.stage()
            from networks.bbb.eps import usable_cores
            out["eps"] = {"source": "host", "floats_per_step": eps._total, "host_draw_ms_per_step": 1e3 * (time.perf_counter() - d0),
                          "host_threads": len(eps._pieces) if eps._pieces else 1,
                          "host_threads_per_rank": {"eps_draw_pieces": len(eps._pieces) if eps._pieces else 1, "eps_drawer": 1, "main": 1},
                          "usable_cores": usable_cores(), "ranks_on_node": int(os.environ.get("LOCAL_WORLD_SIZE") or world),
                          "note": "drawn on the torch CPU generator in the reference's order (bit-identical samples and final generator state) "
                                  "while the previous step runs; the draw is cut into host_threads pieces, each drawn by normal_() on a "
                                  "generator positioned with mlhot_mt19937_advance (MLHOT_EPS_THREADS); a step costs max(GPU time, draw time)"}
        elif eps is not None:
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(eps._dn_stream):
                e0.record()
                eps._dn.draw(out=eps._dn_buf)
                e1.record()
            torch.cuda.synchronize()
            out["eps"] = {"source": "device", "floats_per_step": eps._total, "device_draw_ms_per_step": e0.elapsed_time(e1),
                          "note": "the torch CPU generator's MT19937 stream continued on the GPU (mlhot_mt19937_normal: identical uniforms, "
                                  "normals within 4 ulp of torch's), one step ahead on a side stream; not the bit-exact default"}
            eps.release()
        if extras:
            out["extras"] = extras
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(w)
        else:
            out["cpu_baseline"] = None
        if kernels:
            top = sorted(kernels.items(), key=lambda kv: -kv[1]["us_per_step"])
            out["kernel_us_per_step"] = {k: round(v["us_per_step"], 1) for k, v in top[:14]}
            out["gpu_busy_us_per_step"] = round(sum(v["us_per_step"] for v in kernels.values()), 1)
            out["launches_per_step"] = round(sum(v["launches_per_step"] for v in kernels.values()), 1)
            if os.environ.get("MLHOT_BENCH_KERNELS"):
                with open(os.environ["MLHOT_BENCH_KERNELS"], "w") as f:
                    json.dump({k: v for k, v in top}, f, indent=1)
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


def _library_id():
    """Which build of the HIP library produced this line (profiles/<tag>_MANIFEST.json carries the same sha for the rocprof summaries)."""
    import hashlib
    import mlhot
    path = os.environ.get("MLHOT_LIB") or mlhot.PRODUCT_SO
    try:
        with open(path, "rb") as f:
            from mlhot.build import source_sha256
            return {"path": os.path.relpath(path, ROOT), "sha256": hashlib.sha256(f.read()).hexdigest(), "src_sha256": source_sha256()}
    except OSError:
        return {"path": path, "sha256": None}


def contextlib_null():
    import contextlib
    return contextlib.nullcontext()


if __name__ == "__main__":
    main()
