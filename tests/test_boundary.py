"""Drop-in boundary checks that need no GPU: plugin lookup, state_dict layout, config surface,
C-ABI symbols, loud failure without a HIP device."""
import ctypes
import glob
import importlib
import os
import re

import pytest
import torch
import yaml

from tests import util as U

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", U.model_case_names() + U.resnet_case_names() + U.mr_case_names() + U.fcl_case_names())
def test_plugin_lookup_and_state_dict(name):
    """train.py:41-45: importlib.import_module(f"networks.{method}") + getattr(module, method)(config)."""
    fx, meta = U.load_case(name)
    model = U.build_model(meta, fx=fx)
    assert type(model).__name__ == meta["method"]
    assert list(model.state_dict().keys()) == list(meta["state_sha"].keys())
    assert sum(p.numel() for p in model.parameters()) == meta["n_params"]
    assert model.to(torch.device("cpu")) is model          # train.py:45 relies on .to() returning the module


def test_library_exports_every_declared_symbol():
    import mlhot
    mlhot.build_product()
    lib = ctypes.CDLL(mlhot.PRODUCT_SO)
    header = open(os.path.join(ROOT, "include", "mlhot.h")).read()
    names = set(re.findall(r"\b(mlhot_[a-z0-9_]+)\s*\(", header))
    assert len(names) >= 20
    for n in sorted(names):
        assert hasattr(lib, n), f"libmlhot.so does not export {n}"
    assert mlhot.lib().c.mlhot_version() == mlhot.binding.ABI_VERSION == 7


def test_cpu_tensors_fail_loudly():
    """No CPU fallback: the plugin refuses CPU tensors instead of silently running something else."""
    from mlhot.binding import MlhotError
    fx, meta = U.load_case("s_cnp_shapenet1d_max")
    model = U.build_model(meta)
    cx, qx, cy, _ = U.case_inputs(meta)
    with pytest.raises(MlhotError):
        model(cx, cy, qx)


def test_bad_agg_mode_raises_typeerror():
    """CNPShapeNet1D.py:127-128 / ANPShapeNet1D.py:145-146."""
    fx, meta = U.load_case("s_cnp_shapenet1d_max")
    meta = dict(meta, cfg=dict(meta["cfg"], agg_mode="attention"))
    cfg = U.case_config(meta)
    model = importlib.import_module("networks.CNPShapeNet1D").CNPShapeNet1D(cfg)
    cx, qx, cy, _ = U.case_inputs(dict(meta, input_sha=U.load_case("s_cnp_shapenet1d_max")[1]["input_sha"]))
    with pytest.raises(TypeError):
        model(cx, cy, qx)


IN_SCOPE = {"ANP", "ANPDistractor", "ANPMR", "ANPMRShapeNet1D", "ANPMRShapeNet3D", "ANPShapeNet1D", "ANPVanillaPascal1D",
            "CNPDistractor", "CNPMR", "CNPMRShapeNet1D", "CNPShapeNet1D", "CNPVanillaPascal1D", "CondNeuralProcess",
            "FCLANP", "FCLCNPDistractor", "FCLCNPShapeNet1D"}
OUT_OF_SCOPE = {"MAMLMR", "MAMLMRShapeNet1D", "MAMLShapeNet1D", "MMAMLShapeNet1D", "VanillaMAML",
                "SingleTaskDistractor", "SingleTaskShapeNet1D", "SingleTaskShapeNet3D"}


def test_config_surface(tmp_path, monkeypatch):
    """configs/config.py:33-109: required / optional keys, derived img_size / input_dim / output_dim."""
    from configs.config import Config
    monkeypatch.chdir(tmp_path)
    for path in sorted(glob.glob(os.path.join(ROOT, "what-matters-for-meta-learning_amd", "cfg", "train", "*.yaml"))):
        cfg = Config(path)
        assert (cfg.img_size == [128, 128, 1] and cfg.dim_w == 64) or (cfg.task == "shapenet_3d" and cfg.img_size == [64, 64, 4] and cfg.beta == 1e-7)
        assert os.path.isdir(os.path.join(cfg.save_path, "models"))
        mod = importlib.import_module(f"networks.{cfg.method}")
        getattr(mod, cfg.method)(cfg)


@pytest.mark.skipif(not os.path.isdir("/root/reference/cfg"), reason="reference checkout only exists in the build container")
def test_every_reference_yaml_parses(tmp_path, monkeypatch):
    from configs.config import Config
    monkeypatch.chdir(tmp_path)
    files = sorted(glob.glob("/root/reference/cfg/**/*.yaml", recursive=True))
    assert len(files) == 59
    constructed = set()
    for path in files:
        with open(path, "rb") as f:
            raw = yaml.safe_load(f)
        raw["device"] = "cpu"
        cfg = Config()
        cfg.set_init_values(raw, side_effects=False)
        assert cfg.method == raw["method"]
        # train.py:41-45: the plugin lookup.  In-scope methods construct; the others fail loudly and clearly.
        cls = getattr(importlib.import_module(f"networks.{cfg.method}"), cfg.method)
        if cfg.method in OUT_OF_SCOPE:
            with pytest.raises(NotImplementedError, match="not part of the MI355X hot-path build"):
                cls(cfg)
        else:
            model = cls(cfg)
            assert model.to("cpu") is model and len(list(model.parameters())) > 0
            constructed.add(cfg.method)
    assert constructed == IN_SCOPE


def test_staged_eps_reproduces_the_lazy_draws():
    """networks/bbb/eps.py: pre-drawing the recorded shape sequence gives the numbers the reference's lazy per-layer draws
    (bbb/BBBConv.py:88-95) would have produced from the same generator state - incl. tensors below torch's 16-element
    vectorisation threshold and odd sizes."""
    import torch
    from networks.bbb import eps
    shapes = [(64, 3, 5, 5), (64,), (7,), (1,), (3, 5), (64, 64, 3, 3), (15,), (16,), (17,), (256, 100), (2,)]
    st = eps.StagedEps("cpu")
    torch.manual_seed(7)
    with st.recording():
        lazy = [eps.draw(s, "cpu") for s in shapes] + [eps.draw(s, "cpu") for s in shapes]
    assert st.shapes == shapes + shapes
    for _ in range(2):                     # two consecutive steps: the staging buffers alternate
        torch.manual_seed(7)
        st.stage()
        with st.active():
            staged = [eps.draw(s, "cpu") for s in shapes] + [eps.draw(s, "cpu") for s in shapes]
            with pytest.raises(RuntimeError):
                eps.draw((3,), "cpu")      # more draws than recorded
        for a, b in zip(lazy, staged):
            assert a.shape == b.shape and torch.equal(a, b)
    st.stage()
    with st.active(), pytest.raises(RuntimeError):
        eps.draw((5,), "cpu")              # not the recorded shape
    assert eps._active is None and eps._recorder is None


def test_mt19937_advance_leaves_the_state_the_draws_would():
    """mlhot_mt19937_advance (host only, include/mlhot.h): the engine after n calls without the outputs == torch's CPU generator after
    drawing n 32-bit outputs, from a fresh seed and from mid-block positions, across block boundaries."""
    import numpy as np
    import torch
    import mlhot
    from mlhot import rng
    L = mlhot.lib()
    g = torch.Generator()
    g.manual_seed(20260)
    for n in (0, 1, 5, 622, 623, 624, 625, 1000, 3 * 624, 3 * 624 + 1, 100003, 896016):
        before = g.get_state()
        engine = rng._unpack(before).copy()
        L.mt19937_advance(engine, n)
        if n:
            torch.empty(n).uniform_(generator=g)          # a float uniform is ONE engine call
        assert torch.equal(rng._pack(before, engine), g.get_state()), n
    with pytest.raises(mlhot.MlhotError):
        L.mt19937_advance(np.zeros(10, dtype=np.uint32), 1)


def test_host_f32_to_u8_exact_checks_every_element():
    """mlhot_host_f32_to_u8_exact (host only, ABI 6): the loaders' `u8.astype(float32) / 255.0` (dataset/shapenet_1d.py:189-190) inverted
    with every element checked - all 256 byte values round-trip, and a value one ulp off, a NaN, an infinity, a negative or > 1
    value each count as inexact (the caller then ships the fp32 batch unchanged)."""
    import numpy as np
    import mlhot
    L = mlhot.lib()
    u8 = np.concatenate([np.arange(256, dtype=np.uint8), np.random.RandomState(3).randint(0, 256, size=100003).astype(np.uint8)])
    x = u8.astype(np.float32) / 255.0
    for n in (0, 1, 7, 256, 32768, 32769, x.size):              # block and vector remainders
        dst = np.full(x.size, 77, dtype=np.uint8)
        assert L.host_f32_to_u8_exact(x.ctypes.data, dst.ctypes.data, n) == 0
        assert np.array_equal(dst[:n], u8[:n]) and (dst[n:] == 77).all()
    for bad_value in (np.nextafter(x[300], np.float32(2)), np.nextafter(x[300], np.float32(-2)), np.float32("nan"), np.float32("inf"),
                      np.float32(-0.25), np.float32(1.5), np.float32(0.5)):
        y = x.copy()
        y[300] = bad_value
        y[70000] = bad_value
        assert L.host_f32_to_u8_exact(y.ctypes.data, dst.ctypes.data, y.size) == 2, bad_value
    # several native threads: the same bytes and counts (pieces of >= 64 K elements; a short input stays on one thread)
    big = np.tile(u8, 12)
    xb = big.astype(np.float32) / 255.0
    xb[5], xb[xb.size - 3], xb[xb.size // 2] = 0.3, -1.0, np.float32("nan")
    for thr in (1, 2, 3, 8, 64):
        db = np.zeros_like(big)
        assert L.host_f32_to_u8_exact(xb.ctypes.data, db.ctypes.data, xb.size, threads=thr) == 3
        ok = np.ones(big.size, dtype=bool)
        ok[[5, big.size - 3, big.size // 2]] = False
        assert np.array_equal(db[ok], big[ok])
    assert L.host_f32_to_u8_exact(x.ctypes.data, dst.ctypes.data, 1000, threads=8) == 0
    # another divisor (a loader that scales by 1 / 256 is exact too; by 1 / 100 is not for most bytes)
    x256 = u8.astype(np.float32) / np.float32(256.0)
    assert L.host_f32_to_u8_exact(x256.ctypes.data, dst.ctypes.data, x256.size, div=256.0) == 0 and np.array_equal(dst, u8)
    assert L.host_f32_to_u8_exact(x.ctypes.data, dst.ctypes.data, x.size, div=100.0) > 1000
    with pytest.raises(mlhot.MlhotError):
        L.host_f32_to_u8_exact(0, dst.ctypes.data, 4)


def test_mt19937_advance_from_random_offsets_and_lengths():
    """The same statement from 200 random (offset, length) pairs: the engine is first moved to a random position inside / across
    blocks by real draws, then advanced by a random count - lengths from 1 to a few blocks and a few long ones."""
    import numpy as np
    import torch
    import mlhot
    from mlhot import rng
    L = mlhot.lib()
    r = np.random.RandomState(7)
    g = torch.Generator()
    for i in range(200):
        g.manual_seed(int(r.randint(0, 2 ** 31)))
        off = int(r.randint(0, 3 * 624))
        n = int(r.randint(1, 4 * 624)) if i % 10 else int(r.randint(10 ** 5, 10 ** 6))
        if off:
            torch.empty(off).uniform_(generator=g)
        before = g.get_state()
        engine = rng._unpack(before).copy()
        L.mt19937_advance(engine, n)
        torch.empty(n).uniform_(generator=g)
        assert torch.equal(rng._pack(before, engine), g.get_state()), (off, n)


def test_eps_thread_default_adapts_to_the_ranks_of_the_node(monkeypatch):
    """min(4, usable cores // (2 x ranks)), at least 1; an explicit MLHOT_EPS_THREADS is capped by cores // ranks."""
    from networks.bbb import eps
    monkeypatch.setattr(eps, "usable_cores", lambda: 128)
    monkeypatch.delenv("MLHOT_EPS_THREADS", raising=False)
    for ranks, want in ((1, 4), (8, 4), (16, 4), (32, 2), (64, 1), (128, 1)):
        monkeypatch.setenv("LOCAL_WORLD_SIZE", str(ranks))
        assert eps.default_threads() == want
    monkeypatch.setattr(eps, "usable_cores", lambda: 8)
    for ranks, want in ((1, 4), (2, 2), (4, 1), (8, 1)):
        monkeypatch.setenv("LOCAL_WORLD_SIZE", str(ranks))
        assert eps.default_threads() == want
    monkeypatch.setenv("MLHOT_EPS_THREADS", "16")
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "2")
    assert eps.default_threads() == 4
    assert eps.StagedEps("cpu").threads == 4 and eps.StagedEps("cpu", threads=3).threads == 3


@pytest.mark.parametrize("threads", [2, 3, 8])
def test_staged_eps_on_several_host_threads_draws_the_same_stream(threads):
    """networks/bbb/eps.py with `threads` > 1: the recorded sequence cut into pieces, every piece drawn by normal_() on its own generator
    positioned by mlhot_mt19937_advance - numbers AND final CPU generator state equal the lazy per-layer draws of the reference
    (bbb/BBBConv.py:88-95), for c5's sequence (2 x 26 tensors, 896 k normals) and for a ragged one (sizes that are not multiples of
    16 or 4: each consumes 16 extra outputs)."""
    import torch
    from networks.bbb import eps
    c5 = []
    for _ in range(2):
        c5 += [(64, 3, 5, 5), (64,)]
        for _ in range(12):
            c5 += [(64, 64, 3, 3), (64,)]
    ragged = [(100003,), (64,), (17,), (333, 7, 11), (64, 64, 3, 3), (50,), (16,), (99999,)] * 2
    for shapes in (c5, ragged):
        st = eps.StagedEps("cpu", threads=threads)
        torch.manual_seed(5)
        with st.recording():
            lazy = [eps.draw(s, "cpu") for s in shapes]
        lazy_state = torch.get_rng_state()
        for _ in range(2):
            torch.manual_seed(5)
            st.stage()
            assert st._pieces is not None and len(st._pieces) == threads
            assert torch.equal(torch.get_rng_state(), lazy_state)
            with st.active():
                for a, s_ in zip(lazy, shapes):
                    assert torch.equal(a, eps.draw(s_, "cpu"))
        torch.manual_seed(5)                               # and through the prefetch thread
        st.prefetch()
        st.stage()
        assert torch.equal(torch.get_rng_state(), lazy_state)
        with st.active():
            assert all(torch.equal(a, eps.draw(s_, "cpu")) for a, s_ in zip(lazy, shapes))
    # from a generator in the MIDDLE of a block, three steps in a row (training's normal case: every step starts where the last ended)
    st = eps.StagedEps("cpu", threads=threads)
    torch.manual_seed(11)
    torch.empty(1001).normal_()
    start = torch.get_rng_state()
    with st.recording():
        lazy = [[eps.draw(s, "cpu") for s in c5] for _ in range(1)]
    lazy += [[torch.empty(s).normal_(0, 1) for s in c5] for _ in range(2)]
    lazy_state = torch.get_rng_state()
    torch.set_rng_state(start)
    for step in range(3):
        st.stage()
        with st.active():
            assert all(torch.equal(a, eps.draw(s_, "cpu")) for a, s_ in zip(lazy[step], c5)), step
    assert torch.equal(torch.get_rng_state(), lazy_state)
    one = eps.StagedEps("cpu", threads=1)
    one.shapes = list(c5)
    one._plan()
    one.draw_host(torch.zeros(one._total))
    assert one._pieces is None


def test_committed_bench_line_honours_the_contract():
    """The newest profiles/r*_final_bench_c3.json is a bench.py line from the MI355X box: the driver's contract fields, the
    roofline and cpu_baseline objects, metric / unit as BASELINE.json names them."""
    import glob
    import json
    newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_final_bench_c3.json")))[-1]
    line = json.load(open(newest))
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["higher_is_better"] is True and line["scaling"] == "weak" and line["data"] == "synthetic"
    assert line["dtype"] == "f32" and line["vs_baseline"] is None and "workload" in line["config"] and "model" not in line["config"]
    assert abs(line["value"] - 16 * 1e3 / line["ms_per_step"]) <= 1e-6 * line["value"]            # 16 tasks per step on one GPU
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] is None or r["traffic"] > 0
    c = line["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == line["unit"] and c["sample"]
    if isinstance(base, dict) and "unit" in base:
        assert base["unit"].split("/")[0].strip().lower()[:4] in line["unit"].lower() or line["unit"].lower()[:4] in base["unit"].lower()


def test_library_staleness_is_decided_by_source_content_not_file_times(tmp_path, monkeypatch):
    """mlhot.build: csrc/libmlhot.so carries a sidecar with the sha256 of the sources it was built from; "stale" compares that with the
    tree's sources.  File times are an accident of how the tree travelled (a `git checkout` of unchanged text once made the GPU box
    rebuild an identical library inside the driver's bench run); they decide only for a library without a sidecar."""
    import os
    import time
    from mlhot import build as B
    so, src = tmp_path / "libmlhot.so", tmp_path / "dep.h"
    monkeypatch.setattr(B, "PRODUCT_SO", str(so))
    monkeypatch.setattr(B, "SIDECAR", str(so) + ".src")
    monkeypatch.setattr(B, "_deps", lambda: [str(src)])
    assert B._product_stale()                                   # no library at all
    src.write_text("int a;\n")
    so.write_bytes(b"\x7fELF")
    B.stamp_product()
    assert not B._product_stale()
    os.utime(src, (time.time() + 100, time.time() + 100))       # "newer" source, same text
    assert not B._product_stale()
    src.write_text("int b;\n")                                  # different text, whatever its time
    os.utime(src, (1, 1))
    assert B._product_stale()
    os.remove(str(so) + ".src")                                 # no sidecar: file times decide
    assert not B._product_stale()
    os.utime(src, (time.time() + 100, time.time() + 100))
    assert B._product_stale()
