"""CPU-side checks: the C-ABI library loads and exports every symbol the header declares, host
logic (image preprocessing, scheduler table, generator order, flag handling) matches the golden
fixtures captured from the reference.  No compute calls -- there is no GPU here."""
import json
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def lib():
    from diffsim_amd import _lib, build
    build.build()
    return _lib.lib()


def test_abi_exports_every_declared_symbol(lib):
    from diffsim_amd import _lib
    header = open(os.path.join(ROOT, "include", "diffsim_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(dsim_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name)
    hv = int(re.search(r"#define\s+DSIM_ABI_VERSION\s+(\d+)", header).group(1))
    assert lib.dsim_version() == hv == _lib.ABI_VERSION == 7            # header, library and bindings agree
    # ... and so does the driver's build check (it broke once on a literal)
    assert "_lib.ABI_VERSION" in open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert lib.dsim_strerror(0) == b"ok"
    assert b"workspace" in lib.dsim_strerror(-3)


def test_engine_fails_loudly_without_gpu():
    from diffsim_amd import _lib, config as C, synth as S
    from diffsim_amd.engine import UNetEngine
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.DsimError):
        UNetEngine(C.TINY, {}, torch.float32)


def test_g1_process_image_matches_reference():
    from diffsim_amd.image import load_image, process_image
    g = np.load(os.path.join(G, "g1_process_image.npz"))
    for name in "abcd":
        for size in (64, 128):
            out = process_image(load_image(os.path.join(G, f"g1_img_{name}.png")), size)
            assert out.shape == (1, 3, size, size) and out.dtype == torch.float32
            assert np.array_equal(out.numpy(), g[f"{name}_{size}"])          # bit-exact
            assert float(out.min()) >= -1.0 and float(out.max()) <= 1.0


def test_g2_generator_order():
    from diffsim_amd.diffsim import get_generator
    g2 = json.load(open(os.path.join(G, "g2_generator.json")))
    gen = get_generator(g2["seed"], "cpu")
    for first in g2["first8"]:
        d = torch.randn(tuple(g2["shape"]), generator=gen)
        assert d.flatten()[:8].tolist() == first
    gens = get_generator([1, 2], "cpu")
    assert isinstance(gens, list) and len(gens) == 2
    assert get_generator(None) is None


def test_g7_scheduler_table():
    from diffsim_amd import scheduler as sch
    g = json.load(open(os.path.join(G, "g7_sched.json")))
    ts = sch.pndm_timesteps()
    assert len(ts) == g["pndm_len"] and ts[:4].tolist() == g["pndm_head"] and ts[-3:].tolist() == g["pndm_tail"]
    for i, t in g["pndm_idx"].items():
        assert sch.timestep_from_index(int(i)) == t
    sa, sb = sch.noise_coefficients(401)
    assert abs(sa - g["sqrt_abar_401"]) < 1e-7 and abs(sb - g["sqrt_1m_abar_401"]) < 1e-7
    with pytest.raises(IndexError):
        sch.timestep_from_index(0)
    with pytest.raises(IndexError):
        sch.timestep_from_index(1001)


def test_target_layer_coercion():
    from diffsim_amd.diffsim import _norm_layer
    assert _norm_layer([5]) == 0          # diffsim/diffsim.py:99-100
    assert _norm_layer([0]) == 0
    assert _norm_layer(2) == 2
    with pytest.raises(TypeError):
        _norm_layer([1, 1])


def test_arg_parse_flags():
    from diffsim_amd.cli import arg_parse
    a = arg_parse(["--image_path", "x", "--target_block", "up_blocks", "--target_layer", "0", "--target_step", "600",
                   "--similarity", "cosine", "--metric", "diffsim", "--seed", "2334"])
    assert a.target_layer == [0] and a.target_step == 600 and a.image_size == 512
    d = arg_parse([])
    assert d.similarity == "mse" and d.seed == 2333 and d.prompt == "High quality image"


def test_synthetic_inputs_are_deterministic():
    from diffsim_amd import config as C, synth as S
    a1, b1 = S.make_pair_latents(C.TINY, 7)
    a2, b2 = S.make_pair_latents(C.TINY, 7)
    assert torch.equal(a1, a2) and torch.equal(b1, b2) and not torch.equal(a1, b1)
    sd1 = S.make_state_dict(C.TINY, 0, keys=["conv_in.weight", "mid_block.resnets.0.norm1.weight"])
    sd2 = S.make_state_dict(C.TINY, 0)
    for k in sd1:
        assert torch.equal(sd1[k], sd2[k])


def test_nights_decision_rule():
    from diffsim_amd import harness as Hn
    sl = torch.tensor([0.9, 0.2, 0.5]); sr = torch.tensor([0.1, 0.8, 0.5])
    assert Hn.nights_decisions(sl, sr, "cosine").tolist() == [1, 0, 0]      # night_main.py:160-161
    assert Hn.nights_decisions(sl, sr, "mse").tolist() == [0, 1, 0]         # night_main.py:158-159
    assert abs(Hn.nights_accuracy(sl, sr, [1, 0, 1], "cosine") - 200.0 / 3) < 1e-4
    assert Hn.cute_accuracy(sl, sr) == pytest.approx(100.0 / 3)


def test_bench_kernel_names_match_the_committed_rocprof_summaries():
    """bench.py maps its per-launch profile families to the kernel symbols rocprofv3 prints, so that
    roofline.traffic / mfma_util_pmc come from the committed PMC summaries: every MFMA kernel of the newest
    committed bench line must be found in the same snapshot's kernel-trace and PMC files."""
    import csv
    import glob
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    benches = sorted(glob.glob(os.path.join(root, "profiles", "r*_bench.json")))
    tag = os.path.basename(benches[-1]).split("_")[0]
    line = json.load(open(benches[-1]))
    mf = os.path.join(root, "profiles", f"{tag}_pmc_mfma.json")
    if not os.path.exists(mf):
        pytest.skip("snapshot without an MFMA pass")
    mfma = json.load(open(mf))
    hbm = json.load(open(os.path.join(root, "profiles", f"{tag}_pmc_hbm.json")))
    stats = {r["kernel"] for r in csv.DictReader(open(os.path.join(root, "profiles", f"{tag}_kernel_stats.csv")))}
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    fams = [k for k in line["kernel_breakdown_ms_per_step"] if k.startswith(("gemm_", "attention_"))]
    assert fams
    for fam in fams:
        name = bench.rocprof_name(fam)
        assert name in stats, (fam, name)
        assert name in mfma and name in hbm, (fam, name)
    assert line["roofline"]["kernel"] in stats
    assert bench.pmc_traffic(line["roofline"]["kernel"]) and bench.pmc_mfma_util(line["roofline"]["kernel"])
    # ... and the VAE families of the same snapshot's --pixels-in line (among them the 512 x 128 conv tiles, 8 x 1 waves) in its kernel trace
    px, pstats = (os.path.join(root, "profiles", f"{tag}_{n}") for n in ("bench_pixels_in.json", "pixels_in_kernel_stats.csv"))
    if os.path.exists(px) and os.path.exists(pstats):
        pk = {r["kernel"] for r in csv.DictReader(open(pstats))}
        vfams = [k for k in json.load(open(px))["kernel_breakdown_ms_per_step"] if k.startswith("vae_gemm_")]
        assert vfams
        for fam in vfams:
            assert bench.rocprof_name(fam) in pk, (fam, bench.rocprof_name(fam))


def test_bench_pmc_fields_come_from_the_pass_of_the_same_model():
    """roofline.traffic / mfma_util_pmc of a bench line are read from the newest committed PMC summary OF THAT MODEL:
    the SDXL and DiT passes list the same kernel symbols as the SD1.5 pass (on other shapes), so the lookup goes by exact
    file-name class, never by a glob that the lexically last file wins."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.pmc_file_class("sd15", "pmc_hbm").match("r03d_pmc_hbm.json")
    assert not bench.pmc_file_class("sd15", "pmc_hbm").match("r03d_sdxl_pmc_hbm.json")
    assert not bench.pmc_file_class("sd15", "pmc_hbm").match("r03d_dit_pmc_hbm.json")
    assert bench.pmc_file_class("sdxl", "pmc_hbm").match("r03d_sdxl_pmc_hbm.json")
    assert not bench.pmc_file_class("dit", "pmc_mfma").match("r03d_pmc_mfma.json")
    # r10a sorts after r9z: round number first, then the letter
    for model in ("sd15", "sdxl", "dit"):
        bench.PMC_MODEL = model
        for suffix, field, fn in (("pmc_hbm", "hbm_bytes_per_launch", bench.pmc_traffic), ("pmc_mfma", "mfma_util", bench.pmc_mfma_util)):
            f = bench.newest_pmc_file(model, suffix)
            assert f is not None, (model, suffix)
            base = os.path.basename(f)
            assert ("_sdxl_" in base) == (model == "sdxl") and ("_dit_" in base) == (model == "dit"), base
            d = json.load(open(f))
            for kernel in d:
                f2 = bench.newest_pmc_file(model, suffix, kernel)
                assert fn(kernel) == json.load(open(f2))[kernel][field]
    # the headline line of the newest committed snapshot: its traffic is the number of the SD1.5 pass it names as its source, and that
    # source is an SD1.5 file (never the SDXL / DiT pass that lists the same kernel symbol)
    bench.PMC_MODEL = "sd15"
    f = newest = bench.newest_pmc_file("sd15", "pmc_hbm")
    tag = os.path.basename(newest).split("_")[0]
    bl = os.path.join(root, "profiles", f"{tag}_bench.json")
    if os.path.exists(bl) and int(re.match(r"r(\d+)", tag).group(1)) >= 4:      # lines before round 4 carry the old lookup
        line = json.load(open(bl))
        src = line["roofline"]["pmc_source"]["traffic"]
        assert bench.pmc_file_class("sd15", "pmc_hbm").match(src), src
        assert line["roofline"]["traffic"] == json.load(open(os.path.join(root, "profiles", src)))[line["roofline"]["kernel"]]["hbm_bytes_per_launch"]
        assert src == os.path.basename(f)                 # collect_round.sh points a snapshot's line at the snapshot's own passes


def test_public_header_is_plain_c99(tmp_path):
    """The drop-in boundary is a C ABI: include/diffsim_amd.h must compile as C99 with no C++ or torch types."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "abi.c"
    src.write_text('#include "diffsim_amd.h"\nint main(void) { dsim_unet_cfg c; (void)c; return DSIM_ABI_VERSION == 0; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"),
                        "-fsyntax-only", str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_loader_reads_sharded_checkpoints_and_refuses_ambiguity(tmp_path):
    """loader.load_state_dict: a *.safetensors.index.json checkpoint is merged from all its shards; several unindexed
    files, a missing shard, repeated keys and an unknown compute dtype raise instead of loading a part; of two indexes the
    fp32 one is read."""
    import json as _json
    import torch
    from safetensors.torch import save_file
    from diffsim_amd import loader
    d = tmp_path / "unet"
    d.mkdir()
    save_file({"a.weight": torch.ones(2, 2)}, str(d / "diffusion_pytorch_model-00001-of-00002.safetensors"))
    save_file({"b.weight": torch.zeros(3)}, str(d / "diffusion_pytorch_model-00002-of-00002.safetensors"))
    with pytest.raises(FileNotFoundError, match="several"):
        loader.load_state_dict(str(d))
    idx = {"weight_map": {"a.weight": "diffusion_pytorch_model-00001-of-00002.safetensors",
                          "b.weight": "diffusion_pytorch_model-00002-of-00002.safetensors"}}
    (d / "diffusion_pytorch_model.safetensors.index.json").write_text(_json.dumps(idx))
    sd = loader.load_state_dict(str(d))
    assert sorted(sd) == ["a.weight", "b.weight"] and sd["a.weight"].shape == (2, 2)
    (d / "diffusion_pytorch_model-00002-of-00002.safetensors").unlink()
    with pytest.raises(FileNotFoundError, match="not there"):
        loader.load_state_dict(str(d))
    with pytest.raises(ValueError, match="bf16, fp16 or fp32"):
        loader._torch_dtype("fp8")
    assert loader._torch_dtype("fp16") == torch.float16
    # both indexes present: the fp32 one wins, as for single files (a sorted glob would take the fp16 variant)
    d2 = tmp_path / "unet2"
    d2.mkdir()
    save_file({"a.weight": torch.ones(2)}, str(d2 / "full.safetensors"))
    save_file({"a.weight": torch.zeros(2)}, str(d2 / "half.safetensors"))
    (d2 / "diffusion_pytorch_model.safetensors.index.json").write_text(_json.dumps({"weight_map": {"a.weight": "full.safetensors"}}))
    (d2 / "diffusion_pytorch_model.fp16.safetensors.index.json").write_text(_json.dumps({"weight_map": {"a.weight": "half.safetensors"}}))
    assert float(loader.load_state_dict(str(d2))["a.weight"][0]) == 1.0
    assert loader._json(str(tmp_path / "nowhere")) == {}


def test_decode_pool_processes_match_the_in_process_path(tmp_path):
    """image.DecodePool: worker processes (pipes, PIL + numpy only) return exactly resize_u8's pixels, in order, for threads
    and for processes; a bad path raises in the caller with the worker's message."""
    from PIL import Image
    from diffsim_amd.image import DecodePool, load_image, resize_u8
    rng = np.random.default_rng(0)
    paths = []
    for i in range(5):
        p = str(tmp_path / f"im{i}.png")
        Image.fromarray(rng.integers(0, 255, (40 + i, 50, 3), dtype=np.uint8)).save(p)
        paths.append(p)
    want = torch.cat([resize_u8(load_image(p), 32) for p in paths])
    for procs in (0, 2):
        pool = DecodePool(procs=procs, threads=2)
        got = DecodePool.gather(pool.submit(paths, 32))
        assert got.dtype == torch.uint8 and torch.equal(got, want)
        assert torch.equal(DecodePool.gather(pool.submit(paths[::-1], 32)), want.flip(0))
        if procs:
            with pytest.raises(RuntimeError, match="decode worker failed"):
                DecodePool.gather(pool.submit([str(tmp_path / "missing.png")], 32))
            assert torch.equal(DecodePool.gather(pool.submit(paths[:2], 32)), want[:2])       # the workers survive a failure
        pool.shutdown()


def test_shared_synthetic_weights_are_written_once_and_mapped(tmp_path):
    """bench.py's N-rank runs: rank 0 writes the synthetic state dict once, the other ranks map that file -- same tensors bit for
    bit, and a rank whose file never appears fails with a message instead of hanging."""
    import torch
    from diffsim_amd import config as C, synth as S
    keys = [k for k in C.unet_param_shapes(C.TINY) if k.startswith(("conv_in", "down_blocks.0"))]
    a = S.make_state_dict_shared(C.TINY, 3, keys, rank=0, world=8, cache_dir=str(tmp_path))
    files = list(tmp_path.iterdir())
    assert len(files) == 1 and files[0].suffix == ".safetensors"
    b = S.make_state_dict_shared(C.TINY, 3, keys, rank=5, world=8, cache_dir=str(tmp_path))
    ref = S.make_state_dict(C.TINY, 3, keys)
    assert set(a) == set(b) == set(ref) and all(torch.equal(a[k], ref[k]) and torch.equal(b[k], ref[k]) for k in ref)
    import pytest
    with pytest.raises(TimeoutError):
        S.make_state_dict_shared(C.TINY, 4, keys, rank=1, world=8, cache_dir=str(tmp_path), timeout_s=0.5)


def test_attn160_inline_asm_loads_are_not_touched_before_their_wait():
    """csrc/attn160.hip reads the parked self-attention output back with inline-asm buffer loads hipcc does not track; the audit
    compiles the file (bf16 and fp16) and checks that no instruction between those loads and their wait reads or writes the
    destination registers, and that the kernel has no spills (tools/audit_attn160.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "audit_attn160.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
