"""Command-line driver and checkpoint loader (reference: argprocess.py:5-18, cute_main.py:25-31, 48-132).  CPU: flag
parsing, the CUTE triplet walk, diffusers-layout directory parsing, and the whole object graph built from synthetic
safetensors up to the first GPU touch, which must fail loudly (there is no CPU fallback)."""
import json
import os

import pytest
import torch
from PIL import Image

from diffsim_amd import cli, config as C, loader, synth as S, text as T


def test_flags_keep_reference_names_and_defaults():
    a = cli.arg_parse([])
    # argprocess.py:5-18 defaults
    assert (a.image_size, a.target_block, a.target_layer, a.target_step, a.metric, a.similarity, a.prompt, a.seed) == \
        (512, "up_blocks", 2, 100, "diffsim", "mse", "High quality image", 2333)
    a = cli.arg_parse("--image_path /d --image_size 512 --target_block up_blocks --target_layer 0 --target_step 600 "
                      "--similarity cosine --seed 2334 --metric diffsim".split())          # cute_main.sh:3
    assert a.target_layer == [0] and a.target_step == 600 and a.similarity == "cosine" and a.seed == 2334
    a = cli.arg_parse("--metric diffsim_xl --target_layer 0 1 2".split())
    assert a.target_layer == [0, 1, 2]


def _tree(root, n_cls=2, n_inst=3, n_light=2, n_img=3):
    for c in range(n_cls):
        for i in range(n_inst):
            for l in range(n_light):
                d = os.path.join(root, f"cls{c}", f"inst{i}", f"light{l}")
                os.makedirs(d)
                for k in range(n_img):
                    Image.new("RGB", (8, 8), (c * 40, i * 40, k * 40)).save(os.path.join(d, f"im{k}.png"))


def test_cute_triplet_walk(tmp_path):
    _tree(str(tmp_path))
    t1 = cli.cute_triplets(str(tmp_path), 2334)
    assert t1 == cli.cute_triplets(str(tmp_path), 2334) and t1 != cli.cute_triplets(str(tmp_path), 1)
    # every class: 10 experiments x (os.walk visits the class dir [3 instances] and each instance dir [2 lightings, which
    # have no sub-folders -> skipped]) = 30 triplets per class
    assert len(t1) == 2 * 10 * 3
    for a, b, c, prompt in t1:
        da, db, dc = os.path.dirname(a), os.path.dirname(b), os.path.dirname(c)
        assert da == db and a != b                                  # A, B: two images of one (instance, lighting)
        assert os.path.basename(da) == os.path.basename(dc)         # C: same lighting ...
        assert os.path.dirname(da) != os.path.dirname(dc)           # ... of another instance
        assert os.path.dirname(os.path.dirname(da)) == os.path.dirname(os.path.dirname(dc))    # of the same class
        assert prompt == "The photo of a " + os.path.basename(os.path.dirname(os.path.dirname(da)))
    s_ab, s_ac = torch.tensor([0.9, 0.2, float("nan")]), torch.tensor([0.4, 0.3, 0.1])
    assert cli.cute_counts(s_ab, s_ac, "cosine") == (1, 1)          # cute_main.py:201-205; NaN compares False
    assert cli.cute_counts(s_ab, s_ac, "mse") == (1, 0)


def _write_checkpoint(root):
    from safetensors.torch import save_file
    import dataclasses
    # the text encoder's width is the U-Net's cross_attention_dim, its 77 positions the context length
    ucfg, vcfg, tcfg = C.TINY, C.VAE_TINY, dataclasses.replace(T.CLIP_TINY, hidden_size=C.TINY.cross_attention_dim, projection_dim=0)
    os.makedirs(os.path.join(root, "unet")); os.makedirs(os.path.join(root, "vae")); os.makedirs(os.path.join(root, "text_encoder"))
    save_file(S.make_state_dict(ucfg, seed=0), os.path.join(root, "unet", "diffusion_pytorch_model.safetensors"))
    json.dump({"in_channels": 4, "out_channels": 4, "block_out_channels": list(ucfg.block_out_channels),
               "down_block_types": list(ucfg.down_block_types), "up_block_types": list(ucfg.up_block_types),
               "layers_per_block": 2, "attention_head_dim": ucfg.num_attention_heads, "cross_attention_dim": ucfg.cross_attention_dim,
               "norm_num_groups": 32, "norm_eps": 1e-5, "sample_size": ucfg.sample_size, "use_linear_projection": False},
              open(os.path.join(root, "unet", "config.json"), "w"))
    save_file(S.make_state_dict(vcfg, seed=1), os.path.join(root, "vae", "diffusion_pytorch_model.safetensors"))
    json.dump({"in_channels": 3, "latent_channels": 4, "block_out_channels": list(vcfg.block_out_channels), "layers_per_block": 2,
               "norm_num_groups": 32, "scaling_factor": 0.18215}, open(os.path.join(root, "vae", "config.json"), "w"))
    g = torch.Generator().manual_seed(5)
    te = {"text_model." + k: 0.05 * torch.randn(s, generator=g) for k, s in T.clip_text_param_shapes(tcfg).items() if k != "text_projection.weight"}
    save_file(te, os.path.join(root, "text_encoder", "model.safetensors"))
    json.dump({"vocab_size": tcfg.vocab_size, "hidden_size": tcfg.hidden_size, "intermediate_size": tcfg.intermediate_size,
               "num_hidden_layers": tcfg.num_layers, "num_attention_heads": tcfg.num_heads, "max_position_embeddings": 77,
               "hidden_act": "quick_gelu", "layer_norm_eps": 1e-5, "eos_token_id": 2}, open(os.path.join(root, "text_encoder", "config.json"), "w"))


def test_loader_parses_diffusers_layout(tmp_path):
    _write_checkpoint(str(tmp_path))
    ucfg = loader.unet_config_from_json(loader._json(os.path.join(tmp_path, "unet")), C.SD15)
    assert ucfg == C.UNetConfig(block_out_channels=C.TINY.block_out_channels, num_attention_heads=4, cross_attention_dim=128,
                                sample_size=16, ctx_len=77)
    sd = loader.load_state_dict(os.path.join(tmp_path, "unet"))
    assert {k: tuple(v.shape) for k, v in sd.items()} == C.unet_param_shapes(C.TINY)
    assert loader.vae_config_from_json(loader._json(os.path.join(tmp_path, "vae"))) == C.VAE_TINY
    tc = loader.clip_config_from_json(loader._json(os.path.join(tmp_path, "text_encoder")), T.CLIP_L)
    assert (tc.hidden_size, tc.num_layers, tc.num_heads, tc.vocab_size) == (128, 2, 4, 1000)
    # SDXL's config.json idioms: per-level head counts / transformer depths, text_time conditioning
    xl = loader.unet_config_from_json({"block_out_channels": [320, 640, 1280], "down_block_types": list(C.SDXL.down_block_types),
                                       "up_block_types": list(C.SDXL.up_block_types), "attention_head_dim": [5, 10, 20],
                                       "transformer_layers_per_block": [1, 2, 10], "cross_attention_dim": 2048, "sample_size": 128,
                                       "use_linear_projection": True, "addition_embed_type": "text_time",
                                       "addition_time_embed_dim": 256, "projection_class_embeddings_input_dim": 2816}, C.SDXL)
    assert xl == C.SDXL
    with pytest.raises(FileNotFoundError):
        loader.LazyTokenizer(os.path.join(tmp_path, "tokenizer"))("a cat")


def test_cli_builds_the_object_graph_and_fails_loudly_without_gpu(tmp_path):
    from diffsim_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_gpu_cli.py")
    ck = os.path.join(tmp_path, "ckpt"); os.makedirs(ck)
    _write_checkpoint(ck)
    data = os.path.join(tmp_path, "data"); os.makedirs(data)
    _tree(data, 1, 2, 1, 2)
    argv = ["--metric", "diffsim", "--dataset", "cute", "--model_path", ck, "--image_path", data, "--image_size", "128",
            "--target_block", "up_blocks", "--target_layer", "0", "--target_step", "600", "--similarity", "cosine", "--dtype", "fp32"]
    with pytest.raises(_lib.DsimError, match="no GPU"):
        cli.main(argv)
    for bad in (["--metric", "clip_i"], ["--ip_adapter"], ["--model_path", ""]):
        with pytest.raises(SystemExit):
            cli.main([a for a in argv if a not in ("--metric", "diffsim")] + bad if bad[0] == "--metric" else argv + bad)


def test_sref_triplet_walk(tmp_path):
    """--dataset sref (BASELINE config 4's "Sref style pairs"): the style_main.py:48-76 experiment sampler -- every folder
    with >= 2 images is a style; per experiment random.sample(styles, 2), random.sample(images of the first, 2),
    random.choice(images of the second), seeded with --seed.  Checked against that call sequence stated directly."""
    import random
    from PIL import Image
    root = str(tmp_path)
    for s in range(5):
        d = os.path.join(root, "group%d" % (s % 2), "style%d" % s)
        os.makedirs(d)
        for i in range(3 if s else 1):                 # style0 has one image only: not a style
            Image.new("RGB", (8, 8), (s * 40, i * 60, 7)).save(os.path.join(d, "im%d.png" % i))
    t = cli.sref_triplets(root, 2333, "High quality image", 50)
    assert len(t) == 50 and t == cli.sref_triplets(root, 2333, "High quality image", 50)
    assert t != cli.sref_triplets(root, 1, "High quality image", 50)
    for a, b, c, prompt in t:
        assert os.path.dirname(a) == os.path.dirname(b) and a != b and os.path.dirname(c) != os.path.dirname(a)
        assert "style0" not in a and "style0" not in c and prompt == "High quality image"
    # the reference's sequence of random calls over the same os.walk order
    random.seed(2333)
    styles = {}
    for r, dirs, _ in os.walk(root):
        for d in dirs:
            full = os.path.join(r, d)
            ims = [os.path.join(full, f) for f in os.listdir(full) if f.endswith((".png", ".jpg", ".jpeg"))]
            if len(ims) >= 2:
                styles[full] = ims
    names = list(styles)
    for k in range(50):
        da, dc = random.sample(names, 2)
        a, b = random.sample(styles[da], 2)
        assert t[k][:3] == (a, b, random.choice(styles[dc]))
    args = cli.arg_parse(["--dataset", "sref", "--metric", "diffsim_xl", "--target_layer", "0", "0", "0"])
    assert args.experiments == 2000 and args.target_layer == [0, 0, 0]
