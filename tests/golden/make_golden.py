#!/usr/bin/env python3
"""Generate the committed golden fixtures by running the REFERENCE's own Python.

Runs only in the build container (needs /root/reference); nothing here travels to the GPU
box except the small data files it writes next to itself.  The reference cannot be imported
as-is (diffusers/torchvision/seaborn are not installed), so -- following SURVEY.md
Appendix D -- name-only stubs are registered in ``sys.modules`` for the third-party
packages.  The stub classes that the reference *calls into* (the ``StableDiffusionPipeline``
parent, ``retrieve_timesteps``, ``randn_tensor``, the PNDM scheduler) restate the published
diffusers 0.29.2 semantics of SURVEY.md Appendix A; every line of the reference's own files
(``process_image``, ``get_generator``, ``DiffSim.diffsim``, the pre-hook,
``hacked_AttnProcessor2_0``, ``DiffSimPipeline.step``, the ``hacked_*_forward`` block
control flow) executes unmodified.

Fixtures written:
  g1_img_*.png, g1_process_image.npz   process_image on two non-square PNGs
  g2_generator.json                    draw order of the single per-call generator
  g3_attn_qkv.npz                      hacked_AttnProcessor2_0 q/k/v on seeded input
  g4_tail.npz                          DiffSim.diffsim score tail on seeded q/k/v (cos + mse)
  g5_e2e_tiny.npz                      whole DiffSim.diffsim + DiffSimPipeline.step driven on
                                       the oracle's TINY U-Net: q/k/v of both images + scores
  g6_blocks.npz                        hacked Down/Mid/Up block forwards vs plain forwards
  g7_sched.json                        PNDM table facts (restated) + DiT timestep_map (reference)
"""
import copy
import hashlib
import json
import os
import sys
import types

import numpy as np
import torch
from PIL import Image, ImageOps

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)

from oracle import cpu_ref as R            # noqa: E402
from diffsim_amd import config as C        # noqa: E402
from diffsim_amd import synth as S         # noqa: E402


# ----------------------------------------------------------------------------------------------
# stubs for the third-party namespace the reference imports (names only, plus the few
# diffusers behaviours DiffSimPipeline.step calls -- restated from SURVEY.md Appendix A)
# ----------------------------------------------------------------------------------------------
class _PNDM:
    """PNDMScheduler(SD1.5 scheduler_config): skip_prk_steps, steps_offset 1, scaled_linear."""
    order = 1
    init_noise_sigma = 1.0

    def __init__(self):
        self.alphas_cumprod = R.alphas_cumprod()
        self.timesteps = None

    def set_timesteps(self, n, device=None):
        self.timesteps = torch.from_numpy(R.pndm_timesteps(n))

    def add_noise(self, x, noise, timesteps):
        ac = self.alphas_cumprod.to(x.dtype)
        a = ac[timesteps] ** 0.5
        b = (1 - ac[timesteps]) ** 0.5
        while a.ndim < x.ndim:
            a = a.unsqueeze(-1)
            b = b.unsqueeze(-1)
        return a * x + b * noise

    def scale_model_input(self, x, t):
        return x


class _StableDiffusionPipeline:
    def __init__(self, vae, text_encoder, tokenizer, unet, scheduler, safety_checker,
                 feature_extractor, image_encoder=None, requires_safety_checker=True):
        self.vae, self.text_encoder, self.tokenizer = vae, text_encoder, tokenizer
        self.unet, self.scheduler = unet, scheduler
        self.vae_scale_factor = 8

    # properties the reference's step() reads
    guidance_scale = property(lambda s: s._guidance_scale)
    clip_skip = property(lambda s: s._clip_skip)
    cross_attention_kwargs = property(lambda s: s._cross_attention_kwargs)
    do_classifier_free_guidance = property(
        lambda s: s._guidance_scale > 1 and s.unet.config.time_cond_proj_dim is None)
    _execution_device = property(lambda s: torch.device("cpu"))

    def check_inputs(self, *a, **k):
        pass

    def encode_prompt(self, prompt, device, n, do_cfg, negative_prompt, prompt_embeds=None,
                      negative_prompt_embeds=None, lora_scale=None, clip_skip=None):
        ctx = self.text_encoder(prompt)          # (2,L,D): [uncond, cond]
        return ctx[1:2], ctx[0:1]

    def prepare_latents(self, b, c, h, w, dtype, device, generator, latents=None):
        return latents.to(device) * self.scheduler.init_noise_sigma

    def prepare_extra_step_kwargs(self, generator, eta):
        return {}


def _retrieve_timesteps(scheduler, n, device=None, timesteps=None, sigmas=None, **kw):
    scheduler.set_timesteps(n, device=device)
    return scheduler.timesteps, n


def _randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    return torch.randn(tuple(shape), generator=generator, dtype=dtype)


def _load_image(path):
    im = Image.open(path)
    im = ImageOps.exif_transpose(im)
    return im.convert("RGB")


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Dummy:
        def __init__(self, *a, **k):
            pass

    mod("diffusers", StableDiffusionPipeline=_StableDiffusionPipeline,
        StableDiffusionXLPipeline=_Dummy, DDIMScheduler=_Dummy, AutoencoderKL=_Dummy)
    mod("diffusers.utils", load_image=_load_image,
        PIL_INTERPOLATION={"lanczos": Image.Resampling.LANCZOS}, deprecate=lambda *a, **k: None,
        is_torch_version=lambda *a: True, USE_PEFT_BACKEND=False,
        logging=types.SimpleNamespace(get_logger=lambda *a: None))
    mod("diffusers.utils.torch_utils", apply_freeu=lambda *a, **k: None, randn_tensor=_randn_tensor)
    mod("diffusers.models", AutoencoderKL=_Dummy, ImageProjection=_Dummy, UNet2DConditionModel=_Dummy)
    mod("diffusers.models.transformers")
    mod("diffusers.models.transformers.transformer_2d", Transformer2DModel=_Dummy,
        Transformer2DModelOutput=lambda sample: types.SimpleNamespace(sample=sample))
    mod("diffusers.models.attention", BasicTransformerBlock=_Dummy, _chunked_feed_forward=None)
    mod("diffusers.models.attention_processor", Attention=_Dummy, IPAdapterAttnProcessor=_Dummy,
        IPAdapterAttnProcessor2_0=_Dummy)
    mod("diffusers.image_processor", IPAdapterMaskProcessor=_Dummy, PipelineImageInput=_Dummy,
        VaeImageProcessor=_Dummy)
    mod("diffusers.callbacks", MultiPipelineCallbacks=_Dummy, PipelineCallback=_Dummy)
    mod("diffusers.schedulers", KarrasDiffusionSchedulers=_Dummy)
    mod("diffusers.pipelines")
    mod("diffusers.pipelines.stable_diffusion")
    mod("diffusers.pipelines.stable_diffusion.safety_checker", StableDiffusionSafetyChecker=_Dummy)
    mod("diffusers.pipelines.stable_diffusion.pipeline_stable_diffusion",
        retrieve_timesteps=_retrieve_timesteps, rescale_noise_cfg=None)
    mod("torchvision", transforms=types.ModuleType("torchvision.transforms"))
    mod("torchvision.transforms")
    mod("seaborn")
    mod("matplotlib", pyplot=types.ModuleType("matplotlib.pyplot"))
    mod("matplotlib.pyplot")
    mod("transformers", CLIPImageProcessor=_Dummy, CLIPTextModel=_Dummy, CLIPTokenizer=_Dummy,
        CLIPVisionModelWithProjection=_Dummy)
    sys.path.insert(0, REF)


# ----------------------------------------------------------------------------------------------
# fake (but deterministic, generator-consuming) VAE; the real one is a "next" row (section 8f)
# ----------------------------------------------------------------------------------------------
from tests._fakes import FakeVAE          # noqa: E402  (shared with the parity tests)


class UNetAdapter(torch.nn.Module):
    """Presents the oracle U-Net with diffusers' call signature."""

    def __init__(self, unet: R.UNet2DConditionModel):
        super().__init__()
        self.inner = unet
        self.config = types.SimpleNamespace(in_channels=unet.cfg.in_channels,
                                            sample_size=unet.cfg.sample_size,
                                            time_cond_proj_dim=None)
        self.down_blocks, self.mid_block, self.up_blocks = unet.down_blocks, unet.mid_block, unet.up_blocks

    def forward(self, sample, t, encoder_hidden_states=None, **kw):
        return (self.inner(sample, int(t), encoder_hidden_states),)


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    install_stubs()
    import diffsim.diffsim as ref_ds
    import diffsim.hacked_attn as ref_attn
    import diffsim.hacked_modules as ref_mod
    from diffsim.diffsim_pipeline import DiffSimPipeline

    # ---- G1: process_image -----------------------------------------------------------------
    rs = np.random.RandomState(7)
    g1 = {}
    for name, (h, w) in (("a", (96, 64)), ("b", (50, 80)), ("c", (128, 128)), ("d", (160, 120))):
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([(yy * 255 // h), (xx * 255 // w), ((yy + xx) * 255 // (h + w))], -1)
        img = (0.6 * base + 0.4 * rs.randint(0, 256, (h, w, 3))).astype(np.uint8)
        path = os.path.join(HERE, f"g1_img_{name}.png")
        Image.fromarray(img).save(path)
        for size in (64, 128):
            out = ref_ds.process_image(_load_image(path), size).numpy()
            g1[f"{name}_{size}"] = out.astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "g1_process_image.npz"), **g1)

    # ---- G2: generator draw order ----------------------------------------------------------
    gen = ref_ds.get_generator(2334, "cpu")
    draws = [torch.randn((1, 4, 16, 16), generator=gen) for _ in range(4)]
    g2 = {"seed": 2334, "shape": [1, 4, 16, 16],
          "first8": [d.flatten()[:8].tolist() for d in draws],
          "sha256": [sha(d.numpy()) for d in draws]}
    json.dump(g2, open(os.path.join(HERE, "g2_generator.json"), "w"), indent=1)

    # ---- G3: hacked_AttnProcessor2_0 q/k/v -------------------------------------------------
    torch.manual_seed(3)
    attn = R.Attention(64, 4)
    x = torch.randn(2, 48, 64)
    with torch.no_grad():
        out, q, k, v, res = ref_attn.hacked_AttnProcessor2_0()(attn, x)
    np.savez_compressed(os.path.join(HERE, "g3_attn_qkv.npz"), x=x.numpy(),
                        wq=attn.to_q.weight.detach().numpy(), wk=attn.to_k.weight.detach().numpy(),
                        wv=attn.to_v.weight.detach().numpy(),
                        wo=attn.to_out[0].weight.detach().numpy(), bo=attn.to_out[0].bias.detach().numpy(),
                        out=out.numpy(), q=q.contiguous().numpy(), k=k.contiguous().numpy(),
                        v=v.contiguous().numpy())

    # ---- shared fake pipeline on the oracle's TINY U-Net -----------------------------------
    cfg, rcfg = C.TINY, R.TINY
    sd = S.make_state_dict(cfg, seed=0)
    unet = R.build_unet(rcfg, sd)
    ctx = S.make_context(cfg)
    pipe = DiffSimPipeline(FakeVAE(), lambda prompt: ctx, None, UNetAdapter(unet), _PNDM(), None, None)
    pipe.to = lambda *a, **k: pipe
    ds = ref_ds.DiffSim.__new__(ref_ds.DiffSim)
    ds.pipe, ds.device, ds.ip_adapter = pipe, "cpu", False

    # ---- G4: score tail through DiffSim.diffsim with a stores-filling fake step -------------
    class TailPipe:
        vae = FakeVAE()
        unet = pipe.unet

        def __init__(self):
            self.calls = 0
            self.qkv = None

        def step(self, **kw):
            mod_ = self.unet.up_blocks[1:][0].attentions[-1].transformer_blocks[-1].attn1
            mod_.stores = list(self.qkv[self.calls % 2])
            self.calls += 1
            return 0

    tp = TailPipe()
    ds_tail = ref_ds.DiffSim.__new__(ref_ds.DiffSim)
    ds_tail.pipe, ds_tail.device, ds_tail.ip_adapter = tp, "cpu", False
    img_a, img_b = os.path.join(HERE, "g1_img_c.png"), os.path.join(HERE, "g1_img_d.png")
    g4 = {}
    shapes = [(2, 8, 256, 160), (2, 4, 64, 32), (2, 8, 64, 40), (1, 2, 16, 16)]
    for i in range(10):
        shp = shapes[i % len(shapes)] if i < 8 else shapes[1]
        g = torch.Generator("cpu").manual_seed(100 + i)
        sets = [[torch.randn(shp, generator=g) * (1.5 if j == 0 else 1.0) for j in range(3)]
                for _ in range(2)]
        mixw = 0.3 + 0.07 * i          # B correlated with A so cosine scores spread over (0,1)
        sets[1] = [mixw * a + (1 - mixw) * b for a, b in zip(sets[0], sets[1])]
        if i == 8:      # A == B
            sets[1] = [t.clone() for t in sets[0]]
        for sim in ("cosine", "mse"):
            tp.qkv, tp.calls = sets, 0
            with torch.no_grad():
                s = ds_tail.diffsim(img_a, img_b, 128, "p", "up_blocks", [0], 600, seed=2334,
                                    device="cpu", similarity=sim)
            g4[f"score_{i}_{sim}"] = np.asarray(s.numpy(), dtype=np.float32).reshape(-1)
        g4[f"seed_{i}"] = np.array([100 + i])
        g4[f"shape_{i}"] = np.array(shp)
    np.savez_compressed(os.path.join(HERE, "g4_tail.npz"), **g4)

    # ---- G5: whole DiffSim.diffsim + DiffSimPipeline.step on the oracle's TINY U-Net --------
    g5 = {}
    cases = [("up_blocks", [0], 600, "cosine"), ("up_blocks", [0], 600, "mse"),
             ("up_blocks", [1], 600, "cosine"),          # single-valued -> coerced to 0
             ("up_blocks", [1, 1], 500, "cosine"),       # len 2 list is passed through -> TypeError?
             ("down_blocks", [0], 750, "cosine"), ("mid_blocks", [0], 900, "cosine")]
    for ci, (blk, layer, step, sim) in enumerate(cases):
        try:
            with torch.no_grad():
                s = ds.diffsim(img_a, img_b, 128, "The photo of a cat", blk, layer, step,
                               seed=2334, device="cpu", similarity=sim)
            g5[f"score_{ci}"] = np.asarray(s.numpy(), dtype=np.float32).reshape(-1)
        except Exception as e:                                               # noqa: BLE001
            g5[f"error_{ci}"] = np.array([type(e).__name__])
        g5[f"case_{ci}"] = np.array([blk, json.dumps(layer), str(step), sim])
    # q/k/v of image B (last step) for case 0, to pin the features themselves
    with torch.no_grad():
        ds.diffsim(img_a, img_b, 128, "The photo of a cat", "up_blocks", [0], 600, seed=2334,
                   device="cpu", similarity="cosine")
    mod_ = pipe.unet.up_blocks[1:][0].attentions[-1].transformer_blocks[-1].attn1
    qb, kb, vb = mod_.stores
    g5["qB"], g5["kB"], g5["vB"] = (t.contiguous().numpy() for t in (qb, kb, vb))
    # the latents the fake VAE produced, so the build's latents-in entry can be driven too
    gen = ref_ds.get_generator(2334, "cpu")
    tA = ref_ds.process_image(_load_image(img_a), 128)
    tB = ref_ds.process_image(_load_image(img_b), 128)
    lA = FakeVAE().encode(tA.to(torch.float16)).latent_dist.sample(gen) * 0.18215   # diffsim.py:93 casts to fp16
    lB = FakeVAE().encode(tB.to(torch.float16)).latent_dist.sample(gen) * 0.18215
    nA = torch.randn(lA.shape, generator=gen)
    nB = torch.randn(lB.shape, generator=gen)
    g5["latA"], g5["latB"], g5["noiseA"], g5["noiseB"] = (t.numpy() for t in (lA, lB, nA, nB))
    np.savez_compressed(os.path.join(HERE, "g5_e2e_tiny.npz"), **g5)

    # ---- G6: hacked block control flow bound onto the oracle's modules ----------------------
    g6 = {}
    torch.manual_seed(11)
    temb = torch.randn(2, rcfg.time_embed_dim)
    # down block 1 (CrossAttnDown, 64->128 @ 8x8)
    blk = copy.deepcopy(unet.down_blocks[1]); blk.use_ipa = False; blk.gradient_checkpointing = False
    x = torch.randn(2, 64, 8, 8)
    with torch.no_grad():
        h, outs = ref_mod.hacked_CrossAttnDownBlock2D_forward.__get__(blk, type(blk))(
            x, temb=temb, encoder_hidden_states=ctx)
    g6["down_x"], g6["down_h"] = x.numpy(), h.numpy()
    g6["down_outs"] = np.stack([o.numpy().reshape(-1)[:64] for o in outs])
    g6["down_q"], g6["down_k"], g6["down_v"] = (t.contiguous().numpy() for t in blk.stores[:3])
    # mid
    blk = copy.deepcopy(unet.mid_block); blk.use_ipa = False; blk.gradient_checkpointing = False
    x = torch.randn(2, 256, 2, 2)
    with torch.no_grad():
        h = ref_mod.hacked_UNetMidBlock2DCrossAttn_forward.__get__(blk, type(blk))(
            x, temb=temb, encoder_hidden_states=ctx)
    g6["mid_x"], g6["mid_h"] = x.numpy(), h.numpy()
    g6["mid_q"], g6["mid_k"], g6["mid_v"] = (t.contiguous().numpy() for t in blk.stores[:3])
    # up block 1 (CrossAttnUp @ 4x4; skips 256,256,128)
    blk = copy.deepcopy(unet.up_blocks[1])
    x = torch.randn(2, 256, 4, 4)
    skips = (torch.randn(2, 128, 4, 4), torch.randn(2, 256, 4, 4), torch.randn(2, 256, 4, 4))
    with torch.no_grad():
        h = ref_mod.hacked_CrossAttnUpBlock2D_forward.__get__(blk, type(blk))(
            x, skips, temb=temb, encoder_hidden_states=ctx)
    g6["up_x"], g6["up_h"] = x.numpy(), h.numpy()
    for i, s_ in enumerate(skips):
        g6[f"up_skip{i}"] = s_.numpy()
    g6["up_q"], g6["up_k"], g6["up_v"] = (t.contiguous().numpy() for t in blk.stores[:3])
    g6["temb"] = temb.numpy()
    np.savez_compressed(os.path.join(HERE, "g6_blocks.npz"), **g6)

    # ---- G7: scheduler facts ----------------------------------------------------------------
    ts = R.pndm_timesteps()
    ac = R.alphas_cumprod()
    g7 = {"pndm_len": int(len(ts)), "pndm_head": ts[:4].tolist(), "pndm_tail": ts[-3:].tolist(),
          "pndm_idx": {str(i): int(ts[i]) for i in (500, 600, 750, 900)},
          "abar_401": float(ac[401]), "sqrt_abar_401": float(ac[401] ** 0.5),
          "sqrt_1m_abar_401": float((1 - ac[401]) ** 0.5),
          "source": "restated (diffusers absent); values cross-checked with SURVEY.md section 0"}
    try:
        from DiT.diffusion import create_diffusion
        d = create_diffusion("600")
        g7["dit_map_len"] = len(d.timestep_map)
        g7["dit_map_400"] = int(d.timestep_map[400])
        g7["dit_map_head"] = [int(v) for v in d.timestep_map[:3]]
        g7["dit_map_tail"] = [int(v) for v in d.timestep_map[-3:]]
        g7["dit_source"] = "reference DiT/diffusion/respace.py run here"
    except Exception as e:                                                   # noqa: BLE001
        g7["dit_error"] = repr(e)
    json.dump(g7, open(os.path.join(HERE, "g7_sched.json"), "w"), indent=1)
    print("golden fixtures written to", HERE)
    for f in sorted(os.listdir(HERE)):
        print(f"  {f:28s} {os.path.getsize(os.path.join(HERE, f)):9d} B")


if __name__ == "__main__":
    main()
