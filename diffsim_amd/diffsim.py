"""DiffSim scorer with the reference's entry points, backed by the MI355X engine.

Mirrors ``/root/reference/diffsim/diffsim.py``:
  * ``get_generator``            :16-25
  * ``DiffSim.__init__``         :80-90   (pipeline load -> here: config + state dict + encoders)
  * ``prepare_image_latents``    :92-96
  * ``DiffSim.diffsim``          :98-197  same argument names, returns a (1,) tensor
and folds in what ``DiffSimPipeline.step`` does around the U-Net call
(``/root/reference/diffsim/diffsim_pipeline.py:125-221``): prompt context, index -> timestep,
noise draw, add_noise, CFG duplication.  Deliberate differences (SURVEY.md Appendix C "fix"
rows, none of which changes a score): the U-Net stops at the tap, no hook is leaked per call,
the prompt context and image-slot features are cached, both images run in one batch.

The VAE encoder and CLIP text encoder are "next" rows of the scope table: they are
pluggable (``vae`` with diffusers' ``encode(x).latent_dist.sample(generator)`` surface and an
``encode_prompt(prompt) -> (2, L, Dc)`` callable).  Without them only the latents-in entry
points work; the path-based ones raise.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple, Union

from concurrent.futures import ThreadPoolExecutor

import torch

from . import scheduler as sched
from .config import SD15, UNetConfig
from .engine import UNetEngine, pair_score
from .image import DecodePool, host_threads, load_image, process_image


def get_generator(seed, device="cpu"):
    if seed is not None:
        if isinstance(seed, list):
            generator = [torch.Generator(device).manual_seed(int(s)) for s in seed]
        else:
            generator = torch.Generator(device).manual_seed(int(seed))
    else:
        generator = None
    return generator


def _norm_layer(target_layer) -> int:
    # diffsim/diffsim.py:99-100: a single-valued --target_layer is coerced to 0
    if isinstance(target_layer, int):
        return target_layer
    if len(target_layer) == 1:
        return 0
    # the reference indexes a ModuleList with the list itself -> TypeError; keep the error
    raise TypeError("list indices must be integers or slices, not list")


class DiffSim:
    def __init__(self, torch_dtype=torch.bfloat16, device="cuda", ip_adapter=False, *,
                 unet_config: UNetConfig = SD15, state_dict: Optional[Dict[str, torch.Tensor]] = None,
                 vae=None, encode_prompt: Optional[Callable[[str], torch.Tensor]] = None,
                 vae_dtype=torch.float16, use_graphs: bool = False, noise_dtype=torch.float32, dedup_cfg: bool = True,
                 fusion: Optional[int] = None, decode_procs: Optional[int] = None):
        if ip_adapter:
            raise NotImplementedError("IP-Adapter mode is out of scope (SURVEY.md section 2 row 3)")
        if state_dict is None:
            raise ValueError("state_dict (diffusers-keyed U-Net weights) is required: no checkpoint is bundled")
        # torch.float16 is what every reference driver passes (cute_main.py:31): the kernels then compute in IEEE fp16
        # (v_mfma_f32_16x16x32_f16, same rate as bf16); torch.bfloat16 is the headline mode, torch.float32 the parity mode
        if torch_dtype not in (torch.float32, torch.bfloat16, torch.float16):
            raise ValueError("torch_dtype must be torch.float32, torch.bfloat16 or torch.float16")
        self.dtype = torch_dtype
        self.device = torch.device("cuda:0" if device == "cuda" else device)
        self.ip_adapter = False
        self.cfg = unet_config
        self.state_dict = state_dict
        self.vae = vae
        self.vae_dtype = vae_dtype
        self._encode_prompt = encode_prompt
        self.use_graphs = use_graphs            # replay each U-Net forward as one hipGraph (small, launch-bound batches)
        # dtype of the generator draws and of the add_noise arithmetic.  float32 = the reference run as an fp32 CPU
        # pipeline (the north_star parity setting).  float16 = the reference as its drivers construct it
        # (DiffSim(torch.float16), diffsim.py:79-83): randn_tensor(dtype=latents.dtype) draws in fp16 -- a different
        # random stream from the fp32 draw of the same seed -- the VAE sample is drawn in fp16 too, and
        # scheduler.add_noise runs in fp16 (diffsim_pipeline.py:174-183)
        if noise_dtype not in (torch.float32, torch.float16):
            raise ValueError("noise_dtype must be torch.float32 or torch.float16")
        self.noise_dtype = noise_dtype
        if vae is not None and hasattr(vae, "sample_dtype"):
            vae.sample_dtype = noise_dtype
        # default (opt out with dedup_cfg=False): conv_in, the first resnet and the first transformer's self-attention are
        # identical in the two CFG halves (torch.cat([latents] * 2), diffsim_pipeline.py:208); compute them once per image.
        # Bit-identical scores (tests/test_gpu_round2.py::test_cfg_dedup_is_bit_identical), ~6 % faster.  bench.py's headline
        # line passes dedup_cfg=False: it executes every algorithmic FLOP of the reference's duplicated batch.
        self.dedup_cfg = bool(dedup_cfg)
        self.fusion = fusion                    # None = the library default (every fused kernel); 0 = one launch per layer
        self._base: Optional[UNetEngine] = None
        self._engines: Dict[Tuple[str, int], object] = {}
        self._ctx: Dict[str, torch.Tensor] = {}
        self._pool = ThreadPoolExecutor(max_workers=host_threads())     # host-side image decode / resize: cores / ranks of the node
        # files-in paths: decode + Lanczos resize run ahead of the GPU in worker processes (decode_procs > 0; None = the
        # DSIM_DECODE_PROCS environment default) or threads (0); started lazily, on the first path batch
        self._decode = DecodePool(decode_procs)
        self._streams: List[torch.cuda.Stream] = []          # side streams of score_latent_pairs

    # ------------------------------------------------------------------------------------------
    def engine(self, target_block: str, target_layer: int) -> UNetEngine:
        """The engine positioned at a tap.  ONE packed weight copy (per scorer, i.e. per dtype) serves every tap:
        the handle's tap is moved, nothing is re-packed."""
        key = (target_block, int(target_layer))
        if key not in self._engines:
            if self._base is None:
                self._base = UNetEngine(self.cfg, self.state_dict, self.dtype, target_block, int(target_layer),
                                        str(self.device))
                self._base.use_graphs = self.use_graphs
                if self.dedup_cfg:
                    self._base.set_cfg_dedup(True)
                if self.fusion is not None:
                    self._base.set_fusion(self.fusion)
            self._engines[key] = self._base.view(target_block, int(target_layer))
            self._engines[key].tokens            # moves the tap once: a missing weight raises here, not mid-run
        return self._engines[key]

    def context(self, prompt: Union[str, torch.Tensor]) -> torch.Tensor:
        """[uncond, cond] prompt embeddings (2, L, Dc) f32 on the device; cached per prompt
        (the reference re-encodes the constant prompt on every call, diffsim_pipeline.py:125)."""
        if isinstance(prompt, torch.Tensor):
            return prompt.to(self.device, torch.float32).contiguous()
        if prompt not in self._ctx:
            if self._encode_prompt is None:
                raise RuntimeError("no text encoder plugged in: pass encode_prompt=... or a (2,L,Dc) tensor as prompt")
            self._ctx[prompt] = self._encode_prompt(prompt).to(self.device, torch.float32).contiguous()
        return self._ctx[prompt]

    def prepare_image_latents(self, image, vae=None, device=None, generator=None):
        vae = vae or self.vae
        if vae is None:
            raise RuntimeError("no VAE plugged in: use the latents-in entry points (diffsim_latents / score_latent_pairs)")
        if isinstance(getattr(vae, "device", None), torch.device):
            image = image.to(vae.device)          # cast on the device: the same RNE rounding, without 4 ms of host time
        image = image.to(dtype=self.vae_dtype)
        lat = vae.encode(image).latent_dist.sample(generator=generator)
        return vae.config.scaling_factor * lat

    def _pair_latents(self, tensor_A, tensor_B, generator):
        """Latents of both images.  With the HIP VAE encoder the two images go through ONE encode (its kernels are
        batch-invariant bit for bit) and are then sampled in the reference's order -- A's draw, then B's."""
        vae = self.vae
        if vae is not None and hasattr(vae, "moments") and tensor_A.shape == tensor_B.shape:
            from .engine import _LatentDist
            x = torch.cat([tensor_A, tensor_B]).to(vae.device).to(dtype=self.vae_dtype)
            mom = vae.moments(x)
            sf = vae.config.scaling_factor
            return (sf * _LatentDist(mom[0:1], self.noise_dtype).sample(generator=generator),
                    sf * _LatentDist(mom[1:2], self.noise_dtype).sample(generator=generator))
        return (self.prepare_image_latents(tensor_A, vae, None, generator),
                self.prepare_image_latents(tensor_B, vae, None, generator))

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def features(self, latents: torch.Tensor, noise: torch.Tensor, prompt, target_block, target_layer, target_step):
        """latents/noise (n,4,s,s) -> q,k,v [n][2][N][H*D] (compute dtype, on device)."""
        eng = self.engine(target_block, target_layer)
        t = sched.timestep_from_index(int(target_step))
        eng.set_timestep(t)
        sa, sb = sched.noise_coefficients(t)
        if self.noise_dtype == torch.float16:
            # PNDMScheduler.add_noise on fp16 tensors: alphas_cumprod cast to fp16, sqrt and 1-x in fp16, the two
            # products and the sum each rounded to fp16 (elementwise, so the batch can be done on the device)
            dev = self.device
            ac = sched.alphas_cumprod()[t].to(torch.float16)
            a16, b16 = ac ** 0.5, (1 - ac) ** 0.5
            xt = a16.to(dev) * latents.to(dev, torch.float16) + b16.to(dev) * noise.to(dev, torch.float16)
            lat = xt.float().contiguous()
            ctx16 = self.context(prompt).to(torch.float16).float()         # the fp16 text encoder's output
            return eng.qkv(lat, torch.zeros_like(lat), 1.0, 0.0, ctx16)
        lat = latents.to(self.device, torch.float32).contiguous()
        nz = noise.to(self.device, torch.float32).contiguous()
        return eng.qkv(lat, nz, sa, sb, self.context(prompt))

    def auto_batch_pairs(self, eng, n_pairs: int, streams: int = 2, target: int = 64) -> int:
        """Pairs per chunk when the caller names none: `target` (the sweep's optimum on a 288 GB part), capped by the 2 GiB
        activation bound, by the job and by the free HBM the per-stream arenas may take (half of what is free now)."""
        bp = max(1, min(target, eng.max_images() // 2, max(1, int(n_pairs))))
        try:
            free, _total = torch.cuda.mem_get_info(self.device)
        except Exception:
            return bp
        while bp > 1:
            ns = max(1, min(int(streams), -(-int(n_pairs) // bp)))
            if eng.workspace_bytes(2 * bp) * ns <= 0.5 * free:
                break
            bp = (bp + 1) // 2
        return bp

    @torch.no_grad()
    def score_latent_pairs(self, latA, latB, noiseA, noiseB, prompt, target_block="up_blocks", target_layer=0,
                           target_step=600, similarity="cosine", batch_pairs: Optional[int] = None, streams: int = 2) -> torch.Tensor:
        """Batched latents-in scoring: pair i = (latA[i], latB[i]) -> scores (n,) f32 on device.
        noiseA/noiseB are (1,4,s,s) (shared by every pair: each reference call reseeds) or (n,4,s,s).
        Consecutive chunks of `batch_pairs` pairs are enqueued on `streams` HIP streams in turn, so the HBM-bound kernels of
        one chunk overlap the MFMA-bound kernels of the next (same kernels, same scores).  Each stream in use owns one
        workspace arena of the engine (streams = 2 -> two arenas, ~0.75 GB per pair of the chunk size each).
        batch_pairs=None picks the measured optimum (profiles/r04h_batch_sweep.txt: 64 pairs per chunk, 662 pairs/s against
        603 at 16) within what fits: every activation < 2 GiB and the arenas of the streams in use inside the free HBM."""
        n = latA.shape[0]
        eng = self.engine(target_block, target_layer)
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        if batch_pairs is None:
            batch_pairs = self.auto_batch_pairs(eng, n, streams)
        batch_pairs = max(1, min(batch_pairs, eng.max_images() // 2))      # every activation must stay < 2 GiB
        starts = list(range(0, n, batch_pairs))
        ns = max(1, min(int(streams), len(starts)))
        if self.use_graphs or getattr(eng, "_profiling", False):
            ns = 1          # (per-launch profile brackets are only a kernel's own time when nothing overlaps it)
        main = torch.cuda.current_stream(self.device)
        if ns > 1:
            if len(self._streams) < ns:
                self._streams += [torch.cuda.Stream(device=self.device) for _ in range(ns - len(self._streams))]
            # everything the chunks share is produced on the main stream BEFORE the side streams fork from it: the prompt
            # context and the timestep tables (set_timestep enqueues kernels; a later chunk on another stream would see
            # "already set" on the host while those kernels are still running)
            ctx_ready = self.context(prompt)
            eng.set_timestep(sched.timestep_from_index(int(target_step)))
            for st in self._streams[:ns]:
                st.wait_stream(main)
        for ci, i0 in enumerate(starts):
            i1 = min(n, i0 + batch_pairs)
            m = i1 - i0
            with torch.cuda.stream(self._streams[ci % ns] if ns > 1 else main):
                lat = torch.stack([latA[i0:i1], latB[i0:i1]], dim=1).reshape(2 * m, *latA.shape[1:])
                nA = noiseA[i0:i1] if noiseA.shape[0] == n else noiseA.expand(m, *noiseA.shape[1:])
                nB = noiseB[i0:i1] if noiseB.shape[0] == n else noiseB.expand(m, *noiseB.shape[1:])
                nz = torch.stack([nA, nB], dim=1).reshape(2 * m, *latA.shape[1:])
                q, k, v = self.features(lat, nz, prompt if ns == 1 else ctx_ready, target_block, target_layer, target_step)
                ia = torch.arange(0, 2 * m, 2, dtype=torch.int32, device=self.device)
                out[i0:i1] = pair_score(q, k, v, ia, ia + 1, eng.heads, similarity)
        if ns > 1:
            for st in self._streams[:ns]:
                main.wait_stream(st)
        return out

    @torch.no_grad()
    def diffsim_latents(self, latA, latB, noiseA, noiseB, prompt, target_block="up_blocks", target_layer=0,
                        target_step=600, similarity="cosine") -> torch.Tensor:
        return self.score_latent_pairs(latA, latB, noiseA, noiseB, prompt, target_block, _norm_layer(target_layer),
                                       target_step, similarity)

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def diffsim(self, image_A, image_B, img_size, prompt, target_block, target_layer, target_step, ip_adapter=False,
                seed="2333", device="cuda", similarity="cosine"):
        """Same contract as the reference's ``DiffSim.diffsim`` (diffsim/diffsim.py:98-197)."""
        if ip_adapter:
            raise NotImplementedError("IP-Adapter mode is out of scope")
        target_layer = _norm_layer(target_layer)
        # decode + Lanczos resize of the two images on two host threads (PIL releases the GIL); same tensors as serially
        fa = self._pool.submit(lambda: process_image(load_image(image_A), img_size))
        tensor_B = process_image(load_image(image_B), img_size)
        tensor_A = fa.result()
        generator = get_generator(seed, "cpu")                   # reference CPU path: CPU generator
        latentsA, latentsB = self._pair_latents(tensor_A, tensor_B, generator)
        # DiffSimPipeline.step draws the noise right after prepare_latents: A's step first, then B's
        noiseA = torch.randn(latentsA.shape, generator=generator, dtype=self.noise_dtype)
        noiseB = torch.randn(latentsB.shape, generator=generator, dtype=self.noise_dtype)
        if self.noise_dtype == torch.float16:      # the fp16 pipeline holds its latents in fp16
            latentsA, latentsB = latentsA.to(torch.float16), latentsB.to(torch.float16)
        return self.score_latent_pairs(latentsA.float(), latentsB.float(), noiseA.float(), noiseB.float(), prompt,
                                       target_block, target_layer, target_step, similarity)

    @torch.no_grad()
    def score_pairs(self, pairs: Sequence[Tuple[str, str]], img_size, prompt, target_block, target_layer, target_step,
                    seed="2333", similarity="cosine", batch_pairs: Optional[int] = None) -> torch.Tensor:
        """Batched equivalent of calling :meth:`diffsim` once per (A, B) path pair."""
        target_layer = _norm_layer(target_layer)
        unet_bp = batch_pairs                 # None: score_latent_pairs picks its own chunk (64 where it fits)
        if batch_pairs is None:
            batch_pairs = 16                  # pairs per VAE encode: 32 images at 512 px keep its widest activation < 2 GiB
        lA, lB = [], []
        nA = nB = None
        vae = self.vae
        if vae is not None and hasattr(vae, "moments"):
            # HIP VAE: images decoded on the host thread pool, one encode per chunk of pairs; every pair reseeds the
            # same generator, so its four draws (vaeA, vaeB, noiseA, noiseB) are the same tensors for all pairs
            from .engine import image_preprocess, latent_sample
            g = get_generator(seed, "cpu")
            eps = None
            sf = vae.config.scaling_factor
            nd = self.noise_dtype
            def submit(i0):
                return self._decode.submit([p for ab in pairs[i0:i0 + batch_pairs] for p in ab], img_size)
            starts = list(range(0, len(pairs), batch_pairs))
            pending = [submit(i0) for i0 in starts[:2]]          # decode + resize run two chunks ahead of the GPU
            for ci, i0 in enumerate(starts):
                px = DecodePool.gather(pending.pop(0))
                if ci + 2 < len(starts):
                    pending.append(submit(starts[ci + 2]))
                # process_image's arithmetic and the fp16 image cast on the device (bit-identical, dsim_image_preprocess)
                x = image_preprocess(px.to(vae.device, non_blocking=True), self.vae_dtype == torch.float16)
                mom = vae.moments(x)
                if eps is None:
                    shp = (1, mom.shape[1] // 2) + tuple(mom.shape[2:])
                    eA = torch.randn(shp, generator=g, dtype=nd).float().to(vae.device)
                    eB = torch.randn(shp, generator=g, dtype=nd).float().to(vae.device)
                    nA = torch.randn(shp, generator=g, dtype=nd).float()
                    nB = torch.randn(shp, generator=g, dtype=nd).float()
                    eps = (eA, eB)
                lA.append(latent_sample(mom, eps[0], sf, 0, 2, nd == torch.float16))
                lB.append(latent_sample(mom, eps[1], sf, 1, 2, nd == torch.float16))
            return self.score_latent_pairs(torch.cat(lA), torch.cat(lB), nA, nB, prompt, target_block, target_layer,
                                           target_step, similarity, unet_bp)
        for pa, pb in pairs:
            generator = get_generator(seed, "cpu")
            a = self.prepare_image_latents(process_image(load_image(pa), img_size), self.vae, None, generator)
            b = self.prepare_image_latents(process_image(load_image(pb), img_size), self.vae, None, generator)
            if nA is None:   # same seed every call -> the two noise tensors are identical for every pair
                nA = torch.randn(a.shape, generator=generator, dtype=self.noise_dtype).float()
                nB = torch.randn(b.shape, generator=generator, dtype=self.noise_dtype).float()
            lA.append(a.to(self.noise_dtype).float())
            lB.append(b.to(self.noise_dtype).float())
        return self.score_latent_pairs(torch.cat(lA), torch.cat(lB), nA, nB, prompt, target_block, target_layer,
                                       target_step, similarity, unet_bp)
