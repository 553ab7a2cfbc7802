#!/usr/bin/env python3
"""DiffSim scoring throughput on MI355X: image-pairs/sec, SD1.5 graph, 512 px (64x64 latents),
tap unet.up_blocks[1].attentions[2].transformer_blocks[0].attn1 (--target_block up_blocks
--target_layer 0), --target_step 600 (t = 401), cosine.  BASELINE.json config[1].

A "step" = one pass of the hot path (noising + CFG duplication + U-Net-to-tap + q/k/v + fused
4xSDPA/cosine tail) over one batch of synthetic latent pairs already resident in HBM; the timed steps
cycle through several distinct resident batches.  Weights are seeded random tensors of the real SD1.5
architecture (no checkpoint offline).

  python bench.py --gpus N --steps K --warmup W

N > 1 runs N ranks, one per GPU: either a torch.distributed launcher started this script N times
(WORLD_SIZE in the environment -- how the driver runs it) or the script starts the N ranks itself.  A
run whose --gpus differs from WORLD_SIZE is refused.  Rank 0 prints ONE JSON line with the throughput
(whole job: pairs of all ranks / max-over-ranks time), the roofline of the dominant kernel measured
live with HIP events on the launch stream, and -- at N = 1 -- the CPU oracle timed on the host cores.
Secondary lines (--model sdxl | dit, --pixels-in) keep the same step definition and fields.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from diffsim_amd import config as C          # noqa: E402
from diffsim_amd import scheduler as sched   # noqa: E402
from diffsim_amd import synth as S           # noqa: E402

GFLOP_PER_PAIR = 1580.4          # 4 x 197.2146 GMAC x 2 + 2.684 (tail)
GFLOP_PER_PAIR_PIXELS = 3813.7   # + the VAE encoder of both images: 2 x 558.33 GMAC x 2
PEAK_BF16_TFLOPS = 2500.0        # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBPS = 8000.0
TAP_KEYS_EXCLUDE = ("up_blocks.2", "up_blocks.3", "conv_norm_out", "conv_out")


# kernel family (dsim_*_profile_get) -> symbol rocprofv3 --kernel-trace prints
def rocprof_name(fam: str) -> str:
    if fam.startswith("vae_"):                        # the VAE encoder's launches of a --pixels-in profile: same kernel symbols
        return rocprof_name(fam[4:])
    p = fam.split("_")
    if p[0] == "gemm" and p[1] == "small":           # gemm_small_<dt>_<bm>x<bn>_<mode>[_res]: the small-batch kernel (gemm_skinny.hip)
        bm, bn = p[3].split("x")
        return f"gemm_skinny_kernel<{'1' if p[4] == 'conv3' else '0'}, {'true' if fam.endswith('_res') else 'false'}, {bm}, {bn}>"
    if p[0] == "gemm":
        t = {"bf16": "__bf16", "f16": "_Float16"}.get(p[1], "float")
        bm, bn = p[2].split("x")
        mode = {"conv3": "1", "conv3p": "2"}.get(p[3], "0")      # conv3p: the instantiation for power-of-two output maps
        geglu = "true" if fam.endswith("_geglu") else "false"
        wm, wn = ("8", "1") if bm == "512" else ("4", "2" if bm == "256" else "1")      # 512-row tiles: 8 x 1 waves; 256: 4 x 2; 128: 4 x 1
        # epilogue kind: plain / residual / DiT gate(+act, +residual) / DiT tanh-GELU only; + 4 with GroupNorm statistics (_gn)
        gn = fam.endswith("_gn")
        base = fam[:-3] if gn else fam
        ek = "1" if base.endswith("_res") else ("2" if base.endswith("_dit") else ("3" if base.endswith("_act") else "0"))
        if gn:
            ek = str(int(ek) + 4)
        return f"gemm_kernel<{t}, {bm}, {bn}, {mode}, {geglu}, {wm}, {wn}, {ek}>"
    if p[0] == "attention":
        if p[1] == "fp8":
            return f"attn_fp8_kernel<{p[2][1:]}>"
        t = {"bf16": "__bf16", "f16": "_Float16"}.get(p[1], "float")
        d = p[2][1:]
        kind = p[3] if len(p) > 3 else ""              # the suffix csrc/attention.hip attention_kernel_kind() gave the family
        if kind == "p160":
            return "sdpa160_kernel"
        if kind == "short":
            return f"attn_short_kernel<{d}, true>"          # (the 77-key prompt context: the K80 instantiation)
        if kind == "long":
            return f"attn_long_kernel<{d}, 0>"
        if kind in ("q2", "q2fast"):
            return f"attn_q2_kernel<{d}, {'true' if kind == 'q2fast' else 'false'}>"
        return f"attn_kernel<{t}, {d}, {'true' if kind == 'fast' else 'false'}>"
    if p[0] == "ff":
        return "ff_fused_kernel<0>"
    if p[0] == "ln" and p[1] == "linear":
        return "rowlin_kernel<0>"
    return fam           # groupnorm / layernorm families span several kernel symbols


PMC_MODEL = "sd15"          # which model's PMC passes the roofline fields of this run read (set by main())


def pmc_file_class(model: str, suffix: str):
    """Exact file-name class of a committed PMC summary: profiles/rNN[x]_<suffix>.json for the headline (SD1.5) passes,
    profiles/rNN[x]_sdxl_<suffix>.json / ..._dit_<suffix>.json for the secondary models.  Never a substring glob: the
    SDXL and DiT passes launch kernels with the same symbols as the SD1.5 pass, on other shapes."""
    import re
    mid = "" if model == "sd15" else re.escape(model) + "_"
    return re.compile(r"^r(\d+)([a-z]*)_" + mid + re.escape(suffix) + r"\.json$")


def newest_pmc_file(model: str, suffix: str, kernel: str = None):
    """Newest (round number, then letter) committed summary of `model`'s `suffix` pass that lists `kernel`; None if none."""
    pat = pmc_file_class(model, suffix)
    pdir = os.path.join(ROOT, "profiles")
    cands = []
    for fn in os.listdir(pdir) if os.path.isdir(pdir) else []:
        m = pat.match(fn)
        if m:
            cands.append(((int(m.group(1)), m.group(2)), fn))
    for _key, fn in sorted(cands, reverse=True):
        try:
            d = json.load(open(os.path.join(pdir, fn)))
        except Exception:
            continue
        if kernel is None or kernel in d:
            return os.path.join(pdir, fn)
    return None


PMC_SOURCES = {}            # suffix -> basename of the summary the fields of this line were read from


def _newest_pmc(suffix: str, kernel: str, field: str):
    """`field` of `kernel` from the newest committed rocprofv3 PMC summary of THIS model's pass (separate --pmc passes of
    this same command, gfx950 FETCH_SIZE x2 correction applied: profiles/summarize.py).  PMC counters cannot be read
    from inside a normal run, so this is the committed measurement, or None; the file it came from is named in the
    line's `pmc_source` (profiles/collect_round.sh re-points a snapshot's lines at the snapshot's own passes)."""
    f = newest_pmc_file(PMC_MODEL, suffix, kernel)
    if f is None:
        return None
    PMC_SOURCES[suffix] = os.path.basename(f)
    return json.load(open(f))[kernel][field]


def pmc_traffic(kernel: str):
    return _newest_pmc("pmc_hbm", kernel, "hbm_bytes_per_launch")


def pmc_mfma_util(kernel: str):
    return _newest_pmc("pmc_mfma", kernel, "mfma_util")


def cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def timed_steps(step, a, world, dev):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + synchronize on both sides; the score
    all_gather (the path's only collective) is inside the timed region; time = MAX over ranks.
    Returns (seconds, scores of the first timed step, ranks seen)."""
    import torch.distributed as dist
    first = step(0)
    for i in range(a.warmup):
        step(i)
    kept = torch.empty((a.steps,) + tuple(first.shape), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        kept[i] = step(i)
    if world > 1:
        gathered = [torch.empty_like(kept) for _ in range(world)]
        dist.all_gather(gathered, kept)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    el = time.perf_counter() - t0
    seen = 1
    if world > 1:
        tt = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
        c = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(c)
        seen = int(c.item())
    return el, kept[0], seen


def roofline_fields(recs, peak, tail=None, dump=None):
    """recs: per-launch (family, algorithmic flops, algorithmic bytes, ms, shape) of ONE profiled step.  The dominant
    family by time gives `roofline` (achieved = sum of its launches' algorithmic FLOPs (bytes) / sum of their HIP-event
    durations); the next families and the full breakdown ride along."""
    if dump:
        with open(dump, "w") as f:
            for name, fl, by, ms, shape in recs:
                f.write(json.dumps({"kernel": name, "shape": shape, "gflop": round(fl / 1e9, 2), "mb": round(by / 1e6, 1),
                                    "ms": round(ms, 4)}) + "\n")
    fam = {}
    for name, fl, by, ms, _shape in recs:
        f = fam.setdefault(name, [0, 0.0, 0.0, 0.0])
        f[0] += 1; f[1] += fl; f[2] += by; f[3] += ms

    def entry(name):
        n, fl, by, ms = fam[name]
        kn = rocprof_name(name)
        if fl:
            ach = fl / (ms * 1e-3) / 1e12
            return {"kernel": kn, "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "ms_per_step": round(ms, 3), "launches_per_step": n,
                    "avg_launch_ms": round(ms / n, 4), "algorithmic_gflop_per_launch": round(fl / n / 1e9, 2),
                    "mfma_util_pmc": pmc_mfma_util(kn)}
        ach = by / (ms * 1e-3) / 1e9
        return {"kernel": kn, "bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                "frac": round(ach / PEAK_HBM_GBPS, 4), "ms_per_step": round(ms, 3), "launches_per_step": n,
                "avg_launch_ms": round(ms / n, 4), "algorithmic_mb_per_launch": round(by / n / 1e6, 1)}

    order = sorted(fam, key=lambda n_: -fam[n_][3])
    # The dominant kernel symbol by time.  When the runner-up is within 5 % (the 3x3-conv GEMM and the 64x64 self-attention
    # have been that close: which one led changed from run to run), the choice is made by NAME instead -- MFMA GEMM symbols
    # first, then alphabetically -- so that the same build always reports the same kernel; the other one is always in
    # roofline_top_kernels[0].
    if len(order) > 1 and fam[order[1]][3] >= 0.95 * fam[order[0]][3]:
        close = [n_ for n_ in order if fam[n_][3] >= 0.95 * fam[order[0]][3]]
        first = sorted(close, key=lambda n_: (not n_.startswith("gemm_"), n_))[0]
        order.remove(first)
        order.insert(0, first)
    dom = entry(order[0])
    dom["traffic"] = pmc_traffic(dom["kernel"])
    dom["pmc_source"] = {"traffic": PMC_SOURCES.get("pmc_hbm"), "mfma_util_pmc": PMC_SOURCES.get("pmc_mfma")}
    out = {"roofline": dom, "roofline_top_kernels": [entry(n_) for n_ in order[1:7]],
           "kernel_breakdown_ms_per_step": {
               k_: {"n": v_[0], "ms": round(v_[3], 3),
                    **({"tflops": round(v_[1] / (v_[3] * 1e-3) / 1e12, 1)} if v_[1] else {"gbps": round(v_[2] / (v_[3] * 1e-3) / 1e9, 1)})}
               for k_, v_ in sorted(fam.items(), key=lambda kv: -kv[1][3])},
           "algorithmic_tflop_per_step_profiled": round(sum(v[1] for v in fam.values()) / 1e12, 3)}
    if tail:
        out["kernel_breakdown_ms_per_step"]["pair_tail"] = tail
    return out


def oracle_unet(R, rcfg, cfg, sd):
    """Oracle U-Net for the CPU baseline leg: meta construction + assign (the graph beyond the tap gets zero tensors)."""
    with torch.device("meta"):
        m = R.UNet2DConditionModel(rcfg)
    full = {k: (sd[k].float() if k in sd else torch.zeros(shp)) for k, shp in C.unet_param_shapes(cfg).items()}
    m.load_state_dict(full, strict=True, assign=True)
    return m.eval()


def cpu_baseline_sd15(cfg, sd, lats, noise, gpu_scores, n_pairs):
    """CPU baseline (rank 0, N=1 only): the oracle (fp32 torch CPU restatement of the reference path) on the host
    cores over the first `n_pairs` pairs of the batch -- BASELINE config[0] is 4 pairs -- one pair per call as the
    reference's batch-of-one loop runs (cute_main.py:111-132), U-Net truncated at the tap; then ONE pair in the
    schedule the reference itself executes (full U-Net to conv_out, diffsim_pipeline.py:213-221: same score, about
    twice the work).  Also the parity check of those pairs against the HIP scores."""
    from oracle import cpu_ref as R
    unet = oracle_unet(R, R.SD15, cfg, sd)
    ctx = S.make_context(cfg)
    n_pairs = max(1, min(n_pairs, len(lats)))
    cpu_scores = []
    tc = time.perf_counter()
    for i in range(n_pairs):
        zA, zB = lats[i]
        cpu_scores.append(float(R.diffsim_latents(unet, zA, zB, noise[2], noise[3], ctx, 600, "up_blocks", 0, "cosine")))
    cpu_s = time.perf_counter() - tc
    tc = time.perf_counter()
    so_full = R.diffsim_latents(unet, lats[0][0], lats[0][1], noise[2], noise[3], ctx, 600, "up_blocks", 0, "cosine", full=True)
    cpu_full_s = time.perf_counter() - tc
    g = [float(x) for x in gpu_scores[:n_pairs].float().cpu()]
    errs = [abs(a_ - b_) for a_, b_ in zip(g, cpu_scores)]
    rel = lambda a_, b_: abs(a_ - b_) / max(abs(b_), 1e-6)
    # the north_star tolerance, in the driver-run line: the SAME pairs through the fp32 kernel mode (exact-f32 MFMA chains) against
    # the CPU oracle scores just computed; the run fails above 1e-4 relative
    from diffsim_amd.diffsim import DiffSim
    ds32 = DiffSim(torch_dtype=torch.float32, device="cuda", unet_config=cfg, state_dict=sd, dedup_cfg=False)
    la = torch.cat([lats[i][0] for i in range(n_pairs)]).cuda()
    lb = torch.cat([lats[i][1] for i in range(n_pairs)]).cuda()
    g32 = [float(x) for x in ds32.score_latent_pairs(la, lb, noise[2].cuda(), noise[3].cuda(), ctx, "up_blocks", 0, 600, "cosine",
                                                      batch_pairs=n_pairs, streams=1).float().cpu()]
    del ds32
    torch.cuda.empty_cache()
    rel32 = [rel(a_, b_) for a_, b_ in zip(g32, cpu_scores)]
    return {
        "parity_fp32": {"gpu": [round(x, 7) for x in g32], "cpu_oracle": [round(x, 7) for x in cpu_scores],
                        "max_rel_err": max(rel32), "tolerance": 1e-4, "pass": max(rel32) <= 1e-4,
                        "what": "the %d CPU-baseline pairs through the fp32 kernel mode (exact-f32 MFMA) against the fp32 CPU oracle" % n_pairs},
        "cpu_baseline": {"value": round(n_pairs / cpu_s, 5), "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
                         "sample": "%d pairs (the first pairs of the batch; BASELINE config[0] = 4 pairs), one pair per call, fp32 "
                                   "torch CPU oracle, U-Net truncated at the tap, %.1f s of CPU work" % (n_pairs, cpu_s),
                         "cpu_model": cpu_model(), "host_cpu_count": os.cpu_count(),
                         "reference_schedule_value": round(1.0 / cpu_full_s, 5),
                         "reference_schedule": "pair 0, full U-Net to conv_out as diffsim_pipeline.py:213 runs it "
                                               "(score %.6f, %.1f s)" % (float(so_full), cpu_full_s)},
        "parity_vs_cpu_oracle": {"gpu": [round(x, 6) for x in g], "cpu_oracle": [round(x, 6) for x in cpu_scores],
                                 "max_abs_err": max(errs), "max_rel_err": max(rel(a_, b_) for a_, b_ in zip(g, cpu_scores))},
    }


def product_default_leg(cfg, sd, dtype, dev, lats, noise, ctx, headline_value):
    """What a user of INTEGRATION.md Option A gets: a DiffSim built with NO knobs (its defaults: CFG halves de-duplicated --
    bit-identical scores -- and the chunk size / two streams score_latent_pairs picks itself) scoring the resident pairs through
    `score_latent_pairs`, timed like the headline (warm call, then two timed calls over all resident pairs)."""
    from diffsim_amd.diffsim import DiffSim
    ds = DiffSim(torch_dtype=dtype, device=str(dev), unet_config=cfg, state_dict=sd)
    la = torch.cat([p[0] for p in lats]).to(dev)
    lb = torch.cat([p[1] for p in lats]).to(dev)
    nA, nB = noise[2].to(dev), noise[3].to(dev)
    run = lambda: ds.score_latent_pairs(la, lb, nA, nB, ctx)
    s0 = run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 2
    for _ in range(reps):
        s0 = run()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    v = reps * len(lats) / el
    del ds
    torch.cuda.empty_cache()
    return {"value_product_default": round(v, 3),
            "product_default": {"pairs_per_call": len(lats), "calls_timed": reps, "vs_headline": round(v / headline_value, 4),
                                "what": "DiffSim(dtype, device, unet_config, state_dict).score_latent_pairs(latA, latB, noiseA, noiseB, ctx) "
                                        "with no other argument: dedup_cfg on (bit-identical scores), auto chunk, two streams",
                                "score_sample": [round(float(x), 6) for x in s0[:2].float().cpu()]}}


def pixels_in_leg(a, dtype, dev, eng, noise, nz, sa, sb, ctx, qkv, ia, ib, rank):
    """The reference's real calling path in the driver-run line (`DiffSim.diffsim(image_A, image_B, ...)`,
    /root/reference/diffsim/diffsim.py:103-113: pixels -> VAE encode -> noised U-Net to the tap -> score): the same bp pairs per
    step as the headline with the SD1.5 VAE encoder (synthetic weights, 34 M parameters) in front of the U-Net, pixels resident in
    HBM.  3 timed steps after one warm step (about 1 s of GPU time); the VAE's launches are profiled with HIP events on their
    stream and the dominant one gets its own `roofline` (PMC fields from the committed --pixels-in passes)."""
    global PMC_MODEL
    from diffsim_amd.engine import VAEEncoder, latent_sample, pair_score
    bp = a.batch_pairs
    vae = VAEEncoder(C.VAE_SD15, S.make_state_dict(C.VAE_SD15, seed=1), dtype, str(dev))
    nd = min(bp, 16)                  # the host-side image generator is slow: 16 distinct pairs, tiled to bp
    base = torch.cat([torch.cat(S.make_image_pair(rank * nd + i, 512)) for i in range(nd)])
    imgs = base.repeat((bp + nd - 1) // nd, 1, 1, 1)[:2 * bp].to(dev)
    eps = torch.cat([noise[0], noise[1]] * bp).to(dev).contiguous()

    def step():
        z = latent_sample(vae.moments(imgs), eps, 0.18215)
        q, k, v = eng.qkv(z, nz, sa, sb, ctx, out=qkv)
        return pair_score(q, k, v, ia, ib, eng.heads, "cosine")

    sc = step()
    torch.cuda.synchronize()
    steps = 3
    t0 = time.perf_counter()
    for _ in range(steps):
        sc = step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    v = steps * bp / el
    out = {"value_pixels_in": round(v, 3),
           "pixels_in": {"what": "bp pairs of 512 x 512 pixels resident in HBM -> VAE encoder -> posterior sample -> the headline's step",
                         "pairs_per_step": bp, "distinct_image_pairs": nd, "steps": steps, "ms_per_step": round(1e3 * el / steps, 3),
                         "gflop_per_pair": GFLOP_PER_PAIR_PIXELS,
                         "whole_path_tflops": round(v * GFLOP_PER_PAIR_PIXELS / 1e3, 2),
                         "score_sample": [round(float(x), 6) for x in sc[:2].float().cpu()]}}
    if not a.no_profile:
        vae.profile(True)
        latent_sample(vae.moments(imgs), eps, 0.18215)
        vrecs = [("vae_" + r[0],) + tuple(r[1:]) for r in vae.profile_records(detail=True)]
        vae.profile(False)
        vms = sum(r[3] for r in vrecs)
        prev, PMC_MODEL = PMC_MODEL, "pixels_in"
        PMC_SOURCES.clear()
        rf = roofline_fields(vrecs, PEAK_F32_TFLOPS if a.dtype == "fp32" else PEAK_BF16_TFLOPS)
        PMC_MODEL = prev
        PMC_SOURCES.clear()
        out["pixels_in"].update({"vae_ms_per_image": round(vms / (2 * bp), 4), "vae_ms_per_step": round(vms, 3),
                                 "vae_tflops": round(sum(r[1] for r in vrecs) / (vms * 1e-3) / 1e12, 1),
                                 "vae_roofline": rf["roofline"], "vae_kernel_breakdown_ms_per_step": rf["kernel_breakdown_ms_per_step"]})
    del vae, imgs
    torch.cuda.empty_cache()
    return out


def secondary(a, world, rank, dev):
    """Secondary bench lines: DiffSim-XL (SDXL U-Net, 1024 px, tap up_blocks [0,0,0]; BASELINE config 4) and
    DiffSim-DiT (DiT-XL/2, 256 px, tap blocks[13]; config 5, --fp8-attention) -- same step definition (latents resident
    in HBM -> scores), synthetic weights, same roofline / cpu_baseline fields as the headline line."""
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}.get(a.dtype, torch.float32)
    bp = a.batch_pairs
    g = torch.Generator("cpu").manual_seed(1234 + rank)
    if a.model == "sdxl":
        from diffsim_amd.diffsim_xl import diffsim_xl
        cfg = C.SDXL
        drop = ("up_blocks.1", "up_blocks.2", "conv_norm_out", "conv_out", "up_blocks.0.attentions.1", "up_blocks.0.attentions.2",
                "up_blocks.0.resnets.1", "up_blocks.0.resnets.2", "up_blocks.0.upsamplers")
        sd = S.make_state_dict(cfg, seed=0, keys=[k for k in C.unet_param_shapes(cfg) if not k.startswith(drop)])
        sc = diffsim_xl(dtype, str(dev), unet_config=cfg, state_dict=sd)
        ctx, pooled = S.make_context(cfg), S.make_pooled(cfg)
        shp = (1, 4, cfg.sample_size, cfg.sample_size)
        zA, zB = torch.randn((bp,) + shp[1:], generator=g), torch.randn((bp,) + shp[1:], generator=g)
        n = S.draw_pair_noise(2334, shp)
        run = lambda i=0: sc.score_latent_pairs(zA, zB, n[2], n[3], ctx, pooled, "up_blocks", [0, 0, 0], 600, "cosine", batch_pairs=bp)
        eng = sc.engine("up_blocks", [0, 0, 0])
        name = "DiffSim-XL (SDXL U-Net), synthetic 1024px pairs (latents-in), up_blocks [0,0,0] step 600, cosine"
    else:
        from diffsim_amd.diffsim_dit import diffsim_DiT
        cfg = C.DIT_XL2
        sd = S.make_state_dict(cfg, seed=0, keys=[k for k in C.dit_param_shapes(cfg)
                                                   if not (k.startswith("blocks.") and int(k.split(".")[1]) > 13)])
        sc = diffsim_DiT(256, 600, str(dev), dit_config=cfg, state_dict=sd, torch_dtype=dtype, fp8_attention=a.fp8_attention)
        shp = (1, 4, 32, 32)
        zA, zB = torch.randn((bp,) + shp[1:], generator=g), torch.randn((bp,) + shp[1:], generator=g)
        n = S.draw_pair_noise(2334, shp)
        run = lambda i=0: sc.score_latent_pairs(zA, zB, n[2], n[3], 13, 600, "cosine", batch_pairs=bp)
        eng = sc.engine(13)
        name = "DiffSim-DiT (DiT-XL/2), synthetic 256px pairs (latents-in), blocks[13] step 600, cosine" + \
               (", fp8 (e4m3) MFMA attention" if a.fp8_attention else "")
    zA, zB = zA.to(dev), zB.to(dev)
    n = [t.to(dev) for t in n]
    el, scores, seen = timed_steps(run, a, world, dev)
    out = {"metric": "image-pairs/sec (secondary config)", "value": round(world * a.steps * bp / el, 3), "unit": "pairs/s",
           "n_gpus": world, "n_ranks_seen": seen, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * el / a.steps, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
           "config": {"workload": name, "pairs_per_step_per_gpu": bp},
           "score_sample": [round(float(x), 6) for x in scores[:4].float().cpu()]}
    if rank == 0:
        peak = PEAK_F32_TFLOPS if a.dtype == "fp32" else PEAK_BF16_TFLOPS         # fp16 MFMAs run at the bf16 rate
        if not a.no_profile:
            eng.profile(True)
            run()
            recs = eng.profile_records(detail=True)
            eng.profile(False)
            out.update(roofline_fields(recs, peak, dump=a.dump_launches))
            fl = sum(r[1] for r in recs)          # algorithmic FLOPs of the launches of one step (tail excluded)
            out["config"]["gflop_per_pair"] = round(fl / bp / 1e9, 1)
            out["whole_path_tflops_per_gpu"] = round(fl / (el / a.steps) / 1e12, 1)
        if world == 1 and not a.no_cpu_baseline:
            from oracle import cpu_ref as R
            zA1, zB1 = zA[:1].cpu(), zB[:1].cpu()
            nA, nB = n[2].cpu(), n[3].cpu()
            if a.model == "sdxl":
                unet = oracle_unet(R, R.SDXL, cfg, sd)
                tc = time.perf_counter()
                so = float(R.diffsim_xl_latents(unet, zA1, zB1, nA, nB, ctx, pooled, 600, "up_blocks", [0, 0, 0], "cosine"))
            else:
                with torch.device("meta"):
                    m = R.DiTOracle(R.DIT_XL2)
                m.load_state_dict({k: (sd[k] if k in sd else torch.zeros(v.shape)) for k, v in m.state_dict().items()},
                                  strict=True, assign=True)
                m.eval()
                tc = time.perf_counter()
                so = float(R.diffsim_dit_latents(m, zA1, zB1, nA, nB, 600, 13, "cosine"))
            cpu_s = time.perf_counter() - tc
            out["cpu_baseline"] = {"value": round(1.0 / cpu_s, 5), "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
                                   "sample": "pair 0 of the batch, fp32 torch CPU oracle truncated at the tap, %.1f s of CPU work" % cpu_s,
                                   "cpu_model": cpu_model(), "host_cpu_count": os.cpu_count()}
            out["parity_vs_cpu_oracle"] = {"gpu": [round(float(scores[0]), 6)], "cpu_oracle": [round(so, 6)],
                                           "max_abs_err": abs(float(scores[0]) - so)}
        print(json.dumps(out), flush=True)


def launch_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` without a torch.distributed launcher around it: start N fresh rank processes
    (one per GPU) and relay their output; rank 0 prints the JSON line.  The parent never touches the GPU."""
    from diffsim_amd.parallel import spawn_ranks
    return spawn_ranks(n, [sys.executable, os.path.abspath(__file__)] + list(argv))


def selftest_launch(a, world, rank):
    """CPU stand-in for the N-rank plumbing (tests/test_bench_launcher.py): gloo process group, the same barrier /
    max-over-ranks timing / score all_gather / rank census as the GPU path, no kernels."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
    el = 1e-3 * (rank + 1)
    seen = 1
    scores = torch.full((a.batch_pairs,), float(rank))
    gathered = [scores]
    if world > 1:
        dist.barrier()
        gathered = [torch.empty_like(scores) for _ in range(world)]
        dist.all_gather(gathered, scores)
        tt = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
        c = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(c)
        seen = int(c.item())
    if rank == 0:
        print(json.dumps({"selftest": True, "n_gpus": world, "n_ranks_seen": seen, "steps": a.steps, "warmup": a.warmup,
                          "max_rank_s": el, "ranks_in_gather": [int(g[0]) for g in gathered]}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def headline(a, world, rank, dev):
    from diffsim_amd.diffsim import DiffSim
    from diffsim_amd.engine import pair_score
    cfg = C.SD15
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}.get(a.dtype, torch.float32)
    keys = [k for k in C.unet_param_shapes(cfg) if not k.startswith(TAP_KEYS_EXCLUDE)]
    sd = S.make_state_dict_shared(cfg, seed=0, keys=keys, rank=rank, world=world)      # rank 0 synthesises, the others map its file
    ds = DiffSim(torch_dtype=dtype, device=str(dev), unet_config=cfg, state_dict=sd, dedup_cfg=a.dedup_cfg, fusion=a.fusion)
    if world > 1:
        import torch.distributed as dist
        # every rank has mapped rank 0's weight file -- and proves it: the fingerprints must agree before anything is scored
        fp = torch.tensor([S.shared_fingerprint(sd)], dtype=torch.int64, device=dev)
        lo, hi = fp.clone(), fp.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        S.cleanup_shared()
        if int(lo) != int(hi):
            raise SystemExit(f"bench.py: rank {rank} holds different synthetic weights than another rank (fingerprint {int(fp):x})")
    eng = ds.engine("up_blocks", 0)
    t = sched.timestep_from_index(600)
    eng.set_timestep(t)
    sa, sb = sched.noise_coefficients(t)

    # ---- NB distinct batches of synthetic pairs for this rank, resident in HBM; step i scores batch i % NB --------
    bp = a.batch_pairs
    NB = max(1, a.resident_batches)
    lats = [S.make_pair_latents(cfg, (rank * NB + b) * bp + i) for b in range(NB) for i in range(bp)]
    noise = S.draw_pair_noise(2334, lats[0][0].shape)          # reference draw order; [2],[3] = noise A,B
    lat_all = [torch.cat([torch.cat(p) for p in lats[b * bp:(b + 1) * bp]]).to(dev) for b in range(NB)]   # A0,B0,A1,B1,...
    nz = torch.cat([noise[2], noise[3]] * bp).to(dev)
    ctx = S.make_context(cfg).to(dev)
    ia = torch.arange(0, 2 * bp, 2, dtype=torch.int32, device=dev)
    ib = ia + 1
    shape = (2 * bp, 2, eng.tokens, eng.heads * eng.head_dim)
    # A step scores NS independent sub-batches of bp pairs, each enqueued on its own HIP stream: the HBM-bound kernels of
    # one sub-batch (norms) run under the MFMA-bound kernels of the other (profiles/r02_two_stream.txt)
    NS = 1 if a.pixels_in else max(1, a.streams)
    qkvs = [tuple(torch.empty((3,) + shape, dtype=dtype, device=dev).unbind(0)) for _ in range(NS)]     # (one allocation: the tap's q|k|v is one launch)
    qkv = qkvs[0]
    side = [torch.cuda.Stream(device=dev) for _ in range(NS)] if NS > 1 else []

    if a.pixels_in:
        from diffsim_amd.engine import VAEEncoder
        vae = VAEEncoder(C.VAE_SD15, S.make_state_dict(C.VAE_SD15, seed=1), dtype, str(dev))
        # bp pairs of synthetic 512 x 512 images resident in HBM (the generator is slow on the host: at most 16 distinct pairs, tiled);
        # VAEEncoder.moments encodes them in chunks that keep its widest activation < 2 GiB, the U-Net then runs ONE bp-pair batch
        nd = min(bp, 16)
        base = torch.cat([torch.cat(S.make_image_pair(rank * nd + i, 512)) for i in range(nd)])
        imgs = base.repeat((bp + nd - 1) // nd, 1, 1, 1)[:2 * bp].to(dev)          # [2*bp,3,512,512]
        eps = torch.cat([noise[0], noise[1]] * bp).to(dev).contiguous()         # the two VAE-sample draws (reference order)
        from diffsim_amd.engine import latent_sample

    def step(i=0):
        if a.pixels_in:
            z = latent_sample(vae.moments(imgs), eps, 0.18215)         # posterior sample + scaling: one HIP launch (dsim_latent_sample)
            q, k, v = eng.qkv(z, nz, sa, sb, ctx, out=qkv)
        elif NS == 1:
            q, k, v = eng.qkv(lat_all[i % NB], nz, sa, sb, ctx, out=qkv)
        else:
            main = torch.cuda.current_stream()
            parts = []
            for h, st in enumerate(side):
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    q, k, v = eng.qkv(lat_all[(i * NS + h) % NB], nz, sa, sb, ctx, out=qkvs[h])
                    parts.append(pair_score(q, k, v, ia, ib, eng.heads, "cosine"))
            for st in side:
                main.wait_stream(st)
            return torch.cat(parts)
        return pair_score(q, k, v, ia, ib, eng.heads, "cosine")

    el, scores, seen = timed_steps(step, a, world, dev)
    pairs_per_s = world * a.steps * bp * NS / el
    out = {
        "metric": "image-pairs/sec at 512px, SD1.5 up_blocks[0] t=600",
        "value": round(pairs_per_s, 3), "unit": "pairs/s", "n_gpus": world, "n_ranks_seen": seen, "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": round(1e3 * el / a.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": "DiffSim SD1.5, synthetic 512px pairs (%s), up_blocks[0] t=600 (t=401), cosine"
                               % (("pixels-in incl. VAE encoder" if a.pixels_in else "latents-in") +
                                  (", CFG halves de-duplicated up to the first cross-attention" if a.dedup_cfg else "")),
                   "pairs_per_step_per_gpu": bp * NS, "concurrent_sub_batches": NS, "pairs_per_sub_batch": bp,
                   "distinct_pairs_resident_per_gpu": min(bp, 16) if a.pixels_in else bp * NB,          # (pixels-in: 16 distinct image pairs, tiled to bp)
                   "gflop_per_pair": GFLOP_PER_PAIR_PIXELS if a.pixels_in else GFLOP_PER_PAIR, "parallelism": f"pairs sharded x{world}"},
        "whole_path_tflops_per_gpu": round(pairs_per_s / world * (GFLOP_PER_PAIR_PIXELS if a.pixels_in else GFLOP_PER_PAIR) / 1e3, 2),
        "score_sample": [round(float(x), 6) for x in scores[:4].float().cpu()],
    }
    if rank == 0:
        peak = PEAK_F32_TFLOPS if a.dtype == "fp32" else PEAK_BF16_TFLOPS         # fp16 MFMAs run at the bf16 rate
        if not a.no_profile:
            # ---- roofline of the dominant kernel: HIP events around every launch of one more step
            eng.profile(True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            vrecs = []
            if a.pixels_in:
                # the VAE encoder's launches first (family names prefixed vae_: its 512 x 512-level kernels are the pixels-in path's own)
                vae.profile(True)
                z = latent_sample(vae.moments(imgs), eps, 0.18215)
                vrecs = [("vae_" + r[0],) + tuple(r[1:]) for r in vae.profile_records(detail=True)]
                vae.profile(False)
                q, k, v = eng.qkv(z, nz, sa, sb, ctx, out=qkv)
            else:
                q, k, v = eng.qkv(lat_all[0], nz, sa, sb, ctx, out=qkv)
            e0.record()
            pair_score(q, k, v, ia, ib, eng.heads, "cosine")
            e1.record()
            recs = vrecs + eng.profile_records(detail=True)
            eng.profile(False)
            if a.pixels_in:
                vms = sum(r[3] for r in vrecs)
                out["vae_ms_per_image"] = round(vms / (2 * bp), 4)
                out["vae_tflops"] = round(sum(r[1] for r in vrecs) / (vms * 1e-3) / 1e12, 1)
            tail_ms = e0.elapsed_time(e1)
            out.update(roofline_fields(recs, peak, tail={"n": 2, "ms": round(tail_ms, 3),
                                                         "tflops": round(bp * 2.684e9 / (tail_ms * 1e-3) / 1e12, 1)},
                                       dump=a.dump_launches))
        if world == 1 and not a.pixels_in and not a.no_product_default:
            out.update(product_default_leg(cfg, sd, dtype, dev, lats, noise, ctx, pairs_per_s))
        if world == 1 and not a.pixels_in and not a.no_pixels_leg:
            out.update(pixels_in_leg(a, dtype, dev, eng, noise, nz, sa, sb, ctx, qkv, ia, ib, rank))
        if world == 1 and not a.no_cpu_baseline:
            out.update(cpu_baseline_sd15(cfg, sd, lats, noise, scores, a.cpu_pairs))
        print(json.dumps(out), flush=True)
        if "parity_fp32" in out and not out["parity_fp32"]["pass"]:
            raise SystemExit("bench.py: fp32 kernel mode is %.3g relative from the CPU oracle (tolerance 1e-4)" % out["parity_fp32"]["max_rel_err"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch-pairs", type=int, default=None,
                    help="pairs per sub-batch (one U-Net batch = 4 x this many elements); 64 = the largest whose activations stay < 2 GiB")
    ap.add_argument("--streams", type=int, default=1,
                    help="concurrent sub-batches per step, one HIP stream each (a step then scores streams x batch-pairs pairs).  "
                         "2 is ~2 %% faster (one sub-batch's norm kernels run under the other's GEMMs) and is what "
                         "DiffSim.score_latent_pairs does; the default stays 1 so that every kernel of the timed region runs alone on "
                         "the chip and its rocprofv3 duration is the kernel's own (the roofline leg divides by it)")
    ap.add_argument("--dtype", choices=["bf16", "fp16", "fp32"], default="bf16",
                    help="compute dtype: bf16 = the headline (BASELINE config 2); fp16 = the reference drivers' torch.float16; fp32 = parity mode")
    ap.add_argument("--resident-batches", type=int, default=4,
                    help="distinct synthetic batches kept in HBM; the timed steps cycle through them")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=4, help="pairs the CPU baseline leg scores (config[0] has 4)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-product-default", action="store_true",
                    help="skip the secondary leg that scores the resident pairs through DiffSim.score_latent_pairs with no knobs")
    ap.add_argument("--no-pixels-leg", action="store_true", help="skip the pixels-in leg (VAE encoder in front) of the default run")
    ap.add_argument("--dump-launches", type=str, default=None, help="write the per-launch records of the profiled step (JSON lines)")
    ap.add_argument("--model", choices=["sd15", "sdxl", "dit"], default="sd15",
                    help="sd15 = the headline metric (BASELINE config[1]); sdxl / dit = secondary lines for configs[3], [4]")
    ap.add_argument("--fp8-attention", action="store_true", help="--model dit only: e4m3 MFMA attention in the DiT blocks (config 5)")
    ap.add_argument("--pixels-in", action="store_true",
                    help="secondary line: include the VAE encoder (512x512 pixels in HBM -> score); the headline "
                         "metric is latents-in")
    ap.add_argument("--fusion", type=int, default=None,
                    help="dsim_unet_set_fusion mask (default: every fused kernel; 0 = one launch per layer, for A/B)")
    ap.add_argument("--dedup-cfg", action="store_true",
                    help="secondary line: compute what the two CFG halves share (conv_in, first resnet, first self-attention) once "
                         "per image -- bit-identical scores, 6 %% fewer FLOPs executed than the algorithmic count; the headline line "
                         "keeps the reference's duplicated batch")
    ap.add_argument("--selftest-launch", action="store_true",
                    help="CPU-only check of the N-rank launch + gloo plumbing (no kernels, no throughput)")
    a = ap.parse_args()
    if a.batch_pairs is None:
        a.batch_pairs = {"sd15": 64, "sdxl": 8, "dit": 64}[a.model]

    # ---- N ranks: either a torch.distributed launcher started us (WORLD_SIZE set), or we start them ourselves -----
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # the parent never touches the HIP runtime (not even to count devices): a rank that has no GPU says so itself
        # ("rank r needs GPU k, only n visible") and the supervisor stops the others
        raise SystemExit(launch_ranks(a.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: refusing to report a mislabelled run")
    if a.selftest_launch:
        return selftest_launch(a, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the DiffSim engine has no CPU path")
    if torch.cuda.device_count() <= local:
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local}, only {torch.cuda.device_count()} visible")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        from diffsim_amd.parallel import pin_to_gpu_numa
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        pin_to_gpu_numa(local)             # host threads (weight synthesis, launches) on the cores next to this rank's GPU
        # every rank builds the same synthetic weights on the host: share the cores instead of oversubscribing them
        torch.set_num_threads(max(1, (os.cpu_count() or 8) // (2 * world)))
        dist.init_process_group("nccl", device_id=dev)
    global PMC_MODEL
    PMC_MODEL = "pixels_in" if (a.model == "sd15" and a.pixels_in) else a.model
    if a.model != "sd15":
        secondary(a, world, rank, dev)
    else:
        headline(a, world, rank, dev)
    if world > 1:
        dist.barrier()                     # rank 0 ran one extra (profiled) step: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
