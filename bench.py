#!/usr/bin/env python3
"""DiffSim scoring throughput on MI355X: image-pairs/sec, SD1.5 graph, 512 px (64x64 latents),
tap unet.up_blocks[1].attentions[2].transformer_blocks[0].attn1 (--target_block up_blocks
--target_layer 0), --target_step 600 (t = 401), cosine.  BASELINE.json config[1].

A "step" = one pass of the hot path (noising + CFG duplication + U-Net-to-tap + q/k/v + fused
4xSDPA/cosine tail) over one batch of synthetic latent pairs already resident in HBM.
Weights are seeded random tensors of the real SD1.5 architecture (no checkpoint offline).

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

Prints ONE JSON line (rank 0) with the throughput, the roofline of the dominant kernel measured
live with HIP events on the launch stream, and the CPU oracle timed on the host cores.
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
PEAK_BF16_TFLOPS = 2500.0        # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3
TAP_KEYS_EXCLUDE = ("up_blocks.2", "up_blocks.3", "conv_norm_out", "conv_out")

# kernel family (dsim_unet_profile_get) -> symbol rocprofv3 --kernel-trace prints
def rocprof_name(fam: str) -> str:
    p = fam.split("_")
    if p[0] == "gemm":
        t = "__bf16" if p[1] == "bf16" else "float"
        bm, bn = p[2].split("x")
        mode = "1" if p[3] == "conv3" else "0"
        geglu = "true" if fam.endswith("_geglu") else "false"
        wn = "2" if bm == "256" else "1"          # 256-row tiles run 8 waves as 4x2, 128-row tiles 4x1
        ek = "1" if fam.endswith("_res") else "0"  # epilogue kind template argument: 0 plain, 1 residual
        return f"gemm_kernel<{t}, {bm}, {bn}, {mode}, {geglu}, 4, {wn}, {ek}>"
    if p[0] == "attention":
        return f"attn_kernel<{'__bf16' if p[1] == 'bf16' else 'float'}, {p[2][1:]}>"
    return fam           # groupnorm / layernorm families span several kernel symbols


def pmc_traffic(kernel: str):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary
    (profiles/rNN_pmc_hbm.json: separate --pmc FETCH_SIZE / WRITE_SIZE passes of this same
    command, gfx950 FETCH_SIZE x2 correction applied).  PMC cannot be read from inside a normal
    run, so this is the committed measurement, or None when no summary matches the kernel."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm.json")))
    for f in reversed(files):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if kernel in d:
            return d[kernel]["hbm_bytes_per_launch"]
    return None


def pmc_mfma_util(kernel: str):
    """MFMA utilisation of `kernel` (SQ_VALU_MFMA_BUSY_CYCLES / (active cycles x 1024 SIMDs)) from the newest
    committed PMC summary profiles/rNN_pmc_mfma.json, or None."""
    import glob
    for f in reversed(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_mfma.json")))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if kernel in d:
            return d[kernel]["mfma_util"]
    return None


def secondary(a, world, rank, dev):
    """Secondary bench lines: DiffSim-XL (SDXL U-Net, 1024 px, tap up_blocks [0,0,0]) and DiffSim-DiT (DiT-XL/2,
    256 px, tap blocks[13]) -- same step definition (latents resident in HBM -> scores), synthetic weights."""
    from diffsim_amd.engine import pair_score
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    bp = a.batch_pairs
    if a.model == "sdxl":
        from diffsim_amd.diffsim_xl import diffsim_xl
        cfg = C.SDXL
        drop = ("up_blocks.1", "up_blocks.2", "conv_norm_out", "conv_out", "up_blocks.0.attentions.1", "up_blocks.0.attentions.2",
                "up_blocks.0.resnets.1", "up_blocks.0.resnets.2", "up_blocks.0.upsamplers")
        keys = [k for k in C.unet_param_shapes(cfg) if not k.startswith(drop)]
        sc = diffsim_xl(dtype, str(dev), unet_config=cfg, state_dict=S.make_state_dict(cfg, seed=0, keys=keys))
        ctx, pooled = S.make_context(cfg), S.make_pooled(cfg)
        shp = (1, 4, cfg.sample_size, cfg.sample_size)
        g = torch.Generator("cpu").manual_seed(1234 + rank)
        zA, zB = torch.randn((bp,) + shp[1:], generator=g), torch.randn((bp,) + shp[1:], generator=g)
        n = S.draw_pair_noise(2334, shp)
        run = lambda: sc.score_latent_pairs(zA, zB, n[2], n[3], ctx, pooled, "up_blocks", [0, 0, 0], 600, "cosine", batch_pairs=bp)
        name = "DiffSim-XL (SDXL U-Net), synthetic 1024px pairs (latents-in), up_blocks [0,0,0] step 600, cosine"
    else:
        from diffsim_amd.diffsim_dit import diffsim_DiT
        cfg = C.DIT_XL2
        keys = [k for k in C.dit_param_shapes(cfg) if not (k.startswith("blocks.") and int(k.split(".")[1]) > 13)]
        sc = diffsim_DiT(256, 600, str(dev), dit_config=cfg, state_dict=S.make_state_dict(cfg, seed=0, keys=keys), torch_dtype=dtype,
                         fp8_attention=a.fp8_attention)
        shp = (1, 4, 32, 32)
        g = torch.Generator("cpu").manual_seed(1234 + rank)
        zA, zB = torch.randn((bp,) + shp[1:], generator=g), torch.randn((bp,) + shp[1:], generator=g)
        n = S.draw_pair_noise(2334, shp)
        run = lambda: sc.score_latent_pairs(zA, zB, n[2], n[3], 13, 600, "cosine", batch_pairs=bp)
        name = "DiffSim-DiT (DiT-XL/2), synthetic 256px pairs (latents-in), blocks[13] step 600, cosine" + \
               (", fp8 (e4m3) MFMA attention" if a.fp8_attention else "")
    zA, zB = zA.to(dev), zB.to(dev)
    n = [t.to(dev) for t in n]
    scores = run()
    for _ in range(a.warmup):
        scores = run()
    import torch.distributed as dist
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        scores = run()
    if world > 1:
        allscores = [torch.empty_like(scores) for _ in range(world)]
        dist.all_gather(allscores, scores)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    extra = {}
    if rank == 0 and a.model == "sdxl" and not a.no_profile:
        # algorithmic FLOPs of the path actually launched (per-launch records of one profiled step)
        eng = sc.engine("up_blocks", [0, 0, 0])
        eng.profile(True)
        run()
        recs = eng.profile_records()
        eng.profile(False)
        fl = sum(r[1] for r in recs)
        fam = {}
        for kn_, f_, b_, ms_ in recs:
            e = fam.setdefault(kn_, [0, 0.0, 0.0]); e[0] += 1; e[1] += f_; e[2] += ms_
        extra = {"gflop_per_pair": round(fl / bp / 1e9, 1), "whole_path_tflops_per_gpu": round(fl / (el / a.steps) / 1e12, 1),
                 "kernel_breakdown_ms_per_step": {k_: {"n": v_[0], "ms": round(v_[2], 3)} for k_, v_ in
                                                  sorted(fam.items(), key=lambda kv: -kv[1][2])[:8]}}
    if rank == 0:
        print(json.dumps({**extra, "metric": "image-pairs/sec (secondary config)", "value": round(world * a.steps * bp / el, 3), "unit": "pairs/s",
                          "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * el / a.steps, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
                          "config": {"workload": name, "pairs_per_step_per_gpu": bp},
                          "score_sample": [round(float(x), 6) for x in scores[:4].float().cpu()]}), flush=True)


def cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_sd15(cfg, sd, lats, noise, gpu_scores, n_pairs):
    """CPU baseline (rank 0, N=1 only): the oracle (fp32 torch CPU restatement of the reference path) on the host
    cores over the first `n_pairs` pairs of the batch -- BASELINE config[0] is 4 pairs -- one pair per call as the
    reference's batch-of-one loop runs (cute_main.py:111-132), U-Net truncated at the tap; then ONE pair in the
    schedule the reference itself executes (full U-Net to conv_out, diffsim_pipeline.py:213-221: same score, about
    twice the work).  Also the parity check of those pairs against the HIP scores."""
    from oracle import cpu_ref as R
    full = dict(sd)
    for k_, shp in C.unet_param_shapes(cfg).items():
        if k_ not in full:
            full[k_] = torch.zeros(shp)
    unet = R.build_unet(R.SD15, full)
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
    return {
        "cpu_baseline": {"value": round(n_pairs / cpu_s, 5), "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
                         "sample": "%d pairs (the first pairs of the batch; BASELINE config[0] = 4 pairs), one pair per call, fp32 "
                                   "torch CPU oracle, U-Net truncated at the tap, %.1f s of CPU work" % (n_pairs, cpu_s),
                         "cpu_model": cpu_model(), "host_cpu_count": os.cpu_count(),
                         "reference_schedule_value": round(1.0 / cpu_full_s, 5),
                         "reference_schedule": "pair 0, full U-Net to conv_out as diffsim_pipeline.py:213 runs it "
                                               "(score %.6f, %.1f s)" % (float(so_full), cpu_full_s)},
        "parity_vs_cpu_oracle": {"gpu": [round(x, 6) for x in g], "cpu_oracle": [round(x, 6) for x in cpu_scores],
                                 "max_abs_err": max(errs)},
    }


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch-pairs", type=int, default=32, help="pairs per step per GPU")
    ap.add_argument("--dtype", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--resident-batches", type=int, default=4,
                    help="distinct synthetic batches kept in HBM; the timed steps cycle through them")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=4, help="pairs the CPU baseline leg scores (config[0] has 4)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--model", choices=["sd15", "sdxl", "dit"], default="sd15",
                    help="sd15 = the headline metric (BASELINE config[1]); sdxl / dit = secondary lines for configs[3], [4]")
    ap.add_argument("--fp8-attention", action="store_true", help="--model dit only: e4m3 MFMA attention in the DiT blocks (config 5)")
    ap.add_argument("--pixels-in", action="store_true",
                    help="secondary line: include the VAE encoder (512x512 pixels in HBM -> score); the headline "
                         "metric is latents-in")
    ap.add_argument("--selftest-launch", action="store_true",
                    help="CPU-only check of the N-rank launch + gloo plumbing (no kernels, no throughput)")
    a = ap.parse_args()

    # ---- N ranks: either a torch.distributed launcher started us (WORLD_SIZE set), or we start them ourselves -----
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        if not a.selftest_launch and torch.cuda.device_count() < a.gpus:      # device_count() does not initialise HIP
            raise SystemExit(f"bench.py: --gpus {a.gpus} but only {torch.cuda.device_count()} GPU(s) visible")
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
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # every rank builds the same synthetic weights on the host: share the cores instead of oversubscribing them
        torch.set_num_threads(max(1, (os.cpu_count() or 8) // (2 * world)))
        dist.init_process_group("nccl", device_id=dev)

    from diffsim_amd.diffsim import DiffSim
    from diffsim_amd.engine import pair_score

    if a.model != "sd15":
        return secondary(a, world, rank, dev)
    cfg = C.SD15
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    keys = [k for k in C.unet_param_shapes(cfg) if not k.startswith(TAP_KEYS_EXCLUDE)]
    sd = S.make_state_dict(cfg, seed=0, keys=keys)
    ds = DiffSim(torch_dtype=dtype, device=str(dev), unet_config=cfg, state_dict=sd)
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
    lat = lat_all[0]
    nz = torch.cat([noise[2], noise[3]] * bp).to(dev)
    ctx = S.make_context(cfg).to(dev)
    ia = torch.arange(0, 2 * bp, 2, dtype=torch.int32, device=dev)
    ib = ia + 1
    shape = (2 * bp, 2, eng.tokens, eng.heads * eng.head_dim)
    qkv = tuple(torch.empty(shape, dtype=dtype, device=dev) for _ in range(3))

    if a.pixels_in:
        from diffsim_amd.engine import VAEEncoder
        vae = VAEEncoder(C.VAE_SD15, S.make_state_dict(C.VAE_SD15, seed=1), dtype, str(dev))
        imgs = torch.cat([torch.cat(S.make_image_pair(rank * bp + i, 512)) for i in range(bp)]).to(dev)   # [2*bp,3,512,512]
        eps = torch.cat([noise[0], noise[1]] * bp).to(dev)         # the two VAE-sample draws (reference order)

    def step(i=0):
        if a.pixels_in:
            mom = vae.moments(imgs)
            mean, logvar = mom.chunk(2, dim=1)
            z = (mean + torch.exp(0.5 * logvar.clamp(-30.0, 20.0)) * eps) * 0.18215
            q, k, v = eng.qkv(z.contiguous(), nz, sa, sb, ctx, out=qkv)
        else:
            q, k, v = eng.qkv(lat_all[i % NB], nz, sa, sb, ctx, out=qkv)
        return pair_score(q, k, v, ia, ib, eng.heads, "cosine")

    def barrier():
        if world > 1:
            dist.barrier()

    scores = step(0)
    for i in range(a.warmup):
        scores = step(i)
    step_scores = torch.empty((a.steps, bp), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step_scores[i] = step(i)
    if world > 1:                          # the only collective: one gather of the scalar scores of the whole run
        allscores = [torch.empty_like(step_scores) for _ in range(world)]
        dist.all_gather(allscores, step_scores)
    torch.cuda.synchronize()
    barrier()
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
    scores = step_scores[0]
    pairs_per_s = world * a.steps * bp / el

    out = {
        "metric": "image-pairs/sec at 512px, SD1.5 up_blocks[0] t=600",
        "value": round(pairs_per_s, 3), "unit": "pairs/s", "n_gpus": world, "n_ranks_seen": seen, "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": round(1e3 * el / a.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": "DiffSim SD1.5, synthetic 512px pairs (%s), up_blocks[0] t=600 (t=401), cosine"
                               % ("pixels-in incl. VAE encoder" if a.pixels_in else "latents-in"),
                   "pairs_per_step_per_gpu": bp, "distinct_pairs_resident_per_gpu": bp * (1 if a.pixels_in else NB),
                   "gflop_per_pair": GFLOP_PER_PAIR, "parallelism": f"pairs sharded x{world}"},
        "whole_path_tflops_per_gpu": round(pairs_per_s / world * GFLOP_PER_PAIR / 1e3, 2),
        "score_sample": [round(float(x), 6) for x in scores[:4].float().cpu()],
    }

    if rank == 0:
        peak = PEAK_BF16_TFLOPS if a.dtype == "bf16" else PEAK_F32_TFLOPS
        if not a.no_profile:
            # ---- roofline of the dominant kernel: HIP events around every launch of one more step
            eng.profile(True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            q, k, v = eng.qkv(lat, nz, sa, sb, ctx, out=qkv)
            e0.record()
            pair_score(q, k, v, ia, ib, eng.heads, "cosine")
            e1.record()
            recs = eng.profile_records()
            eng.profile(False)
            fam = {}
            for name, fl, by, ms in recs:
                f = fam.setdefault(name, [0, 0.0, 0.0, 0.0])
                f[0] += 1; f[1] += fl; f[2] += by; f[3] += ms
            tail_ms = e0.elapsed_time(e1)
            dom = max(fam, key=lambda n: fam[n][3])
            n, fl, by, ms = fam[dom]
            ach = fl / (ms * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                               "frac": round(ach / peak, 4), "traffic": pmc_traffic(rocprof_name(dom)),
                               "kernel": rocprof_name(dom),
                               "launches_per_step": n, "avg_launch_ms": round(ms / n, 4),
                               "algorithmic_gflop_per_launch": round(fl / n / 1e9, 2),
                               "mfma_util_pmc": pmc_mfma_util(rocprof_name(dom))}
            # the same figures for the next kernels by time (MFMA-bound: TFLOP/s of 2500; HBM-bound: GB/s of 8000)
            tops = []
            for name in sorted(fam, key=lambda n_: -fam[n_][3])[:6]:
                n_, fl_, by_, ms_ = fam[name]
                if fl_:
                    a_ = fl_ / (ms_ * 1e-3) / 1e12
                    tops.append({"kernel": rocprof_name(name), "bound": "mfma", "achieved": round(a_, 1), "peak": peak,
                                 "unit": "TFLOP/s", "frac": round(a_ / peak, 4), "ms_per_step": round(ms_, 3),
                                 "mfma_util_pmc": pmc_mfma_util(rocprof_name(name))})
                else:
                    a_ = by_ / (ms_ * 1e-3) / 1e9
                    tops.append({"kernel": rocprof_name(name), "bound": "hbm", "achieved": round(a_, 1), "peak": 8000.0,
                                 "unit": "GB/s", "frac": round(a_ / 8000.0, 4), "ms_per_step": round(ms_, 3)})
            out["roofline_top_kernels"] = tops
            out["kernel_breakdown_ms_per_step"] = {
                k_: {"n": v_[0], "ms": round(v_[3], 3),
                     **({"tflops": round(v_[1] / (v_[3] * 1e-3) / 1e12, 1)} if v_[1] else
                        {"gbps": round(v_[2] / (v_[3] * 1e-3) / 1e9, 1)})}
                for k_, v_ in sorted(fam.items(), key=lambda kv: -kv[1][3])}
            out["kernel_breakdown_ms_per_step"]["pair_tail"] = {
                "n": 2, "ms": round(tail_ms, 3), "tflops": round(bp * 2.684e9 / (tail_ms * 1e-3) / 1e12, 1)}
        if world == 1 and not a.no_cpu_baseline:
            out.update(cpu_baseline_sd15(cfg, sd, lats, noise, scores, a.cpu_pairs))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()                     # rank 0 ran one extra (profiled) step: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
