"""Command-line driver: the reference's benchmark loops on the MI355X engine.

``python -m diffsim_amd --dataset cute|nights|sref --metric diffsim|diffsim_xl|dit --model_path <dir> --image_path <dir> ...``

Flags of the reference drivers (``/root/reference/argprocess.py:5-18``) keep their names, meaning and defaults;
``--dataset`` selects which of the reference's loops runs (``cute_main.py:48-226``, ``night_main.py:24-173`` or
``style_main.py:24-196`` -- separate scripts there), ``--model_path / --dtype / --batch / --unet_batch / --ngpu / --noise_dtype`` are additions of this build (the reference
hard-codes NAS checkpoint paths, ``cute_main.py:25-31``, fp16 and one GPU, ``cute_main.sh:1``).

Multi-GPU (``--ngpu N``): the parent starts N rank processes before anything touches the GPU; triplets are sharded
whole over ranks and the per-triplet scores are all-gathered (RCCL) -- rank 0 prints the accuracy.
"""
from __future__ import annotations

import argparse
import os
import random
import sys
from typing import List, Optional, Tuple

METRICS = ["diffsim", "diffsim_xl", "clip_i", "clip_cross", "dino", "dinov1", "dino_cross", "cute", "lpips", "gram",
           "diffeats", "clipfeats", "dinofeats", "ensemble", "dit"]
ENGINE_METRICS = ("diffsim", "diffsim_xl", "dit")


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog="python -m diffsim_amd", description="DiffSim scoring (MI355X-native engine)", allow_abbrev=False)
    p.add_argument("--image_path", type=str, help="Path to image folder")
    p.add_argument("--original_path", type=str, default=None, help="Path to original images for ipref")
    p.add_argument("--out_path", type=str, help="Path to the output folder")
    p.add_argument("--image_size", type=int, default=512, help="(Resized) resolution of compared image")
    p.add_argument("--target_block", type=str, choices=["down_blocks", "mid_blocks", "up_blocks"], default="up_blocks")
    p.add_argument("--target_layer", type=int, default=2, nargs="+")
    p.add_argument("--target_step", type=int, default=100)
    p.add_argument("--metric", type=str, default="diffsim", choices=METRICS)
    p.add_argument("--similarity", type=str, choices=["cosine", "mse"], default="mse")
    p.add_argument("--prompt", type=str, default="High quality image")
    p.add_argument("--ip_adapter", action="store_true")
    p.add_argument("--use_mask", action="store_true")
    p.add_argument("--use_text_attn", action="store_true")
    p.add_argument("--seed", type=int, default=2333)
    # additions of this build
    p.add_argument("--dataset", type=str, choices=["cute", "nights", "sref"], default="cute",
                   help="which reference loop to run: cute_main.py (class/instance/lighting tree), night_main.py (data.csv) or "
                        "style_main.py (Sref / InstantStyle: one folder per style, 2000 sampled experiments)")
    p.add_argument("--experiments", type=int, default=2000, help="--dataset sref: sampled experiments (style_main.py:64)")
    p.add_argument("--model_path", type=str, default=None, help="diffusers-layout checkpoint directory (unet/, vae/, text_encoder/, tokenizer/)")
    p.add_argument("--dtype", type=str, choices=["bf16", "fp16", "fp32"], default="bf16",
                   help="engine compute dtype (fp16 = the reference drivers' torch.float16; fp32 = parity mode)")
    p.add_argument("--noise_dtype", type=str, choices=["fp32", "fp16"], default="fp32",
                   help="generator draws / add_noise arithmetic: fp32 pipeline or the reference's literal fp16 pipeline")
    p.add_argument("--batch", type=int, default=10, help="triplets per decode / VAE-encode chunk")
    p.add_argument("--unet_batch", type=int, default=None,
                   help="triplets per U-Net engine batch (default: chosen from free device memory; lower it on a shared or smaller GPU)")
    p.add_argument("--ngpu", type=int, default=None,
                   help="GPUs of this node to shard the triplets over (one process each); default 1, or the launcher's rank count")
    p.add_argument("--decode_procs", type=str, default="auto",
                   help="worker processes that decode + resize the image files ahead of the GPU: a number, 0 = threads of this "
                        "process, auto = host cores / ranks of the node")
    p.add_argument("--fp8_attention", action="store_true", help="--metric dit: e4m3 MFMA attention")
    p.add_argument("--dedup_cfg", dest="dedup_cfg", action="store_true", default=True,
                   help="--metric diffsim: compute what the two CFG halves share once per image (bit-identical scores, ~6 %% faster); the default")
    p.add_argument("--selftest_shard", action="store_true",
                   help="CPU-only check of the N-rank triplet sharding: the dataset walk, the strided shard, the two score gathers and "
                        "the printed counts over gloo with a stand-in scorer (no GPU, no weights)")
    p.add_argument("--no_dedup_cfg", dest="dedup_cfg", action="store_false",
                   help="--metric diffsim: run the reference's duplicated CFG batch through every layer")
    return p


def arg_parse(argv=None):
    return build_parser().parse_args(argv)


# ---- the CUTE triplet walk (cute_main.py:48-108) ------------------------------------------------------------------
def cute_triplets(image_path: str, seed: int) -> List[Tuple[str, str, str, str]]:
    """(A, B, C, prompt) in the order the reference's loop visits them: for every class, 10 experiments, every
    instance folder: a random lighting, two random images of it (A, B), one image of another instance under the
    same lighting (C).  The same ``random`` call sequence as the reference (scoring consumes no ``random`` state there,
    so collecting the triplets first is equivalent)."""
    random.seed(seed)
    out = []
    ext = (".png", ".jpg", ".jpeg")
    for cls in os.listdir(image_path):
        if cls == "main.py" or cls == ".DS_Store":
            continue
        cls_dir = os.path.join(image_path, cls)
        for _ in range(10):
            for sub1, dirs2, _files in os.walk(cls_dir):
                for d2 in dirs2:
                    d2_path = os.path.join(sub1, d2)
                    subs3 = [d for d in os.listdir(d2_path) if os.path.isdir(os.path.join(d2_path, d))]
                    if not subs3:
                        continue
                    sel3 = random.choice(subs3)
                    sel3_path = os.path.join(d2_path, sel3)
                    files = [f for f in os.listdir(sel3_path) if f.endswith(ext)]
                    if len(files) < 2:
                        continue
                    a, b = random.sample(files, 2)
                    others = [d for d in dirs2 if d != d2]
                    if not others:
                        continue
                    other3 = os.path.join(sub1, random.choice(others), sel3)
                    ofiles = [f for f in os.listdir(other3) if f.endswith(ext)]
                    if not ofiles:
                        continue
                    c = random.choice(ofiles)
                    out.append((os.path.join(sel3_path, a), os.path.join(sel3_path, b), os.path.join(other3, c),
                                f"The photo of a {cls}"))
    return out


# ---- the Sref / InstantStyle walk (style_main.py:48-76) ----------------------------------------------------------------
def sref_triplets(image_path: str, seed: int, prompt: str, experiments: int = 2000) -> List[Tuple[str, str, str, str]]:
    """(A, B, C, prompt) of the style benchmark: every sub-folder with at least two images is a style; each experiment
    draws two different styles, two images of the first (A, B) and one of the second (C).  The ``random`` calls are the
    reference's, in its order (seeded like it, style_main.py:27), so the same tree gives the same experiments."""
    random.seed(seed)
    ext = (".png", ".jpg", ".jpeg")
    styles = {}
    for root, dirs, _files in os.walk(image_path):
        for d in dirs:
            full = os.path.join(root, d)
            ims = [os.path.join(full, f) for f in os.listdir(full) if f.endswith(ext)]
            if len(ims) >= 2:
                styles[full] = ims
    names = list(styles.keys())
    out = []
    for _ in range(experiments):
        if len(names) < 2:
            continue
        dir_a, dir_c = random.sample(names, 2)
        a, b = random.sample(styles[dir_a], 2)
        c = random.choice(styles[dir_c])
        out.append((a, b, c, prompt))
    return out


def cute_counts(s_ab, s_ac, similarity: str) -> Tuple[int, int]:
    """correct / correct_2x of cute_main.py:196-205 (a NaN score compares False: counted wrong, as there)."""
    if similarity == "mse":
        return int((s_ab < s_ac).sum()), int((s_ab * 2 < s_ac).sum())
    return int((s_ab > s_ac).sum()), int((s_ab > 2 * s_ac).sum())


def build_scorer(args):
    """Flags -> scorer object (DiffSim / diffsim_xl / diffsim_DiT): the object graph of cute_main.py:25-31."""
    import torch
    from . import loader
    if args.metric not in ENGINE_METRICS:
        raise SystemExit(f"--metric {args.metric}: only the DiffSim metrics (diffsim, diffsim_xl, dit) run on this engine; the "
                         f"competitor metrics of the reference are out of scope")
    if args.ip_adapter:
        raise SystemExit("--ip_adapter: IP-Adapter mode is out of scope of this build")
    if not args.model_path:
        raise SystemExit("--model_path is required (no checkpoint is bundled)")
    nd = torch.float16 if args.noise_dtype == "fp16" else torch.float32
    dev = "cuda:%d" % int(os.environ.get("LOCAL_RANK", "0"))
    if args.metric == "diffsim":
        return loader.load_diffsim(args.model_path, args.dtype, dev, nd, dedup_cfg=args.dedup_cfg)
    if args.metric == "diffsim_xl":
        return loader.load_diffsim_xl(args.model_path, args.dtype, dev, nd)
    return loader.load_diffsim_dit(args.model_path, args.image_size, args.target_step, args.dtype, dev, args.fp8_attention)


def _selftest_scores(trip, rank, world):
    """Stand-in for harness.score_path_triplets on CPU (--selftest_shard): every rank scores ITS strided shard of the triplets with
    a deterministic function of the file names, then the same two gathers as the real path.  DSIM_SELFTEST_DIE_RANK=r makes
    rank r exit 9 between the two collectives (the supervising launcher must stop the others, which are then waiting in the second)."""
    import torch
    from . import parallel as P

    def fake(a, b):
        return ((sum(map(ord, os.path.basename(a))) * 31 + sum(map(ord, os.path.basename(b))) * 17) % 997) / 997.0
    mine = P.shard_triplets(len(trip), rank, world)
    loc_ab = torch.tensor([fake(trip[j][0], trip[j][1]) for j in mine], dtype=torch.float32)
    loc_ac = torch.tensor([fake(trip[j][0], trip[j][2]) for j in mine], dtype=torch.float32)
    s_ab = P.gather_scores(loc_ab, len(trip), rank, world)
    if os.environ.get("DSIM_SELFTEST_DIE_RANK") == str(rank):
        sys.exit(9)
    s_ac = P.gather_scores(loc_ac, len(trip), rank, world)
    return s_ab, s_ac, 0


def run(args) -> int:
    import torch
    from . import harness as H
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and args.selftest_shard:
        import torch.distributed as dist
        dist.init_process_group("gloo")
    elif world > 1:
        import torch.distributed as dist
        from .parallel import pin_to_gpu_numa
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        pin_to_gpu_numa(int(os.environ.get("LOCAL_RANK", "0")))       # image decode threads next to this rank's GPU
        dist.init_process_group("nccl")
    os.environ.setdefault("DSIM_DECODE_PROCS", str(args.decode_procs))      # the scorers' DecodePool default
    scorer = None if args.selftest_shard else build_scorer(args)
    layer = args.target_layer if isinstance(args.target_layer, list) else [args.target_layer]
    if rank == 0:
        print(f"=========seed {args.seed}=========")
        print(f"Experiment on {args.target_block}, layer {args.target_layer}, timestep {args.target_step}:")
    if args.dataset == "nights":
        rows = H.read_nights_csv(args.image_path)
        trip = [(r["ref"], r["left"], r["right"], r["prompt"]) for r in rows]
    elif args.dataset == "sref":
        trip = sref_triplets(args.image_path, args.seed, args.prompt, args.experiments)
    else:
        trip = cute_triplets(args.image_path, args.seed)
    if args.selftest_shard:
        s_ab, s_ac, bad = _selftest_scores(trip, rank, world)
    else:
        s_ab, s_ac, bad = H.score_path_triplets(scorer, trip, args.image_size, args.target_block, layer, args.target_step, args.seed,
                                                args.similarity, rank, world, args.batch, args.unet_batch)
    if rank == 0:
        total = len(trip)
        if bad:
            print(f"WARNING: {bad} pair score(s) are NaN/inf (counted as wrong, as the reference's comparisons would)")
        if args.dataset == "nights":
            acc = H.nights_accuracy(s_ab, s_ac, [r["vote"] for r in rows], args.similarity)
            print(f"Final validation accuracy: {acc:.2f}%")
        else:
            correct, correct2 = cute_counts(s_ab.cpu(), s_ac.cpu(), args.similarity)
            if total > 0 and args.dataset == "sref":           # style_main.py:186-193
                print(f"Total comparisons: {total}")
                print(f"Accuracy: {correct / total * 100:.2f}%")
                print(f"2x Accuracy: {correct2 / total * 100:.2f}%")
            elif total > 0:                                     # cute_main.py:216-224
                print(f"Total comparisons: {total}")
                print(f"Total {total}; Correct {correct}; Correct 2x {correct2}")
                print(f"Accuracy: {correct / total * 100}%")
                print(f"2x Accuracy: {correct2 / total * 100}%")
            else:
                print("Total comparisons: 0")
                print("No valid comparisons were made.")
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    args = arg_parse(argv)
    if "WORLD_SIZE" in os.environ:
        # --ngpu counts the GPUs of THIS node: under a launcher that is LOCAL_WORLD_SIZE (WORLD_SIZE on one node).  A launch
        # that left --ngpu at its default adopts the launcher's count; an explicit, different --ngpu is refused as mislabelled.
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ["WORLD_SIZE"]))
        explicit = args.ngpu is not None            # (default None: every spelling argparse accepts counts as given)
        if not explicit:
            if local_world != 1:
                print(f"diffsim_amd: --ngpu not given; using the launcher's {local_world} rank(s) on this node", file=sys.stderr)
            args.ngpu = local_world
        elif local_world != args.ngpu:
            raise SystemExit(f"--ngpu {args.ngpu} but the launcher started {local_world} rank(s) on this node "
                             f"(LOCAL_WORLD_SIZE / WORLD_SIZE): refusing to run a mislabelled launch")
    if args.ngpu is None:
        args.ngpu = 1
    if args.ngpu > 1 and "WORLD_SIZE" not in os.environ:
        from .parallel import spawn_ranks              # the parent never touches the GPU
        return spawn_ranks(args.ngpu, [sys.executable, "-m", "diffsim_amd"] + argv)
    return run(args)
