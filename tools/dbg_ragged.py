import sys, torch
sys.path.insert(0, '.')
from diffsim_amd import config as C, synth as S
from diffsim_amd.diffsim import DiffSim
cfg = C.SD15
shapes = C.unet_param_shapes(cfg)
sd = S.make_state_dict(cfg, seed=0, keys=[k for k in shapes if not k.startswith(("conv_norm_out", "conv_out"))])
ctx = S.make_context(cfg)
for side in (28, 26, 30):
    g = torch.Generator("cpu").manual_seed(2800)
    zA, zB = (torch.randn((32, 4, side, side), generator=g) for _ in range(2))
    nA, nB = (torch.randn((1, 4, side, side), generator=g) for _ in range(2))
    for dtype in (torch.float32, torch.bfloat16):
        ds = DiffSim(torch_dtype=dtype, device="cuda", unet_config=cfg, state_dict=sd, dedup_cfg=(len(sys.argv) > 1))
        for t in (("down_blocks", 0), ("down_blocks", 1), ("down_blocks", 2), ("mid_blocks", 0), ("up_blocks", 0), ("up_blocks", 1), ("up_blocks", 2)):
            for n in (1, 2, 8, 32):
                try:
                    s = ds.score_latent_pairs(zA[:n], zB[:n], nA, nB, ctx, t[0], t[1], 600, "cosine", batch_pairs=n)
                    torch.cuda.synchronize()
                    r = "ok %.5f" % float(s[0])
                except Exception as e:
                    r = "FAIL " + str(e)[:80]
                print(side, dtype, t, n, r, flush=True)
        del ds
        torch.cuda.empty_cache()
