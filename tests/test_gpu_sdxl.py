"""DiffSim-XL (config 4): SDXL-topology U-Net on the HIP engine vs the oracle and vs the golden scores the
REFERENCE's diffsim_xl.py + diffsim_xl_pipeline.py produced (tests/golden/make_golden_xl.py)."""
import ast
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from diffsim_amd import config as C
from diffsim_amd import synth as S

REL_F32 = 1e-4


@pytest.fixture(scope="module")
def env():
    from oracle import cpu_ref as R
    sd = S.make_state_dict(C.SDXL_TINY, seed=0)
    return dict(sd=sd, unet=R.build_unet(R.SDXL_TINY, sd), ctx=S.make_context(C.SDXL_TINY), pooled=S.make_pooled(C.SDXL_TINY), R=R)


def test_sdxl_golden_and_oracle(env, golden_dir):
    from diffsim_amd.diffsim_xl import diffsim_xl
    from tests._fakes import FakeVAE
    R, unet, ctx, pooled = env["R"], env["unet"], env["ctx"], env["pooled"]
    g = np.load(os.path.join(golden_dir, "g8_sdxl_tiny.npz"))
    xl = diffsim_xl(torch.float32, "cuda", unet_config=C.SDXL_TINY, state_dict=env["sd"], vae=FakeVAE(),
                    encode_prompt=lambda p: (ctx, pooled))
    xlb = diffsim_xl(torch.bfloat16, "cuda", unet_config=C.SDXL_TINY, state_dict=env["sd"])
    zA, zB, nA, nB = (torch.from_numpy(g[k]) for k in ("latA", "latB", "noiseA", "noiseB"))
    img_a, img_b = os.path.join(golden_dir, "g1_img_c.png"), os.path.join(golden_dir, "g1_img_d.png")
    for ci in range(6):
        blk, tl, step, sim = (str(x) for x in g[f"case_{ci}"])
        tl, step = ast.literal_eval(tl), int(step)
        want = float(g[f"score_{ci}"][0])
        so = float(R.diffsim_xl_latents(unet, zA, zB, nA, nB, ctx, pooled, step, blk, tl, sim))
        assert abs(so - want) <= 2e-5 * abs(want) + 1e-7, ("oracle vs reference", ci, so, want)
        s_lat = float(xl.score_latent_pairs(zA, zB, nA, nB, ctx, pooled, blk, tl, step, sim).cpu())
        assert abs(s_lat - want) <= REL_F32 * abs(want), (ci, s_lat, want)
        s_path = xl.diffsim_score(img_a, img_b, 128, "a cat", blk, tl, step, sim, 2334)
        assert s_path.shape == (1,) and abs(float(s_path.cpu()) - want) <= REL_F32 * abs(want), (ci, float(s_path.cpu()), want)
        s_bf = float(xlb.score_latent_pairs(zA, zB, nA, nB, ctx, pooled, blk, tl, step, sim).cpu())
        assert abs(s_bf - want) <= 3e-2, (ci, s_bf, want)
    q, k, v = xl.features(zB, nB, ctx, pooled, "up_blocks", [0, 1, 2], 600)
    for got, name in ((q, "qB"), (k, "kB"), (v, "vB")):
        want = torch.from_numpy(g[name]).transpose(1, 2).reshape(2, g[name].shape[2], -1)
        assert (got[0].float().cpu() - want).abs().max().item() <= 2e-4 * float(want.abs().max())
