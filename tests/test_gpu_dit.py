"""DiffSim-DiT (config 5): DiT backbone on the HIP engine vs the oracle (fp32 gate) and vs the golden the
REFERENCE's diffsim_dit.py + vendored DiT/modelsdit.py produced in fp16 (structure pin, fp16-sized tolerance)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from diffsim_amd import config as C
from diffsim_amd import synth as S


def test_dit_golden_and_oracle(golden_dir):
    from oracle import cpu_ref as R
    from diffsim_amd.diffsim_dit import diffsim_DiT
    from tests._fakes import FakeVAE
    sd = S.make_state_dict(C.DIT_TINY, seed=0)
    m = R.DiTOracle(R.DIT_TINY)
    m.load_state_dict(sd, strict=True)
    m.eval()
    g = np.load(os.path.join(golden_dir, "g9_dit_tiny.npz"))
    zA, zB, nA, nB = (torch.from_numpy(g[k]) for k in ("latA", "latB", "noiseA", "noiseB"))
    dd = diffsim_DiT(128, 600, "cuda", dit_config=C.DIT_TINY, state_dict=sd, vae=FakeVAE(), torch_dtype=torch.float32)
    ddb = diffsim_DiT(128, 600, "cuda", dit_config=C.DIT_TINY, state_dict=sd, torch_dtype=torch.bfloat16)
    img_a, img_b = os.path.join(golden_dir, "g1_img_c.png"), os.path.join(golden_dir, "g1_img_d.png")
    for ci in range(4):
        layer, step, sim = (str(x) for x in g[f"case_{ci}"])
        layer, step = int(layer), int(step)
        ref16 = float(g[f"score_{ci}"][0])                                   # reference code, fp16 arithmetic
        so = float(R.diffsim_dit_latents(m, zA, zB, nA, nB, step, layer, sim))
        assert abs(so - ref16) <= 2e-3 * max(abs(ref16), 0.05), ("oracle vs reference(fp16)", ci, so, ref16)
        s = float(dd.score_latent_pairs(zA, zB, nA, nB, layer, step, sim).cpu())
        assert abs(s - so) <= 1e-4 * abs(so), (ci, s, so)                    # fp32 gate vs the oracle
        sp = dd.diffsim_score(img_a, img_b, 128, "p", "none", [layer], step, sim, 2334)
        assert sp.shape == (1,) and abs(float(sp.cpu()) - so) <= 1e-4 * abs(so), (ci, float(sp.cpu()), so)
        sb = float(ddb.score_latent_pairs(zA, zB, nA, nB, layer, step, sim).cpu())
        assert abs(sb - so) <= 3e-2, (ci, sb, so)
    q, k, v = dd.features(zB, nB, 2, 600)
    qo, ko, vo = R.dit_features(m, zB, nB, 600, 2)
    for got, want, name in ((q, qo, "qB"), (k, ko, "kB"), (v, vo, "vB")):
        w = want.transpose(1, 2).reshape(2, want.shape[2], -1)
        assert (got[0].float().cpu() - w).abs().max().item() <= 2e-4 * float(w.abs().max())
        r16 = torch.from_numpy(g[name]).transpose(1, 2).reshape(2, g[name].shape[2], -1)
        assert (got[0].float().cpu() - r16).abs().max().item() <= 4e-3 * float(r16.abs().max())     # fp16 reference
    with pytest.raises(IndexError):
        dd.score_latent_pairs(zA, zB, nA, nB, 0, 400)                       # 1000-400 outside the 400-entry schedule


def test_dit_fp8_attention_mode(golden_dir):
    """BASELINE config 5: DiT with fp8 (e4m3) MFMA attention in every block below the tap.  Opt-in; compared with the
    fp32 oracle under a stated fp8-sized tolerance: |score error| <= 3e-2 (the bf16 mode's bound on this config) and
    q/k/v of the tapped block within 6 % of their range; it must differ from the bf16 mode (i.e. really run fp8)."""
    from oracle import cpu_ref as R
    from diffsim_amd import _lib
    from diffsim_amd.diffsim_dit import diffsim_DiT
    sd = S.make_state_dict(C.DIT_TINY, seed=0)
    m = R.DiTOracle(R.DIT_TINY)
    m.load_state_dict(sd, strict=True)
    m.eval()
    g = np.load(os.path.join(golden_dir, "g9_dit_tiny.npz"))
    zA, zB, nA, nB = (torch.from_numpy(g[k]) for k in ("latA", "latB", "noiseA", "noiseB"))
    d8 = diffsim_DiT(128, 600, "cuda", dit_config=C.DIT_TINY, state_dict=sd, torch_dtype=torch.bfloat16, fp8_attention=True)
    d16 = diffsim_DiT(128, 600, "cuda", dit_config=C.DIT_TINY, state_dict=sd, torch_dtype=torch.bfloat16)
    for layer, step, sim in ((2, 600, "cosine"), (1, 700, "cosine"), (2, 600, "mse")):
        so = float(R.diffsim_dit_latents(m, zA, zB, nA, nB, step, layer, sim))
        s8 = float(d8.score_latent_pairs(zA, zB, nA, nB, layer, step, sim).cpu())
        s16 = float(d16.score_latent_pairs(zA, zB, nA, nB, layer, step, sim).cpu())
        assert abs(s8 - so) <= 3e-2 * max(1.0, abs(so)), (layer, step, sim, s8, so)
        assert s8 != s16
    q8, k8, v8 = d8.features(zB, nB, 2, 600)
    qo, ko, vo = R.dit_features(m, zB, nB, 600, 2)
    for got, want in ((q8, qo), (k8, ko), (v8, vo)):
        w = want.transpose(1, 2).reshape(2, want.shape[2], -1)
        assert (got[0].float().cpu() - w).abs().max().item() <= 6e-2 * float(w.abs().max())
    with pytest.raises(_lib.DsimError):                                      # fp32 handles have no fp8 mode
        diffsim_DiT(128, 600, "cuda", dit_config=C.DIT_TINY, state_dict=sd, torch_dtype=torch.float32,
                    fp8_attention=True).score_latent_pairs(zA, zB, nA, nB, 2, 600, "cosine")
