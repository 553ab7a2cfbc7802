"""Host-side scheduler arithmetic of the one-step pipeline.

Mirrors what ``DiffSimPipeline.step`` gets from diffusers' PNDMScheduler with SD1.5's
scheduler_config (reference call sites: diffsim/diffsim_pipeline.py:153-157 ``timesteps[i]``
and :177-183 ``scheduler.add_noise``; semantics: SURVEY.md Appendix A item 10).
"""
from __future__ import annotations

import numpy as np
import torch


def alphas_cumprod(num_train_timesteps: int = 1000, beta_start: float = 0.00085,
                   beta_end: float = 0.012) -> torch.Tensor:
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                           dtype=torch.float32) ** 2          # "scaled_linear"
    return torch.cumprod(1.0 - betas, dim=0)


def pndm_timesteps(num_inference_steps: int = 1000, num_train_timesteps: int = 1000,
                   steps_offset: int = 1) -> np.ndarray:
    """``set_timesteps`` with skip_prk_steps: [1000, 999, 999, 998, ..., 1] (N+1 entries)."""
    ratio = num_train_timesteps // num_inference_steps
    t = (np.arange(0, num_inference_steps) * ratio).round() + steps_offset
    return np.concatenate([t[:-1], t[-2:-1], t[-1:]])[::-1].astype(np.int64).copy()


def timestep_from_index(target_step: int, num_inference_steps: int = 1000) -> int:
    """--target_step is an INDEX into the table above (600 -> t = 401)."""
    ts = pndm_timesteps(num_inference_steps)
    if not 0 <= target_step < len(ts):
        raise IndexError(f"target_step {target_step} outside the {len(ts)}-entry timestep table")
    t = int(ts[target_step])
    if t >= 1000:
        # index 0 -> t = 1000 indexes past alphas_cumprod in the reference (IndexError there too)
        raise IndexError("target_step 0 maps to t=1000, outside the 1000-entry alphas_cumprod table")
    return t


def noise_coefficients(t: int):
    ac = alphas_cumprod()
    return float(ac[t] ** 0.5), float((1.0 - ac[t]) ** 0.5)


# ---- SDXL: EulerDiscreteScheduler with SDXL's scheduler_config ("leading" spacing, steps_offset 1) --------
# reference call sites: diffsim/diffsim_xl_pipeline.py:190-225 (timesteps[i], prepare_latents * init_noise_sigma,
# add_noise) and :309 (scale_model_input); semantics restated from SURVEY.md Appendix A item 14.
def euler_tables(num_inference_steps: int = 1000, num_train_timesteps: int = 1000, steps_offset: int = 1):
    ac = alphas_cumprod().numpy().astype(np.float64)
    sig_all = ((1 - ac) / ac) ** 0.5
    ratio = num_train_timesteps // num_inference_steps
    ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.float32) + steps_offset
    sig = np.interp(ts, np.arange(0, len(sig_all)), sig_all)
    sig = np.concatenate([sig, [0.0]]).astype(np.float32)
    return ts, sig, float((sig.max() ** 2 + 1) ** 0.5)


def sdxl_step_coefficients(target_step: int):
    """(t, a, b) with x_in = a*z0 + b*eps: the reference multiplies the CLEAN latents by init_noise_sigma,
    adds sigma*eps, then divides by sqrt(sigma^2+1) (quirk reproduced)."""
    ts, sig, init = euler_tables()
    if not 0 <= target_step < len(ts):
        raise IndexError(f"target_step {target_step} outside the {len(ts)}-entry timestep table")
    s = float(sig[target_step])
    d = (s * s + 1.0) ** 0.5
    return int(ts[target_step]), init / d, s / d


# ---- DiT: timestep respacing of the vendored OpenAI-style diffusion ------------------------------------
def dit_timestep_map(target_step: int, num_timesteps: int = 1000):
    """``create_diffusion(str(target_step)).timestep_map`` (DiT/diffusion/respace.py:12-88): target_step
    evenly spaced original timesteps, as a sorted list."""
    n = int(target_step)
    stride = 1 if n <= 1 else (num_timesteps - 1) / (n - 1)
    return sorted({round(i * stride) for i in range(n)})


def dit_model_timestep(target_step: int) -> int:
    """The model is called with t = 1000 - target_step remapped through the spaced schedule
    (diffsim/diffsim_dit.py:105; respace.py:117-129) -- noise is added at t = target_step but the network is
    conditioned on timestep_map[1000 - target_step] (quirk reproduced; needs target_step > 500)."""
    m = dit_timestep_map(target_step)
    i = 1000 - int(target_step)
    if not 0 <= i < len(m):
        raise IndexError(f"target_step {target_step}: index {i} outside the {len(m)}-entry spaced schedule")
    return int(m[i])
