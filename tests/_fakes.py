"""Deterministic stand-ins shared by the golden-capture script and the parity tests."""
import types

import torch


class FakeVAE:
    """``encode(x).latent_dist.sample(generator)`` = pooled-mix(x) + noise drawn from the
    caller's generator -- consumes generator state exactly where the real
    ``AutoencoderKL.encode(...).latent_dist.sample`` does (reference: diffsim/diffsim.py:92-96).
    The real VAE encoder is a "next" row."""
    config = types.SimpleNamespace(scaling_factor=0.18215)

    def encode(self, x):
        x = x.float()
        pooled = torch.nn.functional.avg_pool2d(x, 8)                       # (1,3,s,s)
        mix = torch.tensor([[1.0, 0.5, -0.5], [0.25, -1.0, 0.75], [-0.5, 0.5, 1.0],
                            [0.6, 0.6, 0.6]])
        mean = torch.einsum("oc,bchw->bohw", mix, pooled) * (1.0 / 0.18215) * 1.5

        class _D:
            def sample(self, generator=None):
                return mean + (0.1 / 0.18215) * torch.randn(mean.shape, generator=generator)
        return types.SimpleNamespace(latent_dist=_D(), mean=mean)


class FakeVAE16(FakeVAE):
    """The same stand-in as an fp16 module (the reference's SD1.5 VAE runs in fp16, diffsim/diffsim.py:93): the moments
    come out in fp16 and ``sample`` draws ``randn_tensor(dtype=float16)`` and does its arithmetic in fp16, as
    diffusers' DiagonalGaussianDistribution does for fp16 parameters."""

    def encode(self, x):
        base = FakeVAE.encode(self, x)
        mean = base.mean.to(torch.float16)
        std = torch.tensor(0.1 / 0.18215, dtype=torch.float16)

        class _D:
            def sample(self, generator=None):
                return mean + std * torch.randn(mean.shape, generator=generator, dtype=torch.float16)
        return types.SimpleNamespace(latent_dist=_D(), mean=mean)
