"""CLIP text encoder against the third-party implementation the reference calls
(``transformers`` CLIPTextModel / CLIPTextModelWithProjection, /root/reference/diffsim/diffsim_pipeline.py:125-135):
transformers is installed in this image, so this is a direct check against the real dependency with random-init
weights (tolerance 1e-5 absolute on O(1) activations, fp32 CPU)."""
import pytest
import torch

from diffsim_amd import text as T

transformers = pytest.importorskip("transformers")


def _hf(cfg: T.CLIPTextConfig, with_proj: bool):
    hc = transformers.CLIPTextConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size,
                                     intermediate_size=cfg.intermediate_size, num_hidden_layers=cfg.num_layers,
                                     num_attention_heads=cfg.num_heads, max_position_embeddings=cfg.max_positions,
                                     hidden_act=cfg.act, projection_dim=cfg.projection_dim or 32,
                                     bos_token_id=0, eos_token_id=2, pad_token_id=1)
    torch.manual_seed(0)
    cls = transformers.CLIPTextModelWithProjection if with_proj else transformers.CLIPTextModel
    return cls(hc).eval()


def _ids(cfg, B=2, L=77):
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(3, cfg.vocab_size - 1, (B, L), generator=g)
    ids[0, 9:] = cfg.vocab_size - 1          # EOS (highest id, as in the CLIP vocabulary) + padding with EOS
    ids[1, 30:] = cfg.vocab_size - 1
    return ids


@pytest.mark.parametrize("act", ["quick_gelu", "gelu"])
def test_matches_transformers(act):
    import dataclasses
    cfg = dataclasses.replace(T.CLIP_TINY, act=act)
    m = _hf(cfg, True)
    enc = T.CLIPTextEncoder(cfg, m.state_dict(), device="cpu")
    ids = _ids(cfg)
    with torch.no_grad():
        o = m(ids, output_hidden_states=True)
    mine = enc(ids)
    assert torch.allclose(mine["last_hidden_state"], o.last_hidden_state, atol=1e-5)
    assert len(mine["hidden_states"]) == len(o.hidden_states)
    for a, b in zip(mine["hidden_states"], o.hidden_states):
        assert torch.allclose(a, b, atol=1e-5)
    assert torch.allclose(mine["text_embeds"], o.text_embeds, atol=1e-5)


def test_prefix_and_errors():
    cfg = dataclasses_replace_noproj()
    m = _hf(cfg, False)
    sd = {"text_model." + k if not k.startswith("text_model.") else k: v for k, v in m.state_dict().items()}
    enc = T.CLIPTextEncoder(cfg, sd, device="cpu")
    ids = _ids(cfg)
    with torch.no_grad():
        o = m(ids)
    assert torch.allclose(enc(ids)["pooled"], o.pooler_output, atol=1e-5)
    bad = dict(sd)
    bad.pop("text_model.final_layer_norm.weight")
    with pytest.raises(KeyError):
        T.CLIPTextEncoder(cfg, bad, device="cpu")
    with pytest.raises(ValueError):
        enc(torch.zeros(1, 78, dtype=torch.long))


def dataclasses_replace_noproj():
    import dataclasses
    return dataclasses.replace(T.CLIP_TINY, projection_dim=0)


def test_encode_prompt_layouts():
    cfg1, cfg2 = dataclasses_replace_noproj(), T.CLIP_TINY
    e1 = T.CLIPTextEncoder(cfg1, _hf(cfg1, False).state_dict(), device="cpu")
    e2 = T.CLIPTextEncoder(cfg2, _hf(cfg2, True).state_dict(), device="cpu")
    tok = lambda s: torch.tensor([[0] + [3 + (ord(c) % 900) for c in s][:75] + [999] * (76 - min(len(s), 75))])
    ctx = T.make_encode_prompt(e1, tok)("a cat")
    assert ctx.shape == (2, 77, 64)
    assert torch.equal(ctx[0], e1(tok(""))["last_hidden_state"][0])       # [uncond, cond]
    c2, p2 = T.make_encode_prompt_xl(e1, e2, tok, tok)("a cat")
    assert c2.shape == (2, 77, 128) and p2.shape == (2, 32)
    assert float(c2[0].abs().max()) == 0 and float(p2[0].abs().max()) == 0  # force_zeros_for_empty_prompt
    assert torch.equal(c2[1, :, :64], e1(tok("a cat"))["hidden_states"][-2][0])
