"""Pins oracle/clip_ref.py against the reference's real dependency: `transformers.CLIPTextModel` and
`CLIPTextModelWithProjection` (what T/train_util.py:119-120, 139-144 call) on seeded random-init models.  CPU only."""
import pytest
import torch

from oracle import clip_ref as C

transformers = pytest.importorskip("transformers")


def tiny_cfg(act, proj=None):
    return transformers.CLIPTextConfig(vocab_size=1000, hidden_size=64, intermediate_size=256, num_hidden_layers=3,
                                       num_attention_heads=4, max_position_embeddings=77, hidden_act=act,
                                       projection_dim=proj or 512, eos_token_id=999, bos_token_id=998, pad_token_id=0)


def classic(sd):
    """Checkpoint key names (`text_model.` prefix), which transformers >= 5 drops from CLIPTextModel's own state dict."""
    return {(k if k.startswith(("text_model.", "text_projection")) else "text_model." + k): v for k, v in sd.items()}


def token_ids(n=3, seed=0):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(1, 990, (n, 77), generator=g)
    ids[:, 0] = 998
    for i, e in enumerate((10, 30, 76)[:n]):
        ids[i, e:] = 999
    return ids


@pytest.mark.parametrize("act", ["quick_gelu", "gelu"])
def test_clip_text_model_matches_transformers(act):
    torch.manual_seed(0)
    m = transformers.CLIPTextModel(tiny_cfg(act)).eval()
    ids = token_ids()
    with torch.no_grad():
        ref = m(ids, output_hidden_states=True)
    got = C.clip_text_forward(classic(m.state_dict()), ids, 4, act, eos_token_id=999)
    torch.testing.assert_close(got["last_hidden_state"], ref[0], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(got["penultimate"], ref.hidden_states[-2], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(got["pooler_output"], ref.pooler_output, rtol=1e-4, atol=1e-5)


def test_clip_text_model_with_projection_matches_transformers():
    torch.manual_seed(1)
    m = transformers.CLIPTextModelWithProjection(tiny_cfg("gelu", proj=32)).eval()
    ids = token_ids(seed=2)
    with torch.no_grad():
        ref = m(ids, output_hidden_states=True)
    got = C.clip_text_forward(classic(m.state_dict()), ids, 4, "gelu", eos_token_id=999)
    torch.testing.assert_close(got["text_embeds"], ref[0], rtol=1e-4, atol=1e-5)  # what text_encode_xl pools
    torch.testing.assert_close(got["penultimate"], ref.hidden_states[-2], rtol=1e-4, atol=1e-5)


def test_product_container_loads_a_transformers_state_dict_and_fails_loudly_off_gpu():
    import sliders_conceptmod_amd.clip as PC
    from sliders_conceptmod_amd._native import SmiError
    m = transformers.CLIPTextModelWithProjection(tiny_cfg("gelu", proj=32))
    cfg = PC.CLIPTextConfig(vocab_size=1000, hidden_size=64, intermediate_size=256, num_hidden_layers=3,
                            num_attention_heads=4, hidden_act="gelu", projection_dim=32, eos_token_id=999)
    p = PC.CLIPTextModelWithProjection(cfg)
    assert set(p.state_dict().keys()) == {k for k in classic(m.state_dict()) if not k.endswith("position_ids")}
    p.load_state_dict(classic(m.state_dict()))
    with pytest.raises(SmiError):
        p.half()(token_ids())
    big = PC.CLIPTextModel(PC.clip_l_config())
    assert sum(q.numel() for q in big.parameters()) == 123060480  # openai/clip-vit-large-patch14 text tower
