"""CLIP text encoder on the HIP engine (row f-3, T/train_util.py:108-155) against the oracle (oracle/clip_ref.py, itself
pinned to `transformers` in tests/test_oracle_clip.py) and, where importable, against transformers directly."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import clip_ref as C


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def seeded(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.ndim >= 2:
                w = torch.randn(p.shape, generator=g) * (0.02 if "embedding" in name else 0.8 / p.shape[1] ** 0.5)
            elif name.endswith("weight"):
                w = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
            else:
                w = 0.02 * torch.randn(p.shape, generator=g)
            p.copy_(w.to(torch.bfloat16).to(p.dtype))  # bf16-representable: every dtype holds the same weights
    return model


def ids_for(cfg, n=3, seed=0):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(1, cfg.eos_token_id - 2, (n, cfg.max_position_embeddings), generator=g)
    ids[:, 0] = cfg.eos_token_id - 1
    for i, e in enumerate((9, 33, 76)[:n]):
        ids[i, e:] = cfg.eos_token_id
    return ids


@pytest.mark.parametrize("which", ["tiny_quick_gelu", "tiny_gelu_proj", "clip_l", "open_clip_bigg"])
# bars = 1.5 x measured (fp16: 1.49e-3 on OpenCLIP-bigG, <= 9.8e-4 on the others; bf16: 7.9e-3 on CLIP-L)
@pytest.mark.parametrize("dtype,bar", [(torch.float16, 2.25e-3), (torch.bfloat16, 1.2e-2)])
def test_clip_text_encoder_matches_oracle(which, dtype, bar):
    import sliders_conceptmod_amd.clip as PC
    if which == "tiny_quick_gelu":
        cfg, cls = PC.CLIPTextConfig(vocab_size=1000, hidden_size=64, intermediate_size=256, num_hidden_layers=3,
                                     num_attention_heads=4, eos_token_id=999), PC.CLIPTextModel
    elif which == "tiny_gelu_proj":
        cfg, cls = PC.CLIPTextConfig(vocab_size=1000, hidden_size=128, intermediate_size=512, num_hidden_layers=2,
                                     num_attention_heads=2, hidden_act="gelu", projection_dim=64,
                                     eos_token_id=999), PC.CLIPTextModelWithProjection
    elif which == "clip_l":
        cfg, cls = PC.clip_l_config(), PC.CLIPTextModel
    else:
        if dtype == torch.bfloat16:
            pytest.skip("bigG once (fp16) is enough")
        cfg, cls = PC.open_clip_bigg_config(), PC.CLIPTextModelWithProjection
    torch.set_num_threads(16)
    m = seeded(cls(cfg), 3)
    ids = ids_for(cfg)
    ref = C.clip_text_forward({k: v.float() for k, v in m.state_dict().items()}, ids, cfg.num_attention_heads,
                              cfg.hidden_act, cfg.eos_token_id)
    m = m.to("cuda", dtype)
    out = m(ids.cuda(), output_hidden_states=True)
    e_last, e_pen = rel(out.last_hidden_state, ref["last_hidden_state"]), rel(out.hidden_states[-2], ref["penultimate"])
    if cls is PC.CLIPTextModelWithProjection:
        e_pool = rel(out[0], ref["text_embeds"])
        assert out[0].shape == (3, cfg.projection_dim) and out.text_embeds is out[0]
    else:
        e_pool = rel(out.pooler_output, ref["pooler_output"])
        assert out[0] is out.last_hidden_state
    print(f"{which} {dtype}: last {e_last:.2e}, hidden_states[-2] {e_pen:.2e}, pooled {e_pool:.2e}")
    assert max(e_last, e_pen, e_pool) < bar, (e_last, e_pen, e_pool)
    # causal: changing a LATER token must not change earlier positions
    ids2 = ids.clone()
    ids2[:, 5] = (ids2[:, 5] + 7) % (cfg.eos_token_id - 2) + 1
    out2 = m(ids2.cuda(), output_hidden_states=True)
    assert torch.equal(out2.last_hidden_state[:, :5], out.last_hidden_state[:, :5])
    assert not torch.equal(out2.last_hidden_state[:, 5:], out.last_hidden_state[:, 5:])
    if which == "tiny_quick_gelu":  # the engine always runs max_positions tokens; anything else must raise
        from sliders_conceptmod_amd import _native
        with pytest.raises(_native.SmiError, match="padded to"):
            m(ids[:, :10].cuda())


def test_clip_matches_transformers_directly():
    transformers = pytest.importorskip("transformers")
    import sliders_conceptmod_amd.clip as PC
    hf_cfg = transformers.CLIPTextConfig(vocab_size=1000, hidden_size=64, intermediate_size=256, num_hidden_layers=3,
                                         num_attention_heads=4, hidden_act="gelu", projection_dim=32, eos_token_id=999,
                                         bos_token_id=998, pad_token_id=0)
    hf = seeded(transformers.CLIPTextModelWithProjection(hf_cfg), 5).eval()
    cfg = PC.CLIPTextConfig(vocab_size=1000, hidden_size=64, intermediate_size=256, num_hidden_layers=3,
                            num_attention_heads=4, hidden_act="gelu", projection_dim=32, eos_token_id=999)
    p = PC.CLIPTextModelWithProjection(cfg)
    p.load_state_dict({k: v for k, v in hf.state_dict().items()})
    ids = ids_for(cfg, seed=4)
    with torch.no_grad():
        ref = hf(ids, output_hidden_states=True)
    out = p.to("cuda", torch.float16)(ids.cuda(), output_hidden_states=True)
    assert rel(out[0], ref[0]) < 3e-3 and rel(out.hidden_states[-2], ref.hidden_states[-2]) < 3e-3


def test_sdxl_prompt_embeddings_through_the_reference_call_pattern():
    """train_lora_xl.encode_xl with the two native encoders and a stub tokenizer: penultimate states of both encoders
    concatenated, pooled from the second (T/train_util.py:128-155, T/train_lora_xl.py:121-154)."""
    import sliders_conceptmod_amd.clip as PC
    from sliders_conceptmod_amd.train_lora_xl import encode_xl
    c1 = PC.CLIPTextConfig(vocab_size=1000, hidden_size=64, intermediate_size=256, num_hidden_layers=2,
                           num_attention_heads=4, eos_token_id=999)
    c2 = PC.CLIPTextConfig(vocab_size=1000, hidden_size=128, intermediate_size=512, num_hidden_layers=2,
                           num_attention_heads=2, hidden_act="gelu", projection_dim=64, eos_token_id=999)
    e1 = seeded(PC.CLIPTextModel(c1), 1).to("cuda", torch.float16)
    e2 = seeded(PC.CLIPTextModelWithProjection(c2), 2).to("cuda", torch.float16)

    class Tok:
        model_max_length = 77

        def __call__(self, prompt, **kw):
            g = torch.Generator().manual_seed(len(prompt))
            ids = torch.randint(1, 990, (1, 77), generator=g)
            ids[0, 0], ids[0, 1 + len(prompt):] = 998, 999

            class R:
                input_ids = ids
            return R()

    pe = encode_xl([e1, e2], [Tok(), Tok()], "a photo of a person", torch.device("cuda"), torch.float16)
    assert pe.text_embeds.shape == (1, 77, 64 + 128) and pe.pooled_embeds.shape == (1, 64)
    assert torch.isfinite(pe.text_embeds.float()).all() and float(pe.pooled_embeds.float().abs().max()) > 0
