"""ORACLE (test infrastructure, not shipped): CPU restatement of the CLIP text encoder the reference calls through
`transformers` (conceptmod/textsliders/train_util.py:108-155; requirements.txt:21 pins transformers==4.27.4).

PINNED against the dependency itself: `transformers` IS importable in this image (5.x), and tests/test_oracle_clip.py
checks this restatement against `transformers.CLIPTextModel` / `CLIPTextModelWithProjection` on seeded random-init
models (same state dict, same token ids): last hidden state, hidden_states[-2], pooled output and text_embeds.
Plain fp32 PyTorch; used only by tests/."""
import torch
import torch.nn.functional as F


def clip_text_forward(sd: dict, ids: torch.Tensor, num_heads: int, hidden_act: str = "quick_gelu", eos_token_id: int = 49407):
    """sd: transformers CLIPTextModel[WithProjection] state dict (fp32).  Returns dict(last_hidden_state,
    penultimate (= hidden_states[-2]), pooler_output, text_embeds or None)."""
    g = lambda k: sd[k].float()
    n, L = ids.shape
    x = g("text_model.embeddings.token_embedding.weight")[ids] + g("text_model.embeddings.position_embedding.weight")[:L]
    d = x.shape[-1]
    hd = d // num_heads
    nl = 1 + max(int(k.split(".")[3]) for k in sd if k.startswith("text_model.encoder.layers."))
    mask = torch.full((L, L), float("-inf")).triu(1)
    pen = None
    for i in range(nl):
        b = f"text_model.encoder.layers.{i}."
        if i == nl - 1:
            pen = x
        h = F.layer_norm(x, (d,), g(b + "layer_norm1.weight"), g(b + "layer_norm1.bias"), 1e-5)
        q = F.linear(h, g(b + "self_attn.q_proj.weight"), g(b + "self_attn.q_proj.bias"))
        k = F.linear(h, g(b + "self_attn.k_proj.weight"), g(b + "self_attn.k_proj.bias"))
        v = F.linear(h, g(b + "self_attn.v_proj.weight"), g(b + "self_attn.v_proj.bias"))
        q, k, v = (t.view(n, L, num_heads, hd).transpose(1, 2) for t in (q, k, v))
        p = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5 + mask, dim=-1)
        o = (p @ v).transpose(1, 2).reshape(n, L, d)
        x = x + F.linear(o, g(b + "self_attn.out_proj.weight"), g(b + "self_attn.out_proj.bias"))
        h = F.layer_norm(x, (d,), g(b + "layer_norm2.weight"), g(b + "layer_norm2.bias"), 1e-5)
        h = F.linear(h, g(b + "mlp.fc1.weight"), g(b + "mlp.fc1.bias"))
        h = h * torch.sigmoid(1.702 * h) if hidden_act == "quick_gelu" else F.gelu(h)
        x = x + F.linear(h, g(b + "mlp.fc2.weight"), g(b + "mlp.fc2.bias"))
    last = F.layer_norm(x, (d,), g("text_model.final_layer_norm.weight"), g("text_model.final_layer_norm.bias"), 1e-5)
    eos_pos = ids.argmax(dim=-1) if eos_token_id == 2 else (ids == eos_token_id).int().argmax(dim=-1)
    pooled = last[torch.arange(n), eos_pos]
    te = F.linear(pooled, g("text_projection.weight")) if "text_projection.weight" in sd else None
    return {"last_hidden_state": last, "penultimate": pen, "pooler_output": pooled, "text_embeds": te}
