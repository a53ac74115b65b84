"""Plain-PyTorch fp32 CPU restatement of the scoring path — TEST / BASELINE INFRASTRUCTURE ONLY.

Second, independent restatement next to oracle/np_oracle.py (numpy): the same arithmetic written with torch CPU ops, so that
  * bench.py's `cpu_baseline` leg can time what BASELINE.md section 3 specifies — "our own plain-PyTorch fp32 restatement of
    OutfitX._cp_forward ... torch.set_num_threads(os.cpu_count()) and a 1-thread run, 3 warm-up + >= 10 timed iterations,
    median" — on the GPU box's host cores (the reference's files never travel there);
  * tests/studies/operand_scheme_cpu.py can emulate MFMA operand roundings on top of it (the `rnd` hook).
Nothing under outfitx_amd/ imports it.  Parity status: PINNED — tests/test_oracle_golden.py holds it to the golden vectors the
reference itself produced (tests/golden/*.npz, oracle/gen_golden.py) at the same 2e-5 as the numpy oracle.

Reference call sites restated: src/models/outfit_x.py:120-144 (_cp_forward over nn.TransformerEncoder built at :32-45, pre-norm,
Mish, no final norm), src/models/encoders/item_encoder.py:46-61 + base_*_encoder.py (towers, F.normalize per modality, concat),
HF CLIPVisionModelWithProjection / CLIPTextModelWithProjection reached from clip_image_encoder.py:74-76 and
clip_text_encoder.py:56-58 (pre-LN blocks, quick_gelu, causal AND key-padding mask, EOS pooling).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

EOS_ID = 49407


class TorchRef:
    """W: dict name -> fp32 ndarray with the reference's state_dict key names (relative to the sub-model, as synth.* returns them).
    `rnd(x, side, wname)` (default identity) is applied to both operands of every dense contraction: side 'a' | 'w'."""

    def __init__(self, W, rnd=None, rnd_attn=None):
        self.W = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in W.items()}
        self.rnd = rnd or (lambda x, side, name: x)
        self.rnd_attn = rnd_attn or (lambda x: x)
        self._wc = {}

    # ---- primitives
    def lin(self, x, wname, bname=None):
        w = self._wc.get(wname)
        if w is None:
            w = self._wc[wname] = self.rnd(self.W[wname].reshape(self.W[wname].shape[0], -1), "w", wname)
        y = self.rnd(x, "a", wname) @ w.T
        return y if bname is None else y + self.W[bname]

    def ln(self, x, p):
        return F.layer_norm(x, x.shape[-1:], self.W[p + ".weight"], self.W[p + ".bias"], 1e-5)

    def mha(self, q, k, v, dead, n_head):
        B, S, D = q.shape
        dh = D // n_head
        q, k, v = [self.rnd_attn(t).view(B, S, n_head, dh).transpose(1, 2) for t in (q, k, v)]
        s = (q @ k.transpose(-1, -2)) * dh ** -0.5
        p = self.rnd_attn(torch.softmax(s.masked_fill(dead, float("-inf")), -1))
        return (p @ v).transpose(1, 2).reshape(B, S, D)

    # ---- CLIP towers (HF CLIPEncoderLayer: pre-LN, quick_gelu)
    def clip_layers(self, x, dead, prefix, n_layers, n_head):
        for i in range(n_layers):
            p = f"{prefix}encoder.layers.{i}."
            h = self.ln(x, p + "layer_norm1")
            q, k, v = [self.lin(h, p + f"self_attn.{n}_proj.weight", p + f"self_attn.{n}_proj.bias") for n in "qkv"]
            x = x + self.lin(self.mha(q, k, v, dead, n_head), p + "self_attn.out_proj.weight", p + "self_attn.out_proj.bias")
            u = self.lin(self.ln(x, p + "layer_norm2"), p + "mlp.fc1.weight", p + "mlp.fc1.bias")
            x = x + self.lin(u * torch.sigmoid(1.702 * u), p + "mlp.fc2.weight", p + "mlp.fc2.bias")
        return x

    def vit(self, px, n_layers=12, n_head=12):
        """[N,3,224,224] fp32 pixel_values -> UN-normalised image_embeds [N,512]."""
        N = px.shape[0]
        patch = self.W["vision_model.embeddings.patch_embedding.weight"].shape[-1]
        g = px.shape[-1] // patch
        pt = px.view(N, 3, g, patch, g, patch).permute(0, 2, 4, 1, 3, 5).reshape(N, g * g, 3 * patch * patch)
        pe = self.lin(pt, "vision_model.embeddings.patch_embedding.weight")
        cls = self.W["vision_model.embeddings.class_embedding"].expand(N, 1, -1)
        x = torch.cat([cls, pe], 1) + self.W["vision_model.embeddings.position_embedding.weight"][None]
        x = self.ln(x, "vision_model.pre_layrnorm")
        dead = torch.zeros(1, 1, 1, x.shape[1], dtype=torch.bool)
        x = self.clip_layers(x, dead, "vision_model.", n_layers, n_head)
        return self.lin(self.ln(x[:, 0], "vision_model.post_layernorm"), "visual_projection.weight")

    def text(self, ids, att, n_layers=12, n_head=8, eos_token_id=EOS_ID):
        """ids / attention_mask [N,T] int64 -> UN-normalised text_embeds [N,512]."""
        N, T = ids.shape
        x = self.W["text_model.embeddings.token_embedding.weight"][ids] + self.W["text_model.embeddings.position_embedding.weight"][:T][None]
        dead = (att == 0)[:, None, None, :] | torch.triu(torch.ones(T, T, dtype=torch.bool), 1)[None, None]
        x = self.ln(self.clip_layers(x, dead, "text_model.", n_layers, n_head), "text_model.final_layer_norm")
        eos = ids.argmax(-1) if eos_token_id == 2 else (ids == eos_token_id).int().argmax(-1)
        return self.lin(x[torch.arange(N), eos], "text_projection.weight")

    # ---- outfit transformer + CP head (outfit_x.py:120-144)
    def cp(self, emb, mask, n_layers=6, n_head=16):
        """outfit_embedding [B,L,1024], outfit_mask [B,L] bool (True = pad) -> raw logits [B,1]."""
        B, _, D = emb.shape
        x = torch.cat([self.W["outfit_token"].expand(B, 1, -1), emb], 1)
        dead = torch.cat([torch.zeros(B, 1, dtype=torch.bool), mask], 1)[:, None, None, :]
        for i in range(n_layers):
            p = f"transformer_encoder.layers.{i}."
            q, k, v = self.lin(self.ln(x, p + "norm1"), p + "self_attn.in_proj_weight", p + "self_attn.in_proj_bias").split(D, -1)
            x = x + self.lin(self.mha(q, k, v, dead, n_head), p + "self_attn.out_proj.weight", p + "self_attn.out_proj.bias")
            u = self.lin(self.ln(x, p + "norm2"), p + "linear1.weight", p + "linear1.bias")
            x = x + self.lin(F.mish(u), p + "linear2.weight", p + "linear2.bias")
        return self.lin(x[:, 0], "cp_ffn.1.weight", "cp_ffn.1.bias")


def l2n(x):
    return x / x.norm(dim=-1, keepdim=True).clamp_min(1e-12)


def item_encoder(vit_ref: TorchRef, txt_ref: TorchRef, px, ids, att):
    """item_encoder.py:46-61 on tensor inputs: px [B,L,3,H,W], ids / att [B,L,T] -> [B,L,1024] (each modality L2-normalised, concat)."""
    B, L = px.shape[:2]
    img = vit_ref.vit(px.reshape(B * L, *px.shape[2:]))
    txt = txt_ref.text(ids.reshape(B * L, -1), att.reshape(B * L, -1))
    return torch.cat([l2n(img), l2n(txt)], -1).view(B, L, -1)
