#!/usr/bin/env python3
"""CPU emulation of MFMA operand schemes for the cfg2 path (towers -> fuser -> set transformer -> CP logit).

Every dense contraction is emulated as  fp32_matmul(round_A(a), round_W(w))  (products of two <= 11-bit operands are exact in
fp32 and the MFMA accumulates in fp32, so this is what the matrix core returns up to summation order); everything else is fp32.
A scheme names the operand rounding per tower and side:

    f32    exact                         (reference)
    f16    one f16 operand               (11 significant bits)
    bf16   one bf16 operand              (8 bits)
    f16x2  hi + lo, both f16             (22 bits: a second MFMA product per split side)
    bf16x2 hi + lo, both bf16            (16 bits)

Answers, without a GPU: which (activation side, weight side) scheme per tower brings the end-to-end CP logit within 1e-3 of the
fp32 reference on every weight seed.       python tests/studies/operand_scheme_cpu.py [seeds] [outfits]

Test infrastructure only (imports nothing from the product path except the seeded generators).
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from outfitx_amd import synth

torch.set_num_threads(os.cpu_count())


def rnd(x, mode):
    if mode == "f32":
        return x
    if mode == "f16":
        return x.half().float()
    if mode == "bf16":
        return x.bfloat16().float()
    if mode == "f16x2":
        h = x.half().float()
        return h + (x - h).half().float()
    if mode == "bf16x2":
        h = x.bfloat16().float()
        return h + (x - h).bfloat16().float()
    raise ValueError(mode)


class Net:
    def __init__(self, W, a_mode, w_mode, attn_mode=None, sites=None):
        self.W = {k: torch.from_numpy(v) for k, v in W.items()}
        self.a, self.w = a_mode, w_mode
        self.attn = attn_mode or a_mode        # rounding of q, k, v, p inside the attention core
        self.sites = sites or {}               # substring of the weight name -> (a mode, w mode) override
        self.wc = {}

    def modes(self, wname):
        for k, v in self.sites.items():
            if any(t in wname for t in k.split("|")):
                return v
        return self.a, self.w

    def lin(self, x, wname, bname=None):
        am, wm = self.modes(wname)
        w = self.wc.get(wname)
        if w is None:
            w = self.wc[wname] = rnd(self.W[wname].reshape(self.W[wname].shape[0], -1), wm)
        y = rnd(x, am) @ w.T
        if bname is not None:
            y = y + self.W[bname]
        return y

    def ln(self, x, p):
        return torch.nn.functional.layer_norm(x, x.shape[-1:], self.W[p + ".weight"], self.W[p + ".bias"], 1e-5)

    def mha(self, h, dead, names, n_head):
        B, S, D = h.shape
        dh = D // n_head
        q, k, v = [rnd(t, self.attn).view(B, S, n_head, dh).transpose(1, 2) for t in names(h)]
        s = (q @ k.transpose(-1, -2)) * dh ** -0.5
        s = s.masked_fill(dead, float("-inf"))
        p = rnd(torch.softmax(s, -1), self.attn)
        return (p @ v).transpose(1, 2).reshape(B, S, D)

    def clip_layers(self, x, dead, prefix, n_layers, n_head):
        for i in range(n_layers):
            p = f"{prefix}encoder.layers.{i}."
            h = self.ln(x, p + "layer_norm1")
            names = lambda hh: [self.lin(hh, p + f"self_attn.{n}_proj.weight", p + f"self_attn.{n}_proj.bias") for n in "qkv"]
            o = self.mha(h, dead, names, n_head)
            x = x + self.lin(o, p + "self_attn.out_proj.weight", p + "self_attn.out_proj.bias")
            h = self.ln(x, p + "layer_norm2")
            u = self.lin(h, p + "mlp.fc1.weight", p + "mlp.fc1.bias")
            u = u * torch.sigmoid(1.702 * u)
            x = x + self.lin(u, p + "mlp.fc2.weight", p + "mlp.fc2.bias")
        return x

    def vit(self, px):
        N = px.shape[0]
        pt = px.view(N, 3, 7, 32, 7, 32).permute(0, 2, 4, 1, 3, 5).reshape(N, 49, 3072)
        pe = self.lin(pt, "vision_model.embeddings.patch_embedding.weight")
        cls = self.W["vision_model.embeddings.class_embedding"].expand(N, 1, -1)
        x = torch.cat([cls, pe], 1) + self.W["vision_model.embeddings.position_embedding.weight"][None]
        x = self.ln(x, "vision_model.pre_layrnorm")
        dead = torch.zeros(1, 1, 1, 50, dtype=torch.bool)
        x = self.clip_layers(x, dead, "vision_model.", 12, 12)
        return self.lin(self.ln(x[:, 0], "vision_model.post_layernorm"), "visual_projection.weight")

    def text(self, ids, att):
        N, T = ids.shape
        x = self.W["text_model.embeddings.token_embedding.weight"][ids] + self.W["text_model.embeddings.position_embedding.weight"][:T][None]
        dead = (att == 0)[:, None, None, :] | torch.triu(torch.ones(T, T, dtype=torch.bool), 1)[None, None]
        x = self.clip_layers(x, dead, "text_model.", 12, 8)
        x = self.ln(x, "text_model.final_layer_norm")
        eos = (ids == synth.EOS_ID).int().argmax(-1)
        return self.lin(x[torch.arange(N), eos], "text_projection.weight")

    def cp(self, emb, mask):
        B = emb.shape[0]
        x = torch.cat([self.W["outfit_token"].expand(B, 1, -1), emb], 1)
        dead = torch.cat([torch.zeros(B, 1, dtype=torch.bool), mask], 1)[:, None, None, :]
        for i in range(6):
            p = f"transformer_encoder.layers.{i}."
            h = self.ln(x, p + "norm1")
            qkv = self.lin(h, p + "self_attn.in_proj_weight", p + "self_attn.in_proj_bias")
            o = self.mha(h, dead, lambda hh: list(qkv.split(1024, -1)), 16)
            x = x + self.lin(o, p + "self_attn.out_proj.weight", p + "self_attn.out_proj.bias")
            h = self.ln(x, p + "norm2")
            u = self.lin(h, p + "linear1.weight", p + "linear1.bias")
            x = x + self.lin(torch.nn.functional.mish(u), p + "linear2.weight", p + "linear2.bias")
        return self.lin(x[:, 0], "cp_ffn.1.weight", "cp_ffn.1.bias")


def l2n(x):
    return x / x.norm(dim=-1, keepdim=True).clamp_min(1e-12)


def main():
    nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    n = 8
    # scheme = (vit a, vit w, text a, text w); the set transformer is x3-exact (bf16x3 on the GPU) throughout
    schemes = {
        "bf16|bf16": ("bf16", "bf16", "bf16", "bf16"),
        "f16|f16": ("f16", "f16", "f16", "f16"),
        "f16 a, w split (x2)": ("f16", "f16x2", "f16", "f16x2"),
        "a split, f16 w": ("f16x2", "f16", "f16x2", "f16"),
        "vit f16|f16, text x3": ("f16", "f16", "f16x2", "f16x2"),
        "vit x2 (w split), text x3": ("f16", "f16x2", "f16x2", "f16x2"),
        "bf16x2 a | bf16x2 w (bf16x3)": ("bf16x2", "bf16x2", "bf16x2", "bf16x2"),
        "bf16 a | bf16x2 w": ("bf16", "bf16x2", "bf16", "bf16x2"),
    }
    if len(sys.argv) > 3:
        schemes = {s: v for s, v in schemes.items() if s in sys.argv[3:]}
    mean = torch.tensor(synth.CLIP_MEAN).view(1, 3, 1, 1); std = torch.tensor(synth.CLIP_STD).view(1, 3, 1, 1)
    out = {s: {"logit": [], "vit": [], "text": []} for s in schemes}
    for ws in range(1, nseeds + 1):
        Wt, Wv, Wx = synth.outfit_transformer_weights(ws), synth.vision_weights(ws), synth.text_weights(ws)
        g = torch.Generator(); g.manual_seed(9000 + ws)
        u8 = torch.randint(0, 256, (k * n, 3, 224, 224), generator=g, dtype=torch.uint8)
        px = ((u8.float() * (1 / 255.0) - mean) / std).contiguous()
        ids, att = synth.token_batch(9000 + ws, k * n, 64, 8)
        ids, att = torch.from_numpy(ids[:, :8].copy()), torch.from_numpy(att[:, :8].copy())     # causal: tokens past EOS are dead
        mask = torch.zeros(k, n, dtype=torch.bool)
        t0 = time.perf_counter()
        with torch.no_grad():
            ref_net_v, ref_net_t, ref_net_s = Net(Wv, "f32", "f32"), Net(Wx, "f32", "f32"), Net(Wt, "f32", "f32")
            rv, rt = ref_net_v.vit(px), ref_net_t.text(ids, att)
            ref = ref_net_s.cp(torch.cat([l2n(rv), l2n(rt)], -1).view(k, n, -1), mask)
            for s, (va, vw, ta, tw) in schemes.items():
                gv, gt = Net(Wv, va, vw).vit(px), Net(Wx, ta, tw).text(ids, att)
                got = ref_net_s.cp(torch.cat([l2n(gv), l2n(gt)], -1).view(k, n, -1), mask)
                rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
                out[s]["logit"].append(rel(got, ref)); out[s]["vit"].append(rel(gv, rv)); out[s]["text"].append(rel(gt, rt))
        print(f"seed {ws}: {time.perf_counter() - t0:.0f} s  " + "  ".join(f"[{s}] {out[s]['logit'][-1]:.2e}" for s in schemes), file=sys.stderr, flush=True)
    res = {s: {m: {"max": max(v), "all": [float(f"{x:.3g}") for x in v]} for m, v in d.items()} for s, d in out.items()}
    for s, d in out.items():
        print(f"{s:34s} logit max {max(d['logit']):.2e} [" + " ".join(f"{x:.1e}" for x in d["logit"]) + f"]  vit {max(d['vit']):.1e}  text {max(d['text']):.1e}", file=sys.stderr)
    print(json.dumps({"metric": "max|d| / max|ref| over the batch", "outfits": k, "weight_seeds": nseeds, "schemes": res}))


if __name__ == "__main__":
    main()
