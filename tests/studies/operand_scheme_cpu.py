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
from oracle.torch_ref import TorchRef, l2n

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


class Net(TorchRef):
    """TorchRef with operand rounding: a_mode / w_mode per side, `sites` = {substring(s) of the weight name 'a|b': (a mode, w mode)}
    overrides, attn_mode = rounding of q, k, v, p inside the attention core (default: a_mode)."""

    def __init__(self, W, a_mode, w_mode, attn_mode=None, sites=None):
        self.a, self.w, self.sites = a_mode, w_mode, sites or {}
        attn = attn_mode or a_mode
        super().__init__(W, rnd=self._round, rnd_attn=lambda x: rnd(x, attn))

    def modes(self, wname):
        for k, v in self.sites.items():
            if any(t in wname for t in k.split("|")):
                return v
        return self.a, self.w

    def _round(self, x, side, wname):
        am, wm = self.modes(wname)
        return rnd(x, am if side == "a" else wm)


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
    # the shipped default 'f16w2' (DESIGN.md section 2): ViT f16 with split weights on patch / out-proj / fc2 and a three-product
    # projection tail, text tower three-product
    VIT_F16W2 = {"visual_projection": ("f16x2", "f16x2"), "fc2|out_proj|patch_embedding": ("f16", "f16x2")}
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
            runs = {s: (Net(Wv, va, vw), Net(Wx, ta, tw)) for s, (va, vw, ta, tw) in schemes.items()}
            if len(sys.argv) <= 3 or "f16w2" in sys.argv[3:]:
                runs["f16w2 (shipped default)"] = (Net(Wv, "f16", "f16", sites=VIT_F16W2), Net(Wx, "f16x2", "f16x2", attn_mode="f32"))
                out.setdefault("f16w2 (shipped default)", {"logit": [], "vit": [], "text": []})
            for s, (nv, nt) in runs.items():
                gv, gt = nv.vit(px), nt.text(ids, att)
                got = ref_net_s.cp(torch.cat([l2n(gv), l2n(gt)], -1).view(k, n, -1), mask)
                rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
                out[s]["logit"].append(rel(got, ref)); out[s]["vit"].append(rel(gv, rv)); out[s]["text"].append(rel(gt, rt))
        print(f"seed {ws}: {time.perf_counter() - t0:.0f} s  " + "  ".join(f"[{s}] {out[s]['logit'][-1]:.2e}" for s in out), file=sys.stderr, flush=True)
    res = {s: {m: {"max": max(v), "all": [float(f"{x:.3g}") for x in v]} for m, v in d.items()} for s, d in out.items()}
    for s, d in out.items():
        print(f"{s:34s} logit max {max(d['logit']):.2e} [" + " ".join(f"{x:.1e}" for x in d["logit"]) + f"]  vit {max(d['vit']):.1e}  text {max(d['text']):.1e}", file=sys.stderr)
    print(json.dumps({"metric": "max|d| / max|ref| over the batch", "outfits": k, "weight_seeds": nseeds, "schemes": res}))


if __name__ == "__main__":
    main()
