#!/usr/bin/env python3
"""CPU emulation: can the weight-correction product of the split-weight scheme (A . W_lo^T) run on the
block-scaled fp8 matrix instruction (v_mfma_scale_f32_16x16x128_f8f6f4, 2x the f16 rate) without moving the end-to-end error?

    y = f16(a) . f16(w)^T  +  2^-(sa+sw) . q_a(f16(a) 2^sa) . q_w((w - f16(w)) 2^sw)^T

q_a / q_w = rounding to OCP e4m3 (fp8) or e5m2 (bf8); products of two <= 4-bit significands are exact in fp32 and the
instruction accumulates in fp32, so fp32_matmul of the rounded operands is what it returns up to summation order.  sw is a per-tensor
power of two chosen at pack time (max|w_lo| 2^sw <= 256); sa is a compile-time constant.

    python tests/studies/fp8_correction_cpu.py [outfits] [seed ...]

Test infrastructure only."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from outfitx_amd import synth
from oracle.torch_ref import TorchRef, l2n

torch.set_num_threads(int(os.environ.get("OFX_THREADS", "4")))
F8 = {"e4m3": (torch.float8_e4m3fn, 448.0), "e5m2": (torch.float8_e5m2, 57344.0)}


def q8(x, fmt):
    dt, mx = F8[fmt]
    return x.clamp(-mx, mx).to(dt).float()


def f16(x):
    return x.half().float()


class Net(TorchRef):
    """sites: weight-name substring -> mode: 'f16' one product | 'w2' exact f16 lo product | ('f8', a_fmt, sa) fp8 correction |
    'x3' three-product (both sides split).  attn: rounding of q, k, v, p ('f16' or None)."""

    def __init__(self, W, sites, default="f16", attn="f16"):
        super().__init__(W, rnd_attn=(lambda x: f16(x)) if attn == "f16" else None)
        self.sites, self.default, self.amax = sites, default, {}

    def mode(self, wname):
        for k, v in self.sites.items():
            if any(t in wname for t in k.split("|")):
                return v
        return self.default

    def lin(self, x, wname, bname=None):
        m = self.mode(wname)
        w = self.W[wname].reshape(self.W[wname].shape[0], -1)
        if m == "f32":
            y = x @ w.T
        else:
            c = self._wc.get(wname)
            if c is None:
                hi = f16(w); lo = w - hi
                if isinstance(m, tuple):
                    sw = 2.0 ** np.floor(np.log2(256.0 / float(lo.abs().max())))
                    c = (hi, q8(lo * sw, "e4m3"), sw)
                else:
                    c = (hi, f16(lo), 1.0)
                self._wc[wname] = c
            hi, lo, sw = c
            a = f16(x)
            y = a @ hi.T
            if m == "w2":
                y = y + a @ lo.T
            elif m == "x3":
                y = y + a @ lo.T + f16(x - a) @ hi.T
            elif isinstance(m, tuple):
                _, afmt, sa = m
                self.amax[wname.split(".")[-2]] = max(self.amax.get(wname.split(".")[-2], 0.0), float(a.abs().max()))
                y = y + (q8(a * 2.0 ** sa, afmt) @ lo.T) * (2.0 ** -sa / sw)
        return y if bname is None else y + self.W[bname]


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    seeds = [int(s) for s in sys.argv[2:]] or [1, 2, 3, 4, 6, 14, 44, 89, 97, 99]
    n = 8
    ALL = "fc1|fc2|out_proj|q_proj|k_proj|v_proj|patch_embedding"
    vit_schemes = {
        "f16 single": {"visual_projection": "x3"},
        "f16w2x (exact lo product)": {"visual_projection": "x3", ALL: "w2"},
        "f8 corr: a e4m3 2^0": {"visual_projection": "x3", ALL: ("f8", "e4m3", 0)},
        "f8 corr: a e4m3 2^2": {"visual_projection": "x3", ALL: ("f8", "e4m3", 2)},
        "f8 corr: a e5m2": {"visual_projection": "x3", ALL: ("f8", "e5m2", 0)},
    }
    mean = torch.tensor(synth.CLIP_MEAN).view(1, 3, 1, 1); std = torch.tensor(synth.CLIP_STD).view(1, 3, 1, 1)
    out = {s: {"logit": [], "abs": [], "vit": []} for s in vit_schemes}
    amax = {}
    for ws in seeds:
        Wt, Wv, Wx = synth.outfit_transformer_weights(ws), synth.vision_weights(ws), synth.text_weights(ws)
        g = torch.Generator(); g.manual_seed(9000 + ws)
        u8 = torch.randint(0, 256, (k * n, 3, 224, 224), generator=g, dtype=torch.uint8)
        px = ((u8.float() * (1 / 255.0) - mean) / std).contiguous()
        ids, att = synth.token_batch(9000 + ws, k * n, 64, 8)
        ids, att = torch.from_numpy(ids[:, :8].copy()), torch.from_numpy(att[:, :8].copy())
        mask = torch.zeros(k, n, dtype=torch.bool)
        t0 = time.perf_counter()
        with torch.no_grad():
            rs = TorchRef(Wt)
            rv, rt = TorchRef(Wv).vit(px), TorchRef(Wx).text(ids, att)
            ref = rs.cp(torch.cat([l2n(rv), l2n(rt)], -1).view(k, n, -1), mask)
            gt = Net(Wx, {}, default="x3", attn=None).text(ids, att)          # three-product text tower, fp32 attention (the shipped one)
            for s, sites in vit_schemes.items():
                nv = Net(Wv, sites)
                gv = nv.vit(px)
                got = rs.cp(torch.cat([l2n(gv), l2n(gt)], -1).view(k, n, -1), mask)
                d = float((got - ref).abs().max())
                out[s]["logit"].append(d / float(ref.abs().max())); out[s]["abs"].append(d)
                out[s]["vit"].append(float((gv - rv).abs().max() / rv.abs().max()))
                for kk, v in nv.amax.items():
                    amax[kk] = max(amax.get(kk, 0.0), v)
        print(f"seed {ws}: {time.perf_counter() - t0:.0f} s  " + "  ".join(f"[{s}] {out[s]['logit'][-1]:.2e}" for s in out), file=sys.stderr, flush=True)
    for s, d in out.items():
        print(f"{s:30s} logit median {np.median(d['logit']):.2e} max {max(d['logit']):.2e}  abs median {np.median(d['abs']):.2e} max {max(d['abs']):.2e}  vit max {max(d['vit']):.1e}",
              file=sys.stderr)
    print(json.dumps({"metric": "max|d| / max|ref| over the batch", "outfits": k, "weight_seeds": seeds, "max_abs_activation_by_site": amax,
                      "schemes": {s: {m: [float(f"{x:.3g}") for x in v] for m, v in d.items()} for s, d in out.items()}}))


if __name__ == "__main__":
    main()
