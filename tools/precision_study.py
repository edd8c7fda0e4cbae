"""CPU study (no GPU): end-to-end error of Model.forward vs the fp32 oracle when every MFMA-class contraction of the HIP
path (token-major Linear / conv / attention products) is evaluated with a given operand split.  Answers: how many MFMA
products per contraction does the 1e-3 tolerance really need?  Uses the oracle's own forward with F.linear / F.conv1d /
einsum replaced by emulations; conditioning projections ([B, Tc] rows) stay fp32 as in the HIP path.

    python tools/precision_study.py [--dim 128 --depth 6 --n 256 --batch 2]
"""
import argparse, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ns2_oracle as O
from naturalspeech2_pytorch_amd import Model

BF, FH = torch.bfloat16, torch.float16

def split(x, fmt, n):
    """x -> list of n pieces in `fmt` (as float64) whose sum approximates x"""
    out, r = [], x.double()
    for _ in range(n):
        p = r.float().to(fmt).double()
        out.append(p); r = r - p
    return out

def fp8_block(x, axis=-1):
    """MX-style e4m3 with one power-of-two scale per 32 elements along `axis` (as float64)"""
    x = x.double().movedim(axis, -1)
    K = x.shape[-1]; pad = (-K) % 32
    xp = F.pad(x, (0, pad)).reshape(*x.shape[:-1], -1, 32)
    amax = xp.abs().amax(-1, keepdim=True).clamp_min(1e-300)
    scale = torch.exp2(torch.ceil(torch.log2(amax)) - 8)
    q = (xp / scale).float().to(torch.float8_e4m3fn).double() * scale
    return q.reshape(*x.shape[:-1], -1)[..., :K].movedim(-1, axis)

def fp8_row(x, axis=-1, extra=0):
    """e4m3 with ONE power-of-two scale per row (the whole contraction axis); extra = additional binades of head-room"""
    x = x.double().movedim(axis, -1)
    amax = x.abs().amax(-1, keepdim=True).clamp_min(1e-300)
    scale = torch.exp2(torch.ceil(torch.log2(amax)) - 8 + extra)
    q = (x / scale).float().to(torch.float8_e4m3fn).double() * scale
    return q.movedim(-1, axis)

class Scheme:
    def __init__(self, name, units, fn, attn_fn=None, resid_fn=None):
        self.name, self.units, self.fn, self.attn_fn = name, units, fn, attn_fn or fn   # attn_fn: the two attention products
        self.resid_fn = resid_fn or fn          # Linears whose output goes to the residual stream (out features == dim)
        self.site_fn = None                     # optional: (kind, weight shape) -> contraction fn, overrides fn / resid_fn

def per_site(name, units, default, **sites):
    """sites: ffconv / ffin / ffout / qkv / attnout / wavenet (all Wavenet convs) or wavenet3 (dilated k = 3 convs + init conv)
    / wavenet1 (1x1 res / skip / final convs) -> fn (classified by the weight's shape)"""
    sc = Scheme(name, units, default, mk(FH, [(0, 0)]))
    def pick(kind, ws):
        if kind == "conv":
            site = "wavenet" if ws[0] in (DIM, 2 * DIM) or ws[1] == DIM else "ffconv"
            if site == "wavenet" and "wavenet" not in sites:
                site = "wavenet3" if ws[2] == 3 else "wavenet1"
        elif ws[1] == DIM and ws[0] % 64 == 0 and ws[0] != DIM:
            site = "qkv"                                   # to_q [512, dim] / to_kv [1024, dim] (8 heads x 64)
        elif ws[1] == DIM and ws[0] != DIM:
            site = "ffin"                                  # [2 * inner, dim], inner = int(dim * 4 * 2 / 3)
        elif ws[0] == DIM and ws[1] % 64 == 0 and ws[1] != DIM:
            site = "attnout"                               # to_out [dim, 512]
        elif ws[0] == DIM and ws[1] != DIM:
            site = "ffout"
        else:
            site = "other"                                 # [dim, dim]: to_pred / final Linear
        return sites.get(site, default)
    sc.site_fn = pick
    return sc

def mk(fmt, terms):
    """terms: list of (a_piece_index, w_piece_index)"""
    na = 1 + max(t[0] for t in terms); nw = 1 + max(t[1] for t in terms)
    def f(a, w, contract):
        ap, wp = split(a, fmt, na), split(w, fmt, nw)
        return sum(contract(ap[i], wp[j]) for i, j in terms)
    return f

def fp8_cross_row():
    def f(a, w, contract):
        ah, al = split(a, BF, 2); wh, wl = split(w, BF, 2)
        return contract(ah, wh) + contract(fp8_row(ah), fp8_row(wl)) + contract(fp8_row(al), fp8_row(wh))
    return f

def fp8_cross(which):
    def f(a, w, contract):
        ah, al = split(a, BF, 2); wh, wl = split(w, BF, 2)
        y = contract(ah, wh)
        if "aw" in which: y = y + contract(fp8_block(ah), fp8_block(wl))     # a_hi * w_lo on the fp8 MFMA
        else: y = y + contract(ah, wl)
        if "wa" in which: y = y + contract(fp8_block(al), fp8_block(wh))     # a_lo * w_hi on the fp8 MFMA
        else: y = y + contract(al, wh)
        return y
    return f

def f16_e5m2_cross(lo_fmt=torch.float8_e5m2):
    """fp16 hi.hi on the f16 MFMA + both cross terms in ONE fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4): hi8 = e5m2(x)
    (the top byte of the fp16 word, RNE), lo8 = fmt((x - fp16(x)) * 2^12) with the constant 2^-12 applied as the MFMA's block scale"""
    def q(x, fmt): return x.float().to(fmt).double()
    def f(a, w, contract):
        ah = a.float().to(FH).double(); wh = w.float().to(FH).double()
        al = a.double() - ah; wl = w.double() - wh
        a8, w8 = q(ah, torch.float8_e5m2), q(wh, torch.float8_e5m2)
        al8, wl8 = q(al * 4096.0, lo_fmt) / 4096.0, q(wl * 4096.0, lo_fmt) / 4096.0
        return contract(ah, wh) + contract(a8, wl8) + contract(al8, w8)
    return f

def minifloat_block(x, ebits, mbits, bias, axis=-1, block=32):
    """OCP-MX style element format (no inf/nan) with one power-of-two scale per `block` elements along `axis` (as float64)"""
    x = x.double().movedim(axis, -1)
    K = x.shape[-1]; pad = (-K) % block
    xp = F.pad(x, (0, pad)).reshape(*x.shape[:-1], -1, block)
    emax = (1 << ebits) - 1 - bias
    fmax = 2.0 ** emax * (2 - 2.0 ** -mbits)
    amax = xp.abs().amax(-1, keepdim=True).clamp_min(1e-300)
    scale = torch.exp2(torch.ceil(torch.log2(amax / fmax)))
    v = xp / scale
    a = v.abs().clamp_min(1e-300)
    e = torch.floor(torch.log2(a)).clamp(min=1 - bias, max=emax)
    step = torch.exp2(e - mbits)
    q = (torch.round(a / step) * step).clamp(max=fmax) * torch.sign(v)
    return (q * scale).reshape(*x.shape[:-1], -1)[..., :K].movedim(-1, axis)

def f16_mx_cross(ebits, mbits, bias):
    """fp16 hi.hi + both cross terms with MX-block-scaled minifloat operands (fp6 e2m3 / e3m2, fp4 e2m1: 4x the 16-bit MFMA rate)"""
    def f(a, w, contract, axis_a=-1, axis_w=-1):
        ah = a.float().to(FH).double(); wh = w.float().to(FH).double()
        al = a.double() - ah; wl = w.double() - wh
        q = lambda t, ax: minifloat_block(t, ebits, mbits, bias, ax)
        return contract(ah, wh) + contract(q(ah, axis_a), q(wl, axis_w)) + contract(q(al, axis_a), q(wh, axis_w))
    f.takes_axes = True
    return f

SCHEMES = [
    Scheme("bf16 x3  hi.hi+hi.lo+lo.hi (current exact)", 3.0, mk(BF, [(0, 0), (0, 1), (1, 0)])),
    Scheme("bf16 x1  hi.hi (current fast)", 1.0, mk(BF, [(0, 0)])),
    Scheme("bf16 x2  hi.hi+lo.hi (weights hi only)", 2.0, mk(BF, [(0, 0), (1, 0)])),
    Scheme("bf16 x2  hi.hi+hi.lo (activations hi only)", 2.0, mk(BF, [(0, 0), (0, 1)])),
    Scheme("fp16 x1  hi.hi", 1.0, mk(FH, [(0, 0)])),
    Scheme("fp16 x2  hi.hi+lo.hi (weights hi only)", 2.0, mk(FH, [(0, 0), (1, 0)])),
    Scheme("fp16 x2  hi.hi+hi.lo (activations hi only)", 2.0, mk(FH, [(0, 0), (0, 1)])),
    Scheme("fp16 x3", 3.0, mk(FH, [(0, 0), (0, 1), (1, 0)])),
    Scheme("fp16 x1 GEMMs, fp16 x3 attention products", 1.2, mk(FH, [(0, 0)]), mk(FH, [(0, 0), (0, 1), (1, 0)])),
    Scheme("fp16 x3 GEMMs, fp16 x1 attention products", 2.8, mk(FH, [(0, 0), (0, 1), (1, 0)]), mk(FH, [(0, 0)])),
    Scheme("fp16 x1, but x3 for Linears writing the residual stream", 1.2, mk(FH, [(0, 0)]), None, mk(FH, [(0, 0), (0, 1), (1, 0)])),
    Scheme("fp16 hi.hi + cross terms fp8 e5m2 x e5m2(lo*2^12)", 2.0, f16_e5m2_cross()),
    Scheme("fp16 hi.hi + cross terms fp8 e5m2 x e4m3(lo*2^12)", 2.0, f16_e5m2_cross(torch.float8_e4m3fn)),
    Scheme("fp16 hi.hi + cross terms MX fp6 e2m3 (GEMMs), fp16 attention", 1.5, f16_mx_cross(2, 3, 1), mk(FH, [(0, 0)])),
    Scheme("fp16 hi.hi + cross terms MX fp6 e3m2 (GEMMs), fp16 attention", 1.5, f16_mx_cross(3, 2, 3), mk(FH, [(0, 0)])),
    Scheme("fp16 hi.hi + cross terms MX fp4 e2m1 (GEMMs), fp16 attention", 1.5, f16_mx_cross(2, 1, 1), mk(FH, [(0, 0)])),
    Scheme("fp16 hi.hi + cross terms fp8 e5m2 (GEMMs), fp16 attention", 2.0, f16_e5m2_cross(), mk(FH, [(0, 0)])),
    per_site("mixed, FF conv fp16 x1", 1.55, f16_e5m2_cross(), ffconv=mk(FH, [(0, 0)])),
    per_site("mixed, FF conv + wavenet k3 convs fp16 x1 (= hybrid)", 1.4, f16_e5m2_cross(), ffconv=mk(FH, [(0, 0)]), wavenet3=mk(FH, [(0, 0)])),
    per_site("hybrid + FF-in fp16 x1 (= hybrid_ffin)", 1.3, f16_e5m2_cross(), ffconv=mk(FH, [(0, 0)]), wavenet3=mk(FH, [(0, 0)]), ffin=mk(FH, [(0, 0)])),
    per_site("hybrid + FF-out fp16 x1", 1.35, f16_e5m2_cross(), ffconv=mk(FH, [(0, 0)]), wavenet3=mk(FH, [(0, 0)]), ffout=mk(FH, [(0, 0)])),
    per_site("hybrid + whole FF fp16 x1 (= hybrid_ff)", 1.2, f16_e5m2_cross(), ffconv=mk(FH, [(0, 0)]), wavenet3=mk(FH, [(0, 0)]), ffin=mk(FH, [(0, 0)]), ffout=mk(FH, [(0, 0)])),
    per_site("mixed, FF conv + wavenet 1x1 convs fp16 x1", 1.45, f16_e5m2_cross(), ffconv=mk(FH, [(0, 0)]), wavenet1=mk(FH, [(0, 0)])),
    per_site("mixed, FF-in fp16 x1", 1.9, f16_e5m2_cross(), ffin=mk(FH, [(0, 0)])),
    per_site("mixed, FF-out fp16 x1", 1.95, f16_e5m2_cross(), ffout=mk(FH, [(0, 0)])),
    per_site("mixed, qkv fp16 x1", 1.9, f16_e5m2_cross(), qkv=mk(FH, [(0, 0)])),
    per_site("mixed, attention out-projection fp16 x1", 1.95, f16_e5m2_cross(), attnout=mk(FH, [(0, 0)])),
    per_site("mixed, FF conv + FF-in fp16 x1", 1.45, f16_e5m2_cross(), ffconv=mk(FH, [(0, 0)]), ffin=mk(FH, [(0, 0)])),
    per_site("mixed, whole FF fp16 x1", 1.4, f16_e5m2_cross(), ffconv=mk(FH, [(0, 0)]), ffin=mk(FH, [(0, 0)]), ffout=mk(FH, [(0, 0)])),
    per_site("mixed, wavenet fp16 x1", 1.8, f16_e5m2_cross(), wavenet=mk(FH, [(0, 0)])),
    per_site("mixed, FF conv + wavenet fp16 x1", 1.35, f16_e5m2_cross(), ffconv=mk(FH, [(0, 0)]), wavenet=mk(FH, [(0, 0)])),
    per_site("fp16 x1, FF conv mixed", 1.45, mk(FH, [(0, 0)]), ffconv=f16_e5m2_cross()),
    per_site("fp16 x1, wavenet mixed", 1.2, mk(FH, [(0, 0)]), wavenet=f16_e5m2_cross()),
    per_site("fp16 x1, qkv+attnout mixed", 1.1, mk(FH, [(0, 0)]), qkv=f16_e5m2_cross(), attnout=f16_e5m2_cross()),
    per_site("fp16 x1, ffin+ffout mixed", 1.15, mk(FH, [(0, 0)]), ffin=f16_e5m2_cross(), ffout=f16_e5m2_cross()),
    Scheme("bf16 hi.hi + both cross terms on fp8(e4m3, MX32)", 2.0, fp8_cross("aw wa")),
    Scheme("bf16 hi.hi + both cross terms on fp8, one scale per row", 2.0, fp8_cross_row()),
    Scheme("bf16 hi.hi + lo.hi bf16 + hi.lo on fp8", 2.5, fp8_cross("aw")),
    Scheme("bf16 hi.hi + hi.lo bf16 + lo.hi on fp8", 2.5, fp8_cross("wa")),
]

DIM = 0

def run(sd, x, t, scheme, big_rows):
    lin0, conv0, ein0 = F.linear, F.conv1d, torch.einsum
    def lin(inp, w, b=None):
        if inp.numel() // inp.shape[-1] < big_rows: return lin0(inp, w, b)
        f = scheme.resid_fn if (w.shape[0] == DIM and w.shape[1] != DIM) else scheme.fn
        if scheme.site_fn: f = scheme.site_fn("lin", tuple(w.shape))
        y = f(inp, w, lambda a, ww: a @ ww.t())          # contraction axis: last of both
        return (y + (b.double() if b is not None else 0)).float()
    def conv(inp, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        if inp.shape[-1] * inp.shape[0] < big_rows: return conv0(inp, w, b, stride, padding, dilation, groups)
        cf = lambda a, ww: conv0(a, ww, None, stride, padding, dilation, groups)      # contraction axis: channels (dim 1 of both)
        fn = scheme.site_fn("conv", tuple(w.shape)) if scheme.site_fn else scheme.fn
        y = fn(inp, w, cf, 1, 1) if getattr(fn, "takes_axes", False) else fn(inp, w, cf)
        return (y + (b.double()[None, :, None] if b is not None else 0)).float()
    def ein(eq, a, bb):
        y = scheme.attn_fn(a, bb, lambda p, q: ein0(eq, p, q))
        return y.float()
    F.linear, F.conv1d, torch.einsum = lin, conv, ein
    O.F.linear, O.F.conv1d, O.torch.einsum = lin, conv, ein
    try:
        with torch.no_grad(): return O.model_forward(sd, x, t)
    finally:
        F.linear, F.conv1d, torch.einsum = lin0, conv0, ein0
        O.F.linear, O.F.conv1d, O.torch.einsum = lin0, conv0, ein0

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=128); ap.add_argument("--depth", type=int, default=6)
    ap.add_argument("--n", type=int, default=256); ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--seeds", type=int, default=2)
    ap.add_argument("--times", type=float, default=None, help="fix the diffusion time of every utterance")
    ap.add_argument("--only", default="", help="comma-separated substrings of scheme names to run")
    ap.add_argument("--scale-ffin", type=float, default=1.0,
                    help="multiply every FF-in weight (GEGLU output grows with its square): at random init the FF branch is dominated "
                         "by its biases and hides its own rounding; 6 makes its data term dominant, like a trained branch carrying signal")
    a = ap.parse_args()
    global DIM
    DIM = a.dim
    print(f"Model(dim={a.dim}, depth={a.depth}), batch {a.batch} x {a.n} frames, random-init weights" + (f", FF-in weights x{a.scale_ffin:g}" if a.scale_ffin != 1 else "") + f"; rel = |y - y_fp32| / |y_fp32| (Frobenius)")
    rows = []
    for sc in SCHEMES:
        if a.only and not any(k in sc.name for k in a.only.split(",")): continue
        errs = []
        for seed in range(a.seeds):
            torch.manual_seed(seed)
            m = Model(dim=a.dim, depth=a.depth)
            sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
            if a.scale_ffin != 1.0:
                for k, v in sd.items():
                    if v.ndim == 2 and v.shape[1] == a.dim and v.shape[0] != a.dim and v.shape[0] % 64 != 0:
                        v.mul_(a.scale_ffin)
            x = torch.randn(a.batch, a.n, a.dim); t = torch.rand(a.batch)
            if a.times is not None: t = torch.full((a.batch,), a.times)
            with torch.no_grad(): ref = O.model_forward(sd, x, t).double()
            y = run(sd, x, t, sc, big_rows=a.batch * 8).double()
            errs.append(((y - ref).norm() / ref.norm()).item())
        rows.append((sc.name, sc.units, max(errs)))
        print("    per seed: " + " ".join(f"{e:.2e}" for e in errs))
        print(f"  {sc.name:52s} MFMA units {sc.units:3.1f}   rel err (max over {a.seeds} seeds) {max(errs):.2e}", flush=True)

if __name__ == "__main__":
    main()
