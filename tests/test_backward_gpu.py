"""SURVEY §8f-4 on the MI355X: the backward kernels of libns2hip.

  * every `training.HipBackend` call (= one or two C-ABI entry points of include/ns2hip.h "training") against its plain-torch
    restatement (tests/emu_backend.py, the same object the CPU suite checks the chain rule with), ragged and tile-aligned shapes;
  * every parameter's `.grad` and `x.grad` of `Model` through the HIP training path against the REFERENCE's own autograd -- the
    unmodified upstream `Model` from oracle/_ref, fp32 on the same GPU -- at Model(dim=128, depth=6) B = 4 x 1024 (BASELINE
    config 1's training shape) and Model(dim=512, depth=12) B = 2 x 512, relative error per tensor <= 1e-3 (seen: ~1e-5);
  * the conditioned model (cross-attention, aligned conditioning, null selects) against the PyTorch composite;
  * the reference's own `NaturalSpeech2.forward` loss + backward over `compat.HipBackedModel(train_backend="hip")`;
  * determinism (fixed-slot reductions: two backward passes give bit-identical gradients) and the in-place weight refresh.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

from naturalspeech2_pytorch_amd import Model, NaturalSpeech2, ops, training  # noqa: E402
from naturalspeech2_pytorch_amd.compat import hip_backed_model_class  # noqa: E402
from naturalspeech2_pytorch_amd.autograd_path import model_forward_autograd  # noqa: E402
from oracle.ref_stub import load_reference, reference_available  # noqa: E402
from tests.emu_backend import EP, ETP, EPW, EmuBackend, rup  # noqa: E402
from tests.golden.gen import make_input, make_weights  # noqa: E402
from tests.parity_record import record  # noqa: E402

DEV = torch.device("cuda:0")
needs_ref = pytest.mark.skipif(not reference_available(), reason="run oracle/make_ref.py (or __graft_entry__.build()) in the build container")
HB = training.HipBackend()
EB = EmuBackend()


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp(min=1e-30)).item()


def rnd(name, shape, seed, scale=1.0):
    return make_input(name, tuple(shape), seed=seed) * scale


def join(p):
    return ops.join(p).cpu()


def tjoin(tp):
    return ops.join(ops.Planes(tp.buf, tp.rows, tp.ld, True)).cpu()


class _W:       # a stand-in parameter for the pack cache
    def __init__(self, t):
        self.t = torch.nn.Parameter(t.to(DEV))


def hpack(t):
    w = _W(t)
    pw = HB.pack(("t", id(w.t)), (w.t,), lambda: w.t)
    pw._keep = w
    return pw


# ------------------------------------------------------------------------------------------------ backend calls, one by one
@pytest.mark.parametrize("M,C,seq_len", [(256, 64, 64), (200, 72, 40), (1024, 1365, 256), (96, 512, 96)])
def test_grad_prep_rows_transposed_colsum(M, C, seq_len):
    x = rnd("gp", (M, C), 1)
    ld = rup(C, 4) + 4
    xp = torch.full((M, ld), float("nan"))
    xp[:, :C] = x
    row, tp, cs = HB.grad_prep(xp.to(DEV), C, want_row=True, want_t=True, want_colsum=True, t_rows=rup(C, 256))
    er, et, ec = EB.grad_prep(xp, C, want_row=True, want_t=True, want_colsum=True, t_rows=rup(C, 256))
    assert rel(join(row), er.t) < 1e-5 and torch.equal(join(row)[:, C:], er.t[:, C:])
    tj = tjoin(tp)
    assert tj.shape == et.t.shape and rel(tj, et.t) < 1e-5
    assert torch.equal(tj[C:], et.t[C:]) and torch.equal(tj[:, M:], et.t[:, M:])          # zero padding rows / columns
    assert rel(cs.cpu(), ec) < 1e-5
    # per utterance (attention operands)
    if C % 8 == 0:
        _, tpb, _ = HB.grad_prep(xp.to(DEV), C, want_t=True, seq_len=seq_len, per_batch=True)
        _, etb, _ = EB.grad_prep(xp, C, want_t=True, seq_len=seq_len, per_batch=True)
        assert rel(tjoin(tpb), etb.t) < 1e-5


@pytest.mark.parametrize("M,C,seq_len,shifts", [(256, 64, 64, (2, 1, 0)), (240, 96, 40, (8, 4, 0)), (512, 1376, 256, (2, 1, 0)),
                                               (2048, 128, 1024, (256, 128, 0)), (192, 64, 0, (0,))])
def test_planes_transpose_shifted_taps(M, C, seq_len, shifts):
    x = rnd("pt", (M, C + 32), 2)
    p = HB.split(x.to(DEV))
    xr = join(p)                                        # the values the planes hold
    tp = HB.transpose(p, 32, C, seq_len, shifts)
    et = EB.transpose(EP(xr), 32, C, seq_len, shifts)
    tj = tjoin(tp)
    assert tj.shape == et.t.shape and torch.equal(tj, et.t)        # a transposition of hi / lo pairs: exact
    if seq_len:
        assert torch.equal(tjoin(HB.transpose(p, 0, 64, seq_len, per_batch=True)), EB.transpose(EP(xr), 0, 64, seq_len, per_batch=True).t)


@pytest.mark.parametrize("R,K,T,M", [(512, 512, 1, 4096), (64, 64, 3, 480), (1365, 1365, 3, 2048), (2730, 512, 1, 1024), (128, 128, 1, 160),
                                     (1536, 512, 1, 32768)])
def test_wgrad_split_k(R, K, T, M):
    dy = rnd("wg_dy", (M, R), 3, 0.5)
    x = rnd("wg_x", (M, rup(K, 32)), 4)
    x[:, K:] = 0
    seq = M // 2 if T > 1 else 0
    shifts = tuple((T - 1 - t) * 2 for t in range(T))
    _, dyt, _ = HB.grad_prep(dy.to(DEV), R, want_t=True)
    xp = HB.split(x.to(DEV))
    xt = HB.transpose(xp, 0, rup(K, 32), seq, shifts)
    dw = HB.wgrad(dyt, xt, R, T, K).cpu()
    _, edyt, _ = EB.grad_prep(dy, R, want_t=True)
    ext = EB.transpose(EP(x), 0, rup(K, 32), seq, shifts)
    ref = EB.wgrad(edyt, ext, R, T, K)
    assert dw.shape == (R, K, T) and rel(dw, ref) < 2e-5
    dw2 = HB.wgrad(dyt, xt, R, T, K).cpu()
    assert torch.equal(dw, dw2)                         # fixed slots, fixed order


@pytest.mark.parametrize("cout,cin,taps,dil,M,seq", [(512, 512, 3, 4, 2048, 512), (64, 96, 3, 1, 400, 40), (1365, 1365, 3, 1, 1024, 256),
                                                    (512, 1365, 0, 1, 1024, 0), (128, 128, 1, 1, 300, 100)])
def test_dgrad_is_the_forward_kernel_on_the_flipped_weight(cout, cin, taps, dil, M, seq):
    """dX = dY W through ns2_linear_f32 on the second pack (transposed, taps flipped, pad_left = 0) == autograd of the conv"""
    w = rnd("dg_w", (cout, cin, taps) if taps else (cout, cin), 5, 0.05)
    dy = rnd("dg_dy", (M, cout), 6)
    x = torch.zeros(M, cin, requires_grad=True)
    if taps:
        B = M // seq
        xx = torch.nn.functional.pad(x.reshape(B, seq, cin).transpose(1, 2), (dil * (taps - 1), 0))
        y = torch.nn.functional.conv1d(xx, w, dilation=dil).transpose(1, 2).reshape(M, cout)
    else:
        y = x @ w.t()
    (y * dy).sum().backward()
    wp = _W(w)
    pw = training._bwd_pack(HB, wp.t)
    dyp = HB.split(dy.to(DEV))
    dx = HB.gemm_f32(pw, dyp, taps=taps, dil=dil, seq_len=seq if taps else 0, pad_left=0 if taps else -1)[:, :cin].cpu()
    assert rel(dx, x.grad) < 2e-5


def test_film_gate_and_geglu_and_rmsnorm_backward():
    B, N, d, f = 3, 200, 128, 341
    M = B * N
    h, dg = rnd("fg_h", (M, d), 7), rnd("fg_dg", (M, d), 8)
    film = torch.cat((1 + 0.3 * rnd("fg_g", (B, d), 9), 0.3 * rnd("fg_b", (B, d), 10)), -1)
    g = HB.film_gate_fwd(h.to(DEV), film.to(DEV), N, d).cpu()
    assert rel(g, EB.film_gate_fwd(h, film, N, d)) < 1e-5
    dh, dfilm = HB.film_gate_bwd(dg.to(DEV), h.to(DEV), film.to(DEV), B, N, d)
    edh, edfilm = EB.film_gate_bwd(dg, h, film, B, N, d)
    assert rel(dh.cpu(), edh) < 1e-5 and rel(dfilm.cpu(), edfilm) < 1e-5
    pre, dhh = rnd("gg_pre", (M, rup(2 * f, 32)), 11), rnd("gg_dh", (M, rup(f, 32)), 12)
    assert rel(join(HB.geglu_fwd(pre.to(DEV), f)), EB.geglu_fwd(pre, f).t) < 1e-5
    dpre = HB.geglu_bwd(dhh.to(DEV), pre.to(DEV), f).cpu()
    assert rel(dpre[:, :2 * f], EB.geglu_bwd(dhh, pre, f)[:, :2 * f]) < 1e-5
    x, dy, add = rnd("rn_x", (M, d), 13), rnd("rn_dy", (M, d), 14), rnd("rn_add", (M, d), 15)
    gamma = 1 + 0.2 * rnd("rn_g", (d,), 16)
    for kw in (dict(cond=film), dict(gamma=gamma), dict(gamma=gamma, cond=film)):
        dev_kw = {k: v.to(DEV) for k, v in kw.items()}
        dx, dc, dgm = HB.rmsnorm_bwd(x.to(DEV), dy.to(DEV), B, N, d, dx_add=add.to(DEV), **dev_kw)
        edx, edc, edg = EB.rmsnorm_bwd(x, dy, B, N, d, dx_add=add, **kw)
        assert rel(dx.cpu(), edx) < 1e-5
        if edc is not None:
            assert rel(dc.cpu(), edc) < 1e-5
        if edg is not None:
            assert rel(dgm.cpu(), edg) < 1e-5
    # in place on the residual gradient (dx aliases dx_add) -- what the Functions could do; same result
    xin, dyin = x.to(DEV), dy.to(DEV)
    buf = add.to(DEV).clone()
    from naturalspeech2_pytorch_amd._lib import check
    S = HB.lib.ns2_rmsnorm_bwd_slices(N)
    cpart = torch.empty(B * S, 2 * d, device=DEV)
    check(HB.lib.ns2_rmsnorm_bwd(xin.data_ptr(), d, dyin.data_ptr(), d, None, film.to(DEV).data_ptr(), 2 * d, B, N, d, buf.data_ptr(), buf.data_ptr(),
                                 d, cpart.data_ptr(), None, torch.cuda.current_stream().cuda_stream), "ns2_rmsnorm_bwd")
    assert rel(buf.cpu(), EB.rmsnorm_bwd(x, dy, B, N, d, cond=film, dx_add=add)[0]) < 1e-5


@pytest.mark.parametrize("B,H,Nq,Nk,cross", [(2, 8, 256, 256, False), (2, 2, 200, 200, False), (1, 8, 1024, 1024, False), (3, 8, 160, 32, True),
                                             (2, 4, 96, 40, True)])
def test_attention_forward_lse_and_backward(B, H, Nq, Nk, cross):
    a = H * 64
    q, k, v = (rnd(n, (B * L, a), s, 1.0) for n, s, L in (("aq", 17, Nq), ("ak", 18, Nk), ("av", 19, Nk)))
    do = rnd("ado", (B * Nq, a), 20)
    if cross:
        qp, kvp = HB.split(q.to(DEV)), HB.split(torch.cat((k, v), -1).to(DEV))
        kp, vp, qc, kc, vc = kvp, kvp, 0, 0, a
    else:
        qkv = HB.split(torch.cat((q, k, v), -1).to(DEV))
        qp, kp, vp, qc, kc, vc = qkv, qkv, qkv, 0, a, 2 * a
    qr, kr, vr = join(qp)[:, qc:qc + a], join(kp)[:, kc:kc + a], join(vp)[:, vc:vc + a]      # what the planes hold
    vt = HB.transpose(vp, vc, a, Nk, per_batch=True)
    o, lse = HB.attention(qp, qc, kp, kc, vt, B, H, Nq, Nk)
    eq, ek, ev = EP(qr.clone()), EP(kr.clone()), EP(vr.clone())
    eo, else_ = EB.attention(eq, 0, ek, 0, EB.transpose(ev, 0, a, Nk, per_batch=True), B, H, Nq, Nk)
    assert rel(join(o), eo.t) < 2e-5 and (lse.cpu() - else_).abs().max() < 1e-4
    delta = HB.attention_delta(do.to(DEV), o, B, H, Nq)
    assert rel(delta.cpu(), EB.attention_delta(do, EP(join(o)), B, H, Nq)) < 1e-5
    do_row, _, _ = HB.grad_prep(do.to(DEV), a, want_row=True)
    dq = torch.full((B * Nq, a + 32), float("nan"), device=DEV)
    dkv = torch.full((B * Nk, 2 * a), float("nan"), device=DEV)
    HB.attention_bwd(qp, qc, kp, kc, vp, vc, do_row, lse, delta, B, H, Nq, Nk, dq=(dq, 32), dkv=(dkv, 0, a))
    # reference: torch autograd of softmax attention on the plane values
    tq, tk, tv = (t.clone().requires_grad_(True) for t in (qr, kr, vr))
    hd = lambda t, n: t.reshape(B, n, H, 64).transpose(1, 2)          # noqa: E731
    out = torch.nn.functional.scaled_dot_product_attention(hd(tq, Nq), hd(tk, Nk), hd(tv, Nk)).transpose(1, 2).reshape(B * Nq, a)
    (out * join(do_row)).sum().backward()
    e = dict(dq=rel(dq[:, 32:].cpu(), tq.grad), dk=rel(dkv[:, :a].cpu(), tk.grad), dv=rel(dkv[:, a:].cpu(), tv.grad))
    assert max(e.values()) < 5e-5, e
    assert torch.isnan(dq[:, :32]).all()                # the column offset is honoured
    # only one half wanted (cross-attention to a context without gradient)
    dq2 = torch.empty(B * Nq, a, device=DEV)
    HB.attention_bwd(qp, qc, kp, kc, vp, vc, do_row, lse, delta, B, H, Nq, Nk, dq=(dq2, 0))
    assert torch.equal(dq2, dq[:, 32:])
    dkv2 = torch.full((B * Nk, 2 * a), float("nan"), device=DEV)
    HB.attention_bwd(qp, qc, kp, kc, vp, vc, do_row, lse, delta, B, H, Nq, Nk, dkv=(dkv2, 0, a))
    assert torch.equal(dkv2, dkv)                       # two launches, the same bits (no atomics, fixed order)
    if not cross:                                       # round 5: dq | dk | dv as operand planes straight from the kernels = the conversion of the fp32 ones
        gp = HB.new_planes(B * Nq, 3 * a)
        HB.attention_bwd(qp, qc, kp, kc, vp, vc, do_row, lse, delta, B, H, Nq, Nk, planes=(gp, 0, a, 2 * a))
        want = HB.split(torch.cat((dq[:, 32:], dkv), -1))
        assert torch.equal(join(gp), join(want))


def test_weight_update_in_place():
    w = torch.nn.Parameter(rnd("wu", (96, 80, 3), 21).to(DEV))
    x = rnd("wu_x", (120, 96), 22)
    xp = HB.split(x.to(DEV))
    pw = HB.pack(("f", id(w)), (w,), lambda: w)
    y0 = HB.gemm_f32(pw, xp, taps=3, dil=1, seq_len=40)[:, :96].clone()
    with torch.no_grad():
        w.mul_(2.0)                                      # an optimizer step: bumps the version
    pw2 = HB.pack(("f", id(w)), (w,), lambda: w)
    assert pw2 is pw                                     # same packed object, refreshed in place
    y1 = HB.gemm_f32(pw2, xp, taps=3, dil=1, seq_len=40)[:, :96]
    assert rel(y1, 2 * y0) < 1e-5


# ------------------------------------------------------------------------------------------------ whole model
def _grads(m, fwd, x, t, seed=11, **kw):
    for p in m.parameters():
        p.grad = None
    x = x.clone().requires_grad_(True)
    y = fwd(x, t, **kw)
    w = make_input("gw", tuple(y.shape), seed=seed).to(y.device)
    (y * w).sum().backward()
    torch.cuda.synchronize()
    return y.detach(), x.grad.clone(), {k: p.grad.clone() for k, p in m.named_parameters()}


def _compare(tag, g_hip, g_ref, dx_hip, dx_ref, y_hip, y_ref, tol=1e-3):
    errs = {k: rel(g_hip[k], g_ref[k]) for k in g_ref}
    worst = max(errs.items(), key=lambda kv: kv[1])
    rec = dict(n_tensors=len(errs), worst_param=worst[0], worst_rel=worst[1], x_grad_rel=rel(dx_hip, dx_ref), out_rel=rel(y_hip, y_ref),
               median_rel=sorted(errs.values())[len(errs) // 2])
    record(f"backward_vs_reference_autograd/{tag}", rec)
    assert rec["out_rel"] < 1e-4 and rec["x_grad_rel"] < tol, rec
    bad = {k: e for k, e in errs.items() if not e < tol}
    assert not bad, bad
    return rec


@needs_ref
@pytest.mark.parametrize("tag,kw,b,n", [("d128_L6_b4_n1024", dict(dim=128, depth=6), 4, 1024), ("d512_L12_b2_n512", dict(dim=512, depth=12), 2, 512),
                                       ("d64_L2_b3_n200", dict(dim=64, depth=2), 3, 200)])
def test_every_gradient_matches_the_reference_autograd(tag, kw, b, n):
    """NS2:1635 under autograd + NS2:1886: `.grad` of EVERY parameter and of x through the HIP training path vs the reference's own
    Model (unmodified upstream source, fp32 PyTorch ops on the same GPU)"""
    ns2 = load_reference()
    m = Model(**kw, precision="hybrid")
    sd = make_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=61)
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    ref = ns2.Model(**kw)
    ref.load_state_dict(sd)
    ref = ref.to(DEV).train()
    x = make_input("x", (b, n, kw["dim"]), seed=62).to(DEV)
    t = make_input("times", (b,), seed=62, uniform=True).to(DEV)
    y1, dx1, g1 = _grads(m, m, x, t)
    y0, dx0, g0 = _grads(ref, ref, x, t)
    assert set(g1) == set(g0)
    rec = _compare(tag, g1, g0, dx1, dx0, y1, y0)
    print(tag, rec)
    # deterministic: fixed-slot reductions everywhere
    _, dx2, g2 = _grads(m, m, x, t)
    assert torch.equal(dx1, dx2) and all(torch.equal(g1[k], g2[k]) for k in g1)


def test_conditioned_model_gradients_match_the_composite():
    kw = dict(dim=128, depth=2, dim_prompt=128, condition_on_prompt=True, cond_drop_prob=0.)
    m = Model(**kw)
    sd = make_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=63)
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    b, n = 2, 300
    x = make_input("x", (b, n, 128), seed=64).to(DEV)
    t = make_input("times", (b,), seed=64, uniform=True).to(DEV)
    prompt = make_input("prompt", (b, 53, 128), seed=65).to(DEV)
    cond = make_input("cond", (b, 128, 280), seed=65).to(DEV)
    y1, dx1, g1 = _grads(m, m, x, t, prompt=prompt, cond=cond)
    comp = lambda xx, tt, **k: model_forward_autograd(m, xx, tt, **k)          # noqa: E731
    y0, dx0, g0 = _grads(m, comp, x, t, prompt=prompt, cond=cond)
    _compare("conditioned_d128_L2_vs_composite", g1, g0, dx1, dx0, y1, y0)


@pytest.mark.parametrize("tag,kw,b,n,n_p,n_c", [
    ("d96_b1_n50", dict(dim=96, depth=1, wavenet_layers=2, wavenet_stacks=1), 1, 50, 0, 0),
    ("d64_b5_n333", dict(dim=64, depth=1, wavenet_layers=3, wavenet_stacks=2), 5, 333, 0, 0),
    ("cond_pad_lm8", dict(dim=64, depth=2, wavenet_layers=2, wavenet_stacks=1, dim_prompt=64, condition_on_prompt=True, num_latents_m=8), 3, 130, 17, 90),
    ("cond_curtail_proj", dict(dim=64, depth=1, wavenet_layers=2, wavenet_stacks=1, dim_prompt=96, condition_on_prompt=True, num_latents_m=40), 2, 77, 45, 120),
])
def test_backward_ragged_and_odd_shapes(tag, kw, b, n, n_p, n_c):
    """shapes that line up with no tile: one utterance of 50 frames, 333 frames, dim 96, 8 / 40 resampler latents (cross-attention key
    counts below and beside the 64-key tile), aligned conditioning shorter (zero padded, NS2:990-992) and longer (curtailed) than the
    latents, dim_prompt != dim (proj_context) -- every gradient against the PyTorch composite"""
    m = Model(**kw)
    m.load_state_dict(make_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=91))
    m = m.to(DEV).train()
    x = make_input("x", (b, n, kw["dim"]), seed=92).to(DEV)
    t = make_input("times", (b,), seed=92, uniform=True).to(DEV)
    extra = {}
    if n_p:
        extra = dict(prompt=make_input("prompt", (b, n_p, kw["dim_prompt"]), seed=93).to(DEV),
                     cond=make_input("cond", (b, kw["dim_prompt"], n_c), seed=93).to(DEV), cond_drop_prob=0.)
    y1, dx1, g1 = _grads(m, m, x, t, **extra)
    comp = lambda xx, tt, **k: model_forward_autograd(m, xx, tt, **k)          # noqa: E731
    y0, dx0, g0 = _grads(m, comp, x, t, **extra)
    _compare("odd_shapes/" + tag, g1, g0, dx1, dx0, y1, y0)


def test_stochastic_conditioning_dropout_runs_on_the_hip_path():
    """NS2:79-85, 950-958: 0 < cond_drop_prob < 1 draws a per-utterance mask (training / validation).  Rounds 1-3 sent that call to
    the PyTorch composite; it now runs the HIP training forward (with or without autograd) -- same device RNG stream, same masks"""
    kw = dict(dim=128, depth=2, dim_prompt=128, condition_on_prompt=True, cond_drop_prob=0.5)
    m = Model(**kw, precision="hybrid")
    m.load_state_dict(make_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=71))
    m = m.to(DEV).eval()
    b, n = 6, 160
    x = make_input("x", (b, n, 128), seed=72).to(DEV)
    t = make_input("times", (b,), seed=72, uniform=True).to(DEV)
    prompt = make_input("prompt", (b, 40, 128), seed=73).to(DEV)
    cond = make_input("cond", (b, 128, n), seed=73).to(DEV)
    with torch.no_grad():
        torch.manual_seed(5)
        y_hip = m(x, t, prompt=prompt, cond=cond)                      # cond_drop_prob = 0.5 -> per-utterance masks
        torch.manual_seed(5)
        y_ref = model_forward_autograd(m, x, t, prompt=prompt, cond=cond)
        torch.manual_seed(6)
        y_other = m(x, t, prompt=prompt, cond=cond)
    assert rel(y_hip, y_ref) < 1e-4
    assert not torch.equal(y_other, y_hip)                              # another draw, another set of dropped utterances


@needs_ref
def test_reference_wrapper_trains_on_the_hip_kernels():
    """the UNMODIFIED reference NaturalSpeech2.forward (NS2:1503-1684) + loss.backward() over compat.HipBackedModel(train_backend="hip"):
    loss and gradients against the all-reference run with the same RNG state"""
    ns2 = load_reference()
    H = hip_backed_model_class(ns2.Model)
    kw = dict(dim=64, depth=2)
    m = H(**kw, train_backend="hip")
    sd = make_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=66)
    m.load_state_dict(sd)
    ref = ns2.Model(**kw)
    ref.load_state_dict(sd)
    audio = make_input("audio", (2, 96, 64), seed=67).to(DEV)
    torch.manual_seed(9)
    l_ref = ns2.NaturalSpeech2(model=ref.to(DEV), codec=None, target_sample_hz=24000).to(DEV)(audio)
    l_ref.backward()
    torch.manual_seed(9)
    loss = ns2.NaturalSpeech2(model=m.to(DEV), codec=None, target_sample_hz=24000).to(DEV)(audio)
    loss.backward()
    assert abs(loss.item() - l_ref.item()) < 1e-5 * abs(l_ref.item())
    g0 = dict(ref.named_parameters())
    errs = {k: rel(p.grad, g0[k].grad) for k, p in m.named_parameters()}
    assert max(errs.values()) < 1e-3, max(errs.items(), key=lambda kv: kv[1])


def test_optimizer_steps_through_the_hip_path_reduce_the_loss():
    """a few Adam steps of `NaturalSpeech2.forward` (v objective, min-SNR weight) on fixed data: the loss falls, the packed weights
    follow the optimizer (in-place refresh), and the same object samples afterwards"""
    torch.manual_seed(0)
    m = Model(dim=64, depth=2).to(DEV)
    d = NaturalSpeech2(m, codec=None, target_sample_hz=24000, timesteps=4).to(DEV)
    opt = torch.optim.Adam(m.parameters(), lr=3e-4)
    audio = make_input("audio", (4, 128, 64), seed=68).to(DEV)
    times = torch.tensor([0.2, 0.4, 0.6, 0.8], device=DEV)
    noise = make_input("noise", (4, 128, 64), seed=69).to(DEV)
    losses = []
    for _ in range(8):
        opt.zero_grad()
        loss = d(audio, times=times, noise=noise)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < 0.9 * losses[0], losses
    out = d.sample(length=128, batch_size=2)
    assert torch.isfinite(out).all()


def test_gradient_allreducer_over_rccl_world1_with_hip_backward():
    """the data-parallel training step over an RCCL ("nccl") process group with the HIP backward underneath: world 1 on a one-GPU box,
    so the all-reduces degenerate, but hooks, flat-buffer gradient views, ordered bucket launches on RCCL, accumulation and finish()
    are the code path of the 8-GPU run (NS2:1723-1726, 1820, 1877-1886)"""
    import torch.distributed as dist
    from naturalspeech2_pytorch_amd import distributed as D
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(29900 + os.getpid() % 90)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    try:
        torch.manual_seed(0)
        m = Model(dim=64, depth=2).to(DEV)
        d = NaturalSpeech2(m, codec=None, target_sample_hz=24000).to(DEV)
        audio = make_input("audio", (4, 96, 64), seed=81).to(DEV)
        times = torch.tensor([0.1, 0.3, 0.6, 0.9], device=DEV)
        noise = make_input("noise", (4, 96, 64), seed=82).to(DEV)
        d.zero_grad()
        d(audio, times=times, noise=noise).backward()
        ref = [p.grad.clone() for p in d.parameters()]
        red = D.GradientAllReducer(d.parameters(), bucket_bytes=1 << 16)
        assert len(red.buckets) > 3 and red._collective
        red.zero_grad()
        d(audio, times=times, noise=noise).backward()
        red.finish()
        assert all(torch.equal(p.grad, r) for p, r in zip(d.parameters(), ref))           # same kernels, fixed-slot reductions: bit-identical
        # two micro-batches, the first under accumulate(): the mean of the halves' losses
        red.zero_grad()
        with red.accumulate():
            (d(audio[:2], times=times[:2], noise=noise[:2]) / 2).backward()
        (d(audio[2:], times=times[2:], noise=noise[2:]) / 2).backward()
        red.finish()
        d2 = [p.grad.clone() for p in d.parameters()]
        red.remove()
        for p in d.parameters():
            p.grad = None
        (d(audio[:2], times=times[:2], noise=noise[:2]) / 2).backward()
        (d(audio[2:], times=times[2:], noise=noise[2:]) / 2).backward()
        assert max(rel(a, p.grad) for a, p in zip(d2, d.parameters())) < 1e-6
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["uncond_d64", "cond_d64"])
def test_gradients_match_the_committed_reference_autograd_golden(name):
    """tests/golden/grads_*.pt (make_golden.py gen_grad_case: EVERY parameter's gradient, dL/dx and the prediction from the
    unmodified reference's own autograd, generated in the build container): the HIP training path against the committed fixture --
    no reference needed on the GPU box.  Tolerance 1e-3 per tensor (seen: ~1e-5)."""
    fix = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"grads_{name}.pt"), weights_only=False)
    kw = fix["kwargs"]
    m = Model(**kw)
    m.load_state_dict(make_weights(fix["shapes"], seed=fix["weight_seed"]))
    m = m.to(DEV).train()
    b, n, d = fix["batch"], fix["n"], kw["dim"]
    x = make_input("x", (b, n, d), seed=fix["input_seed"]).to(DEV).requires_grad_(True)
    t = make_input("times", (b,), seed=fix["input_seed"], uniform=True).to(DEV)
    extra = {}
    if kw.get("condition_on_prompt"):
        extra = dict(prompt=make_input("prompt", (b, fix["n_prompt"], kw["dim_prompt"]), seed=fix["input_seed"]).to(DEV),
                     cond=make_input("cond", (b, kw["dim_prompt"], fix["n_cond"]), seed=fix["input_seed"]).to(DEV), cond_drop_prob=0.)
    y = training.model_forward_train(m, x, t, **extra)
    (y * make_input("gw", tuple(y.shape), seed=fix["loss_weight_seed"]).to(DEV)).sum().backward()
    errs = {}
    for k, p in m.named_parameters():
        r = fix["grads"][k]
        assert p.grad is not None and p.grad.shape == r.shape, k
        if r.abs().max() == 0:
            assert p.grad.abs().max().item() == 0, k
        else:
            errs[k] = rel(p.grad, r)
    worst = max(errs, key=errs.get)
    res = dict(out_rel=rel(y.detach(), fix["output"]), x_grad_rel=rel(x.grad, fix["x_grad"]), n_tensors=len(errs), worst_param=worst,
               worst_rel=errs[worst], median_rel=sorted(errs.values())[len(errs) // 2])
    record(f"backward_vs_reference_autograd/golden_{name}", res)
    assert res["out_rel"] < 1e-4 and res["x_grad_rel"] < 1e-3 and res["worst_rel"] < 1e-3, res
