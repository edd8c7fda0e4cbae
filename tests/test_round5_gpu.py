"""Round-5 GPU tests.

  * every gradient of the CONDITIONED model at the headline architecture (d512 / L12, 2 x 512 frames) against the
    reference's own autograd -- with the PerceiverResampler's backward now on the HIP Functions (training._resampler).
  * a frozen model with only x.requires_grad runs no weight-gradient GEMM and still returns the reference's dL/dx.
  (The helper loop of the tap-shared conv for half-empty column tiles -- gemm2.hip `helper` -- is covered by the shapes of
  tests/test_kernels_gpu.py::test_causal_conv whose last column tile has at most 128 valid columns: Cout = 300, 100, 341.)
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

from naturalspeech2_pytorch_amd import Model  # noqa: E402
from oracle.ref_stub import load_reference, reference_available  # noqa: E402
from tests.golden.gen import make_input, make_weights  # noqa: E402
from tests.parity_record import record  # noqa: E402

DEV = torch.device("cuda:0")
needs_ref = pytest.mark.skipif(not reference_available(), reason="needs the reference archive (oracle/_ref) or /root/reference")


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _model(kw, seed, precision):
    m = Model(**kw, precision=precision)
    sd = make_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=seed)
    m.load_state_dict(sd)
    return m.to(DEV).eval(), sd


def _grads(m, fwd, x, t, seed=11, **kw):
    for p in m.parameters():
        p.grad = None
    x = x.clone().requires_grad_(True)
    y = fwd(x, t, **kw)
    w = make_input("gw", tuple(y.shape), seed=seed).to(y.device)
    (y * w).sum().backward()
    torch.cuda.synchronize()
    return y.detach(), x.grad.clone(), {k: (p.grad.clone() if p.grad is not None else None) for k, p in m.named_parameters()}


@needs_ref
@pytest.mark.parametrize("tag,kw,b,n,n_p,n_c", [
    ("cond_d512_L12_b2_n512", dict(dim=512, depth=12, dim_prompt=512, condition_on_prompt=True, cond_drop_prob=0.), 2, 512, 103, 512),
    ("cond_d128_L2_b3_n300_proj", dict(dim=128, depth=2, dim_prompt=96, condition_on_prompt=True, cond_drop_prob=0.), 3, 300, 53, 280)])
def test_every_gradient_of_the_conditioned_model_matches_the_reference_autograd(tag, kw, b, n, n_p, n_c):
    """BASELINE config 3's architecture under autograd (NS2:1635, 1886): `.grad` of EVERY parameter -- the perceiver resampler's, the
    cross attentions', `null_*` -- and of x through the HIP training path against the unmodified upstream `Model` on the same GPU"""
    ns2 = load_reference()
    m = Model(**kw, precision="hybrid")
    sd = make_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=81)
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    ref = ns2.Model(**kw)
    ref.load_state_dict(sd)
    ref = ref.to(DEV).train()
    x = make_input("x", (b, n, kw["dim"]), seed=82).to(DEV)
    t = make_input("times", (b,), seed=82, uniform=True).to(DEV)
    prompt = make_input("prompt", (b, n_p, kw["dim_prompt"]), seed=83).to(DEV)
    cond = make_input("cond", (b, kw["dim_prompt"], n_c), seed=83).to(DEV)
    y1, dx1, g1 = _grads(m, m, x, t, prompt=prompt, cond=cond)
    y0, dx0, g0 = _grads(ref, ref, x, t, prompt=prompt, cond=cond)
    assert set(g1) == set(g0)
    errs = {}
    for k, g in g0.items():
        if g is None:                                   # null_* parameters: selected by torch.where with a false mask
            assert g1[k] is None or float(g1[k].abs().max()) == 0.0, k
            continue
        assert g1[k] is not None, k
        errs[k] = rel(g1[k], g) if float(g.abs().max()) > 0 else float(g1[k].abs().max())
    worst = max(errs.items(), key=lambda kv: kv[1])
    rec = dict(n_tensors=len(errs), worst_param=worst[0], worst_rel=worst[1], x_grad_rel=rel(dx1, dx0), out_rel=rel(y1, y0),
               median_rel=sorted(errs.values())[len(errs) // 2],
               worst_resampler=max((e for k, e in errs.items() if "perceiver_resampler" in k), default=0.0))
    record(f"backward_vs_reference_autograd/{tag}", rec)
    print(tag, rec)
    assert rec["out_rel"] < 1e-4 and rec["x_grad_rel"] < 1e-3, rec
    bad = {k: e for k, e in errs.items() if not e < 1e-3}
    assert not bad, bad


def test_frozen_model_returns_dx_without_weight_gradients():
    kw = dict(dim=128, depth=2)
    m, _ = _model(kw, 91, "hybrid")
    m = m.train()
    x = make_input("x", (2, 256, 128), seed=92).to(DEV)
    t = make_input("times", (2,), seed=92, uniform=True).to(DEV)
    _, dx_all, g_all = _grads(m, m, x, t)
    for p in m.parameters():
        p.requires_grad_(False)
    _, dx_frozen, g_frozen = _grads(m, m, x, t)
    assert all(g is None for g in g_frozen.values())
    assert torch.equal(dx_all, dx_frozen)               # the same kernels in the same order produce dL/dx


# ------------------------------------------------------------------------------------------------ the mixed training arithmetic
# (VERDICT r4 missing #2 / item 3c: the reference trains under accelerate's fp16 mixed precision, NS2:1710-1711; train_precision="mixed"
# runs the GEMMs of forward, dgrad and wgrad as one IEEE-half product + fp8 correction terms on FMT_H8 operands under a loss scale)
from naturalspeech2_pytorch_amd import ops, training  # noqa: E402
from tests.emu_backend import EP, EmuBackend, mixed_mm, rup  # noqa: E402

HB4 = training.HipBackend(4)
EB = EmuBackend()


def pjoin(p):
    return ops.join(p).cpu()


def tjoin(tp):
    return ops.join(ops.Planes(tp.buf, tp.rows, tp.ld, True, "h8" if tp.precision == 4 else "bf16")).cpu()


@pytest.mark.parametrize("M,C,seq_len", [(256, 64, 64), (200, 72, 40), (1024, 1365, 256), (96, 512, 96)])
def test_mixed_grad_prep_writes_h8_rows_and_their_exact_transposition(M, C, seq_len):
    x = make_input("gp", (M, C), seed=1) * 3.0
    ld = rup(C, 4) + 4
    xp = torch.full((M, ld), float("nan"))
    xp[:, :C] = x
    row, tp, cs = HB4.grad_prep(xp.to(DEV), C, want_row=True, want_t=True, want_colsum=True, t_rows=rup(C, 256))
    assert row.fmt == "h8" and tp.precision == 4
    rj, tj = pjoin(row), tjoin(tp)
    assert rel(rj[:, :C], x) < 1e-4 and float(rj[:, C:].abs().sum()) == 0.0               # half + e5m2 remainder: ~2^-14 per element
    assert tj.shape == (rup(C, 256), rup(M, 32))
    assert torch.equal(tj[:C, :M], rj[:, :C].t())                                          # the SAME three parts, transposed: exact
    assert float(tj[C:].abs().sum()) == 0.0 and float(tj[:, M:].abs().sum()) == 0.0
    assert rel(cs.cpu(), x.sum(0)) < 1e-5
    if C % 8 == 0:                                                                         # attention operands stay bf16 hi / lo
        r2, t2, _ = HB4.grad_prep(xp.to(DEV), C, want_row=True, want_t=True, seq_len=seq_len, per_batch=True, attn=True)
        assert r2.fmt == "bf16" and t2.precision == 3 and rel(pjoin(r2)[:, :C], x) < 1e-5


@pytest.mark.parametrize("M,C,seq_len,shifts", [(256, 64, 64, (2, 1, 0)), (240, 96, 40, (8, 4, 0)), (512, 1376, 256, (2, 1, 0)),
                                               (2048, 128, 1024, (256, 128, 0)), (192, 64, 0, (0,))])
def test_mixed_planes_transpose_moves_h8_lines_exactly(M, C, seq_len, shifts):
    x = make_input("pt", (M, C + 32), seed=2)
    p = HB4.split(x.to(DEV))
    assert p.fmt == "h8"
    xr = pjoin(p)
    tp = HB4.transpose(p, 32, C, seq_len, shifts)
    et = EB.transpose(EP(xr), 32, C, seq_len, shifts)
    tj = tjoin(tp)
    assert tj.shape == et.t.shape and torch.equal(tj, et.t)


@pytest.mark.parametrize("R,K,T,M", [(512, 512, 1, 4096), (64, 64, 3, 480), (1365, 1365, 3, 2048), (1536, 512, 1, 32768)])
def test_mixed_wgrad(R, K, T, M):
    dy = make_input("wg_dy", (M, R), seed=3) * 0.5
    x = make_input("wg_x", (M, rup(K, 32)), seed=4)
    x[:, K:] = 0
    seq = M // 2 if T > 1 else 0
    shifts = tuple((T - 1 - t) * 2 for t in range(T))
    _, dyt, _ = HB4.grad_prep(dy.to(DEV), R, want_t=True)
    xt = HB4.transpose(HB4.split(x.to(DEV)), 0, rup(K, 32), seq, shifts)
    dw = HB4.wgrad(dyt, xt, R, T, K).cpu()
    _, edyt, _ = EB.grad_prep(dy, R, want_t=True)
    ext = EB.transpose(EP(x), 0, rup(K, 32), seq, shifts)
    ref = EB.wgrad(edyt, ext, R, T, K)
    assert dw.shape == (R, K, T) and rel(dw, ref) < 1e-4, rel(dw, ref)
    assert torch.equal(dw, HB4.wgrad(dyt, xt, R, T, K).cpu())


def test_mixed_attention_output_and_qkv_formats():
    """the attention stays bf16 x3 on bf16 operands (written by the precision-4 q | k | v GEMM through ns2_linear_split_as); its
    output is the FMT_H8 operand of the out-projection, and delta reads it back"""
    B, H, N, d = 2, 8, 256, 512
    a = H * 64
    xn = HB4.split((make_input("xa", (B * N, d), seed=5)).to(DEV))
    w = torch.nn.Parameter((make_input("wa", (3 * a, d), seed=6) * d ** -0.5).to(DEV))
    pw = HB4.pack(("t", id(w)), (w,), lambda: w)
    qkv = HB4.gemm_split(pw, xn, attn=True)
    assert qkv.fmt == "bf16" and qkv.has_lo
    ref = mixed_mm(pjoin(xn), w.detach().cpu().t())
    assert rel(pjoin(qkv), ref) < 2e-5
    vt = HB4.transpose(qkv, 2 * a, a, N, per_batch=True)
    o, lse = HB4.attention(qkv, 0, qkv, a, vt, B, H, N, N)
    assert o.fmt == "h8"
    qr = pjoin(qkv)
    eo, else_ = EB.attention(EP(qr[:, :a].clone()), 0, EP(qr[:, a:2 * a].clone()), 0, EB.transpose(EP(qr[:, 2 * a:].clone()), 0, a, N, per_batch=True),
                             B, H, N, N)
    assert rel(pjoin(o), eo.t) < 1e-4 and (lse.cpu() - else_).abs().max() < 1e-4
    do = make_input("do", (B * N, a), seed=7)
    delta = HB4.attention_delta(do.to(DEV), o, B, H, N)
    assert rel(delta.cpu(), EB.attention_delta(do, EP(pjoin(o)), B, H, N)) < 1e-5


@needs_ref
@pytest.mark.parametrize("tag,kw,b,n", [("d128_L6_b4_n1024", dict(dim=128, depth=6), 4, 1024), ("d512_L12_b2_n512", dict(dim=512, depth=12), 2, 512),
                                       ("d64_L2_b3_n200", dict(dim=64, depth=2), 3, 200)])
def test_every_gradient_in_the_mixed_training_arithmetic_matches_the_reference_autograd(tag, kw, b, n):
    """train_precision="mixed" against the reference's own fp32 autograd, every parameter and x, with an upstream gradient of the
    magnitude a mean-reduced loss produces (1e-7 per element: unusable in IEEE half without the loss scale)"""
    from naturalspeech2_pytorch_amd.ops import saturation_count
    ns2 = load_reference()
    m = Model(**kw, precision="hybrid")
    sd = make_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=61)
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    m.train_precision = "mixed"
    ref = ns2.Model(**kw)
    ref.load_state_dict(sd)
    ref = ref.to(DEV).train()
    x = make_input("x", (b, n, kw["dim"]), seed=62).to(DEV)
    t = make_input("times", (b,), seed=62, uniform=True).to(DEV)

    def grads(mod):
        for p in mod.parameters():
            p.grad = None
        xx = x.clone().requires_grad_(True)
        y = mod(xx, t)
        (y * make_input("gw", tuple(y.shape), seed=11).to(DEV) * 1e-7).sum().backward()
        torch.cuda.synchronize()
        return y.detach(), xx.grad.clone(), {k: p.grad.clone() for k, p in mod.named_parameters()}

    sat0 = saturation_count(reset=False, device=DEV)
    y1, dx1, g1 = grads(m)
    assert saturation_count(reset=False, device=DEV) == sat0, "a value left the IEEE-half range under the chosen loss scale"
    y0, dx0, g0 = grads(ref)
    errs = {k: rel(g1[k], g0[k]) for k in g0}
    worst = max(errs.items(), key=lambda kv: kv[1])
    rec = dict(n_tensors=len(errs), worst_param=worst[0], worst_rel=worst[1], x_grad_rel=rel(dx1, dx0), out_rel=rel(y1, y0),
               median_rel=sorted(errs.values())[len(errs) // 2], loss_scale=float(m._last_loss_scale.s))
    record(f"backward_vs_reference_autograd/mixed/{tag}", rec)
    print(tag, rec)
    assert rec["out_rel"] < 2.5e-4 and rec["x_grad_rel"] < 1e-3, rec
    bad = {k: e for k, e in errs.items() if not e < 1e-3}
    assert not bad, bad
    _, dx2, g2 = grads(m)                                  # deterministic also here: fixed slots, and the scale is a function of the input
    assert torch.equal(dx1, dx2) and all(torch.equal(g1[k], g2[k]) for k in g1)


# ------------------------------------------------------------------------------------------------ the codec's narrow residual block
@pytest.mark.parametrize("B,T,inp", [(2, 300, 0), (3, 77, 2), (1, 1000, 0), (5, 3, 1), (2, 256, 0), (1, 257, 3)])
def test_seanet_resblock_narrow_matches_hf(B, T, inp):
    """ns2_seanet_resblock_narrow (EnCodec's residual block at C = 32 in one fp32 pass: conv1 k = 3 with reflect padding, both ELUs,
    conv2 and the shortcut) against HF's own EncodecResnetBlock on the same weights: rows across workgroup borders, utterance starts
    inside a workgroup, an input layout with prefix rows, T = 3 (the two reflected rows are the whole past)"""
    tf = pytest.importorskip("transformers")
    from transformers.models.encodec.modeling_encodec import EncodecResnetBlock
    from naturalspeech2_pytorch_amd import _lib, seanet
    torch.manual_seed(3)
    blk = EncodecResnetBlock(tf.EncodecConfig(), dim=32, dilations=[1, 1]).eval().to(DEV)
    net = seanet._SEANetHIP.__new__(seanet._SEANetHIP)
    torch.nn.Module.__init__(net)
    net.precision = "exact"
    nr = net._pack_resblock(blk)["narrow"]
    x = (make_input("rb", (B * (inp + T), 32), seed=31) * 1.5).to(DEV)
    y = torch.full((B * T, 32), float("nan"), device=DEV)
    lib = _lib.load()
    rc = lib.ns2_seanet_resblock_narrow(x.data_ptr(), 32, inp, B, T, 32, nr["w1p"].data_ptr(), nr["b1"].data_ptr(), nr["w2p"].data_ptr(),
                                        nr["wsp"].data_ptr(), nr["b2s"].data_ptr(), y.data_ptr(), 32, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    xv = x.reshape(B, inp + T, 32)[:, inp:]
    with torch.no_grad():
        ref = blk(xv.transpose(1, 2)).transpose(1, 2).reshape(B * T, 32)
    e = rel(y, ref)
    assert e < 3e-6, e
    # another width is not served: nothing launched, the caller takes the GEMM path
    assert lib.ns2_seanet_resblock_narrow(x.data_ptr(), 32, inp, B, T, 48, nr["w1p"].data_ptr(), nr["b1"].data_ptr(), nr["w2p"].data_ptr(),
                                          nr["wsp"].data_ptr(), nr["b2s"].data_ptr(), y.data_ptr(), 32, torch.cuda.current_stream().cuda_stream) == _lib.NS2_UNAVAILABLE


# ------------------------------------------------------------------------------------------------ weight gradients from the row planes
# ("wgrad operands through LDS transpose loads ... delete the tplanes passes".  gemm2.hip TR reads both operands of
# dW = dY^T X from the token-major planes with ds_read_b64_tr_b16 / _tr_b8; the conv taps are row offsets of the loads.)
HB3 = training.HipBackend(3)


@pytest.mark.parametrize("prec", [3, 4])
@pytest.mark.parametrize("R,K,T,M,dil,seq", [(512, 512, 1, 4096, 1, 0), (300, 200, 1, 1000, 1, 0), (64, 64, 3, 480, 2, 160), (1365, 1365, 3, 2048, 1, 1024),
                                             (512, 512, 3, 4096, 64, 1024), (128, 512, 1, 2048, 1, 0), (1536, 512, 1, 32768, 1, 0),
                                             (96, 96, 3, 777 * 2, 5, 777)])
def test_wgrad_from_row_planes(prec, R, K, T, M, dil, seq):
    hb = HB3 if prec == 3 else HB4
    dy = make_input("wr_dy", (M, R), seed=5) * 0.5
    x = make_input("wr_x", (M, rup(K, 32) + 32), seed=6)                 # operand planes wider than K: the columns beyond are another tensor's
    x[:, K:rup(K, 32)] = 0
    dyp, xp = hb.split(dy.to(DEV)), hb.split(x.to(DEV))
    dw = hb.wgrad_rows(dyp, xp, R, T, K, dil, seq).cpu()
    # reference: the same rounded operand values (what the planes hold), wide accumulation
    dyr, xr = pjoin(dyp)[:, :R], pjoin(xp)[:, :K]
    ref = torch.zeros(R, K, T, dtype=torch.float64)
    for t in range(T):
        ref[:, :, t] = dyr.double().t() @ EB._shifted(xr, seq if T > 1 else 0, (T - 1 - t) * dil).double()
    assert dw.shape == (R, K, T)
    e = rel(dw, ref)
    assert e < (2e-5 if prec == 3 else 1e-4), e                           # bf16 x3: ~2^-16 per product; mixed: first-order terms in e5m2
    assert torch.equal(dw, hb.wgrad_rows(dyp, xp, R, T, K, dil, seq).cpu())            # fixed slots, fixed order
    # and against the transposed route (same operands, same arithmetic, another order of the fp32 sums)
    _, dyt, _ = hb.grad_prep(dy.to(DEV), R, want_t=True)
    xt = hb.transpose(xp, 0, rup(K, 32), seq if T > 1 else 0, tuple((T - 1 - t) * dil for t in range(T)))
    old = hb.wgrad(dyt, xt, R, T, K).cpu()
    assert rel(dw, old) < (2e-6 if prec == 3 else 2e-5), rel(dw, old)


# ------------------------------------------------------------------------------------------------ optimizers that do not bump version counters
def test_packs_follow_freezing_and_unfreezing():
    """the one-launch re-pack covers the TRAINABLE packs of a pass; a block frozen for three steps and then unfrozen must train exactly as
    it does on the PyTorch composite (the table is rebuilt when trainability changes; a frozen pack is never marked fresh by the launch)"""
    from naturalspeech2_pytorch_amd import NaturalSpeech2
    traj = {}
    for backend in ("hip", "composite"):
        torch.manual_seed(0)
        m = Model(dim=128, depth=2).to(DEV).train()
        m.train_backend = backend
        d = NaturalSpeech2(m, codec=None, target_sample_hz=24000).to(DEV)
        frozen = [p for n, p in m.named_parameters() if n.startswith("wavenet")]
        for p in frozen:
            p.requires_grad_(False)
        opt = torch.optim.Adam(m.parameters(), lr=1e-4, fused=True)
        g = torch.Generator().manual_seed(1)
        audio, times, noise = torch.randn(2, 256, 128, generator=g).to(DEV), torch.rand(2, generator=g).to(DEV), torch.randn(2, 256, 128, generator=g).to(DEV)
        ls = []
        for i in range(8):
            if i == 3:
                for p in frozen:
                    p.requires_grad_(True)
            opt.zero_grad(set_to_none=True)
            loss = d(audio, times=times, noise=noise)
            loss.backward()
            opt.step()
            ls.append(float(loss.detach()))
        traj[backend] = ls
    worst = max(abs(a - b) / abs(b) for a, b in zip(traj["hip"], traj["composite"]))
    assert worst < 2e-3, traj


@pytest.mark.parametrize("fused", [True, False])
def test_fused_optimizer_steps_reach_the_packed_weights(fused):
    """torch.optim.Adam(fused=True) updates parameters in place WITHOUT moving their version counters; the packed GEMM weights of the
    HIP training path must follow anyway (training._PackedCache: refreshed at the first use of every pass).  The same seven steps on the
    PyTorch composite are the witness: same losses step by step (the first step's losses are equal to 1e-6; Adam's sign-like first
    updates then amplify the 1e-5 gradient differences somewhat)."""
    from naturalspeech2_pytorch_amd import NaturalSpeech2
    traj = {}
    for backend in ("hip", "composite"):
        torch.manual_seed(0)
        m = Model(dim=128, depth=2).to(DEV).train()
        m.train_backend = backend
        d = NaturalSpeech2(m, codec=None, target_sample_hz=24000).to(DEV)
        opt = torch.optim.Adam(m.parameters(), lr=1e-4, fused=fused or None)
        g = torch.Generator().manual_seed(1)
        audio, times, noise = torch.randn(2, 256, 128, generator=g).to(DEV), torch.rand(2, generator=g).to(DEV), torch.randn(2, 256, 128, generator=g).to(DEV)
        ls = []
        for _ in range(7):
            opt.zero_grad(set_to_none=True)
            loss = d(audio, times=times, noise=noise)
            loss.backward()
            opt.step()
            ls.append(float(loss.detach()))
        traj[backend] = ls
    h, c = traj["hip"], traj["composite"]
    assert abs(h[0] - c[0]) < 1e-5 * abs(c[0])
    assert h[-1] < 0.97 * h[0] and c[-1] < 0.97 * c[0]                      # both actually train
    worst = max(abs(a - b) / abs(b) for a, b in zip(h, c))
    record(f"training_trajectory_hip_vs_composite/adam_fused_{int(fused)}/worst_rel_loss_difference_over_7_steps", worst)
    assert worst < 2e-3, (h, c)                                             # stale packs gave 4e-2 at step 1 already


# ------------------------------------------------------------------------------------------------ residual update + RMSNorm in one launch (dim = 128)
@pytest.mark.parametrize("prec,tol", [("exact", 1e-4), ("hybrid", 2.5e-4), ("mixed", 2.5e-4), ("half", 1e-3), ("hybrid_ff", 4e-4)])
@pytest.mark.parametrize("conditioned", [False, True])
def test_whole_row_epilogue_runs_the_norm_at_dim_128(prec, tol, conditioned):
    """At dim = 128 the GEMMs that update the residual stream own whole rows (N == BN == 128) and run the following RMSNorm in their
    epilogue when M % 128 == 0 (gemm.hip EPI_F32 nrm_*; model_exec.cpp update_then_norm): adaptive norms in front of self attention,
    cross attention and feed-forward, and to_pred's learned-gamma norm.  Against the fp32 oracle, and a batch whose M is NOT a multiple
    of 128 (separate rmsnorm_kernel) must give the same utterances within the two paths' rounding."""
    from oracle import ns2_oracle as O
    kw = dict(dim=128, depth=2, dim_prompt=128, condition_on_prompt=True) if conditioned else dict(dim=128, depth=2)
    m, sd = _model(kw, seed=31, precision=prec)
    x = make_input("x", (3, 256, 128), seed=32)                       # M = 768: fused;  the first utterance alone, 250 frames: not
    t = make_input("times", (3,), seed=32, uniform=True)
    extra = {}
    if conditioned:
        extra = dict(prompt=make_input("prompt", (3, 40, 128), seed=33), cond=make_input("cond", (3, 128, 256), seed=33))
    with torch.no_grad():
        y = m(x.to(DEV), t.to(DEV), **{k: v.to(DEV) for k, v in extra.items()}).cpu()
        ref = O.model_forward(sd, x, t, **extra)
        y1 = m(x[:1, :250].to(DEV), t[:1].to(DEV), **{k: (v[:1, :, :250] if k == "cond" else v[:1]).to(DEV) for k, v in extra.items()}).cpu()
        ref1 = O.model_forward(sd, x[:1, :250], t[:1], **{k: (v[:1, :, :250] if k == "cond" else v[:1]) for k, v in extra.items()})
        # one utterance of 256 frames: every N = 128 product is split over K and the FINISHING launch runs the norm (gemm.hip
        # splitk_finish_f32_norm_kernel); it must reproduce the utterance of the full batch (rows are independent)
        y2 = m(x[:1].to(DEV), t[:1].to(DEV), **{k: v[:1].to(DEV) for k, v in extra.items()}).cpu()
    e, e1, e2 = rel(y, ref), rel(y1, ref1), rel(y2, ref[:1])
    record(f"fused_norm_d128/{'cond' if conditioned else 'uncond'}/{prec}", max(e, e2))
    assert e < tol and e1 < tol and e2 < tol, (e, e1, e2)
