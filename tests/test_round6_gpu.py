"""Round-6 GPU tests.

  * the dedicated FF causal conv kernel (csrc/ffconv_kernel.h; FeedForward's CausalConv1d(f, f, 3), NS2:1016 / 583-595) against
    (a) the general kernels on the same operands -- bit for bit: same products, same summation order -- and (b) an fp64 restatement of
    the conv on the half-rounded operands, over shapes that reach every code path: full / half-valid / partly valid column tiles,
    first tiles of utterances, K tails of 0 / 32 / 64 padded columns, one and many row tiles, both output formats;
  * the executor on it: a hybrid / half / hybrid_ff model step equals the step with the kernel switched off, bit for bit.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

from naturalspeech2_pytorch_amd import Model, _lib, ops  # noqa: E402
from tests.golden.gen import make_input, make_weights  # noqa: E402

DEV = torch.device("cuda:0")


def _force(k):
    _lib.check(_lib.load().ns2_debug_force_gemm(k), "ns2_debug_force_gemm")


def _conv_ref64(xh, wh, bias, B, N):
    """fp64 causal conv (NS2:583-595: left padding dil * (k - 1) = 2) on operands already rounded to half: xh [B*N, C], wh [R, C, 3]"""
    x = xh.double().reshape(B, N, -1).transpose(1, 2)
    y = torch.nn.functional.conv1d(torch.nn.functional.pad(x, (2, 0)), wh.double(), bias.double())
    return y.transpose(1, 2).reshape(B * N, -1)


CASES = [
    # B, N, C_in, C_out          what the shape reaches
    (8, 1024, 1365, 1365),     # the headline widths: 5 full column tiles + a half-valid one, K tail of 32 padded columns
    (4, 256, 341, 341),        # dim 128: one full + one half-valid column tile, every row tile starts an utterance
    (3, 512, 170, 170),        # K tail: the last K tile is padding only (Cp = 192 in rows of 256)
    (2, 768, 682, 300),        # Cp = 704 in rows of 768; 300 output columns: a full tile + 44 columns
    (1, 256, 96, 96),          # one block, one K-tile pair, a half-valid only column tile
    (5, 256, 200, 200),        # a single partly valid column tile (200 > 128: every wave active, the last quarter beyond N)
    (2, 512, 128, 640),        # lda == Cp (no padding), 2.5 column tiles
]


@pytest.mark.parametrize("B,N,C,R", CASES)
@pytest.mark.parametrize("out_precision", [4, 2])
def test_ffconv3_equals_the_general_kernel_and_the_fp64_conv(B, N, C, R, out_precision):
    g = torch.Generator().manual_seed(1000 + C + R)
    x = torch.randn(B * N, C, generator=g)
    w = torch.randn(R, C, 3, generator=g) * (1.0 / (3 * C) ** 0.5)
    bias = torch.randn(R, generator=g)
    ld = ops.conv3_input_ld(C)
    assert ld % 128 == 0 and 0 <= ld - ops.round_up(C, 32) < 128
    a = ops.split(x.to(DEV), ldo=ld, precision=2)
    # poison the padding columns of the activations: the kernel fetches them but must never multiply them
    a.buf.view(B * N, ld)[:, ops.round_up(C, 32):] = float("nan")
    pw = ops.PackedWeight(w.to(DEV), precision=2).tile_conv3()
    bd = bias.to(DEV)

    def run():
        return ops.linear_split(pw, a, bias=bd, conv_taps=3, dilation=1, seq_len=N, precision=2, out_precision=out_precision)

    try:
        _force(5)                    # the dedicated kernel whatever the size
        new = run()
        new2 = run()
        _force(2)                    # gemm2_kernel (256 x 256 tiles), the kernel it replaces; the 128 x 128 kernel sums in another order
        old = run()
    finally:
        _force(0)
    assert torch.equal(new.buf, new2.buf), "two launches of the dedicated kernel differ"
    assert torch.equal(new.buf, old.buf), "dedicated FF-conv kernel and general kernel differ"
    got = ops.join(new, R).cpu().double()
    ref = _conv_ref64(x.half().float(), w.half().float(), bias, B, N)
    err = ((got - ref).norm() / ref.norm()).item()
    assert err < (6e-4 if out_precision == 2 else 1e-5), err      # the output format's own rounding: half 2^-11, FMT_H8 half + l8 ~ 2^-19


def test_ffconv3_takes_the_big_shapes_by_itself():
    """without any hook: a product of more than 64 output tiles runs on the dedicated kernel (same bits as with it switched off)"""
    B, N, C = 32, 1024, 341
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B * N, C, generator=g)
    w = torch.randn(C, C, 3, generator=g) * 0.03
    a = ops.split(x.to(DEV), ldo=ops.conv3_input_ld(C), precision=2)
    pw = ops.PackedWeight(w.to(DEV), precision=2).tile_conv3()
    new = ops.linear_split(pw, a, conv_taps=3, dilation=1, seq_len=N, precision=2, out_precision=4)
    try:
        _force(4)
        old = ops.linear_split(pw, a, conv_taps=3, dilation=1, seq_len=N, precision=2, out_precision=4)
    finally:
        _force(0)
    assert torch.equal(new.buf, old.buf)


@pytest.mark.parametrize("precision", ["hybrid", "half", "hybrid_ff"])
@pytest.mark.parametrize("kw,B,N", [(dict(dim=128, depth=2), 32, 1024), (dict(dim=512, depth=1), 8, 1024)])
def test_model_step_is_bit_identical_with_and_without_the_ffconv3_kernel(precision, kw, B, N):
    m = Model(**kw, precision=precision)
    sd = make_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=3)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    x = make_input("x", (B, N, kw["dim"]), seed=4).to(DEV)
    t = make_input("times", (B,), seed=4, uniform=True).to(DEV)
    with torch.no_grad():
        try:
            _force(0)
            y_new = m(x, t).clone()
            _force(4)
            y_old = m(x, t).clone()
        finally:
            _force(0)
    assert torch.isfinite(y_new).all()
    assert torch.equal(y_new, y_old)


# ---------------------------------------------------------------------------------------------- the lean mixed linear kernel (gemm3_kernel.h)
LIN_CASES = [
    # M, K, N            what the shape reaches
    (2048, 512, 1536),   # q | k | v widths: even tile count, 6 full column tiles
    (1024, 1365, 512),   # FF-out: K = 1376 = 43 tiles (odd: the peeled steady tile), 2 column tiles
    (512, 96, 200),      # the smallest K (3 tiles: peel + tail only), a partly valid column tile
    (768, 160, 300),     # 5 tiles (peel + one pair + tail), N = 256 + 44
    (256, 128, 128),     # 4 tiles (no steady pair), one row tile, half a column tile
]


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("M,K,N", LIN_CASES)
def test_gemm3_f32_split_equal_gemm2_bit_for_bit(M, K, N):
    g = torch.Generator().manual_seed(7 + K + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * K ** -0.5
    bias, resid = torch.randn(N, generator=g).to(DEV), torch.randn(M, N, generator=g).to(DEV)
    a = ops.split(x.to(DEV), precision=4)
    pw = ops.PackedWeight(w.to(DEV), precision=4).tile_linear()
    outs = {}
    try:
        for k in (5, 2):
            _force(k)
            outs[k] = (ops.linear_f32(pw, a, bias=bias, resid=resid, precision=4), ops.linear_split(pw, a, bias=bias, precision=4).buf.clone(),
                       ops.linear_f32(pw, a, precision=4, act=1))
    finally:
        _force(0)
    for i in range(3):
        assert torch.equal(outs[5][i], outs[2][i]), f"output {i} of the lean kernel differs from gemm2_kernel's"
    ref = x.double() @ w.double().t() + bias.cpu().double() + resid.cpu().double()
    assert _rel(outs[5][0].cpu(), ref) < 2e-4


@pytest.mark.parametrize("B,N,d", [(4, 256, 512), (2, 512, 128)])
def test_gemm3_qkv_and_geglu_equal_gemm2_bit_for_bit(B, N, d):
    g = torch.Generator().manual_seed(11 + d)
    M, f = B * N, int(d * 8 / 3)
    x = torch.randn(M, d, generator=g)
    a = ops.split(x.to(DEV), precision=4)
    wq = ops.PackedWeight((torch.randn(3 * 512, d, generator=g) * d ** -0.5).to(DEV), precision=4).tile_linear()
    w1 = ops.PackedWeight((torch.randn(2 * f, d, generator=g) * d ** -0.5).to(DEV), geglu=True, precision=4).tile_linear()
    pb = ops.geglu_pack_bias(torch.randn(2 * f, generator=g).to(DEV), f)
    outs = {}
    try:
        for k in (5, 2):
            _force(k)
            q, vt = ops.linear_qkv(wq, a, seq_len=N, split_col=1024, precision=4)
            h = ops.linear_geglu(w1, a, pb, precision=4)
            outs[k] = (q.buf.clone(), vt.buf.clone(), h.buf.clone())
    finally:
        _force(0)
    for i in range(3):
        assert torch.equal(outs[5][i], outs[2][i])


@pytest.mark.parametrize("precision", ["hybrid", "mixed"])
@pytest.mark.parametrize("kw,B,N,cond", [(dict(dim=128, depth=2), 16, 1024, False), (dict(dim=512, depth=1), 4, 1024, False),
                                         (dict(dim=512, depth=1, dim_prompt=512, condition_on_prompt=True), 4, 512, True),
                                         (dict(dim=256, depth=1), 16, 512, False),          # wavenet3 with one column tile, K loops of 8 / 12 tiles
                                         (dict(dim=384, depth=1, wavenet_layers=4), 8, 768, False)])   # 1.5 column tiles: wavenet3 not eligible, the others are
def test_model_step_is_bit_identical_with_and_without_the_round6_kernels(precision, kw, B, N, cond):
    m = Model(**kw, precision=precision)
    sd = make_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=3)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    x = make_input("x", (B, N, kw["dim"]), seed=4).to(DEV)
    t = make_input("times", (B,), seed=4, uniform=True).to(DEV)
    extra = {}
    if cond:
        extra = dict(prompt=make_input("prompt", (B, 40, 512), seed=6).to(DEV), cond=make_input("cond", (B, 512, N), seed=7).to(DEV))
    with torch.no_grad():
        try:
            _force(0)
            y_new = m(x, t, **extra).clone()
            _force(4)
            y_old = m(x, t, **extra).clone()
        finally:
            _force(0)
    assert torch.isfinite(y_new).all()
    assert torch.equal(y_new, y_old)


# ---------------------------------------------------------------------------------------------- parity at BASELINE's STATED sizes
def _gpu_oracle_pinned(sd, fn_gpu, fn_cpu_one):
    """the oracle's code on GPU tensors (plain PyTorch fp32 ops) for the full batch, pinned to the CPU oracle on one utterance"""
    ref = fn_gpu()
    pin = _rel(ref[:1].cpu(), fn_cpu_one())
    assert pin < 5e-6, f"GPU-resident oracle drifted from the CPU oracle: {pin}"
    return ref


def test_config3_conditioned_at_its_stated_size_4x1024_cfg():
    """BASELINE config 3 as stated: dim=512 depth=12 dim_prompt=512 condition_on_prompt, batch 4 x 1024 frames, prompt of 103 frames (what a
    32 768-sample prompt becomes), aligned conditioning [4, 512, 1024], classifier-free guidance 1.3 (NS2:914-927)"""
    from oracle import ns2_oracle as O
    from tests.parity_record import record
    kw = dict(dim=512, depth=12, dim_prompt=512, condition_on_prompt=True)
    b, n = 4, 1024
    x = make_input("x", (b, n, 512), seed=31)
    t = make_input("times", (b,), seed=31, uniform=True)
    prompt = make_input("prompt", (b, 103, 512), seed=32)
    cond = make_input("cond", (b, 512, n), seed=33)
    errs = {}
    ref = None
    for precision, tol in (("exact", 1e-4), ("hybrid", 2.5e-4)):
        m = Model(**kw, precision=precision)
        sd = make_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=30)
        m.load_state_dict(sd)
        m = m.to(DEV).eval()
        if ref is None:
            sdg = {k: v.to(DEV) for k, v in sd.items()}
            with torch.no_grad():
                ref = _gpu_oracle_pinned(
                    sd, lambda: O.model_forward_with_cond_scale(sdg, x.to(DEV), t.to(DEV), prompt.to(DEV), cond.to(DEV), 1.3),
                    lambda: O.model_forward_with_cond_scale(sd, x[:1], t[:1], prompt[:1], cond[:1], 1.3))
        with torch.no_grad():
            y = m.forward_with_cond_scale(x.to(DEV), t.to(DEV), prompt=prompt.to(DEV), cond=cond.to(DEV), cond_scale=1.3)
        errs[precision] = _rel(y, ref)
        assert torch.isfinite(y).all() and errs[precision] < tol, (precision, errs[precision])
        del m
    record("config3_stated_size_4x1024_prompt103_cfg1.3", errs)


_HD_REF = {}


@pytest.mark.parametrize("precision,tol", [("exact", 5e-5), ("hybrid", 3e-4), ("half", 1e-3)])
@pytest.mark.parametrize("dim_head,heads", [(32, 8), (128, 2)])
def test_head_dims_32_and_128_on_whole_row_tiles(dim_head, heads, precision, tol):
    """VERDICT r5 #5b beyond the golden sizes: dim 256, 2 x 512 frames, conditioned with guidance -- the shapes at which the executor takes the
    256-row kernels (lean q | k | v with the LDS-transposed V^T, lean Wavenet block ...) -- with heads of 32 and of 128, against the oracle"""
    from oracle import ns2_oracle as O
    kw = dict(dim=256, depth=2, dim_head=dim_head, heads=heads, dim_prompt=64, condition_on_prompt=True, num_latents_m=16)
    m = Model(**kw, precision=precision)
    sd = make_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=81)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    b, n = 2, 512
    x = make_input("x", (b, n, 256), seed=82)
    t = make_input("times", (b,), seed=82, uniform=True)
    prompt = make_input("prompt", (b, 45, 64), seed=83)
    cond = make_input("cond", (b, 64, n), seed=84)
    with torch.no_grad():
        y = m.forward_with_cond_scale(x.to(DEV), t.to(DEV), prompt=prompt.to(DEV), cond=cond.to(DEV), cond_scale=1.4)
        if dim_head not in _HD_REF:                                   # (one CPU-oracle run per head dim serves the three plans)
            _HD_REF[dim_head] = O.model_forward_with_cond_scale(sd, x, t, prompt, cond, 1.4, dim_head=dim_head)
        ref = _HD_REF[dim_head]
    e = _rel(y.cpu(), ref)
    assert torch.isfinite(y).all() and e < tol, (dim_head, precision, e)


def test_long_utterances_4096_frames():
    """the reference takes any length (NS2:929-1000); everything benched is 1024 frames.  4096 frames (16 row tiles per utterance for the lean
    kernels, 64 key tiles in attention, dilation 128 reaching across row tiles): exact and hybrid against the oracle's code on GPU tensors
    (plain PyTorch fp32 ops; the exact plan agreeing with it to 1.2e-5 is the cross-check of both)"""
    from oracle import ns2_oracle as O
    kw = dict(dim=512, depth=2)
    m = Model(**kw, precision="exact")
    sd = make_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=5)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    x = make_input("x", (2, 4096, 512), seed=6).to(DEV)
    t = make_input("times", (2,), seed=6, uniform=True).to(DEV)
    with torch.no_grad():
        ref = O.model_forward({k: v.to(DEV) for k, v in sd.items()}, x, t)
        errs = {}
        for precision, tol in (("exact", 3e-5), ("hybrid", 2.5e-4)):
            m.precision = precision
            y = m(x, t)
            errs[precision] = _rel(y, ref)
            assert torch.isfinite(y).all() and errs[precision] < tol, errs


def test_config2_d128_at_its_stated_size_32x1024():
    """BASELINE config 2 as stated: Model(dim=128, depth=6) unconditional, batch 32 x 1024 latent tokens"""
    from oracle import ns2_oracle as O
    from tests.parity_record import record
    kw = dict(dim=128, depth=6)
    b, n = 32, 1024
    x = make_input("x", (b, n, 128), seed=41)
    t = make_input("times", (b,), seed=41, uniform=True)
    errs, per_utt = {}, {}
    ref = None
    for precision, tol in (("exact", 1e-4), ("hybrid", 2.5e-4)):
        m = Model(**kw, precision=precision)
        sd = make_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=40)
        m.load_state_dict(sd)
        m = m.to(DEV).eval()
        if ref is None:
            sdg = {k: v.to(DEV) for k, v in sd.items()}
            with torch.no_grad():
                ref = _gpu_oracle_pinned(sd, lambda: O.model_forward(sdg, x.to(DEV), t.to(DEV)), lambda: O.model_forward(sd, x[:1], t[:1]))
        with torch.no_grad():
            y = m(x.to(DEV), t.to(DEV))
        errs[precision] = _rel(y, ref)
        d = (y.double() - ref.double()).flatten(1).norm(dim=1) / ref.double().flatten(1).norm(dim=1)
        per_utt[precision] = d.max().item()
        assert torch.isfinite(y).all() and per_utt[precision] < tol, (precision, errs[precision], per_utt[precision])
        del m
    record("config2_stated_size_32x1024", dict(batch=errs, worst_utterance=per_utt))


# ---------------------------------------------------------------------------------------------- mixed training, external conditioning inputs
def test_mixed_training_unscales_the_gradients_of_cond_and_prompt_and_reports_overflow():
    """train_precision="mixed" runs its backward on loss-scaled gradients (training._Scale).  `prompt` and `cond` come from trainable
    modules upstream (SpeechPromptEncoder, the phoneme / pitch encoders: NS2:1635), so their gradients must leave the scaled domain
    like x's -- round 5 returned them 2^27 times too large.  Conditioned d128 / L2 on the MI355X: every gradient incl. d/dprompt and
    d/dcond of the mixed arithmetic against the exact one; `_Scale.overflowed()` is False on a sane pass and True once a conversion
    left the IEEE-half range (activations x 1e6)."""
    kw = dict(dim=128, depth=2, dim_prompt=96, condition_on_prompt=True, cond_drop_prob=0.)
    m = Model(**kw, precision="hybrid")
    m.load_state_dict(make_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=61))
    m = m.to(DEV).train()
    b, n = 2, 256
    x = make_input("x", (b, n, 128), seed=62).to(DEV)
    t = make_input("times", (b,), seed=62, uniform=True).to(DEV)
    prompt = make_input("prompt", (b, 37, 96), seed=63).to(DEV)
    cond = make_input("cond", (b, 96, n), seed=64).to(DEV)

    def grads(tp, xs=1.0):
        m.train_precision = tp
        for p in m.parameters():
            p.grad = None
        xx, pp, cc = (v.clone().requires_grad_(True) for v in (x * xs, prompt, cond))
        y = m(xx, t, prompt=pp, cond=cc)
        (y * make_input("gw", tuple(y.shape), seed=65).to(DEV) * 1e-7).sum().backward()
        torch.cuda.synchronize()
        return dict(x=xx.grad.clone(), prompt=pp.grad.clone(), cond=cc.grad.clone())

    try:
        g0 = grads("exact")
        g1 = grads("mixed")
        sc = m._last_loss_scale
        assert sc is not None and float(sc.s) >= 2.0 ** 16 and not sc.overflowed()
        for k in g0:
            assert torch.isfinite(g1[k]).all() and _rel(g1[k], g0[k]) < 1e-3, (k, _rel(g1[k], g0[k]))
        grads("mixed", xs=1e6)                            # activations far outside the half range: the forward's conversions clamp
        assert m._last_loss_scale.overflowed()
    finally:
        m.train_precision = "exact"
        ops.saturation_count(reset=True)


# ---------------------------------------------------------------------------------------------- the training pass as one HIP graph
@pytest.mark.parametrize("tprec", ["exact", "mixed"])
@pytest.mark.parametrize("conditioned", [False, True], ids=["uncond", "cond"])
def test_graphed_training_step(tprec, conditioned):
    """training.GraphedTrainStep: loss + backward of a fixed shape captured once, replayed per batch (NS2:1635 + NS2:1886 off the Python
    launch path).  The replay must be the eager pass bit for bit -- loss and every parameter's .grad -- on the captured batch AND on a new
    batch copied into the static inputs, after an in-place optimizer step in between (the one repack launch inside the graph reads the
    parameters' own storage), with conditioning dropout drawn inside the graph."""
    from naturalspeech2_pytorch_amd import NaturalSpeech2, training
    kw = dict(dim=128, depth=2)
    if conditioned:
        kw.update(dim_prompt=96, condition_on_prompt=True, cond_drop_prob=0.)
    m = Model(**kw)
    m.load_state_dict(make_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=71))
    m = m.to(DEV).train()
    m.train_backend, m.train_precision = "hip", tprec
    d = NaturalSpeech2(m, codec=None, target_sample_hz=24000).to(DEV)
    b, n = 2, 256
    mk = lambda name, shape, seed, **k: make_input(name, shape, seed=seed, **k).to(DEV)    # noqa: E731
    batch = lambda s: (mk("audio", (b, n, 128), s), mk("times", (b,), s, uniform=True), mk("noise", (b, n, 128), s + 1)) + (   # noqa: E731
        (mk("prompt", (b, 37, 96), s), mk("cond", (b, 96, n), s)) if conditioned else ())

    def loss_fn(a, t, z, *ex):
        if ex:
            return (m(a, t, prompt=ex[0], cond=ex[1]) - z).square().mean()
        return d(a, times=t, noise=z)

    def eager(ins):
        for p in m.parameters():
            p.grad = None
        loss = loss_fn(*ins)
        loss.backward()
        return loss.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}

    opt = torch.optim.SGD(m.parameters(), lr=1e-3)
    try:
        b0, b1 = batch(72), batch(74)
        keep = loss_fn(*b0)                                          # an eager pass whose autograd graph stays referenced (a training loop's
        keep.backward(retain_graph=True)                             # `loss` variable): its AccumulateGrad nodes live on the default stream
        step = training.GraphedTrainStep(loss_fn, b0, m)
        for ins in (b0, b1):
            l_e, g_e = eager(ins)
            for p in m.parameters():
                p.grad = None                                        # (eager grads gone: the replay must write its own static ones)
            l_g = step(*ins).detach().clone()
            assert not step.overflowed()
            got = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
            assert torch.equal(l_g, l_e) and got.keys() == g_e.keys() and len(got) > 20
            bad = [k for k in got if not torch.equal(got[k], g_e[k])]
            assert not bad, bad[:5]
            opt.step()                                               # in place: the next replay / eager pass multiplies the NEW weights
        assert not torch.equal(eager(b0)[0], l_e)
    finally:
        m.train_precision = "exact"
        ops.saturation_count(reset=True)


def test_retile_keeps_the_lean_kernels_images_current_under_the_one_launch_refresh():
    """`ns2_weights_retile` (include/ns2hip.h): packs that carry the lean mixed linear kernel's tile images can be part of the training path's
    one-launch refresh.  Off by default (`_PackedCache.LEAN`: measured slower, profiles/r06_training_graph_ab.txt); here it is switched on:
    mixed-arithmetic losses and gradients over three optimizer steps must equal the default path's bit for bit -- the lean kernel is
    bit-identical to gemm2_kernel<2, *>, so any difference is a stale image."""
    from naturalspeech2_pytorch_amd import training

    def run(lean):
        training._PackedCache.LEAN = lean
        training._HIP.clear()                                        # fresh backends: the flag is read when a cache is created
        m = Model(dim=256, depth=1)
        m.load_state_dict(make_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=97))
        m = m.to(DEV).train()
        m.train_backend, m.train_precision = "hip", "mixed"
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, fused=True)   # fused: no version bumps -- only the per-pass refresh sees the steps
        x = make_input("x", (2, 512, 256), seed=98).to(DEV)
        t = make_input("times", (2,), seed=98, uniform=True).to(DEV)
        out = []
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            loss = m(x, t).square().mean()
            loss.backward()
            out.append((loss.detach().clone(), m.wavenet.init_conv.weight.grad.clone(), getattr(m.transformer.to_pred, "1").weight.grad.clone()))
            opt.step()
        tiled = sum(1 for bk in training._HIP.values() for v in bk.packs.map.values() if getattr(v[0], "tiled", False))
        return out, tiled

    try:
        ref, n0 = run(False)
        got, n1 = run(True)
        assert n0 == 0 and n1 > 4, (n0, n1)
        for (l0, a0, b0), (l1, a1, b1) in zip(ref, got):
            assert torch.equal(l0, l1) and torch.equal(a0, a1) and torch.equal(b0, b1)
        assert not torch.equal(ref[0][0], ref[2][0])                  # the steps did change the weights
    finally:
        training._PackedCache.LEAN = False
        training._HIP.clear()
        ops.saturation_count(reset=True)


def test_weights_unchanged_skips_the_refresh_of_the_packs():
    """gradient accumulation runs several passes per optimizer step; `training.weights_unchanged()` skips the per-pass refresh
    of the packed weights.  Same loss and gradients as an ordinary pass; a weight changed INSIDE the block is (by contract) not seen until the
    first pass outside it (changes that bump the version counter are seen at once, as always)."""
    from naturalspeech2_pytorch_amd import training
    m = Model(dim=128, depth=1)
    m.load_state_dict(make_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=95))
    m = m.to(DEV).train()
    m.train_backend = "hip"
    x = make_input("x", (2, 256, 128), seed=96).to(DEV)
    t = make_input("times", (2,), seed=96, uniform=True).to(DEV)

    def run():
        for p in m.parameters():
            p.grad = None
        y = m(x, t)
        y.square().mean().backward()
        return y.detach().clone(), m.wavenet.init_conv.weight.grad.clone()

    y0, g0 = run()
    with training.weights_unchanged():
        y1, g1 = run()
        assert torch.equal(y1, y0) and torch.equal(g1, g0)
        with torch.no_grad():
            getattr(m.transformer.to_pred, "1").weight.data.mul_(2.0)     # the last Linear (NS2:783): the output doubles.  Through `.data`: no
            #                                                                version bump, like a fused optimizer (an ordinary in-place op IS seen)
        y2, _ = run()
        assert torch.equal(y2, y0)                                   # stale by contract
    y3, _ = run()
    assert _rel(y3, 2.0 * y0) < 1e-5                                 # outside: the packs follow the weights again


# ---------------------------------------------------------------------------------------------- the lean Wavenet block kernel (wavenet3_kernel.h)
@pytest.mark.parametrize("B,N,d,dil", [(4, 1024, 512, 1), (4, 1024, 512, 16), (2, 1024, 512, 128), (3, 256, 256, 2), (2, 512, 256, 64),
                                       (1, 768, 512, 32)])
def test_wavenet3_block_equals_gemm2_bit_for_bit_and_the_fp64_block(B, N, d, dil):
    """WavenetResBlock of the hybrid plan (NS2:597-642): dilated conv as one half product, res conv mixed.  Shapes: first row tiles of
    utterances with every dilation's rows in front of them (dilation 128: the whole first tile of tap 0), utterances of one tile."""
    g = torch.Generator().manual_seed(77 + d + dil)
    M = B * N
    x = torch.randn(M, d, generator=g)
    wc = torch.randn(d, d, 3, generator=g) * (3 * d) ** -0.5
    wr = torch.randn(d, d, 1, generator=g) * d ** -0.5
    bc, br = torch.randn(d, generator=g), torch.randn(d, generator=g)
    film = torch.randn(B, 2 * d, generator=g)
    a = ops.split(x.to(DEV), precision=4)
    pw = ops.PackedWeight(wc.to(DEV), extra1x1=wr.to(DEV), precision=4).tile_wavenet()
    outs = {}
    try:
        for k in (5, 2):
            _force(k)
            outs[k] = ops.wavenet_block(pw, a, N, dil, bc.to(DEV), br.to(DEV), film.to(DEV), precision=5).buf.clone()
    finally:
        _force(0)
    assert torch.equal(outs[5], outs[2]), "lean Wavenet block kernel differs from gemm2_kernel<2, EPI_WAVENET, true, 1>"
    # fp64 restatement on the operands as the kernels see them (x and the conv weights rounded to half for the dilated conv)
    xd = x.double().reshape(B, N, d).transpose(1, 2)
    xh = x.half().double().reshape(B, N, d).transpose(1, 2)
    h = torch.nn.functional.conv1d(torch.nn.functional.pad(xh, (2 * dil, 0)), wc.half().double(), bc.double(), dilation=dil)
    gam, bet = film.double()[:, :d, None], film.double()[:, d:, None]
    h = h * gam + bet
    h = torch.tanh(h) * torch.sigmoid(h)
    ref = (h + torch.nn.functional.conv1d(xd, wr.double(), br.double())).transpose(1, 2).reshape(M, d)
    got = ops.join(ops.Planes(outs[5], M, d, True, "h8"), d).cpu().double()
    assert ((got - ref).norm() / ref.norm()).item() < 3e-4


# ---------------------------------------------------------------------------------------------- classifier-free guidance as one 2B batch
def test_cfg_as_one_batch_equals_two_passes_and_the_oracle():
    """NS2:914-927: null + (cond - null) * scale.  For small batches the conditioned and the null forward run as ONE batch of 2 B utterances
    (Model._forward_hip_cfg: the two prepared conditioning states laid end to end, ns2_model_cond_stack): same result as two passes to fp32
    rounding (another K split may apply), same distance from the oracle; also through a time-table row and along a sampled trajectory."""
    from oracle import ns2_oracle as O
    from naturalspeech2_pytorch_amd import NaturalSpeech2
    kw = dict(dim=128, depth=2, dim_prompt=96, condition_on_prompt=True)
    m = Model(**kw, precision="exact")
    sd = make_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=71)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    b, n = 3, 256
    x = make_input("x", (b, n, 128), seed=72).to(DEV)
    t = make_input("times", (b,), seed=72, uniform=True).to(DEV)
    prompt = make_input("prompt", (b, 37, 96), seed=73).to(DEV)
    cond = make_input("cond", (b, 96, n), seed=74).to(DEV)
    with torch.no_grad():
        assert b <= m.CFG_ONE_BATCH_MAX
        y1 = m.forward_with_cond_scale(x, t, prompt=prompt, cond=cond, cond_scale=1.3)
        y1b = m.forward_with_cond_scale(x, t, prompt=prompt, cond=cond, cond_scale=1.3)
        old, m.CFG_ONE_BATCH_MAX = m.CFG_ONE_BATCH_MAX, 0
        try:
            y2 = m.forward_with_cond_scale(x, t, prompt=prompt, cond=cond, cond_scale=1.3)
        finally:
            m.CFG_ONE_BATCH_MAX = old
        ref = O.model_forward_with_cond_scale(sd, x.cpu(), t.cpu(), prompt.cpu(), cond.cpu(), 1.3)
    assert torch.equal(y1, y1b)
    assert _rel(y1, y2) < 2e-6, _rel(y1, y2)
    assert _rel(y1.cpu(), ref) < 1e-4 and abs(_rel(y1.cpu(), ref) - _rel(y2.cpu(), ref)) < 2e-6
    # the sampler (time table rows + CFG) gives the same trajectory either way
    d = NaturalSpeech2(m, codec=None, target_sample_hz=24000, timesteps=6).to(DEV)
    noise = make_input("noise", (b, n, 128), seed=75).to(DEV)
    o1 = d.sample(length=n, noise=noise, prompt_enc=prompt, cond=cond, cond_scale=1.5)
    old, m.CFG_ONE_BATCH_MAX = m.CFG_ONE_BATCH_MAX, 0
    try:
        o2 = d.sample(length=n, noise=noise, prompt_enc=prompt, cond=cond, cond_scale=1.5)
    finally:
        m.CFG_ONE_BATCH_MAX = old
    assert _rel(o1, o2) < 1e-6, _rel(o1, o2)
