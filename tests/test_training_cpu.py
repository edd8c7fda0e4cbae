"""Host logic of the HIP training path (`naturalspeech2_pytorch_amd/training.py`) on CPU: the autograd Functions run on
`tests/emu_backend.EmuBackend` -- a plain-torch restatement of every backend call with the kernels' layout contracts -- and every
parameter gradient is compared with torch autograd through the PyTorch composite (`autograd_path.model_forward_autograd`, itself
pinned to the reference's goldens in test_host_cpu.py).  What this pins: the chain rule inside the Functions, the transposed /
flipped weight of the dgrad GEMMs (pad_left = 0), the per-tap shifts of the wgrad operands, the stacking offsets, which saved
tensor feeds which kernel.  The kernels themselves are checked on the MI355X (tests/test_backward_gpu.py)."""
import pytest
import torch

from naturalspeech2_pytorch_amd import Model, training
from naturalspeech2_pytorch_amd.autograd_path import model_forward_autograd
from tests.emu_backend import EmuBackend
from tests.golden.gen import make_input, make_weights


@pytest.fixture()
def emu():
    prev = training.set_backend(EmuBackend())
    yield
    training.set_backend(prev)


def _grads(m, fwd, x, t, **kw):
    for p in m.parameters():
        p.grad = None
    x = x.clone().requires_grad_(True)
    y = fwd(m, x, t, **kw)
    w = make_input("gw", tuple(y.shape), seed=11)
    (y * w).sum().backward()
    return y.detach(), x.grad.clone(), {k: (p.grad.clone() if p.grad is not None else None) for k, p in m.named_parameters()}


def _rel(a, b):
    return ((a - b).norm() / b.norm().clamp(min=1e-30)).item()


@pytest.mark.parametrize("cond", [False, True, 96], ids=["uncond", "cond", "cond_proj_context"])
def test_training_functions_match_autograd_composite(emu, cond):
    kw = dict(dim=64, depth=2, wavenet_layers=3, wavenet_stacks=2)
    dpr = 64 if cond is True else cond             # dim_prompt != dim: the resampler's proj_context Linear (NS2:548) through GemmFn
    if cond:
        kw.update(dim_prompt=dpr, condition_on_prompt=True, num_latents_m=8)
    m = Model(**kw)
    sd = make_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=5)
    m.load_state_dict(sd)
    b, n = 2, 40                                     # ragged against every tile size; dilation 4 reaches across 8 of 40 frames
    x = make_input("x", (b, n, 64), seed=6)
    t = make_input("times", (b,), seed=6, uniform=True)
    extra = {}
    if cond:
        extra = dict(prompt=make_input("prompt", (b, 11, dpr), seed=7), cond=make_input("cond", (b, dpr, 33), seed=7), cond_drop_prob=0.)
    y0, dx0, g0 = _grads(m, model_forward_autograd, x, t, **extra)
    y1, dx1, g1 = _grads(m, training.model_forward_train, x, t, **extra)
    assert _rel(y1, y0) < 1e-5
    assert _rel(dx1, dx0) < 1e-4
    worst = ("", 0.0)
    for k in g0:
        if g0[k] is None:
            assert g1[k] is None or g1[k].abs().max() == 0, k
            continue
        assert g1[k] is not None, f"no gradient for {k}"
        assert g1[k].shape == g0[k].shape, k
        e = _rel(g1[k], g0[k])
        worst = max(worst, (k, e), key=lambda z: z[1])
        assert e < 2e-4, (k, e)
    print("worst parameter gradient:", worst)


def test_packed_cache_refreshes_in_place_and_never_serves_a_dead_parameter():
    """the training packs are keyed by id(parameter): a recycled id must not return another tensor's pack"""
    from naturalspeech2_pytorch_amd.training import _PackedCache
    made = []

    class FakePW:
        def __init__(self, w, extra1x1=None, precision=3):
            self.handle = len(made)
            made.append(tuple(w.shape))

    import naturalspeech2_pytorch_amd.training as T
    orig = T.ops.PackedWeight
    T.ops.PackedWeight = FakePW
    try:
        c = _PackedCache()
        w = torch.nn.Parameter(torch.randn(4, 4))
        a = c.get(("f", id(w)), (w,), lambda: w)
        assert c.get(("f", id(w)), (w,), lambda: w) is a and len(made) == 1
        w2 = torch.nn.Parameter(torch.randn(4, 4))
        b = c.get(("f", id(w)), (w2,), lambda: w2)          # same key, another live object: a miss, never `a`
        assert b is not a and len(made) == 2
    finally:
        T.ops.PackedWeight = orig


def test_cond_projections_fn_unused_outputs_and_views(emu):
    """CondProjectionsFn alone against torch autograd: several Linears of different widths on the same rows, one output unused (its
    gradient arrives as None and must count as zeros), parameter gradients returned as row blocks of ONE product."""
    from naturalspeech2_pytorch_amd.training import CondProjectionsFn
    torch.manual_seed(3)
    B, K = 5, 24
    sizes = (16, 8, 40)
    t = torch.randn(B, K, requires_grad=True)
    ws = [torch.randn(n, K, requires_grad=True) for n in sizes]
    bs = [torch.randn(n, requires_grad=True) for n in sizes]
    outs = CondProjectionsFn.apply(t, *[q for w, b in zip(ws, bs) for q in (w, b)])
    assert [tuple(o.shape) for o in outs] == [(B, n) for n in sizes] and all(o.is_contiguous() for o in outs)
    gw = [torch.randn(B, n) for n in sizes]
    (outs[0] * gw[0]).sum().add((outs[2] * gw[2]).sum()).backward()          # outs[1] is never used
    got = [t.grad.clone()] + [w.grad.clone() for w in ws] + [b.grad.clone() for b in bs]
    t.grad = None
    for q in ws + bs:
        q.grad = None
    ref_outs = [t @ w.t() + b for w, b in zip(ws, bs)]
    for o, r in zip(outs, ref_outs):
        assert torch.allclose(o, r, atol=1e-5)
    (ref_outs[0] * gw[0]).sum().add((ref_outs[2] * gw[2]).sum()).backward()
    want = [t.grad] + [w.grad if w.grad is not None else torch.zeros_like(w) for w in ws] + \
           [b.grad if b.grad is not None else torch.zeros_like(b) for b in bs]
    for g, r in zip(got, want):
        assert torch.allclose(g, r, atol=1e-4), (g - r).abs().max()


def _golden_grad_case(name):
    import os
    fix = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"grads_{name}.pt"), weights_only=False)
    kw = fix["kwargs"]
    m = Model(**kw)
    m.load_state_dict(make_weights(fix["shapes"], seed=fix["weight_seed"]))
    b, n, d = fix["batch"], fix["n"], kw["dim"]
    x = make_input("x", (b, n, d), seed=fix["input_seed"])
    t = make_input("times", (b,), seed=fix["input_seed"], uniform=True)
    extra = {}
    if kw.get("condition_on_prompt"):
        extra = dict(prompt=make_input("prompt", (b, fix["n_prompt"], kw["dim_prompt"]), seed=fix["input_seed"]),
                     cond=make_input("cond", (b, kw["dim_prompt"], fix["n_cond"]), seed=fix["input_seed"]), cond_drop_prob=0.)
    return fix, m, x, t, extra


def _check_against_golden_grads(fix, m, fwd, x, t, extra, tol):
    for p in m.parameters():
        p.grad = None
    x = x.clone().requires_grad_(True)
    y = fwd(m, x, t, **extra)
    (y * make_input("gw", tuple(y.shape), seed=fix["loss_weight_seed"]).to(y.device)).sum().backward()
    assert _rel(y.detach().cpu(), fix["output"]) < tol
    assert _rel(x.grad.cpu(), fix["x_grad"]) < tol
    worst = ("", 0.0)
    for k, p in m.named_parameters():
        g, r = p.grad, fix["grads"][k]
        assert g is not None and g.shape == r.shape, k
        if r.abs().max() == 0:
            assert g.abs().max() == 0, k
            continue
        worst = max(worst, (k, _rel(g.cpu(), r)), key=lambda z: z[1])
    assert worst[1] < tol, worst
    return worst


@pytest.mark.parametrize("name", ["uncond_d64", "cond_d64"])
def test_gradients_match_the_reference_autograd_golden(emu, name):
    """tests/golden/grads_*.pt: EVERY parameter's gradient, dL/dx and the prediction from the unmodified reference's own autograd
    (make_golden.py gen_grad_case).  Against it on CPU: the PyTorch composite (autograd_path.py) and the HIP training graph's host
    logic on the emulated backend; the kernels meet the same fixtures in tests/test_backward_gpu.py."""
    fix, m, x, t, extra = _golden_grad_case(name)
    w1 = _check_against_golden_grads(fix, m, model_forward_autograd, x, t, extra, 2e-5)
    w2 = _check_against_golden_grads(fix, m, training.model_forward_train, x, t, extra, 2e-4)
    print("worst vs the reference's autograd:", w1, w2)


def test_frozen_weights_skip_the_weight_gradients(emu):
    """with every parameter frozen and only x.requires_grad (guidance, gradient-based analysis) the backward must not run
    a single wgrad GEMM or operand transpose -- and dL/dx must still be the composite's"""
    kw = dict(dim=64, depth=1, wavenet_layers=2, wavenet_stacks=2)
    m = Model(**kw)
    m.load_state_dict(make_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=5))
    for p in m.parameters():
        p.requires_grad_(False)
    x = make_input("x", (2, 40, 64), seed=6)
    t = make_input("times", (2,), seed=6, uniform=True)
    bk = training.backend()
    calls = {"wgrad": 0, "transpose": 0}
    w0, t0 = bk.wgrad, bk.transpose

    def wgrad(*a, **k):
        calls["wgrad"] += 1
        return w0(*a, **k)

    def transpose(p, col0, C, seq_len, shifts=(0,), per_batch=False, **k):
        calls["transpose"] += 0 if per_batch else 1          # per_batch transposes feed the attention backward, not a wgrad
        return t0(p, col0, C, seq_len, shifts, per_batch=per_batch, **k)

    bk.wgrad, bk.transpose = wgrad, transpose
    try:
        _, dx0, _ = _grads(m, model_forward_autograd, x, t)
        calls["wgrad"] = calls["transpose"] = 0
        _, dx1, g1 = _grads(m, training.model_forward_train, x, t)
    finally:
        bk.wgrad, bk.transpose = w0, t0
    assert calls == {"wgrad": 0, "transpose": 0}, calls
    assert all(g is None for g in g1.values())
    assert _rel(dx1, dx0) < 1e-4


def test_partially_frozen_model_trains_the_rest(emu):
    """only the Wavenet's res convs and the FF-out weights trainable: their gradients match the composite's, the frozen ones get none"""
    m = Model(dim=64, depth=1, wavenet_layers=2, wavenet_stacks=2)
    m.load_state_dict(make_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=5))
    keep = lambda k: "res_conv" in k or k.endswith("5.3.weight")          # noqa: E731
    for k, p in m.named_parameters():
        p.requires_grad_(keep(k))
    x = make_input("x", (2, 40, 64), seed=6)
    t = make_input("times", (2,), seed=6, uniform=True)
    _, dx0, g0 = _grads(m, model_forward_autograd, x, t)
    _, dx1, g1 = _grads(m, training.model_forward_train, x, t)
    assert _rel(dx1, dx0) < 1e-4
    for k in g0:
        if keep(k):
            assert g1[k] is not None and _rel(g1[k], g0[k]) < 1e-4, k
        else:
            assert g1[k] is None, k


def test_training_path_refuses_shapes_the_kernels_are_not_written_for():
    """dim_head != 64 must never reach the HIP training kernels (q / k / v offsets assume heads of 64)"""
    assert training.unsupported_reason(Model(dim=64, depth=1)) is None
    why = training.unsupported_reason(Model(dim=64, depth=1, dim_head=32))
    assert why is not None and "dim_head" in why
    assert "fp32" in training.unsupported_reason(Model(dim=64, depth=1).half())


@pytest.mark.parametrize("cond", [False, True], ids=["uncond", "cond"])
def test_mixed_training_arithmetic_needs_and_gets_its_loss_scale(cond):
    """train_precision="mixed": the GEMMs of forward, dgrad and wgrad multiply FMT_H8 operands (IEEE half + e5m2 correction terms).
    With the gradient of a mean-reduced loss (~1e-7 per element) the unscaled pass loses everything; under training._Scale -- a power
    of two chosen from the incoming gradient, token-sized gradients kept scaled between the Functions, parameter / conditioning /
    input gradients unscaled where they leave -- every gradient matches fp32 autograd like the exact arithmetic does.  Host logic
    on the emulated backend (tests/emu_backend.MixedEmuBackend); the kernels: tests/test_backward_gpu.py."""
    from tests.emu_backend import MixedEmuBackend
    kw = dict(dim=64, depth=2, wavenet_layers=3, wavenet_stacks=2)
    if cond:
        kw.update(dim_prompt=96, condition_on_prompt=True, num_latents_m=8)
    m = Model(**kw)
    m.load_state_dict(make_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=5))
    b, n = 2, 40
    x = make_input("x", (b, n, 64), seed=6)
    t = make_input("times", (b,), seed=6, uniform=True)
    extra = {}
    if cond:
        extra = dict(prompt=make_input("prompt", (b, 11, 96), seed=7), cond=make_input("cond", (b, 96, 33), seed=7), cond_drop_prob=0.)

    def grads(fwd):
        for p in m.parameters():
            p.grad = None
        xx = x.clone().requires_grad_(True)
        # the external conditioning inputs come from trainable modules upstream (prompt / phoneme encoders, NS2:1635): their gradients
        # must leave the scaled domain too (they came out 2^27 times too large)
        ex = {k: (v.clone().requires_grad_(True) if torch.is_tensor(v) else v) for k, v in extra.items()}
        y = fwd(m, xx, t, **ex)
        (y * make_input("gw", tuple(y.shape), seed=11) * 1e-7).sum().backward()          # the magnitude of dL/dy of a mean-reduced loss
        g = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
        g.update({f"<input {k}>": v.grad.clone() for k, v in ex.items() if torch.is_tensor(v)})
        return xx.grad.clone(), g

    dx0, g0 = grads(model_forward_autograd)
    prev = training.set_backend(MixedEmuBackend())
    try:
        m.train_precision = "mixed"
        dx1, g1 = grads(training.model_forward_train)
        s = float(m._last_loss_scale.s)
        assert s == 2.0 ** round(__import__("math").log2(s)) and 2.0 ** 20 <= s <= 2.0 ** 32, s      # a power of two that lifts ~1e-7 to ~2^5
        m.train_precision = "exact"                                                      # same rounding, NO scale: the point of it
        dx2, g2 = grads(training.model_forward_train)
    finally:
        training.set_backend(prev)
        m.train_precision = "exact"
    assert set(g1) == set(g0)
    worst = max(_rel(g1[k], g0[k]) for k in g0)
    assert _rel(dx1, dx0) < 3e-4 and worst < 3e-4, (_rel(dx1, dx0), worst)
    assert max(_rel(g2[k], g0[k]) for k in g0) > 1e-2                                    # unscaled: garbage


def test_repack_parts_describe_the_sources_the_packs_were_made_from():
    """`_PackedCache._part` tells ns2_weights_repack where a rectangle of a packed weight lives in the PARAMETER's own storage (strides in
    elements, possibly negative).  Walking the parameter through those strides on the CPU must rebuild exactly the matrix `make_src`
    hands to the first pack -- for the forward pack, the dgrad pack of a Linear (W^T), of a causal conv (W^T with flipped taps) and
    the q | kv concatenations (two parts, by rows / by columns)."""
    from naturalspeech2_pytorch_amd.training import _PackedCache

    class PW:
        handle = 0

    def rebuild(shape, params, parts):
        out = torch.full(shape, float("nan"))
        for (idx, mode, row0, col0) in parts:
            p = params[idx]
            desc = _PackedCache._part(lambda h, base, sr, sc, st, r0, rows, c0, cols: (base, sr, sc, st, r0, rows, c0, cols), PW(), p, mode, row0, col0)
            base, sr, sc, st, r0, rows, c0, cols = desc
            flat = p.detach().reshape(-1)
            off0 = (base - p.data_ptr()) // 4
            T = shape[2] if len(shape) == 3 else 1
            for r in range(rows):
                for c in range(cols):
                    for t in range(T):
                        v = flat[off0 + r * sr + c * sc + t * st]
                        if len(shape) == 3:
                            out[r0 + r, c0 + c, t] = v
                        else:
                            out[r0 + r, c0 + c] = v
        return out

    g = torch.Generator().manual_seed(0)
    w_lin = torch.nn.Parameter(torch.randn(6, 10, generator=g))
    w_conv = torch.nn.Parameter(torch.randn(5, 7, 3, generator=g))
    wq, wkv = torch.nn.Parameter(torch.randn(8, 6, generator=g)), torch.nn.Parameter(torch.randn(16, 6, generator=g))
    cases = [
        ((w_lin,), [(0, "n", 0, 0)], w_lin.detach()),
        ((w_conv,), [(0, "n", 0, 0)], w_conv.detach()),
        ((w_lin,), [(0, "t", 0, 0)], w_lin.detach().t()),
        ((w_conv,), [(0, "tf", 0, 0)], w_conv.detach().permute(1, 0, 2).flip(-1)),
        ((wq, wkv), [(0, "n", 0, 0), (1, "n", 8, 0)], torch.cat((wq.detach(), wkv.detach()), 0)),
        ((wq, wkv), [(0, "t", 0, 0), (1, "t", 0, 8)], torch.cat((wq.detach(), wkv.detach()), 0).t()),
    ]
    for params, parts, want in cases:
        got = rebuild(tuple(want.shape), params, parts)
        assert torch.equal(got, want.contiguous()), (parts, got, want)
