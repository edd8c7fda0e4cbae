"""GPU parity tests of each HIP kernel family against plain torch fp32/fp64 restatements of the reference ops
(called through the C ABI via naturalspeech2_pytorch_amd.ops).  Per-op tolerances (relative L2) against the fp64
result on the operand values the kernel sees: precision 3 ("exact", bf16x3 split) fp32-class 2e-5; precision 4 ("mixed":
IEEE-half product + both correction terms on the fp8 MFMA) 8e-5; precision 2 ("half", one IEEE-half product) 8e-4 (only
the weight rounding shows, the activations are compared as rounded); precision 1 ("fast", single bf16) bf16-class.
Every GEMM-family and attention test runs at all four precisions and on both GEMM kernels."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

from naturalspeech2_pytorch_amd import ops  # noqa: E402
from oracle import rvq_oracle as R  # noqa: E402
from tests.golden.gen import make_input  # noqa: E402

DEV = torch.device("cuda:0")
TOL = {3: 2e-5, 4: 8e-5, 2: 8e-4, 1: 2e-2}
PRECS = [3, 4, 2, 1]
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def exact(p):        # value the kernels see for a split operand
    return ops.join(p).double()


def asplit(x, prec, ldo=None):
    """attention operands (q, k, V^T): IEEE half also at precision 4 (include/ns2hip.h)"""
    return ops.split(x, ldo=ldo, precision=2 if prec == 4 else prec)


@pytest.fixture(params=[1, 2, 0], ids=["gemm128", "gemm256", "splitk"])
def gemm_kernel(request):
    """run every GEMM-family test on both kernels (gemm.hip 128x128 register-staged, gemm2.hip 256x256 LDS-DMA) and, third, with
    the automatic dispatch plus split-K scratch lent to this thread (ns2_debug_lend_splitk_scratch): these shapes are all "small
    products", so whatever has >= 8 K tiles per tap runs as K slices + the finishing launch (gemm.hip launch_gemm_splitk)."""
    from naturalspeech2_pytorch_amd import _lib
    lib = _lib.load()
    _lib.check(lib.ns2_debug_force_gemm(request.param))
    scratch = None
    if request.param == 0:
        n = int(lib.ns2_splitk_scratch_bytes())
        scratch = torch.full((n // 4,), float("nan"), device=DEV)                    # a slot read before it is written shows
        _lib.check(lib.ns2_debug_lend_splitk_scratch(scratch.data_ptr(), n))
    yield request.param
    torch.cuda.synchronize()
    lib.ns2_debug_lend_splitk_scratch(None, 0)
    lib.ns2_debug_force_gemm(0)
    del scratch


def test_split_join_roundtrip():
    x = rnd(300, 100, seed=1, scale=3.0)
    p = ops.split(x)
    assert (p.rows, p.ld) == (300, 128) and p.buf.shape == (300, 256)      # interleaved [hi32|lo32] rows
    y = ops.join(p, 100)
    assert rel(y, x) < 1e-5
    assert ops.join(p)[:, 100:].abs().sum().item() == 0.0       # zero padding
    hi_only = ops.join(p.hi_only(), 100)
    assert rel(hi_only, x) < 5e-3
    assert torch.equal(p.hi_plane()[:, :100], x.to(torch.bfloat16))      # hi plane == RNE bf16
    d = ops.split(x, lo=False)                                   # dense hi-only layout (precision 1 models)
    assert d.buf.shape == (300, 128) and torch.equal(d.buf[:, :100], x.to(torch.bfloat16))


def test_split_plane_layout_contract():
    """include/ns2hip.h: with a lo plane the row is [hi32|lo32] per 32 logical columns and lo must be hi + 32 elements."""
    x = rnd(70, 96, seed=2, scale=2.0)
    p = ops.split(x)
    raw = p.buf.view(torch.int16).reshape(70, 3, 2, 32)            # [row][32-block][hi|lo][32]
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    assert torch.equal(raw[:, :, 0, :].reshape(70, 96), hi.view(torch.int16))
    assert torch.equal(raw[:, :, 1, :].reshape(70, 96), lo.view(torch.int16))
    out = torch.empty(70, 96, device=DEV)
    from naturalspeech2_pytorch_amd import _lib
    lib = _lib.load()
    rc = lib.ns2_join_f32(p.hi, p.hi + 2 * 64, 96, out.data_ptr(), 96, 70, 96, 3, torch.cuda.current_stream().cuda_stream)
    assert rc != 0 and b"invalid" in lib.ns2_last_error().lower()  # lo pointer that is not hi + 32 elements is rejected


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("M,K,N", [(300, 96, 200), (1024, 512, 512), (128, 64, 64), (4096, 352, 128), (77, 1376, 512), (600, 1376, 300), (512, 96, 1365)])
def test_linear_f32(M, K, N, prec, gemm_kernel):
    x = rnd(M, K, seed=2)
    w = rnd(N, K, seed=3, scale=1 / math.sqrt(K))
    b = rnd(N, seed=4)
    r = rnd(M, N, seed=5)
    pw = ops.PackedWeight(w, precision=prec)
    a = ops.split(x, precision=prec)
    y = ops.linear_f32(pw, a, bias=b, resid=r, precision=prec)
    ref = exact(a)[:, :K] @ w.double().t() + b.double() + r.double()
    e = rel(y, ref)
    assert e < TOL[prec], f"rel err {e}"
    y2 = ops.linear_f32(pw, a, precision=prec)                    # no bias / resid
    assert rel(y2, exact(a)[:, :K] @ w.double().t()) < TOL[prec]


def conv_ref(x_bnc, w, b, dil):
    """reference CausalConv1d (NS2:583-595) on [B, N, C] fp64."""
    xt = x_bnc.transpose(1, 2)
    k = w.shape[-1]
    xt = F.pad(xt, (dil * (k - 1), 0))
    return F.conv1d(xt, w.double(), b.double() if b is not None else None, dilation=dil).transpose(1, 2)


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("B,N,Cin,Cout,dil", [(3, 200, 96, 80, 1), (2, 300, 100, 100, 4), (2, 1024, 64, 64, 128), (1, 50, 170, 170, 1),
                                              # utterances aligned to the 256-row tile, dilation 1: the tap-shared path of the 256x256
                                              # kernel (split epilogue), incl. an odd number of 32-blocks per tap and several column tiles
                                              (2, 512, 170, 200, 1), (3, 256, 96, 300, 1), (1, 768, 100, 100, 1),
                                              # >= 8 K tiles per tap: split-K candidates (uneven slices: 11 tiles as 6 + 5)
                                              (2, 384, 352, 341, 1), (1, 300, 512, 200, 2)])
def test_causal_conv(B, N, Cin, Cout, dil, prec, gemm_kernel):
    x = rnd(B * N, Cin, seed=6)
    w = rnd(Cout, Cin, 3, seed=7, scale=1 / math.sqrt(3 * Cin))
    b = rnd(Cout, seed=8)
    pw = ops.PackedWeight(w, precision=prec)
    a = ops.split(x, precision=prec)
    ref = conv_ref(exact(a)[:, :Cin].reshape(B, N, Cin), w, b, dil).reshape(B * N, Cout)
    y = ops.linear_f32(pw, a, bias=b, conv_taps=3, dilation=dil, seq_len=N, precision=prec)
    e = rel(y, ref)
    assert e < TOL[prec], f"rel err {e}"
    ys = ops.linear_split(pw, a, bias=b, conv_taps=3, dilation=dil, seq_len=N, precision=prec)
    assert rel(ops.join(ys, Cout), ref) < TOL[prec] + (5e-4 if prec == 2 else 1e-5)       # + the output's own rounding
    assert ops.join(ys)[:, Cout:].abs().sum().item() == 0.0


@pytest.mark.parametrize("B,N,Cin,Cout,k,pad", [(2, 100, 64, 96, 9, 4), (3, 37, 128, 300, 9, 4), (2, 60, 96, 64, 9, -1), (1, 5, 64, 64, 9, 4)])
def test_conv_k9_same_padding_silu(B, N, Cin, Cout, k, pad, gemm_kernel):
    """SpeechPromptEncoder convs (NS2:316: Conv1d(k=9, padding=4) + SiLU) and PhonemeEncoder's causal k=9 conv."""
    x = rnd(B * N, Cin, seed=90)
    w = rnd(Cout, Cin, k, seed=91, scale=1 / math.sqrt(k * Cin))
    b = rnd(Cout, seed=92)
    pw = ops.PackedWeight(w)
    a = ops.split(x)
    xe = exact(a)[:, :Cin].reshape(B, N, Cin).transpose(1, 2)
    if pad < 0:
        ref = F.conv1d(F.pad(xe, (k - 1, 0)), w.double(), b.double())
    else:
        ref = F.conv1d(xe, w.double(), b.double(), padding=pad)
    ref = F.silu(ref).transpose(1, 2).reshape(B * N, Cout)
    y = ops.linear_f32(pw, a, bias=b, conv_taps=k, seq_len=N, pad_left=pad, act=1)
    assert rel(y, ref) < 2e-5
    ys = ops.linear_split(pw, a, bias=b, conv_taps=k, seq_len=N, pad_left=pad, act=1)
    assert rel(ops.join(ys, Cout), ref) < 3e-5


@pytest.mark.parametrize("M,Cin,Cout,k", [(700, 32, 16, 3), (300, 64, 32, 3), (513, 256, 128, 1), (260, 512, 300, 3)])
def test_conv_elu_epilogue(M, Cin, Cout, k, gemm_kernel):
    """act = 2: ELU in the epilogue (EnCodec's residual block, HFENC:268-301: conv1 hands ELU(h) to conv2 as operand planes);
    narrow outputs (16 of 32 plane columns are K padding of the consumer and must be zeros)"""
    x = rnd(M, Cin, seed=190)
    w = rnd(Cout, Cin, k, seed=191, scale=3 / math.sqrt(k * Cin))
    b = rnd(Cout, seed=192)
    pw = ops.PackedWeight(w)
    a = ops.split(x)
    xe = exact(a)[:, :Cin].reshape(1, M, Cin).transpose(1, 2)
    ref = F.elu(F.conv1d(F.pad(xe, (k - 1, 0)), w.double(), b.double())).transpose(1, 2).reshape(M, Cout)
    kw = dict(conv_taps=k, seq_len=M) if k > 1 else {}
    y = ops.linear_f32(pw, a, bias=b, act=2, **kw)
    assert rel(y, ref) < 2e-5
    ys = ops.linear_split(pw, a, bias=b, act=2, **kw)
    assert rel(ops.join(ys, Cout), ref) < 3e-5
    assert ops.join(ys)[:, Cout:].abs().sum().item() == 0.0


@pytest.mark.parametrize("prec", [3, 4, 2])
@pytest.mark.parametrize("B,T,C,P,inp", [(2, 50, 16, 2, 0), (3, 33, 32, 2, 2), (1, 20, 70, 0, 1)])
def test_seanet_prep2_windows(B, T, C, P, inp, prec):
    """ns2_seanet_prep2: ELU(x) and x of one fp32 activation (garbage prefix rows skipped on the way in, mirrored prefix rows on
    the way out) into column windows of two plane buffers; columns outside the windows are left alone, padding columns are zeros"""
    from naturalspeech2_pytorch_amd.seanet import _prep2
    x = rnd(B * (inp + T), C, seed=300 + C, scale=2.0)
    cs = ops.round_up(C, 32)
    e = ops._out_planes(B * (P + T), cs, DEV, prec)
    both = ops._out_planes(B * (P + T), 32 + cs, DEV, prec, zero=True)
    _prep2(x, B, T, C, in_prefix=inp, prefix=P, elu_out=(e, 0, cs), raw_out=(both, 32, cs), precision=prec)
    xv = x.reshape(B, inp + T, C)[:, inp:]
    ref = torch.cat([xv[:, 1:P + 1].flip(1), xv], dim=1).reshape(B * (P + T), C)          # reflect: row -j = row j
    tol = {3: 2e-5, 4: 3e-4, 2: 1e-3}[prec]                      # the formats' own rounding: bf16 hi + lo, half + e5m2 residual, half
    je, jb = ops.join(e), ops.join(both)
    assert rel(je[:, :C], F.elu(ref)) < tol and rel(jb[:, 32:32 + C], ref) < tol
    assert je[:, C:].abs().sum().item() == 0.0 and jb[:, 32 + C:].abs().sum().item() == 0.0 and jb[:, :32].abs().sum().item() == 0.0


@pytest.mark.parametrize("B,T,ci,co,inp,elu", [(2, 300, 1, 32, 0, False), (3, 77, 1, 16, 0, True), (2, 300, 32, 1, 2, True),
                                                (1, 45, 64, 1, 0, False), (2, 7, 16, 1, 1, True)])
def test_seanet_conv_narrow(B, T, ci, co, inp, elu):
    """ns2_seanet_conv_narrow: the 1 -> co and ci -> 1 channel k = 7 causal convolutions with reflect padding (and ELU on the
    input) in fp32 against torch's conv1d on the reflect-padded signal; an unsupported shape reports NS2_UNAVAILABLE"""
    from naturalspeech2_pytorch_amd import _lib
    lib = _lib.load()
    k = 7
    x = rnd(B * (inp + T), ci, seed=400 + ci + co, scale=1.5)
    w = rnd(co, ci, k, seed=401, scale=1 / math.sqrt(k * ci))
    b = rnd(co, seed=402)
    y = torch.full((B * T, co), float("nan"), device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.ns2_seanet_conv_narrow(x.data_ptr(), ci, inp, B, T, ci, co, k, int(elu), w.data_ptr(), b.data_ptr(), y.data_ptr(), co, st)
    assert rc == 0, rc
    xv = x.reshape(B, inp + T, ci)[:, inp:].double()
    xv = F.elu(xv) if elu else xv
    ref = F.conv1d(F.pad(xv.transpose(1, 2), (k - 1, 0), mode="reflect"), w.double(), b.double()).transpose(1, 2).reshape(B * T, co)
    assert rel(y, ref) < 2e-6
    assert lib.ns2_seanet_conv_narrow(x.data_ptr(), ci, inp, B, T, ci, co, 5, 0, w.data_ptr(), b.data_ptr(), y.data_ptr(), co, st) == _lib.NS2_UNAVAILABLE


def test_embedding_padding_ids():
    table = rnd(11, 32, seed=93)
    ids = torch.tensor([[0, 5, -1, 9], [-3, 10, 2, 2]], device=DEV)
    out = ops.embedding(ids, table, pad_id=10)
    assert torch.equal(out, table[ids.masked_fill(ids < 0, 10)])


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("M,K,f", [(256, 64, 170), (500, 128, 341), (1024, 512, 1365), (300, 1024, 170)])      # K = 1024: a split-K candidate
def test_geglu(M, K, f, prec, gemm_kernel):
    x = rnd(M, K, seed=9)
    w = rnd(2 * f, K, seed=10, scale=1 / math.sqrt(K))
    b = rnd(2 * f, seed=11)
    pw = ops.PackedWeight(w, geglu=True, precision=prec)
    pb = ops.geglu_pack_bias(b, f)
    a = ops.split(x, precision=prec)
    out = ops.linear_geglu(pw, a, pb, precision=prec)
    h = exact(a)[:, :K] @ w.double().t() + b.double()
    ref = F.gelu(h[:, f:]) * h[:, :f]                              # NS2:1006-1007: first half x, second half gate
    assert out.ld == ops.round_up(f, 32)
    e = rel(ops.join(out, f), ref)
    assert e < TOL[prec] + (5e-4 if prec == 2 else 1e-5), f"rel err {e}"
    assert ops.join(out)[:, f:].abs().sum().item() == 0.0


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("B,N,K", [(2, 200, 64), (3, 135, 128), (2, 1024, 512), (2, 256, 1024), (1, 200, 1024)])   # K = 1024: split-K candidates (fast / generic V^T)
def test_qkv(B, N, K, prec, gemm_kernel):
    a_dim = 512
    x = rnd(B * N, K, seed=12)
    w = rnd(3 * a_dim, K, seed=13, scale=1 / math.sqrt(K))
    pw = ops.PackedWeight(w, precision=prec)
    a = ops.split(x, precision=prec)
    qk, vt = ops.linear_qkv(pw, a, seq_len=N, split_col=2 * a_dim, precision=prec)
    ref = exact(a)[:, :K] @ w.double().t()
    otol = TOL[prec] + (5e-4 if prec in (2, 4) else 1e-5)         # q / k / V^T are written as IEEE half at precisions 2 and 4
    assert (qk.fmt, vt.fmt) == ({3: "bf16", 1: "bf16", 2: "f16", 4: "f16"}[prec],) * 2
    assert rel(ops.join(qk), ref[:, : 2 * a_dim]) < otol
    v = ops.join(vt).reshape(B, a_dim, vt.ld)[:, :, :N]           # [B, a, N]
    vref = ref[:, 2 * a_dim:].reshape(B, N, a_dim).transpose(1, 2)
    assert rel(v, vref) < otol


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("B,N,C,dil", [(2, 200, 64, 2), (2, 300, 128, 64), (1, 1024, 512, 128)])
def test_wavenet_block(B, N, C, dil, prec, gemm_kernel):
    x = rnd(B * N, C, seed=14)
    wc = rnd(C, C, 3, seed=15, scale=1 / math.sqrt(3 * C))
    wr = rnd(C, C, 1, seed=16, scale=1 / math.sqrt(C))
    bc, br = rnd(C, seed=17), rnd(C, seed=18)
    film = rnd(B, 2 * C, seed=19)
    pw = ops.PackedWeight(wc, extra1x1=wr, precision=prec)
    a = ops.split(x, precision=prec)
    out = ops.wavenet_block(pw, a, N, dil, bc, br, film, precision=prec)
    xe = exact(a)[:, :C].reshape(B, N, C)
    h = conv_ref(xe, wc, bc, dil)
    g, bt = film.double()[:, None, :C], film.double()[:, None, C:]
    h = h * g + bt
    h = h.tanh() * h.sigmoid()
    ref = (h + conv_ref(xe, wr, br, 1)).reshape(B * N, C)          # NS2:627-636
    e = rel(ops.join(out, C), ref)
    assert e < TOL[prec] + (5e-4 if prec == 2 else 1e-5), f"rel err {e}"
    if prec == 4:      # the hybrid plan's block: same operands, dilated conv as ONE half product, res_conv with the correction terms
        out5 = ops.wavenet_block(pw, a, N, dil, bc, br, film, precision=5)
        e5 = rel(ops.join(out5, C), ref)
        assert e5 < TOL[2] + 5e-4, f"rel err {e5}"
        assert e5 > 2 * e, (e5, e)             # both kernels really took the half-product phase (the 128x128 one since round 4)


def attn_ref(q, k, v, scale):
    s = torch.einsum("bhid,bhjd->bhij", q, k) * scale
    return torch.einsum("bhij,bhjd->bhid", s.softmax(-1), v)


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("B,H,Nq,Nk", [(2, 8, 200, 200), (1, 8, 1024, 1024), (2, 8, 300, 32), (2, 8, 32, 135), (3, 4, 70, 16),
                                       (1, 2, 129, 65)])
def test_attention(B, H, Nq, Nk, prec):
    a_dim = H * 64
    q = rnd(B * Nq, a_dim, seed=20)
    k = rnd(B * Nk, a_dim, seed=21)
    v = rnd(B * Nk, a_dim, seed=22)
    qp, kp = asplit(q, prec), asplit(k, prec)
    vt_ld = ops.round_up(Nk, 32)
    # V^T planes [B, a, vt_ld]; poison the padding to prove the kernel masks it
    vt_f = torch.full((B, a_dim, vt_ld), float("nan"), device=DEV)
    vt_f[:, :, :Nk] = v.reshape(B, Nk, a_dim).transpose(1, 2)
    vt = asplit(vt_f.reshape(B * a_dim, vt_ld), prec, ldo=vt_ld)
    o = ops.attention(qp, kp, vt, B, H, Nq, Nk, precision=prec)
    assert o.fmt == {3: "bf16", 1: "bf16", 2: "f16", 4: "h8"}[prec]

    def heads(p, n):
        return exact(p).reshape(B, n, H, 64).permute(0, 2, 1, 3)
    ve = exact(vt).reshape(B, a_dim, vt_ld)[:, :, :Nk].reshape(B, H, 64, Nk).transpose(2, 3)
    ref = attn_ref(heads(qp, Nq), heads(kp, Nk), ve, 0.125).permute(0, 2, 1, 3).reshape(B * Nq, a_dim)
    got = ops.join(o)
    assert torch.isfinite(got).all()
    e = rel(got, ref)
    # precisions 2 and 4: one IEEE-half product for S and PV (P is rounded to half), output half / h8
    assert e < {3: 3e-5, 1: 2e-2, 2: 8e-4, 4: 5e-4}[prec], f"rel err {e}"


@pytest.mark.parametrize("prec", [3, 2, 4])
@pytest.mark.parametrize("D", [32, 128])
@pytest.mark.parametrize("B,H,Nq,Nk,masked", [(2, 4, 200, 200, False), (1, 2, 1024, 1024, False), (2, 3, 70, 135, True), (1, 2, 129, 16, False)])
def test_attention_head_dims(B, H, Nq, Nk, masked, D, prec):
    """round 6: the reference's `dim_head` keyword (NS2:814-831; Attend ATT:77-155, scale = dim_head ** -0.5) at 32 and 128:
    same kernel, head dimension as a template parameter; padding poisoned, key-padding mask on one case"""
    a_dim = H * D
    q, k, v = rnd(B * Nq, a_dim, seed=120), rnd(B * Nk, a_dim, seed=121), rnd(B * Nk, a_dim, seed=122)
    qp, kp = asplit(q, prec), asplit(k, prec)
    vt_ld = ops.round_up(Nk, 32)
    vt_f = torch.full((B, a_dim, vt_ld), float("nan"), device=DEV)
    vt_f[:, :, :Nk] = v.reshape(B, Nk, a_dim).transpose(1, 2)
    vt = asplit(vt_f.reshape(B * a_dim, vt_ld), prec, ldo=vt_ld)
    mask = None
    if masked:
        mask = (torch.rand(B, Nk, generator=torch.Generator().manual_seed(123)) > 0.4)
        mask[0, :5] = False
        mask = mask.to(DEV)
    o = ops.attention(qp, kp, vt, B, H, Nq, Nk, precision=prec, key_mask=mask, head_dim=D)

    def heads(p, n):
        return exact(p).reshape(B, n, H, D).permute(0, 2, 1, 3)
    ve = exact(vt).reshape(B, a_dim, vt_ld)[:, :, :Nk].reshape(B, H, D, Nk).transpose(2, 3)
    sim = torch.einsum("bhid,bhjd->bhij", heads(qp, Nq), heads(kp, Nk)) * D ** -0.5
    if masked:
        sim = sim.masked_fill(~mask[:, None, None, :], -torch.finfo(torch.float32).max)
    ref = torch.einsum("bhij,bhjd->bhid", sim.softmax(-1), ve).permute(0, 2, 1, 3).reshape(B * Nq, a_dim)
    got = ops.join(o)
    assert got.shape == ref.shape and torch.isfinite(got).all()
    e = rel(got, ref)
    assert e < {3: 3e-5, 2: 8e-4, 4: 5e-4}[prec], f"rel err {e}"


def test_attention_key_padding_mask():
    """ATT:136-138: masked keys get -finfo.max before the softmax (also when they are the first / middle keys)."""
    B, H, Nq, Nk = 2, 2, 70, 150
    a_dim = H * 64
    q, k, v = rnd(B * Nq, a_dim, seed=60), rnd(B * Nk, a_dim, seed=61), rnd(B * Nk, a_dim, seed=62)
    mask = torch.rand(B, Nk, generator=torch.Generator().manual_seed(63)) > 0.4
    mask[0, :5] = False
    mask = mask.to(DEV)
    qp, kp = ops.split(q), ops.split(k)
    vt_ld = ops.round_up(Nk, 32)
    vt_f = torch.zeros(B, a_dim, vt_ld, device=DEV)
    vt_f[:, :, :Nk] = v.reshape(B, Nk, a_dim).transpose(1, 2)
    vt = ops.split(vt_f.reshape(B * a_dim, vt_ld), ldo=vt_ld)
    o = ops.attention(qp, kp, vt, B, H, Nq, Nk, key_mask=mask)

    def heads(p, n):
        return exact(p).reshape(B, n, H, 64).permute(0, 2, 1, 3)
    ve = exact(vt).reshape(B, a_dim, vt_ld)[:, :, :Nk].reshape(B, H, 64, Nk).transpose(2, 3)
    sim = torch.einsum("bhid,bhjd->bhij", heads(qp, Nq), heads(kp, Nk)) * 0.125
    sim = sim.masked_fill(~mask[:, None, None, :], -torch.finfo(torch.float32).max)
    ref = torch.einsum("bhij,bhjd->bhid", sim.softmax(-1), ve).permute(0, 2, 1, 3).reshape(B * Nq, a_dim)
    assert rel(ops.join(o), ref) < 3e-5


def test_gemm_linearity_at_full_size():
    """size-independent property at the headline GEMM shape: f(a x + b y) == a f(x) + b f(y) for the causal conv."""
    B, N, C = 32, 1024, 512
    w = rnd(C, C, 3, seed=70, scale=0.03)
    pw = ops.PackedWeight(w)
    x, y = rnd(B * N, C, seed=71), rnd(B * N, C, seed=72)
    f = lambda t: ops.linear_f32(pw, ops.split(t), conv_taps=3, dilation=64, seq_len=N)   # noqa: E731
    lhs = f(2.0 * x - 0.5 * y)
    rhs = 2.0 * f(x) - 0.5 * f(y)
    assert rel(lhs, rhs) < 5e-5
    # causality: output at frame n depends only on frames <= n of the same utterance
    x2 = x.clone().reshape(B, N, C)
    x2[:, 700:] += 1.0
    d = (f(x2.reshape(B * N, C)) - f(x)).reshape(B, N, C)
    assert d[:, :700].abs().max().item() == 0.0 and d[:, 700:].abs().max().item() > 0.0


def test_rvq_roundtrip_full_size():
    """BASELINE config 4 size: residual + sum of selected codes reconstructs the latents; decode(codes) == emb."""
    cb = make_input("codebooks", (8, 1024, 128), seed=81).to(DEV)
    x = make_input("latents", (32 * 1024, 128), seed=82).to(DEV)
    codes, emb, resid = ops.rvq_encode(x, cb, want_residual=True)
    assert codes.min().item() >= 0 and codes.max().item() < 1024
    assert torch.equal(ops.rvq_decode(codes, cb), emb)
    assert (emb + resid - x).abs().max().item() < 1e-4
    # each stage picks the nearest code: re-encoding the residual-free reconstruction of stage 0 is idempotent
    c0 = cb[0][codes[:, 0]]
    codes2, _ = ops.rvq_encode(c0.contiguous(), cb[:1].contiguous())
    assert torch.equal(codes2[:, 0], codes[:, 0])


def test_attention_spiked_softmax():
    """online-softmax rescale path: one key dominates late in the sequence (guide rule 26)."""
    B, H, Nq, Nk = 1, 1, 64, 256
    q = rnd(B * Nq, 64, seed=23)
    k = rnd(B * Nk, 64, seed=24)
    v = rnd(B * Nk, 64, seed=25)
    k[200] = q[5] * 6.0                                           # huge score for (q5, k200) in the 4th key tile
    qp, kp = ops.split(q), ops.split(k)
    vp = ops.split(v.reshape(Nk, 64).t().contiguous(), ldo=Nk)
    o = ops.attention(qp, kp, vp, B, H, Nq, Nk)
    ve = exact(vp).t().reshape(1, 1, Nk, 64)
    ref = attn_ref(exact(qp).reshape(1, 1, Nq, 64), exact(kp).reshape(1, 1, Nk, 64), ve, 0.125).reshape(Nq, 64)
    assert rel(ops.join(o), ref) < 3e-5


@pytest.mark.parametrize("M,d,seq,adaptive,gamma", [(400, 64, 100, True, False), (512, 512, 128, True, False),
                                                    (96, 128, 0, False, True), (64, 96, 0, False, False)])
def test_rmsnorm(M, d, seq, adaptive, gamma):
    x = rnd(M, d, seed=26, scale=2.0)
    g = (1 + 0.1 * rnd(d, seed=27)) if gamma else None
    cond = rnd(M // seq, 2 * d + 5, seed=28) if adaptive else None
    out, of = ops.rmsnorm(x, seq_len=seq, gamma=g, cond=cond, want_f32=True)
    ref = F.normalize(x.double(), dim=-1) * math.sqrt(d)
    if gamma:
        ref = ref * g.double()
    if adaptive:
        c = cond.double().repeat_interleave(seq, dim=0)
        ref = ref * c[:, :d] + c[:, d:2 * d]
    assert rel(of, ref) < 1e-6
    assert rel(ops.join(out, d), ref) < 1e-5


def test_rmsnorm_zero_row():
    x = torch.zeros(8, 64, device=DEV)
    out, of = ops.rmsnorm(x, want_f32=True)
    assert of.abs().max().item() == 0.0                            # F.normalize eps clamp, no NaN


@pytest.mark.parametrize("B,K,J,act", [(32, 2048, 3000, 0), (4, 513, 2048, 1), (40, 100, 70, 1), (1, 4096, 1024, 0),
                                       (32, 2048, 40000, 0),                       # the "wide" launch shape (>= 256 blocks of 1024 columns)
                                       (32, 2048, 1024, 0), (32, 1024, 2048, 0), (1024, 32, 2048, 0), (13, 64, 260, 1)])   # training roles: y, dx, dW
def test_skinny_linear(B, K, J, act):
    """elementwise.hip: wide blocks (256 threads x 4 columns x 32 rows) when they fill the chip, else one-wave blocks of 256
    columns x 8 rows; the K split (so every sum's order) is the same in both."""
    x = rnd(B, K, seed=29)
    w = rnd(J, K, seed=30, scale=1 / math.sqrt(K))
    b = rnd(J, seed=31)
    y = ops.skinny_linear(x, w.t().contiguous(), b, act)
    ref = x.double() @ w.double().t() + b.double()
    if act:
        ref = F.silu(ref)
    assert rel(y, ref) < 2e-6


def test_skinny_linear_uniform_batch_rows():
    """Round 3: a block whose batch rows are identical over its K range multiplies once for all rows (the sampler's time
    conditioning).  Same FMA order per row: a uniform batch equals the single-row call bit for bit; a batch that is uniform only
    over part of K (conditioned model: time half shared, prompt half per utterance) matches fp64; a NaN row takes the per-row body."""
    B, K, J = 32, 4096, 3072
    w = rnd(J, K, seed=130, scale=1 / math.sqrt(K)).t().contiguous()
    b = rnd(J, seed=131)
    x1 = rnd(1, K, seed=132)
    y1 = ops.skinny_linear(x1, w, b, 0)
    yu = ops.skinny_linear(x1.expand(B, K).contiguous(), w, b, 0)
    assert torch.equal(yu, y1.expand(B, J))
    xm = x1.expand(B, K).clone()
    xm[:, K // 2:] = rnd(B, K // 2, seed=133)
    ym = ops.skinny_linear(xm, w, b, 1)
    ref = F.silu(xm.double() @ w.double() + b.double())
    assert rel(ym, ref) < 2e-6
    assert rel(ym[5:6], ops.skinny_linear(xm[5:6].contiguous(), w, b, 1)) < 1e-6          # a row in the batch == the row alone
    xn = x1.expand(4, K).clone()
    xn[:, 7] = float("nan")
    assert torch.isnan(ops.skinny_linear(xn, w, b, 0)).all()


def test_time_embed():
    dim, B = 128, 5
    freqs = rnd(dim // 2, seed=32)
    w = rnd(4 * dim, dim + 1, seed=33, scale=0.1)
    b = rnd(4 * dim, seed=34)
    t = torch.rand(B, generator=torch.Generator().manual_seed(35)).to(DEV)
    y = ops.time_embed(t, freqs, w.t().contiguous(), b)
    fr = t[:, None] * freqs[None] * 2 * math.pi                     # NS2:115-119 in fp32 like the reference
    feat = torch.cat((t[:, None], fr.sin(), fr.cos()), dim=-1)
    ref = F.silu(F.linear(feat.double(), w.double(), b.double()))
    assert rel(y, ref) < 5e-6


@pytest.mark.parametrize("objective", ["v", "eps", "x0"])
def test_ddim_step(objective):
    from oracle import ns2_oracle as O
    B, n, d = 3, 50, 64
    audio, mo = rnd(B, n, d, seed=36), rnd(B, n, d, seed=37)
    t = torch.tensor([1.0, 0.6, 0.002])
    tn = torch.tensor([0.9, 0.5, 0.0])
    y = ops.ddim_step(audio, mo, t.to(DEV), tn.to(DEV), objective=objective)
    ref = O.ddim_update(audio.cpu(), mo.cpu(), t, tn, objective=objective)
    assert rel(y.cpu(), ref) < 2e-6


def test_cfg_mix():
    a, b = rnd(1000, seed=38), rnd(1000, seed=39)
    y = ops.cfg_mix(a, b, 1.7)
    assert torch.allclose(y, b + (a - b) * 1.7, atol=1e-6)


@pytest.mark.parametrize("name", ["rvq_small", "rvq_full"])
def test_rvq_golden(name):
    fix = torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)
    cb = make_input("codebooks", (fix["nq"], fix["codes"], fix["d"]), seed=fix["codebook_seed"])
    x = make_input("latents", (fix["m"], fix["d"]), seed=fix["latent_seed"]) * fix["latent_scale"]
    codes, emb, resid, ties = ops.rvq_encode(x.to(DEV), cb.to(DEV), want_residual=True, count_ties=True)
    assert torch.equal(codes.cpu(), fix["indices"]), f"{(codes.cpu() != fix['indices']).sum().item()} index mismatches"
    assert torch.equal(emb.cpu(), fix["emb"])                       # same fp32 add order as HFENC:440-447
    assert torch.equal(ops.rvq_decode(codes, cb.to(DEV)).cpu(), fix["emb"])
    _, _, r_ref = R.rvq_encode(x, cb)
    assert torch.equal(resid.cpu(), r_ref)


@pytest.mark.parametrize("M", [1, 127, 128, 4099])
def test_rvq_vs_oracle_ragged(M):
    cb = make_input("codebooks", (8, 1024, 128), seed=41)
    x = make_input("latents", (M, 128), seed=42)
    codes, emb = ops.rvq_encode(x.to(DEV), cb.to(DEV))
    c_ref, e_ref, _ = R.rvq_encode(x, cb)
    mism = codes.cpu() != c_ref
    if mism.any():                                                 # only fp32 near-ties may differ: report them
        marg = R.top2_margins(x, cb, c_ref)
        first = mism.float().argmax(dim=-1)                        # first differing stage per row
        rows = mism.any(dim=-1).nonzero().flatten()
        assert all(marg[r, first[r]] < 1e-3 for r in rows), "index mismatch that is not an fp32 near-tie"
    else:
        assert torch.equal(emb.cpu(), e_ref)


def test_rvq_exact_tie_prefers_first_index():
    cb = make_input("codebooks", (2, 64, 128), seed=43)
    cb[0, 40] = cb[0, 7]                                           # duplicate code: argmax must return the first
    x = cb[0, 7][None].repeat(5, 1) + 0.01 * make_input("n", (5, 128), seed=44)
    codes, _ = ops.rvq_encode(x.to(DEV), cb.to(DEV))
    assert (codes[:, 0].cpu() == 7).all()
