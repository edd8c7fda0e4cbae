"""Thin torch-tensor wrappers over the op-level C ABI of libns2hip (include/ns2hip.h).

PyTorch is plumbing here: device memory and the current HIP stream.  Every function launches hand-written HIP
kernels through ctypes; nothing in this module computes with torch ops.
"""
from typing import Optional

import torch

from . import _lib
from ._lib import check

def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _f32(t: torch.Tensor) -> torch.Tensor:
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), "expected a contiguous fp32 CUDA tensor"
    return t


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class Planes:
    """A split-plane matrix [rows, ld] held by the library's layout (csrc/ns2_common.h).  `ld` is always the LOGICAL column
    count (a multiple of 32).  Formats (`fmt`):

      "bf16"  with a lo plane ("interleaved", what precision 3 needs) `buf` is ONE bf16 tensor [rows, 2*ld]: every 32 logical
              columns occupy a 128-byte line [hi(32) | lo(32)], so the lo pointer is the hi pointer + 32 elements; without
              (precision 1) `buf` is the dense [rows, ld] hi plane;
      "f16"   one dense IEEE-half plane [rows, ld] (precision 2, and the attention operands of precision 4);
      "h8"    interleaved 128-byte lines [half(32) | e5m2(x)(32 bytes) | e5m2((x - half(x)) * 2^12)(32 bytes)] (precision 4).
    """
    __slots__ = ("buf", "rows", "ld", "has_lo", "fmt")

    def __init__(self, buf: torch.Tensor, rows: int, ld: int, has_lo: bool, fmt: str = "bf16"):
        assert fmt in ("bf16", "f16", "h8")
        assert not (fmt == "f16" and has_lo), "IEEE-half planes have no lo part"
        assert not (fmt == "h8" and not has_lo), "h8 planes are interleaved lines"
        self.buf, self.rows, self.ld, self.has_lo, self.fmt = buf, rows, ld, has_lo, fmt

    @property
    def f16(self):
        return self.fmt == "f16"

    @property
    def precision(self):
        """a precision whose kernels read / write this format"""
        return {"bf16": 3 if self.has_lo else 1, "f16": 2, "h8": 4}[self.fmt]

    @property
    def device(self):
        return self.buf.device

    @property
    def hi(self) -> int:
        """device address of the hi plane (what the C ABI takes as x_hi)"""
        return self.buf.data_ptr()

    @property
    def lo(self) -> Optional[int]:
        """device address of the second half of the lines = hi + 32 elements, or None for a dense single plane"""
        return self.buf.data_ptr() + 64 if self.has_lo else None

    def hi_plane(self) -> torch.Tensor:
        """the logical hi plane as a dense [rows, ld] 16-bit tensor (copy)"""
        if not self.has_lo:
            return self.buf.reshape(self.rows, self.ld).clone()
        return self.buf.reshape(self.rows, self.ld // 32, 2, 32)[:, :, 0, :].reshape(self.rows, self.ld).contiguous()

    def byte_planes(self):
        """h8 only: (e5m2(x), e5m2((x - half(x)) * 2^12)) as uint8 tensors [rows, ld]"""
        assert self.fmt == "h8"
        b = self.buf.reshape(self.rows, self.ld // 32, 2, 32)[:, :, 1, :].contiguous().view(torch.uint8)
        b = b.reshape(self.rows, self.ld // 32, 64)
        return b[:, :, :32].reshape(self.rows, self.ld).contiguous(), b[:, :, 32:].reshape(self.rows, self.ld).contiguous()

    def hi_only(self) -> "Planes":
        return Planes(self.hi_plane(), self.rows, self.ld, False, "f16" if self.fmt != "bf16" else "bf16")


def _fmt_of(precision: int, attention_operand: bool = False):
    """(has_lo, fmt) of what a kernel writes at `precision`; attention operands (q, k, V^T) are IEEE half at precision 4"""
    if precision == 2 or (precision == 4 and attention_operand):
        return False, "f16"
    if precision == 4:
        return True, "h8"
    return True, "bf16"              # precision 1 kernels read the hi plane of the interleaved layout too


def empty_planes(rows: int, cols: int, device, lo: bool = True, zero: bool = False, fmt: str = "bf16") -> Planes:
    lo = (lo and fmt == "bf16") or fmt == "h8"
    assert cols % 32 == 0 or not lo, "interleaved split planes come in 32-column blocks"
    alloc = torch.zeros if zero else torch.empty
    dt = torch.bfloat16 if fmt == "bf16" else torch.float16
    return Planes(alloc(rows, cols * (2 if lo else 1), dtype=dt, device=device), rows, cols, lo, fmt)


def _out_planes(rows, cols, device, precision, attention_operand=False, zero=False):
    lo, fmt = _fmt_of(precision, attention_operand)
    return empty_planes(rows, cols, device, lo, zero, fmt)


def split(x: torch.Tensor, ldo: Optional[int] = None, lo: bool = True, precision: int = 3) -> Planes:
    """fp32 [M, d] -> split planes [M, ldo] (zero padded) in the operand format of `precision`."""
    x = _f32(x)
    M, d = x.shape
    ldo = ldo or round_up(d, 32)
    has_lo, fmt = _fmt_of(precision)
    out = empty_planes(M, ldo, x.device, lo and has_lo, fmt=fmt)
    check(_lib.load().ns2_split_f32(x.data_ptr(), d, M, d, out.hi, out.lo, ldo, out.precision, _stream()), "ns2_split_f32")
    return out


def join(p: Planes, d: Optional[int] = None) -> torch.Tensor:
    M, ld = p.rows, p.ld
    d = d or ld
    out = torch.empty(M, d, dtype=torch.float32, device=p.device)
    check(_lib.load().ns2_join_f32(p.hi, p.lo, ld, out.data_ptr(), d, M, d, p.precision, _stream()), "ns2_join_f32")
    return out


class PackedWeight:
    """Library-owned packed weight (ns2_weight)."""

    def __init__(self, w: torch.Tensor, geglu: bool = False, extra1x1: Optional[torch.Tensor] = None, precision: int = 3):
        """precision 1 / 3: interleaved bf16 planes (serve both); 2: dense IEEE half; 4: h8 lines (each serves only itself)"""
        import ctypes
        w = _f32(w)
        self.precision = precision
        self.rows, self.cols = w.shape[0], w.shape[1]
        self.taps = w.shape[2] if w.ndim == 3 else 1
        self.geglu = geglu
        self.has_extra = extra1x1 is not None
        self.cols_p = round_up(self.cols, 32)
        h = ctypes.c_void_p()
        ex = _f32(extra1x1) if extra1x1 is not None else None
        check(_lib.load().ns2_weight_pack(w.data_ptr(), self.rows, self.cols, self.taps, int(geglu), _p(ex), precision,
                                          ctypes.byref(h), _stream()), "ns2_weight_pack")
        self.handle = h

    def tile_conv3(self) -> "PackedWeight":
        """give a k = 3 conv weight packed at precision 2 the tiled images of the dedicated FF causal conv kernel (include/ns2hip.h)"""
        check(_lib.load().ns2_weight_tile_conv3(self.handle, _stream()), "ns2_weight_tile_conv3")
        return self

    def tile_wavenet(self) -> "PackedWeight":
        """give a WavenetResBlock weight (taps = 3 + extra1x1) packed at precision 4 the tiled images of the lean block kernel (include/ns2hip.h)"""
        check(_lib.load().ns2_weight_tile_wavenet(self.handle, _stream()), "ns2_weight_tile_wavenet")
        return self

    def tile_linear(self) -> "PackedWeight":
        """give a linear weight packed at precision 4 the tiled images of the lean mixed linear kernel (include/ns2hip.h)"""
        check(_lib.load().ns2_weight_tile_linear(self.handle, _stream()), "ns2_weight_tile_linear")
        return self

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.load().ns2_weight_free(self.handle)
                self.handle = None
        except Exception:
            pass


def linear_f32(w: PackedWeight, a: Planes, M: Optional[int] = None, bias=None, resid=None, conv_taps=0, dilation=1,
               seq_len=0, precision=3, pad_left=-1, act=0) -> torch.Tensor:
    M = M or a.rows
    out = torch.empty(M, w.rows, dtype=torch.float32, device=a.device)
    check(_lib.load().ns2_linear_f32(w.handle, a.hi, a.lo, a.ld, M, conv_taps, dilation, seq_len, _p(bias),
                                     _p(resid), w.rows, out.data_ptr(), w.rows, pad_left, act, precision, _stream()), "ns2_linear_f32")
    return out


def conv3_input_ld(cols: int) -> int:
    """row length (elements) of the dense-half activations the dedicated FF causal conv kernel reads for `cols` input channels"""
    return _lib.load().ns2_conv3_input_ld(cols)


def linear_split(w: PackedWeight, a: Planes, bias=None, conv_taps=0, dilation=1, seq_len=0, precision=3, ldo=None, pad_left=-1,
                 act=0, out_precision=None) -> Planes:
    """out_precision: write the output planes in the operand format of another precision (ns2_linear_split_as), e.g. FMT_H8 lines (4)
    from a precision-2 product -- what the hybrid plan's FF causal conv does"""
    M = a.rows
    ldo = ldo or round_up(w.rows, 32)
    out = _out_planes(M, ldo, a.device, out_precision or precision)
    if out_precision is None:
        check(_lib.load().ns2_linear_split(w.handle, a.hi, a.lo, a.ld, M, conv_taps, dilation, seq_len, _p(bias),
                                           out.hi, out.lo, ldo, pad_left, act, precision, _stream()), "ns2_linear_split")
    else:
        check(_lib.load().ns2_linear_split_as(w.handle, a.hi, a.lo, a.ld, M, conv_taps, dilation, seq_len, _p(bias),
                                              out.hi, out.lo, ldo, pad_left, act, precision, out_precision, _stream()), "ns2_linear_split_as")
    return out


def geglu_pack_bias(bias: torch.Tensor, f: int) -> torch.Tensor:
    n = round_up(2 * round_up(f, 32), 256)
    out = torch.empty(n, dtype=torch.float32, device=bias.device)
    check(_lib.load().ns2_geglu_pack_bias(_f32(bias).data_ptr(), f, out.data_ptr(), n, _stream()), "ns2_geglu_pack_bias")
    return out


def linear_geglu(w: PackedWeight, a: Planes, packed_bias: torch.Tensor, precision=3) -> Planes:
    M = a.rows
    f = w.rows // 2
    ldo = round_up(f, 32)
    out = _out_planes(M, ldo, a.device, precision)
    check(_lib.load().ns2_linear_geglu(w.handle, a.hi, a.lo, a.ld, M, packed_bias.data_ptr(), out.hi,
                                       out.lo, ldo, precision, _stream()), "ns2_linear_geglu")
    return out


def linear_qkv(w: PackedWeight, a: Planes, seq_len: int, split_col: int, precision=3):
    """returns (row-major planes [M, split_col], transposed planes [B * (rows - split_col), vt_ld])."""
    M = a.rows
    B = M // seq_len
    vt_ld = round_up(seq_len, 32)
    out = _out_planes(M, split_col, a.device, precision, attention_operand=True)
    vt_rows = w.rows - split_col
    vt = _out_planes(B * vt_rows, vt_ld, a.device, precision, attention_operand=True, zero=True)
    check(_lib.load().ns2_linear_qkv(w.handle, a.hi, a.lo, a.ld, M, seq_len, split_col, out.hi,
                                     out.lo, split_col, vt.hi, vt.lo, vt_ld, precision, _stream()),
          "ns2_linear_qkv")
    return out, vt


def wavenet_block(w: PackedWeight, a: Planes, seq_len: int, dilation: int, conv_bias, res_bias, film: torch.Tensor,
                  precision=3) -> Planes:
    M = a.rows
    ldo = round_up(w.rows, 32)
    out = _out_planes(M, ldo, a.device, 4 if precision == 5 else precision)      # 5: precision-4 planes, dilated conv as one half product
    check(_lib.load().ns2_wavenet_block(w.handle, a.hi, a.lo, a.ld, M, seq_len, dilation, conv_bias.data_ptr(),
                                        res_bias.data_ptr(), _f32(film).data_ptr(), film.shape[1], out.hi,
                                        out.lo, ldo, precision, _stream()), "ns2_wavenet_block")
    return out


def attention(q: Planes, k: Planes, vt: Planes, B: int, H: int, Nq: int, Nk: int, q_col0=0, k_col0=0, scale=None,
              precision=3, key_mask: Optional[torch.Tensor] = None, head_dim: int = 64) -> Planes:
    """vt: transposed value planes [B * H*head_dim, vt_ld]; key_mask: optional bool/uint8 [B, Nk], True = attend (ATT:92-94);
    head_dim 32 / 64 / 128, scale defaults to head_dim ** -0.5 (ATT:128)."""
    if scale is None:
        scale = head_dim ** -0.5
    out = _out_planes(B * Nq, H * head_dim, q.device, precision)
    km = None
    if key_mask is not None:
        km = key_mask.to(torch.uint8).contiguous()
        assert km.shape == (B, Nk)
    check(_lib.load().ns2_attention_hd(q.hi, q.lo, q.ld, q_col0, k.hi, k.lo, k.ld,
                                       k_col0, vt.hi, vt.lo, vt.ld, out.hi, out.lo,
                                       H * head_dim, B, H, Nq, Nk, scale, _p(km), precision, head_dim, _stream()), "ns2_attention_hd")
    return out


def rmsnorm(x: torch.Tensor, seq_len: int = 0, gamma=None, cond=None, want_f32=False, precision: int = 3):
    x = _f32(x)
    M, d = x.shape
    ldo = round_up(d, 32)
    out = _out_planes(M, ldo, x.device, precision)
    of = torch.empty(M, d, dtype=torch.float32, device=x.device) if want_f32 else None
    check(_lib.load().ns2_rmsnorm(x.data_ptr(), d, M, d, seq_len, _p(gamma), _p(cond), cond.shape[1] if cond is not None else 0,
                                  out.hi, out.lo, ldo, _p(of), d, precision, _stream()), "ns2_rmsnorm")
    return (out, of) if want_f32 else out


def embedding(ids: torch.Tensor, table: torch.Tensor, pad_id: int) -> torch.Tensor:
    """ids [..] int64 (negative = padding -> pad_id), table [V, dim] -> [.., dim] fp32."""
    ids = ids.contiguous().to(torch.int64)
    out = torch.empty(*ids.shape, table.shape[1], dtype=torch.float32, device=table.device)
    check(_lib.load().ns2_embedding(ids.data_ptr(), _f32(table).data_ptr(), out.data_ptr(), ids.numel(), table.shape[1], pad_id,
                                    _stream()), "ns2_embedding")
    return out


def _skinny_ws(B, K, J, device):
    n = _lib.load().ns2_skinny_linear_workspace_bytes(B, K, J)
    return torch.empty(max(n, 16), dtype=torch.uint8, device=device), n


def skinny_linear(x: torch.Tensor, wt: torch.Tensor, bias=None, act=0, use_workspace=True) -> torch.Tensor:
    """x [B, K] @ wt [K, J] (+bias) ; act 1 = SiLU.  The split-K scratch is caller-owned (here: a torch tensor)."""
    x, wt = _f32(x), _f32(wt)
    B, K = x.shape
    J = wt.shape[1]
    out = torch.empty(B, J, dtype=torch.float32, device=x.device)
    ws, n = _skinny_ws(B, K, J, x.device) if use_workspace else (None, 0)
    check(_lib.load().ns2_skinny_linear(x.data_ptr(), K, wt.data_ptr(), _p(bias), out.data_ptr(), J, B, K, J, act, _p(ws), n,
                                        _stream()), "ns2_skinny_linear")
    return out


def time_embed(times: torch.Tensor, freqs: torch.Tensor, wt: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    B = times.shape[0]
    dim = freqs.shape[0] * 2
    dt = wt.shape[1]
    feat = torch.empty(B, dim + 1, dtype=torch.float32, device=times.device)
    out = torch.empty(B, dt, dtype=torch.float32, device=times.device)
    ws, n = _skinny_ws(B, dim + 1, dt, times.device)
    check(_lib.load().ns2_time_embed(_f32(times).data_ptr(), _f32(freqs).data_ptr(), _f32(wt).data_ptr(), _f32(bias).data_ptr(),
                                     feat.data_ptr(), out.data_ptr(), dt, B, dim, dt, ws.data_ptr(), n, _stream()), "ns2_time_embed")
    return out


def saturation_count(reset: bool = True, device=None) -> int:
    """conversions to IEEE half (precisions "half" / "mixed") that met a value outside the half range since the last reset, on
    `device` (default: the current one).  Synchronises; see include/ns2hip.h."""
    import ctypes
    n = ctypes.c_int64(0)
    with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
        check(_lib.load().ns2_saturation_count(int(reset), ctypes.byref(n)), "ns2_saturation_count")
    return n.value


OBJECTIVES = {"v": 0, "eps": 1, "x0": 2}
SCHEDULES = {"sigmoid": 0, "cosine": 1, "linear": 2}


def ddim_step(audio, model_out, times, times_next, objective="v", schedule="sigmoid", scale=1.0, out=None):
    audio, model_out = _f32(audio), _f32(model_out)
    B = audio.shape[0]
    per = audio.numel() // B
    out = torch.empty_like(audio) if out is None else out
    check(_lib.load().ns2_ddim_step(audio.data_ptr(), model_out.data_ptr(), out.data_ptr(), _f32(times).data_ptr(),
                                    _f32(times_next).data_ptr(), B, per, OBJECTIVES[objective], SCHEDULES[schedule], float(scale),
                                    _stream()), "ns2_ddim_step")
    return out


def cfg_mix(cond_out, null_out, cond_scale: float, out=None):
    out = torch.empty_like(cond_out) if out is None else out
    check(_lib.load().ns2_cfg_mix(_f32(cond_out).data_ptr(), _f32(null_out).data_ptr(), out.data_ptr(), cond_out.numel(),
                                  float(cond_scale), _stream()), "ns2_cfg_mix")
    return out


def rvq_prepare(codebooks: torch.Tensor) -> torch.Tensor:
    cb = _f32(codebooks)
    Q, C, D = cb.shape
    out = torch.empty(Q, C, dtype=torch.float32, device=cb.device)
    check(_lib.load().ns2_rvq_prepare(cb.data_ptr(), out.data_ptr(), Q, C, D, _stream()), "ns2_rvq_prepare")
    return out


def rvq_encode(x: torch.Tensor, codebooks: torch.Tensor, cb_norm: Optional[torch.Tensor] = None, tie_eps: float = 1e-4,
               want_residual=False, count_ties=False, want_emb=True):
    """x [M, 128] fp32, codebooks [Q, C, 128] -> codes [M, Q] int64, emb [M, 128] (, residual, n_near_ties).
    want_emb=False skips the decode launch that sums the selected code vectors (emb is returned as None)."""
    x, cb = _f32(x), _f32(codebooks)
    M, D = x.shape
    Q, C, _ = cb.shape
    cb_norm = rvq_prepare(cb) if cb_norm is None else cb_norm
    codes = torch.empty(M, Q, dtype=torch.int64, device=x.device)
    emb = torch.empty(M, D, dtype=torch.float32, device=x.device) if want_emb else None
    resid = torch.empty(M, D, dtype=torch.float32, device=x.device) if want_residual else None
    ties = torch.zeros(1, dtype=torch.int32, device=x.device) if count_ties else None
    check(_lib.load().ns2_rvq_encode(x.data_ptr(), cb.data_ptr(), cb_norm.data_ptr(), codes.data_ptr(), _p(emb), _p(resid),
                                     _p(ties), M, Q, C, D, float(tie_eps), _stream()), "ns2_rvq_encode")
    res = [codes, emb]
    if want_residual:
        res.append(resid)
    if count_ties:
        res.append(ties)
    return tuple(res)


def rvq_decode(codes: torch.Tensor, codebooks: torch.Tensor) -> torch.Tensor:
    cb = _f32(codebooks)
    Q, C, D = cb.shape
    M = codes.shape[0]
    emb = torch.empty(M, D, dtype=torch.float32, device=cb.device)
    check(_lib.load().ns2_rvq_decode(codes.contiguous().data_ptr(), cb.data_ptr(), emb.data_ptr(), M, Q, C, D, _stream()),
          "ns2_rvq_decode")
    return emb
