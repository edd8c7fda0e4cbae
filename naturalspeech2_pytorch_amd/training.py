"""Training path of `Model`: forward + backward of the denoiser in the HIP kernels of libns2hip.

Reference call sites: `pred = self.model(noised_audio, times, prompt = prompt, cond = cond)` under autograd (NS2:1635), the
v-target MSE with min-SNR weight (NS2:1637-1666), `accelerator.backward(loss)` (NS2:1886).

How it is cut.  The differentiable graph consists of a handful of coarse `torch.autograd.Function`s whose boundaries are the
fp32 residual-stream tensors [B*N, dim] of the reference's forward (NS2:929-1000):

    GemmFn            nn.Linear / CausalConv1d / 1x1 conv (+ bias, + residual)                 NS2:583-595, 718-725
    WavenetBlockFn    dilated conv -> FiLM -> tanh*sigmoid gate, + res_conv                   NS2:597-642
    AttnFn            adaptive RMSNorm -> q, k, v -> attention -> to_out, + residual         NS2:1029-1069, 727-746, ATT:77-155
    FeedForwardFn     adaptive RMSNorm -> Linear -> GEGLU -> causal conv k3 -> Linear, + res  NS2:1004-1025
    NormLinearFn      RMSNorm(gamma) -> Linear                                                NS2:781-784

Inside a Function everything is a launch of libns2hip (`HipBackend` below; C ABI: include/ns2hip.h "training"): activations
travel between the GEMMs as bf16 hi/lo operand planes, the backward contractions are the FORWARD GEMM kernels on re-packed
weights (dgrad) and on transposed planes with a fixed-slot split-K (wgrad), attention backward is a flash kernel that recomputes
P from the forward's log-sum-exp.  Arithmetic: precision 3 ("exact", bf16 x3 products, fp32 accumulate) whatever inference
precision the `Model` was built with -- gradients match the reference's fp32 autograd to ~1e-5.

Rows = batch entries: every conditioning Linear ([B, dim_cond] -> [B, 2 dim] for FiLM / adaptive norms, `to_time_cond.1`,
`to_prompt_cond.1`: NS2:623, 744, 841, 860) is `SkinnyLinearFn` -- the fp32 weight-streaming kernel of the inference path in all
three roles (y = x W^T + b, dx = dy W, dW = dy^T x).  What stays in PyTorch ops are pointwise glue on [B, *] rows (sin / cos of the
time embedding, SiLU, cat, mean-pool, `torch.where` null selects) and `torch.cat` of the
PerceiverResampler's context (NS2:1060-1061).  Autograd chains them with the Functions above; since round 5 the 32-token resampler
of the conditioned model (NS2:532-579) runs on the same Functions (`_resampler`), so no contraction of `Model`'s backward is a torch op.

`Backend` is the seam the CPU tests use: `tests/emu_backend.py` restates every backend call with plain torch ops on CPU, so the
chain rule, tap flips, shifts and layouts of THIS file are checked against torch autograd without a GPU; the kernels behind
`HipBackend` are checked one by one and end to end on the MI355X (`tests/test_backward_gpu.py`).
"""
import ctypes
import math
import weakref

import torch
import torch.nn.functional as F

from . import _lib, ops
from ._lib import check
from .ops import Planes, round_up


# =============================================================================================== HIP backend
class TPlanes:
    """transposed operand planes: `rows` rows of `ld` token columns, interleaved 128-byte lines along the token axis --
    bf16 [hi32 | lo32] (precision 3) or FMT_H8 [half32 | e5m2 | e5m2 remainder] (precision 4): 4 bytes per element either way"""
    __slots__ = ("buf", "rows", "ld", "precision")

    def __init__(self, rows, ld, device, precision=3):
        self.buf = torch.empty(rows, 2 * ld, dtype=torch.bfloat16, device=device)
        self.rows, self.ld, self.precision = rows, ld, precision

    def ptr(self, row_off=0):
        return self.buf.data_ptr() + row_off * 4 * self.ld          # 2 planes x 2 bytes per logical column


def _s():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


class _PackedCache:
    """Packed weights of a training run: packed once, refreshed IN PLACE (`ns2_weight_update`: no allocation, no
    synchronisation).  When: the first request of every training PASS (`begin_pass`, called by `model_forward_train`) for a weight
    that requires grad -- and whenever a version counter moved.  The version counter alone is not enough: PyTorch's fused optimizers
    (`torch.optim.Adam(fused=True)`, what `accelerate` picks on GPUs) update the parameters in place WITHOUT bumping it, and the
    forward would go on multiplying the weights of step 0 (round 5: found through bench.py's `loss_mixed` / `loss_composite` of the
    same iteration, 0.3700 against 0.3630; tests/test_round5_gpu.py::test_fused_optimizer_steps_reach_the_packed_weights).
    A frozen weight is re-packed only when its version moves."""

    LEAN = False        # tile images of the lean mixed linear kernel for the training packs: measured 100.1 vs 98.8 ms per d512 / L12 step WITH them
                        # (the per-pass re-tiling launches cost more than the eligible products gain; tools/exp_graphed_train.py --lean): off

    def __init__(self, precision=3):
        self.map = {}
        self.precision = precision
        self.pass_id = 0
        self._table = None          # (signature, device table, parts kept alive, n, total_blocks): every trainable pack, one launch per pass
        self.lean = self.LEAN       # give mixed linear packs the tile images of the lean kernel (class attribute: the A/B of tools/exp_graphed_train.py)

    # ---- one launch per pass (ns2_weights_repack): possible once every entry has told where its values live (`parts`)
    @staticmethod
    def _part(lib_part, pw, p, mode, row0, col0):
        """`mode`: "n" = the pack's rows / columns / taps are the parameter's; "t" = transposed ([C, R(, T)] from [R, C(, T)]);
        "tf" = transposed with the taps flipped (the dgrad weight of a causal conv)"""
        R, C = p.shape[0], p.shape[1]
        T = p.shape[2] if p.ndim == 3 else 1
        assert p.is_contiguous() and p.dtype == torch.float32
        base = p.data_ptr()
        if mode == "n":
            sr, sc, st, rows, cols = C * T, T, 1, R, C
        elif mode == "t":
            sr, sc, st, rows, cols = T, C * T, 1, C, R
        else:
            sr, sc, st, rows, cols = T, C * T, -1, C, R
            base += 4 * (T - 1)
        return lib_part(pw.handle, base, sr, sc, st, row0, rows, col0, cols)

    def _signature(self):
        # (storage and trainability of every parameter: a re-allocated, frozen or unfrozen parameter rebuilds the table)
        return tuple((k, tuple((p.data_ptr(), p.requires_grad) for p in (r() for r in v[4]) if p is not None)) for k, v in self.map.items())

    def _repack_all(self):
        """every trainable entry with a `parts` description in one launch; False when some entry cannot be described"""
        if not self.map or any(v[6] is None or any(r() is None for r in v[4]) for v in self.map.values()):
            return False
        lib = _lib.load()
        sig = self._signature()
        if self._table is None or self._table[0] != sig:
            parts, keys = [], set()
            for k, v in self.map.items():
                params = [r() for r in v[4]]
                if not any(p.requires_grad for p in params):
                    continue
                keys.add(k)
                for (idx, mode, row0, col0) in v[6]:
                    parts.append(self._part(_lib.RepackPart, v[0], params[idx], mode, row0, col0))
            if not parts:
                return False
            arr = (_lib.RepackPart * len(parts))(*parts)
            nbytes = lib.ns2_weights_repack_table_bytes(len(parts))
            dev = next(r() for v in self.map.values() for r in v[4]).device
            table = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            total = ctypes.c_int64(0)
            check(lib.ns2_weights_repack_build(arr, len(parts), table.data_ptr(), nbytes, ctypes.byref(total), _s()), "ns2_weights_repack_build")
            tiled = [self.map[k][0].handle.value for k in keys if getattr(self.map[k][0], "tiled", False)]
            self._table = (sig, table, len(parts), total.value, keys, (ctypes.c_void_p * len(tiled))(*tiled) if tiled else None, len(tiled))
        _, table, n, total, _, tiled, nt = self._table
        check(lib.ns2_weights_repack(table.data_ptr(), n, total, _s()), "ns2_weights_repack")
        if nt:                                # the lean kernels' tile images follow their packs (one small launch per tiled weight)
            check(lib.ns2_weights_retile(tiled, nt, _s()), "ns2_weights_retile")
        return True

    def begin_pass(self):
        self.pass_id += 1
        if _WEIGHTS_FROZEN:                  # `with training.weights_unchanged():` -- the packs of the last pass are these weights' (stamp them fresh)
            for k, v in list(self.map.items()):
                self.map[k] = (v[0], v[1], v[2], v[3], v[4], self.pass_id, v[6])
            return
        if self.pass_id > 1 and self._repack_all():
            for k in self._table[4]:                                         # what the launch re-packed is fresh for this pass
                v = self.map[k]                                              # (a frozen weight is not in the table: its version rule stands)
                self.map[k] = (v[0], tuple((p.data_ptr(), p._version) for p in (r() for r in v[4])), v[2], v[3], v[4], self.pass_id, v[6])

    def _purge(self):
        dead = [k for k, v in self.map.items() if any(r() is None for r in v[4])]
        for k in dead:
            del self.map[k]
        if dead:
            self._table = None

    def get(self, key, params, make_src, parts=None, **pack_kw):
        """`key` carries id()s of `params`: an entry is only a hit while those very objects are alive (weak references), so a
        recycled id can never return another model's weights.  `parts`: [(index into params, mode, row0, col0)] -- where the pack's
        values live in the parameters' own storage (`_part`), which lets begin_pass refresh every pack of the model in one launch;
        `make_src` builds the same matrix as a tensor for the first pack (and for entries without `parts`)."""
        sig = tuple((p.data_ptr(), p._version) for p in params)
        hit = self.map.get(key)
        if hit is not None and any(r() is not p for r, p in zip(hit[4], params)):
            hit = None
        refs = tuple(weakref.ref(p) for p in params)
        if hit is None:
            self._purge()                    # a miss is rare (first step of a model): drop the packs of models that no longer exist
            self._table = None
        if hit is not None and hit[1] == sig and (hit[5] == self.pass_id or not any(p.requires_grad for p in params)):
            return hit[0]
        src = make_src()
        src = src if isinstance(src, tuple) else (src, None)
        w, extra = (t.detach().float().contiguous() if t is not None else None for t in src)
        if hit is not None and hit[2] == (tuple(w.shape), None if extra is None else tuple(extra.shape)):
            check(_lib.load().ns2_weight_update(hit[0].handle, w.data_ptr(), _p(extra), _s()), "ns2_weight_update")
            self.map[key] = (hit[0], sig, hit[2], (w, extra), refs, self.pass_id, parts)  # keep the sources alive until the stream has consumed them
            return hit[0]
        pw = ops.PackedWeight(w, extra1x1=extra, precision=self.precision, **pack_kw)
        if self.precision == 4 and w.ndim == 2 and extra is None and not pack_kw.get("geglu") and w.shape[1] >= 96 and self.lean:
            pw.tile_linear()                 # round 6: the lean mixed linear kernel (csrc/gemm3_kernel.h) for the forward and dgrad products
            pw.tiled = True                  # on whole 256-row tiles -- bit-identical to gemm2_kernel<2, *>, 10-15 % faster
        self._table = None                   # (a hit whose source changed shape gets a NEW pack: the re-pack table still names the old one's storage)
        self.map[key] = (pw, sig, (tuple(w.shape), None if extra is None else tuple(extra.shape)), (w, extra), refs, self.pass_id, parts)
        return pw


class HipBackend:
    """every method = launches of libns2hip on the current stream; torch provides the buffers.

    `precision` = the arithmetic of the GEMMs (forward, dgrad, wgrad) and the format of their operand planes: 3 = bf16 hi / lo
    planes, three bf16 products ("exact"); 4 = FMT_H8 lines, one IEEE-half product + both correction terms on the fp8 MFMA
    ("mixed": 2 MFMA units instead of 3; needs the loss scaling of `_Scale` below).  The attention products and their operands
    (q, k, v, dO and the per-utterance transposes) are bf16 x3 at both precisions (`attn=True` arguments)."""
    name = "hip"

    def __init__(self, precision=3):
        assert precision in (3, 4)
        self.lib = _lib.load()
        self.prec = precision
        self.packs = _PackedCache(precision)

    # ---- weights
    def pack(self, key, params, make_src, parts=None):
        return self.packs.get(key, params, make_src, parts=parts)

    # ---- forward pieces
    def split(self, x, C=None):
        x = x if x.is_contiguous() else x.contiguous()
        return ops.split(x, precision=self.prec)

    @staticmethod
    def _rows(t, d):
        """[M, d] fp32 with row stride exactly d (kernels that take no stride for this operand)"""
        return t if (t.stride(1) == 1 and t.stride(0) == d) else t.contiguous()

    def rmsnorm(self, x, seq_len, gamma=None, cond=None):
        return ops.rmsnorm(x, seq_len=seq_len, gamma=gamma, cond=cond, precision=self.prec)

    def rmsnorm_f32(self, x, gamma):
        """RMSNorm(x) * gamma as fp32 [M, d] (the resampler's final norm, NS2:579: its output is a tensor of the graph, not an operand)"""
        return ops.rmsnorm(x, seq_len=0, gamma=gamma, want_f32=True, precision=3)[1]

    def gemm_f32(self, pw, a, bias=None, resid=None, taps=0, dil=1, seq_len=0, pad_left=-1):
        """-> fp32 [M, ldo] with ldo = round_up(N, 32); columns >= N are NOT written"""
        M, N = a.rows, pw.rows
        ldo = round_up(N, 32)
        out = torch.empty(M, ldo, dtype=torch.float32, device=a.device)
        ldr = resid.stride(0) if resid is not None else 0
        check(self.lib.ns2_linear_f32(pw.handle, a.hi, a.lo, a.ld, M, taps, dil, seq_len, _p(bias), _p(resid), ldr, out.data_ptr(), ldo,
                                      pad_left, 0, self.prec, _s()), "ns2_linear_f32")
        return out

    def gemm_split(self, pw, a, bias=None, taps=0, dil=1, seq_len=0, attn=False):
        """-> operand planes; attn=True: attention operands (q | k | v), bf16 hi / lo lines whatever the GEMM arithmetic"""
        if attn and self.prec != 3:
            M, ldo = a.rows, round_up(pw.rows, 32)
            out = ops.empty_planes(M, ldo, a.device)
            check(self.lib.ns2_linear_split_as(pw.handle, a.hi, a.lo, a.ld, M, taps, dil, seq_len, _p(bias), out.hi, out.lo, ldo, -1, 0,
                                               self.prec, 3, _s()), "ns2_linear_split_as")
            return out
        return ops.linear_split(pw, a, bias=bias, conv_taps=taps, dilation=dil, seq_len=seq_len, precision=self.prec)

    def film_gate_fwd(self, h, film, seq_len, d):
        out = torch.empty(h.shape[0], d, dtype=torch.float32, device=h.device)
        check(self.lib.ns2_film_gate_fwd(h.data_ptr(), h.stride(0), film.data_ptr(), film.stride(0), seq_len, h.shape[0], d, out.data_ptr(), d,
                                         _s()), "ns2_film_gate_fwd")
        return out

    def geglu_fwd(self, pre, f):
        M = pre.shape[0]
        out = ops._out_planes(M, round_up(f, 32), pre.device, self.prec)
        check(self.lib.ns2_geglu_fwd(pre.data_ptr(), pre.stride(0), M, f, out.hi, out.lo, out.ld, self.prec, _s()), "ns2_geglu_fwd")
        return out

    def attention(self, q, q_col0, k, k_col0, vt, B, H, Nq, Nk):
        """bf16 x3 products on bf16 operands; the output o (the out-projection's operand) in the GEMM format"""
        o = ops._out_planes(B * Nq, H * 64, q.device, self.prec)
        lse = torch.empty(B, H, Nq, dtype=torch.float32, device=q.device)
        check(self.lib.ns2_attention_lse(q.hi, q.lo, q.ld, q_col0, k.hi, k.lo, k.ld, k_col0, vt.ptr(), vt.ptr() + 64, vt.ld, o.hi, o.lo, H * 64,
                                         B, H, Nq, Nk, 0.125, lse.data_ptr(), 3, self.prec, _s()), "ns2_attention_lse")
        return o, lse

    # ---- rows = batch entries: the conditioning Linears (weight-streaming kernel of the inference path, fp32)
    def skinny(self, x, wt, bias=None):
        """x [R, K] @ wt [K, J] (+ bias) in fp32 (ns2_skinny_linear: deterministic split-K)"""
        return ops.skinny_linear(x if x.is_contiguous() else x.contiguous(), wt if wt.is_contiguous() else wt.contiguous(), bias)

    def transpose_f32(self, x):
        x = x if x.is_contiguous() else x.contiguous()
        R, C = x.shape
        out = torch.empty(C, R, dtype=torch.float32, device=x.device)
        check(self.lib.ns2_transpose_f32(x.data_ptr(), 1, R, C, out.data_ptr(), _s()), "ns2_transpose_f32")
        return out

    def colsum_rows(self, x):
        """sum over the rows of a small [R, J] matrix in a fixed order (bias gradients of the conditioning Linears)"""
        x = x if x.is_contiguous() else x.contiguous()
        out = torch.empty(x.shape[1], dtype=torch.float32, device=x.device)
        check(self.lib.ns2_reduce_slices(x.data_ptr(), 1, x.shape[0], x.shape[1], out.data_ptr(), 0, _s()), "ns2_reduce_slices")
        return out

    # ---- backward pieces
    def grad_prep(self, x, C, want_row=False, want_t=False, want_colsum=False, seq_len=0, per_batch=False, t_rows=None, attn=False):
        """x fp32 [M, >= C] -> (row planes [M, round_up(C, 32)], transposed planes, column sums [C]); attn=True: operands of the
        attention backward (bf16 hi / lo), else GEMM operands in the backend's format"""
        M, dev = x.shape[0], x.device
        if not (want_row or want_t or want_colsum):
            return None, None, None
        prec = 3 if attn else self.prec
        row = ops._out_planes(M, round_up(C, 32), dev, prec) if want_row else None
        tp, ld_t = None, 0
        if want_t:
            ld_t = round_up(seq_len if per_batch else M, 32)
            t_rows = t_rows or C
            tp = TPlanes((M // seq_len) * t_rows if per_batch else t_rows, ld_t, dev, prec)
        part = None
        if want_colsum:
            S = self.lib.ns2_grad_prep_slices(M, ld_t)
            part = torch.empty(S, C, dtype=torch.float32, device=dev)
        check(self.lib.ns2_grad_prep(x.data_ptr(), x.stride(0), M, C, seq_len, 0, row.hi if row else None, row.lo if row else None,
                                     row.ld if row else 0, tp.ptr() if tp else None, tp.ptr() + 64 if tp else None, ld_t, t_rows or 0,
                                     int(per_batch and want_t), _p(part), prec, _s()), "ns2_grad_prep")
        cs = None
        if want_colsum:
            cs = torch.empty(C, dtype=torch.float32, device=dev)
            check(self.lib.ns2_reduce_slices(part.data_ptr(), 1, part.shape[0], C, cs.data_ptr(), 0, _s()), "ns2_reduce_slices")
        return row, tp, cs

    def transpose(self, p, col0, C, seq_len, shifts=(0,), per_batch=False, pad_rows=256):
        """planes [M, ld] columns [col0, col0 + C) -> transposed planes; one row block of Cp = round_up(C, 32) rows per shift
        (the taps of a conv's weight gradient), rows zero-padded to what a W operand of ns2_wgrad may read"""
        M = p.rows
        Cp = round_up(C, 32)
        prec = p.precision                       # the transposed planes keep the format of the planes they come from
        assert prec in (3, 4)
        if per_batch:
            B = M // seq_len
            tp = TPlanes(B * C, round_up(seq_len, 32), p.device, prec)
            check(self.lib.ns2_planes_transpose(p.hi, p.lo, p.ld, col0, M, C, seq_len, 0, tp.ptr(), tp.ptr() + 64, tp.ld, C, 1, prec, _s()),
                  "ns2_planes_transpose")
            return tp
        T = len(shifts)
        # a W operand is read in whole 256-row tiles from wherever a wgrad starts (row 0 for all taps, row 2 Cp for res_conv)
        rows = max((T - 1) * Cp + round_up(Cp, pad_rows), round_up(T * Cp, pad_rows))
        tp = TPlanes(rows, round_up(M, 32), p.device, prec)
        for t, sh in enumerate(shifts):
            t_rows = Cp if t < T - 1 else rows - (T - 1) * Cp
            check(self.lib.ns2_planes_transpose(p.hi, p.lo, p.ld, col0, M, C, seq_len, sh, tp.ptr(t * Cp), tp.ptr(t * Cp) + 64, tp.ld, t_rows, 0,
                                                prec, _s()), "ns2_planes_transpose")
        return tp

    def wgrad(self, dyt, xt, R, T, K, row_off=0):
        """dW [R, K, T] = dY^T X_t ; xt: T blocks of Kp = round_up(K, 32) rows starting at row_off"""
        Kp = round_up(K, 32)
        assert dyt.precision == xt.precision == self.prec, "wgrad operands must be in the backend's GEMM format"
        nbytes = self.lib.ns2_wgrad_workspace_bytes(R, T * Kp, dyt.ld)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dyt.buf.device)
        dw = torch.empty(R, K, T, dtype=torch.float32, device=dyt.buf.device)
        check(self.lib.ns2_wgrad(dyt.ptr(), dyt.ptr() + 64, xt.ptr(row_off), xt.ptr(row_off) + 64, dyt.ld, R, T, Kp, K, dw.data_ptr(), ws.data_ptr(),
                                 nbytes, self.prec, _s()), "ns2_wgrad")
        return dw

    def wgrad_rows_ok(self, R, T, K, seq_len, M):
        """does ns2_wgrad_rows form this gradient from the token-major planes (else: transposed copies + ns2_wgrad)"""
        return (T == 1 or seq_len >= 32) and bool(self.lib.ns2_wgrad_rows_preferred(R, T * round_up(K, 32), M))

    def wgrad_rows(self, dy, x, R, T, K, dil=1, seq_len=0):
        """dW [R, K, T] = sum_m dY[m, r] X[m - (T - 1 - t) dil, k] from the ROW planes dy [M, >= R] and x [M, >= round_up(K, 32)] themselves
        (gemm2.hip TR: LDS transpose reads; the shifts of a conv's taps are row offsets of the loads)"""
        Kp = round_up(K, 32)
        assert dy.precision == x.precision == self.prec and dy.rows == x.rows, "wgrad operands must be planes of the same tokens in the backend's GEMM format"
        M = dy.rows
        nbytes = self.lib.ns2_wgrad_workspace_bytes(R, T * Kp, round_up(M, 32))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dy.device)
        dw = torch.empty(R, K, T, dtype=torch.float32, device=dy.device)
        check(self.lib.ns2_wgrad_rows(dy.hi, dy.lo, dy.ld, x.hi, x.lo, x.ld, M, R, T, Kp, K, dil, seq_len if T > 1 else 0, dw.data_ptr(), ws.data_ptr(),
                                      nbytes, self.prec, _s()), "ns2_wgrad_rows")
        return dw

    def film_gate_bwd(self, dg, h, film, B, seq_len, d):
        S = self.lib.ns2_film_gate_slices(seq_len)
        dh = torch.empty(h.shape[0], d, dtype=torch.float32, device=h.device)
        part = torch.empty(B * S, 2 * d, dtype=torch.float32, device=h.device)
        check(self.lib.ns2_film_gate_bwd(dg.data_ptr(), dg.stride(0), h.data_ptr(), h.stride(0), film.data_ptr(), film.stride(0), B, seq_len, d,
                                         dh.data_ptr(), d, part.data_ptr(), _s()), "ns2_film_gate_bwd")
        dfilm = torch.empty(B, 2 * d, dtype=torch.float32, device=h.device)
        check(self.lib.ns2_reduce_slices(part.data_ptr(), B, S, 2 * d, dfilm.data_ptr(), 0, _s()), "ns2_reduce_slices")
        return dh, dfilm

    def geglu_bwd(self, dh, pre, f):
        M = pre.shape[0]
        dpre = torch.empty(M, round_up(2 * f, 32), dtype=torch.float32, device=pre.device)
        check(self.lib.ns2_geglu_bwd(dh.data_ptr(), dh.stride(0), pre.data_ptr(), pre.stride(0), M, f, dpre.data_ptr(), dpre.stride(0), _s()),
              "ns2_geglu_bwd")
        return dpre

    def rmsnorm_bwd(self, x, dy, B, seq_len, d, gamma=None, cond=None, dx_add=None):
        """-> (dx [M, d] = dx_add + dL/dx, dcond [B, 2 d] or None, dgamma [d] or None)"""
        S = self.lib.ns2_rmsnorm_bwd_slices(seq_len)
        dev = x.device
        dx = torch.empty(B * seq_len, d, dtype=torch.float32, device=dev)
        cpart = torch.empty(B * S, 2 * d, dtype=torch.float32, device=dev) if cond is not None else None
        gpart = torch.empty(B * S, d, dtype=torch.float32, device=dev) if gamma is not None else None
        if dx_add is not None:
            dx_add = self._rows(dx_add[:, :d], d)          # the kernel reads dx_add with dx's row stride (= d)
        check(self.lib.ns2_rmsnorm_bwd(x.data_ptr(), x.stride(0), dy.data_ptr(), dy.stride(0), _p(gamma), _p(cond),
                                       cond.stride(0) if cond is not None else 0, B, seq_len, d, _p(dx_add), dx.data_ptr(), d, _p(cpart),
                                       _p(gpart), _s()), "ns2_rmsnorm_bwd")
        dcond = dgamma = None
        if cond is not None:
            dcond = torch.empty(B, 2 * d, dtype=torch.float32, device=dev)
            check(self.lib.ns2_reduce_slices(cpart.data_ptr(), B, S, 2 * d, dcond.data_ptr(), 0, _s()), "ns2_reduce_slices")
        if gamma is not None:
            dgamma = torch.empty(d, dtype=torch.float32, device=dev)
            check(self.lib.ns2_reduce_slices(gpart.data_ptr(), 1, B * S, d, dgamma.data_ptr(), 0, _s()), "ns2_reduce_slices")
        return dx, dcond, dgamma

    def attention_delta(self, do, o, B, H, Nq):
        delta = torch.empty(B, H, Nq, dtype=torch.float32, device=do.device)
        check(self.lib.ns2_attention_delta(do.data_ptr(), do.stride(0), o.hi, o.lo, o.ld, B, H, Nq, delta.data_ptr(), o.precision, _s()),
              "ns2_attention_delta")
        return delta

    def new_planes(self, M, C):
        """uninitialised operand planes [M, round_up(C, 32)] in the backend's GEMM format (a kernel is about to fill every column)"""
        return ops._out_planes(M, round_up(C, 32), torch.device("cuda", torch.cuda.current_device()), self.prec)

    def attention_bwd(self, q, q_col0, k, k_col0, v, v_col0, do_row, lse, delta, B, H, Nq, Nk, dq=None, dkv=None, planes=None):
        """dq: (fp32 tensor [B*Nq, ld], col0) or None; dkv: (tensor [B*Nk, ld], k col0, v col0) or None;
        planes: (operand planes [B*N, ld], dq col0, dk col0, dv col0) -- self attention: the three gradients leave the kernels as the
        operand of the q | k | v projection's dgrad / wgrad GEMMs instead of fp32 + a conversion pass"""
        a = _lib.AttnBwdArgs()
        a.q_hi, a.q_lo, a.ldq, a.q_col0 = q.hi, q.lo, q.ld, q_col0
        a.k_hi, a.k_lo, a.ldk, a.k_col0 = k.hi, k.lo, k.ld, k_col0
        a.v_hi, a.v_lo, a.ldv, a.v_col0 = v.hi, v.lo, v.ld, v_col0
        a.do_hi, a.do_lo, a.lddo = do_row.hi, do_row.lo, do_row.ld
        a.lse, a.delta = lse.data_ptr(), delta.data_ptr()
        if dq is not None:
            a.dq, a.lddq, a.dq_col0 = dq[0].data_ptr(), dq[0].stride(0), dq[1]
        if dkv is not None:
            a.dk, a.lddk, a.dk_col0 = dkv[0].data_ptr(), dkv[0].stride(0), dkv[1]
            a.dv, a.lddv, a.dv_col0 = dkv[0].data_ptr(), dkv[0].stride(0), dkv[2]
        if planes is not None:
            gp, a.dq_col0, a.dk_col0, a.dv_col0 = planes
            assert gp.precision == self.prec
            a.gp_hi, a.gp_lo, a.gp_ld, a.gp_precision, a.gp_q, a.gp_kv = gp.hi, gp.lo, gp.ld, self.prec, 1, 1
        a.B, a.H, a.Nq, a.Nk, a.scale = B, H, Nq, Nk, 0.125
        check(self.lib.ns2_attention_bwd(ctypes.byref(a), _s()), "ns2_attention_bwd")


_BACKEND = None            # a substitute installed by the tests (tests/emu_backend.py)
_HIP = {}                  # (device index, precision) -> HipBackend, each with its own packed-weight cache and re-pack table (ADVICE r5:
                           # models on two GPUs of one process must not share a device table)
_CUR_DEV = None            # device index of the graph being built (None: the current device)
_CUR_PREC = 3              # the arithmetic of the graph being BUILT (model_forward_train sets it around its Functions' forwards)
_CUR_SCALE = None          # ... and its loss scale (None: exact arithmetic)
TRAIN_PRECISIONS = {"exact": 3, "mixed": 4}


def backend():
    """the backend the Function whose forward is running should use (it keeps it in ctx for its backward)"""
    if _BACKEND is not None:
        return _BACKEND
    dev = _CUR_DEV if _CUR_DEV is not None else (torch.cuda.current_device() if torch.cuda.is_available() else -1)
    key = (dev, _CUR_PREC)
    if key not in _HIP:
        _HIP[key] = HipBackend(_CUR_PREC)
    return _HIP[key]


_WEIGHTS_FROZEN = False


class weights_unchanged:
    """`with training.weights_unchanged():` around training passes that follow another pass WITHOUT an optimizer step in between -- the later
    micro-batches of gradient accumulation (NS2:1877-1885), several losses on one set of weights: the per-pass refresh of every packed weight
    (one launch reading all parameters: 1.3 ms at d512 / L12; needed by default because fused optimizers change parameters without bumping the
    version counters a cache could key on) is skipped and the packs of the previous pass are used as they are.  The caller's promise;
    weights changed inside the block the way a fused optimizer changes them (no version bump) are NOT seen until the first pass outside it."""

    def __enter__(self):
        global _WEIGHTS_FROZEN
        self.prev, _WEIGHTS_FROZEN = _WEIGHTS_FROZEN, True

    def __exit__(self, *exc):
        global _WEIGHTS_FROZEN
        _WEIGHTS_FROZEN = self.prev


def set_backend(b):
    """tests: install a substitute backend (tests/emu_backend.py); returns the previous one"""
    global _BACKEND
    prev, _BACKEND = _BACKEND, b
    return prev


# =============================================================================================== loss scaling of the mixed arithmetic
class _Scale:
    """Loss scale of ONE forward / backward pass of the mixed arithmetic.  IEEE half (and its e5m2 companions) stops at 65504 and
    thins out below 6e-5; the gradient of a mean-reduced loss over 1.7e7 elements is ~1e-7.  Every gradient is linear in the
    gradient that enters the graph, so the pass runs on gradients multiplied by a power of two `s` (exact in fp32) -- what the
    reference's accelerate / fp16 training does with its GradScaler (NS2:1710-1711, 1723-1726).  `s` is chosen where the gradient
    enters (`_ScaleIn` at the model output) from its largest magnitude, on the device, without a synchronisation: the scaled maximum
    becomes 2^5 ... 2^6, which leaves 2^10 of head-room before 65504 for growth inside the graph and ~2^19 below for small values.
    Token-sized gradients stay scaled between the Functions; whatever leaves the scaled domain -- parameter gradients, FiLM /
    adaptive-norm conditioning gradients, x.grad -- is multiplied by 1 / s (`unscale`).  Values that still leave the half range are
    COUNTED by the converting kernels: `overflowed()` (below) compares the device's range counters after the pass with their values
    when the pass began and tells a training loop to skip / repeat the step -- the counterpart of GradScaler's inf check
    (`model._last_loss_scale.overflowed()` after `loss.backward()`)."""
    TARGET = 64.0

    def __init__(self):
        self.s = None
        self.inv = None
        self._before = self._peek()          # stream-ordered: the counters as they stand when the pass is enqueued

    @staticmethod
    def _peek():
        """the five range counters of the device (ns2_saturation_peek_async: forward GEMMs / attention / pointwise;
        ns2_saturation_peek_train_async: the training kernels), copied to pinned memory on the current stream -- no synchronisation"""
        if not torch.cuda.is_available() or torch.cuda.is_current_stream_capturing():
            return None                      # (under capture GraphedTrainStep peeks around the replay instead)
        try:
            lib = _lib.load()
        except Exception:
            return None
        buf = torch.zeros(5, dtype=torch.int32).pin_memory()
        s = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.ns2_saturation_peek_async(buf.data_ptr(), s), "ns2_saturation_peek_async")
        _lib.check(lib.ns2_saturation_peek_train_async(buf.data_ptr() + 16, s), "ns2_saturation_peek_train_async")
        ev = torch.cuda.Event()
        ev.record()
        return buf, ev

    def overflowed(self) -> bool:
        """did a conversion of this pass (forward activations, scaled gradients) leave the IEEE-half range?  Call after `backward()`;
        waits for the pass (one event synchronisation).  True: the gradients of this step are clamped somewhere -- skip the optimizer
        step (and train on with `train_precision="exact"` if it keeps happening: bf16 planes have the fp32 range)."""
        if self._before is None:
            return False
        after, ev = self._peek()
        ev.synchronize()
        return bool((after != self._before[0]).any().item())

    def choose(self, g):
        amax = g.detach().abs().amax().clamp_min(1e-30).float()
        self.s = torch.exp2(torch.floor(torch.log2(self.TARGET / amax))).clamp(2.0 ** -60, 2.0 ** 60)
        self.inv = 1.0 / self.s
        return self.s

    def unscale(self, t):
        return t if t is None else t * self.inv


class _ScaleIn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, sc):
        ctx.sc = sc
        return y.view_as(y)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.sc.choose(g), None


class _ScaleOut(torch.autograd.Function):
    """a tensor entering the scaled domain from outside (x, `null_prompt_tokens`, the resampler's latents ...): its gradient leaves it"""

    @staticmethod
    def forward(ctx, t, sc):
        ctx.sc = sc
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        return ctx.sc.unscale(g), None


def _enter(t):
    return t if _CUR_SCALE is None else _ScaleOut.apply(t, _CUR_SCALE)


def _un(ctx, *ts):
    """parameter / conditioning gradients of a Function leave the scaled domain (one multi-tensor launch per Function: the per-tensor
    multiplications were 325 launches of a d512 / L12 step)"""
    sc = ctx.sc
    if sc is None:
        return ts
    live = [t for t in ts if t is not None]
    if not live:
        return ts
    done = iter(torch._foreach_mul(live, sc.inv))
    return tuple(None if t is None else next(done) for t in ts)


# =============================================================================================== weight sources of the two packs
def _taps(w):
    return w.shape[2] if w.ndim == 3 else 0


def _fwd_pack(bk, w):
    return bk.pack(("f", id(w)), (w,), lambda: w, parts=[(0, "n", 0, 0)])


def _bwd_pack(bk, w):
    """the dgrad weight: dX = dY W  ->  rows = input channels, columns = output channels, taps flipped (the gradient of a causal
    conv reads dY[n + (k - 1 - t) dil], i.e. a conv with pad_left = 0 whose tap t' is the forward's tap k - 1 - t')"""
    if w.ndim == 3:
        return bk.pack(("b", id(w)), (w,), lambda: w.detach().permute(1, 0, 2).flip(-1), parts=[(0, "tf", 0, 0)])
    return bk.pack(("b", id(w)), (w,), lambda: w.detach().t(), parts=[(0, "t", 0, 0)])


def _shifts(taps, dil):
    return tuple((taps - 1 - t) * dil for t in range(taps)) if taps else (0,)


def _dw(dw, w):
    """[R, K, T] from ns2_wgrad -> the parameter's own shape"""
    return dw if w.ndim == 3 else dw[:, :, 0]


def _grads(bk, dy, C, xp, K, taps=0, dil=1, seq_len=0, need_row=True, need_w=True, need_b=False):
    """what the backward of y = conv_or_linear(x) (+ b) needs from dy fp32 [M, >= C]: its row planes (the dgrad GEMM's operand), the
    weight gradient dW [C, K, T] against xp = the operand planes the forward GEMM read, the bias gradient.  The weight gradient comes
    from the row planes themselves where the kernel can (ns2_wgrad_rows: LDS transpose reads), else through transposed copies
    (narrow gradients: the dim = 128 model).  -> (dy_row or None, dW or None, db or None)"""
    T = max(taps, 1)
    if need_w and bk.wgrad_rows_ok(C, T, K, seq_len, dy.shape[0]):
        row, _, db = bk.grad_prep(dy, C, want_row=True, want_colsum=need_b)
        return row, bk.wgrad_rows(row, xp, C, T, K, dil, seq_len), db
    row, dy_t, db = bk.grad_prep(dy, C, want_row=need_row, want_t=need_w, want_colsum=need_b)
    dw = None
    if need_w:
        xt = bk.transpose(xp, 0, round_up(K, 32), seq_len if taps else 0, _shifts(taps, dil))
        dw = bk.wgrad(dy_t, xt, C, T, K)
    return row, dw, db


# =============================================================================================== Functions
class GemmFn(torch.autograd.Function):
    """y = conv_or_linear(x) + b (+ resid);  x [M, Cin] fp32, w [Cout, Cin(, k)], causal with dilation `dil` inside utterances of
    `seq_len` tokens (NS2:583-595)"""

    @staticmethod
    def forward(ctx, x, w, b, resid, seq_len, dil):
        bk = ctx.bk = backend()
        ctx.sc = _CUR_SCALE
        taps = _taps(w)
        xp = bk.split(x)
        y = bk.gemm_f32(_fwd_pack(bk, w), xp, bias=b, resid=resid, taps=taps, dil=dil, seq_len=seq_len if taps else 0)
        ctx.save_for_backward(w, b)
        ctx.xp, ctx.cfg = xp, (taps, dil, seq_len, x.shape[1], resid is not None)
        return y[:, :w.shape[0]]

    @staticmethod
    def backward(ctx, dy):
        bk = ctx.bk
        w, b = ctx.saved_tensors
        taps, dil, seq_len, cin, has_resid = ctx.cfg
        cout = w.shape[0]
        dy = dy if dy.stride(1) == 1 else dy.contiguous()
        # (frozen weights / a gradient wanted for the input only -- guidance: no wgrad GEMM, no transposes, no workspace: ADVICE r4)
        need_w, need_b = ctx.needs_input_grad[1], b is not None and ctx.needs_input_grad[2]
        dy_row, dw, db = _grads(bk, dy, cout, ctx.xp, cin, taps, dil, seq_len, need_row=ctx.needs_input_grad[0], need_w=need_w, need_b=need_b)
        dw = _dw(dw, w) if need_w else None
        dx = None
        if ctx.needs_input_grad[0]:
            dx = bk.gemm_f32(_bwd_pack(bk, w), dy_row, taps=taps, dil=dil, seq_len=seq_len if taps else 0, pad_left=0 if taps else -1)[:, :cin]
        return dx, *_un(ctx, dw, db), (dy if has_resid else None), None, None


class WavenetBlockFn(torch.autograd.Function):
    """WavenetResBlock (NS2:597-642) without its skip conv: out = tanh(z) sigmoid(z) + res_conv(u), z = conv_dil(u) gamma + beta"""

    @staticmethod
    def forward(ctx, u, film, wc, bc, wr, br, seq_len, dil):
        bk = ctx.bk = backend()
        ctx.sc = _CUR_SCALE
        d = u.shape[1]
        up = bk.split(u)
        hc = bk.gemm_f32(_fwd_pack(bk, wc), up, bias=bc, taps=3, dil=dil, seq_len=seq_len)
        g = bk.film_gate_fwd(hc, film, seq_len, d)
        out = bk.gemm_f32(_fwd_pack(bk, wr), up, bias=br, resid=g, taps=1, dil=1, seq_len=seq_len)
        ctx.save_for_backward(film, wc, wr, hc)
        ctx.up, ctx.cfg = up, (seq_len, dil, d)
        return out[:, :d]

    @staticmethod
    def backward(ctx, dout):
        bk = ctx.bk
        film, wc, wr, hc = ctx.saved_tensors
        seq_len, dil, d = ctx.cfg
        B = dout.shape[0] // seq_len
        dout = dout if dout.stride(1) == 1 else dout.contiguous()
        ng = ctx.needs_input_grad                                           # (u, film, wc, bc, wr, br, -, -)
        if bk.wgrad_rows_ok(d, 3, d, seq_len, dout.shape[0]) and bk.wgrad_rows_ok(d, 1, d, seq_len, dout.shape[0]):
            dout_row, dwr, dbr = _grads(bk, dout, d, ctx.up, d, 1, 1, seq_len, need_w=ng[4], need_b=ng[5])
            dhc, dfilm = bk.film_gate_bwd(dout, hc, film, B, seq_len, d)
            dhc_row, dwc, dbc = _grads(bk, dhc, d, ctx.up, d, 3, dil, seq_len, need_w=ng[2], need_b=ng[3])
        else:                                                               # narrow model: transposed copies, one set for both gradients
            dout_row, dout_t, dbr = bk.grad_prep(dout, d, want_row=True, want_t=ng[4], want_colsum=ng[5])
            dhc, dfilm = bk.film_gate_bwd(dout, hc, film, B, seq_len, d)
            dhc_row, dhc_t, dbc = bk.grad_prep(dhc, d, want_row=True, want_t=ng[2], want_colsum=ng[3])
            dwc = dwr = None
            if ng[2] or ng[4]:
                xt = bk.transpose(ctx.up, 0, d, seq_len, _shifts(3, dil))       # taps 0, 1, 2: shifts 2 dil, dil, 0
                if ng[2]:
                    dwc = bk.wgrad(dhc_t, xt, d, 3, d)
                if ng[4]:
                    dwr = bk.wgrad(dout_t, xt, d, 1, d, row_off=2 * round_up(d, 32))    # res_conv reads the unshifted block (tap 2)
        du = bk.gemm_f32(_bwd_pack(bk, wc), dhc_row, taps=3, dil=dil, seq_len=seq_len, pad_left=0)
        du = bk.gemm_f32(_bwd_pack(bk, wr), dout_row, resid=du, taps=1, dil=1, seq_len=seq_len, pad_left=0)
        return du[:, :d], *_un(ctx, dfilm, dwc, dbc, dwr, dbr), None, None


class AttnFn(torch.autograd.Function):
    """h + to_out(attention(q, k, v)) with q = to_q(norm(h)), k, v = to_kv(context or norm(h)); norm = RMSNorm with the adaptive
    (gamma, beta) of `film` (Model, NS2:727-746) -- heads of 64, non-causal, no mask (ATT:77-155).  `film = None`: no norm in front
    (the PerceiverResampler's attention, NS2:574: its context = cat(latents, prompt) is formed by the caller, so the part of the
    context gradient that belongs to the latents reaches them through torch's cat)"""

    @staticmethod
    def forward(ctx, h, film, ctxt, wq, wkv, wout, seq_len, heads, ctx_len):
        bk = ctx.bk = backend()
        ctx.sc = _CUR_SCALE
        M, d = h.shape
        B, a = M // seq_len, heads * 64
        xn = bk.rmsnorm(h, seq_len, cond=film) if film is not None else bk.split(h)
        if ctxt is None:                                                    # self attention: one GEMM for q | k | v
            wqkv = bk.pack(("qkv", id(wq), id(wkv)), (wq, wkv), lambda: torch.cat((wq.detach(), wkv.detach()), 0),
                           parts=[(0, "n", 0, 0), (1, "n", a, 0)])
            qkv = bk.gemm_split(wqkv, xn, attn=True)
            q, k, v, qc, kc, vc, Nk, cp = qkv, qkv, qkv, 0, a, 2 * a, seq_len, None
        else:
            q = bk.gemm_split(_fwd_pack(bk, wq), xn, attn=True)
            cp = bk.split(ctxt)
            kv = bk.gemm_split(_fwd_pack(bk, wkv), cp, attn=True)
            k, v, qc, kc, vc, Nk = kv, kv, 0, 0, a, ctx_len
        vt = bk.transpose(v, vc, a, Nk, per_batch=True)
        o, lse = bk.attention(q, qc, k, kc, vt, B, heads, seq_len, Nk)
        y = bk.gemm_f32(_fwd_pack(bk, wout), o, resid=h)
        ctx.save_for_backward(h, film, wq, wkv, wout, lse)
        ctx.pl, ctx.cfg = (xn, q, k, v, o, cp), (seq_len, heads, Nk, qc, kc, vc, ctxt is not None)
        return y[:, :d]

    @staticmethod
    def backward(ctx, dy):
        bk = ctx.bk
        h, film, wq, wkv, wout, lse = ctx.saved_tensors
        xn, q, k, v, o, cp = ctx.pl
        seq_len, heads, Nk, qc, kc, vc, cross = ctx.cfg
        M, d = h.shape
        B, a = M // seq_len, heads * 64
        dy = dy if dy.stride(1) == 1 else dy.contiguous()
        ng = ctx.needs_input_grad                                           # (h, film, ctxt, wq, wkv, wout, ...)
        dy_row, dwout, _ = _grads(bk, dy, d, o, a, need_w=ng[5])
        dwout = dwout[:, :, 0] if ng[5] else None
        do = bk.gemm_f32(_bwd_pack(bk, wout), dy_row)                       # [M, a]
        delta = bk.attention_delta(do, o, B, heads, seq_len)
        # (the backward kernels form K^T, Q^T and dO^T inside the CU: LDS transpose reads of the row-major tiles)
        do_row, _, _ = bk.grad_prep(do, a, want_row=True, attn=True)
        if not cross:
            need_w = ng[3] or ng[4]
            if not need_w or bk.wgrad_rows_ok(3 * a, 1, d, 0, M):
                # dq | dk | dv feed two GEMMs only (the projection's dgrad and wgrad): they leave the attention backward as operand planes
                g_row = bk.new_planes(M, 3 * a)
                bk.attention_bwd(q, qc, k, kc, v, vc, do_row, lse, delta, B, heads, seq_len, Nk, planes=(g_row, 0, a, 2 * a))
                dwqkv = bk.wgrad_rows(g_row, xn, 3 * a, 1, d) if need_w else None
            else:
                dqkv = torch.empty(M, 3 * a, dtype=torch.float32, device=h.device)
                bk.attention_bwd(q, qc, k, kc, v, vc, do_row, lse, delta, B, heads, seq_len, Nk, dq=(dqkv, 0), dkv=(dqkv, a, 2 * a))
                g_row, dwqkv, _ = _grads(bk, dqkv, 3 * a, xn, d, need_w=need_w)
            dwq = dwkv = None
            if need_w:
                dwq, dwkv = dwqkv[:a, :, 0], dwqkv[a:, :, 0]
            wqkv = bk.pack(("qkv_b", id(wq), id(wkv)), (wq, wkv), lambda: torch.cat((wq.detach(), wkv.detach()), 0).t(),
                           parts=[(0, "t", 0, 0), (1, "t", 0, a)])
            dxn = bk.gemm_f32(wqkv, g_row)
            dctx = None
        else:
            dq = torch.empty(M, a, dtype=torch.float32, device=h.device)
            dkv = torch.empty(B * Nk, 2 * a, dtype=torch.float32, device=h.device)
            bk.attention_bwd(q, qc, k, kc, v, vc, do_row, lse, delta, B, heads, seq_len, Nk, dq=(dq, 0), dkv=(dkv, 0, a))
            q_row, dwq, _ = _grads(bk, dq, a, xn, d, need_w=ng[3])
            dwq = dwq[:, :, 0] if ng[3] else None
            dxn = bk.gemm_f32(_bwd_pack(bk, wq), q_row)
            kv_row, dwkv, _ = _grads(bk, dkv, 2 * a, cp, d, need_row=ng[2], need_w=ng[4])
            dwkv = dwkv[:, :, 0] if ng[4] else None
            dctx = bk.gemm_f32(_bwd_pack(bk, wkv), kv_row)[:, :d] if ng[2] else None
        if film is None:
            dh, dfilm = dy[:, :d] + dxn[:, :d], None
        else:
            dh, dfilm, _ = bk.rmsnorm_bwd(h, dxn, B, seq_len, d, cond=film, dx_add=dy)
        dfilm, dwq, dwkv, dwout = _un(ctx, dfilm, dwq, dwkv, dwout)
        return dh, dfilm, dctx, dwq, dwkv, dwout, None, None, None


class FeedForwardFn(torch.autograd.Function):
    """h + Linear(CausalConv1d_k3(GEGLU(Linear(norm(h)))))   (NS2:1004-1025 with the adaptive RMSNorm in front, NS2:805-807).
    `film = None`: no norm; `wc = None`: no conv -- the PerceiverResampler's FeedForward (NS2:564, 575)"""

    @staticmethod
    def forward(ctx, h, film, w1, b1, wc, bc, w2, b2, seq_len):
        bk = ctx.bk = backend()
        ctx.sc = _CUR_SCALE
        M, d = h.shape
        f = w2.shape[1]
        xn = bk.rmsnorm(h, seq_len, cond=film) if film is not None else bk.split(h)
        pre = bk.gemm_f32(_fwd_pack(bk, w1), xn, bias=b1)                   # [M, 2 f]: x | gate
        hp = bk.geglu_fwd(pre, f)
        cp = bk.gemm_split(_fwd_pack(bk, wc), hp, bias=bc, taps=3, dil=1, seq_len=seq_len) if wc is not None else hp
        y = bk.gemm_f32(_fwd_pack(bk, w2), cp, bias=b2, resid=h)
        ctx.save_for_backward(h, film, w1, wc, w2, pre)
        ctx.pl, ctx.cfg = (xn, hp, cp), (seq_len, f)
        return y[:, :d]

    @staticmethod
    def backward(ctx, dy):
        bk = ctx.bk
        h, film, w1, wc, w2, pre = ctx.saved_tensors
        xn, hp, cp = ctx.pl
        seq_len, f = ctx.cfg
        M, d = h.shape
        B = M // seq_len
        dy = dy if dy.stride(1) == 1 else dy.contiguous()
        ng = ctx.needs_input_grad                                           # (h, film, w1, b1, wc, bc, w2, b2, -)
        planes_dc = wc is not None and bk.wgrad_rows_ok(f, 3, f, seq_len, M)
        dy_row, dw2, db2 = _grads(bk, dy, d, cp, f, need_w=ng[6], need_b=ng[7] or (planes_dc and ng[5]))
        dw2 = dw2[:, :, 0] if ng[6] else None
        dwc = dbc = None
        if planes_dc:
            # dc = dy W2 feeds two GEMMs only (the conv's dgrad and wgrad): it leaves the dgrad GEMM as operand planes and never exists in
            # fp32; the conv's bias gradient, colsum(dy W2), is colsum(dy) W2 -- a [d] x [d, f] product of what the FF-out bias gradient sums anyway
            dc_row = bk.gemm_split(_bwd_pack(bk, w2), dy_row)
            dwc = bk.wgrad_rows(dc_row, hp, f, 3, f, 1, seq_len) if ng[4] else None
            dbc = db2 @ w2.detach() if ng[5] else None
            db2 = db2 if ng[7] else None
            dhh = bk.gemm_f32(_bwd_pack(bk, wc), dc_row, taps=3, dil=1, seq_len=seq_len, pad_left=0)
        else:
            dc = bk.gemm_f32(_bwd_pack(bk, w2), dy_row)                     # [M, f]
            if wc is not None:
                dc_row, dwc, dbc = _grads(bk, dc, f, hp, f, 3, 1, seq_len, need_w=ng[4], need_b=ng[5])
                dhh = bk.gemm_f32(_bwd_pack(bk, wc), dc_row, taps=3, dil=1, seq_len=seq_len, pad_left=0)
            else:
                dhh = dc
        dpre = bk.geglu_bwd(dhh, pre, f)                                    # [M, 2 f]
        p_row, dw1, db1 = _grads(bk, dpre, 2 * f, xn, d, need_w=ng[2], need_b=ng[3])
        dw1 = dw1[:, :, 0] if ng[2] else None
        dxn = bk.gemm_f32(_bwd_pack(bk, w1), p_row)
        if film is None:
            dh, dfilm = dy[:, :d] + dxn[:, :d], None
        else:
            dh, dfilm, _ = bk.rmsnorm_bwd(h, dxn, B, seq_len, d, cond=film, dx_add=dy)
        return dh, *_un(ctx, dfilm, dw1, db1, dwc, dbc, dw2, db2), None


class NormLinearFn(torch.autograd.Function):
    """Linear(RMSNorm(h) * gamma), no bias: `to_pred` (NS2:781-784)"""

    @staticmethod
    def forward(ctx, h, gamma, w, seq_len):
        bk = ctx.bk = backend()
        ctx.sc = _CUR_SCALE
        xn = bk.rmsnorm(h, seq_len, gamma=gamma)
        y = bk.gemm_f32(_fwd_pack(bk, w), xn)
        ctx.save_for_backward(h, gamma, w)
        ctx.xn, ctx.seq_len = xn, seq_len
        return y[:, :w.shape[0]]

    @staticmethod
    def backward(ctx, dy):
        bk = ctx.bk
        h, gamma, w = ctx.saved_tensors
        M, d = h.shape
        dy = dy if dy.stride(1) == 1 else dy.contiguous()
        dy_row, dw, _ = _grads(bk, dy, w.shape[0], ctx.xn, d, need_w=ctx.needs_input_grad[2])
        dw = dw[:, :, 0] if ctx.needs_input_grad[2] else None
        dxn = bk.gemm_f32(_bwd_pack(bk, w), dy_row)
        dh, _, dgamma = bk.rmsnorm_bwd(h, dxn, M // ctx.seq_len, ctx.seq_len, d, gamma=gamma)
        return dh, *_un(ctx, dgamma, dw), None


class RmsNormFn(torch.autograd.Function):
    """RMSNorm(h) * gamma as a tensor of the graph: the PerceiverResampler's final norm (NS2:579, 727-746 with scale = True)"""

    @staticmethod
    def forward(ctx, h, gamma):
        ctx.save_for_backward(h, gamma)
        ctx.bk, ctx.sc = backend(), _CUR_SCALE
        return ctx.bk.rmsnorm_f32(h, gamma)

    @staticmethod
    def backward(ctx, dy):
        h, gamma = ctx.saved_tensors
        M, d = h.shape
        dy = dy if dy.stride(1) == 1 else dy.contiguous()
        dh, _, dgamma = ctx.bk.rmsnorm_bwd(h, dy, 1, M, d, gamma=gamma)      # one "utterance" of M rows: the norm is per row
        return dh, *_un(ctx, dgamma)


class SkinnyLinearFn(torch.autograd.Function):
    """nn.Linear on rows = batch entries: `to_time_cond.1`, every FiLM / adaptive-norm `to_gamma_beta` / block `to_time_cond`
    (NS2:623, 744, 841), `to_prompt_cond.1` (NS2:860).  The weight-streaming fp32 kernel of the inference path in all three roles:
    y = x W^T + b;  dx = dy W (the weight as stored is already K-major for this product);  dW = dy^T x;  db = column sums."""

    @staticmethod
    def forward(ctx, x, w, b):
        bk = ctx.bk = backend()
        wt = bk.transpose_f32(w.detach())                                   # [K, J]
        y = bk.skinny(x.detach(), wt, b.detach() if b is not None else None)
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        bk = ctx.bk
        x, w = ctx.saved_tensors
        dy = dy if dy.is_contiguous() else dy.contiguous()
        dx = bk.skinny(dy, w.detach()) if ctx.needs_input_grad[0] else None  # [B, J] @ [J, K]
        dw = bk.skinny(bk.transpose_f32(dy), x.detach())                     # [J, B] @ [B, K]
        db = bk.colsum_rows(dy) if ctx.has_bias else None
        return dx, dw, db


def _lin(lin, x):
    return SkinnyLinearFn.apply(x if x.is_contiguous() else x.contiguous(), lin.weight, lin.bias)


class CondProjectionsFn(torch.autograd.Function):
    """EVERY FiLM / adaptive-norm projection of one forward (the 32 Wavenet `to_time_cond`, the 24-36 `to_gamma_beta` Linears:
    NS2:623, 744) as ONE weight-streaming product, the way the sampler's executor runs them (model_exec.cpp: one [K, Jtot] matrix).
    They all read the same conditioning rows t [B, K]; per module the kernel had 32 workgroups on 256 CUs and the step paid
    171 launches of ~40 us.  y_all = t W_all^T + b_all;  dt = dy_all W_all;  dW_all = dy_all^T t, whose row blocks ARE the
    parameters' gradients (returned as views);  db_all = column sums.  Same fp32 FMAs per output as the per-module product, in
    another K split (the split follows the column count)."""

    @staticmethod
    def forward(ctx, t, *wb):
        bk = backend()
        ws, bs = wb[0::2], wb[1::2]
        sizes = [int(w.shape[0]) for w in ws]
        W = torch.cat([w.detach() for w in ws], 0)                           # [Jtot, K]
        bias = torch.cat([b.detach() for b in bs], 0)
        y = bk.skinny(t.detach(), bk.transpose_f32(W), bias)                 # [B, Jtot]
        ctx.save_for_backward(t, W)
        ctx.sizes = sizes
        return tuple(o.contiguous() for o in y.split(sizes, dim=1))

    @staticmethod
    def backward(ctx, *dys):
        bk = backend()
        t, W = ctx.saved_tensors
        B = t.shape[0]
        dy = torch.cat([d if d is not None else torch.zeros(B, n, dtype=t.dtype, device=t.device) for d, n in zip(dys, ctx.sizes)], 1)
        dt = bk.skinny(dy, W) if ctx.needs_input_grad[0] else None           # [B, Jtot] @ [Jtot, K]
        dW = bk.skinny(bk.transpose_f32(dy), t.detach())                     # [Jtot, B] @ [B, K]
        db = bk.colsum_rows(dy)
        grads = []
        for gw, gb in zip(dW.split(ctx.sizes, 0), db.split(ctx.sizes, 0)):
            grads += [gw, gb]
        return (dt, *grads)


# =============================================================================================== Model.forward under autograd
def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _resampler(pr, prompt, heads):
    """PerceiverResampler (NS2:532-579) on the Functions above: per layer latents += Attention(latents, context = cat(latents,
    proj(prompt))) and latents += FeedForward(latents) (no norms in front, no conv), then RMSNorm * gamma.  Round 4 ran this part of
    the conditioned model's graph as the PyTorch composite (0.2 % of the FLOPs, but the one piece of `Model`'s backward on torch ops)."""
    b, n_p, _ = prompt.shape
    pc = getattr(pr, "proj_context", None)
    px = _c(prompt).reshape(b * n_p, -1)
    if isinstance(pc, torch.nn.Linear):                                      # dim_prompt != dim (NS2:548)
        px = GemmFn.apply(px, pc.weight, pc.bias, None, 0, 1)
    Lm, d = pr.latents.shape
    px = px.reshape(b, n_p, d)
    lat = _enter(pr.latents)[None].expand(b, -1, -1).reshape(b * Lm, d)      # (a copy: its gradient sums over the batch into pr.latents)
    for attn, ff in pr.layers:
        context = torch.cat((lat.reshape(b, Lm, d), px), dim=1).reshape(b * (Lm + n_p), d)     # cross_attn_include_queries, NS2:1060-1061
        lat = AttnFn.apply(_c(lat), None, context, attn.to_q.weight, attn.to_kv.weight, attn.to_out.weight, Lm, heads, Lm + n_p)
        l1, l2 = getattr(ff, "0"), getattr(ff, "2")
        lat = FeedForwardFn.apply(_c(lat), None, l1.weight, l1.bias, None, None, l2.weight, l2.bias, Lm)
    return RmsNormFn.apply(_c(lat), pr.norm.gamma).reshape(b, Lm, d)


def model_forward_train(m, x, times, prompt=None, cond=None, cond_drop_prob=None):
    """`Model.forward` (NS2:929-1000) as a differentiable graph whose token-sized arithmetic is HIP (module docstring).
    `m` owns the reference's parameters (this package's `Model` or `compat.HipBackedModel`).  `m.train_precision`: "exact"
    (default: bf16 x3) or "mixed" (IEEE-half product + fp8 correction terms on FMT_H8 operands under a loss scale, `_Scale`)."""
    global _CUR_PREC, _CUR_SCALE, _CUR_DEV
    tp = getattr(m, "train_precision", "exact")
    assert tp in TRAIN_PRECISIONS, f"train_precision must be one of {sorted(TRAIN_PRECISIONS)}"
    prev = (_CUR_PREC, _CUR_SCALE, _CUR_DEV)
    _CUR_PREC = TRAIN_PRECISIONS[tp]
    _CUR_DEV = x.device.index if x.is_cuda else None
    _CUR_SCALE = _Scale() if tp == "mixed" else None
    bk = backend()
    if hasattr(bk, "packs"):
        bk.packs.begin_pass()                        # every trainable weight is re-packed at its first use of this pass (see _PackedCache)
    try:
        out = _forward_train(m, x, times, prompt, cond, cond_drop_prob)
        if _CUR_SCALE is not None:
            m._last_loss_scale = _CUR_SCALE          # (inspection / tests: the scale the last backward chose)
            out = _ScaleIn.apply(out, _CUR_SCALE)
        return out
    finally:
        _CUR_PREC, _CUR_SCALE, _CUR_DEV = prev


def _forward_train(m, x, times, prompt, cond, cond_drop_prob):
    b, n, d = x.shape
    M = b * n
    p = m.cond_drop_prob if cond_drop_prob is None else cond_drop_prob
    heads = m._hip_cfg["heads"]
    # ---- rows = batch entries: time embedding and every conditioning projection (PyTorch ops, see the module docstring)
    w = getattr(m.to_time_cond, "0").weights
    tt = times.float()[:, None]
    fr = tt * w[None] * 2 * math.pi
    t = F.silu(_lin(getattr(m.to_time_cond, "1"), torch.cat((tt, fr.sin(), fr.cos()), dim=-1)))
    c = None
    h = _enter(x.float().reshape(M, d))              # (mixed arithmetic: the token-sized gradients below here are loss-scaled)
    if m.condition_on_prompt:
        assert prompt is not None and cond is not None

        def mask():
            if p == 1:
                return torch.ones(b, dtype=torch.bool, device=x.device)
            if p == 0:
                return torch.zeros(b, dtype=torch.bool, device=x.device)
            return torch.rand(b, device=x.device) < p

        dm = mask()
        pc = F.silu(_lin(getattr(m.to_prompt_cond, "1"), prompt.float().mean(dim=1)))
        pc = torch.where(dm[:, None], m.null_prompt_cond, pc)
        t = torch.cat((t, pc), dim=-1)
        # (`prompt` and `cond` come from trainable modules upstream -- the prompt / phoneme encoders, NS2:1635 -- so their gradients
        # leave the scaled domain like x's: ADVICE r5)
        c = torch.where(dm[:, None, None], _enter(m.null_prompt_tokens), _resampler(m.perceiver_resampler, _enter(prompt.float()), heads))   # [b, Lm, d]
        # cond_to_model_dim: 1x1 conv over channel-first cond (NS2:978) = a Linear over the frames
        n_c = cond.shape[-1]
        cm = GemmFn.apply(_enter(_c(cond.float().transpose(1, 2)).reshape(b * n_c, -1)), m.cond_to_model_dim.weight, m.cond_to_model_dim.bias, None,
                          n_c, 1)
        cm = cm.reshape(b, n_c, d)
        cm = torch.where(mask()[:, None, None], _enter(m.null_cond).t()[None], cm)
        if n_c > n:
            cm = cm[:, :n]
        elif n_c < n:
            cm = F.pad(cm, (0, 0, 0, n - n_c))
        h = h + _c(cm).reshape(M, d)
        c2 = _c(c.float()).reshape(b * c.shape[1], d)
    t = _c(t)
    # every conditioning projection of the pass at once (CondProjectionsFn), in the order the graph below consumes them
    lins = [blk.to_time_cond for st in m.wavenet.stacks for blk in st.blocks]
    for layer in m.transformer.layers:
        lins += [getattr(layer, str(i)).to_gamma_beta for i in ((0, 2, 4) if m.condition_on_prompt else (0, 4))]
    if all(l.bias is not None for l in lins):
        films = dict(zip(map(id, lins), CondProjectionsFn.apply(t, *[q for l in lins for q in (l.weight, l.bias)])))
    else:                                                                    # (the reference's Linears all carry a bias)
        films = {}

    def film_of(lin):
        f = films.get(id(lin))
        return f if f is not None else _lin(lin, t)

    wn = m.wavenet
    h0 = GemmFn.apply(h, wn.init_conv.weight, wn.init_conv.bias, None, n, 1)
    cols = [h0] * m._hip_cfg["wavenet_layers"]
    skip = None
    for st in wn.stacks:
        nxt = []
        for i, blk in enumerate(st.blocks):
            z = WavenetBlockFn.apply(_c(cols[i]), film_of(blk.to_time_cond), blk.conv.weight, blk.conv.bias, blk.res_conv.weight,
                                     blk.res_conv.bias, n, 2 ** i)
            nxt.append(z)
            if blk.skip_conv is not None:                                    # sum of the skip convs (NS2:685-686, 725) as a chain of residuals
                skip = GemmFn.apply(_c(z), blk.skip_conv.weight, blk.skip_conv.bias, skip if skip is None else _c(skip), n, 1)
        cols = nxt
    h = GemmFn.apply(_c(skip), wn.final_conv.weight, wn.final_conv.bias, None, n, 1)
    for layer in m.transformer.layers:
        a1 = getattr(layer, "1")
        h = AttnFn.apply(_c(h), film_of(getattr(layer, "0").to_gamma_beta), None, a1.to_q.weight, a1.to_kv.weight, a1.to_out.weight, n, heads, 0)
        if m.condition_on_prompt:
            a3 = getattr(layer, "3")
            h = AttnFn.apply(_c(h), film_of(getattr(layer, "2").to_gamma_beta), c2, a3.to_q.weight, a3.to_kv.weight, a3.to_out.weight, n,
                             heads, c.shape[1])
        ff = getattr(layer, "5")
        l1, cv, l2 = getattr(ff, "0"), getattr(getattr(ff, "2"), "1"), getattr(ff, "3")
        h = FeedForwardFn.apply(_c(h), film_of(getattr(layer, "4").to_gamma_beta), l1.weight, l1.bias, cv.weight, cv.bias, l2.weight, l2.bias, n)
    tp = m.transformer.to_pred
    out = NormLinearFn.apply(_c(h), getattr(tp, "0").gamma, getattr(tp, "1").weight, n)
    return out.reshape(b, n, d).to(x.dtype)


def available(device):
    """the HIP training path needs the library and parameters on an MI355X"""
    return device.type == "cuda" and torch.cuda.is_available()


def unsupported_reason(m):
    """None when `model_forward_train` can run `m`, else why not.  The Functions above are written for what the inference executor
    accepts (ns2_model_create): heads of 64 (q / k / v column offsets, the attention kernels' head dim, scale 1/8), dim a multiple
    of 32 (whole 32-column lines per operand row), fp32 parameters (the packs and the returned gradients are fp32).  Anything else
    must not reach the kernels (a `dim_head=32` model would read past its packed qkv planes)."""
    cfg = m._hip_cfg
    if cfg["dim_head"] != 64:
        return f"dim_head={cfg['dim_head']} (the HIP attention kernels have a head dim of 64)"
    if cfg["dim"] % 32:
        return f"dim={cfg['dim']} is not a multiple of 32"
    for name, p in m.named_parameters():
        if p.dtype != torch.float32:
            return f"parameter {name} is {p.dtype} (fp32 master weights are required)"
    return None


# =============================================================================================== the step as one HIP graph
class _LossOf(torch.nn.Module):
    """`loss_fn` as the forward of a module that owns `root`: lets torch.func.functional_call run it on substitute parameters"""

    def __init__(self, root, loss_fn):
        super().__init__()
        self.root, self.loss_fn = root, loss_fn

    def forward(self, *inputs):
        return self.loss_fn(*inputs)


class GraphedTrainStep:
    """`loss = loss_fn(*inputs); loss.backward()` of a FIXED shape captured once in a HIP graph and replayed per step.

    Why: a d512 / L12 training pass enqueues ~2 000 launches from Python (the autograd Functions above are coarse, but each is a
    dozen library calls); on a host whose cores are shared the enqueue, not the GPU, set the step time (98 ms vs 151 ms for the same
    kernels, profiles/README.md), and at BASELINE config 1's size (d128 / L6, 4 x 1024) the GPU waits for Python: 20 ms eager, 12 ms
    replayed.  Everything this file launches is capture-safe after a warm pass -- the weight packs are refreshed by ONE
    `ns2_weights_repack` launch that reads the parameters' own storage (`_PackedCache._repack_all`), the loss scale of the mixed
    arithmetic is chosen on the device (`_Scale.choose`), buffers come from PyTorch's allocator (its graph-private pool during
    capture), random draws (conditioning dropout, times / noise when `loss_fn` draws them) go through PyTorch's graph-aware
    generator -- so the replay runs the same kernels on the same operands: loss and gradients are bit-identical to the eager pass
    (tests/test_round6_gpu.py::test_graphed_training_step).

        step = GraphedTrainStep(lambda a, t, n: diffusion(a, times=t, noise=n), (audio, times, noise), model)
        loss = step(audio, times, noise)          # .grad of every trainable parameter of `model` is now this batch's gradient
        optimizer.step()                          # outside the graph (`zero_grad` between replays is unnecessary and harmless)

    `module`: the nn.Module (or a list of them) whose trainable parameters `loss_fn` reaches.  During the warm passes and the capture
    the module runs on SUBSTITUTE leaves over the same storage (torch.func.functional_call): a parameter's own AccumulateGrad node lives
    on the stream it was created on -- the default stream whenever an eager pass's `loss` is still referenced -- and the autograd engine
    would tie that stream into the capture (observed: a crash inside hipStreamEndCapture; tools/exp_graph_crash.py).  The gradients
    come back from torch.autograd.grad as the graph's own tensors and are attached as `.grad` after every replay.

    The parameters must be updated IN PLACE (every torch optimizer does); re-create the object after changing shapes, freezing /
    unfreezing parameters or loading a state dict into NEW storage.  `overflowed()` is `_Scale.overflowed` for the replayed pass.
    Tensor hooks on the parameters do not run during a replay (`distributed.GradientAllReducer` overlaps its all-reduces with an EAGER
    backward through such hooks): a data-parallel loop on the graph all-reduces `step.grads` after the call instead."""

    def __init__(self, loss_fn, example_inputs, module, warmup=2):
        root = module if isinstance(module, torch.nn.Module) else torch.nn.ModuleList(list(module))
        named = [(k, p) for k, p in root.named_parameters() if p.requires_grad]
        assert named, "GraphedTrainStep: no trainable parameter"
        self.inputs = [t.detach().clone() for t in example_inputs]
        assert self.inputs and self.inputs[0].is_cuda, "GraphedTrainStep captures a HIP graph: inputs live on the GPU"
        self.params = [p for _, p in named]
        self._subs = {"root." + k: p.detach().requires_grad_(True) for k, p in named}      # kept: the packs of the captured kernels hang on them
        leaves = list(self._subs.values())
        fn = _LossOf(root, loss_fn)
        run = lambda: torch.func.functional_call(fn, self._subs, tuple(self.inputs))       # noqa: E731
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):                     # warm passes: packs, tile images, split-K plans, function attributes all exist
            for _ in range(max(2, warmup)):               # (two at least: the first packs the substitutes' weights, the second builds the
                torch.autograd.grad(run(), leaves, allow_unused=True)      # one-launch repack table -- a synchronising build, not capturable)
        cur.wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            loss = run()
            grads = torch.autograd.grad(loss, leaves, allow_unused=True)
        self.loss = loss.detach()
        self.grads = list(grads)                          # the graph's own tensors: every replay writes them, __call__ attaches them as .grad
        self._before = None

    def __call__(self, *inputs):
        assert len(inputs) == len(self.inputs)
        for s, t in zip(self.inputs, inputs):
            if t is not s:
                s.copy_(t)
        self._before = _Scale._peek()
        self.graph.replay()
        for p, g in zip(self.params, self.grads):         # (whatever a zero_grad(set_to_none=True) or an eager pass in between left there)
            p.grad = g
        return self.loss

    def overflowed(self) -> bool:
        if self._before is None:
            return False
        after, ev = _Scale._peek()
        ev.synchronize()
        return bool((after != self._before[0]).any().item())
