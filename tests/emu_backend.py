"""TEST INFRASTRUCTURE: a plain-torch (CPU) restatement of every `training.HipBackend` call, with the SAME layout contracts
(padded leading dimensions, transposed planes stacked per tap, shifts inside utterances, fixed column offsets), so that the host
logic of `naturalspeech2_pytorch_amd/training.py` -- chain rule, tap flips, shift signs, which tensor is saved for what -- is
checked against torch autograd without a GPU (tests/test_training_cpu.py).  Never imported by the product.

Semantics restated from the kernels' contracts:
  * gemm (gemm.hip / gemm2.hip issue_tile): out[m] = sum_t W[:, :, t] a[m - (pl - t) dil], pl = taps - 1 (causal) or `pad_left`,
    the source row must lie in the utterance of m; columns >= N of an fp32 output are NOT written (NaN here, to catch readers);
  * tplanes (backward.hip): T[c][m] = x[m - shift][c] inside the utterance, zero elsewhere and in the padding rows / columns;
  * attention lse = log2-sum-exp of the scaled scores.
"""
import math

import torch
import torch.nn.functional as F


def rup(x, m):
    return (x + m - 1) // m * m


class EP:                     # operand planes [rows, ld] (values kept in fp32)
    def __init__(self, t):
        self.t, self.rows, self.ld = t, t.shape[0], t.shape[1]

    @property
    def device(self):
        return self.t.device


class ETP:                    # transposed planes [rows, ld]
    def __init__(self, t):
        self.t, self.rows, self.ld = t, t.shape[0], t.shape[1]


class EPW:
    def __init__(self, w):
        self.w = w.detach().float().contiguous()
        self.rows = w.shape[0]


class EmuBackend:
    name = "emu"

    def __init__(self):
        self.calls = []

    def pack(self, key, params, make_src, parts=None):
        src = make_src()
        return EPW(src if not isinstance(src, tuple) else src[0])

    def split(self, x, C=None):
        M, C = x.shape
        out = torch.zeros(M, rup(C, 32))
        out[:, :C] = x
        return EP(out)

    def rmsnorm(self, x, seq_len, gamma=None, cond=None):
        d = x.shape[1]
        y = F.normalize(x, dim=-1) * math.sqrt(d)
        if gamma is not None:
            y = y * gamma
        if cond is not None:
            b = x.shape[0] // seq_len
            g, be = cond[:, :d], cond[:, d:2 * d]
            y = (y.reshape(b, seq_len, d) * g[:, None] + be[:, None]).reshape(-1, d)
        return self.split(y)

    def rmsnorm_f32(self, x, gamma):
        return F.normalize(x, dim=-1) * math.sqrt(x.shape[1]) * gamma

    def _gemm(self, pw, a, taps, dil, seq_len, pad_left):
        w = pw.w
        M = a.rows
        if taps == 0:
            assert w.ndim == 2
            return a.t[:, :w.shape[1]] @ w.t()
        assert w.ndim == 3 and w.shape[2] == taps
        C = w.shape[1]
        pl = taps - 1 if pad_left < 0 else pad_left
        out = torch.zeros(M, w.shape[0])
        x = a.t[:, :C]
        n = torch.arange(M) % seq_len if seq_len > 0 else None
        for t in range(taps):
            shift = (pl - t) * dil
            src = torch.arange(M) - shift
            ok = (src >= 0) & (src < M)
            if n is not None:
                ok &= ((n - shift) >= 0) & ((n - shift) < seq_len)
            xs = torch.zeros_like(x)
            xs[ok] = x[src[ok]]
            out += xs @ w[:, :, t].t()
        return out

    def gemm_f32(self, pw, a, bias=None, resid=None, taps=0, dil=1, seq_len=0, pad_left=-1):
        y = self._gemm(pw, a, taps, dil, seq_len, pad_left)
        N = y.shape[1]
        if bias is not None:
            y = y + bias
        if resid is not None:
            y = y + resid[:, :N]
        out = torch.full((a.rows, rup(N, 32)), float("nan"))
        out[:, :N] = y
        return out

    def gemm_split(self, pw, a, bias=None, taps=0, dil=1, seq_len=0, attn=False):
        y = self._gemm(pw, a, taps, dil, seq_len, -1)
        if bias is not None:
            y = y + bias
        return self.split(y)

    def film_gate_fwd(self, h, film, seq_len, d):
        b = h.shape[0] // seq_len
        z = (h[:, :d].reshape(b, seq_len, d) * film[:, None, :d] + film[:, None, d:2 * d]).reshape(-1, d)
        return z.tanh() * z.sigmoid()

    def geglu_fwd(self, pre, f):
        return self.split(F.gelu(pre[:, f:2 * f]) * pre[:, :f])

    def attention(self, q, q_col0, k, k_col0, vt, B, H, Nq, Nk):
        a = H * 64
        qq = q.t[:, q_col0:q_col0 + a].reshape(B, Nq, H, 64).transpose(1, 2)
        kk = k.t[:, k_col0:k_col0 + a].reshape(B, Nk, H, 64).transpose(1, 2)
        vv = vt.t.reshape(B, H, 64, -1)[..., :Nk].transpose(2, 3)                     # V^T [B][H*64][ld] -> [B, H, Nk, 64]
        s = qq @ kk.transpose(2, 3) * 0.125
        lse = torch.logsumexp(s, dim=-1) / math.log(2.0)
        o = (s.softmax(-1) @ vv).transpose(1, 2).reshape(B * Nq, a)
        return self.split(o), lse

    def skinny(self, x, wt, bias=None):
        y = x @ wt
        return y + bias if bias is not None else y

    def transpose_f32(self, x):
        return x.t().contiguous()

    def colsum_rows(self, x):
        return x.sum(0)

    # ---- backward pieces
    def _shifted(self, x, seq_len, shift):
        M = x.shape[0]
        src = torch.arange(M) - shift
        ok = (src >= 0) & (src < M)
        if seq_len > 0:
            n = torch.arange(M) % seq_len
            ok &= ((n - shift) >= 0) & ((n - shift) < seq_len)
        out = torch.zeros_like(x)
        out[ok] = x[src[ok]]
        return out

    def grad_prep(self, x, C, want_row=False, want_t=False, want_colsum=False, seq_len=0, per_batch=False, t_rows=None, attn=False):
        assert not torch.isnan(x[:, :C]).any(), "grad_prep read an unwritten column"
        M = x.shape[0]
        row = self.split(x[:, :C]) if want_row else None
        tp = None
        if want_t:
            t_rows = t_rows or C
            if per_batch:
                B = M // seq_len
                t = torch.zeros(B * t_rows, rup(seq_len, 32))
                for b in range(B):
                    t[b * t_rows:b * t_rows + C, :seq_len] = x[b * seq_len:(b + 1) * seq_len, :C].t()
            else:
                t = torch.zeros(t_rows, rup(M, 32))
                t[:C, :M] = x[:, :C].t()
            tp = ETP(t)
        return row, tp, (x[:, :C].sum(0) if want_colsum else None)

    def transpose(self, p, col0, C, seq_len, shifts=(0,), per_batch=False, pad_rows=256):
        M = p.rows
        Cp = rup(C, 32)
        x = p.t[:, col0:col0 + C]
        if per_batch:
            B = M // seq_len
            t = torch.zeros(B * C, rup(seq_len, 32))
            for b in range(B):
                t[b * C:(b + 1) * C, :seq_len] = x[b * seq_len:(b + 1) * seq_len].t()
            return ETP(t)
        T = len(shifts)
        rows = max((T - 1) * Cp + rup(Cp, pad_rows), rup(T * Cp, pad_rows))
        t = torch.zeros(rows, rup(M, 32))
        for i, sh in enumerate(shifts):
            t[i * Cp:i * Cp + C, :M] = self._shifted(x, seq_len, sh).t()
        return ETP(t)

    def wgrad(self, dyt, xt, R, T, K, row_off=0):
        Kp = rup(K, 32)
        dw = torch.zeros(R, K, T)
        for t in range(T):
            dw[:, :, t] = dyt.t[:R] @ xt.t[row_off + t * Kp:row_off + t * Kp + K].t()
        return dw

    # the emulation forms weight gradients from the row planes only for wide gradients, like the library (ns2_wgrad_rows_preferred)
    def wgrad_rows_ok(self, R, T, K, seq_len, M):
        return T * rup(K, 32) > 128 and (T == 1 or seq_len >= 32)       # (the library also asks for enough 256 x 256 tiles: a speed rule)

    def _mm_t(self, a_t, b):
        return a_t @ b

    def wgrad_rows(self, dy, x, R, T, K, dil=1, seq_len=0):
        assert dy.rows == x.rows
        dw = torch.zeros(R, K, T)
        for t in range(T):
            dw[:, :, t] = self._mm_t(dy.t[:, :R].t(), self._shifted(x.t[:, :K], seq_len if T > 1 else 0, (T - 1 - t) * dil))
        return dw

    def film_gate_bwd(self, dg, h, film, B, seq_len, d):
        hh = h[:, :d].reshape(B, seq_len, d)
        z = hh * film[:, None, :d] + film[:, None, d:2 * d]
        th, sg = z.tanh(), z.sigmoid()
        dz = dg[:, :d].reshape(B, seq_len, d) * ((1 - th * th) * sg + th * sg * (1 - sg))
        dh = (dz * film[:, None, :d]).reshape(-1, d)
        return dh, torch.cat(((dz * hh).sum(1), dz.sum(1)), dim=-1)

    def geglu_bwd(self, dh, pre, f):
        x, g = pre[:, :f], pre[:, f:2 * f]
        dy = dh[:, :f]
        assert not torch.isnan(dy).any()
        Phi = 0.5 * (1 + torch.erf(g / math.sqrt(2.0)))
        phi = torch.exp(-0.5 * g * g) / math.sqrt(2 * math.pi)
        out = torch.full((pre.shape[0], rup(2 * f, 32)), float("nan"))
        out[:, :f] = dy * g * Phi
        out[:, f:2 * f] = dy * x * (Phi + g * phi)
        return out

    def rmsnorm_bwd(self, x, dy, B, seq_len, d, gamma=None, cond=None, dx_add=None):
        dy = dy[:, :d]
        assert not torch.isnan(dy).any()
        r = math.sqrt(d) / x.norm(dim=-1, keepdim=True).clamp(min=1e-12)
        nh = x * r
        gp = gamma if gamma is not None else torch.ones(d)
        gc = cond[:, :d].repeat_interleave(seq_len, 0) if cond is not None else torch.ones(1, d)
        dn = dy * gc * gp
        dx = r * (dn - nh * (nh * dn).sum(-1, keepdim=True) / d)
        if dx_add is not None:
            dx = dx + dx_add[:, :d]
        dcond = None
        if cond is not None:
            dcond = torch.cat(((dy * nh * gp).reshape(B, seq_len, d).sum(1), dy.reshape(B, seq_len, d).sum(1)), -1)
        dgamma = (dy * gc * nh).sum(0) if gamma is not None else None
        return dx, dcond, dgamma

    def attention_delta(self, do, o, B, H, Nq):
        a = H * 64
        return (do[:, :a] * o.t[:, :a]).reshape(B, Nq, H, 64).sum(-1).transpose(1, 2).contiguous()

    def new_planes(self, M, C):
        return EP(torch.full((M, rup(C, 32)), float("nan")))

    def attention_bwd(self, q, q_col0, k, k_col0, v, v_col0, do_row, lse, delta, B, H, Nq, Nk, dq=None, dkv=None, planes=None):
        a = H * 64
        if planes is not None:                              # the three gradients as operand planes (same values: the emulation keeps fp32)
            dq, dkv = (planes[0].t, planes[1]), (planes[0].t, planes[2], planes[3])
        hd = lambda t, c0, n: t[:, c0:c0 + a].reshape(B, n, H, 64).transpose(1, 2)      # noqa: E731
        qq, kk, vv = hd(q.t, q_col0, Nq), hd(k.t, k_col0, Nk), hd(v.t, v_col0, Nk)
        dO = hd(do_row.t, 0, Nq)
        s = qq @ kk.transpose(2, 3) * 0.125
        P = torch.exp2(s / math.log(2.0) - lse[..., None])
        dP = dO @ vv.transpose(2, 3)
        dS = P * (dP - delta[..., None])
        unhd = lambda t, n: t.transpose(1, 2).reshape(B * n, a)                         # noqa: E731
        if dq is not None:
            dq[0][:, dq[1]:dq[1] + a] = unhd(dS @ kk * 0.125, Nq)
        if dkv is not None:
            dkv[0][:, dkv[1]:dkv[1] + a] = unhd(dS.transpose(2, 3) @ qq * 0.125, Nk)
            dkv[0][:, dkv[2]:dkv[2] + a] = unhd(P.transpose(2, 3) @ dO, Nk)


# ---------------------------------------------------------------------------------------------- the mixed training arithmetic
def _h8_parts(x):
    """x -> (half(x), e5m2((x - half(x)) 2^12) / 2^12, e5m2(x)) as float64: the three parts of an FMT_H8 element (ns2_common.h)"""
    x = x.double().clamp(-57344, 57344)
    h = x.float().to(torch.float16).double()
    l = ((x - h) * 4096).float().to(torch.float8_e5m2).double() / 4096
    return h, l, x.float().to(torch.float8_e5m2).double()


def mixed_mm(a, w_t):
    """a [M, K] @ w_t [K, N] as the precision-4 GEMMs evaluate it: a_h . w_h + e5m2(a) . w_l + a_l . e5m2(w), wide accumulation"""
    ah, al, a8 = _h8_parts(a)
    wh, wl, w8 = _h8_parts(w_t)
    return (ah @ wh + a8 @ wl + al @ w8).float()


class MixedEmuBackend(EmuBackend):
    """EmuBackend whose contractions (forward, dgrad, wgrad) round their operands like FMT_H8 lines: values beyond the IEEE-half range
    are clamped and tiny ones lose their bits -- what the loss scale of training._Scale is for (tests/test_training_cpu.py)"""
    name = "emu-mixed"

    def _mm(self, a, w_t):
        return mixed_mm(a, w_t)

    def _gemm(self, pw, a, taps, dil, seq_len, pad_left):
        w = pw.w
        M = a.rows
        if taps == 0:
            return self._mm(a.t[:, :w.shape[1]], w.t())
        C = w.shape[1]
        pl = taps - 1 if pad_left < 0 else pad_left
        out = torch.zeros(M, w.shape[0])
        x = a.t[:, :C]
        for t in range(taps):
            out += self._mm(self._shifted(x, seq_len, (pl - t) * dil), w[:, :, t].t())
        return out

    def wgrad(self, dyt, xt, R, T, K, row_off=0):
        Kp = rup(K, 32)
        dw = torch.zeros(R, K, T)
        for t in range(T):
            dw[:, :, t] = self._mm(dyt.t[:R], xt.t[row_off + t * Kp:row_off + t * Kp + K].t())
        return dw

    def _mm_t(self, a_t, b):
        return self._mm(a_t, b)
