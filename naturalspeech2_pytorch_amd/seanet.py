"""EnCodec's SEANet encoder / decoder on the HIP kernels (HFENC:81-347 = transformers `modeling_encodec.py`, the
in-container restatement of the un-vendored `encodec` package that audiolm_pytorch's `EncodecWrapper` runs at NS2:1445,
NS2:1496 and NS2:1611).

MI355X-first layout: channel-last rows [B * T, C] throughout (the reference is channel-first), so that every convolution is
a call of the denoiser's own MFMA GEMM family:
  * stride-1 causal convolutions (HFENC:81-181) = shifted-row GEMMs; EnCodec's REFLECT padding is a per-utterance prefix of
    mirrored rows written by `ns2_seanet_prep` (which also applies the ELU in front of the convolution and converts to operand
    planes), so the GEMM's zero fill is never reached;
  * the down-sampling convolutions (kernel 2r, stride r) = 2-tap convolutions over rows REGROUPED r at a time (a free view of
    the same buffer: [T, C] -> [T / r, r C]);
  * the up-sampling transposed convolutions (HFENC:184-244) = 2-tap convolutions producing r output frames per input row
    ([T, r C_out] -> [T r, C_out], again a free view); their causal trimming drops exactly the rows that are never computed;
  * the 2-layer LSTM (HFENC:253-266): input projections of all frames = one GEMM, the recurrence = one small kernel per step;
  * ResnetBlock (HFENC:269-301): three GEMMs, the residual sum is the last one's `resid` operand.

The module wraps an existing SEANet (`transformers.EncodecModel().encoder / .decoder`, or any module with the same layer
structure) and reads its EFFECTIVE weights (weight-norm applied: `conv.weight`); it re-packs when their content changes.
Inference only (like the codec in the reference: `codec.eval()` + `torch.no_grad()`, NS2:1443-1445, 1608-1611).
"""
import os

import torch
from torch import nn

from . import _lib, ops
from ._cache import PackedCache
from ._lib import check
from .model import _PRECISIONS


_NARROW_CONV = os.environ.get("NS2_SEANET_NARROW_CONV", "1") != "0"          # 0: the two 1-channel ends as GEMMs (A/B, tests)
_FUSED_RESBLOCK = os.environ.get("NS2_SEANET_FUSED_RESBLOCK", "1") != "0"    # 0: conv1, shortcut, conv2 as three GEMMs (A/B, tests)
_NARROW_RESBLOCK = os.environ.get("NS2_SEANET_NARROW_RESBLOCK", "1") != "0"  # 0: the 32-channel residual blocks as GEMMs too (A/B, tests)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _prep(x, B, T, C, *, in_prefix=0, elu=False, prefix=0, im2col_k=0, add=None, precision=3, ldo=None):
    """fp32 rows [B, in_prefix + T, C] -> operand planes [B * (prefix + T), ldo] (ELU, reflect prefix, im2col of 1 channel)"""
    ldo = ldo or ops.round_up(max(C, im2col_k), 32)
    out = ops._out_planes(B * (prefix + T), ldo, x.device, precision)
    check(_lib.load().ns2_seanet_prep(x.data_ptr(), x.shape[-1], in_prefix, ops._p(add), add.shape[-1] if add is not None else 0, B, T,
                                      C, int(elu), prefix, im2col_k, out.hi, out.lo, ldo, precision, _stream()), "ns2_seanet_prep")
    return out


def _prep2(x, B, T, C, *, in_prefix=0, prefix=0, elu_out=None, raw_out=None, precision=3):
    """one pass over fp32 rows [B, in_prefix + T, C]: ELU(x) and / or x into column windows (planes, col0, cols) of plane buffers
    with `prefix` mirrored rows per utterance (ns2_seanet_prep2)"""
    def win(o):
        if o is None:
            return None, None, 0, 0, 0
        pl, col0, cols = o
        assert pl.rows == B * (prefix + T) and pl.precision == precision
        return pl.hi, pl.lo, pl.ld, col0, cols
    check(_lib.load().ns2_seanet_prep2(x.data_ptr(), x.shape[-1], in_prefix, B, T, C, prefix, *win(elu_out), *win(raw_out), precision,
                                       _stream()), "ns2_seanet_prep2")


def _unpad(x, B, T, C, prefix):
    out = torch.empty(B * T, C, dtype=torch.float32, device=x.device)
    check(_lib.load().ns2_seanet_unpad(x.data_ptr(), x.shape[-1], prefix, out.data_ptr(), C, B, T, C, _stream()), "ns2_seanet_unpad")
    return out


class _Act:
    """fp32 activation rows [B * (prefix + T), C]: the first `prefix` rows of every utterance are not data"""
    __slots__ = ("x", "B", "T", "C", "prefix")

    def __init__(self, x, B, T, C, prefix=0):
        self.x, self.B, self.T, self.C, self.prefix = x, B, T, C, prefix

    def clean(self):
        if self.prefix == 0:
            return self
        return _Act(_unpad(self.x, self.B, self.T, self.C, self.prefix), self.B, self.T, self.C, 0)


class _SEANetHIP(nn.Module):
    def __init__(self, net: nn.Module, precision="exact"):
        super().__init__()
        assert precision in _PRECISIONS
        self.net = net                       # owns the parameters (weight-norm parametrised convs, LSTM)
        self.precision = precision
        self._cache = PackedCache()

    # ---- weights
    @staticmethod
    def _w(conv_module):
        c = conv_module.conv
        return c.weight.detach().float().contiguous(), (c.bias.detach().float().contiguous() if c.bias is not None else None)

    def _packed(self):
        return self._cache.get(self.net.parameters(), self._build, extra=(self.precision,))

    def refresh_weights(self):
        """re-pack iff the parameter CONTENTS changed since they were packed (`.data` writes bump no version counter); called at
        run boundaries by NaturalSpeech2.sample / .forward and EncodecWrapperHIP.refresh_weights"""
        self._cache.refresh(self.net.parameters())

    def _pack_conv(self, m):
        """EncodecConv1d (HFENC:81-181) -> (PackedWeight, bias, taps, dilation, stride, row prefix)"""
        prec = _PRECISIONS[self.precision]
        w, b = self._w(m)
        co, ci, k = w.shape
        stride, dil = m.conv.stride[0], m.conv.dilation[0]
        assert getattr(m, "causal", True) and getattr(m, "pad_mode", "reflect") == "reflect" and m.norm_type == "weight_norm", \
            "the HIP SEANet implements EnCodec's 24 kHz configuration: causal convolutions, reflect padding, weight norm"
        if stride == 1:
            if ci == 1:                                         # first encoder conv: k taps of one channel as a K = k Linear
                return dict(w=ops.PackedWeight(w.reshape(co, k).contiguous(), precision=prec), b=b, kind="im2col", k=k, co=co, ci=1,
                            w_f32=w)                             # w_f32: the fp32 vector-ALU path of the two 1-channel ends
            return dict(w=ops.PackedWeight(w, precision=prec), b=b, kind="conv", k=k, dil=dil, co=co, ci=ci, prefix=(k - 1) * dil,
                        w_f32=w if (co == 1 and dil == 1) else None)
        assert k == 2 * stride and dil == 1
        # rows regrouped `stride` at a time: W'[co][a][i * ci + c] = W[co][c][a * stride + i]
        w2 = w.reshape(co, ci, 2, stride).permute(0, 2, 3, 1).reshape(co, 2, stride * ci).permute(0, 2, 1).contiguous()
        return dict(w=ops.PackedWeight(w2, precision=prec), b=b, kind="down", r=stride, co=co, ci=ci)

    def _pack_convtr(self, m):
        """EncodecConvTranspose1d (HFENC:184-244), kernel 2r, stride r, causal trim: Out'[n][i * co + c] =
        sum_ci x[n][ci] W[ci][c][i] + x[n - 1][ci] W[ci][c][i + r]  ->  a 2-tap causal convolution with r * co outputs"""
        prec = _PRECISIONS[self.precision]
        c = m.conv
        w = c.weight.detach().float()                           # [ci, co, k]
        ci, co, k = w.shape
        r = c.stride[0]
        assert k == 2 * r and m.causal and m.trim_right_ratio == 1.0
        w = w.reshape(ci, co, 2, r)                             # [ci][co][a][i], tap index j = a * r + i
        cur = w[:, :, 0, :].permute(2, 1, 0).reshape(r * co, ci)    # row i * co + c: current frame n
        prev = w[:, :, 1, :].permute(2, 1, 0).reshape(r * co, ci)   # previous frame n - 1
        w2 = torch.stack((prev, cur), dim=-1).contiguous()      # taps: 0 = shifted by one row, 1 = unshifted
        b = c.bias.detach().float().repeat(r).contiguous() if c.bias is not None else None
        return dict(w=ops.PackedWeight(w2, precision=prec), b=b, kind="up", r=r, co=co, ci=ci)

    def _pack_resblock(self, m):
        convs = [l for l in m.block if not isinstance(l, nn.ELU)]
        assert len(convs) == 2 and not isinstance(m.shortcut, nn.Identity), "EnCodec 24 kHz: two convolutions + conv shortcut"
        p = dict(kind="res", c1=self._pack_conv(convs[0]), c2=self._pack_conv(convs[1]), sc=self._pack_conv(m.shortcut))
        c1, c2, sc = p["c1"], p["c2"], p["sc"]
        if c1["kind"] == c2["kind"] == sc["kind"] == "conv" and c2["k"] == 1 and sc["k"] == 1:
            # conv2(elu(h)) + shortcut(x) = [W_c2 | W_sc] [elu(h) ; x]: one GEMM over the concatenated K, the two operands side by
            # side in one plane buffer (32-column blocks: h in [0, hs), x in [hs, hs + xs))
            (w2, b2), (ws, bs) = self._w(convs[1]), self._w(m.shortcut)
            co, ch, cx = w2.shape[0], w2.shape[1], ws.shape[1]
            hs, xs = ops.round_up(ch, 32), ops.round_up(cx, 32)
            wcat = torch.zeros(co, hs + xs, dtype=torch.float32, device=w2.device)
            wcat[:, :ch] = w2[:, :, 0]
            wcat[:, hs:hs + cx] = ws[:, :, 0]
            zero = torch.zeros(co, dtype=torch.float32, device=w2.device)
            p["cat"] = dict(w=ops.PackedWeight(wcat, precision=_PRECISIONS[self.precision]), hs=hs, xs=xs, co=co,
                            b=(b2 if b2 is not None else zero) + (bs if bs is not None else zero))
            (w1, b1) = self._w(convs[0])
            if c1["k"] == 3 and c1["dil"] == 1 and cx == co and ch * 2 == cx and w1.shape[1] == cx:
                # the one-pass fp32 kernel for the narrow end of the stacks (ns2_seanet_resblock_narrow serves C = 32: the blocks at the
                # full sample rate): weights in its consumption order -- w1p [t][c][h], w2p [h][c], wsp [c][o]
                p["narrow"] = dict(C=cx, w1p=w1.permute(2, 1, 0).contiguous(), b1=(b1 if b1 is not None else torch.zeros(ch, device=w1.device)),
                                   w2p=w2[:, :, 0].t().contiguous(), wsp=ws[:, :, 0].t().contiguous(),
                                   b2s=((b2 if b2 is not None else zero) + (bs if bs is not None else zero)).contiguous())
        return p

    def _pack_lstm(self, m):
        prec = _PRECISIONS[self.precision]
        lstm = m.lstm
        layers = []
        for l in range(lstm.num_layers):
            g = lambda n: getattr(lstm, f"{n}_l{l}").detach().float().contiguous()     # noqa: E731
            # w_ih as a GEMM operand (input projections of a whole sequence) and, past the first layer, as plain fp32 rows: the
            # two-layer launch (ns2_lstm2) forms layer 2's input projections itself, frame by frame
            layers.append(dict(w_ih=ops.PackedWeight(g("weight_ih"), precision=prec), b_ih=g("bias_ih"), w_hh=g("weight_hh"),
                               b_hh=g("bias_hh"), w_ih_f32=g("weight_ih") if l else None))
        return dict(kind="lstm", layers=layers, H=lstm.hidden_size)

    def _build(self):
        seq = []
        pending_elu = False
        for layer in self.net.layers:
            name = type(layer).__name__
            if isinstance(layer, nn.ELU):
                pending_elu = True
                continue
            if name.endswith("ResnetBlock"):
                item = self._pack_resblock(layer)
            elif name.endswith("LSTM"):
                item = self._pack_lstm(layer)
            elif name.endswith("ConvTranspose1d"):
                item = self._pack_convtr(layer)
            elif name.endswith("Conv1d"):
                item = self._pack_conv(layer)
            else:
                raise NotImplementedError(f"SEANet layer {name}")
            item["elu"] = pending_elu                            # the activation in front of this layer
            pending_elu = False
            seq.append(item)
        return seq

    # ---- layers
    def _conv(self, a: _Act, p, elu, resid=None):
        prec = _PRECISIONS[self.precision]
        B, T = a.B, a.T
        if p.get("w_f32") is not None and resid is None and _NARROW_CONV:
            # 1 -> co / ci -> 1 channels, k = 7: fp32 on the vector ALUs, one pass (ns2_seanet_conv_narrow); other shapes: below
            y = torch.empty(B * T, p["co"], dtype=torch.float32, device=a.x.device)
            rc = _lib.load().ns2_seanet_conv_narrow(a.x.data_ptr(), a.x.shape[-1], a.prefix, B, T, p["ci"], p["co"], p["k"], int(elu),
                                                    p["w_f32"].data_ptr(), ops._p(p["b"]), y.data_ptr(), p["co"], _stream())
            if rc != _lib.NS2_UNAVAILABLE:
                check(rc, "ns2_seanet_conv_narrow")
                return _Act(y, B, T, p["co"], 0)
        if p["kind"] == "im2col":
            pl = _prep(a.x, B, T, 1, in_prefix=a.prefix, elu=elu, im2col_k=p["k"], precision=prec)
            y = ops.linear_f32(p["w"], pl, bias=p["b"], precision=prec)
            return _Act(y, B, T, p["co"], 0)
        if p["kind"] == "conv":
            P = p["prefix"]
            pl = _prep(a.x, B, T, a.C, in_prefix=a.prefix, elu=elu, prefix=P, precision=prec)
            if p["k"] == 1:
                y = ops.linear_f32(p["w"], pl, bias=p["b"], resid=resid, precision=prec)
            else:
                assert resid is None
                y = ops.linear_f32(p["w"], pl, bias=p["b"], conv_taps=p["k"], dilation=p["dil"], seq_len=P + T, precision=prec)
            return _Act(y, B, T, p["co"], P)
        if p["kind"] == "down":
            r = p["r"]
            assert T % r == 0, "the frame count must be a multiple of every stride (codec(x) truncates to multiples of 320)"
            pl = _prep(a.x, B, T, a.C, in_prefix=a.prefix, elu=elu, prefix=r, precision=prec)         # one regrouped row of reflection
            rows = B * (r + T) // r
            pl2 = ops.Planes(pl.buf.reshape(rows, -1), rows, r * a.C, pl.has_lo, pl.fmt)            # [T, C] -> [T / r, r C]: a view
            y = ops.linear_f32(p["w"], pl2, bias=p["b"], conv_taps=2, dilation=1, seq_len=1 + T // r, precision=prec)
            return _Act(y, B, T // r, p["co"], 1)
        if p["kind"] == "up":
            r = p["r"]
            pl = _prep(a.x, B, T, a.C, in_prefix=a.prefix, elu=elu, precision=prec)
            y = ops.linear_f32(p["w"], pl, bias=p["b"], conv_taps=2, dilation=1, seq_len=T, precision=prec)   # [B T, r co]
            return _Act(y.reshape(B * T * r, p["co"]), B, T * r, p["co"], 0)                                # -> [B T r, co]: a view
        raise NotImplementedError(p["kind"])

    def _resblock(self, a: _Act, p):
        """EncodecResnetBlock (HFENC:268-301): conv2(elu(conv1(elu(x)))) + shortcut(x) as TWO GEMMs: conv1, then conv2 and the
        shortcut together over the concatenated K (their operands elu(h) | x side by side in one plane buffer, written by two
        passes that also do the ELUs: ns2_seanet_prep2).  Everything runs on the row layout of conv1 -- `prefix` reflected rows
        in front of every utterance, whose outputs are not data and are skipped by whoever reads the result -- so nothing is
        ever copied just to drop prefix rows, and the shortcut's output never exists in memory."""
        prec = _PRECISIONS[self.precision]
        c1, c2, sc, cat = p["c1"], p["c2"], p["sc"], p.get("cat")
        nr = p.get("narrow")
        if nr is not None and _NARROW_RESBLOCK:
            # x read once, y written once: conv1 (k = 3, reflect), both ELUs, conv2 and the shortcut on the vector ALUs in fp32
            y = torch.empty(a.B * a.T, nr["C"], dtype=torch.float32, device=a.x.device)
            rc = _lib.load().ns2_seanet_resblock_narrow(a.x.data_ptr(), a.x.shape[-1], a.prefix, a.B, a.T, nr["C"], nr["w1p"].data_ptr(),
                                                        nr["b1"].data_ptr(), nr["w2p"].data_ptr(), nr["wsp"].data_ptr(), nr["b2s"].data_ptr(),
                                                        y.data_ptr(), nr["C"], _stream())
            if rc != _lib.NS2_UNAVAILABLE:
                check(rc, "ns2_seanet_resblock_narrow")
                return _Act(y, a.B, a.T, nr["C"], 0)
        if cat is None or not _FUSED_RESBLOCK:
            a = a.clean() if a.prefix else a                      # any other block shape: layer by layer
            h = self._conv(a, c1, elu=True)
            s = self._conv(a, sc, elu=False)
            return self._conv(h, c2, elu=True, resid=s.x)
        B, T, P = a.B, a.T, c1["prefix"]
        rows = B * (P + T)
        xe = ops._out_planes(rows, cat["xs"], a.x.device, prec)                                   # elu(x): conv1's operand
        both = ops._out_planes(rows, cat["hs"] + cat["xs"], a.x.device, prec)                     # [elu(h) | x]
        _prep2(a.x, B, T, a.C, in_prefix=a.prefix, prefix=P, elu_out=(xe, 0, cat["xs"]), raw_out=(both, cat["hs"], cat["xs"]),
               precision=prec)
        kw = dict(conv_taps=c1["k"], dilation=c1["dil"], seq_len=P + T) if c1["k"] > 1 else {}
        hf = ops.linear_f32(c1["w"], xe, bias=c1["b"], precision=prec, **kw)                      # [B (P + T), C / 2]
        _prep2(hf, B, P + T, c1["co"], elu_out=(both, 0, cat["hs"]), precision=prec)
        y = ops.linear_f32(cat["w"], both, bias=cat["b"], precision=prec)
        return _Act(y, B, T, cat["co"], P)

    def _lstm(self, a: _Act, p):
        """EncodecLSTM (HFENC:253-266): lstm(x) + x.  Three ways to run the recurrence, fastest first: both layers in one launch
        (ns2_lstm2), one launch per layer, one launch per frame.  The first two need all their workgroups resident at once; the
        launchers check that (NS2_UNAVAILABLE / an internal fallback), and a launch that still had to give up -- a device shared
        with other work -- is reported by ns2_lstm_abort_count: its output is discarded and the next way is taken."""
        import contextlib
        import ctypes
        import warnings
        prec = _PRECISIONS[self.precision]
        a = a.clean()
        B, T, H = a.B, a.T, p["H"]
        lib = _lib.load()
        layers = p["layers"]
        dev = a.x.device
        xproj1 = ops.linear_f32(layers[0]["w_ih"], _prep(a.x, B, T, H, precision=prec), bias=layers[0]["b_ih"], precision=prec)   # [B T, 4H]

        CH = 32                                    # batch rows per one-launch recurrence (rows are independent: larger batches in chunks)
        chunks = [(b0, min(CH, B - b0)) for b0 in range(0, B, CH)]

        def rows(t, b0, nb):                       # rows b0 T .. (b0 + nb) T of a [B T, *] tensor (a view)
            return t[b0 * T:(b0 + nb) * T]

        def fused():
            l1, l2 = layers
            nstate = int(lib.ns2_lstm2_state_floats())
            state = torch.empty(nstate, dtype=torch.float32, device=dev)
            out = torch.empty(B * T, H, dtype=torch.float32, device=dev)
            for b0, nb in chunks:
                rc = lib.ns2_lstm2(rows(xproj1, b0, nb).data_ptr(), 4 * H, l1["w_hh"].data_ptr(), l1["b_hh"].data_ptr(),
                                   l2["w_ih_f32"].data_ptr(), l2["b_ih"].data_ptr(), l2["w_hh"].data_ptr(), l2["b_hh"].data_ptr(),
                                   state.data_ptr(), nstate, rows(a.x, b0, nb).data_ptr(), H, rows(out, b0, nb).data_ptr(), H, nb, T,
                                   _stream())
                if rc == _lib.NS2_UNAVAILABLE:
                    assert b0 == 0, "availability is a property of the device, not of the chunk"
                    return None
                check(rc, "ns2_lstm2")
            return out

        def layer_by_layer(per_frame):
            # the entry point runs one launch per frame when it gets only the minimal scratch (include/ns2hip.h)
            x, out = a.x, None
            for i, l in enumerate(layers):
                xproj = xproj1 if i == 0 else ops.linear_f32(l["w_ih"], _prep(x, B, T, H, precision=prec), bias=l["b_ih"], precision=prec)
                out = torch.empty(B * T, H, dtype=torch.float32, device=dev)
                resid = a.x if i + 1 == len(layers) else None
                for b0, nb in ([(0, B)] if per_frame else chunks):
                    nstate = 3 * nb * H if per_frame else int(lib.ns2_lstm_state_floats(nb, H))
                    state = torch.empty(nstate, dtype=torch.float32, device=dev)
                    check(lib.ns2_lstm_layer(rows(xproj, b0, nb).data_ptr(), 4 * H, l["w_hh"].data_ptr(), l["b_hh"].data_ptr(),
                                             state.data_ptr(), nstate, ops._p(rows(resid, b0, nb) if resid is not None else None), H,
                                             rows(out, b0, nb).data_ptr(), H, nb, T, H, _stream()), "ns2_lstm_layer")
                x = out
            return out

        ways = ([("both layers in one launch", fused)] if len(layers) == 2 and H == 512 else []) + \
               [("one launch per layer", lambda: layer_by_layer(False)), ("one launch per frame", lambda: layer_by_layer(True))]
        for name, run in ways:
            out = run()
            if out is None:
                continue
            n = ctypes.c_int64(0)
            with (torch.cuda.device(dev) if dev.type == "cuda" else contextlib.nullcontext()):   # the counter lives on the device the recurrence ran on
                check(lib.ns2_lstm_abort_count(1, ctypes.byref(n)), "ns2_lstm_abort_count")   # one synchronisation per codec run
            if n.value == 0:
                return _Act(out, B, T, H, 0)
            warnings.warn(f"LSTM recurrence ({name}) gave up waiting for a frame: not all of its workgroups were resident (CU masking / "
                          f"a device shared with other work); running it the next way.  NS2_LSTM_FUSED=0 / NS2_LSTM_PERSISTENT=0 skip "
                          f"the one-launch paths altogether.")
        raise _lib.Ns2Error("the LSTM recurrence could not complete on this device")

    def algorithmic_work(self, B, T, C):
        """(FLOPs, compulsory bytes) of one pass over B utterances of T rows x C channels, counted from the layer shapes:
        2 x MACs of every convolution / transposed convolution / LSTM Linear (HFENC:81-347), and the bytes that must cross HBM
        at least once (input, output, fp32 weights) -- what a fully fused codec would move; used for bench.py's roofline."""
        flops, wbytes = 0.0, 0.0
        t, c = T, C

        def conv(p, t, c):
            k = p.get("k", 1)
            if p["kind"] == "im2col":
                return 2.0 * B * t * k * p["co"], t, p["co"], 4.0 * k * p["co"]
            if p["kind"] == "conv":
                return 2.0 * B * t * p["ci"] * p["co"] * k, t, p["co"], 4.0 * k * p["ci"] * p["co"]
            if p["kind"] == "down":
                r = p["r"]
                return 2.0 * B * (t // r) * p["ci"] * p["co"] * 2 * r, t // r, p["co"], 4.0 * 2 * r * p["ci"] * p["co"]
            r = p["r"]                                           # "up"
            return 2.0 * B * t * p["ci"] * p["co"] * 2 * r, t * r, p["co"], 4.0 * 2 * r * p["ci"] * p["co"]

        for p in self._packed():
            if p["kind"] == "res":
                for q in (p["c1"], p["sc"]):
                    f, _, _, w = conv(q, t, c)
                    flops += f; wbytes += w
                f, _, _, w = conv(p["c2"], t, p["c1"]["co"])
                flops += f; wbytes += w
            elif p["kind"] == "lstm":
                H = p["H"]
                flops += len(p["layers"]) * 2.0 * B * t * (8 * H * H)            # W_ih and W_hh: 4H x H each
                wbytes += len(p["layers"]) * 4.0 * 8 * H * H
            else:
                f, t, c, w = conv(p, t, c)
                flops += f; wbytes += w
        return flops, 4.0 * B * T * C + 4.0 * B * t * c + wbytes

    @torch.no_grad()
    def _run(self, x_rows, B, T, C):
        a = _Act(x_rows, B, T, C, 0)
        for p in self._packed():
            if p["kind"] == "res":
                assert not p["elu"]
                a = self._resblock(a, p)
            elif p["kind"] == "lstm":
                assert not p["elu"]
                a = self._lstm(a, p)
            else:
                a = self._conv(a, p, elu=p["elu"])
        return a.clean()


class SEANetEncoderHIP(_SEANetHIP):
    """`encoder(wav [b, 1, t]) -> latents [b, 128, n]` (the call `EncodecWrapperHIP.forward` makes), HFENC:304-327"""

    @torch.no_grad()
    def forward(self, wav):
        assert wav.ndim == 3 and wav.shape[1] == 1, "mono audio [b, 1, t]"
        B, _, T = wav.shape
        a = self._run(wav.reshape(B * T, 1).float().contiguous(), B, T, 1)
        return a.x.reshape(B, a.T, a.C).transpose(1, 2)


class SEANetDecoderHIP(_SEANetHIP):
    """`decoder(latents [b, 128, n]) -> wav [b, 1, t]`, HFENC:330-358"""

    @torch.no_grad()
    def forward(self, latents):
        B, C, N = latents.shape
        a = self._run(latents.transpose(1, 2).reshape(B * N, C).float().contiguous(), B, N, C)
        return a.x.reshape(B, a.T, a.C).transpose(1, 2)
