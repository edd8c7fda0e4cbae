"""`Model` — drop-in for the reference denoiser class (NS2:811-1000) whose inference forward runs entirely in the
hand-written HIP kernels of libns2hip (csrc/model_exec.cpp enqueues the whole step on the current HIP stream).

Boundary kept from the reference:
  * constructor signature NS2:814-831, attributes `.dim`, `.condition_on_prompt`, `.cond_drop_prob`, `.device`;
  * `forward(x, times, prompt=None, prompt_mask=None, cond=None, cond_drop_prob=None)` NS2:929-937 and
    `forward_with_cond_scale(*args, cond_scale=1., **kwargs)` NS2:914-927;
  * `state_dict()` key names and shapes (checked against the reference's in tests/test_host_cpu.py::
    test_state_dict_contract_matches_reference), so
    reference checkpoints load unchanged;  survives copy.deepcopy (EMA, NS2:1793-1798).

The parameter-holder modules below exist only to own same-named parameters with the reference's default
initialisation; their arithmetic lives in HIP.  PyTorch provides device memory and the stream, nothing else.
"""
import ctypes
import math
import time
from typing import Optional

import torch
from torch import nn

from . import _lib
from ._lib import ModelConfig, check
from ._cache import tensors_fingerprint


# ------------------------------------------------------------------------------------------ parameter holders
class _Holder(nn.Module):
    """container whose children are registered under explicit (possibly numeric) names; None entries are skipped
    exactly like the reference's `Sequential` helper drops them."""

    def __init__(self, **children):
        super().__init__()
        for k, v in children.items():
            if v is not None:
                self.add_module(k.lstrip("_"), v)


def _seq(*mods):
    h = _Holder()
    i = 0
    for m in mods:
        if m is None:
            continue
        h.add_module(str(i), m)
        i += 1
    return h


class _NoParams(nn.Module):      # stands in for parameter-free reference modules (Rearrange, GEGLU, SiLU, Reduce)
    pass


class _SinusoidalWeights(nn.Module):                       # LearnedSinusoidalPosEmb NS2:108-113
    def __init__(self, dim):
        super().__init__()
        assert dim % 2 == 0
        self.weights = nn.Parameter(torch.randn(dim // 2))


class _RMSNorm(nn.Module):                                 # NS2:727-735
    def __init__(self, dim, scale=True, dim_cond=None):
        super().__init__()
        self.to_gamma_beta = nn.Linear(dim_cond, dim * 2) if dim_cond is not None else None
        self.gamma = nn.Parameter(torch.ones(dim)) if scale else None


class _Attention(nn.Module):                               # NS2:1029-1053
    def __init__(self, dim, dim_head, heads, dim_context=None):
        super().__init__()
        inner = dim_head * heads
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_kv = nn.Linear(dim_context or dim, inner * 2, bias=False)
        self.to_out = nn.Linear(inner, dim, bias=False)


def _feedforward(dim, mult, causal_conv):                  # NS2:1009-1025
    inner = int(dim * mult * 2 / 3)
    conv = _seq(_NoParams(), nn.Conv1d(inner, inner, 3), _NoParams()) if causal_conv else None
    return _seq(nn.Linear(dim, inner * 2), _NoParams(), conv, nn.Linear(inner, dim))


class _WavenetBlock(nn.Module):                            # NS2:597-620
    def __init__(self, dim, dilation, skip, dim_cond_mult):
        super().__init__()
        self.to_time_cond = nn.Linear(dim * dim_cond_mult, dim * 2)
        self.conv = nn.Conv1d(dim, dim, 3, dilation=dilation)
        self.res_conv = nn.Conv1d(dim, dim, 1)
        self.skip_conv = nn.Conv1d(dim, dim, 1) if skip else None


class _NativeState:
    """ctypes handle + caches; never copied (a deep copy re-packs lazily from its own parameters)."""

    def __init__(self):
        self.handle = None
        self.sig = None
        self.fingerprint = None
        self.ws = None
        self.cond_cache = {}
        self.keepalive = None
        # staleness / range guards (HipDenoiserMixin._guards)
        self.last_call = None            # wall clock of the last HIP forward
        self.calls_since_refresh = 0
        self.mode_flag = None            # module.training at the last HIP forward
        self.autograd_seen = False       # an autograd forward ran since the last HIP forward (a training step happened)
        self.sat_pinned = None           # pinned int32[4]: target of ns2_saturation_peek_async
        self.sat_event = None
        self.sat_seen = 0                # counter total already accounted for
        self.calls_since_peek = 0
        self.peek_now = True             # read the counters right behind the next forward (first forward of a run)
        # sampled content checksum (ns2_model_param_checksum): what the packed weights were made from, and the read in flight
        self.chk_ref = None              # torch.Tensor (CPU) taken at pack time
        self.chk_pinned = None
        self.chk_event = None
        self.calls_since_chk = 0

    def __deepcopy__(self, memo):
        return _NativeState()

    def __getstate__(self):
        return {}

    def __setstate__(self, state):
        self.__init__()

    def release(self):
        if self.handle is not None:
            try:
                _lib.load().ns2_model_destroy(self.handle)
            except Exception:
                pass
        self.handle = None
        self.sig = None
        self.fingerprint = None
        self.cond_cache = {}
        self.sat_event = None
        self.chk_ref = self.chk_event = None

    def __del__(self):
        self.release()


_CFG_KEYS = ("dim", "depth", "dim_head", "heads", "ff_mult", "wavenet_layers", "wavenet_stacks", "dim_cond_mult",
             "condition_on_prompt", "dim_prompt", "num_latents_m", "resampler_depth")
_PRECISIONS = {"exact": 3, "mixed": 4, "half": 2, "fast": 1}          # op-level arithmetic modes (include/ns2hip.h)
_MODEL_PRECISIONS = dict(_PRECISIONS, hybrid=5, hybrid_ff=6)           # + the per-site plans of the denoiser: "mixed" with the FF causal conv
                                                                       # and the Wavenet's dilated convs "half" (5); the whole FF branch "half" (6)


# ------------------------------------------------------------------------------------------ HIP execution mixin
class HipDenoiserMixin:
    """Runs `forward` / `forward_with_cond_scale` of an nn.Module that OWNS the reference `Model`'s parameters (the state_dict
    key contract of SURVEY §8b) in libns2hip.  Used by this package's `Model` (parameter holders only) and by
    `compat.HipBackedModel` (a subclass of the reference's own `Model` class, for use inside the reference's
    `NaturalSpeech2`).  The host class provides `_forward_autograd` for the calls that need autograd."""

    def _hip_init(self, cfg: dict, precision: str):
        assert precision in _MODEL_PRECISIONS, f"precision must be one of {sorted(_MODEL_PRECISIONS)}"
        assert set(cfg) == set(_CFG_KEYS)
        self._hip_cfg = dict(cfg)
        self.precision = precision
        self._native = _NativeState()

    # ---- cache invalidation: version counters do not see `.data` writes (ema_pytorch updates its shadow model that way)
    REFRESH_IDLE_S = 0.25      # a pause this long between two HIP forwards marks a new sampling run
    REFRESH_EVERY = 256        # ... and the full content check runs at least this often inside a run
    CHECKSUM_EVERY = 8         # forwards between two ASYNCHRONOUS sampled checksums of the parameters (no synchronisation): bounds the
                               # staleness after a `.data` write to CHECKSUM_EVERY + 1 forwards whatever the host's timing
    SAT_PEEK_EVERY = 16        # forwards between two asynchronous reads of the range-guard counters

    def invalidate(self):
        """drop the packed weights; the next inference forward re-packs from the current parameters"""
        self._native.release()

    def refresh_weights(self):
        """re-pack iff the parameter CONTENTS changed since they were packed (one device reduction + one host read)."""
        ns = self._native
        if ns.handle is None:
            return False
        ns.calls_since_refresh = 0
        fp = tensors_fingerprint(list(self.parameters()))
        if fp != ns.fingerprint:
            ns.release()
            return True
        if self._range_guarded() and ns.sat_event is None:
            # run boundary: what the per-device counters hold by now (another model, a codec run) is not this run's
            from . import ops
            ns.sat_seen = ops.saturation_count(reset=False, device=next(self.parameters()).device)
        return False

    def _guards(self):
        """Called at the top of every HIP forward, whoever the caller is (this package's sampler, the reference's own
        `NaturalSpeech2` / `Trainer` around `compat.HipBackedModel`, a bare `Model.forward`):

          * staleness: `(data_ptr, _version)` cannot see writes through `.data`.  The content fingerprint (a device reduction + a
            host read, i.e. a synchronisation) is therefore re-taken at every RUN BOUNDARY -- the first HIP forward after a pause of
            REFRESH_IDLE_S, after an autograd forward (a training step), after a train()/eval() flip -- and every REFRESH_EVERY
            forwards inside a run.  Sampling steps are enqueued milliseconds apart and pay nothing;
          * range: in the IEEE-half modes a clamped activation is finite but wrong.  Every SAT_PEEK_EVERY forwards the per-device
            counters are copied to pinned memory behind the step (no synchronisation) and looked at on a later call; a new count
            raises Ns2Error.  `check_saturation(sync=True)` closes a run (the sampler of this package calls it)."""
        ns = self._native
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            return                           # inside a HIP-graph capture nothing may synchronise; the caller checked before capturing
        now = time.monotonic()
        if ns.handle is not None:
            boundary = (ns.last_call is None or now - ns.last_call > self.REFRESH_IDLE_S or ns.autograd_seen
                        or ns.mode_flag != self.training or ns.calls_since_refresh >= self.REFRESH_EVERY)
            if boundary:
                self.refresh_weights()
                ns.peek_now = True
        ns.autograd_seen, ns.mode_flag = False, self.training     # (last_call is stamped when the forward has been enqueued)
        ns.calls_since_refresh += 1
        self._poll_checksum()
        self.check_saturation(sync=False)

    def _poll_checksum(self):
        """deterministic half of the staleness guard: every CHECKSUM_EVERY forwards a sampled checksum of the parameters is computed
        on the step's stream and copied to pinned memory (no synchronisation); a later call compares it with the one taken when the
        weights were packed and drops the pack on a difference.  Independent of wall-clock time and of the caller."""
        ns = self._native
        if ns.handle is None or ns.chk_ref is None:
            return
        ev = ns.chk_event
        if ev is not None and ev.query():
            ns.chk_event = None
            if not torch.equal(ns.chk_pinned.view(torch.int32), ns.chk_ref.view(torch.int32)):     # bitwise: NaNs compare equal to themselves
                ns.release()                                      # re-packed by this call's _ensure_native()
                ns.peek_now = True
                return
        ns.calls_since_chk += 1
        if ns.chk_event is None and ns.calls_since_chk >= self.CHECKSUM_EVERY:
            dev = next(self.parameters()).device
            with torch.cuda.device(dev):
                check(_lib.load().ns2_model_param_checksum(ns.handle, ns.chk_pinned.data_ptr(), ns.chk_pinned.numel(),
                                                           torch.cuda.current_stream().cuda_stream), "ns2_model_param_checksum")
                ns.chk_event = torch.cuda.Event()
                ns.chk_event.record()
            ns.calls_since_chk = 0

    def _range_guarded(self):
        return self.precision in ("half", "mixed", "hybrid", "hybrid_ff")

    def check_saturation(self, sync=False):
        """look at the last asynchronous counter read (sync=True: take one now and wait for it); raises Ns2Error on a new count"""
        ns = self._native
        if not self._range_guarded() or ns.handle is None:
            return
        if sync:
            self._peek_saturation()
        ev = ns.sat_event
        if ev is None or not (sync or ev.query()):
            return
        if sync:
            ev.synchronize()
        ns.sat_event = None
        tot = int(ns.sat_pinned.sum().item())
        new, ns.sat_seen = tot - ns.sat_seen, tot
        if new > 0:
            raise _lib.Ns2Error(
                f"{new} activation conversions left the IEEE-half range (|x| > 65504) at precision='{self.precision}': the "
                f"results of the last forwards are clamped and wrong.  Use precision='exact' (bf16 planes keep the fp32 exponent "
                f"range) for this checkpoint.")

    def _peek_saturation(self):
        ns = self._native
        dev = next(self.parameters()).device
        if ns.sat_pinned is None:
            ns.sat_pinned = torch.zeros(4, dtype=torch.int32).pin_memory()
        with torch.cuda.device(dev):
            if ns.sat_event is not None:
                ns.sat_event.synchronize()          # the pinned words are about to be rewritten
            check(_lib.load().ns2_saturation_peek_async(ns.sat_pinned.data_ptr(), torch.cuda.current_stream().cuda_stream),
                  "ns2_saturation_peek_async")
            ns.sat_event = torch.cuda.Event()
            ns.sat_event.record()
        ns.calls_since_peek = 0

    def _apply(self, fn, *args, **kwargs):               # .to() / .cuda() / .float(): parameters move or are rewritten
        out = super()._apply(fn, *args, **kwargs)
        if getattr(self, "_native", None) is not None:
            self._native.release()
        return out

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        if getattr(self, "_native", None) is not None:
            self._native.release()
        return out

    # ---- native plumbing
    def _signature(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters()) + (self.precision,)

    def _ensure_native(self):
        ns = self._native
        sig = self._signature()
        if ns.handle is not None and ns.sig == sig:
            return ns
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise _lib.Ns2Error("Model parameters must live on an MI355X (cuda) device: the HIP path has no CPU fallback")
        lib = _lib.load()
        ns.release()
        c = self._hip_cfg
        cfg = ModelConfig(
            dim=c["dim"], depth=c["depth"], dim_head=c["dim_head"], heads=c["heads"], ff_mult=c["ff_mult"],
            wavenet_layers=c["wavenet_layers"], wavenet_stacks=c["wavenet_stacks"], dim_cond_mult=c["dim_cond_mult"],
            condition_on_prompt=int(bool(c["condition_on_prompt"])), dim_prompt=int(c["dim_prompt"] or 0),
            num_latents_m=c["num_latents_m"], resampler_depth=c["resampler_depth"], precision=_MODEL_PRECISIONS[self.precision])
        h = ctypes.c_void_p()
        check(lib.ns2_model_create(ctypes.byref(cfg), ctypes.byref(h)), "ns2_model_create")
        ns.handle = h
        keep = []
        with torch.cuda.device(dev):
            for name, p in self.state_dict().items():
                t = p.detach()
                if t.dtype != torch.float32 or not t.is_contiguous():
                    t = t.float().contiguous()
                keep.append(t)
                dims = (ctypes.c_int64 * t.ndim)(*t.shape)
                check(lib.ns2_model_set_param(h, name.encode(), t.data_ptr(), t.ndim, dims), f"set_param {name}")
            check(lib.ns2_model_finalize(h, torch.cuda.current_stream().cuda_stream), "ns2_model_finalize")
        ns.keepalive = keep           # small vectors (biases, gammas, freqs) are read in place by the executor
        ns.sig = sig
        ns.fingerprint = tensors_fingerprint(list(self.parameters()))
        ns.cond_cache = {}
        ns.calls_since_refresh = 0
        with torch.cuda.device(dev):                                  # the sampled checksum of what was just packed
            n = lib.ns2_model_param_count(h)
            ns.chk_pinned = torch.zeros(2 * n, dtype=torch.float32).pin_memory()
            check(lib.ns2_model_param_checksum(h, ns.chk_pinned.data_ptr(), 2 * n, torch.cuda.current_stream().cuda_stream), "ns2_model_param_checksum")
            torch.cuda.current_stream().synchronize()
        ns.chk_ref, ns.chk_event, ns.calls_since_chk = ns.chk_pinned.clone(), None, 0
        if self._range_guarded():          # what the device counters hold already (other models, earlier runs) is not ours
            from . import ops
            ns.sat_seen = ops.saturation_count(reset=False, device=dev)
            ns.sat_event = None
        return ns

    def _workspace(self, ns, B, N, n_prompt, n_cond):
        dev = next(self.parameters()).device
        need = _lib.load().ns2_model_workspace_bytes(ns.handle, B, N, n_prompt, n_cond)
        if ns.ws is None or ns.ws.numel() < need or ns.ws.device != dev:
            ns.ws = torch.empty(need, dtype=torch.uint8, device=dev)
        return ns.ws

    def _cond_state(self, ns, prompt, cond, drop, B, N):
        key = (prompt.data_ptr(), prompt._version, tuple(prompt.shape), cond.data_ptr(), cond._version, tuple(cond.shape),
               bool(drop), B, N)
        hit = ns.cond_cache.get(bool(drop))
        if hit is not None and hit[0] == key:
            return hit[1]
        lib = _lib.load()
        n_p, n_c = prompt.shape[1], cond.shape[2]
        nbytes = lib.ns2_model_cond_bytes(ns.handle, B, N, n_p, n_c)
        state = torch.empty(nbytes, dtype=torch.uint8, device=prompt.device)
        ws = self._workspace(ns, B, N, n_p, n_c)
        check(lib.ns2_model_prepare_cond(ns.handle, prompt.data_ptr(), n_p, cond.data_ptr(), n_c, int(drop), B, N,
                                         state.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream),
              "ns2_model_prepare_cond")
        ns.cond_cache[bool(drop)] = (key, state, prompt, cond)     # keep the inputs alive with the cache entry
        return state

    def _cond_state_cfg(self, ns, prompt, cond, B, N):
        """the conditioning of ONE 2B-utterance batch [conditioned | null substitutes] (ns2_model_cond_stack), cached like the two states"""
        sa, sb = self._cond_state(ns, prompt, cond, False, B, N), self._cond_state(ns, prompt, cond, True, B, N)
        key = (sa.data_ptr(), sb.data_ptr(), ns.cond_cache[False][0], B, N)
        hit = ns.cond_cache.get("cfg")
        if hit is not None and hit[0] == key:
            return hit[1]
        lib = _lib.load()
        n_p, n_c = prompt.shape[1], cond.shape[2]
        state = torch.empty(lib.ns2_model_cond_bytes(ns.handle, 2 * B, N, n_p, n_c), dtype=torch.uint8, device=prompt.device)
        check(lib.ns2_model_cond_stack(ns.handle, sa.data_ptr(), sb.data_ptr(), B, N, n_c, state.data_ptr(),
                                       torch.cuda.current_stream().cuda_stream), "ns2_model_cond_stack")
        ns.cond_cache["cfg"] = (key, state)
        return state

    # classifier-free guidance as one batch of 2B utterances instead of two passes of B (NS2:914-927): pays while the
    # step is bound by launches and per-block latency rather than by work -- measured (bench.py side.small_batch): b = 4 x 1024 frames
    # conditioned d512 / L12: two passes 2 x 4.4 ms, one batch of 8: 6.0 ms; from 16 utterances up two passes are as fast
    CFG_ONE_BATCH_MAX = 8

    @torch.no_grad()
    def _forward_hip_cfg(self, x, times, prompt, cond, cond_scale, cond_row=None):
        from . import ops
        dim = self._hip_cfg["dim"]
        self._guards()
        ns = self._ensure_native()
        B, N, _ = x.shape
        xin = x if (x.dtype == torch.float32 and x.is_contiguous()) else x.float().contiguous()
        pr = prompt if (prompt.dtype == torch.float32 and prompt.is_contiguous()) else prompt.float().contiguous()
        cd = cond if (cond.dtype == torch.float32 and cond.is_contiguous()) else cond.float().contiguous()
        state = self._cond_state_cfg(ns, pr, cd, B, N)
        n_c = cd.shape[2]
        ws = self._workspace(ns, 2 * B, N, pr.shape[1], n_c)
        x2 = torch.cat((xin, xin), dim=0)
        out2 = torch.empty_like(x2)
        lib = _lib.load()
        if cond_row is None:
            t = times.to(device=xin.device, dtype=torch.float32).contiguous()
            check(lib.ns2_model_forward(ns.handle, x2.data_ptr(), torch.cat((t, t)).data_ptr(), state.data_ptr(), n_c, out2.data_ptr(), 2 * B, N,
                                        ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream), "ns2_model_forward")
        else:
            check(lib.ns2_model_forward_row(ns.handle, x2.data_ptr(), cond_row.data_ptr(), state.data_ptr(), n_c, out2.data_ptr(), 2 * B, N,
                                            ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream), "ns2_model_forward_row")
        self._after_hip_forward(ns)
        out = ops.cfg_mix(out2[:B], out2[B:], cond_scale)
        return out if out.dtype == x.dtype else out.to(x.dtype)

    def clear_cond_cache(self):
        """forget the cached step-invariant conditioning (call when prompt / cond were rewritten in place through `.data`)"""
        self._native.cond_cache = {}

    # ---- step-invariant time conditioning
    @torch.no_grad()
    def time_table(self, times, batch):
        """[T, cols] conditioning table for the sampler's times (1-D tensor, shared by the batch: NS2:1303-1308): EVERY
        time-conditioning projection of the run in one pass before the loop.  `forward(..., cond_row=table[i])` then launches no
        projection in step i; for an unconditional model the result is bit-identical to `forward(x, times_i.expand(batch))`.
        Tied to the packed weights: build it after the last parameter change (the sampler does, per run)."""
        ns = self._ensure_native()
        lib = _lib.load()
        dev = next(self.parameters()).device
        t = times.to(device=dev, dtype=torch.float32).contiguous()
        cols = lib.ns2_model_table_cols(ns.handle)
        table = torch.empty(t.numel(), cols, dtype=torch.float32, device=dev)
        nws = lib.ns2_model_time_table_workspace_bytes(ns.handle, int(batch))
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            check(lib.ns2_model_time_table(ns.handle, t.data_ptr(), t.numel(), int(batch), table.data_ptr(), ws.data_ptr(), nws,
                                           torch.cuda.current_stream().cuda_stream), "ns2_model_time_table")
        return table

    # ---- forward (NS2:929-1000)
    def forward(self, x, times, prompt=None, prompt_mask=None, cond=None, cond_drop_prob=None, *, cond_row=None):
        """`cond_row` (not in the reference): a row of `time_table()` standing in for `times` (inference only)"""
        p = self.cond_drop_prob if cond_drop_prob is None else cond_drop_prob
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(q.requires_grad for q in self.parameters()))
        stochastic = self._hip_cfg["condition_on_prompt"] and p not in (0, 0., 1, 1.)
        if needs_grad or stochastic:
            # training (loss.backward(), NS2:1635/1886) and per-utterance stochastic conditioning dropout (a training /
            # validation feature, NS2:79-85) run the differentiable path: on an MI355X the HIP training kernels (training.py),
            # on the CPU the PyTorch composite; sampling never gets here
            self._native.autograd_seen = True
            return self._forward_autograd(x, times, prompt=prompt, prompt_mask=prompt_mask, cond=cond, cond_drop_prob=cond_drop_prob)
        return self._forward_hip(x, times, prompt, prompt_mask, cond, cond_drop_prob, cond_row=cond_row)

    @torch.no_grad()
    def _forward_hip(self, x, times, prompt=None, prompt_mask=None, cond=None, cond_drop_prob=None, out=None, cond_row=None):
        if prompt_mask is not None:
            raise NotImplementedError(_PROMPT_MASK_MSG)
        p = self.cond_drop_prob if cond_drop_prob is None else cond_drop_prob
        conditional = bool(self._hip_cfg["condition_on_prompt"])
        if conditional and p not in (0, 0., 1, 1.):
            raise NotImplementedError("the HIP inference path supports cond_drop_prob 0 or 1 (what forward_with_cond_scale uses)")
        dim = self._hip_cfg["dim"]
        self._guards()
        ns = self._ensure_native()
        assert x.ndim == 3 and x.shape[-1] == dim, f"x must be [b, n, {dim}]"
        B, N, _ = x.shape
        xin = x if (x.dtype == torch.float32 and x.is_contiguous()) else x.float().contiguous()
        if cond_row is None:
            t = times.to(device=xin.device, dtype=torch.float32).contiguous()
            assert t.shape == (B,)
        else:
            assert cond_row.is_cuda and cond_row.dtype == torch.float32 and cond_row.is_contiguous() and \
                cond_row.numel() == _lib.load().ns2_model_table_cols(ns.handle), "cond_row must be a row of Model.time_table()"
        out = torch.empty_like(xin) if out is None else out
        state_ptr, n_c = None, 0
        if conditional:
            assert prompt is not None and cond is not None, "conditional model needs prompt [b, n_p, dim_prompt] and cond [b, dim_prompt, n_c]"
            pr = prompt if (prompt.dtype == torch.float32 and prompt.is_contiguous()) else prompt.float().contiguous()
            cd = cond if (cond.dtype == torch.float32 and cond.is_contiguous()) else cond.float().contiguous()
            state = self._cond_state(ns, pr, cd, p == 1, B, N)
            state_ptr, n_c = state.data_ptr(), cd.shape[2]
            ws = self._workspace(ns, B, N, pr.shape[1], n_c)
        else:
            ws = self._workspace(ns, B, N, 0, 0)
        if cond_row is None:
            check(_lib.load().ns2_model_forward(ns.handle, xin.data_ptr(), t.data_ptr(), state_ptr, n_c, out.data_ptr(), B, N,
                                                ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream), "ns2_model_forward")
        else:
            check(_lib.load().ns2_model_forward_row(ns.handle, xin.data_ptr(), cond_row.data_ptr(), state_ptr, n_c, out.data_ptr(), B, N,
                                                    ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream),
                  "ns2_model_forward_row")
        self._after_hip_forward(ns)
        return out if out.dtype == x.dtype else out.to(x.dtype)

    def _after_hip_forward(self, ns):
        if self._range_guarded() and not torch.cuda.is_current_stream_capturing():
            ns.calls_since_peek += 1
            if ns.sat_event is None and (ns.peek_now or ns.calls_since_peek >= self.SAT_PEEK_EVERY):
                self._peek_saturation()
                ns.peek_now = False
        ns.last_call = time.monotonic()

    def forward_with_cond_scale(self, *args, cond_scale=1., **kwargs):
        """NS2:914-927."""
        if (cond_scale != 1. and self._hip_cfg["condition_on_prompt"] and not torch.is_grad_enabled() and len(args) == 2 and args[0].is_cuda
                and args[0].shape[0] <= self.CFG_ONE_BATCH_MAX and kwargs.get("prompt") is not None and kwargs.get("cond") is not None
                and kwargs.get("prompt_mask") is None and set(kwargs) <= {"prompt", "cond", "cond_row", "prompt_mask"}
                and not torch.cuda.is_current_stream_capturing()):
            return self._forward_hip_cfg(args[0], args[1], kwargs["prompt"], kwargs["cond"], cond_scale, cond_row=kwargs.get("cond_row"))
        logits = self.forward(*args, cond_drop_prob=0., **kwargs)
        if cond_scale == 1.:
            return logits
        if not self._hip_cfg["condition_on_prompt"] and not torch.is_grad_enabled():
            # an unconditional Model ignores cond_drop_prob, so the reference's second forward reproduces `logits` bit for
            # bit and null + (logits - null) * s == logits exactly: skip it (deterministic kernels, see
            # test_batch_independence_and_determinism)
            return logits
        null_logits = self.forward(*args, cond_drop_prob=1., **kwargs)
        if logits.is_cuda and not torch.is_grad_enabled():
            from . import ops
            return ops.cfg_mix(logits, null_logits, cond_scale)
        return null_logits + (logits - null_logits) * cond_scale

    # ---- parity-test taps
    def debug_forward(self, x, times, taps, prompt=None, cond=None, drop=False):
        """forward + fp32 copies of intermediate buffers; taps: {name: numel}. Test-only helper."""
        ns = self._ensure_native()
        lib = _lib.load()
        dev = next(self.parameters()).device
        bufs = {k: torch.zeros(n, dtype=torch.float32, device=dev) for k, n in taps.items()}
        for k, b in bufs.items():
            check(lib.ns2_model_debug_tap(ns.handle, k.encode(), b.data_ptr(), b.numel()), "debug_tap")
        try:
            ns.cond_cache = {}
            out = self._forward_hip(x, times, prompt=prompt, cond=cond, cond_drop_prob=1. if drop else 0.)
            torch.cuda.synchronize()
        finally:
            for k in bufs:
                lib.ns2_model_debug_tap(ns.handle, k.encode(), None, 0)
        return out, bufs


# `prompt_mask` is in the reference's signature (NS2:929-937) but upstream cannot run with one: the resampler's attention
# concatenates its 32 latents in front of the prompt keys (cross_attn_include_queries, NS2:1060-1061) and then applies the
# [b, n_p] mask to [b, h, 32, 32 + n_p] scores -- a shape error on both of Attend's paths (ATT:92-94, 136-138; verified by
# execution, tests/test_compat_reference.py).  No upstream caller passes it (NS2:1333, 1410, 1635).  Every path of this package
# rejects it the same way instead of guessing a semantics the reference never had.
_PROMPT_MASK_MSG = ("prompt_mask: the reference itself raises on it (the PerceiverResampler applies the [b, n_p] mask to scores over "
                    "32 + n_p keys, NS2:962-968 / 1060-1061 / ATT:92-94) and no reference caller passes one")


def _train_forward(m, x, times, prompt, cond, cond_drop_prob, prompt_mask=None):
    import os
    from . import training
    if prompt_mask is not None:
        raise NotImplementedError(_PROMPT_MASK_MSG)
    which = os.environ.get("NS2_TRAIN_BACKEND") or getattr(m, "train_backend", "hip")
    if which == "hip" and training.available(next(m.parameters()).device):
        why = training.unsupported_reason(m)
        if why is None:
            return training.model_forward_train(m, x, times, prompt=prompt, cond=cond, cond_drop_prob=cond_drop_prob)
        import warnings
        warnings.warn(f"HIP training path not usable for this Model ({why}): running the PyTorch composite")
    from .autograd_path import model_forward_autograd
    return model_forward_autograd(m, x, times, prompt=prompt, cond=cond, cond_drop_prob=cond_drop_prob)


# ------------------------------------------------------------------------------------------ Model
class Model(HipDenoiserMixin, nn.Module):
    def __init__(
        self,
        dim,
        *,
        depth,
        dim_head=64,
        heads=8,
        ff_mult=4,
        wavenet_layers=8,
        wavenet_stacks=4,
        dim_cond_mult=4,
        use_flash_attn=True,
        dim_prompt=None,
        num_latents_m=32,
        resampler_depth=2,
        cond_drop_prob=0.,
        condition_on_prompt=False,
        precision="exact",
    ):
        super().__init__()
        self.dim = dim
        self.depth = depth
        self.dim_head, self.heads, self.ff_mult = dim_head, heads, ff_mult
        self.wavenet_layers, self.wavenet_stacks, self.dim_cond_mult_base = wavenet_layers, wavenet_stacks, dim_cond_mult
        self.use_flash_attn = use_flash_attn          # accepted for signature parity; the HIP attention is always fused
        self.dim_prompt = dim_prompt
        self.num_latents_m, self.resampler_depth = num_latents_m, resampler_depth
        self.cond_drop_prob = cond_drop_prob
        self.condition_on_prompt = condition_on_prompt

        dim_time = dim * dim_cond_mult
        self.to_time_cond = _seq(_SinusoidalWeights(dim), nn.Linear(dim + 1, dim_time), _NoParams())

        self.to_prompt_cond = None
        self.null_cond = None
        self.cond_to_model_dim = None
        if condition_on_prompt:
            assert dim_prompt is not None, "dim_prompt is required when condition_on_prompt=True"
            self.null_prompt_cond = nn.Parameter(torch.randn(dim_time))
            self.null_prompt_tokens = nn.Parameter(torch.randn(num_latents_m, dim))
            nn.init.normal_(self.null_prompt_cond, std=0.02)
            nn.init.normal_(self.null_prompt_tokens, std=0.02)
            self.to_prompt_cond = _seq(_NoParams(), nn.Linear(dim_prompt, dim_time), _NoParams())
            pr = _Holder()
            if dim_prompt != dim:
                pr.proj_context = nn.Linear(dim_prompt, dim)
            pr.latents = nn.Parameter(torch.randn(num_latents_m, dim))
            nn.init.normal_(pr.latents, std=0.02)
            pr.layers = nn.ModuleList([
                nn.ModuleList([_Attention(dim, dim_head, heads), _feedforward(dim, ff_mult, False)])
                for _ in range(resampler_depth)])
            pr.norm = _RMSNorm(dim)
            self.perceiver_resampler = pr
            self.cond_to_model_dim = nn.Conv1d(dim_prompt, dim, 1)
            self.null_cond = nn.Parameter(torch.zeros(dim, 1))

        cm = dim_cond_mult * (2 if condition_on_prompt else 1)      # NS2:884
        wn = _Holder()
        wn.init_conv = nn.Conv1d(dim, dim, 3)
        wn.stacks = nn.ModuleList()
        for s in range(wavenet_stacks):
            st = _Holder()
            st.blocks = nn.ModuleList([
                _WavenetBlock(dim, 2 ** i, skip=(s == wavenet_stacks - 1), dim_cond_mult=cm) for i in range(wavenet_layers)])
            wn.stacks.append(st)
        wn.final_conv = nn.Conv1d(dim, dim, 1)
        self.wavenet = wn

        tr = _Holder()
        tr.layers = nn.ModuleList()
        for _ in range(depth):
            layer = _Holder()
            mods = [
                _RMSNorm(dim, scale=False, dim_cond=dim * cm),
                _Attention(dim, dim_head, heads),
                _RMSNorm(dim, scale=False, dim_cond=dim * cm) if condition_on_prompt else None,
                _Attention(dim, dim_head, heads) if condition_on_prompt else None,
                _RMSNorm(dim, scale=False, dim_cond=dim * cm),
                _feedforward(dim, ff_mult, True),
            ]
            for i, mod in enumerate(mods):                          # reference mlist keeps the None slots' indices
                if mod is not None:
                    layer.add_module(str(i), mod)
            tr.layers.append(layer)
        tr.to_pred = _seq(_RMSNorm(dim), nn.Linear(dim, dim, bias=False))
        self.transformer = tr

        self.train_backend = "hip"                    # "composite": the PyTorch composite also on the GPU (A/B, tests)
        self.train_precision = "exact"                # "mixed": half product + fp8 correction terms under a loss scale (training.py `_Scale`)
        self._hip_init(dict(dim=dim, depth=depth, dim_head=dim_head, heads=heads, ff_mult=ff_mult, wavenet_layers=wavenet_layers,
                            wavenet_stacks=wavenet_stacks, dim_cond_mult=dim_cond_mult, condition_on_prompt=condition_on_prompt,
                            dim_prompt=dim_prompt, num_latents_m=num_latents_m, resampler_depth=resampler_depth), precision)

    # ------------------------------------------------------------------ reference attribute surface
    @property
    def device(self):
        return next(self.parameters()).device

    def _forward_autograd(self, x, times, prompt=None, prompt_mask=None, cond=None, cond_drop_prob=None):
        """forward under autograd (loss.backward(), NS2:1635 / 1886).  Parameters on an MI355X: the HIP training path
        (training.py: forward and backward kernels of libns2hip behind torch.autograd.Functions).  `train_backend = "composite"`
        (or NS2_TRAIN_BACKEND=composite), and parameters on the CPU (BASELINE config 1 as the reference runs it): the PyTorch
        composite of autograd_path.py."""
        return _train_forward(self, x, times, prompt, cond, cond_drop_prob, prompt_mask=prompt_mask)
