#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): denoise steps/sec for Model(dim=512, depth=12) on 32 x 1024 codec-latent frames.

One step = Model.forward_with_cond_scale (cond_scale 1 -> one Model.forward, NS2:914-927) + the DDIM update
(NS2:1396-1430) on one batch of 32 synthetic utterances, inputs resident in HBM.  N GPUs = N data-parallel
replicas each denoising its own 32-utterance shard (weak scaling; no data-path collective inside the loop; the
path's single RCCL all-gather of the generated latents runs once after the K steps and is inside the timed region).

    python bench.py [--gpus N --steps K --warmup W] [--precision hybrid|mixed|half|exact|fast] [--graph]

`--gpus N` with N > 1 re-executes itself under `python -m torch.distributed.run --nproc-per-node N` (one rank per GPU,
backend "nccl" = RCCL) unless it already runs under one (RANK / WORLD_SIZE set by the driver's own launcher).

Default precision is "hybrid", a per-site plan: every contraction is one IEEE-half product plus BOTH first-order
correction terms evaluated in one block-scaled fp8 MFMA per 32-deep k block ("mixed"), except the
feed-forward causal conv (43 % of the FLOPs), which runs as the half product alone -- 6e-5-class error against the fp32
reference (>10x inside the 1e-3 tolerance of BASELINE.json; sweep in tests/test_parity_r2_gpu.py, quoted in `parity`).
The same JSON line reports, from short extra measurements on the same weights, `mixed_mode` (correction terms
everywhere), `half_mode` (the half product alone everywhere: faster, error 8e-4 = a thin margin) and `exact_mode`
(bf16 x3, 1e-5), and `side` = the other BASELINE configs (RVQ config 4, conditioned config 3, d128 config 2), each with
its own parity flag and roofline.

Prints ONE JSON line on rank 0.  `roofline` is measured live with HIP events recorded by the executor on the launch stream
around every launch of the dominant kernel symbol (hybrid: gemm2_kernel<1, EPI_SPLIT, true> = the 12 FF causal convs;
other modes: gemm2_kernel<*, EPI_SPLIT, *> = the 12 FF causal convs + wavenet init conv + skip GEMM); `cpu_baseline` times the
reference's OWN `Model.forward` (unmodified source from the oracle/_ref archive, CPU SDPA attention, the benched weights) on the host
cores: two timed forwards of the full batch of 32 after a warm-up, the minimum reported (kind "port" = the oracle restatement, only
when the archive is absent).  Round 4: the steps read the run's time-conditioning table (built inside the timed region, as
NaturalSpeech2.sample builds it per run; --no-time-table for the A/B), `side.train_step` = a warm training step on the HIP backward.
See DESIGN.md §Measurement.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PMC_TRAFFIC = "r06_pmc_traffic.json"  # profiles/: committed PMC profile of the dominant kernel (tools/pmc_mfma_bench.sh, r06_pmc_mfma.json)
PARITY_RECORD = "r06_parity.json"    # profiles/: the tracked key-wise parity record (tests/parity_record.py)
PEAK_16BIT_TFLOPS = 2500.0         # dense bf16 / f16 MFMA peak, MI355X_MICROARCH.md (spec; 2495 measured)
PEAK_F32_MFMA_TFLOPS = 157.3       # v_mfma_f32_32x32x2_f32 (RVQ)
UTT_GFLOP = {(512, 12, False): 316.37, (512, 12, True): 331.98, (128, 6, False): 26.74}   # per 1024-frame utterance, SURVEY §8d
MFMA_UNITS = {"exact": 3, "mixed": 2, "half": 1, "fast": 1, "hybrid": 1, "hybrid_ff": 1}   # MFMA-pipe time per algorithmic FLOP of the dominant kernel, in 16-bit-product units
KERNEL_NAME = {"exact": "gemm2_kernel<3, 1, false, 0>", "mixed": "gemm2_kernel<2, 1, true, 0>",
               "half": "ffc3::ffconv3_kernel<PF_F16> + gemm2_kernel<1, 1, true, 0>",
               "fast": "gemm2_kernel<1, 1, false, 0>", "hybrid": "ffc3::ffconv3_kernel<PF_H8>", "hybrid_ff": "ffc3::ffconv3_kernel<PF_F16>"}
# gemm2_kernel<NSPLIT, EPI_SPLIT = 1, F16, P1> (csrc/gemm2.hip); ffconv3_kernel = the dedicated FF causal conv kernel of the plans whose conv
# is one IEEE-half product (csrc/ffconv_kernel.h, round 6; NS2_GEMM=4 switches it off: the conv then runs on gemm2_kernel<1, 1, true, 0>)
DTYPE = {"exact": "bf16x3 split operands on the bf16 MFMA, fp32 accumulate",
         "mixed": "fp16 operands on the f16 MFMA + both first-order correction terms as e5m2 on the block-scaled fp8 MFMA, fp32 accumulate",
         "hybrid": "fp16 operands on the f16 MFMA + both first-order correction terms as e5m2 on the block-scaled fp8 MFMA "
                   "(FF causal conv: the fp16 product alone), fp32 accumulate",
         "hybrid_ff": "fp16 operands on the f16 MFMA + both first-order correction terms as e5m2 on the block-scaled fp8 MFMA "
                      "(whole feed-forward branch and the Wavenet's dilated convs: the fp16 product alone), fp32 accumulate",
         "half": "fp16 operands on the f16 MFMA (one product), fp32 accumulate",
         "fast": "bf16 operands, fp32 accumulate"}


def dominant_flops(B, N, dim, depth, ff_mult, wn_layers, conv_only=False, conditioned=False):
    """algorithmic FLOPs (2*MAC) of the launches of gemm2_kernel<*, EPI_SPLIT, *> in one step, and their count
    (conv_only: the FF causal convs alone, which in the hybrid plan are the only launches of their kernel symbol)."""
    M = B * N
    f = int(dim * ff_mult * 2 / 3)
    ffconv = 2.0 * M * f * (3 * f)                 # CausalConv1d(f, f, 3)  NS2:1016
    init = 2.0 * M * dim * (3 * dim)               # wavenet.init_conv      NS2:701
    skip = 2.0 * M * dim * (wn_layers * dim)       # 8 skip convs summed    NS2:639-640, 725
    if conv_only:
        return depth * ffconv, depth
    if conditioned:                                # + the cross-attention query projections (same kernel symbol, NS2:1051)
        return depth * (ffconv + 2.0 * M * dim * dim) + init + skip, 2 * depth + 2
    return depth * ffconv + init + skip, depth + 2


def train_step_side(dev):
    """warm training step (loss + backward + Adam) of BASELINE config 1's shape and of the headline shape on the HIP training path,
    in both training arithmetics (exact = bf16 x3; mixed = half product + fp8 correction terms under a loss scale), with the PyTorch
    composite beside it; every backend runs the same 2 warm-up + `iters` steps, so the reported losses are the SAME iteration"""
    import torch
    from naturalspeech2_pytorch_amd import Model, NaturalSpeech2
    out = {}
    for tag, kw, b, n, iters in (("config1_d128_L6_b4", dict(dim=128, depth=6), 4, 1024, 8), ("headline_d512_L12_b32", dict(dim=512, depth=12), 32, 1024, 6)):
        res = {}
        for name, backend, tprec, k in (("mixed", "hip", "mixed", iters), ("exact", "hip", "exact", iters), ("composite", "composite", "exact", 2)):   # k timed steps (two windows)
            torch.manual_seed(0)
            m = Model(**kw).to(dev).train()
            m.train_backend, m.train_precision = backend, tprec
            d = NaturalSpeech2(m, codec=None, target_sample_hz=24000).to(dev)
            opt = torch.optim.Adam(m.parameters(), lr=1e-4, fused=True)      # PyTorch's single-pass Adam, the same for every backend
            g = torch.Generator().manual_seed(1)
            audio, times, noise = torch.randn(b, n, kw["dim"], generator=g).to(dev), torch.rand(b, generator=g).to(dev), torch.randn(b, n, kw["dim"], generator=g).to(dev)

            def step():
                opt.zero_grad(set_to_none=True)
                loss = d(audio, times=times, noise=noise)
                loss.backward()
                opt.step()
                return loss

            for _ in range(2):
                step()

            def window(fn, kw_):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(kw_):
                    last = fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / kw_, last

            # Round 6: the HIP backends' loss + backward is replayed from ONE HIP graph (training.GraphedTrainStep: same
            # kernels, gradients bit-identical -- tests/test_round6_gpu.py::test_graphed_training_step); fused Adam stays outside it.  The
            # eager pass (~2000 launches enqueued from Python, at the mercy of the host's other tenants) is timed beside it for 2 steps.
            eager_dt = None
            done = 0
            if backend == "hip":
                from naturalspeech2_pytorch_amd import training
                eager_dt, loss = window(step, 2)
                done = 2
                gs = training.GraphedTrainStep(lambda a, t_, z: d(a, times=t_, noise=z), (audio, times, noise), m)

                def step():                                   # noqa: F811
                    loss = gs(audio, times, noise)
                    opt.step()
                    return loss
            wins = []
            for kw_ in ((k - done + 1) // 2, (k - done) // 2):
                if kw_ <= 0:
                    continue
                w, loss = window(step, kw_)
                wins.append(w)
                done += kw_
            dt = sum(wins) / len(wins)                   # the mean of the windows (no best-of)
            for _ in range(iters - done):                # (untimed) so that every backend reports the loss of the SAME iteration
                loss = step()
            res[name + "_windows"] = [round(1e3 * w, 2) for w in wins]
            res[name + "_eager"] = eager_dt
            res[name] = (dt, float(loss.detach()))
            gs = None
            del m, d, opt, gs
            torch.cuda.empty_cache()
        flops = 3.0 * UTT_GFLOP[(kw["dim"], kw["depth"], False)] * 1e9 * b * n / 1024        # forward + dgrad + wgrad
        best = min(("mixed", "exact"), key=lambda q: res[q][0])
        ms = 1e3 * res[best][0]
        out[tag] = dict(metric=f"warm training step (loss + backward + fused Adam), Model(dim={kw['dim']}, depth={kw['depth']}), {b} x {n} frames",
                        ms_per_step=round(ms, 2), train_precision=best, iterations=iters, steps_per_s=round(1e3 / ms, 3),
                        mixed_ms_per_step=round(1e3 * res["mixed"][0], 2), exact_ms_per_step=round(1e3 * res["exact"][0], 2),
                        launch_path="loss + backward replayed from one HIP graph (training.GraphedTrainStep), fused Adam outside it",
                        timed_windows_ms={q: res[q + "_windows"] for q in ("mixed", "exact")},
                        eager_ms_per_step={q: round(1e3 * res[q + "_eager"], 2) for q in ("mixed", "exact")},
                        algorithmic_tflops=round(flops / (ms * 1e-3) / 1e12, 1), algorithmic_flops="3 x the forward's (SURVEY 8d)",
                        frac_of_16bit_peak=round(flops / (ms * 1e-3) / 1e12 / PEAK_16BIT_TFLOPS, 4),
                        arithmetic={"mixed": "IEEE-half product + both correction terms on the fp8 MFMA (2 MFMA units per algorithmic FLOP) on FMT_H8 "
                                             "operands under a power-of-two loss scale chosen on the device; attention bf16 x3; fp32 accumulate, fp32 master weights",
                                    "exact": "bf16 x3 split operands (3 MFMA units per algorithmic FLOP), fp32 accumulate, fp32 master weights"},
                        pytorch_composite_ms_per_step=round(1e3 * res["composite"][0], 2),
                        speedup_vs_pytorch_composite={"exact (same fp32-class arithmetic as the composite)": round(res["composite"][0] / res["exact"][0], 2),
                                                      "mixed (gradients within 1e-3 of the reference's)": round(res["composite"][0] / res["mixed"][0], 2)},
                        loss_mixed=res["mixed"][1], loss_exact=res["exact"][1], loss_composite=res["composite"][1],
                        loss_iteration=f"all after 2 warm-up + {iters} Adam steps from the same init", timing="mean of the two timed windows of graph replays (timed_windows_ms); the composite: 2 steps",
                        parity="every parameter's .grad vs the reference's own autograd in both arithmetics: tests/test_backward_gpu.py, "
                               "tests/test_round5_gpu.py, profiles/r06_parity.json keys backward_vs_reference_autograd/*")
    return out


def physical_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return os.cpu_count() or 1


_REF_FULL = {}      # cpu_baseline keeps (x, t, the reference's full-batch output) for the parity check at the benched shape


def cpu_baseline(sd, dim, depth, B, n):
    """The reference timed on the host cores of this box, beside the GPU number (never the target): ONE forward of the full batch
    after a one-utterance warm-up, fp32, threads = physical cores.
    kind "reference": the reference's OWN `Model` (NS2:811-1000, default use_flash_attn=True -> CPU SDPA, ATT:98-108), imported
    unmodified from oracle/_ref/reference_py.tar.gz (oracle/make_ref.py; /root/reference itself does not exist on the GPU box) with
    the benched weights loaded into it.  kind "port" (only when that archive is absent): the oracle restatement, SDPA attention."""
    import torch
    from oracle import ns2_oracle as O
    from oracle import ref_stub
    cores = physical_cores()
    old = torch.get_num_threads()
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, n, dim, generator=g)
    t = torch.rand(B, generator=g)
    try:
        if ref_stub.reference_available():
            import contextlib
            ns2 = ref_stub.load_reference()
            with contextlib.redirect_stdout(sys.stderr):          # ATT:60-67 print()s its SDPA backend choice: keep stdout = the one JSON line
                ref = ns2.Model(dim=dim, depth=depth).eval()
            ref.load_state_dict(sd)
            with torch.no_grad(), contextlib.redirect_stdout(sys.stderr):
                ref(x[:1], t[:1])                         # warm-up (thread pool, oneDNN primitives)
                els = []
                for _ in range(2):                        # >= 2 timed steps; the minimum is reported
                    t0 = time.perf_counter()
                    y = ref(x, t)
                    els.append(time.perf_counter() - t0)
                el = min(els)
                chk = float(((O.model_forward(sd, x[:1], t[:1]) - y[:1]).norm() / y[:1].norm()).item())   # the oracle against the reference, live
            _REF_FULL.update(x=x, t=t, y=y.detach())
            return dict(value=round(1.0 / el, 5), unit="steps/s", cores=cores, kind="reference",
                        sample=f"the reference's own Model.forward (unmodified source, {ref_stub.reference_source()}), fp32, CPU SDPA "
                               f"attention, TWO timed forwards of the full batch {B} x {n} frames after a 1-utterance warm-up: "
                               f"{els[0]:.2f} s, {els[1]:.2f} s (value = 1 / min), {cores} threads = physical cores; "
                               f"oracle-vs-reference rel err on one utterance {chk:.1e}")
        O.USE_SDPA = True
        with torch.no_grad():
            O.model_forward(sd, x[:1], t[:1])
            t0 = time.perf_counter()
            O.model_forward(sd, x, t)
            el = time.perf_counter() - t0
    finally:
        O.USE_SDPA = False
        torch.set_num_threads(old)
    return dict(value=round(1.0 / el, 5), unit="steps/s", cores=cores, kind="port",
                sample=f"oracle Model.forward fp32 (SDPA attention), ONE forward of the full batch {B} x {n} frames after a "
                       f"1-utterance warm-up: {el:.2f} s, {cores} threads = physical cores (oracle/_ref archive absent)")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", default="hybrid", choices=["hybrid", "hybrid_ff", "mixed", "half", "exact", "fast"],
                    help="hybrid (default): half product + fp8 correction terms, FF causal conv half only, ~6e-5 from the fp32 "
                         "reference; mixed: correction terms everywhere, ~5e-5; half: one fp16 product, ~8e-4 (thin margin); "
                         "exact: bf16x3 split, 1e-5; fast: bf16, ~1e-2 (outside the tolerance)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short mixed/half/exact-mode measurements")
    ap.add_argument("--no-side", action="store_true", help="skip the side workloads (RVQ config 4, conditioned config 3, d128 config 2)")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--depth", type=int, default=12)
    ap.add_argument("--graph", action="store_true", help="replay the step from a HIP graph (no live per-kernel events)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-time-table", action="store_true",
                    help="recompute the time-conditioning projections inside every step (rounds 1-3) instead of reading the run's table (A/B)")
    ap.add_argument("--no-parity", action="store_true", help="skip the live parity forward (profile runs: keeps the kernel trace to the timed workload)")
    ap.add_argument("--conditioned", action="store_true",
                    help="BASELINE config 3 instead of the headline: dim_prompt=512, condition_on_prompt, prompt of 103 codec "
                         "frames, frame-aligned cond")
    ap.add_argument("--cond-scale", type=float, default=1.0, help="with --conditioned: classifier-free guidance scale (NS2:914-927)")
    args = ap.parse_args()

    # ---- N > 1 without a launcher: become one (the driver may call `python bench.py --gpus N` directly)
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd, env=env).returncode)

    import torch
    from naturalspeech2_pytorch_amd import Model, _lib, ops
    from naturalspeech2_pytorch_amd import distributed as D
    import torch.distributed as dist

    rank, local, world = D.init_from_env()
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (the HIP path has no CPU fallback)"
    local_dev = local % torch.cuda.device_count()       # ranks > devices only in the gloo functional test of this script
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    lib = _lib.load()
    B, N = args.batch, args.frames

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    COLL = {}            # --gpus N: the `collective` block of the line (distributed.collective_report)

    def make_model(dim, depth, conditioned):
        torch.manual_seed(1234)                          # same random-init weights on every rank
        mkw = dict(dim_prompt=512, condition_on_prompt=True) if conditioned else {}
        return Model(dim=dim, depth=depth, **mkw).to(dev).eval()

    def measure(model, precision, steps, warmup, conditioned=False, cond_scale=1.0, graph=False, profile=True):
        """W untimed + exactly K timed steps; returns (elapsed s of the K steps [max over ranks], kernel ms, launches)"""
        model.precision = precision                      # re-packs the same parameters in the other plane format
        dim = model.dim
        g = torch.Generator().manual_seed(100 + rank)
        audio = torch.randn(B, N, dim, generator=g).to(dev)
        fwd_kw = {}
        if conditioned:
            fwd_kw = dict(prompt=torch.randn(B, 103, 512, generator=g).to(dev), cond=torch.randn(B, 512, N, generator=g).to(dev))
        n_total = warmup + steps
        ts = torch.linspace(1.0, 0.0, n_total + 1)
        t_dev = [ts[i].expand(B).contiguous().to(dev) for i in range(n_total + 1)]
        t_cur, t_nxt = t_dev[0].clone(), t_dev[1].clone()

        # SURVEY §8f-1 as NaturalSpeech2.sample runs it: the time-conditioning projections of a run's steps are ONE table built at the
        # start of the run (Model.time_table); step i reads row i.  The table of the K timed steps is built INSIDE the timed region.
        use_table = not args.no_time_table
        row_buf = torch.empty(lib.ns2_model_table_cols(model._ensure_native().handle), device=dev) if (use_table and graph) else None

        def step(row=None):
            kw = dict(fwd_kw) if row is None else dict(fwd_kw, cond_row=row)
            out = model.forward_with_cond_scale(audio, t_cur, cond_scale=cond_scale, **kw)
            ops.ddim_step(audio, out, t_cur, t_nxt, "v", "sigmoid", 1.0, out=audio)

        with torch.no_grad():
            cg = None
            tab_w = model.time_table(ts[:max(warmup, 1)].to(dev), B) if use_table else None
            for i in range(warmup):
                t_cur.copy_(t_dev[i]); t_nxt.copy_(t_dev[i + 1])
                step(tab_w[i] if use_table else None)
            if graph:
                torch.cuda.synchronize()
                cg = torch.cuda.CUDAGraph()
                keep = audio.clone()
                if use_table:
                    row_buf.copy_(tab_w[0])
                with torch.cuda.graph(cg):
                    step(row_buf)
                audio.copy_(keep)
            ns = model._ensure_native()
            # the FF causal convs (bit 7); in the other modes init conv + skip GEMM share their kernel symbol (bit 1)
            prof_mask = ((1 << 7) | (0 if precision in ("hybrid", "hybrid_ff") else (1 << 1))) if (profile and not graph) else 0
            barrier()
            if prof_mask:
                lib.ns2_model_profile_begin(ns.handle, prof_mask)
            t0 = time.perf_counter()
            tab = model.time_table(ts[warmup:n_total].to(dev), B) if use_table else None
            for i in range(warmup, n_total):
                t_cur.copy_(t_dev[i]); t_nxt.copy_(t_dev[i + 1])
                if cg is not None:
                    if use_table:
                        row_buf.copy_(tab[i - warmup])
                    cg.replay()
                else:
                    step(tab[i - warmup] if use_table else None)
            if world > 1:                                # the sharded sampler's single collective
                src = audio if dist.get_backend() != "gloo" else audio.cpu()     # gloo (functional test only) gathers on the host
                gathered = torch.empty((world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)   # rank r's shard at rows [r B, (r + 1) B): the form gloo and nccl both take
                dist.all_gather_into_tensor(gathered, src)       # single-buffer form: the collective row measures the wire, not copies
            barrier()
            elapsed = time.perf_counter() - t0
            kern_ms, kern_n = ctypes.c_double(0), ctypes.c_int64(0)
            if prof_mask:
                _lib.check(lib.ns2_model_profile_end(ns.handle, ctypes.byref(kern_ms), ctypes.byref(kern_n)), "profile_end")
        el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
            if profile:        # the headline measurement: what the job's one collective costs, apart from the loop (every rank calls it)
                COLL.update(D.collective_report(audio, 1e3 * elapsed / steps))
        assert torch.isfinite(audio).all()
        return el.item(), kern_ms.value, kern_n.value

    def live_parity(model, sd_cpu, precision, conditioned=False):
        """one small forward of this very model (same packed weights) against the CPU oracle"""
        from oracle import ns2_oracle as O
        model.precision = precision
        g = torch.Generator().manual_seed(5)
        b, n = 1, 256
        x, t = torch.randn(b, n, model.dim, generator=g), torch.rand(b, generator=g)
        kw, okw = {}, {}
        if conditioned:
            p, c = torch.randn(b, 40, 512, generator=g), torch.randn(b, 512, n, generator=g)
            kw, okw = dict(prompt=p.to(dev), cond=c.to(dev)), dict(prompt=p, cond=c)
        with torch.no_grad():
            y = model(x.to(dev), t.to(dev), **kw).cpu()
            ref = O.model_forward(sd_cpu, x, t, **okw)
        return float(((y - ref).double().norm() / ref.double().norm()).item())

    def roofline_obj(precision, dim, depth, kern_ms_v, kern_n_v, conditioned=False):
        if not kern_n_v:
            return None
        fl, nl = dominant_flops(B, N, dim, depth, 4, 8, conv_only=(precision in ("hybrid", "hybrid_ff")), conditioned=conditioned)
        avg_ms = kern_ms_v / kern_n_v
        ach = (fl / nl) / (avg_ms * 1e-3) / 1e12
        traffic, tsrc, busy = None, None, None
        pj = os.path.join(ROOT, "profiles", PMC_TRAFFIC)
        if os.path.exists(pj) and (dim, depth) == (512, 12):
            tj = json.load(open(pj))
            traffic = tj.get("hbm_bytes_per_launch_by_precision", {}).get(precision)
            if traffic is None and tj.get("precision") == precision:         # the raw output of tools/pmc_bench.sh (one precision)
                traffic = tj.get("hbm_bytes_per_launch")
            if traffic is not None:
                tsrc = (f"profiles/{PMC_TRAFFIC}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same command "
                        "(tools/pmc_mfma_bench.sh), read side doubled per MI355X_MICROARCH.md; a committed profile, NOT measured in this run")
                busy = tj.get("mfma_pipe_busy_frac")
        what = f"FF causal conv k3 x{depth}" + ("" if precision in ("hybrid", "hybrid_ff") else ", wavenet init conv, skip-sum GEMM") + \
               (f", cross-attention q projection x{depth}" if conditioned and precision not in ("hybrid", "hybrid_ff") else "")
        return dict(bound="mfma", kernel=f"ns2::{KERNEL_NAME[precision]} ({what})",
                    achieved=round(ach, 2), peak=PEAK_16BIT_TFLOPS, unit="TFLOP/s", frac=round(ach / PEAK_16BIT_TFLOPS, 4),
                    traffic=traffic, traffic_source=tsrc, mfma_pipe_busy_frac_profiled=busy, avg_launch_ms=round(avg_ms, 4), launches=kern_n_v,
                    algorithmic_gflop_per_launch=round(fl / nl / 1e9, 2),
                    mfma_time_units_per_algorithmic_flop=MFMA_UNITS[precision],
                    frac_of_mfma_pipe_time=round(MFMA_UNITS[precision] * ach / PEAK_16BIT_TFLOPS, 4))

    # ================================================================== headline
    dim, depth = args.dim, args.depth
    model = make_model(dim, depth, args.conditioned)
    elapsed, kern_ms_v, kern_n_v = measure(model, args.precision, args.steps, args.warmup, args.conditioned,
                                           args.cond_scale if args.conditioned else 1.0, args.graph)
    steps_per_s = world * args.steps / elapsed
    extra = {}
    if rank == 0 and world == 1 and not args.graph:
        sd_cpu = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        parity = {"live_rel_err_vs_fp32_oracle": {} if args.no_parity else
                  {args.precision: live_parity(model, sd_cpu, args.precision, args.conditioned)}, "tolerance": 1e-3}
        # committed sweep maxima (tests/test_parity_r2_gpu.py via tests/parity_record.py).  A record without the sweep keys is a
        # broken evidence trail: fail loudly instead of quoting an empty sweep.
        pj = os.path.join(ROOT, "profiles", PARITY_RECORD)
        sw = json.load(open(pj))
        missing = [k for k in ("sweep_d512_L12/hybrid", "sweep_d512_L12/mixed", "sweep_d512_L12/half") if k not in sw]
        if missing:
            raise RuntimeError(f"{pj} lacks {missing}: regenerate it from a full `pytest -m gpu` run + tools/merge_parity.py")
        parity["sweep_d512_L12"] = {k.split("/")[-1]: dict(max=v["max"], mean=v["mean"]) for k, v in sw.items()
                                    if k.startswith("sweep_d512_L12/")}
        parity["sweep_source"] = f"profiles/{PARITY_RECORD} (tests/test_parity_r2_gpu.py: 8 weight seeds x times {{0.002, 0.5, 0.999}})"
        if "stress_d512_L12_b2x1024" in sw:              # the same architecture with outlier channels / heavy-tailed weights (committed record)
            parity["stress_d512_L12_b2x1024"] = dict(sw["stress_d512_L12_b2x1024"], source="tests/test_parity_r2_gpu.py::test_precision_plans_with_outlier_channels_and_heavy_tails")
        if not args.no_secondary and not args.conditioned:
            k2 = min(args.steps, 10)
            for other in ("hybrid_ff", "mixed", "half", "exact"):
                if other == args.precision:
                    continue
                e2, km, kn = measure(model, other, k2, 2)
                parity["live_rel_err_vs_fp32_oracle"][other] = live_parity(model, sd_cpu, other)
                r2 = roofline_obj(other, dim, depth, km, kn)
                extra[other + "_mode"] = dict(value=round(k2 / e2, 3), unit="steps/s", steps=k2, ms_per_step=round(1e3 * e2 / k2, 3),
                                              dtype=DTYPE[other], roofline_frac=r2["frac"] if r2 else None,
                                              dominant_kernel_tflops=r2["achieved"] if r2 else None)
                if other == "hybrid_ff":
                    extra[other + "_mode"]["note"] = ("not the default: 1.6e-4 on the random-init sweep but 1.0e-3 with an amplified feed-forward "
                                                      "branch (hybrid: 5.0e-4) -- profiles/r04_plan_hybrid_ff.json, DESIGN.md section 2")
        extra["parity"] = parity
        cpu_sd = sd_cpu if (not args.no_cpu_baseline and not args.conditioned) else None   # timed after the side workloads (below)
        del sd_cpu
    else:
        cpu_sd = None
    del model
    torch.cuda.empty_cache()

    # ================================================================== side workloads (rank 0, single GPU)
    side = None
    if rank == 0 and world == 1 and not args.no_side and not args.conditioned and not args.graph:
        from oracle import ns2_oracle as O
        from oracle import rvq_oracle as R
        side = {}
        # --- BASELINE config 4: EnCodec RVQ encode of 32 x 1024 frames x 8 codebooks x 1024 codes
        g = torch.Generator().manual_seed(11)
        cb = torch.randn(8, 1024, 128, generator=g)
        lat = torch.randn(32 * 1024, 128, generator=g)
        cbd, latd = cb.to(dev), lat.to(dev)
        norm = ops.rvq_prepare(cbd)
        for _ in range(3):
            codes, emb = ops.rvq_encode(latd, cbd, norm)
        torch.cuda.synchronize()
        def timed(fn, n=20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                r = fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n, r
        ms, (codes, emb) = timed(lambda: ops.rvq_encode(latd, cbd, norm))                       # codes + summed embeddings
        ms_k, _ = timed(lambda: ops.rvq_encode(latd, cbd, norm, want_emb=False))              # the encode kernel alone
        c_ref, _, _ = R.rvq_encode(lat[:8192], cb)
        nbad = int((codes[:8192].cpu() != c_ref).any(dim=-1).sum())
        gflop = 2.0 * 32768 * 128 * 1024 * 8 / 1e9
        side["rvq_config4"] = dict(metric="RVQ encode, 32 x 1024 frames x 8 codebooks x 1024 codes", ms=round(ms, 4),
                                   ms_encode_kernel=round(ms_k, 4), frames_per_s=round(32768 / (ms * 1e-3)),
                                   parity=dict(rows_checked=8192, rows_differing_from_fp32_oracle=nbad,
                                   note="every differing row is an fp32 near-tie decided by exact arithmetic (tests/test_parity_r2_gpu.py)"),
                                   roofline=dict(bound="mfma (fp32 v_mfma_f32_16x16x4_f32)", kernel="ns2::rvq_encode_kernel (ms_encode_kernel; "
                                                 "`ms` adds the gather-sum of the 8 selected code vectors per frame)",
                                                 achieved=round(gflop / ms_k, 2), peak=PEAK_F32_MFMA_TFLOPS,
                                                 unit="TFLOP/s", frac=round(gflop / ms_k / PEAK_F32_MFMA_TFLOPS, 4)))
        del cbd, latd, codes, emb
        # --- BASELINE config 4 end to end: raw 24 kHz audio randn(32, 327680) -> SEANet encoder (HIP, seanet.py) -> RVQ codes / latents,
        #     and latents -> SEANet decoder -> audio; checked against HF's own EncodecModel on the same weights
        try:
            import transformers as tf
            from naturalspeech2_pytorch_amd import EncodecWrapperHIP
            torch.manual_seed(0)
            hf = tf.EncodecModel(tf.EncodecConfig()).eval()
            gq = torch.Generator().manual_seed(1)
            with torch.no_grad():
                for layer in hf.quantizer.layers:                    # HF zero-initialises the codebooks
                    layer.codebook.embed.copy_(torch.randn(layer.codebook.embed.shape, generator=gq))
            hf = hf.to(dev)
            codec = EncodecWrapperHIP.from_hf(hf, num_quantizers=8).to(dev)
            nb, nf = 32, 1024                                      # BASELINE config 4: encode of randn(32, 327680)
            wav = torch.randn(nb, nf * 320, generator=g).to(dev)
            with torch.no_grad():
                emb_c, codes_c, _ = codec(wav)                       # warm-up: packs the weights
                rec = codec.decode(emb_c)
                torch.cuda.synchronize()
                t_encs, t_decs = [], []
                for _ in range(3):                                   # median of three: a single call is at the mercy of the allocator
                    t0 = time.perf_counter()
                    emb_c, codes_c, _ = codec(wav)
                    torch.cuda.synchronize()
                    t_encs.append(time.perf_counter() - t0)
                    t0 = time.perf_counter()
                    rec = codec.decode(emb_c)
                    torch.cuda.synchronize()
                    t_decs.append(time.perf_counter() - t0)
                t_enc, t_dec = sorted(t_encs)[1], sorted(t_decs)[1]
                lat_hf = hf.encoder(wav[:2, None])
                codes_hf = hf.quantizer.encode(lat_hf, bandwidth=6.0).permute(1, 2, 0)            # [b, n, Q]
                rec_hf = hf.decoder(emb_c[:2].transpose(1, 2))
            ndiff = int((codes_c[:2] != codes_hf).any(dim=-1).sum())
            dec_err = float(((rec[:2] - rec_hf).double().norm() / rec_hf.double().norm()).item())
            secs = nb * nf * 320 / 24000.0
            # algorithmic work of the encode, counted from the layer shapes (seanet.py): 2 x MACs of every conv / LSTM Linear + the
            # 8-stage RVQ; compulsory bytes = audio in + latents / codes out + fp32 weights.  The arithmetic is bf16 x3 (3 MFMA units
            # per FLOP), so the MFMA floor is 3 x flops / 2.5 PF; the measured time is far above both floors: the path is bound by
            # its 2 x 1024 dependent LSTM steps and by the HBM passes over the early, wide activations (prep -> GEMM -> unpad).
            fl_enc, by_enc = codec.encoder.algorithmic_work(nb, nf * 320, 1)
            fl_enc += 2.0 * nb * nf * 128 * 1024 * 8
            by_enc += nb * nf * 8 * 8
            side["codec_seanet_rvq"] = dict(
                metric=f"EnCodec 24 kHz front end to end on the HIP path: {nb} x {nf * 320} samples -> SEANet encoder -> 8-stage RVQ "
                       f"(codes + latents); latents -> SEANet decoder -> audio; random-init HF EncodecModel weights",
                encode_ms=round(1e3 * t_enc, 2), decode_ms=round(1e3 * t_dec, 2),
                encode_x_realtime=round(secs / t_enc, 1), decode_x_realtime=round(secs / t_dec, 1),
                parity=dict(frames_checked=2 * nf, frames_with_codes_differing_from_hf=ndiff, decode_rel_err_vs_hf=dec_err),
                roofline=dict(bound="latency (1024 dependent frame pairs of the 2-layer LSTM) + hbm (activation passes of the wide early layers)",
                              algorithmic_gflop_per_encode=round(fl_enc / 1e9, 1), algorithmic_mb_per_encode=round(by_enc / 1e6, 1),
                              achieved=round(fl_enc / t_enc / 1e12, 2), unit="TFLOP/s", peak=PEAK_16BIT_TFLOPS,
                              frac=round(fl_enc / t_enc / 1e12 / PEAK_16BIT_TFLOPS, 5),
                              mfma_floor_ms=round(3.0 * fl_enc / (PEAK_16BIT_TFLOPS * 1e12) * 1e3, 3),
                              hbm_floor_ms=round(by_enc / 8e12 * 1e3, 4)),
                note="median of 3 calls; both LSTM layers in ONE launch (layer 2 a frame behind layer 1, frames synchronised through "
                     "tagged values, the batch in independent 8-row groups side by side); residual blocks as two GEMMs (conv2 + shortcut "
                     "over the concatenated K); kernel breakdown: profiles/r03_codec_kernel_stats.csv, profiles/r03_codec_timeline.txt")
            del hf, codec, wav, emb_c, codes_c, rec
        except Exception as e:                                        # transformers missing / API drift: report, do not fail the line
            side["codec_seanet_rvq"] = dict(skipped=f"{type(e).__name__}: {e}")
        torch.cuda.empty_cache()
        # --- BASELINE config 3 shape: conditioned d512/L12, prompt 103 frames, frame-aligned cond
        m3 = make_model(512, 12, True)
        e3, km, kn = measure(m3, args.precision, 10, 2, conditioned=True)
        sd3 = {k: v.detach().cpu() for k, v in m3.state_dict().items()}
        r3 = roofline_obj(args.precision, 512, 12, km, kn, conditioned=True)
        side["conditioned_config3"] = dict(metric="denoise steps/sec, Model(dim=512, depth=12, dim_prompt=512, condition_on_prompt), 32 x 1024 frames, prompt 103 frames, cond_scale 1",
                                           value=round(10 / e3, 3), ms_per_step=round(1e3 * e3 / 10, 3), precision=args.precision,
                                           parity=dict(live_rel_err_vs_fp32_oracle=live_parity(m3, sd3, args.precision, True)),
                                           whole_step_algorithmic_tflops=round(UTT_GFLOP[(512, 12, True)] * B * 1e9 / (e3 / 10) / 1e12, 2),
                                           roofline=dict(frac=r3["frac"], achieved=r3["achieved"], unit="TFLOP/s", kernel=r3["kernel"]) if r3 else None)
        del m3, sd3
        torch.cuda.empty_cache()
        # --- BASELINE config 2: d128/L6, 32 x 1024 frames
        m2 = make_model(128, 6, False)
        e2_, km, kn = measure(m2, args.precision, 20, 3)
        sd2 = {k: v.detach().cpu() for k, v in m2.state_dict().items()}
        r2 = roofline_obj(args.precision, 128, 6, km, kn)
        side["d128_config2"] = dict(metric="denoise steps/sec, Model(dim=128, depth=6), 32 x 1024 frames", value=round(20 / e2_, 3),
                                    ms_per_step=round(1e3 * e2_ / 20, 3), precision=args.precision,
                                    parity=dict(live_rel_err_vs_fp32_oracle=live_parity(m2, sd2, args.precision)),
                                    whole_step_algorithmic_tflops=round(UTT_GFLOP[(128, 6, False)] * B * 1e9 / (e2_ / 20) / 1e12, 2),
                                    roofline=dict(frac=r2["frac"], achieved=r2["achieved"], unit="TFLOP/s", kernel=r2["kernel"]) if r2 else None)
        del m2, sd2
        torch.cuda.empty_cache()
        # --- small batches (BASELINE config 3 is stated at b = 4, config 1 samples b = 1): a step of 1 / 4 utterances has 8 ... 100
        #     output tiles per GEMM on 256 CUs; the executor lends workspace scratch and the long-K products run split-K (gemm.hip
        #     launch_gemm_splitk).  The same steps with the split switched off (ns2_debug_force_gemm(3)) are timed beside it.
        def small_batch(model, b, conditioned):
            model.precision = args.precision
            g = torch.Generator().manual_seed(300)
            audio = torch.randn(b, N, 512, generator=g).to(dev)
            kw = dict(prompt=torch.randn(b, 103, 512, generator=g).to(dev), cond=torch.randn(b, 512, N, generator=g).to(dev)) if conditioned else {}
            ts = torch.linspace(1.0, 0.0, 31)
            out = {}
            with torch.no_grad():
                tab = model.time_table(ts[:30].to(dev), b)
                tcur = [ts[i].expand(b).contiguous().to(dev) for i in range(31)]
                for label, force in (("ms_per_step", 0), ("ms_per_step_without_split_k", 3)):
                    lib.ns2_debug_force_gemm(force)
                    try:
                        x = audio.clone()
                        for i in range(30):
                            if i == 10:
                                torch.cuda.synchronize()
                                t0 = time.perf_counter()
                            o = model.forward_with_cond_scale(x, None, cond_scale=1.0, cond_row=tab[i], **kw)
                            ops.ddim_step(x, o, tcur[i], tcur[i + 1], "v", "sigmoid", 1.0, out=x)
                        torch.cuda.synchronize()
                        out[label] = round(1e3 * (time.perf_counter() - t0) / 20, 3)
                    finally:
                        lib.ns2_debug_force_gemm(0)
                if conditioned:
                    # classifier-free guidance (NS2:914-927) at this batch: ONE batch of 2 b utterances (Model._forward_hip_cfg, round 6)
                    # against the two passes of b it replaces (CFG_ONE_BATCH_MAX = 0 switches it off)
                    keep = model.CFG_ONE_BATCH_MAX
                    for label, lim in (("cfg1.3_ms_per_step", keep), ("cfg1.3_ms_per_step_two_passes", 0)):
                        model.CFG_ONE_BATCH_MAX = lim
                        try:
                            x = audio.clone()
                            for i in range(30):
                                if i == 10:
                                    torch.cuda.synchronize()
                                    t0 = time.perf_counter()
                                o = model.forward_with_cond_scale(x, None, cond_scale=1.3, cond_row=tab[i], **kw)
                                ops.ddim_step(x, o, tcur[i], tcur[i + 1], "v", "sigmoid", 1.0, out=x)
                            torch.cuda.synchronize()
                            out[label] = round(1e3 * (time.perf_counter() - t0) / 20, 3)
                        finally:
                            model.CFG_ONE_BATCH_MAX = keep
                    out["cfg1.3_over_cond_scale_1"] = round(out["cfg1.3_ms_per_step"] / out["ms_per_step"], 3)
            out["utterance_steps_per_s"] = round(b * 1e3 / out["ms_per_step"], 1)
            return out

        try:
            sb = {}
            mu = make_model(512, 12, False)
            sdu = {k: v.detach().cpu() for k, v in mu.state_dict().items()}
            sb["d512_L12_b1x1024"] = small_batch(mu, 1, False)
            sb["d512_L12_b4x1024"] = small_batch(mu, 4, False)
            sb["live_rel_err_vs_fp32_oracle_b1x256"] = live_parity(mu, sdu, args.precision)      # a split-K forward
            del mu, sdu
            mc = make_model(512, 12, True)
            sb["config3_conditioned_b4x1024"] = small_batch(mc, 4, True)
            del mc
            sb["precision"] = args.precision
            side["small_batch"] = sb
        except Exception as e:                                        # report, do not fail the line
            side["small_batch"] = dict(skipped=f"{type(e).__name__}: {e}")
        torch.cuda.empty_cache()
        # --- one WARM training step (NaturalSpeech2.forward loss, NS2:1635-1666 + loss.backward(), NS2:1886 + Adam) on the
        #     HIP training path (training.py: forward and backward kernels of libns2hip, bf16 x3 arithmetic) at BASELINE config 1's
        #     training shape and at the headline shape; the PyTorch composite (fp32 torch ops on the same GPU) timed beside it
        try:
            side["train_step"] = train_step_side(dev)
        except Exception as e:                                        # report, do not fail the line
            side["train_step"] = dict(skipped=f"{type(e).__name__}: {e}")
        torch.cuda.empty_cache()

    if cpu_sd is not None:
        # last: importing the reference installs inert stand-ins for its optional imports (oracle/ref_stub.py), which nothing
        # else in this process should meet half-way
        extra["cpu_baseline"] = cpu_baseline(cpu_sd, dim, depth, B, N)
        if _REF_FULL and "parity" in extra:
            # parity AT THE BENCHED SHAPE, from the reference itself -- the unmodified upstream Model's output on the full
            # batch (the forward cpu_baseline just timed) against the HIP model on the same weights and inputs, in the benched arithmetic
            # and in `exact`; error over the whole batch and the worst single utterance
            from naturalspeech2_pytorch_amd import Model as _M
            hm = _M(dim=dim, depth=depth)
            hm.load_state_dict(cpu_sd)
            hm = hm.to(dev).eval()
            y = _REF_FULL["y"].double()
            res = {}
            for prec in dict.fromkeys((args.precision, "exact")):
                hm.precision = prec
                with torch.no_grad():
                    d = hm(_REF_FULL["x"].to(dev), _REF_FULL["t"].to(dev)).cpu().double() - y
                per = d.flatten(1).norm(dim=1) / y.flatten(1).norm(dim=1)
                res[prec] = dict(rel_err=float((d.norm() / y.norm()).item()), per_utterance_max=float(per.max().item()))
            res["what"] = (f"HIP Model vs the reference's own Model.forward (unmodified source, fp32, CPU) on the benched batch {B} x {N} x {dim}, "
                           f"same weights, x ~ N(0, 1), t ~ U(0, 1): ||y_hip - y_ref|| / ||y_ref|| over the batch and the worst utterance")
            res["tolerance"] = 1e-3
            extra["parity"][f"headline_b{B}_vs_reference"] = res
            del hm
            _REF_FULL.clear()
        del cpu_sd

    if rank == 0:
        whole = None
        key = (dim, depth, bool(args.conditioned))
        if key in UTT_GFLOP and N == 1024:
            whole = round(UTT_GFLOP[key] * B * 1e9 / (elapsed / args.steps) / 1e12, 2)
        headline = not args.conditioned and (dim, depth, B, N) == (512, 12, 32, 1024)
        line = {
            "metric": "denoise steps/sec (dim=512 depth=12, B=32x1024 latents)" if headline
                      else f"denoise steps/sec (side measurement: dim={dim} depth={depth} B={B}x{N}{' conditioned' if args.conditioned else ''})",
            "value": round(steps_per_s, 3), "unit": "steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE[args.precision],
            "data": "synthetic (randn codec latents, random-init weights)",
            "config": {"workload": (f"Model(dim={dim}, depth={depth}) unconditional, batch {B} x {N} latent frames per GPU, "
                                    f"forward_with_cond_scale(cond_scale=1) + DDIM update") if not args.conditioned else
                                   (f"Model(dim={dim}, depth={depth}, dim_prompt=512, condition_on_prompt) batch {B} x {N} frames, "
                                    f"prompt 103 frames, cond_scale={args.cond_scale} + DDIM update (BASELINE config 3 shape)"),
                       "precision": args.precision,
                       "global_batch": B * world, "parallelism": f"dp{world}", "graph_replay": bool(args.graph)},
            "whole_step_algorithmic_tflops_per_gpu": whole,
            "roofline": roofline_obj(args.precision, dim, depth, kern_ms_v, kern_n_v, conditioned=args.conditioned),
            "cpu_baseline": extra.pop("cpu_baseline", None),
            "parity": extra.pop("parity", None),
        }
        line.update(extra)
        if COLL:
            line["collective"] = COLL
        if side is not None:
            line["side"] = side
        if line["cpu_baseline"]:
            line["gpu_over_cpu"] = round(steps_per_s / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
