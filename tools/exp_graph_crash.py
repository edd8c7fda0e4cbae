"""bisect: which state before training.GraphedTrainStep makes hipStreamEndCapture crash (round 6)"""
import os, subprocess, sys
VARIANTS = ["none", "keep_fwd_only", "keep_backward", "keep_backward_deleted", "keep_detached", "grads_set_only", "keep_backward_sync_gc"]
if len(sys.argv) > 1:
    v = sys.argv[1]
    import gc
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from naturalspeech2_pytorch_amd import Model, NaturalSpeech2, training
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = Model(dim=128, depth=2).to(dev).train()
    m.train_backend, m.train_precision = "hip", "exact"
    d = NaturalSpeech2(m, codec=None, target_sample_hz=24000).to(dev)
    a, t, z = torch.randn(2, 256, 128, device=dev), torch.rand(2, device=dev), torch.randn(2, 256, 128, device=dev)
    fn = lambda a_, t_, z_: d(a_, times=t_, noise=z_)
    keep = None
    if v == "keep_fwd_only":
        keep = fn(a, t, z)
    elif v in ("keep_backward", "keep_backward_deleted", "keep_backward_sync_gc"):
        keep = fn(a, t, z); keep.backward(retain_graph=True)
        if v == "keep_backward_deleted":
            del keep; keep = None
        if v == "keep_backward_sync_gc":
            torch.cuda.synchronize(); gc.collect()
    elif v == "keep_detached":
        keep = fn(a, t, z); keep.backward(); keep = keep.detach()
    elif v == "grads_set_only":
        fn(a, t, z).backward()
    step = training.GraphedTrainStep(fn, (a, t, z), m)
    step(a, t, z); torch.cuda.synchronize()
    print("OK", v, float(step.loss))
else:
    for v in VARIANTS:
        r = subprocess.run([sys.executable, __file__, v], capture_output=True, text=True, timeout=300)
        tail = [l for l in (r.stdout + r.stderr).splitlines() if l.startswith("OK") or "Error" in l or "fault" in l.lower()]
        print(f"{v:28s} rc={r.returncode} {tail[-2:]}", flush=True)
