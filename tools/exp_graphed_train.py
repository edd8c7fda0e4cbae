"""Round 6: the training pass as one HIP graph (training.GraphedTrainStep) vs the eager pass: gradients bit for bit, step time.
    python tools/exp_graphed_train.py [--headline]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from naturalspeech2_pytorch_amd import Model, NaturalSpeech2, training  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--headline", action="store_true")
ap.add_argument("--lean", action="store_true", help="mixed linear packs WITH the lean kernel's tile images (A/B; off by default)")
args = ap.parse_args()
dev = torch.device("cuda:0")
if args.lean:
    training._PackedCache.LEAN = True
cases = [("d128_L6_b4", dict(dim=128, depth=6), 4, 1024), ("cond_d128_L2_b3", dict(dim=128, depth=2, dim_prompt=128, condition_on_prompt=True), 3, 512)]
if args.headline:
    cases.append(("d512_L12_b32", dict(dim=512, depth=12), 32, 1024))
for tag, kw, b, n in cases:
    for tprec in ("exact", "mixed"):
        torch.manual_seed(0)
        m = Model(**kw).to(dev).train()
        m.train_backend, m.train_precision = "hip", tprec
        d = NaturalSpeech2(m, codec=None, target_sample_hz=24000).to(dev)
        g = torch.Generator().manual_seed(1)
        mk = lambda *s, u=False: (torch.rand(*s, generator=g) if u else torch.randn(*s, generator=g)).to(dev)
        audio, times, noise = mk(b, n, kw["dim"]), mk(b, u=True), mk(b, n, kw["dim"])
        extra = {}
        if kw.get("condition_on_prompt"):
            extra = dict(prompt=mk(b, 40, kw["dim_prompt"]), cond=mk(b, kw["dim_prompt"], n))
        names = list(extra)
        def loss_fn(a, t, z, *ex):
            if ex:                                # conditioned: the denoiser itself (conditioning dropout drawn inside the pass)
                return (m(a, t, **dict(zip(names, ex))) - z).square().mean()
            return d(a, times=t, noise=z)
        ins = (audio, times, noise) + tuple(extra.values())
        def eager():
            for p in m.parameters():
                p.grad = None
            loss = loss_fn(*ins)
            loss.backward()
            return loss
        for _ in range(2):
            eager()
        l0 = eager().detach().clone()
        g0 = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
        step = training.GraphedTrainStep(loss_fn, ins, m)
        l1 = step(*ins).detach().clone()
        bad = [k for k, p in m.named_parameters() if p.grad is not None and not torch.equal(p.grad, g0[k])]
        miss = [k for k in g0 if dict(m.named_parameters())[k].grad is None]
        # a second batch through the same graph against a fresh eager pass on it
        audio2 = mk(b, n, kw["dim"])
        ins2 = (audio2,) + ins[1:]
        step(*ins2)
        g2 = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
        ins_save = ins; ins = ins2; eager(); ins = ins_save
        bad2 = [k for k, p in m.named_parameters() if p.grad is not None and not torch.equal(p.grad, g2[k])]
        def timeit(fn, k):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(k):
                fn()
            torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
        k = 3 if "d512" in tag else 10
        te = [timeit(eager, k) for _ in range(2)]
        tg = [timeit(lambda: step(*ins), k) for _ in range(2)]
        print(f"{tag:18s} {tprec:6s} loss eager {l0.item():.6f} graph {l1.item():.6f}  grads differing {len(bad)} missing {len(miss)}  second batch differing {len(bad2)}"
              f"  eager {te[0]:.2f} / {te[1]:.2f} ms   graphed {tg[0]:.2f} / {tg[1]:.2f} ms  overflow {step.overflowed()}", flush=True)
        del step, m, d
        torch.cuda.empty_cache()
