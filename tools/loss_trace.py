import sys, torch
sys.path.insert(0, '.')
from naturalspeech2_pytorch_amd import Model, NaturalSpeech2
dev = torch.device("cuda:0")
kw = dict(dim=128, depth=6); b, n = 4, 1024
for backend, tprec in (("hip", "exact"), ("composite", "exact"), ("hip", "mixed")):
    torch.manual_seed(0)
    m = Model(**kw).to(dev).train()
    m.train_backend, m.train_precision = backend, tprec
    d = NaturalSpeech2(m, codec=None, target_sample_hz=24000).to(dev)
    opt = torch.optim.Adam(m.parameters(), lr=1e-4, fused=True)
    g = torch.Generator().manual_seed(1)
    audio, times, noise = torch.randn(b, n, kw["dim"], generator=g).to(dev), torch.rand(b, generator=g).to(dev), torch.randn(b, n, kw["dim"], generator=g).to(dev)
    ls = []
    for i in range(7):
        opt.zero_grad(set_to_none=True)
        loss = d(audio, times=times, noise=noise)
        loss.backward()
        gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters() if p.grad is not None)).item()
        opt.step()
        ls.append((round(float(loss), 6), round(gn, 5)))
    print(backend, tprec, ls)
