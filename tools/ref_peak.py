"""Reference point for the roofline discussion (measurement tool only, not part of the product path): what rate does the
vendor bf16 GEMM (torch.matmul -> hipBLASLt) sustain on this box for random vs all-zero operands?"""
import torch
def bench(m, n, k, zero):
    a = (torch.zeros if zero else torch.randn)(m, k, device="cuda", dtype=torch.bfloat16)
    b = (torch.zeros if zero else torch.randn)(n, k, device="cuda", dtype=torch.bfloat16)
    for _ in range(5): torch.matmul(a, b.t())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): torch.matmul(a, b.t())
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 30
    print(f"bf16 matmul {m}x{n}x{k} {'zeros ' if zero else 'random'}: {us:9.1f} us  {2.0*m*n*k/us/1e6:8.1f} TFLOP/s")
for shape in [(32768, 1536, 4128), (8192, 8192, 8192), (32768, 1536, 12384)]:
    for z in (False, True):
        bench(*shape, z)
