"""GPU-box probe: environment facts + MFMA lane-layout verification (run via gpurun)."""
import ctypes, os, subprocess, sys, json, time
import torch

here = os.path.dirname(os.path.abspath(__file__))
out = {}
out["torch"] = torch.__version__
out["cuda_available"] = torch.cuda.is_available()
out["nproc"] = os.cpu_count()
try:
    out["cpu_model"] = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
except Exception as e:
    out["cpu_model"] = repr(e)
out["mem_gb"] = int(open("/proc/meminfo").readline().split()[1]) // (1 << 20)
if not torch.cuda.is_available():
    print(json.dumps(out)); sys.exit(1)
p = torch.cuda.get_device_properties(0)
out["gpu"] = dict(name=p.name, cus=p.multi_processor_count, mem=p.total_memory, gcn=getattr(p, "gcnArchName", "?"))

lib = ctypes.CDLL(os.path.join(here, "libprobe.so"))
lib.probe_run.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 5
lib.probe_run.restype = ctypes.c_int
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
stream = torch.cuda.current_stream().cuda_stream

def run(which, A, B, dshape, rawshape):
    D = torch.full(dshape, -7.0, device=dev); raw = torch.zeros(rawshape, device=dev)
    rc = lib.probe_run(which, A.data_ptr(), B.data_ptr(), D.data_ptr(), raw.data_ptr(), stream)
    torch.cuda.synchronize()
    return rc, D.cpu(), raw.cpu()

res = {}
for which, (m, n, k, bf) in enumerate([(32, 32, 16, True), (16, 16, 32, True), (32, 32, 2, False), (16, 16, 4, False)]):
    A = torch.randn(m, k, generator=g); B = torch.randn(k, n, generator=g)
    if bf:
        A = A.bfloat16(); B = B.bfloat16()
    ref = A.float() @ B.float()
    Ad = A.to(dev).contiguous(); Bd = B.to(dev).contiguous()
    rc, D, raw = run(which, Ad, Bd, (m, n), (64, m * n // 64))
    err = (D - ref).abs().max().item()
    res[f"mfma_{m}x{n}x{k}"] = dict(rc=rc, max_err=err, ok=bool(err < 1e-4))
    if err >= 1e-4:
        torch.save(dict(A=A, B=B, D=D, raw=raw, ref=ref), os.path.join(here, "..", "..", "gpurun_out", f"probe_fail_{which}.pt"))
out["mfma"] = res
os.makedirs(os.path.join(here, "..", "..", "gpurun_out"), exist_ok=True)
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(here, "..", "..", "gpurun_out", "probe.json"), "w"), indent=1)
