"""CPU-side tests: C-ABI completeness, state_dict contract, host logic (schedules, sharding), the autograd
composite against the reference goldens, and the world_size-2 data-parallel path over gloo."""
import glob
import os
import subprocess
import sys

import pytest
import torch

from naturalspeech2_pytorch_amd import _lib, Model, NaturalSpeech2
from naturalspeech2_pytorch_amd import distributed as D
from naturalspeech2_pytorch_amd.autograd_path import model_forward_autograd
from oracle import ns2_oracle as O
from tests.golden.gen import make_weights, make_input

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_library_exports_whole_abi():
    lib = _lib.load()                                   # raises if the .so or any symbol is missing
    assert set(_lib.header_symbols()) == set(_lib.SIGNATURES)
    assert lib.ns2_version() >= 100
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T ns2_" in l}
    assert set(_lib.header_symbols()) <= exported


def test_split_k_plan_host_arithmetic():
    """gemm2.hip splitk_plan through ns2_debug_splitk_plan (host arithmetic, no device): which products of a small batch are cut
    into K slices, and the invariants the kernels rely on -- no empty slice, every K tile covered once, at least 4 K tiles of a tap
    per slice, slices x output tiles <= 512 (so every slot fits the 32 MiB the forward lends), nothing split once the 128 x 128
    output tiles fill the chip."""
    import ctypes
    lib = _lib.load()

    def plan(M, N, K, taps=1, f32=False):
        kpt = (K + 31) // 32
        S, c = ctypes.c_int(-1), ctypes.c_int(-1)
        _lib.check(lib.ns2_debug_splitk_plan(M, N, kpt * taps, kpt, int(f32), ctypes.byref(S), ctypes.byref(c)), "plan")
        return S.value, c.value, kpt

    scratch = int(lib.ns2_splitk_scratch_bytes()) // 4
    # the d512 / L12 model at 1 x 1024 frames (small batches)
    assert plan(1024, 1365, 1376, taps=3)[:2] == (5, 9)              # FF causal conv: 43 K tiles per tap as 9 + 9 + 9 + 9 + 7
    assert plan(1024, 512, 1376, f32=True)[:2] == (9, 5)             # FF-out
    assert plan(1024, 512, 512, f32=True)[:2] == (4, 4)              # out-proj
    assert plan(1024, 1536, 512)[:2] == (4, 4)                       # QKV
    assert plan(1024, 128, 352, f32=True)[0] == 2                    # dim = 128 FF-out: 11 K tiles as 6 + 5
    assert plan(1024, 128, 128)[0] == 1                              # K = 128: nothing to split
    assert plan(32768, 512, 1376, f32=True)[0] == 1                  # the headline batch fills the chip
    assert plan(32768, 128, 512, f32=True)[0] == 1                   # ... also with one column tile (256 row tiles)
    for M in (1, 77, 256, 1000, 1024, 2048, 4096, 8192, 16384):
        for N in (1, 64, 128, 341, 512, 1365, 1536, 2752):
            for K in (32, 224, 256, 352, 512, 1024, 1376, 4096):
                for taps in (1, 3):
                    for f32 in (False, True):
                        S, c, kpt = plan(M, N, K, taps, f32)
                        if S == 1:
                            continue
                        tiles = -(-M // 128) * -(-N // 128)
                        assert tiles < 256 and kpt * taps >= (11 if f32 else 16) and kpt >= 8
                        assert S >= 2 and c >= 4 and (S - 1) * c < kpt <= S * c          # no empty slice, all tiles covered
                        assert S * tiles <= 512 and S * M * (-(-N // 64) * 64) <= scratch


def test_no_cpu_fallback():
    m = Model(dim=64, depth=1).eval()
    with torch.no_grad(), pytest.raises(_lib.Ns2Error):
        m(torch.zeros(1, 8, 64), torch.zeros(1))        # CPU tensors: the product path refuses, it does not fall back


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "model_*.pt"))), ids=os.path.basename)
def test_state_dict_contract_matches_reference(path):
    fix = torch.load(path, weights_only=False)
    m = Model(**fix["kwargs"])
    own = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    ref = [(k, tuple(v)) for k, v in fix["shapes"].items()]
    assert own == ref                                    # same keys, same order, same shapes as the reference Model


def test_header_is_plain_c_and_a_c_caller_links_against_the_library(tmp_path):
    """the drop-in boundary is a C ABI (extern "C", plain pointers and sizes, no torch types): include/ns2hip.h must compile as C99 with gcc, and
    a C program that takes the address of entry points of every section (and calls the ones that need no GPU) must link against libns2hip.so
    and run -- the binding a maintainer of another host language would write."""
    import shutil
    import subprocess
    from naturalspeech2_pytorch_amd import _lib
    if shutil.which("gcc") is None or not os.path.exists(_lib.LIB_PATH):
        pytest.skip("needs gcc and the built library")
    src = tmp_path / "caller.c"
    src.write_text(r'''
#include <stdio.h>
#include <stdint.h>
#include "ns2hip.h"
int main(void) {
  /* addresses of entry points from every section of the header: unresolved symbols fail the link */
  typedef void (*fn_t)(void);
  fn_t fns[] = {(fn_t)ns2_weight_pack, (fn_t)ns2_linear_f32, (fn_t)ns2_linear_qkv, (fn_t)ns2_attention,
                       (fn_t)ns2_attention_hd, (fn_t)ns2_rmsnorm, (fn_t)ns2_rvq_encode, (fn_t)ns2_model_create,
                       (fn_t)ns2_model_forward, (fn_t)ns2_ddim_step, (fn_t)ns2_weights_repack, (fn_t)ns2_weights_retile,
                       (fn_t)ns2_attention_bwd, (fn_t)ns2_weight_tile_linear, (fn_t)ns2_model_cond_stack};
  int n = 0;
  for (unsigned i = 0; i < sizeof fns / sizeof fns[0]; ++i) n += fns[i] != 0;
  ns2_model_config cfg;                       /* the config struct is plain ints */
  cfg.dim = 64; cfg.depth = 1; cfg.dim_head = 48; cfg.heads = 2;
  printf("%d %d %d %d\n", n, ns2_version(), ns2_conv3_input_ld(1365), (int)sizeof(cfg) % (int)sizeof(int));
  return ns2_debug_force_gemm(99) == 0;       /* argument errors come back as codes, nothing throws across the ABI */
}
''')
    exe = tmp_path / "caller"
    libdir = os.path.dirname(_lib.LIB_PATH)
    inc = os.path.join(ROOT, "include")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", inc, "-fsyntax-only", str(src)], check=True)
    subprocess.run(["gcc", "-std=c99", "-I", inc, str(src), "-o", str(exe), "-L", libdir, "-lns2hip", "-Wl,-rpath," + libdir], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True, timeout=120).stdout.split()
    assert out[0] == "15" and int(out[1]) >= 115 and int(out[2]) == 1408 and out[3] == "0", out


def test_transformer_state_dict_contract():
    from naturalspeech2_pytorch_amd import Transformer
    fix = torch.load(os.path.join(GOLD, "transformer_d64.pt"), weights_only=False)
    m = Transformer(**fix["kwargs"])
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == [(k, tuple(v)) for k, v in fix["shapes"].items()]


def test_encoder_state_dict_contracts():
    from naturalspeech2_pytorch_amd import PhonemeEncoder, SpeechPromptEncoder
    for name, cls in (("speech_prompt_encoder", SpeechPromptEncoder), ("phoneme_encoder", PhonemeEncoder)):
        fix = torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)
        m = cls(**fix["kwargs"])
        assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == [(k, tuple(v)) for k, v in fix["shapes"].items()]


def test_default_init_matches_reference_distributions():
    m = Model(dim=64, depth=1, dim_prompt=64, condition_on_prompt=True)
    assert m.null_cond.abs().sum() == 0                                  # NS2:881
    assert getattr(m.transformer.to_pred, "0").gamma.eq(1).all()         # NS2:734
    assert abs(m.perceiver_resampler.latents.std().item() - 0.02) < 0.01  # NS2:551-552
    w = getattr(m.transformer.layers[0], "1").to_q.weight
    assert w.abs().max().item() <= 1 / 8 + 1e-6                          # nn.Linear default U(-1/sqrt(64), 1/sqrt(64))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "model_*d64*.pt"))), ids=os.path.basename)
def test_autograd_composite_matches_reference_golden(path):
    fix = torch.load(path, weights_only=False)
    kw = fix["kwargs"]
    m = Model(**kw).eval()
    m.load_state_dict(make_weights(fix["shapes"], seed=fix["weight_seed"]))
    b, n = fix["batch"], fix["n"]
    x = make_input("x", (b, n, kw["dim"]), seed=fix["input_seed"])
    t = make_input("times", (b,), seed=fix["input_seed"], uniform=True)
    prompt = cond = None
    if kw.get("condition_on_prompt"):
        prompt = make_input("prompt", (b, fix["n_prompt"], kw["dim_prompt"]), seed=fix["input_seed"])
        cond = make_input("cond", (b, kw["dim_prompt"], fix["n_cond"]), seed=fix["input_seed"])
    y = model_forward_autograd(m, x, t, prompt=prompt, cond=cond, cond_drop_prob=0.)
    ref = fix["outputs"]["cond_scale_1.0"]
    assert ((y - ref).norm() / ref.norm()).item() < 2e-5
    y.sum().backward()
    assert m.wavenet.init_conv.weight.grad is not None


def test_training_loss_cpu():
    m = Model(dim=64, depth=1)
    d = NaturalSpeech2(m, codec=None, target_sample_hz=24000, timesteps=10)
    audio = make_input("audio", (2, 24, 64), seed=3)
    loss = d(audio, times=torch.tensor([0.2, 0.7]), noise=make_input("noise", (2, 24, 64), seed=4))
    loss.backward()
    assert torch.isfinite(loss)
    # v-objective loss restated with the oracle's schedule (NS2:1627-1668)
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    times = torch.tensor([0.2, 0.7])
    g = O.sigmoid_schedule(times)[:, None, None]
    a, s = O.gamma_to_alpha_sigma(g)
    noise = make_input("noise", (2, 24, 64), seed=4)
    with torch.no_grad():
        pred = O.model_forward(sd, a * audio + s * noise, times)
    tgt = a * noise - s * audio
    snr = a * a / (s * s)                                                   # [b, 1, 1]
    # NS2:1668 as written upstream: [b] * [b, 1, 1] broadcasts to [b, 1, b]; the mean is mean(loss) * mean(weight)
    ref = (((pred - tgt) ** 2).flatten(1).mean(1) * (snr.clamp(max=5) / (snr + 1))).mean()
    assert abs(loss.item() - ref.item()) < 1e-5 * max(1.0, abs(ref.item()))


def test_reference_signatures_are_kept():
    """NaturalSpeech2.sample / .forward keep the reference's parameter names and order (NS2:1457-1466, 1503-1515)."""
    import inspect
    fwd = list(inspect.signature(NaturalSpeech2.forward).parameters)
    assert fwd[:9] == ["self", "audio", "text", "text_lens", "mel", "mel_lens", "codes", "prompt", "pitch"]
    smp = inspect.signature(NaturalSpeech2.sample).parameters
    assert list(smp)[:7] == ["self", "length", "prompt", "batch_size", "cond_scale", "text", "text_lens"]
    assert all(smp[k].kind is inspect.Parameter.KEYWORD_ONLY for k in list(smp)[1:])
    dd = list(inspect.signature(NaturalSpeech2.ddim_sample).parameters)
    assert dd[:6] == ["self", "shape", "prompt", "time_difference", "cond_scale", "cond"]
    mfp = inspect.signature(Model.forward).parameters
    mf = list(mfp)
    assert mf[:7] == ["self", "x", "times", "prompt", "prompt_mask", "cond", "cond_drop_prob"]          # NS2:929-937
    assert all(mfp[k].kind is inspect.Parameter.KEYWORD_ONLY and mfp[k].default is None for k in mf[7:])   # extras: keyword-only, optional


def test_conditional_wrapper_accepts_reference_kwargs_and_names_the_missing_module():
    m = Model(dim=64, depth=1, dim_prompt=64, condition_on_prompt=True)
    d = NaturalSpeech2(m, codec=None, target_sample_hz=24000, timesteps=2, dim_codebook=64)
    keys = set(d.state_dict())
    assert any(k.startswith("prompt_enc.transformer.") for k in keys) and any(k.startswith("phoneme_enc.token_emb") for k in keys)
    assert "pitch_emb.weight" in keys
    audio = make_input("audio", (2, 16, 64), seed=1)
    text = torch.randint(0, 100, (2, 10))
    with pytest.raises(NotImplementedError, match="Aligner"):         # out-of-scope branch: needs aligner / duration predictor
        d(audio, text=text, text_lens=torch.tensor([10, 7]), mel=torch.randn(2, 80, 16), pitch=torch.randn(2, 1, 16),
          prompt_enc=torch.randn(2, 5, 64))
    # pre-computed conditioning through the extra keywords: the reference kwargs are accepted alongside
    loss = d(audio, text=text, text_lens=torch.tensor([10, 7]), prompt_enc=torch.randn(2, 5, 64), cond=torch.randn(2, 64, 16))
    loss.backward()
    assert torch.isfinite(loss)
    with pytest.raises(NotImplementedError, match="DurationPitchPredictor"):
        d.sample(length=16, prompt_enc=torch.randn(2, 5, 64), text=text)


def test_rvq_cross_entropy_term():
    """codec.rq(x_start, codes) (NS2:1670-1684): loss = diffusion loss + weight * sum_q CE(-euclid(residual_q, E_q), codes_q)."""
    import torch.nn.functional as F
    from naturalspeech2_pytorch_amd import EncodecWrapperHIP
    cb = make_input("codebooks", (2, 64, 128), seed=5)
    codec = EncodecWrapperHIP(cb)
    x = make_input("x", (2, 6, 128), seed=6).requires_grad_(True)
    # indices = the nearest codes of the running residual (what the codec's own encode yields)
    res, codes = x.detach().clone(), []
    for q in range(2):
        idx = torch.cdist(res, cb[q][None].expand(2, -1, -1)).argmin(-1)
        codes.append(idx)
        res = res - cb[q][idx]
    codes = torch.stack(codes, dim=-1)
    out, ce = codec.rq(x, codes)
    res, ref = x.detach(), 0.
    for q in range(2):
        dist = torch.cdist(res, cb[q][None].expand(2, -1, -1))
        ref = ref + F.cross_entropy((-dist).transpose(1, 2), codes[..., q])
        res = res - cb[q][dist.argmin(-1)]
    assert abs(ce.item() - ref.item()) < 1e-4 * max(1., abs(ref.item()))
    assert torch.allclose(out, x.detach() - res, atol=1e-5)
    ce.backward()
    assert x.grad is not None and torch.isfinite(x.grad).all()


def test_sampling_timesteps_and_schedules():
    m = Model(dim=64, depth=1)
    d = NaturalSpeech2(m, codec=None, target_sample_hz=24000, timesteps=5)
    pairs = d.get_sampling_timesteps(3, device="cpu")
    ref = O.sampling_timesteps(3, 5)
    assert len(pairs) == 5
    for (a, b), (c, e) in zip(pairs, ref):
        assert torch.equal(a, c) and torch.equal(b, e)
    from naturalspeech2_pytorch_amd.diffusion import sigmoid_schedule
    t = torch.linspace(0, 1, 11)
    assert torch.equal(sigmoid_schedule(t), O.sigmoid_schedule(t))
    with pytest.raises(AssertionError):
        NaturalSpeech2(m, codec=None, target_sample_hz=24000, use_ddim=False)   # ddpm_sample is broken upstream (NS2:1361)


def test_shard_ranges_cover_everything():
    for total in (1, 7, 32, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [D.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_utterance_noise_is_shard_invariant():
    full = D.utterance_noise(0, 6, 4, 8, seed=3)
    part = D.utterance_noise(2, 5, 4, 8, seed=3)
    assert torch.equal(full[2:5], part)


WORKER = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from naturalspeech2_pytorch_amd import distributed as D
rank, local, world = D.init_from_env("gloo")
fn = lambda noise: noise * 2.0 + noise.flip(-1).cumsum(dim=1)        # any per-utterance map
out = D.sharded_sample(fn, total=int(sys.argv[2]), length=6, dim=8, seed=5)
single = fn(D.utterance_noise(0, int(sys.argv[2]), 6, 8, seed=5))
assert out.shape == single.shape and torch.equal(out, single), f"rank {rank}: gathered result differs"
# the block bench.py --gpus N prints beside its line (VERDICT r5 item 9): one collective, timed apart from the loop
lo, hi = D.shard_range(int(sys.argv[2]), 0, world)
c = D.collective_report(torch.ones(hi - lo, 6, 8) * (rank + 1), local_ms_per_step=10.0 + rank)
assert c["backend"] == "gloo" and c["world"] == world and c["collectives_inside_the_loop"] == 0 and c["allgather_ms"] > 0
assert c["allgather_bytes_per_rank"] == (hi - lo) * 6 * 8 * 4 and c["per_rank_ms_per_step"] == dict(min=10.0, max=10.0 + world - 1, all=[10.0 + r for r in range(world)])
dist.barrier()
if rank == 0:
    print("OK", world, tuple(out.shape))
dist.destroy_process_group()
'''


@pytest.mark.parametrize("total", [5, 8])
def test_sharded_sample_world2_gloo(tmp_path, total):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = 29600 + os.getpid() % 300 + total
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script), ROOT, str(total)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "OK 2" in r.stdout


def test_planes_layout_helper_cpu():
    """ops.Planes mirrors the library's split-plane layout (csrc/ns2_common.h): [hi32|lo32] per 32 logical columns."""
    import torch
    from naturalspeech2_pytorch_amd import ops
    rows, ld = 5, 96
    hi = torch.arange(rows * ld, dtype=torch.float32).reshape(rows, ld).to(torch.bfloat16)
    lo = (-torch.arange(rows * ld, dtype=torch.float32).reshape(rows, ld)).to(torch.bfloat16)
    buf = torch.stack([hi.reshape(rows, ld // 32, 32), lo.reshape(rows, ld // 32, 32)], dim=2).reshape(rows, 2 * ld).contiguous()
    p = ops.Planes(buf, rows, ld, True)
    assert p.lo == p.hi + 64                                   # lo pointer = hi + 32 bf16 elements
    assert torch.equal(p.hi_plane(), hi)
    d = p.hi_only()
    assert not d.has_lo and d.lo is None and torch.equal(d.buf, hi)
    for c in (0, 31, 32, 63, 64, 95):                          # physical column of logical column c
        assert buf[2, ((c & ~31) << 1) | (c & 31)] == hi[2, c] and buf[2, (((c & ~31) << 1) | (c & 31)) + 32] == lo[2, c]


def test_parity_record_is_complete():
    """The tracked parity record must hold every key bench.py quotes (a partial GPU run once overwrote it with one key)."""
    import json
    import re
    from tests import parity_record as PR
    rec = json.load(open(PR.TRACKED))
    for k in PR.REQUIRED_KEYS:
        assert k in rec and {"max", "mean"} <= set(rec[k]), f"{PR.TRACKED} lacks {k}"
    bench_src = open(os.path.join(ROOT, "bench.py")).read()
    m = re.search(r'PARITY_RECORD = "([^"]+)"', bench_src)
    assert m and os.path.join(ROOT, "profiles", m.group(1)) == PR.TRACKED, "bench.py and tests/parity_record.py must name the same file"


def test_parity_record_merges_key_by_key(tmp_path, monkeypatch):
    """record() carries every other key of the tracked record along: a partial run can never shrink the evidence."""
    import json
    from tests import parity_record as PR
    tracked, scratch = tmp_path / "tracked.json", tmp_path / "out" / "scratch.json"
    tracked.write_text(json.dumps({"a": 1, "b": {"max": 2}}))
    monkeypatch.setattr(PR, "TRACKED", str(tracked))
    monkeypatch.setattr(PR, "SCRATCH", str(scratch))
    PR.record("c", 3)
    PR.record("a", 10)
    got = json.load(open(scratch))
    assert got["a"] == 10 and got["b"] == {"max": 2} and got["c"] == 3
    assert got["_meta"]["updated_keys_r06"] == ["a", "c"]


def test_conditional_training_reaches_the_encoders():
    """the conditional NaturalSpeech2.forward in its default train() mode (SpeechPromptEncoder dropout 0.2) must run
    and give prompt_enc gradients -- the reference trains it jointly (NS2:1542-1543).  Under autograd the encoders run the
    differentiable composite (autograd_path.py); the HIP forwards stay inference-only."""
    from naturalspeech2_pytorch_amd import Model, NaturalSpeech2
    torch.manual_seed(0)
    m = Model(dim=64, depth=1, dim_prompt=512, condition_on_prompt=True)
    d = NaturalSpeech2(m, codec=None, target_sample_hz=24000, dim_codebook=32)
    d.train()
    audio = torch.randn(2, 24, 64)
    prompt = torch.randn(2, 10, 32)                   # codec latents of the prompt [b, n_p, dim_codebook]
    cond = torch.randn(2, 512, 24)
    loss = d(audio, prompt=prompt, cond=cond)
    loss.backward()
    g = [p.grad for p in d.prompt_enc.parameters()]
    assert all(x is not None for x in g) and sum(float(x.abs().sum()) for x in g) > 0
    assert all(p.grad is not None for p in m.parameters() if p.requires_grad and p.numel() and p is not m.null_cond
               and p is not m.null_prompt_cond and p is not m.null_prompt_tokens)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="live reference only exists in the build container")
def test_encoder_composites_equal_the_reference():
    from oracle.ref_stub import load_reference
    from naturalspeech2_pytorch_amd.encoders import PhonemeEncoder, SpeechPromptEncoder
    ns2 = load_reference()
    torch.manual_seed(0)
    r = ns2.SpeechPromptEncoder(dim_codebook=32, dims=(48, 64), depth=2, dropout=0.)
    o = SpeechPromptEncoder(dim_codebook=32, dims=(48, 64), depth=2, dropout=0.)
    o.load_state_dict(r.state_dict())
    x = torch.randn(2, 20, 32, requires_grad=True)
    assert torch.allclose(r(x), o(x), atol=1e-6)
    r = ns2.PhonemeEncoder(num_tokens=50, dim=32, dim_hidden=64, depth=2, conv_dropout=0.).eval()
    o = PhonemeEncoder(num_tokens=50, dim=32, dim_hidden=64, depth=2, conv_dropout=0.)   # training mode (no dropout): the composite; eval() is the HIP path
    o.load_state_dict(r.state_dict())
    ids = torch.randint(0, 50, (2, 17)); ids[1, 12:] = -1
    assert torch.allclose(r(ids, mask=ids >= 0), o(ids, mask=ids >= 0), atol=1e-5)


DP_WORKER = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from naturalspeech2_pytorch_amd import Model, NaturalSpeech2
from naturalspeech2_pytorch_amd import distributed as D
rank, local, world = D.init_from_env("gloo")
torch.manual_seed(0)
m = Model(dim=64, depth=1)
d = NaturalSpeech2(m, codec=None, target_sample_hz=24000)
g = torch.Generator().manual_seed(1)
audio = torch.randn(4, 24, 64, generator=g); times = torch.rand(4, generator=g); noise = torch.randn(4, 24, 64, generator=g)
# single-process reference: mean over the two shards' losses == what DP averages
ref_grads = None
for r in range(world):
    lo, hi = D.shard_range(4, r, world)
    d.zero_grad()
    (d(audio[lo:hi], times=times[lo:hi], noise=noise[lo:hi]) / world).backward()
    gs = [p.grad.clone() if p.grad is not None else torch.zeros_like(p) for p in d.parameters()]
    ref_grads = gs if ref_grads is None else [a + b for a, b in zip(ref_grads, gs)]
# data parallel: this rank's shard, bucketed all-reduce overlapped with backward (tiny buckets: several collectives)
red = D.GradientAllReducer(d.parameters(), bucket_bytes=1 << 16)
assert len(red.buckets) > 3
assert all(p.grad.data_ptr() == red._view[id(p)].data_ptr() for p in red.params)      # gradients live in the flat buffer
red.zero_grad()
lo, hi = D.shard_range(4, rank, world)
d(audio[lo:hi], times=times[lo:hi], noise=noise[lo:hi]).backward()
red.finish()
err = max(float((p.grad - r).abs().max()) for p, r in zip(d.parameters(), ref_grads))
assert err < 1e-5, f"rank {rank}: averaged gradients differ from the single-process gradients: {err}"
# a second step after zero_grad(set_to_none=True): fresh gradient tensors are adopted back into their views
d.zero_grad(set_to_none=True); d(audio[lo:hi], times=times[lo:hi], noise=noise[lo:hi]).backward(); red.finish()
err2 = max(float((p.grad - r).abs().max()) for p, r in zip(d.parameters(), ref_grads))
assert err2 < 1e-5
assert all(p.grad.data_ptr() == red._view[id(p)].data_ptr() for p in red.params)
# gradient accumulation (NS2:1877-1885): two micro-batches per rank, only the last backward reduces
red.zero_grad()
mid = (lo + hi) // 2
with red.accumulate():
    (d(audio[lo:mid], times=times[lo:mid], noise=noise[lo:mid]) / 2).backward()
(d(audio[mid:hi], times=times[mid:hi], noise=noise[mid:hi]) / 2).backward()
red.finish()
ref_acc = None
for r in range(world):
    rlo, rhi = D.shard_range(4, r, world); rm = (rlo + rhi) // 2
    for a, b in ((rlo, rm), (rm, rhi)):
        d2 = [p.grad for p in d.parameters()]
        gs = torch.autograd.grad(d(audio[a:b], times=times[a:b], noise=noise[a:b]) / (2 * world), list(d.parameters()), allow_unused=True)
        gs = [g if g is not None else torch.zeros_like(p) for g, p in zip(gs, d.parameters())]
        ref_acc = gs if ref_acc is None else [x + y for x, y in zip(ref_acc, gs)]
err3 = max(float((p.grad - r).abs().max()) for p, r in zip(d.parameters(), ref_acc))
assert err3 < 1e-5, f"rank {rank}: accumulated gradients differ: {err3}"
# a second reducing backward without finish() must raise, not drop gradients
red.zero_grad()
d(audio[lo:hi], times=times[lo:hi], noise=noise[lo:hi]).backward()
try:
    d(audio[lo:hi], times=times[lo:hi], noise=noise[lo:hi]).backward()
    raise SystemExit("second reducing backward did not raise")
except RuntimeError as e:
    assert "accumulate" in str(e)
red.finish()
# ranks that differ in which parameters got a gradient still issue the same collectives in the same order
red.zero_grad()
loss = d(audio[lo:hi], times=times[lo:hi], noise=noise[lo:hi])
if rank == 0:
    loss.backward()
else:                                       # rank 1: the last layer's parameters receive no gradient from this graph
    frozen = [p for n, p in d.named_parameters() if "to_pred" in n]
    for p in frozen: p.requires_grad_(False)
    d(audio[lo:hi], times=times[lo:hi], noise=noise[lo:hi]).backward()
    for p in frozen: p.requires_grad_(True)
red.finish()
assert all(torch.isfinite(p.grad).all() for p in d.parameters())
dist.barrier()
if rank == 0:
    print("OK", world, len(red.buckets), err)
dist.destroy_process_group()
'''


def test_gradient_allreduce_world2_gloo(tmp_path):
    """SURVEY §8f-4, the collective half: rank-sharded losses + bucketed all-reduce during backward == the single-process gradient
    of the mean loss (world size 2 over gloo here; RCCL on the GPUs)."""
    script = tmp_path / "dp_worker.py"
    script.write_text(DP_WORKER)
    port = 29950 + os.getpid() % 40
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script), ROOT]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "OK 2" in r.stdout


class _FakeLstmLib:
    """stands in for libns2hip in the host-logic test of seanet._lstm: records which recurrence entry points ran, reports the
    scripted availability / give-up outcomes, and writes a marker into `out` so the caller's choice is visible"""

    def __init__(self, fused_rc, aborts):
        self.fused_rc, self.aborts, self.calls, self.rows = fused_rc, list(aborts), [], []

    def ns2_lstm2_state_floats(self):
        return 16

    def ns2_lstm_state_floats(self, B, H):
        return 10 ** 6

    def ns2_lstm2(self, *a):
        self.calls.append("fused")
        self.rows.append(a[-3])                                      # B of this launch
        return self.fused_rc

    def ns2_lstm_layer(self, xproj, ldx, whh, bhh, state, nstate, *rest):
        self.calls.append("frame" if nstate < 10 ** 6 else "layer")
        self.rows.append(rest[-4])
        return 0

    def ns2_lstm_abort_count(self, reset, ref):
        ref._obj.value = self.aborts.pop(0)
        return 0

    def ns2_last_error(self):
        return b""


@pytest.mark.parametrize("fused_rc,aborts,expect", [
    (0, [0], ["fused"]),                                           # the normal case: one launch, nothing gave up
    (1, [0], ["fused", "layer", "layer"]),                         # NS2_UNAVAILABLE (nothing launched): one launch per layer
    (0, [1, 0], ["fused", "layer", "layer"]),                      # the two-layer launch gave up: its output is discarded
    (0, [1, 2, 0], ["fused", "layer", "layer", "frame", "frame"]),  # ... and so did a per-layer launch: one launch per frame
])
def test_seanet_lstm_fallback_chain_host_logic(monkeypatch, fused_rc, aborts, expect):
    """seanet._lstm without a GPU: both layers in one launch -> one launch per layer -> one launch per frame, moving on when the
    entry point reports NS2_UNAVAILABLE or ns2_lstm_abort_count says a launch gave up (its output must not be used)"""
    import warnings
    from naturalspeech2_pytorch_amd import seanet, ops
    fake = _FakeLstmLib(fused_rc, aborts)
    monkeypatch.setattr(seanet._lib, "load", lambda: fake)
    monkeypatch.setattr(seanet, "_prep", lambda x, B, T, C, **kw: x)
    monkeypatch.setattr(seanet, "_stream", lambda: 0)
    monkeypatch.setattr(ops, "linear_f32", lambda w, a, **kw: torch.zeros(a.shape[0], 2048))
    B, T, H = 2, 5, 512
    z = torch.zeros(1)
    layers = [dict(w_ih=None, b_ih=z, w_hh=z, b_hh=z, w_ih_f32=z) for _ in range(2)]
    net = seanet._SEANetHIP.__new__(seanet._SEANetHIP)
    torch.nn.Module.__init__(net)
    net.precision = "exact"
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        act = net._lstm(seanet._Act(torch.zeros(B * T, H), B, T, H, 0), dict(layers=layers, H=H))
    assert fake.calls == expect and fake.aborts == []
    assert len([x for x in w if "gave up waiting" in str(x.message)]) == len(aborts) - 1
    assert (act.B, act.T, act.C, act.prefix) == (B, T, H, 0) and act.x.shape == (B * T, H)


def test_seanet_lstm_chunks_large_batches(monkeypatch):
    """more than 32 utterances: the one-launch recurrences run per chunk of 32 batch rows, the per-frame form takes them all"""
    from naturalspeech2_pytorch_amd import seanet, ops
    fake = _FakeLstmLib(0, [1, 1, 0])
    monkeypatch.setattr(seanet._lib, "load", lambda: fake)
    monkeypatch.setattr(seanet, "_prep", lambda x, B, T, C, **kw: x)
    monkeypatch.setattr(seanet, "_stream", lambda: 0)
    monkeypatch.setattr(ops, "linear_f32", lambda w, a, **kw: torch.zeros(a.shape[0], 2048))
    B, T, H = 70, 3, 512
    z = torch.zeros(1)
    layers = [dict(w_ih=None, b_ih=z, w_hh=z, b_hh=z, w_ih_f32=z) for _ in range(2)]
    net = seanet._SEANetHIP.__new__(seanet._SEANetHIP)
    torch.nn.Module.__init__(net)
    net.precision = "exact"
    with pytest.warns(UserWarning):
        net._lstm(seanet._Act(torch.zeros(B * T, H), B, T, H, 0), dict(layers=layers, H=H))
    assert fake.calls == ["fused"] * 3 + ["layer"] * 6 + ["frame"] * 2
    assert fake.rows == [32, 32, 6] + [32, 32, 6] * 2 + [70, 70]


def test_seanet_resblock_concatenated_weight(monkeypatch):
    """_pack_resblock: conv2(elu(h)) + shortcut(x) as ONE matrix over [elu(h) | x] in 32-column blocks (HFENC:268-301) -- the
    packed matrix applied to the concatenated operand equals the two convolutions of HF's block"""
    tf = pytest.importorskip("transformers")
    from transformers.models.encodec.modeling_encodec import EncodecResnetBlock
    from naturalspeech2_pytorch_amd import seanet, ops
    held = {}

    class Holder:                                                   # keeps the fp32 matrix a PackedWeight would pack
        def __init__(self, w, precision=3):
            self.w = w
            self.rows, self.cols = w.shape[0], w.shape[1]
    monkeypatch.setattr(ops, "PackedWeight", Holder)
    torch.manual_seed(0)
    cfg = tf.EncodecConfig()
    blk = EncodecResnetBlock(cfg, dim=48, dilations=[1, 1]).eval()
    net = seanet._SEANetHIP.__new__(seanet._SEANetHIP)
    torch.nn.Module.__init__(net)
    net.precision = "exact"
    p = net._pack_resblock(blk)
    cat = p["cat"]
    assert (cat["hs"], cat["xs"], cat["co"]) == (32, 64, 48)
    x = torch.randn(48, 30)
    with torch.no_grad():
        convs = [l for l in blk.block if not isinstance(l, torch.nn.ELU)]
        h = convs[0](torch.nn.functional.elu(x[None]))              # [1, 24, 30]
        ref = blk(x[None])[0].t()                                   # [30, 48]
        both = torch.zeros(30, cat["hs"] + cat["xs"])
        both[:, :24] = torch.nn.functional.elu(h[0]).t()
        both[:, cat["hs"]:cat["hs"] + 48] = x.t()
        got = both @ cat["w"].w.t() + cat["b"]
    assert torch.allclose(got, ref, atol=1e-5), (got - ref).abs().max()


def test_training_loss_matches_the_reference_golden():
    """tests/golden/loss_uncond_d64.pt: `NaturalSpeech2.forward(latents)` of the UNMODIFIED reference (NS2:1503-1684) for the three
    objectives with and without the min-SNR weight, random times / noise injected (make_golden.py gen_loss).  This package's
    `NaturalSpeech2.forward` with the same times / noise, the composite forward on CPU."""
    fix = torch.load(os.path.join(GOLD, "loss_uncond_d64.pt"), weights_only=False)
    m = Model(**fix["kwargs"])
    m.load_state_dict(make_weights(fix["shapes"], seed=fix["weight_seed"]))
    b, n = fix["batch"], fix["n"]
    audio = make_input("audio", (b, n, 64), seed=fix["input_seed"])
    times = make_input("times", (b,), seed=fix["input_seed"], uniform=True)
    noise = make_input("noise", (b, n, 64), seed=fix["input_seed"])
    assert len(fix["losses"]) == 6
    for key, ref in fix["losses"].items():
        objective, ms = key.split("/")
        d = NaturalSpeech2(m, codec=None, target_sample_hz=24000, timesteps=10, objective=objective, min_snr_loss_weight=(ms == "min_snr=True"))
        loss = d(audio, times=times, noise=noise)
        assert abs(loss.item() - ref) < 2e-5 * max(1.0, abs(ref)), (key, loss.item(), ref)


def test_sampler_demotes_to_exact_when_a_fast_mode_leaves_the_half_range():
    """diffusion.ddim_sample(on_saturation=...): host logic of the range guard's last resort, with the HIP call substituted (the
    unfused loop: a non-default schedule).  A fast mode whose forward reports clamped activations is repeated ONCE, from the same
    initial latents, in `exact`, and the model stays there; "raise" keeps the error; any other error is not retried."""
    import warnings
    m = Model(dim=64, depth=1, precision="hybrid").eval()
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    calls = []

    def fake_hip(self, x, times, prompt=None, prompt_mask=None, cond=None, cond_drop_prob=None, out=None, cond_row=None):
        calls.append(self.precision)
        if self.precision != "exact" and len(calls) == 2:
            raise _lib.Ns2Error(fake_hip.message)
        return O.model_forward(sd, x, times)

    fake_hip.message = "3 conversions left the IEEE-half range (precision hybrid)"
    d = NaturalSpeech2(m, codec=None, target_sample_hz=24000, timesteps=3, schedule_kwargs=dict(start=-3.0, end=3.0, tau=1.0))
    assert not d._fused_ddim_ok()
    noise = make_input("noise", (2, 12, 64), seed=8)
    try:
        type(m)._forward_hip = fake_hip
        with pytest.warns(UserWarning, match="repeating the sampling run with precision='exact'"):
            out = d.ddim_sample((2, 12, 64), noise=noise)
        assert m.precision == "exact" and calls == ["hybrid", "hybrid", "exact", "exact", "exact"]
        calls.clear()
        with warnings.catch_warnings():
            warnings.simplefilter("error")                                  # already exact: a clean run, no warning
            again = d.ddim_sample((2, 12, 64), noise=noise)
        assert torch.equal(out, again) and calls == ["exact"] * 3           # the repeated run started from the SAME latents
        m.precision = "mixed"
        calls.clear()
        with pytest.raises(_lib.Ns2Error):
            d.ddim_sample((2, 12, 64), noise=noise, on_saturation="raise")
        assert m.precision == "mixed" and calls == ["mixed", "mixed"]
        fake_hip.message = "workspace too small"
        calls.clear()
        with pytest.raises(_lib.Ns2Error):
            d.ddim_sample((2, 12, 64), noise=noise)
        assert m.precision == "mixed" and calls == ["mixed", "mixed"]       # not a range error: no second attempt
    finally:
        del type(m)._forward_hip


def test_prompt_mask_is_rejected_on_every_path_like_upstream():
    """`prompt_mask` is in the reference's signature but the reference raises on it (shape error in the resampler's attention:
    test_compat_reference.py runs that); every path of this package rejects it with the reason instead of guessing a semantics"""
    import pytest
    import torch
    from naturalspeech2_pytorch_amd import Model
    m = Model(dim=64, depth=1, dim_prompt=64, condition_on_prompt=True)
    x, t = torch.randn(2, 16, 64), torch.rand(2)
    prompt, cond = torch.randn(2, 10, 64), torch.randn(2, 64, 16)
    mask = torch.ones(2, 10, dtype=torch.bool)
    with pytest.raises(NotImplementedError, match="reference itself raises"):
        m(x, t, prompt=prompt, prompt_mask=mask, cond=cond)                     # autograd path (parameters require grad)
    with torch.no_grad(), pytest.raises(NotImplementedError, match="reference itself raises"):
        m(x, t, prompt=prompt, prompt_mask=mask, cond=cond)                     # inference path: rejected before any device work


def test_eval_module_without_no_grad_warns_once_and_force_autograd_selects_the_composite():
    """eval() does not switch autograd off in PyTorch; a per-utterance module that keeps its HIP kernels there says so"""
    import warnings
    import torch
    from naturalspeech2_pytorch_amd import transformer as T
    tr = T.Transformer(dim=64, depth=1).eval()
    x = torch.randn(1, 8, 64)
    T._warned_eval_no_graph = False
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert T.needs_autograd(tr, x) is False
        assert T.needs_autograd(tr, x) is False
    assert len([i for i in w if "no autograd graph" in str(i.message)]) == 1
    assert T.needs_autograd(tr, x.clone().requires_grad_(True)) is True
    tr.force_autograd = True
    assert T.needs_autograd(tr, x) is True
    tr.force_autograd = False
    assert T.needs_autograd(tr.train(), x) is True
    with torch.no_grad():
        assert T.needs_autograd(tr, x) is False


def test_seanet_narrow_resblock_weight_arrangement(monkeypatch):
    """_pack_resblock's operands for ns2_seanet_resblock_narrow (w1p [t][c][h], w2p [h][c], wsp [c][o], b2s): the kernel's arithmetic
    restated with them -- reflect-padded k = 3 conv on elu(x), elu, 1 x 1 conv, + 1 x 1 shortcut of x -- equals HF's block (HFENC:268-301)"""
    tf = pytest.importorskip("transformers")
    from transformers.models.encodec.modeling_encodec import EncodecResnetBlock
    from naturalspeech2_pytorch_amd import seanet, ops

    class Holder:
        def __init__(self, w, precision=3):
            self.w = w
    monkeypatch.setattr(ops, "PackedWeight", Holder)
    torch.manual_seed(1)
    blk = EncodecResnetBlock(tf.EncodecConfig(), dim=32, dilations=[1, 1]).eval()
    net = seanet._SEANetHIP.__new__(seanet._SEANetHIP)
    torch.nn.Module.__init__(net)
    net.precision = "exact"
    nr = net._pack_resblock(blk)["narrow"]
    assert nr["C"] == 32 and nr["w1p"].shape == (3, 32, 16) and nr["w2p"].shape == (16, 32) and nr["wsp"].shape == (32, 32)
    T = 21
    x = torch.randn(T, 32)
    with torch.no_grad():
        ref = blk(x.t()[None])[0].t()                                # [T, 32]
        ex = torch.nn.functional.elu(x)
        idx = lambda n: n if n >= 0 else -n                          # noqa: E731  (reflect: x[-i] = x[i])
        h = torch.stack([nr["b1"] + sum(ex[idx(n - 2 + t)] @ nr["w1p"][t] for t in range(3)) for n in range(T)])
        got = nr["b2s"] + torch.nn.functional.elu(h) @ nr["w2p"] + x @ nr["wsp"]
    assert torch.allclose(got, ref, atol=1e-5), (got - ref).abs().max()


def test_wgrad_route_and_repack_table_host_arithmetic():
    """host-only entry points of the round-5 training ABI: which weight gradients take the row-plane (transposed-read) kernel -- the
    headline model's all do, the dim = 128 model's and short batches keep the transposed route where launch_gemm picks the 128 x 128
    kernel -- and the size of the one-launch re-pack table"""
    lib = _lib.load()
    M = 32 * 1024
    for R, ncols in ((512, 512), (512, 3 * 512), (1536, 512), (2730, 512), (512, 1376), (1365, 3 * 1376)):
        assert lib.ns2_wgrad_rows_preferred(R, ncols, M) == 1, (R, ncols)
    assert lib.ns2_wgrad_rows_preferred(128, 512, 4096) == 0            # R = 128: half-empty 256-row tiles
    assert lib.ns2_wgrad_rows_preferred(512, 128, M) == 0               # one 128-column strip
    assert lib.ns2_wgrad_rows_preferred(512, 512, 256) == 0             # a handful of tokens: <= 64 blocks
    assert lib.ns2_wgrad_rows_preferred(512, 512, 0) == 0
    assert lib.ns2_weights_repack_table_bytes(0) == 0
    b1, b270 = lib.ns2_weights_repack_table_bytes(1), lib.ns2_weights_repack_table_bytes(270)
    assert b1 > 0 and b270 == 270 * b1
    # workspace of the row-plane route = the transposed route's (same slots): R x ncols floats per slice
    assert lib.ns2_wgrad_workspace_bytes(512, 512, M) % (512 * 512 * 4) == 0


def test_lds_transpose_read_layouts_deliver_the_fragments_the_mfma_expects():
    """A pure-Python model of the two kernels that read their operands TRANSPOSED out of LDS (round 5), against the probed semantics of
    gfx950's transpose reads (profiles/r05_lds_transpose_read_probe.txt):
        ds_read_b64_tr_b16: lane i of a 16-lane group receives, as element j, the 16-bit word (i % 4) at the address lane 4 j + i // 4 supplied
        ds_read_b64_tr_b8:  ... as byte j, the byte (i % 8) at the address lane 2 j + i // 8 supplied.
    (1) gemm2.hip TR (weight gradients): the DMA lane -> (token, chunk) assignment, the eight-block piece image and the read offsets must hand
    lane (tg, i16, hi) the tokens 16 kc + 8 hi + 4 q .. + 3 (16-bit) / 8 q .. + 7 (bytes) of channel 16 tg + i16.
    (2) backward.hip attn_bwd2: chunk c of tile row r is stored at c ^ f(r); the 16 rows of a ds_read_b128 group and the 4 rows of a
    transpose block must land on distinct bank groups, and the transpose read must return walked rows .. + 3 of the lane's d column."""
    # ---------------- (1) the TR piece image, FMT_H8 (mixed) and bf16 hi / lo (exact)
    for nsplit in (2, 3):
        lds = {}                                            # byte offset inside the 4 pieces of a channel group -> (token, byte of the source line)
        for piece in range(4):                              # tokens 8 piece .. + 7
            for lane in range(64):
                b, w = lane >> 3, lane & 7
                bytes_blk = nsplit == 2 and b >= 4
                tokp = w if bytes_blk else 4 * ((b >> 1) & 1) + (w >> 1)
                lchunk = b if bytes_blk else ((4 * (b >> 2) if nsplit == 3 else 0) + 2 * (b & 1) + (w & 1))
                for k in range(16):
                    lds[piece * 1024 + lane * 16 + k] = (8 * piece + tokp, lchunk * 16 + k)
        assert len(lds) == 4096 and len(set(lds.values())) == 4096          # every byte of the 32 token lines exactly once
        for lane in range(64):
            i16, tg, hi = lane & 15, (lane >> 4) & 1, lane >> 5
            grp = [l for l in range(64) if l >> 4 == lane >> 4]
            for p in range(2 if nsplit == 3 else 1):
                for kc in range(2):
                    for q in range(2):
                        addr = {l: (l >> 5) * 1024 + (4 * p + 2 * q + ((l >> 4) & 1)) * 128 + (l & 15) * 8 + kc * 2048 for l in grp}
                        for j in range(4):
                            src = addr[grp[4 * j + i16 // 4]] + 2 * (i16 % 4)
                            tok, byte = lds[src]
                            assert tok == 16 * kc + 8 * hi + 4 * q + j and byte == p * 64 + 2 * (16 * tg + i16), (nsplit, lane, p, kc, q, j)
            if nsplit == 2:
                for ws in range(2):
                    part = (1 - hi) if ws else hi            # A: lanes 0-31 h8, 32-63 l8;  W: the other way round
                    for q in range(4):
                        addr = {l: (4 + 2 * ((1 - (l >> 5)) if ws else (l >> 5)) + ((l >> 4) & 1)) * 128 + (l & 15) * 8 + q * 1024 for l in grp}
                        for j in range(8):
                            tok, byte = lds[addr[grp[2 * j + i16 // 8]] + i16 % 8]
                            assert tok == 8 * q + j and byte == 64 + 32 * part + 16 * tg + i16, (lane, ws, q, j)
    # ---------------- (2) the attention backward tile image
    f = lambda r: ((r & 3) << 2) | ((r >> 2) & 3)           # noqa: E731
    tile = {}                                               # byte offset -> (row, byte of the row's 256 source bytes)
    for j in range(16):                                     # DMA instruction j: rows 4 j .. + 3
        for lane in range(64):
            row, pos = 4 * j + (lane >> 4), lane & 15
            lc = pos ^ f(row)
            for k in range(16):
                tile[j * 1024 + lane * 16 + k] = (row, lc * 16 + k)
    assert len(set(tile.values())) == 64 * 256
    for r0 in range(0, 64, 16):                             # a ds_read_b128 group: 16 rows, the same logical chunk -> 16 distinct positions
        for c in range(16):
            assert len({(c ^ f(r)) for r in range(r0, r0 + 16)}) == 16
    for r0 in range(0, 64, 4):                              # a transpose block: 4 rows x 4 adjacent chunks -> the four 64-byte quarters
        for c0 in range(0, 16, 4):
            assert len({(c0 ^ f(r)) >> 2 for r in range(r0, r0 + 4)}) == 4
    for lane in range(64):
        i16, tg, hi = lane & 15, (lane >> 4) & 1, lane >> 5
        grp = [l for l in range(64) if l >> 4 == lane >> 4]
        for js in range(2):
            for g1 in range(2):
                for dt in range(2):
                    for p in range(2):
                        for q in range(2):
                            def addr(l):
                                row = 32 * js + 16 * g1 + 8 * (l >> 5) + 4 * q + ((l & 15) >> 2)
                                lc = dt * 8 + p * 4 + 2 * ((l >> 4) & 1) + (((l & 15) >> 1) & 1)
                                return row * 256 + ((lc ^ f(row)) << 4) + (l & 1) * 8
                            for j in range(4):
                                row, byte = tile[addr(grp[4 * j + i16 // 4]) + 2 * (i16 % 4)]
                                d = 16 * tg + i16                      # the lane's d column inside the 32-column half dt
                                assert row == 32 * js + 16 * g1 + 8 * hi + 4 * q + j and byte == dt * 128 + p * 64 + 2 * d, (lane, js, g1, dt, p, q, j)
