"""TEST INFRASTRUCTURE ONLY — CPU oracle for the EnCodec residual-VQ encode (codebook-distance argmax).

The arithmetic lives in third-party packages that are neither vendored under /root/reference nor
installed: `audiolm-pytorch>=0.30.2` (EncodecWrapper; setup.py:24) -> PyPI `encodec`
(`quantization/core_vq.py`: EuclideanCodebook.quantize, ResidualVectorQuantization.encode/decode);
no version is pinned beyond that lower bound.  Reference call sites: NS2:1445, NS2:1611 (encode),
NS2:1496 (decode), NS2:1682 (rq).  The in-container stand-in is HF transformers 5.15.0
`modeling_encodec.py` (HFENC), a line-for-line restatement of encodec's core_vq:
    HFENC:364-369  quantize:  dist = -(x.pow(2).sum(1,keepdim) - 2*x@E.T + E.pow(2).sum(0,keepdim)); argmax (first max)
    HFENC:424-438  encode:    residual -= embedding(idx) per quantizer, indices stacked [q, ...]
    HFENC:440-447  decode:    quantized_out = 0.0 + sum_q embedding(idx_q)   (fp32, in quantizer order)
`tests/golden/make_golden.py` pins this file against HFENC's classes run in the build container;
the reference repo itself holds no test or golden vector for this boundary ("parity unpinned" upstream).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
"""
import torch

Tensor = torch.Tensor


def quantize(x: Tensor, embed: Tensor) -> Tensor:
    """HFENC:364-369.  x: [m, d]; embed: [codes, d] -> idx [m] (int64, first max on ties)."""
    e = embed.t()
    xx = x.pow(2).sum(1, keepdim=True)
    dist = -(xx - 2 * x @ e + e.pow(2).sum(0, keepdim=True))
    return dist.max(dim=-1).indices


def rvq_encode(x: Tensor, codebooks: Tensor):
    """HFENC:424-447.  x: [m, d] latents; codebooks: [q, codes, d].
    Returns codes [m, q] int64, emb [m, d] (sum of selected code vectors), final residual [m, d]."""
    residual = x
    idxs = []
    emb = torch.zeros((), dtype=x.dtype)
    for q in range(codebooks.shape[0]):
        idx = quantize(residual, codebooks[q])
        quantized = torch.nn.functional.embedding(idx, codebooks[q])
        residual = residual - quantized
        emb = emb + quantized
        idxs.append(idx)
    return torch.stack(idxs, dim=-1), emb, residual


def top2_margins(x: Tensor, codebooks: Tensor, codes: Tensor) -> Tensor:
    """fp64 margin between the best and second-best squared distance along the path `codes` took.
    Used by tests to classify any index mismatch as a genuine error or an fp32 near-tie. -> [m, q]."""
    x = x.double()
    cb = codebooks.double()
    out = []
    residual = x
    for q in range(cb.shape[0]):
        d = torch.cdist(residual, cb[q]).pow(2)
        s = d.sort(dim=-1).values
        out.append(s[:, 1] - s[:, 0])
        residual = residual - cb[q][codes[:, q]]
    return torch.stack(out, dim=-1)
