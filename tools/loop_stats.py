"""Instruction mix of the hottest loop of a kernel in an AMDGPU assembly listing (hipcc -S --cuda-device-only):
    python tools/loop_stats.py gemm2.s <mangled kernel name> [...]
The loop = the basic blocks LLVM annotates with the same `Header=` that hold the most MFMAs."""
import re, sys
from collections import Counter, defaultdict

src = open(sys.argv[1]).read().split("\n")
for kname in sys.argv[2:]:
    try:
        a = next(i for i, l in enumerate(src) if l.startswith(kname + ":"))
    except StopIteration:
        print(kname, "not found"); continue
    b = next(i for i in range(a, len(src)) if ".end_amdhsa_kernel" in src[i])
    body = src[a:b]
    meta = {k: next((l.split()[-1] for l in body if k in l), "?") for k in (".amdhsa_next_free_vgpr", ".amdhsa_next_free_sgpr")}
    spills = [next((l for l in src[b:b + 80] if k in l), "") for k in ("; SGPRSpill", "; ScratchSize", "; sgpr_spill_count", "; vgpr_spill_count")]
    loops = defaultdict(Counter)
    cur = None
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            h = re.search(r"Header=BB(\d+_\d+)", l)
            cur = ("BB" + h.group(1)) if h else (m.group(1)[2:] if "Loop Header" in l else None)
            if "Loop Header" in l:
                cur = m.group(1)[2:]
            continue
        if re.match(r"^; %bb", l):
            h = re.search(r"Header=BB(\d+_\d+)", l)
            cur = ("BB" + h.group(1)) if h else None
            continue
        s = l.strip()
        if cur and s and not s.startswith(";") and not s.startswith("."):
            loops[cur][s.split()[0]] += 1
    if not loops:
        print(kname, "no loops"); continue
    hot = max(loops, key=lambda k: sum(v for n, v in loops[k].items() if n.startswith("v_mfma")))
    c = loops[hot]
    mf = sum(v for n, v in c.items() if n.startswith("v_mfma"))
    tot = sum(c.values())
    br = sum(v for n, v in c.items() if n.startswith("s_cbranch") or n == "s_branch")
    lane = sum(v for n, v in c.items() if n in ("v_readlane_b32", "v_writelane_b32"))
    print(f"{kname}\n  vgprs {meta['.amdhsa_next_free_vgpr']}  hottest loop {hot}: {tot} instructions, {mf} MFMAs, {(tot - mf) / max(mf, 1):.2f} other instructions per MFMA, "
          f"{br} branches, {lane} v_readlane / v_writelane")
    print("  " + ", ".join(f"{n} {v}" for n, v in c.most_common(14)))
