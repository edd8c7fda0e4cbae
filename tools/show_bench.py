import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
s=d.get("side",{})
for k,v in (s.get("train_step") or {}).items():
    if isinstance(v,dict): print(k, {q:v.get(q) for q in ("ms_per_step","mixed_ms_per_step","exact_ms_per_step","timed_windows_ms","eager_ms_per_step","pytorch_composite_ms_per_step","loss_mixed","loss_exact","loss_composite")})
    else: print(k, v)
print(json.dumps(s.get("small_batch"),indent=0)[:2500])
print(json.dumps(d.get("parity"),indent=0)[:1500])
