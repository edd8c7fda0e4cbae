"""Fold gpurun_out/parity_r6.json (written by the GPU tests on the box) into the tracked profiles/r06_parity.json, key by key.
Keys absent from the scratch file are kept; nothing is ever dropped.   python tools/merge_parity.py [scratch.json]"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests.parity_record import SCRATCH, TRACKED, _load

src = sys.argv[1] if len(sys.argv) > 1 else SCRATCH
new, old = _load(src), _load(TRACKED)
if not new:
    sys.exit(f"{src}: nothing to merge")
added = [k for k in new if k not in old and k != "_meta"]
changed = [k for k in new if k in old and k != "_meta" and new[k] != old[k]]
upd = set(old.get("_meta", {}).get("updated_keys_r06", [])) | set(new.get("_meta", {}).get("updated_keys_r06", []))
for k, v in new.items():
    if k != "_meta":
        old[k] = v
old.setdefault("_meta", {})["updated_keys_r06"] = sorted(upd)
json.dump(old, open(TRACKED, "w"), indent=1, sort_keys=True)
print(f"merged {src} -> {TRACKED}: {len(added)} new keys, {len(changed)} changed, {len(old) - 1} total")
for k in added: print("  +", k)
for k in changed: print("  ~", k)
