"""Parity evidence record.  GPU tests call record(key, value); values are MERGED key by key, never written as a whole file:

  * the tracked record is profiles/r06_parity.json (TRACKED) -- what DESIGN.md and bench.py quote;
  * on a GPU box only gpurun_out/ travels back, so record() writes gpurun_out/parity_r6.json (SCRATCH), which always starts
    from the tracked record: a partial pytest run therefore carries every other key along unchanged;
  * tools/merge_parity.py folds SCRATCH back into TRACKED key by key (and lists what changed).

tests/test_host_cpu.py::test_parity_record_is_complete fails the CPU suite if the tracked record loses the keys bench.py quotes.
"""
import json
import os

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
TRACKED = os.path.join(ROOT, "profiles", "r06_parity.json")
SCRATCH = os.path.join(ROOT, "gpurun_out", "parity_r6.json")
# keys bench.py quotes in its `parity` object (bench.py refuses to print an empty sweep)
REQUIRED_KEYS = tuple(f"sweep_d512_L12/{p}" for p in ("hybrid", "mixed", "half"))


def _load(path):
    try:
        with open(path) as f:
            d = json.load(f)
        return d if isinstance(d, dict) else {}
    except Exception:
        return {}


def load_current():
    """the record as the next record() call will see it: tracked values overlaid with this box's scratch values"""
    cur, scr = _load(TRACKED), _load(SCRATCH)
    cur.update({k: v for k, v in scr.items() if k != "_meta"})
    upd = set(cur.get("_meta", {}).get("updated_keys_r06", [])) | set(scr.get("_meta", {}).get("updated_keys_r06", []))
    cur.setdefault("_meta", {})["updated_keys_r06"] = sorted(upd)
    return cur


def record(key, value):
    cur = load_current()
    cur[key] = value
    meta = cur.setdefault("_meta", {})
    upd = set(meta.get("updated_keys_r06", []))
    upd.add(key)
    meta["updated_keys_r06"] = sorted(upd)
    os.makedirs(os.path.dirname(SCRATCH), exist_ok=True)
    tmp = SCRATCH + ".tmp"
    with open(tmp, "w") as f:
        json.dump(cur, f, indent=1, sort_keys=True)
    os.replace(tmp, SCRATCH)
