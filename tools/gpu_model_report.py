"""Runs the model-level parity cases without stopping at the first failure; writes gpurun_out/model_report.json."""
import json
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from model_cases import MODEL_CASES  # noqa: E402

only = [a for a in sys.argv[1:] if not a.startswith("-")]
out = []
for name, fn in MODEL_CASES:
    if only and name not in only:
        continue
    try:
        r = fn()
    except Exception as ex:  # noqa: BLE001
        r = {"name": name, "ok": False, "error": repr(ex), "trace": traceback.format_exc()[-1500:]}
    r["case"] = name
    out.append(r)
    print(("PASS " if r.get("ok") else "FAIL ") + json.dumps(r, default=str)[:900], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "model_report.json"), "w") as f:
    json.dump(out, f, indent=1, default=str)
