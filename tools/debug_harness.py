import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import util as U
spec = U.spec_full()
G = U.load_golden("harness.npz")
model, _ = U.build_model(spec, 123, device="cuda")
model.device = torch.device("cpu")
out = U.run_harness(model, spec, torch.device("cuda"))
tags, got, ref = G["trace_tags"], out["trace_vals"], G["trace_vals"]
for i, (t, a, b) in enumerate(zip(tags, got, ref)):
    rel = abs(a - b) / max(abs(b), 1e-9)
    flag = " <<<" if rel > 1e-4 else ""
    print(f"{i:3d} {t:12s} got {a:.6f} ref {b:.6f} rel {rel:.2e}{flag}")
