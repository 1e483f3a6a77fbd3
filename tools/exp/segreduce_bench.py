"""Experiment: the table-gradient segmented reduction (k_segreduce level 1 + final) alone, on the step's batch shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import aread_amd
from aread_amd import presets
from tools import synth
from tools.gemm_bench import timeit

spec = presets.amazon_workload(0.2)
rng = np.random.default_rng(0)
model = presets.build_model(spec, "cuda", precision="bf16x3")
emb = model.embedding
x, y = synth.amazon_batch(spec, rng, 8192)
xd = torch.from_numpy(x).cuda()
plan = aread_amd.RowPlan(xd, model.domain_idx, 25)
emb._ws_for(xd)
emb.sort_lookups(xd, plan.sample_row)
de = torch.randn((plan.max_rows, model.embed_output_dim), device="cuda")
de2 = torch.randn_like(de)
grad = torch.zeros_like(emb.embedding_dict.weight)
t1 = timeit(lambda: emb.reduce_sorted(xd, de, grad), iters=40)
t2 = timeit(lambda: emb.reduce_sorted(xd, de, grad, dout2=de2), iters=40)
print(f"reduce_sorted: {t1:.1f} us (one source), {t2:.1f} us (two sources)")
