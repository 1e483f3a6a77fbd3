"""Run the roofline kernels a few times each (for rocprofv3 --pmc passes): expert-L1 forward GEMM at the bench shape
(fp32 MFMA and split-bf16), the table L2 pass, the fused table optimizer pass and the embedding gather."""
import ctypes as ct
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aread_amd
from aread_amd import _lib as L
from aread_amd import presets
from tools import synth

spec = presets.amazon_workload(0.2)
rng = np.random.default_rng(0)
M, N, K = 9728, 1024, 288
A = torch.randn(M * K, device="cuda"); B = torch.randn(N * K, device="cuda"); C = torch.empty(M * N, device="cuda")
R = spec.n_table_rows
table = torch.randn(R * 32, device="cuda") * 0.5
grad = torch.empty_like(table)
part = torch.empty(L.lib().aread_l2_partials(), device="cuda")
x, _ = synth.amazon_batch(spec, rng, 8192)
xs = torch.from_numpy(x).cuda()
off = torch.from_numpy(np.concatenate([[0], np.cumsum(spec.field_dims)[:-1]]).astype(np.int32)).cuda()
out = torch.empty(8192 * 9 * 32, device="cuda")
from aread_amd.optim import AdamCfg
cfg = AdamCfg(); cfg.lr, cfg.beta1, cfg.beta2, cfg.eps, cfg.weight_decay, cfg.step = 1e-3, 0.9, 0.99, 1e-8, 1e-8, 10
mom, var = torch.zeros_like(table), torch.zeros_like(table)
dz = torch.randn(M * N, device="cuda"); slab = torch.empty(16 * N * K, device="cuda"); kc = M // 16
for _ in range(5):
    L.check(L.lib().aread_gemm(L.ptr(dz), N, kc * N, 0, L.ptr(A), K, kc * K, 0, L.ptr(slab), K, N * K, None, 0, N, K, kc, 16, 0, L.stream()))
    L.check(L.lib().aread_gemm_bf16x3_rc(L.ptr(dz), N, kc * N, L.ptr(A), K, kc * K, L.ptr(slab), K, N * K, N, K, kc, 16, 0, L.stream()))
    L.check(L.lib().aread_gemm_bf16x3(L.ptr(A), K, M * K, L.ptr(B), K, N * K, L.ptr(C), N, M * N, None, 0, M, N, K, 1, 0, L.stream()))
    L.check(L.lib().aread_adam_table_l2(L.ptr(table), L.ptr(mom), L.ptr(var), R, 32, None, None, None, None, 1e-5, ct.byref(cfg), 0,
                                        L.ptr(part), L.stream()))
    L.check(L.lib().aread_gemm(L.ptr(A), K, M * K, 1, L.ptr(B), K, N * K, 1, L.ptr(C), N, M * N, None, 0, M, N, K, 1, 0, L.stream()))
    L.check(L.lib().aread_l2_table(L.ptr(table), table.numel(), 1e-5, 1.0, L.ptr(grad), L.ptr(part), L.stream()))
    L.check(L.lib().aread_embed_fwd(L.ptr(xs), 8192, 17, L.ptr(off), L.ptr(table), R, 32, 7, 2, 5, 2, None, 8192, L.ptr(out), None,
                                    L.stream()))
torch.cuda.synchronize()
# a few whole training steps: the fused tower kernels (k_tower_fwd / k_tower_bwd) and everything else at the bench shape
model = presets.build_model(spec, "cuda", precision="bf16x3"); model.train()
masks = presets.random_masks(model, 0.7, seed=2000)
md = aread_amd.pack_masks(masks, 25, model.edge_num, "cuda")
xb, yb = synth.amazon_batch(spec, rng, 8192)
xbs, ybs = torch.from_numpy(xb).cuda(), torch.from_numpy(yb).cuda()
bufs = model.make_step_buffers(8192)
for _ in range(4):
    model.train_step(xbs, ybs, bufs, masks_dev=md, set_grads=False)
torch.cuda.synchronize()
