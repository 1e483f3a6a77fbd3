"""Experiment: expert-L1 input gradient (M 9728, N 288, K 1024, both operands k-contiguous, split-bf16) as ONE GEMM vs the K axis
cut into S slices run as S groups of the grouped GEMM (a_gs = b_gs = K/S, every slice writes its own output buffer)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from aread_amd import _lib as L
from tools.gemm_bench import timeit

M, N, K = 9728, 288, 1024
A = torch.randn(M, K, device="cuda")
W = torch.randn(N, K, device="cuda")          # transposed weights [in][out]: k-contiguous
for S in (1, 2, 4):
    C = torch.empty(S, M, N, device="cuda")
    fn = lambda: L.check(L.lib().aread_gemm_bf16x3(L.ptr(A), K, K // S, L.ptr(W), K, K // S, L.ptr(C), N, M * N, None, 0,
                                                   M, N, K // S, S, 0, L.stream()))
    t = timeit(fn, iters=40)
    ref = A @ W.t()
    err = float((C.sum(0) - ref).abs().max() / ref.abs().max())
    print(f"K slices {S}: {t:7.1f} us   (max rel err of the summed slices {err:.2e})")
