"""GPU: the grouped fp32-MFMA GEMM against a float64 numpy product, all three operand layouts."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def run(A, B, bias, M, N, K, G, a_kc, b_kc, lda, ldb, a_gs, b_gs, accumulate=False, C0=None):
    from aread_amd import _lib as L
    Ad, Bd = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
    ldc = (N + 3) // 4 * 4 + 4
    C = torch.full((G, M, ldc), 7.0, device="cuda") if C0 is None else torch.from_numpy(C0).cuda()
    bd = None if bias is None else torch.from_numpy(bias).cuda()
    L.check(L.lib().aread_gemm(L.ptr(Ad), lda, a_gs, int(a_kc), L.ptr(Bd), ldb, b_gs, int(b_kc), L.ptr(C), ldc, M * ldc,
                               L.ptr(bd), N if bias is not None else 0, M, N, K, G, int(accumulate), L.stream()))
    torch.cuda.synchronize()
    return C.cpu().numpy()


@pytest.mark.parametrize("M,N,K,G", [(64, 16, 32, 1), (200, 90, 64, 1), (130, 12, 288, 1), (8192, 1024, 288, 1),
                                       (1000, 128, 256, 4), (300, 8, 16, 12), (77, 40, 44, 3), (64, 288, 12, 1)])
@pytest.mark.parametrize("layout", ["kc_kc", "kc_mc", "mc_mc"])
def test_gemm_layouts(M, N, K, G, layout):
    rng = np.random.default_rng(M * 7 + N * 3 + K + G)
    a = rng.standard_normal((G, M, K)).astype(np.float32)
    b = rng.standard_normal((G, N, K)).astype(np.float32)
    bias = rng.standard_normal((G, N)).astype(np.float32)
    ref = np.einsum("gmk,gnk->gmn", a.astype(np.float64), b.astype(np.float64)) + bias[:, None, :]
    pad = lambda n: (n + 3) // 4 * 4
    a_kc, b_kc = layout[:2] == "kc", layout[3:] == "kc"
    if a_kc:
        A = np.zeros((G, M, pad(K)), np.float32); A[:, :, :K] = a; lda, a_gs = pad(K), M * pad(K)
    else:
        A = np.zeros((G, K, pad(M)), np.float32); A[:, :, :M] = a.transpose(0, 2, 1); lda, a_gs = pad(M), K * pad(M)
    if b_kc:
        Bm = np.zeros((G, N, pad(K)), np.float32); Bm[:, :, :K] = b; ldb, b_gs = pad(K), N * pad(K)
    else:
        Bm = np.zeros((G, K, pad(N)), np.float32); Bm[:, :, :N] = b.transpose(0, 2, 1); ldb, b_gs = pad(N), K * pad(N)
    C = run(A, Bm, bias, M, N, K, G, a_kc, b_kc, lda, ldb, a_gs, b_gs)
    err = np.abs(C[:, :, :N] - ref).max() / np.abs(ref).max()
    assert err < 2e-6, err
    assert (C[:, :, N:] == 7.0).all()                                    # nothing written outside [M,N]


def test_gemm_accumulate_and_exact_small_integers():
    rng = np.random.default_rng(3)
    M, N, K = 96, 48, 40
    a = rng.integers(-3, 4, (1, M, K)).astype(np.float32)
    b = rng.integers(-3, 4, (1, N, K)).astype(np.float32)              # asymmetric, exactly representable
    ldc = N + 4
    C0 = rng.integers(-5, 5, (1, M, ldc)).astype(np.float32)
    C = run(a, b, None, M, N, K, 1, True, True, K, K, M * K, N * K, accumulate=True, C0=C0.copy())
    ref = C0.copy(); ref[:, :, :N] += np.einsum("gmk,gnk->gmn", a, b)
    np.testing.assert_array_equal(C, ref)


@pytest.mark.parametrize("M,N,K,G", [(64, 16, 32, 1), (200, 90, 64, 1), (130, 12, 288, 1), (8192, 1024, 288, 1),
                                       (1000, 128, 256, 4), (300, 8, 16, 12), (77, 40, 44, 3), (9728, 288, 1024, 1)])
def test_gemm_bf16x3(M, N, K, G):
    """split-bf16 GEMM: ~1e-6 relative error against float64 (plain bf16 would be ~3e-3)."""
    from aread_amd import _lib as L
    rng = np.random.default_rng(M + N + K + G)
    a = rng.standard_normal((G, M, K)).astype(np.float32)
    b = rng.standard_normal((G, N, K)).astype(np.float32)
    bias = rng.standard_normal((G, N)).astype(np.float32)
    pad = lambda n: (n + 3) // 4 * 4
    A = np.zeros((G, M, pad(K)), np.float32); A[:, :, :K] = a
    Bm = np.zeros((G, N, pad(K)), np.float32); Bm[:, :, :K] = b
    ldc = pad(N) + 4
    C = torch.full((G, M, ldc), 7.0, device="cuda")
    Ad, Bd, bd = torch.from_numpy(A).cuda(), torch.from_numpy(Bm).cuda(), torch.from_numpy(bias).cuda()
    L.check(L.lib().aread_gemm_bf16x3(L.ptr(Ad), pad(K), M * pad(K), L.ptr(Bd), pad(K), N * pad(K), L.ptr(C), ldc, M * ldc,
                                      L.ptr(bd), N, M, N, K, G, 0, L.stream()))
    torch.cuda.synchronize()
    C = C.cpu().numpy()
    ref = np.einsum("gmk,gnk->gmn", a.astype(np.float64), b.astype(np.float64)) + bias[:, None, :]
    err = np.abs(C[:, :, :N] - ref).max() / np.abs(ref).max()
    assert err < 2e-5, err
    assert np.sqrt(((C[:, :, :N] - ref) ** 2).mean()) / np.sqrt((ref ** 2).mean()) < 1e-5
    assert (C[:, :, N:] == 7.0).all()


@pytest.mark.parametrize("M,N,K,G", [(64, 16, 32, 1), (1024, 288, 9728, 1), (128, 256, 2432, 4), (100, 90, 70, 1), (64, 128, 640, 3),
                                       (12, 64, 1000, 1), (32, 32, 96, 6), (16, 16, 64, 12), (288, 4, 9728, 1)])
def test_gemm_bf16x3_row_contiguous(M, N, K, G):
    """weight-gradient shape: both operands row-contiguous (k-major), transposing LDS reads; ~1e-6 relative error
    against float64; exact integer products (also checks the k order / fragment layout bit for bit)."""
    from aread_amd import _lib as L
    rng = np.random.default_rng(M + N + K + G)
    pad = lambda n: (n + 3) // 4 * 4
    a = rng.standard_normal((G, K, M)).astype(np.float32)             # A[g](m,k) stored at [g][k][m]
    b = rng.standard_normal((G, K, N)).astype(np.float32)
    A = np.zeros((G, K, pad(M)), np.float32); A[:, :, :M] = a
    Bm = np.zeros((G, K, pad(N)), np.float32); Bm[:, :, :N] = b
    ldc = pad(N) + 4
    C = torch.full((G, M, ldc), 7.0, device="cuda")
    Ad, Bd = torch.from_numpy(A).cuda(), torch.from_numpy(Bm).cuda()
    L.check(L.lib().aread_gemm_bf16x3_rc(L.ptr(Ad), pad(M), K * pad(M), L.ptr(Bd), pad(N), K * pad(N), L.ptr(C), ldc, M * ldc,
                                         M, N, K, G, 0, L.stream()))
    torch.cuda.synchronize()
    Cn = C.cpu().numpy()
    ref = np.einsum("gkm,gkn->gmn", a.astype(np.float64), b.astype(np.float64))
    assert np.abs(Cn[:, :, :N] - ref).max() / np.abs(ref).max() < 2e-5
    assert np.sqrt(((Cn[:, :, :N] - ref) ** 2).mean()) / np.sqrt((ref ** 2).mean()) < 1e-5
    assert (Cn[:, :, N:] == 7.0).all()
    # small integers are exact in bf16: the product must be bit-exact
    ai = rng.integers(-3, 4, (G, K, pad(M))).astype(np.float32); ai[:, :, M:] = 0
    bi = rng.integers(-3, 4, (G, K, pad(N))).astype(np.float32); bi[:, :, N:] = 0
    C2 = torch.zeros((G, M, ldc), device="cuda")
    aid, bid = torch.from_numpy(ai).cuda(), torch.from_numpy(bi).cuda()
    L.check(L.lib().aread_gemm_bf16x3_rc(L.ptr(aid), pad(M), K * pad(M), L.ptr(bid), pad(N), K * pad(N), L.ptr(C2), ldc, M * ldc,
                                         M, N, K, G, 0, L.stream()))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(C2.cpu().numpy()[:, :, :N], np.einsum("gkm,gkn->gmn", ai[:, :, :M], bi[:, :, :N]))
