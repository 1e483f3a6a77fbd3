// Timing-only experiment (round 3): what does the expert-L1 forward GEMM cost when BOTH operands arrive as pre-split bf16 (hi, lo)
// images by LDS-DMA (no fp32 -> (hi, lo) conversion, no ds_write in the k-loop), against converting the A tile in flight?
// Results are not checked (operands are whatever the buffers hold); build: hipcc -O3 --offload-arch=gfx950 -o gemm_presplit_exp gemm_presplit_exp.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ int bf3_off(int rows, int row, int plane) { return (plane * rows + (row ^ (2 * plane))) * 8; }

// WG tile (32*WM) x (32*FN*WN), waves WM x WN, wave tile 32 x 32*FN; NB LDS buffers
template <int WM, int WN, int FN, int PRESPLIT, int NB>
__global__ __launch_bounds__(64 * WM * WN) void k_exp(const float* __restrict__ A, const __bf16* __restrict__ Aimg, const __bf16* __restrict__ Wimg,
                                                      float* __restrict__ C, int M, int N, int K, int nx, int ny) {
    constexpr int NT = 64 * WM * WN, TM = 32 * WM, TN = 32 * FN * WN;
    constexpr int A_EL = TM * 32, W_EL = TN * 32;
    constexpr int UA = (TM * 4 / NT) > 0 ? (TM * 4 / NT) : 1, UAD = (TM * 8 + NT - 1) / NT, UW = (TN * 8 + NT - 1) / NT;
    constexpr int UWR = (TN * 4 / NT) > 0 ? (TN * 4 / NT) : 1;      // PRESPLIT == 2: W through registers too (fp32 [N][K], split in the kernel)
    const float* Wf = (const float*)Wimg;
    __shared__ __attribute__((aligned(1024))) char s_lds[NB * 2 * (A_EL + W_EL) * 2];
    __bf16* Ab = (__bf16*)s_lds;
    __bf16* Wb = Ab + NB * 2 * A_EL;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WN, wc = wave - wr * WN, l31 = lane & 31, lh = lane >> 5;
    const int KS = K >> 5;
    for (int t = blockIdx.x; t < nx * ny; t += gridDim.x) {
        const int xcd = t & 7, slot = t >> 3;
        int bx = slot % nx, by = (slot / nx) * 8 + xcd;
        if (by >= ny) { bx = t % nx; by = t / nx; }
        const int m0 = by * TM;
        const __bf16* wimg = Wimg + (int64_t)bx * KS * 2 * W_EL;
        const __bf16* aimg = Aimg + (int64_t)by * KS * 2 * A_EL;
        f32x16 acc[FN];
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        float4 av[UA][2];
        auto loadA = [&](int s) {
#pragma unroll
            for (int u = 0; u < UA; ++u) {
                const int idx = tid + u * NT, row = idx >> 2, plane = idx & 3;
                const int m = m0 + row < M ? m0 + row : M - 1;
                const float* src = A + (int64_t)m * K + s * 32 + plane * 8;
                av[u][0] = *(const float4*)src; av[u][1] = *(const float4*)(src + 4);
            }
        };
        auto storeA = [&](int buf) {
            __bf16* Ah = Ab + buf * 2 * A_EL; __bf16* Al = Ah + A_EL;
#pragma unroll
            for (int u = 0; u < UA; ++u) {
                const int idx = tid + u * NT, row = idx >> 2, plane = idx & 3;
                const float x[8] = {av[u][0].x, av[u][0].y, av[u][0].z, av[u][0].w, av[u][1].x, av[u][1].y, av[u][1].z, av[u][1].w};
                bf16x8 h, l;
#pragma unroll
                for (int e = 0; e < 8; ++e) { h[e] = (__bf16)x[e]; l[e] = (__bf16)(x[e] - (float)h[e]); }
                const int o = bf3_off(TM, row, plane);
                *(bf16x8*)(Ah + o) = h; *(bf16x8*)(Al + o) = l;
            }
        };
        float4 wv[UWR][2];
        auto loadW = [&](int s) {
#pragma unroll
            for (int u = 0; u < UWR; ++u) {
                const int idx = tid + u * NT, row = idx >> 2, plane = idx & 3;
                const float* src = Wf + (int64_t)(bx * TN + row) * K + s * 32 + plane * 8;
                wv[u][0] = *(const float4*)src; wv[u][1] = *(const float4*)(src + 4);
            }
        };
        auto storeW = [&](int buf) {
            __bf16* Wh = Wb + buf * 2 * W_EL; __bf16* Wl = Wh + W_EL;
#pragma unroll
            for (int u = 0; u < UWR; ++u) {
                const int idx = tid + u * NT, row = idx >> 2, plane = idx & 3;
                const float x[8] = {wv[u][0].x, wv[u][0].y, wv[u][0].z, wv[u][0].w, wv[u][1].x, wv[u][1].y, wv[u][1].z, wv[u][1].w};
                bf16x8 h, l;
#pragma unroll
                for (int e = 0; e < 8; ++e) { h[e] = (__bf16)x[e]; l[e] = (__bf16)(x[e] - (float)h[e]); }
                const int o = bf3_off(TN, row, plane);
                *(bf16x8*)(Wh + o) = h; *(bf16x8*)(Wl + o) = l;
            }
        };
        auto dma = [&](const __bf16* img, int s, __bf16* dstb, int el, int pieces, int U) {
            const char* src = (const char*)(img + (int64_t)s * 2 * el);
            char* dst = (char*)dstb;
            for (int q = 0; q < U; ++q) {
                const int piece = q * NT + wave * 64;
                if (piece >= pieces) break;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (int64_t)(piece + lane) * 16),
                                                 (__attribute__((address_space(3))) void*)(dst + piece * 16), 16, 0, 0);
            }
        };
        const int a_row = wr * 32 + l31, w_row = wc * 32 * FN + l31;
        if (PRESPLIT == 2) {
            // k_gemm_bf3's structure at this tile size: both operands global -> registers (one k-step ahead) -> split -> LDS, one buffer
            loadA(0); loadW(0);
            for (int s = 0; s < KS; ++s) {
                storeA(0); storeW(0);
                __syncthreads();
                if (s + 1 < KS) { loadA(s + 1); loadW(s + 1); }
                const __bf16* Ah = Ab; const __bf16* Al = Ah + A_EL;
                const __bf16* Wh = Wb; const __bf16* Wl = Wh + W_EL;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int plane = lh + 2 * kk;
                    const bf16x8 ah = *(const bf16x8*)(Ah + bf3_off(TM, a_row, plane));
                    const bf16x8 al = *(const bf16x8*)(Al + bf3_off(TM, a_row, plane));
#pragma unroll
                    for (int i = 0; i < FN; ++i) {
                        const bf16x8 wh = *(const bf16x8*)(Wh + bf3_off(TN, w_row + 32 * i, plane));
                        const bf16x8 wl = *(const bf16x8*)(Wl + bf3_off(TN, w_row + 32 * i, plane));
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, wh, acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wl, acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wh, acc[i], 0, 0, 0);
                    }
                }
                __syncthreads();
            }
        } else {
        // prologue: NB-1 stages in flight
        for (int s = 0; s < NB - 1 && s < KS; ++s) {
            dma(wimg, s, Wb + s * 2 * W_EL, W_EL, TN * 8, UW);
            if (PRESPLIT) dma(aimg, s, Ab + s * 2 * A_EL, A_EL, TM * 8, UAD);
        }
        if (!PRESPLIT) { loadA(0); storeA(0); if (KS > 1) loadA(1); }
        if (NB == 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NB - 2) * (UW + (PRESPLIT ? UAD : 0))) : "memory");
        __builtin_amdgcn_s_barrier();
        for (int s = 0; s < KS; ++s) {
            const int cur = s % NB, nxt = (s + NB - 1) % NB;
            const int sp = s + NB - 1 < KS ? s + NB - 1 : KS - 1;        // (clamped repeats keep the vmcnt arithmetic constant)
            dma(wimg, sp, Wb + nxt * 2 * W_EL, W_EL, TN * 8, UW);
            if (PRESPLIT) dma(aimg, sp, Ab + nxt * 2 * A_EL, A_EL, TM * 8, UAD);
            const __bf16* Ah = Ab + (PRESPLIT ? cur : (s & 1)) * 2 * A_EL; const __bf16* Al = Ah + A_EL;
            const __bf16* Wh = Wb + cur * 2 * W_EL; const __bf16* Wl = Wh + W_EL;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int plane = lh + 2 * kk;
                const bf16x8 ah = *(const bf16x8*)(Ah + bf3_off(TM, a_row, plane));
                const bf16x8 al = *(const bf16x8*)(Al + bf3_off(TM, a_row, plane));
#pragma unroll
                for (int i = 0; i < FN; ++i) {
                    const bf16x8 wh = *(const bf16x8*)(Wh + bf3_off(TN, w_row + 32 * i, plane));
                    const bf16x8 wl = *(const bf16x8*)(Wl + bf3_off(TN, w_row + 32 * i, plane));
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, wh, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wl, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wh, acc[i], 0, 0, 0);
                }
            }
            if (!PRESPLIT) {
                if (s + 1 < KS) storeA((s + 1) & 1);
                loadA(s + 2 < KS ? s + 2 : KS - 1);
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * UA + (NB - 2) * UW) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NB - 2) * (UW + UAD)) : "memory");
            }
            __builtin_amdgcn_s_barrier();
        }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < FN; ++i) {
            float* c0 = C + (int64_t)(m0 + wr * 32 + 4 * lh) * N + bx * TN + wc * 32 * FN + 32 * i + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * 32 + 4 * lh + 8 * (r >> 2) + (r & 3);
                if (m < M) c0[(int64_t)(8 * (r >> 2) + (r & 3)) * N] = acc[i][r];
            }
        }
        __builtin_amdgcn_s_barrier();
    }
}


// ---- mode "ring": loader waves and MFMA waves of ONE workgroup decoupled by an S-slot LDS ring (timing only) -------------------------
// 512 threads: waves 0-3 load one 64 x 128 x 32 operand pair per k-step (fp32 from global memory TWO k-steps ahead, split to bf16
// (hi, lo), ds_write into slot g % S, then one LDS atomic on FULL[slot]); waves 4-7 wait for FULL, read their fragments, release the slot
// (one LDS atomic on FREE[slot]) and issue the 24 MFMAs (16x16x32) of the k-step.  One workgroup per CU, persistent over its tiles.
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int S, int D>
__global__ __launch_bounds__(512) void k_ring(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ C, int M, int N, int K,
                                              int nx, int ny) {
    constexpr int TM = 64, TN = 128, SLOT = 2 * (TM + TN) * 32;           // bf16 elements per slot: Ah | Al | Wh | Wl
    extern __shared__ __attribute__((aligned(1024))) char dyn[];
    __bf16* ring = (__bf16*)dyn;
    unsigned* full = (unsigned*)(dyn + (size_t)S * SLOT * 2);
    unsigned* fre = full + S;
    unsigned* abortf = fre + S;                                          // a wait that gives up (a bug) ends every wave of the workgroup
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 2 * S + 1) full[tid] = 0u;
    __syncthreads();
    const int KS = K >> 5;
    const int n_tiles = nx * ny;
    const int my_tiles = (n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int G = my_tiles * KS;
    auto tile_of = [&](int g, int& bx, int& by, int& ks) {
        const int t = blockIdx.x + (g / KS) * gridDim.x;
        ks = g - (g / KS) * KS;
        const int xcd = t & 7, slot = t >> 3;
        bx = slot % nx; by = (slot / nx) * 8 + xcd;
        if (by >= ny) { bx = t % nx; by = t / nx; }
    };
    if (wave < 4) {
        // ---------------- loaders: 256 lanes; A tile 512 float4 (2 per lane), W tile 1024 float4 (4 per lane) ------------------------
        const int lt = tid;                                   // 0..255
        float4 st[D][6];                                      // D k-steps of operands in flight per lane
        auto issue = [&](float4 (&v)[6], int g) {
            int bx, by, ks; tile_of(g, bx, by, ks);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int idx = lt + u * 256, row = idx >> 3, kq = idx & 7;
                const int m = by * TM + row < M ? by * TM + row : M - 1;
                v[u] = *(const float4*)(A + (int64_t)m * K + ks * 32 + kq * 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = lt + u * 256, row = idx >> 3, kq = idx & 7;
                v[2 + u] = *(const float4*)(W + (int64_t)(bx * TN + row) * K + ks * 32 + kq * 4);
            }
        };
        auto put = [&](const float4 (&v)[6], int slot) {
            __bf16* Ah = ring + (size_t)slot * SLOT; __bf16* Al = Ah + TM * 32; __bf16* Wh = Al + TM * 32; __bf16* Wl = Wh + TN * 32;
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int idx = lt + (u < 2 ? u : u - 2) * 256, row = idx >> 3, kq = idx & 7;
                const float x[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                bf16x4 h, l;
#pragma unroll
                for (int e = 0; e < 4; ++e) { h[e] = (__bf16)x[e]; l[e] = (__bf16)(x[e] - (float)h[e]); }
                const int o = bf3_off(u < 2 ? TM : TN, row, kq >> 1) + 4 * (kq & 1);
                if (u < 2) { *(bf16x4*)(Ah + o) = h; *(bf16x4*)(Al + o) = l; }
                else { *(bf16x4*)(Wh + o) = h; *(bf16x4*)(Wl + o) = l; }
            }
        };
        auto stage = [&](float4 (&v)[6], int g) -> bool {
            // (the other stage's six loads may still be in flight)
            if (g + D - 1 < G) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * (D - 1)) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int slot = g % S;
            const unsigned need = 4u * (unsigned)(g / S);
            for (unsigned spins = 0; __hip_atomic_load(&fre[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need; ++spins) {
                __builtin_amdgcn_s_sleep(0);
                if (spins > (1u << 18)) __hip_atomic_store(abortf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (__hip_atomic_load(abortf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return false;
            }
            put(v, slot);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(&full[slot], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (g + D < G) issue(v, g + D);
            return true;
        };
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (d < G) issue(st[d], d);
        bool alive = true;
        for (int g = 0; g < G && alive; g += D) {
#pragma unroll
            for (int d = 0; d < D; ++d)
                if (alive && g + d < G) alive = stage(st[d], g + d);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        // ---------------- consumers: wave c owns rows 16c .. 16c+15 of the tile, all 128 columns --------------------------------------
        const int c = wave - 4;
        const int fr = lane & 15, fk = lane >> 4;
        const int a_off = bf3_off(TM, c * 16 + fr, fk), b_off = bf3_off(TN, fr, fk);
        f32x4v acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        for (int g = 0; g < G; ++g) {
            const int slot = g % S;
            const unsigned need = 4u * (unsigned)(g / S + 1);
            bool dead = false;
            for (unsigned spins = 0; __hip_atomic_load(&full[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need; ++spins) {
                __builtin_amdgcn_s_sleep(0);
                if (spins > (1u << 18)) __hip_atomic_store(abortf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (__hip_atomic_load(abortf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) { dead = true; break; }
            }
            if (dead) break;
            const __bf16* Ah = ring + (size_t)slot * SLOT; const __bf16* Al = Ah + TM * 32; const __bf16* Wh = Al + TM * 32; const __bf16* Wl = Wh + TN * 32;
            const bf16x8 ah = *(const bf16x8*)(Ah + a_off), al = *(const bf16x8*)(Al + a_off);
            bf16x8 bh[8], bl[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { bh[i] = *(const bf16x8*)(Wh + b_off + i * 128); bl[i] = *(const bf16x8*)(Wl + b_off + i * 128); }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(&fre[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[i], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[i], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[i], acc[i], 0, 0, 0);
            }
            int bx, by, ks; tile_of(g, bx, by, ks);
            if (ks == KS - 1) {                              // tile done: C rows (lane>>4)*4 + r, column lane&15 of every 16-wide fragment
#pragma unroll
                for (int i = 0; i < 8; ++i) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = by * TM + c * 16 + fk * 4 + r;
                        if (m < M) C[(int64_t)m * N + bx * TN + i * 16 + fr] = acc[i][r];
                    }
                    acc[i] = (f32x4v){0.f, 0.f, 0.f, 0.f};
                }
            }
        }
    }
}

template <int S, int D>
static void run_ring(const char* name, const float* A, const float* W, float* C, int M, int N, int K) {
    const int nx = N / 128, ny = (M + 63) / 64;
    const size_t lds = (size_t)S * 2 * (64 + 128) * 32 * 2 + (2 * S + 1) * 4 + 64;
    hipFuncSetAttribute((const void*)k_ring<S, D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_ring<S, D>), dim3(256), dim3(512), lds, 0, A, W, C, M, N, K, nx, ny);
    float best = 1e9f, sum = 0.f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k_ring<S, D>), dim3(256), dim3(512), lds, 0, A, W, C, M, N, K, nx, ny);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const float us = ms * 1000.f / 20.f;
        if (us < best) best = us;
        sum += us;
    }
    hipError_t err = hipGetLastError();
    printf("%-30s depth %d, %d slots (%zu KB LDS), 256 x 512 thr, tiles %4d  M=%d N=%d K=%d : %7.2f us (min %.2f) %s\n", name, D, S, lds >> 10, nx * ny, M, N, K, sum / 5, best,
           err == hipSuccess ? "" : hipGetErrorString(err));
}

template <int WM, int WN, int FN, int PRESPLIT, int NB>
static void run(const char* name, const float* A, const __bf16* Aimg, const __bf16* Wimg, float* C, int M, int N, int K, int per_cu) {
    constexpr int TM = 32 * WM, TN = 32 * FN * WN;
    const int nx = N / TN, ny = (M + TM - 1) / TM;
    int grid = per_cu * 256;
    if (grid > nx * ny) grid = nx * ny;
    grid = grid / 8 * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_exp<WM, WN, FN, PRESPLIT, NB>), dim3(grid), dim3(64 * WM * WN), 0, 0, A, Aimg, Wimg, C, M, N, K, nx, ny);
    float best = 1e9f, sum = 0.f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k_exp<WM, WN, FN, PRESPLIT, NB>), dim3(grid), dim3(64 * WM * WN), 0, 0, A, Aimg, Wimg, C, M, N, K, nx, ny);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const float us = ms * 1000.f / 20.f;
        if (us < best) best = us;
        sum += us;
    }
    hipError_t err = hipGetLastError();
    printf("%-44s tile %3dx%3d  %d thr  grid %4d (tiles %4d)  M=%d N=%d K=%d : %7.2f us (min %.2f) %s\n", name, TM, TN, 64 * WM * WN, grid, nx * ny, M, N, K,
           sum / 5, best, err == hipSuccess ? "" : hipGetErrorString(err));
}

int main() {
    const int M = 9728;
    float *A, *C; __bf16 *Aimg, *Wimg;
    hipMalloc(&A, (size_t)M * 1024 * 4); hipMalloc(&C, (size_t)M * 1024 * 4);
    hipMalloc(&Aimg, (size_t)(M + 256) * 1024 * 4); hipMalloc(&Wimg, (size_t)1024 * 1024 * 4);
    hipMemset(A, 0x3c, (size_t)M * 1024 * 4); hipMemset(Aimg, 0x3c, (size_t)(M + 256) * 1024 * 4); hipMemset(Wimg, 0x3c, (size_t)1024 * 1024 * 4);
    printf("== expert L1 forward shape (N 1024, K 288)\n");
    run<4, 2, 2, 0, 2>("A converted in flight, 2 buffers", A, Aimg, Wimg, C, M, 1024, 288, 2);
    run<4, 2, 2, 1, 2>("A pre-split by DMA, 2 buffers", A, Aimg, Wimg, C, M, 1024, 288, 2);
    run<4, 2, 2, 1, 3>("A pre-split by DMA, 3 buffers (1 WG/CU)", A, Aimg, Wimg, C, M, 1024, 288, 1);
    run<2, 2, 2, 0, 2>("A converted, 64x128 tiles, 2 buffers", A, Aimg, Wimg, C, M, 1024, 288, 3);
    run<2, 2, 2, 1, 2>("A pre-split, 64x128 tiles, 2 buffers", A, Aimg, Wimg, C, M, 1024, 288, 3);
    run<2, 2, 2, 1, 3>("A pre-split, 64x128 tiles, 3 buffers", A, Aimg, Wimg, C, M, 1024, 288, 2);
    run<2, 2, 2, 1, 4>("A pre-split, 64x128 tiles, 4 buffers", A, Aimg, Wimg, C, M, 1024, 288, 2);
    run<2, 4, 2, 1, 2>("A pre-split, 64x256 tiles, 8 waves, 2 buf", A, Aimg, Wimg, C, M, 1024, 288, 2);
    run<4, 4, 2, 1, 2>("A pre-split, 128x256 tiles, 16 waves, 2 buf", A, Aimg, Wimg, C, M, 1024, 288, 1);
    printf("== both operands through registers with the split in the kernel (k_gemm_bf3's structure), by tile size\n");
    run<2, 2, 2, 2, 1>("reg path,  64x128 tiles, 4 waves", A, Aimg, Wimg, C, M, 1024, 288, 4);
    run<4, 2, 2, 2, 1>("reg path, 128x128 tiles, 8 waves", A, Aimg, Wimg, C, M, 1024, 288, 2);
    run<4, 1, 4, 2, 1>("reg path, 128x128 tiles, 4 waves (32x128 each)", A, Aimg, Wimg, C, M, 1024, 288, 2);
    run<4, 2, 1, 2, 1>("reg path, 128x64 tiles, 8 waves", A, Aimg, Wimg, C, M, 1024, 288, 3);
    run<4, 4, 1, 2, 1>("reg path, 128x128 tiles, 16 waves", A, Aimg, Wimg, C, M, 1024, 288, 1);
    run<2, 2, 2, 2, 1>("reg path,  64x128 tiles, dgrad shape", A, Aimg, Wimg, C, M, 256, 1024, 4);
    run<4, 2, 2, 2, 1>("reg path, 128x128 tiles, dgrad shape", A, Aimg, Wimg, C, M, 256, 1024, 2);
    printf("== loader / MFMA waves decoupled by an LDS ring, 64x128 tiles, one persistent 512-thread workgroup per CU\n");
    run_ring<4, 2>("ring, L1 forward shape", A, (const float*)Wimg, C, M, 1024, 288);
    run_ring<4, 4>("ring, L1 forward shape", A, (const float*)Wimg, C, M, 1024, 288);
    run_ring<6, 6>("ring, L1 forward shape", A, (const float*)Wimg, C, M, 1024, 288);
    run_ring<6, 8>("ring, L1 forward shape", A, (const float*)Wimg, C, M, 1024, 288);
    run_ring<6, 6>("ring, L1 dgrad shape", A, (const float*)Wimg, C, M, 256, 1024);
    printf("== expert L1 dgrad shape (N 288 -> 256 here, K 1024)\n");
    run<4, 2, 2, 0, 2>("A converted in flight, 2 buffers", A, Aimg, Wimg, C, M, 256, 1024, 2);
    run<4, 2, 2, 1, 2>("A pre-split by DMA, 2 buffers", A, Aimg, Wimg, C, M, 256, 1024, 2);
    run<2, 2, 2, 1, 3>("A pre-split, 64x128 tiles, 3 buffers", A, Aimg, Wimg, C, M, 256, 1024, 2);
    run<2, 2, 2, 1, 4>("A pre-split, 64x128 tiles, 4 buffers", A, Aimg, Wimg, C, M, 256, 1024, 2);
    printf("== expert L2 forward shape as one group (N 128, K 256)\n");
    run<2, 2, 2, 0, 2>("A converted, 64x128 tiles", A, Aimg, Wimg, C, M * 4, 128, 256, 3);
    run<2, 2, 2, 1, 3>("A pre-split, 64x128 tiles, 3 buffers", A, Aimg, Wimg, C, M * 4, 128, 256, 2);
    return 0;
}
