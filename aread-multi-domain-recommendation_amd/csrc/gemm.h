// gemm.h -- grouped fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32
// products, fp32 accumulate; bit-for-bit a k-ordered fmaf chain).  All dense contractions of the path
// (expert and tower Linear layers, gate logits, their dgrad and wgrad) go through this one template.
//
//   C[g][m][n] (+)= sum_k A[g](m,k) * B[g](n,k)  (+ bias[g][n])
//
// Operand element addressing ("KC" = k contiguous, "MC" = m/n contiguous):
//   A_KC: A[g*a_gs + m*lda + k]      A_MC: A[g*a_gs + k*lda + m]
//   B_KC: B[g*b_gs + n*ldb + k]      B_MC: B[g*b_gs + k*ldb + n]
// forward  Y = X W^T      : A_KC (activations), B_KC (torch Linear weight [out,in])
// dgrad    dX = dY W      : A_KC (dY),          B_MC (W read along its rows)
// wgrad    dW = dY^T X    : A_MC (dY),          B_MC (X), K = batch rows, split-K into slabs
//
// Workgroup = 256 threads = 4 waves stacked along M (16 rows each) -> 64 x (16*NI) output tile,
// BK = 32.  LDS tiles are [rows][BK+2] so that the fragment reads (lane -> row l&15, k l>>4) hit 32
// distinct banks.  Global loads are 16 B per lane along the contiguous axis and are prefetched into
// registers one k-tile ahead of the MFMAs.
#pragma once
#include "common.h"

#define GEMM_THREADS 256
#define GEMM_BK 32
#define GEMM_PITCH (GEMM_BK + 2)

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct GemmP {
    const float* A; int64_t lda, a_gs;
    const float* B; int64_t ldb, b_gs;
    float* C; int64_t ldc, c_gs, c_ks;       // c_ks: slab stride for split-K outputs
    const float* bias; int64_t bias_gs;
    int M, N, K, G;
    int accumulate;
    int k_split, k_chunk;                    // K is cut into k_split slices of k_chunk (multiple of 64)
    // gating.  m_gate: M is the batch axis (tile_seg indexes 64-row m-tiles); k_gate: K is the batch axis
    const int32_t* tile_seg;                 // nullable; -1 = unused tile
    const int32_t* tile_valid;               // valid rows per tile (for the statistics epilogue)
    const uint8_t* active; int active_ld;    // nullable; active[seg*active_ld + g]
    int gate_axis;                           // 0 none, 1 = M, 2 = K
    // BatchNorm statistics epilogue (forward): per m-tile, per column: (mean, M2) over the valid rows
    float* stat_part; int64_t stat_ld;       // stat_part[(tile_m*stat_ld + g*N + n)*2 + {0,1}]
    int dbg;                                 // (unused)
    // inference epilogue (forward with running statistics and no backward to follow): BatchNorm + ReLU applied to the
    // accumulators, C receives the ACTIVATION (the pre-BatchNorm H is not kept), padding rows are written as zeros.
    // BatchNorm is skipped for one-row segments (layer.py:226).  Indexed like bias: [g*bias_gs + n].  Needs gate_axis == 1.
    const float* ep_rmean; const float* ep_rvar; const float* ep_gamma; const float* ep_beta;
    const int32_t* ep_seg_count;
};

// LDS image of an operand tile.  KC operands: [rows][BK+2] (k contiguous, as in global memory).
// MC operands: [BK][PITCH_R] (rows contiguous, as in global memory) -- no transposition on the way in, the
// staging stores are whole 16-byte vectors.  Both pitches put the 4 k-slices of a fragment read on
// disjoint bank groups (PITCH_R = 16 mod 32).
template <int ROWS, bool KC>
struct TileLoader {
    // number of float4 each thread moves per k-tile
    static constexpr int F4 = (ROWS * GEMM_BK / 4 + GEMM_THREADS - 1) / GEMM_THREADS;
    static constexpr int PITCH_R = (ROWS % 32 == 0) ? ROWS + 16 : ROWS;
    static constexpr int LDS_FLOATS = KC ? ROWS * GEMM_PITCH : GEMM_BK * PITCH_R;
    float4 v[F4];

    // fragment element (row r, k) of the staged tile
    static __device__ __forceinline__ int frag_offset(int r, int k) { return KC ? r * GEMM_PITCH + k : k * PITCH_R + r; }
    static constexpr int ROW_STEP = KC ? GEMM_PITCH : 1;     // +1 row
    static constexpr int K_STEP = KC ? 1 : PITCH_R;          // +1 k

    // FULL: the tile lies completely inside the operand (no row / k guards): straight-line 16-byte loads that the
    // compiler can issue back to back and wait for once.  Otherwise every element is guarded (edges, K tails).
    template <bool FULL>
    __device__ __forceinline__ void load(const float* __restrict__ base, int64_t ld, int r0, int r_end, int k0,
                                         int k_end) {
        const int tid = threadIdx.x;
#pragma unroll
        for (int p = 0; p < F4; ++p) {
            const int idx = tid + p * GEMM_THREADS;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (KC) {
                const int row = idx >> 3, kq = idx & 7;                 // 8 float4 per row of BK=32
                if (ROWS * 8 % GEMM_THREADS == 0 || row < ROWS) {
                    const int r = r0 + row, k = k0 + 4 * kq;
                    const float* ptr = base + (int64_t)r * ld + k;
                    if (FULL) t = *(const float4*)ptr;
                    else if (r < r_end) {
                        if (k + 3 < k_end) t = *(const float4*)ptr;
                        else {
                            if (k < k_end) t.x = ptr[0];
                            if (k + 1 < k_end) t.y = ptr[1];
                            if (k + 2 < k_end) t.z = ptr[2];
                        }
                    }
                }
            } else {
                constexpr int Q = ROWS / 4;                              // float4 per k-row
                const int krow = idx / Q, mq = idx - krow * Q;
                if ((Q * GEMM_BK) % GEMM_THREADS == 0 || krow < GEMM_BK) {
                    const int k = k0 + krow, r = r0 + 4 * mq;
                    const float* ptr = base + (int64_t)k * ld + r;
                    if (FULL) t = *(const float4*)ptr;
                    else if (k < k_end) {
                        if (r + 3 < r_end) t = *(const float4*)ptr;
                        else {
                            if (r < r_end) t.x = ptr[0];
                            if (r + 1 < r_end) t.y = ptr[1];
                            if (r + 2 < r_end) t.z = ptr[2];
                        }
                    }
                }
            }
            v[p] = t;
        }
    }

    __device__ __forceinline__ void store(float* __restrict__ lds) const {
        const int tid = threadIdx.x;
#pragma unroll
        for (int p = 0; p < F4; ++p) {
            const int idx = tid + p * GEMM_THREADS;
            if (KC) {
                const int row = idx >> 3, kq = idx & 7;
                if (row < ROWS) {
                    float* d = lds + row * GEMM_PITCH + 4 * kq;
                    *(float2*)d = make_float2(v[p].x, v[p].y);
                    *(float2*)(d + 2) = make_float2(v[p].z, v[p].w);
                }
            } else {
                constexpr int Q = ROWS / 4;
                const int krow = idx / Q, mq = idx - krow * Q;
                if (krow < GEMM_BK) *(float4*)(lds + krow * PITCH_R + 4 * mq) = v[p];
            }
        }
    }
};

// ---- epilogue shared by the fp32 and the split-bf16 kernels ----------------------------------------------
// C/D layout of every 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg.
// s_c: per-wave staging area of 16 x (16*NI + 4) floats (the k-loop's LDS tiles are dead by now and are reused).
template <int NI>
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, f32x4 (&acc)[NI], int g, int ks, int m0, int n0, int by,
                                              float (*s_red)[16 * NI], float* s_c) {
    constexpr int TN = 16 * NI, CP = TN + 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col_in = lane & 15, row_base = wave * 16 + (lane >> 4) * 4;
    float* Cg = p.C + (int64_t)g * p.c_gs + (int64_t)ks * p.c_ks;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int n = n0 + i * 16 + col_in;
        const float bv = (p.bias && n < p.N) ? p.bias[(int64_t)g * p.bias_gs + n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] += bv;
    }
    if (p.ep_gamma) {
        const int seg = p.tile_seg[by];
        const bool bn = p.ep_seg_count[seg] > 1;
        const int nv = p.tile_valid[by];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int n = n0 + i * 16 + col_in;
            if (n < p.N) {
                const int64_t o = (int64_t)g * p.bias_gs + n;
                const float mu = p.ep_rmean[o], rs = 1.0f / sqrtf(p.ep_rvar[o] + 1e-5f), ga = p.ep_gamma[o], be = p.ep_beta[o];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[i][r];
                    if (bn) v = (v - mu) * rs * ga + be;
                    v = v > 0.f ? v : 0.f;
                    acc[i][r] = (row_base + r < nv) ? v : 0.f;
                }
            }
        }
    }
    const bool vec_ok = (n0 + TN <= p.N) && (m0 + 64 <= p.M) && ((p.ldc & 3) == 0) && ((p.c_gs & 3) == 0) &&
                        ((p.c_ks & 3) == 0) && (((uintptr_t)p.C & 15) == 0);
    if (vec_ok) {
        // C/D fragment (col = lane&15, rows (lane>>4)*4 + r) -> LDS [16][CP] -> 16-byte row segments to global
        float* sw = s_c + wave * 16 * CP;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) sw[((lane >> 4) * 4 + r) * CP + i * 16 + col_in] = acc[i][r];
        __builtin_amdgcn_wave_barrier();
        constexpr int Q = TN / 4;                       // float4 per row
#pragma unroll
        for (int it = 0; it < (16 * Q + 63) / 64; ++it) {
            const int idx = it * 64 + lane;
            if (idx < 16 * Q) {
                const int rr = idx / Q, c4 = idx - rr * Q;
                float4 v = *(const float4*)(sw + rr * CP + c4 * 4);
                float4* dst = (float4*)(Cg + (int64_t)(m0 + wave * 16 + rr) * p.ldc + n0 + c4 * 4);
                if (p.accumulate) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                *dst = v;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int n = n0 + i * 16 + col_in;
            if (n < p.N) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + row_base + r;
                    if (m < p.M) {
                        float* c = Cg + (int64_t)m * p.ldc + n;
                        *c = p.accumulate ? *c + acc[i][r] : acc[i][r];
                    }
                }
            }
        }
    }
    if (p.stat_part) {
        const int nvalid = p.tile_valid[by];
        // pass 1: column sums over valid rows
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) s += (row_base + r < nvalid) ? acc[i][r] : 0.f;
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (lane < 16) s_red[wave][i * 16 + lane] = s;
        }
        __syncthreads();
        float mean[NI];
        const float inv = 1.0f / (float)nvalid;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = i * 16 + col_in;
            mean[i] = ((s_red[0][c] + s_red[1][c]) + (s_red[2][c] + s_red[3][c])) * inv;
        }
        __syncthreads();
        // pass 2: centred sum of squares
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = acc[i][r] - mean[i];
                s += (row_base + r < nvalid) ? d * d : 0.f;
            }
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (lane < 16) s_red[wave][i * 16 + lane] = s;
        }
        __syncthreads();
        if (wave == 0 && lane < 16) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int c = i * 16 + lane, n = n0 + c;
                if (n < p.N) {
                    const float m2 = (s_red[0][c] + s_red[1][c]) + (s_red[2][c] + s_red[3][c]);
                    float* o = p.stat_part + ((int64_t)by * p.stat_ld + (int64_t)g * p.N + n) * 2;
                    o[0] = mean[i];
                    o[1] = m2;
                }
            }
        }
    }
}

template <int NI, bool A_KC, bool B_KC>
__global__ __launch_bounds__(GEMM_THREADS) void k_gemm(const GemmP p) {
    constexpr int TM = 64, TN = 16 * NI;
    using LA = TileLoader<TM, A_KC>;
    using LB = TileLoader<TN, B_KC>;
    constexpr int STAGE = 4 * 16 * (TN + 4);            // epilogue staging (floats), overlays the operand tiles
    constexpr int OPER = LA::LDS_FLOATS + LB::LDS_FLOATS;
    __shared__ __attribute__((aligned(16))) float s_lds[OPER > STAGE ? OPER : STAGE];
    float* As = s_lds;
    float* Bs = s_lds + LA::LDS_FLOATS;
    __shared__ float s_red[4][TN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.z / p.k_split, ks = blockIdx.z - g * p.k_split;
    // XCD-aware tile mapping: workgroups are dealt round-robin over the 8 XCDs (each with a private L2), so give
    // all column tiles of one row tile to the same XCD -- the A row tile is then fetched into one L2, not eight.
    int bx = blockIdx.x, by = blockIdx.y;
    {
        const int nx = gridDim.x, ny8 = (gridDim.y / 8) * 8;
        const int id = blockIdx.x + nx * blockIdx.y;
        if (id < nx * ny8) {
            const int xcd = id & 7, slot = id >> 3;
            bx = slot % nx;
            by = (slot / nx) * 8 + xcd;
        }
    }
    const int m0 = by * TM, n0 = bx * TN;

    int seg = 0;
    if (p.gate_axis == 1) {
        seg = p.tile_seg[by];
        if (seg < 0) return;
        if (p.active && !p.active[seg * p.active_ld + g]) return;
    }
    const float* Ag = p.A + (int64_t)g * p.a_gs;
    const float* Bg = p.B + (int64_t)g * p.b_gs;
    const int k_begin = ks * p.k_chunk;
    int k_end = k_begin + p.k_chunk;
    if (k_end > p.K) k_end = p.K;

    f32x4 acc[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto tile_live = [&](int k0) -> bool {
        if (p.gate_axis != 2) return true;
        const int s = p.tile_seg[k0 / TILE_M];
        if (s < 0) return false;
        return !(p.active && !p.active[s * p.active_ld + g]);
    };
    auto next_live = [&](int k0) -> int {
        while (k0 < k_end && !tile_live(k0)) k0 += GEMM_BK;
        return k0;
    };

    LA la;
    LB lb;
    const bool full_mn = (m0 + TM <= p.M) && (n0 + TN <= p.N);
    auto load_tiles = [&](int kk0) {
        if (full_mn && kk0 + GEMM_BK <= k_end) {               // wave-uniform
            la.template load<true>(Ag, p.lda, m0, p.M, kk0, k_end);
            lb.template load<true>(Bg, p.ldb, n0, p.N, kk0, k_end);
        } else {
            la.template load<false>(Ag, p.lda, m0, p.M, kk0, k_end);
            lb.template load<false>(Bg, p.ldb, n0, p.N, kk0, k_end);
        }
    };
    int k0 = next_live(k_begin);
    if (k0 < k_end) load_tiles(k0);
    const int fr = lane & 15, fk = lane >> 4;
    while (k0 < k_end) {
        la.store(As);
        lb.store(Bs);
        __syncthreads();
        const int kn = next_live(k0 + GEMM_BK);
        if (kn < k_end) load_tiles(kn);
        const float* ap = As + LA::frag_offset(wave * 16 + fr, fk);
        const float* bp = Bs + LB::frag_offset(fr, fk);
#pragma unroll
        for (int kk = 0; kk < GEMM_BK / 4; ++kk) {
            const float a = ap[kk * 4 * LA::K_STEP];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const float b = bp[i * 16 * LB::ROW_STEP + kk * 4 * LB::K_STEP];
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            }
        }
        __syncthreads();
        k0 = kn;
    }

    gemm_epilogue<NI>(p, acc, g, ks, m0, n0, by, s_red, s_lds);
}

// ================================================================================================
// split-bf16 ("bf16x3") kernel: C (+)= A B^T with both operands k-contiguous fp32 in memory.
// Every fp32 value x is split while it is staged into LDS: hi = bf16(x), lo = bf16(x - hi); the product is
// formed as hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16 with fp32 accumulation (the lo*lo term,
// 2^-16 relative, is dropped).  Relative error of a dot product ~1e-6: inside the path's 1e-4 logit tolerance,
// at 3/16 of the matrix-pipe time of the fp32 MFMA.  Same tile shape, gating and epilogue as k_gemm.
// LDS images: [rows][32 bf16 + 8 pad] for hi and lo (80-byte rows: the 16-byte fragment reads of 16 rows
// start on 16 different 4-bank groups).
// ================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// (hi, lo) of two fp32 values: hi = bf16(x), lo = bf16(x - hi).  Written on pairs so that it compiles to v_cvt_pk_bf16_f32, the pair
// widened back by shift / mask, ONE packed subtract, v_cvt_pk_bf16_f32: five VALU instructions per pair (the element-wise loop came out
// at seven for every other pair; -4 % on every split-bf16 GEMM of the step)
__device__ __forceinline__ void bf3_split2(float x0, float x1, bf16x2& h, bf16x2& l) {
    const f32x2 x = {x0, x1};
    h = __builtin_convertvector(x, bf16x2);
    l = __builtin_convertvector(x - __builtin_convertvector(h, f32x2), bf16x2);
}
__device__ __forceinline__ void bf3_split4(const float4& v, bf16x4& h, bf16x4& l) {
    bf16x2 h0, l0, h1, l1;
    bf3_split2(v.x, v.y, h0, l0);
    bf3_split2(v.z, v.w, h1, l1);
    h = (bf16x4){h0[0], h0[1], h1[0], h1[1]};
    l = (bf16x4){l0[0], l0[1], l1[0], l1[1]};
}
// LDS image of a split-bf16 operand tile [ROWS][32 k]: four k-planes of [ROWS][8 bf16 = one 16-B slot]; plane f keeps
// row r in slot r ^ (2*f).  With that XOR both the ds_read_b128 fragment reads (16-lane groups {0-3,12-15,20-27}, ...:
// every group sees all 16 fragment rows once, on two neighbouring planes) and the ds_write_b64 staging stores
// (16 contiguous lanes = 2 rows x 4 planes x 2 halves) touch every bank once: conflict-free, no padding.
__device__ __forceinline__ int bf3_off(int rows, int row, int plane) { return (plane * rows + (row ^ (2 * plane))) * 8; }

template <int ROWS>
struct Bf3Loader {
    static constexpr int F4 = (ROWS * GEMM_BK / 4 + GEMM_THREADS - 1) / GEMM_THREADS;
    float4 v[F4];

    template <bool FULL>
    __device__ __forceinline__ void load(const float* __restrict__ base, int64_t ld, int r0, int r_end, int k0, int k_end) {
        const int tid = threadIdx.x;
#pragma unroll
        for (int p = 0; p < F4; ++p) {
            const int idx = tid + p * GEMM_THREADS;
            const int row = idx >> 3, kq = idx & 7;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ROWS * 8 % GEMM_THREADS == 0 || row < ROWS) {
                const int r = r0 + row, k = k0 + 4 * kq;
                const float* ptr = base + (int64_t)r * ld + k;
                if (FULL) t = *(const float4*)ptr;
                else if (r < r_end) {
                    if (k + 3 < k_end) t = *(const float4*)ptr;
                    else {
                        if (k < k_end) t.x = ptr[0];
                        if (k + 1 < k_end) t.y = ptr[1];
                        if (k + 2 < k_end) t.z = ptr[2];
                    }
                }
            }
            v[p] = t;
        }
    }

    __device__ __forceinline__ void store(__bf16* __restrict__ hi, __bf16* __restrict__ lo) const {
        const int tid = threadIdx.x;
#pragma unroll
        for (int p = 0; p < F4; ++p) {
            const int idx = tid + p * GEMM_THREADS;
            const int row = idx >> 3, kq = idx & 7;
            if (ROWS * 8 % GEMM_THREADS == 0 || row < ROWS) {
                bf16x4 h, l;
                bf3_split4(v[p], h, l);
                const int o = bf3_off(ROWS, row, kq >> 1) + 4 * (kq & 1);
                *(bf16x4*)(hi + o) = h;
                *(bf16x4*)(lo + o) = l;
            }
        }
    }
};

template <int NI>
__global__ __launch_bounds__(GEMM_THREADS) void k_gemm_bf3(const GemmP p) {
    constexpr int TM = 64, TN = 16 * NI;
    constexpr int STAGE_B = 4 * 16 * (TN + 4) * 4;       // epilogue staging (bytes), overlays the operand tiles
    constexpr int OPER_B = 2 * (TM + TN) * GEMM_BK * 2;
    __shared__ __attribute__((aligned(16))) char s_lds[OPER_B > STAGE_B ? OPER_B : STAGE_B];
    __bf16* Ah = (__bf16*)s_lds;
    __bf16* Al = Ah + TM * GEMM_BK;
    __bf16* Bh = Al + TM * GEMM_BK;
    __bf16* Bl = Bh + TN * GEMM_BK;
    __shared__ float s_red[4][TN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.z;
    int bx = blockIdx.x, by = blockIdx.y;
    {
        const int nx = gridDim.x, ny8 = (gridDim.y / 8) * 8;
        const int id = blockIdx.x + nx * blockIdx.y;
        if (id < nx * ny8) {
            const int xcd = id & 7, slot = id >> 3;
            bx = slot % nx;
            by = (slot / nx) * 8 + xcd;
        }
    }
    const int m0 = by * TM, n0 = bx * TN;
    if (p.gate_axis == 1) {
        const int seg = p.tile_seg[by];
        if (seg < 0) return;
        if (p.active && !p.active[seg * p.active_ld + g]) return;
    }
    const float* Ag = p.A + (int64_t)g * p.a_gs;
    const float* Bg = p.B + (int64_t)g * p.b_gs;
    const int k_end = p.K;

    f32x4 acc[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    Bf3Loader<TM> la;
    Bf3Loader<TN> lb;
    const bool full_mn = (m0 + TM <= p.M) && (n0 + TN <= p.N);
    auto load_tiles = [&](int kk0) {
        if (full_mn && kk0 + GEMM_BK <= k_end) {
            la.template load<true>(Ag, p.lda, m0, p.M, kk0, k_end);
            lb.template load<true>(Bg, p.ldb, n0, p.N, kk0, k_end);
        } else {
            la.template load<false>(Ag, p.lda, m0, p.M, kk0, k_end);
            lb.template load<false>(Bg, p.ldb, n0, p.N, kk0, k_end);
        }
    };
    load_tiles(0);
    const int fr = lane & 15, fk = lane >> 4;              // fragment: row fr, k = 8*fk .. 8*fk+7
    const int a_off = bf3_off(TM, wave * 16 + fr, fk);
    const int b_off = bf3_off(TN, fr, fk);                 // + i*16 rows: the XOR only touches the low 3 row bits
    for (int k0 = 0; k0 < k_end; k0 += GEMM_BK) {
        la.store(Ah, Al);
        lb.store(Bh, Bl);
        __syncthreads();
        if (k0 + GEMM_BK < k_end) load_tiles(k0 + GEMM_BK);
        const bf16x8 ah = *(const bf16x8*)(Ah + a_off), al = *(const bf16x8*)(Al + a_off);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const bf16x8 bh = *(const bf16x8*)(Bh + b_off + i * 16 * 8);
            const bf16x8 bl = *(const bf16x8*)(Bl + b_off + i * 16 * 8);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[i], 0, 0, 0);
        }
        __syncthreads();
    }
    gemm_epilogue<NI>(p, acc, g, 0, m0, n0, by, s_red, (float*)s_lds);
}

// ================================================================================================
// split-bf16 kernel for ROW-contiguous operands (the weight gradients dW = dH^T X: K = the batch rows, element (r,k) at
// base + k*ld + r for both operands).  The bf16 MFMA wants 8 consecutive k per lane, i.e. a transposed view of the
// k-major tiles as they arrive from memory.  The tiles are staged as they are -- k-major [32 k][rows] bf16 images (hi and
// lo), 8-byte stores of 4 consecutive rows -- and the fragments are read with ds_read_b64_tr_b16, which hands lane i of a
// 16-lane group column i of a 4-row block: exactly (row i, 4 consecutive k).  Two such reads make one 8-k fragment.
// LDS image (bank-conflict-free for the transposed reads, whose two 16-lane groups per half-wave touch k-rows 8g+4h+q
// and 8(g+1)+4h+q, q = 0..3, at one 32-byte chunk of the row):
//   rows <= 64 : k-rows r and r+8 share a 256-byte line (r+8 in the upper half), the 32-byte chunk index is XORed with r&3
//   rows <= 128: one 256-byte line per k-row, chunk index XORed with (r&3) | ((r>>3)&1)<<2
// so the eight 32-byte segments of a half-wave read land on eight different bank windows.
// Same split-K geometry, k-tile gating (dead tiles hold stale data) and epilogue as k_gemm<NI,false,false>.
// ================================================================================================
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <int ROWS>
struct RcImage {
    static constexpr bool WIDE = ROWS > 64;
    static constexpr int ELEMS = (WIDE ? 32 : 16) * 128;                  // bf16 elements of one image (hi or lo)
    static __device__ __forceinline__ int off(int r, int m) {            // element offset of (k-row r, row m of the operand)
        const int c = m >> 4;
        if (WIDE) return r * 128 + ((c ^ ((r & 3) | (((r >> 3) & 1) << 2))) << 4) + (m & 15);
        return ((r & 7) | ((r >> 4) << 3)) * 128 + (((r >> 3) & 1) << 6) + ((c ^ (r & 3)) << 4) + (m & 15);
    }
};

template <int ROWS>
struct Bf3RcLoader {
    static constexpr int Q = ROWS / 4;                                    // float4 per k-row
    static constexpr int F4 = (Q * GEMM_BK + GEMM_THREADS - 1) / GEMM_THREADS;
    float4 v[F4];

    template <bool FULL>
    __device__ __forceinline__ void load(const float* __restrict__ base, int64_t ld, int r0, int r_end, int k0, int k_end) {
        const int tid = threadIdx.x;
#pragma unroll
        for (int p = 0; p < F4; ++p) {
            const int idx = tid + p * GEMM_THREADS;
            const int krow = idx / Q, mq = idx - krow * Q;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((Q * GEMM_BK) % GEMM_THREADS == 0 || krow < GEMM_BK) {
                const int k = k0 + krow, r = r0 + 4 * mq;
                const float* ptr = base + (int64_t)k * ld + r;
                if (FULL) t = *(const float4*)ptr;
                else if (k < k_end) {
                    if (r + 3 < r_end) t = *(const float4*)ptr;
                    else {
                        if (r < r_end) t.x = ptr[0];
                        if (r + 1 < r_end) t.y = ptr[1];
                        if (r + 2 < r_end) t.z = ptr[2];
                    }
                }
            }
            v[p] = t;
        }
    }

    __device__ __forceinline__ void store(__bf16* __restrict__ hi, __bf16* __restrict__ lo) const {
        const int tid = threadIdx.x;
#pragma unroll
        for (int p = 0; p < F4; ++p) {
            const int idx = tid + p * GEMM_THREADS;
            const int krow = idx / Q, mq = idx - krow * Q;
            if ((Q * GEMM_BK) % GEMM_THREADS == 0 || krow < GEMM_BK) {
                bf16x4 h, l;
                bf3_split4(v[p], h, l);
                const int o = RcImage<ROWS>::off(krow, 4 * mq);
                *(bf16x4*)(hi + o) = h;
                *(bf16x4*)(lo + o) = l;
            }
        }
    }
};

// 8-k fragment of 16 rows starting at row m0 (a multiple of 16): two transposed reads (k = 8g .. 8g+3 and 8g+4 .. 8g+7)
template <int ROWS>
__device__ __forceinline__ bf16x8 rc_fragment(const __bf16* img, int m0, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    typedef __attribute__((address_space(3))) s16x4* lds_p;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + RcImage<ROWS>::off(8 * g + q, m0 + 4 * pp)));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + RcImage<ROWS>::off(8 * g + 4 + q, m0 + 4 * pp)));
    union { s16x4 s[2]; bf16x8 v; } u;
    u.s[0] = a; u.s[1] = b;
    return u.v;
}

// one 64 x (16*NI) output tile of one (group, k-slice): shared by k_gemm_bf3_rc (one GEMM per launch) and k_gemm_bf3_rc_multi
// (many small weight-gradient GEMMs in one launch).  s_lds: RC_LDS_BYTES(NI) bytes, 256-byte aligned; s_redf: 4 x 16*NI floats.
#define RC_LDS_BYTES(NI) ((2 * (RcImage<64>::ELEMS + RcImage<16 * (NI)>::ELEMS) * 2) > (4 * 16 * (16 * (NI) + 4) * 4) ? \
                          (2 * (RcImage<64>::ELEMS + RcImage<16 * (NI)>::ELEMS) * 2) : (4 * 16 * (16 * (NI) + 4) * 4))
template <int NI>
__device__ __forceinline__ void gemm_bf3_rc_tile(const GemmP& p, int bx, int by, int bz, char* s_lds, float* s_redf) {
    constexpr int TM = 64, TN = 16 * NI;
    __bf16* Ah = (__bf16*)s_lds;
    __bf16* Al = Ah + RcImage<TM>::ELEMS;
    __bf16* Bh = Al + RcImage<TM>::ELEMS;
    __bf16* Bl = Bh + RcImage<TN>::ELEMS;
    float (*s_red)[TN] = (float (*)[TN])s_redf;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = bz / p.k_split, ks = bz - g * p.k_split;
    const int m0 = by * TM, n0 = bx * TN;
    const float* Ag = p.A + (int64_t)g * p.a_gs;
    const float* Bg = p.B + (int64_t)g * p.b_gs;
    const int k_begin = ks * p.k_chunk;
    int k_end = k_begin + p.k_chunk;
    if (k_end > p.K) k_end = p.K;

    f32x4 acc[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto tile_live = [&](int k0) -> bool {
        if (p.gate_axis != 2) return true;
        const int s = p.tile_seg[k0 / TILE_M];
        if (s < 0) return false;
        return !(p.active && !p.active[s * p.active_ld + g]);
    };
    auto next_live = [&](int k0) -> int {
        while (k0 < k_end && !tile_live(k0)) k0 += GEMM_BK;
        return k0;
    };
    Bf3RcLoader<TM> la;
    Bf3RcLoader<TN> lb;
    const bool full_mn = (m0 + TM <= p.M) && (n0 + TN <= p.N);
    auto load_tiles = [&](int kk0) {
        if (full_mn && kk0 + GEMM_BK <= k_end) {
            la.template load<true>(Ag, p.lda, m0, p.M, kk0, k_end);
            lb.template load<true>(Bg, p.ldb, n0, p.N, kk0, k_end);
        } else {
            la.template load<false>(Ag, p.lda, m0, p.M, kk0, k_end);
            lb.template load<false>(Bg, p.ldb, n0, p.N, kk0, k_end);
        }
    };
    int k0 = next_live(k_begin);
    if (k0 < k_end) load_tiles(k0);
    while (k0 < k_end) {
        la.store(Ah, Al);
        lb.store(Bh, Bl);
        __syncthreads();
        const int kn = next_live(k0 + GEMM_BK);
        if (kn < k_end) load_tiles(kn);
        const bf16x8 ah = rc_fragment<TM>(Ah, wave * 16, lane), al = rc_fragment<TM>(Al, wave * 16, lane);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const bf16x8 bh = rc_fragment<TN>(Bh, i * 16, lane), bl = rc_fragment<TN>(Bl, i * 16, lane);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[i], 0, 0, 0);
        }
        __syncthreads();
        k0 = kn;
    }
    gemm_epilogue<NI>(p, acc, g, ks, m0, n0, by, s_red, (float*)s_lds);
}

// XCD-aware mapping: all (m, n) tiles of one (group, k-slice) read the same k-rows of both operands, so they are given
// to ONE XCD (workgroups are dealt round-robin over the 8 XCDs in dispatch order): the slice is then fetched into one L2
// instead of eight (PMC: 228 MB of HBM traffic per expert-L1 launch without it, 3.3x the compulsory traffic).
// id: block index inside this GEMM's (nx, ny, nz) grid, dealt in dispatch order (id & 7 == the dispatching XCD's slot).
__device__ __forceinline__ void rc_block_map(int id, int nx, int ny, int nz, int* bx, int* by, int* bz) {
    const int bps = nx * ny;
    if ((nz & 7) == 0) {
        const int xcd = id & 7, slot = id >> 3;
        *bz = (slot / bps) * 8 + xcd;
        const int mn = slot - (slot / bps) * bps;
        *by = mn / nx;
        *bx = mn - *by * nx;
    } else {
        *bz = id / bps;
        const int mn = id - *bz * bps;
        *by = mn / nx;
        *bx = mn - *by * nx;
    }
}

template <int NI>
__global__ __launch_bounds__(GEMM_THREADS) void k_gemm_bf3_rc(const GemmP p) {
    __shared__ __attribute__((aligned(256))) char s_lds[RC_LDS_BYTES(NI)];
    __shared__ float s_red[4 * 16 * NI];
    int bx, by, bz;
    rc_block_map(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x, gridDim.y, gridDim.z, &bx, &by, &bz);
    gemm_bf3_rc_tile<NI>(p, bx, by, bz, s_lds, s_red);
}

// ---- many small weight-gradient GEMMs in ONE launch ------------------------------------------------------------------------
// The backward queues ~13 of them (heads, six tower layers, two gate layers, the deeper expert layers): as separate launches
// they are a serial chain of 8-22 us kernels with a dozen workgroups each on the side stream (~190 us, the step's tail waited
// for it); in one launch they run side by side.  Each GEMM keeps its own tile width (NI) and split-K geometry; its blocks
// start at a multiple of 8 so that the XCD mapping above holds per GEMM.
#define RC_MULTI_MAX 20
struct RcDesc {
    const float* A; int64_t lda, a_gs;
    const float* B; int64_t ldb, b_gs;
    float* C; int64_t c_gs, c_ks;
    const uint8_t* active;
    int M, N, K, G, k_split, k_chunk;
    int ni, nx, ny, nz;
};
struct RcMultiP {
    int n;
    int first[RC_MULTI_MAX + 1];                      // first[i]: first block of GEMM i (multiple of 8); first[n]: grid size
    RcDesc d[RC_MULTI_MAX];
    const int32_t* tile_seg; int active_ld, gate_axis;
};
#ifdef GEMM_RC_MULTI_IMPL          // defined by gemm.hip only: a non-template kernel must live in one translation unit
__global__ __launch_bounds__(GEMM_THREADS) void k_gemm_bf3_rc_multi(const RcMultiP a) {
    __shared__ __attribute__((aligned(256))) char s_lds[RC_LDS_BYTES(8)];
    __shared__ float s_red[4 * 16 * 8];
    int i = 0;
    while (i + 1 < a.n && (int)blockIdx.x >= a.first[i + 1]) ++i;        // (block-uniform)
    const RcDesc& d = a.d[i];
    const int id = blockIdx.x - a.first[i];
    if (id >= d.nx * d.ny * d.nz) return;                                // padding up to the next multiple of 8
    GemmP p = {};
    p.A = d.A; p.lda = d.lda; p.a_gs = d.a_gs;
    p.B = d.B; p.ldb = d.ldb; p.b_gs = d.b_gs;
    p.C = d.C; p.ldc = d.N; p.c_gs = d.c_gs; p.c_ks = d.c_ks;
    p.M = d.M; p.N = d.N; p.K = d.K; p.G = d.G; p.k_split = d.k_split; p.k_chunk = d.k_chunk;
    p.tile_seg = a.tile_seg; p.active = d.active; p.active_ld = a.active_ld; p.gate_axis = a.gate_axis;
    int bx, by, bz;
    rc_block_map(id, d.nx, d.ny, d.nz, &bx, &by, &bz);
    switch (d.ni) {
        case 8: gemm_bf3_rc_tile<8>(p, bx, by, bz, s_lds, s_red); break;
        case 6: gemm_bf3_rc_tile<6>(p, bx, by, bz, s_lds, s_red); break;
        case 4: gemm_bf3_rc_tile<4>(p, bx, by, bz, s_lds, s_red); break;
        case 2: gemm_bf3_rc_tile<2>(p, bx, by, bz, s_lds, s_red); break;
        default: gemm_bf3_rc_tile<1>(p, bx, by, bz, s_lds, s_red); break;
    }
}
#endif

// number of 16-column MFMA tiles per wave: 96-wide tiles when they cover N with less padding than 128-wide ones
static inline int gemm_ni(int N) {
    if (N > 64) {
        const int w128 = (N + 127) / 128 * 128, w96 = (N + 95) / 96 * 96;
        return w96 < w128 ? 6 : 8;
    }
    return N > 32 ? 4 : (N > 16 ? 2 : 1);
}
int launch_gemm(const GemmP& p, bool a_kc, bool b_kc, hipStream_t st);
// split-bf16 variant: both operands k-contiguous, no split-K / k-gating
int launch_gemm_bf3(const GemmP& p, hipStream_t st);
// split-bf16 variant for two row-contiguous operands (weight gradients); split-K and k-tile gating as launch_gemm
int launch_gemm_bf3_rc(const GemmP& p, hipStream_t st);
// n such GEMMs (split-K slabs with ldc = N, the same tile_seg / gate_axis) in one launch
int launch_gemm_bf3_rc_multi(const GemmP* p, int n, hipStream_t st);
