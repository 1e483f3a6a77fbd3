// common.h -- shared host/device helpers for libaread_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/aread_hip.h"

#define TILE_M AREAD_TILE_M
#define MAX_SEG AREAD_MAX_SEG
#define WAVE 64

// ---- error plumbing (host) -------------------------------------------------------------------
void aread_set_error(const char* fmt, ...);
#define AR_CHECK_ARG(cond, ...)                          \
    do {                                                 \
        if (!(cond)) {                                   \
            aread_set_error(__VA_ARGS__);                \
            return AREAD_ERR_ARG;                        \
        }                                                \
    } while (0)
#define AR_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            aread_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,  \
                            __LINE__);                                                        \
            return AREAD_ERR_HIP;                                                             \
        }                                                                                     \
    } while (0)
#define AR_LAUNCH_CHECK() AR_HIP(hipGetLastError())

// ---- plan layout (int32 words) ----------------------------------------------------------------
// must match include/aread_hip.h
#define PLAN_B 0
#define PLAN_NSEG 1
#define PLAN_ROWS 2
#define PLAN_NTILES 3
#define PLAN_NBAD 4
#define PLAN_HDR 16
struct PlanView {
    const int32_t* hdr;
    const int32_t* seg_count;   // [MAX_SEG]
    const int32_t* seg_start;   // [MAX_SEG]
    const int32_t* tile_seg;    // [max_tiles]
    const int32_t* tile_valid;  // [max_tiles]
    const int32_t* row_sample;  // [max_rows]
    const int32_t* sample_row;  // [B]
    int64_t max_rows;
    int64_t max_tiles;
};
static inline __host__ __device__ int64_t plan_max_rows(int64_t B, int n_seg) {
    // every segment wastes at most TILE_M-1 rows; empty segments take no rows
    int64_t segs = n_seg < 1 ? 1 : n_seg;
    if (segs > B) segs = B > 0 ? B : 1;
    int64_t r = B + segs * (TILE_M - 1);
    return (r / TILE_M) * TILE_M;
}
static inline __host__ __device__ PlanView plan_view(const int32_t* plan, int64_t B, int n_seg) {
    PlanView v;
    v.max_rows = plan_max_rows(B, n_seg);
    v.max_tiles = v.max_rows / TILE_M;
    v.hdr = plan;
    v.seg_count = plan + PLAN_HDR;
    v.seg_start = v.seg_count + MAX_SEG;
    v.tile_seg = v.seg_start + MAX_SEG;
    v.tile_valid = v.tile_seg + v.max_tiles;
    v.row_sample = v.tile_valid + v.max_tiles;
    v.sample_row = v.row_sample + v.max_rows;
    return v;
}

// ---- dropout hash: bit-identical to oracle/aread_oracle.py::dropout_keep ------------------------
__host__ __device__ static inline uint32_t mix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x7FEB352Du;
    h ^= h >> 15;
    h *= 0x846CA68Bu;
    h ^= h >> 16;
    return h;
}
__host__ __device__ static inline uint32_t drop_row_key(uint32_t seed, uint32_t sample) {
    return mix32((sample * 0x9E3779B1u) ^ seed);
}
__host__ __device__ static inline bool drop_keep(uint32_t row_key, uint32_t site, uint32_t col, uint32_t thr) {
    uint32_t c = site * 4096u + col;
    return mix32(row_key ^ (c * 0x85EBCA77u)) >= thr;
}
// the dropout seed of a launch: the word in device memory when the caller gave one (aread_call.drop_seed_dev), else the argument
__device__ __forceinline__ uint32_t drop_seed_of(uint32_t seed, const uint32_t* seed_dev) { return seed_dev ? *seed_dev : seed; }
static inline uint32_t drop_threshold(float p) {
    double t = (double)p * 4294967296.0;
    t = t + 0.5;
    if (t >= 4294967295.0) return 0xFFFFFFFFu;
    if (t <= 0.0) return 0u;
    return (uint32_t)t;
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
