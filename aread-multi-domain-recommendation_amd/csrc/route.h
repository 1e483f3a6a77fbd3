// route.h -- workspace layout of the lookup routing (route.hip), shared with the fused table optimizer (optim.hip).
#pragma once
#include "common.h"

#define RT_THREADS 256
#define RT_ITEMS 16
#define RT_BLOCK (RT_THREADS * RT_ITEMS)

struct RouteWs {
    int64_t n_keys, n_blk;
    int64_t off_flags, off_slotmap, off_bsum, total;
};
static inline int64_t rt_align(int64_t x) { return (x + 255) & ~(int64_t)255; }
static int route_layout(int64_t n_keys, RouteWs* L) {
    if (n_keys <= 0 || n_keys >= (1ll << 31)) return -1;
    L->n_keys = n_keys;
    L->n_blk = (n_keys + RT_BLOCK - 1) / RT_BLOCK;
    int64_t o = 0;
    L->off_flags = o;   o = rt_align(o + L->n_blk * RT_BLOCK);          // uint8, padded to whole blocks
    L->off_slotmap = o; o = rt_align(o + n_keys * 4);
    L->off_bsum = o;    o = rt_align(o + (L->n_blk + 1) * 4);
    L->total = o;
    return 0;
}

