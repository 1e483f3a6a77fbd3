// l2.hip -- dense L2 regulariser of the embedding table: one streaming pass that produces both
// sum(w^2) (block partials, reduced in fixed order) and the dense gradient 2*l2*w.
// Replaces the table term of BaseModel.get_regularization_loss (model/layer.py:96-112).
// HBM-bound: reads n*4 bytes, writes n*4 bytes; 16 B per lane, grid-stride over 2048 workgroups.
#include "common.h"

#define L2_THREADS 256
#define L2_BLOCKS 2048

typedef float l2_f4 __attribute__((ext_vector_type(4)));

// The pass is cut into L2_BLOCKS "virtual blocks" (one partial sum each, a grid-stride sweep each); a launch of fewer
// real workgroups lets every workgroup walk several virtual blocks.  Result and partial sums are therefore bitwise
// independent of the launch width, which is the throttle of aread_l2_table_throttled.
__global__ __launch_bounds__(L2_THREADS) void k_l2_table(const float* __restrict__ w, int64_t n, float gscale_host,
                                                         float* __restrict__ grad, float* __restrict__ partial,
                                                         const float* __restrict__ gscale_dev) {
    // gscale_dev (nullable): a device scalar factor of the gradient -- autograd's dL/dreg arrives as a device tensor and
    // reading it on the host would stall the stream (aread_l2_table_dev)
    const float gscale = gscale_dev ? gscale_host * gscale_dev[0] : gscale_host;
    const int64_t n4 = n >> 2;
    const float4* w4 = (const float4*)w;
    float4* g4 = (float4*)grad;
    const int64_t stride = (int64_t)L2_BLOCKS * L2_THREADS;
    __shared__ float s[L2_THREADS / WAVE];
    for (int vb = blockIdx.x; vb < L2_BLOCKS; vb += gridDim.x) {
        float acc = 0.f;
        int64_t i = (int64_t)vb * L2_THREADS + threadIdx.x;
        for (; i + 3 * stride < n4; i += 4 * stride) {           // four independent 16-byte loads in flight per lane
            l2_f4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load((const l2_f4*)(w4 + i + u * stride));   // streamed once
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc += v[u].x * v[u].x + v[u].y * v[u].y + v[u].z * v[u].z + v[u].w * v[u].w;
                if (grad) __builtin_nontemporal_store(v[u] * gscale, (l2_f4*)(g4 + i + u * stride));
            }
        }
        for (; i < n4; i += stride) {
            const l2_f4 v = __builtin_nontemporal_load((const l2_f4*)(w4 + i));
            acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            if (grad) __builtin_nontemporal_store(v * gscale, (l2_f4*)(g4 + i));
        }
        if (vb == 0 && threadIdx.x < (n & 3)) {                 // scalar tail
            const int64_t i = (n4 << 2) + threadIdx.x;
            const float v = w[i];
            acc += v * v;
            if (grad) grad[i] = gscale * v;
        }
        if (!partial) continue;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.f;
            for (int i = 0; i < L2_THREADS / WAVE; ++i) t += s[i];
            partial[vb] = t;
        }
    }
}

__global__ __launch_bounds__(256) void k_l2_finish(const float* __restrict__ partial, int n, float l2,
                                                   float* __restrict__ out, int accumulate) {
    __shared__ double s[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += (double)partial[i];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float v = (float)((double)l2 * s[0]);
        out[0] = accumulate ? out[0] + v : v;
    }
}

extern "C" int aread_l2_partials(void) { return L2_BLOCKS; }

extern "C" int aread_l2_table_throttled(const float* w, int64_t n, float l2, float grad_scale, float* grad, float* partial,
                                        int max_workgroups, void* stream) {
    AR_CHECK_ARG(w != nullptr && n > 0, "aread_l2_table: empty input");
    AR_CHECK_ARG(((uintptr_t)w & 15) == 0 && ((uintptr_t)grad & 15) == 0, "aread_l2_table: 16-byte alignment");
    AR_CHECK_ARG(grad || partial, "aread_l2_table: nothing to do");
    int blocks = L2_BLOCKS;
    if (max_workgroups > 0 && max_workgroups < blocks) blocks = max_workgroups;
    hipLaunchKernelGGL(k_l2_table, dim3(blocks), dim3(L2_THREADS), 0, (hipStream_t)stream, w, n,
                       2.0f * l2 * grad_scale, grad, partial, (const float*)nullptr);
    AR_LAUNCH_CHECK();
    return AREAD_OK;
}

extern "C" int aread_l2_table_dev(const float* w, int64_t n, float l2, const float* grad_scale_dev, float* grad, void* stream) {
    AR_CHECK_ARG(w != nullptr && n > 0 && grad && grad_scale_dev, "aread_l2_table_dev: null pointer / empty input");
    AR_CHECK_ARG(((uintptr_t)w & 15) == 0 && ((uintptr_t)grad & 15) == 0, "aread_l2_table_dev: 16-byte alignment");
    hipLaunchKernelGGL(k_l2_table, dim3(L2_BLOCKS), dim3(L2_THREADS), 0, (hipStream_t)stream, w, n, 2.0f * l2, grad,
                       (float*)nullptr, grad_scale_dev);
    AR_LAUNCH_CHECK();
    return AREAD_OK;
}

extern "C" int aread_l2_table(const float* w, int64_t n, float l2, float grad_scale, float* grad, float* partial,
                              void* stream) {
    return aread_l2_table_throttled(w, n, l2, grad_scale, grad, partial, 0, stream);
}

extern "C" int aread_l2_finish(const float* partial, int n_partial, float l2, float* loss_out, int accumulate,
                               void* stream) {
    AR_CHECK_ARG(partial && loss_out && n_partial > 0, "aread_l2_finish: bad arguments");
    hipLaunchKernelGGL(k_l2_finish, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, n_partial, l2, loss_out,
                       accumulate);
    AR_LAUNCH_CHECK();
    return AREAD_OK;
}
