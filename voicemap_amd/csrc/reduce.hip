// Deterministic sum over split-K / per-window partial slabs:  out[i] = sum_k ws[k][i]  in two fixed-order stages
// (SLAB_RCH fp64 partials per element, then their sum), so a long reduction over hundreds of slabs is spread over
// many workgroups instead of being one serial loop per output element.
#include "common.hpp"

namespace vm {

__global__ __launch_bounds__(256) void slab_stage1_kernel(const float* __restrict__ ws, int64_t slabs, int64_t nel,
                                                          double* __restrict__ part) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nel) return;
    const int64_t per = (slabs + gridDim.y - 1) / gridDim.y;
    const int64_t lo = blockIdx.y * per;
    int64_t hi = lo + per;
    if (hi > slabs) hi = slabs;
    // eight slabs in flight per thread (the block-1 weight gradient: 1024 slabs, 16 per thread -- two at a time was eight memory round
    // trips in a row on the step's tail), summed in a fixed order
    double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    int64_t k = lo;
    for (; k + 8 <= hi; k += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ws[(k + u) * nel + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) s[u] += (double)v[u];
    }
    for (; k < hi; ++k) s[0] += (double)ws[k * nel + i];
    part[(int64_t)blockIdx.y * nel + i] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

// one stage: few slabs and enough elements to fill the chip on their own
__global__ __launch_bounds__(256) void slab_single_kernel(const float* __restrict__ ws, int64_t slabs, int64_t nel, int64_t n0,
                                                          float* __restrict__ out0, float* __restrict__ out1) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nel) return;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int64_t k = 0;
    for (; k + 4 <= slabs; k += 4) {
        s0 += (double)ws[k * nel + i];
        s1 += (double)ws[(k + 1) * nel + i];
        s2 += (double)ws[(k + 2) * nel + i];
        s3 += (double)ws[(k + 3) * nel + i];
    }
    for (; k < slabs; ++k) s0 += (double)ws[k * nel + i];
    const double s = (s0 + s1) + (s2 + s3);
    if (i < n0) {
        out0[i] = (float)s;
    } else {
        out1[i - n0] = (float)s;
    }
}

__global__ __launch_bounds__(256) void slab_stage2_kernel(const double* __restrict__ part, int rch, int64_t nel, int64_t n0,
                                                          float* __restrict__ out0, float* __restrict__ out1) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nel) return;
    double s8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // eight partials in flight, fixed order
    int k = 0;
    for (; k + 8 <= rch; k += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(k + u) * nel + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) s8[u] += v[u];
    }
    for (; k < rch; ++k) s8[0] += part[(int64_t)k * nel + i];
    const double s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
    if (i < n0) {
        out0[i] = (float)s;
    } else {
        out1[i - n0] = (float)s;
    }
}

int64_t slab_sum_part_bytes(int64_t nel) { return (int64_t)SLAB_RCH * nel * (int64_t)sizeof(double) + 64; }

int slab_sum(const float* ws, int64_t slabs, int64_t nel, float* out0, int64_t n0, float* out1, void* part, hipStream_t stream) {
    uintptr_t a = reinterpret_cast<uintptr_t>(part);
    a = (a + 63) / 64 * 64;
    double* pp = reinterpret_cast<double*>(a);
    // partial rows: enough workgroups to fill the chip (~2048), at least 4 slabs per partial, at most SLAB_RCH
    const int64_t blocks = cdiv(nel, 256);
    int64_t rch = cdiv(2048, blocks);
    if (rch > slabs / 4) rch = slabs / 4;
    if (rch > SLAB_RCH) rch = SLAB_RCH;
    // up to 64 slabs of >= 256 workgroups' worth of elements (the split-K slabs of the k=3 weight gradients at cfg-A: 16..36 slabs of
    // 98 K..590 K elements) go through ONE launch: the second stage's launch latency (~4.5 us) is more than the parallelism it buys
    if (slabs <= 64 && blocks >= 256) rch = 1;
    if (rch <= 1) {
        hipLaunchKernelGGL(slab_single_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, ws, slabs, nel, n0, out0, out1);
        return check_launch("slab_sum");
    }
    hipLaunchKernelGGL(slab_stage1_kernel, dim3((unsigned)blocks, (unsigned)rch), dim3(256), 0, stream, ws, slabs, nel, pp);
    hipLaunchKernelGGL(slab_stage2_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const double*)pp, (int)rch, nel, n0,
                       out0, out1);
    return check_launch("slab_sum");
}

}  // namespace vm
