// Block 1 of the voicemap encoder: Conv1D(F, 32, padding='same', activation='relu') on the raw waveform
// (voicemap/models.py:13-16).  C_in = 1, so the "GEMM" has K = 32 and is output-bandwidth bound: 4 % of the
// network's FLOPs but the largest activation.  This version keeps the arithmetic in fp32 on the vector ALUs:
// lanes <-> output channels (coalesced channels-last stores), the 32 filter taps live in registers, the
// waveform tile lives in LDS and is read with wave-uniform (broadcast) addresses, 4 positions per pass.
#include "common.hpp"

namespace vm {

constexpr int C1_K = 32;
constexpr int C1_TT = 256;  // positions per workgroup

template <typename T>
__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, int64_t L, int F, int tilesT,
                                                        T* __restrict__ z, float* __restrict__ stat_sum,
                                                        float* __restrict__ stat_sq) {
    __shared__ __attribute__((aligned(16))) float xs[C1_TT + C1_K];
    __shared__ float red[2][4][64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t n = blockIdx.x / tilesT;
    const int tile = (int)(blockIdx.x % tilesT);
    const int64_t t0 = (int64_t)tile * C1_TT;
    const float* xrow = x + n * (L + C1_K - 1);
    for (int i = tid; i < C1_TT + C1_K - 1; i += 256) {
        const int64_t t = t0 + i;
        xs[i] = (t < L + C1_K - 1) ? xrow[t] : 0.f;
    }
    __syncthreads();

    for (int cc = 0; cc < F; cc += 64) {
        const int c = cc + lane;
        const bool cok = c < F;
        float wr[C1_K];
#pragma unroll
        for (int k = 0; k < C1_K; ++k) wr[k] = cok ? w[k * F + c] : 0.f;
        const float bv = cok ? bias[c] : 0.f;
        float csum = 0.f, csq = 0.f;
        for (int q = 0; q < 16; ++q) {
            const int tl = wv * 64 + q * 4;
            float xv[C1_K + 3];
#pragma unroll
            for (int i = 0; i < C1_K + 3; ++i) xv[i] = xs[tl + i];
            float a0 = bv, a1 = bv, a2 = bv, a3 = bv;
#pragma unroll
            for (int k = 0; k < C1_K; ++k) {
                a0 = fmaf(wr[k], xv[k], a0);
                a1 = fmaf(wr[k], xv[k + 1], a1);
                a2 = fmaf(wr[k], xv[k + 2], a2);
                a3 = fmaf(wr[k], xv[k + 3], a3);
            }
            float av[4] = {a0, a1, a2, a3};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t t = t0 + tl + j;
                if (cok && t < L) {
                    const float v = av[j] > 0.f ? av[j] : 0.f;
                    const T tv = Elem<T>::from_f(v);
                    z[(n * L + t) * F + c] = tv;
                    const float vr = Elem<T>::to_f(tv);
                    csum += vr;
                    csq += vr * vr;
                }
            }
        }
        if (stat_sum != nullptr) {
            red[0][wv][lane] = csum;
            red[1][wv][lane] = csq;
            __syncthreads();
            if (wv == 0 && cok) {
                const int64_t row = n * tilesT + tile;
                stat_sum[row * F + c] = red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane];
                stat_sq[row * F + c] = red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane];
            }
            __syncthreads();
        }
    }
}

// dW[k][c] partial for one window: sum_t x[n][t+k] * du[n][t][c].  grid = (n_windows, ceil(F/64)).
constexpr int C1_WT = 1024;  // positions staged per iteration

template <typename T>
__global__ __launch_bounds__(256) void conv1_wgrad_kernel(const float* __restrict__ x, const T* __restrict__ du, int64_t L,
                                                          int F, float* __restrict__ ws) {
    __shared__ __attribute__((aligned(16))) float xs[C1_WT + C1_K];
    __shared__ float red[4][C1_K][64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t n = blockIdx.x;
    const int c = blockIdx.y * 64 + lane;
    const bool cok = c < F;
    const float* xrow = x + n * (L + C1_K - 1);
    const T* durow = du + (n * (L + 2) + 1) * F;  // skip the halo row

    float acc[C1_K];
#pragma unroll
    for (int k = 0; k < C1_K; ++k) acc[k] = 0.f;

    for (int64_t t0 = 0; t0 < L; t0 += C1_WT) {
        __syncthreads();
        for (int i = tid; i < C1_WT + C1_K - 1; i += 256) {
            const int64_t t = t0 + i;
            xs[i] = (t < L + C1_K - 1) ? xrow[t] : 0.f;
        }
        __syncthreads();
        for (int q = 0; q < C1_WT / 4 / 4; ++q) {
            const int tl = wv * (C1_WT / 4) + q * 4;
            float d[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t t = t0 + tl + j;
                d[j] = (cok && t < L) ? Elem<T>::to_f(durow[t * F + c]) : 0.f;
            }
            float xv[C1_K + 3];
#pragma unroll
            for (int i = 0; i < C1_K + 3; ++i) xv[i] = xs[tl + i];
#pragma unroll
            for (int k = 0; k < C1_K; ++k) {
                float a = acc[k];
                a = fmaf(d[0], xv[k], a);
                a = fmaf(d[1], xv[k + 1], a);
                a = fmaf(d[2], xv[k + 2], a);
                a = fmaf(d[3], xv[k + 3], a);
                acc[k] = a;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < C1_K; ++k) red[wv][k][lane] = acc[k];
    __syncthreads();
    if (cok) {
        for (int k = wv; k < C1_K; k += 4) {
            ws[(n * C1_K + k) * F + c] = red[0][k][lane] + red[1][k][lane] + red[2][k][lane] + red[3][k][lane];
        }
    }
}

}  // namespace vm

using namespace vm;

extern "C" int64_t vm_conv1_stat_rows(int64_t L) { return (L + C1_TT - 1) / C1_TT; }

extern "C" int vm_conv1_fwd(const float* x, const float* w, const float* bias, int64_t n_windows, int64_t L, int F, int dtype,
                            void* z, float* stat_sum, float* stat_sq, void* stream) {
    VM_REQUIRE(x && w && bias && z, "vm_conv1_fwd: null pointer");
    VM_REQUIRE(n_windows > 0 && L > 0 && F > 0, "vm_conv1_fwd: bad sizes");
    VM_REQUIRE(F % 8 == 0, "vm_conv1_fwd: filters must be a multiple of 8 (got %d)", F);
    VM_REQUIRE((stat_sum == nullptr) == (stat_sq == nullptr), "vm_conv1_fwd: stat_sum/stat_sq must both be set or NULL");
    const int tilesT = (int)vm_conv1_stat_rows(L);
    const int64_t grid = n_windows * tilesT;
    VM_REQUIRE(grid < (1LL << 31), "vm_conv1_fwd: grid too large");
    VM_DISPATCH_DTYPE(dtype, {
        hipLaunchKernelGGL((conv1_fwd_kernel<T>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, w, bias, L, F,
                           tilesT, (T*)z, stat_sum, stat_sq);
    });
    return check_launch("vm_conv1_fwd");
}

extern "C" int64_t vm_conv1_wgrad_workspace_bytes(int64_t n_windows, int F) {
    return n_windows * C1_K * (int64_t)F * (int64_t)sizeof(float) + slab_sum_part_bytes((int64_t)C1_K * F);
}

extern "C" int vm_conv1_wgrad(const float* x, const void* du, int64_t n_windows, int64_t L, int F, int dtype, float* ws,
                              float* grad_w, void* stream) {
    VM_REQUIRE(x && du && ws && grad_w, "vm_conv1_wgrad: null pointer");
    VM_REQUIRE(n_windows > 0 && L > 0 && F > 0 && F % 8 == 0, "vm_conv1_wgrad: bad sizes");
    VM_DISPATCH_DTYPE(dtype, {
        hipLaunchKernelGGL((conv1_wgrad_kernel<T>), dim3((unsigned)n_windows, (unsigned)((F + 63) / 64)), dim3(256), 0,
                           (hipStream_t)stream, x, (const T*)du, L, F, ws);
    });
    int rc = check_launch("vm_conv1_wgrad");
    if (rc) return rc;
    const int64_t n = (int64_t)C1_K * F;
    return slab_sum((const float*)ws, n_windows, n, grad_w, n, nullptr, ws + n_windows * n, (hipStream_t)stream);
}
