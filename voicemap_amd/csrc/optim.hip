// Keras-2.2.2 Adam(clipnorm=1.) over ONE flat fp32 parameter / gradient buffer
// (experiments/train_siamese.py:56; update rule in SURVEY.md section 8a-Adam).  Multi-tensor by construction:
// the 20 trainable tensors live back to back in one allocation, so the global-norm clip is one reduction and
// the update one streaming pass (3 reads + 3 writes of ~4 MB).
#include "common.hpp"

namespace vm {

constexpr int SQ_BLOCKS = 256;

__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ part) {
    __shared__ double red[4];
    // eight elements in flight per thread (a thread walks ~16 at cfg-A: one load after the other made this 4 MB pass 8 us of the tail)
    constexpr int64_t STRIDE = (int64_t)SQ_BLOCKS * 256;
    double s8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * STRIDE < n; i += 8 * STRIDE) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = g[i + u * STRIDE];
#pragma unroll
        for (int u = 0; u < 8; ++u) s8[u] += (double)v[u] * (double)v[u];
    }
    for (; i < n; i += STRIDE) {
        const double v = (double)g[i];
        s8[0] += v * v;
    }
    double s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void sqnorm_final_kernel(const double* __restrict__ part, float* __restrict__ out) {
    __shared__ double red[4];
    double s = part[threadIdx.x];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)(red[0] + red[1] + red[2] + red[3]);
}

__global__ __launch_bounds__(256) void adam_clip_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, int64_t n, float lr_t, float b1, float b2,
                                                        float eps, float clipnorm, float prescale,
                                                        float* __restrict__ sqnorm, const double* __restrict__ sq_parts,
                                                        int skip_nonfinite, int32_t* __restrict__ skipped) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float sq = 0.f;
    if (sq_parts != nullptr) {
        // the SQ_BLOCKS partials of vm_grad_sqnorm summed here, by every workgroup in sqnorm_final_kernel's order (one launch less
        // in the step's tail); workgroup 0 publishes the value
        __shared__ double red[4];
        double s = wave_sum_d(sq_parts[threadIdx.x]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        sq = (float)(red[0] + red[1] + red[2] + red[3]);
        if (i == 0 && sqnorm != nullptr) sqnorm[0] = sq;
    } else if (sqnorm != nullptr) {
        sq = sqnorm[0];
    }
    if (skip_nonfinite) {  // loss-scaled f16 training: an overflowed gradient costs this step, not the model
        const bool bad = !isfinite(sq);
        if (i == 0 && skipped != nullptr && bad) skipped[0] += 1;   // a running count: one thread of the launch, no atomics needed
        if (bad) return;
    }
    if (i >= n) return;
    float scale = prescale;
    if (clipnorm > 0.f) {
        const float norm = prescale * sqrtf(sq);
        if (norm >= clipnorm) scale = prescale * (clipnorm / norm);
    }
    const float gi = g[i] * scale;
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] - lr_t * mi / (sqrtf(vi) + eps);
}

}  // namespace vm

using namespace vm;

extern "C" int64_t vm_sqnorm_workspace_bytes(int64_t n) { return SQ_BLOCKS * (int64_t)sizeof(double); }

extern "C" int vm_grad_sqnorm(const float* g, int64_t n, void* ws, float* sqnorm, void* stream) {
    VM_REQUIRE(g && ws && n > 0, "vm_grad_sqnorm: bad argument");
    hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(SQ_BLOCKS), dim3(256), 0, (hipStream_t)stream, g, n, (double*)ws);
    if (sqnorm != nullptr)  // NULL: the partials only (vm_adam_clip_step adds them itself when it is given ws as sqnorm_parts)
        hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(SQ_BLOCKS), 0, (hipStream_t)stream, (const double*)ws, sqnorm);
    return check_launch("vm_grad_sqnorm");
}

extern "C" int vm_adam_clip_step(float* p, const float* g, float* m, float* v, int64_t n, float lr_t, float beta1, float beta2,
                                 float eps, float clipnorm, float grad_prescale, float* sqnorm, const void* sqnorm_parts,
                                 int skip_nonfinite, int32_t* skipped, void* stream) {
    VM_REQUIRE(p && g && m && v && n > 0, "vm_adam_clip_step: bad argument");
    VM_REQUIRE((clipnorm <= 0.f && !skip_nonfinite) || sqnorm != nullptr || sqnorm_parts != nullptr,
               "vm_adam_clip_step: clipnorm / skip_nonfinite need sqnorm or sqnorm_parts");
    hipLaunchKernelGGL(adam_clip_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr_t,
                       beta1, beta2, eps, clipnorm, grad_prescale, sqnorm, (const double*)sqnorm_parts, skip_nonfinite, skipped);
    return check_launch("vm_adam_clip_step");
}
