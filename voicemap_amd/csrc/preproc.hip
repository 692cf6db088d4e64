// preprocess_instances(downsampling) + whiten (voicemap/utils.py:22-34, 88-101) fused on the GPU:
// strided decimation (no anti-alias filter, utils.py:29), per-window mean removal (utils.py:94-95) and ONE
// scale per tower rms/sqrt(mean(batch^2)) over the un-centred decimated batch (utils.py:98).  The reference does
// this in float64 numpy on the host; sums are accumulated in fp64 here.  Output already carries conv-1's SAME
// halo (15 zeros left, 16 right), i.e. row length L0 + 31.
#include "common.hpp"

namespace vm {

constexpr int HALO_L = 15, HALO = 31;

template <typename R> __device__ inline float raw_to_f(R v);
template <> __device__ inline float raw_to_f<float>(float v) { return v; }
template <> __device__ inline float raw_to_f<int16_t>(int16_t v) { return (float)v * (1.0f / 32768.0f); }

// pass 1: per-window sum and sum of squares of the decimated samples.  grid = n_windows.
template <typename R>
__global__ __launch_bounds__(256) void whiten_stats_kernel(const R* __restrict__ raw, int64_t raw_len, int ds, int64_t L0,
                                                           double* __restrict__ wsum, double* __restrict__ wsq) {
    __shared__ double red[2][4];
    const int64_t n = blockIdx.x;
    const R* r = raw + n * raw_len;
    double s = 0.0, q = 0.0;
    for (int64_t i = threadIdx.x; i < L0; i += 256) {
        const double v = (double)raw_to_f<R>(r[i * ds]);
        s += v;
        q += v * v;
    }
    s = wave_sum_d(s);
    q = wave_sum_d(q);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = s;
        red[1][threadIdx.x >> 6] = q;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        wsum[n] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        wsq[n] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

// pass 2: write (x - mean_n) * scale_tower with the halo.  grid = (ceil((L0+31)/256), n_windows).
template <typename R>
__global__ __launch_bounds__(256) void whiten_apply_kernel(const R* __restrict__ raw, int64_t raw_len, int ds, int64_t L0,
                                                           int whitening, float rms, int64_t wpt, const double* __restrict__ wsum,
                                                           const double* __restrict__ wsq, float* __restrict__ out) {
    __shared__ double s_mean, s_scale;
    const int64_t n = blockIdx.y;
    if (threadIdx.x == 0) {
        if (whitening) {
            const int64_t tw = n / wpt;
            double q = 0.0;
            for (int64_t j = 0; j < wpt; ++j) q += wsq[tw * wpt + j];
            s_mean = wsum[n] / (double)L0;
            s_scale = (double)rms / sqrt(q / ((double)wpt * (double)L0));
        } else {
            s_mean = 0.0;
            s_scale = 1.0;
        }
    }
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= L0 + HALO) return;
    const int64_t t = i - HALO_L;
    float v = 0.f;
    if (t >= 0 && t < L0) v = (float)(((double)raw_to_f<R>(raw[n * raw_len + t * ds]) - s_mean) * s_scale);
    out[n * (L0 + HALO) + i] = v;
}

}  // namespace vm

using namespace vm;

extern "C" int64_t vm_decimate_whiten_workspace_bytes(int64_t n_windows) { return 2 * n_windows * (int64_t)sizeof(double); }

extern "C" int vm_decimate_whiten(const void* raw, int raw_is_i16, int64_t n_windows, int64_t raw_len, int downsampling,
                                  int whitening, float rms, int64_t windows_per_tower, float* out, void* ws, void* stream) {
    VM_REQUIRE(raw && out && ws, "vm_decimate_whiten: null pointer");
    VM_REQUIRE(n_windows > 0 && raw_len > 0 && downsampling > 0 && windows_per_tower > 0, "vm_decimate_whiten: bad sizes");
    VM_REQUIRE(n_windows % windows_per_tower == 0, "vm_decimate_whiten: n_windows must be a multiple of windows_per_tower");
    const int64_t L0 = (raw_len + downsampling - 1) / downsampling;  // len(x[::d])
    double* wsum = (double*)ws;
    double* wsq = wsum + n_windows;
    const dim3 g2((unsigned)cdiv(L0 + HALO, 256), (unsigned)n_windows);
    if (raw_is_i16) {
        if (whitening)
            hipLaunchKernelGGL((whiten_stats_kernel<int16_t>), dim3((unsigned)n_windows), dim3(256), 0, (hipStream_t)stream,
                               (const int16_t*)raw, raw_len, downsampling, L0, wsum, wsq);
        hipLaunchKernelGGL((whiten_apply_kernel<int16_t>), g2, dim3(256), 0, (hipStream_t)stream, (const int16_t*)raw, raw_len,
                           downsampling, L0, whitening, rms, windows_per_tower, wsum, wsq, out);
    } else {
        if (whitening)
            hipLaunchKernelGGL((whiten_stats_kernel<float>), dim3((unsigned)n_windows), dim3(256), 0, (hipStream_t)stream,
                               (const float*)raw, raw_len, downsampling, L0, wsum, wsq);
        hipLaunchKernelGGL((whiten_apply_kernel<float>), g2, dim3(256), 0, (hipStream_t)stream, (const float*)raw, raw_len,
                           downsampling, L0, whitening, rms, windows_per_tower, wsum, wsq, out);
    }
    return check_launch("vm_decimate_whiten");
}
