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

constexpr int WS_SEG = 8;  // partial-sum segments per window

// pass 1: partial sum and sum of squares of the decimated samples.  grid = (n_windows, WS_SEG).
template <typename R>
__global__ __launch_bounds__(256) void whiten_stats_kernel(const R* __restrict__ raw, const int64_t* __restrict__ offsets,
                                                           int64_t raw_len, int ds, int64_t L0, double* __restrict__ psum,
                                                           double* __restrict__ psq) {
    __shared__ double red[2][4];
    const int64_t n = blockIdx.x;
    const int seg = blockIdx.y;
    const R* r = raw + (offsets ? offsets[n] : n * raw_len);  // window n: its own crop of a resident recording, or row n
    const int64_t per = (L0 + WS_SEG - 1) / WS_SEG;
    const int64_t i1 = (seg + 1) * per < L0 ? (seg + 1) * per : L0;
    double s = 0.0, q = 0.0;
    for (int64_t i = seg * per + threadIdx.x; i < i1; i += 256) {
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
        psum[n * WS_SEG + seg] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        psq[n * WS_SEG + seg] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// pass 2: per-window mean and ONE scale per tower (utils.py:94-98) from the partials -- recomputed by every workgroup in the same
// fixed order (1 K doubles from L2: cheaper than the launch a separate finalize kernel costs) -- then (x - mean_n) * scale_tower
// with the halo.  grid = (WA_SPLIT, n_windows): a workgroup writes one slice of a window's row.
constexpr int WA_SPLIT = 8;
template <typename R>
__global__ __launch_bounds__(256) void whiten_apply_kernel(const R* __restrict__ raw, const int64_t* __restrict__ offsets,
                                                           int64_t raw_len, int ds, int64_t L0, int whitening, int64_t wpt, float rms,
                                                           const double* __restrict__ psum, const double* __restrict__ psq,
                                                           float* __restrict__ out) {
    __shared__ double red[4];
    const int64_t n = blockIdx.y;
    const R* r = raw + (offsets ? offsets[n] : n * raw_len);
    double m = 0.0, sc = 1.0;
    if (whitening) {
        const int64_t tw = n / wpt;
        double q = 0.0;
        for (int64_t j = threadIdx.x; j < wpt * WS_SEG; j += 256) q += psq[tw * wpt * WS_SEG + j];
        q = wave_sum_d(q);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q;
        __syncthreads();
        const double tot = (red[0] + red[1]) + (red[2] + red[3]);
        sc = (double)rms / sqrt(tot / ((double)wpt * (double)L0));
        double s = 0.0;
        for (int k = 0; k < WS_SEG; ++k) s += psum[n * WS_SEG + k];
        m = s / (double)L0;
    }
    const int64_t row = L0 + HALO, per = (row + WA_SPLIT - 1) / WA_SPLIT;
    const int64_t i1 = (blockIdx.x + 1) * per < row ? (blockIdx.x + 1) * per : row;
    for (int64_t i = blockIdx.x * per + threadIdx.x; i < i1; i += 256) {
        const int64_t t = i - HALO_L;
        float v = 0.f;
        if (t >= 0 && t < L0) v = (float)(((double)raw_to_f<R>(r[t * ds]) - m) * sc);
        out[n * row + i] = v;
    }
}

}  // namespace vm

using namespace vm;

extern "C" int64_t vm_decimate_whiten_workspace_bytes(int64_t n_windows) {
    return (2 * WS_SEG + 2) * n_windows * (int64_t)sizeof(double);  // the partials (+ two rows no longer used)
}

static int decimate_whiten_impl(const char* what, const void* raw, int raw_is_i16, const int64_t* offsets, int64_t n_windows,
                                int64_t raw_len, int downsampling, int whitening, float rms, int64_t windows_per_tower, float* out,
                                void* ws, void* stream) {
    const int64_t L0 = (raw_len + downsampling - 1) / downsampling;  // len(x[::d])
    double* psum = (double*)ws;
    double* psq = psum + n_windows * WS_SEG;
    const dim3 g1((unsigned)n_windows, WS_SEG);
    const dim3 g2(WA_SPLIT, (unsigned)n_windows);
    hipStream_t st = (hipStream_t)stream;
    if (raw_is_i16) {
        if (whitening)
            hipLaunchKernelGGL((whiten_stats_kernel<int16_t>), g1, dim3(256), 0, st, (const int16_t*)raw, offsets, raw_len,
                               downsampling, L0, psum, psq);
        hipLaunchKernelGGL((whiten_apply_kernel<int16_t>), g2, dim3(256), 0, st, (const int16_t*)raw, offsets, raw_len, downsampling,
                           L0, whitening, windows_per_tower, rms, psum, psq, out);
    } else {
        if (whitening)
            hipLaunchKernelGGL((whiten_stats_kernel<float>), g1, dim3(256), 0, st, (const float*)raw, offsets, raw_len, downsampling,
                               L0, psum, psq);
        hipLaunchKernelGGL((whiten_apply_kernel<float>), g2, dim3(256), 0, st, (const float*)raw, offsets, raw_len, downsampling, L0,
                           whitening, windows_per_tower, rms, psum, psq, out);
    }
    return check_launch(what);
}

extern "C" int vm_decimate_whiten(const void* raw, int raw_is_i16, int64_t n_windows, int64_t raw_len, int downsampling,
                                  int whitening, float rms, int64_t windows_per_tower, float* out, void* ws, void* stream) {
    VM_REQUIRE(raw && out && ws, "vm_decimate_whiten: null pointer");
    VM_REQUIRE(n_windows > 0 && raw_len > 0 && downsampling > 0 && windows_per_tower > 0, "vm_decimate_whiten: bad sizes");
    VM_REQUIRE(n_windows % windows_per_tower == 0, "vm_decimate_whiten: n_windows must be a multiple of windows_per_tower");
    return decimate_whiten_impl("vm_decimate_whiten", raw, raw_is_i16, nullptr, n_windows, raw_len, downsampling, whitening, rms,
                                windows_per_tower, out, ws, stream);
}

extern "C" int vm_crop_decimate_whiten(const void* audio, int raw_is_i16, const int64_t* offsets, int64_t n_windows,
                                       int64_t raw_len, int downsampling, int whitening, float rms, int64_t windows_per_tower,
                                       float* out, void* ws, void* stream) {
    VM_REQUIRE(audio && offsets && out && ws, "vm_crop_decimate_whiten: null pointer");
    VM_REQUIRE(n_windows > 0 && raw_len > 0 && downsampling > 0 && windows_per_tower > 0, "vm_crop_decimate_whiten: bad sizes");
    VM_REQUIRE(n_windows % windows_per_tower == 0, "vm_crop_decimate_whiten: n_windows must be a multiple of windows_per_tower");
    return decimate_whiten_impl("vm_crop_decimate_whiten", audio, raw_is_i16, offsets, n_windows, raw_len, downsampling, whitening,
                                rms, windows_per_tower, out, ws, stream);
}
