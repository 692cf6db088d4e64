// Log-mel front-end and the window-axis helpers of the 2-D CNN encoder variant (BASELINE.json config 4).
// NOT in the reference: the specification is this repository's own and the CPU oracle is
// oracle/voicemap_oracle.py (logmel_features, encoder2d_*), parity unpinned by construction.
//
//   vm_stft_logmel      raw 16 kHz windows -> log(mel(|STFT|^2) + floor), written as the block-1 input of the 2-D encoder
//   vm_stack_windows    Conv2D(3x3) lowering, forward:  X'[(b,m)][t][(dm, c)] = x[(b, m+dm-1)][t][c]   (zero outside the clip)
//   vm_fold_windows     ... and its adjoint:            dx[(b,m)][t][c]      = sum_dm dX'[(b, m-dm+1)][t][(dm, c)]
//   vm_pool_windows_*   the mel half of MaxPool2D(2, 2): max over window pairs (2m', 2m'+1) / gradient routing (first maximum wins)
//   vm_clip_max_*       the mel half of GlobalMaxPool2D: max over the windows of a clip / gradient routing
//
// Layout: a clip of the 2-D encoder is M "windows" (one per mel band, then per pooled band) of T positions each, window index
// (b, m) = b * M + m, channels last, and -- like every tensor that feeds a k = 3 convolution in this library -- T + 2 rows with
// a zero halo row at both ends.  A Conv2D(3 x 3, SAME) over (T, M) is then the library's k = 3 implicit-GEMM convolution along T
// (vm_conv_fwd / vm_conv_dgrad / vm_conv_wgrad) applied to the band-stacked tensor X' with 3 * C channels: the weights
// W2d[kt][km][ci][co] are W1d[kt][(km, ci)][co], no reshuffling.  BatchNorm, dropout and the time half of the pooling are the
// 1-D kernels' as they are.
#include "common.hpp"

namespace vm {

// ------------------------------------------------------------------------------------------------
// STFT -> power -> mel -> log as two chained fp32 MFMA GEMMs per tile of 32 frames (exact fp32 products: a bf16 DFT would cost
// the log-mel its dynamic range):
//   GEMM 1   D1[bin][frame] = sum_n basis[n][bin] * frame_samples[frame][n]     K = win_length, 256 bins x (re | im) = 512 rows
//            (the analysis window is folded into the basis; bins 0..255 only: with fmin = 0 and fmax = Nyquist the triangular
//             mel filters give bin 0 and bin 256 zero weight)
//   power    P = re^2 + im^2 in the accumulator registers
//   GEMM 2   D2[mel][frame] = sum_bin melw[bin][mel] * P[bin][frame]: the accumulator layout of GEMM 1 (lane <-> frame, register
//            <-> bin) IS a valid K-slot assignment of the B operand of the next MFMA (the P.V trick of attention kernels), so
//            the power spectrum never leaves the registers; the four waves' partial sums over their 64 bins meet in LDS.
// Workgroup = 4 waves = 32 frames of one clip; wave w owns bins [64 w, 64 w + 64).  The 32 frames are staged as an LDS matrix
// [32][win + 1] (odd pitch: the B operand read, one frame per lane, is conflict-free).
// ------------------------------------------------------------------------------------------------
constexpr int SF_FRAMES = 32;

template <typename TOUT, int NMB>  // NMB = n_mels / 32
__global__ __launch_bounds__(256) void stft_logmel_kernel(const void* __restrict__ raw, int is_int16, int64_t raw_len, int win, int hop,
                                                          int T, const float* __restrict__ basis,  // [win][512]: re bins | im bins
                                                          const float* __restrict__ melw,          // [256][32 * NMB]
                                                          float log_floor, TOUT* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xs = reinterpret_cast<float*>(smem);  // [32][win + 1]; later [4][32 * NMB][32] partial mel sums
    const int pitch = win + 1;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t clip = blockIdx.y;
    const int f0 = blockIdx.x * SF_FRAMES;
    const float* rf = reinterpret_cast<const float*>(raw) + clip * raw_len;
    const int16_t* ri = reinterpret_cast<const int16_t*>(raw) + clip * raw_len;
    // a wave per frame, 64 consecutive samples per load, every load of the wave's eight frames unconditional (clamped indices) and in
    // flight before the first store -- see stft_logmel_f16s_kernel: one load per branch was half of a wave's lifetime there
    {
        constexpr int FPW = SF_FRAMES / 4;
        float v[FPW][8];
        auto fetch = [&](auto* src, float scale) {
#pragma unroll
            for (int i = 0; i < FPW; ++i) {
                int fr = f0 + w + 4 * i;
                fr = fr < T ? fr : T - 1;
                const auto* p0 = src + (int64_t)fr * hop;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int n = lane + 64 * j;
                    v[i][j] = (float)p0[n < win ? n : win - 1] * scale;
                }
            }
        };
        if (is_int16) {
            fetch(ri, 1.0f / 32768.0f);
        } else {
            fetch(rf, 1.0f);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < FPW; ++i) {
            const int f = w + 4 * i;
            const bool fok = f0 + f < T;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int n = lane + 64 * j;
                if (n < win) xs[f * pitch + n] = fok ? v[i][j] : 0.f;
            }
        }
    }
    __syncthreads();

    const int col = lane & 31, kh = lane >> 5;
    f32x16 re[2], im[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) re[c][e] = im[c][e] = 0.f;
    const float* bre = basis + 64 * w + col;  // + n * 512 (+ 32 for the second bin block, + 256 for the imaginary part)
    const float* xrow = xs + col * pitch;
    for (int n0 = 0; n0 + 1 < win + 1; n0 += 2) {
        const int n = n0 + kh;
        const bool ok = n < win;  // odd window lengths: the last k-step has one term
        const float x = ok ? xrow[n] : 0.f;
        const float* bn = bre + (int64_t)(ok ? n : 0) * 512;
        const float a0 = bn[0], a1 = bn[32], a2 = bn[256], a3 = bn[288];
        re[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, x, re[0], 0, 0, 0);
        re[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, x, re[1], 0, 0, 0);
        im[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, x, im[0], 0, 0, 0);
        im[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, x, im[1], 0, 0, 0);
    }
    // power spectrum in place; then the mel GEMM over this wave's 64 bins
    f32x16 mel[NMB];
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
        for (int e = 0; e < 16; ++e) mel[mb][e] = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pw = re[c][r] * re[c][r] + im[c][r] * im[c][r];
            const int bin = 64 * w + 32 * c + (r & 3) + 8 * (r >> 2) + 4 * kh;  // the bin this lane's register r holds
            const float* wrow = melw + (int64_t)bin * (32 * NMB) + col;
#pragma unroll
            for (int mb = 0; mb < NMB; ++mb) mel[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[32 * mb], pw, mel[mb], 0, 0, 0);
        }
    }
    __syncthreads();  // every wave is done with the frame matrix
    float* red = xs;  // [4][32 * NMB][32]
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = 32 * mb + (r & 3) + 8 * (r >> 2) + 4 * kh;
            red[(w * 32 * NMB + m) * 32 + col] = mel[mb][r];
        }
    __syncthreads();
    const int n_mels = 32 * NMB;
    for (int i = tid; i < n_mels * 32; i += 256) {
        const int m = i >> 5, f = i & 31;
        if (f0 + f < T) {
            const float s = (red[(0 * n_mels + m) * 32 + f] + red[(1 * n_mels + m) * 32 + f]) +
                            (red[(2 * n_mels + m) * 32 + f] + red[(3 * n_mels + m) * 32 + f]);
            out[((int64_t)clip * n_mels + m) * (T + 2) + 1 + f0 + f] = Elem<TOUT>::from_f(logf(s + log_floor));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same front-end with GEMM 1 on the f16 matrix pipe (16-bit storage modes; the fp32-storage mode keeps the exact kernel above).
// An fp32 MFMA moves K = 2 per 64 clocks, v_mfma_f32_32x32x16_f16 K = 16 per 32: the DFT is 0.43 ms of fp32 MFMA time per 256 clips, the
// whole kernel ran at 87 % of that.  Both operands are split into two halves (v = hi + lo, both f16: 22 significand bits) and the three
// significant products accumulated in fp32: bh*xh + bl*xh + bh*xl -- 3 MFMAs per K = 16 instead of 8 per K = 16: 5.3 x less matrix time at
// ~2^-21 relative per product (fp32: 2^-24).  The samples are scaled by 256 before the split (raw audio is ~0.05: its low halves would
// be subnormal in half) and the power by 2^-16 afterwards (exact).  basis16: [hi | lo][512 rows: re bins, im bins][Kp] f16, Kp = win
// rounded up to 16 (vm_stft_split_basis).  Power, mel GEMM (fp32 MFMA, registers as operand), log: as above.
// ------------------------------------------------------------------------------------------------
template <typename TOUT, int NMB>
__global__ __launch_bounds__(256) void stft_logmel_f16s_kernel(const void* __restrict__ raw, int is_int16, int64_t raw_len, int win, int hop,
                                                               int T, const f16* __restrict__ basis16, int Kp,
                                                               const float* __restrict__ melw, float log_floor, TOUT* __restrict__ out,
                                                               TOUT* __restrict__ out_lo) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int pitch = Kp + 8;   // halves: a row is 2 Kp + 16 bytes, the 16-byte operand reads of 32 frames x 2 K-halves are conflict-free
    f16* xh = reinterpret_cast<f16*>(smem);
    f16* xl = xh + SF_FRAMES * pitch;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t clip = blockIdx.y;
    const int f0 = blockIdx.x * SF_FRAMES;
    const float* rf = reinterpret_cast<const float*>(raw) + clip * raw_len;
    const int16_t* ri = reinterpret_cast<const int16_t*>(raw) + clip * raw_len;
    // A wave per frame, 64 consecutive samples per load, and EVERY load of the wave's eight frames in flight before the first
    // conversion.  The loads are unconditional on clamped indices and the two sample types are two straight-line copies of the code:
    // written as `if (in range) v = is_int16 ? ... : ...` every load sat in its own branch with an s_waitcnt vmcnt(0) behind it --
    // 56 memory round trips in a row, half of a wave's lifetime (s_memtime: staging 49 %, GEMM 1 33 %, mel GEMM 12 %, sums + log 5 %).
    constexpr int FPW = SF_FRAMES / 4;   // frames per wave
    float v[FPW][8];
    auto fetch = [&](auto* src, float scale) {
#pragma unroll
        for (int i = 0; i < FPW; ++i) {
            int fr = f0 + w + 4 * i;
            fr = fr < T ? fr : T - 1;
            const auto* p0 = src + (int64_t)fr * hop;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int n = lane + 64 * j;
                v[i][j] = (float)p0[n < win ? n : win - 1] * scale;
            }
        }
    };
    if (is_int16) {
        fetch(ri, 256.0f / 32768.0f);
    } else {
        fetch(rf, 256.0f);
    }
    __builtin_amdgcn_sched_barrier(0);   // (else the scheduler sinks the loads back between the stores)
#pragma unroll
    for (int i = 0; i < FPW; ++i) {
        const int f = w + 4 * i;
        const bool fok = f0 + f < T;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = lane + 64 * j;
            if (n < Kp) {
                const float x = (fok && n < win) ? v[i][j] : 0.f;
                const f16 h = (f16)x;
                xh[f * pitch + n] = h;
                xl[f * pitch + n] = (f16)(x - (float)h);
            }
        }
    }
    __syncthreads();

    const int col = lane & 31, kh = lane >> 5;
    f32x16 re[2], im[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) re[c][e] = im[c][e] = 0.f;
    // basis16 is stored in operand order: [hi | lo][K step][16 row blocks][lane][8 halves] -- one operand load of a wave is 1 KB
    // contiguous (8 whole cache lines; K-contiguous rows touched 32 lines for 32 bytes each and refetched them every step)
    const f16* bh = basis16 + (int64_t)(2 * w) * 512 + lane * 8;   // row blocks: +0 re, +1 re', +8 im, +9 im'
    const f16* bl = bh + (int64_t)512 * Kp;
    const f16* xhr = xh + col * pitch + 8 * kh;
    const f16* xlr = xl + col * pitch + 8 * kh;
    f16x8 ah[4], al[4];   // the next step's basis operands are in flight under this step's MFMAs
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t ro = (int64_t)((q & 1) + (q >> 1) * 8) * 512;
        ah[q] = *reinterpret_cast<const f16x8*>(bh + ro);
        al[q] = *reinterpret_cast<const f16x8*>(bl + ro);
    }
    for (int n0 = 0; n0 < Kp; n0 += 16) {
        const f16x8 vxh = *reinterpret_cast<const f16x8*>(xhr + n0), vxl = *reinterpret_cast<const f16x8*>(xlr + n0);
        f16x8 ch[4], cl[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { ch[q] = ah[q]; cl[q] = al[q]; }
        const int n1 = n0 + 16 < Kp ? n0 + 16 : n0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t ro = (int64_t)n1 * 512 + ((q & 1) + (q >> 1) * 8) * 512;
            ah[q] = *reinterpret_cast<const f16x8*>(bh + ro);
            al[q] = *reinterpret_cast<const f16x8*>(bl + ro);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x16& acc = q < 2 ? re[q] : im[q - 2];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch[q], vxh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(cl[q], vxh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch[q], vxl, acc, 0, 0, 0);
        }
    }
    f32x16 mel[NMB];
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
        for (int e = 0; e < 16; ++e) mel[mb][e] = 0.f;
    // (the mel matrix operands come straight from global: staging this wave's rows through LDS first measured 0.25 against 0.21 ms)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pw = (re[c][r] * re[c][r] + im[c][r] * im[c][r]) * (1.0f / 65536.0f);   // the samples were scaled by 256
            const int bin = 64 * w + 32 * c + (r & 3) + 8 * (r >> 2) + 4 * kh;
            const float* wrow = melw + (int64_t)bin * (32 * NMB) + col;
#pragma unroll
            for (int mb = 0; mb < NMB; ++mb) mel[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[32 * mb], pw, mel[mb], 0, 0, 0);
        }
    }
    __syncthreads();  // every wave is done with the frame matrices
    float* red = reinterpret_cast<float*>(smem);  // [4][32 * NMB][32]
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = 32 * mb + (r & 3) + 8 * (r >> 2) + 4 * kh;
            red[(w * 32 * NMB + m) * 32 + col] = mel[mb][r];
        }
    __syncthreads();
    const int n_mels = 32 * NMB;
    for (int i = tid; i < n_mels * 32; i += 256) {
        const int m = i >> 5, f = i & 31;
        if (f0 + f < T) {
            const float sum = (red[(0 * n_mels + m) * 32 + f] + red[(1 * n_mels + m) * 32 + f]) +
                              (red[(2 * n_mels + m) * 32 + f] + red[(3 * n_mels + m) * 32 + f]);
            const float lm = logf(sum + log_floor);
            const int64_t o = ((int64_t)clip * n_mels + m) * (T + 2) + 1 + f0 + f;
            const TOUT hi = Elem<TOUT>::from_f(lm);
            out[o] = hi;
            // (vm_stft_logmel_f16s_split) what the storage type dropped: image = out + out_lo to 2 x the significand
            if (out_lo != nullptr) out_lo[o] = Elem<TOUT>::from_f(lm - Elem<TOUT>::to_f(hi));
        }
    }
}

// basis [win][512] fp32 -> [hi | lo][Kp / 16 steps][16 row blocks][64 lanes][8] f16 (lane = row % 32 + 32 * (K half of the step): the
// A operand of v_mfma_f32_32x32x16_f16 as it sits in registers), zero beyond win
__global__ void stft_split_basis_kernel(const float* __restrict__ basis, int win, int Kp, f16* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 512 * Kp) return;
    const int e = i & 7, lane = (i >> 3) & 63, rb = (i >> 9) & 15, step = i >> 13;
    const int row = 32 * rb + (lane & 31), k = 16 * step + 8 * (lane >> 5) + e;
    const float b = k < win ? basis[(int64_t)k * 512 + row] : 0.f;
    const f16 h = (f16)b;
    out[i] = h;
    out[(int64_t)512 * Kp + i] = (f16)(b - (float)h);
}

// ------------------------------------------------------------------------------------------------
// element-wise helpers.  One thread = one V-element vector (V = 16 bytes' worth when the channel count allows, else 1) of one window;
// grid = (windows, ceil(vectors per window / 256)): no 64-bit divisions on the element path, whole-vector loads and stores.
// ------------------------------------------------------------------------------------------------
template <typename T, int V> struct EwVec {   // V elements moved as one unit
    T v[V];
};
template <typename T, int V>
__device__ inline EwVec<T, V> ew_load(const T* p) {
    EwVec<T, V> r;
    if constexpr (V * sizeof(T) == 16) {
        *reinterpret_cast<u32x4*>(r.v) = *reinterpret_cast<const u32x4*>(p);
    } else {
#pragma unroll
        for (int e = 0; e < V; ++e) r.v[e] = p[e];
    }
    return r;
}
template <typename T, int V>
__device__ inline void ew_store(T* p, const EwVec<T, V>& r) {
    if constexpr (V * sizeof(T) == 16) {
        *reinterpret_cast<u32x4*>(p) = *reinterpret_cast<const u32x4*>(r.v);
    } else {
#pragma unroll
        for (int e = 0; e < V; ++e) p[e] = r.v[e];
    }
}

template <typename T, int V>
__global__ __launch_bounds__(256) void stack_windows_kernel(const T* __restrict__ x, int64_t n_clips, int M, int rows, int C, int Cs,
                                                            T* __restrict__ out) {
    const int per_row = Cs / V;
    const int64_t win = blockIdx.x;
    const int j = blockIdx.y * 256 + threadIdx.x;   // (row, vector of the stacked row)
    if (j >= rows * per_row) return;
    const int row = j / per_row, cs = (j - row * per_row) * V;
    const int m = (int)(win % M);
    const int dm = cs / C, c = cs - dm * C;  // V divides C on the vector path: a vector never straddles two bands
    const int ms = m + dm - 1;
    EwVec<T, V> v;
#pragma unroll
    for (int e = 0; e < V; ++e) v.v[e] = Elem<T>::from_f(0.f);
    if (dm < 3 && ms >= 0 && ms < M) v = ew_load<T, V>(x + ((win - m + ms) * rows + row) * (int64_t)C + c);
    ew_store<T, V>(out + (win * rows + row) * (int64_t)Cs + cs, v);
}

template <typename T, int V>
__global__ __launch_bounds__(256) void fold_windows_kernel(const T* __restrict__ dxs, int64_t n_clips, int M, int L, int C, int Cs,
                                                           int src_pad, T* __restrict__ dx) {
    const int per_row = C / V;
    const int64_t win = blockIdx.x;
    const int j = blockIdx.y * 256 + threadIdx.x;
    if (j >= L * per_row) return;
    const int t = j / per_row, c = (j - t * per_row) * V;
    const int m = (int)(win % M);
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
#pragma unroll
    for (int dm = 0; dm < 3; ++dm) {
        const int md = m - dm + 1;  // the window whose stacked band dm is this window
        if (md >= 0 && md < M) {
            const EwVec<T, V> s = ew_load<T, V>(dxs + ((win - m + md) * (L + 2 * src_pad) + src_pad + t) * (int64_t)Cs + dm * C + c);
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] += Elem<T>::to_f(s.v[e]);
        }
    }
    EwVec<T, V> o;
#pragma unroll
    for (int e = 0; e < V; ++e) o.v[e] = Elem<T>::from_f(acc[e]);
    ew_store<T, V>(dx + (win * L + t) * (int64_t)C + c, o);
}

template <typename T, int V>
__global__ __launch_bounds__(256) void pool_windows_fwd_kernel(const T* __restrict__ q, int64_t n_clips, int M, int rows, int C,
                                                               T* __restrict__ out) {
    const int Mo = M / 2;
    const int64_t wo = blockIdx.x;   // output window
    const int j = blockIdx.y * 256 + threadIdx.x;
    if (j >= rows * C / V) return;
    const int64_t e0 = (int64_t)j * V;
    const int64_t b = wo / Mo;
    const int mo = (int)(wo - b * Mo);
    const T* a = q + ((b * M + 2 * mo) * (int64_t)rows * C) + e0;
    const EwVec<T, V> x0 = ew_load<T, V>(a), x1 = ew_load<T, V>(a + (int64_t)rows * C);
    EwVec<T, V> o;
#pragma unroll
    for (int e = 0; e < V; ++e) o.v[e] = Elem<T>::to_f(x0.v[e]) >= Elem<T>::to_f(x1.v[e]) ? x0.v[e] : x1.v[e];
    ew_store<T, V>(out + wo * (int64_t)rows * C + e0, o);
}

// ------------------------------------------------------------------------------------------------
// BatchNorm affine + dropout + MaxPool2D(2, 2) + band stacking of one block boundary in ONE pass over z: what vm_bn_drop_pool_fwd ->
// vm_pool_windows_fwd -> vm_stack_windows do in three (z read once instead of z, q, the pooled tensor three times; the pooled tensor
// itself is never written).  Same arithmetic and rounding points, bit-identical outputs:
//   q[(b, m)][1 + t'][c]  = max_j drop * (scale * z[(b, m)][2 t' + j][c] + shift)            (kept: the backward routes through it)
//   p                     = q[(b, 2 mo)] >= q[(b, 2 mo + 1)] ? the first : the second          (of the stored values)
//   xs[(b, mo + 1 - dm)][1 + t'][dm * C + c] = p  for dm = 0, 1, 2 inside the clip
// The halo rows of q and xs, the out-of-clip band slots and the channels [3 C, Cs) of xs are never written: the caller zeroes both
// tensors once.  A workgroup owns one band pair of one clip: threads = (channel vector, row lane), rows walked with stride 256 / CV.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_pool2d_stack_fwd_kernel(const T* __restrict__ z, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, const float* __restrict__ drop,
                                                                  int64_t wpt, int M, int L, int C, int Cs, T* __restrict__ q,
                                                                  T* __restrict__ xs, const T* __restrict__ z_lo) {
    // z_lo (optional, vm_bn_pool2d_stack_fwd_split): z comes as two planes of the storage type, the affine sees their fp32 sum
    constexpr int V = Elem<T>::kVec;
    const int CV = C / V, RP = 256 / CV;
    const int cv = threadIdx.x % CV, rl = threadIdx.x / CV;
    if (rl >= RP) return;
    const int Mo = M / 2, Mh = (M + 1) / 2;
    const int64_t b = blockIdx.x / Mh;
    const int mo = (int)(blockIdx.x - b * Mh);
    const int64_t n0 = b * M + 2 * mo;
    const bool pair = 2 * mo + 1 < M;   // (odd band count: the last band has no partner and is dropped by the floor pooling)
    const int64_t tw = n0 / wpt;
    const int c0 = cv * V, Lq = L / 2;
    float sc[V], sh[V], d0[V], d1[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        sc[i] = scale[tw * C + c0 + i];
        sh[i] = shift[tw * C + c0 + i];
        d0[i] = drop ? drop[n0 * C + c0 + i] : 1.f;
        d1[i] = (drop && pair) ? drop[(n0 + 1) * C + c0 + i] : 1.f;
    }
    const T* z0 = z + n0 * (int64_t)L * C + c0;
    const T* z1 = z0 + (int64_t)L * C;
    T* q0 = q + (n0 * (Lq + 2) + 1) * (int64_t)C + c0;
    T* q1 = q0 + (int64_t)(Lq + 2) * C;
    const int64_t xrow = (int64_t)(Lq + 2) * Cs;
    T* x1 = xs + ((b * Mo + mo) * (int64_t)(Lq + 2) + 1) * Cs + c0;
    for (int t = rl; t < Lq; t += RP) {
        const Vec16<T> a0 = load16<T>(z0 + (int64_t)(2 * t) * C), a1 = load16<T>(z0 + (int64_t)(2 * t + 1) * C);
        Vec16<T> o0, o1;
        if (z_lo != nullptr) {
            const T* l0 = z_lo + (z0 - z);
            const Vec16<T> e0 = load16<T>(l0 + (int64_t)(2 * t) * C), e1 = load16<T>(l0 + (int64_t)(2 * t + 1) * C);
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const float y0 = fmaf(a0.get(i) + e0.get(i), sc[i], sh[i]) * d0[i], y1 = fmaf(a1.get(i) + e1.get(i), sc[i], sh[i]) * d0[i];
                o0.set(i, y1 > y0 ? y1 : y0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const float y0 = fmaf(a0.get(i), sc[i], sh[i]) * d0[i], y1 = fmaf(a1.get(i), sc[i], sh[i]) * d0[i];
                o0.set(i, y1 > y0 ? y1 : y0);
            }
        }
        store16<T>(q0 + (int64_t)t * C, o0);
        if (!pair) continue;
        const Vec16<T> b0 = load16<T>(z1 + (int64_t)(2 * t) * C), b1 = load16<T>(z1 + (int64_t)(2 * t + 1) * C);
        if (z_lo != nullptr) {
            const T* l1 = z_lo + (z1 - z);
            const Vec16<T> e0 = load16<T>(l1 + (int64_t)(2 * t) * C), e1 = load16<T>(l1 + (int64_t)(2 * t + 1) * C);
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const float y0 = fmaf(b0.get(i) + e0.get(i), sc[i], sh[i]) * d1[i], y1 = fmaf(b1.get(i) + e1.get(i), sc[i], sh[i]) * d1[i];
                o1.set(i, y1 > y0 ? y1 : y0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const float y0 = fmaf(b0.get(i), sc[i], sh[i]) * d1[i], y1 = fmaf(b1.get(i), sc[i], sh[i]) * d1[i];
                o1.set(i, y1 > y0 ? y1 : y0);
            }
        }
        store16<T>(q1 + (int64_t)t * C, o1);
        Vec16<T> pm;
#pragma unroll
        for (int i = 0; i < V; ++i) pm.set(i, o0.get(i) >= o1.get(i) ? o0.get(i) : o1.get(i));
        T* xr = x1 + (int64_t)t * Cs;
        store16<T>(xr + C, pm);                                   // this band: the middle slot of its own window
        if (mo + 1 < Mo) store16<T>(xr + xrow, pm);               // band mo + 1 sees it as its lower neighbour (dm = 0)
        if (mo >= 1) store16<T>(xr - xrow + 2 * C, pm);           // band mo - 1 as its upper neighbour (dm = 2)
    }
}

// vm_fold_windows + vm_pool_windows_bwd in one pass: the gradient of the pooled block input, summed over the three stacked slots it was
// copied to, goes to the band of the pair that held the maximum (first maximum wins), zero to the other.  dxs: the dgrad output
// (n_clips * Mo, L (+ 2), Cs); q: (n_clips * M, L + 2, C) as in the forward; dq: (n_clips * M, L, C).  Same rounding as the two passes.
template <typename T, int V>
__global__ __launch_bounds__(256) void fold_pool_windows_bwd_kernel(const T* __restrict__ dxs, const T* __restrict__ q, int64_t n_clips, int M,
                                                                    int L, int C, int Cs, int src_pad, T* __restrict__ dq,
                                                                    float* __restrict__ s0, float* __restrict__ sa) {
    // A workgroup owns one band pair of one clip; threads = (channel vector, row lane), rows walked with stride 256 / per_row.
    // s0 / sa (optional): per window the sums of dq and dq * q -- the two sums the BatchNorm backward of the block below needs
    // (vm_bn_bwd_from_sums_finalize, one row per window, a_is_act = 1), taken here from registers instead of by a pass over (z, dq)
    __shared__ float red[4][256][V];
    const int Mo = M / 2, Mh = (M + 1) / 2;
    const int per_row = C / V, RP = 256 / per_row;
    const int tid = threadIdx.x;
    const int cvi = tid % per_row, rl = tid / per_row, c = cvi * V;
    const int64_t b = blockIdx.x / Mh;
    const int mo = (int)(blockIdx.x - b * Mh);
    const int64_t n0 = b * M + 2 * mo;
    const bool pair = 2 * mo + 1 < M;   // else: the dropped last band of an odd count
    float sum[4][V];
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int e = 0; e < V; ++e) sum[w][e] = 0.f;
    const int64_t srow = (int64_t)(L + 2 * src_pad) * Cs;
    const T* src = dxs + ((b * Mo + mo) * (int64_t)(L + 2 * src_pad) + src_pad) * Cs + c;   // this band's own window, slot dm = 1 at + C
    const T* qa = q + (n0 * (L + 2) + 1) * (int64_t)C + c;
    T* d0 = dq + n0 * (int64_t)L * C + c;
    for (int t = rl < RP ? rl : L; t < L; t += RP) {
        EwVec<T, V> o0, o1;
#pragma unroll
        for (int e = 0; e < V; ++e) o0.v[e] = o1.v[e] = Elem<T>::from_f(0.f);
        if (pair) {
            float acc[V];
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] = 0.f;
#pragma unroll
            for (int dm = 0; dm < 3; ++dm) {
                const int md = mo - dm + 1;  // the window whose stacked band dm is this pooled band
                if (md >= 0 && md < Mo) {
                    const EwVec<T, V> sv = ew_load<T, V>(src + (1 - dm) * srow + (int64_t)t * Cs + dm * C);
#pragma unroll
                    for (int e = 0; e < V; ++e) acc[e] += Elem<T>::to_f(sv.v[e]);
                }
            }
            const EwVec<T, V> a0 = ew_load<T, V>(qa + (int64_t)t * C), a1 = ew_load<T, V>(qa + (int64_t)(L + 2 + t) * C);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const T g = Elem<T>::from_f(acc[e]);
                const float gf = Elem<T>::to_f(g), q0 = Elem<T>::to_f(a0.v[e]), q1 = Elem<T>::to_f(a1.v[e]);
                if (q0 >= q1) {
                    o0.v[e] = g;
                    sum[0][e] += gf;
                    sum[1][e] = fmaf(gf, q0, sum[1][e]);
                } else {
                    o1.v[e] = g;
                    sum[2][e] += gf;
                    sum[3][e] = fmaf(gf, q1, sum[3][e]);
                }
            }
            ew_store<T, V>(d0 + (int64_t)(L + t) * C, o1);
        }
        ew_store<T, V>(d0 + (int64_t)t * C, o0);
    }
    if (s0 == nullptr) return;
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int e = 0; e < V; ++e) red[w][tid][e] = sum[w][e];
    __syncthreads();
    for (int k = tid; k < (pair ? 4 : 2) * C; k += 256) {   // the row lanes of a channel vector, in lane order
        const int which = k / C, cc = k - which * C, cv = cc / V, e = cc - cv * V;
        float tot = 0.f;
        for (int r = 0; r < RP; ++r) tot += red[which][r * per_row + cv][e];
        ((which & 1) ? sa : s0)[(n0 + (which >> 1)) * C + cc] = tot;
    }
}

// q: the forward input (n_clips * M windows, L + 2 rows with halo); dout: (n_clips * (M/2), L, C); dq: (n_clips * M, L, C)
template <typename T, int V>
__global__ __launch_bounds__(256) void pool_windows_bwd_kernel(const T* __restrict__ q, const T* __restrict__ dout, int64_t n_clips, int M,
                                                               int L, int C, T* __restrict__ dq) {
    const int Mo = M / 2;
    const int64_t win = blockIdx.x;
    const int j = blockIdx.y * 256 + threadIdx.x;
    if (j >= L * C / V) return;
    const int64_t e0 = (int64_t)j * V;
    const int64_t b = win / M;
    const int m = (int)(win - b * M);
    T* dst = dq + win * (int64_t)L * C + e0;
    EwVec<T, V> o;
#pragma unroll
    for (int e = 0; e < V; ++e) o.v[e] = Elem<T>::from_f(0.f);
    if (m < 2 * Mo) {  // (odd band count: the last band is dropped by the floor pooling and gets no gradient)
        const int64_t pair = b * M + (m & ~1);
        const T* q0 = q + (pair * (L + 2) + 1) * (int64_t)C + e0;  // skip the halo row
        const EwVec<T, V> a0 = ew_load<T, V>(q0), a1 = ew_load<T, V>(q0 + (int64_t)(L + 2) * C);
        const EwVec<T, V> g = ew_load<T, V>(dout + (b * Mo + (m >> 1)) * (int64_t)L * C + e0);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const bool first = Elem<T>::to_f(a0.v[e]) >= Elem<T>::to_f(a1.v[e]);
            if (first == ((m & 1) == 0)) o.v[e] = g.v[e];
        }
    }
    ew_store<T, V>(dst, o);
}

// ------------------------------------------------------------------------------------------------
// First Conv2D(3 x 3) of the variant: ONE input channel.  As a band-stacked k = 3 GEMM it had K = 24 of a 32-wide tile and N = 32 of
// a 128-wide one (0.51 ms + a 0.1 ms stacking pass for 4.9 M positions); it is nine multiply-adds per output and a 312 MB store,
// so it runs on the vector ALUs: a thread owns one position x 8 channels (4 adjacent lanes = the 32..128 channels of a position,
// so a wave's stores are contiguous), a workgroup one 128-position statistics row of one window.
//   z[w][t][co] = relu(bias[co] + sum_kT sum_km W[kT][km][co] * in[(b, m + km - 1)][t + kT])      w = (b, m), bands outside the clip: 0
// The weights are rounded to the storage type first (what the GEMM path multiplies by), products and sums in fp32; the BatchNorm
// partial sums are taken of the stored (rounded) z, one row per 128 positions: the layout of vm_conv_stat_rows.
// W: the fp32 kernel (3, Cs, C) of the flat store, Cs >= 3 (entries km >= 3 are padding).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void conv2d_first_fwd_kernel(const T* __restrict__ in, const float* __restrict__ w,
                                                               const float* __restrict__ bias, int M, int L, int Cs, int C,
                                                               T* __restrict__ z, float* __restrict__ stat_sum, float* __restrict__ stat_sq) {
    __shared__ float red[2][4][128];   // [sum | sq][wave][channel]
    __shared__ __attribute__((aligned(16))) float wl[9][128];
    __shared__ float all[2][256][8];
    const int CV = C / 8;              // channel vectors per position (1 .. 16)
    const int64_t win = blockIdx.x;    // a workgroup walks the statistics rows (128 positions each) of one window
    const int rows = (L + 127) / 128;
    const int m = (int)(win % M);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int PPB = 256 / CV;          // positions per pass of the workgroup
    const int vec = tid % CV, pl = tid / CV;
    for (int i = tid; i < 9 * C; i += 256) {   // the nine taps, rounded to the storage type, once per workgroup
        const int k = i / C, c = i - k * C;
        wl[k][c] = Elem<T>::to_f(Elem<T>::from_f(w[((int64_t)(k / 3) * Cs + (k % 3)) * C + c]));
    }
    __syncthreads();
    // (the 72 weights of a thread stay in LDS -- broadcast reads, 4 addresses per wave -- instead of in registers: 136 -> ~70 VGPRs,
    // twice the waves per SIMD for a kernel that waits on its nine 2-byte loads per position)
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = bias[vec * 8 + e];
    const T* rows_in[3];
#pragma unroll
    for (int km = 0; km < 3; ++km) {
        const int ms = m + km - 1;
        rows_in[km] = (ms >= 0 && ms < M) ? in + (win - m + ms) * (int64_t)(L + 2) : nullptr;
    }
    for (int chunk = 0; chunk < rows; ++chunk) {
        float s1[8], s2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
        const int t_end = pl < PPB ? min(L, (chunk + 1) * 128) : 0;   // (C = 96: 252 of the 256 threads have a position)
        for (int t = chunk * 128 + pl; t < t_end; t += PPB) {
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = bv[e];
#pragma unroll
            for (int km = 0; km < 3; ++km) {
                if (rows_in[km] == nullptr) continue;   // (uniform over the workgroup)
#pragma unroll
                for (int kt = 0; kt < 3; ++kt) {
                    const float x = Elem<T>::to_f(rows_in[km][t + kt]);
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(&wl[kt * 3 + km][vec * 8]);
                    const f32x4 w1 = *reinterpret_cast<const f32x4*>(&wl[kt * 3 + km][vec * 8 + 4]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[e] = fmaf(x, w0[e], acc[e]);
                        acc[4 + e] = fmaf(x, w1[e], acc[4 + e]);
                    }
                }
            }
            EwVec<T, 8> o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o.v[e] = Elem<T>::from_f(fmaxf(acc[e], 0.f));
                const float r = Elem<T>::to_f(o.v[e]);
                s1[e] += r;
                s2[e] = fmaf(r, r, s2[e]);
            }
            ew_store<T, 8>(z + (win * L + t) * (int64_t)C + vec * 8, o);
        }
        if (stat_sum == nullptr) continue;
        const int64_t row = win * rows + chunk;
        // lanes that share a channel vector: the same (tid % CV).  CV divides 64 for C = 8 .. 128 except 96 (CV = 12), whose sums go
        // through LDS thread by thread, in a fixed order
        if (64 % CV == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                for (int o = CV; o < 64; o <<= 1) {
                    s1[e] += __shfl_xor(s1[e], o, 64);
                    s2[e] += __shfl_xor(s2[e], o, 64);
                }
            }
            if (lane < CV) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    red[0][wave][lane * 8 + e] = s1[e];
                    red[1][wave][lane * 8 + e] = s2[e];
                }
            }
            __syncthreads();
            if (tid < C) {
                stat_sum[row * C + tid] = (red[0][0][tid] + red[0][1][tid]) + (red[0][2][tid] + red[0][3][tid]);
                stat_sq[row * C + tid] = (red[1][0][tid] + red[1][1][tid]) + (red[1][2][tid] + red[1][3][tid]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                all[0][tid][e] = s1[e];
                all[1][tid][e] = s2[e];
            }
            __syncthreads();
            if (tid < C) {
                const int v = tid / 8, e = tid % 8;
                float a = 0.f, b = 0.f;
                for (int p = 0; p * CV + v < 256; ++p) {
                    a += all[0][p * CV + v][e];
                    b += all[1][p * CV + v][e];
                }
                stat_sum[row * C + tid] = a;
                stat_sq[row * C + tid] = b;
            }
        }
        __syncthreads();   // red / all are rewritten by the next statistics row
    }
}

// ... and its weight gradient: dW[kT][km][co] = sum in[(b, m + km - 1)][t + kT] * du[(b, m)][t][co] over every window and position
// (du padded like an activation tensor).  A workgroup walks `wpb` consecutive windows with the same thread map as the forward and
// leaves one fp32 slab (9, C); slab_sum adds the slabs in a fixed order and the host scatters the nine taps into the (3, Cs, C) kernel
// gradient.  The layer has no input gradient.
template <typename T>
__global__ __launch_bounds__(256) void conv2d_first_wgrad_kernel(const T* __restrict__ in, const T* __restrict__ du, int64_t n_windows, int M,
                                                                 int L, int C, int wpb, float* __restrict__ slabs) {
    __shared__ float red[4][9][128];
    const int CV = C / 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int PPB = 256 / CV;
    const int vec = tid % CV, pl = tid / CV;
    float acc[9][8];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[k][e] = 0.f;
    const int64_t w0 = (int64_t)blockIdx.x * wpb;
    for (int64_t win = w0; win < w0 + wpb && win < n_windows; ++win) {
        const int m = (int)(win % M);
        for (int t = pl; t < (pl < PPB ? L : 0); t += PPB) {
            const EwVec<T, 8> g = ew_load<T, 8>(du + (win * (L + 2) + 1 + t) * (int64_t)C + vec * 8);
            float gf[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) gf[e] = Elem<T>::to_f(g.v[e]);
#pragma unroll
            for (int km = 0; km < 3; ++km) {
                const int ms = m + km - 1;
                if (ms < 0 || ms >= M) continue;
                const T* r = in + (win - m + ms) * (int64_t)(L + 2) + t;
#pragma unroll
                for (int kt = 0; kt < 3; ++kt) {
                    const float x = Elem<T>::to_f(r[kt]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[kt * 3 + km][e] = fmaf(x, gf[e], acc[kt * 3 + km][e]);
                }
            }
        }
    }
    // sum over the threads that share a channel vector, in a fixed order: lanes (where CV divides 64), then waves
    const bool shfl = 64 % CV == 0;
    __shared__ float all[256][9];   // the C = 96 path: one channel at a time
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        if (shfl) {
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                float v = acc[k][e];
                for (int o = CV; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
                if (lane < CV) red[wave][k][lane * 8 + e] = v;
            }
        } else {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 9; ++k) all[tid][k] = acc[k][e];
            __syncthreads();
            if (tid < CV * 9) {
                const int v = tid / 9, k = tid % 9;
                float a = 0.f;
                for (int p = 0; p * CV + v < 256; ++p) a += all[p * CV + v][k];
                red[0][k][v * 8 + e] = a;
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < 9 * C; i += 256) {
        const int k = i / C, c = i % C;
        slabs[(int64_t)blockIdx.x * 9 * C + i] = shfl ? (red[0][k][c] + red[1][k][c]) + (red[2][k][c] + red[3][k][c]) : red[0][k][c];
    }
}

// ------------------------------------------------------------------------------------------------
// The same layer on the matrix pipe (16-bit storage, C % 32 == 0).  The vector-ALU form above issues ~600 vector instructions per
// 32 positions x 32 channels and kept the vector pipes 77 % and the LDS 68 % busy (SQ counters, profiles/r03_logmel_pmc_*.csv): its
// 0.22 ms were arithmetic, not the 312 MB store.  As a GEMM tile the nine taps are K = 9 of ONE v_mfma_f32_32x32x16:
//   A = the 3 x 3 patches of 32 positions (gathered with 2-byte loads: the input has one channel, 10 MB, cache-resident),
//   B = the nine weights of 32 channels (registers, once per wave), accumulator start = bias.
// A wave owns one statistics row (128 positions of one window = 4 tiles).  D puts a channel in a lane and the positions in its
// registers, so the BatchNorm partial sums are in-lane adds (no cross-lane reduction until the row ends); the tile reaches memory
// through a wave-private LDS transpose: 2-byte writes [position][channel], 16-byte reads, 1 KB contiguous per store instruction.
// Tap order k = kT * 3 + km as above; same rounding points (weights to the storage type, fp32 accumulation, statistics of the stored z).
// ------------------------------------------------------------------------------------------------
template <typename T> struct Mma16;
template <> struct Mma16<f16> {
    using Frag = f16x8;
    __device__ static inline f32x16 run32(Frag a, Frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    __device__ static inline f32x4 run16(Frag a, Frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <> struct Mma16<bf16> {
    using Frag = bf16x8;
    __device__ static inline f32x16 run32(Frag a, Frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    __device__ static inline f32x4 run16(Frag a, Frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

// SPLIT (vm_conv2d_first_fwd_split): the image comes as two planes in + in_lo (vm_stft_logmel_f16s_split) and the weights are split the same
// way inside; the three significant products in_hi w_hi + in_lo w_hi + in_hi w_lo are 27 of the 32 K slots of TWO instructions -- a half-wave's
// slots in order: [hi taps | lo taps] x w_hi, then [hi taps] x w_lo -- so the layer sees image and filters at ~2 x the storage significand.
template <typename T, int CB, bool SPLIT>   // CB = C / 32
__global__ __launch_bounds__(256) void conv2d_first_fwd_mfma_kernel(const T* __restrict__ in, const T* __restrict__ in_lo, const float* __restrict__ w,
                                                                    const float* __restrict__ bias, int64_t n_rows, int M, int L, int Cs,
                                                                    T* __restrict__ z, T* __restrict__ z_lo, float* __restrict__ stat_sum,
                                                                    float* __restrict__ stat_sq) {
    // SPLIT: the statistics are those of relu(conv + bias) to two planes of the storage type (z + what z's rounding dropped) -- what
    // vm_conv2d_first_bn_pool_stack recomputes and what vm_bn_pool2d_stack_fwd_split reads back; z_lo (optional) stores the low plane
    constexpr int C = 32 * CB;
    using Frag = typename Mma16<T>::Frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, kh = lane >> 5;
    const int rows = (L + 127) / 128;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;   // statistics row = (window, 128-position chunk); no workgroup barrier below
    if (row >= n_rows) return;
    const int64_t win = row / rows;
    const int chunk = (int)(row - win * rows);
    const int m = (int)(win % M);
    T* tile = reinterpret_cast<T*>(smem) + wave * 32 * C;   // [32 positions][C]

    // this lane's K slots: the lower half-wave carries taps 0 .. 4 in slots 0 .. 4, the upper one taps 5 .. 8 in slots 0 .. 3 (five
    // gather instructions a tile, not eight); the other slots are zero in both operands
    Frag wb[CB];
    Frag wb2[SPLIT ? CB : 1];
    float bv[CB];
    int off[5];
    unsigned ok = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 5 * kh + e, kt = k / 3, km = k - 3 * kt, ms = m + km - 1;
        const bool slot = e < 5 - kh;
        if (e < 5) {
            off[e] = (km - 1) * (L + 2) + kt;
            if (slot && ms >= 0 && ms < M) ok |= 1u << e;
        }
        if constexpr (!SPLIT) {
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
                wb[cb][e] = Elem<T>::from_f(slot ? w[((int64_t)kt * Cs + km) * C + 32 * cb + col] : 0.f);
        }
    }
    if constexpr (SPLIT) {
        // slot s of the 16 this half-wave owns (8 per instruction): n = 5 - kh taps; s < n: hi x w_hi, s < 2 n: lo x w_hi, s < 3 n: hi x w_lo
        const int n = 5 - kh;
#pragma unroll
        for (int sl = 0; sl < 16; ++sl) {
            const int part = sl / n, e = sl - part * n;       // (runtime: once per wave)
            const int k = 5 * kh + e, kt = k / 3, km = k - 3 * kt;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                T v = Elem<T>::from_f(0.f);
                if (part < 3) {
                    const float wf = w[((int64_t)kt * Cs + km) * C + 32 * cb + col];
                    const T wh = Elem<T>::from_f(wf);
                    v = part < 2 ? wh : Elem<T>::from_f(wf - Elem<T>::to_f(wh));
                }
                if (sl < 8) wb[cb][sl] = v; else wb2[cb][sl - 8] = v;
            }
        }
    }
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) bv[cb] = bias[32 * cb + col];
    const T* base = in + win * (int64_t)(L + 2);
    const T* base_lo = SPLIT ? in_lo + win * (int64_t)(L + 2) : nullptr;
    float s1[CB], s2[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) s1[cb] = s2[cb] = 0.f;

    for (int ti = 0; ti < 4; ++ti) {
        const int t0 = chunk * 128 + ti * 32;
        if (t0 >= L) break;
        const int t = t0 + col;
        Frag a;
        Frag a2;
        T zl[SPLIT ? CB : 1][16];
        if constexpr (!SPLIT) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                T v = Elem<T>::from_f(0.f);
                if (e < 5)
                    if (((ok >> e) & 1u) && t < L) v = base[off[e] + t];
                a[e] = v;
            }
        } else {
            const T zero = Elem<T>::from_f(0.f);
            T hi[5], lo[5];
#pragma unroll
            for (int e = 0; e < 5; ++e) {
                hi[e] = lo[e] = zero;
                if (((ok >> e) & 1u) && t < L) {
                    hi[e] = base[off[e] + t];
                    lo[e] = base_lo[off[e] + t];
                }
            }
            // the 16 slots: kh = 0 (5 taps): hi0-4 lo0-4 hi0-4 0;  kh = 1 (4 taps): hi0-3 lo0-3 hi0-3 0 0 0 0
            const bool up = kh != 0;
            a[0] = hi[0]; a[1] = hi[1]; a[2] = hi[2]; a[3] = hi[3];
            a[4] = up ? lo[0] : hi[4];
            a[5] = up ? lo[1] : lo[0];
            a[6] = up ? lo[2] : lo[1];
            a[7] = up ? lo[3] : lo[2];
            a2[0] = up ? hi[0] : lo[3];
            a2[1] = up ? hi[1] : lo[4];
            a2[2] = up ? hi[2] : hi[0];
            a2[3] = up ? hi[3] : hi[1];
            a2[4] = up ? zero : hi[2];
            a2[5] = up ? zero : hi[3];
            a2[6] = up ? zero : hi[4];
            a2[7] = zero;
        }
        const bool whole = t0 + 32 <= L;   // (uniform) every position of the tile counts in the statistics
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = bv[cb];
            acc = Mma16<T>::run32(a, wb[cb], acc);
            if constexpr (SPLIT) acc = Mma16<T>::run32(a2, wb2[cb], acc);
            float rf[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pl = (r & 3) + 8 * (r >> 2) + 4 * kh;
                const float v = fmaxf(acc[r], 0.f);
                const T h = Elem<T>::from_f(v);
                rf[r] = Elem<T>::to_f(h);
                tile[pl * C + 32 * cb + col] = h;
                if constexpr (SPLIT) {
                    zl[cb][r] = Elem<T>::from_f(v - rf[r]);
                    rf[r] += Elem<T>::to_f(zl[cb][r]);   // the statistics of the two-plane value, whether or not its low plane is stored
                }
            }
            if (!whole) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (t0 + (r & 3) + 8 * (r >> 2) + 4 * kh >= L) rf[r] = 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s1[cb] += rf[r];
                s2[cb] = fmaf(rf[r], rf[r], s2[cb]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (wave-private tile: the LDS serves a wave's accesses in order)
        const int nbytes = (L - t0 < 32 ? L - t0 : 32) * C * 2;
        char* dst = reinterpret_cast<char*>(z + (win * L + t0) * (int64_t)C);
#pragma unroll
        for (int i = 0; i < 2 * CB; ++i) {
            const int bo = (i * 64 + lane) * 16;
            if (bo < nbytes) *reinterpret_cast<u32x4*>(dst + bo) = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(tile) + bo);
        }
        asm volatile("" ::: "memory");
        if constexpr (SPLIT) {
            if (z_lo != nullptr) {   // the low plane through the same wave-private tile
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) tile[((r & 3) + 8 * (r >> 2) + 4 * kh) * C + 32 * cb + col] = zl[cb][r];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                char* dl = reinterpret_cast<char*>(z_lo + (win * L + t0) * (int64_t)C);
#pragma unroll
                for (int i = 0; i < 2 * CB; ++i) {
                    const int bo = (i * 64 + lane) * 16;
                    if (bo < nbytes) *reinterpret_cast<u32x4*>(dl + bo) = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(tile) + bo);
                }
                asm volatile("" ::: "memory");
            }
        }
    }
    if (stat_sum != nullptr) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const float a = s1[cb] + __shfl_xor(s1[cb], 32, 64), b = s2[cb] + __shfl_xor(s2[cb], 32, 64);
            if (kh == 0) {
                stat_sum[row * C + 32 * cb + col] = a;
                stat_sq[row * C + 32 * cb + col] = b;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Block 1's boundary WITHOUT the round trip of z (vm_conv2d_first_bn_pool_stack, round 6): once the statistics are known (they come from
// vm_conv2d_first_fwd_split, which also leaves z for the backward), the convolution is nine taps of a 10 MB image -- cheaper to redo than
// to read back -- so this kernel recomputes relu(conv + bias) on the two-plane image (the SPLIT operands above, two matrix instructions
// per band and 32 x 32 tile), applies the BatchNorm affine (+ dropout) to the fp32 accumulator, pools 2 x 2 and writes what
// vm_bn_pool2d_stack_fwd writes: q (per band, pooled along T, rounded to the storage type: the backward routes through it) and the
// band-stacked block-2 input xs = the larger of the two bands' STORED q.  z is neither read nor rounded on the way: the rounding of
// block 1's conv output (5.0e-4 of config 4's embedding error in f16) is gone and so are 312 MB of reads per 256 clips.
// D puts a channel in a lane and 16 positions in its registers (r -> position (r & 3) + 8 (r >> 2) + 4 kh): the pool pair (2 t', 2 t' + 1)
// is registers (r, r + 1) of one lane, the band pair two accumulators of the same lane -- no cross-lane traffic; q0 / q1 / xs tiles
// (16 pooled positions x C) leave through a wave-private LDS transpose as 16-byte stores.
// A wave owns 128 positions (64 pooled) of one band pair of one clip.
// ------------------------------------------------------------------------------------------------
template <typename T, int CB>
__global__ __launch_bounds__(256) void conv2d_first_bn_pool_stack_kernel(const T* __restrict__ in, const T* __restrict__ in_lo,
                                                                         const float* __restrict__ w, const float* __restrict__ bias,
                                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                                         const float* __restrict__ drop, int64_t n_items, int64_t wpt, int M,
                                                                         int L, int Cs, int Cs2, T* __restrict__ q, T* __restrict__ xs) {
    constexpr int C = 32 * CB;
    using Frag = typename Mma16<T>::Frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, kh = lane >> 5;
    const int chunks = (L + 127) / 128;
    const int Mo = M / 2, Mh = (M + 1) / 2, Lq = L / 2;
    const int64_t item = (int64_t)blockIdx.x * 4 + wave;   // (clip, band pair, 128-position chunk); no workgroup barrier below
    if (item >= n_items) return;
    const int chunk = (int)(item % chunks);
    const int64_t bp = item / chunks;
    const int64_t b = bp / Mh;
    const int mo = (int)(bp - b * Mh);
    const int64_t n0 = b * M + 2 * mo;                      // the pair's first band as a window
    const bool pair = 2 * mo + 1 < M;                       // (odd band count: the last band has no partner, the floor pooling drops it)
    const int64_t tw = n0 / wpt;
    T* t0s = reinterpret_cast<T*>(smem) + wave * 3 * 16 * C;   // [q0 | q1 | xs][16 pooled positions][C]
    T* t1s = t0s + 16 * C;
    T* tps = t1s + 16 * C;

    Frag wb[CB], wb2[CB];
    {
        const int n = 5 - kh;
#pragma unroll
        for (int sl = 0; sl < 16; ++sl) {
            const int part = sl / n, e = sl - part * n;
            const int k = 5 * kh + e, kt = k / 3, km = k - 3 * kt;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                T v = Elem<T>::from_f(0.f);
                if (part < 3) {
                    const float wf = w[((int64_t)kt * Cs + km) * C + 32 * cb + col];
                    const T wh = Elem<T>::from_f(wf);
                    v = part < 2 ? wh : Elem<T>::from_f(wf - Elem<T>::to_f(wh));
                }
                if (sl < 8) wb[cb][sl] = v; else wb2[cb][sl - 8] = v;
            }
        }
    }
    int off[5];
    unsigned ok0 = 0, ok1 = 0;   // which taps of band 2 mo / 2 mo + 1 lie inside the clip
#pragma unroll
    for (int e = 0; e < 5; ++e) {
        const int k = 5 * kh + e, kt = k / 3, km = k - 3 * kt;
        off[e] = (km - 1) * (L + 2) + kt;
        if (e < 5 - kh) {
            const int ms0 = 2 * mo + km - 1, ms1 = ms0 + 1;
            if (ms0 >= 0 && ms0 < M) ok0 |= 1u << e;
            if (pair && ms1 >= 0 && ms1 < M) ok1 |= 1u << e;
        }
    }
    float bv[CB], sc[CB], sh[CB], d0[CB], d1[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        const int c = 32 * cb + col;
        bv[cb] = bias[c];
        sc[cb] = scale[tw * C + c];
        sh[cb] = shift[tw * C + c];
        d0[cb] = drop ? drop[n0 * C + c] : 1.f;
        d1[cb] = (drop && pair) ? drop[(n0 + 1) * C + c] : 1.f;
    }
    const T* base_h = in + n0 * (int64_t)(L + 2);
    const T* base_l = in_lo + n0 * (int64_t)(L + 2);
    const T zero = Elem<T>::from_f(0.f);
    const bool up = kh != 0;
    auto gather = [&](const T* bh, const T* bl, unsigned ok, int t, Frag& a, Frag& a2) {
        T hi[5], lo[5];
#pragma unroll
        for (int e = 0; e < 5; ++e) {
            hi[e] = lo[e] = zero;
            if (((ok >> e) & 1u) && t < L) {
                hi[e] = bh[off[e] + t];
                lo[e] = bl[off[e] + t];
            }
        }
        a[0] = hi[0]; a[1] = hi[1]; a[2] = hi[2]; a[3] = hi[3];
        a[4] = up ? lo[0] : hi[4];
        a[5] = up ? lo[1] : lo[0];
        a[6] = up ? lo[2] : lo[1];
        a[7] = up ? lo[3] : lo[2];
        a2[0] = up ? hi[0] : lo[3];
        a2[1] = up ? hi[1] : lo[4];
        a2[2] = up ? hi[2] : hi[0];
        a2[3] = up ? hi[3] : hi[1];
        a2[4] = up ? zero : hi[2];
        a2[5] = up ? zero : hi[3];
        a2[6] = up ? zero : hi[4];
        a2[7] = zero;
    };
    T* q0 = q + (n0 * (Lq + 2) + 1) * (int64_t)C;
    T* q1 = q0 + (int64_t)(Lq + 2) * C;
    const int64_t xrow = (int64_t)(Lq + 2) * Cs2;
    T* x1 = xs + ((b * Mo + mo) * (int64_t)(Lq + 2) + 1) * Cs2;

    for (int ti = 0; ti < 4; ++ti) {
        const int tt0 = chunk * 128 + ti * 32;    // first position of the tile; its pooled positions start at tt0 / 2
        const int p0 = tt0 / 2;
        if (p0 >= Lq) break;
        const int t = tt0 + col;
        Frag a, a2, c1, c2;
        gather(base_h, base_l, ok0, t, a, a2);
        if (pair) gather(base_h + (L + 2), base_l + (L + 2), ok1, t, c1, c2);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = bv[cb];
            acc0 = Mma16<T>::run32(a, wb[cb], acc0);
            acc0 = Mma16<T>::run32(a2, wb2[cb], acc0);
            if (pair) {
                acc1 = Mma16<T>::run32(c1, wb[cb], acc1);
                acc1 = Mma16<T>::run32(c2, wb2[cb], acc1);
            }
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int pp = ((r & 3) >> 1) + 4 * (r >> 2) + 2 * kh;   // pooled position of registers (r, r + 1) inside the tile
                const float ya = fmaf(fmaxf(acc0[r], 0.f), sc[cb], sh[cb]) * d0[cb], yb = fmaf(fmaxf(acc0[r + 1], 0.f), sc[cb], sh[cb]) * d0[cb];
                const T h0 = Elem<T>::from_f(yb > ya ? yb : ya);
                t0s[pp * C + 32 * cb + col] = h0;
                if (pair) {
                    const float yc = fmaf(fmaxf(acc1[r], 0.f), sc[cb], sh[cb]) * d1[cb], yd = fmaf(fmaxf(acc1[r + 1], 0.f), sc[cb], sh[cb]) * d1[cb];
                    const T h1 = Elem<T>::from_f(yd > yc ? yd : yc);
                    t1s[pp * C + 32 * cb + col] = h1;
                    tps[pp * C + 32 * cb + col] = Elem<T>::to_f(h0) >= Elem<T>::to_f(h1) ? h0 : h1;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (wave-private tiles: the LDS serves a wave's accesses in order)
        const int np = Lq - p0 < 16 ? Lq - p0 : 16;          // pooled positions of this tile that exist
        constexpr int VPP = C / 8;                             // 16-byte vectors per position
#pragma unroll
        for (int i = 0; i < (16 * VPP + 63) / 64; ++i) {
            const int idx = i * 64 + lane, pp = idx / VPP, v = idx - pp * VPP;
            if (pp < np) {
                const u32x4 v0 = *reinterpret_cast<const u32x4*>(t0s + pp * C + v * 8);
                *reinterpret_cast<u32x4*>(q0 + (int64_t)(p0 + pp) * C + v * 8) = v0;
                if (pair) {
                    const u32x4 v1 = *reinterpret_cast<const u32x4*>(t1s + pp * C + v * 8);
                    const u32x4 vp = *reinterpret_cast<const u32x4*>(tps + pp * C + v * 8);
                    *reinterpret_cast<u32x4*>(q1 + (int64_t)(p0 + pp) * C + v * 8) = v1;
                    T* xr = x1 + (int64_t)(p0 + pp) * Cs2 + v * 8;
                    *reinterpret_cast<u32x4*>(xr + C) = vp;                                  // this band: the middle slot of its own window
                    if (mo + 1 < Mo) *reinterpret_cast<u32x4*>(xr + xrow) = vp;              // band mo + 1 sees it as its lower neighbour
                    if (mo >= 1) *reinterpret_cast<u32x4*>(xr - xrow + 2 * C) = vp;          // band mo - 1 as its upper neighbour
                }
            }
        }
        asm volatile("" ::: "memory");
    }
}

// ... and the weight gradient as v_mfma_f32_16x16x32 tiles: dW[tap][co] = sum over positions of patch[position][tap] * du[position][co] is a
// GEMM with the POSITIONS as K.  A (16 tap rows x 32 positions): a lane's eight K slots are eight consecutive samples of one band --
// one 16-byte load; B (32 positions x 16 channels): eight 2-byte loads a channel block (du is channel-contiguous).  A wave walks
// (window, 32-position block) items of its workgroup's `wpb` windows; the four waves' tiles meet in LDS in a fixed order.  Same
// slab layout as above.  Positions past the window: du is read as zero there (the A samples beside them belong to the next band: finite).
template <typename T, int CB16>   // CB16 = C / 16
__global__ __launch_bounds__(256) void conv2d_first_wgrad_mfma_kernel(const T* __restrict__ in, const T* __restrict__ du, int64_t n_windows,
                                                                      int M, int L, int wpb, float* __restrict__ slabs) {
    constexpr int C = 16 * CB16;
    using Frag = typename Mma16<T>::Frag;
    __shared__ float red[4][9][C];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kg = lane >> 4;
    const int kt = i / 3, km = i - 3 * kt;   // tap i = kT * 3 + km (rows 9 .. 15 of A: zero)
    f32x4 acc[CB16];
#pragma unroll
    for (int cb = 0; cb < CB16; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nblk = (L + 31) / 32;
    const int64_t w0 = (int64_t)blockIdx.x * wpb;
    const int nwin = (int)(n_windows - w0 < wpb ? n_windows - w0 : wpb);
    for (int item = wave; item < nwin * nblk; item += 4) {
        const int wi = item / nblk, blk = item - wi * nblk;
        const int64_t win = w0 + wi;
        const int m = (int)(win % M), ms = m + km - 1;
        const int p0 = blk * 32 + kg * 8;   // this lane's K slots: positions p0 .. p0 + 7
        Frag a;
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = Elem<T>::from_f(0.f);
        if (i < 9 && ms >= 0 && ms < M) {
            const T* src = in + (win + km - 1) * (int64_t)(L + 2) + p0 + kt;
            if (p0 + kt + 8 <= L + 2) {
                __builtin_memcpy(&a, src, 16);   // 2-byte aligned
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (p0 + kt + j < L + 2) a[j] = src[j];
            }
        }
        // channel blocks in pairs: lane i carries channels 32 q + 2 i (block 2 q) and 32 q + 2 i + 1 (block 2 q + 1) -- one 4-byte load
        const T* g = du + ((win * (L + 2) + 1 + p0) * (int64_t)C + 2 * i);
#pragma unroll
        for (int q = 0; q < CB16 / 2; ++q) {
            Frag b0, b1;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                uint32_t v = 0;
                if (p0 + j < L) v = *reinterpret_cast<const uint32_t*>(g + (int64_t)j * C + 32 * q);
                const uint16_t lo = (uint16_t)(v & 0xffffu), hi = (uint16_t)(v >> 16);
                b0[j] = __builtin_bit_cast(T, lo);
                b1[j] = __builtin_bit_cast(T, hi);
            }
            acc[2 * q] = Mma16<T>::run16(a, b0, acc[2 * q]);
            acc[2 * q + 1] = Mma16<T>::run16(a, b1, acc[2 * q + 1]);
        }
    }
#pragma unroll
    for (int cb = 0; cb < CB16; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int tap = 4 * kg + r;
            if (tap < 9) red[wave][tap][32 * (cb >> 1) + 2 * i + (cb & 1)] = acc[cb][r];
        }
    __syncthreads();
    for (int q = tid; q < 9 * C; q += 256) {
        const int k = q / C, c = q % C;
        slabs[(int64_t)blockIdx.x * 9 * C + q] = (red[0][k][c] + red[1][k][c]) + (red[2][k][c] + red[3][k][c]);
    }
}

template <typename T>
static void launch_c2f_mfma(const T* in, const T* in_lo, const float* w, const float* bias, int64_t n_rows, int M, int L, int Cs, int C, T* z,
                            T* z_lo, float* stat_sum, float* stat_sq, hipStream_t stream) {
    const dim3 grid((unsigned)((n_rows + 3) / 4));
    const size_t lds = (size_t)4 * 32 * C * 2;
#define VM_LAUNCH_C2F(CB)                                                                                                                   \
    do {                                                                                                                                    \
        if (in_lo != nullptr)                                                                                                               \
            hipLaunchKernelGGL((conv2d_first_fwd_mfma_kernel<T, CB, true>), grid, dim3(256), lds, stream, in, in_lo, w, bias, n_rows, M, L, Cs, z, \
                               z_lo, stat_sum, stat_sq);                                                                                         \
        else                                                                                                                                \
            hipLaunchKernelGGL((conv2d_first_fwd_mfma_kernel<T, CB, false>), grid, dim3(256), lds, stream, in, in_lo, w, bias, n_rows, M, L, Cs, z, \
                               z_lo, stat_sum, stat_sq);                                                                                         \
    } while (0)
    switch (C / 32) {
        case 1: VM_LAUNCH_C2F(1); break;
        case 2: VM_LAUNCH_C2F(2); break;
        case 3: VM_LAUNCH_C2F(3); break;
        default: VM_LAUNCH_C2F(4); break;
    }
#undef VM_LAUNCH_C2F
}

template <typename T>
static void launch_c2w_mfma(const T* in, const T* du, int64_t nw, int M, int L, int C, int wpb, float* slabs, unsigned blocks, hipStream_t stream) {
#define VM_LAUNCH_C2W(CB16) \
    hipLaunchKernelGGL((conv2d_first_wgrad_mfma_kernel<T, CB16>), dim3(blocks), dim3(256), 0, stream, in, du, nw, M, L, wpb, slabs)
    switch (C / 32) {
        case 1: VM_LAUNCH_C2W(2); break;
        case 2: VM_LAUNCH_C2W(4); break;
        case 3: VM_LAUNCH_C2W(6); break;
        default: VM_LAUNCH_C2W(8); break;
    }
#undef VM_LAUNCH_C2W
}

__global__ void conv2d_first_scatter_kernel(const float* __restrict__ g9, int Cs, int C, float* __restrict__ grad_w) {
    const int i = blockIdx.x * 256 + threadIdx.x;   // over (3, Cs, C)
    if (i >= 3 * Cs * C) return;
    const int c = i % C, km = (i / C) % Cs, kt = i / (C * Cs);
    grad_w[i] = km < 3 ? g9[(kt * 3 + km) * C + c] : 0.f;
}

__global__ __launch_bounds__(256) void clip_max_fwd_kernel(const float* __restrict__ gmax, int64_t n_clips, int M, int Mv, int C,
                                                           float* __restrict__ out, int32_t* __restrict__ widx) {
    const int64_t i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= n_clips * C) return;
    const int64_t b = i / C;
    const int c = (int)(i % C);
    float best = gmax[(b * M) * (int64_t)C + c];
    int bi = 0;
    for (int m = 1; m < Mv; ++m) {
        const float v = gmax[(b * M + m) * (int64_t)C + c];
        if (v > best) {  // first maximum wins
            best = v;
            bi = m;
        }
    }
    out[i] = best;
    widx[i] = bi;
}

__global__ __launch_bounds__(256) void clip_max_bwd_kernel(const float* __restrict__ dout, const int32_t* __restrict__ widx,
                                                           int64_t n_clips, int M, int C, float* __restrict__ dg) {
    const int64_t i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= n_clips * M * C) return;
    const int c = (int)(i % C);
    const int64_t win = i / C;
    const int64_t b = win / M;
    const int m = (int)(win % M);
    dg[i] = widx[b * C + c] == m ? dout[b * C + c] : 0.f;
}

static dim3 ew_grid(int64_t windows, int64_t vectors_per_window) { return dim3((unsigned)windows, (unsigned)((vectors_per_window + 255) / 256)); }

}  // namespace vm

using namespace vm;

extern "C" int64_t vm_stft_frames(int64_t raw_len, int win_length, int hop) {
    if (raw_len < win_length || hop <= 0) return 0;
    return 1 + (raw_len - win_length) / hop;
}

extern "C" int vm_stft_logmel(const void* raw, int is_int16, int64_t n_clips, int64_t raw_len, int win_length, int hop,
                              const float* basis, const float* melw, int n_mels, float log_floor, int dtype, void* out, void* stream) {
    VM_REQUIRE(raw && basis && melw && out, "vm_stft_logmel: null pointer");
    VM_REQUIRE(n_clips > 0 && n_clips < 65536 && win_length >= 2 && win_length <= 512 && hop > 0 && raw_len >= win_length,
               "vm_stft_logmel: bad sizes (n_fft is 512: win_length <= 512)");
    VM_REQUIRE(n_mels == 32 || n_mels == 64 || n_mels == 96 || n_mels == 128, "vm_stft_logmel: n_mels must be 32, 64, 96 or 128");
    VM_REQUIRE(log_floor > 0.f, "vm_stft_logmel: log_floor must be positive");
    const int n_frames = (int)vm_stft_frames(raw_len, win_length, hop);
    const dim3 grid((unsigned)((n_frames + SF_FRAMES - 1) / SF_FRAMES), (unsigned)n_clips);
    size_t lds = (size_t)SF_FRAMES * (win_length + 1) * 4;
    const size_t red = (size_t)4 * n_mels * 32 * 4;
    if (red > lds) lds = red;
#define VM_LAUNCH_SF(TT, NMB)                                                                                                    \
    hipLaunchKernelGGL((stft_logmel_kernel<TT, NMB>), grid, dim3(256), lds, (hipStream_t)stream, raw, is_int16, raw_len, win_length, \
                       hop, n_frames, basis, melw, log_floor, (TT*)out)
    VM_DISPATCH_DTYPE(dtype, {
        switch (n_mels / 32) {
            case 1: VM_LAUNCH_SF(T, 1); break;
            case 2: VM_LAUNCH_SF(T, 2); break;
            case 3: VM_LAUNCH_SF(T, 3); break;
            default: VM_LAUNCH_SF(T, 4); break;
        }
    });
#undef VM_LAUNCH_SF
    return check_launch("vm_stft_logmel");
}

extern "C" int64_t vm_stft_split_basis_bytes(int win_length) { return 2LL * 512 * ((win_length + 15) / 16 * 16) * 2; }

extern "C" int vm_stft_split_basis(const float* basis, int win_length, void* basis16, void* stream) {
    VM_REQUIRE(basis && basis16 && win_length >= 2 && win_length <= 512, "vm_stft_split_basis: bad argument");
    const int Kp = (win_length + 15) / 16 * 16;
    hipLaunchKernelGGL(stft_split_basis_kernel, dim3((unsigned)((512 * Kp + 255) / 256)), dim3(256), 0, (hipStream_t)stream, basis, win_length, Kp,
                       (f16*)basis16);
    return check_launch("vm_stft_split_basis");
}

static int stft_logmel_f16s(const char* who, const void* raw, int is_int16, int64_t n_clips, int64_t raw_len, int win_length, int hop,
                            const void* basis16, const float* melw, int n_mels, float log_floor, int dtype, void* out, void* out_lo, void* stream) {
    VM_REQUIRE(raw && basis16 && melw && out, "vm_stft_logmel_f16s: null pointer");
    VM_REQUIRE(n_clips > 0 && n_clips < 65536 && win_length >= 2 && win_length <= 512 && hop > 0 && raw_len >= win_length,
               "vm_stft_logmel_f16s: bad sizes (n_fft is 512: win_length <= 512)");
    VM_REQUIRE(n_mels == 32 || n_mels == 64 || n_mels == 96 || n_mels == 128, "vm_stft_logmel_f16s: n_mels must be 32, 64, 96 or 128");
    VM_REQUIRE(log_floor > 0.f, "vm_stft_logmel_f16s: log_floor must be positive");
    const int n_frames = (int)vm_stft_frames(raw_len, win_length, hop);
    const int Kp = (win_length + 15) / 16 * 16;
    const dim3 grid((unsigned)((n_frames + SF_FRAMES - 1) / SF_FRAMES), (unsigned)n_clips);
    size_t lds = (size_t)2 * SF_FRAMES * (Kp + 8) * 2;
    const size_t red = (size_t)4 * n_mels * 32 * 4;
    if (red > lds) lds = red;
#define VM_LAUNCH_SF16(TT, NMB)                                                                                                       \
    hipLaunchKernelGGL((stft_logmel_f16s_kernel<TT, NMB>), grid, dim3(256), lds, (hipStream_t)stream, raw, is_int16, raw_len, win_length, \
                       hop, n_frames, (const f16*)basis16, Kp, melw, log_floor, (TT*)out, (TT*)out_lo)
    VM_DISPATCH_DTYPE(dtype, {
        switch (n_mels / 32) {
            case 1: VM_LAUNCH_SF16(T, 1); break;
            case 2: VM_LAUNCH_SF16(T, 2); break;
            case 3: VM_LAUNCH_SF16(T, 3); break;
            default: VM_LAUNCH_SF16(T, 4); break;
        }
    });
#undef VM_LAUNCH_SF16
    return check_launch(who);
}

extern "C" int vm_stft_logmel_f16s(const void* raw, int is_int16, int64_t n_clips, int64_t raw_len, int win_length, int hop,
                                   const void* basis16, const float* melw, int n_mels, float log_floor, int dtype, void* out, void* stream) {
    return stft_logmel_f16s("vm_stft_logmel_f16s", raw, is_int16, n_clips, raw_len, win_length, hop, basis16, melw, n_mels, log_floor, dtype, out,
                            nullptr, stream);
}

// ... and the part of the image the storage type drops as a second plane of the same layout (out_lo = image - out, rounded to `dtype`):
// the first convolution multiplies both (vm_conv2d_first_fwd_split), so the log-mel image enters the network at 2 x the significand
extern "C" int vm_stft_logmel_f16s_split(const void* raw, int is_int16, int64_t n_clips, int64_t raw_len, int win_length, int hop,
                                         const void* basis16, const float* melw, int n_mels, float log_floor, int dtype, void* out,
                                         void* out_lo, void* stream) {
    VM_REQUIRE(out_lo, "vm_stft_logmel_f16s_split: null pointer");
    VM_REQUIRE(dtype == VM_BF16 || dtype == VM_F16, "vm_stft_logmel_f16s_split: 16-bit storage only");
    return stft_logmel_f16s("vm_stft_logmel_f16s_split", raw, is_int16, n_clips, raw_len, win_length, hop, basis16, melw, n_mels, log_floor, dtype,
                            out, out_lo, stream);
}

#define VM_DISPATCH_VEC(T, C, ...)                       \
    do {                                                 \
        if ((C) % Elem<T>::kVec == 0) {                  \
            constexpr int V = Elem<T>::kVec;             \
            __VA_ARGS__;                                 \
        } else {                                         \
            constexpr int V = 1;                         \
            __VA_ARGS__;                                 \
        }                                                \
    } while (0)

extern "C" int vm_stack_windows(const void* x, int64_t n_clips, int M, int64_t rows, int C, int Cs, int dtype, void* out, void* stream) {
    VM_REQUIRE(x && out, "vm_stack_windows: null pointer");
    VM_REQUIRE(n_clips > 0 && M > 0 && rows > 0 && C > 0 && Cs >= 3 * C, "vm_stack_windows: bad sizes (Cs >= 3 C)");
    VM_DISPATCH_DTYPE(dtype, VM_DISPATCH_VEC(T, ((C % Elem<T>::kVec) || (Cs % Elem<T>::kVec)) ? 1 : C, {
        hipLaunchKernelGGL((stack_windows_kernel<T, V>), ew_grid(n_clips * M, rows * (Cs / V)), dim3(256), 0, (hipStream_t)stream, (const T*)x, n_clips, M,
                           (int)rows, C, Cs, (T*)out);
    }));
    return check_launch("vm_stack_windows");
}

extern "C" int vm_fold_windows(const void* dxs, int64_t n_clips, int M, int64_t L, int C, int Cs, int src_padded, int dtype, void* dx,
                               void* stream) {
    VM_REQUIRE(dxs && dx, "vm_fold_windows: null pointer");
    VM_REQUIRE(n_clips > 0 && M > 0 && L > 0 && C > 0 && Cs >= 3 * C, "vm_fold_windows: bad sizes (Cs >= 3 C)");
    VM_DISPATCH_DTYPE(dtype, VM_DISPATCH_VEC(T, ((C % Elem<T>::kVec) || (Cs % Elem<T>::kVec)) ? 1 : C, {
        hipLaunchKernelGGL((fold_windows_kernel<T, V>), ew_grid(n_clips * M, L * (C / V)), dim3(256), 0, (hipStream_t)stream, (const T*)dxs, n_clips, M,
                           (int)L, C, Cs, src_padded ? 1 : 0, (T*)dx);
    }));
    return check_launch("vm_fold_windows");
}

static int bn_pool2d_stack_fwd(const void* z, const void* z_lo, const float* scale, const float* shift, const float* drop, int64_t n_clips, int M,
                                      int64_t clips_per_tower, int64_t L, int C, int Cs, int dtype, void* q, void* xs, void* stream) {
    VM_REQUIRE(z && scale && shift && q && xs, "vm_bn_pool2d_stack_fwd: null pointer");
    VM_REQUIRE(n_clips > 0 && M >= 2 && clips_per_tower > 0 && L >= 2 && C > 0 && Cs >= 3 * C, "vm_bn_pool2d_stack_fwd: bad sizes (Cs >= 3 C)");
    VM_REQUIRE(n_clips * ((M + 1) / 2) < (1LL << 31), "vm_bn_pool2d_stack_fwd: too many windows");
    VM_DISPATCH_DTYPE(dtype, {
        VM_REQUIRE(C % Elem<T>::kVec == 0 && Cs % Elem<T>::kVec == 0 && C / Elem<T>::kVec <= 256,
                   "vm_bn_pool2d_stack_fwd: C and Cs must be multiples of the 16-byte vector, C / vector <= 256");
        hipLaunchKernelGGL((bn_pool2d_stack_fwd_kernel<T>), dim3((unsigned)(n_clips * ((M + 1) / 2))), dim3(256), 0, (hipStream_t)stream,
                           (const T*)z, scale, shift, drop, clips_per_tower * M, M, (int)L, C, Cs, (T*)q, (T*)xs, (const T*)z_lo);
    });
    return check_launch("vm_bn_pool2d_stack_fwd");
}

extern "C" int vm_bn_pool2d_stack_fwd(const void* z, const float* scale, const float* shift, const float* drop, int64_t n_clips, int M,
                                      int64_t clips_per_tower, int64_t L, int C, int Cs, int dtype, void* q, void* xs, void* stream) {
    return bn_pool2d_stack_fwd(z, nullptr, scale, shift, drop, n_clips, M, clips_per_tower, L, C, Cs, dtype, q, xs, stream);
}

// ... on a z that came as two planes of the storage type (vm_conv2d_first_fwd_split): the affine sees z + z_lo
extern "C" int vm_bn_pool2d_stack_fwd_split(const void* z, const void* z_lo, const float* scale, const float* shift, const float* drop,
                                            int64_t n_clips, int M, int64_t clips_per_tower, int64_t L, int C, int Cs, int dtype, void* q,
                                            void* xs, void* stream) {
    VM_REQUIRE(z_lo, "vm_bn_pool2d_stack_fwd_split: null pointer");
    VM_REQUIRE(dtype == VM_BF16 || dtype == VM_F16, "vm_bn_pool2d_stack_fwd_split: 16-bit storage only");
    return bn_pool2d_stack_fwd(z, z_lo, scale, shift, drop, n_clips, M, clips_per_tower, L, C, Cs, dtype, q, xs, stream);
}

extern "C" int64_t vm_fold_pool_windows_rows(int64_t L, int C, int Cs, int dtype) { return 1; }   // one row of sums per window

extern "C" int vm_fold_pool_windows_bwd(const void* dxs, const void* q, int64_t n_clips, int M, int64_t L, int C, int Cs, int src_padded,
                                        int dtype, void* dq, float* s0, float* sa, void* stream) {
    VM_REQUIRE(dxs && q && dq, "vm_fold_pool_windows_bwd: null pointer");
    VM_REQUIRE((s0 == nullptr) == (sa == nullptr), "vm_fold_pool_windows_bwd: s0 / sa must both be set or NULL");
    VM_REQUIRE(n_clips > 0 && M >= 2 && L > 0 && C > 0 && Cs >= 3 * C, "vm_fold_pool_windows_bwd: bad sizes (Cs >= 3 C)");
    VM_REQUIRE(s0 == nullptr || C <= 256, "vm_fold_pool_windows_bwd: the sums need C <= 256");
    VM_DISPATCH_DTYPE(dtype, VM_DISPATCH_VEC(T, ((C % Elem<T>::kVec) || (Cs % Elem<T>::kVec)) ? 1 : C, {
        VM_REQUIRE(C / V <= 256, "vm_fold_pool_windows_bwd: too many channels");
        hipLaunchKernelGGL((fold_pool_windows_bwd_kernel<T, V>), dim3((unsigned)(n_clips * ((M + 1) / 2))), dim3(256), 0, (hipStream_t)stream,
                           (const T*)dxs, (const T*)q, n_clips, M, (int)L, C, Cs, src_padded ? 1 : 0, (T*)dq, s0, sa);
    }));
    return check_launch("vm_fold_pool_windows_bwd");
}

extern "C" int vm_pool_windows_fwd(const void* q, int64_t n_clips, int M, int64_t rows, int C, int dtype, void* out, void* stream) {
    VM_REQUIRE(q && out, "vm_pool_windows_fwd: null pointer");
    VM_REQUIRE(n_clips > 0 && M >= 2 && rows > 0 && C > 0, "vm_pool_windows_fwd: bad sizes");
    VM_DISPATCH_DTYPE(dtype, VM_DISPATCH_VEC(T, (rows * C), {
        hipLaunchKernelGGL((pool_windows_fwd_kernel<T, V>), ew_grid(n_clips * (M / 2), rows * C / V), dim3(256), 0, (hipStream_t)stream, (const T*)q, n_clips, M,
                           (int)rows, C, (T*)out);
    }));
    return check_launch("vm_pool_windows_fwd");
}

extern "C" int vm_pool_windows_bwd(const void* q, const void* dout, int64_t n_clips, int M, int64_t L, int C, int dtype, void* dq,
                                   void* stream) {
    VM_REQUIRE(q && dout && dq, "vm_pool_windows_bwd: null pointer");
    VM_REQUIRE(n_clips > 0 && M >= 2 && L > 0 && C > 0, "vm_pool_windows_bwd: bad sizes");
    VM_DISPATCH_DTYPE(dtype, VM_DISPATCH_VEC(T, C, {
        hipLaunchKernelGGL((pool_windows_bwd_kernel<T, V>), ew_grid(n_clips * M, L * C / V), dim3(256), 0, (hipStream_t)stream, (const T*)q,
                           (const T*)dout, n_clips, M, (int)L, C, (T*)dq);
    }));
    return check_launch("vm_pool_windows_bwd");
}

extern "C" int vm_conv2d_first_supported(int C, int dtype) { return (C % 8 == 0 && C >= 8 && C <= 128) ? 1 : 0; }

static int conv2d_first_fwd(const void* in, const void* in_lo, const float* w, const float* bias, int64_t n_clips, int M, int64_t L, int Cs, int C,
                            int dtype, void* z, void* z_lo, float* stat_sum, float* stat_sq, void* stream) {
    VM_REQUIRE(in && w && bias && z, "vm_conv2d_first_fwd: null pointer");
    VM_REQUIRE((stat_sum == nullptr) == (stat_sq == nullptr), "vm_conv2d_first_fwd: stat_sum / stat_sq must both be set or NULL");
    VM_REQUIRE(n_clips > 0 && M > 0 && L > 0 && Cs >= 3 && vm_conv2d_first_supported(C, dtype), "vm_conv2d_first_fwd: bad sizes (C % 8 == 0, C <= 128, Cs >= 3)");
    const int64_t rows = (L + 127) / 128;
    VM_REQUIRE(rows < 65536, "vm_conv2d_first_fwd: window too long");
    VM_DISPATCH_DTYPE(dtype, {
        if constexpr (sizeof(T) == 2) {
            if (C % 32 == 0) {   // 16-bit storage: the matrix-pipe form, one wave per statistics row
                const int64_t n_rows = n_clips * M * rows;
                VM_REQUIRE(n_rows / 4 + 1 < (1LL << 31), "vm_conv2d_first_fwd: too many windows");
                launch_c2f_mfma<T>((const T*)in, (const T*)in_lo, w, bias, n_rows, M, (int)L, Cs, C, (T*)z, (T*)z_lo, stat_sum, stat_sq, (hipStream_t)stream);
                return check_launch("vm_conv2d_first_fwd");
            }
        }
        VM_REQUIRE(in_lo == nullptr && z_lo == nullptr, "vm_conv2d_first_fwd_split: 16-bit storage and C % 32 == 0 (the matrix-pipe form) only");
        hipLaunchKernelGGL((conv2d_first_fwd_kernel<T>), dim3((unsigned)(n_clips * M)), dim3(256), 0, (hipStream_t)stream,
                           (const T*)in, w, bias, M, (int)L, Cs, C, (T*)z, stat_sum, stat_sq);
    });
    return check_launch("vm_conv2d_first_fwd");
}

extern "C" int vm_conv2d_first_fwd(const void* in, const float* w, const float* bias, int64_t n_clips, int M, int64_t L, int Cs, int C,
                                   int dtype, void* z, float* stat_sum, float* stat_sq, void* stream) {
    return conv2d_first_fwd(in, nullptr, w, bias, n_clips, M, L, Cs, C, dtype, z, nullptr, stat_sum, stat_sq, stream);
}

extern "C" int vm_conv2d_first_fwd_split(const void* in, const void* in_lo, const float* w, const float* bias, int64_t n_clips, int M, int64_t L,
                                         int Cs, int C, int dtype, void* z, void* z_lo, float* stat_sum, float* stat_sq, void* stream) {
    VM_REQUIRE(in_lo, "vm_conv2d_first_fwd_split: null pointer");
    return conv2d_first_fwd(in, in_lo, w, bias, n_clips, M, L, Cs, C, dtype, z, z_lo, stat_sum, stat_sq, stream);
}

extern "C" int vm_conv2d_first_bn_pool_stack(const void* in, const void* in_lo, const float* w, const float* bias, const float* scale,
                                             const float* shift, const float* drop, int64_t n_clips, int M, int64_t clips_per_tower, int64_t L,
                                             int Cs, int C, int Cs2, int dtype, void* q, void* xs, void* stream) {
    VM_REQUIRE(in && in_lo && w && bias && scale && shift && q && xs, "vm_conv2d_first_bn_pool_stack: null pointer");
    VM_REQUIRE(n_clips > 0 && M >= 2 && clips_per_tower > 0 && L >= 2 && Cs >= 3 && Cs2 >= 3 * C && Cs2 % 8 == 0,
               "vm_conv2d_first_bn_pool_stack: bad sizes (M >= 2, L >= 2, Cs >= 3, Cs2 >= 3 C, Cs2 % 8 == 0)");
    VM_REQUIRE((dtype == VM_BF16 || dtype == VM_F16) && C % 32 == 0 && C >= 32 && C <= 128,
               "vm_conv2d_first_bn_pool_stack: 16-bit storage and C in {32, 64, 96, 128} (the matrix-pipe form) only");
    const int64_t n_items = n_clips * ((M + 1) / 2) * ((L + 127) / 128);
    VM_REQUIRE(n_items / 4 + 1 < (1LL << 31), "vm_conv2d_first_bn_pool_stack: too many windows");
    const dim3 grid((unsigned)((n_items + 3) / 4));
    const size_t lds = (size_t)4 * 3 * 16 * C * 2;
#define VM_LAUNCH_C2P(CB)                                                                                                                     \
    hipLaunchKernelGGL((conv2d_first_bn_pool_stack_kernel<T, CB>), grid, dim3(256), lds, (hipStream_t)stream, (const T*)in, (const T*)in_lo, w, \
                       bias, scale, shift, drop, n_items, clips_per_tower * M, M, (int)L, Cs, Cs2, (T*)q, (T*)xs)
    VM_DISPATCH_16(dtype, {
        switch (C / 32) {
            case 1: VM_LAUNCH_C2P(1); break;
            case 2: VM_LAUNCH_C2P(2); break;
            case 3: VM_LAUNCH_C2P(3); break;
            default: VM_LAUNCH_C2P(4); break;
        }
    });
#undef VM_LAUNCH_C2P
    return check_launch("vm_conv2d_first_bn_pool_stack");
}

static int64_t c2f_blocks(int64_t n_windows) { return n_windows < 2048 ? n_windows : 2048; }

extern "C" int64_t vm_conv2d_first_wgrad_workspace_bytes(int64_t n_clips, int M, int C) {
    const int64_t nel = 9LL * C;
    return (c2f_blocks(n_clips * M) + 1) * nel * (int64_t)sizeof(float) + slab_sum_part_bytes(nel) + 64;
}

extern "C" int vm_conv2d_first_wgrad(const void* in, const void* du, int64_t n_clips, int M, int64_t L, int Cs, int C, int dtype, void* ws,
                                     float* grad_w, void* stream) {
    VM_REQUIRE(in && du && ws && grad_w, "vm_conv2d_first_wgrad: null pointer");
    VM_REQUIRE(n_clips > 0 && M > 0 && L > 0 && Cs >= 3 && vm_conv2d_first_supported(C, dtype), "vm_conv2d_first_wgrad: bad sizes");
    const int64_t nw = n_clips * M, blocks = c2f_blocks(nw), nel = 9LL * C;
    const int wpb = (int)((nw + blocks - 1) / blocks);
    const int64_t used = (nw + wpb - 1) / wpb;
    float* slabs = (float*)ws;
    float* g9 = slabs + used * nel;
    VM_DISPATCH_DTYPE(dtype, {
        bool done = false;
        if constexpr (sizeof(T) == 2) {
            if (C % 32 == 0) {
                launch_c2w_mfma<T>((const T*)in, (const T*)du, nw, M, (int)L, C, wpb, slabs, (unsigned)used, (hipStream_t)stream);
                done = true;
            }
        }
        if (!done)
            hipLaunchKernelGGL((conv2d_first_wgrad_kernel<T>), dim3((unsigned)used), dim3(256), 0, (hipStream_t)stream, (const T*)in,
                               (const T*)du, nw, M, (int)L, C, wpb, slabs);
    });
    int rc = check_launch("vm_conv2d_first_wgrad");
    if (rc) return rc;
    rc = slab_sum(slabs, used, nel, g9, nel, nullptr, g9 + nel, (hipStream_t)stream);
    if (rc) return rc;
    hipLaunchKernelGGL(conv2d_first_scatter_kernel, dim3((unsigned)((3 * Cs * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g9, Cs, C,
                       grad_w);
    return check_launch("vm_conv2d_first_wgrad(scatter)");
}

extern "C" int vm_clip_max_fwd(const float* gmax, int64_t n_clips, int M, int M_valid, int C, float* out, int32_t* widx, void* stream) {
    VM_REQUIRE(gmax && out && widx, "vm_clip_max_fwd: null pointer");
    VM_REQUIRE(n_clips > 0 && M > 0 && M_valid > 0 && M_valid <= M && C > 0, "vm_clip_max_fwd: bad sizes");
    hipLaunchKernelGGL(clip_max_fwd_kernel, dim3((unsigned)((n_clips * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gmax, n_clips,
                       M, M_valid, C, out, widx);
    return check_launch("vm_clip_max_fwd");
}

extern "C" int vm_clip_max_bwd(const float* dout, const int32_t* widx, int64_t n_clips, int M, int C, float* dg, void* stream) {
    VM_REQUIRE(dout && widx && dg, "vm_clip_max_bwd: null pointer");
    VM_REQUIRE(n_clips > 0 && M > 0 && C > 0, "vm_clip_max_bwd: bad sizes");
    hipLaunchKernelGGL(clip_max_bwd_kernel, dim3((unsigned)((n_clips * M * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dout, widx,
                       n_clips, M, C, dg);
    return check_launch("vm_clip_max_bwd");
}
