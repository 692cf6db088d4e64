// BatchNormalization -> SpatialDropout1D -> MaxPool1D between the conv blocks (voicemap/models.py:17-19,
// 23-25, 28-30, 33-35), forward and backward.  All of it is HBM-bound streaming over channels-last tensors:
// 16-byte vectors along C, fp32 math, per-tower statistics (one tower per encoder call, models.py:52-53).
//
// Backward of one block, given dp = dL/d(pooled output):
//   dy[n,t,c]   = drop[n,c] * dp[n,t/pool,c]  at the FIRST maximum of each pool window, else 0
//   zhat        = (z - mean) * invstd
//   dz          = scale * (dy - mean_t(dy) - zhat * mean_t(dy*zhat)),   scale = gamma*invstd   (batch-stat BN)
//   du          = dz * [z > 0]                                           (ReLU fused into the conv)
//   dgamma      = sum(dy*zhat), dbeta = sum(dy)   (both towers added)
// pass 1 (reduce) builds the two sums, pass 2 (apply) writes du into a halo-padded tensor for dgrad/wgrad.
#include "common.hpp"

namespace vm {

constexpr int BN_SEG = 8;  // partial-sum segments per window in the backward kernels

// ---------------------------------------------------------------------------------------------
// Column sums of (rows, C) fp32 partial matrices, two deterministic stages so that the long reduction over
// rows is spread over many workgroups instead of C/64 of them:
//   stage 1: grid (C/64, segments*CR_CHUNKS); a workgroup (64 channels x 16 row lanes) sums one row chunk of
//            one segment (tower) in fp64 -> ws[(seg*CR_CHUNKS + chunk)][which][C]
//   stage 2: the consumer kernel adds the CR_CHUNKS doubles per (segment, channel) in a fixed order.
constexpr int CR_CHUNKS = 32;

__global__ __launch_bounds__(1024) void colreduce_stage1_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                 int64_t rows_per_seg, int C, double* __restrict__ ws) {
    __shared__ double red[2][16][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int seg = blockIdx.y / CR_CHUNKS, chunk = blockIdx.y % CR_CHUNKS;
    const int64_t per = (rows_per_seg + CR_CHUNKS - 1) / CR_CHUNKS;
    const int64_t r_lo = chunk * per;
    int64_t r_hi = r_lo + per;
    if (r_hi > rows_per_seg) r_hi = rows_per_seg;
    double s = 0.0, q = 0.0;
    if (c < C) {
        const int64_t base = (int64_t)seg * rows_per_seg;
        for (int64_t r = r_lo + rg; r < r_hi; r += 16) {
            s += (double)a[(base + r) * C + c];
            if (b != nullptr) q += (double)b[(base + r) * C + c];
        }
    }
    red[0][rg][cl] = s;
    red[1][rg][cl] = q;
    __syncthreads();
    if (rg == 0 && c < C) {
        double ss = 0.0, qq = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            ss += red[0][i][cl];
            qq += red[1][i][cl];
        }
        ws[((int64_t)blockIdx.y * 2 + 0) * C + c] = ss;
        ws[((int64_t)blockIdx.y * 2 + 1) * C + c] = qq;
    }
}

__device__ inline void colreduce_stage2(const double* __restrict__ ws, int seg, int C, int c, double& s, double& q) {
    s = 0.0;
    q = 0.0;
    for (int k = 0; k < CR_CHUNKS; ++k) {
        s += ws[((int64_t)(seg * CR_CHUNKS + k) * 2 + 0) * C + c];
        q += ws[((int64_t)(seg * CR_CHUNKS + k) * 2 + 1) * C + c];
    }
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ ws, int n_towers, int C, double count,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float eps, float momentum, int unbiased, float* moving_mean,
                                                           float* moving_var, float* mean, float* invstd, float* scale,
                                                           float* shift) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float mm = 0.f, mv = 0.f;
    if (moving_mean != nullptr) {
        mm = moving_mean[c];
        mv = moving_var[c];
    }
    for (int tw = 0; tw < n_towers; ++tw) {
        double ss, qq;
        colreduce_stage2(ws, tw, C, c, ss, qq);
        const double m = ss / count;
        double var = qq / count - m * m;
        if (var < 0.0) var = 0.0;
        const float istd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gamma[c] * istd;
        mean[tw * C + c] = (float)m;
        invstd[tw * C + c] = istd;
        scale[tw * C + c] = sc;
        shift[tw * C + c] = beta[c] - (float)m * sc;
        if (moving_mean != nullptr) {
            double vv = var;
            if (unbiased) vv = var * (count / (count - (1.0 + (double)eps)));
            mm = mm - (mm - (float)m) * (1.0f - momentum);
            mv = mv - (mv - (float)vv) * (1.0f - momentum);
        }
    }
    if (moving_mean != nullptr) {
        moving_mean[c] = mm;
        moving_var[c] = mv;
    }
}

__global__ void bn_infer_affine_kernel(const float* gamma, const float* beta, const float* mm, const float* mv, float eps,
                                       int C, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] * (1.0f / sqrtf(mv[c] + eps));
    scale[c] = sc;
    shift[c] = beta[c] - mm[c] * sc;
}

// ---------------------------------------------------------------------------------------------
// grid = (n_windows, BN_SEG); threads: P lanes over channel vectors x RP row lanes (no integer divisions on the
// element path); a block owns pooled rows q = seg, seg+BN_SEG, ... of one window.
template <typename T, int POOL>
__global__ __launch_bounds__(256) void bn_drop_pool_fwd_kernel(const T* __restrict__ z, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, const float* __restrict__ drop,
                                                               int64_t wpt, int64_t L, int C, int P, T* __restrict__ out) {
    constexpr int VEC = Elem<T>::kVec;
    const int tid = threadIdx.x;
    const int RP = 256 / P;
    const int pl = tid % P, rl = tid / P;
    const int CV = C / VEC;
    const int64_t n = blockIdx.x;
    const int seg = blockIdx.y;
    const int64_t tw = n / wpt;
    const int64_t Lq = L / POOL;
    for (int cv = pl; cv < CV; cv += P) {
        const int c0 = cv * VEC;
        float sc[VEC], sh[VEC], dr[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            sc[i] = scale[tw * C + c0 + i];
            sh[i] = shift[tw * C + c0 + i];
            dr[i] = drop ? drop[n * C + c0 + i] : 1.0f;
        }
        const T* zrow = z + n * L * C + c0;
        T* orow = out + (n * (Lq + 2) + 1) * C + c0;
        for (int64_t q = seg + (int64_t)rl * BN_SEG; q < Lq; q += (int64_t)RP * BN_SEG) {
            Vec16<T> v[POOL];
#pragma unroll
            for (int j = 0; j < POOL; ++j) v[j] = load16<T>(zrow + (q * POOL + j) * C);
            Vec16<T> o;
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float best = fmaf(v[0].get(i), sc[i], sh[i]) * dr[i];
#pragma unroll
                for (int j = 1; j < POOL; ++j) {
                    const float y = fmaf(v[j].get(i), sc[i], sh[i]) * dr[i];
                    best = y > best ? y : best;
                }
                o.set(i, best);
            }
            store16<T>(orow + q * C, o);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Shared body of the two backward passes.  grid = (n_windows, BN_SEG); a block owns pool groups
// q = seg, seg+BN_SEG, ... of one window.  threads: P lanes over channel vectors x RP row lanes.
template <typename T, int POOL, bool APPLY>
__global__ __launch_bounds__(256) void bn_pool_bwd_kernel(const T* __restrict__ z, const T* __restrict__ dp,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const float* __restrict__ drop, const float* __restrict__ c1,
                                                          const float* __restrict__ c2, int64_t wpt, int64_t L, int C, int P,
                                                          T* __restrict__ du, float* __restrict__ part_a,
                                                          float* __restrict__ part_b, const float* __restrict__ sp_dg,
                                                          const int32_t* __restrict__ sp_idx) {
    // sp_dg / sp_idx (optional): dp is given in its sparse GlobalMaxPool1D-backward form -- dp[n][q][c] = sp_dg[n][c] if
    // q == sp_idx[n][c] else 0 -- instead of as a dense tensor (saves writing and re-reading it for the last block)
    constexpr int VEC = Elem<T>::kVec;
    __shared__ float red[2][256][VEC];
    const int tid = threadIdx.x;
    const int RP = 256 / P;
    const int pl = tid % P, rl = tid / P;
    const int CV = C / VEC;
    const int64_t n = blockIdx.x;
    const int seg = blockIdx.y;
    const int64_t tw = n / wpt;
    const int64_t Lq = L / POOL;
    const int64_t Q = APPLY ? (L + POOL - 1) / POOL : Lq;  // apply also covers the remainder rows of a floor pool

    for (int cvb = 0; cvb < CV; cvb += P) {
        const int cv = cvb + pl;
        const bool cok = cv < CV;
        const int c0 = cv * VEC;
        // Per-channel constants folded on entry (fewer live registers -> more waves in flight on these HBM-bound passes):
        //   arg-max of y = (z*scale+shift)*drop over a pool window == arg-max of z if scale*drop >= 0, else arg-min
        //   reduce: dy = drop*dp,  dy*zhat = dp * (ka*z + kb),            ka = drop*invstd, kb = -drop*invstd*mean
        //   apply : du = [z>0] * (kc*z + kb + [arg] ka*dp),                ka = scale*drop, kb = scale*(invstd*c2*mean - c1),
        //                                                                  kc = -scale*invstd*c2
        float ka[VEC], kb[VEC], kc[VEC], accA[VEC], accB[VEC];
        bool use_min[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            accA[i] = 0.f;
            accB[i] = 0.f;
            ka[i] = kb[i] = kc[i] = 0.f;
            use_min[i] = false;
            if (cok) {
                const float sc = scale[tw * C + c0 + i];
                const float mu = mean[tw * C + c0 + i];
                const float is = invstd[tw * C + c0 + i];
                const float dr = drop ? drop[n * C + c0 + i] : 1.0f;
                use_min[i] = sc * dr < 0.f;
                if (APPLY) {
                    const float k1 = c1[tw * C + c0 + i], k2 = c2[tw * C + c0 + i];
                    ka[i] = sc * dr;
                    kb[i] = sc * (is * k2 * mu - k1);
                    kc[i] = -sc * is * k2;
                } else {
                    ka[i] = dr * is;
                    kb[i] = -dr * is * mu;
                    kc[i] = dr;
                }
            }
        }
        float spv[VEC];
        int spi[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            spv[i] = 0.f;
            spi[i] = -1;
            if (cok && sp_idx != nullptr) {
                spi[i] = sp_idx[n * C + c0 + i];
                spv[i] = Elem<T>::to_f(Elem<T>::from_f(sp_dg[n * C + c0 + i]));  // same rounding as the dense dp tensor
            }
        }
        if (cok) {
            for (int64_t q = seg + (int64_t)rl * BN_SEG; q < Q; q += (int64_t)RP * BN_SEG) {
                Vec16<T> zv[POOL];
                int nrows = POOL;
                if (q * POOL + POOL > L) nrows = (int)(L - q * POOL);
#pragma unroll
                for (int j = 0; j < POOL; ++j)
                    if (j < nrows) zv[j] = load16<T>(z + (n * L + q * POOL + j) * C + c0);
                const bool has_dp = q < Lq;
                Vec16<T> dv;
                if (has_dp && sp_idx == nullptr) dv = load16<T>(dp + (n * Lq + q) * C + c0);
                Vec16<T> ov[POOL];
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    float zj[POOL];
#pragma unroll
                    for (int j = 0; j < POOL; ++j) zj[j] = j < nrows ? zv[j].get(i) : 0.f;
                    float ext = zj[0];
                    int arg = 0;
#pragma unroll
                    for (int j = 1; j < POOL; ++j) {
                        const bool better = use_min[i] ? (zj[j] < ext) : (zj[j] > ext);  // strict: first extreme wins
                        if (j < nrows && better) {
                            ext = zj[j];
                            arg = j;
                        }
                    }
                    float dpv = 0.f;
                    if (has_dp) dpv = sp_idx != nullptr ? (spi[i] == (int)q ? spv[i] : 0.f) : dv.get(i);
                    if (!APPLY) {
                        accA[i] += kc[i] * dpv;
                        accB[i] += dpv * fmaf(ka[i], ext, kb[i]);
                    } else {
                        const float ady = ka[i] * dpv;
#pragma unroll
                        for (int j = 0; j < POOL; ++j) {
                            float g = fmaf(kc[i], zj[j], kb[i]) + (j == arg ? ady : 0.f);
                            g = zj[j] > 0.f ? g : 0.f;
                            ov[j].set(i, g);
                            if (j < nrows) accA[i] += ov[j].get(i);
                        }
                    }
                }
                if (APPLY) {
#pragma unroll
                    for (int j = 0; j < POOL; ++j)
                        if (j < nrows) store16<T>(du + (n * (L + 2) + 1 + q * POOL + j) * C + c0, ov[j]);
                }
            }
        }
        // reduce over the RP row lanes
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            red[0][tid][i] = accA[i];
            red[1][tid][i] = accB[i];
        }
        __syncthreads();
        if (rl == 0 && cok) {
            const int64_t row = n * BN_SEG + seg;
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float a = 0.f, b = 0.f;
                for (int r = 0; r < RP; ++r) {
                    a += red[0][r * P + pl][i];
                    b += red[1][r * P + pl][i];
                }
                part_a[row * C + c0 + i] = a;
                if (!APPLY) part_b[row * C + c0 + i] = b;
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* __restrict__ ws, int n_towers, int C, double count,
                                                               float* c1, float* c2, float* grad_gamma, float* grad_beta) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double gg = 0.0, gb = 0.0;
    for (int tw = 0; tw < n_towers; ++tw) {
        double ss, qq;
        colreduce_stage2(ws, tw, C, c, ss, qq);
        c1[tw * C + c] = (float)(ss / count);
        c2[tw * C + c] = (float)(qq / count);
        gb += ss;
        gg += qq;
    }
    grad_gamma[c] = (float)gg;
    grad_beta[c] = (float)gb;
}

__global__ __launch_bounds__(256) void colsum_finalize_kernel(const double* __restrict__ ws, int C, float* out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double ss, qq;
    colreduce_stage2(ws, 0, C, c, ss, qq);
    out[c] = (float)ss;
}

static int lanes_for(int cv) {
    int p = 1;
    while (p < cv && p < 256) p <<= 1;
    return p;
}

}  // namespace vm

using namespace vm;

extern "C" int64_t vm_colreduce_workspace_bytes(int n_segments, int C) {
    return (int64_t)n_segments * CR_CHUNKS * 2 * C * (int64_t)sizeof(double);
}

extern "C" int vm_bn_finalize(const float* stat_sum, const float* stat_sq, int64_t rows_per_tower, int n_towers, int C,
                              double count_per_tower, const float* gamma, const float* beta, float eps, float momentum,
                              int unbiased_moving_var, float* moving_mean, float* moving_var, float* mean, float* invstd,
                              float* scale, float* shift, void* ws, void* stream) {
    VM_REQUIRE(stat_sum && stat_sq && gamma && beta && mean && invstd && scale && shift && ws, "vm_bn_finalize: null pointer");
    VM_REQUIRE((moving_mean == nullptr) == (moving_var == nullptr), "vm_bn_finalize: moving stats must both be set or NULL");
    VM_REQUIRE(rows_per_tower > 0 && n_towers > 0 && C > 0 && count_per_tower > 1.0, "vm_bn_finalize: bad sizes");
    hipLaunchKernelGGL(colreduce_stage1_kernel, dim3((C + 63) / 64, n_towers * CR_CHUNKS), dim3(1024), 0, (hipStream_t)stream,
                       stat_sum, stat_sq, rows_per_tower, C, (double*)ws);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const double*)ws,
                       n_towers, C, count_per_tower, gamma, beta, eps, momentum, unbiased_moving_var, moving_mean, moving_var,
                       mean, invstd, scale, shift);
    return check_launch("vm_bn_finalize");
}

extern "C" int vm_bn_infer_affine(const float* gamma, const float* beta, const float* moving_mean, const float* moving_var,
                                  float eps, int C, float* scale, float* shift, void* stream) {
    VM_REQUIRE(gamma && beta && moving_mean && moving_var && scale && shift && C > 0, "vm_bn_infer_affine: bad argument");
    hipLaunchKernelGGL(bn_infer_affine_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                       moving_mean, moving_var, eps, C, scale, shift);
    return check_launch("vm_bn_infer_affine");
}

#define VM_DISPATCH_POOL(pool, ...)                                         \
    do {                                                                    \
        if ((pool) == 2) {                                                  \
            constexpr int POOL = 2;                                         \
            __VA_ARGS__;                                                    \
        } else if ((pool) == 4) {                                           \
            constexpr int POOL = 4;                                         \
            __VA_ARGS__;                                                    \
        } else if ((pool) == 1) {                                           \
            constexpr int POOL = 1;                                         \
            __VA_ARGS__;                                                    \
        } else {                                                            \
            vm::set_error("unsupported pool size %d (1, 2, 4)", (int)(pool)); \
            return VM_ERR_UNSUPPORTED;                                      \
        }                                                                   \
    } while (0)

extern "C" int vm_bn_drop_pool_fwd(const void* z, const float* scale, const float* shift, const float* drop, int64_t n_windows,
                                   int64_t windows_per_tower, int64_t L, int C, int pool, int dtype, void* out, void* stream) {
    VM_REQUIRE(z && scale && shift && out, "vm_bn_drop_pool_fwd: null pointer");
    VM_REQUIRE(n_windows > 0 && windows_per_tower > 0 && L >= pool && C % 8 == 0, "vm_bn_drop_pool_fwd: bad sizes");
    VM_DISPATCH_DTYPE(dtype, VM_DISPATCH_POOL(pool, {
        const int P = lanes_for(C / Elem<T>::kVec);
        hipLaunchKernelGGL((bn_drop_pool_fwd_kernel<T, POOL>), dim3((unsigned)n_windows, BN_SEG), dim3(256), 0,
                           (hipStream_t)stream, (const T*)z, scale, shift, drop, windows_per_tower, L, C, P, (T*)out);
    }));
    return check_launch("vm_bn_drop_pool_fwd");
}

extern "C" int vm_bn_part_rows(void) { return BN_SEG; }

static int bn_pool_bwd_reduce_impl(const void* z, const void* dp, const float* sp_dg, const int32_t* sp_idx, const float* scale,
                                   const float* shift, const float* mean, const float* invstd, const float* drop,
                                   int64_t n_windows, int64_t windows_per_tower, int64_t L, int C, int pool, int dtype,
                                   float* part_dy, float* part_dyz, void* stream) {
    VM_REQUIRE(z && (dp || (sp_dg && sp_idx)) && scale && shift && mean && invstd && part_dy && part_dyz,
               "vm_bn_pool_bwd_reduce: null pointer");
    VM_REQUIRE(n_windows > 0 && windows_per_tower > 0 && L >= pool && C % 8 == 0, "vm_bn_pool_bwd_reduce: bad sizes");
    VM_DISPATCH_DTYPE(dtype, VM_DISPATCH_POOL(pool, {
        const int P = lanes_for(C / Elem<T>::kVec);
        hipLaunchKernelGGL((bn_pool_bwd_kernel<T, POOL, false>), dim3((unsigned)n_windows, BN_SEG), dim3(256), 0,
                           (hipStream_t)stream, (const T*)z, (const T*)dp, scale, shift, mean, invstd, drop,
                           (const float*)nullptr, (const float*)nullptr, windows_per_tower, L, C, P, (T*)nullptr, part_dy,
                           part_dyz, sp_dg, sp_idx);
    }));
    return check_launch("vm_bn_pool_bwd_reduce");
}

extern "C" int vm_bn_pool_bwd_reduce(const void* z, const void* dp, const float* scale, const float* shift, const float* mean,
                                     const float* invstd, const float* drop, int64_t n_windows, int64_t windows_per_tower,
                                     int64_t L, int C, int pool, int dtype, float* part_dy, float* part_dyz, void* stream) {
    VM_REQUIRE(dp, "vm_bn_pool_bwd_reduce: null pointer");
    return bn_pool_bwd_reduce_impl(z, dp, nullptr, nullptr, scale, shift, mean, invstd, drop, n_windows, windows_per_tower, L, C,
                                   pool, dtype, part_dy, part_dyz, stream);
}

extern "C" int vm_bn_pool_bwd_reduce_gmax(const void* z, const float* dg, const int32_t* gidx, const float* scale,
                                          const float* shift, const float* mean, const float* invstd, const float* drop,
                                          int64_t n_windows, int64_t windows_per_tower, int64_t L, int C, int pool, int dtype,
                                          float* part_dy, float* part_dyz, void* stream) {
    VM_REQUIRE(dg && gidx, "vm_bn_pool_bwd_reduce_gmax: null pointer");
    return bn_pool_bwd_reduce_impl(z, nullptr, dg, gidx, scale, shift, mean, invstd, drop, n_windows, windows_per_tower, L, C, pool,
                                   dtype, part_dy, part_dyz, stream);
}

extern "C" int vm_bn_bwd_finalize(const float* part_dy, const float* part_dyz, int64_t n_windows, int64_t windows_per_tower,
                                  int C, double count_per_tower, float* c1, float* c2, float* grad_gamma, float* grad_beta,
                                  void* ws, void* stream) {
    VM_REQUIRE(part_dy && part_dyz && c1 && c2 && grad_gamma && grad_beta && ws, "vm_bn_bwd_finalize: null pointer");
    VM_REQUIRE(n_windows > 0 && windows_per_tower > 0 && n_windows % windows_per_tower == 0,
               "vm_bn_bwd_finalize: n_windows must be a multiple of windows_per_tower");
    const int n_towers = (int)(n_windows / windows_per_tower);
    hipLaunchKernelGGL(colreduce_stage1_kernel, dim3((C + 63) / 64, n_towers * CR_CHUNKS), dim3(1024), 0, (hipStream_t)stream,
                       part_dy, part_dyz, windows_per_tower * BN_SEG, C, (double*)ws);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const double*)ws,
                       n_towers, C, count_per_tower, c1, c2, grad_gamma, grad_beta);
    return check_launch("vm_bn_bwd_finalize");
}

static int bn_pool_bwd_apply_impl(const void* z, const void* dp, const float* sp_dg, const int32_t* sp_idx, const float* scale,
                                  const float* shift, const float* mean, const float* invstd, const float* drop, const float* c1,
                                  const float* c2, int64_t n_windows, int64_t windows_per_tower, int64_t L, int C, int pool,
                                  int dtype, void* du, float* part_du, void* stream) {
    VM_REQUIRE(z && (dp || (sp_dg && sp_idx)) && scale && shift && mean && invstd && c1 && c2 && du && part_du,
               "vm_bn_pool_bwd_apply: null pointer");
    VM_REQUIRE(n_windows > 0 && windows_per_tower > 0 && L >= pool && C % 8 == 0, "vm_bn_pool_bwd_apply: bad sizes");
    VM_DISPATCH_DTYPE(dtype, VM_DISPATCH_POOL(pool, {
        const int P = lanes_for(C / Elem<T>::kVec);
        hipLaunchKernelGGL((bn_pool_bwd_kernel<T, POOL, true>), dim3((unsigned)n_windows, BN_SEG), dim3(256), 0,
                           (hipStream_t)stream, (const T*)z, (const T*)dp, scale, shift, mean, invstd, drop, c1, c2,
                           windows_per_tower, L, C, P, (T*)du, part_du, (float*)nullptr, sp_dg, sp_idx);
    }));
    return check_launch("vm_bn_pool_bwd_apply");
}

extern "C" int vm_bn_pool_bwd_apply(const void* z, const void* dp, const float* scale, const float* shift, const float* mean,
                                    const float* invstd, const float* drop, const float* c1, const float* c2, int64_t n_windows,
                                    int64_t windows_per_tower, int64_t L, int C, int pool, int dtype, void* du, float* part_du,
                                    void* stream) {
    VM_REQUIRE(dp, "vm_bn_pool_bwd_apply: null pointer");
    return bn_pool_bwd_apply_impl(z, dp, nullptr, nullptr, scale, shift, mean, invstd, drop, c1, c2, n_windows, windows_per_tower,
                                  L, C, pool, dtype, du, part_du, stream);
}

extern "C" int vm_bn_pool_bwd_apply_gmax(const void* z, const float* dg, const int32_t* gidx, const float* scale,
                                         const float* shift, const float* mean, const float* invstd, const float* drop,
                                         const float* c1, const float* c2, int64_t n_windows, int64_t windows_per_tower, int64_t L,
                                         int C, int pool, int dtype, void* du, float* part_du, void* stream) {
    VM_REQUIRE(dg && gidx, "vm_bn_pool_bwd_apply_gmax: null pointer");
    return bn_pool_bwd_apply_impl(z, nullptr, dg, gidx, scale, shift, mean, invstd, drop, c1, c2, n_windows, windows_per_tower, L,
                                  C, pool, dtype, du, part_du, stream);
}

extern "C" int vm_colsum(const float* part, int64_t rows, int C, float* out, void* ws, void* stream) {
    VM_REQUIRE(part && out && ws && rows > 0 && C > 0, "vm_colsum: bad argument");
    hipLaunchKernelGGL(colreduce_stage1_kernel, dim3((C + 63) / 64, CR_CHUNKS), dim3(1024), 0, (hipStream_t)stream, part,
                       (const float*)nullptr, rows, C, (double*)ws);
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const double*)ws, C,
                       out);
    return check_launch("vm_colsum");
}
