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
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ stat_sum, const float* __restrict__ stat_sq,
                                                            int64_t rows_per_tower, int n_towers, int C, double count,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float eps, float momentum, int unbiased, float* moving_mean,
                                                            float* moving_var, float* mean, float* invstd, float* scale,
                                                            float* shift) {
    __shared__ double red[2][16][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const bool cok = c < C;
    float mm = 0.f, mv = 0.f;
    if (cok && rg == 0 && moving_mean != nullptr) {
        mm = moving_mean[c];
        mv = moving_var[c];
    }
    for (int tw = 0; tw < n_towers; ++tw) {
        double s = 0.0, q = 0.0;
        if (cok) {
            const int64_t r0 = (int64_t)tw * rows_per_tower;
            for (int64_t r = rg; r < rows_per_tower; r += 16) {
                s += (double)stat_sum[(r0 + r) * C + c];
                q += (double)stat_sq[(r0 + r) * C + c];
            }
        }
        red[0][rg][cl] = s;
        red[1][rg][cl] = q;
        __syncthreads();
        if (rg == 0 && cok) {
            double ss = 0.0, qq = 0.0;
            for (int i = 0; i < 16; ++i) {
                ss += red[0][i][cl];
                qq += red[1][i][cl];
            }
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
        __syncthreads();
    }
    if (cok && rg == 0 && moving_mean != nullptr) {
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
template <typename T, int POOL>
__global__ __launch_bounds__(256) void bn_drop_pool_fwd_kernel(const T* __restrict__ z, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, const float* __restrict__ drop,
                                                               int64_t total, int64_t wpt, int64_t L, int C, T* __restrict__ out) {
    constexpr int VEC = Elem<T>::kVec;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int CV = C / VEC;
    const int64_t Lq = L / POOL;
    const int cv = (int)(idx % CV);
    const int64_t r = idx / CV;
    const int64_t q = r % Lq, n = r / Lq;
    const int c0 = cv * VEC;
    const int64_t tw = n / wpt;
    float sc[VEC], sh[VEC], dr[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        sc[i] = scale[tw * C + c0 + i];
        sh[i] = shift[tw * C + c0 + i];
        dr[i] = drop ? drop[n * C + c0 + i] : 1.0f;
    }
    float best[VEC];
#pragma unroll
    for (int j = 0; j < POOL; ++j) {
        const Vec16<T> v = load16<T>(z + (n * L + q * POOL + j) * C + c0);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const float y = fmaf(v.get(i), sc[i], sh[i]) * dr[i];
            best[i] = (j == 0 || y > best[i]) ? y : best[i];
        }
    }
    Vec16<T> o;
#pragma unroll
    for (int i = 0; i < VEC; ++i) o.set(i, best[i]);
    store16<T>(out + (n * (Lq + 2) + 1 + q) * C + c0, o);
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
                                                          float* __restrict__ part_b) {
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
        float sc[VEC], sh[VEC], dr[VEC], mu[VEC], is[VEC], k1[VEC], k2[VEC], accA[VEC], accB[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            accA[i] = 0.f;
            accB[i] = 0.f;
            if (cok) {
                sc[i] = scale[tw * C + c0 + i];
                sh[i] = shift[tw * C + c0 + i];
                mu[i] = mean[tw * C + c0 + i];
                is[i] = invstd[tw * C + c0 + i];
                dr[i] = drop ? drop[n * C + c0 + i] : 1.0f;
                k1[i] = APPLY ? c1[tw * C + c0 + i] : 0.f;
                k2[i] = APPLY ? c2[tw * C + c0 + i] : 0.f;
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
                if (has_dp) dv = load16<T>(dp + (n * Lq + q) * C + c0);
                int arg[VEC];
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    float best = 0.f;
                    arg[i] = 0;
#pragma unroll
                    for (int j = 0; j < POOL; ++j) {
                        if (j < nrows) {
                            const float y = fmaf(zv[j].get(i), sc[i], sh[i]) * dr[i];
                            if (j == 0 || y > best) {
                                best = y;
                                arg[i] = j;
                            }
                        }
                    }
                }
                if (!APPLY) {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        float zsel = zv[0].get(i);
#pragma unroll
                        for (int j = 1; j < POOL; ++j) zsel = (arg[i] == j) ? zv[j].get(i) : zsel;
                        const float dy = dr[i] * dv.get(i);
                        accA[i] += dy;
                        accB[i] += dy * ((zsel - mu[i]) * is[i]);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < POOL; ++j) {
                        if (j < nrows) {
                            Vec16<T> o;
#pragma unroll
                            for (int i = 0; i < VEC; ++i) {
                                const float zz = zv[j].get(i);
                                const float dy = (has_dp && arg[i] == j) ? dr[i] * dv.get(i) : 0.f;
                                const float zh = (zz - mu[i]) * is[i];
                                float g = sc[i] * (dy - k1[i] - zh * k2[i]);
                                g = zz > 0.f ? g : 0.f;
                                o.set(i, g);
                                accA[i] += o.get(i);
                            }
                            store16<T>(du + (n * (L + 2) + 1 + q * POOL + j) * C + c0, o);
                        }
                    }
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

__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ part_dy, const float* __restrict__ part_dyz,
                                                                int64_t rows_per_tower, int n_towers, int C, double count,
                                                                float* c1, float* c2, float* grad_gamma, float* grad_beta) {
    __shared__ double red[2][16][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const bool cok = c < C;
    double gg = 0.0, gb = 0.0;
    for (int tw = 0; tw < n_towers; ++tw) {
        double s = 0.0, q = 0.0;
        if (cok) {
            const int64_t r0 = (int64_t)tw * rows_per_tower;
            for (int64_t r = rg; r < rows_per_tower; r += 16) {
                s += (double)part_dy[(r0 + r) * C + c];
                q += (double)part_dyz[(r0 + r) * C + c];
            }
        }
        red[0][rg][cl] = s;
        red[1][rg][cl] = q;
        __syncthreads();
        if (rg == 0 && cok) {
            double ss = 0.0, qq = 0.0;
            for (int i = 0; i < 16; ++i) {
                ss += red[0][i][cl];
                qq += red[1][i][cl];
            }
            c1[tw * C + c] = (float)(ss / count);
            c2[tw * C + c] = (float)(qq / count);
            gb += ss;
            gg += qq;
        }
        __syncthreads();
    }
    if (rg == 0 && cok) {
        grad_gamma[c] = (float)gg;
        grad_beta[c] = (float)gb;
    }
}

__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ part, int64_t rows, int C, float* out) {
    __shared__ double red[16][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const bool cok = c < C;
    double s = 0.0;
    if (cok)
        for (int64_t r = rg; r < rows; r += 16) s += (double)part[r * C + c];
    red[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && cok) {
        double ss = 0.0;
        for (int i = 0; i < 16; ++i) ss += red[i][cl];
        out[c] = (float)ss;
    }
}

static int lanes_for(int cv) {
    int p = 1;
    while (p < cv && p < 256) p <<= 1;
    return p;
}

}  // namespace vm

using namespace vm;

extern "C" int vm_bn_finalize(const float* stat_sum, const float* stat_sq, int64_t rows_per_tower, int n_towers, int C,
                              double count_per_tower, const float* gamma, const float* beta, float eps, float momentum,
                              int unbiased_moving_var, float* moving_mean, float* moving_var, float* mean, float* invstd,
                              float* scale, float* shift, void* stream) {
    VM_REQUIRE(stat_sum && stat_sq && gamma && beta && mean && invstd && scale && shift, "vm_bn_finalize: null pointer");
    VM_REQUIRE((moving_mean == nullptr) == (moving_var == nullptr), "vm_bn_finalize: moving stats must both be set or NULL");
    VM_REQUIRE(rows_per_tower > 0 && n_towers > 0 && C > 0 && count_per_tower > 1.0, "vm_bn_finalize: bad sizes");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 63) / 64), dim3(1024), 0, (hipStream_t)stream, stat_sum, stat_sq,
                       rows_per_tower, n_towers, C, count_per_tower, gamma, beta, eps, momentum, unbiased_moving_var,
                       moving_mean, moving_var, mean, invstd, scale, shift);
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
        const int64_t total = n_windows * (L / POOL) * (C / Elem<T>::kVec);
        hipLaunchKernelGGL((bn_drop_pool_fwd_kernel<T, POOL>), dim3((unsigned)cdiv(total, 256)), dim3(256), 0,
                           (hipStream_t)stream, (const T*)z, scale, shift, drop, total, windows_per_tower, L, C, (T*)out);
    }));
    return check_launch("vm_bn_drop_pool_fwd");
}

extern "C" int vm_bn_part_rows(void) { return BN_SEG; }

extern "C" int vm_bn_pool_bwd_reduce(const void* z, const void* dp, const float* scale, const float* shift, const float* mean,
                                     const float* invstd, const float* drop, int64_t n_windows, int64_t windows_per_tower,
                                     int64_t L, int C, int pool, int dtype, float* part_dy, float* part_dyz, void* stream) {
    VM_REQUIRE(z && dp && scale && shift && mean && invstd && part_dy && part_dyz, "vm_bn_pool_bwd_reduce: null pointer");
    VM_REQUIRE(n_windows > 0 && windows_per_tower > 0 && L >= pool && C % 8 == 0, "vm_bn_pool_bwd_reduce: bad sizes");
    VM_DISPATCH_DTYPE(dtype, VM_DISPATCH_POOL(pool, {
        const int P = lanes_for(C / Elem<T>::kVec);
        hipLaunchKernelGGL((bn_pool_bwd_kernel<T, POOL, false>), dim3((unsigned)n_windows, BN_SEG), dim3(256), 0,
                           (hipStream_t)stream, (const T*)z, (const T*)dp, scale, shift, mean, invstd, drop,
                           (const float*)nullptr, (const float*)nullptr, windows_per_tower, L, C, P, (T*)nullptr, part_dy,
                           part_dyz);
    }));
    return check_launch("vm_bn_pool_bwd_reduce");
}

extern "C" int vm_bn_bwd_finalize(const float* part_dy, const float* part_dyz, int64_t n_windows, int64_t windows_per_tower,
                                  int C, double count_per_tower, float* c1, float* c2, float* grad_gamma, float* grad_beta,
                                  void* stream) {
    VM_REQUIRE(part_dy && part_dyz && c1 && c2 && grad_gamma && grad_beta, "vm_bn_bwd_finalize: null pointer");
    VM_REQUIRE(n_windows > 0 && windows_per_tower > 0 && n_windows % windows_per_tower == 0,
               "vm_bn_bwd_finalize: n_windows must be a multiple of windows_per_tower");
    const int n_towers = (int)(n_windows / windows_per_tower);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 63) / 64), dim3(1024), 0, (hipStream_t)stream, part_dy, part_dyz,
                       windows_per_tower * BN_SEG, n_towers, C, count_per_tower, c1, c2, grad_gamma, grad_beta);
    return check_launch("vm_bn_bwd_finalize");
}

extern "C" int vm_bn_pool_bwd_apply(const void* z, const void* dp, const float* scale, const float* shift, const float* mean,
                                    const float* invstd, const float* drop, const float* c1, const float* c2, int64_t n_windows,
                                    int64_t windows_per_tower, int64_t L, int C, int pool, int dtype, void* du, float* part_du,
                                    void* stream) {
    VM_REQUIRE(z && dp && scale && shift && mean && invstd && c1 && c2 && du && part_du, "vm_bn_pool_bwd_apply: null pointer");
    VM_REQUIRE(n_windows > 0 && windows_per_tower > 0 && L >= pool && C % 8 == 0, "vm_bn_pool_bwd_apply: bad sizes");
    VM_DISPATCH_DTYPE(dtype, VM_DISPATCH_POOL(pool, {
        const int P = lanes_for(C / Elem<T>::kVec);
        hipLaunchKernelGGL((bn_pool_bwd_kernel<T, POOL, true>), dim3((unsigned)n_windows, BN_SEG), dim3(256), 0,
                           (hipStream_t)stream, (const T*)z, (const T*)dp, scale, shift, mean, invstd, drop, c1, c2,
                           windows_per_tower, L, C, P, (T*)du, part_du, (float*)nullptr);
    }));
    return check_launch("vm_bn_pool_bwd_apply");
}

extern "C" int vm_colsum(const float* part, int64_t rows, int C, float* out, void* stream) {
    VM_REQUIRE(part && out && rows > 0 && C > 0, "vm_colsum: bad argument");
    hipLaunchKernelGGL(colsum_kernel, dim3((C + 63) / 64), dim3(1024), 0, (hipStream_t)stream, part, rows, C, out);
    return check_launch("vm_colsum");
}
