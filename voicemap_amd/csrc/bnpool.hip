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
#include <mutex>
#include <type_traits>

#include "common.hpp"

#ifndef VM_APPLY_TWO_GROUPS
#define VM_APPLY_TWO_GROUPS 1   // experiment builds: tools/build_variant.sh <name> -DVM_APPLY_TWO_GROUPS=0 bnpool.hip
#endif

namespace vm {

#ifndef VM_BN_SEG
#define VM_BN_SEG 8
#endif
constexpr int BN_SEG = VM_BN_SEG;  // partial-sum rows per window (the layout of every part_* tensor)
// Workgroups per window of the pass kernels = gridDim.y: BN_SEG for the long windows of the 1-D encoder, 1 for short ones (the 2-D
// variant runs 16 384 windows of 149 pooled rows x 4 channel vectors: eight workgroups per window left 70 % of their threads without a
// row and the passes at 0.8-1.6 TB/s).  A launch with fewer segments zero-fills the partial rows it does not produce.
static int bn_segs(int64_t Lq, int C, int vec) { return Lq * (int64_t)(C / vec) >= 8 * 256 ? BN_SEG : 1; }

// VEC consecutive per-channel values (VEC = 4 or 8, 16-byte aligned) as 16-byte accesses: the kernels below run only a
// dozen loop iterations per thread on the last block, so 5 x 8 scalar parameter loads per thread were a visible cost.
template <int VEC>
__device__ inline void loadv(const float* p, float (&v)[VEC]) {
#pragma unroll
    for (int k = 0; k < VEC / 4; ++k) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p + 4 * k);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * k + e] = t[e];
    }
}
template <int VEC>
__device__ inline void loadvi(const int32_t* p, int (&v)[VEC]) {
#pragma unroll
    for (int k = 0; k < VEC / 4; ++k) {
        const u32x4 t = *reinterpret_cast<const u32x4*>(p + 4 * k);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * k + e] = (int)t[e];
    }
}
template <int VEC>
__device__ inline void storev(float* p, const float (&v)[VEC]) {
#pragma unroll
    for (int k = 0; k < VEC / 4; ++k) {
        const f32x4 t = {v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
        *reinterpret_cast<f32x4*>(p + 4 * k) = t;
    }
}
template <int VEC>
__device__ inline void ones(float (&v)[VEC]) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] = 1.0f;
}

// ---------------------------------------------------------------------------------------------
// Column sums of (rows, C) fp32 partial matrices, two deterministic stages so that the long reduction over
// rows is spread over many workgroups instead of C/64 of them:
//   stage 1: grid (C/64, segments*CR_CHUNKS); a workgroup (64 channels x 16 row lanes) sums one row chunk of
//            one segment (tower) in fp64 -> ws[(seg*CR_CHUNKS + chunk)][which][C]
//   stage 2: the consumer kernel adds the CR_CHUNKS doubles per (segment, channel) in a fixed (butterfly) order.
constexpr int CR_CHUNKS = 32;

// ---- stage 1 + stage 2 in ONE launch (round 5).  Every two-stage reduction below used to be a stage-1 launch and a 5 us "finalize"
// launch of a few workgroups -- 15-20 launches of a training step, each a dependent link of the step's chain (at the reference's batch
// sizes the step IS that chain: 62 kernels, 40 of them at the ~5 us floor of a dependent launch).  Now the stage-1 workgroups of a channel block take a
// ticket when their partial row is written; the LAST one to arrive runs the finalize body for the block's channels.  Visibility across
// the eight XCDs' L2s WITHOUT a device-scope fence (a release fence writes back the whole L2 -- measured: with __threadfence() in
// every stage-1 workgroup the cfg-A step went from 2.59 to 4.38 ms): the partials are stored with agent-scope (write-through, sc1)
// stores, every wave waits for its stores' acknowledgements (workgroup-scope release = s_waitcnt vmcnt(0)) before the workgroup's
// barrier, one relaxed agent-scope atomic takes the ticket, and the last arriver reads the partials with agent-scope loads.
// Same partials, same stage-2 butterfly: bit-identical to the two-launch form.
// Tickets: a library-owned pool of zero-initialised words; a launch takes a fresh range (round robin -- far more words than launches
// in flight), the last arriver puts its word back to zero.  The only mutable device state of the library.
// vm_set_tuning("fuse_finalize", mask): which two-stage reductions finish in their stage-1 launch -- bit 0 the BatchNorm statistics
// (vm_bn_finalize), bit 1 the BatchNorm-backward sums (vm_bn_bwd_finalize, vm_bn_bwd_from_sums_finalize), bit 2 plain column sums
// (vm_colsum*), bit 3 vm_du_tower_sums, bit 4: only where the reduction is small (C <= 128 and <= 4096 rows per tower); 0 = the
// two-launch form everywhere (the tests compare 0 with 15 bit for bit).  Measured (interleaved on one box, ms per step; cfg-A 128 pairs
// / cfg-B 32 pairs): bit 0 2.645 -> 2.667 / 0.503 -> 0.496; bit 1 2.650 -> 2.656 / 0.503 -> 0.506; bit 2 no change; bit 3 2.644 ->
// 2.669 / (not on its path): the last workgroup finalises 64 channels in two dependent passes of write-through loads where the
// finalize launch spreads them over C / 8 workgroups reading from L2 -- a launch saved (~5 us) only pays where the layer is narrow.
// Default: the statistics of narrow, short reductions (the reference's cfg-B at its own batch), nothing else.
int g_fuse_finalize = 17;
// Walk of the BatchNorm-backward apply pass over the windows (vm_set_tuning "apply_order"; see bn_pool_bwd_apply_kernel).  The memory-side
// cache (256 MB) keeps what was touched last: the dgrad that produced dp walked the windows front to back, so its last windows' dp rows
// (and the rows of the extreme it read beside them for the BatchNorm-backward sums) are still there when this pass starts --
// tools/probe/mall_recency_probe.py.
int g_apply_order = 2;   // interleaved A/B, one box: cfg-A 128 pairs 2.633 -> 2.626, 2.630 -> 2.627 ms; cfg-B 128 pairs 1.047 -> 1.041; ascending (1): 2.628 -> 2.640

constexpr int TICKET_WORDS = 16384;
__device__ unsigned g_tickets[TICKET_WORDS];
//  the symbol is resolved per device -- a process that drives several GPUs has one g_tickets per device -- and the cursor
// is advanced under a lock, so two host threads can never be handed overlapping words.
static unsigned* ticket_range(int n) {
    constexpr int MAX_DEV = 64;
    static std::mutex mu;
    static unsigned next[MAX_DEV] = {};
    static unsigned* base[MAX_DEV] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV || n > TICKET_WORDS) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (base[dev] == nullptr) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_tickets)) != hipSuccess) return nullptr;
        base[dev] = (unsigned*)p;
    }
    if (next[dev] + (unsigned)n > (unsigned)TICKET_WORDS) next[dev] = 0;
    unsigned* r = base[dev] + next[dev];
    next[dev] += (unsigned)n;
    return r;
}

// true in exactly one workgroup of the ``total`` that call it with this ticket: the last to arrive
__device__ inline bool last_arriver(unsigned* ticket, unsigned total) {
    __shared__ int s_last;
    // Every wave waits until its write-through (sc1) partial stores are ACKNOWLEDGED before the barrier that precedes the ticket.
    // A workgroup-scope release fence does not do that on gfx950 (the compiled code had the stores, s_barrier and the
    // ticket atomic back to back with no vmcnt wait, so the last arriver -- possibly on another XCD -- could read stale partials);
    // the explicit wait does, and it is all that is needed: the stores bypass this XCD's L2 write-back state (sc1), and a wait
    // for their acknowledgement costs no cache operation.  tests/test_isa_lint.py checks the instruction is in the code object.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == total - 1) ? 1 : 0;
        if (s_last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    return s_last != 0;
}
__device__ inline void store_agent(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline double load_agent(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT));
}
// the finalize body ``fin`` over the CL channels of channel block ``cb`` by the 1024 threads of the last workgroup: 32 groups of 32
// lanes, one channel per group and pass (the stage-2 butterflies stay inside a 32-lane group)
template <int CL, typename Fin>
__device__ inline void run_finalize(const Fin& fin, const double* ws, int cb, int C) {
#pragma unroll 1
    for (int cc = threadIdx.x >> 5; cc < CL; cc += 32) {
        const int c = cb * CL + cc;
        if (c < C) fin(ws, c, threadIdx.x & 31);
    }
}
struct FinNone {
    static constexpr bool kOn = false;
    __device__ inline void operator()(const double*, int, int) const {}
};

// CL channels x (1024 / CL) row lanes per workgroup: 64 x 16 for the wide layers, 32 x 32 where 64 lanes would be half empty
// (C = 32, 96: the 2-D variant, whose 131 K partial rows made this kernel 0.4 ms of its step)
template <int CL, typename Fin = FinNone>
__global__ __launch_bounds__(1024) void colreduce_stage1_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                 int64_t rows_per_seg, int C, double* ws, int row_step, Fin fin = Fin(),
                                                                 unsigned* tickets = nullptr) {
    // row_step > 1 (vm_colsum_strided): only every row_step-th row of the matrix is read (the others are known to be zero)
    constexpr int RG = 1024 / CL;
    __shared__ double red[2][RG][CL];
    const int cl = threadIdx.x % CL, rg = threadIdx.x / CL;
    const int c = blockIdx.x * CL + cl;
    const int seg = blockIdx.y / CR_CHUNKS, chunk = blockIdx.y % CR_CHUNKS;
    const int64_t per = (rows_per_seg + CR_CHUNKS - 1) / CR_CHUNKS;
    const int64_t r_lo = chunk * per;
    int64_t r_hi = r_lo + per;
    if (r_hi > rows_per_seg) r_hi = rows_per_seg;
    double s = 0.0, q = 0.0;
    if (c < C) {
        const int64_t base = (int64_t)seg * rows_per_seg;
        const int64_t rc = (int64_t)row_step * C;
        // four rows in flight per thread (a thread of the 2-D variant's launches walks 64 rows: one dependent load after the other
        // made this kernel 25-34 us for 17 MB), combined in a fixed order
        double s4[4] = {0.0, 0.0, 0.0, 0.0}, q4[4] = {0.0, 0.0, 0.0, 0.0};
        int64_t r = r_lo + rg;
        for (; r + 3 * RG < r_hi; r += 4 * RG) {
            float va[4], vb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                va[u] = a[(base + r + u * RG) * rc + c];
                vb[u] = b != nullptr ? b[(base + r + u * RG) * rc + c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s4[u] += (double)va[u];
                q4[u] += (double)vb[u];
            }
        }
        for (; r < r_hi; r += RG) {
            s4[0] += (double)a[(base + r) * rc + c];
            if (b != nullptr) q4[0] += (double)b[(base + r) * rc + c];
        }
        s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        q = (q4[0] + q4[1]) + (q4[2] + q4[3]);
    }
    red[0][rg][cl] = s;
    red[1][rg][cl] = q;
    __syncthreads();
    if (rg == 0 && c < C) {
        double ss = 0.0, qq = 0.0;
#pragma unroll
        for (int i = 0; i < RG; ++i) {
            ss += red[0][i][cl];
            qq += red[1][i][cl];
        }
        if constexpr (Fin::kOn) {
            store_agent(ws + ((int64_t)blockIdx.y * 2 + 0) * C + c, ss);
            store_agent(ws + ((int64_t)blockIdx.y * 2 + 1) * C + c, qq);
        } else {
            ws[((int64_t)blockIdx.y * 2 + 0) * C + c] = ss;
            ws[((int64_t)blockIdx.y * 2 + 1) * C + c] = qq;
        }
    }
    if constexpr (Fin::kOn) {
        if (last_arriver(tickets + blockIdx.x, gridDim.y)) run_finalize<CL>(fin, ws, blockIdx.x, C);
    }
}

static void launch_colreduce(const float* a, const float* b, int64_t rows_per_seg, int C, int segs, double* ws, hipStream_t st,
                             int row_step = 1) {
    if (C % 64 == 0 || C > 128) {
        hipLaunchKernelGGL((colreduce_stage1_kernel<64, FinNone>), dim3((C + 63) / 64, segs * CR_CHUNKS), dim3(1024), 0, st, a, b, rows_per_seg,
                           C, ws, row_step, FinNone(), (unsigned*)nullptr);
    } else {
        hipLaunchKernelGGL((colreduce_stage1_kernel<32, FinNone>), dim3((C + 31) / 32, segs * CR_CHUNKS), dim3(1024), 0, st, a, b, rows_per_seg,
                           C, ws, row_step, FinNone(), (unsigned*)nullptr);
    }
}
// ... with the finalize body run by the last workgroup of every channel block (one launch instead of two)
template <typename Fin>
static bool launch_colreduce_fin(const float* a, const float* b, int64_t rows_per_seg, int C, int segs, double* ws, hipStream_t st,
                                 const Fin& fin, int row_step = 1) {
    const bool wide = C % 64 == 0 || C > 128;
    const int blocks = wide ? (C + 63) / 64 : (C + 31) / 32;
    unsigned* t = ticket_range(blocks);
    if (t == nullptr) return false;
    if (wide) {
        hipLaunchKernelGGL((colreduce_stage1_kernel<64, Fin>), dim3(blocks, segs * CR_CHUNKS), dim3(1024), 0, st, a, b, rows_per_seg, C, ws,
                           row_step, fin, t);
    } else {
        hipLaunchKernelGGL((colreduce_stage1_kernel<32, Fin>), dim3(blocks, segs * CR_CHUNKS), dim3(1024), 0, st, a, b, rows_per_seg, C, ws,
                           row_step, fin, t);
    }
    return true;
}

// Stage 2 spread over 32 lanes (one partial each) + a fixed-order butterfly: the finalize kernels are latency-bound chains
// of dependent loads otherwise (8 us for 512 channels).  Thread layout of the callers: 8 channels x 32 lanes per workgroup.
template <bool SYNC = false>
__device__ inline void colreduce_stage2_par(const double* ws, int seg, int C, int c, int k, double& s, double& q) {
    static_assert(CR_CHUNKS == 32, "one lane per partial");
    // SYNC: the partials were written by OTHER workgroups of this launch (last-arriver finalize): agent-scope loads
    if (SYNC) {
        s = load_agent(ws + ((int64_t)(seg * CR_CHUNKS + k) * 2 + 0) * C + c);
        q = load_agent(ws + ((int64_t)(seg * CR_CHUNKS + k) * 2 + 1) * C + c);
    } else {
        s = ws[((int64_t)(seg * CR_CHUNKS + k) * 2 + 0) * C + c];
        q = ws[((int64_t)(seg * CR_CHUNKS + k) * 2 + 1) * C + c];
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        q += __shfl_xor(q, o, 64);
    }
}

template <bool SYNC>
struct FinBnBwd {
    static constexpr bool kOn = true;
    int n_towers, C;
    double count;
    float* c1;
    float* c2;
    float* grad_gamma;
    float* grad_beta;
    __device__ inline void operator()(const double* ws, int c, int k) const {
    double gg = 0.0, gb = 0.0;
    for (int tw = 0; tw < n_towers; ++tw) {
        double ss, qq;
        colreduce_stage2_par<SYNC>(ws, tw, C, c, k, ss, qq);
        if (k == 0) {
            c1[tw * C + c] = (float)(ss / count);
            c2[tw * C + c] = (float)(qq / count);
        }
        gb += ss;
        gg += qq;
    }
    if (k == 0) {
        grad_gamma[c] = (float)gg;
        grad_beta[c] = (float)gb;
    }
    }
};


// the finalize body as a functor: run by bn_finalize_kernel (its own launch: 8 channels x 32 lanes per workgroup) or, SYNC, by the last
// stage-1 workgroup of a channel block (launch_colreduce_fin)
template <bool SYNC>
struct FinBn {
    static constexpr bool kOn = true;
    int n_towers, C;
    double count;
    const float* gamma;
    const float* beta;
    float eps, momentum;
    int unbiased;
    float* moving_mean;
    float* moving_var;
    float* mean;
    float* invstd;
    float* scale;
    float* shift;
    float* zd_biased;
    float zd_correction;
    const float* center_bias;
    float* shift_adj;
    float* mean_adj;
    const float* tile_center;   // (n_towers, C): the statistics AND the stored extreme are centred by it (vm_conv_fwd_fold e_center)
    __device__ inline void operator()(const double* ws, int c, int k) const {
    float mm = 0.f, mv = 0.f;
    if (moving_mean != nullptr) {
        mm = moving_mean[c];
        mv = moving_var[c];
    }
    for (int tw = 0; tw < n_towers; ++tw) {
        double ss, qq;
        colreduce_stage2_par<SYNC>(ws, tw, C, c, k, ss, qq);
        if (k != 0) continue;
        if (tile_center != nullptr) {
            // the partial sums are over t = z - ctr (rows outside the window are zeros in both): sum z = sum t + n ctr,
            // sum z^2 = sum t^2 + 2 ctr sum t + n ctr^2
            const double cc = (double)tile_center[tw * C + c];
            qq += 2.0 * cc * ss + count * cc * cc;
            ss += count * cc;
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
        if (center_bias != nullptr) {
            // the block's pool extreme is stored as e - ctr, ctr = max(conv bias, 0) (vm_conv1_fused_fwd mode 2): the affine over the
            // STORED value is scale * e' + (shift + scale * ctr), and sum dp * e = sum dp * e' + ctr * sum dp puts ctr into the mean
            const float ctr = fmaxf(center_bias[c], 0.f);
            shift_adj[tw * C + c] = fmaf(sc, ctr, beta[c] - (float)m * sc);
            mean_adj[tw * C + c] = (float)(m - (double)ctr);
        } else if (tile_center != nullptr) {
            const float ctr = tile_center[tw * C + c];
            shift_adj[tw * C + c] = fmaf(sc, ctr, beta[c] - (float)m * sc);
            mean_adj[tw * C + c] = (float)(m - (double)ctr);
        }
        if (moving_mean != nullptr) {
            double vv = var;
            if (unbiased) vv = var * (count / (count - (1.0 + (double)eps)));
            if (zd_biased != nullptr) {
                // Keras 2.2.2 K.moving_average_update = tf.train.assign_moving_average(..., zero_debias=True) (TF 1.10
                // moving_averages._zero_debias): every encoder CALL (tower) keeps its own zero-initialised "biased" accumulator,
                //   biased -= (biased - value) * (1 - momentum);   moving = biased / (1 - momentum^local_step)
                // (zd_correction = 1 / (1 - momentum^t) from the host).  The towers' updates are applied in order: the moving
                // statistic ends up as the last tower's de-biased average.
                float* bm = zd_biased + ((int64_t)tw * 2 + 0) * C + c;
                float* bv = zd_biased + ((int64_t)tw * 2 + 1) * C + c;
                const float nbm = *bm - (*bm - (float)m) * (1.0f - momentum);
                const float nbv = *bv - (*bv - (float)vv) * (1.0f - momentum);
                *bm = nbm;
                *bv = nbv;
                mm = nbm * zd_correction;
                mv = nbv * zd_correction;
            } else {
                mm = mm - (mm - (float)m) * (1.0f - momentum);
                mv = mv - (mv - (float)vv) * (1.0f - momentum);
            }
        }
    }
    if (moving_mean != nullptr && k == 0) {
        moving_mean[c] = mm;
        moving_var[c] = mv;
    }
    }
};

__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ ws, FinBn<false> f) {
    const int c = blockIdx.x * 8 + (threadIdx.x >> 5), k = threadIdx.x & 31;
    if (c >= f.C) return;
    f(ws, c, k);
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
        loadv<VEC>(scale + tw * C + c0, sc);
        loadv<VEC>(shift + tw * C + c0, sh);
        if (drop) {
            loadv<VEC>(drop + n * C + c0, dr);
        } else {
            ones<VEC>(dr);
        }
        const T* zrow = z + n * L * C + c0;
        T* orow = out + (n * (Lq + 2) + 1) * C + c0;
        for (int64_t q = seg + (int64_t)rl * gridDim.y; q < Lq; q += (int64_t)RP * gridDim.y) {
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
// Last block: BatchNorm apply + dropout + max-pool + GlobalMaxPool1D in one pass over z (voicemap/models.py:31-37).  The
// pooled tensor of the last block has no other consumer (its BN backward works from z and the sparse (dg, gidx) form), so
// it is never written: every block keeps the running maximum of the storage-rounded pooled values of its pool groups
// (first maximum wins, as in vm_global_maxpool_fwd) and a second kernel reduces the BN_SEG segment partials.
template <typename T, int POOL>
__global__ __launch_bounds__(256) void bn_drop_pool_gmax_fwd_kernel(const T* __restrict__ z, const float* __restrict__ scale,
                                                                    const float* __restrict__ shift,
                                                                    const float* __restrict__ drop, int64_t wpt, int64_t L,
                                                                    int C, int P, float* __restrict__ part_v,
                                                                    int32_t* __restrict__ part_i, int64_t win_rows, int row0) {
    // win_rows / row0: a window holds win_rows rows of which the L read here start at row0 (L, 0: a plain tensor; L + 2, 1: a
    // padded pool extreme -- vm_bn_drop_pool_gmax_partials_e)
    constexpr int VEC = Elem<T>::kVec;
    __shared__ __attribute__((aligned(16))) float rv[256][VEC];
    __shared__ __attribute__((aligned(16))) int ri[256][VEC];
    const int tid = threadIdx.x;
    const int RP = 256 / P;
    const int pl = tid % P, rl = tid / P;
    const int CV = C / VEC;
    const int64_t n = blockIdx.x;
    const int seg = blockIdx.y;
    const int64_t tw = n / wpt;
    const int64_t Lq = L / POOL;
    for (int cvb = 0; cvb < CV; cvb += P) {
        const int cv = cvb + pl;
        const bool cok = cv < CV;
        const int c0 = cv * VEC;
        float best[VEC];
        int bi[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            best[i] = -INFINITY;
            bi[i] = 0x7fffffff;
        }
        if (cok) {
            float sc[VEC], sh[VEC], dr[VEC];
            loadv<VEC>(scale + tw * C + c0, sc);
            loadv<VEC>(shift + tw * C + c0, sh);
            if (drop) {
                loadv<VEC>(drop + n * C + c0, dr);
            } else {
                ones<VEC>(dr);
            }
            const T* zrow = z + (n * win_rows + row0) * C + c0;
            auto group = [&](int64_t q, const Vec16<T> (&v)[POOL]) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    float y = fmaf(v[0].get(i), sc[i], sh[i]) * dr[i];
#pragma unroll
                    for (int j = 1; j < POOL; ++j) {
                        const float yj = fmaf(v[j].get(i), sc[i], sh[i]) * dr[i];
                        y = yj > y ? yj : y;
                    }
                    y = Elem<T>::to_f(Elem<T>::from_f(y));  // the value the pooled tensor would have held
                    if (y > best[i] || bi[i] == 0x7fffffff) {
                        best[i] = y;
                        bi[i] = (int)q;
                    }
                }
            };
            const int64_t qs = (int64_t)RP * gridDim.y;
            int64_t q = seg + (int64_t)rl * gridDim.y;
            // two pool groups in flight per thread (as the apply pass: a pure stream over z), same groups in the same order
            for (; q + qs < Lq; q += 2 * qs) {
                Vec16<T> v0[POOL], v1[POOL];
#pragma unroll
                for (int j = 0; j < POOL; ++j) v0[j] = load16<T>(zrow + (q * POOL + j) * C);
#pragma unroll
                for (int j = 0; j < POOL; ++j) v1[j] = load16<T>(zrow + ((q + qs) * POOL + j) * C);
                group(q, v0);
                group(q + qs, v1);
            }
            for (; q < Lq; q += qs) {
                Vec16<T> v[POOL];
#pragma unroll
                for (int j = 0; j < POOL; ++j) v[j] = load16<T>(zrow + (q * POOL + j) * C);
                group(q, v);
            }
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            rv[tid][i] = best[i];
            ri[tid][i] = bi[i];
        }
        __syncthreads();
        if (rl == 0 && cok) {
            const int64_t row = n * BN_SEG + seg;
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float b = rv[pl][i];
                int k = ri[pl][i];
                for (int r = 1; r < RP; ++r) {
                    const float y = rv[r * P + pl][i];
                    const int kk = ri[r * P + pl][i];
                    if (kk != 0x7fffffff && (k == 0x7fffffff || y > b || (y == b && kk < k))) {
                        b = y;
                        k = kk;
                    }
                }
                part_v[row * C + c0 + i] = b;
                part_i[row * C + c0 + i] = k;
                for (int e = gridDim.y; seg == 0 && e < BN_SEG; ++e) {   // fewer segments than partial rows: the others are "empty"
                    part_v[(row + e) * C + c0 + i] = -INFINITY;
                    part_i[(row + e) * C + c0 + i] = 0x7fffffff;
                }
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void gmax_segments_kernel(const float* __restrict__ part_v, const int32_t* __restrict__ part_i,
                                                            int64_t n_windows, int C, float* __restrict__ gmax,
                                                            int32_t* __restrict__ gidx) {
    const int64_t e = blockIdx.x * 256LL + threadIdx.x;
    if (e >= n_windows * C) return;
    const int64_t n = e / C;
    const int c = (int)(e - n * C);
    float b = -INFINITY;
    int k = 0x7fffffff;
#pragma unroll
    for (int s = 0; s < BN_SEG; ++s) {
        const float y = part_v[(n * BN_SEG + s) * C + c];
        const int kk = part_i[(n * BN_SEG + s) * C + c];
        if (kk != 0x7fffffff && (k == 0x7fffffff || y > b || (y == b && kk < k))) {
            b = y;
            k = kk;
        }
    }
    gmax[e] = b;
    gidx[e] = k;
}

// ---------------------------------------------------------------------------------------------
// Apply pass of the BatchNorm / ReLU / max-pool backward.  grid = (n_windows, BN_SEG); a block owns pool groups
// q = seg, seg+BN_SEG, ... of one window.  threads: P lanes over channel vectors x RP row lanes.
//   du = [z>0] * (kc*z + kb + [arg] ka*dp),   ka = scale*drop, kb = scale*(invstd*c2*mean - c1), kc = -scale*invstd*c2
// (arg-max of y = (z*scale+shift)*drop over a pool window == arg-max of z if scale*drop >= 0, else arg-min; the first
// extreme wins).  Also emits the per-segment column sums of du (bias gradient of the convolution below).
// SP: dp is given in its sparse GlobalMaxPool1D-backward form -- dp[n][q][c] = sp_dg[n][c] if q == sp_idx[n][c] else 0 --
// instead of as a dense tensor (saves writing and re-reading it for the last block).
// PAIRS (16-bit storage, POOL 2, L even): z is given as the pool extreme e = ``z`` (PADDED, (n_windows, L/2 + 2, C)) and the other
// element of every pair ``zo`` (n_windows, L/2, C) with its sign bit set where the extreme is the pair's second element -- what
// vm_conv_fwd_fold leaves instead of z.
template <typename T, int POOL, bool SP, bool PAIRS = false>
__global__ __launch_bounds__(256) void bn_pool_bwd_apply_kernel(const T* __restrict__ z, const T* __restrict__ zo, const T* __restrict__ dp,
                                                                const float* __restrict__ scale, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, const float* __restrict__ drop,
                                                                const float* __restrict__ c1, const float* __restrict__ c2,
                                                                int64_t wpt, int64_t L, int C, int P, T* __restrict__ du,
                                                                float* __restrict__ part_a, const float* __restrict__ sp_dg,
                                                                const int32_t* __restrict__ sp_idx, const float* __restrict__ e_center,
                                                                int order) {
    constexpr int VEC = Elem<T>::kVec;
    __shared__ __attribute__((aligned(16))) float red[256][VEC];
    const int tid = threadIdx.x;
    const int RP = 256 / P;
    const int pl = tid % P, rl = tid / P;
    const int CV = C / VEC;
    // order 0: grid (windows, segments) -- the dispatcher walks the segments of ALL windows together; 1 / 2: grid (segments, windows),
    // window-major, ascending / DESCENDING: the workgroups resident at any time cover a contiguous range of windows, 2 starts with the
    // windows the producer of dp wrote LAST (g_apply_order).  Same blocks, same sums: bit-identical.
    const int64_t n = order == 0 ? blockIdx.x : (order == 1 ? blockIdx.y : gridDim.y - 1 - blockIdx.y);
    const int seg = order == 0 ? blockIdx.y : blockIdx.x;
    const int nseg = order == 0 ? gridDim.y : gridDim.x;
    const int64_t tw = n / wpt;
    const int64_t Lq = L / POOL;
    const int64_t Q = (L + POOL - 1) / POOL;  // also covers the remainder rows of a floor pool

    for (int cvb = 0; cvb < CV; cvb += P) {
        const int cv = cvb + pl;
        const bool cok = cv < CV;
        const int c0 = cv * VEC;
        float ka[VEC], kb[VEC], kc[VEC], acc[VEC], sgn[VEC], ectr[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            acc[i] = 0.f;
            ka[i] = kb[i] = kc[i] = 0.f;
            sgn[i] = 1.f;
            ectr[i] = 0.f;
        }
        // PAIRS with a centred extreme (vm_conv_fwd_fold e_center): e holds z - ctr, the other element z itself
        if (PAIRS && e_center != nullptr && cok) loadv<VEC>(e_center + tw * C + c0, ectr);
        if (cok) {
            float sc[VEC], mu[VEC], is[VEC], dr[VEC], k1[VEC], k2[VEC];
            loadv<VEC>(scale + tw * C + c0, sc);
            loadv<VEC>(mean + tw * C + c0, mu);
            loadv<VEC>(invstd + tw * C + c0, is);
            loadv<VEC>(c1 + tw * C + c0, k1);
            loadv<VEC>(c2 + tw * C + c0, k2);
            if (drop) {
                loadv<VEC>(drop + n * C + c0, dr);
            } else {
                ones<VEC>(dr);
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                sgn[i] = sc[i] * dr[i] < 0.f ? -1.f : 1.f;
                ka[i] = sc[i] * dr[i];
                kb[i] = sc[i] * (is[i] * k2[i] * mu[i] - k1[i]);
                kc[i] = -sc[i] * is[i] * k2[i];
            }
        }
        float spv[VEC];
        int spi[VEC];
        if constexpr (SP) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                spv[i] = 0.f;
                spi[i] = -1;
            }
            if (cok) {
                loadvi<VEC>(sp_idx + n * C + c0, spi);
                loadv<VEC>(sp_dg + n * C + c0, spv);
#pragma unroll
                for (int i = 0; i < VEC; ++i) spv[i] = Elem<T>::to_f(Elem<T>::from_f(spv[i]));  // same rounding as the dense dp tensor
            }
        }
        if (cok) {
            // FULL: a whole pool group with its dp row (every group when L % POOL == 0) -- no row-count predicates.
            // ``pre``: the group's (e, o, dp) vectors were loaded by the caller (the pair form keeps two groups in flight)
            auto body = [&](int64_t q, auto full_c, const Vec16<T>* pre) {
                constexpr bool FULL = decltype(full_c)::value;
                Vec16<T> zv[POOL];
                int nrows = POOL;
                if (!FULL && q * POOL + POOL > L) nrows = (int)(L - q * POOL);
                typedef uint16_t u16x8 __attribute__((ext_vector_type(8)));
                u16x8 second = (u16x8)0;   // PAIRS: 1 where the extreme is the pair's second element
                if constexpr (PAIRS && sizeof(T) == 2 && POOL == 2) {
                    const Vec16<T> ev = pre ? pre[0] : load16<T>(z + (n * (Lq + 2) + 1 + q) * C + c0);
                    const Vec16<T> ow = pre ? pre[1] : load16<T>(zo + (n * Lq + q) * C + c0);
                    const u16x8 eb = __builtin_bit_cast(u16x8, ev.v), ob = __builtin_bit_cast(u16x8, ow.v);
                    second = ob >> 15;
                    const u16x8 m = (u16x8)0 - (ob >> 15), oa = ob & (uint16_t)0x7fff;  // m: 0xffff where the extreme is element 1
                    zv[0].v = __builtin_bit_cast(decltype(zv[0].v), (u16x8)((eb & ~m) | (oa & m)));
                    zv[1].v = __builtin_bit_cast(decltype(zv[1].v), (u16x8)((oa & ~m) | (eb & m)));
                } else {
                    const T* zp = z + (n * L + q * POOL) * C + c0;
#pragma unroll
                    for (int j = 0; j < POOL; ++j)
                        if (FULL || j < nrows) zv[j] = pre ? pre[j] : load16<T>(zp + j * C);
                }
                const bool has_dp = FULL || q < Lq;
                Vec16<T> dv;
                if (!SP && has_dp) dv = pre ? pre[2] : load16<T>(dp + (n * Lq + q) * C + c0);
                Vec16<T> ov[POOL];
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    float zj[POOL];
#pragma unroll
                    for (int j = 0; j < POOL; ++j) zj[j] = (FULL || j < nrows) ? zv[j].get(i) : 0.f;
                    if constexpr (PAIRS && POOL == 2) {   // the stored extreme back to z (exact where relu clipped: -ctr + ctr)
                        const bool s2 = second[i] != 0;
                        zj[0] += s2 ? 0.f : ectr[i];
                        zj[1] += s2 ? ectr[i] : 0.f;
                    }
                    float ext = sgn[i] * zj[0];
                    int arg = 0;
#pragma unroll
                    for (int j = 1; j < POOL; ++j) {
                        const float y = sgn[i] * zj[j];
                        if ((FULL || j < nrows) && y > ext) {  // strict: the first extreme wins
                            ext = y;
                            arg = j;
                        }
                    }
                    float dpv = 0.f;
                    if (has_dp) dpv = SP ? (spi[i] == (int)q ? spv[i] : 0.f) : dv.get(i);
                    const float ady = ka[i] * dpv;
#pragma unroll
                    for (int j = 0; j < POOL; ++j) {
                        float g = fmaf(kc[i], zj[j], kb[i]) + (j == arg ? ady : 0.f);
                        g = zj[j] > 0.f ? g : 0.f;
                        ov[j].set(i, g);
                        if (FULL || j < nrows) acc[i] += ov[j].get(i);
                    }
                }
                T* op = du + (n * (L + 2) + 1 + q * POOL) * C + c0;
#pragma unroll
                for (int j = 0; j < POOL; ++j)
                    if (FULL || j < nrows) store16<T>(op + j * C, ov[j]);
            };
            int64_t q = seg + (int64_t)rl * nseg;
            const int64_t qs = (int64_t)RP * nseg;
            if constexpr (VM_APPLY_TWO_GROUPS && POOL == 2 && !(PAIRS && sizeof(T) != 2)) {
                // NG pool groups in flight per thread: 3 NG 16-byte loads before the first is consumed (the pass is a pure stream:
                // 4.5 TB/s with three).  Same groups in the same order: bit-identical
                constexpr int NG = VM_APPLY_TWO_GROUPS == 1 ? 2 : VM_APPLY_TWO_GROUPS;
                for (; q + (NG - 1) * qs < Lq; q += NG * qs) {
                    Vec16<T> a[NG][3];
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        if constexpr (PAIRS) {
                            a[g][0] = load16<T>(z + (n * (Lq + 2) + 1 + q + g * qs) * C + c0);
                            a[g][1] = load16<T>(zo + (n * Lq + q + g * qs) * C + c0);
                        } else {   // the dense / sparse-dp forms: the group's two z rows
                            a[g][0] = load16<T>(z + (n * L + (q + g * qs) * 2) * C + c0);
                            a[g][1] = load16<T>(z + (n * L + (q + g * qs) * 2 + 1) * C + c0);
                        }
                        if constexpr (!SP) a[g][2] = load16<T>(dp + (n * Lq + q + g * qs) * C + c0);
                    }
#pragma unroll
                    for (int g = 0; g < NG; ++g) body(q + g * qs, std::true_type{}, a[g]);
                }
            }
            for (; q < Lq; q += qs) body(q, std::true_type{}, nullptr);
            if (q < Q) body(q, std::false_type{}, nullptr);  // the remainder rows of a floor pool (q == Lq)
        }
        // reduce over the RP row lanes
#pragma unroll
        for (int i = 0; i < VEC; ++i) red[tid][i] = acc[i];
        __syncthreads();
        if (rl == 0 && cok) {
            const int64_t row = n * BN_SEG + seg;
            float a[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) a[i] = 0.f;
            for (int r = 0; r < RP; ++r) {
                float t[VEC];
                loadv<VEC>(&red[r * P + pl][0], t);
#pragma unroll
                for (int i = 0; i < VEC; ++i) a[i] += t[i];
            }
            storev<VEC>(part_a + row * C + c0, a);
            if (seg == 0 && nseg < BN_SEG) {   // fewer segments than partial rows: the others are zero
#pragma unroll
                for (int i = 0; i < VEC; ++i) a[i] = 0.f;
                for (int e = nseg; e < BN_SEG; ++e) storev<VEC>(part_a + (row + e) * C + c0, a);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Reduce pass, dense dp.  Same grid/thread mapping as above.  The per-channel constants are factored OUT of the loop:
//   sum dy      = drop * S0,                       S0 = sum dp
//   sum dy*zhat = drop*invstd * (S1 - mean*S0),    S1 = sum dp * ext(z)     (ext = max or min over the pool window, by the
//                                                                            sign of scale*drop, as in the forward pass)
// so the loop carries 2 accumulators + a sign per channel (the general kernel above carries 5 + the sparse operands and
// runs at 4 waves per SIMD); two pool groups are loaded before either is consumed.  Read-only stream: z + dp.
template <typename T, int POOL>
__global__ __launch_bounds__(256) void bn_pool_bwd_reduce_kernel(const T* __restrict__ z, const T* __restrict__ dp,
                                                                 const float* __restrict__ scale, const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, const float* __restrict__ drop,
                                                                 int64_t wpt, int64_t L, int C, int P,
                                                                 float* __restrict__ part_a, float* __restrict__ part_b,
                                                                 const T* __restrict__ act, const float* __restrict__ shift) {
    // act != nullptr (vm_bn_pool_bwd_reduce_pooled): the pool-window extreme of z is recovered from the POOLED forward output
    // instead of being re-derived from z:  act = round(drop * (scale * ext + shift))  =>  ext = act / (scale*drop) - shift/scale.
    // The pass then reads two pooled-size tensors (act, dp) instead of z + dp.  Channels with scale == 0 (the map is not
    // invertible) fall back to z for their whole 8-channel vector.
    constexpr int VEC = Elem<T>::kVec;
    __shared__ __attribute__((aligned(16))) float red[2][256][VEC];
    const int tid = threadIdx.x;
    const int RP = 256 / P;
    const int pl = tid % P, rl = tid / P;
    const int CV = C / VEC;
    const int64_t n = blockIdx.x;
    const int seg = blockIdx.y;
    const int64_t tw = n / wpt;
    const int64_t Lq = L / POOL;
    const int64_t stride = (int64_t)RP * gridDim.y;
    for (int cvb = 0; cvb < CV; cvb += P) {
        const int cv = cvb + pl;
        const bool cok = cv < CV;
        const int c0 = cv * VEC;
        float sgn[VEC], s0[VEC], s1[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            s0[i] = 0.f;
            s1[i] = 0.f;
            sgn[i] = 1.f;
        }
        bool pooled = false;
        float ra[VEC], rb[VEC];
        if (cok) {
            float sc[VEC], dr[VEC];
            loadv<VEC>(scale + tw * C + c0, sc);
            if (drop) {
                loadv<VEC>(drop + n * C + c0, dr);
            } else {
                ones<VEC>(dr);
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) sgn[i] = sc[i] * dr[i] < 0.f ? -1.f : 1.f;
            if (act != nullptr) {
                float sh[VEC];
                loadv<VEC>(shift + tw * C + c0, sh);
                pooled = true;
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    if (sc[i] == 0.f && dr[i] != 0.f) pooled = false;
                    const bool live = sc[i] != 0.f && dr[i] != 0.f;  // drop == 0: dy == 0, the channel contributes nothing
                    ra[i] = live ? 1.0f / (sc[i] * dr[i]) : 0.f;
                    rb[i] = live ? -sh[i] / sc[i] : 0.f;
                }
            }
        }
        if (cok && pooled) {
            const T* ab = act + (n * (Lq + 2) + 1) * C + c0;
            const T* db = dp + n * Lq * C + c0;
            auto consume_p = [&](const Vec16<T>& pv, const Vec16<T>& dv) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const float d = dv.get(i);
                    s0[i] += d;
                    s1[i] = fmaf(d, fmaf(pv.get(i), ra[i], rb[i]), s1[i]);
                }
            };
            int64_t q = seg + (int64_t)rl * gridDim.y;
            for (; q + stride < Lq; q += 2 * stride) {
                const Vec16<T> pa = load16<T>(ab + q * C), da = load16<T>(db + q * C);
                const Vec16<T> pc = load16<T>(ab + (q + stride) * C), dc = load16<T>(db + (q + stride) * C);
                consume_p(pa, da);
                consume_p(pc, dc);
            }
            if (q < Lq) {
                const Vec16<T> pa = load16<T>(ab + q * C), da = load16<T>(db + q * C);
                consume_p(pa, da);
            }
        } else if (cok) {
            const T* zb = z + n * L * C + c0;
            const T* db = dp + n * Lq * C + c0;
            auto consume = [&](const Vec16<T> (&zv)[POOL], const Vec16<T>& dv) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    float m = sgn[i] * zv[0].get(i);
#pragma unroll
                    for (int j = 1; j < POOL; ++j) {
                        const float y = sgn[i] * zv[j].get(i);
                        m = y > m ? y : m;
                    }
                    const float d = dv.get(i);
                    s0[i] += d;
                    s1[i] = fmaf(d, sgn[i] * m, s1[i]);
                }
            };
            int64_t q = seg + (int64_t)rl * gridDim.y;
            for (; q + stride < Lq; q += 2 * stride) {
                Vec16<T> za[POOL], zc[POOL];
#pragma unroll
                for (int j = 0; j < POOL; ++j) za[j] = load16<T>(zb + (q * POOL + j) * C);
                const Vec16<T> da = load16<T>(db + q * C);
#pragma unroll
                for (int j = 0; j < POOL; ++j) zc[j] = load16<T>(zb + ((q + stride) * POOL + j) * C);
                const Vec16<T> dc = load16<T>(db + (q + stride) * C);
                consume(za, da);
                consume(zc, dc);
            }
            if (q < Lq) {
                Vec16<T> za[POOL];
#pragma unroll
                for (int j = 0; j < POOL; ++j) za[j] = load16<T>(zb + (q * POOL + j) * C);
                const Vec16<T> da = load16<T>(db + q * C);
                consume(za, da);
            }
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            red[0][tid][i] = s0[i];
            red[1][tid][i] = s1[i];
        }
        __syncthreads();
        if (rl == 0 && cok) {
            const int64_t row = n * BN_SEG + seg;
            float a[VEC], b[VEC], dr[VEC], mu[VEC], is[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) a[i] = b[i] = 0.f;
            for (int r = 0; r < RP; ++r) {
                float ta[VEC], tb[VEC];
                loadv<VEC>(&red[0][r * P + pl][0], ta);
                loadv<VEC>(&red[1][r * P + pl][0], tb);
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    a[i] += ta[i];
                    b[i] += tb[i];
                }
            }
            loadv<VEC>(mean + tw * C + c0, mu);
            loadv<VEC>(invstd + tw * C + c0, is);
            if (drop) {
                loadv<VEC>(drop + n * C + c0, dr);
            } else {
                ones<VEC>(dr);
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                b[i] = dr[i] * is[i] * (b[i] - mu[i] * a[i]);
                a[i] = dr[i] * a[i];
            }
            storev<VEC>(part_a + row * C + c0, a);
            storev<VEC>(part_b + row * C + c0, b);
            if (seg == 0 && gridDim.y < BN_SEG) {   // fewer segments than partial rows: the others are zero
#pragma unroll
                for (int i = 0; i < VEC; ++i) a[i] = 0.f;
                for (int e = gridDim.y; e < BN_SEG; ++e) {
                    storev<VEC>(part_a + (row + e) * C + c0, a);
                    storev<VEC>(part_b + (row + e) * C + c0, a);
                }
            }
        }
        __syncthreads();
    }
}

// The two BatchNorm-backward sums when S0 = sum dp and SA = sum dp * A were already formed per (window, channel) by the dgrad
// epilogue of the layer above (vm_conv_dgrad_bnred: `rows` partial rows per window).  A is either this block's pooled output
// (a_is_act: ext = A / (scale*drop) - shift/scale as in bn_pool_bwd_reduce_kernel, so S1 = ra*SA + rb*S0) or the pool extreme
// itself (block 1: S1 = SA).  One thread per (window, channel); writes all BN_SEG partial rows of the window (row 0 = value).
// A channel whose scale is exactly 0 has no invertible map: its S1 is re-derived from z here (one slow thread, never in practice).
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_from_sums_kernel(const float* __restrict__ s0p, const float* __restrict__ sap, int rows,
                                                               const T* __restrict__ z, const T* __restrict__ dp,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               const float* __restrict__ mean, const float* __restrict__ invstd,
                                                               const float* __restrict__ drop, int64_t n_windows, int64_t wpt, int64_t L,
                                                               int C, int pool, int a_is_act, float* __restrict__ part_a,
                                                               float* __restrict__ part_b) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_windows * C) return;
    const int64_t n = idx / C;
    const int c = (int)(idx - n * C);
    const int64_t tw = n / wpt;
    float S0 = 0.f, SA = 0.f;
    for (int r = 0; r < rows; ++r) {
        S0 += s0p[(n * rows + r) * C + c];
        SA += sap[(n * rows + r) * C + c];
    }
    const float sc = scale[tw * C + c], dr = drop ? drop[n * C + c] : 1.f;
    float S1 = SA;
    if (a_is_act) {
        if (sc != 0.f && dr != 0.f) {
            S1 = fmaf(SA, 1.0f / (sc * dr), (-shift[tw * C + c] / sc) * S0);
        } else if (dr != 0.f) {  // scale == 0: the forward kept the maximum of z
            const int64_t Lq = L / pool;
            S1 = 0.f;
            for (int64_t q = 0; q < Lq; ++q) {
                float m = Elem<T>::to_f(z[(n * L + q * pool) * C + c]);
                for (int j = 1; j < pool; ++j) {
                    const float y = Elem<T>::to_f(z[(n * L + q * pool + j) * C + c]);
                    m = y > m ? y : m;
                }
                S1 = fmaf(Elem<T>::to_f(dp[(n * Lq + q) * C + c]), m, S1);
            }
        } else {
            S1 = 0.f;
        }
    }
    const float a = dr * S0, b = dr * invstd[tw * C + c] * (S1 - mean[tw * C + c] * S0);
    for (int s = 0; s < BN_SEG; ++s) {
        part_a[(n * BN_SEG + s) * C + c] = s == 0 ? a : 0.f;
        part_b[(n * BN_SEG + s) * C + c] = s == 0 ? b : 0.f;
    }
}

// vm_bn_bwd_from_sums + the first stage of vm_bn_bwd_finalize in ONE launch (the backward chain of a block is a string of small
// dependent kernels, each worth its ~4.5 us of launch latency with the chip idle): the per-(window, channel) map of
// bn_bwd_from_sums_kernel is linear in (S0, SA), so it is applied to every partial row of the dgrad epilogue and the results go
// straight into the fp64 column sums of colreduce_stage1_kernel's layout (grid (C / 64, towers * CR_CHUNKS), 64 channels x 16 row
// lanes; a chunk = a range of windows of one tower, its (window, partial row) items dealt round-robin to the row lanes).  The
// scale == 0 channel (no invertible map: S1 re-derived from z) is handled once per window by the lane that owns its row 0.
template <typename T>
__global__ __launch_bounds__(1024) void bn_bwd_sums_stage1_kernel(const float* __restrict__ s0p, const float* __restrict__ sap, int rows,
                                                                  const T* __restrict__ z, const T* __restrict__ dp,
                                                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                                                  const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                  const float* __restrict__ drop, int64_t wpt, int64_t L, int C, int pool,
                                                                  int a_is_act, double* ws, FinBnBwd<true> fin, unsigned* tickets) {
    __shared__ double red[2][16][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int tw = blockIdx.y / CR_CHUNKS, chunk = blockIdx.y % CR_CHUNKS;
    const int64_t per = (wpt + CR_CHUNKS - 1) / CR_CHUNKS;
    const int64_t w_lo = chunk * per;
    int64_t w_hi = w_lo + per;
    if (w_hi > wpt) w_hi = wpt;
    double sa_ = 0.0, sb_ = 0.0;
    if (c < C && w_hi > w_lo) {
        const float sc = scale[tw * C + c], sh = shift[tw * C + c], mu = mean[tw * C + c], is = invstd[tw * C + c];
        const int64_t items = (w_hi - w_lo) * rows;
        for (int64_t it = rg; it < items; it += 16) {
            const int64_t wl = it / rows;
            const int r = (int)(it - wl * rows);
            const int64_t n = (int64_t)tw * wpt + w_lo + wl;
            const float dr = drop ? drop[n * C + c] : 1.f;
            const float S0 = s0p[(n * rows + r) * C + c], SA = sap[(n * rows + r) * C + c];
            float S1 = SA;
            if (a_is_act) {
                if (sc != 0.f && dr != 0.f) {
                    S1 = fmaf(SA, 1.0f / (sc * dr), (-sh / sc) * S0);
                } else if (dr != 0.f) {  // scale == 0: the forward kept the maximum of z; the whole window once, with its row 0
                    S1 = 0.f;
                    if (r == 0) {
                        const int64_t Lq = L / pool;
                        for (int64_t q = 0; q < Lq; ++q) {
                            float m = Elem<T>::to_f(z[(n * L + q * pool) * C + c]);
                            for (int j = 1; j < pool; ++j) {
                                const float y = Elem<T>::to_f(z[(n * L + q * pool + j) * C + c]);
                                m = y > m ? y : m;
                            }
                            S1 = fmaf(Elem<T>::to_f(dp[(n * Lq + q) * C + c]), m, S1);
                        }
                    }
                } else {
                    S1 = 0.f;
                }
            }
            sa_ += (double)(dr * S0);
            sb_ += (double)(dr * is * (S1 - mu * S0));
        }
    }
    red[0][rg][cl] = sa_;
    red[1][rg][cl] = sb_;
    __syncthreads();
    if (rg == 0 && c < C) {
        double ss = 0.0, qq = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            ss += red[0][i][cl];
            qq += red[1][i][cl];
        }
        store_agent(ws + ((int64_t)blockIdx.y * 2 + 0) * C + c, ss);
        store_agent(ws + ((int64_t)blockIdx.y * 2 + 1) * C + c, qq);
    }
    // (tickets NULL: the two-launch form, vm_set_tuning("fuse_finalize", 0))
    if (tickets != nullptr && last_arriver(tickets + blockIdx.x, gridDim.y)) run_finalize<64>(fin, ws, blockIdx.x, C);
}

// Reduce pass when dp is the sparse GlobalMaxPool1D-backward form: dy is non-zero at ONE pool group per (window, channel),
// so the two sums need z at that group only -- a gather of n*C*POOL elements instead of a pass over z.
// Writes all BN_SEG partial rows of a window (row 0 = the value, the others 0).
template <typename T, int POOL>
__global__ __launch_bounds__(256) void bn_pool_bwd_reduce_gmax_kernel(const T* __restrict__ z, const float* __restrict__ dg,
                                                                      const int32_t* __restrict__ gidx,
                                                                      const float* __restrict__ scale, const float* __restrict__ mean,
                                                                      const float* __restrict__ invstd, const float* __restrict__ drop,
                                                                      int64_t n_windows, int64_t wpt, int64_t L, int C,
                                                                      float* __restrict__ part_a, float* __restrict__ part_b) {
    const int64_t e = blockIdx.x * 256LL + threadIdx.x;
    if (e >= n_windows * C) return;
    const int64_t n = e / C;
    const int c = (int)(e - n * C);
    const int64_t tw = n / wpt;
    const int64_t Lq = L / POOL;
    const int q = gidx[e];
    float a = 0.f, b = 0.f;
    if (q >= 0 && q < Lq) {
        const float dr = drop ? drop[e] : 1.0f;
        const float sc = scale[tw * C + c], mu = mean[tw * C + c], is = invstd[tw * C + c];
        const bool use_min = sc * dr < 0.f;
        const T* zp = z + (n * L + (int64_t)q * POOL) * C + c;
        float ext = Elem<T>::to_f(zp[0]);
#pragma unroll
        for (int j = 1; j < POOL; ++j) {
            const float zj = Elem<T>::to_f(zp[(int64_t)j * C]);
            if (use_min ? (zj < ext) : (zj > ext)) ext = zj;
        }
        const float d = Elem<T>::to_f(Elem<T>::from_f(dg[e]));  // same rounding as the dense dp tensor
        a = dr * d;
        b = dr * is * d * (ext - mu);
    }
#pragma unroll
    for (int s = 0; s < BN_SEG; ++s) {
        part_a[(n * BN_SEG + s) * C + c] = s == 0 ? a : 0.f;
        part_b[(n * BN_SEG + s) * C + c] = s == 0 ? b : 0.f;
    }
}

// The sparse reduce above AND its finalize in one launch (round 6): the sums of the last block run over n_windows x C gathered values
// -- 128 per (tower, channel) at the bench batch -- which one 32-lane group adds in fp64 by itself: three dependent launches
// (gather, column reduction, finalize) of the forward -> backward turn-around become one.  Same per-element arithmetic as
// bn_pool_bwd_reduce_gmax_kernel, fp64 sums of the same fp32 terms (the order differs from the two-stage form; the fp64 sum of a few
// hundred floats rounds to the same float).
template <typename T, int POOL>
__global__ __launch_bounds__(256) void bn_bwd_gmax_finalize_kernel(const T* __restrict__ z, const float* __restrict__ dg,
                                                                   const int32_t* __restrict__ gidx, const float* __restrict__ scale,
                                                                   const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                   const float* __restrict__ drop, int64_t n_windows, int64_t wpt, int64_t L,
                                                                   FinBnBwd<false> f, int64_t win_rows, int row0) {
    const int c = blockIdx.x * 8 + (threadIdx.x >> 5), k = threadIdx.x & 31;
    const int C = f.C;
    if (c >= C) return;
    const int64_t Lq = L / POOL;
    double gg = 0.0, gb = 0.0;
    // (round 6, last session) eight windows per lane in flight -- four of each of (up to) two towers: the index, gradient and mask loads
    // of all of them first, then the gathers they address, then the sums in the same order as before (bit-identical).  One window at a
    // time was two DEPENDENT memory round trips per window, eight windows per lane at the bench batch, on the step's forward -> backward
    // turn-around.  Measured: 22.4 -> 19.2 us, with four or with eight in flight -- the rest is not the lanes' dependent loads (262 k
    // gathers of one 64-byte line each out of a 197 MB tensor: address translation)
    constexpr int U = 4, TW = 2;
    for (int tw0 = 0; tw0 < f.n_towers; tw0 += TW) {
        double sa[TW] = {0.0, 0.0}, sb[TW] = {0.0, 0.0};
        float sc[TW], mu[TW], is[TW];
#pragma unroll
        for (int t = 0; t < TW; ++t) {
            const int tw = tw0 + t < f.n_towers ? tw0 + t : tw0;
            sc[t] = scale[tw * C + c];
            mu[t] = mean[tw * C + c];
            is[t] = invstd[tw * C + c];
        }
        for (int64_t w0 = k; w0 < wpt; w0 += 32 * U) {
            int q[TW][U];
            float d[TW][U], dr[TW][U], zv[TW][U][POOL];
#pragma unroll
            for (int t = 0; t < TW; ++t)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int64_t w = w0 + 32 * u;
                    const bool live = w < wpt && tw0 + t < f.n_towers;
                    const int64_t e = ((tw0 + (tw0 + t < f.n_towers ? t : 0)) * wpt + (w < wpt ? w : w0)) * C + c;   // (clamped: the loads are unconditional)
                    q[t][u] = gidx[e];
                    d[t][u] = dg[e];
                    dr[t][u] = drop ? drop[e] : 1.0f;
                    if (!live) q[t][u] = -1;
                }
#pragma unroll
            for (int t = 0; t < TW; ++t)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool ok = q[t][u] >= 0 && q[t][u] < Lq;
                    const int64_t n = (tw0 + (tw0 + t < f.n_towers ? t : 0)) * wpt + (w0 + 32 * u < wpt ? w0 + 32 * u : w0);
                    const T* zp = z + (n * win_rows + row0 + (int64_t)(ok ? q[t][u] : 0) * POOL) * C + c;
#pragma unroll
                    for (int j = 0; j < POOL; ++j) zv[t][u][j] = Elem<T>::to_f(zp[(int64_t)j * C]);
                }
#pragma unroll
            for (int t = 0; t < TW; ++t)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (q[t][u] >= 0 && q[t][u] < Lq) {
                        const bool use_min = sc[t] * dr[t][u] < 0.f;
                        float ext = zv[t][u][0];
#pragma unroll
                        for (int j = 1; j < POOL; ++j) {
                            const float zj = zv[t][u][j];
                            if (use_min ? (zj < ext) : (zj > ext)) ext = zj;
                        }
                        const float dd = Elem<T>::to_f(Elem<T>::from_f(d[t][u]));  // same rounding as the dense dp tensor
                        sa[t] += (double)(dr[t][u] * dd);
                        sb[t] += (double)(dr[t][u] * is[t] * dd * (ext - mu[t]));
                    }
                }
        }
#pragma unroll
        for (int t = 0; t < TW; ++t) {
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) {
                sa[t] += __shfl_xor(sa[t], o, 64);
                sb[t] += __shfl_xor(sb[t], o, 64);
            }
            if (tw0 + t < f.n_towers) {
                if (k == 0) {
                    f.c1[(tw0 + t) * C + c] = (float)(sa[t] / f.count);
                    f.c2[(tw0 + t) * C + c] = (float)(sb[t] / f.count);
                }
                gb += sa[t];
                gg += sb[t];
            }
        }
    }
    if (k == 0) {
        f.grad_gamma[c] = (float)gg;
        f.grad_beta[c] = (float)gb;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* __restrict__ ws, FinBnBwd<false> f) {
    const int c = blockIdx.x * 8 + (threadIdx.x >> 5), k = threadIdx.x & 31;
    if (c >= f.C) return;
    f(ws, c, k);
}

template <bool SYNC>
struct FinColsum {
    static constexpr bool kOn = true;
    int C;
    float* out;
    __device__ inline void operator()(const double* ws, int c, int k) const {
        double ss, qq;
        colreduce_stage2_par<SYNC>(ws, 0, C, c, k, ss, qq);
        if (k == 0) out[c] = (float)ss;
    }
};

__global__ __launch_bounds__(256) void colsum_finalize_kernel(const double* __restrict__ ws, FinColsum<false> f) {
    const int c = blockIdx.x * 8 + (threadIdx.x >> 5), k = threadIdx.x & 31;
    if (c >= f.C) return;
    f(ws, c, k);
}

// vm_du_tower_sums: per tower t the column sums of du (from the apply pass's partial rows, via colreduce_stage1_kernel with one segment
// per tower) and the three tap sums D_t[k][c] a folded weight gradient needs: the sum over the positions whose tap k lies inside the
// window -- all of them for k = 1, all but position 0 for k = 0, all but position L - 1 for k = 2 (du: padded (n_windows, L + 2, C)).
template <typename T, bool SYNC>
struct FinDuTower {
    static constexpr bool kOn = true;
    const T* du;
    int towers;
    int64_t wpt, L;
    int C;
    float* grad_b;
    float* dsum;
    __device__ inline void operator()(const double* ws, int c, int k) const {
    double gb = 0.0;
    for (int t = 0; t < towers; ++t) {
        double ss, qq;
        colreduce_stage2_par<SYNC>(ws, t, C, c, k, ss, qq);
        double e0 = 0.0, e1 = 0.0;
        for (int64_t w = k; w < wpt; w += 32) {
            const T* row = du + ((int64_t)(t * wpt + w) * (L + 2)) * C + c;
            e0 += (double)(float)row[C];
            e1 += (double)(float)row[L * C];
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            e0 += __shfl_xor(e0, o, 64);
            e1 += __shfl_xor(e1, o, 64);
        }
        if (k == 0) {
            dsum[((int64_t)t * 3 + 0) * C + c] = (float)(ss - e0);
            dsum[((int64_t)t * 3 + 1) * C + c] = (float)ss;
            dsum[((int64_t)t * 3 + 2) * C + c] = (float)(ss - e1);
        }
        gb += ss;
    }
    if (k == 0 && grad_b != nullptr) grad_b[c] = (float)gb;
    }
};

template <typename T>
__global__ __launch_bounds__(256) void du_tower_sums_kernel(const double* __restrict__ ws, FinDuTower<T, false> f) {
    const int c = blockIdx.x * 8 + (threadIdx.x >> 5), k = threadIdx.x & 31;
    if (c >= f.C) return;
    f(ws, c, k);
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
                              float* scale, float* shift, void* ws, float* zd_biased, float zd_correction, const float* center_bias,
                              float* shift_adj, float* mean_adj, const float* tile_center, void* stream) {
    VM_REQUIRE(stat_sum && stat_sq && gamma && beta && mean && invstd && scale && shift && ws, "vm_bn_finalize: null pointer");
    VM_REQUIRE((center_bias == nullptr && tile_center == nullptr) || (shift_adj && mean_adj), "vm_bn_finalize: center_bias / tile_center need shift_adj and mean_adj");
    VM_REQUIRE(center_bias == nullptr || tile_center == nullptr, "vm_bn_finalize: center_bias and tile_center are exclusive");
    VM_REQUIRE((moving_mean == nullptr) == (moving_var == nullptr), "vm_bn_finalize: moving stats must both be set or NULL");
    VM_REQUIRE(rows_per_tower > 0 && n_towers > 0 && C > 0 && count_per_tower > 1.0, "vm_bn_finalize: bad sizes");
    VM_REQUIRE(zd_biased == nullptr || (moving_mean != nullptr && zd_correction >= 1.0f), "vm_bn_finalize: zero-debias needs the moving statistics and a correction >= 1");
    // stage 1 of the column sums, the statistics by the last workgroup of every channel block: one launch
    const FinBn<true> fin{n_towers, C, count_per_tower, gamma, beta, eps, momentum, unbiased_moving_var, moving_mean, moving_var,
                          mean, invstd, scale, shift, zd_biased, zd_correction, center_bias, shift_adj, mean_adj, tile_center};
    const bool small = C <= 128 && rows_per_tower <= 4096;
    if (!(g_fuse_finalize & 1) || ((g_fuse_finalize & 16) && !small) || !launch_colreduce_fin(stat_sum, stat_sq, rows_per_tower, C, n_towers, (double*)ws, (hipStream_t)stream, fin)) {
        launch_colreduce(stat_sum, stat_sq, rows_per_tower, C, n_towers, (double*)ws, (hipStream_t)stream);
        const FinBn<false> f2{n_towers, C, count_per_tower, gamma, beta, eps, momentum, unbiased_moving_var, moving_mean, moving_var,
                              mean, invstd, scale, shift, zd_biased, zd_correction, center_bias, shift_adj, mean_adj, tile_center};
        hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 7) / 8), dim3(256), 0, (hipStream_t)stream, (const double*)ws, f2);
    }
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
        hipLaunchKernelGGL((bn_drop_pool_fwd_kernel<T, POOL>), dim3((unsigned)n_windows, (unsigned)bn_segs(L / POOL, C, Elem<T>::kVec)), dim3(256), 0,
                           (hipStream_t)stream, (const T*)z, scale, shift, drop, windows_per_tower, L, C, P, (T*)out);
    }));
    return check_launch("vm_bn_drop_pool_fwd");
}

extern "C" int64_t vm_bn_drop_pool_gmax_workspace_bytes(int64_t n_windows, int C) {
    return n_windows * BN_SEG * (int64_t)C * 8;
}

extern "C" int vm_bn_drop_pool_gmax_fwd(const void* z, const float* scale, const float* shift, const float* drop,
                                        int64_t n_windows, int64_t windows_per_tower, int64_t L, int C, int pool, int dtype,
                                        float* gmax, int32_t* gidx, void* ws, void* stream) {
    VM_REQUIRE(z && scale && shift && gmax && gidx && ws, "vm_bn_drop_pool_gmax_fwd: null pointer");
    VM_REQUIRE(n_windows > 0 && windows_per_tower > 0 && L >= pool && C % 8 == 0, "vm_bn_drop_pool_gmax_fwd: bad sizes");
    float* part_v = (float*)ws;
    int32_t* part_i = (int32_t*)(part_v + n_windows * BN_SEG * (int64_t)C);
    VM_DISPATCH_DTYPE(dtype, VM_DISPATCH_POOL(pool, {
        const int P = lanes_for(C / Elem<T>::kVec);
        hipLaunchKernelGGL((bn_drop_pool_gmax_fwd_kernel<T, POOL>), dim3((unsigned)n_windows, (unsigned)bn_segs(L / POOL, C, Elem<T>::kVec)), dim3(256), 0,
                           (hipStream_t)stream, (const T*)z, scale, shift, drop, windows_per_tower, L, C, P, part_v, part_i, L, 0);
    }));
    hipLaunchKernelGGL(gmax_segments_kernel, dim3((unsigned)((n_windows * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       part_v, part_i, n_windows, C, gmax, gidx);
    return check_launch("vm_bn_drop_pool_gmax_fwd");
}

// the first launch of vm_bn_drop_pool_gmax_fwd alone: the BN_SEG partial (value, position) rows per window, into two caller-given
// arrays (so that two launches -- the two towers on their streams -- can fill the halves of one pair); vm_tail_fwd_bwd finishes them
extern "C" int vm_bn_drop_pool_gmax_partials(const void* z, const float* scale, const float* shift, const float* drop,
                                             int64_t n_windows, int64_t windows_per_tower, int64_t L, int C, int pool, int dtype,
                                             float* part_v, int32_t* part_i, void* stream) {
    VM_REQUIRE(z && scale && shift && part_v && part_i, "vm_bn_drop_pool_gmax_partials: null pointer");
    VM_REQUIRE(n_windows > 0 && windows_per_tower > 0 && L >= pool && C % 8 == 0, "vm_bn_drop_pool_gmax_partials: bad sizes");
    VM_DISPATCH_DTYPE(dtype, VM_DISPATCH_POOL(pool, {
        const int P = lanes_for(C / Elem<T>::kVec);
        hipLaunchKernelGGL((bn_drop_pool_gmax_fwd_kernel<T, POOL>), dim3((unsigned)n_windows, (unsigned)bn_segs(L / POOL, C, Elem<T>::kVec)), dim3(256), 0,
                           (hipStream_t)stream, (const T*)z, scale, shift, drop, windows_per_tower, L, C, P, part_v, part_i, L, 0);
    }));
    return check_launch("vm_bn_drop_pool_gmax_partials");
}

extern "C" int vm_bn_part_rows(void) { return BN_SEG; }

extern "C" int vm_bn_pool_bwd_reduce(const void* z, const void* dp, const float* scale, const float* shift, const float* mean,
                                     const float* invstd, const float* drop, int64_t n_windows, int64_t windows_per_tower,
                                     int64_t L, int C, int pool, int dtype, float* part_dy, float* part_dyz, void* stream) {
    VM_REQUIRE(z && dp && scale && shift && mean && invstd && part_dy && part_dyz, "vm_bn_pool_bwd_reduce: null pointer");
    VM_REQUIRE(n_windows > 0 && windows_per_tower > 0 && L >= pool && C % 8 == 0, "vm_bn_pool_bwd_reduce: bad sizes");
    VM_DISPATCH_DTYPE(dtype, VM_DISPATCH_POOL(pool, {
        const int P = lanes_for(C / Elem<T>::kVec);
        hipLaunchKernelGGL((bn_pool_bwd_reduce_kernel<T, POOL>), dim3((unsigned)n_windows, (unsigned)bn_segs(L / POOL, C, Elem<T>::kVec)), dim3(256), 0,
                           (hipStream_t)stream, (const T*)z, (const T*)dp, scale, mean, invstd, drop, windows_per_tower, L, C, P,
                           part_dy, part_dyz, (const T*)nullptr, (const float*)nullptr);
    }));
    return check_launch("vm_bn_pool_bwd_reduce");
}

extern "C" int vm_bn_pool_bwd_reduce_pooled(const void* z, const void* act, const void* dp, const float* scale, const float* shift,
                                            const float* mean, const float* invstd, const float* drop, int64_t n_windows,
                                            int64_t windows_per_tower, int64_t L, int C, int pool, int dtype, float* part_dy,
                                            float* part_dyz, void* stream) {
    VM_REQUIRE(z && act && dp && scale && shift && mean && invstd && part_dy && part_dyz,
               "vm_bn_pool_bwd_reduce_pooled: null pointer");
    VM_REQUIRE(n_windows > 0 && windows_per_tower > 0 && L >= pool && C % 8 == 0, "vm_bn_pool_bwd_reduce_pooled: bad sizes");
    VM_DISPATCH_DTYPE(dtype, VM_DISPATCH_POOL(pool, {
        const int P = lanes_for(C / Elem<T>::kVec);
        hipLaunchKernelGGL((bn_pool_bwd_reduce_kernel<T, POOL>), dim3((unsigned)n_windows, (unsigned)bn_segs(L / POOL, C, Elem<T>::kVec)), dim3(256), 0,
                           (hipStream_t)stream, (const T*)z, (const T*)dp, scale, mean, invstd, drop, windows_per_tower, L, C, P,
                           part_dy, part_dyz, (const T*)act, shift);
    }));
    return check_launch("vm_bn_pool_bwd_reduce_pooled");
}

extern "C" int vm_bn_pool_bwd_reduce_gmax(const void* z, const float* dg, const int32_t* gidx, const float* scale,
                                          const float* shift, const float* mean, const float* invstd, const float* drop,
                                          int64_t n_windows, int64_t windows_per_tower, int64_t L, int C, int pool, int dtype,
                                          float* part_dy, float* part_dyz, void* stream) {
    VM_REQUIRE(z && dg && gidx && scale && shift && mean && invstd && part_dy && part_dyz,
               "vm_bn_pool_bwd_reduce_gmax: null pointer");
    VM_REQUIRE(n_windows > 0 && windows_per_tower > 0 && L >= pool && C % 8 == 0, "vm_bn_pool_bwd_reduce_gmax: bad sizes");
    VM_DISPATCH_DTYPE(dtype, VM_DISPATCH_POOL(pool, {
        const int64_t blocks = (n_windows * C + 255) / 256;
        hipLaunchKernelGGL((bn_pool_bwd_reduce_gmax_kernel<T, POOL>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                           (const T*)z, dg, gidx, scale, mean, invstd, drop, n_windows, windows_per_tower, L, C, part_dy,
                           part_dyz);
    }));
    return check_launch("vm_bn_pool_bwd_reduce_gmax");
}

extern "C" int vm_bn_bwd_gmax_finalize(const void* z, const float* dg, const int32_t* gidx, const float* scale, const float* shift,
                                       const float* mean, const float* invstd, const float* drop, int64_t n_windows,
                                       int64_t windows_per_tower, int64_t L, int C, int pool, int dtype, double count_per_tower, float* c1,
                                       float* c2, float* grad_gamma, float* grad_beta, void* stream) {
    VM_REQUIRE(z && dg && gidx && scale && shift && mean && invstd && c1 && c2 && grad_gamma && grad_beta,
               "vm_bn_bwd_gmax_finalize: null pointer");
    VM_REQUIRE(n_windows > 0 && windows_per_tower > 0 && n_windows % windows_per_tower == 0 && L >= pool && C % 8 == 0,
               "vm_bn_bwd_gmax_finalize: bad sizes (n_windows a multiple of windows_per_tower, C a multiple of 8)");
    const int n_towers = (int)(n_windows / windows_per_tower);
    const FinBnBwd<false> fin{n_towers, C, count_per_tower, c1, c2, grad_gamma, grad_beta};
    VM_DISPATCH_DTYPE(dtype, VM_DISPATCH_POOL(pool, {
        hipLaunchKernelGGL((bn_bwd_gmax_finalize_kernel<T, POOL>), dim3((unsigned)((C + 7) / 8)), dim3(256), 0, (hipStream_t)stream,
                           (const T*)z, dg, gidx, scale, mean, invstd, drop, n_windows, windows_per_tower, L, fin, L, 0);
    }));
    return check_launch("vm_bn_bwd_gmax_finalize");
}

extern "C" int vm_bn_bwd_from_sums(const float* s0, const float* sa, int64_t rows_per_window, const void* z, const void* dp,
                                   const float* scale, const float* shift, const float* mean, const float* invstd,
                                   const float* drop, int64_t n_windows, int64_t windows_per_tower, int64_t L, int C, int pool,
                                   int dtype, int a_is_act, float* part_dy, float* part_dyz, void* stream) {
    VM_REQUIRE(s0 && sa && dp && scale && shift && mean && invstd && part_dy && part_dyz, "vm_bn_bwd_from_sums: null pointer");
    VM_REQUIRE(!a_is_act || z, "vm_bn_bwd_from_sums: z is needed next to the pooled output (channels with scale == 0)");
    VM_REQUIRE(n_windows > 0 && windows_per_tower > 0 && rows_per_window > 0 && pool >= 1 && L >= pool && C > 0,
               "vm_bn_bwd_from_sums: bad sizes");
    VM_DISPATCH_DTYPE(dtype, {
        const int64_t blocks = (n_windows * C + 255) / 256;
        hipLaunchKernelGGL((bn_bwd_from_sums_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, s0, sa,
                           (int)rows_per_window, (const T*)z, (const T*)dp, scale, shift, mean, invstd, drop, n_windows,
                           windows_per_tower, L, C, pool, a_is_act, part_dy, part_dyz);
    });
    return check_launch("vm_bn_bwd_from_sums");
}

extern "C" int vm_bn_bwd_finalize(const float* part_dy, const float* part_dyz, int64_t n_windows, int64_t windows_per_tower,
                                  int C, double count_per_tower, float* c1, float* c2, float* grad_gamma, float* grad_beta,
                                  void* ws, void* stream) {
    VM_REQUIRE(part_dy && part_dyz && c1 && c2 && grad_gamma && grad_beta && ws, "vm_bn_bwd_finalize: null pointer");
    VM_REQUIRE(n_windows > 0 && windows_per_tower > 0 && n_windows % windows_per_tower == 0,
               "vm_bn_bwd_finalize: n_windows must be a multiple of windows_per_tower");
    const int n_towers = (int)(n_windows / windows_per_tower);
    const FinBnBwd<true> fin{n_towers, C, count_per_tower, c1, c2, grad_gamma, grad_beta};
    if (!(g_fuse_finalize & 2) ||
        !launch_colreduce_fin(part_dy, part_dyz, windows_per_tower * BN_SEG, C, n_towers, (double*)ws, (hipStream_t)stream, fin)) {
        launch_colreduce(part_dy, part_dyz, windows_per_tower * BN_SEG, C, n_towers, (double*)ws, (hipStream_t)stream);
        const FinBnBwd<false> f2{n_towers, C, count_per_tower, c1, c2, grad_gamma, grad_beta};
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 7) / 8), dim3(256), 0, (hipStream_t)stream, (const double*)ws, f2);
    }
    return check_launch("vm_bn_bwd_finalize");
}

extern "C" int vm_bn_bwd_from_sums_finalize(const float* s0, const float* sa, int64_t rows_per_window, const void* z, const void* dp,
                                            const float* scale, const float* shift, const float* mean, const float* invstd,
                                            const float* drop, int64_t n_windows, int64_t windows_per_tower, int64_t L, int C, int pool,
                                            int dtype, int a_is_act, double count_per_tower, float* c1, float* c2, float* grad_gamma,
                                            float* grad_beta, void* ws, void* stream) {
    VM_REQUIRE(s0 && sa && dp && scale && shift && mean && invstd && c1 && c2 && grad_gamma && grad_beta && ws,
               "vm_bn_bwd_from_sums_finalize: null pointer");
    VM_REQUIRE(!a_is_act || z, "vm_bn_bwd_from_sums_finalize: z is needed next to the pooled output (channels with scale == 0)");
    VM_REQUIRE(n_windows > 0 && windows_per_tower > 0 && n_windows % windows_per_tower == 0 && rows_per_window > 0 && pool >= 1 && L >= pool &&
                   C > 0,
               "vm_bn_bwd_from_sums_finalize: bad sizes");
    const int n_towers = (int)(n_windows / windows_per_tower);
    unsigned* tickets = (g_fuse_finalize & 2) ? ticket_range((C + 63) / 64) : nullptr;
    const FinBnBwd<true> fin{n_towers, C, count_per_tower, c1, c2, grad_gamma, grad_beta};
    VM_DISPATCH_DTYPE(dtype, {
        hipLaunchKernelGGL((bn_bwd_sums_stage1_kernel<T>), dim3((C + 63) / 64, n_towers * CR_CHUNKS), dim3(1024), 0, (hipStream_t)stream, s0,
                           sa, (int)rows_per_window, (const T*)z, (const T*)dp, scale, shift, mean, invstd, drop, windows_per_tower, L, C, pool,
                           a_is_act, (double*)ws, fin, tickets);
    });
    if (tickets == nullptr) {
        const FinBnBwd<false> f2{n_towers, C, count_per_tower, c1, c2, grad_gamma, grad_beta};
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 7) / 8), dim3(256), 0, (hipStream_t)stream, (const double*)ws, f2);
    }
    return check_launch("vm_bn_bwd_from_sums_finalize");
}

static dim3 apply_grid(int64_t n_windows, int segs) {
    return g_apply_order == 0 ? dim3((unsigned)n_windows, (unsigned)segs) : dim3((unsigned)segs, (unsigned)n_windows);
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
        if (sp_idx != nullptr) {
            hipLaunchKernelGGL((bn_pool_bwd_apply_kernel<T, POOL, true>), apply_grid(n_windows, bn_segs(L / POOL, C, Elem<T>::kVec)), dim3(256), 0,
                               (hipStream_t)stream, (const T*)z, (const T*)nullptr, (const T*)dp, scale, mean, invstd, drop, c1, c2, windows_per_tower,
                               L, C, P, (T*)du, part_du, sp_dg, sp_idx, (const float*)nullptr, g_apply_order);
        } else {
            hipLaunchKernelGGL((bn_pool_bwd_apply_kernel<T, POOL, false>), apply_grid(n_windows, bn_segs(L / POOL, C, Elem<T>::kVec)), dim3(256), 0,
                               (hipStream_t)stream, (const T*)z, (const T*)nullptr, (const T*)dp, scale, mean, invstd, drop, c1, c2, windows_per_tower,
                               L, C, P, (T*)du, part_du, sp_dg, sp_idx, (const float*)nullptr, g_apply_order);
        }
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

extern "C" int vm_bn_pool_bwd_apply_pairs(const void* e, const void* o, const void* dp, const float* scale, const float* shift,
                                          const float* mean, const float* invstd, const float* drop, const float* c1, const float* c2,
                                          int64_t n_windows, int64_t windows_per_tower, int64_t L, int C, int dtype, void* du,
                                          float* part_du, const float* e_center, void* stream) {
    VM_REQUIRE(e && o && dp && scale && shift && mean && invstd && c1 && c2 && du && part_du, "vm_bn_pool_bwd_apply_pairs: null pointer");
    VM_REQUIRE(n_windows > 0 && windows_per_tower > 0 && L >= 2 && !(L & 1) && C % 8 == 0, "vm_bn_pool_bwd_apply_pairs: L must be even, C % 8 == 0");
    VM_REQUIRE(dtype == VM_BF16 || dtype == VM_F16, "vm_bn_pool_bwd_apply_pairs: 16-bit storage only (VM_BF16 / VM_F16), got dtype %d", dtype);
    VM_DISPATCH_16(dtype, {
        const int P = lanes_for(C / Elem<T>::kVec);
        hipLaunchKernelGGL((bn_pool_bwd_apply_kernel<T, 2, false, true>), apply_grid(n_windows, bn_segs(L / 2, C, Elem<T>::kVec)), dim3(256), 0,
                           (hipStream_t)stream, (const T*)e, (const T*)o, (const T*)dp, scale, mean, invstd, drop, c1, c2, windows_per_tower,
                           L, C, P, (T*)du, part_du, (const float*)nullptr, (const int32_t*)nullptr, e_center, g_apply_order);
    });
    return check_launch("vm_bn_pool_bwd_apply_pairs");
}

extern "C" int vm_bn_pool_bwd_apply_gmax(const void* z, const float* dg, const int32_t* gidx, const float* scale,
                                         const float* shift, const float* mean, const float* invstd, const float* drop,
                                         const float* c1, const float* c2, int64_t n_windows, int64_t windows_per_tower, int64_t L,
                                         int C, int pool, int dtype, void* du, float* part_du, void* stream) {
    VM_REQUIRE(dg && gidx, "vm_bn_pool_bwd_apply_gmax: null pointer");
    return bn_pool_bwd_apply_impl(z, nullptr, dg, gidx, scale, shift, mean, invstd, drop, c1, c2, n_windows, windows_per_tower, L,
                                  C, pool, dtype, du, part_du, stream);
}

// ---- the LAST block in pair form (round 6): its forward leaves (e, o) like blocks 2..n-1 instead of z, the GlobalMaxPool1D pass reads
// the padded extreme e alone -- max_j fma(z_j, s, h) == fma(ext_j z, s, h): half the bytes of z on the forward -> backward turn-around --
// and the backward's sparse sums and apply pass take (e, o).  Same values, same positions as the z forms.
extern "C" int vm_bn_drop_pool_gmax_partials_e(const void* e, const float* scale, const float* shift, const float* drop, int64_t n_windows,
                                               int64_t windows_per_tower, int64_t Lq, int C, int dtype, float* part_v, int32_t* part_i,
                                               void* stream) {
    VM_REQUIRE(e && scale && shift && part_v && part_i, "vm_bn_drop_pool_gmax_partials_e: null pointer");
    VM_REQUIRE(n_windows > 0 && windows_per_tower > 0 && Lq >= 1 && C % 8 == 0, "vm_bn_drop_pool_gmax_partials_e: bad sizes");
    VM_REQUIRE(dtype == VM_BF16 || dtype == VM_F16, "vm_bn_drop_pool_gmax_partials_e: 16-bit storage only");
    VM_DISPATCH_16(dtype, {
        const int P = lanes_for(C / Elem<T>::kVec);
        hipLaunchKernelGGL((bn_drop_pool_gmax_fwd_kernel<T, 1>), dim3((unsigned)n_windows, (unsigned)bn_segs(Lq, C, Elem<T>::kVec)), dim3(256), 0,
                           (hipStream_t)stream, (const T*)e, scale, shift, drop, windows_per_tower, Lq, C, P, part_v, part_i, Lq + 2, 1);
    });
    return check_launch("vm_bn_drop_pool_gmax_partials_e");
}

extern "C" int vm_bn_bwd_gmax_finalize_e(const void* e, const float* dg, const int32_t* gidx, const float* scale, const float* shift,
                                         const float* mean, const float* invstd, const float* drop, int64_t n_windows,
                                         int64_t windows_per_tower, int64_t Lq, int C, int dtype, double count_per_tower, float* c1,
                                         float* c2, float* grad_gamma, float* grad_beta, void* stream) {
    VM_REQUIRE(e && dg && gidx && scale && shift && mean && invstd && c1 && c2 && grad_gamma && grad_beta,
               "vm_bn_bwd_gmax_finalize_e: null pointer");
    VM_REQUIRE(n_windows > 0 && windows_per_tower > 0 && n_windows % windows_per_tower == 0 && Lq >= 1 && C % 8 == 0,
               "vm_bn_bwd_gmax_finalize_e: bad sizes (n_windows a multiple of windows_per_tower, C a multiple of 8)");
    VM_REQUIRE(dtype == VM_BF16 || dtype == VM_F16, "vm_bn_bwd_gmax_finalize_e: 16-bit storage only");
    const int n_towers = (int)(n_windows / windows_per_tower);
    const FinBnBwd<false> fin{n_towers, C, count_per_tower, c1, c2, grad_gamma, grad_beta};
    VM_DISPATCH_16(dtype, {
        hipLaunchKernelGGL((bn_bwd_gmax_finalize_kernel<T, 1>), dim3((unsigned)((C + 7) / 8)), dim3(256), 0, (hipStream_t)stream,
                           (const T*)e, dg, gidx, scale, mean, invstd, drop, n_windows, windows_per_tower, Lq, fin, Lq + 2, 1);
    });
    return check_launch("vm_bn_bwd_gmax_finalize_e");
}

extern "C" int vm_bn_pool_bwd_apply_pairs_gmax(const void* e, const void* o, const float* dg, const int32_t* gidx, const float* scale,
                                               const float* shift, const float* mean, const float* invstd, const float* drop,
                                               const float* c1, const float* c2, int64_t n_windows, int64_t windows_per_tower, int64_t L,
                                               int C, int dtype, void* du, float* part_du, void* stream) {
    VM_REQUIRE(e && o && dg && gidx && scale && shift && mean && invstd && c1 && c2 && du && part_du,
               "vm_bn_pool_bwd_apply_pairs_gmax: null pointer");
    VM_REQUIRE(n_windows > 0 && windows_per_tower > 0 && L >= 2 && !(L & 1) && C % 8 == 0,
               "vm_bn_pool_bwd_apply_pairs_gmax: L must be even, C % 8 == 0");
    VM_REQUIRE(dtype == VM_BF16 || dtype == VM_F16, "vm_bn_pool_bwd_apply_pairs_gmax: 16-bit storage only");
    VM_DISPATCH_16(dtype, {
        const int P = lanes_for(C / Elem<T>::kVec);
        hipLaunchKernelGGL((bn_pool_bwd_apply_kernel<T, 2, true, true>), apply_grid(n_windows, bn_segs(L / 2, C, Elem<T>::kVec)), dim3(256), 0,
                           (hipStream_t)stream, (const T*)e, (const T*)o, (const T*)nullptr, scale, mean, invstd, drop, c1, c2, windows_per_tower,
                           L, C, P, (T*)du, part_du, dg, gidx, (const float*)nullptr, g_apply_order);
    });
    return check_launch("vm_bn_pool_bwd_apply_pairs_gmax");
}

extern "C" int vm_du_tower_sums(const float* part_du, const void* du, int64_t n_windows, int64_t windows_per_tower, int64_t L, int C,
                                int dtype, float* grad_b, float* dsum, void* ws, void* stream) {
    VM_REQUIRE(part_du && du && dsum && ws, "vm_du_tower_sums: null pointer");
    VM_REQUIRE(n_windows > 0 && windows_per_tower > 0 && n_windows % windows_per_tower == 0 && L > 0 && C > 0,
               "vm_du_tower_sums: n_windows must be a positive multiple of windows_per_tower");
    const int towers = (int)(n_windows / windows_per_tower);
    VM_DISPATCH_DTYPE(dtype, {
        const FinDuTower<T, true> fin{(const T*)du, towers, windows_per_tower, L, C, grad_b, dsum};
        if (!(g_fuse_finalize & 8) ||
            !launch_colreduce_fin(part_du, nullptr, windows_per_tower * (int64_t)BN_SEG, C, towers, (double*)ws, (hipStream_t)stream, fin)) {
            launch_colreduce(part_du, nullptr, windows_per_tower * (int64_t)BN_SEG, C, towers, (double*)ws, (hipStream_t)stream);
            const FinDuTower<T, false> f2{(const T*)du, towers, windows_per_tower, L, C, grad_b, dsum};
            hipLaunchKernelGGL((du_tower_sums_kernel<T>), dim3((C + 7) / 8), dim3(256), 0, (hipStream_t)stream, (const double*)ws, f2);
        }
    });
    return check_launch("vm_du_tower_sums");
}

extern "C" int vm_bn_part_rows_used(int64_t L, int C, int pool, int dtype) {
    const int vec = (dtype == VM_F32 || dtype == VM_F32S) ? 4 : 8;
    return (pool < 1 || C < vec) ? BN_SEG : bn_segs(L / pool, C, vec);
}

extern "C" int vm_colsum_strided(const float* part, int64_t rows, int row_step, int C, float* out, void* ws, void* stream) {
    VM_REQUIRE(part && out && ws && rows > 0 && row_step >= 1 && C > 0, "vm_colsum_strided: bad argument");
    if (!(g_fuse_finalize & 4) || !launch_colreduce_fin(part, nullptr, rows, C, 1, (double*)ws, (hipStream_t)stream, FinColsum<true>{C, out}, row_step)) {
        launch_colreduce(part, nullptr, rows, C, 1, (double*)ws, (hipStream_t)stream, row_step);
        hipLaunchKernelGGL(colsum_finalize_kernel, dim3((C + 7) / 8), dim3(256), 0, (hipStream_t)stream, (const double*)ws, FinColsum<false>{C, out});
    }
    return check_launch("vm_colsum_strided");
}

extern "C" int vm_colsum(const float* part, int64_t rows, int C, float* out, void* ws, void* stream) {
    VM_REQUIRE(part && out && ws && rows > 0 && C > 0, "vm_colsum: bad argument");
    if (!(g_fuse_finalize & 4) || !launch_colreduce_fin(part, nullptr, rows, C, 1, (double*)ws, (hipStream_t)stream, FinColsum<true>{C, out})) {
        launch_colreduce(part, nullptr, rows, C, 1, (double*)ws, (hipStream_t)stream);
        hipLaunchKernelGGL(colsum_finalize_kernel, dim3((C + 7) / 8), dim3(256), 0, (hipStream_t)stream, (const double*)ws, FinColsum<false>{C, out});
    }
    return check_launch("vm_colsum");
}
