// Block 1 of the encoder, fused for 16-bit storage (bf16 / f16; voicemap/models.py:13-19: Conv1D(F,32,'same',relu) ->
// BatchNormalization -> SpatialDropout1D -> MaxPool1D).
//
// C_in = 1 makes this block 4 % of the FLOPs but the largest activation (B x 12000 x F): writing relu(conv) at
// full resolution and streaming it three more times (BN+pool, BN-backward reduce, BN-backward apply) cost more
// HBM time than the whole rest of the network.  Here the full-resolution tensor never exists in HBM:
//
//   forward : MFMA conv tile -> +bias, ReLU -> per-channel sum / sum-of-squares partials (BN statistics)
//             -> MaxPool *in registers*.  BN is a per-channel affine applied after the statistics are known and
//             max-pooling commutes with a monotone map:  max_j(a*z_j + b) = a*max_j z_j + b for a >= 0 and
//             a*min_j z_j + b for a < 0 (sign(a) = sign(gamma), known before the launch).  So only the pooled
//             extreme of z (1/pool of the bytes) is stored; the affine + dropout run over the pooled tensor.
//   backward: the conv tile is *recomputed* from the waveform (32 taps, trivial on the matrix cores), the BN/ReLU/pool
//             backward is evaluated in registers and the result du feeds the weight-gradient MFMA directly as its
//             B operand (the 32x32 accumulator layout of du IS a valid K-slot assignment for the next MFMA, as in
//             flash-attention's P.V step), with the matching waveform samples gathered from LDS as the A operand.
//
// Both kernels are VALU-bound (the matrix pipe idles most of the time), so the per-element instruction count is what is
// tuned here: packed fp32 adds/fmas, wave-uniform "every channel pools a maximum" paths without min/max selects, pair-wise
// bf16 converts, scalar tile addressing, and accumulators kept in VGPRs (build.py: -amdgpu-mfma-vgpr-form) so the
// epilogue reads them without v_accvgpr_read copies.
//
// MFMA operands are bf16, but the waveform and the filters are split hi+lo (x = xh + xl, both bf16) and the three
// significant products accumulated (xh*wh + xl*wh + xh*wl), so block 1 keeps ~16 mantissa bits of its fp32 inputs.
// The waveform tile lives in LDS as 8 sample-shifted copies so that every 8-sample (16-byte) fragment of the
// sliding window  x[p+k .. p+k+8)  is an aligned ds_read_b128 whatever p is.
#include <type_traits>

#include "common.hpp"

// experiment builds (tools/build_variant.sh <name> -DVM_C1_ABL=<bits> conv1_fused.hip; results are wrong by design): 1 no pooled-tensor
// traffic (forward: no stores; backward: no dp loads), 2 no VALU epilogue, 4 no convolution MFMAs / LDS fragment reads, 8 (backward)
// no weight-gradient MFMAs / LDS gathers.
#ifndef VM_C1_ABL
#define VM_C1_ABL 0
#endif

namespace vm {

#if defined(VM_EXPERIMENT_PROFILE)  // where a wave of the block-1 kernels spends its clocks (s_memtime); experiment builds only
constexpr int PROF1_SLOTS = 4096 * 4;
__device__ unsigned int g_prof1[PROF1_SLOTS * 8];
#define VM_PROF1(...) __VA_ARGS__
#else
#define VM_PROF1(...)
#endif

constexpr int F1_K = 32;       // taps
constexpr int F1_CHUNK = 256;  // positions per chunk (8 MFMA row tiles of 32)
constexpr int F1_CPL = 288;    // samples a shifted copy must hold (>= 256 + 31, multiple of 8)
constexpr int F1_PAD = 8;      // slack either side of a copy: the 8 shifted stores of one sample need no range predicate
constexpr int F1_ROW = F1_CPL + 2 * F1_PAD;  // 304 elements = 608 B: consecutive copies start 24 banks apart, so the 16-byte
                                             // fragment reads of 8 lanes (8 copies, same offset) touch 32 distinct banks

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

struct F1Copies {
    bf16 xh[8][F1_ROW];
    bf16 xl[8][F1_ROW];
};

__device__ inline f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

// x_pad row of window n (length L + 31) -> 8 shifted hi/lo copies of local samples [t0, t0 + 295): copy s holds sample
// t0 + m + s at element F1_PAD + m.  Split in two so that the global loads of the next chunk are in flight while the
// current chunk's tiles are computed: fetch (thread tid: samples tid and, for the first 39 threads, 256 + tid) ...
struct F1Fetch {
    float a, b;
    bool va, vb;  // in range (the loads themselves are unconditional on a clamped index: a select in front of a load would
                  // make the compiler wait for it at the join instead of at its first use)
};
__device__ inline F1Fetch f1_fetch(const float* __restrict__ xrow, int64_t t0, int64_t row_len, int tid) {
    F1Fetch f;
    const int64_t t = t0 + tid, last = row_len - 1;
    f.va = t <= last;
    f.vb = tid < F1_CPL + 7 - 256 && t + 256 <= last;
    f.a = xrow[f.va ? t : last];
    f.b = xrow[f.vb ? t + 256 : last];
    return f;
}
// ... and stash: hi/lo split + the 8 shifted 2-byte stores per sample
// PROD (round 6; f16 storage only, vm_set_tuning "f1_products"): how many 16-bit products carry the convolution.
//   3: the split above on the bf16 pipe, xh*wh + xl*wh + xh*wl (both storage types until round 6; bf16 storage always)
//   2: the waveform ROUNDED TO HALF (11 significand bits -- what every other layer's input already is in this storage mode), the
//      filters split hi + lo in halves: xh*wh + xh*wl on the f16 pipe
//   1: xh*wh alone -- input and filters both at the storage precision, like blocks 2-4
// With PROD < 3 the xh copies hold halves and the xl copies hold bf16(x): the operand of the backward's weight-gradient MFMA, whose
// other operand du is a bf16 (gradient range) -- ONE product there too (x at 8 bits beside a du at 8 bits), where the split form
// took two.  NEED_L: the kernel reads the xl copies at all (PROD 3, or the backward).
template <int PROD, bool NEED_L>
__device__ inline void f1_stash_one(F1Copies& sm, float v, int j) {
    bf16 h, l;
    if constexpr (PROD == 3) {
        h = (bf16)v;
        l = (bf16)(v - (float)h);
    } else {
        h = __builtin_bit_cast(bf16, (_Float16)v);
        l = (bf16)v;
    }
    bf16* ph = &sm.xh[0][F1_PAD + j];
    bf16* pl = &sm.xl[0][F1_PAD + j];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        ph[s * (F1_ROW - 1)] = h;  // element F1_PAD + j - s of copy s
        if constexpr (NEED_L) pl[s * (F1_ROW - 1)] = l;
    }
}
template <int PROD, bool NEED_L>
__device__ inline void f1_stash(F1Copies& sm, const F1Fetch& f, int tid) {
    f1_stash_one<PROD, NEED_L>(sm, f.va ? f.a : 0.f, tid);
    if (tid < F1_CPL + 7 - 256) f1_stash_one<PROD, NEED_L>(sm, f.vb ? f.b : 0.f, 256 + tid);
}

struct F1Weights {
    bf16x8 h[2], l[2];  // B fragments for k-steps 0,1: element e <-> tap 16*ks + 8*kh + e
};

template <int PROD>
__device__ inline void f1_load_weights(F1Weights& w, const float* __restrict__ wk, int F, int c, bool cok, int kh) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 16 * ks + 8 * kh + e;
            const float v = cok ? wk[k * F + c] : 0.f;
            if constexpr (PROD == 3) {
                const bf16 hh = (bf16)v;
                w.h[ks][e] = hh;
                w.l[ks][e] = (bf16)(v - (float)hh);
            } else {   // halves in the 16-bit slots
                const _Float16 hh = (_Float16)v;
                w.h[ks][e] = __builtin_bit_cast(bf16, hh);
                w.l[ks][e] = __builtin_bit_cast(bf16, (_Float16)(v - (float)hh));
            }
        }
    }
}
__device__ inline f32x16 mfma_f16(bf16x8 a, bf16x8 b, f32x16 c) {   // the 16-bit slots hold halves
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// u[p][c] for the 32 positions of row tile rt (local p = 32*rt + row) x the wave's 32 channels
// (Starting the accumulator at the bias -- the C operand of the first MFMA -- instead of adding it in the epilogue was measured in
// round 4: 8 v_pk_add_f32 fewer per tile, forward 115 -> 120 us, backward 195 -> 219 us (16 more VGPRs, one wave per SIMD fewer).)
template <int PROD>
__device__ inline f32x16 f1_conv_tile(const F1Copies& sm, const F1Weights& w, int rt, int lane) {
    const int i = lane & 31, kh = lane >> 5;
    const int s = i & 7;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (VM_C1_ABL & 4) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = (float)(lane + rt) * 1e-3f * (float)(r - 7) + (float)w.h[0][r & 7];
        return acc;
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int m0 = F1_PAD + 32 * rt + (i & ~7) + 16 * ks + 8 * kh;
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&sm.xh[s][m0]);
        if constexpr (PROD == 3) {
            const bf16x8 al = *reinterpret_cast<const bf16x8*>(&sm.xl[s][m0]);
            acc = mfma_bf16(ah, w.h[ks], acc);
            acc = mfma_bf16(al, w.h[ks], acc);
            acc = mfma_bf16(ah, w.l[ks], acc);
        } else {
            acc = mfma_f16(ah, w.h[ks], acc);
            if constexpr (PROD == 2) acc = mfma_f16(ah, w.l[ks], acc);
        }
    }
    return acc;
}

// wave -> (column tile, row-tile subset).  CT = number of 32-channel tiles of this block (<= 4).
struct F1Role {
    int ct, rs, RS;
    bool active;
};
__device__ inline F1Role f1_role(int wave, int CT) {
    F1Role r;
    if (CT >= 3) {
        r.RS = 1;
        r.rs = 0;
        r.ct = wave;
        r.active = wave < CT;
    } else {  // CT = 1 or 2: several waves share a column tile and split the 8 row tiles
        r.RS = 4 / CT;
        r.ct = wave % CT;
        r.rs = wave / CT;
        r.active = true;
    }
    return r;
}

// z = relu(conv + bias) of the 4 consecutive positions this lane holds in accumulator group g
__device__ inline void f1_relu4(const f32x16& acc, int g, f32x2 bv2, f32x2& a, f32x2& b) {
    a = f32x2{acc[4 * g], acc[4 * g + 1]} + bv2;  // v_pk_add_f32: the kernels are VALU-bound, so instructions are time
    b = f32x2{acc[4 * g + 2], acc[4 * g + 3]} + bv2;
    a[0] = __builtin_fmaxf(a[0], 0.f);  // the builtin, not fmaxf(): no canonicalising v_max x,x in front of every operand
    a[1] = __builtin_fmaxf(a[1], 0.f);
    b[0] = __builtin_fmaxf(b[0], 0.f);
    b[1] = __builtin_fmaxf(b[1], 0.f);
}

// ------------------------------------------------------------------------------------------------------
// forward.  grid = (n_windows * splits, ceil(F/128)); a block walks `cps` chunks of one window (weights and the per-channel
// constants are loaded once, the waveform copies are double-buffered: one barrier per chunk).
//   TRAIN: e[n][q][c] = stored (TS) pooled extreme of relu(conv+b) (max if gamma >= 0 else min) + stat partials of the fp32 z
//   INFER: act[n][1+q][c] = TS(pooled extreme * scale + shift), sign from scale (moving-statistics affine)
// z itself is never stored, so it has no storage rounding: statistics, pooling and the backward recompute all see the
// fp32 accumulator.
template <typename TS, int POOL, bool INFER, int PROD = 3>
__global__ __launch_bounds__(256) void conv1_fused_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wk,
                                                              const float* __restrict__ bias, const float* __restrict__ sgn,
                                                              const float* __restrict__ shift, int64_t L, int F, int chunks,
                                                              int splits, int cps, TS* __restrict__ out, int e_pad,
                                                              float* __restrict__ stat_sum, float* __restrict__ stat_sq) {
    // e_pad (training): 1 = the pool extreme is written as a padded activation tensor (n_windows, L / POOL + 2, F), rows 1 .. L / POOL
    __shared__ __attribute__((aligned(16))) F1Copies cp[2];
    __shared__ float red[4][32][2];
    VM_PROF1(const long long pq_s0 = __builtin_amdgcn_s_memtime();)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t n = blockIdx.x / splits;
    const int split = (int)(blockIdx.x % splits);
    const int cbase = blockIdx.y * 128;
    const int CT = (min(F - cbase, 128) + 31) / 32;
    const F1Role role = f1_role(wave, CT);
    const int64_t Lq = L / POOL;
    const int ch_lo = split * cps;
    int ch_hi = ch_lo + cps;
    if (ch_hi > chunks) ch_hi = chunks;
    const float* xrow = x + n * (L + F1_K - 1);

    const int c = cbase + role.ct * 32 + (lane & 31);
    const bool cok = role.active && c < F;
    const int hi = lane >> 5;
    F1Weights w;
    f1_load_weights<PROD>(w, wk, F, c, cok, hi);
    const float bv = cok ? bias[c] : 0.f;
    const float sg = cok ? sgn[c] : 1.f;  // gamma (TRAIN) or scale (INFER): only its sign selects max/min ...
    const float sh = (INFER && cok) ? shift[c] : 0.f;
    const bool use_min = sg < 0.f;
    const bool all_max = __builtin_amdgcn_ballot_w64(use_min) == 0;  // wave-uniform: no channel of this wave pools a minimum
    const f32x2 bv2 = {bv, bv};
    // mode 2 (the padded extreme the folded block-2 convolution reads): the value is stored CENTRED, e - max(bias, 0).  The waveform is
    // whitened to an RMS of 0.038, so a bias of a few hundredths is already larger than the convolution it is added to; relu(conv + b)
    // then sits on a pedestal of b and a 16-bit value spends its significand on the pedestal (measured on the CPU oracle: conv1 bias
    // ~ N(0, 0.2) takes the f16 embeddings from 7e-4 to 1.3e-2 off float64; centred 7e-4 again).  Every consumer is linear in e and
    // BatchNorm removes any per-channel constant, so the offset moves into constants they already hold: the fold's shift
    // (shift + scale * ctr), the mean of the BatchNorm-backward sums (mean - ctr) -- vm_bn_finalize writes both (center_bias).
    const float ctr = (!INFER && e_pad) ? __builtin_fmaxf(bv, 0.f) : 0.f;
    const int lane_off = c + (4 / POOL) * hi * F;  // this lane's element inside a tile's pooled rows
    float csum = 0.f, csq = 0.f;
    f32x2 csum2 = {0.f, 0.f}, csq2 = {0.f, 0.f};

    // all 32 positions of a tile pooled and in range (every tile at cfg-A): no per-element predicates, one scalar base
    // pointer per tile.  ALLMAX (wave-uniform): plain maxima, no min/max selects.
    auto fast_tile = [&](const f32x16& acc, TS* ob, auto allmax) {
        constexpr bool ALLMAX = decltype(allmax)::value;
        if (VM_C1_ABL & 2) {
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) v += acc[r];  // keeps the tile alive: 16 adds instead of the epilogue
            csum += v;
            if (!(VM_C1_ABL & 1)) {
#pragma unroll
                for (int g = 0; g < 4; ++g) ob[(2 * g) * F] = (TS)acc[4 * g];
            }
            return;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (INFER && ALLMAX) {
                // no statistics: max-pooling commutes with the ReLU, so the four ReLUs collapse into the last max3
                const f32x2 a = f32x2{acc[4 * g], acc[4 * g + 1]} + bv2, b = f32x2{acc[4 * g + 2], acc[4 * g + 3]} + bv2;
                if (POOL == 4) {
                    const float ext = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(a[0], a[1]), b[0]), __builtin_fmaxf(b[1], 0.f));
                    ob[(2 * g) * F] = (TS)fmaf(ext, sg, sh);
                } else {
                    ob[(4 * g) * F] = (TS)fmaf(__builtin_fmaxf(__builtin_fmaxf(a[0], a[1]), 0.f), sg, sh);
                    ob[(4 * g + 1) * F] = (TS)fmaf(__builtin_fmaxf(__builtin_fmaxf(b[0], b[1]), 0.f), sg, sh);
                }
                continue;
            }
            f32x2 za, zb2;
            f1_relu4(acc, g, bv2, za, zb2);
            if (!INFER) {
                csum2 += za;
                csum2 += zb2;
                csq2 = __builtin_elementwise_fma(za, za, csq2);
                csq2 = __builtin_elementwise_fma(zb2, zb2, csq2);
            }
            const float z[4] = {za[0], za[1], zb2[0], zb2[1]};
#pragma unroll
            for (int pw = 0; pw < 4 / POOL; ++pw) {
                float ext = z[pw * POOL];
#pragma unroll
                for (int j = 1; j < POOL; ++j) {
                    const float v = z[pw * POOL + j];
                    ext = ALLMAX ? __builtin_fmaxf(ext, v) : (use_min ? __builtin_fminf(ext, v) : __builtin_fmaxf(ext, v));
                }
                if (VM_C1_ABL & 1) {
                    csum += ext;  // the pooled value stays live
                } else {
                    ob[((8 * g) / POOL + pw) * F] = INFER ? (TS)fmaf(ext, sg, sh) : (TS)(ext - ctr);
                }
            }
        }
    };

    VM_PROF1(long long pq_bar = 0, pq_conv = 0, pq_epi = 0, pq_stash = 0; const long long pq_s1 = __builtin_amdgcn_s_memtime();)
    F1Fetch nxt = f1_fetch(xrow, (int64_t)ch_lo * F1_CHUNK, L + F1_K - 1, tid);
    f1_stash<PROD, PROD == 3>(cp[0], nxt, tid);
    VM_PROF1(const long long pq_s2 = __builtin_amdgcn_s_memtime();)
    for (int chunk = ch_lo; chunk < ch_hi; ++chunk) {
        const int buf = (chunk - ch_lo) & 1;
        // this chunk's copies are visible, and every wave is done reading the other buffer (previous chunk)
        VM_PROF1(const long long pq_b0 = __builtin_amdgcn_s_memtime();)
        __syncthreads();
        VM_PROF1(pq_bar += __builtin_amdgcn_s_memtime() - pq_b0;)
        const bool more = chunk + 1 < ch_hi;
        if (more) nxt = f1_fetch(xrow, (int64_t)(chunk + 1) * F1_CHUNK, L + F1_K - 1, tid);  // lands while the tiles run
        const F1Copies& sm = cp[buf];
        const int64_t t0 = (int64_t)chunk * F1_CHUNK;
        for (int rt = role.rs; role.active && rt < 8; rt += role.RS) {
            if (t0 + 32 * rt >= L) break;
            VM_PROF1(const long long pq_t0 = __builtin_amdgcn_s_memtime();)
            f32x16 acc = f1_conv_tile<PROD>(sm, w, rt, lane);
            VM_PROF1(asm volatile("" : "+v"(acc)); const long long pq_t1 = __builtin_amdgcn_s_memtime(); pq_conv += pq_t1 - pq_t0;)
            if (cok && t0 + 32 * rt + 32 <= Lq * POOL) {
                TS* ob = (INFER ? out + (n * (Lq + 2) + 1 + (t0 + 32 * rt) / POOL) * F
                                : out + (n * (Lq + 2 * e_pad) + e_pad + (t0 + 32 * rt) / POOL) * F) + lane_off;
                if (all_max) {
                    fast_tile(acc, ob, std::true_type{});
                } else {
                    fast_tile(acc, ob, std::false_type{});
                }
                VM_PROF1(pq_epi += __builtin_amdgcn_s_memtime() - pq_t1;)
                continue;
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int64_t tg = t0 + 32 * rt + 8 * g + 4 * hi;  // first of 4 consecutive positions
                f32x2 za, zb2;
                f1_relu4(acc, g, bv2, za, zb2);
                const float z[4] = {za[0], za[1], zb2[0], zb2[1]};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (!INFER && tg + j < L) {
                        csum += z[j];
                        csq += z[j] * z[j];
                    }
                }
#pragma unroll
                for (int pw = 0; pw < 4 / POOL; ++pw) {
                    float ext = z[pw * POOL];
#pragma unroll
                    for (int j = 1; j < POOL; ++j) ext = use_min ? __builtin_fminf(ext, z[pw * POOL + j]) : __builtin_fmaxf(ext, z[pw * POOL + j]);
                    const int64_t q = tg / POOL + pw;
                    if (cok && q < Lq) {
                        if (INFER) {
                            out[(n * (Lq + 2) + 1 + q) * F + c] = (TS)fmaf(ext, sg, sh);  // ... and here it is the scale
                        } else {
                            out[(n * (Lq + 2 * e_pad) + e_pad + q) * F + c] = (TS)(ext - ctr);
                        }
                    }
                }
            }
        }
        VM_PROF1(const long long pq_h0 = __builtin_amdgcn_s_memtime();)
        if (more) f1_stash<PROD, PROD == 3>(cp[buf ^ 1], nxt, tid);
        VM_PROF1(pq_stash += __builtin_amdgcn_s_memtime() - pq_h0;)
    }
    VM_PROF1(const long long pq_s3 = __builtin_amdgcn_s_memtime();)
    if (!INFER) {
        // one statistics row per chunk (vm_conv1_stat_rows): the block's sum goes to its first chunk's row, zeros to the rest
        csum += csum2[0] + csum2[1];
        csq += csq2[0] + csq2[1];
        csum += __shfl_xor(csum, 32, 64);
        csq += __shfl_xor(csq, 32, 64);
        if (lane < 32) {
            red[wave][lane][0] = role.active ? csum : 0.f;
            red[wave][lane][1] = role.active ? csq : 0.f;
        }
        __syncthreads();
        if (tid < CT * 32) {
            const int ct = tid >> 5, col = tid & 31;
            const int cc = cbase + ct * 32 + col;
            if (cc < F) {
                float s = 0.f, q = 0.f;
                for (int w2 = 0; w2 < 4; ++w2) {
                    const F1Role r2 = f1_role(w2, CT);
                    if (r2.active && r2.ct == ct) {
                        s += red[w2][col][0];
                        q += red[w2][col][1];
                    }
                }
                for (int ch = ch_lo; ch < ch_hi; ++ch) {
                    const int64_t row = n * chunks + ch;
                    stat_sum[row * F + cc] = ch == ch_lo ? s : 0.f;
                    stat_sq[row * F + cc] = ch == ch_lo ? q : 0.f;
                }
            }
        }
    }
#if defined(VM_EXPERIMENT_PROFILE)
    if (!INFER) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long pq_end = __builtin_amdgcn_s_memtime();
        if (lane == 0 && blockIdx.x < 4096 && blockIdx.y == 0) {
            unsigned int* q = g_prof1 + ((int64_t)blockIdx.x * 4 + wave) * 8;
            q[0] = (unsigned int)(pq_end - pq_s0);   // whole wave
            q[1] = (unsigned int)(pq_s1 - pq_s0);    // prologue: weights, constants
            q[2] = (unsigned int)(pq_s2 - pq_s1);    // first fetch + stash
            q[3] = (unsigned int)pq_bar;             // barriers
            q[4] = (unsigned int)pq_conv;            // fragment reads + MFMAs until the accumulator is readable
            q[5] = (unsigned int)pq_epi;             // epilogue VALU + store issue
            q[6] = (unsigned int)pq_stash;           // stash of the next chunk (waits for its fetch)
            q[7] = (unsigned int)(pq_end - pq_s3);   // statistics reduction + store drain
        }
    }
#endif
}


// ------------------------------------------------------------------------------------------------------
// backward.  grid = (n_windows * splits, ceil(F/128)); a block walks `cps` chunks of one window and keeps its
// dW (32 taps x 32 channels per wave) and bias-gradient partials in registers; slab layout (33, F) fp32.
template <typename TS, int POOL, int PROD = 3>
__global__ __launch_bounds__(256) void conv1_fused_bwd_kernel(const float* __restrict__ x, const float* __restrict__ wk,
                                                              const float* __restrict__ bias, const TS* __restrict__ dp,
                                                              const float* __restrict__ scale, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, const float* __restrict__ drop,
                                                              const float* __restrict__ c1, const float* __restrict__ c2,
                                                              int64_t wpt, int64_t L, int F, int chunks, int splits, int cps,
                                                              float* __restrict__ ws) {
    __shared__ __attribute__((aligned(16))) F1Copies cp[2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t n = blockIdx.x / splits;
    const int split = (int)(blockIdx.x % splits);
    const int cbase = blockIdx.y * 128;
    const int CT = (min(F - cbase, 128) + 31) / 32;
    const F1Role role = f1_role(wave, CT);
    const int64_t Lq = L / POOL;
    const int64_t tw = n / wpt;
    const int i = lane & 31, hi = lane >> 5;
    const int c = cbase + role.ct * 32 + i;
    const bool cok = role.active && c < F;
    const float* xrow = x + n * (L + F1_K - 1);

    F1Weights w;
    f1_load_weights<PROD>(w, wk, F, c, cok, hi);
    const float bv = cok ? bias[c] : 0.f;
    const float sc = cok ? scale[tw * F + c] : 0.f;
    const float mu = cok ? mean[tw * F + c] : 0.f;
    const float is = cok ? invstd[tw * F + c] : 0.f;
    const float dr = cok ? (drop ? drop[n * F + c] : 1.0f) : 0.f;
    const float k1 = cok ? c1[tw * F + c] : 0.f;
    const float k2 = cok ? c2[tw * F + c] : 0.f;
    const bool use_min = sc < 0.f;
    const bool all_max = __builtin_amdgcn_ballot_w64(use_min) == 0;
    // du = [z>0] * (gc*z + gb0 + [arg] ga*dp)   (same folding as bn_pool_bwd_kernel)
    const float ga = sc * dr, gb0 = sc * (is * k2 * mu - k1), gc = -sc * is * k2;
    const f32x2 bv2 = {bv, bv};

    f32x16 accw;
    // the bias gradient sum_rows du[row][c] on the matrix pipe too (the kernel is VALU-bound: the converts + adds per tile move there),
    // as v_mfma_f32_4x4x4 with an all-ones A operand: D[lane][0] = the sum of the lane's own four K values (round 6: the 32 x 32 x 16
    // form spent a whole 32 x 32 product -- two of the tile's twelve large MFMAs -- on sixty-four column sums)
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    f32x4 accb0 = {0.f, 0.f, 0.f, 0.f}, accb1 = accb0;
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[r] = 0.f;
    const s16x4 ones4 = {(short)0x3F80, (short)0x3F80, (short)0x3F80, (short)0x3F80};   // bf16 1.0

    const int ch_lo = split * cps;
    int ch_hi = ch_lo + cps;
    if (ch_hi > chunks) ch_hi = chunks;
    constexpr int PW = 4 / POOL;    // pool windows per group of 4 consecutive positions
    constexpr int NDP = 4 * PW;     // pooled gradients one lane needs per tile
    // du of a tile whose 32 positions are all pooled and in range (every tile at cfg-A), dpv = its pooled gradients
    auto fast_tile = [&](const f32x16& acc, const TS (&dpv)[NDP], bf16x2 (&dub)[8], auto allmax) {
        constexpr bool ALLMAX = decltype(allmax)::value;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x2 za, zb2;
            if (ALLMAX) {
                // no ReLU here: the pre-activation v = conv + bias serves.  Among positive values max-pooling v or relu(v) picks the
                // same first maximum; where every v of a window is <= 0 the choice differs but every du of the window is masked to
                // zero either way; and gc * relu(v) + [v > 0] a == [v > 0] (gc * v + a).  Four v_max fewer per group, same bits.
                za = f32x2{acc[4 * g], acc[4 * g + 1]} + bv2;
                zb2 = f32x2{acc[4 * g + 2], acc[4 * g + 3]} + bv2;
            } else {
                f1_relu4(acc, g, bv2, za, zb2);
            }
            const float z[4] = {za[0], za[1], zb2[0], zb2[1]};
            float gz[4];
#pragma unroll
            for (int pw = 0; pw < 4 / POOL; ++pw) {
                const float* zw = &z[pw * POOL];
                const float gb1 = fmaf(ga, (float)dpv[PW * g + pw], gb0);
                // The first extreme of the pool window takes the pooled gradient: its addend is gb1, the others' gb0; where
                // z = 0 the product gc*z vanishes by itself, so only the addend needs the ReLU mask.
                float a[POOL];
                if (ALLMAX) {
                    // running maximum by compare + select: the three compares ARE the "strictly greater than everything
                    // before" flags, and the first-maximum one-hot follows from them on the scalar unit (lane masks)
                    uint64_t cgt[POOL];
                    float run = zw[0];
#pragma unroll
                    for (int j = 1; j < POOL; ++j) {
                        const bool gt = zw[j] > run;
                        cgt[j] = __builtin_amdgcn_ballot_w64(gt);
                        if (j + 1 < POOL) run = gt ? zw[j] : run;
                    }
                    uint64_t later = 0;  // some later position is strictly greater than everything before it
#pragma unroll
                    for (int j = POOL - 1; j >= 1; --j) {
                        a[j] = __builtin_amdgcn_inverse_ballot_w64(cgt[j] & ~later) ? gb1 : gb0;
                        later |= cgt[j];
                    }
                    a[0] = __builtin_amdgcn_inverse_ballot_w64(later) ? gb0 : gb1;
                } else {
                    float ext = zw[0];
                    int arg = 0;
#pragma unroll
                    for (int j = 1; j < POOL; ++j) {
                        const bool better = use_min ? (zw[j] < ext) : (zw[j] > ext);  // strict: the first extreme keeps the gradient
                        if (better) {
                            ext = zw[j];
                            arg = j;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < POOL; ++j) a[j] = j == arg ? gb1 : gb0;
                }
#pragma unroll
                for (int j = 0; j < POOL; ++j) {
                    if (ALLMAX) {
                        const float t = fmaf(gc, zw[j], a[j]);
                        gz[pw * POOL + j] = zw[j] > 0.f ? t : 0.f;
                    } else {
                        gz[pw * POOL + j] = fmaf(gc, zw[j], zw[j] > 0.f ? a[j] : 0.f);
                    }
                }
            }
            dub[2 * g] = __builtin_convertvector(f32x2{gz[0], gz[1]}, bf16x2);
            dub[2 * g + 1] = __builtin_convertvector(f32x2{gz[2], gz[3]}, bf16x2);
        }
    };
    // pooled gradients of the tile at row tile rt of the chunk starting at t0.  Unconditional loads: a tile that is not a
    // fast-path one (or a lane without a channel) reads a valid dummy element instead, so the compiler keeps the loads in
    // flight until their first use instead of waiting at a branch join.
    const int csafe = cok ? c : 0;
    auto load_dp = [&](TS (&dpv)[NDP], int64_t t0, int rt) {
        const bool fast = rt < 8 && t0 + 32 * rt + 32 <= Lq * POOL;  // wave-uniform
        const int64_t Fs = fast ? F : 0;
        const TS* dpb = dp + (fast ? (n * Lq + (t0 + 32 * rt) / POOL) * F : 0) + csafe + PW * hi * Fs;
#pragma unroll
        for (int k = 0; k < NDP; ++k) dpv[k] = (VM_C1_ABL & 1) ? (TS)(float)(k + rt) : dpb[((8 * (k / PW)) / POOL + k % PW) * Fs];
    };

    F1Fetch nxt = f1_fetch(xrow, (int64_t)ch_lo * F1_CHUNK, L + F1_K - 1, tid);
    f1_stash<PROD, true>(cp[0], nxt, tid);
    for (int chunk = ch_lo; chunk < ch_hi; ++chunk) {
        const int buf = (chunk - ch_lo) & 1;
        const int64_t t0 = (int64_t)chunk * F1_CHUNK;
        TS dpn[NDP];  // the next tile's pooled gradients, loaded one tile ahead (their latency hides behind a tile of VALU work)
        load_dp(dpn, t0, role.rs);
        __syncthreads();  // this chunk's copies are visible; the other buffer's readers (previous chunk) are done
        const bool more = chunk + 1 < ch_hi;
        if (more) nxt = f1_fetch(xrow, (int64_t)(chunk + 1) * F1_CHUNK, L + F1_K - 1, tid);  // lands while the tiles run
        const F1Copies& sm = cp[buf];
        for (int rt = role.rs; role.active && rt < 8; rt += role.RS) {
            if (t0 + 32 * rt >= L) break;
            const f32x16 acc = f1_conv_tile<PROD>(sm, w, rt, lane);
            TS dpv[NDP];
#pragma unroll
            for (int k = 0; k < NDP; ++k) dpv[k] = dpn[k];
            load_dp(dpn, t0, rt + role.RS);
            bf16x2 dub[8];  // du of accumulator registers (2k, 2k+1), already packed as the next MFMA wants them
            if (VM_C1_ABL & 2) {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    dub[k] = __builtin_convertvector(f32x2{acc[2 * k] + (float)dpv[k & (NDP - 1)], acc[2 * k + 1]}, bf16x2);
            } else if (cok && t0 + 32 * rt + 32 <= Lq * POOL) {
                if (all_max) {
                    fast_tile(acc, dpv, dub, std::true_type{});
                } else {
                    fast_tile(acc, dpv, dub, std::false_type{});
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int64_t tg = t0 + 32 * rt + 8 * g + 4 * hi;
                    f32x2 za, zb2;
                    f1_relu4(acc, g, bv2, za, zb2);
                    const float z[4] = {za[0], za[1], zb2[0], zb2[1]};
                    float gz[4];
#pragma unroll
                    for (int pw = 0; pw < 4 / POOL; ++pw) {
                        float ext = z[pw * POOL];
                        int arg = 0;
#pragma unroll
                        for (int j = 1; j < POOL; ++j) {
                            const float v = z[pw * POOL + j];
                            const bool better = use_min ? (v < ext) : (v > ext);
                            if (better) {
                                ext = v;
                                arg = j;
                            }
                        }
                        const int64_t q = tg / POOL + pw;
                        float ady = 0.f;
                        if (cok && q < Lq) ady = ga * (float)dp[(n * Lq + q) * F + c];
#pragma unroll
                        for (int j = 0; j < POOL; ++j) {
                            const float zz = z[pw * POOL + j];
                            const float v = fmaf(gc, zz, gb0) + (j == arg ? ady : 0.f);
                            gz[pw * POOL + j] = (zz > 0.f && tg + pw * POOL + j < L) ? v : 0.f;
                        }
                    }
                    dub[2 * g] = __builtin_convertvector(f32x2{gz[0], gz[1]}, bf16x2);
                    dub[2 * g + 1] = __builtin_convertvector(f32x2{gz[2], gz[3]}, bf16x2);
                }
            }
            // weight gradient: dW[tap i][c] += sum_rows x_loc[32*rt + row + i] * du[row][c].  K-slot e of half kh in
            // MFMA m <-> accumulator register 8m+e of this lane <-> row 16m + 8(e>>2) + 4kh + (e&3).
            if (VM_C1_ABL & 8) {
#pragma unroll
                for (int k = 0; k < 8; ++k) accw[k] += (float)dub[k][0] + (float)dub[k][1];
                continue;
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                bf16x8 bfrag;
#pragma unroll
                for (int e = 0; e < 8; ++e) bfrag[e] = dub[4 * m + (e >> 1)][e & 1];
                const int s = i & 3;
                bf16x8 ah, al;
#pragma unroll
                for (int eh = 0; eh < 2; ++eh) {
                    const int S = 32 * rt + 16 * m + 8 * eh + 4 * hi + i;
                    const int m0 = F1_PAD + S - s;
                    // bf16(x): the hi copies of the split form, the xl copies otherwise
                    const bf16x4 vx = *reinterpret_cast<const bf16x4*>(PROD == 3 ? &sm.xh[s][m0] : &sm.xl[s][m0]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) ah[4 * eh + e] = vx[e];
                }
                // ONE product: x at 8 significand bits beside a du at 8 -- the residual xl * du the split form added until round 6 is a
                // zero-mean 2^-9 of terms whose other factor is already rounded at 2^-9, summed over 3 M positions
                accw = mfma_bf16(ah, bfrag, accw);
                (void)al;
                {
                    const u32x4 bw = __builtin_bit_cast(u32x4, bfrag);
                    accb0 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(ones4, __builtin_bit_cast(s16x4, u32x2{bw[0], bw[1]}), accb0, 0, 0, 0);
                    accb1 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(ones4, __builtin_bit_cast(s16x4, u32x2{bw[2], bw[3]}), accb1, 0, 0, 0);
                }
            }
        }
        if (more) f1_stash<PROD, true>(cp[buf ^ 1], nxt, tid);
    }
    // one slab per block.x: (33, F) = 32 tap rows + the bias-gradient row.  Waves that share a column tile
    // (CT < 3: the row tiles are split over several waves) are summed through LDS in a fixed order.
    __shared__ float wred[4][17][64];
#pragma unroll
    for (int r = 0; r < 16; ++r) wred[wave][r][lane] = role.active ? accw[r] : 0.f;
    // every lane holds the sum over ITS K slots (half of the channel's rows): the two lane halves are added below
    wred[wave][16][lane] = role.active ? accb0[0] + accb1[0] : 0.f;
    __syncthreads();
    if (role.active && role.rs == 0) {
        float* slab = ws + (int64_t)blockIdx.x * 33 * F;
        float tot[17];
#pragma unroll
        for (int r = 0; r < 17; ++r) tot[r] = 0.f;
        for (int w2 = 0; w2 < 4; ++w2) {
            const F1Role r2 = f1_role(w2, CT);
            if (r2.active && r2.ct == role.ct) {
#pragma unroll
                for (int r = 0; r < 17; ++r) tot[r] += wred[w2][r][lane];
            }
        }
        if (cok) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tap = (r & 3) + 8 * (r >> 2) + 4 * hi;
                slab[tap * F + c] = tot[r];
            }
        }
        const float bs = tot[16] + __shfl_xor(tot[16], 32, 64);
        if (cok && hi == 0) slab[32 * F + c] = bs;
    }
}

#if defined(VM_EXPERIMENT_PROFILE)
extern "C" int vm_debug_prof1_read(unsigned int* out, int n_slots) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof1), sizeof(unsigned int) * 8 * (size_t)n_slots);
    return 0;
}
#endif

int g_f1_products = 2;       // f16 storage: 16-bit products of the block-1 convolution (see f1_stash_one); vm_set_tuning("f1_products", 1 | 2 | 3)
int g_f1_blocks = 1024;      // target workgroup count of the fused block-1 backward (vm_set_tuning("f1_blocks", n))
int g_f1_fwd_blocks = 1024;  // ... and of the forward (vm_set_tuning("f1_fwd_blocks", n)); round 6, with four MFMAs per tile: 4096 -> 1024 is -0.7 % of the cfg-A step, -0.9 % cfg-B (a workgroup walks 6 chunks per tower launch instead of 2: filters and constants loaded once)

static int f1_splits(int64_t n_windows, int chunks, int target) {
    int s = (int)((target + n_windows - 1) / n_windows);  // aim for >= target workgroups
    if (s < 1) s = 1;
    if (s > chunks) s = chunks;
    const int cps = (chunks + s - 1) / s;
    return (chunks + cps - 1) / cps;
}

int f1_set_blocks(int v) {
    g_f1_blocks = v;
    return 0;
}
int f1_set_fwd_blocks(int v) {
    g_f1_fwd_blocks = v;
    return 0;
}

}  // namespace vm

using namespace vm;

extern "C" int vm_conv1_fused_fwd(const float* x, const float* w, const float* bias, const float* gamma_or_scale,
                                  const float* shift, int64_t n_windows, int64_t L, int F, int pool, int inference, int dtype,
                                  void* out, float* stat_sum, float* stat_sq, void* stream) {
    VM_REQUIRE(x && w && bias && gamma_or_scale && out, "vm_conv1_fused_fwd: null pointer");
    VM_REQUIRE(n_windows > 0 && L > 0 && F > 0 && F % 8 == 0, "vm_conv1_fused_fwd: bad sizes");
    VM_REQUIRE(pool == 2 || pool == 4, "vm_conv1_fused_fwd: pool must be 2 or 4 (got %d)", pool);
    VM_REQUIRE(dtype == VM_BF16 || dtype == VM_F16, "vm_conv1_fused_fwd: 16-bit storage only (VM_BF16 / VM_F16), got dtype %d", dtype);
    VM_REQUIRE(inference >= 0 && inference <= 2, "vm_conv1_fused_fwd: mode must be 0 (training), 1 (inference) or 2 (training, padded extreme)");
    VM_REQUIRE(inference == 1 ? shift != nullptr : (stat_sum && stat_sq), "vm_conv1_fused_fwd: missing shift / stat buffers");
    const int chunks = (int)((L + F1_CHUNK - 1) / F1_CHUNK);
    const int splits = f1_splits(n_windows, chunks, g_f1_fwd_blocks);
    const int cps = (chunks + splits - 1) / splits;
    const int64_t gx = n_windows * splits;
    VM_REQUIRE(gx < (1LL << 31), "vm_conv1_fused_fwd: grid too large");
    const dim3 grid((unsigned)gx, (unsigned)((F + 127) / 128));
#define VM_F1_FWD_P(POOL, INF, PROD)                                                                                      \
    hipLaunchKernelGGL((conv1_fused_fwd_kernel<T, POOL, INF, PROD>), grid, dim3(256), 0, (hipStream_t)stream, x, w, bias,  \
                       gamma_or_scale, shift, L, F, chunks, splits, cps, (T*)out, inference == 2 ? 1 : 0, stat_sum, stat_sq)
#define VM_F1_FWD(POOL, INF)                                                         \
    if constexpr (std::is_same<T, f16>::value) {                                     \
        if (g_f1_products == 1) VM_F1_FWD_P(POOL, INF, 1);                           \
        else if (g_f1_products == 2) VM_F1_FWD_P(POOL, INF, 2);                      \
        else VM_F1_FWD_P(POOL, INF, 3);                                              \
    } else {                                                                         \
        VM_F1_FWD_P(POOL, INF, 3);                                                   \
    }
    VM_DISPATCH_16(dtype, {
        if (pool == 2) {
            if (inference == 1) { VM_F1_FWD(2, true) } else { VM_F1_FWD(2, false) }
        } else {
            if (inference == 1) { VM_F1_FWD(4, true) } else { VM_F1_FWD(4, false) }
        }
    });
#undef VM_F1_FWD
#undef VM_F1_FWD_P
    return check_launch("vm_conv1_fused_fwd");
}

extern "C" int64_t vm_conv1_fused_bwd_workspace_bytes(int64_t n_windows, int64_t L, int F) {
    const int chunks = (int)((L + F1_CHUNK - 1) / F1_CHUNK);
    const int64_t slabs = n_windows * f1_splits(n_windows, chunks, g_f1_blocks);
    return slabs * 33 * (int64_t)F * (int64_t)sizeof(float) + slab_sum_part_bytes(33LL * F);
}

extern "C" int vm_conv1_fused_bwd(const float* x, const float* w, const float* bias, const void* dp, const float* scale,
                                  const float* mean, const float* invstd, const float* drop, const float* c1, const float* c2,
                                  int64_t n_windows, int64_t windows_per_tower, int64_t L, int F, int pool, int dtype, void* ws,
                                  float* grad_w, float* grad_b, void* stream) {
    VM_REQUIRE(x && w && bias && dp && scale && mean && invstd && c1 && c2 && ws && grad_w && grad_b,
               "vm_conv1_fused_bwd: null pointer");
    VM_REQUIRE(n_windows > 0 && windows_per_tower > 0 && L >= pool && F > 0 && F % 8 == 0, "vm_conv1_fused_bwd: bad sizes");
    VM_REQUIRE(pool == 2 || pool == 4, "vm_conv1_fused_bwd: pool must be 2 or 4 (got %d)", pool);
    VM_REQUIRE(dtype == VM_BF16 || dtype == VM_F16, "vm_conv1_fused_bwd: 16-bit storage only (VM_BF16 / VM_F16), got dtype %d", dtype);
    const int chunks = (int)((L + F1_CHUNK - 1) / F1_CHUNK);
    const int splits = f1_splits(n_windows, chunks, g_f1_blocks);
    const int cps = (chunks + splits - 1) / splits;
    const int64_t gx = n_windows * splits;
    VM_REQUIRE(gx < (1LL << 31), "vm_conv1_fused_bwd: grid too large");
    const dim3 grid((unsigned)gx, (unsigned)((F + 127) / 128));
#define VM_F1_BWD_P(POOL, PROD)                                                                                                    \
    hipLaunchKernelGGL((conv1_fused_bwd_kernel<T, POOL, PROD>), grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, (const T*)dp, \
                       scale, mean, invstd, drop, c1, c2, windows_per_tower, L, F, chunks, splits, cps, (float*)ws)
#define VM_F1_BWD(POOL)                                                              \
    if constexpr (std::is_same<T, f16>::value) {                                     \
        if (g_f1_products == 1) VM_F1_BWD_P(POOL, 1);                                \
        else if (g_f1_products == 2) VM_F1_BWD_P(POOL, 2);                           \
        else VM_F1_BWD_P(POOL, 3);                                                   \
    } else {                                                                         \
        VM_F1_BWD_P(POOL, 3);                                                        \
    }
    VM_DISPATCH_16(dtype, {
        if (pool == 2) { VM_F1_BWD(2) } else { VM_F1_BWD(4) }
    });
#undef VM_F1_BWD
#undef VM_F1_BWD_P
    int rc = check_launch("vm_conv1_fused_bwd");
    if (rc) return rc;
    const int64_t nel = 33LL * F;
    return slab_sum((const float*)ws, gx, nel, grad_w, 32LL * F, grad_b, (float*)ws + gx * nel, (hipStream_t)stream);
}
