// Block 1 of the encoder, fused for bf16 storage (voicemap/models.py:13-19: Conv1D(F,32,'same',relu) ->
// BatchNormalization -> SpatialDropout1D -> MaxPool1D).
//
// C_in = 1 makes this block 4 % of the FLOPs but the largest activation (B x 12000 x F): writing relu(conv) at
// full resolution and streaming it three more times (BN+pool, BN-backward reduce, BN-backward apply) cost more
// HBM time than the whole rest of the network.  Here the full-resolution tensor never exists in HBM:
//
//   forward : MFMA conv tile -> +bias, ReLU, bf16 round -> per-channel sum / sum-of-squares partials (BN statistics)
//             -> MaxPool *in registers*.  BN is a per-channel affine applied after the statistics are known and
//             max-pooling commutes with a monotone map:  max_j(a*z_j + b) = a*max_j z_j + b for a >= 0 and
//             a*min_j z_j + b for a < 0 (sign(a) = sign(gamma), known before the launch).  So only the pooled
//             extreme of z (1/pool of the bytes) is stored; the affine + dropout run over the pooled tensor.
//   backward: the conv tile is *recomputed* from the waveform (32 taps, trivial on the matrix cores), the BN/ReLU/pool
//             backward is evaluated in registers and the result du feeds the weight-gradient MFMA directly as its
//             B operand (the 32x32 accumulator layout of du IS a valid K-slot assignment for the next MFMA, as in
//             flash-attention's P.V step), with the matching waveform samples gathered from LDS as the A operand.
//
// MFMA operands are bf16, but the waveform and the filters are split hi+lo (x = xh + xl, both bf16) and the three
// significant products accumulated (xh*wh + xl*wh + xh*wl), so block 1 keeps ~16 mantissa bits of its fp32 inputs.
// The waveform tile lives in LDS as 8 sample-shifted copies so that every 8-sample (16-byte) fragment of the
// sliding window  x[p+k .. p+k+8)  is an aligned ds_read_b128 whatever p is.
#include "common.hpp"

namespace vm {

constexpr int F1_K = 32;       // taps
constexpr int F1_CHUNK = 256;  // positions per chunk (8 MFMA row tiles of 32)
constexpr int F1_CPL = 288;    // samples per shifted copy (>= 256 + 31, multiple of 8)

struct F1Smem {
    bf16 xh[8][F1_CPL];
    bf16 xl[8][F1_CPL];
    float red[4][32][2];
};

__device__ inline f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

__device__ inline float bf16_round(float v) { return (float)(bf16)v; }

// x_pad row of window n (length L + 31) -> 8 shifted hi/lo copies of local samples [t0, t0 + 295)
__device__ inline void f1_build_copies(F1Smem& sm, const float* __restrict__ xrow, int64_t t0, int64_t row_len, int tid) {
    for (int j = tid; j < F1_CPL + 7; j += 256) {
        const int64_t t = t0 + j;
        const float v = t < row_len ? xrow[t] : 0.f;
        const bf16 h = (bf16)v;
        const bf16 l = (bf16)(v - (float)h);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int m = j - s;
            if (m >= 0 && m < F1_CPL) {
                sm.xh[s][m] = h;
                sm.xl[s][m] = l;
            }
        }
    }
}

struct F1Weights {
    bf16x8 h[2], l[2];  // B fragments for k-steps 0,1: element e <-> tap 16*ks + 8*kh + e
};

__device__ inline void f1_load_weights(F1Weights& w, const float* __restrict__ wk, int F, int c, bool cok, int kh) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 16 * ks + 8 * kh + e;
            const float v = cok ? wk[k * F + c] : 0.f;
            const bf16 hh = (bf16)v;
            w.h[ks][e] = hh;
            w.l[ks][e] = (bf16)(v - (float)hh);
        }
    }
}

// u[p][c] for the 32 positions of row tile rt (local p = 32*rt + row) x the wave's 32 channels
__device__ inline f32x16 f1_conv_tile(const F1Smem& sm, const F1Weights& w, int rt, int lane) {
    const int i = lane & 31, kh = lane >> 5;
    const int s = i & 7;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int m0 = 32 * rt + (i & ~7) + 16 * ks + 8 * kh;
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&sm.xh[s][m0]);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(&sm.xl[s][m0]);
        acc = mfma_bf16(ah, w.h[ks], acc);
        acc = mfma_bf16(al, w.h[ks], acc);
        acc = mfma_bf16(ah, w.l[ks], acc);
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

// ------------------------------------------------------------------------------------------------------
// forward.  grid = (n_windows * chunks, ceil(F/128)).
//   TRAIN: e[n][q][c] = pooled extreme of bf16(relu(conv+b)) (max if gamma >= 0 else min) + stat partials
//   INFER: act[n][1+q][c] = bf16(pooled extreme * scale + shift), sign from scale (moving-statistics affine)
template <int POOL, bool INFER>
__global__ __launch_bounds__(256) void conv1_fused_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wk,
                                                              const float* __restrict__ bias, const float* __restrict__ sgn,
                                                              const float* __restrict__ shift, int64_t L, int F, int chunks,
                                                              bf16* __restrict__ out, float* __restrict__ stat_sum,
                                                              float* __restrict__ stat_sq) {
    __shared__ __attribute__((aligned(16))) F1Smem sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t n = blockIdx.x / chunks;
    const int chunk = (int)(blockIdx.x % chunks);
    const int64_t t0 = (int64_t)chunk * F1_CHUNK;
    const int cbase = blockIdx.y * 128;
    const int CT = (min(F - cbase, 128) + 31) / 32;
    const F1Role role = f1_role(wave, CT);
    const int64_t Lq = L / POOL;

    f1_build_copies(sm, x + n * (L + F1_K - 1), t0, L + F1_K - 1, tid);
    __syncthreads();

    const int c = cbase + role.ct * 32 + (lane & 31);
    const bool cok = role.active && c < F;
    const int hi = lane >> 5;
    float csum = 0.f, csq = 0.f;
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    f32x2 csum2 = {0.f, 0.f}, csq2 = {0.f, 0.f};
    if (role.active) {
        F1Weights w;
        f1_load_weights(w, wk, F, c, cok, hi);
        const float bv = cok ? bias[c] : 0.f;
        const float sg = cok ? sgn[c] : 1.f;  // gamma (TRAIN) or scale (INFER): only its sign selects max/min ...
        const float sh = (INFER && cok) ? shift[c] : 0.f;
        const bool use_min = sg < 0.f;
        for (int rt = role.rs; rt < 8; rt += role.RS) {
            if (t0 + 32 * rt >= L) break;
            const f32x16 acc = f1_conv_tile(sm, w, rt, lane);
            if (cok && t0 + 32 * rt + 32 <= Lq * POOL) {
                // Fast path (every tile at cfg-A): all 32 positions are pooled and in range -- no per-element predicates, no
                // 64-bit index arithmetic, one base pointer per tile.  The kernel is VALU-bound, so instructions are time.
                bf16* ob = INFER ? out + (n * (Lq + 2) + 1 + (t0 + 32 * rt) / POOL) * F + c : out + (n * Lq + (t0 + 32 * rt) / POOL) * F + c;
                const f32x2 bv2 = {bv, bv};
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float zb[4];
                    // two-wide fp32 arithmetic (v_pk_add_f32 / v_pk_fma_f32): half the VALU issue slots for bias and statistics
#pragma unroll
                    for (int j2 = 0; j2 < 2; ++j2) {
                        f32x2 v = f32x2{acc[4 * g + 2 * j2], acc[4 * g + 2 * j2 + 1]} + bv2;
                        v[0] = v[0] > 0.f ? v[0] : 0.f;
                        v[1] = v[1] > 0.f ? v[1] : 0.f;
                        const f32x2 zr = {bf16_round(v[0]), bf16_round(v[1])};
                        zb[2 * j2] = zr[0];
                        zb[2 * j2 + 1] = zr[1];
                        if (!INFER) {
                            csum2 += zr;
                            csq2 = __builtin_elementwise_fma(zr, zr, csq2);
                        }
                    }
#pragma unroll
                    for (int pw = 0; pw < 4 / POOL; ++pw) {
                        float ext = zb[pw * POOL];
#pragma unroll
                        for (int j = 1; j < POOL; ++j) {
                            const float v = zb[pw * POOL + j];
                            ext = use_min ? fminf(ext, v) : fmaxf(ext, v);
                        }
                        const int qo = ((8 * g) / POOL + pw) * F + (4 / POOL) * hi * F;  // pooled row offset inside the tile
                        ob[qo] = INFER ? (bf16)fmaf(ext, sg, sh) : (bf16)ext;
                    }
                }
                continue;
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int64_t tg = t0 + 32 * rt + 8 * g + 4 * hi;  // first of 4 consecutive positions
                float zb[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = acc[4 * g + j] + bv;
                    v = v > 0.f ? v : 0.f;
                    zb[j] = bf16_round(v);
                    if (!INFER && tg + j < L) {
                        csum += zb[j];
                        csq += zb[j] * zb[j];
                    }
                }
#pragma unroll
                for (int pw = 0; pw < 4 / POOL; ++pw) {
                    float ext = zb[pw * POOL];
#pragma unroll
                    for (int j = 1; j < POOL; ++j) {
                        const float v = zb[pw * POOL + j];
                        ext = use_min ? fminf(ext, v) : fmaxf(ext, v);
                    }
                    const int64_t q = tg / POOL + pw;
                    if (cok && q < Lq) {
                        if (INFER) {
                            out[(n * (Lq + 2) + 1 + q) * F + c] = (bf16)fmaf(ext, sg, sh);  // ... and here it is the scale
                        } else {
                            out[(n * Lq + q) * F + c] = (bf16)ext;
                        }
                    }
                }
            }
        }
    }
    if (!INFER) {
        csum += csum2[0] + csum2[1];
        csq += csq2[0] + csq2[1];
        csum += __shfl_xor(csum, 32, 64);
        csq += __shfl_xor(csq, 32, 64);
        if (lane < 32) {
            sm.red[wave][lane][0] = role.active ? csum : 0.f;
            sm.red[wave][lane][1] = role.active ? csq : 0.f;
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
                        s += sm.red[w2][col][0];
                        q += sm.red[w2][col][1];
                    }
                }
                const int64_t row = n * chunks + chunk;
                stat_sum[row * F + cc] = s;
                stat_sq[row * F + cc] = q;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// backward.  grid = (n_windows * splits, ceil(F/128)); a block walks `cps` chunks of one window and keeps its
// dW (32 taps x 32 channels per wave) and bias-gradient partials in registers; slab layout (33, F) fp32.
template <int POOL>
__global__ __launch_bounds__(256) void conv1_fused_bwd_kernel(const float* __restrict__ x, const float* __restrict__ wk,
                                                              const float* __restrict__ bias, const bf16* __restrict__ dp,
                                                              const float* __restrict__ scale, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, const float* __restrict__ drop,
                                                              const float* __restrict__ c1, const float* __restrict__ c2,
                                                              int64_t wpt, int64_t L, int F, int chunks, int splits, int cps,
                                                              float* __restrict__ ws) {
    __shared__ __attribute__((aligned(16))) F1Smem sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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

    F1Weights w;
    f1_load_weights(w, wk, F, c, cok, hi);
    const float bv = cok ? bias[c] : 0.f;
    const float sc = cok ? scale[tw * F + c] : 0.f;
    const float mu = cok ? mean[tw * F + c] : 0.f;
    const float is = cok ? invstd[tw * F + c] : 0.f;
    const float dr = cok ? (drop ? drop[n * F + c] : 1.0f) : 0.f;
    const float k1 = cok ? c1[tw * F + c] : 0.f;
    const float k2 = cok ? c2[tw * F + c] : 0.f;
    const bool use_min = sc < 0.f;
    // du = [z>0] * (gc*z + gb0 + [arg] ga*dp)   (same folding as bn_pool_bwd_kernel)
    const float ga = sc * dr, gb0 = sc * (is * k2 * mu - k1), gc = -sc * is * k2;

    f32x16 accw, accb;  // accb: the bias gradient sum_rows du[row][c], as one more MFMA with an all-ones A operand -- the
                        // kernel is VALU-bound and the matrix pipe is 86 % idle, so the 32 converts + adds per tile move there
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[r] = accb[r] = 0.f;
    bf16x8 ones8;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones8[e] = (bf16)1.0f;

    const int ch_lo = split * cps;
    int ch_hi = ch_lo + cps;
    if (ch_hi > chunks) ch_hi = chunks;
    for (int chunk = ch_lo; chunk < ch_hi; ++chunk) {
        const int64_t t0 = (int64_t)chunk * F1_CHUNK;
        __syncthreads();  // previous chunk's fragment reads are done
        f1_build_copies(sm, x + n * (L + F1_K - 1), t0, L + F1_K - 1, tid);
        __syncthreads();
        if (!role.active) continue;
        for (int rt = role.rs; rt < 8; rt += role.RS) {
            if (t0 + 32 * rt >= L) break;
            const f32x16 acc = f1_conv_tile(sm, w, rt, lane);
            bf16 dub[16];
            if (cok && t0 + 32 * rt + 32 <= Lq * POOL) {
                // fast path (every tile at cfg-A): no per-element range predicates, one dp base pointer per tile
                const bf16* dpb = dp + (n * Lq + (t0 + 32 * rt) / POOL) * F + c;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float zb[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float v = acc[4 * g + j] + bv;
                        v = v > 0.f ? v : 0.f;
                        zb[j] = bf16_round(v);
                    }
#pragma unroll
                    for (int pw = 0; pw < 4 / POOL; ++pw) {
                        float ext = zb[pw * POOL];
                        int arg = 0;
#pragma unroll
                        for (int j = 1; j < POOL; ++j) {
                            const float v = zb[pw * POOL + j];
                            const bool better = use_min ? (v < ext) : (v > ext);
                            if (better) {
                                ext = v;
                                arg = j;
                            }
                        }
                        const int qo = ((8 * g) / POOL + pw) * F + (4 / POOL) * hi * F;
                        const float ady = ga * (float)dpb[qo];
#pragma unroll
                        for (int j = 0; j < POOL; ++j) {
                            const float zz = zb[pw * POOL + j];
                            float gz = fmaf(gc, zz, gb0) + (j == arg ? ady : 0.f);
                            gz = zz > 0.f ? gz : 0.f;
                            const bf16 gb = (bf16)gz;
                            dub[4 * g + pw * POOL + j] = gb;
                        }
                    }
                }
            } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int64_t tg = t0 + 32 * rt + 8 * g + 4 * hi;
                float zb[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = acc[4 * g + j] + bv;
                    v = v > 0.f ? v : 0.f;
                    zb[j] = bf16_round(v);
                }
#pragma unroll
                for (int pw = 0; pw < 4 / POOL; ++pw) {
                    float ext = zb[pw * POOL];
                    int arg = 0;
#pragma unroll
                    for (int j = 1; j < POOL; ++j) {
                        const float v = zb[pw * POOL + j];
                        const bool better = use_min ? (v < ext) : (v > ext);  // strict: the first extreme keeps the gradient
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
                        const float zz = zb[pw * POOL + j];
                        float gz = fmaf(gc, zz, gb0) + (j == arg ? ady : 0.f);
                        gz = (zz > 0.f && tg + pw * POOL + j < L) ? gz : 0.f;
                        const bf16 gb = (bf16)gz;
                        dub[4 * g + pw * POOL + j] = gb;
                    }
                }
            }
            }
            // weight gradient: dW[tap i][c] += sum_rows x_loc[32*rt + row + i] * du[row][c].  K-slot e of half kh in
            // MFMA m <-> accumulator register 8m+e of this lane <-> row 16m + 8(e>>2) + 4kh + (e&3).
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                bf16x8 bfrag;
#pragma unroll
                for (int e = 0; e < 8; ++e) bfrag[e] = dub[8 * m + e];
                const int s = i & 3;
                typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
                bf16x8 ah, al;
#pragma unroll
                for (int eh = 0; eh < 2; ++eh) {
                    const int S = 32 * rt + 16 * m + 8 * eh + 4 * hi + i;
                    const int m0 = S - s;
                    const bf16x4 vh = *reinterpret_cast<const bf16x4*>(&sm.xh[s][m0]);
                    const bf16x4 vl = *reinterpret_cast<const bf16x4*>(&sm.xl[s][m0]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ah[4 * eh + e] = vh[e];
                        al[4 * eh + e] = vl[e];
                    }
                }
                accw = mfma_bf16(ah, bfrag, accw);
                accw = mfma_bf16(al, bfrag, accw);
                accb = mfma_bf16(ones8, bfrag, accb);
            }
        }
    }
    // one slab per block.x: (33, F) = 32 tap rows + the bias-gradient row.  Waves that share a column tile
    // (CT < 3: the row tiles are split over several waves) are summed through LDS in a fixed order.
    __shared__ float wred[4][17][64];
#pragma unroll
    for (int r = 0; r < 16; ++r) wred[wave][r][lane] = role.active ? accw[r] : 0.f;
    // every accumulator row of accb holds the column sum over ALL K slots (both lane halves): count it once
    wred[wave][16][lane] = (role.active && hi == 0) ? accb[0] : 0.f;
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

int g_f1_blocks = 2048;  // target workgroup count of the fused block-1 backward (vm_set_tuning("f1_blocks", n))

static int f1_splits(int64_t n_windows, int chunks) {
    int s = (int)((g_f1_blocks + n_windows - 1) / n_windows);  // aim for >= g_f1_blocks workgroups
    if (s < 1) s = 1;
    if (s > chunks) s = chunks;
    const int cps = (chunks + s - 1) / s;
    return (chunks + cps - 1) / cps;
}

int f1_set_blocks(int v) {
    g_f1_blocks = v;
    return 0;
}

}  // namespace vm

using namespace vm;

extern "C" int vm_conv1_fused_fwd(const float* x, const float* w, const float* bias, const float* gamma_or_scale,
                                  const float* shift, int64_t n_windows, int64_t L, int F, int pool, int inference, void* out,
                                  float* stat_sum, float* stat_sq, void* stream) {
    VM_REQUIRE(x && w && bias && gamma_or_scale && out, "vm_conv1_fused_fwd: null pointer");
    VM_REQUIRE(n_windows > 0 && L > 0 && F > 0 && F % 8 == 0, "vm_conv1_fused_fwd: bad sizes");
    VM_REQUIRE(pool == 2 || pool == 4, "vm_conv1_fused_fwd: pool must be 2 or 4 (got %d)", pool);
    VM_REQUIRE(inference ? shift != nullptr : (stat_sum && stat_sq), "vm_conv1_fused_fwd: missing shift / stat buffers");
    const int chunks = (int)((L + F1_CHUNK - 1) / F1_CHUNK);
    const int64_t gx = n_windows * chunks;
    VM_REQUIRE(gx < (1LL << 31), "vm_conv1_fused_fwd: grid too large");
    const dim3 grid((unsigned)gx, (unsigned)((F + 127) / 128));
#define VM_F1_FWD(POOL, INF)                                                                                              \
    hipLaunchKernelGGL((conv1_fused_fwd_kernel<POOL, INF>), grid, dim3(256), 0, (hipStream_t)stream, x, w, bias,           \
                       gamma_or_scale, shift, L, F, chunks, (bf16*)out, stat_sum, stat_sq)
    if (pool == 2) {
        if (inference) VM_F1_FWD(2, true); else VM_F1_FWD(2, false);
    } else {
        if (inference) VM_F1_FWD(4, true); else VM_F1_FWD(4, false);
    }
#undef VM_F1_FWD
    return check_launch("vm_conv1_fused_fwd");
}

extern "C" int64_t vm_conv1_fused_bwd_workspace_bytes(int64_t n_windows, int64_t L, int F) {
    const int chunks = (int)((L + F1_CHUNK - 1) / F1_CHUNK);
    const int64_t slabs = n_windows * f1_splits(n_windows, chunks);
    return slabs * 33 * (int64_t)F * (int64_t)sizeof(float) + slab_sum_part_bytes(33LL * F);
}

extern "C" int vm_conv1_fused_bwd(const float* x, const float* w, const float* bias, const void* dp, const float* scale,
                                  const float* mean, const float* invstd, const float* drop, const float* c1, const float* c2,
                                  int64_t n_windows, int64_t windows_per_tower, int64_t L, int F, int pool, void* ws,
                                  float* grad_w, float* grad_b, void* stream) {
    VM_REQUIRE(x && w && bias && dp && scale && mean && invstd && c1 && c2 && ws && grad_w && grad_b,
               "vm_conv1_fused_bwd: null pointer");
    VM_REQUIRE(n_windows > 0 && windows_per_tower > 0 && L > 0 && F > 0 && F % 8 == 0, "vm_conv1_fused_bwd: bad sizes");
    VM_REQUIRE(pool == 2 || pool == 4, "vm_conv1_fused_bwd: pool must be 2 or 4 (got %d)", pool);
    const int chunks = (int)((L + F1_CHUNK - 1) / F1_CHUNK);
    const int splits = f1_splits(n_windows, chunks);
    const int cps = (chunks + splits - 1) / splits;
    const int64_t gx = n_windows * splits;
    VM_REQUIRE(gx < (1LL << 31), "vm_conv1_fused_bwd: grid too large");
    const dim3 grid((unsigned)gx, (unsigned)((F + 127) / 128));
    if (pool == 2) {
        hipLaunchKernelGGL((conv1_fused_bwd_kernel<2>), grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, (const bf16*)dp,
                           scale, mean, invstd, drop, c1, c2, windows_per_tower, L, F, chunks, splits, cps, (float*)ws);
    } else {
        hipLaunchKernelGGL((conv1_fused_bwd_kernel<4>), grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, (const bf16*)dp,
                           scale, mean, invstd, drop, c1, c2, windows_per_tower, L, F, chunks, splits, cps, (float*)ws);
    }
    int rc = check_launch("vm_conv1_fused_bwd");
    if (rc) return rc;
    const int64_t nel = 33LL * F;
    return slab_sum((const float*)ws, gx, nel, grad_w, 32LL * F, grad_b, (float*)ws + gx * nel, (hipStream_t)stream);
}
