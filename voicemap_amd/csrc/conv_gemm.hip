// Conv1D(k=3, SAME) blocks 2-4 of the voicemap encoder as implicit GEMMs on the CDNA4 matrix cores.
//
// Channels-last + one zero halo row per window turns the im2col matrix into a *view*: the A row of output
// position (n, t) is the 3*C_in contiguous elements starting at padded row t, so no im2col buffer exists.
//   forward : Z[(n,t)][co]  = relu( sum_kk A[(n,t)][kk] * Wf[co][kk] + b[co] )        NT GEMM, K = 3*C_in
//   dgrad   : dX[(n,t)][ci] =       sum_kk dU[(n,t)][kk] * Wd[ci][kk]                 NT GEMM, K = 3*C_out
//   wgrad   : dW[kk][co]    =       sum_(n,t) A[(n,t)][kk] * dU[(n,t+1)][co]          TN GEMM, K = positions
// One compute core serves all three and both storage types: v_mfma_f32_32x32x16_bf16 (8 bf16 per lane) or
// v_mfma_f32_32x32x2_f32 (exact fp32, 1 float per lane).  Workgroup = 4 waves (2x2), 128x128 output tile,
// KB-byte K slices (KB = 128: 64 bf16 / 32 fp32) staged through LDS with a (KB+16)-byte row pitch (conflict-free
// ds_read_b128 fragment reads), global->register prefetch of slice k+1 overlapped with the MFMAs of slice k, two
// LDS buffers, one barrier per slice.  The MFMAs are issued with the operands swapped (D = B.A^T) so that a lane's
// four consecutive accumulator registers are four consecutive *output columns*: the epilogue moves the tile through
// LDS with 16-byte writes and leaves the workgroup as whole 16-byte, fully coalesced row segments (and the wgrad
// slab as 16-byte fp32 stores) instead of 2-byte column-strided stores.
// Tiles never straddle windows (grid = window x t-tile x n-tile), so the halo is never crossed and every row of a
// tile belongs to one BatchNorm tower.
#include <string.h>

#include <type_traits>

#include "common.hpp"

#ifndef VM_MFMA_SETPRIO
#define VM_MFMA_SETPRIO 0  // experiment: raise wave priority around MFMA clusters (build with -DVM_MFMA_SETPRIO=1)
#endif

namespace vm {

// cache-policy bits of the LDS-DMA streams (see glds16); -D overrides are for A/B builds
#if !defined(VM_NT2R_AUX_A)
#define VM_NT2R_AUX_A 0
#endif
#if !defined(VM_NT2R_AUX_B)
#define VM_NT2R_AUX_B 0
#endif
#if !defined(VM_TNX_AUX_A)
#define VM_TNX_AUX_A 0
#endif
#if !defined(VM_TNX_AUX_B)
#define VM_TNX_AUX_B 0
#endif

constexpr int BM = 128, BN = 128;

template <int KB>
struct Geo {
    static constexpr int PITCH = KB + 16;          // LDS row pitch in bytes
    static constexpr int TILE = BM * PITCH;        // one operand tile
    static constexpr int CH = KB / 16;             // 16-byte chunks per row
    static constexpr int NCHUNK = BM * CH / 256;   // chunks per thread per operand
};
constexpr int OUT_PITCH = BN * 4 + 16;  // fp32 epilogue tile row pitch (bytes)

template <typename T> struct Mfma;
template <> struct Mfma<bf16> {
    static constexpr int KSTEP_BYTES = 32;  // one 32x32x16: 16 bf16 of K
    using Frag = bf16x8;
    __device__ static inline Frag load(const char* row_ptr, int s, int kh) {
        return *reinterpret_cast<const Frag*>(row_ptr + (s * 2 + kh) * 16);
    }
    __device__ static inline f32x16 run(Frag a, Frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mfma<float> {
    static constexpr int KSTEP_BYTES = 8;  // one 32x32x2: 2 floats of K
    using Frag = float;
    __device__ static inline Frag load(const char* row_ptr, int s, int kh) {
        return *reinterpret_cast<const float*>(row_ptr + (s * 2 + kh) * 4);
    }
    __device__ static inline f32x16 run(Frag a, Frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
};

// One KB-byte K slice: every wave multiplies its 64 (m) x 64 (n) sub-tile.  lds_a / lds_b: [128][PITCH] bytes.
// acc[im][in] = B_in . A_im^T, i.e. register r of lane l holds  m = 32*im + (l&31),  n = 32*in + (r&3) + 8*(r>>2) + 4*(l>>5).
template <typename T, int KB>
__device__ inline void mma_slice(const char* lds_a, const char* lds_b, int wm, int wn, int lane, f32x16 (&acc)[2][2]) {
    constexpr int PITCH = Geo<KB>::PITCH;
    constexpr int KSTEPS = KB / Mfma<T>::KSTEP_BYTES;
    const int r = lane & 31, kh = lane >> 5;
    const char* pa0 = lds_a + (wm * 64 + r) * PITCH;
    const char* pa1 = pa0 + 32 * PITCH;
    const char* pb0 = lds_b + (wn * 64 + r) * PITCH;
    const char* pb1 = pb0 + 32 * PITCH;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
        typename Mfma<T>::Frag a0 = Mfma<T>::load(pa0, s, kh), a1 = Mfma<T>::load(pa1, s, kh);
        typename Mfma<T>::Frag b0 = Mfma<T>::load(pb0, s, kh), b1 = Mfma<T>::load(pb1, s, kh);
        acc[0][0] = Mfma<T>::run(b0, a0, acc[0][0]);
        acc[0][1] = Mfma<T>::run(b1, a0, acc[0][1]);
        acc[1][0] = Mfma<T>::run(b0, a1, acc[1][0]);
        acc[1][1] = Mfma<T>::run(b1, a1, acc[1][1]);
    }
}

// ---- fp32 storage, split-bf16 arithmetic (dtype VM_F32S) ----------------------------------------------------------------
// An fp32 operand x is staged as two bf16 values hi = bf16(x), lo = bf16(x - hi)  (x = hi + lo up to 2^-17 |x|) and a product
// a * b is formed as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on the bf16 matrix pipe with fp32 accumulation: the dropped lo*lo term
// and the representation error are both ~2^-17 relative, against 2^-9 for plain bf16 operands, at 3 MFMAs of the 16x faster
// kind instead of one fp32 MFMA.  A 128-byte K slice (32 floats) becomes a 64-byte hi plane followed by a 64-byte lo plane in
// the same LDS row, so tile sizes, pitches and the 16-byte fragment reads of the bf16 path carry over unchanged.
// (the elements are copied to scalars first: __builtin_bit_cast applied directly to an ext-vector element, v[i] or v.y, reads
// element 0 for every i with this compiler)
__device__ inline uint32_t pack_bf16x2(float a, float b) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const bf16x2_t r = {(bf16)a, (bf16)b};
    return __builtin_bit_cast(uint32_t, r);
}
__device__ inline void split_f32x4(const u32x4& v, u32x2& hi, u32x2& lo) {
    const uint32_t u0 = v[0], u1 = v[1], u2 = v[2], u3 = v[3];
    const float x0 = __builtin_bit_cast(float, u0), x1 = __builtin_bit_cast(float, u1), x2 = __builtin_bit_cast(float, u2),
                x3 = __builtin_bit_cast(float, u3);
    const uint32_t h01 = pack_bf16x2(x0, x1), h23 = pack_bf16x2(x2, x3);
    const uint32_t a0 = h01 << 16, a1 = h01 & 0xffff0000u, a2 = h23 << 16, a3 = h23 & 0xffff0000u;  // the hi halves as floats
    hi = u32x2{h01, h23};
    lo = u32x2{pack_bf16x2(x0 - __builtin_bit_cast(float, a0), x1 - __builtin_bit_cast(float, a1)),
               pack_bf16x2(x2 - __builtin_bit_cast(float, a2), x3 - __builtin_bit_cast(float, a3))};
}

// mma_slice for a 128-byte slice staged as [hi 64 B | lo 64 B] rows: two bf16 k-steps, three MFMAs per accumulator tile each,
// small terms first.
template <int KB>
__device__ inline void mma_slice_split(const char* lds_a, const char* lds_b, int wm, int wn, int lane, f32x16 (&acc)[2][2]) {
    static_assert(KB == 128, "split-bf16 staging is laid out for 128-byte slices");
    using M = Mfma<bf16>;
    constexpr int PITCH = Geo<KB>::PITCH;
    const int r = lane & 31, kh = lane >> 5;
    const char* pa0 = lds_a + (wm * 64 + r) * PITCH;
    const char* pa1 = pa0 + 32 * PITCH;
    const char* pb0 = lds_b + (wn * 64 + r) * PITCH;
    const char* pb1 = pb0 + 32 * PITCH;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const M::Frag a0h = M::load(pa0, s, kh), a1h = M::load(pa1, s, kh), b0h = M::load(pb0, s, kh), b1h = M::load(pb1, s, kh);
        const M::Frag a0l = M::load(pa0, s + 2, kh), a1l = M::load(pa1, s + 2, kh), b0l = M::load(pb0, s + 2, kh), b1l = M::load(pb1, s + 2, kh);
        acc[0][0] = M::run(b0l, a0h, acc[0][0]);
        acc[0][1] = M::run(b1l, a0h, acc[0][1]);
        acc[1][0] = M::run(b0l, a1h, acc[1][0]);
        acc[1][1] = M::run(b1l, a1h, acc[1][1]);
        acc[0][0] = M::run(b0h, a0l, acc[0][0]);
        acc[0][1] = M::run(b1h, a0l, acc[0][1]);
        acc[1][0] = M::run(b0h, a1l, acc[1][0]);
        acc[1][1] = M::run(b1h, a1l, acc[1][1]);
        acc[0][0] = M::run(b0h, a0h, acc[0][0]);
        acc[0][1] = M::run(b1h, a0h, acc[0][1]);
        acc[1][0] = M::run(b0h, a1h, acc[1][0]);
        acc[1][1] = M::run(b1h, a1h, acc[1][1]);
    }
}

__device__ inline void zero_acc(f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

// accumulators -> fp32 tile in LDS, out_tile[m][n] with OUT_PITCH-byte rows (16-byte writes, conflict-free)
__device__ inline void acc_to_lds(char* out_tile, int wm, int wn, int lane, const f32x16 (&acc)[2][2]) {
    const int hi = lane >> 5;
#pragma unroll
    for (int im = 0; im < 2; ++im) {
        const int m = wm * 64 + im * 32 + (lane & 31);
#pragma unroll
        for (int in = 0; in < 2; ++in) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = wn * 64 + in * 32 + 8 * g + 4 * hi;
                f32x4 v = {acc[im][in][4 * g], acc[im][in][4 * g + 1], acc[im][in][4 * g + 2], acc[im][in][4 * g + 3]};
                *reinterpret_cast<f32x4*>(out_tile + m * OUT_PITCH + n * 4) = v;
            }
        }
    }
}

// EPI_FWD_POOL (conv_nt2r_kernel only): forward in inference mode with the BatchNorm affine and MaxPool1D(2) applied in the epilogue
enum { EPI_FWD = 0, EPI_DGRAD = 1, EPI_FWD_POOL = 2 };

template <typename T>
struct NtArgs {
    const T* a;            // padded activations (n_windows, L+2, a_c)
    const T* bt;           // (N, Ktot)
    const float* bias;     // (N) or nullptr
    T* out;                // (n_windows, L, N)
    float* stat_sum;       // (n_windows*tilesL, N) or nullptr
    float* stat_sq;
    int64_t a_win_stride;  // elements
    int a_c;               // row stride of a (elements)
    int L, N, Ktot;
    int tilesL, tilesN;
    int order;   // tile order of the LDS-DMA kernels: 0 sequential n-tiles per workgroup, 1 concurrent n-tiles per XCD
    int korder;  // conv_nt8_kernel: 1 = walk K as (channel chunk, tap) instead of (tap, channel chunk): consecutive K tiles then
                 // re-read the same input cache lines one row later (needs a_c % 64 == 0)
    int skew;    // conv_nt8_kernel: start delay (units of 127*64 clocks) per workgroup phase (blockIdx >> 3) & 3
    int ablate;  // timing experiments only (results are wrong when != 0): 1 no epilogue stores, 2 no epilogue,
                 // 4 no MFMA, 8 no K-loop global loads after the first slice
    // dgrad + BatchNorm-backward reduce (conv_nt2r_kernel only; vm_conv_dgrad_bnred): red_a = the tensor A whose rows line up with
    // the output rows (window stride / first row in elements / rows), stat_sum / stat_sq then receive sum(out) and sum(out * A)
    const T* red_a = nullptr;
    int64_t red_a_win_stride = 0;
    int red_a_row0 = 0;
    int split = 0;  // fp32 storage only (dtype VM_F32S): split-bf16 products on the bf16 matrix pipe instead of fp32 MFMAs
    // EPI_FWD_POOL: per-channel scale / shift of the inference-mode BatchNorm; out is then the padded pooled tensor (L / 2 + 2 rows)
    const float* aff_scale = nullptr;
    const float* aff_shift = nullptr;
    // EPI_FWD (training, vm_conv_fwd_e): also write the pool-window extreme of the (2q, 2q + 1) position pairs -- the maximum where
    // aff_scale (= the BatchNorm gamma) is >= 0, the minimum where it is negative -- as an unpadded (n_windows, L / 2, N) tensor
    T* pool_e = nullptr;
};

template <int KB>
constexpr int nt_lds_bytes() {
    return (2 * 2 * Geo<KB>::TILE > BM * OUT_PITCH + 4096) ? 2 * 2 * Geo<KB>::TILE : BM * OUT_PITCH + 4096;
}

template <typename T, int EPI, int KB, bool SPLIT = false>
__global__ __launch_bounds__(256) void conv_nt_kernel(NtArgs<T> p, int64_t n_groups) {
    static_assert(!SPLIT || (sizeof(T) == 4 && KB == 128), "split-bf16 arithmetic: fp32 storage, 128-byte slices");
    using G = Geo<KB>;
    constexpr int VEC = Elem<T>::kVec;
    constexpr int BK = KB / (int)sizeof(T);
    __shared__ __attribute__((aligned(16))) char lds[nt_lds_bytes<KB>()];  // [buf][A|B] in the K loop, then the fp32 tile

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1;
    const int nk = (p.Ktot + BK - 1) / BK;

    // staging geometry that does not depend on the tile
    int lds_off[G::NCHUNK], kch[G::NCHUNK], srow[G::NCHUNK];
#pragma unroll
    for (int i = 0; i < G::NCHUNK; ++i) {
        // SPLIT: chunks 2j, 2j + 1 of a thread are neighbours in a row (8 floats), so that their bf16 halves leave as one 16-byte
        // LDS write per plane (the 8-byte writes of one chunk at a time ran at half the LDS write rate)
        const int id = SPLIT ? (tid + (i >> 1) * 256) * 2 + (i & 1) : tid + i * 256;
        srow[i] = id / G::CH;
        const int ch = id % G::CH;
        lds_off[i] = srow[i] * G::PITCH + ch * (SPLIT ? 8 : 16);  // SPLIT: 8 bytes into the hi plane, 8 into the lo plane (+64)
        kch[i] = ch * VEC;
    }

    // A workgroup walks (window, t-tile) groups and, inside a group, all n-tiles: the A tile is re-read from this
    // CU's caches.  The first K slice of the NEXT tile is requested before the epilogue of the current one, so its
    // HBM latency hides behind the LDS round trip and the stores.
    const T* a_base = nullptr;
    int a_off[G::NCHUNK], b_off[G::NCHUNK];
    u32x4 ra[G::NCHUNK], rb[G::NCHUNK];
    auto setup = [&](int64_t group, int tn) {
        const int tl = (int)(group % p.tilesL);
        const int64_t n = group / p.tilesL;
        a_base = p.a + n * p.a_win_stride;
#pragma unroll
        for (int i = 0; i < G::NCHUNK; ++i) {
            int t = tl * BM + srow[i];
            t = t < p.L ? t : p.L - 1;
            int nn = tn * BN + srow[i];
            nn = nn < p.N ? nn : p.N - 1;
            a_off[i] = t * p.a_c + kch[i];
            b_off[i] = nn * p.Ktot + kch[i];
        }
    };
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < G::NCHUNK; ++i) {
            const int kk = kt * BK + kch[i];
            if (kk < p.Ktot) {
                ra[i] = *reinterpret_cast<const u32x4*>(a_base + a_off[i] + kt * BK);
                rb[i] = *reinterpret_cast<const u32x4*>(p.bt + b_off[i] + kt * BK);
            } else {
                ra[i] = u32x4{0, 0, 0, 0};
                rb[i] = u32x4{0, 0, 0, 0};
            }
        }
    };

    int64_t group = blockIdx.x;
    int tn = 0;
    if (group < n_groups) {
        setup(group, tn);
        gload(0);
    }
    while (group < n_groups) {
        f32x16 acc[2][2];
        zero_acc(acc);
        for (int kt = 0; kt < nk; ++kt) {
            char* ta = lds + ((kt & 1) * 2 + 0) * G::TILE;
            char* tb = lds + ((kt & 1) * 2 + 1) * G::TILE;
#pragma unroll
            for (int i = 0; i < G::NCHUNK; ++i) {
                if constexpr (SPLIT) {
                    if (i & 1) {
                        u32x2 h0, l0, h1, l1;
                        split_f32x4(ra[i - 1], h0, l0);
                        split_f32x4(ra[i], h1, l1);
                        *reinterpret_cast<u32x4*>(ta + lds_off[i - 1]) = u32x4{h0[0], h0[1], h1[0], h1[1]};
                        *reinterpret_cast<u32x4*>(ta + lds_off[i - 1] + 64) = u32x4{l0[0], l0[1], l1[0], l1[1]};
                        split_f32x4(rb[i - 1], h0, l0);
                        split_f32x4(rb[i], h1, l1);
                        *reinterpret_cast<u32x4*>(tb + lds_off[i - 1]) = u32x4{h0[0], h0[1], h1[0], h1[1]};
                        *reinterpret_cast<u32x4*>(tb + lds_off[i - 1] + 64) = u32x4{l0[0], l0[1], l1[0], l1[1]};
                    }
                } else {
                    *reinterpret_cast<u32x4*>(ta + lds_off[i]) = ra[i];
                    *reinterpret_cast<u32x4*>(tb + lds_off[i]) = rb[i];
                }
            }
            __syncthreads();
            if (kt + 1 < nk && !(p.ablate & 8)) gload(kt + 1);
            if constexpr (SPLIT) {
                mma_slice_split<KB>(ta, tb, wm, wn, lane, acc);
            } else {
                if (!(p.ablate & 4)) mma_slice<T, KB>(ta, tb, wm, wn, lane, acc);
            }
        }
        // coordinates of the finished tile
        const int tl = (int)(group % p.tilesL);
        const int64_t n = group / p.tilesL;
        const int t0 = tl * BM, n0 = tn * BN;
        // advance and prefetch
        if (++tn == p.tilesN) {
            tn = 0;
            group += gridDim.x;
        }
        if (group < n_groups) {
            setup(group, tn);
            gload(0);
        }

        // ---- epilogue: accumulators -> fp32 LDS tile -> (bias, ReLU, convert) -> 16-byte coalesced row segments ----
        __syncthreads();
        if (p.ablate & 2) continue;
        acc_to_lds(lds, wm, wn, lane, acc);
        __syncthreads();
        const int c8 = tid & 15, rg = tid >> 4;  // 8-column chunk, row group
        const int ncol = n0 + c8 * 8;
        const bool cok = ncol < p.N;  // N is a multiple of 8: a chunk is entirely in or out
        float bias8[8], s8[8], q8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            bias8[i] = (EPI == EPI_FWD && cok) ? p.bias[ncol + i] : 0.f;
            s8[i] = 0.f;
            q8[i] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = rg + 16 * j;
            const int t = t0 + row;
            if (cok && t < p.L) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(lds + row * OUT_PITCH + c8 * 32);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(lds + row * OUT_PITCH + c8 * 32 + 16);
                const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                Vec16<T> o0, o1;  // 8 outputs: one 16-byte vector for bf16, two for fp32
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float x = v[i] + bias8[i];
                    if (EPI == EPI_FWD) x = x > 0.f ? x : 0.f;
                    const T tx = Elem<T>::from_f(x);
                    if (EPI == EPI_FWD) {
                        const float xr = Elem<T>::to_f(tx);
                        s8[i] += xr;
                        q8[i] += xr * xr;
                    }
                    if (sizeof(T) == 2) {
                        o0.set(i, x);
                    } else if (i < 4) {
                        o0.set(i, x);
                    } else {
                        o1.set(i - 4, x);
                    }
                }
                T* dst = p.out + (n * p.L + t) * (int64_t)p.N + ncol;
                if (!(p.ablate & 1)) {
                    store16<T>(dst, o0);
                    if (sizeof(T) == 4) store16<T>(dst + 4, o1);
                }
            }
        }
        if (EPI == EPI_FWD && p.stat_sum != nullptr) {
            // the 4 row groups of a wave (lanes 0-15, 16-31, 32-47, 48-63) hold the same column chunk
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                s8[i] += __shfl_xor(s8[i], 16, 64);
                s8[i] += __shfl_xor(s8[i], 32, 64);
                q8[i] += __shfl_xor(q8[i], 16, 64);
                q8[i] += __shfl_xor(q8[i], 32, 64);
            }
            float* red = reinterpret_cast<float*>(lds + BM * OUT_PITCH);  // [4 waves][2][128]
            if (lane < 16) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    red[(w * 2 + 0) * 128 + c8 * 8 + i] = s8[i];
                    red[(w * 2 + 1) * 128 + c8 * 8 + i] = q8[i];
                }
            }
            __syncthreads();
            if (tid < 128 && n0 + tid < p.N) {
                const int64_t row = n * p.tilesL + tl;
                p.stat_sum[row * p.N + n0 + tid] =
                    (red[0 * 128 + tid] + red[2 * 128 + tid]) + (red[4 * 128 + tid] + red[6 * 128 + tid]);
                p.stat_sq[row * p.N + n0 + tid] =
                    (red[1 * 128 + tid] + red[3 * 128 + tid]) + (red[5 * 128 + tid] + red[7 * 128 + tid]);
            }
        }
        __syncthreads();  // the fp32 tile is consumed before the next tile's K loop overwrites the buffers
    }
}

// ------------------------------------------------------------------------------------------------
// NT GEMM, direct-to-LDS variant.  The register-staged kernel above is bound by its LDS *writes*
// (ds_write_b128 sustains ~79 B/clk/CU: 32 KB per 128-byte slice = ~415 clk, more than the 16 MFMAs per wave it feeds;
// measured: removing the MFMAs saves 10 % of its time, removing nothing else saves more).  Here the K slices go from
// global memory straight into LDS with global_load_lds_dwordx4 (no VGPR round trip, no ds_write).  An LDS-DMA writes
// wave-uniform base + lane*16, so the tile is stored UNPADDED ([128 rows][KB bytes]) and bank conflicts of the
// ds_read_b128 fragment reads are removed by an XOR swizzle of the 16-byte chunk index that is applied to the per-lane
// SOURCE address and again when reading:
//     KB = 128: chunk' = chunk ^ ((row >> 1) & 7)      (two rows per 256-byte bank line)
//     KB =  64: chunk' = chunk ^ ((row >> 2) & 3)      (four rows per bank line)
// Requires K*sizeof(T) to be a multiple of KB (no zero-filled K tail); other shapes use the kernel above.
// 16 bytes per lane from global memory straight into LDS at (wave-uniform) lds_base + lane*16.  The body only exists in
// the device pass: the host pass of hipcc cannot type-check the LDS address-space cast and would silently drop the
// kernel's launch stub.
// AUX: cache-policy bits of the instruction (1 = sc0, 2 = nt); all streams use 0.  Measured (same-box A/B of builds, -DVM_*_AUX_*):
// nt on the forward / dgrad kernels' streams +34 % (their half-line pieces and tap re-reads live on cache hits); nt on the wgrad
// kernel's streams -1 % for wgrad and +2 % for the step (the kernels that follow lose their hits); sc0 anywhere: no change.  (A build
// with nt everywhere showed wgrad at -5 %: the slower neighbours let the chip clock higher -- not a property of the kernel.)
template <int AUX = 0>
__device__ inline void glds16(const char* gsrc, char* lds_base) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds(gsrc, (__attribute__((address_space(3))) void*)lds_base, 16, 0, AUX);
#endif
}

template <int KB>
__device__ inline int swz(int row, int chunk) {
    return KB == 128 ? (chunk ^ ((row >> 1) & 7)) : (chunk ^ ((row >> 2) & 3));
}

template <typename T, int KB>
__device__ inline void mma_slice_swz(const char* lds_a, const char* lds_b, int wm, int wn, int lane, f32x16 (&acc)[2][2]) {
    constexpr int KSTEPS = KB / Mfma<T>::KSTEP_BYTES;
    constexpr int SUB = 16 / (Mfma<T>::KSTEP_BYTES / 2);  // fragments per 16-byte chunk: 1 (bf16), 4 (fp32)
    using Frag = typename Mfma<T>::Frag;
    const int r = lane & 31, kh = lane >> 5;
    const int ra0 = wm * 64 + r, ra1 = ra0 + 32, rb0 = wn * 64 + r, rb1 = rb0 + 32;
    auto frag = [&](const char* base, int row, int s) -> Frag {
        const int f = s * 2 + kh;  // fragment index along K
        return *reinterpret_cast<const Frag*>(base + row * KB + swz<KB>(row, f / SUB) * 16 + (f % SUB) * (16 / SUB));
    };
    // software-pipelined by one k-step: the fragments of step s+1 are requested before the MFMAs of step s are issued,
    // so an LDS read has a whole k-step (4 MFMAs = 128 cycles) plus the other wave's time to return
    Frag a0 = frag(lds_a, ra0, 0), a1 = frag(lds_a, ra1, 0), b0 = frag(lds_b, rb0, 0), b1 = frag(lds_b, rb1, 0);
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
        Frag na0 = a0, na1 = a1, nb0 = b0, nb1 = b1;
        if (s + 1 < KSTEPS) {
            na0 = frag(lds_a, ra0, s + 1);
            na1 = frag(lds_a, ra1, s + 1);
            nb0 = frag(lds_b, rb0, s + 1);
            nb1 = frag(lds_b, rb1, s + 1);
        }
#if VM_MFMA_SETPRIO
        __builtin_amdgcn_s_setprio(1);
#endif
        acc[0][0] = Mfma<T>::run(b0, a0, acc[0][0]);
        acc[0][1] = Mfma<T>::run(b1, a0, acc[0][1]);
        acc[1][0] = Mfma<T>::run(b0, a1, acc[1][0]);
        acc[1][1] = Mfma<T>::run(b1, a1, acc[1][1]);
#if VM_MFMA_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        a0 = na0;
        a1 = na1;
        b0 = nb0;
        b1 = nb1;
    }
}

template <int KB>
constexpr int glds_lds_bytes() {
    return (2 * 2 * BM * KB > BM * OUT_PITCH + 4096) ? 2 * 2 * BM * KB : BM * OUT_PITCH + 4096;
}

// Shared epilogue of the NT kernels: accumulators -> fp32 LDS tile -> (bias, ReLU, convert, BN partial statistics) ->
// 16-byte coalesced row segments.  `lds` must hold BM*OUT_PITCH + 4096 bytes; callers barrier before and after.
template <typename T, int EPI>
__device__ inline void nt_epilogue(const NtArgs<T>& p, char* lds, const f32x16 (&acc)[2][2], int64_t n, int tl, int t0, int n0,
                                   int tid, int lane, int w, int wm, int wn) {
    __syncthreads();
    acc_to_lds(lds, wm, wn, lane, acc);
    __syncthreads();
    const int c8 = tid & 15, rg = tid >> 4;
    const int ncol = n0 + c8 * 8;
    const bool cok = ncol < p.N;
    float bias8[8], s8[8], q8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        bias8[i] = (EPI == EPI_FWD && cok) ? p.bias[ncol + i] : 0.f;
        s8[i] = 0.f;
        q8[i] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = rg + 16 * j;
        const int t = t0 + row;
        if (cok && t < p.L) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(lds + row * OUT_PITCH + c8 * 32);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(lds + row * OUT_PITCH + c8 * 32 + 16);
            const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            Vec16<T> o0, o1;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float x = v[i] + bias8[i];
                if (EPI == EPI_FWD) x = x > 0.f ? x : 0.f;
                const T tx = Elem<T>::from_f(x);
                if (EPI == EPI_FWD) {
                    const float xr = Elem<T>::to_f(tx);
                    s8[i] += xr;
                    q8[i] += xr * xr;
                }
                if (sizeof(T) == 2) {
                    o0.set(i, x);
                } else if (i < 4) {
                    o0.set(i, x);
                } else {
                    o1.set(i - 4, x);
                }
            }
            T* dst = p.out + (n * p.L + t) * (int64_t)p.N + ncol;
            store16<T>(dst, o0);
            if (sizeof(T) == 4) store16<T>(dst + 4, o1);
        }
    }
    if (EPI == EPI_FWD && p.stat_sum != nullptr) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            s8[i] += __shfl_xor(s8[i], 16, 64);
            s8[i] += __shfl_xor(s8[i], 32, 64);
            q8[i] += __shfl_xor(q8[i], 16, 64);
            q8[i] += __shfl_xor(q8[i], 32, 64);
        }
        float* red = reinterpret_cast<float*>(lds + BM * OUT_PITCH);
        if (lane < 16) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                red[(w * 2 + 0) * 128 + c8 * 8 + i] = s8[i];
                red[(w * 2 + 1) * 128 + c8 * 8 + i] = q8[i];
            }
        }
        __syncthreads();
        if (tid < 128 && n0 + tid < p.N) {
            const int64_t row = n * p.tilesL + tl;
            p.stat_sum[row * p.N + n0 + tid] =
                (red[0 * 128 + tid] + red[2 * 128 + tid]) + (red[4 * 128 + tid] + red[6 * 128 + tid]);
            p.stat_sq[row * p.N + n0 + tid] =
                (red[1 * 128 + tid] + red[3 * 128 + tid]) + (red[5 * 128 + tid] + red[7 * 128 + tid]);
        }
    }
}

// Epilogue variant that stages the tile in the STORAGE type: bias + ReLU + convert happen in registers (the swapped MFMA
// layout gives every lane 4 consecutive output columns, so the bias is one float4 per register group), the LDS tile is
// [128][128*sizeof(T) + 16] (35 KB for bf16 instead of 68 KB of fp32) and the read-back is a pure 16-byte copy plus the
// BatchNorm partial sums.  With 64-byte K slices this brings the workgroup's LDS to 39 KB: three workgroups per CU.
template <typename T>
constexpr int tile_pitch_t() { return BN * (int)sizeof(T) + 16; }

template <typename T, int EPI>
__device__ inline void nt_epilogue_t(const NtArgs<T>& p, char* lds, const f32x16 (&acc)[2][2], int64_t n, int tl, int t0, int n0,
                                     int tid, int lane, int w, int wm, int wn) {
    constexpr int TP = tile_pitch_t<T>();
    __syncthreads();
    {
        const int hi = lane >> 5;
#pragma unroll
        for (int in = 0; in < 2; ++in) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = wn * 64 + in * 32 + 8 * g + 4 * hi;
                f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
                if (EPI == EPI_FWD && n0 + nl < p.N) b4 = *reinterpret_cast<const f32x4*>(p.bias + n0 + nl);
#pragma unroll
                for (int im = 0; im < 2; ++im) {
                    const int m = wm * 64 + im * 32 + (lane & 31);
                    T o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float x = acc[im][in][4 * g + j] + b4[j];
                        if (EPI == EPI_FWD) x = x > 0.f ? x : 0.f;
                        o[j] = Elem<T>::from_f(x);
                    }
                    char* dst = lds + m * TP + nl * (int)sizeof(T);
                    if (sizeof(T) == 2) {
                        *reinterpret_cast<u32x2*>(dst) = *reinterpret_cast<const u32x2*>(o);
                    } else {
                        *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(o);
                    }
                }
            }
        }
    }
    __syncthreads();
    const int c8 = tid & 15, rg = tid >> 4;
    const int ncol = n0 + c8 * 8;
    const bool cok = ncol < p.N;
    float s8[8], q8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        s8[i] = 0.f;
        q8[i] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = rg + 16 * j;
        const int t = t0 + row;
        if (cok && t < p.L) {
            T* dst = p.out + (n * p.L + t) * (int64_t)p.N + ncol;
            const char* src = lds + row * TP + c8 * 8 * (int)sizeof(T);
            const Vec16<T> v0 = *reinterpret_cast<const Vec16<T>*>(src);
            store16<T>(dst, v0);
            if (sizeof(T) == 4) {
                const Vec16<T> v1 = *reinterpret_cast<const Vec16<T>*>(src + 16);
                store16<T>(dst + 4, v1);
                if (EPI == EPI_FWD) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        s8[i] += v0.get(i);
                        q8[i] += v0.get(i) * v0.get(i);
                        s8[4 + i] += v1.get(i);
                        q8[4 + i] += v1.get(i) * v1.get(i);
                    }
                }
            } else if (EPI == EPI_FWD) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float xr = v0.get(i);
                    s8[i] += xr;
                    q8[i] += xr * xr;
                }
            }
        }
    }
    if (EPI == EPI_FWD && p.stat_sum != nullptr) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            s8[i] += __shfl_xor(s8[i], 16, 64);
            s8[i] += __shfl_xor(s8[i], 32, 64);
            q8[i] += __shfl_xor(q8[i], 16, 64);
            q8[i] += __shfl_xor(q8[i], 32, 64);
        }
        float* red = reinterpret_cast<float*>(lds + BM * TP);
        if (lane < 16) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                red[(w * 2 + 0) * 128 + c8 * 8 + i] = s8[i];
                red[(w * 2 + 1) * 128 + c8 * 8 + i] = q8[i];
            }
        }
        __syncthreads();
        if (tid < 128 && n0 + tid < p.N) {
            const int64_t row = n * p.tilesL + tl;
            p.stat_sum[row * p.N + n0 + tid] =
                (red[0 * 128 + tid] + red[2 * 128 + tid]) + (red[4 * 128 + tid] + red[6 * 128 + tid]);
            p.stat_sq[row * p.N + n0 + tid] =
                (red[1 * 128 + tid] + red[3 * 128 + tid]) + (red[5 * 128 + tid] + red[7 * 128 + tid]);
        }
    }
}

template <typename T, int KB>
constexpr int glds_t_lds_bytes() {
    return (2 * 2 * BM * KB > BM * tile_pitch_t<T>() + 4096) ? 2 * 2 * BM * KB : BM * tile_pitch_t<T>() + 4096;
}

template <typename T, int EPI, int KB, bool TEPI = false>
__global__ __launch_bounds__(256, (TEPI && sizeof(T) == 2) ? 3 : 2) void conv_nt_glds_kernel(NtArgs<T> p, int64_t n_groups) {
    constexpr int BK = KB / (int)sizeof(T);
    constexpr int RPI = 1024 / KB;        // rows per wave-instruction (64 lanes x 16 bytes)
    constexpr int CPR = KB / 16;          // chunks per row
    constexpr int NI = BM / RPI / 4;      // instructions per wave per operand per slice
    constexpr int OPB = BM * KB;          // bytes of one operand tile
    __shared__ __attribute__((aligned(16))) char lds[TEPI ? glds_t_lds_bytes<T, KB>() : glds_lds_bytes<KB>()];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1;
    const int nk = p.Ktot / BK;

    // lane -> (row within the instruction's rows, physical chunk); the source chunk undoes the swizzle
    int srow[NI], src_chunk_bytes[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        srow[i] = (w + 4 * i) * RPI + lane / CPR;
        src_chunk_bytes[i] = swz<KB>(srow[i], lane % CPR) * 16;
    }

    // Tile order.  order 1 (default when the group count is a multiple of 8): the n-tiles of one (window, t-tile) group --
    // which all stream the SAME A rows -- are given to workgroups that run at the same time on the same XCD
    // (workgroup b is dispatched to XCD b % 8), so the A tile is fetched into that L2 once instead of once per n-tile
    // (PMC: FETCH_SIZE of the forward launches was 4.5x the algorithmic bytes with the sequential order 0).
    const int64_t total_tiles = n_groups * p.tilesN;
    const bool xcd_order = p.order == 1 && (n_groups & 7) == 0;
    for (int64_t it = 0;; ++it) {
        int64_t group;
        int tn;
        if (xcd_order) {
            const int64_t v = blockIdx.x + it * gridDim.x;
            if (v >= total_tiles) break;
            const int64_t j = v >> 3;
            tn = (int)(j % p.tilesN);
            group = (j / p.tilesN) * 8 + (v & 7);
        } else {
            group = blockIdx.x + (it / p.tilesN) * gridDim.x;
            tn = (int)(it % p.tilesN);
            if (group >= n_groups) break;
        }
        const int tl = (int)(group % p.tilesL);
        const int64_t n = group / p.tilesL;
        const int t0 = tl * BM;
        const char* a_rows[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int t = t0 + srow[i];
            t = t < p.L ? t : p.L - 1;
            a_rows[i] = reinterpret_cast<const char*>(p.a + n * p.a_win_stride + (int64_t)t * p.a_c) + src_chunk_bytes[i];
        }
        {
            const int n0 = tn * BN;
            const char* b_rows[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                int nn = n0 + srow[i];
                nn = nn < p.N ? nn : p.N - 1;
                b_rows[i] = reinterpret_cast<const char*>(p.bt + (int64_t)nn * p.Ktot) + src_chunk_bytes[i];
            }
            // K slice kt -> byte offset in an (im2col / weight) row; korder: (channel chunk, tap) instead of (tap, channel chunk),
            // so that consecutive slices re-read the same input cache lines one row later
            const int row_bytes = p.a_c * (int)sizeof(T);
            auto issue = [&](int kt) {
                char* bufbase = lds + (kt & 1) * 2 * OPB;
                const int64_t ko = p.korder ? (int64_t)(kt % 3) * row_bytes + (int64_t)(kt / 3) * KB : (int64_t)kt * KB;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int dst = __builtin_amdgcn_readfirstlane((w + 4 * i) * 1024);
                    glds16(a_rows[i] + ko, bufbase + dst);
                    glds16(b_rows[i] + ko, bufbase + OPB + dst);
                }
            };

            f32x16 acc[2][2];
            zero_acc(acc);
            issue(0);
            for (int kt = 0; kt < nk; ++kt) {
                __syncthreads();  // slice kt has landed for every wave (the compiler drains vmcnt before the barrier)
                if (kt + 1 < nk) issue(kt + 1);
                const char* bufbase = lds + (kt & 1) * 2 * OPB;
                mma_slice_swz<T, KB>(bufbase, bufbase + OPB, wm, wn, lane, acc);
            }

            // ---- epilogue ----
            if (TEPI) {
                nt_epilogue_t<T, EPI>(p, lds, acc, n, tl, t0, n0, tid, lane, w, wm, wn);
            } else {
                nt_epilogue<T, EPI>(p, lds, acc, n, tl, t0, n0, tid, lane, w, wm, wn);
            }
            __syncthreads();  // the fp32 tile is consumed before the next tile's DMA overwrites the buffers
        }
    }
}

// ------------------------------------------------------------------------------------------------
// NT GEMM, ring-pipelined direct-to-LDS variant.  With two buffers the DMA of slice k+1 only has the MFMAs of slice k to
// land, and measurements say the K loop is bound by that latency (HBM-class ~2 us under load), not by MFMA or LDS
// throughput.  Here a ring of RING = 4 buffers of 64-byte slices keeps RING-1 = 3 slices in flight: the wait before the
// barrier is a *counted* s_waitcnt vmcnt(N) (N = DMAs of the slices that may still be flying) and the barrier is the raw
// s_barrier -- __syncthreads() would make hipcc drain vmcnt to 0 and collapse the pipeline to depth 1.
// Ordering argument: the DMA of slice kt+3 overwrites the buffer read in iteration kt-1; every wave has finished those
// reads (their MFMAs consumed them) before it arrives at barrier kt, and the DMA is issued after that barrier.
template <int N>
__device__ inline void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <typename T, int EPI>
__global__ __launch_bounds__(256) void conv_nt_ring_kernel(NtArgs<T> p, int64_t n_groups) {
    constexpr int KB = 64, RING = 4;
    constexpr int BK = KB / (int)sizeof(T);
    constexpr int RPI = 1024 / KB;        // 16 rows per wave-instruction
    constexpr int CPR = KB / 16;          // 4 chunks per row
    constexpr int NI = BM / RPI / 4;      // 2 instructions per wave per operand per slice
    constexpr int G = 2 * NI;             // DMAs per wave per slice
    constexpr int OPB = BM * KB;          // 8 KB per operand tile
    constexpr int LDS_BYTES = (RING * 2 * OPB > BM * OUT_PITCH + 4096) ? RING * 2 * OPB : BM * OUT_PITCH + 4096;
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1;
    const int nk = p.Ktot / BK;

    int srow[NI], src_chunk_bytes[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        srow[i] = (w + 4 * i) * RPI + lane / CPR;
        src_chunk_bytes[i] = swz<KB>(srow[i], lane % CPR) * 16;
    }

    // Tile order.  order 1 (default when the group count is a multiple of 8): the n-tiles of one (window, t-tile) group --
    // which all stream the SAME A rows -- are given to workgroups that run at the same time on the same XCD
    // (workgroup b is dispatched to XCD b % 8), so the A tile is fetched into that L2 once instead of once per n-tile
    // (PMC: FETCH_SIZE of the forward launches was 4.5x the algorithmic bytes with the sequential order 0).
    const int64_t total_tiles = n_groups * p.tilesN;
    const bool xcd_order = p.order == 1 && (n_groups & 7) == 0;
    for (int64_t it = 0;; ++it) {
        int64_t group;
        int tn;
        if (xcd_order) {
            const int64_t v = blockIdx.x + it * gridDim.x;
            if (v >= total_tiles) break;
            const int64_t j = v >> 3;
            tn = (int)(j % p.tilesN);
            group = (j / p.tilesN) * 8 + (v & 7);
        } else {
            group = blockIdx.x + (it / p.tilesN) * gridDim.x;
            tn = (int)(it % p.tilesN);
            if (group >= n_groups) break;
        }
        const int tl = (int)(group % p.tilesL);
        const int64_t n = group / p.tilesL;
        const int t0 = tl * BM;
        const char* a_rows[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int t = t0 + srow[i];
            t = t < p.L ? t : p.L - 1;
            a_rows[i] = reinterpret_cast<const char*>(p.a + n * p.a_win_stride + (int64_t)t * p.a_c) + src_chunk_bytes[i];
        }
        {
            const int n0 = tn * BN;
            const char* b_rows[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                int nn = n0 + srow[i];
                nn = nn < p.N ? nn : p.N - 1;
                b_rows[i] = reinterpret_cast<const char*>(p.bt + (int64_t)nn * p.Ktot) + src_chunk_bytes[i];
            }
            auto issue = [&](int kt) {
                char* bufbase = lds + (kt % RING) * 2 * OPB;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int dst = __builtin_amdgcn_readfirstlane((w + 4 * i) * 1024);
                    glds16(a_rows[i] + (int64_t)kt * KB, bufbase + dst);
                    glds16(b_rows[i] + (int64_t)kt * KB, bufbase + OPB + dst);
                }
            };
            f32x16 acc[2][2];
            zero_acc(acc);
#pragma unroll
            for (int k0 = 0; k0 < RING - 1; ++k0)
                if (k0 < nk) issue(k0);
            for (int kt = 0; kt < nk; ++kt) {
                const int ahead = nk - 1 - kt;  // slices after kt that have been issued and may still be in flight
                if (ahead >= RING - 2) {
                    wait_vmcnt<(RING - 2) * G>();
                } else if (ahead == 1) {
                    wait_vmcnt<G>();
                } else {
                    wait_vmcnt<0>();
                }
                __builtin_amdgcn_s_barrier();
                if (kt + RING - 1 < nk) issue(kt + RING - 1);
                const char* bufbase = lds + (kt % RING) * 2 * OPB;
                mma_slice_swz<T, KB>(bufbase, bufbase + OPB, wm, wn, lane, acc);
            }
            nt_epilogue<T, EPI>(p, lds, acc, n, tl, t0, n0, tid, lane, w, wm, wn);
            __syncthreads();  // the fp32 tile is consumed before the next tile's DMA overwrites the buffers
        }
    }
}

// ------------------------------------------------------------------------------------------------
// NT GEMM, 256 x 256 output tile, 8 waves (2 x 4, 128 x 64 each), phase-interleaved LDS-DMA pipeline (bf16 only).
//
// The 128^2 kernels above are bound by their issue structure (one DMA wait + barrier per K slice; every technique that
// keeps that structure measured within noise).  This kernel changes the structure:
//  * A K tile of 64 elements is staged as four 16 KB half tiles (A_lo / A_hi: rows 0-63 / 64-127 of every wave's
//    128 rows; B_c0 / B_c1: columns 0-31 / 32-63 of every wave's 64 columns: a half tile is what ONE phase reads);
//    two K tiles are resident (128 KB).  A K tile is computed in four phases (one 64 x 32 quadrant of every wave's
//    128 x 64 sub-tile each); every phase is a READ slot (fragment ds_reads + the DMA of one half tile of a K tile up to
//    two ahead) and an MFMA slot (8 x v_mfma_f32_32x32x16_bf16), each closed by a raw s_barrier.
//  * The waves of rows 128-255 (waves 4-7: they share SIMDs pairwise with waves 0-3) run ONE SLOT BEHIND the others (one
//    extra barrier up front, one less at the end): while one wave of a SIMD feeds the matrix pipe the other one does its
//    LDS reads and address arithmetic, and s_setprio keeps the MFMA wave ahead.
//  * The only DMA wait is a counted s_waitcnt vmcnt(6) once per K tile, so three half tiles stay in flight across
//    barriers; the stream of K tiles runs across output tiles, so the first K tiles of the next output tile are landing
//    while a wave runs its (wave-private, barrier-free) epilogue.
// Stream position g (K tile g of this workgroup's stream) lives in buffer g & 1.  Phases of K tile g:
//   phase 0: read A_lo, B_c0 | DMA B0(g+1) | MFMA A_lo x B_c0
//   phase 1: read B_c1       | DMA A0(g+2) | MFMA A_lo x B_c1
//   phase 2: read A_hi       | DMA B1(g+2) | MFMA A_hi x B_c1
//   phase 3: read B_c0       | DMA A1(g+2) | vmcnt(6): all of K tile g+1 has landed | MFMA A_hi x B_c0
// (A_lo / A_hi: the wave's rows 0-63 / 64-127; B_c0 / B_c1: its columns 0-31 / 32-63.)
// Slots: group y (0: waves 0-3, 1: waves 4-7) runs READ(p) in slot 2p+y and MFMA(p) in slot 2p+y+1.
// WAR: every READ slot ends with lgkmcnt(0) before its barrier, so the fragment reads of phase p are complete at the end
//   of slot 2p (group 0) / 2p+1 (group 1); a half tile last read in phase p is re-staged in READ(p+1): slots 2p+2 / 2p+3.
//   A0(g) is last read in phase 0 -> re-staged in phase 1; B1(g): 1 -> 2; A1(g): 2 -> 3; B0(g): 3 -> phase 0 of g+1.
// RAW: group y waits (counted) for its own DMAs of K tile g+1 at the end of READ(g,3) = slot 8g+6+y; the first reads of
//   K tile g+1 are in slot 8g+8, after the barriers closing both of those slots.
// ------------------------------------------------------------------------------------------------
namespace p8 {
constexpr int HALF = 128 * 128;        // 16 KB: 128 rows x 128 bytes (64 bf16 of K)
constexpr int BUF = 4 * HALF;          // A0 A1 B0 B1
constexpr int SCR = 4096;              // per-wave epilogue scratch: 32 rows x 32 fp32
constexpr int LDS_BYTES = 2 * BUF + 8 * SCR;  // 160 KB: one workgroup per CU
enum { H_A0 = 0, H_A1 = 1, H_B0 = 2, H_B1 = 3 };
typedef bf16x8 FragA[2][4];
typedef bf16x8 FragB[4];
}  // namespace p8

template <int EPI, int PH>
__global__ __launch_bounds__(512) void conv_nt8_kernel(NtArgs<bf16> p, int n_groups) {
    using namespace p8;
    __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 2, wn = w & 3;
    const int nk = p.Ktot / 64;

    // ---- this workgroup's tiles (same XCD-concurrent order as the kernels above; here tilesL counts 256-row tiles) ----
    const int total_tiles = n_groups * p.tilesN;
    if ((int)blockIdx.x >= total_tiles) return;
    const int my_tiles = (total_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1;
    const int G = my_tiles * nk;  // K tiles in the stream
    const bool xcd_order = p.order == 1 && (n_groups & 7) == 0;
    auto decode = [&](int it, int& n, int& tl, int& tn) {
        const unsigned v = blockIdx.x + (unsigned)it * gridDim.x;
        unsigned group;
        if (xcd_order) {
            const unsigned j = v >> 3;
            tn = (int)(j % (unsigned)p.tilesN);
            group = (j / (unsigned)p.tilesN) * 8 + (v & 7);
        } else {
            tn = (int)(v % (unsigned)p.tilesN);
            group = v / (unsigned)p.tilesN;
        }
        // wave-uniform by construction; say so, so that tile coordinates live in SGPRs and the DMA addresses use the
        // scalar-base form (the divisions above are done on the vector ALU)
        tn = __builtin_amdgcn_readfirstlane(tn);
        tl = __builtin_amdgcn_readfirstlane((int)(group % (unsigned)p.tilesL));
        n = __builtin_amdgcn_readfirstlane((int)(group / (unsigned)p.tilesL));
    };

    // ---- DMA geometry: a half tile is 2 rounds of 64 rows; wave w fills rows 8w..8w+7 of a round (lane-linear) ----
    const int rr = w * 8 + (lane >> 3);
    const int schunk = ((lane & 7) ^ (((w & 1) << 2) | (lane >> 4))) * 16;  // source chunk that undoes swz<128>
    const char* const a_base = reinterpret_cast<const char*>(p.a);
    const char* const b_base = reinterpret_cast<const char*>(p.bt);
    const int a_pitch = p.a_c * 2, b_pitch = p.Ktot * 2;
    const int64_t a_win = p.a_win_stride * 2;
    // A half h holds, for both row groups, the rows read in the same phase: LDS row q <-> tile row (q>>6)*128 + h*64 + (q&63);
    // B half h likewise: LDS row q <-> tile column (q>>5)*64 + h*32 + (q&31).  (The WAR schedule above is per half tile.)
    const int b_col = (rr >> 5) * 64 + (rr & 31);
    // byte offset of K tile kt inside an (im2col / weight) row
    auto koff = [&](int kt) { return p.korder ? ((kt % 3) * p.a_c + (kt / 3) * 64) * 2 : kt * 128; };
    auto stage = [&](int h, int buf, int n, int t0, int n0, int kt) {
        char* dst = lds + buf * BUF + h * HALF + w * 1024;
        if (h < 2) {
            const char* src = a_base + n * a_win + koff(kt);  // wave-uniform
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                int t = t0 + j * 128 + h * 64 + rr;
                t = t < p.L ? t : p.L - 1;
                glds16(src + ((unsigned)t * (unsigned)a_pitch + (unsigned)schunk), dst + j * 8192);
            }
        } else {
            const char* src = b_base + (int64_t)(n0 + (h - 2) * 32) * b_pitch + koff(kt);  // wave-uniform
#pragma unroll
            for (int j = 0; j < 2; ++j)
                glds16(src + ((unsigned)(b_col + j * 128) * (unsigned)b_pitch + (unsigned)schunk), dst + j * 8192);
        }
    };

    // ---- fragment geometry ----
    const int r = lane & 31, kh = lane >> 5;
    int off[4];  // byte offset of k-step s of row r inside a 32-row block (swizzled chunk)
#pragma unroll
    for (int s = 0; s < 4; ++s) off[s] = r * 128 + (((s * 2 + kh) ^ ((r >> 1) & 7)) * 16);
    const int a_rows = wm * 64 * 128;  // this wave's 64 rows inside an A half tile
    const int b_rows = wn * 32 * 128;  // its 32 columns inside a B half tile
    auto read_a = [&](FragA& fa, int buf, int ih) {
        if (p.ablate & 16) return;
        const char* base = lds + buf * BUF + ih * HALF + a_rows;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < 4; ++s) fa[i][s] = *reinterpret_cast<const bf16x8*>(base + i * 32 * 128 + off[s]);
    };
    auto read_b = [&](FragB& fb, int buf, int jn) {
        if (p.ablate & 16) return;
        const char* base = lds + buf * BUF + (2 + jn) * HALF + b_rows;
#pragma unroll
        for (int s = 0; s < 4; ++s) fb[s] = *reinterpret_cast<const bf16x8*>(base + off[s]);
    };
    auto mma = [&](const FragA& fa, const FragB& fb, f32x16& c0, f32x16& c1) {
        if (p.ablate & 4) return;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[s], fa[0][s], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[s], fa[1][s], c1, 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    auto slot_end = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto read_done = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };

    // ---- tile state ----
    int n_c, n_n = 0, tl_c, tn_c, tl_n = 0, tn_n = 0;
    decode(0, n_c, tl_c, tn_c);
    if (my_tiles > 1) decode(1, n_n, tl_n, tn_n);
    int it = 0, kt = 0;
    // DMA of half h of stream position g + d (d = 1, 2): current tile or the next one
    const int abl = p.ablate;  // timing experiments only: 2 no epilogue, 4 no MFMA, 8 no in-loop DMA, 16 no fragment reads
    auto stage_ahead = [&](int h, int g, int d) {
        if (g + d >= G || (abl & 8)) return;
        if ((abl & 64) && h >= 2) return;  // 64: no B DMA
        const int k2 = kt + d;
        if (abl & 32) {  // 32: A rows always from window 0, tile 0 (cache-hot source)
            stage(h, (g + d) & 1, 0, 0, tn_c * 256, k2 < nk ? k2 : k2 - nk);
        } else if (k2 < nk) {
            stage(h, (g + d) & 1, n_c, tl_c * 256, tn_c * 256, k2);
        } else {
            stage(h, (g + d) & 1, n_n, tl_n * 256, tn_n * 256, k2 - nk);
        }
    };

    // ---- bias of this wave's columns in read-back layout (lane -> 4 consecutive columns of a 32-column block) ----
    f32x4 bias4[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    int bias_tn = -1;
    auto load_bias = [&](int tn) {
        if (EPI != EPI_FWD || tn == bias_tn) return;
        bias_tn = tn;
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
            bias4[jn] = *reinterpret_cast<const f32x4*>(p.bias + (tn * 256 + wn * 64 + jn * 32) + (unsigned)((lane & 7) * 4));
        // make the loads complete HERE: the compiler's own wait for them must not land inside the pipelined loop
        asm volatile("" : "+v"(bias4[0]), "+v"(bias4[1]));
    };

    f32x16 acc[4][2];
    auto zero = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    };

    // ---- epilogue of the current tile: wave-private, through 4 KB of fp32 scratch per (row block, column block) ----
    auto epilogue = [&]() {
        char* scr = lds + 2 * BUF + w * SCR;
        // The epilogue sits inside the K-tile loop: without this the compiler hoists every lane-dependent offset below out
        // of the loop and keeps ~40 of them live across the MFMA pipeline (spills).  Recompute them per tile instead.
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int hi = ln >> 5, m = ln & 31;
        const int rb = ln >> 3, c = ln & 7;  // read-back: row within a group of 8, 16-byte chunk
        const int t_base = tl_c * 256 + wm * 128;
        const int col_base = tn_c * 256 + wn * 64;
        // wave-uniform base + 32-bit lane offset: the stores use the scalar-base form (no 64-bit address VGPRs)
        bf16* const out_u = p.out + ((int64_t)n_c * p.L + t_base) * p.N + col_base;
        if (EPI == EPI_DGRAD) {
            // no bias / ReLU / statistics: convert in the accumulator layout, stage 32 rows x 64 columns in the STORAGE type
            // (8-byte writes, 16-byte read-back) and leave as whole 128-byte row segments -- half the LDS traffic and half the
            // store instructions of the fp32 form below
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) {
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        bf16 o[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (bf16)acc[i][jn][4 * g4 + e];
                        *reinterpret_cast<u32x2*>(scr + m * 128 + (((jn * 4 + g4) ^ (m & 7)) * 16) + hi * 8) = *reinterpret_cast<const u32x2*>(o);
                    }
                }
                u32x4 vv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) vv[k] = *reinterpret_cast<const u32x4*>(scr + (k * 8 + rb) * 128 + ((c ^ rb) * 16));
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = k * 8 + rb;
                    if (t_base + i * 32 + row < p.L) *reinterpret_cast<u32x4*>(out_u + (unsigned)((i * 32 + row) * p.N + c * 8)) = vv[k];
                }
            }
            return;
        }
        float s4[2][4], q4[2][4];
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int e = 0; e < 4; ++e) s4[jn][e] = q4[jn][e] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 v = {acc[i][jn][4 * g4], acc[i][jn][4 * g4 + 1], acc[i][jn][4 * g4 + 2], acc[i][jn][4 * g4 + 3]};
                    *reinterpret_cast<f32x4*>(scr + m * 128 + (((2 * g4 + hi) ^ (m & 7)) * 16)) = v;
                }
                f32x4 vv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    vv[k] = *reinterpret_cast<const f32x4*>(scr + (k * 8 + rb) * 128 + ((c ^ rb) * 16));
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = k * 8 + rb;
                    const f32x4 v = vv[k];
                    const bool ok = t_base + i * 32 + row < p.L;
                    bf16 o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = v[e];
                        if (EPI == EPI_FWD) {
                            x += bias4[jn][e];
                            x = x > 0.f ? x : 0.f;
                        }
                        o[e] = (bf16)x;
                        if (EPI == EPI_FWD) {
                            const float xr = ok ? (float)o[e] : 0.f;
                            s4[jn][e] += xr;
                            q4[jn][e] += xr * xr;
                        }
                    }
                    if (ok) *reinterpret_cast<u32x2*>(out_u + (unsigned)((i * 32 + row) * p.N + jn * 32 + c * 4)) = *reinterpret_cast<const u32x2*>(o);
                }
            }
        }
        if (EPI == EPI_FWD && p.stat_sum != nullptr) {
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int o = 8; o < 64; o <<= 1) {
                        s4[jn][e] += __shfl_xor(s4[jn][e], o, 64);
                        q4[jn][e] += __shfl_xor(q4[jn][e], o, 64);
                    }
                }
            const int tiles128 = (p.L + 127) / 128;  // rows of the statistics buffers per window (vm_conv_stat_rows)
            const int srow = tl_c * 2 + wm;
            if (ln < 8 && srow < tiles128) {
                const int64_t row = (int64_t)n_c * tiles128 + srow;
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) {
                    const f32x4 sv = {s4[jn][0], s4[jn][1], s4[jn][2], s4[jn][3]};
                    const f32x4 qv = {q4[jn][0], q4[jn][1], q4[jn][2], q4[jn][3]};
                    *reinterpret_cast<f32x4*>(p.stat_sum + (row * p.N + col_base + jn * 32) + (unsigned)(c * 4)) = sv;
                    *reinterpret_cast<f32x4*>(p.stat_sq + (row * p.N + col_base + jn * 32) + (unsigned)(c * 4)) = qv;
                }
            }
        }
    };

    // Optional start skew: with one lock-step workgroup per CU every CU reaches its epilogue at the same time and the chip
    // writes a 33 MB burst per tile round; delaying the workgroups of an XCD by 0..3 units spreads the bursts.
    if (p.skew > 0) {
        const int reps = (((int)blockIdx.x >> 3) & 3) * p.skew;
        for (int i = 0; i < reps; ++i) __builtin_amdgcn_s_sleep(127);
    }

    // ---- prologue: K tile 0 and what the last phases of stream position -1 would have issued (nk >= 3) ----
    load_bias(tn_c);
    stage(H_A0, 0, n_c, tl_c * 256, tn_c * 256, 0);
    stage(H_B1, 0, n_c, tl_c * 256, tn_c * 256, 0);
    stage(H_A1, 0, n_c, tl_c * 256, tn_c * 256, 0);
    stage(H_B0, 0, n_c, tl_c * 256, tn_c * 256, 0);
    if (PH == 4) {
        stage(H_A0, 1, n_c, tl_c * 256, tn_c * 256, 1);
        stage(H_B1, 1, n_c, tl_c * 256, tn_c * 256, 1);
        stage(H_A1, 1, n_c, tl_c * 256, tn_c * 256, 1);
        wait_vmcnt<6>();
    } else {
        stage(H_A0, 1, n_c, tl_c * 256, tn_c * 256, 1);
        stage(H_B0, 1, n_c, tl_c * 256, tn_c * 256, 1);
        wait_vmcnt<4>();
    }
    slot_end();
    zero();
    if (wm == 1) slot_end();  // rows 128-255 run one slot behind

    // ---- the stream ----
    FragA fa;
    FragB fb, fb1;
    if (abl & 16) {  // defined (garbage-free) operands for the no-read timing experiment
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            fa[0][s] = fa[1][s] = fb[s] = fb1[s] = bf16x8{};
        }
    }
    for (int g = 0; g < G; ++g) {
        const int buf = g & 1;
        if (PH == 4) {
            // phase 0
            read_a(fa, buf, 0);
            read_b(fb, buf, 0);
            stage_ahead(H_B0, g, 1);
            read_done();
            slot_end();
            mma(fa, fb, acc[0][0], acc[1][0]);
            slot_end();
            // phase 1
            read_b(fb, buf, 1);
            stage_ahead(H_A0, g, 2);
            read_done();
            slot_end();
            mma(fa, fb, acc[0][1], acc[1][1]);
            slot_end();
            // phase 2
            read_a(fa, buf, 1);
            stage_ahead(H_B1, g, 2);
            read_done();
            slot_end();
            mma(fa, fb, acc[2][1], acc[3][1]);
            slot_end();
            // phase 3
            read_b(fb, buf, 0);
            stage_ahead(H_A1, g, 2);
            if (g + 1 < G) {
                if (abl & 64) {
                    wait_vmcnt<4>();
                } else if (g + 2 < G) {
                    wait_vmcnt<6>();
                } else {
                    wait_vmcnt<0>();
                }
            }
            read_done();
            slot_end();
            mma(fa, fb, acc[2][0], acc[3][0]);
            slot_end();
        } else {
            // Two phases of 16 MFMAs (half the barriers, no second read of B_c0):
            //   phase 0: read A_lo, B_c0, B_c1 | DMA B1(g+1), A1(g+1) | MFMA A_lo x (B_c0, B_c1)
            //   phase 1: read A_hi             | DMA A0(g+2), B0(g+2) | vmcnt(4): K tile g+1 has landed | MFMA A_hi x (B_c0, B_c1)
            // WAR: A0, B0, B1 are last read in phase 0 (re-staged from phase 1 on), A1 in phase 1 (re-staged in phase 0 of
            // g+1).  RAW: group y waits at the end of READ(g,1) = slot 4g+2+y, the first reads of g+1 are in slot 4g+4.
            read_a(fa, buf, 0);
            read_b(fb, buf, 0);
            read_b(fb1, buf, 1);
            stage_ahead(H_B1, g, 1);
            stage_ahead(H_A1, g, 1);
            read_done();
            slot_end();
            mma(fa, fb, acc[0][0], acc[1][0]);
            mma(fa, fb1, acc[0][1], acc[1][1]);
            slot_end();
            read_a(fa, buf, 1);
            stage_ahead(H_A0, g, 2);
            stage_ahead(H_B0, g, 2);
            if (g + 1 < G) {
                if (g + 2 < G && !(abl & 8)) {
                    wait_vmcnt<4>();
                } else {
                    wait_vmcnt<0>();
                }
            }
            read_done();
            slot_end();
            mma(fa, fb, acc[2][0], acc[3][0]);
            mma(fa, fb1, acc[2][1], acc[3][1]);
            slot_end();
        }
        if (++kt == nk) {
            if (!(abl & 2)) epilogue();
            zero();
            kt = 0;
            ++it;
            n_c = n_n;
            tl_c = tl_n;
            tn_c = tn_n;
            if (it < my_tiles) load_bias(tn_c);
            if (it + 1 < my_tiles) decode(it + 1, n_n, tl_n, tn_n);
        }
    }
    if (wm == 0) slot_end();  // balance the barrier count of the two groups
}

// ------------------------------------------------------------------------------------------------
// NT conv GEMM with ONE wave per SIMD (bf16, opt-in: vm_set_tuning("nt_w4", 1 | 2); the round-2 direction of DESIGN.md 8.1,
// developed as tools/probe/conv_w4_probe.hip).  256-thread workgroup, every wave owns a 128 x 128 output tile in 256
// accumulator registers (0.5 KB of fragment reads per MFMA against 0.75 KB in conv_nt8_kernel); tile = 254 output positions x
// 256 channels; the K walk is (channel chunk, tap) with an INPUT-RESIDENT A: the 256 padded rows of a 64-channel chunk are
// staged once and read at row offsets 0 / 1 / 2 for the three taps (a third of the A pieces to issue).
// LDS: 2 A blocks + 2 B stages of 32 KB = 128 KB; the epilogue reuses it as one 32 KB transpose region per wave.
// Every MFMA is followed by one filler in its issue shadow (fragment read of the next k-step, or a DMA piece);
// per K tile g = 3*chunk + tap there is one counted DMA wait + one s_barrier in the middle of its last k-step:
//   tap 0: k-step 0 issues B(g+1) [8 pieces], k-step 1 the first half of A(chunk+1) [4] -> wait vmcnt(4): B(g+1) landed
//   tap 1: likewise with the second half of A(chunk+1)                                 -> wait vmcnt(4)
//   tap 2: k-step 0 issues B(g+1) [8]                                                  -> wait vmcnt(0): B(g+1) and A(chunk+1)
// (loads complete in order, and inside a K tile every B piece precedes the A pieces).  WAR: a B stage / A block is re-staged
// only after the barrier that follows its last fragment reads.  Requires Ktot == 3*a_c, a_c % 64 == 0, N % 256 == 0.
// ------------------------------------------------------------------------------------------------
namespace w4 {
constexpr int TROWS = 254;
constexpr int ROWB = 128;
constexpr int OPB = 256 * ROWB;
constexpr int B0 = 2 * OPB;
constexpr int LDS_BYTES = 4 * OPB;
struct Frag {
    bf16x8 a[4], b[4];
};
}  // namespace w4

template <int EPI>
__global__ __attribute__((amdgpu_flat_work_group_size(256, 256), amdgpu_waves_per_eu(1, 1))) void conv_w4_kernel(NtArgs<bf16> p) {
    using namespace w4;
    __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int tn = __builtin_amdgcn_readfirstlane((int)blockIdx.x % p.tilesN);
    const int grp = (int)blockIdx.x / p.tilesN;
    const int tl = __builtin_amdgcn_readfirstlane(grp % p.tilesL), n = __builtin_amdgcn_readfirstlane(grp / p.tilesL);
    const int t0 = tl * TROWS;
    const int chunks = p.a_c / 64, nk = chunks * 3;
    const int a_pitch = p.a_c * 2, b_pitch = p.Ktot * 2;

    // ---- DMA geometry: a piece = 8 rows x 128 B, lane-linear in LDS; wave w stages rows [64w, 64w+64) of a block; LDS row R
    // keeps 16-byte chunk c at position c ^ ((R >> 1) & 7), so the source chunk of a lane is permuted accordingly ----
    const int prow = lane >> 3;
    const char* a_win = reinterpret_cast<const char*>(p.a) + (int64_t)n * p.a_win_stride * 2;
    const char* b_base = reinterpret_cast<const char*>(p.bt) + (int64_t)(tn * 256 + w * 64) * b_pitch;
    unsigned a_off[8], b_off[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int key = (4 * q + (prow >> 1)) & 7;
        int pr = t0 + w * 64 + q * 8 + prow;  // padded row of the window; the tail tile reads its last halo row again
        pr = pr < p.L + 2 ? pr : p.L + 1;
        a_off[q] = (unsigned)(pr * a_pitch + (((lane & 7) ^ key) * 16));
        b_off[q] = (unsigned)((q * 8 + prow) * b_pitch + (((lane & 7) ^ key) * 16));
    }
    auto stage_a = [&](int blk, int chunk, int q) { glds16(a_win + (int64_t)chunk * ROWB + a_off[q], lds + blk * OPB + (w * 64 + q * 8) * ROWB); };
    auto stage_b = [&](int stg, int g, int q) {  // K tile g = 3*chunk + tap -> weight columns tap*a_c + 64*chunk
        const int chunk = g / 3, tap = g - 3 * chunk;
        glds16(b_base + (int64_t)(tap * p.a_c + chunk * 64) * 2 + b_off[q], lds + B0 + stg * OPB + (w * 64 + q * 8) * ROWB);
    };

    // ---- fragment geometry: lane (r, kh) reads 16 B of row r (+ tap for A), k-chunk 2s + kh ----
    const int r = lane & 31, kh = lane >> 5;
    int foff_b[4], foff_a[3][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        foff_b[s] = r * ROWB + (((2 * s + kh) ^ ((r >> 1) & 7)) * 16);
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) foff_a[tap][s] = (r + tap) * ROWB + (((2 * s + kh) ^ (((r + tap) >> 1) & 7)) * 16);
    }
    const int a_rows = wm * 128 * ROWB, b_rows = wn * 128 * ROWB;
    auto one_read = [&](Frag& f, int blk, int stg, int tap, int s, int t) {  // t in 0..7: a[0..3], b[0..3]
        if (t < 4) {
            f.a[t] = *reinterpret_cast<const bf16x8*>(lds + blk * OPB + a_rows + t * 32 * ROWB + foff_a[tap][s]);
        } else {
            f.b[t - 4] = *reinterpret_cast<const bf16x8*>(lds + B0 + stg * OPB + b_rows + (t - 4) * 32 * ROWB + foff_b[s]);
        }
    };

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto kstep = [&](const Frag& cur, Frag& nxt, int rblk, int rstg, int rtap, int rs, int rslot0, int bstg, int bg, int ablk, int achunk, int aq0,
                     int wait_n) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int i = t >> 2, j = t & 3;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.b[j], cur.a[i], acc[i][j], 0, 0, 0);
            if (wait_n >= 0 && t == 7) {
                if (wait_n == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            if (rslot0 == 0) {
                if ((t & 1) == 0) one_read(nxt, rblk, rstg, rtap, rs, t >> 1);
            } else if (t >= 8) {
                one_read(nxt, rblk, rstg, rtap, rs, t - 8);
            }
            if (bg >= 0 && (t & 1)) stage_b(bstg, bg, t >> 1);                        // k-step 0: the 8 B pieces, odd slots
            if (achunk >= 0 && (t & 3) == 1) stage_a(ablk, achunk, aq0 + (t >> 2));  // k-step 1: 4 A pieces, slots 1, 5, 9, 13
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- prologue: A(0), B(0) ----
#pragma unroll
    for (int q = 0; q < 8; ++q) stage_a(0, 0, q);
#pragma unroll
    for (int q = 0; q < 8; ++q) stage_b(0, 0, q);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    Frag f0, f1;
#pragma unroll
    for (int t = 0; t < 8; ++t) one_read(f0, 0, 0, 0, 0, t);

    // ---- K-tile stream; past the end the DMAs re-stage K tile 0 / chunk 0 into memory nobody reads ----
    for (int c = 0; c < chunks; ++c) {
        const int ablk = c & 1;
        const int cn = c + 1 < chunks ? c + 1 : 0;
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
            const int g = 3 * c + tap, sb = g & 1;
            const int gn = g + 1 < nk ? g + 1 : 0;
            const int ntap = tap == 2 ? 0 : tap + 1, nblk = tap == 2 ? (ablk ^ 1) : ablk;  // where the next K tile reads
            kstep(f0, f1, ablk, sb, tap, 1, 0, sb ^ 1, gn, 0, -1, 0, -1);
            kstep(f1, f0, ablk, sb, tap, 2, 0, 0, -1, ablk ^ 1, tap == 2 ? -1 : cn, tap == 0 ? 0 : 4, -1);
            kstep(f0, f1, ablk, sb, tap, 3, 0, 0, -1, 0, -1, 0, -1);
            kstep(f1, f0, nblk, sb ^ 1, ntap, 0, 8, 0, -1, 0, -1, 0, tap == 2 ? 0 : 4);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave is done with the operand memory: it becomes the epilogue scratch

    // ---- epilogue: (bias + ReLU,) bf16, through LDS (the wave's 128 x 128 tile, rows of 256 B, 16-byte chunk c of row R at
    // c ^ (R & 15)) to whole-row 16-byte stores; forward: statistics of the stored (rounded) values ----
    char* scr = lds + w * 32768;
    const int col0 = tn * 256 + wn * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cc = j * 32 + 8 * g + 4 * kh;  // first of this lane's 4 consecutive channels in the wave tile
                bf16 o[4];
                if (EPI == EPI_FWD) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + col0 + cc);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = acc[i][j][4 * g + e] + bv[e];
                        o[e] = (bf16)(x > 0.f ? x : 0.f);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (bf16)acc[i][j][4 * g + e];
                }
                const int row = i * 32 + r, cb = cc * 2;
                *reinterpret_cast<u32x2*>(scr + row * 256 + (((cb >> 4) ^ (row & 15)) << 4) + (cb & 15)) = *reinterpret_cast<const u32x2*>(o);
            }
        }
    }
    float s8[8], q8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s8[e] = q8[e] = 0.f;
    bf16* zbase = p.out + ((int64_t)n * p.L + t0 + wm * 128) * p.N + col0;
    const int c16 = lane & 15;
    const bool stats = EPI == EPI_FWD && p.stat_sum != nullptr;
#pragma unroll 4
    for (int it = 0; it < 32; ++it) {
        const int row = it * 4 + (lane >> 4);
        const int trow = wm * 128 + row;  // row inside the 256-row MFMA tile
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(scr + row * 256 + ((c16 ^ (row & 15)) << 4));
        if (trow < TROWS && t0 + trow < p.L) {
            *reinterpret_cast<bf16x8*>(zbase + (int64_t)row * p.N + c16 * 8) = v;
            if (stats) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float x = (float)v[e];
                    s8[e] += x;
                    q8[e] = fmaf(x, x, q8[e]);
                }
            }
        }
    }
    if (stats) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s8[e] += __shfl_xor(s8[e], 16, 64);
            s8[e] += __shfl_xor(s8[e], 32, 64);
            q8[e] += __shfl_xor(q8[e], 16, 64);
            q8[e] += __shfl_xor(q8[e], 32, 64);
        }
        if (lane < 16) {
            // statistics rows per window = (L + 127) / 128 (vm_conv_stat_rows); the launch guarantees 2 * tilesL of them
            const int64_t srow = (int64_t)n * ((p.L + 127) / 128) + tl * 2 + wm;
            float* ps = p.stat_sum + srow * p.N + col0 + c16 * 8;
            float* pq = p.stat_sq + srow * p.N + col0 + c16 * 8;
            *reinterpret_cast<f32x4*>(ps) = f32x4{s8[0], s8[1], s8[2], s8[3]};
            *reinterpret_cast<f32x4*>(ps + 4) = f32x4{s8[4], s8[5], s8[6], s8[7]};
            *reinterpret_cast<f32x4*>(pq) = f32x4{q8[0], q8[1], q8[2], q8[3]};
            *reinterpret_cast<f32x4*>(pq + 4) = f32x4{q8[4], q8[5], q8[6], q8[7]};
        }
    }
}

// ------------------------------------------------------------------------------------------------
// NT GEMM, 256 (positions) x 128 (channels) output tile, 4 waves (2 x 2, 128 x 64 each), TWO workgroups per CU (bf16).
//
// Why this shape (round 2).  At cfg-A an output tile sees only K = 384..1536, i.e. a 256 x 256 tile is 6..24 K tiles of 64 and
// then an epilogue that has to push 128 KB through a store path that issues ~7..10 B/clk/CU: in the one-workgroup-per-CU
// kernels (conv_nt8_kernel, conv_w4_kernel) that epilogue and the cold start of the next tile are a quarter of the tile time
// with the matrix pipes idle (profiles/r01_gemm_w4_structure_probe.txt: cold start 6.5k + epilogue 10.5k of 69k cycles), and
// making the K stream persistent inside a workgroup did not help (tools/probe/conv_w4p_probe: 364 us against 280 us).  The
// 128^2 kernels do overlap epilogues with other workgroups' main loops (2-3 per CU) but their 64 x 64 wave tiles need 1 KB of
// fragment reads per MFMA and 2 x 128 rows of DMA per 4 MFMA-steps: LDS-bound outright.  This kernel keeps the 128 x 64 wave
// tile of conv_nt8_kernel (0.75 KB of fragment reads per MFMA, 128 accumulator registers -> 256 registers per wave -> two
// waves per SIMD) but gives each group of four waves its OWN output tile and its own 72 KB of LDS, as an independent workgroup:
// two of them share a CU, drift apart, and one's epilogue / cold start runs under the other's MFMAs.  Per MFMA the DMA bytes
// are (256 + 128) / (256 * 128) against (128 + 128) / (128 * 128) of the 128^2 kernels: -25 %.
//
// K slices of 64 bytes (32 bf16) in a ring of three 24 KB stages (A 256 rows + B 128 rows, unpadded, 16-byte chunk c of row R at
// c ^ ((R >> 2) & 3): swizzle applied to the per-lane SOURCE address of the LDS-DMA and again by the fragment reads); K is
// walked (channel chunk, tap) so that consecutive slices re-read the same input cache lines one row later.  One counted
// s_waitcnt vmcnt(6) (= the 6 pieces of the slice that may still fly) + one raw s_barrier per slice.
//   RAW: a wave waits for its own pieces of slice kt, then the barrier: after it every wave's pieces of slice kt have landed.
//   WAR: slice kt+2 is staged into the stage read in iteration kt-1, after barrier kt, which every wave reaches only with all
//        its fragment reads of iteration kt-1 issued and consumed by its MFMAs.
// Epilogue: (bias + ReLU,) bf16 in registers, tile through LDS ([256][272 B]), whole-row 16-byte stores, forward statistics of
// the stored (rounded) values as two partial rows per tile (one per 128 positions: the layout of vm_conv_stat_rows).
// Requires a_c % 32 == 0, Ktot == 3 * a_c, N % 128 == 0.
// ------------------------------------------------------------------------------------------------
namespace n2 {
constexpr int TM = 256, TN = 128, KB = 64, RING = 3;
constexpr int A_BYTES = TM * KB, B_BYTES = TN * KB, STAGE = A_BYTES + B_BYTES;
constexpr int TP = TN * 2 + 16;
constexpr int LDS_BYTES = (RING * STAGE > TM * TP) ? RING * STAGE : TM * TP;  // operand rings / the epilogue tile
static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
}  // namespace n2

// accumulators start at the bias of their channel (forward) or 0 (dgrad): register 4g + e of block j <-> channel c0 + 32j + 8g + e
template <int EPI>
__device__ inline void n2_load_bias(const NtArgs<bf16>& p, f32x4 (&b4)[2][4], int c0) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            b4[j][g] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (EPI != EPI_DGRAD) b4[j][g] = *reinterpret_cast<const f32x4*>(p.bias + c0 + 32 * j + 8 * g);
        }
}
__device__ inline void n2_fill_acc(f32x16 (&acc)[4][2], const f32x4 (&b4)[2][4]) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] = b4[j][g][e];
}
template <int EPI>
__device__ inline void n2_init_acc(const NtArgs<bf16>& p, f32x16 (&acc)[4][2], int c0) {
    f32x4 b4[2][4];
    n2_load_bias<EPI>(p, b4, c0);
    n2_fill_acc(acc, b4);
}

// Shared epilogue of the 256 x 128 kernels below.  ``trows``: valid MFMA-tile rows (256, or 254 for the input-resident kernel).
//
// Forward statistics ON THE MATRIX PIPE.  The BatchNorm partial sums (sum z, sum z^2 per channel over the tile's positions) used
// to be 3 VALU instructions per output element in the read-back loop -- 37..47 us of a 205..237 us launch at cfg-A, un-hidden
// (ablation: forward 0.662 -> 0.540 ms per step without them).  They are column sums of the bf16 tile Z that sits in LDS anyway:
//     sum_r Z[r][c]        = (1^T Z)[c]                    sum_r Z[r][c]^2 = diag(Z^T Z)[c]
// so wave w takes channels [32 w, 32 w + 32), reads the K-major fragment X (lane <-> channel, 8 consecutive positions per lane)
// of 16 positions with two transposing LDS reads (ds_read_b64_tr_b16, the wgrad kernels' read) and issues
//     D2 += X X^T  (32 x 32 block of Z^T Z: its diagonal are the squares)     D1 += 1 X^T  (every row = the column sums)
// -- the SAME registers serve as both operands of the first MFMA.  32 transposing reads + 32 MFMAs per wave replace 384 VALU
// instructions per thread; products of bf16 values are exact in fp32, so the sums are those of the stored (rounded) values as
// before, in a different (fixed) order.  Rows that are not positions of the window (254 / 255 of the input-resident tile, the
// tail of the last tile) are written to LDS as zeros so that they drop out.
template <int EPI>
__device__ inline void n2_epilogue(const NtArgs<bf16>& p, char* lds, const f32x16 (&acc)[4][2], int64_t n, int tl, int t0, int n0, int trows,
                                   int tid, int lane, int w, int wm, int wn) {
    using namespace n2;
    const int r = lane & 31, kh = lane >> 5;
    const bool stats = EPI == EPI_FWD && p.stat_sum != nullptr && !(p.ablate & 64);
    const bool red = EPI == EPI_DGRAD && p.red_a != nullptr;
    const int valid = (p.L - t0) < trows ? (p.L - t0) : trows;  // MFMA-tile rows that are positions of the window
    __syncthreads();  // every wave is done with the operand stages: they become the epilogue tile

    // ---- registers -> bf16 tile in LDS.  Forward: the bias is already in the accumulators (they were initialised with it) and
    // ReLU is applied to the PACKED bf16 pairs as a signed 16-bit max with 0 (a negative bf16 is a negative int16, -0.0 included;
    // rounding is monotone, so relu(round(x)) == round(relu(x))): 1 VALU instruction per element instead of 2.5 ----
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = wm * 128 + i * 32 + r;
        const bool partial = wm * 128 + i * 32 + 32 > valid;  // wave-uniform: this 32-row block has rows outside the window
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = wn * 64 + j * 32 + 8 * g + 4 * kh;  // first of this lane's 4 consecutive channels
                bf16 o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16)acc[i][j][4 * g + e];
                u32x2 pk = *reinterpret_cast<const u32x2*>(o);
                if (EPI != EPI_DGRAD) {
                    uint32_t lo = pk[0], hi = pk[1];
                    asm("v_pk_max_i16 %0, %1, 0" : "=v"(lo) : "v"(lo));
                    asm("v_pk_max_i16 %0, %1, 0" : "=v"(hi) : "v"(hi));
                    pk[0] = lo;
                    pk[1] = hi;
                }
                if ((EPI == EPI_FWD || red) && partial && m >= valid) pk[0] = pk[1] = 0u;
                *reinterpret_cast<u32x2*>(lds + m * TP + nl * 2) = pk;
            }
        }
    }
    __syncthreads();
    const int c8 = tid & 15, rg = tid >> 4;
    if (EPI == EPI_FWD_POOL) {
        // ---- inference: y = z * scale + shift per channel, max over the position pairs (2q, 2q + 1) -- the arithmetic of
        // bn_drop_pool_fwd_kernel on the same storage-rounded z, so the pooled tensor is bit-identical to the two-kernel path; z itself
        // is never written.  Tiles start at even positions and L is even (checked by the launch): a pair never straddles tiles ----
        float sc[8], sh[8];
        {
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(p.aff_scale + n0 + c8 * 8), s1 = *reinterpret_cast<const f32x4*>(p.aff_scale + n0 + c8 * 8 + 4);
            const f32x4 h0 = *reinterpret_cast<const f32x4*>(p.aff_shift + n0 + c8 * 8), h1 = *reinterpret_cast<const f32x4*>(p.aff_shift + n0 + c8 * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sc[e] = s0[e];
                sc[4 + e] = s1[e];
                sh[e] = h0[e];
                sh[4 + e] = h1[e];
            }
        }
        const int vq = valid >> 1;
        bf16* pbase = p.out + (n * (int64_t)(p.L / 2 + 2) + 1 + (t0 >> 1)) * (int64_t)p.N + n0 + c8 * 8;
        bf16x8 r0[8], r1[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int q = rg + 16 * jj;
            r0[jj] = *reinterpret_cast<const bf16x8*>(lds + (2 * q) * TP + c8 * 16);
            r1[jj] = *reinterpret_cast<const bf16x8*>(lds + (2 * q + 1) * TP + c8 * 16);
        }
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int q = rg + 16 * jj;
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float y0 = fmaf((float)r0[jj][e], sc[e], sh[e]), y1 = fmaf((float)r1[jj][e], sc[e], sh[e]);
                o[e] = (bf16)(y1 > y0 ? y1 : y0);
            }
            if (q < vq) *reinterpret_cast<bf16x8*>(pbase + (int64_t)q * p.N) = o;
        }
        return;
    }
    // ---- read-back: 8 rows per thread and half, ALL tile reads first, then the 8 whole-row stores back to back; interior tiles
    // take a predicate-free path ----
    bf16* obase = p.out + (n * p.L + t0) * (int64_t)p.N + n0 + c8 * 8;
    const bool interior = t0 + trows <= p.L;  // every valid MFMA row of the tile is a position of the window
    const bool no_store = (p.ablate & 1) != 0;
    auto half = [&](int h, auto interior_c) {
        constexpr bool INTERIOR = decltype(interior_c)::value;
        bf16x8 v[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) v[jj] = *reinterpret_cast<const bf16x8*>(lds + (h * 128 + rg + 16 * jj) * TP + c8 * 16);
        if (!no_store) {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int row = h * 128 + rg + 16 * jj;
                // rows >= trows exist only in the last 16 rows of the tile (trows >= 240)
                const bool ok = INTERIOR ? (h == 0 || jj < 7 || row < trows) : row < valid;
                if (ok) *reinterpret_cast<bf16x8*>(obase + (int64_t)row * p.N) = v[jj];
            }
        }
    };
    if (interior) {
        half(0, std::true_type{});
        half(1, std::true_type{});
    } else {
        half(0, std::false_type{});
        half(1, std::false_type{});
    }
    if (EPI == EPI_FWD && p.pool_e != nullptr) {
        // ---- pool-window extreme of z for the BatchNorm / pool pass that follows the statistics (it then reads a pooled-size tensor
        // instead of z: max_j fma(z_j, s, h) == fma(ext_j z, s, h), the extreme being the maximum for s >= 0 and the minimum for
        // s < 0; sign(s) = sign(gamma) is known before the statistics are).  z >= 0 after ReLU, so the packed 16-bit integer max / min
        // order the bf16 values. ----
        u32x4 neg;  // 0xFFFF in the 16-bit lanes of channels with gamma < 0
        {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.aff_scale + n0 + c8 * 8), g1 = *reinterpret_cast<const f32x4*>(p.aff_scale + n0 + c8 * 8 + 4);
            neg[0] = (g0[0] < 0.f ? 0xFFFFu : 0u) | (g0[1] < 0.f ? 0xFFFF0000u : 0u);
            neg[1] = (g0[2] < 0.f ? 0xFFFFu : 0u) | (g0[3] < 0.f ? 0xFFFF0000u : 0u);
            neg[2] = (g1[0] < 0.f ? 0xFFFFu : 0u) | (g1[1] < 0.f ? 0xFFFF0000u : 0u);
            neg[3] = (g1[2] < 0.f ? 0xFFFFu : 0u) | (g1[3] < 0.f ? 0xFFFF0000u : 0u);
        }
        const int vq = valid >> 1;
        bf16* ebase = p.pool_e + (n * (int64_t)(p.L / 2) + (t0 >> 1)) * (int64_t)p.N + n0 + c8 * 8;
        u32x4 r0[8], r1[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int q = rg + 16 * jj;
            r0[jj] = *reinterpret_cast<const u32x4*>(lds + (2 * q) * TP + c8 * 16);
            r1[jj] = *reinterpret_cast<const u32x4*>(lds + (2 * q + 1) * TP + c8 * 16);
        }
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int q = rg + 16 * jj;
            u32x4 o;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const uint32_t a = r0[jj][d], b = r1[jj][d];
                uint32_t mx, mn;
                asm("v_pk_max_i16 %0, %1, %2" : "=v"(mx) : "v"(a), "v"(b));
                asm("v_pk_min_i16 %0, %1, %2" : "=v"(mn) : "v"(a), "v"(b));
                const uint32_t m = neg[d];
                o[d] = (mx & ~m) | (mn & m);
            }
            if (q < vq) *reinterpret_cast<u32x4*>(ebase + (int64_t)q * p.N) = o;
        }
    }
    if (stats) {
#if defined(__HIP_DEVICE_COMPILE__)
        const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)lds;
#else
        const uint32_t lds0 = 0;
#endif
        // fragment geometry of the transposing read: lane (kh, lg, li) supplies the 8-byte word of row kh * 8 + (li >> 2), channels
        // 16 lg + 4 (li & 3) .. + 3 and receives channel 16 lg + li, rows kh * 8 .. + 3 (second read: + 4 rows)
        const int li = lane & 15, lg = (lane >> 4) & 1;
        const uint32_t xoff = lds0 + (kh * 8 + (li >> 2)) * TP + (32 * w + 16 * lg + 4 * (li & 3)) * 2;
        const u32x4 ones4 = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
        const bf16x8 ones = __builtin_bit_cast(bf16x8, ones4);
        const int srows = (p.L + 127) / 128;  // vm_conv_stat_rows
        const int cn = lane & 31;             // the channel (of this wave's 32) whose sums this lane extracts
        const int rsel = (cn & 3) + 4 * (cn >> 3);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x16 d1, d2;
#pragma unroll
            for (int e = 0; e < 16; ++e) d1[e] = d2[e] = 0.f;
            u32x2 lo[8], hi[8];   // all 16 transposing reads of the half first, then the 16 MFMAs (one latency, not eight)
#pragma unroll
            for (int rs = 0; rs < 8; ++rs) {
                const uint32_t a = xoff + (h * 128 + rs * 16) * TP;
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo[rs]) : "v"(a));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1088" : "=v"(hi[rs]) : "v"(a));  // + 4 rows of 272 bytes
            }
#pragma unroll
            for (int rs = 0; rs < 8; ++rs) {
                if (rs == 0) {
                    asm volatile("s_waitcnt lgkmcnt(14)" : "+v"(lo[0]), "+v"(hi[0]));
                } else if (rs == 1) {
                    asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(lo[1]), "+v"(hi[1]));
                } else if (rs == 2) {
                    asm volatile("s_waitcnt lgkmcnt(10)" : "+v"(lo[2]), "+v"(hi[2]));
                } else if (rs == 3) {
                    asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(lo[3]), "+v"(hi[3]));
                } else if (rs == 4) {
                    asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(lo[4]), "+v"(hi[4]));
                } else if (rs == 5) {
                    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(lo[5]), "+v"(hi[5]));
                } else if (rs == 6) {
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(lo[6]), "+v"(hi[6]));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[7]), "+v"(hi[7]));
                }
                const u32x4 xv = {lo[rs][0], lo[rs][1], hi[rs][0], hi[rs][1]};
                const bf16x8 x = __builtin_bit_cast(bf16x8, xv);
                d2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, d2, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, x, d1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // D1[m][n] = sum over the 16 x 8 rows of Z[row][32 w + n] for every m: register 0 of lane n.  D2[m][n] with
            // m = (r & 3) + 8 (r >> 2) + 4 kh: the diagonal element of channel n is register rsel(n) of the lane with kh == (n >> 2) & 1.
            float dq = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) dq = e == rsel ? d2[e] : dq;
            if (2 * tl + h < srows) {
                const int64_t srow = n * srows + 2 * tl + h;
                if (kh == 0) p.stat_sum[srow * p.N + n0 + 32 * w + cn] = d1[0];
                if (kh == ((cn >> 2) & 1)) p.stat_sq[srow * p.N + n0 + 32 * w + cn] = dq;
            }
        }
    }
    if (red) {
        // ---- BatchNorm-backward partial sums of the layer below, from the tile that is in LDS anyway (vm_conv_dgrad_bnred):
        //   S0[c] = sum_r dp[r][c] = (1^T DP)[c],   S1[c] = sum_r dp[r][c] * A[r][c] = diag(DP^T A)[c]
        // with the statistics machinery above.  The K-major fragments of dp are kept in registers while the A tile (the pooled
        // activation / pooled extreme of the layer below: same rows, same channels) replaces the dp tile in LDS by LDS-DMA. ----
#if defined(__HIP_DEVICE_COMPILE__)
        const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)lds;
#else
        const uint32_t lds0 = 0;
#endif
        const int li = lane & 15, lg = (lane >> 4) & 1;
        const uint32_t xoff = lds0 + (kh * 8 + (li >> 2)) * TP + (32 * w + 16 * lg + 4 * (li & 3)) * 2;
        const u32x4 ones4 = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
        const bf16x8 ones = __builtin_bit_cast(bf16x8, ones4);
        u32x2 dlo[16], dhi[16];
        f32x16 d1[2], d2[2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 16; ++e) d1[h][e] = d2[h][e] = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const uint32_t a = xoff + (q * 16) * TP;
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dlo[q]) : "v"(a));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1088" : "=v"(dhi[q]) : "v"(a));
            if (q == 7) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dlo[0]), "+v"(dhi[0]), "+v"(dlo[1]), "+v"(dhi[1]), "+v"(dlo[2]), "+v"(dhi[2]),
                                     "+v"(dlo[3]), "+v"(dhi[3]), "+v"(dlo[4]), "+v"(dhi[4]), "+v"(dlo[5]), "+v"(dhi[5]), "+v"(dlo[6]), "+v"(dhi[6]),
                                     "+v"(dlo[7]), "+v"(dhi[7]));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dlo[8]), "+v"(dhi[8]), "+v"(dlo[9]), "+v"(dhi[9]), "+v"(dlo[10]), "+v"(dhi[10]), "+v"(dlo[11]),
                     "+v"(dhi[11]), "+v"(dlo[12]), "+v"(dhi[12]), "+v"(dlo[13]), "+v"(dhi[13]), "+v"(dlo[14]), "+v"(dhi[14]), "+v"(dlo[15]),
                     "+v"(dhi[15]));
        __syncthreads();  // every wave has its dp fragments and has issued its output stores: the tile memory is free
        // A tile: 256 rows x 128 channels, 256-byte rows (unpadded: the LDS-DMA destination is lane-linear), 64 pieces of 4 rows
        {
            const int arow = lane >> 4, achunk = lane & 15;
            const bf16* abase = p.red_a + n * p.red_a_win_stride + (int64_t)(t0 + p.red_a_row0) * p.N + n0 + achunk * 8;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int pi = w + 4 * k;
                int R = pi * 4 + arow;
                R = R < valid ? R : valid - 1;  // rows outside the window: any finite value (their dp rows are zero)
                glds16(reinterpret_cast<const char*>(abase + (int64_t)R * p.N), lds + __builtin_amdgcn_readfirstlane(pi * 1024));
            }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const u32x4 xv = {dlo[q][0], dlo[q][1], dhi[q][0], dhi[q][1]};
            d1[q >> 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, __builtin_bit_cast(bf16x8, xv), d1[q >> 3], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // stores and loads retire out of order with each other: no counted wait here
        __syncthreads();  // the A tile has landed for every wave
        const uint32_t aoff = lds0 + (kh * 8 + (li >> 2)) * 256 + (32 * w + 16 * lg + 4 * (li & 3)) * 2;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32x2 alo[8], ahi[8];
#pragma unroll
            for (int rs = 0; rs < 8; ++rs) {
                const uint32_t a = aoff + (h * 128 + rs * 16) * 256;
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(alo[rs]) : "v"(a));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(ahi[rs]) : "v"(a));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(alo[0]), "+v"(ahi[0]), "+v"(alo[1]), "+v"(ahi[1]), "+v"(alo[2]), "+v"(ahi[2]), "+v"(alo[3]),
                         "+v"(ahi[3]), "+v"(alo[4]), "+v"(ahi[4]), "+v"(alo[5]), "+v"(ahi[5]), "+v"(alo[6]), "+v"(ahi[6]), "+v"(alo[7]), "+v"(ahi[7]));
#pragma unroll
            for (int rs = 0; rs < 8; ++rs) {
                const int q = h * 8 + rs;
                const u32x4 dv = {dlo[q][0], dlo[q][1], dhi[q][0], dhi[q][1]};
                const u32x4 av = {alo[rs][0], alo[rs][1], ahi[rs][0], ahi[rs][1]};
                d2[h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, dv), __builtin_bit_cast(bf16x8, av), d2[h], 0, 0, 0);
            }
        }
        const int cn = lane & 31, rsel = (cn & 3) + 4 * (cn >> 3);
        const int rows2 = 2 * p.tilesL;  // partial rows per window (vm_conv_dgrad_bnred_rows)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float dq = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) dq = e == rsel ? d2[h][e] : dq;
            const int64_t srow = n * rows2 + 2 * tl + h;
            if (kh == 0) p.stat_sum[srow * p.N + n0 + 32 * w + cn] = d1[h][0];
            if (kh == ((cn >> 2) & 1)) p.stat_sq[srow * p.N + n0 + 32 * w + cn] = dq;
        }
    }
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void conv_nt2_kernel(NtArgs<bf16> p, int64_t n_groups) {
    using namespace n2;
    __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w >> 1, wn = w & 1;
    const int chunks = p.a_c / 32, nk = chunks * 3;
    const int row_bytes = p.a_c * 2;

    // tile of this workgroup; the n-tiles of one (window, t-tile) group go to workgroups 8 apart (same XCD, dispatched together)
    int64_t group;
    int tn;
    {
        const int64_t v = blockIdx.x;
        if (p.order == 1 && (n_groups & 7) == 0) {
            const int64_t j = v >> 3;
            tn = (int)(j % p.tilesN);
            group = (j / p.tilesN) * 8 + (v & 7);
        } else {
            group = v / p.tilesN;
            tn = (int)(v % p.tilesN);
        }
    }
    const int tl = (int)(group % p.tilesL);
    const int64_t n = group / p.tilesL;
    const int t0 = tl * TM, n0 = tn * TN;

    // ---- DMA sources: one instruction = 16 rows x 64 B, lane -> (row lane/4, physical chunk lane%4) ----
    const int lrow = lane >> 2, lchunk = lane & 3;
    const char* a_src[4];
    const char* b_src[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (w + 4 * i) * 16 + lrow;
        int t = t0 + row;
        t = t < p.L ? t : p.L - 1;
        a_src[i] = reinterpret_cast<const char*>(p.a + n * p.a_win_stride + (int64_t)t * p.a_c) + ((lchunk ^ ((row >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (w + 4 * i) * 16 + lrow;
        b_src[i] = reinterpret_cast<const char*>(p.bt + (int64_t)(n0 + row) * p.Ktot) + ((lchunk ^ ((row >> 2) & 3)) << 4);
    }
    auto issue = [&](int stage, int ko) {
        char* base = lds + stage * STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(a_src[i] + ko, base + __builtin_amdgcn_readfirstlane((w + 4 * i) * 1024));
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(b_src[i] + ko, base + A_BYTES + __builtin_amdgcn_readfirstlane((w + 4 * i) * 1024));
    };

    // ---- fragment geometry: lane (r, kh) reads chunk 2s + kh of its row ----
    const int r = lane & 31, kh = lane >> 5;
    int fa[4], fb[2], key_a[4], key_b[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wm * 128 + 32 * i + r;
        fa[i] = row * KB;
        key_a[i] = (row >> 2) & 3;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = wn * 64 + 32 * j + r;
        fb[j] = A_BYTES + row * KB;
        key_b[j] = (row >> 2) & 3;
    }

    f32x16 acc[4][2];
    n2_init_acc<EPI>(p, acc, n0 + wn * 64 + 4 * (lane >> 5));

    // K walk (chunk, tap): slice kt -> byte offset tap * row_bytes + chunk * 64 in an im2col / weight row
    int i_tap = 0, i_ko = 0, i_chunk_off = 0;  // of the next slice to ISSUE
    auto advance = [&]() {
        ++i_tap;
        i_ko += row_bytes;
        if (i_tap == 3) {
            i_tap = 0;
            i_chunk_off += KB;
            i_ko = i_chunk_off;
        }
    };
    issue(0, i_ko);
    advance();
    if (nk > 1) {
        issue(1, i_ko);
        advance();
    }
    int stage = 0, istage = 2;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) {
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk && !(p.ablate & 8)) {
            issue(istage, i_ko);
            advance();
            istage = istage == 2 ? 0 : istage + 1;
        }
        const char* base = lds + stage * STAGE;
        stage = stage == 2 ? 0 : stage + 1;
        // all 12 fragment reads of the slice are issued up front, in the order the MFMAs consume them (the compiler then waits with
        // counted lgkmcnt): the k-step-1 reads land under the 8 MFMAs of k-step 0, and what is exposed of the first reads is covered
        // by the other workgroup's wave on this SIMD
        bf16x8 a0[4], b0[2], a1[4], b1[2];
        b0[0] = *reinterpret_cast<const bf16x8*>(base + fb[0] + ((kh ^ key_b[0]) << 4));
        a0[0] = *reinterpret_cast<const bf16x8*>(base + fa[0] + ((kh ^ key_a[0]) << 4));
        b0[1] = *reinterpret_cast<const bf16x8*>(base + fb[1] + ((kh ^ key_b[1]) << 4));
#pragma unroll
        for (int i = 1; i < 4; ++i) a0[i] = *reinterpret_cast<const bf16x8*>(base + fa[i] + ((kh ^ key_a[i]) << 4));
#pragma unroll
        for (int j = 0; j < 2; ++j) b1[j] = *reinterpret_cast<const bf16x8*>(base + fb[j] + (((2 + kh) ^ key_b[j]) << 4));
#pragma unroll
        for (int i = 0; i < 4; ++i) a1[i] = *reinterpret_cast<const bf16x8*>(base + fa[i] + (((2 + kh) ^ key_a[i]) << 4));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0[j], a0[i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1[j], a1[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (p.ablate & 2) {  // timing experiments: no epilogue at all (one store keeps the accumulators live)
        if (acc[0][0][0] == 123.456f) p.out[0] = (bf16)acc[3][1][15];
        return;
    }
    n2_epilogue<EPI>(p, lds, acc, n, tl, t0, n0, TM, tid, lane, w, wm, wn);
}

// ------------------------------------------------------------------------------------------------
// conv_nt2_kernel with an INPUT-RESIDENT A operand (the structure of conv_w4_kernel / conv_tn8x_kernel applied to the
// two-workgroups-per-CU tile).  Ablation of conv_nt2_kernel at cfg-A (us; forward 256->384 / dgrad 384->512): full 278 / 244,
// no epilogue 204 / 196, no in-loop DMA 197 / 153, neither 142 / 136, DMA + fragment reads alone (no MFMA, no epilogue) 179 /
// 175: the global -> LDS stream is the longest pole -- it moves 24 KB per 64 MFMAs, 85 FLOP per byte of L2 -> LDS traffic.
// The A row of tap k at position t is input row t + k, so ONE staged block of 256 input rows x 32 channels serves the three
// taps of a channel chunk: per chunk 16 KB of A + 3 x 8 KB of B instead of 3 x 24 KB (153 FLOP/B).  An output tile is 254
// positions (MFMA rows 254, 255 are computed and dropped) so that a block is exactly 16 DMA instructions.
//
// Stream: K tile kt = 3 * chunk + tap.  B slices in a ring of three 8 KB stages (two ahead), A blocks in a ring of three 16 KB
// blocks (block chunk+2 is issued during taps 0 and 1 of `chunk`: four slices of lead for the HBM-streamed operand).  Per wave
// and iteration kt the issue order is  B(kt+2) [2 pieces], then half of A(chunk+2) [2 pieces, taps 0 and 1 only].
//   RAW: iteration kt needs B(kt) (first issue of iteration kt-2; kt = 0, 1: prologue) and, at tap 0, A(chunk) (issued four or
//        more slices earlier).  Loads complete in order, so s_waitcnt vmcnt(N) with N = pieces issued after B(kt), then the
//        barrier.  N is computed from the schedule (tail iterations issue less).
//   WAR: B(kt+2) overwrites the stage read in iteration kt-1, A(chunk+2) the block read during chunk-1; both are issued after
//        barrier kt, which every wave reaches only when its reads of iteration kt-1 have been consumed.
// Requires a_c % 32 == 0, Ktot == 3 * a_c, N % 128 == 0; forward with statistics: 2 * ceil(L / 254) == ceil(L / 128).
// Tried on this kernel without gain (round 2, interleaved A/B on the six cfg-A launches): s_setprio(2) around the K loop (+1 %
// time), a start offset between the two workgroups of a CU or across the chip (0 .. +3 %), the two operand streams issued by
// different waves so that vmcnt's in-order retirement does not tie the input blocks to the weight slices (+3 %), non-temporal
// output stores (-1.5 %, not kept: the next kernel reads the tensor).  Ablations: see DESIGN.md 4.2 -- and read them with care:
// an ablation that corrupts the DATA (NaN / zero tensors downstream) makes every later launch faster by itself, the chip clocks
// higher on such operands (a build whose statistics were broken ran the whole step 6 % faster, wgrad included).
// ------------------------------------------------------------------------------------------------
namespace n2r {
constexpr int TROWS = 254;
constexpr int A_BLK = 256 * 64, B_STG = 128 * 64;
constexpr int B0 = 3 * A_BLK;
constexpr int OPER = 3 * A_BLK + 3 * B_STG;  // 72 KB
static_assert(OPER <= n2::LDS_BYTES, "operand rings fit under the epilogue tile");
}  // namespace n2r

__device__ inline void wait_vm_0246(int n) {
    if (n >= 6) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else if (n == 4) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else if (n == 2) {
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

#if defined(VM_EXPERIMENT_PROFILE)  // where a wave of conv_nt2r_kernel spends its clocks (s_memtime); experiment builds only
constexpr int PROF_SLOTS = 8192 * 4;
__device__ unsigned int g_prof[PROF_SLOTS * 8];  // per (workgroup, wave): total, first K tile, other K tiles, epilogue
#define VM_PROF(...) __VA_ARGS__
#else
#define VM_PROF(...)
#endif

template <int EPI>
__global__ __launch_bounds__(256, 2) void conv_nt2r_kernel(NtArgs<bf16> p, int64_t n_groups) {
    VM_PROF(const long long pt_start = __builtin_amdgcn_s_memtime(); long long pt_first = 0, pt_bar = 0;)
    using namespace n2;
    using namespace n2r;
    __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w >> 1, wn = w & 1;
    const int chunks = p.a_c / 32, nk = chunks * 3;
    const int row_bytes = p.a_c * 2;

    // tile coordinates in 32-bit unsigned arithmetic (the launch guarantees grid < 2^31): the 64-bit divisions this replaces were a
    // visible part of the ~4 000 clocks a workgroup spent before its first DMA (profiles/r02_nt2r_wave_profile.txt)
    unsigned group;
    int tn;
    {
        const unsigned v = blockIdx.x, tiles_n = (unsigned)p.tilesN;
        if (p.order == 1 && (n_groups & 7) == 0) {
            const unsigned j = v >> 3, q = j / tiles_n;
            tn = (int)(j - q * tiles_n);
            group = q * 8 + (v & 7);
        } else {
            group = v / tiles_n;
            tn = (int)(v - group * tiles_n);
        }
    }
    const unsigned nw = group / (unsigned)p.tilesL;
    const int tl = (int)(group - nw * (unsigned)p.tilesL);
    const int64_t n = nw;
    const int t0 = tl * TROWS, n0 = tn * TN;
    // forward: the bias loads go out first and are consumed (accumulator init) only after the prologue DMA has been issued
    f32x4 bias4[2][4];
    n2_load_bias<EPI>(p, bias4, n0 + wn * 64 + 4 * (lane >> 5));

    // ---- DMA sources: one instruction = 16 rows x 64 B; A block row R <-> padded input row t0 + R (clamped to the L + 2 rows) ----
    const int lrow = lane >> 2, lchunk = lane & 3;
    const char* a_src[4];
    const char* b_src[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (w + 4 * i) * 16 + lrow;
        int pr = t0 + row;
        pr = pr < p.L + 2 ? pr : p.L + 1;
        a_src[i] = reinterpret_cast<const char*>(p.a + n * p.a_win_stride + (int64_t)pr * p.a_c) + ((lchunk ^ ((row >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (w + 4 * i) * 16 + lrow;
        b_src[i] = reinterpret_cast<const char*>(p.bt + (int64_t)(n0 + row) * p.Ktot) + ((lchunk ^ ((row >> 2) & 3)) << 4);
    }
    auto issue_a = [&](int blk, int chunk, int i0) {  // pieces i0, i0 + 1 of this wave's four
        char* base = lds + blk * A_BLK;
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16<VM_NT2R_AUX_A>(a_src[i0 + i] + chunk * KB, base + __builtin_amdgcn_readfirstlane((w + 4 * (i0 + i)) * 1024));
    };
    auto issue_b = [&](int stg, int ko) {
        char* base = lds + B0 + stg * B_STG;
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16<VM_NT2R_AUX_B>(b_src[i] + ko, base + __builtin_amdgcn_readfirstlane((w + 4 * i) * 1024));
    };

    // ---- fragment geometry: A row of tap k = block row m + k.  The four 32-row blocks of a wave are 2048 bytes apart and share the
    // swizzle key ((row >> 2) & 3 is unchanged by +32), so one address per (tap, k-step) serves them through the ds_read offset
    // field; rows 256 / 257 (taps 1, 2 of the dropped outputs 254, 255) read whatever follows the block -- valid LDS, results unused ----
    const int r = lane & 31, kh = lane >> 5;
    int a_addr[3][2], b_addr[2];
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) {
        const int row = wm * 128 + r + tap;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) a_addr[tap][ks] = row * KB + (((2 * ks + kh) ^ ((row >> 2) & 3)) << 4);
    }
    {
        const int row = wn * 64 + r;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) b_addr[ks] = B0 + row * KB + (((2 * ks + kh) ^ ((row >> 2) & 3)) << 4);
    }

#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)lds;
#else
    const uint32_t lds0 = 0;
#endif

    VM_PROF(const long long pt_s2 = __builtin_amdgcn_s_memtime();)
    // ---- prologue: A(0), B(0), A(1), B(1) in that order ----
    issue_a(0, 0, 0);
    issue_a(0, 0, 2);
    issue_b(0, 0);
    if (chunks > 1) {
        issue_a(1, 1, 0);
        issue_a(1, 1, 2);
    }
    issue_b(1, row_bytes);  // K tile 1 = (chunk 0, tap 1)
    VM_PROF(const long long pt_s3 = __builtin_amdgcn_s_memtime();)
    f32x16 acc[4][2];
    n2_fill_acc(acc, bias4);
    int n_wait = (chunks > 1 ? 4 : 0) + 2;  // pieces issued after B(0)
    int ia_prev = 0;                        // A pieces issued in the previous iteration (after its B pieces)
    // of the next B slice to issue (K tile kt + 2)
    int b_tap = 2, b_chunk_off = 0, b_stage = 2;
    int a_blk = 0, b_cur = 0;  // ring slots read by the current K tile
    // the other workgroup of this CU is usually in a different phase: its VALU-dense epilogue must not starve this wave's MFMA
    // issue (two waves per SIMD share one issue port), so the K loop runs at raised priority
    if (p.skew & 1) __builtin_amdgcn_s_setprio(2);
    if ((p.skew >> 1) > 0 && blockIdx.x < 512) {
        // experiment: de-synchronise the chip.  Every first-round workgroup starts ((blockIdx >> 3) & 7) x (skew >> 1) x 1016 clocks
        // late, so that the epilogue store bursts of the 512 resident workgroups do not all fall into the same time window.
        if (w == 0) {
            const int units = (int)((blockIdx.x >> 3) & 7) * (p.skew >> 1);
            for (int k = 0; k < units; ++k) __builtin_amdgcn_s_sleep(16);
        }
        __builtin_amdgcn_s_barrier();
    }
    for (int c = 0; c < chunks; ++c) {
        const bool more_a = c + 2 < chunks;
        const int a_next_blk = a_blk == 0 ? 2 : a_blk - 1;  // (c + 2) % 3
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
            const int kt = 3 * c + tap;
            wait_vm_0246(n_wait);
            __builtin_amdgcn_s_barrier();
            VM_PROF(if (kt == 0) pt_bar = __builtin_amdgcn_s_memtime();)
            int ib = 0, ia = 0;
            if (!(p.ablate & 8)) {
                if (kt + 2 < nk) {
                    issue_b(b_stage, b_tap * row_bytes + b_chunk_off);
                    b_stage = b_stage == 2 ? 0 : b_stage + 1;
                    if (++b_tap == 3) {
                        b_tap = 0;
                        b_chunk_off += KB;
                    }
                    ib = 2;
                }
                if (tap < 2 && more_a) {
                    issue_a(a_next_blk, c + 2, 2 * tap);
                    ia = 2;
                }
            }
            n_wait = ia_prev + ib + ia;  // pieces issued after B(kt + 1): the A pieces of kt - 1, then everything of kt
            ia_prev = ia;
            // ---- fragments: 12 x ds_read_b128 issued up front in consumption order, then COUNTED lgkmcnt waits (LDS returns in order):
            // the first MFMAs start when 3 reads have landed, the k-step-1 reads land under the MFMAs of k-step 0.  The reads are
            // inline asm because hipcc waits lgkmcnt(0) before the first use of any of them; every wait names the registers it
            // releases as in/out operands, which orders the MFMAs behind it ----
            const uint32_t aa0 = lds0 + a_blk * A_BLK + a_addr[tap][0], aa1 = lds0 + a_blk * A_BLK + a_addr[tap][1];
            const uint32_t bb0 = lds0 + b_cur * B_STG + b_addr[0], bb1 = lds0 + b_cur * B_STG + b_addr[1];  // b_addr contains B0
            b_cur = b_cur == 2 ? 0 : b_cur + 1;
            u32x4 a0[4], b0[2], a1[4], b1[2];
            asm volatile("ds_read_b128 %0, %1" : "=v"(b0[0]) : "v"(bb0));
            asm volatile("ds_read_b128 %0, %1" : "=v"(a0[0]) : "v"(aa0));
            asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(b0[1]) : "v"(bb0));
            asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(a0[1]) : "v"(aa0));
            asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(a0[2]) : "v"(aa0));
            asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(a0[3]) : "v"(aa0));
            asm volatile("ds_read_b128 %0, %1" : "=v"(b1[0]) : "v"(bb1));
            asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(b1[1]) : "v"(bb1));
            asm volatile("ds_read_b128 %0, %1" : "=v"(a1[0]) : "v"(aa1));
            asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(a1[1]) : "v"(aa1));
            asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(a1[2]) : "v"(aa1));
            asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(a1[3]) : "v"(aa1));
#define VM_MM(A, B, I, J) acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, B), __builtin_bit_cast(bf16x8, A), acc[I][J], 0, 0, 0)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(9)" : "+v"(b0[0]), "+v"(a0[0]), "+v"(b0[1]));
            VM_MM(a0[0], b0[0], 0, 0);
            VM_MM(a0[0], b0[1], 0, 1);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(a0[1]));
            VM_MM(a0[1], b0[0], 1, 0);
            VM_MM(a0[1], b0[1], 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(a0[2]));
            VM_MM(a0[2], b0[0], 2, 0);
            VM_MM(a0[2], b0[1], 2, 1);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a0[3]));
            VM_MM(a0[3], b0[0], 3, 0);
            VM_MM(a0[3], b0[1], 3, 1);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(b1[0]), "+v"(b1[1]), "+v"(a1[0]));
            VM_MM(a1[0], b1[0], 0, 0);
            VM_MM(a1[0], b1[1], 0, 1);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a1[1]));
            VM_MM(a1[1], b1[0], 1, 0);
            VM_MM(a1[1], b1[1], 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(a1[2]));
            VM_MM(a1[2], b1[0], 2, 0);
            VM_MM(a1[2], b1[1], 2, 1);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a1[3]));
            VM_MM(a1[3], b1[0], 3, 0);
            VM_MM(a1[3], b1[1], 3, 1);
            __builtin_amdgcn_sched_barrier(0);
#undef VM_MM
            __builtin_amdgcn_sched_barrier(0);
            VM_PROF(if (kt == 0) pt_first = __builtin_amdgcn_s_memtime();)
        }
        a_blk = a_blk == 2 ? 0 : a_blk + 1;
    }
    if (p.skew & 1) __builtin_amdgcn_s_setprio(0);
    if (p.ablate & 2) {
        if (acc[0][0][0] == 123.456f) p.out[0] = (bf16)acc[3][1][15];
        return;
    }
    VM_PROF(const long long pt_loop = __builtin_amdgcn_s_memtime();)
    n2_epilogue<EPI>(p, lds, acc, n, tl, t0, n0, TROWS, tid, lane, w, wm, wn);
#if defined(VM_EXPERIMENT_PROFILE)
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long pt_end = __builtin_amdgcn_s_memtime();
        if (lane == 0 && blockIdx.x < 8192) {
            unsigned int* q = g_prof + ((int64_t)blockIdx.x * 4 + w) * 8;
            q[0] = (unsigned int)(pt_end - pt_start);    // the whole wave-tile (incl. the drain of its stores)
            q[1] = (unsigned int)(pt_first - pt_start);  // start -> end of the first K tile
            q[2] = (unsigned int)(pt_loop - pt_first);   // the other nk - 1 K tiles
            q[3] = (unsigned int)(pt_end - pt_loop);     // epilogue + store drain
            q[4] = (unsigned int)(pt_s2 - pt_start);     // tile coordinates, bias loads, DMA / fragment addresses
            q[5] = (unsigned int)(pt_s3 - pt_s2);        // issue of the 10-14 prologue DMA instructions
            q[6] = (unsigned int)(pt_bar - pt_s3);       // accumulator init, first data wait, first barrier
            q[7] = (unsigned int)(pt_first - pt_bar);    // fragment reads + 16 MFMAs of the first K tile
        }
    }
#endif
}

#if defined(VM_EXPERIMENT_PROFILE)
extern "C" int vm_debug_prof_read(unsigned int* out, int n_slots) {  // out: n_slots x 4 host values
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned int) * 8 * (size_t)n_slots);
    return 0;
}
#endif

// ------------------------------------------------------------------------------------------------
// wgrad: TN GEMM with a transposing stager.  Output tile 128 (kk) x 128 (co); reduction over the positions
// of windows [w_begin, w_end).  Each stage brings BKP positions x 128 columns of both operands; a thread loads
// 4 consecutive positions x 16 bytes per item and writes them position-contiguous, so the fragment reads are the
// same 16-byte K-contiguous reads as in the NT kernel.
// ------------------------------------------------------------------------------------------------
template <typename T>
struct TnArgs {
    const T* x;   // padded input activations  (n_windows, L+2, c_in)
    const T* du;  // padded output gradients   (n_windows, L+2, c_out)
    float* ws;    // (splits, 3*c_in, c_out)
    int64_t x_win_stride, du_win_stride;
    int c_in, c_out, L;
    int Kk;  // 3*c_in
    int tilesI, tilesJ, splits;
    int xcd_remap;
    int ablate;  // conv_tn8_kernel timing experiments (wrong results): 4 no MFMA, 8 no in-loop DMA, 16 no fragment reads,
                 // 32 no A DMA, 64 no B DMA
    int64_t n_windows, win_per_split;
    int split = 0;  // fp32 storage only (dtype VM_F32S): split-bf16 products
};

template <typename T, int PITCH> struct Transpose4;
template <int PITCH> struct Transpose4<bf16, PITCH> {
    // 4 position rows of 8 bf16 -> 8 columns of 4 bf16 (8 bytes each)
    // v_perm_b32: result bytes selected from {first operand = bytes 7..4, second = bytes 3..0}
    __device__ static inline void store(char* lds_tile, int col0, int pg, const u32x4 (&v)[4]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t sel = (j & 1) ? 0x07060302u : 0x05040100u;  // high / low halves of the two dwords
            u32x2 o;
            o[0] = __builtin_amdgcn_perm(v[1][j >> 1], v[0][j >> 1], sel);
            o[1] = __builtin_amdgcn_perm(v[3][j >> 1], v[2][j >> 1], sel);
            *reinterpret_cast<u32x2*>(lds_tile + (col0 + j) * PITCH + pg * 8) = o;
        }
    }
};
template <int PITCH> struct Transpose4<float, PITCH> {
    __device__ static inline void store(char* lds_tile, int col0, int pg, const u32x4 (&v)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u32x4 o = {v[0][j], v[1][j], v[2][j], v[3][j]};
            *reinterpret_cast<u32x4*>(lds_tile + (col0 + j) * PITCH + pg * 16) = o;
        }
    }
};

// fp32 rows -> per channel column the 4 positions as 4 bf16 hi (8 bytes, hi plane) + 4 bf16 lo (8 bytes, lo plane at +64)
template <int PITCH>
struct Transpose4Split {
    __device__ static inline void store(char* lds_tile, int col0, int pg, const u32x4 (&v)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32x4 o = {v[0][j], v[1][j], v[2][j], v[3][j]};
            u32x2 h, l;
            split_f32x4(o, h, l);
            *reinterpret_cast<u32x2*>(lds_tile + (col0 + j) * PITCH + pg * 8) = h;
            *reinterpret_cast<u32x2*>(lds_tile + (col0 + j) * PITCH + 64 + pg * 8) = l;
        }
    }
};

template <typename T, int KB, bool SPLIT = false>
__global__ __launch_bounds__(256) void conv_tn_kernel(TnArgs<T> p) {
    static_assert(!SPLIT || (sizeof(T) == 4 && KB == 128), "split-bf16 arithmetic: fp32 storage, 128-byte stages");
    using G = Geo<KB>;
    constexpr int VEC = Elem<T>::kVec;
    constexpr int BKP = KB / (int)sizeof(T);  // positions per stage
    constexpr int PG = BKP / 4;               // groups of 4 positions
    constexpr int ITEMS = (128 / VEC) * PG;   // (column group, position group) items per operand
    constexpr int NIT = ITEMS / 128;          // items per thread (threads 0..127 stage X, 128..255 stage dU)
    __shared__ __attribute__((aligned(16))) char lds[2 * 2 * G::TILE];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1;
    // All output tiles of one split stream the SAME positions of X and dU, so they should share an L2: workgroup b
    // runs on XCD b % 8 (observed dispatch order; only speed depends on it), hence split s is given the workgroups
    // {b : b % 8 == s % 8}.  Splits beyond the last multiple of 8 fall back to the plain order.
    int64_t b = blockIdx.x;
    {
        const int64_t NT = (int64_t)p.tilesI * p.tilesJ;
        const int64_t full = (int64_t)(p.splits / 8) * 8 * NT;
        if (p.xcd_remap && b < full) {
            const int64_t xcd = b & 7, local = b >> 3;
            b = ((local / NT) * 8 + xcd) * NT + local % NT;
        }
    }
    const int tj = (int)(b % p.tilesJ);
    b /= p.tilesJ;
    const int ti = (int)(b % p.tilesI);
    const int split = (int)(b / p.tilesI);
    const int i0 = ti * BM, j0 = tj * BN;

    const bool is_x = tid < 128;
    const T* base0 = is_x ? p.x : p.du + p.c_out;  // +1 halo row: dU row t lives at padded row t+1
    const int64_t win_stride = is_x ? p.x_win_stride : p.du_win_stride;
    const int row_c = is_x ? p.c_in : p.c_out;
    const int which = is_x ? 0 : 1;
    int pg[NIT], col0[NIT], toff[NIT];
    bool col_ok[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int item = (tid & 127) + it * 128;
        pg[it] = item % PG;
        col0[it] = (item / PG) * VEC;
        const int gcol = (is_x ? i0 : j0) + col0[it];
        col_ok[it] = gcol < (is_x ? p.Kk : p.c_out);
        toff[it] = pg[it] * 4 * row_c + (col_ok[it] ? gcol : 0);  // element offset of this item's first row in a stage
    }

    const int64_t w_begin = (int64_t)split * p.win_per_split;
    int64_t w_end = w_begin + p.win_per_split;
    if (w_end > p.n_windows) w_end = p.n_windows;
    const int stages_per_win = (p.L + BKP - 1) / BKP;
    const int64_t n_stages = (w_end - w_begin) * stages_per_win;

    f32x16 acc[2][2];
    zero_acc(acc);

    // (window, stage-in-window) cursor of the NEXT stage to load -- incremented, never divided
    int64_t ld_n = w_begin;
    int ld_s = 0;
    u32x4 rv[NIT][4];
    auto gload = [&]() {
        const int tb = ld_s * BKP;
        const T* wbase = base0 + ld_n * win_stride + (int64_t)tb * row_c;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = tb + pg[it] * 4 + r;
                if (col_ok[it] && t < p.L) {
                    rv[it][r] = *reinterpret_cast<const u32x4*>(wbase + toff[it] + r * row_c);
                } else {
                    rv[it][r] = u32x4{0, 0, 0, 0};
                }
            }
        }
        if (++ld_s == stages_per_win) {
            ld_s = 0;
            ++ld_n;
        }
    };
    if (n_stages > 0) gload();
    for (int64_t st = 0; st < n_stages; ++st) {
        const int buf = (int)(st & 1);
        char* mine = lds + (buf * 2 + which) * G::TILE;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if constexpr (SPLIT) {
                Transpose4Split<G::PITCH>::store(mine, col0[it], pg[it], rv[it]);
            } else {
                Transpose4<T, G::PITCH>::store(mine, col0[it], pg[it], rv[it]);
            }
        }
        __syncthreads();
        if (st + 1 < n_stages) gload();
        if constexpr (SPLIT) {
            mma_slice_split<KB>(lds + (buf * 2 + 0) * G::TILE, lds + (buf * 2 + 1) * G::TILE, wm, wn, lane, acc);
        } else {
            mma_slice<T, KB>(lds + (buf * 2 + 0) * G::TILE, lds + (buf * 2 + 1) * G::TILE, wm, wn, lane, acc);
        }
    }

    // slab tile: row = kk (m side), 4 consecutive co per register group -> 16-byte fp32 stores
    float* out = p.ws + (int64_t)split * p.Kk * p.c_out;
    const int hi = lane >> 5;
#pragma unroll
    for (int im = 0; im < 2; ++im) {
        const int row = i0 + wm * 64 + im * 32 + (lane & 31);
        if (row >= p.Kk) continue;
#pragma unroll
        for (int in = 0; in < 2; ++in) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = j0 + wn * 64 + in * 32 + 8 * g + 4 * hi;
                if (col < p.c_out) {  // c_out is a multiple of 8 -> the 4 columns are all valid
                    f32x4 v = {acc[im][in][4 * g], acc[im][in][4 * g + 1], acc[im][in][4 * g + 2], acc[im][in][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(out + (int64_t)row * p.c_out + col) = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// wgrad, 256 x 256 output tile.  The 128 x 128 kernel above is bound by LDS traffic, not by the matrix cores: per
// 128-byte stage it writes 32 KB (transposed 8-byte writes, ~85 B/clk) and reads 64 KB for 16 MFMAs per wave.  With a
// 256 x 256 tile and 64 x 128 per wave (8 waves) a stage writes 64 KB and reads 192 KB for 32 MFMAs per wave: LDS
// cycles per MFMA cycle drop from 1.25 to 0.75.  One workgroup per CU (144 KB of LDS, 128 accumulator registers).
template <typename T, int KB, bool SPLIT = false>
__global__ __launch_bounds__(512) void conv_tn256_kernel(TnArgs<T> p) {
    static_assert(!SPLIT || (sizeof(T) == 4 && KB == 128), "split-bf16 arithmetic: fp32 storage, 128-byte stages");
    using G = Geo<KB>;
    constexpr int VEC = Elem<T>::kVec;
    constexpr int BKP = KB / (int)sizeof(T);  // positions per stage
    constexpr int PG = BKP / 4;
    constexpr int TM = 256, TN_ = 256;
    constexpr int OPB = TM * G::PITCH;        // bytes of one operand tile (256 rows)
    constexpr int KSTEPS = KB / Mfma<T>::KSTEP_BYTES;
    static_assert((TM / VEC) * PG == 512, "one item per thread and operand");
    __shared__ __attribute__((aligned(16))) char lds[2 * 2 * OPB];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1;
    int64_t b = blockIdx.x;
    {
        const int64_t NT = (int64_t)p.tilesI * p.tilesJ;
        const int64_t full = (int64_t)(p.splits / 8) * 8 * NT;
        if (p.xcd_remap && b < full) {
            const int64_t xcd = b & 7, local = b >> 3;
            b = ((local / NT) * 8 + xcd) * NT + local % NT;
        }
    }
    const int tj = (int)(b % p.tilesJ);
    b /= p.tilesJ;
    const int ti = (int)(b % p.tilesI);
    const int split = (int)(b / p.tilesI);
    const int i0 = ti * TM, j0 = tj * TN_;

    // every thread stages one item of X and one of dU per stage
    const int pg = tid % PG, col0 = (tid / PG) * VEC;
    const bool x_ok = i0 + col0 < p.Kk, d_ok = j0 + col0 < p.c_out;
    const int x_toff = pg * 4 * p.c_in + (x_ok ? i0 + col0 : 0);
    const int d_toff = pg * 4 * p.c_out + (d_ok ? j0 + col0 : 0);
    const T* d_base0 = p.du + p.c_out;  // dU row t lives at padded row t+1

    const int64_t w_begin = (int64_t)split * p.win_per_split;
    int64_t w_end = w_begin + p.win_per_split;
    if (w_end > p.n_windows) w_end = p.n_windows;
    const int stages_per_win = (p.L + BKP - 1) / BKP;
    const int64_t n_stages = (w_end - w_begin) * stages_per_win;

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Two register sets: the loads of stage st+2 are issued while stage st is multiplied, so a load has two full stages
    // (~2 us of MFMA at one workgroup per CU) to come back -- with one set the K loop was bound by that latency.
    int64_t ld_n = w_begin;
    int ld_s = 0;
    u32x4 rx0[4], rd0[4], rx1[4], rd1[4];
    auto gload = [&](u32x4 (&rx)[4], u32x4 (&rd)[4]) {
        const int tb = ld_s * BKP;
        const T* xb = p.x + ld_n * p.x_win_stride + (int64_t)tb * p.c_in;
        const T* db = d_base0 + ld_n * p.du_win_stride + (int64_t)tb * p.c_out;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool tok = tb + pg * 4 + r < p.L;
            rx[r] = (x_ok && tok) ? *reinterpret_cast<const u32x4*>(xb + x_toff + r * p.c_in) : u32x4{0, 0, 0, 0};
            rd[r] = (d_ok && tok) ? *reinterpret_cast<const u32x4*>(db + d_toff + r * p.c_out) : u32x4{0, 0, 0, 0};
        }
        if (++ld_s == stages_per_win) {
            ld_s = 0;
            ++ld_n;
        }
    };
    const int r = lane & 31, kh = lane >> 5;
    auto step = [&](int64_t st, u32x4 (&rx)[4], u32x4 (&rd)[4]) {
        char* ta = lds + (int)(st & 1) * 2 * OPB;
        char* tb_ = ta + OPB;
        if constexpr (SPLIT) {
            Transpose4Split<G::PITCH>::store(ta, col0, pg, rx);
            Transpose4Split<G::PITCH>::store(tb_, col0, pg, rd);
        } else {
            Transpose4<T, G::PITCH>::store(ta, col0, pg, rx);
            Transpose4<T, G::PITCH>::store(tb_, col0, pg, rd);
        }
        __syncthreads();
        if (st + 2 < n_stages) gload(rx, rd);
        const char* pa = ta + (wm * 64 + r) * G::PITCH;
        const char* pb = tb_ + (wn * 128 + r) * G::PITCH;
        if constexpr (SPLIT) {
            using M = Mfma<bf16>;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const M::Frag a0h = M::load(pa, s, kh), a1h = M::load(pa + 32 * G::PITCH, s, kh);
                const M::Frag a0l = M::load(pa, s + 2, kh), a1l = M::load(pa + 32 * G::PITCH, s + 2, kh);
#pragma unroll
                for (int in = 0; in < 4; ++in) {
                    const M::Frag bh = M::load(pb + in * 32 * G::PITCH, s, kh), bl = M::load(pb + in * 32 * G::PITCH, s + 2, kh);
                    acc[0][in] = M::run(bl, a0h, acc[0][in]);
                    acc[1][in] = M::run(bl, a1h, acc[1][in]);
                    acc[0][in] = M::run(bh, a0l, acc[0][in]);
                    acc[1][in] = M::run(bh, a1l, acc[1][in]);
                    acc[0][in] = M::run(bh, a0h, acc[0][in]);
                    acc[1][in] = M::run(bh, a1h, acc[1][in]);
                }
            }
            return;
        }
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            typename Mfma<T>::Frag a0 = Mfma<T>::load(pa, s, kh), a1 = Mfma<T>::load(pa + 32 * G::PITCH, s, kh);
#pragma unroll
            for (int in = 0; in < 4; ++in) {
                typename Mfma<T>::Frag bf = Mfma<T>::load(pb + in * 32 * G::PITCH, s, kh);
                acc[0][in] = Mfma<T>::run(bf, a0, acc[0][in]);
                acc[1][in] = Mfma<T>::run(bf, a1, acc[1][in]);
            }
        }
    };
    if (n_stages > 0) gload(rx0, rd0);
    if (n_stages > 1) gload(rx1, rd1);
    for (int64_t st = 0; st < n_stages; st += 2) {
        step(st, rx0, rd0);
        if (st + 1 < n_stages) step(st + 1, rx1, rd1);
    }

    float* out = p.ws + (int64_t)split * p.Kk * p.c_out;
    const int hi = lane >> 5;
#pragma unroll
    for (int im = 0; im < 2; ++im) {
        const int row = i0 + wm * 64 + im * 32 + (lane & 31);
        if (row >= p.Kk) continue;
#pragma unroll
        for (int in = 0; in < 4; ++in) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = j0 + wn * 128 + in * 32 + 8 * g + 4 * hi;
                if (col < p.c_out) {
                    f32x4 v = {acc[im][in][4 * g], acc[im][in][4 * g + 1], acc[im][in][4 * g + 2], acc[im][in][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(out + (int64_t)row * p.c_out + col) = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// wgrad, 256 x 256 output tile, LDS-DMA staging + hardware-transposing fragment reads (bf16 only).
//
// Both operands of the TN GEMM are position-major in memory (X[pos][ci], dU[pos][co]) while an MFMA lane wants 8
// consecutive POSITIONS of one channel.  The kernels above transpose in registers (4 x 16-byte loads, v_perm_b32,
// 8-byte ds_writes).  gfx950 can transpose on the LDS read instead: ds_read_b64_tr_b16 gives lane i of a 16-lane group
// element (i & 3) of the 8-byte words read by lanes 4j + (i >> 2), j = 0..3 (tools/probe/tr_read_probe.hip) -- i.e. column
// i of a [4 positions][16 channels] block.  So the tiles are staged UNtransposed by global_load_lds (no VALU, no ds_write)
// as blocks of [64 positions][32 channels] (64-byte rows: the 32 lanes served per LDS cycle read 4 rows x 64 bytes = 256
// contiguous bytes, conflict-free), and a fragment is two transposing reads.
//
// The pipeline is the one of conv_nt8_kernel (2-phase form): 8 waves as 2 (kk) x 4 (co), 128 x 64 per wave, a stage of 64
// positions staged as four 16 KB half tiles (A_lo / A_hi: kk rows 0-63 / 64-127 of every wave's 128; B_c0 / B_c1: co
// columns 0-31 / 32-63 of every wave's 64), two stages resident, waves 4-7 one slot behind waves 0-3, counted vmcnt.
// No epilogue inside the stream: the 256 x 256 fp32 tile is written once, to the split's slab.
//   phase 0: read A_lo, B_c0, B_c1 | DMA B1(g+1), A1(g+1) | MFMA A_lo x (B_c0, B_c1)
//   phase 1: read A_hi             | DMA A0(g+2), B0(g+2) | vmcnt(4): stage g+1 has landed | MFMA A_hi x (B_c0, B_c1)
// Positions past the end of a window are neutralised on the dU side: their source row is the (zero) halo row L+1 of the
// padded dU tensor, so whatever X row is fetched for them contributes nothing.
// ------------------------------------------------------------------------------------------------
namespace t8 {
constexpr int BLK = 64 * 64;           // one block: 64 positions x 32 channels (64-byte rows)
constexpr int HALF = 4 * BLK;          // 16 KB
constexpr int BUF = 4 * HALF;          // A_lo A_hi B_c0 B_c1
constexpr int LDS_BYTES = 2 * BUF;     // 128 KB
typedef __attribute__((ext_vector_type(4))) short s16x4;
}  // namespace t8

__global__ __launch_bounds__(512) void conv_tn8_kernel(TnArgs<bf16> p) {
    using namespace t8;
    using p8::FragA;
    using p8::FragB;
    using p8::H_A0;
    using p8::H_A1;
    using p8::H_B0;
    using p8::H_B1;
    __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 2, wn = w & 3;

    int64_t b = blockIdx.x;
    {
        const int64_t NT = (int64_t)p.tilesI * p.tilesJ;
        const int64_t full = (int64_t)(p.splits / 8) * 8 * NT;
        if (p.xcd_remap && b < full) {
            const int64_t xcd = b & 7, local = b >> 3;
            b = ((local / NT) * 8 + xcd) * NT + local % NT;
        }
    }
    const int tj = __builtin_amdgcn_readfirstlane((int)(b % p.tilesJ));
    b /= p.tilesJ;
    const int ti = __builtin_amdgcn_readfirstlane((int)(b % p.tilesI));
    const int split = __builtin_amdgcn_readfirstlane((int)(b / p.tilesI));
    const int i0 = ti * 256, j0 = tj * 256;

    const int w_begin = (int)((int64_t)split * p.win_per_split);
    int w_end = w_begin + (int)p.win_per_split;
    if (w_end > (int)p.n_windows) w_end = (int)p.n_windows;
    const int spw = (p.L + 63) / 64;  // stages per window
    const int G = w_end > w_begin ? (w_end - w_begin) * spw : 0;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if (G > 0) {
        // ---- DMA geometry: a half tile is 16 wave-instructions of 1 KB (16 positions x 64 bytes of one block) ----
        // instruction q = 8 j + w (j = 0, 1): block q >> 2, position quarter q & 3; lane -> position lane >> 2, chunk lane & 3
        const int dpos = (w & 3) * 16 + (lane >> 2);
        const int dchunk = (lane & 3) * 8;  // elements
        const char* const x_base = reinterpret_cast<const char*>(p.x);
        const char* const d_base = reinterpret_cast<const char*>(p.du);
        const int abl = p.ablate;
        auto stage = [&](int h, int buf, int n, int st) {
            if ((abl & 32) && h < 2) return;
            if ((abl & 64) && h >= 2) return;
            char* dst = lds + buf * BUF + h * HALF + w * 1024;
            int t = st * 64 + dpos;
            if (h < 2) {
                t = t < p.L ? t : p.L - 1;
                const char* src = x_base + n * p.x_win_stride * 2;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int bq = j * 2 + (w >> 2);  // block: row group bq >> 1, 32-row block bq & 1
                    int kk0 = i0 + (bq >> 1) * 128 + h * 64 + (bq & 1) * 32;
                    kk0 = kk0 < p.Kk ? kk0 : 0;
                    glds16(src + (unsigned)(t * p.c_in + kk0 + dchunk) * 2u, dst + j * 8192);
                }
            } else {
                t = (t < p.L ? t : p.L) + 1;  // row L+1 of the padded tensor is the zero halo
                const char* src = d_base + n * p.du_win_stride * 2;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int bq = j * 2 + (w >> 2);  // block = column group (wave column) bq
                    int co0 = j0 + bq * 64 + (h - 2) * 32;
                    co0 = co0 < p.c_out ? co0 : 0;
                    glds16(src + (unsigned)(t * p.c_out + co0 + dchunk) * 2u, dst + j * 8192);
                }
            }
        };

        // ---- fragment geometry: two transposing 8-byte reads per fragment ----
        const int li = lane & 15, lg = (lane >> 4) & 1, kh = lane >> 5;
        const int lane_off = (kh * 8 + (li >> 2)) * 64 + lg * 32 + (li & 3) * 8;
        // Inline asm on purpose: for the ds_read_tr builtin hipcc inserts s_waitcnt vmcnt(0) before every read that follows
        // a global_load_lds (it cannot tell which LDS bytes the DMA writes), which would drain the DMA pipeline twice per
        // stage.  The asm form is invisible to that pass -- and to its lgkmcnt bookkeeping: read_done() below is the wait.
        auto tr8 = [&](const char* ptr) -> bf16x8 {
            const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)ptr;
            u32x2 lo, hi;
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:256" : "=v"(hi) : "v"(a));
            const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
            return __builtin_bit_cast(bf16x8, v);
        };
        auto read_a = [&](FragA& fa, int buf, int ih) {
            if (abl & 16) return;
            const char* base = lds + buf * BUF + ih * HALF + wm * 2 * BLK + lane_off;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int s = 0; s < 4; ++s) fa[i][s] = tr8(base + i * BLK + s * 1024);
        };
        auto read_b = [&](FragB& fb, int buf, int jn) {
            if (abl & 16) return;
            const char* base = lds + buf * BUF + (2 + jn) * HALF + wn * BLK + lane_off;
#pragma unroll
            for (int s = 0; s < 4; ++s) fb[s] = tr8(base + s * 1024);
        };
        auto mma = [&](const FragA& fa, const FragB& fb, f32x16& c0, f32x16& c1) {
            if (abl & 4) return;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[s], fa[0][s], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[s], fa[1][s], c1, 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
        };
        auto slot_end = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        auto read_done = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };

        // stream cursors: (n1, s1) = position g+1, (n2, s2) = position g+2
        int n1 = w_begin, s1 = 0, n2, s2;
        auto advance = [&](int& n, int& s) {
            if (++s == spw) {
                s = 0;
                ++n;
            }
        };
        // ---- prologue: stage 0 whole, A0/B0 of stage 1 ----
        stage(H_A0, 0, n1, s1);
        stage(H_B1, 0, n1, s1);
        stage(H_A1, 0, n1, s1);
        stage(H_B0, 0, n1, s1);
        advance(n1, s1);  // -> position 1
        n2 = n1;
        s2 = s1;
        if (G > 1) {
            stage(H_A0, 1, n1, s1);
            stage(H_B0, 1, n1, s1);
            wait_vmcnt<4>();
        } else {
            wait_vmcnt<0>();
        }
        advance(n2, s2);  // -> position 2
        slot_end();
        if (wm == 1) slot_end();  // kk rows 128-255 run one slot behind

        FragA fa;
        FragB fb, fb1;
        if (abl & 16) {
#pragma unroll
            for (int s = 0; s < 4; ++s) fa[0][s] = fa[1][s] = fb[s] = fb1[s] = bf16x8{};
        }
        for (int g = 0; g < G; ++g) {
            const int buf = g & 1;
            read_a(fa, buf, 0);
            read_b(fb, buf, 0);
            read_b(fb1, buf, 1);
            if (g + 1 < G && !(abl & 8)) {
                stage(H_B1, buf ^ 1, n1, s1);
                stage(H_A1, buf ^ 1, n1, s1);
            }
            read_done();
            slot_end();
            mma(fa, fb, acc[0][0], acc[1][0]);
            mma(fa, fb1, acc[0][1], acc[1][1]);
            slot_end();
            read_a(fa, buf, 1);
            if (g + 2 < G && !(abl & 8)) {
                stage(H_A0, buf, n2, s2);
                stage(H_B0, buf, n2, s2);
                if (abl & 96) {
                    wait_vmcnt<2>();
                } else {
                    wait_vmcnt<4>();
                }
            } else {
                wait_vmcnt<0>();
            }
            read_done();
            slot_end();
            mma(fa, fb, acc[2][0], acc[3][0]);
            mma(fa, fb1, acc[2][1], acc[3][1]);
            slot_end();
            n1 = n2;
            s1 = s2;
            advance(n2, s2);
        }
        if (wm == 0) slot_end();  // balance the barrier count of the two groups
    }

    // ---- the split's slab tile ----
    float* out = p.ws + (int64_t)split * p.Kk * p.c_out;
    const int hi = lane >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = i0 + wm * 128 + i * 32 + (lane & 31);
        if (row >= p.Kk) continue;
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int col = j0 + wn * 64 + jn * 32 + 8 * g4 + 4 * hi;
                if (col < p.c_out) {
                    const f32x4 v = {acc[i][jn][4 * g4], acc[i][jn][4 * g4 + 1], acc[i][jn][4 * g4 + 2], acc[i][jn][4 * g4 + 3]};
                    *reinterpret_cast<f32x4*>(out + (int64_t)row * p.c_out + col) = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// wgrad, input-resident form: output tile = (3 taps x 128 input channels) x 128 output channels.
//
// Ablating conv_tn8_kernel shows that it is bound by its LDS-DMA traffic, not by the matrix cores (block 3, us: full 316,
// no MFMA 300, no DMA 202): with kk = tap * C_in + ci tiled 256-wide, a tile's A operand is re-fetched for every tap and
// every output-channel tile, and 256-wide tiles waste 10-25 % on C_out = 384 / Kk = 384.  Here a workgroup owns 128 input
// channels for ALL THREE taps: the A operand of tap t at position p is X[p + t], so one staged block of X rows serves the
// three taps (the fragment reads just start 0, 1 or 2 rows later) and the DMA bytes per MFMA drop by ~45 %; the tiles
// (384 x 128) divide every layer of the model exactly.
//   * A lives in a ring of 256 position rows per 32-channel block (4 blocks, 64 KB): stage g occupies rows (g & 3) * 64..+63,
//     tap reads run up to 2 rows into the next stage's rows (ring indices wrap with an AND).  B (dU) has 4 stage buffers of
//     4 blocks [64 positions][32 channels] (64 KB).  A stage is 64 positions; a window takes ceil((L + 2) / 64) stages so that
//     its last stage holds the zero halo row L + 1; positions >= L are neutralised on the dU side (source row L + 1 = zero
//     halo), so what the A rows of such positions hold does not matter as long as it is finite (the ring is zeroed once).
//   * 8 waves = 4 (input-channel blocks of 32) x 2 (64 output channels): a wave owns 3 taps x 32 ci x 64 co = 6 accumulator
//     tiles; 24 MFMAs per stage in clusters of 8 and 16; waves 4-7 (channel blocks 2, 3) run one slot behind waves 0-3.
//   * DMA runs three stages ahead: stage g + 3 is issued in the second READ slot of stage g (into the ring slot of stage
//     g - 1, whose last reads completed a phase earlier) and the counted vmcnt(4) there retires stage g + 2 -- stage g + 1
//     needs it for its tap overflow rows.
//   phase 0: read A(t0), A(t1), B_c0, B_c1 | MFMA t0 x c0, t0 x c1, t1 x c0
//   phase 1: read A(t2) | DMA stage g+3 | vmcnt(4) | MFMA t1 x c1, t2 x c0, t2 x c1
// ------------------------------------------------------------------------------------------------
namespace t8x {
constexpr int ROWS = 256;                  // ring rows per A block
constexpr int ABLK = ROWS * 128;           // 32 KB: one block = 64 channels, 128-byte rows (whole cache lines per DMA row)
constexpr int A_BYTES = 2 * ABLK;          // 64 KB
constexpr int BBLK = 64 * 128;             // 8 KB
constexpr int BSTAGE = 2 * BBLK;           // 16 KB
constexpr int LDS_BYTES = A_BYTES + 4 * BSTAGE;  // 128 KB
struct Frag4 {  // 4 k-steps; the two 8-byte halves are only joined at the MFMA, i.e. after the lgkmcnt wait
    u32x2 lo[4], hi[4];
};
}  // namespace t8x

template <bool FREE>
__global__ __launch_bounds__(512) void conv_tn8x_kernel(TnArgs<bf16> p) {
    using namespace t8x;
    __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;  // channel block (0..3), output-column half (0..1); waves 4-7 = blocks 2, 3

    int64_t b = blockIdx.x;
    {
        const int64_t NT = (int64_t)p.tilesI * p.tilesJ;
        const int64_t full = (int64_t)(p.splits / 8) * 8 * NT;
        if (p.xcd_remap && b < full) {
            const int64_t xcd = b & 7, local = b >> 3;
            b = ((local / NT) * 8 + xcd) * NT + local % NT;
        }
    }
    const int tj = __builtin_amdgcn_readfirstlane((int)(b % p.tilesJ));
    b /= p.tilesJ;
    const int ti = __builtin_amdgcn_readfirstlane((int)(b % p.tilesI));
    const int split = __builtin_amdgcn_readfirstlane((int)(b / p.tilesI));
    const int ci0 = ti * 128, j0 = tj * 128;

    const int w_begin = (int)((int64_t)split * p.win_per_split);
    int w_end = w_begin + (int)p.win_per_split;
    if (w_end > (int)p.n_windows) w_end = (int)p.n_windows;
    const int spw = (p.L + 2 + 63) / 64;  // stages per window (the last one holds the halo row L + 1)
    const int G = w_end > w_begin ? (w_end - w_begin) * spw : 0;

    f32x16 acc[3][2];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][j][e] = 0.f;

    if (G > 0) {
        // zero the A ring once: tap-overflow reads of never-staged rows must be finite
        for (int i = tid * 16; i < A_BYTES; i += 512 * 16) *reinterpret_cast<u32x4*>(lds + i) = u32x4{0, 0, 0, 0};
        __syncthreads();

        // ---- DMA geometry: per stage 16 wave-instructions for A and 16 for B; one instruction = 8 position rows x 128 bytes
        // (64 channels: whole 128-byte lines -- with 64-byte rows two instructions fetched the halves of every line and the
        // vector L1 spent half its time on hits-on-miss).  Inside a row the two 64-byte halves (32 channels each) are swapped
        // when bit 1 of the row index is set, so that the 4 rows x 64 bytes a transposing read touches fall in 4 different
        // 64-byte bank segments; the swap is applied to the per-lane SOURCE chunk here and again in the read addresses.
        const int drow = w * 8 + (lane >> 3);
        const int dchunk = ((lane & 7) ^ (((lane >> 4) & 1) << 2)) * 8;  // elements; (row >> 1) & 1 == (lane >> 4) & 1
        const char* const x_base = reinterpret_cast<const char*>(p.x);
        const char* const d_base = reinterpret_cast<const char*>(p.du);
        const int abl = p.ablate;
        auto stage = [&](int slot, int n, int st) {
            const int t = st * 64 + drow;
            if (!(abl & 32)) {
                int r = t < p.L + 1 ? t : p.L + 1;  // padded row of tap 0 at position t; rows past the halo are never used
                const char* src = x_base + n * p.x_win_stride * 2;
                char* dst = lds + slot * 8192 + w * 1024;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    int c0 = ci0 + j * 64;
                    c0 = c0 < p.c_in ? c0 : 0;
                    glds16<VM_TNX_AUX_A>(src + (unsigned)(r * p.c_in + c0 + dchunk) * 2u, dst + j * ABLK);
                }
            }
            if (!(abl & 64)) {
                const int r = (t < p.L ? t : p.L) + 1;  // row L + 1 of the padded dU tensor is the zero halo
                const char* src = d_base + n * p.du_win_stride * 2;
                char* dst = lds + A_BYTES + slot * BSTAGE + w * 1024;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    int co0 = j0 + j * 64;
                    co0 = co0 < p.c_out ? co0 : 0;
                    glds16<VM_TNX_AUX_B>(src + (unsigned)(r * p.c_out + co0 + dchunk) * 2u, dst + j * BBLK);
                }
            }
        };

        // ---- fragment reads (transposing, see conv_tn8_kernel) ----
        // lane -> row (kh * 8 + (li >> 2)) of the 16 rows of a k-step, channel lg * 16 + li of the wave's 32-channel half
        const int li = lane & 15, lg = (lane >> 4) & 1, kh = lane >> 5;
        const int rowl = kh * 8 + (li >> 2);
        const int sub = lg * 32 + (li & 3) * 8;
        // A: half (wm & 1) of 64-channel block (wm >> 1); the swap bit of ring row U + rowl + tap (U % 4 == 0) depends on tap
        int a_off[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) a_off[t] = (rowl + t) * 128 + (((wm & 1) ^ (((rowl + t) >> 1) & 1)) * 64) + sub;
        const int b_off = rowl * 128 + ((((rowl >> 1) & 1)) * 64) + sub;  // xor with the column half jn below
        const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)lds;
        auto tr_pair = [&](u32x2& lo, u32x2& hi, uint32_t a_lo, uint32_t a_hi) {
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a_lo));
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(a_hi));
        };
        auto read_a = [&](Frag4& fa, int slot, int tap) {
            if (abl & 16) return;
            const uint32_t blk = lds0 + (wm >> 1) * ABLK;
            const uint32_t u = slot * 8192 + a_off[tap];
#pragma unroll
            for (int s = 0; s < 4; ++s)
                tr_pair(fa.lo[s], fa.hi[s], blk + ((u + s * 2048) & (ABLK - 1)), blk + ((u + s * 2048 + 512) & (ABLK - 1)));
        };
        auto read_b = [&](Frag4& fb, int slot, int jn) {
            if (abl & 16) return;
            const uint32_t a = lds0 + A_BYTES + slot * BSTAGE + wn * BBLK + (b_off ^ (jn * 64));
#pragma unroll
            for (int s = 0; s < 4; ++s) tr_pair(fb.lo[s], fb.hi[s], a + s * 2048, a + s * 2048 + 512);
        };
        // clusters of 8 (one tap) and 16 (two taps) MFMAs, k-steps interleaved over the accumulator tiles
        auto opf = [](const Frag4& f, int s) {
            const u32x4 v = {f.lo[s][0], f.lo[s][1], f.hi[s][0], f.hi[s][1]};
            return __builtin_bit_cast(bf16x8, v);
        };
        auto mma_a = [&](const Frag4& a0, const Frag4& b0, const Frag4& b1, f32x16& c0, f32x16& c1) {
            if (abl & 4) return;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(opf(b0, s), opf(a0, s), c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(opf(b1, s), opf(a0, s), c1, 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
        };
        auto mma_b = [&](const Frag4& a0, const Frag4& a1, const Frag4& b0, const Frag4& b1, f32x16& c00, f32x16& c01, f32x16& c10,
                         f32x16& c11) {
            if (abl & 4) return;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                c00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(opf(b0, s), opf(a0, s), c00, 0, 0, 0);
                c01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(opf(b1, s), opf(a0, s), c01, 0, 0, 0);
                c10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(opf(b0, s), opf(a1, s), c10, 0, 0, 0);
                c11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(opf(b1, s), opf(a1, s), c11, 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
        };
        auto slot_end = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        auto read_done = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };

        // stream cursor of the stage to be staged next
        int nn = w_begin, ss = 0, staged = 0;
        auto stage_next = [&]() {
            stage(staged & 3, nn, ss);
            ++staged;
            if (++ss == spw) {
                ss = 0;
                ++nn;
            }
        };
        // ---- prologue: stages 0, 1, 2 ----
        stage_next();
        if (G > 1) stage_next();
        if (G > 2) stage_next();

        Frag4 fa0, fa1, fb0, fb1;
        if (abl & 16) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                fa0.lo[s] = fa0.hi[s] = fa1.lo[s] = fa1.hi[s] = u32x2{0, 0};
                fb0.lo[s] = fb0.hi[s] = fb1.lo[s] = fb1.hi[s] = u32x2{0, 0};
            }
        }
        if (FREE) {
            // Free-running form: ONE barrier per stage (after the counted DMA wait); inside a stage every wave pipelines its
            // own fragment reads against its MFMAs with counted lgkmcnt waits (LDS returns in order; the asm reads are
            // invisible to the compiler, hence the explicit waits and scheduling fences), and the two waves of a SIMD
            // interleave freely.
            auto fence = [&]() { __builtin_amdgcn_sched_barrier(0); };
            auto mma2 = [&](const Frag4& a, const Frag4& b0, f32x16& c0, const Frag4& b1, f32x16& c1, int s0, int s1) {
                if (abl & 4) return;
                auto op = [](const Frag4& f, int s) {
                    const u32x4 v = {f.lo[s][0], f.lo[s][1], f.hi[s][0], f.hi[s][1]};
                    return __builtin_bit_cast(bf16x8, v);
                };
#pragma unroll
                for (int s = s0; s < s1; ++s) {
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op(b0, s), op(a, s), c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op(b1, s), op(a, s), c1, 0, 0, 0);
                }
            };
            for (int g = 0; g < G; ++g) {
                const int slot = g & 3;
                // stages g and g+1 landed (stage g+2 may still be in flight)
                if (g + 2 < G && !(abl & 8)) {
                    if (abl & 96) {
                        wait_vmcnt<2>();
                    } else {
                        wait_vmcnt<4>();
                    }
                } else {
                    wait_vmcnt<0>();
                }
                slot_end();
                if (staged < G && !(abl & 8)) stage_next();  // stage g+3 -> ring slot of stage g-1 (every wave is past it)
                read_a(fa0, slot, 0);
                read_b(fb0, slot, 0);
                read_b(fb1, slot, 1);
                read_a(fa1, slot, 1);
                fence();
                asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");  // A(t0), B_c0, B_c1 (A(t1) may be in flight)
                fence();
                mma2(fa0, fb0, acc[0][0], fb1, acc[0][1], 0, 4);
                fence();
                read_done();  // A(t1)
                fence();
                read_a(fa0, slot, 2);
                fence();
                mma2(fa1, fb0, acc[1][0], fb1, acc[1][1], 0, 4);
                fence();
                read_done();  // A(t2)
                fence();
                mma2(fa0, fb0, acc[2][0], fb1, acc[2][1], 0, 4);
                fence();
            }
        } else {
            if (G > 2) {
                if (abl & 96) {
                    wait_vmcnt<2>();
                } else {
                    wait_vmcnt<4>();
                }
            } else {
                wait_vmcnt<0>();
            }
            slot_end();
            if (w >= 4) slot_end();  // channel blocks 2, 3 run one slot behind
            // A wave keeps at most 15 LDS reads in flight (lgkmcnt is 4 bits), so a READ slot costs about one LDS round trip
            // per 15 reads: the slots are paired big-with-big -- READ0 (24 reads) runs beside the other group's MFMA1 (16
            // MFMAs), READ1 (16 reads + the DMA) beside its MFMA0 (8).
            for (int g = 0; g < G; ++g) {
                const int slot = g & 3;
                read_a(fa0, slot, 0);
                read_b(fb0, slot, 0);
                read_b(fb1, slot, 1);
                read_done();
                slot_end();
                mma_a(fa0, fb0, fb1, acc[0][0], acc[0][1]);
                slot_end();
                read_a(fa1, slot, 1);
                read_a(fa0, slot, 2);
                if (staged < G && !(abl & 8)) {
                    stage_next();
                    if (abl & 96) {
                        wait_vmcnt<2>();
                    } else {
                        wait_vmcnt<4>();
                    }
                } else {
                    wait_vmcnt<0>();
                }
                read_done();
                slot_end();
                mma_b(fa1, fa0, fb0, fb1, acc[1][0], acc[1][1], acc[2][0], acc[2][1]);
                slot_end();
            }
            if (w < 4) slot_end();  // balance the barrier count of the two groups
        }
    }

    // ---- the split's slab tile: rows kk = tap * C_in + ci ----
    float* out = p.ws + (int64_t)split * p.Kk * p.c_out;
    const int hi = lane >> 5;
    const int ci = ci0 + wm * 32 + (lane & 31);
    if (ci0 + wm * 32 < p.c_in) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int64_t row = (int64_t)t * p.c_in + ci;
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int col = j0 + wn * 64 + jn * 32 + 8 * g4 + 4 * hi;
                    if (col < p.c_out) {
                        const f32x4 v = {acc[t][jn][4 * g4], acc[t][jn][4 * g4 + 1], acc[t][jn][4 * g4 + 2], acc[t][jn][4 * g4 + 3]};
                        *reinterpret_cast<f32x4*>(out + row * p.c_out + col) = v;
                    }
                }
            }
        }
    }
}

// fp32 Keras kernel (3, c_in, c_out) -> wf[co][k*c_in + ci] = W[k][ci][co];  wd[ci][j*c_out + co] = W[2-j][ci][co]
template <typename T>
__global__ void prep_weights_kernel(const float* w, int c_in, int c_out, T* wf, T* wd) {
    const int64_t total = 3LL * c_in * c_out;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    // i indexes the source: ((k*c_in)+ci)*c_out + co
    const int co = (int)(i % c_out);
    const int64_t r = i / c_out;
    const int ci = (int)(r % c_in);
    const int k = (int)(r / c_in);
    const T v = Elem<T>::from_f(w[i]);
    wf[(int64_t)co * 3 * c_in + (int64_t)k * c_in + ci] = v;
    wd[(int64_t)ci * 3 * c_out + (int64_t)(2 - k) * c_out + co] = v;
}

int g_gemm_kb = 128;  // K-slice bytes (tuning knob, vm_set_tuning("gemm_kb", 64 | 128))

}  // namespace vm

using namespace vm;

static int tiles(int64_t x, int t) { return (int)((x + t - 1) / t); }

extern "C" int64_t vm_conv_stat_rows(int64_t L) { return (L + BM - 1) / BM; }

int g_nt_ablate = 0;
int g_tn_xcd = 1;
int g_nt_blocks = 512;  // persistent NT grid (2 workgroups per CU on 256 CUs); vm_set_tuning("nt_blocks", n)

int g_nt_tepi = 1;      // bf16: 64-byte slices + storage-typed epilogue tile (39 KB LDS, 3 workgroups per CU)
int g_nt_blocks3 = 768;  // persistent grid of that variant
int g_nt_order = 1;
int g_nt_ring = 0;  // ring-pipelined LDS-DMA NT kernel (4 x 64-byte slices in flight); vm_set_tuning("nt_ring", 0 | 1)
int g_nt_glds = 1;  // direct-to-LDS NT kernel when the shape allows it; vm_set_tuning("nt_glds", 0 | 1)
// 256 x 256 phase-interleaved kernel for bf16 shapes with N % 256 == 0; vm_set_tuning("nt_p8", 0 | 1 | 2): 0 off, 1 every
// eligible shape, 2 (default) only K >= 1152 -- at cfg-A the block-4 forward (312 -> 275 us) and the block-3 dgrad (275 ->
// 245 us), where it clearly beats the 128^2 kernels; on the K = 384 forward of block 2 it only ties.  Step: 3.758 -> 3.708 ms
// (interleaved A/B, three repetitions each).  An earlier A/B, before the BN / block-1 / wgrad work, showed no step gain at
// all: the chip runs this workload against its power limit and a faster GEMM then slowed its neighbours down by as much.
// Ablations (block-3 dgrad, us): full 249, no epilogue 186, no DMA 147, neither 116 (= 78 % of the MFMA peak), barrier
// skeleton alone 47.
int g_nt_p8 = 2;
int g_nt_p8_blocks = 256;
int g_nt_p8_skew = 0;
int g_nt_korder = 0;  // K walk of the 128^2 LDS-DMA kernels: 0 (tap, chunk), 1 (chunk, tap)
int g_nt_p8_korder = 1;  // -0.5 % step (interleaved A/B): the same cache lines are re-read one K tile later instead of six
int g_nt_p8_phases = 2;  // MFMA clusters per K tile: 2 x 16 or 4 x 8; vm_set_tuning("nt_p8_phases", 2 | 4)
// conv_w4_kernel (one wave per SIMD, input-resident A): vm_set_tuning("nt_w4", 0 | 1 | 2): 0 off (default in round 1: it was
// finished after the last evidence run), 1 every eligible shape, 2 only K >= 1152.  Stand-alone form measured at 226 us on the
// block-4 forward against 265 us for conv_nt8_kernel (profiles/r01_conv_w4_probe.txt).
int g_nt_w4 = 0;

// conv_nt2_kernel (256 x 128 tiles, two workgroups per CU): vm_set_tuning("nt_n2", 0 | 1 | 2 | 3): 0 off, 1 forward, 2 dgrad, 3 both
int g_nt_n2 = 3;
int g_nt_n2_prio = 0;  // experiment: s_setprio(2) around the K loop of conv_nt2r_kernel
int g_nt_n2r = 1;  // prefer the input-resident form (conv_nt2r_kernel) where its tiling fits; vm_set_tuning("nt_n2r", 0 | 1)
template <int EPI>
static bool launch_n2(const NtArgs<bf16>& a, int64_t n_windows, hipStream_t stream) {
    if (!(g_nt_n2 & (EPI == EPI_DGRAD ? 2 : 1)) || a.N % n2::TN != 0 || a.a_c % 32 != 0 || a.Ktot != 3 * a.a_c) return false;
    // short windows (the 2-D variant runs L = 298 .. 37): a 256-row tile that is mostly padding loses to the 128-row kernels
    {
        const double u256 = (double)a.L / (256.0 * ((a.L + 255) / 256)), u128 = (double)a.L / (128.0 * ((a.L + 127) / 128));
        if (u256 + 0.10 < u128) return false;
    }
    NtArgs<bf16> b = a;
    b.skew = g_nt_n2_prio & 63;
    b.tilesN = a.N / n2::TN;
    const int t254 = (a.L + n2r::TROWS - 1) / n2r::TROWS;
    // input-resident variant: its statistics rows (two per 254-position tile) must be exactly the (L + 127) / 128 rows of
    // vm_conv_stat_rows -- checked for the inference launch as well so that a layer runs the same kernel in both modes
    if (g_nt_n2r && (EPI != EPI_FWD || 2 * t254 == (a.L + 127) / 128)) {
        b.tilesL = t254;
        const int64_t n_groups = n_windows * b.tilesL;
        const int64_t grid = n_groups * b.tilesN;
        if (grid >= (1LL << 31)) return false;
        // experiment (nt_n2_prio & 64): 8 KB of unused dynamic LDS -> one workgroup per CU instead of two
        hipLaunchKernelGGL((conv_nt2r_kernel<EPI>), dim3((unsigned)grid), dim3(256), (g_nt_n2_prio & 64) ? 8192 : 0, stream, b, n_groups);
        return true;
    }
    if (EPI == EPI_FWD_POOL || a.pool_e != nullptr) return false;  // the pooled epilogues are served by the input-resident kernel only
    b.tilesL = (a.L + n2::TM - 1) / n2::TM;
    const int64_t n_groups = n_windows * b.tilesL;
    const int64_t grid = n_groups * b.tilesN;
    if (grid >= (1LL << 31)) return false;
    if constexpr (EPI != EPI_FWD_POOL) hipLaunchKernelGGL((conv_nt2_kernel<EPI>), dim3((unsigned)grid), dim3(256), 0, stream, b, n_groups);
    return true;
}

template <int EPI>
static bool launch_w4(const NtArgs<bf16>& a, int64_t n_windows, hipStream_t stream) {
    const int tl254 = (a.L + w4::TROWS - 1) / w4::TROWS;
    if (!g_nt_w4 || (g_nt_w4 == 2 && a.Ktot < 1152) || a.N % 256 != 0 || a.a_c % 64 != 0 || a.Ktot != 3 * a.a_c || a.ablate != 0) return false;
    // forward statistics: one partial row per (tile, row half) must fill exactly the (L + 127) / 128 rows per window (checked
    // for the inference launch as well, so that a layer runs the same kernel -- the same summation order -- in both modes)
    if (EPI == EPI_FWD && 2 * tl254 != (a.L + 127) / 128) return false;
    const int64_t grid = n_windows * tl254 * (a.N / 256);
    if (grid >= (1LL << 31)) return false;
    NtArgs<bf16> b = a;
    b.tilesL = tl254;
    b.tilesN = a.N / 256;
    hipLaunchKernelGGL((conv_w4_kernel<EPI>), dim3((unsigned)grid), dim3(256), 0, stream, b);
    return true;
}

template <typename T, int EPI>
static bool launch_nt8(const NtArgs<T>&, int64_t, hipStream_t) { return false; }
template <int EPI>
static bool launch_nt8_bf16(const NtArgs<bf16>& a, int64_t n_windows, hipStream_t stream) {
    if (launch_n2<EPI>(a, n_windows, stream)) return true;
    if (launch_w4<EPI>(a, n_windows, stream)) return true;
    if (!g_nt_p8 || (g_nt_p8 == 2 && a.Ktot < 1152) || a.N % 256 != 0 || a.Ktot % 64 != 0 || a.Ktot < 192 || a.a_c % 8 != 0 ||
        n_windows * ((a.L + 255) / 256) * (a.N / 256) >= (1LL << 30)) return false;
    NtArgs<bf16> b = a;
    b.skew = g_nt_p8_skew;
    b.korder = (g_nt_p8_korder && a.a_c % 64 == 0 && a.Ktot == 3 * a.a_c) ? 1 : 0;
    b.tilesL = (a.L + 255) / 256;
    b.tilesN = a.N / 256;
    const int64_t n_groups = n_windows * b.tilesL;
    const int64_t total = n_groups * b.tilesN;
    const int64_t grid = total < g_nt_p8_blocks ? total : g_nt_p8_blocks;
    if (g_nt_p8_phases == 4) {
        hipLaunchKernelGGL((conv_nt8_kernel<EPI, 4>), dim3((unsigned)grid), dim3(512), 0, stream, b, (int)n_groups);
    } else {
        hipLaunchKernelGGL((conv_nt8_kernel<EPI, 2>), dim3((unsigned)grid), dim3(512), 0, stream, b, (int)n_groups);
    }
    return true;
}
template <>
bool launch_nt8<bf16, EPI_FWD>(const NtArgs<bf16>& a, int64_t n_windows, hipStream_t s) { return launch_nt8_bf16<EPI_FWD>(a, n_windows, s); }
template <>
bool launch_nt8<bf16, EPI_DGRAD>(const NtArgs<bf16>& a, int64_t n_windows, hipStream_t s) { return launch_nt8_bf16<EPI_DGRAD>(a, n_windows, s); }

template <typename T, int EPI>
static void launch_nt(const NtArgs<T>& a, int64_t n_groups, hipStream_t stream) {
    const int64_t grid = n_groups < g_nt_blocks ? n_groups : g_nt_blocks;
    const int64_t kbytes = (int64_t)a.Ktot * (int64_t)sizeof(T);
    if constexpr (sizeof(T) == 4) {
        if (a.split) {
            hipLaunchKernelGGL((conv_nt_kernel<T, EPI, 128, true>), dim3((unsigned)grid), dim3(256), 0, stream, a, n_groups);
            return;
        }
    }
    if (launch_nt8<T, EPI>(a, n_groups / a.tilesL, stream)) return;
    // measured at cfg-A: the 3-workgroup variant wins for the forward (epilogue-heavy, K = 384..1152: 0.94 -> 0.87 ms) and
    // loses for dgrad (K = 768..1536, light epilogue: 0.73 -> 0.81 ms), which keeps the 128-byte-slice kernel
    if (g_nt_tepi && EPI == EPI_FWD && a.ablate == 0 && kbytes % 64 == 0 && sizeof(T) == 2) {
        const int64_t g3 = n_groups * a.tilesN < g_nt_blocks3 ? n_groups * a.tilesN : g_nt_blocks3;
        hipLaunchKernelGGL((conv_nt_glds_kernel<T, EPI, 64, true>), dim3((unsigned)g3), dim3(256), 0, stream, a, n_groups);
        return;
    }
    if (g_nt_ring && a.ablate == 0 && kbytes % 64 == 0) {
        hipLaunchKernelGGL((conv_nt_ring_kernel<T, EPI>), dim3((unsigned)grid), dim3(256), 0, stream, a, n_groups);
        return;
    }
    if (g_nt_glds && a.ablate == 0 && kbytes % 128 == 0 && g_gemm_kb == 128) {
        hipLaunchKernelGGL((conv_nt_glds_kernel<T, EPI, 128>), dim3((unsigned)grid), dim3(256), 0, stream, a, n_groups);
        return;
    }
    if (g_nt_glds && a.ablate == 0 && kbytes % 64 == 0) {
        hipLaunchKernelGGL((conv_nt_glds_kernel<T, EPI, 64>), dim3((unsigned)grid), dim3(256), 0, stream, a, n_groups);
        return;
    }
    if (g_gemm_kb == 64) {
        hipLaunchKernelGGL((conv_nt_kernel<T, EPI, 64>), dim3((unsigned)grid), dim3(256), 0, stream, a, n_groups);
    } else {
        hipLaunchKernelGGL((conv_nt_kernel<T, EPI, 128>), dim3((unsigned)grid), dim3(256), 0, stream, a, n_groups);
    }
}

extern "C" int vm_conv_fwd(const void* in, const void* wf, const float* bias, int64_t n_windows, int64_t L, int c_in,
                           int c_out, int dtype, void* z, float* stat_sum, float* stat_sq, void* stream) {
    VM_REQUIRE(in && wf && bias && z, "vm_conv_fwd: null pointer");
    VM_REQUIRE(n_windows > 0 && L > 0 && c_in > 0 && c_out > 0, "vm_conv_fwd: bad sizes");
    VM_REQUIRE(c_in % 8 == 0 && c_out % 8 == 0, "vm_conv_fwd: channels must be multiples of 8 (got %d, %d)", c_in, c_out);
    VM_REQUIRE((stat_sum == nullptr) == (stat_sq == nullptr), "vm_conv_fwd: stat_sum/stat_sq must both be set or NULL");
    VM_REQUIRE((L + 2) * (int64_t)c_in < (1LL << 31) && 3LL * c_in * c_out < (1LL << 31), "vm_conv_fwd: window too large");
    VM_DISPATCH_DTYPE(dtype, {
        NtArgs<T> a;
        a.a = (const T*)in;
        a.bt = (const T*)wf;
        a.bias = bias;
        a.out = (T*)z;
        a.stat_sum = stat_sum;
        a.stat_sq = stat_sq;
        a.a_win_stride = (L + 2) * (int64_t)c_in;
        a.a_c = c_in;
        a.L = (int)L;
        a.N = c_out;
        a.Ktot = 3 * c_in;
        a.tilesL = tiles(L, BM);
        a.tilesN = tiles(c_out, BN);
        a.ablate = g_nt_ablate;
        a.order = g_nt_order;
        a.skew = 0;
        a.korder = g_nt_korder && (c_in * (int)sizeof(T)) % 128 == 0;
        a.split = dtype == VM_F32S;
        launch_nt<T, EPI_FWD>(a, n_windows * a.tilesL, (hipStream_t)stream);
    });
    return check_launch("vm_conv_fwd");
}

extern "C" int vm_conv_dgrad(const void* du, const void* wd, int64_t n_windows, int64_t L, int c_in, int c_out, int dtype,
                             void* dx, void* stream) {
    VM_REQUIRE(du && wd && dx, "vm_conv_dgrad: null pointer");
    VM_REQUIRE(n_windows > 0 && L > 0, "vm_conv_dgrad: bad sizes");
    VM_REQUIRE(c_in % 8 == 0 && c_out % 8 == 0, "vm_conv_dgrad: channels must be multiples of 8");
    VM_REQUIRE((L + 2) * (int64_t)c_out < (1LL << 31) && 3LL * c_in * c_out < (1LL << 31), "vm_conv_dgrad: window too large");
    VM_DISPATCH_DTYPE(dtype, {
        NtArgs<T> a;
        a.a = (const T*)du;
        a.bt = (const T*)wd;
        a.bias = nullptr;
        a.out = (T*)dx;
        a.stat_sum = nullptr;
        a.stat_sq = nullptr;
        a.a_win_stride = (L + 2) * (int64_t)c_out;
        a.a_c = c_out;
        a.L = (int)L;
        a.N = c_in;
        a.Ktot = 3 * c_out;
        a.tilesL = tiles(L, BM);
        a.tilesN = tiles(c_in, BN);
        a.ablate = g_nt_ablate;
        a.order = g_nt_order;
        a.skew = 0;
        a.korder = g_nt_korder && (c_out * (int)sizeof(T)) % 128 == 0;
        a.split = dtype == VM_F32S;
        launch_nt<T, EPI_DGRAD>(a, n_windows * a.tilesL, (hipStream_t)stream);
    });
    return check_launch("vm_conv_dgrad");
}

// ---- training forward that also emits the pool-window extreme (conv_nt2r_kernel only) ----
static bool fwd_e_shape(int64_t n_windows, int64_t L, int c_in, int c_out, int dtype) {
    if (dtype != VM_BF16 || !(g_nt_n2 & 1) || !g_nt_n2r || c_out % n2::TN != 0 || c_in % 32 != 0 || L < 2 || (L & 1) || n_windows <= 0) return false;
    const double u256 = (double)L / (256.0 * ((L + 255) / 256)), u128 = (double)L / (128.0 * ((L + 127) / 128));
    if (u256 + 0.10 < u128) return false;
    const int64_t t254 = (L + n2r::TROWS - 1) / n2r::TROWS;
    if (2 * t254 != (L + 127) / 128) return false;  // the statistics rows of vm_conv_stat_rows must be the kernel's two per tile
    return n_windows * t254 * (c_out / n2::TN) < (1LL << 31);
}

extern "C" int vm_conv_fwd_e_supported(int64_t n_windows, int64_t L, int c_in, int c_out, int dtype) {
    return fwd_e_shape(n_windows, L, c_in, c_out, dtype) ? 1 : 0;
}

extern "C" int vm_conv_fwd_e(const void* in, const void* wf, const float* bias, const float* gamma, int64_t n_windows, int64_t L,
                             int c_in, int c_out, int dtype, void* z, float* stat_sum, float* stat_sq, void* e, void* stream) {
    VM_REQUIRE(in && wf && bias && gamma && z && stat_sum && stat_sq && e, "vm_conv_fwd_e: null pointer");
    VM_REQUIRE(n_windows > 0 && L > 0, "vm_conv_fwd_e: bad sizes");
    VM_REQUIRE((L + 2) * (int64_t)c_in < (1LL << 31) && 3LL * c_in * c_out < (1LL << 31), "vm_conv_fwd_e: window too large");
    if (!fwd_e_shape(n_windows, L, c_in, c_out, dtype)) {
        set_error("vm_conv_fwd_e: shape/dtype/tuning not served by the 256 x 128 input-resident kernel (ask vm_conv_fwd_e_supported)");
        return VM_ERR_UNSUPPORTED;
    }
    NtArgs<bf16> a;
    a.a = (const bf16*)in;
    a.bt = (const bf16*)wf;
    a.bias = bias;
    a.out = (bf16*)z;
    a.stat_sum = stat_sum;
    a.stat_sq = stat_sq;
    a.a_win_stride = (L + 2) * (int64_t)c_in;
    a.a_c = c_in;
    a.L = (int)L;
    a.N = c_out;
    a.Ktot = 3 * c_in;
    a.tilesL = tiles(L, BM);
    a.tilesN = tiles(c_out, BN);
    a.ablate = 0;
    a.order = g_nt_order;
    a.skew = 0;
    a.korder = 0;
    a.aff_scale = gamma;
    a.pool_e = (bf16*)e;
    if (!launch_n2<EPI_FWD>(a, n_windows, (hipStream_t)stream)) {
        set_error("vm_conv_fwd_e: launch refused");
        return VM_ERR_UNSUPPORTED;
    }
    return check_launch("vm_conv_fwd_e");
}

// ---- inference forward with BatchNorm affine + MaxPool1D(2) in the epilogue (conv_nt2r_kernel only) ----
static bool fwd_pool_shape(int64_t n_windows, int64_t L, int c_in, int c_out, int dtype) {
    if (dtype != VM_BF16 || !(g_nt_n2 & 1) || !g_nt_n2r || c_out % n2::TN != 0 || c_in % 32 != 0 || L < 2 || (L & 1) || n_windows <= 0) return false;
    const double u256 = (double)L / (256.0 * ((L + 255) / 256)), u128 = (double)L / (128.0 * ((L + 127) / 128));
    if (u256 + 0.10 < u128) return false;
    const int64_t t254 = (L + n2r::TROWS - 1) / n2r::TROWS;
    return n_windows * t254 * (c_out / n2::TN) < (1LL << 31);
}

extern "C" int vm_conv_fwd_pool_supported(int64_t n_windows, int64_t L, int c_in, int c_out, int dtype) {
    return fwd_pool_shape(n_windows, L, c_in, c_out, dtype) ? 1 : 0;
}

extern "C" int vm_conv_fwd_pool(const void* in, const void* wf, const float* bias, const float* scale, const float* shift,
                                int64_t n_windows, int64_t L, int c_in, int c_out, int dtype, void* act, void* stream) {
    VM_REQUIRE(in && wf && bias && scale && shift && act, "vm_conv_fwd_pool: null pointer");
    VM_REQUIRE(n_windows > 0 && L > 0, "vm_conv_fwd_pool: bad sizes");
    VM_REQUIRE((L + 2) * (int64_t)c_in < (1LL << 31) && 3LL * c_in * c_out < (1LL << 31), "vm_conv_fwd_pool: window too large");
    if (!fwd_pool_shape(n_windows, L, c_in, c_out, dtype)) {
        set_error("vm_conv_fwd_pool: shape/dtype/tuning not served by the 256 x 128 input-resident kernel (ask vm_conv_fwd_pool_supported)");
        return VM_ERR_UNSUPPORTED;
    }
    NtArgs<bf16> a;
    a.a = (const bf16*)in;
    a.bt = (const bf16*)wf;
    a.bias = bias;
    a.out = (bf16*)act;
    a.stat_sum = nullptr;
    a.stat_sq = nullptr;
    a.a_win_stride = (L + 2) * (int64_t)c_in;
    a.a_c = c_in;
    a.L = (int)L;
    a.N = c_out;
    a.Ktot = 3 * c_in;
    a.tilesL = tiles(L, BM);
    a.tilesN = tiles(c_out, BN);
    a.ablate = 0;
    a.order = g_nt_order;
    a.skew = 0;
    a.korder = 0;
    a.aff_scale = scale;
    a.aff_shift = shift;
    if (!launch_n2<EPI_FWD_POOL>(a, n_windows, (hipStream_t)stream)) {
        set_error("vm_conv_fwd_pool: launch refused");
        return VM_ERR_UNSUPPORTED;
    }
    return check_launch("vm_conv_fwd_pool");
}

// ---- dgrad with the BatchNorm-backward partial sums of the layer below fused into its epilogue (conv_nt2r_kernel only) ----
static bool bnred_shape(int64_t n_windows, int64_t L, int c_in, int c_out, int dtype) {
    if (dtype != VM_BF16 || !(g_nt_n2 & 2) || !g_nt_n2r || c_in % n2::TN != 0 || c_out % 32 != 0 || L <= 0 || n_windows <= 0) return false;
    const double u256 = (double)L / (256.0 * ((L + 255) / 256)), u128 = (double)L / (128.0 * ((L + 127) / 128));
    if (u256 + 0.10 < u128) return false;
    const int64_t t254 = (L + n2r::TROWS - 1) / n2r::TROWS;
    return n_windows * t254 * (c_in / n2::TN) < (1LL << 31);
}

extern "C" int64_t vm_conv_dgrad_bnred_rows(int64_t L) { return 2 * ((L + n2r::TROWS - 1) / n2r::TROWS); }

extern "C" int vm_conv_dgrad_bnred_supported(int64_t n_windows, int64_t L, int c_in, int c_out, int dtype) {
    return bnred_shape(n_windows, L, c_in, c_out, dtype) ? 1 : 0;
}

extern "C" int vm_conv_dgrad_bnred(const void* du, const void* wd, int64_t n_windows, int64_t L, int c_in, int c_out, int dtype,
                                   void* dx, const void* red_a, int red_a_padded, float* red_s0, float* red_s1, void* stream) {
    VM_REQUIRE(du && wd && dx && red_a && red_s0 && red_s1, "vm_conv_dgrad_bnred: null pointer");
    VM_REQUIRE(n_windows > 0 && L > 0, "vm_conv_dgrad_bnred: bad sizes");
    VM_REQUIRE((L + 2) * (int64_t)c_out < (1LL << 31) && 3LL * c_in * c_out < (1LL << 31), "vm_conv_dgrad_bnred: window too large");
    if (!bnred_shape(n_windows, L, c_in, c_out, dtype)) {
        set_error("vm_conv_dgrad_bnred: shape/dtype/tuning not served by the 256 x 128 input-resident kernel (ask vm_conv_dgrad_bnred_supported)");
        return VM_ERR_UNSUPPORTED;
    }
    NtArgs<bf16> a;
    a.a = (const bf16*)du;
    a.bt = (const bf16*)wd;
    a.bias = nullptr;
    a.out = (bf16*)dx;
    a.stat_sum = red_s0;
    a.stat_sq = red_s1;
    a.a_win_stride = (L + 2) * (int64_t)c_out;
    a.a_c = c_out;
    a.L = (int)L;
    a.N = c_in;
    a.Ktot = 3 * c_out;
    a.tilesL = tiles(L, BM);
    a.tilesN = tiles(c_in, BN);
    a.ablate = 0;
    a.order = g_nt_order;
    a.skew = 0;
    a.korder = 0;
    a.red_a = (const bf16*)red_a;
    a.red_a_win_stride = (L + (red_a_padded ? 2 : 0)) * (int64_t)c_in;
    a.red_a_row0 = red_a_padded ? 1 : 0;
    if (!launch_n2<EPI_DGRAD>(a, n_windows, (hipStream_t)stream)) {
        set_error("vm_conv_dgrad_bnred: launch refused");
        return VM_ERR_UNSUPPORTED;
    }
    return check_launch("vm_conv_dgrad_bnred");
}

int g_tn_tile = 256;  // wgrad output tile: 256 (8 waves, one workgroup per CU) or 128; vm_set_tuning("tn_tile", ..)

extern int g_tn_x;
int g_tn_x = 1;   // (2: free-running form) input-resident (3 taps x 128 ci) x 128 co wgrad kernel (bf16, channels % 32 == 0); vm_set_tuning("tn_x", 0 | 1)
static bool tn_x_shape(int c_in, int c_out) { return g_tn_x && c_in % 64 == 0 && c_out % 64 == 0; }
int g_tn_p8 = 1;  // LDS-DMA + transposing-read wgrad kernel (bf16, channels % 32 == 0); vm_set_tuning("tn_p8", 0 | 1)
template <typename T>
static void launch_tn8x(const TnArgs<T>&, int64_t, hipStream_t) {}
template <>
void launch_tn8x<bf16>(const TnArgs<bf16>& a, int64_t grid, hipStream_t stream) {
    if (g_tn_x == 2) {
        hipLaunchKernelGGL(conv_tn8x_kernel<true>, dim3((unsigned)grid), dim3(512), 0, stream, a);
    } else {
        hipLaunchKernelGGL(conv_tn8x_kernel<false>, dim3((unsigned)grid), dim3(512), 0, stream, a);
    }
}
template <typename T>
static bool launch_tn8(const TnArgs<T>&, int64_t, hipStream_t) { return false; }
template <>
bool launch_tn8<bf16>(const TnArgs<bf16>& a, int64_t grid, hipStream_t stream) {
    if (!g_tn_p8 || a.c_in % 32 != 0 || a.c_out % 32 != 0 || a.n_windows >= (1LL << 30)) return false;
    hipLaunchKernelGGL(conv_tn8_kernel, dim3((unsigned)grid), dim3(512), 0, stream, a);
    return true;
}

static bool tn_use_256(int c_in, int c_out) { return g_tn_tile == 256 && 3 * c_in >= 192 && c_out >= 192; }

// Split of the position reduction over windows.  All workgroups of a launch do the same amount of work
// (windows_per_split windows) and a fixed number of them is resident at a time (2 per CU for the 128-tile kernel, 1 per
// CU for the 256-tile kernel), so the launch takes rounds = ceil(tiles * splits / slots) rounds of windows_per_split
// windows each -- a launch of 3 rounds + 12 workgroups pays a whole 4th round -- plus the write + re-read of one fp32
// slab per split.  Pick the split that minimises   rounds * wps * t_window  +  splits * t_slab.
extern "C" int vm_conv_wgrad_splits(int64_t n_windows, int64_t L, int c_in, int c_out) {
    const bool big = tn_use_256(c_in, c_out);
    const bool xres = tn_x_shape(c_in, c_out);  // (the fp32 kernels then run with a split count tuned for the bf16 tiling)
    const int tile = big ? 256 : 128;
    const int64_t t = xres ? (int64_t)tiles(c_in, 128) * tiles(c_out, 128) : (int64_t)tiles(3 * c_in, tile) * tiles(c_out, tile);
    const int64_t slots = (big || xres) ? 256 : 512;
    const double t_window = xres ? 2.0 * 384 * 128 * (double)L / 5.0e12 : 2.0 * tile * tile * (double)L / (big ? 4.0e12 : 1.0e12);
    const double t_slab = 8.0 * 3.0 * c_in * c_out / 3.0e12;
    int64_t best_wps = 1;
    double best_cost = -1.0;
    for (int64_t wps = 1; wps <= n_windows; ++wps) {
        const int64_t splits = (n_windows + wps - 1) / wps;
        const int64_t rounds = (t * splits + slots - 1) / slots;
        const double cost = (double)(rounds * wps) * t_window + (double)splits * t_slab;
        if (best_cost < 0.0 || cost < best_cost) {
            best_cost = cost;
            best_wps = wps;
        }
    }
    return (int)((n_windows + best_wps - 1) / best_wps);
}

extern "C" int64_t vm_conv_wgrad_workspace_bytes(int64_t n_windows, int64_t L, int c_in, int c_out) {
    return (int64_t)vm_conv_wgrad_splits(n_windows, L, c_in, c_out) * 3 * c_in * c_out * (int64_t)sizeof(float) +
           slab_sum_part_bytes(3LL * c_in * c_out);
}

extern "C" int vm_conv_wgrad(const void* in, const void* du, int64_t n_windows, int64_t L, int c_in, int c_out, int dtype,
                             void* ws, float* grad_w, void* stream) {
    VM_REQUIRE(in && du && ws && grad_w, "vm_conv_wgrad: null pointer");
    VM_REQUIRE(n_windows > 0 && L > 0, "vm_conv_wgrad: bad sizes");
    VM_REQUIRE(c_in % 8 == 0 && c_out % 8 == 0, "vm_conv_wgrad: channels must be multiples of 8");
    const int splits = vm_conv_wgrad_splits(n_windows, L, c_in, c_out);
    VM_DISPATCH_DTYPE(dtype, {
        TnArgs<T> a;
        a.x = (const T*)in;
        a.du = (const T*)du;
        a.ws = (float*)ws;
        a.x_win_stride = (L + 2) * (int64_t)c_in;
        a.du_win_stride = (L + 2) * (int64_t)c_out;
        a.c_in = c_in;
        a.c_out = c_out;
        a.L = (int)L;
        a.Kk = 3 * c_in;
        const bool big = tn_use_256(c_in, c_out);
        const bool xres = sizeof(T) == 2 && tn_x_shape(c_in, c_out) && n_windows < (1LL << 30);
        a.tilesI = xres ? tiles(c_in, 128) : tiles(3 * c_in, big ? 256 : BM);
        a.tilesJ = xres ? tiles(c_out, 128) : tiles(c_out, big ? 256 : BN);
        a.splits = splits;
        a.xcd_remap = g_tn_xcd;
        a.ablate = g_nt_ablate;
        a.n_windows = n_windows;
        a.win_per_split = (n_windows + splits - 1) / splits;
        const int64_t grid = (int64_t)splits * a.tilesI * a.tilesJ;
        a.split = dtype == VM_F32S;
        bool split_done = false;
        if constexpr (sizeof(T) == 4) {
            if (a.split && big) {
                hipLaunchKernelGGL((conv_tn256_kernel<T, 128, true>), dim3((unsigned)grid), dim3(512), 0, (hipStream_t)stream, a);
                split_done = true;
            } else if (a.split) {
                hipLaunchKernelGGL((conv_tn_kernel<T, 128, true>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
                split_done = true;
            }
        }
        if (split_done) {
        } else if (xres) {
            launch_tn8x<T>(a, grid, (hipStream_t)stream);
        } else if (big && launch_tn8<T>(a, grid, (hipStream_t)stream)) {
        } else if (big) {
            hipLaunchKernelGGL((conv_tn256_kernel<T, 128>), dim3((unsigned)grid), dim3(512), 0, (hipStream_t)stream, a);
        } else if (g_gemm_kb == 64) {
            hipLaunchKernelGGL((conv_tn_kernel<T, 64>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
        } else {
            hipLaunchKernelGGL((conv_tn_kernel<T, 128>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
        }
    });
    int rc = check_launch("vm_conv_wgrad");
    if (rc) return rc;
    const int64_t n = 3LL * c_in * c_out;
    return slab_sum((const float*)ws, splits, n, grad_w, n, nullptr, (float*)ws + (int64_t)splits * n, (hipStream_t)stream);
}

extern "C" int vm_prep_conv_weights(const float* w, int c_in, int c_out, int dtype, void* wf, void* wd, void* stream) {
    VM_REQUIRE(w && wf && wd, "vm_prep_conv_weights: null pointer");
    const int64_t n = 3LL * c_in * c_out;
    VM_DISPATCH_DTYPE(dtype, {
        hipLaunchKernelGGL((prep_weights_kernel<T>), dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, w,
                           c_in, c_out, (T*)wf, (T*)wd);
    });
    return check_launch("vm_prep_conv_weights");
}

// All layers of an encoder in one launch (the optimizer step re-derives every copy; three 5 us launches back to back cost more
// than the copies themselves).  blockIdx.y = layer.
constexpr int PREP_MAX_LAYERS = 8;
struct PrepBatch {
    const float* w[PREP_MAX_LAYERS];
    void* wf[PREP_MAX_LAYERS];
    void* wd[PREP_MAX_LAYERS];
    int c_in[PREP_MAX_LAYERS], c_out[PREP_MAX_LAYERS];
};
template <typename T>
__global__ void prep_weights_batch_kernel(PrepBatch pb) {
    const int l = blockIdx.y;
    const int c_in = pb.c_in[l], c_out = pb.c_out[l];
    const int64_t total = 3LL * c_in * c_out;
    const float* w = pb.w[l];
    T* wf = (T*)pb.wf[l];
    T* wd = (T*)pb.wd[l];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % c_out);
        const int64_t r = i / c_out;
        const int ci = (int)(r % c_in);
        const int k = (int)(r / c_in);
        const T v = Elem<T>::from_f(w[i]);
        wf[(int64_t)co * 3 * c_in + (int64_t)k * c_in + ci] = v;
        wd[(int64_t)ci * 3 * c_out + (int64_t)(2 - k) * c_out + co] = v;
    }
}

extern "C" int vm_prep_conv_weights_batch(int n_layers, const float* const* w, const int* c_in, const int* c_out, int dtype,
                                          void* const* wf, void* const* wd, void* stream) {
    VM_REQUIRE(w && c_in && c_out && wf && wd, "vm_prep_conv_weights_batch: null pointer");
    VM_REQUIRE(n_layers > 0 && n_layers <= PREP_MAX_LAYERS, "vm_prep_conv_weights_batch: 1..%d layers per call (got %d)", PREP_MAX_LAYERS,
               n_layers);
    PrepBatch pb;
    int64_t most = 0;
    for (int l = 0; l < n_layers; ++l) {
        VM_REQUIRE(w[l] && wf[l] && wd[l] && c_in[l] > 0 && c_out[l] > 0, "vm_prep_conv_weights_batch: bad layer %d", l);
        pb.w[l] = w[l];
        pb.wf[l] = wf[l];
        pb.wd[l] = wd[l];
        pb.c_in[l] = c_in[l];
        pb.c_out[l] = c_out[l];
        const int64_t n = 3LL * c_in[l] * c_out[l];
        most = n > most ? n : most;
    }
    VM_DISPATCH_DTYPE(dtype, {
        hipLaunchKernelGGL((prep_weights_batch_kernel<T>), dim3((unsigned)cdiv(most, 256), (unsigned)n_layers), dim3(256), 0,
                           (hipStream_t)stream, pb);
    });
    return check_launch("vm_prep_conv_weights_batch");
}

// Tuning hook for A/B measurements (not part of the drop-in surface): returns 0 if the key/value is known.
extern "C" int vm_set_tuning(const char* key, int value) {
    if (key != nullptr && strcmp(key, "gemm_kb") == 0 && (value == 64 || value == 128)) {
        g_gemm_kb = value;
        return VM_OK;
    }
    if (key != nullptr && strcmp(key, "f1_fwd_blocks") == 0 && value > 0) {
        return f1_set_fwd_blocks(value);
    }
    if (key != nullptr && strcmp(key, "f1_blocks") == 0 && value > 0) {
        return f1_set_blocks(value);
    }
    if (key != nullptr && strcmp(key, "nt_tepi") == 0) {
        g_nt_tepi = value;
        return VM_OK;
    }
    if (key != nullptr && strcmp(key, "nt_blocks3") == 0 && value > 0) {
        g_nt_blocks3 = value;
        return VM_OK;
    }
    if (key != nullptr && strcmp(key, "nt_order") == 0) {
        g_nt_order = value;
        return VM_OK;
    }
    if (key != nullptr && strcmp(key, "nt_ring") == 0) {
        g_nt_ring = value;
        return VM_OK;
    }
    if (key != nullptr && strcmp(key, "nt_glds") == 0) {
        g_nt_glds = value;
        return VM_OK;
    }
    if (key != nullptr && strcmp(key, "tn_x") == 0) {
        g_tn_x = value;
        return VM_OK;
    }
    if (key != nullptr && strcmp(key, "tn_p8") == 0) {
        g_tn_p8 = value;
        return VM_OK;
    }
    if (key != nullptr && strcmp(key, "nt_p8") == 0) {
        g_nt_p8 = value;
        return VM_OK;
    }
    if (key != nullptr && strcmp(key, "nt_p8_phases") == 0 && (value == 2 || value == 4)) {
        g_nt_p8_phases = value;
        return VM_OK;
    }
    if (key != nullptr && strcmp(key, "nt_korder") == 0) {
        g_nt_korder = value;
        return VM_OK;
    }
    if (key != nullptr && strcmp(key, "nt_p8_korder") == 0) {
        g_nt_p8_korder = value;
        return VM_OK;
    }
    if (key != nullptr && strcmp(key, "nt_n2_prio") == 0) {
        g_nt_n2_prio = value;
        return VM_OK;
    }
    if (key != nullptr && strcmp(key, "nt_n2r") == 0 && value >= 0 && value <= 1) {
        g_nt_n2r = value;
        return VM_OK;
    }
    if (key != nullptr && strcmp(key, "nt_n2") == 0 && value >= 0 && value <= 3) {
        g_nt_n2 = value;
        return VM_OK;
    }
    if (key != nullptr && strcmp(key, "nt_w4") == 0 && value >= 0 && value <= 2) {
        g_nt_w4 = value;
        return VM_OK;
    }
    if (key != nullptr && strcmp(key, "nt_p8_skew") == 0 && value >= 0 && value <= 64) {
        g_nt_p8_skew = value;
        return VM_OK;
    }
    if (key != nullptr && strcmp(key, "nt_p8_blocks") == 0) {
        VM_REQUIRE(value >= 1 && value <= 4096, "vm_set_tuning: nt_p8_blocks out of range");
        g_nt_p8_blocks = value;
        return VM_OK;
    }
    if (key != nullptr && strcmp(key, "nt_ablate") == 0) {
#ifndef VM_ENABLE_ABLATION
        if (value != 0) {
            vm::set_error("vm_set_tuning: nt_ablate produces wrong results and is only available in builds with -DVM_ENABLE_ABLATION");
            return VM_ERR_UNSUPPORTED;
        }
#endif
        g_nt_ablate = value;
        return VM_OK;
    }
    if (key != nullptr && strcmp(key, "tn_tile") == 0 && (value == 128 || value == 256)) {
        g_tn_tile = value;
        return VM_OK;
    }
    if (key != nullptr && strcmp(key, "tn_xcd") == 0) {
        g_tn_xcd = value;
        return VM_OK;
    }
    if (key != nullptr && strcmp(key, "nt_blocks") == 0 && value > 0) {
        g_nt_blocks = value;
        return VM_OK;
    }
    vm::set_error("vm_set_tuning: unknown key/value");
    return VM_ERR_ARG;
}
