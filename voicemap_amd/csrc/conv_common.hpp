// Pieces shared by the k=3 convolution GEMM kernels (conv_gemm.hip: forward / dgrad, conv_wgrad.hip: weight gradient).
#pragma once
#include <string.h>

#include <type_traits>

#include "common.hpp"

namespace vm {

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
template <> struct Mfma<f16> {
    static constexpr int KSTEP_BYTES = 32;  // one 32x32x16: 16 halves of K (same rate as the bf16 instruction)
    using Frag = f16x8;
    __device__ static inline Frag load(const char* row_ptr, int s, int kh) {
        return *reinterpret_cast<const Frag*>(row_ptr + (s * 2 + kh) * 16);
    }
    __device__ static inline f32x16 run(Frag a, Frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
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

// eight ones in a 16-bit storage type (the all-ones MFMA operand of the column-sum tricks)
template <typename T>
__device__ inline typename Mfma<T>::Frag ones16() {
    static_assert(sizeof(T) == 2, "16-bit storage types only");
    const uint32_t w = std::is_same<T, bf16>::value ? 0x3F803F80u : 0x3C003C00u;  // 1.0 as bf16 / f16, twice
    const u32x4 v = {w, w, w, w};
    return __builtin_bit_cast(typename Mfma<T>::Frag, v);
}

// v_mfma_f32_4x4x4 (16 independent 4 x 4 x 4 blocks of four lanes): D[lane 4b + j][reg i] = sum_k A[lane 4b + i][k] * B[lane 4b + j][k]
// (tools/probe/mfma4x4_probe.hip).  Operands: four 16-bit K values per lane, as the two registers of a ds_read_b64_tr_b16.
template <typename T> struct Mfma4;
template <> struct Mfma4<f16> {
    __device__ static inline f32x4 run(u32x2 a, u32x2 b, f32x4 c) {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        return __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(h4, a), __builtin_bit_cast(h4, b), c, 0, 0, 0);
    }
};
template <> struct Mfma4<bf16> {
    __device__ static inline f32x4 run(u32x2 a, u32x2 b, f32x4 c) {
        typedef short s4 __attribute__((ext_vector_type(4)));
        return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(s4, a), __builtin_bit_cast(s4, b), c, 0, 0, 0);
    }
};
template <typename T>
__device__ inline u32x2 ones4() {   // four ones in a 16-bit storage type
    static_assert(sizeof(T) == 2, "16-bit storage types only");
    const uint32_t w = std::is_same<T, bf16>::value ? 0x3F803F80u : 0x3C003C00u;
    return u32x2{w, w};
}

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
// EPI_FWD_FOLD (conv_nt2r_kernel only): EPI_FWD on the pool extreme of the layer below with that layer's BatchNorm affine folded in
enum { EPI_FWD = 0, EPI_DGRAD = 1, EPI_FWD_POOL = 2, EPI_FWD_FOLD = 3 };

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
    int pool_e_pad = 0;  // 1: pool_e is padded like an activation tensor, (n_windows, L / 2 + 2, N) with the data in rows 1 .. L / 2
    // with pool_e: the OTHER element of every position pair, unpadded (n_windows, L / 2, N), its sign bit set where the extreme is the
    // second element of the pair (z >= 0 after ReLU, so the bit is free) -- (pool_e, pool_o) then hold all of z and `out` is not written
    T* pool_o = nullptr;
    // EPI_FWD_FOLD = EPI_FWD with the BatchNorm affine of the layer below folded into the weights (vm_conv_fwd_fold): the
    // input is the pool extreme e of that layer, bt holds W * scale[ci], and fold_hb (4, N) the per-tap constants
    // hb[k][co] = sum_ci W[k][ci][co] * shift[ci] (rows 0..2) and row 3 = bias + hb[0] + hb[1] + hb[2], which the accumulators start
    // at; position 0 of a window has no tap 0 and position L - 1 no tap 2 (the SAME padding pads the BatchNorm OUTPUT with zeros), so
    // hb[0] / hb[2] come off there.
    const float* fold_hb = nullptr;
    // ... and (f16 storage, (pool_e, pool_o) output) fold_ctr (towers, N): the tile is computed and stored CENTRED, t = relu(z) - ctr[co]
    // with ctr >= 0 exactly representable in the storage type (row 3 of fold_hb then already has ctr taken off): what a 16-bit
    // value spends its significand on is the distance from the pedestal, not the pedestal (vm_fold_bn_weights `ctr_out`)
    const float* fold_ctr = nullptr;
    // ... per tower (BatchNorm statistics are per encoder call): windows [t * tower_windows, (t + 1) * tower_windows) use the weights at
    // bt + t * bt_tower_stride and the constants at fold_hb + t * 4 * N
    int64_t tower_windows = 0, bt_tower_stride = 0;
    // vm_conv_fwd_flat (the 128-row kernels' epilogues): the input is the concatenation of windows of flat_valid positions, each with
    // its two zero halo rows (flat_period = flat_valid + 2 rows per window), run as ONE window; tile row t belongs to window
    // t / flat_period, position t % flat_period, and rows with position >= flat_valid -- the halo positions, whose results are junk --
    // are neither stored nor counted; the others go to the UN-padded output row window * flat_valid + position.  0 = off.
    int flat_period = 0, flat_valid = 0;
    // conv_nt3_kernel: bt in MFMA fragment order (vm_pack_nt_weights; all towers).  NULL: conv_nt2r_kernel streams bt through LDS.
    const T* bt_packed = nullptr;
};

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

template <int N>
__device__ inline void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

}  // namespace vm
