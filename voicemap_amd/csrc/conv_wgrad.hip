// Weight gradient of the k=3 convolutions (blocks 2-4 of voicemap/models.py:22-35) as a TN GEMM on the CDNA4 matrix cores:
//   dW[kk][co] = sum_(n,t) A[(n,t)][kk] * dU[(n,t+1)][co],   kk = tap * C_in + ci, reduction over every position of every window.
// Both operands are position-major in memory while an MFMA lane wants 8 consecutive POSITIONS of one channel.  Three kernels:
//   conv_tn8x_kernel   16-bit storage (bf16 / f16), channel counts % 64 == 0: operands staged UNtransposed by LDS-DMA and
//                      transposed on the read by ds_read_b64_tr_b16; a workgroup owns 128 input channels for all three taps
//                      (one staged ring of input rows serves them) -- what every cfg-A launch runs
//   conv_tn256_kernel  256 x 256 tile, register transposes (v_perm_b32): fp32 storage (and the split-bf16 arithmetic of VM_F32S)
//   conv_tn_kernel     128 x 128 tile, register transposes: any shape
// The reduction is split over windows into fp32 slabs that are summed in a fixed order (reduce.hip): no float atomics, run-to-run
// bit-identical gradients.  (conv_tn8_kernel, the 256 x 256 LDS-DMA form that re-fetched the input for every tap, and the
// free-running form of conv_tn8x_kernel were removed in round 3: history up to commit 7ccb023, measurements in DESIGN.md 4.2.)
#include "conv_common.hpp"

namespace vm {

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
    int64_t n_windows, win_per_split;
    int split = 0;  // fp32 storage only (dtype VM_F32S): split-bf16 products
    // vm_conv_wgrad_fold: the windows come in towers of tower_windows and no slab straddles two of them (slab s belongs to tower
    // s / splits_per_tower); the plain entry point has one "tower" of all windows
    int64_t tower_windows = 0;
    int splits_per_tower = 0;
    // conv_tn9_kernel only: a split is a range of 64-position STAGES of its tower's stream (window after window), not of whole
    // windows -- 128 windows over 21 splits are 6.1 windows each, not 7 (0 = whole windows, win_per_split)
    int64_t stages_per_split = 0;
};

// windows [w_begin, w_end) of split ``split``
template <typename T>
__device__ inline void split_windows(const TnArgs<T>& p, int split, int64_t& w_begin, int64_t& w_end) {
    const int tw = split / p.splits_per_tower, j = split - tw * p.splits_per_tower;
    const int64_t t0 = (int64_t)tw * p.tower_windows;
    int64_t t1 = t0 + p.tower_windows;
    if (t1 > p.n_windows) t1 = p.n_windows;
    w_begin = t0 + (int64_t)j * p.win_per_split;
    w_end = w_begin + p.win_per_split;
    if (w_end > t1) w_end = t1;
    if (w_begin > w_end) w_begin = w_end;
}

template <typename T, int PITCH, int SZ = (int)sizeof(T)> struct Transpose4;
template <typename T, int PITCH> struct Transpose4<T, PITCH, 2> {
    // 4 position rows of 8 16-bit values -> 8 columns of 4 (8 bytes each)
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
template <typename T, int PITCH> struct Transpose4<T, PITCH, 4> {
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

    int64_t w_begin, w_end;
    split_windows(p, split, w_begin, w_end);
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

    int64_t w_begin, w_end;
    split_windows(p, split, w_begin, w_end);
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

template <typename T>
__global__ __launch_bounds__(512) void conv_tn8x_kernel(TnArgs<T> p) {
    static_assert(sizeof(T) == 2, "16-bit storage types (bf16 / f16)");
    using V8 = typename Mfma<T>::Frag;
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

    int w_begin, w_end;
    {
        int64_t wb, we;
        split_windows(p, split, wb, we);
        w_begin = (int)wb;
        w_end = (int)we;
    }
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
        auto stage = [&](int slot, int n, int st) {
            const int t = st * 64 + drow;
            {
                int r = t < p.L + 1 ? t : p.L + 1;  // padded row of tap 0 at position t; rows past the halo are never used
                const char* src = x_base + n * p.x_win_stride * 2;
                char* dst = lds + slot * 8192 + w * 1024;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    int c0 = ci0 + j * 64;
                    c0 = c0 < p.c_in ? c0 : 0;
                    glds16(src + (unsigned)(r * p.c_in + c0 + dchunk) * 2u, dst + j * ABLK);
                }
            }
            {
                const int r = (t < p.L ? t : p.L) + 1;  // row L + 1 of the padded dU tensor is the zero halo
                const char* src = d_base + n * p.du_win_stride * 2;
                char* dst = lds + A_BYTES + slot * BSTAGE + w * 1024;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    int co0 = j0 + j * 64;
                    co0 = co0 < p.c_out ? co0 : 0;
                    glds16(src + (unsigned)(r * p.c_out + co0 + dchunk) * 2u, dst + j * BBLK);
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
            const uint32_t blk = lds0 + (wm >> 1) * ABLK;
            const uint32_t u = slot * 8192 + a_off[tap];
#pragma unroll
            for (int s = 0; s < 4; ++s)
                tr_pair(fa.lo[s], fa.hi[s], blk + ((u + s * 2048) & (ABLK - 1)), blk + ((u + s * 2048 + 512) & (ABLK - 1)));
        };
        auto read_b = [&](Frag4& fb, int slot, int jn) {
            const uint32_t a = lds0 + A_BYTES + slot * BSTAGE + wn * BBLK + (b_off ^ (jn * 64));
#pragma unroll
            for (int s = 0; s < 4; ++s) tr_pair(fb.lo[s], fb.hi[s], a + s * 2048, a + s * 2048 + 512);
        };
        // clusters of 8 (one tap) and 16 (two taps) MFMAs, k-steps interleaved over the accumulator tiles
        auto opf = [](const Frag4& f, int s) {
            const u32x4 v = {f.lo[s][0], f.lo[s][1], f.hi[s][0], f.hi[s][1]};
            return __builtin_bit_cast(V8, v);
        };
        auto mma_a = [&](const Frag4& a0, const Frag4& b0, const Frag4& b1, f32x16& c0, f32x16& c1) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                c0 = Mfma<T>::run(opf(b0, s), opf(a0, s), c0);
                c1 = Mfma<T>::run(opf(b1, s), opf(a0, s), c1);
            }
            __builtin_amdgcn_s_setprio(0);
        };
        auto mma_b = [&](const Frag4& a0, const Frag4& a1, const Frag4& b0, const Frag4& b1, f32x16& c00, f32x16& c01, f32x16& c10,
                         f32x16& c11) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                c00 = Mfma<T>::run(opf(b0, s), opf(a0, s), c00);
                c01 = Mfma<T>::run(opf(b1, s), opf(a0, s), c01);
                c10 = Mfma<T>::run(opf(b0, s), opf(a1, s), c10);
                c11 = Mfma<T>::run(opf(b1, s), opf(a1, s), c11);
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
        {
            if (G > 2) {
                wait_vmcnt<4>();
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
                if (staged < G) {
                    stage_next();
                    wait_vmcnt<4>();
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

// ------------------------------------------------------------------------------------------------
// conv_tn9_kernel (round 4): conv_tn8x_kernel's tile, ring and DMA geometry with a FREE-RUNNING K loop -- no READ / MFMA slots, one
// barrier per stage (64 positions) instead of four.  A wave's stream is one MFMA, two transposing LDS reads of the NEXT k-step, one
// MFMA, ...: the fragments of a k-step (16 positions: 3 tap fragments of x, 2 column fragments of dU = 10 ds_read_b64_tr_b16, 20
// registers) are double-buffered in registers, every read has five to six MFMAs (>= 160 clocks) to return and every wait is a counted
// lgkmcnt(6); the four DMA instructions of stage g + 3 go one per k-step behind the fourth MFMA.  The phase kernel kept the matrix
// pipe of a SIMD busy only while the OTHER wave's READ slot was shorter than this wave's MFMA slot (8 MFMAs beside 16 reads + 4 DMA
// + the vmcnt wait); here the two waves of a SIMD interleave instruction by instruction, as in conv_nt3_kernel.
//   * RAW: stage g + 2 is waited for (vmcnt(4): the four youngest loads are stage g + 3's) and published by the barrier in k-step 3
//     of stage g; stage g reads its own rows from k-step 0 on and rows 0, 1 of stage g + 1 (tap overflow) in k-step 3.
//   * WAR: the DMA of stage g + 3 overwrites the ring slot of stage g - 1.  Its first instruction is issued in k-step 0 of stage g,
//     i.e. behind the barrier of stage g - 1, which every wave enters with lgkmcnt(0) -- all its reads of that slot returned.
//   * The loop is ROTATED: its back edge sits right behind that barrier, where no LDS read is in flight (hipcc copies registers at
//     loop phis; a copy of a register that a read is still writing would move stale bytes -- tools/isa_lint_inflight.py).  The last
//     two MFMAs of a stage's k-step 3 are issued at the top of the next iteration, behind the ten reads of its k-step 0 (the first
//     iteration runs them on zero fragments).
//   * A fragment register is overwritten by a read no earlier than four MFMAs after the last MFMA that read it.
// ------------------------------------------------------------------------------------------------
// experiment builds (tools/build_variant.sh <name> -DVM_TN9_ABL=<bits> conv_wgrad.hip; results are wrong by design): 1 no in-loop DMA,
// 2 no fragment reads, 4 no per-stage barrier, 8 no MFMAs
#ifndef VM_TN9_ABL
#define VM_TN9_ABL 0
#endif
// PW (producer waves): a 768-thread workgroup whose waves 8..11 (one per SIMD: a workgroup's waves go to the SIMDs cyclically) issue ALL
// the LDS-DMA -- a DMA instruction holds a wave's in-order issue for 60-250 ticks, and the 152 registers of this kernel leave room for a
// third wave per SIMD.  Producer iteration g: the 8 instructions of stage g + 3 for the rows of compute waves 2 (w - 8), 2 (w - 8) + 1,
// vmcnt(8) (stage g + 2 landed), the stage barrier; the compute waves carry no vector-memory instruction at all.
template <typename T, bool PW>
__global__ __launch_bounds__(PW ? 768 : 512) void conv_tn9_kernel(TnArgs<T> p) {
    static_assert(sizeof(T) == 2, "16-bit storage types (bf16 / f16)");
    using V8 = typename Mfma<T>::Frag;
    using namespace t8x;
    __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;  // channel block (0..3), output-column half (0..1)

    int64_t b = blockIdx.x;
    {
        const int64_t NT = (int64_t)p.tilesI * p.tilesJ;
        const int64_t full = (int64_t)(p.splits / 8) * 8 * NT;
        if (p.xcd_remap && b < full) {
            const int64_t xcd = b & 7, local = b >> 3;
            b = ((local / NT) * 8 + xcd) * NT + local % NT;
        } else if (p.xcd_remap) {
            // the splits % 8 splits behind the whole groups of 8: left in launch order their co-tiles went round the eight XCDs and every
            // L2 fetched the split's operands for itself (cfg-A block 4: 4 of 20 splits, 3.5 x their bytes -- most of the kernel's
            // counted excess over its algorithmic traffic).  Each XCD takes a contiguous run of (split, tile) pairs instead: a split's
            // 12 tiles sit on two XCDs
            const int64_t rest = (int64_t)p.splits * NT - full, per = rest / 8, bb = b - full;
            if (bb < per * 8) b = full + (bb & 7) * per + (bb >> 3);
        }
    }
    const int tj = __builtin_amdgcn_readfirstlane((int)(b % p.tilesJ));
    b /= p.tilesJ;
    const int ti = __builtin_amdgcn_readfirstlane((int)(b % p.tilesI));
    const int split = __builtin_amdgcn_readfirstlane((int)(b / p.tilesI));
    const int ci0 = ti * 128, j0 = tj * 128;

    const int spw = (p.L + 2 + 63) / 64;  // stages per window (the last one holds the halo row L + 1)
    // G stages are computed, Gs >= G are staged; the stream starts at stage ss0 of window nn0
    int nn0, ss0 = 0, G, Gs;
    if (p.stages_per_split > 0) {
        // stages [g0, g1) of the tower's stream.  A split that ends INSIDE a window stages one stage more than it computes: taps 1, 2 of
        // its last two positions read the first two input rows of the next stage (at a window's end those positions are halo rows
        // whose gradient rows are zero, and nothing more is needed)
        const int tw = split / p.splits_per_tower, j = split - tw * p.splits_per_tower;
        const int64_t t0 = (int64_t)tw * p.tower_windows;
        int64_t t1 = t0 + p.tower_windows;
        if (t1 > p.n_windows) t1 = p.n_windows;
        const int64_t S = (t1 - t0) * spw;
        int64_t g0 = (int64_t)j * p.stages_per_split, g1 = g0 + p.stages_per_split;
        g0 = g0 < S ? g0 : S;
        g1 = g1 < S ? g1 : S;
        G = __builtin_amdgcn_readfirstlane((int)(g1 - g0));
        nn0 = __builtin_amdgcn_readfirstlane((int)(t0 + g0 / spw));
        ss0 = __builtin_amdgcn_readfirstlane((int)(g0 % spw));
        Gs = G + ((G > 0 && g1 % spw != 0) ? 1 : 0);
    } else {
        int64_t wb, we;
        split_windows(p, split, wb, we);
        nn0 = (int)wb;
        G = we > wb ? (int)(we - wb) * spw : 0;
        Gs = G;
    }

    f32x16 acc[3][2];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][j][e] = 0.f;

    if (G > 0) {
        // zero the A ring once: tap-overflow reads of never-staged rows must be finite
        for (int i = tid * 16; i < A_BYTES; i += (PW ? 768 : 512) * 16) *reinterpret_cast<u32x4*>(lds + i) = u32x4{0, 0, 0, 0};
        __syncthreads();

        // ---- DMA geometry (conv_tn8x_kernel): one instruction = 8 position rows x 128 bytes, halves swapped by row bit 1 ----
        const int dchunk = ((lane & 7) ^ (((lane >> 4) & 1) << 2)) * 8;
        const char* const x_base = reinterpret_cast<const char*>(p.x);
        const char* const d_base = reinterpret_cast<const char*>(p.du);
        auto piece_of = [&](int wr, int slot, int n, int st, int q) {  // rows 8 wr .. 8 wr + 7; q = 0, 1: the A blocks; 2, 3: the B blocks
            const int t = st * 64 + wr * 8 + (lane >> 3);
            const int j = q & 1;
            if (q < 2) {
                const int r = t < p.L + 1 ? t : p.L + 1;
                int c0 = ci0 + j * 64;
                c0 = c0 < p.c_in ? c0 : 0;
                glds16(x_base + n * p.x_win_stride * 2 + (unsigned)(r * p.c_in + c0 + dchunk) * 2u, lds + slot * 8192 + wr * 1024 + j * ABLK);
            } else {
                const int r = (t < p.L ? t : p.L) + 1;
                int co0 = j0 + j * 64;
                co0 = co0 < p.c_out ? co0 : 0;
                glds16(d_base + n * p.du_win_stride * 2 + (unsigned)(r * p.c_out + co0 + dchunk) * 2u,
                       lds + A_BYTES + slot * BSTAGE + wr * 1024 + j * BBLK);
            }
        };
        auto piece = [&](int slot, int n, int st, int q) { piece_of(w, slot, n, st, q); };
        int nn = nn0, ss = ss0, staged = 0;  // stream cursor of the stage to be staged next
        auto advance = [&]() {
            ++staged;
            if (++ss == spw) {
                ss = 0;
                ++nn;
            }
        };

        // ---- fragment geometry (conv_tn8x_kernel) ----
        const int li = lane & 15, lg = (lane >> 4) & 1, kh = lane >> 5;
        const int rowl = kh * 8 + (li >> 2);
        const int sub = lg * 32 + (li & 3) * 8;
        int a_off[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) a_off[t] = (rowl + t) * 128 + (((wm & 1) ^ (((rowl + t) >> 1) & 1)) * 64) + sub;
        const int b_off = rowl * 128 + ((((rowl >> 1) & 1)) * 64) + sub;
#if defined(__HIP_DEVICE_COMPILE__)
        const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)lds;
#else
        const uint32_t lds0 = 0;
#endif
        const uint32_t ablk = lds0 + (wm >> 1) * ABLK;
        const uint32_t bblk = lds0 + A_BYTES + wn * BBLK;

        // ---- prologue: stages 0, 1, 2 ----
        if constexpr (PW) {
            if (w >= 8) {
                const int w0 = 2 * (w - 8);
                auto stage_all = [&]() {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int q = 0; q < 4; ++q) piece_of(w0 + i, staged & 3, nn, ss, q);
                    advance();
                };
                for (int i = 0; i < 3 && staged < Gs; ++i) stage_all();
                if (Gs > 2) {
                    wait_vmcnt<8>();
                } else {
                    wait_vmcnt<0>();
                }
                __builtin_amdgcn_s_barrier();
                for (int g = 0; g < G; ++g) {
                    if (staged < Gs) {
                        stage_all();
                        wait_vmcnt<8>();
                    } else {
                        wait_vmcnt<0>();
                    }
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
                return;
            }
            __builtin_amdgcn_s_barrier();
        } else {
            for (int i = 0; i < 3 && staged < Gs; ++i) {
#pragma unroll
                for (int q = 0; q < 4; ++q) piece(staged & 3, nn, ss, q);
                advance();
            }
            if (Gs > 2) {
                wait_vmcnt<4>();
            } else {
                wait_vmcnt<0>();
            }
            __builtin_amdgcn_s_barrier();
        }

        struct Fr {
            u32x2 lo, hi;
        };
        Fr fa[2][3], fb[2][2];  // [register set][tap] / [register set][column half]; constant indices only
#pragma unroll
        for (int t = 0; t < 3; ++t) fa[1][t].lo = fa[1][t].hi = u32x2{0u, 0u};
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[1][j].lo = fb[1][j].hi = u32x2{0u, 0u};

#define VM_TR(DST, ADDR, OFF) \
    if constexpr (!(VM_TN9_ABL & 2)) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #OFF : "=v"(DST) : "v"(ADDR))
        // k-step S of the stage whose addresses are in aaddr / a3lo / a3hi / baddr -> register set SET
#define VM_RDA(SET, TAP, S)                                      \
    if constexpr ((S) == 0) {                                    \
        VM_TR(fa[SET][TAP].lo, aaddr[TAP], 0);                   \
        VM_TR(fa[SET][TAP].hi, aaddr[TAP], 512);                 \
    } else if constexpr ((S) == 1) {                             \
        VM_TR(fa[SET][TAP].lo, aaddr[TAP], 2048);                \
        VM_TR(fa[SET][TAP].hi, aaddr[TAP], 2560);                \
    } else if constexpr ((S) == 2) {                             \
        VM_TR(fa[SET][TAP].lo, aaddr[TAP], 4096);                \
        VM_TR(fa[SET][TAP].hi, aaddr[TAP], 4608);                \
    } else {                                                     \
        VM_TR(fa[SET][TAP].lo, a3lo[TAP], 0);                    \
        VM_TR(fa[SET][TAP].hi, a3hi[TAP], 0);                    \
    }
#define VM_RDB(SET, JN, S)                                       \
    if constexpr ((S) == 0) {                                    \
        VM_TR(fb[SET][JN].lo, baddr[JN], 0);                     \
        VM_TR(fb[SET][JN].hi, baddr[JN], 512);                   \
    } else if constexpr ((S) == 1) {                             \
        VM_TR(fb[SET][JN].lo, baddr[JN], 2048);                  \
        VM_TR(fb[SET][JN].hi, baddr[JN], 2560);                  \
    } else if constexpr ((S) == 2) {                             \
        VM_TR(fb[SET][JN].lo, baddr[JN], 4096);                  \
        VM_TR(fb[SET][JN].hi, baddr[JN], 4608);                  \
    } else {                                                     \
        VM_TR(fb[SET][JN].lo, baddr[JN], 6144);                  \
        VM_TR(fb[SET][JN].hi, baddr[JN], 6656);                  \
    }
#define VM_LGKM(N, F) \
    if constexpr (!(VM_TN9_ABL & 2)) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"((F).lo), "+v"((F).hi))
#define VM_FRAG(F) __builtin_bit_cast(V8, (u32x4{(F).lo[0], (F).lo[1], (F).hi[0], (F).hi[1]}))
#define VM_MMA(SET, TAP, JN)                                                                                                   \
    if constexpr (!(VM_TN9_ABL & 8)) acc[TAP][JN] = Mfma<T>::run(VM_FRAG(fb[SET][JN]), VM_FRAG(fa[SET][TAP]), acc[TAP][JN]);   \
    else acc[TAP][JN][0] += __builtin_bit_cast(float, fb[SET][JN].lo[0] ^ fa[SET][TAP].hi[1]);                                 \
    __builtin_amdgcn_sched_barrier(0)
        // one k-step out of set CUR while k-step SN of the same stage loads into set NXT; Q: the DMA piece of stage g + 3 it carries
#define VM_KSTEP(CUR, NXT, SN, Q)                     \
    VM_LGKM(6, fb[CUR][0]);                           \
    VM_LGKM(6, fa[CUR][0]);                           \
    VM_MMA(CUR, 0, 0);                                \
    VM_RDB(NXT, 0, SN);                               \
    VM_LGKM(6, fa[CUR][1]);                           \
    VM_MMA(CUR, 1, 0);                                \
    VM_RDA(NXT, 0, SN);                               \
    VM_LGKM(6, fa[CUR][2]);                           \
    VM_MMA(CUR, 2, 0);                                \
    VM_RDA(NXT, 1, SN);                               \
    VM_LGKM(6, fb[CUR][1]);                           \
    VM_MMA(CUR, 0, 1);                                \
    VM_RDA(NXT, 2, SN);                               \
    if (!PW && more && !(VM_TN9_ABL & 1)) piece(staged & 3, nn, ss, Q); \
    __builtin_amdgcn_sched_barrier(0);                \
    VM_MMA(CUR, 1, 1);                                \
    VM_RDB(NXT, 1, SN);                               \
    VM_MMA(CUR, 2, 1)

        for (int g = 0; g < G; ++g) {
            const int slot = g & 3;
            const bool more = staged < Gs;
            uint32_t aaddr[3], a3lo[3], a3hi[3], baddr[2];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const uint32_t u = slot * 8192 + a_off[t];
                aaddr[t] = ablk + u;
                a3lo[t] = ablk + ((u + 3 * 2048) & (ABLK - 1));
                a3hi[t] = ablk + ((u + 3 * 2048 + 512) & (ABLK - 1));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) baddr[j] = bblk + slot * BSTAGE + (b_off ^ (j * 64));
            // k-step 0 of this stage -> set 0 (last read in k-step 2 of the previous stage)
            VM_RDB(0, 0, 0);
            VM_RDA(0, 0, 0);
            VM_RDA(0, 1, 0);
            VM_RDA(0, 2, 0);
            VM_RDB(0, 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            // the last two MFMAs of the previous stage's k-step 3 (set 1; zero fragments in the first iteration)
            VM_MMA(1, 1, 1);
            VM_MMA(1, 2, 1);
            VM_KSTEP(0, 1, 1, 0);
            VM_KSTEP(1, 0, 2, 1);
            VM_KSTEP(0, 1, 3, 2);
            // k-step 3 (set 1), nothing to prefetch: the waits count down
            VM_LGKM(6, fb[1][0]);
            VM_LGKM(6, fa[1][0]);
            VM_MMA(1, 0, 0);
            VM_LGKM(4, fa[1][1]);
            VM_MMA(1, 1, 0);
            VM_LGKM(2, fa[1][2]);
            VM_MMA(1, 2, 0);
            VM_LGKM(0, fb[1][1]);
            VM_MMA(1, 0, 1);
            if constexpr (!PW) {
                if (more) {
                    if (!(VM_TN9_ABL & 1)) piece(staged & 3, nn, ss, 3);
                    advance();
                    if (!(VM_TN9_ABL & 1)) wait_vmcnt<4>();
                } else {
                    wait_vmcnt<0>();
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" ::: "memory");
            if (!(VM_TN9_ABL & 4)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        VM_MMA(1, 1, 1);
        VM_MMA(1, 2, 1);
#undef VM_KSTEP
#undef VM_MMA
#undef VM_FRAG
#undef VM_LGKM
#undef VM_RDB
#undef VM_RDA
#undef VM_TR
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

}  // namespace vm

using namespace vm;

static int tiles(int64_t x, int t) { return (int)((x + t - 1) / t); }

namespace vm {
int g_tn9 = 1;        // conv_tn9_kernel (free-running K loop; 2 = with producer waves) instead of conv_tn8x_kernel (READ / MFMA slots); vm_set_tuning("tn9", 0 | 1 | 2)
int g_tn9_stages = 1;  // conv_tn9_kernel's splits are ranges of 64-position stages, not of whole windows; vm_set_tuning("tn9_stages", 0 | 1)
int g_tn_x = 1;       // conv_tn8x_kernel for 16-bit storage with channel counts % 64 == 0; vm_set_tuning("tn_x", 0 | 1)
int g_tn_tile = 256;  // tile of the register-transposing kernels: 256 (8 waves, one workgroup per CU) or 128; vm_set_tuning("tn_tile", ..)
}  // namespace vm

static bool tn_x_shape(int c_in, int c_out) { return g_tn_x && c_in % 64 == 0 && c_out % 64 == 0; }
static bool tn_use_256(int c_in, int c_out) { return g_tn_tile == 256 && 3 * c_in >= 192 && c_out >= 192; }
static bool tn_stage_splits(int c_in, int c_out) { return tn_x_shape(c_in, c_out) && g_tn9 != 0 && g_tn9_stages != 0; }
static int64_t tn9_stages_per_window(int64_t L) { return (L + 2 + 63) / 64; }

// Split of the position reduction over windows.  All workgroups of a launch do the same amount of work
// (windows_per_split windows) and a fixed number of them is resident at a time (2 per CU for the 128-tile kernel, 1 per
// CU for the others), so the launch takes rounds = ceil(tiles * splits / slots) rounds of windows_per_split
// windows each -- a launch of 3 rounds + 12 workgroups pays a whole 4th round -- plus the write + re-read of one fp32
// slab per split.  Pick the split that minimises   rounds * wps * t_window  +  splits * t_slab.
// ``tower_windows`` < n_windows (vm_conv_wgrad_fold): the same search with every tower split separately, so that a slab never holds
// windows of two towers; returns the splits PER TOWER.
static int wgrad_splits_per_tower(int64_t n_windows, int64_t tower_windows, int64_t L, int c_in, int c_out) {
    const bool big = tn_use_256(c_in, c_out);
    const bool xres = tn_x_shape(c_in, c_out);  // (the fp32 kernels then run with a split count tuned for the 16-bit tiling)
    const int tile = big ? 256 : 128;
    const int64_t t = xres ? (int64_t)tiles(c_in, 128) * tiles(c_out, 128) : (int64_t)tiles(3 * c_in, tile) * tiles(c_out, tile);
    const int64_t slots = (big || xres) ? 256 : 512;
    const double t_window = xres ? 2.0 * 384 * 128 * (double)L / 5.0e12 : 2.0 * tile * tile * (double)L / (big ? 4.0e12 : 1.0e12);
    const double t_slab = 8.0 * 3.0 * c_in * c_out / 3.0e12;
    const int64_t towers = (n_windows + tower_windows - 1) / tower_windows;
    if (tn_stage_splits(c_in, c_out)) {
        // conv_tn9_kernel: a split is a range of stages (64 positions) of its tower's stream, so the work of a launch is balanced to one
        // stage, not to one window: cfg-A block 3 (128 windows per tower, 6 output tiles) runs 2 x 21 splits of 147 stages (252 of 256
        // workgroup slots, 6.1 windows each) where whole windows gave 2 x 19 of 168.  A split that ends inside a window stages one
        // stage more than it computes (the + 1).  (The fp32 kernels run whole windows under the same split count: ceil(128 / 21) = 7
        // windows fill 19 splits, the last two write zero slabs.)
        const int64_t spw = tn9_stages_per_window(L), S = tower_windows * spw;
        const double t_stage = t_window / (double)spw;
        int64_t best_spt = 1;
        double best = -1.0;
        for (int64_t spt = 1; spt <= S && spt <= 4096; ++spt) {
            const int64_t splits = towers * spt, sps = (S + spt - 1) / spt;
            const int64_t rounds = (t * splits + slots - 1) / slots;
            const double cost = (double)(rounds * (sps + 1)) * t_stage + (double)splits * t_slab;
            if (best < 0.0 || cost < best) {
                best = cost;
                best_spt = spt;
            }
        }
        return (int)best_spt;
    }
    int64_t best_wps = 1;
    double best_cost = -1.0;
    for (int64_t wps = 1; wps <= tower_windows; ++wps) {
        const int64_t splits = towers * ((tower_windows + wps - 1) / wps);
        const int64_t rounds = (t * splits + slots - 1) / slots;
        const double cost = (double)(rounds * wps) * t_window + (double)splits * t_slab;
        if (best_cost < 0.0 || cost < best_cost) {
            best_cost = cost;
            best_wps = wps;
        }
    }
    return (int)((tower_windows + best_wps - 1) / best_wps);
}

extern "C" int vm_conv_wgrad_splits(int64_t n_windows, int64_t L, int c_in, int c_out) {
    return wgrad_splits_per_tower(n_windows, n_windows, L, c_in, c_out);
}

extern "C" int64_t vm_conv_wgrad_workspace_bytes(int64_t n_windows, int64_t L, int c_in, int c_out) {
    return (int64_t)vm_conv_wgrad_splits(n_windows, L, c_in, c_out) * 3 * c_in * c_out * (int64_t)sizeof(float) +
           slab_sum_part_bytes(3LL * c_in * c_out);
}

// The split-K GEMM launch shared by vm_conv_wgrad and vm_conv_wgrad_fold: one fp32 slab (3 * c_in, c_out) per split into ws.
// Returns the number of slabs.
static int launch_wgrad(const void* in, const void* du, int64_t n_windows, int64_t tower_windows, int64_t L, int c_in, int c_out,
                        int dtype, void* ws, hipStream_t st) {
    const int spt = wgrad_splits_per_tower(n_windows, tower_windows, L, c_in, c_out);
    const int splits = spt * (int)((n_windows + tower_windows - 1) / tower_windows);
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
        a.xcd_remap = 1;
        a.n_windows = n_windows;
        a.tower_windows = tower_windows;
        a.splits_per_tower = spt;
        a.win_per_split = (tower_windows + spt - 1) / spt;
        if (xres && tn_stage_splits(c_in, c_out)) a.stages_per_split = (tower_windows * tn9_stages_per_window(L) + spt - 1) / spt;
        a.split = dtype == VM_F32S;
        const dim3 grid((unsigned)((int64_t)splits * a.tilesI * a.tilesJ));
        if constexpr (sizeof(T) == 4) {
            if (a.split && big) {
                hipLaunchKernelGGL((conv_tn256_kernel<T, 128, true>), grid, dim3(512), 0, st, a);
            } else if (a.split) {
                hipLaunchKernelGGL((conv_tn_kernel<T, 128, true>), grid, dim3(256), 0, st, a);
            } else if (big) {
                hipLaunchKernelGGL((conv_tn256_kernel<T, 128>), grid, dim3(512), 0, st, a);
            } else {
                hipLaunchKernelGGL((conv_tn_kernel<T, 128>), grid, dim3(256), 0, st, a);
            }
        } else {
            if (xres) {
                if (g_tn9 == 2) {
                    hipLaunchKernelGGL((conv_tn9_kernel<T, true>), grid, dim3(768), 0, st, a);
                } else if (g_tn9) {
                    hipLaunchKernelGGL((conv_tn9_kernel<T, false>), grid, dim3(512), 0, st, a);
                } else {
                    hipLaunchKernelGGL((conv_tn8x_kernel<T>), grid, dim3(512), 0, st, a);
                }
            } else if (big) {
                hipLaunchKernelGGL((conv_tn256_kernel<T, 128>), grid, dim3(512), 0, st, a);
            } else {
                hipLaunchKernelGGL((conv_tn_kernel<T, 128>), grid, dim3(256), 0, st, a);
            }
        }
    });
    return splits;
}

extern "C" int vm_conv_wgrad(const void* in, const void* du, int64_t n_windows, int64_t L, int c_in, int c_out, int dtype,
                             void* ws, float* grad_w, void* stream) {
    VM_REQUIRE(in && du && ws && grad_w, "vm_conv_wgrad: null pointer");
    VM_REQUIRE(n_windows > 0 && L > 0, "vm_conv_wgrad: bad sizes");
    VM_REQUIRE(c_in % 8 == 0 && c_out % 8 == 0, "vm_conv_wgrad: channels must be multiples of 8");
    const int splits = launch_wgrad(in, du, n_windows, n_windows, L, c_in, c_out, dtype, ws, (hipStream_t)stream);
    if (splits < 0) return splits;  // unknown dtype
    int rc = check_launch("vm_conv_wgrad");
    if (rc) return rc;
    const int64_t n = 3LL * c_in * c_out;
    return slab_sum((const float*)ws, splits, n, grad_w, n, nullptr, (float*)ws + (int64_t)splits * n, (hipStream_t)stream);
}

// ---- weight gradient of a layer that ran on the pool extreme e of the layer below with that layer's BatchNorm affine folded into
// its weights (vm_conv_fwd_fold).  Its true input is y = scale_t[ci] * e + shift_t[ci] inside the window and 0 in the padding (t =
// the tower of the window: BatchNorm statistics are per encoder call), so
//     dW[k][ci][co] = sum_t  scale_t[ci] * (sum_pos e[pos + k - 1][ci] du[pos][co])  +  shift_t[ci] * D_t[k][co]
// with D_t[k][co] = sum of du[pos][co] over the positions whose tap k lies inside the window (vm_du_tower_sums).  The first sum is
// the GEMM above run on e, one set of slabs per tower; this kernel adds the slabs of each tower in fp64 in a fixed order and applies
// the two per-channel factors ----
__global__ __launch_bounds__(256) void slab_fold_kernel(const float* __restrict__ ws, int towers, int spt, int c_in, int c_out,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        const float* __restrict__ dsum, float* __restrict__ out) {
    const int64_t nel = 3LL * c_in * c_out;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nel) return;
    const int co = (int)(i % c_out);
    const int64_t r = i / c_out;
    const int ci = (int)(r % c_in), k = (int)(r / c_in);
    double total = 0.0;
    for (int t = 0; t < towers; ++t) {
        double s0 = 0.0, s1 = 0.0;
        int j = 0;
        for (; j + 2 <= spt; j += 2) {
            s0 += (double)ws[(int64_t)(t * spt + j) * nel + i];
            s1 += (double)ws[(int64_t)(t * spt + j + 1) * nel + i];
        }
        if (j < spt) s0 += (double)ws[(int64_t)(t * spt + j) * nel + i];
        total += (double)scale[t * c_in + ci] * (s0 + s1) + (double)shift[t * c_in + ci] * (double)dsum[((int64_t)t * 3 + k) * c_out + co];
    }
    out[i] = (float)total;
}

extern "C" int64_t vm_conv_wgrad_fold_workspace_bytes(int64_t n_windows, int64_t windows_per_tower, int64_t L, int c_in, int c_out) {
    if (windows_per_tower <= 0 || n_windows <= 0) return 0;
    const int64_t splits = (int64_t)wgrad_splits_per_tower(n_windows, windows_per_tower, L, c_in, c_out) *
                           ((n_windows + windows_per_tower - 1) / windows_per_tower);
    return splits * 3 * c_in * c_out * (int64_t)sizeof(float) + 64;
}

extern "C" int vm_conv_wgrad_fold_finish(const void* ws, int64_t n_windows, int64_t windows_per_tower, int64_t L, int c_in, int c_out,
                                         const float* scale, const float* shift, const float* dsum, float* grad_w, void* stream) {
    VM_REQUIRE(ws && scale && shift && dsum && grad_w, "vm_conv_wgrad_fold_finish: null pointer");
    VM_REQUIRE(n_windows > 0 && L > 0 && windows_per_tower > 0 && n_windows % windows_per_tower == 0,
               "vm_conv_wgrad_fold_finish: n_windows must be a positive multiple of windows_per_tower");
    const int towers = (int)(n_windows / windows_per_tower);
    const int spt = wgrad_splits_per_tower(n_windows, windows_per_tower, L, c_in, c_out);
    const int64_t n = 3LL * c_in * c_out;
    hipLaunchKernelGGL(slab_fold_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)ws, towers, spt,
                       c_in, c_out, scale, shift, dsum, grad_w);
    return check_launch("vm_conv_wgrad_fold_finish");
}

extern "C" int vm_conv_wgrad_fold(const void* in_e, const void* du, int64_t n_windows, int64_t windows_per_tower, int64_t L, int c_in,
                                  int c_out, int dtype, const float* scale, const float* shift, const float* dsum, void* ws,
                                  float* grad_w, void* stream) {
    VM_REQUIRE(in_e && du && ws, "vm_conv_wgrad_fold: null pointer");
    VM_REQUIRE(dsum == nullptr || (scale && shift && grad_w), "vm_conv_wgrad_fold: scale / shift / grad_w go with dsum");
    VM_REQUIRE(n_windows > 0 && L > 0 && windows_per_tower > 0 && n_windows % windows_per_tower == 0,
               "vm_conv_wgrad_fold: n_windows must be a positive multiple of windows_per_tower");
    VM_REQUIRE(c_in % 8 == 0 && c_out % 8 == 0, "vm_conv_wgrad_fold: channels must be multiples of 8");
    const int splits = launch_wgrad(in_e, du, n_windows, windows_per_tower, L, c_in, c_out, dtype, ws, (hipStream_t)stream);
    if (splits < 0) return splits;  // unknown dtype
    int rc = check_launch("vm_conv_wgrad_fold");
    if (rc || dsum == nullptr) return rc;   // dsum NULL: the slabs only; vm_conv_wgrad_fold_finish later (once dsum exists)
    return vm_conv_wgrad_fold_finish(ws, n_windows, windows_per_tower, L, c_in, c_out, scale, shift, dsum, grad_w, stream);
}

// ---- the weights vm_conv_fwd_fold runs on, per tower t: wf[t][co][k * c_in + ci] = W[k][ci][co] * scale[t][ci] in the storage type,
// and the per-tap constants hb[t][k][co] = sum_ci W[k][ci][co] * shift[t][ci] (fp64, fixed order), k = 0..2, plus row 3 = bias[co] +
// hb[t][0][co] + hb[t][1][co] + hb[t][2][co]: the value the forward's accumulators start from (one vector load per lane instead of
// four).  Input: wt = the fp32 kernel in wf's layout (vm_prep_conv_weights_batch), so that reads and writes are row-contiguous.
// One wave = one output channel (its three tap rows); a lane moves 8 consecutive input channels at a time: one 16-byte piece of wf
// and / or of the fragment-order copy conv_nt3_kernel streams (vm_pack_nt_weights' layout, written here directly) ----
template <typename T>
__global__ __launch_bounds__(256) void fold_bn_weights_kernel(const float* __restrict__ wt, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, const float* __restrict__ bias, int c_in,
                                                              int c_out, T* __restrict__ wf, T* __restrict__ wfp, float* __restrict__ hb,
                                                              float* __restrict__ ctr_out) {
    const int lane = threadIdx.x & 63, co = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int t = blockIdx.y;
    if (co >= c_out) return;
    const float* sc = scale + (int64_t)t * c_in;
    const float* sh = shift + (int64_t)t * c_in;
    const int ppr = c_in / 8;  // 16-byte pieces per tap row
    const int nk = 3 * (c_in / 32);
    double acc[3] = {0.0, 0.0, 0.0};
    for (int idx = lane; idx < 3 * ppr; idx += 64) {
        const int k = idx / ppr, ci = (idx - k * ppr) * 8;
        const float* src = wt + ((int64_t)co * 3 + k) * c_in + ci;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc + ci), s1 = *reinterpret_cast<const f32x4*>(sc + ci + 4);
        const f32x4 h0 = *reinterpret_cast<const f32x4*>(sh + ci), h1 = *reinterpret_cast<const f32x4*>(sh + ci + 4);
        Vec16<T> o;
        double a = 0.0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o.set(e, v0[e] * s0[e]);
            o.set(4 + e, v1[e] * s1[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) a += (double)v0[e] * (double)h0[e];   // ascending input channel, as the scalar loop did per lane
#pragma unroll
        for (int e = 0; e < 4; ++e) a += (double)v1[e] * (double)h1[e];
        acc[0] += k == 0 ? a : 0.0;
        acc[1] += k == 1 ? a : 0.0;
        acc[2] += k == 2 ? a : 0.0;
        if (wf != nullptr) store16<T>(wf + (((int64_t)t * c_out + co) * 3 + k) * c_in + ci, o);
        if (wfp != nullptr) {
            // element (row co, column k * c_in + ci .. + 7) of tower t in fragment order (see vm_pack_nt_weights)
            const int chunk = ci >> 5, ks = (ci >> 4) & 1, kh = (ci >> 3) & 1;
            const int64_t piece = (((((int64_t)t * (c_out >> 6) + (co >> 6)) * nk + 3 * chunk + k) * 2 + ((co >> 5) & 1)) * 2 + ks) * 64 +
                                  kh * 32 + (co & 31);
            store16<T>(wfp + piece * 8, o);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[k] += __shfl_xor(acc[k], o, 64);
    if (lane == 0) {
        float* h = hb + (int64_t)t * 4 * c_out + co;
        const float f0 = (float)acc[0], f1 = (float)acc[1], f2 = (float)acc[2];
        h[0] = f0;
        h[c_out] = f1;
        h[2 * c_out] = f2;
        const float start = bias[co] + ((f0 + f1) + f2);   // the sum conv_nt2r_kernel used to form per lane
        if (ctr_out != nullptr) {
            // the centre of this channel's tile (vm_conv_fwd_fold `e_center`): the pedestal the accumulators start from, where it is
            // positive, rounded to the storage type so that relu's clip value -ctr and z = t + ctr are exact
            const float ctr = (float)(T)fmaxf(start, 0.f);
            ctr_out[(int64_t)t * c_out + co] = ctr;
            h[3 * c_out] = start - ctr;
        } else {
            h[3 * c_out] = start;
        }
    }
}

extern "C" int vm_fold_bn_weights(const float* wt, const float* scale, const float* shift, const float* bias, int towers, int c_in,
                                  int c_out, int dtype, void* wf_folded, void* wf_packed, float* hb, float* ctr_out, void* stream) {
    VM_REQUIRE(wt && scale && shift && bias && hb && (wf_folded || wf_packed), "vm_fold_bn_weights: null pointer");
    VM_REQUIRE(c_in > 0 && c_out > 0 && towers > 0 && towers < 65536, "vm_fold_bn_weights: bad sizes");
    VM_REQUIRE(c_in % 8 == 0, "vm_fold_bn_weights: c_in must be a multiple of 8 (got %d)", c_in);
    VM_REQUIRE(wf_packed == nullptr || (c_in % 32 == 0 && c_out % 128 == 0),
               "vm_fold_bn_weights: the fragment-order copy needs c_in %% 32 == 0 and c_out %% 128 == 0 (vm_pack_nt_weights_supported)");
    VM_REQUIRE(dtype == VM_BF16 || dtype == VM_F16, "vm_fold_bn_weights: 16-bit storage only (VM_BF16 / VM_F16), got dtype %d", dtype);
    VM_DISPATCH_16(dtype, {
        hipLaunchKernelGGL((fold_bn_weights_kernel<T>), dim3((unsigned)cdiv(c_out, 4), (unsigned)towers), dim3(256), 0,
                           (hipStream_t)stream, wt, scale, shift, bias, c_in, c_out, (T*)wf_folded, (T*)wf_packed, hb, ctr_out);
    });
    return check_launch("vm_fold_bn_weights");
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
    float* wt[PREP_MAX_LAYERS];  // optional: the fp32 kernel itself in wf's layout (the input of vm_fold_bn_weights)
    int c_in[PREP_MAX_LAYERS], c_out[PREP_MAX_LAYERS];
};
// One workgroup = one 32 (c_in) x 32 (c_out) tile of one tap of one layer (blockIdx.y): the source rows run along c_out, wf / wt rows
// along c_in, so the tile goes through LDS and both sides move whole 64..128-byte segments (the element-per-thread version wrote
// 2-byte scatters: 10..16 us at the end of every step for 1.2 M weights); wd keeps the source's orientation.
template <typename T>
__global__ __launch_bounds__(256) void prep_weights_batch_kernel(PrepBatch pb) {
    __shared__ float tile[32][33];
    const int l = blockIdx.y;
    const int c_in = pb.c_in[l], c_out = pb.c_out[l];
    const int ti = (c_in + 31) / 32, tj = (c_out + 31) / 32;
    if ((int)blockIdx.x >= 3 * ti * tj) return;
    const int k = blockIdx.x / (ti * tj), r = blockIdx.x - k * ti * tj;
    const int ci0 = (r / tj) * 32, co0 = (r % tj) * 32;
    const float* w = pb.w[l];
    T* wf = (T*)pb.wf[l];
    T* wd = (T*)pb.wd[l];
    float* wt = pb.wt[l];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ci = ci0 + ty + 8 * j, co = co0 + tx;
        float v = 0.f;
        if (ci < c_in && co < c_out) {
            v = w[((int64_t)k * c_in + ci) * c_out + co];
            wd[(int64_t)ci * 3 * c_out + (int64_t)(2 - k) * c_out + co] = Elem<T>::from_f(v);
        }
        tile[ty + 8 * j][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int co = co0 + ty + 8 * j, ci = ci0 + tx;
        if (ci < c_in && co < c_out) {
            const float v = tile[tx][ty + 8 * j];
            const int64_t o = (int64_t)co * 3 * c_in + (int64_t)k * c_in + ci;
            wf[o] = Elem<T>::from_f(v);
            if (wt != nullptr) wt[o] = v;
        }
    }
}

extern "C" int vm_prep_conv_weights_batch(int n_layers, const float* const* w, const int* c_in, const int* c_out, int dtype,
                                          void* const* wf, void* const* wd, float* const* wt, void* stream) {
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
        pb.wt[l] = wt != nullptr ? wt[l] : nullptr;
        pb.c_in[l] = c_in[l];
        pb.c_out[l] = c_out[l];
        const int64_t n = 3LL * ((c_in[l] + 31) / 32) * ((c_out[l] + 31) / 32);   // tiles
        most = n > most ? n : most;
    }
    VM_DISPATCH_DTYPE(dtype, {
        hipLaunchKernelGGL((prep_weights_batch_kernel<T>), dim3((unsigned)most, (unsigned)n_layers), dim3(256), 0,
                           (hipStream_t)stream, pb);
    });
    return check_launch("vm_prep_conv_weights_batch");
}
