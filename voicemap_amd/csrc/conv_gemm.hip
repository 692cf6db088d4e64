// Conv1D(k=3, SAME) blocks 2-4 of the voicemap encoder as implicit GEMMs on the CDNA4 matrix cores.
//
// Channels-last + one zero halo row per window turns the im2col matrix into a *view*: the A row of output
// position (n, t) is the 3*C_in contiguous elements starting at padded row t, so no im2col buffer exists.
//   forward : Z[(n,t)][co]  = relu( sum_kk A[(n,t)][kk] * Wf[co][kk] + b[co] )        NT GEMM, K = 3*C_in
//   dgrad   : dX[(n,t)][ci] =       sum_kk dU[(n,t)][kk] * Wd[ci][kk]                 NT GEMM, K = 3*C_out
//   wgrad   : dW[kk][co]    =       sum_(n,t) A[(n,t)][kk] * dU[(n,t+1)][co]          TN GEMM, K = positions
// One compute core serves all three and both storage types: v_mfma_f32_32x32x16_bf16 (8 bf16 per lane) or
// v_mfma_f32_32x32x2_f32 (exact fp32, 1 float per lane).  Workgroup = 4 waves (2x2), 128x128 output tile,
// 64-byte K slices staged through LDS (80-byte row pitch: conflict-free ds_read_b128 fragment reads),
// global->register prefetch of slice k+1 overlapped with the MFMAs of slice k, two LDS buffers, one barrier
// per slice.  Tiles never straddle windows (grid = window x t-tile x n-tile), so the halo is never crossed
// and every row of a tile belongs to one BatchNorm tower.
#include "common.hpp"

namespace vm {

constexpr int BM = 128, BN = 128;
constexpr int KBYTES = 64;    // bytes of K per slice and row
constexpr int PITCH = 80;     // LDS row pitch in bytes
constexpr int TILE_BYTES = BM * PITCH;

template <typename T> struct Mfma;
template <> struct Mfma<bf16> {
    static constexpr int KSTEPS = 2;  // 2 x (32x32x16) per 32-element slice
    using Frag = bf16x8;
    __device__ static inline Frag load(const char* row_ptr, int s, int kh) {
        return *reinterpret_cast<const Frag*>(row_ptr + (s * 2 + kh) * 16);
    }
    __device__ static inline f32x16 run(Frag a, Frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mfma<float> {
    static constexpr int KSTEPS = 8;  // 8 x (32x32x2) per 16-element slice
    using Frag = float;
    __device__ static inline Frag load(const char* row_ptr, int s, int kh) {
        return *reinterpret_cast<const float*>(row_ptr + (s * 2 + kh) * 4);
    }
    __device__ static inline f32x16 run(Frag a, Frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
};

// One 64-byte K slice: every wave multiplies its 64x64 sub-tile.  lds_a / lds_b: [128][PITCH] bytes.
template <typename T>
__device__ inline void mma_slice(const char* lds_a, const char* lds_b, int wm, int wn, int lane, f32x16 (&acc)[2][2]) {
    const int r = lane & 31, kh = lane >> 5;
    const char* pa0 = lds_a + (wm * 64 + r) * PITCH;
    const char* pa1 = pa0 + 32 * PITCH;
    const char* pb0 = lds_b + (wn * 64 + r) * PITCH;
    const char* pb1 = pb0 + 32 * PITCH;
#pragma unroll
    for (int s = 0; s < Mfma<T>::KSTEPS; ++s) {
        typename Mfma<T>::Frag a0 = Mfma<T>::load(pa0, s, kh), a1 = Mfma<T>::load(pa1, s, kh);
        typename Mfma<T>::Frag b0 = Mfma<T>::load(pb0, s, kh), b1 = Mfma<T>::load(pb1, s, kh);
        acc[0][0] = Mfma<T>::run(a0, b0, acc[0][0]);
        acc[0][1] = Mfma<T>::run(a0, b1, acc[0][1]);
        acc[1][0] = Mfma<T>::run(a1, b0, acc[1][0]);
        acc[1][1] = Mfma<T>::run(a1, b1, acc[1][1]);
    }
}

// Accumulator element -> tile coordinates (MFMA 32x32 C/D layout: col = lane&31,
// row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)).
__device__ inline int acc_row(int wm, int im, int reg, int lane) {
    return wm * 64 + im * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}
__device__ inline int acc_col(int wn, int in, int lane) { return wn * 64 + in * 32 + (lane & 31); }

enum { EPI_FWD = 0, EPI_DGRAD = 1 };

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
};

template <typename T, int EPI>
__global__ __launch_bounds__(256) void conv_nt_kernel(NtArgs<T> p) {
    constexpr int VEC = Elem<T>::kVec;
    constexpr int BK = KBYTES / (int)sizeof(T);
    __shared__ __attribute__((aligned(16))) char lds[2][2][TILE_BYTES];  // [buf][A|B]

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1;
    int64_t b = blockIdx.x;
    const int tn = (int)(b % p.tilesN);
    b /= p.tilesN;
    const int tl = (int)(b % p.tilesL);
    const int64_t n = b / p.tilesL;
    const int t0 = tl * BM, n0 = tn * BN;

    // staging assignment: 2 x 16-byte chunks of A and of B per thread
    const T* a_ptr[2];
    const T* b_ptr[2];
    int lds_off[2], kch[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = tid + i * 256, row = id >> 2, ch = id & 3;
        int t = t0 + row;
        t = t < p.L ? t : p.L - 1;
        int nn = n0 + row;
        nn = nn < p.N ? nn : p.N - 1;
        a_ptr[i] = p.a + n * p.a_win_stride + (int64_t)t * p.a_c + ch * VEC;
        b_ptr[i] = p.bt + (int64_t)nn * p.Ktot + ch * VEC;
        lds_off[i] = row * PITCH + ch * 16;
        kch[i] = ch * VEC;
    }
    const int nk = (p.Ktot + BK - 1) / BK;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    u32x4 ra[2], rb[2];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int kk = kt * BK + kch[i];
            if (kk < p.Ktot) {
                ra[i] = *reinterpret_cast<const u32x4*>(a_ptr[i] + (int64_t)kt * BK);
                rb[i] = *reinterpret_cast<const u32x4*>(b_ptr[i] + (int64_t)kt * BK);
            } else {
                ra[i] = u32x4{0, 0, 0, 0};
                rb[i] = u32x4{0, 0, 0, 0};
            }
        }
    };
    gload(0);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<u32x4*>(&lds[buf][0][lds_off[i]]) = ra[i];
            *reinterpret_cast<u32x4*>(&lds[buf][1][lds_off[i]]) = rb[i];
        }
        __syncthreads();
        if (kt + 1 < nk) gload(kt + 1);
        mma_slice<T>(lds[buf][0], lds[buf][1], wm, wn, lane, acc);
    }

    // ---- epilogue ----
    if (EPI == EPI_FWD) {
        float csum[2] = {0.f, 0.f}, csq[2] = {0.f, 0.f};
#pragma unroll
        for (int in = 0; in < 2; ++in) {
            const int col = n0 + acc_col(wn, in, lane);
            const bool cok = col < p.N;
            const float bias = cok ? p.bias[col] : 0.f;
#pragma unroll
            for (int im = 0; im < 2; ++im) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int t = t0 + acc_row(wm, im, r, lane);
                    if (cok && t < p.L) {
                        float v = acc[im][in][r] + bias;
                        v = v > 0.f ? v : 0.f;
                        const T tv = Elem<T>::from_f(v);
                        p.out[(n * p.L + t) * (int64_t)p.N + col] = tv;
                        const float vr = Elem<T>::to_f(tv);
                        csum[in] += vr;
                        csq[in] += vr * vr;
                    }
                }
            }
        }
        if (p.stat_sum != nullptr) {
            __syncthreads();  // all waves done with the K-loop LDS
            float* red = reinterpret_cast<float*>(&lds[0][0][0]);  // [2 (sum|sq)][2 (wm)][128]
#pragma unroll
            for (int in = 0; in < 2; ++in) {
                float s = csum[in] + __shfl_xor(csum[in], 32, 64);
                float q = csq[in] + __shfl_xor(csq[in], 32, 64);
                if (lane < 32) {
                    const int c = wn * 64 + in * 32 + lane;
                    red[(0 * 2 + wm) * 128 + c] = s;
                    red[(1 * 2 + wm) * 128 + c] = q;
                }
            }
            __syncthreads();
            if (tid < 128 && n0 + tid < p.N) {
                const int64_t row = n * p.tilesL + tl;
                p.stat_sum[row * p.N + n0 + tid] = red[0 * 128 + tid] + red[1 * 128 + tid];
                p.stat_sq[row * p.N + n0 + tid] = red[2 * 128 + tid] + red[3 * 128 + tid];
            }
        }
    } else {
#pragma unroll
        for (int in = 0; in < 2; ++in) {
            const int col = n0 + acc_col(wn, in, lane);
            if (col >= p.N) continue;
#pragma unroll
            for (int im = 0; im < 2; ++im) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int t = t0 + acc_row(wm, im, r, lane);
                    if (t < p.L) p.out[(n * p.L + t) * (int64_t)p.N + col] = Elem<T>::from_f(acc[im][in][r]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// wgrad: TN GEMM with a transposing stager.  Output tile 128 (kk) x 128 (co); reduction over the positions
// of windows [w_begin, w_end).  Each stage brings BK positions x 128 columns of both operands; a thread loads
// 4 consecutive positions x 16 bytes and writes them position-contiguous so the fragment reads are the same
// 16-byte K-contiguous reads as in the NT kernel.
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
    int64_t n_windows, win_per_split;
};

template <typename T> struct Transpose4;
template <> struct Transpose4<bf16> {
    // 4 position rows of 8 bf16 -> 8 columns of 4 bf16 (8 bytes each)
    __device__ static inline uint32_t half(const u32x4& v, int j) { return (v[j >> 1] >> ((j & 1) * 16)) & 0xffffu; }
    __device__ static inline void store(char* lds_tile, int col0, int pg, const u32x4 (&v)[4]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            u32x2 o;
            o[0] = half(v[0], j) | (half(v[1], j) << 16);
            o[1] = half(v[2], j) | (half(v[3], j) << 16);
            *reinterpret_cast<u32x2*>(lds_tile + (col0 + j) * PITCH + pg * 8) = o;
        }
    }
};
template <> struct Transpose4<float> {
    __device__ static inline void store(char* lds_tile, int col0, int pg, const u32x4 (&v)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u32x4 o = {v[0][j], v[1][j], v[2][j], v[3][j]};
            *reinterpret_cast<u32x4*>(lds_tile + (col0 + j) * PITCH + pg * 16) = o;
        }
    }
};

template <typename T>
__global__ __launch_bounds__(256) void conv_tn_kernel(TnArgs<T> p) {
    constexpr int VEC = Elem<T>::kVec;
    constexpr int BKP = KBYTES / (int)sizeof(T);  // positions per stage
    constexpr int PG = BKP / 4;                   // groups of 4 positions
    __shared__ __attribute__((aligned(16))) char lds[2][2][TILE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1;
    int64_t b = blockIdx.x;
    const int tj = (int)(b % p.tilesJ);
    b /= p.tilesJ;
    const int ti = (int)(b % p.tilesI);
    const int split = (int)(b / p.tilesI);
    const int i0 = ti * BM, j0 = tj * BN;

    // staging role: threads 0..127 stage X (rows of the output tile), 128..255 stage dU (columns)
    const bool is_x = tid < 128;
    const int item = tid & 127;
    const int pg = item % PG, cg = item / PG;
    const int col0 = cg * VEC;  // first of VEC tile columns handled by this thread
    const T* base;
    int64_t win_stride;
    int row_c;
    bool col_ok;
    if (is_x) {
        const int kk = i0 + col0;
        col_ok = kk < p.Kk;
        base = p.x + (col_ok ? kk : 0);
        win_stride = p.x_win_stride;
        row_c = p.c_in;
    } else {
        const int co = j0 + col0;
        col_ok = co < p.c_out;
        base = p.du + p.c_out + (col_ok ? co : 0);  // +1 halo row: dU row t lives at padded row t+1
        win_stride = p.du_win_stride;
        row_c = p.c_out;
    }
    char* my_tile0 = &lds[0][is_x ? 0 : 1][0];
    char* my_tile1 = &lds[1][is_x ? 0 : 1][0];

    const int64_t w_begin = (int64_t)split * p.win_per_split;
    int64_t w_end = w_begin + p.win_per_split;
    if (w_end > p.n_windows) w_end = p.n_windows;
    const int stages_per_win = (p.L + BKP - 1) / BKP;
    const int64_t n_stages = (w_end - w_begin) * stages_per_win;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    u32x4 rv[4];
    auto gload = [&](int64_t st) {
        const int64_t n = w_begin + st / stages_per_win;
        const int t0 = (int)(st % stages_per_win) * BKP + pg * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = t0 + r;
            if (col_ok && t < p.L) {
                rv[r] = *reinterpret_cast<const u32x4*>(base + n * win_stride + (int64_t)t * row_c);
            } else {
                rv[r] = u32x4{0, 0, 0, 0};
            }
        }
    };
    if (n_stages > 0) gload(0);
    for (int64_t st = 0; st < n_stages; ++st) {
        char* tile = (st & 1) ? my_tile1 : my_tile0;
        Transpose4<T>::store(tile, col0, pg, rv);
        __syncthreads();
        if (st + 1 < n_stages) gload(st + 1);
        mma_slice<T>(lds[st & 1][0], lds[st & 1][1], wm, wn, lane, acc);
    }

    float* out = p.ws + (int64_t)split * p.Kk * p.c_out;
#pragma unroll
    for (int in = 0; in < 2; ++in) {
        const int col = j0 + acc_col(wn, in, lane);
        if (col >= p.c_out) continue;
#pragma unroll
        for (int im = 0; im < 2; ++im) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + acc_row(wm, im, r, lane);
                if (row < p.Kk) out[(int64_t)row * p.c_out + col] = acc[im][in][r];
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

extern "C" int64_t vm_conv_stat_rows(int64_t L) { return (L + BM - 1) / BM; }

extern "C" int vm_conv_fwd(const void* in, const void* wf, const float* bias, int64_t n_windows, int64_t L, int c_in,
                           int c_out, int dtype, void* z, float* stat_sum, float* stat_sq, void* stream) {
    VM_REQUIRE(in && wf && bias && z, "vm_conv_fwd: null pointer");
    VM_REQUIRE(n_windows > 0 && L > 0 && c_in > 0 && c_out > 0, "vm_conv_fwd: bad sizes");
    VM_REQUIRE(c_in % 8 == 0 && c_out % 8 == 0, "vm_conv_fwd: channels must be multiples of 8 (got %d, %d)", c_in, c_out);
    VM_REQUIRE((stat_sum == nullptr) == (stat_sq == nullptr), "vm_conv_fwd: stat_sum/stat_sq must both be set or NULL");
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
        const int64_t grid = n_windows * a.tilesL * a.tilesN;
        VM_REQUIRE(grid < (1LL << 31), "vm_conv_fwd: grid too large");
        hipLaunchKernelGGL((conv_nt_kernel<T, EPI_FWD>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    });
    return check_launch("vm_conv_fwd");
}

extern "C" int vm_conv_dgrad(const void* du, const void* wd, int64_t n_windows, int64_t L, int c_in, int c_out, int dtype,
                             void* dx, void* stream) {
    VM_REQUIRE(du && wd && dx, "vm_conv_dgrad: null pointer");
    VM_REQUIRE(n_windows > 0 && L > 0, "vm_conv_dgrad: bad sizes");
    VM_REQUIRE(c_in % 8 == 0 && c_out % 8 == 0, "vm_conv_dgrad: channels must be multiples of 8");
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
        const int64_t grid = n_windows * a.tilesL * a.tilesN;
        VM_REQUIRE(grid < (1LL << 31), "vm_conv_dgrad: grid too large");
        hipLaunchKernelGGL((conv_nt_kernel<T, EPI_DGRAD>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    });
    return check_launch("vm_conv_dgrad");
}

extern "C" int vm_conv_wgrad_splits(int64_t n_windows, int64_t L, int c_in, int c_out) {
    const int64_t t = (int64_t)tiles(3 * c_in, BM) * tiles(c_out, BN);
    int64_t s = (1536 + t - 1) / t;
    if (s > n_windows) s = n_windows;
    if (s < 1) s = 1;
    const int64_t wps = (n_windows + s - 1) / s;
    return (int)((n_windows + wps - 1) / wps);
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
        a.tilesI = tiles(3 * c_in, BM);
        a.tilesJ = tiles(c_out, BN);
        a.splits = splits;
        a.n_windows = n_windows;
        a.win_per_split = (n_windows + splits - 1) / splits;
        const int64_t grid = (int64_t)splits * a.tilesI * a.tilesJ;
        hipLaunchKernelGGL((conv_tn_kernel<T>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
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
