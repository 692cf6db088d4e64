// Conv1D(k=3, SAME) blocks 2-4 of the voicemap encoder (voicemap/models.py:22-35) as implicit GEMMs on the CDNA4 matrix cores:
// forward and input gradient (the weight gradient lives in conv_wgrad.hip).
//
// Channels-last + one zero halo row per window turns the im2col matrix into a *view*: the A row of output
// position (n, t) is the 3*C_in contiguous elements starting at padded row t, so no im2col buffer exists.
//   forward : Z[(n,t)][co]  = relu( sum_kk A[(n,t)][kk] * Wf[co][kk] + b[co] )        NT GEMM, K = 3*C_in
//   dgrad   : dX[(n,t)][ci] =       sum_kk dU[(n,t)][kk] * Wd[ci][kk]                 NT GEMM, K = 3*C_out
// Three kernels, every one parity-tested against the oracle on every shape it serves (tests/test_gpu_kernels.py):
//   conv_nt2r_kernel     16-bit storage (bf16 / f16), C_K % 32 == 0, N % 128 == 0: 254 x 128 output tiles, two workgroups per CU,
//                        input-resident A operand by LDS-DMA -- what every cfg-A launch runs
//   conv_nt_glds_kernel  128 x 128 tiles staged by LDS-DMA: fp32 storage and the 16-bit shapes the first one does not take
//   conv_nt_kernel       128 x 128 tiles staged through registers: any shape (K tails, odd channel counts) and the split-bf16
//                        arithmetic of dtype VM_F32S
// The MFMAs are issued with the operands swapped (D = B.A^T) so that a lane's four consecutive accumulator registers are four
// consecutive *output columns*: the epilogue moves the tile through LDS with 16-byte writes and leaves the workgroup as whole
// 16-byte, fully coalesced row segments.  Tiles never straddle windows (grid = window x t-tile x n-tile), so the halo is never
// crossed and every row of a tile belongs to one BatchNorm tower.
// (The kernel variants of rounds 1-2 that lost their A/Bs -- conv_nt_ring, conv_nt8, conv_w4, conv_nt2, the storage-typed 128^2
// epilogue -- were removed in round 3; they are in the history up to commit 7ccb023 and their measurements in DESIGN.md 4.2-4.4.)
#include "conv_common.hpp"

// Timing experiments of conv_nt2r_kernel (WRONG results: parts of the kernel are switched off); compiled in only by
// tools/build_variant.sh -DVM_ABL=<bits>: 1 no in-loop weight DMA, 2 no in-loop input DMA, 4 no output stores, 8 no statistics.
#ifndef VM_ABL
#define VM_ABL 0
#endif

namespace vm {

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
            if (kt + 1 < nk) gload(kt + 1);
            if constexpr (SPLIT) {
                mma_slice_split<KB>(ta, tb, wm, wn, lane, acc);
            } else {
                mma_slice<T, KB>(ta, tb, wm, wn, lane, acc);
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
            int64_t orow = n * p.L + t;
            bool rok = t < p.L;
            if (p.flat_period) {   // vm_conv_fwd_flat: (window, position) of this flat row; halo positions are dropped
                const int wq = t / p.flat_period, loc = t - wq * p.flat_period;
                rok = rok && loc < p.flat_valid;
                orow = (int64_t)wq * p.flat_valid + loc;
            }
            if (cok && rok) {
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
                T* dst = p.out + orow * (int64_t)p.N + ncol;
                store16<T>(dst, o0);
                if (sizeof(T) == 4) store16<T>(dst + 4, o1);
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
        acc[0][0] = Mfma<T>::run(b0, a0, acc[0][0]);
        acc[0][1] = Mfma<T>::run(b1, a0, acc[0][1]);
        acc[1][0] = Mfma<T>::run(b0, a1, acc[1][0]);
        acc[1][1] = Mfma<T>::run(b1, a1, acc[1][1]);
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
        int64_t orow = n * p.L + t;
        bool rok = t < p.L;
        if (p.flat_period) {   // vm_conv_fwd_flat: (window, position) of this flat row; halo positions are dropped
            const int wq = t / p.flat_period, loc = t - wq * p.flat_period;
            rok = rok && loc < p.flat_valid;
            orow = (int64_t)wq * p.flat_valid + loc;
        }
        if (cok && rok) {
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
            T* dst = p.out + orow * (int64_t)p.N + ncol;
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

template <typename T, int EPI, int KB>
__global__ __launch_bounds__(256, 2) void conv_nt_glds_kernel(NtArgs<T> p, int64_t n_groups) {
    constexpr int BK = KB / (int)sizeof(T);
    constexpr int RPI = 1024 / KB;        // rows per wave-instruction (64 lanes x 16 bytes)
    constexpr int CPR = KB / 16;          // chunks per row
    constexpr int NI = BM / RPI / 4;      // instructions per wave per operand per slice
    constexpr int OPB = BM * KB;          // bytes of one operand tile
    __shared__ __attribute__((aligned(16))) char lds[glds_lds_bytes<KB>()];

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
    const bool xcd_order = (n_groups & 7) == 0;
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
            auto issue = [&](int kt) {  // K slice kt = bytes [kt * KB, kt * KB + KB) of an (im2col / weight) row
                char* bufbase = lds + (kt & 1) * 2 * OPB;
                const int64_t ko = (int64_t)kt * KB;
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
            nt_epilogue<T, EPI>(p, lds, acc, n, tl, t0, n0, tid, lane, w, wm, wn);
            __syncthreads();  // the fp32 tile is consumed before the next tile's DMA overwrites the buffers
        }
    }
}

// ------------------------------------------------------------------------------------------------
// NT GEMM for 16-bit storage (bf16 / f16): 256 (positions) x 128 (channels) MFMA tile, 4 waves (2 x 2, 128 x 64 each), TWO
// workgroups per CU.
//
// Why this shape (round 2).  At cfg-A an output tile sees only K = 384..1536, i.e. 12..48 K tiles of 32 and then an epilogue that
// has to push 64 KB through a store path that issues ~7..10 B/clk/CU: with one workgroup per CU (the 256 x 256 kernels of round 1)
// that epilogue and the cold start of the next tile were a quarter of the tile time with the matrix pipes idle, and making the K
// stream persistent inside a workgroup did not help (tools/probe/conv_w4p_probe: 364 us against 280 us).  The 128^2 kernels above do
// overlap epilogues with other workgroups' main loops but their 64 x 64 wave tiles need 1 KB of fragment reads per MFMA and 2 x 128
// rows of DMA per 4 MFMA-steps: LDS-bound outright.  Here a wave owns 128 x 64 (0.75 KB of fragment reads per MFMA, 128
// accumulator registers -> 256 registers per wave -> two waves per SIMD) and every four waves are their OWN workgroup with their
// own output tile and 72 KB of LDS: two of them share a CU, drift apart, and one's epilogue / cold start runs under the other's
// MFMAs.  Operand tiles are unpadded (the LDS-DMA destination is lane-linear), 16-byte chunk c of row R at c ^ ((R >> 2) & 3): the
// swizzle is applied to the per-lane SOURCE address of the LDS-DMA and again by the fragment reads.
// Epilogue: (bias + ReLU,) 16-bit values in registers, tile through LDS ([256][272 B]), whole-row 16-byte stores, forward
// statistics of the stored (rounded) values as two partial rows per tile (one per 128 positions: the layout of vm_conv_stat_rows).
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
template <typename T, int EPI>
__device__ inline void n2_load_bias(const NtArgs<T>& p, f32x4 (&b4)[2][4], int c0) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            b4[j][g] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (EPI != EPI_DGRAD) b4[j][g] = *reinterpret_cast<const f32x4*>(p.bias + c0 + 32 * j + 8 * g);
        }
}
// vm_conv_fwd_fold: tile row ``row`` is the first (which = 0) or last (which = 1) position of the window -- take the constant of the
// tap that falls into the padding off its accumulators (lane r <-> row 32 i + r of the wave's 128, registers <-> channels)
// WIDE (conv_nt3_kernel's 256 x 32 wave tile, round 6 experiment): acc[i][j] is row block i + 4 j of the tile's eight, all of them over
// the wave's ONE 32-channel block -- every wave holds every row, so every wave takes the constant off ITS channels (h[0] only).
template <typename T, bool WIDE = false>
__device__ inline void n2_fold_edge(const NtArgs<T>& p, f32x16 (&acc)[4][2], int row, int which, int wm, int r, int c0) {
    if (!WIDE && (row >> 7) != wm) return;  // wave-uniform
    const float* hb = p.fold_hb + (which ? 2 * p.N : 0) + c0;
    f32x4 h[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) h[j][g] = *reinterpret_cast<const f32x4*>(hb + (WIDE ? 0 : 32 * j) + 8 * g);
    const int rw = WIDE ? row : (row & 127);
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // every 32-row block with a lane mask: a runtime block index would put acc into scratch
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float m = (32 * (WIDE ? i + 4 * j : i) + r == rw) ? 1.f : 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] -= m * h[j][g][e];
        }
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
// The epilogue in two halves (so that different waves COULD run them: conv_nt4_kernel, the persistent compute-wave / drain-wave
// experiment of round 4 -- bit-identical, 5-16 % slower, removed; profiles/r04_nt4_wave_profile.txt): n2_tile_write moves a wave's accumulators into
// the 16-bit LDS tile, n2_tile_drain does everything that reads the tile (stores, pool pairs, statistics, the fused BatchNorm-backward
// sums).  ``bar``: the drain's synchronisation policy -- sync() = a barrier the draining waves NEED between two of its phases
// (only the fused sums have them), point() = a place where a barrier may be put for balance and nothing depends on it.
#if defined(VM_EXPERIMENT_PROFILE_EPI)  // experiment builds only: raw s_memtime stamps of the epilogue's phases (tools/probe/nt3_prof.py epi)
extern __device__ unsigned int g_prof[];
__device__ inline void vm_epi_mark(int k) {
    const long long t = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 8192) g_prof[((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + k] = (unsigned int)t;
}
#if VM_EXPERIMENT_PROFILE_EPI + 0 == 2   // =2: the stamps go to the dgrad's fused BatchNorm-backward sums instead (EPI=2 nt3_prof.py)
#define VM_EPI_MARK(k)
#define VM_RED_MARK(k) vm_epi_mark(k)
#else
#define VM_EPI_MARK(k) vm_epi_mark(k)
#define VM_RED_MARK(k)
#endif
#else
#define VM_EPI_MARK(k)
#define VM_RED_MARK(k)
#endif
struct N2DrainSync {   // the drain run by the four waves of a 256-thread workgroup that also computed the tile
    __device__ inline void sync() { __syncthreads(); }
    __device__ inline void point() {}
};

template <typename T, int EPI, bool WIDE = false>
__device__ inline void n2_tile_write(const NtArgs<T>& p, char* lds, const f32x16 (&acc)[4][2], int t0, int n0, int trows, int lane, int wm, int wn,
                                     const f32x4 (&negc)[2][4]) {
    using namespace n2;
    const int r = lane & 31, kh = lane >> 5;
    constexpr bool FWD = EPI == EPI_FWD || EPI == EPI_FWD_FOLD;
    const bool red = EPI == EPI_DGRAD && p.red_a != nullptr;
    const int valid = (p.L - t0) < trows ? (p.L - t0) : trows;  // MFMA-tile rows that are positions of the window
    // centred tile (fold_ctr): the accumulators hold z_pre - ctr (the start vector had ctr taken off), so ReLU is max(., -ctr), with
    // ONE rounding to the storage type
    // (negc = -ctr of this lane's 32 channels, loaded by n2_epilogue in front of its first barrier)
    constexpr bool CAN_CENTRE = EPI == EPI_FWD_FOLD && std::is_same<T, f16>::value;
    const bool ctrd = CAN_CENTRE && p.fold_ctr != nullptr;
    // ---- registers -> bf16 tile in LDS.  Forward: the bias is already in the accumulators (they were initialised with it) and
    // ReLU is applied to the PACKED bf16 pairs as a signed 16-bit max with 0 (a negative bf16 is a negative int16, -0.0 included;
    // rounding is monotone, so relu(round(x)) == round(relu(x))): 1 VALU instruction per element instead of 2.5 ----
    // (the run-time switch is hoisted into two straight-line copies: left inside, hipcc branched on it once per 4 values -- 64
    // branches per wave-tile, 4.8 k clocks of tile write against the dgrad form's 2.0 k; profiles/r05_nt3_epilogue_phases.txt)
    // The centred form takes its maximum with -ctr AFTER the conversion, as a packed half maximum: rounding is monotone and -ctr is a
    // half, so max(rn(x), -ctr) == rn(max(x, -ctr)) -- still one rounding, 2 operations per 4 values instead of 4 fp32 maxima (a VALU
    // operation costs 12-20 clocks here, beside a wave that issues MFMAs back to back: centred tile write 3.0 k -> 2.0 k clocks, the
    // un-centred form's).  (Zeroing the rows outside the window only in the 32-row blocks that have any -- a wave-uniform branch per
    // block instead of two selects per group -- was measured too: no faster, and the four copies of the loop cost the dgrad 1 %.)
    auto body = [&](auto ctrd_c) {
        constexpr bool CTRD = decltype(ctrd_c)::value;
        uint32_t nc16[2][4][2];   // -ctr of the lane's channels as packed halves (exact: ctr is a half)
        if constexpr (CTRD) {
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    nc16[j][g][0] = __builtin_bit_cast(uint32_t, h2{(_Float16)negc[j][g][0], (_Float16)negc[j][g][1]});
                    nc16[j][g][1] = __builtin_bit_cast(uint32_t, h2{(_Float16)negc[j][g][2], (_Float16)negc[j][g][3]});
                }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                // (WIDE: wn is the wave's 32-channel block 0..3, (i, j) its row block i + 4 j)
                const int mb = WIDE ? (i + 4 * j) * 32 : wm * 128 + i * 32;
                const int m = mb + r;
                const bool partial = mb + 32 > valid;  // wave-uniform: this 32-row block has rows outside the window
                const bool zero = (FWD || red) && partial && m >= valid;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nl = (WIDE ? wn * 32 : wn * 64 + j * 32) + 8 * g + 4 * kh;  // first of this lane's 4 consecutive channels
                    T o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = Elem<T>::from_f(acc[i][j][4 * g + e]);
                    u32x2 pk = *reinterpret_cast<const u32x2*>(o);
                    if constexpr (CTRD) {
                        uint32_t lo = pk[0], hi = pk[1];
                        asm("v_pk_max_f16 %0, %1, %2" : "=v"(lo) : "v"(lo), "v"(nc16[j][g][0]));
                        asm("v_pk_max_f16 %0, %1, %2" : "=v"(hi) : "v"(hi), "v"(nc16[j][g][1]));
                        pk[0] = lo;
                        pk[1] = hi;
                    } else if constexpr (EPI != EPI_DGRAD) {
                        uint32_t lo = pk[0], hi = pk[1];
                        asm("v_pk_max_i16 %0, %1, 0" : "=v"(lo) : "v"(lo));
                        asm("v_pk_max_i16 %0, %1, 0" : "=v"(hi) : "v"(hi));
                        pk[0] = lo;
                        pk[1] = hi;
                    }
                    pk[0] = zero ? 0u : pk[0];
                    pk[1] = zero ? 0u : pk[1];
                    *reinterpret_cast<u32x2*>(lds + m * TP + nl * 2) = pk;
                }
            }
        }
    };
    if constexpr (CAN_CENTRE) {
        if (ctrd) body(std::true_type{});
        else body(std::false_type{});
    } else {
        body(std::false_type{});
    }
}

template <typename T, int EPI, typename BarT>
__device__ inline void n2_tile_drain(const NtArgs<T>& p, char* lds, int64_t n, int tl, int t0, int n0, int trows, int tid, int lane, int w,
                                     BarT& bar) {
    using namespace n2;
    using V8 = typename Mfma<T>::Frag;  // eight 16-bit values
    const int kh = lane >> 5;
    constexpr bool FWD = EPI == EPI_FWD || EPI == EPI_FWD_FOLD;  // the training forward (statistics, optional pool extreme)
    const bool stats = FWD && p.stat_sum != nullptr && !(VM_ABL & 8);
    const bool red = EPI == EPI_DGRAD && p.red_a != nullptr;
    const int valid = (p.L - t0) < trows ? (p.L - t0) : trows;  // MFMA-tile rows that are positions of the window
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
        T* pbase = p.out + (n * (int64_t)(p.L / 2 + 2) + 1 + (t0 >> 1)) * (int64_t)p.N + n0 + c8 * 8;
        V8 r0[8], r1[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int q = rg + 16 * jj;
            r0[jj] = *reinterpret_cast<const V8*>(lds + (2 * q) * TP + c8 * 16);
            r1[jj] = *reinterpret_cast<const V8*>(lds + (2 * q + 1) * TP + c8 * 16);
        }
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int q = rg + 16 * jj;
            V8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float y0 = fmaf((float)r0[jj][e], sc[e], sh[e]), y1 = fmaf((float)r1[jj][e], sc[e], sh[e]);
                o[e] = (T)(y1 > y0 ? y1 : y0);
            }
            if (q < vq) *reinterpret_cast<V8*>(pbase + (int64_t)q * p.N) = o;
            bar.point();
        }
        return;
    }
    // ---- read-back: 8 rows per thread and half, ALL tile reads first, then the 8 whole-row stores back to back; interior tiles
    // take a predicate-free path ----
    T* obase = p.out + (n * p.L + t0) * (int64_t)p.N + n0 + c8 * 8;
    const bool interior = t0 + trows <= p.L;  // every valid MFMA row of the tile is a position of the window
    auto half = [&](int h, auto interior_c) {
        constexpr bool INTERIOR = decltype(interior_c)::value;
        u32x4 v[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) v[jj] = *reinterpret_cast<const u32x4*>(lds + (h * 128 + rg + 16 * jj) * TP + c8 * 16);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int row = h * 128 + rg + 16 * jj;
            // rows >= trows exist only in the last 16 rows of the tile (trows >= 240)
            const bool ok = INTERIOR ? (h == 0 || jj < 7 || row < trows) : row < valid;
            if (ok && !((VM_ABL & 4) && row > 0)) *reinterpret_cast<u32x4*>(obase + (int64_t)row * p.N) = v[jj];
        }
    };
    if (FWD && p.pool_o != nullptr) {
        // (pool_e, pool_o) below carry the tile: z itself is not written
    } else if (interior) {
        half(0, std::true_type{});
        bar.point();
        half(1, std::true_type{});
        bar.point();
    } else {
        half(0, std::false_type{});
        bar.point();
        half(1, std::false_type{});
        bar.point();
    }
    if (FWD && p.pool_e != nullptr) {
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
        // centred tile (f16, fold_ctr): values may be negative -- the extreme by PACKED HALF max / min, and the pair is taken back to
        // z = t + ctr (one packed add: exact where relu clipped, t = -ctr) for the "other" element, whose flag bit needs z >= 0
        constexpr bool CAN_CENTRE = EPI == EPI_FWD_FOLD && std::is_same<T, f16>::value;
        const bool ctrd = CAN_CENTRE && p.fold_ctr != nullptr;
        u32x4 c16 = {0u, 0u, 0u, 0u};
        if (CAN_CENTRE && ctrd) {
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(p.fold_ctr + n0 + c8 * 8), q1 = *reinterpret_cast<const f32x4*>(p.fold_ctr + n0 + c8 * 8 + 4);
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            c16[0] = __builtin_bit_cast(uint32_t, h2{(_Float16)q0[0], (_Float16)q0[1]});
            c16[1] = __builtin_bit_cast(uint32_t, h2{(_Float16)q0[2], (_Float16)q0[3]});
            c16[2] = __builtin_bit_cast(uint32_t, h2{(_Float16)q1[0], (_Float16)q1[1]});
            c16[3] = __builtin_bit_cast(uint32_t, h2{(_Float16)q1[2], (_Float16)q1[3]});
        }
        const int vq = valid >> 1;
        T* ebase = p.pool_e + (n * (int64_t)(p.L / 2 + 2 * p.pool_e_pad) + p.pool_e_pad + (t0 >> 1)) * (int64_t)p.N + n0 + c8 * 8;
        T* obase2 = p.pool_o + (n * (int64_t)(p.L / 2) + (t0 >> 1)) * (int64_t)p.N + n0 + c8 * 8;
        u32x4 r0[8], r1[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int q = rg + 16 * jj;
            r0[jj] = *reinterpret_cast<const u32x4*>(lds + (2 * q) * TP + c8 * 16);
            r1[jj] = *reinterpret_cast<const u32x4*>(lds + (2 * q + 1) * TP + c8 * 16);
        }
        // (the two run-time switches select one of four straight-line copies: inside the loops hipcc branched on them per 32-bit word)
        auto pairs = [&](auto ctrd_c, auto other_c) {
            constexpr bool CTRD = decltype(ctrd_c)::value, OTHER = decltype(other_c)::value;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int q = rg + 16 * jj;
                u32x4 o, oth;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    uint32_t a = r0[jj][d], b = r1[jj][d];
                    const uint32_t m = neg[d];
                    if constexpr (CTRD) {
                        uint32_t mx, mn;
                        asm("v_pk_max_f16 %0, %1, %2" : "=v"(mx) : "v"(a), "v"(b));
                        asm("v_pk_min_f16 %0, %1, %2" : "=v"(mn) : "v"(a), "v"(b));
                        o[d] = (mx & ~m) | (mn & m);                      // the CENTRED extreme: what block i + 1 reads
                        if constexpr (OTHER) {
                            asm("v_pk_add_f16 %0, %1, %2" : "=v"(a) : "v"(a), "v"(c16[d]));
                            asm("v_pk_add_f16 %0, %1, %2" : "=v"(b) : "v"(b), "v"(c16[d]));
                        }
                    }
                    if constexpr (!CTRD || OTHER) {
                        // a, b >= 0 here (ReLU, or un-centred above), so they order as 15-bit integers; XOR with 0x7FFF reverses that
                        // order: in the flipped domain of a gamma < 0 channel the wanted minimum is a maximum too, and one packed
                        // max / min / subtract serve both kinds of channel (8 VALU operations per word instead of 11)
                        const uint32_t fl = m & 0x7FFF7FFFu;
                        const uint32_t af = a ^ fl, bf = b ^ fl;
                        uint32_t mx, mn;
                        asm("v_pk_max_i16 %0, %1, %2" : "=v"(mx) : "v"(af), "v"(bf));
                        if constexpr (!CTRD) o[d] = mx ^ fl;
                        if constexpr (OTHER) {
                            // the other element, flagged (bit 15) where the extreme is the pair's SECOND element: bf > af -- the sign of
                            // the 16-bit difference of two non-negative values; ties: the first
                            uint32_t df;
                            asm("v_pk_min_i16 %0, %1, %2" : "=v"(mn) : "v"(af), "v"(bf));
                            asm("v_pk_sub_i16 %0, %1, %2" : "=v"(df) : "v"(af), "v"(bf));
                            oth[d] = (mn ^ fl) | (df & 0x80008000u);
                        }
                    }
                }
                if (q < vq) {
                    *reinterpret_cast<u32x4*>(ebase + (int64_t)q * p.N) = o;
                    if constexpr (OTHER) *reinterpret_cast<u32x4*>(obase2 + (int64_t)q * p.N) = oth;
                }
                bar.point();
            }
        };
        const bool other = p.pool_o != nullptr;
        if constexpr (CAN_CENTRE) {
            if (ctrd && other) pairs(std::true_type{}, std::true_type{});
            else if (ctrd) pairs(std::true_type{}, std::false_type{});
            else if (other) pairs(std::false_type{}, std::true_type{});
            else pairs(std::false_type{}, std::false_type{});
        } else {
            if (other) pairs(std::false_type{}, std::true_type{});
            else pairs(std::false_type{}, std::false_type{});
        }
    }
    VM_EPI_MARK(5);
    if (stats) {
#if defined(__HIP_DEVICE_COMPILE__)
        const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)lds;
#else
        const uint32_t lds0 = 0;
#endif
        // fragment geometry of the transposing read: lane (kh, lg, li) supplies the 8-byte word of row 4 (li >> 2) + 2 kh, channels
        // 16 lg + 4 (li & 3) .. + 3 and receives channel 16 lg + li of the four rows 2 kh + {0, 4, 8, 12} (second read: + 1 row).
        // WHICH 16 positions a k-step holds does not matter to a sum over positions (both MFMA operands are these registers), and with
        // the rows of one read 4 apart their four 64-byte segments fall in four different quarters of the bank line (a 272-byte pitch
        // moves a row by 4 banks: consecutive rows overlapped in 12 of their 16 banks -- the 4-way conflicts of the round-3 counters)
        const int li = lane & 15, lg = (lane >> 4) & 1;
        const uint32_t xoff = lds0 + (4 * (li >> 2) + 2 * kh) * TP + (32 * w + 16 * lg + 4 * (li & 3)) * 2;
        const u32x2 ones = ones4<T>();
        const int srows = (p.L + 127) / 128;  // vm_conv_stat_rows
        const int cn = lane & 31;             // the channel (of this wave's 32) whose sums this lane holds: 16 lg + li
        // (round 6) The sums ride on v_mfma_f32_4x4x4 (16 blocks of 4 lanes, K = the four rows of a transposing read): with A = B = the
        // lane's own four values, D[lane][reg lane & 3] is the lane's sum of squares, with A = ones D[lane][0] its sum -- 1024
        // multiply-adds per instruction where the 32 x 32 x 16 form spent 16384 on a 32 x 32 product of which only the diagonal (or
        // one row) was wanted.  On the power limit that is what counts: the statistics phase was 13 / 15 / 6 us of the three forward
        // launches (VM_ABL=8, profiles/r06_nt3_stats_4x4.txt).  The two row halves (kh) of a channel are added across lanes 32 apart.
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 d1a = {0.f, 0.f, 0.f, 0.f}, d1b = d1a, d2a = d1a, d2b = d1a;
            u32x2 lo[8], hi[8];   // all 16 transposing reads of the half first, then the MFMAs (one latency, not eight)
#pragma unroll
            for (int rs = 0; rs < 8; ++rs) {
                const uint32_t a = xoff + (h * 128 + rs * 16) * TP;
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo[rs]) : "v"(a));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:272" : "=v"(hi[rs]) : "v"(a));  // + 1 row of 272 bytes
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
                d2a = Mfma4<T>::run(lo[rs], lo[rs], d2a);
                d2b = Mfma4<T>::run(hi[rs], hi[rs], d2b);
                d1a = Mfma4<T>::run(ones, lo[rs], d1a);
                d1b = Mfma4<T>::run(ones, hi[rs], d1b);
                __builtin_amdgcn_sched_barrier(0);
            }
            float ds = d1a[0] + d1b[0], dq = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) dq = e == (lane & 3) ? d2a[e] + d2b[e] : dq;
            ds += __shfl_xor(ds, 32, 64);
            dq += __shfl_xor(dq, 32, 64);
            if (2 * tl + h < srows && kh == 0) {
                const int64_t srow = n * srows + 2 * tl + h;
                p.stat_sum[srow * p.N + n0 + 32 * w + cn] = ds;
                p.stat_sq[srow * p.N + n0 + 32 * w + cn] = dq;
            }
            bar.point();
        }
    }
    VM_EPI_MARK(6);
    VM_RED_MARK(1);
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
        // (the row order of the statistics reads above: 4 (li >> 2) + 2 kh, second read + 1 row -- the same for dp and for A, whose
        // k indices must agree)
        const int li = lane & 15, lg = (lane >> 4) & 1;
        const uint32_t xoff = lds0 + (4 * (li >> 2) + 2 * kh) * TP + (32 * w + 16 * lg + 4 * (li & 3)) * 2;
        const u32x2 ones = ones4<T>();
        u32x2 dlo[16], dhi[16];
        f32x4 d1[2][2], d2[2][2];   // [half][lo / hi read]: v_mfma_f32_4x4x4 accumulators (see the forward statistics above)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int u = 0; u < 2; ++u) d1[h][u] = d2[h][u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const uint32_t a = xoff + (q * 16) * TP;
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dlo[q]) : "v"(a));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:272" : "=v"(dhi[q]) : "v"(a));
            if (q == 7) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dlo[0]), "+v"(dhi[0]), "+v"(dlo[1]), "+v"(dhi[1]), "+v"(dlo[2]), "+v"(dhi[2]),
                                     "+v"(dlo[3]), "+v"(dhi[3]), "+v"(dlo[4]), "+v"(dhi[4]), "+v"(dlo[5]), "+v"(dhi[5]), "+v"(dlo[6]), "+v"(dhi[6]),
                                     "+v"(dlo[7]), "+v"(dhi[7]));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dlo[8]), "+v"(dhi[8]), "+v"(dlo[9]), "+v"(dhi[9]), "+v"(dlo[10]), "+v"(dhi[10]), "+v"(dlo[11]),
                     "+v"(dhi[11]), "+v"(dlo[12]), "+v"(dhi[12]), "+v"(dlo[13]), "+v"(dhi[13]), "+v"(dlo[14]), "+v"(dhi[14]), "+v"(dlo[15]),
                     "+v"(dhi[15]));
        VM_RED_MARK(2);
        bar.sync();  // every wave has its dp fragments and has issued its output stores: the tile memory is free
        VM_RED_MARK(3);
        // A tile: 256 rows x 128 channels, 256-byte rows (unpadded: the LDS-DMA destination is lane-linear), 64 pieces of 4 rows.  A
        // row is a whole bank line, so the four rows of a transposing read would sit on the same banks: the 64-byte quarters of row R are
        // stored XOR-ed with (R >> 2) & 3 (applied to the per-lane SOURCE address here and to the read address below)
        {
            const int arow = lane >> 4, achunk = lane & 15;
            const T* abase = p.red_a + n * p.red_a_win_stride + (int64_t)(t0 + p.red_a_row0) * p.N + n0;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int pi = w + 4 * k;
                int R = pi * 4 + arow;
                const int sw = (achunk ^ ((pi & 3) << 2)) * 8;   // (R >> 2) & 3 == pi & 3
                R = R < valid ? R : valid - 1;  // rows outside the window: any finite value (their dp rows are zero)
                glds16(reinterpret_cast<const char*>(abase + (int64_t)R * p.N + sw), lds + __builtin_amdgcn_readfirstlane(pi * 1024));
                if ((k & 3) == 3) bar.point();
            }
        }
        VM_RED_MARK(4);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            d1[q >> 3][0] = Mfma4<T>::run(ones, dlo[q], d1[q >> 3][0]);
            d1[q >> 3][1] = Mfma4<T>::run(ones, dhi[q], d1[q >> 3][1]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // stores and loads retire out of order with each other: no counted wait here
        VM_RED_MARK(5);
        bar.sync();  // the A tile has landed for every wave
        VM_RED_MARK(6);
        const uint32_t aoff = lds0 + (4 * (li >> 2) + 2 * kh) * 256 + (((32 * w + 16 * lg + 4 * (li & 3)) * 2) ^ ((li >> 2) << 6));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32x2 alo[8], ahi[8];
#pragma unroll
            for (int rs = 0; rs < 8; ++rs) {
                const uint32_t a = aoff + (h * 128 + rs * 16) * 256;
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(alo[rs]) : "v"(a));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:256" : "=v"(ahi[rs]) : "v"(a));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(alo[0]), "+v"(ahi[0]), "+v"(alo[1]), "+v"(ahi[1]), "+v"(alo[2]), "+v"(ahi[2]), "+v"(alo[3]),
                         "+v"(ahi[3]), "+v"(alo[4]), "+v"(ahi[4]), "+v"(alo[5]), "+v"(ahi[5]), "+v"(alo[6]), "+v"(ahi[6]), "+v"(alo[7]), "+v"(ahi[7]));
#pragma unroll
            for (int rs = 0; rs < 8; ++rs) {
                const int q = h * 8 + rs;
                d2[h][0] = Mfma4<T>::run(dlo[q], alo[rs], d2[h][0]);
                d2[h][1] = Mfma4<T>::run(dhi[q], ahi[rs], d2[h][1]);
            }
        }
        const int cn = lane & 31;
        const int rows2 = 2 * p.tilesL;  // partial rows per window (vm_conv_dgrad_bnred_rows)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float ds = d1[h][0][0] + d1[h][1][0], dq = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) dq = e == (lane & 3) ? d2[h][0][e] + d2[h][1][e] : dq;
            ds += __shfl_xor(ds, 32, 64);   // the two row halves (kh) of the channel
            dq += __shfl_xor(dq, 32, 64);
            const int64_t srow = n * rows2 + 2 * tl + h;
            if (kh == 0) {
                p.stat_sum[srow * p.N + n0 + 32 * w + cn] = ds;
                p.stat_sq[srow * p.N + n0 + 32 * w + cn] = dq;
            }
        }
    }
}


template <typename T, int EPI, bool WIDE = false>
__device__ inline void n2_epilogue(const NtArgs<T>& p, char* lds, const f32x16 (&acc)[4][2], int64_t n, int tl, int t0, int n0, int trows,
                                   int tid, int lane, int w, int wm, int wn) {
    f32x4 negc[2][4];
    if constexpr (EPI == EPI_FWD_FOLD && std::is_same<T, f16>::value) {
        if (p.fold_ctr != nullptr) {   // issued here: the barrier below hides their latency
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 c = *reinterpret_cast<const f32x4*>(p.fold_ctr + n0 + (WIDE ? wn * 32 : wn * 64 + j * 32) + 8 * g + 4 * (lane >> 5));
                    negc[j][g] = f32x4{-c[0], -c[1], -c[2], -c[3]};
                }
        }
    }
    VM_EPI_MARK(1);
    __syncthreads();  // every wave is done with the operand stages: they become the epilogue tile
    VM_EPI_MARK(2);
    n2_tile_write<T, EPI, WIDE>(p, lds, acc, t0, n0, trows, lane, wm, wn, negc);
    VM_EPI_MARK(3);
    __syncthreads();
    VM_EPI_MARK(4);
    N2DrainSync bar;
    n2_tile_drain<T, EPI>(p, lds, n, tl, t0, n0, trows, tid, lane, w, bar);
}

// ------------------------------------------------------------------------------------------------
// conv_nt2r_kernel: the tile above with an INPUT-RESIDENT A operand.  Staging one (A 256 rows + B 128 rows) x 64-byte slice per K
// tile -- the first form of this kernel -- moved 24 KB per 64 MFMAs, 85 FLOP per byte of L2 -> LDS traffic, and its ablations
// (us; forward 256->384 / dgrad 384->512: full 278 / 244, no epilogue 204 / 196, no in-loop DMA 197 / 153, neither 142 / 136)
// named the global -> LDS stream as the longest pole.
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

template <typename T, int EPI>
__global__ __launch_bounds__(256, 2) void conv_nt2r_kernel(NtArgs<T> p, int64_t n_groups) {
    VM_PROF(const long long pt_start = __builtin_amdgcn_s_memtime(); long long pt_first = 0, pt_bar = 0;)
    using namespace n2;
    using namespace n2r;
    using V8 = typename Mfma<T>::Frag;
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
        if ((n_groups & 7) == 0) {
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
    if constexpr (EPI == EPI_FWD_FOLD) {  // this window's tower: its folded weights and constants
        const unsigned tw = nw / (unsigned)p.tower_windows;
        p.bt += tw * p.bt_tower_stride;
        p.fold_hb += tw * 4 * p.N;
        if (p.fold_ctr != nullptr) p.fold_ctr += tw * p.N;
    }
    // forward: the bias loads go out first and are consumed (accumulator init) only after the prologue DMA has been issued
    f32x4 bias4[2][4];
    if constexpr (EPI == EPI_FWD_FOLD) p.bias = p.fold_hb + 3 * p.N;  // row 3 of hb: bias + the three per-tap constants (vm_fold_bn_weights)
    n2_load_bias<T, EPI>(p, bias4, n0 + wn * 64 + 4 * (lane >> 5));

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
        for (int i = 0; i < 2; ++i) glds16(a_src[i0 + i] + chunk * KB, base + __builtin_amdgcn_readfirstlane((w + 4 * (i0 + i)) * 1024));
    };
    auto issue_b = [&](int stg, int ko) {
        char* base = lds + B0 + stg * B_STG;
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(b_src[i] + ko, base + __builtin_amdgcn_readfirstlane((w + 4 * i) * 1024));
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
            {
                if (kt + 2 < nk && !(VM_ABL & 1)) {
                    issue_b(b_stage, b_tap * row_bytes + b_chunk_off);
                    b_stage = b_stage == 2 ? 0 : b_stage + 1;
                    if (++b_tap == 3) {
                        b_tap = 0;
                        b_chunk_off += KB;
                    }
                    ib = 2;
                }
                if (tap < 2 && more_a && !(VM_ABL & 2)) {
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
#define VM_MM(A, B, I, J) acc[I][J] = Mfma<T>::run(__builtin_bit_cast(V8, B), __builtin_bit_cast(V8, A), acc[I][J])
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
    VM_PROF(const long long pt_loop = __builtin_amdgcn_s_memtime();)
    if constexpr (EPI == EPI_FWD_FOLD) {
        const int c0 = n0 + wn * 64 + 4 * (lane >> 5), rl = p.L - 1 - t0;
        if (t0 == 0) n2_fold_edge<T>(p, acc, 0, 0, wm, lane & 31, c0);
        if (rl >= 0 && rl < TROWS) n2_fold_edge<T>(p, acc, rl, 1, wm, lane & 31, c0);
    }
    n2_epilogue<T, EPI>(p, lds, acc, n, tl, t0, n0, TROWS, tid, lane, w, wm, wn);
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


// ------------------------------------------------------------------------------------------------
// conv_nt3_kernel (round 4): the tile of conv_nt2r_kernel with the WEIGHT operand taken from L2 straight into registers.
//
// conv_nt2r_kernel's K loop runs at 1 340 .. 1 410 clocks per K tile against 1 024 of MFMA issue (two waves share a SIMD), and its
// ablation put 13 % of forward + dgrad on the in-loop weight LDS-DMA alone: a K tile costs a wave 2 weight pieces + 1.3 input pieces of
// LDS-DMA (~250 clocks of issue each under load), 4 of its 12 fragment reads, and a workgroup barrier whose only purpose is the reuse
// of the 8 KB weight stages.  The weights are small (0.2 .. 1.2 MB per layer), read by every workgroup and therefore L2-resident.
// Here vm_pack_nt_weights lays them out in MFMA fragment order -- [tower][64-channel block][K tile][j][k-step][lane][8 values], so
// that one global_load_dwordx4 of a wave IS one 32 x 16 B fragment, 1 KB contiguous, and a wave's stream is 4 KB per K tile in the
// order the loop walks -- and the loop keeps three register sets of four fragments: K tile kt multiplies out of set kt % 3 while the
// loads of K tile kt + 2 fill set (kt + 2) % 3.  What is left in LDS is the input operand: a ring of FOUR 16 KB blocks (256 rows x
// 32 channels, the layout and swizzle of conv_nt2r_kernel; c_in = 128 -- the block-2 forward -- is resident outright), block
// c + 3 requested during chunk c, and ONE barrier per chunk (48 MFMAs per wave) instead of one per K tile:
//   RAW  A(c) was issued three chunks earlier, B(kt) two K tiles earlier; loads return in order, so the counted wait for B(kt) at the
//        top of iteration kt covers both; the barrier at tap 0 makes the other waves' pieces of A(c) visible.
//   WAR  A(c + 3) overwrites the block read during chunk c - 1; it is issued after the barrier of chunk c, which a wave passes only
//        with its chunk c - 1 reads consumed.  Register set (kt + 2) % 3 was consumed by the MFMAs of iteration kt - 1.
// The global loads are inline asm: beside an LDS-DMA hipcc waits vmcnt(0) before the first use of any ordinary load's result (and it
// cannot count asm loads at all), so every wait is written by hand from the issue schedule:
//   iteration kt:  B(kt + 2) [4 loads]  ->  s_waitcnt vmcnt(N)  ->  barrier (tap 0)  ->  A pieces [2; taps 0, 1]  ->  8 ds_read + 16 MFMA
//   N = loads issued after B(kt) = A pieces of kt - 2  +  B(kt + 1)  +  A pieces of kt - 1  +  B(kt + 2)
// The epilogue is conv_nt2r_kernel's (n2_epilogue).  Same requirements: a_c % 32 == 0, Ktot == 3 * a_c, N % 128 == 0.
// ------------------------------------------------------------------------------------------------
// timing experiments of conv_nt3_kernel (WRONG results; tools/build_variant.sh <name> -DVM_NT3_ABL=<bits>): 1 no in-loop weight loads,
// 2 no in-loop input DMA, 4 no K-loop MFMAs, 8 no epilogue
#ifndef VM_NT3_ABL
#define VM_NT3_ABL 0
#endif
namespace n3 {
constexpr int NBLK = 4;
constexpr int A_BLK = 256 * 64;
static_assert(NBLK * A_BLK <= n2::LDS_BYTES, "the input ring fits under the epilogue tile");
constexpr int KT_BYTES = 4096;  // one wave's weight fragments of one K tile: [j 2][k-step 2][lane 64][16 B]
}  // namespace n3

// one fragment: lane l gets the 16 bytes at sbase + imm + l * 16 (voff = l * 16); the result is valid after the counted wait that
// names the register
#define VM_GLOAD_FRAG(dst, voff, sbase, imm) \
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #imm : "=v"(dst) : "v"(voff), "s"(sbase) : "memory")

template <int N>
__device__ inline void n3_wait_b(u32x4 (&b)[4]) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N) : "memory");
}

// The issue schedule of conv_nt3_kernel as constexpr functions: which input (A) pieces iteration i = 3 c + tap issues under its last
// eight MFMAs, and -- by replaying the whole schedule -- the number of vector-memory operations a wave has issued after the last
// fragment of B(kt) when it reaches the wait that opens iteration kt: the N of that iteration's s_waitcnt vmcnt(N).
//   prologue          A(0) [4]   B(0) [4]   B(1) [4]   A(1) [4, chunks > 1]            (LEAN; the first wait needs A(0), B(0) only)
//                     A(0) [4]   B(0) [4]   A(1) [4]   B(1) [4]   A(2) [4, chunks > 2]  (!LEAN: the first form)
//   iteration i       WAIT   B(i + 2) [4, i + 2 < nk, under the first eight MFMAs]   A pieces [under the last eight]:
//                     LEAN: i = 0: all four of A(2); i = 1: all four of A(3); i = 2: none; from chunk 1 on as below
//                     pieces 2 tap, 2 tap + 1 of A(c + 3) at taps 0, 1 while c + 3 < chunks
// LEAN issues four LDS-DMA instructions (~250 ticks each under load) fewer before the first MFMA of a tile.
struct N3Pieces {
    int count, block, first;
};
constexpr N3Pieces n3_pieces(int i, int chunks, bool lean) {
    const int c = i / 3, tap = i - 3 * c;
    if (lean && i == 0) return N3Pieces{chunks > 2 ? 4 : 0, 2, 0};
    if (lean && i == 1) return N3Pieces{chunks > 3 ? 4 : 0, 3, 0};
    if (lean && c == 0) return N3Pieces{0, 0, 0};
    return N3Pieces{(tap < 2 && c + 3 < chunks) ? 2 : 0, c + 3, 2 * tap};
}
constexpr int n3_nwait(int kt, int chunks, bool lean, int bl = 4) {   // bl: loads per weight set (4; the 256 x 32 wave tile: 2)
    const int nk = 3 * chunks;
    int after = -1;  // operations issued since B(kt) completed its issue; -1: B(kt) not issued yet
    auto issue = [&](int count, bool is_bkt) {
        if (is_bkt) {
            after = 0;
        } else if (after >= 0) {
            after += count;
        }
    };
    issue(4, false);                       // A(0)
    issue(bl, kt == 0);                    // B(0)
    if (lean) {
        issue(bl, kt == 1);                // B(1)
        if (chunks > 1) issue(4, false);   // A(1)
    } else {
        if (chunks > 1) issue(4, false);   // A(1)
        issue(bl, kt == 1);                // B(1)
        if (chunks > 2) issue(4, false);   // A(2)
    }
    for (int i = 0; i < kt; ++i) {         // (the wait of iteration kt opens it)
        if (i + 2 < nk) issue(bl, i + 2 == kt);
        issue(n3_pieces(i, chunks, lean).count, false);
    }
    return after;
}
static_assert(n3_nwait(0, 4, false) == 12 && n3_nwait(1, 4, false) == 10 && n3_nwait(2, 4, false) == 8 && n3_nwait(11, 4, false) == 0 &&
              n3_nwait(5, 8, false) == 8, "conv_nt3_kernel wait schedule");
static_assert(n3_nwait(0, 4, true) == 8 && n3_nwait(1, 4, true) == 12 && n3_nwait(2, 4, true) == 12 && n3_nwait(3, 4, true) == 8 &&
              n3_nwait(4, 8, true) == 6 && n3_nwait(11, 4, true) == 0, "conv_nt3_kernel wait schedule (lean prologue)");

// (Round 6, measured and removed: a start offset of half a tile for the workgroup in a CU's second slot in the first round of the grid.
// The slot timelines of tools/probe/nt3_slots.py show why it cannot pay: the two workgroups of a CU already run half a tile out of
// phase on their own -- median phase of the second slot's starts 0.48-0.52 on all six launches, profiles/r06_nt3_slots.txt -- and the
// launches gained nothing at any offset, 0 .. +3 % with the delay itself; round 2 had the same result on conv_nt2r_kernel.)
// LEAN: the prologue that leaves A(2), A(3) to the first two K tiles (A/B switch nt3_lean)
// WIDE (round 6 experiment, vm_set_tuning "nt3_wide"): the wave tile is 256 rows x 32 channels instead of 128 x 64 -- every wave reads
// ALL of the block's rows from LDS (16 fragment reads per K tile instead of 8) and HALF the weight fragments from L2 (2 instead of
// 4: the weight stream, 1.77 GB per launch, is the largest movable slice of a launch's energy: profiles/r06_nt3_ablation.txt).
template <typename T, int EPI, int CHUNKS, bool LEAN, bool WIDE = false>
__global__ __launch_bounds__(256, 2) void conv_nt3_kernel(NtArgs<T> p, int64_t n_groups) {
    VM_PROF(const long long pt_start = __builtin_amdgcn_s_memtime(); long long pt_first = 0, pt_bar = 0;)
    using namespace n2;
    using V8 = typename Mfma<T>::Frag;
    constexpr int A_BLK = n3::A_BLK, NK = 3 * CHUNKS;
    __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = WIDE ? 0 : (w >> 1), wn = WIDE ? w : (w & 1);
    constexpr int WC = WIDE ? 32 : 64;   // channels per wave

    unsigned group;
    int tn;
    {
        const unsigned v = blockIdx.x, tiles_n = (unsigned)p.tilesN;
        if ((n_groups & 7) == 0) {
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
    const int t0 = tl * n2r::TROWS, n0 = tn * TN;
    unsigned tw = 0;
    if constexpr (EPI == EPI_FWD_FOLD) {  // this window's tower: its folded weights and constants
        tw = nw / (unsigned)p.tower_windows;
        p.fold_hb += tw * 4 * p.N;
        if (p.fold_ctr != nullptr) p.fold_ctr += tw * p.N;
    }
    if constexpr (EPI == EPI_FWD_FOLD) p.bias = p.fold_hb + 3 * p.N;  // row 3 of hb: bias + the three per-tap constants (vm_fold_bn_weights)
    // The wave's 64 bias values by SCALAR loads (the address is wave-uniform; a lane wants the 32 of its half kh).  As 8 vector loads
    // per lane they cost ~3000 ticks of VMEM issue in the setup, and -- hipcc cannot see the counted waits behind the inline-asm loads
    // -- an s_waitcnt vmcnt(0) in front of the first MFMA, which drained the whole prologue (first K tile 2500 ticks against 760).
    // They must stay IN FRONT of the first inline-asm statement with a memory clobber: behind one, hipcc no longer proves the bias
    // unclobbered and falls back to vector loads (16 of them, and the vmcnt(0) again).
    f32x4 bias_lo[2][4], bias_hi[2][4];
    if constexpr (EPI != EPI_DGRAD) {
        const float* bp = p.bias + n0 + wn * WC;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {   // (WIDE: j is a row block, both take the wave's one 32-channel block)
                bias_lo[j][g] = *reinterpret_cast<const f32x4*>(bp + (WIDE ? 0 : 32 * j) + 8 * g);
                bias_hi[j][g] = *reinterpret_cast<const f32x4*>(bp + (WIDE ? 0 : 32 * j) + 8 * g + 4);
            }
    }

    // ---- input DMA sources (as conv_nt2r_kernel): one instruction = 16 rows x 64 B ----
    const int lrow = lane >> 2, lchunk = lane & 3;
    const char* a_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (w + 4 * i) * 16 + lrow;
        int pr = t0 + row;
        pr = pr < p.L + 2 ? pr : p.L + 1;
        a_src[i] = reinterpret_cast<const char*>(p.a + n * p.a_win_stride + (int64_t)pr * p.a_c) + ((lchunk ^ ((row >> 2) & 3)) << 4);
    }
    auto issue_a = [&](int blk, int chunk, int i0) {  // pieces i0, i0 + 1 of this wave's four
        char* base = lds + blk * A_BLK;
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(a_src[i0 + i] + chunk * KB, base + __builtin_amdgcn_readfirstlane((w + 4 * (i0 + i)) * 1024));
    };
    auto issue_a1 = [&](int blk, int chunk, int i) {  // piece i of this wave's four
        glds16(a_src[i] + chunk * KB, lds + blk * A_BLK + __builtin_amdgcn_readfirstlane((w + 4 * i) * 1024));
    };
    // ---- weight stream of this wave: (tower, 64-channel block n0 / 64 + wn), NK x 4 KB, base in SGPRs ----
    uint64_t bbase;
    {
        // (WIDE: the wave's 32 channels are fragment row j = wn & 1 of the 64-channel block (n0 >> 6) + (wn >> 1): 2 KB into each K tile)
        const uint64_t q = (uint64_t)(uintptr_t)p.bt_packed +
                           ((uint64_t)tw * (unsigned)(p.N >> 6) + (unsigned)((n0 >> 6) + (WIDE ? (wn >> 1) : wn))) * (uint64_t)(NK * n3::KT_BYTES) +
                           (WIDE ? (uint64_t)((wn & 1) * 2048) : 0);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)q), hi = __builtin_amdgcn_readfirstlane((uint32_t)(q >> 32));
        bbase = ((uint64_t)hi << 32) | lo;
    }
    const uint32_t bvoff = lane * 16;
    u32x4 bs[3][4];  // register sets of K tiles kt % 3 = 0, 1, 2: fragments (j, k-step) = 00, 01, 10, 11 (constant indices only)
#define VM_LOAD_SET(KT)                                                           \
    {                                                                             \
        const uint64_t sb_ = bbase + (uint64_t)((KT) * n3::KT_BYTES);             \
        VM_GLOAD_FRAG(bs[(KT) % 3][0], bvoff, sb_, 0);                            \
        VM_GLOAD_FRAG(bs[(KT) % 3][1], bvoff, sb_, 1024);                         \
        if constexpr (!WIDE) {                                                    \
            VM_GLOAD_FRAG(bs[(KT) % 3][2], bvoff, sb_, 2048);                     \
            VM_GLOAD_FRAG(bs[(KT) % 3][3], bvoff, sb_, 3072);                     \
        }                                                                         \
    }

    const int r = lane & 31, kh = lane >> 5;
    int a_addr[3][2];
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) {
        const int row = wm * 128 + r + tap;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) a_addr[tap][ks] = row * KB + (((2 * ks + kh) ^ ((row >> 2) & 3)) << 4);
    }
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)lds;
#else
    const uint32_t lds0 = 0;
#endif

    VM_PROF(const long long pt_s2 = __builtin_amdgcn_s_memtime();)
    // ---- prologue (see n3_nwait) ----
    issue_a(0, 0, 0);
    issue_a(0, 0, 2);
    VM_LOAD_SET(0);
    if constexpr (LEAN) VM_LOAD_SET(1);
    if constexpr (CHUNKS > 1) {
        issue_a(1, 1, 0);
        issue_a(1, 1, 2);
    }
    if constexpr (!LEAN) {
        VM_LOAD_SET(1);
        if constexpr (CHUNKS > 2) {
            issue_a(2, 2, 0);
            issue_a(2, 2, 2);
        }
    }
    VM_PROF(const long long pt_s3 = __builtin_amdgcn_s_memtime();)
    f32x16 acc[4][2];
    {
        f32x4 bias4[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bias4[j][g] = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (EPI != EPI_DGRAD) bias4[j][g] = (lane >> 5) ? bias_hi[j][g] : bias_lo[j][g];
            }
        n2_fill_acc(acc, bias4);
    }
#if VM_NT3_ABL & 4
#define VM_MM(A, B, I, J) asm volatile("" : "+v"(acc[I][J]) : "v"(A), "v"(B))
#else
#define VM_MM(A, B, I, J) acc[I][J] = Mfma<T>::run(__builtin_bit_cast(V8, B), __builtin_bit_cast(V8, A), acc[I][J])
#endif
    // KT is a literal in the macros below: every index, every wait count and every branch is a compile-time constant, the loop is
    // straight-line code and no register that a load is still writing ever meets a phi (the rolled form made hipcc copy them)
    // ---- the interleaved loop.  Every memory operation of a K tile sits INSIDE its MFMA stream, one per pair of MFMAs: the
    // k-step-1 fragments of this tile and the weight fragments of tile kt + 2 under the k-step-0 MFMAs, the k-step-0 fragments of the
    // NEXT tile (their registers are free by then) and the input DMA pieces under the k-step-1 MFMAs.  An LDS read has eight MFMAs
    // (256 clocks) to return, a wave never has an issue-only phase, and LDS returns in order with exactly three reads younger than
    // the one an MFMA pair needs: every wait is lgkmcnt(3) until the last tile drains.  The chunk barrier moves to the middle of
    // tap 2 (before the first read of the next block), behind lgkmcnt(0): all reads of this chunk's block have RETURNED when a wave
    // passes it, so the DMA pieces that recycle the block -- issued at least one tile later -- cannot overtake a read.
    u32x4 f0[WIDE ? 8 : 4], f1[WIDE ? 8 : 4];
#define VM_FRAG_READ(dst, base, I)                                                              \
    if constexpr ((I) == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(base));          \
    if constexpr ((I) == 1) asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(dst) : "v"(base)); \
    if constexpr ((I) == 2) asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(dst) : "v"(base)); \
    if constexpr ((I) == 3) asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(dst) : "v"(base)); \
    if constexpr ((I) == 4) asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(dst) : "v"(base)); \
    if constexpr ((I) == 5) asm volatile("ds_read_b128 %0, %1 offset:10240" : "=v"(dst) : "v"(base)); \
    if constexpr ((I) == 6) asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(dst) : "v"(base)); \
    if constexpr ((I) == 7) asm volatile("ds_read_b128 %0, %1 offset:14336" : "=v"(dst) : "v"(base));
#define VM_P_STEP0(I)                                                                                                                 \
    asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(f0[I]));                                                                               \
    VM_MM(f0[I], bs[cur_][0], I, 0);                                                                                                  \
    VM_MM(f0[I], bs[cur_][2], I, 1);                                                                                                  \
    VM_FRAG_READ(f1[I], aa1_, I)                                                                                                      \
    if constexpr (kt_ + 2 < NK && !(VM_NT3_ABL & 1)) {                                                                                \
        if constexpr ((I) == 0) VM_GLOAD_FRAG(bs[nxt_][0], bvoff, sb_, 0);                                                            \
        if constexpr ((I) == 1) VM_GLOAD_FRAG(bs[nxt_][1], bvoff, sb_, 1024);                                                         \
        if constexpr ((I) == 2) VM_GLOAD_FRAG(bs[nxt_][2], bvoff, sb_, 2048);                                                         \
        if constexpr ((I) == 3) VM_GLOAD_FRAG(bs[nxt_][3], bvoff, sb_, 3072);                                                         \
    }                                                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);
#define VM_P_STEP1(I)                                                                                                                 \
    if constexpr (kt_ + 1 < NK) {                                                                                                     \
        asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(f1[I]));                                                                           \
    } else {                                                                                                                          \
        if constexpr ((I) == 0) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(f1[I]));                                                   \
        if constexpr ((I) == 1) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(f1[I]));                                                   \
        if constexpr ((I) == 2) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(f1[I]));                                                   \
        if constexpr ((I) == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f1[I]));                                                   \
    }                                                                                                                                 \
    VM_MM(f1[I], bs[cur_][1], I, 0);                                                                                                  \
    VM_MM(f1[I], bs[cur_][3], I, 1);                                                                                                  \
    if constexpr (kt_ + 1 < NK) { VM_FRAG_READ(f0[I], an0_, I) }                                                                      \
    if constexpr ((I) < n3_pieces(kt_, CHUNKS, LEAN).count && !(VM_NT3_ABL & 2))                                                      \
        issue_a1(n3_pieces(kt_, CHUNKS, LEAN).block % 4, n3_pieces(kt_, CHUNKS, LEAN).block, n3_pieces(kt_, CHUNKS, LEAN).first + (I)); \
    __builtin_amdgcn_sched_barrier(0);
#define VM_KTILE_P(KT)                                                                                                                \
    if constexpr ((KT) < NK) {                                                                                                        \
        constexpr int kt_ = (KT), c_ = kt_ / 3, tap_ = kt_ - 3 * c_, cur_ = kt_ % 3, nxt_ = (kt_ + 2) % 3, ablk_ = c_ % 4;             \
        constexpr int nc_ = (kt_ + 1) / 3, ntap_ = (kt_ + 1) - 3 * nc_;   /* chunk and tap of the next tile */                          \
        n3_wait_b<n3_nwait(kt_, CHUNKS, LEAN)>(bs[cur_]);                                                                             \
        if constexpr (kt_ == 0) {                                                                                                     \
            __builtin_amdgcn_s_barrier();                                                                                             \
            VM_PROF(pt_bar = __builtin_amdgcn_s_memtime();)                                                                           \
            const uint32_t a00_ = lds0 + a_addr[0][0];                                                                                \
            VM_FRAG_READ(f0[0], a00_, 0) VM_FRAG_READ(f0[1], a00_, 1) VM_FRAG_READ(f0[2], a00_, 2) VM_FRAG_READ(f0[3], a00_, 3)         \
        }                                                                                                                             \
        const uint32_t aa1_ = lds0 + ablk_ * A_BLK + a_addr[tap_][1];                                                                  \
        const uint32_t an0_ = lds0 + (nc_ % 4) * A_BLK + a_addr[ntap_][0];                                                             \
        const uint64_t sb_ = bbase + (uint64_t)((kt_ + 2) * n3::KT_BYTES);                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                                            \
        VM_P_STEP0(0) VM_P_STEP0(1) VM_P_STEP0(2) VM_P_STEP0(3)                                                                       \
        if constexpr (tap_ == 2 && kt_ + 1 < NK) {                                                                                    \
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f1[0]), "+v"(f1[1]), "+v"(f1[2]), "+v"(f1[3]));                                 \
            __builtin_amdgcn_s_barrier();                                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                                                        \
        }                                                                                                                             \
        VM_P_STEP1(0) VM_P_STEP1(1) VM_P_STEP1(2) VM_P_STEP1(3)                                                                       \
        VM_PROF(if (kt_ == 0) pt_first = __builtin_amdgcn_s_memtime();)                                                               \
    }
    // ---- the same loop for the 256 x 32 wave tile: ONE MFMA per fragment read (row block I = 0..7 against the wave's one weight
    // fragment of the k-step), so a read has seven younger reads behind it when its MFMA needs it: every wait is lgkmcnt(7) until the
    // last tile drains; the weight fragments of tile kt + 2 ride under row blocks 0, 1 of k-step 0, the input DMA pieces under row
    // blocks 0..3 of k-step 1 as before.  acc[I & 3][I >> 2] is row block I.
#define VM_W_WAIT(N, R) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(R))
#define VM_W_STEP0(I)                                                                                                                 \
    VM_W_WAIT(7, f0[I]);                                                                                                              \
    VM_MM(f0[I], bs[cur_][0], (I) & 3, (I) >> 2);                                                                                     \
    VM_FRAG_READ(f1[I], aa1_, I)                                                                                                      \
    if constexpr (kt_ + 2 < NK && !(VM_NT3_ABL & 1)) {                                                                                \
        if constexpr ((I) == 0) VM_GLOAD_FRAG(bs[nxt_][0], bvoff, sb_, 0);                                                            \
        if constexpr ((I) == 1) VM_GLOAD_FRAG(bs[nxt_][1], bvoff, sb_, 1024);                                                         \
    }                                                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);
#define VM_W_STEP1(I)                                                                                                                 \
    if constexpr (kt_ + 1 < NK) {                                                                                                     \
        VM_W_WAIT(7, f1[I]);                                                                                                          \
    } else {                                                                                                                          \
        if constexpr ((I) == 0) VM_W_WAIT(7, f1[I]);                                                                                  \
        if constexpr ((I) == 1) VM_W_WAIT(6, f1[I]);                                                                                  \
        if constexpr ((I) == 2) VM_W_WAIT(5, f1[I]);                                                                                  \
        if constexpr ((I) == 3) VM_W_WAIT(4, f1[I]);                                                                                  \
        if constexpr ((I) == 4) VM_W_WAIT(3, f1[I]);                                                                                  \
        if constexpr ((I) == 5) VM_W_WAIT(2, f1[I]);                                                                                  \
        if constexpr ((I) == 6) VM_W_WAIT(1, f1[I]);                                                                                  \
        if constexpr ((I) == 7) VM_W_WAIT(0, f1[I]);                                                                                  \
    }                                                                                                                                 \
    VM_MM(f1[I], bs[cur_][1], (I) & 3, (I) >> 2);                                                                                     \
    if constexpr (kt_ + 1 < NK) { VM_FRAG_READ(f0[I], an0_, I) }                                                                      \
    if constexpr ((I) < n3_pieces(kt_, CHUNKS, LEAN).count && !(VM_NT3_ABL & 2))                                                      \
        issue_a1(n3_pieces(kt_, CHUNKS, LEAN).block % 4, n3_pieces(kt_, CHUNKS, LEAN).block, n3_pieces(kt_, CHUNKS, LEAN).first + (I)); \
    __builtin_amdgcn_sched_barrier(0);
#define VM_KTILE_W(KT)                                                                                                                \
    if constexpr ((KT) < NK) {                                                                                                        \
        constexpr int kt_ = (KT), c_ = kt_ / 3, tap_ = kt_ - 3 * c_, cur_ = kt_ % 3, nxt_ = (kt_ + 2) % 3, ablk_ = c_ % 4;             \
        constexpr int nc_ = (kt_ + 1) / 3, ntap_ = (kt_ + 1) - 3 * nc_;   /* chunk and tap of the next tile */                          \
        asm volatile("s_waitcnt vmcnt(%2)" : "+v"(bs[cur_][0]), "+v"(bs[cur_][1]) : "n"(n3_nwait(kt_, CHUNKS, LEAN, 2)) : "memory");  \
        if constexpr (kt_ == 0) {                                                                                                     \
            __builtin_amdgcn_s_barrier();                                                                                             \
            VM_PROF(pt_bar = __builtin_amdgcn_s_memtime();)                                                                           \
            const uint32_t a00_ = lds0 + a_addr[0][0];                                                                                \
            VM_FRAG_READ(f0[0], a00_, 0) VM_FRAG_READ(f0[1], a00_, 1) VM_FRAG_READ(f0[2], a00_, 2) VM_FRAG_READ(f0[3], a00_, 3)         \
            VM_FRAG_READ(f0[4], a00_, 4) VM_FRAG_READ(f0[5], a00_, 5) VM_FRAG_READ(f0[6], a00_, 6) VM_FRAG_READ(f0[7], a00_, 7)         \
        }                                                                                                                             \
        const uint32_t aa1_ = lds0 + ablk_ * A_BLK + a_addr[tap_][1];                                                                  \
        const uint32_t an0_ = lds0 + (nc_ % 4) * A_BLK + a_addr[ntap_][0];                                                             \
        const uint64_t sb_ = bbase + (uint64_t)((kt_ + 2) * n3::KT_BYTES);                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                                            \
        VM_W_STEP0(0) VM_W_STEP0(1) VM_W_STEP0(2) VM_W_STEP0(3) VM_W_STEP0(4) VM_W_STEP0(5) VM_W_STEP0(6) VM_W_STEP0(7)               \
        if constexpr (tap_ == 2 && kt_ + 1 < NK) {                                                                                    \
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f1[0]), "+v"(f1[1]), "+v"(f1[2]), "+v"(f1[3]), "+v"(f1[4]), "+v"(f1[5]),       \
                         "+v"(f1[6]), "+v"(f1[7]));                                                                                   \
            __builtin_amdgcn_s_barrier();                                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                                                        \
        }                                                                                                                             \
        VM_W_STEP1(0) VM_W_STEP1(1) VM_W_STEP1(2) VM_W_STEP1(3) VM_W_STEP1(4) VM_W_STEP1(5) VM_W_STEP1(6) VM_W_STEP1(7)               \
        VM_PROF(if (kt_ == 0) pt_first = __builtin_amdgcn_s_memtime();)                                                               \
    }
#define VM_CHUNK(C) VM_KTILE_P(3 * (C)) VM_KTILE_P(3 * (C) + 1) VM_KTILE_P(3 * (C) + 2)
#define VM_CHUNK_W(C) VM_KTILE_W(3 * (C)) VM_KTILE_W(3 * (C) + 1) VM_KTILE_W(3 * (C) + 2)
    if constexpr (WIDE) {
        VM_CHUNK_W(0) VM_CHUNK_W(1) VM_CHUNK_W(2) VM_CHUNK_W(3) VM_CHUNK_W(4) VM_CHUNK_W(5) VM_CHUNK_W(6) VM_CHUNK_W(7)
        VM_CHUNK_W(8) VM_CHUNK_W(9) VM_CHUNK_W(10) VM_CHUNK_W(11) VM_CHUNK_W(12) VM_CHUNK_W(13) VM_CHUNK_W(14) VM_CHUNK_W(15)
    } else {
        VM_CHUNK(0) VM_CHUNK(1) VM_CHUNK(2) VM_CHUNK(3) VM_CHUNK(4) VM_CHUNK(5) VM_CHUNK(6) VM_CHUNK(7)
        VM_CHUNK(8) VM_CHUNK(9) VM_CHUNK(10) VM_CHUNK(11) VM_CHUNK(12) VM_CHUNK(13) VM_CHUNK(14) VM_CHUNK(15)
    }
    static_assert(CHUNKS <= 16, "conv_nt3_kernel: at most 16 channel chunks are written out");
#undef VM_CHUNK
#undef VM_CHUNK_W
#undef VM_KTILE_W
#undef VM_W_STEP0
#undef VM_W_STEP1
#undef VM_W_WAIT
#undef VM_KTILE_P
#undef VM_P_STEP0
#undef VM_P_STEP1
#undef VM_FRAG_READ
#undef VM_MM
#undef VM_LOAD_SET
    if constexpr (EPI == EPI_FWD_FOLD) {
        const int c0 = n0 + wn * WC + 4 * (lane >> 5), rl = p.L - 1 - t0;
        if (t0 == 0) n2_fold_edge<T, WIDE>(p, acc, 0, 0, wm, lane & 31, c0);
        if (rl >= 0 && rl < n2r::TROWS) n2_fold_edge<T, WIDE>(p, acc, rl, 1, wm, lane & 31, c0);
    }
    VM_PROF(const long long pt_loop = __builtin_amdgcn_s_memtime();)
#if defined(VM_NT3_EPI_PRIO)   // experiment build: the epilogue at a raised wave priority (its VALU beside the co-resident wave's MFMAs)
    __builtin_amdgcn_s_setprio(VM_NT3_EPI_PRIO);
#endif
#if VM_NT3_ABL & 8
    if (acc[0][0][0] == 123.456f && acc[3][1][15] == 1.5f) n2_epilogue<T, EPI, WIDE>(p, lds, acc, n, tl, t0, n0, n2r::TROWS, tid, lane, w, wm, wn);
#else
    n2_epilogue<T, EPI, WIDE>(p, lds, acc, n, tl, t0, n0, n2r::TROWS, tid, lane, w, wm, wn);
#endif
#if defined(VM_EXPERIMENT_PROFILE)
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long pt_end = __builtin_amdgcn_s_memtime();
#if defined(VM_EXPERIMENT_PROFILE_HW)
        // where and when a workgroup ran (tools/probe/nt3_slots.py): raw stamps + the hardware ids of its CU / workgroup slot
        if (lane == 0 && blockIdx.x < 8192) {
            unsigned int* q = g_prof + ((int64_t)blockIdx.x * 4 + w) * 8;
            q[0] = (unsigned int)pt_start;
            q[1] = (unsigned int)pt_end;
            q[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID: wave 3:0, simd 5:4, cu 11:8, sh 12, se 15:13, tg 19:16
            q[3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
            q[4] = (unsigned int)pt_bar;
            q[5] = (unsigned int)pt_loop;
            q[6] = (unsigned int)(pt_start >> 32);
            q[7] = (unsigned int)pt_s3;
        }
        (void)pt_first; (void)pt_s2;
#elif defined(VM_EXPERIMENT_PROFILE_EPI)
        if (lane == 0 && blockIdx.x < 8192) {   // raw stamps: slots 1..6 were written by the epilogue (VM_EPI_MARK)
            unsigned int* q = g_prof + ((int64_t)blockIdx.x * 4 + w) * 8;
            q[0] = (unsigned int)pt_start;
            q[7] = (unsigned int)pt_end;
        }
        (void)pt_first; (void)pt_bar; (void)pt_s2; (void)pt_s3; (void)pt_loop;
#else
        if (lane == 0 && blockIdx.x < 8192) {
            unsigned int* q = g_prof + ((int64_t)blockIdx.x * 4 + w) * 8;
            q[0] = (unsigned int)(pt_end - pt_start);    // the whole wave-tile (incl. the drain of its stores)
            q[1] = (unsigned int)(pt_first - pt_start);  // start -> end of the first K tile
            q[2] = (unsigned int)(pt_loop - pt_first);   // the other nk - 1 K tiles
            q[3] = (unsigned int)(pt_end - pt_loop);     // epilogue + store drain
            q[4] = (unsigned int)(pt_s2 - pt_start);     // tile coordinates, bias loads, DMA / fragment addresses
            q[5] = (unsigned int)(pt_s3 - pt_s2);        // issue of the prologue DMA / weight loads
            q[6] = (unsigned int)(pt_bar - pt_s3);       // accumulator init, first data wait, first barrier
            q[7] = (unsigned int)(pt_first - pt_bar);    // fragment reads + 16 MFMAs of the first K tile
        }
#endif
    }
#endif
}

// (N, 3 * a_c) row-major GEMM-layout weights (vm_prep_conv_weights' wf / wd, vm_fold_bn_weights' wf_folded; `towers` of them back to
// back) -> the fragment order conv_nt3_kernel streams: one thread per 16-byte piece of the OUTPUT (coalesced writes; the reads are
// 16-byte gathers out of L2).  out piece index = ((((t * N/64 + b64) * nk + kt) * 2 + j) * 2 + ks) * 64 + lane.
constexpr int PACK_MAX = 8;
struct PackBatch {   // several matrices in one launch (blockIdx.y): the wd copies of an encoder after every optimizer step
    const void* bt[PACK_MAX];
    void* out[PACK_MAX];
    int towers[PACK_MAX], rows[PACK_MAX], a_c[PACK_MAX];
};
template <typename T>
__global__ __launch_bounds__(256) void pack_nt_weights_kernel(PackBatch pb) {
    const int m = blockIdx.y;
    const T* __restrict__ bt = (const T*)pb.bt[m];
    T* __restrict__ out = (T*)pb.out[m];
    const int towers = pb.towers[m], N = pb.rows[m], a_c = pb.a_c[m];
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int nk = 3 * (a_c / 32);
    const int64_t total = (int64_t)towers * N * 3 * a_c / 8;
    if (o >= total) return;
    const int lane = (int)(o & 63);
    int64_t q = o >> 6;
    const int ks = (int)(q & 1);
    q >>= 1;
    const int j = (int)(q & 1);
    q >>= 1;
    const int kt = (int)(q % nk);
    q /= nk;
    const int b64 = (int)(q % (N / 64));
    const int t = (int)(q / (N / 64));
    const int nrow = b64 * 64 + j * 32 + (lane & 31);
    const int chunk = kt / 3, tap = kt - 3 * chunk;
    const int col = tap * a_c + chunk * 32 + ks * 16 + (lane >> 5) * 8;
    const u32x4 v = *reinterpret_cast<const u32x4*>(bt + ((int64_t)t * N + nrow) * (3 * (int64_t)a_c) + col);
    *reinterpret_cast<u32x4*>(out + o * 8) = v;
}

#if defined(VM_EXPERIMENT_PROFILE)
extern "C" int vm_debug_prof_read(unsigned int* out, int n_slots) {  // out: n_slots x 4 host values
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned int) * 8 * (size_t)n_slots);
    return 0;
}
#endif

}  // namespace vm

using namespace vm;

static int tiles(int64_t x, int t) { return (int)((x + t - 1) / t); }

extern "C" int64_t vm_conv_stat_rows(int64_t L) { return (L + BM - 1) / BM; }

// ---- kernel selection.  Every selectable kernel computes the same result (each is parity-tested on every shape it serves); the
// switches exist so that the tests can pin the fallback kernels on shapes the default dispatch would give to conv_nt2r_kernel ----
namespace vm {
int g_nt_n2 = 3;       // conv_nt2r_kernel for 16-bit storage: bit 0 forward, bit 1 dgrad; vm_set_tuning("nt_n2", 0..3)
int g_nt3 = 3;         // conv_nt3_kernel (weights L2 -> registers) where the caller supplies packed weights: bit 0 forward, bit 1 dgrad
int g_nt3_wide = 0;    // conv_nt3_kernel's 256 x 32 wave tile (half the weight stream, twice the fragment reads): bit 0 forward, bit 1 dgrad
int g_nt3_lean = 3;    // conv_nt3_kernel's lean prologue (A(2), A(3) requested under the first two K tiles): bit 0 forward, bit 1 dgrad
int g_nt_glds = 1;     // the LDS-DMA 128^2 kernel where K * sizeof(T) % 64 == 0, else register staging; vm_set_tuning("nt_glds", 0 | 1)
int g_nt_blocks = 512; // persistent grid of the 128^2 kernels (2 workgroups per CU on 256 CUs)
extern int g_tn_x, g_tn_tile, g_tn9, g_tn9_stages;  // conv_wgrad.hip
extern int g_fuse_finalize, g_apply_order;          // bnpool.hip
extern int g_f1_products;            // conv1_fused.hip
}  // namespace vm

static bool is16(int dtype) { return dtype == VM_BF16 || dtype == VM_F16; }

// Does conv_nt2r_kernel take (L, K-side channels ck, N-side channels n)?  Short windows (the 2-D variant runs L = 298 .. 37) whose
// 256-row tiles would be mostly padding stay with the 128-row kernels; a forward that also emits BatchNorm statistics needs its two
// partial rows per 254-position tile to be exactly the ceil(L / 128) rows of vm_conv_stat_rows (asked for the inference launch of
// the same layer as well, so that a layer runs the same kernel -- the same summation order -- in both modes).
static bool n2r_shape(int64_t n_windows, int64_t L, int ck, int n, bool stats_layout) {
    if (n % n2::TN != 0 || ck % 32 != 0 || L <= 0 || n_windows <= 0) return false;
    const double u256 = (double)L / (256.0 * ((L + 255) / 256)), u128 = (double)L / (128.0 * ((L + 127) / 128));
    if (u256 + 0.10 < u128) return false;
    const int64_t t254 = (L + n2r::TROWS - 1) / n2r::TROWS;
    if (stats_layout && 2 * t254 != (L + 127) / 128) return false;
    return n_windows * t254 * (n / n2::TN) < (1LL << 31);
}

static bool nt3_chunks(int a_c) { return a_c == 128 || a_c == 256 || a_c == 384 || a_c == 512; }

template <typename T, int EPI>
static void launch_n2r(const NtArgs<T>& a, int64_t n_windows, hipStream_t stream) {
    if constexpr (sizeof(T) == 2) {
        NtArgs<T> b = a;
        b.tilesN = a.N / n2::TN;
        b.tilesL = (a.L + n2r::TROWS - 1) / n2r::TROWS;
        const int64_t n_groups = n_windows * b.tilesL;
        if constexpr (EPI != EPI_FWD) {  // the entry points that take packed weights (vm_pack_nt_weights)
            if (b.bt_packed != nullptr && (g_nt3 & (EPI == EPI_DGRAD ? 2 : 1)) && nt3_chunks(a.a_c)) {
                const dim3 grid((unsigned)(n_groups * b.tilesN));
#define VM_NT3(CH, PIPE) hipLaunchKernelGGL((conv_nt3_kernel<T, EPI, CH, PIPE>), grid, dim3(256), 0, stream, b, n_groups)
                if constexpr (EPI == EPI_FWD_FOLD || EPI == EPI_DGRAD) {   // the 256 x 32 wave tile (experiment; lean prologue only)
                    if (g_nt3_wide & (EPI == EPI_DGRAD ? 2 : 1)) {
#define VM_NT3W(CH) hipLaunchKernelGGL((conv_nt3_kernel<T, EPI, CH, true, true>), grid, dim3(256), 0, stream, b, n_groups)
                        switch (a.a_c / 32) {
                            case 4: VM_NT3W(4); break;
                            case 8: VM_NT3W(8); break;
                            case 12: VM_NT3W(12); break;
                            default: VM_NT3W(16); break;
                        }
#undef VM_NT3W
                        return;
                    }
                }
                const bool pipe = (g_nt3_lean & (EPI == EPI_DGRAD ? 2 : 1)) != 0;
                switch (a.a_c / 32) {  // the K loop is written out per channel count: 128, 256, 384, 512 channels on the K side
                    case 4: if (pipe) VM_NT3(4, true); else VM_NT3(4, false); break;
                    case 8: if (pipe) VM_NT3(8, true); else VM_NT3(8, false); break;
                    case 12: if (pipe) VM_NT3(12, true); else VM_NT3(12, false); break;
                    default: if (pipe) VM_NT3(16, true); else VM_NT3(16, false); break;
                }
#undef VM_NT3
                return;
            }
        }
        hipLaunchKernelGGL((conv_nt2r_kernel<T, EPI>), dim3((unsigned)(n_groups * b.tilesN)), dim3(256), 0, stream, b, n_groups);
    }
}

template <typename T, int EPI>
static void launch_nt(const NtArgs<T>& a, int64_t n_windows, hipStream_t stream) {
    if constexpr (sizeof(T) == 2) {
        if (!a.flat_period && (g_nt_n2 & (EPI == EPI_DGRAD ? 2 : 1)) && a.Ktot == 3 * a.a_c && n2r_shape(n_windows, a.L, a.a_c, a.N, EPI == EPI_FWD || EPI == EPI_FWD_FOLD)) {
            launch_n2r<T, EPI>(a, n_windows, stream);
            return;
        }
    }
    const int64_t n_groups = n_windows * a.tilesL;
    const int64_t grid = n_groups < g_nt_blocks ? n_groups : g_nt_blocks;
    const int64_t kbytes = (int64_t)a.Ktot * (int64_t)sizeof(T);
    if constexpr (sizeof(T) == 4) {
        if (a.split) {
            hipLaunchKernelGGL((conv_nt_kernel<T, EPI, 128, true>), dim3((unsigned)grid), dim3(256), 0, stream, a, n_groups);
            return;
        }
    }
    if (g_nt_glds && kbytes % 128 == 0) {
        hipLaunchKernelGGL((conv_nt_glds_kernel<T, EPI, 128>), dim3((unsigned)grid), dim3(256), 0, stream, a, n_groups);
    } else if (g_nt_glds && kbytes % 64 == 0) {
        hipLaunchKernelGGL((conv_nt_glds_kernel<T, EPI, 64>), dim3((unsigned)grid), dim3(256), 0, stream, a, n_groups);
    } else {
        hipLaunchKernelGGL((conv_nt_kernel<T, EPI, 128>), dim3((unsigned)grid), dim3(256), 0, stream, a, n_groups);
    }
}

template <typename T>
static NtArgs<T> fwd_args(const void* in, const void* wf, const float* bias, void* out, float* stat_sum, float* stat_sq, int64_t L,
                          int c_in, int c_out, int dtype) {
    NtArgs<T> a;
    a.a = (const T*)in;
    a.bt = (const T*)wf;
    a.bias = bias;
    a.out = (T*)out;
    a.stat_sum = stat_sum;
    a.stat_sq = stat_sq;
    a.a_win_stride = (L + 2) * (int64_t)c_in;
    a.a_c = c_in;
    a.L = (int)L;
    a.N = c_out;
    a.Ktot = 3 * c_in;
    a.tilesL = tiles(L, BM);
    a.tilesN = tiles(c_out, BN);
    a.split = dtype == VM_F32S;
    return a;
}

template <typename T>
static NtArgs<T> dgrad_args(const void* du, const void* wd, void* dx, int64_t L, int c_in, int c_out, int dtype) {
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
    a.split = dtype == VM_F32S;
    return a;
}

extern "C" int vm_conv_fwd(const void* in, const void* wf, const float* bias, int64_t n_windows, int64_t L, int c_in,
                           int c_out, int dtype, void* z, float* stat_sum, float* stat_sq, void* stream) {
    VM_REQUIRE(in && wf && bias && z, "vm_conv_fwd: null pointer");
    VM_REQUIRE(n_windows > 0 && L > 0 && c_in > 0 && c_out > 0, "vm_conv_fwd: bad sizes");
    VM_REQUIRE(c_in % 8 == 0 && c_out % 8 == 0, "vm_conv_fwd: channels must be multiples of 8 (got %d, %d)", c_in, c_out);
    VM_REQUIRE((stat_sum == nullptr) == (stat_sq == nullptr), "vm_conv_fwd: stat_sum/stat_sq must both be set or NULL");
    VM_REQUIRE((L + 2) * (int64_t)c_in < (1LL << 31) && 3LL * c_in * c_out < (1LL << 31), "vm_conv_fwd: window too large");
    VM_DISPATCH_DTYPE(dtype, {
        launch_nt<T, EPI_FWD>(fwd_args<T>(in, wf, bias, z, stat_sum, stat_sq, L, c_in, c_out, dtype), n_windows, (hipStream_t)stream);
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
        launch_nt<T, EPI_DGRAD>(dgrad_args<T>(du, wd, dx, L, c_in, c_out, dtype), n_windows, (hipStream_t)stream);
    });
    return check_launch("vm_conv_dgrad");
}

// ---- vm_conv_fwd over windows too short for a tile: their concatenation (every window carries its own zero halo rows) run as one
// sequence on the 128-row kernels, the halo positions dropped by the epilogue.  Same z, bit for bit; the statistics rows are per
// 128-row tile of the concatenation (vm_conv_flat_stat_rows rows in all) instead of per window ----
extern "C" int64_t vm_conv_flat_stat_rows(int64_t n_windows, int64_t L) { return (n_windows * (L + 2) - 2 + BM - 1) / BM; }

extern "C" int vm_conv_fwd_flat(const void* in, const void* wf, const float* bias, int64_t n_windows, int64_t L, int c_in, int c_out,
                                int dtype, void* z, float* stat_sum, float* stat_sq, void* stream) {
    VM_REQUIRE(in && wf && bias && z, "vm_conv_fwd_flat: null pointer");
    VM_REQUIRE(n_windows > 0 && L > 0 && c_in > 0 && c_out > 0, "vm_conv_fwd_flat: bad sizes");
    VM_REQUIRE(c_in % 8 == 0 && c_out % 8 == 0, "vm_conv_fwd_flat: channels must be multiples of 8 (got %d, %d)", c_in, c_out);
    VM_REQUIRE((stat_sum == nullptr) == (stat_sq == nullptr), "vm_conv_fwd_flat: stat_sum/stat_sq must both be set or NULL");
    const int64_t Lf = n_windows * (L + 2) - 2;
    VM_REQUIRE((Lf + 2) * (int64_t)(c_in > c_out ? c_in : c_out) < (1LL << 31) && 3LL * c_in * c_out < (1LL << 31),
               "vm_conv_fwd_flat: the concatenated windows exceed 2^31 elements (launch fewer windows per call)");
    VM_DISPATCH_DTYPE(dtype, {
        NtArgs<T> a = fwd_args<T>(in, wf, bias, z, stat_sum, stat_sq, Lf, c_in, c_out, dtype);
        a.flat_period = (int)(L + 2);
        a.flat_valid = (int)L;
        launch_nt<T, EPI_FWD>(a, 1, (hipStream_t)stream);
    });
    return check_launch("vm_conv_fwd_flat");
}

// ---- training forward that also emits the pool-window extreme (conv_nt2r_kernel only) ----
extern "C" int vm_conv_fwd_e_supported(int64_t n_windows, int64_t L, int c_in, int c_out, int dtype) {
    return (is16(dtype) && (g_nt_n2 & 1) && L >= 2 && !(L & 1) && n2r_shape(n_windows, L, c_in, c_out, true)) ? 1 : 0;
}

extern "C" int vm_conv_fwd_e(const void* in, const void* wf, const float* bias, const float* gamma, int64_t n_windows, int64_t L,
                             int c_in, int c_out, int dtype, void* z, float* stat_sum, float* stat_sq, void* e, void* stream) {
    VM_REQUIRE(in && wf && bias && gamma && z && stat_sum && stat_sq && e, "vm_conv_fwd_e: null pointer");
    VM_REQUIRE(n_windows > 0 && L > 0, "vm_conv_fwd_e: bad sizes");
    VM_REQUIRE((L + 2) * (int64_t)c_in < (1LL << 31) && 3LL * c_in * c_out < (1LL << 31), "vm_conv_fwd_e: window too large");
    if (!vm_conv_fwd_e_supported(n_windows, L, c_in, c_out, dtype)) {
        set_error("vm_conv_fwd_e: shape/dtype/tuning not served by the 256 x 128 input-resident kernel (ask vm_conv_fwd_e_supported)");
        return VM_ERR_UNSUPPORTED;
    }
    VM_DISPATCH_16(dtype, {
        NtArgs<T> a = fwd_args<T>(in, wf, bias, z, stat_sum, stat_sq, L, c_in, c_out, dtype);
        a.aff_scale = gamma;
        a.pool_e = (T*)e;
        launch_n2r<T, EPI_FWD>(a, n_windows, (hipStream_t)stream);
    });
    return check_launch("vm_conv_fwd_e");
}

// ---- training forward over the pool extreme of the layer below, that layer's BatchNorm affine folded into the weights
// (vm_fold_bn_weights makes wf_folded and hb; conv_nt2r_kernel only).  e (optional): this layer's own pool extreme, PADDED ----
extern "C" int vm_conv_fwd_fold_supported(int64_t n_windows, int64_t L, int c_in, int c_out, int dtype, int with_e) {
    if (!(is16(dtype) && (g_nt_n2 & 1) && n2r_shape(n_windows, L, c_in, c_out, true))) return 0;
    return (!with_e || (L >= 2 && !(L & 1))) ? 1 : 0;
}

extern "C" int vm_conv_fwd_fold(const void* in_e, const void* wf_folded, const float* bias, const float* hb, const float* gamma,
                                int64_t n_windows, int64_t windows_per_tower, int64_t L, int c_in, int c_out, int dtype, void* z,
                                float* stat_sum, float* stat_sq, void* e, void* o, const void* wf_packed, const float* e_center,
                                void* stream) {
    VM_REQUIRE(in_e && wf_folded && bias && hb && stat_sum && stat_sq, "vm_conv_fwd_fold: null pointer");
    VM_REQUIRE(e_center == nullptr || (dtype == VM_F16 && e != nullptr && o != nullptr),
               "vm_conv_fwd_fold: a centred tile (e_center) needs VM_F16 storage and the (e, o) pair output");
    VM_REQUIRE(e == nullptr || gamma != nullptr, "vm_conv_fwd_fold: the pool extreme needs gamma (its sign picks max / min)");
    VM_REQUIRE(o == nullptr || e != nullptr, "vm_conv_fwd_fold: o (the other element of each pair) goes with e");
    VM_REQUIRE(z != nullptr || o != nullptr, "vm_conv_fwd_fold: z may only be NULL when (e, o) carry the output");
    VM_REQUIRE(n_windows > 0 && L > 0 && windows_per_tower > 0 && n_windows % windows_per_tower == 0,
               "vm_conv_fwd_fold: n_windows must be a positive multiple of windows_per_tower");
    VM_REQUIRE((L + 2) * (int64_t)c_in < (1LL << 31) && 3LL * c_in * c_out < (1LL << 31), "vm_conv_fwd_fold: window too large");
    if (!vm_conv_fwd_fold_supported(n_windows, L, c_in, c_out, dtype, e != nullptr)) {
        set_error("vm_conv_fwd_fold: shape/dtype/tuning not served by the 256 x 128 input-resident kernel (ask vm_conv_fwd_fold_supported)");
        return VM_ERR_UNSUPPORTED;
    }
    VM_DISPATCH_16(dtype, {
        NtArgs<T> a = fwd_args<T>(in_e, wf_folded, bias, z, stat_sum, stat_sq, L, c_in, c_out, dtype);
        a.fold_hb = hb;
        a.tower_windows = windows_per_tower;
        a.bt_tower_stride = 3LL * c_in * c_out;
        a.aff_scale = gamma;
        a.pool_e = (T*)e;
        a.pool_o = (T*)o;
        a.pool_e_pad = 1;
        a.bt_packed = (const T*)wf_packed;
        a.fold_ctr = e_center;
        launch_n2r<T, EPI_FWD_FOLD>(a, n_windows, (hipStream_t)stream);
    });
    return check_launch("vm_conv_fwd_fold");
}

// ---- inference forward with BatchNorm affine + MaxPool1D(2) in the epilogue (conv_nt2r_kernel only) ----
extern "C" int vm_conv_fwd_pool_supported(int64_t n_windows, int64_t L, int c_in, int c_out, int dtype) {
    return (is16(dtype) && (g_nt_n2 & 1) && L >= 2 && !(L & 1) && n2r_shape(n_windows, L, c_in, c_out, false)) ? 1 : 0;
}

extern "C" int vm_conv_fwd_pool(const void* in, const void* wf, const float* bias, const float* scale, const float* shift,
                                int64_t n_windows, int64_t L, int c_in, int c_out, int dtype, void* act, const void* wf_packed,
                                void* stream) {
    VM_REQUIRE(in && wf && bias && scale && shift && act, "vm_conv_fwd_pool: null pointer");
    VM_REQUIRE(n_windows > 0 && L > 0, "vm_conv_fwd_pool: bad sizes");
    VM_REQUIRE((L + 2) * (int64_t)c_in < (1LL << 31) && 3LL * c_in * c_out < (1LL << 31), "vm_conv_fwd_pool: window too large");
    if (!vm_conv_fwd_pool_supported(n_windows, L, c_in, c_out, dtype)) {
        set_error("vm_conv_fwd_pool: shape/dtype/tuning not served by the 256 x 128 input-resident kernel (ask vm_conv_fwd_pool_supported)");
        return VM_ERR_UNSUPPORTED;
    }
    VM_DISPATCH_16(dtype, {
        NtArgs<T> a = fwd_args<T>(in, wf, bias, act, nullptr, nullptr, L, c_in, c_out, dtype);
        a.aff_scale = scale;
        a.aff_shift = shift;
        a.bt_packed = (const T*)wf_packed;
        launch_n2r<T, EPI_FWD_POOL>(a, n_windows, (hipStream_t)stream);
    });
    return check_launch("vm_conv_fwd_pool");
}

// ---- dgrad with the BatchNorm-backward partial sums of the layer below fused into its epilogue (conv_nt2r_kernel only) ----
extern "C" int64_t vm_conv_dgrad_bnred_rows(int64_t L) { return 2 * ((L + n2r::TROWS - 1) / n2r::TROWS); }

extern "C" int vm_conv_dgrad_bnred_supported(int64_t n_windows, int64_t L, int c_in, int c_out, int dtype) {
    return (is16(dtype) && (g_nt_n2 & 2) && n2r_shape(n_windows, L, c_out, c_in, false)) ? 1 : 0;
}

extern "C" int vm_conv_dgrad_bnred(const void* du, const void* wd, int64_t n_windows, int64_t L, int c_in, int c_out, int dtype,
                                   void* dx, const void* red_a, int red_a_padded, float* red_s0, float* red_s1, const void* wd_packed,
                                   void* stream) {
    VM_REQUIRE(du && wd && dx && red_a && red_s0 && red_s1, "vm_conv_dgrad_bnred: null pointer");
    VM_REQUIRE(n_windows > 0 && L > 0, "vm_conv_dgrad_bnred: bad sizes");
    VM_REQUIRE((L + 2) * (int64_t)c_out < (1LL << 31) && 3LL * c_in * c_out < (1LL << 31), "vm_conv_dgrad_bnred: window too large");
    if (!vm_conv_dgrad_bnred_supported(n_windows, L, c_in, c_out, dtype)) {
        set_error("vm_conv_dgrad_bnred: shape/dtype/tuning not served by the 256 x 128 input-resident kernel (ask vm_conv_dgrad_bnred_supported)");
        return VM_ERR_UNSUPPORTED;
    }
    VM_DISPATCH_16(dtype, {
        NtArgs<T> a = dgrad_args<T>(du, wd, dx, L, c_in, c_out, dtype);
        a.stat_sum = red_s0;
        a.stat_sq = red_s1;
        a.red_a = (const T*)red_a;
        a.red_a_win_stride = (L + (red_a_padded ? 2 : 0)) * (int64_t)c_in;
        a.red_a_row0 = red_a_padded ? 1 : 0;
        a.bt_packed = (const T*)wd_packed;
        launch_n2r<T, EPI_DGRAD>(a, n_windows, (hipStream_t)stream);
    });
    return check_launch("vm_conv_dgrad_bnred");
}

// ---- (rows, 3 * a_c) GEMM-layout weights -> the fragment order of conv_nt3_kernel (same bytes, permuted) ----
extern "C" int vm_pack_nt_weights_supported(int n_rows, int a_c, int dtype) {
    return (is16(dtype) && n_rows > 0 && a_c > 0 && n_rows % n2::TN == 0 && a_c % 32 == 0) ? 1 : 0;
}

extern "C" int vm_pack_nt_weights(const void* bt, int towers, int n_rows, int a_c, int dtype, void* packed, void* stream) {
    VM_REQUIRE(bt && packed, "vm_pack_nt_weights: null pointer");
    VM_REQUIRE(towers > 0, "vm_pack_nt_weights: bad sizes");
    if (!vm_pack_nt_weights_supported(n_rows, a_c, dtype)) {
        set_error("vm_pack_nt_weights: 16-bit storage, rows %% 128 == 0 and channels %% 32 == 0 only (ask vm_pack_nt_weights_supported)");
        return VM_ERR_UNSUPPORTED;
    }
    const int64_t pieces = (int64_t)towers * n_rows * 3 * a_c / 8;
    PackBatch pb;
    pb.bt[0] = bt;
    pb.out[0] = packed;
    pb.towers[0] = towers;
    pb.rows[0] = n_rows;
    pb.a_c[0] = a_c;
    VM_DISPATCH_16(dtype, {
        hipLaunchKernelGGL((pack_nt_weights_kernel<T>), dim3((unsigned)cdiv(pieces, 256), 1), dim3(256), 0, (hipStream_t)stream, pb);
    });
    return check_launch("vm_pack_nt_weights");
}

extern "C" int vm_pack_nt_weights_batch(int n, const void* const* bt, const int* towers, const int* n_rows, const int* a_c, int dtype,
                                        void* const* packed, void* stream) {
    VM_REQUIRE(bt && towers && n_rows && a_c && packed, "vm_pack_nt_weights_batch: null pointer");
    VM_REQUIRE(n > 0 && n <= PACK_MAX, "vm_pack_nt_weights_batch: 1..%d matrices per call (got %d)", PACK_MAX, n);
    PackBatch pb;
    int64_t most = 0;
    for (int i = 0; i < n; ++i) {
        VM_REQUIRE(bt[i] && packed[i] && towers[i] > 0, "vm_pack_nt_weights_batch: bad matrix %d", i);
        if (!vm_pack_nt_weights_supported(n_rows[i], a_c[i], dtype)) {
            set_error("vm_pack_nt_weights_batch: matrix %d: 16-bit storage, rows %% 128 == 0 and channels %% 32 == 0 only", i);
            return VM_ERR_UNSUPPORTED;
        }
        pb.bt[i] = bt[i];
        pb.out[i] = packed[i];
        pb.towers[i] = towers[i];
        pb.rows[i] = n_rows[i];
        pb.a_c[i] = a_c[i];
        const int64_t pieces = (int64_t)towers[i] * n_rows[i] * 3 * a_c[i] / 8;
        most = pieces > most ? pieces : most;
    }
    VM_DISPATCH_16(dtype, {
        hipLaunchKernelGGL((pack_nt_weights_kernel<T>), dim3((unsigned)cdiv(most, 256), (unsigned)n), dim3(256), 0, (hipStream_t)stream, pb);
    });
    return check_launch("vm_pack_nt_weights_batch");
}

// ---- measurement aid: the dense 16-bit MFMA rate this device SUSTAINS from registers.  The cfg-A step runs at the package power
// limit (profiles/r06_kernel_power.txt: 1400 W under every GEMM launch, clocks 1.47-1.86 GHz), so the 2.5 PFLOP/s of the 2.4 GHz
// nominal clock is not a rate the part holds; this launch -- every SIMD of every CU issuing v_mfma_f32_32x32x16 back to back on
// non-zero operands, nothing else -- is what bench.py times beside the step to say how far the GEMM launches are from THAT.
template <typename T>
__global__ __launch_bounds__(256, 2) void mfma_rate_kernel(float* __restrict__ sink, int iters) {
    using V8 = typename Mfma<T>::Frag;
    const int lane = threadIdx.x & 63;
    u32x4 ra, rb;   // dense, lane-dependent operands of ordinary magnitude (zero operands draw less power and clock higher)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t h = (uint32_t)(lane * 2654435761u + k * 40503u + blockIdx.x * 97u);
        // two 16-bit values per word, exponent field fixed near 1.0 (0x3C00 / 0x3F80 region), random sign and fraction bits
        const uint32_t base = std::is_same<T, bf16>::value ? 0x3F003F00u : 0x38003800u;
        ra[k] = base | (h & 0x80FF80FFu);
        rb[k] = base | ((h >> 3) & 0x80FF80FFu);
    }
    const V8 a = __builtin_bit_cast(V8, ra), b = __builtin_bit_cast(V8, rb);
    f32x16 acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = Mfma<T>::run(a, b, acc[k]);
    }
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += acc[k][0] + acc[k][7] + acc[k][15];
    if (t == 123.456f) sink[blockIdx.x * 256 + threadIdx.x] = t;   // never true: keeps the chain alive
}

constexpr int MFMA_RATE_BLOCKS = 512;   // two 4-wave workgroups per CU: two waves per SIMD, like the GEMM kernels

extern "C" int64_t vm_mfma_rate_probe_flops(int iters) {
    return (int64_t)MFMA_RATE_BLOCKS * 4 * (int64_t)iters * 8 * (2LL * 32 * 32 * 16);
}

extern "C" int vm_mfma_rate_probe(int dtype, int iters, float* sink, void* stream) {
    VM_REQUIRE(sink && iters > 0, "vm_mfma_rate_probe: bad argument (sink: %d floats)", MFMA_RATE_BLOCKS * 256);
    VM_REQUIRE(dtype == VM_BF16 || dtype == VM_F16, "vm_mfma_rate_probe: VM_BF16 or VM_F16, got dtype %d", dtype);
    if (dtype == VM_BF16) {
        hipLaunchKernelGGL((mfma_rate_kernel<bf16>), dim3(MFMA_RATE_BLOCKS), dim3(256), 0, (hipStream_t)stream, sink, iters);
    } else {
        hipLaunchKernelGGL((mfma_rate_kernel<f16>), dim3(MFMA_RATE_BLOCKS), dim3(256), 0, (hipStream_t)stream, sink, iters);
    }
    return check_launch("vm_mfma_rate_probe");
}

// Kernel-selection hook for the tests and A/B measurements (not part of the drop-in surface): returns 0 if the key/value is known.
extern "C" int vm_set_tuning(const char* key, int value) {
    struct Knob { const char* key; int* var; int lo, hi; };
    static const Knob knobs[] = {{"nt_n2", &g_nt_n2, 0, 3}, {"nt3", &g_nt3, 0, 3}, {"nt3_lean", &g_nt3_lean, 0, 3}, {"nt3_wide", &g_nt3_wide, 0, 3}, {"nt_glds", &g_nt_glds, 0, 1}, {"tn_x", &g_tn_x, 0, 1}, {"tn9", &g_tn9, 0, 2}, {"tn9_stages", &g_tn9_stages, 0, 1}, {"fuse_finalize", &g_fuse_finalize, 0, 31}, {"apply_order", &g_apply_order, 0, 2}, {"f1_products", &g_f1_products, 1, 3}};
    if (key == nullptr) {
        set_error("vm_set_tuning: null key");
        return VM_ERR_ARG;
    }
    for (const Knob& k : knobs) {
        if (strcmp(key, k.key) == 0) {
            VM_REQUIRE(value >= k.lo && value <= k.hi, "vm_set_tuning: %s out of range [%d, %d]", key, k.lo, k.hi);
            *k.var = value;
            return VM_OK;
        }
    }
    if (strcmp(key, "tn_tile") == 0 && (value == 128 || value == 256)) {
        g_tn_tile = value;
        return VM_OK;
    }
    if (strcmp(key, "f1_fwd_blocks") == 0 && value > 0) return f1_set_fwd_blocks(value);
    if (strcmp(key, "f1_blocks") == 0 && value > 0) return f1_set_blocks(value);
    set_error("vm_set_tuning: unknown key/value");
    return VM_ERR_ARG;
}
