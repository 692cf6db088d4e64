// Probe for the round-2 forward/dgrad GEMM structure: ONE wave per SIMD, 128x128 per wave.
//   C[M][N] (bf16) = A[M][K] * B[N][K]^T, bf16 operands, fp32 accumulate; 256 x 256 workgroup tile, 4 waves (2 x 2),
//   K tiles of 64 staged by LDS-DMA into two 64 KB stages, one s_barrier per K tile, fragments double-buffered in
//   registers (read one k-step ahead), 16 v_mfma_f32_32x32x16_bf16 per k-step and wave.
// It is a correct GEMM (checked against the host on sampled rows, integer-valued data => exact) and is timed against the
// sustained MFMA rate.   hipcc --offload-arch=gfx950 -O3 gemm_w4_probe.hip -o gemm_w4_probe && ./gemm_w4_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

constexpr int TM = 256, TN = 256, KT = 64;  // workgroup tile, K tile (elements)
constexpr int ROWB = KT * 2;                 // 128 bytes of K per LDS row
constexpr int OPB = 256 * ROWB;              // one operand of one stage: 32 KB
constexpr int NA = 3, NB = 2;                // A stages (streamed from HBM: two K tiles of lead), B stages (weights: L2)
constexpr int B0 = NA * OPB;                 // B stages start here
constexpr int LDS_BYTES = (NA + NB) * OPB;   // 160 KB: one workgroup per CU

__device__ inline void glds16(const char* gsrc, char* lds_base) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds(gsrc, (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
#endif
}

struct Args {
    const bf16* a;
    const bf16* b;
    bf16* c;
    int M, N, K;
    int tilesN;
    unsigned long long* stall;  // MODE & 64: per wave {prologue wait, sum of vmcnt waits, sum of barrier waits, whole kernel}
};

template <int MODE>
__global__ __attribute__((amdgpu_flat_work_group_size(256, 256), amdgpu_waves_per_eu(1, 1))) void gemm_w4(Args p) {
    __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    // MODE & 16: XCD-aware order -- workgroup b runs on XCD b % 8; consecutive workgroups OF ONE XCD take the column tiles of
    // one row block, so the second one finds the A rows in that XCD's L2
    int tn, tm;
    if (MODE & 16) {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        tn = __builtin_amdgcn_readfirstlane(j % p.tilesN);
        tm = __builtin_amdgcn_readfirstlane((j / p.tilesN) * 8 + x);
        if (tm * TM >= p.M) return;
    } else {
        tn = __builtin_amdgcn_readfirstlane((int)blockIdx.x % p.tilesN);
        tm = __builtin_amdgcn_readfirstlane((int)blockIdx.x / p.tilesN);
    }
    const int nk = p.K / KT;
    const int pitch = p.K * 2;

    // ---- DMA geometry: a piece = 8 rows x 128 B (lane-linear in LDS); wave w owns rows [64w, 64w+64) of A and of B ----
    // LDS row r keeps 16-byte chunk c at position c ^ ((r >> 1) & 7): the source chunk of a lane is permuted accordingly
    const int prow = lane >> 3;                                  // row within a piece
    const char* a_base = reinterpret_cast<const char*>(p.a) + (int64_t)(((MODE & 8) ? 0 : tm * TM) + w * 64) * pitch;
    const char* b_base = reinterpret_cast<const char*>(p.b) + (int64_t)(tn * TN + w * 64) * pitch;
    // lane offset inside a piece for piece index q (rows 8q .. 8q+7 of the wave's 64): key = ((8q + prow) >> 1) & 7 = (4q + (prow >> 1)) & 7
    auto lane_off = [&](int q) { return (unsigned)((q * 8 + prow) * pitch + (((lane & 7) ^ ((4 * q + (prow >> 1)) & 7)) * 16)); };
    unsigned loff[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) loff[q] = lane_off(q);
    auto stage_piece = [&](int stage, int kt, int q, int operand) {  // stage: A stage 0..NA-1 or B stage 0..NB-1
        char* dst = lds + (operand ? B0 : 0) + stage * OPB + (w * 64 + q * 8) * ROWB;
        const char* src = (operand ? b_base : a_base) + (int64_t)kt * ROWB;
        glds16(src + loff[q], dst);
    };

    // ---- fragment geometry: lane (r = lane & 31, kh = lane >> 5) reads 16 B of row r, k-chunk 2s + kh ----
    const int r = lane & 31, kh = lane >> 5;
    int foff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) foff[s] = r * ROWB + (((2 * s + kh) ^ ((r >> 1) & 7)) * 16);  // 32-row blocks are 16-row aligned
    const int a_rows = wm * 128 * ROWB, b_rows = wn * 128 * ROWB;
    struct Frag {
        bf16x8 a[4], b[4];
    };
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // One k-step = 16 MFMAs, issued in order; after MFMA t one "filler" is issued in its shadow (<= 4-5 single-issue
    // instructions per MFMA are free: tools/probe/mfma_valu_probe): the 8 fragment reads of the NEXT k-step in slots 0-7
    // (rslot) and up to 8 DMA pieces of the next K tile in slots 8-15.  sched_barrier(0) freezes that order.
    auto one_read = [&](Frag& f, int sa, int sb, int s, int t) {  // t in 0..7: a[0..3], b[0..3]; sa / sb: A / B stage
        if (MODE & 2) return;
        const char* base = lds + (t < 4 ? sa * OPB + a_rows : B0 + sb * OPB + b_rows) + foff[s] + (t & 3) * 32 * ROWB;
        if (t < 4) f.a[t] = *reinterpret_cast<const bf16x8*>(base); else f.b[t - 4] = *reinterpret_cast<const bf16x8*>(base);
    };
    unsigned long long st_wait = 0, st_bar = 0;
    // DMA of one K tile: B pieces of K tile kt+1 first, then A pieces of K tile kt+2, so that the counted wait vmcnt(8) in
    // the middle of k-step 3 covers B(kt+1) and the older A(kt+1) while the 8 newest pieces, A(kt+2), stay in flight.
    auto kstep = [&](const Frag& cur, Frag& nxt, int rsa, int rsb, int rs, int rslot0, int dma_op, int dstage, int dkt, bool sync_mid, int dpiece0 = 0) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int i = t >> 2, j = t & 3;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.b[j], cur.a[i], acc[i][j], 0, 0, 0);
            if (sync_mid && t == 7) {
                unsigned long long t0 = 0, t1 = 0;
                if (MODE & 64) t0 = __builtin_amdgcn_s_memtime();
                if (MODE & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                if (MODE & 64) t1 = __builtin_amdgcn_s_memtime();
                __builtin_amdgcn_s_barrier();
                if (MODE & 64) {
                    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
                    st_wait += t1 - t0;
                    st_bar += t2 - t1;
                }
            }
            if (MODE & 128) {
                // spread form: reads in the even slots (or 8-15 after the barrier), DMA pieces in odd slots -- never two DMA
                // instructions back to back, so a backed-up request queue delays one MFMA, not a run of them
                if (rslot0 == 0) {
                    if ((t & 1) == 0) one_read(nxt, rsa, rsb, rs, t >> 1);
                } else if (t >= 8) {
                    one_read(nxt, rsa, rsb, rs, t - 8);
                }
                if (dma_op == 1 && !(MODE & 1) && (t & 1)) stage_piece(dstage, dkt, t >> 1, 1);            // 8 B pieces: k-step 0
                if (dma_op == 0 && !(MODE & 1) && (t & 3) == 1) stage_piece(dstage, dkt, dpiece0 + (t >> 2), 0);  // 4 A pieces per k-step
            } else {
                if (t >= rslot0 && t < rslot0 + 8) one_read(nxt, rsa, rsb, rs, t - rslot0);
                if (dma_op >= 0 && !(MODE & 1) && t >= 8) stage_piece(dstage, dkt, t - 8, dma_op);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    const unsigned long long t_begin = (MODE & 64) ? __builtin_amdgcn_s_memtime() : 0;
    // ---- prologue: A(0), B(0), then A(1); wait for the first two ----
#pragma unroll
    for (int q = 0; q < 8; ++q) stage_piece(0, 0, q, 0);
#pragma unroll
    for (int q = 0; q < 8; ++q) stage_piece(0, 0, q, 1);
    if (!(MODE & 1)) {
#pragma unroll
        for (int q = 0; q < 8; ++q) stage_piece(1, nk > 1 ? 1 : 0, q, 0);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    const unsigned long long t_pro = (MODE & 64) ? __builtin_amdgcn_s_memtime() : 0;
    Frag f0, f1;
    if (MODE & 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                f0.a[i][e] = f1.a[i][e] = (bf16)(float)(lane + e);
                f0.b[i][e] = f1.b[i][e] = (bf16)(float)(lane - e);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) one_read(f0, 0, 0, 0, t);

    int sa = 0;  // A stage of K tile kt (kt % 3 without the division)
    for (int kt = 0; kt < nk; ++kt) {
        const int sb = kt & 1;
        const int sa1 = sa == NA - 1 ? 0 : sa + 1, sa2 = sa1 == NA - 1 ? 0 : sa1 + 1;
        const int kt1 = kt + 1 < nk ? kt + 1 : 0, kt2 = kt + 2 < nk ? kt + 2 : 0;  // branch-free: past the end re-stage tile 0
        kstep(f0, f1, sa, sb, 1, 0, 1, sb ^ 1, kt1, false);   // B pieces of K tile kt+1 (B stage last read in K tile kt-1)
        kstep(f1, f0, sa, sb, 2, 0, 0, sa2, kt2, false, 0);   // A pieces of K tile kt+2 (A stage last read in K tile kt-1)
        kstep(f0, f1, sa, sb, 3, 0, (MODE & 128) ? 0 : -1, sa2, kt2, false, 4);  // spread form: its second half
        kstep(f1, f0, sa1, sb ^ 1, 0, 8, -1, 0, 0, true);     // counted wait + barrier after MFMA 7, then the next tile's first reads
        sa = sa1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the dummy re-stages past the end must not outlive the workgroup
    const unsigned long long t_loop = (MODE & 64) ? __builtin_amdgcn_s_memtime() : 0;

    // ---- epilogue: D = B.A^T puts 4 consecutive output columns in 4 consecutive registers: 8-byte stores ----
    if (MODE & 4) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][9];
        if (s == 12345.678f) p.c[tid] = (bf16)s;
        return;
    }
    bf16* cbase = p.c + (int64_t)(tm * TM + wm * 128) * p.N + tn * TN + wn * 128;
    if (MODE & 32) {
        // through LDS: the wave's 128 x 128 bf16 tile (32 KB, rows of 256 B, 16-byte chunk c of row R kept at c ^ (R & 15)) is
        // written from the accumulator layout (8 bytes per lane) and leaves as whole rows: 16 bytes per lane, 4 rows per store
        __builtin_amdgcn_s_barrier();  // every wave is done with the operand stages (all DMA was drained above)
        char* scr = lds + w * 32768;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16 o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (bf16)acc[i][j][4 * g + e];
                    const int row = i * 32 + r, cb = (j * 32 + 8 * g + 4 * kh) * 2;
                    *reinterpret_cast<u32x2*>(scr + row * 256 + (((cb >> 4) ^ (row & 15)) << 4) + (cb & 15)) = *reinterpret_cast<const u32x2*>(o);
                }
            }
        }
        // wave-private region: no barrier needed, only this wave's own LDS writes must have landed (lgkmcnt, compiler-tracked)
        typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
#pragma unroll 8
        for (int it = 0; it < 32; ++it) {
            const int row = it * 4 + (lane >> 4), c = lane & 15;
            const u32x4 v = *reinterpret_cast<const u32x4*>(scr + row * 256 + ((c ^ (row & 15)) << 4));
            *reinterpret_cast<u32x4*>(cbase + (int64_t)row * p.N + c * 8) = v;
        }
        if ((MODE & 64) && lane == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            unsigned long long* o = p.stall + ((int64_t)blockIdx.x * 4 + w) * 5;
            o[0] = t_pro - t_begin; o[1] = st_wait; o[2] = st_bar; o[3] = t_loop - t_pro; o[4] = __builtin_amdgcn_s_memtime() - t_loop;
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16 o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16)acc[i][j][4 * g + e];
                *reinterpret_cast<u32x2*>(cbase + (int64_t)(i * 32 + r) * p.N + j * 32 + 8 * g + 4 * kh) = *reinterpret_cast<const u32x2*>(o);
            }
        }
    }
}

static float bf2f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

int main() {
    struct Case {
        int M, N, K;
        const char* what;
    } cases[] = {{192000, 512, 1152, "block-4 forward"}, {384000, 256, 1152, "block-3 dgrad"}, {768000, 256, 384, "block-2 forward"}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (const Case& cs : cases) {
        const int M = cs.M, N = cs.N, K = cs.K;
        std::vector<uint16_t> ha((size_t)M * K), hb((size_t)N * K);
        uint32_t s = 12345;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (int)((s >> 16) % 7) - 3; };  // integers in [-3, 3]: exact sums
        for (auto& v : ha) v = f2bf((float)rnd());
        for (auto& v : hb) v = f2bf((float)rnd());
        bf16 *da, *db, *dc;
        hipMalloc(&da, ha.size() * 2);
        hipMalloc(&db, hb.size() * 2);
        hipMalloc(&dc, (size_t)M * N * 2);
        hipMemcpy(da, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(db, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
        hipMemset(dc, 0xff, (size_t)M * N * 2);
        unsigned long long* dst;
        hipMalloc(&dst, (size_t)((M / TM + 8) * (N / TN)) * 4 * 5 * 8);
        Args a{da, db, dc, M, N, K, N / TN, dst};
        const int grid = (M / TM) * (N / TN);
        const int grid16 = ((M / TM + 7) / 8) * 8 * (N / TN);
        hipLaunchKernelGGL(gemm_w4<160>, dim3(grid), dim3(256), 0, 0, a);
        if (hipDeviceSynchronize() != hipSuccess) {
            printf("launch failed: %s\n", hipGetErrorString(hipGetLastError()));
            return 1;
        }
        // check 96 sampled rows (first, last, and scattered) against the host
        std::vector<uint16_t> hc((size_t)M * N);
        hipMemcpy(hc.data(), dc, hc.size() * 2, hipMemcpyDeviceToHost);
        long bad = 0, checked = 0;
        for (int t = 0; t < 96; ++t) {
            const int m = t < 32 ? t * 9 : (t < 64 ? M - 1 - (t - 32) * 7 : (int)(((long)t * 2654435761u) % M));
            for (int n = 0; n < N; ++n) {
                float ref = 0.f;
                for (int k = 0; k < K; ++k) ref += bf2f(ha[(size_t)m * K + k]) * bf2f(hb[(size_t)n * K + k]);
                const float got = bf2f(hc[(size_t)m * N + n]);
                ++checked;
                if (got != bf2f(f2bf(ref))) {
                    if (bad < 5) printf("  mismatch m=%d n=%d got %g want %g\n", m, n, got, ref);
                    ++bad;
                }
            }
        }
        printf("%-16s M=%d N=%d K=%d  check: %ld / %ld wrong\n", cs.what, M, N, K, bad, checked);
        auto launch = [&](int mode) {
            switch (mode) {
                case 0: hipLaunchKernelGGL(gemm_w4<0>, dim3(grid), dim3(256), 0, 0, a); break;
                case 1: hipLaunchKernelGGL(gemm_w4<1>, dim3(grid), dim3(256), 0, 0, a); break;
                case 2: hipLaunchKernelGGL(gemm_w4<2>, dim3(grid), dim3(256), 0, 0, a); break;
                case 3: hipLaunchKernelGGL(gemm_w4<3>, dim3(grid), dim3(256), 0, 0, a); break;
                case 5: hipLaunchKernelGGL(gemm_w4<5>, dim3(grid), dim3(256), 0, 0, a); break;
                case 4: hipLaunchKernelGGL(gemm_w4<4>, dim3(grid), dim3(256), 0, 0, a); break;
                case 160: hipLaunchKernelGGL(gemm_w4<160>, dim3(grid), dim3(256), 0, 0, a); break;
                case 132: hipLaunchKernelGGL(gemm_w4<132>, dim3(grid), dim3(256), 0, 0, a); break;
                case 224: hipLaunchKernelGGL(gemm_w4<224>, dim3(grid), dim3(256), 0, 0, a); break;
                case 96: hipLaunchKernelGGL(gemm_w4<96>, dim3(grid), dim3(256), 0, 0, a); break;
                case 32: hipLaunchKernelGGL(gemm_w4<32>, dim3(grid), dim3(256), 0, 0, a); break;
                case 48: hipLaunchKernelGGL(gemm_w4<48>, dim3(grid16), dim3(256), 0, 0, a); break;
                case 16: hipLaunchKernelGGL(gemm_w4<16>, dim3(grid16), dim3(256), 0, 0, a); break;
                case 20: hipLaunchKernelGGL(gemm_w4<20>, dim3(grid16), dim3(256), 0, 0, a); break;
                case 8: hipLaunchKernelGGL(gemm_w4<8>, dim3(grid), dim3(256), 0, 0, a); break;
                case 12: hipLaunchKernelGGL(gemm_w4<12>, dim3(grid), dim3(256), 0, 0, a); break;
                default: hipLaunchKernelGGL(gemm_w4<7>, dim3(grid), dim3(256), 0, 0, a); break;
            }
        };
        {
            hipLaunchKernelGGL(gemm_w4<224>, dim3(grid), dim3(256), 0, 0, a);
            hipDeviceSynchronize();
            std::vector<unsigned long long> hs((size_t)grid * 20);
            hipMemcpy(hs.data(), dst, hs.size() * 8, hipMemcpyDeviceToHost);
            double s5[5] = {0, 0, 0, 0, 0};
            for (int b = 0; b < grid; ++b) for (int k = 0; k < 5; ++k) s5[k] += (double)hs[(size_t)b * 20 + k];  // wave 0 of every workgroup
            printf("   cycles per tile (wave 0, mean over %d tiles): prologue wait %.0f | K loop %.0f of which DMA waits %.0f, barriers %.0f | epilogue %.0f\n", grid, s5[0] / grid, s5[3] / grid, s5[1] / grid, s5[2] / grid, s5[4] / grid);
        }
        for (int mode : {32, 160, 4, 132, 32, 160}) {
            for (int i = 0; i < 3; ++i) launch(mode);
            hipEventRecord(e0);
            const int reps = 10;
            for (int i = 0; i < reps; ++i) launch(mode);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1e3 / reps, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
            printf("   mode %3d (%s%s%s%s%s%s%s)  %8.1f us  %7.1f TFLOP/s\n", mode, mode & 1 ? "no-DMA " : "", mode & 2 ? "no-frag-reads " : "",
                   mode & 4 ? "no-stores " : "", mode & 8 ? "A-from-tile-0 " : "", mode & 16 ? "xcd-order " : "", mode & 32 ? "lds-epilogue " : "", mode & 128 ? "spread-dma" : (mode ? "" : "full"), us, tf);
        }
        hipFree(da);
        hipFree(db);
        hipFree(dc);
    }
    return 0;
}
