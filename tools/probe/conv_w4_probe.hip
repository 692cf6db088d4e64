// Round-2 candidate for the k=3 forward convolution: tools/probe/gemm_w4_probe.hip turned into the
// conv -- ONE wave per SIMD (128 x 128 per wave, 256 accumulator registers), INPUT-RESIDENT A (the 256-row block of a
// 64-channel chunk is staged once and read at row offsets 0 / 1 / 2 for the three taps: a third of the A pieces to issue),
// fused bias + ReLU + bf16 + BatchNorm partial statistics in an LDS-transposed epilogue with 16-byte whole-row stores.
//
//   act  : (n, L + 2, Cin) bf16, halo rows zero (the engine's padded activation layout: the im2col row of output position t is
//          the 3*Cin contiguous elements starting at padded row t)
//   wf   : (Cout, 3*Cin) bf16, K index = tap*Cin + ci          bias : (Cout) fp32
//   z    : (n, L, Cout) bf16 = relu(conv + bias)                stat_sum / stat_sq : (n * tilesL * 2, Cout) fp32 partials of z, z^2
//
// Tiles: 254 output positions x 256 channels per workgroup (the block of 256 padded rows serves positions t0 .. t0+253 for all
// three taps; the two last rows of the 256 x 256 MFMA tile are computed on whatever follows and never stored or counted).
// LDS: 2 A blocks (32 KB each, one per channel chunk, double-buffered) + 2 B stages (32 KB each, one per K tile = (chunk, tap))
//      = 128 KB; the epilogue reuses it (one 32 KB region per wave).
// K-tile g = 3*chunk + tap; per K tile one counted DMA wait + one s_barrier in the middle of its last k-step:
//   K tile 3c   : k-step 0 issues B(3c+1) [8 pieces], k-step 1 A(c+1) first half [4] -> at its end wait vmcnt(4): B(3c+1) landed
//   K tile 3c+1 : k-step 0 issues B(3c+2) [8],        k-step 1 A(c+1) second half [4] -> vmcnt(4): B(3c+2), older A half landed
//   K tile 3c+2 : k-step 0 issues B(3c+3) [8]                                          -> vmcnt(0): B(3c+3), all of A(c+1) landed
// (loads complete in order, so "all but the newest 4" covers every B piece: inside a K tile all B pieces precede the A pieces)
// WAR: a B stage was last read in K tile g-1, whose reads completed before the barrier inside K tile g-1; the A block of chunk
//      c+1 overwrites the block of chunk c-1, last read in K tile 3c-1, likewise.
// Run once at the very end of round 1 (profiles/r01_conv_w4_probe.txt): outputs and statistics exact on both check cases (the
// ragged one included); block-4 forward (256 windows) 226.5 us against 265 us for conv_nt8_kernel in the library; block-2
// forward (K = 384) 308.7 us against 277 us for the 128^2 kernel -- short K tiles need the persistent form (no cold start per
// tile, epilogue under the next tile's MFMAs) before this structure pays there.
//   hipcc --offload-arch=gfx950 -O3 conv_w4_probe.hip -o conv_w4_probe && ./conv_w4_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

constexpr int TROWS = 254;                    // output positions per tile
constexpr int ROWB = 128;                     // bytes of K per LDS row (64 bf16)
constexpr int OPB = 256 * ROWB;               // one A block / one B stage: 32 KB
constexpr int B0 = 2 * OPB;                   // B stages start here
constexpr int LDS_BYTES = 4 * OPB;            // 128 KB

__device__ inline void glds16(const char* gsrc, char* lds_base) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds(gsrc, (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
#endif
}

struct Args {
    const bf16* act;
    const bf16* wf;
    const float* bias;
    bf16* z;
    float* stat_sum;
    float* stat_sq;
    int n, L, Cin, Cout;
    int tilesL, tilesN;
};

__global__ __attribute__((amdgpu_flat_work_group_size(256, 256), amdgpu_waves_per_eu(1, 1))) void conv_w4_fwd(Args p) {
    __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    // workgroup -> (window, position tile, channel tile); channel tiles of one position tile are neighbours
    const int tn = __builtin_amdgcn_readfirstlane((int)blockIdx.x % p.tilesN);
    const int grp = (int)blockIdx.x / p.tilesN;
    const int tl = __builtin_amdgcn_readfirstlane(grp % p.tilesL), n = __builtin_amdgcn_readfirstlane(grp / p.tilesL);
    const int t0 = tl * TROWS;
    const int chunks = p.Cin / 64, nk = chunks * 3;
    const int a_pitch = p.Cin * 2, b_pitch = 3 * p.Cin * 2;

    // ---- DMA geometry (as gemm_w4_probe): a piece = 8 rows x 128 B; wave w stages rows [64w, 64w+64) of a block ----
    const int prow = lane >> 3;
    // A block row R <-> padded row t0 + R of window n (clamped to the window's L + 2 padded rows: the tail tile reads its last row again)
    const char* a_win = reinterpret_cast<const char*>(p.act) + (int64_t)n * (p.L + 2) * a_pitch;
    const char* b_base = reinterpret_cast<const char*>(p.wf) + (int64_t)(tn * 256 + w * 64) * b_pitch;
    unsigned a_off[8], b_off[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int key = (4 * q + (prow >> 1)) & 7;  // ((row >> 1) & 7) of block row 64w + 8q + prow
        int pr = t0 + w * 64 + q * 8 + prow;
        pr = pr < p.L + 2 ? pr : p.L + 1;
        a_off[q] = (unsigned)(pr * a_pitch + (((lane & 7) ^ key) * 16));
        b_off[q] = (unsigned)((q * 8 + prow) * b_pitch + (((lane & 7) ^ key) * 16));
    }
    auto stage_a = [&](int blk, int chunk, int q) { glds16(a_win + (int64_t)chunk * ROWB + a_off[q], lds + blk * OPB + (w * 64 + q * 8) * ROWB); };
    auto stage_b = [&](int stg, int g, int q) {  // K tile g = 3*chunk + tap -> weight columns tap*Cin + 64*chunk
        const int chunk = g / 3, tap = g - 3 * chunk;
        glds16(b_base + (int64_t)(tap * p.Cin + chunk * 64) * 2 + b_off[q], lds + B0 + stg * OPB + (w * 64 + q * 8) * ROWB);
    };

    // ---- fragment geometry ----
    const int r = lane & 31, kh = lane >> 5;
    int foff_b[4], foff_a[3][4];  // B rows are channels (no shift); A rows are shifted by the tap
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        foff_b[s] = r * ROWB + (((2 * s + kh) ^ ((r >> 1) & 7)) * 16);
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) foff_a[tap][s] = (r + tap) * ROWB + (((2 * s + kh) ^ (((r + tap) >> 1) & 7)) * 16);
    }
    const int a_rows = wm * 128 * ROWB, b_rows = wn * 128 * ROWB;
    struct Frag {
        bf16x8 a[4], b[4];
    };
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

    // one k-step: 16 MFMAs, one filler after each (fragment reads of the next k-step in slots rslot0 .. rslot0+7; DMA pieces in
    // the odd slots of the first three k-steps; the counted wait + barrier after MFMA 7 of a K tile's last k-step)
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
            if (bg >= 0 && (t & 1)) stage_b(bstg, bg, t >> 1);                          // k-step 0: the 8 B pieces, odd slots
            if (achunk >= 0 && (t & 3) == 1) stage_a(ablk, achunk, aq0 + (t >> 2));    // k-step 1: 4 A pieces, slots 1, 5, 9, 13
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

    // K-tile stream (DMA plan in the header).  Past the end the DMAs re-stage K tile 0 / chunk 0 into memory nobody reads.
    for (int c = 0; c < chunks; ++c) {
        const int ablk = c & 1;
        const int cn = c + 1 < chunks ? c + 1 : 0;
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
            const int g = 3 * c + tap, sb = g & 1;
            const int gn = g + 1 < nk ? g + 1 : 0;  // past the end: re-stage K tile 0 into the stage nobody reads next
            const int ntap = tap == 2 ? 0 : tap + 1, nblk = tap == 2 ? (ablk ^ 1) : ablk;  // where the next K tile reads
            kstep(f0, f1, ablk, sb, tap, 1, 0, sb ^ 1, gn, 0, -1, 0, -1);                                  // B(g+1)
            kstep(f1, f0, ablk, sb, tap, 2, 0, 0, -1, ablk ^ 1, tap == 2 ? -1 : cn, tap == 0 ? 0 : 4, -1);  // half of A(c+1)
            kstep(f0, f1, ablk, sb, tap, 3, 0, 0, -1, 0, -1, 0, -1);
            kstep(f1, f0, nblk, sb ^ 1, ntap, 0, 8, 0, -1, 0, -1, 0, tap == 2 ? 0 : 4);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave is done with the operand memory: it becomes the epilogue scratch

    // ---- epilogue: bias + ReLU + bf16, statistics of the stored values, whole-row 16-byte stores ----
    // scratch: the wave's 128 x 128 bf16 tile, rows of 256 B, 16-byte chunk c of row R kept at c ^ (R & 15)
    char* scr = lds + w * 32768;
    const int col0 = tn * 256 + wn * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cc = j * 32 + 8 * g + 4 * kh;  // first of this lane's 4 consecutive channels in the wave tile
                const float4 bv = *reinterpret_cast<const float4*>(p.bias + col0 + cc);
                bf16 o[4];
                o[0] = (bf16)fmaxf(acc[i][j][4 * g + 0] + bv.x, 0.f);
                o[1] = (bf16)fmaxf(acc[i][j][4 * g + 1] + bv.y, 0.f);
                o[2] = (bf16)fmaxf(acc[i][j][4 * g + 2] + bv.z, 0.f);
                o[3] = (bf16)fmaxf(acc[i][j][4 * g + 3] + bv.w, 0.f);
                const int row = i * 32 + r, cb = cc * 2;
                *reinterpret_cast<u32x2*>(scr + row * 256 + (((cb >> 4) ^ (row & 15)) << 4) + (cb & 15)) = *reinterpret_cast<const u32x2*>(o);
            }
        }
    }
    // read back row-wise: lane -> chunk (lane & 15) = 8 channels, rows it*4 + (lane >> 4); the statistics are taken from the
    // rounded values that are stored (BatchNorm normalises what it will read)
    float s8[8], q8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s8[e] = q8[e] = 0.f;
    bf16* zbase = p.z + ((int64_t)n * p.L + t0 + wm * 128) * p.Cout + col0;
    const int c16 = lane & 15;
#pragma unroll 4
    for (int it = 0; it < 32; ++it) {
        const int row = it * 4 + (lane >> 4);
        const int trow = wm * 128 + row;  // row inside the 256-row MFMA tile
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(scr + row * 256 + ((c16 ^ (row & 15)) << 4));
        const bool ok = trow < TROWS && t0 + trow < p.L;
        if (ok) {
            *reinterpret_cast<bf16x8*>(zbase + (int64_t)row * p.Cout + c16 * 8) = v;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = (float)v[e];
                s8[e] += x;
                q8[e] = fmaf(x, x, q8[e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        s8[e] += __shfl_xor(s8[e], 16, 64);
        s8[e] += __shfl_xor(s8[e], 32, 64);
        q8[e] += __shfl_xor(q8[e], 16, 64);
        q8[e] += __shfl_xor(q8[e], 32, 64);
    }
    if (lane < 16) {
        const int64_t srow = ((int64_t)n * p.tilesL + tl) * 2 + wm;
        float* ps = p.stat_sum + srow * p.Cout + col0 + c16 * 8;
        float* pq = p.stat_sq + srow * p.Cout + col0 + c16 * 8;
        *reinterpret_cast<float4*>(ps) = float4{s8[0], s8[1], s8[2], s8[3]};
        *reinterpret_cast<float4*>(ps + 4) = float4{s8[4], s8[5], s8[6], s8[7]};
        *reinterpret_cast<float4*>(pq) = float4{q8[0], q8[1], q8[2], q8[3]};
        *reinterpret_cast<float4*>(pq + 4) = float4{q8[4], q8[5], q8[6], q8[7]};
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
        int n, L, Cin, Cout;
        const char* what;
    } cases[] = {{3, 750, 384, 512, "block-4 forward, 3 windows (check)"}, {5, 700, 128, 256, "ragged L, 5 windows (check)"},
                 {256, 750, 384, 512, "block-4 forward"}, {256, 3000, 128, 256, "block-2 forward"}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (const Case& cs : cases) {
        const int n = cs.n, L = cs.L, Cin = cs.Cin, Cout = cs.Cout;
        const int tilesL = (L + TROWS - 1) / TROWS, tilesN = Cout / 256;
        std::vector<uint16_t> ha((size_t)n * (L + 2) * Cin, 0), hw((size_t)Cout * 3 * Cin);
        std::vector<float> hb(Cout);
        uint32_t s = 777;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (int)((s >> 16) % 7) - 3; };
        for (int w = 0; w < n; ++w)
            for (int t = 1; t <= L; ++t)
                for (int c = 0; c < Cin; ++c) ha[((size_t)w * (L + 2) + t) * Cin + c] = f2bf((float)rnd());
        for (auto& v : hw) v = f2bf((float)rnd());
        for (auto& v : hb) v = (float)rnd() * 4.f;
        bf16 *da, *dw, *dz;
        float *dbias, *dss, *dsq;
        const size_t srows = (size_t)n * tilesL * 2;
        hipMalloc(&da, ha.size() * 2);
        hipMalloc(&dw, hw.size() * 2);
        hipMalloc(&dz, (size_t)n * L * Cout * 2);
        hipMalloc(&dbias, Cout * 4);
        hipMalloc(&dss, srows * Cout * 4);
        hipMalloc(&dsq, srows * Cout * 4);
        hipMemcpy(da, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dbias, hb.data(), Cout * 4, hipMemcpyHostToDevice);
        hipMemset(dz, 0xff, (size_t)n * L * Cout * 2);
        Args a{da, dw, dbias, dz, dss, dsq, n, L, Cin, Cout, tilesL, tilesN};
        const int grid = n * tilesL * tilesN;
        hipLaunchKernelGGL(conv_w4_fwd, dim3(grid), dim3(256), 0, 0, a);
        if (hipDeviceSynchronize() != hipSuccess) {
            printf("launch failed: %s\n", hipGetErrorString(hipGetLastError()));
            return 1;
        }
        if (n <= 16) {  // full check of z and of the summed statistics against the host (integer data: exact in fp32)
            std::vector<uint16_t> hz((size_t)n * L * Cout);
            std::vector<float> hss(srows * Cout), hsq(srows * Cout);
            hipMemcpy(hz.data(), dz, hz.size() * 2, hipMemcpyDeviceToHost);
            hipMemcpy(hss.data(), dss, hss.size() * 4, hipMemcpyDeviceToHost);
            hipMemcpy(hsq.data(), dsq, hsq.size() * 4, hipMemcpyDeviceToHost);
            long bad = 0, checked = 0;
            std::vector<double> rs((size_t)n * Cout, 0.0), rq((size_t)n * Cout, 0.0);
            for (int w = 0; w < n; ++w)
                for (int t = 0; t < L; ++t)
                    for (int co = 0; co < Cout; ++co) {
                        float ref = hb[co];
                        const uint16_t* arow = &ha[((size_t)w * (L + 2) + t) * Cin];
                        const uint16_t* wrow = &hw[(size_t)co * 3 * Cin];
                        for (int k = 0; k < 3 * Cin; ++k) ref += bf2f(arow[k]) * bf2f(wrow[k]);
                        ref = ref > 0.f ? ref : 0.f;
                        const float want = bf2f(f2bf(ref)), got = bf2f(hz[((size_t)w * L + t) * Cout + co]);
                        rs[(size_t)w * Cout + co] += want;
                        rq[(size_t)w * Cout + co] += (double)want * want;
                        ++checked;
                        if (got != want && bad++ < 5) printf("  mismatch n=%d t=%d co=%d got %g want %g\n", w, t, co, got, want);
                    }
            long sbad = 0;
            for (int w = 0; w < n; ++w)
                for (int co = 0; co < Cout; ++co) {
                    double gs = 0, gq = 0;
                    for (int k = 0; k < tilesL * 2; ++k) {
                        gs += hss[((size_t)w * tilesL * 2 + k) * Cout + co];
                        gq += hsq[((size_t)w * tilesL * 2 + k) * Cout + co];
                    }
                    if (fabs(gs - rs[(size_t)w * Cout + co]) > 1e-3 * (1 + fabs(gs)) || fabs(gq - rq[(size_t)w * Cout + co]) > 1e-3 * (1 + fabs(gq))) ++sbad;
                }
            printf("%-40s n=%d L=%d Cin=%d Cout=%d  z: %ld / %ld wrong   statistics: %ld / %d wrong\n", cs.what, n, L, Cin, Cout, bad, checked, sbad,
                   n * Cout);
        } else {
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(conv_w4_fwd, dim3(grid), dim3(256), 0, 0, a);
            hipEventRecord(e0);
            const int reps = 10;
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(conv_w4_fwd, dim3(grid), dim3(256), 0, 0, a);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1e3 / reps, tf = 2.0 * n * L * 3.0 * Cin * Cout / (us * 1e-6) / 1e12;
            printf("%-40s n=%d L=%d Cin=%d Cout=%d  %8.1f us  %7.1f TFLOP/s (algorithmic)\n", cs.what, n, L, Cin, Cout, us, tf);
        }
        hipFree(da);
        hipFree(dw);
        hipFree(dz);
        hipFree(dbias);
        hipFree(dss);
        hipFree(dsq);
    }
    return 0;
}
