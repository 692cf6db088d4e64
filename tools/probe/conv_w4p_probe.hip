// PERSISTENT form of tools/probe/conv_w4_probe.hip (k=3 forward convolution, one wave per SIMD, input-resident A): the K-tile
// stream of a workgroup runs ACROSS its output tiles, so a tile has no cold start -- while the last K tiles of tile i run, the
// first A block and B stage of tile i+1 are already being staged -- and the epilogue no longer needs the operand memory: it
// transposes the wave's 128 x 128 tile through a private 8 KB region in four 32-row passes (LDS: 128 KB operands + 32 KB).
// conv_w4_probe's in-kernel accounting put the cold start at 6 500 of a tile's 69 000 cycles.
//
// Differences from conv_w4_probe.hip, everything else (layouts, swizzles, per-K-tile DMA plan, waits) is the same:
//   * grid = min(tiles, 256 * WG_PER_CU=1) workgroups; tile index it = blockIdx.x + k * gridDim.x, decoded (tn fastest);
//   * "next K tile" / "next chunk" of the DMA plan roll over into the next tile of this workgroup (its own window, position
//     tile and channel tile: a second set of source offsets, recomputed when a tile's last chunk starts);
//   * the B-stage and A-block parities follow stream counters instead of the tile-local g and chunk;
//   * past the workgroup's last tile the DMAs re-stage its first tile's first operands (never read).
// NOT YET RUN ON HARDWARE (round 1 ended without GPU budget): compiles with 512 registers.  First action:
//   hipcc --offload-arch=gfx950 -O3 conv_w4p_probe.hip -o conv_w4p_probe && ./conv_w4p_probe
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
constexpr int SCR0 = 4 * OPB;                 // per-wave epilogue scratch (4 x 8 KB) starts here
constexpr int LDS_BYTES = 5 * OPB;            // 160 KB

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
    int total_tiles;
};

__global__ __attribute__((amdgpu_flat_work_group_size(256, 256), amdgpu_waves_per_eu(1, 1))) void conv_w4p_fwd(Args p) {
    __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int chunks = p.Cin / 64, nk = chunks * 3;
    const int a_pitch = p.Cin * 2, b_pitch = 3 * p.Cin * 2;
    const int my_tiles = (p.total_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1;

    // ---- tile state: current (c) and next (n) tile of this workgroup; all of it wave-uniform (SGPRs) ----
    struct Tile {
        int n, tl, tn, t0;
    };
    const int prow = lane >> 3;
    unsigned b_off[8], swz[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        swz[q] = (unsigned)(((lane & 7) ^ ((4 * q + (prow >> 1)) & 7)) * 16);
        b_off[q] = (unsigned)((q * 8 + prow) * b_pitch) + swz[q];
    }
    auto decode = [&](int it, Tile& T) {
        const int v = (int)blockIdx.x + (it < my_tiles ? it : 0) * (int)gridDim.x;  // past the end: the first tile again (dummy DMA source)
        T.tn = __builtin_amdgcn_readfirstlane(v % p.tilesN);
        const int grp = v / p.tilesN;
        T.tl = __builtin_amdgcn_readfirstlane(grp % p.tilesL);
        T.n = __builtin_amdgcn_readfirstlane(grp / p.tilesL);
        T.t0 = T.tl * TROWS;
    };
    // A piece q of `chunk` of tile (n, t0): rows t0 + 64w + 8q + prow of the window, clamped to its L + 2 padded rows (offsets
    // are recomputed per piece -- four VALU instructions in an MFMA shadow -- instead of holding two tiles' worth of them)
    auto stage_a = [&](int tn_, int t0_, int blk, int chunk, int q) {
        int pr = t0_ + w * 64 + q * 8 + prow;
        pr = pr < p.L + 2 ? pr : p.L + 1;
        const char* src = reinterpret_cast<const char*>(p.act) + (int64_t)tn_ * (p.L + 2) * a_pitch + (int64_t)chunk * ROWB;
        glds16(src + ((unsigned)(pr * a_pitch) + swz[q]), lds + blk * OPB + (w * 64 + q * 8) * ROWB);
    };
    auto stage_b = [&](int tn_, int stg, int g, int q) {
        const int chunk = g / 3, tap = g - 3 * chunk;
        const char* src = reinterpret_cast<const char*>(p.wf) + (int64_t)(tn_ * 256 + w * 64) * b_pitch + (int64_t)(tap * p.Cin + chunk * 64) * 2;
        glds16(src + b_off[q], lds + B0 + stg * OPB + (w * 64 + q * 8) * ROWB);
    };

    // ---- fragment geometry ----
    const int r = lane & 31, kh = lane >> 5;
    int foff_b[4], foff_a[3][4];
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
    auto one_read = [&](Frag& f, int blk, int stg, int tap, int s, int t) {
        if (t < 4) {
            f.a[t] = *reinterpret_cast<const bf16x8*>(lds + blk * OPB + a_rows + t * 32 * ROWB + foff_a[tap][s]);
        } else {
            f.b[t - 4] = *reinterpret_cast<const bf16x8*>(lds + B0 + stg * OPB + b_rows + (t - 4) * 32 * ROWB + foff_b[s]);
        }
    };

    f32x16 acc[4][4];
    auto zero = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    };
    zero();

    // do_b: stage the 8 B pieces of K tile bg of channel tile b_tn; do_a: stage 4 A pieces of `achunk` of window a_n at a_t0 (the flags
    // are literals at the call sites: the loop body stays branch-free, which is what keeps the compiler's wait counts exact)
    auto kstep = [&](const Frag& cur, Frag& nxt, int rblk, int rstg, int rtap, int rs, int rslot0, bool do_b, int b_tn, int bstg, int bg, bool do_a, int a_n,
                     int a_t0, int ablk, int achunk, int aq0, int wait_n) {
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
            if (do_b && (t & 1)) stage_b(b_tn, bstg, bg, t >> 1);
            if (do_a && (t & 3) == 1) stage_a(a_n, a_t0, ablk, achunk, aq0 + (t >> 2));
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    Tile Tc, Tn;
    decode(0, Tc);
    decode(1, Tn);
    // ---- prologue of the stream: A(0), B(0) of the first tile ----
#pragma unroll
    for (int q = 0; q < 8; ++q) stage_a(Tc.n, Tc.t0, 0, 0, q);
#pragma unroll
    for (int q = 0; q < 8; ++q) stage_b(Tc.tn, 0, 0, q);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    Frag f0, f1;
#pragma unroll
    for (int t = 0; t < 8; ++t) one_read(f0, 0, 0, 0, 0, t);

    int sb = 0, ablk = 0;  // stream parities: B stage of the current K tile, A block of the current chunk
    char* scr = lds + SCR0 + w * 8192;
    const int c16 = lane & 15;
    for (int it = 0; it < my_tiles; ++it) {
        for (int c = 0; c < chunks; ++c) {
            const bool last_c = c + 1 == chunks;  // the chunk after the tile's last one is chunk 0 of the next tile
            const int a_n = last_c ? Tn.n : Tc.n, a_t0 = last_c ? Tn.t0 : Tc.t0;
            const int cn = last_c ? 0 : c + 1;
#pragma unroll
            for (int tap = 0; tap < 3; ++tap) {
                const int g = 3 * c + tap;
                const bool last_g = g + 1 == nk;
                const int b_tn = last_g ? Tn.tn : Tc.tn;
                const int gn = last_g ? 0 : g + 1;
                const int ntap = tap == 2 ? 0 : tap + 1, nblk = tap == 2 ? (ablk ^ 1) : ablk;
                kstep(f0, f1, ablk, sb, tap, 1, 0, true, b_tn, sb ^ 1, gn, false, 0, 0, 0, 0, 0, -1);
                kstep(f1, f0, ablk, sb, tap, 2, 0, false, 0, 0, 0, tap != 2, a_n, a_t0, ablk ^ 1, cn, tap == 0 ? 0 : 4, -1);
                kstep(f0, f1, ablk, sb, tap, 3, 0, false, 0, 0, 0, false, 0, 0, 0, 0, 0, -1);
                kstep(f1, f0, nblk, sb ^ 1, ntap, 0, 8, false, 0, 0, 0, false, 0, 0, 0, 0, 0, tap == 2 ? 0 : 4);
                sb ^= 1;
            }
            ablk ^= 1;
        }
        // ---- epilogue of tile `it` (f0 already holds the first fragments of the next tile; its operands are in LDS, nothing is in
        // flight: the last wait of a chunk is vmcnt(0)); four 32-row passes through the wave's private 8 KB ----
        const int col0 = Tc.tn * 256 + wn * 128;
        bf16* zbase = p.z + ((int64_t)Tc.n * p.L + Tc.t0 + wm * 128) * p.Cout + col0;
        float s8[8], q8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s8[e] = q8[e] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int cc = j * 32 + 8 * g4 + 4 * kh;
                    const float4 bv = *reinterpret_cast<const float4*>(p.bias + col0 + cc);
                    bf16 o[4];
                    o[0] = (bf16)fmaxf(acc[i][j][4 * g4 + 0] + bv.x, 0.f);
                    o[1] = (bf16)fmaxf(acc[i][j][4 * g4 + 1] + bv.y, 0.f);
                    o[2] = (bf16)fmaxf(acc[i][j][4 * g4 + 2] + bv.z, 0.f);
                    o[3] = (bf16)fmaxf(acc[i][j][4 * g4 + 3] + bv.w, 0.f);
                    const int cb = cc * 2;  // row r of the 32-row pass; 16-byte chunk c of row R kept at c ^ (R & 15)
                    *reinterpret_cast<u32x2*>(scr + r * 256 + (((cb >> 4) ^ (r & 15)) << 4) + (cb & 15)) = *reinterpret_cast<const u32x2*>(o);
                }
            }
#pragma unroll
            for (int it8 = 0; it8 < 8; ++it8) {
                const int row = it8 * 4 + (lane >> 4);       // row inside the pass
                const int trow = wm * 128 + i * 32 + row;    // row inside the 256-row MFMA tile
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(scr + row * 256 + ((c16 ^ (row & 15)) << 4));
                if (trow < TROWS && Tc.t0 + trow < p.L) {
                    *reinterpret_cast<bf16x8*>(zbase + (int64_t)(i * 32 + row) * p.Cout + c16 * 8) = v;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float x = (float)v[e];
                        s8[e] += x;
                        q8[e] = fmaf(x, x, q8[e]);
                    }
                }
            }
            // the next pass overwrites the scratch: this wave's reads above are complete when their values have been used
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s8[e] += __shfl_xor(s8[e], 16, 64);
            s8[e] += __shfl_xor(s8[e], 32, 64);
            q8[e] += __shfl_xor(q8[e], 16, 64);
            q8[e] += __shfl_xor(q8[e], 32, 64);
        }
        if (lane < 16) {
            const int64_t srow = ((int64_t)Tc.n * p.tilesL + Tc.tl) * 2 + wm;
            float* ps = p.stat_sum + srow * p.Cout + col0 + c16 * 8;
            float* pq = p.stat_sq + srow * p.Cout + col0 + c16 * 8;
            *reinterpret_cast<float4*>(ps) = float4{s8[0], s8[1], s8[2], s8[3]};
            *reinterpret_cast<float4*>(ps + 4) = float4{s8[4], s8[5], s8[6], s8[7]};
            *reinterpret_cast<float4*>(pq) = float4{q8[0], q8[1], q8[2], q8[3]};
            *reinterpret_cast<float4*>(pq + 4) = float4{q8[4], q8[5], q8[6], q8[7]};
        }
        zero();
        Tc = Tn;
        decode(it + 2, Tn);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing may still be landing in LDS when the workgroup ends
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
        const int total = n * tilesL * tilesN;
        Args a{da, dw, dbias, dz, dss, dsq, n, L, Cin, Cout, tilesL, tilesN, total};
        const int grid = total < 256 ? total : 256;
        hipLaunchKernelGGL(conv_w4p_fwd, dim3(grid), dim3(256), 0, 0, a);
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
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(conv_w4p_fwd, dim3(grid), dim3(256), 0, 0, a);
            hipEventRecord(e0);
            const int reps = 10;
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(conv_w4p_fwd, dim3(grid), dim3(256), 0, 0, a);
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
