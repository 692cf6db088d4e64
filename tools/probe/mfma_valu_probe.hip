// How do VALU instructions and MFMAs share a SIMD on gfx950?  Register-only loops, cycles read in-kernel with s_memtime
// (clock-independent), wall time from HIP events beside it (shows the DVFS state).  One workgroup per CU.
//   hipcc --offload-arch=gfx950 -O3 mfma_valu_probe.hip -o mfma_valu_probe && ./mfma_valu_probe
// Per loop iteration a "M" wave issues 4 v_mfma_f32_32x32x16_bf16 (independent accumulators unless DEP) with NV plain
// v_fma_f32 (or NP v_pk_fma_f32) placed after each MFMA; a "V" wave issues only VALU.  Reported: cycles per iteration of the
// slowest wave, i.e. per 4 MFMAs (an MFMA occupies the matrix pipe for 32 cycles: 128 is the floor with one M wave per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

struct Cfg {
    int waves;      // waves per workgroup (4: one per SIMD, 8: two per SIMD)
    int split;      // 1: waves >= 4 are V waves (VALU only, 4*(NV+NP) per iteration), waves < 4 M waves without VALU
    int mfma;       // M waves issue MFMAs (0: VALU only everywhere)
};

template <int NV, int NP, bool DEP>
__global__ __launch_bounds__(512) void probe(Cfg c, int iters, unsigned long long* cyc, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bf16x8 fa, fb;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        fa[e] = (__bf16)(0.37f * ((lane * 7 + e * 3) % 11) - 1.5f);
        fb[e] = (__bf16)(0.21f * ((lane * 5 + e) % 13) - 1.1f);
    }
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float v[8];
    f32x2 pv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        v[i] = lane * 0.01f + i;
        pv[i] = f32x2{v[i], v[i] + 1.f};
    }
    const float a = 0.999f + 1e-6f * lane, b = 1e-3f;
    const f32x2 a2 = {a, a}, b2 = {b, b};
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const bool vwave = c.split && wave_u >= 4;
    const bool do_mfma = c.mfma && !vwave;
    const bool do_valu = !c.split || vwave;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    // the role is wave-uniform and fixed: one branch-free loop body per role
#define MFMA_I(i) acc[DEP ? 0 : i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[DEP ? 0 : i], 0, 0, 0)
#define VALU_GROUP()                                                                                                   \
    _Pragma("unroll") for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(a), "v"(b)); \
    _Pragma("unroll") for (int j = 0; j < NP; ++j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pv[j & 7]) : "v"(a2), "v"(b2))
    if (do_mfma && do_valu) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                MFMA_I(i);
                VALU_GROUP();
            }
        }
    } else if (do_mfma) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) MFMA_I(i);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                VALU_GROUP();
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i] + pv[i][0] + pv[i][1];
    asm volatile("" : "+v"(s));  // results are complete before the second timestamp
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    if (s == 123.456f) out[threadIdx.x] = s;
}

static unsigned long long* d_cyc;
static float* d_out;
static hipEvent_t e0, e1;

template <int NV, int NP, bool DEP>
static void run(const char* name, Cfg c) {
    const int iters = 40000, blocks = 256;
    hipMemset(d_cyc, 0, blocks * 8 * 8);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<NV, NP, DEP>), dim3(blocks), dim3(c.waves * 64), 0, 0, c, iters, d_cyc, d_out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long h[256 * 8];
    hipMemcpy(h, d_cyc, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long mx_m = 0, mx_v = 0;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < c.waves; ++w) {
            unsigned long long x = h[b * 8 + w];
            if (c.split && w >= 4) mx_v = x > mx_v ? x : mx_v; else mx_m = x > mx_m ? x : mx_m;
        }
    const double cm = (double)mx_m / iters, cv = (double)mx_v / iters;
    const double tot = cm > cv ? cm : cv;
    printf("%-58s %7.1f cyc/iter (M %6.1f V %6.1f)  %7.3f ms  -> %.2f GHz\n", name, tot, cm, cv, ms, tot * iters / (ms * 1e6));
}

int main() {
    hipMalloc(&d_cyc, 256 * 8 * 8);
    hipMalloc(&d_out, 4096);
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const Cfg w4 = {4, 0, 1}, w8 = {8, 0, 1}, sp = {8, 1, 1}, v4 = {4, 0, 0}, v8 = {8, 0, 0};
    for (int warm = 0; warm < 6; ++warm) run<0, 0, false>("warm-up", w8);  // let the clocks settle
    printf("--- one wave per SIMD, MFMA + n plain VALU after each MFMA (same wave)\n");
    run<0, 0, false>("4 MFMA", w4);
    run<2, 0, false>("4 MFMA + 4x2 v_fma", w4);
    run<4, 0, false>("4 MFMA + 4x4 v_fma", w4);
    run<5, 0, false>("4 MFMA + 4x5 v_fma", w4);
    run<6, 0, false>("4 MFMA + 4x6 v_fma", w4);
    run<8, 0, false>("4 MFMA + 4x8 v_fma", w4);
    run<12, 0, false>("4 MFMA + 4x12 v_fma", w4);
    run<16, 0, false>("4 MFMA + 4x16 v_fma", w4);
    run<0, 4, false>("4 MFMA + 4x4 v_pk_fma", w4);
    run<0, 8, false>("4 MFMA + 4x8 v_pk_fma", w4);
    run<0, 0, true>("4 dependent MFMA", w4);
    run<4, 0, true>("4 dependent MFMA + 4x4 v_fma", w4);
    run<8, 0, true>("4 dependent MFMA + 4x8 v_fma", w4);
    printf("--- VALU only\n");
    run<8, 0, false>("4x8 v_fma, one wave per SIMD", v4);
    run<8, 0, false>("4x8 v_fma, two waves per SIMD", v8);
    run<0, 8, false>("4x8 v_pk_fma, one wave per SIMD", v4);
    run<0, 8, false>("4x8 v_pk_fma, two waves per SIMD", v8);
    printf("--- two waves per SIMD\n");
    run<0, 0, false>("both waves: 4 MFMA", w8);
    run<4, 0, false>("both waves: 4 MFMA + 4x4 v_fma", w8);
    run<8, 0, false>("both waves: 4 MFMA + 4x8 v_fma", w8);
    run<4, 0, false>("split: M wave 4 MFMA | V wave 4x4 v_fma", sp);
    run<8, 0, false>("split: M wave 4 MFMA | V wave 4x8 v_fma", sp);
    run<16, 0, false>("split: M wave 4 MFMA | V wave 4x16 v_fma", sp);
    run<0, 8, false>("split: M wave 4 MFMA | V wave 4x8 v_pk_fma", sp);
    run<8, 0, true>("split: M wave 4 dependent MFMA | V wave 4x8 v_fma", sp);
    return 0;
}
