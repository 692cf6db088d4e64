// How long does a wave spend ISSUING global_load_lds_dwordx4 (LDS-DMA) instructions?  conv_nt2r_kernel's waves need ~250 clocks per
// instruction in the tile prologue (profile build).  One instruction moves 1 KB: either 16 rows x 64 B (half cache lines: the
// forward/dgrad kernels' A and B pieces) or 8 rows x 128 B (whole lines: the wgrad kernel's pieces).  `wgs` workgroups of 4 waves per
// CU each issue `n` instructions back to back over a large tensor (row pitch 1024 B), then wait; clocks from s_memtime.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int ROWB>
__global__ __launch_bounds__(256) void k(const char* src, long rows, int n, int rounds, long long* out) {
    __shared__ __attribute__((aligned(1024))) char lds[64 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    constexpr int LPR = ROWB / 16;            // lanes per row
    const int lrow = lane / LPR, lch = lane % LPR;
    constexpr int RPI = 64 / LPR;             // rows per instruction
    long long t_issue = 0, t_total = 0;
    long base_row = ((long)blockIdx.x * 4 + w) * 977 % (rows - 256);
    for (int r = 0; r < rounds; ++r) {
        const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
        for (int i = 0; i < n; ++i) {
            const char* g = src + (base_row + (long)i * RPI + lrow) * 1024 + (r & 7) * ROWB + lch * 16;
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)(lds + __builtin_amdgcn_readfirstlane((w * 16 + (i & 15)) * 1024)), 16, 0, 0);
#endif
        }
        const long long t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long t2 = __builtin_amdgcn_s_memtime();
        t_issue += t1 - t0;
        t_total += t2 - t0;
        base_row = (base_row + 4099) % (rows - 256);
        __syncthreads();
    }
    if (lane == 0) {
        out[((long)blockIdx.x * 4 + w) * 2] = t_issue;
        out[((long)blockIdx.x * 4 + w) * 2 + 1] = t_total;
    }
    if (lds[tid] == 77 && src[0] == 99) out[0] = 0;
}

template <int ROWB>
static void run(const char* name, const char* buf, long rows, int wgs_per_cu, int n) {
    long long* out;
    const int grid = 256 * wgs_per_cu, rounds = 200;
    hipMallocManaged(&out, (size_t)grid * 4 * 2 * 8);
    hipLaunchKernelGGL(k<ROWB>, dim3(grid), dim3(256), 0, 0, buf, rows, n, rounds, out);
    hipDeviceSynchronize();
    double a = 0, b = 0;
    for (int i = 0; i < grid * 4; ++i) { a += out[2 * i]; b += out[2 * i + 1]; }
    a /= (double)grid * 4 * rounds; b /= (double)grid * 4 * rounds;
    printf("%-18s %d WG/CU, %2d instr per burst: issue %7.0f clk (%5.0f per instr), issue + wait %7.0f clk\n", name, wgs_per_cu, n, a, a / n, b);
    hipFree(out);
}

int main(int argc, char** argv) {
    // default: 2 GB source (every piece comes from HBM); "l2": 8 MB source (the pieces hit the 4 MB-per-XCD L2 / Infinity Cache)
    const bool l2 = argc > 1 && argv[1][0] == 'l';
    const long rows = l2 ? 8 * 1024 : 2 * 1024 * 1024;  // x 1024 B
    char* buf;
    hipMalloc(&buf, rows * 1024);
    hipMemset(buf, 1, rows * 1024);
    for (int wgs : {1, 2}) {
        for (int n : {4, 14}) {
            run<64>("16 rows x 64 B", buf, rows, wgs, n);
            run<128>("8 rows x 128 B", buf, rows, wgs, n);
            run<256>("4 rows x 256 B", buf, rows, wgs, n);
        }
    }
    return 0;
}
