// LDS read throughput per CU for the fragment-read instructions the GEMM kernels use: ds_read_b128 (forward / dgrad fragments),
// ds_read_b64 and ds_read_b64_tr_b16 (the transposing read of the wgrad kernels and of the statistics epilogues), with the
// conflict-free address patterns of those kernels.  One workgroup per CU, `waves` waves, each wave issues `N` reads per loop
// iteration (independent destinations), cycles from s_memtime.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(512) void k(uint32_t* out, int iters, long long* cyc) {
    __shared__ __attribute__((aligned(1024))) char lds[64 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid * 16; i < 64 * 1024; i += blockDim.x * 16) *reinterpret_cast<u32x4*>(lds + i) = u32x4{(uint32_t)i, 1, 2, 3};
    __syncthreads();
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)lds;
    uint32_t addr;
    if (MODE == 0) {
        // nt2r fragment read: lane (r = lane & 31, kh = lane >> 5) reads 16 bytes of row r, 64-byte rows, xor swizzle by (row >> 2) & 3
        const int r = lane & 31, kh = lane >> 5;
        addr = lds0 + w * 2048 + r * 64 + (((kh) ^ ((r >> 2) & 3)) * 16);
    } else {
        // tn8x / statistics transposing read: lane (kh, lg, li): row kh * 8 + (li >> 2), 128-byte rows, halves swapped by (row >> 1) & 1
        const int li = lane & 15, lg = (lane >> 4) & 1, kh = lane >> 5;
        const int rowl = kh * 8 + (li >> 2);
        addr = lds0 + w * 4096 + rowl * 128 + ((((rowl >> 1) & 1)) * 64) + lg * 32 + (li & 3) * 8;
    }
    u32x4 a[8];
    u32x2 b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = u32x4{0, 0, 0, 0}; b[j] = u32x2{0, 0}; }
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[j]) : "v"(addr), "n"(j * 4096 % 32768));
            if (MODE == 1) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(b[j]) : "v"(addr), "n"(j * 2048 % 16384));
            if (MODE == 2) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(b[j]) : "v"(addr), "n"(j * 2048 % 16384));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += a[j][0] + a[j][3] + b[j][0] + b[j][1];
    if (acc == 0x12345678u) out[tid] = acc;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int waves, int bytes_per_lane) {
    uint32_t* out;
    long long* cyc;
    hipMalloc(&out, 4096);
    hipMallocManaged(&cyc, 256 * 8);
    const int iters = 4000;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(waves * 64), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(waves * 64), 0, 0, out, iters, cyc);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    double avg = 0;
    for (int i = 0; i < 256; ++i) avg += (double)cyc[i];
    avg /= 256;
    // s_memtime counts at 100 MHz on this part; report bytes per memtime tick and let the caller compare modes
    const double bytes = (double)iters * 8 * waves * 64 * bytes_per_lane;
    printf("%-22s waves %d: %10.0f ticks  %8.1f bytes/tick/CU   kernel %.3f ms -> %.1f GB/s per CU (%.1f B/clk at 2.4 GHz)\n", name, waves, avg,
           bytes / avg, ms, bytes / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 2.4e9);
    hipFree(out);
    hipFree(cyc);
}

int main() {
    for (int waves : {4, 8}) {
        run<0>("ds_read_b128", waves, 16);
        run<1>("ds_read_b64", waves, 8);
        run<2>("ds_read_b64_tr_b16", waves, 8);
    }
    return 0;
}
