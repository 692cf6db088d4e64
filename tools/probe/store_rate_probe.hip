// Per-CU global store throughput: is the ~7-10 B/clk/CU seen in GEMM epilogues a per-CU limit or a chip (HBM/fabric) limit?
// `nwg` workgroups of 256 threads each write `kb` KB (16 B per lane, whole 256-byte row segments, 4 rows per wave-instruction with
// a row pitch of `pitch` bytes) -- 1 workgroup per CU when nwg <= 256.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__global__ __launch_bounds__(256) void st(u32x4* out, long per_wg_bytes, int pitch16, int reps, long long* cyc) {
    const int tid = threadIdx.x, c = tid & 15, rg = tid >> 4;
    u32x4 v = {(uint32_t)tid, 1u, 2u, 3u};
    char* base = (char*)out + (long)blockIdx.x * per_wg_bytes;
    const long rows = per_wg_bytes / ((long)pitch16 * 16);
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; ++rep)
        for (long r0 = 0; r0 + 16 <= rows; r0 += 16) {
            *reinterpret_cast<u32x4*>(base + (r0 + rg) * pitch16 * 16 + c * 16) = v;
            v.x += 1;
        }
    __builtin_amdgcn_s_waitcnt(0);
    long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    u32x4* buf;
    const long total = 1024L * 1024 * 1024;
    hipMalloc(&buf, total);
    long long* cyc;
    hipMallocManaged(&cyc, 8192 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("%6s %8s %8s | %10s %12s %12s\n", "nwg", "KB/wg", "pitch", "us", "GB/s total", "B/clk/WG(avg)");
    for (int pitch : {256, 768}) {
        for (int nwg : {8, 32, 128, 256, 512, 1024, 2048}) {
            const long per = pitch == 256 ? 512 * 1024 : 3 * 512 * 1024;  // address range per WG; bytes actually written = per * 256 / pitch
            if ((long)nwg * per > total) continue;
            const int reps = 4;
            hipLaunchKernelGGL(st, dim3(nwg), dim3(256), 0, 0, buf, per, pitch / 16, 1, cyc);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(st, dim3(nwg), dim3(256), 0, 0, buf, per, pitch / 16, reps, cyc);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double bytes = (double)nwg * (per * 256.0 / pitch) * reps;
            double c = 0;
            for (int i = 0; i < nwg; ++i) c += (double)cyc[i];
            c /= nwg;
            printf("%6d %8ld %8d | %10.1f %12.0f %12.2f\n", nwg, per * 256 / pitch / 1024, pitch, ms * 1e3, bytes / (ms * 1e-3) / 1e9,
                   (per * 256.0 / pitch) * reps / c * 1.0);
        }
    }
    return 0;
}
