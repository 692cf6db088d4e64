// HBM read rate of the access pattern a GEMM's A-operand staging produces: a wave instruction fetches 8 rows x CHUNK bytes
// (16 bytes per lane), rows `pitch` bytes apart, and a workgroup walks the row's K extent chunk by chunk -- against the same
// bytes read as one contiguous stream.   hipcc --offload-arch=gfx950 -O3 strided_read_probe.hip -o strided_read_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// mode 0: contiguous (lane-linear over the whole buffer); mode 1: tile pattern with 128-byte chunks (8 lanes per row);
// mode 2: 256-byte chunks (16 lanes per row); mode 3: 512-byte chunks
template <int MODE>
__global__ __launch_bounds__(256) void rd(const char* __restrict__ a, int64_t rows, int pitch, uint32_t* out) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    u32x4 acc = {0, 0, 0, 0};
    if (MODE == 0) {
        const int64_t total = rows * pitch / 16;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(a + i * 16);
            acc ^= v;
        }
    } else {
        constexpr int CH = MODE == 1 ? 128 : (MODE == 2 ? 256 : 512);
        constexpr int LPR = CH / 16, RPI = 64 / LPR;  // lanes per row, rows per wave instruction
        // a workgroup owns 256 consecutive rows (like a 256-row GEMM tile), its 4 waves 64 rows each
        for (int64_t t = blockIdx.x; t < rows / 256; t += gridDim.x) {
            const char* base = a + (t * 256 + w * 64) * pitch;
            for (int k = 0; k < pitch; k += CH) {
#pragma unroll
                for (int q = 0; q < 64 / RPI; ++q) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(base + (int64_t)(q * RPI + lane / LPR) * pitch + k + (lane % LPR) * 16);
                    acc ^= v;
                }
            }
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[threadIdx.x] = acc[0];
}

int main() {
    const int64_t rows = 192000;
    const int pitch = 2304;  // K = 1152 bf16
    char* a;
    uint32_t* out;
    hipMalloc(&a, rows * pitch);
    hipMalloc(&out, 4096);
    hipMemset(a, 1, rows * pitch);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char* names[] = {"contiguous stream", "256-row tiles, 128 B per row and step", "256-row tiles, 256 B per row and step", "256-row tiles, 512 B per row and step"};
    for (int wgs : {256, 1024, 2048}) {
        for (int mode = 0; mode < 4; ++mode) {
            auto launch = [&]() {
                switch (mode) {
                    case 0: hipLaunchKernelGGL(rd<0>, dim3(wgs), dim3(256), 0, 0, a, rows, pitch, out); break;
                    case 1: hipLaunchKernelGGL(rd<1>, dim3(wgs), dim3(256), 0, 0, a, rows, pitch, out); break;
                    case 2: hipLaunchKernelGGL(rd<2>, dim3(wgs), dim3(256), 0, 0, a, rows, pitch, out); break;
                    default: hipLaunchKernelGGL(rd<3>, dim3(wgs), dim3(256), 0, 0, a, rows, pitch, out); break;
                }
            };
            for (int i = 0; i < 3; ++i) launch();
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%4d workgroups  %-42s %7.1f us  %6.2f TB/s\n", wgs, names[mode], ms * 100, rows * pitch / (ms * 1e-4) / 1e12);
        }
    }
    return 0;
}
