// Does the 256 MiB Infinity Cache keep what a streaming kernel just wrote (or read), and what does a consumer gain by walking
// the tensor in the OPPOSITE direction of its producer (most recently touched data first) instead of the same direction (LRU
// streaming: the head of the tensor is gone by the time the tail has been written)?
//   producer: writes (or reads) `chunks` contiguous chunks of CH bytes, workgroup b <-> chunk b (dispatch order = ascending)
//   consumer: reads chunk b (same direction) or chunk chunks-1-b (reversed), 16 B per lane, sums into a sink
// hipcc --offload-arch=gfx950 -O3 mall_order_probe.hip -o mall_order_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// SPLIT workgroups per chunk (a chunk = one window); per_chunk16 below is the per-WORKGROUP piece, consecutive workgroups are
// consecutive pieces, so ascending dispatch order = ascending addresses
__global__ __launch_bounds__(256) void writer(u32x4* p, long per_chunk16, uint32_t v) {
    u32x4* c = p + (long)blockIdx.x * per_chunk16;
    for (long i = threadIdx.x; i < per_chunk16; i += 256) c[i] = u32x4{v, v + 1, v + 2, (uint32_t)i};
}

__global__ __launch_bounds__(256) void reader(const u32x4* p, long per_chunk16, int reversed, uint32_t* sink) {
    const long b = reversed ? (long)gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const u32x4* c = p + b * per_chunk16;
    uint32_t acc = 0;
    for (long i = threadIdx.x; i < per_chunk16; i += 1024) {
        u32x4 a = c[i], b2 = i + 256 < per_chunk16 ? c[i + 256] : u32x4{0, 0, 0, 0};
        u32x4 c2 = i + 512 < per_chunk16 ? c[i + 512] : u32x4{0, 0, 0, 0}, d = i + 768 < per_chunk16 ? c[i + 768] : u32x4{0, 0, 0, 0};
        acc += a.x ^ a.w ^ b2.y ^ c2.z ^ d.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// BN-forward-like pass: read a chunk, write half as many bytes (same chunk index of the destination)
__global__ __launch_bounds__(256) void halver(const u32x4* p, u32x4* q, long per_chunk16, int reversed) {
    const long b = reversed ? (long)gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const u32x4* c = p + b * per_chunk16;
    u32x4* o = q + b * (per_chunk16 / 2);
    for (long i = threadIdx.x; i < per_chunk16 / 2; i += 256) {
        u32x4 a = c[2 * i], b2 = c[2 * i + 1];
        o[i] = u32x4{a.x ^ b2.x, a.y ^ b2.y, a.z ^ b2.z, a.w ^ b2.w};
    }
}

int main() {
    const long CH = 1536 * 1024;  // one window of a (3000 x 256) bf16 tensor
    const int SPLIT = 16;
    const long per16 = CH / 16 / SPLIT;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    uint32_t* sink;
    hipMalloc(&sink, 64);
    const long maxb = 1024L * 1024 * 1024;
    u32x4 *buf, *buf2, *flush;
    hipMalloc(&buf, maxb);
    hipMalloc(&buf2, maxb / 2);
    hipMalloc(&flush, maxb);
    printf("%8s %10s | consumer read, GB/s: %10s %10s | after READ producer: %10s %10s | halving pass GB/s(read+write): %8s %8s\n", "MB", "producer", "same-dir",
           "reversed", "same-dir", "reversed", "same", "rev");
    for (long mb : {96L, 147L, 196L, 295L, 393L, 590L}) {
        const int chunks = (int)(mb * 1024 * 1024 / CH) * SPLIT;
        const double bytes = (double)chunks * CH / SPLIT;
        double res[6];
        for (int mode = 0; mode < 6; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                // evict: stream 1 GiB through the caches
                hipLaunchKernelGGL(writer, dim3((unsigned)(maxb / CH * SPLIT)), dim3(256), 0, 0, flush, per16, 7u);
                if (mode < 2 || mode >= 4) {
                    hipLaunchKernelGGL(writer, dim3(chunks), dim3(256), 0, 0, buf, per16, (uint32_t)rep);
                } else {
                    hipLaunchKernelGGL(writer, dim3(chunks), dim3(256), 0, 0, buf, per16, (uint32_t)rep);
                    hipLaunchKernelGGL(writer, dim3((unsigned)(maxb / CH * SPLIT)), dim3(256), 0, 0, flush, per16, 7u);
                    hipLaunchKernelGGL(reader, dim3(chunks), dim3(256), 0, 0, buf, per16, 0, sink);
                }
                hipEventRecord(e0);
                if (mode < 4) {
                    hipLaunchKernelGGL(reader, dim3(chunks), dim3(256), 0, 0, buf, per16, mode & 1, sink);
                } else {
                    hipLaunchKernelGGL(halver, dim3(chunks), dim3(256), 0, 0, buf, buf2, per16, mode & 1);
                }
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            res[mode] = (mode < 4 ? bytes : 1.5 * bytes) / (best * 1e-3) / 1e9;
        }
        printf("%8ld %10s | %32.0f %10.0f | %30.0f %10.0f | %39.0f %8.0f\n", mb, "write", res[0], res[1], res[2], res[3], res[4], res[5]);
    }
    return 0;
}
