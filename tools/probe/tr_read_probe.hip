// Probe of ds_read_b64_tr_b16 semantics on gfx950: LDS holds u16 value == its element index; every lane reads at byte
// address addr(l) and we print what each lane received.  hipcc --offload-arch=gfx950 tr_read_probe.hip -o tr_probe && ./tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void probe(int mode, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    int addr_elems;
    if (mode == 0) addr_elems = l * 4;                               // lane-linear: lane l reads elements 4l..4l+3
    else if (mode == 1) addr_elems = (l & 15) / 4 * 32 + (l & 3) * 4 + (l >> 4) * 16;  // [key][32ch] rows of 32 elems: key=(i>>2), group cols
    else addr_elems = (l >> 2) * 64 + (l & 3) * 4;                   // rows of 64 elements
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + addr_elems));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}
int main() {
    uint16_t* d;
    hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, mode, d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    return 0;
}
