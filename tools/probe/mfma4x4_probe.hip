// Lane / register layout of v_mfma_f32_4x4x4_16B_f16 on gfx950 (16 independent 4x4x4 blocks per instruction), checked against two
// hypotheses on random operands:   hipcc --offload-arch=gfx950 -O2 mfma4x4_probe.hip -o mfma4x4_probe && ./mfma4x4_probe
//   H1: D[lane 4b + j][reg i] = sum_k A[lane 4b + i][k] * B[lane 4b + j][k]        H2: the transpose (lane <-> i, reg <-> j)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const h4* a, const h4* b, f4* d) {
    const int l = threadIdx.x;
    f4 c = {0.f, 0.f, 0.f, 0.f};
    d[l] = __builtin_amdgcn_mfma_f32_4x4x4f16(a[l], b[l], c, 0, 0, 0);
}
int main() {
    h4 ha[64], hb[64];
    float fa[64][4], fb[64][4];
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 4; ++e) {
            fa[l][e] = (float)((rand() % 17) - 8) * 0.25f;
            fb[l][e] = (float)((rand() % 13) - 6) * 0.5f;
            ha[l][e] = (_Float16)fa[l][e];
            hb[l][e] = (_Float16)fb[l][e];
        }
    h4 *da, *db;
    f4* dd;
    hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dd, 64 * sizeof(f4));
    hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd);
    f4 out[64];
    hipMemcpy(out, dd, sizeof(out), hipMemcpyDeviceToHost);
    int ok1 = 1, ok2 = 1;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int b = l / 4, j = l % 4;
            float h1 = 0.f, h2 = 0.f;
            for (int kk = 0; kk < 4; ++kk) {
                h1 += fa[4 * b + r][kk] * fb[4 * b + j][kk];
                h2 += fa[4 * b + j][kk] * fb[4 * b + r][kk];
            }
            if (out[l][r] != h1) ok1 = 0;
            if (out[l][r] != h2) ok2 = 0;
        }
    printf("v_mfma_f32_4x4x4_16B_f16: H1 (lane = column j of B, reg = row i of A) %s, H2 (transpose) %s\n", ok1 ? "MATCHES" : "no", ok2 ? "MATCHES" : "no");
    printf("lane 5: %g %g %g %g\n", out[5][0], out[5][1], out[5][2], out[5][3]);
    return 0;
}
