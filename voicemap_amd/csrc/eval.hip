// n-shot k-way evaluation distances (voicemap/utils.py:159-206): per task, k class prototypes from n support
// embeddings each, distance to the query embedding, argmin.  The reference does this per task in float64 numpy
// after two predict() calls; here all tasks of an evaluation run go through one launch (fp64 accumulation).
#include "common.hpp"

namespace vm {

// grid = tasks, block = 64 (one wave): lane <-> class (k <= 64 per pass, looped above that).
__global__ __launch_bounds__(64) void nshot_kernel(const float* __restrict__ query, const float* __restrict__ support, int k,
                                                   int n, int E, int dist_kind, float* __restrict__ pred,
                                                   int32_t* __restrict__ argmin_out) {
    const int64_t task = blockIdx.x;
    const float* q = query + task * E;
    const float* s = support + task * (int64_t)k * n * E;
    double best = INFINITY;
    int besti = 0x7fffffff, bestn = 0;
    for (int cb = 0; cb < k; cb += 64) {
        const int cls = cb + (int)threadIdx.x;
        double d = INFINITY;
        if (cls < k) {
            const float* sc = s + (int64_t)cls * n * E;
            if (dist_kind == VM_DIST_EUCLIDEAN) {
                double acc = 0.0;
                for (int j = 0; j < E; ++j) {
                    double mean = 0.0;
                    for (int i = 0; i < n; ++i) mean += (double)sc[i * E + j];
                    mean /= (double)n;
                    const double df = (double)q[j] - mean;
                    acc += df * df;
                }
                d = sqrt(acc);
            } else {
                // per-sample magnitudes
                double magsum = 0.0, dot = 0.0, mu2 = 0.0, q2 = 0.0;
                for (int j = 0; j < E; ++j) q2 += (double)q[j] * (double)q[j];
                for (int i = 0; i < n; ++i) {
                    double m2 = 0.0;
                    for (int j = 0; j < E; ++j) m2 += (double)sc[i * E + j] * (double)sc[i * E + j];
                    magsum += sqrt(m2);
                }
                for (int j = 0; j < E; ++j) {
                    double mu = 0.0;  // mean unit vector component
                    for (int i = 0; i < n; ++i) {
                        double m2 = 0.0;
                        for (int jj = 0; jj < E; ++jj) m2 += (double)sc[i * E + jj] * (double)sc[i * E + jj];
                        mu += (double)sc[i * E + j] / sqrt(m2);
                    }
                    mu /= (double)n;
                    dot += (double)q[j] * mu;
                    mu2 += mu * mu;
                }
                if (dist_kind == VM_DIST_COSINE) {
                    d = 1.0 - dot / (sqrt(q2) * sqrt(mu2));
                } else {
                    d = -(dot * (magsum / (double)n));
                }
            }
            pred[task * k + cls] = (float)d;
        }
        // wave argmin, first minimum wins; a NaN distance orders below everything, like numpy.argmin (utils.py:200 takes
        // np.argmin of the distances: the first NaN if there is one)
        double dv = (cls < k && d != d) ? -INFINITY : d;
        int dn = (cls < k && d != d) ? 1 : 0;
        int di = cls < k ? cls : 0x7fffffff;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double od = __shfl_xor(dv, o, 64);
            const int on = __shfl_xor(dn, o, 64);
            const int oi = __shfl_xor(di, o, 64);
            if (on > dn || (on == dn && (od < dv || (od == dv && oi < di)))) {
                dv = od;
                dn = on;
                di = oi;
            }
        }
        if (dn > bestn || (dn == bestn && (dv < best || (dv == best && di < besti)))) {
            best = dv;
            bestn = dn;
            besti = di;
        }
    }
    if (threadIdx.x == 0) argmin_out[task] = besti;
}

}  // namespace vm

using namespace vm;

extern "C" int vm_nshot_distances(const float* query, const float* support, int64_t tasks, int k, int n, int E, int dist_kind,
                                  float* pred, int32_t* argmin, void* stream) {
    VM_REQUIRE(query && support && pred && argmin, "vm_nshot_distances: null pointer");
    VM_REQUIRE(tasks > 0 && k > 0 && n > 0 && E > 0, "vm_nshot_distances: bad sizes");
    VM_REQUIRE(dist_kind >= VM_DIST_EUCLIDEAN && dist_kind <= VM_DIST_DOT,
               "vm_nshot_distances: Distance must be in (euclidean, cosine, dot_product)");
    hipLaunchKernelGGL(nshot_kernel, dim3((unsigned)tasks), dim3(64), 0, (hipStream_t)stream, query, support, k, n, E, dist_kind,
                       pred, argmin);
    return check_launch("vm_nshot_distances");
}
