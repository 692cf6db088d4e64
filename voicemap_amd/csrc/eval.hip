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

// ------------------------------------------------------------------------------------------------------------------------------
// Batched evaluation over a CACHED embedding matrix (SURVEY 8f.2 / BASELINE.json config 5: "batched embedding forward +
// pairwise-distance matrix over train-clean-360, sharded").  The reference embeds k*n + 1 windows per task with two predict()
// calls (voicemap/utils.py:121-212, experiments/k_way_accuracy.py:52-69); with an evaluation set whose crops are deterministic
// (stochastic=False, the reference's validation datasets) every file has ONE embedding, so the corpus is embedded once into an
// (N, E) fp32 matrix and a task is k*n + 1 row indices.
// ------------------------------------------------------------------------------------------------------------------------------
namespace vm {

constexpr int NSI_MAX_EL = 4;  // elements of an embedding per lane: E <= 256

// vm_nshot_indexed: one wave per task, lane <-> embedding component(s); the arithmetic of voicemap/utils.py:159-206 in float64
// (per-class mean of the support embeddings -> L2; mean of the unit vectors -> cosine; mean magnitude x mean unit vector -> negative
// dot product), the rows gathered from the cached matrix by index.
__global__ __launch_bounds__(256) void nshot_indexed_kernel(const float* __restrict__ emb, int64_t n_rows, const int32_t* __restrict__ query_idx,
                                                            const int32_t* __restrict__ support_idx, int64_t tasks, int k, int n, int E,
                                                            int dist_kind, float* __restrict__ pred, int32_t* __restrict__ argmin_out) {
    const int lane = threadIdx.x & 63;
    const int64_t task = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (task >= tasks) return;
    const int EL = (E + 63) / 64;
    double q[NSI_MAX_EL];
    // an index outside [0, n_rows) never leaves the matrix: it is clamped (and the task is then simply wrong -- the host checks ranges)
    auto row_of = [&](int32_t i) { return emb + (int64_t)(i < 0 ? 0 : (i >= n_rows ? n_rows - 1 : i)) * E; };
    const float* qrow = row_of(query_idx[task]);
    double q2 = 0.0;
#pragma unroll
    for (int u = 0; u < NSI_MAX_EL; ++u) {
        const int e = lane + 64 * u;
        q[u] = (u < EL && e < E) ? (double)qrow[e] : 0.0;
        q2 += q[u] * q[u];
    }
    if (dist_kind != VM_DIST_EUCLIDEAN) q2 = wave_sum_d(q2);
    const int32_t* sidx = support_idx + task * (int64_t)k * n;
    double best = INFINITY;
    int besti = 0x7fffffff, bestn = 0;
    for (int cls = 0; cls < k; ++cls) {
        double acc[NSI_MAX_EL];
#pragma unroll
        for (int u = 0; u < NSI_MAX_EL; ++u) acc[u] = 0.0;
        double magsum = 0.0;
        for (int i = 0; i < n; ++i) {
            const float* srow = row_of(sidx[cls * n + i]);
            double v[NSI_MAX_EL], m2 = 0.0;
#pragma unroll
            for (int u = 0; u < NSI_MAX_EL; ++u) {
                const int e = lane + 64 * u;
                v[u] = (u < EL && e < E) ? (double)srow[e] : 0.0;
                m2 += v[u] * v[u];
            }
            if (dist_kind == VM_DIST_EUCLIDEAN) {
#pragma unroll
                for (int u = 0; u < NSI_MAX_EL; ++u) acc[u] += v[u];
            } else {
                const double mag = sqrt(wave_sum_d(m2));
                magsum += mag;
#pragma unroll
                for (int u = 0; u < NSI_MAX_EL; ++u) acc[u] += v[u] / mag;
            }
        }
        double d;
        if (dist_kind == VM_DIST_EUCLIDEAN) {
            double s = 0.0;
#pragma unroll
            for (int u = 0; u < NSI_MAX_EL; ++u) {
                const double df = q[u] - acc[u] / (double)n;
                s += df * df;
            }
            d = sqrt(wave_sum_d(s));
        } else {
            double dot = 0.0, mu2 = 0.0;
#pragma unroll
            for (int u = 0; u < NSI_MAX_EL; ++u) {
                const double mu = acc[u] / (double)n;
                dot += q[u] * mu;
                mu2 += mu * mu;
            }
            dot = wave_sum_d(dot);
            mu2 = wave_sum_d(mu2);
            d = dist_kind == VM_DIST_COSINE ? 1.0 - dot / (sqrt(q2) * sqrt(mu2)) : -(dot * (magsum / (double)n));
        }
        if (lane == 0 && pred != nullptr) pred[task * k + cls] = (float)d;
        // first minimum wins; a NaN distance orders below everything, like numpy.argmin (utils.py:200)
        const int dn = d != d ? 1 : 0;
        const double dv = dn ? -INFINITY : d;
        if (dn > bestn || (dn == bestn && dv < best)) {
            best = dv;
            bestn = dn;
            besti = cls;
        }
    }
    if (lane == 0) argmin_out[task] = besti;
}

// vm_pairdist_argmin: the (M, N) distance matrix between a block of query rows and the whole reference matrix, fp32, with the
// nearest reference row per query (first minimum; the query's own row can be excluded).  Workgroup = 16 x 16 threads, tile 64
// queries x 64 references, thread (ty, tx) owns queries 4ty..4ty+3 x references 4tx..4tx+3; the query tile stays in LDS for the
// whole walk over the references (split over blockIdx.y), reference tiles are staged 64 components at a time.  The euclidean
// distance is the direct form sqrt(sum (a - b)^2) -- no ||a||^2 + ||b||^2 - 2ab cancellation -- 2 VALU instructions per
// component pair: VALU-bound, ~5 ms for 13 K x 104 K x 64 (one rank's share of train-clean-360 at cfg-A's E).
constexpr int PD_T = 64, PD_EC = 64, PD_MAX_E = 256;

__global__ __launch_bounds__(256) void rowsq_kernel(const float* __restrict__ x, int64_t rows, int E, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float s = 0.f;
    for (int e = lane; e < E; e += 64) s = fmaf(x[r * E + e], x[r * E + e], s);
    s = wave_sum(s);
    if (lane == 0) out[r] = s;
}

__global__ __launch_bounds__(256) void pairdist_kernel(const float* __restrict__ q, const float* __restrict__ ref, int64_t M, int64_t N,
                                                       int E, int dist_kind, int64_t q_row0, const float* __restrict__ qsq,
                                                       const float* __restrict__ rsq, int tiles_per_split, float* __restrict__ dist,
                                                       float* __restrict__ part_val, int32_t* __restrict__ part_idx) {
    extern __shared__ __attribute__((aligned(16))) float pd_lds[];
    const int EP = ((E + 3) / 4) * 4;     // query rows are zero-extended to a multiple of 4 components
    const int QP = EP + 4;                // row pitch of the query tile (floats)
    constexpr int RP = PD_EC + 4;         // ... of a reference chunk
    float* qs = pd_lds;                   // [64][QP]
    float* rs = qs + PD_T * QP;           // [64][RP]
    float* red_v = rs + PD_T * RP;        // [64][16]
    int* red_i = reinterpret_cast<int*>(red_v + PD_T * 16);
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int64_t m0 = (int64_t)blockIdx.x * PD_T;
    for (int i = tid; i < PD_T * EP; i += 256) {
        const int r = i / EP, e = i - r * EP;
        qs[r * QP + e] = (m0 + r < M && e < E) ? q[(m0 + r) * E + e] : 0.f;
    }
    float qn[4], bestv[4];
    int besti[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + ty * 4 + i;
        qn[i] = (dist_kind == VM_DIST_COSINE && m < M) ? sqrtf(qsq[m]) : 1.f;
        bestv[i] = INFINITY;
        besti[i] = 0x7fffffff;
    }
    const int n_tiles = (int)((N + PD_T - 1) / PD_T);
    const int t_lo = blockIdx.y * tiles_per_split;
    const int t_hi = min(n_tiles, t_lo + tiles_per_split);
    for (int t = t_lo; t < t_hi; ++t) {
        const int64_t n0 = (int64_t)t * PD_T;
        float acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
        for (int ec = 0; ec < EP; ec += PD_EC) {
            const int ew = min(PD_EC, EP - ec);
            __syncthreads();  // the previous chunk's readers are done (and, the first time, the query tile is complete)
            for (int i = tid; i < PD_T * PD_EC; i += 256) {
                const int r = i >> 6, e = i & 63;
                rs[r * RP + e] = (n0 + r < N && ec + e < E) ? ref[(n0 + r) * E + ec + e] : 0.f;
            }
            __syncthreads();
            for (int e = 0; e < ew; e += 4) {
                f32x4 qv[4], rv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) qv[i] = *reinterpret_cast<const f32x4*>(qs + (ty * 4 + i) * QP + ec + e);
#pragma unroll
                for (int j = 0; j < 4; ++j) rv[j] = *reinterpret_cast<const f32x4*>(rs + (tx * 4 + j) * RP + e);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            if (dist_kind == VM_DIST_EUCLIDEAN) {
                                const float d = qv[i][c] - rv[j][c];
                                acc[i][j] = fmaf(d, d, acc[i][j]);
                            } else {
                                acc[i][j] = fmaf(qv[i][c], rv[j][c], acc[i][j]);
                            }
                        }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t m = m0 + ty * 4 + i;
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t nn = n0 + tx * 4 + j;
                float d;
                if (dist_kind == VM_DIST_EUCLIDEAN) {
                    d = sqrtf(acc[i][j]);
                } else if (dist_kind == VM_DIST_COSINE) {
                    d = 1.f - acc[i][j] / (qn[i] * sqrtf(nn < N ? rsq[nn] : 1.f));
                } else {
                    d = -acc[i][j];
                }
                o[j] = d;
                if (nn < N && m < M && (q_row0 < 0 || nn != q_row0 + m) && d < bestv[i]) {  // ascending nn: the first minimum is kept
                    bestv[i] = d;
                    besti[i] = (int)nn;
                }
            }
            if (dist != nullptr && m < M) {
                const int64_t nn = n0 + tx * 4;
                if (nn + 3 < N && (N & 3) == 0) {
                    *reinterpret_cast<f32x4*>(dist + m * N + nn) = o;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (nn + j < N) dist[m * N + nn + j] = o[j];
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        red_v[(ty * 4 + i) * 16 + tx] = bestv[i];
        red_i[(ty * 4 + i) * 16 + tx] = besti[i];
    }
    __syncthreads();
    if (tid < PD_T && m0 + tid < M) {
        float bv = INFINITY;
        int bi = 0x7fffffff;
        for (int x = 0; x < 16; ++x) {
            const float v = red_v[tid * 16 + x];
            const int ix = red_i[tid * 16 + x];
            if (v < bv || (v == bv && ix < bi)) {
                bv = v;
                bi = ix;
            }
        }
        part_val[(int64_t)blockIdx.y * M + m0 + tid] = bv;
        part_idx[(int64_t)blockIdx.y * M + m0 + tid] = bi;
    }
}

__global__ __launch_bounds__(256) void pairdist_final_kernel(const float* __restrict__ part_val, const int32_t* __restrict__ part_idx,
                                                             int64_t M, int splits, float* __restrict__ best_val,
                                                             int32_t* __restrict__ best_idx) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    float bv = INFINITY;
    int bi = 0x7fffffff;
    for (int s = 0; s < splits; ++s) {  // splits cover ascending reference ranges: strict < keeps the first minimum
        const float v = part_val[(int64_t)s * M + m];
        if (v < bv) {
            bv = v;
            bi = part_idx[(int64_t)s * M + m];
        }
    }
    best_val[m] = bv;
    best_idx[m] = bi == 0x7fffffff ? -1 : bi;
}

static int pd_splits(int64_t M, int64_t N) {
    const int64_t mt = (M + PD_T - 1) / PD_T, nt = (N + PD_T - 1) / PD_T;
    int64_t s = (2048 + mt - 1) / mt;  // aim for >= 2048 workgroups (8 per CU)
    if (s > nt) s = nt;
    if (s < 1) s = 1;
    return (int)s;
}

}  // namespace vm

extern "C" int vm_nshot_indexed(const float* emb, int64_t n_rows, const int32_t* query_idx, const int32_t* support_idx, int64_t tasks,
                                int k, int n, int E, int dist_kind, float* pred, int32_t* argmin, void* stream) {
    VM_REQUIRE(emb && query_idx && support_idx && argmin, "vm_nshot_indexed: null pointer");
    VM_REQUIRE(tasks > 0 && k > 0 && n > 0 && E > 0 && n_rows > 0, "vm_nshot_indexed: bad sizes");
    VM_REQUIRE(E <= 64 * vm::NSI_MAX_EL, "vm_nshot_indexed: embedding dimension %d > %d", E, 64 * vm::NSI_MAX_EL);
    VM_REQUIRE(dist_kind >= VM_DIST_EUCLIDEAN && dist_kind <= VM_DIST_DOT,
               "vm_nshot_indexed: Distance must be in (euclidean, cosine, dot_product)");
    VM_REQUIRE((tasks + 3) / 4 < (1LL << 31), "vm_nshot_indexed: too many tasks for one launch");
    hipLaunchKernelGGL(vm::nshot_indexed_kernel, dim3((unsigned)((tasks + 3) / 4)), dim3(256), 0, (hipStream_t)stream, emb, n_rows, query_idx,
                       support_idx, tasks, k, n, E, dist_kind, pred, argmin);
    return vm::check_launch("vm_nshot_indexed");
}

extern "C" int64_t vm_pairdist_workspace_bytes(int64_t M, int64_t N) {
    if (M <= 0 || N <= 0) return 0;
    return (int64_t)vm::pd_splits(M, N) * M * 8 + (M + N) * 4 + 256;
}

extern "C" int vm_pairdist_argmin(const float* q, const float* ref, int64_t M, int64_t N, int E, int dist_kind, int64_t q_row0,
                                  float* dist, float* best_val, int32_t* best_idx, void* ws, void* stream) {
    using namespace vm;
    VM_REQUIRE(q && ref && best_val && best_idx && ws, "vm_pairdist_argmin: null pointer");
    VM_REQUIRE(M > 0 && N > 0 && E > 0 && E <= PD_MAX_E, "vm_pairdist_argmin: bad sizes (E <= %d)", PD_MAX_E);
    VM_REQUIRE(N < (1LL << 31) && (M + PD_T - 1) / PD_T < (1LL << 31), "vm_pairdist_argmin: matrix too large for one launch");
    VM_REQUIRE(dist_kind >= VM_DIST_EUCLIDEAN && dist_kind <= VM_DIST_DOT,
               "vm_pairdist_argmin: Distance must be in (euclidean, cosine, dot_product)");
    const int splits = pd_splits(M, N);
    float* part_val = (float*)ws;
    int32_t* part_idx = (int32_t*)(part_val + (int64_t)splits * M);
    float* qsq = (float*)(part_idx + (int64_t)splits * M);
    float* rsq = qsq + M;
    hipStream_t st = (hipStream_t)stream;
    if (dist_kind == VM_DIST_COSINE) {
        hipLaunchKernelGGL(rowsq_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, q, M, E, qsq);
        hipLaunchKernelGGL(rowsq_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, ref, N, E, rsq);
    }
    const int EP = ((E + 3) / 4) * 4;
    const size_t lds = (size_t)(PD_T * (EP + 4) + PD_T * (PD_EC + 4) + PD_T * 16 * 2) * 4;
    const int n_tiles = (int)((N + PD_T - 1) / PD_T);
    const int tps = (n_tiles + splits - 1) / splits;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pairdist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(pairdist_kernel, dim3((unsigned)((M + PD_T - 1) / PD_T), (unsigned)splits), dim3(256), lds, st, q, ref, M, N, E,
                       dist_kind, q_row0, qsq, rsq, tps, dist, part_val, part_idx);
    hipLaunchKernelGGL(pairdist_final_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, part_val, part_idx, M, splits, best_val,
                       best_idx);
    return check_launch("vm_pairdist_argmin");
}
