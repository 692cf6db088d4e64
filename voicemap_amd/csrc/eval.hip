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
// nearest reference row per query (first minimum; the query's own row can be excluded).  The euclidean distance is the direct form
// sqrt(sum (a - b)^2) -- no ||a||^2 + ||b||^2 - 2ab cancellation -- 2 VALU instructions per component pair: the kernel's bound is
// fp32 VALU issue (2 x M x N x E / 64 wave instructions; v_pk_* do not issue faster on gfx950).  Round 6, on the 13 002 x 104 014 x 64
// shard of BASELINE.json config 5 (profiles/r06_pairdist.txt): the round-5 kernel (16 x 16 threads, 4 x 4 register tiles, both operands
// from LDS) ran 8.9 ms -- it was LDS-bound: a thread's four ADJACENT reference rows put lanes tx, tx + 4, tx + 8, tx + 12 on the same
// banks (4-way conflicts on every reference read).  Rows 16 apart + register prefetch of the next stage: 5.5 ms, and the dot product
// at 3.5 ms showed the LDS still level with the VALU (8 reads per 64 component pairs).  Now the QUERIES are wave-uniform and come through
// the scalar path (pairdist_qt_kernel lays them out for it), lanes are references: one LDS read per 32 component pairs.
constexpr int PD_T = 64, PD_RT = 128, PD_EC = 64, PD_MAX_E = 256;   // queries per workgroup, references per stage, components per stage

__global__ __launch_bounds__(256) void rowsq_kernel(const float* __restrict__ x, int64_t rows, int E, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float s = 0.f;
    for (int e = lane; e < E; e += 64) s = fmaf(x[r * E + e], x[r * E + e], s);
    s = wave_sum(s);
    if (lane == 0) out[r] = s;
}

// q (M, E) -> qT: per group of 8 queries and per 4 components the 8 x 4 values back to back (128 bytes): what a wave of
// pairdist_kernel takes per component step through the SCALAR path.  [group][e4][query in group][4]; rows past M / components past E: 0.
__global__ __launch_bounds__(256) void pairdist_qt_kernel(const float* __restrict__ q, int64_t M, int E, int EP, float* __restrict__ qT) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t groups = (M + 7) / 8;
    if (i >= groups * 8 * EP) return;
    const int c = (int)(i & 3), qi = (int)((i >> 2) & 7);
    const int64_t r = i >> 5;
    const int e4 = (int)(r % (EP / 4));
    const int64_t g = r / (EP / 4);
    const int64_t m = g * 8 + qi;
    const int e = e4 * 4 + c;
    qT[i] = (m < M && e < E) ? q[m * E + e] : 0.f;
}

// Workgroup = 8 waves; wave w owns the 8 queries m0 + 8 w .. + 7 -- WAVE-UNIFORM, so their components arrive through scalar loads
// (SGPR operands of the subtract / FMA: no LDS read, no vector register) -- and lane l the references n0 + l and n0 + l + 64 of a
// 128-reference stage, whose 64 components per stage it reads from LDS 16 bytes at a time: ONE LDS read per 32 component pairs.
template <int DIST>
__global__ __launch_bounds__(512) void pairdist_kernel(const float* __restrict__ qT, const float* __restrict__ ref, int64_t M, int64_t N,
                                                       int E, int64_t q_row0, const float* __restrict__ qsq,
                                                       const float* __restrict__ rsq, int tiles_per_split, float* __restrict__ dist,
                                                       float* __restrict__ part_val, int32_t* __restrict__ part_idx) {
    constexpr int RT = PD_RT, RP = PD_EC + 4;   // 128 references per stage; row pitch 68 floats = 17 bank quads
    __shared__ __attribute__((aligned(16))) float rs[RT * RP];
    const int EP = ((E + 3) / 4) * 4, E4 = EP / 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t m0 = (int64_t)blockIdx.x * PD_T + 8 * w;          // this wave's first query
    // its group's rows of qT (wave-uniform).  A wave past the last query (M not a multiple of 64) computes on the LAST group's rows --
    // qT holds ceil(M / 8) groups and nothing behind them -- and stores nothing (every use below is guarded by m < M).
    const int64_t last_group = (M - 1) >> 3;
    const float* qg = qT + ((m0 >> 3) < last_group ? (m0 >> 3) : last_group) * (int64_t)E4 * 32;
    float qn[8], bestv[8];
    int besti[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        qn[i] = (DIST == VM_DIST_COSINE && m0 + i < M) ? sqrtf(qsq[m0 + i]) : 1.f;
        bestv[i] = INFINITY;
        besti[i] = 0x7fffffff;
    }
    const int n_tiles = (int)((N + RT - 1) / RT);
    const int t_lo = blockIdx.y * tiles_per_split;
    const int t_hi = min(n_tiles, t_lo + tiles_per_split);
    const int nchunk = (EP + PD_EC - 1) / PD_EC;
    const int n_stage = (t_hi - t_lo) * nchunk;
    const bool vec = (E & 3) == 0;
    // stage s = (tile t_lo + s / nchunk, components (s % nchunk) * 64 ..): this thread's four 16-byte pieces, piece = tid + 512 k
    auto fetch = [&](int s, f32x4 (&v)[4]) {
        const int64_t n0 = (int64_t)(t_lo + s / nchunk) * RT;
        const int ec = (s % nchunk) * PD_EC;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int piece = tid + 512 * k, r = piece >> 4, c = (piece & 15) * 4;
            const int64_t row = n0 + r;
            const int col = ec + c;
            if (vec) {
                const bool ok = row < N && col < E;
                const f32x4 x = *reinterpret_cast<const f32x4*>(ref + (ok ? row * E + col : 0));
                v[k] = ok ? x : f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) v[k][u] = (row < N && col + u < E) ? ref[row * E + col + u] : 0.f;
            }
        }
    };
    auto stash = [&](const f32x4 (&v)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int piece = tid + 512 * k, r = piece >> 4, c = (piece & 15) * 4;
            *reinterpret_cast<f32x4*>(rs + r * RP + c) = v[k];
        }
    };
    f32x4 nxt[4];
    if (n_stage > 0) fetch(0, nxt);
    float acc[8][2];
    for (int s = 0; s < n_stage; ++s) {
        const int t = t_lo + s / nchunk, ck = s % nchunk;
        const int ec = ck * PD_EC;
        const int ew4 = min(PD_EC, EP - ec) / 4;
        const int64_t n0 = (int64_t)t * RT;
        __syncthreads();  // the previous stage's readers are done
        stash(nxt);
        if (s + 1 < n_stage) fetch(s + 1, nxt);   // in flight while this stage is consumed
        __syncthreads();
        if (ck == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i][0] = acc[i][1] = 0.f;
        }
        const float* qe = qg + (ec / 4) * 32;
#pragma unroll 2
        for (int e4 = 0; e4 < ew4; ++e4) {
            const f32x4 r0 = *reinterpret_cast<const f32x4*>(rs + lane * RP + e4 * 4);
            const f32x4 r1 = *reinterpret_cast<const f32x4*>(rs + (lane + 64) * RP + e4 * 4);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x4 qv = *reinterpret_cast<const f32x4*>(qe + e4 * 32 + i * 4);   // uniform address: a scalar load
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (DIST == VM_DIST_EUCLIDEAN) {
                        const float d0 = qv[c] - r0[c], d1 = qv[c] - r1[c];
                        acc[i][0] = fmaf(d0, d0, acc[i][0]);
                        acc[i][1] = fmaf(d1, d1, acc[i][1]);
                    } else {
                        acc[i][0] = fmaf(qv[c], r0[c], acc[i][0]);
                        acc[i][1] = fmaf(qv[c], r1[c], acc[i][1]);
                    }
                }
            }
        }
        if (ck != nchunk - 1) continue;
        float rn[2] = {1.f, 1.f};
        if (DIST == VM_DIST_COSINE) {
#pragma unroll
            for (int j = 0; j < 2; ++j) rn[j] = sqrtf(n0 + lane + 64 * j < N ? rsq[n0 + lane + 64 * j] : 1.f);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {   // ascending reference index within a lane: the first minimum is kept
            const int64_t nn = n0 + lane + 64 * j;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int64_t m = m0 + i;
                float d;
                if (DIST == VM_DIST_EUCLIDEAN) {
                    d = sqrtf(acc[i][j]);
                } else if (DIST == VM_DIST_COSINE) {
                    d = 1.f - acc[i][j] / (qn[i] * rn[j]);
                } else {
                    d = -acc[i][j];
                }
                if (dist != nullptr && m < M && nn < N) dist[m * N + nn] = d;   // a wave = 256 contiguous bytes of a row
                if (nn < N && m < M && (q_row0 < 0 || nn != q_row0 + m) && d < bestv[i]) {
                    bestv[i] = d;
                    besti[i] = (int)nn;
                }
            }
        }
    }
    // a lane's minimum is the first in ITS references; across lanes the smaller index wins a tie
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float bv = bestv[i];
        int bi = besti[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov < bv || (ov == bv && oi < bi)) {
                bv = ov;
                bi = oi;
            }
        }
        if (lane == 0 && m0 + i < M) {
            part_val[(int64_t)blockIdx.y * M + m0 + i] = bv;
            part_idx[(int64_t)blockIdx.y * M + m0 + i] = bi;
        }
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
    const int64_t mt = (M + PD_T - 1) / PD_T, nt = (N + PD_RT - 1) / PD_RT;
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
    // partial minima per split, the two squared-norm vectors, and the scalar-path copy of the queries (8-row groups x up to PD_MAX_E)
    return (int64_t)vm::pd_splits(M, N) * M * 8 + (M + N) * 4 + 256 + ((M + 7) / 8) * 8 * (int64_t)vm::PD_MAX_E * 4 + 256;
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
    // the scalar-path copy of the queries behind the squared norms (16-byte aligned)
    float* qT = (float*)(((uintptr_t)(rsq + N) + 255) & ~(uintptr_t)255);
    const int64_t qt_n = ((M + 7) / 8) * 8 * (int64_t)EP;
    hipLaunchKernelGGL(pairdist_qt_kernel, dim3((unsigned)((qt_n + 255) / 256)), dim3(256), 0, st, q, M, E, EP, qT);
    const int n_tiles = (int)((N + PD_RT - 1) / PD_RT);
    const int tps = (n_tiles + splits - 1) / splits;
    VM_REQUIRE((E & 3) != 0 || (((uintptr_t)ref) & 15) == 0, "vm_pairdist_argmin: ref must be 16-byte aligned when E %% 4 == 0");
    const dim3 grid((unsigned)((M + PD_T - 1) / PD_T), (unsigned)splits);
#define VM_PD(D) hipLaunchKernelGGL(pairdist_kernel<D>, grid, dim3(512), 0, st, qT, ref, M, N, E, q_row0, qsq, rsq, tps, dist, part_val, part_idx)
    if (dist_kind == VM_DIST_EUCLIDEAN) VM_PD(VM_DIST_EUCLIDEAN);
    else if (dist_kind == VM_DIST_COSINE) VM_PD(VM_DIST_COSINE);
    else VM_PD(VM_DIST_DOT);
#undef VM_PD
    hipLaunchKernelGGL(pairdist_final_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, part_val, part_idx, M, splits, best_val,
                       best_idx);
    return check_launch("vm_pairdist_argmin");
}
