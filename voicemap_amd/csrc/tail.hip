// Tail of the encoder and the siamese / classifier heads: GlobalMaxPool1D -> Dense(E) (voicemap/models.py:37-39),
// twin distance -> Dense(1, sigmoid) (models.py:55-69), contrastive / binary-cross-entropy loss
// (voicemap/utils.py:77-85, experiments/train_siamese.py:57) and softmax + categorical CE
// (experiments/train_classifier.py:112-115).  A few hundred KB of fp32 work per step: small kernels, fixed
// summation order, everything in fp32.
#include "common.hpp"

namespace vm {

// ---- GlobalMaxPool1D ------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void global_maxpool_fwd_kernel(const T* __restrict__ act, int64_t L, int C, int P,
                                                                 float* __restrict__ gmax, int32_t* __restrict__ gidx) {
    constexpr int VEC = Elem<T>::kVec;
    __shared__ float rv[256][VEC];
    __shared__ int ri[256][VEC];
    const int tid = threadIdx.x, RP = 256 / P, pl = tid % P, rl = tid / P;
    const int CV = C / VEC;
    const int64_t n = blockIdx.x;
    for (int cvb = 0; cvb < CV; cvb += P) {
        const int cv = cvb + pl;
        const bool cok = cv < CV;
        const int c0 = cv * VEC;
        float best[VEC];
        int bi[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            best[i] = -INFINITY;
            bi[i] = 0x7fffffff;
        }
        if (cok) {
            for (int64_t t = rl; t < L; t += RP) {
                const Vec16<T> v = load16<T>(act + (n * (L + 2) + 1 + t) * C + c0);
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const float y = v.get(i);
                    if (y > best[i] || bi[i] == 0x7fffffff) {
                        best[i] = y;
                        bi[i] = (int)t;
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            rv[tid][i] = best[i];
            ri[tid][i] = bi[i];
        }
        __syncthreads();
        if (rl == 0 && cok) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float b = rv[pl][i];
                int k = ri[pl][i];
                for (int r = 1; r < RP; ++r) {
                    const float y = rv[r * P + pl][i];
                    const int kk = ri[r * P + pl][i];
                    if (kk != 0x7fffffff && (k == 0x7fffffff || y > b || (y == b && kk < k))) {
                        b = y;
                        k = kk;
                    }
                }
                gmax[n * C + c0 + i] = b;
                gidx[n * C + c0 + i] = k;
            }
        }
        __syncthreads();
    }
}

template <typename T>
__global__ __launch_bounds__(256) void global_maxpool_bwd_kernel(const float* __restrict__ dg, const int32_t* __restrict__ gidx,
                                                                 int64_t total, int64_t L, int C, T* __restrict__ dp) {
    constexpr int VEC = Elem<T>::kVec;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int CV = C / VEC;
    const int cv = (int)(idx % CV);
    const int64_t r = idx / CV;
    const int64_t t = r % L, n = r / L;
    const int c0 = cv * VEC;
    Vec16<T> o;
#pragma unroll
    for (int i = 0; i < VEC; ++i) o.set(i, gidx[n * C + c0 + i] == (int)t ? dg[n * C + c0 + i] : 0.f);
    store16<T>(dp + (n * L + t) * C + c0, o);
}

// ---- Dense ------------------------------------------------------------------------------------------
// Tiny fp32 GEMMs (256 x 512 x 64 at cfg-A).  blockIdx.y carries the index of the broadcast operand's row, so
// that operand is read with wave-uniform (scalar) loads and the other one coalesced across the 64 lanes.
constexpr int DENSE_KS = 16;  // reduction slices per output in the dense kernels (one wave each)

__global__ __launch_bounds__(64 * DENSE_KS) void dense_fwd_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                                  const float* __restrict__ b, int n_in, int n_out,
                                                                  float* __restrict__ out) {
    // 64 outputs x DENSE_KS slices of the reduction per workgroup.  These layers are tiny (cfg-A: 256 x 512 -> 64) and bound by
    // the dependent-load chain of one thread, not by bandwidth: many short slices, 8 loads in flight per slice, fixed-order tree.
    __shared__ float red[DENSE_KS][64];
    const int ol = threadIdx.x & 63, kq = threadIdx.x >> 6;
    const int o = blockIdx.x * 64 + ol;
    const int64_t r = blockIdx.y;
    const int per = (n_in + DENSE_KS - 1) / DENSE_KS;
    const int i0 = kq * per, i1 = min(n_in, i0 + per);
    const float* ir = in + r * n_in;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (o < n_out) {
        int i = i0;
        for (; i + 8 <= i1; i += 8) {
            float x[8], y[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x[u] = ir[i + u];
                y[u] = w[(int64_t)(i + u) * n_out + o];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u & 3] = fmaf(x[u], y[u], a[u & 3]);
        }
        for (; i < i1; ++i) a[0] = fmaf(ir[i], w[(int64_t)i * n_out + o], a[0]);
    }
    red[kq][ol] = (a[0] + a[1]) + (a[2] + a[3]);
    __syncthreads();
    if (kq == 0 && o < n_out) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < DENSE_KS; q += 4) t += (red[q][ol] + red[q + 1][ol]) + (red[q + 2][ol] + red[q + 3][ol]);
        out[r * n_out + o] = (b ? b[o] : 0.f) + t;
    }
}

// grad_w[i][o] = sum_r in[r][i]*dout[r][o]; blockIdx.y = i.  The last y-row computes grad_b.
__device__ inline void dense_bwd_w_body(const float* __restrict__ in, const float* __restrict__ dout, int64_t rows, int n_in, int n_out,
                                        float* __restrict__ grad_w, float* __restrict__ grad_b, int bx, int i, float (&red)[DENSE_KS][64]) {
    const int ol = threadIdx.x & 63, rq = threadIdx.x >> 6;
    const int o = bx * 64 + ol;
    const int64_t per = (rows + DENSE_KS - 1) / DENSE_KS;
    const int64_t r0 = rq * per, r1 = (r0 + per < rows) ? r0 + per : rows;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (o < n_out) {
        int64_t r = r0;
        if (i < n_in) {
            for (; r + 8 <= r1; r += 8) {
                float x[8], y[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    x[u] = in[(r + u) * n_in + i];
                    y[u] = dout[(r + u) * n_out + o];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) a[u & 3] = fmaf(x[u], y[u], a[u & 3]);
            }
            for (; r < r1; ++r) a[0] = fmaf(in[r * n_in + i], dout[r * n_out + o], a[0]);
        } else {
            for (; r + 4 <= r1; r += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) a[u] += dout[(r + u) * n_out + o];
            }
            for (; r < r1; ++r) a[0] += dout[r * n_out + o];
        }
    }
    red[rq][ol] = (a[0] + a[1]) + (a[2] + a[3]);
    __syncthreads();
    if (rq == 0 && o < n_out) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < DENSE_KS; q += 4) t += (red[q][ol] + red[q + 1][ol]) + (red[q + 2][ol] + red[q + 3][ol]);
        if (i < n_in) {
            grad_w[(int64_t)i * n_out + o] = t;
        } else {
            grad_b[o] = t;
        }
    }
}

__global__ __launch_bounds__(64 * DENSE_KS) void dense_bwd_w_kernel(const float* __restrict__ in, const float* __restrict__ dout,
                                                                    int64_t rows, int n_in, int n_out, float* __restrict__ grad_w,
                                                                    float* __restrict__ grad_b) {
    __shared__ float red[DENSE_KS][64];
    dense_bwd_w_body(in, dout, rows, n_in, n_out, grad_w, grad_b, blockIdx.x, blockIdx.y, red);
}

// din[r][i] = sum_o dout[r][o]*w[i][o]; blockIdx.y = r, lanes over i (w rows are short: L2-resident).
__device__ inline void dense_bwd_in_body(const float* __restrict__ w, const float* __restrict__ dout, int n_in, int n_out,
                                         float* __restrict__ din, int i, int64_t r) {
    if (i >= n_in) return;
    const float* wr = w + (int64_t)i * n_out;
    const float* dr = dout + r * n_out;
    float a0 = 0.f, a1 = 0.f;
    int o = 0;
    for (; o + 2 <= n_out; o += 2) {
        a0 = fmaf(dr[o], wr[o], a0);
        a1 = fmaf(dr[o + 1], wr[o + 1], a1);
    }
    for (; o < n_out; ++o) a0 = fmaf(dr[o], wr[o], a0);
    din[r * n_in + i] = a0 + a1;
}

__global__ __launch_bounds__(64) void dense_bwd_in_kernel(const float* __restrict__ w, const float* __restrict__ dout,
                                                          int n_in, int n_out, float* __restrict__ din) {
    dense_bwd_in_body(w, dout, n_in, n_out, din, blockIdx.x * 64 + threadIdx.x, blockIdx.y);
}

// ---- siamese head + loss, forward and backward ----------------------------------------------------------
constexpr float KERAS_EPS = 1e-7f;

__device__ inline float dloss_dpred(float p, float y, int loss_kind) {
    if (loss_kind == VM_LOSS_CONTRASTIVE) {
        const float m = fmaxf(1.0f - p, 0.f);
        return 2.f * (1.f - y) * p - 2.f * y * m;
    }
    if (p < KERAS_EPS || p > 1.0f - KERAS_EPS) return 0.f;  // clip_by_value passes no gradient outside
    return (p - y) / (p * (1.0f - p));
}

__device__ inline float loss_value(float p, float y, int loss_kind) {
    if (loss_kind == VM_LOSS_CONTRASTIVE) {
        const float m = fmaxf(1.0f - p, 0.f);
        return (1.f - y) * p * p + y * m * m;
    }
    const float pc = fminf(fmaxf(p, KERAS_EPS), 1.0f - KERAS_EPS);
    const float x = logf(pc / (1.0f - pc));
    return fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
}

__device__ inline float block_sum_256(float v, float* red) {
    const int tid = threadIdx.x;
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// pass 1: one wave per pair (lanes over the embedding dimension): distance, sigmoid, per-pair loss terms, d loss / d emb.
// pair_ws[b] = {loss term, accuracy hit, dL/da, dL/da * distance}.
__global__ __launch_bounds__(256) void siamese_head_pair_kernel(const float* __restrict__ emb, const float* __restrict__ hw,
                                                                const float* __restrict__ hb, const float* __restrict__ y,
                                                                int64_t pairs, int E, int head_kind, int loss_kind, float grad_scale,
                                                                float* __restrict__ pred, float* __restrict__ demb,
                                                                float* __restrict__ pair_ws) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= pairs) return;
    const float* e1 = emb + b * E;
    const float* e2 = emb + (pairs + b) * E;
    float acc = 0.f;
    for (int j = lane; j < E; j += 64) {
        const float df = e1[j] - e2[j];
        acc += head_kind == VM_HEAD_UNIFORM_EUCLIDEAN ? df * df : hw[j] * fabsf(df);
    }
    acc = wave_sum(acc);
    float d = 0.f, a;
    if (head_kind == VM_HEAD_UNIFORM_EUCLIDEAN) {
        d = sqrtf(acc);
        a = fmaf(hw[0], d, hb[0]);
    } else {
        a = acc + hb[0];
    }
    const float p = 1.0f / (1.0f + expf(-a));
    if (lane == 0) pred[b] = p;
    if (y == nullptr) return;
    const float yy = y[b];
    const float dlda = grad_scale * (dloss_dpred(p, yy, loss_kind) * p * (1.0f - p) / (float)pairs);  // x 1.0f is exact
    if (lane == 0) {
        pair_ws[b * 4 + 0] = loss_value(p, yy, loss_kind);
        pair_ws[b * 4 + 1] = (rintf(p) == yy) ? 1.f : 0.f;
        pair_ws[b * 4 + 2] = dlda;
        pair_ws[b * 4 + 3] = dlda * d;
    }
    const float k = head_kind == VM_HEAD_UNIFORM_EUCLIDEAN ? dlda * hw[0] / d : 0.f;  // d == 0 -> inf/NaN like sqrt'(0) in the reference
    for (int j = lane; j < E; j += 64) {
        const float df = e1[j] - e2[j];
        float g;
        if (head_kind == VM_HEAD_UNIFORM_EUCLIDEAN) {
            g = k * df;
        } else {
            g = dlda * hw[j] * (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f));
        }
        demb[b * E + j] = g;
        demb[(pairs + b) * E + j] = -g;
    }
}

// pass 2: fixed-order sums over the pairs: loss, accuracy, head gradients.
__global__ __launch_bounds__(256) void siamese_head_reduce_kernel(const float* __restrict__ emb, const float* __restrict__ pair_ws,
                                                                  int64_t pairs, int E, int head_kind,
                                                                  float* __restrict__ loss_acc, float* __restrict__ grad_hw,
                                                                  float* __restrict__ grad_hb) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    float l = 0.f, h = 0.f, gb = 0.f, gw = 0.f;
    for (int64_t b = tid; b < pairs; b += 256) {
        l += pair_ws[b * 4 + 0];
        h += pair_ws[b * 4 + 1];
        gb += pair_ws[b * 4 + 2];
        gw += pair_ws[b * 4 + 3];
    }
    const float ls = block_sum_256(l, red);
    const float hs = block_sum_256(h, red);
    const float gbs = block_sum_256(gb, red);
    const float gws = block_sum_256(gw, red);
    if (tid == 0) {
        loss_acc[0] = ls / (float)pairs;
        loss_acc[1] = hs / (float)pairs;
        grad_hb[0] = gbs;
        if (head_kind == VM_HEAD_UNIFORM_EUCLIDEAN) grad_hw[0] = gws;
    }
    if (head_kind == VM_HEAD_WEIGHTED_L1) {
        for (int j = tid; j < E; j += 256) {
            float acc = 0.f;
            for (int64_t b = 0; b < pairs; ++b) acc = fmaf(pair_ws[b * 4 + 2], fabsf(emb[b * E + j] - emb[(pairs + b) * E + j]), acc);
            grad_hw[j] = acc;
        }
    }
}

// ---- the whole tail of a siamese training step in two launches (vm_tail_fwd_bwd) ----------------------------------------
// GlobalMaxPool1D (finish of the segment partials) -> Dense(E) -> twin distance -> Dense(1, sigmoid) -> loss -> d loss / d emb
// -> d loss / d gmax is PAIR-LOCAL: nothing of it crosses pairs, so one workgroup runs it for the two windows of a pair out of LDS
// (tail_pair_kernel).  What crosses pairs -- loss, accuracy, the head's and the dense layer's parameter gradients -- is nobody's
// input before the optimizer: tail_reduce_kernel, which the caller may put on another stream.  Six launches of 4-9 us (gmax_segments,
// dense_fwd, siamese_head_pair, siamese_head_reduce, dense_bwd_w, dense_bwd_in) become two, one of them on the critical path.
// Every sum keeps the order of the kernel it replaces (the device functions are shared or restated term by term), so the fused
// path is bit-identical to the six-launch path.
constexpr int TAIL_MAX_C = 1024, TAIL_MAX_E = 256;

__global__ __launch_bounds__(64 * DENSE_KS) void tail_pair_kernel(const float* __restrict__ part_v, const int32_t* __restrict__ part_i,
                                                                  int seg_rows, float* __restrict__ gmax, int32_t* __restrict__ gidx,
                                                                  const float* __restrict__ dw, const float* __restrict__ db,
                                                                  const float* __restrict__ hw, const float* __restrict__ hb,
                                                                  const float* __restrict__ y, int64_t pairs, int C, int E, int head_kind,
                                                                  int loss_kind, float grad_scale, float* __restrict__ emb,
                                                                  float* __restrict__ pred, float* __restrict__ demb,
                                                                  float* __restrict__ dgmax, float* __restrict__ pair_ws) {
    __shared__ float g[2][TAIL_MAX_C];
    __shared__ float ev[2][TAIL_MAX_E];
    __shared__ float de[2][TAIL_MAX_E];
    __shared__ float red[DENSE_KS][64];
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.x;
    const int64_t win[2] = {b, pairs + b};
    // 1. GlobalMaxPool1D: the segment partials of vm_bn_drop_pool_gmax_partials (first maximum wins: gmax_segments_kernel's rule)
    for (int idx = tid; idx < 2 * C; idx += 64 * DENSE_KS) {
        const int w = idx >= C ? 1 : 0, c = idx - w * C;
        const int64_t n = win[w];
        float bv;
        if (part_v != nullptr) {
            bv = -INFINITY;
            int k = 0x7fffffff;
            for (int s = 0; s < seg_rows; ++s) {
                const float yv = part_v[(n * seg_rows + s) * C + c];
                const int kk = part_i[(n * seg_rows + s) * C + c];
                if (kk != 0x7fffffff && (k == 0x7fffffff || yv > bv || (yv == bv && kk < k))) {
                    bv = yv;
                    k = kk;
                }
            }
            gmax[n * C + c] = bv;
            gidx[n * C + c] = k;
        } else {
            bv = gmax[n * C + c];
        }
        g[w][c] = bv;
    }
    __syncthreads();
    // 2. Dense(E), linear (voicemap/models.py:39): dense_fwd_kernel's slices and order, the input row in LDS
    {
        const int ol = tid & 63, kq = tid >> 6;
        const int per = (C + DENSE_KS - 1) / DENSE_KS;
        const int i0 = kq * per, i1 = min(C, i0 + per);
        for (int w = 0; w < 2; ++w) {
            for (int ob = 0; ob * 64 < E; ++ob) {
                const int o = ob * 64 + ol;
                const float* ir = g[w];
                float a[4] = {0.f, 0.f, 0.f, 0.f};
                if (o < E) {
                    int i = i0;
                    for (; i + 8 <= i1; i += 8) {
                        float x[8], yv[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            x[u] = ir[i + u];
                            yv[u] = dw[(int64_t)(i + u) * E + o];
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) a[u & 3] = fmaf(x[u], yv[u], a[u & 3]);
                    }
                    for (; i < i1; ++i) a[0] = fmaf(ir[i], dw[(int64_t)i * E + o], a[0]);
                }
                red[kq][ol] = (a[0] + a[1]) + (a[2] + a[3]);
                __syncthreads();
                if (kq == 0 && o < E) {
                    float t = 0.f;
#pragma unroll
                    for (int q = 0; q < DENSE_KS; q += 4) t += (red[q][ol] + red[q + 1][ol]) + (red[q + 2][ol] + red[q + 3][ol]);
                    const float v = (db ? db[o] : 0.f) + t;
                    emb[win[w] * E + o] = v;
                    ev[w][o] = v;
                }
                __syncthreads();
            }
        }
    }
    // 3. distance -> Dense(1, sigmoid) -> loss terms -> d loss / d emb: siamese_head_pair_kernel's wave, on the LDS copy
    if (tid < 64) {
        const int lane = tid;
        const float* e1 = ev[0];
        const float* e2 = ev[1];
        float acc = 0.f;
        for (int j = lane; j < E; j += 64) {
            const float df = e1[j] - e2[j];
            acc += head_kind == VM_HEAD_UNIFORM_EUCLIDEAN ? df * df : hw[j] * fabsf(df);
        }
        acc = wave_sum(acc);
        float d = 0.f, a;
        if (head_kind == VM_HEAD_UNIFORM_EUCLIDEAN) {
            d = sqrtf(acc);
            a = fmaf(hw[0], d, hb[0]);
        } else {
            a = acc + hb[0];
        }
        const float p = 1.0f / (1.0f + expf(-a));
        if (lane == 0) pred[b] = p;
        const float yy = y[b];
        const float dlda = grad_scale * (dloss_dpred(p, yy, loss_kind) * p * (1.0f - p) / (float)pairs);
        if (lane == 0) {
            pair_ws[b * 4 + 0] = loss_value(p, yy, loss_kind);
            pair_ws[b * 4 + 1] = (rintf(p) == yy) ? 1.f : 0.f;
            pair_ws[b * 4 + 2] = dlda;
            pair_ws[b * 4 + 3] = dlda * d;
        }
        const float k = head_kind == VM_HEAD_UNIFORM_EUCLIDEAN ? dlda * hw[0] / d : 0.f;
        for (int j = lane; j < E; j += 64) {
            const float df = e1[j] - e2[j];
            float gv;
            if (head_kind == VM_HEAD_UNIFORM_EUCLIDEAN) {
                gv = k * df;
            } else {
                gv = dlda * hw[j] * (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f));
            }
            demb[b * E + j] = gv;
            demb[(pairs + b) * E + j] = -gv;
            de[0][j] = gv;
            de[1][j] = -gv;
        }
    }
    __syncthreads();
    // 4. d loss / d gmax = demb W^T (dense_bwd_in_body's two-accumulator order)
    for (int idx = tid; idx < 2 * C; idx += 64 * DENSE_KS) {
        const int w = idx >= C ? 1 : 0, i = idx - w * C;
        const float* wr = dw + (int64_t)i * E;
        const float* dr = de[w];
        float a0 = 0.f, a1 = 0.f;
        int o = 0;
        for (; o + 2 <= E; o += 2) {
            a0 = fmaf(dr[o], wr[o], a0);
            a1 = fmaf(dr[o + 1], wr[o + 1], a1);
        }
        for (; o < E; ++o) a0 = fmaf(dr[o], wr[o], a0);
        dgmax[win[w] * C + i] = a0 + a1;
    }
}

// blockIdx.y <= n_in: dense_bwd_w_kernel (row n_in: the bias gradient); blockIdx.y == n_in + 1 (blockIdx.x == 0): the fixed-order
// sums of siamese_head_reduce_kernel, taken by the first four waves
__global__ __launch_bounds__(64 * DENSE_KS) void tail_reduce_kernel(const float* __restrict__ gmax, const float* __restrict__ demb,
                                                                    const float* __restrict__ emb, const float* __restrict__ pair_ws,
                                                                    int64_t pairs, int C, int E, int head_kind, float* __restrict__ loss_acc,
                                                                    float* __restrict__ grad_dw, float* __restrict__ grad_db,
                                                                    float* __restrict__ grad_hw, float* __restrict__ grad_hb) {
    __shared__ float red[DENSE_KS][64];
    if ((int)blockIdx.y <= C) {
        dense_bwd_w_body(gmax, demb, 2 * pairs, C, E, grad_dw, grad_db, blockIdx.x, blockIdx.y, red);
        return;
    }
    if (blockIdx.x != 0) return;
    const int tid = threadIdx.x;
    float l = 0.f, h = 0.f, gb = 0.f, gw = 0.f;
    if (tid < 256) {
        for (int64_t b = tid; b < pairs; b += 256) {
            l += pair_ws[b * 4 + 0];
            h += pair_ws[b * 4 + 1];
            gb += pair_ws[b * 4 + 2];
            gw += pair_ws[b * 4 + 3];
        }
    }
    l = wave_sum(l);
    h = wave_sum(h);
    gb = wave_sum(gb);
    gw = wave_sum(gw);
    if (tid < 256 && (tid & 63) == 0) {
        red[0][tid >> 6] = l;
        red[1][tid >> 6] = h;
        red[2][tid >> 6] = gb;
        red[3][tid >> 6] = gw;
    }
    __syncthreads();
    if (tid == 0) {
        loss_acc[0] = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (float)pairs;
        loss_acc[1] = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / (float)pairs;
        grad_hb[0] = red[2][0] + red[2][1] + red[2][2] + red[2][3];
        if (head_kind == VM_HEAD_UNIFORM_EUCLIDEAN) grad_hw[0] = red[3][0] + red[3][1] + red[3][2] + red[3][3];
    }
    if (head_kind == VM_HEAD_WEIGHTED_L1 && tid < 256) {
        for (int j = tid; j < E; j += 256) {
            float acc = 0.f;
            for (int64_t b = 0; b < pairs; ++b) acc = fmaf(pair_ws[b * 4 + 2], fabsf(emb[b * E + j] - emb[(pairs + b) * E + j]), acc);
            grad_hw[j] = acc;
        }
    }
}

// ---- softmax + categorical cross-entropy ----------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_cce_kernel(const float* __restrict__ logits, const int32_t* __restrict__ labels,
                                                          int64_t rows, int n_classes, float* __restrict__ prob,
                                                          float* __restrict__ row_loss, float* __restrict__ row_hit,
                                                          float* __restrict__ dlogits, float grad_scale) {
    __shared__ float red[4];
    __shared__ float redm[4];
    __shared__ int redi[4];
    const int tid = threadIdx.x;
    const int64_t r = blockIdx.x;
    const float* x = logits + r * n_classes;
    float m = -INFINITY;
    int mi = 0x7fffffff;
    for (int c = tid; c < n_classes; c += 256) {
        if (x[c] > m) {
            m = x[c];
            mi = c;
        }
    }
    // block argmax (first maximum)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o, 64);
        const int oi = __shfl_xor(mi, o, 64);
        if (om > m || (om == m && oi < mi)) {
            m = om;
            mi = oi;
        }
    }
    if ((tid & 63) == 0) {
        redm[tid >> 6] = m;
        redi[tid >> 6] = mi;
    }
    __syncthreads();
    m = redm[0];
    mi = redi[0];
    for (int i = 1; i < 4; ++i)
        if (redm[i] > m || (redm[i] == m && redi[i] < mi)) {
            m = redm[i];
            mi = redi[i];
        }
    float s = 0.f;
    for (int c = tid; c < n_classes; c += 256) s += expf(x[c] - m);
    const float S = block_sum_256(s, red);
    float psum = 0.f;
    for (int c = tid; c < n_classes; c += 256) {
        const float p = expf(x[c] - m) / S;
        prob[r * n_classes + c] = p;
        psum += p;
    }
    const float PS = block_sum_256(psum, red);  // Keras renormalises the softmax output before the clip
    if (labels == nullptr) return;
    const int lab = labels[r];
    const float ql_raw = (expf(x[lab] - m) / S) / PS;
    const bool in_range = ql_raw >= KERAS_EPS && ql_raw <= 1.0f - KERAS_EPS;
    if (tid == 0) {
        const float ql = fminf(fmaxf(ql_raw, KERAS_EPS), 1.0f - KERAS_EPS);
        row_loss[r] = -logf(ql);
        row_hit[r] = (mi == lab) ? 1.f : 0.f;
    }
    if (dlogits != nullptr) {
        for (int c = tid; c < n_classes; c += 256) {
            const float q = (expf(x[c] - m) / S) / PS;
            float g = in_range ? (q - (c == lab ? 1.f : 0.f)) : 0.f;
            dlogits[r * n_classes + c] = grad_scale * (g / (float)rows);
        }
    }
}

__global__ __launch_bounds__(256) void mean2_kernel(const float* a, const float* b, int64_t n, float* out) {
    __shared__ float red[4];
    float sa = 0.f, sb = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        sa += a[i];
        sb += b[i];
    }
    const float A = block_sum_256(sa, red);
    const float B = block_sum_256(sb, red);
    if (threadIdx.x == 0) {
        out[0] = A / (float)n;
        out[1] = B / (float)n;
    }
}

static int lanes_for_t(int cv) {
    int p = 1;
    while (p < cv && p < 256) p <<= 1;
    return p;
}

}  // namespace vm

using namespace vm;

extern "C" int vm_global_maxpool_fwd(const void* act, int64_t n_windows, int64_t L, int C, int dtype, float* gmax,
                                     int32_t* gidx, void* stream) {
    VM_REQUIRE(act && gmax && gidx, "vm_global_maxpool_fwd: null pointer");
    VM_REQUIRE(n_windows > 0 && L > 0 && C % 8 == 0, "vm_global_maxpool_fwd: bad sizes");
    VM_DISPATCH_DTYPE(dtype, {
        const int P = lanes_for_t(C / Elem<T>::kVec);
        hipLaunchKernelGGL((global_maxpool_fwd_kernel<T>), dim3((unsigned)n_windows), dim3(256), 0, (hipStream_t)stream,
                           (const T*)act, L, C, P, gmax, gidx);
    });
    return check_launch("vm_global_maxpool_fwd");
}

extern "C" int vm_global_maxpool_bwd(const float* dg, const int32_t* gidx, int64_t n_windows, int64_t L, int C, int dtype,
                                     void* dp, void* stream) {
    VM_REQUIRE(dg && gidx && dp, "vm_global_maxpool_bwd: null pointer");
    VM_REQUIRE(n_windows > 0 && L > 0 && C % 8 == 0, "vm_global_maxpool_bwd: bad sizes");
    VM_DISPATCH_DTYPE(dtype, {
        const int64_t total = n_windows * L * (C / Elem<T>::kVec);
        hipLaunchKernelGGL((global_maxpool_bwd_kernel<T>), dim3((unsigned)cdiv(total, 256)), dim3(256), 0,
                           (hipStream_t)stream, dg, gidx, total, L, C, (T*)dp);
    });
    return check_launch("vm_global_maxpool_bwd");
}

extern "C" int vm_dense_fwd(const float* in, const float* w, const float* b, int64_t rows, int n_in, int n_out, float* out,
                            void* stream) {
    VM_REQUIRE(in && w && out && rows > 0 && n_in > 0 && n_out > 0, "vm_dense_fwd: bad argument");
    for (int64_t r0 = 0; r0 < rows; r0 += 65535) {  // grid.y is limited to 65535
        const int64_t nr = rows - r0 < 65535 ? rows - r0 : 65535;
        hipLaunchKernelGGL(dense_fwd_kernel, dim3((n_out + 63) / 64, (unsigned)nr), dim3(64 * DENSE_KS), 0, (hipStream_t)stream,
                           in + r0 * n_in, w, b, n_in, n_out, out + r0 * n_out);
    }
    return check_launch("vm_dense_fwd");
}

extern "C" int vm_dense_bwd(const float* in, const float* w, const float* dout, int64_t rows, int n_in, int n_out,
                            float* grad_w, float* grad_b, float* din, void* stream) {
    VM_REQUIRE(in && w && dout && rows > 0 && n_in > 0 && n_out > 0, "vm_dense_bwd: bad argument");
    VM_REQUIRE((grad_w == nullptr) == (grad_b == nullptr) && (grad_w != nullptr || din != nullptr), "vm_dense_bwd: grad_w / grad_b go together; one of (grad_w, din) is needed");
    VM_REQUIRE(n_in < 65535, "vm_dense_bwd: n_in too large");
    int rc = VM_OK;
    if (grad_w != nullptr) {   // NULL: the input gradient only (the caller puts the parameter half on another stream)
        hipLaunchKernelGGL(dense_bwd_w_kernel, dim3((n_out + 63) / 64, n_in + 1), dim3(64 * DENSE_KS), 0, (hipStream_t)stream, in, dout,
                           rows, n_in, n_out, grad_w, grad_b);
        rc = check_launch("vm_dense_bwd(w)");
    }
    if (rc || din == nullptr) return rc;
    for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
        const int64_t nr = rows - r0 < 65535 ? rows - r0 : 65535;
        hipLaunchKernelGGL(dense_bwd_in_kernel, dim3((n_in + 63) / 64, (unsigned)nr), dim3(64), 0, (hipStream_t)stream, w,
                           dout + r0 * n_out, n_in, n_out, din + r0 * n_in);
    }
    return check_launch("vm_dense_bwd(in)");
}

extern "C" int vm_siamese_head_loss(const float* emb, const float* head_w, const float* head_b, const float* y, int64_t pairs,
                                    int E, int head_kind, int loss_kind, float grad_scale, float* pred, float* loss_acc,
                                    float* demb, float* grad_hw, float* grad_hb, float* ws, void* stream) {
    VM_REQUIRE(emb && head_w && head_b && pred, "vm_siamese_head_loss: null pointer");
    VM_REQUIRE(pairs > 0 && E > 0, "vm_siamese_head_loss: bad sizes");
    VM_REQUIRE(head_kind == VM_HEAD_UNIFORM_EUCLIDEAN || head_kind == VM_HEAD_WEIGHTED_L1,
               "vm_siamese_head_loss: head_kind %d not implemented (the reference raises NotImplementedError too)", head_kind);
    VM_REQUIRE(loss_kind == VM_LOSS_CONTRASTIVE || loss_kind == VM_LOSS_BCE, "vm_siamese_head_loss: unknown loss %d", loss_kind);
    VM_REQUIRE(y == nullptr || (demb && ws && (loss_acc == nullptr || (grad_hw && grad_hb))), "vm_siamese_head_loss: training outputs missing");
    hipLaunchKernelGGL(siamese_head_pair_kernel, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, (hipStream_t)stream, emb, head_w,
                       head_b, y, pairs, E, head_kind, loss_kind, grad_scale, pred, demb, ws);
    int rc = check_launch("vm_siamese_head_loss");
    if (rc || y == nullptr || loss_acc == nullptr) return rc;   // loss_acc NULL: the caller runs vm_siamese_head_reduce itself
    hipLaunchKernelGGL(siamese_head_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, emb, (const float*)ws, pairs, E,
                       head_kind, loss_acc, grad_hw, grad_hb);
    return check_launch("vm_siamese_head_loss(reduce)");
}

extern "C" int vm_siamese_head_reduce(const float* emb, const float* ws, int64_t pairs, int E, int head_kind, float* loss_acc,
                                      float* grad_hw, float* grad_hb, void* stream) {
    VM_REQUIRE(emb && ws && loss_acc && grad_hw && grad_hb && pairs > 0 && E > 0, "vm_siamese_head_reduce: bad argument");
    VM_REQUIRE(head_kind == VM_HEAD_UNIFORM_EUCLIDEAN || head_kind == VM_HEAD_WEIGHTED_L1, "vm_siamese_head_reduce: head_kind %d", head_kind);
    hipLaunchKernelGGL(siamese_head_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, emb, ws, pairs, E, head_kind, loss_acc,
                       grad_hw, grad_hb);
    return check_launch("vm_siamese_head_reduce");
}

extern "C" int vm_softmax_cce(const float* logits, const int32_t* labels, int64_t rows, int n_classes, float grad_scale, float* prob,
                              float* loss_acc, float* dlogits, float* ws, void* stream) {
    VM_REQUIRE(logits && prob && rows > 0 && n_classes > 0, "vm_softmax_cce: bad argument");
    VM_REQUIRE(labels == nullptr || (loss_acc && ws), "vm_softmax_cce: loss_acc and ws (2*rows floats) required with labels");
    hipLaunchKernelGGL(softmax_cce_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, logits, labels, rows,
                       n_classes, prob, ws, ws ? ws + rows : nullptr, dlogits, grad_scale);
    int rc = check_launch("vm_softmax_cce");
    if (rc || labels == nullptr) return rc;
    hipLaunchKernelGGL(mean2_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)ws, (const float*)(ws + rows),
                       rows, loss_acc);
    return check_launch("vm_softmax_cce(mean)");
}

extern "C" int vm_tail_fwd_bwd_supported(int C, int E) { return C > 0 && E > 0 && C <= TAIL_MAX_C && E <= TAIL_MAX_E; }

extern "C" int vm_tail_fwd_bwd(const float* gmax_part_v, const int32_t* gmax_part_i, int seg_rows, float* gmax, int32_t* gidx, const float* dense_w, const float* dense_b,
                               const float* head_w, const float* head_b, const float* y, int64_t pairs, int C, int E, int head_kind,
                               int loss_kind, float grad_scale, float* emb, float* pred, float* demb, float* dgmax, float* ws, void* stream) {
    VM_REQUIRE(gmax && gidx && dense_w && head_w && head_b && y && emb && pred && demb && dgmax && ws, "vm_tail_fwd_bwd: null pointer");
    VM_REQUIRE(pairs > 0 && vm_tail_fwd_bwd_supported(C, E), "vm_tail_fwd_bwd: needs pairs > 0, 0 < C <= %d, 0 < E <= %d (got %d, %d)",
               TAIL_MAX_C, TAIL_MAX_E, C, E);
    VM_REQUIRE((gmax_part_v == nullptr) == (gmax_part_i == nullptr) && (gmax_part_v == nullptr || seg_rows > 0),
               "vm_tail_fwd_bwd: gmax_part_v / gmax_part_i go together, with seg_rows > 0");
    VM_REQUIRE(head_kind == VM_HEAD_UNIFORM_EUCLIDEAN || head_kind == VM_HEAD_WEIGHTED_L1,
               "vm_tail_fwd_bwd: head_kind %d not implemented (the reference raises NotImplementedError too)", head_kind);
    VM_REQUIRE(loss_kind == VM_LOSS_CONTRASTIVE || loss_kind == VM_LOSS_BCE, "vm_tail_fwd_bwd: unknown loss %d", loss_kind);
    const float* part_v = gmax_part_v;
    const int32_t* part_i = gmax_part_i;
    hipLaunchKernelGGL(tail_pair_kernel, dim3((unsigned)pairs), dim3(64 * DENSE_KS), 0, (hipStream_t)stream, part_v, part_i, seg_rows, gmax,
                       gidx, dense_w, dense_b, head_w, head_b, y, pairs, C, E, head_kind, loss_kind, grad_scale, emb, pred, demb, dgmax, ws);
    return check_launch("vm_tail_fwd_bwd");
}

extern "C" int vm_tail_param_grads(const float* gmax, const float* demb, const float* emb, const float* ws, int64_t pairs, int C, int E,
                                   int head_kind, float* loss_acc, float* grad_dense_w, float* grad_dense_b, float* grad_hw, float* grad_hb,
                                   void* stream) {
    VM_REQUIRE(gmax && demb && emb && ws && loss_acc && grad_dense_w && grad_dense_b && grad_hw && grad_hb, "vm_tail_param_grads: null pointer");
    VM_REQUIRE(pairs > 0 && C > 0 && E > 0 && C < 65534, "vm_tail_param_grads: bad sizes");
    VM_REQUIRE(head_kind == VM_HEAD_UNIFORM_EUCLIDEAN || head_kind == VM_HEAD_WEIGHTED_L1, "vm_tail_param_grads: head_kind %d", head_kind);
    hipLaunchKernelGGL(tail_reduce_kernel, dim3((E + 63) / 64, C + 2), dim3(64 * DENSE_KS), 0, (hipStream_t)stream, gmax, demb, emb, ws, pairs,
                       C, E, head_kind, loss_acc, grad_dense_w, grad_dense_b, grad_hw, grad_hb);
    return check_launch("vm_tail_param_grads");
}
