// Channel ("transposed") attention, a5: the c x c attention map of each head is built from
// L2-normalised q and k rows, so the expensive part is two reductions over all N pixels:
//   G[i][j] = sum_n q_i[n] k_j[n],   |q_i|^2,   |k_j|^2.
// gram_kernel streams q and k once (HBM-bound, c/4 flop per byte) and forms 16x16 Gram tiles with
// v_mfma_f32_16x16x4_f32, the pixel axis being the MFMA K dimension.  The row norms are the
// diagonals of q q^T and k k^T and reuse the SAME registers as A and B operand (for this
// instruction A[i][k] and B[k][j] have the same lane map), so they cost one extra MFMA each.
// Pixel slabs are reduced through per-slab partials (no float atomics: results are bitwise
// reproducible run to run).
// attn_fold_kernel finishes the softmax and folds it into project_out:
//   project_out(attn @ v) = (W_out * blockdiag(attn_b)) v,
// a per-image C x C matrix written directly in MFMA operand order, so attn @ v and the 1x1
// projection become ONE GEMM over v (rf_conv1x1.hip) instead of two more passes over N.
#include "rf_common.h"

namespace rf {

static constexpr int kMaxBand = 4;            // k tiles a q tile can need (head size <= 64)
static constexpr int kRowW = kMaxBand * 16 + 2;   // partial row: band columns, |q|^2, |k|^2

__host__ __device__ static inline void band_of(int tq, int C, int c, int* tklo, int* nb) {
    const int lo_ch = 16 * tq;
    const int hi_ch = (16 * tq + 15 < C - 1) ? 16 * tq + 15 : C - 1;
    const int hlo = lo_ch / c, hhi = hi_ch / c;
    const int klo = hlo * c, khi = (hhi + 1) * c - 1;
    *tklo = klo / 16;
    *nb = khi / 16 - *tklo + 1;
}

int gram_plan(int B, int C, int heads, int P, int* nslab, int* slab, size_t* partial_floats) {
    RF_CHECK_ARG(heads > 0 && C % heads == 0, "chan_attn: C=%d not divisible by heads=%d", C, heads);
    const int c = C / heads;
    RF_CHECK_ARG(c <= 64, "chan_attn: head size %d > 64 not supported", c);
    const int NTq = cdiv(C, 16);
    // a 16-channel q tile may straddle heads when the head size is not 16-aligned; the k tiles it then needs must fit
    // the partial row (kMaxBand tiles): c = 40 or 56 would need 5 / 7 and are rejected instead of computed wrongly
    for (int tq = 0; tq < NTq; ++tq) {
        int tklo, nb;
        band_of(tq, C, c, &tklo, &nb);
        RF_CHECK_ARG(nb <= kMaxBand, "chan_attn: C=%d with %d heads (head size %d) needs %d key tiles per query tile, at most %d supported",
                     C, heads, c, nb, kMaxBand);
    }
    // the slab split depends only on (C, P), never on B: an image's result is bitwise the same
    // alone and inside a batch
    int target = P / 16;                       // pixels per slab: at most 16 slabs per image (a slab's fixed cost -- ramp,
                                               // LDS reduction, 1056 stores -- is worth about 600 pixels),
    if (target < 1024) target = 1024;          // at least 1024 px so small levels still fill the GPU
    if (target > 8192) target = 8192;
    int ns = cdiv(P, target);
    const int maxs = cdiv(P, 256);
    if (ns > maxs) ns = maxs;
    if (ns < 1) ns = 1;
    int sl = cdiv(cdiv(P, ns), 256) * 256;
    ns = cdiv(P, sl);
    *nslab = ns;
    *slab = sl;
    *partial_floats = (size_t)B * ns * NTq * 16 * kRowW;
    return RF_OK;
}

__global__ void __launch_bounds__(256) gram_kernel(GramArgs a, int vec) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int slab_id = blockIdx.x, tq = blockIdx.y, b = blockIdx.z;
    const int C = a.C, P = a.P, c = C / a.heads;
    int tklo, nb;
    band_of(tq, C, c, &tklo, &nb);
    const float* qb = a.q + (size_t)b * a.bstride;
    const float* kb = a.k + (size_t)b * a.bstride;
    // the slab's pixels, cut to the rows that enter the statistics (a spatial shard counts its interior rows only)
    const int m_hi = a.p_hi > 0 ? a.p_hi : P;
    int n_lo = slab_id * a.slab;
    int n_hi = (n_lo + a.slab < m_hi) ? n_lo + a.slab : m_hi;
    if (n_lo < a.p_lo) n_lo = a.p_lo;
    if (n_hi < n_lo) n_hi = n_lo;

    // |q_i|^2, |k_i|^2: per-lane sums of squares on the VALU (as diagonals of two more MFMA products they cost 8 of the 12 MFMAs of a
    // step at one band tile; MFMA and VALU time add up on gfx950)
    f32x4 g[kMaxBand];
    float nq = 0.f, nk = 0.f;
#pragma unroll
    for (int t = 0; t < kMaxBand; ++t) g[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto load4 = [&](const float* base, int ch, int n) -> float4 {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ch < C) {
            const float* p = base + (size_t)ch * P + n;
            if (vec) {
                if (n < n_hi) v = *reinterpret_cast<const float4*>(p);
            } else {
                if (n < n_hi) v.x = p[0];
                if (n + 1 < n_hi) v.y = p[1];
                if (n + 2 < n_hi) v.z = p[2];
                if (n + 3 < n_hi) v.w = p[3];
            }
        }
        return v;
    };

    // a wave step covers 16 pixels: lane (i, kq) holds pixels n + 4kq .. 4kq+3 of channel row i.
    // The k rows of the diagonal tile (for |k|^2) are the band tile tq - tklo: no separate load.
    const int td = tq - tklo;
    auto mfma_step = [&](const float4& qv, const float4 (&kv)[kMaxBand]) __attribute__((always_inline)) {
        const float qa[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
        for (int m = 0; m < 4; ++m) nq = fmaf(qa[m], qa[m], nq);
#pragma unroll
        for (int t = 0; t < kMaxBand; ++t) {
            if (t < nb) {
                const float kt[4] = {kv[t].x, kv[t].y, kv[t].z, kv[t].w};
#pragma unroll
                for (int m = 0; m < 4; ++m) g[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[m], kt[m], g[t], 0, 0, 0);
                if (t == td) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) nk = fmaf(kt[m], kt[m], nk);
                }
            }
        }
    };
    const bool fast = vec && 16 * tq + 16 <= C && 16 * (tklo + nb) <= C && ((n_hi - n_lo) % 64 == 0);
    if (fast) {
        // every load in range: branch-free, and the next step's loads are issued before this step's MFMAs
        const float* qp = qb + (size_t)(16 * tq + i) * P + 4 * kq;
        const float* kp = kb + (size_t)(16 * tklo + i) * P + 4 * kq;
        auto load_step = [&](int n, float4& qv, float4 (&kv)[kMaxBand]) __attribute__((always_inline)) {
            qv = *reinterpret_cast<const float4*>(qp + n);
#pragma unroll
            for (int t = 0; t < kMaxBand; ++t)
                kv[t] = *reinterpret_cast<const float4*>(kp + (size_t)(t < nb ? t : 0) * 16 * P + n);
        };
        float4 q0, q1, k0[kMaxBand], k1[kMaxBand];
        int n = n_lo + wave * 16;
        load_step(n, q0, k0);
        for (; n < n_hi; n += 128) {
            const bool more1 = n + 64 < n_hi;
            load_step(more1 ? n + 64 : n, q1, k1);
            mfma_step(q0, k0);
            if (!more1) break;
            load_step(n + 128 < n_hi ? n + 128 : n, q0, k0);
            mfma_step(q1, k1);
        }
    } else {
        for (int n = n_lo + wave * 16; n < n_hi; n += 64) {
            const int nn = n + 4 * kq;
            const float4 qv = load4(qb, 16 * tq + i, nn);
            float4 kv[kMaxBand];
#pragma unroll
            for (int t = 0; t < kMaxBand; ++t)
                kv[t] = (t < nb) ? load4(kb, 16 * (tklo + t) + i, nn) : make_float4(0.f, 0.f, 0.f, 0.f);
            mfma_step(qv, kv);
        }
    }

    // cross-wave reduction in a fixed order, then one plain store per value
    __shared__ float red[4][16][kRowW];
    nq += __shfl_xor(nq, 16); nq += __shfl_xor(nq, 32);       // channel i: the four kq lanes hold disjoint pixels
    nk += __shfl_xor(nk, 16); nk += __shfl_xor(nk, 32);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * kq + r;
#pragma unroll
        for (int t = 0; t < kMaxBand; ++t) red[wave][row][t * 16 + i] = g[t][r];
        if (row == i) {
            red[wave][row][kMaxBand * 16] = nq;
            red[wave][row][kMaxBand * 16 + 1] = nk;
        }
    }
    __syncthreads();
    float* dst = a.partial + (((size_t)b * a.nslab + slab_id) * gridDim.y + tq) * 16 * kRowW;
    for (int idx = threadIdx.x; idx < 16 * kRowW; idx += 256) {
        const int row = idx / kRowW, col = idx % kRowW;
        dst[idx] = ((red[0][row][col] + red[1][row][col]) + red[2][row][col]) + red[3][row][col];
    }
}

int launch_gram(const GramArgs& a, hipStream_t st) {
    RF_CHECK_ARG(a.B > 0 && a.B <= 65535 && a.C > 0 && a.P > 0, "gram: bad sizes");
    const int vec = (a.P % 4 == 0) && aligned16(a.q) && aligned16(a.k) && (a.bstride % 4 == 0);
    dim3 grid((unsigned)a.nslab, (unsigned)cdiv(a.C, 16), (unsigned)a.B);
    const double c = (double)a.C / a.heads;
    ProfScope prof(st, "gram_kernel", 2.0 * a.C * c * a.P * a.B, 8.0 * a.C * (double)a.P * a.B);
    gram_kernel<<<grid, 256, 0, st>>>(a, vec);
    return check_launch("gram");
}

// one workgroup per (head, image)
__global__ void __launch_bounds__(256) attn_fold_kernel(const float* __restrict__ partial, int nslab,
                                                        const float* __restrict__ temperature,
                                                        const float* __restrict__ w_out, float* __restrict__ wp_out,
                                                        unsigned short* __restrict__ wp3_out, int C, int heads, int log_temperature) {
    const int hd = blockIdx.x, b = blockIdx.y;
    const int c = C / heads;
    const int NT = (C + 15) >> 4;
    __shared__ float S[64][65];
    __shared__ float nq[64], nk[64];
    const float* pb = partial + (size_t)b * nslab * NT * 16 * kRowW;
    // 1. reduce the slab partials: value v is summed by RS threads over interleaved slabs, then the RS
    //    sub-sums are combined in a fixed order (deterministic, and parallel when c*c + 2c < 256)
    __shared__ float sub[256];
    const int nval = c * c + 2 * c;
    int RS = 256 / nval;
    if (RS < 1) RS = 1;
    if (RS > 16) RS = 16;
    for (int base = 0; base < nval; base += 256 / RS) {
        const int vi = base + threadIdx.x / RS, rs = threadIdx.x % RS;
        float s = 0.f;
        if (vi < nval && threadIdx.x / RS < 256 / RS) {
            int qch, col;
            if (vi < c * c) {
                const int ii = vi / c, jj = vi % c;
                qch = hd * c + ii;
                const int kch = hd * c + jj;
                int tklo, nb;
                band_of(qch >> 4, C, c, &tklo, &nb);
                col = ((kch >> 4) - tklo) * 16 + (kch & 15);
            } else if (vi < c * c + c) {
                qch = hd * c + (vi - c * c);
                col = kMaxBand * 16;
            } else {
                qch = hd * c + (vi - c * c - c);
                col = kMaxBand * 16 + 1;
            }
            const float* src = pb + ((size_t)(qch >> 4) * 16 + (qch & 15)) * kRowW + col;
            // loads of 8 slabs in flight, summed in slab order (same result, one exposed round trip per 8 instead of per slab)
            int sl = rs;
            for (; sl + 7 * RS < nslab; sl += 8 * RS) {
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = src[(size_t)(sl + u * RS) * NT * 16 * kRowW];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += t[u];
            }
            for (; sl < nslab; sl += RS) s += src[(size_t)sl * NT * 16 * kRowW];
        }
        sub[threadIdx.x] = s;
        __syncthreads();
        if (rs == 0 && vi < nval && threadIdx.x / RS < 256 / RS) {
            float t = 0.f;
            for (int q = 0; q < RS; ++q) t += sub[threadIdx.x + q];
            if (vi < c * c) S[vi / c][vi % c] = t;
            else if (vi < c * c + c) nq[vi - c * c] = t;
            else nk[vi - c * c - c] = t;
        }
        __syncthreads();
    }
    // 2. cosine similarity * temperature, softmax over j (F.normalize clamps the norm at 1e-12)
    if (threadIdx.x < c) {
        const int ii = threadIdx.x;
        const float T = log_temperature ? expf(temperature[hd]) : temperature[hd];
        const float rq = 1.0f / fmaxf(sqrtf(nq[ii]), 1e-12f);
        float m = -INFINITY;
        for (int jj = 0; jj < c; ++jj) {
            const float v = S[ii][jj] * rq * (1.0f / fmaxf(sqrtf(nk[jj]), 1e-12f)) * T;
            S[ii][jj] = v;
            m = fmaxf(m, v);
        }
        float sum = 0.f;
        for (int jj = 0; jj < c; ++jj) {
            const float e = expf(S[ii][jj] - m);
            S[ii][jj] = e;
            sum += e;
        }
        const float inv = 1.0f / sum;
        for (int jj = 0; jj < c; ++jj) S[ii][jj] *= inv;
    }
    __syncthreads();
    // 3. W'[co][hd*c + j] = sum_i W_out[co][hd*c + i] * attn[i][j], stored in MFMA operand order
    float* dst = wp_out + (size_t)b * NT * (C / 4) * 64;
    const int NB = (C + 31) >> 5;
    unsigned short* dst3 = wp3_out ? wp3_out + (size_t)b * NT * NB * 1536 : nullptr;   // b3 form of the same matrix (rf_common.h)
    for (int idx = threadIdx.x; idx < NT * 16 * c; idx += 256) {
        const int co = idx / c, jj = idx % c;
        float s = 0.f;
        if (co < C) {
            const float* wr = w_out + (size_t)co * C + hd * c;
            for (int ii = 0; ii < c; ++ii) s = fmaf(wr[ii], S[ii][jj], s);
        }
        const int k = hd * c + jj;
        dst[((size_t)(k >> 2) * NT + (co >> 4)) * 64 + (co & 15) + 16 * (k & 3)] = s;
        if (dst3) b3_store(dst3, NT, co, k, s);
    }
    // columns C .. 32 NB - 1 of the padded b3 matrix (C not a multiple of 32): zero, written by the last head's workgroup
    if (dst3 && hd == heads - 1)
        for (int idx = threadIdx.x; idx < NT * 16 * (32 * NB - C); idx += 256)
            b3_store(dst3, NT, idx / (32 * NB - C), C + idx % (32 * NB - C), 0.f);
}

int launch_attn_fold(const float* partial, int nslab, const float* temperature, const float* w_out,
                     float* wp_out, void* wp3_out, int B, int C, int heads, hipStream_t st, int log_temperature) {
    RF_CHECK_ARG(C % heads == 0 && C / heads <= 64 && C % 4 == 0, "attn_fold: unsupported C=%d heads=%d", C, heads);
    ProfScope prof(st, "attn_fold_kernel", 0.0, 0.0);
    attn_fold_kernel<<<dim3((unsigned)heads, (unsigned)B), 256, 0, st>>>(partial, nslab, temperature, w_out, wp_out, (unsigned short*)wp3_out, C, heads, log_temperature);
    return check_launch("attn_fold");
}

}  // namespace rf
