// Luminance-aware token attention (Attenblock.py:143-220).
//
//   q,k,v = split(to_qkv(x))                        [B, heads*d, N]   (1x1 GEMM, rf_gemm1x1.hip)
//   t' = gamma * t + beta  for t in q,k,v;  q' += alpha * (avgpool3(1 - luma) - mean)     luma_film_kernel
//   out[:, i] = sum_j softmax_j(q'_i . k'_j * d^-1/2) v'_j   per (image, head)               token_attn_kernel
//
// token_attn_kernel is a flash-style kernel: the N x N score matrix (8.6 GB for one 128 x 128 stage-0
// image of config 1) never exists.  A wave owns 16 queries and streams the keys in blocks of 64:
//   S^T tile  (16 keys x 16 queries) = K^T Q     one v_mfma_f32_16x16x4_f32 per 4 head channels; the lane
//             holds keys 4kq..4kq+3 (registers r) of query column c, so a query's running max / sum need
//             4 in-lane values and two cross-lane steps (xor 16, 32) per 64 keys
//   O^T (d x 16 queries) += V P^T                the D registers of S^T ARE the B operand of this MFMA:
//             MFMA r contracts keys {4kk + r}, its A operand is component r of ONE float4 of V -- P never
//             moves between lanes or through LDS
// Online softmax in base 2 (scores pre-multiplied by scale * log2 e).  d <= 32, any N.
#include <cmath>
#include <type_traits>
#include "rf_common.h"

namespace rf {

template <int DT>   // 16-row tiles of the head dimension: d <= 16 * DT
__global__ void __launch_bounds__(256) token_attn_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                         const float* __restrict__ v, float* __restrict__ out,
                                                         int64_t bstride_qkv, int64_t bstride_out, int heads, int d, int N,
                                                         float scale_log2e, int vec) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, kq = lane >> 4;
    const int i0 = (blockIdx.x * 4 + wave) * 16;
    if (i0 >= N) return;                                  // whole wave (no barriers in this kernel)
    const int hd = blockIdx.y;
    const size_t b = blockIdx.z;
    const float* qh = q + b * bstride_qkv + (size_t)hd * d * N;
    const float* kh = k + b * bstride_qkv + (size_t)hd * d * N;
    const float* vh = v + b * bstride_qkv + (size_t)hd * d * N;
    float* oh = out + b * bstride_out + (size_t)hd * d * N;
    const int KS = (d + 3) >> 2;                          // k-sets of the score contraction (<= 8)

    // B operand of the score MFMAs: lane (c, kq) holds q[4s + kq][i0 + c]
    float bq[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int dd = 4 * s + kq;
        bq[s] = (s < KS && dd < d && i0 + c < N) ? qh[(size_t)dd * N + i0 + c] : 0.f;
    }
    float m = -INFINITY, lsum = 0.f;
    f32x4 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) o[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Rows beyond d need no masks: a k row >= d meets bq = 0 and a v row >= d only feeds output rows that are
    // never stored, so both are read from row d - 1 (finite data) instead.  FULL = the 64 keys all exist
    // (wave-uniform): no per-key masks either.
    auto block = [&](int j0, auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
        // ---- S^T = K^T Q for 4 key tiles
        f32x4 st[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            st[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int j = j0 + 16 * tt + c;               // A operand: lane (c, kq) holds k[4s + kq][j]
            const int jc = FULL ? j : min(j, N - 1);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if (s >= KS) break;
                const int dd = min(4 * s + kq, d - 1);
                const float ak = kh[(size_t)dd * N + jc];
                st[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ak, bq[s], st[tt], 0, 0, 0);
            }
        }
        // ---- online softmax over the 64 keys (this lane: keys j0 + 16 tt + 4 kq + r of query c)
        float mloc = -INFINITY;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = j0 + 16 * tt + 4 * kq + r;
                const float sv = (FULL || j < N) ? st[tt][r] * scale_log2e : -INFINITY;
                st[tt][r] = sv;
                mloc = fmaxf(mloc, sv);
            }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 16));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
        const float mnew = fmaxf(m, mloc);                // finite: every block holds at least one valid key
        const float corr = __builtin_amdgcn_exp2f(m - mnew);
        float psum = 0.f;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(st[tt][r] - mnew);
                st[tt][r] = p;
                psum += p;
            }
        psum += __shfl_xor(psum, 16);
        psum += __shfl_xor(psum, 32);
        lsum = fmaf(lsum, corr, psum);
        m = mnew;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[t][r] *= corr;
        // ---- O^T += V P^T : MFMA r of tile tt contracts keys j0 + 16 tt + 4 kk + r
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int jb = j0 + 16 * tt + 4 * kq;         // A operand: lane (c = head channel, kq) holds v[16t + c][jb + r]
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                const float* vrow = vh + (size_t)min(16 * t + c, d - 1) * N;
                float av[4];
                if (FULL && vec) {
                    const float4 t4 = *reinterpret_cast<const float4*>(vrow + jb);
                    av[0] = t4.x; av[1] = t4.y; av[2] = t4.z; av[3] = t4.w;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) av[r] = vrow[min(jb + r, N - 1)];     // p = 0 beyond N
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) o[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], st[tt][r], o[t], 0, 0, 0);
            }
        }
    };
    int j0 = 0;
    for (; j0 + 64 <= N; j0 += 64) block(j0, std::true_type{});
    if (j0 < N) block(j0, std::false_type{});
    // ---- normalise and store: lane (c, kq) holds head channels 16t + 4kq + r of query i0 + c
    const float inv = 1.0f / lsum;
    if (i0 + c < N) {
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int dd = 16 * t + 4 * kq + r;
                if (dd < d) oh[(size_t)dd * N + i0 + c] = o[t][r] * inv;
            }
    }
}

// ---- luma bias: pooled = avg_pool2d(1 - luma, 3, stride 1, padding 1) (count_include_pad: always / 9),
// per-block sums in a fixed order for the per-image mean
__global__ void __launch_bounds__(256) luma_pool_kernel(const float* __restrict__ luma, float* __restrict__ pooled,
                                                        float* __restrict__ partial, int h, int w, int nblk) {
    const size_t b = blockIdx.y;
    const int P = h * w;
    const float* lb = luma + b * P;
    __shared__ float red[4];
    float s = 0.f;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p < P) {
        const int y = p / w, x = p - y * w;
        float a = 0.f;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int yy = y + dy, xx = x + dx;
                if (yy >= 0 && yy < h && xx >= 0 && xx < w) a += 1.0f - lb[yy * w + xx];
            }
        s = a / 9.0f;
        pooled[b * P + p] = s;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[b * nblk + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(256) luma_mean_kernel(const float* __restrict__ partial, float* __restrict__ mean, int nblk, int P) {
    const size_t b = blockIdx.x;
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) s += partial[b * nblk + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) mean[b] = red[0] / (float)P;
}

// FiLM + query bias: out[b][part*inner + ch][n] = gamma[b][ch][n] * qkv[b][part*inner + ch][n] + beta[b][ch][n]
//                                                 (+ alpha * (pooled[b][n] - mean[b]) for part 0 = q)
__global__ void __launch_bounds__(256) luma_film_kernel(const float* __restrict__ qkv, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int64_t gb_bstride,
                                                        const float* __restrict__ pooled, const float* __restrict__ mean,
                                                        const float* __restrict__ alpha, float* __restrict__ out,
                                                        int inner, int P) {
    const size_t b = blockIdx.z;
    const int ch = blockIdx.y;
    const float al = (pooled && alpha) ? *alpha : 0.f;
    const float mu = pooled ? mean[b] : 0.f;
    const float* g = gamma + b * gb_bstride + (size_t)ch * P;
    const float* be = beta + b * gb_bstride + (size_t)ch * P;
    for (int n = blockIdx.x * 256 + threadIdx.x; n < P; n += gridDim.x * 256) {
        const float gv = g[n], bv = be[n];
        float qb = pooled ? al * (pooled[b * P + n] - mu) : 0.f;
        asm volatile("" : "+v"(qb));                    // products rounded on their own: no fma contraction
#pragma unroll
        for (int part = 0; part < 3; ++part) {
            const size_t idx = (b * 3 * inner + (size_t)part * inner + ch) * P + n;
            float t = gv * qkv[idx];                      // torch: gamma * q + beta, then q + alpha * invL, every
            asm volatile("" : "+v"(t));                   // operation rounded separately (bit-exact FiLM)
            t += bv;
            if (part == 0 && pooled) t += qb;
            out[idx] = t;
        }
    }
}

// ---- BayerLuma (Attenblock.py:79-138): the three 3x3 "mask" convolutions pick neighbours of the mosaic
//   rggb: r = in[y-1][x-1], g = (in[y-1][x] + in[y][x-1]) / 2, b = in[y][x]      (zero padded; other patterns permute)
// luma = .299 r + .587 g + .114 b, then (luma - min) / (max - min + 1e-6) per image.  Pass 1 writes luma and per-block
// min / max, pass 2 reduces them (min / max are order independent) and normalises.
__device__ __forceinline__ float nofma(float v) { asm volatile("" : "+v"(v)); return v; }   // keeps a product from contracting

__global__ void __launch_bounds__(256) bayer_luma_kernel(const float* __restrict__ in, float* __restrict__ luma,
                                                         float* __restrict__ pmin, float* __restrict__ pmax, int H, int W, int pattern, int nblk) {
    const size_t b = blockIdx.y;
    const int P = H * W;
    const float* ib = in + b * P;
    const int p = blockIdx.x * 256 + threadIdx.x;
    float v = 0.f, lo = INFINITY, hi = -INFINITY;
    if (p < P) {
        const int y = p / W, x = p - y * W;
        const float c = ib[p];
        const float u = y > 0 ? ib[p - W] : 0.f, l = x > 0 ? ib[p - 1] : 0.f, ul = (y > 0 && x > 0) ? ib[p - W - 1] : 0.f;
        // tap (0,0) = ul, (0,1) = u, (1,0) = l, (1,1) = c
        float r, g, bl;
        if (pattern == 0) { r = ul; g = nofma(0.5f * u) + nofma(0.5f * l); bl = c; }          // rggb
        else if (pattern == 1) { bl = ul; g = nofma(0.5f * u) + nofma(0.5f * l); r = c; }     // bggr
        else if (pattern == 2) { g = nofma(0.5f * ul) + nofma(0.5f * c); r = u; bl = l; }     // grbg
        else { g = nofma(0.5f * ul) + nofma(0.5f * c); bl = u; r = l; }                       // gbrg
        v = (nofma(r * 0.299f) + nofma(g * 0.587f)) + nofma(bl * 0.114f);
        luma[b * P + p] = v;
        lo = hi = v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
    __shared__ float rl[4], rh[4];
    if ((threadIdx.x & 63) == 0) { rl[threadIdx.x >> 6] = lo; rh[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        pmin[b * nblk + blockIdx.x] = fminf(fminf(rl[0], rl[1]), fminf(rl[2], rl[3]));
        pmax[b * nblk + blockIdx.x] = fmaxf(fmaxf(rh[0], rh[1]), fmaxf(rh[2], rh[3]));
    }
}

__global__ void __launch_bounds__(256) bayer_luma_norm_kernel(float* __restrict__ luma, const float* __restrict__ pmin,
                                                              const float* __restrict__ pmax, int P, int nblk) {
    const size_t b = blockIdx.y;
    __shared__ float rl[256], rh[256];
    float lo = INFINITY, hi = -INFINITY;
    for (int i = threadIdx.x; i < nblk; i += 256) { lo = fminf(lo, pmin[b * nblk + i]); hi = fmaxf(hi, pmax[b * nblk + i]); }
    rl[threadIdx.x] = lo; rh[threadIdx.x] = hi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { rl[threadIdx.x] = fminf(rl[threadIdx.x], rl[threadIdx.x + o]); rh[threadIdx.x] = fmaxf(rh[threadIdx.x], rh[threadIdx.x + o]); }
        __syncthreads();
    }
    const float mn = rl[0], den = (rh[0] - rl[0]) + 1e-6f;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < P; p += gridDim.x * 256) luma[b * P + p] = (luma[b * P + p] - mn) / den;
}

}  // namespace rf

using namespace rf;

extern "C" {

int rf_token_attn(const float* q, const float* k, const float* v, float* out, long long bstride_qkv, long long bstride_out,
                  int B, int heads, int d, int N, float scale, void* stream) {
    RF_CHECK_ARG(q && k && v && out && B > 0 && B <= 65535 && heads > 0 && heads <= 65535 && N > 0, "token_attn: bad arguments");
    RF_CHECK_ARG(d >= 1 && d <= 32, "token_attn: head dimension %d not in 1..32", d);
    hipStream_t st = (hipStream_t)stream;
    const int vec = (N % 4 == 0) && aligned16(v) && (bstride_qkv % 4 == 0);
    const dim3 grid((unsigned)cdiv(N, 64), (unsigned)heads, (unsigned)B);
    const double work = 4.0 * (double)N * N * d * heads * B;
    ProfScope prof(st, d <= 16 ? "token_attn_kernel<1>" : "token_attn_kernel<2>", work, 16.0 * (double)B * heads * d * N);
    const float sl2 = scale * 1.44269504088896340736f;
    if (d <= 16) token_attn_kernel<1><<<grid, 256, 0, st>>>(q, k, v, out, bstride_qkv, bstride_out, heads, d, N, sl2, vec);
    else token_attn_kernel<2><<<grid, 256, 0, st>>>(q, k, v, out, bstride_qkv, bstride_out, heads, d, N, sl2, vec);
    return check_launch("token_attn");
}

int rf_luma_film_scratch_bytes(int B, int h, int w, size_t* bytes) {
    RF_CHECK_ARG(bytes && B > 0 && h > 0 && w > 0, "luma_film_scratch_bytes: bad arguments");
    const size_t P = (size_t)h * w;
    *bytes = sizeof(float) * (align_up((size_t)B * P, 64) + align_up((size_t)B * cdiv((int)P, 256), 64) + align_up((size_t)B, 64));
    return RF_OK;
}

int rf_luma_film(const float* qkv, const float* gamma, const float* beta, long long gb_bstride, const float* luma, const float* alpha,
                 float* out, void* scratch, int B, int inner, int h, int w, void* stream) {
    RF_CHECK_ARG(qkv && gamma && beta && out && B > 0 && B <= 65535 && inner > 0 && inner <= 65535 && h > 0 && w > 0, "luma_film: bad arguments");
    RF_CHECK_ARG(!luma || scratch, "luma_film: the luma bias needs the scratch buffer");
    hipStream_t st = (hipStream_t)stream;
    const int P = h * w, nblk = cdiv(P, 256);
    float* pooled = nullptr;
    float* mean = nullptr;
    ProfScope prof(st, "luma_film(3 kernels)", 0.0, 4.0 * B * (8.0 * inner + 2.0) * P);
    if (luma) {
        pooled = static_cast<float*>(scratch);
        float* partial = pooled + align_up((size_t)B * P, 64);
        mean = partial + align_up((size_t)B * nblk, 64);
        luma_pool_kernel<<<dim3((unsigned)nblk, (unsigned)B), 256, 0, st>>>(luma, pooled, partial, h, w, nblk);
        luma_mean_kernel<<<dim3((unsigned)B), 256, 0, st>>>(partial, mean, nblk, P);
    }
    int gx = cdiv(P, 256);
    if (gx > 64) gx = 64;
    luma_film_kernel<<<dim3((unsigned)gx, (unsigned)inner, (unsigned)B), 256, 0, st>>>(qkv, gamma, beta, gb_bstride, pooled, mean,
                                                                                    luma ? alpha : nullptr, out, inner, P);
    return check_launch("luma_film");
}

int rf_bayer_luma_scratch_bytes(int B, int H, int W, size_t* bytes) {
    RF_CHECK_ARG(bytes && B > 0 && H > 0 && W > 0, "bayer_luma_scratch_bytes: bad arguments");
    *bytes = sizeof(float) * 2 * align_up((size_t)B * cdiv(H * W, 256), 64);
    return RF_OK;
}

int rf_bayer_luma(const float* mosaic, float* luma, void* scratch, int B, int H, int W, int pattern, void* stream) {
    RF_CHECK_ARG(mosaic && luma && scratch && B > 0 && B <= 65535 && H > 0 && W > 0, "bayer_luma: bad arguments");
    RF_CHECK_ARG(pattern >= 0 && pattern <= 3, "bayer_luma: pattern %d (0 rggb, 1 bggr, 2 grbg, 3 gbrg)", pattern);
    hipStream_t st = (hipStream_t)stream;
    const int P = H * W, nblk = cdiv(P, 256);
    float* pmin = static_cast<float*>(scratch);
    float* pmax = pmin + align_up((size_t)B * nblk, 64);
    ProfScope prof(st, "bayer_luma(2 kernels)", 0.0, 16.0 * B * P);
    bayer_luma_kernel<<<dim3((unsigned)nblk, (unsigned)B), 256, 0, st>>>(mosaic, luma, pmin, pmax, H, W, pattern, nblk);
    int gx = nblk > 1024 ? 1024 : nblk;
    bayer_luma_norm_kernel<<<dim3((unsigned)gx, (unsigned)B), 256, 0, st>>>(luma, pmin, pmax, P, nblk);
    return check_launch("bayer_luma");
}
}  // extern "C"
