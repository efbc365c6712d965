// HBM-bound operators of the RawFormer path: Bayer pack / unpack, 2x2 wavelet analysis and
// synthesis, per-pixel LayerNorm, depthwise 3x3.  One lane owns 4 consecutive pixels of a row
// (16-byte accesses, 1 KiB per wave instruction) whenever the row length allows; a scalar
// path covers ragged widths (e.g. w = 266 at level 3 of a 1424x2128 frame).
#include <cstdlib>
#include "rf_common.h"

namespace rf {

static constexpr int kBlock = 256;
static inline int grid_for(size_t items) {
    size_t g = (items + kBlock - 1) / kBlock;
    if (g > 256 * 16) g = 256 * 16;   // grid-stride the rest: 16 blocks per CU keeps the chip full
    if (g < 1) g = 1;
    return (int)g;
}

// ------------------------------------------------------------------------------------------
// a1: downshuffle(var, 2)  (RawFomer_WFB_FFAB/model.py:287-298)
//   out[b][4c + 2i + j][y][x] = in[b][c][2y + i][2x + j]
// Lane owns a 2x8 input patch -> one float4 for each of the four output planes.
// ------------------------------------------------------------------------------------------
template <int VEC>
__global__ void __launch_bounds__(kBlock) pixel_unshuffle2_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                  int planes, int h, int w) {
    const int wv = w / VEC;
    const size_t items = (size_t)planes * h * wv;
    for (size_t it = blockIdx.x * (size_t)kBlock + threadIdx.x; it < items; it += (size_t)gridDim.x * kBlock) {
        const int xv = (int)(it % wv);
        const int y = (int)((it / wv) % h);
        const size_t pl = it / ((size_t)wv * h);
        const float* r0 = in + (pl * 2 * h + 2 * y) * (size_t)(2 * w) + 2 * xv * VEC;
        const float* r1 = r0 + 2 * w;
        float* o = out + (pl * 4 * h + y) * (size_t)w + xv * VEC;
        const size_t ps = (size_t)h * w;
        if constexpr (VEC == 4) {
            const float4 a0 = *reinterpret_cast<const float4*>(r0), a1 = *reinterpret_cast<const float4*>(r0 + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(r1), b1 = *reinterpret_cast<const float4*>(r1 + 4);
            *reinterpret_cast<float4*>(o) = make_float4(a0.x, a0.z, a1.x, a1.z);
            *reinterpret_cast<float4*>(o + ps) = make_float4(a0.y, a0.w, a1.y, a1.w);
            *reinterpret_cast<float4*>(o + 2 * ps) = make_float4(b0.x, b0.z, b1.x, b1.z);
            *reinterpret_cast<float4*>(o + 3 * ps) = make_float4(b0.y, b0.w, b1.y, b1.w);
        } else {
            o[0] = r0[0]; o[ps] = r0[1]; o[2 * ps] = r1[0]; o[3 * ps] = r1[1];
        }
    }
}

int launch_pixel_unshuffle2(const float* in, float* out, int B, int C, int h, int w, hipStream_t st) {
    const int planes = B * C;
    ProfScope prof(st, "pixel_unshuffle2_kernel", 0.0, 32.0 * planes * h * w);
    if ((w & 3) == 0 && aligned16(in) && aligned16(out))
        pixel_unshuffle2_kernel<4><<<grid_for((size_t)planes * h * (w / 4)), kBlock, 0, st>>>(in, out, planes, h, w);
    else
        pixel_unshuffle2_kernel<1><<<grid_for((size_t)planes * h * w), kBlock, 0, st>>>(in, out, planes, h, w);
    return check_launch("pixel_unshuffle2");
}

// a10: nn.PixelShuffle(2)   out[b][c][2y + i][2x + j] = in[b][4c + 2i + j][y][x]
template <int VEC>
__global__ void __launch_bounds__(kBlock) pixel_shuffle2_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                int planes, int h, int w) {
    const int wv = w / VEC;
    const size_t items = (size_t)planes * h * wv;
    for (size_t it = blockIdx.x * (size_t)kBlock + threadIdx.x; it < items; it += (size_t)gridDim.x * kBlock) {
        const int xv = (int)(it % wv);
        const int y = (int)((it / wv) % h);
        const size_t pl = it / ((size_t)wv * h);
        const float* i0 = in + (pl * 4 * h + y) * (size_t)w + xv * VEC;
        const size_t ps = (size_t)h * w;
        float* r0 = out + (pl * 2 * h + 2 * y) * (size_t)(2 * w) + 2 * xv * VEC;
        float* r1 = r0 + 2 * w;
        if constexpr (VEC == 4) {
            const float4 p0 = *reinterpret_cast<const float4*>(i0), p1 = *reinterpret_cast<const float4*>(i0 + ps);
            const float4 p2 = *reinterpret_cast<const float4*>(i0 + 2 * ps), p3 = *reinterpret_cast<const float4*>(i0 + 3 * ps);
            *reinterpret_cast<float4*>(r0) = make_float4(p0.x, p1.x, p0.y, p1.y);
            *reinterpret_cast<float4*>(r0 + 4) = make_float4(p0.z, p1.z, p0.w, p1.w);
            *reinterpret_cast<float4*>(r1) = make_float4(p2.x, p3.x, p2.y, p3.y);
            *reinterpret_cast<float4*>(r1 + 4) = make_float4(p2.z, p3.z, p2.w, p3.w);
        } else {
            r0[0] = i0[0]; r0[1] = i0[ps]; r1[0] = i0[2 * ps]; r1[1] = i0[3 * ps];
        }
    }
}

int launch_pixel_shuffle2(const float* in, float* out, int B, int C, int h, int w, hipStream_t st) {
    const int planes = B * C;
    ProfScope prof(st, "pixel_shuffle2_kernel", 0.0, 32.0 * planes * h * w);
    if ((w & 3) == 0 && aligned16(in) && aligned16(out))
        pixel_shuffle2_kernel<4><<<grid_for((size_t)planes * h * (w / 4)), kBlock, 0, st>>>(in, out, planes, h, w);
    else
        pixel_shuffle2_kernel<1><<<grid_for((size_t)planes * h * w), kBlock, 0, st>>>(in, out, planes, h, w);
    return check_launch("pixel_shuffle2");
}

// ------------------------------------------------------------------------------------------
// a11 / a13 / a14: 2x2 analysis   out_s[y][x] = sum_t K[s][t] * in[2y + (t >> 1)][2x + (t & 1)]
// Each lane owns whole 2x2 tiles (two 16-byte row loads per row), so the Haar butterflies are
// register adds; no cross-lane traffic is needed.  Algorithmic bytes: 8 per input element.
// ------------------------------------------------------------------------------------------
struct K16 { float v[16]; };

__device__ __forceinline__ void analysis4(const K16& k, int exact, float t0, float t1, float t2, float t3,
                                          float& s0, float& s1, float& s2, float& s3) {
    if (exact) {
        // dwt_init, RawFomer_WFB_FFAB/blocks.py:104-113: x1=(0,0) x2=(1,0) x3=(0,1) x4=(1,1), each /2,
        // then the sums exactly in the order Python evaluates them.
        const float x1 = t0 / 2, x2 = t2 / 2, x3 = t1 / 2, x4 = t3 / 2;
        s0 = ((x1 + x2) + x3) + x4;
        s1 = ((-x1 - x2) + x3) + x4;
        s2 = ((-x1 + x2) - x3) + x4;
        s3 = ((x1 - x2) - x3) + x4;
    } else {
        s0 = fmaf(k.v[3], t3, fmaf(k.v[2], t2, fmaf(k.v[1], t1, k.v[0] * t0)));
        s1 = fmaf(k.v[7], t3, fmaf(k.v[6], t2, fmaf(k.v[5], t1, k.v[4] * t0)));
        s2 = fmaf(k.v[11], t3, fmaf(k.v[10], t2, fmaf(k.v[9], t1, k.v[8] * t0)));
        s3 = fmaf(k.v[15], t3, fmaf(k.v[14], t2, fmaf(k.v[13], t1, k.v[12] * t0)));
    }
}

template <int VEC>
__global__ void __launch_bounds__(kBlock) dwt2x2_kernel(const float* __restrict__ in, float* __restrict__ out, K16 k,
                                                         int exact, int B, int C, int h, int w, int hin, int win,
                                                         size_t ob, size_t os, size_t oc) {
    const int wv = w / VEC;
    const size_t items = (size_t)B * C * h * wv;
    for (size_t it = blockIdx.x * (size_t)kBlock + threadIdx.x; it < items; it += (size_t)gridDim.x * kBlock) {
        const int xv = (int)(it % wv);
        const int y = (int)((it / wv) % h);
        const size_t pl = it / ((size_t)wv * h);
        const int c = (int)(pl % C);
        const int b = (int)(pl / C);
        float* o = out + b * ob + c * oc + (size_t)y * w + xv * VEC;
        const float* base = in + pl * (size_t)hin * win;
        if constexpr (VEC == 4) {
            const float* r0 = base + (size_t)(2 * y) * win + 8 * xv;
            const float* r1 = r0 + win;
            const float4 a0 = *reinterpret_cast<const float4*>(r0), a1 = *reinterpret_cast<const float4*>(r0 + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(r1), b1 = *reinterpret_cast<const float4*>(r1 + 4);
            float4 q0, q1, q2, q3;
            analysis4(k, exact, a0.x, a0.y, b0.x, b0.y, q0.x, q1.x, q2.x, q3.x);
            analysis4(k, exact, a0.z, a0.w, b0.z, b0.w, q0.y, q1.y, q2.y, q3.y);
            analysis4(k, exact, a1.x, a1.y, b1.x, b1.y, q0.z, q1.z, q2.z, q3.z);
            analysis4(k, exact, a1.z, a1.w, b1.z, b1.w, q0.w, q1.w, q2.w, q3.w);
            *reinterpret_cast<float4*>(o) = q0;
            *reinterpret_cast<float4*>(o + os) = q1;
            *reinterpret_cast<float4*>(o + 2 * os) = q2;
            *reinterpret_cast<float4*>(o + 3 * os) = q3;
        } else {
            // reflect padding on the right / bottom when the input size is odd (HaarDWT)
            const int y0 = 2 * y, y1 = (2 * y + 1 < hin) ? 2 * y + 1 : hin - 2;
            const int x0 = 2 * xv, x1 = (2 * xv + 1 < win) ? 2 * xv + 1 : win - 2;
            const float t0 = base[(size_t)y0 * win + x0], t1 = base[(size_t)y0 * win + x1];
            const float t2 = base[(size_t)y1 * win + x0], t3 = base[(size_t)y1 * win + x1];
            float s0, s1, s2, s3;
            analysis4(k, exact, t0, t1, t2, t3, s0, s1, s2, s3);
            o[0] = s0; o[os] = s1; o[2 * os] = s2; o[3 * os] = s3;
        }
    }
}

int launch_dwt2x2(const float* in, float* out, const float kk[16], int layout, int exact_haar,
                  int B, int C, int h, int w, int hin, int win, hipStream_t st) {
    K16 k;
    for (int i = 0; i < 16; ++i) k.v[i] = kk[i];
    const size_t hw = (size_t)h * w;
    const size_t ob = layout == 1 ? 4 * (size_t)C * hw : (size_t)C * hw;
    const size_t os = layout == 1 ? (size_t)C * hw : (size_t)B * C * hw;
    const bool vec = (w & 3) == 0 && hin == 2 * h && win == 2 * w && aligned16(in) && aligned16(out);
    // algorithmic traffic: 4 B read + 4 B written per input element
    ProfScope prof(st, vec ? "dwt2x2_kernel<4>" : "dwt2x2_kernel<1>", 7.0 * 4 * B * C * hw, 8.0 * 4 * B * C * hw);
    if (vec)
        dwt2x2_kernel<4><<<grid_for((size_t)B * C * h * (w / 4)), kBlock, 0, st>>>(in, out, k, exact_haar, B, C, h, w, hin, win, ob, os, hw);
    else
        dwt2x2_kernel<1><<<grid_for((size_t)B * C * hw), kBlock, 0, st>>>(in, out, k, exact_haar, B, C, h, w, hin, win, ob, os, hw);
    return check_launch("dwt2x2");
}

// a12 / a13: synthesis   out[2y + (t >> 1)][2x + (t & 1)] = sum_s K[s][t] * band_s[y][x]
__device__ __forceinline__ void synthesis4(const K16& k, int exact, float b0, float b1, float b2, float b3,
                                           float& t0, float& t1, float& t2, float& t3) {
    if (exact) {
        // iwt_init, RawFomer_WFB_FFAB/blocks.py:123-134
        const float x1 = b0 / 2, x2 = b1 / 2, x3 = b2 / 2, x4 = b3 / 2;
        t0 = ((x1 - x2) - x3) + x4;   // (0,0)
        t2 = ((x1 - x2) + x3) - x4;   // (1,0)
        t1 = ((x1 + x2) - x3) - x4;   // (0,1)
        t3 = ((x1 + x2) + x3) + x4;   // (1,1)
    } else {
        t0 = fmaf(k.v[12], b3, fmaf(k.v[8], b2, fmaf(k.v[4], b1, k.v[0] * b0)));
        t1 = fmaf(k.v[13], b3, fmaf(k.v[9], b2, fmaf(k.v[5], b1, k.v[1] * b0)));
        t2 = fmaf(k.v[14], b3, fmaf(k.v[10], b2, fmaf(k.v[6], b1, k.v[2] * b0)));
        t3 = fmaf(k.v[15], b3, fmaf(k.v[11], b2, fmaf(k.v[7], b1, k.v[3] * b0)));
    }
}

template <int VEC>
__global__ void __launch_bounds__(kBlock) idwt2x2_kernel(const float* __restrict__ in, float* __restrict__ out, K16 k,
                                                          int exact, int B, int C, int h, int w,
                                                          size_t ib, size_t is, size_t ic) {
    const int wv = w / VEC;
    const size_t items = (size_t)B * C * h * wv;
    for (size_t it = blockIdx.x * (size_t)kBlock + threadIdx.x; it < items; it += (size_t)gridDim.x * kBlock) {
        const int xv = (int)(it % wv);
        const int y = (int)((it / wv) % h);
        const size_t pl = it / ((size_t)wv * h);
        const int c = (int)(pl % C);
        const int b = (int)(pl / C);
        const float* i0 = in + b * ib + c * ic + (size_t)y * w + xv * VEC;
        float* r0 = out + (pl * 2 * h + 2 * y) * (size_t)(2 * w) + 2 * xv * VEC;
        float* r1 = r0 + 2 * w;
        if constexpr (VEC == 4) {
            const float4 p0 = *reinterpret_cast<const float4*>(i0), p1 = *reinterpret_cast<const float4*>(i0 + is);
            const float4 p2 = *reinterpret_cast<const float4*>(i0 + 2 * is), p3 = *reinterpret_cast<const float4*>(i0 + 3 * is);
            float4 ra, rb, rc, rd;   // row0 cols 0-3, row0 cols 4-7, row1 cols 0-3, row1 cols 4-7
            synthesis4(k, exact, p0.x, p1.x, p2.x, p3.x, ra.x, ra.y, rc.x, rc.y);
            synthesis4(k, exact, p0.y, p1.y, p2.y, p3.y, ra.z, ra.w, rc.z, rc.w);
            synthesis4(k, exact, p0.z, p1.z, p2.z, p3.z, rb.x, rb.y, rd.x, rd.y);
            synthesis4(k, exact, p0.w, p1.w, p2.w, p3.w, rb.z, rb.w, rd.z, rd.w);
            *reinterpret_cast<float4*>(r0) = ra;
            *reinterpret_cast<float4*>(r0 + 4) = rb;
            *reinterpret_cast<float4*>(r1) = rc;
            *reinterpret_cast<float4*>(r1 + 4) = rd;
        } else {
            float t0, t1, t2, t3;
            synthesis4(k, exact, i0[0], i0[is], i0[2 * is], i0[3 * is], t0, t1, t2, t3);
            r0[0] = t0; r0[1] = t1; r1[0] = t2; r1[1] = t3;
        }
    }
}

int launch_idwt2x2(const float* in, float* out, const float kk[16], int layout, int exact_haar,
                   int B, int C, int h, int w, hipStream_t st) {
    K16 k;
    for (int i = 0; i < 16; ++i) k.v[i] = kk[i];
    const size_t hw = (size_t)h * w;
    const size_t ib = layout == 1 ? 4 * (size_t)C * hw : (size_t)C * hw;
    const size_t is = layout == 1 ? (size_t)C * hw : (size_t)B * C * hw;
    ProfScope prof(st, ((w & 3) == 0 && aligned16(in) && aligned16(out)) ? "idwt2x2_kernel<4>" : "idwt2x2_kernel<1>",
                   7.0 * 4 * B * C * hw, 8.0 * 4 * B * C * hw);
    if ((w & 3) == 0 && aligned16(in) && aligned16(out))
        idwt2x2_kernel<4><<<grid_for((size_t)B * C * h * (w / 4)), kBlock, 0, st>>>(in, out, k, exact_haar, B, C, h, w, ib, is, hw);
    else
        idwt2x2_kernel<1><<<grid_for((size_t)B * C * hw), kBlock, 0, st>>>(in, out, k, exact_haar, B, C, h, w, ib, is, hw);
    return check_launch("idwt2x2");
}

// ------------------------------------------------------------------------------------------
// a4: LayerNorm over the channel axis of NCHW (per pixel), biased variance, two-pass.
// Lanes run along pixels, so every channel row is read coalesced.  (The forward itself uses
// the prologue fused into the 1x1 GEMM; this is the standalone operator.)
// ------------------------------------------------------------------------------------------
template <int VEC>
__global__ void __launch_bounds__(kBlock) layernorm2d_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                              const float* __restrict__ gw, const float* __restrict__ gb,
                                                              float eps, int B, int C, int P) {
    const int pv = P / VEC;
    const size_t items = (size_t)B * pv;
    const float invC = 1.0f / (float)C;
    for (size_t it = blockIdx.x * (size_t)kBlock + threadIdx.x; it < items; it += (size_t)gridDim.x * kBlock) {
        const int p = (int)(it % pv) * VEC;
        const size_t b = it / pv;
        const float* x = in + b * (size_t)C * P + p;
        float* o = out + b * (size_t)C * P + p;
        float mu[VEC], var[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) { mu[v] = 0.f; var[v] = 0.f; }
        for (int c = 0; c < C; ++c) {
            if constexpr (VEC == 4) {
                const float4 t = *reinterpret_cast<const float4*>(x + (size_t)c * P);
                mu[0] += t.x; mu[1] += t.y; mu[2] += t.z; mu[3] += t.w;
            } else {
                mu[0] += x[(size_t)c * P];
            }
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) mu[v] *= invC;
        for (int c = 0; c < C; ++c) {
            if constexpr (VEC == 4) {
                const float4 t = *reinterpret_cast<const float4*>(x + (size_t)c * P);
                var[0] += (t.x - mu[0]) * (t.x - mu[0]); var[1] += (t.y - mu[1]) * (t.y - mu[1]);
                var[2] += (t.z - mu[2]) * (t.z - mu[2]); var[3] += (t.w - mu[3]) * (t.w - mu[3]);
            } else {
                const float d = x[(size_t)c * P] - mu[0];
                var[0] += d * d;
            }
        }
        float rs[VEC], sh[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            rs[v] = 1.0f / sqrtf(var[v] * invC + eps);
            sh[v] = gb ? mu[v] : 0.f;   // BiasFree_LayerNorm keeps the mean in the numerator
        }
        for (int c = 0; c < C; ++c) {
            const float g = gw[c], bb = gb ? gb[c] : 0.f;
            if constexpr (VEC == 4) {
                const float4 t = *reinterpret_cast<const float4*>(x + (size_t)c * P);
                float4 r;
                r.x = (t.x - sh[0]) * rs[0] * g + bb; r.y = (t.y - sh[1]) * rs[1] * g + bb;
                r.z = (t.z - sh[2]) * rs[2] * g + bb; r.w = (t.w - sh[3]) * rs[3] * g + bb;
                *reinterpret_cast<float4*>(o + (size_t)c * P) = r;
            } else {
                o[(size_t)c * P] = (x[(size_t)c * P] - sh[0]) * rs[0] * g + bb;
            }
        }
    }
}

// Register-resident form for the channel counts of the model (C = CPL * KS <= 128): the 64 lanes of a wave are 64 / KS groups of
// 4 consecutive pixels x KS channel slices; a lane keeps its CPL channels of its 4 pixels in registers (one 16-byte load per
// channel, all issued before the first use), the exact two-pass statistics are completed with KS-wide butterflies, and the
// result is stored once: x is read ONCE and written once (the generic kernel above reads it three times, channel by channel,
// with one dependent load in flight: 0.15 of HBM at 4 x 32 x 512 x 512).  Same arithmetic order per pixel as the generic
// kernel up to the order of the channel sum.
template <int CPL, int KS>
__global__ void __launch_bounds__(kBlock) layernorm2d_reg_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                  const float* __restrict__ gw, const float* __restrict__ gb,
                                                                  float eps, int C, int P) {
    constexpr int G = 64 / KS;                     // pixel groups per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane % G, ks = lane / G;
    const size_t img = blockIdx.y;
    const float* xb = in + img * (size_t)C * P;
    float* ob = out + img * (size_t)C * P;
    const int ngroups = P / 4, per_wg = 4 * G;
    float gam[CPL], bet[CPL];
#pragma unroll
    for (int s = 0; s < CPL; ++s) { gam[s] = gw[ks + KS * s]; bet[s] = gb ? gb[ks + KS * s] : 0.f; }
    const float invC = 1.0f / (float)C;
    for (int g0 = blockIdx.x * per_wg; g0 < ngroups; g0 += gridDim.x * per_wg) {
        const int gi = g0 + wave * G + j;
        const bool ok = gi < ngroups;
        const size_t off = (size_t)(ok ? gi : 0) * 4;
        float4 x[CPL];
#pragma unroll
        for (int s = 0; s < CPL; ++s) x[s] = *reinterpret_cast<const float4*>(xb + (size_t)(ks + KS * s) * P + off);
        float mu[4] = {0.f, 0.f, 0.f, 0.f}, var[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < CPL; ++s) { mu[0] += x[s].x; mu[1] += x[s].y; mu[2] += x[s].z; mu[3] += x[s].w; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int o = G; o < 64; o <<= 1) mu[q] += __shfl_xor(mu[q], o);
            mu[q] *= invC;
        }
#pragma unroll
        for (int s = 0; s < CPL; ++s) {
            const float d0 = x[s].x - mu[0], d1 = x[s].y - mu[1], d2 = x[s].z - mu[2], d3 = x[s].w - mu[3];
            var[0] = fmaf(d0, d0, var[0]); var[1] = fmaf(d1, d1, var[1]); var[2] = fmaf(d2, d2, var[2]); var[3] = fmaf(d3, d3, var[3]);
        }
        float rs[4], sh[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int o = G; o < 64; o <<= 1) var[q] += __shfl_xor(var[q], o);
            rs[q] = 1.0f / sqrtf(var[q] * invC + eps);
            sh[q] = gb ? mu[q] : 0.f;              // BiasFree_LayerNorm keeps the mean in the numerator
        }
        if (ok) {
#pragma unroll
            for (int s = 0; s < CPL; ++s) {
                float4 r;
                r.x = (x[s].x - sh[0]) * rs[0] * gam[s] + bet[s]; r.y = (x[s].y - sh[1]) * rs[1] * gam[s] + bet[s];
                r.z = (x[s].z - sh[2]) * rs[2] * gam[s] + bet[s]; r.w = (x[s].w - sh[3]) * rs[3] * gam[s] + bet[s];
                *reinterpret_cast<float4*>(ob + (size_t)(ks + KS * s) * P + off) = r;
            }
        }
    }
}

template <int CPL, int KS>
static void launch_ln_reg(const float* in, float* out, const float* w, const float* b, float eps, int B, int C, int P, hipStream_t st) {
    const int per_wg = 4 * (64 / KS);
    int gx = cdiv(P / 4, per_wg);
    const int cap = cdiv(256 * 8, B);              // about eight workgroups per CU over the batch, grid-stride beyond
    if (gx > cap) gx = cap;
    layernorm2d_reg_kernel<CPL, KS><<<dim3((unsigned)gx, (unsigned)B), kBlock, 0, st>>>(in, out, w, b, eps, C, P);
}

int launch_layernorm2d(const float* in, float* out, const float* w, const float* b, float eps,
                       int B, int C, int P, hipStream_t st) {
    ProfScope prof(st, "layernorm2d_kernel", 8.0 * B * C * P, 8.0 * B * C * P);
    if ((P & 3) == 0 && aligned16(in) && aligned16(out) && B <= 65535) {
        switch (C) {
            case 16: launch_ln_reg<4, 4>(in, out, w, b, eps, B, C, P, st); return check_launch("layernorm2d");
            case 32: launch_ln_reg<8, 4>(in, out, w, b, eps, B, C, P, st); return check_launch("layernorm2d");
            case 48: launch_ln_reg<12, 4>(in, out, w, b, eps, B, C, P, st); return check_launch("layernorm2d");
            case 64: launch_ln_reg<16, 4>(in, out, w, b, eps, B, C, P, st); return check_launch("layernorm2d");
            case 96: launch_ln_reg<12, 8>(in, out, w, b, eps, B, C, P, st); return check_launch("layernorm2d");
            case 128: launch_ln_reg<16, 8>(in, out, w, b, eps, B, C, P, st); return check_launch("layernorm2d");
            case 192: launch_ln_reg<12, 16>(in, out, w, b, eps, B, C, P, st); return check_launch("layernorm2d");
            case 256: launch_ln_reg<16, 16>(in, out, w, b, eps, B, C, P, st); return check_launch("layernorm2d");
            case 384: launch_ln_reg<12, 32>(in, out, w, b, eps, B, C, P, st); return check_launch("layernorm2d");
            case 512: launch_ln_reg<16, 32>(in, out, w, b, eps, B, C, P, st); return check_launch("layernorm2d");
            default: break;
        }
    }
    if ((P & 3) == 0 && aligned16(in) && aligned16(out))
        layernorm2d_kernel<4><<<grid_for((size_t)B * (P / 4)), kBlock, 0, st>>>(in, out, w, b, eps, B, C, P);
    else
        layernorm2d_kernel<1><<<grid_for((size_t)B * P), kBlock, 0, st>>>(in, out, w, b, eps, B, C, P);
    return check_launch("layernorm2d");
}

// ------------------------------------------------------------------------------------------
// Depthwise 3x3, padding 1 (+bias, optional exact GELU).  A lane computes a 4-wide, ROWS-tall
// strip of one plane: ROWS+2 row reads of (float4 + two edge scalars), so each input row is
// fetched from HBM once and re-served from L1 for the neighbouring strips.
// Algorithmic bytes: 8 per element.
// ------------------------------------------------------------------------------------------

template <int VEC, int ROWS>
__global__ void __launch_bounds__(kBlock) dwconv3x3_kernel(DwConvArgs a) {
    const int w = a.w_, h = a.h;
    const int wv = (w + VEC - 1) / VEC;
    const int hr = (h + ROWS - 1) / ROWS;
    const size_t items = (size_t)a.B * a.C * hr * wv;
    for (size_t it = blockIdx.x * (size_t)kBlock + threadIdx.x; it < items; it += (size_t)gridDim.x * kBlock) {
        const int xv = (int)(it % wv);
        const int yr = (int)((it / wv) % hr);
        const size_t pl = it / ((size_t)wv * hr);
        const int c = (int)(pl % a.C);
        const size_t b = pl / a.C;
        const float* x = a.x + b * a.x_bstride + (size_t)c * h * w;
        float* o = a.out + b * a.out_bstride + (size_t)c * h * w;
        float* o2 = a.out2 ? a.out2 + b * a.out_bstride + (size_t)c * h * w : nullptr;
        float k[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) k[i] = a.w[c * 9 + i];
        const float bias = a.bias ? a.bias[c] : 0.f;
        const int x0 = xv * VEC, y0 = yr * ROWS;
        float acc[ROWS][VEC];
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[r][v] = bias;
#pragma unroll
        for (int rr = 0; rr < ROWS + 2; ++rr) {
            const int y = y0 + rr - 1;
            float v[VEC + 2];
            if constexpr (VEC == 4) {
                // Row segment as one 16-byte load plus the two edge taps, all three UNCONDITIONAL (clamped
                // addresses, values masked afterwards): a load inside a conditional block is followed by
                // s_waitcnt vmcnt(0), which serialised the 10 rows of a thread into 10 round trips.
                const bool rok = y >= 0 && y < h;
                const float* row = x + (size_t)(rok ? y : 0) * w;
                const float4 t = *reinterpret_cast<const float4*>(row + x0);
                const float l = row[x0 > 0 ? x0 - 1 : 0], r = row[x0 + 4 < w ? x0 + 4 : x0];
                v[1] = rok ? t.x : 0.f; v[2] = rok ? t.y : 0.f; v[3] = rok ? t.z : 0.f; v[4] = rok ? t.w : 0.f;
                v[0] = (rok && x0 > 0) ? l : 0.f;
                v[5] = (rok && x0 + 4 < w) ? r : 0.f;
            } else if constexpr (VEC == 2) {      // even widths that are not multiples of 4 (w = 266 at level 3 of a 2848 x 4256 frame)
                const bool rok = y >= 0 && y < h;
                const float* row = x + (size_t)(rok ? y : 0) * w;
                const float2 t = *reinterpret_cast<const float2*>(row + x0);
                const float l = row[x0 > 0 ? x0 - 1 : 0], r = row[x0 + 2 < w ? x0 + 2 : x0];
                v[1] = rok ? t.x : 0.f; v[2] = rok ? t.y : 0.f;
                v[0] = (rok && x0 > 0) ? l : 0.f;
                v[3] = (rok && x0 + 2 < w) ? r : 0.f;
            } else if (y >= 0 && y < h) {
                const float* row = x + (size_t)y * w;
                v[1] = row[x0];
                v[0] = x0 > 0 ? row[x0 - 1] : 0.f;
                v[VEC + 1] = x0 + VEC < w ? row[x0 + VEC] : 0.f;
            } else {
#pragma unroll
                for (int i = 0; i < VEC + 2; ++i) v[i] = 0.f;
            }
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                const int ky = rr - r;   // kernel row that maps input row rr to output row r
                if (ky >= 0 && ky < 3) {
#pragma unroll
                    for (int p = 0; p < VEC; ++p)
                        acc[r][p] = fmaf(k[ky * 3 + 2], v[p + 2], fmaf(k[ky * 3 + 1], v[p + 1], fmaf(k[ky * 3], v[p], acc[r][p])));
                }
            }
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int y = y0 + r;
            if (y < h) {
                if (a.gelu) {
#pragma unroll
                    for (int p = 0; p < VEC; ++p) acc[r][p] = gelu_fast(acc[r][p]);
                }
                if constexpr (VEC == 4)
                    *reinterpret_cast<float4*>(o + (size_t)y * w + x0) = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
                else if constexpr (VEC == 2)
                    *reinterpret_cast<float2*>(o + (size_t)y * w + x0) = make_float2(acc[r][0], acc[r][1]);
                else
                    o[(size_t)y * w + x0] = acc[r][0];
                if (o2) {      // the training forward keeps both sides of the activation (rf_trainstep.hip)
                    float gv[VEC];
#pragma unroll
                    for (int p = 0; p < VEC; ++p) gv[p] = 0.5f * acc[r][p] * (1.0f + erff(acc[r][p] * 0.70710678118654752440f));
                    if constexpr (VEC == 4)
                        *reinterpret_cast<float4*>(o2 + (size_t)y * w + x0) = make_float4(gv[0], gv[1], gv[2], gv[3]);
                    else if constexpr (VEC == 2)
                        *reinterpret_cast<float2*>(o2 + (size_t)y * w + x0) = make_float2(gv[0], gv[1]);
                    else
                        o2[(size_t)y * w + x0] = gv[0];
                }
            }
        }
    }
}

// The same strips with the two edge taps of every row taken from the NEIGHBOURING LANES instead of from memory.  In the kernel
// above a row costs three load instructions -- the 16-byte segment and two dwords one pixel outside it -- and the dword loads
// touch as many cache lines as the segment load: 3 x the L1 work of the data, and the kernel sat at 0.55 x HBM.  Here a wave
// covers 62 consecutive strips of the x-fastest strip order with lanes 1 .. 62; lanes 0 and 63 load the strips before and after
// them and only lend their edge pixels (3 % of the lanes do no arithmetic).  Strips at the ends of a row need no neighbour (zero
// padding), so the order may run on across rows and planes.  w % 4 == 0.
template <int ROWS>
__global__ void __launch_bounds__(kBlock) dwconv3x3_halo_kernel(DwConvArgs a) {
    const int w = a.w_, h = a.h;
    const int wv = w >> 2;
    const int hr = (h + ROWS - 1) / ROWS;
    const long items = (long)a.B * a.C * hr * wv;
    const int lane = threadIdx.x & 63;
    const long nwave = (items + 61) / 62;
    for (long wid = (blockIdx.x * (long)kBlock + threadIdx.x) >> 6; wid < nwave; wid += ((long)gridDim.x * kBlock) >> 6) {
        const long itr = wid * 62 + lane - 1;                        // lanes 0 / 63: the strips before / after this wave's 62
        const bool mine = lane >= 1 && lane <= 62 && itr < items;
        const long it = itr < 0 ? 0 : (itr < items ? itr : items - 1);
        const int xv = (int)(it % wv);
        const int yr = (int)((it / wv) % hr);
        const size_t pl = (size_t)(it / ((long)wv * hr));
        const int c = (int)(pl % a.C);
        const size_t b = pl / a.C;
        const float* x = a.x + b * a.x_bstride + (size_t)c * h * w;
        float* o = a.out + b * a.out_bstride + (size_t)c * h * w;
        float* o2 = a.out2 ? a.out2 + b * a.out_bstride + (size_t)c * h * w : nullptr;
        float k[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) k[i] = a.w[c * 9 + i];
        const float bias = a.bias ? a.bias[c] : 0.f;
        const int x0 = xv * 4, y0 = yr * ROWS;
        const bool has_l = x0 > 0, has_r = x0 + 4 < w;              // inside a row the neighbouring lane holds the neighbouring strip
        float acc[ROWS][4];
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[r][v] = bias;
#pragma unroll
        for (int rr = 0; rr < ROWS + 2; ++rr) {
            const int y = y0 + rr - 1;
            const bool rok = y >= 0 && y < h;
            const float4 t = *reinterpret_cast<const float4*>(x + (size_t)(rok ? y : 0) * w + x0);
            const float lft = __shfl_up(t.w, 1), rgt = __shfl_down(t.x, 1);
            float v[6];
            v[0] = (rok && has_l) ? lft : 0.f;
            v[1] = rok ? t.x : 0.f; v[2] = rok ? t.y : 0.f; v[3] = rok ? t.z : 0.f; v[4] = rok ? t.w : 0.f;
            v[5] = (rok && has_r) ? rgt : 0.f;
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                const int ky = rr - r;
                if (ky >= 0 && ky < 3) {
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        acc[r][p] = fmaf(k[ky * 3 + 2], v[p + 2], fmaf(k[ky * 3 + 1], v[p + 1], fmaf(k[ky * 3], v[p], acc[r][p])));
                }
            }
        }
        if (!mine) continue;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int y = y0 + r;
            if (y < h) {
                if (a.gelu) {
#pragma unroll
                    for (int p = 0; p < 4; ++p) acc[r][p] = gelu_fast(acc[r][p]);
                }
                *reinterpret_cast<float4*>(o + (size_t)y * w + x0) = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
                if (o2) {
                    float gv[4];
#pragma unroll
                    for (int p = 0; p < 4; ++p) gv[p] = 0.5f * acc[r][p] * (1.0f + erff(acc[r][p] * 0.70710678118654752440f));
                    *reinterpret_cast<float4*>(o2 + (size_t)y * w + x0) = make_float4(gv[0], gv[1], gv[2], gv[3]);
                }
            }
        }
    }
}

int launch_dwconv3x3(const DwConvArgs& a, hipStream_t st) {
    RF_CHECK_ARG(!a.out2 || (!a.gelu && aligned16(a.out2)), "dwconv3x3: out2 is the activated copy of a pre-activation out (gelu = 0), 16-byte aligned");
    const bool vec = (a.w_ & 3) == 0 && aligned16(a.x) && aligned16(a.out) && (a.x_bstride & 3) == 0 && (a.out_bstride & 3) == 0;
    const double el = (double)a.B * a.C * a.h * a.w_;
    const bool tall = vec && a.h % 8 == 0;   // 8 output rows per thread: 10 row loads per 8 rows instead of 6 per 4
    const bool vec2 = !vec && (a.w_ & 1) == 0 && aligned16(a.x) && aligned16(a.out) && (a.x_bstride & 1) == 0 && (a.out_bstride & 1) == 0 &&
                      (((size_t)a.h * a.w_) & 1) == 0;
    ProfScope prof(st, vec ? (tall ? "dwconv3x3_halo_kernel<8>" : "dwconv3x3_halo_kernel<4>") : vec2 ? "dwconv3x3_kernel<2, 8>" : "dwconv3x3_kernel<1, 4>",
                   18.0 * el, 8.0 * el);
    if (tall) {
        const size_t items = (size_t)a.B * a.C * (a.h / 8) * (a.w_ / 4);
        dwconv3x3_halo_kernel<8><<<grid_for((items + 61) / 62 * 64), kBlock, 0, st>>>(a);
    } else if (vec) {
        const size_t items = (size_t)a.B * a.C * cdiv(a.h, 4) * (a.w_ / 4);
        dwconv3x3_halo_kernel<4><<<grid_for((items + 61) / 62 * 64), kBlock, 0, st>>>(a);
    } else if (vec2) {
        const size_t items = (size_t)a.B * a.C * cdiv(a.h, 8) * (a.w_ / 2);
        dwconv3x3_kernel<2, 8><<<grid_for(items), kBlock, 0, st>>>(a);
    } else {
        const size_t items = (size_t)a.B * a.C * cdiv(a.h, 4) * a.w_;
        dwconv3x3_kernel<1, 4><<<grid_for(items), kBlock, 0, st>>>(a);
    }
    return check_launch("dwconv3x3");
}

}  // namespace rf
