// TrueColorRawFormer-specific kernels (BayerTORGBColorMultiLvl.py): everything the variant adds around the shared U-Net.
//   tc_front        EnhancedBayerProcessor up to the small convolutions (:100-125): softplus white balance, 3x3 colour
//                   matrix + bias, BT.709 luma and its per-image maximum; writes the inputs of the two conv chains
//   tc_normalise    y / max(amax, eps) (:124), Haar pyramid of y (EnhancedFLCA._pyramid_y, :236-247: LL and high-band
//                   magnitude of every level)
//   tc_guide_level  the seven guidance planes an EnhancedFLCA block sees at its feature size (:255-275):
//                   [y, cr, cb, R, G, y_low (deepest LL), y_high (mean of the resized magnitudes)], bilinear, align_corners=False
//   tc_spatial      feat * (1 + sigmoid(conv3x3(5 planes)) + tanh(sigmoid(conv3x3(y_low)) + tanh(conv3x3(y_high))))   (:277-283)
//   tc_residual     x + 0.2 tanh(res_proj(x)) and the per-block channel sums for the squeeze-excite pooling (:285-292); the
//                   two 1x1 convolutions of res_proj are the GEMM kernels of rf_gemm1x1.hip
//   tc_color_head   CameraAwareColorCorrection (:160-176) per pixel: clamp, pow(1/gamma), 3->64->3 MLP, per-channel tone curve
// All HBM-bound or VALU-bound element-wise work; the dense parts of the variant run on the shared conv / GEMM kernels.
#include "rf_common.h"

namespace rf {

namespace {

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_f(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

__device__ __forceinline__ void atomic_max_float(int* addr, float v) {
    if (v >= 0.f) atomicMax(addr, __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

__device__ __forceinline__ float packed_at(const float* in, int mosaic, size_t b, int ch, int y, int x, int H, int W) {
    return mosaic ? in[(b * 2 * H + 2 * y + (ch >> 1)) * (size_t)(2 * W) + 2 * x + (ch & 1)] : in[((b * 4 + ch) * H + y) * (size_t)W + x];
}

__global__ void tc_init_kernel(int* amax, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) amax[i] = (int)0xff800000u;   // -inf
}

// cin4 [B,4,H,W] = (r, g, b, y_raw) after white balance; lin3 [B,3,H,W] = colour matrix output
__global__ void __launch_bounds__(256) tc_front_kernel(const float* __restrict__ in, int mosaic, const float* __restrict__ wb_gains,
                                                       const float* __restrict__ cm, float* __restrict__ cin4, float* __restrict__ lin3,
                                                       int* __restrict__ amax, int H, int W) {
    const size_t b = blockIdx.y, hw = (size_t)H * W;
    float gn[4], m[12];
#pragma unroll
    for (int i = 0; i < 4; ++i) gn[i] = softplus_f(wb_gains[i]) + 1e-6f;
#pragma unroll
    for (int i = 0; i < 12; ++i) m[i] = cm[i];
    float mx = -INFINITY;
    for (size_t p = blockIdx.x * 256ull + threadIdx.x; p < hw; p += (size_t)gridDim.x * 256) {
        const int y = (int)(p / W), x = (int)(p % W);
        const float r = packed_at(in, mosaic, b, 0, y, x, H, W) * gn[0];
        const float g = 0.5f * (packed_at(in, mosaic, b, 1, y, x, H, W) * gn[1] + packed_at(in, mosaic, b, 2, y, x, H, W) * gn[2]);
        const float bl = packed_at(in, mosaic, b, 3, y, x, H, W) * gn[3];
        float l[3];
#pragma unroll
        for (int o = 0; o < 3; ++o) l[o] = ((r * m[4 * o] + g * m[4 * o + 1]) + bl * m[4 * o + 2]) + m[4 * o + 3];
        const float yr = (l[0] * 0.2126f + l[1] * 0.7152f) + l[2] * 0.0722f;
        float* c4 = cin4 + b * 4 * hw + p;
        c4[0] = r; c4[hw] = g; c4[2 * hw] = bl; c4[3 * hw] = yr;
        float* l3 = lin3 + b * 3 * hw + p;
        l3[0] = l[0]; l3[hw] = l[1]; l3[2 * hw] = l[2];
        mx = fmaxf(mx, yr);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        if (mx > -INFINITY) atomic_max_float(amax + b, mx);
    }
}

// y = y_raw / max(amax, eps) written back into plane 3 of cin4; first pyramid level (LL1, mag1) from 2x2 blocks of y.
// One thread per 2x2 block (H, W even: packed sizes are multiples of 8).
__global__ void __launch_bounds__(256) tc_normalise_kernel(float* __restrict__ cin4, const int* __restrict__ amax, float* __restrict__ ll1,
                                                           float* __restrict__ mag1, int H, int W) {
    const size_t b = blockIdx.y, hw = (size_t)H * W;
    const int H2 = H / 2, W2 = W / 2;
    const float inv = 1.0f / fmaxf(__int_as_float(amax[b]), 1e-6f);
    float* yp = cin4 + (b * 4 + 3) * hw;
    for (size_t q = blockIdx.x * 256ull + threadIdx.x; q < (size_t)H2 * W2; q += (size_t)gridDim.x * 256) {
        const int y2 = (int)(q / W2), x2 = (int)(q % W2);
        float v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const size_t p = (size_t)(2 * y2 + (t >> 1)) * W + 2 * x2 + (t & 1);
            v[t] = yp[p] * inv;
            yp[p] = v[t];
        }
        const float lh = (v[0] - v[1] + v[2] - v[3]) * 0.5f, hl = (v[0] + v[1] - v[2] - v[3]) * 0.5f, hh = (v[0] - v[1] - v[2] + v[3]) * 0.5f;
        ll1[b * (size_t)H2 * W2 + q] = (v[0] + v[1] + v[2] + v[3]) * 0.5f;
        mag1[b * (size_t)H2 * W2 + q] = sqrtf(lh * lh + hl * hl + hh * hh + 1e-8f);
    }
}

// deeper pyramid levels: HaarDWT of the previous LL (reflect padding when a side is odd: HaarDWT.forward, :54-58)
__global__ void __launch_bounds__(256) tc_pyramid_kernel(const float* __restrict__ src, float* __restrict__ ll, float* __restrict__ mag, int hs, int ws) {
    const size_t b = blockIdx.y;
    const int h2 = (hs + 1) / 2, w2 = (ws + 1) / 2;
    const float* s = src + b * (size_t)hs * ws;
    for (size_t q = blockIdx.x * 256ull + threadIdx.x; q < (size_t)h2 * w2; q += (size_t)gridDim.x * 256) {
        const int y2 = (int)(q / w2), x2 = (int)(q % w2);
        float v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            int y = 2 * y2 + (t >> 1), x = 2 * x2 + (t & 1);
            if (y >= hs) y = hs - 2;            // reflect (no edge repeat): index hs -> hs - 2
            if (x >= ws) x = ws - 2;
            v[t] = s[(size_t)(y < 0 ? 0 : y) * ws + (x < 0 ? 0 : x)];
        }
        const float lh = (v[0] - v[1] + v[2] - v[3]) * 0.5f, hl = (v[0] + v[1] - v[2] - v[3]) * 0.5f, hh = (v[0] - v[1] - v[2] + v[3]) * 0.5f;
        ll[b * (size_t)h2 * w2 + q] = (v[0] + v[1] + v[2] + v[3]) * 0.5f;
        mag[b * (size_t)h2 * w2 + q] = sqrtf(lh * lh + hl * hl + hh * hh + 1e-8f);
    }
}

__device__ __forceinline__ float bilerp(const float* __restrict__ src, int hi, int wi, int ho, int wo, int y, int x) {
    const float sy = fmaxf(((float)y + 0.5f) * ((float)hi / (float)ho) - 0.5f, 0.f);
    const float sx = fmaxf(((float)x + 0.5f) * ((float)wi / (float)wo) - 0.5f, 0.f);
    int y0 = (int)sy, x0 = (int)sx;
    if (y0 > hi - 1) y0 = hi - 1;
    if (x0 > wi - 1) x0 = wi - 1;
    const int y1 = y0 + (y0 < hi - 1), x1 = x0 + (x0 < wi - 1);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float top = src[(size_t)y0 * wi + x0] * (1.f - lx) + src[(size_t)y0 * wi + x1] * lx;
    const float bot = src[(size_t)y1 * wi + x0] * (1.f - lx) + src[(size_t)y1 * wi + x1] * lx;
    return top * (1.f - ly) + bot * ly;
}

struct TcGuideSrc {
    const float* y;        // plane 3 of cin4, image stride 4 * H * W
    const float* crcb;     // [B,2,H,W]
    const float* rgb;      // refined rgb [B,3,H,W]
    const float* ll[3];    // LL of pyramid level i (size level_h[i] x level_w[i])
    const float* mag[3];
    int lh[3], lw[3];
    int levels;
};

__global__ void __launch_bounds__(256) tc_guide_level_kernel(TcGuideSrc s, float* __restrict__ guide, int H, int W, int hf, int wf) {
    const size_t b = blockIdx.y, hw = (size_t)H * W, pf = (size_t)hf * wf;
    for (size_t p = blockIdx.x * 256ull + threadIdx.x; p < pf; p += (size_t)gridDim.x * 256) {
        const int y = (int)(p / wf), x = (int)(p % wf);
        float* o = guide + b * 7 * pf + p;
        o[0] = bilerp(s.y + b * 4 * hw, H, W, hf, wf, y, x);
        o[pf] = bilerp(s.crcb + b * 2 * hw, H, W, hf, wf, y, x);
        o[2 * pf] = bilerp(s.crcb + (b * 2 + 1) * hw, H, W, hf, wf, y, x);
        o[3 * pf] = bilerp(s.rgb + b * 3 * hw, H, W, hf, wf, y, x);
        o[4 * pf] = bilerp(s.rgb + (b * 3 + 1) * hw, H, W, hf, wf, y, x);
        const int L = s.levels - 1;
        o[5 * pf] = bilerp(s.ll[L] + b * (size_t)s.lh[L] * s.lw[L], s.lh[L], s.lw[L], hf, wf, y, x);
        float hsum = 0.f;
        for (int i = 0; i < s.levels; ++i) hsum += bilerp(s.mag[i] + b * (size_t)s.lh[i] * s.lw[i], s.lh[i], s.lw[i], hf, wf, y, x);
        o[6 * pf] = s.levels > 1 ? hsum / (float)s.levels : hsum;
    }
}

// Spatial gate.  PX = pixels per thread along x (4: float4 feature traffic, needs w % 4 == 0; 1: any shape).
struct TcSpatialArgs {
    const float* feat; float* xs; const float* guide;         // [B,C,P], [B,C,P], [B,7,P]
    const float* w_col; const float* b_col;                   // [C,5,3,3], [C]
    const float* w_low; const float* b_low; const float* w_high; const float* b_high;   // [C,1,3,3], [C]
    int B, C, h, w;
};

template <int PX>
__global__ void __launch_bounds__(256) tc_spatial_kernel(TcSpatialArgs a) {
    constexpr int CG = 32;
    const size_t b = blockIdx.z;
    const int h = a.h, w = a.w, P = h * w, C = a.C;
    const int c_lo = blockIdx.y * CG, c_hi = (c_lo + CG < C) ? c_lo + CG : C;
    const int p = (blockIdx.x * 256 + threadIdx.x) * PX;
    const bool live = p < P;
    const int y = live ? p / w : 0, x = live ? p - (p / w) * w : 0;
    float nb[7][3][PX + 2];
    const float* gb = a.guide + b * 7 * (size_t)P;
#pragma unroll
    for (int pl = 0; pl < 7; ++pl)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < PX + 2; ++dx) {
                const int yy = y + dy - 1, xx = x + dx - 1;
                const bool ok = live && yy >= 0 && yy < h && xx >= 0 && xx < w;
                nb[pl][dy][dx] = ok ? gb[(size_t)pl * P + (size_t)yy * w + xx] : 0.f;
            }
    const float* fb = a.feat + b * (size_t)C * P + (live ? p : 0);
    float* xb = a.xs + b * (size_t)C * P + p;
    for (int c = c_lo; c < c_hi; ++c) {
        const float* wc = a.w_col + c * 45;
        const float* wl = a.w_low + c * 9;
        const float* wh = a.w_high + c * 9;
        float sc[PX], sl[PX], sh[PX];
#pragma unroll
        for (int q = 0; q < PX; ++q) { sc[q] = a.b_col[c]; sl[q] = a.b_low[c]; sh[q] = a.b_high[c]; }
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int t = dy * 3 + dx;
                const float k0 = wc[t], k1 = wc[9 + t], k2 = wc[18 + t], k3 = wc[27 + t], k4 = wc[36 + t], kl = wl[t], kh = wh[t];
#pragma unroll
                for (int q = 0; q < PX; ++q) {
                    sc[q] = fmaf(k4, nb[4][dy][q + dx], fmaf(k3, nb[3][dy][q + dx], fmaf(k2, nb[2][dy][q + dx],
                            fmaf(k1, nb[1][dy][q + dx], fmaf(k0, nb[0][dy][q + dx], sc[q])))));
                    sl[q] = fmaf(kl, nb[5][dy][q + dx], sl[q]);
                    sh[q] = fmaf(kh, nb[6][dy][q + dx], sh[q]);
                }
            }
        if (live) {
            float fv[PX], v[PX];
            if constexpr (PX == 4) {
                const float4 f4 = *reinterpret_cast<const float4*>(fb + (size_t)c * P);
                fv[0] = f4.x; fv[1] = f4.y; fv[2] = f4.z; fv[3] = f4.w;
            } else {
                fv[0] = fb[(size_t)c * P];
            }
#pragma unroll
            for (int q = 0; q < PX; ++q) v[q] = fv[q] * (1.0f + sigmoid_f(sc[q]) + tanh_f(sigmoid_f(sl[q]) + tanh_f(sh[q])));
            if constexpr (PX == 4) *reinterpret_cast<float4*>(xb + (size_t)c * P) = make_float4(v[0], v[1], v[2], v[3]);
            else xb[(size_t)c * P] = v[0];
        }
    }
}

// x2 = xs + 0.2 tanh(r); partial[(b * nblk + blk) * C + c] = sum over the block's pixels of x2 (fixed-order: no atomics)
template <int PX>
__global__ void __launch_bounds__(256) tc_residual_kernel(const float* __restrict__ xs, const float* __restrict__ r, float* __restrict__ x2,
                                                          float* __restrict__ partial, int nblk, int C, int P) {
    const int blk = blockIdx.x;
    const size_t b = blockIdx.y;
    const int p = (blk * 256 + threadIdx.x) * PX;
    const bool live = p < P;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ float red[512][4];
    const size_t base = b * (size_t)C * P + (live ? p : 0);
    for (int c = 0; c < C; ++c) {
        float s = 0.f;
        if (live) {
            if constexpr (PX == 4) {
                const float4 a = *reinterpret_cast<const float4*>(xs + base + (size_t)c * P);
                const float4 t = *reinterpret_cast<const float4*>(r + base + (size_t)c * P);
                const float4 v = make_float4(a.x + 0.2f * tanh_f(t.x), a.y + 0.2f * tanh_f(t.y), a.z + 0.2f * tanh_f(t.z), a.w + 0.2f * tanh_f(t.w));
                *reinterpret_cast<float4*>(x2 + base + (size_t)c * P) = v;
                s = (v.x + v.y) + (v.z + v.w);
            } else {
                const float v = xs[base + (size_t)c * P] + 0.2f * tanh_f(r[base + (size_t)c * P]);
                x2[base + (size_t)c * P] = v;
                s = v;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) red[c][wave] = s;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256)
        partial[(b * nblk + blk) * C + c] = (red[c][0] + red[c][1]) + (red[c][2] + red[c][3]);
}

// refined = lin + d (demosaic_refine residual, :131)
__global__ void __launch_bounds__(256) tc_add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = a[i] + b[i];
}

// CameraAwareColorCorrection in place on [B,3,P]
__global__ void __launch_bounds__(256) tc_color_head_kernel(float* __restrict__ x, const float* __restrict__ gamma_param,
                                                            const float* __restrict__ w0, const float* __restrict__ b0,   // [64,3], [64]
                                                            const float* __restrict__ w2, const float* __restrict__ b2,   // [3,64], [3]
                                                            const float* __restrict__ t0w, const float* __restrict__ t0b, // [32], [32]
                                                            const float* __restrict__ t2w, const float* __restrict__ t2b, // [32], [1]
                                                            size_t P) {
    __shared__ float s_w0[192], s_b0[64], s_w2[192], s_t0w[32], s_t0b[32], s_t2w[32];
    for (int i = threadIdx.x; i < 192; i += 256) { s_w0[i] = w0[i]; s_w2[i] = w2[i]; }
    for (int i = threadIdx.x; i < 64; i += 256) s_b0[i] = b0[i];
    for (int i = threadIdx.x; i < 32; i += 256) { s_t0w[i] = t0w[i]; s_t0b[i] = t0b[i]; s_t2w[i] = t2w[i]; }
    __syncthreads();
    const float inv_gamma = 1.0f / (softplus_f(*gamma_param) + 1e-6f);
    const float bb[3] = {b2[0], b2[1], b2[2]}, tb = t2b[0];
    const size_t b = blockIdx.y;
    float* xb = x + b * 3 * P;
    for (size_t p = blockIdx.x * 256ull + threadIdx.x; p < P; p += (size_t)gridDim.x * 256) {
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = powf(fminf(fmaxf(xb[c * P + p], 0.f), 1.f), inv_gamma);
        float o[3] = {bb[0], bb[1], bb[2]};
        for (int m = 0; m < 64; ++m) {
            const float hid = fmaxf(fmaf(s_w0[3 * m + 2], v[2], fmaf(s_w0[3 * m + 1], v[1], fmaf(s_w0[3 * m], v[0], s_b0[m]))), 0.f);
#pragma unroll
            for (int c = 0; c < 3; ++c) o[c] = fmaf(s_w2[64 * c + m], hid, o[c]);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float t = tb;
            for (int m = 0; m < 32; ++m) t = fmaf(s_t2w[m], fmaxf(fmaf(s_t0w[m], o[c], s_t0b[m]), 0.f), t);
            const float mod = 1.0f / (1.0f + expf(-t));
            xb[c * P + p] = fminf(fmaxf(o[c] * (0.8f + 0.4f * mod), 0.f), 1.f);
        }
    }
}

int grid_for(size_t n, int cap = 4096) {
    int g = (int)((n + 255) / 256);
    if (g > cap) g = cap;
    return g < 1 ? 1 : g;
}

}  // namespace

// scratch of the front end (floats): cin4 4hw | lin3 3hw | t16 16hw | crcb 2hw | t32 32hw | d3 3hw | rgb 3hw | pyramid | amax
size_t tc_front_scratch_floats(int B, int H, int W, int levels) {
    const size_t hw = (size_t)H * W;
    size_t pyr = 0;
    int h = H, w = W;
    for (int i = 0; i < levels; ++i) { h = (h + 1) / 2; w = (w + 1) / 2; pyr += 2 * (size_t)h * w; }
    return (size_t)B * ((4 + 3 + 16 + 2 + 32 + 3 + 3) * hw + pyr) + align_up((size_t)B, 64) + 64 * 16;
}

struct TcFrontBufs { float *cin4, *lin3, *t16, *crcb, *t32, *d3, *rgb, *ll[3], *mag[3]; int lh[3], lw[3]; int* amax; };

static TcFrontBufs tc_front_layout(float* s, int B, int H, int W, int levels) {
    const size_t hw = (size_t)H * W;
    TcFrontBufs f{};
    size_t off = 0;
    auto take = [&](size_t n) { float* p = s + off; off += align_up(n, 64); return p; };
    f.cin4 = take(B * 4 * hw); f.lin3 = take(B * 3 * hw); f.t16 = take(B * 16 * hw); f.crcb = take(B * 2 * hw);
    f.t32 = take(B * 32 * hw); f.d3 = take(B * 3 * hw); f.rgb = take(B * 3 * hw);
    int h = H, w = W;
    for (int i = 0; i < levels; ++i) {
        h = (h + 1) / 2; w = (w + 1) / 2;
        f.lh[i] = h; f.lw[i] = w;
        f.ll[i] = take((size_t)B * h * w); f.mag[i] = take((size_t)B * h * w);
    }
    f.amax = reinterpret_cast<int*>(take(B));
    return f;
}

// EnhancedBayerProcessor + pyramid.  wp_* = packed 3x3 weights of the four small convolutions.
int launch_tc_front(const float* in, int mosaic, const float* wb_gains, const float* color_matrix, const float* wp_c0, const float* b_c0,
                    const float* wp_c2, const float* b_c2, const float* wp_d0, const float* b_d0, const float* wp_d2, const float* b_d2,
                    float* scratch, int B, int H, int W, int levels, hipStream_t st) {
    RF_CHECK_ARG(levels >= 1 && levels <= 3 && H % 2 == 0 && W % 2 == 0 && B <= 65535, "truecolor front: levels=%d H=%d W=%d unsupported", levels, H, W);
    const TcFrontBufs f = tc_front_layout(scratch, B, H, W, levels);
    const size_t hw = (size_t)H * W;
    ProfScope prof(st, "tc_front(EnhancedBayerProcessor)", 2.0 * B * hw * 9.0 * (4 * 16 + 16 * 2 + 3 * 32 + 32 * 3), 4.0 * B * hw * 80.0);
    tc_init_kernel<<<cdiv(B, 256), 256, 0, st>>>(f.amax, B);
    const int cap = 2048 / B > 32 ? 2048 / B : 32;
    tc_front_kernel<<<dim3((unsigned)grid_for(hw, cap), (unsigned)B), 256, 0, st>>>(in, mosaic, wb_gains, color_matrix, f.cin4, f.lin3, f.amax, H, W);
    tc_normalise_kernel<<<dim3((unsigned)grid_for(hw / 4, 1024), (unsigned)B), 256, 0, st>>>(f.cin4, f.amax, f.ll[0], f.mag[0], H, W);
    for (int i = 1; i < levels; ++i)
        tc_pyramid_kernel<<<dim3((unsigned)grid_for((size_t)f.lh[i] * f.lw[i], 1024), (unsigned)B), 256, 0, st>>>(f.ll[i - 1], f.ll[i], f.mag[i], f.lh[i - 1], f.lw[i - 1]);
    if (int rc = check_launch("tc_front")) return rc;
    auto conv = [&](const float* x, float* out, const float* wp, const float* bias, int cin, int cout, int act) {
        Conv3x3Args a{};
        a.x = x; a.x_bstride = (int64_t)cin * hw; a.wp = wp; a.bias = bias; a.out = out; a.out_bstride = (int64_t)cout * hw;
        a.B = B; a.Cin = cin; a.Cout = cout; a.h = H; a.w = W; a.act = act;
        return launch_conv3x3(a, st);
    };
    if (int rc = conv(f.cin4, f.t16, wp_c0, b_c0, 4, 16, 2)) return rc;       // chroma_extractor: conv, ReLU, conv, Tanh
    if (int rc = conv(f.t16, f.crcb, wp_c2, b_c2, 16, 2, 4)) return rc;
    if (int rc = conv(f.lin3, f.t32, wp_d0, b_d0, 3, 32, 3)) return rc;       // demosaic_refine: conv, GELU, conv (+ residual)
    if (int rc = conv(f.t32, f.d3, wp_d2, b_d2, 32, 3, 0)) return rc;
    tc_add_kernel<<<grid_for(B * 3 * hw), 256, 0, st>>>(f.lin3, f.d3, f.rgb, B * 3 * hw);
    return check_launch("tc_front");
}

int launch_tc_guide_level(const float* scratch, float* guide, int B, int H, int W, int levels, int hf, int wf, hipStream_t st) {
    const TcFrontBufs f = tc_front_layout(const_cast<float*>(scratch), B, H, W, levels);
    TcGuideSrc s{};
    s.y = f.cin4 + 3 * (size_t)H * W; s.crcb = f.crcb; s.rgb = f.rgb; s.levels = levels;
    for (int i = 0; i < levels; ++i) { s.ll[i] = f.ll[i]; s.mag[i] = f.mag[i]; s.lh[i] = f.lh[i]; s.lw[i] = f.lw[i]; }
    ProfScope prof(st, "tc_guide_level_kernel", 0.0, 4.0 * B * hf * wf * 14);
    tc_guide_level_kernel<<<dim3((unsigned)grid_for((size_t)hf * wf, 1024), (unsigned)B), 256, 0, st>>>(s, guide, H, W, hf, wf);
    return check_launch("tc_guide_level");
}

// the bayer-processor outputs themselves, for operator-level parity tests: y [B,1,H,W], crcb [B,2,H,W], rgb [B,3,H,W]
int tc_front_outputs(const float* scratch, const float** y, const float** crcb, const float** rgb, int B, int H, int W, int levels) {
    const TcFrontBufs f = tc_front_layout(const_cast<float*>(scratch), B, H, W, levels);
    *y = f.cin4 + 3 * (size_t)H * W; *crcb = f.crcb; *rgb = f.rgb;
    return RF_OK;
}

int tc_nblk(int h, int w) { return (w % 4 == 0) ? cdiv(h * w, 1024) : cdiv(h * w, 256); }

int launch_tc_spatial(const float* feat, float* xs, const float* guide, const float* w_col, const float* b_col, const float* w_low,
                      const float* b_low, const float* w_high, const float* b_high, int B, int C, int h, int w, hipStream_t st) {
    RF_CHECK_ARG(C <= 512 && B <= 65535, "truecolor flca: C=%d > 512 not supported", C);
    TcSpatialArgs a{feat, xs, guide, w_col, b_col, w_low, b_low, w_high, b_high, B, C, h, w};
    const double el = (double)B * C * h * w;
    ProfScope prof(st, "tc_spatial_kernel", 130.0 * el, 8.0 * el);
    const bool vec = (w % 4 == 0) && aligned16(feat) && aligned16(xs);
    if (vec) tc_spatial_kernel<4><<<dim3((unsigned)cdiv(h * w, 1024), (unsigned)cdiv(C, 32), (unsigned)B), 256, 0, st>>>(a);
    else tc_spatial_kernel<1><<<dim3((unsigned)cdiv(h * w, 256), (unsigned)cdiv(C, 32), (unsigned)B), 256, 0, st>>>(a);
    return check_launch("tc_spatial");
}

int launch_tc_residual(const float* xs, const float* r, float* x2, float* partial, int B, int C, int h, int w, hipStream_t st) {
    RF_CHECK_ARG(C <= 512 && B <= 65535, "truecolor flca: C=%d > 512 not supported", C);
    const int P = h * w, nblk = tc_nblk(h, w);
    ProfScope prof(st, "tc_residual_kernel", 8.0 * B * C * P, 12.0 * B * C * P);
    const bool vec = (w % 4 == 0) && aligned16(xs) && aligned16(r) && aligned16(x2);
    RF_CHECK_ARG(vec || w % 4 != 0, "truecolor flca: buffers must be 16-byte aligned");
    if (vec) tc_residual_kernel<4><<<dim3((unsigned)nblk, (unsigned)B), 256, 0, st>>>(xs, r, x2, partial, nblk, C, P);
    else tc_residual_kernel<1><<<dim3((unsigned)nblk, (unsigned)B), 256, 0, st>>>(xs, r, x2, partial, nblk, C, P);
    return check_launch("tc_residual");
}

int launch_tc_color_head(float* x, const float* const* prm /* gamma, ct.0.w, ct.0.b, ct.2.w, ct.2.b, tone.0.w, tone.0.b, tone.2.w, tone.2.b */,
                         int B, size_t P, hipStream_t st) {
    ProfScope prof(st, "tc_color_head_kernel", 2.0 * B * P * (3 * 64 * 2 + 3 * 64), 24.0 * B * P);
    const int cap = 4096 / B > 64 ? 4096 / B : 64;
    tc_color_head_kernel<<<dim3((unsigned)grid_for(P, cap), (unsigned)B), 256, 0, st>>>(x, prm[0], prm[1], prm[2], prm[3], prm[4], prm[5], prm[6], prm[7], prm[8], P);
    return check_launch("tc_color_head");
}

}  // namespace rf
