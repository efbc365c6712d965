// FLCA branch (a15, FrequencyawareLumaChromaAttentionRAWFormer.py:79-162) and its guidance.
//   guidance_base : packed RGGB (or the mosaic itself) -> y, cr, cb at HxW, Haar LL and
//                   high-band magnitude at H/2 x W/2 (a14 on the 1-channel luma)
//   guidance_level: bilinear resize (align_corners=False) of those planes to a stage size
//   flca_spatial  : feat * (1 + a*sigmoid(conv3(y_low)) + b*tanh(conv3(y_high)) + g*sigmoid(conv3(cr,cb)))
//                   + per-block channel sums for the squeeze-excite pooling
//   flca_se_fold  : SE MLP, then folds the per-image channel gate into the following
//                   channel_reduce 1x1:  W_cr [x*ch ; trans] = [W_a diag(ch) | W_b] [x ; trans]
#include <type_traits>
#include "rf_common.h"

namespace rf {

size_t guidance_scratch_floats(int B, int H, int W) {
    const size_t hw = (size_t)H * W;
    return (size_t)B * (3 * hw + 2 * (hw / 4)) + align_up((size_t)B, 64);
}

struct GuideLayout {
    float *y, *cr, *cb, *ll, *mag;
    int* amax;
};
__host__ __device__ static inline GuideLayout guide_layout(float* s, int B, int H, int W) {
    const size_t hw = (size_t)H * W;
    GuideLayout g;
    g.y = s;
    g.cr = g.y + B * hw;
    g.cb = g.cr + B * hw;
    g.ll = g.cb + B * hw;
    g.mag = g.ll + B * (hw / 4);
    g.amax = reinterpret_cast<int*>(g.mag + B * (hw / 4));
    return g;
}

__device__ __forceinline__ void atomic_max_float(int* addr, float v) {
    if (v >= 0.f) atomicMax(addr, __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

__device__ __forceinline__ float packed_at(const float* in, int mosaic, int clamp_in, size_t b, int ch, int y, int x, int H, int W) {
    float v = mosaic ? in[(b * 2 * H + 2 * y + (ch >> 1)) * (size_t)(2 * W) + 2 * x + (ch & 1)]
                     : in[((b * 4 + ch) * H + y) * (size_t)W + x];
    return clamp_in ? fminf(fmaxf(v, 0.f), 1.f) : v;
}

__global__ void guide_init_kernel(int* amax, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) amax[i] = (int)0xff800000u;   // -inf
}

// y_raw = 0.299 R + 0.587 * 0.5 (G1 + G2) + 0.114 B and its per-image maximum
__global__ void __launch_bounds__(256) guide_luma_kernel(const float* __restrict__ in, int mosaic, int clamp_in,
                                                         float* __restrict__ yraw, int* __restrict__ amax, int H, int W) {
    const size_t b = blockIdx.y;
    const size_t hw = (size_t)H * W;
    float m = -INFINITY;
    for (size_t p = blockIdx.x * 256ull + threadIdx.x; p < hw; p += (size_t)gridDim.x * 256) {
        const int y = (int)(p / W), x = (int)(p % W);
        const float r = packed_at(in, mosaic, clamp_in, b, 0, y, x, H, W);
        const float g = 0.5f * (packed_at(in, mosaic, clamp_in, b, 1, y, x, H, W) + packed_at(in, mosaic, clamp_in, b, 2, y, x, H, W));
        const float bl = packed_at(in, mosaic, clamp_in, b, 3, y, x, H, W);
        const float v = (0.299f * r + 0.587f * g) + 0.114f * bl;
        yraw[b * hw + p] = v;
        m = fmaxf(m, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {   // one atomic per workgroup: all of an image's atomics hit one address
        m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        if (m > -INFINITY) atomic_max_float(amax + b, m);
    }
}

// normalise luma, chroma differences, Haar LL / high-band magnitude (one thread per 2x2 block)
__global__ void __launch_bounds__(256) guide_haar_kernel(const float* __restrict__ in, int mosaic, int clamp_in, float* __restrict__ scratch,
                                                         int B, int H, int W) {
    const GuideLayout g = guide_layout(scratch, B, H, W);
    const size_t b = blockIdx.y;
    const int H2 = H / 2, W2 = W / 2;
    const size_t hw = (size_t)H * W, hw4 = (size_t)H2 * W2;
    const float inv = 1.0f / fmaxf(__int_as_float(g.amax[b]), 1e-6f);
    for (size_t q = blockIdx.x * 256ull + threadIdx.x; q < hw4; q += (size_t)gridDim.x * 256) {
        const int y2 = (int)(q / W2), x2 = (int)(q % W2);
        float yv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int y = 2 * y2 + (t >> 1), x = 2 * x2 + (t & 1);
            const size_t p = (size_t)y * W + x;
            const float yn = g.y[b * hw + p] * inv;
            g.y[b * hw + p] = yn;
            g.cr[b * hw + p] = packed_at(in, mosaic, clamp_in, b, 0, y, x, H, W) - yn;
            g.cb[b * hw + p] = packed_at(in, mosaic, clamp_in, b, 3, y, x, H, W) - yn;
            yv[t] = yn;
        }
        const float ll = (yv[0] + yv[1] + yv[2] + yv[3]) * 0.5f;
        const float lh = (yv[0] - yv[1] + yv[2] - yv[3]) * 0.5f;
        const float hl = (yv[0] + yv[1] - yv[2] - yv[3]) * 0.5f;
        const float hh = (yv[0] - yv[1] - yv[2] + yv[3]) * 0.5f;
        g.ll[b * hw4 + q] = ll;
        g.mag[b * hw4 + q] = sqrtf(lh * lh + hl * hl + hh * hh + 1e-8f);
    }
}

int launch_guidance_base(const float* in, int mosaic, int clamp_in, float* scratch, int B, int H, int W, hipStream_t st,
                         void (*allreduce)(void*, float*, size_t, int, void*), void* allreduce_user) {
    RF_CHECK_ARG(H % 2 == 0 && W % 2 == 0 && B <= 65535, "guidance: H=%d W=%d must be even", H, W);
    const GuideLayout g = guide_layout(scratch, B, H, W);
    ProfScope prof(st, "guidance_base(3 kernels)", 0.0, 4.0 * B * H * W * (4 + 1 + 1 + 4 + 3.5));
    guide_init_kernel<<<cdiv(B, 256), 256, 0, st>>>(g.amax, B);
    const size_t hw = (size_t)H * W;
    int gx = (int)((hw + 255) / 256);
    const int cap = 2048 / B > 32 ? 2048 / B : 32;   // grid-stride: few workgroups (= few atomics) per image
    if (gx > cap) gx = cap;
    guide_luma_kernel<<<dim3((unsigned)gx, (unsigned)B), 256, 0, st>>>(in, mosaic, clamp_in, g.y, g.amax, H, W);
    // spatial shard: the luma normaliser is the maximum over the whole frame = the max over the ranks' windows (the stored bit
    // patterns are the floats themselves)
    if (allreduce) allreduce(allreduce_user, reinterpret_cast<float*>(g.amax), (size_t)B, 1, (void*)st);
    int gx2 = (int)((hw / 4 + 255) / 256);
    if (gx2 > 1024) gx2 = 1024;
    guide_haar_kernel<<<dim3((unsigned)gx2, (unsigned)B), 256, 0, st>>>(in, mosaic, clamp_in, scratch, B, H, W);
    return check_launch("guidance_base");
}

// F.interpolate(mode='bilinear', align_corners=False)
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

__global__ void __launch_bounds__(256) guide_level_kernel(const float* __restrict__ scratch, float* __restrict__ guide,
                                                          int B, int H, int W, int hf, int wf) {
    const GuideLayout g = guide_layout(const_cast<float*>(scratch), B, H, W);
    const size_t b = blockIdx.y;
    const size_t pf = (size_t)hf * wf, hw = (size_t)H * W;
    for (size_t p = blockIdx.x * 256ull + threadIdx.x; p < pf; p += (size_t)gridDim.x * 256) {
        const int y = (int)(p / wf), x = (int)(p % wf);
        float* o = guide + b * 4 * pf + p;
        o[0] = bilerp(g.ll + b * (hw / 4), H / 2, W / 2, hf, wf, y, x);
        o[pf] = bilerp(g.mag + b * (hw / 4), H / 2, W / 2, hf, wf, y, x);
        o[2 * pf] = bilerp(g.cr + b * hw, H, W, hf, wf, y, x);
        o[3 * pf] = bilerp(g.cb + b * hw, H, W, hf, wf, y, x);
    }
}

int launch_guidance_level(const float* scratch, float* guide, int B, int H, int W, int hf, int wf, hipStream_t st) {
    int gx = cdiv(hf * wf, 256);
    if (gx > 1024) gx = 1024;
    ProfScope prof(st, "guide_level_kernel", 0.0, 4.0 * B * hf * wf * 8);
    guide_level_kernel<<<dim3((unsigned)gx, (unsigned)B), 256, 0, st>>>(scratch, guide, B, H, W, hf, wf);
    return check_launch("guidance_level");
}

// ------------------------------------------------------------------------------------------
// Spatial modulation.  Vector path (w % 4 == 0): a lane owns 4 consecutive pixels of a row and
// keeps their 3x6 guidance neighbourhoods of all four planes in registers (72 floats); the channel
// loop prefetches the next channel's feature float4 while the current one is in the VALU;
// sigmoid / tanh use v_exp_f32 + v_rcp_f32 (about 1 ulp each, far below the parity budget).
// The 36 weights of a channel are wave-uniform (scalar loads).  Per-block channel sums feed the
// squeeze-excite pooling without atomics.
static constexpr int kFlcaCG = 32;   // channels per workgroup: (pixel block, channel group, image) grid keeps small levels parallel
int flca_nblk(int h, int w) { return (w % 4 == 0) ? cdiv(h * w, 1024) : (w % 2 == 0) ? cdiv(h * w, 512) : cdiv(h * w, 256); }

__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

template <int PX>   // pixels of a row per lane: 4 (w % 4 == 0) or 2 (w % 4 == 2, e.g. level 3 of a 2848 x 4256 frame: w = 266)
__global__ void __launch_bounds__(256) flca_spatial_vec_kernel(FlcaSpatialArgs a, const float* __restrict__ w_low,
                                                               const float* __restrict__ w_high, const float* __restrict__ w_chr,
                                                               const float* __restrict__ feat, float* __restrict__ xs, int cg) {
    const int blk = blockIdx.x;
    const size_t b = blockIdx.z;
    const int h = a.h, w = a.w, P = h * w, C = a.C;
    const int c_lo = blockIdx.y * cg, c_hi = (c_lo + cg < C) ? c_lo + cg : C;   // this workgroup's channels (cg <= kFlcaCG)
    const int p = (blk * 256 + threadIdx.x) * PX;
    const bool live = p < P;
    const int y = live ? p / w : 0, x = live ? p - (p / w) * w : 0;
    const bool counted = y >= a.ylo && y < (a.yhi > 0 ? a.yhi : h);      // rows of the squeeze-excite pooling
    float nb[4][3][PX + 2];
    const float* gb = a.guide + b * 4 * (size_t)P;
#pragma unroll
    for (int pl = 0; pl < 4; ++pl)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = y + dy - 1;
            const bool rok = live && yy >= 0 && yy < h;
            const float* row = gb + (size_t)pl * P + (size_t)(rok ? yy : 0) * w;
            float m[PX];
            if constexpr (PX == 4) {
                const float4 t = *reinterpret_cast<const float4*>(row + x);
                m[0] = t.x; m[1] = t.y; m[2] = t.z; m[3] = t.w;
            } else {
                const float2 t = *reinterpret_cast<const float2*>(row + x);
                m[0] = t.x; m[1] = t.y;
            }
            const float l = row[x > 0 ? x - 1 : 0], r = row[x + PX < w ? x + PX : 0];
            nb[pl][dy][0] = (rok && x > 0) ? l : 0.f;
#pragma unroll
            for (int q = 0; q < PX; ++q) nb[pl][dy][1 + q] = rok ? m[q] : 0.f;
            nb[pl][dy][PX + 1] = (rok && x + PX < w) ? r : 0.f;
        }
    const float al = *a.alpha, be = *a.beta, ga = *a.gamma;
    __shared__ float red[kFlcaCG][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* fb = feat + b * (size_t)C * P + (live ? p : 0);
    float* xb = xs + b * (size_t)C * P + p;
    typedef typename std::conditional<PX == 4, float4, float2>::type vecT;
    vecT fcur = *reinterpret_cast<const vecT*>(fb + (size_t)c_lo * P);
    for (int c = c_lo; c < c_hi; ++c) {
        const vecT fnext = *reinterpret_cast<const vecT*>(fb + (size_t)(c + 1 < c_hi ? c + 1 : c) * P);
        const float* wl = w_low + c * 9;
        const float* wh = w_high + c * 9;
        const float* wc = w_chr + c * 18;
        float sl[PX], sh[PX], sc[PX];
#pragma unroll
        for (int q = 0; q < PX; ++q) { sl[q] = 0.f; sh[q] = 0.f; sc[q] = 0.f; }
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const float k0 = wl[dy * 3 + dx], k1 = wh[dy * 3 + dx], k2 = wc[dy * 3 + dx], k3 = wc[9 + dy * 3 + dx];
#pragma unroll
                for (int q = 0; q < PX; ++q) {
                    sl[q] = fmaf(k0, nb[0][dy][q + dx], sl[q]);
                    sh[q] = fmaf(k1, nb[1][dy][q + dx], sh[q]);
                    sc[q] = fmaf(k3, nb[3][dy][q + dx], fmaf(k2, nb[2][dy][q + dx], sc[q]));
                }
            }
        float v[PX], fv[PX];
        if constexpr (PX == 4) { fv[0] = fcur.x; fv[1] = fcur.y; fv[2] = fcur.z; fv[3] = fcur.w; }
        else { fv[0] = fcur.x; fv[1] = fcur.y; }
#pragma unroll
        for (int q = 0; q < PX; ++q)
            v[q] = fv[q] * (1.0f + al * fast_sigmoid(sl[q]) + be * fast_tanh(sh[q]) + ga * fast_sigmoid(sc[q]));
        float s = 0.f;
        if (live) {
            if constexpr (PX == 4) {
                *reinterpret_cast<float4*>(xb + (size_t)c * P) = make_float4(v[0], v[1], v[2], v[3]);
                if (counted) s = (v[0] + v[1]) + (v[2] + v[3]);
            } else {
                *reinterpret_cast<float2*>(xb + (size_t)c * P) = make_float2(v[0], v[1]);
                if (counted) s = v[0] + v[1];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) red[c - c_lo][wave] = s;
        fcur = fnext;
    }
    __syncthreads();
    for (int c = c_lo + threadIdx.x; c < c_hi; c += 256)
        a.partial[(b * a.nblk + blk) * C + c] = (red[c - c_lo][0] + red[c - c_lo][1]) + (red[c - c_lo][2] + red[c - c_lo][3]);
}

__global__ void __launch_bounds__(256) flca_spatial_kernel(FlcaSpatialArgs a) {
    const int blk = blockIdx.x;
    const size_t b = blockIdx.y;
    const int h = a.h, w = a.w, P = h * w, C = a.C;
    const int p = blk * 256 + threadIdx.x;
    const bool live = p < P;
    const int y = live ? p / w : 0, x = live ? p % w : 0;
    const bool counted = y >= a.ylo && y < (a.yhi > 0 ? a.yhi : h);
    // 3x3 neighbourhoods of the four guidance planes, zero padded
    float nb[4][9];
    const float* gb = a.guide + b * 4 * (size_t)P;
#pragma unroll
    for (int pl = 0; pl < 4; ++pl)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int yy = y + dy - 1, xx = x + dx - 1;
                nb[pl][dy * 3 + dx] = (live && yy >= 0 && yy < h && xx >= 0 && xx < w) ? gb[(size_t)pl * P + (size_t)yy * w + xx] : 0.f;
            }
    const float al = *a.alpha, be = *a.beta, ga = *a.gamma;
    __shared__ float red[512][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* fb = a.feat + b * (size_t)C * P;
    float* xb = a.xs + b * (size_t)C * P;
    for (int c = 0; c < C; ++c) {
        const float* wl = a.w_low + c * 9;
        const float* wh = a.w_high + c * 9;
        const float* wc = a.w_chr + c * 18;
        float sl = 0.f, sh = 0.f, sc = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            sl = fmaf(wl[t], nb[0][t], sl);
            sh = fmaf(wh[t], nb[1][t], sh);
            sc = fmaf(wc[9 + t], nb[3][t], fmaf(wc[t], nb[2][t], sc));
        }
        float v = 0.f;
        if (live) {
            v = fb[(size_t)c * P + p] * (1.0f + al * fast_sigmoid(sl) + be * fast_tanh(sh) + ga * fast_sigmoid(sc));
            xb[(size_t)c * P + p] = v;
            if (!counted) v = 0.f;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[c][wave] = v;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256)
        a.partial[(b * a.nblk + blk) * C + c] = (red[c][0] + red[c][1]) + (red[c][2] + red[c][3]);
}

int launch_flca_spatial(const FlcaSpatialArgs& a, hipStream_t st) {
    RF_CHECK_ARG(a.C <= 512 && a.B <= 65535, "flca: C=%d > 512 not supported", a.C);
    RF_CHECK_ARG(a.nblk == flca_nblk(a.h, a.w), "flca: partial-sum block count mismatch");
    const double el = (double)a.B * a.C * a.h * a.w;
    const bool vec = (a.w % 2 == 0) && aligned16(a.feat) && aligned16(a.xs) && aligned16(a.guide) && ((size_t)a.h * a.w) % 2 == 0;
    RF_CHECK_ARG(vec || a.w % 2 != 0, "flca: feature / guidance buffers must be 16-byte aligned");
    ProfScope prof(st, vec ? (a.w % 4 == 0 ? "flca_spatial_vec_kernel" : "flca_spatial_vec_kernel<2>") : "flca_spatial_kernel", 80.0 * el, 8.0 * el);
    // channels per workgroup: 32, or 8 when that leaves most of the chip idle (one frame: a lane then walks 8 channels instead
    // of 32 -- the per-channel load -> gate -> store chain is what a small launch waits on)
    const int cg = ((long)a.nblk * cdiv(a.C, kFlcaCG) * a.B < 512) ? 8 : kFlcaCG;
    if (vec && a.w % 4 == 0)
        flca_spatial_vec_kernel<4><<<dim3((unsigned)a.nblk, (unsigned)cdiv(a.C, cg), (unsigned)a.B), 256, 0, st>>>(a, a.w_low, a.w_high, a.w_chr, a.feat, a.xs, cg);
    else if (vec)
        flca_spatial_vec_kernel<2><<<dim3((unsigned)a.nblk, (unsigned)cdiv(a.C, cg), (unsigned)a.B), 256, 0, st>>>(a, a.w_low, a.w_high, a.w_chr, a.feat, a.xs, cg);
    else
        flca_spatial_kernel<<<dim3((unsigned)a.nblk, (unsigned)a.B), 256, 0, st>>>(a);
    return check_launch("flca_spatial");
}

// squeeze-excite gate of one image: pooled mean (fixed-order reduction of the per-block sums,
// all 256 threads busy: thread = (block slice, channel)), 2-layer MLP, sigmoid
__global__ void __launch_bounds__(256) flca_se_kernel(const float* __restrict__ partial, int nblk, int P,
                                                      const float* __restrict__ se1_w, const float* __restrict__ se1_b,
                                                      const float* __restrict__ se3_w, const float* __restrict__ se3_b, int hidden,
                                                      float* __restrict__ ch_out, int C) {
    const size_t b = blockIdx.x;
    __shared__ float part[256], mean[512], hid[64];
    const int cw = C < 256 ? C : 256;          // channels handled per pass
    const int nsl = 256 / cw;                  // block slices summed in parallel (C is a multiple of 8)
    for (int c0 = 0; c0 < C; c0 += cw) {
        const int c = c0 + threadIdx.x % cw, sl = threadIdx.x / cw;
        float s = 0.f;
        if (sl < nsl && c < C) {
            const float* src = partial + b * nblk * C + c;
            int k = sl;
            for (; k + 7 * nsl < nblk; k += 8 * nsl) {      // 8 loads in flight, summed in block order
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = src[(size_t)(k + u * nsl) * C];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += t[u];
            }
            for (; k < nblk; k += nsl) s += src[(size_t)k * C];
        }
        part[threadIdx.x] = s;
        __syncthreads();
        if (threadIdx.x < cw && c0 + threadIdx.x < C) {
            float t = 0.f;
            for (int q = 0; q < nsl; ++q) t += part[q * cw + threadIdx.x];
            mean[c0 + threadIdx.x] = t / (float)P;
        }
        __syncthreads();
    }
    for (int m = threadIdx.x; m < hidden; m += 256) {
        float s = se1_b[m];
        for (int c = 0; c < C; ++c) s = fmaf(se1_w[m * C + c], mean[c], s);
        hid[m] = fmaxf(s, 0.f);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = se3_b[c];
        for (int m = 0; m < hidden; ++m) s = fmaf(se3_w[c * hidden + m], hid[m], s);
        ch_out[b * C + c] = 1.0f / (1.0f + expf(-s));
    }
}

// wp_out[b] = pack([W_a diag(ch_b) | W_b]) in MFMA operand order
// (and, when wp3_out is given, the same matrix in b3 form for the bf16x3 GEMM: rf_common.h).
// With `wc` (tail_compose_kernel below: W_b W_2, [C][hc]) the matrix is [W_a diag(ch_b) | W_b | W_b W_2], K = 2C + hc, written in
// b3 form only: the weights of the composed stage tail (rf_model.hip, run_stage).  ch == nullptr: no gate (plain variant).
__global__ void __launch_bounds__(256) flca_fold_kernel(const float* __restrict__ w_cr, const float* __restrict__ ch, const float* __restrict__ wc,
                                                        float* __restrict__ wp_out, unsigned short* __restrict__ wp3_out, int C, int hc) {
    const size_t b = blockIdx.y;
    const int K = 2 * C + (wc ? hc : 0);
    const int NT = (C + 15) >> 4, NS = K >> 2, NB = (K + 31) >> 5;
    float* dst = wp_out ? wp_out + b * (size_t)NT * NS * 64 : nullptr;
    unsigned short* dst3 = wp3_out ? wp3_out + b * (size_t)NT * NB * 1536 : nullptr;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < NT * NB * 512; idx += gridDim.x * 256) {
        const int l = idx & 63, t = (idx >> 6) % NT, s = (idx >> 6) / NT;      // s runs over the 8 NB k-sets of the padded matrix
        const int co = 16 * t + (l & 15), k = 4 * s + (l >> 4);
        float v = 0.f;
        if (co < C && k < 2 * C) v = w_cr[(size_t)co * 2 * C + k] * ((k < C && ch) ? ch[b * C + k] : 1.0f);
        else if (co < C && k < K) v = wc[(size_t)co * hc + (k - 2 * C)];
        if (dst && s < NS) dst[((size_t)s * NT + t) * 64 + l] = v;
        if (dst3) b3_store(dst3, NT, co, k, v);
    }
}

// Composed stage tail, once per parameter load:  channel_reduce([xs ; x1 + W_2 g + b_2]) = W_a xs + W_b x1 + (W_b W_2) g + (W_b b_2 + b_cr)
// out[co * hc + k] = sum_m W_cr[co][C + m] W_2[m][k]   (k < hc),   out[C * hc + co] = b_cr[co] + sum_m W_cr[co][C + m] b_2[m]
__global__ void __launch_bounds__(256) tail_compose_kernel(const float* __restrict__ w_cr, const float* __restrict__ b_cr, const float* __restrict__ w2,
                                                           const float* __restrict__ b2, float* __restrict__ out, int C, int hc) {
    const int n = C * hc + C;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < n; idx += gridDim.x * 256) {
        float v;
        if (idx < C * hc) {
            const int co = idx / hc, k = idx - co * hc;
            v = 0.f;
            for (int m = 0; m < C; ++m) v = fmaf(w_cr[(size_t)co * 2 * C + C + m], w2[(size_t)m * hc + k], v);
        } else {
            const int co = idx - C * hc;
            v = b_cr ? b_cr[co] : 0.f;
            if (b2) for (int m = 0; m < C; ++m) v = fmaf(w_cr[(size_t)co * 2 * C + C + m], b2[m], v);
        }
        out[idx] = v;
    }
}

size_t tail_composed_floats(int C, int hc) { return (size_t)C * hc + C; }

int pack_tail(const float* w_cr, const float* b_cr, const float* w2, const float* b2, float* composed, int C, int hc, hipStream_t st) {
    tail_compose_kernel<<<cdiv(C * hc + C, 256), 256, 0, st>>>(w_cr, b_cr, w2, b2, composed, C, hc);
    return check_launch("pack_tail");
}

static int fold_grid(int C, int K) {
    const int gx = cdiv(cdiv(C, 16) * cdiv(K, 32) * 512, 256 * 4);
    return gx < 1 ? 1 : gx;
}

// [W_a diag(ch_b) | W_b | W_b W_2] in b3 form for B images (ch == nullptr: B = 1 static set without a gate)
int launch_tail_fold(const float* w_cr, const float* ch, const float* composed, void* wp3_out, int B, int C, int hc, hipStream_t st) {
    RF_CHECK_ARG(w_cr && composed && wp3_out && C % 32 == 0 && hc % 32 == 0, "tail_fold: bad arguments (C = %d, hidden = %d)", C, hc);
    flca_fold_kernel<<<dim3((unsigned)fold_grid(C, 2 * C + hc), (unsigned)B), 256, 0, st>>>(w_cr, ch, composed, nullptr, (unsigned short*)wp3_out, C, hc);
    return check_launch("tail_fold");
}

int launch_flca_se(const float* partial, int nblk, int P, const float* se1_w, const float* se1_b,
                   const float* se3_w, const float* se3_b, int hidden, float* ch_out, int B, int C, hipStream_t st) {
    RF_CHECK_ARG(C <= 512 && hidden <= 64 && C % 8 == 0 && ch_out, "flca_se: C=%d hidden=%d unsupported", C, hidden);
    flca_se_kernel<<<B, 256, 0, st>>>(partial, nblk, P, se1_w, se1_b, se3_w, se3_b, hidden, ch_out, C);
    return check_launch("flca_se");
}

// out = in * ch[b][c]  (in == out: in place), 16 bytes per lane when the plane size allows
__global__ void __launch_bounds__(256) scale_channels_kernel(const float* in, float* out, const float* __restrict__ ch, int P, size_t total) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) out[i] = in[i] * ch[i / P];
}
__global__ void __launch_bounds__(256) scale_channels4_kernel(const float4* in, float4* out, const float* __restrict__ ch, int P4, size_t total4) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const float s = ch[i / P4];
        float4 v = in[i];
        v.x *= s; v.y *= s; v.z *= s; v.w *= s;
        out[i] = v;
    }
}

int launch_scale_channels_to(const float* in, float* out, const float* ch, int B, int C, int P, hipStream_t st) {
    const size_t total = (size_t)B * C * P;
    if (P % 4 == 0 && aligned16(in) && aligned16(out)) {
        size_t g = (total / 4 + 255) / 256;
        if (g > 16384) g = 16384;
        scale_channels4_kernel<<<(unsigned)g, 256, 0, st>>>(reinterpret_cast<const float4*>(in), reinterpret_cast<float4*>(out), ch, P / 4, total / 4);
    } else {
        int g = (int)((total + 255) / 256);
        if (g > 4096) g = 4096;
        scale_channels_kernel<<<g, 256, 0, st>>>(in, out, ch, P, total);
    }
    return check_launch("scale_channels");
}
int launch_scale_channels(float* x, const float* ch, int B, int C, int P, hipStream_t st) { return launch_scale_channels_to(x, x, ch, B, C, P, st); }

int launch_flca_se_fold(const float* partial, int nblk, int P, const float* se1_w, const float* se1_b,
                        const float* se3_w, const float* se3_b, int hidden, const float* w_cr,
                        float* wp_out, void* wp3_out, float* ch_out, int B, int C, hipStream_t st, const float* composed, int hc) {
    ProfScope prof(st, "flca_se_kernel+flca_fold_kernel", 0.0, 0.0);
    const int rc = launch_flca_se(partial, nblk, P, se1_w, se1_b, se3_w, se3_b, hidden, ch_out, B, C, st);
    if (rc) return rc;
    if (composed) return launch_tail_fold(w_cr, ch_out, composed, wp3_out, B, C, hc, st);
    flca_fold_kernel<<<dim3((unsigned)fold_grid(C, 2 * C), (unsigned)B), 256, 0, st>>>(w_cr, ch_out, nullptr, wp_out, (unsigned short*)wp3_out, C, 0);
    return check_launch("flca_se_fold");
}

}  // namespace rf
