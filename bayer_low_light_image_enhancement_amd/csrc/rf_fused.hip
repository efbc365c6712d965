// Fused transformer-block kernels for U-Net level 0 of RawFormer-S (C = 32 channels; the templates are written
// for any C % 16 == 0, but C = 64 needs 400+ registers and lost to the op-by-op schedule, so only <32> is built).
//
// Un-fused, one TransformerBlock moves ~26 C floats per pixel through HBM (qkv 1x1: C in / 3C out,
// depthwise 3x3: 3C/3C, Gram: 2C, ...; ffn: C/2C, 2C/2C, 2C+C/C).  Here the wide intermediates
// (qkv before/after the depthwise conv, the FFN hidden tensor) never leave the CU:
//
//   ffn_fused_kernel    x1 -> LN2 -> 1x1 (C->2C) -> dw3x3 -> GELU -> 1x1 (2C->C) + x1      reads C, writes C
//   attn_front_kernel   x  -> LN1 -> 1x1 (C->3C) -> dw3x3 -> { Gram partials of q,k ; v }   reads C, writes C
//
// Tile = 4 rows x 64 px per workgroup (4 waves, one output row each) plus a 1-pixel halo, held as
// 6 rows x 72 columns (columns x0-4 .. x0+67, so every 4-pixel group is 16-byte aligned in HBM and
// either wholly inside or wholly outside the image).
//   Phase A (per part of 32 intermediate channels): each wave owns 27 of the 108 halo'd pixel
//     groups as two MFMA steps; its LayerNorm'd input (C x 64 px per step) stays in registers for
//     all parts (the resident-input GEMM of rf_gemm1x1.hip), D tiles go to LDS (+bias, zero outside
//     the image: the depthwise conv pads ITS input, i.e. the 1x1 output).
//   Phase B: a lane reads the 3x6 neighbourhood of its 4 pixels from LDS, runs the 9-tap stencil
//     in registers and feeds the result straight into the next MFMA as B operand (FFN: second 1x1;
//     attention: q k^T, q q^T, k k^T with the pixel axis as K) or stores it (v).
// Weights of the current part are staged in LDS in MFMA lane order.  All cross-workgroup
// reductions go through fixed-order partials (bitwise reproducible).
#include <cstdio>
#include <cstdlib>
#include "rf_common.h"

namespace rf {

#ifdef RF_STAMP
// Diagnostic build only (python -m ...build --stamp): per-phase cycle sums of wave 0 lanes, read back
// with rf_debug_stamps().  Never compiled into the shipped library.
__device__ unsigned long long g_stamp[16];
#define STAMP_DECL unsigned long long st_prev = __builtin_amdgcn_s_memtime(), st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_prev; st_prev = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define STAMP_FLUSH do { if ((threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_stamp[i_], st_acc[i_]); } while (0)
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH
#endif

// Workgroup barrier that orders LDS traffic only.  __syncthreads() makes hipcc drain vmcnt(0) first,
// which would stall every wave on the next tile's prefetch loads that are deliberately in flight.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

namespace fused {
constexpr int TH = 4, TW = 64;          // output tile
constexpr int HR = TH + 2;              // halo'd rows
constexpr int HC = 72;                  // halo'd columns held (18 groups of 4 px)
constexpr int NG = HR * (HC / 4);       // 108 pixel groups per tile
constexpr int GPW = NG / 4;             // 27 groups per wave in phase A (two MFMA steps: 16 + 11)
constexpr int PART = 32;                // intermediate channels per part
// Plane stride of the Gram rounds, where lane (j, kq) reads 16 bytes at plane j, column 4 kq (+ 16 st): a ds_read_b128 is
// served in four groups of 16 lanes, each holding every j once with two different kq (MI355X_MICROARCH.md, LDS), so the
// groups are conflict-free iff (PSG j + 4 kq) mod 64 are 16 distinct multiples of 4: PSG = 8 mod 64.  (452 = 4 mod 64 made
// lanes (j, 1) and (j + 1, 0) collide: SQ_LDS_BANK_CONFLICT was 55 % of the LDS-active cycles of attn_front.)
constexpr int PSG = 456;
}  // namespace fused

// ---- shared phase-A machinery ------------------------------------------------------------------
// The wave's two steps of halo'd pixel groups: geometry of step st for lane j.
struct GroupGeom {
    int lds_off;     // row * HC + 4 * cg
    int goff;        // y * w + x  (clamped to 0 when outside)
    bool valid;
};
__device__ __forceinline__ GroupGeom group_geom(int wave, int st, int j, int y0, int x0, int h, int w) {
    using namespace fused;
    GroupGeom g;
    const int gi = wave * GPW + st * 16 + j;
    const bool in_step = (st == 0) || (j < GPW - 16);
    const int row = gi / (HC / 4), cg = gi % (HC / 4);
    const int y = y0 - 1 + row, x = x0 - 4 + 4 * cg;
    g.valid = in_step && y >= 0 && y < h && x >= 0 && x < w;
    g.lds_off = in_step ? row * HC + 4 * cg : -1;
    g.goff = g.valid ? y * w + x : 0;
    return g;
}

// Input tile of one step: raw loads (issue early: the registers are dead between a tile's last
// phase A and the next tile's first one, so the next tile is fetched behind the current phase B) ...
template <int C>
__device__ __forceinline__ void load_step(const float* __restrict__ xb, int P, int kq, const GroupGeom& g, float4 (&xh)[C / 4]) {
    const unsigned voff = (unsigned)kq * (unsigned)P + (unsigned)g.goff;
#pragma unroll
    for (int s = 0; s < C / 4; ++s) xh[s] = *reinterpret_cast<const float4*>(xb + (size_t)(4 * s) * P + voff);
}

// ... and the exact two-pass LayerNorm over channels, in place: xh[s] = LN(x)[channel 4s + kq][4 px].
template <int C>
__device__ __forceinline__ void ln_step(int kq, const float* __restrict__ gam_l, const float* __restrict__ bet_l,
                                        float eps, float4 (&xh)[C / 4]) {
    constexpr int NS = C / 4;
    float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s) { sum[0] += xh[s].x; sum[1] += xh[s].y; sum[2] += xh[s].z; sum[3] += xh[s].w; }
    float mu[4], var[4] = {0.f, 0.f, 0.f, 0.f}, rstd[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        sum[q] += __shfl_xor(sum[q], 16);
        sum[q] += __shfl_xor(sum[q], 32);
        mu[q] = sum[q] * (1.0f / C);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const float d0 = xh[s].x - mu[0], d1 = xh[s].y - mu[1], d2 = xh[s].z - mu[2], d3 = xh[s].w - mu[3];
        var[0] = fmaf(d0, d0, var[0]); var[1] = fmaf(d1, d1, var[1]);
        var[2] = fmaf(d2, d2, var[2]); var[3] = fmaf(d3, d3, var[3]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        var[q] += __shfl_xor(var[q], 16);
        var[q] += __shfl_xor(var[q], 32);
        rstd[q] = 1.0f / sqrtf(var[q] * (1.0f / C) + eps);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const float gk = gam_l[4 * s + kq], bk = bet_l[4 * s + kq];
        xh[s].x = fmaf((xh[s].x - mu[0]) * rstd[0], gk, bk);
        xh[s].y = fmaf((xh[s].y - mu[1]) * rstd[1], gk, bk);
        xh[s].z = fmaf((xh[s].z - mu[2]) * rstd[2], gk, bk);
        xh[s].w = fmaf((xh[s].w - mu[3]) * rstd[3], gk, bk);
    }
}

// One phase-A step: 2 output tiles (32 intermediate channels) of the 1x1 GEMM, written to the LDS
// plane array `mid` (plane stride PS) with bias; groups outside the image are written as zeros.
template <int C, int WT>
__device__ __forceinline__ void phase_a_step(const float4 (&xh)[C / 4], const float* __restrict__ wl /* [C/4][WT][64] + lane */,
                                             int tile0, int tile1, const float* __restrict__ bias0, const float* __restrict__ bias1,
                                             float* __restrict__ mid, int PS, int kq, const GroupGeom& g) {
    f32x4 acc[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[t][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* w0 = wl + tile0 * 64;
    const float* w1 = wl + tile1 * 64;
#pragma unroll
    for (int s = 0; s < C / 4; ++s) {
        const float xb[4] = {xh[s].x, xh[s].y, xh[s].z, xh[s].w};
        const float a0 = w0[s * WT * 64], a1 = w1[s * WT * 64];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[0][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, xb[q], acc[0][q], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[1][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, xb[q], acc[1][q], 0, 0, 0);
    }
    if (g.lds_off >= 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float bs = (t ? bias1 : bias0)[4 * kq + r];
                float4 v = make_float4(acc[t][0][r] + bs, acc[t][1][r] + bs, acc[t][2][r] + bs, acc[t][3][r] + bs);
                if (!g.valid) v = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(mid + (16 * t + 4 * kq + r) * PS + g.lds_off) = v;
            }
    }
}

// ---- b3 forms of the phase-A helpers (rf_common.h): lane (j, kq) holds channels 32 kb + 8 kq + i, i = 0..7 ------------------
template <int C>
__device__ __forceinline__ void load_step_b3(const float* __restrict__ xb, int P, int kq, const GroupGeom& g, float4 (&xh)[C / 4]) {
    const unsigned voff = (unsigned)(8 * kq) * (unsigned)P + (unsigned)g.goff;
#pragma unroll
    for (int s = 0; s < C / 4; ++s) xh[s] = *reinterpret_cast<const float4*>(xb + (size_t)(32 * (s >> 3) + (s & 7)) * P + voff);
}

template <int C>
__device__ __forceinline__ void ln_step_b3(int kq, const float* __restrict__ gam_l, const float* __restrict__ bet_l, float eps, float4 (&xh)[C / 4]) {
    constexpr int NS = C / 4;
    float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s) { sum[0] += xh[s].x; sum[1] += xh[s].y; sum[2] += xh[s].z; sum[3] += xh[s].w; }
    float mu[4], var[4] = {0.f, 0.f, 0.f, 0.f}, rstd[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        sum[q] += __shfl_xor(sum[q], 16);
        sum[q] += __shfl_xor(sum[q], 32);
        mu[q] = sum[q] * (1.0f / C);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const float d0 = xh[s].x - mu[0], d1 = xh[s].y - mu[1], d2 = xh[s].z - mu[2], d3 = xh[s].w - mu[3];
        var[0] = fmaf(d0, d0, var[0]); var[1] = fmaf(d1, d1, var[1]);
        var[2] = fmaf(d2, d2, var[2]); var[3] = fmaf(d3, d3, var[3]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        var[q] += __shfl_xor(var[q], 16);
        var[q] += __shfl_xor(var[q], 32);
        rstd[q] = 1.0f / sqrtf(var[q] * (1.0f / C) + eps);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int ch = 32 * (s >> 3) + 8 * kq + (s & 7);
        const float gk = gam_l[ch], bk = bet_l[ch];
        xh[s].x = fmaf((xh[s].x - mu[0]) * rstd[0], gk, bk);
        xh[s].y = fmaf((xh[s].y - mu[1]) * rstd[1], gk, bk);
        xh[s].z = fmaf((xh[s].z - mu[2]) * rstd[2], gk, bk);
        xh[s].w = fmaf((xh[s].w - mu[3]) * rstd[3], gk, bk);
    }
}

template <int C>
__device__ __forceinline__ void split_step(const float4 (&xh)[C / 4], u32x4 (&bp)[C / 32][4][3]) {
#pragma unroll
    for (int kb = 0; kb < C / 32; ++kb)
#pragma unroll
        for (int hp = 0; hp < 4; ++hp) {
            const float4 va = xh[8 * kb + 2 * hp], vb = xh[8 * kb + 2 * hp + 1];
            const float xa[4] = {va.x, va.y, va.z, va.w}, xc[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned a0, a1, a2, b0, b1, b2;
                b3_split(xa[g], a0, a1, a2);
                b3_split(xc[g], b0, b1, b2);
                bp[kb][g][0][hp] = b3_pack(a0, b0);
                bp[kb][g][1][hp] = b3_pack(a1, b1);
                bp[kb][g][2][hp] = b3_pack(a2, b2);
            }
        }
}

template <int C, int WT>
__device__ __forceinline__ void phase_a_step_b3(const u32x4 (&bp)[C / 32][4][3], const u32x4* __restrict__ wl,
                                                int tile0, int tile1, const float* __restrict__ bias0, const float* __restrict__ bias1,
                                                float* __restrict__ mid, int PS, int kq, const GroupGeom& g) {
    f32x4 acc[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[t][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < C / 32; ++kb) {
        const u32x4* w0 = wl + (size_t)(kb * WT + tile0) * 192;
        const u32x4* w1 = wl + (size_t)(kb * WT + tile1) * 192;
        const u32x4 a0[3] = {w0[0], w0[64], w0[128]}, a1[3] = {w1[0], w1[64], w1[128]};
        b3_mfma4(a0, bp[kb], acc[0]);
        b3_mfma4(a1, bp[kb], acc[1]);
    }
    if (g.lds_off >= 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float bs = (t ? bias1 : bias0)[4 * kq + r];
                float4 v = make_float4(acc[t][0][r] + bs, acc[t][1][r] + bs, acc[t][2][r] + bs, acc[t][3][r] + bs);
                if (!g.valid) v = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(mid + (16 * t + 4 * kq + r) * PS + g.lds_off) = v;
            }
    }
}

// 3x3 depthwise stencil for 4 consecutive pixels from an LDS plane: `p` points at the plane's
// (row of the output pixel - 1, column of the first pixel), 16-byte aligned.  The two edge taps of
// every row must not be scalar LDS reads (lanes 4 floats apart are a 4-way bank conflict on
// ds_read_b32: measured 58 % of all LDS cycles), so:
//   stencil4_dpp   lanes j-1 / j+1 of the same 16-lane row hold the neighbouring 4-pixel groups:
//                  edges come over DPP row shifts; only lanes 0 and 15 read their outer tap.
//   stencil4_wide  neighbouring groups are not in this wave's registers: three aligned
//                  ds_read_b128 per row (conflict-free with the plane stride used there).
__device__ __forceinline__ float dpp_row_shr1(float keep, float v) {   // lane i <- lane i-1 (i % 16 == 0 keeps `keep`)
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(keep), __float_as_int(v), 0x111, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_row_shl1(float keep, float v) {   // lane i <- lane i+1 (i % 16 == 15 keeps `keep`)
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(keep), __float_as_int(v), 0x101, 0xf, 0xf, false));
}

__device__ __forceinline__ void stencil4_dpp(const float* __restrict__ p, int j, const float* __restrict__ k9, float bias, float (&out)[4]) {
    using namespace fused;
#pragma unroll
    for (int q = 0; q < 4; ++q) out[q] = bias;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const float* row = p + dy * HC;
        const float4 m = *reinterpret_cast<const float4*>(row);
        float el = 0.f, er = 0.f;
        if (j == 0) el = row[-1];
        if (j == 15) er = row[4];
        const float v[6] = {dpp_row_shr1(el, m.w), m.x, m.y, m.z, m.w, dpp_row_shl1(er, m.x)};
        const float k0 = k9[dy * 3], k1 = k9[dy * 3 + 1], k2 = k9[dy * 3 + 2];
#pragma unroll
        for (int q = 0; q < 4; ++q) out[q] = fmaf(k2, v[q + 2], fmaf(k1, v[q + 1], fmaf(k0, v[q], out[q])));
    }
}

// the same two stencils on rows at explicit offsets ro[dy] (a circular row window in LDS: attn_mid_kernel)
__device__ __forceinline__ void stencil4_dpp_r(const float* __restrict__ p, const int (&ro)[3], int j, const float* __restrict__ k9, float bias, float (&out)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) out[q] = bias;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const float* row = p + ro[dy];
        const float4 m = *reinterpret_cast<const float4*>(row);
        float el = 0.f, er = 0.f;
        if (j == 0) el = row[-1];
        if (j == 15) er = row[4];
        const float v[6] = {dpp_row_shr1(el, m.w), m.x, m.y, m.z, m.w, dpp_row_shl1(er, m.x)};
        const float k0 = k9[dy * 3], k1 = k9[dy * 3 + 1], k2 = k9[dy * 3 + 2];
#pragma unroll
        for (int q = 0; q < 4; ++q) out[q] = fmaf(k2, v[q + 2], fmaf(k1, v[q + 1], fmaf(k0, v[q], out[q])));
    }
}
__device__ __forceinline__ void stencil4_wide_r(const float* __restrict__ p, const int (&ro)[3], const float* __restrict__ k9, float bias, float (&out)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) out[q] = bias;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const float* row = p + ro[dy];
        const float4 lft = *reinterpret_cast<const float4*>(row - 4);
        const float4 m = *reinterpret_cast<const float4*>(row);
        const float4 rgt = *reinterpret_cast<const float4*>(row + 4);
        const float v[6] = {lft.w, m.x, m.y, m.z, m.w, rgt.x};
        const float k0 = k9[dy * 3], k1 = k9[dy * 3 + 1], k2 = k9[dy * 3 + 2];
#pragma unroll
        for (int q = 0; q < 4; ++q) out[q] = fmaf(k2, v[q + 2], fmaf(k1, v[q + 1], fmaf(k0, v[q], out[q])));
    }
}

__device__ __forceinline__ void stencil4_wide(const float* __restrict__ p, const float* __restrict__ k9, float bias, float (&out)[4]) {
    using namespace fused;
#pragma unroll
    for (int q = 0; q < 4; ++q) out[q] = bias;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const float* row = p + dy * HC;
        const float4 lft = *reinterpret_cast<const float4*>(row - 4);
        const float4 m = *reinterpret_cast<const float4*>(row);
        const float4 rgt = *reinterpret_cast<const float4*>(row + 4);
        const float v[6] = {lft.w, m.x, m.y, m.z, m.w, rgt.x};
        const float k0 = k9[dy * 3], k1 = k9[dy * 3 + 1], k2 = k9[dy * 3 + 2];
#pragma unroll
        for (int q = 0; q < 4; ++q) out[q] = fmaf(k2, v[q + 2], fmaf(k1, v[q + 1], fmaf(k0, v[q], out[q])));
    }
}

// ================================================================================================
// FFN:  out = x1 + W2 gelu(dw3x3(W1 LN2(x1) + b1) + bd) + b2
// ================================================================================================
struct FfnArgs {
    const float* x;        // [B][C][h][w] block input (also the residual)
    float* out;            // [B][C][h][w]
    const float* ln_w; const float* ln_b;
    const void* w1p;       // b3-packed pw1 weight [C/32][2C/16][3][64] 16-byte elements
    const float* b1;       // [2C]
    const float* wd;       // [2C][9]
    const float* bd;       // [2C]
    const float* w2p;      // packed [2C/4][C/16][64]
    const float* b2;       // [C]
    int B, h, w, tiles_x, ntiles;
};

template <int C>
__global__ void __launch_bounds__(256, 2) ffn_fused_kernel(FfnArgs a) {
    using namespace fused;
    constexpr int NS = C / 4;            // k-sets of the first GEMM
    constexpr int NT1 = 2 * C / 16;      // output tiles of the first GEMM (hidden)
    constexpr int NTO = C / 16;          // output tiles of the second GEMM
    constexpr int NPART = 2 * C / PART;  // parts of 32 hidden channels
    constexpr int PS = 448;              // LDS plane stride (multiple of 64: kq planes on disjoint slots)
    // all weights live in LDS for the lifetime of the (persistent) workgroup
    __shared__ __attribute__((aligned(16))) float mid[PART * PS + 8];
    __shared__ __attribute__((aligned(16))) u32x4 w1_l[(C / 32) * NT1 * 192];      // b3 form (phase A runs on the bf16 instruction)
    __shared__ __attribute__((aligned(16))) float w2_l[(2 * C / 4) * NTO * 64];
    __shared__ float wd_l[2 * C * 9], bd_l[2 * C], b1_l[2 * C], b2_l[C], gam_l[C], bet_l[C];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int b = blockIdx.y;
    const int h = a.h, w = a.w, P = h * w;
    const float* xb = a.x + (size_t)b * C * P;
    float* ob = a.out + (size_t)b * C * P;

    for (int i = tid; i < (C / 32) * NT1 * 192; i += 256) w1_l[i] = reinterpret_cast<const u32x4*>(a.w1p)[i];
    for (int i = tid; i < (2 * C / 4) * NTO * 16; i += 256) *reinterpret_cast<float4*>(w2_l + i * 4) = *reinterpret_cast<const float4*>(a.w2p + i * 4);
    for (int i = tid; i < 2 * C * 9; i += 256) wd_l[i] = a.wd[i];
    for (int i = tid; i < 2 * C; i += 256) { bd_l[i] = a.bd[i]; b1_l[i] = a.b1[i]; }
    for (int i = tid; i < C; i += 256) { gam_l[i] = a.ln_w[i]; bet_l[i] = a.ln_b[i]; b2_l[i] = a.b2[i]; }
    __syncthreads();
    STAMP_DECL

    // A workgroup's tiles are CONSECUTIVE and numbered down the columns of the tile grid (ty fastest): successive tiles share two
    // of their six halo'd rows, which are then still in L2 (strided tiles, numbered along x, re-fetched every halo row from HBM)
    const int tiles_y = a.ntiles / a.tiles_x;
    const int per = (a.ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int t_begin = blockIdx.x * per, t_end = (t_begin + per < a.ntiles) ? t_begin + per : a.ntiles;
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int tx = tile / tiles_y, ty = tile % tiles_y;
        const int x0 = tx * TW, y0 = ty * TH;
        STAMP(0);
        // input tile, loaded here (see attn_front_kernel: the b3 pieces take the registers a tile in flight would need):
        // LayerNorm in registers, then the three-piece split that both parts of phase A reuse
        const GroupGeom gw0 = group_geom(wave, 0, j, y0, x0, h, w), gw1 = group_geom(wave, 1, j, y0, x0, h, w);
        u32x4 bp0[C / 32][4][3], bp1[C / 32][4][3];
        {
            float4 xh0[NS], xh1[NS];
            load_step_b3<C>(xb, P, kq, gw0, xh0);
            load_step_b3<C>(xb, P, kq, gw1, xh1);
            ln_step_b3<C>(kq, gam_l, bet_l, 1e-5f, xh0);
            ln_step_b3<C>(kq, gam_l, bet_l, 1e-5f, xh1);
            split_step<C>(xh0, bp0);
            split_step<C>(xh1, bp1);
        }
        STAMP(1);

        const int yo = y0 + wave, xo = x0 + 4 * j;          // this lane's 4 output pixels
        const bool live = yo < h && xo < w;
        const unsigned voff = (unsigned)(4 * kq) * (unsigned)P + (unsigned)(live ? yo * w + xo : 0);
        float4 resv[NTO * 4];
        f32x4 acc[NTO][4];
#pragma unroll
        for (int t = 0; t < NTO; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[t][q] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
        for (int part = 0; part < NPART; ++part) {
            lds_barrier();                                 // previous phase B is done with mid
            STAMP(0);
            // ---- phase A: hidden[32 of part][halo tile] = W1 x^ + b1 -> LDS
            phase_a_step_b3<C, NT1>(bp0, w1_l + lane, 2 * part, 2 * part + 1, b1_l + part * PART, b1_l + part * PART + 16, mid, PS, kq, gw0);
            phase_a_step_b3<C, NT1>(bp1, w1_l + lane, 2 * part, 2 * part + 1, b1_l + part * PART, b1_l + part * PART + 16, mid, PS, kq, gw1);
            STAMP(2);
            if (part == NPART - 1) {
                // this tile's residual rows, behind the last phase B (issued before this tile's stores)
#pragma unroll
                for (int t = 0; t < NTO; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) resv[t * 4 + r] = *reinterpret_cast<const float4*>(xb + (size_t)(16 * t + r) * P + voff);
            }
            lds_barrier();
            STAMP(0);
            // ---- phase B: depthwise 3x3 + GELU in registers, straight into the second GEMM
#pragma unroll
            for (int s = 0; s < PART / 4; ++s) {
                const int hc = 4 * s + kq;
                float v[4];
                stencil4_dpp(mid + hc * PS + wave * HC + 4 * j + 4, j, wd_l + (part * PART + hc) * 9, bd_l[part * PART + hc], v);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = gelu_fast(v[q]);
#pragma unroll
                for (int t = 0; t < NTO; ++t) {
                    const float av = w2_l[((part * (PART / 4) + s) * NTO + t) * 64 + lane];
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[t][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, v[q], acc[t][q], 0, 0, 0);
                }
            }
            STAMP(3);
        }
        // ---- epilogue: + b2 + residual
        if (live) {
#pragma unroll
            for (int t = 0; t < NTO; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int cu = 16 * t + r;
                    const float bs = b2_l[cu + 4 * kq];
                    const float4 rv = resv[t * 4 + r];
                    *reinterpret_cast<float4*>(ob + (size_t)cu * P + voff) =
                        make_float4(acc[t][0][r] + bs + rv.x, acc[t][1][r] + bs + rv.y, acc[t][2][r] + bs + rv.z, acc[t][3][r] + bs + rv.w);
                }
        }
        STAMP(4);
    }
    STAMP_FLUSH;
}

// ================================================================================================
// The same FFN for C = 48 and C = 64 (level 0 of RawFormer-B / -L, level 1 of RawFormer-S) on EIGHT-wave workgroups, one per CU.
// Two steps of b3 pieces per wave (the C = 32 kernel) would need 192 registers at C = 64; here the 108 halo'd pixel groups are
// spread over eight waves -- 14 each, ONE phase-A step and 96 registers of pieces per wave -- and phase B splits the K dimension
// of the second GEMM instead of the pixels: waves 0-3 and 4-7 own the same four output rows, wave half `hf` runs the
// depthwise stencil + GELU and the MFMAs of k-steps [4 hf, 4 hf + 4) of every part (no stencil is evaluated twice), and the two
// partial accumulator sets are added through LDS once per tile (64 KB = the `mid` planes, free by then).  K of the first GEMM
// is padded to a multiple of 32 with zero pieces (C = 48: the packed weight is zero-padded the same way).
// LDS at C = 64: 64 KB mid / reduction + 48 KB W1 (b3) + 32 KB W2 + 6 KB of vectors = 150 KB.
// ================================================================================================
namespace fused8 {
constexpr int GPW = 14;                 // pixel groups per wave in phase A (8 x 14 = 112 >= 108)
}

__device__ __forceinline__ GroupGeom group_geom8(int wave, int j, int y0, int x0, int h, int w) {
    using namespace fused;
    GroupGeom g;
    const int gi = wave * fused8::GPW + j;
    const bool in_step = j < fused8::GPW && gi < NG;
    const int row = gi / (HC / 4), cg = gi % (HC / 4);
    const int y = y0 - 1 + row, x = x0 - 4 + 4 * cg;
    g.valid = in_step && y >= 0 && y < h && x >= 0 && x < w;
    g.lds_off = in_step ? row * HC + 4 * cg : -1;
    g.goff = g.valid ? y * w + x : 0;
    return g;
}

// input of one step, K padded to KP: lane (j, kq) holds channels 32 kb + 8 kq + i; channels >= C are zeros (loaded from a
// clamped address, selected away: the loads stay branch-free)
template <int C, int KP>
__device__ __forceinline__ void load_step_b3p(const float* __restrict__ xb, int P, int kq, const GroupGeom& g, float4 (&xh)[KP / 4]) {
#pragma unroll
    for (int s = 0; s < KP / 4; ++s) {
        const int ch = 32 * (s >> 3) + 8 * kq + (s & 7);
        xh[s] = *reinterpret_cast<const float4*>(xb + (size_t)(ch < C ? ch : 0) * P + (unsigned)g.goff);
    }
}

template <int C, int KP>
__device__ __forceinline__ void ln_step_b3p(int kq, const float* __restrict__ gam_l, const float* __restrict__ bet_l, float eps, float4 (&xh)[KP / 4]) {
    constexpr int NS = KP / 4;
    float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const bool ok = 32 * (s >> 3) + 8 * kq + (s & 7) < C;
        if (!ok) xh[s] = make_float4(0.f, 0.f, 0.f, 0.f);
        sum[0] += xh[s].x; sum[1] += xh[s].y; sum[2] += xh[s].z; sum[3] += xh[s].w;
    }
    float mu[4], var[4] = {0.f, 0.f, 0.f, 0.f}, rstd[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        sum[q] += __shfl_xor(sum[q], 16);
        sum[q] += __shfl_xor(sum[q], 32);
        mu[q] = sum[q] * (1.0f / C);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const float m = (32 * (s >> 3) + 8 * kq + (s & 7) < C) ? 1.0f : 0.f;
        const float d0 = m * (xh[s].x - mu[0]), d1 = m * (xh[s].y - mu[1]), d2 = m * (xh[s].z - mu[2]), d3 = m * (xh[s].w - mu[3]);
        var[0] = fmaf(d0, d0, var[0]); var[1] = fmaf(d1, d1, var[1]);
        var[2] = fmaf(d2, d2, var[2]); var[3] = fmaf(d3, d3, var[3]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        var[q] += __shfl_xor(var[q], 16);
        var[q] += __shfl_xor(var[q], 32);
        rstd[q] = 1.0f / sqrtf(var[q] * (1.0f / C) + eps);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int ch = 32 * (s >> 3) + 8 * kq + (s & 7);
        const bool ok = ch < C;
        const float gk = gam_l[ok ? ch : 0], bk = bet_l[ok ? ch : 0];
        xh[s].x = ok ? fmaf((xh[s].x - mu[0]) * rstd[0], gk, bk) : 0.f;
        xh[s].y = ok ? fmaf((xh[s].y - mu[1]) * rstd[1], gk, bk) : 0.f;
        xh[s].z = ok ? fmaf((xh[s].z - mu[2]) * rstd[2], gk, bk) : 0.f;
        xh[s].w = ok ? fmaf((xh[s].w - mu[3]) * rstd[3], gk, bk) : 0.f;
    }
}

template <int C>
__global__ void __launch_bounds__(512, 1) ffn_fused8_kernel(FfnArgs a) {
    using namespace fused;
    constexpr int KP = (C + 31) / 32 * 32; // K of the first GEMM, padded
    constexpr int NT1 = 2 * C / 16;      // output tiles of the first GEMM (hidden)
    constexpr int NTO = C / 16;          // output tiles of the second GEMM
    constexpr int NPART = 2 * C / PART;  // parts of 32 hidden channels
    constexpr int PS = 448;
    constexpr int MIDF = (PART * PS + 8 > NTO * 4 * 256 * 4) ? PART * PS + 8 : NTO * 4 * 256 * 4;      // planes, or the K-split reduction
    __shared__ __attribute__((aligned(16))) float mid[MIDF];
    __shared__ __attribute__((aligned(16))) u32x4 w1_l[(KP / 32) * NT1 * 192];
    __shared__ __attribute__((aligned(16))) float w2_l[(2 * C / 4) * NTO * 64];
    __shared__ float wd_l[2 * C * 9], bd_l[2 * C], b1_l[2 * C], b2_l[C], gam_l[C], bet_l[C];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = wave & 3, hf = wave >> 2;
    const int b = blockIdx.y;
    const int h = a.h, w = a.w, P = h * w;
    const float* xb = a.x + (size_t)b * C * P;
    float* ob = a.out + (size_t)b * C * P;

    for (int i = tid; i < (KP / 32) * NT1 * 192; i += 512) w1_l[i] = reinterpret_cast<const u32x4*>(a.w1p)[i];
    for (int i = tid; i < (2 * C / 4) * NTO * 16; i += 512) *reinterpret_cast<float4*>(w2_l + i * 4) = *reinterpret_cast<const float4*>(a.w2p + i * 4);
    for (int i = tid; i < 2 * C * 9; i += 512) wd_l[i] = a.wd[i];
    for (int i = tid; i < 2 * C; i += 512) { bd_l[i] = a.bd[i]; b1_l[i] = a.b1[i]; }
    for (int i = tid; i < C; i += 512) { gam_l[i] = a.ln_w[i]; bet_l[i] = a.ln_b[i]; b2_l[i] = a.b2[i]; }
    __syncthreads();

    const int tiles_y = a.ntiles / a.tiles_x;
    const int per = (a.ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int t_begin = blockIdx.x * per, t_end = (t_begin + per < a.ntiles) ? t_begin + per : a.ntiles;
    const int j0 = lane & 15, kq0 = lane >> 4, lane0 = lane;
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int tx = tile / tiles_y, ty = tile % tiles_y;
        const int x0 = tx * TW, y0 = ty * TH;
        // the lane coordinates are made opaque once per tile: everything derived from them (LDS plane / weight / stencil addresses)
        // would otherwise be hoisted out of the tile loop into long-lived registers and spilled (112 of them)
        int j = j0, kq = kq0, lane = lane0;
        asm volatile("" : "+v"(j), "+v"(kq), "+v"(lane));
        const GroupGeom gw = group_geom8(wave, j, y0, x0, h, w);
        u32x4 bp[KP / 32][4][3];
        {
            float4 xh[KP / 4];
            load_step_b3p<C, KP>(xb, P, kq, gw, xh);
            ln_step_b3p<C, KP>(kq, gam_l, bet_l, 1e-5f, xh);
            split_step<KP>(xh, bp);
        }
        const int yo = y0 + row, xo = x0 + 4 * j;            // this lane's 4 output pixels (both wave halves)
        const bool live = yo < h && xo < w;
        const unsigned voff = (unsigned)(4 * kq) * (unsigned)P + (unsigned)(live ? yo * w + xo : 0);
        f32x4 acc[NTO][4];
#pragma unroll
        for (int t = 0; t < NTO; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[t][q] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
        for (int part = 0; part < NPART; ++part) {
            lds_barrier();                                 // previous phase B (or the previous tile's reduction) is done with mid
            phase_a_step_b3<KP, NT1>(bp, w1_l + lane, 2 * part, 2 * part + 1, b1_l + part * PART, b1_l + part * PART + 16, mid, PS, kq, gw);
            lds_barrier();
            // ---- phase B: this wave half's four k-steps of the part (not unrolled: hoisted stencil loads would spill)
#pragma unroll 1
            for (int s4 = 0; s4 < PART / 8; ++s4) {
                const int s = 4 * hf + s4, hc = 4 * s + kq;
                float v[4];
                stencil4_dpp(mid + hc * PS + row * HC + 4 * j + 4, j, wd_l + (part * PART + hc) * 9, bd_l[part * PART + hc], v);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = gelu_fast(v[q]);
#pragma unroll
                for (int t = 0; t < NTO; ++t) {
                    const float av = w2_l[((part * (PART / 4) + s) * NTO + t) * 64 + lane];
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[t][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, v[q], acc[t][q], 0, 0, 0);
                }
            }
        }
        // ---- the two K halves: waves 4-7 hand their accumulators to waves 0-3 through LDS (mid is free after the barrier)
        lds_barrier();
        float4* red = reinterpret_cast<float4*>(mid);
        if (hf == 1) {
#pragma unroll
            for (int t = 0; t < NTO; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) red[(t * 4 + q) * 256 + row * 64 + lane] = make_float4(acc[t][q][0], acc[t][q][1], acc[t][q][2], acc[t][q][3]);
        }
        float4 resv[NTO * 4];
        if (hf == 0) {                                     // residual rows, in flight across the barrier
#pragma unroll
            for (int t = 0; t < NTO; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) resv[t * 4 + r] = *reinterpret_cast<const float4*>(xb + (size_t)(16 * t + r) * P + voff);
        }
        lds_barrier();
        if (hf == 0) {
#pragma unroll
            for (int t = 0; t < NTO; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 o = red[(t * 4 + q) * 256 + row * 64 + lane];
                    acc[t][q][0] += o.x; acc[t][q][1] += o.y; acc[t][q][2] += o.z; acc[t][q][3] += o.w;
                }
            if (live) {
#pragma unroll
                for (int t = 0; t < NTO; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int cu = 16 * t + r;
                        const float bs = b2_l[cu + 4 * kq];
                        const float4 rv = resv[t * 4 + r];
                        *reinterpret_cast<float4*>(ob + (size_t)cu * P + voff) =
                            make_float4(acc[t][0][r] + bs + rv.x, acc[t][1][r] + bs + rv.y, acc[t][2][r] + bs + rv.z, acc[t][3][r] + bs + rv.w);
                    }
            }
        }
    }
}

// C = 64 is built and tested (diagnostic twin: RF_FFN8_64=1) but NOT dispatched: measured on MI355X the eight-wave kernel ties the
// op-by-op chain there (RawFormer-L level 0, 3 M pixels: 1.98 ms against 1.85 ms; RawFormer-S level 1: 0.33 against 0.30 ms) --
// at 64 channels the kernel is bound by MFMA + VALU issue (they do not overlap on this chip) where the chain is bound by HBM,
// and both take the same time.  C = 48 (RawFormer-B level 0) gains 8 % of its FFN.
bool fused_ffn_supported(int C, int hidden, int h, int w) {
#ifdef RF_DIAG
    if (C == 64 && getenv("RF_FFN8_64") && hidden == 2 * C && (w % 4 == 0) && ((double)C * h * w * 4.0 < 4.0e9)) return true;
#endif
    return (C == 32 || C == 48) && hidden == 2 * C && (w % 4 == 0) && ((double)C * h * w * 4.0 < 4.0e9);
}

int launch_ffn_fused(const float* x, float* out, const float* ln_w, const float* ln_b, const void* w1p, const float* b1,
                     const float* wd, const float* bd, const float* w2p, const float* b2, int B, int C, int h, int w, hipStream_t st) {
    RF_CHECK_ARG((C == 32 || C == 48 || C == 64) && w % 4 == 0 && (double)C * h * w * 4.0 < 4.0e9 && B <= 65535, "ffn_fused: unsupported shape C=%d %dx%d", C, h, w);
    RF_CHECK_ARG(aligned16(x) && aligned16(out) && aligned16(w1p), "ffn_fused: buffers must be 16-byte aligned");
    FfnArgs a{x, out, ln_w, ln_b, w1p, b1, wd, bd, w2p, b2, B, h, w, cdiv(w, fused::TW), 0};
    a.ntiles = a.tiles_x * cdiv(h, fused::TH);
    int wgs = cdiv(512, B);                       // persistent: two workgroups per CU over the whole batch
    if (wgs > a.ntiles) wgs = a.ntiles;
    const dim3 grid((unsigned)wgs, (unsigned)B);
    const double px = (double)B * h * w;
    if (C != 32) {
        // eight-wave form: one workgroup per CU, persistent over the whole batch
        int wg8 = cdiv(256, B);
        if (wg8 > a.ntiles) wg8 = a.ntiles;
        const dim3 grid8((unsigned)wg8, (unsigned)B);
        ProfScope prof(st, C == 64 ? "ffn_fused8_kernel<64>" : "ffn_fused8_kernel<48>", px * (8.0 * C * C + 36.0 * C), px * 8.0 * C);
        if (C == 64) ffn_fused8_kernel<64><<<grid8, 512, 0, st>>>(a);
        else ffn_fused8_kernel<48><<<grid8, 512, 0, st>>>(a);
        return check_launch("ffn_fused8");
    }
    ProfScope prof(st, "ffn_fused_kernel<32>", px * (8.0 * C * C + 36.0 * C), px * 8.0 * C);
    ffn_fused_kernel<32><<<grid, 256, 0, st>>>(a);
    return check_launch("ffn_fused");
}

// ================================================================================================
// Attention front:  qkv = dw3x3(Wqkv LN1(x) + b);  Gram partials of (q, k) per head;  v -> HBM
// ================================================================================================
struct AttnFrontArgs {
    const float* x;        // [B][C][h][w]
    float* v;              // [B][C][h][w]  depthwise-convolved v
    float* partial;        // [B][nslab][C/16][16][66]  (layout of rf_attn.hip: band of one k tile)
    const float* ln_w; const float* ln_b;
    const void* wp;        // b3-packed qkv weight [C/32][3C/16][3][64] 16-byte elements
    const float* bq;       // [3C]
    const float* wd;       // [3C][9]
    const float* bd;       // [3C]
    int B, h, w, tiles_x, ntiles, nslab;
    int ylo, yhi;          // rows [ylo, yhi) enter the Gram statistics (a spatial shard's interior; the whole image otherwise)
};

template <int C>
__global__ void __launch_bounds__(256, 2) attn_front_kernel(AttnFrontArgs a) {
    using namespace fused;
    constexpr int NS = C / 4;
    constexpr int NQT = C / 16;          // q (and k) tiles = Gram rounds
    constexpr int NVP = C / PART;        // v parts
    constexpr int NT3 = 3 * C / 16;      // tiles of the qkv weight
    constexpr int PSG = fused::PSG;      // plane stride for the Gram rounds (see fused::PSG)
    constexpr int PSV = 448;             // plane stride for the v parts: lanes run along pixels
    constexpr int ROWW = 4 * 16 + 2;     // partial row width of rf_attn.hip (kMaxBand * 16 + 2)
    __shared__ __attribute__((aligned(16))) float mid[PART * PSG + 8];
    __shared__ __attribute__((aligned(16))) u32x4 w_l[(C / 32) * NT3 * 192];     // whole qkv weight (b3), resident
    __shared__ float wd_l[3 * C * 9], bd_l[3 * C], bq_l[3 * C], gam_l[C], bet_l[C];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int slab = blockIdx.x, b = blockIdx.y;
    const int h = a.h, w = a.w, P = h * w;
    const float* xb = a.x + (size_t)b * C * P;
    float* vb = a.v + (size_t)b * C * P;

    for (int i = tid; i < (C / 32) * NT3 * 192; i += 256) w_l[i] = reinterpret_cast<const u32x4*>(a.wp)[i];
    for (int i = tid; i < 3 * C * 9; i += 256) wd_l[i] = a.wd[i];
    for (int i = tid; i < 3 * C; i += 256) { bd_l[i] = a.bd[i]; bq_l[i] = a.bq[i]; }
    for (int i = tid; i < C; i += 256) { gam_l[i] = a.ln_w[i]; bet_l[i] = a.ln_b[i]; }

    // |q_j|^2 and |k_j|^2 are plain per-lane sums of squares (2 VALU per value): as the diagonals of q q^T and k k^T on the matrix
    // pipe they cost eight 33-cycle MFMAs per step for 32 useful numbers -- and MFMA time and VALU time add up on this chip
    f32x4 gq[NQT];
    float nq[NQT], nk[NQT];
#pragma unroll
    for (int r = 0; r < NQT; ++r) { gq[r] = (f32x4){0.f, 0.f, 0.f, 0.f}; nq[r] = 0.f; nk[r] = 0.f; }
    STAMP_DECL

    // consecutive tiles down the columns of the tile grid (see ffn_fused_kernel)
    const int tiles_y = a.ntiles / a.tiles_x;
    const int per = (a.ntiles + a.nslab - 1) / a.nslab;
    const int t_begin = slab * per, t_end = (t_begin + per < a.ntiles) ? t_begin + per : a.ntiles;
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int tx = tile / tiles_y, ty = tile % tiles_y;
        const int x0 = tx * TW, y0 = ty * TH;
        // The input tile is loaded here, not a tile ahead: the b3 pieces (96 registers) already fill the budget that the
        // f32 kernel of round 1 spent on the next tile's raw values.  The loads are issued before the barrier so that their
        // latency overlaps the wait (throw-away loads warming L2 for the next tile were measured: 3 % slower).
        const GroupGeom g0 = group_geom(wave, 0, j, y0, x0, h, w), g1 = group_geom(wave, 1, j, y0, x0, h, w);
        float4 xh0[NS], xh1[NS];
        load_step_b3<C>(xb, P, kq, g0, xh0);
        load_step_b3<C>(xb, P, kq, g1, xh1);
        lds_barrier();                                   // weights visible; previous tile finished with mid
        STAMP(0);
        ln_step_b3<C>(kq, gam_l, bet_l, 1e-5f, xh0);
        ln_step_b3<C>(kq, gam_l, bet_l, 1e-5f, xh1);
        u32x4 bp0[C / 32][4][3], bp1[C / 32][4][3];
        split_step<C>(xh0, bp0);
        split_step<C>(xh1, bp1);
        const int yo = y0 + wave;
        STAMP(1);

        // ---- Gram rounds: q tile r (plane 0-15) with k tile r (planes 16-31); heads never straddle a tile here
#pragma unroll
        for (int r = 0; r < NQT; ++r) {
            if (r) lds_barrier();
            STAMP(0);
            phase_a_step_b3<C, NT3>(bp0, w_l + lane, r, NQT + r, bq_l + 16 * r, bq_l + C + 16 * r, mid, PSG, kq, g0);
            phase_a_step_b3<C, NT3>(bp1, w_l + lane, r, NQT + r, bq_l + 16 * r, bq_l + C + 16 * r, mid, PSG, kq, g1);
            STAMP(2);
            lds_barrier();
            STAMP(0);
            // phase B: lane (i = j, kq) owns channel i of the q tile and of the k tile at pixels x0 + 16*st + 4*kq + m
            const int cq = 16 * r + j, ck = C + 16 * r + j;
#pragma unroll 1                                         // unrolled, the hoisted stencil loads push the kernel into scratch
            for (int st = 0; st < 4; ++st) {
                const int xo = x0 + 16 * st + 4 * kq;
                const bool ok = yo >= a.ylo && yo < a.yhi && xo < w;
                float qa[4], kb[4];
                stencil4_wide(mid + j * PSG + wave * HC + 16 * st + 4 * kq + 4, wd_l + cq * 9, bd_l[cq], qa);
                stencil4_wide(mid + (16 + j) * PSG + wave * HC + 16 * st + 4 * kq + 4, wd_l + ck * 9, bd_l[ck], kb);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const float qv = ok ? qa[m] : 0.f, kv = ok ? kb[m] : 0.f;
                    gq[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(qv, kv, gq[r], 0, 0, 0);
                    nq[r] = fmaf(qv, qv, nq[r]);
                    nk[r] = fmaf(kv, kv, nk[r]);
                }
            }
            STAMP(3);
        }
        // ---- v parts: 1x1 -> LDS -> depthwise -> HBM
#pragma unroll
        for (int vp = 0; vp < NVP; ++vp) {
            lds_barrier();
            STAMP(0);
            const int t0 = 2 * NQT + 2 * vp;
            phase_a_step_b3<C, NT3>(bp0, w_l + lane, t0, t0 + 1, bq_l + 2 * C + vp * PART, bq_l + 2 * C + vp * PART + 16, mid, PSV, kq, g0);
            phase_a_step_b3<C, NT3>(bp1, w_l + lane, t0, t0 + 1, bq_l + 2 * C + vp * PART, bq_l + 2 * C + vp * PART + 16, mid, PSV, kq, g1);
            STAMP(2);
            lds_barrier();
            STAMP(0);
            const int xo = x0 + 4 * j;
            if (yo < h && xo < w) {
#pragma unroll
                for (int s = 0; s < PART / 4; ++s) {
                    const int hc = 4 * s + kq, cv = 2 * C + vp * PART + hc;
                    float v[4];
                    stencil4_dpp(mid + hc * PSV + wave * HC + 4 * j + 4, j, wd_l + cv * 9, bd_l[cv], v);
                    // uniform base + 32-bit lane offset (C P < 2^30 elements, checked by fused_attn_supported): no 64-bit per-lane pointer to keep
                    *reinterpret_cast<float4*>(vb + (unsigned)((vp * PART + hc) * P + yo * w + xo)) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
            STAMP(4);
        }
    }
    STAMP_FLUSH;
    // ---- cross-wave reduction of the Gram tiles in a fixed order, one partial per workgroup.
    // The zero key tiles of a partial row are written by their own strided loop, not next to the Gram values: hipcc 7.2 paired
    // `rr[j] = g; rr[16 + j] = 0; rr[32 + j] = 0; rr[48 + j] = 0` into ds_write2_b32 with offset0 4 instead of 16 when the
    // accumulators sat in AGPRs (seen with __launch_bounds__(256, 1)), i.e. zeros over the neighbouring Gram columns.
    __syncthreads();
    float* red = mid;                                      // [4 waves][16][ROWW] floats = 4224 <= PART * PSG
#pragma unroll
    for (int r = 0; r < NQT; ++r) {
        for (int i = tid; i < 64 * 48; i += 256) red[(i / 48) * ROWW + 16 + i % 48] = 0.f;
        // channel j's sums of squares: the four kq lanes of a channel hold disjoint pixels
        float nqt = nq[r], nkt = nk[r];
        nqt += __shfl_xor(nqt, 16); nqt += __shfl_xor(nqt, 32);
        nkt += __shfl_xor(nkt, 16); nkt += __shfl_xor(nkt, 32);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = 4 * kq + q;
            float* rr = red + (wave * 16 + row) * ROWW;
            rr[j] = gq[r][q];
            if (row == j) { rr[64] = nqt; rr[65] = nkt; }
        }
        __syncthreads();
        float* dst = a.partial + (((size_t)b * a.nslab + slab) * NQT + r) * 16 * ROWW;
        for (int i = tid; i < 16 * ROWW; i += 256)
            dst[i] = ((red[i] + red[16 * ROWW + i]) + red[2 * 16 * ROWW + i]) + red[3 * 16 * ROWW + i];
        __syncthreads();
    }
}

bool fused_attn_supported(int C, int heads, int h, int w) {
    const int c = heads > 0 ? C / heads : 0;
    return C == 32 && heads > 0 && C % heads == 0 && c <= 16 && 16 % c == 0 && (w % 4 == 0) &&
           ((double)C * h * w * 4.0 < 4.0e9);
}

int fused_attn_plan(int h, int w, int* nslab, size_t* partial_floats, int B, int C) {
    const int ntiles = cdiv(w, fused::TW) * cdiv(h, fused::TH);
    int ns = cdiv(ntiles, 4);                   // 4 tiles (1024 px) per workgroup: ONE 1024x1024 frame gives the chip 256 workgroups
                                                // (8 tiles: 131 us per launch for one frame, 4 tiles: 75 us; a batch of 8 pays
                                                // 0.6 %); depends on the image only, never on B: an image's reduction order is
                                                // batch-invariant
    if (ntiles <= 256) ns = ntiles;             // frames up to 256 x 256 packed: one tile per workgroup (a workgroup's tiles are a
                                                // chain of ~15 us each: 62 -> 25 us per launch for one 128 x 128 frame)
                                                // (two tiles per workgroup at 512 x 512: -17 % for one frame, +11 % for a batch of 8)
    if (ns < 1) ns = 1;
    *nslab = ns;
    *partial_floats = (size_t)B * ns * (C / 16) * 16 * 66;
    return RF_OK;
}

int launch_attn_front(const float* x, float* v, float* partial, int nslab, const float* ln_w, const float* ln_b,
                      const void* wp, const float* bq, const float* wd, const float* bd, int B, int C, int h, int w, hipStream_t st,
                      int ylo, int yhi) {
    RF_CHECK_ARG(C == 32 && w % 4 == 0 && B <= 65535, "attn_front: unsupported shape C=%d %dx%d", C, h, w);
    RF_CHECK_ARG(aligned16(x) && aligned16(v), "attn_front: buffers must be 16-byte aligned");
    AttnFrontArgs a{x, v, partial, ln_w, ln_b, wp, bq, wd, bd, B, h, w, cdiv(w, fused::TW), 0, nslab, ylo, (yhi > 0 && yhi < h) ? yhi : h};
    a.ntiles = a.tiles_x * cdiv(h, fused::TH);
    const double px = (double)B * h * w;
    ProfScope prof(st, "attn_front_kernel<32>", px * (6.0 * C * C + 54.0 * C + 4.0 * C * 16), px * 8.0 * C);
    const dim3 grid((unsigned)nslab, (unsigned)B);
    attn_front_kernel<32><<<grid, 256, 0, st>>>(a);
    return check_launch("attn_front");
}

// ================================================================================================
// Attention middle for levels where the qkv 1x1 stays a separate GEMM (C = 64, 128):
//   qkv [B,3C,h,w] (HBM) -> depthwise 3x3 -> { Gram partials of (q, k) per head ; v -> HBM }
// i.e. attn_front_kernel with phase A replaced by staging halo'd qkv tiles from HBM: the depthwise-convolved
// q and k never exist in memory (un-fused: dwconv writes 3C and the Gram kernel reads 2C of it back).
// Round r stages q tile r (planes 0-15) and k tile r (planes 16-31) -- heads never straddle a 16-channel tile
// here -- and the v rounds 32 channels each; a round's 14 x 16-byte loads per thread are issued before the
// previous round's phase B and land in LDS after it (hardware zero fill outside the image, like rf_conv3x3.hip).
// ================================================================================================
struct AttnMidArgs {
    const float* qkv;      // [B][3C][h][w]
    float* v;              // [B][C][h][w]  depthwise-convolved v
    float* partial;        // [B][nslab][C/16][16][66]  (layout of attn_front_kernel / rf_attn.hip)
    const float* wd;       // [3C][9]
    const float* bd;       // [3C]
    int B, h, w, tiles_x, ntiles, nslab;
    int ylo, yhi;          // as AttnFrontArgs
    int rgroups;           // gridDim.z: the rounds (Gram tiles, then v parts) are split over this many workgroups per slab
};

template <int C>
__global__ void __launch_bounds__(256, 2) attn_mid_kernel(AttnMidArgs a) {
    using namespace fused;
    constexpr int NQT = C / 16;          // Gram rounds
    constexpr int NVP = C / PART;        // v rounds
    constexpr int NR = NQT + NVP;
    constexpr int PSG = fused::PSG, PSV = 448, ROWW = 4 * 16 + 2;
    constexpr int NF4 = PART * HR * (HC / 4);            // 16-byte elements of one staged round (32 planes x 6 x 18)
    constexpr int FPT = (NF4 + 255) / 256;
    constexpr unsigned OOB = 0x80000000u;
    __shared__ __attribute__((aligned(16))) float mid[PART * PSG + 8];
    __shared__ float wd_l[3 * C * 9], bd_l[3 * C];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int slab = blockIdx.x, b = blockIdx.y;
    const int h = a.h, w = a.w, P = h * w;
    const float* qb = a.qkv + (size_t)b * 3 * C * P;
    float* vb = a.v + (size_t)b * C * P;
    for (int i = tid; i < 3 * C * 9; i += 256) wd_l[i] = a.wd[i];
    for (int i = tid; i < 3 * C; i += 256) bd_l[i] = a.bd[i];

    // element e = tid + 256 i of a round = (plane p, halo row, 4-pixel group g)
    int pk[FPT];
#pragma unroll
    for (int i = 0; i < FPT; ++i) {
        const int e = tid + 256 * i;
        const int pl = e / (HR * (HC / 4)), rem = e % (HR * (HC / 4));
        pk[i] = e < NF4 ? (pl << 16) | ((rem / (HC / 4)) << 8) | (rem % (HC / 4)) : -1;
    }
    // A workgroup's tiles are consecutive and run DOWN a column of the tile grid, and the six halo'd rows of a plane live in a
    // circular window of LDS rows (image row y in slot (y + 1) mod 6): a tile directly below the previous one of the same round
    // finds its first two rows already there and stages only the four new ones -- 4.5 instead of 6.75 floats read per pixel
    // and channel (the measured HBM traffic of this kernel had been 2.1-2.3 x its algorithmic bytes).
    const int tiles_y = a.ntiles / a.tiles_x;
    const int per = (a.ntiles + a.nslab - 1) / a.nslab;
    unsigned voff[FPT];       // pending tile: byte offset of (row, group) inside a plane, OOB outside the image / already in LDS
    int pend_ys = 0;          // ... slot of its halo row 0
    bool pend_full = true;    // ... all six rows are staged (first tile of a round or of a column)
    auto plan_tile = [&](int tile, bool full) {
        const int y0 = (tile % tiles_y) * TH, x0 = (tile / tiles_y) * TW;
        pend_ys = y0 % HR; pend_full = full;
#pragma unroll
        for (int i = 0; i < FPT; ++i) {
            int e = pk[i];
            asm volatile("" : "+v"(e));
            const int r = (e >> 8) & 255;
            const int y = y0 - 1 + r, x = x0 - 4 + 4 * (e & 255);
            const bool ok = e >= 0 && (full || r >= 2) && (unsigned)y < (unsigned)h && (unsigned)x < (unsigned)w;      // w % 4 == 0: whole groups
            voff[i] = ok ? (unsigned)((y * w + x) * 4) : OOB;
        }
    };
    float4 stg[FPT];
    // round rd < NQT: q tile rd | k tile rd;  rd >= NQT: v channels 32 (rd - NQT) ..
    auto load_round = [&](int rd) {
        const bool gram = rd < NQT;
        const size_t base = gram ? (size_t)16 * rd * P : (size_t)(2 * C + PART * (rd - NQT)) * P;
        // num_records = the planes this round touches (q tile + k tile, C planes apart; or 32 v planes), never the rest of the
        // tensor: it must stay below the OOB offset 2^31 for the zero fill to work on large frames (3C planes of a 1424x2128
        // level are 2.3 GB)
        const size_t span = (gram ? (size_t)(C + 16) : (size_t)PART) * P, rest = (size_t)3 * C * P - base;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(qb) + base, 0, (int)((span < rest ? span : rest) * 4), 0x00020000);
        const unsigned kjump = gram ? (unsigned)((C - 16) * P) * 4u : 0u;      // planes 16-31 of a Gram round are the k tile
#pragma unroll
        for (int i = 0; i < FPT; ++i) {
            const int pl = pk[i] >> 16;
            const unsigned off = voff[i] + (unsigned)pl * (unsigned)P * 4u + (pl >= 16 ? kjump : 0u);   // OOB stays >= 2^31
            typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
            const u32x4_t v4 = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
            stg[i] = make_float4(__uint_as_float(v4.x), __uint_as_float(v4.y), __uint_as_float(v4.z), __uint_as_float(v4.w));
        }
    };
    auto store_round = [&](int rd) {      // the pending tile's rows into their slots (rows 0-1 of a sliding tile are already there)
        const int PSX = rd < NQT ? PSG : PSV;
#pragma unroll
        for (int i = 0; i < FPT; ++i) {
            int e = pk[i];
            asm volatile("" : "+v"(e));       // opaque: nothing derived from the element id is hoisted out of the step loop (and spilled)
            const int r = (e >> 8) & 255;
            const int slot = pend_ys + r - (pend_ys + r >= HR ? HR : 0);
            if (e >= 0 && (pend_full || r >= 2)) *reinterpret_cast<float4*>(mid + (e >> 16) * PSX + slot * HC + 4 * (e & 255)) = stg[i];
        }
    };

    // Rounds are the OUTER loop and this workgroup's tiles the inner one, so one register set holds the Gram tile of the
    // round across all tiles (a round index into a register array would go to scratch); (round, tile) is one flattened
    // pipeline: the next step's loads are issued before this step's phase B.
    const int t_begin = slab * per;
    if (t_begin >= a.ntiles) {                             // (whole workgroup) no tiles: its Gram partials are zero
        if ((int)blockIdx.z == 0)
            for (int i = tid; i < NQT * 16 * ROWW; i += 256) a.partial[((size_t)b * a.nslab + slab) * NQT * 16 * ROWW + i] = 0.f;
        return;
    }
    const int ntw = (t_begin + per < a.ntiles) ? per : a.ntiles - t_begin;      // tiles of this workgroup: t_begin, t_begin + 1, ...
    f32x4 gq = {0.f, 0.f, 0.f, 0.f};
    float nq = 0.f, nk = 0.f;                              // sums of squares on the VALU (see attn_front_kernel)
    // rounds of this workgroup: every round is independent (its own Gram partial or its own v channels), so a launch with few
    // slabs (one frame) spreads them over gridDim.z workgroups per slab -- same partials, same results
    const int rd_lo = (int)blockIdx.z * NR / a.rgroups, rd_hi = ((int)blockIdx.z + 1) * NR / a.rgroups;
    plan_tile(t_begin, true);
    load_round(rd_lo);
    __syncthreads();                                      // wd_l / bd_l visible
    for (int rd = rd_lo; rd < rd_hi; ++rd) {
        for (int ti = 0; ti < ntw; ++ti) {
            const int tile = t_begin + ti;
            const int x0 = (tile / tiles_y) * TW, y0 = (tile % tiles_y) * TH;
            const int yo = y0 + wave;
            lds_barrier();                                // everyone is done reading the previous step
            store_round(rd);
            // LDS row offsets of this wave's three stencil rows (halo rows wave .. wave + 2 of this tile)
            int ro[3];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int sl = pend_ys + wave + dy;
                ro[dy] = (sl - (sl >= HR ? HR : 0)) * HC;
            }
            if (ti + 1 < ntw) {                           // next step: same round, next tile / next round, first tile
                plan_tile(tile + 1, (tile + 1) % tiles_y == 0);      // a new column starts with a full window
                load_round(rd);
            } else if (rd + 1 < rd_hi) {
                plan_tile(t_begin, true);
                load_round(rd + 1);
            }
            lds_barrier();
            if (rd < NQT) {
                // Gram: lane (i = j, kq) owns channel j of the q tile and of the k tile at pixels x0 + 16 st + 4 kq + m
                const int cq = 16 * rd + j, ck = C + 16 * rd + j;
#pragma unroll 1
                for (int st = 0; st < 4; ++st) {
                    const int xo = x0 + 16 * st + 4 * kq;
                    const bool ok = yo >= a.ylo && yo < a.yhi && xo < w;
                    float qa[4], kb[4];
                    stencil4_wide_r(mid + j * PSG + 16 * st + 4 * kq + 4, ro, wd_l + cq * 9, bd_l[cq], qa);
                    stencil4_wide_r(mid + (16 + j) * PSG + 16 * st + 4 * kq + 4, ro, wd_l + ck * 9, bd_l[ck], kb);
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const float qv = ok ? qa[m] : 0.f, kv = ok ? kb[m] : 0.f;
                        gq = __builtin_amdgcn_mfma_f32_16x16x4f32(qv, kv, gq, 0, 0, 0);
                        nq = fmaf(qv, qv, nq);
                        nk = fmaf(kv, kv, nk);
                    }
                }
            } else {
                const int vp = rd - NQT, xo = x0 + 4 * j;
                if (yo < h && xo < w) {
#pragma unroll
                    for (int s = 0; s < PART / 4; ++s) {
                        const int hc = 4 * s + kq, cv = 2 * C + vp * PART + hc;
                        float v[4];
                        stencil4_dpp_r(mid + hc * PSV + 4 * j + 4, ro, j, wd_l + cv * 9, bd_l[cv], v);
                        *reinterpret_cast<float4*>(vb + (size_t)(vp * PART + hc) * P + (size_t)yo * w + xo) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                }
            }
        }
        if (rd < NQT) {
            // ---- cross-wave reduction of this round's Gram tile in a fixed order, one partial per workgroup
            // (the next step's data is still in registers: mid is free between the two barriers)
            __syncthreads();
            float* red = mid;
            for (int i = tid; i < 64 * 48; i += 256) red[(i / 48) * ROWW + 16 + i % 48] = 0.f;      // zero key tiles: see attn_front_kernel
            int kq_ = kq;                     // opaque here: the row addresses below were hoisted out of the round loop and SPILLED
            asm volatile("" : "+v"(kq_));     // (scratch traffic next to the prefetched loads of the next round)
            float nqt = nq, nkt = nk;         // channel j's sums of squares over the four kq lanes
            nqt += __shfl_xor(nqt, 16); nqt += __shfl_xor(nqt, 32);
            nkt += __shfl_xor(nkt, 16); nkt += __shfl_xor(nkt, 32);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = 4 * kq_ + q;
                float* rr = red + (wave * 16 + row) * ROWW;
                rr[j] = gq[q];
                if (row == j) { rr[64] = nqt; rr[65] = nkt; }
            }
            __syncthreads();
            float* dst = a.partial + (((size_t)b * a.nslab + slab) * NQT + rd) * 16 * ROWW;
            for (int i = tid; i < 16 * ROWW; i += 256)
                dst[i] = ((red[i] + red[16 * ROWW + i]) + red[2 * 16 * ROWW + i]) + red[3 * 16 * ROWW + i];
            gq = (f32x4){0.f, 0.f, 0.f, 0.f}; nq = 0.f; nk = 0.f;
        }
    }
}

// slabs of the image (= workgroups per image): enough of them to fill the chip at a batch of 8, at most 8 tiles each;
// a function of the image only (batch-invariant reduction order)
int attn_mid_plan(int h, int w, int* nslab, size_t* partial_floats, int B, int C) {
    const int ntiles = cdiv(w, fused::TW) * cdiv(h, fused::TH);
    int per = ntiles / 64;                      // (finer slabs -- ntiles / 128 -- are 40 % faster for ONE frame and 9 % slower for a
                                                // batch of 8: the tile loop is what hides this kernel's load latency)
    if (per < 1) per = 1;
    if (per > 8) per = 8;
    *nslab = cdiv(ntiles, per);
    *partial_floats = (size_t)B * *nslab * (C / 16) * 16 * 66;
    return RF_OK;
}

bool attn_mid_supported(int C, int heads, int h, int w) {
    const int c = heads > 0 ? C / heads : 0;
    return (C == 64 || C == 128) && heads > 0 && C % heads == 0 && c <= 16 && 16 % c == 0 && (w % 4 == 0) &&
           ((double)(C + 16) * h * w * 4.0 < 2.0e9);      // byte offsets inside one round's buffer window stay below 2^31
}

int launch_attn_mid(const float* qkv, float* v, float* partial, int nslab, const float* wd, const float* bd,
                    int B, int C, int h, int w, hipStream_t st, int ylo, int yhi) {
    RF_CHECK_ARG((C == 64 || C == 128) && w % 4 == 0 && B <= 65535, "attn_mid: unsupported shape C=%d %dx%d", C, h, w);
    RF_CHECK_ARG(aligned16(qkv) && aligned16(v), "attn_mid: buffers must be 16-byte aligned");
    AttnMidArgs a{qkv, v, partial, wd, bd, B, h, w, cdiv(w, fused::TW), 0, nslab, ylo, (yhi > 0 && yhi < h) ? yhi : h, 1};
    a.ntiles = a.tiles_x * cdiv(h, fused::TH);
    if ((long)nslab * B < 256) a.rgroups = 3;               // C / 16 + C / 32 rounds: 6 (C = 64) or 12 (C = 128)
    const double px = (double)B * h * w;
    ProfScope prof(st, C == 64 ? "attn_mid_kernel<64>" : "attn_mid_kernel<128>", px * (54.0 * C + 4.0 * C * 16), px * 16.0 * C);
    const dim3 grid((unsigned)nslab, (unsigned)B, (unsigned)a.rgroups);
    if (C == 64) attn_mid_kernel<64><<<grid, 256, 0, st>>>(a);
    else attn_mid_kernel<128><<<grid, 256, 0, st>>>(a);
    return check_launch("attn_mid");
}

#ifdef RF_STAMP
extern "C" int rf_debug_stamps(unsigned long long* out8) {   // diagnostic build only: read and reset the phase cycle sums
    unsigned long long z[16] = {0};
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_stamp), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif

}  // namespace rf
