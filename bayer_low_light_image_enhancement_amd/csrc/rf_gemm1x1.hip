// 1x1 convolution = per-image GEMM  out[co][p] = sum_k W[co][k] * x[k][p]  on the f32 matrix
// cores (v_mfma_f32_16x16x4_f32: exact f32 fmaf chain, so RawFormer keeps reference numerics).
//
// Operand mapping (one wave = 64 pixels):
//   * B: lane (j = l & 15, kq = l >> 4) reads ONE float4 = pixels p0 + 4j .. 4j+3 of channel
//     4s + kq: 16 lanes cover 256 contiguous bytes of a channel row, one wave instruction 1 KiB.
//     Component g of that float4 is the B operand of MFMA g, so x never goes through LDS (each
//     element is used by exactly one wave) and every HBM byte of x is fetched once.
//   * A: one float per lane from the lane-ordered packed weights, staged in LDS per workgroup
//     together with the bias and the LayerNorm scale/shift.
//   * D: lane holds channels 16t + 4kq + r of its 4 pixels -> 16-byte stores, 256 B per 16 lanes.
// Addresses are (wave-uniform channel base) + (one per-lane 32-bit offset).
//
// gfx950 has ONE in-order counter (vmcnt) for loads and stores: a load that is waited for drags
// every older store with it.  The kernels therefore issue every global load a tile needs
// (next tile's input, this tile's residual) BEFORE the tile's stores, keep loads branch-free
// (clamped addresses instead of predicates, so the compiler emits counted waits), and read all
// small operands (bias, gamma, beta, weights) from LDS, which has its own counter.
//
//   conv1x1_res_kernel    K <= 128 (U-Net levels 0-1, HBM-bound): persistent workgroups; a wave
//                         keeps its whole [K x 64 px] input tile in registers, computes exact
//                         two-pass LayerNorm statistics on it, then sweeps ALL output-channel
//                         groups over the resident tile (x is read from HBM once whatever Cout
//                         is) while the next tile's loads are already in flight (K <= 64).
//   conv1x1_stream_kernel K  > 128 (levels 2-3, MFMA-bound): accumulators resident, K streamed
//                         in chunks; the next chunk's x (registers) and weights (registers ->
//                         LDS double buffer) are fetched while the current one is in the matrix
//                         pipe; one barrier per chunk.
//   conv1x1_scalar_kernel any ragged shape (P % 4 != 0 or unaligned): plain FMA loop.
// Fused: LayerNorm prologue (a4), second source (torch.cat without the copy), bias, residual,
// LeakyReLU, ConvTranspose2d(k=2,s=2) scatter (a9), per-image weights (wp_bstride != 0) for the
// attention / squeeze-excite matrices folded into the projection.
#include <cstdio>
#include <cstdlib>
#include "rf_common.h"

namespace rf {

// wave-uniform base of the 4-channel k-set s (channels 4s .. 4s+3 come from one source)
__device__ __forceinline__ const float* kset_base(const Conv1x1Args& a, int b, int s) {
    const int k = 4 * s;
    const bool first = k < a.C1;                       // selects, not branches: a branch around the loads that use this
    const float* src = first ? a.x1 : a.x2;            // base makes hipcc close each block with s_waitcnt vmcnt(0)
    const size_t off = (size_t)b * (size_t)(first ? a.x1_bstride : a.x2_bstride) + (size_t)(first ? k : k - a.C1) * (size_t)a.P;
    return src + off;
}

__device__ __forceinline__ float4 ldv(const float* __restrict__ base, unsigned off) {
    return *reinterpret_cast<const float4*>(base + off);
}

// Store NCO accumulator tiles.  bias_l: LDS bias of the first tile; res: prefetched residual rows
// (RES) laid out [tile][r] as float4.
template <int NCO, bool RES>
__device__ __forceinline__ void epilogue(const Conv1x1Args& a, f32x4 (&acc)[NCO][4], int tfirst, int ntiles,
                                         const float* __restrict__ bias_l, const float4* res,
                                         int b, int p0, int kq, bool live) {
    const int P = a.P;
    float* outb = a.out + (size_t)b * a.out_bstride;
    if (a.mode == 0) {
        const unsigned voff = (unsigned)(4 * kq) * (unsigned)P + (unsigned)p0;   // channel 4kq of a tile
#pragma unroll
        for (int t = 0; t < NCO; ++t) {
            if (t >= ntiles) break;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cu = 16 * (tfirst + t) + r;                  // wave-uniform part of the channel
                const float bs = bias_l[16 * t + 4 * kq + r];
                float v[4] = {acc[t][0][r] + bs, acc[t][1][r] + bs, acc[t][2][r] + bs, acc[t][3][r] + bs};
                if constexpr (RES) {
                    const float4 rv = res[t * 4 + r];
                    v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                }
                if (a.act == 1 || a.act == 3) {
                    const float slope = a.act == 1 ? 0.2f : 0.1f;
#pragma unroll
                    for (int g = 0; g < 4; ++g) v[g] = v[g] > 0.f ? v[g] : slope * v[g];
                } else if (a.act == 2) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) v[g] = fmaxf(v[g], 0.f);
                } else if (a.act == 4) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) v[g] = fminf(fmaxf(v[g], 0.f), 1e4f);
                }
                if (live && cu + 4 * kq < a.Cout)
                    *reinterpret_cast<float4*>(outb + (size_t)cu * P + voff) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    } else {
        // ConvTranspose2d(k=2, s=2): GEMM row 4*o + 2*i + jj is output channel o at sub-position
        // (i, jj); the lane holds the whole 2x2 patch of each of its pixels for channel o.  The width only has to be
        // even: pixels (p0, p0+1) and (p0+2, p0+3) are each inside one row (p0 % 4 == 0), so every pixel PAIR is one
        // aligned 16-byte store per output row (at w = 266, level 3 of a 1424x2128 frame, the two pairs of a lane may
        // sit in different rows).
        const int w = a.w, w2 = 2 * a.w;
        const int ya = p0 / w, xa = p0 - ya * w;
        const int yb = (p0 + 2) / w, xb = (p0 + 2) - yb * w;
#pragma unroll
        for (int t = 0; t < NCO; ++t) {
            if (t >= ntiles) break;
            const int co = 16 * (tfirst + t) + 4 * kq;
            const int o = co >> 2;
            const float bs = bias_l[4 * t + kq];      // bias per real output channel o (staged as [NT*4])
            float* op = outb + (size_t)o * 4 * P;
            if (live && co < a.Cout) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    *reinterpret_cast<float4*>(op + (size_t)(2 * ya + i) * w2 + 2 * xa) =
                        make_float4(acc[t][0][2 * i] + bs, acc[t][0][2 * i + 1] + bs, acc[t][1][2 * i] + bs, acc[t][1][2 * i + 1] + bs);
                    *reinterpret_cast<float4*>(op + (size_t)(2 * yb + i) * w2 + 2 * xb) =
                        make_float4(acc[t][2][2 * i] + bs, acc[t][2][2 * i + 1] + bs, acc[t][3][2 * i] + bs, acc[t][3][2 * i + 1] + bs);
                }
            }
        }
    }
}

// bias (or zeros) for tiles [tbeg, tbeg + tpad) into LDS; mode 1 stores one value per output channel o
__device__ __forceinline__ void stage_bias(const Conv1x1Args& a, float* bias_l, int tbeg, int tpad, int tid) {
    if (a.mode == 0) {
        for (int i = tid; i < tpad * 16; i += 256) {
            const int co = 16 * tbeg + i;
            bias_l[i] = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
        }
    } else {
        for (int i = tid; i < tpad * 4; i += 256) {
            const int o = 4 * tbeg + i;
            bias_l[i] = (a.bias && 4 * o < a.Cout) ? a.bias[o] : 0.f;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Resident-input kernel.  KS = k-sets held in registers (K <= 4 * KS; missing k-sets are
// duplicates of the last one against zero weights).  A workgroup is persistent over the
// 256-pixel tiles of ONE image; its weight slice (output tiles [grp*ntw, grp*ntw+ntw), padded to
// a multiple of NCO = 2) sits in LDS as [KS][tpad][64] floats.  LDS map:
//   [0, KS*tpad*64) weights | tpad*16 bias | 4*KS gamma | 4*KS beta
// ---------------------------------------------------------------------------------------------
template <int KS, bool LN, int RT, bool DBUF>   // RT = residual tiles prefetched per workgroup (0, 2 or 4)
__global__ void __launch_bounds__(256, (KS <= 8 ? 3 : 2)) conv1x1_res_kernel(Conv1x1Args a, int ntw, int ngroups) {
    constexpr int NCO = 2;
    constexpr bool RES = RT > 0;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int grp = blockIdx.x % ngroups;
    const int wg = blockIdx.x / ngroups, nwg = gridDim.x / ngroups;
    const int b = blockIdx.y;
    const int P = a.P;
    const int K = a.C1 + a.C2;
    const int NS = K >> 2;
    const int NT = (a.Cout + 15) >> 4;
    const int tbeg = grp * ntw;
    const int tcnt = (NT - tbeg < ntw) ? NT - tbeg : ntw;
    const int tpad = (tcnt + NCO - 1) / NCO * NCO;
    const int ntiles = (P + 255) >> 8;
    float* lds_w = lds;
    float* bias_l = lds + KS * tpad * 64;
    float* gam_l = bias_l + tpad * 16;
    float* bet_l = gam_l + 4 * KS;

    // branch-free tile load: dead lanes (p0 >= P) read pixel 0, missing k-sets re-read the last one
    auto load_tile = [&](int tile, float4 (&dst)[KS]) {
        const int p0 = (tile * 4 + wave) * 64 + 4 * j;
        const unsigned voff = (unsigned)kq * (unsigned)P + (unsigned)(p0 < P ? p0 : 0);
#pragma unroll
        for (int s = 0; s < KS; ++s) dst[s] = ldv(kset_base(a, b, s < NS ? s : NS - 1), voff);
    };

    float4 xr[KS], xn[DBUF ? KS : 1];
    int tile = wg;
    load_tile(tile < ntiles ? tile : 0, xr);
    {   // weights, bias, LayerNorm affine -> LDS, once per workgroup (overlaps the first tile's loads)
        const float* wp = a.wp + (size_t)b * a.wp_bstride;
        const int n4 = KS * tpad * 16;
        for (int idx = tid; idx < n4; idx += 256) {
            const int l4 = idx & 15;
            const int t = (idx >> 4) % tpad;
            const int s = (idx >> 4) / tpad;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < tcnt && s < NS) v = *reinterpret_cast<const float4*>(wp + ((size_t)s * NT + tbeg + t) * 64 + l4 * 4);
            *reinterpret_cast<float4*>(lds_w + ((size_t)s * tpad + t) * 64 + l4 * 4) = v;
        }
        stage_bias(a, bias_l, tbeg, tpad, tid);
        if constexpr (LN) {
            for (int i = tid; i < 4 * KS; i += 256) {
                gam_l[i] = i < K ? a.ln_w[i] : 0.f;
                bet_l[i] = (a.ln_b && i < K) ? a.ln_b[i] : 0.f;
            }
        }
    }
    __syncthreads();

    for (; tile < ntiles; tile += nwg) {
        const int p0 = (tile * 4 + wave) * 64 + 4 * j;
        const bool live = p0 < P;
        const int tnext = tile + nwg;
        // ---- every global load this iteration needs is issued here, ahead of the tile's stores
        if constexpr (DBUF) {
            load_tile(tnext < ntiles ? tnext : tile, xn);
        }
        float4 res[RES ? RT * 4 : 1];   // residual rows of this workgroup's output tiles (Cout % 16 == 0 here)
        if constexpr (RES) {
            const float* resb = a.res + (size_t)b * a.res_bstride;
            const unsigned voff = (unsigned)(4 * kq) * (unsigned)P + (unsigned)(live ? p0 : 0);
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)   // tiles past tcnt re-read the last one: loads stay branch-free
                    res[t * 4 + r] = ldv(resb + (size_t)(16 * (tbeg + (t < tcnt ? t : tcnt - 1)) + r) * P, voff);
        }
        // ---- LayerNorm on the resident tile: exact two-pass statistics over the K channels
        if constexpr (LN) {
            float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s)
                if (s < NS) { sum[0] += xr[s].x; sum[1] += xr[s].y; sum[2] += xr[s].z; sum[3] += xr[s].w; }
            const float invK = 1.0f / (float)K;
            float mu[4], var[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                sum[g] += __shfl_xor(sum[g], 16);
                sum[g] += __shfl_xor(sum[g], 32);
                mu[g] = sum[g] * invK;
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s < NS) {
                    const float d0 = xr[s].x - mu[0], d1 = xr[s].y - mu[1], d2 = xr[s].z - mu[2], d3 = xr[s].w - mu[3];
                    var[0] = fmaf(d0, d0, var[0]); var[1] = fmaf(d1, d1, var[1]);
                    var[2] = fmaf(d2, d2, var[2]); var[3] = fmaf(d3, d3, var[3]);
                }
            }
            float rstd[4], sh[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                var[g] += __shfl_xor(var[g], 16);
                var[g] += __shfl_xor(var[g], 32);
                rstd[g] = 1.0f / sqrtf(var[g] * invK + a.ln_eps);
                sh[g] = a.ln_b ? mu[g] : 0.f;   // BiasFree_LayerNorm keeps the mean in the numerator
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const float gk = gam_l[4 * s + kq], bk = bet_l[4 * s + kq];
                xr[s].x = fmaf((xr[s].x - sh[0]) * rstd[0], gk, bk);
                xr[s].y = fmaf((xr[s].y - sh[1]) * rstd[1], gk, bk);
                xr[s].z = fmaf((xr[s].z - sh[2]) * rstd[2], gk, bk);
                xr[s].w = fmaf((xr[s].w - sh[3]) * rstd[3], gk, bk);
            }
        }
        // ---- sweep the output-channel groups over the resident tile
        auto do_group = [&](int tg, const float4* resp) {
            f32x4 acc[NCO][4];
#pragma unroll
            for (int t = 0; t < NCO; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[t][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* wl = lds_w + (size_t)tg * 64 + lane;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const float xb[4] = {xr[s].x, xr[s].y, xr[s].z, xr[s].w};
#pragma unroll
                for (int t = 0; t < NCO; ++t) {
                    const float av = wl[((size_t)s * tpad + t) * 64];
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        acc[t][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xb[g], acc[t][g], 0, 0, 0);
                }
            }
            epilogue<NCO, RES>(a, acc, tbeg + tg, tcnt - tg, bias_l + (a.mode == 0 ? 16 : 4) * tg, resp, b, p0, kq, live);
        };
        if constexpr (RES) {   // at most two groups, unrolled so the residual registers are indexed statically
            do_group(0, &res[0]);
            if constexpr (RT > NCO) {
                if (tpad > NCO) do_group(NCO, &res[NCO * 4]);
            }
        } else {
#pragma unroll 1
            for (int tg = 0; tg < tpad; tg += NCO) do_group(tg, res);
        }
        if constexpr (DBUF) {
#pragma unroll
            for (int s = 0; s < KS; ++s) xr[s] = xn[s];
        } else {
            if (tnext < ntiles) load_tile(tnext, xr);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Streaming kernel: accumulators for NCO tiles resident, K in chunks of KCH k-sets.
// ---------------------------------------------------------------------------------------------
static constexpr int kStreamLnMaxK = 1024;   // LayerNorm gamma/beta staged in LDS (RawFormer-L level 3 has K = 512)

template <int NCO, int KCH, bool LN>
__global__ void __launch_bounds__(256, 2) conv1x1_stream_kernel(Conv1x1Args a, int ngroups) {
    __shared__ __attribute__((aligned(16))) float lds_w[2][KCH * NCO * 64];
    __shared__ float bias_l[NCO * 16];
    __shared__ float gam_l[LN ? kStreamLnMaxK : 4], bet_l[LN ? kStreamLnMaxK : 4];
    constexpr int WPT = KCH * NCO * 16 / 256;   // float4 of weights each thread moves per chunk
    static_assert(KCH * NCO * 16 % 256 == 0, "weight chunk must split evenly over the workgroup");
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int grp = blockIdx.x % ngroups;
    const int tile = blockIdx.x / ngroups;
    const int b = blockIdx.y;
    const int P = a.P;
    const int K = a.C1 + a.C2;
    const int NS = K >> 2;
    const int NT = (a.Cout + 15) >> 4;
    const int t0 = grp * NCO;
    const int tcnt = (NT - t0 < NCO) ? NT - t0 : NCO;
    const int nch = (NS + KCH - 1) / KCH;
    const int p0 = (tile * 4 + wave) * 64 + 4 * j;
    const bool live = p0 < P;
    const unsigned voff = (unsigned)kq * (unsigned)P + (unsigned)(live ? p0 : 0);
    const float* wp = a.wp + (size_t)b * a.wp_bstride;
    stage_bias(a, bias_l, t0, NCO, tid);
    if constexpr (LN) {
        for (int i = tid; i < 4 * nch * KCH; i += 256) {          // zero beyond K: those k-sets meet zero weights anyway
            gam_l[i] = i < K ? a.ln_w[i] : 0.f;
            bet_l[i] = (a.ln_b && i < K) ? a.ln_b[i] : 0.f;
        }
    }

    // The residual initialises the accumulators: its loads travel with the first chunk's (one exposed
    // round trip instead of two) and the epilogue stays store-only.
    f32x4 acc[NCO][4];
#pragma unroll
    for (int t = 0; t < NCO; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[t][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.res && a.mode == 0) {   // (wave-uniform)
        // branch-free: a conditional block around each load would be closed with s_waitcnt vmcnt(0), i.e. 32
        // serialised round trips to HBM; rows that do not exist read row 0 of the image and are masked to zero
        const float* resb = a.res + (size_t)b * a.res_bstride;
        const unsigned vo = (unsigned)(live ? p0 : 0);
#pragma unroll
        for (int t = 0; t < NCO; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = 16 * (t0 + t) + 4 * kq + r;
                const bool ok = live && t < tcnt && co < a.Cout;
                const float4 rv = ldv(resb + (size_t)(ok ? co : 0) * P, vo);
                acc[t][0][r] = ok ? rv.x : 0.f; acc[t][1][r] = ok ? rv.y : 0.f;
                acc[t][2][r] = ok ? rv.z : 0.f; acc[t][3][r] = ok ? rv.w : 0.f;
            }
    }

    // LayerNorm statistics (shifted single pass; lanes kq = 0..3 split the channels)
    float lnA[4] = {1.f, 1.f, 1.f, 1.f}, lnB[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (LN) {
        const float4 s4 = ldv(kset_base(a, b, 0), (unsigned)(live ? p0 : 0));
        const float sh[4] = {s4.x, s4.y, s4.z, s4.w};
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 16
        for (int s = 0; s < NS; ++s) {
            const float4 t = ldv(kset_base(a, b, s), voff);
            const float d[4] = {t.x - sh[0], t.y - sh[1], t.z - sh[2], t.w - sh[3]};
#pragma unroll
            for (int g = 0; g < 4; ++g) { s1[g] += d[g]; s2[g] = fmaf(d[g], d[g], s2[g]); }
        }
        const float invK = 1.0f / (float)K;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            s1[g] += __shfl_xor(s1[g], 16); s1[g] += __shfl_xor(s1[g], 32);
            s2[g] += __shfl_xor(s2[g], 16); s2[g] += __shfl_xor(s2[g], 32);
            const float md = s1[g] * invK;
            const float var = fmaxf(fmaf(-md, md, s2[g] * invK), 0.f);
            const float rstd = 1.0f / sqrtf(var + a.ln_eps);
            lnA[g] = rstd;
            lnB[g] = a.ln_b ? -(sh[g] + md) * rstd : 0.f;
        }
    }

    float4 xa[KCH], xb_[KCH], wr[WPT];
    auto load_x_chunk = [&](int c, float4 (&dst)[KCH]) {
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            const int s = c * KCH + i;
            dst[i] = ldv(kset_base(a, b, s < NS ? s : NS - 1), voff);   // k-sets past K meet zero weights
        }
    };
    auto load_w_chunk = [&](int c) {
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int idx = tid + 256 * i;
            const int l4 = idx & 15, t = (idx >> 4) % NCO, s = c * KCH + (idx >> 4) / NCO;
            const bool ok = s < NS && t < tcnt;
            const float4 v = *reinterpret_cast<const float4*>(wp + ((size_t)(ok ? s : 0) * NT + t0 + (ok ? t : 0)) * 64 + l4 * 4);
            wr[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_w_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < WPT; ++i) *reinterpret_cast<float4*>(&lds_w[buf][(tid + 256 * i) * 4]) = wr[i];
    };
    // one chunk: the next chunk's x goes into the OTHER register set (no copies), its weights into the other LDS buffer
    auto chunk = [&](int c, float4 (&xc)[KCH], float4 (&xn)[KCH]) {
        if (c + 1 < nch) {   // next chunk in flight while this one computes
            load_x_chunk(c + 1, xn);
            load_w_chunk(c + 1);
        }
        const float* wl = &lds_w[c & 1][lane];
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            float xv[4] = {xc[i].x, xc[i].y, xc[i].z, xc[i].w};
            if constexpr (LN) {
                const int k = 4 * (c * KCH + i) + kq;
                const float gk = gam_l[k], bk = bet_l[k];
#pragma unroll
                for (int g = 0; g < 4; ++g) xv[g] = fmaf(fmaf(xv[g], lnA[g], lnB[g]), gk, bk);
            }
            // (reading the A operands one k-set ahead, pinned with sched_barrier, was measured: no gain -- the LDS
            // latency is already covered by the other wave of the SIMD)
#pragma unroll
            for (int t = 0; t < NCO; ++t) {
                const float av = wl[(i * NCO + t) * 64];
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    acc[t][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xv[g], acc[t][g], 0, 0, 0);
            }
        }
        if (c + 1 < nch) store_w_chunk((c + 1) & 1);
        __syncthreads();
    };

    load_x_chunk(0, xa);
    load_w_chunk(0);
    store_w_chunk(0);
    __syncthreads();
    for (int c = 0; c < nch; c += 2) {
        chunk(c, xa, xb_);
        if (c + 1 < nch) chunk(c + 1, xb_, xa);
    }
    epilogue<NCO, false>(a, acc, t0, tcnt, bias_l, nullptr, b, p0, kq, live);
}


// ---------------------------------------------------------------------------------------------
// b3 streaming kernel: the same GEMM on the bf16 matrix pipe with three-piece operands (rf_common.h).
// K is streamed in blocks of 32 channels = one v_mfma_f32_16x16x32_bf16 K extent.  Lane (j, kq) loads, for i = 0..7,
// one float4 = pixels p0 + 4j .. 4j+3 of channel 32c + 8kq + i; component g of those eight registers is -- after the
// split -- the B operand of pixel group g (the D layout, and with it the epilogue, is that of the f32 kernels).
// Per block and wave: 176 VALU for the split of 32 values x 4 pixels, 3 * NCO ds_read_b128 for the weight pieces,
// 24 * NCO MFMAs of 17 cycles (f32 form: 32 * NCO MFMAs of 33 cycles).  The next block's x is loaded into the registers
// the split has just vacated, its weights go registers -> LDS behind the MFMAs; one barrier per block.
// ---------------------------------------------------------------------------------------------
// channels 32c + 8kq + i (i = 0..7) of K block c, 4 pixels each: the block lies in ONE of up to three sources (launch_conv1x1
// checks that they are cut at multiples of 32 channels); selects, not branches (see kset_base)
__device__ __forceinline__ void b3_load_x_block(float4 (&xr)[8], const float* xb1, const float* xb2, const float* xb3, int C1, int C2, int C3,
                                                int c, int kq, unsigned P, unsigned pl) {
    const int cb = 32 * c;
    const bool first = cb < C1, third = cb >= C1 + C2;
    const float* src = first ? xb1 : third ? xb3 : xb2;
    const int cloc = first ? cb : third ? cb - C1 - C2 : cb - C1, cmax = (first ? C1 : third ? C3 : C2) - 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int ch = cloc + 8 * kq + i;
        ch = ch < cmax ? ch : cmax;
        xr[i] = ldv(src, (unsigned)ch * P + pl);
    }
}

template <int NCO, bool LN>
__global__ void __launch_bounds__(256, 2) conv1x1_b3_kernel(Conv1x1Args a, int ngroups) {
    __shared__ __attribute__((aligned(16))) u32x4 lds_w[2][NCO * 3 * 64];
    __shared__ float bias_l[NCO * 16];
    __shared__ float gam_l[LN ? kStreamLnMaxK : 4], bet_l[LN ? kStreamLnMaxK : 4];
    constexpr int WPT = (NCO * 3 * 64 + 255) / 256;   // 16-byte weight elements each thread moves per block
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int grp = blockIdx.x % ngroups;
    const int tile = blockIdx.x / ngroups;
    const int b = blockIdx.y;
    const int P = a.P;
    const int K = a.C1 + a.C2 + a.C3;
    const int NB = (K + 31) >> 5;
    const int NT = (a.Cout + 15) >> 4;
    const int t0 = grp * NCO;
    const int tcnt = (NT - t0 < NCO) ? NT - t0 : NCO;
    const int p0 = (tile * 4 + wave) * 64 + 4 * j;
    const bool live = p0 < P;
    const unsigned pl = (unsigned)(live ? p0 : 0);
    const u32x4* wp3 = reinterpret_cast<const u32x4*>(reinterpret_cast<const float*>(a.wp3) + (size_t)b * a.wp3_bstride);
    stage_bias(a, bias_l, t0, NCO, tid);
    if constexpr (LN) {
        for (int i = tid; i < 32 * NB; i += 256) {
            gam_l[i] = i < K ? a.ln_w[i] : 0.f;
            bet_l[i] = (a.ln_b && i < K) ? a.ln_b[i] : 0.f;
        }
    }

    f32x4 acc[NCO][4];
#pragma unroll
    for (int t = 0; t < NCO; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[t][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.res && a.mode == 0) {   // the residual initialises the accumulators (branch-free loads, see conv1x1_stream_kernel)
        const float* resb = a.res + (size_t)b * a.res_bstride;
#pragma unroll
        for (int t = 0; t < NCO; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = 16 * (t0 + t) + 4 * kq + r;
                const bool ok = live && t < tcnt && co < a.Cout;
                const float4 rv = ldv(resb + (size_t)(ok ? co : 0) * P, pl);
                acc[t][0][r] = ok ? rv.x : 0.f; acc[t][1][r] = ok ? rv.y : 0.f;
                acc[t][2][r] = ok ? rv.z : 0.f; acc[t][3][r] = ok ? rv.w : 0.f;
            }
    }

    // LayerNorm statistics (shifted single pass; lanes kq = 0..3 split the channels)
    float lnA[4] = {1.f, 1.f, 1.f, 1.f}, lnB[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (LN) {
        const unsigned voff = (unsigned)kq * (unsigned)P + pl;
        const float4 s4 = ldv(kset_base(a, b, 0), pl);
        const float sh[4] = {s4.x, s4.y, s4.z, s4.w};
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 16
        for (int s = 0; s < (K >> 2); ++s) {
            const float4 t = ldv(kset_base(a, b, s), voff);
            const float d[4] = {t.x - sh[0], t.y - sh[1], t.z - sh[2], t.w - sh[3]};
#pragma unroll
            for (int g = 0; g < 4; ++g) { s1[g] += d[g]; s2[g] = fmaf(d[g], d[g], s2[g]); }
        }
        const float invK = 1.0f / (float)K;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            s1[g] += __shfl_xor(s1[g], 16); s1[g] += __shfl_xor(s1[g], 32);
            s2[g] += __shfl_xor(s2[g], 16); s2[g] += __shfl_xor(s2[g], 32);
            const float md = s1[g] * invK;
            const float var = fmaxf(fmaf(-md, md, s2[g] * invK), 0.f);
            const float rstd = 1.0f / sqrtf(var + a.ln_eps);
            lnA[g] = rstd;
            lnB[g] = a.ln_b ? -(sh[g] + md) * rstd : 0.f;
        }
    }

    float4 xr[8];
    u32x4 wr[WPT];
    // channel 32c + 8kq + i of block c: sources are cut at multiples of 32 channels (launch_conv1x1 checks), channels past K
    // re-read the last one (their weights are zero)
    // (the sources are passed BY VALUE to a function: a conditional between two variables captured by a lambda is a conditional
    // between two addresses of its closure, and hipcc then keeps the closure -- and the xr registers it refers to -- in scratch)
    const float* const xb1 = a.x1 + (size_t)b * (size_t)a.x1_bstride;
    const float* const xb2 = a.x2 + (size_t)b * (size_t)a.x2_bstride;      // only dereferenced when C2 / C3 > 0
    const float* const xb3 = a.x3 + (size_t)b * (size_t)a.x3_bstride;
    auto load_x_block = [&](int c) { b3_load_x_block(xr, xb1, xb2, xb3, a.C1, a.C2, a.C3, c, kq, (unsigned)P, pl); };
    auto load_w_block = [&](int c) {
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int idx = tid + 256 * i;                       // (t, piece, lane) of this workgroup's slice
            const int t = idx / 192, rem = idx - t * 192;
            const bool ok = idx < NCO * 192 && t < tcnt;
            const u32x4 v = wp3[((size_t)c * NT + t0 + (ok ? t : 0)) * 192 + (ok ? rem : 0)];
            wr[i] = ok ? v : (u32x4){0u, 0u, 0u, 0u};
        }
    };
    auto store_w_block = [&](int buf) {
#pragma unroll
        for (int i = 0; i < WPT; ++i)
            if (tid + 256 * i < NCO * 192) lds_w[buf][tid + 256 * i] = wr[i];
    };

    load_x_block(0);
    load_w_block(0);
    store_w_block(0);
    __syncthreads();
    for (int c = 0; c < NB; ++c) {
        // ---- LayerNorm + three-piece split of this block's 8 channels x 4 pixels
        u32x4 bp[4][3];
#pragma unroll
        for (int hp = 0; hp < 4; ++hp) {          // channel pair (2 hp, 2 hp + 1) -> dword hp of every operand
            float xa[4] = {xr[2 * hp].x, xr[2 * hp].y, xr[2 * hp].z, xr[2 * hp].w};
            float xb[4] = {xr[2 * hp + 1].x, xr[2 * hp + 1].y, xr[2 * hp + 1].z, xr[2 * hp + 1].w};
            if constexpr (LN) {
                const int k = 32 * c + 8 * kq + 2 * hp;
                const float ga = gam_l[k], ba = bet_l[k], gb = gam_l[k + 1], bb = bet_l[k + 1];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    xa[g] = fmaf(fmaf(xa[g], lnA[g], lnB[g]), ga, ba);
                    xb[g] = fmaf(fmaf(xb[g], lnA[g], lnB[g]), gb, bb);
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned a0, a1, a2, b0, b1, b2;
                b3_split(xa[g], a0, a1, a2);
                b3_split(xb[g], b0, b1, b2);
                bp[g][0][hp] = b3_pack(a0, b0);
                bp[g][1][hp] = b3_pack(a1, b1);
                bp[g][2][hp] = b3_pack(a2, b2);
            }
        }
        if (c + 1 < NB) {   // next block in flight while this one is in the matrix pipe (x into the registers just vacated)
            load_x_block(c + 1);
            load_w_block(c + 1);
        }
        const u32x4* wl = &lds_w[c & 1][lane];
#pragma unroll
        for (int t = 0; t < NCO; ++t) {
            const u32x4 ap[3] = {wl[(t * 3 + 0) * 64], wl[(t * 3 + 1) * 64], wl[(t * 3 + 2) * 64]};
            b3_mfma4(ap, bp, acc[t]);
        }
        if (c + 1 < NB) store_w_block((c + 1) & 1);
        __syncthreads();
    }
    epilogue<NCO, false>(a, acc, t0, tcnt, bias_l, nullptr, b, p0, kq, live);
}

// ---------------------------------------------------------------------------------------------
// LayerNorm + 1x1 at levels 2-3 (K = 128 / 256, Cout = 2K or 3K): the input tile is normalised and split ONCE.
//   conv1x1_b3_kernel gives every workgroup 4-6 output tiles and re-reads, re-normalises and re-splits its 256 pixels for
//   each of the Cout / 64 output groups (6 passes over x for the level-2 qkv).  Here a workgroup owns 64 pixels and ALL
//   output tiles: the four waves load K / 4 channels each, reduce the two-pass statistics through LDS, write the three
//   bf16 pieces of the normalised tile to LDS in B-operand order ([K block][pixel q][piece][lane], 12 KB per 32 channels),
//   and then split the OUTPUT tiles: wave w computes tiles [w tpw, (w + 1) tpw) in chunks of NCO, reading the shared B
//   pieces from LDS (once per chunk) and its A pieces straight from L2 (each weight element is used by exactly one wave of
//   the workgroup), double-buffered one K block ahead.
// ---------------------------------------------------------------------------------------------
template <int KB, int NCO>
__global__ void __launch_bounds__(256, KB <= 4 ? 2 : 1) conv1x1_b3_ln_kernel(Conv1x1Args a, int tpw) {
    __shared__ __attribute__((aligned(16))) u32x4 Bl[KB * 768];
    __shared__ __attribute__((aligned(16))) float red[2 * 4 * 64];
    __shared__ float bias_l[1024];
    constexpr int KPW = KB >= 4 ? KB / 4 : 1; // K blocks each wave loads ...
    constexpr int CPT = KB >= 4 ? 8 : 4;      // ... and channels of a block per lane: K = 64 splits each block over two waves
    constexpr int K = 32 * KB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int b = blockIdx.y, P = a.P;
    const int NT = a.Cout >> 4;
    const int ntile = (P + 63) >> 6;
    const float* xb = a.x1 + (size_t)b * a.x1_bstride;
    const u32x4* wp3 = reinterpret_cast<const u32x4*>(reinterpret_cast<const float*>(a.wp3) + (size_t)b * a.wp3_bstride) + lane;
    for (int i = tid; i < a.Cout; i += 256) bias_l[i] = a.bias ? a.bias[i] : 0.f;

    // ---- input tile: wave w holds K blocks [w KPW, (w + 1) KPW), lane (j, kq) channels 8 kq .. 8 kq + 7 of each, 4 pixels.
    // Persistent over pixel tiles: the next tile's loads are issued as soon as this one is split, i.e. behind its whole GEMM
    // (without that the kernel ran at HBM time + MFMA time: all workgroups load, then all multiply).
    const int kb0 = KB >= 4 ? wave * KPW : wave >> 1;                     // this wave's first K block
    const int cbase = KB >= 4 ? 8 * kq : 8 * kq + 4 * (wave & 1);         // first of the lane's CPT channels inside a block
    float4 xr[KPW][CPT];
    auto load_x = [&](int tile) {
        const int q0 = tile * 64 + 4 * j;
        const unsigned ql = (unsigned)(q0 < P ? q0 : 0);
#pragma unroll
        for (int i = 0; i < KPW; ++i)
#pragma unroll
            for (int c = 0; c < CPT; ++c) xr[i][c] = ldv(xb, (unsigned)(32 * (kb0 + i) + cbase + c) * (unsigned)P + ql);
    };
    constexpr bool PREF = KB != 4;         // K = 128 runs two workgroups per CU with 48 B-piece registers: none left for a tile in flight
    if (PREF) load_x(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    if (!PREF) load_x(tile);
    const int p0 = tile * 64 + 4 * j;
    const bool live = p0 < P;
    // two-pass statistics: lanes, then waves (fixed order)
    float mean[4], rstd[4];
    {
        float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < KPW; ++i)
#pragma unroll
            for (int c = 0; c < CPT; ++c) { s[0] += xr[i][c].x; s[1] += xr[i][c].y; s[2] += xr[i][c].z; s[3] += xr[i][c].w; }
#pragma unroll
        for (int g = 0; g < 4; ++g) { s[g] += __shfl_xor(s[g], 16); s[g] += __shfl_xor(s[g], 32); }
        if (kq == 0) *reinterpret_cast<float4*>(red + wave * 64 + 4 * j) = make_float4(s[0], s[1], s[2], s[3]);
        __syncthreads();
        const float4 r0 = *reinterpret_cast<const float4*>(red + 4 * j), r1 = *reinterpret_cast<const float4*>(red + 64 + 4 * j);
        const float4 r2 = *reinterpret_cast<const float4*>(red + 128 + 4 * j), r3 = *reinterpret_cast<const float4*>(red + 192 + 4 * j);
        mean[0] = (((r0.x + r1.x) + r2.x) + r3.x) * (1.0f / K); mean[1] = (((r0.y + r1.y) + r2.y) + r3.y) * (1.0f / K);
        mean[2] = (((r0.z + r1.z) + r2.z) + r3.z) * (1.0f / K); mean[3] = (((r0.w + r1.w) + r2.w) + r3.w) * (1.0f / K);
        float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < KPW; ++i)
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                const float d0 = xr[i][c].x - mean[0], d1 = xr[i][c].y - mean[1], d2 = xr[i][c].z - mean[2], d3 = xr[i][c].w - mean[3];
                v[0] = fmaf(d0, d0, v[0]); v[1] = fmaf(d1, d1, v[1]); v[2] = fmaf(d2, d2, v[2]); v[3] = fmaf(d3, d3, v[3]);
            }
#pragma unroll
        for (int g = 0; g < 4; ++g) { v[g] += __shfl_xor(v[g], 16); v[g] += __shfl_xor(v[g], 32); }
        if (kq == 0) *reinterpret_cast<float4*>(red + 256 + wave * 64 + 4 * j) = make_float4(v[0], v[1], v[2], v[3]);
        __syncthreads();
        const float4 q0 = *reinterpret_cast<const float4*>(red + 256 + 4 * j), q1 = *reinterpret_cast<const float4*>(red + 320 + 4 * j);
        const float4 q2 = *reinterpret_cast<const float4*>(red + 384 + 4 * j), q3 = *reinterpret_cast<const float4*>(red + 448 + 4 * j);
        rstd[0] = 1.0f / sqrtf((((q0.x + q1.x) + q2.x) + q3.x) * (1.0f / K) + a.ln_eps);
        rstd[1] = 1.0f / sqrtf((((q0.y + q1.y) + q2.y) + q3.y) * (1.0f / K) + a.ln_eps);
        rstd[2] = 1.0f / sqrtf((((q0.z + q1.z) + q2.z) + q3.z) * (1.0f / K) + a.ln_eps);
        rstd[3] = 1.0f / sqrtf((((q0.w + q1.w) + q2.w) + q3.w) * (1.0f / K) + a.ln_eps);
    }
    // normalise, split into three bf16 pieces, publish in B-operand order
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        const int kb = kb0 + i;
        unsigned bpd[4][3][CPT / 2];              // [pixel][piece][channel pair]
#pragma unroll
        for (int hp = 0; hp < CPT / 2; ++hp) {
            const int k = 32 * kb + cbase + 2 * hp;
            const float ga = a.ln_w[k], gb = a.ln_w[k + 1];
            const float ba = a.ln_b ? a.ln_b[k] : 0.f, bb = a.ln_b ? a.ln_b[k + 1] : 0.f;
            const float xa[4] = {xr[i][2 * hp].x, xr[i][2 * hp].y, xr[i][2 * hp].z, xr[i][2 * hp].w};
            const float xc[4] = {xr[i][2 * hp + 1].x, xr[i][2 * hp + 1].y, xr[i][2 * hp + 1].z, xr[i][2 * hp + 1].w};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float mu = a.ln_b ? mean[g] : 0.f;     // BiasFree_LayerNorm scales x, not x - mean
                const float ya = fmaf((xa[g] - mu) * rstd[g], ga, ba), yb = fmaf((xc[g] - mu) * rstd[g], gb, bb);
                unsigned a0, a1, a2, b0, b1, b2;
                b3_split(ya, a0, a1, a2);
                b3_split(yb, b0, b1, b2);
                bpd[g][0][hp] = b3_pack(a0, b0);
                bpd[g][1][hp] = b3_pack(a1, b1);
                bpd[g][2][hp] = b3_pack(a2, b2);
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
                u32x4* dst = &Bl[((kb * 4 + g) * 3 + pc) * 64 + lane];
                if constexpr (CPT == 8) *dst = (u32x4){bpd[g][pc][0], bpd[g][pc][1], bpd[g][pc][2], bpd[g][pc][3]};
                else reinterpret_cast<uint2*>(dst)[wave & 1] = make_uint2(bpd[g][pc][0], bpd[g][pc][1]);   // this wave's half of the element
            }
    }
    __syncthreads();
    if (PREF && tile + (int)gridDim.x < ntile) load_x(tile + gridDim.x);

    // ---- output tiles of this wave, NCO at a time
    for (int chunk = 0; chunk * NCO < tpw; ++chunk) {
        const int t0 = ((int)blockIdx.z * 4 + wave) * tpw + chunk * NCO;    // blockIdx.z: output-channel split of small launches
        f32x4 acc[NCO][4];
#pragma unroll
        for (int t = 0; t < NCO; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[t][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // A pieces straight from L2, in a ring of RING (K block, tile) steps: a step's three 16-byte loads are issued
        // RING - 1 steps (x 24 MFMAs = 408 cycles each) before their use.
        constexpr int STEPS = KB * NCO, RING = KB > 4 ? 12 : (STEPS < 5 ? STEPS : 5);
        u32x4 A[RING][3];
        auto load_a = [&](int slot, int step) {
            const int kb = step / NCO, t = step - kb * NCO;
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) A[slot][pc] = wp3[((size_t)kb * NT + t0 + t) * 192 + pc * 64];
        };
#pragma unroll
        for (int i = 0; i < RING - 1; ++i) load_a(i, i);
        // B pieces from LDS, double-buffered per K block: the 12 reads of block kb + 1 are spread over the NCO steps of block kb
        u32x4 bp[2][4][3];
        auto load_b = [&](int buf, int kb, int first, int last) {
#pragma unroll
            for (int i = first; i < last; ++i) bp[buf][i / 3][i % 3] = Bl[(kb * 12 + i) * 64 + lane];
        };
        load_b(0, 0, 0, 12);
#pragma unroll
        for (int step = 0; step < STEPS; ++step) {
            const int kb = step / NCO, t = step % NCO;
            if (step + RING - 1 < STEPS) load_a((step + RING - 1) % RING, step + RING - 1);
            if (kb + 1 < KB) load_b((kb + 1) & 1, kb + 1, 12 * t / NCO, 12 * (t + 1) / NCO);
            b3_mfma4(A[step % RING], bp[kb & 1], acc[t]);
            __builtin_amdgcn_sched_barrier(0);      // keep the ring: without it hipcc hoists every load to the top and spills
        }
        epilogue<NCO, false>(a, acc, t0, NCO, bias_l + 16 * t0, nullptr, b, p0, kq, live);
    }
    }   // pixel tiles (the first barrier of the next tile's statistics also frees Bl)
}

static bool b3_ln_supported(const Conv1x1Args& a, int* nco) {
    const int K = a.C1 + a.C2;
    if (!(a.ln_w && a.wp3 && a.C2 == 0 && (K == 64 || K == 128 || K == 256) && a.mode == 0 && !a.res && a.Cout % 64 == 0 && a.Cout <= 1024 &&
          aligned16(a.wp3) && a.wp3_bstride % 4 == 0))
        return false;
    const int tpw = a.Cout / 64;
    // K = 128 runs two workgroups per CU (256 registers): 4 tiles at once plus the next pixel tile in flight would spill
    *nco = (K == 256 && tpw % 4 == 0) ? 4 : tpw % 3 == 0 ? 3 : tpw % 2 == 0 ? 2 : 0;
    return *nco != 0;
}

// ---------------------------------------------------------------------------------------------
// Ragged shapes (P % 4 != 0, unaligned views): one thread per output pixel and channel, plain
// FMA over K with a two-pass LayerNorm.  Correctness path only: tiny odd test frames (a RawFormer level always has
// an even width and a pixel count divisible by 4 except at 1-2 pixel extents).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) conv1x1_scalar_kernel(Conv1x1Args a) {
    const int P = a.P, K = a.C1 + a.C2, NT = (a.Cout + 15) >> 4;
    const int b = blockIdx.y;
    const float* wp = a.wp + (size_t)b * a.wp_bstride;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < P; p += gridDim.x * 256) {
        auto xat = [&](int k) {
            return (k < a.C1) ? a.x1[(size_t)b * a.x1_bstride + (size_t)k * P + p]
                              : a.x2[(size_t)b * a.x2_bstride + (size_t)(k - a.C1) * P + p];
        };
        float mu = 0.f, rstd = 1.f;
        if (a.ln_w) {
            for (int k = 0; k < K; ++k) mu += xat(k);
            mu /= (float)K;
            float var = 0.f;
            for (int k = 0; k < K; ++k) { const float d = xat(k) - mu; var = fmaf(d, d, var); }
            rstd = 1.0f / sqrtf(var / (float)K + a.ln_eps);
            if (!a.ln_b) mu = 0.f;
        }
        for (int co = 0; co < a.Cout; ++co) {
            float s = 0.f;
            for (int k = 0; k < K; ++k) {
                float xv = xat(k);
                if (a.ln_w) xv = fmaf((xv - mu) * rstd, a.ln_w[k], a.ln_b ? a.ln_b[k] : 0.f);
                s = fmaf(wp[((size_t)(k >> 2) * NT + (co >> 4)) * 64 + (co & 15) + 16 * (k & 3)], xv, s);
            }
            if (a.mode == 0) {
                s += a.bias ? a.bias[co] : 0.f;
                if (a.res) s += a.res[(size_t)b * a.res_bstride + (size_t)co * P + p];
                if (a.act == 1) s = s > 0.f ? s : 0.2f * s;
                else if (a.act == 2) s = fmaxf(s, 0.f);
                else if (a.act == 3) s = s > 0.f ? s : 0.1f * s;
                else if (a.act == 4) s = fminf(fmaxf(s, 0.f), 1e4f);
                a.out[(size_t)b * a.out_bstride + (size_t)co * P + p] = s;
            } else {
                const int o = co >> 2, y = p / a.w, x = p - y * a.w;
                s += a.bias ? a.bias[o] : 0.f;
                a.out[(size_t)b * a.out_bstride + (size_t)o * 4 * P + (size_t)(2 * y + ((co >> 1) & 1)) * (2 * a.w) + 2 * x + (co & 1)] = s;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
template <int KS, int RT>
static void launch_res_rt(const Conv1x1Args& a, int ntw, int ngroups, dim3 grid, size_t lds, hipStream_t st) {
    constexpr bool DBUF = KS <= 16 && !(KS > 8 && RT == 4);
    if (a.ln_w) conv1x1_res_kernel<KS, true, RT, DBUF><<<grid, 256, lds, st>>>(a, ntw, ngroups);
    else conv1x1_res_kernel<KS, false, RT, DBUF><<<grid, 256, lds, st>>>(a, ntw, ngroups);
}

template <int KS>
static void launch_res(const Conv1x1Args& a, int ntw, int ngroups, dim3 grid, size_t lds, hipStream_t st) {
    if constexpr (KS <= 16) {
        if (a.res && ntw > 2) launch_res_rt<KS, 4>(a, ntw, ngroups, grid, lds, st);
        else if (a.res) launch_res_rt<KS, 2>(a, ntw, ngroups, grid, lds, st);
        else launch_res_rt<KS, 0>(a, ntw, ngroups, grid, lds, st);
    } else {
        conv1x1_res_kernel<KS, false, 0, false><<<grid, 256, lds, st>>>(a, ntw, ngroups);
    }
}

// shapes the b3 kernel takes: K-blocks of 32 channels must not straddle the two sources; K < 128 stays on the resident-input
// f32 kernels, which read x once whatever Cout is and are HBM-bound there (measured per shape with tools/kbench.py:
// b3 wins 1.05-1.4 x from K = 128 up, loses 0.75-0.95 x at K = 64)
static bool b3_supported(const Conv1x1Args& a) {
    const int K = a.C1 + a.C2 + a.C3;
    return K >= 128 && a.C1 % 8 == 0 && a.C2 % 8 == 0 && a.C3 % 8 == 0 && (a.C2 + a.C3 == 0 || a.C1 % 32 == 0) &&
           (a.C3 == 0 || a.C2 % 32 == 0) && (!a.ln_w || (K <= kStreamLnMaxK && a.C3 == 0)) && aligned16(a.wp3) && (a.wp3_bstride % 4 == 0);
}

// Does the LayerNorm prologue of this GEMM read x once?  Yes on the split-once bf16x3 kernel (K = 64 / 128 / 256) and on the
// resident-input kernels (K <= 64).  The streaming kernels re-read and re-normalise their pixels once per group of 64-96 output
// channels (RawFormer-L level 3, K = 512 -> 1536: 24 passes, 0.77 ms for a 0.2 ms product; every level of RawFormer-B, K = 96 /
// 192 / 384): their callers run layernorm2d first and the plain GEMM on its output (rf_block.hip).
bool conv1x1_ln_single_pass(const Conv1x1Args& a) {
    int nco = 0;
    return !a.ln_w || a.C1 + a.C2 <= 64 || (a.wp3 != nullptr && b3_ln_supported(a, &nco));
}

int launch_conv1x1(const Conv1x1Args& a, hipStream_t st) {
    RF_CHECK_ARG(a.B > 0 && a.P > 0 && a.C1 > 0 && a.C2 >= 0 && a.Cout > 0, "conv1x1: bad sizes B=%d P=%d C1=%d C2=%d Cout=%d",
                 a.B, a.P, a.C1, a.C2, a.Cout);
    RF_CHECK_ARG(a.C1 % 4 == 0 && a.C2 % 4 == 0, "conv1x1: input channel counts (%d, %d) must be multiples of 4", a.C1, a.C2);
    RF_CHECK_ARG(a.C2 == 0 || a.x2 != nullptr, "conv1x1: second source missing");
    RF_CHECK_ARG(a.C3 >= 0 && a.C3 % 4 == 0 && (a.C3 == 0 || (a.x3 != nullptr && a.C2 > 0)), "conv1x1: bad third source (C3 = %d)", a.C3);
    RF_CHECK_ARG(a.mode == 0 || (a.Cout % 4 == 0 && a.w > 0 && a.P % a.w == 0 && !a.res), "conv1x1: bad ConvTranspose geometry");
    RF_CHECK_ARG(a.B <= 65535 && (double)a.P * 4.0 * 16.0 < 4.0e9, "conv1x1: batch %d / plane %d too large", a.B, a.P);
    const int K = a.C1 + a.C2 + a.C3;
    const int NS = K / 4, NT = cdiv(a.Cout, 16);
    bool vec = (a.P % 4 == 0) && aligned16(a.x1) && aligned16(a.out) && (a.x1_bstride % 4 == 0) && (a.out_bstride % 4 == 0);
    if (a.x2) vec = vec && aligned16(a.x2) && (a.x2_bstride % 4 == 0);
    if (a.C3) vec = vec && aligned16(a.x3) && (a.x3_bstride % 4 == 0);
    if (a.res) vec = vec && aligned16(a.res) && (a.res_bstride % 4 == 0);
    if (a.mode == 1) vec = vec && (a.w % 2 == 0);
    const double px = (double)a.B * a.P;
    const double work_flops = 2.0 * K * a.Cout * px, work_bytes = 4.0 * px * (K + a.Cout + (a.res ? a.Cout : 0));
    char key[64];
    bool b3_ok = a.wp3 != nullptr;
#ifdef RF_DIAG   // diagnostic build only (build.py --diag): force the f32 MFMA kernels
    if (getenv("RF_NO_B3")) b3_ok = false;
#endif
    const bool use_b3 = b3_ok && b3_supported(a);
    // three sources exist for the bf16x3 streaming kernel only (its caller checks the shape; there is no slower form to fall to)
    RF_CHECK_ARG(a.C3 == 0 || (use_b3 && vec && !a.ln_w && !(a.res && a.Cout % 16 != 0)), "conv1x1: three sources need the bf16x3 kernel (K = %d, P = %d)", K, a.P);
    if (!vec || (a.res && a.Cout % 16 != 0) || (a.ln_w && K > kStreamLnMaxK)) {
        ProfScope prof(st, "conv1x1_scalar_kernel", work_flops, work_bytes);
        int gx = cdiv(a.P, 256);
        if (gx > 4096) gx = 4096;
        conv1x1_scalar_kernel<<<dim3((unsigned)gx, (unsigned)a.B), 256, 0, st>>>(a);
    } else if (int nco_ln = 0; b3_ok && b3_ln_supported(a, &nco_ln)) {
        const int kb = K / 32;
        int tpw = a.Cout / 64;
        const int slots = (kb <= 4 ? 2 : 1) * 256;                // resident workgroups (LDS: 31 / 55 / 104 KB; registers: 2 per CU)
        int gx = slots / a.B;
        gx = gx < 1 ? 1 : gx;
        gx = gx < cdiv(a.P, 64) ? gx : cdiv(a.P, 64);
        // one small frame (levels 2-3: 4-64 pixel tiles): a workgroup's waves walk tpw / NCO chunks of output tiles one after the
        // other, each a chain of K-block steps -- split the output channels over blockIdx.z (every workgroup normalises its
        // 64 pixels again: K x 64 values) until the launch has a few hundred workgroups.  Same arithmetic per output: same bits.
        int zs = 1;
        if ((long)gx * a.B < 128)
            for (int z = tpw / nco_ln; z > 1; --z)
                if ((tpw / nco_ln) % z == 0 && (long)gx * a.B * z <= 512) { zs = z; break; }
        tpw /= zs;
        dim3 grid((unsigned)gx, (unsigned)a.B, (unsigned)zs);
        snprintf(key, sizeof(key), "conv1x1_b3_ln_kernel<%d, %d>", kb, nco_ln);
        ProfScope prof(st, key, work_flops, work_bytes);
        if (kb == 2) {
            if (nco_ln == 3) conv1x1_b3_ln_kernel<2, 3><<<grid, 256, 0, st>>>(a, tpw);
            else conv1x1_b3_ln_kernel<2, 2><<<grid, 256, 0, st>>>(a, tpw);
        } else if (kb == 4) {
            if (nco_ln == 3) conv1x1_b3_ln_kernel<4, 3><<<grid, 256, 0, st>>>(a, tpw);
            else conv1x1_b3_ln_kernel<4, 2><<<grid, 256, 0, st>>>(a, tpw);
        } else {
            if (nco_ln == 4) conv1x1_b3_ln_kernel<8, 4><<<grid, 256, 0, st>>>(a, tpw);
            else if (nco_ln == 3) conv1x1_b3_ln_kernel<8, 3><<<grid, 256, 0, st>>>(a, tpw);
            else conv1x1_b3_ln_kernel<8, 2><<<grid, 256, 0, st>>>(a, tpw);
        }
    } else if (use_b3) {
        // bf16x3 streaming kernel (K > 32: below that the resident-input f32 kernels are HBM-bound anyway)
        // 6 tiles per workgroup where that divides the output evenly; never with the LayerNorm prologue (that instantiation
        // needs 2 registers more than the 256 a wave has at two waves per SIMD)
        int nco = (!a.ln_w && (NT % 6 == 0 || (NT % 4 != 0 && NT > 8))) ? 6 : 4;
        // one frame at levels 2-3: fewer than 256 workgroups -- two output tiles per workgroup instead of four fills more CUs
        if (!a.ln_w && nco == 4 && NT % 2 == 0 && (long)cdiv(a.P, 256) * cdiv(NT, 4) * a.B < 256) nco = 2;
        const int ngroups = cdiv(NT, nco);
        dim3 grid((unsigned)(cdiv(a.P, 256) * ngroups), (unsigned)a.B, 1);
        snprintf(key, sizeof(key), "conv1x1_b3_kernel<%d, %s>", nco, a.ln_w ? "true" : "false");
        ProfScope prof(st, key, work_flops, work_bytes);
        if (nco == 6) {
            conv1x1_b3_kernel<6, false><<<grid, 256, 0, st>>>(a, ngroups);
        } else if (nco == 2) {
            conv1x1_b3_kernel<2, false><<<grid, 256, 0, st>>>(a, ngroups);
        } else {
            if (a.ln_w) conv1x1_b3_kernel<4, true><<<grid, 256, 0, st>>>(a, ngroups);
            else conv1x1_b3_kernel<4, false><<<grid, 256, 0, st>>>(a, ngroups);
        }
    } else if (NS <= 16 || (NS <= 32 && !a.ln_w && !a.res)) {
        const int ks = NS <= 4 ? 4 : NS <= 8 ? 8 : NS <= 12 ? 12 : NS <= 16 ? 16 : NS <= 24 ? 24 : 32;
        // output tiles per workgroup: weight slice <= ~60 KB, and <= 8 tiles when the residual rows
        // are prefetched into registers
        int cap = (240 / ks) & ~1;
        if (a.res && cap > 4) cap = 4;
        // the four-residual-tile instantiations with K <= 32 or a LayerNorm prologue spill (28-296 bytes of scratch) next to
        // their prefetched loads: such shapes (no layer of RawFormer; reachable through rf_conv1x1) take two tiles per workgroup
        if (a.res && (ks <= 8 || a.ln_w) && cap > 2) cap = 2;
        int ntw = NT < cap ? NT : cap;
        const int ngroups = cdiv(NT, ntw);
        const int tpad = cdiv(ntw, 2) * 2;
        const size_t lds = ((size_t)ks * tpad * 64 + tpad * 16 + 8 * ks) * sizeof(float);
        // persistent workgroups: enough to fill every CU at this kernel's occupancy, split over the images
        const int slots = 256 * (ks <= 8 ? 3 : 2);
        int wgs = cdiv(slots, a.B * ngroups);
        if (wgs > cdiv(a.P, 256)) wgs = cdiv(a.P, 256);
        dim3 grid((unsigned)(wgs * ngroups), (unsigned)a.B, 1);
        // the key is the instantiation launch_res picks (= the kernel name rocprofv3 reports): <KS, LN, RT, DBUF>
        const int rt = ks <= 16 ? (a.res ? (ntw > 2 ? 4 : 2) : 0) : 0;
        const bool dbuf = ks <= 16 && !(ks > 8 && rt == 4);
        snprintf(key, sizeof(key), "conv1x1_res_kernel<%d, %s, %d, %s>", ks, (ks <= 16 && a.ln_w) ? "true" : "false", rt, dbuf ? "true" : "false");
        ProfScope prof(st, key, work_flops, work_bytes);
        switch (ks) {
            case 4: launch_res<4>(a, ntw, ngroups, grid, lds, st); break;
            case 8: launch_res<8>(a, ntw, ngroups, grid, lds, st); break;
            case 12: launch_res<12>(a, ntw, ngroups, grid, lds, st); break;
            case 16: launch_res<16>(a, ntw, ngroups, grid, lds, st); break;
            case 24: launch_res<24>(a, ntw, ngroups, grid, lds, st); break;
            default: launch_res<32>(a, ntw, ngroups, grid, lds, st); break;
        }
    } else {
        // accumulator tiles per workgroup: 8 (128 channels) unless that would be mostly padding -- or unless the grid
        // would fill the 512 resident workgroup slots badly (level 3 of config 2: 256 or 768 workgroups = half-empty
        // rounds); 4 tiles per workgroup double the grid.  The k order per output is the same either way (same bits).
        int nco = (NT % 8 == 0 || NT > 12) ? 8 : 4;
        if (nco == 8) {
            const long units = (long)cdiv(a.P, 256) * a.B;
            const long cost8 = ((units * cdiv(NT, 8) + 511) / 512) * 8, cost4 = ((units * cdiv(NT, 4) + 511) / 512) * 4;
            if (cost4 < cost8) nco = 4;
        }
        const int ngroups = cdiv(NT, nco);
        dim3 grid((unsigned)(cdiv(a.P, 256) * ngroups), (unsigned)a.B, 1);
        snprintf(key, sizeof(key), "conv1x1_stream_kernel<%d, %d, %s>", nco, nco == 8 ? 4 : 8, a.ln_w ? "true" : "false");
        ProfScope prof(st, key, work_flops, work_bytes);
        if (nco == 8) {
            if (a.ln_w) conv1x1_stream_kernel<8, 4, true><<<grid, 256, 0, st>>>(a, ngroups);
            else conv1x1_stream_kernel<8, 4, false><<<grid, 256, 0, st>>>(a, ngroups);
        } else {
            if (a.ln_w) conv1x1_stream_kernel<4, 8, true><<<grid, 256, 0, st>>>(a, ngroups);
            else conv1x1_stream_kernel<4, 8, false><<<grid, 256, 0, st>>>(a, ngroups);
        }
    }
    return check_launch("conv1x1");
}

}  // namespace rf
