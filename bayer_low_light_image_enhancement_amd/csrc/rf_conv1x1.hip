// 1x1 convolution = per-image GEMM  out[co][p] = sum_k W[co][k] * x[k][p]  on the f32 matrix
// cores (v_mfma_f32_16x16x4_f32: exact f32 fmaf chain, RawFormer keeps reference numerics).
//
// Mapping (one wave = 64 pixels x (NCO * 16) output channels):
//   * lane (j = l & 15, kq = l >> 4) reads ONE float4 = pixels p0 + 4j .. 4j+3 of channel 4s + kq:
//     16 lanes cover 256 contiguous bytes of a channel row, a wave instruction covers 1 KiB.
//     Component g of that float4 is the B operand of MFMA g, so the x tile is never staged in
//     LDS (each element is used by exactly one wave) and is read from HBM exactly once.
//   * the A operand is one float per lane from the lane-ordered packed weights (L1/L2 resident).
//   * D: lane holds channels 16t + 4kq + r for its 4 pixels -> float4 stores, 256 B per 16 lanes.
// Fused in: LayerNorm prologue (per-pixel statistics over the K channels, a4), a second source
// (torch.cat along channels without the copy), bias, residual add, LeakyReLU, and the
// ConvTranspose2d(k=2, s=2) scatter (a9).  Per-image weight sets (wp_bstride != 0) carry the
// attention / squeeze-excite matrices folded into the projection.
#include <cstdio>
#include "rf_common.h"

namespace rf {

template <int NCO, bool LN>
__global__ void __launch_bounds__(256) conv1x1_kernel(Conv1x1Args a, int ngroups, int vec) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int grp = blockIdx.x % ngroups;
    const int tile = blockIdx.x / ngroups;
    const int b = blockIdx.y;
    const int P = a.P;
    const int K = a.C1 + a.C2;
    const int NS = (K + 3) >> 2;
    const int NT = (a.Cout + 15) >> 4;
    const int t0 = grp * NCO;
    const int p0 = (tile * 4 + wave) * 64 + 4 * j;
    const float* x1 = a.x1 + (size_t)b * a.x1_bstride;
    const float* x2 = a.x2 ? a.x2 + (size_t)b * a.x2_bstride : nullptr;

    auto load_x = [&](int k) -> float4 {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < K) {
            const float* src = (k < a.C1) ? x1 + (size_t)k * P : x2 + (size_t)(k - a.C1) * P;
            if (vec) {
                if (p0 < P) v = *reinterpret_cast<const float4*>(src + p0);
            } else {
                if (p0 < P) v.x = src[p0];
                if (p0 + 1 < P) v.y = src[p0 + 1];
                if (p0 + 2 < P) v.z = src[p0 + 2];
                if (p0 + 3 < P) v.w = src[p0 + 3];
            }
        }
        return v;
    };

    // ---- LayerNorm statistics for my 4 pixels (shifted sums; lanes kq = 0..3 split the channels)
    float lnA[4] = {1.f, 1.f, 1.f, 1.f}, lnB[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (LN) {
        const float4 s4 = load_x(0);
        const float sh[4] = {s4.x, s4.y, s4.z, s4.w};
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < NS; ++s) {
            const int k = 4 * s + kq;
            if (k < K) {
                const float4 t = load_x(k);
                const float d[4] = {t.x - sh[0], t.y - sh[1], t.z - sh[2], t.w - sh[3]};
#pragma unroll
                for (int g = 0; g < 4; ++g) { s1[g] += d[g]; s2[g] = fmaf(d[g], d[g], s2[g]); }
            }
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
            lnB[g] = a.ln_b ? -(sh[g] + md) * rstd : 0.f;   // BiasFree form keeps the mean
        }
    }

    f32x4 acc[NCO][4];
#pragma unroll
    for (int t = 0; t < NCO; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[t][g] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float* wp = a.wp + (size_t)b * a.wp_bstride + (size_t)t0 * 64 + lane;
    float4 xv = load_x(kq);
    float av[NCO];
#pragma unroll
    for (int t = 0; t < NCO; ++t) av[t] = (t0 + t < NT) ? wp[(size_t)t * 64] : 0.f;

    for (int s = 0; s < NS; ++s) {
        // prefetch the next k-set while this one is in the matrix pipe
        float4 xn = make_float4(0.f, 0.f, 0.f, 0.f);
        float an[NCO];
#pragma unroll
        for (int t = 0; t < NCO; ++t) an[t] = 0.f;
        if (s + 1 < NS) {
            xn = load_x(4 * (s + 1) + kq);
#pragma unroll
            for (int t = 0; t < NCO; ++t)
                if (t0 + t < NT) an[t] = wp[((size_t)(s + 1) * NT + t) * 64];
        }
        float xb[4] = {xv.x, xv.y, xv.z, xv.w};
        if constexpr (LN) {
            const int k = 4 * s + kq;
            if (k < K) {
                const float gk = a.ln_w[k], bk = a.ln_b ? a.ln_b[k] : 0.f;
#pragma unroll
                for (int g = 0; g < 4; ++g) xb[g] = fmaf(fmaf(xb[g], lnA[g], lnB[g]), gk, bk);
            }
        }
#pragma unroll
        for (int t = 0; t < NCO; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                acc[t][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], xb[g], acc[t][g], 0, 0, 0);
        xv = xn;
#pragma unroll
        for (int t = 0; t < NCO; ++t) av[t] = an[t];
    }

    // ---- epilogue
    if (p0 >= P) return;
    float* outb = a.out + (size_t)b * a.out_bstride;
    if (a.mode == 0) {
        const float* resb = a.res ? a.res + (size_t)b * a.res_bstride : nullptr;
#pragma unroll
        for (int t = 0; t < NCO; ++t) {
            if (t0 + t >= NT) break;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = 16 * (t0 + t) + 4 * kq + r;
                if (co >= a.Cout) continue;
                const float bs = a.bias ? a.bias[co] : 0.f;
                float v[4] = {acc[t][0][r] + bs, acc[t][1][r] + bs, acc[t][2][r] + bs, acc[t][3][r] + bs};
                const size_t off = (size_t)co * P + p0;
                if (vec) {
                    if (resb) {
                        const float4 rv = *reinterpret_cast<const float4*>(resb + off);
                        v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                    }
                    if (a.act == 1) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) v[g] = v[g] > 0.f ? v[g] : 0.2f * v[g];
                    }
                    *reinterpret_cast<float4*>(outb + off) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (p0 + g < P) {
                            float u = v[g] + (resb ? resb[off + g] : 0.f);
                            if (a.act == 1) u = u > 0.f ? u : 0.2f * u;
                            outb[off + g] = u;
                        }
                    }
                }
            }
        }
    } else {
        // ConvTranspose2d(k=2, s=2): row 4*o + 2*i + jj of the GEMM is output channel o at
        // sub-position (i, jj); lane holds the whole 2x2 patch of its pixels for channel o.
        const int w = a.w, w2 = 2 * a.w;
#pragma unroll
        for (int t = 0; t < NCO; ++t) {
            if (t0 + t >= NT) break;
            const int co = 16 * (t0 + t) + 4 * kq;
            if (co >= a.Cout) continue;
            const int o = co >> 2;
            const float bs = a.bias ? a.bias[o] : 0.f;
            float* op = outb + (size_t)o * 4 * P;
            if (vec) {
                const int y = p0 / w, x = p0 - y * w;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    float* row = op + (size_t)(2 * y + i) * w2 + 2 * x;
                    *reinterpret_cast<float4*>(row) = make_float4(acc[t][0][2 * i] + bs, acc[t][0][2 * i + 1] + bs,
                                                                  acc[t][1][2 * i] + bs, acc[t][1][2 * i + 1] + bs);
                    *reinterpret_cast<float4*>(row + 4) = make_float4(acc[t][2][2 * i] + bs, acc[t][2][2 * i + 1] + bs,
                                                                      acc[t][3][2 * i] + bs, acc[t][3][2 * i + 1] + bs);
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int p = p0 + g;
                    if (p < P) {
                        const int y = p / w, x = p - y * w;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            op[(size_t)(2 * y + (r >> 1)) * w2 + 2 * x + (r & 1)] = acc[t][g][r] + bs;
                    }
                }
            }
        }
    }
}

static int pick_nco(int NT) {
    if (NT % 8 == 0) return 8;
    if (NT % 6 == 0) return 6;
    if (NT % 4 == 0) return 4;
    if (NT % 3 == 0) return 3;
    if (NT % 2 == 0) return 2;
    if (NT == 1) return 1;
    return 4;   // ragged: the last group runs partly empty
}

template <int NCO>
static void launch_t(const Conv1x1Args& a, int ngroups, int vec, dim3 grid, hipStream_t st) {
    if (a.ln_w)
        conv1x1_kernel<NCO, true><<<grid, 256, 0, st>>>(a, ngroups, vec);
    else
        conv1x1_kernel<NCO, false><<<grid, 256, 0, st>>>(a, ngroups, vec);
}

int launch_conv1x1(const Conv1x1Args& a, hipStream_t st) {
    RF_CHECK_ARG(a.B > 0 && a.P > 0 && a.C1 > 0 && a.C2 >= 0 && a.Cout > 0, "conv1x1: bad sizes B=%d P=%d C1=%d C2=%d Cout=%d",
                 a.B, a.P, a.C1, a.C2, a.Cout);
    RF_CHECK_ARG(a.C2 == 0 || a.x2 != nullptr, "conv1x1: second source missing");
    RF_CHECK_ARG(a.mode == 0 || (a.Cout % 4 == 0 && a.w > 0 && a.P % a.w == 0), "conv1x1: bad ConvTranspose geometry");
    RF_CHECK_ARG(a.B <= 65535, "conv1x1: batch %d too large", a.B);
    const int NT = cdiv(a.Cout, 16);
    const int nco = pick_nco(NT);
    const int ngroups = cdiv(NT, nco);
    int vec = (a.P % 4 == 0) && aligned16(a.x1) && aligned16(a.out) && (a.x1_bstride % 4 == 0) && (a.out_bstride % 4 == 0);
    if (a.x2) vec = vec && aligned16(a.x2) && (a.x2_bstride % 4 == 0);
    if (a.res) vec = vec && aligned16(a.res) && (a.res_bstride % 4 == 0);
    if (a.mode == 1) vec = vec && (a.w % 4 == 0);
    dim3 grid((unsigned)(cdiv(a.P, 256) * ngroups), (unsigned)a.B, 1);
    char key[64];
    snprintf(key, sizeof(key), "conv1x1_kernel<%d, %s>", nco, a.ln_w ? "true" : "false");
    const double px = (double)a.B * a.P, K = a.C1 + a.C2;
    const double outs = a.mode == 1 ? a.Cout : a.Cout;   // mode 1 writes Cout/4 channels at 4x the pixels
    ProfScope prof(st, key, 2.0 * K * a.Cout * px, 4.0 * px * (K + outs + (a.res ? a.Cout : 0)));
    switch (nco) {
        case 8: launch_t<8>(a, ngroups, vec, grid, st); break;
        case 6: launch_t<6>(a, ngroups, vec, grid, st); break;
        case 4: launch_t<4>(a, ngroups, vec, grid, st); break;
        case 3: launch_t<3>(a, ngroups, vec, grid, st); break;
        case 2: launch_t<2>(a, ngroups, vec, grid, st); break;
        default: launch_t<1>(a, ngroups, vec, grid, st); break;
    }
    return check_launch("conv1x1");
}

}  // namespace rf
