// Decoder step  up = ConvTranspose2d(2C, C, 2, stride 2)(x);  out = Conv2d(2C, C, 1)(cat[up, skip])
// (RawFomer_WFB_FFAB/model.py:461-468, 494-503) as ONE kernel.  Both maps are linear, so
//   out[co][2y+i][2x+j] = b'[co] + sum_k Wc[ij][co][k] x[k][y][x] + sum_m Wb[co][m] skip[m][2y+i][2x+j]
//   Wc[ij][co][k] = sum_m Wr[co][m] Wu[k][m][i][j],   Wb = Wr[:, C:],   b' = bR + Wr[:, :C] bU
// with Wc, b' composed once per parameter load (upcat_compose_kernel).  Against the two-kernel form this is
// 24 C^2 instead of 32 C^2 flop per low-resolution pixel and 2.5 C instead of 4.5 C floats of HBM traffic per
// output pixel: `up` is never written or read.
//
// A wave owns 64 low-resolution pixels (lane (j, kq): 4 consecutive pixels of one row, channel 4s + kq) = 256
// output pixels, and NCOG * 16 output channels for all four sub-positions ij: accumulator tile (ij, t).  The x
// k-sets feed every tile; a skip k-set of sub-position ij feeds only the tiles of that ij, its B operand being the
// even (j = 0) or odd (j = 1) pixels of two 16-byte loads of skip row 2y + i.  Weights stream through an LDS
// double buffer (one barrier per chunk of 128 MFMAs per wave).
#include "rf_common.h"

namespace rf {

static constexpr int kUpNCOG = 2;    // output tiles (16 channels) per workgroup and sub-position

static bool upcat_b3(int C) { return (2 * C) % 32 == 0; }     // phase 1 on three-piece bf16 operands: whole K blocks of 32 channels
static size_t upcat_f32_floats(int C) {
    const int ngrp = cdiv(cdiv(C, 16), kUpNCOG);
    const size_t wc = (size_t)(2 * C / 4) * ngrp * 4 * kUpNCOG * 64;
    const size_t wb = (size_t)(C / 4) * ngrp * kUpNCOG * 64;
    return wc + wb + (size_t)ngrp * kUpNCOG * 16;
}
// [Wc | Wb | b' | Wc in b3 form: [K block][T][piece][lane] 16-byte elements = 768 floats per (K block, tile)]
size_t upcat_packed_floats(int C) {
    const int ngrp = cdiv(cdiv(C, 16), kUpNCOG);
    return upcat_f32_floats(C) + (upcat_b3(C) ? (size_t)(2 * C / 32) * ngrp * 4 * kUpNCOG * 768 : 0);
}

// packed Wc: [s = k/4][T = (grp*4 + ij)*NCOG + t][l],  l = i16 + 16 kk;  packed Wb: [s][grp*NCOG + t][l];  then b'
__global__ void __launch_bounds__(256) upcat_compose_kernel(const float* __restrict__ wu, const float* __restrict__ bu,
                                                            const float* __restrict__ wr, const float* __restrict__ br,
                                                            float* __restrict__ packed, int C) {
    const int ngrp = ((C + 15) / 16 + kUpNCOG - 1) / kUpNCOG;
    const int NTc = ngrp * 4 * kUpNCOG, NTb = ngrp * kUpNCOG;
    const size_t nwc = (size_t)(2 * C / 4) * NTc * 64, nwb = (size_t)(C / 4) * NTb * 64, nb = (size_t)NTb * 16;
    for (size_t idx = blockIdx.x * 256ull + threadIdx.x; idx < nwc + nwb + nb; idx += (size_t)gridDim.x * 256) {
        float v = 0.f;
        if (idx < nwc) {
            const int l = (int)(idx & 63), T = (int)((idx >> 6) % NTc), s = (int)((idx >> 6) / NTc);
            const int t = T % kUpNCOG, ij = (T / kUpNCOG) & 3, grp = T / (4 * kUpNCOG);
            const int co = 16 * (grp * kUpNCOG + t) + (l & 15), k = 4 * s + (l >> 4);
            if (co < C) {
                const float* wrow = wr + (size_t)co * 2 * C;                 // Wr[co][0..C) : the `up` half
                const float* wcol = wu + (size_t)k * C * 4 + ij;             // Wu[k][m][i][j], m stride 4
                for (int m = 0; m < C; ++m) v = fmaf(wrow[m], wcol[(size_t)m * 4], v);
            }
        } else if (idx < nwc + nwb) {
            const size_t e = idx - nwc;
            const int l = (int)(e & 63), T = (int)((e >> 6) % NTb), s = (int)((e >> 6) / NTb);
            const int co = 16 * T + (l & 15), m = 4 * s + (l >> 4);
            if (co < C) v = wr[(size_t)co * 2 * C + C + m];
        } else {
            const int co = (int)(idx - nwc - nwb);
            if (co < C) {
                v = br ? br[co] : 0.f;
                if (bu) for (int m = 0; m < C; ++m) v = fmaf(wr[(size_t)co * 2 * C + m], bu[m], v);
            }
        }
        packed[idx] = v;
    }
}

// composed Wc (f32, MFMA operand order [s][T][l]) -> three bf16 pieces in the operand order of v_mfma_f32_16x16x32_bf16 (rf_common.h)
__global__ void __launch_bounds__(256) upcat_b3_kernel(const float* __restrict__ wc, unsigned short* __restrict__ wc3, int C) {
    const int ngrp = ((C + 15) / 16 + kUpNCOG - 1) / kUpNCOG;
    const int NTc = ngrp * 4 * kUpNCOG, K = 2 * C;
    const size_t total = (size_t)NTc * 16 * K;
    for (size_t idx = blockIdx.x * 256ull + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int k = (int)(idx % K), row = (int)(idx / K);          // row = 16 T + i
        const float v = wc[((size_t)(k >> 2) * NTc + (row >> 4)) * 64 + (row & 15) + 16 * (k & 3)];
        b3_store(wc3, NTc, row, k, v);
    }
}

int pack_upcat(const float* up_w, const float* up_b, const float* cr_w, const float* cr_b, float* packed, int C, hipStream_t st) {
    const size_t total = upcat_f32_floats(C);
    int g = (int)((total + 255) / 256);
    if (g > 8192) g = 8192;
    upcat_compose_kernel<<<g, 256, 0, st>>>(up_w, up_b, cr_w, cr_b, packed, C);
    if (upcat_b3(C)) upcat_b3_kernel<<<g, 256, 0, st>>>(packed, reinterpret_cast<unsigned short*>(packed + total), C);
    return check_launch("pack_upcat");
}

struct UpcatArgs {
    const float* x;      // [B][2C][h][w]
    const float* skip;   // [B][C][2h][2w]
    float* out;          // [B][C][2h][2w]
    const float* wc;     // packed composed convT weights
    const float* wb;     // packed skip weights
    const float* bias;   // b'
    const void* wc3;     // Wc in b3 form (B3 kernels)
    int B, C, h, w, ngroups;
};

template <int NCOG, bool B3>
__global__ void __launch_bounds__(256, 2) upcat_kernel(UpcatArgs a) {
    constexpr int KCX = 4, KCS = 2;                        // k-sets per chunk: 4 * 4 NCOG * 4 = 2 * 4 * NCOG * 4 = 128 MFMAs (NCOG = 2)
    constexpr int WX4 = KCX * 4 * NCOG * 16;               // float4 of weights per x chunk
    constexpr int WS4 = KCS * NCOG * 16;                   // ... per skip chunk
    constexpr int WPTX = (WX4 + 255) / 256;
    constexpr int W3 = 4 * NCOG * 192;                      // 16-byte elements of one K block's A pieces (b3 phase 1)
    __shared__ __attribute__((aligned(16))) float lds_raw[B3 ? 2 * W3 * 4 : 2 * WX4 * 4];
    float (*lds_w)[WX4 * 4] = reinterpret_cast<float (*)[WX4 * 4]>(lds_raw);      // f32 view (phase 2; phase 1 without b3)
    __shared__ float bias_l[NCOG * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int grp = blockIdx.x % a.ngroups, tile = blockIdx.x / a.ngroups;
    const size_t b = blockIdx.y;
    const int C = a.C, h = a.h, w = a.w, P = h * w;
    const int NSx = 2 * C / 4, NSs = C / 4;
    const int NTc = a.ngroups * 4 * NCOG, NTb = a.ngroups * NCOG;
    const int ncx = (NSx + KCX - 1) / KCX, ncs = (NSs + KCS - 1) / KCS;
    const int p0 = (tile * 4 + wave) * 64 + 4 * j;
    const bool live = p0 < P;
    const int pc = live ? p0 : 0;
    const int y = pc / w, x = pc - y * w;
    if (tid < NCOG * 16) bias_l[tid] = a.bias[grp * NCOG * 16 + tid];

    f32x4 acc[4][NCOG][4];
#pragma unroll
    for (int ij = 0; ij < 4; ++ij)
#pragma unroll
        for (int t = 0; t < NCOG; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[ij][t][g] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---------------- phase 1: composed transposed convolution, K = 2C channels of x ----------------
    if constexpr (B3) {
        // K blocks of 32 channels on v_mfma_f32_16x16x32_bf16 with three-piece operands (rf_common.h): lane (j, kq) loads channels
        // 32 c + 8 kq + i of its 4 pixels, splits them once per block (176 VALU) and feeds all 4 * NCOG accumulator tiles
        // (192 MFMAs of 17 cycles per block instead of 256 of 33 for the f32 instruction)
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4* lw3 = reinterpret_cast<u32x4*>(lds_raw);
        const u32x4* wg = reinterpret_cast<const u32x4*>(a.wc3);
        const int nkb = 2 * C / 32;
        constexpr int WPT3 = (W3 + 255) / 256;
        const float* xb = a.x + b * 2 * C * (size_t)P + (size_t)(8 * kq) * P + pc;
        float4 xr[8];
        u32x4 wr[WPT3];
        auto load_x = [&](int c) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 8; ++i) xr[i] = *reinterpret_cast<const float4*>(xb + (size_t)(32 * c + i) * P);
        };
        auto load_w = [&](int c) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < WPT3; ++i) {
                const int idx = tid + 256 * i;                         // (T, piece, lane) of this group's slice: contiguous in wc3
                wr[i] = wg[((size_t)c * NTc + grp * 4 * NCOG) * 192 + (idx < W3 ? idx : 0)];
            }
        };
        auto store_w = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < WPT3; ++i)
                if (tid + 256 * i < W3) lw3[buf * W3 + tid + 256 * i] = wr[i];
        };
        load_x(0);
        load_w(0);
        store_w(0);
        __syncthreads();
        for (int c = 0; c < nkb; ++c) {
            u32x4 bp[4][3];
#pragma unroll
            for (int hp = 0; hp < 4; ++hp) {
                const float xa[4] = {xr[2 * hp].x, xr[2 * hp].y, xr[2 * hp].z, xr[2 * hp].w};
                const float xc[4] = {xr[2 * hp + 1].x, xr[2 * hp + 1].y, xr[2 * hp + 1].z, xr[2 * hp + 1].w};
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    unsigned a0, a1, a2, b0, b1, b2;
                    b3_split(xa[g], a0, a1, a2);
                    b3_split(xc[g], b0, b1, b2);
                    bp[g][0][hp] = b3_pack(a0, b0);
                    bp[g][1][hp] = b3_pack(a1, b1);
                    bp[g][2][hp] = b3_pack(a2, b2);
                }
            }
            const int cn = c + 1 < nkb ? c + 1 : c;                    // branch-free prefetch (the last one re-reads)
            load_x(cn);
            load_w(cn);
            const u32x4* wl = lw3 + (c & 1) * W3 + lane;
#pragma unroll
            for (int T = 0; T < 4 * NCOG; ++T) {
                const u32x4 ap[3] = {wl[(T * 3 + 0) * 64], wl[(T * 3 + 1) * 64], wl[(T * 3 + 2) * 64]};
                b3_mfma4(ap, bp, acc[T / NCOG][T % NCOG]);
            }
            if (c + 1 < nkb) store_w((c + 1) & 1);
            __syncthreads();
        }
    } else {
        const float* xb = a.x + b * 2 * C * (size_t)P + (size_t)kq * P + pc;
        float4 xa[KCX], xn[KCX], wr[WPTX];
        auto load_x = [&](int c, float4 (&dst)[KCX]) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < KCX; ++i) {
                const int s = min(c * KCX + i, NSx - 1);             // k-sets past K meet zero weights
                dst[i] = *reinterpret_cast<const float4*>(xb + (size_t)(4 * s) * P);
            }
        };
        auto load_w = [&](int c) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < WPTX; ++i) {
                const int idx = tid + 256 * i;
                const int l4 = idx & 15, T = (idx >> 4) % (4 * NCOG), s = c * KCX + (idx >> 4) / (4 * NCOG);
                const bool ok = idx < WX4 && s < NSx;
                const float4 v = *reinterpret_cast<const float4*>(a.wc + ((size_t)(ok ? s : 0) * NTc + grp * 4 * NCOG + T) * 64 + l4 * 4);
                wr[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        auto store_w = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < WPTX; ++i)
                if (tid + 256 * i < WX4) *reinterpret_cast<float4*>(&lds_w[buf][(tid + 256 * i) * 4]) = wr[i];
        };
        auto chunk = [&](int c, float4 (&xc)[KCX], float4 (&xnext)[KCX]) __attribute__((always_inline)) {
            load_x(c + 1 < ncx ? c + 1 : c, xnext);                  // branch-free prefetch
            load_w(c + 1 < ncx ? c + 1 : c);
            const float* wl = &lds_w[c & 1][lane];
#pragma unroll
            for (int i = 0; i < KCX; ++i) {
                const float xv[4] = {xc[i].x, xc[i].y, xc[i].z, xc[i].w};
#pragma unroll
                for (int T = 0; T < 4 * NCOG; ++T) {
                    const float av = wl[(i * 4 * NCOG + T) * 64];
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        acc[T / NCOG][T % NCOG][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xv[g], acc[T / NCOG][T % NCOG][g], 0, 0, 0);
                }
            }
            if (c + 1 < ncx) store_w((c + 1) & 1);
            __syncthreads();
        };
        load_x(0, xa);
        load_w(0);
        store_w(0);
        __syncthreads();
        for (int c = 0; c < ncx; c += 2) {
            chunk(c, xa, xn);
            if (c + 1 < ncx) chunk(c + 1, xn, xa);
        }
    }
    // ---------------- phase 2: skip connection, K = C channels at the output resolution ----------------
    {
        const int w2 = 2 * w;
        const float* sb = a.skip + (b * C + kq) * (size_t)(4 * P) + (size_t)(2 * y) * w2 + 2 * x;
        float4 sa[KCS][4], sn[KCS][4];      // [k-set][2 i + half]: skip row 2y + i, pixels 2x .. 2x+3 | 2x+4 .. 2x+7
        float4 wr;
        auto load_s = [&](int c, float4 (&dst)[KCS][4]) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < KCS; ++q) {
                const int s = min(c * KCS + q, NSs - 1);
                const float* p = sb + (size_t)(4 * s) * (4 * P);
                dst[q][0] = *reinterpret_cast<const float4*>(p);
                dst[q][1] = *reinterpret_cast<const float4*>(p + 4);
                dst[q][2] = *reinterpret_cast<const float4*>(p + w2);
                dst[q][3] = *reinterpret_cast<const float4*>(p + w2 + 4);
            }
        };
        auto load_w = [&](int c) __attribute__((always_inline)) {
            const int idx = tid < WS4 ? tid : 0;
            const int l4 = idx & 15, t = (idx >> 4) % NCOG, s = c * KCS + (idx >> 4) / NCOG;
            const bool ok = s < NSs;
            const float4 v = *reinterpret_cast<const float4*>(a.wb + ((size_t)(ok ? s : 0) * NTb + grp * NCOG + t) * 64 + l4 * 4);
            wr = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        };
        auto store_w = [&](int buf) __attribute__((always_inline)) {
            if (tid < WS4) *reinterpret_cast<float4*>(&lds_w[buf][tid * 4]) = wr;
        };
        auto chunk = [&](int c, float4 (&sc)[KCS][4], float4 (&snext)[KCS][4]) __attribute__((always_inline)) {
            load_s(c + 1 < ncs ? c + 1 : c, snext);
            load_w(c + 1 < ncs ? c + 1 : c);
            const float* wl = &lds_w[c & 1][lane];
#pragma unroll
            for (int q = 0; q < KCS; ++q) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float4 lo = sc[q][2 * i], hi = sc[q][2 * i + 1];
                    const float bv[2][4] = {{lo.x, lo.z, hi.x, hi.z}, {lo.y, lo.w, hi.y, hi.w}};     // j = 0: even pixels, j = 1: odd
#pragma unroll
                    for (int t = 0; t < NCOG; ++t) {
                        const float av = wl[(q * NCOG + t) * 64];
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                            for (int g = 0; g < 4; ++g)
                                acc[2 * i + jj][t][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[jj][g], acc[2 * i + jj][t][g], 0, 0, 0);
                    }
                }
            }
            if (c + 1 < ncs) store_w((c + 1) & 1);
            __syncthreads();
        };
        load_s(0, sa);
        load_w(0);
        store_w(0);
        __syncthreads();
        for (int c = 0; c < ncs; c += 2) {
            chunk(c, sa, sn);
            if (c + 1 < ncs) chunk(c + 1, sn, sa);
        }
    }
    // ---------------- epilogue: rows 2y and 2y+1, pixels 2x .. 2x+7 of channels 16 t + 4 kq + r ----------------
    if (live) {
        const int w2 = 2 * w;
#pragma unroll
        for (int t = 0; t < NCOG; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = 16 * (grp * NCOG + t) + 4 * kq + r;
                if (co >= C) continue;
                const float bs = bias_l[16 * t + 4 * kq + r];
                float* op = a.out + (b * C + co) * (size_t)(4 * P) + (size_t)(2 * y) * w2 + 2 * x;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const f32x4* e = acc[2 * i][t];       // j = 0
                    const f32x4* o = acc[2 * i + 1][t];   // j = 1
                    *reinterpret_cast<float4*>(op + i * w2) = make_float4(e[0][r] + bs, o[0][r] + bs, e[1][r] + bs, o[1][r] + bs);
                    *reinterpret_cast<float4*>(op + i * w2 + 4) = make_float4(e[2][r] + bs, o[2][r] + bs, e[3][r] + bs, o[3][r] + bs);
                }
            }
    }
}

bool upcat_supported(int C, int h, int w, const void* x, const void* skip, const void* out) {
    return C % 4 == 0 && w % 4 == 0 && aligned16(x) && aligned16(skip) && aligned16(out) && (double)C * h * w * 16.0 < 4.0e9;
}

// packed = [Wc | Wb | b'] as laid out by pack_upcat
int launch_upcat(const float* x, const float* skip, float* out, const float* packed, int B, int C, int h, int w, hipStream_t st) {
    RF_CHECK_ARG(upcat_supported(C, h, w, x, skip, out) && B > 0 && B <= 65535, "upcat: unsupported shape C=%d %dx%d", C, h, w);
    const int ngrp = cdiv(cdiv(C, 16), kUpNCOG);
    UpcatArgs a{};
    a.x = x; a.skip = skip; a.out = out;
    a.wc = packed;
    a.wb = packed + (size_t)(2 * C / 4) * ngrp * 4 * kUpNCOG * 64;
    a.bias = a.wb + (size_t)(C / 4) * ngrp * kUpNCOG * 64;
    a.wc3 = packed + upcat_f32_floats(C);
    a.B = B; a.C = C; a.h = h; a.w = w; a.ngroups = ngrp;
    const double px = (double)B * h * w;
    ProfScope prof(st, upcat_b3(C) ? "upcat_kernel<2, true>" : "upcat_kernel<2, false>", px * 24.0 * C * C, px * 4.0 * (2.0 * C + 8.0 * C));
    if (upcat_b3(C)) upcat_kernel<kUpNCOG, true><<<dim3((unsigned)(cdiv(h * w, 256) * ngrp), (unsigned)B), 256, 0, st>>>(a);
    else upcat_kernel<kUpNCOG, false><<<dim3((unsigned)(cdiv(h * w, 256) * ngrp), (unsigned)B), 256, 0, st>>>(a);
    return check_launch("upcat");
}

}  // namespace rf
