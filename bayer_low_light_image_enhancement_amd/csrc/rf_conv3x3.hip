// Dense 3x3 convolution (padding 1) as an implicit GEMM on the f32 matrix cores
// (v_mfma_f32_16x16x4_f32; K = 4 input channels at one tap per instruction).
//
// Workgroup = 4 waves; it owns a (4*RW rows) x (64/RW cols) pixel tile and NCO*16 output
// channels.  Per 8-input-channel chunk the halo'd input tile and the matching slice of the
// lane-ordered packed weights sit in LDS; every wave then walks 2 k-sets x 9 taps.
//   * B operand: lane (kq, j) owns 4 consecutive pixels of one row.  Two ds_read_b128 per
//     (k-set, dy) fetch the 6 neighbours those pixels need; tap dx of pixel g is v[g + dx], so
//     one LDS read feeds 12 * NCO MFMAs.  The plane stride (448 floats) keeps the four kq
//     planes on disjoint LDS slots.
//   * A operand: ds_read_b32, lane-linear (conflict-free).
//   * D: channels 16t + 4kq + r for the lane's 4 pixels -> 16-byte stores.
// Pipeline: LDS is double buffered; the global loads of chunk c+1 (input tile with zero
// padding, weight slice) are issued into registers before the MFMA block of chunk c and written
// to the other LDS buffer after it: one barrier per chunk, HBM/L2 latency hidden behind the
// matrix pipe.  Per-thread gather offsets are computed once, not per chunk.
// Fused: Bayer pack on the input side (a1, the embedding conv reads the mosaic directly),
// input/output clamps, bias, LeakyReLU(0.2), and the pixel-unshuffle (Downsample, a8) or
// pixel-shuffle (conv_out + PixelShuffle, a10) store.
#include <cstdio>
#include "rf_common.h"

namespace rf {

static constexpr int KC = 8;        // input channels per LDS chunk

template <int NCO, int LOG2_RW, int RPW>
__global__ void __launch_bounds__(256, 2) conv3x3_kernel(Conv3x3Args a, int ngroups, int tiles_x, int vec) {
    constexpr int RW = 1 << LOG2_RW;         // rows one MFMA pixel group spans (narrow images)
    constexpr int TW = 64 / RW;              // tile width
    constexpr int TH = 4 * RW * RPW;         // tile height: every wave owns RPW row groups (RPW = 2 for few output
                                             // channels: twice the MFMAs per staged weight, 25 % less halo)
    constexpr int RS = TW + 8;               // LDS row stride
    static_assert(RPW == 1 || RW == 1, "two row groups per wave only with one row per group");
    constexpr int PS = ((TH + 2) * RS + 63) / 64 * 64;   // LDS plane stride in floats (multiple of 64)
    constexpr int NIN = KC * (TH + 2) * (TW + 2);       // staged input elements per chunk
    constexpr int EPT = (NIN + 255) / 256;              // ... per thread
    constexpr int NW4 = 2 * 9 * NCO * 16;               // staged weight float4 per chunk
    constexpr int WPT = (NW4 + 255) / 256;
    constexpr int BUF = KC * PS + 2 * 9 * NCO * 64;     // floats per LDS buffer
    __shared__ __attribute__((aligned(16))) float lds[2 * BUF];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int rowj = j / (16 / RW), cg = j % (16 / RW);
    const int grp = blockIdx.x % ngroups;
    const int tile = blockIdx.x / ngroups;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int b = blockIdx.y;
    const int h = a.h, w = a.w;
    const int x0 = tx * TW, y0 = ty * TH;
    const int NT = (a.Cout + 15) >> 4;
    const int t0 = grp * NCO;
    const int nchunks = (a.Cin + KC - 1) / KC;
    const float* xb = a.x + (size_t)b * a.x_bstride;
    const size_t chunk_stride = a.unshuffle_in ? (size_t)(KC / 4) * 4 * h * w : (size_t)KC * h * w;

    // ---- per-thread gather plan for the input tile (same for every chunk up to the channel base)
    int goff[EPT];        // offset inside the chunk's KC planes, or -1 = zero padding / beyond the tile
    short loff[EPT];      // LDS offset inside the buffer
    short gcl[EPT];       // channel inside the chunk
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int idx = tid + 256 * i;
        const int c = idx % (TW + 2);
        const int r = (idx / (TW + 2)) % (TH + 2);
        const int cl = idx / ((TW + 2) * (TH + 2));
        const int y = y0 - 1 + r, x = x0 - 1 + c;
        const bool ok = idx < NIN && y >= 0 && y < h && x >= 0 && x < w;
        int off;
        if (a.unshuffle_in)   // packed channel cl = 2*i + jj of mosaic plane 0 lives at (2y+i, 2x+jj); KC = 8 covers 2 planes
            off = ((cl >> 2) * 2 * h + 2 * y + ((cl >> 1) & 1)) * (2 * w) + 2 * x + (cl & 1);
        else
            off = (cl * h + y) * w + x;
        goff[i] = ok ? off : -1;
        loff[i] = (short)(idx < NIN ? cl * PS + r * RS + c : -1);
        gcl[i] = (short)cl;
    }
    float xin[EPT];
    float4 win[WPT];
    auto load_chunk = [&](int ch) {
        const float* src = xb + (size_t)ch * chunk_stride;
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const bool ok = goff[i] >= 0 && ch * KC + gcl[i] < a.Cin;
            float v = src[ok ? goff[i] : 0];          // branch-free: clamped address, value masked below
            if (a.clamp_in) v = fminf(fmaxf(v, 0.f), 1.f);
            xin[i] = ok ? v : 0.f;
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int idx = tid + 256 * i;
            const int l4 = idx % 16;
            const int t = (idx / 16) % NCO;
            const int kt = idx / (16 * NCO);          // ks * 9 + tap
            const bool ok = idx < NW4 && t0 + t < NT;
            const float4 v = *reinterpret_cast<const float4*>(a.wp + (((size_t)(ch * 2) * 9 + (ok ? kt : 0)) * NT + t0 + (ok ? t : 0)) * 64 + l4 * 4);
            win[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_chunk = [&](int buf) {
        float* li = lds + buf * BUF;
        float* lw = li + KC * PS;
#pragma unroll
        for (int i = 0; i < EPT; ++i)
            if (loff[i] >= 0) li[loff[i]] = xin[i];
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int idx = tid + 256 * i;
            if (idx < NW4) *reinterpret_cast<float4*>(lw + idx * 4) = win[i];   // [kt][t][64] is linear in idx
        }
    };

    f32x4 acc[RPW][NCO][4];
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
        for (int t = 0; t < NCO; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[rr][t][g] = (f32x4){0.f, 0.f, 0.f, 0.f};

    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        if (ch + 1 < nchunks) load_chunk(ch + 1);       // in flight during the MFMA block below
        const float* lds_in = lds + (ch & 1) * BUF;
        const float* lds_w = lds_in + KC * PS;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const float* lp = lds_in + (ks * 4 + kq) * PS + (wave * RW * RPW + rowj) * RS + 4 * cg;
#pragma unroll
            for (int ir = 0; ir < RPW + 2; ++ir) {          // input row group ir feeds output row groups ir-2 .. ir
                const float4 lo = *reinterpret_cast<const float4*>(lp + ir * RS);
                const float4 hi = *reinterpret_cast<const float4*>(lp + ir * RS + 4);
                const float v[6] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y};
#pragma unroll
                for (int rr = 0; rr < RPW; ++rr) {
                    const int dy = ir - rr;
                    if (dy < 0 || dy > 2) continue;
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const float* wl = lds_w + ((ks * 9 + dy * 3 + dx) * NCO) * 64 + lane;
#pragma unroll
                        for (int t = 0; t < NCO; ++t) {
                            const float av = wl[t * 64];
#pragma unroll
                            for (int g = 0; g < 4; ++g)
                                acc[rr][t][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, v[g + dx], acc[rr][t][g], 0, 0, 0);
                        }
                    }
                }
            }
        }
        if (ch + 1 < nchunks) store_chunk((ch + 1) & 1);   // the other buffer: last read one barrier ago
        __syncthreads();
    }

    // ---- epilogue
    float* outb = a.out + (size_t)b * a.out_bstride;
    const int x = x0 + 4 * cg;
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
    const int y = y0 + (wave * RPW + rr) * RW + rowj;
    if (y >= h || x >= w) continue;
#pragma unroll
    for (int t = 0; t < NCO; ++t) {
        if (t0 + t >= NT) break;
        float v[4][4];   // [r][g]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = 16 * (t0 + t) + 4 * kq + r;
            const float bs = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float u = acc[rr][t][g][r] + bs;
                if (a.act == 1) u = u > 0.f ? u : 0.2f * u;
                if (a.clamp_out) u = fminf(fmaxf(u, 0.f), 1.f);
                v[r][g] = u;
            }
        }
        const int cobase = 16 * (t0 + t) + 4 * kq;
        if (a.store == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = cobase + r;
                if (co >= a.Cout) continue;
                float* o = outb + ((size_t)co * h + y) * w + x;
                if (vec) {
                    *reinterpret_cast<float4*>(o) = make_float4(v[r][0], v[r][1], v[r][2], v[r][3]);
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        if (x + g < w) o[g] = v[r][g];
                }
            }
        } else if (a.store == 1) {
            // Downsample: out[4*co + 2*(y&1) + (x&1)][y>>1][x>>1]
            const int h2 = h >> 1, w2 = w >> 1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = cobase + r;
                if (co >= a.Cout) continue;
                float* o = outb + (((size_t)(4 * co + 2 * (y & 1))) * h2 + (y >> 1)) * w2 + (x >> 1);
                const size_t ps = (size_t)h2 * w2;
                if (vec) {
                    *reinterpret_cast<float2*>(o) = make_float2(v[r][0], v[r][2]);
                    *reinterpret_cast<float2*>(o + ps) = make_float2(v[r][1], v[r][3]);
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        if (x + g < w) o[(size_t)(g & 1) * ps + (g >> 1)] = v[r][g];
                }
            }
        } else {
            // conv_out + PixelShuffle(2): GEMM row 4*c + 2*i + jj -> out[c][2y+i][2x+jj]
            if (cobase >= a.Cout) continue;
            const int c = cobase >> 2;
            float* op = outb + (size_t)c * 4 * h * w;
            const int w2 = 2 * w;
            if (vec) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    float* row = op + (size_t)(2 * y + i) * w2 + 2 * x;
                    *reinterpret_cast<float4*>(row) = make_float4(v[2 * i][0], v[2 * i + 1][0], v[2 * i][1], v[2 * i + 1][1]);
                    *reinterpret_cast<float4*>(row + 4) = make_float4(v[2 * i][2], v[2 * i + 1][2], v[2 * i][3], v[2 * i + 1][3]);
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (x + g < w) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            op[(size_t)(2 * y + (r >> 1)) * w2 + 2 * (x + g) + (r & 1)] = v[r][g];
                    }
            }
        }
    }
    }
}

template <int NCO>
static void launch_rw(const Conv3x3Args& a, int ngroups, int vec, hipStream_t st) {
    // pick the tile shape from the image width: 64x4, 32x8 or 16x16 pixels
    if (a.w > 32) {
        // (two row groups per wave, RPW = 2, was measured for NCO <= 2: no gain -- 231 VGPRs -- so RPW = 1 everywhere)
        const int txs = cdiv(a.w, 64), tys = cdiv(a.h, 4);
        conv3x3_kernel<NCO, 0, 1><<<dim3((unsigned)(ngroups * txs * tys), (unsigned)a.B), 256, 0, st>>>(a, ngroups, txs, vec);
    } else if (a.w > 16) {
        const int txs = cdiv(a.w, 32), tys = cdiv(a.h, 8);
        conv3x3_kernel<NCO, 1, 1><<<dim3((unsigned)(ngroups * txs * tys), (unsigned)a.B), 256, 0, st>>>(a, ngroups, txs, vec);
    } else {
        const int txs = cdiv(a.w, 16), tys = cdiv(a.h, 16);
        conv3x3_kernel<NCO, 2, 1><<<dim3((unsigned)(ngroups * txs * tys), (unsigned)a.B), 256, 0, st>>>(a, ngroups, txs, vec);
    }
}

int launch_conv3x3(const Conv3x3Args& a, hipStream_t st) {
    RF_CHECK_ARG(a.B > 0 && a.B <= 65535 && a.Cin > 0 && a.Cout > 0 && a.h > 0 && a.w > 0, "conv3x3: bad sizes");
    RF_CHECK_ARG(a.store == 0 || (a.store == 1 && a.h % 2 == 0 && a.w % 2 == 0) || (a.store == 2 && a.Cout % 4 == 0),
                 "conv3x3: store mode %d incompatible with Cout=%d h=%d w=%d", a.store, a.Cout, a.h, a.w);
    RF_CHECK_ARG(!a.unshuffle_in || a.Cin == 4, "conv3x3: the packed-mosaic input path takes exactly 4 channels");
    RF_CHECK_ARG((double)a.Cin * a.h * a.w * 4.0 < 2.0e9, "conv3x3: image too large for 32-bit gather offsets");
    const int NT = cdiv(a.Cout, 16);
    int nco = 4;
    if (NT % 4 != 0) nco = (NT % 3 == 0) ? 3 : (NT % 2 == 0 ? 2 : (NT == 1 ? 1 : 4));
    const int ngroups = cdiv(NT, nco);
    const int vec = (a.w % 4 == 0) && aligned16(a.out) && (a.out_bstride % 4 == 0);
    char key[64];
    snprintf(key, sizeof(key), "conv3x3_kernel<%d, %d, 1>", nco, a.w > 32 ? 0 : (a.w > 16 ? 1 : 2));
    const double px = (double)a.B * a.h * a.w;
    ProfScope prof(st, key, 18.0 * a.Cin * a.Cout * px, 4.0 * px * (a.Cin + a.Cout));
    switch (nco) {
        case 4: launch_rw<4>(a, ngroups, vec, st); break;
        case 3: launch_rw<3>(a, ngroups, vec, st); break;
        case 2: launch_rw<2>(a, ngroups, vec, st); break;
        default: launch_rw<1>(a, ngroups, vec, st); break;
    }
    return check_launch("conv3x3");
}

}  // namespace rf
