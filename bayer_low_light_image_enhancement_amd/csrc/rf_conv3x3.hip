// Dense 3x3 convolution (padding 1) as an implicit GEMM on the f32 matrix cores
// (v_mfma_f32_16x16x4_f32; K = 4 input channels per instruction), with the Winograd F(4,3) minimal filtering
// algorithm along x: the lane's 4 output pixels of one kernel row cost 6 products instead of 12,
//   m = (G g) * (B^T d),   y = A^T m
//   G g   = (g0/4, -(g0+g1+g2)/6, -(g0-g1+g2)/6, g0/24+g1/12+g2/6, g0/24-g1/12+g2/6, g2): done when the weights
//           are packed (rf_pack.hip), 18 taps per k-set instead of 9
//   B^T d = (4d0-5d2+d4, (d4-4d2)+(d3-4d1), (d4-4d2)-(d3-4d1), (d4-d2)+2(d3-d1), (d4-d2)-2(d3-d1), 4d1-5d3+d5):
//           12 VALU operations on the two ds_read_b128 of a (k-set, input row), next to 6*NCO MFMAs
//   A^T m = (m0+m1+m2+m3+m4, (m1-m2)+2(m3-m4), (m1+m2)+4(m3+m4), (m1-m2)+8(m3-m4)+m5): once per tile, in the epilogue
// A chunk takes half the MFMAs of the direct form.  Rounding error 2.2 x the direct form's (f32 simulated against
// f64: 5e-7 on O(0.2) outputs at Cin = 32, 1.3e-6 at Cin = 256), inside every parity tolerance.
//
// Workgroup = NWV waves; it owns a (NWV*RW*RPW rows) x (64/RW cols) pixel tile and NCO*16 output
// channels, and is persistent over several such tiles.  Per 8-input-channel chunk the halo'd input tile and the matching slice of the
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
#include <cstdlib>
#include "rf_common.h"

namespace rf {

static constexpr int KC = 8;        // input channels per LDS chunk
static constexpr int kKsCounters = 4096;   // floats at the head of Conv3x3Args::ks_scratch that hold the tickets of a K-split launch
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// NWV = waves per workgroup: 8 on wide images (the 18-tap weight slice is shared by twice the waves, so that two
// waves per SIMD fit the 160 KB of LDS with one workgroup per CU), 4 for the narrow-image tile shapes.
// KS (small launches only, one tile per workgroup): the input-channel chunks are split over gridDim.z workgroups.  Each writes
// its partial tile (after the output transform, before bias and activation) to scratch; the LAST one to arrive -- a ticket per
// (image, tile, output group) -- adds the partials in split order (deterministic whatever the arrival order), applies bias /
// activation / clamp and stores.  One 16 x 16 frame at level 3 is a chain of 32 chunk barriers on 4 workgroups otherwise.
template <int NCO, int LOG2_RW, int RPW, int NWV, bool KS = false>
__global__ void __launch_bounds__(64 * NWV, (NWV >= 8 || NCO >= 3) ? 1 : 2) conv3x3_kernel(Conv3x3Args a, int ngroups, int tiles_x, int ntiles, int vec) {
    constexpr int NTHR = 64 * NWV;
    constexpr int TAPS = 18;                 // transformed taps per k-set (3 kernel rows x 6)
    constexpr int NM = 6;                    // Winograd products per lane and output tile
    constexpr int RW = 1 << LOG2_RW;         // rows one MFMA pixel group spans (narrow images)
    constexpr int TW = 64 / RW;              // tile width
    constexpr int TH = NWV * RW * RPW;       // tile height: a wave owns RPW row groups (RPW = 2 for few output channels:
                                             // twice the MFMAs per barrier and per staged weight, 17 % less halo)
    static_assert(RPW == 1 || RW == 1, "two row groups per wave only with one row per group");
    constexpr int RS = TW + 8;               // LDS row stride
    constexpr int PS = ((TH + 2) * RS + 63) / 64 * 64;   // LDS plane stride in floats (multiple of 64)
    constexpr int NIN = KC * (TH + 2) * (TW + 2);       // staged input elements per chunk
    constexpr int EPT = (NIN + NTHR - 1) / NTHR;        // ... per thread
    constexpr int NW4 = 2 * TAPS * NCO * 16;            // staged weight float4 per chunk
    constexpr int WPT = (NW4 + NTHR - 1) / NTHR;
    constexpr int BUF = KC * PS + 2 * TAPS * NCO * 64;  // floats per LDS buffer
    __shared__ __attribute__((aligned(16))) float lds[2 * BUF];
    __shared__ float bias_l[NCO * 16];       // the epilogue must not read global memory: a load there is followed by
                                             // s_waitcnt vmcnt(0), one exposed round trip per bias value and tile

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int rowj = j / (16 / RW), cg = j % (16 / RW);
    const int grp = blockIdx.x % ngroups;
    const int wg = blockIdx.x / ngroups;
    const int b = blockIdx.y;
    const int h = a.h, w = a.w;
    const int NT = (a.Cout + 15) >> 4;
    const int t0 = grp * NCO;
    const int nchunks_all = (a.Cin + KC - 1) / KC;
    const int ksplit = KS ? (int)gridDim.z : 1, kz = KS ? (int)blockIdx.z : 0;
    const int ch_lo = KS ? (int)((long)kz * nchunks_all / ksplit) : 0, ch_hi = KS ? (int)((long)(kz + 1) * nchunks_all / ksplit) : nchunks_all;
    const float* xb = a.x + (size_t)b * a.x_bstride;
    const size_t chunk_stride = a.unshuffle_in ? (size_t)(KC / 4) * 4 * h * w : (size_t)KC * h * w;

    // ---- staging plan.  Every per-chunk instruction of the staging path is a memory instruction with
    // precomputed operands: f32 MFMA time and VALU time add up on this hardware (measured: ~250 VALU per
    // chunk cost 10 %), so the chunk loop carries no address arithmetic, selects or masks.
    //   * input: buffer_load_dword through a per-chunk descriptor (base = first plane of the chunk,
    //     num_records = the planes that exist).  Zero padding, "no element" and channels beyond Cin
    //     are out-of-range offsets -> the hardware returns 0.
    //   * weights: buffer_load_dwordx4, same trick for output tiles beyond Cout.
    // Element i of this thread is halo'd-tile element idx = tid + NTHR i = (channel cl, row r, col c).
    constexpr unsigned OOB = 0x80000000u;   // > any num_records (images are < 2 GiB, checked by the launcher)
    int prc[EPT];             // (cl << 16) | (r << 8) | c, or -1 = no element
    short loff[EPT];          // LDS float offset inside a buffer ("no element" -> a spare slot in the plane padding,
                              // so the LDS writes need no predicate; 128 spare slots, 2 threads of different waves each)
    static_assert(PS - (TH + 2) * RS >= 16, "plane padding holds the dummy slots");
    const int dummy = (tid & 7) * PS + (TH + 2) * RS + ((tid >> 3) & 15);   // shared only by threads of different waves
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int idx = tid + NTHR * i;
        const int c = idx % (TW + 2);
        const int r = (idx / (TW + 2)) % (TH + 2);
        const int cl = idx / ((TW + 2) * (TH + 2));
        prc[i] = idx < NIN ? (cl << 16) | (r << 8) | c : -1;
        loff[i] = (short)(idx < NIN ? cl * PS + r * RS + c : dummy);
    }
    unsigned wvoff[WPT];      // weight byte offsets inside one chunk's packed slice
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const int idx = tid + NTHR * i;
        const int l4 = idx % 16;
        const int t = (idx / 16) % NCO;
        const int kt = idx / (16 * NCO);          // ks * TAPS + dy * (TAPS / 3) + j
        const bool ok = idx < NW4 && t0 + t < NT;
        wvoff[i] = ok ? (unsigned)(((kt * NT + t0 + t) * 64 + l4 * 4) * 4) : OOB;
    }
    const int w_chunk_bytes = 2 * TAPS * NT * 64 * 4;
    const size_t plane_bytes = (size_t)h * w * 4;      // one packed-resolution plane

    unsigned voff[EPT];       // current tile: byte offset inside the chunk's planes, OOB = zero
    int x0 = 0, y0 = 0;       // origin of the tile the plan (and the loads in flight) belong to
    auto plan_tile = [&](int tile) {
        const int tx = tile % tiles_x, ty = tile / tiles_x;
        x0 = tx * TW;
        y0 = ty * TH;
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            int e = prc[i];
            asm volatile("" : "+v"(e));   // opaque: nothing derived from the element id stays live across the MFMA blocks
            const int cl = e >> 16;
            const int y = y0 - 1 + ((e >> 8) & 255), x = x0 - 1 + (e & 255);
            const bool ok = e >= 0 && (unsigned)y < (unsigned)h && (unsigned)x < (unsigned)w;
            // packed channel cl = 2*i + jj of mosaic plane 0 lives at (2y+i, 2x+jj); KC = 8 covers 2 mosaic planes
            const int off_m = ((cl >> 2) * 2 * h + 2 * y + ((cl >> 1) & 1)) * (2 * w) + 2 * x + (cl & 1);
            const int off_p = (cl * h + y) * w + x;
            voff[i] = ok ? (unsigned)(4 * (a.unshuffle_in ? off_m : off_p)) : OOB;
        }
    };
    float xin[EPT];
    float4 win[WPT];
    auto load_chunk = [&](int ch) {
        // planes of this chunk that exist: min(KC, Cin - ch*KC) channel planes (mosaic input: 4 channels = 1 plane of 4hw)
        const int cl_lim = min(KC, a.Cin - ch * KC);
        const float* src = xb + (size_t)ch * chunk_stride;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(src), 0, (int)(a.unshuffle_in ? (size_t)(cl_lim / 4) * 4 * plane_bytes : (size_t)cl_lim * plane_bytes), 0x00020000);
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.wp) + (size_t)ch * (w_chunk_bytes / 4), 0, w_chunk_bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < EPT; ++i) xin[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, (int)voff[i], 0, 0));
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rw, (int)wvoff[i], 0, 0);
            win[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
    };
    auto store_chunk = [&](int buf) {
        float* li = lds + buf * BUF;
        float* lw = li + KC * PS;
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            float v = xin[i];
            if (a.clamp_in) v = fminf(fmaxf(v, 0.f), 1.f);      // uniform; only the clamp_io embedding conv
            li[loff[i]] = v;
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int idx = tid + NTHR * i;
            if (idx < NW4) *reinterpret_cast<float4*>(lw + idx * 4) = win[i];   // [kt][t][64] is linear in idx
        }
    };

    f32x4 acc[RPW][NCO][NM];     // [row group][output tile][Winograd product m_0..5]

    // ---- persistent over a contiguous range of tiles: the first chunk of the next tile is prefetched
    // behind the last MFMA block of the current one, so only the very first load is exposed
    // (tiles wg, wg + nwg, ...: the workgroups running at one time cover a compact band of the image that moves
    // down it -- contiguous runs per workgroup put them 8 rows x 2^k bytes apart and lost 10 % at 512 x 512)
    const int nwg = gridDim.x / ngroups;
    int tile = wg;
    const int tile_end = ntiles;
    if (tile >= tile_end) return;
    if (tid < NCO * 16) {
        const int co = 16 * t0 + tid;
        bias_l[tid] = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
    }
    plan_tile(tile);
    load_chunk(ch_lo);
    store_chunk(0);
    __syncthreads();
    int buf = 0;
    float* outb = a.out + (size_t)b * a.out_bstride;
    for (; tile < tile_end; tile += nwg) {
        const int ex0 = x0, ey0 = y0;                  // this tile's origin (the plan moves on before the epilogue)
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
            for (int t = 0; t < NCO; ++t)
#pragma unroll
                for (int g = 0; g < NM; ++g) acc[rr][t][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int ch = ch_lo; ch < ch_hi; ++ch) {
            const bool last = ch + 1 == ch_hi;
            const bool more = !last || tile + nwg < tile_end;
            const int nch = last ? ch_lo : ch + 1;
            if (last && more) plan_tile(tile + nwg);
            if (more) load_chunk(nch);                  // in flight during the MFMA block below
            const float* lds_in = lds + buf * BUF;
            const float* lds_w = lds_in + KC * PS;
            const int nks = (a.Cin - ch * KC > 4) ? 2 : 1;   // a 4-channel tail (the embedding conv) skips the empty k-set
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (ks >= nks) break;
                const float* lp = lds_in + (ks * 4 + kq) * PS + (wave * RW * RPW + rowj) * RS + 4 * cg;
#pragma unroll
                for (int ir = 0; ir < RPW + 2; ++ir) {          // input row group ir feeds output row groups ir-2 .. ir
                    const float4 lo = *reinterpret_cast<const float4*>(lp + ir * RS);
                    const float4 hi = *reinterpret_cast<const float4*>(lp + ir * RS + 4);
                    float d[NM];
                    {
                        const float p = fmaf(-4.f, lo.z, hi.x), q = fmaf(-4.f, lo.y, lo.w);      // d4 - 4 d2, d3 - 4 d1
                        const float c2 = hi.x - lo.z, e2 = lo.w - lo.y;                         // d4 - d2,   d3 - d1
                        d[0] = fmaf(4.f, lo.x, fmaf(-5.f, lo.z, hi.x));
                        d[1] = p + q; d[2] = p - q;
                        d[3] = fmaf(2.f, e2, c2); d[4] = fmaf(-2.f, e2, c2);
                        d[5] = fmaf(4.f, lo.y, fmaf(-5.f, lo.w, hi.y));
                    }
#pragma unroll
                    for (int rr = 0; rr < RPW; ++rr) {
                        const int dy = ir - rr;
                        if (dy < 0 || dy > 2) continue;
#pragma unroll
                        for (int j4 = 0; j4 < TAPS / 3; ++j4) {
                            const float* wl = lds_w + ((ks * TAPS + dy * (TAPS / 3) + j4) * NCO) * 64 + lane;
#pragma unroll
                            for (int t = 0; t < NCO; ++t) {
                                acc[rr][t][j4] = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[t * 64], d[j4], acc[rr][t][j4], 0, 0, 0);
                            }
                        }
                    }
                }
            }
            if (more) store_chunk(buf ^ 1);             // the other buffer: last read one barrier ago
            __syncthreads();
            buf ^= 1;
        }

    // ---- epilogue of this tile (its stores drain behind the next tile's first MFMA block)
    int kq_ = kq;                      // opaque per tile: keeps the bias values and output row pointers from being
    asm volatile("" : "+v"(kq_));      // hoisted out of the tile loop (they would stay live across every MFMA block)
    const int x = ex0 + 4 * cg;
    // output transform A^T m of the lane's F(4,3) tiles: u[r][g], channel 16 (t0 + t) + 4 kq + r, pixel x + g
    auto transform = [&](int rr, int t, float (&u)[4][4]) {
        const f32x4* mm = acc[rr][t];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float s12 = mm[1][r] + mm[2][r], d12 = mm[1][r] - mm[2][r], s34 = mm[3][r] + mm[4][r], d34 = mm[3][r] - mm[4][r];
            u[r][0] = (mm[0][r] + s12) + s34;
            u[r][1] = fmaf(2.f, d34, d12);
            u[r][2] = fmaf(4.f, s34, s12);
            u[r][3] = fmaf(8.f, d34, d12) + mm[5][r];
        }
    };
    // bias, activation, clamp
    auto finish = [&](int t, float (&v)[4][4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float bs = bias_l[16 * t + 4 * kq_ + r];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float u = v[r][g] + bs;
                if (a.act == 1) u = u > 0.f ? u : 0.2f * u;
                else if (a.act == 2) u = fmaxf(u, 0.f);
                else if (a.act == 3) u = gelu_fast(u);
                else if (a.act == 4) u = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * u));      // tanh
                if (a.clamp_out) u = fminf(fmaxf(u, 0.f), 1.f);
                v[r][g] = u;
            }
        }
    };
    auto store_tile = [&](int t, int y, const float (&v)[4][4]) {
        const int cobase = 16 * (t0 + t) + 4 * kq_;
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
            if (cobase >= a.Cout) return;
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
    };
    if constexpr (!KS) {
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int y = ey0 + (wave * RPW + rr) * RW + rowj;
            if (y >= h || x >= w) continue;
#pragma unroll
            for (int t = 0; t < NCO; ++t) {
                if (t0 + t >= NT) break;
                float v[4][4];   // [r][g]
                transform(rr, t, v);
                finish(t, v);
                store_tile(t, y, v);
            }
        }
    } else {
        // partial tiles [split][image][channel][y][x] (the launcher checks w % 4 == 0), then the ticket
        unsigned* counter = reinterpret_cast<unsigned*>(a.ks_scratch) + ((size_t)b * ntiles + tile) * ngroups + grp;
        float* part = a.ks_scratch + kKsCounters;
        const size_t split_stride = (size_t)a.B * a.Cout * h * w;
        float* mine = part + (size_t)kz * split_stride + (size_t)b * a.Cout * h * w;
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int y = ey0 + (wave * RPW + rr) * RW + rowj;
            if (y >= h || x >= w) continue;
#pragma unroll
            for (int t = 0; t < NCO; ++t) {
                if (t0 + t >= NT) break;
                float v[4][4];
                transform(rr, t, v);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = 16 * (t0 + t) + 4 * kq_ + r;
                    if (co < a.Cout) *reinterpret_cast<float4*>(mine + ((size_t)co * h + y) * w + x) = make_float4(v[r][0], v[r][1], v[r][2], v[r][3]);
                }
            }
        }
        __shared__ unsigned ticket;
        // release (every workgroup: its partial tile is written back before the ticket is taken) / acquire (the last workgroup
        // only: it must not read stale lines).  A full __threadfence() on both sides made EVERY workgroup invalidate its L2.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (tid == 0) ticket = atomicAdd(counter, 1u);
        __syncthreads();
        if (ticket != (unsigned)(ksplit - 1)) return;  // not the last split of this tile
        if (tid == 0) *counter = 0u;                   // ready for the next launch (stream order)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int y = ey0 + (wave * RPW + rr) * RW + rowj;
            if (y >= h || x >= w) continue;
#pragma unroll
            for (int t = 0; t < NCO; ++t) {
                if (t0 + t >= NT) break;
                float v[4][4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = 16 * (t0 + t) + 4 * kq_ + r;
                    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (co < a.Cout) {
                        const float* src = part + (size_t)b * a.Cout * h * w + ((size_t)co * h + y) * w + x;
                        for (int z = 0; z < ksplit; ++z) {          // split order, whatever the arrival order
                            const float4 pz = *reinterpret_cast<const float4*>(src + (size_t)z * split_stride);
                            sum.x += pz.x; sum.y += pz.y; sum.z += pz.z; sum.w += pz.w;
                        }
                    }
                    v[r][0] = sum.x; v[r][1] = sum.y; v[r][2] = sum.z; v[r][3] = sum.w;
                }
                finish(t, v);
                store_tile(t, y, v);
            }
        }
        return;
    }
    }
}

size_t conv3x3_ksplit_counter_bytes() { return (size_t)kKsCounters * sizeof(float); }
size_t conv3x3_ksplit_floats(int B, int Cout, int h, int w) {
    return ((long)B * h * w <= 16384) ? (size_t)kKsCounters + (size_t)8 * B * Cout * h * w : 0;
}

// input-channel split of a launch whose every tile has its own workgroup: 1 = none
static int ksplit_for(const Conv3x3Args& a, int ntiles, int ngroups, int vec, const dim3& grid) {
    if (!a.ks_scratch || !vec || !aligned16(a.ks_scratch)) return 1;
    const long total = (long)ntiles * ngroups * a.B;
    const int nchunks = cdiv(a.Cin, KC);
    // only for launches of at most 64 workgroups, split into at most 256: a 256-workgroup launch split three ways was 2.2-2.8 x
    // SLOWER (46 -> 103 us with release / acquire fences, 129 us with a full fence on both sides): the hand-over costs every
    // workgroup an L2 write-back
    if (total > 64 || total > kKsCounters || nchunks < 8 || (long)grid.x != (long)ngroups * ntiles) return 1;
    int S = nchunks / 4;
    if (S > 8) S = 8;
    while (S > 1 && total * S > 256) --S;
    if (S < 2 || (size_t)kKsCounters + (size_t)S * a.B * a.Cout * a.h * a.w > a.ks_floats) return 1;
    return S;
}

template <int NCO>
static void launch_rw(const Conv3x3Args& a, int ngroups, int vec, bool small, hipStream_t st) {
    // Wide images (w > 32): 8 waves share one 18-tap weight slice, one workgroup per CU; a wave owns one pixel
    // row of 64 (NCO >= 3) or two (NCO <= 2: 16x64 tile, 144 MFMAs per barrier).  Narrow images fall back to
    // 4-wave tiles of 32x8 or 16x16 pixels.  Workgroups are persistent over tiles of one (image, output group).
    const int lrw = a.w > 32 ? 0 : (a.w > 16 ? 1 : 2);
    constexpr int RPWB = NCO <= 2 ? 2 : 1;
    auto grid_for_tiles = [&](int ntiles) {
        const long total = (long)ntiles * ngroups * a.B;
        const int wgs = cdiv(ntiles, (int)((total + 255) / 256));      // one resident workgroup per CU
        return dim3((unsigned)(ngroups * wgs), (unsigned)a.B);
    };
    if (small && lrw == 0) {
        // few pixels (one frame at levels 2-3): 4-wave workgroups on 4x64 tiles, more of them
        const int txs = cdiv(a.w, 64), ntiles = txs * cdiv(a.h, 4);
        dim3 grid = grid_for_tiles(ntiles);
        const int S = ksplit_for(a, ntiles, ngroups, vec, grid);
        grid.z = (unsigned)S;
        if (S > 1) conv3x3_kernel<NCO, 0, 1, 4, true><<<grid, 256, 0, st>>>(a, ngroups, txs, ntiles, vec);
        else conv3x3_kernel<NCO, 0, 1, 4><<<grid, 256, 0, st>>>(a, ngroups, txs, ntiles, vec);
    } else if (NCO == 1 && lrw == 0 && a.h >= 16) {
        // one output tile: 16 waves of one row each (103 registers, four waves per SIMD cover each other's LDS and
        // barrier waits) beat 8 waves of two rows by 7 %; with NCO = 2 the same shape spills at the 128-register cap
        const int txs = cdiv(a.w, 64), ntiles = txs * cdiv(a.h, 16);
        conv3x3_kernel<1, 0, 1, 16><<<grid_for_tiles(ntiles), 1024, 0, st>>>(a, ngroups, txs, ntiles, vec);
    } else if (lrw == 0 && a.h >= 8 * RPWB) {
        const int txs = cdiv(a.w, 64), ntiles = txs * cdiv(a.h, 8 * RPWB);
        conv3x3_kernel<NCO, 0, RPWB, 8><<<grid_for_tiles(ntiles), 512, 0, st>>>(a, ngroups, txs, ntiles, vec);
    } else if (lrw == 0 && a.h >= 8) {
        const int txs = cdiv(a.w, 64), ntiles = txs * cdiv(a.h, 8);
        conv3x3_kernel<NCO, 0, 1, 8><<<grid_for_tiles(ntiles), 512, 0, st>>>(a, ngroups, txs, ntiles, vec);
    } else {
        const int txs = cdiv(a.w, 64 >> lrw), ntiles = txs * cdiv(a.h, 4 << lrw);
        dim3 grid = grid_for_tiles(ntiles);
        const int S = ksplit_for(a, ntiles, ngroups, vec, grid);
        grid.z = (unsigned)S;
        if (S > 1) {
            if (lrw == 0) conv3x3_kernel<NCO, 0, 1, 4, true><<<grid, 256, 0, st>>>(a, ngroups, txs, ntiles, vec);
            else if (lrw == 1) conv3x3_kernel<NCO, 1, 1, 4, true><<<grid, 256, 0, st>>>(a, ngroups, txs, ntiles, vec);
            else conv3x3_kernel<NCO, 2, 1, 4, true><<<grid, 256, 0, st>>>(a, ngroups, txs, ntiles, vec);
        } else if (lrw == 0) conv3x3_kernel<NCO, 0, 1, 4><<<grid, 256, 0, st>>>(a, ngroups, txs, ntiles, vec);
        else if (lrw == 1) conv3x3_kernel<NCO, 1, 1, 4><<<grid, 256, 0, st>>>(a, ngroups, txs, ntiles, vec);
        else conv3x3_kernel<NCO, 2, 1, 4><<<grid, 256, 0, st>>>(a, ngroups, txs, ntiles, vec);
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
    // A launch that would put fewer than 128 workgroups on the 256 CUs (one frame at the deep levels: 8 tiles x 4 output
    // groups at level 3) trades per-workgroup efficiency for parallelism: 4x64 tiles on 4-wave workgroups and fewer output tiles
    // per workgroup.  Batches of 8 never come here (the grid already fills the chip).
    bool small = false;
    if (a.w > 32 && a.h >= 8) {
        const long big = (long)cdiv(a.w, 64) * cdiv(a.h, nco <= 2 ? 16 : 8) * cdiv(NT, nco) * a.B;
        if (big < 128) {
            small = true;
            const long t4 = (long)cdiv(a.w, 64) * cdiv(a.h, 4) * a.B;
            while (nco > 1 && t4 * cdiv(NT, nco) < 192) nco = (nco == 4 || nco == 2) ? nco / 2 : 1;
        }
    }
    const int ngroups = cdiv(NT, nco);
    const int vec = (a.w % 4 == 0) && aligned16(a.out) && (a.out_bstride % 4 == 0);
    char key[64];
    {
        const int lrw = a.w > 32 ? 0 : (a.w > 16 ? 1 : 2), rpwb = nco <= 2 ? 2 : 1;
        const int wide = lrw == 0 && a.h >= 8;
        snprintf(key, sizeof(key), "conv3x3_kernel<%d, %d, %d, %d>", nco, lrw, (wide && a.h >= 8 * rpwb) ? rpwb : 1, wide ? 8 : 4);
        if (nco == 1 && lrw == 0 && a.h >= 16) snprintf(key, sizeof(key), "conv3x3_kernel<1, 0, 1, 16>");
        if (small) snprintf(key, sizeof(key), "conv3x3_kernel<%d, 0, 1, 4>", nco);
    }
    const double px = (double)a.B * a.h * a.w;
    ProfScope prof(st, key, 18.0 * a.Cin * a.Cout * px, 4.0 * px * (a.Cin + a.Cout));
    switch (nco) {
        case 4: launch_rw<4>(a, ngroups, vec, small, st); break;
        case 3: launch_rw<3>(a, ngroups, vec, small, st); break;
        case 2: launch_rw<2>(a, ngroups, vec, small, st); break;
        default: launch_rw<1>(a, ngroups, vec, small, st); break;
    }
    return check_launch("conv3x3");
}

}  // namespace rf
