// Backward kernels of the training step (/root/reference train.py:127-147, RawFomer_WFB_FFAB/train.py:124).
// The reference trains through torch.autograd; here every adjoint is a hand-written gfx950 kernel or one of the forward
// kernels run on transposed / flipped weights:
//
//   conv 1x1 / 3x3 / ConvTranspose2d   dX: the forward kernels on W^T (3x3: taps flipped);  dW: gram2_kernel (below)
//   depthwise 3x3                      dX: the forward kernel on flipped taps;              dW, db: dw_wgrad_kernel
//   LayerNorm over channels            ln_bwd_kernel (dx per pixel; dgamma, dbeta through fixed-order partials)
//   GELU, LeakyReLU, residual adds     element-wise kernels
//   L1 / Charbonnier loss              loss_kernel (value through fixed-order partials, gradient in the same pass)
//   Adam                               adam_kernel on the flat parameter / gradient / moment buffers
//
// gram2: G[i][j] = sum over images and pixels of A[i][p] * B[j][p + shift], the weight gradient of a convolution with A = dOut
// and B = the layer input (zero padded), one shift per tap.  Like the attention Gram it is a contraction over pixels: the pixel
// axis is the MFMA K dimension (v_mfma_f32_16x16x4_f32, f32 accumulate: reference numerics), every lane streams rows of 4
// consecutive pixels.  Pixel slabs are reduced through per-slab partials summed in a fixed order: gradients are bitwise
// reproducible run to run (no float atomics).
#include "rf_common.h"

namespace rf {

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// ---------------------------------------------------------------------------------------------
// gram2.  Workgroup = (slab of pixels of one image, tile ti of 16 A rows, group tj of TJ*16 B rows).  NTAP = 1: one shift
// (sy, sx), TJ = 4 B tiles per workgroup; NTAP = 9: the nine taps of a 3x3 window at once, TJ = 1.
// partial[((slab * NTAP + tap) * Ca + i) * Cb + j]
// ---------------------------------------------------------------------------------------------
struct Gram2Args {
    const float* a; int64_t a_bstride; int Ca;     // dOut  [B][Ca][h][w]
    const float* b; int64_t b_bstride; int Cb;     // input [B][Cb][h][w]; rows >= Cb1 come from b2 (a concatenated input read in place)
    const float* b2; int64_t b2_bstride; int Cb1;
    float* partial;
    int B, h, w, sy, sx, slab_px, slabs_per_image;
    float* bias_partial;                           // [slab][Ca] row sums of A (the bias gradient of the same layer) or nullptr
    int nslab, nty, ntz;                           // grid decomposition (gram2_block)
};

// Workgroup -> (slab, A tile group ti, B tile group tjg).  The nty * ntz tile workgroups of one slab read the same pixels of A
// and B; the hardware deals consecutive workgroup ids round-robin over the 8 XCDs (one L2 each), so the ids are arranged for the
// tiles of a slab to follow each other ON ONE XCD: its L2 serves the re-reads (with slab as the fast grid index the tiles of a
// slab ran far apart in time and every one of them came from HBM: 2.5 x the algorithmic bytes at levels 1-3).
__device__ __forceinline__ bool gram2_block(const Gram2Args& g, int* slab, int* ti, int* tjg) {
    const int T = g.nty * g.ntz, bid = blockIdx.x;
    const int tile = (bid >> 3) % T;
    *slab = (bid & 7) + 8 * (bid / (8 * T));
    *ti = tile % g.nty;
    *tjg = tile / g.nty;
    return *slab < g.nslab;
}

// NA (single-tap form only): A tiles per workgroup.  With one A tile the B rows were re-read once per 16 output rows (Ca = 96:
// 2.25 x the algorithmic bytes, 0.22 of HBM); NA = Ca / 16 (up to 6) reads both operands once.
// TJ: B tiles per workgroup (single-tap form: 4, or 2 when Cb <= 32 -- with four, half the loads and MFMAs of every level-0
// layer were spent on clamped duplicate rows).  The row sums of A -- the bias gradient of the layer whose weight gradient this
// is -- are accumulated per lane on the way (4 NA additions beside 16 NA TJ MFMAs) instead of by a second pass over dOut.
// MASK (single-tap form): a wave's 16-pixel step may straddle the end of the image (P % 16 != 0); the 3x3 form always masks.
template <int NTAP, int NA, int TJ, bool MASK = true>
__global__ void __launch_bounds__(256) gram2_kernel(Gram2Args g) {
    static_assert(NTAP == 1 || TJ == 1, "several B tiles only in the single-tap form");
    // the wave index as a SCALAR: the step loop and everything that guards a load is then a scalar branch.  With `wave` in a
    // vector register hipcc treated the (wave-uniform) loop control as divergent, wrapped every prefetch in an EXEC-masked block
    // and closed it with s_waitcnt vmcnt(2..3): the loads of the NEXT step were waited for as soon as they were issued.
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, kq = lane >> 4;
    int slab, ti, tjg;
    if (!gram2_block(g, &slab, &ti, &tjg)) return;          // padding of the slab count to a multiple of 8 (whole workgroups)
    const int img = slab / g.slabs_per_image, sl = slab - img * g.slabs_per_image;
    const int h = g.h, w = g.w, P = h * w;
    const int n_lo = sl * g.slab_px, n_hi = (n_lo + g.slab_px < P) ? n_lo + g.slab_px : P;
    const float* arow[NA];
#pragma unroll
    for (int u = 0; u < NA; ++u) {
        const int ia = 16 * (ti * NA + u) + r;
        arow[u] = g.a + (size_t)img * g.a_bstride + (size_t)(ia < g.Ca ? ia : g.Ca - 1) * P;
    }
    const float* brow[TJ];
#pragma unroll
    for (int t = 0; t < TJ; ++t) {
        const int jb = 16 * (tjg * TJ + t) + r, jc = jb < g.Cb ? jb : g.Cb - 1;
        brow[t] = jc < g.Cb1 ? g.b + (size_t)img * g.b_bstride + (size_t)jc * P : g.b2 + (size_t)img * g.b2_bstride + (size_t)(jc - g.Cb1) * P;
    }
    // single-tap form: wave-uniform tile bases + 32-bit lane offsets (the loads take the scalar-base addressing mode: one VALU
    // addition per load instead of a 64-bit pointer sum), no row / column arithmetic (the shift is zero: launch_gram2 checks), and
    // no masks when every 16-pixel step of a wave is entirely inside the image (P % 16 == 0).  The VALU instructions of a step are
    // what limited this kernel: f32 MFMA time and VALU time add up, and there were four of them for every MFMA.
    const float* abase[NA];
    unsigned aoff[NA];
    const float* bbase[TJ];
    unsigned boff[TJ];
    if constexpr (NTAP == 1) {
        const int nta = (g.Ca + 15) >> 4, ntb = (g.Cb + 15) >> 4;
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int tile = (ti * NA + u < nta) ? ti * NA + u : nta - 1;         // uniform; a clamped duplicate tile is never stored
            const int rr = (16 * tile + r < g.Ca) ? r : g.Ca - 1 - 16 * tile;
            abase[u] = g.a + (size_t)img * g.a_bstride + (size_t)(16 * tile) * P;
            aoff[u] = (unsigned)rr * (unsigned)P;
        }
#pragma unroll
        for (int t = 0; t < TJ; ++t) {
            const int tile = (tjg * TJ + t < ntb) ? tjg * TJ + t : ntb - 1;
            const int rr = (16 * tile + r < g.Cb) ? r : g.Cb - 1 - 16 * tile;
            const bool second = 16 * tile >= g.Cb1;                                // launch_gram2: Cb1 % 16 == 0 when there are two inputs
            bbase[t] = second ? g.b2 + (size_t)img * g.b2_bstride + (size_t)(16 * tile - g.Cb1) * P : g.b + (size_t)img * g.b_bstride + (size_t)(16 * tile) * P;
            boff[t] = (unsigned)rr * (unsigned)P;
        }
    }
    f32x4 acc[NTAP * NA][TJ];       // [tap * NA + A tile]
#pragma unroll
    for (int k = 0; k < NTAP * NA; ++k)
#pragma unroll
        for (int t = 0; t < TJ; ++t) acc[k][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum[NA];
#pragma unroll
    for (int u = 0; u < NA; ++u) bsum[u] = 0.f;

    // The 4 pixels of a lane sit in one row (w % 4 == 0).  A step's loads -- the A row, and per B tile three rows of (16 bytes +
    // the two outer taps) for the 3x3 window, or one shifted row -- are all UNCONDITIONAL on clamped addresses (values masked
    // afterwards) and issued one step ahead: loads behind `if (row inside the image)` branches had been followed by
    // s_waitcnt vmcnt(0) each, 15 serialised round trips per step (gram2<9> ran at 16 % of the f32 matrix rate).
    constexpr int NROW = NTAP == 9 ? 3 : 1;
    struct Step { float4 a[NA]; float4 c[TJ][NROW]; float l[TJ][NROW], rg[TJ][NROW]; bool ok; bool rok[NROW]; bool lok, rgk; };
    auto load_step = [&](Step& q, int n0) {
        const int nn = n0 + 4 * kq;
        q.ok = nn < n_hi;
        const int n = q.ok ? nn : n_lo;
        if constexpr (NTAP == 1) {
#pragma unroll
            for (int u = 0; u < NA; ++u) q.a[u] = ld4(abase[u] + (aoff[u] + (unsigned)n));
#pragma unroll
            for (int t = 0; t < TJ; ++t) q.c[t][0] = ld4(bbase[t] + (boff[t] + (unsigned)n));
            return;
        }
        const int y = n / w, x = n - y * w;
#pragma unroll
        for (int u = 0; u < NA; ++u) q.a[u] = ld4(arow[u] + n);
        q.lok = x > 0; q.rgk = x + 4 < w;
#pragma unroll
        for (int dy = 0; dy < NROW; ++dy) {
            const int yy = y + (NTAP == 9 ? dy - 1 : g.sy);
            q.rok[dy] = q.ok && yy >= 0 && yy < h;
            const size_t off = (size_t)(q.rok[dy] ? yy : 0) * w + x;
#pragma unroll
            for (int t = 0; t < TJ; ++t) {
                const float* p = brow[t] + off;
                q.c[t][dy] = ld4(p);
                if constexpr (NTAP == 9) {        // the single-tap form only shifts rows (launch_gram2 checks sx == 0)
                    q.l[t][dy] = p[q.lok ? -1 : 0];
                    q.rg[t][dy] = p[q.rgk ? 4 : 0];
                } else {
                    q.l[t][dy] = 0.f; q.rg[t][dy] = 0.f;
                }
            }
        }
    };
    auto compute = [&](const Step& q) {
        float aa[NA][4];
#pragma unroll
        for (int u = 0; u < NA; ++u) { aa[u][0] = q.a[u].x; aa[u][1] = q.a[u].y; aa[u][2] = q.a[u].z; aa[u][3] = q.a[u].w; }
        if constexpr (NTAP != 1 || MASK) {   // a zero A value silences its products: B needs no mask in the single-tap form
#pragma unroll
            for (int u = 0; u < NA; ++u)
#pragma unroll
                for (int m = 0; m < 4; ++m) aa[u][m] = q.ok ? aa[u][m] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < NA; ++u) bsum[u] += (aa[u][0] + aa[u][1]) + (aa[u][2] + aa[u][3]);
        if constexpr (NTAP == 1) {
#pragma unroll
            for (int t = 0; t < TJ; ++t) {
                const float bv[4] = {q.c[t][0].x, q.c[t][0].y, q.c[t][0].z, q.c[t][0].w};
#pragma unroll
                for (int u = 0; u < NA; ++u)
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[u][m], bv[m], acc[u][t], 0, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int t = 0; t < TJ; ++t) {
            float v[NROW][6];
#pragma unroll
            for (int dy = 0; dy < NROW; ++dy) {
                const bool rk = q.rok[dy];
                v[dy][0] = (rk && q.lok) ? q.l[t][dy] : 0.f;
                v[dy][1] = rk ? q.c[t][dy].x : 0.f; v[dy][2] = rk ? q.c[t][dy].y : 0.f;
                v[dy][3] = rk ? q.c[t][dy].z : 0.f; v[dy][4] = rk ? q.c[t][dy].w : 0.f;
                v[dy][5] = (rk && q.rgk) ? q.rg[t][dy] : 0.f;
            }
            if constexpr (NTAP == 1) {
#pragma unroll
                for (int u = 0; u < NA; ++u)
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[u][m], v[0][m + 1], acc[u][t], 0, 0, 0);
            } else {
#pragma unroll
                for (int k = 0; k < 9; ++k)
#pragma unroll
                    for (int u = 0; u < NA; ++u)
#pragma unroll
                        for (int m = 0; m < 4; ++m)
                            acc[k * NA + u][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[u][m], v[k / 3][m + k % 3], acc[k * NA + u][t], 0, 0, 0);
            }
        }
    };
    // the loop is wave-uniform: an MFMA takes its operands from ALL 64 lanes whatever EXEC says, so lanes past the end of the
    // slab stay in the loop and feed zeros (clamped address, masked value)
    // Steps of this wave: k = 0 .. nsteps - 1 at pixels n_lo + 16 wave + 64 k (scalar loop control).  The loads of step k + 1 are
    // issued UNCONDITIONALLY before step k is multiplied (past the end they re-read the slab's first pixels: clamped address,
    // result unused), so that no branch separates a load from the wait that belongs to it.
    Step s0, s1;
    const int first = n_lo + wave * 16;
    const int nsteps = first < n_hi ? (n_hi - first + 63) >> 6 : 0;
    // (the compiler barriers pin the loads where they are written: without them hipcc sinks every load_step down to the
    // compute that uses it -- load, s_waitcnt vmcnt(0), multiply -- and the prefetch is gone)
    if (nsteps > 0) load_step(s0, first);
    int k = 0;
    for (; k + 2 <= nsteps; k += 2) {           // two steps per trip, no exit in the middle: s0 / s1 stay in their registers
        load_step(s1, first + 64 * (k + 1));
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);      // ... and the machine scheduler from moving them down into the MFMA block
        compute(s0);
        load_step(s0, first + 64 * (k + 2));
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        compute(s1);
    }
    if (k < nsteps) compute(s0);                // odd step count
    __shared__ float red[4][16][17];
#pragma unroll
    for (int k = 0; k < NTAP * NA; ++k)
#pragma unroll
        for (int t = 0; t < TJ; ++t) {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) red[wave][4 * kq + q][r] = acc[k][t][q];
            __syncthreads();
            const int i = threadIdx.x >> 4, j = threadIdx.x & 15;       // 256 threads = the 16 x 16 tile
            const int tap = k / NA, u = k % NA;
            const int gi = 16 * (ti * NA + u) + i, gj = 16 * (tjg * TJ + t) + j;
            if (gi < g.Ca && gj < g.Cb)
                g.partial[(((size_t)slab * NTAP + tap) * g.Ca + gi) * g.Cb + gj] = ((red[0][i][j] + red[1][i][j]) + red[2][i][j]) + red[3][i][j];
        }
    if (g.bias_partial && tjg == 0) {      // row sums: the four pixel groups of a row, then the four waves, in a fixed order
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            float v = bsum[u];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (kq == 0) red[wave][u][r] = v;
        }
        __syncthreads();
        if (threadIdx.x < NA * 16) {
            const int u = threadIdx.x >> 4, i = threadIdx.x & 15, gi = 16 * (ti * NA + u) + i;
            if (gi < g.Ca) g.bias_partial[(size_t)slab * g.Ca + gi] = ((red[0][u][i] + red[1][u][i]) + red[2][u][i]) + red[3][u][i];
        }
    }
}

// out[e] (+)= sum over slabs of partial[slab][e], in slab order
// Sums over slabs: a block handles 32 elements x 8 slab parts (thread = (element e0 + t % 32, part t / 32) adds slabs part,
// part + 8, ... in order; the 8 part sums are combined in a fixed order through LDS), so that a reduction over a few hundred
// slabs of a small matrix is 8 x shorter per thread and fills more than a handful of workgroups.  Deterministic.
constexpr int kRedElems = 32, kRedParts = 8;
__device__ __forceinline__ float slab_sum(const float* __restrict__ base, size_t stride, int nslab, bool ok, float (&lds)[kRedParts][kRedElems]) {
    const int el = threadIdx.x % kRedElems, part = threadIdx.x / kRedElems;
    float s = 0.f;
    if (ok)
        for (int k = part; k < nslab; k += kRedParts) s += base[(size_t)k * stride];
    lds[part][el] = s;
    __syncthreads();
    float t = 0.f;
    if (part == 0) {
#pragma unroll
        for (int q = 0; q < kRedParts; ++q) t += lds[q][el];
    }
    __syncthreads();
    return t;       // valid in part 0
}

__global__ void __launch_bounds__(256) reduce_partials_kernel(const float* __restrict__ partial, float* __restrict__ out, int nslab, size_t n, int accumulate) {
    __shared__ float lds[kRedParts][kRedElems];
    for (size_t e0 = (size_t)blockIdx.x * kRedElems; e0 < n; e0 += (size_t)gridDim.x * kRedElems) {
        const size_t e = e0 + threadIdx.x % kRedElems;
        const float s = slab_sum(partial + (e < n ? e : 0), n, nslab, e < n, lds);
        if (threadIdx.x < kRedElems && e < n) out[e] = accumulate ? out[e] + s : s;
    }
}

// gram2 partials [group][slab_in_group][tap][Ca][Cb] -> out[group][(i * ld + j) * ntap + tap]   (weight layout [Cout][Cin][taps];
// group = image when the result is wanted per image, else one group over all slabs)
// bias_partial / db (optional, one group only): db[i] (+)= sum over slabs of bias_partial[slab][i], handled by the blocks that
// follow the weight elements
__global__ void __launch_bounds__(256) reduce_gram2_kernel(const float* __restrict__ partial, float* __restrict__ out, int nslab_per_group, int ntap,
                                                           int Ca, int Cb, int ld, size_t out_gstride, int accumulate,
                                                           const float* __restrict__ bias_partial, float* __restrict__ db) {
    __shared__ float lds[kRedParts][kRedElems];
    const size_t n = (size_t)ntap * Ca * Cb, nb = db ? (size_t)Ca : 0, n_pad = (n + kRedElems - 1) / kRedElems * kRedElems;
    const int grp = blockIdx.y;
    const float* pg = partial + (size_t)grp * nslab_per_group * n;
    for (size_t e0 = (size_t)blockIdx.x * kRedElems; e0 < n_pad + nb; e0 += (size_t)gridDim.x * kRedElems) {
        const size_t e = e0 + threadIdx.x % kRedElems;
        if (e0 >= n_pad) {                                   // block-uniform: a block of bias elements
            const size_t i = e - n_pad;
            const float s = slab_sum(bias_partial + (i < nb ? i : 0), (size_t)Ca, nslab_per_group, i < nb, lds);
            if (threadIdx.x < kRedElems && i < nb) db[i] = accumulate ? db[i] + s : s;
            continue;
        }
        const float s = slab_sum(pg + (e < n ? e : 0), n, nslab_per_group, e < n, lds);
        if (threadIdx.x < kRedElems && e < n) {
            const int j = (int)(e % Cb), i = (int)((e / Cb) % Ca), tap = (int)(e / ((size_t)Ca * Cb));
            float* o = out + (size_t)grp * out_gstride + ((size_t)i * ld + j) * ntap + tap;
            *o = accumulate ? *o + s : s;
        }
    }
}

// depthwise partials [(img, blk)][C][10] -> dw[c][9] and db[c]
__global__ void __launch_bounds__(256) reduce_dw_kernel(const float* __restrict__ partial, float* __restrict__ dw, float* __restrict__ db, int nslab, int C, int accumulate) {
    __shared__ float lds[kRedParts][kRedElems];
    const int n = C * 10;
    for (int e0 = blockIdx.x * kRedElems; e0 < n; e0 += gridDim.x * kRedElems) {
        const int e = e0 + threadIdx.x % kRedElems;
        const float s = slab_sum(partial + (e < n ? e : 0), (size_t)n, nslab, e < n, lds);
        if (threadIdx.x < kRedElems && e < n) {
            const int c = e / 10, t = e % 10;
            float* o = t < 9 ? dw + c * 9 + t : (db ? db + c : nullptr);
            if (o) *o = accumulate ? *o + s : s;
        }
    }
}

// channel sums (bias gradients): partial[(img * nblk + blk) * C + c]
__global__ void __launch_bounds__(256) chan_sum_kernel(const float* __restrict__ x, int64_t bstride, float* __restrict__ partial, int C, int P, int nblk) {
    const int blk = blockIdx.x, c = blockIdx.y, img = blockIdx.z;
    const float* row = x + (size_t)img * bstride + (size_t)c * P;
    const int per = ((P + nblk - 1) / nblk + 3) & ~3;
    const int lo = blk * per, hi = (lo + per < P) ? lo + per : P;
    float s = 0.f;
    if ((P & 3) == 0 && (((size_t)row) & 15) == 0) {        // 16 bytes per lane, four loads in flight
        float s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int p = lo + 4 * threadIdx.x;
        for (; p + 3 * 1024 < hi; p += 4 * 1024) {
            const float4 a = ld4(row + p), b = ld4(row + p + 1024), cc = ld4(row + p + 2048), d = ld4(row + p + 3072);
            s += (a.x + a.y) + (a.z + a.w); s1 += (b.x + b.y) + (b.z + b.w); s2 += (cc.x + cc.y) + (cc.z + cc.w); s3 += (d.x + d.y) + (d.z + d.w);
        }
        for (; p < hi; p += 1024) { const float4 a = ld4(row + p); s += (a.x + a.y) + (a.z + a.w); }
        s = (s + s1) + (s2 + s3);
    } else {
        for (int p = lo + threadIdx.x; p < hi; p += 256) s += row[p];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    __shared__ float wsum[4];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[((size_t)img * nblk + blk) * C + c] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over channels, backward.
//   ln_bwd_kernel      one thread per pixel (lanes along pixels: every channel row is a coalesced stream):
//                      dx = rstd (g - mean_c(g) - xhat mean_c(g xhat)),  g = dy gamma;  keeps mu and rstd of its pixel
//   ln_wgrad_kernel    workgroup = (pixel block, channel, image): dgamma[c] = sum_p dy xhat, dbeta[c] = sum_p dy with ONE block
//                      reduction per workgroup;  partial[((img * nblk + blk) * 2 + {0,1}) * C + c]
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ gamma,
                                                     float* __restrict__ dx, float* __restrict__ stats, int C, int P, float eps, int accumulate_dx) {
    const int img = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const size_t base = (size_t)img * C * P + p;
    float mu = 0.f;
    for (int c = 0; c < C; ++c) mu += x[base + (size_t)c * P];
    mu /= (float)C;
    float var = 0.f;
    for (int c = 0; c < C; ++c) { const float d = x[base + (size_t)c * P] - mu; var = fmaf(d, d, var); }
    const float rstd = 1.0f / sqrtf(var / (float)C + eps);
    float s1 = 0.f, s2 = 0.f;
    for (int c = 0; c < C; ++c) {
        const float g = dy[base + (size_t)c * P] * gamma[c];
        s1 += g;
        s2 = fmaf(g, (x[base + (size_t)c * P] - mu) * rstd, s2);
    }
    s1 /= (float)C; s2 /= (float)C;
    for (int c = 0; c < C; ++c) {
        const float xh = (x[base + (size_t)c * P] - mu) * rstd;
        const float v = rstd * (dy[base + (size_t)c * P] * gamma[c] - s1 - xh * s2);
        dx[base + (size_t)c * P] = accumulate_dx ? dx[base + (size_t)c * P] + v : v;
    }
    stats[(size_t)img * 2 * P + p] = mu;
    stats[(size_t)img * 2 * P + P + p] = rstd;
}

__global__ void __launch_bounds__(256) ln_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ stats,
                                                       float* __restrict__ partial, int C, int P, int nblk) {
    const int blk = blockIdx.x, c = blockIdx.y, img = blockIdx.z;
    const float* xr = x + ((size_t)img * C + c) * P;
    const float* dr = dy + ((size_t)img * C + c) * P;
    const float* mu = stats + (size_t)img * 2 * P;
    const float* rs = mu + P;
    const int per = ((P + nblk - 1) / nblk + 3) & ~3, lo = blk * per, hi = (lo + per < P) ? lo + per : P;
    float a = 0.f, b = 0.f;
    for (int p = lo + threadIdx.x; p < hi; p += 256) {
        const float d = dr[p];
        a = fmaf(d, (xr[p] - mu[p]) * rs[p], a);
        b += d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    __shared__ float red[2][4];
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x < 2)
        partial[(((size_t)img * nblk + blk) * 2 + threadIdx.x) * C + c] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// Register-resident fused form (C = CPL * KS <= 128; the layout of layernorm2d_reg_kernel, rf_pointwise.hip): a lane keeps its
// CPL channels of x AND dy for its 4 pixels, so x and dy are read ONCE and dx written once, and the per-channel sums of
// dgamma / dbeta accumulate in registers over the workgroup's pixel chunks -- one butterfly + LDS reduction per workgroup at
// the end (the per-pixel kernel above walks the channels four times with one dependent scalar load in flight, and the weight
// pass re-reads x and dy once more: 0.18 of HBM).   partial[((img * nblk + blk) * 2 + {0,1}) * C + c], as ln_wgrad_kernel.
// `res` (may be null): dx = res + (LayerNorm adjoint); res may be a strided view (res_bstride) -- the residual branch of
// TransformerBlock without a separate add pass.
template <int CPL, int KS>
__global__ void __launch_bounds__(256) ln_bwd_reg_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ gamma,
                                                         const float* __restrict__ res, int64_t res_bstride, float* __restrict__ dx,
                                                         float* __restrict__ partial, int C, int P, float eps, int accumulate_dx) {
    constexpr int G = 64 / KS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane % G, ks = lane / G;
    const size_t img = blockIdx.y;
    const float* xb = x + img * (size_t)C * P;
    const float* db = dy + img * (size_t)C * P;
    const float* rb = res ? res + img * (size_t)res_bstride : nullptr;
    float* ob = dx + img * (size_t)C * P;
    const int ngroups = P / 4, per_wg = 4 * G;
    float gam[CPL], ag[CPL], ab[CPL];
#pragma unroll
    for (int s = 0; s < CPL; ++s) { gam[s] = gamma[ks + KS * s]; ag[s] = 0.f; ab[s] = 0.f; }
    const float invC = 1.0f / (float)C;
    for (int g0 = blockIdx.x * per_wg; g0 < ngroups; g0 += gridDim.x * per_wg) {
        const int gi = g0 + wave * G + j;
        const bool ok = gi < ngroups;
        const size_t off = (size_t)(ok ? gi : 0) * 4;
        float4 xv[CPL], dv[CPL];
#pragma unroll
        for (int s = 0; s < CPL; ++s) {
            xv[s] = *reinterpret_cast<const float4*>(xb + (size_t)(ks + KS * s) * P + off);
            dv[s] = *reinterpret_cast<const float4*>(db + (size_t)(ks + KS * s) * P + off);
        }
        float mu[4] = {0.f, 0.f, 0.f, 0.f}, var[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < CPL; ++s) { mu[0] += xv[s].x; mu[1] += xv[s].y; mu[2] += xv[s].z; mu[3] += xv[s].w; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int o = G; o < 64; o <<= 1) mu[q] += __shfl_xor(mu[q], o);
            mu[q] *= invC;
        }
#pragma unroll
        for (int s = 0; s < CPL; ++s) {
            xv[s].x -= mu[0]; xv[s].y -= mu[1]; xv[s].z -= mu[2]; xv[s].w -= mu[3];
            var[0] = fmaf(xv[s].x, xv[s].x, var[0]); var[1] = fmaf(xv[s].y, xv[s].y, var[1]);
            var[2] = fmaf(xv[s].z, xv[s].z, var[2]); var[3] = fmaf(xv[s].w, xv[s].w, var[3]);
        }
        float rstd[4], s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int o = G; o < 64; o <<= 1) var[q] += __shfl_xor(var[q], o);
            rstd[q] = 1.0f / sqrtf(var[q] * invC + eps);
        }
        const float m = ok ? 1.0f : 0.f;              // lanes past the end contribute nothing to the channel sums
#pragma unroll
        for (int s = 0; s < CPL; ++s) {
            // xhat in place; channel sums of dy xhat and dy; g = dy gamma
            xv[s].x *= rstd[0]; xv[s].y *= rstd[1]; xv[s].z *= rstd[2]; xv[s].w *= rstd[3];
            ag[s] += m * (fmaf(dv[s].x, xv[s].x, fmaf(dv[s].y, xv[s].y, fmaf(dv[s].z, xv[s].z, dv[s].w * xv[s].w))));
            ab[s] += m * ((dv[s].x + dv[s].y) + (dv[s].z + dv[s].w));
            dv[s].x *= gam[s]; dv[s].y *= gam[s]; dv[s].z *= gam[s]; dv[s].w *= gam[s];
            s1[0] += dv[s].x; s1[1] += dv[s].y; s1[2] += dv[s].z; s1[3] += dv[s].w;
            s2[0] = fmaf(dv[s].x, xv[s].x, s2[0]); s2[1] = fmaf(dv[s].y, xv[s].y, s2[1]);
            s2[2] = fmaf(dv[s].z, xv[s].z, s2[2]); s2[3] = fmaf(dv[s].w, xv[s].w, s2[3]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int o = G; o < 64; o <<= 1) { s1[q] += __shfl_xor(s1[q], o); s2[q] += __shfl_xor(s2[q], o); }
            s1[q] *= invC; s2[q] *= invC;
        }
        if (ok) {
#pragma unroll
            for (int s = 0; s < CPL; ++s) {
                float4 v;
                v.x = rstd[0] * (dv[s].x - s1[0] - xv[s].x * s2[0]); v.y = rstd[1] * (dv[s].y - s1[1] - xv[s].y * s2[1]);
                v.z = rstd[2] * (dv[s].z - s1[2] - xv[s].z * s2[2]); v.w = rstd[3] * (dv[s].w - s1[3] - xv[s].w * s2[3]);
                float* o = ob + (size_t)(ks + KS * s) * P + off;
                if (rb) { const float4 r = *reinterpret_cast<const float4*>(rb + (size_t)(ks + KS * s) * P + off); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
                if (accumulate_dx) { const float4 r = *reinterpret_cast<const float4*>(o); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
                *reinterpret_cast<float4*>(o) = v;
            }
        }
    }
    // channel sums: over the G pixel-group lanes of a slice (butterfly), then over the four waves in a fixed order
    __shared__ float red[4][2][CPL * KS];
#pragma unroll
    for (int s = 0; s < CPL; ++s) {
#pragma unroll
        for (int o = 1; o < G; o <<= 1) { ag[s] += __shfl_xor(ag[s], o); ab[s] += __shfl_xor(ab[s], o); }
        if (j == 0) { red[wave][0][ks + KS * s] = ag[s]; red[wave][1][ks + KS * s] = ab[s]; }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * C; e += 256) {
        const int k = e / C, c = e % C;
        partial[((img * gridDim.x + blockIdx.x) * 2 + k) * C + c] = (red[0][k][c] + red[1][k][c]) + (red[2][k][c] + red[3][k][c]);
    }
}

// ---------------------------------------------------------------------------------------------
// depthwise 3x3: weight and bias gradients.  partial[((img * nblk + blk) * C + c) * 10 + {9 taps, bias}]
// ---------------------------------------------------------------------------------------------
// VEC = 4 (w % 4 == 0): a lane owns a 4-pixel x 8-row strip like dwconv3x3_kernel<4, 8> -- 8 rows of dy and 10 rows of x
// (16-byte loads + the two edge taps, all unconditional on clamped addresses) feed the nine tap sums; VEC = 1: one pixel per
// lane with predicated neighbour loads (odd widths).
template <int VEC>
__global__ void __launch_bounds__(256) dw_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ partial,
                                                       int C, int h, int w, int nblk) {
    const int blk = blockIdx.x, c = blockIdx.y, img = blockIdx.z;
    const int P = h * w;
    const float* xr = x + ((size_t)img * C + c) * P;
    const float* dr = dy + ((size_t)img * C + c) * P;
    float s[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) s[k] = 0.f;
    if constexpr (VEC == 4) {
        constexpr int ROWS = 8;
        const int wv = w / 4, hr = (h + ROWS - 1) / ROWS, items = wv * hr;
        const int per = (items + nblk - 1) / nblk;
        const int lo = blk * per, hi = (lo + per < items) ? lo + per : items;
        for (int it = lo + threadIdx.x; it < hi; it += 256) {
            const int x0 = (it % wv) * 4, y0 = (it / wv) * ROWS;
            float d[ROWS][4];
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                const bool ok = y0 + r < h;
                const float4 t = *reinterpret_cast<const float4*>(dr + (size_t)(ok ? y0 + r : 0) * w + x0);
                d[r][0] = ok ? t.x : 0.f; d[r][1] = ok ? t.y : 0.f; d[r][2] = ok ? t.z : 0.f; d[r][3] = ok ? t.w : 0.f;
                s[9] += (d[r][0] + d[r][1]) + (d[r][2] + d[r][3]);
            }
#pragma unroll
            for (int rr = 0; rr < ROWS + 2; ++rr) {
                const int y = y0 + rr - 1;
                const bool rok = y >= 0 && y < h;
                const float* row = xr + (size_t)(rok ? y : 0) * w;
                const float4 t = *reinterpret_cast<const float4*>(row + x0);
                const float l = row[x0 > 0 ? x0 - 1 : 0], rg = row[x0 + 4 < w ? x0 + 4 : x0];
                const float v[6] = {(rok && x0 > 0) ? l : 0.f, rok ? t.x : 0.f, rok ? t.y : 0.f, rok ? t.z : 0.f, rok ? t.w : 0.f,
                                    (rok && x0 + 4 < w) ? rg : 0.f};
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    const int ky = rr - r;          // input row rr is tap row ky of output row r
                    if (ky >= 0 && ky < 3) {
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                            for (int q = 0; q < 4; ++q) s[ky * 3 + kx] = fmaf(d[r][q], v[q + kx], s[ky * 3 + kx]);
                    }
                }
            }
        }
    } else {
        const int per = (P + nblk - 1) / nblk;
        const int lo = blk * per, hi = (lo + per < P) ? lo + per : P;
        for (int p = lo + threadIdx.x; p < hi; p += 256) {
            const int y = p / w, xx = p - y * w;
            const float d = dr[p];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int yy = y + k / 3 - 1, xc = xx + k % 3 - 1;
                if (yy >= 0 && yy < h && xc >= 0 && xc < w) s[k] = fmaf(d, xr[(size_t)yy * w + xc], s[k]);
            }
            s[9] += d;
        }
    }
    __shared__ float red[10][4];
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        float v = s[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 10)
        partial[(((size_t)img * nblk + blk) * C + c) * 10 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// ---------------------------------------------------------------------------------------------
// element-wise
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_grad(float v) {      // d/dv [ v Phi(v) ] = Phi(v) + v phi(v)
    const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
    return cdf + v * 0.39894228040143267794f * expf(-0.5f * v * v);
}
// mode 0: out = a + b;  1: out = dy * gelu'(x) (a = dy, b = x);  2: out = dy * (y > 0 ? 1 : slope) (a = dy, b = y);
// 3: out = gelu(a) (exact erf);  4: out += a;  5: out = a
__device__ __forceinline__ float ewise_op(float a, float b, float o, int mode, float slope) {
    if (mode == 0) return a + b;
    if (mode == 1) return a * gelu_grad(b);
    if (mode == 2) return a * (b > 0.f ? 1.0f : slope);
    if (mode == 3) return 0.5f * a * (1.0f + erff(a * 0.70710678118654752440f));
    if (mode == 5) return a;
    return o + a;
}
// VEC = 4: 16 bytes per lane and access (n % 4 == 0, 16-byte aligned operands); VEC = 1: any n
template <int VEC>
__global__ void __launch_bounds__(256) ewise_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, size_t n, int mode, float slope) {
    const bool need_b = mode <= 2, need_o = mode == 4;
    for (size_t i = (blockIdx.x * 256ull + threadIdx.x) * VEC; i < n; i += (size_t)gridDim.x * 256 * VEC) {
        if constexpr (VEC == 4) {
            const float4 av = *reinterpret_cast<const float4*>(a + i);
            const float4 bv = need_b ? *reinterpret_cast<const float4*>(b + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 ov = need_o ? *reinterpret_cast<const float4*>(out + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(out + i) = make_float4(ewise_op(av.x, bv.x, ov.x, mode, slope), ewise_op(av.y, bv.y, ov.y, mode, slope),
                                                              ewise_op(av.z, bv.z, ov.z, mode, slope), ewise_op(av.w, bv.w, ov.w, mode, slope));
        } else {
            out[i] = ewise_op(a[i], need_b ? b[i] : 0.f, need_o ? out[i] : 0.f, mode, slope);
        }
    }
}

// [B][2C][P] -> the two channel halves as contiguous tensors [B][C][P] (one pass; replaces 2 B device-to-device copies)
__global__ void __launch_bounds__(256) split_halves_kernel(const float4* __restrict__ src, float4* __restrict__ a, float4* __restrict__ b, size_t half4, size_t total4) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const size_t img = i / (2 * half4), r = i - img * 2 * half4;
        const float4 v = src[i];
        if (r < half4) a[img * half4 + r] = v; else b[img * half4 + (r - half4)] = v;
    }
}

// w'[c][8 - t] = w[c][t] (depthwise);  w'[ci][co][2-dy][2-dx] = w[co][ci][dy][dx] (dense 3x3)
__global__ void __launch_bounds__(256) flip_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int dense) {
    const size_t total = (size_t)Cout * (dense ? Cin : 1) * 9;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int t = (int)(i % 9);
        if (!dense) { out[i - t + 8 - t] = w[i]; continue; }
        const int ci = (int)((i / 9) % Cin), co = (int)(i / ((size_t)9 * Cin));
        out[((size_t)ci * Cout + co) * 9 + 8 - t] = w[i];
    }
}

// loss (mode 0: L1 = mean |d|; 1: Charbonnier = mean sqrt(d^2 + eps^2), train.py:16-25) and its gradient w.r.t. pred
__global__ void __launch_bounds__(256) loss_kernel(const float* __restrict__ pred, const float* __restrict__ gt, float* __restrict__ grad,
                                                   float* __restrict__ partial, size_t n, int mode, float eps, float inv_n) {
    float s = 0.f;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float d = pred[i] - gt[i];
        if (mode == 0) { s += fabsf(d); grad[i] = (d > 0.f ? inv_n : d < 0.f ? -inv_n : 0.f); }
        else { const float r = sqrtf(fmaf(d, d, eps * eps)); s += r; grad[i] = d / r * inv_n; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    __shared__ float wsum[4];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = ((wsum[0] + wsum[1]) + (wsum[2] + wsum[3])) * inv_n;
}

// torch.optim.Adam (train.py:113; weight_decay > 0 with decoupled = 1 gives AdamW), one launch over the flat buffers
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
                                                   float lr, float b1, float b2, float eps, float wd, int decoupled, float bc1, float bc2, float gscale) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float gi = g[i] * gscale, pi = p[i];
        if (wd != 0.f) { if (decoupled) pi *= 1.0f - lr * wd; else gi = fmaf(wd, pi, gi); }
        const float mi = fmaf(b1, m[i], (1.0f - b1) * gi);
        const float vi = fmaf(b2, v[i], (1.0f - b2) * gi * gi);
        m[i] = mi; v[i] = vi;
        p[i] = pi - lr / bc1 * mi / (sqrtf(vi) / sqrtf(bc2) + eps);
    }
}

int red_grid(size_t n) {          // blocks of the slab reducers: 32 elements each
    size_t g = (n + kRedElems - 1) / kRedElems;
    return (int)(g > 2048 ? 2048 : g < 1 ? 1 : g);
}

int grid1d(size_t n, int cap = 4096) {
    int g = (int)((n + 255) / 256);
    if (g > cap) g = cap;
    return g < 1 ? 1 : g;
}

}  // namespace

// ---- launchers (shared with the training schedule, rf_trainstep.hip) --------------------------------------------
// slabs: pixels per slab a multiple of 64, about 1024 workgroups in flight
static void gram2_slabs(int B, int P, int tiles, int* slab_px, int* per_image) {
    int per = 1;
    while ((long)B * per * tiles < 1024 && P / (per + 1) >= 2048) ++per;
    int px = (P + per - 1) / per;
    px = (px + 63) / 64 * 64;
    *slab_px = px;
    *per_image = (P + px - 1) / px;
}

// A tiles per workgroup of the single-tap form: all of Ca when that is at most 6 tiles, else the divisor that leaves fewest groups
static int gram2_na(int Ca, int ntap) {
    // 3x3 form: two A tiles share the nine shifted B operands of a step (the row loads, masks and selects are per B tile)
    if (ntap != 1) return cdiv(Ca, 16) % 2 == 0 ? 2 : 1;
    const int nt = cdiv(Ca, 16);
    if (nt <= 4 || nt == 6) return nt;
    if (nt % 6 == 0) return 6;
    if (nt % 4 == 0) return 4;
    if (nt % 3 == 0) return 3;
    return nt % 2 == 0 ? 2 : 1;
}

static int gram2_tj(int Cb, int ntap) { return ntap != 1 ? 1 : Cb <= 32 ? 2 : 4; }

size_t gram2_partial_floats(int B, int Ca, int Cb, int h, int w, int ntap) {
    int px, per;
    gram2_slabs(B, h * w, cdiv(Ca, 16 * gram2_na(Ca, ntap)) * cdiv(Cb, 16 * gram2_tj(Cb, ntap)), &px, &per);
    return (size_t)B * per * ((size_t)ntap * Ca * Cb + Ca);      // + the row sums of A (bias gradient)
}

// out[(i * ld + j) * ntap + tap] = weight layout [Ca][ld >= Cb][taps] (ntap = 9: the 3x3 window in (dy, dx) row-major order;
// ntap = 1: the single shift (sy, sx)); per_image: out[b * out_istride + ...] without the sum over images; accumulate adds
int launch_gram2(const float* a, int64_t a_bstride, int Ca, const float* b, int64_t b_bstride, int Cb, float* out, int ld, float* partial,
                 int B, int h, int w, int ntap, int sy, int sx, int per_image, size_t out_istride, int accumulate, hipStream_t st, float* db,
                 const float* b2, int64_t b2_bstride, int Cb2) {
    RF_CHECK_ARG(w % 4 == 0 && aligned16(a) && aligned16(b) && a_bstride % 4 == 0 && b_bstride % 4 == 0,
                 "gram2: width %d must be a multiple of 4 and the operands 16-byte aligned", w);
    RF_CHECK_ARG(ntap == 1 || ntap == 9, "gram2: ntap must be 1 or 9");
    RF_CHECK_ARG(ntap == 9 || (sx == 0 && sy == 0), "gram2: the single-tap form takes no shift (sy = %d, sx = %d)", sy, sx);
    RF_CHECK_ARG(Cb2 == 0 || Cb % 16 == 0, "gram2: a second input needs the first to hold a multiple of 16 channels (%d)", Cb);
    RF_CHECK_ARG(!db || !per_image, "gram2: the bias gradient is a sum over all images");
    RF_CHECK_ARG(Cb2 == 0 || (b2 && aligned16(b2) && b2_bstride % 4 == 0), "gram2: bad second input");
    const int Cb1 = Cb;
    Cb += Cb2;                                      // rows [Cb1, Cb1 + Cb2) of the input are b2's channels
    Gram2Args g{a, a_bstride, Ca, b, b_bstride, Cb, Cb2 ? b2 : b, Cb2 ? b2_bstride : b_bstride, Cb1, partial, B, h, w, sy, sx, 0, 0, nullptr, 0, 0, 0};
    // P % 16 != 0 (no frame of the reference's pipeline): the masked single-tap instantiations exist for one A tile only
    const bool masked = ntap == 1 && ((h * w) & 15) != 0;
    const int na = masked ? 1 : gram2_na(Ca, ntap), tj = gram2_tj(Cb, ntap);
    gram2_slabs(B, h * w, cdiv(Ca, 16 * na) * cdiv(Cb, 16 * tj), &g.slab_px, &g.slabs_per_image);
    const int nslab = B * g.slabs_per_image;
    g.nslab = nslab; g.nty = cdiv(Ca, 16 * na); g.ntz = cdiv(Cb, 16 * tj);
    const size_t n = (size_t)ntap * Ca * Cb;
    if (db) g.bias_partial = partial + (size_t)nslab * n;
    ProfScope prof(st, ntap == 1 ? "gram2_kernel<1>" : "gram2_kernel<9>", 2.0 * ntap * Ca * Cb * (double)B * h * w, 4.0 * (double)B * h * w * (Ca + Cb));
    const dim3 grid1((unsigned)(cdiv(nslab, 8) * 8 * g.nty * g.ntz));
#define RF_G2(NA_, TJ_) gram2_kernel<1, NA_, TJ_, false><<<grid1, 256, 0, st>>>(g)
#define RF_G2_TJ(NA_) do { if (tj == 2) RF_G2(NA_, 2); else RF_G2(NA_, 4); } while (0)
    if (ntap == 9 && na == 2) gram2_kernel<9, 2, 1><<<grid1, 256, 0, st>>>(g);
    else if (ntap == 9) gram2_kernel<9, 1, 1><<<grid1, 256, 0, st>>>(g);
    else if (masked && tj == 2) gram2_kernel<1, 1, 2, true><<<grid1, 256, 0, st>>>(g);
    else if (masked) gram2_kernel<1, 1, 4, true><<<grid1, 256, 0, st>>>(g);
    else if (na == 6) RF_G2_TJ(6);
    else if (na == 4) RF_G2_TJ(4);
    else if (na == 3) RF_G2_TJ(3);
    else if (na == 2) RF_G2_TJ(2);
    else RF_G2_TJ(1);
#undef RF_G2_TJ
#undef RF_G2
    const size_t n_pad = (n + kRedElems - 1) / kRedElems * kRedElems;
    reduce_gram2_kernel<<<dim3((unsigned)red_grid(n_pad + (db ? Ca : 0)), per_image ? (unsigned)B : 1u), 256, 0, st>>>(
        partial, out, per_image ? g.slabs_per_image : nslab, ntap, Ca, Cb, ld, out_istride, accumulate, g.bias_partial, db);
    return check_launch("gram2");
}

// out[e] (+)= sum over rows of partial[row][e]
int launch_reduce_rows(const float* partial, float* out, int nrows, size_t n, int accumulate, hipStream_t st) {
    reduce_partials_kernel<<<red_grid(n), 256, 0, st>>>(partial, out, nrows, n, accumulate);
    return check_launch("reduce_rows");
}

int chan_sum_nblk(int P) { int n = P / 16384; return n < 1 ? 1 : n > 16 ? 16 : n; }

int launch_chan_sum(const float* x, int64_t bstride, float* out, float* partial, int B, int C, int P, int accumulate, hipStream_t st) {
    const int nblk = chan_sum_nblk(P);
    chan_sum_kernel<<<dim3((unsigned)nblk, (unsigned)C, (unsigned)B), 256, 0, st>>>(x, bstride, partial, C, P, nblk);
    reduce_partials_kernel<<<red_grid(C), 256, 0, st>>>(partial, out, B * nblk, (size_t)C, accumulate);
    return check_launch("chan_sum");
}

// dgb = [dgamma (C) | dbeta (C)], contiguous (the two LayerNorm parameters are neighbours in the flat gradient buffer);
// partial: ln_bwd_partial_floats floats = per-pixel statistics + the per-block sums.  dx must not alias dy (the weight pass
// reads dy after dx has been written).
static int ln_nblk(int P) { int n = P / 4096; return n < 1 ? 1 : n > 64 ? 64 : n; }
static int ln_reg_nblk(int B, int P, int per_wg) {
    int gx = cdiv(P / 4, per_wg);
    const int cap = cdiv(256 * 4, B);              // about four workgroups per CU over the batch: each loops over its pixel chunks
    return gx > cap ? cap : gx;
}
template <int CPL, int KS>
static int launch_ln_bwd_reg(const float* x, const float* dy, const float* gamma, const float* res, int64_t res_bstride, float* dx, float* dgb,
                             float* partial, int B, int C, int P, float eps, int accumulate_dx, int accumulate_w, hipStream_t st) {
    const int nblk = ln_reg_nblk(B, P, 4 * (64 / KS));
    ln_bwd_reg_kernel<CPL, KS><<<dim3((unsigned)nblk, (unsigned)B), 256, 0, st>>>(x, dy, gamma, res, res_bstride, dx, partial, C, P, eps, accumulate_dx);
    reduce_partials_kernel<<<red_grid(2 * C), 256, 0, st>>>(partial, dgb, B * nblk, (size_t)2 * C, accumulate_w);
    return check_launch("ln_bwd");
}
// channel counts of the register-resident kernel (which also takes a strided residual); anything else runs the generic path
bool ln_bwd_fused_shape(int C, int P) {
    switch (C) {
        case 16: case 32: case 48: case 64: case 96: case 128: case 192: case 256: case 384: case 512: return (P & 3) == 0;
        default: return false;
    }
}
int launch_ln_bwd(const float* x, const float* dy, const float* gamma, float* dx, float* dgb, float* partial,
                  int B, int C, int P, float eps, int accumulate_dx, int accumulate_w, hipStream_t st, const float* res, int64_t res_bstride) {
    ProfScope prof(st, "ln_bwd", 14.0 * B * C * P, (res ? 16.0 : 12.0) * B * C * P);
    if ((P & 3) == 0 && aligned16(x) && aligned16(dy) && aligned16(dx) && (!res || (aligned16(res) && res_bstride % 4 == 0)) && B <= 65535) {
#define RF_LN_BWD(CPL, KS) return launch_ln_bwd_reg<CPL, KS>(x, dy, gamma, res, res_bstride, dx, dgb, partial, B, C, P, eps, accumulate_dx, accumulate_w, st)
        switch (C) {
            case 16: RF_LN_BWD(4, 4);
            case 32: RF_LN_BWD(8, 4);
            case 48: RF_LN_BWD(12, 4);
            case 64: RF_LN_BWD(16, 4);
            case 96: RF_LN_BWD(12, 8);
            case 128: RF_LN_BWD(16, 8);
            case 192: RF_LN_BWD(12, 16);
            case 256: RF_LN_BWD(16, 16);
            case 384: RF_LN_BWD(12, 32);
            case 512: RF_LN_BWD(16, 32);
            default: break;
        }
#undef RF_LN_BWD
    }
    // generic channel counts (level 3: C = 256 ...): per-pixel kernel + weight pass; the residual is added by a separate pass
    const int nblk = ln_nblk(P);
    float* stats = partial;
    float* sums = partial + align_up((size_t)B * 2 * P, 64);
    if (res) {
        RF_CHECK_ARG(res_bstride == (int64_t)C * P, "ln_bwd: the generic path takes a contiguous residual");
        // dx = res (or dx += res), then the adjoint accumulates
        if (int rc = launch_ewise(res, accumulate_dx ? dx : nullptr, dx, (size_t)B * C * P, accumulate_dx ? 0 : 5, 0.f, st)) return rc;
        accumulate_dx = 1;
    }
    ln_bwd_kernel<<<dim3((unsigned)cdiv(P, 256), (unsigned)B), 256, 0, st>>>(x, dy, gamma, dx, stats, C, P, eps, accumulate_dx);
    ln_wgrad_kernel<<<dim3((unsigned)nblk, (unsigned)C, (unsigned)B), 256, 0, st>>>(x, dy, stats, sums, C, P, nblk);
    reduce_partials_kernel<<<red_grid(2 * C), 256, 0, st>>>(sums, dgb, B * nblk, (size_t)2 * C, accumulate_w);
    return check_launch("ln_bwd");
}
size_t ln_bwd_partial_floats(int B, int C, int P) {
    const size_t generic = align_up((size_t)B * 2 * P, 64) + (size_t)B * ln_nblk(P) * 2 * C;
    const size_t reg = (size_t)B * 1024 * 2 * C;
    return generic > reg ? generic : reg;
}

int dw_wgrad_nblk(int P) { int n = P / 8192; return n < 1 ? 1 : n > 32 ? 32 : n; }
size_t dw_wgrad_partial_floats(int B, int C, int P) { return (size_t)B * dw_wgrad_nblk(P) * C * 10; }

// dw[c][9] tap gradients, db[c] bias gradient (db may be null)
int launch_dw_wgrad(const float* x, const float* dy, float* dw, float* db, float* partial, int B, int C, int h, int w, int accumulate, hipStream_t st) {
    const int nblk = dw_wgrad_nblk(h * w);
    ProfScope prof(st, "dw_wgrad_kernel", 20.0 * B * C * h * w, 8.0 * B * C * h * w);
    if (w % 4 == 0 && aligned16(x) && aligned16(dy) && ((size_t)h * w) % 4 == 0)
        dw_wgrad_kernel<4><<<dim3((unsigned)nblk, (unsigned)C, (unsigned)B), 256, 0, st>>>(x, dy, partial, C, h, w, nblk);
    else
        dw_wgrad_kernel<1><<<dim3((unsigned)nblk, (unsigned)C, (unsigned)B), 256, 0, st>>>(x, dy, partial, C, h, w, nblk);
    reduce_dw_kernel<<<red_grid((size_t)C * 10), 256, 0, st>>>(partial, dw, db, B * nblk, C, accumulate);
    return check_launch("dw_wgrad");
}

int launch_ewise(const float* a, const float* b, float* out, size_t n, int mode, float slope, hipStream_t st) {
    ProfScope prof(st, "ewise_kernel", 0.0, (mode == 3 || mode == 5 ? 8.0 : 12.0) * n);
    if (n % 4 == 0 && aligned16(a) && aligned16(out) && (!b || aligned16(b))) ewise_kernel<4><<<grid1d(n / 4), 256, 0, st>>>(a, b, out, n, mode, slope);
    else ewise_kernel<1><<<grid1d(n), 256, 0, st>>>(a, b, out, n, mode, slope);
    return check_launch("ewise");
}

int launch_split_halves(const float* src, float* a, float* b, int B, int C, int P, hipStream_t st) {
    RF_CHECK_ARG(((size_t)C * P) % 4 == 0 && aligned16(src) && aligned16(a) && aligned16(b), "split_halves: C * P must be a multiple of 4, buffers 16-byte aligned");
    const size_t half4 = (size_t)C * P / 4, total4 = 2 * half4 * B;
    ProfScope prof(st, "split_halves_kernel", 0.0, 32.0 * half4 * B);
    split_halves_kernel<<<grid1d(total4), 256, 0, st>>>(reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(a), reinterpret_cast<float4*>(b), half4, total4);
    return check_launch("split_halves");
}

int launch_flip3x3(const float* w, float* out, int Cout, int Cin, int dense, hipStream_t st) {
    flip_kernel<<<grid1d((size_t)Cout * (dense ? Cin : 1) * 9), 256, 0, st>>>(w, out, Cout, Cin, dense);
    return check_launch("flip3x3");
}

int loss_nblk() { return 1024; }
int launch_loss(const float* pred, const float* gt, float* grad, float* loss_out, float* partial, size_t n, int mode, float eps, hipStream_t st) {
    loss_kernel<<<loss_nblk(), 256, 0, st>>>(pred, gt, grad, partial, n, mode, eps, 1.0f / (float)n);
    reduce_partials_kernel<<<1, 256, 0, st>>>(partial, loss_out, loss_nblk(), 1, 0);
    return check_launch("loss");
}

int launch_adam(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, float wd, int decoupled,
                int step, float gscale, hipStream_t st) {
    const float bc1 = 1.0f - powf(b1, (float)step), bc2 = 1.0f - powf(b2, (float)step);
    ProfScope prof(st, "adam_kernel", 12.0 * n, 28.0 * n);
    adam_kernel<<<grid1d(n), 256, 0, st>>>(p, g, m, v, n, lr, b1, b2, eps, wd, decoupled, bc1, bc2, gscale);
    return check_launch("adam");
}

}  // namespace rf

// =================================================================================================
// FLCA branch, backward (FrequencyawareLumaChromaAttentionRAWFormer.py:103-162).  Forward:
//   S = 1 + alpha sigmoid(conv(g_low; w_low)) + beta tanh(conv(g_high; w_high)) + gamma sigmoid(conv(g_cr, g_cb; w_chr))
//   xs = feat S;   ch = sigmoid(W3 relu(W1 mean_p(xs) + b1) + b3) per image;   z = xs ch
// Backward of z w.r.t. feat and the 10 parameter tensors (the guidance planes carry no gradient: they come from the input):
//   dch[b][c] = sum_p dz xs           -> squeeze-excite MLP backward (tiny, per image) -> dm[b][c] (gradient of the pooled mean)
//   dxs = dz ch + dm / P;   dfeat = dxs S;   dS = dxs feat;   dalpha = sum dS a_low, ...;   d(conv outputs) -> tap sums
// =================================================================================================
namespace rf {
namespace {

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

// dch partial: partial[((b * nblk + blk) * C + c)] = sum over the block's pixels of dz * xs
__global__ void __launch_bounds__(256) flca_dch_kernel(const float* __restrict__ dz, int64_t dz_bstride, const float* __restrict__ xs, float* __restrict__ partial,
                                                       int C, int P, int nblk) {
    const int blk = blockIdx.x, c = blockIdx.y, b = blockIdx.z;
    const float* d = dz + (size_t)b * dz_bstride + (size_t)c * P;
    const float* x = xs + ((size_t)b * C + c) * P;
    const int per = (P + nblk - 1) / nblk, lo = blk * per, hi = (lo + per < P) ? lo + per : P;
    float s = 0.f;
    for (int p = lo + threadIdx.x; p < hi; p += 256) s = fmaf(d[p], x[p], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    __shared__ float ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[((size_t)b * nblk + blk) * C + c] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

// squeeze-excite backward, one workgroup per image (a single workgroup walking the images in turn took 120 us per stage):
// in: pooled sums partial (forward: flca partial / P = mean), dch partials;  out: dm[b][c] / P and the image's CONTRIBUTION to the
// gradients of se.1.{w,b}, se.3.{w,b} in contrib[b][hid C | hid | C hid | C] (the order of the four tensors in the flat gradient
// buffer), summed over the images in order by reduce_partials_kernel: deterministic.
__global__ void __launch_bounds__(256) flca_se_bwd_kernel(const float* __restrict__ pool_partial, int pool_nblk, const float* __restrict__ dch_partial, int dch_nblk,
                                                          const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w3, const float* __restrict__ b3,
                                                          float* __restrict__ dmP, float* __restrict__ contrib, int C, int hid, int P) {
    __shared__ float mean[512], dch[512], ds3[512], hv[64], dh[64];
    __shared__ float part[2][256];
    const size_t b = blockIdx.x;
    // column sums of the two partial tables: cw channels x nsl row slices per pass, 8 loads in flight per thread, slices
    // combined in a fixed order (the forward's flca_se_kernel sums its table the same way)
    const int cw = C < 256 ? C : 256, nsl = 256 / cw;
    for (int c0 = 0; c0 < C; c0 += cw) {
        const int c = c0 + threadIdx.x % cw, sl = threadIdx.x / cw;
        float s = 0.f, d = 0.f;
        if (sl < nsl && c < C) {
            const float* src = pool_partial + b * pool_nblk * C + c;
            int k = sl;
            for (; k + 7 * nsl < pool_nblk; k += 8 * nsl) {
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = src[(size_t)(k + u * nsl) * C];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += t[u];
            }
            for (; k < pool_nblk; k += nsl) s += src[(size_t)k * C];
            const float* dsrc = dch_partial + b * dch_nblk * C + c;
            for (int q = sl; q < dch_nblk; q += nsl) d += dsrc[(size_t)q * C];
        }
        part[0][threadIdx.x] = s;
        part[1][threadIdx.x] = d;
        __syncthreads();
        if (threadIdx.x < cw && c0 + threadIdx.x < C) {
            float t = 0.f, u = 0.f;
            for (int q = 0; q < nsl; ++q) { t += part[0][q * cw + threadIdx.x]; u += part[1][q * cw + threadIdx.x]; }
            mean[c0 + threadIdx.x] = t / (float)P;
            dch[c0 + threadIdx.x] = u;
        }
        __syncthreads();
    }
    for (int m = threadIdx.x; m < hid; m += 256) {
        float s = b1[m];
        for (int c = 0; c < C; ++c) s = fmaf(w1[m * C + c], mean[c], s);
        hv[m] = fmaxf(s, 0.f);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = b3[c];
        for (int m = 0; m < hid; ++m) s = fmaf(w3[c * hid + m], hv[m], s);
        const float chv = sigm(s);
        ds3[c] = dch[c] * chv * (1.0f - chv);
    }
    __syncthreads();
    for (int m = threadIdx.x; m < hid; m += 256) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) s = fmaf(w3[c * hid + m], ds3[c], s);
        dh[m] = hv[m] > 0.f ? s : 0.f;
    }
    __syncthreads();
    float* gw1 = contrib + b * (size_t)(2 * C * hid + hid + C);
    float* gb1 = gw1 + (size_t)hid * C;
    float* gw3 = gb1 + hid;
    float* gb3 = gw3 + (size_t)C * hid;
    for (int e = threadIdx.x; e < hid * C; e += 256) gw1[e] = dh[e / C] * mean[e % C];          // [hid][C]
    for (int m = threadIdx.x; m < hid; m += 256) gb1[m] = dh[m];
    for (int e = threadIdx.x; e < C * hid; e += 256) gw3[e] = ds3[e / hid] * hv[e % hid];       // [C][hid]
    for (int c = threadIdx.x; c < C; c += 256) gb3[c] = ds3[c];
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int m = 0; m < hid; ++m) s = fmaf(w1[m * C + c], dh[m], s);
        dmP[b * C + c] = s / (float)P;
    }
}

// spatial part, element-wise: one thread per pixel, channels in a loop (the guidance neighbourhoods stay in registers).
//   dfeat = dxs S;   ds_low = alpha dS a_l (1 - a_l),  ds_high = beta dS (1 - a_h^2),  ds_chr = gamma dS a_c (1 - a_c)   (dS = dxs feat)
// are written out: the tap sums dW[c][tap] = sum_p ds[c][p] g[p + tap] are then weight gradients of a 3x3 convolution with a
// 1- or 2-plane input = gram2<9>.  dalpha, dbeta, dgamma accumulate per thread over all channels: one block reduction at the end.
struct FlcaBwdArgs {
    const float* feat; const float* guide; const float* dz; int64_t dz_bstride; const float* ch; const float* dmP;
    const float* w_low; const float* w_high; const float* w_chr; const float* alpha; const float* beta; const float* gamma;
    float* dfeat; float* ds;        // ds: [3][B][C][P]
    float* abg_partial;             // [(b * nblk + blk)][3]
    int B, C, h, w, nblk, accumulate;
};

__global__ void __launch_bounds__(256) flca_spatial_bwd_kernel(FlcaBwdArgs a) {
    const int blk = blockIdx.x;
    const size_t b = blockIdx.y;
    const int h = a.h, w = a.w, P = h * w, C = a.C;
    const int p = blk * 256 + threadIdx.x;
    const bool live = p < P;
    const int y = live ? p / w : 0, x = live ? p % w : 0;
    float nb[4][9];
    const float* gb = a.guide + b * 4 * (size_t)P;
#pragma unroll
    for (int pl = 0; pl < 4; ++pl)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            nb[pl][t] = (live && yy >= 0 && yy < h && xx >= 0 && xx < w) ? gb[(size_t)pl * P + (size_t)yy * w + xx] : 0.f;
        }
    const float al = *a.alpha, be = *a.beta, ga = *a.gamma;
    const size_t plane = (size_t)a.B * C * P;
    float sa = 0.f, sb = 0.f, sg = 0.f;
    if (live) {
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
            const float a_l = sigm(sl), a_h = tanhf(sh), a_c = sigm(sc);
            const size_t idx = (b * C + c) * (size_t)P + p;
            const float dxs = a.dz[b * a.dz_bstride + (size_t)c * P + p] * a.ch[b * C + c] + a.dmP[b * C + c];
            const float df = dxs * (1.0f + al * a_l + be * a_h + ga * a_c);
            a.dfeat[idx] = a.accumulate ? a.dfeat[idx] + df : df;
            const float dS = dxs * a.feat[idx];
            a.ds[idx] = al * dS * a_l * (1.0f - a_l);
            a.ds[plane + idx] = be * dS * (1.0f - a_h * a_h);
            a.ds[2 * plane + idx] = ga * dS * a_c * (1.0f - a_c);
            sa = fmaf(dS, a_l, sa); sb = fmaf(dS, a_h, sb); sg = fmaf(dS, a_c, sg);
        }
    }
    __shared__ float red[3][4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sa += __shfl_xor(sa, o); sb += __shfl_xor(sb, o); sg += __shfl_xor(sg, o); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sa; red[1][threadIdx.x >> 6] = sb; red[2][threadIdx.x >> 6] = sg; }
    __syncthreads();
    if (threadIdx.x < 3)
        a.abg_partial[(b * a.nblk + blk) * 3 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// Spatial part and tap sums in ONE kernel (w % 4 == 0): the element-wise kernel above writes the three ds tensors for a tap-sum pass to
// read back (6 of the 9 streams of the pair).  Here a lane owns channel 16 ti + r and the 4 pixels of its pixel group kq -- the
// A-operand layout of the tap contraction below -- recomputes the three gates of its channel at those pixels from the guidance
// neighbourhood (36 weights of the channel in registers), writes dfeat, and feeds ds_low / ds_high / ds_chr straight into the
// MFMAs whose B rows are the TAPS: G[c][col] = sum_p ds_k[c][p] * plane_k[p + tap] is a contraction over pixels; lane (r, kq) of
// the B operand loads tap r of its plane at the lane's 4 pixels, four accumulator tiles per A tile: low x plane 0, high x plane 1,
// chroma x plane 2, chroma x plane 3 (columns 0-8 = the 3 x 3 window).  dz and feat are read once, dfeat written once, nothing
// else touches HBM (the guidance planes are cache resident).
// partial[(slab * C + c) * 36 + {low 0-8, high 9-17, chroma 18-35}], slabs reduced in order by reduce_partials_kernel;
// abg_partial[(slab * gridDim.y + ti) * 3 + k].
__device__ __forceinline__ float fsig(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

__global__ void __launch_bounds__(256) flca_bwd_fused_kernel(FlcaBwdArgs a, float* __restrict__ partial, int slab_px, int slabs_per_image) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const int slab = blockIdx.x, ti = blockIdx.y;
    const int img = slab / slabs_per_image, sl = slab - img * slabs_per_image;
    const int h = a.h, w = a.w, P = h * w, C = a.C;
    const int n_lo = sl * slab_px, n_hi = (n_lo + slab_px < P) ? n_lo + slab_px : P;
    const bool cvalid = 16 * ti + r < C;
    const int c = cvalid ? 16 * ti + r : C - 1;
    float wl[9], wh[9], wc[18];
#pragma unroll
    for (int t = 0; t < 9; ++t) { wl[t] = a.w_low[c * 9 + t]; wh[t] = a.w_high[c * 9 + t]; wc[t] = a.w_chr[c * 18 + t]; wc[9 + t] = a.w_chr[c * 18 + 9 + t]; }
    const float al = *a.alpha, be = *a.beta, ga = *a.gamma;
    const float chv = a.ch[(size_t)img * C + c], dm = a.dmP[(size_t)img * C + c];
    const float* dzr = a.dz + (size_t)img * a.dz_bstride + (size_t)c * P;
    const float* fr = a.feat + ((size_t)img * C + c) * P;
    float* dfr = a.dfeat + ((size_t)img * C + c) * P;
    const float* gpl = a.guide + (size_t)img * 4 * P;
    const int tdy = r < 9 ? r / 3 - 1 : 0, tdx = r < 9 ? r % 3 - 1 : 0;      // this lane's tap (B operand rows 9-15 are zero)
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float sa = 0.f, sb = 0.f, sg = 0.f;
    for (int n0 = n_lo + wave * 16; n0 < n_hi; n0 += 64) {
        const int nn = n0 + 4 * kq;
        const bool ok = nn < n_hi;
        const int n = ok ? nn : n_lo;
        const int y = n / w, x = n - y * w;
        const float4 dz4 = ld4(dzr + n), f4 = ld4(fr + n);
        // B operand: this lane's tap of the four planes at the lane's 4 pixels
        float bv[4][4];
        {
            const int yy = y + tdy;
            const bool rowok = ok && r < 9 && yy >= 0 && yy < h;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int xx = x + m + tdx;
                const bool in = rowok && xx >= 0 && xx < w;
                const size_t off = in ? (size_t)yy * w + xx : 0;
#pragma unroll
                for (int pl = 0; pl < 4; ++pl) { const float v = gpl[(size_t)pl * P + off]; bv[pl][m] = in ? v : 0.f; }
            }
        }
        // gates of the lane's channel at its 4 pixels: pre-activations from the 3 x 6 neighbourhood of each plane
        float s_l[4] = {0.f, 0.f, 0.f, 0.f}, s_h[4] = {0.f, 0.f, 0.f, 0.f}, s_c[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = y + dy - 1;
            const bool rok = yy >= 0 && yy < h;
            const size_t rowoff = (size_t)(rok ? yy : 0) * w + x;
            float nb[4][6];
#pragma unroll
            for (int pl = 0; pl < 4; ++pl) {
                const float* row = gpl + (size_t)pl * P + rowoff;
                const float4 t = ld4(row);
                const float lft = row[x > 0 ? -1 : 0], rgt = row[x + 4 < w ? 4 : 0];
                nb[pl][0] = (rok && x > 0) ? lft : 0.f;
                nb[pl][1] = rok ? t.x : 0.f; nb[pl][2] = rok ? t.y : 0.f; nb[pl][3] = rok ? t.z : 0.f; nb[pl][4] = rok ? t.w : 0.f;
                nb[pl][5] = (rok && x + 4 < w) ? rgt : 0.f;
            }
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const float k0 = wl[dy * 3 + dx], k1 = wh[dy * 3 + dx], k2 = wc[dy * 3 + dx], k3 = wc[9 + dy * 3 + dx];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    s_l[q] = fmaf(k0, nb[0][q + dx], s_l[q]);
                    s_h[q] = fmaf(k1, nb[1][q + dx], s_h[q]);
                    s_c[q] = fmaf(k3, nb[3][q + dx], fmaf(k2, nb[2][q + dx], s_c[q]));
                }
            }
        }
        const float dzv[4] = {dz4.x, dz4.y, dz4.z, dz4.w}, fv[4] = {f4.x, f4.y, f4.z, f4.w};
        const float live = (ok && cvalid) ? 1.0f : 0.f;
        float dsl[4], dsh[4], dsc[4], df[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float a_l = fsig(s_l[q]), a_h = 2.0f * fsig(2.0f * s_h[q]) - 1.0f, a_c = fsig(s_c[q]);
            const float dxs = live * fmaf(dzv[q], chv, dm);
            df[q] = dxs * (1.0f + al * a_l + be * a_h + ga * a_c);
            const float dS = dxs * fv[q];
            dsl[q] = al * dS * a_l * (1.0f - a_l);
            dsh[q] = be * dS * (1.0f - a_h * a_h);
            dsc[q] = ga * dS * a_c * (1.0f - a_c);
            sa = fmaf(dS, a_l, sa); sb = fmaf(dS, a_h, sb); sg = fmaf(dS, a_c, sg);
        }
        if (ok && cvalid) {
            float4 o = make_float4(df[0], df[1], df[2], df[3]);
            if (a.accumulate) { const float4 p = ld4(dfr + n); o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
            *reinterpret_cast<float4*>(dfr + n) = o;
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(dsl[m], bv[0][m], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(dsh[m], bv[1][m], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(dsc[m], bv[2][m], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(dsc[m], bv[3][m], acc[3], 0, 0, 0);
        }
    }
    __shared__ float red[4][16][17];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) red[wave][4 * kq + q][r] = acc[t][q];
        __syncthreads();
        const int i = threadIdx.x >> 4, jj = threadIdx.x & 15;
        const int gi = 16 * ti + i;
        if (gi < C && jj < 9)
            partial[((size_t)slab * C + gi) * 36 + 9 * t + jj] = ((red[0][i][jj] + red[1][i][jj]) + red[2][i][jj]) + red[3][i][jj];
    }
    // dalpha, dbeta, dgamma: wave butterflies, then the four waves in order
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sa += __shfl_xor(sa, o); sb += __shfl_xor(sb, o); sg += __shfl_xor(sg, o); }
    if (lane == 0) { red[0][0][wave] = sa; red[0][1][wave] = sb; red[0][2][wave] = sg; }
    __syncthreads();
    if (threadIdx.x < 3)
        a.abg_partial[((size_t)slab * gridDim.y + ti) * 3 + threadIdx.x] =
            (red[0][threadIdx.x][0] + red[0][threadIdx.x][1]) + (red[0][threadIdx.x][2] + red[0][threadIdx.x][3]);
}

// the 36 tap sums of a channel -> the three weight gradients: low [C][9], high [C][9], chroma [C][2][9]
__global__ void __launch_bounds__(256) flca_taps_scatter_kernel(const float* __restrict__ sums, float* __restrict__ g_low, float* __restrict__ g_high,
                                                                float* __restrict__ g_chr, int C) {
    for (int e = blockIdx.x * 256 + threadIdx.x; e < C * 36; e += gridDim.x * 256) {
        const int c = e / 36, t = e % 36;
        if (t < 9) g_low[c * 9 + t] += sums[e];
        else if (t < 18) g_high[c * 9 + t - 9] += sums[e];
        else g_chr[c * 18 + t - 18] += sums[e];
    }
}

__global__ void __launch_bounds__(256) flca_abg_kernel(const float* __restrict__ partial, int nrec, float* __restrict__ ga, float* __restrict__ gb, float* __restrict__ gg) {
    __shared__ float part[3][256];
    float s[3] = {0.f, 0.f, 0.f};
    for (int k = threadIdx.x; k < nrec; k += 256) {
#pragma unroll
        for (int t = 0; t < 3; ++t) s[t] += partial[(size_t)k * 3 + t];
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) part[t][threadIdx.x] = s[t];
    __syncthreads();
    if (threadIdx.x < 3) {
        float v = 0.f;
        for (int q = 0; q < 256; ++q) v += part[threadIdx.x][q];      // fixed order
        float* o = threadIdx.x == 0 ? ga : threadIdx.x == 1 ? gb : gg;
        *o += v;
    }
}

}  // namespace

size_t flca_bwd_scratch_floats(int B, int C, int h, int w) {
    const int P = h * w;
    const int hid = C / 8 > 8 ? C / 8 : 8;
    return 3 * (size_t)B * C * P + (size_t)B * cdiv(P, 256) * 3 + (size_t)B * chan_sum_nblk(P) * C + 2 * (size_t)B * C + 256 +
           (size_t)B * (2 * C * hid + hid + C) + 64 +
           gram2_partial_floats(B, C, 2, h, w, 9) + gram2_partial_floats(B, C, 16, h, w, 1) * 3 + (size_t)C * 36 + 64;
}

// prm / grd: alpha, beta, gamma, low_attn.0.w, high_attn.0.w, chroma_attn.0.w, se.1.w, se.1.b, se.3.w, se.3.b (parameters / their gradients)
int launch_flca_backward(const float* feat, const float* guide, const float* xs, const float* dz, int64_t dz_bstride, const float* ch,
                         const float* pool_partial, int pool_nblk, const float* const* prm, float* const* grd, float* dfeat, int accumulate,
                         float* scratch, int B, int C, int h, int w, hipStream_t st) {
    RF_CHECK_ARG(C <= 512 && B <= 65535 && w % 4 == 0, "flca backward: C=%d, w=%d unsupported", C, w);
    const int P = h * w, nblk = cdiv(P, 256), dnblk = chan_sum_nblk(P), hid = C / 8 > 8 ? C / 8 : 8;
    const size_t plane = (size_t)B * C * P;
    float* ds = scratch;
    float* abg = ds + 3 * plane;
    float* dch_part = abg + (size_t)B * nblk * 3;
    float* dmP = dch_part + (size_t)B * dnblk * C;
    float* se_contrib = dmP + (size_t)B * C + 64;
    const size_t n_se = (size_t)2 * C * hid + hid + C;
    float* gpart = se_contrib + align_up((size_t)B * n_se, 64);
    // the four squeeze-excite tensors follow each other in the flat gradient buffer (registry order, sizes multiples of 4)
    RF_CHECK_ARG(grd[7] == grd[6] + (size_t)hid * C && grd[8] == grd[7] + hid && grd[9] == grd[8] + (size_t)C * hid,
                 "flca backward: the gradients of se.1.weight, se.1.bias, se.3.weight, se.3.bias must be contiguous");
    // squeeze-excite part: dch -> per-image MLP adjoint -> dm, the four se gradients
    auto se_part = [&]() -> int {
        flca_dch_kernel<<<dim3((unsigned)dnblk, (unsigned)C, (unsigned)B), 256, 0, st>>>(dz, dz_bstride, xs, dch_part, C, P, dnblk);
        flca_se_bwd_kernel<<<B, 256, 0, st>>>(pool_partial, pool_nblk, dch_part, dnblk, prm[6], prm[7], prm[8], prm[9], dmP, se_contrib, C, hid, P);
        reduce_partials_kernel<<<red_grid(n_se), 256, 0, st>>>(se_contrib, grd[6], B, n_se, 1);
        return check_launch("flca_backward (squeeze-excite)");
    };
    if (w % 4 == 0 && aligned16(dz) && dz_bstride % 4 == 0 && aligned16(feat) && aligned16(dfeat) && aligned16(guide)) {
        // spatial part + tap sums in one kernel; abg partials live in the (unused) ds area
        int slab_px, per_image;
        gram2_slabs(B, P, cdiv(C, 16), &slab_px, &per_image);
        const int nslab = B * per_image, nti = cdiv(C, 16);
        float* sums = gpart + (size_t)nslab * C * 36;
        float* abg2 = ds;
        ProfScope prof(st, "flca_backward(fused)", 2.0 * 36 * (double)B * C * P + 200.0 * B * C * P, 12.0 * (double)B * C * P + 8.0 * B * C * P);
        if (int rc = se_part()) return rc;
        FlcaBwdArgs a{feat, guide, dz, dz_bstride, ch, dmP, prm[3], prm[4], prm[5], prm[0], prm[1], prm[2], dfeat, nullptr, abg2, B, C, h, w, nblk, accumulate};
        flca_bwd_fused_kernel<<<dim3((unsigned)nslab, (unsigned)nti), 256, 0, st>>>(a, gpart, slab_px, per_image);
        flca_abg_kernel<<<1, 256, 0, st>>>(abg2, nslab * nti, grd[0], grd[1], grd[2]);
        reduce_partials_kernel<<<red_grid((size_t)C * 36), 256, 0, st>>>(gpart, sums, nslab, (size_t)C * 36, 0);
        flca_taps_scatter_kernel<<<cdiv(C * 36, 256), 256, 0, st>>>(sums, grd[3], grd[4], grd[5], C);
        return check_launch("flca_backward (fused)");
    }
    {
        ProfScope prof(st, "flca_backward(elementwise)", 200.0 * B * C * P, 32.0 * B * C * P);
        if (int rc = se_part()) return rc;
        FlcaBwdArgs a{feat, guide, dz, dz_bstride, ch, dmP, prm[3], prm[4], prm[5], prm[0], prm[1], prm[2], dfeat, ds, abg, B, C, h, w, nblk, accumulate};
        flca_spatial_bwd_kernel<<<dim3((unsigned)nblk, (unsigned)B), 256, 0, st>>>(a);
        flca_abg_kernel<<<1, 256, 0, st>>>(abg, B * nblk, grd[0], grd[1], grd[2]);
        if (int rc = check_launch("flca_backward")) return rc;
    }
    if (int rc = launch_gram2(ds, (int64_t)C * P, C, guide, (int64_t)4 * P, 1, grd[3], 1, gpart, B, h, w, 9, 0, 0, 0, 0, 1, st)) return rc;
    if (int rc = launch_gram2(ds + plane, (int64_t)C * P, C, guide + P, (int64_t)4 * P, 1, grd[4], 1, gpart, B, h, w, 9, 0, 0, 0, 0, 1, st)) return rc;
    return launch_gram2(ds + 2 * plane, (int64_t)C * P, C, guide + 2 * (size_t)P, (int64_t)4 * P, 2, grd[5], 2, gpart, B, h, w, 9, 0, 0, 0, 0, 1, st);
}

}  // namespace rf
