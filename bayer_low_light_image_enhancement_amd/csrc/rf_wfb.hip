// WFB extras reachable without Mamba: RawFomer_WFB_FFAB/model.py:17-87, 174-200.
//
//   FeedForward   x = project_in(x);  x1 = x + rep_conv1(x) + rep_conv2(x);  x2 = dwconv(x)
//                 out = project_out(gelu(x2) * x1 + gelu(x1) * x2) + identity
//                 In eval mode the two Conv2d_BN branches and the identity fold into ONE depthwise 3x3 with
//                 bias (FeedForward.fuse(), model.py:66-87; the host does that fold), so the middle is two
//                 depthwise 3x3 filters over the same input and a gate: dwgate_kernel reads x once and writes
//                 the gated tensor once (8 B / element instead of 5 passes).
//   Illumination_Estimator   conv1([img ; mean_c(img)]) -> depthwise 5x5 -> conv2:  the channel mean folds
//                 into conv1's weights on the host; dwconv5x5_kernel is the only new arithmetic.
// The 1x1 convolutions are rf_gemm1x1.hip.
#include "rf_common.h"

namespace rf {

static constexpr int kBlk = 256;

// exact GELU (erff) here: these operators are off the benchmarked path and the gate multiplies two
// GELUs into O(1) values, so the reference's erf is kept to the last bit of libm
__device__ __forceinline__ float gelu_exact(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

// a thread owns 4 pixels x 4 rows of one channel plane (the dwconv3x3_kernel<4,4> shape)
__global__ void __launch_bounds__(kBlk) dwgate_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                      const float* __restrict__ wa, const float* __restrict__ ba,
                                                      const float* __restrict__ wb, const float* __restrict__ bb,
                                                      int B, int C, int h, int w, int vec) {
    const int wv = (w + 3) / 4, hr = (h + 3) / 4;
    const size_t items = (size_t)B * C * hr * wv;
    for (size_t it = blockIdx.x * (size_t)kBlk + threadIdx.x; it < items; it += (size_t)gridDim.x * kBlk) {
        const int xv = (int)(it % wv);
        const int yr = (int)((it / wv) % hr);
        const size_t pl = it / ((size_t)wv * hr);
        const int c = (int)(pl % C);
        const float* xp = x + pl * (size_t)h * w;
        float* op = out + pl * (size_t)h * w;
        float ka[9], kb[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) { ka[i] = wa[c * 9 + i]; kb[i] = wb[c * 9 + i]; }
        const float bia = ba ? ba[c] : 0.f, bib = bb ? bb[c] : 0.f;
        const int x0 = xv * 4, y0 = yr * 4;
        float a[4][4], g[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int p = 0; p < 4; ++p) { a[r][p] = bia; g[r][p] = bib; }
#pragma unroll
        for (int rr = 0; rr < 6; ++rr) {
            const int y = y0 + rr - 1;
            float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (y >= 0 && y < h) {
                const float* row = xp + (size_t)y * w;
                if (vec) {
                    const float4 t = *reinterpret_cast<const float4*>(row + x0);
                    v[1] = t.x; v[2] = t.y; v[3] = t.z; v[4] = t.w;
                } else {
#pragma unroll
                    for (int p = 0; p < 4; ++p) v[1 + p] = x0 + p < w ? row[x0 + p] : 0.f;
                }
                v[0] = x0 > 0 ? row[x0 - 1] : 0.f;
                v[5] = x0 + 4 < w ? row[x0 + 4] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ky = rr - r;
                if (ky >= 0 && ky < 3) {
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        a[r][p] = fmaf(ka[ky * 3 + 2], v[p + 2], fmaf(ka[ky * 3 + 1], v[p + 1], fmaf(ka[ky * 3], v[p], a[r][p])));
                        g[r][p] = fmaf(kb[ky * 3 + 2], v[p + 2], fmaf(kb[ky * 3 + 1], v[p + 1], fmaf(kb[ky * 3], v[p], g[r][p])));
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int y = y0 + r;
            if (y >= h) continue;
            float o[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) o[p] = gelu_exact(g[r][p]) * a[r][p] + gelu_exact(a[r][p]) * g[r][p];   // model.py:59
            if (vec) {
                *reinterpret_cast<float4*>(op + (size_t)y * w + x0) = make_float4(o[0], o[1], o[2], o[3]);
            } else {
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    if (x0 + p < w) op[(size_t)y * w + x0 + p] = o[p];
            }
        }
    }
}

// depthwise 5x5, padding 2 (model.py:182-183): a thread owns 4 pixels x 2 rows
__global__ void __launch_bounds__(kBlk) dwconv5x5_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                         const float* __restrict__ wgt, const float* __restrict__ bias,
                                                         int B, int C, int h, int w) {
    const int wv = (w + 3) / 4, hr = (h + 1) / 2;
    const size_t items = (size_t)B * C * hr * wv;
    for (size_t it = blockIdx.x * (size_t)kBlk + threadIdx.x; it < items; it += (size_t)gridDim.x * kBlk) {
        const int xv = (int)(it % wv);
        const int yr = (int)((it / wv) % hr);
        const size_t pl = it / ((size_t)wv * hr);
        const int c = (int)(pl % C);
        const float* xp = x + pl * (size_t)h * w;
        float* op = out + pl * (size_t)h * w;
        const float* k = wgt + c * 25;
        const float bi = bias ? bias[c] : 0.f;
        const int x0 = xv * 4, y0 = yr * 2;
        float acc[2][4];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[r][p] = bi;
#pragma unroll
        for (int rr = 0; rr < 6; ++rr) {
            const int y = y0 + rr - 2;
            if (y < 0 || y >= h) continue;
            const float* row = xp + (size_t)y * w;
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int xx = x0 - 2 + i;
                v[i] = (xx >= 0 && xx < w) ? row[xx] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int ky = rr - r;
                if (ky >= 0 && ky < 5) {
#pragma unroll
                    for (int p = 0; p < 4; ++p)
#pragma unroll
                        for (int kx = 0; kx < 5; ++kx) acc[r][p] = fmaf(k[ky * 5 + kx], v[p + kx], acc[r][p]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int y = y0 + r;
            if (y >= h) continue;
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if (x0 + p < w) op[(size_t)y * w + x0 + p] = acc[r][p];
        }
    }
}

}  // namespace rf

using namespace rf;

extern "C" {

int rf_dwgate3x3(const float* in, float* out, const float* wa, const float* ba, const float* wb, const float* bb,
                 int B, int C, int h, int w, void* stream) {
    RF_CHECK_ARG(in && out && wa && wb && B > 0 && C > 0 && h > 0 && w > 0, "dwgate3x3: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int vec = (w % 4 == 0) && aligned16(in) && aligned16(out);
    const size_t items = (size_t)B * C * cdiv(h, 4) * cdiv(w, 4);
    size_t gx = (items + kBlk - 1) / kBlk;
    if (gx > 8192) gx = 8192;
    const double el = (double)B * C * h * w;
    ProfScope prof(st, "dwgate_kernel", 40.0 * el, 8.0 * el);
    dwgate_kernel<<<dim3((unsigned)gx), kBlk, 0, st>>>(in, out, wa, ba, wb, bb, B, C, h, w, vec);
    return check_launch("dwgate3x3");
}

int rf_dwconv5x5(const float* in, float* out, const float* weight, const float* bias, int B, int C, int h, int w, void* stream) {
    RF_CHECK_ARG(in && out && weight && B > 0 && C > 0 && h > 0 && w > 0, "dwconv5x5: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const size_t items = (size_t)B * C * cdiv(h, 2) * cdiv(w, 4);
    size_t gx = (items + kBlk - 1) / kBlk;
    if (gx > 8192) gx = 8192;
    const double el = (double)B * C * h * w;
    ProfScope prof(st, "dwconv5x5_kernel", 50.0 * el, 8.0 * el);
    dwconv5x5_kernel<<<dim3((unsigned)gx), kBlk, 0, st>>>(in, out, weight, bias, B, C, h, w);
    return check_launch("dwconv5x5");
}

}  // extern "C"
