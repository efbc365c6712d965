// FEB / FFAB (RawFomer_WFB_FFAB/blocks.py:11-92): the frequency block of the WFB variant.
//
//   FEB(x) = clamp( irfft2( polar( clamp(MLP1(|F| + 1e-6), 0, 1e4), MLP2(angle F) ) ) + clamp(x) ),  F = rfft2(fpre(clamp x))
//
// with norm='ortho' transforms over the last two axes.  Both transforms are hand-written, line-in-LDS kernels:
//
//   rfft2  = [row pass]    real row of length w  -> half spectrum (wf = w/2 + 1 complex), scaled 1/sqrt(w)
//            [column pass] complex FFT of length h over every one of the wf columns, scaled 1/sqrt(h); its epilogue
//                          turns the spectrum into the two REAL planes the MLPs read: |F| + 1e-6 and atan2(Im, Re)
//   irfft2 = [column pass] its prologue turns (mag, pha) into mag cos(pha) + i mag sin(pha); inverse FFT of length h
//            [row pass]    Hermitian half spectrum -> real row (Im of the DC / Nyquist bins ignored, as pocketfft's c2r
//                          does); epilogue: + clamp(x, -10, 10), clamp(-10, 10)
//
// A line of length n lives in LDS as n complex values.  n = 2^k: radix-2 decimation in time (bit-reversed load, k
// in-place stages, one barrier each).  Any other n (e.g. 712 x 1064 = the LL band of a 1424 x 2128 frame): direct O(n^2)
// DFT out of the same LDS line with an exact twiddle index (j k mod n) -- a correctness path, ~n/log2(n) times the work.
// Twiddles come from sincospif (argument exact in float for n <= 2^23), so the error growth matches a table-driven host
// FFT (pocketfft in the reference: O(1e-7 log n) relative).
//
// The four bins of a real 2-D transform that are real by symmetry -- (0|h/2, 0|w/2) -- get Im = +0 exactly, as the
// reference produces them: angle() of a negative real must be +pi, not -pi (it feeds a 1x1 MLP, i.e. is NOT 2 pi
// periodic downstream).  Elsewhere a phase within rounding of the +-pi cut is ill-conditioned in the reference too;
// tests state their tolerance accordingly.
#include "rf_common.h"

namespace rf {

namespace {

constexpr int kMaxLine = 4096;          // longest line held in LDS (32 KB as float2)

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

__device__ __forceinline__ float2 twiddle(int k, int n, bool inverse) {   // exp(-+ 2 pi i k / n)
    float s, c;
    sincospif(2.0f * (float)k / (float)n, &s, &c);
    return make_float2(c, inverse ? s : -s);
}

// In-place transform of L independent lines of length n held in `buf` (line l at buf + l * n); `tmp` (same size) is
// only used by the generic path.  All 256 threads take part; ends with a barrier.  Unscaled.
__device__ void lds_fft_lines(float2* __restrict__ buf, float2* __restrict__ tmp, int n, int log2n, int L, bool inverse) {
    const int tid = threadIdx.x;
    if (log2n >= 0) {
        // bit reversal (in place: swap pairs once)
        for (int i = tid; i < L * n; i += 256) {
            const int l = i / n, k = i - l * n;
            const int r = (int)(__brev((unsigned)k) >> (32 - log2n));
            if (log2n > 0 && r > k) {
                float2* b = buf + l * n;
                const float2 t = b[k];
                b[k] = b[r];
                b[r] = t;
            }
        }
        __syncthreads();
        for (int s = 0; s < log2n; ++s) {
            const int half = 1 << s;
            for (int i = tid; i < L * (n >> 1); i += 256) {
                const int l = i / (n >> 1), q = i - l * (n >> 1);
                const int j = q & (half - 1), base = ((q >> s) << (s + 1)) + j;
                float2* b = buf + l * n;
                const float2 w = twiddle(j << (log2n - 1 - s), n, inverse);   // exp(-+2 pi i j / (2 half))
                const float2 u = b[base], v = cmul(b[base + half], w);
                b[base] = make_float2(u.x + v.x, u.y + v.y);
                b[base + half] = make_float2(u.x - v.x, u.y - v.y);
            }
            __syncthreads();
        }
    } else {
        for (int i = tid; i < L * n; i += 256) {
            const int l = i / n, k = i - l * n;
            const float2* b = buf + l * n;
            float2 acc = make_float2(0.f, 0.f);
            int jk = 0;                                   // j * k mod n, exact
            for (int j = 0; j < n; ++j) {
                const float2 w = twiddle(jk, n, inverse);
                const float2 v = b[j];
                acc.x = fmaf(v.x, w.x, fmaf(-v.y, w.y, acc.x));
                acc.y = fmaf(v.x, w.y, fmaf(v.y, w.x, acc.y));
                jk += k;
                if (jk >= n) jk -= n;
            }
            tmp[i] = acc;
        }
        __syncthreads();
        for (int i = tid; i < L * n; i += 256) buf[i] = tmp[i];
        __syncthreads();
    }
}

static int ilog2_exact(int n) {
    int l = 0;
    while ((1 << l) < n) ++l;
    return (1 << l) == n ? l : -1;
}

// ---- forward row pass: real rows [rows][w] -> half spectrum [rows][wf] complex, scaled by `scale`
__global__ void __launch_bounds__(256) fft_rows_r2c_kernel(const float* __restrict__ in, float2* __restrict__ out, int rows, int w,
                                                           int log2w, int L, float scale) {
    extern __shared__ __attribute__((aligned(16))) float2 fft_lds[];
    float2* buf = fft_lds;
    float2* tmp = fft_lds + L * w;
    const int wf = w / 2 + 1;
    for (int r0 = blockIdx.x * L; r0 < rows; r0 += gridDim.x * L) {
        const int nl = rows - r0 < L ? rows - r0 : L;
        for (int i = threadIdx.x; i < L * w; i += 256) {
            const int l = i / w, k = i - l * w;
            buf[i] = make_float2(l < nl ? in[(size_t)(r0 + l) * w + k] : 0.f, 0.f);
        }
        __syncthreads();
        lds_fft_lines(buf, tmp, w, log2w, L, false);
        for (int i = threadIdx.x; i < nl * wf; i += 256) {
            const int l = i / wf, k = i - l * wf;
            float2 v = buf[l * w + k];
            v.x *= scale; v.y *= scale;
            if (k == 0 || 2 * k == w) v.y = 0.f;          // real by symmetry
            out[(size_t)(r0 + l) * wf + k] = v;
        }
        __syncthreads();
    }
}

// ---- column passes.  One workgroup owns TC consecutive columns of one [h][wf] plane: line c is column c0 + c.
// forward: complex in -> FFT -> (|F| + 1e-6, angle F) planes.   inverse: (mag, pha) planes -> polar -> inverse FFT -> complex out
template <bool INVERSE>
__global__ void __launch_bounds__(256) fft_cols_kernel(const float2* __restrict__ cin, float2* __restrict__ cout_, const float* __restrict__ mag_in,
                                                       const float* __restrict__ pha_in, float* __restrict__ mag_out, float* __restrict__ pha_out,
                                                       int planes, int h, int wf, int w_even_nyq, int log2h, int TC, float scale) {
    extern __shared__ __attribute__((aligned(16))) float2 fft_lds[];
    float2* buf = fft_lds;                    // [TC][h]
    float2* tmp = fft_lds + TC * h;
    const int ctiles = (wf + TC - 1) / TC;
    for (int unit = blockIdx.x; unit < planes * ctiles; unit += gridDim.x) {
        const int pl = unit / ctiles, c0 = (unit - pl * ctiles) * TC;
        const size_t pbase = (size_t)pl * h * wf;
        for (int i = threadIdx.x; i < TC * h; i += 256) {
            const int y = i / TC, c = i - y * TC;          // consecutive threads read consecutive columns of a row
            float2 v = make_float2(0.f, 0.f);
            if (c0 + c < wf) {
                const size_t g = pbase + (size_t)y * wf + c0 + c;
                if constexpr (INVERSE) {
                    const float m = mag_in[g], p = pha_in[g];
                    float s, co;
                    sincosf(p, &s, &co);
                    v = make_float2(m * co, m * s);
                } else {
                    v = cin[g];
                }
            }
            buf[c * h + y] = v;
        }
        __syncthreads();
        lds_fft_lines(buf, tmp, h, log2h, TC, INVERSE);
        for (int i = threadIdx.x; i < TC * h; i += 256) {
            const int y = i / TC, c = i - y * TC;
            if (c0 + c >= wf) continue;
            float2 v = buf[c * h + y];
            v.x *= scale; v.y *= scale;
            const size_t g = pbase + (size_t)y * wf + c0 + c;
            if constexpr (INVERSE) {
                cout_[g] = v;
            } else {
                const int k2 = c0 + c;
                if ((k2 == 0 || k2 == w_even_nyq) && (y == 0 || 2 * y == h)) v.y = 0.f;     // the four real bins: Im = +0
                mag_out[g] = sqrtf(v.x * v.x + v.y * v.y) + 1e-6f;
                pha_out[g] = atan2f(v.y, v.x);
            }
        }
        __syncthreads();
    }
}

// ---- inverse row pass: half spectrum [rows][wf] -> real rows [rows][w]; out = clamp(scale * x + clamp(res))
__global__ void __launch_bounds__(256) fft_rows_c2r_kernel(const float2* __restrict__ in, const float* __restrict__ res, float* __restrict__ out,
                                                           int rows, int w, int log2w, int L, float scale, float lim) {
    extern __shared__ __attribute__((aligned(16))) float2 fft_lds[];
    float2* buf = fft_lds;
    float2* tmp = fft_lds + L * w;
    const int wf = w / 2 + 1;
    for (int r0 = blockIdx.x * L; r0 < rows; r0 += gridDim.x * L) {
        const int nl = rows - r0 < L ? rows - r0 : L;
        for (int i = threadIdx.x; i < L * w; i += 256) {
            const int l = i / w, k = i - l * w;
            float2 v = make_float2(0.f, 0.f);
            if (l < nl) {
                if (k < wf) {
                    v = in[(size_t)(r0 + l) * wf + k];
                    if (k == 0 || 2 * k == w) v.y = 0.f;                   // c2r ignores Im of DC / Nyquist
                } else {
                    v = in[(size_t)(r0 + l) * wf + (w - k)];
                    v.y = -v.y;                                            // Hermitian extension
                }
            }
            buf[i] = v;
        }
        __syncthreads();
        lds_fft_lines(buf, tmp, w, log2w, L, true);
        for (int i = threadIdx.x; i < nl * w; i += 256) {
            const int l = i / w, k = i - l * w;
            const size_t g = (size_t)(r0 + l) * w + k;
            float v = buf[l * w + k].x * scale;
            if (res) v += fminf(fmaxf(res[g], -lim), lim);
            out[g] = fminf(fmaxf(v, -lim), lim);
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) clamp_kernel(const float* __restrict__ in, float* __restrict__ out, size_t n, float lo, float hi) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = fminf(fmaxf(in[i], lo), hi);
}

}  // namespace

static int fft_geometry(int n, const char* what, int* log2n) {
    RF_CHECK_ARG(n >= 2 && n <= kMaxLine, "%s: transform length %d outside [2, %d]", what, n, kMaxLine);
    *log2n = ilog2_exact(n);
    RF_CHECK_ARG(*log2n >= 0 || n <= 2048, "%s: non-power-of-two length %d > 2048 not supported", what, n);
    return RF_OK;
}

// rfft2 (ortho) of [planes][h][w] followed by |.| + 1e-6 and angle: mag / pha are [planes][h][w/2 + 1]; `cscratch` holds
// planes * h * (w/2 + 1) complex values
int launch_rfft2_polar(const float* in, float* mag, float* pha, float2* cscratch, int planes, int h, int w, hipStream_t st) {
    RF_CHECK_ARG(w % 2 == 0, "rfft2: width %d must be even (irfft2 with s=(H,W) needs it to invert)", w);
    int l2w, l2h;
    if (int rc = fft_geometry(w, "rfft2", &l2w)) return rc;
    if (int rc = fft_geometry(h, "rfft2", &l2h)) return rc;
    const int wf = w / 2 + 1, rows = planes * h;
    int L = 2048 / w; if (L < 1) L = 1;
    int gx = cdiv(rows, L); if (gx > 4096) gx = 4096;
    ProfScope prof(st, "rfft2_polar(2 kernels)", 0.0, 4.0 * planes * ((double)h * w + 6.0 * h * wf));
    fft_rows_r2c_kernel<<<gx, 256, (size_t)2 * L * w * sizeof(float2), st>>>(in, cscratch, rows, w, l2w, L, 1.0f / sqrtf((float)w));
    int TC = 2048 / h; if (TC < 1) TC = 1; if (TC > 16) TC = 16;
    int gy = planes * cdiv(wf, TC); if (gy > 4096) gy = 4096;
    fft_cols_kernel<false><<<gy, 256, (size_t)2 * TC * h * sizeof(float2), st>>>(cscratch, nullptr, nullptr, nullptr, mag, pha, planes, h, wf,
                                                                                  w / 2, l2h, TC, 1.0f / sqrtf((float)h));
    return check_launch("rfft2_polar");
}

// irfft2 (ortho, s = (h, w)) of mag * exp(i pha), then out = clamp(. + clamp(res, +-lim), +-lim)   (res may be null)
int launch_polar_irfft2(const float* mag, const float* pha, const float* res, float* out, float2* cscratch, int planes, int h, int w,
                        float lim, hipStream_t st) {
    RF_CHECK_ARG(w % 2 == 0, "irfft2: width %d must be even", w);
    int l2w, l2h;
    if (int rc = fft_geometry(w, "irfft2", &l2w)) return rc;
    if (int rc = fft_geometry(h, "irfft2", &l2h)) return rc;
    const int wf = w / 2 + 1, rows = planes * h;
    ProfScope prof(st, "polar_irfft2(2 kernels)", 0.0, 4.0 * planes * (2.0 * h * w + 6.0 * h * wf));
    int TC = 2048 / h; if (TC < 1) TC = 1; if (TC > 16) TC = 16;
    int gy = planes * cdiv(wf, TC); if (gy > 4096) gy = 4096;
    fft_cols_kernel<true><<<gy, 256, (size_t)2 * TC * h * sizeof(float2), st>>>(nullptr, cscratch, mag, pha, nullptr, nullptr, planes, h, wf,
                                                                                 w / 2, l2h, TC, 1.0f / sqrtf((float)h));
    int L = 2048 / w; if (L < 1) L = 1;
    int gx = cdiv(rows, L); if (gx > 4096) gx = 4096;
    fft_rows_c2r_kernel<<<gx, 256, (size_t)2 * L * w * sizeof(float2), st>>>(cscratch, res, out, rows, w, l2w, L, 1.0f / sqrtf((float)w), lim);
    return check_launch("polar_irfft2");
}

int launch_clamp(const float* in, float* out, size_t n, float lo, float hi, hipStream_t st) {
    int gx = (int)((n + 255) / 256); if (gx > 4096) gx = 4096;
    clamp_kernel<<<gx, 256, 0, st>>>(in, out, n, lo, hi);
    return check_launch("clamp");
}

}  // namespace rf
