/* Plain-C restatement of the byte-moving / wavelet operators of the RawFormer path.
 * TEST INFRASTRUCTURE (oracle), not product: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load the library built from this file (oracle/Makefile -> oracle/_build/).
 * Parity: pinned against the reference's outputs in tests/golden/per_op.npz
 * (tests/test_oracle_c.py).  Citations are relative to the reference tree.
 * All tensors float32, NCHW, contiguous.
 */
#include <stddef.h>

/* downshuffle(var, 2): RawFomer_WFB_FFAB/model.py:287-298
 * out[b][4c + 2i + j][y][x] = in[b][c][2y + i][2x + j] */
void ref_pixel_unshuffle2(const float* in, float* out, int B, int C, int h, int w) {
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j)
                    for (int y = 0; y < h; ++y)
                        for (int x = 0; x < w; ++x)
                            out[(((size_t)b * 4 * C + 4 * c + 2 * i + j) * h + y) * w + x] =
                                in[(((size_t)b * C + c) * 2 * h + 2 * y + i) * 2 * w + 2 * x + j];
}

/* nn.PixelShuffle(2): RawFomer_WFB_FFAB/model.py:471,507
 * out[b][c][2y + i][2x + j] = in[b][4c + 2i + j][y][x] */
void ref_pixel_shuffle2(const float* in, float* out, int B, int C, int h, int w) {
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j)
                    for (int y = 0; y < h; ++y)
                        for (int x = 0; x < w; ++x)
                            out[(((size_t)b * C + c) * 2 * h + 2 * y + i) * 2 * w + 2 * x + j] =
                                in[(((size_t)b * 4 * C + 4 * c + 2 * i + j) * h + y) * w + x];
}

/* dwt_init: RawFomer_WFB_FFAB/blocks.py:102-115.  [B,C,2h,2w] -> [4B,C,h,w], bands LL,HL,LH,HH on
 * the batch axis; the sums are evaluated in the order the Python expression evaluates them. */
void ref_dwt_init(const float* in, float* out, int B, int C, int h, int w) {
    const size_t band = (size_t)B * C * h * w;
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int y = 0; y < h; ++y)
                for (int x = 0; x < w; ++x) {
                    const float* p = in + (((size_t)b * C + c) * 2 * h + 2 * y) * 2 * w + 2 * x;
                    const float x1 = p[0] / 2, x2 = p[2 * w] / 2, x3 = p[1] / 2, x4 = p[2 * w + 1] / 2;
                    const size_t o = (((size_t)b * C + c) * h + y) * w + x;
                    out[o] = ((x1 + x2) + x3) + x4;
                    out[band + o] = ((-x1 - x2) + x3) + x4;
                    out[2 * band + o] = ((-x1 + x2) - x3) + x4;
                    out[3 * band + o] = ((x1 - x2) - x3) + x4;
                }
}

/* iwt_init: RawFomer_WFB_FFAB/blocks.py:119-136.  [4B,C,h,w] -> [B,C,2h,2w] */
void ref_iwt_init(const float* in, float* out, int B, int C, int h, int w) {
    const size_t band = (size_t)B * C * h * w;
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int y = 0; y < h; ++y)
                for (int x = 0; x < w; ++x) {
                    const size_t i = (((size_t)b * C + c) * h + y) * w + x;
                    const float x1 = in[i] / 2, x2 = in[band + i] / 2, x3 = in[2 * band + i] / 2, x4 = in[3 * band + i] / 2;
                    float* p = out + (((size_t)b * C + c) * 2 * h + 2 * y) * 2 * w + 2 * x;
                    p[0] = ((x1 - x2) - x3) + x4;
                    p[2 * w] = ((x1 - x2) + x3) - x4;
                    p[1] = ((x1 + x2) - x3) - x4;
                    p[2 * w + 1] = ((x1 + x2) + x3) + x4;
                }
}

/* CustomDWT: README.md:92-117.  k = 4x4 row-major (already halved when norm=True).
 * out[b][s*C + c][y][x] = sum_t k[s][t] * in[b][c][2y + (t >> 1)][2x + (t & 1)] */
void ref_custom_dwt(const float* in, float* out, const float* k, int B, int C, int h, int w) {
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int y = 0; y < h; ++y)
                for (int x = 0; x < w; ++x) {
                    const float* p = in + (((size_t)b * C + c) * 2 * h + 2 * y) * 2 * w + 2 * x;
                    const float t[4] = {p[0], p[1], p[2 * w], p[2 * w + 1]};
                    for (int s = 0; s < 4; ++s) {
                        double acc = 0.0;
                        for (int q = 0; q < 4; ++q) acc += (double)k[4 * s + q] * t[q];
                        out[(((size_t)b * 4 * C + s * C + c) * h + y) * w + x] = (float)acc;
                    }
                }
}

/* CustomIDWT: README.md:120-144.
 * out[b][c][2y + (t >> 1)][2x + (t & 1)] = sum_s k[s][t] * in[b][s*C + c][y][x] */
void ref_custom_idwt(const float* in, float* out, const float* k, int B, int C, int h, int w) {
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int y = 0; y < h; ++y)
                for (int x = 0; x < w; ++x) {
                    float v[4];
                    for (int s = 0; s < 4; ++s) v[s] = in[(((size_t)b * 4 * C + s * C + c) * h + y) * w + x];
                    float* p = out + (((size_t)b * C + c) * 2 * h + 2 * y) * 2 * w + 2 * x;
                    for (int t = 0; t < 4; ++t) {
                        double acc = 0.0;
                        for (int s = 0; s < 4; ++s) acc += (double)k[4 * s + t] * v[s];
                        p[(t >> 1) * 2 * w + (t & 1)] = (float)acc;
                    }
                }
}

/* HaarDWT: FrequencyawareLumaChromaAttentionRAWFormer.py:39-73.  Orthonormal Haar, reflect pad on
 * the right / bottom when the size is odd.  out = 4 planes [B,C,ceil(hin/2),ceil(win/2)]: LL,LH,HL,HH */
void ref_haar_dwt(const float* in, float* out, int B, int C, int hin, int win) {
    const int h = (hin + 1) / 2, w = (win + 1) / 2;
    const size_t band = (size_t)B * C * h * w;
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int y = 0; y < h; ++y)
                for (int x = 0; x < w; ++x) {
                    const float* pl = in + ((size_t)b * C + c) * hin * win;
                    const int y0 = 2 * y, y1 = 2 * y + 1 < hin ? 2 * y + 1 : hin - 2;
                    const int x0 = 2 * x, x1 = 2 * x + 1 < win ? 2 * x + 1 : win - 2;
                    const float a = pl[(size_t)y0 * win + x0], bb = pl[(size_t)y0 * win + x1];
                    const float cc = pl[(size_t)y1 * win + x0], d = pl[(size_t)y1 * win + x1];
                    const size_t o = (((size_t)b * C + c) * h + y) * w + x;
                    out[o] = 0.5f * (a + bb + cc + d);
                    out[band + o] = 0.5f * (a - bb + cc - d);
                    out[2 * band + o] = 0.5f * (a + bb - cc - d);
                    out[3 * band + o] = 0.5f * (a - bb - cc + d);
                }
}
