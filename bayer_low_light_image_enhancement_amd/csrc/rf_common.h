// Shared declarations for the gfx950 RawFormer kernels (internal; the public ABI is
// include/rawformer_hip.h).  Everything here is written for CDNA4 only: 64-lane waves,
// v_mfma_f32_16x16x4_f32, 160 KB LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/rawformer_hip.h"

namespace rf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);
int check_launch(const char* what);

#define RF_CHECK_ARG(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            rf::set_error(__VA_ARGS__);    \
            return RF_E_INVALID;           \
        }                                  \
    } while (0)

// Optional per-launch HIP-event bracket (rf_profile_begin / rf_profile_end in the C ABI):
// bench.py uses it to time each kernel class on the stream it is launched on.
struct ProfScope {
    ProfScope(hipStream_t st, const char* key, double flops, double bytes);
    ~ProfScope();
    hipStream_t st_;
    int rec_;
};
bool profiling_active();   // between rf_profile_begin and rf_profile_end: schedules stay on ONE stream so the brackets mean what they say

// GELU(v) = v/2 (1 + erf(v/sqrt 2)) with erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7), arranged for the fewest
// VALU operations (this sits next to MFMAs, and f32 VALU time adds to f32 MFMA time): with h = v/2 and
// q = 1 - erf(|v|/sqrt 2) = poly(t) t exp(-v^2/2), t = 1 / (1 + p |v|/sqrt 2):
//   GELU = h + |h| (1 - q) = h + fma(-|h|, q, |h|)            (no sign transfer: h sign(v) = |h|)
// 11 plain operations + v_rcp_f32 + v_exp_f32.
__device__ __forceinline__ float gelu_fast(float v) {
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, fabsf(v), 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float ex = __builtin_amdgcn_exp2f((v * v) * (-0.5f * 1.44269504088896340736f));   // exp(-v^2 / 2)
    const float q = (p * t) * ex;
    const float h = 0.5f * v;
    return h + fmaf(-fabsf(h), q, fabsf(h));
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---------------------------------------------------------------------------------------------
// MFMA operand packing.  All dense contractions use v_mfma_f32_16x16x4_f32:
//   A (weights)   lane l holds A[i = l & 15][k = l >> 4]
//   B (pixels)    lane l holds B[k = l >> 4][j = l & 15]
//   C/D           lane l, reg r holds D[row = 4 * (l >> 4) + r][col = l & 15]
// A packed weight matrix [Cout][K] is stored as tiles of 64 floats in lane order:
//   packed[(s * NT + t) * 64 + l] = W[16 t + (l & 15)][4 s + (l >> 4)]      (zero outside)
// with NT = ceil(Cout / 16) and s over ceil(K / 4) k-sets.  A 3x3 weight [Cout][Cin][3][3] is
//   packed[((s * TAPS + dy * TAPS/3 + j) * NT + t) * 64 + l] = u_j(W[16 t + (l & 15)][4 s + (l >> 4)][dy][0..2]),
// the Winograd transform G g of the three taps of kernel row dy (rf_pack.hip; TAPS = 18).
// ---------------------------------------------------------------------------------------------
static inline size_t packed1x1_floats(int K, int Cout) { return (size_t)cdiv(K, 4) * cdiv(Cout, 16) * 64; }
// 3x3 weights are packed in Winograd F(4,3) form along x: 18 transformed taps per k-set
static inline size_t packed3x3_floats(int Cin, int Cout) { return (size_t)cdiv(Cin, 8) * 2 * 18 * cdiv(Cout, 16) * 64; }

// ---------------------------------------------------------------------------------------------
// "b3" operands: an f32 value as the exact sum of three bf16 pieces (truncation split: x = p0 + p1 + p2, 8 significant
// bits each), contracted on the bf16 matrix pipe as six cross terms with f32 accumulation
//   a b ~= a2 b0 + a0 b2 + a1 b1 + a1 b0 + a0 b1 + a0 b0        (dropped: a1 b2, a2 b1, a2 b2 <= 2^-24 |a b|)
// Measured on MI355X (tools/ubench/bf16x3.hip, profiles/r02_ubench_*.txt): the error against an f64 reference equals
// the f32 MFMA chain's (7e-8 .. 6e-7 of sum |a b|, K = 32 .. 2304), and v_mfma_f32_16x16x32_bf16 issues every 17 cycles
// for K = 32 where v_mfma_f32_16x16x4_f32 needs 8 x 33: 6 x 17 = 102 cycles per 16x16x32 block instead of 264.
// (The f32 MFMA runs at the f32 VECTOR rate and is mutually exclusive with VALU work on its SIMD -- tools/ubench/
// mfma_valu.hip -- so it buys operand reuse, not throughput.)
//   v_mfma_f32_16x16x32_bf16:  A lane l holds A[i = l & 15][k = 8 (l >> 4) + e], e = 0..7 (16 bytes, e = ushort index)
//                              B lane l holds B[k = 8 (l >> 4) + e][j = l & 15];   C/D as for the f32 instruction
// A packed b3 weight matrix [Cout][K]: per (k-block s32 of 32 channels, output tile t of 16) three 1 KiB pieces
//   packed3[(((s32 * NT + t) * 3 + piece) * 64 + l) * 8 + e] = piece(W[16 t + (l & 15)][32 s32 + 8 (l >> 4) + e])   (ushort)
// ---------------------------------------------------------------------------------------------
static inline size_t packed1x1_b3_floats(int K, int Cout) { return (size_t)cdiv(K, 32) * cdiv(Cout, 16) * 768; }
int pack_1x1_b3(const float* w, void* packed3, int Cout, int K, int64_t row_stride, int64_t col_stride, hipStream_t st);
// device helpers shared by every producer / consumer of b3 operands
#ifdef __HIPCC__
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void b3_split(float x, unsigned& p0, unsigned& p1, unsigned& p2) {   // pieces = HIGH halves of p0, p1, p2
    p0 = __float_as_uint(x);
    const float r1 = x - __uint_as_float(p0 & 0xffff0000u);
    p1 = __float_as_uint(r1);
    p2 = __float_as_uint(r1 - __uint_as_float(p1 & 0xffff0000u));
}
__device__ __forceinline__ unsigned b3_pack(unsigned even_elem, unsigned odd_elem) {   // high halves -> one dword (element e at ushort e)
    return __builtin_amdgcn_perm(odd_elem, even_elem, 0x07060302u);
}
__device__ __forceinline__ size_t b3_index(int NT, int co, int k, int piece) {   // ushort index of W[co][k]'s piece
    const int lane = (co & 15) + 16 * ((k & 31) >> 3);
    return ((((size_t)(k >> 5) * NT + (co >> 4)) * 3 + piece) * 64 + lane) * 8 + (k & 7);
}
__device__ __forceinline__ void b3_store(unsigned short* dst, int NT, int co, int k, float v) {
    unsigned p0, p1, p2;
    b3_split(v, p0, p1, p2);
    dst[b3_index(NT, co, k, 0)] = (unsigned short)(p0 >> 16);
    dst[b3_index(NT, co, k, 1)] = (unsigned short)(p1 >> 16);
    dst[b3_index(NT, co, k, 2)] = (unsigned short)(p2 >> 16);
}
// six cross terms, smallest first
__device__ __forceinline__ f32x4 b3_mfma(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x4 c) {
#define RF_B3(x) __builtin_bit_cast(bf16x8, x)
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(RF_B3(a[2]), RF_B3(b[0]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(RF_B3(a[0]), RF_B3(b[2]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(RF_B3(a[1]), RF_B3(b[1]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(RF_B3(a[1]), RF_B3(b[0]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(RF_B3(a[0]), RF_B3(b[1]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(RF_B3(a[0]), RF_B3(b[0]), c, 0, 0, 0);
#undef RF_B3
    return c;
}
// The same six terms for the four pixel sub-groups of a lane, term-major: an accumulator's six MFMAs form a dependent chain, and
// a dependent v_mfma_f32_16x16x32_bf16 issued straight after its producer waits out the producer's 8 passes (measured: chains
// issued one after the other run at half the 17-cycle issue rate).  Interleaved, three independent MFMAs sit between an
// accumulator's consecutive terms.  Every accumulator still sees its terms in b3_mfma's order: the results are bit-identical.
__device__ __forceinline__ void b3_mfma4(const u32x4 (&a)[3], const u32x4 (&b)[4][3], f32x4 (&c)[4]) {
#define RF_B3(x) __builtin_bit_cast(bf16x8, x)
#define RF_B3_TERM(ia, ib)                                                                                     \
    _Pragma("unroll") for (int g = 0; g < 4; ++g)                                                              \
        c[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(RF_B3(a[ia]), RF_B3(b[g][ib]), c[g], 0, 0, 0);
    RF_B3_TERM(2, 0) RF_B3_TERM(0, 2) RF_B3_TERM(1, 1) RF_B3_TERM(1, 0) RF_B3_TERM(0, 1) RF_B3_TERM(0, 0)
#undef RF_B3_TERM
#undef RF_B3
}
#endif

// ---- weight repacking (rf_pack.hip)
// 1x1: W[Cout][K] row-major -> packed.  `row_stride`/`col_stride` let the source be a
// ConvTranspose2d weight [Cin][Cout][2][2] viewed as rows (o,i,j) x cols k.
int pack_1x1(const float* w, float* packed, int Cout, int K, int64_t row_stride, int64_t col_stride, hipStream_t st);
int pack_3x3(const float* w, float* packed, int Cout, int Cin, hipStream_t st);
int pack_convT(const float* w, float* packed, int Cin, int Cout, hipStream_t st);

// ---- batched weight packing (rf_pack.hip; the training step packs every weight form it needs in a few launches)
struct PackDesc {
    const float* src; float* dst;
    int kind;              // 0: 1x1 generic strides, 1: 3x3 Winograd generic strides (+ tap flip), 2: depthwise taps flipped, 3: 1x1 in b3 form
    int rows, cols;        // of the packed matrix (rows = its output channels)
    int64_t rs, cs;        // floats between rows / columns of src
    int flip;
};
constexpr int kPackBatch = 48;
struct PackBatch { PackDesc d[kPackBatch]; };
size_t pack_desc_floats(const PackDesc& d);
int launch_pack_batch(const PackDesc* d, int n, hipStream_t st);

// ---- decoder step ConvTranspose2d(2C,C,2,2) + cat skip + Conv2d(2C,C,1) on composed weights (rf_upcat.hip)
size_t upcat_packed_floats(int C);
int pack_upcat(const float* up_w, const float* up_b, const float* cr_w, const float* cr_b, float* packed, int C, hipStream_t st);
bool upcat_supported(int C, int h, int w, const void* x, const void* skip, const void* out);
int launch_upcat(const float* x, const float* skip, float* out, const float* packed, int B, int C, int h, int w, hipStream_t st);

// ---- conv1x1 (rf_conv1x1.hip)
struct Conv1x1Args {
    const float* x1;       // first source, channel 0 of image 0
    const float* x2;       // second source or nullptr
    int C1, C2;            // channels taken from each source (K = C1 + C2)
    int64_t x1_bstride;    // floats between images in source 1
    int64_t x2_bstride;
    const float* x3;       // third source (bf16x3 streaming kernel only: the composed Conv_Transformer tail, rf_model.hip) or nullptr
    int C3;                // K = C1 + C2 + C3
    int64_t x3_bstride;
    const float* wp;       // packed weights
    int64_t wp_bstride;    // floats between per-image weight sets (0 = shared)
    const void* wp3;       // the same weights in b3 form (packed1x1_b3_floats) or nullptr: selects the bf16x3 kernels
    int64_t wp3_bstride;   // floats between per-image b3 weight sets (0 = shared)
    const float* bias;     // [Cout] or nullptr
    const float* ln_w;     // LayerNorm prologue over the K channels (nullptr = none)
    const float* ln_b;
    float ln_eps;
    const float* res;      // residual, same layout as out (nullptr = none)
    int64_t res_bstride;
    float* out;
    int64_t out_bstride;
    int Cout;              // rows of W (for mode 1: 4 * output channels)
    int B, P;              // images, pixels per image
    int w;                 // image width (mode 1 only)
    int mode;              // 0: out[co][p]; 1: ConvTranspose2d 2x2 scatter, row co = 4*o + 2*i + j
    int act;               // 0 none, 1 LeakyReLU(0.2), 2 ReLU, 3 LeakyReLU(0.1), 4 clamp to [0, 1e4] (FEB, blocks.py:14-30)
};
int launch_conv1x1(const Conv1x1Args& a, hipStream_t st);
bool conv1x1_ln_single_pass(const Conv1x1Args& a);   // false: cheaper as layernorm2d + the plain GEMM (the prologue would re-read x per output group)

// ---- conv3x3 (rf_conv3x3.hip)
struct Conv3x3Args {
    const float* x;        // [B][Cin][h][w]; with unshuffle_in: mosaic [B][Cin/4][2h][2w]
    int64_t x_bstride;
    const float* wp;       // packed 3x3 weights
    const float* bias;     // [Cout] or nullptr
    float* out;
    int64_t out_bstride;
    int B, Cin, Cout, h, w;
    int act;               // 0 none, 1 LeakyReLU(0.2), 2 ReLU, 3 GELU, 4 tanh (EnhancedBayerProcessor, BayerTORGBColorMultiLvl.py:86-98)
    int store;             // 0 plain, 1 pixel-unshuffle, 2 pixel-shuffle
    int unshuffle_in;      // read the input through the Bayer pack (a1)
    int clamp_in;          // clamp input to [0,1] while loading
    int clamp_out;         // clamp output to [0,1] before storing
    // optional scratch for the input-channel split of small launches (rf_conv3x3.hip, KS): conv3x3_ksplit_floats() floats whose
    // first 4096 are zero before the first launch (the kernels leave them zero); nullptr = never split.  One launch at a time.
    float* ks_scratch = nullptr;
    size_t ks_floats = 0;
};
int launch_conv3x3(const Conv3x3Args& a, hipStream_t st);
size_t conv3x3_ksplit_floats(int B, int Cout, int h, int w);   // 0: launches of this size never split
size_t conv3x3_ksplit_counter_bytes();

// ---- memory-bound ops (rf_pointwise.hip)
int launch_layernorm2d(const float* in, float* out, const float* w, const float* b, float eps,
                       int B, int C, int P, hipStream_t st);
struct DwConvArgs {
    const float* x; int64_t x_bstride;
    float* out; int64_t out_bstride;
    const float* w;        // [C][9]
    const float* bias;     // [C] or nullptr
    int B, C, h, w_;
    int gelu;
    float* out2 = nullptr; // training forward: out keeps the pre-activation, out2 (layout of out) receives GELU(out) with the exact erf
};
int launch_dwconv3x3(const DwConvArgs& a, hipStream_t st);
int launch_pixel_unshuffle2(const float* in, float* out, int B, int C, int h, int w, hipStream_t st);
int launch_pixel_shuffle2(const float* in, float* out, int B, int C, int h, int w, hipStream_t st);
// 2x2 analysis/synthesis with an arbitrary 4x4 matrix.  layout 0: bands on the batch axis
// (dwt_init), 1: band-major channels (CustomDWT), 2: four separate [B,C,h,w] planes (HaarDWT).
// exact_haar selects the reference's summation order for dwt_init / iwt_init (bit-exact).
int launch_dwt2x2(const float* in, float* out, const float k[16], int layout, int exact_haar,
                  int B, int C, int h, int w, int hin, int win, hipStream_t st);
int launch_idwt2x2(const float* in, float* out, const float k[16], int layout, int exact_haar,
                   int B, int C, int h, int w, hipStream_t st);

// ---- channel attention (rf_attn.hip)
struct GramArgs {
    const float* q; const float* k;   // channel 0 of image 0 for q and k
    int64_t bstride;                  // floats between images
    int B, C, heads, P;
    float* partial;                   // [B][nslab][C][bandw + 2]  (see gram_partial_floats)
    int nslab, slab;
    int p_lo = 0, p_hi = 0;           // pixels [p_lo, p_hi) enter the statistics (p_hi = 0: all); see rf_set_shard
};
int gram_plan(int B, int C, int heads, int P, int* nslab, int* slab, size_t* partial_floats);
int launch_gram(const GramArgs& a, hipStream_t st);
// softmax + fold into project_out: wp_out[b] = pack(W_out * blockdiag(attn_b))
int launch_attn_fold(const float* partial, int nslab, const float* temperature, const float* w_out,
                     float* wp_out, void* wp3_out /* b3 form too, or nullptr */, int B, int C, int heads, hipStream_t st,
                     int log_temperature = 0);

// ---- fused transformer-block kernels for C = 32 / 64 (rf_fused.hip)
bool fused_ffn_supported(int C, int hidden, int h, int w);
int launch_ffn_fused(const float* x, float* out, const float* ln_w, const float* ln_b, const void* w1p /* b3 */, const float* b1,
                     const float* wd, const float* bd, const float* w2p, const float* b2, int B, int C, int h, int w, hipStream_t st);
bool fused_attn_supported(int C, int heads, int h, int w);
int fused_attn_plan(int h, int w, int* nslab, size_t* partial_floats, int B, int C);
int launch_attn_front(const float* x, float* v, float* partial, int nslab, const float* ln_w, const float* ln_b,
                      const void* wp /* b3 */, const float* bq, const float* wd, const float* bd, int B, int C, int h, int w, hipStream_t st,
                      int ylo = 0, int yhi = 0 /* rows [ylo, yhi) enter the Gram statistics; yhi = 0: all */);

// qkv [B,3C,h,w] -> depthwise 3x3 -> Gram partials of (q,k) + v, for C = 64 / 128
bool attn_mid_supported(int C, int heads, int h, int w);
int attn_mid_plan(int h, int w, int* nslab, size_t* partial_floats, int B, int C);
int launch_attn_mid(const float* qkv, float* v, float* partial, int nslab, const float* wd, const float* bd,
                    int B, int C, int h, int w, hipStream_t st, int ylo = 0, int yhi = 0);

// ---- TransformerBlock schedule (rf_block.hip)
struct TbParams {
    const float *ln1_w, *ln1_b, *temperature;
    const float *qkv_wp /* packed */, *qkv_b, *qkv_dw_w, *qkv_dw_b, *proj_w /* raw [C][C] */, *proj_b;
    const float *ln2_w, *ln2_b, *pw1_wp /* packed */, *pw1_b, *dw_w, *dw_b, *pw2_wp /* packed */, *pw2_b;
    const void *qkv_wp3, *pw1_wp3, *pw2_wp3;   // b3 forms of the three packed weights (nullptr: f32 kernels only)
    int log_temperature;                       // `temperature` holds log T (TrueColorRawFormer, BayerTORGBColorMultiLvl.py:331,344)
    // spatial shard (rf_set_shard): rows [ylo, yhi) of this level enter the Gram statistics, and `allreduce` sums the
    // slab partials over the ranks before the softmax (nullptr: single device)
    int ylo = 0, yhi = 0;
    void (*allreduce)(void* user, float* buf, size_t n, int op, void* stream) = nullptr;
    void* allreduce_user = nullptr;
    // composed stage tail: on the op-by-op FFN path stop after depthwise 3x3 + GELU -- x + attn(..) stays in the x1 buffer, the
    // hidden tensor in bufB, `out` is not written; the caller's channel_reduce GEMM applies pointwise2 (transformer_ffn_is_fused
    // tells it which path runs)
    bool defer_pw2 = false;
};
bool transformer_ffn_is_fused(const TbParams& p, int C, int hc, int hh, int ww);
struct TbBufOffsets { size_t bufA, bufB, x1, partial, wfold, wfold3; };   // float offsets into one scratch area
size_t transformer_scratch_floats(int B, int C, int heads, int hc, int h, int w, TbBufOffsets* o);
int run_transformer(const TbParams& p, const float* in, float* out, float* ws, const TbBufOffsets& o,
                    int B, int C, int heads, int hc, int hh, int ww, hipStream_t st);

// ---- rfft2 / irfft2 with the polar maps of FEB (rf_fft.hip); cscratch: planes * h * (w/2 + 1) complex values
int launch_rfft2_polar(const float* in, float* mag, float* pha, float2* cscratch, int planes, int h, int w, hipStream_t st);
int launch_polar_irfft2(const float* mag, const float* pha, const float* res, float* out, float2* cscratch, int planes, int h, int w,
                        float lim, hipStream_t st);
int launch_clamp(const float* in, float* out, size_t n, float lo, float hi, hipStream_t st);

// ---- TrueColorRawFormer extras (rf_truecolor.hip)
size_t tc_front_scratch_floats(int B, int H, int W, int levels);
int launch_tc_front(const float* in, int mosaic, const float* wb_gains, const float* color_matrix, const float* wp_c0, const float* b_c0,
                    const float* wp_c2, const float* b_c2, const float* wp_d0, const float* b_d0, const float* wp_d2, const float* b_d2,
                    float* scratch, int B, int H, int W, int levels, hipStream_t st);
int launch_tc_guide_level(const float* scratch, float* guide, int B, int H, int W, int levels, int hf, int wf, hipStream_t st);
int tc_front_outputs(const float* scratch, const float** y, const float** crcb, const float** rgb, int B, int H, int W, int levels);
int tc_nblk(int h, int w);
int launch_tc_spatial(const float* feat, float* xs, const float* guide, const float* w_col, const float* b_col, const float* w_low,
                      const float* b_low, const float* w_high, const float* b_high, int B, int C, int h, int w, hipStream_t st);
int launch_tc_residual(const float* xs, const float* r, float* x2, float* partial, int B, int C, int h, int w, hipStream_t st);
int launch_tc_color_head(float* x, const float* const* prm, int B, size_t P, hipStream_t st);

// ---- training kernels (rf_train.hip)
size_t gram2_partial_floats(int B, int Ca, int Cb, int h, int w, int ntap);
int launch_gram2(const float* a, int64_t a_bstride, int Ca, const float* b, int64_t b_bstride, int Cb, float* out, int ld, float* partial,
                 int B, int h, int w, int ntap, int sy, int sx, int per_image, size_t out_istride, int accumulate, hipStream_t st,
                 float* db = nullptr /* [Ca] (+)= row sums of a over all images and pixels: the bias gradient of the same layer */,
                 const float* b2 = nullptr, int64_t b2_bstride = 0, int Cb2 = 0 /* input = cat(b, b2) along channels, read in place */);
int launch_reduce_rows(const float* partial, float* out, int nrows, size_t n, int accumulate, hipStream_t st);
int chan_sum_nblk(int P);
int launch_chan_sum(const float* x, int64_t bstride, float* out, float* partial, int B, int C, int P, int accumulate, hipStream_t st);
size_t ln_bwd_partial_floats(int B, int C, int P);
bool ln_bwd_fused_shape(int C, int P);   // true: launch_ln_bwd takes a strided residual for this shape
int launch_ln_bwd(const float* x, const float* dy, const float* gamma, float* dx, float* dgb, float* partial,
                  int B, int C, int P, float eps, int accumulate_dx, int accumulate_w, hipStream_t st,
                  const float* res = nullptr, int64_t res_bstride = 0);
size_t dw_wgrad_partial_floats(int B, int C, int P);
int launch_dw_wgrad(const float* x, const float* dy, float* dw, float* db, float* partial, int B, int C, int h, int w, int accumulate, hipStream_t st);
int launch_ewise(const float* a, const float* b, float* out, size_t n, int mode, float slope, hipStream_t st);
int launch_split_halves(const float* src, float* a, float* b, int B, int C, int P, hipStream_t st);
int launch_flip3x3(const float* w, float* out, int Cout, int Cin, int dense, hipStream_t st);
int loss_nblk();
int launch_loss(const float* pred, const float* gt, float* grad, float* loss_out, float* partial, size_t n, int mode, float eps, hipStream_t st);
int launch_adam(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, float wd, int decoupled,
                int step, float gscale, hipStream_t st);

size_t flca_bwd_scratch_floats(int B, int C, int h, int w);
int launch_flca_backward(const float* feat, const float* guide, const float* xs, const float* dz, int64_t dz_bstride, const float* ch,
                         const float* pool_partial, int pool_nblk, const float* const* prm, float* const* grd, float* dfeat, int accumulate,
                         float* scratch, int B, int C, int h, int w, hipStream_t st);

// ---- FLCA (rf_flca.hip)
size_t guidance_scratch_floats(int B, int H, int W);
// packed-or-mosaic input -> base planes in scratch (y, cr, cb at HxW; LL, mag at H/2 x W/2)
int launch_guidance_base(const float* in, int mosaic, int clamp_in, float* scratch, int B, int H, int W, hipStream_t st,
                         void (*allreduce)(void*, float*, size_t, int, void*) = nullptr, void* allreduce_user = nullptr);
int launch_guidance_level(const float* scratch, float* guide, int B, int H, int W, int hf, int wf, hipStream_t st);
struct FlcaSpatialArgs {
    const float* feat; float* xs; const float* guide;   // feat/xs [B][C][P], guide [B][4][h][w]
    const float* w_low; const float* w_high; const float* w_chr;   // [C][1][3][3], [C][1][3][3], [C][2][3][3]
    const float* alpha; const float* beta; const float* gamma;     // device scalars
    float* partial;                                      // [B][nblk][C] per-block channel sums
    int B, C, h, w, nblk;
    int ylo = 0, yhi = 0;                                // rows [ylo, yhi) enter the channel sums (yhi = 0: all); see rf_set_shard
};
int flca_nblk(int h, int w);
int launch_flca_spatial(const FlcaSpatialArgs& a, hipStream_t st);
// squeeze-excite gate only: ch_out[b][c]
int launch_flca_se(const float* partial, int nblk, int P, const float* se1_w, const float* se1_b,
                   const float* se3_w, const float* se3_b, int hidden, float* ch_out, int B, int C, hipStream_t st);
// x[b][c][:] *= ch[b][c]   (operator-level FLCA output; the forward folds the gate into the next 1x1 instead)
int launch_scale_channels(float* x, const float* ch, int B, int C, int P, hipStream_t st);
int launch_scale_channels_to(const float* in, float* out, const float* ch, int B, int C, int P, hipStream_t st);   // out = in * ch (in == out allowed)
// SE + fold into channel_reduce: wp_out[b] = pack([Wa * diag(ch_b) | Wb])
int launch_flca_se_fold(const float* partial, int nblk, int P, const float* se1_w, const float* se1_b,
                        const float* se3_w, const float* se3_b, int hidden, const float* w_cr,
                        float* wp_out, void* wp3_out /* b3 form too, or nullptr */, float* ch_out, int B, int C, hipStream_t st,
                        const float* composed = nullptr /* pack_tail: emit [Wa diag(ch_b) | Wb | Wb W2] in b3 form only */, int hc = 0);
// Composed stage tail (rf_model.hip run_stage): channel_reduce(cat(xs, x1 + W2 g + b2)) as ONE GEMM over [xs ; x1 ; g].
// pack_tail writes Wb W2 ([C][hc]) and the composed bias ([C]) once per parameter load; launch_tail_fold the b3 weights.
size_t tail_composed_floats(int C, int hc);
int pack_tail(const float* w_cr, const float* b_cr, const float* w2, const float* b2, float* composed, int C, int hc, hipStream_t st);
int launch_tail_fold(const float* w_cr, const float* ch /* [B][C] gate or nullptr */, const float* composed, void* wp3_out, int B, int C, int hc,
                     hipStream_t st);

}  // namespace rf
