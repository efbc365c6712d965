/*
 * rawformer_hip.h -- C ABI of the MI355X (gfx950) RawFormer inference path.
 *
 * This is the drop-in boundary.  The reference is pure
 * Python/PyTorch with no FFI of its own; what a maintainer would bind is the
 * call `pred = model(inp)` in test.py:116 (RawFomer_WFB_FFAB/test.py:78), i.e.
 * `RawFormer.forward` (RawFomer_WFB_FFAB/model.py:473-508 ==
 * FrequencyawareLumaChromaAttentionRAWFormer.py:330-370), plus the operators it
 * is made of.  Every entry point below names the reference function it
 * replaces.  INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - plain C types only; every pointer is a DEVICE pointer to float32 unless
 *     the comment says host;
 *   - tensors are NCHW, contiguous;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all
 *     calls are asynchronous on it and allocate nothing;
 *   - return value 0 = ok, negative = error (RF_E_*); rf_last_error() gives the
 *     message of the calling thread's last failure;
 *   - a handle is re-entrant across handles, not thread-safe on one handle.
 */
#ifndef RAWFORMER_HIP_H
#define RAWFORMER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RF_OK 0
#define RF_E_INVALID (-22)   /* bad argument / shape          */
#define RF_E_NOMEM (-12)     /* workspace too small           */
#define RF_E_MISSING (-2)    /* a required parameter not set  */
#define RF_E_DEVICE (-5)     /* HIP runtime error             */

#define RF_VARIANT_FLCA 0    /* FrequencyawareLumaChromaAttentionRAWFormer.py:257-278 branch */
#define RF_VARIANT_PLAIN 1   /* conv branch: RawFomer_WFB_FFAB/model.py:393-412, model.py:94-108 */
#define RF_VARIANT_TRUECOLOR 2 /* TrueColorRawFormer, BayerTORGBColorMultiLvl.py:387-462: learned Bayer front end
                                  (EnhancedBayerProcessor), EnhancedFLCA branch, exp(log_temperature) attention, ReLU before
                                  the PixelShuffle, CameraAwareColorCorrection head */

typedef struct rf_handle rf_handle;

typedef struct rf_config {
    int32_t dim;              /* 32 / 48 / 64 (any multiple of 8)                     */
    int32_t heads[4];         /* per U-Net level, reference default {8,8,8,8}         */
    int32_t inp_channels;     /* 1 (Bayer mosaic)                                     */
    int32_t out_channels;     /* 3                                                    */
    int32_t ffn_expansion;    /* 2                                                    */
    int32_t variant;          /* RF_VARIANT_*                                         */
    int32_t branch_lrelu;     /* plain variant: LeakyReLU on the conv branch (WFB) or not (model.py) */
    int32_t clamp_io;         /* clamp input and output to [0,1]: RawFomer_WFB_FFAB/model.py:475,508 */
    int32_t flca_levels;      /* TRUECOLOR: wavelet pyramid depth of EnhancedFLCA (reference default 2; 0 = 2) */
} rf_config;

const char* rf_last_error(void);
int rf_version(void);

/* Diagnostic: bracket every kernel launch made by this library between the two calls with
 * HIP events on the launch stream; rf_profile_end writes a JSON array of
 * {kernel, launches, ms, flops, bytes} (algorithmic work per kernel class) into `json`.
 * Single-threaded use only; adds event overhead, so never leave it on in a timed region. */
int rf_profile_begin(void);
int rf_profile_end(char* json, size_t len);

/* ---- whole model: RawFormer.__init__/load_state_dict/forward ------------------------------ */
int rf_create(const rf_config* cfg, rf_handle** out);
void rf_destroy(rf_handle* h);
/* Parameter registry: names are the reference's state_dict keys
 * (FrequencyawareLumaChromaAttentionRAWFormer.py:297-328).  Pointers are borrowed. */
int rf_param_count(const rf_handle* h);
int rf_param_info(const rf_handle* h, int index, const char** name, int64_t shape[4], int* ndim);
int rf_set_param(rf_handle* h, const char* name, const float* dev_ptr, const int64_t* shape, int ndim);
/* Repack the registered weights into MFMA operand order inside caller memory.  Call again
 * whenever a parameter tensor changed. */
int rf_packed_bytes(const rf_handle* h, size_t* bytes);
int rf_pack_params(rf_handle* h, void* packed_dev, size_t bytes, void* stream);
/* H, W are the PACKED sizes (mosaic is 2H x 2W); both must be multiples of 8. */
int rf_workspace_bytes(const rf_handle* h, int B, int H, int W, size_t* bytes);
/* in: mosaic [B, inp_channels, 2H, 2W] (packed_input = 0) or packed [B, 4*inp_channels, H, W];
 * out: [B, out_channels, 2H, 2W].  Replaces RawFormer.forward (test.py:116). */
int rf_forward(rf_handle* h, const float* in, float* out, void* workspace, size_t workspace_bytes,
               int B, int H, int W, int packed_input, void* stream);

/* Exact spatial sharding of ONE frame over several devices (the reference has no counterpart: it
 * runs whole frames on one device, test.py:116).  Each rank runs rf_forward on a WINDOW of the packed frame: its interior
 * rows [y_lo, y_hi) (local row indices, multiples of 8) plus enough halo rows for the U-Net's receptive field (77 packed rows from 8-aligned cuts; tiling.HALO_ROWS = 80)
 * and the same window height on every rank.  What is global in RawFormer -- the channel attention's Gram / norm statistics
 * (FrequencyawareLumaChromaAttentionRAWFormer.py:205-221) and FLCA's squeeze-excite pooling (:150-154) -- is accumulated over the
 * interior rows only and reduced over the ranks through `allreduce(user, buf, n, op, stream)` (n floats, in place, ordered on
 * `stream`: RCCL all-reduce; op 0 = sum, 1 = max) before it is used; FLCA's luma normaliser (the frame maximum, :59-62) is
 * the max over the windows; total_rows is the frame's packed height (the pooled mean's denominator).
 * Interior rows of the output then equal the whole-frame forward up to summation order.  y_lo = y_hi = 0 and a null
 * callback switch sharding off.  Variants flca and plain. */
typedef void (*rf_allreduce_fn)(void* user, float* buf, size_t n, int op, void* stream);
int rf_set_shard(rf_handle* h, int y_lo, int y_hi, int total_rows, rf_allreduce_fn allreduce, void* user);

/* One Conv_Transformer stage of the model, exactly as rf_forward schedules it (branch || TransformerBlock -> cat ->
 * 1x1 -> 3x3 -> LeakyReLU; FrequencyawareLumaChromaAttentionRAWFormer.py:257-278, RawFomer_WFB_FFAB/model.py:393-412,
 * model.py:94-108), including the squeeze-excite fold into channel_reduce.  stage = 1..7 (conv_tran<stage>), at U-Net
 * level l = stage-1 (encoder) or 7-stage (decoder): in/out [B, dim*2^l, H>>l, W>>l]; packed [B,4,H,W] feeds the FLCA
 * guidance (may be NULL for the plain variant).  Workspace as for rf_forward(B, H, W). */
int rf_forward_stage(rf_handle* h, int stage, const float* in, const float* packed, float* out, void* workspace,
                     size_t workspace_bytes, int B, int H, int W, void* stream);

/* ---- training step (reference train.py:127-147: forward, loss, backward; optimiser train.py:113) ----
 * Parameters, gradients and the Adam moments live in FLAT float buffers: parameter i of the registry sits at float offset
 * rf_flat_offset(i) (16-byte aligned), rf_flat_param_floats floats in all.  The host points rf_set_param at views of its flat
 * parameter buffer, all-reduces the flat gradient buffer across ranks (RCCL) and calls rf_adam_step on the three buffers.
 * rf_train_step: in [B,1,2H,2W] mosaic, gt [B,3,2H,2W]; writes grads (flat), *loss_out (device float), optionally the prediction.
 * loss_mode 0: L1 (RawFomer_WFB_FFAB/train.py:124); 1: Charbonnier sqrt(d^2 + eps^2) (train.py:16-25).
 * Adjoint schedule for variants PLAIN and FLCA (no clamp_io); packed W must be a multiple of 32. */
int rf_flat_param_floats(const rf_handle* h, size_t* floats);
int rf_flat_offset(const rf_handle* h, int index, size_t* offset);
int rf_train_workspace_bytes(const rf_handle* h, int B, int H, int W, size_t* bytes);
/* Overlap of the gradient all-reduce with the backward pass (the reference reduces inside backward through nn.DataParallel,
 * train.py:108-111).  The registry -- and so the flat buffer -- is in FORWARD order, the backward finishes modules in reverse
 * order: rf_train_step calls `ready(user, offset, count, stream)` each time the gradients of flat floats [offset, offset + count)
 * are final (their last kernel has been enqueued on `stream`), from the end of the buffer towards its start, every float
 * exactly once.  The host starts its collective for that range behind those kernels.  NULL switches the notifications off.
 * rf_grad_range_count / rf_grad_range: the ranges a step will announce, in announcement order (static per model). */
typedef void (*rf_grad_ready_fn)(void* user, size_t offset, size_t count, void* stream);
int rf_set_grad_ready(rf_handle* h, rf_grad_ready_fn ready, void* user);
int rf_grad_range_count(const rf_handle* h, int* count);
int rf_grad_range(const rf_handle* h, int index, size_t* offset, size_t* count);
int rf_train_step(rf_handle* h, const float* in, const float* gt, float* grads, float* loss_out, float* pred_out, void* workspace,
                  size_t workspace_bytes, int B, int H, int W, int loss_mode, float loss_eps, void* stream);
/* torch.optim.Adam / AdamW (decoupled != 0) on flat buffers; grads are multiplied by grad_scale first (1 / world size). */
int rf_adam_step(float* params, const float* grads, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                 float weight_decay, int decoupled, int step, float grad_scale, void* stream);

/* ---- operators (parity-test surface) -----------------------------------------------------------
 * Each entry runs the kernels the forward uses for that operator at that shape.  Where the forward
 * fuses ACROSS operators the fused kernel belongs to the wider entry point: rf_chan_attn (no LayerNorm
 * in its contract) takes the depthwise+Gram kernel of levels 1-2 (attn_mid) but not the level-0 kernel
 * that also contains LayerNorm and the qkv 1x1 (attn_front) -- that one, and the fused FFN, are reached
 * through rf_transformer_block. */
/* downshuffle(var, 2): RawFomer_WFB_FFAB/model.py:287-298.  [B,C,2h,2w] -> [B,4C,h,w] */
int rf_pixel_unshuffle2(const float* in, float* out, int B, int C, int h, int w, void* stream);
/* nn.PixelShuffle(2): RawFomer_WFB_FFAB/model.py:471,507.    [B,4C,h,w] -> [B,C,2h,2w] */
int rf_pixel_shuffle2(const float* in, float* out, int B, int C, int h, int w, void* stream);
/* dwt_init / DWT: RawFomer_WFB_FFAB/blocks.py:102-115.       [B,C,2h,2w] -> [4B,C,h,w] (LL,HL,LH,HH) */
int rf_dwt_haar(const float* in, float* out, int B, int C, int h, int w, void* stream);
/* iwt_init / IWT: RawFomer_WFB_FFAB/blocks.py:119-136.       [4B,C,h,w] -> [B,C,2h,2w] */
int rf_idwt_haar(const float* in, float* out, int B, int C, int h, int w, void* stream);
/* CustomDWT: README.md:92-117.  k16 = HOST pointer to the 4x4 kernel, row-major; norm!=0 halves it.
 * [B,C,2h,2w] -> [B,4C,h,w], sub-band-major channels. */
int rf_dwt_custom(const float* in, float* out, const float* k16, int norm, int B, int C, int h, int w, void* stream);
/* CustomIDWT: README.md:120-144.  [B,4C,h,w] -> [B,C,2h,2w] */
int rf_idwt_custom(const float* in, float* out, const float* k16, int norm, int B, int C, int h, int w, void* stream);
/* HaarDWT: FrequencyawareLumaChromaAttentionRAWFormer.py:39-73 (odd sizes reflect-padded).
 * [B,C,hin,win] -> out[4][B,C,ceil(hin/2),ceil(win/2)] in the order LL, LH, HL, HH. */
int rf_haar_dwt(const float* in, float* out, int B, int C, int hin, int win, void* stream);
/* LayerNorm over channels: FrequencyawareLumaChromaAttentionRAWFormer.py:180-187;
 * bias == NULL selects BiasFree_LayerNorm (RawFomer_WFB_FFAB/model.py:89-103). */
int rf_layernorm2d(const float* in, float* out, const float* weight, const float* bias, float eps,
                   int B, int C, int h, int w, void* stream);
/* nn.Conv2d(Cin, Cout, 1) with raw [Cout,Cin] weights; scratch holds the repacked weights
 * (rf_conv1x1_scratch_bytes).  Optional fused LayerNorm prologue (ln_w != NULL), residual add
 * (res != NULL) and a second input (in2, C2 channels, concatenated after the first). */
int rf_conv1x1_scratch_bytes(int Cin_total, int Cout, size_t* bytes);
int rf_conv1x1(const float* in, const float* in2, float* out, const float* weight, const float* bias,
               const float* ln_w, const float* ln_b, const float* res, void* scratch,
               int B, int C1, int C2, int Cout, int h, int w, void* stream);
/* nn.Conv2d(C, C, 3, padding=1, groups=C) (+bias), optional exact GELU: conv_ffn middle
 * (RawFomer_WFB_FFAB/model.py:326-334). */
int rf_dwconv3x3(const float* in, float* out, const float* weight, const float* bias, int gelu,
                 int B, int C, int h, int w, void* stream);
/* nn.Conv2d(Cin, Cout, 3, padding=1) with raw [Cout,Cin,3,3] weights.
 * act: 0 none, 1 LeakyReLU(0.2), 2 ReLU (LumaCond, Attenblock.py:148-153).  store: 0 plain, 1 pixel-unshuffle (Downsample,
 * RawFomer_WFB_FFAB/model.py:300-307), 2 pixel-shuffle (conv_out + PixelShuffle, :505-507). */
int rf_conv3x3_scratch_bytes(int Cin, int Cout, size_t* bytes);
int rf_conv3x3(const float* in, float* out, const float* weight, const float* bias, void* scratch,
               int act, int store, int B, int Cin, int Cout, int h, int w, void* stream);
/* nn.ConvTranspose2d(Cin, Cout, 2, stride=2) with raw [Cin,Cout,2,2] weights
 * (RawFomer_WFB_FFAB/model.py:461).  [B,Cin,h,w] -> [B,Cout,2h,2w] */
int rf_convT2x2_scratch_bytes(int Cin, int Cout, size_t* bytes);
int rf_convT2x2(const float* in, float* out, const float* weight, const float* bias, void* scratch,
                int B, int Cin, int Cout, int h, int w, void* stream);
/* Attention.forward: FrequencyawareLumaChromaAttentionRAWFormer.py:212-235.
 * Takes the 6 raw parameter tensors; scratch from rf_chan_attn_scratch_bytes. */
int rf_chan_attn_scratch_bytes(int B, int C, int heads, int h, int w, size_t* bytes);
int rf_chan_attn(const float* in, float* out, const float* qkv_w, const float* qkv_b,
                 const float* dw_w, const float* dw_b, const float* temperature,
                 const float* proj_w, const float* proj_b, void* scratch,
                 int B, int C, int heads, int h, int w, void* stream);
/* TransformerBlock.forward: FrequencyawareLumaChromaAttentionRAWFormer.py:238-254 (model.py:81-92).
 * prm = HOST array of 17 device pointers in state_dict order: norm1.{w,b}, attn.temperature,
 * attn.qkv.{w,b}, attn.qkv_dwconv.{w,b}, attn.project_out.{w,b}, norm2.{w,b},
 * ffn.pointwise1.{w,b}, ffn.depthwise.{w,b}, ffn.pointwise2.{w,b}.  Uses the fused gfx950 kernels
 * (rf_fused.hip) where the shape allows, exactly as rf_forward does. */
int rf_transformer_block_scratch_bytes(int B, int C, int heads, int ffn_expansion, int h, int w, size_t* bytes);
int rf_transformer_block(const float* in, float* out, const float* const* prm, void* scratch,
                         int B, int C, int heads, int ffn_expansion, int h, int w, void* stream);
/* FLCA.forward given the guidance planes: FrequencyawareLumaChromaAttentionRAWFormer.py:134-162.
 * guide = [B,4,h,w] from rf_flca_guidance at the feature size; prm = HOST array of 10 device pointers:
 * alpha, beta, gamma, low_attn.0.w, high_attn.0.w, chroma_attn.0.w, se.1.{w,b}, se.3.{w,b}. */
int rf_flca_scratch_bytes(int B, int C, int h, int w, size_t* bytes);
int rf_flca(const float* feat, const float* guide, float* out, const float* const* prm, void* scratch,
            int B, int C, int h, int w, void* stream);
/* BayerLumaChroma + HaarDWT + bilinear resize = the guidance planes of FLCA
 * (FrequencyawareLumaChromaAttentionRAWFormer.py:79-97,138-149).
 * packed [B,4,H,W] -> guide [B,4,hf,wf] = (y_low, y_high, cr, cb); scratch >= rf_guidance_scratch_bytes.
 * On return the scratch area starts with the BayerLumaChroma output itself: y | cr | cb, each [B,H,W] floats. */
int rf_guidance_scratch_bytes(int B, int H, int W, size_t* bytes);
int rf_flca_guidance(const float* packed, float* guide, void* scratch, int B, int H, int W, int hf, int wf, void* stream);

/* ---- FFAB / FEB: the frequency block of the WFB variant (RawFomer_WFB_FFAB/blocks.py:11-92) ------------------------
 * rfft2 / irfft2 are hand-written LDS transforms (power-of-two lines: radix 2; any other length up to 2048: direct DFT);
 * norm='ortho'.  Widths must be even (irfft2 with s=(H,W) inverts rfft2 only then).
 * rf_rfft2_polar: [planes,h,w] -> |F| + 1e-6 and angle(F), each [planes,h,w/2+1]  (FEB, blocks.py:27-29)
 * rf_polar_irfft2: mag, pha [planes,h,w/2+1] -> irfft2(mag e^{i pha}) [planes,h,w]  (blocks.py:32-35) */
int rf_rfft2_polar_scratch_bytes(int planes, int h, int w, size_t* bytes);
int rf_rfft2_polar(const float* in, float* mag, float* pha, void* scratch, int planes, int h, int w, void* stream);
int rf_polar_irfft2(const float* mag, const float* pha, float* out, void* scratch, int planes, int h, int w, void* stream);
/* FEB.forward (blocks.py:23-39).  prm = HOST array of 10 device pointers in state_dict order:
 * fpre.{w,b}, process1.0.{w,b}, process1.2.{w,b}, process2.0.{w,b}, process2.2.{w,b}.  [B,nc,h,w] -> same. */
int rf_feb_scratch_bytes(int B, int nc, int h, int w, size_t* bytes);
int rf_feb(const float* in, float* out, const float* const* prm, void* scratch, int B, int nc, int h, int w, void* stream);
/* FFAB.forward (blocks.py:83-92).  prm = HOST array of the module's 92 tensors in state_dict order
 * (conv0.0.{w,b}, conv0.1.<ProcessBlock: 10 FEB tensors, cat.{w,b}>, conv1..conv3, conv4.0.<PB>, conv4.1.{w,b}, conv5.*, convout.*). */
int rf_ffab_scratch_bytes(int B, int nc, int h, int w, size_t* bytes);
int rf_ffab(const float* in, float* out, const float* const* prm, void* scratch, int B, int nc, int h, int w, void* stream);
/* out = add + clamp(in * scale + shift, lo, hi): inverse_data_transform + residual of WMB (model.py:13-15, 241-243). */
int rf_affine_clamp_add(const float* in, const float* add, float* out, size_t n, float scale, float shift, float lo, float hi, void* stream);

/* Decoder step of RawFormer.forward (RawFomer_WFB_FFAB/model.py:461-468, 494-503) as one kernel on composed weights:
 *   out = Conv2d(2C, C, 1)(cat[ConvTranspose2d(2C, C, 2, stride=2)(x), skip])
 * x [B,2C,h,w], skip and out [B,C,2h,2w]; up_w [2C,C,2,2], up_b [C], cr_w [C,2C,1,1], cr_b [C]; w % 4 == 0. */
int rf_upcat_scratch_bytes(int C, size_t* bytes);
int rf_upcat(const float* x, const float* skip, float* out, const float* up_w, const float* up_b, const float* cr_w, const float* cr_b,
             void* scratch, int B, int C, int h, int w, void* stream);

/* ---- evaluation harness: test.py:117-124 on the device ----------- */
/* (clamp(pred,0,1) * 255).astype(uint8), CHW float32 -> HWC uint8 (truncation): test.py:117-118 */
int rf_to_uint8_hwc(const float* in, unsigned char* out, int B, int C, int h, int w, void* stream);
/* per-image sum of squared differences of two uint8 images, exact uint64: PSNR (test.py:123) =
 * 10 log10(255^2 * n_per_image / sse) */
int rf_u8_sse(const unsigned char* a, const unsigned char* b, unsigned long long* sse, int B, size_t n_per_image, void* stream);
/* per-image, per-channel sums of HWC uint8 images (auto_correct_rb compares channel means, test.py:31-40) */
int rf_u8_channel_sums(const unsigned char* a, unsigned long long* sums, int B, int C, size_t hw, void* stream);

/* ---- SID front-end: correctdataloader.py:58-72 (pack_raw), :86 (x ratio),
 * :103 (min 1) fused into one pass over the uint16 Bayer frame [B, 2h, 2w]:
 *   v = min(clip((raw - black) / (white - black), 0, 1) * ratio, 1), evaluated in double, rounded once.
 * mode 0: the loader's packing [B,4,h,w] = sites (0,0) (0,1) (1,1) (1,0); mode 1: pixel_unshuffle order
 * (0,0) (0,1) (1,0) (1,1) (a1); mode 2: normalised mosaic [B,1,2h,2w] (RawFormer.forward's input).
 * black = min(black_level_per_channel); w % 4 == 0. */
int rf_sid_pack(const unsigned short* raw, float* out, int B, int h, int w, int black, int white, double ratio, int mode, void* stream);

/* ---- luminance-aware token attention: Attenblock.py:161-220 -----------------
 * softmax(q_i . k_j * scale) v_j per (image, head), flash style (the N x N scores are never stored).
 * q, k, v: [B, heads*d, N] with `bstride_qkv` floats between images (so they may be the three thirds of one
 * [B, 3*heads*d, N] tensor); out: [B, heads*d, N].  Einsums of Attenblock.py:212-216.  d <= 32. */
int rf_token_attn(const float* q, const float* k, const float* v, float* out, long long bstride_qkv, long long bstride_out,
                  int B, int heads, int d, int N, float scale, void* stream);
/* BayerLuma (Attenblock.py:79-138): mosaic [B,1,H,W] -> luma [B,1,H,W] in [0,1]: the 3x3 mask convolutions of the
 * pattern (0 rggb, 1 bggr, 2 grbg, 3 gbrg), .299 r + .587 g + .114 b, per-image (x - min) / (max - min + 1e-6). */
int rf_bayer_luma_scratch_bytes(int B, int H, int W, size_t* bytes);
int rf_bayer_luma(const float* mosaic, float* luma, void* scratch, int B, int H, int W, int pattern, void* stream);
/* FiLM of q,k,v and the query luma bias (Attenblock.py:193-210): out = gamma * qkv + beta on each third of
 * qkv [B, 3*inner, h*w]; the q third also gets alpha * (avg_pool3(1 - luma) - mean) when luma [B,1,h,w] != NULL.
 * gamma, beta: [B, inner, h*w] with `gb_bstride` floats between images; alpha: device scalar. */
int rf_luma_film_scratch_bytes(int B, int h, int w, size_t* bytes);
int rf_luma_film(const float* qkv, const float* gamma, const float* beta, long long gb_bstride, const float* luma, const float* alpha,
                 float* out, void* scratch, int B, int inner, int h, int w, void* stream);

/* ---- WFB extras without Mamba: RawFomer_WFB_FFAB/model.py:42-87, 174-200 -------
 * Middle of the gated FeedForward in eval mode: a = dw3x3(x; wa, ba) (the fused rep-conv branch, fuse() at
 * model.py:66-87), g = dw3x3(x; wb, bb) (dwconv), out = gelu(g) * a + gelu(a) * g (model.py:59); one pass. */
int rf_dwgate3x3(const float* in, float* out, const float* wa, const float* ba, const float* wb, const float* bb,
                 int B, int C, int h, int w, void* stream);
/* nn.Conv2d(C, C, 5, padding=2, groups=C) (+bias): Illumination_Estimator.depth_conv (model.py:182-183) */
int rf_dwconv5x5(const float* in, float* out, const float* weight, const float* bias, int B, int C, int h, int w, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RAWFORMER_HIP_H */
