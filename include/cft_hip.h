/*
 * cft_hip.h - C ABI of libcft_hip.so: the MI355X (gfx950) kernels behind the two-stream
 * YOLOv5 + CFT inference forward.
 *
 * The reference (DocF/multispectral-object-detection) has no native code and no FFI: its hot
 * path is Python nn.Modules dispatching to ATen.  Each entry point below therefore replaces
 * the ATen call sequence of one reference module forward (file:line given per function,
 * relative to /root/reference).  The Python modules in multispectral-object-detection_amd/
 * models/common.py keep the reference's class names, constructor signatures and state-dict
 * keys and call these functions through ctypes.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller; nothing is allocated or freed here;
 *  - activations are NHWC ("channels-last"): element (b,y,x,c) of a tensor with `ld` channels
 *    per pixel lives at ((b*H + y)*W + x)*ld + off + c, so a tensor may be a channel slice
 *    [off, off+C) of a wider buffer (this is how Concat / C3 / SPP avoid copies);
 *  - dtype codes: CFT_BF16 (bfloat16), CFT_F16 (IEEE half - the precision the reference's GPU callers use,
 *    test.py:66-68 `model.half()`) or CFT_F32; `dtype` is the compute/activation type; all three accumulate in fp32;
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default);
 *  - return value: CFT_OK (0) or a negative CFT_E* code; nothing is launched on error.
 */
#ifndef CFT_HIP_H
#define CFT_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

enum { CFT_BF16 = 0, CFT_F32 = 1, CFT_F16 = 2 };
enum { CFT_ACT_NONE = 0, CFT_ACT_SILU = 1, CFT_ACT_GELU = 2 };
enum {
  CFT_OK = 0,
  CFT_EINVAL = -1,   /* bad argument (alignment, size, dtype) */
  CFT_ELAUNCH = -2,  /* HIP launch error */
  CFT_ENODEV = -3    /* no gfx950 device / wrong architecture */
};

/* Library / device probe.  cft_abi_version() never touches the GPU. */
int cft_abi_version(void);
int cft_device_check(void);               /* CFT_OK iff the current device is gfx950 */
const char* cft_last_error(void);
/* Measurement only (bench.py's sustained leg; no reference counterpart): a one-wave kernel that spins for spin_us microseconds of the
 * constant-rate wall clock and writes out[0] = shader-clock ticks (s_memtime), out[1] = wall-clock ticks (s_memrealtime),
 * out[2] = dependent v_fma_f32 executed meanwhile, out[3] = scratch; *wall_khz = rate of the wall clock.  out: 4 x uint64 in device memory. */
int cft_clock_probe(void* out4_u64, int spin_us, int* wall_khz, void* stream);

/*
 * Convolution as implicit GEMM with fused epilogue:
 *   y[m, yoff+n] = act( sum_{kh,kw,ci} x[b, ho*s+kh-p, wo*s+kw-p, xoff+ci] * w[n][kh][kw][ci] + bias[n] )
 *                  (+ res[m, roff+n] if res != NULL),   m = (b*Ho + ho)*Wo + wo,  p = k/2
 * Replaces Conv.forward / Conv.fuseforward (models/common.py:45-50: conv2d + BatchNorm(eval,
 * folded into w/bias, utils/torch_utils.py:181-201) + SiLU), the Bottleneck residual add
 * (models/common.py:108-109), the channel concat of C3/SPP (via ldy/yoff; :142-143,:163-165),
 * nn.Linear(+bias)(+GELU)(+residual) of the CFT block (models/common.py:450-453,532-546; call
 * with B=1,H=1,W=rows,k=1) and the Detect 1x1 conv (models/yolo_test.py:46).
 *   x    : dtype, NHWC, ldx channels per pixel, slice offset xoff, cin channels used
 *   w    : dtype, [n][kpad] with (kh,kw,ci) flattened, ci fastest, zero padded to kpad
 *          (kpad % 64 == 0 for bf16, % 32 == 0 for f32; cin % 8 (bf16) / % 4 (f32) == 0)
 *   bias : float[n] or NULL
 *   res  : residual, res_dtype, ldr/roff; NULL for none.  May alias y (same element).
 *   y    : out_dtype, ldy/yoff.  n % 8 == 0, ldy % 8 == 0, yoff % 8 == 0.
 */
int cft_conv2d(const void* x, const void* w, const float* bias, const void* res, void* y,
               int B, int H, int W, int cin, int ldx, int xoff,
               int n, int kpad, int ksize, int stride,
               int ldy, int yoff, int ldr, int roff,
               int act, int dtype, int out_dtype, int res_dtype, void* stream);

/*
 * Two layers in one launch: a Conv (models/common.py:45-50, SiLU) and the pointwise Conv(s) that consume its output - in the CFT
 * networks the stride-2 Conv in front of a C3 and that C3's cv1 | cv2 (models/common.py:141-143, both read the same input and are
 * packed as one [n2][n1] weight).  y = act2(conv1x1(SiLU(conv(x)))):  the first layer's output tile is rounded to dtype in LDS and
 * becomes the second GEMM's A operand, so the n1-channel tensor between the layers is never written or read.  Results are
 * bit-identical to the two cft_conv2d launches.  x / w1 / bias1 / geometry as cft_conv2d; w2: dtype [n2][n1] (k = 1), bias2
 * float[n2] or NULL; y: dtype, ldy / yoff.  Eligible pairs only (cft_conv2d_chain_ok returns 1): bf16 / fp16, n1 == 128 (192 x 128
 * tile, two workgroups per CU, second layer's weights resident in LDS) or n1 == 256 (256 x 256 tile, 160 KiB of LDS, second
 * layer's weights streamed a K step at a time), cin % 64 == 0, kpad1 == k*k*cin, n2 <= n1; anything else is CFT_EINVAL.
 */
int cft_conv2d_chain(const void* x, const void* w1, const float* bias1, const void* w2, const float* bias2, void* y,
                     int B, int H, int W, int cin, int ldx, int xoff,
                     int n1, int kpad1, int ksize, int stride, int n2,
                     int ldy, int yoff, int act2, int dtype, void* stream);
/* 1 when cft_conv2d_chain accepts the pair (it runs the launcher's own validation, incl. the 2^31-element limits on the ldx / ldy extents). */
int cft_conv2d_chain_ok(int B, int H, int W, int cin, int ldx, int n1, int kpad1, int ksize, int stride, int n2, int ldy, int dtype);

/*
 * The chained pair with a SHORTCUT on the first layer - inside a C3 with shortcuts (models/common.py:99-109, :138-142) Bottleneck j's 3x3
 * conv and Bottleneck j+1's 1x1 conv:  y1 = SiLU(conv(x) + b1) + res  (one rounding; the Bottleneck's output AND the next shortcut),
 * y2 = act2(conv1x1(y1) + b2).  The shortcut tile is DMA-ed into the LDS images, added in fp32 before the rounding, and the images are
 * stored to y1 while the second GEMM runs: y1 is never re-read from memory and the 1x1 launch disappears.  Bit-identical to
 * cft_conv2d(x, w1, res=res) followed by cft_conv2d(y1, w2).  n1 == 256 only; otherwise the eligibility of cft_conv2d_chain_ok.
 * res: dtype, ldr / roff; y1: dtype, ldy1 / yoff1 (may alias res); y2: dtype, ldy2 / yoff2.
 */
int cft_conv2d_chain_res(const void* x, const void* w1, const float* bias1, const void* res, void* y1,
                         const void* w2, const float* bias2, void* y2,
                         int B, int H, int W, int cin, int ldx, int xoff,
                         int n1, int kpad1, int ksize, int stride, int ldr, int roff, int ldy1, int yoff1,
                         int n2, int ldy2, int yoff2, int act2, int dtype, void* stream);

/*
 * split-K nn.Linear (models/common.py:511 out_proj, :532-538 the MLP's second Linear): parts[s] (float [rows][n], s < splits) =
 * x[:, s*K/splits : (s+1)*K/splits] . w[:, same]^T (+ bias in s == 0).  For GEMMs with few output tiles and a long K loop: splits x the
 * workgroups, 1 / splits of the K steps each.  The partial sums are folded into the fp32 residual stream in a FIXED order (reproducible
 * results) by cft_layernorm_reduce.  cin % K-step == 0, kpad == cin, (kpad / K-step) % splits == 0, 2 <= splits <= 8.
 */
int cft_linear_splitk(const void* x, const void* w, const float* bias, float* parts,
                      int rows, int cin, int ldx, int n, int kpad, int splits, int dtype, void* stream);

/*
 * Bottleneck as one kernel (models/common.py:99-109 with e = 1.0, the form C3 uses :138):
 *   y = (shortcut ? x : 0) + SiLU(conv3x3(SiLU(conv1x1(x) + b1)) + b2),  c -> c -> c channels, 16-bit dtype.
 * x, y: NHWC channel slices (ldx/xoff, ldy/yoff) that must not overlap (the kernel reads a halo of x);
 * w1 [c][kpad1] and w2 [c][kpad2] in the cft_conv2d layout (BN folded).  Bit-identical to two cft_conv2d calls
 * (1x1 + SiLU, then 3x3 + SiLU + residual); the hidden tensor never reaches HBM.
 * w2_stages (required for c = 128, ignored for c = 64): the same 3x3 weights as 36 stage images of 8 KiB written by
 * cft_bottleneck_pack_w2; the kernel streams them as contiguous 1-KiB requests through a 4-slot LDS ring.
 */
int cft_bottleneck(const void* x, int ldx, int xoff, const void* w1, int kpad1, const float* b1,
                   const void* w2, int kpad2, const void* w2_stages, const float* b2, void* y, int ldy, int yoff,
                   int B, int H, int W, int c, int shortcut, int dtype, void* stream);

/* w2 [128][1152] (cft_conv2d layout of a 128 -> 128 3x3 conv) -> w2_stages: 36 x 8 KiB, stage u = k 32u .. 32u+31 of
 * every row in the order the kernel keeps it in LDS (c * kpad2 * 2 bytes, the size of w2). */
int cft_bottleneck_pack_w2(const void* w2, int kpad2, int c, void* w2_stages, int dtype, void* stream);

/* Tuning knob: force one tile configuration of cft_conv2d (0 = automatic, the default; see
 * csrc/conv_gemm.hip for the table).  The setting is PER HOST THREAD (thread_local): it reaches the launches the calling
 * thread issues and no other's.  Returns the previous value.  Not needed for normal use. */
int cft_set_conv_variant(int variant);

/*
 * Focus space-to-depth (models/common.py:176-179, the torch.cat of four strided slices):
 *   out[b, y, x, q*3 + c] = in[b, c, 2y+dy, 2x+dx],  q = dy + 2*dx, channels 12..15 = 0
 * in : float NCHW [B,3,H,W] contiguous (the image batch, values in [0,1)); out : dtype NHWC
 * [B,H/2,W/2,16].  The 3x3 Conv of Focus then runs through cft_conv2d with cin = 16.
 */
int cft_focus_s2d(const float* in, void* out, int B, int H, int W, int dtype, void* stream);

/*
 * Same, for the uint8 images the reference's callers hold (one [B,6,H,W] uint8 tensor: RGB = channels 0-2,
 * IR = 3-5; test.py:106-113 does `.float()/255` and the split before calling the model).  `in` points at the
 * first channel of the stream, element strides in bytes for batch / channel / row (row elements contiguous);
 * out[...] = in[...] * scale (scale = 1/255).
 */
int cft_focus_s2d_u8(const unsigned char* in, long stride_b, long stride_c, long stride_h, void* out,
                     int B, int H, int W, float scale, int dtype, void* stream);

/*
 * Focus in one kernel: the space-to-depth above plus its 3x3 Conv (+ folded BN, + SiLU; models/common.py:168-179
 * with :45-50) straight from the image, 16-bit compute (dtype = CFT_BF16 or CFT_F16), no intermediate tensor.
 * `in` is the first channel of the stream: float (in_kind = 0, scale = 1), unsigned char (in_kind = 1,
 * scale = 1/255; test.py:106-113) or half (in_kind = 2, the `img.half()` of test.py:107; dtype CFT_F16 only);
 * element strides for batch / channel / row, row elements contiguous, pointer and strides multiples of 2 elements.
 * w: dtype [n][192], k = (kh*3 + kw)*16 + ci, ci < 12 real (the cft_conv2d layout for cin = 16); n in {32,48,64,80};
 * y: dtype NHWC [B,H/2,W/2] with ldy/yoff.  Bit-identical to cft_focus_s2d(_u8) followed by cft_conv2d.
 */
int cft_focus_conv(const void* in, int in_kind, long stride_b, long stride_c, long stride_h, float scale,
                   const void* w, int kpad, const float* bias, void* y, int ldy, int yoff,
                   int B, int H, int W, int n, int act, int dtype, void* stream);

/*
 * SPP max pools (models/common.py:161-165): reads channels [0,C) of the NHWC buffer `buf`
 * (ld channels/pixel) and writes max_pool2d(k, stride 1, pad k/2) for k = k1,k2,k3 to channel
 * slices [C,2C), [2C,3C), [3C,4C) of the same buffer.  k odd, <= 13, k1 <= k2 <= k3.
 */
int cft_spp_maxpool(void* buf, int B, int H, int W, int C, int ld, int k1, int k2, int k3,
                    int dtype, void* stream);

/*
 * Channel-slice copy with optional nearest-neighbour upsampling (nn.Upsample(None,2,'nearest')
 * + Concat, yaml rows 33-34 / models/common.py:217-219):
 *   out[b, y, x, ooff + c] = in[b, y >> up, x >> up, ioff + c],  c < C;  out is [B,Ho,Wo,ldo].
 */
int cft_copy_channels(const void* in, int ldi, int ioff, void* out, int ldo, int ooff,
                      int B, int Ho, int Wo, int C, int up, int dtype, void* stream);

/*
 * Layout / dtype conversion at the boundary: any strided [B,C,H,W] tensor (in_dtype, element strides) -> NHWC
 * channel slice [ooff, ooff + cpad) of `out` in `dtype`, channels [C, cpad) zero (cpad a granule multiple).  This is what lets a module of
 * models/common.py be called with an ordinary NCHW torch tensor (as the reference's modules are) without ATen.
 */
int cft_to_nhwc(const void* in, int in_dtype, long stride_b, long stride_c, long stride_h, long stride_w,
                void* out, int ldo, int ooff, int B, int C, int cpad, int H, int W, int dtype, void* stream);

/*
 * Letterbox on the device (utils/datasets.py:1698-1728 `letterbox`: cv2.resize(INTER_LINEAR) to resized_w x resized_h,
 * then cv2.copyMakeBorder with a constant colour), 8-bit 3-channel images.  src: HWC, row stride in bytes; dst element
 * (y, x, c) at dst + y*stride_y + x*stride_x + c'*stride_c with c' = flip_channels ? 2 - c : c - strides (w*3, 3, 1) give
 * cv2's HWC image, (w, 1, h*w) with flip = 1 gives the CHW RGB plane the callers build next (datasets.py:1276-1281).
 * The geometry (resized size, top/left) is computed by the caller exactly as the reference does; colour = border value.
 */
int cft_letterbox_u8(const unsigned char* src, int src_h, int src_w, long src_row_stride,
                     unsigned char* dst, int dst_h, int dst_w, long dst_stride_y, long dst_stride_x, long dst_stride_c, int flip_channels,
                     int resized_h, int resized_w, int top, int left, int color0, int color1, int color2, void* stream);

/* Elementwise out = a + b over M pixels x C channels (Add / Add2, models/common.py:228-243). */
int cft_add(const void* a, int lda, int aoff, const void* b, int ldb, int boff,
            void* out, int ldo, int ooff, long M, int C, int dtype, void* stream);

/*
 * CFT tokeniser (models/common.py:608-621): AdaptiveAvgPool2d((8,8)) of both streams, flatten,
 * concat on the token axis (RGB tokens 0..63, IR tokens 64..127), + pos_emb.
 *   tokens[b, s*64 + i*8 + j, c] = mean(window(i,j) of stream s)[c] + pos_emb[s*64+i*8+j, c]
 * window rows [floor(i*H/8), ceil((i+1)*H/8)).  rgb/ir: dtype NHWC; tokens: float [B,128,C].
 */
int cft_gpt_tokenize(const void* rgb, int ld_rgb, int off_rgb, const void* ir, int ld_ir, int off_ir,
                     const float* pos_emb, float* tokens, int B, int H, int W, int C,
                     int dtype, void* stream);

/* LayerNorm over the last dim (eps 1e-5, affine; models/common.py:529-530,572):
 * x float [rows, C] -> y out_dtype [rows, C]. */
int cft_layernorm(const float* x, const float* gamma, const float* beta, void* y,
                  long rows, int C, float eps, int out_dtype, void* stream);
/* x (float [rows, C], updated in place) += parts[0] + ... + parts[nparts-1] (float [nparts][rows][C], cft_linear_splitk), then
 * y = LayerNorm(x): the residual add of models/common.py:543-544 and the LayerNorm of the next sub-block (:529-530, :572) in one pass. */
int cft_layernorm_reduce(float* x, const float* parts, int nparts, const float* gamma, const float* beta, void* y,
                         long rows, int C, float eps, int out_dtype, void* stream);

/*
 * Multi-head self-attention core (models/common.py:491-510): for each (b, head)
 *   O = softmax(Q K^T / sqrt(dk)) V  over T = 128 tokens.
 * qkv : dtype [B*128, 3*heads*dkp]: row = token, columns [which(q,k,v)][head][dkp]; dkp is the
 * head width padded with zeros to a multiple of 32 (bf16) / 16 (f32); dk the true head width.
 * out : dtype [B*128, heads*dkp].
 * attn_pdrop / seed: training-mode dropout of the attention probabilities (models/common.py:507 `attn_drop`), applied
 * inside the kernel with the counter-based mask of cft_dropout; 0 = inference.
 */
int cft_attention(const void* qkv, void* out, int B, int heads, int dk, int dkp,
                  int dtype, float attn_pdrop, unsigned long long seed, void* stream);

/*
 * CFT de-tokeniser fused with the residual add (models/common.py:626-637 + Add2 :238-243):
 *   out[b,y,x,c] = (base ? base[b,y,x,c] : 0) + bilinear_{8x8 -> HxW, align_corners=False}(tokens[b, s*64 + ., c])
 * tokens: float [B,128,C] (already through ln_f); s selects the stream (0 RGB, 1 IR).
 */
int cft_gpt_upsample_add(const float* tokens, int s, const void* base, int ldb, int boff,
                         void* out, int ldo, int ooff, int B, int H, int W, int C,
                         int dtype, void* stream);

/*
 * The same for BOTH streams of a CFT block in one launch, plus the Add that consumes the two results (models/common.py:626-637 twice,
 * Add2 :238-243 twice, Add :228-229):  out0 = base0 + up(tokens[:, :64]), out1 = base1 + up(tokens[:, 64:]),
 * sum (may be NULL) = out0 + out1 formed in fp32 before the one rounding.  out0 / out1 are bit-identical to two cft_gpt_upsample_add calls.
 */
int cft_gpt_upsample_add2(const float* tokens, const void* base0, int ldb0, int boff0, const void* base1, int ldb1, int boff1,
                          void* out0, int ldo0, int ooff0, void* out1, int ldo1, int ooff1, void* sum, int lds, int soff,
                          int B, int H, int W, int C, int dtype, void* stream);

/*
 * Detect decode (models/yolo_test.py:47-57).  logits: float [B,ny,nx,ldl] holding na*no valid
 * channels (channel = a*no + o), the output of the 1x1 conv.  Writes
 *   raw [B,na,ny,nx,no]            = logits permuted (the reference's x[i])
 *   pred[B, row0 + (a*ny+y)*nx+x, o] with total_rows rows per image:
 *        xy = (2*sig - 0.5 + grid) * stride,  wh = (2*sig)^2 * anchor[a],  rest = sig
 * anchors: float[na*2] in pixels (anchor_grid of this level).
 */
int cft_detect_decode(const float* logits, int ldl, float* raw, float* pred, const float* anchors,
                      int B, int ny, int nx, int na, int no, float stride,
                      long row0, long total_rows, void* stream);

/*
 * Training-mode forward (forward only, no autograd).
 *
 * cft_batchnorm_train: BatchNorm2d with BATCH statistics on the fp32 conv output x [M, ldx] (channels xoff..xoff+C), as
 * `act(bn(conv(x)))` does when `bn.training` (models/common.py:45-47): biased variance for the normalisation, running
 * statistics updated in place with `momentum` and the unbiased variance (torch semantics; NULL = do not track), then
 * SiLU / none, optional residual add (Bottleneck shortcut, :108-109) and the store into an NHWC channel slice in
 * `out_dtype`.  workspace: cft_batchnorm_train_workspace(M, C) bytes of device memory.
 *
 * cft_dropout: in-place nn.Dropout(p) in training mode on a contiguous tensor of n elements (GPT.drop :611, resid_drop
 * :511, the MLP's Dropout :537): element i is kept iff hash(seed, i) >= p * 2^32 and scaled by 1/(1-p).
 */
long cft_batchnorm_train_workspace(long M, int C);
int cft_batchnorm_train(const float* x, int ldx, int xoff, long M, int C,
                        const float* gamma, const float* beta, float* running_mean, float* running_var,
                        float momentum, float eps, const void* res, int ldr, int roff, int res_dtype,
                        void* y, int ldy, int yoff, int act, int out_dtype,
                        void* workspace, long workspace_bytes, void* stream);
int cft_dropout(void* x, long n, float p, unsigned long long seed, int dtype, void* stream);

/*
 * Batched NMS on the decoded predictions (utils/general.py:455-543 `non_max_suppression`, incl. the
 * torchvision.ops.nms call at :527): per image keep rows with obj > conf_thres, conf = obj*cls, best class
 * (multi_label = 0) or every class above conf_thres (multi_label = 1), optional class filter (class_allow:
 * device array of no-5 bytes, non-zero = class kept, :505-506; NULL = all classes), xywh -> xyxy, the max_nms
 * pre-truncation to the highest confidences (:469,:515-516; 0 = off), per-class greedy NMS (class offset
 * 4096 px unless agnostic) with IoU > iou_thres suppression, at most max_det detections.
 *   pred    : float [B, rows, no]            dets : float [B, max_det, 6] (x1,y1,x2,y2,conf,cls), first counts[b] rows valid,
 *                                                   the others zeroed
 *   scratch : >= B * round_up(rows * (multi_label ? no-5 : 1), 4) * 32 bytes of device memory, 16-byte aligned
 */
int cft_nms(const float* pred, int B, int rows, int no, float conf_thres, float iou_thres,
            int agnostic, int multi_label, const unsigned char* class_allow, int max_det, int max_nms,
            void* scratch, long scratch_bytes, float* dets, int* counts, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CFT_HIP_H */
