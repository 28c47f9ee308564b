/*
 * unscreen_hip.h - C ABI of libunscreen_hip.so: the MI355X (gfx950) pixel stages of the
 * per-frame video-unscreen path.
 *
 * The reference (AnyiRao/video_unscreen) is pure Python; the interface this library sits behind is
 * its agent API (unscreen/<stage>/agent.py classes and unscreen/utils functions).  Each entry point
 * names the reference code it replaces as file:line (paths relative to the reference repo root).
 * The Python side binds these with ctypes (video_unscreen_amd/_native.py); INTEGRATION.md shows the
 * stub a reference maintainer would add.
 *
 * Conventions
 *  - plain pointers and sizes only; images are uint8, HWC, BGR, densely packed; masks uint8 HW;
 *  - every pointer not marked "host" is a DEVICE pointer (e.g. torch.Tensor.data_ptr());
 *  - every call enqueues work on `stream` (a hipStream_t, e.g. torch.cuda.current_stream().cuda_stream)
 *    and returns without synchronising; no allocation, no global mutable state, re-entrant;
 *  - the caller owns every buffer, including the scratch `ws` whose size the matching
 *    *_ws_bytes() function reports;
 *  - return value: 0 = ok, < 0 = error (USC_E_*); usc_last_error() gives the thread-local message.
 */
#ifndef UNSCREEN_HIP_H
#define UNSCREEN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* usc_stream_t;               /* hipStream_t */

#define USC_OK            0
#define USC_E_ARG        -1               /* bad argument (null pointer, non-positive size, ...) */
#define USC_E_UNSUPPORTED -2              /* parameter combination outside the kernels' limits */
#define USC_E_WORKSPACE  -3               /* workspace too small */
#define USC_E_LAUNCH     -4               /* HIP reported a launch error */

#define USC_MORPH_DILATE 0
#define USC_MORPH_ERODE  1
#define USC_LAYOUT_NCHW  0
#define USC_LAYOUT_NHWC  1
#define USC_LAYOUT_NHWC8 2                /* 7 channels + one zero channel per pixel (32-byte pixels for the HIP convs) */

int         usc_abi_version(void);
const char* usc_last_error(void);

/* Geometry helpers (host, pure): unscreen/utils/imgprocess.py:164-192 get_target_size,
 * :67-97 pad_resize (size of the resized image inside the padded canvas),
 * :100-132 inv_pad_resize (size of the un-padded region). */
void usc_get_target_size(int h, int w, int long_side, int division, int* target_h, int* target_w);
void usc_pad_resize_geometry(int h, int w, int target_h, int target_w, int* new_h, int* new_w);
void usc_inv_pad_geometry(int h, int w, int ori_h, int ori_w, int* resized_h, int* resized_w);

/* The NumPy-level forms of the same helpers, behind the names the reference's tools and agents import from unscreen.utils
 * (tools/unscreen/bg_offline.py:16, tools/replace/replace.py:14, unscreen/iseg/agent.py:5); the frame loops use the fused
 * kernels further down.
 * usc_pad_resize_u8: unscreen/utils/imgprocess.py:67-97 pad_resize on uint8, cn = 1 | 3: cv2.resize (default INTER_LINEAR, incl.
 *   OpenCV's same-size copy and exact-2x box reroute) to the aspect-preserving size, cv2.copyMakeBorder(BORDER_REFLECT) at the
 *   bottom / right up to (target_h, target_w) - one kernel.
 * usc_inv_pad_resize: imgprocess.py:100-132 inv_pad_resize: the bottom / right pad is cut, the rest resized to (ori_h, ori_w);
 *   elem = USC_ELEM_U8 (fixed point), USC_ELEM_F32 / USC_ELEM_F64 (float weights; the float32 alpha of vmatting/agent.py:127 and
 *   probability of iseg/agent.py:117, the float64 scores of binseg/agent.py:233), cn = 1..4 interleaved channels.
 * usc_imnormalize_u8: imgprocess.py:135-161 imnormalize on n pixels: BGR -> RGB IN PLACE on `img` when to_rgb (as the reference
 *   does to its argument, :157), out = ((float32(img) / 255) - mean) * (1 / std), float32 operations; mean / std: host, 3 doubles.
 * usc_gray_moments_u8: imgprocess.py:7-14 get_center (rgb_order = 1: cv2.COLOR_RGB2GRAY; moments3 = m00, m10, m01 of the gray
 *   image as exact integers) and :17-30 get_mask (rgb_order = 0, mask = 255 where gray > thr); either output may be NULL.
 * usc_gray_moments_pair_u8: the moments of TWO images of one size in one launch (moments6 = a's m00, m10, m01, then b's): the two
 *   get_center calls per frame of tools/replace/replace.py:32-35 (comp_dx_dy). */
#define USC_ELEM_U8  0
#define USC_ELEM_F32 1
#define USC_ELEM_F64 2
int usc_pad_resize_u8(const uint8_t* src, int h, int w, int cn, uint8_t* dst, int target_h, int target_w, usc_stream_t stream);
int usc_inv_pad_resize(const void* src, int h, int w, int cn, int elem, void* dst, int ori_h, int ori_w, usc_stream_t stream);
int usc_imnormalize_u8(uint8_t* img, size_t n, const double* mean3_host, const double* std3_host, int to_rgb, float* out, usc_stream_t stream);
int usc_gray_moments_u8(const uint8_t* img, int H, int W, int rgb_order, int thr, uint8_t* mask, unsigned long long* moments3,
                        usc_stream_t stream);
int usc_gray_moments_pair_u8(const uint8_t* img_a, const uint8_t* img_b, int H, int W, int rgb_order, unsigned long long* moments6,
                             usc_stream_t stream);

/* ISegAgent, no-BRS predictor (unscreen/iseg/agent.py:92-119, iseg/brs/brs_predictor.py:35-106).  The network's dense layers run
 * on usc_conv_* (video_unscreen_amd/iseg/net_hip.py); these are its pixel stages and the one non-dense layer type.
 * usc_iseg_pre: agent.py:92-104 preprocess + the model's input stage (iseg/brs/models/distmap.py:16-22, 30-34, 70-124): pad_resize of
 *   the uint8 BGR frame to (in_h, in_w), imnormalize, the two click distance maps tanh(2 sqrt(min_p |x - p|^2) / 260) and rgb_conv
 *   (Conv 5->8, LeakyReLU 0.2, BN, Conv 8->3).  points_yx_host: 2 P (y, x) float pairs in NETWORK-INPUT coordinates, first the P
 *   positive slots then the P negative ones, (-1, -1) = empty slot (brs_predictor.py:64-99 get_points_nd; 1 <= P <=
 *   USC_ISEG_MAX_CLICKS = its net_clicks_limit).  rgb_params_host: 75 floats w1[8][5], b1[8], w2[3][8], b2[3] with the BatchNorm
 *   folded into the second layer.  out: (in_h, in_w, 4) float32 NHWC (R', G', B', 0); with flip a second image follows, the
 *   mirror of the first (transforms.py:34-58 AddHorizontalFlip mirrors image and clicks: the same values, mirrored).
 * usc_dwconv3x3_nhwc_f32: the depthwise 3x3 (padding 1, no bias) of SeparableConv2d (iseg/brs/models/basic_blocks.py:101-113) on
 *   `batch` stacked NHWC images; w9c = weights [9][C] tap-major.  Not a dense contraction: a direct kernel, off the matrix cores.
 * usc_iseg_post: brs_predictor.py:46-51 (bilinear align_corners up to the input size, flip pair averaged on the logits, sigmoid)
 *   -> prob (in_h, in_w) float32; then, when mask != NULL, agent.py:116-119: inv_pad_resize of prob to (H, W), > prob_thresh -> 255 / 0.
 *   logits: channel 0 of (nimg * lh, lw, cbuf) float32, nimg = 2 with flip. */
#define USC_ISEG_MAX_CLICKS 20
int usc_iseg_pre(const uint8_t* frame, int H, int W, int in_h, int in_w, const float* points_yx_host, int P,
                 const float* rgb_params_host, int flip, float* out, usc_stream_t stream);
int usc_dwconv3x3_nhwc_f32(const float* src, int h, int w, int C, int batch, const float* w9c, float* dst, usc_stream_t stream);
int usc_iseg_post(const float* logits, int lh, int lw, int cbuf, int in_h, int in_w, int flip, int H, int W, double prob_thresh,
                  float* prob, uint8_t* mask, usc_stream_t stream);

/* unscreen/utils/maskprocess.py:7-34 dilate_mask / erode_mask: grey-level max / min filter with
 * cv2.getStructuringElement(MORPH_ELLIPSE, (ksize, ksize)), `iters` passes, out-of-image taps ignored.
 * src and dst may not alias. */
size_t usc_morph_ws_bytes(int H, int W, int ksize, int iters);
int usc_morph_u8(const uint8_t* src, uint8_t* dst, int H, int W, int ksize, int iters, int op,
                 void* ws, size_t ws_bytes, usc_stream_t stream);

/* unscreen/trimap/agent.py:35-61 TrimapAgent.generate_trimap: NEAREST down to
 * get_target_size(H, W, long_side), dilate + erode (ksize, iters), {0,128,255}, bilinear back to H x W,
 * intermediate values -> 128. */
size_t usc_trimap_ws_bytes(int H, int W, int long_side);
int usc_trimap_u8(const uint8_t* mask, int H, int W, int long_side, int ksize, int iters, uint8_t* out,
                  void* ws, size_t ws_bytes, usc_stream_t stream);

/* unscreen/utils/fgfuncs.py:9-65 is_pixel_inrange (long_side_input <= 0): out[i] = 1 when every 8-bit HSV channel of
 * img lies within clip(bg -/+ winsize//2, 10, 255).  Exactly one of bg_img (device, H*W*3) and bg_color_host
 * (HOST pointer to 3 bytes BGR) must be given; winsize is a HOST pointer to 3 ints (h, s, v). */
int usc_inrange_hsv_u8(const uint8_t* img, const uint8_t* bg_img, const uint8_t* bg_color_host, const int* winsize,
                       int H, int W, uint8_t* out, usc_stream_t stream);

/* unscreen/trimap/agent.py:63-101 TrimapAgent.generate_trimap_withbg, entirely on the device (the
 * fuzzy / mask ratio test of :94 is evaluated by the kernels from device-side counters, no host sync). */
size_t usc_trimap_withbg_ws_bytes(int H, int W, int long_side);
int usc_trimap_withbg_u8(const uint8_t* mask, const uint8_t* img, const uint8_t* bg_img, const uint8_t* bg_color_host,
                         const int* winsize, int H, int W, int long_side, int ksize, int iters, uint8_t* out,
                         void* ws, size_t ws_bytes, usc_stream_t stream);

/* tools/unscreen/bg_offline.py:150-160 (same arithmetic: tools/unscreen/bg.py:85-92): background fusion
 * bg = u8(bg_i*beta + (1-beta)*bg_always) and the background-difference gate
 * alpha = gray(seg) * (dilate4x4x2(gray(|frame - bg|) > thr) // 255).
 * bg_always may be NULL with beta = 1 (bg.py: a single background image).  seg_cn is 1 or 3.  One kernel (bgdiff_gate_kernel):
 * the difference image lives as bits in LDS; ws may be NULL (usc_bgdiff_ws_bytes returns 0 since round 3 and stays for callers
 * that size a buffer with it).  out_bg must not alias an input. */
size_t usc_bgdiff_ws_bytes(int H, int W);
int usc_bgdiff_mask_u8(const uint8_t* frame, const uint8_t* bg_i, const uint8_t* bg_always,
                       const uint8_t* seg, int seg_cn, double beta, int thr, int H, int W,
                       uint8_t* out_bg, uint8_t* out_alpha, void* ws, size_t ws_bytes, usc_stream_t stream);

/* unscreen/vmatting/agent.py:76-106 preprocess (+ :44-74 to_tensor, unscreen/utils/imgprocess.py:135-161
 * imnormalize, model.py:564 torch.cat): frame, previous alpha and trimap -> the network's packed
 * 7-channel float input of size inH x inW (bilinear pad_resize + BORDER_REFLECT, ImageNet normalise in RGB
 * order, alpha/255, one-hot trimap). */
int usc_vmat_pre(const uint8_t* frame, const uint8_t* alpha_pre, const uint8_t* trimap, int H, int W,
                 int inH, int inW, float* out7, int layout, usc_stream_t stream);

/* unscreen/vmatting/agent.py:108-132 postprocess: crop the pad, float bilinear back to H x W, force known
 * trimap regions, (a * 255) truncated to uint8.  alpha_f32 (optional, may be NULL) gets the float alpha. */
int usc_vmat_post(const float* pred, int inH, int inW, const uint8_t* trimap, int H, int W,
                  uint8_t* alpha_u8, float* alpha_f32, usc_stream_t stream);

/* unscreen/utils/fgfuncs.py:84-110 get_fg, optionally preceded by the caller's paste
 * bg[alpha < paste_thr] = frame[alpha < paste_thr] (tools/unscreen/bg_offline.py:171 and bg.py:99 use
 * alpha == 0 -> paste_thr 1; green.py:125 uses alpha < 128; 0 = no paste).
 * out_bg (optional, may be NULL) receives the pasted background. */
int usc_get_fg_u8(const uint8_t* frame, const uint8_t* alpha, const uint8_t* bg, int paste_thr, int H, int W,
                  uint8_t* out_fg, uint8_t* out_bg, usc_stream_t stream);

/* The two calls above as ONE launch, for loops that run them back to back on every frame (tools/unscreen/bg_offline.py:165-173:
 * vmattingagent.forward(...) followed by bg[alpha == 0] = frame[alpha == 0]; get_fg(frame, alpha, bg)): alpha_u8 and out_fg receive
 * exactly what usc_vmat_post followed by usc_get_fg_u8(paste_thr) writes (the same device code; the alpha of a pixel stays in a register). */
int usc_vmat_post_fg(const float* pred, int inH, int inW, const uint8_t* trimap, const uint8_t* frame, const uint8_t* bg, int paste_thr,
                     int H, int W, uint8_t* alpha_u8, float* alpha_f32, uint8_t* out_fg, usc_stream_t stream);

/* unscreen/utils/fgfuncs.py:113-137 get_bg. */
int usc_get_bg_u8(const uint8_t* alpha, const uint8_t* bg, int H, int W, uint8_t* out, usc_stream_t stream);

/* tools/replace/replace.py:74-76 composite: u8(fg*m + bg*(1-m)), m = mask/255 in float64; mask_cn 1 or 3. */
int usc_blend_u8(const uint8_t* fg, const uint8_t* mask, int mask_cn, const uint8_t* bg, int H, int W,
                 uint8_t* out, usc_stream_t stream);

/* Replace mode (tools/replace/replace.py:59-82) geometry:
 * usc_resize_u8: cv2.resize on uint8, 1 or 3 channels; interp 0 = INTER_NEAREST, 1 = INTER_LINEAR (incl. OpenCV's exact-2x
 *   box reroute), 2 = INTER_CUBIC (unscreen/utils/imgprocess.py:33-37 adaptive_resize uses 1).
 * usc_rescale_fg_u8: imgprocess.py:40-52 rescale_fg - cubic zoom by `scale` and centre crop, fused (only the crop is computed).
 * usc_warp_affine_u8: cv2.warpAffine(LINEAR, zero border) with a host 2x3 double matrix (imgprocess.py:55-64 shift_fg
 *   passes [[1,0,dx],[0,1,dy]] as float32).
 * usc_composite_fgbg_u8: unscreen/utils/fgfuncs.py:172-214 composite_fgbg (extend=False), background resize fused in. */
int usc_resize_u8(const uint8_t* src, int sh, int sw, int cn, uint8_t* dst, int dh, int dw, int interp, usc_stream_t stream);
int usc_rescale_fg_u8(const uint8_t* src, int h, int w, int cn, double scale, uint8_t* dst, usc_stream_t stream);
int usc_warp_affine_u8(const uint8_t* src, int sh, int sw, int cn, const double* M2x3_host, uint8_t* dst, int dh, int dw,
                       usc_stream_t stream);
int usc_composite_fgbg_u8(const uint8_t* fg, const uint8_t* alpha, int fh, int fw, const uint8_t* bg, int bh, int bw,
                          uint8_t* out, usc_stream_t stream);

/* The per-frame body of tools/replace/replace.py:59-82 as ONE launch: res = u8(fg' * m' + bg * (1 - m')) with fg', m' =
 * rescale_fg(shift_fg(., M), scale) (imgprocess.py:55-64, :40-52; mask_cn = 1 | 3), m' / 255 and the blend in float64 (:74-76), and
 * compare = [adaptive_resize(src, (H, W)) | res] (:64, :82; (H, 2 W, 3)).  `bg` is the background already at (H, W) (:66 resizes it
 * once; a same-size cv2.resize is a copy).  Either output may be NULL (compare needs src).  Bit-identical to usc_warp_affine_u8 ->
 * usc_rescale_fg_u8 -> usc_blend_u8 (+ usc_resize_u8 of src); 21 bytes of HBM traffic per pixel instead of ~60. */
int usc_replace_compose_u8(const uint8_t* fg, const uint8_t* mask, int mask_cn, int H, int W, const double* M2x3_host, double scale,
                           const uint8_t* bg, const uint8_t* src, int sh, int sw, uint8_t* res, uint8_t* compare, usc_stream_t stream);

/* Green-screen mode (tools/unscreen/green.py:70-126).
 * usc_cvt_color_u8: cv2.cvtColor on uint8, n pixels; code USC_CVT_* (BGR2HSV: colorfiltering/agent.py:311; HSV2BGR: :352;
 *   BGR2LAB: utils/imgprocess.py:281-283, harmonization/agent.py:63-64; BGR2GRAY: tools/unscreen/bg_offline.py:152).  BGR2LAB
 *   needs `lab_tables`: a device copy of the usc_lab_tables_bytes() blob that usc_lab_tables_host() fills (others: NULL).
 * usc_gmm_lut_host: unscreen/colorfiltering/agent.py:201-230 get_prob_by_gmm evaluated at the 256 uint8 sample values
 *   (HOST, float32 arithmetic; means / stds / weights of one fitted 1-D mixture).
 * usc_gmm_alpha_u8: colorfiltering/agent.py:232-257 get_alpha_by_gmm; luts = device float[6][256] (bg h,s,v then fg h,s,v);
 *   prob_out (float per pixel) may be NULL.
 * usc_cf_threshold_u8: colorfiltering/agent.py:259-280 postprocess, adaptive threshold (in place on alpha); the
 *   close/open that follows is four usc_morph_u8 calls.
 * usc_color_correct_u8: unscreen/utils/imgprocess.py:263-300 color_correct; bg_color_host = 3 host bytes (B, G, R);
 *   *n_sqrt_out (device int, may be NULL) receives the number of sqrt passes the while loop took.
 * usc_box_filter_u8: cv2.boxFilter(img, -1, (k, k)) of unscreen/harmonization/agent.py:91-92, 110-111. */
#define USC_CVT_BGR2HSV  0
#define USC_CVT_HSV2BGR  1
#define USC_CVT_BGR2GRAY 2
#define USC_CVT_BGR2LAB  3
#define USC_CVT_LAB2BGR  4         /* cv2.COLOR_LAB2BGR, harmonization/agent.py:73; `lab_tables` = device copy of the usc_lab_inv_tables_host() blob */
size_t usc_lab_tables_bytes(void);
void usc_lab_tables_host(void* blob);
size_t usc_lab_inv_tables_bytes(void);
void usc_lab_inv_tables_host(void* blob);
int usc_cvt_color_u8(const uint8_t* src, size_t n, int code, uint8_t* dst, const void* lab_tables, usc_stream_t stream);
int usc_gmm_lut_host(const double* means, const double* stds, const double* weights, int K, float* lut256);
int usc_gmm_alpha_u8(const uint8_t* hsv, size_t n, const float* luts, uint8_t* alpha, float* prob_out, usc_stream_t stream);
size_t usc_cf_threshold_ws_bytes(void);
int usc_cf_threshold_u8(uint8_t* alpha, const uint8_t* mask, size_t n, double thr_ratio, void* ws, size_t ws_bytes,
                        usc_stream_t stream);
size_t usc_color_correct_ws_bytes(int H, int W, int long_side);
int usc_color_correct_u8(const uint8_t* img, const uint8_t* alpha, int H, int W, const uint8_t* bg_color_host, int long_side,
                         double mean_exp, const void* lab_tables, uint8_t* out, int* n_sqrt_out, void* ws, size_t ws_bytes,
                         usc_stream_t stream);
int usc_box_filter_u8(const uint8_t* src, int H, int W, int cn, int ksize, uint8_t* dst, usc_stream_t stream);

/* SegAgent (unscreen/binseg/agent.py), everything but the convolutions.
 * usc_segagent_target_size: agent.py:70-100 get_target_size (host).
 * usc_grid_crop_locations: agent.py:102-141 grid_crop locations, 4 ints (s_h, e_h, s_w, e_w) per crop, a mirrored crop has
 *   s_w > e_w; returns the crop count (locs == NULL: count only) or a negative error (host).
 * usc_segagent_pre: agent.py:176-198 preprocess of ONE crop (loc_host = its 4 ints): pad_resize + imnormalize + crop
 *   (+ cv2.flip), out = (crop_h, crop_w, 4) float32 NHWC (R, G, B, 0).
 * usc_im2col_nhwc_f32: patch matrix [OH*OW][Kpad] (k = (ky*kw + kx)*cuse + c, zero beyond kh*kw*cuse) of a k x k stride-s
 *   convolution over the first `cuse` channels of an NHWC input: the 7x7/2 ResNet stem (deeplab/backbone/resnet.py:183-184)
 *   then runs as a 1x1 layer of usc_conv_*.
 * usc_upsample_nc_nhwc_f32: F.interpolate(bilinear, align_corners=False) (deeplab/_deeplab.py:51-55) into channels
 *   [c0, c0 + C) of a destination with ld channels per pixel.
 * usc_segagent_post: agent.py:200-237 postprocess: the model's last upsample (deeplab/utils.py:20-21), softmax, float64 merge
 *   of the crops (agent.py:143-174), inv_pad_resize (utils/imgprocess.py:100-132), argmax * 255.  logits = ncrops
 *   consecutive (lh, lw, cbuf) NHWC buffers (classes in channels 0, 1); score_out (H*W*2 doubles) may be NULL. */
void usc_segagent_target_size(int h, int w, int long_side, int division, int crop_h, int crop_w, int* th, int* tw);
int usc_grid_crop_locations(int h, int w, int crop_h, int crop_w, double stride_ratio, int flip, int* locs, int max_crops);
int usc_segagent_pre(const uint8_t* img, int H, int W, int th, int tw, const int* loc_host, float* out, usc_stream_t stream);
int usc_im2col_nhwc_f32(const float* in, int IH, int IW, int cpitch, int cuse, int kh, int kw, int stride, int pad, float* out,
                        int OH, int OW, int Kpad, usc_stream_t stream);
int usc_upsample_nc_nhwc_f32(const float* src, int sh, int sw, int C, float* dst, int dh, int dw, int ld, int c0,
                             usc_stream_t stream);
/* nn.AdaptiveAvgPool2d(s) on channels [c0, c0 + C) of an (H, W, ld) NHWC float32 tensor -> dense (s, s, C)
 * (binseg/human_parse.py:92-96, deeplab/_deeplab.py:166); windows summed in a fixed order. */
int usc_adaptive_avgpool_nhwc_f32(const float* src, int H, int W, int ld, int c0, int C, float* dst, int s, usc_stream_t stream);
/* The same for `batch` images stacked along y in src ((batch*H, W, ld)) -> dst (batch*s, s, C); per image bit-identical. */
int usc_adaptive_avgpool_nhwc_batch_f32(const float* src, int H, int W, int ld, int c0, int C, float* dst, int s, int batch,
                                        usc_stream_t stream);
/* nn.MaxPool2d(kernel_size=3, stride=2, padding=1) of `batch` images stacked along y: src (batch*H, W, C) float32 NHWC -> dst
 * (batch*OH, OW, C), OH = (H - 1) / 2 + 1 (the ResNet stems: binseg/human_parse.py:210, binseg/deeplab/backbone/resnet.py:185,
 * stm/model.py:46,58,73,83, iseg/brs/models/resnetv1b.py:191).  C % 4 == 0, 16-byte aligned tensors. */
int usc_maxpool3x3s2_nhwc_f32(const float* src, int H, int W, int C, float* dst, int batch, usc_stream_t stream);
size_t usc_segagent_post_ws_bytes(int th, int tw);
int usc_segagent_post(const float* logits, int ncrops, int lh, int lw, int cbuf, const int* locs_host, int crop_h, int crop_w,
                      int th, int tw, int H, int W, uint8_t* out, double* score_out, void* ws, size_t ws_bytes,
                      usc_stream_t stream);

/* STMAgent (unscreen/stm/agent.py), everything but the convolutions.
 * usc_stm_pre: agent.py:62-98 preprocess of one frame (+ the first frame's mask, or NULL): out = (in_h, in_w, 8) float32 NHWC
 *   (R, G, B, object, background, 0, 0, 0), (in_h, in_w) = usc_get_target_size(H, W, long_side, 16).
 * usc_pack_kn_f32: dst[k/4][n][k%4] = src[k*sk + n*sn] (zero padded to Kpad x Npad): puts a matrix that lives in HBM (memory
 *   keys / values) into the weight layout of usc_conv_*, so the two products of the memory read (stm/model.py:137-156) run as
 *   1x1 layers.
 * usc_row_softmax_f32: in place softmax(x / div) over the first `cols` of `ld` columns of every row (stm/model.py:148-150).
 * usc_stm_post: decoder logits p2 (lh, lw, cbuf) -> x4 bilinear, softmax, soft aggregation, softmax (stm/model.py:135-136,
 *   195-219; agent.py:164) -> prob (in_h, in_w, 2) float32 (written) -> inv_pad_resize + argmax * 255 (agent.py:172-201). */
int usc_stm_pre(const uint8_t* frame, const uint8_t* mask, int H, int W, int in_h, int in_w, float* out, usc_stream_t stream);
int usc_pack_kn_f32(const float* src, int K, int N, long long sk, long long sn, int Kpad, int Npad, float* dst, usc_stream_t stream);
int usc_row_softmax_f32(float* x, int rows, int cols, int ld, float div, usc_stream_t stream);
int usc_stm_post(const float* p2, int lh, int lw, int cbuf, int in_h, int in_w, int H, int W, uint8_t* out, float* prob,
                 usc_stream_t stream);
/* The second half of usc_stm_post alone: (in_h, in_w, 2) float32 probabilities -> inv_pad_resize + argmax * 255 (the
 * reference derives the first frame's output mask this way from the one-hot of mask0, agent.py:127, 172-201). */
int usc_stm_mask_from_prob(const float* prob, int in_h, int in_w, int H, int W, uint8_t* out, usc_stream_t stream);

/* Background inpainting of the online mode (unscreen/utils/region_fill.py:7-63 regionfill, per channel from
 * tools/unscreen/bg.py:79): inside mask > 0 the three channels of img are replaced by the solution of the discrete Laplace
 * equation with the surrounding pixels as boundary values - conjugate gradients preconditioned by a multigrid V-cycle, float64
 * (csrc/usc_regionfill.hip; the reference calls a sparse direct solver: results agree to ~1e4 * rtol grey levels, i.e. +-1 level
 * after the reference's astype(uint8) truncation, only next to integers).
 * Like every device entry it only enqueues work: iterations [first_iter, first_iter + n_iters) are launched, the convergence test
 * (every channel's residual norm reduced by rtol) runs ON THE DEVICE every two iterations, and once it has passed the kernels
 * still queued return at once - so n_iters is a budget, not a count.  first_iter = 0 starts a solve (ws is initialised);
 * first_iter > 0 continues the solve whose state is in ws (same img / mask / size) when a budget turned out too small.  `out`
 * receives the current iterate truncated to uint8 either way.  status (DEVICE int[2], may be NULL) = {iterations run so far,
 * converged 0 | 1}.  No host synchronisation, no state outside ws. */
size_t usc_regionfill_ws_bytes(int H, int W);
int usc_regionfill_u8(const uint8_t* img, const uint8_t* mask, int H, int W, uint8_t* out, double rtol, int first_iter, int n_iters,
                      int* status, void* ws, size_t ws_bytes, usc_stream_t stream);

/* remove_invalid_objects on the device (unscreen/utils/maskprocess.py:77-152) - no border following, no host round trip:
 * connected-component labels + per-crack shoelace terms + containment-tree sums (csrc/usc_objects.hip).  alpha is modified
 * in place when src is NULL (or == alpha); otherwise alpha receives the filtered copy of src, which stays untouched (the copy
 * rides on the first scan: the reference's `alpha = remove_invalid_objects(cfg, alphaor.copy())` pattern without a pass of
 * its own).  seg may be NULL (consensus against alpha itself); (center_y, center_x) = the score map's centre as ratios
 * (build_score_map, maskprocess.py:181-189: landscape / portrait entry of cfg) - the map itself (maskprocess.py:155-178) is
 * evaluated per pixel inside the kernels, and its sums are kept in integer fixed point: results are the same on every run.
 * usc_score_map_f64 builds the map as a float64 table (tests; ws8 = 8 bytes of device scratch). */
int usc_score_map_f64(int H, int W, double center_y, double center_x, double* out, void* ws8, usc_stream_t stream);
size_t usc_remove_invalid_objects_ws_bytes(int H, int W);
int usc_remove_invalid_objects_u8(const uint8_t* src, uint8_t* alpha, const uint8_t* seg, int H, int W, double center_y, double center_x,
                                  double saliency_thr, double consensus_thr, void* ws, size_t ws_bytes, usc_stream_t stream);

/* Overall background of a clip (tools/unscreen/bg_offline.py:99-130; the stage is switched off with `if 0:` in the reference).
 * usc_bg_accumulate_f32: one frame into the float32 accumulators raw (H*W*3) and cont (H*W; the reference keeps three equal
 *   channels): raw += frame * (1 - mask // 255), cont += (mask < 250); mask = the frame's segmentation mask after dilate 3x3 x2.
 *   Sums of integers: exact in float32, so per-rank partial sums can be all-reduced (the one collective on a data path).
 * usc_bg_finalize_u8: mask_always = cont <= 10 (255 / 0), bg_always = uint8(clip(raw / max(cont, 1))) zeroed where always masked.
 *   The reference then inpaints dilate3x3x2(mask_always) with cv2.inpaint(TELEA); this build fills it with usc_regionfill_u8. */
int usc_bg_accumulate_f32(const uint8_t* frame, const uint8_t* mask_dilated, int H, int W, float* raw, float* cont, usc_stream_t stream);
int usc_bg_finalize_u8(const float* raw, const float* cont, int H, int W, uint8_t* bg_always, uint8_t* mask_always, usc_stream_t stream);

/* unscreen/utils/maskprocess.py:56-60 exist_foreground: *out_count (device, 8 bytes) = #(mask >= thr). */
int usc_count_ge_u8(const uint8_t* mask, size_t n, int thr, unsigned long long* out_count, usc_stream_t stream);

/* unscreen/utils/maskprocess.py:77-189 remove_invalid_objects (+ get_score_map / build_score_map).
 * HOST function on HOST pointers: contour tracing is sequential, so this stage runs on the CPU between
 * two device stages exactly where the reference runs it.  alpha (H*W) is modified in place; seg may be NULL
 * (then alpha itself is the consensus mask).  score_cache is optional caller-owned host scratch of
 * H*W doubles reused across calls of the same geometry (NULL = rebuilt per call like the reference). */
int usc_remove_invalid_objects_host(uint8_t* alpha, const uint8_t* seg, int H, int W,
                                    double center_y, double center_x, double saliency_thr, double consensus_thr,
                                    double* score_cache, int* score_cache_valid);

/* ---------------------------------------------------------------------------------------------
 * HumanSegAgent pre / post processing (reference unscreen/binseg/agent_human.py:262-312).
 * usc_humanseg_affine (host, pure): :247-260 _xywh2cs for the whole-frame box + :91-125 get_affine_transform
 *   (cv2.getAffineTransform = 6x6 LU in double); inv = 0 maps frame -> network input, inv = 1 the reverse.
 * usc_humanseg_pre: :277-294 cv2.warpAffine(frame -> in_h x in_w, LINEAR, zero border) + imnormalize; writes NHWC
 *   float32 with 4 channels (R, G, B, 0).
 * usc_humanseg_post: :296-311 + transform_logits :58-73: logits (lh x lw, NHWC with Cbuf >= C channels per pixel)
 *   -> bilinear align_corners to in_h x in_w -> inverse affine warp of every class to H x W -> argmax > 0 -> 255.  One kernel
 *   since round 3 (the upsampled tensor is staged per tile in LDS); ws may be NULL, usc_humanseg_post_ws_bytes returns 0. */
void   usc_humanseg_affine(int w, int h, int in_w, int in_h, int inv, double* M2x3);
int    usc_humanseg_pre(const uint8_t* frame, int H, int W, int in_h, int in_w, float* out_nhwc4, usc_stream_t stream);
size_t usc_humanseg_post_ws_bytes(int C, int in_h, int in_w);
int    usc_humanseg_post(const float* logits, int C, int Cbuf, int lh, int lw, int in_h, int in_w, int H, int W,
                         uint8_t* out, void* ws, size_t ws_bytes, usc_stream_t stream);
/* F.interpolate(bilinear, align_corners=True) of an NHWC float32 tensor (sh x sw x C) into channels [c0, c0 + C) of a
 * (dh x dw x ld) tensor: the glue between the conv segments of the parsing net (human_parse.py:108, 147-150, 186). */
int    usc_upsample_ac_nhwc_batch_f32(const float* src, int sh, int sw, int C, float* dst, int dh, int dw, int ld, int c0, int batch,
                                      usc_stream_t stream);    /* `batch` images stacked along y in src and dst */
int    usc_upsample_ac_nhwc_f32(const float* src, int sh, int sw, int C, float* dst, int dh, int dw, int ld, int c0,
                                usc_stream_t stream);

/* conv3x3(padding 1) of a bilinearly upsampled low-resolution map without the convolution (binseg/human_parse.py:83-111: the
 * PSP priors are s x s maps, s = 1, 2, 3, 6, upsampled with align_corners=True to the feature size and fed to a 3x3 bottleneck;
 * binseg/deeplab/_deeplab.py:51-55: the ASPP output, upsampled x4 with align_corners=False, is 256 of the 304 input channels of
 * the decoder's 3x3 conv).  The channel mixing of a tap commutes with the upsampling: conv(up(p)) = sum_taps shift_tap(up(p W_tap)).
 * q = (batch*sh, sw, 9*C) holds p W_tap in channels [tap*C, (tap+1)*C), taps in (dy, dx) row-major order; dst (batch*dh, dw, C)
 * = (accumulate ? dst : 0) + sum over the 9 taps of up(q_tap) sampled at (y + dy, x + dx), zero outside the image. */
int    usc_upsample_taps_accum_f32(const float* q, int sh, int sw, int C, float* dst, int dh, int dw, int batch, int accumulate,
                                   int align_corners, usc_stream_t stream);
const char* usc_seg_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Matting network convolutions (reference unscreen/vmatting/model.py:435-568: every nn.Conv2d /
 * nn.ConvTranspose2d of ResShortCut + ResShortCut_Dec with its BatchNorm / activation / residual).
 * One descriptor = one fused layer on NHWC float32 tensors:
 *   out[Y, X, n] = post( act( sum_{t, ci} in[oy*in_stride + dy[t], ox*in_stride + dx[t], ci] * W[t, ci, n]
 *                             + bias[n] + res1[Y >> res1_shift, X >> res1_shift, n] ) ) + res2[Y, X, n]
 *   (Y, X) = (oy*out_stride + out_py, ox*out_stride + out_px),  post(v) = v*scale[n] + shift[n]
 * for (oy, ox) in [0, OH) x [0, OW); out-of-image taps contribute zero.  Optional pointers may be NULL.
 * `in` (and res1 / res2) may themselves be channel slices of wider tensors only if dense; `out` may be a slice (out_ld).
 * wpk is W padded to a multiple of 32 rows of k = t*Cin + ci and packed [k/4][Cout][4]
 * (kind USC_CONV_HEAD: plain [k], Cout = 1, and the (tanh(.)+1)/2 of model.py:566 is applied).
 * Cin must be a power of two >= 4 or a multiple of 32, Cout a multiple of 32 (or 1 for the head), ntaps <= 16. */
#define USC_ACT_NONE  0
#define USC_ACT_RELU  1
#define USC_ACT_LEAKY 2            /* LeakyReLU(0.2) */
#define USC_ACT_LEAKY01 3          /* LeakyReLU(0.01): InPlaceABNSync of the human-parsing net */
#define USC_CONV_MFMA 0
#define USC_CONV_HEAD 1

typedef struct usc_conv_desc {
    const float* in;
    float*       out;
    const float* wpk;
    const float* bias;
    const float* res1;
    const float* scale;
    const float* shift;
    const float* res2;
    int IH, IW, Cin;
    int OH, OW;                    /* logical output grid of this launch */
    int OHf, OWf, Cout;            /* dimensions of the tensor `out` points to */
    int in_stride;
    int out_stride, out_py, out_px;
    int res1_shift;
    int act;
    int ntaps;                     /* taps per phase */
    int kind;
    signed char dy[16], dx[16];    /* nphase * ntaps entries */
    int nphase;                    /* 1, or 4 = the (py, px) output phases of ConvTranspose2d(k4, s2, p1) in one launch:
                                      phase p uses taps [p*ntaps, (p+1)*ntaps), weights wpk + p*wphase_stride and
                                      writes (2*oy + (p >> 1), 2*ox + (p & 1)); out_py / out_px are then ignored */
    int wphase_stride;             /* floats between the packed weights of consecutive phases */
    int ksplit;                    /* set by the library */
    int in_relu;                   /* 1: the layer computes conv(relu(in)) (pre-activation residual blocks, stm/model.py:25-30) */
    float* slab;                   /* optional scratch for split-K partial sums (small-M layers), may be NULL.  Its LAST 64 KB hold
                                      arrival counters (Winograd layers that split K over two workgroups): the library zeroes
                                      them when it plans such a layer, kernels leave them even; partial sums never reach them.
                                      Layers that may run concurrently must not share a slab. */
    size_t slab_bytes;
    int out_ld, out_c0;            /* 0 / 0: `out` is a dense [pixel][Cout] tensor; otherwise `out` has out_ld channels per pixel
                                      and this layer writes channels [out_c0, out_c0 + Cout) (concat buffers) */
    int batch;                     /* 0 / 1: one image.  > 1: `batch` images stacked along y in every tensor; OH and OHf count the
                                      rows of all of them, IH is per image. */
    int plan_batch;                /* 0 / 1 or the batch the tile shape and the K partition are planned for (independent of `batch`).
                                      Layers of the same plan_batch give bit for bit the same per-image results at any `batch`:
                                      a pipeline that runs a batch-4 network and, for single frames, a batch-1 network gives
                                      both plan_batch = 4. */
    const float* wino;             /* optional, 3x3 / stride 1 / pad 1 layers with Cin % 8 == 0 only: the Winograd F(2x2, 3x3) operand
                                      U = G g G^T of the same weights, floats [Cin/8][16][2][Cout][4] with cin = 8c + 4h + q
                                      (vmatting/net_hip.py pack_wino).  When given, the library may run the layer as 16 per-position
                                      products on transformed 4x4 patches (2.25x fewer multiplications; results equal to the
                                      direct form up to float rounding, ~2e-7 relative).  NULL: direct form.
                                      Four-phase layers (nphase = 4: ConvTranspose2d(k4, s2, p1)) with Cin % 8 == 0, no first
                                      residual and no affine: the Winograd F(3x3, 2x2) operands of the four phases instead,
                                      floats [4][Cin/8][16][2][Cout][4] (vmatting/net_hip.py pack_wino_t: G = [1 0; 1/2 1/2;
                                      1/2 -1/2; 0 -1], phase (py, px) = index 2 py + px). */
    float* out2;                   /* optional second output tensor (direct, single-phase layers without residuals / affine only): output
                                      channels [split_c, Cout) go to `out2` - a dense [pixel][Cout - split_c] tensor - with activation
                                      `act2`, channels [0, split_c) to `out` ([pixel][split_c] unless out_ld says otherwise) with `act`.
                                      Two convolutions that read the same input with the same geometry run as ONE launch this way (a
                                      ResNet stage's 3x3 / stride-2 conv and its AvgPool + 1x1 shortcut: unscreen/vmatting/model.py:119-157 -
                                      the shortcut's weights sit on four of the nine taps, zeros on the others).  NULL: one output. */
    int split_c, act2;             /* split_c: a multiple of 32 (of 64 when Cout % 64 == 0 tiles are planned), 0 < split_c < Cout */
} usc_conv_desc;

/* Tuning hook (tools/dev/tune_convs.py): force the tile width (1 | 2 x 32 columns), the row groups per workgroup (4 | 2 | 1
 * x 32 pixels), the grid K split and the K groups per workgroup of every layer planned from the calling thread; 0 = the
 * library's own plan (cost model). */
void usc_conv_force_plan(int nt, int rowgroups, int ksplit, int kgroups);
/* Tests / tuning: -1 = the library decides per layer (default), 0 = never use the Winograd form, 1 = use it wherever it applies. */
void usc_conv_force_winograd(int mode);
/* Tests / tuning: which Winograd FORM layers planned from the calling thread may take.  t3 (transposed F(3x3, 2x2) form of the
 * four-phase layers) and lin (blocks of 32 consecutive tiles instead of 8 x 4 rectangles): -1 = cost model (default), 0 = never,
 * 1 = wherever admissible; split (K halves of a unit on two workgroups): 0 = cost model (default), 1 = never, 2 = wherever
 * admissible.  Like the two hooks above this is per-thread planner state, not process state: the library reads no environment
 * variable, and a plan is a function of the layer shape and these arguments only. */
void usc_conv_force_winograd_form(int t3, int lin, int split);

/* The launch plan the library chooses for one layer (no device work): tile width in 32-column units, row groups and K groups
 * per workgroup, grid K split, workgroups and threads of the main launch, launches per layer (2 when the grid splits K),
 * winograd = 1 | 2 | 3 when the layer runs in the Winograd form (8 x 4 blocks | blocks of 32 consecutive tiles | transposed
 * form; then nt = rowgroups = 0 and ksplit = 2 means K halves on two workgroups of the SAME launch).
 * Used by the profiling tools to label kernel traces (tools/dev/layer_table.py). */
typedef struct usc_conv_plan_info { int nt, rowgroups, ksplit, kgroups, workgroups, threads, launches, winograd; } usc_conv_plan_info;
int usc_conv_plan_query(const usc_conv_desc* desc, usc_conv_plan_info* out);

/* sizeof(usc_conv_desc) as the library was built: bindings check their struct definition against it. */
size_t usc_conv_desc_bytes(void);

/* Enqueue `n` layers in order on `stream` (one kernel launch each, no host synchronisation). */
int usc_conv_sequence_f32(const usc_conv_desc* descs, int n, usc_stream_t stream);
const char* usc_conv_last_error(void);

/* The same layer list planned once and kept behind a handle; launched either as stream launches or as a pre-instantiated hipGraph:
 * one node per launch, chained in list order (a linear chain: executable
 * graphs with parallel branches fault in hipGraphLaunch on the ROCm runtime torch 2.10 bundles, see usc_conv.hip).  The
 * descriptors are copied; the buffers they point to must stay valid until usc_conv_graph_destroy.  A handle is instantiated once
 * per stream it is launched on (at that stream's first launch); launches of one handle must come from one host thread at a time,
 * and - its layers sharing their activation buffers - must not overlap on the device (launch it on one stream, or order the
 * streams). */
#define USC_CONV_LAUNCH_STREAM 0   /* default: the planned launches are enqueued one by one on the stream (no re-planning, ~3 us of host
                                    * time per layer; measured 1.49 ms per matting forward against 1.67 ms for the graph replay) */
#define USC_CONV_LAUNCH_GRAPH  1   /* one hipGraphLaunch per call: least host time (one call per forward) */
#define USC_CONV_LAUNCH_RUNS   2   /* as STREAM, but every run of consecutive same-shape Winograd layers (each reading the one before) goes
                                    * out as ONE launch: workgroups pull (layer, block) items from a counter and wait on per-block
                                    * arrival counters of the previous layer instead of a kernel boundary; bit-identical results */
int  usc_conv_graph_create(const usc_conv_desc* descs, int n, void** handle);
int  usc_conv_graph_set_mode(void* handle, int mode);
int  usc_conv_graph_launch(void* handle, usc_stream_t stream);
/* USC_CONV_LAUNCH_RUNS bookkeeping (blocks: reads device memory): how many runs the handle's layer list holds, how many layers they
 * cover, and how many bounded polls of an arrival counter have ever run out (0 unless a hand-over is broken: such a launch has
 * gone on with possibly stale inputs instead of hanging - callers that use the mode check this: HipMattingNet.forward() does after every
 * forward and raises).  The queue, sequence words and arrival counters of the mode belong to the HANDLE: at most one launch of a handle may
 * be in flight at a time in this mode (launches of one handle on one stream are ordered and therefore fine; two streams are not). */
int  usc_conv_graph_run_info(void* handle, int* nruns, int* layers_in_runs, unsigned* poll_timeouts);
void usc_conv_graph_destroy(void* handle);   /* the one entry of this group that blocks: waits for the device to drain (a launch
                                              * of the graph may still be queued), then destroys the graph, its execs and plans */

#ifdef __cplusplus
}
#endif
#endif
