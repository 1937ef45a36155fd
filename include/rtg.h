/* rtg.h — C ABI of librtg.so: the MI355X (gfx950) kernels behind the RetuneGAN train-step hot path.
 *
 * The reference (Kahsolt/TransTacoS-RetuneGAN) is 100 % Python and has no FFI: its "operator interface" for this
 * path is the set of torch library calls made by retunegan/models/{generator,discrminator,loss}.py, retunegan/audio.py and retunegan/train.py.
 * Each entry point below names the reference call sites it replaces (paths relative to /root/reference).
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C: raw device pointers (float32 unless said otherwise), ints, floats; no torch / C++ types.
 *   - every buffer (inputs, outputs, workspaces) is allocated and owned by the caller; kernels never allocate.
 *   - asynchronous launch on the caller's `stream` (a hipStream_t passed as void*); no host synchronisation,
 *     no global mutable state: safe under hipGraph capture and from several host threads on distinct streams.
 *   - return 0 on success, a negative RTG_E* code on bad arguments, or -(1000 + hipError_t) on a launch failure.
 *   - tensors are contiguous NCW: [batch][channel][time].
 */
#ifndef RTG_H
#define RTG_H

#ifdef __cplusplus
extern "C" {
#endif

#define RTG_ABI_VERSION 11

#define RTG_OK 0
#define RTG_EINVAL (-1)   /* inconsistent descriptor               */
#define RTG_ERANGE (-2)   /* shape outside what the kernels tile   */
#define RTG_ENULL  (-3)   /* required pointer is NULL              */

/* input transform applied while a tile is staged into LDS */
enum { RTG_PRE_NONE = 0, RTG_PRE_LRELU = 1, RTG_PRE_MUL_DLRELU = 2, RTG_PRE_MUL_DTANH = 3 };
/* output activation */
enum { RTG_ACT_NONE = 0, RTG_ACT_LRELU = 1, RTG_ACT_TANH = 2 };
/* packed-weight layouts produced by rtg_weights_pack (see RtgPackJob.mode) */
enum { RTG_PACK_FWD = 0, RTG_PACK_DGRAD_S1 = 1, RTG_PACK_DGRAD_POLY = 2, RTG_PACK_CONVT_POLY = 3, RTG_PACK_DGRAD_2D = 4,
       /* ABI 7: the plain layouts rtg_gconv_forward / rtg_gconv_backward_data read (what rtg_gconv_prepare[_bwd] write),
        * produced by rtg_weights_pack with the other images of a model instead of one launch per layer and pass:
        * [group][ci][tap][oc] resp. [group][oc][tap][ci], dst_size = groups * Mg * Cg * K floats */
       RTG_PACK_GCONV_FWD = 5, RTG_PACK_GCONV_BWD = 6,
       /* [group][oc][ci][44]: the taps of a (row, channel) pair padded to 44 with zeros — what rtg_gmfma_forward reads
        * (dst_size = groups * Mg * Cg * 44 floats; RtgPackJob.S = 44).  With RtgPackJob.KH = the layer's stride (the layer
        * of 8 output channels per group): the position-pair image [group][16][ci][48], row (r, oc) = w[oc][ci][u - KH * r]
        * (dst_size = groups * 16 * Cg * 48 floats; RtgPackJob.S = 48) */
       RTG_PACK_GMFMA_FWD = 7 };

/* ------------------------------------------------------------------------------------------------------------
 * rtg_conv1d — implicit-GEMM 1-D convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32 / 16x16x4_f32).
 *
 *   acc[b, g*Mg+m, q] = sum_{c<Cg, j<K} W[g][m][c][j] * pre( xcat[b, g*Cg+c, q*stride - pad + j*dil] )
 *   out               = act( ((acc + bias[m']) * dmask + res) * out_scale )
 *
 * Replaces: F.conv1d / nn.Conv1d forward (generator.py:41-56,139-141,682-722; discrminator.py:37-45), the
 * (k,1) Conv2d of DiscriminatorP with the period folded into the batch (discrminator.py:156-163), and - through
 * repacked weights (rtg_weights_pack) - ConvTranspose1d forward (generator.py:697-700, polyphase: M = C_out*S rows,
 * ceil(K/S) taps, "shuffle" store) and every convolution_backward w.r.t. the input (same kernel, RTG_PACK_DGRAD_*).
 *
 * xcat is the virtual concatenation along channels of x1 [B,C1,L_in] and x2 [B,C2,L_in] (torch.cat at
 * generator.py:755,769); x2 may be NULL with C2 = 0.
 * pre(): RTG_PRE_LRELU -> leaky_relu(x, pre_slope) (F.leaky_relu / nn.LeakyReLU call sites generator.py:40,147,745..);
 *        RTG_PRE_MUL_DLRELU -> x * (aux > 0 ? 1 : pre_slope); RTG_PRE_MUL_DTANH -> x * (1 - aux^2)  (aux shaped like x1).
 * dmask (optional, `mask` != NULL): (mask > 0 ? 1 : mask_slope), mask shaped like out (leaky-relu backward).
 * Shuffle store (shuf_S > 1): row m' = g*Mg+m writes channel m'/shuf_S at time q*shuf_S + m'%shuf_S - shuf_P;
 *        bias is then indexed by the channel m'/shuf_S.
 * `wp` is the packed weight buffer of this layer (RtgPackJob).  tile_m selects the MFMA shape (32 or 16 rows).
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct RtgConv1dDesc {
  int B, C1, C2, L_in;         /* input: C_in = C1 + C2                                                      */
  int groups, Cg, Mg;          /* channels per group (in), logical output rows per group                    */
  int K, stride, dil, pad;     /* taps, stride, dilation, left padding                                       */
  int Q;                       /* number of output positions computed                                        */
  int out_C, out_L;            /* shape of the stored tensor [B, out_C, out_L]                               */
  int shuf_S, shuf_P;          /* 1, 0 for a normal store                                                    */
  int pre_mode; float pre_slope;
  float mask_slope;
  float out_scale;
  int act; float act_slope;
  int accumulate;              /* out += result instead of out = result                                      */
  int tile_m;                  /* 32 or 16                                                                   */
  int out_split;               /* 0, or: channels >= out_split are stored to `out2` ([B, out_C-out_split, out_L]);
                                  either half may be skipped by passing NULL (backward of a torch.cat input pair) */
  /* Second dimension (Conv2d of StftDiscriminator, discrminator.py:255-262).  All zero (or h_k = h_n = 1) = 1-D.
   * The 2-D convolution over [items, C, H, W] runs as this 1-D operator along W: a "clip" (index 0..B-1) is one
   * (item, output row) pair, B = items * h_n; a "channel" (0..C1-1) is one (c, kernel row kh) pair, C1 = C * h_k;
   * L_in = W.  The patch row of clip (item, r) and channel (c, kh) is input row
   *     h_mode 0 (forward / weight gradient):  r * h_stride - h_pad + kh
   *     h_mode 1 (backward-data over dy):      (r + h_pad - kh) / h_stride   when divisible, else a zero row
   *     h_mode 2 (ABI 10: the forward again, with the channels ordered (kernel row kh, c) — channel index kh * C + c, the
   *              weight image packed with RtgPackJob.kh_major; served by the dense-layer block shapes only, C a multiple of
   *              the chunk (16; 32 with bf16 operands): rtg_conv1d_tile_candidates lists them or returns 0.  Same sums in
   *              another order: results differ from h_mode 0 by summation-order rounding)
   * of the [items, C, h_in, W] tensor; rows outside [0, h_in) are zero padding.  The output tensor is
   * [items, out_C, h_n, out_L].  Requires groups == 1, C2 == 0, out_split == 0. */
  int h_in, h_k, h_stride, h_pad, h_n, h_mode;
  int tap_major;               /* weights packed in the tap-major order (RtgPackJob.tap_major): for layers with few input
                                  channels per group (C_in = 1 first layers, grouped MSD convs) the MFMA K dimension
                                  walks (channel, 64/tile_m consecutive taps) instead of padding the channels to 16   */
  int tile_cfg;                /* block shape: 0 = the library's heuristic, else MT*100 + NT*10 + WM as listed by
                                  rtg_conv1d_tile_candidates (wave tile MT x NT MFMA tiles, WM x 4/WM waves).  Every
                                  shape gives bit-identical results; callers time the candidates once per layer      */
  int bf16;                    /* 1: operands rounded to bf16 (activations when they are staged, weights packed by
                                  rtg_weights_pack with RtgPackJob.bf16) and multiplied on the bf16 matrix cores
                                  (v_mfma_f32_32x32x8_bf16_1k / 16x16x16), fp32 accumulate, fp32 tensors in HBM
                                  (BASELINE configs[2]).  Not with tap_major.                                  */
  int wp16;                    /* 1 (ABI 6): `wp` carries a second image of the layer's weights right behind the standard
                                  one (at wp + rtg_packed_size(groups, Mg, Cg, K, tile_m) floats): the 16-byte-fragment
                                  image RtgPackJob.frag16 writes (rtg_packed_size_frag16 floats).  The dense-layer kernel
                                  of rtg_dconv.hip (block-shape codes 8xxx) is only listed / accepted with it.
                                  2: ... and the caller also accepts the split-K kernel of rtg_sconv.hip (codes 9004 / 9008:
                                  few columns, the reduction split over the waves of a block — a summation order of its
                                  own, where every other block shape gives identical bits)                             */
  int io_bf16;                 /* ABI 9: bf16 feature maps in HBM (BASELINE configs[2]) — a bit mask of RTG_IO_*: which of
                                  the tensor arguments are bf16 NCW instead of fp32 (the pointer parameters stay `float*`).
                                  RTG_IO_X_BF16: x1 holds bf16 values that are ALREADY activated — what a producer with
                                  RTG_IO_OUT_BF16 stored, or a gradient; the kernel applies no pre-activation to it
                                  (pre_mode / pre_slope then describe the tensor, not work to do).
                                  RTG_IO_OUT_BF16: out = bf16(leaky_relu(result, enc_slope)), rounded to nearest even — the
                                  consumer's leaky-relu applied once by the producer (enc_slope = 1: a plain bf16 store,
                                  used for gradients); the sign, all a leaky-relu backward needs, survives.
                                  RTG_IO_MASK_BF16 / RTG_IO_RES_BF16: `mask` / `res` are bf16.
                                  A bf16 x1 (and the bf16 x / dy of rtg_conv1d_wgrad) must be followed by 16 READABLE
                                  bytes: the rows are read with 16-byte loads at 2-byte granularity, a load that starts
                                  at the tensor's last elements runs past its end (what it reads there is discarded).
                                  Needs bf16 = 1 and a block shape of the dense-layer kernel (codes 8xxx):
                                  rtg_conv1d_tile_candidates lists only those for such a descriptor, possibly none —
                                  the caller then converts (rtg_bf16_decode / rtg_bf16_encode) around an fp32 launch.   */
  float enc_slope;
} RtgConv1dDesc;
enum { RTG_IO_X_BF16 = 1, RTG_IO_OUT_BF16 = 2, RTG_IO_MASK_BF16 = 4, RTG_IO_RES_BF16 = 8 };

int rtg_conv1d(const RtgConv1dDesc* d, const float* x1, const float* x2, const float* aux, const float* wp,
               const float* bias, const float* mask, const float* res, float* out, float* out2, void* stream);

/* Up to RTG_MAX_GROUP (4) independent convolutions in ONE launch: the sub-discriminators of a stack (3 MSD scales, 4 MPD
 * periods, 3 MTD resolutions; discrminator.py:104-129,225-244,311-330) run the same layer on different clips with
 * different weights; each alone often gives a CU less than one workgroup, side by side they fill the chip.  Every
 * descriptor must carry the same non-zero tile_cfg and the same tile_m (one kernel instance serves the group); results
 * are bit-identical to n separate rtg_conv1d calls. */
#define RTG_MAX_GROUP 4
typedef struct RtgConvPtrs {
  const float *x1, *x2, *aux, *wp, *bias, *mask, *res;
  float *out, *out2;
} RtgConvPtrs;
int rtg_conv1d_group(int n, const RtgConv1dDesc* descs, const RtgConvPtrs* ptrs, void* stream);

/* which template instantiation rtg_conv1d would launch for this descriptor: tile_m*100 + MT*10 + NT (wave tile =
 * MT x NT MFMA tiles), or a negative RTG_E* code.  Used by bench.py to attribute time per kernel. */
int rtg_conv1d_variant(const RtgConv1dDesc* d);
/* the block shapes (RtgConv1dDesc.tile_cfg codes) valid for this descriptor, best-guess first; returns how many were
 * written to cfgs[0..max) or a negative RTG_E* code */
int rtg_conv1d_tile_candidates(const RtgConv1dDesc* d, int* cfgs, int max);

/* number of floats of the packed weight buffer for a layer with the given logical shape */
long long rtg_packed_size(int groups, int Mg, int Cg, int K, int tile_m);
/* ... of the 16-byte-fragment image (RtgPackJob.frag16; groups == 1): ceil(Mg/16) x ceil(Cg/16) x K x 256 */
long long rtg_packed_size_frag16(int Mg, int Cg, int K);
/* ... of its bf16 form (RtgPackJob.frag16 with .bf16): ceil(Mg/16) x ceil(Cg/32) x K x 256 floats (8 bf16 per 16 bytes) */
long long rtg_packed_size_frag16_bf16(int Mg, int Cg, int K);
/* the same for the tap-major order, and whether that order needs fewer MFMAs than the channel-major one (1 / 0) */
long long rtg_packed_size_bf16(int groups, int Mg, int Cg, int K, int tile_m);   /* floats (2 bf16 each) */
long long rtg_packed_size_tapmajor(int groups, int Mg, int Cg, int K, int tile_m);
int rtg_tapmajor_pays(int Cg, int K, int tile_m);

/* ------------------------------------------------------------------------------------------------------------
 * rtg_conv1d_wgrad — convolution_backward w.r.t. weight and bias (train.py:158,191 -> autograd of every conv above).
 *
 *   dW[g*Mg+m][c][j] = sum_{b,q} gy(b, g*Mg+m, q) * pre( xcat[b, g*Cg+c, q*stride - pad + j*dil] )
 *   db[g*Mg+m]       = sum_{b,q} gy(b, g*Mg+m, q)
 *   gy = gy_scale * dy * dact(gy_aux)   (gy_mode RTG_PRE_MUL_DLRELU / MUL_DTANH, slope gy_slope),
 *        gy_scale * leaky_relu(dy)       (gy_mode RTG_PRE_LRELU: the transposed-conv case where the roles are swapped),
 *        gy_scale * dy                   (RTG_PRE_NONE)
 *
 * The (b,q) reduction is split `splits` ways; split s writes its partial to  part + s*part_stride  laid out as
 * [rows = groups*Mg][Cg*K] followed (at part + s*part_stride + rows*Cg*K) by the `rows` bias partials.
 * rtg_weightnorm_backward sums the partials in fixed order (bitwise reproducible).
 * For ConvTranspose1d the caller swaps the roles (x := dy, dy := x) and gets dW in the [C_in][C_out][K] layout.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct RtgWgradDesc {
  int B, C1, C2, L_in;
  int groups, Cg, Mg;
  int K, stride, dil, pad;
  int Q;                       /* positions of dy used (dy is [B, groups*Mg, dy_L])                          */
  int dy_L;
  int pre_mode; float pre_slope;
  int gy_mode; float gy_slope; float gy_scale;
  int splits; long long part_stride;
  /* second dimension, as in RtgConv1dDesc (h_mode is always 0 here): x is [items, C1/h_k, h_in, L_in], dy is
   * [items, groups*Mg, h_n, dy_L], B = items * h_n clips */
  int h_in, h_k, h_stride, h_pad, h_n;
  int shape_cfg;               /* block shape: 0 = the library's heuristic, else a code listed by
                                  rtg_wgrad_shape_candidates.  Codes 1-6 (the general kernel's block shapes) do not
                                  change the summation order (the number of splits does); 7 (one-channel layers), 8
                                  (UNet-G residual convs), 9 (thin-group layers on the vector ALUs) and 10 .. 14 (dense
                                  discriminator layers, rtg_dwgrad.hip: 1 / 2 / 4 / 8 channel chunks per block; 12 for
                                  3-tap or bf16 layers, 13 for 3-tap bf16 layers, 14 = 64-row blocks for 3-tap layers of 64 rows) are kernels of their own:
                                  rounding-level differences, each reproducible run to run                           */
  int bf16;                    /* 1: both operands rounded to bf16 as they are read from LDS, bf16 matrix cores, fp32
                                  accumulation and fp32 split partials (BASELINE configs[2])                         */
  int io_bf16;                 /* ABI 9: bf16 feature maps in HBM — bit 0 (RTG_IO_X_BF16): x1 is bf16 and already activated
                                  (no pre-activation is applied); bit 1 (RTG_IO_OUT_BF16): dy is bf16.  Needs bf16 = 1 and
                                  a shape of the dense-layer kernel (codes 10 .. 14): rtg_wgrad_shape_candidates lists only
                                  those for such a descriptor, possibly none (the caller then converts).  Partials stay fp32 */
} RtgWgradDesc;

int rtg_conv1d_wgrad(const RtgWgradDesc* d, const float* x1, const float* x2, const float* dy, const float* gy_aux,
                     float* part, void* stream);
/* ------------------------------------------------------------------------------------------------------------
 * Thin-group strided convolutions on the vector ALUs (rtg_gconv.hip): the grouped k41 layers of DiscriminatorS
 * (retunegan/models/discrminator.py:39-43: 4-8 input and 8-16 output channels per group, stride 2 / 4), whose groups
 * are too small for the matrix-core tiles.
 *   rtg_gconv_ok         1 when an instance serves the shape (K = 41; (Mg, Cg, stride) in (16, 8, 2), (16, 4, 4), (8, 8, 4))
 *   rtg_gconv_workspace  floats of the layer's weight buffer w
 *   rtg_gconv_prepare    w[group][ci][tap][oc] = v * scale from the weight-norm parameters (v [C_out][Cg][K], scale[r] =
 *                        g[r] / ||v[r]|| as written by rtg_weightnorm_scales); once per weight update
 *   rtg_gconv_forward    out = conv(leaky_relu(x, pre_slope)) + bias
 *   rtg_gconv_prepare_bwd / rtg_gconv_backward_data
 *                        dx = res + lrelu'(mask) * conv_transpose(dy) with w in [group][oc][tap][ci] order (mask / res
 *                        NULL: none);
 *                        d describes the FORWARD problem (L_in = length of dx, L_out = length of dy)
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct RtgGconvDesc {
  int B, groups, Cg, Mg, K, stride, pad, L_in, L_out;
  float pre_slope;             /* 1 = no input activation */
} RtgGconvDesc;
int rtg_gconv_ok(const RtgGconvDesc* d);
long long rtg_gconv_workspace(const RtgGconvDesc* d);
int rtg_gconv_prepare(const RtgGconvDesc* d, const float* v, const float* scale, float* w, void* stream);
int rtg_gconv_forward(const RtgGconvDesc* d, const float* x, const float* w, const float* bias, float* out, void* stream);
int rtg_gconv_prepare_bwd(const RtgGconvDesc* d, const float* v, const float* scale, float* w, void* stream);
int rtg_gconv_backward_data(const RtgGconvDesc* d, const float* dy, const float* w, const float* mask, const float* res,
                            float* dx, void* stream);
/* The same forward on the matrix cores with exact-fit tiles (rtg_gmfma.hip, ABI 7): a v_mfma_f32_16x16x4_f32 tile is one
 * group (16 output channels x 16 positions x 4 input channels at one tap).  w: the image RTG_PACK_GMFMA_FWD of
 * rtg_weights_pack ([group][oc][ci][44], 16-byte aligned; rtg_gmfma_workspace floats).  Same operator and descriptor as
 * rtg_gconv_forward (replaces F.conv1d(..., groups=g) of discrminator.py:39-43); rows of at least 16 positions.  The layer
 * with 8 output channels per group (discrminator.py:43) fills its tiles with PAIRS of neighbouring positions (rows =
 * (parity, channel), a stride-8 walk of 45 taps) and reads the pair image of RTG_PACK_GMFMA_FWD. */
int rtg_gmfma_ok(const RtgGconvDesc* d);
long long rtg_gmfma_workspace(const RtgGconvDesc* d);
int rtg_gmfma_forward(const RtgGconvDesc* d, const float* x, const float* w, const float* bias, float* out, void* stream);

/* n <= RTG_WGRAD_MAX_GROUP problems in ONE launch (the parallel ResBlock branches of a UNet-G decoder stage,
 * generator.py:776-778; the six convs of a ResidualStack, generator.py:33-77).  Every descriptor names the same general
 * block shape (shape_cfg 1..6, as listed by rtg_wgrad_shape_candidates) and its own splits / part_stride; the members
 * must share one kernel instance (fp32, 1-D, Mg >= 32, same tiling mode, patch width <= 128): RTG_EINVAL otherwise and
 * the caller launches them one by one.  Bit-identical to the members' own rtg_conv1d_wgrad launches. */
#define RTG_WGRAD_MAX_GROUP 6
typedef struct RtgWgradPtrs {
  const float *x1, *x2, *dy, *gy_aux;
  float* part;
} RtgWgradPtrs;
int rtg_conv1d_wgrad_group(int n, const RtgWgradDesc* descs, const RtgWgradPtrs* ptrs, void* stream);
/* suggested number of splits for a problem and its block shape (>= 1) */
int rtg_wgrad_splits(const RtgWgradDesc* d);
/* the block shapes (RtgWgradDesc.shape_cfg codes) valid for this problem, best-guess first; returns the count written */
int rtg_wgrad_shape_candidates(const RtgWgradDesc* d, int* cfgs, int max);

/* ------------------------------------------------------------------------------------------------------------
 * Weight bank: old-style weight norm (torch.nn.utils.weight_norm, dim=0: every conv of the path, e.g.
 * generator.py:682, discrminator.py:38) for ALL layers of a model in one launch each.
 *   scale[r] = g[r] / ||v[r,:]||,  w[r,:] = v[r,:] * scale[r]
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct RtgNormJob {      /* one per weight-normed tensor                                              */
  long long g_off, v_off;        /* offsets (floats) into the flat parameter buffer                           */
  long long scale_off;           /* offset into the scale buffer (rows floats) ; inv-norm stored at +rows     */
  int rows, inner;
} RtgNormJob;

typedef struct RtgPackJob {      /* one per packed layout of a layer                                           */
  long long v_off;               /* source v in the flat parameter buffer                                      */
  long long scale_off;           /* scale[] of the source tensor                                               */
  long long dst_off;             /* destination offset in the packed buffer                                    */
  long long dst_size;            /* = rtg_packed_size(groups, Mg, Cg, K, tile_m)                                */
  int mode;                      /* RTG_PACK_*                                                                 */
  int groups, Mg, Cg, K;         /* logical shape of the PACKED operator (rows per group, in-ch per group, taps) */
  int src_K, src_inner_c;        /* source taps; source dim-1 size (channels per group of the source tensor)  */
  int S;                         /* stride of the source conv (polyphase modes)                                */
  int tile_m;
  int KH;                        /* RTG_PACK_DGRAD_2D: kernel rows of the source [C_out][C_in][KH][src_K] weight;
                                    packed rows = (ci, phase), packed channels = (kh, co): kernel-row major                    */
  int tap_major;                 /* 1: [g][m-tile][k-step group][k-step][kk][m] with k-step = (channel, tap group)  */
  int bf16;                      /* 1: bf16 fragments [g][m-tile][chunk][tap][mfma][lane][4] (dst_size in floats =
                                    rtg_packed_size_bf16); lane (kk, m) holds channels 8*mfma + 4*kk .. +3 of the chunk
                                    (tile_m 32, two MFMAs per chunk) or 4*kk .. +3 (tile_m 16, one MFMA)             */
  int frag16;                    /* 1 (ABI 6; channel-major, groups == 1): the 16-byte-fragment image
                                    [16-row tile][chunk][tap][kgrp 4][m 16][kq 4] with channel = 4*kq + kgrp of the chunk
                                    (dst_size = rtg_packed_size_frag16): what rtg_dconv.hip (codes 8xxx) reads.  With bf16:
                                    [16-row tile][32-channel chunk][tap][kgrp 4][m 16][8 bf16], channel = 8*kgrp + element
                                    (dst_size = rtg_packed_size_frag16_bf16 floats)                                       */
  int first_block, n_blocks;     /* (ABI 6) the job's range of workgroups in the launch: n_blocks = rtg_pack_job_blocks(job),
                                    first_block = sum of n_blocks of the jobs before it in the table                     */
  int src_T;                     /* (ABI 9) 1: the layer runs on its input with the last two axes swapped (StftDiscriminator
                                    along the frequency axis): the operator's kernel rows are the source tensor's LAST kernel
                                    axis and its taps the one before — the source [C_out][C_in][src_K][KH] is read where the
                                    operator means [C_out][C_in][KH][src_K]                                              */
  int kh_major;                  /* (ABI 10; RTG_PACK_FWD with KH > 1, Cg = C_in * KH) 1: packed channel kh * C_in + ci instead
                                    of ci * KH + kh — the image RtgConv1dDesc.h_mode 2 reads                               */
} RtgPackJob;

typedef struct RtgWnBwdJob {     /* one per weight-normed tensor                                               */
  long long g_off, v_off, b_off; /* offsets into the flat parameter / gradient buffers (b_off < 0: no bias)    */
  long long scale_off;
  long long part_off;            /* offset into the partial buffer                                             */
  long long part_stride;
  int splits;
  int rows, inner;
  int t_rows, t_taps;            /* (ABI 9) both 0, or the kernel rows / taps of a layer packed with RtgPackJob.src_T: its
                                    partials are in the operator's order [C_in][t_rows][t_taps], the gradient is written in
                                    the tensor's [C_in][t_taps][t_rows]                                                  */
} RtgWnBwdJob;

int rtg_weightnorm_scales(const RtgNormJob* jobs_dev, int n_jobs, int max_rows, const float* params, float* scales,
                          void* stream);
/* workgroups job *job (HOST memory; first_block / n_blocks not read) needs in rtg_weights_pack; < 0: invalid job */
int rtg_pack_job_blocks(const RtgPackJob* job);
/* floats of LDS the job stages a slab of its image in (0: none); < 0: invalid job */
int rtg_pack_job_lds(const RtgPackJob* job);
/* total_blocks = sum of the jobs' n_blocks (the table is ordered by first_block); lds_floats = the largest rtg_pack_job_lds
 * of the jobs (a job whose slab does not fit takes the slower gather path) */
int rtg_weights_pack(const RtgPackJob* jobs_dev, int n_jobs, long long total_blocks, int lds_floats, const float* params,
                     const float* scales, float* packed, void* stream);
/* grads[g_off..], grads[v_off..], grads[b_off..] += weight-norm backward of (sum of partials) */
int rtg_weightnorm_backward(const RtgWnBwdJob* jobs_dev, int n_jobs, int max_rows, int max_inner, const float* params,
                            const float* scales, const float* partials, float* grads, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * rtg_stft_* — framed rFFT of get_stft_torch (retunegan/audio.py:150-170) + the epilogues of multi_stft_loss
 * (retunegan/models/loss.py:32-52): reflect pad n_fft/2, periodic hann(win) centred in n_fft, rFFT,
 * S = |D + 1e-9|, mel = melW @ S, outputs logS = log S and phase/PI  (PI = 3.14159265358979, utils.py:12).
 *   y [B,T] -> mel [B,n_mel,frames]; spec [B,2,F,frames] (channel 0 = log S, 1 = angle/PI) if spec != NULL;
 *   re/im [B,F,frames] saved for the backward if not NULL.
 * Backward: given dmel [B,n_mel,frames] (may be NULL) and dspec [B,2,F,frames] (may be NULL) -> dy [B,T] (+=).
 * melW is passed as CSR-like band tables: for filter m the non-zero bins are [lo[m], lo[m]+len[m]) with weights
 * at wts + woff[m].  twiddle: [n_fft/2] cos then [n_fft/2] sin of 2*pi*k/n_fft (fp64-rounded).  window: [win].
 * Sizes: n_fft a power of two in 128 .. 4096 (RTG_ERANGE otherwise; the reference's multi_stft_params use 2048 / 1024 /
 * 512, torch.stft itself takes any size), 1 <= win <= n_fft, n_fft/2 < T (reflect padding), n_mel <= 256.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct RtgStftDesc {
  int B, T, n_fft, win, hop, frames, n_mel;
  int spec_T;                    /* (ABI 9) 0: spec / dspec are [B][2][F][frames] (the reference's stack, loss.py:36-44);
                                    1: [B][2][frames][F] — frequency contiguous: coalesced stores, and the layout the
                                    spectrogram discriminators walk (callers hand out the transposed view)          */
} RtgStftDesc;

int rtg_stft_forward(const RtgStftDesc* d, const float* y, const float* window, const float* twiddle,
                     const int* mel_lo, const int* mel_len, const int* mel_woff, const float* mel_w,
                     float* mel, float* spec, float* re, float* im, void* stream);
int rtg_stft_backward(const RtgStftDesc* d, const float* re, const float* im, const float* dmel, const float* dspec,
                      const float* window, const float* twiddle,
                      const int* binmel_idx, const float* binmel_w,   /* per bin: 2 (filter, weight) pairs          */
                      float* frame_ws,                                 /* workspace [B*frames*win]                   */
                      float* dy, void* stream);

/* (ABI 11) Several (resolution, signal) jobs in ONE launch per kernel: multi_stft_loss (retunegan/models/loss.py:30-52) runs
 * hparam.multi_stft_params' three resolutions on the real and on the generated wave — six calls of get_stft_torch
 * (retunegan/audio.py:150-170), launch-shaped rather than byte-shaped on the GPU.  A job = the descriptor and the operands
 * of rtg_stft_forward / rtg_stft_backward.  The `jobs` array is HOST memory (copied into the launch's arguments).
 * rtg_stft_backward_multi: the jobs are the resolutions of ONE wave [B, T] (equal B, T or RTG_EINVAL); dy[b,t] =
 * (accumulate ? dy[b,t] : 0) + the jobs' contributions in job order — no zero-filled dy needed, no atomics. */
#define RTG_STFT_MAX_JOBS 8
typedef struct RtgStftFwdJob {
  RtgStftDesc d;
  const float *y, *window, *twiddle;
  const int *mel_lo, *mel_len, *mel_woff;
  const float* mel_w;
  float *mel, *spec, *re, *im;       /* mel, spec, (re, im) may be NULL as in rtg_stft_forward */
} RtgStftFwdJob;
typedef struct RtgStftBwdJob {
  RtgStftDesc d;
  const float *re, *im, *dmel, *dspec, *window, *twiddle;     /* dmel, dspec may be NULL */
  const int* binmel_idx;
  const float* binmel_w;
  float* frame_ws;                   /* [B * frames * win] scratch of this job */
} RtgStftBwdJob;
int rtg_stft_forward_multi(int n, const RtgStftFwdJob* jobs, void* stream);
int rtg_stft_backward_multi(int n, const RtgStftBwdJob* jobs, float* dy, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Small fused element-wise / reduction kernels of the path
 * ------------------------------------------------------------------------------------------------------------ */
/* GaussianNoise (generator.py:19-30): out = leaky_relu(x + u*w, slope), u ~ U[0,1) from a counter-based generator
 * keyed by (seed ^ f(*salt_dev), element index); salt_dev (optional) is a device word that changes every step (the
 * optimizer's step counter) so that a captured hipGraph draws fresh noise at every replay.
 * If `u_in` != NULL the noise is read from it instead (parity tests). */
int rtg_noise_lrelu_fwd(const float* x, const float* w, const float* u_in, float* out, long long n, float slope,
                        unsigned long long seed, const float* salt_dev, void* stream);
/* dx = dy * lrelu'(x + u*w);  dw_part[block] = sum dy * lrelu' * u   (n_blocks partials, summed by the caller) */
int rtg_noise_lrelu_bwd(const float* x, const float* w, const float* u_in, const float* dy, float* dx,
                        float* dw_part, int n_blocks, long long n, float slope, unsigned long long seed,
                        const float* salt_dev, void* stream);
/* (ABI 8) the same, followed by a one-block launch that adds the n_blocks partials in a fixed order and does
 * *dw_acc += sum — the gradient slot of the shared scalar `w` of GaussianNoise (generator.py:19-30); replaces part.sum()
 * and the gradient accumulation of the autograd engine */
int rtg_noise_lrelu_bwd_acc(const float* x, const float* w, const float* u_in, const float* dy, float* dx,
                            float* dw_part, int n_blocks, long long n, float slope, unsigned long long seed,
                            const float* salt_dev, float* dw_acc, void* stream);

/* out[c] += sum_{b,t} x[b,c,t]   (bias gradient of ConvTranspose1d layers; x is [B, C, L]); ws: 32 * C floats of scratch
 * (two fixed-order stages: partial sums over every 32nd clip, then their sum) */
int rtg_channel_sum(const float* x, float* out, int B, int C, int L, float* ws, void* stream);
/* out[i] (+)= alpha * a[i] + beta * b[i]   (b may be NULL) */
int rtg_axpby(const float* a, const float* b, float* out, long long n, float alpha, float beta, int accumulate,
              void* stream);
/* dx = dy * (ref > 0 ? 1 : slope) */
int rtg_lrelu_bwd(const float* dy, const float* ref, float* dx, long long n, float slope, void* stream);
/* (ABI 9) bf16 feature maps in HBM, the conversions at the edge of the kernels that read / write them natively
 * (RtgConv1dDesc.io_bf16):  rtg_bf16_encode: dst[i] = bf16(leaky_relu(src[i], slope)) rounded to nearest even (slope 1: a
 * plain conversion, gradients);  rtg_bf16_decode: dst[i] = v > 0 ? v : v / slope for v = float(src[i]).  `dst` / `src` of
 * the bf16 side are arrays of 2-byte elements */
int rtg_bf16_encode(const float* src, void* dst_bf16, long long n, float slope, void* stream);
int rtg_bf16_decode(const void* src_bf16, float* dst, long long n, float slope, void* stream);

/* nn.AvgPool1d(4, 2, 1) (discrminator.py:113) forward / backward on [rows, L] -> [rows, L/2] */
int rtg_avgpool4s2_fwd(const float* x, float* out, int rows, int L, void* stream);
int rtg_avgpool4s2_bwd(const float* dy, float* dx, int rows, int L, void* stream);

/* DiscriminatorP fold (discrminator.py:203-210): y [B,T] -> out [B*p, H] with out[b*p+w, h] = reflect_pad(y)[b, h*p+w] */
int rtg_period_fold_fwd(const float* y, float* out, int B, int T, int p, int H, void* stream);
int rtg_period_fold_bwd(const float* dout, float* dy, int B, int T, int p, int H, void* stream);

/* Scalar losses, multi-tensor: ONE launch covers a whole list of (a, b) pairs (all feature maps of a discriminator
 * stack, all logits, the three mel resolutions).   *loss_out += sum_j w_j * mean_j(term)   with
 *   RTG_LOSS_L1         |a - b|                         F.l1_loss                    (loss.py:154)
 *   RTG_LOSS_L1_L1LOG   |a - b| + |log a - log b|       mel + log-mel L1             (loss.py:51-52)
 *   RTG_LOSS_MSE_TARGET (target - a)^2                  LSGAN terms, equal-length rows (loss.py:121-122,142)
 *   RTG_LOSS_MSE_REL    (target - (a - b))^2            relative LSGAN terms, b detached: no gradient to b
 *                                                        (loss.py:116,136: hparam.relative_gan_loss)
 *   RTG_LOSS_L1_ENC     |dec(a) - dec(b)|               (ABI 9) F.l1_loss over bf16 feature maps as RTG_IO_OUT_BF16 stores
 *                                                        them: a, b are bf16 arrays of leaky_relu(x, s) values, dec(v) =
 *                                                        v > 0 ? v : v / s with s = the job's `target` field; da, db are
 *                                                        written as bf16 (gradients w.r.t. the decoded values)
 * Partials are summed in fixed order (ws needs 64 * n_jobs floats).  The backward writes (not accumulates)
 * d loss / d a into da and / or d loss / d b into db, scaled by w_j / n_j and by the device scalar *gscale (NULL = 1). */
#define RTG_MAX_LOSS_JOBS 48
enum { RTG_LOSS_L1 = 0, RTG_LOSS_L1_L1LOG = 1, RTG_LOSS_MSE_TARGET = 2, RTG_LOSS_MSE_REL = 3, RTG_LOSS_L1_ENC = 4 };
typedef struct RtgLossJob {
  const float* a; const float* b; float* da; float* db;
  long long n; float w; float target;
} RtgLossJob;
int rtg_loss_fwd(int kind, const RtgLossJob* jobs_host, int n_jobs, float* ws, float* loss_out, void* stream);
int rtg_loss_bwd(int kind, const RtgLossJob* jobs_host, int n_jobs, const float* gscale, void* stream);
/* dynamic_loss (loss.py:76-82): mean | |max_k(y)+max_k(-y)| - |max_k(g)+max_k(-g)| | over windows of k (=160) samples,
 * nn.MaxPool1d(k) semantics (stride k, tail dropped, first maximum wins).  ws: 256 floats.  Backward w.r.t. g. */
int rtg_dyn_loss_fwd(const float* y, const float* g, int rows, int L, int k, float w, float* ws, float* loss_out,
                     void* stream);
int rtg_dyn_loss_bwd(const float* y, const float* g, int rows, int L, int k, float w, const float* gscale, float* dg,
                     void* stream);
/* envelope_loss (loss.py:66-72): mean |max_k(y) - max_k(g)| + mean |max_k(-y) - max_k(-g)|, same windows and tie rule as
 * rtg_dyn_loss_*.  ws: 256 floats.  Backward w.r.t. g. */
int rtg_env_loss_fwd(const float* y, const float* g, int rows, int L, int k, float w, float* ws, float* loss_out,
                     void* stream);
int rtg_env_loss_bwd(const float* y, const float* g, int rows, int L, int k, float w, const float* gscale, float* dg,
                     void* stream);
/* strip_mirror_loss (loss.py:86-98) of y [rows, L] (an odd L drops the last sample): with u_i = y[2i] - y[2i+1] and
 * d_i = u_i - mean(u) (= the difference of the separately de-meaned even and odd strips, means taken over ALL rows),
 * loss = w * mean_i( -log(min(|d_i| + 1e-9, 1)) ).  stats: 4 device floats kept for the backward (sum u, sum f'(d), ...);
 * ws: 256 floats.  Backward w.r.t. y (writes every element of dy, the dropped tail sample gets 0). */
int rtg_strip_mirror_fwd(const float* y, int rows, int L, float w, float* ws, float* stats, float* loss_out, void* stream);
int rtg_strip_mirror_bwd(const float* y, int rows, int L, float w, const float* stats, const float* gscale, float* dy,
                         void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * rtg_adamw — torch.optim.AdamW step (train.py:80-81,160,193) over a flat fp32 parameter buffer:
 * decoupled weight decay, bias correction from the device-resident step counter state[0] (float),
 * skipped entirely (counter included) when *loss_flag is NaN (NaN guard of train.py:158,191 made device-side).
 * Hyper-parameters are doubles, as torch holds them: the scalar factors (1 - beta^t, lr / bc1, 1 - lr*wd) are formed in
 * double and the element arithmetic runs in fp32 in torch's order (m = lerp(m, g, 1-b1), ...).  grads are read as
 * grads[i] * grad_scale (1 / world size under data parallelism: the all-reduce sums).
 * ------------------------------------------------------------------------------------------------------------ */
int rtg_adamw(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n, float* step_state,
              const float* loss_flag, double lr, double beta1, double beta2, double eps, double weight_decay,
              float grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * rtg_resstack_* — a whole ResidualStack (generator.py:33-77: three times r = conv_d(lrelu(x)), x = x + conv_1(lrelu(r)),
 * k = 3, d = 1, 3, 9; optional leaky-relu on the final x) in ONE launch per direction, for the stacks whose clips fit in
 * LDS: (C, L) = (128, 32) and (64, 256).  Replaces the 6 Conv1d forward and 6 backward-data launches of a stack.
 *   forward   x [B,C,L]; wp / bias: the 6 layers' packed forward weights (RTG_PACK_FWD, tile_m 32) and biases in execution
 *             order; outs: the 6 results r1, x1, r2, x2, r3, y (y = the final x, activated when final_act)
 *   backward  dy [B,C,L] (cotangent of y), y (for the activation's derivative when final_act); wpb: the packed
 *             backward-data weights (RTG_PACK_DGRAD_S1) of the layers in REVERSE order; masks: the forward tensors whose
 *             sign masks each step: r3, x2, r2, x1, r1, x0; gouts: g_r3, g_x2, g_r2, g_x1, g_r1, dx0 — the first five are
 *             the output cotangents the weight-gradient launches of the layers need.
 * rtg_resstack_ok: the instance number (1: (C, L) = (128, 32), 2: (64, >= 64), 3: (32, >= 128)) that serves the shape, 0: none
 * (only instance 1 measured faster than six launches; a caller chooses which it uses: rtg/ops.py RTG_RESSTACK_KINDS).
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct RtgResStackDesc {
  int B, C, L;
  int dil[6];                  /* dilation (= padding) of the layers in execution order                          */
  float pre_slope;             /* leaky-relu slope of every conv input (0.01)                                       */
  int final_act; float act_slope;
} RtgResStackDesc;
int rtg_resstack_ok(const RtgResStackDesc* d);
int rtg_resstack_forward(const RtgResStackDesc* d, const float* x, const float* const* wp, const float* const* bias,
                         float* const* outs, void* stream);
int rtg_resstack_backward(const RtgResStackDesc* d, const float* dy, const float* y, const float* const* wpb,
                          const float* const* masks, float* const* gouts, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * HIP streams of the library's own (ABI 7).  torch.cuda.Stream() hands out streams of a pool of 32 per priority, round
 * robin — ProcessGroupNCCL draws its collective stream from the same pool, so a stream a caller creates for HIP-graph
 * capture (or forks inside one) can BE the RCCL stream: the process group's watchdog then polls an event whose stream is
 * capturing and hipEventQuery fails with hipErrorCapturedEvent (the watchdog ends the process).  Streams created here are
 * outside that pool; the host side wraps them in torch.cuda.ExternalStream (train.py, models/layers.py).  Replaces the
 * implicit stream pool behind the reference's single-stream step (retunegan/train.py:121-193 runs on the default stream).
 * priority: 0 normal, < 0 high (clamped to the device's range).  The stream stays alive until rtg_stream_destroy.
 * ------------------------------------------------------------------------------------------------------------ */
int rtg_stream_create(int priority, void** stream);
int rtg_stream_destroy(void* stream);
/* End whatever capture `stream` is the origin of and drop the graph: 0 = the stream was not capturing, 1 = a capture
 * (possibly invalidated by an illegal call) was ended.  The host side calls it when a HIP-graph capture raised, so that the
 * process can go on with eager launches (a stream left in an invalidated capture fails every later allocation with
 * hipErrorStreamCaptureImplicit). */
int rtg_stream_end_capture(void* stream);

/* library self-description */
/* (ABI 8) out[0] = sum_i w[i] * *p[i] over n <= RTG_MAX_SCALAR_TERMS device scalars, in list order (the loss totals of
 * retunegan/train.py:137-158,170-189), and the backward of that sum: out[i] = w[i] * g[0].  The term table is read on the
 * host and travels by value (legal under stream capture). */
#define RTG_MAX_SCALAR_TERMS 16
typedef struct RtgScalarTerms {
  const float* p[RTG_MAX_SCALAR_TERMS];
  float w[RTG_MAX_SCALAR_TERMS];
  int n;
} RtgScalarTerms;
int rtg_scalar_wsum(const RtgScalarTerms* terms, float* out, void* stream);
int rtg_scalar_fanout(const RtgScalarTerms* terms, const float* g, float* out, void* stream);

int rtg_abi_version(void);
const char* rtg_build_info(void);

#ifdef __cplusplus
}
#endif
#endif /* RTG_H */
