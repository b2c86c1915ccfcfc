/*
 * ssde.h -- C ABI of libssde_hip.so: the MI355X (gfx950) native hot path of
 * yang-song/score_sde_pytorch (NCSN++/DDPM++ U-Net forward, PC-sampler update,
 * DSM-training glue).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer borrowed from the caller (PyTorch owns
 *     the memory); the library never allocates or frees tensor storage;
 *   - all activations are NHWC fp32, weights are pre-packed by the host (see
 *     "weight packing" below); no torch types appear in any signature;
 *   - every launch function takes the hipStream_t to enqueue on (as void*) and
 *     returns 0 on success, a negative code on failure; ssde_last_error()
 *     returns a thread-local message for the last failure;
 *   - no global mutable state except graph handles created by ssde_graph_*.
 *
 * Reference interfaces each entry point replaces are cited as file:line into
 * yang-song/score_sde_pytorch.
 */
#ifndef SSDE_H_
#define SSDE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSDE_ABI_VERSION 10  /* (still 10, no layout change: SSDE_CONVF_X6_WIDE / SSDE_CONVF_X6_NO_WIDE -- the 128 x 256 tile of the bf16x6 GEMM -- are new routing flags old callers never set) 10: ssde_conv_args.gn_in_part0 / gn_in_part1 / gn_in_slices0 / gn_in_slices1 / gn_in_eps (the consuming launch merges the GroupNorm partials of its main source itself: no ssde_gn_finalize launch in front of it), ssde_gn_finalize merges with teams of 16 lanes, ssde_attn_args.flags (SSDE_ATTNF_BF16X6); 9: SSDE_CONVF_NO_SMALL_COUT (3x3 convolutions onto at most four channels have their own kernel, conv_small.hip), the register-fed F(4x4,3x3) matrix kernel splits its reduction (no interface change); 8: routing switches moved from environment variables into ssde_conv_args.flags / ssde_wgrad_args.flags / ssde_gn_bwd_reduce_args.flags, SSDE_TILE_WINOGRAD4R + SSDE_PACK_WINO4R (register-fed F(4x4,3x3) matrix kernel), SSDE_TILE_WINOGRAD4X removed; 7: ssde_gn_bwd_reduce_args.g0 / g1 (GroupNorm backward in one pass over dp and x); 6: ssde_conv_args.wino_v / ssde_wgrad_args.v_pre (forward by-product feeds the weight gradient); 5: SSDE_PACK_WINO4 image re-ordered per wave (plan blobs of version 4 carry the old image), ODE ops in programs */

/* ---- prologue applied to a source tensor while it is staged into LDS ---- */
enum {
  SSDE_PRO_NONE = 0,    /* x                                            */
  SSDE_PRO_GN = 1,      /* GroupNorm(x)            (AttnBlockpp, layerspp.py:77)   */
  SSDE_PRO_GN_SILU = 2, /* SiLU(GroupNorm(x))      (ResnetBlockBigGANpp, layerspp.py:243,264) */
  SSDE_PRO_SILU = 3     /* SiLU(x)                 (Dense_0(act(temb)), layerspp.py:263) */
};

/* A source operand: the channel-concatenation of up to two NHWC tensors
 * (torch.cat([h, hs.pop()], dim=1), ncsnpp.py:318, without materialising it)
 * plus an optional fused GroupNorm/SiLU prologue. */
typedef struct ssde_src {
  const float* p0;       /* [N, H, W, c0]                                  */
  const float* p1;       /* [N, H, W, c1] or NULL                          */
  int32_t c0, c1;
  int32_t pro_mode;      /* SSDE_PRO_*                                     */
  int32_t gn_groups;     /* G; channels-per-group = (c0+c1)/G              */
  const float* gn_mean;  /* [N, G]  from ssde_groupnorm_stats              */
  const float* gn_rstd;  /* [N, G]                                         */
  const float* gn_gamma; /* [c0+c1]                                        */
  const float* gn_beta;  /* [c0+c1]                                        */
  /* train-mode dropout applied AFTER the prologue (Dropout_0(act(GroupNorm_1(h))), layerspp.py:264-265):
   * element e of the virtual concat tensor is zeroed when hash32(e * 0x9E3779B1 + (*drop_seed ^ drop_salt))
   * < drop_thresh, kept and multiplied by drop_scale = 1/(1-p) otherwise.  The mask is a pure function of
   * (seed word, salt, element index), so forward, weight-gradient and input-gradient kernels regenerate it
   * instead of storing it.  drop_thresh == 0 disables it. */
  uint32_t drop_thresh;
  float drop_scale;
  const uint32_t* drop_seed;  /* device word, rewritten by the host / a step kernel every training step */
  uint32_t drop_salt;
  int32_t _pad;
} ssde_src;

/* ---- convolution / pointwise GEMM -------------------------------------------
 * out = scale * ( conv_kxk(pro(main)) + conv_1x1(pro(aux)) + bias + chan_add + resid )
 * replaces: nn.Conv2d 3x3/1x1 via ddpm_conv3x3 / ddpm_conv1x1 (models/layers.py:100-124),
 * NIN (models/layers.py:546-555), nn.Linear (ncsnpp.py:86-91, layerspp.py:227),
 * the strided conv of conv_downsample_2d (models/up_or_down_sampling.py:178) and
 * the residual tail "(x + h) / sqrt(2)" (layerspp.py:268-274, 87-91).
 * Weight packing (host side): w_main [ceil(Cin/8)][k*k][cout_pad][8],
 * w_aux [ceil(Cx/8)][cout_pad][8], cout_pad = roundup(c_out, 64), zero filled.
 * Under SSDE_TILE_AUTO the library routes by shape: 1x1-only launches to the GEMM kernels (conv1x1.hip), 3x3 / stride 1 /
 * pad 1 launches onto at most four channels -- the image heads, ncsnpp.py:329-337,368-375 -- to conv_small.hip (ABI 9; same
 * weight packing; ssde_conv_gn_slices is 0 for them), everything else to the direct matrix kernel (conv_mfma.hip); the
 * Winograd kernels are taken by naming their tile (their weights are packed differently). */
typedef struct ssde_conv_args {
  ssde_src main;         /* k x k source; ksize == 0 -> unused             */
  ssde_src aux;          /* 1 x 1 source at OUTPUT resolution; p0 == NULL -> unused */
  const float* w_main;
  const float* w_aux;
  int32_t n, h_in, w_in; /* spatial size of main                           */
  int32_t h_out, w_out, c_out;
  int32_t ksize;         /* 0 or 3                                         */
  int32_t stride;        /* 1 or 2                                         */
  int32_t pad;           /* zero padding of main (applied after prologue)  */
  int32_t tile;          /* SSDE_TILE_*; 0 = let the library choose        */
  const float* bias;     /* [c_out] or NULL                                */
  const float* chan_add; /* [N, chan_add_ld] per-(sample, channel) addend (Dense_0(act(temb))) or NULL */
  int32_t chan_add_ld;
  int32_t resid_post;    /* 0: out = scale*(... + resid) (residual tail); 1: out = scale*(...) + resid (gradient accumulation) */
  const float* resid;    /* [N, h_out, w_out, c_out] or NULL (may alias dst) */
  float out_scale;       /* 1 or 1/sqrt(2)                                 */
  uint32_t flags;        /* SSDE_CONVF_* routing switches (ABI 8; 0 = the library's own choice everywhere).  They replace the
                          * environment variables rounds 2-4 read inside the launchers: the route of a launch is part of its
                          * arguments (and of a plan blob), not of the process */
  float* dst;            /* [N, h_out, w_out, c_out]                       */
  float* gn_part;        /* optional: GroupNorm partial statistics of dst, [N][S][c_out/4][3] = (mean, M2, count) per
                          * (image, slice, channel quad), written by the epilogue; S = ssde_conv_gn_slices(args) > 0
                          * (every workgroup tile lies inside one image).  Merged by ssde_gn_finalize. */
  float* wino_v;         /* optional, SSDE_TILE_WINOGRAD4 only (ABI 6): the kernel also leaves the transformed input
                          * V[pos][(c0+c1)/4][t][4] = B^T pro(main) B (36 positions, channel quads, t = ((image * H/4) + tile
                          * row) * W/4 + tile column, fp32; 36 * N*H*W/16 * (c0+c1) floats, < 4 GB) -- what the F(4x4,3x3)
                          * weight gradient of the same layer multiplies with
                          * (ssde_wgrad_args.v_pre); a by-product of the first 64-cout tile's staging.  NULL: not written.
                          * SSDE_TILE_WINOGRAD4R: REQUIRED -- the same tensor in the same layout, written by the launch's
                          * transform pass and read by its matrix kernel (and still good for v_pre afterwards). */
  /* ABI 10 -- optional: main.gn_mean / main.gn_rstd are NOT filled yet; the launch merges them from the partials the producers
   * of main.p0 / main.p1 wrote (their ssde_conv_args.gn_part, [N][slices][c/4][3]) exactly as ssde_gn_finalize(part0 =
   * gn_in_part0, part1 = gn_in_part1, c0 = main.c0, c1 = main.c1, groups = main.gn_groups, eps = gn_in_eps) would -- bit for
   * bit -- and also leaves them in main.gn_mean / main.gn_rstd (a training program's backward reads them).  With
   * SSDE_TILE_WINOGRAD4R the transform pass does it for the (image, group) pairs of each workgroup while its pixel loads are
   * in flight (95 finalize launches of 6-9 us per U-Net evaluation were 3.5 % of it); every other route issues the
   * ssde_gn_finalize launch itself, in front of the kernel.  nn.GroupNorm: layerspp.py:67,219,231. */
  const float* gn_in_part0;   /* NULL: main.gn_mean / gn_rstd are valid as they are */
  const float* gn_in_part1;   /* main.c1 > 0 */
  int32_t gn_in_slices0, gn_in_slices1;
  float gn_in_eps;
  int32_t _pad_gn_in;
} ssde_conv_args;

enum { SSDE_CONVF_V_GIVEN = 1u,      /* SSDE_TILE_WINOGRAD4R: wino_v already holds B^T pro(main) B -- skip the transform pass */
       SSDE_CONVF_BF16X6 = 2u,       /* contractions that have a split kernel (1x1 / NIN / Linear GEMMs) run on the BF16 matrix pipe as
                                      * exact-fp32 products of a 3-way bf16 split, fp32 accumulation (conv1x1.hip) */
       SSDE_CONVF_NO_KSPLIT = 4u,    /* never split the reduction of a launch over several workgroups */
       SSDE_CONVF_BKC8 = 8u,         /* direct kernel: 8-channel stages everywhere (default: 64-channel stages where they fit) */
       SSDE_CONVF_GEMM_PIPE = 16u,   /* 1x1 GEMM: force the persistent pipelined kernel (default: where it pays, bf16x6 only) */
       SSDE_CONVF_NO_GEMM_PIPE = 32u,/* 1x1 GEMM: never take it */
       SSDE_CONVF_X6_BM64 = 64u,     /* bf16x6 GEMM: 64 rows per workgroup instead of 128 */
       SSDE_CONVF_X6_PF2 = 128u,     /* bf16x6 GEMM: rows loaded two stages ahead instead of one */
       SSDE_CONVF_NO_SMALL_COUT = 256u,/* 3x3 / stride 1 convolutions with at most four output channels (the image heads) on the general
                                          direct kernel instead of conv_small.hip (A/B runs, tests) */
       SSDE_CONVF_X6_WIDE = 512u,    /* bf16x6 GEMM: force the 128 x 256 tile wherever c_out % 256 == 0 (default: where its workgroups fill the device) */
       SSDE_CONVF_X6_NO_WIDE = 1024u };/* bf16x6 GEMM: never take it */
enum { SSDE_TILE_AUTO = 0, SSDE_TILE_256x64 = 1, SSDE_TILE_128x64 = 2, SSDE_TILE_64x64 = 3, SSDE_TILE_256x32 = 4,
       /* Winograd F(2x2,3x3) kernel (3x3, stride 1, pad 1, even output, no aux): w_main must then be packed as
        * [ceil(Cin/8)][ceil(Cout/64)][16 positions][4 channel pairs][64 couts, bit 4 ^= pair parity][2], G g G^T */
       SSDE_TILE_WINOGRAD = 5,
       /* Winograd F(4x4,3x3) kernel (3x3, stride 1, pad 1, output a multiple of 4, no aux): w_main packed as
        * [ceil(Cin/4)][ceil(Cout/64)][36 positions][64 couts][4 channels], G g G^T with the 6x3 G of F(4,3) */
       SSDE_TILE_WINOGRAD4 = 6,
       /* 7 was SSDE_TILE_WINOGRAD4X (F(4x4,3x3) on the BF16 matrix pipe through a 3-way bf16 split): parity-green but 6-24 % slower
        * than the fp32 kernel in both structures built; removed from the library in ABI 8, sources + logs under
        * tools/experiments/conv_wino4x/ */
       /* 8 was SSDE_TILE_WINOGRAD4G (two-kernel F(4x4,3x3) whose matrix kernel took both operands through LDS by LDS-DMA,
        * round 4): superseded by SSDE_TILE_WINOGRAD4R, which reproduces it bit for bit 7-14 % faster; removed in ABI 8,
        * tools/experiments/conv_wino4g_lds_fed/ */
       /* F(4x4,3x3) in two kernels: the input transform V = B^T pro(x) B as its own HBM-bound pass (wino4_xform.hip) into
        * ssde_conv_args.wino_v (REQUIRED here: 36 * N*H*W/16 * Cin floats, Cin % 8 == 0), then a matrix kernel without prologue or
        * transform that is fed from REGISTERS (conv_wino4r.hip, ABI 8): no LDS and no barrier in its main loop, every wave loads
        * the V runs and its private weights straight into the MFMA operands, two stages ahead.  Pays where a V tile feeds four or
        * more 64-cout tiles, and in training programs (V is what the weight gradient wants).  w_main packed per lane
        * (SSDE_PACK_WINO4R):
        * [ceil(Cin/4)][ceil(Cout/64)][8 waves (q, h)][4 pieces x [64 lanes][4] | [64 lanes][2]] where lane (lh, li) of wave (q, h)
        * holds, of cout 32 h + li and channels 2 lh, 2 lh + 1, the positions q + 4 (2 i), q + 4 (2 i + 1) in piece i and q + 32 last */
       SSDE_TILE_WINOGRAD4R = 9 };

/* ---- GroupNorm statistics: mean / rstd per (sample, group) -----------------
 * replaces the reduction half of nn.GroupNorm(min(C/4,32), C, eps=1e-6)
 * (layerspp.py:67,219,231; ncsnpp.py:194-227); the normalise+affine(+SiLU) half is
 * fused into the consumer's prologue. */
typedef struct ssde_gn_stats_args {
  const float* p0; const float* p1; int32_t c0, c1;   /* virtual concat */
  int32_t n, hw, groups; float eps;
  float* mean; float* rstd;      /* [N, G] */
  float* scratch;                /* >= N*slices*G*2 floats when slices > 1 */
  int32_t slices; int32_t _pad0;
} ssde_gn_stats_args;

/* GroupNorm statistics from the PRODUCERS' partials instead of a pass over the tensor: merges, in a fixed order,
 * the (mean, M2, count) triples the convolution epilogues wrote (ssde_conv_args.gn_part) over the slices and the
 * channel quads of every group of a (possibly concatenated) tensor. */
typedef struct ssde_gn_finalize_args {
  const float* part0; const float* part1;   /* [N][slices][c/4][3]; part1 NULL when not a concat */
  int32_t c0, c1, slices0, slices1;
  int32_t n, groups; float eps; int32_t _pad0;
  float* mean; float* rstd;                 /* [N, G] */
} ssde_gn_finalize_args;

/* ---- upfirdn2d (NHWC): zero-insert up, pad, FIR, decimate ------------------
 * replaces op/upfirdn2d.py:145-200 + op/upfirdn2d_kernel.cu:107-207 as used by
 * upsample_2d / downsample_2d / conv_downsample_2d (up_or_down_sampling.py:144-257)
 * and naive_upsample_2d / naive_downsample_2d (:59-69, as 2x2 box kernels). */
typedef struct ssde_upfirdn_args {
  ssde_src src;                  /* p1 must be NULL; prologue allowed */
  int32_t n, h_in, w_in, c;
  int32_t h_out, w_out;
  int32_t up, down, pad0, pad1;
  int32_t kh, kw;                /* <= 4 */
  float k[16];                   /* row-major [kh][kw], UNflipped (the op flips, as upfirdn2d does) */
  float* dst;                    /* [N, h_out, w_out, c] */
  int32_t accumulate;            /* dst += result (gradient accumulation in backward programs) */
  int32_t _pad0;
  float* dst2;                   /* optional second output: the same filter applied to the source WITHOUT its prologue
                                    (a residual block resamples act(GroupNorm(x)) and x, layerspp.py:250-258) */
} ssde_upfirdn_args;

/* ---- single-head self-attention core ----------------------------------------
 * replaces layerspp.py:82-86: w = softmax(q.k * C^-1/2) over keys; h = w.v
 * qkv: [N, L, 3C] (q | k | v per token, output of the fused NIN_0..2 GEMM). */
typedef struct ssde_attn_args {
  const float* qkv; float* dst;  /* dst [N, L, C] */
  int32_t n, l, c; float scale;
  uint32_t flags;                /* SSDE_ATTNF_* (ABI 10; 0 = the fp32-MFMA kernel) */
  int32_t _pad0;
} ssde_attn_args;
enum { SSDE_ATTNF_BF16X6 = 1u }; /* both contractions (Q K^T and P V) on the BF16 matrix pipe as exact-fp32 products of a 3-way bf16
                                  * split, fp32 accumulation and an fp32 softmax (attention.hip: attn_x6_kernel) -- the forward at
                                  * L = 256 tokens and C <= 256 channels; other shapes and the backward stay on the fp32 kernels */

/* ---- time / noise-level embeddings -------------------------------------------
 * kind 0: GaussianFourierProjection(log(cond)) (layerspp.py:39-41, ncsnpp.py:239)
 * kind 1: get_timestep_embedding(cond, dim)     (layers.py:515-529)              */
typedef struct ssde_embed_args {
  const float* cond;             /* [N] sigma (kind 0) or timestep label (kind 1) */
  const float* w;                /* kind 0: [dim/2] Fourier frequencies */
  float* dst;                    /* [N, dim] */
  int32_t n, dim, kind, _pad0;
} ssde_embed_args;

/* ---- layout boundary (reference tensors are NCHW, ncsnpp.py:232) ------------ */
typedef struct ssde_to_nhwc_args {   /* dst[n,h,w,c] = a*f(n)*src[n,c,h,w]+b, channels c..c_pad-1 zero */
  const float* src; float* dst; int32_t n, c, h, w, c_pad; float a, b;
  int32_t mode;                  /* f(n) as for ssde_to_nchw: backward of the output head */
  const float* v;                /* [N] or NULL */
} ssde_to_nhwc_args;
typedef struct ssde_to_nchw_args {   /* dst[n,c,h,w] = alpha * f(n) * src[n,h,w,c] */
  const float* src; float* dst; int32_t n, c, h, w, c_src;
  int32_t mode;                  /* 0: f=1; 1: f=1/v[n] (scale_by_sigma, ncsnpp.py:377-379); 2: f=-1/v[n] (VP score, models/utils.py:159) */
  const float* v;                /* [N] */
  float alpha;                   /* 0 is read as 1 (forward programs leave it unset) */
  int32_t accumulate;            /* dst += ... */
} ssde_to_nchw_args;

/* ---- fused bias + activation (API parity with op/fused_act.py:86-97) -------- */
typedef struct ssde_bias_act_args {
  const float* src; const float* bias; float* dst;
  int64_t numel; int32_t channels; int32_t inner;   /* bias index = (i / inner) % channels */
  int32_t act;                    /* 1 linear, 3 leaky relu */
  float alpha, scale;
  int32_t grad;                   /* 0: forward; 1: first-order gradient mode, dst = src * (ref > 0 ? 1 : alpha) * scale
                                     (op/fused_bias_act_kernel.cu:36-44, used by FusedLeakyReLUFunctionBackward) */
  const float* ref;               /* grad == 1: the forward OUTPUT whose sign selects the slope */
} ssde_bias_act_args;

/* ---- predictor-corrector update (sampling.py:195-200, 262-282, 181-187) ----- */
typedef struct ssde_sumsq_args {  /* out_a[n] = sum(a[n,:]^2), out_b likewise */
  const float* a; const float* b; float* out_a; float* out_b; int32_t n, per;
} ssde_sumsq_args;
typedef struct ssde_randn_args {  /* Philox4x32-10 (rocRAND device API) standard normals */
  float* dst; int64_t numel; uint64_t seed; const int32_t* step_ptr; int32_t stream_id; int32_t _pad0;
  const uint64_t* seed_ptr;  /* optional device word ADDED to `seed` at run time: one captured graph serves every seed
                                (the reference draws fresh torch.randn_like noise on every call, sampling.py:197,275) */
} ssde_randn_args;
typedef struct ssde_langevin_args {
  /* step = (snr * mean_n(||z||) / mean_n(||g||))^2 * 2 * alpha ; x_mean = x + step*g ; x = x_mean + sqrt(2*step)*z */
  float* x; float* x_mean; const float* grad; const float* noise;
  const float* grad_sumsq; const float* noise_sumsq;  /* [N] from ssde_sumsq */
  const float* alpha_tab; const int32_t* step_ptr;    /* alpha = alpha_tab ? alpha_tab[*step_ptr] : 1 */
  int32_t n, per; float snr; int32_t _pad0;
} ssde_langevin_args;
typedef struct ssde_predictor_args {
  /* x_mean = a*x + b*score ; x = x_mean + c*z   with (a,b,c) = coef[3*(*step_ptr) .. +2]
   * reverse diffusion VE: a=1, b=G^2, c=G (sde_lib.py:246-254, sampling.py:195-200) */
  float* x; float* x_mean; const float* score; const float* noise;
  const float* coef; const int32_t* step_ptr;
  int64_t numel;
} ssde_predictor_args;
typedef struct ssde_sample_update_args {
  /* One predictor / corrector update with PER-SAMPLE coefficients (the generic path of sampling.py, where `t` may differ per
   * sample and the score comes from any model):   x_mean = a[n] x + b[n] y ;   x_out = x_mean + c[n] z
   * fp32, one rounding per product and per sum (the reference's separate torch multiply / add kernels).  a == NULL: 1;
   * y == NULL: no second term; z == NULL: x_out = x_mean.  Covers Euler-Maruyama (sampling.py:181-187: y = drift, b = dt,
   * c = g sqrt(-dt)), reverse diffusion (:195-200: y = f, b = -1, c = G), ancestral sampling (:213-239), Langevin and
   * annealed Langevin steps (:262-282, :300-319: y = score, b = step size, c = sqrt(2 step size)). */
  const float* x; const float* y; const float* z; const float* a; const float* b; const float* c;
  float* x_mean; float* x_out; int32_t n; int32_t per;
} ssde_sample_update_args;
typedef struct ssde_fill_args {   /* dst[i] = tab[*step_ptr] (vec_t / labels of the current step, sampling.py:405) */
  float* dst; const float* tab; const int32_t* step_ptr; int32_t n; int32_t _pad0;
} ssde_fill_args;
typedef struct ssde_step_inc_args { int32_t* step_ptr; int32_t delta; int32_t _pad0; } ssde_step_inc_args;
typedef struct ssde_project_args {
  /* data-consistency projection of controllable generation, applied in place to the NCHW sampler state after a
   * corrector / predictor update (controllable_generation.py:44-52 inpainting, :136-144 colorization):
   *   y      = A x                                   A = identity, or the colour decoupling M when use_matrix
   *   known  = m D + s z                             (m, s) = coef[2*(*step_ptr) .. +1]: sde.marginal_prob's mean
   *                                                  coefficient and std at this step; D = data in A's space
   *   x      = A^-1 (y (1 - mask) + known mask)
   *   x_mean = A^-1 ((A x)(1 - mask) + m D mask)     with the NEW x, as the reference computes it */
  float* x; float* x_mean; const float* data; const float* mask; const float* noise;
  const float* coef; const int32_t* step_ptr;
  int32_t n, c, hw;               /* state is [n, c, hw]; use_matrix needs c == 3 */
  int32_t use_matrix;
  float M[9], invM[9];            /* row-major [i][j]: y_j = sum_i x_i M[i][j]  (einsum 'bihw,ij->bjhw', :108-113) */
} ssde_project_args;


/* =============================== training path ====================================
 * Backward of the U-Net program and the DSM training step (losses.py:73-99,177-208,
 * models/ema.py:32-51).  Gradients of parameters are written in the REFERENCE layouts
 * (OIHW conv weights, [out,in] Linear, [in,out] NIN) straight into a flat gradient
 * buffer, so a reference optimizer / checkpoint sees the usual tensors.
 * The input-gradient of a convolution is the forward kernel itself (ssde_conv2d) run on
 * the output gradient with host-repacked (transposed, 180-degree rotated) weights. */

/* ---- weight gradient: dw += scale * sum_pixels g[pix, co] * pro(src)[pix (+) tap, ci] -------- */
typedef struct ssde_wgrad_args {
  ssde_src src;            /* the convolution's input operand incl. its prologue (recomputed, not stored) */
  const float* g;          /* gradient of the convolution output, rows = output pixels [N*h_out*w_out] */
  int32_t g_ld, g_off;     /* row stride of g and first column used */
  int32_t n, h_in, w_in, h_out, w_out;
  int32_t c_out;           /* number of g columns (output channels of this weight) */
  int32_t ksize;           /* 3 or 1 */
  int32_t stride, pad;
  int32_t cin_store;       /* input channels that exist in dw (<= c0+c1: padded channels are skipped) */
  int32_t transpose_out;   /* ksize 1 only: dw is [cin_store][c_out] (NIN.W, models/layers.py:550) */
  int32_t splits;          /* pixel-dimension split (0 = library chooses, bounded by scratch_floats) */
  float scale;
  uint32_t flags;          /* SSDE_WGRADF_* routing switches (ABI 8; 0 = the library's own choice) */
  float* dw;               /* [c_out][cin_store][k][k] (OIHW); dw += result */
  /* split > 1: every workgroup writes its partial tile to `scratch` with coalesced stores and a second
   * kernel sums the splits in a fixed order (deterministic, no atomics).  ssde_wgrad_scratch_floats()
   * returns the floats needed for the split the library would choose with unlimited scratch. */
  float* scratch;
  int64_t scratch_floats;
  const float* v_pre;      /* optional (ABI 6): the transformed input the FORWARD launch of this layer left behind
                            * (ssde_conv_args.wino_v, same src and prologue); taken only when ssde_wgrad_wants_winograd4()
                            * is true for these arguments -- the input-transform pass of the weight gradient is then skipped */
} ssde_wgrad_args;

enum { SSDE_WGRADF_DIRECT = 1u,        /* never a Winograd weight-gradient kernel */
       SSDE_WGRADF_F2 = 2u,            /* F(2x2,3x3) wherever it is legal, never F(4x4,3x3) */
       SSDE_WGRADF_F4_FORCE = 4u,      /* F(4x4,3x3) wherever it is legal (default: where it is legal and pays) */
       SSDE_WGRADF_NO_STREAMK = 8u,    /* F(4x4,3x3) GEMM: the plain split over K instead of the stream-K runs */
       SSDE_WGRADF_NO_XCD_ORDER = 16u, /* F(4x4,3x3) GEMM: workgroups in launch order */
       SSDE_WGRADF_1X1_CHUNKED = 32u,  /* 1x1 / NIN / Linear: the chunked kernel instead of the pipelined GEMM */
       SSDE_WGRADF_XVEC1 = 64u };      /* F(4x4,3x3) transforms: one channel per thread instead of two */

/* ---- column sums of a gradient: bias and Dense_0(temb) addend gradients ------------------- */
typedef struct ssde_colsum_args {
  const float* g; int32_t g_ld, g_off;   /* [N*hw, g_ld], columns g_off .. g_off+c */
  int32_t n, hw, c;
  float scale;
  float* per_sample;       /* [N, ps_ld] (+ ps_off): per_sample[n, ps_off+j] = scale * sum_hw g  or NULL */
  int32_t ps_ld, ps_off;
  float* total;            /* [c]: scale * sum_{n,hw} g  or NULL */
  float* total2;           /* optional second destination of the same sums (Conv_1.bias and Conv_2.bias share one) */
  float* scratch;          /* N * (min(32, max(1, hw/64)) + 1) * c floats (pixel-slice partials + per-sample sums) */
  uint32_t flags;          /* SSDE_COLSUMF_DEFER (ABI 8): run only the pass over g -- pixel-slice partials [N][slices][c] into scratch,
                            * slices = min(32, max(1, hw/64)); per_sample / total / total2 are written by a later ssde_colsum_finish
                            * that lists this call among its jobs (one launch finishes up to SSDE_FINISH_JOBS column sums) */
  int32_t _pad0;
} ssde_colsum_args;
enum { SSDE_COLSUMF_DEFER = 1u };

/* ---- batched finishing launches (ABI 8) ------------------------------------------------------
 * A training step has ~100 column sums and ~95 GroupNorm backward passes; each ended in one or two launches of 5-8 us that
 * reduce a few thousand floats (690 launches under 13 us = 8 % of the round-4 step).  The producers now leave their partials
 * behind (SSDE_COLSUMF_DEFER, SSDE_GNBWDF_DEFER_PARAMS) and ONE launch finishes up to SSDE_FINISH_JOBS of them, in the same
 * fixed summation order as the per-call kernels (bit-identical results). */
#define SSDE_FINISH_JOBS 16
typedef struct ssde_colsum_job {
  const float* part;       /* [n][slices][c] pixel-slice partials a deferred ssde_colsum left in its scratch */
  float* per_sample;       /* [n, ps_ld] (+ ps_off) or NULL */
  float* total; float* total2;   /* [c] or NULL */
  int32_t n, slices, c, ps_ld, ps_off, _pad0;
} ssde_colsum_job;
typedef struct ssde_colsum_finish_args { int32_t count, _pad0; ssde_colsum_job job[SSDE_FINISH_JOBS]; } ssde_colsum_finish_args;
typedef struct ssde_gn_bwd_job {
  const float* scratch;    /* [rows][c][2]: per-(sample, slice) sums of du xhat and du a deferred ssde_gn_bwd_reduce left behind */
  float* dgamma; float* dbeta;   /* [c] written */
  int32_t rows, c;
} ssde_gn_bwd_job;
typedef struct ssde_gn_bwd_finish_args { int32_t count, _pad0; ssde_gn_bwd_job job[SSDE_FINISH_JOBS]; } ssde_gn_bwd_finish_args;

/* ---- backward through a prologue: dx = d pro(x) / dx applied to dp ------------------------ *
 * GroupNorm backward (nn.GroupNorm autograd): with u = xhat*gamma+beta, y = silu(u) [* dropout mask],
 *   du = dp * mask * silu'(u);  dgamma += sum du*xhat;  dbeta += sum du;  dxh = du*gamma
 *   dx = rstd * (dxh - mean_g(dxh) - xhat * mean_g(dxh*xhat))
 * ssde_gn_bwd_reduce computes the per-(sample,group) means and the per-channel dgamma/dbeta;
 * ssde_prologue_bwd applies the formula (or the plain / SiLU-only variants) and adds the result
 * into the gradient tensors of the (possibly concatenated) sources. */
typedef struct ssde_gn_bwd_reduce_args {
  ssde_src src;            /* x with its GroupNorm prologue (mean/rstd/gamma/beta as in the forward) */
  const float* dp;         /* [N*hw, c0+c1] gradient w.r.t. the prologue output */
  int32_t n, hw;
  float* sums;             /* [N, G, 2]: mean_g(dxh), mean_g(dxh*xhat) */
  float* dgamma; float* dbeta;   /* [c0+c1] written (not accumulated) */
  float* scratch;          /* >= N*slices*(G*2 + C*2) floats */
  int32_t slices;
  uint32_t flags;          /* SSDE_GNBWDF_* (ABI 8) */
  /* ABI 7, optional: with g0 or g1 set the call also applies the formula (what ssde_prologue_bwd would do with dp_ld = c0+c1,
   * dp_off = 0): where one sample's run of whole groups fits the registers of a workgroup (hw <= 8192 at 4 channels per
   * group) dp and x are read ONCE by a single kernel that reduces and applies; elsewhere the call runs the reduction, the
   * finalize and the apply kernel one after the other.  `sums` may then be NULL unless the three-kernel path is taken
   * (callers that cannot know pass it). */
  float* g0; float* g1;    /* gradients of p0 [N*hw, c0] and p1 [N*hw, c1] (NULL: that source needs no gradient) */
  int32_t acc0, acc1;      /* 1: g += ..., 0: g = ... */
  float scale; int32_t _pad1;
} ssde_gn_bwd_reduce_args;

enum { SSDE_GNBWDF_THREE_KERNELS = 1u,     /* ssde_gn_bwd_reduce with g0 / g1: always reduce + finalize + apply, never the one-pass kernel */
       SSDE_GNBWDF_DEFER_PARAMS = 2u };    /* dgamma / dbeta are NOT written: the per-(sample, slice) channel sums stay in scratch
                                            * ([n * slices][c][2]; slices = 1 on the one-pass path) for a later ssde_gn_bwd_finish */

typedef struct ssde_prologue_bwd_args {
  ssde_src src;            /* forward source (p0/p1 may be NULL for SSDE_PRO_NONE) */
  const float* dp;         /* [N*hw, dp_ld] */
  int32_t dp_ld, dp_off;
  int32_t n, hw;
  const float* sums;       /* from ssde_gn_bwd_reduce (GN modes) */
  float scale;
  int32_t acc0, acc1;      /* 1: g += ..., 0: g = ... */
  float* g0; float* g1;    /* gradients of p0 [N*hw, c0] and p1 [N*hw, c1] (NULL: that source needs no gradient) */
} ssde_prologue_bwd_args;

/* ---- attention backward (autograd of layerspp.py:82-86) ---------------------------------- */
typedef struct ssde_attn_bwd_args {
  const float* qkv;        /* [N, L, 3C] forward input */
  const float* o;          /* [N, L, C]  forward output */
  const float* d_o;        /* [N, L, C]  gradient of the output */
  float* dqkv;             /* [N, L, 3C] written */
  float* stats;            /* [N, L, 4] scratch: row max, row sum, D = sum_c dO*O */
  int32_t n, l, c; float scale;
} ssde_attn_bwd_args;

/* ---- denoising-score-matching loss head (losses.py:84-99) --------------------------------- *
 * perturb: x_t = a[n]*x + s[n]*z ;  loss: r = score*s[n] + z (likelihood_weighting=False) or
 * r = score + z/s[n] (True, weighted by g2[n]); per-sample reduce (0.5*sum or mean), batch mean;
 * dscore = d loss / d score, fused in the same pass. */
typedef struct ssde_perturb_args {
  const float* x; const float* z; const float* a; const float* s; float* dst; int32_t n, per;
} ssde_perturb_args;
typedef struct ssde_dsm_loss_args {
  const float* score; const float* z; const float* s; const float* g2;  /* g2 NULL unless likelihood weighting */
  float* dscore;           /* [N, per] or NULL (eval) */
  float* losses;           /* [N] per-sample loss */
  float* loss;             /* [1] batch mean */
  int32_t n, per; int32_t reduce_mean; int32_t likelihood_weighting;
  float grad_scale;        /* multiplies dscore (1/world_size for data-parallel mean) */
  int32_t _pad0;
} ssde_dsm_loss_args;

/* ---- fused optimizer: global-norm clip + Adam + EMA over flat buffers ---------------------- *
 * torch.nn.utils.clip_grad_norm_ (losses.py:49-50), torch.optim.Adam.step (losses.py:29,51) and
 * ExponentialMovingAverage.update (models/ema.py:46-51) in two launches. hyper (device, 8 floats):
 * [lr, beta1, beta2, eps, weight_decay, grad_clip(<0 off), bias_corr1, bias_corr2_sqrt] + [ema one_minus_decay] at [8] */
typedef struct ssde_sumsq_flat_args { const float* x; int64_t numel; float* partial; float* out; } ssde_sumsq_flat_args;
typedef struct ssde_adam_args {
  float* p; const float* g; float* m; float* v; float* ema;   /* ema may be NULL */
  int64_t numel;
  const float* hyper;      /* device [12] */
  const float* gnorm_sq;   /* device [1] from ssde_sumsq_flat, or NULL (no clipping) */
} ssde_adam_args;

typedef struct ssde_memset_args { void* dst; int64_t bytes; int32_t value; int32_t _pad0; } ssde_memset_args;

/* ---- weight re-packing after an optimizer step ------------------------------------------------- *
 * The kernels read weights in their own layouts (conv: [cin/8][tap][cout_pad][8]; Winograd: the LDS image of
 * G g G^T; 1x1 / NIN / Linear: [cin/8][cout_pad][8]; input-gradient variants transposed + rotated), the
 * parameters stay in the reference layouts (state_dict compatibility).  One launch per kind re-packs
 * every weight of the model from a device-resident descriptor table. */
enum { SSDE_PACK_CONV3 = 1, SSDE_PACK_WINO3 = 2, SSDE_PACK_MATRIX = 3, SSDE_PACK_VECTOR = 4,
       SSDE_PACK_WINO4 = 5 /* conv -> the F(4x4,3x3) image of SSDE_TILE_WINOGRAD4 */,
       SSDE_PACK_WINO4R = 6 /* conv -> the per-lane F(4x4,3x3) image of SSDE_TILE_WINOGRAD4R */ };
typedef struct ssde_pack_desc {
  const float* src;        /* parameter: conv [cout][cin][3][3]; matrix [cout][cin]; vector [n] */
  const float* src2;       /* vector: optional second addend (Conv_1.bias + Conv_2.bias) */
  float* dst;
  int32_t kind;
  int32_t cout, cin;       /* parameter dims */
  int32_t cout_l, cin_l;   /* conv: output / input channels of the packed operand; matrix: padded row count of dst */
  int32_t flags;           /* 1: conv = input-gradient weights (transposed, rotated 180 degrees); matrix = transposed */
  int32_t r_off, c_off;    /* matrix / vector: destination row (element) and column offsets */
  int64_t n;               /* conv: elements of dst; matrix / vector: elements of src */
} ssde_pack_desc;
typedef struct ssde_pack_args { const ssde_pack_desc* table; int32_t count; int32_t kind; int64_t max_n; } ssde_pack_args;
typedef struct ssde_axpy_args {   /* dst = (acc ? dst : 0) + alpha * x, optional SiLU' gate: * silu'(gate) */
  const float* x; const float* gate; float* dst; int64_t numel; float alpha; int32_t acc;
} ssde_axpy_args;

/* ---- adaptive RK45 (Dormand-Prince) stage arithmetic on the device ------------------
 * replaces the numpy side of scipy.integrate.solve_ivp(method='RK45') as the reference drives it
 * (sampling.py:466-475, likelihood.py:90-99): fp64 state and stage slopes K[7][n] stay in HBM. */
typedef struct ssde_rk_coefs { double v[7]; } ssde_rk_coefs;
typedef struct ssde_rk_combine_args {     /* dst = y + sum_{j < terms} coef[j] * K[j]   (coef = a_sj * h); dst32 = (float)dst or NULL */
  const double* y; const double* k; int64_t n; int32_t terms; int32_t _pad0; ssde_rk_coefs coef; double* dst; float* dst32;
  int64_t n32;                             /* elements of dst32 (0 = n): the likelihood state is [x | delta log p], only x feeds the U-Net */
} ssde_rk_combine_args;
typedef struct ssde_rk_error_args {       /* out[0] = sqrt(mean(((sum_j coef[j] K[j]) / (atol + max(|y|,|y_new|) rtol))^2)), coef = E_j * h */
  const double* y; const double* y_new; const double* k; int64_t n; ssde_rk_coefs coef; double atol, rtol;
  double* partial; int32_t partial_len; int32_t _pad0; double* out;
} ssde_rk_error_args;
/* Per-evaluation scalars of an ODE right-hand side, in DEVICE memory, so that one captured hipGraph (fill labels ->
 * U-Net program -> drift [-> input-gradient program -> divergence]) serves every evaluation of an adaptive solve: the
 * host uploads this record (24 bytes) before each replay.  label / std are read by SSDE_OP_FILL (tab = &rec.label, ...). */
typedef struct ssde_ode_dyn {
  float label, std;                        /* network label and marginal std of this evaluation (models/utils.py:147-166) */
  float a, g2;                             /* drift coefficient f(x, t) = a x and g(t)^2 (sde_lib.py:61-64, 119-123, 171-176) */
  double* dst;                             /* the integrator's slope row this evaluation fills */
} ssde_ode_dyn;
typedef struct ssde_pf_drift_args {       /* dst = (double)(a * x - (g2 * score) * 0.5) in fp32, RSDE.sde with probability_flow (sde_lib.py:93-97) */
  const float* x; const float* score; double* dst; int64_t numel; float a, g2;
  const ssde_ode_dyn* dyn;                 /* non-NULL: a, g2 and dst come from this device record */
} ssde_pf_drift_args;
/* Hutchinson-Skilling divergence of the probability-flow drift (likelihood.py:26-37, 59-67):
 *   dst[dst_off + b] = sum_i (a eps_i - 0.5 g2 gx_i) eps_i,   gx = (d score / d x)^T eps  (the input-gradient program's result)
 * i.e. eps^T (d drift / d x) eps for drift = a x - g2 score / 2; one workgroup per sample, fixed summation order. */
typedef struct ssde_hutch_div_args {
  const float* gx; const float* eps; double* dst; int64_t dst_off; int32_t n; int32_t per; float a, g2;
  const ssde_ode_dyn* dyn;                 /* non-NULL: a, g2 and dst come from this device record */
} ssde_hutch_div_args;

/* ---- single-op launch entry points ------------------------------------------ */
int ssde_conv2d(const ssde_conv_args* a, void* stream);
int ssde_groupnorm_stats(const ssde_gn_stats_args* a, void* stream);
int ssde_gn_finalize(const ssde_gn_finalize_args* a, void* stream);
/* slices per image of the GroupNorm partials this launch would write (plan only, no device access); 0 = this
 * launch cannot produce them (a workgroup tile would span several images, or c_out % 4 != 0) */
int ssde_conv_gn_slices(const ssde_conv_args* a);
int ssde_upfirdn2d(const ssde_upfirdn_args* a, void* stream);
int ssde_attention(const ssde_attn_args* a, void* stream);
int ssde_embed(const ssde_embed_args* a, void* stream);
int ssde_to_nhwc(const ssde_to_nhwc_args* a, void* stream);
int ssde_to_nchw(const ssde_to_nchw_args* a, void* stream);
int ssde_fused_bias_act(const ssde_bias_act_args* a, void* stream);
int ssde_sumsq(const ssde_sumsq_args* a, void* stream);
int ssde_randn(const ssde_randn_args* a, void* stream);
int ssde_langevin_update(const ssde_langevin_args* a, void* stream);
int ssde_predictor_update(const ssde_predictor_args* a, void* stream);
int ssde_sample_update(const ssde_sample_update_args* a, void* stream);
int ssde_fill_from_table(const ssde_fill_args* a, void* stream);
int ssde_step_inc(const ssde_step_inc_args* a, void* stream);
int ssde_project_update(const ssde_project_args* a, void* stream);
int ssde_rk_combine(const ssde_rk_combine_args* a, void* stream);
int ssde_rk_error_norm(const ssde_rk_error_args* a, void* stream);
int ssde_pf_drift(const ssde_pf_drift_args* a, void* stream);
int ssde_hutch_div(const ssde_hutch_div_args* a, void* stream);
int ssde_conv_wgrad(const ssde_wgrad_args* a, void* stream);
/* 1 when ssde_conv_wgrad would run these arguments on the F(4x4,3x3) path (wgrad_wino4.hip), i.e. when a forward launch
 * may usefully fill ssde_conv_args.wino_v for it; shape-only query (pointers are not dereferenced) */
int ssde_wgrad_wants_winograd4(const ssde_wgrad_args* a);
int ssde_colsum(const ssde_colsum_args* a, void* stream);
int ssde_colsum_finish(const ssde_colsum_finish_args* a, void* stream);
int ssde_gn_bwd_finish(const ssde_gn_bwd_finish_args* a, void* stream);
/* rows of the scratch a deferred ssde_gn_bwd_reduce (SSDE_GNBWDF_DEFER_PARAMS) leaves behind: n on the one-pass path, n * slices
 * on the three-kernel path -- what ssde_gn_bwd_job.rows must be (shape-only query) */
int ssde_gn_bwd_scratch_rows(const ssde_gn_bwd_reduce_args* a);
int ssde_gn_bwd_reduce(const ssde_gn_bwd_reduce_args* a, void* stream);
int ssde_prologue_bwd(const ssde_prologue_bwd_args* a, void* stream);
int ssde_attention_bwd(const ssde_attn_bwd_args* a, void* stream);
int ssde_perturb(const ssde_perturb_args* a, void* stream);
int ssde_dsm_loss(const ssde_dsm_loss_args* a, void* stream);
int ssde_sumsq_flat(const ssde_sumsq_flat_args* a, void* stream);
int ssde_adam_clip_ema(const ssde_adam_args* a, void* stream);
int ssde_memset(const ssde_memset_args* a, void* stream);
int ssde_axpy(const ssde_axpy_args* a, void* stream);
int ssde_pack_weights(const ssde_pack_args* a, void* stream);

/* ---- programs: a whole U-Net forward / PC step as one call -------------------
 * A program is a flat array of tagged ops built once by the host (it replaces the
 * Python module walk of NCSNpp.forward, ncsnpp.py:232-381, and of pc_sampler's
 * loop body, sampling.py:403-407). */
enum {
  SSDE_OP_CONV = 1, SSDE_OP_GN_STATS = 2, SSDE_OP_UPFIRDN = 3, SSDE_OP_ATTN = 4, SSDE_OP_EMBED = 5,
  SSDE_OP_TO_NHWC = 6, SSDE_OP_TO_NCHW = 7, SSDE_OP_BIAS_ACT = 8, SSDE_OP_SUMSQ = 9, SSDE_OP_RANDN = 10,
  SSDE_OP_LANGEVIN = 11, SSDE_OP_PREDICTOR = 12, SSDE_OP_FILL = 13, SSDE_OP_STEP_INC = 14,
  SSDE_OP_WGRAD = 15, SSDE_OP_COLSUM = 16, SSDE_OP_GN_BWD_REDUCE = 17, SSDE_OP_PROLOGUE_BWD = 18,
  SSDE_OP_ATTN_BWD = 19, SSDE_OP_PERTURB = 20, SSDE_OP_DSM_LOSS = 21, SSDE_OP_SUMSQ_FLAT = 22,
  SSDE_OP_ADAM = 23, SSDE_OP_MEMSET = 24, SSDE_OP_AXPY = 25, SSDE_OP_PACK = 26, SSDE_OP_PROJECT = 27,
  SSDE_OP_GN_FINALIZE = 28, SSDE_OP_PF_DRIFT = 29, SSDE_OP_HUTCH_DIV = 30, SSDE_OP_COLSUM_FINISH = 31, SSDE_OP_GN_BWD_FINISH = 32
};
typedef struct ssde_op {
  int32_t kind; int32_t flops_class;   /* flops_class: free tag echoed by timing */
  union {
    ssde_conv_args conv; ssde_gn_stats_args gn; ssde_upfirdn_args fir; ssde_attn_args attn;
    ssde_embed_args embed; ssde_to_nhwc_args to_nhwc; ssde_to_nchw_args to_nchw;
    ssde_bias_act_args bias_act; ssde_sumsq_args sumsq; ssde_randn_args randn;
    ssde_langevin_args langevin; ssde_predictor_args predictor; ssde_fill_args fill;
    ssde_step_inc_args step_inc;
    ssde_wgrad_args wgrad; ssde_colsum_args colsum; ssde_gn_bwd_reduce_args gn_bwd; ssde_prologue_bwd_args pro_bwd;
    ssde_attn_bwd_args attn_bwd; ssde_perturb_args perturb; ssde_dsm_loss_args dsm_loss;
    ssde_sumsq_flat_args sumsq_flat; ssde_adam_args adam; ssde_memset_args memset; ssde_axpy_args axpy;
    ssde_pack_args pack; ssde_project_args project; ssde_gn_finalize_args gn_fin;
    ssde_pf_drift_args pf_drift; ssde_hutch_div_args hutch_div;
    ssde_colsum_finish_args colsum_fin; ssde_gn_bwd_finish_args gn_bwd_fin;
  } u;
} ssde_op;

int ssde_program_run(const ssde_op* ops, int32_t n_ops, void* stream);
/* same, bracketing every op with HIP events on `stream`; ms[i] = duration of op i */
int ssde_program_run_timed(const ssde_op* ops, int32_t n_ops, void* stream, float* ms);
/* capture the program into a hipGraph on `stream` (must be a non-default stream) */
int ssde_graph_capture(const ssde_op* ops, int32_t n_ops, void* stream, void** graph_out);
int ssde_graph_launch(void* graph, void* stream);
int ssde_graph_destroy(void* graph);

/* ---- misc --------------------------------------------------------------------- */
/* ---- plans: a lowered program as a position-independent blob, for hosts without Python --------------------------
 * SURVEY 8(b).  The lowering of NCSNpp.forward (models/ncsnpp.py:232-381) and of pc_sampler's loop body
 * (sampling.py:403-407) lives in ONE place, score_sde_pytorch_amd/engine.py + pc_engine.py; plan_export.py serialises
 * its result.  Layout of a blob (little endian, all structs below packed as declared):
 *   ssde_plan_header | ssde_plan_region[n_regions] | ssde_op[n_ops] | ssde_op[n_refresh_ops] |
 *   ssde_plan_reloc[n_relocs] | ssde_plan_param_entry[n_params] | data[data_bytes]
 * Every pointer field of every op (and of the weight re-pack descriptor tables, which live in constant regions) is
 * zero in the blob and listed as a relocation (region, byte offset).  ssde_plan_load allocates the regions with
 * hipMalloc, uploads the constant ones, patches the pointers. */
enum { SSDE_REGION_ZERO = 0,    /* activations, I/O, sampler state: zero-initialised                      */
       SSDE_REGION_CONST = 1 }; /* initial contents in the blob: packed weights, parameters, tables       */
enum { SSDE_RELOC_OP = 0, SSDE_RELOC_REFRESH_OP = 1, SSDE_RELOC_REGION = 2 };
enum { SSDE_PLAN_UNET = 0, SSDE_PLAN_PC = 1, SSDE_PLAN_TRAIN = 2 };
enum { SSDE_IO_X = 0, SSDE_IO_COND = 1, SSDE_IO_SIGMA = 2, SSDE_IO_STD = 3, SSDE_IO_OUT = 4, SSDE_IO_XMEAN = 5,
       SSDE_IO_STEP = 6, SSDE_IO_SEED = 7,
       /* training plans (losses.FusedTrainStep): clean batch, noise, per-sample mean coefficient / std / g^2, scalar loss,
        * the 12-float hyper-parameter record of ssde_adam_clip_ema, the dropout seed word, d loss / d out, d loss / d x,
        * the flat gradient and the flat parameter buffer (reference layouts, state_dict order) */
       SSDE_IO_BATCH = 8, SSDE_IO_Z = 9, SSDE_IO_A = 10, SSDE_IO_S = 11, SSDE_IO_G2 = 12, SSDE_IO_LOSS = 13, SSDE_IO_HYPER = 14,
       SSDE_IO_DROP_SEED = 15, SSDE_IO_GOUT = 16, SSDE_IO_GX = 17, SSDE_IO_GRAD = 18, SSDE_IO_PARAMS = 19, SSDE_IO_SLOTS = 24 };
typedef struct ssde_plan_header {
  char magic[8];                /* "SSDEPLN1" */
  int32_t abi_version, sizeof_op;
  int32_t n_regions, n_ops, n_refresh_ops, n_relocs, n_params;
  int32_t kind;                 /* SSDE_PLAN_* */
  int32_t batch, channels, height, width;
  int32_t nfe_per_iteration;    /* sampler plans: U-Net evaluations per PC iteration; else 1 */
  int32_t sde_steps;            /* sampler plans: N (length of the step tables) */
  int32_t io[SSDE_IO_SLOTS];    /* region ids by SSDE_IO_*; -1 = absent */
  int32_t seg[4];               /* training plans: first op of the forward / loss head / backward / optimizer segments
                                   (ops [0, seg[0]) perturb the batch: losses.py:84-88) */
  int64_t n_flat;               /* training plans: floats of the flat parameter / gradient buffers */
  int64_t data_bytes;
} ssde_plan_header;
typedef struct ssde_plan_region { int64_t bytes; int64_t data_offset; int32_t kind; int32_t _pad0; char name[32]; } ssde_plan_region;
typedef struct ssde_plan_reloc {
  int32_t target_kind;          /* SSDE_RELOC_*: where the pointer is stored */
  int32_t target;               /* op index / region id */
  int64_t byte_offset;          /* of the pointer field inside the op / the region */
  int32_t region; int32_t _pad0;/* what it points to */
  int64_t offset;
} ssde_plan_reloc;
typedef struct ssde_plan_param_entry { char name[96]; int32_t region; int32_t _pad0; int64_t offset; int64_t numel; } ssde_plan_param_entry;
typedef struct ssde_plan ssde_plan;

int ssde_plan_load(const void* blob, size_t bytes, ssde_plan** out);
int ssde_plan_load_file(const char* path, ssde_plan** out);
int ssde_plan_destroy(ssde_plan* p);
int ssde_plan_info(const ssde_plan* p, ssde_plan_header* out);
/* device address of a parameter in the reference's layout, by state_dict name (index < 0) or by index */
int ssde_plan_param(const ssde_plan* p, const char* name, int32_t index, float** dev, int64_t* numel, const char** name_out);
/* after writing parameters: rebuild every kernel-layout weight copy on the device (ssde_pack_weights) */
int ssde_plan_refresh_weights(ssde_plan* p, void* stream);
/* replaces NCSNpp.forward (models/ncsnpp.py:232): device pointers x, out [B,C,H,W], cond [B]; sigma / std NULL unless
 * the plan has those inputs (discrete labels with scale_by_sigma; VP score head, models/utils.py:147-159) */
int ssde_unet_forward(ssde_plan* p, const float* x, const float* cond, const float* sigma, const float* std_, float* out, void* stream);
/* replaces pc_sampler's loop (sampling.py:390-409): load the prior sample, run iterations, read the state */
int ssde_pc_reset(ssde_plan* p, const float* x_T, uint64_t seed, void* stream);
int ssde_pc_run(ssde_plan* p, int32_t n_iterations, int32_t use_graph, void* stream);
/* ---- training plans (SSDE_PLAN_TRAIN, exported from losses.FusedTrainStep by plan_export.export_train_plan) ----
 * replaces step_fn (losses.py:179-208) for one optimisation step of the denoising score matching loss: perturb the batch
 * (x_t = a[n] batch + s[n] z, losses.py:84-88), forward, loss head + d loss / d score, backward, gradient clipping + Adam +
 * EMA (losses.py:41-51, models/ema.py:46-51), re-pack of the kernel-layout weights.  All tensors are DEVICE pointers:
 * batch, z [B,C,H,W]; a, s, labels [B] (mean coefficient, marginal std and network label of every sample's t -- the
 * host evaluates sde.marginal_prob, a handful of scalars); g2 [B] or NULL (likelihood weighting); loss_out: 1 float or
 * NULL.  `hyper` is a HOST pointer to 9 floats: lr, beta1, beta2, eps, weight_decay, grad_clip, 1 - beta1^t,
 * sqrt(1 - beta2^t), 1 - ema_decay. */
int ssde_train_step(ssde_plan* p, const float* batch, const float* z, const float* a, const float* s, const float* labels,
                    const float* g2, const float* hyper, uint32_t dropout_seed, float* loss_out, void* stream);
/* the two halves of autograd through the network on a training plan: out = model(x, cond) in train mode (activations
 * stay resident, dropout masks from `dropout_seed`), then the vector-Jacobian products of d loss / d out = dout:
 * dx [B,C,H,W] (plans exported with the input gradient; else pass NULL) and dparams = the flat parameter gradient
 * (n_flat floats, reference layouts in state_dict order; NULL to leave it in the plan).  Replaces loss.backward()
 * (losses.py:196) for hosts that bring their own loss. */
int ssde_train_forward(ssde_plan* p, const float* x, const float* cond, const float* sigma, const float* std_, uint32_t dropout_seed,
                       float* out, void* stream);
int ssde_unet_backward(ssde_plan* p, const float* dout, float* dx, float* dparams, void* stream);
/* device-to-device copy between an I/O region of the plan (SSDE_IO_*) and the caller's buffer: to_plan = 0 reads the region
 * (e.g. SSDE_IO_PARAMS: the flat parameter buffer after training steps), 1 writes it */
int ssde_plan_copy_io(ssde_plan* p, int32_t slot, void* buf, int64_t bytes, int32_t to_plan, void* stream);
int ssde_pc_state(ssde_plan* p, float* x, float* x_mean, void* stream);

int ssde_abi_version(void);
int ssde_sizeof_op(void);             /* lets the ctypes mirror verify its layout */
const char* ssde_last_error(void);
int ssde_conv_lds_bytes(const ssde_conv_args* a);   /* diagnostic: LDS a launch would use */
int64_t ssde_wgrad_scratch_floats(const ssde_wgrad_args* a);   /* scratch the preferred split needs (0: none), < 0: error */
/* diagnostic: `workgroups` workgroups of 4 waves (pass the CU count: one wave per SIMD) issue `iters` x 8 back-to-back
 * v_mfma_f32_32x32x2_f32 on register operands (nothing else: no LDS, no memory).  Timed by the caller,
 * workgroups x 4 x iters x 8 x 4096 FLOP / t is the fp32 matrix rate this device SUSTAINS at the clock it settles on
 * under load -- bench.py reports it beside the 157.3 TFLOP/s data-sheet peak (2.4 GHz) the roofline fractions are quoted
 * against.  sink: at least 64 floats (never written). */
int ssde_mfma_probe(int32_t workgroups, int32_t iters, float* sink, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SSDE_H_ */
