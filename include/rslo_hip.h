/*
 * rslo_hip.h -- C ABI of librslo_hip.so, the MI355X (gfx950) implementation of the
 * RSLO two-frame LiDAR-odometry hot path.
 *
 * Boundary rules
 *   - extern "C", plain device pointers + sizes, no torch types.  Every pointer is a
 *     DEVICE pointer unless the parameter name starts with h_.
 *   - The caller owns every buffer (inputs, outputs, workspaces).  Nothing is allocated
 *     or freed inside the library; *_ws_bytes() functions size the workspaces.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing
 *     synchronises.  Data-dependent counts are written to device ints the caller reads.
 *   - Return value: 0 on success, a negative RSLO_E* code otherwise; rslo_last_error()
 *     gives the text (thread-local).
 *   - Coordinates are int32 (b, z, y, x); features are row-major fp32 [rows, channels];
 *     sparse-conv weights are [K = kz*ky*kx, Cin, Cout] (a view of spconv's
 *     [kz,ky,kx,Cin,Cout]); a neighbour table nbr[rows*K] holds the contributing row of
 *     the other side for kernel offset k, or -1.
 *
 * Each entry point cites the reference interface it replaces (paths relative to the
 * DecaYale/RSLO tree).  The spconv_plus / apex / kornia sources are not part of that tree
 * (Dockerfile:55-61,93-97; freeze.yml:212); for those the call site is cited.
 */
#ifndef RSLO_HIP_H_
#define RSLO_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define RSLO_API __attribute__((visibility("default")))
#else
#define RSLO_API
#endif

#define RSLO_OK 0
#define RSLO_EINVAL (-1)   /* bad argument / unsupported shape                         */
#define RSLO_ELAUNCH (-2)  /* hipLaunch / runtime error                                */
#define RSLO_ERANGE (-3)   /* linear voxel index does not fit the 32-bit key space     */
#define RSLO_EWS (-4)      /* workspace too small                                      */

RSLO_API int rslo_abi_version(void);
RSLO_API const char *rslo_last_error(void);

/* ------------------------------------------------------------------------------------
 * a1  Voxelization.  Replaces spconv.utils.VoxelGenerator.generate
 *     (rslo/builder/voxel_builder.py:36-54,83-94; called rslo/data/preprocess.py:493).
 * points [P,F] fp32 (x,y,z first).  First-come voxel numbering in point order, at most
 * T points per voxel kept in point order, processing stops at the first point that would
 * open voxel number max_voxels, coordinates stored (z,y,x).
 * Outputs are sized for max_voxels; rows >= *d_nvox are zero.
 * ------------------------------------------------------------------------------------ */
RSLO_API size_t rslo_voxelize_ws_bytes(int64_t P);
RSLO_API int rslo_voxelize(const float *points, int64_t P, int F, const float *h_range6,
                  const float *h_vsize3, const int32_t *h_grid_xyz, int T, int max_voxels,
                  void *ws, size_t ws_bytes, float *voxels /*[max_voxels,T,F]*/,
                  int32_t *coords /*[max_voxels,3] zyx*/, int32_t *num_points /*[max_voxels]*/,
                  int32_t *d_nvox /*[1]*/, void *stream);

/* a4  SimpleVoxel_XYZINormalC.forward (rslo/models/voxel_encoder.py:258-280):
 *     per-voxel mean of the F features, channels 4:7 divided by (norm + 1e-12). */
RSLO_API int rslo_vfe_mean(const float *voxels /*[M,T,F]*/, const int32_t *num_points /*[M]*/, int64_t M,
                  int T, int F, float *out /*[M,F]*/, void *stream);

/* ------------------------------------------------------------------------------------
 * a5  Site index + rulebooks.  Replace the implicit indice-pair builds behind
 *     spconv.SparseConvTensor / SubMConv3d / SparseConv3d / SparseInverseConv3d
 *     (rslo/models/middle.py:119-213,223-225).
 * The site index is an open-addressing hash: keys[cap] (uint32 linear index, 0xFFFFFFFF =
 * empty), vals[cap] (row).  cap = rslo_hash_capacity(n) (power of two >= 2n).
 * ------------------------------------------------------------------------------------ */
RSLO_API int64_t rslo_hash_capacity(int64_t n);
RSLO_API int rslo_hash_build(const int32_t *coords /*[N,4]*/, int64_t N, int B, const int32_t *h_dims3,
                    uint32_t *keys, int32_t *vals, int64_t cap, void *stream);
/* SubM: output set == input set in the given order; nbr[o][k] = row at coords[o]+(k-ks/2). */
RSLO_API int rslo_rulebook_subm(const int32_t *coords, int64_t N, int B, const int32_t *h_dims3,
                       const int32_t *h_ks3, const uint32_t *keys, const int32_t *vals, int64_t cap,
                       int32_t *nbr /*[N,K]*/, void *stream);
/* Strided conv, step 1: mark the output sites (in = out*stride - pad + k) in a bitmap over
 * the output volume and rank them; *d_count receives the number of output sites M.
 * words = rslo_conv_bitmap_words(B, out_dims).  scan_ws: rslo_scan_ws_bytes(words). */
RSLO_API int64_t rslo_conv_bitmap_words(int B, const int32_t *h_out_dims3);
RSLO_API size_t rslo_scan_ws_bytes(int64_t n);
RSLO_API int rslo_conv_out_count(const int32_t *coords_in, int64_t N, int B, const int32_t *h_ks3,
                        const int32_t *h_stride3, const int32_t *h_pad3, const int32_t *h_out_dims3,
                        uint32_t *bitmap /*[words]*/, int32_t *word_prefix /*[words]*/, int64_t words,
                        void *scan_ws, size_t scan_ws_bytes, int32_t *d_count /*[1]*/, void *stream);
/* step 2: emit the M output coordinates in ascending linear-index order. */
RSLO_API int rslo_conv_out_coords(const uint32_t *bitmap, const int32_t *word_prefix, int64_t words, int B,
                         const int32_t *h_out_dims3, int32_t *out_coords /*[M,4]*/, int64_t M,
                         void *stream);
/* step 3: nbr[o][k] = input row at out*stride - pad + k (probe of the INPUT site index). */
RSLO_API int rslo_rulebook_conv(const int32_t *coords_out, int64_t M, int B, const int32_t *h_in_dims3,
                       const int32_t *h_ks3, const int32_t *h_stride3, const int32_t *h_pad3,
                       const uint32_t *in_keys, const int32_t *in_vals, int64_t in_cap,
                       int32_t *nbr /*[M,K]*/, void *stream);
/* step 4: nbrT[i][k] = output row o with o*stride - pad + k == in (probe of the OUTPUT index);
 * the table the dgrad and SparseInverseConv3d use. */
RSLO_API int rslo_rulebook_conv_T(const int32_t *coords_in, int64_t N, int B, const int32_t *h_out_dims3,
                         const int32_t *h_ks3, const int32_t *h_stride3, const int32_t *h_pad3,
                         const uint32_t *out_keys, const int32_t *out_vals, int64_t out_cap,
                         int32_t *nbrT /*[N,K]*/, void *stream);

/* ------------------------------------------------------------------------------------
 * a6/a7  Sparse convolution arithmetic (spconv SubMConv3d / SparseConv3d /
 *        SparseInverseConv3d forward + backward; call sites middle.py:119-213).
 * fwd : out[o] = act( bias + sum_k in[nbr[o][k]] . W[kk] ),  kk = flip_k ? K-1-k : k
 *       act = LeakyReLU(slope) fused when slope != 1 (middle.py:99-101,123...).
 * dgrad: din[i] = sum_k dout[nbrT[i][k]] . W[kk]^T   (for SubM pass nbr and flip_k=1)
 * wgrad: dW[k]  = sum_o in[nbr[o][k]]^T dout[o]      (deterministic two-stage reduce)
 * Supported channel counts: 1..64 on both sides.
 * ------------------------------------------------------------------------------------ */
/*     row_order (int32 [n_out] or NULL): the order in which the table rows are grouped into the kernels' 16/32-row tiles
 *     (rslo_rulebook_row_order).  Purely a scheduling hint: out[o] is written for every o and does not depend on it. */
RSLO_API int rslo_spconv_fwd(const float *in, int cin, const float *W, const float *bias, const int32_t *nbr,
                    const int32_t *row_order, int64_t n_out, int K, int cout, int flip_k, float act_slope,
                    float *out, void *stream);
RSLO_API int rslo_spconv_dgrad(const float *dout, int cout, const float *W, const int32_t *nbrT,
                      const int32_t *row_order, int64_t n_in, int K, int cin, int flip_k, float *din, void *stream);
/*     W [K,Cin,Cout] -> Wt [K,Cout,Cin].  rslo_spconv_fwd(dout, Cout, Wt, NULL, nbrT, ...) then equals
 *     rslo_spconv_dgrad(dout, Cout, W, nbrT, ...) with weight reads contiguous along the lane index
 *     (12-50 % faster on the 64-channel layers); the host mirror uses this form. */
RSLO_API int rslo_weight_transpose(const float *W, int K, int cin, int cout, float *Wt, void *stream);
/*     fp32-accurate sparse conv on the bf16 matrix cores (channel counts 32 / 64): every fp32 value is split exactly
 *     into three bf16 pieces (hi + mid + lo); six bf16 MFMAs (hh, hm, mh, hl, mm, lh) with fp32 accumulation replace
 *     the fp32 MFMAs -- error below one fp32 ulp per product, 2.5x fewer matrix-core cycles.
 *     rslo_weight_split: W [K,Cin_w,Cout_w] fp32 -> Ws (3 planes of bf16, K*cin_op*cout_op each, MFMA operand order);
 *     transpose = 1 gives the operator of the data gradient (cin_op = Cout_w, cout_op = Cin_w).
 *     rslo_spconv_fwd_split == rslo_spconv_fwd with the split weights. */
RSLO_API size_t rslo_weight_split_bytes(int K, int cin, int cout);
RSLO_API int rslo_weight_split(const float *W, int K, int cin_op, int cout_op, int transpose, void *Ws, void *stream);
/*      rslo_weight_split_many: the same split for every 32/64-channel layer of a model, both orientations (forward and
 *      data gradient), in ONE launch -- the weights are constant between optimizer steps.  desc_dev = device array of
 *      n_layers descriptors, max_weight_elems = max over layers of K * cin * cout. */
typedef struct {
  const float *W;   /* [K,cin,cout] */
  void *ws_fwd;     /* rslo_weight_split_bytes(K, cin, cout) bytes: operand of rslo_spconv_fwd_split(x, .., cin, cout) */
  void *ws_dgrad;   /* same size: operand of the data-gradient call rslo_spconv_fwd_split(dout, .., cout, cin) */
  int32_t K, cin, cout;
} RsloWeightSplitDesc;
RSLO_API int rslo_weight_split_many(const RsloWeightSplitDesc *desc_dev, int n_layers, int64_t max_weight_elems,
                                    void *stream);
RSLO_API int rslo_spconv_fwd_split(const float *in, int cin, const void *Ws, const float *bias, const int32_t *nbr,
                                   const int32_t *row_order, int64_t n_out, int K, int cout, int flip_k,
                                   float act_slope, float *out, void *stream);
/*     Tiling of k_spconv_v6 behind rslo_spconv_fwd_split: a tile is 16*rbw rows (rbw 1, 2, 4) and ks waves (1, 2, 4) share
 *     it, each walking 1/ks of the tile's active kernel offsets; their accumulators are added through LDS in wave order
 *     (a fixed summation order, results within fp32 rounding of the one-wave form).  0 = the library chooses per layer
 *     shape (default: 32 rows and two waves except for 32 -> 32 channels).  Same as the switches spconv_rbw / spconv_ks. */
RSLO_API void rslo_spconv_set_tiling(int rbw, int ks);
/*     Round 6, the replayed inference pass (evaluate.py:363-408 -> rslo/models/middle.py:219-245 per frame; rslo_amd/inference.py):
 *     a capacity-laid-out rulebook has more rows than the scan has sites, the rows past the level's count being padding rows
 *     (rslo_plan_encoder_pad_tails), and the count lives in a device word.  rslo_spconv_set_live_rows(p) hands that word to the
 *     NEXT rslo_spconv_fwd / rslo_spconv_fwd_split launch (and only that one): workgroups whose rows all lie at or past *p
 *     return at once -- their output rows are left unwritten; nothing reads a padding row.  Ignored with a row order.
 *     Per calling thread. */
RSLO_API void rslo_spconv_set_live_rows(const int32_t *n_live_dev);
/*     Tuning switches of the launch code.  The library reads NO environment variable: tile shapes and kernel variants that
 *     exist for A/B measurements and for the parity tests that pin every tiling against the oracle are set through these
 *     calls (process-wide; the defaults are the measured choices).  Names (rslo_tuning_name(i), i = 0 .. until NULL):
 *     conv2d_wgrad_s2_fullres, conv2d_wgrad_nb, conv2d_wgrad_wgs, conv2d_fwd_tr, conv2d_fwd_mtw, conv2d_fwd_occ,
 *     conv2d_fwd_kc, conv2d_fwd_lean, conv2d_fwd_xsc, conv2d_s2_mtw, conv2d_s2_xsc, bn_small_rc, spconv_rbw, spconv_ks,
 *     spconv_v, spconv_wgrad_split, wgrad_xcd, vfe_lds, chamfer, chamfer_segments, dense_tiled, conv1x1_split, conv2d_fwd_wl
 *     (meanings: csrc/rslo_common.h RsloTune).  Every setting
 *     computes the same products; only tiling, summation grouping and launch geometry change.  Unknown name -> RSLO_EINVAL.
 *     (The reference has no counterpart: spconv / cuDNN pick their algorithms internally.) */
RSLO_API int rslo_tuning_set(const char *name, int value);
RSLO_API int rslo_tuning_get(const char *name, int *value);
RSLO_API const char *rslo_tuning_name(int index);
/*     bf16 feature path (BASELINE config C4: bf16 features, int32 rulebook, fp32 accumulate): in / out are bf16 rows
 *     [N,C], Wb the weights rounded to bf16 in MFMA operand order (rslo_weight_to_bf16, K*cin*cout*2 bytes; transpose
 *     = 1 for the data gradient), bias fp32.  Channel counts 32 / 64. */
RSLO_API int rslo_weight_to_bf16(const float *W, int K, int cin_op, int cout_op, int transpose, void *Wb, void *stream);
RSLO_API int rslo_spconv_fwd_bf16(const void *in, int cin, const void *Wb, const float *bias, const int32_t *nbr,
                                  const int32_t *row_order, int64_t n_out, int K, int cout, int flip_k,
                                  float act_slope, void *out, void *stream);
/*     bf16 feature path, backward: g = dout * act'(y) on bf16 rows (+ per-block column sums of g in fp32, stage 1 of the
 *     bias gradient; partial [rslo_leaky_bwd_colsum_bf16_blocks(rows, cols), cols] or NULL; cols must divide 2048), and
 *     the pair-list weight gradient with bf16 rows on both sides, fp32 accumulation and fp32 dW / dbias (workspace size:
 *     rslo_spconv_wgrad_pairs_ws_bytes; dbias requires bias_partial).  The data gradient is rslo_spconv_fwd_bf16 on the
 *     transposed operand (rslo_weight_to_bf16(..., transpose = 1)). */
RSLO_API int64_t rslo_leaky_bwd_colsum_bf16_blocks(int64_t rows, int cols);
RSLO_API int rslo_leaky_bwd_colsum_bf16(const void *y, const void *dout, int64_t rows, int cols, float slope, void *g,
                                        float *partial, void *stream);
RSLO_API int rslo_spconv_wgrad_pairs_bf16(const void *in, int cin, const void *dout, int cout, const int32_t *pairs_in,
                                          const int32_t *pairs_out, const int32_t *koff, int64_t n_out, int K, void *ws,
                                          size_t ws_bytes, float *dW, float *dbias, const float *bias_partial,
                                          int n_bias_partial, void *stream);
RSLO_API size_t rslo_spconv_wgrad_ws_bytes(int64_t n_out, int K, int cin, int cout);
RSLO_API int rslo_spconv_wgrad(const float *in, int cin, const float *dout, int cout, const int32_t *nbr,
                      int64_t n_out, int K, void *ws, size_t ws_bytes, float *dW /*[K,cin,cout]*/,
                      float *dbias /*[cout] or NULL*/, void *stream);
/* Scheduling order of a neighbour table's rows for rslo_spconv_fwd / _dgrad / _fwd_split / _fwd_bf16: inside every
 * window of 2048 consecutive rows, rows are sorted by their neighbour mask so that the kernels' 16/32-row tiles issue
 * matrix operations for fewer kernel offsets no row of the tile needs.  order [n_rows] is a permutation of 0..n_rows-1
 * (deterministic).  flip_k = 1: order for calls made with flip_k = 1 (SubM data gradient).  Replaces nothing in the
 * reference: spconv's gather-GEMM-scatter walks complete pair lists per offset and has no tile skipping to feed. */
RSLO_API int rslo_rulebook_row_order(const int32_t *nbr, int64_t n_rows, int K, int flip_k, int32_t *order,
                                     void *stream);
/* Pair-list view of a neighbour table (the spconv-1.x rulebook layout): pairs of offset k are contiguous in
 * [koff[k], koff[k+1]) in ascending output row; pairs_in/out need room for n_rows*K entries (upper bound),
 * koff [K+1] lives on the device.  Built once per indice_key; feeds the weight-gradient kernel. */
RSLO_API size_t rslo_rulebook_pairs_ws_bytes(int64_t n_rows, int K);
RSLO_API int rslo_rulebook_pairs(const int32_t *nbr, int64_t n_rows, int K, void *ws, size_t ws_bytes,
                        int32_t *pairs_in, int32_t *pairs_out, int32_t *koff /*[K+1]*/, void *stream);
/* a11 / a12 / a15  The element-wise tail of the BEV head (rslo/models/odom_pred.py:227-264):
 *   rslo_tq_normalize_*   tq_map = cat(t, q / |q|) per cell (odom_pred.py:229-234), tq [B,7,cells];
 *   rslo_conf_softmax_*   both ConfidenceModule softmaxes (rslo/layers/confidence.py:26-34): where(outside, -1000, logit) / T,
 *                         softmax over the cells of a sample; T = 1 -> t_conf, r_conf [B,cells] (with gradient) and
 *                         T = temperature (20) -> conf_temp [B,2,cells] (no gradient) from one pass; outside: 1 byte per cell;
 *   rslo_head_masks_*     occupancy pyramid (mask_gen_pools: MaxPool 3/2/1, odom_pred_base.py:228-231), loss weights
 *                         w_0 = mask * conf_temp, w_{k+1} = occ_{k+1} * AvgPool(3,2,1)(w_k) (hier_weight_gen, odom_pred.py:148,262-264),
 *                         masked pyramid predictions pred_k * (occ_k > 0), masked pose maps tq * mask, tq_g * mask. */
RSLO_API int rslo_tq_normalize_fwd(const float *tq, int B, int64_t cells, float *out, void *stream);
RSLO_API int rslo_tq_normalize_bwd(const float *tq, const float *grad, int B, int64_t cells, float *dtq, void *stream);
RSLO_API int rslo_conf_softmax_fwd(const float *t_logit, const float *r_logit, const unsigned char *outside, int B,
                                   int cells, float temperature, float *t_conf, float *r_conf, float *conf_temp,
                                   void *stream);
RSLO_API int rslo_conf_softmax_bwd(const float *t_conf, const float *r_conf, const float *g_t, const float *g_r,
                                   const unsigned char *outside, int B, int cells, float *d_t_logit, float *d_r_logit,
                                   void *stream);
typedef struct {
  const float *mask;        /* [B,1,H,W] 0/1 */
  const float *conf;        /* [B,2,H,W] */
  const float *tq, *tq_g;   /* [B,7,H,W] */
  const float *pred[3];     /* pred[k-1]: [B,7,H>>k,W>>k], k = 1 .. levels-1 */
  float *w[4];              /* out: w[k] [B,2,H>>k,W>>k] */
  float *occ[4];            /* out: occ[k], k >= 1: [B,1,H>>k,W>>k] */
  float *mpred[3];          /* out: masked predictions */
  float *mtq, *mtq_g;       /* out */
  int32_t B, H, W, levels;  /* levels = 1 + number of pyramid predictions (<= 4) */
} RsloHeadMasks;
typedef struct {
  const float *mask;
  const float *occ[4];
  const float *g_mpred[3];  /* may be NULL */
  const float *g_mtq;       /* may be NULL */
  float *d_pred[3];
  float *d_tq;
  int32_t B, H, W, levels;
} RsloHeadMasksBwd;
RSLO_API int rslo_head_masks_fwd(const RsloHeadMasks *h_a, void *stream);
RSLO_API int rslo_head_masks_bwd(const RsloHeadMasksBwd *h_a, void *stream);
/*     Input of a deblock (odom_pred.py:219-221: deblock(cat([x, skip], 1)) with deblock[0] = nn.Upsample(scale)):
 *     out [B,Ca+Cb,scale*H,scale*W] = nearest-neighbour upsampling of the channel concatenation of a [B,Ca,H,W] and
 *     b [B,Cb,H,W] in one launch; _bwd: da / db (either may be NULL) = the scale x scale window sums of grad, summed in
 *     upsample_nearest2d_backward's order. */
RSLO_API int rslo_cat_upsample_fwd(const float *a, const float *b, int B, int Ca, int Cb, int H, int W, int scale, float *out,
                                   void *stream);
RSLO_API int rslo_cat_upsample_bwd(const float *grad, int B, int Ca, int Cb, int H, int W, int scale, float *da, float *db,
                                   void *stream);
/*     The voted pose odom [B,7] = (t, q) -> t [B,3], r [B,4] = q / (|q| + 1e-12) (odom_pred.py:279-288) and its gradient
 *     (g_t / g_r may be NULL = zeros), one launch each way. */
RSLO_API int rslo_pose_tail_fwd(const float *odom, int B, float *t, float *r, void *stream);
RSLO_API int rslo_pose_tail_bwd(const float *odom, const float *g_t, const float *g_r, int B, float *d_odom, void *stream);

/* a21  Loss assembly in one launch each way: AdaptiveWeightedL2Loss of the voted pose against the ICP pseudo-targets
 *      (rslo/core/losses.py:144-197; mask = ones, focal_gamma = 0), the same reduction of the pyramid levels' per-sample
 *      terms (rslo/models/voxel_odom_net.py:743-798; loss_b from rslo_pyramid_l2_fwd), the consistency loss's reduce over
 *      the per-pair terms (losses.py:496-506) and the weighted total (voxel_odom_net.py:324-376).
 *      out5 = (total, T, R, pyramid, C).  Pointers are device pointers; alpha_* point at the modules' log-variance
 *      parameters (they may alias: the shipped configuration uses the SAME modules for pose and pyramid terms). */
#define RSLO_LOSS_TAIL_MAX_LEVELS 8
#define RSLO_LOSS_TAIL_ALPHA_STRIDE 4
typedef struct {
  const float *t_pred, *t_tgt;        /* [B,3] */
  const float *q_pred, *q_tgt;        /* [B,4] (w,x,y,z) */
  const float *alpha_T, *alpha_R;     /* [1] each */
  const float *pyr_loss_b;            /* [L,B,2] or NULL */
  const float *alpha_pT, *alpha_pR;
  const float *pair_loss;             /* [n_pairs] or NULL */
  const float *alpha_C;
  int32_t B, L, n_pairs, reserved;
  float w_T, w_R, w_pT, w_pR;         /* the modules' _loss_weight */
  float c_scale;                      /* (1 - warm_weight) * prediction weight * consistency _loss_weight */
  float level_w[RSLO_LOSS_TAIL_MAX_LEVELS];   /* pyloss_exp_w_base ** (L - l) */
} RsloLossTail;
RSLO_API int rslo_loss_tail_fwd(const RsloLossTail *h_p, float *out5, void *stream);
RSLO_API int rslo_loss_tail_bwd(const RsloLossTail *h_p, const float *grad_out /*[1]*/, float *d_t /*[B,3]*/,
                                float *d_q /*[B,4]*/, float *d_pyr /*[L,B,2]*/, float *d_pair /*[n_pairs]*/,
                                float *d_alpha5 /*(T, R, pT, pR, C): 5 x RSLO_LOSS_TAIL_ALPHA_STRIDE floats*/, void *stream);
/*     d_alpha5: entry i is written to d_alpha5[i * RSLO_LOSS_TAIL_ALPHA_STRIDE] -- 16-byte slots, so that each one can be
 *     handed on as a parameter's gradient tensor (the optimizer's tables take 16-byte aligned gradients).  A module that
 *     serves several terms (the same alpha pointer more than once): its FIRST entry holds the sum of all of them, added in
 *     entry order. */

/* ------------------------------------------------------------------------------------
 * a1 + a2 + a5 in ONE call: voxelization of all clouds of a step (frames x samples) and the complete rulebook chain
 * of a chain-structured sparse encoder -- site hashes, SubM tables, strided-conv output sets + both tables, tile row
 * orders, weight-gradient pair lists -- without a host read.  Replaces, for a whole training step, the DataLoader-side
 * VoxelGenerator.generate calls + merge_second_batch coordinate padding (rslo/data/preprocess.py:493,75-89) and the
 * implicit indice-pair builds of SpMiddleFHDWithCov2_3 (rslo/models/middle.py:119-213, keys subm0 .. dsubm1).
 *
 * Levels: level 0 = the voxel grid (dims0 = sparse_shape, z y x); level l+1 = output of the strided conv l
 * (conv_ks/stride/pad[l]); subm_ks[l] != 0 asks for the SubM table of level l.  Batch index of cloud c in the batched
 * encoder tensor = c (clouds are given frame-major: c = t * clouds_per_frame + b); its index inside its frame = b.
 * Everything lives in ONE caller-owned arena (rslo_plan_encoder_layout gives the offsets; capacities instead of exact
 * sizes: level 0 = min(n_clouds * max_voxels, sum P), level l+1 = capacity of level l unless cap_rows[] says otherwise).
 * Actual sizes are device words in the counts block, copied asynchronously to h_counts (pinned host memory, may be
 * NULL) at the end; the caller reads them after synchronising with an event recorded behind the call:
 *   counts[RSLO_PLAN_CNT_OVERFLOW]      bit l set: level l had more sites than its capacity (rows past it dropped:
 *                                       the plan is unusable, re-plan with larger cap_rows)
 *   counts[RSLO_PLAN_CNT_ROWS + l]      active sites of level l
 *   counts[RSLO_PLAN_CNT_NVOX + c]      voxels of cloud c
 *   counts[RSLO_PLAN_CNT_BOFF + l*(RSLO_PLAN_MAX_CLOUDS+1) + j]   first row of batch element j on level l (j = 0..n)
 * Results equal the stand-alone entry points above bit for bit (same kernels, row counts read from the device).
 * ------------------------------------------------------------------------------------ */
#define RSLO_PLAN_MAX_LEVELS 8
#define RSLO_PLAN_MAX_CLOUDS 64
#define RSLO_PLAN_CNT_OVERFLOW 0
#define RSLO_PLAN_CNT_ROWS 1
#define RSLO_PLAN_CNT_RAW 9
#define RSLO_PLAN_CNT_NVOX 32
#define RSLO_PLAN_CNT_BASE 96
#define RSLO_PLAN_CNT_BOFF 176
#define RSLO_PLAN_CNT_WORDS (176 + RSLO_PLAN_MAX_LEVELS * (RSLO_PLAN_MAX_CLOUDS + 1))
typedef struct {
  int32_t n_levels;
  int32_t dims0[3];                              /* sparse_shape (z, y, x) of level 0 */
  int32_t subm_ks[RSLO_PLAN_MAX_LEVELS][3];      /* 0,0,0: no SubM table on this level */
  int32_t conv_ks[RSLO_PLAN_MAX_LEVELS][3], conv_stride[RSLO_PLAN_MAX_LEVELS][3], conv_pad[RSLO_PLAN_MAX_LEVELS][3];
  int32_t want_pairs, want_orders;               /* want_orders: bit l = a mask-sorted row order for the transposed table of level l's strided conv */
  float range6[6], vsize3[3];                    /* voxelizer geometry (rslo_voxelize) */
  int32_t grid_xyz[3], max_points, max_voxels, n_features;
  int64_t cap_rows[RSLO_PLAN_MAX_LEVELS];        /* 0 = default capacity */
} RsloEncoderSpec;
typedef struct {
  uint64_t total_bytes;
  int32_t dims[RSLO_PLAN_MAX_LEVELS][3];
  int64_t cap_rows[RSLO_PLAN_MAX_LEVELS], hash_cap[RSLO_PLAN_MAX_LEVELS];
  uint64_t counts_off;                           /* int32 [RSLO_PLAN_CNT_WORDS] */
  uint64_t voxels_off, num_points_off;           /* fp32 [cap0, T, F], int32 [cap0]: all clouds, frame-major */
  uint64_t coords_frame_off;                     /* int32 [cap0, 4]: (index inside the frame, z, y, x) */
  uint64_t coords_off[RSLO_PLAN_MAX_LEVELS];     /* int32 [cap_l, 4]: (batch, z, y, x); level 0 = all clouds */
  uint64_t keys_off[RSLO_PLAN_MAX_LEVELS], vals_off[RSLO_PLAN_MAX_LEVELS];
  uint64_t subm_nbr_off[RSLO_PLAN_MAX_LEVELS], subm_pin_off[RSLO_PLAN_MAX_LEVELS], subm_pout_off[RSLO_PLAN_MAX_LEVELS],
      subm_koff_off[RSLO_PLAN_MAX_LEVELS];
  uint64_t conv_nbr_off[RSLO_PLAN_MAX_LEVELS], conv_nbrT_off[RSLO_PLAN_MAX_LEVELS], conv_order_off[RSLO_PLAN_MAX_LEVELS],
      conv_pin_off[RSLO_PLAN_MAX_LEVELS], conv_pout_off[RSLO_PLAN_MAX_LEVELS], conv_koff_off[RSLO_PLAN_MAX_LEVELS];
  int64_t scratch_words;
  uint64_t vox_ws_off, bitmap_off, prefix_off, scan_ws_off, pair_ws_off;
  uint64_t keys_end_off;                                 /* keys_off[0] .. keys_end_off: the keys of all levels (one fill) */
  uint64_t bitmap_level_off[RSLO_PLAN_MAX_LEVELS], bitmap_end_off;   /* output bitmap of conv l, all levels contiguous */
} RsloPlanLayout;
RSLO_API int rslo_plan_encoder_layout(const RsloEncoderSpec *h_spec, int n_clouds, const int64_t *h_n_points,
                                      RsloPlanLayout *h_layout);
RSLO_API int rslo_plan_encoder(const RsloEncoderSpec *h_spec, const RsloPlanLayout *h_layout, int n_clouds,
                               int clouds_per_frame, const float *const *h_points /*n_clouds device pointers [P,F]*/,
                               const int64_t *h_n_points, void *arena, size_t arena_bytes,
                               int32_t *h_counts /*pinned, [RSLO_PLAN_CNT_WORDS] or NULL*/, void *stream);

/*     rslo_plan_encoder_pad_tails (round 5): behind rslo_plan_encoder on the same stream, turns the rows of every table past
 *     its level's device-side count -- up to the level's CAPACITY -- into padding rows: coordinates -1 (rslo_dense_scatter
 *     skips them), neighbour entries -1 (a padding output row gathers nothing), tile orders continued as the identity.  The
 *     encoder's kernels can then be launched for capacity-sized tensors whose shapes do not depend on the scan: what an
 *     inference loop needs to replay ONE hipGraph per arena (evaluate.py:363-408 runs the same forward frame after frame). */
RSLO_API int rslo_plan_encoder_pad_tails(const RsloEncoderSpec *h_spec, const RsloPlanLayout *h_layout, void *arena,
                                         void *stream);

/* wgrad over pair lists: dW[k] = sum_{p in [koff[k],koff[k+1])} in[pairs_in[p]]^T dout[pairs_out[p]];
 * dbias = column sums of dout (all n_out rows).  Deterministic (fixed-order two-stage reduction). */
RSLO_API size_t rslo_spconv_wgrad_pairs_ws_bytes(int64_t n_out, int K, int cin, int cout);
RSLO_API int rslo_spconv_wgrad_pairs(const float *in, int cin, const float *dout, int cout, const int32_t *pairs_in,
                            const int32_t *pairs_out, const int32_t *koff, int64_t n_out, int K, void *ws,
                            size_t ws_bytes, float *dW /*[K,cin,cout]*/, float *dbias /*[cout] or NULL*/,
                            const float *bias_partial /*[n_bias_partial,cout] from rslo_leaky_bwd_colsum, or NULL*/,
                            int n_bias_partial, void *stream);
/* LeakyReLU backward from the saved OUTPUT (sign-preserving): g = dout * (y > 0 ? 1 : slope). */
RSLO_API int rslo_leaky_bwd(const float *y, const float *dout, int64_t n, float slope, float *g, void *stream);
/*     The same plus per-block column sums of g (stage 1 of the bias gradient; cols must divide 1024):
 *     partial [rslo_leaky_bwd_colsum_blocks(rows, cols), cols] is handed to rslo_spconv_wgrad_pairs. */
RSLO_API int64_t rslo_leaky_bwd_colsum_blocks(int64_t rows, int cols);
RSLO_API int rslo_leaky_bwd_colsum(const float *y, const float *dout, int64_t rows, int cols, float slope, float *g,
                                   float *partial, void *stream);

/* a7  Per-frame BatchNorm1d (+ fused LeakyReLU) of the covariance branch (nn.BatchNorm1d at
 *     rslo/models/middle.py:181-198; the reference feeds one frame per call, so statistics are per frame).
 * Rows are grouped by frame: seg_off [S+1] (device).  Training-mode semantics of S consecutive module calls:
 * biased variance for normalisation, running estimates updated in segment order with the unbiased variance.
 * y = act(gamma * (x - mean_s) * invstd_s + beta); save_mean / save_invstd [S,C] feed the backward. */
RSLO_API size_t rslo_segbn_ws_bytes(int S, int64_t max_seg_len, int C);
RSLO_API int rslo_segbn_fwd(const float *x, int C, const int32_t *seg_off, int S, int64_t max_seg_len,
                   const float *gamma, const float *beta, float *running_mean /*or NULL*/,
                   float *running_var /*or NULL*/, float momentum, float eps, float act_slope, void *ws,
                   size_t ws_bytes, float *y, float *save_mean, float *save_invstd, void *stream);
/*     Eval mode (evaluate.py:363-408; the nn.BatchNorm1d layers of rslo/models/middle.py:181-213 with their running statistics):
 *     y = act((x - running_mean) / sqrt(running_var + eps) * gamma + beta) over [n, C] rows in one launch; act_slope 1 = no
 *     activation; gamma / beta may be NULL; n_live_dev (or NULL): device count of the live rows (rows past it stay unwritten). */
RSLO_API int rslo_bn1d_eval_act(const float *x, int64_t n, int C, const float *running_mean, const float *running_var,
                                const float *gamma, const float *beta, float eps, float act_slope, const int32_t *n_live_dev,
                                float *y, void *stream);
/* ws_bytes >= rslo_segbn_ws_bytes(...) + 2*S*C*4 */
RSLO_API int rslo_segbn_bwd(const float *x, const float *y, const float *gy, int C, const int32_t *seg_off, int S,
                   int64_t max_seg_len, const float *gamma, const float *save_mean, const float *save_invstd,
                   float act_slope, void *ws, size_t ws_bytes, float *gx, float *dgamma, float *dbeta,
                   void *stream);

/* a8  SparseConvTensor.dense() + view (middle.py:240-243): [M,C] rows -> [B,C,D,H,W]
 *     (zero-filled here) and its backward gather. */
RSLO_API int rslo_dense_scatter(const float *feat, const int32_t *coords, int64_t M, int C, int B,
                       const int32_t *h_dims3, float *out, void *stream);
RSLO_API int rslo_dense_gather(const float *dense, const int32_t *coords, int64_t M, int C, int B,
                      const int32_t *h_dims3, float *dfeat, void *stream);
/*     The same with the frames of a sample side by side: the batch holds frame t of sample b at index t * (B / frames) + b
 *     and the dense tensor is [B / frames, frames, C, D, H, W] = the channel concatenation the BEV head builds from a
 *     pair's frames (rslo/models/odom_pred.py:170, odom_pred_base.py:305-324), produced without a copy. */
RSLO_API int rslo_dense_scatter_frames(const float *feat, const int32_t *coords, int64_t M, int C, int B, int frames,
                                       const int32_t *h_dims3, float *out, void *stream);
RSLO_API int rslo_dense_gather_frames(const float *dense, const int32_t *coords, int64_t M, int C, int B, int frames,
                                      const int32_t *h_dims3, float *dfeat, void *stream);
/*     Per-cell sums over each of the G channel groups of a BEV tensor [B, G*Cg, HW] -> [B, G, HW] in one pass: feeds the
 *     occupancy masks (odom_pred.py:165-168; voxel_odom_net.py:519-527) and the logged channel means. */
RSLO_API int rslo_bev_channel_sums(const float *in, int B, int G, int Cg, int64_t HW, float *out, void *stream);
/*     The same pass also writing the occupancy of channel group 0 in the three forms the head uses (odom_pred.py:165-168 and
 *     rslo/layers/confidence.py:26-34): mask_f [B,HW] float 0/1, mask_b [B,HW] bytes 0/1, outside_b = !mask_b.  Any may be NULL. */
RSLO_API int rslo_bev_channel_sums_masks(const float *in, int B, int G, int Cg, int64_t HW, float *out, float *mask_f,
                                         unsigned char *mask_b, unsigned char *outside_b, void *stream);
/*     The logged extras from those sums [B,T,HW] in two launches (voxel_odom_net.py:455-464; ws: rslo_bev_display_ws_bytes(T) bytes of partial extrema): mask [B,HW] = (sum over t) != 0;
 *     disp [T,B,HW] = the per-frame channel mean sums / Cg, min-max normalised over the frame's batch
 *     ((d - min) / (max - min + 1e-12)) -- `middle_feature` and `feature_mask` of the training forward. */
RSLO_API size_t rslo_bev_display_ws_bytes(int T);
RSLO_API int rslo_bev_display(const float *sums, int B, int T, int Cg, int64_t HW, float *mask, float *disp, void *ws,
                              size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * a17  Chamfer nearest neighbour.  Replaces cd.forward_cuda_one_direction /
 *      cd.backward_cuda_one_direction (thirdparty/chamfer_distance/chamfer_distance.cpp:62-76,
 *      94-113; kernels chamfer_distance.cu:6-137,177-206).  Arithmetic follows the CPU
 *      nnsearch (chamfer_distance.cpp:116-144): fp32 products and sums without contraction,
 *      strict '<' so the lowest index wins ties -- bit-exact dist and idx vs that path.
 * ------------------------------------------------------------------------------------ */
RSLO_API size_t rslo_chamfer_ws_bytes(int B, int N, int M);
RSLO_API int rslo_chamfer_nn(const float *xyz1 /*[B,N,3]*/, const float *xyz2 /*[B,M,3]*/, int B, int N, int M,
                    float *dist /*[B,N]*/, int32_t *idx /*[B,N]*/, void *ws, size_t ws_bytes,
                    void *stream);
/* Ragged batch: pair b uses only its first ncnt[b] queries and mcnt[b] targets of the padded [B,N,3]/[B,M,3]
 * arrays (device int32 [B]; NULL = all).  Padding queries get dist = +inf, idx = 0. */
RSLO_API int rslo_chamfer_nn_ragged(const float *xyz1, const float *xyz2, int B, int N, int M, const int32_t *ncnt,
                           const int32_t *mcnt, float *dist, int32_t *idx, void *ws, size_t ws_bytes,
                           void *stream);
/*     Exact search with spatial pruning (Morton bucket sort of both clouds, 64-point target tiles with bounding
 *     boxes, box lower bound in the distance's own operation order => bit-identical to the exhaustive scan).
 *     rslo_chamfer_nn / _ragged dispatch to it for N >= 1024 and M >= 2048 (tuning switch `chamfer`: 1 = exhaustive, 2 = pruned);
 *     rslo_chamfer_ws_bytes covers both. */
RSLO_API int rslo_chamfer_brute_nn(const float *xyz1, const float *xyz2, int B, int N, int M, const int32_t *ncnt,
                                   const int32_t *mcnt, float *dist, int32_t *idx, void *ws, size_t ws_bytes,
                                   void *stream);   /* the exhaustive scan, explicitly */
RSLO_API size_t rslo_chamfer_grid_ws_bytes(int B, int N, int M);
RSLO_API int rslo_chamfer_grid_nn(const float *xyz1, const float *xyz2, int B, int N, int M, const int32_t *ncnt,
                                  const int32_t *mcnt, float *dist, int32_t *idx, void *ws, size_t ws_bytes,
                                  void *stream);
RSLO_API int rslo_chamfer_grad(const float *xyz1, const float *xyz2, int B, int N, int M,
                      const float *graddist1, const int32_t *idx1, float *gradxyz1 /*[B,N,3]*/,
                      float *gradxyz2 /*[B,M,3]*/, void *stream);

/* ------------------------------------------------------------------------------------
 * a18-a20  Consistency loss (rslo/core/losses.py:337-507, Aleat5_1ChamferL2NormalWeightedALLSVDLoss).
 * All arrays are per frame pair b in [0,B): p1 [B,N,3] source voxel means, n1 [B,N,3] their normals,
 * tgt [B,M,3] posed target means, cov1 [B,N,7] / cov2 [B,M,7] covariance parameters
 * (3 eigenvalue increments + quaternion read as x,y,z,w -- losses.py:348-363), idx/dist [B,N] the
 * chamfer association, thr [B] = max(kth(dist), 1) (losses.py:326-334), Rd [B,9] the detached pose.
 *
 * cov_residual_fwd: loss[b] = mean_roi(d^T sigma^-1 d) + reg * mean_roi(0.5 log det sigma),
 *                   sigma = S1 + Rd S2[idx] Rd^T, d = p1 - tgt[idx], roi = dist < thr (losses.py:396-435;
 *                   closed-form 3x3 inverse/det instead of torch.inverse/torch.det over ~30k matrices).
 * cov_residual_bwd: gradients of sum_b gloss[b]*loss[b] w.r.t. tgt, cov1, cov2 (and p1 if gp1 != NULL).
 * icp_step        : one SVDHead refinement (rslo/layers/svd.py:14-64 called from losses.py:451-465):
 *                   weighted Kabsch over the ROI with w = cos(n1, tgt[idx]-p1)^2, unweighted centroids,
 *                   reflection fix, returns the inverse motion and composes res_r = R res_r,
 *                   res_t = R res_t + t in place.  No host round trip (the reference branches on det(R)).
 * transform_points: out = R x + t per pair (losses.py:471-473).
 * ------------------------------------------------------------------------------------ */
RSLO_API size_t rslo_cov_residual_ws_bytes(int B, int N);
RSLO_API int rslo_cov_residual_fwd(const float *p1, const float *tgt, const float *cov1, const float *cov2,
                          const int32_t *idx, const float *dist, const float *thr, const float *Rd, int B, int N,
                          int M, float reg_weight, void *ws, size_t ws_bytes, float *loss /*[B]*/,
                          float *cnt /*[B]*/, void *stream);
RSLO_API int rslo_cov_residual_bwd(const float *p1, const float *tgt, const float *cov1, const float *cov2,
                          const int32_t *idx, const float *dist, const float *thr, const float *Rd,
                          const float *gloss /*[B]*/, const float *cnt /*[B]*/, int B, int N, int M,
                          float reg_weight, float *gp1 /*[B,N,3] or NULL*/, float *gtgt /*[B,M,3]*/,
                          float *gcov1 /*[B,N,7]*/, float *gcov2 /*[B,M,7]*/, void *ws, size_t ws_bytes, void *stream);
/*     The gradients that land on the PARTNER rows (gtgt, gcov2: several sources may share a partner) are added in ascending
 *     source order: every source leaves its ten values and the key (partner row, source row), the keys are sorted (radix
 *     sort over B N 64-bit keys) and each partner's run is added by one thread (runs of up to 32 sources) or one wave
 *     (fixed lane assignment + fixed tree) -- bit-reproducible from run to run, where per-run atomics (rounds 1-4, deleted in
 *     round 6) made every training step unique.  ws: rslo_cov_residual_bwd_ws_bytes(B, N, M). */
RSLO_API size_t rslo_cov_residual_bwd_ws_bytes(int B, int N, int M);
RSLO_API size_t rslo_icp_ws_bytes(int B, int N);
RSLO_API int rslo_icp_step(const float *p1, const float *n1, const float *tgt, const int32_t *idx, const float *dist,
                  const float *thr, int B, int N, int M, void *ws, size_t ws_bytes, float *res_r /*[B,9] in/out*/,
                  float *res_t /*[B,3] in/out*/, float *step_R /*[B,9] or NULL*/, float *step_t /*[B,3] or NULL*/,
                  void *stream);
/*     _first: the first refinement of a pair -- res_r / res_t are outputs only (the running motion starts as the identity,
 *     losses.py:452-453), the caller does not fill them. */
RSLO_API int rslo_icp_step_first(const float *p1, const float *n1, const float *tgt, const int32_t *idx, const float *dist,
                  const float *thr, int B, int N, int M, void *ws, size_t ws_bytes, float *res_r /*[B,9] out*/,
                  float *res_t /*[B,3] out*/, float *step_R /*[B,9] or NULL*/, float *step_t /*[B,3] or NULL*/,
                  void *stream);
/* a18  ROI threshold (rslo/core/losses.py:326-334): thr[b] = max(k-th smallest of dist[b][0..counts[b]), 1) with
 *      k = 1 + int(counts[b] * ratio); counts NULL = N everywhere.  Exact (radix select), no sort. */
RSLO_API int rslo_roi_threshold(const float *dist, int B, int N, const int32_t *counts, double ratio, float *thr,
                                void *stream);
RSLO_API int rslo_transform_points(const float *x, const float *R, const float *t, int B, int M, float *out,
                          void *stream);
/*     The same rigid map on rows of `row_stride` floats (t may be NULL), and its pose gradient
 *     dR[b] = sum_j g[b][j] x[b][j]^T, dt[b] = sum_j g[b][j]  (p2_moved / n2_moved of the loss assembly,
 *     rslo/models/voxel_odom_net.py:668-676).  done: int32 [B] counters, zero on entry, left zero. */
RSLO_API int rslo_transform_rows(const float *x, int row_stride, const float *R, const float *t, int B, int M,
                                 float *out, void *stream);
RSLO_API size_t rslo_transform_rows_bwd_ws_bytes(int B, int M);
RSLO_API int rslo_transform_rows_bwd(const float *x, int row_stride, const float *gout, int B, int M, void *ws,
                                     size_t ws_bytes, int32_t *done, float *dR, float *dt, void *stream);

/* a21  pyramid supervision of the per-cell local transformation maps, all levels in one launch
 *      (rslo/models/voxel_odom_net.py:706-760 pyramid part of create_loss; target map of gen_tq_maps
 *      voxel_odom_net.py:575-600 = generate_pointwise_local_transformation_tch rslo/data/dataset.py:118-168,
 *      nearest-resampled per level by F.interpolate; per-sample masked L2 of AdaptiveWeightedL2Loss
 *      rslo/core/losses.py:144-197).
 *      level l: pred [B,7,h,w] (t_l xyz, q wxyz), mask [B,Cm,h,w] (channel 0 weighs the translation part,
 *      channel Cm-1 the rotation part).  tq [B,7] = global pose targets (t, q wxyz).  The target of a cell is
 *      t_l = R(q)^-1 (t_g - x) + x with x the centre of the nearest cell of the H0 x W0 map.
 *      loss_b [L,B,2] = sum(diff^2 mask) / (n_ch sum(mask) + 1e-12) for (T, R);  den [L,B,2] = n_ch sum(mask).
 *      done: [L*B] int32 counters, zero on entry, left zero on exit.  bwd writes levels[l].dpred [B,7,h,w]. */
typedef struct {
  const float *pred;
  const float *mask;
  float *dpred;          /* bwd only */
  int32_t h, w, mask_channels;
} RsloPyramidLevel;
RSLO_API size_t rslo_pyramid_l2_ws_bytes(const RsloPyramidLevel *h_levels, int n_levels, int B);
RSLO_API int rslo_pyramid_l2_fwd(const RsloPyramidLevel *h_levels, int n_levels, int B, const float *tq, int H0,
                                 int W0, const float *h_origin3, const float *h_vsize3, void *ws, size_t ws_bytes,
                                 int32_t *done, float *loss_b, float *den, void *stream);
RSLO_API int rslo_pyramid_l2_bwd(const RsloPyramidLevel *h_levels, int n_levels, int B, const float *tq, int H0,
                                 int W0, const float *h_origin3, const float *h_vsize3, const float *grad_loss_b,
                                 const float *den, void *stream);

/* a20  ragged batch assembly of the consistency loss (rslo/models/voxel_odom_net.py:629-651: each sample's frames are
 *      truncated to the shortest; all samples of one frame are gathered into ONE zero-padded batch):
 *      out[b, r, :] = r < len[b] ? src[off[b] + r, :] : 0   (src [N,C], out [B,Lmax,C]);  bwd is the inverse copy,
 *      rows of src outside every [off[b], off[b]+len[b]) get 0.  off/len: int32 [B] on the device. */
RSLO_API int rslo_pad_rows_fwd(const float *src, int64_t N, int C, const int32_t *off, const int32_t *len, int B,
                               int Lmax, float *out, void *stream);
RSLO_API int rslo_pad_rows_bwd(const float *dout, int64_t N, int C, const int32_t *off, const int32_t *len, int B,
                               int Lmax, float *dsrc, void *stream);
/*      rslo_pair_rows_fwd: the three operands of one frame in one launch (voxel_odom_net.py:630-660 selects the xyz and
 *      normal columns of the voxel features, joins them with the covariance head's rows and pads): feats [N,F] (F = 7:
 *      xyz, intensity, normal -- or F = 6: xyz, normal), conf [N,Cc] -> xyz [B,Lmax,3], nrm [B,Lmax,3], cov [B,Lmax,Cc],
 *      zero padded as above.  Only conf carries a gradient: rslo_pad_rows_bwd with C = Cc is its backward. */
RSLO_API int rslo_pair_rows_fwd(const float *feats, int64_t N, int F, const float *conf, int Cc, const int32_t *off,
                                const int32_t *len, int B, int Lmax, float *xyz, float *nrm, float *cov, void *stream);

/* a10-a12  training-mode (Sync)BatchNorm2d of the dense head fused with activation and residual add, NCHW fp32
 *      (apex.parallel.SyncBatchNorm via rslo/layers/SparseConv.py:96-113; rslo/models/custom_resnet_spc.py:224-298).
 *      stats [2C+1] doubles = per-channel sum, sum of squares, element count: the caller all-reduces them across ranks
 *      between rslo_bn2d_stats and rslo_bn2d_apply (nothing to do on one rank).  apply: y = act(gamma (x-mean) invstd +
 *      beta (+ res)), running statistics updated with `momentum` (unbiased variance), mean / invstd saved.
 *      backward: rslo_bn2d_bwd_reduce gives red [2C] = sum g, sum g x^ (g = dy act'(y); all-reduce across ranks) and
 *      the LOCAL dgamma / dbeta; rslo_bn2d_bwd_apply gives dx (and dres = g).  done: int32 [C], zero on entry / exit. */
RSLO_API size_t rslo_bn2d_ws_bytes(int N, int C, int HW);
RSLO_API int rslo_bn2d_stats(const float *x, int N, int C, int HW, void *ws, size_t ws_bytes, int32_t *done,
                             double *stats, void *stream);
RSLO_API int rslo_bn2d_apply(const float *x, const float *res, const double *stats, const float *gamma,
                             const float *beta, int N, int C, int HW, float eps, float momentum, float act_slope,
                             float *running_mean, float *running_var, float *save_mean, float *save_invstd, float *y,
                             void *stream);
RSLO_API int rslo_bn2d_bwd_reduce(const float *dy, const float *y, const float *x, const float *save_mean,
                                  const float *save_invstd, int N, int C, int HW, float act_slope, int has_act,
                                  void *ws, size_t ws_bytes, int32_t *done, double *red, float *dgamma, float *dbeta,
                                  void *stream);
RSLO_API int rslo_bn2d_bwd_apply(const float *dy, const float *y, const float *x, const float *gamma,
                                 const float *save_mean, const float *save_invstd, const double *red, double count,
                                 const double *count_dev /* or NULL: element count over all ranks on the device, e.g.
                                 &stats[2C] after the forward all-reduce (uneven per-rank batches); overrides count */,
                                 int N, int C, int HW, float act_slope, int has_act, float *dx, float *dres,
                                 void *stream);

/*      Single-rank forms (no statistics exchange): statistics slices + apply in two launches, the apply kernels add
 *      the slice partials themselves.  Same results as stats -> apply / bwd_reduce -> bwd_apply with one rank. */
RSLO_API int rslo_bn2d_fwd_local(const float *x, const float *res, const float *gamma, const float *beta, int N, int C,
                                 int HW, float eps, float momentum, float act_slope, float *running_mean,
                                 float *running_var, float *save_mean, float *save_invstd, float *y, void *ws,
                                 size_t ws_bytes, void *stream);
RSLO_API int rslo_bn2d_bwd_local(const float *dy, const float *y, const float *x, const float *gamma,
                                 const float *save_mean, const float *save_invstd, int N, int C, int HW,
                                 float act_slope, int has_act, float *dx, float *dres, float *dgamma, float *dbeta,
                                 void *ws, size_t ws_bytes, void *stream);

/* a13 / a14  local -> global transformation of every BEV cell + confidence-weighted ego-motion vote
 *      (from_pointwise_local_transformation_tch rslo/data/dataset.py:121-208, rotate_vec_by_q
 *       rslo/utils/pose_utils.py:130-142, aggregate_tq rslo/models/odom_pred.py:347-357).
 *      tq_map [B,7,H,W] (t_l xyz, q wxyz), t_conf / r_conf [B,1,H,W]  ->  tq_map_g [B,7,H,W], odom [B,7] (t, q),
 *      sums [B,2] = (sum t_conf + 1e-12, sum r_conf + 1e-12).  bwd: gradient of odom -> d tq_map, d t_conf, d r_conf
 *      (tq_map_g carries no gradient on this path).  done: int32 [B], zero on entry / exit. */
RSLO_API size_t rslo_vote_ws_bytes(int B, int H, int W);
RSLO_API int rslo_vote_fwd(const float *tq_map, const float *t_conf, const float *r_conf, int B, int H, int W,
                           const float *h_origin3, const float *h_vsize3, void *ws, size_t ws_bytes, int32_t *done,
                           float *tq_map_g, float *odom, float *sums, void *stream);
RSLO_API int rslo_vote_bwd(const float *tq_map, const float *t_conf, const float *r_conf, int B, int H, int W,
                           const float *h_origin3, const float *h_vsize3, const float *odom, const float *sums,
                           const float *g_odom, float *d_tq_map, float *d_t_conf, float *d_r_conf, void *stream);

/* a10 - a12  weight gradient of the BEV head's dense 3x3 convolutions (torch.nn.Conv2d / MaskConv.conv1 built in
 *      rslo/models/odom_pred.py:65-134,398-426 and rslo/models/custom_resnet_spc.py:224-298; in the reference the
 *      gradient comes from cuDNN through autograd).  NCHW fp32, kernel 3x3, padding 1, stride 1 or 2, cin % 16 == 0,
 *      cout % 32 == 0:  dW [cout,cin,3,3] = sum over batch and output pixels of dout [B,cout,Ho,Wo] x shifted
 *      in [B,cin,H,W].  bf16 matrix cores on exactly split fp32 operands, fixed summation order (no atomics).
 *      rslo_conv2d_wgrad_supported returns 1 for shapes the kernel takes (others stay on the library path).
 *      dbias [cout] (optional, stride 1 only): the bias gradient = per-channel sum of dout, from the same pass. */
RSLO_API int rslo_conv2d_wgrad_supported(int cin, int cout, int H, int W, int stride);
RSLO_API size_t rslo_conv2d_wgrad_ws_bytes(int B, int cin, int cout, int H, int W, int stride);
RSLO_API int rslo_conv2d_wgrad(const float *in, const float *dout, int B, int cin, int cout, int H, int W, int stride,
                               float *dW, float *dbias, void *ws, size_t ws_bytes, void *stream);

/* a10 - a12  forward and data gradient of the same layers (3x3, stride 1, padding 1, NCHW fp32; channels % 32 == 0):
 *      out[b][m] = bias[m] + sum_k A[m][k] (*) in[b][k].  rslo_conv2d_wsplit turns the layer's weight W [cout,cin,3,3]
 *      into split-bf16 MFMA operands: transpose = 0 for the forward pass (m = cout, k = cin), transpose = 1 for the
 *      data gradient (m = cin, k = cout, taps flipped; pass dout as `in`).  rslo_conv2d_fwd's cin / cout are the
 *      contraction / produced channel counts of THAT call.  bias may be NULL. */
RSLO_API int rslo_conv2d_fwd_supported(int cin, int cout, int H, int W);
RSLO_API size_t rslo_conv2d_wsplit_bytes(int cin, int cout);
RSLO_API int rslo_conv2d_wsplit(const float *W, int cin, int cout, int transpose, void *Ws, void *stream);
/*      rslo_conv2d_wsplit_many: the same split for all layers of a model, both orientations, in ONE launch (the weights
 *      are constant between optimizer steps): desc_dev = device array of n_layers descriptors, max_weight_elems =
 *      max over layers of cin * cout * 9. */
typedef struct {
  const float *W;   /* [cout,cin,3,3] */
  void *ws_fwd;     /* rslo_conv2d_wsplit_bytes(cin, cout) bytes, transpose = 0 */
  void *ws_dgrad;   /* same size, transpose = 1 */
  int32_t cin, cout;
  int32_t ntap;     /* 9 (or 0): 3x3 weight; 1: 1x1 weight (operand of rslo_conv2d_fwd_s2 / _dgrad_s2 with ksize 1) */
  int32_t reserved;
} RsloConv2dSplitDesc;
RSLO_API int rslo_conv2d_wsplit_many(const RsloConv2dSplitDesc *desc_dev, int n_layers, int64_t max_weight_elems,
                                     void *stream);
RSLO_API int rslo_conv2d_fwd(const float *in, const void *Ws, const float *bias, int B, int cin, int cout, int H, int W,
                             float *out, void *stream);
/*      rslo_conv2d_fwd_add: out = conv + bias + res, res [B,cout,H,W] read in the epilogue -- the data gradient of a
 *      BasicBlock's first convolution plus the gradient of its identity branch (custom_resnet_spc.py:74-96 backward),
 *      the same bits as the convolution followed by a separate element-wise add.  res must not alias out. */
RSLO_API int rslo_conv2d_fwd_add(const float *in, const void *Ws, const float *bias, const float *res, int B, int cin,
                                 int cout, int H, int W, float *out, void *stream);
/*      Operand planes (round 5).  A dense activation [B,C,H,W] fp32 (C % 8 == 0) stored ONCE, by its producer, in the
 *      form the convolution's MFMA operands have:  P[b][C/8][3 = hi|mid|lo][H][W][8] bf16 bit patterns, hi + mid + lo ==
 *      the fp32 value exactly (the split k_conv2d_fwd performs per workgroup while staging).  rslo_opl_bytes: size of P;
 *      rslo_opl_from_nchw: the stand-alone producer; the BatchNorm apply entry points with a `planes` argument write the
 *      same layout from their epilogue.  rslo_conv2d_fwd_p = rslo_conv2d_fwd / _fwd_add reading P instead of the fp32
 *      tensor (Ws with transpose = 1 and the planes of dout: the data gradient): same arithmetic, bit-identical
 *      results, no operand split and no strided loads in the convolution.  Replaces the same reference calls:
 *      nn.Conv2d inside rslo/layers/MaskConv.py:30-63 as used by custom_resnet_spc.py:224-298 and
 *      rslo/models/odom_pred_base.py:155-207. */
RSLO_API size_t rslo_opl_bytes(int B, int C, int H, int W);
RSLO_API int rslo_opl_from_nchw(const float *in, int B, int C, int H, int W, void *planes, void *stream);
RSLO_API int rslo_conv2d_fwd_p_supported(int cin, int cout, int H, int W);
RSLO_API int rslo_conv2d_fwd_p(const void *planes, const void *Ws, const float *bias, const float *res /* or NULL */,
                               int B, int cin, int cout, int H, int W, float *out, void *stream);
/*      The stride-2 layers of the BEV encoder (first 3x3 convolution and 1x1 downsample of every stage,
 *      rslo/models/odom_pred.py:398-426): ksize 3 (padding 1) or 1 (padding 0), no bias (the reference builds them
 *      bias-free in front of a BatchNorm).  in [B,cin,H,W] -> out [B,cout,Ho,Wo], Ho = (H - 1) / 2 + 1.
 *      rslo_conv2d_wsplit_k: the split operand of a [cout,cin,ksize,ksize] weight (transpose = 1: data gradient);
 *      rslo_conv2d_dgrad_s2 writes every element of din [B,cin,H,W] (per output parity class: no zero-stuffing,
 *      no atomics). */
RSLO_API int rslo_conv2d_s2_supported(int cin, int cout, int ksize);
RSLO_API int rslo_conv2d_wsplit_k(const float *W, int cin, int cout, int ksize, int transpose, void *Ws, void *stream);
RSLO_API int rslo_conv2d_fwd_s2(const float *in, const void *Ws, int B, int cin, int cout, int H, int W, int ksize,
                                float *out, void *stream);
RSLO_API int rslo_conv2d_dgrad_s2(const float *dout, const void *Ws, int B, int cin, int cout, int H, int W, int ksize,
                                  float *din, void *stream);
/*      din = data gradient + res (res [B,cin,H,W]): the input gradients of the two branches of a stride-2 BasicBlock
 *      (custom_resnet_spc.py:74-96) joined in the second branch's epilogue, same bits as the add.  ksize 1 only: res may
 *      BE din (in place) -- the 1x1 gradient lands on the pixels (2y, 2x), the other three quarters of din are not touched. */
RSLO_API int rslo_conv2d_dgrad_s2_add(const float *dout, const void *Ws, const float *res, int B, int cin, int cout, int H,
                                      int W, int ksize, float *din, void *stream);
/*      C4 (bf16 operands, fp32 accumulation and storage): the same kernels issuing only the product of the
 *      round-to-nearest bf16 values of activations and weights (1 MFMA instead of 6).  Ws is the operand block of
 *      rslo_conv2d_wsplit (its first plane IS the bf16-rounded weight); the stride-2 weight gradient keeps the split form. */
RSLO_API int rslo_conv2d_fwd_bf16(const float *in, const void *Ws, const float *bias, int B, int cin, int cout, int H,
                                  int W, float *out, void *stream);
/*      Weight gradient of the 1x1 / stride-2 / padding-0 downsample layers (custom_resnet_spc.py:224-260 `downsample`):
 *      in [B,cin,H,W], dout [B,cout,Ho,Wo] -> dW [cout,cin,1,1]; the centre tap of the stride-2 3x3 kernel, same fixed-order
 *      slab reduction (bit-reproducible).  cin % 16 == 0, cout % 32 == 0, W >= 15. */
RSLO_API int rslo_conv1x1s2_wgrad_supported(int cin, int cout, int H, int W);
RSLO_API size_t rslo_conv1x1s2_wgrad_ws_bytes(int B, int cin, int cout, int H, int W);
RSLO_API int rslo_conv1x1s2_wgrad(const float *in, const float *dout, int B, int cin, int cout, int H, int W, float *dW,
                                  void *ws, size_t ws_bytes, void *stream);
RSLO_API int rslo_conv2d_fwd_add_bf16(const float *in, const void *Ws, const float *bias, const float *res, int B,
                                      int cin, int cout, int H, int W, float *out, void *stream);
RSLO_API int rslo_conv2d_wgrad_bf16(const float *in, const float *dout, int B, int cin, int cout, int H, int W,
                                    int stride, float *dW, float *dbias, void *ws, size_t ws_bytes, void *stream);

/* a11 / a12  the 1x1 output convolutions of the head (torch.nn.Conv2d(c, 7, 1) / (32, 1, 1): rslo/models/odom_pred.py:71,
 *      rslo/models/odom_pred_base.py:22-24,107-109), cout <= 8, NCHW fp32, HW = H * W:
 *      forward (+ bias), data gradient, weight gradient (+ bias gradient from the same pass, fixed summation order). */
RSLO_API int rslo_conv1x1_supported(int cin, int cout);
RSLO_API int rslo_conv1x1_fwd(const float *x, const float *W, const float *bias, int B, int cin, int cout, int HW,
                              float *out, void *stream);
RSLO_API int rslo_conv1x1_dgrad(const float *dy, const float *W, int B, int cin, int cout, int HW, float *dx, void *stream);
RSLO_API size_t rslo_conv1x1_wgrad_ws_bytes(int B, int cin, int cout, int HW);
RSLO_API int rslo_conv1x1_wgrad(const float *x, const float *dy, int B, int cin, int cout, int HW, float *dW, float *dbias,
                                void *ws, size_t ws_bytes, void *stream);

/* a16 / a20  per-pair pose algebra of the loss assembly (one thread per frame pair):
 *      rslo_quat_to_rot: q (w,x,y,z) -> R [B,9] with kornia 0.4.0 semantics (normalise with eps 1e-12 first;
 *      rslo/models/voxel_odom_net.py:675) and its backward;  rslo_pose_targets: pseudo-targets of the ICP refinement
 *      (voxel_odom_net.py:709-735): q* = sign-fixed (w,x,y,z) quaternion of res_R R_pred (kornia's four-branch
 *      matrix -> quaternion, eps 1e-8), t* = res_R T_pred + res_t. */
RSLO_API int rslo_quat_to_rot(const float *q_wxyz, int B, float *R, void *stream);
RSLO_API int rslo_quat_to_rot_bwd(const float *q_wxyz, const float *gR, int B, float *gq, void *stream);
RSLO_API int rslo_pose_targets(const float *res_r, const float *res_t, const float *R_pred, const float *T_pred, int B,
                               float *rot_targets_wxyz, float *trans_targets, void *stream);
/*      _tq: additionally the rows (t*, q*) [B,7] the pyramid supervision reads (voxel_odom_net.py:747), NULL = skip. */
RSLO_API int rslo_pose_targets_tq(const float *res_r, const float *res_t, const float *R_pred, const float *T_pred, int B,
                                  float *rot_targets_wxyz, float *trans_targets, float *tq, void *stream);

/* f1   optimizer step of the training driver: global gradient-norm clipping (train_hdf5.py:671,
 *      torch.nn.utils.clip_grad_norm_) and torch.optim.Adam under the fastai OptimWrapper's decoupled weight decay
 *      (rslo/torchplus/train/fastai_optim.py:176-187; 8 parameter groups, rslo/builder/optimizer_builder.py:49-66) over
 *      all parameter tensors at once.  tensors_dev: device array, one entry per trainable tensor (grad = NULL: no
 *      gradient this step -- excluded from the norm, decayed but not stepped, like the wrapper does);
 *      chunks_dev: device array cutting the tensors into pieces of <= 4096 elements (offsets multiples of 4), one
 *      workgroup each.  State lives in the caller's tensors (exp_avg, exp_avg_sq, step = torch.optim.Adam's).
 *      rslo_opt_clip_grad_norm: total_norm[0] = || all grads ||_2 (device scalar, no host read); grads are scaled by
 *      max_norm / (total_norm + 1e-6) when that is < 1.  partial_ws: n_chunks doubles.
 *      rslo_opt_adam_step: p *= 1 - wd lr; m, v moment updates; p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
 *      with t = step (the count AFTER this update, >= 1); writes t into every tensor's step scalar. */
typedef struct {
  float *param, *grad, *exp_avg, *exp_avg_sq, *step;   /* step may be NULL */
  int32_t group;                                       /* index into RsloOptHyper.group */
  int32_t step_offset;  /* this tensor's own step count = `step` + step_offset (torch.optim.Adam counts per tensor: one
                           that got its first gradient late, or skipped steps, lags the others) */
} RsloOptTensor;
typedef struct {
  int32_t tensor, count;
  int64_t offset;
} RsloOptChunk;
typedef struct {
  double lr, beta1, beta2, eps, weight_decay;          /* weight_decay = the wrapper's decoupled wd (0: none) */
} RsloOptGroup;
#define RSLO_OPT_MAX_GROUPS 16
typedef struct {
  RsloOptGroup group[RSLO_OPT_MAX_GROUPS];
} RsloOptHyper;
RSLO_API int rslo_opt_clip_grad_norm(const RsloOptTensor *tensors_dev, const RsloOptChunk *chunks_dev, int n_chunks,
                                     float max_norm, double *partial_ws, float *total_norm, void *stream);
RSLO_API int rslo_opt_adam_step(const RsloOptTensor *tensors_dev, const RsloOptChunk *chunks_dev, int n_chunks,
                                const RsloOptHyper *hyper, float step, void *stream);

/* ---- a6 / a10 second stage (round 6): all weight-gradient reduces of a backward pass in ONE launch.
 *      Every weight-gradient entry point (rslo_conv2d_wgrad[_bf16], rslo_conv1x1s2_wgrad, rslo_spconv_wgrad_pairs[_bf16]) is a
 *      kernel that leaves slab / chunk partials in the caller's workspace + a small kernel that adds them in a fixed order.  While
 *      a sink is installed (rslo_wgrad_reduce_defer(sink, capacity, &count)) they append a descriptor of that second kernel to
 *      the caller's array instead of launching it (count is advanced; a full sink: launched at once, as without one);
 *      rslo_wgrad_reduce_defer(NULL, 0, NULL) removes the sink (the sink is per calling THREAD).  rslo_wgrad_reduce_many(reduces, n, stream) then runs the n
 *      reduces as one grid -- the same block bodies, the same bits.  The caller keeps every workspace / bias-partial buffer a
 *      descriptor points to alive and unmodified until that launch, and orders it behind the kernels that wrote them (same
 *      stream or an event).  Nothing of the reference reads a gradient before the pass is over (train_hdf5.py:663-672). */
typedef struct RsloWgradReduce {
  int32_t kind;               /* 0: dense slab partials (conv2d.hip); 2: sparse pair-chunk partials (spconv.hip) */
  int32_t n_blocks;           /* workgroups of 256 threads */
  const void *ws;
  void *dW;
  const void *aux;            /* bias-gradient partial rows */
  void *dbias;
  const void *koff;           /* sparse: pair offsets per kernel offset */
  int32_t p[8];               /* shape parameters, written by the launch code */
} RsloWgradReduce;
RSLO_API int rslo_wgrad_reduce_defer(RsloWgradReduce *sink, int capacity, int *count);
RSLO_API int rslo_wgrad_reduce_many(const RsloWgradReduce *reduces, int n, void *stream);

/* ---- a22 (SyncBN statistics): sum of a few hundred doubles over the ranks of ONE node, as a kernel on the caller's stream
 *      Replaces the per-layer all-reduce of apex SyncBatchNorm (rslo/layers/SparseConv.py:96-132 -> apex
 *      sync_batchnorm: 45 layers x 2 directions per step, train_hdf5.py:463) between rslo_bn2d_stats and rslo_bn2d_apply
 *      (and rslo_bn2d_bwd_reduce / _bwd_apply).  Every rank owns a slice (4 slots of {flag, payload}) the others can read;
 *      rslo_peer_allreduce_f64 writes the own payload + flag, waits for every peer's flag of the same exchange number, and
 *      sums the payloads in rank order (identical bits on every rank) into t, in place.  All ranks must issue the same
 *      sequence of exchanges, and all exchanges of one comm go to ONE stream (slot reuse relies on their order).  A peer that does not arrive within the timeout (default 600 s, rslo_peer_set_timeout_ms) poisons t with NaN and is
 *      reported by rslo_peer_status -- the kernel never hangs the GPU.
 *      Transports:  host   = one POSIX shared-memory segment `name` (every rank passes the same name) registered with the
 *                            HIP runtime: any GPUs of one host, also several ranks on one GPU;
 *                   device = each rank's slice in its own HBM, exported as an IPC handle of rslo_peer_ipc_handle_bytes()
 *                            bytes (begin), the handles of ALL ranks in rank order handed back (finish): peers read it
 *                            over xGMI.  world <= 16, max_n <= 1024. */
RSLO_API int rslo_peer_create_host(const char *name, int rank, int world, int max_n, void **comm_out);
RSLO_API int rslo_peer_host_unlink(void *comm);   /* after the caller's barrier: drop the segment's name (mappings stay) */
RSLO_API int rslo_peer_ipc_handle_bytes(void);
RSLO_API int rslo_peer_create_device_begin(int rank, int world, int max_n, void **comm_out, void *handle_out);
RSLO_API int rslo_peer_create_device_finish(void *comm, const void *all_handles);
RSLO_API int rslo_peer_set_timeout_ms(void *comm, int ms);
RSLO_API int rslo_peer_allreduce_f64(void *comm, double *t, int n, void *stream);
RSLO_API unsigned long long rslo_peer_status(void *comm, int *peer);
/*      Diagnostics for the first real N > 1 run: microseconds each of the most recent exchanges (<= 4096) spent spinning for its
 *      slowest peer, oldest first (arrival skew + transport latency); returns the sample count. */
RSLO_API int rslo_peer_wait_samples(void *comm, float *out_us, int max_out);
/*      Round 5: SyncBN of the maps a workgroup holds in registers (N*HW <= 17408 values per channel, C <= 512) in ONE launch
 *      per direction on any number of ranks.  The workgroup of channel c publishes its sums in the comm's slice, meets the
 *      workgroups of channel c on the other ranks (per-channel flags; sums in rank order: identical bits everywhere) and
 *      applies from its registers -- no statistics kernel, no exchange kernel, no second pass over the activation.  Same
 *      exchange sequence rule as rslo_peer_allreduce_f64 (one sequence number per call, all on one stream); a missing peer
 *      poisons the outputs with NaN and is reported by rslo_peer_status.  count_out / count_all: the element count over
 *      all ranks (device double), produced by the forward pass and read by the backward pass.  Replaces apex
 *      SyncBatchNorm forward / backward (rslo/layers/SparseConv.py:96-132) as rslo_bn2d_fwd_local / _bwd_local do on one rank. */
RSLO_API int rslo_bn2d_peer_supported(int N, int C, int HW);
RSLO_API int rslo_bn2d_fwd_peer(void *comm, const float *x, const float *res, const float *gamma, const float *beta, int N,
                                int C, int HW, float eps, float momentum, float act_slope, float *running_mean,
                                float *running_var, float *save_mean, float *save_invstd, double *count_out, float *y,
                                void *stream);
RSLO_API int rslo_bn2d_bwd_peer(void *comm, const float *dy, const float *y, const float *x, const float *gamma,
                                const float *save_mean, const float *save_invstd, const double *count_all, int N, int C,
                                int HW, float act_slope, int has_act, float *dx, float *dres, float *dgamma, float *dbeta,
                                void *stream);
/*      Round 6: exchanges inside a REPLAYED stream capture (the BEV head's forward as one hipGraph, rslo_amd/headgraph.py --
 *      the reference runs the same 45 SyncBatchNorm layers per step under apex DDP, train_hdf5.py:463).  Between
 *      rslo_peer_capture_begin and _end every exchange launched on the comm (rslo_peer_allreduce_f64, rslo_bn2d_fwd_peer /
 *      _bwd_peer) carries its number RELATIVE to a device word: its launch arguments are replay-invariant.  _end returns how
 *      many exchanges the capture holds.  In front of EVERY replay the caller issues rslo_peer_replay_prepare(comm, n, stream)
 *      on the replaying stream: it writes the word (= exchanges issued so far) in stream order and advances the comm's counter
 *      past the replay's n exchanges.  Every rank must capture the same exchanges in the same order (they run the same model). */
RSLO_API int rslo_peer_capture_begin(void *comm);
RSLO_API int rslo_peer_capture_end(void *comm, int *n_exchanges);
RSLO_API int rslo_peer_replay_prepare(void *comm, int n_exchanges, void *stream);
RSLO_API int rslo_peer_destroy(void *comm);

#ifdef __cplusplus
}
#endif
#endif /* RSLO_HIP_H_ */
