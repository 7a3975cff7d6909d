/* occformer_hip.h -- C ABI of liboccformer_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the OccFormer forward hot path.  Every entry point takes plain
 * device pointers + sizes + a hipStream_t (as void*), launches asynchronously on that
 * stream and returns 0 on success, a positive hipError_t, or a negative OCCF_E* code for
 * bad arguments.  No torch types cross this boundary.
 *
 * Citations are into the reference tree (zhangyp15/OccFormer):
 *   M/ = mmdetection3d/mmdet3d/      P/ = projects/mmdet3d_plugin/
 */
#ifndef OCCFORMER_HIP_H
#define OCCFORMER_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ view transform ---- */

/* crc32 of this header as the library was compiled against it; the loader refuses a mismatch. */
int occf_abi_hash(void);

/* Replaces bev_pool_ext.bev_pool_forward  (M/ops/bev_pool/src/bev_pool.cpp:22-57,
 * kernel bev_pool_cuda.cu:20-42).  x[n,c] f32 sorted by voxel rank, geom[n,4] i32 =
 * (x,y,z,b), interval starts/lengths [m] i32.  out[b,d,h,w,c] is zero-filled by the
 * callee (as the reference does) and out[b, z, x, y, :] = in-order fp32 sum of the
 * interval's rows -- bit-identical to the reference kernel.  Unlike the reference it
 * honours `stream` (the reference launches on the default stream, bev_pool_cuda.cu:88). */
int occf_bev_pool_fwd(const float* x, const int32_t* geom, const int32_t* interval_starts,
                      const int32_t* interval_lengths, float* out, int b, int d, int h, int w,
                      int n, int c, int n_intervals, void* stream);

/* Replaces bev_pool_ext.bev_pool_backward (bev_pool.cpp:60-87, bev_pool_cuda.cu:61-84):
 * x_grad[row,:] = out_grad[voxel(row),:]. */
int occf_bev_pool_bwd(const float* out_grad, const int32_t* geom, const int32_t* interval_starts,
                      const int32_t* interval_lengths, float* x_grad, int b, int d, int h, int w,
                      int n, int c, int n_intervals, void* stream);

/* Replaces get_geometry + the quantise/range-mask half of voxel_pooling
 * (P/occformer/image2bev/ViewTransformerLSSBEVDepth.py:117-150,
 *  P/occformer/image2bev/ViewTransformerLSSVoxel.py:83-94).
 * frustum[DHW,3]; cam[B*N,27] = inv(post_rots)[9] post_trans[3] (rots@inv(K))[9] trans[3]
 * K[:3,3][3]; bda[B,12] (3x4, row-major); grid[9] = lo[3] dx[3] nx[3];
 * vox[B*N*DHW] <- channels-last voxel row ((b*X+x)*Y+y)*Z+z, or -1 when outside. */
int occf_lss_voxel_index(const float* frustum, const float* cam, const float* bda,
                         const float* grid, int32_t* vox, int B, int N, int DHW, int X, int Y,
                         int Z, int bda4, void* stream);

/* Fused lift + splat: subsumes the volume materialisation, the two boolean-mask gathers,
 * bev_pool and the permutes of ViewTransformerLSSVoxel.py:112-118,77-100.
 * depth[BN,D,HW] (softmaxed), feat[BN,HW,C] channels-last, CSR over all n_vox voxels
 * (offsets[n_vox+1], sorted_pts = original point indices, ascending inside a voxel).
 * out[n_vox, C] = channels-last [B,X,Y,Z,C]; every row written exactly once. */
int occf_lift_splat_fwd(const float* depth, const float* feat, const int32_t* offsets,
                        const int32_t* sorted_pts, float* out, long n_vox, int BN, int D, int HW,
                        int C, void* stream);

/* Backward of the fused op (QuickCumsumCuda.backward + the lift's autograd,
 * M/ops/bev_pool/bev_pool.py:63-80).  vox[n_pts] as produced by occf_lss_voxel_index. */
int occf_lift_splat_bwd(const float* out_grad, const float* depth, const float* feat,
                        const int32_t* vox, float* d_depth, float* d_feat, long n_pts, int BN,
                        int D, int HW, int C, void* stream);

/* ------------------------------------------------------------------ dual-path encoder -- */

/* Windowed MSA core of the shared SwinBlock: replaces F.pad + torch.roll + window_partition +
 * WindowMSA's (q@k^T + rel-pos bias + shift mask -> softmax -> @v) + window_reverse + un-roll +
 * crop  (P/occformer/backbones/modules/window_attention.py:69-107,168-242).
 * qkv[n_tok, 3C] = fused qkv projection of the layer-normed tokens, token order
 * ((b*X + x)*Y + y)*S + s (S slices per batch: Z height slices + the BEV mean slice);
 * qkv_bias[3C] (value of padded tokens); bias_table[(2*7-1)^2, heads];
 * out[n_tok, C] = attention output before `proj`.  head_dim is 32 (dualpath_block.py:31-32),
 * window 7, shift in {0, 3}. */
int occf_window_attn_fwd(const float* qkv, const float* qkv_bias, const float* bias_table,
                         float* out, int B, int X, int Y, int S, int C, int heads, int shift,
                         void* stream);

/* Attention half of the shared SwinBlock fused (embed_dims 128, 4 heads, 7x7 windows):
 * out = x + proj(WindowMSA(LayerNorm(x)))  -- P/occformer/backbones/modules/window_attention.py:346-372 (first
 * residual), :69-107, :168-242.  x/out[B*X*Y*S, 128] token rows as in occf_window_attn_fwd; wqkv[384, 128] and
 * wproj[128, 128] pre-split bf16 (hi, lo); bias_table[(2*7-1)^2, 4].  x must not alias out.  Returns OCCF_ESHAPE
 * (-2) for other widths (the caller composes layernorm / linear / window_attn / linear). */
int occf_swin_attn_fused_fwd(const float* x, const float* ln_gamma, const float* ln_beta, float eps,
                             const uint16_t* wqkv_hi, const uint16_t* wqkv_lo, const float* bqkv,
                             const float* bias_table, const uint16_t* wproj_hi, const uint16_t* wproj_lo,
                             const float* bproj, float* out, int B, int X, int Y, int S, int C, int heads,
                             int shift, int weights_packed, void* stream);
/* weights_packed != 0: wqkv / wproj (hi and lo) are given in MFMA-fragment order, as written by occf_swin_attn_pack
 * (w[rows, 128] -> [rows / 32][8 k-steps][64 lanes][8]): a wave's weight load then reads 1 KB contiguous. */
int occf_swin_attn_pack(const uint16_t* w_hi, const uint16_t* w_lo, uint16_t* f_hi, uint16_t* f_lo, int rows, int C,
                        void* stream);

/* ------------------------------------------------------------------ pixel decoder ------ */

/* Sampling core of MultiScaleDeformableAttention3D: replaces the location arithmetic, the
 * softmax over levels*points and multi_scale_deformable_attn_pytorch
 * (P/occformer/necks/multi_scale_deform_attn_3d.py:246-273, 17-80).
 * value[B, Nq, heads*head_dim] (projected), sampling_offsets[B, Nq, heads, L, P, 3] raw linear
 * output (last dim ordered z, y, x), attn_logits[B, Nq, heads, L*P] raw, out[B, Nq, heads*head_dim]
 * (before output_proj).  level_shapes is a HOST array [L][3] = (X, Y, Z) per level, coarse->fine
 * as the decoder concatenates them; the queries are the level cells themselves (Nq = sum XYZ).
 * value_head_major = 1: value is [B, heads, Nq, head_dim] and lanes walk the queries of one head
 * (adjacent queries sample adjacent cells -> shared cache lines); out stays [B, Nq, heads*head_dim].
 * offsets_ld / logits_ld: floats between consecutive query rows of the two inputs (0 = dense), so both
 * may be column blocks of ONE fused projection output. */
int occf_msda3d_fwd(const float* value, const float* sampling_offsets, const float* attn_logits,
                    float* out, const int32_t* level_shapes, int num_levels, int B, int Nq,
                    int heads, int head_dim, int num_points, int value_head_major, long offsets_ld,
                    long logits_ld, void* stream);

/* ------------------------------------------------------------------ occupancy decoder -- */

/* Preserve-pooling: F.adaptive_max_pool3d(mask_pred, (ox,oy,oz)) -> sigmoid < 0.5
 * (P/occformer/mask2former/mask2former_nusc_occ.py:457-466).  mask_pred[BQ, X, Y, Z];
 * pooled[BQ, L] logits, blocked[BQ, L] bytes (1 = masked out), row_open[BQ] (1 when any key of
 * the row is open; zeroed by the callee). */
int occf_mask_pool_fwd(const float* mask_pred, float* pooled, uint8_t* blocked, int32_t* row_open,
                       long BQ, int X, int Y, int Z, int ox, int oy, int oz, void* stream);

/* Fused mask_embed x mask_feature contraction + preserve-pooling for decoder layers whose full
 * mask logits are not consumed (every layer but the last in simple_test,
 * mask2former_nusc_occ.py:448-466, 713-731): the [B, Q, X, Y, Z] logits are never written.
 * mask_embed[B, Q, E] fp32, feat_hi/lo[B, V, E] = bf16 split of the channels-last mask features,
 * outputs as occf_mask_pool_fwd (pooled[B*Q, L], blocked, row_open).  Uniform pooling windows only
 * (ox | X, oy | Y, oz | Z, 128 % Z == 0, window rows dividing the 128/Z rows of a tile), Q <= 128,
 * E % 32 == 0; otherwise OCCF_ESHAPE (-2) / workspace 0 and the caller uses GEMM + occf_mask_pool_fwd.
 * workspace: occf_mask_gemm_pool_workspace(...) floats.
 * reverse != 0 walks the volume from the far end (results are identical): the ten prediction sets of a
 * forward read the same features, which exceed the last-level cache -- alternating the direction between
 * consecutive calls lets each pass start on the tail the previous one left there. */
long occf_mask_gemm_pool_workspace(int B, int Q, int E, int X, int Y, int Z, int ox, int oy, int oz);
int occf_mask_gemm_pool_fwd(const float* mask_embed, const uint16_t* feat_hi, const uint16_t* feat_lo,
                            float* pooled, uint8_t* blocked, int32_t* row_open, float* workspace, int B, int Q,
                            int E, int X, int Y, int Z, int ox, int oy, int oz, int terms, int reverse,
                            void* stream);

/* Masked multi-head cross-attention core (scaled dot product + boolean mask + softmax + @V) of
 * the decoder layers, incl. the all-masked-row fix (mask2former_nusc_occ.py:652-667; mmcv
 * MultiheadAttention -> torch.nn.MultiheadAttention).  q[B, Q, E], k/v[B, L, E] already
 * projected; blocked may be NULL (plain attention); out[B, Q, E] before out_proj; head_dim 32.
 * workspace: occf_masked_xattn_workspace(...) floats of scratch. */
int occf_masked_xattn_fwd(const float* q, const float* k, const float* v, const uint8_t* blocked,
                          const int32_t* row_open, float* out, float* workspace,
                          long workspace_floats, int B, int Q, int L, int E, int heads,
                          void* stream);
long occf_masked_xattn_workspace(int B, int Q, int L, int heads);

/* simple_test's tail fused: F.interpolate(mask_pred, occ_size, trilinear, align_corners=True) ->
 * sigmoid -> einsum('bqc,bqxyz->bcxyz') with softmax(cls)[..., :-1]
 * (mask2former_nusc_occ.py:725-733, 691-696).  cls[B, Q, K+1]; out[B, K, X2, Y2, Z2]; K <= 24;
 * workspace: B*Q*24 floats (the class probabilities). */
int occf_upsample_classify_fwd(const float* mask_pred, const float* cls, float* out, float* workspace, int B,
                               int Q, int K, int X, int Y, int Z, int X2, int Y2, int Z2, void* stream);

/* forward_lidarseg (eval branch, mask2former_nusc_occ.py:505-542): class volume at mask
 * resolution sampled at points (bilinear=trilinear, align_corners=True, border padding) and
 * soft-maxed.  pts[P, 4] = (batch, gx, gy, gz) in [-1, 1] along (X, Y, Z); out[P, K]. */
int occf_lidarseg_sample_fwd(const float* mask_pred, const float* cls, const float* pts, float* out,
                             int P, int B, int Q, int K, int X, int Y, int Z, void* stream);

/* ------------------------------------------------------------------ dense contractions - */

/* out[M, N] = act(x[M, K] @ weight[N, K]^T + bias[N]) + residual[M, N]  on the fp32 matrix
 * cores (v_mfma_f32_32x32x2_f32).  Replaces nn.Linear / mmcv FFN / the 1x1x1 convolutions on
 * channels-last tokens (e.g. P/occformer/backbones/modules/window_attention.py:63-66,336-344;
 * P/occformer/mask2former/mask2former_nusc_occ.py:448-455).  bias/residual may be NULL;
 * act: 0 none, 1 ReLU, 2 exact GELU; ldx/ldo/ldr = row strides in floats; K % 4 == 0. */
int occf_linear_fwd(const float* x, const float* weight, const float* bias, const float* residual,
                    float* out, long M, int N, int K, long ldx, long ldo, long ldr, int act,
                    void* stream);

/* occf_linear_fwd for tiny problems (the 100-query decoder): one thread per output, exact fp32,
 * no tile padding.  Same arguments; intended for M <= 128 and M*N <= 256k. */
int occf_linear_small_fwd(const float* x, const float* weight, const float* bias, const float* residual,
                          float* out, int M, int N, int K, long ldx, long ldo, long ldr, int act,
                          void* stream);

/* Implicit-GEMM convolution over a channels-last volume (nn.Conv3d / nn.Conv2d as Zi = kZ = 1):
 * x[B, Xi, Yi, Zi, Cin] addressed by element strides (in_sb, in_sx, in_sy, in_sz; channel stride
 * 1), weight_tapmajor[Cout, kX*kY*kZ*Cin] (k = ((dx*kY + dy)*kZ + dz)*Cin + cin), zero padding,
 * out[B*Xo*Yo*Zo, Cout] channels-last; same epilogue as occf_linear_fwd.  Replaces the 3^3 / 1^3
 * convolutions of P/occformer/backbones/dualpath_block.py:36-48 and
 * P/occformer/necks/multiscale_deformattn_3d.py:70-116.  Cin % 4 == 0 (fast path Cin % 16 == 0). */
int occf_conv3d_fwd(const float* x, const float* weight_tapmajor, const float* bias,
                    const float* residual, float* out, int B, int Xi, int Yi, int Zi, int Cin, int Cout,
                    int kX, int kY, int kZ, int stride, int dil, int pad_x, int pad_y, int pad_z,
                    long in_sb, long in_sx, long in_sy, long in_sz, int act, void* stream);

/* Fused token MLP: out = LNpost?( x + W2 . act(W1 . LNpre?(x) + b1) + b2 ), x/out[M, C] fp32,
 * W1[H, C] / W2[C, H] pre-split bf16 (hi, lo), act 1 = ReLU, 2 = exact GELU, ln_mode 0 none /
 * 1 pre-LN (MLP input only; SwinBlock norm2 + FFN, window_attention.py:356-361) / 2 post-LN
 * (pixel-decoder layer 'ffn','norm').  C in {128, 192, 256}, H % 128 == 0; the [M, H] hidden
 * activation never reaches HBM. */
int occf_mlp_fused_fwd(const float* x, const float* ln_gamma, const float* ln_beta, const uint16_t* w1_hi,
                       const uint16_t* w1_lo, const float* b1, const uint16_t* w2_hi, const uint16_t* w2_lo,
                       const float* b2, float* out, long M, int C, int H, int act, int ln_mode, float eps,
                       int terms, void* stream);

/* ------------------------------------------------------------------ norms / fusion ----- */

/* GroupNorm over channels-last x[B, V, C] (nn.GroupNorm, eps inside the sqrt): stats[B, G, 2] =
 * (mean, rstd); deterministic two-stage reduction.  workspace: occf_groupnorm_workspace floats. */
long occf_groupnorm_workspace(int B, long V, int C, int G);
int occf_groupnorm_stats(const float* x, float* stats, float* workspace, int B, long V, int C, int G,
                         float eps, void* stream);
/* second stage alone, for the per-channel partial sums written by a convolution / GEMM epilogue:
 * partial[B][nblk][C][2] -> stats[B, G, 2]; count = elements per (batch, group) = V * C / G. */
int occf_groupnorm_finalize(const float* partial, float* stats, int B, long nblk, int C, int G, double count,
                            float eps, void* stream);
/* y = (x - mean) * rstd * gamma + beta [ReLU] [+ residual];  x[B, P, Z, C] -> out[B, P, Zs, C].
 * tokens = 1: Zs = Z + 1 and slot Z = mean_z(y): builds the dual-path token buffer (the Z height
 * slices plus the BEV slice, dualpath_block.py:70-76) in the same pass. */
int occf_groupnorm_apply(const float* x, const float* stats, const float* gamma, const float* beta,
                         const float* residual, float* out, int B, long P, int Z, int C, int G, int relu,
                         int tokens, void* stream);

/* Row LayerNorm (nn.LayerNorm over the last dim), x[M, C] -> out[M, C], C <= 1024. */
int occf_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* out, long M, int C,
                       float eps, void* stream);

/* Dual-path fusion (dualpath_block.py:79-82): out = tok + sigmoid(<tok, w> + b) * bev + identity.
 * tokens[BP, Z+1, C], bev[BP, C], identity/out[BP, Z, C]; coeff_bias: device scalar or NULL. */
int occf_dualpath_combine(const float* tokens, const float* bev, const float* coeff_weight,
                          const float* coeff_bias, const float* identity, float* out, long BP, int Z, int C,
                          void* stream);

/* FPN top-down step (P/occformer/necks/multiscale_deformattn_3d.py:233-243):
 * out = lateral + trilinear_upsample(coarse -> lateral's size, align_corners=False); channels-last
 * coarse[B, X, Y, Z, C], lateral/out[B, X2, Y2, Z2, C]. */
int occf_upsample_add(const float* coarse, const float* lateral, float* out, int B, int X, int Y, int Z,
                      int X2, int Y2, int Z2, int C, void* stream);

/* ------------------------------------------------------------------ split-bf16 contractions */

/* Same contracts as occf_linear_fwd / occf_conv3d_fwd, products on the bf16 matrix cores
 * (v_mfma_f32_32x32x16_bf16) with fp32 accumulation.  terms = 3: every operand is split
 * a = a_hi + a_lo (bf16) and a_hi*b_hi + a_hi*b_lo + a_lo*b_hi is accumulated -- fp32-class
 * accuracy (~2^-17 rel. per product) at 16/3 of the fp32-MFMA rate; terms = 1: plain bf16 (w_lo may
 * be NULL).  The weight is given pre-split (occf_split_bf16) as two bf16 arrays [N, K]; the
 * activation is split while it is staged.  K % 32 == 0 (conv: Cin % 32 == 0).
 * out_head_dim > 0 (linear only): the output is written head-major, out[b, n / out_head_dim, q,
 * n % out_head_dim] with row m = b * out_head_rows + q -- the layout occf_msda3d_fwd gathers from
 * (value_head_major), produced by the projection itself instead of a transposing copy; ldo is ignored,
 * N % out_head_dim == 0, out_head_dim % 4 == 0, no residual.
 * gn_partial != NULL: the epilogue also emits per-column partial sums of the stored outputs,
 * gn_partial[M-tile][N][2] (sum, sum of squares; M-tile = 128 rows; a batch element must consist of whole
 * tiles, or B = 1), to be reduced by occf_groupnorm_finalize -- the GroupNorm statistics pass over the
 * convolution output (one full read of the tensor) disappears.  No split-K in that case. */
int occf_linear_bf16_fwd(const float* x, const uint16_t* w_hi, const uint16_t* w_lo, const float* bias,
                         const float* residual, float* out, long M, int N, int K, long ldx, long ldo,
                         long ldr, int act, int terms, float* workspace, long workspace_floats,
                         int out_head_dim, long out_head_rows, float* gn_partial, void* stream);
/* Linear only: the streaming shapes -- M >= 16 384 rows (OCCF_GEMM_STREAM=n: from n rows; 0: never), K in
 * {64, 96, ..., 256}, N % 32 == 0, no head-major output / GroupNorm partials -- run on a weight-resident persistent
 * kernel (csrc/gemm_stream.h: the weight block stays in LDS, rows stream global -> registers -> MFMA -> global).
 * occf_linear_stream_launches: how many calls took it since the library was loaded (diagnostics / tests). */
long occf_linear_stream_launches(void);
/* The same kernel with the epilogues of the TRAINING graph's Swin block (window_attention.py:300-344: DropPath around
 * the attention projection and around the FFN; mmcv FFN = Linear, GELU, Linear):
 *   act = 2 with pre_out != NULL: out = GELU(z), pre_out = z = x W^T + b  (the backward needs z; no separate GELU pass)
 *   row_scale != NULL: out = residual + row_scale[sample(row)] * (x W^T + b), sample(row) = (row / xy_s) * s_slices +
 *     row % s_slices  (token rows ((b XY + xy) S + s), xy_s = XY * S: mmcv DropPath, one draw per slice) -- no
 *     separate DropPath pass
 *   act = 3: out = (x W^T + b) * GELU'(aux[row, col]) with aux = residual_or_aux (row stride ldr): the data gradient
 *     through the second FFN linear AND the GELU in one pass.
 * Returns OCCF_ESHAPE when the shape is outside the kernel's envelope (the caller then runs the unfused sequence). */
/* 1 when a [M, K] x [N, K]^T linear (contiguous rows, ld % 4 == 0) is inside that kernel's envelope right now */
int occf_linear_stream_takes(long M, int N, int K);
int occf_linear_stream_fwd(const float* x, const uint16_t* w_hi, const uint16_t* w_lo, const float* bias,
                           const float* residual_or_aux, float* out, float* pre_out, const float* row_scale, long M,
                           int N, int K, long ldx, long ldo, long ldr, int act, int terms, long xy_s, int s_slices,
                           void* stream);
int occf_conv3d_bf16_fwd(const float* x, const uint16_t* w_hi, const uint16_t* w_lo, const float* bias,
                         const float* residual, float* out, int B, int Xi, int Yi, int Zi, int Cin, int Cout,
                         int kX, int kY, int kZ, int stride, int dil, int pad_x, int pad_y, int pad_z,
                         long in_sb, long in_sx, long in_sy, long in_sz, int act, int terms, float* workspace,
                         long workspace_floats, float* gn_partial, void* stream);
/* Small-M problems (few output tiles, long K: the coarse encoder stages) are split along K over
 * blockIdx.y into partial slabs and reduced in fixed order; workspace (may be NULL = never split)
 * needs occf_gemm_bf16_workspace(M, N, K) floats (0 = this shape is not split). */
long occf_gemm_bf16_workspace(long M, int N, int K);
/* hi = bf16_rne(x), lo = bf16_rne(x - hi), n elements. */
int occf_split_bf16(const float* x, uint16_t* hi, uint16_t* lo, long n, void* stream);

/* 3x3x3 / stride 1 / pad 1 convolution with an LDS-resident halo tile (same contract and weight
 * layout as occf_conv3d_bf16_fwd restricted to that geometry): the input is staged and split to
 * bf16 once per 32-channel chunk instead of once per tap.  Returns OCCF_ESHAPE (-2) for shapes
 * outside its envelope (Cin % 32, Cout % 64, Z in {4, 8, 16k}); the caller then uses
 * occf_conv3d_bf16_fwd.  x strides as in occf_conv3d_fwd; out[B*X*Y*Z, Cout] channels-last. */
int occf_conv3x3x3_halo_fwd(const float* x, const uint16_t* w_hi, const uint16_t* w_lo, const float* bias,
                            const float* residual, float* out, int B, int X, int Y, int Z, int Cin, int Cout,
                            long in_sb, long in_sx, long in_sy, long in_sz, int act, int terms, float* gn_partial,
                            const uint16_t* wfrag_hi, const uint16_t* wfrag_lo, void* stream);
/* Optional: the same weights in MFMA-fragment order (wfrag_hi / wfrag_lo above; NULL = weight slabs staged through
 * LDS with a barrier per tap).  occf_conv3x3x3_halo_pack_elems = uint16 elements per array (0: shape not packed);
 * occf_conv3x3x3_halo_pack permutes w_hi / w_lo [Cout, 27*Cin] into [Cin/32][27][2][Cout/32][64 lanes][8]. */
long occf_conv3x3x3_halo_pack_elems(int Cin, int Cout);
int occf_conv3x3x3_halo_pack(const uint16_t* w_hi, const uint16_t* w_lo, uint16_t* f_hi, uint16_t* f_lo, int Cin,
                             int Cout, void* stream);
/* gn_partial (both convolutions): like occf_linear_bf16_fwd -- [B][tiles per batch element][Cout][2];
 * tiles per batch element = ceil(Xo*Yo*Zo / 128) for occf_conv3d_bf16_fwd (must divide unless B = 1) and
 * occf_conv3x3x3_halo_gn_blocks(X, Y, Z) for the halo kernel (-1: shape not taken). */
long occf_conv3x3x3_halo_gn_blocks(int X, int Y, int Z);

/* The same convolution as Winograd F(2, 3) along x over the same LDS halo tile (csrc/conv_wino.hip): per output pair
 * (x0, x0 + 1) four 9-tap contractions of x-transformed planes against x-transformed filters instead of two 27-tap ones
 * -- 2/3 of the matrix-core products, 3-term bf16 split of the transformed values.  The filter transform happens once
 * per weight version: occf_conv3x3x3_wino_pack reads the fp32 tap-major weight [Cout, 27*Cin] (tap = (dx*3+dy)*3+dz) and
 * writes U (hi, lo) in fragment order [Cin/32][4][9][2][Cout/32][64 lanes][8] (occf_conv3x3x3_wino_pack_elems uint16
 * elements per array; 0 = shape outside the envelope: Cin % 32, Cout % 64, or OCCF_WINO=0).  occf_conv3x3x3_wino_fwd
 * returns OCCF_ESHAPE (-2) outside its envelope (Z in {4, 8, 16k}); the caller then uses occf_conv3x3x3_halo_fwd.
 * gn_partial rows per batch element: occf_conv3x3x3_wino_gn_blocks (tiles of one x-pair x 64 (y, z) positions). */
long occf_conv3x3x3_wino_pack_elems(int Cin, int Cout);
int occf_conv3x3x3_wino_pack(const float* w_tapmajor, uint16_t* f_hi, uint16_t* f_lo, int Cin, int Cout, int f16,
                             void* stream);
long occf_conv3x3x3_wino_gn_blocks(int X, int Y, int Z);
int occf_conv3x3x3_wino_fwd(const float* x, const uint16_t* wfrag_hi, const uint16_t* wfrag_lo, const float* bias,
                            const float* residual, float* out, int B, int X, int Y, int Z, int Cin, int Cout,
                            long in_sb, long in_sx, long in_sy, long in_sz, int act, float* gn_partial,
                            const uint32_t* f16_scale, void* stream);
/* (wfrag_lo == NULL together with f16_scale != NULL selects the ONE-product form of the two-product kernel below: the
 * transformed filters too as one fp16 piece -- the hi fragments alone.) */
/* Two-product form of the same kernel (f16_scale != NULL, fragments packed with f16 = 1): x enters as ONE fp16 piece of
 * (transformed x) * 2^k, k from max |x| (f16_scale = a scale slot filled by occf_absmax_f32 on the same stream), the
 * filters as fp16 (hi, lo); the output leaves multiplied by 2^-k.  The op layer uses it for the data gradients.
 * occf_absmax_f32: max |x| of x[rows][cols] (row stride ld, cols % 4 == 0) into slot[0] as an fp32 bit pattern;
 * slot = occf_absmax_slot_words() uint32 words of device memory; two launches, no atomics. */
long occf_absmax_slot_words(void);
int occf_absmax_f32(const float* x, long rows, int cols, long ld, uint32_t* slot, void* stream);
int occf_absmax_flat(const float* x, long n, uint32_t* slot, void* stream);   /* any length / alignment */

/* Reproducible scatter sums (the op layer's ``deterministic`` mode, OCCF_DETERMINISTIC=1): the scatter kernels of the
 * backward with their float atomics replaced by 64-bit FIXED-POINT integer atomics -- the same contributions add up to the
 * same bits in any arrival order.  acc = a ZERO-FILLED int64 buffer shaped like the float result, slot = a scale slot
 * holding the maximum of the scattered tensor (occf_absmax_*: contributions are that tensor times weights <= 1, scaled
 * to < 2^30); occf_fx_to_f32 turns the sums back into floats (precision 2^-30 of that maximum). */
int occf_point_sample_3d_bwd_fx(const float* dout, const float* pts, long long* acc, const uint32_t* slot, int N, int C,
                                int X, int Y, int Z, long P, int shared_pts, int align_corners, int border_padding,
                                long voxel_major_ld, void* stream);
int occf_deform_col2im_fx(const float* x, const float* offset, const float* mask, const float* dcol, long long* acc,
                          float* doffset, float* dmask, const uint32_t* slot, int BN, int H, int W, int C, int K,
                          int stride, int pad, int dil, int groups, int deform_groups, void* stream);
int occf_fx_to_f32(const long long* acc, float* out, long n, const uint32_t* slot, void* stream);

/* ------------------------------------------------------------------ decoder rows (inference) ------ */

/* The occupancy decoder's per-query work between its attention kernels as two kernels per layer
 * (DetrTransformerDecoderLayer ('cross_attn','norm','self_attn','norm','ffn','norm') + forward_head's per-query half,
 * P/occformer/mask2former/mask2former_nusc_occ.py:426-456, 640-667; csrc/decoder_rows.hip).  rows = B * Q query rows of
 * E channels; qpos [Q, E] is added per sample (row % Q).  Every weight matrix W [N, K] enters as a PAIR of fragment
 * arrays {hi, lo} (a host array of two device pointers) made by occf_decoder_rows_pack (bf16 split, order
 * [ceil(N/16)][K/32][64 lanes][8]; occf_decoder_rows_pack_elems uint16 elements per array, 0: K % 32 != 0).
 *   K1: q1 = LN(attn_out Wo^T + bo + q_in); [qs | ks] = (q1 + qpos) [Wq; Wk]^T + b_qk; vs = q1 Wv^T + b_v
 *   K2 (mode 1): q2 = LN1(attn_out Wo^T + bo + q_in); q3 = LN2(relu(q2 W1^T + b1) W2^T + b2 + q2);
 *       d = LNpost(q3); cls = d Wc^T + bc [rows, n_cls]; mask_embed = W_me2 relu(W_me1 relu(W_me0 d)) (+ biases);
 *       qx = (q3 + qpos) Wqnext^T + b_qnext (f_qnext NULL: skipped).  mode 0: q_in IS q3 (the set before layer 0).
 * E % 32 == 0, H % 32 == 0, LDS: 16 rows of all activations (160 KB at E = 192, H = 1536); else OCCF_ESHAPE. */
long occf_decoder_rows_pack_elems(int N, int K);
int occf_decoder_rows_pack(const float* w, long ld, int N, int K, uint16_t* f_hi, uint16_t* f_lo, void* stream);
int occf_decoder_rows_k1(const float* attn_out, const float* q_in, const float* qpos, int rows, int E, int Q,
                         const uint16_t* const* f_out, const float* b_out, const float* ln_gamma,
                         const float* ln_beta, float ln_eps, const uint16_t* const* f_qk, const float* b_qk,
                         const uint16_t* const* f_v, const float* b_v, float* q1, float* qs, float* ks,
                         float* vs, void* stream);
int occf_decoder_rows_k2(int mode, const float* attn_out, const float* q_in, const float* qpos, int rows,
                         int E, int H, int Q, const uint16_t* const* f_out, const float* b_out,
                         const float* ln1_gamma, const float* ln1_beta, float ln1_eps,
                         const uint16_t* const* f_ffn1, const float* b_ffn1, const uint16_t* const* f_ffn2,
                         const float* b_ffn2, const float* ln2_gamma, const float* ln2_beta, float ln2_eps,
                         const float* post_gamma, const float* post_beta, float post_eps,
                         const uint16_t* const* f_cls, const float* b_cls, int n_cls,
                         const uint16_t* const* f_me0, const float* b_me0, const uint16_t* const* f_me1,
                         const float* b_me1, const uint16_t* const* f_me2, const float* b_me2,
                         const uint16_t* const* f_qnext, const float* b_qnext, float* q3, float* cls,
                         float* mask_embed, float* qx, void* stream);

/* ------------------------------------------------------------------ DepthNet's DCN ------ */

/* Deformable im2col of mmcv-full 1.4.0 `deform_conv2d` (DCNv1; third-party op behind
 * `build_conv_layer(dict(type='DCN', ...))`, P/occformer/image2bev/ViewTransformerLSSBEVDepth.py:479-487).
 * x[BN, H, W, C] channels-last, offset[BN, dg*2*K*K, Ho, Wo] (conv_offset's NCHW output, (dy, dx)
 * pairs per tap), col[BN*Ho*Wo, groups, K*K, C/groups]; the grouped contraction is occf_linear_*
 * per group on the contiguous K-slabs. */
int occf_deform_im2col(const float* x, const float* offset, float* col, int BN, int H, int W, int C, int K,
                       int stride, int pad, int dil, int groups, int deform_groups, void* stream);

/* DCNv2 (mmcv-full 1.4.0 `modulated_deform_conv2d`, third-party op behind `dcn=dict(type='DCNv2', ...)` of the
 * R101-DCN image backbone, projects/configs/occformer_nusc/occformer_nusc_r101_896x1600.py:78-79): the same
 * columns with every tap's sample scaled by mask[BN, dg*K*K, Ho, Wo] (sigmoid already applied). */
int occf_modulated_deform_im2col(const float* x, const float* offset, const float* mask, float* col, int BN, int H,
                                 int W, int C, int K, int stride, int pad, int dil, int groups, int deform_groups,
                                 void* stream);

/* Inference epilogue of the 2-D image branch (ResNet Bottleneck / SECONDFPN, outside the north-star path but inside
 * the reference's from-images forward, occupancyformer.py:59-67): y[p, c] = act(y[p, c] * scale[c] + shift[c]
 * (+ residual[p, c])) IN PLACE over rows x C channels-last elements -- eval-mode BatchNorm + identity add + ReLU as one
 * pass over the MIOpen convolution's output.  bf16 != 0: y / residual are bf16 (C % 8 == 0), else fp32 (C % 4 == 0);
 * scale / shift are fp32 [C]. */
int occf_scale_shift_act(void* y, const float* scale, const float* shift, const void* residual, long rows, int C,
                         int relu, int bf16, void* stream);

/* ------------------------------------------------------------------ training-time sampling */

/* point_sample_3d (P/occformer/mask2former/base/mmdet_utils.py:21-47 = F.grid_sample on
 * points in [0,1]): vol[N, C, X, Y, Z], pts[N (or 1 if shared_pts), P, 3] with the last dim in
 * grid_sample order (pts[..,0] -> Z, [..,1] -> Y, [..,2] -> X), out[N, C, P]; trilinear,
 * align_corners flag, zeros (0) or border (1) padding. */
int occf_point_sample_3d_fwd(const float* vol, const float* pts, float* out, int N, int C, int X, int Y, int Z,
                             long P, int shared_pts, int align_corners, int border_padding, void* stream);

/* The same with a ROW INDIRECTION and one channel: sample n reads the volume vol[rows[n]] of a stack vol[R, X, Y, Z]
 * (rows int64 [N], values in [0, R)), out[N, P].  Replaces gt_masks[pos_assigned_gt_inds] followed by point_sample_3d
 * (mask2former_nusc_occ.py:253-262, 383-390): the matched ground-truth rows are never copied. */
int occf_point_sample_3d_rows_fwd(const float* vol, const int64_t* rows, const float* pts, float* out, int N, int X,
                                  int Y, int Z, long P, int shared_pts, int align_corners, int border_padding,
                                  void* stream);

/* The same sampling on a CHANNELS-LAST volume: tok[X*Y*Z, C] (row stride ld >= C, C % 4 == 0), pts[P, 3] ->
 * out[P, C].  Used for the matching cost of the training step (mask2former_nusc_occ.py:232-238,
 * mask2former_occ.py:258-262 sample all Q query logits at the matching points): the query logits are a linear map of
 * the mask features, einsum('qc,cxyz->qxyz') (mask2former_nusc_occ.py:455), and trilinear sampling is linear, so
 * sampling the features (8 contiguous C-float rows per point) and contracting with mask_embed afterwards gives the
 * same [Q, P] matrix without 8 x Q scattered 4-byte gathers per point from the [Q, X, Y, Z] logits. */
int occf_point_sample_tokens_fwd(const float* tok, const float* pts, float* out, int X, int Y, int Z, int C, long ld,
                                 long P, int align_corners, int border_padding, void* stream);

/* Weighted sampling without replacement = torch.multinomial(weights, k, replacement=False) of the
 * class-guided sampler (mmdet_utils.py:91-136): the k largest exponential-race keys
 * weights[i] / -log(uniforms[r, i]) per row, found by radix select + wavefront compaction.
 * weights[R (or 1 if weights_shared), V], noise[R, V] = uniforms in (0, 1] or, with
 * noise_is_exponential, the Exp(1) draws q themselves (key = w / q), out_indices[R, k] int64 (an
 * unordered set), workspace: occf_sample_wor_workspace(R, V) floats. */
long occf_sample_wor_workspace(int R, long V);
int occf_sample_wor_fwd(const float* weights, const float* noise, int64_t* out_indices, float* workspace,
                        int R, long V, long k, int weights_shared, int noise_is_exponential, void* stream);

/* Importance sampling (mmdet_utils.py:138-177, 179-246: topk(-|logit|, k)): indices of the k values
 * with the smallest magnitude per row; values[R, V], out_indices[R, k] int64 (unordered set),
 * workspace as occf_sample_wor_workspace(R, V). */
int occf_topk_smallest_abs_fwd(const float* values, int64_t* out_indices, float* workspace, int R, long V, long k,
                               void* stream);

/* Row sums behind the point-sampled mask losses (mask2former_nusc_occ.py:396-417; losses/dice_loss.py:8-61):
 * out[R, 4] = { sum BCE-with-logits(x, t), sum sigmoid(x)*t, sum sigmoid(x), sum t } over logits/targets[R, P]. */
int occf_point_loss_rows_fwd(const float* logits, const float* targets, float* out, int R, long P, void* stream);

/* Hungarian matching (assigners/mask_hungarian_assigner.py:104-126: scipy.optimize.linear_sum_assignment on
 * cost.cpu()): P independent problems, cost[P, Q, G] (queries x ground-truth rows, finite fp32).
 * match_gt[P, G] int32 = query matched to GT g (-1 if G > Q left it unmatched);
 * assigned_gt[P, Q] int32 = the reference's assigned_gt_inds (0 = background, g + 1 = matched to GT g).
 * Minimises the same total cost as scipy (shortest augmenting paths in double); Q, G <= 1024. */
int occf_hungarian_fwd(const float* cost, int* match_gt, int* assigned_gt, int P, int Q, int G, void* stream);

/* ================================================================== training step: backward kernels
 * The reference's training step is ATen autograd through the modules above plus QuickCumsumCuda.backward
 * (M/ops/bev_pool/bev_pool.py:63-80; P/occformer/detectors/occupancyformer.py:132-199,
 * P/occformer/apis/mmdet_train.py:72-80).  Each forward entry point of the path has its backward here; the host
 * side wraps the pairs as torch.autograd.Function (occformer_amd/autograd.py). */

/* out[N] = sum over rows of x[M, N] (row stride ldx): bias gradients.  workspace: occf_colsum_workspace floats. */
long occf_colsum_workspace(long M, int N);
int occf_colsum(const float* x, float* out, float* workspace, long M, int N, long ldx, void* stream);

/* nn.LayerNorm backward: x/dy/dx[M, C]; dgamma/dbeta[C] (deterministic two-stage sums).
 * workspace: occf_layernorm_bwd_workspace floats.  addend (may be NULL) [M, C]: dx = addend + (the LayerNorm gradient) --
 * the gradient that reaches x through the residual connection around the normalised branch, added in the same pass. */
long occf_layernorm_bwd_workspace(long M, int C);
int occf_layernorm_bwd(const float* x, const float* gamma, const float* dy, const float* addend, float* dx,
                       float* dgamma, float* dbeta, float* workspace, long M, int C, float eps, void* stream);

/* Backward of occf_groupnorm_apply (same x / stats / flags): dy[B, P, Zs, C] (Zs = Z + 1 in token mode: the
 * gradient of the z-mean slot is spread over the Z slices), dx[B, P, Z, C], dgamma/dbeta[C]; dresidual (may be
 * NULL) receives the gradient of the forward's residual operand.  workspace: occf_groupnorm_bwd_workspace. */
long occf_groupnorm_bwd_workspace(int B, long V, int C, int G);
int occf_groupnorm_bwd(const float* x, const float* stats, const float* gamma, const float* beta, const float* dy,
                       float* dx, float* dgamma, float* dbeta, float* dresidual, float* workspace, int B, long P,
                       int Z, int C, int G, int relu, int tokens, void* stream);

/* y = act(x) / dx = dy * act'(x), act 1 = ReLU, 2 = exact GELU (mmcv FFN of the SwinBlock, window_attention.py
 * :336-344); n % 4 == 0. */
int occf_act_fwd(const float* x, float* y, long n, int act, void* stream);
int occf_act_bwd(const float* x, const float* dy, float* dx, long n, int act, void* stream);

/* DropPath of the shared SwinBlock in training mode (window_attention.py:311,332): out = identity + branch *
 * scale[sample]; token row ((b*XY + xy)*S + s) belongs to sample b*S + s (the block's batch are the slices of
 * the token buffer); scale = 0 or 1 / keep_prob.  identity == NULL: out = branch * scale (= the backward). */
int occf_droppath(const float* identity, const float* branch, const float* scale, float* out, long rows, int C,
                  long XY, int S, void* stream);

/* Backward of occf_dualpath_combine: dtokens[BP, Z+1, C] (slot Z zero-filled), dbev[BP, C], dweight[C], dbias[1];
 * the identity operand's gradient is dout itself.  workspace: occf_dualpath_combine_bwd_workspace floats. */
long occf_dualpath_combine_bwd_workspace(long BP, int C);
int occf_dualpath_combine_bwd(const float* tokens, const float* bev, const float* coeff_weight,
                              const float* coeff_bias, const float* dout, float* dtokens, float* dbev,
                              float* dweight, float* dbias, float* workspace, long BP, int Z, int C, void* stream);

/* Backward of occf_upsample_add w.r.t. the coarse operand (gather form, no atomics); the lateral operand's
 * gradient is dout itself.  dout[B, X2, Y2, Z2, C] -> dcoarse[B, X, Y, Z, C]. */
int occf_upsample_add_bwd(const float* dout, float* dcoarse, int B, int X, int Y, int Z, int X2, int Y2, int Z2,
                          int C, void* stream);

/* Backward of occf_point_sample_3d_fwd w.r.t. the volume (F.grid_sample's input gradient): dout[N, C, P] scattered
 * into dvol[N, C, X, Y, Z], which the CALLER zero-fills.  voxel_major_ld > 0: dvol is [X*Y*Z, voxel_major_ld] with
 * column n*C + c instead (voxel-major, the operand layout of the mask-logit contraction's backward). */
int occf_point_sample_3d_bwd(const float* dout, const float* pts, float* dvol, int N, int C, int X, int Y, int Z,
                             long P, int shared_pts, int align_corners, int border_padding, long voxel_major_ld,
                             void* stream);

/* Backward of occf_point_loss_rows_fwd: grad_rows[R, 4] = d(loss)/d(out) -> dlogits[R, P]. */
int occf_point_loss_rows_bwd(const float* logits, const float* targets, const float* grad_rows, float* dlogits,
                             int R, long P, void* stream);

/* Weight (and bias) gradient of occf_linear_*: dw[N, K] = dy[M, N]^T x[M, K], dbias[N] = column sums of dy (NULL
 * to skip); bf16 matrix cores with terms = 3 (fp32-class) or 1, or -- terms = 2 -- TWO fp16-piece products per
 * product (dy as ONE fp16 piece after a per-tensor power-of-two scale taken from max |dy| on the device, x as fp16
 * (hi, lo): ~2^-12 per dy element, weight gradients are leaf quantities); M split into slabs reduced in fixed order.
 * workspace: occf_linear_wgrad_workspace floats (0: none needed).  Small or odd shapes (M <= 1024, N or K not a
 * multiple of 4) run an exact fp32 kernel. */
long occf_linear_wgrad_workspace(long M, int N, int K);
int occf_linear_wgrad(const float* dy, const float* x, float* dw, float* dbias, float* workspace,
                      long workspace_floats, long M, int N, int K, long ldy, long ldx, int terms, void* stream);

/* Weight gradient of occf_conv3d_*_fwd in the tap-major layout dw[Cout, kX*kY*kZ*Cin]; x addressed by strides as in
 * the forward, dy[B*Xo*Yo*Zo, Cout] contiguous.  terms as occf_linear_wgrad, plus terms = 4: ONE product per product
 * -- dy (scaled) and x each as one fp16 piece -- on the G8 kernel's shapes (stride 1, 3^3, Z % 8 == 0, channels % 64
 * == 0, dense x); every other shape computes terms = 2. */
long occf_conv3d_wgrad_workspace(int B, int Xi, int Yi, int Zi, int Cin, int Cout, int kX, int kY, int kZ, int stride,
                                 int dil, int pad_x, int pad_y, int pad_z);
int occf_conv3d_wgrad(const float* dy, const float* x, float* dw_tapmajor, float* dbias, float* workspace,
                      long workspace_floats, int B, int Xi, int Yi, int Zi, int Cin, int Cout, int kX, int kY, int kZ,
                      int stride, int dil, int pad_x, int pad_y, int pad_z, long in_sb, long in_sx, long in_sy,
                      long in_sz, int terms, void* stream);

/* Data gradient of occf_conv3d_bf16_fwd (any stride / dilation): dy[B, Xo, Yo, Zo, Cout] contiguous ->
 * dx[B, Xi, Yi, Zi, Cin] contiguous; wt = the weight re-laid as [Cin, taps*Cout] (k = tap*Cout + co), pre-split.
 * Cout % 32 == 0.  (The 3^3 / stride-1 case is also served by occf_conv3x3x3_halo_fwd on the tap-flipped weight.) */
int occf_conv3d_bf16_dgrad(const float* dy, const uint16_t* wt_hi, const uint16_t* wt_lo, float* dx, int B, int Xi,
                           int Yi, int Zi, int Cin, int Cout, int kX, int kY, int kZ, int stride, int dil, int pad_x,
                           int pad_y, int pad_z, int terms, float* workspace, long workspace_floats, void* stream);

/* Backward of occf_window_attn_fwd (and of the attention core inside occf_swin_attn_fused_fwd): qkv / qkv_bias /
 * bias_table as in the forward, attn_out = the forward's output, dout its gradient.  dqkv[n_tok, 3C] (every row
 * written once); dqkv_bias[3C] receives ONLY the key / value gradients of zero-padded window positions (whose
 * qkv is the bias) and must be ZERO-FILLED by the caller -- the bias gradient of the projection itself is the
 * column sum of dqkv; dbias_table[(2*7-1)^2, heads].  workspace: occf_window_attn_bwd_workspace floats. */
long occf_window_attn_bwd_workspace(int B, int X, int Y, int S, int heads);
int occf_window_attn_bwd(const float* qkv, const float* qkv_bias, const float* bias_table, const float* attn_out,
                         const float* dout, float* dqkv, float* dqkv_bias, float* dbias_table, float* workspace,
                         int B, int X, int Y, int S, int C, int heads, int shift, void* stream);

/* Backward of occf_masked_xattn_fwd: out = the forward's output, dout its gradient; dq[B, Q, E], dk/dv[B, L, E].
 * Q <= 128, head_dim 32.  workspace: occf_masked_xattn_bwd_workspace floats. */
long occf_masked_xattn_bwd_workspace(int B, int Q, int L, int heads);
int occf_masked_xattn_bwd(const float* q, const float* k, const float* v, const uint8_t* blocked,
                          const int32_t* row_open, const float* out, const float* dout, float* dq, float* dk,
                          float* dv, float* workspace, int B, int Q, int L, int E, int heads, void* stream);

/* Backward of occf_msda3d_fwd.  dout[B, Nq, heads*head_dim]; dvalue[B, Nq, heads*head_dim] TOKEN-major (whatever
 * layout `value` has) and ZERO-FILLED by the caller (scatter with float atomics, as F.grid_sample's backward);
 * doffsets / dlogits in the forward's layouts with their own row strides (0 = dense).
 * workspace (occf_msda3d_bwd_workspace floats; 0 / NULL = plain atomic scatter): with head_dim 12 / 24 the value
 * gradient is accumulated in LDS tiles of the sampled level and reduced deterministically from per-tile slabs. */
long occf_msda3d_bwd_workspace(const int32_t* level_shapes, int num_levels, int B, int heads, int head_dim);
int occf_msda3d_bwd(const float* value, const float* sampling_offsets, const float* attn_logits, const float* dout,
                    float* dvalue, float* doffsets, float* dlogits, const int32_t* level_shapes, int num_levels, int B,
                    int Nq, int heads, int head_dim, int num_points, int value_head_major, long offsets_ld,
                    long logits_ld, long doffsets_ld, long dlogits_ld, float* workspace, long workspace_floats,
                    void* stream);

/* Backward of occf_deform_im2col (mmcv deformable_col2im + deformable_col2im_coord): dcol in the forward's column
 * layout -> dx[BN, H, W, C] (scatter with float atomics, ZERO-FILLED by the caller) and doffset in conv_offset's
 * layout (every element written). */
int occf_deform_col2im(const float* x, const float* offset, const float* dcol, float* dx, float* doffset, int BN,
                       int H, int W, int C, int K, int stride, int pad, int dil, int groups, int deform_groups,
                       void* stream);

/* DCNv2 backward (mmcv-full 1.4.0 `modulated_deformable_col2im` + `modulated_deformable_col2im_coord`, behind the
 * R101-DCN image backbone, occformer_nusc_r101_896x1600.py:78-79): as occf_deform_col2im with the forward's modulation
 * mask[BN, dg*K*K, Ho, Wo]; dmask of the same shape receives the gradient of the (already sigmoid-ed) modulation. */
int occf_modulated_deform_col2im(const float* x, const float* offset, const float* mask, const float* dcol, float* dx,
                                 float* doffset, float* dmask, int BN, int H, int W, int C, int K, int stride, int pad,
                                 int dil, int groups, int deform_groups, void* stream);

/* One launch for a TABLE of strided-gather + bf16-split jobs (the per-step "prepare weights" pass: tap-major,
 * transposed, tap-flipped layouts and (hi, lo) splits of every parameter).  table[(n + 1) * 16] int64 on the device, row
 * r = {in, f32_out or 0, hi_out, lo_out, first pair index of its span, dims[5], input strides[5] (elements, base offset
 * folded into `in`), mode}; row n carries the total span.  A pair = two consecutive outputs of the last dimension
 * (even).  Spans are whole SLOTS of 512 pairs (one workgroup each): mode 0 = ceil(pairs / 512) slots in output order;
 * mode 1 (the output's last dimension is the input's slowest: transposes) = ceil(M / 32) * ceil(N / 32) slots, one
 * 32 x 32 tile of out[M = d0 d1 d2 d3][N = d4] each, staged through LDS.  total_pairs = the total span. */
int occf_prep_weights(const int64_t* table, int n, long total_pairs, void* stream);

/* ------------------------------------------------------------------ input pipeline / evaluation ---- */

/* Replaces CreateDepthFromLiDAR.__call__ (P/datasets/pipelines/lidar2depth.py:15-87; no FFI in the reference: torch
 * ops on the host inside the data loader).  points[P, >=3] (row stride points_ld floats), cam[N, 36] =
 * inv(rots)[9] | trans[3] | intrins (3x3 in the first 9 of 16, or 4x4 when kitti) [16] | post_rots[:2,:2][4] |
 * post_trans[:2][2] | pad[2]; gt_depths[N, H, W] <- nearest valid LiDAR depth per pixel, 0 where none.
 * workspace: N*H*W uint32. */
int occf_lidar_depth_fwd(const float* points, long points_ld, const float* cam, float* gt_depths, uint32_t* workspace,
                         long P, int N, int H, int W, int kitti, void* stream);

/* Replaces SSCMetrics.update (P/utils/ssc_metric.py:62-175).  pred[B*V] int64 labels, or scores[B, C, V] (arg-max
 * over C taken here; apis/test.py:64); target[B*V] uint8 (255 = ignore); nonempty / nonsurface[B*V] uint8 or NULL.
 * counts[C*C + 3] int64 are ACCUMULATED: conf[t*C + p] over the semantic mask, then completion tp, fp, fn. */
int occf_ssc_confusion_fwd(const int64_t* pred, const float* scores, const uint8_t* target, const uint8_t* nonempty,
                           const uint8_t* nonsurface, int64_t* counts, long B, long V, int C, void* stream);

/* img_inputs producer, image half (P/datasets/pipelines/loading_nusc_imgs.py:57-64 img_transform_core = PIL
 * Image.resize / crop / FLIP_LEFT_RIGHT / rotate, :179-193 mmlabNormalize; no FFI in the reference: PIL in the data-loader
 * workers).  occf_image_resample_fwd: ONE pass of Pillow's separable antialiased resampling on uint8 [H][W][C] (host
 * pointers are not accepted: every pointer is device memory); bounds [out_size][2] = (first tap, taps), kk
 * [out_size][ksize] = 22-bit fixed-point coefficients (Resample.c precompute_coeffs + normalize_coeffs_8bpc, computed by
 * the caller in double); vertical = 0: out [H][out_size][C], 1: out [out_size][W][C]. */
int occf_image_resample_fwd(const uint8_t* in, uint8_t* out, const int32_t* bounds, const int32_t* kk, int ksize, int H,
                            int W, int C, int out_size, int vertical, void* stream);
/* crop (cx0, cy0, fW x fH, zero outside) -> horizontal flip -> rotate (rot_mode 0 none | 1 = 180 degrees | 2 = Pillow's
 * 16.16 fixed-point affine map, affine[6] on the HOST) -> out [3][fH][fW] = (pixel - mean) * stdinv per channel (mean /
 * stdinv [3] on the HOST; to_rgb swaps channels 0 and 2 first, mmcv imnormalize), canvas (may be NULL) uint8 [fH][fW][3]
 * = the un-normalised frame (results['canvas']). */
int occf_image_crop_rotate_normalize_fwd(const uint8_t* in, float* out, uint8_t* canvas, int Hn, int Wn, int cx0, int cy0,
                                         int fW, int fH, int flip, int rot_mode, const int64_t* affine, const float* mean,
                                         const float* stdinv, int to_rgb, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OCCFORMER_HIP_H */
