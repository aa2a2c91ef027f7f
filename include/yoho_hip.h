/*
 * yoho_hip.h - C ABI of libyoho_hip.so: MI355X (gfx950) kernels for YOHO's 60-rotation
 * group-equivariant descriptor path and the YOHO-O / YOHO-C transformation estimators.
 *
 * The reference (HpWang-whu/YOHO) is pure Python/PyTorch on this path and has no FFI of its
 * own; the entry points below are what a binding for that path replaces, one per reference
 * call site (file:line relative to the reference root, see DESIGN.md / INTEGRATION.md):
 *
 *   yoho_partI_forward      PartI_test.forward            utils/network.py:86-105,140-147
 *   yoho_group_mean_np      np.mean(feats, axis=-1)       tests/matcher.py:35-36
 *   yoho_nn_search          modified_knn_matcher.__call__ utils/knn_search.py:17-66,138-154
 *   yoho_mutual_nn          matcher_dual.match inner loop tests/matcher.py:37-48
 *   yoho_des2r              Batch_Des2R_torch             tests/extractor.py:74-78
 *   yoho_partII_forward     PartII_test.forward           utils/network.py:259-278
 *   yoho_hyp_from_quat      quat -> [R|t] loops           tests/extractor.py:187-199, utils/r_eval.py:94-110
 *   yoho_o_score            yohoo.ransac scoring loop     tests/estimator.py:330-336, :286-290
 *   yoho_c_ransac           yohoc.ransac loop body        tests/estimator.py:119-137, :55-70
 *   yoho_c_ransac_device    yohoc.ransac incl. sampling   tests/estimator.py:34-51,:119-137 (device RNG)
 *   yoho_range_status       (fp16 range guard, no counterpart: the reference computes in fp32)
 *   yoho_group_gather       60-fold FCGF feature gather   YOHO_testset.py:153-166
 *   yoho_group_scatter      kpts_f[:, :, i] = pci_f[inds]  simple_yoho/yoho_extract.py:38-39,52
 *   yoho_set_nn_grid             (speed hint, no counterpart)    voxel size of YOHO_testset.py:39-49 / simple_yoho/fcgf_feat.py:33-43
 *   yoho_partI_forward_pair      the two PartI passes of a pair  tests/extractor.py:37-62 (one pass per fragment there)
 *   yoho_des2r_indexed           feats[match[:,0]] + Des2R       tests/extractor.py:80-103
 *   yoho_partII_forward_indexed  batch_create + PartII_test      tests/extractor.py:125-141,178-186, utils/network.py:259-278
 *   yoho_gconv_layer             Comb_Conv / Residual_Comb_Conv  utils/network.py:35-62 (forward) and its autograd (training)
 *   yoho_load_fcgf / _voxelize / _forward / _forward_batch        fcgf_model/resunet.py:10-190, simple_yoho/fcgf_feat.py:33-54
 *   yoho_fcgf_voxelize_rotated / yoho_rotate_select               YOHO_testset.py:143-147,92, simple_yoho/yoho_extract.py:46-53
 *   yoho_group_transfer_batch                                     the feature-transfer body of those loops for the copies of a pass
 *   yoho_register_pair           one iteration of the pair loop  tests/evaluator.py:112-117 / :41-47 (matcher -> Des2R -> PartII + vote | YOHO-C)
 *   yoho_vote_order              np.random.shuffle(index)        tests/estimator.py:321-323 (seeded RandomState, host code)
 *   yoho_c_draw_np               the np.random.choice draws      tests/estimator.py:113-128 (numpy's legacy stream continued in C, host code)
 *
 * Conventions
 *   - return 0 on success, a negative YOHO_E* code on error; yoho_last_error() gives a
 *     thread-local message.  No C++ exception crosses the ABI.
 *   - every data pointer is a DEVICE pointer owned by the caller (contiguous, 16-byte
 *     aligned), except the yoho_load_* / yoho_ctx_create inputs which are HOST pointers.
 *   - calls are asynchronous on the given hipStream_t (pass NULL for the default stream; yoho_register_pair and
 *     yoho_range_status, which return results to the host, wait for it);
 *     the library allocates only its own workspace inside the ctx (which may synchronise
 *     the stream the first time a larger problem is seen).
 *   - a ctx belongs to one device and is used from one stream / thread at a time (it owns one workspace); different ctxs - also
 *     several on one device, also on different group tables - are independent: the library has no global mutable state.
 *   - tensors use the reference's layouts: group features (K,32,60) f32 with the group axis
 *     innermost, keypoints (K,3) f64, transforms (.,3,4) f64, indices int64.
 *   - environment: yoho_ctx_create is the ONLY place the library reads the environment (every variable once, into the new context);
 *     no other entry point does, so what a context computes and how is fixed at its creation and two contexts of a process may differ.
 *     Defaults of settings that also have a setter:   YOHO_GCONV, YOHO_PARTII (yoho_set_gconv_mode / yoho_set_partII_mode),
 *     YOHO_PARTI_CHUNK (yoho_set_partI_schedule), YOHO_NN=brute (yoho_set_nn_prefilter), YOHO_FCGF_SORT, YOHO_FCGF_CELLS,
 *     YOHO_FCGF_COORDS=hash (yoho_set_fcgf_sort).   A/B and diagnostic switches without a setter (struct yoho_env_switches in
 *     csrc/common.h; every value gives valid results):   YOHO_PARTII_TAIL=staged, YOHO_TRANSFER=staged, YOHO_XF_STEAL=0,
 *     YOHO_NN_SPLITS=<n>, YOHO_FCGF=f32 (takes effect at the context's yoho_load_fcgf), YOHO_FCGF_MAPS=full, YOHO_FCGF_NORM=staged, YOHO_FCGF_HEADS=staged, YOHO_PARTII_L1=3 (PartII's first layer on the 256 x 256-tile GEMM kernel),
 *     YOHO_WS_LIMIT_MB=<n> (workspace requests above n MiB fail with YOHO_ENOMEM as on an exhausted device: test hook for the recoveries),
 *     YOHO_SPCONV_DEBUG=<bits> (only in a -DYOHO_SPCONV_ABLATE build).   The timing experiments YOHO_PARTI_DEBUG / YOHO_FGEMM_DEBUG /
 *     YOHO_SPCONV_VAR exist only in the separate -DYOHO_EXPERIMENTS library, which no product code loads.
 *   - argument checks: a NULL context, a NULL required pointer, a negative count or a count beyond a stated capacity returns
 *     YOHO_EINVAL with a message naming the entry point (tests/test_gpu_abi.py drives every entry point that way); a count of 0 is
 *     valid wherever it is meaningful (no rows: nothing is launched).
 */
#ifndef YOHO_HIP_H
#define YOHO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YOHO_OK            0
#define YOHO_EINVAL       -1   /* bad argument */
#define YOHO_EHIP         -2   /* HIP runtime error (message has the hipError string) */
#define YOHO_ENOWEIGHTS   -3   /* forward called before yoho_load_* */
#define YOHO_ENOMEM       -4
#define YOHO_ERANGE       -5   /* a value left the fp16 range of the fp16x2 arithmetic (yoho_range_status) */

typedef struct yoho_ctx yoho_ctx;

/* host pointers into a checkpoint's tensors, torch layouts (conv: (Cout,Cin,1,K) row-major) */
typedef struct { const float *weight, *bias; } yoho_conv_w;
typedef struct { const float *gamma, *beta, *mean, *var; } yoho_bn_w;   /* BatchNorm2d eval, eps 1e-5 */

/* PartI_test state dict (utils/network.py:76-78) */
typedef struct {
    yoho_conv_w conv_in;                       /* PartI_net.Conv_in.0                   32->256, K=13 */
    yoho_bn_w   res_in_bn;   yoho_conv_w res_in;   /* SO3_Conv_layers.0.comb_layer_in.{0,2}  256->512 */
    yoho_bn_w   res_out_bn;  yoho_conv_w res_out;  /* SO3_Conv_layers.0.comb_layer_out.{0,2} 512->256 */
    yoho_bn_w   out_bn;      yoho_conv_w conv_out; /* Conv_out.comb_layer.{0,2}              256->32  */
} yoho_partI_weights;

/* PartII_test state dict (utils/network.py:228-241) */
typedef struct {
    yoho_bn_w   init_bn;     yoho_conv_w init;     /* Conv_init.comb_layer.{0,2}            128->256 */
    yoho_bn_w   res_in_bn;   yoho_conv_w res_in;   /* PartII_SO3_Conv_layers.0.comb_layer_in  256->512 */
    yoho_bn_w   res_out_bn;  yoho_conv_w res_out;  /* PartII_SO3_Conv_layers.0.comb_layer_out 512->256 */
    yoho_conv_w fc0;  yoho_bn_w fc0_bn;            /* PartII_To_R_FC.{0,1}  256->512, K=1 */
    yoho_conv_w fc1;  yoho_bn_w fc1_bn;            /* PartII_To_R_FC.{3,4}  512->128 */
    yoho_conv_w fc2;                               /* PartII_To_R_FC.6      128->4   */
} yoho_partII_weights;

const char* yoho_last_error(void);
const char* yoho_version(void);

/* Rotation.npy as (60,3,3) f32, Nei_Index_in_SO3_ordered_13.npy as (60,13) u8, 60_60.npy as (60,60) u8 */
int yoho_ctx_create(int device, const float* R60x9, const uint8_t* N60x13, const uint8_t* P60x60, yoho_ctx** out);
int yoho_ctx_destroy(yoho_ctx* ctx);

int yoho_load_partI(yoho_ctx* ctx, const yoho_partI_weights* w);
int yoho_load_partII(yoho_ctx* ctx, const yoho_partII_weights* w);

/* x (B,32,60) -> eqv (B,32,60) L2-normalised over channels, inv (B,32) (network's normalised
 * invariant feature; may be NULL), inv_np (B,32) = numpy-order fp32 mean of eqv over the group
 * axis, i.e. what the matcher consumes (may be NULL).  Any B >= 1. */
int yoho_partI_forward(yoho_ctx* ctx, const float* x, int B, float* eqv, float* inv, float* inv_np, void* stream);

/* both fragments of a scene pair in one descriptor pass without concatenating them first: rows [0,B0) of the outputs
 * belong to x0, rows [B0,B0+B1) to x1.  Default arithmetic mode only (YOHO_EINVAL otherwise), B0+B1 <= 16384. */
int yoho_partI_forward_pair(yoho_ctx* ctx, const float* x0, int B0, const float* x1, int B1, float* eqv, float* inv, float* inv_np,
                            void* stream);

/* out (B,32) = np.mean(eqv (B,32,60), axis=-1) bit-exactly (numpy pairwise order, fp32) */
int yoho_group_mean_np(yoho_ctx* ctx, const float* eqv, int B, float* out, void* stream);

/* for every row of src (Ns,D) the index (and distance, may be NULL) of the nearest row of
 * tgt (Nt,D), fp32, first minimum wins.  dist_type YOHO_DIST_L2: argmin_j sqrt(sum_f (s_f - t_jf)^2
 * + 1e-7) (pdist 'L2'); YOHO_DIST_SQUARE_L2: argmin_j sum_f (s_f - t_jf)^2 (pdist 'SquareL2').
 * D = 32 (descriptor matching, torch-CPU summation order) or D = 3. */
#define YOHO_DIST_L2        0
#define YOHO_DIST_SQUARE_L2 1
int yoho_nn_search(yoho_ctx* ctx, const float* src, int Ns, const float* tgt, int Nt, int D, int dist_type,
                   int64_t* idx, float* dist, void* stream);

/* mutual nearest neighbours of a (Na,32) in b (Nb,32): pairs (M,2) int64 in ascending a-index,
 * buffer must hold Na rows; *M_out (device int) receives M. */
int yoho_mutual_nn(yoho_ctx* ctx, const float* a, int Na, const float* b, int Nb,
                   int64_t* pairs, int* M_out, void* stream);

/* idx[m] = argmax_a sum_{f,g} d1[m,f,P[a,g]] * d2[m,f,g]; cor (M,60) may be NULL */
int yoho_des2r(yoho_ctx* ctx, const float* d1, const float* d2, int M, int64_t* idx, float* cor, void* stream);

/* inputs as the network receives them (after batch_create's 0<->1 exchange); NOT modified.
 * quat (M,4) unit quaternions. */
int yoho_partII_forward(yoho_ctx* ctx, const float* before_eqv0, const float* before_eqv1,
                        const float* after_eqv0, const float* after_eqv1, const int64_t* pre_idx,
                        int M, float* quat, void* stream);

/* HBM-resident variants for a caller that holds the full descriptor arrays and a match list: the rows are addressed in
 * place through index arrays instead of being gathered first.  i* point at the first index, consecutive indices are
 * `istride` elements apart (2 for the columns of an (M,2) int64 match array).
 *   yoho_des2r_indexed:          d1 = e1[i1[m]], d2 = e2[i2[m]]
 *   yoho_partII_forward_indexed: before_eqv0 = s0[i0[m]], before_eqv1 = s1[i1[m]], after_eqv0 = s2[i2[m]], after_eqv1 = s3[i3[m]];
 *                                default PartII arithmetic mode only (YOHO_EINVAL otherwise) */
int yoho_des2r_indexed(yoho_ctx* ctx, const float* e1, const int64_t* i1, const float* e2, const int64_t* i2, int istride, int M,
                       int64_t* idx, float* cor, void* stream);
int yoho_partII_forward_indexed(yoho_ctx* ctx, const float* s0, const int64_t* i0, const float* s1, const int64_t* i1, const float* s2,
                                const int64_t* i2, const float* s3, const int64_t* i3, int istride, const int64_t* pre_idx, int M,
                                float* quat, void* stream);

/* T[m] = [R | t], R = quat2mat_f32(quat[m]) * Rgroup_f32[idx[m]] (f64), t = k0[m] - R k1[m] */
int yoho_hyp_from_quat(yoho_ctx* ctx, const float* quat, const int64_t* idx, const double* k0,
                       const double* k1, int M, double* T, void* stream);

/* inlier counts of hypotheses T[order[h]], h < H, over M matches with threshold d; best_h /
 * best_count (device ints) = first strict maximum (best_count 0 -> no hypothesis beats 0);
 * counts (H) int32 may be NULL; order may be NULL (identity). */
int yoho_o_score(yoho_ctx* ctx, const double* k0, const double* k1, int M, const double* T,
                 const int64_t* order, int H, double d, int* best_h, int* best_count,
                 int32_t* counts, void* stream);

/* 3-point Kabsch per triple + inlier vote.  reflect (I) u8 may be NULL (proper rotations);
 * reflect[i] != 0 reproduces the reference's det = -1 solution for that iteration.
 * T_out (I,3,4) may be NULL; best_T (3,4); best_iter is 1-based, 0 = none. */
int yoho_c_ransac(yoho_ctx* ctx, const double* k0, const double* k1, int M, const int64_t* triples,
                  const uint8_t* reflect, int I, double d, double* best_T, int* best_iter,
                  int* best_count, double* T_out, int32_t* counts, void* stream);

/* YOHO-C with no host work (throughput mode of yohoc.ransac, tests/estimator.py:28-141): the histogram of dr_index over
 * the M matches, the weights n (n - .01)(n - .02) (:34-51), the two draws of every iteration (:119-128) from a
 * counter-based Philox4x32-10 stream keyed by `seed`, the 3-point Kabsch (proper rotation) and the inlier vote run on the
 * device; exactly max_iter iterations are scored.  keys0 / keys1 (.,3) f64 are addressed through the row indices
 * i0[m * istride] / i1[m * istride] (e.g. the two columns of a (M,2) match list, istride 2), or directly when i0 / i1 are
 * NULL.  best_T (3,4) f64, best_iter = 1-based index of the first best iteration (0: no inlier anywhere, best_T = eye;
 * 50001: fewer than two matches share a coarse rotation, the reference's 'no estimate' code), best_count = its inliers.
 * triples_out (max_iter,3) int64 receives the sampled matches, or NULL.  Deviation from the reference, stated: numpy's
 * MT19937 stream cannot be continued on the device, so for a given seed the triples are those of
 * oracle/yoho_oracle.py:yohoc_device_triples, not np.random's, and reflections (LAPACK's arbitrary det = -1 answers, see
 * yoho_c_ransac) are never scored.  The host-RNG parity mode is yoho_c_ransac. */
int yoho_c_ransac_device(yoho_ctx* ctx, const double* keys0, const int64_t* i0, const double* keys1, const int64_t* i1,
                         int istride, const int64_t* dr_index, int M, int max_iter, uint64_t seed, double d,
                         double* best_T, int* best_iter, int* best_count, int64_t* triples_out, void* stream);

/* ---- one scene pair in one call (the pair loop of tests/evaluator.py:112-117 / :41-47; yoho_amd/pipeline.py:run_pair composes the
 * same entries from Python).  Both fragments are already described: feat (n,32,60) FCGF group features (YOHO-O only, may be NULL
 * for YOHO-C), eqv (n,32,60) / inv (n,32) = yoho_partI_forward's eqv / inv_np, keys (n,3) f64; all device pointers.
 *   mutual NN of inv0 / inv1 -> Des2R -> YOHO_ESTIMATOR_O: vote order = numpy's RandomState(seed & 0xFFFFFFFF).shuffle(arange(M))
 *   (tests/estimator.py:321-323; yoho_vote_order), PartII + [R|t] for the min(max_iter, M) matches the vote reads (selected != 0)
 *   or for every match (selected = 0: the reference's Trans_pre stage), inlier vote with threshold inlier_dist;
 *   YOHO_ESTIMATOR_C: yoho_c_ransac_device with max_iter iterations from the Philox stream `seed`.
 * Waits for `stream` twice (match count, winner) and fills *out on the host.  YOHO-O needs the default PartII arithmetic mode
 * (YOHO_EINVAL otherwise); out->range_flag != 0 reports (and clears) the PartII fp16 range flag: the result is then not to be
 * trusted and the caller repeats the pair through the staged entries after yoho_set_partII_mode(ctx, 1). */
#define YOHO_ESTIMATOR_O 0
#define YOHO_ESTIMATOR_C 1
typedef struct yoho_pair_result {
    double trans[12];      /* (3,4) row-major [R|t] of the winner; not meaningful when best_count == 0 (the reference returns eye(4)) */
    int32_t matches;       /* M */
    int32_t best_h;        /* YOHO-O: position of the winner in the vote order; YOHO-C: its 1-based iteration (both: 'recalltime') */
    int32_t best_count;    /* inliers of the winner; 0 = no estimate */
    int32_t hypotheses;    /* hypotheses scored */
    int32_t range_flag;
    int32_t reserved;
} yoho_pair_result;
int yoho_register_pair(yoho_ctx* ctx, const float* feat0, const float* feat1, const float* eqv0, const float* eqv1, const float* inv0,
                       const float* inv1, const double* keys0, const double* keys1, int n0, int n1, int estimator, int max_iter,
                       double inlier_dist, uint64_t seed, int selected, yoho_pair_result* out, void* stream);
/* order[0..M) = numpy.random.RandomState(seed).shuffle applied to arange(M) (host memory; no device work, no context) */
int yoho_vote_order(uint32_t seed, int M, int64_t* order);
/* The sampling half of the reference's YOHO-C loop (tests/estimator.py:113-128) on numpy's legacy MT19937 stream, host memory only (no
 * device work, no context): per accepted iteration `np.random.choice(range(60), p=prob)` then `np.random.choice(bucket, 3)`, a
 * rotation whose bucket has fewer than two matches consuming its draw and being skipped, until max_iter iterations or 50001 draws.
 *   mt_key[624], *mt_pos     np.random.get_state()[1:3]; advanced in place, so np.random.set_state continues the reference's stream
 *   prob[60]                 DR_statictic's normalised weights as handed to choice (:34-51)
 *   bucket_start[61], bucket_members   matches of rotation r, ascending: bucket_members[bucket_start[r] .. bucket_start[r+1])
 *   triples (max_iter,3)     receives the sampled match indices; *n_triples accepted iterations, *n_draws rotation draws made */
int yoho_c_draw_np(uint32_t* mt_key, int* mt_pos, const double* prob, const int64_t* bucket_start, const int64_t* bucket_members,
                   int max_iter, int64_t* triples, int* n_triples, int* n_draws);

/* one group element of the 60-fold gather: rotate keys (K,3) f64 by Rg (3x3 f64, host ptr),
 * 1-NN among pts (n,3) f32 in f64, copy feat (n,32) rows into out[:, :, g] of (K,32,60);
 * nn_idx (K) int64 may be NULL. */
int yoho_group_gather(yoho_ctx* ctx, const double* keys, int K, const float* pts, const float* feat,
                      int n, int g, const double* Rg_host, float* out, int64_t* nn_idx, void* stream);

/* out[k, :, g] = feat[idx[k], :] for k < K: the row transfer of one group element once the nearest neighbours are known
 * (simple_yoho/yoho_extract.py:38-39,52: kpts_f[:, :, i] = pci_f[inds]).  feat (n,32) f32, idx (K) int64, out (K,32,60) f32. */
int yoho_group_scatter(yoho_ctx* ctx, const float* feat, int n, const int64_t* idx, int K, int g, float* out, void* stream);

/* ---- training path (reference train/trainer.py on utils/network.py *_train): one (1,13) group-conv layer on
 * device-resident parameters, training layout.  weight (cout,cin,1,13), bias (cout) or NULL: device pointers.
 *   transpose = 0:  y (B,cout,60) = bias + conv(x (B,cin,60))            (utils/network.py:46-52 + Conv2d(cin,cout,(1,13)))
 *   transpose = 1:  y (B,cin,60)  = data gradient of that layer for the output gradient x (B,cout,60); bias ignored.
 * The weight / bias gradient of the layer is yoho_gconv_wgrad below. */
int yoho_gconv_layer(yoho_ctx* ctx, const float* x, int B, int cin, int cout, const float* weight, const float* bias, int transpose,
                     float* y, void* stream);

/* weight (and bias) gradient of the same layer: dW (cout,cin,1,13) = sum over (b, g) of dy[b,o,g] * x[b,c,N[g,k]], db (cout) =
 * sum of dy over (b, g) or NULL; x (B,cin,60), dy (B,cout,60) device pointers, cin and cout multiples of 32.  Together with
 * yoho_gconv_layer(transpose = 1) this is the whole backward pass of Conv2d(cin,cout,(1,13)) on the neighbour gather
 * (the reference's autograd of utils/network.py:46-52,18). */
int yoho_gconv_wgrad(yoho_ctx* ctx, const float* x, const float* dy, int B, int cin, int cout, float* dW, float* db, void* stream);

/* BatchNorm2d (train or eval statistics) + ReLU in front of a conv (utils/network.py:16-17,28-29,33-34), on (B,C,60) device tensors.
 *   yoho_bn_stats          mean (C), biased var (C) over the B x 60 values of every channel (f64 sums);
 *   yoho_bn_relu_apply     y = relu(x * scale[c] + shift[c]), scale = gamma * rsqrt(var + eps), shift = beta - mean * scale;
 *   yoho_bn_relu_backward  dy -> dx, dgamma, dbeta through ReLU and the normalisation (batch_stats = 1: the statistics depend on x
 *                          as in training; 0: running statistics, the normalisation is a fixed affine map). */
int yoho_bn_stats(yoho_ctx* ctx, const float* x, int B, int C, float* mean, float* var, void* stream);
int yoho_bn_relu_apply(yoho_ctx* ctx, const float* x, int B, int C, const float* scale, const float* shift, float* y, void* stream);
int yoho_bn_relu_backward(yoho_ctx* ctx, const float* x, const float* y, const float* dy, int B, int C, const float* gamma,
                          const float* mean, const float* rstd, int batch_stats, float* dx, float* dgamma, float* dbeta, void* stream);

/* ---- FCGF backbone (reference fcgf_model/resunet.py ResUNet2 family, simple_yoho/fcgf_feat.py) ------------------------
 * Sparse 3-D ResUNet forward pass, fp32.  channels / tr_channels = CHANNELS / TR_CHANNELS of the model class (index 0
 * unused), e.g. ResUNetBN2C: {0,32,64,128,256} / {0,64,64,64,128}.  tensors: host pointers to the state_dict entries in
 * the model's registration order with the num_batches_tracked entries left out (conv1.kernel, norm1.bn.{weight,bias,
 * running_mean,running_var}, block1.conv1.kernel, ...; yoho_amd.weights.fcgf_spec()). */
typedef struct yoho_fcgf_config {
    int channels[5];
    int tr_channels[5];
    int out_channels;
    int conv1_kernel_size;
    int in_channels;          /* 1: the input feature is a column of ones (fcgf_feat.py:41) */
    int normalize_feature;
} yoho_fcgf_config;
int yoho_load_fcgf(yoho_ctx* ctx, const yoho_fcgf_config* cfg, const float* const* tensors, int ntensors);
/* fcgf_feat.py:33-43: voxel = floor(p / voxel_size); sel = index of the first point of every voxel, ascending;
 * coords = their integer voxel coordinates.  pts (n,3) f64, sel (n) i64, coords (n,3) i32 device buffers sized for n;
 * *count (host) receives the number of voxels (the call synchronises the stream). */
int yoho_fcgf_voxelize(yoho_ctx* ctx, const double* pts, int n, double voxel_size, int64_t* sel, int32_t* coords, int* count,
                       void* stream);
/* the same on a rotated copy of the cloud, p' = R p in f64 (R: 9 doubles, row major, HOST pointer), without materialising it:
 * one of the 60 rotated copies YOHO_testset.py:143 / simple_yoho/yoho_extract.py:47 feed to the backbone.  pts_sel (n,3) f32
 * (may be NULL) receives the rotated selected points, i.e. pcd[sel] cast to float (YOHO_testset.py:92). */
int yoho_fcgf_voxelize_rotated(yoho_ctx* ctx, const double* pts, int n, const double* R, double voxel_size, int64_t* sel,
                               int32_t* coords, float* pts_sel, int* count, void* stream);
/* yoho_fcgf_voxelize_rotated for nb (1..64) rotations of the same cloud in one call: R_host holds nb row-major 3x3 matrices, the
 * outputs hold n rows per copy - sel (nb, n), coords (nb, n, 3), pts_sel (nb, n, 3) or NULL - of which the first counts[b] of copy b
 * are valid.  The copies' stages are queued back to back and the counts come back with one read-back (the 60-rotation loops of
 * YOHO_testset.py:143-147 / simple_yoho/yoho_extract.py:46-53 in groups). */
int yoho_fcgf_voxelize_rotated_batch(yoho_ctx* ctx, const double* pts, int n, const double* R_host, int nb, double voxel_size,
                                     int64_t* sel, int32_t* coords, float* pts_sel, int* counts, void* stream);

/* out (m,3) f32 = (float)(R pts[sel[i]]) with the arithmetic of yoho_fcgf_voxelize_rotated (R may be NULL: plain gather + cast):
 * the rotated keypoints of the feature transfer. */
int yoho_rotate_select(yoho_ctx* ctx, const double* pts, const double* R, const int64_t* sel, int m, float* out, void* stream);

/* The NN feature transfer of nb (1..64) rotated copies of one cloud in one call - the body of the 60-rotation loop of
 * simple_yoho/yoho_extract.py:46-53 / YOHO_testset.py:143-166 for the copies of one backbone pass.  For copy b:
 *   q = (float)(R_b pts[kidx])                          (yoho_rotate_select; R_host = nb row-major 3x3 f64 matrices, host)
 *   idx = argmin_j |q - ds[b][j]|^2, fp32 'SquareL2'    (yoho_nn_search with D = 3: through the hash grid when yoho_set_nn_grid is set)
 *   out[:, :, g0 + b] = feat[b][idx]                    (yoho_group_scatter; out (K,32,60) f32)
 * ds / feat: HOST arrays of nb device pointers ((m[b],3) and (m[b],32) f32), m: host array of row counts; q_scratch (K,3) f32 and
 * idx_scratch (K) int64 are caller-owned device scratch.  Same results as the three calls per copy, without their 3 nb
 * host round trips through the binding. */
int yoho_group_transfer_batch(yoho_ctx* ctx, const double* pts, const int64_t* kidx, int K, const double* R_host, int nb,
                              const float* const* ds, const float* const* feat, const int* m, int g0, float* out,
                              float* q_scratch, int64_t* idx_scratch, void* stream);
/* resunet.py:141-190 + the final normalisation of fcgf_feat.py:48.  coords (n,3) i32 distinct voxels, out (n,out_channels). */
int yoho_fcgf_forward(yoho_ctx* ctx, const int32_t* coords, int n, float* out, void* stream);
/* several clouds in one pass (the 60 rotated copies of a fragment, or the reference's DataLoader batch, YOHO_testset.py:172-180):
 * coords = the clouds' voxel rows one after the other, offsets (host, nb+1 entries, offsets[0] = 0) their row ranges,
 * nb <= 64.  Same result per cloud as nb separate yoho_fcgf_forward calls.  The workspace is sized from the voxel count; clouds that are
 * sparse in large bounding boxes need more for their occupancy bitmaps, which the library only learns once the pass has started: it
 * then restarts the pass once on a workspace grown by that much (and keeps it), or runs the hash-table coordinate maps if the
 * allocation fails - the caller sees one call with the same result either way, never YOHO_ENOMEM for a pass that fits the device. */
int yoho_fcgf_forward_batch(yoho_ctx* ctx, const int32_t* coords, const int32_t* offsets, int nb, float* out, void* stream);

/* PartI group-conv formulation: 0 = direct 13-tap conv on fp32 MFMA (v_mfma_f32_32x32x2_f32), 1 = direct conv with an
 * fp32-accurate 3-way bf16 split on v_mfma_f32_32x32x16_bf16 (6 products per term), 2 = group-Fourier domain conv
 * (244 instead of 780 slab products per chunk) on fp32 MFMA, 3 = direct conv with a 2-way fp16 split on
 * v_mfma_f32_32x32x16_f16 (3 products per term, error <= 3*2^-22 per product; activations must stay below 4094 in
 * magnitude, beyond that the result is inf/NaN), 4 = group-Fourier domain with every layer as five dense irrep GEMMs on
 * the fp16x2 split MFMA and fp16x2 transform kernels between them (default); 5 / 6 = the same arithmetic with the other two GEMM
 * blockings; 7 (opt-in, YOHO_GCONV=fgemm8) = 4 with the two correction products of the split of the two large layers evaluated in fp8
 * e4m3 on v_mfma_scale_f32_32x32x64_f8f6f4 (2 / 3 of the matrix time of those layers: ~7 % per pair; descriptors within ~1e-5 of the fp32
 * reference instead of ~1e-6, i.e. nearest-neighbour near-ties may resolve differently from the reference's).  All meet the 1e-4 parity tolerance. */
int yoho_set_gconv_mode(yoho_ctx* ctx, int mode);

/* 3-D nearest-neighbour searches (yoho_nn_search with D = 3, yoho_group_gather) through a uniform hash grid of the given
 * cell size instead of brute force; 0 switches back.  The answers are identical for ANY cell size (queries the grid
 * cannot settle within 2 cells are redone by brute force), the hint only decides the speed: pass the voxel size the
 * target cloud was down-sampled with (YOHO_testset.py:39-49 / simple_yoho/fcgf_feat.py:33-43), where every query has a
 * target point within a voxel diagonal. */
int yoho_set_nn_grid(yoho_ctx* ctx, double cell);

/* yoho_mutual_nn on large sets (Na * Nb >= 2^20): 1 (default) = Gram matrix on the fp16 MFMA as a pre-filter, then the exact
 * explicit-difference distance of every candidate within a proven error band of its row / column minimum; 0 = brute force.  The
 * match list is identical either way (the deciding arithmetic is always utils/knn_search.py:17-20's); inputs the pre-filter
 * cannot take (non-finite values, magnitudes beyond the fp16 range) go to the brute-force kernels by themselves. */
int yoho_set_nn_prefilter(yoho_ctx* ctx, int enable);

/* FCGF backbone, internal row orders (outputs are bit-identical either way, rows come back in the caller's order).
 * Coordinate maps: by default every level's map is a rank-ordered occupancy bitmap (row = prefix popcount; rows of every level in
 * brick order of the bitmap) whenever every cloud of the pass fits one (< 2^24 words) and the rows are distinct voxels; otherwise, or
 * with bit 2 of cell_sort set (cell_sort | 4, YOHO_FCGF_COORDS=hash), hash tables over packed voxel keys with rows in first-occurrence
 * order as MinkowskiEngine's CPU manager numbers them - the options below then apply:
 *   cell_sort   - 0 never, 1 (default) for passes of at least 2^18 voxels, 2 always: the level-0 rows of a pass are grouped
 *                 by 8^3-voxel cell (Morton order of the cells), so the rows a workgroup gathers are shared by its output
 *                 rows and stay in the L2 (the sort pays for itself only on large passes);
 *   parity_sort - (default on) the transposed convolutions walk the rows of their output level sorted by the parity class of the
 *                 coordinates (a row is reached from 1, 2, 4 or 8 of the 27 offsets, the same for a whole class) and every
 *                 tile skips the offsets none of its rows reaches (skipped terms are exact zeros). */
int yoho_set_fcgf_sort(yoho_ctx* ctx, int parity_sort, int cell_sort);

/* PartII group-conv layers: 0 = fp32 MFMA, 1 = bf16x3 split MFMA, 2 = fp16x2 split MFMA (first layer in the
 * group-Fourier domain, 13-rotation cone layer direct, last layer as one dense product at the identity), 3 = 2 with the
 * 13-rotation cone layer as ONE implicit GEMM over (tap, channel) whose B operand stages are picked per tap from the 45 cone
 * elements (YOHO_PARTII=cgemm), 4 = 3 with the two correction products of the fp16 split on the fp8 matrix pipe
 * (YOHO_PARTII=cgemm8; quaternions ~1e-5 from the fp32 reference instead of ~1e-6, tolerance 1e-4; PartII feeds no index decision). */
int yoho_set_partII_mode(yoho_ctx* ctx, int mode);

/* fp16 range guard.  The default arithmetic (PartI mode 4, PartII mode 2; also PartI mode 3) keeps activations and
 * Fourier coefficients as fixed power-of-two multiples in fp16 planes (|activation| < 4094, |coefficient| < 16376).
 * Every kernel that writes such planes raises a device-side flag when a value falls outside; nothing else in the
 * forward calls changes (they stay asynchronous, and their outputs are then not to be trusted).
 * yoho_range_status waits for `stream`, reports and clears the flag of every network whose out pointer is given (a NULL out
 * pointer leaves that network's flag pending for a later call) and returns YOHO_ERANGE if a reported flag was set, 0 otherwise; the caller then repeats the pass after yoho_set_gconv_mode(ctx, 1) /
 * yoho_set_partII_mode(ctx, 1) (bf16x3 planes carry the fp32 exponent range) - yoho_amd/hip.py does exactly that.
 * The reference computes in fp32 throughout (utils/network.py:12-105, 259-278), so it has no counterpart. */
int yoho_range_status(yoho_ctx* ctx, int* partI_overflow, int* partII_overflow, void* stream);

/* Schedule of the PartI pass in the default arithmetic mode (no counterpart in the reference, whose extractor walks batches of
 * test_batch_size keypoints breadth-first, tests/extractor.py:51-59).  chunk_kp = 0 (default): every layer sweeps all keypoints of
 * the pass.  chunk_kp > 0 (rounded up to a multiple of 256): depth-first - the pass is cut into chunks of chunk_kp keypoints, each
 * running head -> 4 irrep GEMMs + 3 transforms -> tail on its own workspace slice, so a chunk's intermediates (0.5 MB per
 * keypoint) can stay in the 256 MB Infinity Cache between layers.  streams = 2: chunks alternate between the caller's stream and
 * a stream owned by the context, forked from / joined into the caller's stream with events, so the call keeps its contract
 * (asynchronous, ordered on the caller's stream).  Results are bit-identical for every schedule in every arithmetic mode except the
 * opt-in mode 7 ('fgemm8'): its fp8 correction planes are scaled by one amax word per launch, so there a keypoint's bits depend on
 * which keypoints share its launch (differences within the mode's 1e-5 of the fp32 reference). */
int yoho_set_partI_schedule(yoho_ctx* ctx, int chunk_kp, int streams);

/* timing hook for bench.py: device time (ms) of the stages of the last yoho_partI_forward pass, measured with hipEvents
 * on the launch streams.  which 0..3: the four group-conv launches; 4: head; 5: tail (inverse transform + finalize);
 * 6: the three inter-layer transform launches together, 7 / 8 / 9 each of them; 10: inverse transform; 11: finalize
 * (a chunked pass reports the sums over its chunks); 12: the whole pass, first launch to last. */
int yoho_set_profiling(yoho_ctx* ctx, int enable);
int yoho_get_kernel_ms(yoho_ctx* ctx, int which, float* ms);

/* measurement hook for bench.py's `fcgf` leg: phase profile of the entries that are many launches long (the raw-cloud path
 * yoho_fcgf_voxelize* -> yoho_fcgf_forward* -> yoho_group_transfer_batch; the reference has no counterpart, its extractor is timed
 * from outside, simple_yoho/yoho_extract.py:57-77).  With the profile enabled those entries record events on their launch stream at
 * phase boundaries; yoho_phase_read waits for `stream`, returns the accumulated milliseconds per category (ms[16]), the fp16 MFMA
 * flops the category's launches issued (flops[16], counted at launch; null to skip), the number of such launches (launches[16]), and
 * starts the accumulation again.  Categories: 0 voxelisation, 1 coordinate maps (hash tables, strided maps, cell sort, bounding
 * boxes), 2 kernel maps (3^3 / strided / transposed maps, occupancy bitmaps, parity orders), 3 first convolution, 4..7 the 3^3
 * stride-1 convolutions of level 0..3, 8..10 the strided convolution into level 1..3, 11..13 the transposed convolution out to level
 * 0..2, 14 the 1x1 heads + normalisation, 15 nearest-neighbour feature transfer. */
int yoho_phase_profile(yoho_ctx* ctx, int enable);
int yoho_phase_read(yoho_ctx* ctx, double* ms, double* flops, double* launches, void* stream);

/* measurement hook for bench.py: the shader clock the part actually sustains while other streams are loaded.  One wave spins for
 * `microseconds` of the constant-rate wall counter and writes out[0] = shader cycles elapsed (s_memtime), out[1] = wall-counter
 * ticks elapsed, out[2] = the wall counter's rate in kHz (device int64[3]); shader MHz = out[0] / out[1] * out[2] / 1000.  Launch
 * it on a stream of its own beside the kernels under test. */
int yoho_clock_probe(yoho_ctx* ctx, int microseconds, long long* out3, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* YOHO_HIP_H */
