/*
 * epipolar_hip.h -- C ABI of libepipolar_hip.so (MI355X / gfx950).
 *
 * The reference (mkocabas/EpipolarPose) is pure Python/PyTorch and has no FFI of its own; the drop-in
 * boundary is the set of Python callables listed in SURVEY.md section 8(b).  This header declares the
 * device entry points those callables bind to (through ctypes).  Each entry point
 * names the reference code it replaces (path:line under the reference repository).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; the caller owns all memory;
 *   - nothing is allocated, freed or synchronised inside; work is enqueued on `stream` (a hipStream_t
 *     passed as void*, NULL = default stream); the library holds no global state and is re-entrant;
 *   - return value: EPI_OK (0) or an epi_status error code (> 0); epi_status_string() names it;
 *   - tensors are dense, row-major in the stated index order.
 */
#ifndef EPIPOLAR_HIP_H
#define EPIPOLAR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* epi_stream_t;

enum epi_status {
    EPI_OK = 0,
    EPI_ERR_INVALID_ARGUMENT = 1,   /* NULL pointer, non-positive size, bad enum */
    EPI_ERR_UNSUPPORTED = 2,        /* shape/dtype combination the kernels do not cover */
    EPI_ERR_WORKSPACE = 3,          /* workspace too small */
    EPI_ERR_LAUNCH = 4              /* hipGetLastError() != hipSuccess after a launch */
};

enum epi_dtype { EPI_F32 = 0, EPI_BF16 = 1, EPI_F64 = 2 };
enum epi_layout {
    EPI_NCHW = 0,   /* logits[b][j*D+d][h][w]  -- the reference layout (integral_loss.py:52)     */
    EPI_NHWC = 1    /* logits[b][h][w][j*D+d]  -- torch channels_last / the MFMA head's output  */
};
enum epi_loss_kind { EPI_LOSS_L1 = 0, EPI_LOSS_L2 = 1, EPI_LOSS_SMOOTH_L1 = 2 };

const char* epi_version(void);
const char* epi_status_string(int status);

/* Deterministic mode -- the reference's CUDNN.DETERMINISTIC key (lib/core/config.py:21, scripts/train.py:80: torch.backends.cudnn.deterministic).
 * on = 1: every cross-workgroup floating-point sum of the library runs in a fixed order (no fp32 atomics): the GEMM launches withhold their fused
 * BatchNorm statistics / BatchNorm-backward sums (*_done = 0: the caller's own pass runs), and those passes write per-workgroup partial sums to a
 * library-owned 16 MB scratch PER (device, stream) (allocated at the first use on that pair) that a second kernel adds in index order -- launches
 * of one stream are ordered and never share the scratch in time, launches of different streams have their own.  on = 0: atomics (default).
 * on < 0: query.  Returns the previous setting (-1: the scratch could not be allocated).  Bit-identical reruns of a training step:
 * tests/test_hip_deterministic.py. */
int epi_set_deterministic(int on);

/* ------------------------------------------------------------------------------------------------
 * Integral regression (soft-argmax) -- replaces lib/core/integral_loss.py:49-86
 * (softmax_integral_tensor + generate_3d_integral_preds_tensor): global softmax over each joint's
 * D*H*W voxels and the 3-D expectation, in ONE pass over the logits.
 *   xyz      [B][3J]  (x,y,z per joint), x = E[w]/W - 0.5, y = E[h]/H - 0.5, z = E[d]/D - 0.5
 *   row_max  [B*J]    max logit of each row      (saved for the backward pass)
 *   row_sum  [B*J]    sum exp(logit - row_max)   (saved for the backward pass)
 * workspace: epi_softargmax3d_workspace_bytes() bytes.
 * ------------------------------------------------------------------------------------------------ */
size_t epi_softargmax3d_workspace_bytes(int B, int J, int D, int H, int W);

int epi_softargmax3d_fwd(const void* logits, int dtype, int layout, int B, int J, int D, int H, int W,
                         float* xyz, float* row_max, float* row_sum,
                         void* workspace, size_t workspace_bytes, epi_stream_t stream);

/* Backward of the above contracted with grad_xyz [B][3J] (what autograd produces in the reference
 * through integral_loss.py:71-86): one read of the logits, one write of dlogits (same dtype/layout).
 *   dlogit_i = p_i * ( gx*(w_i/W - (x+.5)) + gy*(h_i/H - (y+.5)) + gz*(d_i/D - (z+.5)) ) * grad_scale
 * grad_scale: device scalar (upstream d loss) or NULL (= 1). */
int epi_softargmax3d_bwd(const void* logits, int dtype, int layout, int B, int J, int D, int H, int W,
                         const float* row_max, const float* row_sum, const float* xyz,
                         const float* grad_xyz, const float* grad_scale,
                         void* dlogits, epi_stream_t stream);
/* The same, and in the same pass the per-channel sums of the gradient it writes (channels-last logits only): the bias gradient of the final 1x1
 * convolution (pose3d_resnet.py:116-122; autograd's `grad_output.sum((0, 2, 3))`), which otherwise re-reads the whole gradient.
 *   col_sums [J*D] f32, ZERO on entry.  *col_sums_done = 1: col_sums[c] = sum over batch and pixels of dlogits[.., c] as stored (rounded to the
 *   logits' dtype); 0: col_sums untouched (NCHW, a depth extent the kernel's thread mapping does not cover, deterministic mode -- the sums are
 *   fp32 atomics): use epi_column_sums_bf16. */
int epi_softargmax3d_bwd_colsums(const void* logits, int dtype, int layout, int B, int J, int D, int H, int W,
                                 const float* row_max, const float* row_sum, const float* xyz,
                                 const float* grad_xyz, const float* grad_scale,
                                 void* dlogits, float* col_sums, int* col_sums_done, epi_stream_t stream);

/* Weighted joint-location loss value and its gradient w.r.t. the predicted coordinates -- replaces
 * lib/core/integral_loss.py:7-47 (weighted_mse_loss / weighted_l1_loss / weighted_smooth_l1_loss),
 * including norm=True (whole-tensor L1 normalisation, :9-11) and the sum/len(input) reduction (:16).
 *   pred, target, weight, grad_pred: [B][n]  (n = 3J);  loss: device scalar;  grad_pred may be NULL. */
int epi_joint_loss(const float* pred, const float* target, const float* weight, int B, int n,
                   int kind, int norm, int size_average, float* loss, float* grad_pred, epi_stream_t stream);

/* Hard arg-max of each row -- the device half of lib/core/inference.py:12-40 (get_max_preds):
 * first-maximum flat index (NumPy argmax tie rule, NaN counts as maximum) and the maximum.
 *   x [rows][n] (dtype f32 or bf16);  idx [rows] int64;  val [rows] f32
 * workspace: epi_argmax_workspace_bytes(rows, n). */
size_t epi_argmax_workspace_bytes(int rows, int n);
int epi_argmax_rows(const void* x, int dtype, int rows, int n, int64_t* idx, float* val,
                    void* workspace, size_t workspace_bytes, epi_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Self-supervision geometry -- replaces lib/utils/img_utils.py:141-243, lib/utils/triangulation.py,
 * lib/utils/prep_h36m.py:170-204.  A batch of B = V*G samples is view-major: sample (view v, group g)
 * sits at index v*G + g (img_utils.py:194-199: first half of the batch = view 1, second half = view 2).
 * All per-sample camera / crop arrays are float64, as default_collate produces them from the
 * reference's meta dict (lib/dataset/h36m.py:73-86).
 * ------------------------------------------------------------------------------------------------ */
typedef struct epi_view_meta {
    const double* center_x;   /* [B] crop centre in the original image (px)                         */
    const double* center_y;   /* [B]                                                                */
    const double* width;      /* [B] crop box width before augmentation                             */
    const double* height;     /* [B]                                                                */
    const double* scale;      /* [B] scale augmentation                                             */
    const double* rot;        /* [B] rotation augmentation, degrees                                 */
    const double* R;          /* [B][3][3] world->camera rotation                                   */
    const double* T;          /* [B][3]    camera centre in the world (X_c = R (X - T))             */
    const double* f;          /* [B][2]    focal lengths                                            */
    const double* c;          /* [B][2]    principal point                                          */
    const double* P;          /* [B][3][4] projection matrix K [R | -R T] (cameras.py:126-131)      */
} epi_view_meta;

/* img_utils.py:141-155,171-185 (trans_coords_from_patch_to_org_3d): soft-argmax output -> original
 * image pixels + depth in mm.  xyz [B][3J] f32 (normalised, as epi_softargmax3d_fwd writes it);
 * kps_img [B][J][3] f64 (u, v, z_mm). patch = 256, rect3d = 2000 in the reference (:178-179). */
int epi_decode_to_image(const float* xyz, int B, int J, const epi_view_meta* meta_host,
                        double patch_w, double patch_h, double rect3d, double* kps_img, epi_stream_t stream);

/* Triangulators.  kps [B][J][kps_stride] (first two entries = u, v), P [B][3][4]; dtype EPI_F64 or
 * EPI_F32 selects storage AND arithmetic of kps/P/X.  X [G][J][3]; status [G][J] int32 or NULL.
 *   iterls: triangulation.py:104-181 (iterative_LS_triangulation; V=2 is the reference, V>2 the same
 *           iteration with one depth per view).  status: 1, 0, -1, -2, -3 as the reference (:176-179).
 *   ls:     triangulation.py:34-97  (linear_LS_triangulation), V views.
 *   dlt:    triangulation.py:8-27   (linear_eigen_triangulation == cv2.triangulatePoints), 2Vx4
 *           homogeneous system, smallest right-singular vector by one-sided Jacobi in float64;
 *           status = 1 where max|X| <= 1e16 (:25). */
int epi_triangulate_iterls(const void* kps, int kps_stride, const void* P, int dtype, int G, int V, int J,
                           double tolerance, int max_iter, void* X, int32_t* status, epi_stream_t stream);
int epi_triangulate_ls(const void* kps, int kps_stride, const void* P, int dtype, int G, int V, int J,
                       void* X, int32_t* status, epi_stream_t stream);
int epi_triangulate_dlt(const void* kps, int kps_stride, const void* P, int dtype, int G, int V, int J,
                        void* X, int32_t* status, epi_stream_t stream);
/* Bulk launches (>= 256 (group, joint) items, 16-byte aligned kps / P / X) of the linear solve (ls) run a staged kernel:
 * the projection matrices of a workgroup's groups through LDS with 16-byte loads, one (u, v) vector load per view, 16-byte result stores --
 * the same arithmetic as the per-item kernel.  epi_triangulate_staged: 0 always the per-item kernel, 1 the default above, 2 the staged
 * kernel for every method (the parity tests run both on the same inputs), negative only queries; returns the previous setting. */
int epi_triangulate_staged(int on);

/* triangulation.py:184-220 (polynomial_triangulation), V must be 2: F = [t]_x R from the two projection
 * matrices (:196-204), the matches moved onto the closest exactly-epipolar pair (cv2.correctMatches, :210;
 * Hartley & Sturm / HZ Alg. 12.1, the degree-6 polynomial solved by real-root isolation in float64), then
 * the dlt solve above on the corrected matches (:220).  The reference's fallback to an 8-point F when the
 * correction returns NaN (:213-217) lives one level up: this entry point gives such points status 0 and the host-side
 * mirror (epipolarpose_amd/utils/triangulation.py:polynomial_triangulation) re-runs them with epi_fundamental_8point +
 * epi_correct_matches as the reference does. */
int epi_triangulate_poly(const void* kps, int kps_stride, const void* P, int dtype, int G, int V, int J,
                         void* X, int32_t* status, epi_stream_t stream);

/* cv2.findFundamentalMat(points1, points2, FM_8POINT) (triangulation.py:216; cameras.py:136-143 uses FM_LMEDS, whose random
 * sampling is not reproducible), batched: u1/u2 [G][J][2] f64 -> F [G][3][3] f64 with x2^T F x1 = 0, rank 2, F[2][2] = 1;
 * status [G] int32 or NULL: 1, or 0 for degenerate input (J < 8, coincident points: F = 0). */
int epi_fundamental_8point(const double* u1, const double* u2, int G, int J, double* F, int32_t* status, epi_stream_t stream);

/* The part of cv2.findFundamentalMat(points1, points2, FM_LMEDS) (cameras.py:136-143) that scales with the number of correspondences:
 * F [H][3][3] f64 candidate matrices (the 7-point solutions of the random samples: host side, utils/triangulation.py
 * find_fundamental_mat_lmeds), u1/u2 [N][2] f64 -> medians [H] f64 = median over the N pairs of the float32 symmetric epipolar error
 * max(d(x1, F^T x2)^2, d(x2, F x1)^2) (mean of the two middle elements for even N); epi_fundamental_errors: the errors [N] f32 of one
 * matrix (the inlier test err <= sigma^2).  OpenCV 4.1.0 ptsetreg.cpp / fundam.cpp restated: parity with OpenCV itself is unpinned. */
int epi_fundamental_lmeds_medians(const double* F, int H, const double* u1, const double* u2, int N, double* medians, epi_stream_t stream);
int epi_fundamental_errors(const double* F, const double* u1, const double* u2, int N, float* err, epi_stream_t stream);

/* cv2.correctMatches(F, points1, points2) (called at triangulation.py:210,216), batched: F [G][3][3],
 * u1/u2/out1/out2 [G][J][2], all f64 device pointers.  out may alias in. */
int epi_correct_matches(const double* F, const double* u1, const double* u2, int G, int J,
                        double* out1, double* out2, epi_stream_t stream);

/* img_utils.py:212-243 + prep_h36m.py:177-204 (get_batch_labels_from_global_coords):
 * world joints X [G][J][3] f64 -> per-view pseudo labels label [B][3J] f32, weight [B][3J] f32 (ones). */
int epi_reproject_labels(const double* X, int G, int V, int J, const epi_view_meta* meta_host,
                         double patch_w, double patch_h, double rect3d, int root_joint,
                         float* label, float* weight, epi_stream_t stream);

/* img_utils.py:166-190 (self_supervision) fused into ONE launch: decode -> triangulate -> re-project.
 * method: 0 iterative LS (reference), 1 linear LS, 2 DLT, 3 polynomial (V = 2 only).  X_out [G][J][3] f64 may be NULL. */
int epi_self_supervision(const float* xyz, int G, int V, int J, const epi_view_meta* meta_host,
                         double patch_w, double patch_h, double rect3d, int root_joint,
                         int method, double tolerance, int max_iter,
                         float* label, float* weight, double* X_out, epi_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Deconvolution head on the matrix cores -- replaces the cuDNN calls behind
 * lib/models/pose3d_resnet.py:158-183 (ConvTranspose2d k=4 s=2 p=1, no bias) and :116-122 (final 1x1 conv).
 * Activations are NHWC bf16; accumulation is fp32 (v_mfma_f32_32x32x16_bf16).
 * ------------------------------------------------------------------------------------------------ */

/* C[M][N] (bf16 or f32, row stride ldc) = A[M][K] (bf16, row stride lda) * Bt[N][K]^T (bf16, row stride ldb)
 * (+ bias[N] f32 or NULL).  K % 8 == 0, lda/ldb % 8 == 0, ldc % 4 == 0, 16-byte aligned bases.
 * The final 1x1 convolution is this with A = activations [B*H*W][Cin], Bt = weight [Cout][Cin]; its
 * backward-data is this with A = dlogits [B*H*W][Cout], Bt = weight^T [Cin][Cout].
 * workspace: epi_gemm_workspace_bytes(M, N, K, nphase) bytes (split-K slabs; 0 when the problem already fills
 * the chip; nphase = 1 except 4 for epi_deconv4x4s2_fwd). */
size_t epi_gemm_workspace_bytes(int M, int N, int K, int nphase);
int epi_gemm_bf16(const void* A, int lda, const void* Bt, int ldb, void* C, int ldc, int c_dtype,
                  int M, int N, int K, const float* bias, void* workspace, size_t workspace_bytes, epi_stream_t stream);

/* Re-pack a ConvTranspose2d weight [Cin][Cout][4][4] (bf16) into the two GEMM operand forms:
 *   w_phase [4][Cout][4*Cin] : per output-parity phase (2*(oh&1)+(ow&1)) the 2x2 taps that reach it
 *   w_bwd   [Cin][16*Cout]   : all 16 taps, for backward-data.                   Either may be NULL. */
int epi_deconv4x4s2_pack_weight(const void* w_bf16, int Cin, int Cout, void* w_phase, void* w_bwd,
                                epi_stream_t stream);
/* The same for a weight kept in channels_last memory [Cin][kh][kw][Cout] (bf16): that memory IS w_bwd already, only w_phase is
 * produced (32x32 tile transposes per tap).  _fill_row writes a row of the multi-layer table of epi_conv2d_pack_weight_bwd_multi,
 * so that the deconvolution weights are re-packed by the same single launch after every optimizer step. */
int epi_deconv4x4s2_pack_phase_cl(const void* w_cl, int Cin, int Cout, void* w_phase, epi_stream_t stream);
int epi_deconv4x4s2_pack_fill_row(void* row_host, const void* w_cl, void* w_phase, int Cin, int Cout, long long tile_begin,
                                  long long* ntiles);

/* y [B][2H][2W][Cout] = ConvTranspose2d(x [B][H][W][Cin]) as 4 implicit (gather) GEMMs with K = 4*Cin.
 * Cin % 64 == 0, Cout % 4 == 0.  Raw output (BatchNorm + ReLU follow, pose3d_resnet.py:180-181).
 * workspace: epi_gemm_workspace_bytes(B*H*W, Cout, 4*Cin, 4). */
int epi_deconv4x4s2_fwd(const void* x, const void* w_phase, void* y, int B, int H, int W, int Cin, int Cout,
                        void* workspace, size_t workspace_bytes, epi_stream_t stream);
/* The same with the BatchNorm batch sums of the result (cf. epi_conv2d_fwd): bn_sums [epi_bn_sum_copies(Cout)][2 Cout] f32 (zeroed by the
 * caller) += per-channel (sum, sum of squares) of the bf16 outputs when the launch can do it from its epilogue (*stats_done = 1: the
 * BatchNorm that follows needs no statistics pass), else left alone (*stats_done = 0).  Either pointer may be NULL. */
int epi_deconv4x4s2_fwd_stats(const void* x, const void* w_phase, void* y, int B, int H, int W, int Cin, int Cout, float* bn_sums,
                              int* stats_done, void* workspace, size_t workspace_bytes, epi_stream_t stream);

/* dx [B][H][W][Cin] from dy [B][2H][2W][Cout]: a 4x4 stride-2 implicit GEMM with K = 16*Cout.  Cout % 64 == 0.
 * workspace: epi_gemm_workspace_bytes(B*H*W, Cin, 16*Cout, 1). */
int epi_deconv4x4s2_bwd_data(const void* dy, const void* w_bwd, void* dx, int B, int H, int W, int Cin, int Cout,
                             void* workspace, size_t workspace_bytes, epi_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * BatchNorm2d (+ residual add) (+ ReLU), NHWC bf16 -- replaces every nn.BatchNorm2d(momentum=0.1) + nn.ReLU
 * (+ `out += residual`) of lib/models/pose3d_resnet.py:31-47,68-88,158-183,186-188.
 *   x, residual, y : [R][C] bf16 (R = B*H*W, C % 8 == 0);  gamma, beta, running_mean, running_var, mean, rstd : [C] f32
 *   scale_shift    : [2C] f32 out (scale = gamma*rstd, shift = beta - mean*scale), reused by the backward
 *   sums_ws        : [epi_bn_sum_copies(C)][2C] f32 accumulator (training) that must be ZERO on entry; on return its copies add
 *                    up to the batch sums (the producers -- row blocks of the statistics kernel, M tiles of a convolution
 *                    epilogue -- spread their fp32 atomics over the copies; what serialises in L2 is atomics per 128-byte line).
 *                    Keep one per layer and hand it to the layer's backward call as `fwd_sums_clear`, which zeroes
 *                    it again (steady-state training and graph replays then need no memset); a caller that runs a
 *                    training-mode forward without the backward clears it itself before the next forward.
 *   training != 0  : batch statistics, running stats updated with `momentum` (unbiased variance),
 *                    num_batches_tracked += 1;   training == 0: running statistics;   training == 2: like 1, but sums_ws
 *                    ALREADY holds the batch sums (accumulated by the producing convolution, epi_conv2d_fwd) -- no statistics pass.
 *   bwd_sums       : [2C] f32 or NULL: zeroed by the forward so that it can serve as `dbeta_dgamma` of this
 *                    layer's next backward pass without a separate memset
 * Backward (training):  dbeta_dgamma [2C] f32, ZERO on entry, out = (sum dz, sum dz*xhat);  dx [R][C];  dres [R][C] or NULL
 *   = gradient of the residual input;  y = saved forward output, required when relu && dres;
 *   fwd_sums_clear [epi_bn_sum_copies(C)][2C] f32 or NULL: zeroed (the forward accumulator above).
 *   param_grads [2C] f32 or NULL: a COPY of (dbeta | dgamma) in caller-owned memory outside the accumulator protocol -- hand
 *   THIS to the optimizer / autograd, never dbeta_dgamma itself: the layer's next forward clears that accumulator, which would
 *   wipe a gradient that is still waiting (gradient accumulation, a forward between backward and optimizer.step).
 * ------------------------------------------------------------------------------------------------ */
int epi_bn_sum_copies(int C);       /* accumulator copies of a C-channel layer (4 for C <= 512, else 1) */
int epi_bn_act_fwd(const void* x, const void* residual, long long R, int C, const float* gamma, const float* beta,
                   float eps, float momentum, int training, int relu, float* running_mean, float* running_var,
                   long long* num_batches_tracked, float* mean, float* rstd, float* scale_shift, float* sums_ws,
                   float* bwd_sums, void* y, epi_stream_t stream);
int epi_bn_act_bwd(const void* dy, const void* x, const void* y, long long R, int C, const float* gamma,
                   const float* mean, const float* rstd, const float* scale_shift, int relu,
                   float* dbeta_dgamma, void* dx, void* dres, float* fwd_sums_clear, float* param_grads, epi_stream_t stream);

/* The last BatchNorm of a residual unit whose shortcut is a projection, and that projection's own BatchNorm, in ONE pass
 * (`residual = self.downsample(x)`; `out = self.bn3(out); out += residual; out = self.relu(out)`, pose3d_resnet.py:68-88 with :130-136):
 *   y = relu(bn_main(x) + bn_proj(x_proj)),  x / x_proj / y [R][C] bf16 -- the projection's normalised output is never written.
 * Each EpiBnLayer carries the arguments epi_bn_act_fwd takes per layer (same accumulator protocol, same saved statistics, running
 * estimates updated); training: 1 = compute both layers' batch sums here, 2 = both producers delivered them, 0 = running statistics.
 * The backward pass is unchanged: epi_bn_act_bwd of the main layer yields dres, which is the projection BatchNorm's dy. */
typedef struct EpiBnLayer {
    const float* gamma;
    const float* beta;
    float* running_mean;
    float* running_var;
    long long* num_batches_tracked;
    float* mean;          /* out [C] */
    float* rstd;          /* out [C] */
    float* scale_shift;   /* out [2C] */
    float* sums_ws;       /* [epi_bn_sum_copies(C)][2C] */
    float* bwd_sums;      /* [2C] */
} EpiBnLayer;
int epi_bn_act_fwd_dual(const void* x, const void* x_proj, long long R, int C, const EpiBnLayer* main_bn, const EpiBnLayer* proj_bn,
                        float eps, float momentum, int training, void* y, epi_stream_t stream);
/* The stem's `self.maxpool(self.relu(self.bn1(...)))` (pose3d_resnet.py:186-188) without the normalised tensor in between: epi_bn_finalize does the
 * per-channel part of epi_bn_act_fwd alone (statistics from sums_ws -- training 2: the producer delivered them; 0: running statistics --, scale / shift,
 * running estimates, accumulator hand-over), epi_maxpool3x3s2_bn_relu_fwd pools bf16(relu(x * scale + shift)) straight from the raw convolution output
 * (same result and window positions as epi_bn_act_fwd + epi_maxpool3x3s2_fwd, bit for bit).  Backward: epi_maxpool3x3s2_bwd, then epi_bn_act_bwd
 * with the mask from x (y = NULL). */
int epi_bn_finalize(const EpiBnLayer* layer, long long R, int C, float eps, float momentum, int training, epi_stream_t stream);
int epi_maxpool3x3s2_bn_relu_fwd(const void* x, const float* scale_shift, void* y, void* pos, int B, int H, int W, int C, epi_stream_t stream);

/* ---- The BatchNorm-backward reduction fused into the backward-data GEMM that produces the layer's dy ---------------------------
 * (autograd of `out = relu(bn(conv(x)) [+ residual])`, pose3d_resnet.py:33-46,68-88,186-199: the gradient of a BatchNorm output is
 * produced by the backward-data pass of the convolution that consumed it.)  epi_bn_act_bwd starts with a pass over (dy, x, y) for
 * sum(dz) and sum(dz * xhat), dz = dy * [output > 0].  The *_bnred variants of the backward-data entries do that in their epilogue:
 *   red->z    raw forward output the BatchNorm normalised, bf16, same rows / stride as the dx being produced
 *   red->y    saved forward output (mask = y > 0) for residual + ReLU layers, or NULL: mask = scale*z + shift > 0 (ignored without relu)
 *   red->bn   [mean | rstd | scale | shift], C floats each (the `mean` pointer epi_bn_act_fwd filled, when the four are contiguous)
 *   red->sums [2C] f32 accumulator, ZERO before the first contribution: += (sum dz | sum dz * xhat)
 * *red_done = 1: dx holds dz (the masked gradient) and the sums are in -- finish with epi_bn_act_bwd_reduced; dz is also the gradient
 *   of the residual input, so no second output is written.  *red_done = 0 (split-K launches, rows not 16-byte aligned, EPI_BN_BWD_FUSE=0):
 *   dx holds the plain gradient, red->sums is untouched -- call epi_bn_act_bwd as before.
 * epi_bn_act_bwd_reduced: dx = gamma*rstd * (dz - dbeta/R - xhat * dgamma/R) from dz and the finished sums (no mask, no reduction). */
typedef struct EpiBnReduce {
    const void* z;
    const void* y;
    const float* bn;
    float* sums;
    int relu;
} EpiBnReduce;
int epi_bn_act_bwd_reduced(const void* dz, const void* x, long long R, int C, const float* gamma, const float* mean, const float* rstd,
                           const float* scale_shift, const float* dbeta_dgamma, void* dx, float* fwd_sums_clear, float* param_grads,
                           epi_stream_t stream);
/* addend_step (epi_conv2d_bwd_data_bnred; 1 = as epi_conv2d_bwd_data): 2 = `addend` is [B][H/2][W/2][Cin] and holds the contribution of the EVEN
 * pixels only -- the shortcut gradient through a 1x1 stride-2 projection (`downsample`, pose3d_resnet.py:130-136), whose backward-data is non-zero at
 * every other pixel: the projection's gradient stays at half resolution (a plain 1x1 backward-data on the [B][H/2][W/2] grid) and is never expanded with
 * zeros.  1x1 / stride-1 problems on an unsplit launch only: ask epi_conv2d_bwd_data_half_addend_ok first.  red may be NULL (then red_done may be too). */
int epi_conv2d_bwd_data_bnred(const void* dy, const void* w_bwd, void* dx, int B, int H, int W, int Cin, int Cout, int KH, int KW,
                              int stride, int pad, const void* addend, int addend_step, const EpiBnReduce* red, int* red_done, void* workspace,
                              size_t workspace_bytes, epi_stream_t stream);
int epi_conv2d_bwd_data_half_addend_ok(int B, int H, int W, int Cin, int Cout);
int epi_deconv4x4s2_bwd_data_bnred(const void* dy, const void* w_bwd, void* dx, int B, int H, int W, int Cin, int Cout,
                                   const EpiBnReduce* red, int* red_done, void* workspace, size_t workspace_bytes, epi_stream_t stream);
int epi_gemm_bf16_bnred(const void* A, int lda, const void* Bt, int ldb, void* C, int ldc, int M, int N, int K, const EpiBnReduce* red,
                        int* red_done, void* workspace, size_t workspace_bytes, epi_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Backbone convolutions on the matrix cores -- replace the cuDNN/MIOpen calls behind every bias-free nn.Conv2d of
 * lib/models/pose3d_resnet.py:21-88,130-136 (BasicBlock / Bottleneck conv1..conv3, downsample.0; groups 1, dilation 1).
 * NHWC bf16 activations, fp32 accumulation; the same implicit-GEMM kernel as the deconvolution head.
 *   x [B][H][W][Cin],  y / dy [B][Ho][Wo][Cout],  Ho = (H + 2*pad - KH)/stride + 1
 *   w      [Cout][KH][KW][Cin]  bf16 -- the memory order of a channels_last weight tensor (no repacking for the forward)
 *   w_bwd  Cin*KH*KW*Cout bf16, written by epi_conv2d_pack_weight_bwd: [Cin][KH][KW][Cout] for stride 1; for stride 2
 *          the four output-parity phase blocks [Cin][taps of the phase][Cout] one after the other
 * Coverage: KH*KW <= 16, Cout % 4 == 0, Cin % 4 == 0, and Cin % 64 == 0 (forward) / Cout % 64 == 0 (backward-data) unless
 * the convolution is 1x1 / stride 1 / pad 0 (a plain GEMM: % 8); stride 1 or 2; stride-2 backward-data needs even H, W and
 * at most 4 taps per parity phase (3x3 pad 1 and 1x1 pad 0 qualify).  Anything else: EPI_ERR_UNSUPPORTED (the caller keeps
 * the library convolution for it, e.g. the 7x7 stem on 3 input channels).
 * workspace: epi_conv2d_workspace_bytes(...) covers forward and backward-data of one layer.
 * ------------------------------------------------------------------------------------------------ */
size_t epi_conv2d_workspace_bytes(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
/* 3x3 / stride 1 / pad 1 forward and backward-data have a second kernel that stages each pixel patch once for all nine taps
 * (csrc/head_gemm.hip, conv_patch_kernel).  mode 0: never; 1: only where its tiles fill the chip without a split over channel
 * chunks; 2 (default; EPI_CONV3X3_PATCH overrides): always.  Sets the mode (other values only query) and returns the previous one.  Same results
 * either way (fp32 accumulation over the same products; the summation order differs). */
int epi_conv3x3_patch_mode(int mode);
/* Tuning hooks of the NT implicit-GEMM launches (tools/gemm_lab.hip, tools/gemm_trace.hip; results never depend on them beyond fp32 summation
 * order).  epi_gemm_tune: tile 0 by shape | 1 128x128 | 2 256x256 | 3 64x128 | 4 64x64 (negative: keep), pipe -1 per EPI_GEMM_PIPE | 0 two LDS
 * stages | 1 ring when the K loop has >= 6 tiles | 2 ring always (< -1: keep); returns the tile override in force.  epi_gemm_store_policy: result
 * stores of the coalesced epilogue 0 plain | 1 non-temporal (negative: query; EPI_GEMM_STORES=nt sets the default); returns the previous policy. */
int epi_gemm_tune(int tile, int pipe);
int epi_gemm_store_policy(int policy);
/* bn_sums [epi_bn_sum_copies(Cout)][2*Cout] f32 or NULL: the accumulator of the BatchNorm that follows (sums_ws of epi_bn_act_fwd, ZERO on entry).  When
 * the launch can do it (unsplit result), the GEMM epilogue adds the per-channel (sum, sum of squares) of the bf16 outputs and sets
 * *bn_sums_done = 1 -- pass training = 2 to epi_bn_act_fwd then (its statistics pass is skipped); otherwise *bn_sums_done = 0 and
 * bn_sums is untouched. */
int epi_conv2d_fwd(const void* x, const void* w, void* y, int B, int H, int W, int Cin, int Cout, int KH, int KW,
                   int stride, int pad, float* bn_sums, int* bn_sums_done, void* workspace, size_t workspace_bytes,
                   epi_stream_t stream);
/* The 1x1 / stride-1 convolution behind a BatchNorm + ReLU WITHOUT the normalised tensor in between (`out = self.conv3(self.relu(self.bn2(out)))`,
 * pose3d_resnet.py:76-82): z [B][H][W][Cin] is the raw output of the convolution in front, in_bn that layer's BatchNorm with the batch sums of z already in
 * in_bn->sums_ws (in_training = 2; 0: running statistics).  The launch derives scale / shift itself, applies relu(z * scale + shift) to its A operand
 * on the fly, fills in_bn->mean / rstd / scale_shift, updates the running estimates and clears in_bn->bwd_sums as epi_bn_act_fwd(training = 2) would;
 * bn_sums / bn_sums_done as epi_conv2d_fwd.  Cin % 64 == 0, Cin <= 2048, Cout % 8 == 0; EPI_ERR_UNSUPPORTED otherwise and in deterministic mode (the
 * caller then normalises first).  The weight gradient of such a layer: EpiWgradItem::x_scale_shift. */
int epi_conv1x1_fwd_bn_in(const void* z, const EpiBnLayer* in_bn, int in_training, float eps, float momentum, const void* w, void* y, int B, int H, int W,
                          int Cin, int Cout, float* bn_sums, int* bn_sums_done, void* workspace, size_t workspace_bytes, epi_stream_t stream);
int epi_conv2d_pack_weight_bwd(const void* w, int Cout, int Cin, int KH, int KW, int stride, int pad, void* w_bwd,
                               epi_stream_t stream);
/* The same for MANY layers in one launch (every backbone weight after an optimizer step).  The caller keeps a table of
 * epi_conv2d_pack_row_bytes()-sized rows: fill each row ONCE on the host with epi_conv2d_pack_fill_row (tile_begin = the sum of
 * the *ntiles of the rows before it), copy the table to the device, then launch with the total tile count. */
size_t epi_conv2d_pack_row_bytes(void);
int epi_conv2d_pack_fill_row(void* row_host, const void* w, void* w_bwd, int Cout, int Cin, int KH, int KW, int stride, int pad,
                             long long tile_begin, long long* ntiles);
int epi_conv2d_pack_weight_bwd_multi(const void* rows, int nrows, long long total_tiles, epi_stream_t stream);
/* addend [B][H][W][Cin] bf16 or NULL: dx = (backward-data result, rounded to bf16) + addend -- the gradient that reaches the same
 * tensor through the other branch of a residual junction (`out += residual`, pose3d_resnet.py:44,85), added in the GEMM epilogue
 * instead of a separate element-wise launch. */
int epi_conv2d_bwd_data(const void* dy, const void* w_bwd, void* dx, int B, int H, int W, int Cin, int Cout, int KH, int KW,
                        int stride, int pad, const void* addend, void* workspace, size_t workspace_bytes, epi_stream_t stream);

/* Weight gradient of a Conv2d (groups 1, dilation 1) on NHWC bf16 tensors -- the backbone convolutions' backward-weight
 * (autograd of nn.Conv2d in lib/models/pose3d_resnet.py:21-88), which the reference leaves to cuDNN:
 *   x [B][H][W][Cin], dy [B][Ho][Wo][Cout] -> dw [Cout][KH][KW][Cin] (channels_last weight order), dw_dtype EPI_F32 | EPI_BF16;
 *   KH*KW <= 16, Cin % 8 == 0, Cout % 8 == 0.  workspace: epi_gemm_tn_workspace_bytes(B*Ho*Wo, Cout, Cin, KH*KW). */
int epi_conv2d_bwd_weight(const void* x, const void* dy, void* dw, int dw_dtype, int B, int H, int W, int Cin, int Cout,
                          int KH, int KW, int stride, int pad, void* workspace, size_t workspace_bytes, epi_stream_t stream);
/* Deferred reduction.  A weight gradient whose reduction over batch*pixels was split leaves one fp32 slab per split in the
 * workspace; summing them is a separate small launch per layer (55 per ResNet-50 backward).  The _deferred entry point runs the
 * GEMM only: the slabs stay in `workspace` -- memory the caller must then leave untouched until the reduce has run -- and
 * *pending describes the outstanding sum (pending->nsplit == 0: nothing outstanding, dw is complete).  epi_slab_reduce_multi
 * performs MANY of them in one launch: rows_dev = a device array of the descriptors with chunk_begin filled in (row r starts at
 * chunk  sum over the rows before it of epi_slab_reduce_chunks(n)), total_chunks = the sum over all rows.  pending == NULL
 * behaves like epi_conv2d_bwd_weight. */
typedef struct EpiSlabReduce {
    const float* slabs;     /* [nsplit][n] fp32 partial results */
    void* out;              /* n elements, f32 or bf16 */
    long long n;
    long long chunk_begin;  /* filled by the caller: first 512-element chunk of this row in the multi launch */
    int nsplit, out_bf16;
} EpiSlabReduce;
int epi_conv2d_bwd_weight_deferred(const void* x, const void* dy, void* dw, int dw_dtype, int B, int H, int W, int Cin, int Cout,
                                   int KH, int KW, int stride, int pad, void* workspace, size_t workspace_bytes,
                                   EpiSlabReduce* pending, epi_stream_t stream);
long long epi_slab_reduce_chunks(long long n);
int epi_slab_reduce_multi(const EpiSlabReduce* rows_dev, int nrows, long long total_chunks, epi_stream_t stream);

/* Grouped weight gradients (round 3): MANY backward-weight GEMMs -- the convolutions of several residual units, a whole ResNet stage --
 * in at most two launches (128 x 128 output tiles; 64 x 128 tiles for the layers with <= 64 output channels).  The autograd of
 * nn.Conv2d / nn.ConvTranspose2d in lib/models/pose3d_resnet.py:50-88,158-183 produces these gradients one layer at a time; at batch
 * 32 a single deep layer has 16 .. 144 output tiles and had to cut its reduction over batch*pixels into 2 .. 16 slices to give every
 * compute unit a workgroup, each slice writing an fp32 slab of the whole result (866 MB of slabs per ResNet-50 step).  Together the
 * tiles of ~10 .. 20 layers fill the chip with unsplit (or barely split) reductions.
 *   kind EPI_WGRAD_CONV2D:       x [B][H][W][Cin], dy [B][Ho][Wo][Cout] -> dw [Cout][KH][KW][Cin]    (= epi_conv2d_bwd_weight)
 *   kind EPI_WGRAD_DECONV4X4S2:  x [B][H][W][Cin], dy [B][2H][2W][Cout] -> dw [Cin][16 taps][Cout]  (= epi_deconv4x4s2_bwd_weight; KH, KW,
 *                                stride, pad ignored)
 * epi_wgrad_group_plan (host only): the slab bytes the group needs (0: no reduction is split) and, per item, its reduction slices.
 * epi_wgrad_group: launches; pending[r] describes item r's outstanding slab sum (nsplit == 0: dw complete with the launch), to be run
 * with epi_slab_reduce_multi; slab_ws must stay untouched until then.  n <= epi_wgrad_group_max(). */
enum { EPI_WGRAD_CONV2D = 0, EPI_WGRAD_DECONV4X4S2 = 1 };
typedef struct EpiWgradItem {
    const void* x;
    const void* dy;
    void* dw;
    int dw_dtype, kind;
    int B, H, W, Cin, Cout, KH, KW, stride, pad;
    const float* x_scale_shift;   /* NULL, or (EPI_WGRAD_CONV2D, 1x1 / stride 1 only) [2 Cin] f32: `x` is the RAW output z of the convolution in front and the
                                     layer's input was relu(z * scale[c] + shift[c]) -- the tensor epi_conv1x1_fwd_bn_in never wrote; the kernel re-applies it */
} EpiWgradItem;
int epi_wgrad_group_max(void);
int epi_wgrad_group_plan(const EpiWgradItem* items, int n, size_t* slab_bytes, int* nsplit);
int epi_wgrad_group(const EpiWgradItem* items, int n, void* slab_ws, size_t slab_bytes, EpiSlabReduce* pending, epi_stream_t stream);
/* One item by itself, with its own reduction split (= epi_conv2d_bwd_weight_deferred / epi_deconv4x4s2_bwd_weight for what those can describe, plus
 * x_scale_shift).  workspace: epi_gemm_tn_workspace_bytes(rows, Cout, Cin, taps) of the item; pending NULL: the slab sum runs behind the GEMM. */
int epi_wgrad_item(const EpiWgradItem* item, void* workspace, size_t workspace_bytes, EpiSlabReduce* pending, epi_stream_t stream);

/* Weight gradients of the head (reduction over batch*pixels, fp32 results).  workspace: the split-K slabs,
 * epi_gemm_tn_workspace_bytes(R, I, J, ntap) bytes (J = columns per filter tap; ntap = 1 for epi_gemm_tn_bf16, 16 for the
 * deconvolution; may be 0: a reduction that needs no split writes its result directly).
 *   epi_gemm_tn_bf16:            C[I][J] = A[R][I]^T * B[R][J]  (final conv: A = dlogits, B = activations -> dW[Cout][Cin])
 *   epi_deconv4x4s2_bwd_weight:  dw_taps[Cin][16][Cout] (EPI_F32 | EPI_BF16), tap = kh*4+kw -- the memory order of a channels_last
 *                                [Cin, Cout, 4, 4] weight -- from x [B][H][W][Cin], dy [B][2H][2W][Cout]
 *   epi_column_sums_bf16:        sums[2C] += per-column (sum, sum of squares) of x [R][C]  (bias gradient; zero it first) */
size_t epi_gemm_tn_workspace_bytes(int R, int I, int J, int ntap);
/* The plan behind that figure (host only, nothing is launched): plan[0] tile configuration (0: 128 x 128, 1: 64 x 128, 2: 256 x 256),
 * plan[1] output tiles, plan[2] reduction splits (each writes one fp32 slab of the whole result), plan[3] rows per split. */
int epi_gemm_tn_plan(int R, int I, int J, int ntap, long long* plan);
int epi_gemm_tn_bf16(const void* A, int lda, const void* B, int ldb, float* C, int R, int I, int J,
                     void* workspace, size_t workspace_bytes, epi_stream_t stream);
int epi_deconv4x4s2_bwd_weight(const void* x, const void* dy, void* dw_taps, int dw_dtype, int B, int H, int W, int Cin, int Cout,
                               void* workspace, size_t workspace_bytes, epi_stream_t stream);
int epi_column_sums_bf16(const void* x, long long R, int C, float* sums, epi_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fused multi-tensor Adam -- replaces torch.optim.Adam as built by lib/utils/utils.py:55-59 (lr only, betas
 * (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad), one launch for all parameters.
 *   table  : device array of rows {float* p; const void* g; float* m; float* v; uint16* shadow; int64 n; int32
 *            g_bf16; int32 pad} (epi_adam_tensor_bytes() bytes each).  g is bf16 when g_bf16 != 0, else f32;
 *            shadow (or NULL) receives the bf16 copy of the updated parameter.
 *   chunks : device array of int32 pairs (tensor index, chunk index), chunk = epi_adam_chunk_elems() elements.
 *   step   : 1-based step count (bias correction is computed on the host from it).
 * ------------------------------------------------------------------------------------------------ */
size_t epi_adam_tensor_bytes(void);
int epi_adam_chunk_elems(void);
int epi_adam_step(const void* table, const void* chunks, int nchunks, float lr, float beta1, float beta2, float eps,
                  long long step, epi_stream_t stream);
/* The same with torch.nn.utils.clip_grad_norm_(parameters, max_norm) folded in (refiner/main.py:53-54): norm_sq is a device scalar,
 * ZERO on entry, that receives the squared total gradient norm; every gradient is scaled by min(1, max_norm / (norm + 1e-6)) as the
 * Adam launch reads it (the stored gradients are left untouched). */
int epi_adam_step_clipped(const void* table, const void* chunks, int nchunks, float lr, float beta1, float beta2, float eps,
                          long long step, float max_norm, float* norm_sq, epi_stream_t stream);

/* Inverted dropout, bf16 (refiner/model.py:26: nn.Dropout(p)): y_i = keep_i ? x_i / (1 - p) : 0 with keep_i a stateless hash of
 * (seed, i); calling it on the output gradient with the same seed IS the backward pass.  y may alias x. */
int epi_dropout_bf16(const void* x, void* y, long long n, float p, unsigned long long seed, epi_stream_t stream);

/* nn.MaxPool2d(kernel_size=3, stride=2, padding=1) of the ResNet stem (lib/models/pose3d_resnet.py:104,186), NHWC bf16, C % 8 == 0.
 *   x [B][H][W][C] -> y [B][Ho][Wo][C], Ho = (H - 1) / 2 + 1;  pos [B][Ho][Wo][C] uint8: the window position (kh*3 + kw) each output
 *   element was taken from -- the library's selection rule (first of equal maxima in scan order, a NaN sticks) -- which is all the
 *   backward pass needs: dx [B][H][W][C] gathers dy from the <= 4 windows that selected the pixel (no atomics, no zero fill). */
int epi_maxpool3x3s2_fwd(const void* x, void* y, void* pos, int B, int H, int W, int C, epi_stream_t stream);
int epi_maxpool3x3s2_bwd(const void* dy, const void* pos, void* dx, int B, int H, int W, int C, epi_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Input pipeline -- replaces, per batch instead of per sample on the host, generate_patch_image_cv
 * (lib/utils/img_utils.py:114-127: cv2.warpAffine, INTER_LINEAR, constant border) and the colour / normalisation stage of
 * get_single_patch_sample (:265-279: BGR -> RGB, per-channel colour scale, clip to [0, 255], (x - mean) / std).
 *   frames        device bytes holding B decoded BGR frames; sample b starts at frames + frame_offset[b], size frame_hw[b] = (h, w)
 *   trans         [B][2][3] f64: the FORWARD affine frame -> patch of gen_trans_from_patch_cv (:72-105); inverted in the kernel
 *                 exactly as cv::warpAffine does, sampled in OpenCV's 5-bit fixed point with its 2^15-scaled integer weights
 *   do_flip       [B] int32 or NULL: mirror the frame horizontally first (:120-122)
 *   color_scale   [B][3] f32 per RGB channel or NULL;   mean_host / std_host: 3 floats each (HOST pointers) or both NULL (no normalisation)
 *   out           [B][3][ph][pw] (EPI_NCHW) or [B][ph][pw][3] (EPI_NHWC), f32 or bf16, channels in RGB order
 * Not covered: the synthetic-occlusion augmentation (lib/utils/augmentation.py:61-114 pastes Pascal-VOC objects; needs that dataset).
 * ------------------------------------------------------------------------------------------------ */
int epi_crop_patches(const void* frames, const long long* frame_offset, const int* frame_hw, const double* trans,
                     const int* do_flip, const float* color_scale, const float* mean_host, const float* std_host, int B,
                     int patch_h, int patch_w, void* out, int out_dtype, int out_layout, epi_stream_t stream);
/* The same with the synthetic-occlusion augmentation (lib/utils/augmentation.py:61-114 occlude_with_objects / paste_over, applied to the
 * uint8 RGB patch between the warp and the colour stage, img_utils.py:271-272):
 *   occ_bank    device bytes of N RGBA occluder images (alpha 255 inside, 192 on the border ring, 0 outside, as load_occluders builds them),
 *               occluder n at occ_bank + occ_offset[n], native size occ_hw[n] = (h, w)
 *   occ_place   [B][max_occ][5] int32 per sample, in pasting order: occluder index (-1 ends the list), pasted width and height (<= native:
 *               cv2.resize INTER_AREA restated as an exact box-filter average), top-left corner (x0, y0) in the patch (may be outside)
 * The random draws (count, choice, scale, centre) stay with the caller (dataset/synthetic_frames.py, in the reference's order).  The
 * reference's occluders come from Pascal VOC (not on the boxes): the bank holds whatever RGBA images the caller supplies. */
int epi_crop_patches_occluded(const void* frames, const long long* frame_offset, const int* frame_hw, const double* trans,
                              const int* do_flip, const float* color_scale, const float* mean_host, const float* std_host, int B,
                              int patch_h, int patch_w, const void* occ_bank, const long long* occ_offset, const int* occ_hw,
                              const int* occ_place, int max_occ, void* out, int out_dtype, int out_layout, epi_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Pose evaluation -- replaces the per-sample loop of H36M_Integral.evaluate
 * (lib/dataset/h36m.py:168-378) incl. compute_similarity_transform (lib/utils/prep_h36m.py:108-168).
 *   pred_img, gt_img [N][J][3] f64: (u, v, root-relative depth mm) in image coordinates; pelvis_z [N]: camera-space
 *   root depth; fl, c_p [N][2]; root: root joint index; j14 [n14] int32 (device): joints of the 14-joint protocol.
 *   metrics [N][9] f64: MPJPE, PA-MPJPE, N-MPJPE, the same three over j14, mean |dx|, |dy|, |dz|  (all root-centred);
 *   per_joint [N][J] f64: per-joint errors.  J <= 32.
 * ------------------------------------------------------------------------------------------------ */
int epi_evaluate_poses(const double* pred_img, const double* gt_img, const double* pelvis_z, const double* fl,
                       const double* c_p, int N, int J, int root, const int32_t* j14, int n14, double* metrics,
                       double* per_joint, epi_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Stem convolution (round 3): nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False) of pose3d_resnet.py:99,186 on the same
 * implicit-GEMM kernels as every other convolution (rounds 1-2 left it to MIOpen: 3 input channels cannot feed a 16-byte DMA chunk).
 * The image is rewritten space-to-depth -- s2d [B][H/2 + 3][W/2 + 3][16] bf16, 2 x 2 pixel blocks of the 3-zero-padded image as 12 (+4 zero)
 * channels -- which turns the layer into a 4 x 4 stride-1 convolution whose four horizontal taps are 64 contiguous elements: a gather GEMM
 * with K = 256 (csrc/head_gemm.hip, "stem").  H, W even.
 *   epi_stem7x7s2_s2d            x [B,3,H,W] (EPI_NCHW) or [B,H,W,3] (EPI_NHWC), EPI_F32 | EPI_BF16 -> s2d (epi_stem7x7s2_s2d_bytes bytes)
 *   epi_stem7x7s2_pack_weight    w bf16 [Cout,3,7,7] (contiguous, or channels_last memory [Cout][7][7][3]) -> wp [Cout][256] bf16
 *   epi_stem7x7s2_fwd            -> y [B][H/2][W/2][Cout] bf16 (raw); bn_sums / bn_sums_done as epi_conv2d_fwd
 *   epi_stem7x7s2_bwd_weight     s2d, dy [B][H/2][W/2][Cout] bf16 -> dwp [Cout][256] f32 (packed order)
 *   epi_stem7x7s2_unpack_weight_grad   dwp -> dw in the parameter's layout, EPI_F32 | EPI_BF16
 * workspace: epi_stem7x7s2_workspace_bytes.  No input gradient (the image needs none).
 * ------------------------------------------------------------------------------------------------ */
size_t epi_stem7x7s2_s2d_bytes(int B, int H, int W);
size_t epi_stem7x7s2_workspace_bytes(int B, int H, int W, int Cout);
int epi_stem7x7s2_s2d(const void* x, int x_dtype, int x_layout, int B, int H, int W, void* s2d, epi_stream_t stream);
int epi_stem7x7s2_pack_weight(const void* w_bf16, int channels_last, int Cout, void* wp, epi_stream_t stream);
int epi_stem7x7s2_unpack_weight_grad(const float* dwp, int Cout, int channels_last, void* dw, int dw_dtype, epi_stream_t stream);
int epi_stem7x7s2_fwd(const void* s2d, const void* wp, void* y, int B, int H, int W, int Cout, float* bn_sums, int* bn_sums_done,
                      void* workspace, size_t workspace_bytes, epi_stream_t stream);
int epi_stem7x7s2_bwd_weight(const void* s2d, const void* dy, float* dwp, int B, int H, int W, int Cout, void* workspace,
                             size_t workspace_bytes, epi_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * fp32-grade verification mode (round 3; epipolarpose_amd/models/precise.py, csrc/precise.hip) -- NOT on the training hot path.
 * The training path computes bf16 x bf16 -> fp32 with bf16 activations, so against the reference's fp32 network
 * (lib/models/pose3d_resnet.py:185-201, lib/core/integral_loss.py:140-160) it can only be held to a bf16 yardstick.  These entry
 * points let the SAME MFMA GEMM kernels carry fp32 activations: every operand is split into bf16 pieces x = hi + lo (+ lo2)
 * (epi_split_bf16) laid out along the GEMM's reduction dimension, the kernels accumulate in fp32 as always and write fp32
 * (the *_f32 entry points; epi_conv2d_bwd_weight / epi_deconv4x4s2_bwd_weight / epi_gemm_bf16 / epi_gemm_tn_bf16 already could),
 * and BatchNorm / max-pool / bias sums run on fp32 storage (same kernels instantiated for float where they are templates).
 *   epi_split_bf16: x [rows][C] f32 -> bf16 pieces; pieces[b] in {0: bf16(x), 1: bf16(x - hi), 2: bf16(x - hi - lo)};
 *                   row_concat 0: out [rows][nblk*C] (channel blocks -- forward / backward-data operands),
 *                   row_concat 1: out [nblk][rows][C] (row blocks -- weight-gradient operands); C % 4 == 0, nblk <= 8.
 * ------------------------------------------------------------------------------------------------ */
int epi_split_bf16(const float* x, long long rows, int C, const int* pieces, int nblk, int row_concat, void* out, epi_stream_t stream);
int epi_conv2d_fwd_f32(const void* x, const void* w, void* y, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                       void* workspace, size_t workspace_bytes, epi_stream_t stream);
int epi_conv2d_bwd_data_f32(const void* dy, const void* w_bwd, void* dx, int B, int H, int W, int Cin, int Cout, int KH, int KW,
                            int stride, int pad, void* workspace, size_t workspace_bytes, epi_stream_t stream);
int epi_deconv4x4s2_fwd_f32(const void* x, const void* w_phase, void* y, int B, int H, int W, int Cin, int Cout, void* workspace,
                            size_t workspace_bytes, epi_stream_t stream);
int epi_deconv4x4s2_bwd_data_f32(const void* dy, const void* w_bwd, void* dx, int B, int H, int W, int Cin, int Cout, void* workspace,
                                 size_t workspace_bytes, epi_stream_t stream);
int epi_bn_act_fwd_f32(const void* x, const void* residual, long long R, int C, const float* gamma, const float* beta, float eps,
                       float momentum, int training, int relu, float* running_mean, float* running_var,
                       long long* num_batches_tracked, float* mean, float* rstd, float* scale_shift, float* sums_ws, float* bwd_sums,
                       void* y, epi_stream_t stream);
int epi_bn_act_bwd_f32(const void* dy, const void* x, const void* y, long long R, int C, const float* gamma, const float* mean,
                       const float* rstd, const float* scale_shift, int relu, float* dbeta_dgamma, void* dx, void* dres,
                       float* fwd_sums_clear, float* param_grads, epi_stream_t stream);
int epi_column_sums_f32(const void* x, long long R, int C, float* sums, epi_stream_t stream);
int epi_maxpool3x3s2_fwd_f32(const void* x, void* y, void* pos, int B, int H, int W, int C, epi_stream_t stream);
int epi_maxpool3x3s2_bwd_f32(const void* dy, const void* pos, void* dx, int B, int H, int W, int C, epi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* EPIPOLAR_HIP_H */
