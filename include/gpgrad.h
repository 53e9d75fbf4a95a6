/* gpgrad.h -- C ABI of the MI355X-native gradient-enhanced GP likelihood hot path.
 *
 * This is the drop-in boundary: every entry point replaces one step of the reference
 * (marchildon/gpgradpy @ v2; citations are file:line in that tree).  Plain pointers and sizes only;
 * the caller owns every host buffer, the library owns every device allocation.
 *
 * Return convention (all int-returning functions):
 *    0   success
 *   >0   LAPACK-style info: 1-based index of the first non-positive pivot met by the Cholesky
 *        factorisation -> the caller treats it like the `except` branch of Kernel.py:253-264
 *        (Kcov_chofac = None, b_chofac_good = False)
 *   <0   argument / HIP runtime error; text via gpg_last_error()
 *
 * Threading: one context = one device + one HIP stream, not re-entrant (the reference object is not
 * thread-safe either: memoised `_last_hp_vec`, mutable `_time_chofac`).
 *
 * Matrix layout: derivative-major blocks (KernelSqExp.py:381-408, CommonFun.py:151-173): row
 * r = blk * n + a, blk 0 = function value, blk i+1 = d/dx_i at point a.  Device storage is
 * column-major, lower triangle, padded to a multiple of 128.
 */
#ifndef GPGRAD_H
#define GPGRAD_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gpg_ctx gpg_ctx;

/* RatQu = the rational quadratic kernel of KernelRatQuad.py, with its own hyperparameter alpha (gpg_hp.hp_kernel). */
enum { GPG_KERNEL_SQEXP = 0, GPG_KERNEL_MA5F2 = 1, GPG_KERNEL_RATQU = 2 };
enum { GPG_WELLCOND_BASE = 0, GPG_WELLCOND_PRECON = 1 };

/* Hyperparameters + regularisation of ONE evaluation.
 * Mirrors HparaOptzVal (GpHpara.py:12-19) plus the scalars calc_all_K_w_chofac reads from the object
 * (Kernel.py:186-236: theta, varK, noise, self._etaK, self.wellcond_mtd). */
typedef struct {
  const double* theta; /* [d] > 0                                                              */
  double varK_mat;     /* varK the matrix is built with: 1 for noise-free data and for          */
                       /* b_normlz_w_varK (Kernel.py:137-138,196-197), hp_vals.varK otherwise   */
  double var_fval;     /* >= 0: noise variance of every function value (unknown-noise case,     */
                       /*       Kernel.py:348); < 0: use the per-row vector given to set_data   */
  double var_fgrad;    /* same for every gradient entry (Kernel.py:355)                         */
  double eta;          /* nugget etaK (GpWellCond.py:116-154, Kernel.py:229-236)                */
  int wellcond;        /* GPG_WELLCOND_PRECON (Kernel.py:220-266) or _BASE (Kernel.py:268-302)  */
  int closed_form_varK;/* 1: noise-free path, varK = max(1e-32, r'K^-1 r / N) (CalcLkd.py:159); */
                       /* 0: noisy path, ln_lkd = -(ln_det + r'K^-1 r)/2 (CalcLkd.py:226)       */
  double hp_kernel;    /* hyperparameter of the kernel itself (HparaOptzVal.kernel): alpha > 0   */
                       /* of GPG_KERNEL_RATQU (KernelRatQuad.py:468); ignored by the others      */
} gpg_hp;

/* Result of one likelihood evaluation = the scalar fields of LkdInfo (CalcLkd.py:14-26). */
typedef struct {
  double ln_lkd;  /* marginal log-likelihood, without the N ln(2 pi) constant (CalcLkd.py:168,226) */
  double ln_det;  /* 2 sum log diag(P L)                               (CalcLkd.py:165,225)       */
  double beta;    /* GLS constant mean                                  (GpMeanFun.py:103-107)     */
  double varK;    /* closed-form varK, or varK_mat on the noisy path    (CalcLkd.py:159)           */
  double rKr;     /* r' Kcov^-1 r with r = y - V beta                   (CalcLkd.py:154,221)       */
  int info;       /* 0, or first non-positive pivot (1-based)                                      */
  int pad_;
} gpg_lkd_out;

/* Lifetime --------------------------------------------------------------------------------------- */

/* Replaces GaussianProcess.__init__ + the shape part of set_data (GaussianProcess.py:138-190,
 * 244-262).  n_eval points of dimension dim; use_grad != 0 -> N = n_eval * (dim + 1) rows.
 * Allocates the (N_pad + 128) x N_pad fp64 workspace once.  dim <= 16. */
int gpg_create(gpg_ctx** out, int device, int n_eval, int dim, int use_grad, int kernel);
void gpg_destroy(gpg_ctx* ctx);
const char* gpg_last_error(const gpg_ctx* ctx); /* ctx may be NULL: error of the last failed create */

/* Optional, before gpg_set_data: replaces the bvec_use_grad handling of set_data / calc_KernGrad
 * (GaussianProcess.py:246-262, KernelSqExp.py:349-377).  use_grad_pt [n_eval] != 0 marks the points whose
 * gradient enters the model; NULL restores "all".  N becomes n_eval + n_grad * dim and data_vec / noise_var
 * of gpg_set_data are [N] with the gradient part laid out [n_grad, dim] column-major (CommonFun.py:170-171). */
int gpg_set_grad_mask(gpg_ctx* ctx, const unsigned char* use_grad_pt);

/* Replaces the data ingest of set_data (GaussianProcess.py:296-302,363) and make_data_vec
 * (CommonFun.py:151-173).  x [n_eval, dim] row-major; data_vec [N] = [f, d1 f(all pts), ...];
 * noise_var [N] = known noise variances in the same ordering (Kernel.py:343-353) or NULL (zeros).
 * The [d, n, n] Rtensor of the reference is never materialised. */
int gpg_set_data(gpg_ctx* ctx, const double* x, const double* data_vec, const double* noise_var);

/* Replace the per-row noise variances only (host [N], NULL = zeros): what calc_all_K_w_chofac(..., noise_vec = ...) passes in
 * (Kernel.py:140-143, 207-208, 218).  Likelihood-side state is invalidated; a posterior set up by gpg_setup_eval keeps its factor. */
int gpg_set_noise(gpg_ctx* ctx, const double* noise_var);

/* Likelihood ------------------------------------------------------------------------------------- */

/* Replaces CalcLkd.calc_lkd_all(hp, calc_lkd=True, calc_cond=False, calc_grad=False)
 * (CalcLkd.py:270-346): assembly + noise + preconditioner + nugget (Kernel.py:213-237), Cholesky
 * (Kernel.py:251), GLS mean (GpMeanFun.py:69-122), alpha-free evaluation of r'K^-1 r and ln det
 * (CalcLkd.py:149-181 / 185-251).  Returns out->info as well. */
int gpg_lkd(gpg_ctx* ctx, const gpg_hp* hp, gpg_lkd_out* out);

/* Replaces CalcLkd.calc_lkd_all(hp, calc_grad=True) with the adjoint method (CalcLkd.py:170-177, 230-235;
 * kernel derivatives GpHparaGrad.py:13-155, KernelSqExp.py:470-568, KernelMatern5f2.py:532-642).
 * out as gpg_lkd.  g_aa, g_inv hold dim + 4 entries: hyperparameter slot k = theta_0..theta_(dim-1), varK, var_fval,
 * var_fgrad, hp_kernel (the last one zero unless the kernel has a hyperparameter of its own: alpha of GPG_KERNEL_RATQU,
 * KernelRatQuad.py:752-843):
 *   g_aa[k]  = sum_{r,c} G_k[r,c] alpha_r alpha_c,    g_inv[k] = sum_{r,c} G_k[r,c] (-1/2 Kcov^-1)[r,c],
 * G_k = d Kcov / d hp_k, alpha = Kcov^-1 (y - V beta).  The caller forms ln_lkd_grad = s g_aa + g_inv with
 * s = 1/2 (noisy, CalcLkd.py:233) or pnlt'/N + 1/(2 varK) (noise-free, CalcLkd.py:173-175).  Neither the
 * [n_hp, N, N] derivative tensor nor a dense identity solve is formed; costs two more N^3/3 sweeps. */
int gpg_lkd_grad(gpg_ctx* ctx, const gpg_hp* hp, gpg_lkd_out* out, double* g_aa, double* g_inv);

/* Replaces the serial restart loop of GpHparaX0.select_hp_optz_x0 (GpHparaX0.py:33-59) on ONE
 * device: m hyperparameter rows, hp_rows [m, row_len] with row = [theta(d), varK_mat, var_fval,
 * var_fgrad (, hp_kernel if row_len >= d + 4: alpha of GPG_KERNEL_RATQU)] already decoded from log10
 * (GpHpara.py:56-103); eta / wellcond / closed_form_varK are shared (row_len > dim + 4: column dim + 4 carries a nugget per row instead).  out [m].  Rows whose factorisation fails get info > 0 and NaN ln_lkd (GpHparaX0.py:34,
 * 43-45).  Returns 0 unless an argument / runtime error occurred.  All m evaluations are queued on
 * the stream back-to-back and synchronised once. */
int gpg_lkd_batch(gpg_ctx* ctx, int m, const double* hp_rows, int row_len, double eta, int wellcond,
                  int closed_form_varK, gpg_lkd_out* out);

/* Value AND adjoint gradient of m restart rows in one call: what m SLSQP runs of the reference's multi-start
 * (OptzLkd.py:249-290, one calc_store_likelihood per iterate: OptzLkd.py:15-100) ask for when they advance in lock step.
 * Rows as gpg_lkd_batch; out [m]; g_aa, g_inv [m, dim + 4] row-major with the slots of gpg_lkd_grad.  Rows whose
 * factorisation fails get info > 0 and NaN gradients.  The factorisations, the two N^3/3 sweeps of the explicit inverse
 * (W = L^-T, -(W W^T)) run as ONE dataflow launch each per group of up to 8 rows; results are bit-identical to m calls
 * of gpg_lkd_grad. */
int gpg_lkd_grad_batch(gpg_ctx* ctx, int m, const double* hp_rows, int row_len, double eta, int wellcond,
                       int closed_form_varK, gpg_lkd_out* out, double* g_aa, double* g_inv);

/* Several devices, one process -------------------------------------------------------------------- */

/* The restart table of GpHparaX0.select_hp_optz_x0 (GpHparaX0.py:33-59) sharded over the devices of ONE process: one
 * context per listed device (devices == NULL: 0 .. ndev-1; a device may be listed more than once), one host thread per
 * context, contiguous row blocks (rows differ by at most one between devices), the results gathered in `out` [m] and
 * the row of the highest ln_lkd among the successfully factorised ones in *best (-1 if none; may be NULL).  The data
 * set is replicated by gpg_multi_set_data.  This is the in-library form of the partition DESIGN.md section 5 runs with
 * one process per GPU over torch.distributed / RCCL. */
typedef struct gpg_multi gpg_multi;
int gpg_multi_create(gpg_multi** out, int ndev, const int* devices, int n_eval, int dim, int use_grad, int kernel);
void gpg_multi_destroy(gpg_multi* m);
const char* gpg_multi_last_error(const gpg_multi* m); /* m may be NULL: error of the last failed create */
int gpg_multi_count(const gpg_multi* m);
int gpg_multi_set_data(gpg_multi* m, const double* x, const double* data_vec, const double* noise_var);
int gpg_multi_lkd_batch(gpg_multi* m, int nrows, const double* hp_rows, int row_len, double eta, int wellcond,
                        int closed_form_varK, gpg_lkd_out* out, int* best);

/* Posterior -------------------------------------------------------------------------------------- */

/* Replaces GpEvalModel.setup_eval_model (GpEvalModel.py:17-57): factorises the matrix for hp (the
 * caller passes varK_mat = 1, noise undivided: the b_normlz_w_varK quirk of Kernel.py:196-197,218)
 * and keeps the factor plus alpha = Kcov^-1 (y - V beta) on the device.  alpha_out [N] may be NULL.
 * The factor lives in a workspace of its own (allocated by the first call: one more (N_pad + 128) x N_pad fp64
 * array): gpg_lkd / gpg_lkd_grad / gpg_lkd_batch / gpg_get_matrix calls that follow do NOT disturb it, and
 * gpg_predict* keep answering from it until the next gpg_setup_eval / gpg_set_data -- as KernEta_chofac of the
 * reference survives later calc_lkd_all calls. */
int gpg_setup_eval(gpg_ctx* ctx, const gpg_hp* hp, double beta, double* alpha_out);

/* Replaces GpEvalModel.eval_model(x, calc_grad=False) (GpEvalModel.py:59-198): cross-kernel
 * (GpEvalModel.py:133-139), K^-1 Kyx (GpEvalModel.py:154), mu = beta + Kyx' alpha (:168),
 * sig = sqrt(max(0, 1 - diag(Kxy K^-1 Kyx))) * sqrt(varK) (:162-166).  xq [nx, dim] row-major.
 * sig2_raw [nx] (may be NULL) receives the unclipped 1 - diag(...) that the reference asserts to be
 * non-negative (GpEvalModel.py:163). */
int gpg_predict(gpg_ctx* ctx, int nx, const double* xq, double varK, double* mu, double* sig,
                double* sig2_raw);

/* Replaces eval_model(x, calc_grad=True) (GpEvalModel.py:133-140,170-172,319-354): additionally returns
 * dmudx, dsigdx [nx, dim] row-major.  The derivative blocks of the cross-kernel are recomputed on the fly
 * (the [N, nx*dim] matrix dKxy_dx is never formed); K^-1 Kyx costs one extra triangular sweep per query. */
int gpg_predict_grad(gpg_ctx* ctx, int nx, const double* xq, double varK, double* mu, double* sig,
                     double* sig2_raw, double* dmudx, double* dsigdx);

/* Replaces GpEvalModel.eval_model_var(x, calc_grad) (GpEvalModel.py:200-317): the variance form of the posterior,
 * sig2 [nx] = varK (1 - diag(Kxy K^-1 Kyx)) (:297, NOT clipped at zero: the reference asserts min >= 0 on it) and, when
 * dsig2dx != NULL, its gradient [nx, dim] row-major = -2 varK sum_r dKxy_dx[., r] (K^-1 Kyx)[r] (calc_dsig2dx :327-337;
 * defined where sig = 0, unlike dsigdx). */
int gpg_predict_var(gpg_ctx* ctx, int nx, const double* xq, double varK, double* sig2, double* dsig2dx);

/* Posterior Hessians at ONE query point xq[d] (the reference evaluates them one point per call):
 * d2mudx2[d*d], d2sigdx2[d*d] row-major, plus everything gpg_predict_grad returns for that point.
 * Replaces eval_model(calc_grad=True, calc_hess=True) -- GpEvalModel.py:170-181, calc_d2mudx2 :355-363,
 * calc_d2sigdx2 :365-380, calc_Kern_hess_x (KernelSqExp.py:48-88,412-468, KernelMatern5f2.py:54-94,453-530).
 * d2sigdx2 is NaN where sig == 0, as in the reference. */
int gpg_predict_hess(gpg_ctx* ctx, const double* xq, double varK, double* mu, double* sig, double* dmudx,
                     double* dsigdx, double* d2mudx2, double* d2sigdx2);

/* Materialisation on request (the 7-tuple of Kernel.py:307 carries N x N arrays; the fast path never
 * copies them).  out is [N, N] column-major == row-major (symmetric) for which = 0..2:
 *   0 Kern (Kernel.py:213-216), 1 Kcov (Kernel.py:237 / 277), 2 the matrix that is factorised
 *   (Kcov_precon, Kernel.py:236; equals Kcov for wellcond = base);
 *   3 the Cholesky factor of the most recent factorisation (gpg_lkd / gpg_lkd_grad / gpg_setup_eval), as the
 *     lower-triangular P L of Kernel.py:252 (row-major [N, N], zeros above the diagonal) usable with
 *     scipy.linalg.cho_solve((out, True), b);
 *   4 the same for the factor KEPT by the last successful gpg_setup_eval, whatever likelihood calls came after it
 *     (hp ignored for 3 and 4). */
int gpg_get_matrix(gpg_ctx* ctx, const gpg_hp* hp, int which, double* out);

/* Kernel table of the reference (Kernel.py:27-126): calc_KernBase / calc_KernGrad as functions of the difference
 * tensor, exactly as the reference's bound attributes take it -- replaces sq_exp_calc_KernBase / _KernGrad
 * (KernelSqExp.py:16-46, 320-410), matern_5f2_calc_KernBase / _KernGrad (KernelMatern5f2.py:16-52, 352-450),
 * rat_quad_calc_KernBase / _KernGrad (KernelRatQuad.py:439-554).  rtensor [dim, n1, n2] row-major with
 * R[k, a, b] = X1[a, k] - X2[b, k] (CommonFun.py:56-84); two DIFFERENT point sets are fine.  use_grad = 0: out
 * [n1, n2] = KernBase.  use_grad != 0: out [n1 + n1g dim, n2 + n2g dim] row-major = KernGrad in the derivative-major
 * block layout, where use_grad1 [n1] / use_grad2 [n2] (NULL = all) are bvec_use_grad1 / 2 and n1g, n2g their counts.
 * No context: a self-contained call on `device` (uploads the tensor, downloads the matrix); errors through
 * gpg_last_error(NULL). */
int gpg_kern_rtensor(int device, int kernel, int dim, int n1, int n2, const double* rtensor, const double* theta,
                     double hp_kernel, int use_grad, const unsigned char* use_grad1, const unsigned char* use_grad2,
                     double* out);

/* Products with the matrix that was factorised last (gpg_lkd / gpg_setup_eval), through its factor in HBM:
 * op 0: out = (L L^T) v, op 1: out = (L L^T)^-1 v; v, out host [N].  L L^T is Kcov_precon = varK (Kcor + eta I)
 * for 'precon' and Kcov for 'base' -- the matrices whose 2-norm condition number Kernel.py:239-245, 279-285 report.
 * The host side runs a Lanczos iteration on these two operators to get lambda_max and 1 / lambda_min
 * (gpgradpy_amd/cond_number.py; replaces np.linalg.cond of an N x N matrix). */
int gpg_factor_apply(gpg_ctx* ctx, int op, const double* v, double* out);

/* d K / d theta_k [dim, N, N] and, for GPG_KERNEL_RATQU, d K / d alpha [N, N] (either may be NULL) of the kernel matrix of ONE point
 * set with itself from its difference tensor rtensor [dim, n, n] (R[k, a, b] = x[a, k] - x[b, k]); N = n (dim + 1) with use_grad, n
 * without; host arrays, row-major.  Replaces the kernel-table entries calc_KernBase_grad_th / calc_KernGrad_grad_th /
 * calc_Kern*_grad_alpha (Kernel.py:43-48, 69-75, 96-102; KernelSqExp.py:91-123, 470-568, KernelMatern5f2.py:100-135, 532-642,
 * KernelRatQuad.py:133-163, 640-840) for callers that want the tensors themselves (the reference's unit_test/test_grad_Kmat.py,
 * calc_KernGrad_hp / calc_Kcov_grad_hp); the likelihood gradient never forms them (gpg_lkd_grad). */
int gpg_kern_rtensor_grad_hp(int device, int kernel, int dim, int n, const double* rtensor, const double* theta, double hp_kernel,
                             int use_grad, double* out_theta, double* out_alpha);

/* The x-derivative entries of the kernel table, from the difference tensor rtensor [dim, n1, n2] of two point sets (R = X1 - X2):
 * out [dim, n1 dim, n2 (+ n2g dim with use_grad)], out[k][i n1 + a][b] = d2 K(a, b) / d x1_i d x1_k and, in gradient column (j, b),
 * d3 K / d x1_i d x1_k d x2_j; use_grad2 [n2] masks the gradient columns (NULL = all).  Replaces calc_KernBase_hess_x (use_grad = 0) /
 * calc_KernGrad_grad_x (use_grad = 1) (Kernel.py:56-57, 82-83, 109-110; KernelSqExp.py:48-88, 412-468); the posterior Hessian itself
 * (gpg_predict_hess) never forms the tensor. */
int gpg_kern_rtensor_hess_x(int device, int kernel, int dim, int n1, int n2, const double* rtensor, const double* theta, double hp_kernel,
                            int use_grad, const unsigned char* use_grad2, double* out);

/* The reference's matrix-dependent nugget (cond_eta_is_const = False, i.e. wellcond_mtd 'rescale_eta_vary'; Kernel.py:229-236,
 * 269-276): rowsum[i] = sum_j |M_ij| over the N rows of M = Kcor = P^-1 (K + diag(noise / varK)) P^-1 (wellcond PRECON) or of the
 * kernel matrix K itself (BASE), host [N]; hp->eta is ignored.  The caller takes argmax / max and sets
 * eta = rowsum[argmax] / (cond_max_target - 1) in the gpg_hp of the evaluation that follows. */
int gpg_abs_rowsum(gpg_ctx* ctx, const gpg_hp* hp, double* rowsum);

/* alpha = Kcov^-1 (y - V beta), host [N], of the most recent successful gpg_lkd_grad on this context (no likelihood call in between):
 * with gpg_factor_apply and gpg_dcov_quadform it gives the remaining outputs of the reference's gradient code without the
 * [n_hp, N, N] tensors -- hp_beta_grad_k = -u' G_k alpha with u = Kcov^-1 V / (V' Kcov^-1 V) (GpMeanFun.py:114-117, as the polarisation
 * (q_k(u + alpha) - q_k(u - alpha)) / 4 of the quadratic form), hp_varK_grad_k = -alpha' G_k alpha / N (CalcLkd.py:112-113; V' alpha = 0),
 * ln_det_Kmat_grad_k = tr(Kcov^-1 G_k) = -2 g_inv[k] (CalcLkd.py:361-365). */
int gpg_lkd_alpha(gpg_ctx* ctx, double* alpha);

/* eta >= 0: the derivatives d Kcov / d hp_k (gpg_lkd_grad, gpg_lkd_grad_batch, gpg_dcov_quadform, gpg_cond_fro) are formed with
 * this nugget instead of gpg_hp.eta -- the reference differentiates with self._etaK even when the matrix was built with the
 * row-sum nugget above (GpHparaGrad.py:43,107,126).  eta < 0 (default): use gpg_hp.eta. */
int gpg_set_gradient_nugget(gpg_ctx* ctx, double eta);

/* out[k] = v^T (d Kcov / d hp_k) v for the dim + 4 hyperparameter slots of gpg_lkd_grad (theta, varK, var_fval,
 * var_fgrad, hp_kernel), v host [N]; call after a successful gpg_lkd / gpg_setup_eval with the same hp (it uses that
 * call's preconditioner vector).  With the extreme eigenvectors of the host-side Lanczos runs this gives the
 * gradient of the condition number, d cond / d hp_k = (q_k(v_max) - cond q_k(v_min)) / lambda_min
 * (GpHparaCon.py:163-207), without the [n_hp, N, N] derivative tensor or a dense eigendecomposition. */
int gpg_dcov_quadform(gpg_ctx* ctx, const gpg_hp* hp, const double* v, double* out);

/* Frobenius-norm condition number of the matrix that is factorised for hp -- cond_norm = 'fro' of the reference:
 * np.linalg.cond(Kcov_precon | Kcov, 'fro') = ||K||_F ||K^-1||_F (Kernel.py:239-245, 279-285; calc_cond_fronorm_w_grad,
 * GpHparaCon.py:209-236) -- and, when cond_grad != NULL (dim + 4 slots as gpg_lkd_grad; wellcond 'base' only, as in the
 * reference), its gradient sum_{r,c} (frac K - K^-3 / frac)[r,c] (d K / d hp_k)[r,c], frac = ||K^-1||_F / ||K||_F.
 * Everything stays on the device: the norms are reductions over the assembled matrix and over the explicit inverse of
 * the adjoint-gradient path, K^-2 and K^-3 are two full MFMA products, the contraction recomputes d K / d hp_k on the fly.
 * Returns > 0 (first non-positive pivot) when the Cholesky fails. */
int gpg_cond_fro(gpg_ctx* ctx, const gpg_hp* hp, double* cond, double* cond_grad);

/* Instrumentation (replaces the wall-clock accumulator _time_chofac, Kernel.py:247,304-305) ------- */

enum {
  GPG_PROF_ASSEMBLY = 0,   /* fused kernel build: algorithmic bytes 8 N (N+1) / 2                   */
  GPG_PROF_POTRF = 1,      /* diagonal-block factorisations                                         */
  GPG_PROF_TRSM = 2,       /* panel triangular solves                                               */
  GPG_PROF_GEMM_PANEL = 3, /* updates inside a panel                                                */
  GPG_PROF_GEMM_TRAIL = 4, /* fp64 MFMA factorisation kernel (dataflow Cholesky launch, or the trailing
                            * update of the blocked schedule): algorithmic flops                      */
  GPG_PROF_REDUCE = 5,     /* log-det / GLS reductions                                              */
  GPG_PROF_NCAT = 6
};
/* mask: bit c enables HIP-event timing of category c launches on the context's stream. */
int gpg_prof_enable(gpg_ctx* ctx, unsigned mask);
/* Synchronises, then returns per category: ms[c] summed event time, count[c] launches,
 * work[c] algorithmic units (bytes for ASSEMBLY, flops otherwise) since the last call; resets. */
int gpg_prof_read(gpg_ctx* ctx, double ms[GPG_PROF_NCAT], long long count[GPG_PROF_NCAT],
                  double work[GPG_PROF_NCAT]);

/* Tuning knobs (defaults are the measured best): panel width (multiple of 128 in [128, 1024]). */
int gpg_set_panel(gpg_ctx* ctx, int nb_outer);
/* bit 0 = 1 (default): look-ahead Cholesky (the next diagonal block is factorised on a second,
 * high-priority stream under the trailing update); 0: single stream.  bit 1 = 1: register-staged
 * 128x128 update kernel instead of the LDS-DMA one (A/B measurements only). */
int gpg_set_lookahead(gpg_ctx* ctx, int on);

/* Factorisation schedule.  GPG_FACTOR_AUTO (default): one dataflow launch -- the 64 x 64-tile kernel up to
 * 12288 padded columns, the 128 x 128-tile kernel above (batched launches of gpg_lkd_batch switch to the 128-tile
 * kernel much earlier, see gpg_set_batch).  GPG_FACTOR_BLOCKED: right-looking blocked
 * algorithm (panel solve + trailing update per panel, look-ahead on a second stream; A/B measurements and
 * the reference point for the parity tests).  GPG_FACTOR_TILE64 / GPG_FACTOR_TILE128 force one kernel. */
enum gpg_factor_mode { GPG_FACTOR_AUTO = 0, GPG_FACTOR_BLOCKED = 1, GPG_FACTOR_TILE64 = 2, GPG_FACTOR_TILE128 = 3 };
int gpg_set_factor_mode(gpg_ctx* ctx, int mode);
/* Two dataflow launches sharing one device (two processes / contexts on the same GPU) can starve each other: every
 * wait inside the launch is bounded (0.25 s), the launch then drains, and the library repeats the call with the
 * blocked schedule and keeps the context on it until gpg_set_factor_mode is called again.  Returns how often that
 * happened (results are unaffected; rc -4 is only returned if the blocked repeat fails too). */
int gpg_factor_fallbacks(gpg_ctx* ctx);
/* The same bounded waits guard the overlapped inverse of gpg_lkd_grad / gpg_lkd_grad_batch (W = L^-T launched behind the
 * factorisation it waits for) and the dataflow triangular solves.  A timeout there does NOT cost the dataflow factorisation:
 * the call is repeated once without the overlap (the context stops overlapping), respectively with the blocked solve sweeps
 * (the context keeps them for its solves), until gpg_set_factor_mode re-arms both.  These return how often each happened. */
int gpg_overlap_fallbacks(gpg_ctx* ctx);
int gpg_solve_fallbacks(gpg_ctx* ctx);
/* Schedule of the most recent factorisation launch (for logs / bench): *kernel = 0 blocked, 1 dataflow with 64 x 64
 * tiles, 2 dataflow with 128 x 128 tiles, 3 dataflow with pairs of 128 x 128 tiles per workgroup (batched launches of large
 * matrices); *matrices = how many matrices that launch factorised.  Either may be NULL. */
int gpg_last_factor(gpg_ctx* ctx, int* kernel, int* matrices);
/* Batched launches of the 128 x 128-tile schedule (gpg_lkd_batch on large matrices) can give every 512-thread workgroup a PAIR of tiles of
 * one tile column: both tiles then share the row panel they read as the second operand (a quarter less memory traffic), at the
 * price of finalisations that are not hidden behind another workgroup's MFMA loop; measured 1-3 % faster for every batched size.
 * mode 0: never, 1: always (also for one matrix per launch, where it is 1-2 % slower), 2 (default): whenever a launch holds at
 * least two matrices.  Results are bit-identical in all three modes. */
int gpg_set_pair_mode(gpg_ctx* ctx, int mode);

/* Caps the number of persistent workgroups of every dataflow launch (0 = default: as many as the device holds at once).  The
 * launches are correct for ANY number >= 1 -- a workgroup only ever waits for tasks with smaller tickets, all of which are held
 * by running workgroups or finished -- which is what tests/test_gpu_smoke.py checks with 1, 3 and 17 workgroups; also a way to
 * leave part of the device to another tenant. */
int gpg_set_max_workgroups(gpg_ctx* ctx, int n);

/* gpg_lkd_batch: up to max_matrices restart rows are assembled into separate workspaces and factorised by ONE
 * dataflow launch (task lists interleaved tile column by tile column).  A single small factorisation is
 * latency-bound and leaves most of the chip idle; a large one does so at its two ends.  Results are bit-identical to
 * the one-at-a-time path.  -1 (default): automatic (8 .. 64 rows, fewer the larger the matrix, 8 from ~18000 padded
 * columns up to 32768, 1 above, where 2 .. 3 can be asked for explicitly); 0 / 1: off.  With enough rows per launch (rows x Npad / 128 >= 320, Npad >= 2048)
 * the 128 x 128-tile kernel is used whatever the single-matrix choice of GPG_FACTOR_AUTO would be. */
int gpg_set_batch(gpg_ctx* ctx, int max_matrices);
/* Allocates the batch workspaces a gpg_lkd_batch call with `rows` rows will use (otherwise done by the first such
 * call): setup, like gpg_set_data, for callers that time gpg_lkd_batch. */
int gpg_reserve_batch(gpg_ctx* ctx, int rows);

/* Library / device facts for logs: writes "gfx950 MI355X ..." style text. */
int gpg_device_info(int device, char* buf, int buflen);

#ifdef __cplusplus
}
#endif
#endif /* GPGRAD_H */
