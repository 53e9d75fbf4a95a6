// Internal declarations shared by the HIP translation units of libgpgrad_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <map>
#include <string>
#include <vector>
#include "../../include/gpgrad.h"

#define GPG_MAX_DIM 16
#define GPG_GRAD_SLOTS_MAX (GPG_MAX_DIM + 4)   // theta(d), varK, var_fval, var_fgrad, hp_kernel
#define GPG_TILE 128      // padding / GEMM tile granularity
#define GPG_NBI 64        // inner (diagonal block) width of the panel factorisation
#define GPG_INFO_INTERNAL 0x7fffffff   // info value: dataflow factorisation aborted (dependency wait timed out)
#define GPG_RHS_ROWS 128  // right-hand-side rows appended below the matrix (ride along the Cholesky)

// Arguments of the fused assembly kernels (passed by value -> SGPRs / kernarg segment).
struct AsmParams {
  int n, d, use_grad, kernel;
  int ng;          // number of points that carry a gradient (== n unless a bvec_use_grad mask is set)
  const int* gpos; // [n] position of point a among the gradient points, -1 if its gradient is not used
  int N, Npad, ld;
  int precon;      // 1: write varK*(P^-1 Kw P^-1 + eta I); 0: varK*(Kw + eta I)
  int mode;        // 0: matrix to factorise, 1: raw Kern, 2: Kcov (P Kp P for precon)
  double varK, eta;
  double hp_kernel; // alpha of the rational quadratic kernel
  double theta[GPG_MAX_DIM];
};

// One matrix of a batched likelihood evaluation (gpg_lkd_batch): what the small kernels need besides the shape.
struct gpg_batch_item { AsmParams p; double var_fval, var_fgrad; };

struct TileMap { int* dev; int n; };

struct ProfEvent { hipEvent_t e0, e1; int cat; };

// One factorisation workspace with the vectors that belong to the matrix in it.  The context keeps two: set 0 serves the
// likelihood evaluations (gpg_lkd / gpg_lkd_grad / gpg_lkd_batch / gpg_get_matrix 0..2), set 1 holds the factor kept by
// gpg_setup_eval for gpg_predict* -- the reference keeps KernEta_chofac untouched by later likelihood calls
// (GpEvalModel.py:17-57), so the posterior must not share storage with them.  Set 1 is allocated by the first
// gpg_setup_eval.  The "active" set is mirrored in gpg_ctx::A, dvec, ... (what every launch helper reads).
struct gpg_ws {
  double *A = nullptr, *dvec = nullptr, *invp = nullptr, *dinv = nullptr, *zvec = nullptr, *tmpv = nullptr;
  bool factor_valid = false, prep_valid = false;
  int precon = 0;
};

struct gpg_ctx {
  int device = 0;
  hipStream_t stream = nullptr;      // main stream
  hipStream_t stream_upd = nullptr;  // high-priority stream: look-ahead factorisation of the next diagonal block (blocked schedule)
  hipStream_t stream_inv = nullptr;  // LOWEST-priority stream: the overlapped inverse W = L^-T, which waits for the factorisation on `stream`
  int lookahead = 1;
  int gemm_impl = 1;                 // 1: LDS-DMA ring kernel for the 128x128 updates, 0: register-staged kernel (A/B runs)
  std::vector<hipEvent_t> ev_panel, ev_upd;
  std::map<unsigned long long, TileMap> tilemaps;   // live-tile lists of the trailing updates, per shape
  int n = 0, d = 0, use_grad = 0, kernel = 0;
  int N = 0, Npad = 0, R = GPG_RHS_ROWS, ld = 0;
  int ng = 0;                // gradient points in use (KernelSqExp.py:349-377: bvec_use_grad)
  int* gpos = nullptr;       // device [n]
  size_t A_elems = 0;        // allocated size of A (doubles), sized for all gradients
  int panel_impl = 1;                // 1: fused panel_solve_kernel for B_p, 0: trsm64 + small gemm launches (A/B runs)
  int nb_outer = 256;   // panel width
  bool launch_error = false;  // a launch helper could not allocate its scratch / task list: nothing was launched (checked by the API call)
  int factor_fallbacks = 0;   // times a dataflow launch timed out and the call was repeated with the blocked schedule
  int last_factor_kernel = 0; // schedule of the most recent factorisation launch: 0 blocked, 1 64-tile, 2 128-tile dataflow, 3 128-tile pairs
  bool last_launch_paired = false;
  int last_factor_batch = 0;  // matrices it factorised
  int chol_impl = 0;    // 1: whole factorisation by the 128-tile dataflow kernel
  int tail_cols = 0;    // trailing block of at most this many columns goes to the dataflow tile kernel (0: off)
  int rows_max_tasks = 1 << 15;  // dataflow row solves (posterior at many points): at most this many 64 x 64 tile tasks per launch; beyond, the
                                 // blocked forward sweep is as fast (cfg3: ~5000 points; tools/post_many.py; env GPG_ROWS_MAX_TASKS overrides)
  int inv_tile64_cols = 4096;   // explicit inverse: W = L^-T on 64 x 64 tiles up to this many padded columns (0: always 128-tiles)
  int* tile_flags = nullptr;   // device: completion flags of the dataflow kernel + abort word + ticket counter
  // value + gradient of ONE small matrix: W = L^-T is launched on the second stream while the factorisation is still running and waits
  // for the factorisation's own diagonal-tile flags (gpg_overlap_inverse_* in cholesky_dataflow.hip).  Both launches keep their flags in
  // keep_flags, which nothing else clears meanwhile.
  int overlap_inverse = 2;     // 0: off, 1: W = L^-T behind the factorisation, 2: -(W W^T) behind W as well (env GPG_OVERLAP_INVERSE)
  int* keep_flags = nullptr;
  size_t keep_flags_cap = 0;
  int* chol_flags_override = nullptr;   // launch_tile_chol: use this (large enough) buffer for the next launch and record ev_flags after clearing it
  hipEvent_t ev_flags = nullptr, ev_trinv = nullptr;
  hipEvent_t ev_winit = nullptr;   // recorded on stream_inv after W's flags / abort word / tickets / identity are set and before its kernel starts
  int grid_cap = 0;            // > 0: cap on the next persistent launches' grids (overlapped inverse: never every co-resident slot)
  int overlap_fallbacks = 0;   // times a call with the overlapped inverse timed out and was repeated without the overlap
  bool overlap_used = false;   // the call in progress launched the overlapped inverse
  int fail_kind = 0;           // what the last -4 came from: 1 factorisation (or unknown), 2 a dataflow triangular solve
  int solve_dataflow = 1;      // triangular solves as dataflow launches (0 after one of them timed out: blocked sweeps)
  int solve_fallbacks = 0;
  int num_cus = 0;             // compute units of the device (grid of the persistent launches)
  bool alpha_valid = false;    // zvec of workspace set 0 holds p * alpha of the last gpg_lkd_grad (gpg_lkd_alpha)
  bool zero_info_in_prep = false;    // the next gpg_launch_prep also clears *c->info (consumed by it)
  int fuse_subdiag_max_tiles = 48;   // ... up to this many 64-column tile columns (3072 matrix columns)
  int fuse_subdiag = 1;        // 64-tile factorisation: diagonal tasks own the sub-diagonal tile below the previous diagonal tile (tile_chol_task)
  int pair_mode = 2;           // batched 128-tile launches (B >= 2) by pair128_chol_kernel (two tiles of a tile column per 512-thread workgroup):
                               // 0 never, 1 always, 2 whenever a launch holds two or more matrices (env GPG_PAIR)
  int pair_single_cols = 1 << 30;   // pair_mode 2: ONE matrix per launch goes to pair128_chol_kernel from this many padded columns on
  int task_order = -1;         // ticket order of the dataflow factorisation: 0 column-major, 1 critical path first, -1 measured choice (chol_task_order)
  double grad_eta = -1.0;      // >= 0: nugget the hyperparameter derivatives are formed with instead of hp->eta (gpg_set_gradient_nugget)
  int max_workgroups = 0;      // > 0: cap on the grid of every persistent launch (gpg_set_max_workgroups; 0 = co-resident capacity)
  std::map<const void*, int> occupancy;   // workgroups per compute unit of each persistent kernel (occupancy query, cached)
  size_t tile_flags_cap = 0;
  int nb_big = 0;       // wide-panel width used while at least big_rows columns remain (0: off)
  int big_rows = 0;
  // device buffers
  double* A = nullptr;       // [ld x Npad] column-major; lower triangle + RHS rows
  double* Xt = nullptr;      // [d x n]   (coordinate-major copy of x for coalesced loads)
  double* y = nullptr;       // [N]
  double* noise = nullptr;   // [N] known noise variances (zeros if none)
  double* dvec = nullptr;    // [Npad] diag(Kern) + noise / varK
  double* invp = nullptr;    // [Npad] 1/sqrt(dvec) (precon) or 1
  double* zvec = nullptr;    // [Npad] L^-T L^-1 P^-1 (y - V beta)  (= p * alpha)
  double* tmpv = nullptr;    // [Npad] scratch for the backward solve
  double* dinv = nullptr;    // [Npad] reciprocal pivots 1 / L_jj of the factor in A
  double* scal = nullptr;    // device scalars of the reductions
  int* info = nullptr;       // device: first failing pivot (0 = none)
  double* vec_rows = nullptr;   // [64 x vec_rows_cols] carrier tile of the single-vector backward solve (on first use)
  int vec_rows_cols = 0;        // = Npad of the full-gradient shape (allocation size)
  double* apply_buf = nullptr;  // [2 x Npad] vectors of gpg_factor_apply (on first use)
  double* vec_x = nullptr;      // [4 x vec_x_cols] compact solution rows of the vector solves (on first use)
  int vec_x_cols = 0;
  double* Wt = nullptr;      // prediction RHS rows [wt_rows x Npad]
  int wt_rows = 0;
  double* xq_dev = nullptr;  // [d x nxp]
  double* musig = nullptr;   // [2 x nxp]
  double* gradbuf = nullptr; // [2 x nxp x d] posterior-gradient reductions + [nxp x 64] backward-solve scratch
  int xq_cap = 0;
  double* batchA = nullptr;     // [batch_cap x A_elems] workspaces of the batched small-matrix factorisation (on first use)
  double* batchV = nullptr;     // [batch_cap x 3 x Npad] dvec / invp / dinv of each batched matrix
  int batch_cap = 0;
  double* batchW = nullptr;     // [gbatch_cap x Npad^2] L^-T of each matrix of a batched gradient call (on first use)
  double* batchM = nullptr;     // [gbatch_cap x Npad^2] -(L L^T)^-1, lower triangle
  double* batchZ = nullptr;     // [gbatch_cap x Npad]   p * alpha of each matrix
  int gbatch_cap = 0;
  double* gres = nullptr;       // [gres_cap x 2 GPG_GRAD_SLOTS_MAX] gradient sums per restart row
  int gres_cap = 0;
  int batch_max = -1;           // matrices per batched launch (-1: auto, 0 / 1: off)
  gpg_batch_item* items_dev = nullptr;   // [items_cap] per-row parameters of a batched call (device / pinned host)
  gpg_batch_item* items_host = nullptr;
  int items_cap = 0;
  double* dense_tmp = nullptr;  // [N x N] materialisation buffer (on request)
  double* Wfull = nullptr;      // [Npad x Npad] L^-T (likelihood gradient, on first use)
  double* Minv = nullptr;       // [Npad x Npad] -(L L^T)^-1, lower triangle
  double* Kbuf = nullptr;       // [Npad x Npad] copy of the assembled matrix (gpg_cond_fro with gradient, on first use)
  double* Tbuf = nullptr;       // [Npad x Npad] K^-2 (same)
  double* gpartial = nullptr;   // per-workgroup partial sums of the gradient contraction + 2 (d+3) results
  // pinned host staging
  double* h_scal = nullptr;
  double* h_pin = nullptr;           // pinned host scratch for small results (a copy into pageable memory costs ~20 us more)
  int* h_info = nullptr;
  // state
  bool have_data = false;
  bool factor_valid = false;   // the active workspace holds a finished factor
  bool prep_valid = false;     // dvec / invp of the active workspace belong to the hyperparameters of the last single-matrix call
  gpg_ws ws[2];                // parked copies of the workspace sets (the active one is stale here until it is parked)
  int ws_cur = 0;              // which set is active
  int last_factor_ws = 0;      // set of the most recent factorisation (gpg_factor_apply, gpg_get_matrix 3, gpg_dcov_quadform)
  bool eval_ready = false;     // set 1 holds factor + alpha of a successful gpg_setup_eval
  double eval_beta = 0.0;
  int last_precon = 0;         // wellcond of the matrix currently in A
  int scal_slots = 0;
  AsmParams eval_params;       // hyperparameters of the factor kept for gpg_predict
  // profiling
  unsigned prof_mask = 0;
  std::vector<ProfEvent> prof_pending;
  int prof_open = -1;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_pool;
  double prof_work[GPG_PROF_NCAT] = {0, 0, 0, 0, 0, 0};
  std::string err;
};

#define GPG_SCAL_COUNT 16

// ---- launchers (each enqueues on ctx->stream; none synchronises) ---------------------------------
void gpg_launch_prep(gpg_ctx* c, const AsmParams& p, double var_fval, double var_fgrad,
                     double s0, double t0, double s1, double t1);
void gpg_launch_assembly(gpg_ctx* c, const AsmParams& p);
void gpg_launch_prep_assembly_batch(gpg_ctx* c, const AsmParams& p, int B, const gpg_batch_item* items, size_t v_stride,
                                    size_t a_stride);
void gpg_launch_lkd_reduce_batch(gpg_ctx* c, int slot0, int B, size_t v_stride, size_t a_stride, const int* info0 /* B words */);
void gpg_launch_cross(gpg_ctx* c, const AsmParams& p, int nx, int nxp);   // Wt <- P^-1 Kyx (transposed)
void gpg_launch_abs_rowsum(gpg_ctx* c, double scale, double* out_dev);    // out[i] = scale * sum_j |M_ij|, M symmetric, lower triangle in A
void gpg_launch_tile_chol(gpg_ctx* c, int c0);                          // dataflow factorisation of A[c0:, c0:], 64-tiles
void gpg_launch_tile128_chol(gpg_ctx* c);
// dataflow W <- W L^-T / Z <- Z L^-1 for a few 64-row tiles; rows >= valid must be zero (their substitution is skipped);
// false: not applicable (caller falls back on the blocked sweep)
// triangular solves as dataflow launches?  (off with the blocked factor mode, and after one of them timed out: api.hip with_fallback)
static inline bool gpg_dataflow_solves(const gpg_ctx* c) { return c->solve_dataflow != 0 && (c->chol_impl != 0 || c->tail_cols != 0); }
bool gpg_overlap_inverse_begin(gpg_ctx* c, int B);          // eligible? then the next (batched) factorisation keeps its flags (call before enqueueing it)
bool gpg_overlap_inverse_trinv(gpg_ctx* c, int B, const double* Abase, size_t a_stride, const double* dinv_base, int d_stride, double* Wbase,
                               int* info_base);             // after the factorisation was enqueued: W = L^-T on the second stream
void gpg_overlap_inverse_cancel(gpg_ctx* c);                // the overlapped factorisation failed: make the W launch drain
bool gpg_overlap_inverse_wwt(gpg_ctx* c, int B, const double* Abase, size_t a_stride, const double* dinv_base, int d_stride, double* Wbase,
                             double* Mbase, int* info_base);   // Minv = -W W^T on the main stream, after W
bool gpg_launch_tile128_inverse(gpg_ctx* c, double* W, double* Minv);   // Minv <- -(L L^T)^-1 by two dataflow launches
bool gpg_launch_tile128_inverse_batch(gpg_ctx* c, int B, const double* Abase, size_t a_stride, const double* dinv_base, int d_stride,
                                      double* Wbase, double* Mbase, int* info_base);   // the same for B factors at once
void gpg_launch_frob_lower(gpg_ctx* c, const double* M, int ld, double* partial, double* out_dev);   // squared Frobenius norm (lower storage)
void gpg_launch_symmetrize(gpg_ctx* c, double* M, int ld);                                              // upper <- lower^T
bool gpg_launch_full_abt(gpg_ctx* c, const double* Sa, const double* Sb, double* M);                    // M (lower) = -Sa Sb^T, full operands
bool gpg_launch_rows_fwd(gpg_ctx* c, double* W, int ldw, int rows, int valid);
bool gpg_launch_rows_bwd(gpg_ctx* c, double* Z, int ldz, int rows, int valid);
void gpg_launch_tile_chol_batch(gpg_ctx* c, int B, double* Abase, size_t a_stride, double* dinv_base, int d_stride,
                                int* info_base);                              // dataflow factorisation, 128-tiles (whole matrix)
void gpg_cholesky(gpg_ctx* c);                                           // blocked right-looking, in place
void gpg_forward_rows(gpg_ctx* c, double* W, int ldw, int rows, int valid = -1);   // W <- W L^-T (rows >= valid are zero)
void gpg_launch_lkd_reduce(gpg_ctx* c, int slot);                        // writes scal[slot*8 ..]
void gpg_backward_solve(gpg_ctx* c);                                     // zvec <- L^-T (RHS row 0)
void gpg_launch_alpha(gpg_ctx* c, double* alpha_dev);                    // alpha = zvec * invp
void gpg_launch_predict_reduce(gpg_ctx* c, int nx, int nxp, double beta, double varK, int phase);
void gpg_backward_rows(gpg_ctx* c, double* Z, int ldz, int nrhs, double* tbuf);  // Z <- Z L^-1 (multi-RHS)
void gpg_launch_cross_grad(gpg_ctx* c, const AsmParams& p, int nx, int nxp, double* g1, double* g2);
void gpg_launch_hess_stage(gpg_ctx* c, const AsmParams& p, int nxp, double* h1, double* h2, double* T, int stage);
void gpg_launch_combine_rows(gpg_ctx* c, int slot);                       // RHS row 0 <- L^-1 P^-1 (y - V beta)
void gpg_launch_identity(gpg_ctx* c, double* W, int ldw);
void gpg_inverse_from_factor(gpg_ctx* c, double* W, double* Minv);       // Minv <- -(L L^T)^-1 (lower)
void gpg_launch_grad_contract(gpg_ctx* c, const AsmParams& p, double* partial, double* out_dev, const double* zvec,
                              const double* Minv);
void gpg_launch_unscale(gpg_ctx* c, const double* v, double* z);          // z = v / invp
int gpg_grad_partial_blocks(const gpg_ctx* c);
int gpg_factor_apply_dev(gpg_ctx* c, int op, double* v, double* out);       // (L L^T) v or (L L^T)^-1 v, device vectors [Npad]
void gpg_launch_extract(gpg_ctx* c, int which);                          // dense_tmp <- sym / P L
int gpg_kern_rtensor_run(int kernel, int d, int n1, int n2, int n1g, int n2g, int use_grad, const double* theta, double hp_kernel,
                         const double* rt_dev, const int* gpos1_dev, const int* gpos2_dev, double* out_dev, hipStream_t stream);
int gpg_kern_rtensor_dhp_run(int kernel, int d, int n, int use_grad, const double* theta, double hp_kernel, const double* rt_dev,
                             double* out_th_dev, double* out_al_dev, hipStream_t stream);
int gpg_kern_rtensor_hess_x_run(int kernel, int d, int n1, int n2, int n2g, int use_grad, const double* theta, double hp_kernel,
                                const double* rt_dev, const int* gpos2_dev, double* out_dev, hipStream_t stream);
int gpg_ws_activate(gpg_ctx* c, int which);                              // make workspace set `which` the active one (allocates set 1 on first use)

// Device allocation inside a launch helper (task lists, flags, carrier tiles): on failure the pointer stays null, the
// context is marked and the helper returns WITHOUT launching; the API call reports it (GPG_LAUNCH_OK) instead of a
// kernel dereferencing a null pointer.
template <typename T>
inline bool gpg_dev_alloc(gpg_ctx* c, T** p, size_t bytes) {
  *p = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(p), bytes) == hipSuccess && *p != nullptr) return true;
  (void)hipGetLastError();
  *p = nullptr;
  c->launch_error = true;
  return false;
}

// profiling helpers
void gpg_prof_begin(gpg_ctx* c, int cat, double work);
void gpg_prof_end(gpg_ctx* c);

#define GPG_LAUNCH_OK(c)                                                                 \
  do {                                                                                   \
    if ((c)->launch_error) {                                                             \
      (c)->launch_error = false;                                                         \
      (c)->factor_valid = (c)->eval_ready = false;                                       \
      (c)->err = "out of device memory inside a launch helper (task list / flags / carrier tile)"; \
      return -2;                                                                         \
    }                                                                                    \
  } while (0)

#define GPG_HIP_OK(c, call)                                                              \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess) {                                                              \
      (c)->err = std::string(#call) + ": " + hipGetErrorString(e_);                      \
      return -2;                                                                         \
    }                                                                                    \
  } while (0)
