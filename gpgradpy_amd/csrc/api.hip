// C ABI of libgpgrad_hip.so (see include/gpgrad.h for the contract and the reference citations).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <cstring>
#include <functional>
#include <limits>
#include "gpg_internal.h"

static thread_local std::string g_create_err;

// ---- profiling helpers ---------------------------------------------------------------------------
void gpg_prof_begin(gpg_ctx* c, int cat, double work) {
  c->prof_open = -1;
  if (!(c->prof_mask & (1u << cat))) return;
  std::pair<hipEvent_t, hipEvent_t> ev;
  if (!c->prof_pool.empty()) {
    ev = c->prof_pool.back();
    c->prof_pool.pop_back();
  } else {
    (void)hipEventCreate(&ev.first);
    (void)hipEventCreate(&ev.second);
  }
  (void)hipEventRecord(ev.first, c->stream);
  c->prof_pending.push_back(ProfEvent{ev.first, ev.second, cat});
  c->prof_open = (int)c->prof_pending.size() - 1;
  c->prof_work[cat] += work;
}

void gpg_prof_end(gpg_ctx* c) {
  if (c->prof_open < 0) return;
  (void)hipEventRecord(c->prof_pending[c->prof_open].e1, c->stream);
  c->prof_open = -1;
}

static AsmParams make_params(const gpg_ctx* c, const gpg_hp* hp, int mode) {
  AsmParams p;
  memset(&p, 0, sizeof(p));
  p.n = c->n; p.d = c->d; p.use_grad = c->use_grad; p.kernel = c->kernel;
  p.ng = c->ng > 0 ? c->ng : 1; p.gpos = c->gpos;
  p.N = c->N; p.Npad = c->Npad; p.ld = c->ld;
  p.precon = hp->wellcond == GPG_WELLCOND_PRECON ? 1 : 0;
  p.mode = mode;
  p.varK = hp->varK_mat;
  p.eta = hp->eta;
  for (int k = 0; k < c->d; ++k) p.theta[k] = hp->theta[k];
  p.hp_kernel = hp->hp_kernel;
  return p;
}

// ---- workspace sets ---------------------------------------------------------------------------------
static void ws_park(gpg_ctx* c) {
  gpg_ws& w = c->ws[c->ws_cur];
  w.A = c->A; w.dvec = c->dvec; w.invp = c->invp; w.dinv = c->dinv; w.zvec = c->zvec; w.tmpv = c->tmpv;
  w.factor_valid = c->factor_valid; w.prep_valid = c->prep_valid; w.precon = c->last_precon;
}

int gpg_ws_activate(gpg_ctx* c, int which) {
  if (which == c->ws_cur) return 0;
  ws_park(c);
  gpg_ws& w = c->ws[which];
  if (!w.A) {   // first gpg_setup_eval: a second workspace of the context's (full-gradient) shape
    const size_t nv = sizeof(double) * (size_t)c->vec_rows_cols;
    double** ptrs[] = {&w.dvec, &w.invp, &w.dinv, &w.zvec, &w.tmpv};
    bool ok = hipMalloc(&w.A, sizeof(double) * c->A_elems) == hipSuccess;
    for (double** q : ptrs) ok = ok && hipMalloc(q, nv) == hipSuccess;
    ok = ok && hipMemsetAsync(w.A, 0, sizeof(double) * c->A_elems, c->stream) == hipSuccess;
    if (!ok) {
      (void)hipGetLastError();
      if (w.A) (void)hipFree(w.A);
      for (double** q : ptrs) if (*q) { (void)hipFree(*q); *q = nullptr; }
      w = gpg_ws();
      c->err = "out of device memory for the posterior's own factor workspace";
      return -2;
    }
  }
  c->A = w.A; c->dvec = w.dvec; c->invp = w.invp; c->dinv = w.dinv; c->zvec = w.zvec; c->tmpv = w.tmpv;
  c->factor_valid = w.factor_valid; c->prep_valid = w.prep_valid; c->last_precon = w.precon;
  c->ws_cur = which;
  return 0;
}
#define GPG_WS(c, which) do { if (gpg_ws_activate((c), (which)) != 0) return -2; } while (0)

static int check_hp(gpg_ctx* c, const gpg_hp* hp) {
  if (!c) return -1;
  if (!hp || !hp->theta) { c->err = "hp / hp->theta is NULL"; return -1; }
  if (!c->have_data) { c->err = "gpg_set_data has not been called"; return -1; }
  for (int k = 0; k < c->d; ++k)
    if (std::isnan(hp->theta[k])) { c->err = "There are nan values in theta"; return -1; }  // Kernel.py:201
  if (!(hp->varK_mat > 0.0)) { c->err = "varK must be positive"; return -1; }               // Kernel.py:199
  if (c->kernel == GPG_KERNEL_RATQU && !(hp->hp_kernel > 0.0)) { c->err = "hp_kernel (alpha of RatQu) must be positive"; return -1; }
  if (hp->wellcond == GPG_WELLCOND_PRECON && !c->use_grad) {
    c->err = "wellcond 'precon' requires use_grad (Kernel.py:222)";
    return -1;
  }
  return 0;
}

// Streams and pinned host blocks are the expensive part of creating / destroying a context (~1 ms each on this runtime; the device
// allocations are ~0.1 ms each), and a Bayesian-optimisation loop makes a new context for every added point: a destroyed context
// parks them here (per device, a handful at most) and the next gpg_create takes them over.
struct ParkedHost { int device; hipStream_t stream, stream_upd, stream_inv; double* h_scal; int* h_info; double* h_pin; };
static std::mutex g_park_mutex;
static std::vector<ParkedHost> g_parked;
constexpr int kScalSlotsDefault = 64;
constexpr size_t kParkMax = 4;

// Pinned host scratch of n doubles, or nullptr (too large / no memory): the caller then copies into its own pageable buffer.
constexpr size_t kPinDoubles = 16384;
static double* pinned_scratch(gpg_ctx* c, size_t n) {
  if (n > kPinDoubles) return nullptr;
  if (!c->h_pin && hipHostMalloc(&c->h_pin, sizeof(double) * kPinDoubles) != hipSuccess) { (void)hipGetLastError(); c->h_pin = nullptr; }
  return c->h_pin;
}

static int ensure_scal(gpg_ctx* c, int slots) {
  if (slots <= c->scal_slots) return 0;
  if (c->scal) (void)hipFree(c->scal);
  if (c->info) (void)hipFree(c->info);
  const bool keep_host = c->scal_slots == 0 && slots == kScalSlotsDefault && c->h_scal && c->h_info;   // taken over from a parked context
  if (!keep_host) {
    if (c->h_scal) (void)hipHostFree(c->h_scal);
    if (c->h_info) (void)hipHostFree(c->h_info);
    c->h_scal = nullptr; c->h_info = nullptr;
  }
  c->scal = nullptr; c->info = nullptr; c->scal_slots = 0;
  GPG_HIP_OK(c, hipMalloc(&c->scal, sizeof(double) * 8 * slots));
  GPG_HIP_OK(c, hipMalloc(&c->info, sizeof(int) * slots));
  if (!keep_host) {
    GPG_HIP_OK(c, hipHostMalloc(&c->h_scal, sizeof(double) * 8 * slots));
    GPG_HIP_OK(c, hipHostMalloc(&c->h_info, sizeof(int) * slots));
  }
  c->scal_slots = slots;
  return 0;
}

// queue one likelihood evaluation; results land in scal[slot], info[slot]
static void enqueue_lkd(gpg_ctx* c, const gpg_hp* hp, int slot) {
  AsmParams p = make_params(c, hp, 0);
  c->alpha_valid = false;
  c->last_precon = p.precon;
  c->prep_valid = true;
  c->last_factor_ws = c->ws_cur;
  int* info_save = c->info;
  c->info = info_save + slot;   // potrf writes through c->info
  gpg_launch_prep(c, p, hp->var_fval, hp->var_fgrad, 1.0, 0.0, 0.0, 1.0);
  gpg_launch_assembly(c, p);
  gpg_cholesky(c);
  gpg_launch_lkd_reduce(c, slot);
  c->info = info_save;
}

static void finish_lkd(const gpg_ctx* c, const gpg_hp* hp, const double* s, int info, gpg_lkd_out* out) {
  const double nan = std::numeric_limits<double>::quiet_NaN();
  out->info = info;
  out->pad_ = 0;
  if (info != 0) {
    out->ln_lkd = out->ln_det = out->beta = out->varK = out->rKr = nan;
    return;
  }
  const double N = (double)c->N;
  out->ln_det = s[0];
  out->beta = s[1];
  out->rKr = s[2];
  if (hp->closed_form_varK) {
    double vk = s[2] / N;
    if (!(vk > 1e-32)) vk = 1e-32;                        // CalcLkd.py:151,159
    out->varK = vk;
    out->ln_lkd = -(N * log(vk) + s[0]) / 2.0;            // CalcLkd.py:168 (penalty off by default)
  } else {
    out->varK = hp->varK_mat;
    out->ln_lkd = -(s[0] + s[2]) / 2.0;                   // CalcLkd.py:226
  }
}

extern "C" {

int gpg_create(gpg_ctx** out, int device, int n_eval, int dim, int use_grad, int kernel) {
  if (!out) return -1;
  *out = nullptr;
  if (n_eval < 1 || dim < 1 || dim > GPG_MAX_DIM) { g_create_err = "n_eval >= 1 and 1 <= dim <= 16 required"; return -1; }
  if (kernel != GPG_KERNEL_SQEXP && kernel != GPG_KERNEL_MA5F2 && kernel != GPG_KERNEL_RATQU) { g_create_err = "unknown kernel id"; return -1; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { g_create_err = "no HIP device visible"; return -2; }
  if (device < 0 || device >= ndev) { g_create_err = "device index out of range"; return -1; }
  gpg_ctx* c = new gpg_ctx();
  c->device = device; c->n = n_eval; c->d = dim; c->use_grad = use_grad ? 1 : 0; c->kernel = kernel;
  c->N = use_grad ? n_eval * (dim + 1) : n_eval;
  c->Npad = ((c->N + GPG_TILE - 1) / GPG_TILE) * GPG_TILE;
  c->R = GPG_RHS_ROWS;
  c->ld = c->Npad + c->R;
  c->nb_outer = c->Npad >= 8192 ? 512 : 256;   // wider panels amortise the C-tile traffic of the trailing update (measured)
  // factorisation schedule (measured, profiles/r01_tile_probe.log): one dataflow launch, 64-tile kernel for small
  // matrices (shorter dependency chain), 128-tile kernel above; chol_impl 0 = blocked right-looking (A/B runs)
  c->chol_impl = 1;
  if (const char* e = getenv("GPG_OVERLAP_INVERSE")) c->overlap_inverse = atoi(e);   // diagnostic override (A/B runs)
  if (const char* e = getenv("GPG_PAIR")) c->pair_mode = atoi(e);                    // diagnostic override (A/B runs)
  if (const char* e = getenv("GPG_ROWS_MAX_TASKS")) { const int v = atoi(e); if (v >= 64) c->rows_max_tasks = v; }   // diagnostic override
  c->tail_cols = 12288;
#define CREATE_OK(call)                                                              \
  do {                                                                               \
    hipError_t e_ = (call);                                                          \
    if (e_ != hipSuccess) {                                                          \
      g_create_err = std::string(#call) + ": " + hipGetErrorString(e_);              \
      gpg_destroy(c);                                                                \
      return -2;                                                                     \
    }                                                                                \
  } while (0)
  CREATE_OK(hipSetDevice(device));
  CREATE_OK(hipDeviceGetAttribute(&c->num_cus, hipDeviceAttributeMultiprocessorCount, device));
  {
    std::lock_guard<std::mutex> lock(g_park_mutex);
    for (size_t k = 0; k < g_parked.size(); ++k)
      if (g_parked[k].device == device) {
        c->stream = g_parked[k].stream; c->stream_upd = g_parked[k].stream_upd; c->stream_inv = g_parked[k].stream_inv;
        c->h_scal = g_parked[k].h_scal; c->h_info = g_parked[k].h_info; c->h_pin = g_parked[k].h_pin;
        g_parked.erase(g_parked.begin() + k);
        break;
      }
  }
  if (!c->stream) {
    int prio_least = 0, prio_greatest = 0;
    CREATE_OK(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));   // numerically lowest = highest priority
    // main stream in the middle of the range (the least priority when the range has two levels only), the look-ahead stream of the
    // blocked schedule above it, the overlapped inverse -- which only ever WAITS for work of the main stream -- below or level with it
    const int prio_main = prio_least - (prio_least - prio_greatest) / 2;
    CREATE_OK(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_main));
    CREATE_OK(hipStreamCreateWithPriority(&c->stream_upd, hipStreamNonBlocking, prio_greatest));
    CREATE_OK(hipStreamCreateWithPriority(&c->stream_inv, hipStreamNonBlocking, prio_least));
  }
  c->ng = c->use_grad ? n_eval : 0;
  c->A_elems = (size_t)c->ld * c->Npad;
  c->vec_rows_cols = c->Npad;
  CREATE_OK(hipMalloc(&c->A, sizeof(double) * c->A_elems));
  CREATE_OK(hipMalloc(&c->gpos, sizeof(int) * c->n));
  {
    std::vector<int> ident(c->n);
    for (int a = 0; a < c->n; ++a) ident[a] = c->use_grad ? a : -1;
    CREATE_OK(hipMemcpy(c->gpos, ident.data(), sizeof(int) * c->n, hipMemcpyHostToDevice));
  }
  CREATE_OK(hipMalloc(&c->Xt, sizeof(double) * (size_t)c->n * c->d));
  CREATE_OK(hipMalloc(&c->y, sizeof(double) * c->N));
  CREATE_OK(hipMalloc(&c->noise, sizeof(double) * c->N));
  CREATE_OK(hipMalloc(&c->dvec, sizeof(double) * c->Npad));
  CREATE_OK(hipMalloc(&c->invp, sizeof(double) * c->Npad));
  CREATE_OK(hipMalloc(&c->zvec, sizeof(double) * c->Npad));
  CREATE_OK(hipMalloc(&c->tmpv, sizeof(double) * c->Npad));
  CREATE_OK(hipMalloc(&c->dinv, sizeof(double) * c->Npad));
  CREATE_OK(hipMemsetAsync(c->A, 0, sizeof(double) * (size_t)c->ld * c->Npad, c->stream));   // ordered on the context's own
  CREATE_OK(hipStreamSynchronize(c->stream));                                                  // (non-blocking) stream
#undef CREATE_OK
  if (ensure_scal(c, kScalSlotsDefault) != 0) { g_create_err = c->err; gpg_destroy(c); return -2; }
  *out = c;
  return 0;
}

void gpg_destroy(gpg_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->stream_upd) (void)hipStreamSynchronize(c->stream_upd);
  if (c->stream_inv) (void)hipStreamSynchronize(c->stream_inv);
  for (auto& pe : c->prof_pending) { (void)hipEventDestroy(pe.e0); (void)hipEventDestroy(pe.e1); }
  for (auto& ev : c->prof_pool) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  ws_park(c);
  double* bufs[] = {c->ws[0].A, c->ws[0].dvec, c->ws[0].invp, c->ws[0].zvec, c->ws[0].tmpv, c->ws[0].dinv,
                    c->ws[1].A, c->ws[1].dvec, c->ws[1].invp, c->ws[1].zvec, c->ws[1].tmpv, c->ws[1].dinv,
                    c->Xt, c->y, c->noise, c->scal, c->Wt, c->xq_dev,
                    c->musig, c->gradbuf, c->dense_tmp, c->Wfull, c->Minv, c->gpartial, c->batchA, c->batchV, c->vec_rows, c->vec_x, c->apply_buf,
                    c->batchW, c->batchM, c->batchZ, c->gres, c->Kbuf, c->Tbuf};
  for (double* b : bufs) if (b) (void)hipFree(b);
  if (c->info) (void)hipFree(c->info);
  if (c->gpos) (void)hipFree(c->gpos);
  bool parked = false;
  if (c->stream && c->stream_upd && c->stream_inv && c->h_scal && c->h_info && c->scal_slots == kScalSlotsDefault) {
    std::lock_guard<std::mutex> lock(g_park_mutex);
    if (g_parked.size() < kParkMax) {
      g_parked.push_back(ParkedHost{c->device, c->stream, c->stream_upd, c->stream_inv, c->h_scal, c->h_info, c->h_pin});   // (the streams are idle: synchronised above)
      parked = true;
    }
  }
  if (!parked) {
    if (c->h_scal) (void)hipHostFree(c->h_scal);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    if (c->h_info) (void)hipHostFree(c->h_info);
  }
  if (c->items_dev) (void)hipFree(c->items_dev);
  if (c->items_host) (void)hipHostFree(c->items_host);
  for (auto& kv : c->tilemaps) if (kv.second.dev) (void)hipFree(kv.second.dev);
  if (c->ev_flags) (void)hipEventDestroy(c->ev_flags);
  if (c->ev_trinv) (void)hipEventDestroy(c->ev_trinv);
  if (c->ev_winit) (void)hipEventDestroy(c->ev_winit);
  if (c->keep_flags) (void)hipFree(c->keep_flags);
  for (auto e : c->ev_panel) (void)hipEventDestroy(e);
  for (auto e : c->ev_upd) (void)hipEventDestroy(e);
  if (!parked) {
    if (c->stream_upd) (void)hipStreamDestroy(c->stream_upd);
    if (c->stream_inv) (void)hipStreamDestroy(c->stream_inv);
    if (c->stream) (void)hipStreamDestroy(c->stream);
  }
  delete c;
}

const char* gpg_last_error(const gpg_ctx* c) { return c ? c->err.c_str() : g_create_err.c_str(); }

int gpg_set_grad_mask(gpg_ctx* c, const unsigned char* use_grad_pt) {
  if (!c) return -1;
  if (!c->use_grad) { c->err = "gradient mask given but use_grad = 0 (GaussianProcess.py:257)"; return -1; }
  GPG_HIP_OK(c, hipSetDevice(c->device));
  GPG_HIP_OK(c, hipStreamSynchronize(c->stream));
  std::vector<int> gp(c->n);
  int ng = 0;
  for (int a = 0; a < c->n; ++a) gp[a] = (!use_grad_pt || use_grad_pt[a]) ? ng++ : -1;
  GPG_HIP_OK(c, hipMemcpy(c->gpos, gp.data(), sizeof(int) * c->n, hipMemcpyHostToDevice));
  c->ng = ng;
  c->N = c->n + ng * c->d;                                       // GaussianProcess.py:260-262
  c->Npad = ((c->N + GPG_TILE - 1) / GPG_TILE) * GPG_TILE;
  c->ld = c->Npad + c->R;                                        // never exceeds the allocation made for all gradients
  c->have_data = false;
  c->alpha_valid = false;
  c->factor_valid = c->eval_ready = c->prep_valid = false;
  c->ws[0].factor_valid = c->ws[1].factor_valid = c->ws[0].prep_valid = c->ws[1].prep_valid = false;
  if (c->dense_tmp) { (void)hipFree(c->dense_tmp); c->dense_tmp = nullptr; }
  if (c->batchA) { (void)hipFree(c->batchA); (void)hipFree(c->batchV); c->batchA = c->batchV = nullptr; c->batch_cap = 0; }
  if (c->Wfull) { (void)hipFree(c->Wfull); (void)hipFree(c->Minv); c->Wfull = c->Minv = nullptr; }
  if (c->Kbuf) { (void)hipFree(c->Kbuf); (void)hipFree(c->Tbuf); c->Kbuf = c->Tbuf = nullptr; }
  if (c->batchW) { (void)hipFree(c->batchW); (void)hipFree(c->batchM); (void)hipFree(c->batchZ); c->batchW = c->batchM = c->batchZ = nullptr; c->gbatch_cap = 0; }
  if (c->Wt) { (void)hipFree(c->Wt); (void)hipFree(c->xq_dev); (void)hipFree(c->musig); (void)hipFree(c->gradbuf); c->Wt = c->xq_dev = c->musig = c->gradbuf = nullptr; c->xq_cap = 0; }
  return 0;
}

int gpg_set_data(gpg_ctx* c, const double* x, const double* data_vec, const double* noise_var) {
  if (!c) return -1;
  if (!x || !data_vec) { c->err = "x / data_vec is NULL"; return -1; }
  GPG_HIP_OK(c, hipSetDevice(c->device));
  std::vector<double> xt((size_t)c->n * c->d);
  for (int a = 0; a < c->n; ++a)
    for (int k = 0; k < c->d; ++k) xt[(size_t)k * c->n + a] = x[(size_t)a * c->d + k];
  GPG_HIP_OK(c, hipStreamSynchronize(c->stream));
  GPG_HIP_OK(c, hipMemcpy(c->Xt, xt.data(), sizeof(double) * xt.size(), hipMemcpyHostToDevice));
  GPG_HIP_OK(c, hipMemcpy(c->y, data_vec, sizeof(double) * c->N, hipMemcpyHostToDevice));
  if (noise_var) GPG_HIP_OK(c, hipMemcpy(c->noise, noise_var, sizeof(double) * c->N, hipMemcpyHostToDevice));
  else {   // on the context's stream (created non-blocking: the null stream is not ordered against it), then waited for
    GPG_HIP_OK(c, hipMemsetAsync(c->noise, 0, sizeof(double) * c->N, c->stream));
    GPG_HIP_OK(c, hipStreamSynchronize(c->stream));
  }
  c->have_data = true;
  c->alpha_valid = false;
  c->factor_valid = c->eval_ready = c->prep_valid = false;
  c->ws[0].factor_valid = c->ws[1].factor_valid = c->ws[0].prep_valid = c->ws[1].prep_valid = false;
  return 0;
}

// A dependency wait of the dataflow factorisation timed out (never expected: it would mean a lost flag or a
// dispatch-order assumption broken).  The kernel drains and marks info with GPG_INFO_INTERNAL.
int gpg_set_noise(gpg_ctx* c, const double* noise_var) {
  if (!c) return -1;
  if (!c->have_data) { c->err = "gpg_set_data has not been called"; return -1; }
  GPG_HIP_OK(c, hipSetDevice(c->device));
  GPG_HIP_OK(c, hipStreamSynchronize(c->stream));
  if (noise_var) GPG_HIP_OK(c, hipMemcpy(c->noise, noise_var, sizeof(double) * c->N, hipMemcpyHostToDevice));
  else {
    GPG_HIP_OK(c, hipMemsetAsync(c->noise, 0, sizeof(double) * c->N, c->stream));
    GPG_HIP_OK(c, hipStreamSynchronize(c->stream));
  }
  c->ws[0].factor_valid = c->ws[0].prep_valid = false;      // likelihood-side state only: the posterior keeps its own factor
  if (c->ws_cur == 0) c->factor_valid = c->prep_valid = false;
  c->alpha_valid = false;
  return 0;
}

static int internal_failure(gpg_ctx* c, const int* infos, int m) {
  for (int i = 0; i < m; ++i)
    if (infos[i] == GPG_INFO_INTERNAL) {
      c->err = "dataflow Cholesky: a dependency wait timed out (another dataflow launch on this device?)";
      c->fail_kind = 1;
      c->factor_valid = false;
      if (c->ws_cur == 1) c->eval_ready = false;
      return -4;
    }
  return 0;
}

// The dataflow triangular solves of the posterior have the same bounded waits as the factorisation; the factor in A
// is untouched by them, so the evaluation is simply repeated with the blocked sweeps.
static int solve_failure(gpg_ctx* c) {
  if (*c->h_info != GPG_INFO_INTERNAL) return 0;
  c->err = "dataflow triangular solve: a dependency wait timed out (another dataflow launch on this device?)";
  c->fail_kind = 2;
  return -4;
}

static int gpg_lkd_once(gpg_ctx* c, const gpg_hp* hp, gpg_lkd_out* out) {
  int rc = check_hp(c, hp);
  if (rc) return rc;
  if (!out) { c->err = "out is NULL"; return -1; }
  GPG_HIP_OK(c, hipSetDevice(c->device));
  GPG_WS(c, 0);
  c->zero_info_in_prep = true;                             // one launch fewer than a memset of the info word
  enqueue_lkd(c, hp, 0);
  GPG_HIP_OK(c, hipMemcpyAsync(c->h_scal, c->scal, sizeof(double) * 8, hipMemcpyDeviceToHost, c->stream));
  GPG_HIP_OK(c, hipStreamSynchronize(c->stream));
  GPG_HIP_OK(c, hipGetLastError());
  GPG_LAUNCH_OK(c);
  c->h_info[0] = (int)c->h_scal[7];                        // lkd_reduce_kernel passes the info word along
  if (internal_failure(c, c->h_info, 1)) return -4;
  c->factor_valid = (c->h_info[0] == 0);
  finish_lkd(c, hp, c->h_scal, c->h_info[0], out);
  return out->info;
}

static int gpg_lkd_grad_once(gpg_ctx* c, const gpg_hp* hp, gpg_lkd_out* out, double* g_aa, double* g_inv) {
  int rc = check_hp(c, hp);
  if (rc) return rc;
  if (!out || !g_aa || !g_inv) { c->err = "out / g_aa / g_inv is NULL"; return -1; }
  GPG_HIP_OK(c, hipSetDevice(c->device));
  GPG_WS(c, 0);
  const size_t nn = (size_t)c->Npad * c->Npad;
  if (!c->Wfull) {
    GPG_HIP_OK(c, hipMalloc(&c->Wfull, sizeof(double) * nn));
    GPG_HIP_OK(c, hipMalloc(&c->Minv, sizeof(double) * nn));
  }
  const int nblk = gpg_grad_partial_blocks(c);
  const int nval = 2 * GPG_GRAD_SLOTS_MAX;
  if (!c->gpartial) GPG_HIP_OK(c, hipMalloc(&c->gpartial, sizeof(double) * (size_t)nval * (nblk + 1)));
  c->zero_info_in_prep = true;
  // small matrix on the dataflow schedule: W = L^-T goes to the second stream right behind the factorisation (cholesky_dataflow.hip)
  const bool overlap = c->Npad <= 32768 && gpg_overlap_inverse_begin(c, 1);
  enqueue_lkd(c, hp, 0);                                   // factor + beta + r'K^-1 r + ln det (scal slot 0)
  c->chol_flags_override = nullptr;                        // (consumed by the 64-tile launch; cleared in case another schedule ran)
  if (overlap && !gpg_overlap_inverse_trinv(c, 1, c->A, 0, c->dinv, 0, c->Wfull, c->info)) { c->err = "overlapped inverse: launch refused"; return -2; }
  GPG_HIP_OK(c, hipMemcpyAsync(c->h_scal, c->scal, sizeof(double) * 8, hipMemcpyDeviceToHost, c->stream));
  // Large matrices: look at the factorisation's info before spending two more N^3/3 sweeps on a failed factor.  Small ones
  // (the sweeps cost less than a host round trip is worth): everything is enqueued at once and judged at the end; the kernels
  // run to completion on whatever a failed factorisation left behind (their waits depend on flags, not on values).
  const bool one_sync = c->Npad <= 2048 || (overlap && c->Npad <= 12288);
  if (!one_sync) {
    GPG_HIP_OK(c, hipStreamSynchronize(c->stream));
    GPG_HIP_OK(c, hipGetLastError());
    GPG_LAUNCH_OK(c);
    c->h_info[0] = (int)c->h_scal[7];
    if (c->h_info[0] != 0 && overlap) gpg_overlap_inverse_cancel(c);   // W = L^-T is already running behind the factorisation: make it drain
    if (internal_failure(c, c->h_info, 1)) return -4;
    c->factor_valid = (c->h_info[0] == 0);
    finish_lkd(c, hp, c->h_scal, c->h_info[0], out);
    if (out->info != 0) return out->info;
  }
  // alpha = Kcov^-1 (y - V beta):  z = L^-T (w2 - beta w1) = p * alpha
  AsmParams p = make_params(c, hp, 0);
  gpg_launch_combine_rows(c, 0);
  gpg_backward_solve(c);
  if (overlap) { if (!gpg_overlap_inverse_wwt(c, 1, c->A, 0, c->dinv, 0, c->Wfull, c->Minv, c->info)) { c->err = "overlapped inverse: launch refused"; return -2; } }
  else gpg_inverse_from_factor(c, c->Wfull, c->Minv);
  double* res = c->gpartial + (size_t)nval * nblk;
  gpg_launch_grad_contract(c, p, c->gpartial, res, c->zvec, c->Minv);
  std::vector<double> hvec;
  double* h = pinned_scratch(c, nval);
  if (!h) { hvec.resize(nval); h = hvec.data(); }
  const int ns = c->d + 3 + (c->kernel == GPG_KERNEL_RATQU ? 1 : 0);
  GPG_HIP_OK(c, hipMemcpyAsync(h, res, sizeof(double) * 2 * ns, hipMemcpyDeviceToHost, c->stream));
  GPG_HIP_OK(c, hipMemcpyAsync(c->h_info, c->info, sizeof(int), hipMemcpyDeviceToHost, c->stream));   // the dataflow solve
  GPG_HIP_OK(c, hipStreamSynchronize(c->stream));                                                      // reports through info
  GPG_HIP_OK(c, hipGetLastError());
  GPG_LAUNCH_OK(c);
  if (one_sync) {
    const int info_solves = c->h_info[0];
    c->h_info[0] = (int)c->h_scal[7];                      // the word as the factorisation left it
    if (internal_failure(c, c->h_info, 1)) return -4;
    c->factor_valid = (c->h_info[0] == 0);
    finish_lkd(c, hp, c->h_scal, c->h_info[0], out);
    if (out->info != 0) return out->info;
    c->h_info[0] = info_solves;
  }
  if (solve_failure(c)) return -4;
  g_aa[c->d + 3] = g_inv[c->d + 3] = 0.0;                  // hp_kernel slot: RatQu only
  for (int k = 0; k < ns; ++k) { g_aa[k] = h[k]; g_inv[k] = h[ns + k]; }
  c->alpha_valid = true;
  return 0;
}

// How many restart rows of the context's shape one dataflow launch factorises (gpg_set_batch), and the workspaces
// for them.  Small matrices (64-tile regime): one factorisation is latency-bound and leaves most of the chip idle;
// large ones (128-tile regime) do so at their two ends.  Workspaces are sized once for the largest batch the shape
// will use, not for this call's m: a later, larger call must not pay a multi-GB reallocation.
static int batch_plan(gpg_ctx* c, int m) {
  const bool small = c->tail_cols > 0 && c->Npad <= c->tail_cols;
  const bool large_df = !small && c->chol_impl == 1;
  int bmax = c->batch_max;
  if (bmax < 0) {
    if (c->Npad < 2048) {   // 64-tile kernel: enough matrices to put ~8k tiles in flight
      const long mt = c->Npad / 64, ntask = mt * (mt + 5) / 2;
      bmax = (int)(8192 / (ntask > 0 ? ntask : 1));
    } else if (c->Npad <= 32768) {
      // ~1280 tasks of the 128-tile kernel per tile-column round (measured, tools/tile_probe 5: 2560 columns 32 / 37 /
      // 41 TF with 16 / 32 / 64 matrices, 4608: 42 / 51 / 52 with 8 / 32 / 64, 9216: 59 / 60 with 8 / 16, 18048: +0.3 %
      // from 8 to 16); none at 68k columns
      // (r03, pair kernel with the cheaper finalisation: twice as many still pay -- 2560 columns 48.5 -> 51.6 TF from 64 to 128
      // matrices, 9216: 64.9 -> 65.6 from 16 to 32, 18048: 68.6 -> 69.1 from 10 to 20 -- the two ends of a launch are shared by more work)
      bmax = (2816 + c->Npad / 128 - 1) / (c->Npad / 128);
    } else {
      // very large matrices (cfg5: 68096 padded columns, 37 GB per workspace): the chain-bound ends are ~3 % of one
      // factorisation there and several matrices per launch gain nothing (measured: 65.7 TF with three per launch against
      // 66.0 one at a time, profiles/r02b_bench_cfg5_3perlaunch.json); gpg_set_batch(2..3) still works (288 GB hold them)
      bmax = 1;
    }
    if (bmax > 1) bmax = bmax < 8 ? 8 : (bmax > 128 ? 128 : bmax);
  }
  if (c->batch_max > 1 && c->Npad > 32768 && bmax > 3) bmax = 3;
  if (!(small || large_df) || bmax <= 1 || m <= 1) return 1;
  const size_t bytesA = sizeof(double) * c->A_elems;
  int Bcap = bmax;
  while (Bcap > 1 && bytesA * Bcap > ((size_t)120 << 30)) --Bcap;  // at most 120 GB of extra workspaces (288 GB per GPU)
  if (Bcap > c->batch_cap) {
    if (c->batchA) (void)hipFree(c->batchA);
    if (c->batchV) (void)hipFree(c->batchV);
    c->batchA = c->batchV = nullptr; c->batch_cap = 0;
    while (Bcap > 1) {
      if (hipMalloc(&c->batchA, bytesA * Bcap) == hipSuccess && hipMalloc(&c->batchV, sizeof(double) * 3 * c->Npad * Bcap) == hipSuccess) {
        c->batch_cap = Bcap;
        break;
      }
      if (c->batchA) (void)hipFree(c->batchA);
      c->batchA = c->batchV = nullptr;
      (void)hipGetLastError();
      Bcap /= 2;                                                   // no room: try half, then one matrix at a time
    }
  }
  const int B = m < c->batch_cap ? m : c->batch_cap;
  return B < 1 ? 1 : B;
}

static int gpg_lkd_batch_once(gpg_ctx* c, int m, const double* hp_rows, int row_len, double eta, int wellcond,
                              int closed_form_varK, gpg_lkd_out* out) {
  if (!c) return -1;
  if (m < 1 || !hp_rows || !out || row_len < c->d + 3) { c->err = "bad batch arguments (row_len >= d + 3)"; return -1; }
  GPG_HIP_OK(c, hipSetDevice(c->device));
  GPG_WS(c, 0);
  c->alpha_valid = false;                                  // the vectors of set 0 are redirected / rewritten by the groups
  int rc = ensure_scal(c, m);
  if (rc) return rc;
  if (m > c->items_cap) {
    if (c->items_dev) (void)hipFree(c->items_dev);
    if (c->items_host) (void)hipHostFree(c->items_host);
    c->items_dev = c->items_host = nullptr; c->items_cap = 0;
    GPG_HIP_OK(c, hipMalloc(&c->items_dev, sizeof(gpg_batch_item) * m));
    GPG_HIP_OK(c, hipHostMalloc(&c->items_host, sizeof(gpg_batch_item) * m));
    c->items_cap = m;
  }
  std::vector<gpg_hp> hps(m);
  for (int i = 0; i < m; ++i) {
    const double* row = hp_rows + (size_t)i * row_len;
    hps[i].theta = row;
    hps[i].varK_mat = row[c->d];
    hps[i].var_fval = row[c->d + 1];
    hps[i].var_fgrad = row[c->d + 2];
    hps[i].eta = row_len > c->d + 4 ? hp_rows[(size_t)i * row_len + c->d + 4] : eta;
    hps[i].wellcond = wellcond;
    hps[i].closed_form_varK = closed_form_varK;
    hps[i].hp_kernel = row_len >= c->d + 4 ? row[c->d + 3] : 0.0;   // RatQu: alpha in column d + 3
    rc = check_hp(c, &hps[i]);
    if (rc) return rc;
  }
  GPG_HIP_OK(c, hipMemsetAsync(c->info, 0, sizeof(int) * m, c->stream));
  int B = batch_plan(c, m);
  if (B > 1) {   // equal groups (10 rows with room for 8 -> 5 + 5, not 8 + 2)
    const int ngroups = (m + B - 1) / B;
    B = (m + ngroups - 1) / ngroups;
  }
  if (B > 1) {
    // the launch helpers read the context's pointers: they are redirected to the batch workspaces for the loop and put
    // back by the guard on EVERY way out of this scope (an early error return must not leave them redirected)
    struct Restore {
      gpg_ctx* c; double *A, *dvec, *invp, *dinv; int* info;
      ~Restore() { c->A = A; c->dvec = dvec; c->invp = invp; c->dinv = dinv; c->info = info; }
    } restore{c, c->A, c->dvec, c->invp, c->dinv, c->info};
    int* info0 = c->info;
    c->prep_valid = false;                                        // dvec / invp no longer belong to one hyperparameter row
    c->last_factor_ws = 0;
    for (int r0 = 0; r0 < m; r0 += B) {
      const int Bg = (m - r0) < B ? (m - r0) : B;
      // matrix b of the group: workspace batchA + b A_elems, vectors dvec | invp | dinv at batchV + 3 b Npad
      c->A = c->batchA;
      c->dvec = c->batchV;
      c->invp = c->dvec + c->Npad;
      c->dinv = c->invp + c->Npad;
      if (Bg > 1) {   // ONE launch each of prep_diag / prep_rows / assemble / factorise / reduce for the whole group
        for (int b = 0; b < Bg; ++b) {
          gpg_batch_item& it = c->items_host[r0 + b];
          it.p = make_params(c, &hps[r0 + b], 0);
          it.var_fval = hps[r0 + b].var_fval;
          it.var_fgrad = hps[r0 + b].var_fgrad;
        }
        c->last_precon = c->items_host[r0].p.precon;             // wellcond is shared by the rows of a call
        GPG_HIP_OK(c, hipMemcpyAsync(c->items_dev + r0, c->items_host + r0, sizeof(gpg_batch_item) * Bg, hipMemcpyHostToDevice,
                                     c->stream));
        gpg_launch_prep_assembly_batch(c, c->items_host[r0].p, Bg, c->items_dev + r0, 3 * (size_t)c->Npad, c->A_elems);
        gpg_launch_tile_chol_batch(c, Bg, c->batchA, c->A_elems, c->batchV + 2 * (size_t)c->Npad, 3 * c->Npad, info0 + r0);
        gpg_launch_lkd_reduce_batch(c, r0, Bg, 3 * (size_t)c->Npad, c->A_elems, info0 + r0);
      } else {
        AsmParams p = make_params(c, &hps[r0], 0);
        c->last_precon = p.precon;
        gpg_launch_prep(c, p, hps[r0].var_fval, hps[r0].var_fgrad, 1.0, 0.0, 0.0, 1.0);
        gpg_launch_assembly(c, p);
        c->info = info0 + r0;
        gpg_cholesky(c);
        gpg_launch_lkd_reduce(c, r0);
      }
    }
  } else {
    for (int i = 0; i < m; ++i) enqueue_lkd(c, &hps[i], i);
  }
  GPG_HIP_OK(c, hipMemcpyAsync(c->h_scal, c->scal, sizeof(double) * 8 * m, hipMemcpyDeviceToHost, c->stream));
  GPG_HIP_OK(c, hipStreamSynchronize(c->stream));
  GPG_HIP_OK(c, hipGetLastError());
  GPG_LAUNCH_OK(c);
  for (int i = 0; i < m; ++i) c->h_info[i] = (int)c->h_scal[(size_t)8 * i + 7];
  if (internal_failure(c, c->h_info, m)) return -4;
  for (int i = 0; i < m; ++i) finish_lkd(c, &hps[i], c->h_scal + (size_t)8 * i, c->h_info[i], &out[i]);
  c->factor_valid = false;
  return 0;
}

// Value + adjoint gradient of m restart rows (gpg_lkd_grad_batch).  Groups of up to B rows: ONE launch each of
// prep / assembly / factorisation / reduction for the group (as gpg_lkd_batch), the short per-matrix steps (alpha by the
// vector solve) one after the other, then ONE launch each of W = L^-T and -(W W^T) for the whole group, then the fused
// contraction per matrix.  One synchronisation at the end.  Same arithmetic per row as gpg_lkd_grad (bit-identical).
static int gpg_lkd_grad_batch_once(gpg_ctx* c, int m, const double* hp_rows, int row_len, double eta, int wellcond,
                                   int closed_form_varK, gpg_lkd_out* out, double* g_aa, double* g_inv) {
  if (!c) return -1;
  if (m < 1 || !hp_rows || !out || !g_aa || !g_inv || row_len < c->d + 3) { c->err = "bad batch arguments (row_len >= d + 3)"; return -1; }
  GPG_HIP_OK(c, hipSetDevice(c->device));
  GPG_WS(c, 0);
  c->alpha_valid = false;                                  // the vectors of set 0 are redirected / rewritten by the groups
  int rc = ensure_scal(c, m);
  if (rc) return rc;
  if (m > c->items_cap) {
    if (c->items_dev) (void)hipFree(c->items_dev);
    if (c->items_host) (void)hipHostFree(c->items_host);
    c->items_dev = c->items_host = nullptr; c->items_cap = 0;
    GPG_HIP_OK(c, hipMalloc(&c->items_dev, sizeof(gpg_batch_item) * m));
    GPG_HIP_OK(c, hipHostMalloc(&c->items_host, sizeof(gpg_batch_item) * m));
    c->items_cap = m;
  }
  std::vector<gpg_hp> hps(m);
  for (int i = 0; i < m; ++i) {
    const double* row = hp_rows + (size_t)i * row_len;
    hps[i].theta = row;
    hps[i].varK_mat = row[c->d];
    hps[i].var_fval = row[c->d + 1];
    hps[i].var_fgrad = row[c->d + 2];
    hps[i].eta = row_len > c->d + 4 ? hp_rows[(size_t)i * row_len + c->d + 4] : eta;
    hps[i].wellcond = wellcond;
    hps[i].closed_form_varK = closed_form_varK;
    hps[i].hp_kernel = row_len >= c->d + 4 ? row[c->d + 3] : 0.0;
    rc = check_hp(c, &hps[i]);
    if (rc) return rc;
  }
  const int slots = GPG_GRAD_SLOTS_MAX, nval = 2 * slots;
  const int ns = c->d + 3 + (c->kernel == GPG_KERNEL_RATQU ? 1 : 0);
  // rows per group: what the value path batches (batch_plan also allocates batchA / batchV), at most 8, and within
  // 96 GB for the two Npad^2 arrays (W, Minv) per matrix
  int B = batch_plan(c, m);
  const size_t nn = (size_t)c->Npad * c->Npad;
  if (B > 8) B = 8;
  while (B > 1 && 2 * nn * sizeof(double) * B > ((size_t)96 << 30)) --B;
  const bool dataflow = c->chol_impl != 0 || c->tail_cols != 0;
  if (!dataflow || c->Npad < 512) B = 1;           // blocked mode / tiny matrices: one row at a time through gpg_lkd_grad's steps
  if (B > c->gbatch_cap) {
    if (c->batchW) (void)hipFree(c->batchW);
    if (c->batchM) (void)hipFree(c->batchM);
    if (c->batchZ) (void)hipFree(c->batchZ);
    c->batchW = c->batchM = c->batchZ = nullptr; c->gbatch_cap = 0;
    GPG_HIP_OK(c, hipMalloc(&c->batchW, sizeof(double) * nn * B));
    GPG_HIP_OK(c, hipMalloc(&c->batchM, sizeof(double) * nn * B));
    GPG_HIP_OK(c, hipMalloc(&c->batchZ, sizeof(double) * (size_t)c->vec_rows_cols * B));
    c->gbatch_cap = B;
  }
  const int nblk = gpg_grad_partial_blocks(c);
  if (!c->gpartial) GPG_HIP_OK(c, hipMalloc(&c->gpartial, sizeof(double) * (size_t)nval * (nblk + 1)));
  if (m > c->gres_cap) {
    if (c->gres) (void)hipFree(c->gres);
    c->gres = nullptr; c->gres_cap = 0;
    GPG_HIP_OK(c, hipMalloc(&c->gres, sizeof(double) * (size_t)nval * m));
    c->gres_cap = m;
  }
  GPG_HIP_OK(c, hipMemsetAsync(c->info, 0, sizeof(int) * m, c->stream));
  GPG_HIP_OK(c, hipMemsetAsync(c->gres, 0, sizeof(double) * (size_t)nval * m, c->stream));
  if (B <= 1) {   // one row at a time, still queued back-to-back with one synchronisation
    if (!c->Wfull) {
      GPG_HIP_OK(c, hipMalloc(&c->Wfull, sizeof(double) * nn));
      GPG_HIP_OK(c, hipMalloc(&c->Minv, sizeof(double) * nn));
    }
    for (int i = 0; i < m; ++i) {
      enqueue_lkd(c, &hps[i], i);
      AsmParams p = make_params(c, &hps[i], 0);
      gpg_launch_combine_rows(c, i);
      gpg_backward_solve(c);
      gpg_inverse_from_factor(c, c->Wfull, c->Minv);
      gpg_launch_grad_contract(c, p, c->gpartial, c->gres + (size_t)nval * i, c->zvec, c->Minv);
    }
  } else {
    struct Restore {
      gpg_ctx* c; double *A, *dvec, *invp, *dinv, *zvec; int* info;
      ~Restore() { c->A = A; c->dvec = dvec; c->invp = invp; c->dinv = dinv; c->zvec = zvec; c->info = info; }
    } restore{c, c->A, c->dvec, c->invp, c->dinv, c->zvec, c->info};
    int* info0 = c->info;
    c->prep_valid = false;
    c->last_factor_ws = 0;
    const size_t vs = 3 * (size_t)c->Npad;
    const int ngroups = (m + B - 1) / B;
    const int Bq = (m + ngroups - 1) / ngroups;                   // equal groups
    for (int r0 = 0; r0 < m; r0 += Bq) {
      const int Bg = (m - r0) < Bq ? (m - r0) : Bq;
      c->A = c->batchA;
      c->dvec = c->batchV;
      c->invp = c->dvec + c->Npad;
      c->dinv = c->invp + c->Npad;
      c->info = info0;
      for (int b = 0; b < Bg; ++b) {
        gpg_batch_item& it = c->items_host[r0 + b];
        it.p = make_params(c, &hps[r0 + b], 0);
        it.var_fval = hps[r0 + b].var_fval;
        it.var_fgrad = hps[r0 + b].var_fgrad;
      }
      c->last_precon = c->items_host[r0].p.precon;
      GPG_HIP_OK(c, hipMemcpyAsync(c->items_dev + r0, c->items_host + r0, sizeof(gpg_batch_item) * Bg, hipMemcpyHostToDevice, c->stream));
      gpg_launch_prep_assembly_batch(c, c->items_host[r0].p, Bg, c->items_dev + r0, vs, c->A_elems);
      // small matrices: W = L^-T of the group on the second stream behind the factorisation (cholesky_dataflow.hip)
      const bool overlap = c->Npad <= 4096 && gpg_overlap_inverse_begin(c, Bg);
      if (Bg > 1) gpg_launch_tile_chol_batch(c, Bg, c->batchA, c->A_elems, c->batchV + 2 * (size_t)c->Npad, 3 * c->Npad, info0 + r0);
      else { c->info = info0 + r0; gpg_cholesky(c); c->info = info0; }
      c->chol_flags_override = nullptr;
      if (overlap && !gpg_overlap_inverse_trinv(c, Bg, c->batchA, c->A_elems, c->batchV + 2 * (size_t)c->Npad, 3 * c->Npad, c->batchW, info0 + r0)) {
        c->err = "overlapped batched inverse: launch refused";
        return -2;
      }
      gpg_launch_lkd_reduce_batch(c, r0, Bg, vs, c->A_elems, info0 + r0);
      for (int b = 0; b < Bg; ++b) {     // alpha of matrix b: RHS row 0 <- w2 - beta w1, then z = L^-T (.) by the vector solve
        c->A = c->batchA + (size_t)b * c->A_elems;
        c->dinv = c->batchV + (size_t)b * vs + 2 * (size_t)c->Npad;
        c->zvec = c->batchZ + (size_t)b * c->vec_rows_cols;
        c->info = info0 + r0 + b;
        gpg_launch_combine_rows(c, r0 + b);
        gpg_backward_solve(c);
      }
      c->info = info0;
      if (overlap ? !gpg_overlap_inverse_wwt(c, Bg, c->batchA, c->A_elems, c->batchV + 2 * (size_t)c->Npad, 3 * c->Npad, c->batchW, c->batchM,
                                             info0 + r0)
                  : !gpg_launch_tile128_inverse_batch(c, Bg, c->batchA, c->A_elems, c->batchV + 2 * (size_t)c->Npad, 3 * c->Npad, c->batchW,
                                                      c->batchM, info0 + r0)) {
        c->err = "batched inverse could not be launched";
        return -2;
      }
      for (int b = 0; b < Bg; ++b) {
        c->invp = c->batchV + (size_t)b * vs + c->Npad;
        gpg_launch_grad_contract(c, c->items_host[r0 + b].p, c->gpartial, c->gres + (size_t)nval * (r0 + b),
                                 c->batchZ + (size_t)b * c->vec_rows_cols, c->batchM + (size_t)b * nn);
      }
    }
  }
  std::vector<double> hres((size_t)nval * m);
  GPG_HIP_OK(c, hipMemcpyAsync(hres.data(), c->gres, sizeof(double) * hres.size(), hipMemcpyDeviceToHost, c->stream));
  GPG_HIP_OK(c, hipMemcpyAsync(c->h_scal, c->scal, sizeof(double) * 8 * m, hipMemcpyDeviceToHost, c->stream));
  GPG_HIP_OK(c, hipMemcpyAsync(c->h_info, c->info, sizeof(int) * m, hipMemcpyDeviceToHost, c->stream));
  GPG_HIP_OK(c, hipStreamSynchronize(c->stream));
  GPG_HIP_OK(c, hipGetLastError());
  GPG_LAUNCH_OK(c);
  if (internal_failure(c, c->h_info, m)) return -4;
  for (int i = 0; i < m; ++i) {
    finish_lkd(c, &hps[i], c->h_scal + (size_t)8 * i, c->h_info[i], &out[i]);
    double* ga = g_aa + (size_t)i * (c->d + 4);
    double* gi = g_inv + (size_t)i * (c->d + 4);
    for (int k = 0; k < c->d + 4; ++k) ga[k] = gi[k] = 0.0;
    if (c->h_info[i] != 0) { for (int k = 0; k < c->d + 4; ++k) ga[k] = gi[k] = std::numeric_limits<double>::quiet_NaN(); continue; }
    for (int k = 0; k < ns; ++k) { ga[k] = hres[(size_t)nval * i + k]; gi[k] = hres[(size_t)nval * i + ns + k]; }
    if (c->kernel == GPG_KERNEL_RATQU) { ga[c->d + 3] = hres[(size_t)nval * i + ns - 1]; gi[c->d + 3] = hres[(size_t)nval * i + 2 * ns - 1]; }
  }
  c->factor_valid = false;
  return 0;
}

static int gpg_setup_eval_once(gpg_ctx* c, const gpg_hp* hp, double beta, double* alpha_out) {
  int rc = check_hp(c, hp);
  if (rc) return rc;
  GPG_HIP_OK(c, hipSetDevice(c->device));
  GPG_WS(c, 1);                                   // the posterior's own factor storage (GpEvalModel.py:17-57)
  c->eval_ready = false;
  AsmParams p = make_params(c, hp, 0);
  c->last_precon = p.precon;
  c->prep_valid = true;
  c->last_factor_ws = 1;
  GPG_HIP_OK(c, hipMemsetAsync(c->info, 0, sizeof(int), c->stream));
  gpg_launch_prep(c, p, hp->var_fval, hp->var_fgrad, -beta, 1.0, 0.0, 0.0);   // RHS row 0 = (y - V beta) P^-1
  gpg_launch_assembly(c, p);
  gpg_cholesky(c);
  GPG_HIP_OK(c, hipMemcpyAsync(c->h_info, c->info, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  GPG_HIP_OK(c, hipStreamSynchronize(c->stream));
  GPG_HIP_OK(c, hipGetLastError());
  GPG_LAUNCH_OK(c);
  if (internal_failure(c, c->h_info, 1)) return -4;
  if (c->h_info[0] != 0) { c->factor_valid = false; return c->h_info[0]; }
  c->factor_valid = true;
  gpg_backward_solve(c);
  if (alpha_out) {
    gpg_launch_alpha(c, c->tmpv);
    GPG_HIP_OK(c, hipMemcpyAsync(alpha_out, c->tmpv, sizeof(double) * c->N, hipMemcpyDeviceToHost, c->stream));
  }
  GPG_HIP_OK(c, hipMemcpyAsync(c->h_info, c->info, sizeof(int), hipMemcpyDeviceToHost, c->stream));   // the dataflow backward solve
  GPG_HIP_OK(c, hipStreamSynchronize(c->stream));                                                      // reports a timed-out wait here
  GPG_HIP_OK(c, hipGetLastError());
  GPG_LAUNCH_OK(c);
  if (solve_failure(c)) return -4;
  c->eval_ready = true;
  c->eval_beta = beta;
  c->eval_params = p;   // theta etc. for the cross kernel
  c->err.clear();
  return 0;
}

// Dataflow launches that share the GPU with another such launch (another process / context on the same device) can
// starve each other: every wait inside them is bounded, the launch drains and reports GPG_INFO_INTERNAL (-4 here).  What is
// repeated, and what the context gives up, depends on WHAT timed out:
//   * a call that had the inverse W = L^-T overlapped with its factorisation: once more without the overlap (and the context
//     stops overlapping; gpg_overlap_fallbacks counts) -- the dataflow factorisation and solves stay;
//   * a dataflow triangular solve (the factor is untouched): once more with the blocked sweeps, which the context keeps for its
//     solves (gpg_solve_fallbacks counts) -- the dataflow factorisation stays;
//   * the factorisation itself: once more with the blocked schedule, which has no inter-workgroup waits, and the context stays
//     on it (gpg_factor_fallbacks counts).
// gpg_set_factor_mode re-arms all three.  Before anything is repeated every stream of the context is drained: workgroups of a
// timed-out launch may still be on their way out and write the info word or W.
static void drain_streams(gpg_ctx* c) {
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->stream_inv) (void)hipStreamSynchronize(c->stream_inv);
  if (c->stream_upd) (void)hipStreamSynchronize(c->stream_upd);
  (void)hipGetLastError();
}
static int with_fallback(gpg_ctx* c, const std::function<int()>& f) {
  if (c) { c->overlap_used = false; c->fail_kind = 0; }
  int rc = f();
  if (rc == -4 && c && c->overlap_used) {
    drain_streams(c);
    c->overlap_inverse = 0;
    c->overlap_fallbacks += 1;
    c->overlap_used = false; c->fail_kind = 0;
    rc = f();
  }
  if (rc == -4 && c && c->fail_kind == 2 && c->solve_dataflow) {
    drain_streams(c);
    c->solve_dataflow = 0;
    c->solve_fallbacks += 1;
    c->fail_kind = 0;
    rc = f();
  }
  if (rc == -4 && c && (c->chol_impl != 0 || c->tail_cols != 0)) {
    drain_streams(c);
    c->chol_impl = 0;
    c->tail_cols = 0;
    c->factor_fallbacks += 1;
    rc = f();
  }
  return rc;
}

int gpg_lkd(gpg_ctx* c, const gpg_hp* hp, gpg_lkd_out* out) {
  return with_fallback(c, [&] { return gpg_lkd_once(c, hp, out); });
}
int gpg_lkd_grad(gpg_ctx* c, const gpg_hp* hp, gpg_lkd_out* out, double* g_aa, double* g_inv) {
  return with_fallback(c, [&] { return gpg_lkd_grad_once(c, hp, out, g_aa, g_inv); });
}
int gpg_lkd_batch(gpg_ctx* c, int m, const double* hp_rows, int row_len, double eta, int wellcond, int closed_form_varK,
                  gpg_lkd_out* out) {
  return with_fallback(c, [&] { return gpg_lkd_batch_once(c, m, hp_rows, row_len, eta, wellcond, closed_form_varK, out); });
}
int gpg_lkd_grad_batch(gpg_ctx* c, int m, const double* hp_rows, int row_len, double eta, int wellcond, int closed_form_varK,
                       gpg_lkd_out* out, double* g_aa, double* g_inv) {
  return with_fallback(c, [&] { return gpg_lkd_grad_batch_once(c, m, hp_rows, row_len, eta, wellcond, closed_form_varK, out, g_aa, g_inv); });
}
int gpg_setup_eval(gpg_ctx* c, const gpg_hp* hp, double beta, double* alpha_out) {
  return with_fallback(c, [&] { return gpg_setup_eval_once(c, hp, beta, alpha_out); });
}
int gpg_factor_fallbacks(gpg_ctx* c) { return c ? c->factor_fallbacks : -1; }
int gpg_overlap_fallbacks(gpg_ctx* c) { return c ? c->overlap_fallbacks : -1; }
int gpg_solve_fallbacks(gpg_ctx* c) { return c ? c->solve_fallbacks : -1; }
int gpg_last_factor(gpg_ctx* c, int* kernel, int* matrices) {
  if (!c) return -1;
  if (kernel) *kernel = c->last_factor_kernel;
  if (matrices) *matrices = c->last_factor_batch;
  return 0;
}

static int predict_impl(gpg_ctx* c, int nx, const double* xq, double varK, double* mu, double* sig, double* sig2_raw,
                        double* dmudx, double* dsigdx, double* dsig2dx = nullptr);


int gpg_predict(gpg_ctx* c, int nx, const double* xq, double varK, double* mu, double* sig, double* sig2_raw) {
  return with_fallback(c, [&] { return predict_impl(c, nx, xq, varK, mu, sig, sig2_raw, nullptr, nullptr); });
}

int gpg_predict_grad(gpg_ctx* c, int nx, const double* xq, double varK, double* mu, double* sig, double* sig2_raw,
                     double* dmudx, double* dsigdx) {
  if (c && (!dmudx || !dsigdx)) { c->err = "dmudx / dsigdx is NULL"; return -1; }
  return with_fallback(c, [&] { return predict_impl(c, nx, xq, varK, mu, sig, sig2_raw, dmudx, dsigdx); });
}

int gpg_predict_var(gpg_ctx* c, int nx, const double* xq, double varK, double* sig2, double* dsig2dx) {
  if (c && !sig2) { c->err = "sig2 is NULL"; return -1; }
  if (!c || nx < 1) { if (c) c->err = "bad predict arguments"; return -1; }
  std::vector<double> mu(nx), sg(nx), dmu(dsig2dx ? (size_t)nx * c->d : 0), dsg(dmu.size());
  int rc = with_fallback(c, [&] {
    return predict_impl(c, nx, xq, varK, mu.data(), sg.data(), sig2, dsig2dx ? dmu.data() : nullptr,
                        dsig2dx ? dsg.data() : nullptr, dsig2dx);
  });
  if (rc) return rc;
  for (int j = 0; j < nx; ++j) sig2[j] *= varK;            // GpEvalModel.py:297: varK (1 - diag(Kxy K^-1 Kyx)), not clipped
  return 0;
}

static int predict_impl(gpg_ctx* c, int nx, const double* xq, double varK, double* mu, double* sig, double* sig2_raw,
                        double* dmudx, double* dsigdx, double* dsig2dx) {
  if (!c) return -1;
  if (!c->eval_ready) { c->err = "gpg_setup_eval must succeed before gpg_predict"; return -1; }
  if (nx < 1 || !xq || !mu || !sig) { c->err = "bad predict arguments"; return -1; }
  GPG_HIP_OK(c, hipSetDevice(c->device));
  GPG_WS(c, 1);
  const int nxp = ((nx + 63) / 64) * 64;
  if (nxp > c->xq_cap) {
    if (c->Wt) (void)hipFree(c->Wt);
    if (c->xq_dev) (void)hipFree(c->xq_dev);
    if (c->musig) (void)hipFree(c->musig);
    if (c->gradbuf) (void)hipFree(c->gradbuf);
    c->Wt = c->xq_dev = c->musig = c->gradbuf = nullptr; c->xq_cap = 0;
    GPG_HIP_OK(c, hipMalloc(&c->Wt, sizeof(double) * (size_t)nxp * c->Npad));
    GPG_HIP_OK(c, hipMalloc(&c->xq_dev, sizeof(double) * (size_t)nxp * c->d));
    GPG_HIP_OK(c, hipMalloc(&c->musig, sizeof(double) * 2 * nxp));
    GPG_HIP_OK(c, hipMalloc(&c->gradbuf, sizeof(double) * ((size_t)2 * nxp * GPG_MAX_DIM + (size_t)nxp * 64)));
    c->xq_cap = nxp;
  }
  // the allocation may be larger than this call's nxp: kernels index with this call's nxp
  std::vector<double> xt((size_t)nxp * c->d, 0.0);
  for (int j = 0; j < nx; ++j)
    for (int k = 0; k < c->d; ++k) xt[(size_t)k * nxp + j] = xq[(size_t)j * c->d + k];
  GPG_HIP_OK(c, hipMemcpyAsync(c->xq_dev, xt.data(), sizeof(double) * xt.size(), hipMemcpyHostToDevice, c->stream));
  GPG_HIP_OK(c, hipMemsetAsync(c->Wt, 0, sizeof(double) * (size_t)nxp * c->Npad, c->stream));
  GPG_HIP_OK(c, hipMemsetAsync(c->info, 0, sizeof(int), c->stream));
  AsmParams p = c->eval_params;
  gpg_launch_cross(c, p, nx, nxp);
  gpg_launch_predict_reduce(c, nx, nxp, c->eval_beta, varK, 0);
  gpg_forward_rows(c, c->Wt, nxp, nxp, nx);
  gpg_launch_predict_reduce(c, nx, nxp, c->eval_beta, varK, 1);
  std::vector<double> hgrad_vec;
  double* hgrad = nullptr;
  size_t n_hgrad = 0;
  double* pin = pinned_scratch(c, 2 * (size_t)nxp * (1 + GPG_MAX_DIM));   // mu | sig2 | the two gradient blocks
  if (dmudx) {
    // K^-1 Kyx needs the second triangular sweep too (GpEvalModel.py:154), then the fused gradient reductions
    double* g1 = c->gradbuf;
    double* g2 = c->gradbuf + (size_t)nxp * GPG_MAX_DIM;
    double* tbuf = c->gradbuf + (size_t)2 * nxp * GPG_MAX_DIM;
    gpg_backward_rows(c, c->Wt, nxp, nx, tbuf);
    gpg_launch_cross_grad(c, p, nx, nxp, g1, g2);
    n_hgrad = (size_t)2 * nxp * GPG_MAX_DIM;
    hgrad = pin ? pin + 2 * (size_t)nxp : nullptr;
    if (!hgrad) { hgrad_vec.resize(n_hgrad); hgrad = hgrad_vec.data(); }
    GPG_HIP_OK(c, hipMemcpyAsync(hgrad, c->gradbuf, sizeof(double) * n_hgrad, hipMemcpyDeviceToHost, c->stream));
  }
  std::vector<double> host_vec;
  double* host = pin;
  if (!host) { host_vec.resize(2 * (size_t)nxp); host = host_vec.data(); }
  GPG_HIP_OK(c, hipMemcpyAsync(host, c->musig, sizeof(double) * 2 * nxp, hipMemcpyDeviceToHost, c->stream));
  GPG_HIP_OK(c, hipMemcpyAsync(c->h_info, c->info, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  GPG_HIP_OK(c, hipStreamSynchronize(c->stream));
  GPG_HIP_OK(c, hipGetLastError());
  GPG_LAUNCH_OK(c);
  if (solve_failure(c)) return -4;
  const double sigK = sqrt(varK);
  for (int j = 0; j < nx; ++j) {
    mu[j] = host[j];
    double s2 = host[nxp + j];
    if (sig2_raw) sig2_raw[j] = s2;
    sig[j] = sqrt(s2 < 0.0 ? 0.0 : s2) * sigK;   // GpEvalModel.py:165-166
    if (dmudx) {
      const double inv_sig = sig[j] != 0.0 ? 1.0 / sig[j] : 0.0;      // GpEvalModel.py:346
      for (int k = 0; k < c->d; ++k) {
        dmudx[(size_t)j * c->d + k] = hgrad[(size_t)j * c->d + k];                                        // :319-325
        dsigdx[(size_t)j * c->d + k] = -inv_sig * (hgrad[(size_t)nxp * GPG_MAX_DIM + (size_t)j * c->d + k] * varK);   // :339-354
        if (dsig2dx) dsig2dx[(size_t)j * c->d + k] = -2.0 * hgrad[(size_t)nxp * GPG_MAX_DIM + (size_t)j * c->d + k] * varK;   // :327-337
      }
    }
  }
  return 0;
}

static int predict_hess_once(gpg_ctx* c, const double* xq, double varK, double* mu, double* sig, double* dmudx,
                             double* dsigdx, double* d2mudx2, double* d2sigdx2);

int gpg_predict_hess(gpg_ctx* c, const double* xq, double varK, double* mu, double* sig, double* dmudx, double* dsigdx,
                     double* d2mudx2, double* d2sigdx2) {
  if (c && (!d2mudx2 || !d2sigdx2 || !dmudx || !dsigdx)) { c->err = "Hessian / gradient output is NULL"; return -1; }
  return with_fallback(c, [&] { return predict_hess_once(c, xq, varK, mu, sig, dmudx, dsigdx, d2mudx2, d2sigdx2); });
}

static int predict_hess_once(gpg_ctx* c, const double* xq, double varK, double* mu, double* sig, double* dmudx,
                             double* dsigdx, double* d2mudx2, double* d2sigdx2) {
  double s2 = 0.0;
  int rc = predict_impl(c, 1, xq, varK, mu, sig, &s2, dmudx, dsigdx);   // leaves K^-1 Kyx in row 0 of Wt
  if (rc) return rc;
  const int d = c->d, nxp = 64;
  AsmParams p = c->eval_params;
  // gradbuf: [2 * nxp * 16] gradient reductions (consumed), then 3 * 256 doubles for H1, H2, T
  double* h1 = c->gradbuf;
  double* h2 = h1 + GPG_MAX_DIM * GPG_MAX_DIM;
  double* T = h2 + GPG_MAX_DIM * GPG_MAX_DIM;
  gpg_launch_hess_stage(c, p, nxp, h1, h2, T, 0);
  GPG_HIP_OK(c, hipMemsetAsync(c->Wt, 0, sizeof(double) * (size_t)nxp * c->Npad, c->stream));
  gpg_launch_hess_stage(c, p, nxp, h1, h2, T, 1);
  gpg_forward_rows(c, c->Wt, nxp, nxp, d + 1);
  gpg_launch_hess_stage(c, p, nxp, h1, h2, T, 2);
  std::vector<double> h(3 * GPG_MAX_DIM * GPG_MAX_DIM);
  GPG_HIP_OK(c, hipMemcpyAsync(h.data(), h1, sizeof(double) * h.size(), hipMemcpyDeviceToHost, c->stream));
  GPG_HIP_OK(c, hipMemcpyAsync(c->h_info, c->info, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  GPG_HIP_OK(c, hipStreamSynchronize(c->stream));
  GPG_HIP_OK(c, hipGetLastError());
  GPG_LAUNCH_OK(c);
  if (solve_failure(c)) return -4;
  const double* H1 = h.data();
  const double* H2 = H1 + GPG_MAX_DIM * GPG_MAX_DIM;
  const double* TT = H2 + GPG_MAX_DIM * GPG_MAX_DIM;
  const double sg = sig[0];
  for (int k = 0; k < d; ++k)
    for (int i = 0; i < d; ++i) {
      d2mudx2[k * d + i] = H1[k * d + i];                                               // GpEvalModel.py:355-363
      const double d2s2 = -2.0 * varK * (H2[k * d + i] + TT[k * d + i]);                // :370-372
      d2sigdx2[k * d + i] = sg != 0.0 ? (d2s2 - 2.0 * dsigdx[k] * dsigdx[i]) / (2.0 * sg) : NAN;   // :375-378
    }
  return 0;
}

int gpg_get_matrix(gpg_ctx* c, const gpg_hp* hp, int which, double* out) {
  if (!c) return -1;
  if (!out || which < 0 || which > 4) { c->err = "bad get_matrix arguments"; return -1; }
  GPG_HIP_OK(c, hipSetDevice(c->device));
  if (!c->dense_tmp) GPG_HIP_OK(c, hipMalloc(&c->dense_tmp, sizeof(double) * (size_t)c->N * c->N));
  if (which == 4) {
    if (!c->eval_ready) { c->err = "no factor kept by gpg_setup_eval on the device"; return -1; }
    GPG_WS(c, 1);
  } else if (which == 3) {
    GPG_WS(c, c->last_factor_ws);
    if (!c->factor_valid) { c->err = "no valid factor on the device"; return -1; }
  } else {
    int rc = check_hp(c, hp);
    if (rc) return rc;
    GPG_WS(c, 0);
    c->prep_valid = true;
    AsmParams p = make_params(c, hp, which == 0 ? 1 : (which == 1 ? 2 : 0));
    c->last_precon = p.precon;
    gpg_launch_prep(c, p, hp->var_fval, hp->var_fgrad, 0.0, 0.0, 0.0, 0.0);
    unsigned save = c->prof_mask; c->prof_mask = 0;
    gpg_launch_assembly(c, p);
    c->prof_mask = save;
    c->factor_valid = false;
  }
  gpg_launch_extract(c, which == 4 ? 3 : which);
  GPG_HIP_OK(c, hipMemcpyAsync(out, c->dense_tmp, sizeof(double) * (size_t)c->N * c->N, hipMemcpyDeviceToHost,
                               c->stream));
  GPG_HIP_OK(c, hipStreamSynchronize(c->stream));
  GPG_HIP_OK(c, hipGetLastError());
  GPG_LAUNCH_OK(c);
  return 0;
}

int gpg_kern_rtensor(int device, int kernel, int dim, int n1, int n2, const double* rtensor, const double* theta, double hp_kernel,
                     int use_grad, const unsigned char* use_grad1, const unsigned char* use_grad2, double* out) {
  if (!rtensor || !theta || !out || dim < 1 || dim > GPG_MAX_DIM || n1 < 1 || n2 < 1) { g_create_err = "bad gpg_kern_rtensor arguments (1 <= dim <= 16)"; return -1; }
  if (kernel != GPG_KERNEL_SQEXP && kernel != GPG_KERNEL_MA5F2 && kernel != GPG_KERNEL_RATQU) { g_create_err = "unknown kernel id"; return -1; }
  if (kernel == GPG_KERNEL_RATQU && !(hp_kernel > 0.0)) { g_create_err = "hp_kernel (alpha of RatQu) must be positive"; return -1; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { g_create_err = "no HIP device visible"; return -2; }
  if (device < 0 || device >= ndev) { g_create_err = "device index out of range"; return -1; }
  std::vector<int> gp1(n1, -1), gp2(n2, -1);
  int n1g = 0, n2g = 0;
  if (use_grad) {
    for (int a = 0; a < n1; ++a) if (!use_grad1 || use_grad1[a]) gp1[a] = n1g++;
    for (int b = 0; b < n2; ++b) if (!use_grad2 || use_grad2[b]) gp2[b] = n2g++;
  }
  const size_t R1 = (size_t)n1 + (size_t)n1g * dim, C2 = (size_t)n2 + (size_t)n2g * dim;
  const size_t nrt = (size_t)dim * n1 * n2;
  double *d_rt = nullptr, *d_out = nullptr;
  int *d_g1 = nullptr, *d_g2 = nullptr;
  hipStream_t st = nullptr;
  int rc = 0;
  auto fail = [&](const char* what, hipError_t e) { g_create_err = std::string(what) + ": " + hipGetErrorString(e); rc = -2; };
  hipError_t e;
  if ((e = hipSetDevice(device)) != hipSuccess) fail("hipSetDevice", e);
  if (!rc && (e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking)) != hipSuccess) fail("hipStreamCreate", e);
  if (!rc && (e = hipMalloc(&d_rt, sizeof(double) * nrt)) != hipSuccess) fail("hipMalloc(rtensor)", e);
  if (!rc && (e = hipMalloc(&d_out, sizeof(double) * R1 * C2)) != hipSuccess) fail("hipMalloc(out)", e);
  if (!rc && (e = hipMalloc(&d_g1, sizeof(int) * n1)) != hipSuccess) fail("hipMalloc(gpos1)", e);
  if (!rc && (e = hipMalloc(&d_g2, sizeof(int) * n2)) != hipSuccess) fail("hipMalloc(gpos2)", e);
  if (!rc && (e = hipMemcpyAsync(d_rt, rtensor, sizeof(double) * nrt, hipMemcpyHostToDevice, st)) != hipSuccess) fail("copy rtensor", e);
  if (!rc && (e = hipMemcpyAsync(d_g1, gp1.data(), sizeof(int) * n1, hipMemcpyHostToDevice, st)) != hipSuccess) fail("copy gpos1", e);
  if (!rc && (e = hipMemcpyAsync(d_g2, gp2.data(), sizeof(int) * n2, hipMemcpyHostToDevice, st)) != hipSuccess) fail("copy gpos2", e);
  if (!rc && (e = hipMemsetAsync(d_out, 0, sizeof(double) * R1 * C2, st)) != hipSuccess) fail("memset out", e);
  if (!rc && gpg_kern_rtensor_run(kernel, dim, n1, n2, n1g, n2g, use_grad ? 1 : 0, theta, hp_kernel, d_rt, d_g1, d_g2, d_out, st) != 0) {
    g_create_err = "rtensor kernel launch failed"; rc = -2;
  }
  if (!rc && (e = hipMemcpyAsync(out, d_out, sizeof(double) * R1 * C2, hipMemcpyDeviceToHost, st)) != hipSuccess) fail("copy out", e);
  if (!rc && (e = hipStreamSynchronize(st)) != hipSuccess) fail("hipStreamSynchronize", e);
  if (d_rt) (void)hipFree(d_rt);
  if (d_out) (void)hipFree(d_out);
  if (d_g1) (void)hipFree(d_g1);
  if (d_g2) (void)hipFree(d_g2);
  if (st) (void)hipStreamDestroy(st);
  return rc;
}

int gpg_kern_rtensor_grad_hp(int device, int kernel, int dim, int n, const double* rtensor, const double* theta, double hp_kernel, int use_grad,
                             double* out_theta, double* out_alpha) {
  if (!rtensor || !theta || (!out_theta && !out_alpha) || dim < 1 || dim > GPG_MAX_DIM || n < 1) { g_create_err = "bad gpg_kern_rtensor_grad_hp arguments (1 <= dim <= 16)"; return -1; }
  if (kernel != GPG_KERNEL_SQEXP && kernel != GPG_KERNEL_MA5F2 && kernel != GPG_KERNEL_RATQU) { g_create_err = "unknown kernel id"; return -1; }
  if (kernel == GPG_KERNEL_RATQU && !(hp_kernel > 0.0)) { g_create_err = "hp_kernel (alpha of RatQu) must be positive"; return -1; }
  if (out_alpha && kernel != GPG_KERNEL_RATQU) { g_create_err = "There are no kernel hyperparameters for this kernel"; return -1; }   // KernelSqExp.py:125-127
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { g_create_err = "no HIP device visible"; return -2; }
  if (device < 0 || device >= ndev) { g_create_err = "device index out of range"; return -1; }
  const size_t N = use_grad ? (size_t)n * (dim + 1) : (size_t)n, nrt = (size_t)dim * n * n;
  double *d_rt = nullptr, *d_th = nullptr, *d_al = nullptr;
  hipStream_t st = nullptr;
  int rc = 0;
  auto fail = [&](const char* what, hipError_t e) { g_create_err = std::string(what) + ": " + hipGetErrorString(e); rc = -2; };
  hipError_t e;
  if ((e = hipSetDevice(device)) != hipSuccess) fail("hipSetDevice", e);
  if (!rc && (e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking)) != hipSuccess) fail("hipStreamCreate", e);
  if (!rc && (e = hipMalloc(&d_rt, sizeof(double) * nrt)) != hipSuccess) fail("hipMalloc(rtensor)", e);
  if (!rc && out_theta && (e = hipMalloc(&d_th, sizeof(double) * dim * N * N)) != hipSuccess) fail("hipMalloc(out_theta)", e);
  if (!rc && out_alpha && (e = hipMalloc(&d_al, sizeof(double) * N * N)) != hipSuccess) fail("hipMalloc(out_alpha)", e);
  if (!rc && (e = hipMemcpyAsync(d_rt, rtensor, sizeof(double) * nrt, hipMemcpyHostToDevice, st)) != hipSuccess) fail("copy rtensor", e);
  if (!rc && gpg_kern_rtensor_dhp_run(kernel, dim, n, use_grad ? 1 : 0, theta, hp_kernel, d_rt, d_th, d_al, st) != 0) {
    g_create_err = "rtensor derivative kernel launch failed"; rc = -2;
  }
  if (!rc && out_theta && (e = hipMemcpyAsync(out_theta, d_th, sizeof(double) * dim * N * N, hipMemcpyDeviceToHost, st)) != hipSuccess) fail("copy out_theta", e);
  if (!rc && out_alpha && (e = hipMemcpyAsync(out_alpha, d_al, sizeof(double) * N * N, hipMemcpyDeviceToHost, st)) != hipSuccess) fail("copy out_alpha", e);
  if (!rc && (e = hipStreamSynchronize(st)) != hipSuccess) fail("hipStreamSynchronize", e);
  if (d_rt) (void)hipFree(d_rt);
  if (d_th) (void)hipFree(d_th);
  if (d_al) (void)hipFree(d_al);
  if (st) (void)hipStreamDestroy(st);
  return rc;
}

int gpg_kern_rtensor_hess_x(int device, int kernel, int dim, int n1, int n2, const double* rtensor, const double* theta, double hp_kernel,
                            int use_grad, const unsigned char* use_grad2, double* out) {
  if (!rtensor || !theta || !out || dim < 1 || dim > GPG_MAX_DIM || n1 < 1 || n2 < 1) { g_create_err = "bad gpg_kern_rtensor_hess_x arguments (1 <= dim <= 16)"; return -1; }
  if (kernel != GPG_KERNEL_SQEXP && kernel != GPG_KERNEL_MA5F2 && kernel != GPG_KERNEL_RATQU) { g_create_err = "unknown kernel id"; return -1; }
  if (kernel == GPG_KERNEL_RATQU && !(hp_kernel > 0.0)) { g_create_err = "hp_kernel (alpha of RatQu) must be positive"; return -1; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { g_create_err = "no HIP device visible"; return -2; }
  if (device < 0 || device >= ndev) { g_create_err = "device index out of range"; return -1; }
  std::vector<int> gp2(n2, -1);
  int n2g = 0;
  if (use_grad)
    for (int b = 0; b < n2; ++b) if (!use_grad2 || use_grad2[b]) gp2[b] = n2g++;
  const size_t Ccols = (size_t)n2 + (size_t)n2g * dim, nout = (size_t)dim * n1 * dim * Ccols, nrt = (size_t)dim * n1 * n2;
  double *d_rt = nullptr, *d_out = nullptr;
  int* d_g2 = nullptr;
  hipStream_t st = nullptr;
  int rc = 0;
  auto fail = [&](const char* what, hipError_t e) { g_create_err = std::string(what) + ": " + hipGetErrorString(e); rc = -2; };
  hipError_t e;
  if ((e = hipSetDevice(device)) != hipSuccess) fail("hipSetDevice", e);
  if (!rc && (e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking)) != hipSuccess) fail("hipStreamCreate", e);
  if (!rc && (e = hipMalloc(&d_rt, sizeof(double) * nrt)) != hipSuccess) fail("hipMalloc(rtensor)", e);
  if (!rc && (e = hipMalloc(&d_out, sizeof(double) * nout)) != hipSuccess) fail("hipMalloc(out)", e);
  if (!rc && (e = hipMalloc(&d_g2, sizeof(int) * n2)) != hipSuccess) fail("hipMalloc(gpos2)", e);
  if (!rc && (e = hipMemcpyAsync(d_rt, rtensor, sizeof(double) * nrt, hipMemcpyHostToDevice, st)) != hipSuccess) fail("copy rtensor", e);
  if (!rc && (e = hipMemcpyAsync(d_g2, gp2.data(), sizeof(int) * n2, hipMemcpyHostToDevice, st)) != hipSuccess) fail("copy gpos2", e);
  if (!rc && (e = hipMemsetAsync(d_out, 0, sizeof(double) * nout, st)) != hipSuccess) fail("memset out", e);
  if (!rc && gpg_kern_rtensor_hess_x_run(kernel, dim, n1, n2, n2g, use_grad ? 1 : 0, theta, hp_kernel, d_rt, d_g2, d_out, st) != 0) {
    g_create_err = "rtensor x-derivative kernel launch failed"; rc = -2;
  }
  if (!rc && (e = hipMemcpyAsync(out, d_out, sizeof(double) * nout, hipMemcpyDeviceToHost, st)) != hipSuccess) fail("copy out", e);
  if (!rc && (e = hipStreamSynchronize(st)) != hipSuccess) fail("hipStreamSynchronize", e);
  if (d_rt) (void)hipFree(d_rt);
  if (d_out) (void)hipFree(d_out);
  if (d_g2) (void)hipFree(d_g2);
  if (st) (void)hipStreamDestroy(st);
  return rc;
}

int gpg_abs_rowsum(gpg_ctx* c, const gpg_hp* hp, double* rowsum) {
  int rc = check_hp(c, hp);
  if (rc) return rc;
  if (!rowsum) { c->err = "rowsum is NULL"; return -1; }
  GPG_HIP_OK(c, hipSetDevice(c->device));
  GPG_WS(c, 0);
  if (!c->apply_buf) GPG_HIP_OK(c, hipMalloc(&c->apply_buf, sizeof(double) * 2 * (size_t)c->vec_rows_cols));
  gpg_hp h0 = *hp;
  h0.eta = 0.0;
  const bool precon = hp->wellcond == GPG_WELLCOND_PRECON;
  AsmParams p = make_params(c, &h0, precon ? 0 : 1);      // Kcor (Kernel.py:227, times varK) or the raw kernel matrix (:272)
  c->last_precon = p.precon;
  c->prep_valid = true;
  c->factor_valid = false;
  gpg_launch_prep(c, p, hp->var_fval, hp->var_fgrad, 0.0, 0.0, 0.0, 0.0);
  unsigned save = c->prof_mask; c->prof_mask = 0;
  gpg_launch_assembly(c, p);
  c->prof_mask = save;
  gpg_launch_abs_rowsum(c, precon ? 1.0 / hp->varK_mat : 1.0, c->apply_buf);
  GPG_HIP_OK(c, hipMemcpyAsync(rowsum, c->apply_buf, sizeof(double) * c->N, hipMemcpyDeviceToHost, c->stream));
  GPG_HIP_OK(c, hipStreamSynchronize(c->stream));
  GPG_HIP_OK(c, hipGetLastError());
  GPG_LAUNCH_OK(c);
  return 0;
}

int gpg_lkd_alpha(gpg_ctx* c, double* alpha) {
  if (!c) return -1;
  if (!alpha) { c->err = "alpha is NULL"; return -1; }
  GPG_HIP_OK(c, hipSetDevice(c->device));
  GPG_WS(c, 0);
  if (!c->alpha_valid) { c->err = "gpg_lkd_alpha follows a successful gpg_lkd_grad (no other call on the context in between)"; return -1; }
  gpg_launch_alpha(c, c->tmpv);
  GPG_HIP_OK(c, hipMemcpyAsync(alpha, c->tmpv, sizeof(double) * c->N, hipMemcpyDeviceToHost, c->stream));
  GPG_HIP_OK(c, hipStreamSynchronize(c->stream));
  GPG_HIP_OK(c, hipGetLastError());
  return 0;
}

int gpg_set_gradient_nugget(gpg_ctx* c, double eta) {
  if (!c) return -1;
  c->grad_eta = eta >= 0.0 ? eta : -1.0;
  return 0;
}

int gpg_dcov_quadform(gpg_ctx* c, const gpg_hp* hp, const double* v, double* out) {
  int rc = check_hp(c, hp);
  if (rc) return rc;
  if (!v || !out) { c->err = "v / out is NULL"; return -1; }
  GPG_HIP_OK(c, hipSetDevice(c->device));
  GPG_WS(c, c->last_factor_ws);
  if (!c->prep_valid) { c->err = "gpg_dcov_quadform follows a gpg_lkd / gpg_setup_eval call with the same hp"; return -1; }
  const int nblk = gpg_grad_partial_blocks(c);
  const int nval = 2 * GPG_GRAD_SLOTS_MAX;
  if (!c->gpartial) GPG_HIP_OK(c, hipMalloc(&c->gpartial, sizeof(double) * (size_t)nval * (nblk + 1)));
  if (!c->apply_buf) GPG_HIP_OK(c, hipMalloc(&c->apply_buf, sizeof(double) * 2 * (size_t)c->vec_rows_cols));
  double* dv = c->apply_buf;
  double* dz = c->apply_buf + c->vec_rows_cols;
  GPG_HIP_OK(c, hipMemcpyAsync(dv, v, sizeof(double) * c->N, hipMemcpyHostToDevice, c->stream));
  gpg_launch_unscale(c, dv, dz);                          // the contraction kernel forms "alpha" = z * invp = v
  AsmParams p = make_params(c, hp, 0);
  double* res = c->gpartial + (size_t)nval * nblk;
  gpg_launch_grad_contract(c, p, c->gpartial, res, dz, nullptr);
  const int ns = c->d + 3 + (c->kernel == GPG_KERNEL_RATQU ? 1 : 0);
  std::vector<double> h(ns);
  GPG_HIP_OK(c, hipMemcpyAsync(h.data(), res, sizeof(double) * ns, hipMemcpyDeviceToHost, c->stream));
  GPG_HIP_OK(c, hipStreamSynchronize(c->stream));
  GPG_HIP_OK(c, hipGetLastError());
  GPG_LAUNCH_OK(c);
  out[c->d + 3] = 0.0;
  for (int k = 0; k < ns; ++k) out[k] = h[k];
  return 0;
}

int gpg_cond_fro(gpg_ctx* c, const gpg_hp* hp, double* cond, double* cond_grad) {
  int rc = check_hp(c, hp);
  if (rc) return rc;
  if (!cond) { c->err = "cond is NULL"; return -1; }
  if (cond_grad && hp->wellcond == GPG_WELLCOND_PRECON) {
    c->err = "Not setup to calculate the gradient of the condition number if wellcond_mtd = \"precon\" (GpHparaCon.py:249)";
    return -1;
  }
  GPG_HIP_OK(c, hipSetDevice(c->device));
  GPG_WS(c, 0);
  const size_t nn = (size_t)c->Npad * c->Npad;
  if (!c->Wfull) {
    GPG_HIP_OK(c, hipMalloc(&c->Wfull, sizeof(double) * nn));
    GPG_HIP_OK(c, hipMalloc(&c->Minv, sizeof(double) * nn));
  }
  if (cond_grad && !c->Kbuf) {
    GPG_HIP_OK(c, hipMalloc(&c->Kbuf, sizeof(double) * nn));
    GPG_HIP_OK(c, hipMalloc(&c->Tbuf, sizeof(double) * nn));
  }
  const int nblk = gpg_grad_partial_blocks(c);
  const int nval = 2 * GPG_GRAD_SLOTS_MAX;
  if (!c->gpartial) GPG_HIP_OK(c, hipMalloc(&c->gpartial, sizeof(double) * (size_t)nval * (nblk + 1)));
  if (!c->apply_buf) GPG_HIP_OK(c, hipMalloc(&c->apply_buf, sizeof(double) * 2 * (size_t)c->vec_rows_cols));
  double* colsum = c->apply_buf;                       // [N] per-column partial sums of the two norms
  double* norms = c->scal + 8;                         // scal slot 1: ||K||_F^2, ||K^-1||_F^2
  AsmParams p = make_params(c, hp, 0);
  c->last_precon = p.precon;
  c->prep_valid = true;
  c->last_factor_ws = 0;
  GPG_HIP_OK(c, hipMemsetAsync(c->info, 0, sizeof(int), c->stream));
  gpg_launch_prep(c, p, hp->var_fval, hp->var_fgrad, 0.0, 0.0, 0.0, 0.0);
  gpg_launch_assembly(c, p);                           // the matrix that is factorised: Kcov_precon / Kcov (Kernel.py:240, 280)
  gpg_launch_frob_lower(c, c->A, c->ld, colsum, norms);
  if (cond_grad)                                       // keep K: the factorisation overwrites it (columns of ld doubles -> Npad doubles)
    GPG_HIP_OK(c, hipMemcpy2DAsync(c->Kbuf, sizeof(double) * c->Npad, c->A, sizeof(double) * c->ld, sizeof(double) * c->Npad, c->Npad,
                                   hipMemcpyDeviceToDevice, c->stream));
  gpg_cholesky(c);
  GPG_HIP_OK(c, hipMemcpyAsync(c->h_info, c->info, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  GPG_HIP_OK(c, hipStreamSynchronize(c->stream));
  GPG_HIP_OK(c, hipGetLastError());
  GPG_LAUNCH_OK(c);
  if (internal_failure(c, c->h_info, 1)) return -4;
  c->factor_valid = (c->h_info[0] == 0);
  if (c->h_info[0] != 0) return c->h_info[0];
  gpg_inverse_from_factor(c, c->Wfull, c->Minv);       // Minv = -(L L^T)^-1, lower
  gpg_launch_frob_lower(c, c->Minv, c->Npad, colsum, norms + 1);
  double h_norm[2] = {0.0, 0.0};
  GPG_HIP_OK(c, hipMemcpyAsync(h_norm, norms, sizeof(double) * 2, hipMemcpyDeviceToHost, c->stream));
  GPG_HIP_OK(c, hipMemcpyAsync(c->h_info, c->info, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  GPG_HIP_OK(c, hipStreamSynchronize(c->stream));
  GPG_HIP_OK(c, hipGetLastError());
  GPG_LAUNCH_OK(c);
  if (solve_failure(c)) return -4;
  const double nK = sqrt(h_norm[0]), nKi = sqrt(h_norm[1]);
  *cond = nK * nKi;                                    // GpHparaCon.py:217-220
  if (!cond_grad) return 0;
  // d cond / d K = frac K - K^-3 / frac, frac = ||K^-1||_F / ||K||_F (GpHparaCon.py:232-234), contracted with d K / d hp_k.
  // With S = -K^-1 (full):  T = -S S^T = -K^-2,  M2 = -T S^T = -K^-3 (lower).
  gpg_launch_symmetrize(c, c->Minv, c->Npad);                                    // S
  if (!gpg_launch_full_abt(c, c->Minv, nullptr, c->Tbuf)) { c->err = "full product launch failed"; return -2; }
  gpg_launch_symmetrize(c, c->Tbuf, c->Npad);                                    // T
  if (!gpg_launch_full_abt(c, c->Tbuf, c->Minv, c->Wfull)) { c->err = "full product launch failed"; return -2; }   // M2 -> Wfull (lower)
  GPG_HIP_OK(c, hipMemsetAsync(c->zvec, 0, sizeof(double) * c->Npad, c->stream));   // the alpha-alpha^T half of the contraction is not used
  double* res = c->gpartial + (size_t)nval * nblk;
  const int ns = c->d + 3 + (c->kernel == GPG_KERNEL_RATQU ? 1 : 0);
  std::vector<double> h1(2 * ns), h2(2 * ns);
  gpg_launch_grad_contract(c, p, c->gpartial, res, c->zvec, c->Kbuf);            // g_inv = sum G_k o (K / 2)
  GPG_HIP_OK(c, hipMemcpyAsync(h1.data(), res, sizeof(double) * 2 * ns, hipMemcpyDeviceToHost, c->stream));
  GPG_HIP_OK(c, hipStreamSynchronize(c->stream));
  gpg_launch_grad_contract(c, p, c->gpartial, res, c->zvec, c->Wfull);           // g_inv = sum G_k o (-K^-3 / 2)
  GPG_HIP_OK(c, hipMemcpyAsync(h2.data(), res, sizeof(double) * 2 * ns, hipMemcpyDeviceToHost, c->stream));
  GPG_HIP_OK(c, hipStreamSynchronize(c->stream));
  GPG_HIP_OK(c, hipGetLastError());
  GPG_LAUNCH_OK(c);
  const double frac = nKi / nK;
  for (int k = 0; k < c->d + 4; ++k) cond_grad[k] = 0.0;
  for (int k = 0; k < ns; ++k) cond_grad[k] = 2.0 * frac * h1[ns + k] + (2.0 / frac) * h2[ns + k];
  return 0;
}

int gpg_factor_apply(gpg_ctx* c, int op, const double* v, double* out) {
  if (!c) return -1;
  if (!v || !out || (op != 0 && op != 1)) { c->err = "bad factor_apply arguments"; return -1; }
  GPG_HIP_OK(c, hipSetDevice(c->device));
  GPG_WS(c, c->last_factor_ws);
  if (!c->factor_valid) { c->err = "no valid factor on the device (gpg_lkd / gpg_setup_eval first)"; return -1; }
  if (!c->apply_buf) GPG_HIP_OK(c, hipMalloc(&c->apply_buf, sizeof(double) * 2 * (size_t)c->vec_rows_cols));
  double* dv = c->apply_buf;
  double* dout = c->apply_buf + c->vec_rows_cols;
  GPG_HIP_OK(c, hipMemsetAsync(dv, 0, sizeof(double) * c->Npad, c->stream));
  GPG_HIP_OK(c, hipMemcpyAsync(dv, v, sizeof(double) * c->N, hipMemcpyHostToDevice, c->stream));
  GPG_HIP_OK(c, hipMemsetAsync(c->info, 0, sizeof(int), c->stream));
  if (gpg_factor_apply_dev(c, op, dv, dout) != 0) { c->err = "factor_apply launch failed"; return -2; }
  GPG_HIP_OK(c, hipMemcpyAsync(out, dout, sizeof(double) * c->N, hipMemcpyDeviceToHost, c->stream));
  GPG_HIP_OK(c, hipMemcpyAsync(c->h_info, c->info, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  GPG_HIP_OK(c, hipStreamSynchronize(c->stream));
  GPG_HIP_OK(c, hipGetLastError());
  GPG_LAUNCH_OK(c);
  if (solve_failure(c)) return -4;
  return 0;
}

int gpg_prof_enable(gpg_ctx* c, unsigned mask) {
  if (!c) return -1;
  c->prof_mask = mask;
  return 0;
}

int gpg_prof_read(gpg_ctx* c, double ms[GPG_PROF_NCAT], long long count[GPG_PROF_NCAT], double work[GPG_PROF_NCAT]) {
  if (!c) return -1;
  GPG_HIP_OK(c, hipSetDevice(c->device));
  GPG_HIP_OK(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < GPG_PROF_NCAT; ++i) { ms[i] = 0.0; count[i] = 0; work[i] = c->prof_work[i]; c->prof_work[i] = 0.0; }
  for (auto& pe : c->prof_pending) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, pe.e0, pe.e1) == hipSuccess) { ms[pe.cat] += t; count[pe.cat] += 1; }
    c->prof_pool.push_back({pe.e0, pe.e1});
  }
  c->prof_pending.clear();
  return 0;
}

int gpg_set_factor_mode(gpg_ctx* c, int mode) {
  if (!c) return -1;
  switch (mode) {
    case GPG_FACTOR_AUTO:    c->chol_impl = 1; c->tail_cols = 12288; break;
    case GPG_FACTOR_BLOCKED: c->chol_impl = 0; c->tail_cols = 0; break;
    case GPG_FACTOR_TILE64:  c->chol_impl = 0; c->tail_cols = 1 << 30; break;
    case GPG_FACTOR_TILE128: c->chol_impl = 1; c->tail_cols = 0; break;
    default: c->err = "unknown factor mode"; return -1;
  }
  c->solve_dataflow = 1;                                    // re-arm what a timed-out wait switched off (with_fallback)
  if (c->overlap_inverse == 0 && c->overlap_fallbacks > 0) c->overlap_inverse = 2;
  return 0;
}

int gpg_set_pair_mode(gpg_ctx* c, int mode) {
  if (!c) return -1;
  if (mode < 0 || mode > 2) { c->err = "pair mode must be 0, 1 or 2"; return -1; }
  c->pair_mode = mode;
  return 0;
}

int gpg_reserve_batch(gpg_ctx* c, int rows) {
  if (!c) return -1;
  if (!c->have_data) { c->err = "gpg_set_data must be called first"; return -1; }
  GPG_HIP_OK(c, hipSetDevice(c->device));
  return batch_plan(c, rows) >= 1 ? 0 : -2;
}

int gpg_set_batch(gpg_ctx* c, int max_matrices) {
  if (!c) return -1;
  if (max_matrices < -1 || max_matrices > 64) { c->err = "batch size must be in [-1, 64]"; return -1; }
  c->batch_max = max_matrices;
  return 0;
}

int gpg_set_max_workgroups(gpg_ctx* c, int n) {
  if (!c) return -1;
  if (n < 0) { c->err = "max_workgroups must be >= 0 (0 = as many as the device holds)"; return -1; }
  c->max_workgroups = n;
  return 0;
}

int gpg_set_lookahead(gpg_ctx* c, int on) {
  if (!c) return -1;
  c->lookahead = (on & 1) ? 1 : 0;
  c->gemm_impl = (on & 2) ? 0 : 1;   // bit 1: fall back to the register-staged 128x128 kernel (A/B runs)
  c->panel_impl = (on & 4) ? 0 : 1;  // bit 2: fall back to per-64-column trsm + gemm launches for B_p (A/B runs)
  return 0;
}

int gpg_set_panel(gpg_ctx* c, int nb_outer) {
  if (!c) return -1;
  if (nb_outer < 128 || nb_outer > 1024 || nb_outer % 128) { c->err = "panel width must be a multiple of 128 in [128, 1024]"; return -1; }
  c->nb_outer = nb_outer;
  return 0;
}

int gpg_device_info(int device, char* buf, int buflen) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return -2;
  snprintf(buf, buflen, "%s %s CUs=%d clock=%dkHz mem=%.1fGiB", prop.name, prop.gcnArchName, prop.multiProcessorCount,
           prop.clockRate, prop.totalGlobalMem / 1073741824.0);
  return 0;
}

}  // extern "C"
