// gpg_multi_*: the restart table of one multi-start sharded over several devices of ONE process (SURVEY.md 8b / 8e;
// reference loop GpHparaX0.py:33-59).  One gpg_ctx per device, one host thread per context, contiguous row blocks
// (the partition of gpgradpy_amd/multistart.py::shard_rows), results gathered in host memory, nanargmax on the host.
// No collective library: inside one process the "all_gather" is a join.  The torch.distributed path
// (one process per GPU, RCCL) is the deployment of DESIGN.md section 5; this is the same partition for callers below Python.
#include <cmath>
#include <string>
#include <thread>
#include <vector>
#include "gpg_internal.h"

struct gpg_multi {
  std::vector<gpg_ctx*> ctx;
  std::vector<int> device;
  std::string err;
};

static thread_local std::string g_multi_create_err;

extern "C" {

int gpg_multi_create(gpg_multi** out, int ndev, const int* devices, int n_eval, int dim, int use_grad, int kernel) {
  if (!out) return -1;
  *out = nullptr;
  if (ndev < 1 || ndev > 64) { g_multi_create_err = "1 <= ndev <= 64 required"; return -1; }
  gpg_multi* m = new gpg_multi();
  for (int i = 0; i < ndev; ++i) {
    gpg_ctx* c = nullptr;
    const int dev = devices ? devices[i] : i;
    const int rc = gpg_create(&c, dev, n_eval, dim, use_grad, kernel);
    if (rc != 0) {
      g_multi_create_err = std::string("device ") + std::to_string(dev) + ": " + gpg_last_error(nullptr);
      for (gpg_ctx* p : m->ctx) gpg_destroy(p);
      delete m;
      return rc;
    }
    m->ctx.push_back(c);
    m->device.push_back(dev);
  }
  *out = m;
  return 0;
}

void gpg_multi_destroy(gpg_multi* m) {
  if (!m) return;
  for (gpg_ctx* c : m->ctx) gpg_destroy(c);
  delete m;
}

const char* gpg_multi_last_error(const gpg_multi* m) { return m ? m->err.c_str() : g_multi_create_err.c_str(); }

int gpg_multi_count(const gpg_multi* m) { return m ? (int)m->ctx.size() : -1; }

int gpg_multi_set_data(gpg_multi* m, const double* x, const double* data_vec, const double* noise_var) {
  if (!m) return -1;
  for (size_t i = 0; i < m->ctx.size(); ++i) {       // the data set is replicated: <= 1.1 MB even at n=4000, d=16
    const int rc = gpg_set_data(m->ctx[i], x, data_vec, noise_var);
    if (rc != 0) { m->err = std::string("device ") + std::to_string(m->device[i]) + ": " + gpg_last_error(m->ctx[i]); return rc; }
  }
  return 0;
}

int gpg_multi_lkd_batch(gpg_multi* m, int nrows, const double* hp_rows, int row_len, double eta, int wellcond,
                        int closed_form_varK, gpg_lkd_out* out, int* best) {
  if (!m) return -1;
  if (nrows < 1 || !hp_rows || !out) { m->err = "bad batch arguments"; return -1; }
  const int G = (int)m->ctx.size();
  std::vector<int> rc(G, 0);
  std::vector<std::thread> th;
  const int base = nrows / G, rem = nrows % G;
  for (int g = 0; g < G; ++g) {
    const int lo = g * base + (g < rem ? g : rem), cnt = base + (g < rem ? 1 : 0);
    if (cnt == 0) continue;
    th.emplace_back([=, &rc] {
      rc[g] = gpg_lkd_batch(m->ctx[g], cnt, hp_rows + (size_t)lo * row_len, row_len, eta, wellcond, closed_form_varK, out + lo);
    });
  }
  for (auto& t : th) t.join();
  for (int g = 0; g < G; ++g)
    if (rc[g] != 0) { m->err = std::string("device ") + std::to_string(m->device[g]) + ": " + gpg_last_error(m->ctx[g]); return rc[g]; }
  if (best) {       // nanargmax (GpHparaX0.py:58): rows whose factorisation failed are skipped
    int b = -1;
    for (int i = 0; i < nrows; ++i)
      if (out[i].info == 0 && !std::isnan(out[i].ln_lkd) && (b < 0 || out[i].ln_lkd > out[b].ln_lkd)) b = i;
    *best = b;
  }
  return 0;
}

}  // extern "C"
