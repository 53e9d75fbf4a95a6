// Reductions and triangular sweeps around the factorisation (gfx950).
//   lkd_reduce      : ln det, GLS mean beta, r' K^-1 r from the diagonal of L and the two RHS rows
//                     (reference GpMeanFun.py:102-108, CalcLkd.py:153-168 / 220-226) -- wave-reduced
//   backward solve  : z = L^-T w for the single vector needed by alpha (GpEvalModel.py:57)
//   predict_reduce  : mu = beta + Kyx' alpha, sig2 = 1 - diag(Kxy K^-1 Kyx) (GpEvalModel.py:162-168)
//   extract         : dense N x N copies on request (tests / drop-in 7-tuple)
#include <algorithm>
#include "gpg_internal.h"

namespace {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// block-wide sum of NV values per thread; result valid in every thread
template <int NV>
__device__ void block_sum(double (&v)[NV], double* sh /* [NV * 16] */) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    double s = wave_sum(v[q]);
    if (lane == 0) sh[q * 16 + w] = s;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    double s = 0.0;
    for (int i = 0; i < nw; ++i) s += sh[q * 16 + i];
    v[q] = s;
  }
  __syncthreads();
}

// scal[0] = ln_det, [1] = beta, [2] = r'K^-1 r, [3] = V'K^-1 V, [4] = V'K^-1 y, [7] = the factorisation's info word (so that the
// host fetches ONE buffer per evaluation)
__global__ void __launch_bounds__(1024) lkd_reduce_kernel(const double* __restrict__ A, int ld, int N, int Npad,
                                                          const double* __restrict__ dvec, int precon,
                                                          double* __restrict__ scal, size_t v_stride, size_t a_stride,
                                                          const int* __restrict__ info) {
  A += blockIdx.x * a_stride; dvec += blockIdx.x * v_stride; scal += blockIdx.x * 8;   // batched: one workgroup per matrix
  __shared__ double sh[4 * 16];
  double v[4] = {0.0, 0.0, 0.0, 0.0};
  for (int c = threadIdx.x; c < N; c += blockDim.x) {
    const size_t col = (size_t)c * ld;
    v[0] += log(A[col + c]);
    if (precon) v[1] += 0.5 * log(dvec[c]);
    const double w1 = A[col + Npad], w2 = A[col + Npad + 1];
    v[2] += w1 * w1;
    v[3] += w1 * w2;
  }
  block_sum<4>(v, sh);
  const double beta = v[3] / v[2];
  double rr[1] = {0.0};
  for (int c = threadIdx.x; c < N; c += blockDim.x) {
    const size_t col = (size_t)c * ld;
    const double t = A[col + Npad + 1] - beta * A[col + Npad];
    rr[0] += t * t;
  }
  block_sum<1>(rr, sh);
  if (threadIdx.x == 0) {
    scal[0] = 2.0 * (v[0] + v[1]);
    scal[1] = beta;
    scal[2] = rr[0];
    scal[3] = v[2];
    scal[4] = v[3];
    scal[7] = (double)info[blockIdx.x];
  }
}

// rows[64 c + r] = (r == 0) ? src[c ld] : 0 -- a vector as row 0 of a zeroed 64-row tile row (one launch instead of a fill and a
// strided copy)
__global__ void __launch_bounds__(256) vec_rows_load_kernel(double* __restrict__ rows, const double* __restrict__ src, int ld, int Npad) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= 64 * (size_t)Npad) return;
  rows[i] = (i & 63) == 0 ? src[(i >> 6) * (size_t)ld] : 0.0;
}

// t[j] = sum_{i >= k0+64} L[i, k0+j] z[i]   (one workgroup per column j)
__global__ void __launch_bounds__(256) bs_dot_kernel(const double* __restrict__ A, int ld, int Npad, int k0,
                                                     const double* __restrict__ z, double* __restrict__ t) {
  __shared__ double sh[16];
  const int j = blockIdx.x;
  const double* colp = A + (size_t)(k0 + j) * ld;
  double v[1] = {0.0};
  for (int i = k0 + 64 + threadIdx.x; i < Npad; i += 256) v[0] += colp[i] * z[i];
  block_sum<1>(v, sh);
  if (threadIdx.x == 0) t[j] = v[0];
}

// solve L_kk^T z_k = w_k - t (one wave, lane j <-> unknown j)
__global__ void __launch_bounds__(64) bs_tri_kernel(const double* __restrict__ A, int ld, int Npad, int k0,
                                                    const double* __restrict__ t, int has_t, double* __restrict__ z) {
  __shared__ double Ls[64][65];
  const int j = threadIdx.x;
  for (int q = 0; q < 64; ++q) Ls[j][q] = A[(size_t)(k0 + j) + (size_t)(k0 + q) * ld];  // Ls[i][q] = L[i][q]
  __syncthreads();
  // right-hand side: forward-solved RHS row 0
  double rhs = A[(size_t)(k0 + j) * ld + Npad] - (has_t ? t[j] : 0.0);
  double zj = 0.0;
  for (int i = 63; i >= 0; --i) {
    double zi = __shfl(rhs, i, 64) / Ls[i][i];
    if (j == i) zj = zi;
    if (j < i) rhs -= Ls[i][j] * zi;
  }
  z[k0 + j] = zj;
}

__global__ void alpha_kernel(const double* __restrict__ z, const double* __restrict__ invp, int N,
                             double* __restrict__ alpha) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < N) alpha[r] = z[r] * invp[r];
}

// Neumaier's compensated accumulation: s + comp carries the sum to ~eps |result| + eps^2 sum |terms|.  The mean
// mu = beta + sum_c Wt[j, c] z[c] cancels by a factor ~kappa on ill-conditioned cases (|alpha| ~ kappa |mu|), so the
// order of a plain summation shows up at eps kappa (measured 4e-8 .. 2e-7 at kappa = 7e9 depending on the slicing).
__device__ __forceinline__ void kb_add(double& s, double& comp, double x) {
  const double t = s + x;
  comp += (fabs(s) >= fabs(x)) ? (s - t) + x : (x - t) + s;
  s = t;
}

// phase 0: out[j] = beta + sum_c Wt[j, c] z[c] ; phase 1: out[j] = 1 - sum_c Wt[j, c]^2
__global__ void __launch_bounds__(1024) predict_reduce_kernel(const double* __restrict__ Wt, int nxp, int N,
                                                              const double* __restrict__ z, double beta, int phase,
                                                              double* __restrict__ out) {
  __shared__ double sh[16][64];
  const int jl = threadIdx.x & 63, cg = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + jl;
  double acc = 0.0, comp = 0.0;
  if (phase == 0) {
    for (int c = cg; c < N; c += 16) kb_add(acc, comp, Wt[(size_t)c * nxp + j] * z[c]);
  } else {
    for (int c = cg; c < N; c += 16) { double w = Wt[(size_t)c * nxp + j]; acc += w * w; }
  }
  sh[cg][jl] = acc + comp;
  __syncthreads();
  if (cg == 0) {
    double s = 0.0, cs = 0.0;
    for (int q = 0; q < 16; ++q) kb_add(s, cs, sh[q][jl]);
    s += cs;
    out[j] = phase == 0 ? beta + s : 1.0 - s;
  }
}

// The same reductions for a few query rows, spread over S slices of the columns (grid = (nxp / 64, S)) so that a
// single-point evaluation is not one workgroup streaming the whole row set; the partial sums are added in slice order
// by predict_final_kernel (deterministic).
__global__ void __launch_bounds__(1024) predict_partial_kernel(const double* __restrict__ Wt, int nxp, int N, int chunk,
                                                               const double* __restrict__ z, int phase,
                                                               double* __restrict__ partial) {
  __shared__ double sh[16][64];
  const int jl = threadIdx.x & 63, cg = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + jl;
  const int c0 = blockIdx.y * chunk, c1 = min(N, c0 + chunk);
  double acc = 0.0, comp = 0.0;
  if (phase == 0) {
    for (int c = c0 + cg; c < c1; c += 16) kb_add(acc, comp, Wt[(size_t)c * nxp + j] * z[c]);
  } else {
    for (int c = c0 + cg; c < c1; c += 16) { double w = Wt[(size_t)c * nxp + j]; acc += w * w; }
  }
  sh[cg][jl] = acc + comp;
  __syncthreads();
  if (cg == 0) {
    double s = 0.0, cs = 0.0;
    for (int q = 0; q < 16; ++q) kb_add(s, cs, sh[q][jl]);
    partial[(size_t)blockIdx.y * nxp + j] = s + cs;
  }
}

__global__ void __launch_bounds__(1024) predict_final_kernel(const double* __restrict__ partial, int nxp, int S, double beta,
                                                             int phase, double* __restrict__ out) {
  __shared__ double sh[16][64];
  const int jl = threadIdx.x & 63, cg = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + jl;
  double acc = 0.0, comp = 0.0;
  for (int q = cg; q < S; q += 16) kb_add(acc, comp, partial[(size_t)q * nxp + j]);
  sh[cg][jl] = acc + comp;
  __syncthreads();
  if (cg == 0) {
    double s = 0.0, cs = 0.0;
    for (int q = 0; q < 16; ++q) kb_add(s, cs, sh[q][jl]);
    s += cs;
    out[j] = phase == 0 ? beta + s : 1.0 - s;
  }
}

// ---- multi right-hand-side backward solve on the "RHS rows" layout Z[r + i * ldz] (in place) ----------
// t[r * 64 + j] = sum_{i >= k0+64} L[i, k0+j] Z[r, i]   (workgroup = column j x a group of 16 right-hand sides)
__global__ void __launch_bounds__(256) bs_dot_multi_kernel(const double* __restrict__ A, int ld, int Npad, int k0,
                                                           const double* __restrict__ Z, int ldz, int nrhs,
                                                           double* __restrict__ t) {
  __shared__ double sh[16 * 16];
  const int j = blockIdx.x, r0 = blockIdx.y * 16;
  const double* colp = A + (size_t)(k0 + j) * ld;
  double v[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) v[q] = 0.0;
  for (int i = k0 + 64 + threadIdx.x; i < Npad; i += 256) {
    const double l = colp[i];
    const double* zp = Z + (size_t)i * ldz + r0;
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] += l * zp[q];
  }
  block_sum<16>(v, sh);
  if (threadIdx.x < 16 && r0 + threadIdx.x < nrhs) t[(size_t)(r0 + threadIdx.x) * 64 + j] = v[threadIdx.x];
}

// solve L_kk^T z_k = w_k - t for right-hand side r = blockIdx.x (one wave, lane j <-> unknown j)
__global__ void __launch_bounds__(64) bs_tri_multi_kernel(const double* __restrict__ A, int ld, int k0,
                                                          const double* __restrict__ t, int has_t,
                                                          double* __restrict__ Z, int ldz) {
  __shared__ double Ls[64][65];
  const int j = threadIdx.x, r = blockIdx.x;
  for (int q = 0; q < 64; ++q) Ls[j][q] = A[(size_t)(k0 + j) + (size_t)(k0 + q) * ld];
  __syncthreads();
  double rhs = Z[(size_t)(k0 + j) * ldz + r] - (has_t ? t[(size_t)r * 64 + j] : 0.0);
  double zj = 0.0;
  for (int i = 63; i >= 0; --i) {
    double zi = __shfl(rhs, i, 64) / Ls[i][i];
    if (j == i) zj = zi;
    if (j < i) rhs -= Ls[i][j] * zi;
  }
  Z[(size_t)(k0 + j) * ldz + r] = zj;
}

// Posterior gradient reductions (reference GpEvalModel.py:135-140, 319-354): for query q and direction j'
//   g1[q, j'] = sum_c dKxy_dx[(j', q), c] alpha_c              -> dmu/dx
//   g2[q, j'] = sum_c dKxy_dx[(j', q), c] (K^-1 Kyx)[c, q]     -> d sig/dx = -varK g2 / sig
// dKxy_dx[(j', q), c] = block (I, j'+1) of the gradient-enhanced kernel at (x_a, xq_q), c = (I, a); it is
// recomputed on the fly (never stored).  zvec = p * alpha, Z = L^-T L^-1 P^-1 Kyx (RHS-rows layout).
template <int KERN, int D>
__global__ void __launch_bounds__(256) cross_grad_kernel(AsmParams P, const double* __restrict__ Xt,
                                                         const double* __restrict__ Xq, int nxp,
                                                         const double* __restrict__ invp, const double* __restrict__ zvec,
                                                         const double* __restrict__ Z, double* __restrict__ g1o,
                                                         double* __restrict__ g2o) {
  __shared__ double sh[2 * D * 16];
  const int q = blockIdx.x, n = P.n, ng = P.ng;
  const int nblk = P.use_grad ? D + 1 : 1;
  double g[2 * D];
#pragma unroll
  for (int k = 0; k < 2 * D; ++k) g[k] = 0.0;
  double xq[D], th[D];
#pragma unroll
  for (int k = 0; k < D; ++k) { xq[k] = Xq[(size_t)k * nxp + q]; th[k] = P.theta[k]; }
  const double sqrt5 = sqrt(5.0);
  const double rq_const = 4.0 * (1.0 + 1.0 / P.hp_kernel);       // KernelRatQuad.py:529 (unused by the other kernels)
  for (int a = threadIdx.x; a < n; a += 256) {
    double R[D], E, M1 = 0.0;
    if (KERN == GPG_KERNEL_SQEXP) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) { R[k] = Xt[(size_t)k * n + a] - xq[k]; s -= th[k] * (R[k] * R[k]); }
      E = exp(s);
    } else if (KERN == GPG_KERNEL_RATQU) {       // KernelRatQuad.py:463-476: M1 = B^(-alpha-1), E <- B^(-alpha-2)
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) { R[k] = Xt[(size_t)k * n + a] - xq[k]; s += th[k] * (R[k] * R[k]); }
      const double Bq = 1.0 + s / P.hp_kernel;
      M1 = pow(Bq, -P.hp_kernel - 1.0);
      E = pow(Bq, -P.hp_kernel - 2.0);
    } else {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) { R[k] = Xt[(size_t)k * n + a] - xq[k]; s += th[k] * (R[k] * R[k]); }
      const double nu = sqrt(s);
      E = exp(-sqrt5 * nu);
      M1 = ((5.0 / 3.0) * (1.0 + sqrt5 * nu)) * E;
    }
    const int gpa = P.gpos[a];
    // I = 0 : block (0, j'+1) = +2 th R E  (SqExp) / + th R mat1 (Matern)
    {
      const double wa = zvec[a] * invp[a], ws = Z[(size_t)a * nxp + q] * invp[a];
#pragma unroll
      for (int jp = 0; jp < D; ++jp) {
        const double v = KERN == GPG_KERNEL_SQEXP ? ((2.0 * th[jp]) * R[jp]) * E
                         : KERN == GPG_KERNEL_RATQU ? ((2.0 * th[jp]) * R[jp]) * M1 : (th[jp] * R[jp]) * M1;   // RatQu: :541
        g[jp] += v * wa;
        g[D + jp] += v * ws;
      }
    }
    if (nblk > 1 && gpa >= 0) {
#pragma unroll
      for (int i = 0; i < D; ++i) {
        const size_t c = (size_t)n + (size_t)i * ng + gpa;
        const double ip = invp[c];
        const double wa = zvec[c] * ip, ws = Z[c * nxp + q] * ip;
#pragma unroll
        for (int jp = 0; jp < D; ++jp) {
          double v;
          if (KERN == GPG_KERNEL_SQEXP) {
            v = (i == jp) ? (2.0 * th[i] - (4.0 * (th[i] * th[i])) * (R[i] * R[i])) * E
                          : ((-4.0 * th[i]) * th[jp]) * ((R[i] * R[jp]) * E);
          } else if (KERN == GPG_KERNEL_RATQU) {     // KernelRatQuad.py:544, 554
            v = (i == jp) ? (2.0 * th[i]) * M1 - ((rq_const * (th[i] * th[i])) * (R[i] * R[i])) * E
                          : (((((-rq_const) * th[i]) * th[jp]) * R[i]) * R[jp]) * E;
          } else {
            v = (i == jp) ? th[i] * M1 - (((25.0 / 3.0) * (th[i] * th[i])) * (R[i] * R[i])) * E
                          : (((((-(25.0 / 3.0)) * th[i]) * th[jp]) * R[i]) * R[jp]) * E;
          }
          g[jp] += v * wa;
          g[D + jp] += v * ws;
        }
      }
    }
  }
  block_sum<2 * D>(g, sh);
  if (threadIdx.x < D) {
    g1o[(size_t)q * D + threadIdx.x] = g[threadIdx.x];
    g2o[(size_t)q * D + threadIdx.x] = g[D + threadIdx.x];
  }
}

template <int KERN>
void launch_cross_grad_d(gpg_ctx* c, const AsmParams& p, int nx, int nxp, double* g1, double* g2) {
#define CASE_D(DD)                                                                                          \
  case DD:                                                                                                  \
    hipLaunchKernelGGL((cross_grad_kernel<KERN, DD>), dim3(nx), dim3(256), 0, c->stream, p, c->Xt, c->xq_dev, nxp, \
                       c->invp, c->zvec, c->Wt, g1, g2);                                                    \
    break;
  switch (p.d) {
    CASE_D(1) CASE_D(2) CASE_D(3) CASE_D(4) CASE_D(5) CASE_D(6) CASE_D(7) CASE_D(8)
    CASE_D(9) CASE_D(10) CASE_D(11) CASE_D(12) CASE_D(13) CASE_D(14) CASE_D(15) CASE_D(16)
  }
#undef CASE_D
}

// ------------------------------------------------------------------------------------------------
// Posterior Hessians at ONE query point (reference GpEvalModel.py:355-382, kernel second / third derivatives
// KernelSqExp.py:48-88,412-468, KernelMatern5f2.py:54-94,453-530).  With R = x_a - xq (this file's sign):
//   base column a          : d2K[k,i] = (-2 th_i d_ik + 4 th_i th_k R_i R_k) E                     (SqExp)
//                                       -th_k M1 d_ik + (25/3) th_i R_i th_k R_k E                 (Matern 5/2)
//   gradient column (j, a) : d2K[k,i] = (4 th_i th_j (d_ik R_j + d_jk R_i) + 4 d_ij th_i th_k R_k
//                                        - 8 th_i th_j th_k R_i R_j R_k) E                          (SqExp)
//                                       (25/3) (th_i d_ik th_j R_j + th_j d_jk th_i R_i + th_i d_ij th_k R_k
//                                        - sqrt5 / nu th_i R_i th_j R_j th_k R_k) E                 (Matern 5/2)
// hess_contract_kernel: workgroup k -> row k of  H1 = sum_c d2K[k,:,c] alpha_c  and  H2 = sum_c d2K[k,:,c] v_c,
// v = K^-1 Kyx (row 0 of Z).  Nothing of the [d, d, N] tensor is stored.
// ------------------------------------------------------------------------------------------------
template <int KERN, int D>
__global__ void __launch_bounds__(256) hess_contract_kernel(AsmParams P, const double* __restrict__ Xt,
                                                            const double* __restrict__ Xq, int nxp,
                                                            const double* __restrict__ invp, const double* __restrict__ zvec,
                                                            const double* __restrict__ Z, double* __restrict__ h1o,
                                                            double* __restrict__ h2o) {
  __shared__ double sh[2 * D * 16];
  const int k = blockIdx.x, n = P.n, ng = P.ng;
  const int nblk = P.use_grad ? D + 1 : 1;
  double g[2 * D];
#pragma unroll
  for (int i = 0; i < 2 * D; ++i) g[i] = 0.0;
  double xq[D], th[D];
#pragma unroll
  for (int i = 0; i < D; ++i) { xq[i] = Xq[(size_t)i * nxp]; th[i] = P.theta[i]; }
  const double sqrt5 = sqrt(5.0);
  const double rq_s1 = 1.0 + 1.0 / P.hp_kernel, rq_s2 = rq_s1 * (1.0 + 2.0 / P.hp_kernel);   // RatQu only
  for (int a = threadIdx.x; a < n; a += 256) {
    double R[D], tR[D], E, M1 = 0.0, c3 = 0.0, F3 = 0.0;
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) { R[i] = Xt[(size_t)i * n + a] - xq[i]; tR[i] = th[i] * R[i]; s += tR[i] * R[i]; }
    if (KERN == GPG_KERNEL_SQEXP) {
      E = exp(-s);
    } else if (KERN == GPG_KERNEL_RATQU) {               // f_p = B^(-alpha-p): M1 = f_1, E = f_2, F3 = f_3
      const double Bq = 1.0 + s / P.hp_kernel;
      M1 = pow(Bq, -P.hp_kernel - 1.0);
      E = pow(Bq, -P.hp_kernel - 2.0);
      F3 = pow(Bq, -P.hp_kernel - 3.0);
    } else {
      const double nu = sqrt(s);
      E = exp(-sqrt5 * nu);
      M1 = ((5.0 / 3.0) * (1.0 + sqrt5 * nu)) * E;
      c3 = sqrt5 / fmax(nu, 1e-16);                      // KernelMatern5f2.py:500
    }
    const double tRk = tR[k];
    {
      const double wa = zvec[a] * invp[a], ws = Z[(size_t)a * nxp] * invp[a];
#pragma unroll
      for (int i = 0; i < D; ++i) {
        double v;
        if (KERN == GPG_KERNEL_SQEXP) v = (4.0 * tR[i] * tRk - (i == k ? 2.0 * th[i] : 0.0)) * E;
        else if (KERN == GPG_KERNEL_RATQU) v = (4.0 * rq_s1) * tR[i] * tRk * E - (i == k ? 2.0 * th[i] * M1 : 0.0);   // KernelRatQuad.py:51-136
        else v = (25.0 / 3.0) * tR[i] * tRk * E - (i == k ? th[k] * M1 : 0.0);
        g[i] += v * wa;
        g[D + i] += v * ws;
      }
    }
    const int gpa = P.gpos[a];
    if (nblk > 1 && gpa >= 0) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const size_t c = (size_t)n + (size_t)j * ng + gpa;
        const double ip = invp[c];
        const double wa = zvec[c] * ip, ws = Z[c * nxp] * ip;
#pragma unroll
        for (int i = 0; i < D; ++i) {
          double v;
          if (KERN == GPG_KERNEL_SQEXP) {
            v = 4.0 * ((i == k ? th[i] * tR[j] : 0.0) + (j == k ? th[j] * tR[i] : 0.0) + (i == j ? th[i] * tRk : 0.0))
                - 8.0 * tR[i] * tR[j] * tRk;
            v *= E;
          } else if (KERN == GPG_KERNEL_RATQU) {           // KernelRatQuad.py:556-632
            v = (4.0 * rq_s1) * ((i == k ? th[i] * tR[j] : 0.0) + (j == k ? th[j] * tR[i] : 0.0) + (i == j ? th[i] * tRk : 0.0)) * E
                - (8.0 * rq_s2) * tR[i] * tR[j] * tRk * F3;
          } else {
            v = (i == k ? th[i] * tR[j] : 0.0) + (j == k ? th[j] * tR[i] : 0.0) + (i == j ? th[i] * tRk : 0.0)
                - c3 * tR[i] * tR[j] * tRk;
            v *= (25.0 / 3.0) * E;
          }
          g[i] += v * wa;
          g[D + i] += v * ws;
        }
      }
    }
  }
  block_sum<2 * D>(g, sh);
  if (threadIdx.x < D) {
    h1o[(size_t)k * D + threadIdx.x] = g[threadIdx.x];
    h2o[(size_t)k * D + threadIdx.x] = g[D + threadIdx.x];
  }
}

// Rows 1 .. d of Wt (RHS-rows layout, leading dimension nxp) <- P^-1 dKxy_dx[(j', q=0), :]: the d derivative rows
// of the cross-covariance at the query point (same entries as cross_grad_kernel), for the forward sweep that gives
// dKxy_dx K^-1 dKxy_dx^T (GpEvalModel.py:367).  Row 0 is cleared.
template <int KERN, int D>
__global__ void __launch_bounds__(256) cross_dx_rows_kernel(AsmParams P, const double* __restrict__ Xt,
                                                            const double* __restrict__ Xq, int nxp,
                                                            const double* __restrict__ invp, double* __restrict__ Wt) {
  const int a = blockIdx.x * 256 + threadIdx.x, n = P.n, ng = P.ng;
  if (a >= n) return;
  const int nblk = P.use_grad ? D + 1 : 1;
  double R[D], th[D], E, M1 = 0.0, s = 0.0;
  const double sqrt5 = sqrt(5.0);
#pragma unroll
  for (int i = 0; i < D; ++i) { th[i] = P.theta[i]; R[i] = Xt[(size_t)i * n + a] - Xq[(size_t)i * nxp]; s += th[i] * (R[i] * R[i]); }
  if (KERN == GPG_KERNEL_SQEXP) {
    E = exp(-s);
  } else if (KERN == GPG_KERNEL_RATQU) {
    const double Bq = 1.0 + s / P.hp_kernel;
    M1 = pow(Bq, -P.hp_kernel - 1.0);
    E = pow(Bq, -P.hp_kernel - 2.0);
  } else {
    const double nu = sqrt(s);
    E = exp(-sqrt5 * nu);
    M1 = ((5.0 / 3.0) * (1.0 + sqrt5 * nu)) * E;
  }
  {
    const double ip = invp[a];
    Wt[(size_t)a * nxp] = 0.0;
#pragma unroll
    for (int jp = 0; jp < D; ++jp)
      Wt[(size_t)a * nxp + 1 + jp] = ip * (KERN == GPG_KERNEL_SQEXP ? ((2.0 * th[jp]) * R[jp]) * E
                                          : KERN == GPG_KERNEL_RATQU ? ((2.0 * th[jp]) * R[jp]) * M1 : (th[jp] * R[jp]) * M1);
  }
  const int gpa = P.gpos[a];
  if (nblk > 1 && gpa >= 0) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      const size_t c = (size_t)n + (size_t)i * ng + gpa;
      const double ip = invp[c];
      Wt[c * nxp] = 0.0;
#pragma unroll
      for (int jp = 0; jp < D; ++jp) {
        double v;
        if (KERN == GPG_KERNEL_SQEXP) {
          v = (i == jp) ? (2.0 * th[i] - (4.0 * (th[i] * th[i])) * (R[i] * R[i])) * E
                        : ((-4.0 * th[i]) * th[jp]) * ((R[i] * R[jp]) * E);
        } else if (KERN == GPG_KERNEL_RATQU) {
          const double rq_c = 4.0 * (1.0 + 1.0 / P.hp_kernel);
          v = (i == jp) ? (2.0 * th[i]) * M1 - ((rq_c * (th[i] * th[i])) * (R[i] * R[i])) * E
                        : (((((-rq_c) * th[i]) * th[jp]) * R[i]) * R[jp]) * E;
        } else {
          v = (i == jp) ? th[i] * M1 - (((25.0 / 3.0) * (th[i] * th[i])) * (R[i] * R[i])) * E
                        : (((((-(25.0 / 3.0)) * th[i]) * th[jp]) * R[i]) * R[jp]) * E;
        }
        Wt[c * nxp + 1 + jp] = ip * v;
      }
    }
  }
}

// T[i, i'] = sum_c W[1 + i, c] W[1 + i', c]  (workgroup i; rows of the RHS-rows layout after the forward sweep)
template <int D>
__global__ void __launch_bounds__(256) rows_gram_kernel(const double* __restrict__ Wt, int nxp, int Npad,
                                                        double* __restrict__ T) {
  __shared__ double sh[D * 16];
  const int i = blockIdx.x;
  double g[D];
#pragma unroll
  for (int q = 0; q < D; ++q) g[q] = 0.0;
  for (int c = threadIdx.x; c < Npad; c += 256) {
    const double* w = Wt + (size_t)c * nxp + 1;
    const double wi = w[i];
#pragma unroll
    for (int q = 0; q < D; ++q) g[q] += wi * w[q];
  }
  block_sum<D>(g, sh);
  if (threadIdx.x < D) T[(size_t)i * D + threadIdx.x] = g[threadIdx.x];
}

template <int KERN>
void launch_hess_d(gpg_ctx* c, const AsmParams& p, int nxp, double* h1, double* h2, double* T, int stage) {
#define CASE_D(DD)                                                                                                   \
  case DD:                                                                                                           \
    if (stage == 0)                                                                                                  \
      hipLaunchKernelGGL((hess_contract_kernel<KERN, DD>), dim3(DD), dim3(256), 0, c->stream, p, c->Xt, c->xq_dev, nxp,  \
                         c->invp, c->zvec, c->Wt, h1, h2);                                                           \
    else if (stage == 1)                                                                                             \
      hipLaunchKernelGGL((cross_dx_rows_kernel<KERN, DD>), dim3((p.n + 255) / 256), dim3(256), 0, c->stream, p, c->Xt,   \
                         c->xq_dev, nxp, c->invp, c->Wt);                                                            \
    else                                                                                                             \
      hipLaunchKernelGGL((rows_gram_kernel<DD>), dim3(DD), dim3(256), 0, c->stream, c->Wt, nxp, c->Npad, T);          \
    break;
  switch (p.d) {
    CASE_D(1) CASE_D(2) CASE_D(3) CASE_D(4) CASE_D(5) CASE_D(6) CASE_D(7) CASE_D(8)
    CASE_D(9) CASE_D(10) CASE_D(11) CASE_D(12) CASE_D(13) CASE_D(14) CASE_D(15) CASE_D(16)
  }
#undef CASE_D
}

// which 0..2: symmetric copy of the lower triangle; which 3: P L (lower), zeros above
__global__ void extract_kernel(const double* __restrict__ A, int ld, int N, const double* __restrict__ dvec,
                               int precon, int which, double* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (c >= N) return;
  double v;
  if (which == 3) {
    v = r >= c ? A[(size_t)r + (size_t)c * ld] * (precon ? sqrt(dvec[r]) : 1.0) : 0.0;
  } else {
    v = r >= c ? A[(size_t)r + (size_t)c * ld] : A[(size_t)c + (size_t)r * ld];
  }
  out[(size_t)r * N + c] = v;
}

}  // namespace

void gpg_launch_lkd_reduce(gpg_ctx* c, int slot) {
  gpg_prof_begin(c, GPG_PROF_REDUCE, 0.0);
  hipLaunchKernelGGL(lkd_reduce_kernel, dim3(1), dim3(1024), 0, c->stream, c->A, c->ld, c->N, c->Npad, c->dvec,
                     c->last_precon, c->scal + (size_t)slot * 8, (size_t)0, (size_t)0, c->info);   // c->info: this evaluation's word
  gpg_prof_end(c);
}

void gpg_launch_lkd_reduce_batch(gpg_ctx* c, int slot0, int B, size_t v_stride, size_t a_stride, const int* info0) {
  gpg_prof_begin(c, GPG_PROF_REDUCE, 0.0);
  hipLaunchKernelGGL(lkd_reduce_kernel, dim3(B), dim3(1024), 0, c->stream, c->A, c->ld, c->N, c->Npad, c->dvec,
                     c->last_precon, c->scal + (size_t)slot0 * 8, v_stride, a_stride, info0);
  gpg_prof_end(c);
}

void gpg_backward_solve(gpg_ctx* c) {
  const int Npad = c->Npad;
  if (gpg_dataflow_solves(c)) {
    // the vector rides as row 0 of a zeroed 64-row tile through the dataflow backward solve (one launch instead of
    // 2 Npad / 64 dependent ones)
    if (!c->vec_rows) (void)gpg_dev_alloc(c, &c->vec_rows, sizeof(double) * 64 * (size_t)c->vec_rows_cols);
    if (c->vec_rows) {
      // right-hand side: the forward-solved RHS row 0 of the factorisation workspace (row Npad of A); rows 1..63 of the tile zero
      hipLaunchKernelGGL(vec_rows_load_kernel, dim3((unsigned)((64 * (size_t)Npad + 255) / 256)), dim3(256), 0, c->stream, c->vec_rows,
                         c->A + Npad, c->ld, Npad);
      if (gpg_launch_rows_bwd(c, c->vec_rows, 64, 64, 1)) {
        (void)hipMemcpy2DAsync(c->zvec, sizeof(double), c->vec_rows, 64 * sizeof(double), sizeof(double), Npad,
                               hipMemcpyDeviceToDevice, c->stream);
        return;
      }
    }
  }
  for (int k0 = Npad - 64; k0 >= 0; k0 -= 64) {
    const int has_t = (k0 + 64 < Npad);
    if (has_t)
      hipLaunchKernelGGL(bs_dot_kernel, dim3(64), dim3(256), 0, c->stream, c->A, c->ld, Npad, k0, c->zvec, c->tmpv);
    hipLaunchKernelGGL(bs_tri_kernel, dim3(1), dim3(64), 0, c->stream, c->A, c->ld, Npad, k0, c->tmpv, has_t, c->zvec);
  }
}

// Z (nrhs x Npad, RHS-rows layout, leading dimension ldz) <- Z L^-1, i.e. every row solved against L^T
void gpg_backward_rows(gpg_ctx* c, double* Z, int ldz, int nrhs, double* tbuf) {
  // one dataflow launch over the 64-row tiles that hold the nrhs rows (the other rows of a tile ride along)
  const int rows = ((nrhs + 63) / 64) * 64;
  if (gpg_dataflow_solves(c)) {
    if (gpg_launch_rows_bwd(c, Z, ldz, rows, nrhs)) return;
    // more row tiles than one dataflow launch takes: one launch per group of row tiles (rows are the fast index of Z)
    const int chunk = (c->rows_max_tasks / (c->Npad / 64)) * 64;
    if (chunk >= 64) {
      bool ok = true;
      for (int r0 = 0; r0 < rows && ok; r0 += chunk) {
        const int nr = std::min(chunk, rows - r0);
        ok = gpg_launch_rows_bwd(c, Z + r0, ldz, nr, std::min(nr, nrhs - r0));
      }
      if (ok) return;      // (a refusal can only come before the first launch: same shape every time)
    }
  }
  const int Npad = c->Npad;
  for (int k0 = Npad - 64; k0 >= 0; k0 -= 64) {
    const int has_t = (k0 + 64 < Npad);
    if (has_t)
      hipLaunchKernelGGL(bs_dot_multi_kernel, dim3(64, (nrhs + 15) / 16), dim3(256), 0, c->stream, c->A, c->ld, Npad, k0, Z,
                         ldz, nrhs, tbuf);
    hipLaunchKernelGGL(bs_tri_multi_kernel, dim3(nrhs), dim3(64), 0, c->stream, c->A, c->ld, k0, tbuf, has_t, Z, ldz);
  }
}

void gpg_launch_cross_grad(gpg_ctx* c, const AsmParams& p, int nx, int nxp, double* g1, double* g2) {
  if (p.kernel == GPG_KERNEL_SQEXP) launch_cross_grad_d<GPG_KERNEL_SQEXP>(c, p, nx, nxp, g1, g2);
  else if (p.kernel == GPG_KERNEL_RATQU) launch_cross_grad_d<GPG_KERNEL_RATQU>(c, p, nx, nxp, g1, g2);
  else launch_cross_grad_d<GPG_KERNEL_MA5F2>(c, p, nx, nxp, g1, g2);
}

// stage 0: H1 / H2 contraction, 1: derivative rows into Wt rows 1..d, 2: Gram of those rows after the forward sweep
void gpg_launch_hess_stage(gpg_ctx* c, const AsmParams& p, int nxp, double* h1, double* h2, double* T, int stage) {
  if (p.kernel == GPG_KERNEL_SQEXP) launch_hess_d<GPG_KERNEL_SQEXP>(c, p, nxp, h1, h2, T, stage);
  else if (p.kernel == GPG_KERNEL_RATQU) launch_hess_d<GPG_KERNEL_RATQU>(c, p, nxp, h1, h2, T, stage);
  else launch_hess_d<GPG_KERNEL_MA5F2>(c, p, nxp, h1, h2, T, stage);
}

// ---- products with the factor in place (condition number) ------------------------------------
// w = L^T v: one wave per column j (column-major storage: the column is contiguous), w[j] = sum_{i >= j} L[i][j] v[i]
__global__ void __launch_bounds__(256) trmv_t_kernel(const double* __restrict__ A, int ld, int n, const double* __restrict__ v,
                                                     double* __restrict__ w) {
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (j >= n) return;
  const double* col = A + (size_t)j * ld;
  double s = 0.0;
  for (int i = j + lane; i < n; i += 64) s += col[i] * v[i];
  s = wave_sum(s);
  if (lane == 0) w[j] = s;
}

// u = L w: thread <-> row i of a block of 256 rows, columns streamed in chunks of 256 (w chunk in LDS; every load of
// L is coalesced over the rows), u[i] = sum_{j <= i} L[i][j] w[j]
__global__ void __launch_bounds__(256) trmv_n_kernel(const double* __restrict__ A, int ld, int n, const double* __restrict__ w,
                                                     double* __restrict__ u) {
  __shared__ double ws[256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int jend = min(n, (int)(blockIdx.x + 1) * 256);
  double s = 0.0;
  for (int j0 = 0; j0 < jend; j0 += 256) {
    __syncthreads();
    ws[threadIdx.x] = (j0 + (int)threadIdx.x < n) ? w[j0 + threadIdx.x] : 0.0;
    __syncthreads();
    if (i < n) {
      const int jm = min(256, i - j0 + 1);               // columns j0 .. j0 + jm - 1 are <= i
      const double* p = A + (size_t)i + (size_t)j0 * ld;
#pragma unroll 8
      for (int jj = 0; jj < jm; ++jj) s += p[(size_t)jj * ld] * ws[jj];
    }
  }
  if (i < n) u[i] = s;
}

// out (device, Npad) <- (L L^T) v (op 0) or (L L^T)^-1 v (op 1) with the factor in A; v (device, Npad, zero tail) is
// overwritten.  The solves run through the single-row dataflow kernels on a carrier tile.
int gpg_factor_apply_dev(gpg_ctx* c, int op, double* v, double* out) {
  const int Npad = c->Npad;
  if (op == 0) {
    hipLaunchKernelGGL(trmv_t_kernel, dim3((Npad + 3) / 4), dim3(256), 0, c->stream, c->A, c->ld, Npad, v, out);   // out = L^T v
    hipLaunchKernelGGL(trmv_n_kernel, dim3((Npad + 255) / 256), dim3(256), 0, c->stream, c->A, c->ld, Npad, out, v);  // v = L out
    return hipMemcpyAsync(out, v, sizeof(double) * Npad, hipMemcpyDeviceToDevice, c->stream) == hipSuccess ? 0 : -2;
  }
  if (!c->vec_rows && !gpg_dev_alloc(c, &c->vec_rows, sizeof(double) * 64 * (size_t)c->vec_rows_cols)) return -2;
  hipLaunchKernelGGL(vec_rows_load_kernel, dim3((unsigned)((64 * (size_t)Npad + 255) / 256)), dim3(256), 0, c->stream, c->vec_rows, v, 1,
                     Npad);                                                       // row 0 of the carrier <- v
  if (gpg_dataflow_solves(c)) {
    if (!gpg_launch_rows_fwd(c, c->vec_rows, 64, 64, 1) || !gpg_launch_rows_bwd(c, c->vec_rows, 64, 64, 1)) return -2;
  } else {
    gpg_forward_rows(c, c->vec_rows, 64, 64, 1);
    gpg_backward_rows(c, c->vec_rows, 64, 1, c->gradbuf ? c->gradbuf : v);
  }
  return hipMemcpy2DAsync(out, sizeof(double), c->vec_rows, 64 * sizeof(double), sizeof(double), Npad, hipMemcpyDeviceToDevice,
                          c->stream) == hipSuccess ? 0 : -2;
}

void gpg_launch_alpha(gpg_ctx* c, double* alpha_dev) {
  hipLaunchKernelGGL(alpha_kernel, dim3((c->N + 255) / 256), dim3(256), 0, c->stream, c->zvec, c->invp, c->N,
                     alpha_dev);
}

void gpg_launch_predict_reduce(gpg_ctx* c, int nx, int nxp, double beta, double varK, int phase) {
  (void)nx; (void)varK;
  const int S = std::min(128, c->Npad / nxp);      // slices; the partial sums live in tmpv [Npad]
  if (S >= 2) {
    const int chunk = (c->N + S - 1) / S;
    hipLaunchKernelGGL(predict_partial_kernel, dim3(nxp / 64, S), dim3(1024), 0, c->stream, c->Wt, nxp, c->N, chunk, c->zvec,
                       phase, c->tmpv);
    hipLaunchKernelGGL(predict_final_kernel, dim3(nxp / 64), dim3(1024), 0, c->stream, c->tmpv, nxp, S, beta, phase,
                       c->musig + (size_t)phase * nxp);
    return;
  }
  hipLaunchKernelGGL(predict_reduce_kernel, dim3(nxp / 64), dim3(1024), 0, c->stream, c->Wt, nxp, c->N, c->zvec, beta,
                     phase, c->musig + (size_t)phase * nxp);
}

void gpg_launch_extract(gpg_ctx* c, int which) {
  dim3 grid((c->N + 255) / 256, c->N);
  hipLaunchKernelGGL(extract_kernel, grid, dim3(256), 0, c->stream, c->A, c->ld, c->N, c->dvec, c->last_precon, which,
                     c->dense_tmp);
}
