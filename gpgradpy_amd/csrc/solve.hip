// Reductions and triangular sweeps around the factorisation (gfx950).
//   lkd_reduce      : ln det, GLS mean beta, r' K^-1 r from the diagonal of L and the two RHS rows
//                     (reference GpMeanFun.py:102-108, CalcLkd.py:153-168 / 220-226) -- wave-reduced
//   backward solve  : z = L^-T w for the single vector needed by alpha (GpEvalModel.py:57)
//   predict_reduce  : mu = beta + Kyx' alpha, sig2 = 1 - diag(Kxy K^-1 Kyx) (GpEvalModel.py:162-168)
//   extract         : dense N x N copies on request (tests / drop-in 7-tuple)
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

// scal[0] = ln_det, [1] = beta, [2] = r'K^-1 r, [3] = V'K^-1 V, [4] = V'K^-1 y
__global__ void __launch_bounds__(1024) lkd_reduce_kernel(const double* __restrict__ A, int ld, int N, int Npad,
                                                          const double* __restrict__ dvec, int precon,
                                                          double* __restrict__ scal) {
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
  }
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

// phase 0: out[j] = beta + sum_c Wt[j, c] z[c] ; phase 1: out[j] = 1 - sum_c Wt[j, c]^2
__global__ void __launch_bounds__(1024) predict_reduce_kernel(const double* __restrict__ Wt, int nxp, int N,
                                                              const double* __restrict__ z, double beta, int phase,
                                                              double* __restrict__ out) {
  __shared__ double sh[16][64];
  const int jl = threadIdx.x & 63, cg = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + jl;
  double acc = 0.0;
  if (phase == 0) {
    for (int c = cg; c < N; c += 16) acc += Wt[(size_t)c * nxp + j] * z[c];
  } else {
    for (int c = cg; c < N; c += 16) { double w = Wt[(size_t)c * nxp + j]; acc += w * w; }
  }
  sh[cg][jl] = acc;
  __syncthreads();
  if (cg == 0) {
    double s = 0.0;
    for (int q = 0; q < 16; ++q) s += sh[q][jl];
    out[j] = phase == 0 ? beta + s : 1.0 - s;
  }
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
                     c->last_precon, c->scal + (size_t)slot * 8);
  gpg_prof_end(c);
}

void gpg_backward_solve(gpg_ctx* c) {
  const int Npad = c->Npad;
  for (int k0 = Npad - 64; k0 >= 0; k0 -= 64) {
    const int has_t = (k0 + 64 < Npad);
    if (has_t)
      hipLaunchKernelGGL(bs_dot_kernel, dim3(64), dim3(256), 0, c->stream, c->A, c->ld, Npad, k0, c->zvec, c->tmpv);
    hipLaunchKernelGGL(bs_tri_kernel, dim3(1), dim3(64), 0, c->stream, c->A, c->ld, Npad, k0, c->tmpv, has_t, c->zvec);
  }
}

void gpg_launch_alpha(gpg_ctx* c, double* alpha_dev) {
  hipLaunchKernelGGL(alpha_kernel, dim3((c->N + 255) / 256), dim3(256), 0, c->stream, c->zvec, c->invp, c->N,
                     alpha_dev);
}

void gpg_launch_predict_reduce(gpg_ctx* c, int nx, int nxp, double beta, double varK, int phase) {
  (void)nx; (void)varK;
  hipLaunchKernelGGL(predict_reduce_kernel, dim3(nxp / 64), dim3(1024), 0, c->stream, c->Wt, nxp, c->N, c->zvec, beta,
                     phase, c->musig + (size_t)phase * nxp);
}

void gpg_launch_extract(gpg_ctx* c, int which) {
  dim3 grid((c->N + 255) / 256, c->N);
  hipLaunchKernelGGL(extract_kernel, grid, dim3(256), 0, c->stream, c->A, c->ld, c->N, c->dvec, c->last_precon, which,
                     c->dense_tmp);
}
