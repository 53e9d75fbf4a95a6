// Fused assembly of the gradient-enhanced covariance matrix (gfx950).
//
// One pass produces, directly in the form that is factorised,
//     Kp = varK * (P^-1 (K + diag(noise / varK)) P^-1 + eta I)           (reference Kernel.py:213-236)
// from the n x d design alone: the [d, n, n] difference tensor of the reference
// (CommonFun.py:56-84) is recomputed in registers, the diagonal preconditioner and the nugget are
// applied in-register, and only the lower triangle (column-major) is written.
// Kernel formulas: KernelSqExp.py:338-408 and KernelMatern5f2.py:369-448; the operation order of those
// lines is kept (this file is compiled with -ffp-contract=off) so that entries agree with the NumPy
// reference to the last ulp of exp()/sqrt().
//
// Mapping: lane <-> data point a (rows r = I*n + a are contiguous in a, so every store instruction of
// a wave writes 512 contiguous bytes of one matrix column); each thread walks TB columns b and emits
// the (d+1)(d+2)/2 block entries of the ordered pair (a, b) that fall in the lower triangle.
#include <cstring>
#include "gpg_internal.h"

namespace {

constexpr int kTB = 16;  // columns (points b) per workgroup

__device__ __forceinline__ double kdiag_of(int kernel, int blk, const double* theta) {
  if (blk == 0) return 1.0;
  return kernel == GPG_KERNEL_MA5F2 ? theta[blk - 1] * (5.0 / 3.0) : 2.0 * theta[blk - 1];   // SqExp and RatQu: 2 theta
}

// dvec = diag(Kern) + noise / varK ; invp = 1/sqrt(dvec) (Kernel.py:218,224-226)
// Batched launches (gpg_lkd_batch): items != nullptr, workgroup coordinate `bz` selects the matrix; its parameters
// come from items[bz] and its buffers sit at bz * stride behind the base pointers.
__global__ void prep_diag_kernel(AsmParams P, const double* __restrict__ noise, double var_fval, double var_fgrad,
                                 double* __restrict__ dvec, double* __restrict__ invp,
                                 const gpg_batch_item* __restrict__ items, size_t v_stride, int* __restrict__ zero_info) {
  if (items) {
    const int bz = blockIdx.y;
    P = items[bz].p; var_fval = items[bz].var_fval; var_fgrad = items[bz].var_fgrad;
    dvec += bz * v_stride; invp += bz * v_stride;
  }
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (zero_info && r == 0) *zero_info = 0;               // the evaluation's info word (first launch of the evaluation)
  if (r >= P.Npad) return;
  if (r >= P.N) { dvec[r] = 1.0; invp[r] = 1.0; return; }
  int blk = (P.use_grad && r >= P.n) ? 1 + (r - P.n) / P.ng : 0;
  double kd = kdiag_of(P.kernel, blk, P.theta);
  double nz = blk == 0 ? (var_fval >= 0.0 ? var_fval : noise[r]) : (var_fgrad >= 0.0 ? var_fgrad : noise[r]);
  double dv = kd + nz / P.varK;
  dvec[r] = dv;
  invp[r] = P.precon ? 1.0 / sqrt(dv) : 1.0;
}

// rows [N, ld) of every column: identity padding and the right-hand-side rows
//   rhs row 0 = (s0 V + t0 y) P^-1, rhs row 1 = (s1 V + t1 y) P^-1, V = [1_n; 0] (GpMeanFun.py:172-191)
__global__ void prep_rows_kernel(AsmParams P, const double* __restrict__ y, const double* __restrict__ invp,
                                 double s0, double t0, double s1, double t1, double* __restrict__ A,
                                 const gpg_batch_item* __restrict__ items, size_t v_stride, size_t a_stride) {
  if (items) {
    const int bz = blockIdx.y;
    P = items[bz].p;
    invp += bz * v_stride; A += bz * a_stride;
  }
  int c = blockIdx.x;
  size_t col = (size_t)c * P.ld;
  for (int r = P.N + threadIdx.x; r < P.ld; r += blockDim.x) {
    double v = 0.0;
    if (r < P.Npad) {
      v = (r == c) ? 1.0 : 0.0;
    } else if (c < P.N) {
      int j = r - P.Npad;
      double vc = c < P.n ? 1.0 : 0.0;
      if (j == 0) v = (s0 * vc + t0 * y[c]) * invp[c];
      else if (j == 1) v = (s1 * vc + t1 * y[c]) * invp[c];
    }
    A[col + r] = v;
  }
}

template <int KERN, int D>
__global__ void __launch_bounds__(256) assemble_kernel(AsmParams P, const double* __restrict__ Xt,
                                                       const double* __restrict__ dvec,
                                                       const double* __restrict__ invp, double* __restrict__ A,
                                                       const gpg_batch_item* __restrict__ items, size_t v_stride,
                                                       size_t a_stride, int tb /* points b per workgroup, <= kTB */) {
  if (items) {
    const int bz = blockIdx.z;
    P = items[bz].p;
    dvec += bz * v_stride; invp += bz * v_stride; A += bz * a_stride;
  }
  __shared__ double xb[kTB][D];
  __shared__ double ipb[kTB][D + 1];
  __shared__ int gpb[kTB];
  const int n = P.n, ng = P.ng;
  const int a = blockIdx.x * 256 + threadIdx.x;
  const int b0 = blockIdx.y * tb;
  const int nblk = P.use_grad ? D + 1 : 1;
  // gradient rows / columns exist only for points whose gradient is used: index n + (I-1) ng + gpos[pt]
  for (int t = threadIdx.x; t < tb * D; t += 256) {
    int bb = t / D, k = t % D, b = b0 + bb;
    xb[bb][k] = b < n ? Xt[(size_t)k * n + b] : 0.0;
  }
  for (int t = threadIdx.x; t < tb * (D + 1); t += 256) {
    int bb = t / (D + 1), J = t % (D + 1), b = b0 + bb;
    double v = 0.0;
    if (b < n && J < nblk) {
      const int gp = P.gpos[b];
      if (J == 0) v = invp[b];
      else if (gp >= 0) v = invp[(size_t)n + (size_t)(J - 1) * ng + gp];
    }
    ipb[bb][J] = v;
  }
  if (threadIdx.x < tb) gpb[threadIdx.x] = (b0 + threadIdx.x < n) ? P.gpos[b0 + threadIdx.x] : -1;
  __syncthreads();
  if (a >= n) return;

  const int gpa = P.gpos[a];
  double xa[D], ipa[D + 1], th[D];
#pragma unroll
  for (int k = 0; k < D; ++k) { xa[k] = Xt[(size_t)k * n + a]; th[k] = P.theta[k]; }
  ipa[0] = invp[a];
#pragma unroll
  for (int I = 1; I <= D; ++I) ipa[I] = (I < nblk && gpa >= 0) ? invp[(size_t)n + (size_t)(I - 1) * ng + gpa] : 0.0;

  const int mode = P.mode, precon = P.precon, ld = P.ld;
  const double varK = P.varK, eta = P.eta;
  const int bend = min(tb, n - b0);
  const double sqrt5 = sqrt(5.0);
  const double rq_alpha = P.hp_kernel, rq_const = 4.0 * (1.0 + 1.0 / P.hp_kernel);   // KernelRatQuad.py:529

  for (int bb = 0; bb < bend; ++bb) {
    const int b = b0 + bb;
    double R[D];
    double E, M1 = 0.0, K00;
    if (KERN == GPG_KERNEL_SQEXP) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) { R[k] = xa[k] - xb[bb][k]; s -= th[k] * (R[k] * R[k]); }
      E = exp(s);
      K00 = E;
    } else if (KERN == GPG_KERNEL_RATQU) {   // KernelRatQuad.py:463-476: B = 1 + sum theta R^2 / alpha
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) { R[k] = xa[k] - xb[bb][k]; s += th[k] * (R[k] * R[k]); }
      const double Bq = 1.0 + s / rq_alpha;
      K00 = pow(Bq, -rq_alpha);
      M1 = pow(Bq, -rq_alpha - 1.0);
      E = pow(Bq, -rq_alpha - 2.0);            // B^(-alpha-2) takes the place of E in the second derivatives
    } else {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) { R[k] = xa[k] - xb[bb][k]; s += th[k] * (R[k] * R[k]); }
      double nu = sqrt(s);
      E = exp(-sqrt5 * nu);                       // Abase
      M1 = ((5.0 / 3.0) * (1.0 + sqrt5 * nu)) * E;  // mat1
      K00 = (1.0 + sqrt5 * nu + (5.0 / 3.0) * (nu * nu)) * E;
    }
    const bool diag_pt = (a == b);
    const bool low_pt = (a >= b);
    const int gpbb = gpb[bb];

    // emit one entry: block (I, J), kernel value v
    auto emit = [&](int I, int J, double v) {
      if ((I > 0 && gpa < 0) || (J > 0 && gpbb < 0)) return;   // gradient of that point not in use
      const size_t r = I == 0 ? (size_t)a : (size_t)n + (size_t)(I - 1) * ng + gpa;
      const size_t c = J == 0 ? (size_t)b : (size_t)n + (size_t)(J - 1) * ng + gpbb;
      double o;
      if (mode == 1) {
        o = v;
      } else if (diag_pt && I == J) {
        double dv = dvec[r];
        if (precon) {
          double kc = (ipa[I] * dv) * ipa[I];
          o = varK * (kc + eta);
          if (mode == 2) { double p = sqrt(dv); o = (p * o) * p; }
        } else {
          o = varK * (dv + eta);
        }
      } else {
        if (precon) {
          o = varK * ((ipa[I] * v) * ipb[bb][J]);
          if (mode == 2) o = ((1.0 / ipa[I]) * o) * (1.0 / ipb[bb][J]);
        } else {
          o = varK * v;
        }
      }
      A[c * ld + r] = o;
    };

    if (low_pt) emit(0, 0, K00);
    if (nblk > 1) {
#pragma unroll
      for (int i = 0; i < D; ++i) {
        double vi0, vii;
        if (KERN == GPG_KERNEL_SQEXP) {
          vi0 = ((-2.0 * th[i]) * R[i]) * E;
          vii = (2.0 * th[i] - (4.0 * (th[i] * th[i])) * (R[i] * R[i])) * E;
        } else if (KERN == GPG_KERNEL_RATQU) {   // KernelRatQuad.py:539-544
          vi0 = ((-2.0 * th[i]) * R[i]) * M1;
          vii = (2.0 * th[i]) * M1 - ((rq_const * (th[i] * th[i])) * (R[i] * R[i])) * E;
        } else {
          vi0 = ((-th[i]) * R[i]) * M1;
          vii = th[i] * M1 - (((25.0 / 3.0) * (th[i] * th[i])) * (R[i] * R[i])) * E;
        }
        emit(i + 1, 0, vi0);
        if (low_pt) emit(i + 1, i + 1, vii);
#pragma unroll
        for (int j = 0; j < D; ++j) {
          if (j < i) {  // block row i+1 > block column j+1; the reference's (lo, hi) = (j, i) ordering
            double vij;
            if (KERN == GPG_KERNEL_SQEXP) vij = ((-4.0 * th[j]) * th[i]) * ((R[j] * R[i]) * E);
            else if (KERN == GPG_KERNEL_RATQU) vij = (((((-rq_const) * th[j]) * th[i]) * R[j]) * R[i]) * E;   // :554
            else vij = (((((-(25.0 / 3.0)) * th[j]) * th[i]) * R[j]) * R[i]) * E;
            emit(i + 1, j + 1, vij);
          }
        }
      }
    }
  }
}

// Cross matrix for the posterior (GpEvalModel.py:133-139): Wt[j, c] = Kyx[c, j] / p_c, c = I*n + a,
// stored as RHS rows (query index j contiguous).
template <int KERN, int D>
__global__ void __launch_bounds__(256) cross_kernel(AsmParams P, const double* __restrict__ Xt,
                                                    const double* __restrict__ Xq, int nx, int nxp,
                                                    const double* __restrict__ invp, double* __restrict__ Wt) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int j = (int)(t % nxp);
  const long long a_ll = t / nxp;
  if (a_ll >= P.n) return;
  const int a = (int)a_ll, n = P.n;
  const int nblk = P.use_grad ? D + 1 : 1;
  double R[D];
  double E, M1 = 0.0, K00;
  const double sqrt5 = sqrt(5.0);
  if (j < nx) {
    if (KERN == GPG_KERNEL_SQEXP) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) { R[k] = Xt[(size_t)k * n + a] - Xq[(size_t)k * nxp + j]; s -= P.theta[k] * (R[k] * R[k]); }
      E = exp(s);
      K00 = E;
    } else if (KERN == GPG_KERNEL_RATQU) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) { R[k] = Xt[(size_t)k * n + a] - Xq[(size_t)k * nxp + j]; s += P.theta[k] * (R[k] * R[k]); }
      const double Bq = 1.0 + s / P.hp_kernel;
      K00 = pow(Bq, -P.hp_kernel);
      M1 = pow(Bq, -P.hp_kernel - 1.0);
      E = 0.0;
    } else {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) { R[k] = Xt[(size_t)k * n + a] - Xq[(size_t)k * nxp + j]; s += P.theta[k] * (R[k] * R[k]); }
      double nu = sqrt(s);
      E = exp(-sqrt5 * nu);
      M1 = ((5.0 / 3.0) * (1.0 + sqrt5 * nu)) * E;
      K00 = (1.0 + sqrt5 * nu + (5.0 / 3.0) * (nu * nu)) * E;
    }
  }
  const int gpa = P.gpos[a];
  for (int I = 0; I < nblk; ++I) {
    if (I > 0 && gpa < 0) break;
    const size_t c = I == 0 ? (size_t)a : (size_t)n + (size_t)(I - 1) * P.ng + gpa;
    double v = 0.0;
    if (j < nx) {
      if (I == 0) v = K00;
      else if (KERN == GPG_KERNEL_SQEXP) v = ((-2.0 * P.theta[I - 1]) * R[I - 1]) * E;
      else if (KERN == GPG_KERNEL_RATQU) v = ((-2.0 * P.theta[I - 1]) * R[I - 1]) * M1;   // KernelRatQuad.py:540
      else v = ((-P.theta[I - 1]) * R[I - 1]) * M1;
      v *= invp[c];
    }
    Wt[c * nxp + j] = v;
  }
}

// Kernel table entry points of the reference (Kernel.py:27-126: calc_KernBase / calc_KernGrad take the [d, n1, n2]
// difference tensor, NOT point sets): one thread per pair (a, b) reads R[:, a, b] and emits the (d+1)^2 block entries of
// KernelSqExp.py:381-408 / KernelMatern5f2.py:421-447 / KernelRatQuad.py:523-552 in those lines' operation order
// (this file is compiled with -ffp-contract=off).  out is row-major [n1 + n1g d, n2 + n2g d] (NumPy C order); gradient
// rows / columns exist for the points selected by bvec_use_grad1 / 2 (gpos = position among them, -1 = not used).
struct RtParams { int d, n1, n2, n1g, n2g, use_grad; double hp_kernel; double theta[GPG_MAX_DIM]; };

template <int KERN>
__global__ void __launch_bounds__(256) rtensor_kern_kernel(RtParams P, const double* __restrict__ Rt, const int* __restrict__ gpos1,
                                                           const int* __restrict__ gpos2, double* __restrict__ out) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long np = (long long)P.n1 * P.n2;
  if (t >= np) return;
  const int a = (int)(t / P.n2), b = (int)(t - (long long)a * P.n2);
  const int d = P.d;
  double R[GPG_MAX_DIM];
  double s = 0.0;
  for (int k = 0; k < d; ++k) {
    R[k] = Rt[(size_t)k * np + t];
    if (KERN == GPG_KERNEL_SQEXP) s -= P.theta[k] * (R[k] * R[k]);
    else s += P.theta[k] * (R[k] * R[k]);
  }
  double E, M1 = 0.0, K00, c2;                      // c2: factor of the mixed second derivatives
  if (KERN == GPG_KERNEL_SQEXP) {
    E = exp(s); K00 = E; c2 = -4.0;
  } else if (KERN == GPG_KERNEL_RATQU) {
    const double Bq = 1.0 + s / P.hp_kernel;
    K00 = pow(Bq, -P.hp_kernel);
    M1 = pow(Bq, -P.hp_kernel - 1.0);
    E = pow(Bq, -P.hp_kernel - 2.0);
    c2 = -(4.0 * (1.0 + 1.0 / P.hp_kernel));
  } else {
    const double sqrt5 = sqrt(5.0), nu = sqrt(s);
    E = exp(-sqrt5 * nu);
    M1 = ((5.0 / 3.0) * (1.0 + sqrt5 * nu)) * E;
    K00 = (1.0 + sqrt5 * nu + (5.0 / 3.0) * (nu * nu)) * E;
    c2 = -(25.0 / 3.0);
  }
  const size_t C2 = (size_t)P.n2 + (P.use_grad ? (size_t)P.n2g * d : 0);
  out[(size_t)a * C2 + b] = K00;
  if (!P.use_grad) return;
  const int g1 = gpos1[a], g2 = gpos2[b];
  for (int i = 0; i < d; ++i) {
    const double th = P.theta[i];
    double v10, v01, vii;
    if (KERN == GPG_KERNEL_SQEXP) {
      v10 = ((-2.0 * th) * R[i]) * E;
      v01 = ((2.0 * th) * R[i]) * E;
      vii = (2.0 * th - (4.0 * (th * th)) * (R[i] * R[i])) * E;
    } else if (KERN == GPG_KERNEL_RATQU) {
      v10 = ((-2.0 * th) * R[i]) * M1;
      v01 = ((2.0 * th) * R[i]) * M1;
      vii = (2.0 * th) * M1 - (((-c2) * (th * th)) * (R[i] * R[i])) * E;
    } else {
      v10 = ((-th) * R[i]) * M1;
      v01 = (th * R[i]) * M1;
      vii = th * M1 - (((25.0 / 3.0) * (th * th)) * (R[i] * R[i])) * E;
    }
    const size_t ri = (size_t)P.n1 + (size_t)i * P.n1g + g1, ci = (size_t)P.n2 + (size_t)i * P.n2g + g2;
    if (g1 >= 0) out[ri * C2 + b] = v10;
    if (g2 >= 0) out[(size_t)a * C2 + ci] = v01;
    if (g1 >= 0 && g2 >= 0) {
      out[ri * C2 + ci] = vii;
      for (int j = i + 1; j < d; ++j) {
        const double term = KERN == GPG_KERNEL_SQEXP ? ((-4.0 * th) * P.theta[j]) * ((R[i] * R[j]) * E)
                                                     : (((c2 * th) * P.theta[j]) * R[i]) * R[j] * E;
        const size_t rj = (size_t)P.n1 + (size_t)j * P.n1g + g1, cj = (size_t)P.n2 + (size_t)j * P.n2g + g2;
        out[ri * C2 + cj] = term;
        out[rj * C2 + ci] = term;
      }
    }
  }
}

template <int KERN>
void launch_assemble_d(gpg_ctx* c, const AsmParams& p, int B, const gpg_batch_item* items, size_t v_stride, size_t a_stride) {
  // every thread owns one point a and walks the tb points b of its workgroup one after the other (~2.5 us per pair at d = 4): small
  // data sets get fewer points b per workgroup, so that the launch is ~1000 workgroups wide instead of n / 16 deep chains
  const long wide = (long)((p.n + 255) / 256) * p.n * B;
  int tb = wide >= 8192 ? kTB : (int)(wide / 1024);
  tb = tb < 1 ? 1 : (tb > kTB ? kTB : tb);
  dim3 grid((p.n + 255) / 256, (p.n + tb - 1) / tb, B);
#define CASE_D(DD)                                                                                  \
  case DD:                                                                                          \
    hipLaunchKernelGGL((assemble_kernel<KERN, DD>), grid, dim3(256), 0, c->stream, p, c->Xt, c->dvec, \
                       c->invp, c->A, items, v_stride, a_stride, tb);                               \
    break;
  switch (p.d) {
    CASE_D(1) CASE_D(2) CASE_D(3) CASE_D(4) CASE_D(5) CASE_D(6) CASE_D(7) CASE_D(8)
    CASE_D(9) CASE_D(10) CASE_D(11) CASE_D(12) CASE_D(13) CASE_D(14) CASE_D(15) CASE_D(16)
  }
#undef CASE_D
}

template <int KERN>
void launch_cross_d(gpg_ctx* c, const AsmParams& p, int nx, int nxp) {
  long long total = (long long)p.n * nxp;
  dim3 grid((unsigned)((total + 255) / 256));
#define CASE_D(DD)                                                                                   \
  case DD:                                                                                           \
    hipLaunchKernelGGL((cross_kernel<KERN, DD>), grid, dim3(256), 0, c->stream, p, c->Xt, c->xq_dev, nx, \
                       nxp, c->invp, c->Wt);                                                         \
    break;
  switch (p.d) {
    CASE_D(1) CASE_D(2) CASE_D(3) CASE_D(4) CASE_D(5) CASE_D(6) CASE_D(7) CASE_D(8)
    CASE_D(9) CASE_D(10) CASE_D(11) CASE_D(12) CASE_D(13) CASE_D(14) CASE_D(15) CASE_D(16)
  }
#undef CASE_D
}

}  // namespace

void gpg_launch_prep(gpg_ctx* c, const AsmParams& p, double var_fval, double var_fgrad, double s0, double t0,
                     double s1, double t1) {
  if (c->ws_cur == 0) c->alpha_valid = false;   // invp of set 0 is rewritten
  int* zero_info = c->zero_info_in_prep ? c->info : nullptr;
  c->zero_info_in_prep = false;
  hipLaunchKernelGGL(prep_diag_kernel, dim3((p.Npad + 255) / 256), dim3(256), 0, c->stream, p, c->noise, var_fval,
                     var_fgrad, c->dvec, c->invp, (const gpg_batch_item*)nullptr, (size_t)0, zero_info);
  hipLaunchKernelGGL(prep_rows_kernel, dim3(p.Npad), dim3(256), 0, c->stream, p, c->y, c->invp, s0, t0, s1, t1, c->A,
                     (const gpg_batch_item*)nullptr, (size_t)0, (size_t)0);
}

void gpg_launch_assembly(gpg_ctx* c, const AsmParams& p) {
  double bytes = 8.0 * (double)p.N * ((double)p.N + 1.0) / 2.0;
  gpg_prof_begin(c, GPG_PROF_ASSEMBLY, bytes);
  if (p.kernel == GPG_KERNEL_SQEXP) launch_assemble_d<GPG_KERNEL_SQEXP>(c, p, 1, nullptr, 0, 0);
  else if (p.kernel == GPG_KERNEL_RATQU) launch_assemble_d<GPG_KERNEL_RATQU>(c, p, 1, nullptr, 0, 0);
  else launch_assemble_d<GPG_KERNEL_MA5F2>(c, p, 1, nullptr, 0, 0);
  gpg_prof_end(c);
}

// The three launches above for B matrices at once: c->dvec / c->invp / c->A are the buffers of matrix 0, matrix b
// sits b * v_stride / b * a_stride behind them, items[b] (device) carries its parameters; p = the parameters of any
// of them (shape only).  rhs rows as for a likelihood evaluation: row 0 = V P^-1, row 1 = y P^-1.
void gpg_launch_prep_assembly_batch(gpg_ctx* c, const AsmParams& p, int B, const gpg_batch_item* items, size_t v_stride,
                                    size_t a_stride) {
  c->alpha_valid = false;
  hipLaunchKernelGGL(prep_diag_kernel, dim3((p.Npad + 255) / 256, B), dim3(256), 0, c->stream, p, c->noise, 0.0, 0.0, c->dvec,
                     c->invp, items, v_stride, (int*)nullptr);
  hipLaunchKernelGGL(prep_rows_kernel, dim3(p.Npad, B), dim3(256), 0, c->stream, p, c->y, c->invp, 1.0, 0.0, 0.0, 1.0, c->A,
                     items, v_stride, a_stride);
  double bytes = 8.0 * (double)p.N * ((double)p.N + 1.0) / 2.0 * B;
  gpg_prof_begin(c, GPG_PROF_ASSEMBLY, bytes);
  if (p.kernel == GPG_KERNEL_SQEXP) launch_assemble_d<GPG_KERNEL_SQEXP>(c, p, B, items, v_stride, a_stride);
  else if (p.kernel == GPG_KERNEL_RATQU) launch_assemble_d<GPG_KERNEL_RATQU>(c, p, B, items, v_stride, a_stride);
  else launch_assemble_d<GPG_KERNEL_MA5F2>(c, p, B, items, v_stride, a_stride);
  gpg_prof_end(c);
}

// Row sums of |M| for the symmetric matrix whose lower triangle sits in A -- the quantity the reference's variable nugget
// is made of (Kernel.py:232-234, 272-274: sum_rows = np.sum(np.abs(K), axis=1)).  One wave per row i: the row part
// A(i, 0..i) is a strided read, the column part A(i+1.., i) a contiguous one; lanes add in a fixed order, so the result is
// reproducible from run to run.  HBM-read bound: N^2 / 2 entries, the strided half in 64-byte sectors.
__global__ void __launch_bounds__(256) abs_rowsum_kernel(const double* __restrict__ A, int ld, int N, double scale,
                                                         double* __restrict__ out) {
  const int i = (int)(((size_t)blockIdx.x * 256 + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if (i >= N) return;
  double s = 0.0;
  for (int j = lane; j <= i; j += 64) s += fabs(A[(size_t)j * ld + i]);
  const double* col = A + (size_t)i * ld;
  for (int j = i + 1 + lane; j < N; j += 64) s += fabs(col[j]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) out[i] = s * scale;
}

void gpg_launch_abs_rowsum(gpg_ctx* c, double scale, double* out_dev) {
  hipLaunchKernelGGL(abs_rowsum_kernel, dim3((unsigned)(((size_t)c->N * 64 + 255) / 256)), dim3(256), 0, c->stream, c->A, c->ld, c->N,
                     scale, out_dev);
}

void gpg_launch_cross(gpg_ctx* c, const AsmParams& p, int nx, int nxp) {
  if (p.kernel == GPG_KERNEL_SQEXP) launch_cross_d<GPG_KERNEL_SQEXP>(c, p, nx, nxp);
  else if (p.kernel == GPG_KERNEL_RATQU) launch_cross_d<GPG_KERNEL_RATQU>(c, p, nx, nxp);
  else launch_cross_d<GPG_KERNEL_MA5F2>(c, p, nx, nxp);
}

// d K / d theta_k (and d K / d alpha for the rational quadratic kernel) of the kernel matrix of ONE point set with itself, from
// its difference tensor -- calc_KernBase_grad_th / calc_KernGrad_grad_th / calc_Kern*_grad_alpha of the reference's kernel table
// (KernelSqExp.py:91-123, 470-568; KernelMatern5f2.py:100-135, 532-642; KernelRatQuad.py:96-163, 640-840), materialised:
// out_th [d, N, N], out_al [N, N], row-major, N = n (d + 1) with gradients.  One thread per point pair (a, b), R = x_a - x_b:
// block (0, 0) is even in R, blocks (i+1, 0) odd -- block (0, i+1) at (a, b) is minus block (i+1, 0) at (a, b) -- blocks
// (i+1, j+1) even and symmetric in (i, j).  The formulas are those of grad_contract_kernel (gradient.hip), which contracts the
// same derivatives without storing them; the Matern-5/2 value block uses the exact derivative -(1/2) R_k^2 (5/3)(1 + sqrt5 nu) e^(-sqrt5 nu)
// (the reference's gradient-free variant differentiates the exponential only, KernelMatern5f2.py:100-135: see tests/tolerances.py).
template <int KERN>
__global__ void __launch_bounds__(256) rtensor_kern_dhp_kernel(RtParams P, const double* __restrict__ Rt, double* __restrict__ out_th,
                                                               double* __restrict__ out_al) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int n = P.n1, d = P.d;
  const long long np = (long long)n * n;
  if (t >= np) return;
  const int a = (int)(t / n), b = (int)(t - (long long)a * n);
  const size_t N = P.use_grad ? (size_t)n * (d + 1) : (size_t)n, NN = N * N;
  double R[GPG_MAX_DIM], th[GPG_MAX_DIM];
  double s = 0.0;
  for (int k = 0; k < d; ++k) { R[k] = Rt[(size_t)k * np + t]; th[k] = P.theta[k]; s += th[k] * (R[k] * R[k]); }
  const double sqrt5 = sqrt(5.0), al = P.hp_kernel;
  // f0 = value, f1 / f2 / f3: the factors the first / second / third derivative levels multiply (E, M1, ... of gradient.hip)
  double f0, f1 = 0.0, f2, f3 = 0.0, inu = 0.0, G0 = 0.0, G1 = 0.0, G2 = 0.0;
  const double s1 = 1.0 + 1.0 / al, s2 = s1 * (1.0 + 2.0 / al), cq = 4.0 * s1, dcq = 4.0 / (al * al);
  if (KERN == GPG_KERNEL_SQEXP) {
    f2 = exp(-s); f0 = f2;
  } else if (KERN == GPG_KERNEL_RATQU) {
    const double Bq = 1.0 + s / al;
    f0 = pow(Bq, -al); f1 = pow(Bq, -al - 1.0); f2 = pow(Bq, -al - 2.0); f3 = pow(Bq, -al - 3.0);
    const double lnB = log(Bq), om = 1.0 - 1.0 / Bq;
    G0 = -lnB + om; G1 = -lnB + s1 * om; G2 = -lnB + (1.0 + 2.0 / al) * om;
  } else {
    const double nu = sqrt(s);
    f2 = exp(-sqrt5 * nu);
    f1 = ((5.0 / 3.0) * (1.0 + sqrt5 * nu)) * f2;
    f0 = (1.0 + sqrt5 * nu + (5.0 / 3.0) * (nu * nu)) * f2;
    inu = 1.0 / fmax(nu, 1e-16);
  }
  auto put = [&](size_t r, size_t c, int k, double v) { out_th[(size_t)k * NN + r * N + c] = v; };
  // ---- block (0, 0)
  for (int k = 0; k < d; ++k) {
    const double rk2 = R[k] * R[k];
    if (out_th) put(a, b, k, KERN == GPG_KERNEL_SQEXP ? -rk2 * f2 : KERN == GPG_KERNEL_RATQU ? -rk2 * f1 : -0.5 * (rk2 * f1));
  }
  if (out_al) out_al[(size_t)a * N + b] = KERN == GPG_KERNEL_RATQU ? f0 * G0 : 0.0;
  if (!P.use_grad) return;
  for (int i = 0; i < d; ++i) {
    const size_t ri = (size_t)(i + 1) * n + a, ci = (size_t)(i + 1) * n + b;
    // ---- blocks (i+1, 0) and (0, i+1)
    const double v = KERN == GPG_KERNEL_SQEXP ? ((-2.0 * th[i]) * R[i]) * f2 : KERN == GPG_KERNEL_RATQU ? ((-2.0 * th[i]) * R[i]) * f1 : ((-th[i]) * R[i]) * f1;
    if (out_th)
      for (int k = 0; k < d; ++k) {
        const double rk2 = R[k] * R[k];
        double dv;
        if (KERN == GPG_KERNEL_SQEXP) dv = -rk2 * v + (k == i ? -2.0 * R[i] * f2 : 0.0);
        else if (KERN == GPG_KERNEL_RATQU) dv = (2.0 * s1) * th[i] * R[i] * rk2 * f2 + (k == i ? -2.0 * R[i] * f1 : 0.0);
        else dv = (25.0 / 6.0) * th[i] * R[i] * rk2 * f2 + (k == i ? -R[i] * f1 : 0.0);
        put(ri, b, k, dv);
        put(a, ci, k, -dv);
      }
    if (out_al) {
      const double da = KERN == GPG_KERNEL_RATQU ? v * G1 : 0.0;
      out_al[ri * N + b] = da;
      out_al[(size_t)a * N + ci] = -da;
    }
    // ---- blocks (i+1, j+1) and (j+1, i+1), j <= i
    for (int j = 0; j <= i; ++j) {
      const size_t rj = (size_t)(j + 1) * n + a, cj = (size_t)(j + 1) * n + b;
      const double rij = R[i] * R[j];
      double w, daw = 0.0;
      if (KERN == GPG_KERNEL_SQEXP) w = i == j ? (2.0 * th[i] - (4.0 * (th[i] * th[i])) * rij) * f2 : ((-4.0 * th[j]) * th[i]) * (rij * f2);
      else if (KERN == GPG_KERNEL_RATQU) {
        w = i == j ? (2.0 * th[i]) * f1 - ((cq * (th[i] * th[i])) * rij) * f2 : (((-cq) * th[j]) * th[i]) * rij * f2;
        daw = (i == j ? (2.0 * th[i]) * f1 * G1 : 0.0) + (dcq - cq * G2) * th[i] * th[j] * rij * f2;
      } else w = i == j ? th[i] * f1 - (((25.0 / 3.0) * (th[i] * th[i])) * rij) * f2 : ((((-(25.0 / 3.0)) * th[j]) * th[i]) * rij) * f2;
      if (out_th)
        for (int k = 0; k < d; ++k) {
          const double rk2 = R[k] * R[k];
          double dw;
          if (KERN == GPG_KERNEL_SQEXP) {
            dw = -rk2 * w;
            if (i == j) dw += k == i ? (2.0 - 8.0 * th[i] * rij) * f2 : 0.0;
            else dw += (k == i ? -4.0 * th[j] * rij * f2 : 0.0) + (k == j ? -4.0 * th[i] * rij * f2 : 0.0);
          } else if (KERN == GPG_KERNEL_RATQU) {
            dw = (4.0 * s2) * th[i] * th[j] * rij * rk2 * f3;
            if (i == j) dw += -(2.0 * s1) * th[i] * rk2 * f2 + (k == i ? 2.0 * f1 - (8.0 * s1) * th[i] * rij * f2 : 0.0);
            else dw += (k == i ? -(4.0 * s1) * th[j] * rij * f2 : 0.0) + (k == j ? -(4.0 * s1) * th[i] * rij * f2 : 0.0);
          } else {
            dw = ((25.0 * sqrt5 / 6.0) * th[i] * th[j]) * rij * rk2 * inu * f2;
            if (i == j) dw += -((25.0 / 6.0) * th[i]) * rk2 * f2 + (k == i ? f1 - ((50.0 / 3.0) * th[i]) * rij * f2 : 0.0);
            else dw += (k == i ? -((25.0 / 3.0) * th[j]) * rij * f2 : 0.0) + (k == j ? -((25.0 / 3.0) * th[i]) * rij * f2 : 0.0);
          }
          put(ri, cj, k, dw);
          if (i != j) put(rj, ci, k, dw);
        }
      if (out_al) {
        out_al[ri * N + cj] = daw;
        if (i != j) out_al[rj * N + ci] = daw;
      }
    }
  }
}

// d (d K / d x1) / d x1: the x-derivative entries of the kernel table -- calc_KernBase_hess_x [d, n1 d, n2] and calc_KernGrad_grad_x
// [d, n1 d, n2 + n2g d] (KernelSqExp.py:48-88, 412-468; KernelMatern5f2.py:53-97, 452-530; KernelRatQuad.py:51-131, 556-638) from the
// difference tensor R = X1 - X2 of two point sets: out[k][i n1 + a][b] = d2 K(a, b) / d x1_i d x1_k and, in the gradient columns
// (j, b), d3 K / d x1_i d x1_k d x2_j.  The formulas are those of hess_contract_kernel (solve.hip), which contracts them for the
// posterior Hessian without storing them (there R = x_train - x_query, hence the opposite sign of the odd, third-order entries).
template <int KERN>
__global__ void __launch_bounds__(256) rtensor_kern_hess_x_kernel(RtParams P, const double* __restrict__ Rt, const int* __restrict__ gpos2,
                                                                  double* __restrict__ out) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long np = (long long)P.n1 * P.n2;
  if (t >= np) return;
  const int a = (int)(t / P.n2), b = (int)(t - (long long)a * P.n2);
  const int d = P.d, n1 = P.n1, n2 = P.n2;
  const size_t C = P.use_grad ? (size_t)n2 + (size_t)P.n2g * d : (size_t)n2, Rw = (size_t)n1 * d;
  double tR[GPG_MAX_DIM], th[GPG_MAX_DIM];
  double s = 0.0;
  for (int k = 0; k < d; ++k) { const double r = Rt[(size_t)k * np + t]; th[k] = P.theta[k]; tR[k] = th[k] * r; s += tR[k] * r; }
  const double sqrt5 = sqrt(5.0), al = P.hp_kernel, s1 = 1.0 + 1.0 / al, s2 = s1 * (1.0 + 2.0 / al);
  double E, M1 = 0.0, F3 = 0.0, c3 = 0.0;
  if (KERN == GPG_KERNEL_SQEXP) E = exp(-s);
  else if (KERN == GPG_KERNEL_RATQU) { const double Bq = 1.0 + s / al; M1 = pow(Bq, -al - 1.0); E = pow(Bq, -al - 2.0); F3 = pow(Bq, -al - 3.0); }
  else { const double nu = sqrt(s); E = exp(-sqrt5 * nu); M1 = ((5.0 / 3.0) * (1.0 + sqrt5 * nu)) * E; c3 = sqrt5 / fmax(nu, 1e-16); }
  const int gb = P.use_grad ? gpos2[b] : -1;
  for (int k = 0; k < d; ++k)
    for (int i = 0; i < d; ++i) {
      double* row = out + ((size_t)k * Rw + (size_t)i * n1 + a) * C;
      double v;
      if (KERN == GPG_KERNEL_SQEXP) v = (4.0 * tR[i] * tR[k] - (i == k ? 2.0 * th[i] : 0.0)) * E;
      else if (KERN == GPG_KERNEL_RATQU) v = (4.0 * s1) * tR[i] * tR[k] * E - (i == k ? 2.0 * th[i] * M1 : 0.0);
      else v = (25.0 / 3.0) * tR[i] * tR[k] * E - (i == k ? th[k] * M1 : 0.0);
      row[b] = v;
      if (gb < 0) continue;
      for (int j = 0; j < d; ++j) {
        const double lin = (i == k ? th[i] * tR[j] : 0.0) + (j == k ? th[j] * tR[i] : 0.0) + (i == j ? th[i] * tR[k] : 0.0);
        const double cub = tR[i] * tR[j] * tR[k];
        double w;
        if (KERN == GPG_KERNEL_SQEXP) w = (-4.0 * lin + 8.0 * cub) * E;
        else if (KERN == GPG_KERNEL_RATQU) w = -(4.0 * s1) * lin * E + (8.0 * s2) * cub * F3;
        else w = -(25.0 / 3.0) * E * (lin - c3 * cub);
        row[(size_t)n2 + (size_t)j * P.n2g + gb] = w;
      }
    }
}

int gpg_kern_rtensor_hess_x_run(int kernel, int d, int n1, int n2, int n2g, int use_grad, const double* theta, double hp_kernel,
                                const double* rt_dev, const int* gpos2_dev, double* out_dev, hipStream_t stream) {
  RtParams P;
  memset(&P, 0, sizeof(P));
  P.d = d; P.n1 = n1; P.n2 = n2; P.n1g = n1; P.n2g = n2g; P.use_grad = use_grad; P.hp_kernel = hp_kernel > 0.0 ? hp_kernel : 1.0;
  for (int k = 0; k < d; ++k) P.theta[k] = theta[k];
  const dim3 grid((unsigned)(((long long)n1 * n2 + 255) / 256));
  if (kernel == GPG_KERNEL_SQEXP) hipLaunchKernelGGL(rtensor_kern_hess_x_kernel<GPG_KERNEL_SQEXP>, grid, dim3(256), 0, stream, P, rt_dev, gpos2_dev, out_dev);
  else if (kernel == GPG_KERNEL_RATQU) hipLaunchKernelGGL(rtensor_kern_hess_x_kernel<GPG_KERNEL_RATQU>, grid, dim3(256), 0, stream, P, rt_dev, gpos2_dev, out_dev);
  else hipLaunchKernelGGL(rtensor_kern_hess_x_kernel<GPG_KERNEL_MA5F2>, grid, dim3(256), 0, stream, P, rt_dev, gpos2_dev, out_dev);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

int gpg_kern_rtensor_dhp_run(int kernel, int d, int n, int use_grad, const double* theta, double hp_kernel, const double* rt_dev,
                             double* out_th_dev, double* out_al_dev, hipStream_t stream) {
  RtParams P;
  memset(&P, 0, sizeof(P));
  P.d = d; P.n1 = n; P.n2 = n; P.n1g = n; P.n2g = n; P.use_grad = use_grad; P.hp_kernel = hp_kernel > 0.0 ? hp_kernel : 1.0;
  for (int k = 0; k < d; ++k) P.theta[k] = theta[k];
  const dim3 grid((unsigned)(((long long)n * n + 255) / 256));
  if (kernel == GPG_KERNEL_SQEXP) hipLaunchKernelGGL(rtensor_kern_dhp_kernel<GPG_KERNEL_SQEXP>, grid, dim3(256), 0, stream, P, rt_dev, out_th_dev, out_al_dev);
  else if (kernel == GPG_KERNEL_RATQU) hipLaunchKernelGGL(rtensor_kern_dhp_kernel<GPG_KERNEL_RATQU>, grid, dim3(256), 0, stream, P, rt_dev, out_th_dev, out_al_dev);
  else hipLaunchKernelGGL(rtensor_kern_dhp_kernel<GPG_KERNEL_MA5F2>, grid, dim3(256), 0, stream, P, rt_dev, out_th_dev, out_al_dev);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// Host side of gpg_kern_rtensor (api.hip): device buffers are temporaries of the call.
int gpg_kern_rtensor_run(int kernel, int d, int n1, int n2, int n1g, int n2g, int use_grad, const double* theta, double hp_kernel,
                         const double* rt_dev, const int* gpos1_dev, const int* gpos2_dev, double* out_dev, hipStream_t stream) {
  RtParams P;
  memset(&P, 0, sizeof(P));
  P.d = d; P.n1 = n1; P.n2 = n2; P.n1g = n1g; P.n2g = n2g; P.use_grad = use_grad; P.hp_kernel = hp_kernel;
  for (int k = 0; k < d; ++k) P.theta[k] = theta[k];
  const long long np = (long long)n1 * n2;
  const dim3 grid((unsigned)((np + 255) / 256));
  if (kernel == GPG_KERNEL_SQEXP) hipLaunchKernelGGL(rtensor_kern_kernel<GPG_KERNEL_SQEXP>, grid, dim3(256), 0, stream, P, rt_dev, gpos1_dev, gpos2_dev, out_dev);
  else if (kernel == GPG_KERNEL_RATQU) hipLaunchKernelGGL(rtensor_kern_kernel<GPG_KERNEL_RATQU>, grid, dim3(256), 0, stream, P, rt_dev, gpos1_dev, gpos2_dev, out_dev);
  else hipLaunchKernelGGL(rtensor_kern_kernel<GPG_KERNEL_MA5F2>, grid, dim3(256), 0, stream, P, rt_dev, gpos1_dev, gpos2_dev, out_dev);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
