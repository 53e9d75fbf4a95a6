// Adjoint gradient of the marginal log-likelihood with respect to the hyperparameters (gfx950).
//
// Reference: CalcLkd.py:170-177 (noise-free) / :230-235 (noisy):  ln_lkd_grad_k = sum_{r,c} G_k[r,c] Lam[r,c],
//   Lam = s * alpha alpha^T - 1/2 Kcov^-1      (s = pnlt'/N + 1/(2 varK) noise-free, 1/2 noisy),
//   G_k = d(regularised covariance)/d(hp_k):  GpHparaGrad.py:13-98 with the kernel derivatives of
//   KernelSqExp.py:470-568 and KernelMatern5f2.py:532-642 (exact d/d theta of every block entry).
// The reference materialises G as an [n_hp, N, N] tensor (20.7 GB at n=2000, d=8) and Kcov^-1 through
// cho_solve(eye(N)).  Here:
//   1. Kp^-1 = L^-T L^-1 on the MFMA kernels: W = I L^-T by the row-restricted forward sweep (W is upper
//      triangular, rows below the current panel are skipped), then Minv = - W W^T panel by panel with the
//      trailing-update kernel (lower triangle only).
//   2. one fused pass recomputes every G_k entry from the n x d design like the assembly kernel does and
//      contracts it on the fly with alpha alpha^T and Minv; per-workgroup partial sums are reduced in a
//      fixed order (deterministic).
// Output per hyperparameter slot k in [0, d+3): theta_0..theta_(d-1), varK, var_fval, var_fgrad:
//   g_aa[k] = sum G_k o (alpha alpha^T),  g_inv[k] = sum G_k o (-1/2 Kcov^-1);  the host combines s*g_aa + g_inv.
#include "gpg_internal.h"

namespace {

constexpr int kTBg = 16;          // b-points staged per workgroup (at most)
// Small designs: fewer b-points per workgroup, so that the launch has enough workgroups for the chip (n = 500: 64 -> 250).
static inline int grad_tb(int n) { return n <= 256 ? 1 : n <= 1024 ? 4 : kTBg; }

__device__ __forceinline__ double wave_sum_g(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// RHS row 0 <- RHS row 1 - beta * RHS row 0  ( = L^-1 P^-1 (y - V beta) ), beta from the reduction scalars
__global__ void combine_rows_kernel(double* __restrict__ A, int ld, int Npad, const double* __restrict__ scal) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Npad) return;
  const double beta = scal[1];
  double* col = A + (size_t)c * ld + Npad;
  col[0] = col[1] - beta * col[0];
}

// W (ldw x Npad, ldw even) <- identity, two entries per thread: one launch instead of a fill and a diagonal pass
__global__ void __launch_bounds__(256) set_identity_kernel(double* __restrict__ W, int ldw, int Npad) {
  const size_t half = (size_t)ldw / 2, total = half * Npad;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t col = i / half, r = 2 * (i - col * half);
    double2 v;
    v.x = r == col ? 1.0 : 0.0;
    v.y = r + 1 == col ? 1.0 : 0.0;
    *reinterpret_cast<double2*>(W + col * (size_t)ldw + r) = v;
  }
}

// z[r] = v[r] / invp[r] (zero tail): the vector whose "alpha" in grad_contract_kernel is v itself
__global__ void unscale_kernel(const double* __restrict__ v, const double* __restrict__ invp, int N, int Npad, double* __restrict__ z) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < Npad) z[r] = r < N ? v[r] / invp[r] : 0.0;
}

template <int KERN, int D>
__global__ void __launch_bounds__(256, 2) grad_contract_kernel(AsmParams P, const double* __restrict__ Xt,
                                                            const double* __restrict__ invp,
                                                            const double* __restrict__ zvec,
                                                            const double* __restrict__ Minv, int ldm,
                                                            double* __restrict__ partial, int tb) {
  constexpr int RQ = KERN == GPG_KERNEL_RATQU ? 1 : 0;
  constexpr int NS = D + 3 + RQ;            // theta_0..D-1, varK, var_fval, var_fgrad (, alpha of RatQu)
  __shared__ double xb[kTBg][D];
  __shared__ double ipb[kTBg][D + 1];
  __shared__ double alb[kTBg][D + 1];
  __shared__ int gpb[kTBg];
  __shared__ double red[2 * NS][4];
  const int n = P.n, ng = P.ng;
  const int a = blockIdx.x * 256 + threadIdx.x;
  const int b0 = blockIdx.y * tb;
  const int nblk = P.use_grad ? D + 1 : 1;
  for (int t = threadIdx.x; t < tb * D; t += 256) {
    int bb = t / D, k = t % D, b = b0 + bb;
    xb[bb][k] = b < n ? Xt[(size_t)k * n + b] : 0.0;
  }
  for (int t = threadIdx.x; t < tb * (D + 1); t += 256) {
    int bb = t / (D + 1), J = t % (D + 1), b = b0 + bb;
    double v = 0.0, al = 0.0;
    if (b < n && J < nblk) {
      const int gp = P.gpos[b];
      size_t c = (size_t)b;
      bool ok = true;
      if (J > 0) { ok = gp >= 0; c = (size_t)n + (size_t)(J - 1) * ng + gp; }
      if (ok) { v = invp[c]; al = zvec[c] * v; }             // alpha = zvec * invp
    }
    ipb[bb][J] = v;
    alb[bb][J] = al;
  }
  if (threadIdx.x < tb) gpb[threadIdx.x] = (b0 + threadIdx.x < n) ? P.gpos[b0 + threadIdx.x] : -1;
  __syncthreads();

  double ga[NS], gi[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) { ga[k] = 0.0; gi[k] = 0.0; }

  if (a < n) {
    const int gpa = P.gpos[a];
    double xa[D], ipa[D + 1], ala[D + 1], th[D];
#pragma unroll
    for (int k = 0; k < D; ++k) { xa[k] = Xt[(size_t)k * n + a]; th[k] = P.theta[k]; }
    ipa[0] = invp[a];
    ala[0] = zvec[a] * ipa[0];
#pragma unroll
    for (int I = 1; I <= D; ++I) {
      const bool ok = I < nblk && gpa >= 0;
      const size_t r = (size_t)n + (size_t)(I - 1) * ng + (ok ? gpa : 0);
      ipa[I] = ok ? invp[r] : 0.0;
      ala[I] = ok ? zvec[r] * ipa[I] : 0.0;
    }
    const double vK = P.varK, eta = P.eta;
    const int precon = P.precon;
    const int bend = min(tb, n - b0);
    const double sqrt5 = sqrt(5.0);
    const double rq_alpha = P.hp_kernel, rq_s1 = 1.0 + 1.0 / P.hp_kernel, rq_s2 = rq_s1 * (1.0 + 2.0 / P.hp_kernel);   // scalar1, scalar2 (:704-705)
    const double rq_c = 4.0 * rq_s1, rq_dc = 4.0 / (P.hp_kernel * P.hp_kernel);          // const (:529) and -d const / d alpha

    for (int bb = 0; bb < bend; ++bb) {
      const int b = b0 + bb;
      if (b > a) continue;                                   // lower triangle of point pairs only ...
      const bool diag_pt = (a == b);
      const int gpbb = gpb[bb];
      double R[D], E, M1 = 0.0, inu = 0.0, K00, F3 = 0.0, G0 = 0.0, G1 = 0.0, G2 = 0.0;
      if (KERN == GPG_KERNEL_SQEXP) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) { R[k] = xa[k] - xb[bb][k]; s -= th[k] * (R[k] * R[k]); }
        E = exp(s);
        K00 = E;
      } else if (KERN == GPG_KERNEL_RATQU) {
        // KernelRatQuad.py:656-667, 770-790: f_p = B^(-alpha-p), B = 1 + sum theta R^2 / alpha;
        //   d f_p / d theta_k = -(1 + p/alpha) R_k^2 f_(p+1),   d f_p / d alpha = f_p g_p,
        //   g_p = -ln B + (1 + p/alpha) (1 - 1/B)
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) { R[k] = xa[k] - xb[bb][k]; s += th[k] * (R[k] * R[k]); }
        const double Bq = 1.0 + s / rq_alpha;
        K00 = pow(Bq, -rq_alpha);
        M1 = pow(Bq, -rq_alpha - 1.0);          // f_1
        E = pow(Bq, -rq_alpha - 2.0);           // f_2
        F3 = pow(Bq, -rq_alpha - 3.0);          // f_3
        const double lnB = log(Bq), om = 1.0 - 1.0 / Bq;
        G0 = -lnB + om;
        G1 = -lnB + (1.0 + 1.0 / rq_alpha) * om;
        G2 = -lnB + (1.0 + 2.0 / rq_alpha) * om;
      } else {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) { R[k] = xa[k] - xb[bb][k]; s += th[k] * (R[k] * R[k]); }
        const double nu = sqrt(s);
        E = exp(-sqrt5 * nu);
        M1 = ((5.0 / 3.0) * (1.0 + sqrt5 * nu)) * E;
        K00 = (1.0 + sqrt5 * nu + (5.0 / 3.0) * (nu * nu)) * E;
        inu = 1.0 / fmax(nu, 1e-16);                         // KernelMatern5f2.py:592
      }
      // one matrix entry (I, J) at the point pair (a, b): value v, d v / d theta_k in dv[k]
      auto contract = [&](int I, int J, double v, const double (&dv)[D], double wgt, double da = 0.0) {
        const size_t r = I == 0 ? (size_t)a : (size_t)n + (size_t)(I - 1) * ng + gpa;
        const size_t c = J == 0 ? (size_t)b : (size_t)n + (size_t)(J - 1) * ng + gpbb;
        const size_t rr = r >= c ? r : c, cc = r >= c ? c : r;            // Minv holds the lower triangle
        const double laa = wgt * (ala[I] * alb[bb][J]);
        const double lin = Minv ? wgt * (0.5 * ((ipa[I] * Minv[rr + cc * (size_t)ldm]) * ipb[bb][J])) : 0.0;   // -1/2 Kcov^-1 = +1/2 P^-1 Minv P^-1
        const bool dg = (r == c);
        const double dscale = (dg && precon) ? (1.0 + eta) : 1.0;        // precon: eta * diag(d K_rr) rides along
#pragma unroll
        for (int k = 0; k < D; ++k) {
          const double g = vK * dv[k] * dscale;                           // GpHparaGrad.py:39-51 / :100-111
          ga[k] += g * laa;
          gi[k] += g * lin;
        }
        if (RQ) {                                                         // d / d alpha (GpHparaGrad.py:53-66 / :113-126)
          const double g = vK * da * dscale;
          ga[NS - 1] += g * laa;
          gi[NS - 1] += g * lin;
        }
        // d/d varK : Kern + eta diag(Kern) (precon) or Kern + eta I (base)   GpHparaGrad.py:128-137
        const double gv = dg ? (precon ? v * (1.0 + eta) : v + eta) : v;
        ga[D] += gv * laa;
        gi[D] += gv * lin;
        if (dg) {                                                         // GpHparaGrad.py:139-155
          const double gn = precon ? 1.0 + eta : 1.0;
          const int slot = I == 0 ? D + 1 : D + 2;
          ga[slot] += gn * laa;
          gi[slot] += gn * lin;
        }
      };
      double dv[D];
      // ---- block (0, 0)
#pragma unroll
      for (int k = 0; k < D; ++k)
        dv[k] = KERN == GPG_KERNEL_SQEXP ? -(R[k] * R[k]) * E
              : KERN == GPG_KERNEL_RATQU ? -(R[k] * R[k]) * M1 : -0.5 * ((R[k] * R[k]) * M1);
      contract(0, 0, K00, dv, diag_pt ? 1.0 : 2.0, K00 * G0);
      if (nblk > 1) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
          // ---- block (i+1, 0) at (a, b)  and, for a != b, block (i+1, 0) at (b, a) = -value (R -> -R)
          double v, dav = 0.0;
          if (KERN == GPG_KERNEL_SQEXP) {
            v = ((-2.0 * th[i]) * R[i]) * E;
#pragma unroll
            for (int k = 0; k < D; ++k) dv[k] = -(R[k] * R[k]) * v + (k == i ? -2.0 * R[i] * E : 0.0);
          } else if (KERN == GPG_KERNEL_RATQU) {
            v = ((-2.0 * th[i]) * R[i]) * M1;
#pragma unroll
            for (int k = 0; k < D; ++k)
              dv[k] = (2.0 * rq_s1) * th[i] * R[i] * (R[k] * R[k]) * E + (k == i ? -2.0 * R[i] * M1 : 0.0);   // :712-722
            dav = v * G1;
          } else {
            v = ((-th[i]) * R[i]) * M1;
#pragma unroll
            for (int k = 0; k < D; ++k)
              dv[k] = (25.0 / 6.0) * th[i] * R[i] * (R[k] * R[k]) * E + (k == i ? -R[i] * M1 : 0.0);
          }
          // entry (row (i+1, a), col (0, b)) and its mirror (row (i+1, b), col (0, a)) carry opposite signs;
          // each is an off-diagonal matrix entry (weight 2 for the symmetric pair)
          if (gpa >= 0) contract(i + 1, 0, v, dv, 2.0, dav);
          if (!diag_pt && gpbb >= 0) {
            // mirror: swap the roles of a and b.  value and derivative change sign; alpha / invp / Minv
            // indices are those of row (i+1, b), column (0, a)
            const size_t r = (size_t)n + (size_t)i * ng + gpbb, c = (size_t)a;
            const double laa = 2.0 * (alb[bb][i + 1] * ala[0]);
            const double lin = Minv ? 2.0 * (0.5 * ((ipb[bb][i + 1] * Minv[r + c * (size_t)ldm]) * ipa[0])) : 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) {
              const double g = vK * (-dv[k]);
              ga[k] += g * laa;
              gi[k] += g * lin;
            }
            if (RQ) {
              const double g = vK * (-dav);
              ga[NS - 1] += g * laa;
              gi[NS - 1] += g * lin;
            }
            ga[D] += (-v) * laa;
            gi[D] += (-v) * lin;
          }
          // ---- blocks (i+1, j+1), j <= i : symmetric in (a, b)
          if (gpa >= 0 && gpbb >= 0) {
#pragma unroll
            for (int j = 0; j < D; ++j) {
              if (j > i) continue;
              double w, daw = 0.0;
              if (KERN == GPG_KERNEL_RATQU) {
                if (i == j) {                                             // KernelRatQuad.py:544 / :709,729-736 / :820-830
                  w = (2.0 * th[i]) * M1 - ((rq_c * (th[i] * th[i])) * (R[i] * R[i])) * E;
#pragma unroll
                  for (int k = 0; k < D; ++k)
                    dv[k] = -(2.0 * rq_s1) * th[i] * (R[k] * R[k]) * E +
                            (4.0 * rq_s2) * (th[i] * th[i]) * (R[i] * R[i]) * (R[k] * R[k]) * F3 +
                            (k == i ? 2.0 * M1 - (8.0 * rq_s1) * th[i] * (R[i] * R[i]) * E : 0.0);
                  daw = (2.0 * th[i]) * M1 * G1 + (rq_dc - rq_c * G2) * (th[i] * th[i]) * (R[i] * R[i]) * E;
                } else {                                                  // :554 / :724-727,738-743
                  w = (((((-rq_c) * th[j]) * th[i]) * R[j]) * R[i]) * E;
#pragma unroll
                  for (int k = 0; k < D; ++k)
                    dv[k] = (4.0 * rq_s2) * th[i] * th[j] * (R[i] * R[j]) * (R[k] * R[k]) * F3 +
                            (k == i ? -(4.0 * rq_s1) * th[j] * (R[i] * R[j]) * E : 0.0) +
                            (k == j ? -(4.0 * rq_s1) * th[i] * (R[i] * R[j]) * E : 0.0);
                  daw = (rq_dc - rq_c * G2) * th[i] * th[j] * (R[i] * R[j]) * E;
                }
              } else if (KERN == GPG_KERNEL_SQEXP) {
                if (i == j) {
                  w = (2.0 * th[i] - (4.0 * (th[i] * th[i])) * (R[i] * R[i])) * E;
#pragma unroll
                  for (int k = 0; k < D; ++k)
                    dv[k] = -(R[k] * R[k]) * w + (k == i ? (2.0 - 8.0 * th[i] * (R[i] * R[i])) * E : 0.0);
                } else {
                  w = ((-4.0 * th[j]) * th[i]) * ((R[j] * R[i]) * E);
#pragma unroll
                  for (int k = 0; k < D; ++k)
                    dv[k] = -(R[k] * R[k]) * w + (k == i ? -4.0 * th[j] * (R[i] * R[j]) * E : 0.0) +
                            (k == j ? -4.0 * th[i] * (R[i] * R[j]) * E : 0.0);
                }
              } else {
                if (i == j) {
                  w = th[i] * M1 - (((25.0 / 3.0) * (th[i] * th[i])) * (R[i] * R[i])) * E;
#pragma unroll
                  for (int k = 0; k < D; ++k)
                    dv[k] = -((25.0 / 6.0) * th[i]) * (R[k] * R[k]) * E +
                            ((25.0 * sqrt5 / 6.0) * (th[i] * th[i])) * (R[i] * R[i]) * (R[k] * R[k]) * inu * E +
                            (k == i ? M1 - ((50.0 / 3.0) * th[i]) * (R[i] * R[i]) * E : 0.0);
                } else {
                  w = (((((-(25.0 / 3.0)) * th[j]) * th[i]) * R[j]) * R[i]) * E;
#pragma unroll
                  for (int k = 0; k < D; ++k)
                    dv[k] = ((25.0 * sqrt5 / 6.0) * th[i] * th[j]) * (R[i] * R[j]) * (R[k] * R[k]) * inu * E +
                            (k == i ? -((25.0 / 3.0) * th[j]) * (R[i] * R[j]) * E : 0.0) +
                            (k == j ? -((25.0 / 3.0) * th[i]) * (R[i] * R[j]) * E : 0.0);
                }
              }
              // matrix entries of this (a, b, i, j) orbit: (i+1,a ; j+1,b), (j+1,a ; i+1,b) and their two
              // transposes.  Blocks are symmetric in (a, b), so every distinct entry has the same value.
              if (i == j) {
                contract(i + 1, i + 1, w, dv, diag_pt ? 1.0 : 2.0, daw);
              } else {
                contract(i + 1, j + 1, w, dv, 2.0, daw);                 // (i+1,a ; j+1,b) + transpose
                if (!diag_pt) {                                          // (j+1,a ; i+1,b) + transpose
                  const size_t r = (size_t)n + (size_t)i * ng + gpbb, c = (size_t)n + (size_t)j * ng + gpa;
                  const double laa = 2.0 * (alb[bb][i + 1] * ala[j + 1]);
                  const double lin = Minv ? 2.0 * (0.5 * ((ipb[bb][i + 1] * Minv[r + c * (size_t)ldm]) * ipa[j + 1])) : 0.0;
#pragma unroll
                  for (int k = 0; k < D; ++k) {
                    const double g = vK * dv[k];
                    ga[k] += g * laa;
                    gi[k] += g * lin;
                  }
                  if (RQ) {
                    const double g = vK * daw;
                    ga[NS - 1] += g * laa;
                    gi[NS - 1] += g * lin;
                  }
                  ga[D] += w * laa;
                  gi[D] += w * lin;
                }
              }
            }
          }
        }
      }
    }
  }
  // block reduction -> partial[block][2 * NS]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const double s1 = wave_sum_g(ga[k]), s2 = wave_sum_g(gi[k]);
    if (lane == 0) { red[k][w] = s1; red[NS + k][w] = s2; }
  }
  __syncthreads();
  if (threadIdx.x < 2 * NS) {
    const double s = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
    partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (2 * GPG_GRAD_SLOTS_MAX) + threadIdx.x] = s;
  }
}

// fixed-order sum of the per-workgroup partials -> out[2 * NS]: one wave per value, lane l adds the partials of the
// workgroups l, l + 64, ... in order, then the 64 lane sums are combined by a fixed shuffle tree (deterministic)
__global__ void grad_final_reduce_kernel(const double* __restrict__ partial, int nblocks, int nvals, double* __restrict__ out) {
  const int k = blockIdx.x, lane = threadIdx.x;
  if (k >= nvals) return;
  double s = 0.0;
  for (int b = lane; b < nblocks; b += 64) s += partial[(size_t)b * (2 * GPG_GRAD_SLOTS_MAX) + k];
  s = wave_sum_g(s);
  if (lane == 0) out[k] = s;
}

template <int KERN>
void launch_contract_d(gpg_ctx* c, const AsmParams& p, double* partial, dim3 grid, const double* zvec, const double* Minv, int tb) {
#define CASE_D(DD)                                                                                               \
  case DD:                                                                                                       \
    hipLaunchKernelGGL((grad_contract_kernel<KERN, DD>), grid, dim3(256), 0, c->stream, p, c->Xt, c->invp, zvec,   \
                       Minv, c->Npad, partial, tb);                                                              \
    break;
  switch (p.d) {
    CASE_D(1) CASE_D(2) CASE_D(3) CASE_D(4) CASE_D(5) CASE_D(6) CASE_D(7) CASE_D(8)
    CASE_D(9) CASE_D(10) CASE_D(11) CASE_D(12) CASE_D(13) CASE_D(14) CASE_D(15) CASE_D(16)
  }
#undef CASE_D
}

}  // namespace

void gpg_launch_combine_rows(gpg_ctx* c, int slot) {
  hipLaunchKernelGGL(combine_rows_kernel, dim3((c->Npad + 255) / 256), dim3(256), 0, c->stream, c->A, c->ld, c->Npad,
                     c->scal + (size_t)slot * 8);
}

void gpg_launch_identity(gpg_ctx* c, double* W, int ldw) {
  const size_t total = (size_t)ldw / 2 * c->Npad;
  const size_t nb = (total + 255) / 256;
  hipLaunchKernelGGL(set_identity_kernel, dim3((unsigned)(nb < 65536 ? nb : 65536)), dim3(256), 0, c->stream, W, ldw, c->Npad);
}

// out_dev[0 .. ns) = g_aa, out_dev[ns .. 2 ns) = g_inv, ns = d + 3 (+ 1 for RatQu: alpha last)
// zvec = P alpha-like vector (the kernel forms alpha = zvec * invp), Minv = -(L L^T)^-1 or nullptr (then only the
// quadratic forms g_aa[k] = alpha^T G_k alpha are meaningful: gpg_dcov_quadform)
void gpg_launch_grad_contract(gpg_ctx* c, const AsmParams& p_in, double* partial, double* out_dev, const double* zvec,
                              const double* Minv) {
  AsmParams p = p_in;
  if (c->grad_eta >= 0.0) p.eta = c->grad_eta;   // the reference differentiates with self._etaK whatever nugget the matrix got
  const int tb = grad_tb(p.n);
  dim3 grid((p.n + 255) / 256, (p.n + tb - 1) / tb);
  if (p.kernel == GPG_KERNEL_SQEXP) launch_contract_d<GPG_KERNEL_SQEXP>(c, p, partial, grid, zvec, Minv, tb);
  else if (p.kernel == GPG_KERNEL_RATQU) launch_contract_d<GPG_KERNEL_RATQU>(c, p, partial, grid, zvec, Minv, tb);
  else launch_contract_d<GPG_KERNEL_MA5F2>(c, p, partial, grid, zvec, Minv, tb);
  const int ns = p.d + 3 + (p.kernel == GPG_KERNEL_RATQU ? 1 : 0);
  hipLaunchKernelGGL(grad_final_reduce_kernel, dim3(2 * ns), dim3(64), 0, c->stream, partial, (int)(grid.x * grid.y),
                     2 * ns, out_dev);
}

void gpg_launch_unscale(gpg_ctx* c, const double* v, double* z) {
  hipLaunchKernelGGL(unscale_kernel, dim3((c->Npad + 255) / 256), dim3(256), 0, c->stream, v, c->invp, c->N, c->Npad, z);
}

int gpg_grad_partial_blocks(const gpg_ctx* c) { const int tb = grad_tb(c->n); return ((c->n + 255) / 256) * ((c->n + tb - 1) / tb); }
