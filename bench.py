#!/usr/bin/env python3
"""Headline benchmark: marginal-likelihood evaluations per second, gradient-enhanced SqExp GP,
n=2000, d=8 (K is 18000 x 18000), fp64, on N MI355X of one node (BASELINE.json metric / configs[2]).

A "step" is one likelihood evaluation (assembly + preconditioner + nugget + Cholesky + GLS mean +
ln det + r'K^-1 r) of one restart row on each GPU; the data set is resident in HBM before the timed
region (set_data excluded).  Multi-GPU is weak scaling over independent restart rows
(SURVEY.md 8e): every rank evaluates `steps` rows of the shared restart table on its own device, then
ONE all_gather (RCCL) of the ln_lkd values selects the best row -- no data-path collective.

    python bench.py --gpus 1 --steps 8 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # v_mfma_f64_16x16x4_f64: 64 cycles / instr / SIMD measured (profiles/r01_fp64_pipe_probe.log)
                               # = 32 flop/clk/SIMD x 1024 SIMDs x 2.4 GHz; equals AMD's datasheet FP64 matrix figure
HBM_PEAK_GBS = 8000.0
# HBM bytes per launch of the dominant kernel at n=2000, d=8 (tile128_chol_kernel: ONE launch per evaluation) from
# the PMC passes of profiles/r01_d_pmc_summary.txt (rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes
# over this same command, 2 evaluations = 2 launches): WRITE_SIZE 7.10 GB / 2; FETCH_SIZE 114.96 GB / 2 raw,
# doubled per MI355X_MICROARCH.md (gfx950 counts wide streamed reads at half their bytes).  The left-looking
# tile sweep reads 2 x 128 x K x 8 B of finished columns per 128 x 128 tile: 128 GB per factorisation, i.e. the
# measured traffic is that operand stream with almost no L2 reuse between tiles (L2 hit 33 %); the matrix itself
# (1.3 GB) is read and written once.
PMC_TRAFFIC_BYTES_PER_LAUNCH_CFG3 = (2 * 114.96e9 + 7.10e9) / 2.0


def make_workload(n, d, cfg="cfg3"):
    """BASELINE.md section 3 synthetic inputs: X ~ U(-2,2)^(n x d) seed 0, Rosenbrock(a=10) values + gradients,
    restart table default_rng(1).uniform(-2.5, -0.5, (64, d)) = log10 theta (cfg2/3/4); cfg5: rows
    [log10 theta ~ U(-3,-1)^d, log10 varK ~ U(-1,1)] seed 2 with known noise std_f = 1e-2, std_g = 1e-1."""
    rng = np.random.default_rng(0)
    X = rng.uniform(-2.0, 2.0, (n, d))
    f = np.zeros(n)
    g = np.zeros((n, d))
    a = 10.0
    for k in range(d - 1):
        t = X[:, k + 1] - X[:, k] ** 2
        f += a * t ** 2 + (1 - X[:, k]) ** 2
        g[:, k] += -4 * a * X[:, k] * t - 2 * (1 - X[:, k])
        g[:, k + 1] += 2 * a * t
    if cfg == "cfg5":
        rng2 = np.random.default_rng(2)
        hp_table = np.hstack((rng2.uniform(-3.0, -1.0, (64, d)), rng2.uniform(-1.0, 1.0, (64, 1))))
    else:
        hp_table = np.random.default_rng(1).uniform(-2.5, -0.5, (64, d))
    return X, f, g, hp_table


def cpu_baseline(n, d, X, f, g, theta, threads=16):
    """Oracle (NumPy/SciPy port of the reference's CPU path) timed on this box's host cores, rank 0, N=1.
    (1) full-size evaluation with element-wise preconditioner scaling + LAPACK dpotrf/dpotrs -- the
        best-practice CPU path, reported as `value` so the GPU/CPU ratio is not inflated by (2);
    (2) the reference as written (five dense diag-matrix GEMMs, Kernel.py:224-252) on a bounded sample,
        extrapolated with N^3."""
    from oracle import gp_oracle as orc
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=threads)
    except Exception:  # pragma: no cover
        limiter = None
    y = orc.make_data_vec(f, g)
    N = y.size
    eb, eta = orc.calc_nugget(n, d, "SqExp", True, "precon")
    # warm BLAS threads
    Xs, fs, gs = X[:64], f[:64], g[:64]
    orc.calc_lkd(Xs, orc.make_data_vec(fs, gs), theta, "SqExp", True, "precon", eta, np.zeros(64 * (d + 1)), False)
    t0 = time.perf_counter()
    r = orc.calc_lkd(X, y, theta, "SqExp", True, "precon", eta, np.zeros(N), False)
    t_best = time.perf_counter() - t0
    ns = 500
    Ns = ns * (d + 1)
    Xs, fs, gs = X[:ns], f[:ns], g[:ns]
    ys = orc.make_data_vec(fs, gs)
    _, eta_s = orc.calc_nugget(ns, d, "SqExp", True, "precon")
    t0 = time.perf_counter()
    orc.calc_lkd(Xs, ys, theta, "SqExp", True, "precon", eta_s, np.zeros(Ns), False, as_written=True)
    t_aw = time.perf_counter() - t0
    scale = (N / Ns) ** 3
    if limiter is not None:
        limiter.unregister() if hasattr(limiter, "unregister") else None
    return {
        "value": 1.0 / t_best, "unit": "evals/s", "cores": threads, "kind": "port",
        "sample": f"1 full-size evaluation (n={n}, d={d}, N={N}) of oracle/gp_oracle.py with element-wise "
                  f"preconditioner scaling + LAPACK dpotrf/dpotrs: {t_best:.2f} s",
        "as_written_value": 1.0 / (t_aw * scale),
        "as_written_sample": f"reference-as-written path (dense diag GEMMs, Kernel.py:224-252) at n={ns} "
                             f"(N={Ns}): {t_aw:.2f} s, extrapolated x(N/Ns)^3 = {scale:.0f}",
        "ln_lkd_cpu": r.ln_lkd,
    }, r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=8)   # same batch size as the timed region: every factorisation launch of a default run is alike
    ap.add_argument("--config", default="cfg3", choices=["cfg2", "cfg3", "cfg5"],
                    help="BASELINE.json config: cfg2 n=500 d=4 SqExp, cfg3 n=2000 d=8 SqExp (headline), cfg5 n=4000 d=16 Ma5f2 noisy")
    ap.add_argument("--n", type=int, default=0)
    ap.add_argument("--d", type=int, default=0)
    ap.add_argument("--panel", type=int, default=0, help="panel width override")
    ap.add_argument("--lookahead", type=int, default=-1, help="1/0 force the two-stream look-ahead on/off (blocked schedule)")
    ap.add_argument("--factor-mode", default="auto", choices=["auto", "blocked", "tile64", "tile128"],
                    help="Cholesky schedule (include/gpgrad.h gpg_factor_mode); default: one dataflow launch")
    ap.add_argument("--batch", type=int, default=-1, help="restart rows per batched launch on small matrices (gpg_set_batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prof-all", action="store_true", help="time every kernel category (adds event overhead)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import gpgradpy_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: launch with torch.distributed.run", file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    if not torch.cuda.is_available():
        print("[bench] no GPU visible: the product path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    # one rank per GPU -> RCCL ("nccl"); more ranks than GPUs (a rehearsal on a 1-GPU box) -> gloo on host tensors
    backend = "nccl" if world <= ndev else "gloo"
    coll_dev = torch.device("cuda", dev_index) if backend == "nccl" else torch.device("cpu")
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    n0, d0 = {"cfg2": (500, 4), "cfg3": (2000, 8), "cfg5": (4000, 16)}[args.config]
    n, d = args.n or n0, args.d or d0
    kernel = "Ma5f2" if args.config == "cfg5" else "SqExp"
    X, f, g, hp_table = make_workload(n, d, args.config)
    N = n * (d + 1)
    GP = gpgradpy_amd.GaussianProcess(d, True, kernel, "precon", device=dev_index)
    if args.config == "cfg5":
        GP.set_data(X, f, np.full(n, 1e-2), g, np.full((n, d), 1e-1))   # known noise -> varK is a hyperparameter
    else:
        GP.set_data(X, f, np.zeros(n), g, np.zeros((n, d)))      # resident in HBM before the timed region
    if args.panel:
        GP.set_panel(args.panel)
    if args.lookahead >= 0:
        GP.set_lookahead(args.lookahead)
    GP.set_factor_mode(args.factor_mode)
    if args.batch >= 0:
        GP.set_batch(args.batch)
    Npad = -(-N // 128) * 128
    DOM_KERNELS = {
        "blocked": "gemm_dma_kernel<4> (trailing update C -= A_p A_p^T of the blocked schedule, 128x128 tiles)",
        "tile64": "tile_chol_kernel (whole Cholesky as one dataflow launch, 64x64 tiles, left-looking)",
        "tile128": "tile128_chol_kernel (whole Cholesky as one dataflow launch, 128x128 tiles, left-looking)"}

    # rank r owns rows [8r, 8r+8) of the 64-row table (BASELINE cfg4), cycled when steps > 8
    def rows_for(k):
        idx = [(8 * rank + i) % hp_table.shape[0] for i in range(k)]
        return hp_table[idx]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    GP.reserve_batch(args.steps)                                  # setup (like set_data): workspaces of the batched launches
    if args.warmup > 0:
        GP.calc_lkd_batch(rows_for(args.warmup))
    cats = list(gpgradpy_amd._lib.PROF_CATS) if args.prof_all else ["gemm_trail", "assembly"]
    GP.prof_enable(cats)
    GP.prof_read()
    barrier()
    t0 = time.perf_counter()
    ln_local = GP.calc_lkd_batch(rows_for(args.steps))           # K evaluations queued back-to-back, one sync
    if world > 1:                                                # the single collective: gather ln_lkd, pick best
        buf = torch.from_numpy(ln_local).to(coll_dev)
        out = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(out, buf)
        ln_all = torch.cat(out).cpu().numpy()
    else:
        ln_all = ln_local
    barrier()
    elapsed = time.perf_counter() - t0
    best = int(np.nanargmax(ln_all))
    dom_kernel = DOM_KERNELS[GP.last_factor()[0]]                # what the library actually launched
    prof = GP.prof_read()
    GP.prof_enable([])
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        tr = prof["gemm_trail"]
        achieved = tr["work"] / (tr["ms"] * 1e-3) * 1e-12 if tr["ms"] > 0 else 0.0
        # restart rows are batched into one factorisation launch (gpg_set_batch): flops per launch = B x Npad^3 / 3
        mats_per_launch = int(round(tr["work"] / max(1, tr["count"]) / (Npad ** 3 / 3.0))) if tr["count"] else 1
        asm = prof["assembly"]
        result = {
            "metric": "marginal-likelihood evals/sec (grad-enh, n=2000 d=8)" if (n, d, args.config) == (2000, 8, "cfg3")
                      else f"marginal-likelihood evals/sec (grad-enh, n={n} d={d}, {args.config})",
            "value": world * args.steps / elapsed, "unit": "evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.config}: gradient-enhanced {kernel} GP, n={n} d={d} (K is {N}x{N}), "
                                   + ("known noise on f and grad f (varK a hyperparameter), " if args.config == "cfg5" else "noise-free, ")
                                   + "precon + nugget, value-only likelihood evaluation per restart row "
                                   "(rows of the BASELINE.md section 3 restart table, 8 per rank)",
                       "n": n, "d": d, "N": N, "kernel": kernel, "wellcond": "precon",
                       "evals_per_gpu": args.steps, "parallelism": f"restarts sharded over {world} rank(s), one all_gather ({backend if world > 1 else 'none'})"},
            "roofline": {"bound": "mfma", "kernel": dom_kernel,
                         "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
                         "matrices_per_launch": mats_per_launch,
                         "traffic": PMC_TRAFFIC_BYTES_PER_LAUNCH_CFG3 * mats_per_launch
                                    if (n, d, args.config, args.factor_mode) == (2000, 8, "cfg3", "auto") else None,
                         "traffic_unit": "bytes/launch = matrices per launch x the single-matrix PMC figure (FETCH_SIZE x2 + "
                                         "WRITE_SIZE, profiles/r01_d_pmc_summary.txt)",
                         "launches": tr["count"], "avg_launch_ms": tr["ms"] / max(1, tr["count"]),
                         "algorithmic_flops": tr["work"],
                         "share_of_step_time": tr["ms"] * 1e-3 / elapsed if elapsed > 0 else None},
            "assembly": {"bound": "hbm", "achieved": asm["work"] / (asm["ms"] * 1e-3) * 1e-9 if asm["ms"] > 0 else 0.0,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "avg_launch_ms": asm["ms"] / max(1, asm["count"])},
            "best_row": best, "ln_lkd_best": float(ln_all[best]),
        }
        if args.prof_all:
            result["kernel_ms_per_eval"] = {c: prof[c]["ms"] / args.steps for c in prof}
        if world == 1 and not args.no_cpu_baseline and args.config == "cfg3":
            theta0 = 10.0 ** hp_table[0]
            cb, r = cpu_baseline(n, d, X, f, g, theta0)
            result["cpu_baseline"] = cb
            result["parity_ln_lkd_rel_err_row0"] = abs(ln_local[0] - r.ln_lkd) / abs(r.ln_lkd) if args.steps >= 1 else None
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
