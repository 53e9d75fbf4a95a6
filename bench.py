#!/usr/bin/env python3
"""Headline benchmark: marginal-likelihood evaluations per second, gradient-enhanced SqExp GP,
n=2000, d=8 (K is 18000 x 18000), fp64, on N MI355X of one node (BASELINE.json metric / configs[2]).

A "step" is one likelihood evaluation (assembly + preconditioner + nugget + Cholesky + GLS mean +
ln det + r'K^-1 r) of one restart row on each GPU; the data set is resident in HBM before the timed
region (set_data excluded).  Multi-GPU is weak scaling over independent restart rows
(SURVEY.md 8e): the global restart table has `steps` rows per rank, `select_best_restart` (the product's
own sharding function) gives every rank its contiguous block, each rank evaluates it on its own device,
then ONE all_gather (RCCL) of the ln_lkd values selects the best row -- no data-path collective.

    python bench.py --gpus 1 --steps 8 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus 1 --force-dist      # one rank, but through torch.distributed / RCCL like N > 1
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
PARITY_RTOL = 1e-8             # ln_lkd of row 0 against the CPU oracle (tests/tolerances.py LN_LKD_RTOL)
# HBM traffic of the dominant kernel: NOT a constant of this file.  tools/pmc_collect.sh runs this very command under
# rocprofv3 --pmc (FETCH_SIZE and WRITE_SIZE in separate passes) and tools/pmc_summarize.py writes the per-launch
# figures (FETCH_SIZE doubled per MI355X_MICROARCH.md: gfx950 counts wide streamed reads at half their bytes) into
# this file, keyed by config and matrices per launch; `roofline.traffic` is read from it, or null when no entry matches.
PMC_SUMMARY = os.path.join(ROOT, "profiles", "pmc_traffic.json")


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


def host_description():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.lower().startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"nproc": os.cpu_count(), "cpu_model": model}


def cpu_baseline(cfg, n, d, kernel, X, f, g, hp_row, threads_to_try=(16, 64, 128)):
    """Oracle (NumPy/SciPy port of the reference's CPU path) timed on this box's host cores, rank 0, N=1.
    (1) evaluation with element-wise preconditioner scaling + LAPACK dpotrf/dpotrs -- the best-practice CPU
        path, reported as `value` so the GPU/CPU ratio is not inflated by (2); at full size for cfg2 / cfg3, on a
        bounded sample extrapolated with N^3 for cfg5 (N = 68000 does not finish in the budget of a bench run).
        Timed at every BLAS thread count of `threads_to_try` that the box has CPUs for; `value` is the BEST of
        them (`cores` = its thread count), `threads_tried` keeps all of them;
    (2) the reference as written (five dense diag-matrix GEMMs, Kernel.py:224-252) on a bounded sample at the
        best thread count, extrapolated with N^3.
    Returns (record, oracle result of the full-size evaluation or None)."""
    from oracle import gp_oracle as orc
    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # pragma: no cover
        threadpool_limits = None
    noisy = cfg == "cfg5"
    ncpu = os.cpu_count() or 1
    tries = sorted({min(t, ncpu) for t in threads_to_try})

    def one(ns, as_written=False, reps=1):
        Xs, fs, gs = X[:ns], f[:ns], g[:ns]
        ys = orc.make_data_vec(fs, gs)
        Ns = ys.size
        _, eta = orc.calc_nugget(ns, d, kernel, True, "precon")
        theta = 10.0 ** hp_row[:d]
        if noisy:
            nv = orc.calc_noise_vec(ns, d, True, np.full(ns, 1e-2), np.full((ns, d), 1e-1), None, None)
            kw = dict(varK=10.0 ** hp_row[d])
        else:
            nv, kw = np.zeros(Ns), {}
        best, r = np.inf, None
        for _ in range(reps):
            t0 = time.perf_counter()
            r = orc.calc_lkd(Xs, ys, theta, kernel, True, "precon", eta, nv, noisy, as_written=as_written, **kw)
            best = min(best, time.perf_counter() - t0)
        return best, r, Ns

    def limited(t):
        return threadpool_limits(limits=t) if threadpool_limits is not None else None

    N = n * (d + 1)
    ns_full = 500 if cfg == "cfg5" else n                       # cfg5: N = 8500, ~2 s
    tried, r_full, Ns = {}, None, N
    for t in tries:
        lim = limited(t)
        one(64)                                                 # warm this pool's BLAS threads
        t_best, r, Ns = one(ns_full, reps=3 if cfg == "cfg2" else 1)
        tried[t] = t_best
        if cfg != "cfg5" and r_full is None:
            r_full = r
        if lim is not None and hasattr(lim, "unregister"):
            lim.unregister()
    threads = min(tried, key=tried.get)
    t_best = tried[threads]
    scale = (N / Ns) ** 3
    value = 1.0 / (t_best * scale)
    if cfg == "cfg5":
        sample = (f"bounded sample n={ns_full} (N={Ns}) of the {kernel} noisy workload with element-wise preconditioner "
                  f"scaling + LAPACK dpotrf/dpotrs: {t_best:.2f} s, extrapolated x(N/Ns)^3 = {scale:.0f}")
    else:
        sample = (f"{'best of 3' if cfg == 'cfg2' else '1'} full-size evaluation(s) (n={n}, d={d}, N={N}) of oracle/gp_oracle.py with "
                  f"element-wise preconditioner scaling + LAPACK dpotrf/dpotrs: {t_best:.3f} s")
    sample += f" at {threads} BLAS threads (best of {sorted(tried)})"
    lim = limited(threads)
    ns_aw = min(n, 500)
    t_aw, _, Ns_aw = one(ns_aw, as_written=True)
    scale_aw = (N / Ns_aw) ** 3
    if lim is not None and hasattr(lim, "unregister"):
        lim.unregister()
    rec = {"value": value, "unit": "evals/s", "cores": threads, "kind": "port", "sample": sample,
           "threads_tried": {str(t): {"seconds": tried[t] * scale, "value": 1.0 / (tried[t] * scale)} for t in sorted(tried)},
           "as_written_value": 1.0 / (t_aw * scale_aw),
           "as_written_sample": f"reference-as-written path (dense diag GEMMs, Kernel.py:224-252) at n={ns_aw} "
                                f"(N={Ns_aw}), {threads} BLAS threads: {t_aw:.2f} s"
                                + (f", extrapolated x(N/Ns)^3 = {scale_aw:.0f}" if scale_aw > 1.0 else ""),
           "blas_threads": threads}
    rec.update(host_description())
    if r_full is not None:
        rec["ln_lkd_cpu"] = r_full.ln_lkd
    return rec, r_full


def pmc_traffic(cfg, mats_per_launch, kernel_name):
    """Per-launch HBM bytes of the dominant kernel from the tracked PMC summary (or None)."""
    try:
        with open(PMC_SUMMARY) as fh:
            table = json.load(fh)
    except (OSError, ValueError):
        return None, None
    for e in table.get("entries", []):
        if e.get("config") == cfg and int(e.get("matrices_per_launch", -1)) == int(mats_per_launch) \
                and e.get("kernel", "").split("(")[0].strip() == kernel_name.split("(")[0].strip():
            return float(e["traffic_bytes_per_launch"]), e.get("source")
    return None, None


def self_launch(nranks):
    """`python3 bench.py --gpus N` without a launcher: this process -- which has not imported torch, loaded the library or
    touched HIP -- starts N fresh child processes of this very command, one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT in their environment, a free port), relays rank 0's JSON line and returns non-zero when any
    rank does.  No process that has initialised the GPU is ever replaced; under torch.distributed.run (WORLD_SIZE set)
    this function is not reached."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(nranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nranks), LOCAL_WORLD_SIZE=str(nranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # rank 0's stdout is the bench line; the other ranks print nothing there, but keep it off our stdout anyway
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    import threading
    buf = []
    reader = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rcs = [None] * nranks
    t_fail = None
    while any(c is None for c in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
                if rcs[r] not in (None, 0) and t_fail is None:
                    t_fail = time.time()                         # a rank failed: the others may be blocked in a collective
        if t_fail is not None and time.time() - t_fail > 20.0:
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    p.kill()                                     # this exact child, by handle
                    rcs[r] = p.wait()
        time.sleep(0.05)
    reader.join(timeout=10.0)
    out0 = buf[0] if buf else b""
    for ln in out0.decode(errors="replace").splitlines():        # ONE JSON line on our stdout; whatever else rank 0 (or a library in it,
        ln_s = ln.strip()                                        # e.g. gloo's connection notice) wrote there goes to stderr
        if ln_s.startswith("{") and ln_s.endswith("}"):
            sys.stdout.write(ln_s + "\n")
        elif ln_s:
            print(ln, file=sys.stderr)
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(rcs) if c != 0]
    if bad:
        print(f"[bench] rank(s) failed: {bad}", file=sys.stderr)
        return next(c for _, c in bad) if 0 < bad[0][1] < 256 else 1
    return 0


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
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed even for one rank: the N > 1 code path (RCCL init, sharded selection, "
                         "device-tensor all_gather) on a single GPU")
    ap.add_argument("--backend", default="auto", choices=["auto", "nccl", "gloo"],
                    help="auto: nccl (RCCL) with one rank per GPU, gloo when there are more ranks than GPUs (rehearsal)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))                         # plain `python3 bench.py --gpus N`: start the ranks ourselves

    import torch
    import torch.distributed as dist
    import gpgradpy_amd
    from gpgradpy_amd import multistart

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
    # one rank per GPU -> RCCL ("nccl"); more ranks than GPUs (a rehearsal on a 1-GPU box) -> gloo on host tensors
    backend = args.backend if args.backend != "auto" else ("nccl" if world <= ndev else "gloo")
    if backend == "nccl" and local_rank >= ndev:
        print(f"[bench] LOCAL_RANK={local_rank} but only {ndev} GPU(s) visible: RCCL needs one GPU per rank "
              f"(use --backend gloo to rehearse more ranks than GPUs)", file=sys.stderr)
        sys.exit(4)
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    use_dist = world > 1 or args.force_dist
    coll_dev = torch.device("cuda", dev_index) if backend == "nccl" else torch.device("cpu")
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:                       # --force-dist without a launcher: any free port
            import socket
            with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    group = dist.group.WORLD if use_dist else None

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
        "tile128": "tile128_chol_kernel (whole Cholesky as one dataflow launch, 128x128 tiles, left-looking)",
        "pair128": "pair128_chol_kernel (whole Cholesky of the batch as one dataflow launch, 128x128 tiles in pairs per 512-thread "
                   "workgroup, left-looking)"}

    # global restart table of the run: rank r's block is rows [8r, 8r+8) of the 64-row table (BASELINE cfg4: 8 per GPU),
    # cycled when steps > 8; shard_rows(world * k, world, r) hands rank r exactly its block
    def table_for(k):
        idx = [(8 * r + i) % hp_table.shape[0] for r in range(world) for i in range(k)]
        return hp_table[idx]

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    GP.reserve_batch(args.steps)                                  # setup (like set_data): workspaces of the batched launches
    if args.warmup > 0:
        multistart.select_best_restart(table_for(args.warmup), GP.calc_lkd_batch, group=group, device=coll_dev if use_dist else None)
    cats = list(gpgradpy_amd._lib.PROF_CATS) if args.prof_all else ["gemm_trail", "assembly"]
    GP.prof_enable(cats)
    GP.prof_read()
    local_s = [0.0]

    def timed_eval(rows):                                         # this rank's share: K evaluations queued back-to-back, one sync
        t = time.perf_counter()
        out = GP.calc_lkd_batch(rows)
        local_s[0] = time.perf_counter() - t
        return out

    barrier()
    t0 = time.perf_counter()
    # the product's own sharded selection: local block on this device, then the single collective + nanargmax
    _, ln_all, best = multistart.select_best_restart(table_for(args.steps), timed_eval, group=group,
                                                     device=coll_dev if use_dist else None)
    barrier()
    elapsed = time.perf_counter() - t0
    collective_ms = multistart.last_collective_s * 1e3 if use_dist else 0.0
    dom_kernel = DOM_KERNELS[GP.last_factor()[0]]                # what the library actually launched
    prof = GP.prof_read()
    GP.prof_enable([])
    per_rank_ms = local_s[0] * 1e3                               # one rank: one number; N ranks: [min, max]
    if use_dist:
        t = torch.tensor([elapsed, local_s[0], -local_s[0], collective_ms], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, collective_ms = float(t[0].item()), float(t[3].item())
        per_rank_ms = [-float(t[2].item()) * 1e3, float(t[1].item()) * 1e3]      # [min, max] over ranks of the local evaluation time

    rc = 0
    if rank == 0:
        tr = prof["gemm_trail"]
        achieved = tr["work"] / (tr["ms"] * 1e-3) * 1e-12 if tr["ms"] > 0 else 0.0
        # restart rows are batched into one factorisation launch (gpg_set_batch): flops per launch = B x N^3 / 3
        mats_per_launch = int(round(tr["work"] / max(1, tr["count"]) / (N ** 3 / 3.0))) if tr["count"] else 1
        asm = prof["assembly"]
        traffic, traffic_src = pmc_traffic(args.config, mats_per_launch, dom_kernel)
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
                       "n": n, "d": d, "N": N, "Npad": Npad, "kernel": kernel, "wellcond": "precon",
                       "evals_per_gpu": args.steps, "collective_backend": backend if use_dist else "none",
                       "parallelism": f"restarts sharded over {world} rank(s), one all_gather ({backend if use_dist else 'none'})"},
            "per_rank_ms": per_rank_ms, "collective_ms": collective_ms,
            "roofline": {"bound": "mfma", "kernel": dom_kernel,
                         "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
                         "matrices_per_launch": mats_per_launch,
                         "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                         "launches": tr["count"], "avg_launch_ms": tr["ms"] / max(1, tr["count"]),
                         "algorithmic_flops": tr["work"], "algorithmic_flops_def": "matrices x N^3 / 3 (N, not the padded size)",
                         "share_of_step_time": tr["ms"] * 1e-3 / elapsed if elapsed > 0 else None},
            "assembly": {"bound": "hbm", "achieved": asm["work"] / (asm["ms"] * 1e-3) * 1e-9 if asm["ms"] > 0 else 0.0,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "avg_launch_ms": asm["ms"] / max(1, asm["count"])},
            "best_row": int(best), "ln_lkd_best": float(ln_all[best]),
            "factor_fallbacks": GP.factor_fallbacks(),
        }
        if args.prof_all:
            result["kernel_ms_per_eval"] = {c: prof[c]["ms"] / args.steps for c in prof}
        if world == 1 and not args.no_cpu_baseline:
            cb, r = cpu_baseline(args.config, n, d, kernel, X, f, g, hp_table[0])
            result["cpu_baseline"] = cb
            if r is not None and args.steps >= 1:
                err = abs(ln_all[0] - r.ln_lkd) / abs(r.ln_lkd)
                result["parity_ln_lkd_rel_err_row0"] = err
                result["parity_rtol"] = PARITY_RTOL
                if not (err <= PARITY_RTOL):
                    print(f"[bench] PARITY FAILURE: ln_lkd of row 0 is {ln_all[0]!r} on the device, {r.ln_lkd!r} on the CPU oracle "
                          f"(rel {err:.3e} > {PARITY_RTOL:g})", file=sys.stderr)
                    rc = 5
        print(json.dumps(result))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
