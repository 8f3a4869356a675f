#!/usr/bin/env python3
"""bench.py -- log-posterior evaluations per second of the device-resident stretch-move sampler.

Workload (BASELINE.json configs[2], "C3"): N_design = 1000, N_obs = 500, 10 PCs, d = 6 parameters,
1024 walkers, RBF + White kernel with fixed hyper-parameters.  One "step" = one
stretch-move step = 2 half-ensemble updates = 1024 log-posterior evaluations.  Model state and the
ensemble are resident in HBM before the timed region; the chain is kept on the device.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

With N > 1 the SAME 1024 walkers are sharded over the ranks (strong scaling): each rank evaluates
its block of every half's proposals and the new log-probabilities are all-gathered with RCCL.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (trmm_vsq_kernel, the fp64
triangular GEMM): algorithmic FLOPs per launch (k * N^2 per evaluation, SURVEY 8d, x evaluations per
launch) / its mean launch duration measured with HIP events in a second pass of the same K steps.
`cpu_baseline` times the CPU oracle (reference-form per-walker log_posterior) on this host.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "bayesian-inference_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

N_DESIGN, N_OBS, N_PC, N_WALKERS = 1000, 500, 10, 1024
# Rehearsal knob (tests only): several ranks sharing ONE GPU cannot co-run the full-size kernels -- a rank's waves would
# spin on stores of a rank that is not scheduled -- so the 2-rank test of the N > 1 path runs a small ensemble.  The
# JSON line then carries "rehearsal": true and is not a measurement of the headline workload.
if os.environ.get("GPEMU_BENCH_REHEARSAL_WALKERS"):
    N_WALKERS = int(os.environ["GPEMU_BENCH_REHEARSAL_WALKERS"])
FP64_MATRIX_PEAK_TFLOPS = 78.6   # MI355X datasheet fp64 matrix rate (the local guide lists none)


def build_workload(device=0):
    """Synthetic C3 model, built with the product's own device fit path (setup, not timed):
    standardise + PCA (gpemu_pca_fit), then kernel matrix / Cholesky / alpha at the fixed
    hyper-parameters of SURVEY.md 8d (gpemu_fit_factor).  Same generator as the goldens."""
    from gpemu import estimators, synthetic
    from gpemu.fit import DeviceFit
    prob = synthetic.make_problem(N_DESIGN, N_OBS, seed=0)
    scaler, pca, Y_pca = estimators.scale_and_pca(prob["Y"], device=device)
    ls = (prob["hi"] - prob["lo"]) * 0.5
    noise = 0.05
    theta = np.log(np.r_[ls, noise])
    fit = DeviceFit(prob["design"], kernel_kind=0, has_noise=True, jitter=1e-10, device=device)
    Ls, alphas = [], []
    for i in range(N_PC):
        L, alpha, _ = fit.factor(Y_pca[:, i], theta)
        Ls.append(L)
        alphas.append(alpha)
    fit.close()
    cun = estimators.truncation_covariance(pca, N_PC, device=device)
    return dict(prob=prob, ls=np.tile(ls, (N_PC, 1)), noise=np.full(N_PC, noise), alpha=np.stack(alphas),
                L=np.stack(Ls), components=pca.components_[:N_PC], mean=scaler.mean_, scale=scaler.scale_,
                cun=cun)


def measure_predict(dm, n_samples=1024, reps=5):
    """Metric 2 (BASELINE.json): emulation.predict throughput with outputs resident in HBM.
    Algorithmic bytes: 8 (B F^2 + B F) out + 8 [k N (N+1)/2 + k N + N d + B d + F k + 2 F + F^2] in."""
    import torch
    from gpemu import synthetic
    B, F, k, N, d = n_samples, dm.F, dm.k, dm.N, dm.d
    dev = torch.device("cuda", dm.device)
    X = torch.from_numpy(synthetic.make_walkers(B, seed=2)).to(dev)
    cv = torch.empty((B, F), dtype=torch.float64, device=dev)
    cov = torch.empty((B, F, F), dtype=torch.float64, device=dev)
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        dm.predict_full_dev(X.data_ptr(), B, float(B), cv.data_ptr(), cov.data_ptr(), stream=st.cuda_stream)
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            dm.predict_full_dev(X.data_ptr(), B, float(B), cv.data_ptr(), cov.data_ptr(), stream=st.cuda_stream)
        e1.record(st)
        st.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nbytes = 8 * (B * F * F + B * F) + 8 * (k * N * (N + 1) // 2 + k * N + N * d + B * d + F * k + 2 * F + F * F)
    return {"metric": "GP predict GB/s (emulation.predict, full covariance, outputs in HBM)",
            "value": nbytes / (ms * 1e-3) / 1e9, "unit": "GB/s", "samples_per_s": B / (ms * 1e-3),
            "batch": B, "ms_per_batch": ms, "algorithmic_bytes": nbytes,
            "roofline": {"bound": "hbm", "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                         "frac": nbytes / (ms * 1e-3) / 1e9 / 8000.0}}


def measure_fit_c5(device=0):
    """BASELINE configs[4] ("C5": N_design = 5000, N_obs = 2000): the fit side on the device -- StandardScaler + PCA of the
    5000 x 2000 matrix and one log-marginal-likelihood evaluation at N = 5000 (kernel matrix, blocked Cholesky with MFMA
    SYRK trailing updates, blocked triangular inverse, alpha; with gradient: K^-1 = W^T W on the lower triangle + the
    contraction), timed against the fp64 MFMA peak.  Algorithmic FLOPs: N^3/3 (Cholesky) + N^3/3 (inverse of the factor)
    [+ N^3/3 (K^-1, symmetric, from the triangular W)], SURVEY.md 8d."""
    from gpemu import estimators, synthetic
    from gpemu.fit import DeviceFit
    N, F = 5000, 2000
    prob = synthetic.make_problem(N, F, seed=3)
    estimators.scale_and_pca(prob["Y"][:96, :40], device=device)      # the code objects' first load is not the PCA's time
    t0 = time.perf_counter()
    scaler, pca, Y_pca = estimators.scale_and_pca(prob["Y"], device=device)
    t_pca = time.perf_counter() - t0
    theta = np.log(np.r_[(prob["hi"] - prob["lo"]) * 0.5, 0.05])
    fit = DeviceFit(prob["design"], kernel_kind=0, has_noise=True, jitter=1e-10, device=device)
    y = Y_pca[:, 0]
    out = {"workload": "C5: N_design=5000 x N_obs=2000", "pca_5000x2000_ms": t_pca * 1e3}
    for grad in (False, True):
        fit.lml(y, theta, eval_gradient=grad)
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            fit.lml(y, theta, eval_gradient=grad)
        dt = (time.perf_counter() - t0) / reps
        flop = N ** 3 / 3 * (3 if grad else 2)
        key = "lml_grad" if grad else "lml"
        out[key + "_ms"] = dt * 1e3
        out[key + "_roofline"] = {"bound": "mfma", "achieved": flop / dt / 1e12, "peak": FP64_MATRIX_PEAK_TFLOPS,
                                  "unit": "TFLOP/s", "frac": flop / dt / 1e12 / FP64_MATRIX_PEAK_TFLOPS,
                                  "algorithmic_gflop": flop / 1e9}
    # the form the fit itself uses: several (target, theta) problems through one launch chain (gpemu_fit_lml_batch)
    nbatch = 8
    rng = np.random.default_rng(1)
    ys = np.stack([Y_pca[:, i % Y_pca.shape[1]] for i in range(nbatch)])
    thetas = np.stack([theta + 0.1 * rng.normal(size=theta.size) for _ in range(nbatch)])
    fit.lml_batch(ys, thetas)
    t0 = time.perf_counter()
    fit.lml_batch(ys, thetas)
    dt = (time.perf_counter() - t0) / nbatch
    flop = N ** 3
    out["lml_grad_batched_ms_per_problem"] = dt * 1e3
    out["lml_grad_batched_roofline"] = {"bound": "mfma", "achieved": flop / dt / 1e12, "peak": FP64_MATRIX_PEAK_TFLOPS,
                                        "unit": "TFLOP/s", "frac": flop / dt / 1e12 / FP64_MATRIX_PEAK_TFLOPS,
                                        "problems_per_launch_chain": nbatch}
    fit.close()
    return out


def measure_fit_c3(device=0, n_restarts=50):
    """The whole C3 emulator fit as the shipped configuration runs it (ref: emulation.py:169-172 with
    config/jet_substructure.yaml:80, n_restarts: 50): 10 GPs x 51 L-BFGS-B maximisations of the log-marginal
    likelihood at N = 1000, the optimisers advancing in lock step on the host, their evaluations batched on the device."""
    from gpemu import estimators, synthetic
    prob = synthetic.make_problem(N_DESIGN, N_OBS, seed=0)
    estimators.scale_and_pca(prob["Y"][:96, :40], device=device)      # the code objects' first load is not the PCA's time
    t0 = time.perf_counter()
    scaler, pca, Y_pca = estimators.scale_and_pca(prob["Y"], device=device)
    t_pca = time.perf_counter() - t0
    ls0 = prob["hi"] - prob["lo"]
    kern = estimators.ARDKernel(estimators.RBF_KIND, length_scale=ls0, length_scale_bounds=np.outer(ls0, (0.01, 100.0)),
                                noise_level=0.1, noise_level_bounds=(1e-3, 10.0))
    np.random.seed(2026)
    t0 = time.perf_counter()
    gps = estimators.fit_gps(prob["design"], Y_pca[:, :N_PC], kern, alpha=1e-10, n_restarts_optimizer=n_restarts,
                             device=device)
    dt = time.perf_counter() - t0
    return {"workload": f"C3 fit: 10 GPs x (1 + {n_restarts}) L-BFGS-B runs at N_design=1000", "seconds": dt,
            "pca_1000x500_ms": t_pca * 1e3, "lml_evaluations": int(getattr(gps[0], "n_lml_evaluations_", 0)),
            "mean_lml": float(np.mean([g.log_marginal_likelihood_value_ for g in gps]))}


def committed_traffic(world):
    """HBM/fabric bytes per launch of the dominant kernel from the newest committed rocprofv3 PMC passes
    (profiles/rNN*_traffic.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs, gfx950 correction applied; made by
    tools/collect_profiles.sh + tools/summarise_profiles.py).  The counters cannot be collected from inside this
    process; null when no file holds the kernel at this shape."""
    if world != 1:
        return None
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*traffic.json")), reverse=True):
        try:
            with open(path) as f:
                doc = json.load(f)
            kernels = doc.get("batches", {}).get("512") or doc.get("kernels", {})
            for name, k in kernels.items():
                if "trmm_vsq_dma_kernel" in name:
                    return {"bytes_per_launch": k["bytes_per_launch_corrected"],
                            "source": "profiles/" + os.path.basename(path)}
        except Exception:
            continue
    return None


def cpu_baseline(seconds_budget=20.0):
    """Reference-form per-walker log_posterior (oracle port, incl. the per-call recomputation of the
    truncation covariance) over a spawn Pool, one walker per task (ref: mcmc.py:77-85)."""
    import multiprocessing as mp
    ncores = min(os.cpu_count() or 1, 16)
    ctx = mp.get_context("spawn")
    per_worker = 2
    t_setup = time.time()
    with ctx.Pool(ncores, initializer=_cpu_init) as pool:
        pool.map(_cpu_eval, range(ncores))                      # warm-up: builds the model per worker
        t_setup = time.time() - t_setup
        n = ncores * per_worker
        t0 = time.time()
        pool.map(_cpu_eval, range(n), chunksize=1)
        dt = time.time() - t0
        # size the timed sample for ~seconds_budget/2 of wall time (>= 10 s of CPU work on >= 2 cores)
        n = int(max(n, min(20000, n * (seconds_budget / 2) / max(dt, 1e-3))))
        t0 = time.time()
        pool.map(_cpu_eval, range(n), chunksize=4)
        dt = time.time() - t0
    return {"value": n / dt, "unit": "log-posterior evals/s", "cores": ncores, "kind": "port",
            "sample": f"{n} per-walker reference-form evaluations of the C3 workload over a "
                      f"{ncores}-process spawn pool, BLAS threads = 1 per process "
                      f"(per-process model build {t_setup:.1f} s not timed)"}


_CPU = {}


def _cpu_init():
    os.environ["OMP_NUM_THREADS"] = "1"
    os.environ["OPENBLAS_NUM_THREADS"] = "1"
    try:
        from threadpoolctl import threadpool_limits
        _CPU["limit"] = threadpool_limits(1)
    except Exception:
        pass
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_util as GU
    from gpemu import synthetic
    model, prob, _ = GU.fixed_theta_model(N_DESIGN, N_OBS, N_PC, seed=0)
    _CPU.update(model=model, prob=prob, X=synthetic.make_walkers(N_WALKERS, seed=1))


def _cpu_eval(i):
    from oracle import gp_oracle as O
    p = _CPU["prob"]
    x = _CPU["X"][i % N_WALKERS]
    return float(O.log_posterior(x, {"g": _CPU["model"]}, p["lo"], p["hi"], p["y_exp"], p["y_err"])[0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fit", action="store_true", help="skip the fit-side legs (fit_c5, fit_c3)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="diagnostic: time ONE rank's share of an N-GPU step on this GPU (world-1 RCCL gather); "
                         "the printed value is NOT a throughput claim")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    if world > 1 or args.emulate_world:
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"] = "127.0.0.1"
            os.environ["MASTER_PORT"] = str(29400 + os.getpid() % 500)
        # RCCL; GPEMU_DIST_BACKEND=gloo only to rehearse the N > 1 path with several ranks on ONE GPU (which RCCL refuses)
        backend = os.environ.get("GPEMU_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    coll_dev = "cuda" if (not dist.is_initialized() or dist.get_backend() == "nccl") else "cpu"

    from gpemu import synthetic
    from gpemu.model import DeviceModel
    from gpemu.sampler import DeviceSampler

    wl = build_workload(dev_index)
    prob = wl["prob"]
    dm = DeviceModel(X_train=prob["design"], ls=wl["ls"], alpha=wl["alpha"], L=wl["L"],
                     components=wl["components"], scaler_mean=wl["mean"], scaler_scale=wl["scale"],
                     kernel_kind=0, noise=wl["noise"], cov_unexplained=wl["cun"], device=dev_index)
    dm.likelihood_setup(prob["y_exp"], prob["y_err"], prob["lo"], prob["hi"], 1.0)
    ds = DeviceSampler([dm], N_WALKERS, a=2.0, seed=1)
    ds.set_state(synthetic.make_walkers(N_WALKERS, seed=1))

    def run(steps, store=True):
        if args.emulate_world:
            ds.run_sharded(steps, store=store, force=True, emulate_world=args.emulate_world)
        elif world > 1:
            ds.run_sharded(steps, store=store)
        else:
            ds.run(steps, store=store)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dm.sync()

    ds.reserve(args.warmup + 2 * args.steps)      # chain storage for the warm-up, timed and profiled passes
    # Untimed pre-warm before the W warm-up steps: the clocks of an idle MI355X take tens of ms of load to
    # ramp; with a short W the ramp otherwise lands inside the timed region (seen as a bimodal ms_per_step).
    # Every rank must make the same number of passes (a sharded pass is a collective): rank 0's clock decides.
    t_pre = time.perf_counter()
    while True:
        run(100, store=False)
        barrier()
        go = time.perf_counter() - t_pre < 0.3
        if world > 1:
            flag = torch.tensor([1 if go else 0], dtype=torch.int32, device=coll_dev)
            dist.broadcast(flag, src=0)
            go = bool(int(flag.item()))
        if not go:
            break
    run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    evals = N_WALKERS * args.steps

    # second pass of the same K steps with HIP events around every launch of the hot kernels
    dm.profile(True)
    run(args.steps)
    barrier()
    prof = dm.profile_read()
    dm.profile(False)
    ms_tot, n_launch = prof["trmm_vsq"]
    roofline = None
    if n_launch > 0:
        split = args.emulate_world or world
        evals_per_launch = -(-(N_WALKERS // 2) // split)            # this rank's block of a half: 512 at N = 1
        flop_per_launch = N_PC * N_DESIGN ** 2 * evals_per_launch
        avg_s = ms_tot / n_launch * 1e-3
        achieved = flop_per_launch / avg_s / 1e12
        kern = "trmm_vsq_dma_kernel" if evals_per_launch > 128 else "trmm_vsq_small_kernel"
        roofline = {"bound": "mfma", "kernel": kern, "achieved": achieved,
                    "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / FP64_MATRIX_PEAK_TFLOPS, "traffic": committed_traffic(split),
                    "avg_launch_us": avg_s * 1e6, "launches": n_launch,
                    "kstar_avg_launch_us": prof["kstar"][0] / max(prof["kstar"][1], 1) * 1e3}

    predict = None
    if rank == 0:
        try:
            predict = measure_predict(dm)
        except Exception as e:
            predict = {"value": None, "error": repr(e)}

    fit_c5 = fit_c3 = None
    if rank == 0 and world == 1 and not args.no_fit and not args.emulate_world:
        try:
            fit_c5 = measure_fit_c5(dev_index)
            fit_c3 = measure_fit_c3(dev_index)
        except Exception as e:
            fit_c5 = fit_c5 or {"error": repr(e)}
            fit_c3 = fit_c3 or {"error": repr(e)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline()
        except Exception as e:  # the baseline is a reported figure; never lose the GPU line over it
            cpu = {"value": None, "error": repr(e)}

    nacc, iters, _ = ds.counts()
    if rank == 0:
        out = {"metric": "log-posterior evals/sec", "value": evals / dt, "unit": "evals/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "emulate_world": args.emulate_world or None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "C3: N_design=1000 x N_obs=500, 10 PCs, d=6, 1024-walker "
                                      "stretch-move MCMC, RBF+White fixed theta",
                          "n_walkers": N_WALKERS, "evals_per_step": N_WALKERS,
                          "parallelism": f"walkers sharded over {world} GPU(s)"},
               "acceptance_fraction_mean": float((nacc / max(iters, 1)).mean()),
               "roofline": roofline, "cpu_baseline": cpu, "gp_predict": predict, "fit_c5": fit_c5, "fit_c3": fit_c3}
        if os.environ.get("GPEMU_BENCH_REHEARSAL_WALKERS"):
            out["rehearsal"] = True
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()          # rank 0 also measured metric 2 / printed; tear down together
    ds.close()
    dm.close()
    if world > 1 or args.emulate_world:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
