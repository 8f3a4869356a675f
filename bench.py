#!/usr/bin/env python3
"""bench.py -- log-posterior evaluations per second of the device-resident stretch-move sampler.

Workload (BASELINE.json configs[2], "C3"): N_design = 1000, N_obs = 500, 10 PCs, d = 6 parameters,
1024 walkers, RBF + White kernel with fixed hyper-parameters.  One "step" = one
stretch-move step = 2 half-ensemble updates = 1024 log-posterior evaluations.  Model state and the
ensemble are resident in HBM before the timed region; the chain is kept on the device.

    python bench.py [--gpus N --steps K --warmup W]        (N > 1: starts its N ranks itself, as child processes)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1, launcher environment)

With N > 1 the SAME 1024 walkers are sharded over the ranks (strong scaling): each rank evaluates
its block of every half's proposals and the new log-probabilities are exchanged -- timed once with one RCCL
all-gather per half-step and once with peer stores over xGMI (`transports`); `value` is the faster of the two,
`transport` names it, `ranks_seen` / `peer_selftest_per_rank` / `fallback_vote` say what the ranks really did.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (trmm_vsq_dma_kernel, the fp64
triangular GEMM): algorithmic FLOPs per launch (k * N^2 per evaluation, SURVEY 8d, x evaluations per
launch) / its mean launch duration measured with HIP events in a second pass of the same K steps
(`achieved` / `frac`); `step_achieved` / `step_frac` is SURVEY 8(d)'s own definition for the whole step:
evaluations per second x FLOP_eval (1.02e7 at C3) / peak -- the kernels around the GEMM count against it.
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


def build_workload(device=0, n_design=None, n_obs=None, n_pc=None, seed=0, kernel_kind=0, nu=np.inf):
    """Synthetic C3 model (or one of another size), built with the product's own device fit path (setup, not timed):
    standardise + PCA (gpemu_pca_fit), then kernel matrix / Cholesky / alpha at the fixed
    hyper-parameters of SURVEY.md 8d (gpemu_fit_factor).  Same generator as the goldens."""
    from gpemu import estimators, synthetic
    from gpemu.fit import DeviceFit
    N_DESIGN, N_OBS, N_PC = (n_design or globals()["N_DESIGN"], n_obs or globals()["N_OBS"], n_pc or globals()["N_PC"])
    prob = synthetic.make_problem(N_DESIGN, N_OBS, seed=seed)
    scaler, pca, Y_pca = estimators.scale_and_pca(prob["Y"], device=device)
    ls = (prob["hi"] - prob["lo"]) * 0.5
    noise = 0.05
    theta = np.log(np.r_[ls, noise])
    fit = DeviceFit(prob["design"], kernel_kind=kernel_kind, nu=nu, has_noise=True, jitter=1e-10, device=device)
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


def measure_shipped_shape(device=0, n_walkers=200, steps=3000, observable_blocks=False):
    """One chain at the size the reference ships (ref: config/jet_substructure.yaml: ~150 design points, d = 6, 200
    walkers, three emulation groups of 5 / 11 / 25 PCs, block-diagonal likelihood over the groups): the regime where a
    stretch-move step is a handful of ~10 us launches, nothing like C3.  Models built by the product's own device fit.
    observable_blocks: the covariance of a group block diagonal over its OBSERVABLES as well (ref: emulation.py:370-388),
    2 / 4 / 10 of them as in the shipped configuration (golden G7): a k x k factorisation per observable and proposal."""
    from gpemu import synthetic
    from gpemu.model import DeviceModel
    from gpemu.sampler import DeviceSampler
    dms = []
    n_blocks = {60: 2, 120: 4, 215: 10}
    for gi, (n_obs, n_pc) in enumerate([(60, 5), (120, 11), (215, 25)]):
        wl = build_workload(device, 150, n_obs, n_pc, seed=gi)
        prob = wl["prob"]
        dmg = DeviceModel(X_train=prob["design"], ls=wl["ls"], alpha=wl["alpha"], L=wl["L"], components=wl["components"],
                          scaler_mean=wl["mean"], scaler_scale=wl["scale"], kernel_kind=0, noise=wl["noise"],
                          cov_unexplained=wl["cun"], device=device)
        blocks = None
        if observable_blocks:
            blocks = [int(round(i * n_obs / n_blocks[n_obs])) for i in range(n_blocks[n_obs] + 1)]
        dmg.likelihood_setup(prob["y_exp"], prob["y_err"], prob["lo"], prob["hi"], 1.0, block_start=blocks)
        dms.append(dmg)
    ds = DeviceSampler(dms, n_walkers, seed=11)
    ds.set_state(synthetic.make_walkers(n_walkers, seed=3))
    ds.run(300, store=False)
    dms[0].sync()
    t0 = time.perf_counter()
    ds.run(steps, store=False)
    dms[0].sync()
    dt = time.perf_counter() - t0
    ds.close()
    for dmg in dms:
        dmg.close()
    return {"workload": f"shipped shape: N_design=150, groups of 5 + 11 + 25 PCs (60 + 120 + 215 observables"
                        + (" in 2 + 4 + 10 observable blocks" if observable_blocks else "") + f"), {n_walkers} walkers",
            "us_per_step": dt / steps * 1e6, "evals_per_s": n_walkers * steps / dt, "steps": steps}


def measure_steady(ds, dm, n_walkers, min_seconds=0.5, block=500):
    """The headline sampler again, for at least `min_seconds` of device time and whatever --steps says (VERDICT r4 item 4:
    the driver's --steps 20 is a 4.6 ms timed region): blocks of `block` steps until the clock says so, one timed region,
    chain stored as in the headline pass."""
    n_blocks = 0
    ds.run(50)
    dm.sync()
    t0 = time.perf_counter()
    while True:
        ds.run(block)
        n_blocks += 1
        if n_blocks >= 4:                    # (>= 2 000 steps before the clock is looked at: no sync inside the first ones)
            dm.sync()
            if time.perf_counter() - t0 >= min_seconds:
                break
    dm.sync()
    dt = time.perf_counter() - t0
    steps = n_blocks * block
    return {"steps": steps, "seconds": dt, "ms_per_step": dt / steps * 1e3, "value": n_walkers * steps / dt}


def measure_matern15(device=0, steps=1500):
    """The C3 workload with the kernel the reference SHIPS (ref: config/jet_substructure.yaml:58-61, rehlers.yaml:63-66:
    Matern nu = 1.5 + White) instead of the RBF the headline is defined on: same sizes, same fixed length scales and noise
    level, same sampler; `ms_per_step` and the cross-kernel's launch time (its base kernel is the only thing that changes)."""
    from gpemu import synthetic
    from gpemu.model import DeviceModel
    from gpemu.sampler import DeviceSampler
    wl = build_workload(device, kernel_kind=1, nu=1.5)
    prob = wl["prob"]
    dmm = DeviceModel(X_train=prob["design"], ls=wl["ls"], alpha=wl["alpha"], L=wl["L"], components=wl["components"],
                      scaler_mean=wl["mean"], scaler_scale=wl["scale"], kernel_kind=1, nu=1.5, noise=wl["noise"],
                      cov_unexplained=wl["cun"], device=device)
    dmm.likelihood_setup(prob["y_exp"], prob["y_err"], prob["lo"], prob["hi"], 1.0)
    ds = DeviceSampler([dmm], N_WALKERS, a=2.0, seed=1)
    ds.set_state(synthetic.make_walkers(N_WALKERS, seed=1))
    ds.reserve(steps + 200)
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.3:
        ds.run(100, store=False)
        dmm.sync()
    t0 = time.perf_counter()
    ds.run(steps)
    dmm.sync()
    dt = time.perf_counter() - t0
    dmm.profile(True)
    ds.run(200, store=False)
    dmm.sync()
    prof = dmm.profile_read()
    dmm.profile(False)
    nacc, iters, _ = ds.counts()
    ds.close()
    dmm.close()
    return {"workload": "C3 sizes with the shipped kernel: Matern nu=1.5 + White, fixed theta, 1024 walkers",
            "steps": steps, "ms_per_step": dt / steps * 1e3, "value": N_WALKERS * steps / dt,
            "kstar_avg_launch_us": prof["kstar"][0] / max(prof["kstar"][1], 1) * 1e3,
            "trmm_avg_launch_us": prof["trmm_vsq"][0] / max(prof["trmm_vsq"][1], 1) * 1e3,
            "acceptance_fraction_mean": float((nacc / max(iters, 1)).mean())}


def measure_scaling_model(ds, dm, single_ms_per_step, steps=300):
    """What the strong-scaling curve can be at 1024 walkers, measured on THIS GPU (no hardware curve exists: DESIGN 6):
    rank 0's share of a 2- / 4- / 8-rank run through the fused two-launch half-step (entries of the other ranks' shares
    pre-filled: the exchange costs its local part only, the xGMI hop is NOT in it), split into its terms with HIP events:
    the triangular GEMM of the share, the front kernel (likelihood of the share + exchange + accept + proposal +
    cross-kernel: latency, not throughput), the rest (RNG batch, gaps).  `floor`: the same step with the GEMM at 100 % of
    the fp64 matrix peak and the front kernel as measured -- the speed-up NO schedule of the GEMM can exceed."""
    out = {"single_gpu_ms_per_step": single_ms_per_step, "note": "emulated on one GPU; no cross-GPU hop; not a throughput claim; gemm_us / front_us are HIP-event spans on the launch "
                   "stream (each takes in part of the launch gap beside it), so their sum can exceed half of ms_per_step"}
    for world in (2, 4, 8):
        try:
            t_pre = time.perf_counter()
            while time.perf_counter() - t_pre < 0.2:
                ds.run_emulated(100, world)
                dm.sync()
            t0 = time.perf_counter()
            ds.run_emulated(steps, world)
            dm.sync()
            ms = (time.perf_counter() - t0) / steps * 1e3
            dm.profile(True)
            ds.run_emulated(100, world)
            dm.sync()
            prof = dm.profile_read()
            dm.profile(False)
            gemm_us = prof["trmm_vsq"][0] / max(prof["trmm_vsq"][1], 1) * 1e3
            front_us = prof["kstar"][0] / max(prof["kstar"][1], 1) * 1e3
            share = -(-(N_WALKERS // 2) // world)
            gemm_floor_us = N_PC * N_DESIGN ** 2 * share / (FP64_MATRIX_PEAK_TFLOPS * 1e12) * 1e6
            floor_ms = 2.0 * (gemm_floor_us + front_us) * 1e-3
            out[str(world)] = {"ms_per_step": ms, "speedup": single_ms_per_step / ms, "proposals_per_rank_and_half": share,
                               "gemm_us": gemm_us, "front_us": front_us, "half_step_us": ms * 1e3 / 2,
                               "gemm_frac_of_peak": gemm_floor_us / gemm_us,
                               "floor": {"gemm_at_peak_us": gemm_floor_us, "ms_per_step": floor_ms,
                                         "speedup_bound": single_ms_per_step / floor_ms}}
        except Exception as e:
            out[str(world)] = {"error": repr(e)}
    return out


def measure_predict(dm, n_samples=1024, reps=20, prewarm_s=0.3):
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
        # untimed pre-warm, as for metric 1 (timed_pass): workspace allocation and schedules, and >= 0.3 s of the hot
        # path itself -- the clocks of the part take tens of ms of load to settle, and 20 calls are only ~12 ms: measured
        # right after three warm-up calls the triangular GEMM of this path runs at 120-134 us per launch, in steady state
        # (the 600th launch, in-kernel stamps) at the sampler's 94 us
        t_pre = time.perf_counter()
        while True:
            for _ in range(8):
                dm.predict_full_dev(X.data_ptr(), B, float(B), cv.data_ptr(), cov.data_ptr(), stream=st.cuda_stream)
            st.synchronize()
            if time.perf_counter() - t_pre > prewarm_s:
                break
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
    from gpemu import estimators
    nbatch = estimators.fit_batch_size(N)              # what fit_gps would put through one chain at this N (58 at N = 5000)
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
    # untimed: a two-GP fit of the first 128 design points -- the first launch of every kernel of the fit loads its code
    # object (0.2-0.3 s in all), which is not the fit's time (the PCA and metric 2 are warmed the same way)
    np.random.seed(2025)
    estimators.fit_gps(prob["design"][:128], Y_pca[:128, :2], kern, alpha=1e-10, n_restarts_optimizer=1, device=device)
    np.random.seed(2026)
    t0 = time.perf_counter()
    gps = estimators.fit_gps(prob["design"], Y_pca[:, :N_PC], kern, alpha=1e-10, n_restarts_optimizer=n_restarts,
                             device=device)
    dt = time.perf_counter() - t0
    t_all, t_lib = getattr(gps[0], "fit_seconds_", (dt, None))
    return {"workload": f"C3 fit: 10 GPs x (1 + {n_restarts}) L-BFGS-B runs at N_design=1000", "seconds": dt,
            "seconds_in_library": t_lib, "seconds_host_optimiser": None if t_lib is None else t_all - t_lib,
            "lbfgsb_driver": getattr(gps[0], "fit_driver_", None),
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


def host_description():
    """What the CPU baseline ran on (core count, CPU model, BLAS vendor / threads, library versions)."""
    import numpy
    import scipy
    info = {"os_cpu_count": os.cpu_count(), "numpy": numpy.__version__, "scipy": scipy.__version__}
    try:
        info["affinity_cpus"] = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:        # cgroup v2 CPU quota of the container, if any ("max 100000" = none)
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        info["cgroup_cpu_quota"] = None if quota == "max" else float(quota) / float(period)
    except Exception:
        pass
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                info["cpu_model"] = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    try:
        from threadpoolctl import threadpool_info
        info["blas"] = [{k: lib.get(k) for k in ("internal_api", "version", "num_threads", "threading_layer")}
                        for lib in threadpool_info() if lib.get("user_api") == "blas"]
    except Exception:
        pass
    return info


def cpu_baseline(seconds_budget=24.0):
    """The reference's CPU path beside the GPU number: the oracle's reference-form per-walker
    log_posterior (kind "port"), one walker per task (ref: mcmc.py:77-85), BLAS threads = 1 per process.
      value                   spawn pool of all usable cores, truncation covariance recomputed in every call (what the
                              reference does: ref: emulation.py:214-224 returns None, so :445-448 recomputes it)
      precomputed_cov         the same pool with the truncation covariance computed once
      single_process          run A: one process, per-core rate, both variants"""
    import multiprocessing as mp
    host = host_description()
    usable = [host.get("os_cpu_count") or 1, host.get("affinity_cpus") or 10 ** 9]
    if host.get("cgroup_cpu_quota"):
        usable.append(max(1, int(host["cgroup_cpu_quota"])))
    ncores = max(1, min(usable))
    ctx = mp.get_context("spawn")

    def timed(pool, fn, nproc, budget):
        n = nproc * 2
        t0 = time.time()
        pool.map(fn, range(n), chunksize=1)
        dt = time.time() - t0
        n = int(max(n, min(20000, n * budget / max(dt, 1e-3))))
        t0 = time.time()
        pool.map(fn, range(n), chunksize=max(1, min(4, n // (4 * nproc))))
        dt = time.time() - t0
        return n / dt, n

    t_setup = time.time()
    with ctx.Pool(ncores, initializer=_cpu_init) as pool:
        pool.map(_cpu_eval, range(ncores))                      # warm-up: builds the model per worker
        t_setup = time.time() - t_setup
        rate, n = timed(pool, _cpu_eval, ncores, seconds_budget * 0.35)
        rate_pre, n_pre = timed(pool, _cpu_eval_pre, ncores, seconds_budget * 0.25)
    with ctx.Pool(1, initializer=_cpu_init) as pool:
        pool.map(_cpu_eval, range(1))
        rate1, n1 = timed(pool, _cpu_eval, 1, seconds_budget * 0.2)
        rate1_pre, n1_pre = timed(pool, _cpu_eval_pre, 1, seconds_budget * 0.2)
    return {"value": rate, "unit": "log-posterior evals/s", "cores": ncores, "kind": "port",
            "sample": f"{n} per-walker reference-form evaluations of the C3 workload over a "
                      f"{ncores}-process spawn pool, BLAS threads = 1 per process, truncation covariance recomputed in "
                      f"every call as the reference does (per-process model build {t_setup:.1f} s not timed)",
            "precomputed_cov": {"value": rate_pre, "sample": f"{n_pre} evaluations, same pool, truncation covariance "
                                                             "computed once"},
            "single_process": {"value": rate1, "precomputed_cov_value": rate1_pre, "cores": 1,
                               "sample": f"{n1} / {n1_pre} evaluations in one process, BLAS threads = 1"},
            "host": host}


_CPU = {}


def _cpu_init():
    os.environ["OMP_NUM_THREADS"] = "1"
    os.environ["OPENBLAS_NUM_THREADS"] = "1"
    try:
        from threadpoolctl import threadpool_limits
        _CPU["limit"] = threadpool_limits(1)
    except Exception:
        pass
    from gpemu import synthetic
    from oracle import gp_oracle as O
    from oracle import workloads
    model, prob, _ = workloads.fixed_theta_model(N_DESIGN, N_OBS, N_PC, seed=0)
    _CPU.update(model=model, prob=prob, X=synthetic.make_walkers(N_WALKERS, seed=1),
                cun={"g": O.cov_unexplained(model)})


def _cpu_eval(i, cun=None):
    from oracle import gp_oracle as O
    p = _CPU["prob"]
    x = _CPU["X"][i % N_WALKERS]
    return float(O.log_posterior(x, {"g": _CPU["model"]}, p["lo"], p["hi"], p["y_exp"], p["y_err"],
                                 cov_unexpl=cun)[0])


def _cpu_eval_pre(i):
    return _cpu_eval(i, _CPU["cun"])


def self_launch(args):
    """`python bench.py --gpus N` without a launcher environment: start the N ranks as fresh child processes
    (torch.distributed.run, one per GPU) BEFORE this process touches a GPU, let them print (rank 0: the JSON line)
    and return their exit code.  Nothing is re-executed in a process that has initialised HIP."""
    import socket
    import subprocess
    rehearsal = bool(os.environ.get("GPEMU_BENCH_REHEARSAL_WALKERS"))
    try:
        import torch
        ndev = torch.cuda.device_count()          # does not initialise the runtime on this image
    except Exception:
        ndev = None
    if ndev is not None and ndev < args.gpus and not rehearsal:
        print(f"bench.py: --gpus {args.gpus} but only {ndev} GPU(s) are visible", file=sys.stderr, flush=True)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the peer transport shares device memory across processes (dmabuf IPC)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fit", action="store_true", help="skip the fit-side legs (fit_c5, fit_c3)")
    ap.add_argument("--no-predict", action="store_true", help="skip metric 2 (gp_predict); for kernel profiles of the sampler alone")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the scaling_model and matern15 legs: they launch the headline's kernels on other amounts of work, "
                         "which would mix into a kernel profile's per-kernel averages")
    ap.add_argument("--transport", default="both", choices=["both", "peer", "rccl", "torch"],
                    help="N > 1: how the new log-probabilities are exchanged; 'both' times the RCCL all-gather run and "
                         "the peer-store run in one invocation and reports the faster as `value`")
    ap.add_argument("--weak", action="store_true",
                    help="N > 1: also time 1024 x N walkers (weak scaling), reported under `weak_scaling`")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="diagnostic: time ONE rank's share of an N-GPU step on this GPU (world-1 RCCL gather); "
                         "the printed value is NOT a throughput claim")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.emulate_world:
        sys.exit(self_launch(args))

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE={world}")
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

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dm.sync()

    def all_ok(ok):
        """True on every rank only if every rank succeeded (the ranks take the same branch afterwards)."""
        if world == 1:
            return ok
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))

    def timed_pass(ds, n_walkers, transport):
        """Pre-warm, W warm-up steps, then EXACTLY K timed steps between barrier + synchronize; max over ranks."""
        def run(steps, store=True):
            if args.emulate_world:
                ds.run_sharded(steps, store=store, force=True, emulate_world=args.emulate_world)
            elif world > 1:
                ds.run_sharded(steps, store=store, transport=transport)
            else:
                ds.run(steps, store=store)
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
        return {"ms_per_step": dt / args.steps * 1e3, "value": n_walkers * args.steps / dt, "seconds": dt,
                "transport_taken": ds.last_transport or "single", "run": run}

    def measure(n_walkers, seed):
        """One ensemble of `n_walkers`, every requested transport timed on it.  Returns (sampler, results, best)."""
        X0 = synthetic.make_walkers(n_walkers, seed=seed)
        ds = DeviceSampler([dm], n_walkers, a=2.0, seed=1)
        ds.set_state(X0)
        if world == 1 or args.emulate_world:
            wanted = ["single"]
        else:
            wanted = ["rccl", "peer"] if args.transport == "both" else [args.transport]
        ds.reserve((args.warmup + args.steps) * len(wanted) + args.steps)   # no hipMalloc inside a timed region
        results = {}
        for t in wanted:
            err = None
            try:
                res = timed_pass(ds, n_walkers, None if t == "single" else t)
            except Exception as e:              # e.g. a lost peer exchange (GPEMU_ERR_STATE on every rank)
                err, res = repr(e), None
            if all_ok(err is None):
                results[t] = res
            else:
                results[t] = {"error": err or "another rank failed"}
                torch.cuda.synchronize()
                ds.set_state(X0)                # the ensemble of a failed pass is not a valid state
        ok = {t: r for t, r in results.items() if "error" not in r}
        best = min(ok, key=lambda t: ok[t]["ms_per_step"]) if ok else None
        return ds, results, best

    ds, results, best = measure(N_WALKERS, seed=1)
    if best is None:
        if rank == 0:
            print(json.dumps({"metric": "log-posterior evals/sec", "value": None, "n_gpus": world,
                              "error": {t: r.get("error") for t, r in results.items()}}), flush=True)
        raise SystemExit(1)
    headline = results[best]
    run = headline["run"]

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
        # FLOP_eval = k N^2 (L^-1 k_*) + k N (3 d + 2) (kernel row + mean) + 2 k N (|v|^2) + ~k^3/3 + 4 k^2
        d_par = 6
        flop_eval = (N_PC * N_DESIGN ** 2 + N_PC * N_DESIGN * (3 * d_par + 2) + 2 * N_PC * N_DESIGN
                     + N_PC ** 3 / 3 + 4 * N_PC ** 2)
        step_tf = headline["value"] * flop_eval / 1e12 / (1 if not args.emulate_world else args.emulate_world)
        roofline.update(flop_per_eval=flop_eval, step_achieved=step_tf, step_frac=step_tf / (FP64_MATRIX_PEAK_TFLOPS * world))
    nacc, iters, _ = ds.counts()
    tinfo = dict(ds.transport_info)

    weak = None
    if args.weak and world > 1:
        # weak scaling, clearly NOT the headline: 1024 walkers per GPU (per-rank work as on one GPU)
        try:
            dsw, wres, wbest = measure(N_WALKERS * world, seed=5)
            weak = {"label": "weak scaling: 1024 walkers per GPU; NOT the headline value",
                    "n_walkers": N_WALKERS * world, "transport": wbest,
                    "value": wres[wbest]["value"] if wbest else None,
                    "ms_per_step": wres[wbest]["ms_per_step"] if wbest else None,
                    "transports": {t: {k: v for k, v in r.items() if k != "run"} for t, r in wres.items()}}
            dsw.close()
        except Exception as e:
            weak = {"error": repr(e)}

    steady = matern = scaling_model = None
    if world == 1 and not args.emulate_world:
        try:
            ds.reserve(8000)
            steady = measure_steady(ds, dm, N_WALKERS)
        except Exception as e:
            steady = {"error": repr(e)}
        try:
            scaling_model = None if args.no_extra else measure_scaling_model(ds, dm, headline["ms_per_step"])
        except Exception as e:
            scaling_model = {"error": repr(e)}
        if not args.no_fit and not args.no_extra:
            try:
                matern = measure_matern15(dev_index)
            except Exception as e:
                matern = {"error": repr(e)}

    predict = None
    if rank == 0 and not args.no_predict:
        try:
            predict = measure_predict(dm)
        except Exception as e:
            predict = {"value": None, "error": repr(e)}

    fit_c5 = fit_c3 = shipped = None
    if rank == 0 and world == 1 and not args.no_fit and not args.emulate_world:
        try:
            fit_c5 = measure_fit_c5(dev_index)
            fit_c3 = measure_fit_c3(dev_index)
        except Exception as e:
            fit_c5 = fit_c5 or {"error": repr(e)}
            fit_c3 = fit_c3 or {"error": repr(e)}
        try:
            shipped = measure_shipped_shape(dev_index)
            shipped["with_observable_blocks"] = measure_shipped_shape(dev_index, steps=2000, observable_blocks=True)
        except Exception as e:
            shipped = shipped or {"error": repr(e)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline()
        except Exception as e:  # the baseline is a reported figure; never lose the GPU line over it
            cpu = {"value": None, "error": repr(e)}

    if rank == 0:
        rehearsal = bool(os.environ.get("GPEMU_BENCH_REHEARSAL_WALKERS"))
        workload = (f"C3: N_design={N_DESIGN} x N_obs={N_OBS}, {N_PC} PCs, d=6, {N_WALKERS}-walker "
                    "stretch-move MCMC, RBF+White fixed theta")
        if rehearsal:
            workload = "REHEARSAL (not the headline ensemble): " + workload
        ranks_seen = {"torch_distributed": world}
        if "rccl_ranks_seen" in tinfo:
            ranks_seen["rccl_communicator"] = tinfo["rccl_ranks_seen"]
        if "peer_ranks_seen" in tinfo:
            ranks_seen["peer_selftest"] = tinfo["peer_ranks_seen"]
        out = {"metric": "log-posterior evals/sec", "value": headline["value"], "unit": "evals/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": headline["ms_per_step"], "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "emulate_world": args.emulate_world or None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": workload,
                          "n_walkers": N_WALKERS, "evals_per_step": N_WALKERS,
                          "parallelism": f"walkers sharded over {world} GPU(s)"},
               "transport": headline["transport_taken"],
               "transports": {t: {k: v for k, v in r.items() if k != "run"} for t, r in results.items()},
               "ranks_seen": ranks_seen,
               "peer_selftest_per_rank": tinfo.get("peer_selftest_per_rank"),
               "fallback_vote": {"peer_to_collective": bool(tinfo.get("fallback_from_peer", False)),
                                 "rccl_to_torch": tinfo.get("rccl_fallback_reason")},
               "acceptance_fraction_mean": float((nacc / max(iters, 1)).mean()),
               "roofline": roofline, "cpu_baseline": cpu, "gp_predict": predict, "fit_c5": fit_c5, "fit_c3": fit_c3,
               "shipped_shape": shipped, "weak_scaling": weak,
               "steady": steady, "matern15": matern, "scaling_model": scaling_model}
        if rehearsal:
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
