"""Stretch-move ensemble samplers.

``DeviceSampler``    the ensemble lives on one MI355X (libgpemu ``gpemu_sampler_*``); with
                     ``torch.distributed`` initialised the proposing half is sharded over the ranks
                     and the new log-probabilities are all-gathered (RCCL over xGMI) per half-step.
``HostEnsemble``     the same move for an arbitrary Python ``log_prob_fn`` (the emcee calling
                     convention the reference uses, ref: mcmc.py:83-85), vectorised over the
                     proposing half; also shardable over ranks (any torch.distributed backend).
``EnsembleSampler``  emcee-compatible facade (the subset the reference touches: ref: mcmc.py:83-116,
                     187-204, plot_mcmc.py) that picks the device path when ``log_prob_fn`` is bound
                     to device models.
"""
from __future__ import annotations

import ctypes as C
import logging
import os

import numpy as np

from . import _lib
from ._lib import as_f64, check, ptr


def shard_bounds(n, world, rank):
    """Contiguous block of ``ceil(n/world)`` proposals per rank: (lo, hi, per)."""
    per = (n + world - 1) // world
    lo = min(rank * per, n)
    hi = min(lo + per, n)
    return lo, hi, per


class AutocorrError(Exception):
    """Raised when the chain is too short for a reliable estimate (emcee.autocorr.AutocorrError)."""

    def __init__(self, tau, *args, **kwargs):
        self.tau = tau
        super().__init__(*args, **kwargs)


def _next_pow_two(n):
    i = 1
    while i < n:
        i = i << 1
    return i


def _rfft(a, n):
    """Real-input FFT along axis 0, on all host threads where scipy's pocketfft is there."""
    try:
        import scipy.fft
        return scipy.fft.rfft(a, n=n, axis=0, workers=-1)
    except Exception:
        return np.fft.rfft(a, n=n, axis=0)


def _irfft(a, n):
    try:
        import scipy.fft
        return scipy.fft.irfft(a, n=n, axis=0, workers=-1)
    except Exception:
        return np.fft.irfft(a, n=n, axis=0)


def function_1d(x):
    """Normalised autocorrelation function of a 1-D series via FFT (emcee.autocorr.function_1d)."""
    x = np.atleast_1d(x)
    n = _next_pow_two(len(x))
    f = np.fft.fft(x - np.mean(x), n=2 * n)
    acf = np.fft.ifft(f * np.conjugate(f))[: len(x)].real
    acf /= acf[0]
    return acf


def integrated_time(x, c=5, tol=50, quiet=False):
    """Integrated autocorrelation time per dimension with Sokal's window
    (emcee.autocorr.integrated_time; x has shape (steps, walkers, ndim))."""
    x = np.atleast_1d(x)
    if x.ndim == 1:
        x = x[:, np.newaxis, np.newaxis]
    if x.ndim == 2:
        x = x[:, :, np.newaxis]
    n_t, n_w, n_d = x.shape
    tau_est = np.empty(n_d)
    windows = np.empty(n_d, dtype=int)
    n2 = 2 * _next_pow_two(n_t)
    chunk = max(1, min(n_w, (64 << 20) // (16 * n2)))          # walkers per batched transform (~64 MB of complex128)
    for d in range(n_d):
        f = np.zeros(n_t)
        for k0 in range(0, n_w, chunk):
            # function_1d for a block of walkers in one batched real-input FFT over all host threads (emcee transforms
            # walker by walker with a complex FFT: the same numbers to rounding), added walker by walker in its order
            xc = x[:, k0:k0 + chunk, d]
            ft = _rfft(xc - np.mean(xc, axis=0), n2)
            acf = _irfft(ft * np.conjugate(ft), n2)[:n_t]
            acf /= acf[0]
            for k in range(acf.shape[1]):
                f += acf[:, k]
        f /= n_w
        taus = 2.0 * np.cumsum(f) - 1.0
        m = np.arange(len(taus)) < c * taus
        windows[d] = np.argmin(m) if np.any(m) else len(taus) - 1
        tau_est[d] = taus[windows[d]]
    flag = tol * tau_est > n_t
    if np.any(flag):
        msg = ("The chain is shorter than {0} times the integrated autocorrelation time for {1} "
               "parameter(s). Use this estimate with caution and run a longer chain!\n"
               ).format(tol, np.sum(flag))
        msg += "N/{0} = {1:.0f};\ntau: {2}".format(tol, n_t / tol, tau_est)
        if not quiet:
            raise AutocorrError(tau_est, msg)
        logging.getLogger(__name__).warning(msg)            # emcee logs the message when quiet (autocorr.py)
    return tau_est


# ------------------------------------------------------------------------------------------------
class DeviceSampler:
    """Ensemble resident on the device; log-posterior = sum over the given DeviceModels."""

    def __init__(self, models, n_walkers, a=2.0, seed=0, seeds=None):
        """``seeds`` (a sequence): that many INDEPENDENT chains of ``n_walkers`` walkers each, stacked in one
        sampler and evaluated in the same launches (the closure tests of ref: steer_analysis.py:168-183); chain c is
        the chain ``DeviceSampler(models, n_walkers, a, seed=seeds[c])`` produces on data vector c of the models'
        ``likelihood_setup(y_exp (C, F), ...)``.  State and chain arrays then hold ``C * n_walkers`` walkers,
        chain after chain."""
        _lib.require_device()
        self.models = list(models)
        arr = (C.c_void_p * len(self.models))(*[m.handle for m in self.models])
        h = C.c_void_p()
        if seeds is None:
            check(_lib.lib().gpemu_sampler_create(C.byref(h), arr, len(self.models), int(n_walkers), float(a),
                                                  C.c_uint64(int(seed) & (2 ** 64 - 1))))
            self.n_chains = 1
        else:
            sd = np.array([int(v) & (2 ** 64 - 1) for v in seeds], dtype=np.uint64)
            check(_lib.lib().gpemu_sampler_create_chains(C.byref(h), arr, len(self.models), int(n_walkers), float(a),
                                                         ptr(sd), int(sd.size)))
            self.n_chains = int(sd.size)
        self._h = h
        self.walkers_per_chain = int(n_walkers)
        self.W, self.d = int(n_walkers) * self.n_chains, self.models[0].d
        self.ns = ((self.W + 1) // 2, self.W // 2)
        self.device = self.models[0].device
        self.last_transport = None       # what the last run_sharded call really took: "peer" / "rccl" / "torch" / "single" / "replicated"
        self.transport_info = {}         # self-test results, communicator sizes, fall-back reasons

    def close(self):
        for h in self.__dict__.pop("_comms", {}).values():
            if h is not None:
                _lib.lib().gpemu_comm_destroy(h)
        if getattr(self, "_h", None):
            _lib.lib().gpemu_sampler_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_state(self, X0, logp0=None):
        X0 = as_f64(X0, (self.W, self.d))
        lp = None if logp0 is None else as_f64(logp0, (self.W,))
        check(_lib.lib().gpemu_sampler_set_state(self._h, ptr(X0), ptr(lp)))

    def get_state(self):
        X = np.empty((self.W, self.d))
        lp = np.empty(self.W)
        check(_lib.lib().gpemu_sampler_get_state(self._h, ptr(X), ptr(lp)))
        return X, lp

    def reset(self):
        check(_lib.lib().gpemu_sampler_reset(self._h))

    def reserve(self, steps):
        """Grow the device chain buffer for ``steps`` more stored steps now (otherwise it grows on demand)."""
        check(_lib.lib().gpemu_sampler_reserve_chain(self._h, int(steps)))

    def run(self, steps, store=True):
        rc = _lib.lib().gpemu_sampler_run(self._h, int(steps), int(bool(store)))
        if rc == 1:
            raise ValueError("Probability function returned NaN")
        check(rc)

    def step_host_rng(self, inds, zz, rint, logu, store=True):
        inds = np.ascontiguousarray(inds, dtype=np.int32)
        zz = as_f64(np.concatenate(zz), (self.W,))
        logu = as_f64(np.concatenate(logu), (self.W,))
        rint = np.ascontiguousarray(np.concatenate(rint), dtype=np.int64)
        rc = _lib.lib().gpemu_sampler_step_host_rng(self._h, ptr(inds), ptr(zz), ptr(rint), ptr(logu),
                                                    int(bool(store)))
        if rc == 1:
            raise ValueError("Probability function returned NaN")
        check(rc)

    def counts(self):
        nacc = np.zeros(self.W, dtype=np.int64)
        it, cl = C.c_int64(), C.c_int64()
        check(_lib.lib().gpemu_sampler_get_counts(self._h, ptr(nacc), C.byref(it), C.byref(cl)))
        return nacc, int(it.value), int(cl.value)

    def get_chain(self, first=0, n=None):
        _, _, cl = self.counts()
        n = cl - first if n is None else n
        chain = np.empty((n, self.W, self.d))
        lp = np.empty((n, self.W))
        check(_lib.lib().gpemu_sampler_get_chain(self._h, int(first), int(n), ptr(chain), ptr(lp)))
        return chain, lp

    def acf_block(self, lag0, n_lags, first=0, n=None, w0=0, nw=None):
        """Walker-averaged normalised autocorrelation function, lags [lag0, lag0 + n_lags), of the chain stored on
        the device: (n_lags, d).  ``lag0`` a multiple of 16, the first block of an estimate at 0."""
        _, _, cl = self.counts()
        n = cl - first if n is None else n
        nw = self.W - w0 if nw is None else nw
        f = np.empty((int(n_lags), self.d))
        check(_lib.lib().gpemu_sampler_acf(self._h, int(first), int(n), int(w0), int(nw), int(lag0), int(n_lags), ptr(f)))
        return f

    def integrated_time(self, first=0, n=None, w0=0, nw=None, c=5, tol=50, quiet=False, block=256):
        """emcee.autocorr.integrated_time of the stored chain (rows [first, first + n), walkers [w0, w0 + nw)) WITHOUT
        bringing it to the host: the lag products are formed on the device, ``block`` lags at a time, until Sokal's
        window (the first M with M >= c tau(M)) has closed for every parameter -- a few hundred lags instead of FFTs over
        the whole chain.  Same estimate as ``gpemu.sampler.integrated_time`` (rounding apart)."""
        _, _, cl = self.counts()
        n_t = int(cl - first if n is None else n)
        d = self.d
        tau_est = np.full(d, np.nan)
        windows = np.full(d, -1, dtype=int)
        run = np.zeros(d)                    # cumulative sum of f so far
        lag0 = 0
        while lag0 < n_t and np.any(windows < 0):
            nl = int(min(block, n_t - lag0))
            f = self.acf_block(lag0, nl, first=first, n=n_t, w0=w0, nw=nw)
            cs = run[None, :] + np.cumsum(f, axis=0)
            taus = 2.0 * cs - 1.0
            if lag0 == 0:
                tau_first = taus[0].copy()
            idx = lag0 + np.arange(nl)
            for dd in range(d):
                if windows[dd] >= 0:
                    continue
                closed = ~(idx < c * taus[:, dd])           # NaN compares False: the window "closes" at once, as in emcee
                if np.any(closed):
                    m = int(np.argmax(closed))
                    windows[dd] = lag0 + m
                    tau_est[dd] = taus[m, dd]
                elif lag0 + nl >= n_t:
                    # never closed (idx < c tau at every lag): emcee's auto_window takes np.argmin of an all-True mask,
                    # i.e. window 0 and tau = taus[0] (which then fails the tol test: the short-chain AutocorrError).
                    # (A NaN series is the other case: its mask has no True entry, emcee returns len - 1 there and a
                    # NaN tau; here the window "closes" at lag 0 with the same NaN.)
                    windows[dd] = 0
                    tau_est[dd] = tau_first[dd]
            run = cs[-1]
            lag0 += nl
            block = min(block * 2, 4096)
        flag = tol * tau_est > n_t
        if np.any(flag):
            msg = ("The chain is shorter than {0} times the integrated autocorrelation time for {1} "
                   "parameter(s). Use this estimate with caution and run a longer chain!\n"
                   ).format(tol, np.sum(flag))
            msg += "N/{0} = {1:.0f};\ntau: {2}".format(tol, n_t / tol, tau_est)
            if not quiet:
                raise AutocorrError(tau_est, msg)
            logging.getLogger(__name__).warning(msg)        # emcee logs the message when quiet (autocorr.py)
        return tau_est

    # -- multi-GPU: one process per GPU, the ensemble replicated, proposals sharded -------------
    def _rccl_comm_agreed(self, group):
        """RCCL communicator owned by the library (one per sampler and process group), bootstrapped through
        torch.distributed -- or None on EVERY rank if any rank cannot provide it, so that all ranks take the same
        transport.  Every rank goes through the same sequence of collectives whatever fails locally:
          1. rank 0 makes the 128-byte id; it is broadcast together with a status byte (zeroed id + 0 on failure);
          2. every rank probes that it can load librccl; the ranks vote (all-reduce MIN) BEFORE anyone enters
             ncclCommInitRank, which would otherwise wait for the ranks that could not follow;
          3. the ranks join, run one all-gather of their indices through the new communicator, and vote again."""
        import warnings
        import torch
        import torch.distributed as dist
        key = id(group) if group is not None else 0
        comms = self.__dict__.setdefault("_comms", {})
        if key in comms:
            return comms[key]
        L = _lib.lib()
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        dev = torch.device("cuda", self.device)
        bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        path = bundled.encode() if os.path.exists(bundled) else None   # the copy torch already loaded
        err = None

        def vote(ok):
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            return int(t.item()) == 1

        def give_up(reason):
            warnings.warn(f"library-owned RCCL communicator unavailable ({reason}); using torch.distributed's all-gather")
            self.transport_info["rccl_fallback_reason"] = str(reason)
            comms[key] = None
            return None

        # 1. the id, always broadcast
        ident = (C.c_char * 128)()
        status = 1
        if rank == 0:
            try:
                check(L.gpemu_comm_unique_id(path, C.cast(ident, C.c_void_p)))
            except Exception as e:
                err, status = e, 0
                ident = (C.c_char * 128)()
        t = torch.tensor(list(ident.raw) + [status], dtype=torch.uint8, device=dev)
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast(t, src=src, group=group)
        raw = bytes(t.cpu().tolist())
        if raw[128] == 0:
            return give_up(repr(err) if err is not None else "rank 0 could not create the id")
        # 2. can this rank load the library?  (an id of its own, thrown away, exercises dlopen + the symbols)
        try:
            scratch = (C.c_char * 128)()
            check(L.gpemu_comm_unique_id(path, C.cast(scratch, C.c_void_p)))
            loadable = True
        except Exception as e:
            err, loadable = e, False
        if not vote(loadable):
            return give_up(repr(err) if err is not None else "another rank cannot load librccl")
        # 3. join, self-check, vote
        comm = None
        try:
            buf = C.create_string_buffer(raw[:128], 128)
            h = C.c_void_p()
            check(L.gpemu_comm_create(C.byref(h), int(self.device), int(rank), int(world), C.cast(buf, C.c_void_p), path))
            comm = h
            src_t = torch.full((1,), float(rank), dtype=torch.float64, device=dev)
            dst_t = torch.full((world,), -1.0, dtype=torch.float64, device=dev)
            torch.cuda.synchronize(dev)
            check(L.gpemu_comm_all_gather(comm, C.c_void_p(src_t.data_ptr()), C.c_void_p(dst_t.data_ptr()), 1, None))
            torch.cuda.synchronize(dev)
            if dst_t.cpu().tolist() != [float(r) for r in range(world)]:
                raise RuntimeError(f"RCCL all-gather self-check returned {dst_t.cpu().tolist()}")
            good = True
        except Exception as e:      # ncclCommInitRank failure, wrong data ...
            err, good = e, False
        if vote(good):
            comms[key] = comm
            r_seen, w_seen = C.c_int(-1), C.c_int(-1)
            L.gpemu_comm_dims(comm, C.byref(r_seen), C.byref(w_seen))
            self.transport_info["rccl_ranks_seen"] = int(w_seen.value)      # ncclCommCount of the new communicator
            return comm
        if comm is not None:
            L.gpemu_comm_destroy(comm)
        return give_up(repr(err) if err is not None else "another rank failed the self-check")

    def _peer_ready(self, group):
        """Exchange the ranks' gather-buffer IPC handles (once per sampler and group) so that every rank can
        store its log-probabilities straight into the others' memory.  True on every rank, or False on every
        rank (the ranks vote), e.g. when the model is outside the fused run's limits or IPC is unavailable."""
        import torch
        import torch.distributed as dist
        key = id(group) if group is not None else 0
        cache = self.__dict__.setdefault("_peer_ok", {})
        if key in cache:
            return cache[key]
        L = _lib.lib()
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        where = torch.device("cuda", self.device) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        # ranks sharing this GPU (one-GPU rehearsals): the library then checks that all their launches fit on it
        # together; one rank per GPU -- the production layout -- needs no such rule
        import hashlib
        import socket
        bus = C.create_string_buffer(64)
        check(L.gpemu_device_bus_id(int(self.device), bus, 64))
        key16 = hashlib.sha256(socket.gethostname().encode() + b"/" + bus.value).digest()[:16]
        mykey = torch.tensor(list(key16), dtype=torch.uint8, device=where)
        keys = torch.zeros(16 * world, dtype=torch.uint8, device=where)
        dist.all_gather_into_tensor(keys, mykey, group=group)
        keys = bytes(keys.cpu().tolist())
        share = sum(1 for r in range(world) if keys[16 * r:16 * r + 16] == key16)
        check(L.gpemu_sampler_peer_share(self._h, int(share)))
        self.transport_info["ranks_on_this_device"] = int(share)
        mine = (C.c_char * 64)()
        ok = 1
        stage = "ok"
        if L.gpemu_sampler_peer_export(self._h, C.cast(mine, C.c_void_p)) != 0:
            ok, stage = 0, "export: " + _lib.last_error()
        t = torch.tensor(list(mine.raw), dtype=torch.uint8, device=where)
        everyone = torch.zeros(64 * world, dtype=torch.uint8, device=where)
        dist.all_gather_into_tensor(everyone, t, group=group)
        if ok:
            raw = bytes(everyone.cpu().tolist())
            buf = C.create_string_buffer(raw, 64 * world)
            if L.gpemu_sampler_peer_import(self._h, int(world), int(rank), C.cast(buf, C.c_void_p)) != 0:
                ok, stage = 0, "import: " + _lib.last_error()
        # the data path itself, before a chain depends on it: every rank stores a token into every rank's buffer and
        # waits (bounded) for all tokens in its own -- a mapping that opens but does not carry stores (or carries them
        # too late) makes the ranks fall back to the collective transports together instead of losing an exchange
        dist.barrier(group=group)
        if ok and L.gpemu_sampler_peer_selftest(self._h) != 0:
            ok, stage = 0, "selftest: " + _lib.last_error()
        # every rank's result, for the record (bench.py prints it), and the vote
        mine_ok = torch.tensor([ok], dtype=torch.int32, device=where)
        all_ok = torch.zeros(world, dtype=torch.int32, device=where)
        dist.all_gather_into_tensor(all_ok, mine_ok, group=group)
        per_rank = [int(v) for v in all_ok.cpu().tolist()]
        cache[key] = all(v == 1 for v in per_rank)
        self.transport_info.update(peer_selftest_per_rank=per_rank, peer_ranks_seen=sum(per_rank),
                                   peer_local_stage=stage, peer_vote=cache[key])
        return cache[key]

    def worth_sharding(self):
        """Is a half-step of this sampler large enough for walker sharding to pay?  Arithmetic of one half-step on one
        GPU, sum over the groups of k Npad^2 x proposals (the triangular GEMM), against GPEMU_SHARD_MIN_GFLOP
        (default 1.0).  Measured: C3 (5.4 GFLOP per half-step) 116 us on one GPU against 36 us for a rank's share of 8;
        the shipped three-group shape (0.27 GFLOP) 42 us on one GPU against 59 us through the fused half-step at ANY world
        size (profiles/r04_shipped_shape_step.txt) -- below about 1 GFLOP the step is launch latency, which sharding adds
        to."""
        gflop = sum(m.k * (-(-m.N // 128) * 128) ** 2 for m in self.models) * self.ns[0] / self.n_chains / 1e9
        return gflop >= float(os.environ.get("GPEMU_SHARD_MIN_GFLOP", "1.0"))

    def _run_peer_block(self, steps, store, vote):
        """One block of steps through the fused peer-store run with a way back: the chain state is snapshot on the
        device first; if the run fails with a LOST EXCHANGE on any rank (``vote(ok)``: every rank's outcome, all-reduced
        -- the ranks must take the same branch), every rank restores the snapshot and False is returned: the caller
        reruns the block over another transport.  True: the block is done.  A NaN log-probability is the model's, not
        the transport's: raised as emcee raises it."""
        L = _lib.lib()
        check(L.gpemu_sampler_snapshot(self._h))
        rc = L.gpemu_sampler_run_peer(self._h, int(steps), int(bool(store)))
        lost = rc == -4 and "timed out" in _lib.last_error()            # GPEMU_ERR_STATE from the bounded waits
        if rc != 0 and rc != 1 and not lost:
            vote(False)                                                  # (the peers must not wait for this rank's vote)
            check(rc)
        if vote(rc in (0, 1)):
            if rc == 1:
                raise ValueError("Probability function returned NaN")
            return True
        check(L.gpemu_sampler_restore(self._h))
        return False

    def run_emulated(self, steps, world, store=False):
        """Timing aid (bench.py's ``scaling_model``): rank 0's share of a ``world``-rank sharded run on this GPU through
        the fused two-launch half-step, without a process group or communicator (the entries of the other ranks' shares
        are pre-filled, so the exchange costs its local part only).  NOT a valid chain."""
        self.last_transport = "emulated"
        check(_lib.lib().gpemu_sampler_run_sharded(self._h, None, int(steps), int(bool(store)), int(world)))

    def run_sharded(self, steps, store=True, group=None, force=False, emulate_world=None, transport=None):
        """Same chain as ``run`` (every rank draws identical randomness); rank r evaluates its
        block of each half's proposals and the log-probabilities are exchanged.

        transport "peer" (default where it applies: up to 8 emulation groups of up to 64 PCs, launches that fit
        on the chip at once): a front launch + one triangular GEMM per group per half-step, every new
        log-probability stored straight into all ranks' memory (xGMI peer stores, no collective):
        ``gpemu_sampler_run_peer``.  transport "rccl" (backend "nccl"): the loop runs
        inside the library on its own RCCL communicator with one all-gather per half-step
        (``gpemu_sampler_run_sharded``).  transport "torch": per-phase calls with torch.distributed's all-gather
        (staged through the host on non-nccl backends).  GPEMU_SHARDED_TRANSPORT overrides the default.
        """
        import torch
        import torch.distributed as dist
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        if world == 1 and not force:
            self.last_transport = "single"
            return self.run(steps, store)
        L = _lib.lib()
        dev = torch.device("cuda", self.device)
        on_device = dist.get_backend(group) == "nccl"
        if transport is None and "GPEMU_SHARDED_TRANSPORT" not in os.environ and not emulate_world and not self.worth_sharding():
            # A small model: a half-step on ONE GPU is a handful of ~10 us launches (the reference's shipped shape:
            # 84 us per step), and the two-launch fused half-step every transport of a sharded run pays is slower than
            # that (119 us at the same shape, before any hop): sharding would be a regression.  Every rank holds the whole
            # ensemble and draws the same randomness, so each simply runs the chain itself -- the same chain, no exchange.
            self.last_transport = "replicated"
            self.transport_info["requested"] = "replicated (small model: below GPEMU_SHARD_MIN_GFLOP per half-step)"
            return self.run(steps, store)
        transport = transport or os.environ.get("GPEMU_SHARDED_TRANSPORT", "peer")
        self.transport_info["requested"] = transport
        if transport == "peer" and not emulate_world:
            if world > 1 and self._peer_ready(group):
                # the ranks enter every fused run together: all launches of the previous run (their gather-slot
                # hand-backs included) have completed everywhere before anyone stores into a peer's buffer again
                dist.barrier(group=group)
                self.last_transport = "peer"

                def vote(ok):
                    where = dev if on_device else torch.device("cpu")
                    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=where)
                    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
                    return int(t.item()) == 1
                if self._run_peer_block(int(steps), bool(store), vote):
                    return None
                # a peer's values did not arrive in time on some rank: every rank is back at the state before the block
                # (restored from the device snapshot) and the block is rerun over the collective transport; the peer
                # stores are not trusted again for this sampler
                self.__dict__.setdefault("_peer_ok", {})[id(group) if group is not None else 0] = False
                self.transport_info["peer_blocks_recovered"] = self.transport_info.get("peer_blocks_recovered", 0) + 1
                logging.getLogger(__name__).warning(
                    "sharded run: a peer exchange was lost; the block of %d steps is rerun over the collective transport "
                    "from the state before it (same random stream: the chain is that of an unbroken run)", int(steps))
            else:
                self.transport_info["fallback_from_peer"] = True
            transport = "rccl"
        elif transport == "peer":
            transport = "rccl"          # the emulation switch lives in gpemu_sampler_run_sharded
        if on_device and transport == "rccl":
            comm = self._rccl_comm_agreed(group)
            if comm is None:
                transport = "torch"
        if on_device and transport == "rccl":
            self.last_transport = "rccl"
            rc = L.gpemu_sampler_run_sharded(self._h, comm, int(steps), int(bool(store)), int(emulate_world or 0))
            if rc == 1:
                raise ValueError("Probability function returned NaN")
            check(rc)
            return None
        self.last_transport = "torch"
        # a dedicated (non-null) torch stream carries the library's launches AND the collectives, so
        # they are ordered by the stream; torch's default stream has handle 0, which the C ABI reads
        # as "use the handle's own stream"
        if getattr(self, "_tstream", None) is None:
            self._tstream = torch.cuda.Stream(device=dev)
        stream = self._tstream
        stream.wait_stream(torch.cuda.current_stream(dev))
        check(L.gpemu_sampler_set_stream(self._h, C.c_void_p(stream.cuda_stream)))
        if store:
            check(L.gpemu_sampler_reserve_chain(self._h, int(steps)))
        if emulate_world:   # measurement aid: do one rank's share of an `emulate_world`-GPU run (results are NOT a valid chain)
            bounds = [shard_bounds(self.ns[h], int(emulate_world), 0) for h in (0, 1)]
            bounds = [(lo, hi, self.ns[h]) for h, (lo, hi, _p) in enumerate(bounds)]
        else:
            bounds = [shard_bounds(self.ns[h], world, rank) for h in (0, 1)]
        with torch.cuda.stream(stream):
            # emulation: proposals nobody evaluates carry -inf, i.e. are rejected
            mine = [torch.full((b[2],), float("-inf") if emulate_world else 0.0, dtype=torch.float64, device=dev)
                    for b in bounds]
            full = [torch.zeros(b[2] * world, dtype=torch.float64, device=dev) for b in bounds]
        try:
          with torch.cuda.stream(stream):
            for _ in range(int(steps)):
                check(L.gpemu_sampler_begin_step(self._h))
                for h in (0, 1):
                    lo, hi, _per = bounds[h]
                    check(L.gpemu_sampler_half_propose_eval(self._h, h, lo, hi, C.c_void_p(mine[h].data_ptr())))
                    if on_device:
                        dist.all_gather_into_tensor(full[h], mine[h], group=group)
                    else:
                        host_mine = mine[h].cpu()
                        host_full = torch.empty(full[h].shape, dtype=torch.float64)
                        dist.all_gather_into_tensor(host_full, host_mine, group=group)
                        full[h].copy_(host_full)
                    check(L.gpemu_sampler_half_accept(self._h, h, C.c_void_p(full[h].data_ptr()), int(bool(store))))
                check(L.gpemu_sampler_end_step(self._h, int(bool(store))))
            rc = L.gpemu_sampler_check(self._h)
            if rc == 1:
                raise ValueError("Probability function returned NaN")
            check(rc)
        finally:
            stream.synchronize()
            L.gpemu_sampler_set_stream(self._h, None)


# ------------------------------------------------------------------------------------------------
class HostEnsemble:
    """Stretch move for an arbitrary vectorised ``log_prob_fn(X (n,d)) -> (n,)`` on the host.

    Randomness: numpy ``RandomState`` consumed in emcee's order (choice, shuffle, rand, randint,
    rand...).  With torch.distributed initialised (``sharded=True``) rank r evaluates its block of
    the proposals and the values are all-gathered; every rank keeps the whole ensemble.
    """

    def __init__(self, n_walkers, ndim, log_prob_fn, a=2.0, seed=None, sharded=False, group=None):
        if n_walkers < 2 * ndim:
            raise RuntimeError("It is unadvisable to use a red-blue move with fewer walkers than twice "
                               "the number of dimensions.")
        self.W, self.d, self.fn, self.a = n_walkers, ndim, log_prob_fn, a
        self.random = np.random.RandomState(seed)
        self.sharded, self.group = sharded, group
        self.X = None
        self.lp = None
        self.reset()

    def reset(self):
        self.chain, self.lps = [], []
        self.naccepted = np.zeros(self.W, dtype=np.int64)
        self.iterations = 0

    def _eval(self, q):
        if not self.sharded:
            return np.asarray(self.fn(q), dtype=np.float64).reshape(-1)
        import torch
        import torch.distributed as dist
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        lo, hi, per = shard_bounds(q.shape[0], world, rank)
        mine = torch.zeros(per, dtype=torch.float64)
        if hi > lo:
            mine[: hi - lo] = torch.from_numpy(np.asarray(self.fn(q[lo:hi]), dtype=np.float64).reshape(-1))
        full = torch.zeros(per * world, dtype=torch.float64)
        dist.all_gather_into_tensor(full, mine, group=self.group)
        return full.numpy()[: q.shape[0]].copy()

    def set_state(self, X0, logp0=None):
        self.X = np.array(X0, dtype=np.float64)
        if self.X.shape != (self.W, self.d):
            raise ValueError("incompatible input dimensions")
        self.lp = self._eval(self.X) if logp0 is None else np.array(logp0, dtype=np.float64)
        if np.any(np.isnan(self.lp)):
            raise ValueError("The initial log_prob was NaN")

    def step(self, store=True):
        rs, W, a = self.random, self.W, self.a
        rs.choice(1, p=[1.0])
        inds = np.arange(W) % 2
        rs.shuffle(inds)
        for split in range(2):
            S1 = inds == split
            s, c = self.X[S1], self.X[~S1]
            ns, nc = s.shape[0], c.shape[0]
            zz = ((a - 1.0) * rs.rand(ns) + 1) ** 2.0 / a
            factors = (self.d - 1.0) * np.log(zz)
            rint = rs.randint(nc, size=(ns,))
            q = c[rint] - (c[rint] - s) * zz[:, None]
            new_lp = self._eval(q)
            if np.any(np.isnan(new_lp)):
                raise ValueError("Probability function returned NaN")
            with np.errstate(divide="ignore"):
                logu = np.log(np.array([rs.rand() for _ in range(ns)]))
            acc = factors + new_lp - self.lp[S1] > logu
            widx = np.flatnonzero(S1)[acc]
            self.X[widx] = q[acc]
            self.lp[widx] = new_lp[acc]
            self.naccepted[widx] += 1
        self.iterations += 1
        if store:
            self.chain.append(self.X.copy())
            self.lps.append(self.lp.copy())

    def run(self, steps, store=True):
        for _ in range(int(steps)):
            self.step(store)


# ------------------------------------------------------------------------------------------------
def walkers_independent(coords):
    """emcee's initial-state check: the walker cloud must span the space (condition number <= 1e8)."""
    if not np.all(np.isfinite(coords)):
        return False
    C = coords - np.mean(coords, axis=0)[None, :]
    C_colmax = np.amax(np.abs(C), axis=0)
    if np.any(C_colmax == 0):
        return False
    C = C / C_colmax
    C = C / np.sqrt(np.sum(C ** 2, axis=0))
    return np.linalg.cond(C.astype(float)) <= 1e8


class State:
    """Minimal emcee.State: ``coords``, ``log_prob``; unpacks / indexes like emcee's
    (``sampler.run_mcmc(...)[0]`` is the coordinate array, ref: mcmc.py:101)."""

    def __init__(self, coords, log_prob=None, blobs=None, random_state=None):
        self.coords = np.atleast_2d(np.asarray(coords, dtype=np.float64))
        self.log_prob = log_prob
        self.blobs = blobs
        self.random_state = random_state

    def __iter__(self):
        return iter((self.coords, self.log_prob, self.random_state))

    def __getitem__(self, i):
        return (self.coords, self.log_prob, self.random_state)[i]

    def __len__(self):
        return 3


class EnsembleSampler:
    """emcee.EnsembleSampler look-alike (the subset the reference uses: ref: mcmc.py:83-116, 187-204 and
    plot_mcmc.py): stretch move a = 2, red/blue split re-drawn every step.

    * ``log_prob_fn`` bound to device models (``bayesian_inference.log_posterior.log_posterior`` carries a
      ``_gpemu_device_models`` hook): the ensemble lives on the GPU (``DeviceSampler``); with
      torch.distributed initialised the proposals are sharded over the ranks.
    * any other callable: host stretch move (``HostEnsemble``), one call per walker like emcee
      (``vectorize=True`` passes the whole half-ensemble), ``pool.map`` if a pool is given.
    """

    def __init__(self, nwalkers, ndim, log_prob_fn, pool=None, a=2.0, seed=None, vectorize=False,
                 args=None, kwargs=None, sharded=None, **_ignored):
        self.nwalkers, self.ndim = int(nwalkers), int(ndim)
        self._sharded = sharded          # None: shard whenever torch.distributed has more than one rank
        self.log_prob_fn = log_prob_fn
        self.pool = pool
        self.a = float(a)
        self.vectorize = vectorize
        self._args, self._kwargs = tuple(args or ()), dict(kwargs or {})
        self._seed = int(np.random.randint(0, 2 ** 31 - 1)) if seed is None else int(seed)
        self._impl = None
        self._cache = None

    # -- backends -------------------------------------------------------------------------------
    @property
    def world_size(self):
        if self._sharded is False:       # an independent chain on this rank (replica parallelism)
            return 1
        from . import dist as gdist
        dist = gdist._dist()             # None in a single process that never imported torch (no ~1 s import)
        if dist is not None and dist.is_initialized():
            return dist.get_world_size()
        return 1

    def _call_rows(self, q):
        f = self.log_prob_fn
        if self.vectorize:
            return np.asarray(f(q, *self._args, **self._kwargs), dtype=np.float64).reshape(-1)
        mapper = self.pool.map if self.pool is not None else map
        vals = list(mapper(_RowCall(f, self._args, self._kwargs), list(q)))
        return np.array([float(np.asarray(v).reshape(-1)[0]) for v in vals])

    def _ensure(self):
        if self._impl is not None:
            return
        if self.world_size > 1:
            # a sharded chain needs the same random stream on every rank (the ranks only exchange
            # log-probabilities): rank 0's seed wins, whether it was given or drawn from numpy's global state
            from . import dist as gdist
            self._seed = gdist.rank0_int(self._seed)
        hook = getattr(self.log_prob_fn, "_gpemu_device_models", None)
        if hook is not None:
            self._impl = DeviceSampler(hook(), self.nwalkers, a=self.a, seed=self._seed)
            self._device = True
        else:
            self._impl = HostEnsemble(self.nwalkers, self.ndim, self._call_rows, a=self.a, seed=self._seed,
                                      sharded=self.world_size > 1)
            self._device = False

    # -- running ----------------------------------------------------------------------------------
    def advance(self, initial_state, nsteps, store=True, want_state=True):
        """Run ``nsteps`` steps (from ``initial_state`` if given, else from the current state).  ``want_state`` = False:
        the ensemble is not downloaded and None is returned (the blocks between two log lines of a long run)."""
        self._ensure()
        self._cache = None
        if initial_state is not None:
            X0 = initial_state.coords if isinstance(initial_state, State) else np.asarray(initial_state, dtype=np.float64)
            if X0.shape != (self.nwalkers, self.ndim):
                raise ValueError("incompatible input dimensions")
            if self.nwalkers < 2 * self.ndim:
                raise RuntimeError("It is unadvisable to use a red-blue move with fewer walkers than twice the "
                                   "number of dimensions.")
            if not walkers_independent(X0):
                raise ValueError("Initial state has a large condition number. Make sure that your walkers are "
                                 "linearly independent for the best performance")
            self._impl.set_state(X0)
            lp0 = self._impl.get_state()[1] if self._device else self._impl.lp
            if np.any(np.isnan(lp0)):
                raise ValueError("The initial log_prob was NaN")
        if self._device and self.world_size > 1:
            self._impl.run_sharded(nsteps, store)
        else:
            self._impl.run(nsteps, store)
        if not want_state:
            return None
        if self._device:
            X, lp = self._impl.get_state()
        else:
            X, lp = self._impl.X.copy(), self._impl.lp.copy()
        return State(X, log_prob=lp)

    def run_mcmc(self, initial_state, nsteps, **kwargs):
        return self.advance(initial_state, nsteps, **kwargs)

    def sample(self, initial_state, iterations=1, store=True, **_ignored):
        """Generator over single steps (emcee's ``sample``)."""
        state = None
        for i in range(int(iterations)):
            state = self.advance(initial_state if i == 0 else None, 1, store=store)
            yield state

    def reset(self):
        self._cache = None
        if self._impl is not None:
            self._impl.reset()

    # -- results ----------------------------------------------------------------------------------
    def _results(self):
        if self._cache is None:
            if self.__dict__.get("_frozen") or self._impl is None:
                chain = np.empty((0, self.nwalkers, self.ndim)); lp = np.empty((0, self.nwalkers))
                nacc = np.zeros(self.nwalkers, dtype=np.int64); it = 0
            elif self._device:
                chain, lp = self._impl.get_chain()
                nacc, it, _ = self._impl.counts()
            else:
                h = self._impl
                chain = np.stack(h.chain) if h.chain else np.empty((0, self.nwalkers, self.ndim))
                lp = np.stack(h.lps) if h.lps else np.empty((0, self.nwalkers))
                nacc, it = h.naccepted.copy(), h.iterations
            self._cache = (chain, lp, nacc, it)
        return self._cache

    def _counts(self):
        """(accepted per walker, iterations) without the chain: the reference logs the acceptance fraction every
        ``n_logging_steps`` (ref: mcmc.py:98-107), and a download of the whole chain so far each time was 0.4 s of a
        3.4 s ``run_mcmc`` at C3."""
        if self._cache is not None:
            return self._cache[2], self._cache[3]
        if self.__dict__.get("_frozen") or self._impl is None:
            return np.zeros(self.nwalkers, dtype=np.int64), 0
        if self._device:
            nacc, it, _ = self._impl.counts()
            return nacc, it
        return self._impl.naccepted.copy(), self._impl.iterations

    @property
    def iteration(self):
        return self._counts()[1]

    @staticmethod
    def _thin(v, discard, thin, flat):
        v = v[discard + thin - 1::thin]
        if flat:
            v = v.reshape((-1,) + v.shape[2:])
        return v

    def get_chain(self, discard=0, thin=1, flat=False):
        return self._thin(self._results()[0], discard, thin, flat)

    def get_log_prob(self, discard=0, thin=1, flat=False):
        return self._thin(self._results()[1], discard, thin, flat)

    @property
    def flatchain(self):
        return self.get_chain(flat=True)

    @property
    def flatlnprobability(self):
        return self.get_log_prob(flat=True)

    @property
    def acceptance_fraction(self):
        nacc, it = self._counts()
        return nacc / float(max(it, 1))

    def get_autocorr_time(self, discard=0, thin=1, **kwargs):
        # the chain still lives on the device: estimate there (no FFTs over a 500 MB host copy); GPEMU_ACF_HOST=1 or a
        # thinned / unpickled sampler takes the host routine
        if (self._impl is not None and getattr(self, "_device", False) and thin == 1 and not self.__dict__.get("_frozen")
                and not os.environ.get("GPEMU_ACF_HOST")):
            kw = {k: kwargs[k] for k in ("c", "tol", "quiet") if k in kwargs}
            if set(kwargs) <= {"c", "tol", "quiet"}:
                return self._impl.integrated_time(first=int(discard), **kw)
        return thin * integrated_time(self.get_chain(discard=discard, thin=thin), **kwargs)

    # -- pickling (ref: mcmc.py:131-132 pickles the sampler) --------------------------------------
    def __getstate__(self):
        st = dict(self.__dict__)
        st["_cache"] = self._results()
        st["_impl"] = None
        st["pool"] = None
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        self._frozen = True

class _RowCall:
    def __init__(self, f, args, kwargs):
        self.f, self.args, self.kwargs = f, args, kwargs

    def __call__(self, x):
        return self.f(x, *self.args, **self.kwargs)
