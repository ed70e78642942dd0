"""``torchfilter.filters`` on MI355X: the particle filter and the virtual-sensor EKF.

API as the reference uses it -- ``ParticleFilter(dynamics_model=, measurement_model=,
num_particles=)`` with public mutable ``num_particles`` / ``dynamics_model`` /
``measurement_model`` (``/root/reference/crossmodal/door_models/pf.py:14-27``,
``train_helpers.py:46,94``); ``VirtualSensorExtendedKalmanFilter(dynamics_model=,
virtual_sensor_model=)`` with ``_belief_mean`` / ``_belief_covariance``
(``door_models/kf.py:14-28``, ``base_models/crossmodal_kf.py:180``) -- with the recursion
itself running in hand-written HIP: ``mmf_pf_init_particles`` for the initial belief, K1
(``mmf_pf_reweight_resample``) for reweight / normalise / estimate / resample / gather, K3
(``mmf_ekf_step``) for the Kalman algebra, and the native step loops
(``mmf_pf_forward_loop`` / ``mmf_ekf_forward_loop``) behind ``forward_loop``.
Step order follows upstream torchfilter.
"""
import ctypes
import math
from typing import Optional

import torch

from . import _abi, base, engine
from .engine import _timed, check_range, require_device, reserve_memory, use_autograd
from .utils import CounterBlock, NoiseSource, tree_index, tree_leading_shape, tree_map

_MODES = {"none": 0, "systematic": 1, "multinomial": 2}


class ParticleFilter(base.Filter):
    """Bootstrap particle filter (T1).

    ``resample=None`` resamples iff ``not self.training`` (upstream behaviour).
    ``soft_resample_alpha < 1`` (upstream option): ancestors are drawn from the
    mixture ``alpha w + (1 - alpha) / M`` and keep the importance weights ``w / mixture``
    (``mmf_pf_reweight_resample_soft``; inside the native loop too: ``MmfPfLoopArgs.soft_alpha``).
    ``resample_mode``: ``"systematic"`` (low variance, one uniform per trajectory; what
    ``north_star`` asks for) or ``"multinomial"`` (upstream's distribution, one uniform per
    particle).  Both use the fixed-point CDF of ``csrc/pf_resample.hip``.

    Belief: ``particle_states (N, M, d)``, ``particle_log_weights (N, M)``.
    Randomness comes from ``self.noise`` (``utils.NoiseSource``), never from global RNG state.
    """

    def __init__(self, *, dynamics_model: base.DynamicsModel,
                 measurement_model: base.ParticleFilterMeasurementModel,
                 num_particles: int = 100, resample: Optional[bool] = None,
                 resample_mode: str = "systematic",
                 estimation_method: str = "weighted_average", soft_resample_alpha: float = 1.0):
        super().__init__(state_dim=dynamics_model.state_dim)
        assert 0.0 < soft_resample_alpha <= 1.0
        self.soft_resample_alpha = soft_resample_alpha
        assert isinstance(dynamics_model, base.DynamicsModel)
        assert isinstance(measurement_model, base.ParticleFilterMeasurementModel)
        assert measurement_model.state_dim == self.state_dim
        assert resample_mode in ("systematic", "multinomial")
        assert estimation_method in ("weighted_average", "argmax")
        self.dynamics_model = dynamics_model
        self.measurement_model = measurement_model
        self.num_particles = num_particles
        self.resample = resample
        self.resample_mode = resample_mode
        self.estimation_method = estimation_method
        self.noise = NoiseSource(0)
        # parity certificates: with record_indices set, every step (and the native loop, per step)
        # keeps its ancestors, the log-likelihoods K2 produced and the log-weights K1 started from
        self.record_indices = False
        self.last_resample_indices = None
        self.last_log_likelihoods = None
        self.last_log_weights_in = None
        self.use_native_loop = True   # False: forward_loop keeps the step-by-step Python loop
        self.particle_states: torch.Tensor = None
        self.particle_log_weights: torch.Tensor = None
        self._spare_states = None
        self._initialized = False

    # ------------------------------------------------------------------ belief
    def initialize_beliefs(self, *, mean: torch.Tensor, covariance: torch.Tensor) -> None:
        N, d = mean.shape
        assert d == self.state_dim
        assert covariance.shape == (N, d, d)
        require_device(mean, "ParticleFilter.initialize_beliefs")
        M = self.num_particles
        eps = self.noise.gaussian((N, M, d), like=mean)
        if use_autograd(self) and (mean.requires_grad or covariance.requires_grad):
            # differentiable initialisation (training): torch ops
            L = torch.linalg.cholesky(covariance.to(torch.float32))
            self.particle_states = (mean[:, None, :] + torch.einsum("nij,nmj->nmi", L, eps)).contiguous()
            self.particle_log_weights = mean.new_full((N, M), -math.log(M))
        else:
            states = torch.empty((N, M, d), dtype=torch.float32, device=mean.device)
            logw = torch.empty((N, M), dtype=torch.float32, device=mean.device)
            not_pd = torch.zeros(1, dtype=torch.int32, device=mean.device)
            _abi.pf_init_particles(mean.detach().to(torch.float32).contiguous(),
                                   covariance.detach().to(torch.float32).contiguous(),
                                   eps.to(torch.float32).contiguous(), states, logw, not_pd)
            if engine.CAPTURING:  # no host read inside a hipGraph capture: bit 16 of the range flag, read after the replay
                engine.range_flag(mean.device).bitwise_or_(not_pd.ne(0).to(torch.int32) * 16)
            elif int(not_pd.item()):
                raise ValueError("initialize_beliefs: covariance is not positive definite")
            self.particle_states, self.particle_log_weights = states, logw
        self._spare_states = None
        self._initialized = True

    def reserve(self, *, steps: int, batch: int, particles: int = None) -> int:
        """Plan memory for ``forward_loop`` over ``steps`` x ``batch`` trajectories: grows the
        allocator once so the loop itself never calls ``hipMalloc``.  Returns the bytes reserved."""
        M = self.num_particles if particles is None else particles
        d = self.state_dim
        dev = next(self.parameters()).device
        per_row = 64 * 4 * 12               # encoder contexts, image features, program outputs
        per_step = batch * M * 4 * (2 * d + 4)  # particle ping-pong, log-weights, log-lik, noise views
        # the image-encoder workspace is persistent: create it now, outside the reserved block
        if steps * batch > 0:
            engine._image_workspace(dev, min(steps * batch, engine._IMAGE_CHUNK), 2)
        # 8 steps' worth of particle buffers: initialize_beliefs() builds the new belief while the
        # previous run's belief and scratch are still alive (measured: 4x left the first loop at
        # a new length one 12 MB segment short = one stream-draining hipMalloc)
        nbytes = steps * batch * per_row + 8 * per_step + (64 << 20)
        reserve_memory(dev, nbytes)
        return nbytes

    def _adapt_particle_count(self) -> None:
        """Upstream torchfilter's particle-count adaptation for steps that do not resample
        (the reference flips 30 <-> 300 in ``train()``,
        ``/root/reference/crossmodal/door_models/pf.py:24-27``, so a train-mode step right after
        an eval-mode belief lands here): the first ``(M_new // M) * M`` slots are whole copies of
        the particle set, the rest a sample without replacement (one permutation shared by the
        batch -- drawn from ``self.noise``: the arg-sort of ``M`` uniforms -- where upstream calls
        ``torch.randperm``); log-weights are gathered alongside and re-normalised."""
        N, M, d = self.particle_states.shape
        Mo = int(self.num_particles)
        dev = self.particle_states.device
        copies = (Mo // M) * M
        parts = []
        if copies > 0:
            parts.append(torch.arange(M, device=dev).repeat(copies // M))
        if Mo - copies > 0:
            perm = torch.argsort(self.noise.uniform((M,), like=self.particle_states), stable=True)
            parts.append(perm[:Mo - copies])
        idx = torch.cat(parts)[None, :].expand(N, Mo)
        self.particle_states = torch.gather(self.particle_states, 1, idx[:, :, None].expand(N, Mo, d)).contiguous()
        lw = torch.gather(self.particle_log_weights, 1, idx)
        self.particle_log_weights = (lw - torch.logsumexp(lw, dim=1, keepdim=True)).contiguous()
        self._spare_states = None

    # ------------------------------------------------------------------ one step
    def _propagate(self, controls, ctrl_ctx, N, M, d):
        eps = self.noise.gaussian((N, M, d), like=self.particle_states)
        dyn = self.dynamics_model
        if hasattr(dyn, "propagate_encoded"):
            if ctrl_ctx is None:
                ctrl_ctx = dyn.encode_controls(controls)
            spare = self._spare_states
            if spare is not None and spare.shape != self.particle_states.shape:
                spare = None
            return dyn.propagate_encoded(self.particle_states, ctrl_ctx, eps, out=spare)
        # generic user model (torch ops on the device): same control for a trajectory's particles
        flat = self.particle_states.reshape(N * M, d)
        rep = tree_map(controls, lambda t: torch.repeat_interleave(t, repeats=M, dim=0))
        pred, tril = dyn(initial_states=flat, controls=rep)
        return (pred + torch.einsum("rij,rj->ri", tril, eps.reshape(N * M, d))).reshape(N, M, d).contiguous()

    def _measure(self, states, observations, obs_ctx):
        meas = self.measurement_model
        if hasattr(meas, "forward_encoded"):
            if obs_ctx is None:
                obs_ctx = meas.encode_observations(observations)
            return meas.forward_encoded(states, obs_ctx)
        return meas(states=states, observations=observations).to(torch.float32).contiguous()

    def _step(self, observations, controls, obs_ctx=None, ctrl_ctx=None) -> torch.Tensor:
        assert self._initialized, "Particle filter not initialized!"
        N, M, d = self.particle_states.shape
        do_resample = (not self.training) if self.resample is None else bool(self.resample)
        if not do_resample and self.num_particles != M:
            self._adapt_particle_count()
            N, M, d = self.particle_states.shape

        with torch.no_grad():
            states = self._propagate(controls, ctrl_ctx, N, M, d)
            loglik = self._measure(states, observations, obs_ctx)
            assert loglik.shape == (N, M)
            if self.record_indices:
                self.last_log_likelihoods, self.last_log_weights_in = loglik, self.particle_log_weights

            estimate = torch.empty((N, d), dtype=torch.float32, device=states.device)
            if do_resample:
                Mo = self.num_particles
                mode = _MODES[self.resample_mode]
                u = self.noise.uniform((N,) if mode == 1 else (N, Mo), like=states)
                # two buffers ping-pong: dynamics wrote `states`; the old belief is free again
                out = self.particle_states
                if out.shape != (N, Mo, d) or out.data_ptr() == states.data_ptr():
                    out = torch.empty((N, Mo, d), dtype=torch.float32, device=states.device)
                logw_out = torch.empty((N, Mo), dtype=torch.float32, device=states.device)
                idx = (torch.empty((N, Mo), dtype=torch.int32, device=states.device)
                       if self.record_indices else None)
                lw_in = self.particle_log_weights
                _timed("pf_reweight_resample", 0.0, N * M * 4.0 * (2 + d) + N * Mo * 4.0 * d,
                       lambda: _abi.pf_reweight_resample(loglik, lw_in, states, u, estimate,
                                                         out, logw_out, idx, mode, self.soft_resample_alpha))
                self._spare_states = states
                self.last_resample_indices = idx
            else:
                out, logw_out = states, torch.empty_like(loglik)
                if states.data_ptr() != self.particle_states.data_ptr():
                    self._spare_states = self.particle_states  # the old belief is the next scratch
                _abi.pf_reweight_resample(loglik, self.particle_log_weights, states, None, estimate,
                                          None, logw_out, None, 0)
            if self.estimation_method == "argmax":
                # arg-max of the *pre-resampling* normalised weights
                tot = self.particle_log_weights + loglik
                best = torch.argmax(tot, dim=1)
                estimate = states[torch.arange(N, device=states.device), best]
            self.particle_states = out
            self.particle_log_weights = logw_out
        return estimate

    def _step_autograd(self, observations, controls, dyn_bias=None, meas_ctx=None) -> torch.Tensor:
        """Differentiable torch formulation of the step (training backend "autograd"): gradients
        flow through the reparameterised noise and the log-weights; resampling, when requested,
        runs through K1 on detached tensors (it stops gradients upstream as well)."""
        assert self._initialized, "Particle filter not initialized!"
        N, M, d = self.particle_states.shape
        do_resample = (not self.training) if self.resample is None else bool(self.resample)
        if not do_resample and self.num_particles != M:
            self._adapt_particle_count()
            N, M, d = self.particle_states.shape
        if engine.use_hip_backward() and hasattr(self.dynamics_model, "forward_particles"):
            # K6: the N*M-row network evaluates and differentiates in HIP
            pred = self.dynamics_model.forward_particles(states=self.particle_states, controls=controls,
                                                         bias=dyn_bias)
            eps = self.noise.gaussian((N, M, d), like=pred)
            states = pred + eps @ self.dynamics_model.scale_tril().t()
        else:
            flat = self.particle_states.reshape(N * M, d)
            rep = tree_map(controls, lambda t: torch.repeat_interleave(t, repeats=M, dim=0))
            pred, tril = self.dynamics_model(initial_states=flat, controls=rep)
            eps = self.noise.gaussian((N, M, d), like=pred).reshape(N * M, d)
            states = (pred + torch.einsum("rij,rj->ri", tril, eps)).reshape(N, M, d)
        if meas_ctx is not None:
            loglik = self.measurement_model.forward_encoded_autograd(states, meas_ctx)
        else:
            loglik = self.measurement_model(states=states, observations=observations)
        if engine.use_hip_backward() and self.estimation_method == "weighted_average":
            # K6: reweight + normalise + estimate forward (K1 mode 0) and backward in HIP
            estimate, logw = engine.ReweightEstimateFunction.apply(loglik, self.particle_log_weights, states)
        else:
            logw = self.particle_log_weights + loglik
            logw = logw - torch.logsumexp(logw, dim=1, keepdim=True)
            if self.estimation_method == "weighted_average":
                estimate = torch.sum(torch.exp(logw)[:, :, None] * states, dim=1)
            else:
                estimate = states[torch.arange(N, device=states.device), torch.argmax(logw, dim=1)]
        self.particle_states, self.particle_log_weights = states, logw
        if do_resample:
            Mo = self.num_particles
            mode = _MODES[self.resample_mode]
            u = self.noise.uniform((N,) if mode == 1 else (N, Mo), like=states)
            out = torch.empty((N, Mo, d), dtype=torch.float32, device=states.device)
            logw_out = torch.empty((N, Mo), dtype=torch.float32, device=states.device)
            scratch = torch.empty((N, d), dtype=torch.float32, device=states.device)
            soft = self.soft_resample_alpha < 1.0
            idx = torch.empty((N, Mo), dtype=torch.int32, device=states.device) if soft else None
            with torch.no_grad():
                _abi.pf_reweight_resample(torch.zeros_like(logw), logw.detach().contiguous(),
                                          states.detach().contiguous(), u, scratch, out, logw_out, idx, mode,
                                          self.soft_resample_alpha)
            if soft:
                # upstream's soft resampling is differentiable: the ancestors come from K1, the
                # survivors' states and importance weights are re-derived with torch ops so that
                # gradients reach the pre-resampling weights and particles
                a = self.soft_resample_alpha
                gi = idx.long()
                mix = torch.logaddexp(logw + math.log(a), torch.full_like(logw, math.log((1.0 - a) / M)))
                new = torch.gather(logw - mix, 1, gi)
                out = torch.gather(states, 1, gi[:, :, None].expand(N, Mo, d))
                logw_out = new - torch.logsumexp(new, dim=1, keepdim=True)
            self.particle_states, self.particle_log_weights = out, logw_out
        return estimate

    def _native_loop(self, obs_all, ctrl_all, T, N):
        """All ``T`` steps through ``mmf_pf_forward_loop`` (one C call, no per-step Python) when
        both models are fused networks; ``None`` -> the caller runs the step-by-step loop."""
        dyn, meas = self.dynamics_model, self.measurement_model
        if obs_all is None or ctrl_all is None or not hasattr(dyn, "_net") or not hasattr(meas, "fused_measurements"):
            return None
        if not self.use_native_loop:
            return None
        plan = meas.fused_measurements(obs_all)
        if plan is None or T == 0:
            return None
        nets, stride = plan
        Nb, M, d = self.particle_states.shape
        do_resample = (not self.training) if self.resample is None else bool(self.resample)
        if Nb != N or self.num_particles != M or len(nets) > _abi.LOOP_MAX_MEAS:
            return None
        assert self._initialized, "Particle filter not initialized!"
        mode = _MODES[self.resample_mode] if do_resample else 0
        like = self.particle_states
        u_shape = None if mode == 0 else ((N,) if mode == 1 else (N, M))
        eps, u = self.noise.draw_steps(T, (N, M, d), u_shape, like=like)
        dev = like.device
        states_a = like.contiguous()
        states_b = self._spare_states if (self._spare_states is not None and self._spare_states.shape == states_a.shape
                                          and self._spare_states.data_ptr() != states_a.data_ptr()) else torch.empty_like(states_a)
        logw_a = self.particle_log_weights.contiguous()
        logw_b = torch.empty_like(logw_a)
        loglik = torch.empty_like(logw_a)
        est = torch.empty((T, N, d), dtype=torch.float32, device=dev)
        tril = dyn.scale_tril().contiguous()
        keep = [nets, eps, u, tril]  # keep every operand alive until the launches are enqueued
        P = lambda t: None if t is None else ctypes.c_void_p(_abi.ptr(t))
        a = _abi.MmfPfLoopArgs()
        a.T, a.N, a.M, a.d, a.n_meas, a.resample_mode = T, N, M, d, len(nets), mode
        a.precision = dyn._net.precision_code()
        a.n_res_dyn, a.n_res_meas, a.logw_stride = dyn._net.n_res, nets[0][0].n_res, stride
        a.dyn_packed, a.dyn_bias = P(dyn._net.blob()), P(ctrl_all["bias"])
        for k, (net, bias, lw) in enumerate(nets):
            a.meas_packed[k], a.meas_bias[k], a.meas_logw[k] = P(net.blob()), P(bias), P(lw)
        if isinstance(eps, CounterBlock):   # counter-based noise: generated inside the dynamics kernel
            a.noise, a.noise_mode = None, 2
            a.noise_seed, a.noise_step0, a.noise_traj0 = eps.seed, eps.step0, eps.traj0
        else:
            a.noise = P(eps)
        a.scale_tril, a.uniforms = P(tril), P(u)
        a.states_a, a.states_b, a.logw_a, a.logw_b = P(states_a), P(states_b), P(logw_a), P(logw_b)
        a.loglik, a.estimates = P(loglik), P(est)
        if self.record_indices:
            self.last_log_weights_in = logw_a.clone()
            self.last_log_likelihoods = torch.empty((T, N, M), dtype=torch.float32, device=dev)
            a.loglik_steps = P(self.last_log_likelihoods)
            if mode != 0:
                self.last_resample_indices = torch.empty((T, N, M), dtype=torch.int32, device=dev)
                a.indices_steps = ctypes.c_void_p(_abi.ptr(self.last_resample_indices, dtype=torch.int32))
        a.range_flag = ctypes.c_void_p(engine.range_flag(dev).data_ptr())
        if do_resample and self.soft_resample_alpha < 1.0:
            a.soft_alpha = float(self.soft_resample_alpha)  # survivors carry importance weights (mmf_pf_reweight_resample_soft)
        if self.estimation_method == "argmax":
            est_scratch = torch.empty((N, d), dtype=torch.float32, device=dev)
            keep.append(est_scratch)
            a.estimate_argmax, a.estimate_scratch = 1, P(est_scratch)
        timer = engine.kernel_timer()
        if (engine.PF_PERSISTENT and mode == 1 and timer is None and not self.record_indices
                and a.soft_alpha == 0.0 and not a.estimate_argmax and d in (2, 3)
                and dyn._net.n_res == 3 and all(net.n_res == 2 for net, _b, _l in nets)
                and _abi.pf_persistent_plan(N, M, len(nets)) > 0):
            # small problem: ONE launch for all T steps (csrc/pf_persistent.inc); same bits as the loop of launches
            n_words = _abi.pf_persistent_sync_words(N, M, d, len(nets))
            sync = torch.empty(n_words, dtype=torch.int32, device=dev)  # tagged granules of the hand-offs (zeroed by the call)
            keep.append(sync)
            a.persistent, a.n_sync_words = 1, n_words
            a.sync_words = ctypes.c_void_p(_abi.ptr(sync, dtype=torch.int32))
            # the persistent launch needs ALL its workgroups resident; if it gives up (another process on this GPU),
            # the loop is re-run from this copy of the belief as a loop of launches -- see below
            belief_backup = (states_a.clone(), logw_a.clone())
        events = None
        names = ["particle_net_dynamics"] + ["particle_net_measure"] * len(nets) + ["pf_reweight_resample"]
        stride = 1
        if timer is not None:
            stride = max(1, int(timer.loop_stride))
            events = timer.loop_events(2 * len(names) * len(range(stride // 2, T, stride)))  # pf_loop.hip samples t % stride == stride // 2
        loc = _abi.pf_forward_loop(a, like, events, stride)
        if a.persistent and engine.persistent_loop_gave_up(dev):
            # bounded spins ran out (a workgroup of the launch was not resident): nothing of this call can be used.
            # Restore the belief, take the launch-per-step path for this call and for the rest of the process.
            states_a.copy_(belief_backup[0])
            logw_a.copy_(belief_backup[1])
            a.persistent = 0
            loc = _abi.pf_forward_loop(a, like, events, stride)
        if timer is not None:
            R = N * M
            dflops = 2.0 * R * engine.particle_net_macs(d, dyn._net.n_res, dyn._net.n_out)
            work = [(dflops, R * 4.0 * 3 * d)]
            mwork = [(2.0 * R * engine.particle_net_macs(d, net.n_res, net.n_out), R * 4.0 * (d + 1 + (k > 0)))
                     for k, (net, _, _) in enumerate(nets)]
            work += mwork
            work.append((0.0, R * 4.0 * (2 + 2 * d)))
            timer.add_loop_records(names, work, events)
        self.particle_states = states_b if loc & 1 else states_a
        self._spare_states = states_a if loc & 1 else states_b
        self.particle_log_weights = logw_b if loc & 2 else logw_a
        del keep
        return est

    def _native_train_loop(self, dyn_all, meas_all, T, N):
        """K6, whole recursion (``engine.PfTrainLoopFunction``): all ``T`` train-mode steps forward in one C
        call, backward in one; ``None`` -> the caller's step-by-step autograd loop (user models, resampling
        while training, ``argmax`` estimates, a belief of another particle count)."""
        dyn, meas = self.dynamics_model, self.measurement_model
        do_resample = (not self.training) if self.resample is None else bool(self.resample)
        if (dyn_all is None or meas_all is None or not hasattr(dyn, "_net") or not hasattr(meas, "train_plan")
                or do_resample or self.estimation_method != "weighted_average" or T == 0 or not self.use_native_loop):
            return None
        assert self._initialized, "Particle filter not initialized!"
        Nb, M, d = self.particle_states.shape
        if Nb != N or self.num_particles != M:
            return None
        plan = meas.train_plan(meas_all)
        if plan is None:
            return None
        nets, biases, beta, K_all = plan
        if len(nets) > _abi.LOOP_MAX_MEAS:
            return None
        # the native recursion returns no gradient for the process-noise factor (the reference's models freeze Q)
        # and takes ONE depth for all measurement networks: a trainable / state-dependent Q or networks of
        # different depths keep the step-by-step autograd loop, which differentiates `eps @ scale_tril^T`
        if dyn.scale_tril().requires_grad or len({net.n_res for net, _col in nets}) != 1:
            return None
        eps, _ = self.noise.draw_steps(T, (N, M, d), None, like=self.particle_states)
        if isinstance(eps, CounterBlock):  # the training recursion reads its noise from a tensor
            blk = eps
            eps = torch.empty((T, N, M, d), dtype=torch.float32, device=self.particle_states.device)
            for t in range(T):
                _abi.philox_normals(blk.seed, blk.step0 + t, blk.traj0, eps[t])
        params = list(dyn._net._sources())
        for net, _col in nets:
            params += net._sources()
        empty = torch.empty(0, dtype=torch.float32, device=self.particle_states.device)
        est, states, logw = engine.PfTrainLoopFunction.apply(
            (dyn._net, nets, K_all), T, N, M, self.particle_states, self.particle_log_weights, eps,
            dyn.scale_tril(), dyn_all, beta if beta is not None else empty, *biases, *params)
        self.particle_states, self.particle_log_weights = states, logw
        self._spare_states = None
        return est

    @engine.checked_step
    def forward(self, *, observations, controls) -> torch.Tensor:
        if use_autograd(self):
            return self._step_autograd(observations, controls)
        return self._step(observations, controls)

    @engine.checked_loop
    def forward_loop(self, *, observations, controls) -> torch.Tensor:
        """Sequential in ``t``.  Everything that does not depend on the belief (control and
        observation encoders, image CNNs, modality weights) is evaluated ahead of the
        recursion for all ``T*N`` rows at once: rows are independent, so the per-trajectory
        work costs one launch sequence per ``forward_loop`` instead of one per step."""
        T, N = tree_leading_shape(controls)[:2]
        assert tree_leading_shape(observations)[:2] == (T, N)
        flat = lambda x: x.reshape((T * N,) + tuple(x.shape[2:]))
        if use_autograd(self):
            if not engine.use_hip_backward():
                return base.Filter.forward_loop(self, observations=observations, controls=controls)
            # training, K6 backend: the per-trajectory networks (image CNNs, encoders, weight
            # model) are differentiable torch ops evaluated ONCE on the T*N flattened rows
            dyn_all = meas_all = None
            if hasattr(self.dynamics_model, "encode_controls_autograd"):
                dyn_all = self.dynamics_model.encode_controls_autograd(tree_map(controls, flat))
            if hasattr(self.measurement_model, "encode_observations_autograd"):
                meas_all = self.measurement_model.encode_observations_autograd(tree_map(observations, flat))
            native = self._native_train_loop(dyn_all, meas_all, T, N)
            if native is not None:
                return native
            out = []
            for t in range(T):
                sl = slice(t * N, (t + 1) * N)
                out.append(self._step_autograd(
                    tree_index(observations, t), tree_index(controls, t),
                    None if dyn_all is None else dyn_all[sl],
                    None if meas_all is None else {k: v[sl] for k, v in meas_all.items()}))
            return torch.stack(out, dim=0)
        obs_all = ctrl_all = None
        with torch.no_grad():
            if hasattr(self.measurement_model, "forward_encoded"):
                obs_all = self.measurement_model.encode_observations(tree_map(observations, flat))
            if hasattr(self.dynamics_model, "propagate_encoded"):
                ctrl_all = self.dynamics_model.encode_controls(tree_map(controls, flat))
            native = self._native_loop(obs_all, ctrl_all, T, N)
        if native is not None:
            return native
        out = []
        for t in range(T):
            sl = slice(t * N, (t + 1) * N)
            out.append(self._step(
                tree_index(observations, t), tree_index(controls, t),
                None if obs_all is None else {k: v[sl] for k, v in obs_all.items()},
                None if ctrl_all is None else {k: v[sl] for k, v in ctrl_all.items()}))
        return torch.stack(out, dim=0)


class VirtualSensorExtendedKalmanFilter(base.Filter):
    """EKF whose measurement is a learned virtual sensor ``(z, R^1/2)`` observed through
    ``C = I`` (T2).  predict: ``S- = A S A^T + L L^T`` with ``A`` the dynamics Jacobian;
    correct: ``K = S-(S- + R)^-1``, ``mu = mu- + K(z - mu-)``, ``S = (I - K) S-``."""

    def __init__(self, *, dynamics_model: base.DynamicsModel,
                 virtual_sensor_model: base.VirtualSensorModel):
        super().__init__(state_dim=dynamics_model.state_dim)
        assert isinstance(dynamics_model, base.DynamicsModel)
        assert isinstance(virtual_sensor_model, base.VirtualSensorModel)
        self.dynamics_model = dynamics_model
        self.virtual_sensor_model = virtual_sensor_model
        self._belief_mean = None
        self._belief_covariance = None
        self._initialized = False

    @property
    def belief_mean(self):
        return self._belief_mean

    @belief_mean.setter
    def belief_mean(self, v):
        self._belief_mean = v

    @property
    def belief_covariance(self):
        return self._belief_covariance

    @belief_covariance.setter
    def belief_covariance(self, v):
        self._belief_covariance = v

    def initialize_beliefs(self, *, mean, covariance):
        N, d = mean.shape
        assert d == self.state_dim
        assert covariance.shape == (N, d, d)
        require_device(mean, "VirtualSensorExtendedKalmanFilter.initialize_beliefs")
        self._belief_mean = mean.to(torch.float32).contiguous().clone()
        self._belief_covariance = covariance.to(torch.float32).contiguous().clone()
        self._initialized = True

    def _predict_pieces(self, controls, ctrl_ctx=None):
        """``(mu-, A, L)`` for the current belief mean; ``L`` is ``(d, d)`` (constant noise)."""
        dyn = self.dynamics_model
        mu = self._belief_mean
        if hasattr(dyn, "predict_with_jacobian"):
            if ctrl_ctx is None:
                ctrl_ctx = dyn.encode_controls(controls)
            return dyn.predict_with_jacobian(mu, ctrl_ctx)
        mu_pred, tril = dyn(initial_states=mu, controls=controls)
        A = dyn.jacobian(initial_states=mu, controls=controls)
        # the C ABI takes one scale_tril per sub-filter: user models must keep it constant
        return mu_pred.detach().contiguous(), A.detach().contiguous(), tril[0].detach().contiguous()

    def _step(self, observations, controls, sensor_out=None, ctrl_ctx=None):
        assert self._initialized, "Kalman filter not initialized!"
        with torch.no_grad():
            z, r_tril = sensor_out if sensor_out is not None else self.virtual_sensor_model(observations=observations)
            mu_pred, A, L = self._predict_pieces(controls, ctrl_ctx)
            N, d = mu_pred.shape
            mu = torch.empty((1, N, d), dtype=torch.float32, device=mu_pred.device)
            Sigma = self._belief_covariance.reshape(1, N, d, d).clone()
            _abi.ekf_step(A.reshape(1, N, d, d), mu_pred.reshape(1, N, d), L.reshape(1, d, d).contiguous(),
                          z.to(torch.float32).reshape(1, N, d).contiguous(),
                          r_tril.to(torch.float32).reshape(1, N, d, d).contiguous(), None,
                          mu, Sigma, None, None, fusion=0, feedback=0)
            self._belief_mean, self._belief_covariance = mu[0], Sigma[0]
        return self._belief_mean

    def _step_autograd(self, observations, controls):
        """Differentiable torch formulation (training backend "autograd")."""
        assert self._initialized, "Kalman filter not initialized!"
        mu, Sigma = self._belief_mean, self._belief_covariance
        z, r_tril = self.virtual_sensor_model(observations=observations)
        dyn = self.dynamics_model
        if engine.use_hip_backward() and hasattr(dyn, "predict_with_jacobian_autograd"):
            # K6: dynamics network + forward-mode Jacobian forward and backward in HIP
            mu_pred, A = dyn.predict_with_jacobian_autograd(mu, controls)
            L = dyn.scale_tril()[None]
        else:
            mu_pred, L = dyn(initial_states=mu, controls=controls)
            A = dyn.jacobian(initial_states=mu, controls=controls)
        # K6's Kalman step takes ONE (d, d) process-noise factor and returns no gradient for it: right for
        # the reference's models (constant Q, requires_grad=False), wrong for a user model whose
        # scale_tril depends on the state / control or is trainable -- those keep the torch algebra below
        q_const = (not L.requires_grad) and (L.shape[0] == 1 or bool((L == L[:1]).all()))
        if engine.use_hip_backward() and q_const:
            # K6: the Kalman algebra forward (K3) and backward (closed-form adjoints) in HIP; the
            # networks around it (sensor, dynamics, Jacobian) keep their autograd form
            self._belief_mean, self._belief_covariance = engine.EkfStepFunction.apply(A, mu_pred, L[0], z, r_tril, Sigma)
            return self._belief_mean
        Sp = A @ Sigma @ A.transpose(-1, -2) + L @ L.transpose(-1, -2)
        K = Sp @ torch.inverse(Sp + r_tril @ r_tril.transpose(-1, -2))
        self._belief_mean = mu_pred + (K @ (z - mu_pred)[:, :, None]).squeeze(-1)
        self._belief_covariance = (torch.eye(K.shape[-1], device=K.device) - K) @ Sp
        return self._belief_mean

    @engine.checked_step
    def forward(self, *, observations, controls):
        if use_autograd(self):
            return self._step_autograd(observations, controls)
        return self._step(observations, controls)

    def _native_loop(self, observations, ctrl_all, T, N, flat):
        """All ``T`` steps through ``mmf_ekf_forward_loop`` (K = 1, no fusion) when the virtual
        sensor is row-wise (so it can be evaluated on the ``T*N`` flattened rows at once) and
        the dynamics model is a fused network; ``None`` -> Python loop."""
        dyn, vs = self.dynamics_model, self.virtual_sensor_model
        if ctrl_all is None or T == 0 or not hasattr(dyn, "_net") or not getattr(vs, "row_wise", False):
            return None
        assert self._initialized, "Kalman filter not initialized!"
        d = self.state_dim
        z, r = vs(observations=tree_map(observations, flat))
        z = z.to(torch.float32).reshape(T, 1, N, d).contiguous()
        r = r.to(torch.float32).reshape(T, 1, N, d, d).contiguous()
        mu = self._belief_mean.reshape(1, N, d).contiguous().clone()
        Sigma = self._belief_covariance.reshape(1, N, d, d).contiguous().clone()
        q = dyn.scale_tril().to(torch.float32).reshape(1, d, d).contiguous()
        mu_pred, A = torch.empty_like(mu), torch.empty_like(Sigma)
        est = torch.empty((T, N, d), dtype=torch.float32, device=mu.device)
        prec = dyn._net.precision_code()
        blob = dyn._net.blob(prec)
        P = lambda t: ctypes.c_void_p(_abi.ptr(t))
        a = _abi.MmfEkfLoopArgs()
        a.T, a.N, a.d, a.K, a.fusion, a.feedback = T, N, d, 1, 0, 0
        a.n_res_dyn, a.precision = dyn._net.n_res, prec
        a.range_flag = (ctypes.c_void_p(_abi.ptr(engine.range_flag(mu.device), dtype=torch.int32))
                        if prec != _abi.PREC_F32 else None)
        a.dyn_packed[0], a.dyn_bias[0] = P(blob), P(ctrl_all["bias"])
        a.q_tril, a.z, a.r_tril = P(q), P(z), P(r)
        a.mu, a.Sigma, a.mu_pred, a.A, a.estimates = P(mu), P(Sigma), P(mu_pred), P(A), P(est)
        engine.run_ekf_loop(a, mu, Sigma)
        self._belief_mean, self._belief_covariance = mu[0], Sigma[0]
        return est

    @engine.checked_loop
    def forward_loop(self, *, observations, controls):
        if use_autograd(self):
            return base.Filter.forward_loop(self, observations=observations, controls=controls)
        T, N = tree_leading_shape(controls)[:2]
        flat = lambda t: t.reshape((T * N,) + tuple(t.shape[2:]))
        with torch.no_grad():
            ctrl_all = None
            if hasattr(self.dynamics_model, "predict_with_jacobian"):
                ctrl_all = self.dynamics_model.encode_controls(tree_map(controls, flat))
            native = self._native_loop(observations, ctrl_all, T, N, flat)
            if native is not None:
                return native
            sensors = [self.virtual_sensor_model(observations=tree_index(observations, t)) for t in range(T)]
        out = []
        for t in range(T):
            sl = slice(t * N, (t + 1) * N)
            out.append(self._step(tree_index(observations, t), tree_index(controls, t), sensors[t],
                                  None if ctrl_all is None else {k: v[sl] for k, v in ctrl_all.items()}))
        return torch.stack(out, dim=0)


# ------------------------------------------------------------------------------ unscented filter
class JulierSigmaPointStrategy:
    """``lambda = 3 - d`` unless given; mean and covariance weights coincide (upstream
    ``torchfilter.utils.JulierSigmaPointStrategy``, its UKFs' default)."""

    def __init__(self, lambd: Optional[float] = None):
        self.lambd = lambd

    def compute_lambda(self, dim: int) -> float:
        return 3.0 - dim if self.lambd is None else float(self.lambd)

    def compute_sigma_weights(self, dim: int):
        """``(wc0, wm0, wi)``: covariance / mean weight of the central point, weight of the rest."""
        lambd = self.compute_lambda(dim)
        w0 = lambd / (dim + lambd)
        return w0, w0, 1.0 / (2.0 * (dim + lambd))


class MerweSigmaPointStrategy:
    """Van der Merwe's scaled points: ``lambda = alpha^2 (d + kappa) - d``, ``kappa = 3 - d`` unless
    given; ``wc0 = wm0 + 1 - alpha^2 + beta`` (upstream ``torchfilter.utils.MerweSigmaPointStrategy``)."""

    def __init__(self, alpha: float = 1e-2, beta: float = 2.0, kappa: Optional[float] = None):
        self.alpha, self.beta, self.kappa = alpha, beta, kappa

    def compute_lambda(self, dim: int) -> float:
        kappa = 3.0 - dim if self.kappa is None else float(self.kappa)
        return self.alpha ** 2 * (dim + kappa) - dim

    def compute_sigma_weights(self, dim: int):
        lambd = self.compute_lambda(dim)
        wm0 = lambd / (dim + lambd)
        return wm0 + 1.0 - self.alpha ** 2 + self.beta, wm0, 1.0 / (2.0 * (dim + lambd))


class VirtualSensorUnscentedKalmanFilter(VirtualSensorExtendedKalmanFilter):
    """UKF whose measurement is a learned virtual sensor observed through ``C = I`` (SURVEY.md 8f
    rank 2; upstream ``torchfilter.filters.VirtualSensorUnscentedKalmanFilter``, absent and
    un-pinned: published algorithm, restated in ``oracle/tf/filters.py``).

    predict: ``2d + 1`` sigma points of the belief (``mmf_ukf_sigma_points``) go through the
    dynamics model -- for the built-in networks they are rows of K2, ``N x (2d+1)`` "particles"
    without noise -- and ``mmf_ukf_moments`` forms ``mu-`` and ``Sigma- = sum wc (X - mu-)(X - mu-)^T +
    Q``; no Jacobian is evaluated.  correct: with an identity measurement the unscented update is
    the Kalman update on ``(mu-, Sigma-)``, i.e. K3 with ``A = I`` and no added noise.
    Same belief attributes and ``forward`` / ``forward_loop`` contract as the EKF."""

    def __init__(self, *, dynamics_model, virtual_sensor_model, sigma_point_strategy=None):
        super().__init__(dynamics_model=dynamics_model, virtual_sensor_model=virtual_sensor_model)
        self.sigma_point_strategy = sigma_point_strategy if sigma_point_strategy is not None else JulierSigmaPointStrategy()

    def _unscented_predict(self, controls, ctrl_ctx=None, not_pd=None):
        """``not_pd``: a loop-wide device flag (checked by the caller after the loop) -- ``None`` checks
        this step's own flag right away (one blocking 4-byte read)."""
        dyn = self.dynamics_model
        mu, Sigma = self._belief_mean.contiguous(), self._belief_covariance.contiguous()
        N, d = mu.shape
        P = 2 * d + 1
        lambd = self.sigma_point_strategy.compute_lambda(d)
        wc0, wm0, wi = self.sigma_point_strategy.compute_sigma_weights(d)
        points = torch.empty((N, P, d), dtype=torch.float32, device=mu.device)
        own = not_pd is None
        if own:
            not_pd = torch.zeros(1, dtype=torch.int32, device=mu.device)
        _abi.ukf_sigma_points(mu, Sigma, math.sqrt(d + lambd), points, not_pd)
        # a non-PD belief collapses its points onto the mean inside the kernel (finite rows for the
        # networks); a single step raises here, a forward_loop once after its last step
        if own and int(not_pd.item()):
            raise ValueError("unscented predict: belief covariance is not positive definite")
        if hasattr(dyn, "propagate_encoded"):
            if ctrl_ctx is None:
                ctrl_ctx = dyn.encode_controls(controls)
            moved = dyn.propagate_encoded(points, ctrl_ctx, None)
            L = dyn.scale_tril().to(torch.float32).contiguous()
        else:  # forward-only user model: one call on the N * (2d+1) flattened rows
            rep = tree_map(controls, lambda t: torch.repeat_interleave(t, repeats=P, dim=0))
            pred, _ = dyn(initial_states=points.reshape(N * P, d), controls=rep)
            moved = pred.reshape(N, P, d).to(torch.float32).contiguous()
            # noise evaluated at the belief mean (upstream); must be constant over the batch for the C ABI
            L = dyn(initial_states=mu, controls=controls)[1][0].detach().to(torch.float32).contiguous()
        mu_pred = torch.empty((N, d), dtype=torch.float32, device=mu.device)
        Sigma_pred = torch.empty((N, d, d), dtype=torch.float32, device=mu.device)
        _abi.ukf_moments(moved, wm0, wc0, wi, L, mu_pred, Sigma_pred)
        return mu_pred, Sigma_pred

    def _step(self, observations, controls, sensor_out=None, ctrl_ctx=None, not_pd=None):
        assert self._initialized, "Kalman filter not initialized!"
        with torch.no_grad():
            z, r_tril = sensor_out if sensor_out is not None else self.virtual_sensor_model(observations=observations)
            mu_pred, Sigma_pred = self._unscented_predict(controls, ctrl_ctx, not_pd)
            N, d = mu_pred.shape
            eye = torch.eye(d, dtype=torch.float32, device=mu_pred.device)[None].expand(N, d, d).contiguous()
            mu = torch.empty((1, N, d), dtype=torch.float32, device=mu_pred.device)
            Sigma = Sigma_pred.reshape(1, N, d, d)
            _abi.ekf_step(eye.reshape(1, N, d, d), mu_pred.reshape(1, N, d),
                          torch.zeros((1, d, d), dtype=torch.float32, device=mu_pred.device),
                          z.to(torch.float32).reshape(1, N, d).contiguous(),
                          r_tril.to(torch.float32).reshape(1, N, d, d).contiguous(), None,
                          mu, Sigma, None, None, fusion=0, feedback=0)
            self._belief_mean, self._belief_covariance = mu[0], Sigma[0]
        return self._belief_mean

    def _step_autograd(self, observations, controls):
        raise NotImplementedError("the unscented filter is evaluation-only here; train the models with the EKF / PF")

    def _native_loop(self, observations, ctrl_all, T, N, flat):
        return None  # the step is sigma points -> K2 -> moments -> K3: driven from Python, sensors batched over T*N

    @engine.checked_loop
    def forward_loop(self, *, observations, controls):
        T, N = tree_leading_shape(controls)[:2]
        flat = lambda t: t.reshape((T * N,) + tuple(t.shape[2:]))
        with torch.no_grad():
            ctrl_all = None
            if hasattr(self.dynamics_model, "propagate_encoded"):
                ctrl_all = self.dynamics_model.encode_controls(tree_map(controls, flat))
            if getattr(self.virtual_sensor_model, "row_wise", False):
                z, r = self.virtual_sensor_model(observations=tree_map(observations, flat))
                sensors = [(z[t * N:(t + 1) * N], r[t * N:(t + 1) * N]) for t in range(T)]
            else:
                sensors = [self.virtual_sensor_model(observations=tree_index(observations, t)) for t in range(T)]
        out = []
        not_pd = torch.zeros(1, dtype=torch.int32, device=self._belief_mean.device)  # one flag, one read per loop
        for t in range(T):
            sl = slice(t * N, (t + 1) * N)
            out.append(self._step(tree_index(observations, t), tree_index(controls, t), sensors[t],
                                  None if ctrl_all is None else {k: v[sl] for k, v in ctrl_all.items()}, not_pd))
        if T > 0 and int(not_pd.item()):
            raise ValueError("unscented predict: belief covariance is not positive definite")
        return torch.stack(out, dim=0)
