"""Training steps: the end-to-end filter step and the pre-training losses of
the reference's curricula (8f rank 4).

Mirrors ``torchfilter.train.train_filter`` as the reference calls it
(``/root/reference/crossmodal/train_helpers.py:124-162``; upstream behaviour SURVEY.md A.2):
a batch of subsequences ``(L, N, ...)``, belief initialised at a sample
``~ N(states[0], initial_covariance)`` (or from the first observation), ``forward_loop`` over
``[1:]``, mean-squared error against ``states[1:]``, one optimiser step.  Randomness is explicit
(``noise``), as everywhere in this package.

Pre-training (``train_helpers.py:31-121`` -> ``torchfilter.train.train_dynamics_single_step`` /
``train_dynamics_recurrent`` / ``train_particle_filter_measurement`` / ``train_virtual_sensor``; the
package is absent and un-pinned, its published losses are restated -- same formulas as
``oracle/tf/train.py``): ``dynamics_single_step_loss``, ``dynamics_recurrent_loss``,
``particle_filter_measurement_loss``, ``virtual_sensor_loss`` and ``pretrain_step``; batches come
from ``data.SingleStepBatcher`` / ``data.SubsequenceBatcher`` /
``data.ParticleFilterMeasurementBatcher``.

The filter must be in ``train()`` mode with a training backend selected
(``engine.set_training_backend("hip")``: per-particle networks forward + backward in HIP,
K6; ``"autograd"``: torch ops throughout).  With ``torch.distributed`` initialised,
``all_reduce=True`` averages the gradients over ranks before the step (X2).
"""
from typing import Dict, Optional

import torch

from . import distributed, engine
from .utils import NoiseSource

# initial-belief perturbations when neither the caller nor the filter brings a noise source: ONE
# module-level generator that advances from step to step (upstream ``train_filter`` samples a
# fresh ``MultivariateNormal`` per batch), seeded per data-parallel rank on first use
_DEFAULT_NOISE = NoiseSource(20201025)


def default_noise(filter_model) -> NoiseSource:
    """The persistent source a training step draws from by default: the filter's own ``noise``
    (particle filters carry one) or the module-level one -- never a fresh seed-0 generator."""
    own = getattr(filter_model, "noise", None)
    return own if type(own) is NoiseSource else _DEFAULT_NOISE  # replayed / stacked blocks are the steps' own


def filter_loss(filter_model, batch: Dict[str, torch.Tensor], *, initial_covariance: torch.Tensor,
                noise: Optional[NoiseSource] = None, measurement_initialize: bool = False,
                initial_scale_tril: Optional[torch.Tensor] = None) -> torch.Tensor:
    """MSE of ``forward_loop`` on ``batch`` = ``{"states" (L, N, d), "controls" (L, N, 7), "image",
    "gripper_pos", "gripper_sensors" (L, N, ...)}`` (time-major)."""
    assert filter_model.training, "call filter_model.train() first"
    assert engine.TRAINING_BACKEND is not None, "select engine.set_training_backend('hip' | 'autograd')"
    states = batch["states"]
    L, N, d = states.shape
    obs = {k: batch[k] for k in ("image", "gripper_pos", "gripper_sensors")}
    if measurement_initialize and hasattr(filter_model, "measurement_initialize_beliefs"):
        filter_model.measurement_initialize_beliefs({k: v[0] for k, v in obs.items()})
    else:
        noise = noise if noise is not None else default_noise(filter_model)
        tril = initial_scale_tril if initial_scale_tril is not None else torch.linalg.cholesky(initial_covariance.to(torch.float32))
        mean = states[0] + noise.gaussian((N, d), like=states) @ tril.t()
        filter_model.initialize_beliefs(mean=mean, covariance=initial_covariance[None].expand(N, d, d))
    pred = filter_model.forward_loop(observations={k: v[1:] for k, v in obs.items()}, controls=batch["controls"][1:])
    return torch.mean((pred - states[1:]) ** 2)


def train_filter_step(filter_model, batch, optimizer: torch.optim.Optimizer, *, initial_covariance: torch.Tensor,
                      noise: Optional[NoiseSource] = None, measurement_initialize: bool = False,
                      all_reduce: bool = False) -> float:
    """One optimisation step on one batch of subsequences; returns the loss."""
    optimizer.zero_grad(set_to_none=True)
    loss = filter_loss(filter_model, batch, initial_covariance=initial_covariance, noise=noise,
                       measurement_initialize=measurement_initialize)
    loss.backward()
    if all_reduce:
        distributed.all_reduce_gradients(filter_model)
    optimizer.step()
    return float(loss.detach())


class GraphedFilterStep:
    """``train_filter_step`` captured ONCE as a hipGraph and replayed: the reference-sized training step (32 trajectories x
    30 particles x 16 steps, ``train_helpers.py:124-162``) is ~430 kernel launches of a few microseconds each, and the host
    -- Python, autograd, ~8 us between launches -- not the GPU bounds it (GPU busy 64-69 %, profiles/r06/
    train_refsize_fwd_ab.txt).  A replay is one launch of the whole step: same kernels, same order, same arithmetic.

        step = GraphedFilterStep(filter_model, optimizer, initial_covariance=cov)
        for batch in batches:           # every batch of the SAME shapes (the reference's fixed batch / subsequence sizes)
            loss = step(batch)

    The first ``eager_steps`` calls run ``train_filter_step`` eagerly (they build the packed blobs, per-trajectory programs
    and allocator blocks a capture must not create); the next call captures forward + backward + ``optimizer.step()`` on static
    copies of the batch and replays it; later calls copy the batch in and replay.  Randomness: the explicit ``NoiseSource``
    generators are registered with the graph, so every replay draws fresh numbers, in the order the eager step would.  The
    checks that need a host read (range flag, "covariance not positive definite") accumulate on the device during a replay and are
    read once after it.  Requirements: a capturable optimiser (``torch.optim.SGD``, or Adam with ``capturable=True``), no
    host-dependent control flow in user models, batches of one shape (a new shape raises)."""

    def __init__(self, filter_model, optimizer: torch.optim.Optimizer, *, initial_covariance: torch.Tensor,
                 noise: Optional[NoiseSource] = None, measurement_initialize: bool = False, all_reduce: bool = False,
                 eager_steps: int = 2):
        assert not all_reduce, "the gradient all-reduce is not captured: use train_filter_step for data-parallel training"
        self.model, self.optimizer, self.cov = filter_model, optimizer, initial_covariance
        self.noise = noise if noise is not None else default_noise(filter_model)
        self.measurement_initialize, self.eager_left = measurement_initialize, int(eager_steps)
        self.graph, self.static, self.loss, self.tril = None, None, None, None
        self.stream = None   # eager steps and the capture share ONE side stream: autograd remembers the stream a parameter's
                             # gradient accumulator was created on, and a capture must not reach back to the default stream

    def _capture(self, batch):
        dev = batch["states"].device
        self.static = {k: v.detach().clone() for k, v in batch.items()}
        self.tril = torch.linalg.cholesky(self.cov.to(torch.float32))   # outside the capture: the factorisation reads its status on the host
        self.graph = torch.cuda.CUDAGraph()
        sources = {id(s): s for s in (self.noise, getattr(self.model, "noise", None)) if type(s) is NoiseSource}
        for src in sources.values():
            self.graph.register_generator_state(src._gen(dev))
        self.optimizer.zero_grad(set_to_none=True)
        engine.clear_range(dev)
        engine.CAPTURING = True
        try:
            with torch.cuda.graph(self.graph, stream=self.stream):
                loss = filter_loss(self.model, self.static, initial_covariance=self.cov, noise=self.noise,
                                   measurement_initialize=self.measurement_initialize, initial_scale_tril=self.tril)
                loss.backward()
                self.optimizer.step()
                self.loss = loss.detach()
        finally:
            engine.CAPTURING = False

    def __call__(self, batch: Dict[str, torch.Tensor]) -> float:
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=batch["states"].device)
        if self.eager_left > 0:
            self.eager_left -= 1
            self.stream.wait_stream(torch.cuda.current_stream(batch["states"].device))
            with torch.cuda.stream(self.stream):
                loss = train_filter_step(self.model, batch, self.optimizer, initial_covariance=self.cov, noise=self.noise,
                                         measurement_initialize=self.measurement_initialize)   # (reads the loss: synchronises)
            return loss
        if self.graph is None:
            self._capture(batch)
        else:
            for k, v in self.static.items():
                assert batch[k].shape == v.shape, f"GraphedFilterStep was captured for {k} of shape {tuple(v.shape)}, got {tuple(batch[k].shape)}"
                v.copy_(batch[k])
        self.graph.replay()
        engine.check_range(batch["states"].device)   # the replay's deferred checks: one 4-byte read
        return float(self.loss)


# ------------------------------------------------------------------------------ pre-training losses
def _check_training(module):
    assert module.training, "call .train() first"
    assert engine.TRAINING_BACKEND is not None, "select engine.set_training_backend('hip' | 'autograd')"


def dynamics_single_step_loss(dynamics_model, batch: Dict[str, torch.Tensor], *, loss_function: str = "mse") -> torch.Tensor:
    """``mse(f(x_t, u_{t+1}), x_{t+1})`` on a ``data.SingleStepBatcher`` batch (``"nll"``: negative
    log-likelihood under the model's noise); ``train_helpers.py:31-48`` uses ``"mse"``."""
    _check_training(dynamics_model)
    pred, tril = dynamics_model(initial_states=batch["initial_states"], controls=batch["controls"])
    if loss_function == "mse":
        return torch.nn.functional.mse_loss(pred, batch["next_states"])
    assert loss_function == "nll", loss_function
    dist = torch.distributions.MultivariateNormal(loc=pred, scale_tril=tril)
    return -torch.mean(dist.log_prob(batch["next_states"]))


def dynamics_recurrent_loss(dynamics_model, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
    """Open-loop rollout from ``states[0]`` over ``controls[1:]``, mse against ``states[1:]``
    (time-major ``data.SubsequenceBatcher`` batch; ``train_helpers.py:51-74``)."""
    _check_training(dynamics_model)
    pred, _ = dynamics_model.forward_loop(initial_states=batch["states"][0], controls=batch["controls"][1:])
    return torch.nn.functional.mse_loss(pred, batch["states"][1:])


def particle_filter_measurement_loss(measurement_model, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
    """``mse(loglik(noisy_state, observation), log N(noisy_state; state, covariance))`` with one
    particle per sample (``data.ParticleFilterMeasurementBatcher`` batch; ``train_helpers.py:76-96``).
    With the ``"hip"`` backend the per-particle network runs forward and backward in K6."""
    _check_training(measurement_model)
    obs = {k: batch[k] for k in ("image", "gripper_pos", "gripper_sensors")}
    pred = measurement_model(states=batch["noisy_states"][:, None, :].contiguous(), observations=obs)
    assert pred.shape == (batch["noisy_states"].shape[0], 1)
    return torch.nn.functional.mse_loss(pred, batch["log_likelihoods"][:, None])


def virtual_sensor_loss(virtual_sensor_model, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
    """``mse(z(o_{t+1}), x_{t+1})`` on a ``data.SingleStepBatcher`` batch (``train_helpers.py:98-121``)."""
    _check_training(virtual_sensor_model)
    obs = {k: batch[k] for k in ("image", "gripper_pos", "gripper_sensors")}
    z, _tril = virtual_sensor_model(observations=obs)
    return torch.nn.functional.mse_loss(z, batch["next_states"])


def pretrain_step(loss_fn, module, batch, optimizer: torch.optim.Optimizer, *, all_reduce: bool = False, **kw) -> float:
    """One optimisation step of a pre-training loss (``loss_fn(module, batch, **kw)``)."""
    optimizer.zero_grad(set_to_none=True)
    loss = loss_fn(module, batch, **kw)
    loss.backward()
    if all_reduce:
        distributed.all_reduce_gradients(module)
    optimizer.step()
    return float(loss.detach())
