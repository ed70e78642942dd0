"""``torchfilter.train`` restated as LOSS functions (``crossmodal/train_helpers.py:45,71,93,116,155``
call the loops; an optimiser step around a loss is all a loop adds).  TEST INFRASTRUCTURE.

Published behaviour of the absent, un-pinned package -- PARITY UNPINNED (see ``data.py``):

* ``train_dynamics_single_step(..., loss_function="mse")``: ``mse(f(x_t, u_{t+1}), x_{t+1})``
  (``"nll"``: negative log-likelihood under the model's ``scale_tril``).
* ``train_dynamics_recurrent``: open-loop rollout from ``x_0`` over ``u_{1:}``, mse against ``x_{1:}``.
* ``train_particle_filter_measurement``: the measurement model evaluated on ONE particle per
  sample, ``mse(loglik(noisy_state, observation), log N(noisy_state; state, covariance))``.
* ``train_virtual_sensor``: ``mse(z(o_{t+1}), x_{t+1})`` (the predicted noise is not trained here).
* ``train_filter``: see ``multimodalfilter_amd/train.py::filter_loss`` / the training tests.
"""
import torch
import torch.nn.functional as F


def dynamics_single_step_loss(dynamics_model, *, initial_states, next_states, controls, loss_function="mse"):
    pred, tril = dynamics_model(initial_states=initial_states, controls=controls)
    if loss_function == "mse":
        return F.mse_loss(pred, next_states)
    assert loss_function == "nll"
    dist = torch.distributions.MultivariateNormal(loc=pred, scale_tril=tril)
    return -torch.mean(dist.log_prob(next_states))


def dynamics_recurrent_loss(dynamics_model, *, states, controls):
    """``states (T, N, d)``, ``controls (T, N, 7)`` (time-major)."""
    pred, _ = dynamics_model.forward_loop(initial_states=states[0], controls=controls[1:])
    return F.mse_loss(pred, states[1:])


def particle_filter_measurement_loss(measurement_model, *, noisy_states, observations, log_likelihoods):
    pred = measurement_model(states=noisy_states[:, None, :], observations=observations)
    assert pred.shape == (noisy_states.shape[0], 1)
    return F.mse_loss(pred, log_likelihoods[:, None])


def virtual_sensor_loss(virtual_sensor_model, *, observations, states):
    z, _tril = virtual_sensor_model(observations=observations)
    return F.mse_loss(z, states)


def _unavailable(*_a, **_k):
    raise RuntimeError("torchfilter.train's Buddy-driven loops are not restated; use the *_loss functions")


train_dynamics_single_step = train_dynamics_recurrent = _unavailable
train_particle_filter_measurement = train_virtual_sensor = train_filter = _unavailable
