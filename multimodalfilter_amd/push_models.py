"""Push-task models (``state_dim = 2``) under the reference's class names
(``/root/reference/crossmodal/push_models/__init__.py:5-21``; LSTM baseline out of scope)."""
from . import task_models as _tm

_ns = _tm.make_task_models(_tm.PUSH)
model_types = _ns.model_types

PushDynamicsModel = _ns.PushDynamicsModel
PushMeasurementModel = _ns.PushMeasurementModel
PushCrossmodalWeightModel = _ns.PushCrossmodalWeightModel
PushVirtualSensorModel = _ns.PushVirtualSensorModel
PushCrossmodalKalmanFilterWeightModel = _ns.PushCrossmodalKalmanFilterWeightModel
PushParticleFilter = _ns.PushParticleFilter
PushCrossmodalParticleFilter = _ns.PushCrossmodalParticleFilter
PushCrossmodalParticleFilterSeq5 = _ns.PushCrossmodalParticleFilterSeq5
PushUnimodalParticleFilter = _ns.PushUnimodalParticleFilter
PushKalmanFilter = _ns.PushKalmanFilter
PushCrossmodalKalmanFilter = _ns.PushCrossmodalKalmanFilter
PushUnimodalKalmanFilter = _ns.PushUnimodalKalmanFilter
PushMeasurementCrossmodalKalmanFilter = _ns.PushMeasurementCrossmodalKalmanFilter
PushMeasurementUnimodalKalmanFilter = _ns.PushMeasurementUnimodalKalmanFilter
