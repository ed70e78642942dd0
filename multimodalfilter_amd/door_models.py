"""Door-task models (``state_dim = 3``) under the reference's class names
(``/root/reference/crossmodal/door_models/__init__.py:5-19``; the LSTM baseline is out of
scope)."""
from . import task_models as _tm

_ns = _tm.make_task_models(_tm.DOOR)
model_types = _ns.model_types

DoorDynamicsModel = _ns.DoorDynamicsModel
DoorDynamicsModelBrent = _ns.DoorDynamicsModelBrent
DoorMeasurementModel = _ns.DoorMeasurementModel
DoorCrossmodalWeightModel = _ns.DoorCrossmodalWeightModel
DoorVirtualSensorModel = _ns.DoorVirtualSensorModel
DoorCrossmodalKalmanFilterWeightModel = _ns.DoorCrossmodalKalmanFilterWeightModel
DoorParticleFilter = _ns.DoorParticleFilter
DoorCrossmodalParticleFilter = _ns.DoorCrossmodalParticleFilter
DoorCrossmodalParticleFilterSeq5 = _ns.DoorCrossmodalParticleFilterSeq5
DoorUnimodalParticleFilter = _ns.DoorUnimodalParticleFilter
DoorKalmanFilter = _ns.DoorKalmanFilter
DoorCrossmodalKalmanFilter = _ns.DoorCrossmodalKalmanFilter
DoorUnimodalKalmanFilter = _ns.DoorUnimodalKalmanFilter
DoorMeasurementCrossmodalKalmanFilter = _ns.DoorMeasurementCrossmodalKalmanFilter
DoorMeasurementUnimodalKalmanFilter = _ns.DoorMeasurementUnimodalKalmanFilter
