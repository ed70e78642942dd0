import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multimodalfilter_amd as mmf
from multimodalfilter_amd import engine
from oracle import models as om
dev = torch.device("cuda:0")
for scale in (1.0, 1e3, 1e5, 1e7):
    dyn = mmf.door_models.DoorDynamicsModelBrent()
    dyn.load_state_dict(om.seeded_state_dict(dyn, seed=0, gain=1.4))
    with torch.no_grad(): dyn.state_layers[0].weight.mul_(scale)
    dyn.to(dev)
    x = torch.randn((2, 64, 3), device=dev)
    ctx = dyn.encode_controls(torch.randn((2, 7), device=dev))
    for prec in ("f16x3", "f32"):
        engine.set_default_precision(prec)
        engine.range_flag(dev).zero_()
        out = dyn.propagate_encoded(x, ctx, None)
        torch.cuda.synchronize()
        print(scale, prec, "flag", int(engine.range_flag(dev).item()), "out absmax", float(out.abs().max()), "nan", bool(torch.isnan(out).any()))
