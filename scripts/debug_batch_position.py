"""Do K4 / K7 results depend on the row's position in the batch?"""
import torch
import multimodalfilter_amd as mmf
from multimodalfilter_amd import engine, layers
from multimodalfilter_amd.trajprog import TrajProgram

dev = torch.device("cuda:0")
torch.manual_seed(0)
enc = layers.image_encoder(64).to(dev)
imgs = (torch.randn(15, 32, 32, device=dev) * 0.5).clamp(-1, 1)
big = engine.encode_images([enc], imgs)[0]
for t in range(3):
    small = engine.encode_images([enc], imgs[5 * t:5 * t + 5].contiguous())[0]
    print("K4 t", t, (small - big[5 * t:5 * t + 5]).abs().max().item())

ve = layers.vector_encoder(7, 64).to(dev)
x = torch.randn(15, 7, device=dev)
p = TrajProgram()
s = p.load("x", 7)
h = p.vector_encoder(ve, s, 7)
p.store("y", h, 64)
def run(xx):
    y = torch.empty((xx.shape[0], 64), device=dev)
    p.run({"x": xx.contiguous(), "y": y}, xx.shape[0])
    return y
big = run(x)
for t in range(3):
    print("K7 t", t, (run(x[5 * t:5 * t + 5]) - big[5 * t:5 * t + 5]).abs().max().item())
for r in range(15):
    print("K7 single row", r, (run(x[r:r + 1]) - big[r:r + 1]).abs().max().item())
