"""Is the first forward_loop at a new length slower (allocator growth inside the timed region)?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from multimodalfilter_amd import synthetic, evaluation

dev = torch.device("cuda:0")
wl = dict(bench.WORKLOADS["door_ekf"])
B, d = wl["batch"], 3
f = bench.build_filter(wl, dev)
synthetic.stabilise_dynamics(f)
_, t5 = bench.make_inputs(wl, 5, B, 1, dev, d)
_, t20 = bench.make_inputs(wl, 20, B, 2, dev, d)
def run(tr, tag):
    torch.cuda.synchronize(); t0 = time.perf_counter(); evaluation.run_filter(f, tr); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(tag, f"{1e3*dt:.2f} ms  ({1e3*dt/(tr['states'].shape[0]-1):.3f} ms/step)  reserved {torch.cuda.memory_reserved()/2**20:.0f} MiB")
run(t5, "T=5 first")
run(t20, "T=20 first")
run(t20, "T=20 again")
run(t20, "T=20 again")
