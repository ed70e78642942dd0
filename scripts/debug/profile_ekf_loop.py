"""Which torch ops run in one EKF forward_loop (the copies rocprofv3 shows as __amd_rocclr_copyBuffer)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from multimodalfilter_amd import synthetic, evaluation
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
wl = dict(bench.WORKLOADS["door_ekf"])
K, B, d = 20, wl["batch"], 3
f = bench.build_filter(wl, dev)
synthetic.stabilise_dynamics(f)
_, traj = bench.make_inputs(wl, K, B, 1, dev, d)
for _ in range(3):
    evaluation.run_filter(f, traj)
torch.cuda.synchronize()
t0 = time.perf_counter(); evaluation.run_filter(f, traj); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0):.2f} ms, until GPU done {1e3*(t2-t0):.2f} ms ({1e3*(t2-t0)/K:.3f} ms/step)")
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    evaluation.run_filter(f, traj)
torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=4).table(sort_by="self_cpu_time_total", row_limit=25, max_name_column_width=40, max_src_column_width=90))
for ev in prof.key_averages(group_by_stack_n=8):
    if ev.key in ("aten::nonzero", "aten::_to_copy", "aten::_local_scalar_dense") :
        print(ev.key, ev.count, ev.self_cpu_time_total)
        for s in ev.stack[:8]:
            print("     ", s)
