"""Training-step throughput (SURVEY.md 8d config C5 shape, scaled to one launch's memory):
push unimodal particle filter, train mode (no resampling), M particles, subsequences of L steps,
forward + backward + SGD step.  Compares the two training backends:

    python scripts/bench_train.py [--particles 8192] [--batch 32] [--length 16] [--steps 3]

Prints one JSON line per backend: particle-steps/s counts N * M * (L - 1) per optimisation step.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--particles", type=int, default=8192)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--length", type=int, default=16)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--model", default="PushUnimodalParticleFilter")
    ap.add_argument("--backends", default="hip,autograd")
    ap.add_argument("--cnn-precision", default=None, choices=[None, "f32", "f16x3", "bf16"],
                    help="image-encoder training forward (hip backend): default exact fp32; bf16 = BASELINE config 5")
    args = ap.parse_args()

    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import engine, synthetic, train

    dev = torch.device("cuda:0")
    task = "push" if args.model.startswith("Push") else "door"
    d = 2 if task == "push" else 3
    N, M, L = args.batch, args.particles, args.length
    batch = {k: v.to(dev) for k, v in synthetic.make_trajectories(state_dim=d, T=L - 1, N=N, seed=11).items()}
    cov = torch.eye(d, device=dev) * 0.1
    for backend in args.backends.split(","):
        torch.manual_seed(0)
        f = mmf.model_types(task)[args.model]().to(dev).train()
        f.num_particles = M
        engine.set_training_backend(backend)
        engine.set_image_encoder_precision(args.cnn_precision if backend == "hip" else None)
        opt = torch.optim.SGD(f.parameters(), lr=1e-4)
        f.noise = mmf.NoiseSource(seed=5)
        times, losses = [], []
        torch.cuda.reset_peak_memory_stats()
        for it in range(args.steps + 1):  # first iteration is the warm-up
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            losses.append(train.train_filter_step(f, batch, opt, initial_covariance=cov, noise=f.noise))
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        best = min(times[1:])
        print(json.dumps({"backend": backend, "cnn_forward": (args.cnn_precision or "f32") if backend == "hip" else "torch",
                          "model": args.model, "batch": N, "particles": M, "length": L,
                          "ms_per_train_step": 1e3 * best,
                          "particle_steps_per_s_fwd_bwd": N * M * (L - 1) / best,
                          "peak_memory_GB": torch.cuda.max_memory_allocated() / 2 ** 30,
                          "loss_first_last": [losses[0], losses[-1]]}), flush=True)
        engine.set_training_backend(None)
        engine.set_image_encoder_precision(None)
        del f, opt
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
