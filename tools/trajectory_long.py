"""GPU box: the N-step comparison of tests/test_gpu_training_trajectory.py over a LONGER horizon (default 300 Adam steps):
the plugin's training loop on the HIP path next to torch autograd over the CPU oracle, same data, same jitter uniforms.
Per-step losses drift apart (the trajectories are chaotic at the 1e-3 level after ~50 steps, see the test's self-sensitivity
measurement); what is compared here is where the two runs END: PSNR of their eval images against the teacher's.
    python tools/trajectory_long.py [--steps 300]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import pnr_oracle as O  # noqa: E402
import trajectory as T  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    args = ap.parse_args()
    O.build_c_oracle()
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    dev = torch.device("cuda:0")
    prob = T.make_problem(O)
    ps0 = [T.psnr(a, v["target"]) for a, v in zip(T.eval_images(O, prob, prob["points"], prob["weights"]), prob["views"])]
    t0 = time.time()
    losses_h, seeds, model = T.run_hip(prob, args.steps, dev)
    t_hip = time.time() - t0
    t0 = time.time()
    losses_o, pts_o, w_o = T.run_oracle(O, prob, args.steps, seeds)
    t_cpu = time.time() - t0
    ps_h = [T.psnr(a, v["target"]) for a, v in zip(T.hip_eval_images(model, prob, dev), prob["views"])]
    ps_o = [T.psnr(a, v["target"]) for a, v in zip(T.eval_images(O, prob, pts_o, w_o), prob["views"])]
    drift = [abs(a - b) / max(abs(b), 1e-12) for a, b in zip(losses_h, losses_o)]
    marks = [m for m in (10, 50, 100, 200, 300, 500, 1000) if m <= args.steps]
    print(json.dumps({"steps": args.steps, "psnr_start_db": ps0, "psnr_hip_db": ps_h, "psnr_oracle_db": ps_o,
                      "psnr_difference_db": [a - b for a, b in zip(ps_h, ps_o)],
                      "max_loss_drift_up_to_step": {str(m): max(drift[:m]) for m in marks},
                      "loss_hip_last10": sum(losses_h[-10:]) / 10, "loss_oracle_last10": sum(losses_o[-10:]) / 10,
                      "seconds_hip": t_hip, "seconds_oracle_cpu": t_cpu}))


if __name__ == "__main__":
    main()
