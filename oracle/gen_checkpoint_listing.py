"""Generates tests/golden/shipped_checkpoint_keys.json: key -> shape / dtype of the `best_net_ray_marching.pth` files the
reference ships under pointnerf/mvsnet_checkpoints/init/ (data, not source; runs only in the build container).

TEST INFRASTRUCTURE.  Pins `PointNerf.AGGREGATOR_MAP` (pointnerf2studio_amd/model.py): the legacy module names and
layer shapes the opt-in warm start (`hip_load_aggregator_weights`) expects are the ones the shipped files hold.
The shipped files are MVSNet-initialisation checkpoints: they carry `aggregator.*` (+ `net_fine_decoder.*`) and no
`neural_points.*` tensors -- those appear in the per-scene `{iter}_net_ray_marching.pth` that training writes
(models/base_model.py:85-120), whose layout tests/test_gpu_checkpoint.py reproduces.

  python oracle/gen_checkpoint_listing.py
"""
import glob
import json
import os

import torch

REF = "/root/reference/pointnerf/mvsnet_checkpoints/init"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "shipped_checkpoint_keys.json")


def main():
    listing = {}
    for path in sorted(glob.glob(os.path.join(REF, "*", "best_net_ray_marching.pth"))):
        sd = torch.load(path, map_location="cpu", weights_only=True)
        name = os.path.basename(os.path.dirname(path))
        listing[name] = {k: {"shape": list(v.shape), "dtype": str(v.dtype).replace("torch.", "")} for k, v in sd.items()}
    with open(OUT, "w") as f:
        json.dump({"source": "pointnerf/mvsnet_checkpoints/init/<name>/best_net_ray_marching.pth", "checkpoints": listing},
                  f, indent=1, sort_keys=True)
    print("wrote", os.path.abspath(OUT), {k: len(v) for k, v in listing.items()})


if __name__ == "__main__":
    main()
