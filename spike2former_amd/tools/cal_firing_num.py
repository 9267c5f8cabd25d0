"""Firing-count tool: counterpart of the reference's tools/cal_firing_num.py (:138-174 hook, :203-225 loop, :272-285 output).

Runs `test_num` images through `model.predict` in eval mode WITHOUT resetting the membranes between images (exactly the
reference's loop, so neurons carry state across images), accumulates mean(output * quant) / test_num per Q_IFNode name
from the counters the neuron kernels maintain, prints the JSON `{"t0": {name: rate}}` and writes `fr_rate.csv`
(index = module name, column `T`).  Datasets / checkpoints are out of scope here: images are synthetic unless
`--images` points at a `.npy` array [n, 3, H, W]; `--checkpoint` loads a reference `.pth` (same state_dict keys).

    python -m spike2former_amd.tools.cal_firing_num --workload C2 --test-num 4 --out-dir /tmp/firing
"""
import argparse
import json
import os

import torch

import spike2former_amd as s2f
from spike2former_amd.init_utils import seeded_init


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C2", choices=sorted(s2f.WORKLOADS))
    ap.add_argument("--test-num", type=int, default=4)
    ap.add_argument("--quant", type=int, default=8, help="quantisation step count, as the reference's --quant (:57-59)")
    ap.add_argument("--checkpoint", default=None)
    ap.add_argument("--images", default=None)
    ap.add_argument("--out-dir", default=".")
    ap.add_argument("--reset-between-images", action="store_true", help="NOT what the reference does; for comparison")
    args = ap.parse_args(argv)
    assert torch.cuda.is_available(), "the firing tool runs the HIP path; no GPU visible"
    w = s2f.WORKLOADS[args.workload]
    model = s2f.MODELS.build(s2f.model_cfg(args.workload))
    if args.checkpoint:
        sd = torch.load(args.checkpoint, map_location="cpu")
        model.load_state_dict(sd.get("state_dict", sd), strict=False)
    else:
        seeded_init(model)
    model.cuda().eval()
    s2f.reset_net(model)
    if args.images:
        import numpy as np
        imgs = torch.from_numpy(np.load(args.images)).float()
    else:
        g = torch.Generator().manual_seed(0)
        imgs = torch.randn(args.test_num, 3, w["H"], w["W"], generator=g)
    metas = [dict(img_shape=(w["H"], w["W"]), batch_input_shape=(w["H"], w["W"]))]
    with torch.no_grad(), s2f.FiringRecorder(model, quant=args.quant) as rec:
        for i in range(args.test_num):
            if args.reset_between_images:
                s2f.reset_net(model)
            model(imgs[i % len(imgs)][None].cuda(), metas, mode="logits")
            rec.collect()
    result = rec.result(args.test_num)
    print(json.dumps(result))
    os.makedirs(args.out_dir, exist_ok=True)
    rec.to_csv(os.path.join(args.out_dir, "fr_rate.csv"), args.test_num)
    with open(os.path.join(args.out_dir, "fr_rate.json"), "w") as f:
        json.dump(result, f, indent=1)
    return result


if __name__ == "__main__":
    main()
