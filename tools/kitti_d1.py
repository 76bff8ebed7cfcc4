#!/usr/bin/env python3
"""D1 error of a Stereo DNN model on the KITTI 2015 stereo training set (200 pairs), end to end on the GPU:
PNG -> rt_preprocess_bgr8 (INTER_AREA to the network size) -> network -> disparity in pixels of the original image.

    python tools/kitti_d1.py <kitti>/training  <trt_weights.bin>  [--model resnet18_2D] [--width 1025 --height 321]

The reference quotes 9.8 % (ResNet-18 2D), 7.7 % (NVSmall), 11.1 % (NVTiny) in stereoDNN/README.md:26-36.
Needs the dataset (not redistributable) and an MI355X; nothing of this runs in the test suites."""
import argparse
import glob
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redtail_amd import capi, kitti  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("weights")
    ap.add_argument("--model", default="resnet18_2D")
    ap.add_argument("--width", type=int, default=1025)
    ap.add_argument("--height", type=int, default=321)
    args = ap.parse_args()
    lib = capi.NetLib()
    net = lib.create(args.model, args.width, args.height, weights_path=args.weights)
    lefts = sorted(glob.glob(os.path.join(args.root, "image_2", "*_10.png")))
    if not lefts:
        sys.exit("no image_2/*_10.png under %s" % args.root)
    k = lib.kernels
    d_l = torch.empty(1, 3, args.height, args.width, device="cuda")
    d_r = torch.empty_like(d_l)
    disp = torch.empty(1, 1, args.height, args.width, device="cuda")
    scores = []
    for lp in lefts:
        name = os.path.basename(lp)
        l8 = torch.from_numpy(kitti.read_image_bgr(lp)).cuda()
        r8 = torch.from_numpy(kitti.read_image_bgr(os.path.join(args.root, "image_3", name))).cuda()
        h, w = l8.shape[:2]
        k.preprocess_bgr8(l8, h, w, d_l, args.height, args.width)
        k.preprocess_bgr8(r8, h, w, d_r, args.height, args.width)
        net.execute(d_l, d_r, disp, 1)
        torch.cuda.synchronize()
        d = disp[0, 0]
        if args.model == "resnet18_2D":
            d = d * args.width                         # sigmoid output = disparity / width
        # back to the geometry of the original image: disparities scale with the width ratio
        d = torch.nn.functional.interpolate(d[None, None] * (w / args.width), size=(h, w), mode="bilinear", align_corners=False)[0, 0]
        gt = kitti.read_disparity_png(os.path.join(args.root, "disp_occ_0", name))
        scores.append(kitti.d1_all(d.cpu().numpy(), gt))
        print("%s  D1 %.2f %%" % (name, scores[-1]), flush=True)
    print("mean D1-all over %d pairs: %.2f %%" % (len(scores), float(np.nanmean(scores))))


if __name__ == "__main__":
    main()
