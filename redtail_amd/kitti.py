"""KITTI 2015 stereo evaluation helpers: the D1 outlier rate the reference's README quotes
(stereoDNN/README.md:26-36) and the benchmark's 16-bit PNG disparity encoding (value / 256, 0 = invalid), which is also
what sample_app/main.cpp:324-330 writes."""
import numpy as np


def d1_all(est, gt):
    """D1-all in percent: share of pixels with ground truth (gt > 0) whose disparity error is > 3 px AND > 5 % of
    the true disparity (KITTI 2015 devkit, evaluate_scene_flow.cpp)."""
    est, gt = np.asarray(est, np.float64), np.asarray(gt, np.float64)
    valid = gt > 0
    n = int(valid.sum())
    if n == 0:
        return float("nan")
    err = np.abs(est - gt)[valid]
    return 100.0 * float(((err > 3.0) & (err > 0.05 * gt[valid])).sum()) / n


def read_disparity_png(path):
    """KITTI disparity map: uint16 PNG, disparity = value / 256, 0 marks pixels without ground truth"""
    from PIL import Image
    a = np.asarray(Image.open(path))
    if a.dtype != np.uint16 and a.dtype != np.int32:
        raise ValueError("%s is not a 16-bit disparity PNG" % path)
    return a.astype(np.float32) / 256.0


def write_disparity_png(path, disp_px):
    """inverse of read_disparity_png; same rounding and saturation as rt_disparity_to_u16 (cvRound, saturate_cast)"""
    from PIL import Image
    v = np.clip(np.rint(np.asarray(disp_px, np.float32) * np.float32(256)), 0, 65535).astype(np.uint16)
    Image.fromarray(v).save(path)


def read_image_bgr(path):
    """8-bit colour image as the HWC BGR array cv::imread would return (input of rt_preprocess_bgr8)"""
    from PIL import Image
    return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])
