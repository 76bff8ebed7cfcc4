"""Where the reference's trained weight files live inside this repository.

The reference ships `stereoDNN/models/<Model>/TensorRT/trt_weights{,_fp16}.bin` (written by
scripts/tensorrt_model_builder.py:52-60, read by sample_app/main.cpp:111-134).  `/root/reference` does not exist
on the GPU box, so `stage_reference_weights()` (called from `__graft_entry__.build()`) copies the files into
`weights/_ref/` -- git-ignored like every other built artefact, but shipped with the gpurun snapshot -- and tests /
bench.py look them up through `weight_file()`, which raises when a file is missing: there is no silent fall-back to
synthetic weights anywhere a test or a benchmark says "real weights".
"""
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WEIGHT_DIR = os.path.join(ROOT, "weights", "_ref")

# (model id of include/rt_stereo_net.h, fp16 file?) -> path below stereoDNN/models
FILES = {
    ("resnet18_2D", False): "ResNet-18_2D/TensorRT/trt_weights.bin",
    ("resnet18_2D", True): "ResNet-18_2D/TensorRT/trt_weights_fp16.bin",
    ("nvtiny", False): "NVTiny/TensorRT/trt_weights.bin",
    ("nvtiny", True): "NVTiny/TensorRT/trt_weights_fp16.bin",
    ("nvsmall", True): "NVSmall/TensorRT/trt_weights_fp16.bin",     # the reference ships no fp32 file
}


def stage_reference_weights(reference=None):
    """Copies the weight files from the reference tree into weights/_ref/.  Returns the list of staged paths;
    a no-op (empty list) when the reference tree is absent (GPU box)."""
    reference = reference or os.environ.get("RT_REFERENCE", "/root/reference")
    src_root = os.path.join(reference, "stereoDNN", "models")
    staged = []
    if not os.path.isdir(src_root):
        return staged
    todo = [(os.path.join(src_root, rel), os.path.join(WEIGHT_DIR, rel)) for rel in sorted(set(FILES.values()))]
    # the sample application's stereo pair (sample_app/data/img_{left,right}.png)
    todo += [(os.path.join(reference, "stereoDNN", "sample_app", "data", f), os.path.join(WEIGHT_DIR, "sample", f))
             for f in ("img_left.png", "img_right.png", "img_left.bin", "img_right.bin")]
    # the golden tensors of the reference's plugin tests (stereoDNN/tests/data/*.bin), for oracle/_ref/nvstereo_tests
    data = os.path.join(reference, "stereoDNN", "tests", "data")
    if os.path.isdir(data):
        todo += [(os.path.join(data, f), os.path.join(WEIGHT_DIR, "tests_data", f)) for f in sorted(os.listdir(data)) if f.endswith(".bin")]
    for src, dst in todo:
        if not os.path.exists(src):
            continue
        if not os.path.exists(dst) or os.path.getsize(dst) != os.path.getsize(src):
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            shutil.copyfile(src, dst)
        staged.append(dst)
    return staged


def sample_image(side):
    """the reference sample application's left / right PNG, staged next to the weights"""
    path = os.path.join(WEIGHT_DIR, "sample", "img_%s.png" % side)
    if not os.path.exists(path):
        raise FileNotFoundError("%s is missing: run `python __graft_entry__.py` where /root/reference exists" % path)
    return path


def sample_bin(side):
    """the reference's own pre-processed network input for that image (sample_app/data/img_<side>.bin: 3 x 321 x 1025 float32), or None"""
    path = os.path.join(WEIGHT_DIR, "sample", "img_%s.bin" % side)
    return path if os.path.exists(path) else None


def tests_data_dir():
    """stereoDNN/tests/data as staged next to the weights"""
    path = os.path.join(WEIGHT_DIR, "tests_data")
    if not os.path.isdir(path):
        raise FileNotFoundError("%s is missing: run `python __graft_entry__.py` where /root/reference exists" % path)
    return path


def weight_file(model, fp16=False):
    """Path of the staged reference weight file; FileNotFoundError (never a fall-back) when it is not there."""
    key = (model, bool(fp16))
    if key not in FILES:
        raise FileNotFoundError("the reference ships no %s weight file for %s" % ("fp16" if fp16 else "fp32", model))
    path = os.path.join(WEIGHT_DIR, FILES[key])
    if not os.path.exists(path):
        raise FileNotFoundError("%s is missing: run `python __graft_entry__.py` where /root/reference exists "
                                "(it stages the reference's weight files; they travel with the gpurun snapshot)" % path)
    return path
