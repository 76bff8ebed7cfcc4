"""TensorFlow-free model conversion: TF checkpoint bundle -> trt_weights.bin."""
