/*
 * rt_stereo_net.h -- whole-network C ABI of the Stereo DNN inference path on MI355X.
 *
 * What sample_app/main.cpp:136-340 and ros/packages/stereo_dnn_ros/src/stereo_dnn_ros_node.cpp:224-377 of
 * the reference do with TensorRT objects (read trt_weights.bin, build the network through the
 * IPluginContainer API, build an engine, create a context, execute on device buffers) behind five
 * plain-C calls, so that non-C++ hosts (ctypes in bench.py / tests, cgo, JNI ...) can drive the same
 * code path.  Implemented by redtail_amd/csrc/host/net_capi.cpp on top of the NvInfer.h shim.
 */
#ifndef RT_STEREO_NET_H
#define RT_STEREO_NET_H

#include <stddef.h>
#include <stdint.h>

#include "rt_stereo.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rtStereoNet rtStereoNet;

/* model_type strings of sample_app/main.cpp:155-161 */
enum { RT_MODEL_RESNET18_2D = 0, RT_MODEL_NVSMALL = 1, RT_MODEL_NVTINY = 2, RT_MODEL_RESNET18 = 3 };

/* Build an engine for `width` x `height` images (any size = 1 mod 8 for ResNet-18 2D).
 * weights_dtype: RT_F32 / RT_F16 = element type of the weight blob (trt_weights.bin / trt_weights_fp16.bin,
 * layout of scripts/tensorrt_model_builder.py:52-60).  Activations are fp32.  max_disp <= 0 selects the
 * model's default half-resolution disparity range (48 / 48 / 24 / 68). */
int rt_net_create(rtStereoNet** net, int model, int width, int height, int max_batch, int weights_dtype,
                  int max_disp, const char* weights_path);
/* Same, from an in-memory image of the weight file (what rank 0 broadcasts over RCCL). */
int rt_net_create_from_memory(rtStereoNet** net, int model, int width, int height, int max_batch, int weights_dtype,
                              int max_disp, const void* blob, size_t bytes);

/* Multi-GPU start-up without Python: every rank of `comm` (rt_stereo.h: rt_comm_init_rank / rt_comm_adopt / rt_comm_init_all)
 * calls this on its own device; rank `root` passes the weight-file image, the others pass NULL / 0 and receive it over RCCL
 * (ncclBroadcast: first its size, then its bytes), then every rank builds the same engine.  What redtail_amd/parallel.py does
 * through torch.distributed, for hosts that have no torch (apps/stereo_throughput.cpp).  comm == NULL: world of one rank. */
int rt_net_create_broadcast(rtStereoNet** net, int model, int width, int height, int max_batch, int weights_dtype,
                            int max_disp, const void* blob, size_t bytes, rtComm* comm, int root);
/* crc32 (zlib polynomial) of the weight-file image the engine was built from: every rank prints it, they must agree */
int rt_net_weights_crc32(const rtStereoNet* net, uint32_t* crc);
/* The image itself (owned by the engine, valid until rt_net_destroy): a rank that received it by broadcast builds its further
 * execution contexts from it (rt_net_create_from_memory). */
int rt_net_weights_image(const rtStereoNet* net, const void** data, size_t* bytes);

/* Everything above in one call, options as a struct (zero-initialise, then set what is needed): weights from `weights_path` or from the
 * image `blob` / `bytes` (the root rank's when `comm` is set, see rt_net_create_broadcast); flags: RT_CONV_EXACT_FP32 keeps every 2-D
 * convolution on the fp32 fmaf-chain kernels (IBuilder::setExactFp32Mode). */
typedef struct rtNetOptions {
    int model, width, height, max_batch;
    int weights_dtype;         /* RT_F32 / RT_F16 */
    int max_disp;              /* <= 0: the model's default */
    const char* weights_path;  /* or NULL */
    const void* blob;          /* or NULL */
    size_t bytes;
    unsigned flags;            /* RT_CONV_EXACT_FP32 */
    rtComm* comm;              /* or NULL: no broadcast */
    int root;
} rtNetOptions;
int rt_net_create_opt(rtStereoNet** net, const rtNetOptions* options);

/* left/right: device (N,3,H,W) fp32 in [0,1]; disp: device (N,1,H,W) fp32 (ResNet-18 2D: disparity / width;
 * 3-D models: pixels).  stream == NULL: synchronous (IExecutionContext::execute); otherwise asynchronous on
 * that HIP stream (IExecutionContext::enqueue). */
int rt_net_execute(rtStereoNet* net, const void* left, const void* right, void* disp, int batch, rtStream stream);

/* Per-launch timing through nvinfer1::IProfiler (single stream, one event pair per launch):
 * writes "name<TAB>milliseconds\n" lines into buf.  Returns 0 or an error. */
int rt_net_profile(rtStereoNet* net, const void* left, const void* right, void* disp, int batch, char* buf,
                   size_t buf_bytes);

/* Engine plan: ICudaEngine::serialize() (sample_app/main.cpp:269-275).  buf == NULL queries the size.  Fails for
 * networks with non-serialisable plugins (the 3-D models), as the reference does. */
int rt_net_serialize(rtStereoNet* net, void* buf, size_t buf_bytes, size_t* plan_bytes);
/* IRuntime::deserializeCudaEngine(plan, size, &StereoDnnPluginFactory) + createExecutionContext
 * (sample_app/main.cpp:198-220; lib/internal_utils.cpp:289-313). */
int rt_net_create_from_plan(rtStereoNet** net, const void* plan, size_t plan_bytes);
int rt_net_num_layers(const rtStereoNet* net);     /* layers of the network definition  */
/* HIP streams the context issues its launches on: 2 (default) = the right-image encoder on a second stream, best for one
 * context (latency); 1 = everything on the caller's stream, best when several handles are kept busy side by side
 * (IExecutionContext::setExecutionStreams, an extension of the NvInfer.h subset; bench.py uses 1 with its six contexts). */
int rt_net_set_streams(rtStereoNet* net, int streams);
/* Graph mode (IExecutionContext::setGraphMode, an extension): the second rt_net_execute with the same device pointers, batch and stream
 * is captured as a hipGraph (rt_stereo.h: rt_graph_*), every further one is a single graph launch; other pointers capture another graph.
 * Off by default (the host is not the bottleneck of this path). */
int rt_net_set_graph(rtStereoNet* net, int on);
/* Debug mode (IExecutionContext::setDebugSync): every launch is synchronised and the input of every fp16-pipe convolution is range-checked
 * first (rt_check_range): an execute() whose activations leave the fp16-split domain fails with the layer's name in rt_net_last_error(). */
int rt_net_set_debug(rtStereoNet* net, int on);
/* Launch trace (IExecutionContext::setLaunchTrace, a debugging aid): while on, the output of every launch is hashed on the launch's own
 * stream (rt_stereo.h: rt_hash_buffer).  rt_net_read_launch_trace waits for the last pass and returns one value per launch (count, or -1);
 * two passes over the same input must agree launch by launch -- the first index that differs names the kernel that deviated.
 * rt_net_launch_name: the launch's layer name; rt_net_read_launch_output: its output tensor as stored (host == NULL: size in bytes). */
int rt_net_set_launch_trace(rtStereoNet* net, int on);
int rt_net_read_launch_trace(rtStereoNet* net, unsigned long long* hashes, int max);
const char* rt_net_launch_name(const rtStereoNet* net, int launch);
long long rt_net_read_launch_output(rtStereoNet* net, int launch, void* host, long long bytes);
int rt_net_num_launches(const rtStereoNet* net);   /* kernel launches after fusion       */
int rt_net_destroy(rtStereoNet* net);
const char* rt_net_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
