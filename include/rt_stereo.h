/*
 * rt_stereo.h -- thin C ABI over the MI355X (gfx950) HIP kernels of the Stereo DNN hot path.
 *
 * This is the drop-in boundary "host C++ calls HIP through": every plugin `enqueue` in
 * redtail_amd/csrc/host/plugins.cpp and every TensorRT-native layer of the shim executor ends in
 * exactly one of these entry points.  Plain pointers, ints and an opaque stream handle only -- no
 * torch, HIP or C++ types -- so the same functions bind from C++, ctypes, cgo, JNI ...
 * (see INTEGRATION.md).  Each entry cites the reference interface it replaces
 * (paths relative to /root/reference/stereoDNN).
 *
 * Conventions
 *   - all tensor pointers are DEVICE pointers; tensors are dense, row-major, batch outermost;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous;
 *   - return value 0 = success; >0 = hipError_t from the runtime; <0 = RT_E_* argument errors.
 *     rt_last_error_string() describes the last failure of the calling thread
 *     (reference convention: enqueue returns 0 / non-zero, lib/cost_volume_plugin.cpp:138);
 *   - dtype: RT_F32 everywhere; RT_F16 where noted (fp16 storage, fp32 accumulate, as the
 *     reference does in lib/kernels.cu:218-223).
 */
#ifndef RT_STEREO_H
#define RT_STEREO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* rtStream;

enum { RT_F32 = 0, RT_F16 = 1 };                     /* nvinfer1::DataType kFLOAT / kHALF            */
enum { RT_NCHW = 0, RT_NC2HW2 = 1 };                 /* nvinfer1::PluginFormat                       */
enum { RT_ACT_NONE = 0, RT_ACT_ELU = 1, RT_ACT_SIGMOID = 2 };
enum { RT_E_BADARG = -1, RT_E_UNSUPPORTED = -2, RT_E_NODEVICE = -3, RT_E_NOMEM = -4, RT_E_RUNTIME = -5 };

/* ---- library / device ------------------------------------------------------------------- */
const char* rt_last_error_string(void);
const char* rt_backend_name(void);                   /* "hip:gfx950 (<device name>)"                 */
int rt_device_count(void);
int rt_set_device(int ordinal);

/* Device memory, copies, streams and timing, so that a host library needs no HIP headers
 * (replaces the cudaMalloc / cudaMemcpy / cudaStream / cudaEvent calls in
 * sample_app/main.cpp:295-315 and lib/conv3d_plugin.cpp:122-133). */
int rt_malloc(void** dptr, size_t bytes);
int rt_free(void* dptr);
int rt_memcpy_h2d(void* dst, const void* src, size_t bytes, rtStream stream);
int rt_memcpy_d2h(void* dst, const void* src, size_t bytes, rtStream stream);
int rt_memcpy_d2d(void* dst, const void* src, size_t bytes, rtStream stream);
int rt_memset(void* dst, int value, size_t bytes, rtStream stream);
int rt_stream_create(rtStream* stream);             /* non-blocking w.r.t. the NULL stream */
int rt_stream_destroy(rtStream stream);
int rt_stream_sync(rtStream stream);
int rt_stream_wait_event(rtStream stream, void* ev);   /* later work on `stream` waits for `ev` */
int rt_event_create(void** ev);
int rt_event_create_ordering(void** ev);              /* no timestamps: for rt_stream_wait_event only (cudaEventDisableTiming) */
int rt_event_destroy(void* ev);
int rt_event_record(void* ev, rtStream stream);
int rt_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms);   /* synchronises on ev_stop */
/* hipGraph of everything issued on `stream` (and on streams that join it through events) between begin and end: an executor
 * captures one pass over fixed bindings and replays it with ONE rt_graph_launch (TensorRT's enqueue is graph-capturable the same
 * way; cudaStreamBeginCapture / cudaGraphInstantiate / cudaGraphLaunch).  Relaxed capture mode; RT_E_UNSUPPORTED on the emulator. */
typedef struct rtGraph rtGraph;
int rt_graph_begin_capture(rtStream stream);
int rt_graph_end_capture(rtStream stream, rtGraph** graph);          /* ends the capture and instantiates; NULL graph on failure */
int rt_graph_launch(rtGraph* graph, rtStream stream);
int rt_graph_destroy(rtGraph* graph);

/* ---- element-wise ------------------------------------------------------------------------ */
/* ELU, alpha = 1: y = x > 0 ? x : exp(x) - 1.  Replaces EluPlugin::enqueue ->
 * cudnnActivationForward (lib/elu_plugin.cpp:123-135).  n = element count (any layout: for
 * kNC2HW2 pass the padded count, lib/elu_plugin.cpp:165-168).  In-place allowed. */
int rt_elu(const void* x, void* y, int64_t n, int dtype, rtStream stream);

/* y = act(a + b) -- TensorRT addElementWise(kSUM) (+ a following ELU plugin when fused),
 * e.g. sample_app/resnet18_2D_513x257_net.cpp:95-103.  In-place allowed. */
int rt_add_act(const void* a, const void* b, void* y, int64_t n, int act, int dtype, rtStream stream);

/* y = act(x) for RT_ACT_SIGMOID / RT_ACT_ELU -- addActivation(kSIGMOID),
 * sample_app/resnet18_2D_513x257_net.cpp:766. */
int rt_activation(const void* x, void* y, int64_t n, int act, int dtype, rtStream stream);

/* ---- cost volumes ------------------------------------------------------------------------ */
/* Correlation cost volume: cv[n,d,y,x] = sum_c L[n,c,y,x] * R[n,c,y,x-d], 0 for x < d.
 * left/right (N,C,H,W) -> cv (N,D,H,W).  Replaces CudaKernels::computeCorrCostVolume /
 * corrCostVolumeKernel (lib/kernels.cu:168-200,252-287).  RT_F16 + RT_NC2HW2 follows
 * corrCostVolumeFP16NC2HW2Kernel (lib/kernels.cu:203-250): channel pairs packed in 32-bit
 * words, fp32 accumulate, output planes packed in disparity pairs. */
int rt_corr_cost_volume(const void* left, const void* right, void* cost_vol, int batch, int C, int H, int W,
                        int max_disp, int dtype, int format, rtStream stream);
/* The same with flags.  fp32 NCHW maps of a network's size (W >= 64, 16 <= C <= 32, max_disp <= 64) are correlated on the matrix
 * cores through the 3-term fp16 split of the convolutions (relative error <= 2^-21 of sum |l r|, inputs |x| < 65504);
 * RT_CONV_EXACT_FP32 keeps the fp32 fmaf chain of corrCostVolumeKernel for them too (CostVolumePlugin::enqueue passes it in engines
 * built with IBuilder::setExactFp32Mode). */
int rt_corr_cost_volume_flags(const void* left, const void* right, void* cost_vol, int batch, int C, int H, int W,
                              int max_disp, int dtype, int format, unsigned flags, rtStream stream);

/* Default (concatenation) cost volume: cv[n,d,0:C]=L ; cv[n,d,C:2C,y,x]=R[n,:,y,x-d] (0 for x<d).
 * (N,C,H,W) x2 -> (N,D,2C,H,W).  Replaces CudaKernels::computeCostVolume (lib/kernels.cu:50-97,136-161). */
int rt_cost_volume(const void* left, const void* right, void* cost_vol, int batch, int C, int H, int W,
                   int max_disp, int dtype, rtStream stream);

/* Soft-argmax (is_min = 0) / soft-argmin (1): out[n,y,x] = sum_d d * softmax_d(+-vol[n,d,y,x]).
 * vol (N,D,H,W) -> out (N,1,H,W), one pass, no workspace.  Replaces SoftargmaxPlugin::enqueue
 * (5 cuDNN/memcpy passes over a 2x workspace, lib/softargmax_plugin.cpp:167-205). */
int rt_softargmax(const void* vol, void* out, int batch, int D, int H, int W, int is_min, int dtype,
                  rtStream stream);

/* Fused correlation cost volume + soft-argmax/min: the D-volume never reaches HBM.
 * out element (n,y,x) is written at out + n*out_batch_stride + y*W + x (elements), which lets the
 * caller place the result directly in a channel of a concatenation buffer
 * (sample_app/resnet18_2D_513x257_net.cpp:601-615). */
int rt_corr_softargmax(const void* left, const void* right, void* out, int batch, int C, int H, int W,
                       int max_disp, int is_min, int64_t out_batch_stride, int dtype, rtStream stream);
/* (fp32 maps of a network's size -- W >= 64, 16 <= C <= 32, max_disp <= 64 -- are correlated on the matrix cores through the 3-term fp16
 * split of the convolutions, inputs |x| < 65504; smaller ones by the fp32 fmaf chain of the reference kernel.)
 * Same with row pitches (elements, 0 = dense) for the feature maps and for the output plane; see
 * rt_conv_plan_set_pitch.  This entry always computes the fp32 chain: it is what engines built with
 * IBuilder::setExactFp32Mode call. */
int rt_corr_softargmax_pitched(const void* left, const void* right, void* out, int batch, int C, int H, int W,
                               int max_disp, int is_min, int in_pitch, int out_pitch, int64_t out_batch_stride,
                               int dtype, rtStream stream);
/* The same fused correlation + soft-argmax on channel-interleaved (C/4, H, pitch, 4) fp32 feature maps -- the executor's
 * internal layout between 3x3 convolutions -- computed on the matrix cores (C a multiple of 4 up to 32, D <= 64). */
int rt_corr_softargmax_il(const void* left, const void* right, void* out, int batch, int C, int H, int W, int D,
                          int is_min, int in_pitch, int out_pitch, int64_t out_bstride, rtStream stream);
/* ... with the map written as lane 0 of the 16-byte pixel slots of a channel-interleaved group, (H, out_pitch, 4) with zeros in
 * lanes 1..3 (out_slot = 4; 1 = the plane above): the disparity channel of conv2D_1's 33-channel input when the executor keeps
 * that concatenation interleaved (sample_app/resnet18_2D_513x257_net.cpp:601-615). */
int rt_corr_softargmax_il_slot(const void* left, const void* right, void* out, int batch, int C, int H, int W, int D,
                               int is_min, int in_pitch, int out_pitch, int64_t out_bstride, int out_slot, rtStream stream);
/* ... in half2 mode: fp16 channel-interleaved feature maps (C/8, H, pitch, 8), the map as an fp16 plane; the stored values are the
 * operands of the matrix instructions (fp32 accumulation of exact fp16 products, lib/kernels.cu:203-250 computes the same in fp32).
 * out_slot = 8: the map as lane 0 of the 16-byte slots of an interleaved group of 8 fp16 channels, zeros in lanes 1 .. 7. */
int rt_corr_softargmax_il8_f16(const void* left, const void* right, void* out, int batch, int C, int H, int W, int D,
                               int is_min, int in_pitch, int out_pitch, int64_t out_bstride, int out_slot, rtStream stream);

/* ---- layout glue of the 3-D models ------------------------------------------------------- */
/* 4-D permute of a (N, d0,d1,d2,d3) tensor: out dim i = in dim order[i].  Replaces
 * TransformPlugin::enqueue -> cudnnTransformTensor (lib/transform_plugin.cpp:94-108). */
int rt_permute4d(const void* x, void* y, int batch, int d0, int d1, int d2, int d3, const int order[4],
                 int dtype, rtStream stream);
/* Format conversion at a plugin boundary -- the "reformat" TensorRT inserts between its fp32 tensors and an IPluginExt that
 * asked for kHALF (reference lib/elu_plugin.cpp:45-53, lib/cost_volume_plugin.cpp:60-66; tests_main.cpp:301-321, 988-1026).
 * Kinds: 0 = fp32 NCHW, 1 = fp16 NCHW, 2 = fp16 NC2HW2 (channel pairs of a pixel in one 4-byte slot, odd C zero padded).
 * x / y: (batch, C, inner) with inner = the product of the remaining dims. */
int rt_convert_format(const void* x, void* y, int batch, int C, int64_t inner, int src_kind, int dst_kind, rtStream stream);

/* (N,D,inner) -> (N,D+pad_end,inner): copy + zero tail.  PaddingPlugin::enqueue (lib/padding_plugin.cpp:79-94). */
int rt_pad_d(const void* x, void* y, int batch, int D, int64_t inner, int pad_end, int dtype, rtStream stream);
/* (N,D,inner) -> (N,end-start,inner).  SlicePlugin::enqueue (lib/slice_plugin.cpp:80-92). */
int rt_slice_d(const void* x, void* y, int batch, int D, int64_t inner, int start, int end, int dtype,
               rtStream stream);
/* Copy `C` channels of x (N,C,inner) into channels [c_off, c_off+C) of y (N,Ctot,inner):
 * TensorRT addConcatenation, sample_app/resnet18_2D_513x257_net.cpp:612-615. */
int rt_concat_channels(const void* x, void* y, int batch, int C, int Ctot, int c_off, int64_t inner, int dtype,
                       rtStream stream);

/* ---- image front-end / back-end of the sample application ----------------------------------- */
/* u8 BGR HWC image(s) (N, src_h, src_w, 3) -> float RGB CHW (N, 3, dst_h, dst_w) in [0,1]: convertTo(CV_32F),
 * cv::resize(INTER_AREA) (down-scaling or same size), BGR -> RGB, HWC -> CHW, / 255 of readImgFile
 * (sample_app/main.cpp:83-98; stereo_dnn_ros_node.cpp:42-103), on the device. */
int rt_preprocess_bgr8(const void* src_u8, int src_h, int src_w, void* dst_f32, int dst_h, int dst_w, int batch,
                       rtStream stream);
/* disparity map -> 16-bit KITTI encoding: saturate_cast<ushort>(round(disp * scale)), scale = 256 (x width for the
 * normalised output of ResNet-18 2D), sample_app/main.cpp:324-330. */
int rt_disparity_to_u16(const void* disp_f32, void* out_u16, int64_t n, float scale, rtStream stream);

/* ---- convolutions (MFMA implicit GEMM) ----------------------------------------------------- */
/* A plan owns the device copy of the (re-packed) weights, bias and gather tables of one layer,
 * like Conv3DPlugin::configure owns kernel_weights_d_ (lib/conv3d_plugin.cpp:122-133). */
typedef struct rtConvPlan rtConvPlan;

typedef struct rtConv2dDesc {
    int Cin, Cout;          /* channels                                                          */
    int Hin, Win;           /* input plane                                                       */
    int KH, KW;             /* 3x3 or 5x5                                                        */
    int stride;             /* 1 or 2 (same in H and W)                                          */
    int pad_h, pad_w;       /* symmetric zero padding (TRT setPadding)                           */
    int act;                /* RT_ACT_* fused after bias (+ residual)                            */
    int has_residual;       /* enqueue takes a residual tensor shaped like the output            */
    int dtype;              /* RT_F32 | RT_F16: storage type of activations AND weights          */
    int flags;              /* RT_CONV_* option bits, 0 = defaults                               */
} rtConv2dDesc;

/* rtConv2dDesc::flags / rtConv3dDesc::flags.  RT_CONV_EXACT_FP32: keep the convolution on the fp32 fmaf-chain kernels (fp32 MFMA /
 * Winograd F(2x2,3x3)) instead of the default 3-term fp16 split on the fp16 matrix pipe (22-bit operands, fp32 accumulation; domain
 * |x| < 65504, see rt_check_range).  IBuilder::setExactFp32Mode / rtNetOptions::flags set it for a whole engine. */
#define RT_CONV_EXACT_FP32 1

/* 2-D convolution (cross-correlation), weights KCRS, bias K (may be NULL): TensorRT
 * addConvolution as called at sample_app/resnet18_2D_513x257_net.cpp:48-53 (no source in the
 * reference: TensorFlow conv2d semantics, scripts/tensorrt_model_builder.py:149-228).
 * x (N,Cin,Hin,Win) -> y (N,Cout,Hout,Wout), Hout = (Hin + 2*pad - KH)/stride + 1. */
int rt_conv2d_plan_create(rtConvPlan** plan, const rtConv2dDesc* desc, const void* weights_host,
                          const void* bias_host);
/* 2-D transposed convolution, weights (Cin,Cout,R,S), stride 2, 3x3: TensorRT addDeconvolution
 * (sample_app/resnet18_2D_513x257_net.cpp:722-727) = TF conv2d_transpose
 * (scripts/tensorrt_model_builder.py:230-288).  Hout = (Hin-1)*stride - 2*pad + KH. */
int rt_deconv2d_plan_create(rtConvPlan** plan, const rtConv2dDesc* desc, const void* weights_host,
                            const void* bias_host);

/* Fused residual block  y = act2(conv3x3(act1(conv3x3(x) + b1)) + b2 + x): the unit of the ResNet-18 feature towers
 * (reference resnet18_2D_513x257_net.cpp:66-575: resblockN_conv1, ELU plugin, resblockN_conv2, addElementWise(kSUM),
 * ELU plugin).  d1 / d2 describe the two addConvolution layers as rt_conv2d_plan_create would take them (d2->has_residual
 * = 1); RT_E_UNSUPPORTED unless both are 3x3, stride 1, pad 1, with <= 32 channels and equal input / output channel
 * counts.  Enqueue with residual = x (or NULL). */
int rt_resblock_plan_create(rtConvPlan** plan, const rtConv2dDesc* d1, const void* w1, const void* b1,
                            const rtConv2dDesc* d2, const void* w2, const void* b2);

/* Pre-split tensors between the blocks of a feature tower (not in the reference: TensorRT owns its internal formats): per sample
 * (C/8, H, pitch, [8 x fp16 hi | 8 x fp16 lo]) with x = hi + lo * 2^-11 -- the operands of the split-fp16 matrix instructions as stored
 * values, 4 bytes per element like the fp32 tensor it replaces (same pitch, same sample stride).  A block that READS one fills its LDS by
 * direct global -> LDS loads; a block that WRITES one splits its results in the epilogue.  Only the tower block takes them (32 -> 32 -> 32
 * channels, ELU after both convolutions of resblockN, reference resnet18_2D_513x257_net.cpp:66-575, on channel-interleaved fp32 tensors:
 * rt_conv_plan_set_layouts(plan, 1, 1, 1) first); rt_resblock_plan_supports_split says whether this plan does.  The skip connection of a
 * block with x_split is taken from the split tensor (22 bits). */
int rt_resblock_plan_supports_split(const rtConvPlan* plan);
int rt_resblock_plan_set_split(rtConvPlan* plan, int x_split, int y_split);

/* Row pitch (in elements, >= the row length; 0 = dense) of the input and of the output/residual planes of a 2-D
 * plan.  Not in the reference (TensorRT owns its internal layouts): lets the executor keep internal activations
 * 128-byte aligned per row.  Tensors are then (N, C, H, pitch) in memory with W valid columns. */
int rt_conv_plan_set_pitch(rtConvPlan* plan, int in_pitch, int out_pitch);
/* Per-sample strides in elements (0 = unchanged) of input / output / residual: a tensor may live inside a larger buffer,
 * e.g. as a channel range of the tensor TensorRT's addConcatenation would have copied it into
 * (reference resnet18_2D_513x257_net.cpp:612-615).  Call after rt_conv_plan_set_pitch. */
int rt_conv_plan_set_batch_strides(rtConvPlan* plan, int64_t x_bstride, int64_t y_bstride, int64_t r_bstride);

/* Storage type (RT_F32 / RT_F16) of the input and of the output + residual tensors of a 2-D plan: TensorRT's half2
 * mode (IBuilder::setHalf2Mode, sample_app/main.cpp:256-262) keeps activations in fp16 between layers.
 * fp16 -> fp16 (3x3 stride 1/2, transposed 3x3 stride 2): the plan's weights are re-packed as fp16 and the stored
 * values become the operands of the fp16 matrix instructions, accumulation stays fp32 (RT_NO_F16MMA=1 in the
 * environment keeps fp32 arithmetic).  fp32 -> fp16: the network's first layer (5x5 stride 2, <= 3 input channels)
 * rounds image and weights to fp16 and runs on the fp16 matrix instructions too, as TensorRT's half2 mode does; other
 * fp32 -> fp16 layers (direct-form kernels) and fp16 -> fp32 (last layer, small-output transposed kernel) compute in
 * fp32.  Other combinations return RT_E_UNSUPPORTED. */
int rt_conv_plan_set_io_types(rtConvPlan* plan, int x_dtype, int y_dtype);

/* Channel-interleaved tensors, (C/8, H, pitch, 8) per sample in fp16 and (C/4, H, pitch, 4) in fp32: one 16-byte slot
 * per pixel and channel group.
 * Not in the reference (TensorRT owns its internal layouts; its own fp16 formats are of this kind, PluginFormat
 * kNC2HW2 / kNHWC8 in NvInfer.h).  The 3x3 stride-1 kernels (fp32 Winograd, fp16 arithmetic) move such tensors in full
 * cache lines with a quarter of the memory instructions; the executor
 * uses the layout for tensors that only 3x3 stride-1 plans in fp16 arithmetic (and, as output, the first layer)
 * touch.  rt_conv_plan_supports_il8: which tensors of the plan may be interleaved (bit 0 input, bit 1 output, bit 2
 * residual, bit 3: an interleaved output only together with an interleaved input, bit 4: an interleaved input whose channel
 * count is padded up to a whole group -- the pad channels must hold finite values, bit 5: the input must stay planar fp32 (the
 * factored cost-volume fold of the first Conv3D); call it after
 * rt_conv_plan_set_io_types); rt_conv_plan_set_layouts: layout (0 planar, 1 interleaved) of
 * the input, the output and the residual tensor. */
int rt_conv_plan_supports_il8(const rtConvPlan* plan);
int rt_conv_plan_set_layouts(rtConvPlan* plan, int x_il8, int y_il8, int r_il8);

/* The soft-argmax that follows the last Conv3DTranspose of the 3-D models (disp_softargmax: reference lib/softargmax_plugin.cpp:167-205
 * over the volume lib/conv3d_transpose_plugin.cpp:137-166 wrote) inside that layer's launch: mode 1 = soft-argmax, 2 = soft-argmin over the
 * output depth, 0 = off.  rt_conv_enqueue then writes the (batch, 1, H, W) fp32 map to y (plain pitch W) and the (D, 1, H, W) volume is
 * never stored.  Only the depth-walking form of that layer has it -- one output channel, 32 input channels, fp16 channel-interleaved input,
 * fp32 output, no residual; any other plan returns RT_E_UNSUPPORTED and stays as it is (the caller then runs rt_softargmax on the volume).
 * Call it last: rt_conv_plan_set_io_types / rt_conv_plan_set_layouts switch it off. */
int rt_conv_plan_set_softarg(rtConvPlan* plan, int mode);

typedef struct rtConv3dDesc {
    int C, K;               /* conv: input channels C, output channels K.  Transposed op: K = INPUT   */
                            /* channels (tensor KDHW), C = OUTPUT channels (tensor DCHW)             */
    int D, H, W;            /* conv: input dims (D,C,H,W); transposed: OUTPUT dims (D,C,H,W)         */
    int kernel[3];          /* (V,R,S): filter is (K,V,C,R,S); R = S in {1,3}                        */
    int stride[3];          /* (d,h,w), h == w                                                      */
    int pad_start[3];       /* (d,h,w) -- what the reference hands to cuDNN                         */
    int pad_end[3];         /* validated like lib/conv3d_plugin.cpp:43-49, otherwise unused          */
    int act;                /* RT_ACT_* (0 at the plugin boundary; used by the fusing executor)     */
    int out_dchw;           /* conv: write (Do,K,Ho,Wo) instead of (K,Do,Ho,Wo); transposed: write (C,D,H,W)  */
                            /* instead of (D,C,H,W) -- the Transform plugin that follows, in the same pass  */
    int has_residual;       /* residual tensor: shaped like the output; for the transposed op always (D,C,H,W) */
    int dtype;
    int out_depth;          /* transposed only: keep output depth slices [0, out_depth) (0 = all) -- the      */
                            /* Slice plugin that follows an even-depth conv3d_transpose, in the same pass      */
    int in_pad_end;         /* conv only: the last in_pad_end of the D input slices are zeros that do not exist  */
                            /* in memory (x is (D - in_pad_end, C, H, W)) -- the Pad plugin emitted before every  */
                            /* stride-2 Conv3D (scripts/tensorrt_model_builder.py:331-345), in the same pass     */
    int cv_fold;            /* conv only, executor: F > 0 = the input is the DEFAULT COST VOLUME of two (F,H,W) feature  */
                            /* maps (C == 2F; CostVolumePlugin kDefault, lib/kernels.cu:50-97) that is never built:     */
                            /* x points to the (2F, H, W) tensor [left | right] and slice d of the volume is gathered   */
                            /* as cv[d, 0:F] = L, cv[d, F:2F, y, x] = R[:, y, x - d] (0 for x < d)                       */
    int flags;              /* RT_CONV_* option bits (see rtConv2dDesc), 0 = defaults                               */
} rtConv3dDesc;

/* TensorFlow-compatible 3-D convolution.  x (N, D,C,H,W) , w (K,V,C,R,S) 3x3x3 -> y (N, K,Do,Ho,Wo).
 * Replaces Conv3DPlugin::enqueue -> cudnnConvolutionForward on the (D*C)-merged descriptor +
 * cudnnAddTensor bias (lib/conv3d_plugin.cpp:187-216, lib/conv_utils.cpp:27-32,58-72). */
int rt_conv3d_plan_create(rtConvPlan** plan, const rtConv3dDesc* desc, const void* weights_host,
                          const void* bias_host);
/* TensorFlow-compatible 3-D transposed convolution.  y (N, K,Dy,Hy,Wy), w (K,V,C,R,S) -> x (N, D,C,H,W),
 * bias per C.  Replaces Conv3DTransposePlugin::enqueue -> cudnnConvolutionBackwardData +
 * addDBiasTo3DConvKernel (lib/conv3d_transpose_plugin.cpp:205-243, lib/kernels.cu:292-335).
 * in_dims = (Dy,Hy,Wy) of the input. */
int rt_conv3d_transpose_plan_create(rtConvPlan** plan, const rtConv3dDesc* desc, const int in_dims[3],
                                    const void* weights_host, const void* bias_host);

/* 1 when the library was built with -DRT_EXPERIMENTAL: the measured-and-rejected kernel families (persistent split kernel, per-tile fused
 * residual block, 8-row / 8-wave tiles, Winograd on interleaved tensors) are then compiled in and selectable through development knobs.
 * The product build returns 0; the CPU test tier's emulator build returns 1. */
int rt_has_experimental(void);

/* The domain of a plan's arithmetic.  The default fp32 path multiplies 22-bit fp16 splits on the fp16 matrix pipe: an input value with
 * |x| >= 65504 (or a non-finite one) becomes inf / NaN in the result -- loud, but far from its cause.  rt_conv_plan_input_limit gives the
 * bound a plan needs (65504 for split / fp16-operand plans, +inf for RT_CONV_EXACT_FP32 and the small direct kernels);
 * rt_check_range scans a device tensor of `rows` rows with `valid` leading elements out of `pitch` each (dense: rows = 1, valid = pitch = n):
 * max |x| and the number of elements with |x| >= limit or non-finite.  Blocking.  IExecutionContext::setDebugSync(true) /
 * rt_net_set_debug make the executor run it on the input of every such launch and fail the execute() with the layer's name. */
int rt_conv_plan_input_limit(const rtConvPlan* plan, float* limit);
int rt_check_range(const void* x, int64_t rows, int64_t valid, int64_t pitch, int dtype, float limit, float* max_abs, int64_t* violations,
                   rtStream stream);

/* Order-independent 64-bit checksum of `bytes` (a multiple of 4) of device memory, written to the device word `out_dev` on `stream`
 * (asynchronous).  Two buffers with the same bits give the same value; the executor's launch trace (rt_net_set_launch_trace) hashes every
 * launch's output with it so that two passes over the same input can be compared launch by launch without copying tensors. */
int rt_hash_buffer(const void* x, size_t bytes, unsigned long long* out_dev, rtStream stream);

/* Output dims of a plan: 2-D -> (Cout,Hout,Wout,1); 3-D -> 4 dims in the order they are written. */
int rt_conv_plan_out_dims(const rtConvPlan* plan, int dims[4]);
/* Run: x, y (and residual, shaped like y, or NULL) are device pointers for `batch` samples. */
int rt_conv_enqueue(const rtConvPlan* plan, const void* x, void* y, const void* residual, int batch,
                    rtStream stream);
/* The same with launch hints (0 = rt_conv_enqueue).  RT_HINT_THROUGHPUT: the caller keeps several launches in flight on the device
 * (execution contexts with one stream each, TensorRT's `trtexec --streams` set-up; IExecutionContext::setExecutionStreams(1)): kernels
 * may then trade the duration of one launch for device time per result -- the streaming residual block walks 64-row instead of 32-row
 * segments (fewer pipeline fill / drain steps per row, half the workgroups per launch); without the hint, launches that would leave a
 * SIMD a single wave split their contraction over wave groups (split-K: another fp32 summation order).  Results are deterministic for
 * a given (hints, device CU count) and agree across hints to fp32 rounding (<= 1e-5 on the networks' outputs), not bit for bit. */
#define RT_HINT_THROUGHPUT 1
int rt_conv_enqueue_hint(const rtConvPlan* plan, const void* x, void* y, const void* residual, int batch,
                         rtStream stream, int hints);
/* The first layer of both feature towers in one launch: samples 0 .. batch - 1 from x (left images), batch .. 2 batch - 1 from x2 (right
 * images -- the reference's two input bindings, sample_app/main.cpp:290-300), y = 2 * batch samples.  The towers share their weights, so
 * this is the [left | right] launch the executor makes of every tower layer; only first-layer plans (5x5 stride 2, <= 3 input channels,
 * rt_conv_plan_supports_twin_input) have it. */
int rt_conv_plan_supports_twin_input(const rtConvPlan* plan);
int rt_conv_enqueue_twin_input(const rtConvPlan* plan, const void* x, const void* x2, void* y, int batch, rtStream stream, int hints);

/* Scratch a plan's launches need for `batch` samples, in bytes (0 for most plans; the first Conv3D over a folded cost volume in its
 * factored form keeps four small maps per sample).  The caller owns it -- an execution context passes its workspace, as TensorRT hands a
 * plugin's getWorkspaceSize() bytes to enqueue (reference lib/conv3d_plugin.cpp:179-185, 187-190) -- so that contexts sharing a plan do
 * not share scratch: 16-byte aligned device memory of at least that size, private to the stream until the launches have run.
 * rt_conv_enqueue / _hint on such a plan use blocks owned by the plan instead, one per stream they are called with (safe across streams;
 * a block grows after its stream has drained, which a stream capture cannot record: capturing callers pass a workspace). */
size_t rt_conv_plan_workspace_bytes(const rtConvPlan* plan, int batch);
int rt_conv_enqueue_ws(const rtConvPlan* plan, const void* x, void* y, const void* residual, int batch, void* workspace,
                       size_t workspace_bytes, rtStream stream, int hints);
int rt_conv_plan_destroy(rtConvPlan* plan);

/* ---- multi-GPU: RCCL communicator and byte broadcast ---------------------------------------------------------
 * Stereo pairs are independent (the reference is single-GPU, batch 1: sample_app/main.cpp:260, 303-305), so ranks share
 * nothing but the weight-file image: rank `root` reads trt_weights.bin, every other rank receives it over RCCL / xGMI
 * (ncclBroadcast) at start-up, and no collective touches the data path.  librccl is loaded on first use (dlopen), so
 * single-GPU users never pay for it.  One communicator per (process or thread, device): call rt_set_device first.
 *   rank 0:  rt_comm_unique_id(id)  -> ship the 128 bytes to the other ranks by any means (file, socket, MPI, torchrun's store)
 *   all:     rt_comm_init_rank(&comm, world, rank, id)          (ncclCommInitRank)
 *   or:      rt_comm_adopt(&comm, existing ncclComm_t)           (a communicator the host application already has; not destroyed)
 *   or:      rt_comm_init_all(comms, ndev, devices)              (one process driving several devices: ncclCommInitAll)
 *   all:     rt_comm_broadcast(comm, host_buffer, bytes, root, stream)   blocking; bytes must agree on every rank
 * Calls of one process that must progress together (several comms driven from one thread) go between
 * rt_comm_group_start / rt_comm_group_end. */
typedef struct rtComm rtComm;
#define RT_COMM_ID_BYTES 128
int rt_comm_unique_id(void* id_bytes);
int rt_comm_init_rank(rtComm** comm, int world, int rank, const void* id_bytes);
int rt_comm_init_all(rtComm** comms, int ndev, const int* devices);
int rt_comm_adopt(rtComm** comm, void* nccl_comm);
int rt_comm_info(const rtComm* comm, int* world, int* rank);
int rt_comm_broadcast(rtComm* comm, void* host_buf, size_t bytes, int root, rtStream stream);
int rt_comm_group_start(void);
int rt_comm_group_end(void);
int rt_comm_destroy(rtComm* comm);

#ifdef __cplusplus
}
#endif
#endif /* RT_STEREO_H */
