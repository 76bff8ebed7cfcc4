// Internal view of the Stereo DNN plugins for the fusing executor (engine.cpp).  Not part of the
// public API: TensorRT only ever saw IPlugin; our executor owns both sides and may look inside.
#ifndef REDTAIL_AMD_PLUGIN_INTERNAL_H
#define REDTAIL_AMD_PLUGIN_INTERNAL_H

#include <string>

#include "redtail_tensorrt_plugins.h"

namespace redtail { namespace tensorrt { namespace internal {

enum class Kind { kElu, kCostVolume, kSoftargmax, kConv3D, kConv3DTranspose, kTransform, kPadding, kSlice };

struct ConvFusion {            // what the executor asks a Conv3D / Conv3DTranspose plugin to absorb
    int act = 0;               // RT_ACT_*
    bool out_dchw = false;     // write the transposed layout directly (elides the Transform {1,0,2,3} plugin):
                               // Conv3D (K,D,H,W) -> (D,K,H,W); Conv3DTranspose (D,C,H,W) -> (C,D,H,W)
    bool residual = false;     // add a residual tensor before the activation (Conv3DTranspose: always (D,C,H,W))
    int out_depth = 0;         // Conv3DTranspose only: keep output slices [0, out_depth) (elides the Slice plugin)
    int in_pad_end = 0;        // Conv3D only: the last input slices are implicit zeros (elides the Pad plugin)
    int cv_fold = 0;           // Conv3D only: F > 0 = the input is the default cost volume of two (F,H,W) feature maps, gathered
                               // from the (2F,H,W) tensor [left | right] instead of being built (elides the CostVolume plugin)
};

class IStereoPlugin {
public:
    virtual Kind kind() const = 0;
    virtual const std::string& pluginName() const = 0;
    // cost volume / soft-argmax
    virtual CostVolumeType costVolumeType() const { return CostVolumeType::kDefault; }
    virtual int maxDisparity() const { return 0; }
    // engines built with IBuilder::setExactFp32Mode keep a stand-alone correlation on the fp32 fmaf kernel (rt_corr_cost_volume_flags)
    virtual void setExactFp32(bool) {}
    virtual SoftargmaxType softargmaxType() const { return SoftargmaxType::kMax; }
    // transform
    virtual Permutation permutation() const { return Permutation{{0, 1, 2, 3}}; }
    // padding / slice (outermost dim only)
    virtual int padEnd() const { return 0; }
    virtual int sliceStart() const { return 0; }
    virtual int sliceEnd() const { return 0; }
    // 3-D convolutions: re-plan with fused epilogue; enqueueFused takes the residual pointer
    virtual bool setFusion(const ConvFusion&) { return false; }
    virtual ConvFusion fusion() const { return ConvFusion(); }
    // half2 mode: store the 3-D input / output (+ residual) tensors as fp16 (kept across setFusion)
    virtual bool setIoTypes(bool, bool) { return false; }
    // ... and which of them may be channel-interleaved, (D, C/8, H, W, 8) (rt_conv_plan_supports_il8: bit 0 input, 1 output, 2 residual,
    // 3 = output only together with the input), and their layouts (kept across setFusion)
    virtual int ilCaps() const { return 0; }
    virtual bool setLayouts(bool, bool, bool) { return false; }
    // Conv3DTranspose: end the launch in the soft-argmax (1) / soft-argmin (2) over the output depth that follows it (rt_conv_plan_set_softarg;
    // declared after the types and layouts, which reset it); false = the plan has no such form
    virtual bool setSoftarg(int) { return false; }
    virtual int softarg() const { return 0; }       // what the plan currently ends in (the executor checks it against what it planned for)
    // (workspace: getWorkspaceSize(maxBatchSize) bytes private to the stream, or null -- the plan then uses a block of its own)
    virtual int enqueueFused(int, const void*, void*, const void*, void*, size_t, cudaStream_t) { return -1; }
    virtual ~IStereoPlugin() {}
};

void logError(ILogger& log, int status, const char* file, int line, const char* func);

} } }  // namespace redtail::tensorrt::internal

// Logs "<file>:<line>: <func>: ..." at kERROR when status != 0 (reference: lib/internal_macros.h:14-21).
#define RT_CHECKL(status, log)                                                                        \
    do {                                                                                              \
        int res_ = (int)(status);                                                                     \
        if (res_ != 0) ::redtail::tensorrt::internal::logError(log, res_, __FILE__, __LINE__, __func__); \
    } while (false)

#endif
