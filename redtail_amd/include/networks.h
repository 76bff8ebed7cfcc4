// networks.h -- Stereo DNN network builders for the NvInfer.h shim.
//
// Same four entry points as /root/reference/stereoDNN/sample_app/networks.h:20-36 (the reference ships
// them as ~2 700 lines of generated code, one file per fixed resolution).  Here each model is a short
// programmatic builder that follows the generator scripts (scripts/model_{nvsmall,resnet18,resnet18_2D}.py)
// and accepts ANY legal resolution, which is what lets ResNet-18 2D run at BASELINE's 1257x369.  The
// weight names and layouts are those of scripts/tensorrt_model_builder.py (trt_weights.bin).
#ifndef REDTAIL_AMD_NETWORKS_H
#define REDTAIL_AMD_NETWORKS_H

#include <NvInfer.h>

#include <string>
#include <unordered_map>

namespace redtail { namespace tensorrt {

using namespace nvinfer1;

using weight_map = std::unordered_map<std::string, Weights>;

class IPluginContainer;

// NVSmall: 1025x321 input, 96 max disparity (cost volume D = 48 at half resolution).
INetworkDefinition* createNVSmall1025x321Network(IBuilder& builder, IPluginContainer& plugin_factory, DimsCHW img_dims,
                                                 const weight_map& weights, DataType data_type, ILogger& log);
// NVTiny: 513x161 input, 48 max disparity (D = 24).
INetworkDefinition* createNVTiny513x161Network(IBuilder& builder, IPluginContainer& plugin_factory, DimsCHW img_dims,
                                               const weight_map& weights, DataType data_type, ILogger& log);
// ResNet-18 (3-D): 1025x321 input, 136 max disparity (D = 68).
INetworkDefinition* createResNet18_1025x321Network(IBuilder& builder, IPluginContainer& plugin_factory, DimsCHW img_dims,
                                                   const weight_map& weights, DataType data_type, ILogger& log);
// ResNet-18 2D: 513x257 input in the reference; any W,H = 1 (mod 8) here.  Correlation D = 48.
INetworkDefinition* createResNet18_2D_513x257Network(IBuilder& builder, IPluginContainer& plugin_factory, DimsCHW img_dims,
                                                     const weight_map& weights, DataType data_type, ILogger& log);

// Resolution-generic builders behind the four entry points above (max_disp = half-resolution D).
INetworkDefinition* createResNet18_2DNetwork(IBuilder& builder, IPluginContainer& plugin_factory, DimsCHW img_dims,
                                             const weight_map& weights, DataType data_type, int max_disp, ILogger& log);
enum class Stereo3DModel { kNVSmall, kNVTiny, kResNet18 };
INetworkDefinition* createStereo3DNetwork(IBuilder& builder, IPluginContainer& plugin_factory, Stereo3DModel model,
                                          DimsCHW img_dims, const weight_map& weights, DataType data_type, int max_disp,
                                          ILogger& log);

} }  // namespace redtail::tensorrt

#endif
