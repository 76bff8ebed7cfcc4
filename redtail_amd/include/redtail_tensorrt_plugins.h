// redtail_tensorrt_plugins.h -- public API of the Stereo DNN plugin library, MI355X build.
//
// This header is the drop-in boundary: it declares, name for name and argument for argument, the API of
// /root/reference/stereoDNN/lib/redtail_tensorrt_plugins.h (enums :18-40, IPluginContainer :49-86, the
// add* graph helpers :91-124, StereoDnnPluginFactory :129-146), so that the reference's generated
// network builders (sample_app/*_net.cpp) and application code compile against it unchanged.  The
// implementation behind it (redtail_amd/csrc/host/plugins.cpp) has no cuDNN / CUDA: every enqueue is
// one call into the HIP kernels through include/rt_stereo.h.
#ifndef REDTAIL_TENSORRT_PLUGINS_H
#define REDTAIL_TENSORRT_PLUGINS_H

#include <NvInfer.h>

#include <memory>
#include <string>

namespace redtail { namespace tensorrt {

using namespace nvinfer1;

// How the 3-D convolution plugins interpret their tensors: kCuDnn = (C,D,H,W) volumes, kTensorFlow =
// (D,C,H,W) input / (K,V,C,R,S) filter as produced by the TF -> TRT converter.
enum class Conv3DType { kCuDnn = 0, kTensorFlow = 1 };

// kDefault: concatenation volume, two (C,H,W) inputs -> (D,2C,H,W); kCorrelation: -> (D,H,W).
enum class CostVolumeType { kDefault = 0, kCorrelation = 1 };

enum class SoftargmaxType { kMax = 0, kMin = 1 };

// Factory and owner of every plugin instance (plugins must outlive the engine that references them;
// create* returns a naked pointer exactly as TensorRT expects).
class IPluginContainer {
public:
    virtual ~IPluginContainer() = default;

    virtual IPlugin* createEluPlugin(DataType data_type, std::string name) = 0;
    virtual IPlugin* deserializeEluPlugin(const char* name, const void* data, size_t size) = 0;

    virtual IPlugin* createCostVolumePlugin(DataType data_type, CostVolumeType cv_type, int max_disparity,
                                            std::string name) = 0;
    virtual IPlugin* deserializeCostVolumePlugin(const char* name, const void* data, size_t size) = 0;

    virtual IPlugin* createConv3DPlugin(Conv3DType conv_type, Dims kernel_dims, Dims stride_dims, Dims pad_start_dims,
                                        Dims pad_end_dims, Weights kernel_weights, Weights bias_weights,
                                        std::string name) = 0;

    virtual IPlugin* createConv3DTransposePlugin(Conv3DType conv_type, Dims kernel_dims, Dims out_dims, Dims stride_dims,
                                                 Dims pad_start_dims, Dims pad_end_dims, Weights kernel_weights,
                                                 Weights bias_weights, std::string name) = 0;

    virtual IPlugin* createTransformPlugin(Permutation permutation, std::string name) = 0;

    virtual IPlugin* createPaddingPlugin(DimsNCHW pad_start, DimsNCHW pad_end, std::string name) = 0;

    virtual IPlugin* createSlicePlugin(Dims dims, Dims slice_start, Dims slice_end, std::string name) = 0;

    virtual IPlugin* createSoftargmaxPlugin(DataType data_type, SoftargmaxType sm_type, std::string name) = 0;
    virtual IPlugin* deserializeSoftargmaxPlugin(const char* name, const void* data, size_t size) = 0;

    static std::unique_ptr<IPluginContainer> create(ILogger& log);
};

// Graph helpers: create the plugin through the container and append it to the network.
ILayer* addElu(IPluginContainer& plugin_factory, INetworkDefinition& network, ITensor& input, DataType data_type,
               const std::string& name);

ILayer* addCostVolume(IPluginContainer& plugin_factory, INetworkDefinition& network, ITensor& left_input,
                      ITensor& right_input, CostVolumeType cv_type, int max_disparity, DataType data_type,
                      const std::string& name);

ILayer* addConv3D(IPluginContainer& plugin_factory, INetworkDefinition& network, ITensor& input, Conv3DType conv_type,
                  Dims kernel_dims, Dims stride_dims, Dims pad_start_dims, Dims pad_end_dims, Weights kernel_weights,
                  Weights bias_weights, const std::string& name);

ILayer* addConv3DTranspose(IPluginContainer& plugin_factory, INetworkDefinition& network, ITensor& input,
                           Conv3DType conv_type, Dims kernel_dims, Dims out_dims, Dims stride_dims, Dims pad_start_dims,
                           Dims pad_end_dims, Weights kernel_weights, Weights bias_weights, const std::string& name);

ILayer* addSlice(IPluginContainer& plugin_factory, INetworkDefinition& network, ITensor& input, Dims dims,
                 Dims slice_start, Dims slice_end, const std::string& name);

ILayer* addTransform(IPluginContainer& plugin_factory, INetworkDefinition& network, ITensor& input,
                     Permutation permutation, const std::string& name);

ILayer* addPad(IPluginContainer& plugin_factory, INetworkDefinition& network, ITensor& input, DimsNCHW pad_start,
               DimsNCHW pad_end, const std::string& name);

ILayer* addSoftargmax(IPluginContainer& plugin_factory, INetworkDefinition& network, ITensor& input,
                      SoftargmaxType sm_type, DataType data_type, const std::string& name);

// Re-creates serialisable plugins (ELU, cost volume, soft-argmax) when an engine plan is loaded;
// the blob starts with an int32 PluginType.
class StereoDnnPluginFactory : public IPluginFactory {
public:
    enum class PluginType { kElu = 0, kCostVolume = 1, kSoftargmax = 2 };

    StereoDnnPluginFactory(IPluginContainer& container);

    IPlugin* createPlugin(const char* layerName, const void* serialData, size_t serialLength) override;

private:
    IPluginContainer& container_;
};

} }  // namespace redtail::tensorrt

#endif  // REDTAIL_TENSORRT_PLUGINS_H
