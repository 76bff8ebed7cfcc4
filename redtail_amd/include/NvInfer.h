// NvInfer.h -- the slice of the TensorRT 3/4 C++ API that redtail's Stereo DNN sources are written
// against, re-declared from scratch so that those sources (generated network builders, plugin
// tests, sample application logic) compile and link on ROCm, where TensorRT does not exist.
//
// Nothing here comes from NVIDIA headers (TensorRT is closed source and is not part of the
// reference tree); the surface was enumerated from the call sites in
//   /root/reference/stereoDNN/lib/redtail_tensorrt_plugins.h:18-146   (plugin API)
//   /root/reference/stereoDNN/lib/*_plugin.cpp                        (IPlugin / IPluginExt overrides)
//   /root/reference/stereoDNN/sample_app/*_net.cpp, main.cpp:136-340  (network building, execution)
//   /root/reference/stereoDNN/tests/tests_main.cpp:96-250             (test harness)
// and is implemented by redtail_amd/csrc/host/engine.cpp: a static-graph builder with a fusing
// executor that launches the gfx950 kernels through the C ABI of include/rt_stereo.h.
#ifndef REDTAIL_AMD_NVINFER_SHIM_H
#define REDTAIL_AMD_NVINFER_SHIM_H

#include <cstddef>
#include <cstdint>

#include "cuda_runtime_api.h"   // cudaStream_t / cudaEvent_t as opaque HIP handles

#define NV_TENSORRT_MAJOR 4
#define NV_TENSORRT_MINOR 0
#define NV_TENSORRT_PATCH 0

namespace nvinfer1 {

template <typename T> inline int EnumMax();

enum class DataType : int { kFLOAT = 0, kHALF = 1, kINT8 = 2, kINT32 = 3 };
template <> inline int EnumMax<DataType>() { return 4; }

enum class DimensionType : int { kSPATIAL = 0, kCHANNEL = 1, kINDEX = 2, kSEQUENCE = 3 };
template <> inline int EnumMax<DimensionType>() { return 4; }

// ---- dimensions (aggregate, so `Dims{3, {1, 1, 1}}` works as in the reference) ----------------------
class Dims {
public:
    static const int MAX_DIMS = 8;
    int nbDims;
    int d[MAX_DIMS];
    DimensionType type[MAX_DIMS];
};

class Dims2 : public Dims {
public:
    Dims2() { nbDims = 2; d[0] = d[1] = 0; type[0] = type[1] = DimensionType::kSPATIAL; }
    Dims2(int d0, int d1) { nbDims = 2; d[0] = d0; d[1] = d1; type[0] = type[1] = DimensionType::kSPATIAL; }
};
class DimsHW : public Dims2 {
public:
    DimsHW() : Dims2() {}
    DimsHW(int height, int width) : Dims2(height, width) {}
    int& h() { return d[0]; }
    int h() const { return d[0]; }
    int& w() { return d[1]; }
    int w() const { return d[1]; }
};
class Dims3 : public Dims {
public:
    Dims3() { nbDims = 3; d[0] = d[1] = d[2] = 0; type[0] = DimensionType::kCHANNEL; type[1] = type[2] = DimensionType::kSPATIAL; }
    Dims3(int d0, int d1, int d2) { nbDims = 3; d[0] = d0; d[1] = d1; d[2] = d2; type[0] = DimensionType::kCHANNEL; type[1] = type[2] = DimensionType::kSPATIAL; }
};
class DimsCHW : public Dims3 {
public:
    DimsCHW() : Dims3() {}
    DimsCHW(int channels, int height, int width) : Dims3(channels, height, width) {}
    int& c() { return d[0]; }
    int c() const { return d[0]; }
    int& h() { return d[1]; }
    int h() const { return d[1]; }
    int& w() { return d[2]; }
    int w() const { return d[2]; }
};
class Dims4 : public Dims {
public:
    Dims4() { nbDims = 4; for (int i = 0; i < 4; i++) { d[i] = 0; type[i] = DimensionType::kSPATIAL; } type[0] = DimensionType::kINDEX; type[1] = DimensionType::kCHANNEL; }
    Dims4(int d0, int d1, int d2, int d3) { nbDims = 4; d[0] = d0; d[1] = d1; d[2] = d2; d[3] = d3; for (int i = 0; i < 4; i++) type[i] = DimensionType::kSPATIAL; type[0] = DimensionType::kINDEX; type[1] = DimensionType::kCHANNEL; }
};
class DimsNCHW : public Dims4 {
public:
    DimsNCHW() : Dims4() {}
    DimsNCHW(int batchSize, int channels, int height, int width) : Dims4(batchSize, channels, height, width) {}
    int& n() { return d[0]; }
    int n() const { return d[0]; }
    int& c() { return d[1]; }
    int c() const { return d[1]; }
    int& h() { return d[2]; }
    int h() const { return d[2]; }
    int& w() { return d[3]; }
    int w() const { return d[3]; }
};

struct Permutation { int order[Dims::MAX_DIMS]; };

class Weights {
public:
    DataType type;
    const void* values;
    int64_t count;
};

class IHostMemory {
public:
    virtual void* data() const = 0;
    virtual std::size_t size() const = 0;
    virtual DataType type() const = 0;
    virtual void destroy() = 0;
protected:
    virtual ~IHostMemory() {}
};

class ILogger {
public:
    enum class Severity { kINTERNAL_ERROR = 0, kERROR = 1, kWARNING = 2, kINFO = 3 };
    virtual void log(Severity severity, const char* msg) = 0;
    virtual ~ILogger() {}
};
template <> inline int EnumMax<ILogger::Severity>() { return 4; }

class IProfiler {
public:
    virtual void reportLayerTime(const char* layerName, float ms) = 0;
    virtual ~IProfiler() {}
};

// ---- network definition -----------------------------------------------------------------------------
enum class LayerType : int {
    kCONVOLUTION = 0, kFULLY_CONNECTED = 1, kACTIVATION = 2, kPOOLING = 3, kLRN = 4, kSCALE = 5, kSOFTMAX = 6,
    kDECONVOLUTION = 7, kCONCATENATION = 8, kELEMENTWISE = 9, kPLUGIN = 10, kRNN = 11, kUNARY = 12,
    kPADDING = 13, kSHUFFLE = 14
};
enum class ActivationType : int { kRELU = 0, kSIGMOID = 1, kTANH = 2 };
enum class ScaleMode : int { kUNIFORM = 0, kCHANNEL = 1, kELEMENTWISE = 2 };
enum class ElementWiseOperation : int { kSUM = 0, kPROD = 1, kMAX = 2, kMIN = 3, kSUB = 4, kDIV = 5, kPOW = 6 };
enum class PluginFormat : uint8_t { kNCHW = 0, kNC2HW2 = 1, kNHWC8 = 2 };
template <> inline int EnumMax<PluginFormat>() { return 3; }

class ILayer;

class ITensor {
public:
    virtual void setName(const char* name) = 0;
    virtual const char* getName() const = 0;
    virtual void setDimensions(Dims dimensions) = 0;
    virtual Dims getDimensions() const = 0;
    virtual DataType getType() const = 0;
    virtual bool isNetworkInput() const = 0;
    virtual bool isNetworkOutput() const = 0;
protected:
    virtual ~ITensor() {}
};

class ILayer {
public:
    virtual LayerType getType() const = 0;
    virtual void setName(const char* name) = 0;
    virtual const char* getName() const = 0;
    virtual int getNbInputs() const = 0;
    virtual ITensor* getInput(int index) const = 0;
    virtual int getNbOutputs() const = 0;
    virtual ITensor* getOutput(int index) const = 0;
protected:
    virtual ~ILayer() {}
};

class IConvolutionLayer : public ILayer {
public:
    virtual void setKernelSize(DimsHW kernelSize) = 0;
    virtual DimsHW getKernelSize() const = 0;
    virtual void setNbOutputMaps(int nbOutputMaps) = 0;
    virtual int getNbOutputMaps() const = 0;
    virtual void setStride(DimsHW stride) = 0;
    virtual DimsHW getStride() const = 0;
    virtual void setPadding(DimsHW padding) = 0;
    virtual DimsHW getPadding() const = 0;
    virtual void setKernelWeights(Weights weights) = 0;
    virtual Weights getKernelWeights() const = 0;
    virtual void setBiasWeights(Weights weights) = 0;
    virtual Weights getBiasWeights() const = 0;
protected:
    virtual ~IConvolutionLayer() {}
};
class IDeconvolutionLayer : public ILayer {
public:
    virtual void setKernelSize(DimsHW kernelSize) = 0;
    virtual DimsHW getKernelSize() const = 0;
    virtual void setNbOutputMaps(int nbOutputMaps) = 0;
    virtual int getNbOutputMaps() const = 0;
    virtual void setStride(DimsHW stride) = 0;
    virtual DimsHW getStride() const = 0;
    virtual void setPadding(DimsHW padding) = 0;
    virtual DimsHW getPadding() const = 0;
    virtual void setKernelWeights(Weights weights) = 0;
    virtual Weights getKernelWeights() const = 0;
    virtual void setBiasWeights(Weights weights) = 0;
    virtual Weights getBiasWeights() const = 0;
protected:
    virtual ~IDeconvolutionLayer() {}
};
class IActivationLayer : public ILayer {
public:
    virtual void setActivationType(ActivationType type) = 0;
    virtual ActivationType getActivationType() const = 0;
protected:
    virtual ~IActivationLayer() {}
};
class IScaleLayer : public ILayer {
public:
    virtual ScaleMode getMode() const = 0;
    virtual Weights getShift() const = 0;
    virtual Weights getScale() const = 0;
    virtual Weights getPower() const = 0;
protected:
    virtual ~IScaleLayer() {}
};
class IElementWiseLayer : public ILayer {
public:
    virtual void setOperation(ElementWiseOperation type) = 0;
    virtual ElementWiseOperation getOperation() const = 0;
protected:
    virtual ~IElementWiseLayer() {}
};
class IConcatenationLayer : public ILayer {
public:
    virtual void setAxis(int axis) = 0;
    virtual int getAxis() const = 0;
protected:
    virtual ~IConcatenationLayer() {}
};
class IPaddingLayer : public ILayer {
public:
    virtual DimsHW getPrePadding() const = 0;
    virtual DimsHW getPostPadding() const = 0;
protected:
    virtual ~IPaddingLayer() {}
};
class IShuffleLayer : public ILayer {
public:
    virtual void setFirstTranspose(Permutation permutation) = 0;
    virtual void setReshapeDimensions(Dims dimensions) = 0;
    virtual Dims getReshapeDimensions() const = 0;
    virtual void setSecondTranspose(Permutation permutation) = 0;
protected:
    virtual ~IShuffleLayer() {}
};

// ---- plugins (the boundary the reference's stereoDNN/lib implements) -----------------------------------
class IPlugin {
public:
    virtual int getNbOutputs() const = 0;
    virtual Dims getOutputDimensions(int index, const Dims* inputs, int nbInputDims) = 0;
    virtual void configure(const Dims* inputDims, int nbInputs, const Dims* outputDims, int nbOutputs, int maxBatchSize) = 0;
    virtual int initialize() = 0;
    virtual void terminate() = 0;
    virtual size_t getWorkspaceSize(int maxBatchSize) const = 0;
    virtual int enqueue(int batchSize, const void* const* inputs, void** outputs, void* workspace, cudaStream_t stream) = 0;
    virtual size_t getSerializationSize() = 0;
    virtual void serialize(void* buffer) = 0;
protected:
    virtual ~IPlugin() {}
};

class IPluginExt : public IPlugin {
public:
    virtual int getTensorRTVersion() const { return NV_TENSORRT_MAJOR * 1000 + NV_TENSORRT_MINOR * 100 + NV_TENSORRT_PATCH; }
    virtual bool supportsFormat(DataType type, PluginFormat format) const = 0;
    virtual void configureWithFormat(const Dims* inputDims, int nbInputs, const Dims* outputDims, int nbOutputs,
                                     DataType type, PluginFormat format, int maxBatchSize) = 0;
protected:
    void configure(const Dims*, int, const Dims*, int, int) final {}
    virtual ~IPluginExt() {}
};

class IPluginLayer : public ILayer {
public:
    virtual IPlugin& getPlugin() = 0;
protected:
    virtual ~IPluginLayer() {}
};

class IPluginFactory {
public:
    virtual IPlugin* createPlugin(const char* layerName, const void* serialData, size_t serialLength) = 0;
    virtual ~IPluginFactory() {}
};

class INetworkDefinition {
public:
    virtual ITensor* addInput(const char* name, DataType type, Dims dimensions) = 0;
    virtual void markOutput(ITensor& tensor) = 0;
    virtual IConvolutionLayer* addConvolution(ITensor& input, int nbOutputMaps, DimsHW kernelSize, Weights kernelWeights, Weights biasWeights) = 0;
    virtual IDeconvolutionLayer* addDeconvolution(ITensor& input, int nbOutputMaps, DimsHW kernelSize, Weights kernelWeights, Weights biasWeights) = 0;
    virtual IActivationLayer* addActivation(ITensor& input, ActivationType type) = 0;
    virtual IScaleLayer* addScale(ITensor& input, ScaleMode mode, Weights shift, Weights scale, Weights power) = 0;
    virtual IElementWiseLayer* addElementWise(ITensor& input1, ITensor& input2, ElementWiseOperation op) = 0;
    virtual IConcatenationLayer* addConcatenation(ITensor* const* inputs, int nbInputs) = 0;
    virtual IPaddingLayer* addPadding(ITensor& input, DimsHW prePadding, DimsHW postPadding) = 0;
    virtual IShuffleLayer* addShuffle(ITensor& input) = 0;
    virtual IPluginLayer* addPlugin(ITensor* const* inputs, int nbInputs, IPlugin& plugin) = 0;
    virtual IPluginLayer* addPluginExt(ITensor* const* inputs, int nbInputs, IPluginExt& plugin) = 0;
    virtual int getNbLayers() const = 0;
    virtual ILayer* getLayer(int index) const = 0;
    virtual int getNbInputs() const = 0;
    virtual ITensor* getInput(int index) const = 0;
    virtual int getNbOutputs() const = 0;
    virtual ITensor* getOutput(int index) const = 0;
    virtual void destroy() = 0;
protected:
    virtual ~INetworkDefinition() {}
};

// ---- engine / execution ----------------------------------------------------------------------------------
class ICudaEngine;

class IExecutionContext {
public:
    virtual bool execute(int batchSize, void** bindings) = 0;
    virtual bool enqueue(int batchSize, void** bindings, cudaStream_t stream, cudaEvent_t* inputConsumed) = 0;
    virtual void setDebugSync(bool sync) = 0;
    virtual bool getDebugSync() const = 0;
    virtual void setProfiler(IProfiler*) = 0;
    virtual IProfiler* getProfiler() const = 0;
    virtual const ICudaEngine& getEngine() const = 0;
    virtual void destroy() = 0;
    // Extension (not in TensorRT): HIP streams this context issues its launches on.  2 (default) = independent branches of the
    // network -- the right-image encoder -- run on a second stream: best for ONE context (latency).  1 = everything on the
    // caller's stream, no cross-stream events: best when several contexts are kept busy side by side (throughput; measured on
    // MI355X, ResNet-18 2D 1257x369: six one-stream contexts 2270 pairs/s, four two-stream contexts 2100).
    virtual void setExecutionStreams(int) {}
    virtual int getExecutionStreams() const { return 2; }
    // Extension: graph mode.  The second execute() / enqueue() with the same bindings, batch size and stream is captured as a
    // hipGraph and every further one replays it with a single graph launch (a change of bindings captures another graph).
    // Off by default: measured on MI355X the 25 launches of ResNet-18 2D cost the host 70-95 us, well under the 0.55 ms the GPU needs.
    virtual void setGraphMode(bool) {}
    virtual bool getGraphMode() const { return false; }
    // Extension: launch trace (a debugging aid).  While it is on, every launch's output tensor is hashed on the launch's own stream
    // (rt_hash_buffer) right after the launch; readLaunchTrace() waits for the pass and returns one 64-bit value per launch, in launch
    // order, so two passes over the same input can be compared launch by launch.  getLaunchName / readLaunchOutput name a launch and copy
    // its output tensor (as stored: fp32 / fp16, planar / interleaved, pitched) to the host.
    virtual void setLaunchTrace(bool) {}
    virtual int readLaunchTrace(unsigned long long* /*hashes*/, int /*max*/) { return 0; }
    virtual const char* getLaunchName(int) const { return nullptr; }
    virtual long long readLaunchOutput(int /*launch*/, void* /*host*/, long long /*bytes*/) { return -1; }
protected:
    virtual ~IExecutionContext() {}
};

class ICudaEngine {
public:
    virtual int getNbBindings() const = 0;
    virtual int getBindingIndex(const char* name) const = 0;
    virtual const char* getBindingName(int bindingIndex) const = 0;
    virtual bool bindingIsInput(int bindingIndex) const = 0;
    virtual Dims getBindingDimensions(int bindingIndex) const = 0;
    virtual DataType getBindingDataType(int bindingIndex) const = 0;
    virtual int getMaxBatchSize() const = 0;
    virtual int getNbLayers() const = 0;
    virtual std::size_t getWorkspaceSize() const = 0;
    virtual IHostMemory* serialize() const = 0;
    virtual IExecutionContext* createExecutionContext() = 0;
    virtual void destroy() = 0;
protected:
    virtual ~ICudaEngine() {}
};

class IBuilder {
public:
    virtual INetworkDefinition* createNetwork() = 0;
    virtual void setMaxBatchSize(int batchSize) = 0;
    virtual int getMaxBatchSize() const = 0;
    virtual void setMaxWorkspaceSize(std::size_t workspaceSize) = 0;
    virtual std::size_t getMaxWorkspaceSize() const = 0;
    virtual void setHalf2Mode(bool mode) = 0;
    virtual bool getHalf2Mode() const = 0;
    virtual void setDebugSync(bool sync) = 0;
    virtual bool getDebugSync() const = 0;
    virtual void setMinFindIterations(int minFind) = 0;
    virtual int getMinFindIterations() const = 0;
    virtual void setAverageFindIterations(int avgFind) = 0;
    virtual int getAverageFindIterations() const = 0;
    virtual bool platformHasFastFp16() const = 0;
    virtual bool platformHasFastInt8() const = 0;
    virtual ICudaEngine* buildCudaEngine(INetworkDefinition& network) = 0;
    virtual void destroy() = 0;
    // Extension (not in TensorRT): keep every 2-D convolution of the engine on the fp32 fmaf-chain kernels instead of the default 3-term
    // fp16 split on the fp16 matrix pipe (rtConv2dDesc::flags = RT_CONV_EXACT_FP32; 1610 vs 2380 pairs/s at 1257x369 on MI355X).
    virtual void setExactFp32Mode(bool) {}
    virtual bool getExactFp32Mode() const { return false; }
protected:
    virtual ~IBuilder() {}
};

class IRuntime {
public:
    virtual ICudaEngine* deserializeCudaEngine(const void* blob, std::size_t size, IPluginFactory* pluginFactory) = 0;
    virtual void destroy() = 0;
protected:
    virtual ~IRuntime() {}
};

IBuilder* createInferBuilder(ILogger& logger);
IRuntime* createInferRuntime(ILogger& logger);

}  // namespace nvinfer1

#endif  // REDTAIL_AMD_NVINFER_SHIM_H
