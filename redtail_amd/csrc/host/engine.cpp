// Static-graph builder and fusing executor behind the NvInfer.h shim.
//
// TensorRT is the (closed-source) runtime the reference plugs into; on MI355X this file takes its
// place for the Stereo DNN graphs: INetworkDefinition records layers, buildCudaEngine() runs shape
// inference + the plugin life cycle (getOutputDimensions -> supportsFormat -> configure[WithFormat]
// -> getWorkspaceSize, as listed in SURVEY.md 8b), lowers the graph to a short list of kernel
// launches and fuses what TensorRT could not fuse across a plugin boundary:
//     conv/deconv (+ residual add) (+ ELU plugin | sigmoid)  ->  one MFMA kernel with fused epilogue
//     correlation cost volume + soft-argmax                    ->  one kernel, volume never in HBM
//     identity scale / reshape                                 ->  aliases
//     Conv3D + Transform(+ELU), Pad before Conv3D, Conv3DTranspose + Slice (+add +ELU) -> one kernel
// Independent branches (the left and right encoders) are issued on two HIP streams.
// Every launch goes through the C ABI of include/rt_stereo.h; there is no CPU compute path.
#include <algorithm>
#include <cassert>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <map>
#include <set>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "NvInfer.h"
#include "plugin_internal.h"
#include "rt_stereo.h"

namespace nvinfer1 {
namespace {

// Development knobs (RT_* environment variables, the A/B switches of tests / tools / DESIGN.md's measurements) are honoured only in a
// process that opts in with RT_DEV_KNOBS=1 (tests/conftest.py, tools/): ambient environment does not change what the library does.
const char* knob(const char* name) {
    static const bool on = [] { const char* e = std::getenv("RT_DEV_KNOBS"); return e && std::atoi(e) != 0; }();
    return on ? std::getenv(name) : nullptr;
}

using redtail::tensorrt::CostVolumeType;
using redtail::tensorrt::SoftargmaxType;
using redtail::tensorrt::internal::ConvFusion;
using redtail::tensorrt::internal::IStereoPlugin;
using redtail::tensorrt::internal::Kind;

size_t volume(const Dims& d) {
    size_t n = 1;
    for (int i = 0; i < d.nbDims; i++) n *= (size_t)d.d[i];
    return n;
}

struct LayerData;

// ---- tensors -----------------------------------------------------------------------------------------
struct TensorImpl : public ITensor {
    std::string name;
    Dims dims{};
    int id = -1;
    LayerData* producer = nullptr;
    int producer_out = 0;
    bool is_input = false, is_output = false;
    // executor state
    int alias_of = -1;          // shares the buffer of another tensor (identity scale, reshape)
    int64_t alias_off = 0;      // element offset inside the aliased buffer (per sample)
    int64_t bstride = 0;        // per-sample stride in elements (0 = dense volume)
    void* dev = nullptr;
    int stream = 0;             // which of the two execution streams produces it
    int pitch = 0;              // row pitch in elements of an internal (C,H,W) tensor, 0 = dense
    bool f16 = false;           // stored as fp16 (half2 mode); only tensors that only layout-aware launches touch
    bool il8 = false;           // channel-interleaved, (C/8, H, pitch, 8) in fp16, (C/4, H, pitch, 4) in fp32: only layout-aware launches touch it
    bool split = false;         // fp32 il8 tensor stored PRE-SPLIT, (C/8, H, pitch, [8 x fp16 hi | 8 x fp16 lo]): between two tower blocks (fuseResBlocks)
    int cpad = 0;               // channels ALLOCATED per sample when > dims.d[0]: an interleaved tensor that hosts a folded concatenation
                                // (its own channels, then the other members' groups); the sample stride of everything that touches it
    // siamese towers (mergeSiamese): this right-tower tensor lives in the second half of its left-tower twin's buffer -- `batch`
    // samples after the twin's first sample, so that one launch over 2 * batch samples serves both towers
    int twin_of = -1;
    bool has_twin = false;      // the left-tower tensor of such a pair: its buffer holds 2 * maxBatchSize samples

    void setName(const char* n) override { name = n ? n : ""; }
    const char* getName() const override { return name.c_str(); }
    void setDimensions(Dims d) override { dims = d; }
    Dims getDimensions() const override { return dims; }
    DataType getType() const override { return DataType::kFLOAT; }
    bool isNetworkInput() const override { return is_input; }
    bool isNetworkOutput() const override { return is_output; }
};

// ---- layers ------------------------------------------------------------------------------------------
struct LayerData {
    LayerType type;
    std::string name;
    std::vector<TensorImpl*> in, out;
    // convolution / deconvolution
    int nb_maps = 0;
    DimsHW ksize{1, 1}, stride{1, 1}, padding{0, 0};
    Weights kernel{DataType::kFLOAT, nullptr, 0}, bias{DataType::kFLOAT, nullptr, 0};
    // others
    ActivationType act = ActivationType::kRELU;
    ScaleMode scale_mode = ScaleMode::kUNIFORM;
    Weights shift{DataType::kFLOAT, nullptr, 0}, scale{DataType::kFLOAT, nullptr, 0}, power{DataType::kFLOAT, nullptr, 0};
    ElementWiseOperation ew = ElementWiseOperation::kSUM;
    Dims reshape{};
    DimsHW pre_pad{0, 0}, post_pad{0, 0};
    IPlugin* plugin = nullptr;
    bool plugin_ext = false;
    virtual void recomputeShapes() {}
    virtual ~LayerData() {}
};

template <typename Iface>
struct LayerImpl : public Iface, public LayerData {
    LayerType getType() const override { return type; }
    void setName(const char* n) override { name = n ? n : ""; }
    const char* getName() const override { return name.c_str(); }
    int getNbInputs() const override { return (int)in.size(); }
    ITensor* getInput(int i) const override { return i >= 0 && i < (int)in.size() ? in[i] : nullptr; }
    int getNbOutputs() const override { return (int)out.size(); }
    ITensor* getOutput(int i) const override { return i >= 0 && i < (int)out.size() ? out[i] : nullptr; }
};

template <typename Iface, bool kDeconv>
struct ConvLayerImpl : public LayerImpl<Iface> {
    void setKernelSize(DimsHW k) override { this->ksize = k; recomputeShapes(); }
    DimsHW getKernelSize() const override { return this->ksize; }
    void setNbOutputMaps(int n) override { this->nb_maps = n; recomputeShapes(); }
    int getNbOutputMaps() const override { return this->nb_maps; }
    void setStride(DimsHW s) override { this->stride = s; recomputeShapes(); }
    DimsHW getStride() const override { return this->stride; }
    void setPadding(DimsHW p) override { this->padding = p; recomputeShapes(); }
    DimsHW getPadding() const override { return this->padding; }
    void setKernelWeights(Weights w) override { this->kernel = w; }
    Weights getKernelWeights() const override { return this->kernel; }
    void setBiasWeights(Weights w) override { this->bias = w; }
    Weights getBiasWeights() const override { return this->bias; }
    void recomputeShapes() override {
        const Dims& x = this->in[0]->dims;
        int ho, wo;
        if (kDeconv) {
            ho = (x.d[1] - 1) * this->stride.h() - 2 * this->padding.h() + this->ksize.h();
            wo = (x.d[2] - 1) * this->stride.w() - 2 * this->padding.w() + this->ksize.w();
        } else {
            ho = (x.d[1] + 2 * this->padding.h() - this->ksize.h()) / this->stride.h() + 1;
            wo = (x.d[2] + 2 * this->padding.w() - this->ksize.w()) / this->stride.w() + 1;
        }
        this->out[0]->dims = DimsCHW(this->nb_maps, ho, wo);
    }
};
struct ActivationImpl : public LayerImpl<IActivationLayer> {
    void setActivationType(ActivationType t) override { act = t; }
    ActivationType getActivationType() const override { return act; }
};
struct ScaleImpl : public LayerImpl<IScaleLayer> {
    ScaleMode getMode() const override { return scale_mode; }
    Weights getShift() const override { return shift; }
    Weights getScale() const override { return scale; }
    Weights getPower() const override { return power; }
};
struct ElementWiseImpl : public LayerImpl<IElementWiseLayer> {
    void setOperation(ElementWiseOperation o) override { ew = o; }
    ElementWiseOperation getOperation() const override { return ew; }
};
struct ConcatImpl : public LayerImpl<IConcatenationLayer> {
    void setAxis(int) override {}
    int getAxis() const override { return 0; }
};
struct PaddingImpl : public LayerImpl<IPaddingLayer> {
    DimsHW getPrePadding() const override { return pre_pad; }
    DimsHW getPostPadding() const override { return post_pad; }
};
struct ShuffleImpl : public LayerImpl<IShuffleLayer> {
    void setFirstTranspose(Permutation) override {}
    void setSecondTranspose(Permutation) override {}
    void setReshapeDimensions(Dims d) override {
        assert(volume(d) == volume(in[0]->dims) && "reshape must preserve the volume");
        reshape = d;
        out[0]->dims = d;
    }
    Dims getReshapeDimensions() const override { return reshape; }
};
struct PluginLayerImpl : public LayerImpl<IPluginLayer> {
    IPlugin& getPlugin() override { return *plugin; }
};

// ---- network -----------------------------------------------------------------------------------------
class NetworkImpl : public INetworkDefinition {
public:
    explicit NetworkImpl(ILogger& log) : log_(log) {}
    ~NetworkImpl() override {}

    ITensor* addInput(const char* name, DataType, Dims dims) override {
        TensorImpl* t = newTensor(name);
        t->dims = dims;
        t->is_input = true;
        inputs_.push_back(t);
        return t;
    }
    void markOutput(ITensor& t) override {
        auto* ti = static_cast<TensorImpl*>(&t);
        ti->is_output = true;
        outputs_.push_back(ti);
    }
    IConvolutionLayer* addConvolution(ITensor& x, int maps, DimsHW k, Weights w, Weights b) override {
        auto* l = add<ConvLayerImpl<IConvolutionLayer, false>>(LayerType::kCONVOLUTION, {&x}, 1);
        l->nb_maps = maps; l->ksize = k; l->kernel = w; l->bias = b;
        l->recomputeShapes();
        return l;
    }
    IDeconvolutionLayer* addDeconvolution(ITensor& x, int maps, DimsHW k, Weights w, Weights b) override {
        auto* l = add<ConvLayerImpl<IDeconvolutionLayer, true>>(LayerType::kDECONVOLUTION, {&x}, 1);
        l->nb_maps = maps; l->ksize = k; l->kernel = w; l->bias = b;
        l->recomputeShapes();
        return l;
    }
    IActivationLayer* addActivation(ITensor& x, ActivationType t) override {
        auto* l = add<ActivationImpl>(LayerType::kACTIVATION, {&x}, 1);
        l->act = t;
        l->out[0]->dims = l->in[0]->dims;
        return l;
    }
    IScaleLayer* addScale(ITensor& x, ScaleMode mode, Weights shift, Weights scale, Weights power) override {
        auto* l = add<ScaleImpl>(LayerType::kSCALE, {&x}, 1);
        l->scale_mode = mode; l->shift = shift; l->scale = scale; l->power = power;
        l->out[0]->dims = l->in[0]->dims;
        return l;
    }
    IElementWiseLayer* addElementWise(ITensor& a, ITensor& b, ElementWiseOperation op) override {
        auto* l = add<ElementWiseImpl>(LayerType::kELEMENTWISE, {&a, &b}, 1);
        l->ew = op;
        assert(volume(l->in[0]->dims) == volume(l->in[1]->dims));
        l->out[0]->dims = l->in[0]->dims;
        return l;
    }
    IConcatenationLayer* addConcatenation(ITensor* const* ins, int n) override {
        std::vector<ITensor*> v(ins, ins + n);
        auto* l = add<ConcatImpl>(LayerType::kCONCATENATION, v, 1);
        Dims d = l->in[0]->dims;
        for (int i = 1; i < n; i++) d.d[0] += l->in[i]->dims.d[0];
        l->out[0]->dims = d;
        return l;
    }
    IPaddingLayer* addPadding(ITensor& x, DimsHW pre, DimsHW post) override {
        auto* l = add<PaddingImpl>(LayerType::kPADDING, {&x}, 1);
        l->pre_pad = pre; l->post_pad = post;
        Dims d = l->in[0]->dims;
        d.d[1] += pre.h() + post.h();
        d.d[2] += pre.w() + post.w();
        l->out[0]->dims = d;
        return l;
    }
    IShuffleLayer* addShuffle(ITensor& x) override {
        auto* l = add<ShuffleImpl>(LayerType::kSHUFFLE, {&x}, 1);
        l->out[0]->dims = l->in[0]->dims;
        l->reshape = l->in[0]->dims;
        return l;
    }
    IPluginLayer* addPlugin(ITensor* const* ins, int n, IPlugin& plugin) override { return addPluginImpl(ins, n, plugin, false); }
    IPluginLayer* addPluginExt(ITensor* const* ins, int n, IPluginExt& plugin) override { return addPluginImpl(ins, n, plugin, true); }

    int getNbLayers() const override { return (int)layers_.size(); }
    ILayer* getLayer(int i) const override { return ilayers_[i]; }
    int getNbInputs() const override { return (int)inputs_.size(); }
    ITensor* getInput(int i) const override { return inputs_[i]; }
    int getNbOutputs() const override { return (int)outputs_.size(); }
    ITensor* getOutput(int i) const override { return outputs_[i]; }
    void destroy() override { delete this; }

    // the engine takes ownership of the graph
    std::vector<std::unique_ptr<TensorImpl>> tensors_;
    std::vector<std::unique_ptr<LayerData>> layers_;
    std::vector<ILayer*> ilayers_;
    std::vector<TensorImpl*> inputs_, outputs_;
    ILogger& log_;

private:
    TensorImpl* newTensor(const char* name) {
        tensors_.emplace_back(new TensorImpl());
        TensorImpl* t = tensors_.back().get();
        t->id = (int)tensors_.size() - 1;
        t->name = name ? name : "";
        return t;
    }
    template <typename L>
    L* add(LayerType type, std::vector<ITensor*> ins, int nout) {
        L* l = new L();
        l->type = type;
        l->name = "(Unnamed Layer* " + std::to_string(layers_.size()) + ")";
        for (ITensor* t : ins) l->in.push_back(static_cast<TensorImpl*>(t));
        for (int i = 0; i < nout; i++) {
            TensorImpl* o = newTensor((l->name + "_output_" + std::to_string(i)).c_str());
            o->producer = l;
            o->producer_out = i;
            l->out.push_back(o);
        }
        layers_.emplace_back(l);
        ilayers_.push_back(l);
        return l;
    }
    IPluginLayer* addPluginImpl(ITensor* const* ins, int n, IPlugin& plugin, bool ext) {
        std::vector<ITensor*> v(ins, ins + n);
        const int nout = plugin.getNbOutputs();
        auto* l = add<PluginLayerImpl>(LayerType::kPLUGIN, v, nout);
        l->plugin = &plugin;
        l->plugin_ext = ext;
        std::vector<Dims> in_dims;
        for (auto* t : l->in) in_dims.push_back(t->dims);
        for (int i = 0; i < nout; i++) l->out[i]->dims = plugin.getOutputDimensions(i, in_dims.data(), (int)in_dims.size());
        return l;
    }
};

// ---- lowered program -----------------------------------------------------------------------------------
enum class OpKind { kConv, kPlugin, kConv3D, kAdd, kAct, kConcat, kCorrSoftargmax, kCopy };

struct Op {
    OpKind kind;
    std::string name;                  // layer name reported to the profiler (first layer of a fused group)
    std::vector<int> in;               // tensor ids
    int out = -1;
    int resid = -1;
    int act = RT_ACT_NONE;
    rtConvPlan* plan = nullptr;        // kConv
    const struct LayerData* conv_layer = nullptr;   // kConv: the (de)convolution layer the plan was made from
    IPlugin* plugin = nullptr;         // kPlugin / kConv3D
    IStereoPlugin* splugin = nullptr;  // kConv3D
    int max_disp = 0, is_min = 0;      // kCorrSoftargmax
    bool il_in = false;                // kCorrSoftargmax: channel-interleaved feature maps -> matrix-core kernel
    bool il_out = false;               // kCorrSoftargmax: the map is lane 0 of an interleaved group (member of an interleaved concatenation)
    int half_kind = 0;                 // kPlugin created for kHALF: 1 = fp16 NCHW, 2 = fp16 NC2HW2 (tensors are converted around it)
    int stream = 0;
    std::vector<int> wait_on;          // tensors produced on the other stream that this op consumes
    bool publish = false;              // another stream consumes the result: record an event after the launch
    std::vector<int> lays;             // indices (layers_) of the layers this launch covers, in fusion order
    // merged siamese launch (mergeSiamese): the right-tower op this launch also executes, as the second half of a 2 * batch launch
    bool twin = false;
    std::vector<int> twin_in;          // right-tower tensors it reads (inputs, residual): dependencies for the stream schedule
    int twin_out = -1;                 // right-tower tensor it writes
    int softarg = 0;                   // kConv3D: the launch ends in the soft-argmax (1) / soft-argmin (2) that followed it (fuseSoftargmax3D)
    bool twin_bind = false;            // ... whose INPUT is the other input binding (first layers: rt_conv_enqueue_twin_input), not a twin-placed tensor
};

class EngineImpl;

class ContextImpl : public IExecutionContext {
public:
    explicit ContextImpl(EngineImpl& e);
    ~ContextImpl() override;
    bool execute(int batchSize, void** bindings) override;
    bool enqueue(int batchSize, void** bindings, cudaStream_t stream, cudaEvent_t* inputConsumed) override;
    void setDebugSync(bool s) override { debug_sync_ = s; }
    bool getDebugSync() const override { return debug_sync_; }
    void setExecutionStreams(int n) override { streams_ = n < 2 ? 1 : 2; }
    int getExecutionStreams() const override { return streams_; }
    void setGraphMode(bool on) override { graph_mode_ = on; }
    bool getGraphMode() const override { return graph_mode_; }
    void setProfiler(IProfiler* p) override { profiler_ = p; }
    IProfiler* getProfiler() const override { return profiler_; }
    const ICudaEngine& getEngine() const override;
    void destroy() override { delete this; }
    void setLaunchTrace(bool on) override { trace_ = on; }
    int readLaunchTrace(unsigned long long* hashes, int max) override;
    const char* getLaunchName(int i) const override;
    long long readLaunchOutput(int launch, void* host, long long bytes) override;

private:
    bool run(int batch, void** bindings, cudaStream_t stream, bool sync);
    bool issue(int batch, void** bindings, cudaStream_t stream, bool sync, bool capturing);
    bool checkInputRange(const Op& op, int batch, void** bindings, rtStream st);
    bool ensureBuffers(int batch);
    void* addr(int tensor, int batch, void** bindings) const;

    EngineImpl& eng_;
    IProfiler* profiler_ = nullptr;
    bool debug_sync_ = false;
    int streams_ = 2;
    int alloc_batch_ = 0;
    std::vector<void*> buffers_;       // per tensor id (owned)
    void* workspace_ = nullptr;        // 2 x workspace_bytes_: the launches of the main / the side stream never share scratch
    size_t workspace_bytes_ = 0;
    void* wsOf(bool side) const { return workspace_ ? static_cast<char*>(workspace_) + (side ? workspace_bytes_ : 0) : nullptr; }
    rtStream side_stream_ = nullptr;
    rtStream main_stream_ = nullptr;   // execute() runs on its own stream, never on the NULL stream
    std::vector<void*> events_;        // per tensor id, lazily created
    std::vector<void*> prof_events_;   // start/stop pair per launch (IProfiler)
    void* ev_fork_ = nullptr;
    void* ev_join_ = nullptr;
    void* ev_null_ = nullptr;
    // graph mode: one captured pass per (stream, batch, schedule, bindings); second == nullptr: seen once, captured on the next call
    bool graph_mode_ = false;
    std::map<std::vector<uintptr_t>, rtGraph*> graphs_;
    std::set<std::vector<uintptr_t>> graph_failed_;
    std::map<std::pair<int, int>, std::pair<void*, size_t>> half_buf_;   // (op, slot) -> fp16 staging buffer of a kHALF plugin
    void* halfBuffer(int op, int slot, size_t bytes);
    // launch trace (setLaunchTrace): one device word per launch, the stream and the extent of the last pass's outputs
    bool trace_ = false;
    void* trace_dev_ = nullptr;
    rtStream trace_stream_ = nullptr;
    int trace_batch_ = 0;
    std::vector<void*> trace_ptr_;
    size_t outBytes(const Op& op, int batch) const;
    void dropGraphs();
};

class HostMemory : public IHostMemory {
public:
    std::string blob;
    void* data() const override { return const_cast<char*>(blob.data()); }
    std::size_t size() const override { return blob.size(); }
    DataType type() const override { return DataType::kINT8; }
    void destroy() override { delete this; }
};

class EngineImpl : public ICudaEngine {
public:
    EngineImpl(NetworkImpl& net, int max_batch, bool half2, ILogger& log, bool exact_fp32 = false);
    ~EngineImpl() override;
    bool ok() const { return ok_; }

    int getNbBindings() const override { return (int)bindings_.size(); }
    int getBindingIndex(const char* name) const override {
        for (size_t i = 0; i < bindings_.size(); i++)
            if (tensors_[bindings_[i]]->name == name) return (int)i;
        return -1;
    }
    const char* getBindingName(int i) const override { return tensors_[bindings_[i]]->name.c_str(); }
    bool bindingIsInput(int i) const override { return tensors_[bindings_[i]]->is_input; }
    Dims getBindingDimensions(int i) const override { return tensors_[bindings_[i]]->dims; }
    DataType getBindingDataType(int) const override { return DataType::kFLOAT; }
    int getMaxBatchSize() const override { return max_batch_; }
    int getNbLayers() const override { return (int)ops_.size(); }
    std::size_t getWorkspaceSize() const override { return 2 * workspace_bytes_; }      // what a context allocates: one block per execution stream
    IHostMemory* serialize() const override;
    IExecutionContext* createExecutionContext() override {
        for (auto& l : layers_)
            if (l->plugin && l->plugin->initialize() != 0)
                log_.log(ILogger::Severity::kERROR, (l->name + ": plugin initialize() failed").c_str());
        return new ContextImpl(*this);
    }
    void destroy() override { delete this; }

    // graph (moved out of the network)
    std::vector<std::unique_ptr<TensorImpl>> tensors_;
    std::vector<std::unique_ptr<LayerData>> layers_;
    std::vector<int> bindings_;        // tensor ids: inputs in declaration order, then outputs
    std::vector<Op> ops_;
    int max_batch_;
    size_t workspace_bytes_ = 0;
    bool two_streams_ = false;
    std::vector<size_t> issue_sync_;   // op indices in the issue order of execute() (two streams alternating); empty = ops_ order
    ILogger& log_;

    // elements of one sample as the tensor lies in memory (row pitch, channels padded for a hosted concatenation)
    static int64_t sampleElems(const TensorImpl& t) {
        if (t.dims.nbDims != 3) return (int64_t)volume(t.dims);
        return (int64_t)(t.cpad ? t.cpad : t.dims.d[0]) * t.dims.d[1] * (t.pitch ? t.pitch : t.dims.d[2]);
    }

private:
    int root(int t) const {
        while (tensors_[t]->alias_of >= 0) t = tensors_[t]->alias_of;
        return t;
    }
    bool concatFoldable(size_t ci) const;
    void applySampleStrides();
    int consumers(int t) const {
        int n = 0;
        for (auto& l : layers_)
            for (auto* i : l->in)
                if (i->id == t) n++;
        return n + (tensors_[t]->is_output ? 1 : 0);
    }
    LayerData* soleConsumer(int t) const {
        if (consumers(t) != 1 || tensors_[t]->is_output) return nullptr;
        for (auto& l : layers_)
            for (auto* i : l->in)
                if (i->id == t) return l.get();
        return nullptr;
    }
    static IStereoPlugin* stereo(LayerData* l) { return l && l->plugin ? dynamic_cast<IStereoPlugin*>(l->plugin) : nullptr; }
    static bool isKind(LayerData* l, Kind k) { auto* s = stereo(l); return s && s->kind() == k; }
    bool lower();
    void assignStreams();
    void planIssueOrder();
    void assignPitch();
    void foldConcats();
    void foldCostVolumes();
    void assignHalf3D();
    void assignInterleaved3D();
    void fuseSoftargmax3D();
    void fuseResBlocks();
    void mergeSiamese();
    bool ok_ = false;
    bool half2_ = false;
    bool exact_fp32_ = false;      // IBuilder::setExactFp32Mode: rtConv2dDesc::flags |= RT_CONV_EXACT_FP32 on every 2-D plan
    std::vector<IPlugin*> half_plugins_;   // IPluginExt instances that only accept kHALF ...
    std::vector<int> half_kinds_;          // ... and the format they were configured with (1 = NCHW, 2 = NC2HW2)
public:
    std::vector<std::unique_ptr<char[]>> weight_store_;   // weight bytes of a deserialised plan (Weights.values point here)
};

// ---- building ------------------------------------------------------------------------------------------------
EngineImpl::EngineImpl(NetworkImpl& net, int max_batch, bool half2, ILogger& log, bool exact_fp32)
    : max_batch_(max_batch), log_(log), half2_(half2), exact_fp32_(exact_fp32 || (knob("RT_CONV_EXACT_FP32") && atoi(knob("RT_CONV_EXACT_FP32")) != 0)) {
    tensors_ = std::move(net.tensors_);
    layers_ = std::move(net.layers_);
    for (auto* t : net.inputs_) bindings_.push_back(t->id);
    for (auto* t : net.outputs_) bindings_.push_back(t->id);
    net.inputs_.clear();
    net.outputs_.clear();
    net.ilayers_.clear();

    // plugin life cycle (TensorRT build phase)
    for (auto& l : layers_) {
        if (!l->plugin) continue;
        std::vector<Dims> in_dims, out_dims;
        for (auto* t : l->in) in_dims.push_back(t->dims);
        for (auto* t : l->out) out_dims.push_back(t->dims);
        if (l->plugin_ext) {
            auto* ext = static_cast<IPluginExt*>(l->plugin);
            // activations are fp32 NCHW in this build; a plugin created for kHALF is asked for its own type
            DataType type = ext->supportsFormat(DataType::kFLOAT, PluginFormat::kNCHW) ? DataType::kFLOAT : DataType::kHALF;
            PluginFormat fmt = PluginFormat::kNCHW;
            if (!ext->supportsFormat(type, fmt)) fmt = PluginFormat::kNC2HW2;
            // A plugin created for kHALF (the sample application passes its fp16 data type to the ResNet-18 2D builder,
            // sample_app/main.cpp:248, 256-262) is configured with the type it asks for; it may only be used through the fused
            // launches below -- which take whatever storage type the executor gives the tensors -- never through its own enqueue()
            if (type != DataType::kFLOAT) { half_plugins_.push_back(l->plugin); half_kinds_.push_back(fmt == PluginFormat::kNCHW ? 1 : 2); }
            ext->configureWithFormat(in_dims.data(), (int)in_dims.size(), out_dims.data(), (int)out_dims.size(), type, fmt, max_batch_);
        } else {
            l->plugin->configure(in_dims.data(), (int)in_dims.size(), out_dims.data(), (int)out_dims.size(), max_batch_);
        }
        workspace_bytes_ = std::max(workspace_bytes_, l->plugin->getWorkspaceSize(max_batch_));
    }
    ok_ = lower();
    // a kHALF plugin that was not fused away runs on its own enqueue(): the executor converts its fp32 tensors to the format
    // it was configured with and back (TensorRT's reformat layers; reference tests_main.cpp:301-321, 988-1026)
    if (ok_)
        for (auto& op : ops_) {
            if (op.kind != OpKind::kPlugin) continue;
            auto it = std::find(half_plugins_.begin(), half_plugins_.end(), op.plugin);
            if (it != half_plugins_.end()) op.half_kind = half_kinds_[it - half_plugins_.begin()];
        }
    if (ok_) assignStreams();           // preliminary: assignPitch asks whether a concatenation will fold (same-stream producers)
    if (ok_) assignPitch();
    if (ok_) fuseResBlocks();           // after the layouts are known: the streaming kernel serves interleaved tensors
    if (ok_) assignStreams();           // preliminary: foldConcats wants to know which stream an op is on
    if (ok_) foldConcats();
    if (ok_) foldCostVolumes();
    if (ok_) assignHalf3D();
    if (ok_) assignInterleaved3D();
    if (ok_) fuseSoftargmax3D();        // after the layouts: only the depth-walking form of the last layer has the fused reduction
    if (ok_) mergeSiamese();
    if (ok_) applySampleStrides();
    if (ok_) assignStreams();           // final: the passes above removed and merged ops (waits and publish flags are recomputed)
    if (ok_) planIssueOrder();
    // the launches' scratch (IPlugin::getWorkspaceSize) as it is AFTER fusion and layout negotiation -- the factored first Conv3D of the
    // 3-D models only exists once the cost volume is folded into it.  A context keeps one block per stream it issues on.
    if (ok_)
        for (auto& op : ops_)
            if (op.kind == OpKind::kConv3D && op.plugin) workspace_bytes_ = std::max(workspace_bytes_, op.plugin->getWorkspaceSize(max_batch_));
    workspace_bytes_ = (workspace_bytes_ + 255) / 256 * 256;
}

EngineImpl::~EngineImpl() {
    for (auto& op : ops_)
        if (op.plan) rt_conv_plan_destroy(op.plan);
    for (auto& l : layers_)
        if (l->plugin) l->plugin->terminate();
}

static bool isIdentityScale(const LayerData& l) {
    auto val = [](const Weights& w, float dflt) {
        if (w.count == 0 || !w.values) return dflt;
        if (w.type == DataType::kHALF) {
            uint16_t h;
            std::memcpy(&h, w.values, 2);
            // only exact 0 / 1 matter here
            return h == 0 ? 0.f : (h == 0x3c00 ? 1.f : -1234.f);
        }
        float f;
        std::memcpy(&f, w.values, 4);
        return f;
    };
    return l.scale_mode == ScaleMode::kUNIFORM && val(l.shift, 0.f) == 0.f && val(l.scale, 1.f) == 1.f && val(l.power, 1.f) == 1.f;
}

bool EngineImpl::lower() {
    std::vector<bool> done(layers_.size(), false);
    auto index_of = [&](LayerData* l) {
        for (size_t i = 0; i < layers_.size(); i++)
            if (layers_[i].get() == l) return (int)i;
        return -1;
    };
    auto fail = [&](const std::string& msg) {
        log_.log(ILogger::Severity::kERROR, msg.c_str());
        return false;
    };
    const bool fuse = knob("RT_NO_FUSION") == nullptr;

    for (size_t li = 0; li < layers_.size(); li++) {
        if (done[li]) continue;
        LayerData* l = layers_[li].get();
        done[li] = true;
        Op op;
        op.name = l->name;
        op.lays.push_back((int)li);
        for (auto* t : l->in) op.in.push_back(t->id);
        op.out = l->out[0]->id;

        switch (l->type) {
            case LayerType::kSCALE: {
                if (!isIdentityScale(*l)) return fail(l->name + ": only identity scale layers occur in the Stereo DNN graphs");
                l->out[0]->alias_of = l->in[0]->id;
                continue;
            }
            case LayerType::kSHUFFLE: {
                l->out[0]->alias_of = l->in[0]->id;
                continue;
            }
            case LayerType::kCONVOLUTION:
            case LayerType::kDECONVOLUTION: {
                const bool deconv = l->type == LayerType::kDECONVOLUTION;
                const Dims& x = l->in[0]->dims;
                if (l->ksize.h() != l->ksize.w() || l->stride.h() != l->stride.w())
                    return fail(l->name + ": non-square kernels / strides are not used by the Stereo DNN graphs");
                // greedy epilogue fusion: [+ residual] [+ ELU plugin | sigmoid]
                TensorImpl* cur = l->out[0];
                int resid = -1, act = RT_ACT_NONE;
                if (fuse) {
                    LayerData* nx = soleConsumer(cur->id);
                    if (nx && nx->type == LayerType::kELEMENTWISE && nx->ew == ElementWiseOperation::kSUM) {
                        TensorImpl* other = nx->in[0] == cur ? nx->in[1] : nx->in[0];
                        auto same_dims = [](const Dims& a, const Dims& b) {
                            if (a.nbDims != b.nbDims) return false;
                            for (int i = 0; i < a.nbDims; i++)
                                if (a.d[i] != b.d[i]) return false;
                            return true;
                        };
                        // the residual must already exist when the conv runs, and is read with the output's strides
                        if (other != cur && same_dims(other->dims, cur->dims) &&
                            (other->producer == nullptr || index_of(other->producer) < (int)li || done[index_of(other->producer)])) {
                            resid = other->id;
                            { done[index_of(nx)] = true; op.lays.push_back(index_of(nx)); }
                            cur = nx->out[0];
                            nx = soleConsumer(cur->id);
                        }
                    }
                    if (nx && isKind(nx, Kind::kElu)) {
                        act = RT_ACT_ELU;
                        { done[index_of(nx)] = true; op.lays.push_back(index_of(nx)); }
                        cur = nx->out[0];
                    } else if (nx && nx->type == LayerType::kACTIVATION && nx->act == ActivationType::kSIGMOID) {
                        act = RT_ACT_SIGMOID;
                        { done[index_of(nx)] = true; op.lays.push_back(index_of(nx)); }
                        cur = nx->out[0];
                    }
                }
                rtConv2dDesc d{};
                d.Cin = x.d[0]; d.Cout = l->nb_maps; d.Hin = x.d[1]; d.Win = x.d[2];
                d.KH = l->ksize.h(); d.KW = l->ksize.w(); d.stride = l->stride.h();
                d.pad_h = l->padding.h(); d.pad_w = l->padding.w();
                d.act = act; d.has_residual = resid >= 0;
                d.dtype = l->kernel.type == DataType::kHALF ? RT_F16 : RT_F32;
                d.flags = exact_fp32_ ? RT_CONV_EXACT_FP32 : 0;
                const int64_t expect = (int64_t)d.Cin * d.Cout * d.KH * d.KW;
                if (l->kernel.count != expect) return fail(l->name + ": kernel weight count does not match the layer shape");
                if (l->bias.count != 0 && l->bias.count != d.Cout) return fail(l->name + ": bias count does not match");
                int rc = deconv ? rt_deconv2d_plan_create(&op.plan, &d, l->kernel.values, l->bias.count ? l->bias.values : nullptr)
                                : rt_conv2d_plan_create(&op.plan, &d, l->kernel.values, l->bias.count ? l->bias.values : nullptr);
                if (rc) return fail(l->name + ": " + rt_last_error_string());
                op.kind = OpKind::kConv;
                op.conv_layer = l;
                op.resid = resid;
                op.act = act;
                op.out = cur->id;
                ops_.push_back(op);
                continue;
            }
            case LayerType::kELEMENTWISE: {
                if (l->ew != ElementWiseOperation::kSUM) return fail(l->name + ": only kSUM is used by the Stereo DNN graphs");
                op.kind = OpKind::kAdd;
                TensorImpl* cur = l->out[0];
                LayerData* nx = fuse ? soleConsumer(cur->id) : nullptr;
                if (nx && isKind(nx, Kind::kElu)) {
                    op.act = RT_ACT_ELU;
                    { done[index_of(nx)] = true; op.lays.push_back(index_of(nx)); }
                    cur = nx->out[0];
                }
                op.out = cur->id;
                ops_.push_back(op);
                continue;
            }
            case LayerType::kACTIVATION: {
                if (l->act != ActivationType::kSIGMOID) return fail(l->name + ": only sigmoid activations are used");
                op.kind = OpKind::kAct;
                op.act = RT_ACT_SIGMOID;
                ops_.push_back(op);
                continue;
            }
            case LayerType::kCONCATENATION: {
                op.kind = OpKind::kConcat;
                ops_.push_back(op);
                continue;
            }
            case LayerType::kPLUGIN: {
                IStereoPlugin* sp = stereo(l);
                // correlation cost volume feeding only a soft-argmax: one fused kernel
                if (fuse && sp && sp->kind() == Kind::kCostVolume && sp->costVolumeType() == CostVolumeType::kCorrelation &&
                    sp->maxDisparity() <= 64) {
                    LayerData* nx = soleConsumer(l->out[0]->id);
                    if (nx && isKind(nx, Kind::kSoftargmax)) {
                        op.kind = OpKind::kCorrSoftargmax;
                        op.max_disp = sp->maxDisparity();
                        op.is_min = stereo(nx)->softargmaxType() == SoftargmaxType::kMin;
                        op.out = nx->out[0]->id;
                        { done[index_of(nx)] = true; op.lays.push_back(index_of(nx)); }
                        ops_.push_back(op);
                        continue;
                    }
                }
                // Conv3D [+ Transform {1,0,2,3}] [+ ELU]   (Pad before a Conv3D is folded by the table)
                if (fuse && sp && (sp->kind() == Kind::kConv3D || sp->kind() == Kind::kConv3DTranspose)) {
                    ConvFusion f;
                    TensorImpl* cur = l->out[0];
                    std::vector<int> absorbed;
                    LayerData* nx = soleConsumer(cur->id);
                    if (sp->kind() == Kind::kConv3D && nx && isKind(nx, Kind::kTransform)) {
                        Permutation p = stereo(nx)->permutation();
                        if (p.order[0] == 1 && p.order[1] == 0 && p.order[2] == 2 && p.order[3] == 3) {
                            f.out_dchw = true;
                            absorbed.push_back(index_of(nx));
                            cur = nx->out[0];
                            nx = soleConsumer(cur->id);
                        }
                    }
                    // Pad (zero slices appended in D) feeding only this Conv3D: read the unpadded tensor, the gather
                    // table turns the missing slices into zeros
                    int folded_pad_op = -1;
                    if (sp->kind() == Kind::kConv3D && l->in[0]->producer && isKind(l->in[0]->producer, Kind::kPadding) &&
                        consumers(l->in[0]->id) == 1) {
                        for (size_t oi = 0; oi < ops_.size(); oi++)
                            if (ops_[oi].kind == OpKind::kPlugin && ops_[oi].out == l->in[0]->id) folded_pad_op = (int)oi;
                        if (folded_pad_op >= 0) f.in_pad_end = stereo(l->in[0]->producer)->padEnd();
                    }
                    // Conv3DTranspose [+ Slice [0,d)] [+ skip add] [+ ELU] [+ Transform {1,0,2,3}]: the decoder pattern
                    // of the 3-D models (nvsmall_1025x321_net.cpp:331-398) in one launch
                    if (sp->kind() == Kind::kConv3DTranspose && nx && isKind(nx, Kind::kSlice) && stereo(nx)->sliceStart() == 0) {
                        f.out_depth = stereo(nx)->sliceEnd();
                        absorbed.push_back(index_of(nx));
                        cur = nx->out[0];
                        nx = soleConsumer(cur->id);
                    }
                    int resid = -1;
                    if (nx && nx->type == LayerType::kELEMENTWISE && nx->ew == ElementWiseOperation::kSUM &&
                        volume(nx->out[0]->dims) == volume(cur->dims)) {
                        TensorImpl* other = nx->in[0] == cur ? nx->in[1] : nx->in[0];
                        if (other != cur && (other->producer == nullptr || done[index_of(other->producer)])) {
                            resid = other->id;
                            f.residual = true;
                            absorbed.push_back(index_of(nx));
                            cur = nx->out[0];
                            nx = soleConsumer(cur->id);
                        }
                    }
                    if (nx && isKind(nx, Kind::kElu)) {
                        f.act = RT_ACT_ELU;
                        absorbed.push_back(index_of(nx));
                        cur = nx->out[0];
                        nx = soleConsumer(cur->id);
                    }
                    if (sp->kind() == Kind::kConv3DTranspose && nx && isKind(nx, Kind::kTransform)) {
                        Permutation p = stereo(nx)->permutation();
                        if (p.order[0] == 1 && p.order[1] == 0 && p.order[2] == 2 && p.order[3] == 3) {
                            f.out_dchw = true;
                            absorbed.push_back(index_of(nx));
                            cur = nx->out[0];
                        }
                    }
                    if ((f.act || f.out_dchw || f.residual || f.out_depth || f.in_pad_end) && sp->setFusion(f)) {
                        for (int a : absorbed) { done[a] = true; op.lays.push_back(a); }
                        if (f.in_pad_end) {                       // drop the Pad launch, consume its input
                            op.in[0] = ops_[folded_pad_op].in[0];
                            ops_.erase(ops_.begin() + folded_pad_op);
                        }
                        op.kind = OpKind::kConv3D;
                        op.plugin = l->plugin;
                        op.splugin = sp;
                        op.resid = resid;
                        op.out = cur->id;
                        ops_.push_back(op);
                        continue;
                    }
                }
                op.kind = OpKind::kPlugin;
                op.plugin = l->plugin;
                if (sp) sp->setExactFp32(exact_fp32_);
                if (l->out.size() != 1) return fail(l->name + ": multi-output plugins are not supported");
                if (l->in.size() > 8) return fail(l->name + ": plugins with more than 8 inputs are not supported");
                ops_.push_back(op);
                continue;
            }
            default:
                return fail(l->name + ": layer type not supported by the Stereo DNN engine");
        }
    }
    std::ostringstream s;
    s << "engine: " << layers_.size() << " layers lowered to " << ops_.size() << " launches";
    log_.log(ILogger::Severity::kINFO, s.str().c_str());
    return true;
}

// Two-stream schedule: everything that depends only on the SECOND network input runs on the side
// stream (the right-image encoder); ops that mix both wait on events.
// Internal activations that only pitch-aware launches touch (2-D convolution plans, the fused correlation, channel
// concatenation) get a row pitch of a multiple of 32 floats: W is odd in every Stereo DNN network, so dense rows
// start at arbitrary 4-byte offsets and every 128-byte access of a tile straddles two cache lines (measured
// +6 % on the 3x3 layers with aligned rows).  Bindings and everything a generic plugin launch sees stay dense,
// as TensorRT's plugin contract requires.
void EngineImpl::assignPitch() {
    if (knob("RT_NO_PITCH")) return;
    std::vector<char> ok(tensors_.size(), 1);
    for (auto& t : tensors_) {
        if (t->is_input || t->is_output || t->alias_of >= 0 || t->dims.nbDims != 3 || t->bstride != 0) ok[t->id] = 0;
        if (t->alias_of >= 0) ok[root(t->id)] = 0;         // shared buffers stay dense
    }
    auto clear = [&](int t) { if (t >= 0) ok[root(t)] = 0; };
    auto P = [&](int t) { return t >= 0 && ok[root(t)] != 0; };
    for (auto& op : ops_) {
        const bool aware = op.kind == OpKind::kConv || op.kind == OpKind::kCorrSoftargmax || op.kind == OpKind::kConcat;
        if (aware) continue;
        for (int i : op.in) clear(i);
        clear(op.out);
        clear(op.resid);
    }
    for (bool changed = true; changed;) {
        changed = false;
        for (auto& op : ops_) {
            if (op.kind == OpKind::kConv && op.resid >= 0 && P(op.resid) != P(op.out)) {
                clear(op.resid); clear(op.out); changed = true;
            }
            if (op.kind == OpKind::kCorrSoftargmax && (P(op.in[0]) != P(op.in[1]) || P(op.in[0]) != P(op.out))) {
                clear(op.in[0]); clear(op.in[1]); clear(op.out); changed = true;   // one storage type for the fused kernel
            }
            if (op.kind == OpKind::kConcat) {
                bool all = P(op.out), any = P(op.out);
                for (int i : op.in) { all = all && P(i); any = any || P(i); }
                if (any && !all) {
                    clear(op.out);
                    for (int i : op.in) clear(i);
                    changed = true;
                }
            }
        }
    }
    // half2 mode (IBuilder::setHalf2Mode, sample_app/main.cpp:256-262): the same tensors are also stored as fp16 --
    // TensorRT keeps activations in fp16 between its own layers and hands plugins what they asked for; arithmetic
    // stays fp32 in the kernels.  Rows are then padded to 64 elements (128 bytes, and the even pitch the 2-pixel
    // stores need).
    const bool f16 = half2_ && !knob("RT_NO_F16");
    for (auto& t : tensors_) {
        if (!ok[t->id]) continue;
        const char* ex = knob("RT_PITCH_EXTRA");          // A/B knob: extra 128-byte lines per row
        const int q = f16 ? 64 : 32;
        const int w = t->dims.d[2], pitch = (w + q - 1) / q * q + q * (ex ? atoi(ex) : 0);
        t->pitch = pitch == w ? 0 : pitch;
        t->f16 = f16;
    }
    auto apply = [&](bool with_f16) {
        for (auto& op : ops_) {
            if (op.kind != OpKind::kConv) continue;
            const TensorImpl& x = *tensors_[root(op.in[0])];
            const TensorImpl& y = *tensors_[root(op.out)];
            if ((x.pitch || y.pitch) && rt_conv_plan_set_pitch(op.plan, x.pitch, y.pitch) != 0) return false;
            if (rt_conv_plan_set_io_types(op.plan, with_f16 && x.f16 ? RT_F16 : RT_F32, with_f16 && y.f16 ? RT_F16 : RT_F32) != 0)
                return false;
        }
        return true;
    };
    if (!apply(f16)) {
        if (f16) {       // a layer has no fp16 form (e.g. Winograd reading an fp32 binding): keep the whole engine fp32
            log_.log(ILogger::Severity::kWARNING, (std::string("half2 mode: ") + rt_last_error_string() + " -- activations stay fp32").c_str());
            for (auto& t : tensors_) t->f16 = false;
            if (apply(false)) return;
        }
        log_.log(ILogger::Severity::kERROR, rt_last_error_string());
        ok_ = false;
        return;
    }
    // Channel-interleaved storage -- (C/4, H, pitch, 4) in fp32, (C/8, H, pitch, 8) in fp16: one 16-byte slot per pixel
    // and channel group -- for the tensors that only 3x3 stride-1 plans read and write (the inside of the two feature
    // towers and of the bottleneck: 34 of the 49 launches of ResNet-18 2D write one).  Those kernels then move whole
    // cache lines with a quarter of the memory instructions (conv_wino.hip.h, conv_f16.hip.h).  A plan takes any mix
    // of planar and interleaved tensors, so the layout is a per-tensor property and nothing is converted anywhere.
    std::vector<char> il(tensors_.size(), 0);
    auto corr_takes_il = [&](const Op& op) {        // fused correlation + soft-argmax on the matrix cores (corr_mfma.hip.h)
        const Dims& f = tensors_[op.in[0]]->dims;
        // (half2 mode, round 6: fp16 maps in groups of 8 channels, corr_softargmax_mfma_f16_kernel -- the map stays an fp16 plane)
        return op.kind == OpKind::kCorrSoftargmax && f.d[0] % (f16 ? 8 : 4) == 0 && f.d[0] <= 32 && op.max_disp <= 64 &&
               !knob("RT_NO_CORR_MFMA") && !knob("RT_NO_IL8") && !(f16 && knob("RT_NO_CORR_MFMA_F16"));
    };
    // A concatenation that foldConcats() will fold can stay interleaved (fp32): every member but the last holds whole groups of 4
    // channels, the last one is padded to a group -- the 33-channel input of conv2D_1 is the 8 groups of left_conv1_act and a ninth
    // with the soft-argmax map in lane 0 (resnet18_2D_513x257_net.cpp:601-615).  The first member then serves the left tower's first
    // residual block in the layout the streaming kernel wants, and that block merges with the right tower's (round 3: two extra launches).
    std::vector<char> catpad(tensors_.size(), 0), cat_il(ops_.size(), 0);
    for (size_t ci = 0; ci < ops_.size(); ci++) {
        const Op& op = ops_[ci];
        // (half2 mode, round 6: the same with groups of 8 fp16 channels -- conv_f16mma_kernel takes the padded input, the fp16 matrix-core
        //  correlation writes lane 0 of a fifth group)
        if (op.kind != OpKind::kConcat || knob("RT_NO_IL8") || knob("RT_NO_IL_CONCAT") || (f16 && knob("RT_NO_IL_CONCAT_F16")) || !concatFoldable(ci)) continue;
        const int G = f16 ? 8 : 4;
        bool cand = ok[op.out] != 0;
        for (size_t k = 0; k < op.in.size(); k++) {
            const TensorImpl& t = *tensors_[op.in[k]];
            cand = cand && ok[t.id] && (t.dims.d[0] % G == 0 || k + 1 == op.in.size());
            if (t.dims.d[0] % G != 0)                                   // written as lane 0 of a group: the matrix-core correlation does that
                for (const Op& pr : ops_)
                    if (pr.out == t.id) cand = cand && t.dims.d[0] == 1 && corr_takes_il(pr);
        }
        for (const Op& rd : ops_)                                       // readers of the whole: plans that take a padded interleaved input
            for (int x : rd.in)
                if (x == op.out && &rd != &op) cand = cand && rd.kind == OpKind::kConv && (rt_conv_plan_supports_il8(rd.plan) & (1 | 16)) != 0;
        if (!cand) continue;
        cat_il[ci] = 1;
        catpad[op.out] = 1;
        catpad[op.in.back()] = 1;
    }
    for (auto& t : tensors_) il[t->id] = ok[t->id] && t->f16 == f16 && (t->dims.d[0] % (f16 ? 8 : 4) == 0 || catpad[t->id]);
    for (size_t oi = 0; oi < ops_.size(); oi++) {
        const Op& op = ops_[oi];
        if (cat_il[oi]) continue;
        int caps = op.kind == OpKind::kConv ? rt_conv_plan_supports_il8(op.plan) : 0;     // bit 0 input, 1 output, 2 residual, 4 padded input
        if (corr_takes_il(op)) caps = 1 | (catpad[op.out] ? 2 : 0);
        if (!(caps & 1)) for (int i : op.in) il[root(i)] = (caps & 16) && catpad[root(i)] ? il[root(i)] : 0;
        if (!(caps & 2)) il[root(op.out)] = 0;
        if (op.resid >= 0 && !(caps & 4)) il[root(op.resid)] = 0;
    }
    for (bool changed = true; changed;) {
        changed = false;
        for (auto& op : ops_)                            // both feature maps or neither
            if (op.kind == OpKind::kCorrSoftargmax) {
                if (il[root(op.in[0])] != il[root(op.in[1])]) { il[root(op.in[0])] = il[root(op.in[1])] = 0; changed = true; }
                if (il[op.out] && !il[root(op.in[0])]) { il[op.out] = 0; changed = true; }     // no interleaved map without the matrix-core kernel
            }
        for (size_t ci = 0; ci < ops_.size(); ci++) {    // a concatenation and all its members, or none of them
            if (!cat_il[ci]) continue;
            const Op& op = ops_[ci];
            bool all = il[op.out] != 0;
            for (int i : op.in) all = all && il[i];
            if (all) continue;
            il[op.out] = 0;
            for (int i : op.in) il[i] = 0;
            cat_il[ci] = 0;
            changed = true;
        }
    }
    for (auto& op : ops_) {
        if (op.kind != OpKind::kConv) continue;
        const int xi = il[root(op.in[0])], yi = il[root(op.out)], ri = op.resid >= 0 ? il[root(op.resid)] : 0;
        if (knob("RT_IL_TRACE")) fprintf(stderr, "[rt] %-28s x%d y%d r%d caps %d\n", op.name.c_str(), xi, yi, ri, rt_conv_plan_supports_il8(op.plan));
        if (!(xi || yi || ri)) continue;
        if (rt_conv_plan_set_layouts(op.plan, xi, yi, ri) != 0) {
            log_.log(ILogger::Severity::kERROR, rt_last_error_string());
            ok_ = false;
            return;
        }
    }
    for (auto& op : ops_)
        if (op.kind == OpKind::kCorrSoftargmax) { op.il_in = il[root(op.in[0])] != 0; op.il_out = op.il_in && il[op.out] != 0; }
    int n_il = 0;
    for (auto& t : tensors_) t->il8 = il[t->id] != 0;
    for (auto& op : ops_) n_il += il[root(op.out)] != 0;         // tensors that launches really write
    log_.log(ILogger::Severity::kINFO, (std::string(f16 ? "half2 mode: " : "fp32: ") + std::to_string(n_il) + " of " + std::to_string(ops_.size()) + " launches write channel-interleaved tensors").c_str());
}

void EngineImpl::assignStreams() {
    // (re)computed from scratch: the engine calls this once before the folding passes and once after them
    two_streams_ = false;
    for (auto& op : ops_) { op.stream = 0; op.wait_on.clear(); op.publish = false; }
    for (auto& t : tensors_) t->stream = 0;
    if (knob("RT_SINGLE_STREAM")) return;
    int n_inputs = 0;
    for (int b : bindings_)
        if (tensors_[b]->is_input) n_inputs++;
    if (n_inputs != 2) return;
    const int second = bindings_[1];
    std::vector<int> dep(tensors_.size(), 0);      // bit0: depends on input 0 / others, bit1: depends on second input
    for (auto& t : tensors_)
        if (t->is_input) dep[t->id] = t->id == second ? 2 : 1;
    auto dep_of = [&](int t) { return dep[root(t)]; };
    int side_ops = 0;
    for (auto& op : ops_) {
        int d = 0;
        for (int i : op.in) d |= dep_of(i);
        if (op.resid >= 0) d |= dep_of(op.resid);
        for (int i : op.twin_in) d |= dep_of(i);    // a merged siamese launch reads both towers
        dep[root(op.out)] |= d;
        dep[op.out] |= d;
        op.stream = d == 2 ? 1 : 0;
        tensors_[op.out]->stream = op.stream;
        tensors_[root(op.out)]->stream = op.stream;
        if (op.twin_out >= 0) {
            dep[root(op.twin_out)] |= d;
            dep[op.twin_out] |= d;
            tensors_[op.twin_out]->stream = op.stream;
            tensors_[root(op.twin_out)]->stream = op.stream;
        }
        side_ops += op.stream;
    }
    if (side_ops < 4) {          // not worth a second stream
        for (auto& op : ops_) op.stream = 0;
        for (auto& t : tensors_) t->stream = 0;
        return;
    }
    two_streams_ = true;
    for (auto& op : ops_) {
        auto consider = [&](int t) {
            const int r = root(t);
            if (!tensors_[r]->is_input && tensors_[r]->stream != op.stream) op.wait_on.push_back(r);
        };
        for (int i : op.in) consider(i);
        if (op.resid >= 0) consider(op.resid);
        for (int i : op.twin_in) consider(i);
    }
    for (auto& op : ops_)
        for (auto& o2 : ops_)
            for (int w : o2.wait_on) op.publish |= (w == root(op.out)) || (op.twin_out >= 0 && w == root(op.twin_out));
    // every wait must have a launch that records the event it waits for (a pass that removes an op after this one would break that)
    for (auto& op : ops_)
        for (int w : op.wait_on) {
            bool published = false;
            for (auto& o2 : ops_) published = published || (o2.publish && (root(o2.out) == w || (o2.twin_out >= 0 && root(o2.twin_out) == w)));
            if (!published) {
                log_.log(ILogger::Severity::kERROR, ("engine: " + op.name + " waits for tensor '" + tensors_[w]->name + "' of the other stream, which no launch publishes").c_str());
                ok_ = false;
            }
        }
}

void EngineImpl::planIssueOrder() {
    if (!two_streams_) return;
    // Issue order of the synchronous API (execute(): the reference's loop, sample_app/main.cpp:303-309): alternate between
    // the two streams.  The network definition lists the whole left tower before the right one and a launch costs the host
    // 3-4 us: with an idle GPU at the call the side stream's first kernel would start 18 launches (60-90 us) after the main
    // stream's -- measured on MI355X, one context, round 2: 645 -> 613 us per pair.  Per stream the order is unchanged, and an
    // op that waits for a tensor of the other stream is issued after that tensor's producer (the event it waits for must
    // have been recorded).  enqueue() keeps the definition order: there the host runs ahead of the GPU and alternating
    // streams costs it more per launch (167 vs 127 us per step; 2087 vs 2097 pairs/s with four contexts).
    {
        std::vector<size_t> q[2];
        for (size_t i = 0; i < ops_.size(); i++) q[ops_[i].stream].push_back(i);
        std::vector<char> produced(tensors_.size(), 0);
        auto ready = [&](const Op& op) {
            for (int w : op.wait_on)
                if (!produced[w]) return false;
            return true;
        };
        size_t h[2] = {0, 0};
        int turn = 1;
        while (issue_sync_.size() < ops_.size()) {
            int st = turn;
            if (h[st] >= q[st].size() || !ready(ops_[q[st][h[st]]])) st ^= 1;
            if (h[st] >= q[st].size() || !ready(ops_[q[st][h[st]]])) break;        // cannot happen for a topological order
            const size_t i = q[st][h[st]++];
            issue_sync_.push_back(i);
            for (int r = ops_[i].out; r >= 0; r = tensors_[r]->alias_of) produced[r] = 1;
            for (int r = ops_[i].twin_out; r >= 0; r = tensors_[r]->alias_of) produced[r] = 1;
            turn = st ^ 1;
        }
        if (issue_sync_.size() != ops_.size()) issue_sync_.clear();
    }
}

// Residual blocks in one launch: conv3x3 (+ELU) -> conv3x3 + skip (+ELU) with the skip connection being the block's input
// (reference resnet18_2D_513x257_net.cpp:66-575, 8 blocks per feature tower) become one rt_resblock_plan: the intermediate
// tensor never leaves the CU (conv_split.hip.h, conv_s3rb_kernel).  In half2 mode the two layers run on fp16 operands with an fp16
// tensor in between; the fused form (conv_rbh.hip.h) rounds the intermediate to fp16 in LDS and is bit-identical to them.
void EngineImpl::fuseResBlocks() {
    // Default: the blocks the STREAMING kernel serves (conv_rbs.hip.h: 32 -> 32 -> 32 channels, ELU after both, channel-interleaved fp32
    // tensors) on images large enough to fill the GPU with its strips x segments -- measured on MI355X, round 2, ResNet-18 2D at
    // 1257x369 with four contexts: 2185 vs 2057 pairs/s, 2.8x less HBM traffic per block.  Everything else stays two launches: the
    // per-tile form of the fused block (conv_s3rb_kernel) is slower than its two layers (1934 vs 2069 pairs/s).  The pass runs AFTER
    // assignPitch(), so it sees the layouts the tensors really have (a block whose input is written straight into a concatenation
    // reads a planar tensor and is left alone).  Development knobs: RT_RB=1 fuses every block rt_resblock_plan_create accepts
    // (tests, A/B), RT_RB=0 / RT_NO_RB none.
    const char* e = knob("RT_RB");
    const bool force = e && atoi(e) != 0;
    // half2 mode (round 6): the same block on fp16 tensors and fp16 operands, conv_f16rbd_kernel -- bit-identical to its two layers
    const bool h2 = half2_ && !knob("RT_NO_F16");
    if ((e && !force) || knob("RT_NO_FUSION") || knob("RT_NO_RB") || knob("RT_NO_S3") || exact_fp32_ || (h2 && (knob("RT_NO_RBH") || knob("RT_NO_F16MMA")))) return;
    int fused = 0;
    for (size_t ia = 0; ia < ops_.size(); ia++) {
        Op& a = ops_[ia];
        if (a.kind != OpKind::kConv || !a.conv_layer || a.conv_layer->type != LayerType::kCONVOLUTION || a.resid >= 0) continue;
        if (tensors_[a.out]->is_output) continue;
        // the one consumer of the intermediate
        int ib = -1, uses = 0;
        for (size_t j = 0; j < ops_.size(); j++) {
            if (j == ia) continue;
            for (int x : ops_[j].in)
                if (root(x) == root(a.out)) { uses++; ib = (int)j; }
            if (ops_[j].resid >= 0 && root(ops_[j].resid) == root(a.out)) uses += 2;
        }
        if (uses != 1 || ib < (int)ia) continue;
        Op& b = ops_[ib];
        if (b.kind != OpKind::kConv || !b.conv_layer || b.conv_layer->type != LayerType::kCONVOLUTION || b.in.size() != 1 ||
            root(b.in[0]) != root(a.out) || b.resid < 0 || root(b.resid) != root(a.in[0]))
            continue;
        auto desc = [&](const Op& op, bool resid) {
            const LayerData* l = op.conv_layer;
            const Dims& x = l->in[0]->dims;
            rtConv2dDesc d{};
            d.Cin = x.d[0]; d.Cout = l->nb_maps; d.Hin = x.d[1]; d.Win = x.d[2];
            d.KH = l->ksize.h(); d.KW = l->ksize.w(); d.stride = l->stride.h();
            d.pad_h = l->padding.h(); d.pad_w = l->padding.w();
            d.act = op.act; d.has_residual = resid;
            d.dtype = l->kernel.type == DataType::kHALF ? RT_F16 : RT_F32;
            return d;
        };
        const rtConv2dDesc d1 = desc(a, false), d2 = desc(b, true);
        const TensorImpl &x = *tensors_[root(a.in[0])], &y = *tensors_[root(b.out)];
        {
            bool ok = d1.Cin == 32 && d1.Cout == 32 && d2.Cout == 32 && a.act == RT_ACT_ELU && b.act == RT_ACT_ELU;
            ok = ok && x.il8 && y.il8 && x.f16 == h2 && y.f16 == h2;              // what the streaming kernels read and write
            if (h2 && !ok) continue;                                              // (half2: no other form of the fused block)
            // forced: also small images, and -- in RT_EXPERIMENTAL builds of the kernel library -- every other block through the per-tile form
            if (force ? (!ok && !rt_has_experimental()) : !(ok && ((d1.Win + 29) / 30) * ((d1.Hin + 15) / 16) >= 200)) continue;   // default: enough strips x segments for every CU
        }
        const LayerData *la = a.conv_layer, *lb = b.conv_layer;
        rtConvPlan* rb = nullptr;
        if (rt_resblock_plan_create(&rb, &d1, la->kernel.values, la->bias.count ? la->bias.values : nullptr, &d2, lb->kernel.values,
                                    lb->bias.count ? lb->bias.values : nullptr) != 0)
            continue;                                   // not of that form: the two launches stay
        // the tensors' properties, as assignPitch() gave them to the two plans this one replaces
        bool cfg = (!(x.pitch || y.pitch) || rt_conv_plan_set_pitch(rb, x.pitch, y.pitch) == 0) &&
                   rt_conv_plan_set_io_types(rb, x.f16 ? RT_F16 : RT_F32, y.f16 ? RT_F16 : RT_F32) == 0 &&
                   (!(x.il8 || y.il8) || rt_conv_plan_set_layouts(rb, x.il8, y.il8, x.il8) == 0);
        if (!cfg) {                                     // the fused plan cannot take these tensors: keep the two launches
            rt_conv_plan_destroy(rb);
            continue;
        }
        rt_conv_plan_destroy(a.plan);
        rt_conv_plan_destroy(b.plan);
        a.plan = rb;
        a.out = b.out;
        a.resid = b.resid;
        a.act = b.act;
        a.name += "+" + b.name;                         // IProfiler rows name both layers
        a.lays.insert(a.lays.end(), b.lays.begin(), b.lays.end());
        ops_.erase(ops_.begin() + ib);
        fused++;
    }
    if (fused)
        log_.log(ILogger::Severity::kINFO, ("engine: " + std::to_string(fused) + " residual blocks fused into one launch each; " +
                                            std::to_string(ops_.size()) + " launches").c_str());
    // Tensors that only travel from one tower block to the next are stored PRE-SPLIT (rt_resblock_plan_set_split, conv_rbd.hip.h): the
    // producing block writes the fp16 hi / lo operand pairs its successor's matrix instructions consume, the successor fills its LDS by
    // direct global -> LDS loads.  Such a tensor has exactly one reader -- a tower block that takes it as input and skip connection -- and
    // is no binding; the first block of a tower (fp32 input) and the last one (fp32 output) stay at the edges.  RT_NO_RBD=1: none.
    if (!fused || knob("RT_NO_RBD")) return;
    auto tower = [&](const Op& op) { return op.kind == OpKind::kConv && op.plan && rt_resblock_plan_supports_split(op.plan) != 0; };
    int nsplit = 0;
    for (size_t ia = 0; ia < ops_.size(); ia++) {
        const Op& a = ops_[ia];
        if (!tower(a) || root(a.out) != a.out) continue;
        TensorImpl& t = *tensors_[a.out];
        if (t.is_output || t.is_input || t.bstride != 0 || t.cpad != 0 || !t.il8 || t.f16) continue;
        int readers = 0;
        bool ok = true;
        for (size_t j = 0; j < ops_.size(); j++) {
            const Op& b = ops_[j];
            bool reads = b.resid >= 0 && root(b.resid) == t.id;
            for (int x : b.in) reads = reads || root(x) == t.id;
            if (!reads) continue;
            readers++;
            ok = ok && j > ia && tower(b) && b.in.size() == 1 && root(b.in[0]) == t.id && b.resid >= 0 && root(b.resid) == t.id;
        }
        if (readers == 1 && ok) { t.split = true; nsplit++; }
    }
    if (!nsplit) return;
    for (Op& op : ops_) {
        if (!tower(op)) continue;
        const bool xs = tensors_[root(op.in[0])]->split, ys = tensors_[root(op.out)]->split;
        if ((xs || ys) && rt_resblock_plan_set_split(op.plan, xs, ys) != 0) {
            log_.log(ILogger::Severity::kERROR, rt_last_error_string());
            ok_ = false;
            return;
        }
    }
    log_.log(ILogger::Severity::kINFO, ("engine: " + std::to_string(nsplit) + " tensors between tower blocks stored pre-split (fp16 hi / lo operand pairs)").c_str());
}

// Siamese towers in ONE launch per layer.  The two feature towers of every Stereo DNN share their weights (the reference exports
// the TF variables once per side: left_* / right_* tensors of trt_weights.bin are byte-identical; resnet18_2D_513x257_net.cpp:48-575
// builds the same layer sequence twice), so a left-tower launch and its right-tower twin differ only in the tensors they touch.
// When the twin tensors of a pair are dense internal tensors of the same layout, the right one is PLACED `batch` samples behind the
// left one in one buffer of 2 * maxBatchSize samples and the two launches become one launch over 2 * batch samples: twice the
// workgroups per launch (the streaming residual block fills every CU: 252 instead of 126 workgroups at 1257x369), weights fetched
// once, half the launches, and no second stream / cross-stream events for the towers.  The arithmetic per sample is untouched, so
// results are bit-identical to the separate launches (tests/test_net_parity.py).
void EngineImpl::mergeSiamese() {
    if (knob("RT_NO_SIAMESE") || knob("RT_NO_FUSION")) return;
    std::vector<int> ins;
    for (int b : bindings_)
        if (tensors_[b]->is_input) ins.push_back(b);
    if (ins.size() != 2) return;
    auto same_dims = [](const Dims& a, const Dims& b) {
        if (a.nbDims != b.nbDims) return false;
        for (int i = 0; i < a.nbDims; i++)
            if (a.d[i] != b.d[i]) return false;
        return true;
    };
    if (!same_dims(tensors_[ins[0]]->dims, tensors_[ins[1]]->dims)) return;
    auto same_weights = [](const Weights& a, const Weights& b) {
        if (a.count != b.count || a.type != b.type) return false;
        if (a.count == 0) return true;
        const size_t es = a.type == DataType::kHALF ? 2 : 4;
        return a.values && b.values && (a.values == b.values || std::memcmp(a.values, b.values, (size_t)a.count * es) == 0);
    };
    // 1. twin layers: same type and parameters, byte-identical weights, input i of the right one is the twin of input i of the left one
    std::vector<int> twin_t(tensors_.size(), -1), twin_l(layers_.size(), -1);
    std::vector<char> is_right(layers_.size(), 0);
    twin_t[ins[0]] = ins[1];
    auto same_layer = [&](const LayerData& a, const LayerData& b) {
        if (a.type != b.type || a.in.size() != b.in.size() || a.out.size() != b.out.size()) return false;
        for (size_t i = 0; i < a.in.size(); i++)
            if (twin_t[a.in[i]->id] != b.in[i]->id) return false;
        for (size_t i = 0; i < a.out.size(); i++)
            if (!same_dims(a.out[i]->dims, b.out[i]->dims)) return false;
        switch (a.type) {
            case LayerType::kCONVOLUTION:
            case LayerType::kDECONVOLUTION:
                return a.nb_maps == b.nb_maps && a.ksize.h() == b.ksize.h() && a.ksize.w() == b.ksize.w() && a.stride.h() == b.stride.h() &&
                       a.stride.w() == b.stride.w() && a.padding.h() == b.padding.h() && a.padding.w() == b.padding.w() &&
                       same_weights(a.kernel, b.kernel) && same_weights(a.bias, b.bias);
            case LayerType::kSCALE: return isIdentityScale(a) && isIdentityScale(b);
            case LayerType::kELEMENTWISE: return a.ew == b.ew;
            case LayerType::kACTIVATION: return a.act == b.act;
            case LayerType::kPLUGIN: {
                IStereoPlugin *pa = a.plugin ? dynamic_cast<IStereoPlugin*>(a.plugin) : nullptr, *pb = b.plugin ? dynamic_cast<IStereoPlugin*>(b.plugin) : nullptr;
                return pa && pb && pa->kind() == Kind::kElu && pb->kind() == Kind::kElu;
            }
            default: return false;
        }
    };
    for (size_t ia = 0; ia < layers_.size(); ia++) {
        const LayerData& a = *layers_[ia];
        if (is_right[ia] || a.in.empty()) continue;
        bool have = true;
        for (auto* t : a.in) have = have && twin_t[t->id] >= 0;
        if (!have) continue;
        for (size_t ib = 0; ib < layers_.size(); ib++) {
            if (ib == ia || is_right[ib] || twin_l[ib] >= 0 || !same_layer(a, *layers_[ib])) continue;
            twin_l[ia] = (int)ib;
            is_right[ib] = 1;
            for (size_t i = 0; i < a.out.size(); i++) twin_t[a.out[i]->id] = layers_[ib]->out[i]->id;
            break;
        }
    }
    // 2. tensor pairs that can share a buffer [left samples | right samples]
    auto plain = [&](const TensorImpl& t) { return !t.is_input && !t.is_output && t.alias_of < 0 && t.bstride == 0 && t.twin_of < 0 && !t.has_twin; };
    auto pairable = [&](int l, int r) {
        if (l < 0 || r < 0 || l == r) return false;
        const TensorImpl &a = *tensors_[l], &b = *tensors_[r];
        return plain(a) && plain(b) && same_dims(a.dims, b.dims) && a.pitch == b.pitch && a.f16 == b.f16 && a.il8 == b.il8 && a.split == b.split && b.cpad == 0;
    };
    // 3. op pairs: convolution plans covering twin layers, all of whose tensors pair up
    int merged = 0;
    for (size_t ia = 0; ia < ops_.size(); ia++) {
        Op& a = ops_[ia];
        if (a.kind != OpKind::kConv || a.twin || a.in.size() != 1 || a.lays.empty() || twin_l[a.lays[0]] < 0) continue;
        int ib = -1;
        for (size_t j = 0; j < ops_.size() && ib < 0; j++) {
            const Op& b = ops_[j];
            if (j == ia || b.kind != OpKind::kConv || b.twin || b.lays.size() != a.lays.size() || b.in.size() != 1 || b.act != a.act) continue;
            bool eq = true;
            for (size_t k = 0; k < a.lays.size(); k++) eq = eq && twin_l[a.lays[k]] == b.lays[k];
            if (eq) ib = (int)j;
        }
        if (ib < 0) continue;
        const Op& b = ops_[ib];
        // the towers' first layers read the two input bindings (through the identity scale layers of the generated networks: aliases) --
        // separate buffers of the caller: the first-layer kernels take both pointers (round 6), everything else needs its inputs placed
        // behind one another
        const bool from_bindings = root(a.in[0]) == ins[0] && root(b.in[0]) == ins[1] && a.resid < 0 && !knob("RT_NO_TWIN_INPUT") &&
                                   tensors_[a.in[0]]->alias_off == 0 && tensors_[b.in[0]]->alias_off == 0 && rt_conv_plan_supports_twin_input(a.plan) != 0;
        // the tensors as the ops see them (a tensor the op reads through an alias chain must be the plain buffer itself)
        if ((!from_bindings && (root(a.in[0]) != a.in[0] || root(b.in[0]) != b.in[0])) || root(a.out) != a.out || root(b.out) != b.out) continue;
        if ((a.resid >= 0) != (b.resid >= 0) || (a.resid >= 0 && (root(a.resid) != a.resid || root(b.resid) != b.resid))) continue;
        auto ok_pair = [&](int l, int r) {       // already placed together, or can be
            return l >= 0 && r >= 0 && ((tensors_[r]->twin_of == l && tensors_[l]->has_twin) || pairable(l, r));
        };
        if (!(from_bindings || ok_pair(a.in[0], b.in[0])) || !ok_pair(a.out, b.out) || (a.resid >= 0 && !ok_pair(a.resid, b.resid))) continue;
        auto place = [&](int l, int r) {
            tensors_[r]->twin_of = l;
            tensors_[l]->has_twin = true;
            tensors_[r]->cpad = tensors_[l]->cpad;         // same sample size as its twin (the host of an interleaved concatenation)
        };
        if (!from_bindings) place(a.in[0], b.in[0]);
        a.twin_bind = from_bindings;
        place(a.out, b.out);
        if (a.resid >= 0) place(a.resid, b.resid);
        a.twin = true;
        a.twin_in.push_back(b.in[0]);
        if (b.resid >= 0) a.twin_in.push_back(b.resid);
        a.twin_out = b.out;
        a.name += " | " + b.name;
        rt_conv_plan_destroy(b.plan);
        ops_.erase(ops_.begin() + ib);
        if ((size_t)ib < ia) ia--;
        merged++;
    }
    if (!merged) return;
    // 4. a merged launch needs the right tower's earlier (unmerged) launches before it: stable topological order by buffer
    {
        const size_t n = ops_.size();
        // tensors as the ops name them; two tensors are the same data when one is reached from the other along alias_of (a channel
        // range of a folded concatenation and the whole buffer; an identity scale and its input) -- siblings of a concatenation are not
        auto reads = [&](const Op& op) {
            std::vector<int> r(op.in);
            if (op.resid >= 0) r.push_back(op.resid);
            r.insert(r.end(), op.twin_in.begin(), op.twin_in.end());
            return r;
        };
        auto writes = [&](const Op& op) {
            std::vector<int> w{op.out};
            if (op.twin_out >= 0) w.push_back(op.twin_out);
            return w;
        };
        // ... i.e. when their element ranges inside the root buffer overlap (first sample; the other samples repeat the picture).  The
        // host of an interleaved concatenation and the members placed behind its own channels are the SAME buffer and different data.
        auto range = [&](int t, int& r, int64_t& lo, int64_t& hi) {
            const TensorImpl& x = *tensors_[t];
            int64_t n = (int64_t)volume(x.dims);
            lo = 0;
            for (r = t; tensors_[r]->alias_of >= 0; r = tensors_[r]->alias_of) lo += tensors_[r]->alias_off;
            if (x.dims.nbDims == 3 && tensors_[r]->pitch) n = (int64_t)x.dims.d[0] * x.dims.d[1] * tensors_[r]->pitch;
            hi = lo + n;
        };
        auto same_data = [&](int a, int b) {
            int ra, rb;
            int64_t la, ha, lb, hb;
            range(a, ra, la, ha);
            range(b, rb, lb, hb);
            return ra == rb && la < hb && lb < ha;
        };
        std::vector<std::vector<int>> rd(n), wr(n);
        for (size_t i = 0; i < n; i++) { rd[i] = reads(ops_[i]); wr[i] = writes(ops_[i]); }
        std::vector<char> placed(n, 0);
        std::vector<Op> sorted;
        for (size_t done = 0; done < n; done++) {
            size_t pick = n;
            for (size_t i = 0; i < n && pick == n; i++) {
                if (placed[i]) continue;
                bool ready = true;
                for (size_t j = 0; j < n && ready; j++) {
                    if (j == i || placed[j]) continue;
                    for (int w : wr[j])
                        for (int r : rd[i]) ready = ready && !same_data(w, r);
                }
                if (ready) pick = i;
            }
            if (pick == n) {        // a cycle cannot happen in a feed-forward graph; keep the original order if it does
                log_.log(ILogger::Severity::kERROR, "engine: siamese merge produced a dependency cycle");
                ok_ = false;
                return;
            }
            placed[pick] = 1;
            sorted.push_back(ops_[pick]);
        }
        ops_ = std::move(sorted);
    }
    log_.log(ILogger::Severity::kINFO, ("engine: " + std::to_string(merged) + " left / right tower launches merged into one launch over both images each; " +
                                        std::to_string(ops_.size()) + " launches").c_str());
}

// The default cost volume is never built (reference CostVolumePlugin kDefault, lib/kernels.cu:50-97: a (D, 2F, H, W) tensor
// of D shifted copies -- 1.0 GB per pair for NVSmall, 1.44 GB for ResNet-18 3D -- that the first Conv3D reads three times):
// when its only consumer is a Conv3D plan, the two (F,H,W) feature maps become the halves of one (2F,H,W) buffer and the
// convolution gathers slice d of the volume from it (rtConv3dDesc::cv_fold; the gather table's plane offsets + an x shift of
// d for the right-image half).  The CostVolume launch disappears.
void EngineImpl::foldCostVolumes() {
    if (knob("RT_NO_CV_FOLD") || knob("RT_NO_FUSION")) return;
    for (size_t ci = 0; ci < ops_.size(); ci++) {
        const Op cv = ops_[ci];
        if (cv.kind != OpKind::kPlugin) continue;
        IStereoPlugin* sp = cv.plugin ? dynamic_cast<IStereoPlugin*>(cv.plugin) : nullptr;
        if (!sp || sp->kind() != Kind::kCostVolume || sp->costVolumeType() != CostVolumeType::kDefault || cv.half_kind || cv.in.size() != 2) continue;
        // the one consumer: a Conv3D launch reading the volume as its input
        int ib = -1, uses = 0;
        for (size_t j = 0; j < ops_.size(); j++) {
            if (j == ci) continue;
            for (int x : ops_[j].in)
                if (root(x) == root(cv.out)) { uses++; ib = (int)j; }
            if (ops_[j].resid >= 0 && root(ops_[j].resid) == root(cv.out)) uses += 2;
        }
        if (uses != 1 || ib < (int)ci || tensors_[cv.out]->is_output) continue;
        Op& conv = ops_[ib];
        if (conv.kind != OpKind::kConv3D || !conv.splugin || conv.splugin->kind() != Kind::kConv3D || root(conv.in[0]) != root(cv.out)) continue;
        TensorImpl &l = *tensors_[cv.in[0]], &r = *tensors_[cv.in[1]];
        bool ok = l.id != r.id && l.dims.nbDims == 3 && r.dims.nbDims == 3 && l.dims.d[0] % 4 == 0;
        for (TensorImpl* t : {&l, &r}) {
            ok = ok && !t->is_input && !t->is_output && t->alias_of < 0 && t->pitch == 0 && !t->f16 && !t->il8 && t->bstride == 0;
            int producers = 0;
            for (size_t oi = 0; oi < ops_.size() && ok; oi++) {
                const Op& op = ops_[oi];
                if (oi == ci) continue;
                if (op.out == t->id) { producers++; ok = ok && op.kind == OpKind::kConv && oi < ci; }
                if (op.resid == t->id) ok = false;
                for (int x : op.in) ok = ok && x != t->id;
            }
            ok = ok && producers == 1;
        }
        if (!ok) continue;
        const int F = l.dims.d[0];
        ConvFusion f = conv.splugin->fusion();
        f.cv_fold = F;
        if (!conv.splugin->setFusion(f)) continue;             // window without a split-fp16 kernel: the volume is built as before
        // the (2F, H, W) buffer [left | right]
        std::unique_ptr<TensorImpl> fused(new TensorImpl());
        fused->id = (int)tensors_.size();
        fused->name = cv.name + "_features";
        fused->dims = DimsCHW(2 * F, l.dims.d[1], l.dims.d[2]);
        const int64_t plane = (int64_t)l.dims.d[1] * l.dims.d[2];
        const int fid = fused->id;
        fused->stream = conv.stream;
        tensors_.push_back(std::move(fused));
        bool failed = false;
        int half = 0;
        for (TensorImpl* t : {&l, &r}) {
            t->alias_of = fid;
            t->alias_off = (int64_t)half * F * plane;
            t->bstride = 2 * (int64_t)F * plane;
            half++;
            for (auto& op : ops_)
                if (op.kind == OpKind::kConv && op.out == t->id && rt_conv_plan_set_batch_strides(op.plan, 0, t->bstride, 0) != 0) failed = true;
        }
        if (failed) {
            log_.log(ILogger::Severity::kERROR, rt_last_error_string());
            ok_ = false;
            return;
        }
        conv.in[0] = fid;
        for (int w : cv.wait_on) conv.wait_on.push_back(root(w));          // the cost-volume launch's cross-stream waits move to the convolution
        ops_.erase(ops_.begin() + ci);
        ci--;
        log_.log(ILogger::Severity::kINFO, (cv.name + ": default cost volume folded into " + ops_[ib - 1].name + " (never materialised)").c_str());
    }
}

// half2 mode of the 3-D models (fp16 weight file -> IBuilder::setHalf2Mode, sample_app/main.cpp:256-262; the reference's Conv3D
// plugins then convert fp32 <-> fp16 around every cuDNN call, lib/conv3d_plugin.cpp:247-274): the 4-D tensors that only fused
// Conv3D / Conv3DTranspose launches touch are STORED as fp16 (dense, same (D,C,H,W) / (K,D,H,W) layouts) -- half the bytes of
// the tensors that make these networks HBM-heavy -- and multiplied as fp16 operands with fp32 accumulation
// (conv_s3_kernel<.., TIN, TOUT>).  The 2-D feature towers keep fp32 activations; so do the volume the last layer writes
// and the soft-argmin.
void EngineImpl::assignHalf3D() {
    if (!half2_ || knob("RT_NO_F16") || knob("RT_NO_F16_3D")) return;
    std::vector<char> h(tensors_.size(), 0);
    for (auto& t : tensors_) h[t->id] = !t->is_input && !t->is_output && t->alias_of < 0 && t->dims.nbDims == 4 && !t->f16;
    auto clear = [&](int t) { if (t >= 0) h[root(t)] = 0; };
    for (auto& op : ops_) {
        if (op.kind == OpKind::kConv3D) continue;
        for (int i : op.in) clear(i);
        clear(op.out);
        clear(op.resid);
    }
    // The two feature maps of a folded cost volume ((2F, H, W) buffer [left | right], foldCostVolumes): when the tower launches that
    // write them can store fp16, channel-interleaved (2F/8, H, W, 8), the first Conv3D gathers 16-byte slots of fp16 operands
    // (conv_f16mma_kernel with the x-shift table) instead of fp32 planes it has to split: 1.18 -> 0.6 ms for NVSmall's conv3D_1.
    struct Feat { int tensor; size_t conv3d; std::vector<size_t> producers; };
    std::vector<Feat> feats;
    if (!knob("RT_NO_IL8") && !knob("RT_NO_IL8_3D") && !knob("RT_NO_F16_FEAT"))
        for (size_t ci = 0; ci < ops_.size(); ci++) {
            const Op& cv = ops_[ci];
            if (cv.kind != OpKind::kConv3D || !cv.splugin->fusion().cv_fold) continue;
            if (cv.splugin->ilCaps() & 32) continue;                    // the factored fold (fold_factor.hip.h) wants the maps as they are: planar fp32
            const int x = root(cv.in[0]);
            TensorImpl& ft = *tensors_[x];
            if (ft.dims.nbDims != 3 || ft.f16 || ft.il8 || ft.pitch || ft.dims.d[0] % 16 != 0) continue;
            Feat f{x, ci, {}};
            bool ok = true;
            for (size_t i = 0; i < ops_.size() && ok; i++) {
                const Op& op = ops_[i];
                if (i == ci) continue;
                bool reads = op.resid >= 0 && root(op.resid) == x;
                for (int t : op.in) reads = reads || root(t) == x;
                if (reads) ok = false;                                  // only the Conv3D reads the maps
                if (root(op.out) == x) { ok = ok && op.kind == OpKind::kConv && !op.twin; f.producers.push_back(i); }
            }
            if (!ok || f.producers.size() != 2) continue;
            size_t done = 0;
            for (; done < f.producers.size(); done++) {
                Op& p = ops_[f.producers[done]];
                const TensorImpl& pin = *tensors_[root(p.in[0])];
                if (rt_conv_plan_set_io_types(p.plan, pin.f16 ? RT_F16 : RT_F32, RT_F16) != 0 || !(rt_conv_plan_supports_il8(p.plan) & 2) ||
                    rt_conv_plan_set_layouts(p.plan, pin.il8, 1, 0) != 0)
                    break;
            }
            if (done != f.producers.size()) {                            // not every producer can: back to fp32 planar maps
                for (size_t k = 0; k <= done && k < f.producers.size(); k++) {
                    Op& p = ops_[f.producers[k]];
                    const TensorImpl& pin = *tensors_[root(p.in[0])];
                    rt_conv_plan_set_io_types(p.plan, pin.f16 ? RT_F16 : RT_F32, RT_F32);
                    rt_conv_plan_set_layouts(p.plan, pin.il8, 0, 0);
                }
                continue;
            }
            h[x] = 1;
            feats.push_back(f);
        }
    for (bool changed = true; changed;) {
        changed = false;
        for (auto& op : ops_) {
            if (op.kind != OpKind::kConv3D) continue;
            const int x = root(op.in[0]), y = root(op.out), r = op.resid >= 0 ? root(op.resid) : -1;
            if (r >= 0 && h[r] != h[y]) { h[r] = h[y] = 0; changed = true; }       // the residual is read with the output's type
            if (!h[x] && !h[y]) continue;
            if (!op.splugin->setIoTypes(h[x] != 0, h[y] != 0)) {                     // no fp16-storage kernel for this launch
                op.splugin->setIoTypes(false, false);
                if (h[x] || h[y] || (r >= 0 && h[r])) changed = true;
                h[x] = h[y] = 0;
                if (r >= 0) h[r] = 0;
            }
        }
    }
    int n = 0;
    for (auto& op : ops_)
        if (op.kind == OpKind::kConv3D) {
            const int x = root(op.in[0]), y = root(op.out);
            op.splugin->setIoTypes(h[x] != 0, h[y] != 0);
            n += h[y] != 0;
        }
    for (const Feat& f : feats) {
        if (h[f.tensor]) {                                               // the maps are fp16, interleaved: so are the tensors that alias into them
            for (auto& t : tensors_)
                if (t->id == f.tensor || (t->alias_of >= 0 && root(t->id) == f.tensor)) { t->f16 = true; t->il8 = true; }
            log_.log(ILogger::Severity::kINFO, ("half2 mode: " + ops_[f.conv3d].name + " reads the two feature maps as fp16, channel-interleaved").c_str());
            continue;
        }
        for (size_t i : f.producers) {                                   // the Conv3D could not take them: the producers write fp32 again
            Op& p = ops_[i];
            const TensorImpl& pin = *tensors_[root(p.in[0])];
            rt_conv_plan_set_io_types(p.plan, pin.f16 ? RT_F16 : RT_F32, RT_F32);
            rt_conv_plan_set_layouts(p.plan, pin.il8, 0, 0);
        }
    }
    for (auto& t : tensors_)
        if (h[t->id] && t->dims.nbDims == 4) t->f16 = true;
    if (n) log_.log(ILogger::Severity::kINFO, ("half2 mode: " + std::to_string(n) + " Conv3D / Conv3DTranspose launches write fp16 tensors").c_str());
}

// half2 mode of the 3-D models, continued: the fp16 4-D tensors in depth-major form (D, C, H, W) are stored channel-interleaved,
// (D, C/8, H, W, 8) -- one 16-byte slot per pixel and group of 8 channels, as the 2-D fp16 tensors of the towers are (assignPitch).
// A Conv3D between two such tensors then runs on fp16 operands (conv_f16mma_kernel: one 16-byte load per pixel and channel group
// instead of eight 2-byte loads, 8-byte stores instead of four 2-byte ones: 0.36 vs 0.54 ms for the 32 -> 32 layer of NVSmall
// as a 2-D proxy, tools/dev/il_probe_f16.py); the folded-cost-volume Conv3D writes one; a fused Conv3DTranspose reads its skip
// tensor that way.  Which tensors qualify follows from the launches' own capabilities (IStereoPlugin::ilCaps).
// The soft-argmax at the end of the 3-D models (disp_softargmax over the last Conv3DTranspose's volume, nvsmall_1025x321_net.cpp:401-425)
// reduces over the axis that layer's kernel walks: when the plan has the fused form (rt_conv_plan_set_softarg), the launch writes the
// (1, H, W) map itself and the volume tensor -- 127 MB per pair at 1025 x 321, written once and read once -- is never allocated.
void EngineImpl::fuseSoftargmax3D() {
    if (knob("RT_NO_SOFTARG_FUSE") || knob("RT_NO_FUSION")) return;
    for (size_t si = 0; si < ops_.size(); si++) {
        const Op sa = ops_[si];
        if (sa.kind != OpKind::kPlugin || sa.half_kind || sa.in.size() != 1) continue;
        IStereoPlugin* sp = sa.plugin ? dynamic_cast<IStereoPlugin*>(sa.plugin) : nullptr;
        if (!sp || sp->kind() != Kind::kSoftargmax) continue;
        const int vol = root(sa.in[0]);
        const TensorImpl& vt = *tensors_[vol];
        if (vt.is_output || vt.is_input || vt.f16 || vt.il8 || vt.pitch || vt.bstride || vt.alias_of >= 0) continue;
        int producer = -1, uses = 0;
        for (size_t j = 0; j < ops_.size(); j++) {
            if (j == si) continue;
            for (int x : ops_[j].in) uses += root(x) == vol;
            if (ops_[j].resid >= 0 && root(ops_[j].resid) == vol) uses++;
            if (root(ops_[j].out) == vol) { uses += producer >= 0; producer = (int)j; }
        }
        if (uses != 0 || producer < 0 || producer > (int)si) continue;
        Op& conv = ops_[producer];
        if (conv.kind != OpKind::kConv3D || !conv.splugin || conv.splugin->kind() != Kind::kConv3DTranspose || conv.stream != sa.stream) continue;
        // the soft-argmax must reduce exactly the volume this launch produces, (D, 1, H, W): root() also passes through reshaping aliases
        const Dims vd = tensors_[conv.out]->dims, sd = tensors_[sa.in[0]]->dims, md = tensors_[sa.out]->dims;
        if (vd.nbDims != 4 || vd.d[1] != 1 || sd.nbDims != 4 || sd.d[0] != vd.d[0] || sd.d[1] != 1 || sd.d[2] != vd.d[2] || sd.d[3] != vd.d[3] ||
            md.d[md.nbDims - 2] != vd.d[2] || md.d[md.nbDims - 1] != vd.d[3]) continue;
        const int mode = sp->softargmaxType() == SoftargmaxType::kMin ? 2 : 1;
        if (!conv.splugin->setSoftarg(mode)) continue;     // no fused form for this plan
        conv.softarg = mode;
        conv.out = sa.out;
        for (int w : sa.wait_on) conv.wait_on.push_back(root(w));
        conv.lays.insert(conv.lays.end(), sa.lays.begin(), sa.lays.end());
        const std::string sname = sa.name;
        ops_.erase(ops_.begin() + si);
        log_.log(ILogger::Severity::kINFO, (sname + ": folded into " + ops_[producer].name + " (the volume is never materialised)").c_str());
        si--;
    }
}

void EngineImpl::assignInterleaved3D() {
    if (knob("RT_NO_IL8") || knob("RT_NO_IL8_3D")) return;
    // half2 mode: the fp16 4-D tensors, groups of 8 channels.  fp32 engines (round 4): the fp32 4-D tensors between Conv3D launches,
    // groups of 4 -- (D, C/4, H, W, 4), the layout of the 2-D tower tensors (assignPitch) -- and the skip tensors the decoder reads.
    const bool f16_mode = half2_ && !knob("RT_NO_F16") && !knob("RT_NO_F16_3D");
    if (!f16_mode && (exact_fp32_ || knob("RT_NO_IL8_3D_F32"))) return;
    std::vector<char> il(tensors_.size(), 0);
    for (auto& t : tensors_) il[t->id] = t->f16 == f16_mode && t->dims.nbDims == 4 && !t->is_input && !t->is_output && t->alias_of < 0 && t->twin_of < 0 && !t->has_twin;
    for (auto& op : ops_) {
        if (op.kind == OpKind::kConv3D) continue;
        for (int i : op.in) il[root(i)] = 0;
        il[root(op.out)] = 0;
        if (op.resid >= 0) il[root(op.resid)] = 0;
    }
    // fp32 engines: the two feature maps of a folded cost volume ((2F, H, W) buffer, foldCostVolumes) join the negotiation when the two
    // tower launches that write them can store (2F/4, H, W, 4): the first Conv3D then gathers 16-byte slots instead of 4-byte planes
    struct Feat { int tensor; std::vector<size_t> producers; };
    std::vector<Feat> feats;
    if (!f16_mode && !knob("RT_NO_IL_FEAT_F32"))
        for (size_t ci = 0; ci < ops_.size(); ci++) {
            const Op& cv = ops_[ci];
            if (cv.kind != OpKind::kConv3D || !cv.splugin->fusion().cv_fold) continue;
            if (cv.splugin->ilCaps() & 32) continue;                    // (the factored fold reads planar maps)
            const int x = root(cv.in[0]);
            const TensorImpl& ft = *tensors_[x];
            if (ft.dims.nbDims != 3 || ft.f16 || ft.il8 || ft.pitch || ft.dims.d[0] % 8 != 0) continue;
            Feat f{x, {}};
            bool ok = true;
            for (size_t i = 0; i < ops_.size() && ok; i++) {
                const Op& op = ops_[i];
                if (i == ci) continue;
                bool reads = op.resid >= 0 && root(op.resid) == x;
                for (int t : op.in) reads = reads || root(t) == x;
                if (reads) ok = false;                                  // only the Conv3D reads the maps
                if (root(op.out) == x) { ok = ok && op.kind == OpKind::kConv && !op.twin && (rt_conv_plan_supports_il8(op.plan) & 2) != 0; f.producers.push_back(i); }
            }
            if (!ok || f.producers.size() != 2) continue;
            il[x] = 1;
            feats.push_back(f);
        }
    std::vector<char> fixed(tensors_.size(), 0);        // fp16 interleaved feature maps of a folded cost volume (assignHalf3D): decided already
    for (auto& op : ops_)
        if (op.kind == OpKind::kConv3D && op.splugin->fusion().cv_fold) {
            const int x = root(op.in[0]);
            if (tensors_[x]->f16 && tensors_[x]->il8) il[x] = fixed[x] = 1;
        }
    for (bool changed = true; changed;) {
        changed = false;
        for (auto& op : ops_) {
            if (op.kind != OpKind::kConv3D) continue;
            const int caps = op.splugin->ilCaps();
            const int x = root(op.in[0]), y = root(op.out), r = op.resid >= 0 ? root(op.resid) : -1;
            auto drop = [&](int t) {
                if (t < 0 || !il[t]) return;
                if (fixed[t]) { log_.log(ILogger::Severity::kERROR, (op.name + ": cannot read the interleaved fp16 feature maps").c_str()); ok_ = false; return; }
                il[t] = 0;
                changed = true;
            };
            if (!(caps & 1)) drop(x);
            if (!(caps & 2)) drop(y);
            if (!(caps & 4)) drop(r);
            if ((caps & 8) && !il[x]) drop(y);          // this launch writes an interleaved tensor only when it reads one
            // a Conv3D on fp16 tensors reads an interleaved skip tensor only on fp16 operands, i.e. with an interleaved input
            // (rt_conv_plan_set_layouts refuses the combination: negotiate it away instead of failing the engine)
            if (f16_mode && op.splugin->kind() == Kind::kConv3D && !il[x]) drop(r);
        }
    }
    int n = 0;
    for (auto& op : ops_) {
        if (op.kind != OpKind::kConv3D) continue;
        const int x = root(op.in[0]), y = root(op.out), r = op.resid >= 0 ? root(op.resid) : -1;
        const bool xi = il[x] != 0, yi = il[y] != 0, ri = r >= 0 && il[r] != 0;
        if (!(xi || yi || ri)) continue;
        if (!op.splugin->setLayouts(xi, yi, ri)) {
            log_.log(ILogger::Severity::kERROR, (op.name + ": " + rt_last_error_string()).c_str());
            ok_ = false;
            return;
        }
        n += yi;
    }
    if (!ok_) return;
    for (const Feat& f : feats) {
        if (!il[f.tensor]) continue;
        for (size_t i : f.producers) {
            Op& p = ops_[i];
            const TensorImpl& pin = *tensors_[root(p.in[0])];
            if (rt_conv_plan_set_layouts(p.plan, pin.il8, 1, p.resid >= 0 && tensors_[root(p.resid)]->il8) != 0) {
                log_.log(ILogger::Severity::kERROR, (p.name + ": " + rt_last_error_string()).c_str());
                ok_ = false;
                return;
            }
        }
        for (auto& t : tensors_)                                         // the maps are interleaved: so are the tensors that alias into them
            if (t->alias_of >= 0 && root(t->id) == f.tensor) t->il8 = true;
        log_.log(ILogger::Severity::kINFO, "fp32: the first Conv3D reads the two feature maps channel-interleaved (2F/4, H, W, 4)");
    }
    for (auto& t : tensors_)
        if (il[t->id]) t->il8 = true;
    if (n) log_.log(ILogger::Severity::kINFO, (std::string(f16_mode ? "half2 mode: " : "fp32: ") + std::to_string(n) + " Conv3D launches write channel-interleaved " +
                                               (f16_mode ? "(D, C/8, H, W, 8)" : "(D, C/4, H, W, 4)") + " tensors").c_str());
}

// Channel concatenation without copies (the reference concatenates left_conv1_act and the soft-argmax map into the
// 33-channel input of conv2D_1, resnet18_2D_513x257_net.cpp:612-615): when every input of a concatenation is written
// by a launch that takes an output batch stride (2-D convolution plans, the fused correlation) and read only by
// convolution plans, the inputs BECOME channel ranges of the concatenated buffer -- same planar layout and row pitch, a
// per-sample stride of the whole buffer -- and the copy launches disappear.
// Will foldConcats() fold concatenation ops_[ci]?  (Layouts aside: the caller compares those.)
bool EngineImpl::concatFoldable(size_t ci) const {
    if (knob("RT_NO_CONCAT_FOLD") || knob("RT_NO_FUSION")) return false;
    const Op& cat = ops_[ci];
    const TensorImpl& out = *tensors_[cat.out];
    if (out.is_input || out.is_output || out.alias_of >= 0 || out.dims.nbDims != 3) return false;
    bool ok = true;
    for (int i : cat.in) {
        const TensorImpl& t = *tensors_[i];
        ok = ok && !t.is_input && !t.is_output && t.alias_of < 0 && t.dims.nbDims == 3 && t.bstride == 0;
        int producers = 0;
        for (size_t oi = 0; oi < ops_.size() && ok; oi++) {
            const Op& op = ops_[oi];
            if (oi == ci) continue;
            if (op.out == i) {
                producers++;
                ok = ok && (op.kind == OpKind::kConv || op.kind == OpKind::kCorrSoftargmax) && op.stream == cat.stream && oi < ci;
            }
            if (op.resid == i) ok = ok && op.kind == OpKind::kConv && op.stream == cat.stream;
            for (int x : op.in)
                if (x == i) ok = ok && op.kind == OpKind::kConv && op.stream == cat.stream;
            if (op.kind == OpKind::kConcat)
                for (int x : op.in) ok = ok && x != i;               // member of one concatenation only
        }
        ok = ok && producers == 1;
    }
    return ok;
}

void EngineImpl::foldConcats() {
    for (size_t ci = 0; ci < ops_.size(); ci++) {
        if (ops_[ci].kind != OpKind::kConcat) continue;
        const Op cat = ops_[ci];
        TensorImpl& out = *tensors_[cat.out];
        bool ok = concatFoldable(ci);
        for (int i : cat.in) {
            const TensorImpl& t = *tensors_[i];
            ok = ok && t.il8 == out.il8 && t.f16 == out.f16 && t.pitch == out.pitch;
        }
        // readers of the concatenated tensor must come after the concatenation (they do: it is their producer)
        if (!ok) {
            if (out.il8) {          // assignPitch() only keeps a concatenation interleaved when this pass folds it (same predicate)
                log_.log(ILogger::Severity::kERROR, (cat.name + ": interleaved concatenation that cannot be folded").c_str());
                ok_ = false;
                return;
            }
            continue;
        }
        const int64_t P = out.pitch ? out.pitch : out.dims.d[2], plane = (int64_t)out.dims.d[1] * P;
        bool failed = false;
        if (out.il8) {
            // interleaved: the FIRST member hosts the whole -- its buffer is allocated for the padded channel count, its own groups come
            // first, the other members are further groups of it.  It stays a plain tensor (so that the siamese merge can pair it with
            // its right-tower twin); the concatenated tensor is the same memory under its own name.  Sample strides: applySampleStrides().
            TensorImpl& host = *tensors_[cat.in[0]];
            const int G = out.f16 ? 8 : 4;
            host.cpad = (out.dims.d[0] + G - 1) / G * G;
            out.alias_of = host.id;
            out.alias_off = 0;
            int coff = host.dims.d[0];
            for (size_t k = 1; k < cat.in.size(); k++) {
                TensorImpl& t = *tensors_[cat.in[k]];
                t.alias_of = out.id;                        // (through the whole: a reader of the whole depends on this member's producer)
                t.alias_off = (int64_t)coff * plane;        // group g of (C/4, H, pitch, 4) starts where planar channel 4 g would
                t.bstride = (int64_t)host.cpad * plane;
                coff += t.dims.d[0];
                for (auto& op : ops_) {
                    if (op.kind != OpKind::kConv) continue;
                    const int64_t xb = op.in[0] == t.id ? t.bstride : 0, yb = op.out == t.id ? t.bstride : 0, rb = op.resid == t.id ? t.bstride : 0;
                    if ((xb || yb || rb) && rt_conv_plan_set_batch_strides(op.plan, xb, yb, rb) != 0) failed = true;
                }
            }
        } else {
            int coff = 0;
            for (int i : cat.in) {
                TensorImpl& t = *tensors_[i];
                t.alias_of = out.id;
                t.alias_off = (int64_t)coff * plane;
                t.bstride = (int64_t)out.dims.d[0] * plane;
                coff += t.dims.d[0];
                for (auto& op : ops_) {
                    if (op.kind != OpKind::kConv) continue;
                    const int64_t xb = op.in[0] == i ? t.bstride : 0, yb = op.out == i ? t.bstride : 0, rb = op.resid == i ? t.bstride : 0;
                    if ((xb || yb || rb) && rt_conv_plan_set_batch_strides(op.plan, xb, yb, rb) != 0) failed = true;
                }
            }
        }
        if (failed) {
            log_.log(ILogger::Severity::kERROR, rt_last_error_string());
            ok_ = false;
            return;
        }
        ops_.erase(ops_.begin() + ci);
        ci--;
        log_.log(ILogger::Severity::kINFO, (cat.name + ": concatenation folded into its producers (no copy launch)" +
                                            (out.il8 ? ", interleaved: the first member hosts the whole" : "")).c_str());
    }
}

// Launches that touch a tensor with padded channels (the host of an interleaved concatenation, its right-tower twin, the concatenated
// tensor itself) step from sample to sample by the PADDED size.
void EngineImpl::applySampleStrides() {
    auto stride = [&](int t) -> int64_t {
        if (t < 0 || tensors_[t]->bstride) return 0;               // (a member placed inside another buffer: set where it was folded)
        while (tensors_[t]->alias_of >= 0 && tensors_[t]->alias_off == 0 && tensors_[t]->bstride == 0) t = tensors_[t]->alias_of;
        const TensorImpl& r = *tensors_[t];
        return r.cpad ? sampleElems(r) : 0;
    };
    for (auto& op : ops_) {
        if (op.kind != OpKind::kConv) continue;
        const int64_t xb = stride(op.in[0]), yb = stride(op.out), rb = stride(op.resid);
        if ((xb || yb || rb) && rt_conv_plan_set_batch_strides(op.plan, xb, yb, rb) != 0) {
            log_.log(ILogger::Severity::kERROR, rt_last_error_string());
            ok_ = false;
            return;
        }
    }
}

// ---- plan (de)serialisation ------------------------------------------------------------------------------------
// A plan is the recorded sequence of network-building calls (inputs, layers with their parameters and weights,
// plugin blobs exactly as IPlugin::serialize() wrote them, outputs) plus the builder settings.  Deserialising replays
// the calls into a fresh INetworkDefinition -- plugins come back through IPluginFactory::createPlugin(layerName,
// blob, size), as TensorRT does it (sample_app/main.cpp:198-220, lib/internal_utils.cpp:289-313) -- and builds the
// engine again, which takes milliseconds here because there is no kernel autotuning step.  Like in the reference only
// networks whose plugins are all serialisable have a plan (ELU / cost volume / soft-argmax: ResNet-18 2D); the 3-D
// convolution plugins are not serialisable there either (lib/conv3d_plugin.cpp: getSerializationSize() == 0).
namespace plan {
constexpr char kMagic[8] = {'R', 'T', 'S', 'D', 'P', 'L', 'N', '1'};
struct Writer {
    std::string out;
    void raw(const void* p, size_t n) { out.append(static_cast<const char*>(p), n); }
    template <typename T> void pod(T v) { raw(&v, sizeof(T)); }
    void str(const std::string& s) { pod<uint32_t>((uint32_t)s.size()); raw(s.data(), s.size()); }
    void dims(const Dims& d) { pod<int32_t>(d.nbDims); for (int i = 0; i < 8; i++) pod<int32_t>(d.d[i]); }
    void hw(const DimsHW& d) { pod<int32_t>(d.h()); pod<int32_t>(d.w()); }
    void weights(const Weights& w) {
        pod<int32_t>((int32_t)w.type);
        pod<int64_t>(w.count);
        raw(w.values, (size_t)w.count * (w.type == DataType::kHALF ? 2 : 4));
    }
};
struct Reader {
    const char* p; const char* end; bool ok = true;
    bool need(size_t n) { if ((size_t)(end - p) < n) ok = false; return ok; }
    template <typename T> T pod() { T v{}; if (need(sizeof(T))) { std::memcpy(&v, p, sizeof(T)); p += sizeof(T); } return v; }
    std::string str() { uint32_t n = pod<uint32_t>(); std::string s; if (need(n)) { s.assign(p, n); p += n; } return s; }
    Dims dims() { Dims d{}; d.nbDims = pod<int32_t>(); for (int i = 0; i < 8; i++) d.d[i] = pod<int32_t>(); return d; }
    DimsHW hw() { int h = pod<int32_t>(), w = pod<int32_t>(); return DimsHW{h, w}; }
};
}  // namespace plan

IHostMemory* EngineImpl::serialize() const {
    plan::Writer w;
    w.raw(plan::kMagic, 8);
    w.pod<int32_t>(max_batch_);
    w.pod<uint8_t>(half2_ ? 1 : 0);
    std::vector<const TensorImpl*> inputs, outputs;
    for (int b : bindings_) (tensors_[b]->is_input ? inputs : outputs).push_back(tensors_[b].get());
    // tensor numbers in the plan are canonical (inputs, then layer outputs in layer order), so a plan survives a
    // deserialise / serialise round trip byte for byte
    std::map<int, int32_t> num;
    for (auto* t : inputs) num.emplace(t->id, (int32_t)num.size());
    for (auto& l : layers_)
        for (auto* t : l->out) num.emplace(t->id, (int32_t)num.size());
    auto id_of = [&](const TensorImpl* t) { return num.at(t->id); };
    w.pod<uint32_t>((uint32_t)inputs.size());
    for (auto* t : inputs) { w.pod<int32_t>(id_of(t)); w.str(t->name); w.dims(t->dims); }
    w.pod<uint32_t>((uint32_t)layers_.size());
    for (auto& l : layers_) {
        w.pod<int32_t>((int32_t)l->type);
        w.str(l->name);
        w.pod<uint32_t>((uint32_t)l->in.size());
        for (auto* t : l->in) w.pod<int32_t>(id_of(t));
        w.pod<uint32_t>((uint32_t)l->out.size());
        for (auto* t : l->out) { w.pod<int32_t>(id_of(t)); w.str(t->name); }
        switch (l->type) {
            case LayerType::kCONVOLUTION:
            case LayerType::kDECONVOLUTION:
                w.pod<int32_t>(l->nb_maps); w.hw(l->ksize); w.hw(l->stride); w.hw(l->padding);
                w.weights(l->kernel); w.weights(l->bias);
                break;
            case LayerType::kSCALE:
                w.pod<int32_t>((int32_t)l->scale_mode); w.weights(l->shift); w.weights(l->scale); w.weights(l->power);
                break;
            case LayerType::kACTIVATION: w.pod<int32_t>((int32_t)l->act); break;
            case LayerType::kELEMENTWISE: w.pod<int32_t>((int32_t)l->ew); break;
            case LayerType::kCONCATENATION: break;
            case LayerType::kPADDING: w.hw(l->pre_pad); w.hw(l->post_pad); break;
            case LayerType::kSHUFFLE: w.dims(l->reshape); break;
            case LayerType::kPLUGIN: {
                const size_t n = l->plugin->getSerializationSize();
                if (!n) {
                    log_.log(ILogger::Severity::kERROR, (l->name + ": plugin is not serialisable, this engine has no plan "
                             "(same as the reference's Conv3D / Transform / Pad / Slice plugins)").c_str());
                    return nullptr;
                }
                std::string blob(n, '\0');
                l->plugin->serialize(&blob[0]);
                w.pod<uint8_t>(l->plugin_ext ? 1 : 0);
                w.str(blob);
                break;
            }
            default:
                log_.log(ILogger::Severity::kERROR, (l->name + ": layer type cannot be serialised").c_str());
                return nullptr;
        }
    }
    w.pod<uint32_t>((uint32_t)outputs.size());
    for (auto* t : outputs) w.pod<int32_t>(id_of(t));
    auto* m = new HostMemory();
    m->blob = std::move(w.out);
    return m;
}

// ---- execution -----------------------------------------------------------------------------------------------
ContextImpl::ContextImpl(EngineImpl& e) : eng_(e) {
    buffers_.assign(e.tensors_.size(), nullptr);
    events_.assign(e.tensors_.size(), nullptr);
    debug_sync_ = knob("RT_DEBUG_SYNC") != nullptr;      // serialise every launch (A/B measurements only)
}

ContextImpl::~ContextImpl() {
    rt_stream_sync(nullptr);
    for (void* b : buffers_)
        if (b) rt_free(b);
    if (workspace_) rt_free(workspace_);
    if (trace_dev_) rt_free(trace_dev_);
    for (auto& kv : half_buf_)
        if (kv.second.first) rt_free(kv.second.first);
    for (void* ev : events_)
        if (ev) rt_event_destroy(ev);
    for (void* ev : prof_events_)
        if (ev) rt_event_destroy(ev);
    if (ev_fork_) rt_event_destroy(ev_fork_);
    if (ev_join_) rt_event_destroy(ev_join_);
    if (ev_null_) rt_event_destroy(ev_null_);
    for (auto& g : graphs_) rt_graph_destroy(g.second);
    if (side_stream_) rt_stream_destroy(side_stream_);
    if (main_stream_) rt_stream_destroy(main_stream_);
}

const ICudaEngine& ContextImpl::getEngine() const { return eng_; }

// A captured graph bakes in the addresses of the internal buffers: whenever one of them is freed or reallocated, every captured pass
// is stale and must be captured again (batch 1, batch 2, batch 1 in graph mode replayed freed memory).
void ContextImpl::dropGraphs() {
    for (auto& g : graphs_)
        if (g.second) rt_graph_destroy(g.second);
    graphs_.clear();
    graph_failed_.clear();
}

void* ContextImpl::halfBuffer(int op, int slot, size_t bytes) {
    auto& e = half_buf_[{op, slot}];
    if (e.second < bytes) {
        if (e.first) { rt_stream_sync(nullptr); dropGraphs(); }
        if (e.first) rt_free(e.first);
        e.first = nullptr;
        e.second = 0;
        if (rt_malloc(&e.first, bytes) != 0) return nullptr;
        e.second = bytes;
    }
    return e.first;
}

bool ContextImpl::ensureBuffers(int batch) {
    if (batch <= alloc_batch_) return true;
    if (alloc_batch_ > 0) {             // growing: earlier passes may still be in flight on the old buffers, captured graphs point at them
        rt_stream_sync(nullptr);
        if (main_stream_) rt_stream_sync(main_stream_);
        if (side_stream_) rt_stream_sync(side_stream_);
        dropGraphs();
    }
    for (void*& b : buffers_) {
        if (b) rt_free(b);
        b = nullptr;
    }
    for (auto& t : eng_.tensors_) {
        if (t->is_input || t->is_output || t->alias_of >= 0 || t->twin_of >= 0) continue;     // a right-tower twin lives in its left twin's buffer
        bool used = false;
        for (auto& op : eng_.ops_) {
            used |= op.out == t->id || op.resid == t->id;
            for (int i : op.in) used |= i == t->id;
        }
        if (!used) continue;
        const size_t bytes = (size_t)EngineImpl::sampleElems(*t) * (t->f16 ? 2 : 4) * (size_t)batch * (t->has_twin ? 2 : 1);
        if (rt_malloc(&buffers_[t->id], bytes) != 0) {
            eng_.log_.log(ILogger::Severity::kERROR, (std::string("engine: device allocation failed: ") + rt_last_error_string()).c_str());
            return false;
        }
        // padded channels are read (16-byte groups) and meet zero weights: they must be finite from the first pass on
        if (t->cpad && (rt_memset(buffers_[t->id], 0, bytes, nullptr) != 0 || rt_stream_sync(nullptr) != 0)) return false;
    }
    if (eng_.workspace_bytes_ && !workspace_) {
        if (rt_malloc(&workspace_, 2 * eng_.workspace_bytes_) != 0) return false;        // one block per stream of the context (main, side)
        workspace_bytes_ = eng_.workspace_bytes_;
    }
    alloc_batch_ = batch;
    return true;
}

void* ContextImpl::addr(int tensor, int batch, void** bindings) const {
    int t = tensor;
    int64_t off = 0;                   // element offset inside the buffer it aliases (folded concatenation)
    while (eng_.tensors_[t]->alias_of >= 0) { off += eng_.tensors_[t]->alias_off; t = eng_.tensors_[t]->alias_of; }
    if (eng_.tensors_[t]->twin_of >= 0) {      // right-tower twin: `batch` samples behind the left tensor's first sample
        const TensorImpl& r = *eng_.tensors_[t];
        off += (int64_t)batch * EngineImpl::sampleElems(r);
        t = r.twin_of;
    }
    const TensorImpl& ti = *eng_.tensors_[t];
    if (ti.is_input || ti.is_output) {
        for (size_t b = 0; b < eng_.bindings_.size(); b++)
            if (eng_.bindings_[b] == t) return static_cast<char*>(bindings[b]) + off * 4;
    }
    return static_cast<char*>(buffers_[t]) + off * (ti.f16 ? 2 : 4);
}

// events that only order streams carry no timestamps (RT_SYNC_EVENTS_TIMED=1: the round-1 behaviour, for A/B timing)
static void order_event(void** ev) {
    static const bool timed = knob("RT_SYNC_EVENTS_TIMED") && atoi(knob("RT_SYNC_EVENTS_TIMED")) != 0;
    if (timed) rt_event_create(ev); else rt_event_create_ordering(ev);
}

bool ContextImpl::execute(int batchSize, void** bindings) {
    if (!main_stream_ && rt_stream_create(&main_stream_) != 0) return false;
    // execute() is the synchronous API: whatever the caller queued on the NULL stream (uploads, fills) must be complete before
    // our own non-blocking streams touch the bindings -- ordered on the device (an event of the NULL stream), not by blocking
    // the host before the first launch is issued
    if (!ev_null_) order_event(&ev_null_);
    if (rt_event_record(ev_null_, nullptr) != 0 || rt_stream_wait_event(main_stream_, ev_null_) != 0) return false;
    return run(batchSize, bindings, (cudaStream_t)main_stream_, true);
}

bool ContextImpl::enqueue(int batchSize, void** bindings, cudaStream_t stream, cudaEvent_t* inputConsumed) {
    bool ok = run(batchSize, bindings, stream, false);
    if (ok && inputConsumed && *inputConsumed) ok = rt_event_record(*inputConsumed, stream) == 0;
    return ok;
}

// Debug mode (setDebugSync(true)): the input of every launch that multiplies on the fp16 matrix pipe is scanned for values outside the
// domain of the fp16 split (|x| >= 65504, inf, NaN) BEFORE the launch, so that a violation is reported with the layer's name instead of
// surfacing as a NaN disparity many layers later.  (The reference's cuDNN / TensorRT fp32 kernels have no such domain.)
bool ContextImpl::checkInputRange(const Op& op, int batch, void** bindings, rtStream st) {
    if (op.kind != OpKind::kConv || !op.plan) return true;
    float limit = 0.f;
    if (rt_conv_plan_input_limit(op.plan, &limit) != 0 || !(limit < 3e38f)) return true;
    std::vector<int> ins{op.in[0]};
    if (op.twin) ins.push_back(op.twin_in[0]);
    for (int tin : ins) {
        int t = tin;
        while (eng_.tensors_[t]->alias_of >= 0 && eng_.tensors_[t]->alias_off == 0 && eng_.tensors_[t]->bstride == 0) t = eng_.tensors_[t]->alias_of;
        const TensorImpl& ti = *eng_.tensors_[t];
        if (ti.alias_of >= 0 || ti.bstride != 0 || ti.dims.nbDims != 3) continue;       // a channel range of another buffer: checked where it is produced
        // (a pre-split tensor is scanned as the fp16 values it holds: 16 per pixel and group of 8 channels)
        // the host of an interleaved concatenation: a reader of the host itself is checked on the host's own channels (the groups behind
        // them belong to other members and may not be written yet), a reader of the WHOLE on every group (the disparity map of conv2D_1's
        // input is produced by the correlation, which no other check sees; padding lanes are zeros)
        const bool whole = ti.cpad && tin != t && eng_.tensors_[tin]->dims.d[0] > ti.dims.d[0];
        const int C = whole ? ti.cpad : (ti.split ? 2 * ti.dims.d[0] : ti.dims.d[0]), H = ti.dims.d[1], W = ti.dims.d[2], P = ti.pitch ? ti.pitch : W,
                  G = ti.split ? 16 : (ti.il8 ? (ti.f16 ? 8 : 4) : 1);
        float mx = 0.f;
        int64_t bad = 0;
        // (sample by sample: the samples of a padded tensor are cpad channels apart)
        const int calls = ti.cpad ? batch : 1;
        const int64_t rows = (int64_t)(ti.cpad ? 1 : batch) * (C / G) * H;
        for (int n = 0; n < calls; n++) {
            float m1 = 0.f;
            int64_t b1 = 0;
            const char* px = static_cast<const char*>(addr(tin, batch, bindings)) + (size_t)n * EngineImpl::sampleElems(ti) * (ti.f16 ? 2 : 4);
            if (rt_check_range(px, rows, (int64_t)G * W, (int64_t)G * P, (ti.f16 || ti.split) ? RT_F16 : RT_F32, limit, &m1, &b1, st) != 0) {
                eng_.log_.log(ILogger::Severity::kERROR, (op.name + ": range check failed: " + rt_last_error_string()).c_str());
                return false;
            }
            mx = std::max(mx, m1);
            bad += b1;
        }
        if (bad) {
            char msg[256];
            std::snprintf(msg, sizeof msg, ": input '%s' leaves the domain of the fp16-split convolution: %lld value(s) with |x| >= %.0f or non-finite, max finite |x| = %g "
                          "(IBuilder::setExactFp32Mode keeps the engine on fp32 kernels)", ti.name.c_str(), (long long)bad, limit, mx);
            eng_.log_.log(ILogger::Severity::kERROR, (op.name + msg).c_str());
            return false;
        }
    }
    return true;
}

// Graph mode (setGraphMode): the first pass over a set of bindings runs as usual (buffers, streams and events come into being), the
// second is captured -- the side stream joins the capture through the fork event and leaves it through the join event -- and from then
// on a pass is one rt_graph_launch.  Profiling and debug mode keep the launch-by-launch path.
bool ContextImpl::run(int batch, void** bindings, cudaStream_t stream, bool sync) {
    if (!graph_mode_ || profiler_ || debug_sync_ || trace_ || !stream) return issue(batch, bindings, stream, sync, false);
    std::vector<uintptr_t> key{(uintptr_t)stream, (uintptr_t)batch, (uintptr_t)streams_, (uintptr_t)sync};
    for (int b = 0; b < eng_.getNbBindings(); b++) key.push_back((uintptr_t)bindings[b]);
    auto it = graphs_.find(key);
    if (it != graphs_.end() && it->second) {
        if (rt_graph_launch(it->second, (rtStream)stream) != 0) {
            eng_.log_.log(ILogger::Severity::kERROR, (std::string("graph launch failed: ") + rt_last_error_string()).c_str());
            return false;
        }
        return !sync || rt_stream_sync((rtStream)stream) == 0;
    }
    if (it == graphs_.end() || graph_failed_.count(key)) {       // first pass with these bindings (or graphs are not available): launch by launch
        if (it == graphs_.end()) graphs_[key] = nullptr;
        return issue(batch, bindings, stream, sync, false);
    }
    if (rt_graph_begin_capture((rtStream)stream) != 0) {
        eng_.log_.log(ILogger::Severity::kWARNING, (std::string("graph mode: capture unavailable (") + rt_last_error_string() + "), launching directly").c_str());
        graph_failed_.insert(key);
        return issue(batch, bindings, stream, sync, false);
    }
    const bool issued = issue(batch, bindings, stream, sync, true);
    rtGraph* g = nullptr;
    const int rc = rt_graph_end_capture((rtStream)stream, &g);
    if (!issued || rc != 0 || !g) {
        eng_.log_.log(ILogger::Severity::kWARNING, (std::string("graph mode: capture failed (") + rt_last_error_string() + "), launching directly").c_str());
        graph_failed_.insert(key);
        return issue(batch, bindings, stream, sync, false);
    }
    it->second = g;
    if (rt_graph_launch(g, (rtStream)stream) != 0) return false;
    return !sync || rt_stream_sync((rtStream)stream) == 0;
}

bool ContextImpl::issue(int batch, void** bindings, cudaStream_t stream, bool sync, bool capturing) {
    ILogger& log = eng_.log_;
    if (batch < 1 || batch > eng_.max_batch_) {
        log.log(ILogger::Severity::kERROR, "execute: batch size exceeds the engine's max batch size");
        return false;
    }
    if (!ensureBuffers(batch)) return false;
    if (trace_ && !capturing) {
        if (!trace_dev_ && rt_malloc(&trace_dev_, 8 * eng_.ops_.size() + 8) != 0) return false;
        trace_ptr_.assign(eng_.ops_.size(), nullptr);
        trace_stream_ = stream;
        trace_batch_ = batch;
    }
    const bool profile = profiler_ != nullptr;
    const bool two = eng_.two_streams_ && streams_ > 1;
    rtStream main = stream;
    if (two) {
        if (!side_stream_ && rt_stream_create(&side_stream_) != 0) return false;
        if (!ev_fork_) order_event(&ev_fork_);
        if (!ev_join_) order_event(&ev_join_);
        // the side stream must not start before work already queued on the caller's stream
        // (e.g. the H2D copies of the inputs) is done
        rt_event_record(ev_fork_, main);
        rt_stream_wait_event(side_stream_, ev_fork_);
    }
    // profiling keeps the production schedule (both streams): one event pair per launch, recorded on the
    // stream the launch goes to and read back after the final synchronisation
    if (profile && prof_events_.size() != 2 * eng_.ops_.size()) {
        prof_events_.assign(2 * eng_.ops_.size(), nullptr);
        for (void*& e : prof_events_) rt_event_create(&e);
    }
    // issue order: see assignStreams(); event pairs and half buffers are indexed by the op's position in ops_
    const int want = knob("RT_INTERLEAVE") ? atoi(knob("RT_INTERLEAVE")) : -1;     // A/B: 0 never, 1 always
    const bool interleave = two && eng_.issue_sync_.size() == eng_.ops_.size() && (want < 0 ? sync : want != 0);

    bool ok = true;
    for (size_t k = 0; k < eng_.ops_.size(); k++) {
        const size_t op_index = interleave ? eng_.issue_sync_[k] : k;
        const Op& op = eng_.ops_[op_index];
        rtStream st = (two && op.stream == 1) ? side_stream_ : main;
        if (two)
            for (int w : op.wait_on) {
                if (!events_[w]) {          // the producer on the other stream must have been issued (and its event recorded) before
                    log.log(ILogger::Severity::kERROR, (op.name + ": cross-stream input '" + eng_.tensors_[w]->name + "' has no recorded event").c_str());
                    return false;
                }
                rt_stream_wait_event(st, events_[w]);
            }
        if (debug_sync_ && !checkInputRange(op, batch, bindings, st)) {
            ok = false;
            break;
        }
        if (profile) rt_event_record(prof_events_[2 * op_index], st);
        const TensorImpl& out = *eng_.tensors_[op.out];
        void* y = addr(op.out, batch, bindings);
        int rc = 0;
        switch (op.kind) {
            case OpKind::kConv:
                // a merged siamese launch covers the left samples and, right behind them, the right samples
                if (op.twin_bind)       // first layers of both towers: left images | right images from their own bindings
                    rc = rt_conv_enqueue_twin_input(op.plan, addr(op.in[0], batch, bindings), addr(op.twin_in[0], batch, bindings), y, batch, st,
                                                    streams_ == 1 ? RT_HINT_THROUGHPUT : 0);
                else
                    rc = rt_conv_enqueue_hint(op.plan, addr(op.in[0], batch, bindings), y, op.resid >= 0 ? addr(op.resid, batch, bindings) : nullptr,
                                              op.twin ? 2 * batch : batch, st, streams_ == 1 ? RT_HINT_THROUGHPUT : 0);
                break;
            case OpKind::kConv3D:
                if (op.softarg != op.splugin->softarg()) {      // (the plan was rebuilt and lost the reduction: its output would not fit the map's buffer)
                    eng_.log_.log(ILogger::Severity::kERROR, (op.name + ": the plan no longer ends in the soft-argmax the executor planned for").c_str());
                    return false;
                }
                rc = op.splugin->enqueueFused(batch, addr(op.in[0], batch, bindings), y, op.resid >= 0 ? addr(op.resid, batch, bindings) : nullptr,
                                              wsOf(st == side_stream_ && two), workspace_bytes_, (cudaStream_t)st);
                break;
            case OpKind::kPlugin: {
                const void* ins[8];
                for (size_t i = 0; i < op.in.size() && i < 8; i++) ins[i] = addr(op.in[i], batch, bindings);
                void* outs[1] = {y};
                if (op.half_kind) {         // fp32 tensors <-> the fp16 format the plugin was configured with
                    auto halfs = [&](const Dims& d) { return (size_t)(op.half_kind == 2 ? (d.d[0] + 1) / 2 * 2 : d.d[0]) * (volume(d) / d.d[0]); };
                    for (size_t i = 0; i < op.in.size() && i < 8 && rc == 0; i++) {
                        const Dims& d = eng_.tensors_[op.in[i]]->dims;
                        void* hb = halfBuffer((int)op_index, (int)i, halfs(d) * 2 * (size_t)batch);
                        rc = hb ? rt_convert_format(ins[i], hb, batch, d.d[0], (int64_t)(volume(d) / d.d[0]), 0, op.half_kind, st) : RT_E_NOMEM;
                        ins[i] = hb;
                    }
                    void* ho = rc == 0 ? halfBuffer((int)op_index, 8, halfs(out.dims) * 2 * (size_t)batch) : nullptr;
                    if (rc == 0 && !ho) rc = RT_E_NOMEM;
                    if (rc == 0) {
                        outs[0] = ho;
                        rc = op.plugin->enqueue(batch, ins, outs, wsOf(st == side_stream_ && two), (cudaStream_t)st);
                    }
                    if (rc == 0) rc = rt_convert_format(ho, y, batch, out.dims.d[0], (int64_t)(volume(out.dims) / out.dims.d[0]), op.half_kind, 0, st);
                    break;
                }
                rc = op.plugin->enqueue(batch, ins, outs, wsOf(st == side_stream_ && two), (cudaStream_t)st);
                break;
            }
            case OpKind::kAdd:
                rc = rt_add_act(addr(op.in[0], batch, bindings), addr(op.in[1], batch, bindings), y,
                                (int64_t)volume(out.dims) * batch, op.act, RT_F32, st);
                break;
            case OpKind::kAct:
                rc = rt_activation(addr(op.in[0], batch, bindings), y, (int64_t)volume(out.dims) * batch, op.act, RT_F32, st);
                break;
            case OpKind::kConcat: {
                const int ctot = out.dims.d[0];
                const int64_t inner = out.pitch ? (int64_t)out.dims.d[1] * out.pitch : (int64_t)volume(out.dims) / ctot;
                int coff = 0;
                for (int i : op.in) {
                    const Dims& d = eng_.tensors_[i]->dims;
                    if (rc == 0)
                        rc = rt_concat_channels(addr(i, batch, bindings), y, batch, d.d[0], ctot, coff, inner, out.f16 ? RT_F16 : RT_F32, st);
                    coff += d.d[0];
                }
                break;
            }
            case OpKind::kCorrSoftargmax: {
                const Dims& f = eng_.tensors_[op.in[0]]->dims;
                if (op.il_in && eng_.tensors_[op.in[0]]->f16)
                    rc = rt_corr_softargmax_il8_f16(addr(op.in[0], batch, bindings), addr(op.in[1], batch, bindings), y, batch, f.d[0], f.d[1], f.d[2],
                                                    op.max_disp, op.is_min, eng_.tensors_[op.in[0]]->pitch, out.pitch, out.bstride, op.il_out ? 8 : 1, st);
                else if (op.il_in)
                    rc = rt_corr_softargmax_il_slot(addr(op.in[0], batch, bindings), addr(op.in[1], batch, bindings), y, batch, f.d[0], f.d[1],
                                                    f.d[2], op.max_disp, op.is_min, eng_.tensors_[op.in[0]]->pitch, out.pitch, out.bstride,
                                                    op.il_out ? 4 : 1, st);
                else
                    rc = rt_corr_softargmax_pitched(addr(op.in[0], batch, bindings), addr(op.in[1], batch, bindings), y, batch, f.d[0],
                                                    f.d[1], f.d[2], op.max_disp, op.is_min, eng_.tensors_[op.in[0]]->pitch, out.pitch,
                                                    out.bstride, out.f16 ? RT_F16 : RT_F32, st);
                break;
            }
            case OpKind::kCopy:
                break;
        }
        if (rc != 0) {
            log.log(ILogger::Severity::kERROR, (op.name + ": launch failed: " + rt_last_error_string()).c_str());
            ok = false;
            break;
        }
        if (profile) rt_event_record(prof_events_[2 * op_index + 1], st);
        if (trace_ && !capturing) {     // hash the output where it was produced: same stream, right behind the launch
            trace_ptr_[op_index] = y;
            if (rt_hash_buffer(y, outBytes(op, batch), static_cast<unsigned long long*>(trace_dev_) + op_index, st) != 0) {
                log.log(ILogger::Severity::kERROR, (op.name + ": launch trace failed: " + rt_last_error_string()).c_str());
                ok = false;
                break;
            }
        }
        if (two && op.publish) {       // make the result visible to the consumer on the other stream
            int r = op.out;
            while (eng_.tensors_[r]->alias_of >= 0) r = eng_.tensors_[r]->alias_of;
            if (!events_[r]) order_event(&events_[r]);
            rt_event_record(events_[r], st);
            if (op.twin_out >= 0) {
                int r2 = op.twin_out;
                while (eng_.tensors_[r2]->alias_of >= 0) r2 = eng_.tensors_[r2]->alias_of;
                if (!events_[r2]) order_event(&events_[r2]);
                rt_event_record(events_[r2], st);
            }
        }
        if (debug_sync_) rt_stream_sync(st);
    }
    if (two) {
        rt_event_record(ev_join_, side_stream_);
        rt_stream_wait_event(main, ev_join_);
    }
    if ((sync || profile) && !capturing) ok = (rt_stream_sync(main) == 0) && ok;
    if (profile && ok) {
        for (size_t i = 0; i < eng_.ops_.size(); i++) {
            float ms = 0.f;
            rt_event_elapsed_ms(prof_events_[2 * i], prof_events_[2 * i + 1], &ms);
            profiler_->reportLayerTime(eng_.ops_[i].name.c_str(), ms);
            if (knob("RT_PROFILE_TIMELINE")) {            // start/end of every launch relative to the first one
                float t0 = 0.f, t1 = 0.f;
                rt_event_elapsed_ms(prof_events_[0], prof_events_[2 * i], &t0);
                rt_event_elapsed_ms(prof_events_[0], prof_events_[2 * i + 1], &t1);
                std::fprintf(stderr, "timeline s%d %9.1f %9.1f  %s\n", eng_.ops_[i].stream, t0 * 1e3f, t1 * 1e3f, eng_.ops_[i].name.c_str());
            }
        }
    }
    return ok;
}

// ---- launch trace (IExecutionContext::setLaunchTrace) --------------------------------------------------------------------------
// bytes of a launch's output as it lies in memory: the dense (or pitched) tensor for `batch` samples -- both towers' samples for a merged
// siamese launch; a tensor placed inside another buffer with its own sample stride (folded concatenation) is covered for its first
// sample only (the samples are not contiguous there)
size_t ContextImpl::outBytes(const Op& op, int batch) const {
    const TensorImpl& t = *eng_.tensors_[op.out];
    const size_t elems = t.pitch ? (size_t)t.dims.d[0] * t.dims.d[1] * t.pitch : (size_t)volume(t.dims);
    const size_t es = (t.f16 && !t.is_output && !t.is_input) ? 2 : 4;
    size_t n = elems * es;
    if (t.bstride == 0 && t.alias_of < 0 && t.cpad == 0) n *= (size_t)batch * (op.twin ? 2 : 1);
    return n & ~(size_t)3;
}

int ContextImpl::readLaunchTrace(unsigned long long* hashes, int max) {
    if (!trace_dev_ || !hashes || max <= 0) return 0;
    if (trace_stream_) rt_stream_sync(trace_stream_);
    else if (main_stream_) rt_stream_sync(main_stream_);
    const int n = std::min<int>(max, (int)eng_.ops_.size());
    if (rt_memcpy_d2h(hashes, trace_dev_, 8 * (size_t)n, nullptr) != 0 || rt_stream_sync(nullptr) != 0) return -1;
    return n;
}

const char* ContextImpl::getLaunchName(int i) const {
    return (i >= 0 && i < (int)eng_.ops_.size()) ? eng_.ops_[i].name.c_str() : nullptr;
}

long long ContextImpl::readLaunchOutput(int launch, void* host, long long bytes) {
    if (launch < 0 || launch >= (int)eng_.ops_.size() || launch >= (int)trace_ptr_.size() || !trace_ptr_[launch]) return -1;
    const long long n = (long long)outBytes(eng_.ops_[launch], trace_batch_);
    if (!host) return n;
    if (bytes < n) return -1;
    if (trace_stream_) rt_stream_sync(trace_stream_);
    if (rt_memcpy_d2h(host, trace_ptr_[launch], (size_t)n, nullptr) != 0 || rt_stream_sync(nullptr) != 0) return -1;
    return n;
}

// ---- builder / runtime ------------------------------------------------------------------------------------------
class BuilderImpl : public IBuilder {
public:
    explicit BuilderImpl(ILogger& log) : log_(log) {}
    INetworkDefinition* createNetwork() override { return new NetworkImpl(log_); }
    void setMaxBatchSize(int b) override { max_batch_ = b; }
    int getMaxBatchSize() const override { return max_batch_; }
    void setMaxWorkspaceSize(std::size_t w) override { workspace_ = w; }
    std::size_t getMaxWorkspaceSize() const override { return workspace_; }
    void setHalf2Mode(bool m) override { half2_ = m; }
    bool getHalf2Mode() const override { return half2_; }
    void setExactFp32Mode(bool m) override { exact_fp32_ = m; }
    bool getExactFp32Mode() const override { return exact_fp32_; }
    void setDebugSync(bool s) override { debug_sync_ = s; }
    bool getDebugSync() const override { return debug_sync_; }
    void setMinFindIterations(int v) override { min_find_ = v; }
    int getMinFindIterations() const override { return min_find_; }
    void setAverageFindIterations(int v) override { avg_find_ = v; }
    int getAverageFindIterations() const override { return avg_find_; }
    bool platformHasFastFp16() const override { return true; }      // CDNA4: fp16 MFMA at 16x the fp32 rate
    bool platformHasFastInt8() const override { return true; }
    ICudaEngine* buildCudaEngine(INetworkDefinition& network) override {
        auto* e = new EngineImpl(static_cast<NetworkImpl&>(network), max_batch_, half2_, log_, exact_fp32_);
        if (!e->ok()) {
            log_.log(ILogger::Severity::kERROR, "buildCudaEngine failed");
            delete e;
            return nullptr;
        }
        return e;
    }
    void destroy() override { delete this; }

private:
    ILogger& log_;
    int max_batch_ = 1, min_find_ = 1, avg_find_ = 1;
    std::size_t workspace_ = 0;
    bool half2_ = false, debug_sync_ = false, exact_fp32_ = false;
};

class RuntimeImpl : public IRuntime {
public:
    explicit RuntimeImpl(ILogger& log) : log_(log) {}
    ICudaEngine* deserializeCudaEngine(const void* blob, std::size_t size, IPluginFactory* factory) override {
        auto fail = [&](const std::string& msg) -> ICudaEngine* {
            log_.log(ILogger::Severity::kERROR, ("deserializeCudaEngine: " + msg).c_str());
            return nullptr;
        };
        if (!blob || size < 8 || std::memcmp(blob, plan::kMagic, 8) != 0)
            return fail("not a plan of this runtime (TensorRT plan files are not portable: rebuild the engine from the network "
                        "builder, which takes milliseconds here)");
        plan::Reader r{static_cast<const char*>(blob) + 8, static_cast<const char*>(blob) + size};
        const int max_batch = r.pod<int32_t>();
        const bool half2 = r.pod<uint8_t>() != 0;
        std::unique_ptr<NetworkImpl> net(new NetworkImpl(log_));
        std::vector<std::unique_ptr<char[]>> store;
        std::map<int, ITensor*> tensor;
        auto weights = [&]() {
            Weights w{};
            w.type = (DataType)r.pod<int32_t>();
            w.count = r.pod<int64_t>();
            const size_t bytes = (size_t)w.count * (w.type == DataType::kHALF ? 2 : 4);
            if (w.count < 0 || !r.need(bytes)) { r.ok = false; w.count = 0; return w; }
            store.emplace_back(new char[bytes ? bytes : 1]);
            std::memcpy(store.back().get(), r.p, bytes);
            r.p += bytes;
            w.values = w.count ? store.back().get() : nullptr;
            return w;
        };
        const uint32_t nin = r.pod<uint32_t>();
        for (uint32_t i = 0; i < nin && r.ok; i++) {
            const int id = r.pod<int32_t>();
            const std::string name = r.str();
            const Dims d = r.dims();
            tensor[id] = net->addInput(name.c_str(), DataType::kFLOAT, d);
        }
        const uint32_t nl = r.pod<uint32_t>();
        for (uint32_t li = 0; li < nl && r.ok; li++) {
            const LayerType type = (LayerType)r.pod<int32_t>();
            const std::string name = r.str();
            std::vector<ITensor*> ins;
            const uint32_t ni = r.pod<uint32_t>();
            for (uint32_t i = 0; i < ni && r.ok; i++) {
                auto it = tensor.find(r.pod<int32_t>());
                if (it == tensor.end()) return fail(name + ": input tensor defined after use");
                ins.push_back(it->second);
            }
            std::vector<std::pair<int, std::string>> outs;
            const uint32_t no = r.pod<uint32_t>();
            for (uint32_t i = 0; i < no && r.ok; i++) { const int id = r.pod<int32_t>(); outs.emplace_back(id, r.str()); }
            if (!r.ok || ins.empty()) return fail("truncated plan");
            ILayer* layer = nullptr;
            switch (type) {
                case LayerType::kCONVOLUTION:
                case LayerType::kDECONVOLUTION: {
                    const int maps = r.pod<int32_t>();
                    const DimsHW k = r.hw(), st = r.hw(), pad = r.hw();
                    const Weights kw = weights(), bw = weights();
                    if (!r.ok) break;
                    if (type == LayerType::kCONVOLUTION) {
                        auto* c = net->addConvolution(*ins[0], maps, k, kw, bw);
                        c->setStride(st); c->setPadding(pad);
                        layer = c;
                    } else {
                        auto* c = net->addDeconvolution(*ins[0], maps, k, kw, bw);
                        c->setStride(st); c->setPadding(pad);
                        layer = c;
                    }
                    break;
                }
                case LayerType::kSCALE: {
                    const ScaleMode mode = (ScaleMode)r.pod<int32_t>();
                    const Weights a = weights(), b = weights(), c = weights();
                    if (r.ok) layer = net->addScale(*ins[0], mode, a, b, c);
                    break;
                }
                case LayerType::kACTIVATION: layer = net->addActivation(*ins[0], (ActivationType)r.pod<int32_t>()); break;
                case LayerType::kELEMENTWISE:
                    if (ins.size() != 2) return fail(name + ": element-wise layer needs two inputs");
                    layer = net->addElementWise(*ins[0], *ins[1], (ElementWiseOperation)r.pod<int32_t>());
                    break;
                case LayerType::kCONCATENATION: layer = net->addConcatenation(ins.data(), (int)ins.size()); break;
                case LayerType::kPADDING: { const DimsHW a = r.hw(), b = r.hw(); layer = net->addPadding(*ins[0], a, b); break; }
                case LayerType::kSHUFFLE: {
                    auto* sh = net->addShuffle(*ins[0]);
                    sh->setReshapeDimensions(r.dims());
                    layer = sh;
                    break;
                }
                case LayerType::kPLUGIN: {
                    const bool ext = r.pod<uint8_t>() != 0;
                    const std::string pb = r.str();
                    if (!r.ok) break;
                    if (!factory) return fail(name + ": the plan contains plugin layers but no IPluginFactory was given");
                    IPlugin* plugin = factory->createPlugin(name.c_str(), pb.data(), pb.size());
                    if (!plugin) return fail(name + ": the plugin factory could not re-create the plugin");
                    layer = ext ? net->addPluginExt(ins.data(), (int)ins.size(), *static_cast<IPluginExt*>(plugin))
                                : net->addPlugin(ins.data(), (int)ins.size(), *plugin);
                    break;
                }
                default: return fail(name + ": unknown layer type in plan");
            }
            if (!r.ok || !layer) return fail("truncated or inconsistent plan at layer " + name);
            layer->setName(name.c_str());
            if ((int)outs.size() != layer->getNbOutputs()) return fail(name + ": output count mismatch");
            for (size_t i = 0; i < outs.size(); i++) {
                layer->getOutput((int)i)->setName(outs[i].second.c_str());
                tensor[outs[i].first] = layer->getOutput((int)i);
            }
        }
        const uint32_t nout = r.pod<uint32_t>();
        for (uint32_t i = 0; i < nout && r.ok; i++) {
            auto it = tensor.find(r.pod<int32_t>());
            if (it == tensor.end()) return fail("output tensor missing");
            net->markOutput(*it->second);
        }
        if (!r.ok) return fail("truncated plan");
        auto* e = new EngineImpl(*net, max_batch, half2, log_);
        if (!e->ok()) {
            delete e;
            return fail("engine build failed");
        }
        e->weight_store_ = std::move(store);
        return e;
    }
    void destroy() override { delete this; }
private:
    ILogger& log_;
};

}  // namespace

IBuilder* createInferBuilder(ILogger& logger) { return new BuilderImpl(logger); }
IRuntime* createInferRuntime(ILogger& logger) { return new RuntimeImpl(logger); }

}  // namespace nvinfer1
