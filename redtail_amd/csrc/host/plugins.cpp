// Stereo DNN plugins on MI355X: the IPlugin / IPluginExt implementations behind
// redtail_tensorrt_plugins.h.  Behavioural contract = the reference's stereoDNN/lib/*_plugin.cpp
// (shape rules, lifecycle, serialisation blobs, error convention); the arithmetic lives in the HIP
// kernels and is reached exclusively through the C ABI of include/rt_stereo.h.
#include <cassert>
#include <cstring>
#include <iomanip>
#include <mutex>
#include <sstream>
#include <vector>

#include "plugin_internal.h"
#include "rt_stereo.h"

namespace redtail { namespace tensorrt {

using namespace nvinfer1;
using internal::IStereoPlugin;
using internal::Kind;

namespace internal {
void logError(ILogger& log, int status, const char* file, int line, const char* func) {
    std::ostringstream s;
    s << file << ":" << line << ": " << func << ": HIP/rt error " << status << " (" << rt_last_error_string() << ").";
    log.log(ILogger::Severity::kERROR, s.str().c_str());
}
}  // namespace internal

namespace {

size_t volume(const Dims& d) {
    size_t n = 1;
    for (int i = 0; i < d.nbDims; i++) n *= (size_t)d.d[i];
    return n;
}
bool sameDims(const Dims& a, const Dims& b) {
    if (a.nbDims != b.nbDims) return false;
    for (int i = 0; i < a.nbDims; i++)
        if (a.d[i] != b.d[i]) return false;
    return true;
}
std::string dimsStr(const Dims& d) {
    std::ostringstream s;
    s << "{";
    for (int i = 0; i < d.nbDims; i++) s << std::setw(4) << d.d[i] << (i + 1 < d.nbDims ? "," : "");
    s << "}";
    return s.str();
}
const char* typeStr(DataType t) { return t == DataType::kFLOAT ? "Float" : t == DataType::kHALF ? "Half" : "Other"; }
const char* fmtStr(PluginFormat f) { return f == PluginFormat::kNCHW ? "NCHW" : f == PluginFormat::kNC2HW2 ? "NC2HW2" : "NHWC8"; }
int rtType(DataType t) { return t == DataType::kHALF ? RT_F16 : RT_F32; }
Dims dims4(int a, int b, int c, int d) { return DimsNCHW(a, b, c, d); }

// little-endian POD (de)serialisation of the plan blobs
struct BlobWriter {
    std::string s;
    template <typename T> void put(T v) { s.append(reinterpret_cast<const char*>(&v), sizeof(T)); }
    void putDims(const Dims& d) { put<int32_t>(d.nbDims); for (int i = 0; i < d.nbDims; i++) put<int32_t>(d.d[i]); }
};
struct BlobReader {
    const char* p; size_t left;
    BlobReader(const void* data, size_t size) : p(static_cast<const char*>(data)), left(size) {}
    template <typename T> T get() {
        T v{};
        if (left >= sizeof(T)) { std::memcpy(&v, p, sizeof(T)); p += sizeof(T); left -= sizeof(T); } else { left = 0; }
        return v;
    }
    Dims getDims() {
        Dims d{};
        d.nbDims = get<int32_t>();
        if (d.nbDims < 0 || d.nbDims > Dims::MAX_DIMS) d.nbDims = 0;
        for (int i = 0; i < d.nbDims; i++) d.d[i] = get<int32_t>();
        return d;
    }
};

// ------------------------------------------------------------------------------------------------------
// ELU  (reference: lib/elu_plugin.cpp)
// ------------------------------------------------------------------------------------------------------
class EluPlugin : public IPluginExt, public IStereoPlugin {
public:
    EluPlugin(DataType type, ILogger& log, std::string name) : type_(type), log_(log), name_(std::move(name)) {
        assert(type_ == DataType::kFLOAT || type_ == DataType::kHALF);
    }
    EluPlugin(const char* name, const void* data, size_t size, ILogger& log) : log_(log), name_(name) {
        BlobReader r(data, size);                     // plugin-type word already consumed by the factory
        type_ = (DataType)r.get<int32_t>();
        format_ = (PluginFormat)r.get<uint8_t>();
        dims_ = r.getDims();
        assert(r.left == 0);
    }
    Kind kind() const override { return Kind::kElu; }
    const std::string& pluginName() const override { return name_; }

    bool supportsFormat(DataType type, PluginFormat format) const override {
        return type == type_ && (format == PluginFormat::kNCHW || format == PluginFormat::kNC2HW2);
    }
    int getNbOutputs() const override { return 1; }
    Dims getOutputDimensions(int, const Dims* inputs, int nbInputDims) override {
        assert(nbInputDims == 1);
        (void)nbInputDims;
        dims_ = inputs[0];
        return dims_;
    }
    void configureWithFormat(const Dims* in, int nbIn, const Dims* out, int nbOut, DataType type, PluginFormat format,
                             int) override {
        assert(nbIn == 1 && nbOut == 1 && sameDims(in[0], dims_) && sameDims(out[0], dims_) && type == type_);
        (void)in; (void)nbIn; (void)out; (void)nbOut;
        format_ = format;
        log_.log(ILogger::Severity::kINFO, (name_ + ": Dims: " + dimsStr(dims_) + ", Format: [" + typeStr(type) + ", " +
                                            fmtStr(format) + "]").c_str());
    }
    int initialize() override { return 0; }
    void terminate() override {}
    size_t getWorkspaceSize(int) const override { return 0; }
    int enqueue(int batchSize, const void* const* inputs, void** outputs, void*, cudaStream_t stream) override {
        // kNC2HW2 packs channel pairs: an odd C is padded by one channel (elu_plugin.cpp:165-168)
        size_t n = volume(dims_);
        if (format_ == PluginFormat::kNC2HW2 && dims_.nbDims >= 1 && (dims_.d[0] & 1)) n = n / dims_.d[0] * (dims_.d[0] + 1);
        int rc = rt_elu(inputs[0], outputs[0], (int64_t)batchSize * (int64_t)n, rtType(type_), stream);
        RT_CHECKL(rc, log_);
        return rc == 0 ? 0 : -1;
    }
    size_t getSerializationSize() override { return blob().size(); }
    void serialize(void* buffer) override { auto b = blob(); std::memcpy(buffer, b.data(), b.size()); }

private:
    std::string blob() const {
        BlobWriter w;
        w.put<int32_t>((int32_t)StereoDnnPluginFactory::PluginType::kElu);
        w.put<int32_t>((int32_t)type_);
        w.put<uint8_t>((uint8_t)format_);
        w.putDims(dims_);
        return w.s;
    }
    DataType type_ = DataType::kFLOAT;
    PluginFormat format_ = PluginFormat::kNCHW;
    Dims dims_{};
    ILogger& log_;
    std::string name_;
};

// ------------------------------------------------------------------------------------------------------
// Cost volume  (reference: lib/cost_volume_plugin.cpp)
// ------------------------------------------------------------------------------------------------------
class CostVolumePlugin : public IPluginExt, public IStereoPlugin {
public:
    CostVolumePlugin(DataType type, CostVolumeType cv, int max_disp, ILogger& log, std::string name)
        : type_(type), cv_(cv), max_disp_(max_disp), log_(log), name_(std::move(name)) {
        assert(type_ == DataType::kFLOAT || type_ == DataType::kHALF);
        assert(max_disp_ > 0);
    }
    CostVolumePlugin(const char* name, const void* data, size_t size, ILogger& log) : log_(log), name_(name) {
        BlobReader r(data, size);
        type_ = (DataType)r.get<int32_t>();
        format_ = (PluginFormat)r.get<uint8_t>();
        cv_ = (CostVolumeType)r.get<int32_t>();
        max_disp_ = r.get<int32_t>();
        in_ = r.getDims();
        out_ = r.getDims();
        assert(r.left == 0);
    }
    Kind kind() const override { return Kind::kCostVolume; }
    const std::string& pluginName() const override { return name_; }
    CostVolumeType costVolumeType() const override { return cv_; }
    int maxDisparity() const override { return max_disp_; }
    void setExactFp32(bool on) override { exact_ = on; }

    bool supportsFormat(DataType type, PluginFormat format) const override {
        bool ok = (type == DataType::kFLOAT && format == PluginFormat::kNCHW) ||
                  (type == DataType::kHALF && format == PluginFormat::kNC2HW2);
        return type == type_ && ok;
    }
    int getNbOutputs() const override { return 1; }
    Dims getOutputDimensions(int, const Dims* inputs, int nbInputDims) override {
        assert(nbInputDims == 2 && inputs[0].nbDims == 3 && sameDims(inputs[0], inputs[1]));
        (void)nbInputDims;
        in_ = inputs[0];
        if (cv_ == CostVolumeType::kDefault) out_ = dims4(max_disp_, 2 * in_.d[0], in_.d[1], in_.d[2]);
        else out_ = DimsCHW(max_disp_, in_.d[1], in_.d[2]);
        return out_;
    }
    void configureWithFormat(const Dims* in, int nbIn, const Dims* out, int nbOut, DataType type, PluginFormat format,
                             int) override {
        assert(nbIn == 2 && nbOut == 1 && sameDims(in[0], in_) && sameDims(in[1], in_) && sameDims(out[0], out_) && type == type_);
        (void)in; (void)nbIn; (void)out; (void)nbOut; (void)type;
        format_ = format;
        log_.log(ILogger::Severity::kINFO, (name_ + ": InDims(x2): " + dimsStr(in_)).c_str());
        log_.log(ILogger::Severity::kINFO, (name_ + ": OutDims   : " + dimsStr(out_)).c_str());
    }
    int initialize() override { return 0; }
    void terminate() override {}
    size_t getWorkspaceSize(int) const override { return 0; }
    int enqueue(int batchSize, const void* const* inputs, void** outputs, void*, cudaStream_t stream) override {
        // unlike the reference (batch asserted to 1, cost_volume_plugin.cpp:124) the batch is real here
        int rc;
        if (cv_ == CostVolumeType::kDefault)
            rc = rt_cost_volume(inputs[0], inputs[1], outputs[0], batchSize, in_.d[0], in_.d[1], in_.d[2], max_disp_,
                                rtType(type_), stream);
        else
            rc = rt_corr_cost_volume_flags(inputs[0], inputs[1], outputs[0], batchSize, in_.d[0], in_.d[1], in_.d[2], max_disp_,
                                           rtType(type_), format_ == PluginFormat::kNC2HW2 ? RT_NC2HW2 : RT_NCHW,
                                           exact_ ? RT_CONV_EXACT_FP32 : 0u, stream);
        RT_CHECKL(rc, log_);
        return rc;
    }
    size_t getSerializationSize() override { return blob().size(); }
    void serialize(void* buffer) override { auto b = blob(); std::memcpy(buffer, b.data(), b.size()); }

private:
    std::string blob() const {
        BlobWriter w;
        w.put<int32_t>((int32_t)StereoDnnPluginFactory::PluginType::kCostVolume);
        w.put<int32_t>((int32_t)type_);
        w.put<uint8_t>((uint8_t)format_);
        w.put<int32_t>((int32_t)cv_);
        w.put<int32_t>(max_disp_);
        w.putDims(in_);
        w.putDims(out_);
        return w.s;
    }
    DataType type_ = DataType::kFLOAT;
    PluginFormat format_ = PluginFormat::kNCHW;
    CostVolumeType cv_ = CostVolumeType::kDefault;
    int max_disp_ = 0;
    bool exact_ = false;           // (an engine setting, not part of the blob)
    Dims in_{}, out_{};
    ILogger& log_;
    std::string name_;
};

// ------------------------------------------------------------------------------------------------------
// Soft-argmax / soft-argmin  (reference: lib/softargmax_plugin.cpp) -- one kernel, no workspace
// ------------------------------------------------------------------------------------------------------
class SoftargmaxPlugin : public IPluginExt, public IStereoPlugin {
public:
    SoftargmaxPlugin(DataType type, SoftargmaxType sm, ILogger& log, std::string name)
        : type_(type), sm_(sm), log_(log), name_(std::move(name)) {}
    SoftargmaxPlugin(const char* name, const void* data, size_t size, ILogger& log) : log_(log), name_(name) {
        BlobReader r(data, size);
        type_ = (DataType)r.get<int32_t>();
        sm_ = (SoftargmaxType)r.get<int32_t>();
        in_ = r.getDims();
        out_ = r.getDims();
        assert(r.left == 0);
    }
    Kind kind() const override { return Kind::kSoftargmax; }
    const std::string& pluginName() const override { return name_; }
    SoftargmaxType softargmaxType() const override { return sm_; }

    bool supportsFormat(DataType type, PluginFormat format) const override { return type == type_ && format == PluginFormat::kNCHW; }
    int getNbOutputs() const override { return 1; }
    Dims getOutputDimensions(int, const Dims* inputs, int nbInputDims) override {
        assert(nbInputDims == 1 && (inputs[0].nbDims == 3 || inputs[0].nbDims == 4));
        (void)nbInputDims;
        if (inputs[0].nbDims == 3) in_ = inputs[0];
        else {                                            // (D,1,H,W): C must be 1 (softargmax_plugin.cpp:66-72)
            assert(inputs[0].d[1] == 1);
            in_ = DimsCHW(inputs[0].d[0], inputs[0].d[2], inputs[0].d[3]);
        }
        out_ = DimsCHW(1, in_.d[1], in_.d[2]);
        return out_;
    }
    void configureWithFormat(const Dims*, int nbIn, const Dims* out, int nbOut, DataType type, PluginFormat, int) override {
        assert(nbIn == 1 && nbOut == 1 && sameDims(out[0], out_) && type == type_);
        (void)nbIn; (void)out; (void)nbOut; (void)type;
        log_.log(ILogger::Severity::kINFO, (name_ + ": InDims : " + dimsStr(in_)).c_str());
        log_.log(ILogger::Severity::kINFO, (name_ + ": OutDims: " + dimsStr(out_)).c_str());
    }
    int initialize() override { return 0; }
    void terminate() override {}
    size_t getWorkspaceSize(int) const override { return 0; }
    int enqueue(int batchSize, const void* const* inputs, void** outputs, void*, cudaStream_t stream) override {
        int rc = rt_softargmax(inputs[0], outputs[0], batchSize, in_.d[0], in_.d[1], in_.d[2], sm_ == SoftargmaxType::kMin,
                               rtType(type_), stream);
        RT_CHECKL(rc, log_);
        return rc == 0 ? 0 : -1;
    }
    size_t getSerializationSize() override { return blob().size(); }
    void serialize(void* buffer) override { auto b = blob(); std::memcpy(buffer, b.data(), b.size()); }

private:
    std::string blob() const {
        BlobWriter w;
        w.put<int32_t>((int32_t)StereoDnnPluginFactory::PluginType::kSoftargmax);
        w.put<int32_t>((int32_t)type_);
        w.put<int32_t>((int32_t)sm_);
        w.putDims(in_);
        w.putDims(out_);
        return w.s;
    }
    DataType type_ = DataType::kFLOAT;
    SoftargmaxType sm_ = SoftargmaxType::kMax;
    Dims in_{}, out_{};
    ILogger& log_;
    std::string name_;
};

// ------------------------------------------------------------------------------------------------------
// Conv3D / Conv3DTranspose  (reference: lib/conv3d_plugin.cpp, lib/conv3d_transpose_plugin.cpp)
// The plan (re-packed weights + gather tables on the device) is created in configure() and released
// in terminate(), like the reference's cudaMalloc'ed weights (conv3d_plugin.cpp:122-133,144-177).
// ------------------------------------------------------------------------------------------------------
class Conv3DPluginBase : public IPlugin, public IStereoPlugin {
public:
    Conv3DPluginBase(bool transposed, Conv3DType conv_type, Dims w_dims, Dims stride, Dims pad_start, Dims pad_end,
                     Weights kernel, Weights bias, ILogger& log, std::string name)
        : transposed_(transposed), conv_type_(conv_type), w_dims_(w_dims), stride_(stride), pad_start_(pad_start),
          pad_end_(pad_end), kernel_(kernel), bias_(bias), log_(log), name_(std::move(name)) {
        assert(conv_type_ == Conv3DType::kTensorFlow && "only the TF layout is used by the Stereo DNN models");
        assert(w_dims_.nbDims == 5 && stride_.nbDims == 3 && pad_start_.nbDims == 3 && pad_end_.nbDims == 3);
        assert(kernel_.type == DataType::kFLOAT || kernel_.type == DataType::kHALF);
        assert(kernel_.count > 0 && kernel_.values != nullptr);
        assert((bias_.count > 0 && bias_.values != nullptr) || (bias_.count == 0 && bias_.values == nullptr));
        assert(bias_.count == 0 || bias_.type == kernel_.type);
    }
    const std::string& pluginName() const override { return name_; }
    int getNbOutputs() const override { return 1; }
    int initialize() override { return plan_ ? 0 : -1; }
    void terminate() override {
        if (plan_) rt_conv_plan_destroy(plan_);
        plan_ = nullptr;
    }
    // scratch of the plan's launches (the factored first Conv3D over a folded cost volume keeps four small maps per sample): the runtime
    // owns it and hands it to enqueue, as TensorRT does for the reference's plugins (lib/conv3d_plugin.cpp:179-190)
    size_t getWorkspaceSize(int maxBatchSize) const override { return plan_ ? rt_conv_plan_workspace_bytes(plan_, maxBatchSize) : 0; }
    int enqueue(int batchSize, const void* const* inputs, void** outputs, void* workspace, cudaStream_t stream) override {
        return enqueueFused(batchSize, inputs[0], outputs[0], nullptr, workspace, workspace ? getWorkspaceSize(batchSize) : 0, stream);
    }
    int enqueueFused(int batchSize, const void* x, void* y, const void* residual, void* workspace, size_t workspace_bytes, cudaStream_t stream) override {
        if (!plan_) return -1;
        int rc = rt_conv_enqueue_ws(plan_, x, y, residual, batchSize, workspace, workspace_bytes, stream, 0);
        RT_CHECKL(rc, log_);
        return rc == 0 ? 0 : -1;
    }
    bool setFusion(const internal::ConvFusion& f) override {
        if (f.out_depth && !transposed_) return false;
        if ((f.in_pad_end || f.cv_fold) && transposed_) return false;
        const internal::ConvFusion old = fusion_;
        fusion_ = f;
        if (rebuildPlan()) return true;
        fusion_ = old;                       // e.g. cv_fold on a window the split kernels do not cover: keep the plan we had
        rebuildPlan();
        return false;
    }
    internal::ConvFusion fusion() const override { return fusion_; }
    bool setIoTypes(bool x16, bool y16) override {
        if (!plan_) return false;
        if (rt_conv_plan_set_io_types(plan_, x16 ? RT_F16 : RT_F32, y16 ? RT_F16 : RT_F32) != 0) return false;
        x16_ = x16; y16_ = y16;
        xil_ = yil_ = ril_ = false;          // a change of the storage types resets the layouts
        softarg_ = 0;
        return true;
    }
    int ilCaps() const override { return plan_ ? rt_conv_plan_supports_il8(plan_) : 0; }
    bool setLayouts(bool x, bool y, bool r) override {
        if (!plan_ || rt_conv_plan_set_layouts(plan_, x, y, r) != 0) return false;
        xil_ = x; yil_ = y; ril_ = r;
        softarg_ = 0;
        return true;
    }
    bool setSoftarg(int mode) override {
        if (!plan_ || !transposed_ || rt_conv_plan_set_softarg(plan_, mode) != 0) return false;
        softarg_ = mode;
        return true;
    }
    int softarg() const override { return softarg_; }
    size_t getSerializationSize() override { return 0; }        // not serialisable, as in the reference
    void serialize(void*) override {}

protected:
    bool rebuildPlan() {
        terminate();
        rtConv3dDesc d{};
        fillDesc(d);
        d.act = fusion_.act;
        d.out_dchw = fusion_.out_dchw ? 1 : 0;
        d.has_residual = fusion_.residual ? 1 : 0;
        d.out_depth = fusion_.out_depth;
        d.in_pad_end = fusion_.in_pad_end;
        d.cv_fold = fusion_.cv_fold;
        d.dtype = kernel_.type == DataType::kHALF ? RT_F16 : RT_F32;   // storage type of the weight blob
        int rc = createPlan(d);
        RT_CHECKL(rc, log_);
        // The executor sized its tensors for these settings: a rebuilt plan that does not take them again is an error, not a fallback
        // (a plan without its fused soft-argmax would write the whole volume into the map's buffer).  The mirrors say what the plan has.
        bool kept = true;
        if (rc == 0 && (x16_ || y16_) && rt_conv_plan_set_io_types(plan_, x16_ ? RT_F16 : RT_F32, y16_ ? RT_F16 : RT_F32) != 0) { x16_ = y16_ = false; kept = false; }
        if (rc == 0 && (xil_ || yil_ || ril_) && rt_conv_plan_set_layouts(plan_, xil_, yil_, ril_) != 0) { xil_ = yil_ = ril_ = false; kept = false; }
        if (rc == 0 && softarg_ && rt_conv_plan_set_softarg(plan_, softarg_) != 0) { softarg_ = 0; kept = false; }
        if (rc == 0 && !kept) RT_CHECKL(RT_E_UNSUPPORTED, log_);
        return rc == 0 && kept;
    }
    virtual void fillDesc(rtConv3dDesc& d) const = 0;
    virtual int createPlan(const rtConv3dDesc& d) = 0;

    bool transposed_;
    Conv3DType conv_type_;
    Dims w_dims_, stride_, pad_start_, pad_end_;
    Weights kernel_, bias_;
    internal::ConvFusion fusion_;
    bool x16_ = false, y16_ = false;
    bool xil_ = false, yil_ = false, ril_ = false;
    int softarg_ = 0;
    rtConvPlan* plan_ = nullptr;
    Dims x_dims_{}, y_dims_{};
    ILogger& log_;
    std::string name_;
};

class Conv3DPlugin : public Conv3DPluginBase {
public:
    using Conv3DPluginBase::Conv3DPluginBase;
    Kind kind() const override { return Kind::kConv3D; }

    Dims getOutputDimensions(int index, const Dims* inputs, int nbInputDims) override {
        assert(index == 0 && nbInputDims == 1 && inputs[0].nbDims == 4);
        (void)index; (void)nbInputDims;
        x_dims_ = dims4(inputs[0].d[0], inputs[0].d[1], inputs[0].d[2], inputs[0].d[3]);        // (D,C,H,W)
        assert(x_dims_.d[1] == w_dims_.d[2] && "input channels must match the filter's C");
        // cuDNN output size with the *start* pads only (conv3d_plugin.cpp:74-100): out = (in + 2p - k)/s + 1
        int o[3];
        const int in[3] = {x_dims_.d[0], x_dims_.d[2], x_dims_.d[3]};
        const int k[3] = {w_dims_.d[1], w_dims_.d[3], w_dims_.d[4]};
        for (int i = 0; i < 3; i++) o[i] = (in[i] + 2 * pad_start_.d[i] - k[i]) / stride_.d[i] + 1;
        y_dims_ = dims4(w_dims_.d[0], o[0], o[1], o[2]);                                         // (K,Do,Ho,Wo)
        return y_dims_;
    }
    void configure(const Dims* in, int nbIn, const Dims* out, int nbOut, int) override {
        assert(nbIn == 1 && nbOut == 1 && sameDims(in[0], x_dims_) && sameDims(out[0], y_dims_));
        (void)in; (void)nbIn; (void)out; (void)nbOut;
        rebuildPlan();
        log_.log(ILogger::Severity::kINFO, (name_ + ": InDims  : " + dimsStr(x_dims_)).c_str());
        log_.log(ILogger::Severity::kINFO, (name_ + ": OutDims : " + dimsStr(y_dims_)).c_str());
    }

protected:
    void fillDesc(rtConv3dDesc& d) const override {
        d.K = w_dims_.d[0]; d.C = w_dims_.d[2];
        d.D = x_dims_.d[0]; d.H = x_dims_.d[2]; d.W = x_dims_.d[3];
        d.kernel[0] = w_dims_.d[1]; d.kernel[1] = w_dims_.d[3]; d.kernel[2] = w_dims_.d[4];
        for (int i = 0; i < 3; i++) { d.stride[i] = stride_.d[i]; d.pad_start[i] = pad_start_.d[i]; d.pad_end[i] = pad_end_.d[i]; }
    }
    int createPlan(const rtConv3dDesc& d) override {
        return rt_conv3d_plan_create(&plan_, &d, kernel_.values, bias_.count > 0 ? bias_.values : nullptr);
    }
};

class Conv3DTransposePlugin : public Conv3DPluginBase {
public:
    Conv3DTransposePlugin(Conv3DType conv_type, Dims w_dims, Dims out_dims, Dims stride, Dims pad_start, Dims pad_end,
                          Weights kernel, Weights bias, ILogger& log, std::string name)
        : Conv3DPluginBase(true, conv_type, w_dims, stride, pad_start, pad_end, kernel, bias, log, std::move(name)) {
        assert(out_dims.nbDims == 4);
        x_dims_ = dims4(out_dims.d[0], out_dims.d[1], out_dims.d[2], out_dims.d[3]);            // output (D,C,H,W)
    }
    Kind kind() const override { return Kind::kConv3DTranspose; }

    Dims getOutputDimensions(int index, const Dims* inputs, int nbInputDims) override {
        assert(index == 0 && nbInputDims == 1 && inputs[0].nbDims == 4);
        (void)index; (void)nbInputDims;
        y_dims_ = dims4(inputs[0].d[0], inputs[0].d[1], inputs[0].d[2], inputs[0].d[3]);        // input (K,Dy,Hy,Wy)
        assert(y_dims_.d[0] == w_dims_.d[0] && x_dims_.d[1] == w_dims_.d[2]);
        return x_dims_;
    }
    void configure(const Dims* in, int nbIn, const Dims* out, int nbOut, int) override {
        assert(nbIn == 1 && nbOut == 1 && sameDims(in[0], y_dims_) && sameDims(out[0], x_dims_));
        (void)in; (void)nbIn; (void)out; (void)nbOut;
        rebuildPlan();
        log_.log(ILogger::Severity::kINFO, (name_ + ": InDims  : " + dimsStr(y_dims_)).c_str());
        log_.log(ILogger::Severity::kINFO, (name_ + ": OutDims : " + dimsStr(x_dims_)).c_str());
    }

protected:
    void fillDesc(rtConv3dDesc& d) const override {
        d.K = w_dims_.d[0]; d.C = w_dims_.d[2];
        d.D = x_dims_.d[0]; d.H = x_dims_.d[2]; d.W = x_dims_.d[3];
        d.kernel[0] = w_dims_.d[1]; d.kernel[1] = w_dims_.d[3]; d.kernel[2] = w_dims_.d[4];
        for (int i = 0; i < 3; i++) { d.stride[i] = stride_.d[i]; d.pad_start[i] = pad_start_.d[i]; d.pad_end[i] = pad_end_.d[i]; }
    }
    int createPlan(const rtConv3dDesc& d) override {
        const int in_dims[3] = {y_dims_.d[1], y_dims_.d[2], y_dims_.d[3]};
        return rt_conv3d_transpose_plan_create(&plan_, &d, in_dims, kernel_.values, bias_.count > 0 ? bias_.values : nullptr);
    }
};

// ------------------------------------------------------------------------------------------------------
// Transform / Padding / Slice  (reference: lib/transform_plugin.cpp, padding_plugin.cpp, slice_plugin.cpp)
// ------------------------------------------------------------------------------------------------------
class SimplePluginBase : public IPlugin, public IStereoPlugin {
public:
    SimplePluginBase(ILogger& log, std::string name) : log_(log), name_(std::move(name)) {}
    const std::string& pluginName() const override { return name_; }
    int getNbOutputs() const override { return 1; }
    void configure(const Dims* in, int nbIn, const Dims* out, int nbOut, int) override {
        assert(nbIn == 1 && nbOut == 1 && sameDims(in[0], in_) && sameDims(out[0], out_));
        (void)in; (void)nbIn; (void)out; (void)nbOut;
        log_.log(ILogger::Severity::kINFO, (name_ + ": InDims : " + dimsStr(in_)).c_str());
        log_.log(ILogger::Severity::kINFO, (name_ + ": OutDims: " + dimsStr(out_)).c_str());
    }
    int initialize() override { return 0; }
    void terminate() override {}
    size_t getWorkspaceSize(int) const override { return 0; }
    size_t getSerializationSize() override { return 0; }
    void serialize(void*) override {}

protected:
    Dims in_{}, out_{};
    ILogger& log_;
    std::string name_;
};

class TransformPlugin : public SimplePluginBase {
public:
    TransformPlugin(Permutation perm, ILogger& log, std::string name) : SimplePluginBase(log, std::move(name)), perm_(perm) {}
    Kind kind() const override { return Kind::kTransform; }
    Permutation permutation() const override { return perm_; }
    Dims getOutputDimensions(int, const Dims* inputs, int nbInputDims) override {
        assert(nbInputDims == 1 && inputs[0].nbDims == 4);
        (void)nbInputDims;
        in_ = dims4(inputs[0].d[0], inputs[0].d[1], inputs[0].d[2], inputs[0].d[3]);
        int seen = 0;
        for (int i = 0; i < 4; i++) seen |= 1 << perm_.order[i];
        assert(seen == 15 && "order must be a permutation of {0,1,2,3}");
        (void)seen;
        out_ = dims4(in_.d[perm_.order[0]], in_.d[perm_.order[1]], in_.d[perm_.order[2]], in_.d[perm_.order[3]]);
        return out_;
    }
    int enqueue(int batchSize, const void* const* inputs, void** outputs, void*, cudaStream_t stream) override {
        int rc = rt_permute4d(inputs[0], outputs[0], batchSize, in_.d[0], in_.d[1], in_.d[2], in_.d[3], perm_.order, RT_F32, stream);
        RT_CHECKL(rc, log_);
        return rc == 0 ? 0 : -1;
    }
private:
    Permutation perm_;
};

class PaddingPlugin : public SimplePluginBase {
public:
    PaddingPlugin(DimsNCHW pad_start, DimsNCHW pad_end, ILogger& log, std::string name)
        : SimplePluginBase(log, std::move(name)), pad_start_(pad_start), pad_end_(pad_end) {
        // only end-padding of the outermost (D) dimension exists in the models (padding_plugin.cpp:20-30)
        assert(pad_start_.n() == 0 && pad_start_.c() == 0 && pad_start_.h() == 0 && pad_start_.w() == 0);
        assert(pad_end_.n() >= 0 && pad_end_.c() == 0 && pad_end_.h() == 0 && pad_end_.w() == 0);
    }
    Kind kind() const override { return Kind::kPadding; }
    int padEnd() const override { return pad_end_.n(); }
    Dims getOutputDimensions(int, const Dims* inputs, int nbInputDims) override {
        assert(nbInputDims == 1 && inputs[0].nbDims == 4);
        (void)nbInputDims;
        in_ = dims4(inputs[0].d[0], inputs[0].d[1], inputs[0].d[2], inputs[0].d[3]);
        out_ = dims4(in_.d[0] + pad_end_.n(), in_.d[1], in_.d[2], in_.d[3]);
        return out_;
    }
    int enqueue(int batchSize, const void* const* inputs, void** outputs, void*, cudaStream_t stream) override {
        const int64_t inner = (int64_t)in_.d[1] * in_.d[2] * in_.d[3];
        int rc = rt_pad_d(inputs[0], outputs[0], batchSize, in_.d[0], inner, pad_end_.n(), RT_F32, stream);
        RT_CHECKL(rc, log_);
        return rc;
    }
private:
    DimsNCHW pad_start_, pad_end_;
};

class SlicePlugin : public SimplePluginBase {
public:
    SlicePlugin(Dims dims, Dims start, Dims end, ILogger& log, std::string name)
        : SimplePluginBase(log, std::move(name)), start_(start), end_(end) {
        in_ = dims;
        assert(in_.nbDims == 4 && start_.nbDims == 4 && end_.nbDims == 4);
        assert(0 <= start_.d[0] && start_.d[0] < end_.d[0] && end_.d[0] <= in_.d[0]);
        for (int i = 1; i < 4; i++) assert(start_.d[i] == 0 && end_.d[i] == in_.d[i]);
    }
    Kind kind() const override { return Kind::kSlice; }
    int sliceStart() const override { return start_.d[0]; }
    int sliceEnd() const override { return end_.d[0]; }
    Dims getOutputDimensions(int, const Dims* inputs, int nbInputDims) override {
        assert(nbInputDims == 1 && inputs[0].nbDims == 4);
        (void)nbInputDims;
        out_ = dims4(end_.d[0] - start_.d[0], inputs[0].d[1], inputs[0].d[2], inputs[0].d[3]);
        return out_;
    }
    int enqueue(int batchSize, const void* const* inputs, void** outputs, void*, cudaStream_t stream) override {
        const int64_t inner = (int64_t)in_.d[1] * in_.d[2] * in_.d[3];
        int rc = rt_slice_d(inputs[0], outputs[0], batchSize, in_.d[0], inner, start_.d[0], end_.d[0], RT_F32, stream);
        RT_CHECKL(rc, log_);
        return rc;
    }
private:
    Dims start_, end_;
};

// ------------------------------------------------------------------------------------------------------
// Container  (reference: lib/internal_utils.h:114-168)
// ------------------------------------------------------------------------------------------------------
class PluginContainer : public IPluginContainer {
public:
    explicit PluginContainer(ILogger& log) : log_(log) {}
    ~PluginContainer() noexcept override = default;     // plugins are intentionally not deleted (TRT contract)

    IPlugin* createEluPlugin(DataType t, std::string name) override { return keep(new EluPlugin(t, log_, name)); }
    IPlugin* deserializeEluPlugin(const char* name, const void* d, size_t n) override { return keep(new EluPlugin(name, d, n, log_)); }
    IPlugin* createCostVolumePlugin(DataType t, CostVolumeType cv, int md, std::string name) override {
        return keep(new CostVolumePlugin(t, cv, md, log_, name));
    }
    IPlugin* deserializeCostVolumePlugin(const char* name, const void* d, size_t n) override {
        return keep(new CostVolumePlugin(name, d, n, log_));
    }
    IPlugin* createConv3DPlugin(Conv3DType ct, Dims k, Dims s, Dims ps, Dims pe, Weights w, Weights b, std::string name) override {
        return keep(new Conv3DPlugin(false, ct, k, s, ps, pe, w, b, log_, name));
    }
    IPlugin* createConv3DTransposePlugin(Conv3DType ct, Dims k, Dims od, Dims s, Dims ps, Dims pe, Weights w, Weights b,
                                         std::string name) override {
        return keep(new Conv3DTransposePlugin(ct, k, od, s, ps, pe, w, b, log_, name));
    }
    IPlugin* createTransformPlugin(Permutation p, std::string name) override { return keep(new TransformPlugin(p, log_, name)); }
    IPlugin* createPaddingPlugin(DimsNCHW ps, DimsNCHW pe, std::string name) override { return keep(new PaddingPlugin(ps, pe, log_, name)); }
    IPlugin* createSlicePlugin(Dims d, Dims s, Dims e, std::string name) override { return keep(new SlicePlugin(d, s, e, log_, name)); }
    IPlugin* createSoftargmaxPlugin(DataType t, SoftargmaxType sm, std::string name) override {
        return keep(new SoftargmaxPlugin(t, sm, log_, name));
    }
    IPlugin* deserializeSoftargmaxPlugin(const char* name, const void* d, size_t n) override {
        return keep(new SoftargmaxPlugin(name, d, n, log_));
    }

private:
    template <typename P> IPlugin* keep(P* p) {
        std::lock_guard<std::mutex> lock(lock_);
        plugins_.push_back(p);
        return p;
    }
    std::vector<IPlugin*> plugins_;
    std::mutex lock_;
    ILogger& log_;
};

IPluginLayer* appendPlugin(INetworkDefinition& network, ITensor* const* inputs, int n, IPlugin* plugin) {
    auto ext = dynamic_cast<IPluginExt*>(plugin);
    return ext ? network.addPluginExt(inputs, n, *ext) : network.addPlugin(inputs, n, *plugin);
}

}  // namespace

std::unique_ptr<IPluginContainer> IPluginContainer::create(ILogger& log) {
    return std::unique_ptr<IPluginContainer>(new PluginContainer(log));
}

ILayer* addElu(IPluginContainer& f, INetworkDefinition& net, ITensor& input, DataType data_type, const std::string& name) {
    ITensor* in[] = {&input};
    return appendPlugin(net, in, 1, f.createEluPlugin(data_type, name));
}
ILayer* addCostVolume(IPluginContainer& f, INetworkDefinition& net, ITensor& left, ITensor& right, CostVolumeType cv_type,
                      int max_disparity, DataType data_type, const std::string& name) {
    ITensor* in[] = {&left, &right};
    return appendPlugin(net, in, 2, f.createCostVolumePlugin(data_type, cv_type, max_disparity, name));
}
ILayer* addConv3D(IPluginContainer& f, INetworkDefinition& net, ITensor& input, Conv3DType conv_type, Dims kernel_dims,
                  Dims stride_dims, Dims pad_start_dims, Dims pad_end_dims, Weights kernel_weights, Weights bias_weights,
                  const std::string& name) {
    ITensor* in[] = {&input};
    return appendPlugin(net, in, 1, f.createConv3DPlugin(conv_type, kernel_dims, stride_dims, pad_start_dims, pad_end_dims,
                                                         kernel_weights, bias_weights, name));
}
ILayer* addConv3DTranspose(IPluginContainer& f, INetworkDefinition& net, ITensor& input, Conv3DType conv_type,
                           Dims kernel_dims, Dims out_dims, Dims stride_dims, Dims pad_start_dims, Dims pad_end_dims,
                           Weights kernel_weights, Weights bias_weights, const std::string& name) {
    ITensor* in[] = {&input};
    return appendPlugin(net, in, 1, f.createConv3DTransposePlugin(conv_type, kernel_dims, out_dims, stride_dims, pad_start_dims,
                                                                  pad_end_dims, kernel_weights, bias_weights, name));
}
ILayer* addSlice(IPluginContainer& f, INetworkDefinition& net, ITensor& input, Dims dims, Dims slice_start, Dims slice_end,
                 const std::string& name) {
    ITensor* in[] = {&input};
    return appendPlugin(net, in, 1, f.createSlicePlugin(dims, slice_start, slice_end, name));
}
ILayer* addTransform(IPluginContainer& f, INetworkDefinition& net, ITensor& input, Permutation permutation,
                     const std::string& name) {
    ITensor* in[] = {&input};
    return appendPlugin(net, in, 1, f.createTransformPlugin(permutation, name));
}
ILayer* addPad(IPluginContainer& f, INetworkDefinition& net, ITensor& input, DimsNCHW pad_start, DimsNCHW pad_end,
               const std::string& name) {
    ITensor* in[] = {&input};
    return appendPlugin(net, in, 1, f.createPaddingPlugin(pad_start, pad_end, name));
}
ILayer* addSoftargmax(IPluginContainer& f, INetworkDefinition& net, ITensor& input, SoftargmaxType sm_type, DataType data_type,
                      const std::string& name) {
    ITensor* in[] = {&input};
    return appendPlugin(net, in, 1, f.createSoftargmaxPlugin(data_type, sm_type, name));
}

StereoDnnPluginFactory::StereoDnnPluginFactory(IPluginContainer& container) : container_(container) {}

IPlugin* StereoDnnPluginFactory::createPlugin(const char* layerName, const void* serialData, size_t serialLength) {
    if (serialLength < sizeof(int32_t)) return nullptr;
    int32_t type;
    std::memcpy(&type, serialData, sizeof(type));
    const char* rest = static_cast<const char*>(serialData) + sizeof(type);
    const size_t n = serialLength - sizeof(type);
    switch ((PluginType)type) {
        case PluginType::kElu: return container_.deserializeEluPlugin(layerName, rest, n);
        case PluginType::kCostVolume: return container_.deserializeCostVolumePlugin(layerName, rest, n);
        case PluginType::kSoftargmax: return container_.deserializeSoftargmaxPlugin(layerName, rest, n);
    }
    return nullptr;
}

} }  // namespace redtail::tensorrt
