// Whole-network C ABI (include/rt_stereo_net.h): weight-file parsing, network/engine/context life cycle.
#include "rt_stereo_net.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "networks.h"
#include "redtail_tensorrt_plugins.h"

using namespace nvinfer1;
using namespace redtail::tensorrt;

namespace {

thread_local std::string g_net_err;

class NetLogger : public ILogger {
public:
    NetLogger() {
        const char* e = getenv("RT_LOG_LEVEL");
        level_ = e ? atoi(e) : 1;
    }
    void log(Severity severity, const char* msg) override {
        if ((int)severity <= (int)Severity::kERROR) last_error = msg;
        if ((int)severity > level_) return;
        static const char* tag[] = {"INTERNAL_ERROR", "ERROR", "WARNING", "INFO"};
        fprintf(stderr, "[redtail_amd %s] %s\n", tag[(int)severity & 3], msg);
    }
    std::string last_error;
private:
    int level_;
};

class TextProfiler : public IProfiler {
public:
    void reportLayerTime(const char* name, float ms) override { rows.emplace_back(name, ms); }
    std::vector<std::pair<std::string, float>> rows;
};

int fail(const std::string& msg) {
    g_net_err = msg;
    return RT_E_BADARG;
}

}  // namespace

struct rtStereoNet {
    NetLogger log;
    std::vector<char> blob;                       // weight file image; Weights::values point into it
    weight_map weights;
    std::unique_ptr<IPluginContainer> plugins;
    ICudaEngine* engine = nullptr;
    IExecutionContext* context = nullptr;
    int layers = 0, width = 0, height = 0, max_batch = 1;
    ~rtStereoNet() {
        if (context) context->destroy();
        if (engine) engine->destroy();
    }
};

namespace {

// name '\0' uint32 count, count elements (reference reader: sample_app/main.cpp:111-134)
bool parseWeights(rtStereoNet& n, int dtype) {
    const size_t el = dtype == RT_F16 ? 2 : 4;
    size_t off = 0;
    const std::vector<char>& b = n.blob;
    while (off < b.size()) {
        const void* z = memchr(b.data() + off, 0, b.size() - off);
        if (!z) return false;
        std::string name(b.data() + off);
        off += name.size() + 1;
        if (off + 4 > b.size()) return false;
        uint32_t count;
        memcpy(&count, b.data() + off, 4);
        off += 4;
        if (off + (size_t)count * el > b.size()) return false;
        if (n.weights.count(name)) return false;
        n.weights[name] = Weights{dtype == RT_F16 ? DataType::kHALF : DataType::kFLOAT, b.data() + off, (int64_t)count};
        off += (size_t)count * el;
    }
    return !n.weights.empty();
}

int build(rtStereoNet** out, int model, int width, int height, int max_batch, int dtype, int max_disp,
          std::vector<char>&& blob, unsigned flags = 0) {
    if (!out) return fail("rt_net_create: null out pointer");
    if (width < 17 || height < 17 || max_batch < 1) return fail("rt_net_create: bad dimensions");
    if (dtype != RT_F32 && dtype != RT_F16) return fail("rt_net_create: weights_dtype must be RT_F32 or RT_F16");
    std::unique_ptr<rtStereoNet> n(new rtStereoNet());
    n->blob = std::move(blob);
    n->width = width; n->height = height; n->max_batch = max_batch;
    if (!parseWeights(*n, dtype)) return fail("rt_net_create: malformed weight file (expected name\\0, uint32 count, data ...)");
    n->plugins = IPluginContainer::create(n->log);
    IBuilder* builder = createInferBuilder(n->log);
    const DimsCHW dims{3, height, width};
    // The sample application keeps the plugins' DataType at kFLOAT for the 3-D models and passes its command-line data
    // type to the ResNet-18 2D builder (sample_app/main.cpp:228-256).  Here the plugin type stays kFLOAT for every model: what
    // an fp16 weight file switches on is half2 mode of the builder (fp16 storage between fused launches), and the executor
    // also accepts plugins created for kHALF as long as they are fused away (tests/test_sample_app.py drives that path)
    const DataType act_type = DataType::kFLOAT;
    INetworkDefinition* net = nullptr;
    switch (model) {
        case RT_MODEL_RESNET18_2D:
            net = createResNet18_2DNetwork(*builder, *n->plugins, dims, n->weights, act_type, max_disp > 0 ? max_disp : 48, n->log);
            break;
        case RT_MODEL_NVSMALL:
            net = createStereo3DNetwork(*builder, *n->plugins, Stereo3DModel::kNVSmall, dims, n->weights, act_type, max_disp > 0 ? max_disp : 48, n->log);
            break;
        case RT_MODEL_NVTINY:
            net = createStereo3DNetwork(*builder, *n->plugins, Stereo3DModel::kNVTiny, dims, n->weights, act_type, max_disp > 0 ? max_disp : 24, n->log);
            break;
        case RT_MODEL_RESNET18:
            net = createStereo3DNetwork(*builder, *n->plugins, Stereo3DModel::kResNet18, dims, n->weights, act_type, max_disp > 0 ? max_disp : 68, n->log);
            break;
        default:
            builder->destroy();
            return fail("rt_net_create: unknown model");
    }
    if (!net) {
        builder->destroy();
        return fail("rt_net_create: network construction failed: " + n->log.last_error);
    }
    n->layers = net->getNbLayers();
    builder->setMaxBatchSize(max_batch);
    builder->setHalf2Mode(dtype == RT_F16);        // fp16 weight file = fp16 inference, as in sample_app/main.cpp:256-262
    builder->setExactFp32Mode((flags & RT_CONV_EXACT_FP32) != 0);
    builder->setMaxWorkspaceSize((size_t)1 << 30);
    n->engine = builder->buildCudaEngine(*net);
    net->destroy();
    builder->destroy();
    if (!n->engine) return fail("rt_net_create: engine build failed: " + n->log.last_error);
    if (n->engine->getNbBindings() != 3 || n->engine->getBindingIndex("left") != 0 || n->engine->getBindingIndex("right") != 1 ||
        n->engine->getBindingIndex("disp") != 2)
        return fail("rt_net_create: unexpected bindings");
    n->context = n->engine->createExecutionContext();
    if (!n->context) return fail("rt_net_create: context creation failed");
    *out = n.release();
    return 0;
}

}  // namespace

extern "C" const char* rt_net_last_error(void) { return g_net_err.c_str(); }

extern "C" int rt_net_create(rtStereoNet** net, int model, int width, int height, int max_batch, int dtype, int max_disp,
                             const char* path) {
    if (!path) return fail("rt_net_create: null weights path");
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f.is_open()) return fail(std::string("rt_net_create: cannot open ") + path);
    std::vector<char> blob((size_t)f.tellg());
    f.seekg(0);
    f.read(blob.data(), (std::streamsize)blob.size());
    return build(net, model, width, height, max_batch, dtype, max_disp, std::move(blob));
}

extern "C" int rt_net_create_from_memory(rtStereoNet** net, int model, int width, int height, int max_batch, int dtype,
                                         int max_disp, const void* blob, size_t bytes) {
    if (!blob || !bytes) return fail("rt_net_create_from_memory: empty blob");
    std::vector<char> copy(static_cast<const char*>(blob), static_cast<const char*>(blob) + bytes);
    return build(net, model, width, height, max_batch, dtype, max_disp, std::move(copy));
}

// Multi-GPU start-up (include/rt_stereo_net.h): the weight image travels root -> everyone over RCCL, then every rank builds its engine.
extern "C" int rt_net_create_broadcast(rtStereoNet** net, int model, int width, int height, int max_batch, int dtype, int max_disp,
                                       const void* blob, size_t bytes, rtComm* comm, int root) {
    int world = 1, rank = 0;
    if (comm && rt_comm_info(comm, &world, &rank) != 0) return fail(std::string("rt_net_create_broadcast: ") + rt_last_error_string());
    if (root < 0 || root >= world) return fail("rt_net_create_broadcast: root is not a rank of the communicator");
    if (rank == root && (!blob || !bytes)) return fail("rt_net_create_broadcast: the root rank has no weight image");
    uint64_t n = rank == root ? (uint64_t)bytes : 0;
    if (comm && rt_comm_broadcast(comm, &n, sizeof(n), root, nullptr) != 0) return fail(std::string("rt_net_create_broadcast: ") + rt_last_error_string());
    if (n == 0 || n > ((uint64_t)1 << 32)) return fail("rt_net_create_broadcast: implausible weight image size received");
    std::vector<char> image((size_t)n);
    if (rank == root) memcpy(image.data(), blob, (size_t)n);
    if (comm && rt_comm_broadcast(comm, image.data(), (size_t)n, root, nullptr) != 0) return fail(std::string("rt_net_create_broadcast: ") + rt_last_error_string());
    return build(net, model, width, height, max_batch, dtype, max_disp, std::move(image));
}

extern "C" int rt_net_create_opt(rtStereoNet** net, const rtNetOptions* o) {
    if (!o) return fail("rt_net_create_opt: null options");
    std::vector<char> image;
    if (o->weights_path && !o->blob) {
        std::ifstream f(o->weights_path, std::ios::binary | std::ios::ate);
        if (!f.is_open()) return fail(std::string("rt_net_create_opt: cannot open ") + o->weights_path);
        image.resize((size_t)f.tellg());
        f.seekg(0);
        f.read(image.data(), (std::streamsize)image.size());
    } else if (o->blob && o->bytes) {
        image.assign(static_cast<const char*>(o->blob), static_cast<const char*>(o->blob) + o->bytes);
    }
    if (o->comm) {
        int world = 1, rank = 0;
        if (rt_comm_info(o->comm, &world, &rank) != 0) return fail(std::string("rt_net_create_opt: ") + rt_last_error_string());
        if (o->root < 0 || o->root >= world) return fail("rt_net_create_opt: root is not a rank of the communicator");
        if (rank == o->root && image.empty()) return fail("rt_net_create_opt: the root rank has no weight image");
        uint64_t n = rank == o->root ? (uint64_t)image.size() : 0;
        if (rt_comm_broadcast(o->comm, &n, sizeof(n), o->root, nullptr) != 0) return fail(std::string("rt_net_create_opt: ") + rt_last_error_string());
        if (n == 0 || n > ((uint64_t)1 << 32)) return fail("rt_net_create_opt: implausible weight image size received");
        image.resize((size_t)n);
        if (rt_comm_broadcast(o->comm, image.data(), (size_t)n, o->root, nullptr) != 0) return fail(std::string("rt_net_create_opt: ") + rt_last_error_string());
    }
    if (image.empty()) return fail("rt_net_create_opt: no weights (weights_path, blob or a broadcast)");
    return build(net, o->model, o->width, o->height, o->max_batch, o->weights_dtype, o->max_disp, std::move(image), o->flags);
}

extern "C" int rt_net_set_debug(rtStereoNet* net, int on) {
    if (!net || !net->context) return fail("rt_net_set_debug: null pointer");
    net->context->setDebugSync(on != 0);
    return 0;
}

extern "C" int rt_net_weights_crc32(const rtStereoNet* net, uint32_t* crc) {
    if (!net || !crc) return fail("rt_net_weights_crc32: null pointer");
    uint32_t c = 0xffffffffu;
    for (unsigned char b : net->blob) {
        c ^= b;
        for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xedb88320u & (0u - (c & 1u)));
    }
    *crc = c ^ 0xffffffffu;
    return 0;
}

extern "C" int rt_net_weights_image(const rtStereoNet* net, const void** data, size_t* bytes) {
    if (!net || !data || !bytes) return fail("rt_net_weights_image: null pointer");
    *data = net->blob.data();
    *bytes = net->blob.size();
    return 0;
}

// Engine plan = what ICudaEngine::serialize() returns (sample_app/main.cpp:269-275 caches it in a .plan file).
// Call with buf == NULL to query the size.  Networks with non-serialisable plugins (the 3-D models) have no plan,
// exactly as in the reference.
extern "C" int rt_net_serialize(rtStereoNet* net, void* buf, size_t buf_bytes, size_t* plan_bytes) {
    if (!net || !plan_bytes) return fail("rt_net_serialize: null pointer");
    IHostMemory* m = net->engine->serialize();
    if (!m) return fail("rt_net_serialize: " + net->log.last_error);
    *plan_bytes = m->size();
    int rc = 0;
    if (buf) {
        if (buf_bytes < m->size()) rc = fail("rt_net_serialize: buffer too small");
        else memcpy(buf, m->data(), m->size());
    }
    m->destroy();
    return rc;
}

// Counterpart of sample_app/main.cpp:198-220: IRuntime::deserializeCudaEngine(plan, size, &StereoDnnPluginFactory).
extern "C" int rt_net_create_from_plan(rtStereoNet** out, const void* plan, size_t plan_bytes) {
    if (!out || !plan || !plan_bytes) return fail("rt_net_create_from_plan: null pointer");
    std::unique_ptr<rtStereoNet> n(new rtStereoNet());
    n->plugins = IPluginContainer::create(n->log);
    StereoDnnPluginFactory factory(*n->plugins);
    IRuntime* runtime = createInferRuntime(n->log);
    n->engine = runtime->deserializeCudaEngine(plan, plan_bytes, &factory);
    runtime->destroy();
    if (!n->engine) return fail("rt_net_create_from_plan: " + n->log.last_error);
    if (n->engine->getNbBindings() != 3 || n->engine->getBindingIndex("left") != 0 || n->engine->getBindingIndex("right") != 1 ||
        n->engine->getBindingIndex("disp") != 2)
        return fail("rt_net_create_from_plan: unexpected bindings");
    n->max_batch = n->engine->getMaxBatchSize();
    n->context = n->engine->createExecutionContext();
    if (!n->context) return fail("rt_net_create_from_plan: context creation failed");
    *out = n.release();
    return 0;
}

extern "C" int rt_net_execute(rtStereoNet* net, const void* left, const void* right, void* disp, int batch, rtStream stream) {
    if (!net || !left || !right || !disp) return fail("rt_net_execute: null pointer");
    void* bindings[3] = {const_cast<void*>(left), const_cast<void*>(right), disp};
    const bool ok = stream ? net->context->enqueue(batch, bindings, (cudaStream_t)stream, nullptr) : net->context->execute(batch, bindings);
    if (!ok) return fail("rt_net_execute: " + net->log.last_error);
    return 0;
}

extern "C" int rt_net_profile(rtStereoNet* net, const void* left, const void* right, void* disp, int batch, char* buf,
                              size_t buf_bytes) {
    if (!net || !buf || !buf_bytes) return fail("rt_net_profile: null pointer");
    TextProfiler prof;
    net->context->setProfiler(&prof);
    void* bindings[3] = {const_cast<void*>(left), const_cast<void*>(right), disp};
    const bool ok = net->context->execute(batch, bindings);
    net->context->setProfiler(nullptr);
    if (!ok) return fail("rt_net_profile: " + net->log.last_error);
    std::ostringstream s;
    for (auto& r : prof.rows) s << r.first << "\t" << r.second << "\n";
    const std::string t = s.str();
    snprintf(buf, buf_bytes, "%s", t.c_str());
    return 0;
}

extern "C" int rt_net_set_streams(rtStereoNet* net, int streams) {
    if (!net || !net->context) return fail("rt_net_set_streams: null pointer");
    if (streams < 1 || streams > 2) return fail("rt_net_set_streams: 1 or 2 streams");
    net->context->setExecutionStreams(streams);
    return 0;
}

extern "C" int rt_net_set_graph(rtStereoNet* net, int on) {
    if (!net || !net->context) return fail("rt_net_set_graph: null pointer");
    net->context->setGraphMode(on != 0);
    return 0;
}

extern "C" int rt_net_set_launch_trace(rtStereoNet* net, int on) {
    if (!net || !net->context) return fail("rt_net_set_launch_trace: null pointer");
    net->context->setLaunchTrace(on != 0);
    return 0;
}

extern "C" int rt_net_read_launch_trace(rtStereoNet* net, unsigned long long* hashes, int max) {
    if (!net || !net->context || !hashes) { fail("rt_net_read_launch_trace: null pointer"); return -1; }
    const int n = net->context->readLaunchTrace(hashes, max);
    if (n < 0) fail("rt_net_read_launch_trace: read-back failed");
    return n;
}

extern "C" const char* rt_net_launch_name(const rtStereoNet* net, int launch) {
    return net && net->context ? net->context->getLaunchName(launch) : nullptr;
}

extern "C" long long rt_net_read_launch_output(rtStereoNet* net, int launch, void* host, long long bytes) {
    if (!net || !net->context) { fail("rt_net_read_launch_output: null pointer"); return -1; }
    const long long n = net->context->readLaunchOutput(launch, host, bytes);
    if (n < 0) fail("rt_net_read_launch_output: no traced pass, bad launch index or buffer too small");
    return n;
}

extern "C" int rt_net_num_layers(const rtStereoNet* net) { return net ? net->layers : 0; }
extern "C" int rt_net_num_launches(const rtStereoNet* net) { return net && net->engine ? net->engine->getNbLayers() : 0; }
extern "C" int rt_net_destroy(rtStereoNet* net) {
    delete net;
    return 0;
}
