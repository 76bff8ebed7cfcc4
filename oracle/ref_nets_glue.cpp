// TEST INFRASTRUCTURE.  C entry point around the REFERENCE's generated network builders
// (/root/reference/stereoDNN/sample_app/*_net.cpp, compiled untouched into oracle/_ref/libref_nets.so by
// redtail_amd/build.py:build_ref_link_check) so that tests can execute the reference-defined graphs on top
// of our plugins/executor and compare them with our programmatic builders (include/networks.h).
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "NvInfer.h"
#include "redtail_tensorrt_plugins.h"

namespace redtail { namespace tensorrt {
using weight_map = std::unordered_map<std::string, nvinfer1::Weights>;
// declarations of the reference's sample_app/networks.h:20-36 (the definitions come from its *_net.cpp)
nvinfer1::INetworkDefinition* createNVSmall1025x321Network(nvinfer1::IBuilder&, IPluginContainer&, nvinfer1::DimsCHW, const weight_map&, nvinfer1::DataType, nvinfer1::ILogger&);
nvinfer1::INetworkDefinition* createNVTiny513x161Network(nvinfer1::IBuilder&, IPluginContainer&, nvinfer1::DimsCHW, const weight_map&, nvinfer1::DataType, nvinfer1::ILogger&);
nvinfer1::INetworkDefinition* createResNet18_1025x321Network(nvinfer1::IBuilder&, IPluginContainer&, nvinfer1::DimsCHW, const weight_map&, nvinfer1::DataType, nvinfer1::ILogger&);
nvinfer1::INetworkDefinition* createResNet18_2D_513x257Network(nvinfer1::IBuilder&, IPluginContainer&, nvinfer1::DimsCHW, const weight_map&, nvinfer1::DataType, nvinfer1::ILogger&);
} }

using namespace nvinfer1;
using namespace redtail::tensorrt;

namespace {
struct QuietLogger : ILogger {
    std::string last;
    void log(Severity s, const char* m) override { if ((int)s <= 1) last = m; }
};
struct RefNet {
    QuietLogger log;
    std::vector<char> blob;
    weight_map weights;
    std::unique_ptr<IPluginContainer> plugins;
    ICudaEngine* engine = nullptr;
    IExecutionContext* ctx = nullptr;
};
}

// model: 0 resnet18_2D (513x257), 1 nvsmall (1025x321), 2 nvtiny (513x161), 3 resnet18 (1025x321)
// half != 0: the blob holds fp16 weights and the network is built the way sample_app/main.cpp does for `fp16`
// (main.cpp:126, 248, 256-262): Weights of type kHALF, plugins created for kHALF (ResNet-18 2D only), setHalf2Mode(true)
static void* create(int model, int width, int height, const void* blob, size_t bytes, int half) {
    auto* n = new RefNet();
    n->blob.assign((const char*)blob, (const char*)blob + bytes);
    size_t off = 0;
    while (off < n->blob.size()) {
        std::string name(n->blob.data() + off);
        off += name.size() + 1;
        uint32_t count;
        memcpy(&count, n->blob.data() + off, 4);
        off += 4;
        n->weights[name] = Weights{half ? DataType::kHALF : DataType::kFLOAT, n->blob.data() + off, (int64_t)count};
        off += (size_t)count * (half ? 2 : 4);
    }
    n->plugins = IPluginContainer::create(n->log);
    IBuilder* b = createInferBuilder(n->log);
    DimsCHW d{3, height, width};
    INetworkDefinition* net = nullptr;
    if (model == 0) net = createResNet18_2D_513x257Network(*b, *n->plugins, d, n->weights, half ? DataType::kHALF : DataType::kFLOAT, n->log);
    if (model == 1) net = createNVSmall1025x321Network(*b, *n->plugins, d, n->weights, DataType::kFLOAT, n->log);
    if (model == 2) net = createNVTiny513x161Network(*b, *n->plugins, d, n->weights, DataType::kFLOAT, n->log);
    if (model == 3) net = createResNet18_1025x321Network(*b, *n->plugins, d, n->weights, DataType::kFLOAT, n->log);
    if (net) {
        b->setMaxBatchSize(1);
        b->setHalf2Mode(half != 0);
        n->engine = b->buildCudaEngine(*net);
        net->destroy();
    }
    b->destroy();
    if (!n->engine) { delete n; return nullptr; }
    n->ctx = n->engine->createExecutionContext();
    return n;
}
extern "C" void* ref_net_create(int model, int width, int height, const void* blob, size_t bytes) {
    return create(model, width, height, blob, bytes, 0);
}
extern "C" void* ref_net_create_half(int model, int width, int height, const void* blob, size_t bytes) {
    return create(model, width, height, blob, bytes, 1);
}
extern "C" int ref_net_execute(void* h, void* left, void* right, void* disp) {
    auto* n = (RefNet*)h;
    void* bindings[3] = {left, right, disp};
    return n->ctx->execute(1, bindings) ? 0 : -1;
}
extern "C" int ref_net_num_launches(void* h) { return ((RefNet*)h)->engine->getNbLayers(); }
extern "C" void ref_net_destroy(void* h) {
    auto* n = (RefNet*)h;
    if (!n) return;
    n->ctx->destroy();
    n->engine->destroy();
    delete n;
}
