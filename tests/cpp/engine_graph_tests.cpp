// Executor passes on graphs that the four Stereo DNN models do not contain -- built through the public C++ API (NvInfer.h subset +
// redtail_tensorrt_plugins.h), executed on whatever library the binary is linked with (GPU build, or the SIMT emulator build in the
// CPU test tier).  Every case compares the default engine (all fusions, two streams, siamese merge) with the same network built with
// RT_NO_FUSION=1 RT_SINGLE_STREAM=1 (one launch per layer, one stream).  Test infrastructure; driven by tests/test_engine_graphs.py.
//
//   1. two_tower_concat : a concatenation that only depends on the SECOND input (side stream) feeding a main-stream convolution
//                         (foldConcats() erased the concatenation after the stream pass -- the consumer's event was never recorded)
//   2. siamese_custom   : twin towers with shared weights on a graph of our own, maxBatchSize 3, batches 1..3 (EngineImpl::mergeSiamese)
//   3. siamese_unequal  : the same with one differing bias -- nothing downstream of it may be merged, results still right
//   4. standalone_corr  : CostVolumePlugin(kCorrelation) as a launch of its own on maps of a network's width: the default engine runs it
//                         on the matrix cores (3-term fp16 split), an engine built with IBuilder::setExactFp32Mode on the fp32 kernel
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <random>
#include <string>
#include <vector>

#include "NvInfer.h"
#include "cuda_runtime_api.h"
#include "redtail_tensorrt_plugins.h"
#include "rt_stereo.h"

using namespace nvinfer1;
using namespace redtail::tensorrt;

namespace {

struct Logger : public ILogger {
    std::vector<std::string> info;
    void log(Severity s, const char* msg) override {
        if ((int)s <= (int)Severity::kERROR) fprintf(stderr, "[engine] %s\n", msg);
        else info.push_back(msg);
    }
};

struct WeightPool {
    std::mt19937 rng{42};
    std::vector<std::unique_ptr<std::vector<float>>> store;
    Weights make(size_t n, float scale) {
        std::normal_distribution<float> nd(0.f, scale);
        store.emplace_back(new std::vector<float>(n));
        for (auto& v : *store.back()) v = nd(rng);
        return Weights{DataType::kFLOAT, store.back()->data(), (int64_t)n};
    }
    Weights copy(const Weights& w) {          // same bytes at a different address (what a weight file gives the two sides)
        store.emplace_back(new std::vector<float>((const float*)w.values, (const float*)w.values + w.count));
        return Weights{DataType::kFLOAT, store.back()->data(), w.count};
    }
};

struct ConvW { Weights k, b; };

ITensor* conv(INetworkDefinition& net, IPluginContainer& pc, ITensor& x, int maps, const ConvW& w, const std::string& name, bool elu) {
    auto* l = net.addConvolution(x, maps, DimsHW{3, 3}, w.k, w.b);
    l->setName(name.c_str());
    l->setStride(DimsHW{1, 1});
    l->setPadding(DimsHW{1, 1});
    ITensor* t = l->getOutput(0);
    if (elu) {
        auto* e = addElu(pc, net, *t, DataType::kFLOAT, name + "_act");
        e->setName((name + "_act").c_str());
        t = e->getOutput(0);
    }
    return t;
}

typedef std::function<void(INetworkDefinition&, IPluginContainer&, WeightPool&)> GraphFn;

struct Result {
    std::vector<float> out;
    int launches = 0;
    bool ok = false;
    std::vector<std::string> info;
};

Result run(const GraphFn& graph, int C, int H, int W, int max_batch, int batch, const std::vector<float>& l, const std::vector<float>& r, int out_c,
           bool exact_fp32 = false) {
    Result res;
    Logger log;
    auto plugins = IPluginContainer::create(log);
    IBuilder* builder = createInferBuilder(log);
    INetworkDefinition* net = builder->createNetwork();
    WeightPool pool;
    graph(*net, *plugins, pool);
    builder->setMaxBatchSize(max_batch);
    if (exact_fp32) builder->setExactFp32Mode(true);
    ICudaEngine* engine = builder->buildCudaEngine(*net);
    net->destroy();
    builder->destroy();
    if (!engine) return res;
    IExecutionContext* ctx = engine->createExecutionContext();
    res.launches = engine->getNbLayers();
    const size_t in_bytes = (size_t)batch * C * H * W * 4, out_elems = (size_t)batch * out_c * H * W;
    void *dl = nullptr, *dr = nullptr, *dout = nullptr;
    if (rt_malloc(&dl, in_bytes) || rt_malloc(&dr, in_bytes) || rt_malloc(&dout, out_elems * 4)) return res;
    rt_memcpy_h2d(dl, l.data(), in_bytes, nullptr);
    rt_memcpy_h2d(dr, r.data(), in_bytes, nullptr);
    void* bindings[3];
    bindings[engine->getBindingIndex("left")] = dl;
    bindings[engine->getBindingIndex("right")] = dr;
    bindings[engine->getBindingIndex("out")] = dout;
    res.out.assign(out_elems, 0.f);
    bool ok = true;
    for (int rep = 0; rep < 3 && ok; rep++) ok = ctx->execute(batch, bindings);      // repeated: events and buffers are reused
    rt_memcpy_d2h(res.out.data(), dout, out_elems * 4, nullptr); rt_stream_sync(nullptr);
    rt_free(dl); rt_free(dr); rt_free(dout);
    ctx->destroy();
    engine->destroy();
    res.ok = ok;
    res.info = log.info;
    return res;
}

double max_diff(const std::vector<float>& a, const std::vector<float>& b) {
    if (a.size() != b.size()) return 1e30;
    double m = 0;
    for (size_t i = 0; i < a.size(); i++) {
        const double d = std::fabs((double)a[i] - b[i]);
        if (!(d <= m)) m = d;          // NaN propagates
    }
    return m;
}

int g_failed = 0, g_ran = 0;
#define CHECK(cond, ...) do { if (!(cond)) { printf("  FAILED %s:%d: %s -- ", __FILE__, __LINE__, #cond); printf(__VA_ARGS__); printf("\n"); g_failed++; return; } } while (0)

std::vector<float> image(int n, int C, int H, int W, unsigned seed) {
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    std::vector<float> v((size_t)n * C * H * W);
    for (auto& x : v) x = u(rng);
    return v;
}

void with_env(const char* const* names, const std::function<void()>& f) {
    for (const char* const* n = names; *n; n++) setenv(*n, "1", 1);
    f();
    for (const char* const* n = names; *n; n++) unsetenv(*n);
}

// ---- 1 ---------------------------------------------------------------------------------------------------------------------
void two_tower_concat() {
    g_ran++;
    printf("[ RUN ] two_tower_concat\n");
    const int H = 21, W = 37;
    GraphFn g = [&](INetworkDefinition& net, IPluginContainer& pc, WeightPool& p) {
        ITensor* L = net.addInput("left", DataType::kFLOAT, DimsCHW{3, H, W});
        ITensor* R = net.addInput("right", DataType::kFLOAT, DimsCHW{3, H, W});
        auto cw = [&](int co, int ci) { return ConvW{p.make((size_t)co * ci * 9, 0.25f), p.make(co, 0.1f)}; };
        ITensor* a = conv(net, pc, *L, 8, cw(8, 3), "a", true);                 // main stream
        ITensor* b1 = conv(net, pc, *R, 4, cw(4, 3), "b1", true);               // side stream from here ...
        ITensor* b2 = conv(net, pc, *b1, 4, cw(4, 4), "b2", true);
        ITensor* b3 = conv(net, pc, *b2, 4, cw(4, 4), "b3", true);
        ITensor* b4 = conv(net, pc, *b3, 4, cw(4, 4), "b4", true);
        ITensor* b5 = conv(net, pc, *b3, 4, cw(4, 4), "b5", true);
        ITensor* cat_in[] = {b4, b5};
        auto* cat = net.addConcatenation(cat_in, 2);                            // ... to here: depends on the second input only
        cat->setName("cat");
        ITensor* c = conv(net, pc, *cat->getOutput(0), 8, cw(8, 8), "c", false);
        auto* add = net.addElementWise(*c, *a, ElementWiseOperation::kSUM);     // the consumer mixes both towers: main stream
        add->setName("c_add");
        auto* e = addElu(pc, net, *add->getOutput(0), DataType::kFLOAT, "c_act");
        e->setName("c_act");
        e->getOutput(0)->setName("out");
        net.markOutput(*e->getOutput(0));
    };
    const auto l = image(2, 3, H, W, 1), r = image(2, 3, H, W, 2);
    Result fused = run(g, 3, H, W, 2, 2, l, r, 8), plain;
    const char* env[] = {"RT_NO_FUSION", "RT_SINGLE_STREAM", nullptr};
    with_env(env, [&] { plain = run(g, 3, H, W, 2, 2, l, r, 8); });
    CHECK(fused.ok && plain.ok, "engine did not build / run");
    CHECK(fused.launches < plain.launches, "nothing was fused: %d vs %d launches", fused.launches, plain.launches);
    bool folded = false;
    for (auto& s : fused.info) folded = folded || s.find("concatenation folded") != std::string::npos;
    CHECK(folded, "the side-stream concatenation was not folded (the case this test is about)");
    const double d = max_diff(fused.out, plain.out);
    CHECK(d <= 2e-5, "fused two-stream engine differs from the layer-by-layer one by %g", d);
    printf("[  OK  ] two_tower_concat: %d vs %d launches, max diff %.3g\n", fused.launches, plain.launches, d);
}

// ---- 2, 3 ------------------------------------------------------------------------------------------------------------------
GraphFn siamese_graph(int H, int W, bool unequal) {
    return [=](INetworkDefinition& net, IPluginContainer& pc, WeightPool& p) {
        ITensor* in[2] = {net.addInput("left", DataType::kFLOAT, DimsCHW{3, H, W}), net.addInput("right", DataType::kFLOAT, DimsCHW{3, H, W})};
        auto cw = [&](int co, int ci) { return ConvW{p.make((size_t)co * ci * 9, 0.2f), p.make(co, 0.1f)}; };
        const ConvW w1 = cw(8, 3), w2 = cw(8, 8), w3 = cw(8, 8), w4 = cw(8, 8);
        ITensor* feat[2];
        for (int s = 0; s < 2; s++) {
            const std::string n = s ? "r_" : "l_";
            auto side = [&](const ConvW& w) { return s ? ConvW{p.copy(w.k), p.copy(w.b)} : w; };
            ConvW w3s = side(w3);
            if (s && unequal) {
                w3s.b = p.copy(w3.b);
                ((float*)w3s.b.values)[2] += 0.25f;
            }
            ITensor* t1 = conv(net, pc, *in[s], 8, side(w1), n + "c1", true);
            ITensor* t2 = conv(net, pc, *t1, 8, side(w2), n + "c2", true);
            ITensor* t3 = conv(net, pc, *t2, 8, w3s, n + "c3", false);
            auto* add = net.addElementWise(*t3, *t1, ElementWiseOperation::kSUM);
            add->setName((n + "add").c_str());
            auto* e = addElu(pc, net, *add->getOutput(0), DataType::kFLOAT, n + "add_act");
            e->setName((n + "add_act").c_str());
            feat[s] = conv(net, pc, *e->getOutput(0), 8, side(w4), n + "c4", true);
        }
        auto* sum = net.addElementWise(*feat[0], *feat[1], ElementWiseOperation::kSUM);
        sum->setName("join");
        ITensor* y = conv(net, pc, *sum->getOutput(0), 4, cw(4, 8), "head", false);
        y->setName("out");
        net.markOutput(*y);
    };
}

void siamese(bool unequal) {
    g_ran++;
    const char* name = unequal ? "siamese_unequal" : "siamese_custom";
    printf("[ RUN ] %s\n", name);
    const int H = 19, W = 33;
    const GraphFn g = siamese_graph(H, W, unequal);
    const bool quick = getenv("RT_TEST_QUICK") != nullptr;      // the CPU tier (SIMT emulator): batch 1 and the maximum, not the one in between
    for (int batch = 1; batch <= 3; batch++) {
        if (quick && batch == 2) continue;
        const auto l = image(batch, 3, H, W, 10 + batch), r = image(batch, 3, H, W, 20 + batch);
        Result merged = run(g, 3, H, W, 3, batch, l, r, 4), apart, plain;
        const char* e1[] = {"RT_NO_SIAMESE", nullptr};
        with_env(e1, [&] { apart = run(g, 3, H, W, 3, batch, l, r, 4); });
        const char* e2[] = {"RT_NO_FUSION", "RT_SINGLE_STREAM", nullptr};
        with_env(e2, [&] { plain = run(g, 3, H, W, 3, batch, l, r, 4); });
        CHECK(merged.ok && apart.ok && plain.ok, "engine did not build / run (batch %d)", batch);
        // conv1 reads the bindings and cannot pair up; c2, c3 (+add+ELU), c4 can -- c3 and c4 not when the towers differ from c3 on
        const int expect = unequal ? 1 : 3;
        CHECK(apart.launches - merged.launches == expect, "batch %d: %d launches merged, expected %d", batch, apart.launches - merged.launches, expect);
        const double d0 = max_diff(merged.out, apart.out), d1 = max_diff(merged.out, plain.out);
        CHECK(d0 == 0.0, "batch %d: merged towers differ from separate launches by %g (must be bit-identical)", batch, d0);
        CHECK(d1 <= 2e-5, "batch %d: differs from the layer-by-layer engine by %g", batch, d1);
    }
    printf("[  OK  ] %s\n", name);
}

// ---- 4 ---------------------------------------------------------------------------------------------------------------------
void standalone_corr() {
    g_ran++;
    printf("[ RUN ] standalone_corr\n");
    const int C = 16, H = 3, W = 70, D = 12;
    GraphFn g = [&](INetworkDefinition& net, IPluginContainer& pc, WeightPool&) {
        ITensor* L = net.addInput("left", DataType::kFLOAT, DimsCHW{C, H, W});
        ITensor* R = net.addInput("right", DataType::kFLOAT, DimsCHW{C, H, W});
        auto* cv = addCostVolume(pc, net, *L, *R, CostVolumeType::kCorrelation, D, DataType::kFLOAT, "corr");
        cv->setName("corr");
        cv->getOutput(0)->setName("out");
        net.markOutput(*cv->getOutput(0));
    };
    const auto l = image(2, C, H, W, 31), r = image(2, C, H, W, 32);
    Result mm = run(g, C, H, W, 2, 2, l, r, D), exact = run(g, C, H, W, 2, 2, l, r, D, true), valu;
    const char* env[] = {"RT_NO_CORR_MFMA_PLANAR", nullptr};
    with_env(env, [&] { valu = run(g, C, H, W, 2, 2, l, r, D); });
    CHECK(mm.ok && exact.ok && valu.ok, "engine did not build / run");
    // fp64 reference and the sum of magnitudes the split product's error is relative to
    double worst = 0, worst_exact = 0;
    for (int n = 0; n < 2; n++)
        for (int d = 0; d < D; d++)
            for (int y = 0; y < H; y++)
                for (int x = 0; x < W; x++) {
                    double ref = 0, mag = 0;
                    if (x >= d)
                        for (int c = 0; c < C; c++) {
                            const double a = l[(((size_t)n * C + c) * H + y) * W + x], b = r[(((size_t)n * C + c) * H + y) * W + x - d];
                            ref += a * b; mag += std::fabs(a * b);
                        }
                    const size_t o = (((size_t)n * D + d) * H + y) * W + x;
                    const double e = std::fabs(mm.out[o] - ref), ee = std::fabs(exact.out[o] - ref);
                    if (mag > 0) { worst = std::max(worst, e / mag); worst_exact = std::max(worst_exact, ee / mag); }
                    else CHECK(mm.out[o] == 0.f && exact.out[o] == 0.f, "x < d must be exactly zero");
                }
    CHECK(worst <= 4.8e-7, "matrix-core correlation: |err| / sum |l r| = %g", worst);
    CHECK(worst_exact <= 4.8e-7, "fp32 correlation: |err| / sum |l r| = %g", worst_exact);
    CHECK(max_diff(exact.out, valu.out) == 0.0, "the exact-fp32 engine did not run the fp32 kernel");
    CHECK(max_diff(mm.out, valu.out) > 0.0, "the default engine did not run the matrix-core kernel (same bits as the fp32 kernel)");
    printf("[  OK  ] standalone_corr: |err| / sum |l r| %.3g (matrix cores), %.3g (exact fp32)\n", worst, worst_exact);
}

}  // namespace

int main() {
    setenv("RT_DEV_KNOBS", "1", 1);       // RT_NO_FUSION / RT_SINGLE_STREAM / RT_NO_SIAMESE below are development knobs
    two_tower_concat();
    siamese(false);
    siamese(true);
    standalone_corr();
    printf("%s %d of %d engine graph tests\n", g_failed ? "FAILED" : "PASSED", g_failed ? g_failed : g_ran, g_ran);
    return g_failed ? 1 : 0;
}
