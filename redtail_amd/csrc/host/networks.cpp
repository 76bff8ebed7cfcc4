// Programmatic Stereo DNN network builders (see include/networks.h).  Graph topology, layer names and
// weight names follow the reference generator: scripts/model_resnet18_2D.py:16-45,
// scripts/model_resnet18.py:12-85, scripts/model_nvsmall.py:12-73 and the TF-SAME padding rule of
// scripts/tensorrt_model_builder.py:140-147; the emitted layer sequence is the one found in
// sample_app/*_net.cpp.
#include "networks.h"

#include <cassert>
#include <stdexcept>
#include <vector>

#include "redtail_tensorrt_plugins.h"

namespace redtail { namespace tensorrt {

namespace {

struct Pad { int start, end; };
Pad tfSamePad(int in, int k, int s) {
    int along = (in % s == 0) ? std::max(k - s, 0) : std::max(k - (in % s), 0);
    return Pad{along / 2, along - along / 2};
}

struct Ctx {
    INetworkDefinition* net;
    IPluginContainer& plugins;
    const weight_map& w;
    DataType dt;
    ILogger& log;
    const Weights& at(const std::string& name) const {
        auto it = w.find(name);
        if (it == w.end()) {
            log.log(ILogger::Severity::kERROR, ("weights: tensor '" + name + "' is missing").c_str());
            throw std::out_of_range(name);
        }
        return it->second;
    }
};

ITensor* conv2d(Ctx& c, ITensor* x, const std::string& name, int maps, int k, int stride) {
    const Dims d = x->getDimensions();
    const Pad ph = tfSamePad(d.d[1], k, stride), pw = tfSamePad(d.d[2], k, stride);
    if (ph.start != ph.end || pw.start != pw.end) {
        // same restriction as the reference converter (tensorrt_model_builder.py:203-207)
        c.log.log(ILogger::Severity::kERROR, (name + ": input size needs asymmetric 2-D padding; use W,H = 1 (mod 8)").c_str());
        throw std::invalid_argument(name);
    }
    auto* l = c.net->addConvolution(*x, maps, DimsHW{k, k}, c.at(name + "_k"), c.at(name + "_b"));
    l->setName(name.c_str());
    l->setStride(DimsHW{stride, stride});
    l->setPadding(DimsHW{ph.start, pw.start});
    return l->getOutput(0);
}

ITensor* elu(Ctx& c, ITensor* x, const std::string& name) {
    auto* l = addElu(c.plugins, *c.net, *x, c.dt, name);
    l->setName(name.c_str());
    return l->getOutput(0);
}

ITensor* add(Ctx& c, ITensor* a, ITensor* b, const std::string& name) {
    auto* l = c.net->addElementWise(*a, *b, ElementWiseOperation::kSUM);
    l->setName(name.c_str());
    return l->getOutput(0);
}

ITensor* input(Ctx& c, const char* name, DimsCHW dims) {
    ITensor* t = c.net->addInput(name, DataType::kFLOAT, dims);
    const std::string s = std::string(name) + "_scale";
    auto* l = c.net->addScale(*t, ScaleMode::kUNIFORM, c.at(s + "_shift"), c.at(s + "_scale"), c.at(s + "_power"));
    l->setName(s.c_str());
    return l->getOutput(0);
}

// ResNet-style 2-D encoder of one side (model_resnet18.py:12-45); returns {encoder2D_out, conv1_act}
std::pair<ITensor*, ITensor*> resnetEncoder(Ctx& c, ITensor* x, const std::string& side) {
    ITensor* cur = elu(c, conv2d(c, x, side + "_conv1", 32, 5, 2), side + "_conv1_act");
    ITensor* conv1_act = cur;
    for (int i = 1; i <= 8; i++) {
        const std::string p = side + "_resblock" + std::to_string(i);
        ITensor* t = elu(c, conv2d(c, cur, p + "_conv1", 32, 3, 1), p + "_conv1_act");
        t = conv2d(c, t, p + "_conv2", 32, 3, 1);
        cur = elu(c, add(c, t, cur, p + "_conv2_add"), p + "_conv2_add_act");
    }
    return {conv2d(c, cur, side + "_encoder2D_out", 32, 3, 1), conv1_act};
}

// plain 5-conv encoder of NVSmall / NVTiny (model_nvsmall.py:13-28)
ITensor* plainEncoder(Ctx& c, ITensor* x, const std::string& side, int feat) {
    ITensor* cur = elu(c, conv2d(c, x, side + "_conv1", 32, 5, 2), side + "_conv1_act");
    for (const char* l : {"conv2", "conv3", "conv4"}) cur = elu(c, conv2d(c, cur, side + "_" + l, 32, 3, 1), side + "_" + l + "_act");
    return conv2d(c, cur, side + "_conv5", feat, 3, 1);
}

void markOutput(Ctx& c, ITensor* t, const char* name) {
    t->setName(name);
    c.net->markOutput(*t);
}

}  // namespace

INetworkDefinition* createResNet18_2DNetwork(IBuilder& builder, IPluginContainer& plugins, DimsCHW img_dims,
                                             const weight_map& weights, DataType data_type, int max_disp, ILogger& log) {
    INetworkDefinition* net = builder.createNetwork();
    Ctx c{net, plugins, weights, data_type, log};
    try {
        ITensor* left = input(c, "left", img_dims);
        ITensor* right = input(c, "right", img_dims);
        auto l = resnetEncoder(c, left, "left");
        auto r = resnetEncoder(c, right, "right");
        auto* cv = addCostVolume(plugins, *net, *l.first, *r.first, CostVolumeType::kCorrelation, max_disp, data_type, "cost_vol");
        cv->setName("cost_vol");
        auto* sa = addSoftargmax(plugins, *net, *cv->getOutput(0), SoftargmaxType::kMax, data_type, "softargmax_softargmax");
        sa->setName("softargmax");
        ITensor* cat_in[] = {l.second, sa->getOutput(0)};
        auto* cat = net->addConcatenation(cat_in, 2);
        cat->setName("concat");
        ITensor* cur = cat->getOutput(0);
        struct L { const char* name; int maps, stride; };
        ITensor* skip2 = nullptr;
        ITensor* skip5 = nullptr;
        for (const L& s : {L{"conv2D_1", 32, 1}, L{"conv2D_2", 32, 1}, L{"conv2D_3ds", 64, 2}, L{"conv2D_4", 64, 1},
                           L{"conv2D_5", 64, 1}, L{"conv2D_6ds", 128, 2}, L{"conv2D_7", 128, 1}, L{"conv2D_8", 128, 1}}) {
            cur = elu(c, conv2d(c, cur, s.name, s.maps, 3, s.stride), std::string(s.name) + "_act");
            if (std::string(s.name) == "conv2D_2") skip2 = cur;
            if (std::string(s.name) == "conv2D_5") skip5 = cur;
        }
        struct D { const char* name; int maps; ITensor* skip; };
        for (const D& s : {D{"deconv2D_1", 64, skip5}, D{"deconv2D_2", 32, skip2}, D{"deconv2D_3", 1, nullptr}}) {
            const std::string n = s.name;
            auto* dl = net->addDeconvolution(*cur, s.maps, DimsHW{3, 3}, c.at(n + "_k"), c.at(n + "_b"));
            dl->setName(s.name);
            dl->setStride(DimsHW{2, 2});
            dl->setPadding(DimsHW{1, 1});
            cur = dl->getOutput(0);
            if (s.skip) cur = elu(c, add(c, cur, s.skip, n + "_add_skip"), n + "_act");
        }
        auto* sig = net->addActivation(*cur, ActivationType::kSIGMOID);
        sig->setName("disp");
        markOutput(c, sig->getOutput(0), "disp");
    } catch (const std::exception&) {
        net->destroy();
        return nullptr;
    }
    return net;
}

INetworkDefinition* createStereo3DNetwork(IBuilder& builder, IPluginContainer& plugins, Stereo3DModel model, DimsCHW img_dims,
                                          const weight_map& weights, DataType data_type, int max_disp, ILogger& log) {
    struct C3 { std::string name; int K, C, stride; };
    struct D3 { std::string name; int K, C; std::string skip; };
    std::vector<C3> convs;
    std::vector<D3> deconvs;
    int feat = 32;
    if (model == Stereo3DModel::kResNet18) {
        convs = {{"conv3D_1a", 32, 64, 1}, {"conv3D_1b", 32, 32, 1}, {"conv3D_1ds", 64, 32, 2}, {"conv3D_2a", 64, 64, 1},
                 {"conv3D_2b", 64, 64, 1}, {"conv3D_2ds", 64, 64, 2}, {"conv3D_3a", 64, 64, 1}, {"conv3D_3b", 64, 64, 1},
                 {"conv3D_3ds", 64, 64, 2}, {"conv3D_4a", 64, 64, 1}, {"conv3D_4b", 64, 64, 1}, {"conv3D_4ds", 128, 64, 2},
                 {"conv3D_5a", 128, 128, 1}, {"conv3D_5b", 128, 128, 1}};
        deconvs = {{"deconv3D_1", 128, 64, "conv3D_4b"}, {"deconv3D_2", 64, 64, "conv3D_3b"}, {"deconv3D_3", 64, 64, "conv3D_2b"},
                   {"deconv3D_4", 64, 32, "conv3D_1b"}, {"deconv3D_5", 32, 1, ""}};
    } else {
        const int f = model == Stereo3DModel::kNVTiny ? 8 : 32;      // encoder feature maps
        const int b = model == Stereo3DModel::kNVTiny ? 16 : 32;     // base 3-D width
        feat = f;
        convs = {{"conv3D_1", b, 2 * f, 1}, {"conv3D_2", b, b, 1}, {"conv3D_3ds", 2 * b, b, 2}, {"conv3D_4", 2 * b, 2 * b, 1},
                 {"conv3D_5", 2 * b, 2 * b, 1}, {"conv3D_6ds", 4 * b, 2 * b, 2}, {"conv3D_7", 4 * b, 4 * b, 1},
                 {"conv3D_8", 4 * b, 4 * b, 1}};
        deconvs = {{"deconv3D_1", 4 * b, 2 * b, "conv3D_5"}, {"deconv3D_2", 2 * b, b, "conv3D_2"}, {"deconv3D_3", b, 1, ""}};
    }

    INetworkDefinition* net = builder.createNetwork();
    Ctx c{net, plugins, weights, data_type, log};
    try {
        ITensor* left = input(c, "left", img_dims);
        ITensor* right = input(c, "right", img_dims);
        ITensor *lf, *rf;
        if (model == Stereo3DModel::kResNet18) {
            lf = resnetEncoder(c, left, "left").first;
            rf = resnetEncoder(c, right, "right").first;
        } else {
            lf = plainEncoder(c, left, "left", feat);
            rf = plainEncoder(c, right, "right", feat);
        }
        auto* cv = addCostVolume(plugins, *net, *lf, *rf, CostVolumeType::kDefault, max_disp, data_type, "cost_vol");
        cv->setName("cost_vol");
        ITensor* cur = cv->getOutput(0);                                   // (D, 2C, H, W)
        std::unordered_map<std::string, ITensor*> acts;
        for (size_t i = 0; i < convs.size(); i++) {
            const C3& s = convs[i];
            const Dims in = cur->getDimensions();                          // (D, C, H, W)
            Dims stride{3, {s.stride, s.stride, s.stride}}, ps{3, {1, 1, 1}}, pe{3, {1, 1, 1}};
            if (s.stride == 2) {
                // a Pad plugin precedes every stride-2 conv (model_nvsmall.py:39-41); the D pads handed to the
                // conv are TF-SAME for the unpadded depth (tensorrt_model_builder.py:331-345)
                auto* pad = addPad(plugins, *net, *cur, {0, 0, 0, 0}, {1, 0, 0, 0}, s.name + "_pad");
                pad->setName((s.name + "_pad").c_str());
                cur = pad->getOutput(0);
                const Pad pd = tfSamePad(in.d[0], 3, 2), ph = tfSamePad(in.d[2], 3, 2), pw = tfSamePad(in.d[3], 3, 2);
                ps = Dims{3, {pd.start, ph.start, pw.start}};
                pe = Dims{3, {pd.end, ph.end, pw.end}};
            }
            auto* conv = addConv3D(plugins, *net, *cur, Conv3DType::kTensorFlow, Dims{5, {s.K, 3, s.C, 3, 3}}, stride, ps, pe,
                                   c.at(s.name + "_k"), c.at(s.name + "_b"), s.name);
            conv->setName(s.name.c_str());
            cur = conv->getOutput(0);                                      // (K, D, H, W)
            if (i + 1 != convs.size()) {                                   // the last conv feeds the decoder in KDHW
                auto* tr = addTransform(plugins, *net, *cur, {1, 0, 2, 3}, s.name + "_tran_transform");
                tr->setName((s.name + "_tran").c_str());
                cur = tr->getOutput(0);
            }
            cur = elu(c, cur, s.name + "_act");
            acts[s.name] = cur;
        }
        for (const D3& s : deconvs) {
            const Dims in = cur->getDimensions();                          // (K, Dy, Hy, Wy)
            int dx, hx, wx;
            if (!s.skip.empty()) {
                const Dims sk = acts.at(s.skip)->getDimensions();          // (D, C, H, W)
                dx = sk.d[0]; hx = sk.d[2]; wx = sk.d[3];
            } else {
                dx = 2 * in.d[1]; hx = 2 * in.d[2] - 1; wx = 2 * in.d[3] - 1;
            }
            const Pad pd = tfSamePad(dx, 3, 2);
            const bool asym = pd.start != pd.end;                          // even depth: compute D+1, then Slice
            Dims out_dims{4, {asym ? dx + 1 : dx, s.C, hx, wx}};
            Dims ps{3, {asym ? 0 : pd.start, 1, 1}}, pe{3, {asym ? 0 : pd.end, 1, 1}};
            auto* dc = addConv3DTranspose(plugins, *net, *cur, Conv3DType::kTensorFlow, Dims{5, {s.K, 3, s.C, 3, 3}}, out_dims,
                                          Dims{3, {2, 2, 2}}, ps, pe, c.at(s.name + "_k"), c.at(s.name + "_b"), s.name);
            dc->setName(s.name.c_str());
            cur = dc->getOutput(0);
            if (asym) {
                auto* sl = addSlice(plugins, *net, *cur, out_dims, {4, {0, 0, 0, 0}}, {4, {dx, s.C, hx, wx}}, s.name + "_slice");
                sl->setName((s.name + "_slice_layer").c_str());
                cur = sl->getOutput(0);
            }
            if (!s.skip.empty()) {
                cur = elu(c, add(c, cur, acts.at(s.skip), s.name + "_add_skip"), s.name + "_act");
                auto* tr = addTransform(plugins, *net, *cur, {1, 0, 2, 3}, s.name + "_transform_transform");
                tr->setName((s.name + "_transform").c_str());
                cur = tr->getOutput(0);
            }
        }
        auto* sa = addSoftargmax(plugins, *net, *cur, SoftargmaxType::kMin, data_type, "disp_softargmax");
        sa->setName("disp");
        markOutput(c, sa->getOutput(0), "disp");
    } catch (const std::exception&) {
        net->destroy();
        return nullptr;
    }
    return net;
}

INetworkDefinition* createResNet18_2D_513x257Network(IBuilder& b, IPluginContainer& p, DimsCHW d, const weight_map& w, DataType t, ILogger& l) {
    return createResNet18_2DNetwork(b, p, d, w, t, 48, l);
}
INetworkDefinition* createNVSmall1025x321Network(IBuilder& b, IPluginContainer& p, DimsCHW d, const weight_map& w, DataType t, ILogger& l) {
    return createStereo3DNetwork(b, p, Stereo3DModel::kNVSmall, d, w, t, 48, l);
}
INetworkDefinition* createNVTiny513x161Network(IBuilder& b, IPluginContainer& p, DimsCHW d, const weight_map& w, DataType t, ILogger& l) {
    return createStereo3DNetwork(b, p, Stereo3DModel::kNVTiny, d, w, t, 24, l);
}
INetworkDefinition* createResNet18_1025x321Network(IBuilder& b, IPluginContainer& p, DimsCHW d, const weight_map& w, DataType t, ILogger& l) {
    return createStereo3DNetwork(b, p, Stereo3DModel::kResNet18, d, w, t, 68, l);
}

} }  // namespace redtail::tensorrt
