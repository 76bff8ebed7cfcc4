// Multi-GPU Stereo DNN throughput without Python: one process, one host thread per MI355X, the weight file read once and broadcast
// over RCCL / xGMI, stereo pairs sharded contiguously over the devices, no data-path collective.
//
// What sample_app/main.cpp:136-340 of the reference does for one GPU and one pair (read trt_weights.bin, build, execute in a loop, report
// ms per pair) for N GPUs: rank 0 reads the file; rt_comm_init_all (ncclCommInitAll) gives every device thread a communicator;
// rt_net_create_broadcast ships the image (ncclBroadcast, first the size, then the bytes) and builds the engine on each device; every
// thread then runs its shard of the pairs round-robin over --contexts execution contexts with one stream each (the throughput
// set-up of bench.py) on synthetic KITTI-shaped pairs resident in HBM.  Prints one JSON line: pairs/s of the whole job (pairs of all
// devices / slowest device's time), per-device rates and the crc32 of the weight image each device received.
//
//   stereo_throughput <model: resnet18_2D|nvsmall|nvtiny|resnet18> <width> <height> <weights.bin> [--fp16] [--gpus N] [--pairs P]
//                     [--batch B] [--contexts C] [--warmup W]
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include "rt_stereo.h"
#include "rt_stereo_net.h"

namespace {

struct Shard { int first, count; };
// contiguous shards, remainder to the first ranks (redtail_amd/parallel.py: shard)
Shard shard(int total, int rank, int world) {
    const int q = total / world, r = total % world;
    return Shard{rank * q + (rank < r ? rank : r), q + (rank < r ? 1 : 0)};
}

struct Barrier {
    std::atomic<int> count{0}, gen{0};
    int n = 1;
    void wait() {
        const int g = gen.load();
        if (count.fetch_add(1) + 1 == n) { count.store(0); gen.fetch_add(1); }
        else while (gen.load() == g) std::this_thread::yield();
    }
};

struct Result {
    double seconds = 0, pairs = 0;
    uint32_t crc = 0;
    std::string error;
};

}  // namespace

int main(int argc, char** argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: %s <resnet18_2D|nvsmall|nvtiny|resnet18> <width> <height> <weights.bin> [--fp16] [--gpus N] [--pairs P] [--batch B] [--contexts C] [--warmup W]\n", argv[0]);
        return 2;
    }
    const std::string model_name = argv[1];
    const int width = atoi(argv[2]), height = atoi(argv[3]);
    const char* path = argv[4];
    int gpus = 0, pairs = 240, batch = 1, contexts = 3, warmup = 12, fp16 = 0;      // (three contexts: profiles/r06_contexts.txt)
    for (int i = 5; i < argc; i++) {
        const std::string a = argv[i];
        auto val = [&]() { return i + 1 < argc ? atoi(argv[++i]) : 0; };
        if (a == "--fp16") fp16 = 1;
        else if (a == "--gpus") gpus = val();
        else if (a == "--pairs") pairs = val();
        else if (a == "--batch") batch = val();
        else if (a == "--contexts") contexts = val();
        else if (a == "--warmup") warmup = val();
        else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    int model = -1;
    const char* names[] = {"resnet18_2D", "nvsmall", "nvtiny", "resnet18"};
    for (int m = 0; m < 4; m++)
        if (model_name == names[m]) model = m;
    if (model < 0 || width < 17 || height < 17 || batch < 1 || contexts < 1 || pairs < 1) { fprintf(stderr, "bad arguments\n"); return 2; }
    const int ndev = rt_device_count();
    if (ndev < 1) { fprintf(stderr, "no HIP device: %s\n", rt_last_error_string()); return 1; }
    if (gpus <= 0) gpus = ndev;
    if (gpus > ndev) { fprintf(stderr, "--gpus %d but only %d device(s) are visible\n", gpus, ndev); return 1; }

    std::vector<char> blob;
    {
        std::ifstream f(path, std::ios::binary | std::ios::ate);
        if (!f.is_open()) { fprintf(stderr, "cannot open %s\n", path); return 1; }
        blob.resize((size_t)f.tellg());
        f.seekg(0);
        f.read(blob.data(), (std::streamsize)blob.size());
    }
    std::vector<rtComm*> comms((size_t)gpus, nullptr);
    std::vector<int> devices((size_t)gpus);
    for (int d = 0; d < gpus; d++) devices[d] = d;
    if (rt_comm_init_all(comms.data(), gpus, devices.data()) != 0) { fprintf(stderr, "rt_comm_init_all: %s\n", rt_last_error_string()); return 1; }

    std::vector<Result> res((size_t)gpus);
    Barrier bar;
    bar.n = gpus;
    auto worker = [&](int rank) {
        Result& out = res[rank];
        auto fail = [&](const char* what, const char* msg) { out.error = std::string(what) + ": " + msg; };
        if (rt_set_device(rank) != 0) { fail("rt_set_device", rt_last_error_string()); bar.wait(); bar.wait(); return; }
        // 1. weights: rank 0 owns the file image, everyone else receives it over RCCL
        std::vector<rtStereoNet*> nets((size_t)contexts, nullptr);
        bool ok = true;
        for (int c = 0; c < contexts && ok; c++) {
            // the first context of a device takes part in the broadcast; further contexts of the same device build from that image
            int rc;
            if (c == 0) rc = rt_net_create_broadcast(&nets[c], model, width, height, batch, fp16 ? RT_F16 : RT_F32, 0, rank == 0 ? blob.data() : nullptr,
                                                     rank == 0 ? blob.size() : 0, comms[rank], 0);
            else {                                        // ... from the image that arrived over RCCL, not from the file
                const void* image = nullptr;
                size_t image_bytes = 0;
                rc = rt_net_weights_image(nets[0], &image, &image_bytes);
                if (rc == 0) rc = rt_net_create_from_memory(&nets[c], model, width, height, batch, fp16 ? RT_F16 : RT_F32, 0, image, image_bytes);
            }
            if (rc != 0) { fail("engine", rt_net_last_error()); ok = false; }
            if (ok) rt_net_set_streams(nets[c], contexts > 1 ? 1 : 2);
        }
        if (ok) rt_net_weights_crc32(nets[0], &out.crc);
        // 2. this rank's shard of the pairs, synthetic, resident in HBM
        const Shard sh = shard(pairs, rank, gpus);
        const size_t img = (size_t)batch * 3 * height * width, dsp = (size_t)batch * height * width;
        void *left = nullptr, *right = nullptr;
        std::vector<void*> disp((size_t)contexts, nullptr), streams((size_t)contexts, nullptr);
        if (ok) {
            std::vector<float> l(img), r(img);
            for (size_t i = 0; i < img; i++) {            // smooth texture + shift: values in [0, 1]
                const int x = (int)(i % width), y = (int)((i / width) % height);
                l[i] = 0.5f + 0.5f * std::sin(0.05f * x + 0.031f * y + 0.7f * rank);
                r[i] = 0.5f + 0.5f * std::sin(0.05f * (x + 9) + 0.031f * y + 0.7f * rank);
            }
            ok = rt_malloc(&left, img * 4) == 0 && rt_malloc(&right, img * 4) == 0 && rt_memcpy_h2d(left, l.data(), img * 4, nullptr) == 0 &&
                 rt_memcpy_h2d(right, r.data(), img * 4, nullptr) == 0;
            for (int c = 0; c < contexts && ok; c++) ok = rt_malloc(&disp[c], dsp * 4) == 0 && rt_stream_create(&streams[c]) == 0;
            if (!ok) fail("device memory", rt_last_error_string());
        }
        auto step = [&](int i) { return rt_net_execute(nets[i % contexts], left, right, disp[i % contexts], batch, streams[i % contexts]); };
        auto sync_all = [&]() { for (int c = 0; c < contexts; c++) rt_stream_sync(streams[c]); };
        if (ok) {
            for (int i = 0; i < warmup && ok; i++) ok = step(i) == 0;
            sync_all();
        }
        bar.wait();                                       // every device starts its timed region together
        const int steps = (sh.count + batch - 1) / batch;
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < steps && ok; i++) ok = step(i) == 0;
        if (ok) sync_all();
        out.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        out.pairs = (double)steps * batch;
        if (!ok && out.error.empty()) fail("execute", rt_net_last_error());
        bar.wait();
        if (ok) {                                         // the disparity must be finite
            std::vector<float> h(dsp);
            rt_memcpy_d2h(h.data(), disp[0], dsp * 4, nullptr);
            rt_stream_sync(nullptr);
            for (float v : h)
                if (!std::isfinite(v)) { fail("result", "non-finite disparity"); break; }
        }
        for (auto* n : nets)
            if (n) rt_net_destroy(n);
        for (int c = 0; c < contexts; c++) { if (disp[c]) rt_free(disp[c]); if (streams[c]) rt_stream_destroy(streams[c]); }
        if (left) rt_free(left);
        if (right) rt_free(right);
    };
    std::vector<std::thread> threads;
    for (int d = 0; d < gpus; d++) threads.emplace_back(worker, d);
    for (auto& t : threads) t.join();
    for (auto* c : comms) rt_comm_destroy(c);

    double slowest = 0, total = 0;
    bool ok = true;
    for (int d = 0; d < gpus; d++) {
        if (!res[d].error.empty()) { fprintf(stderr, "device %d: %s\n", d, res[d].error.c_str()); ok = false; }
        slowest = res[d].seconds > slowest ? res[d].seconds : slowest;
        total += res[d].pairs;
        if (res[d].crc != res[0].crc) { fprintf(stderr, "device %d received a different weight image (crc32 %08x vs %08x)\n", d, res[d].crc, res[0].crc); ok = false; }
    }
    if (!ok) return 1;
    printf("{\"metric\": \"stereo pairs/sec, %s %dx%d (native multi-GPU driver)\", \"value\": %.2f, \"unit\": \"pairs/s\", \"n_gpus\": %d, \"pairs\": %.0f, "
           "\"batch\": %d, \"contexts_per_gpu\": %d, \"seconds\": %.4f, \"scaling\": \"weak\", \"weights_crc32\": \"%08x\", \"ranks\": [",
           model_name.c_str(), width, height, total / slowest, gpus, total, batch, contexts, slowest, res[0].crc);
    for (int d = 0; d < gpus; d++) printf("%s{\"rank\": %d, \"pairs_per_s\": %.2f, \"weights_crc32\": \"%08x\"}", d ? ", " : "", d, res[d].pairs / res[d].seconds, res[d].crc);
    printf("]}\n");
    return 0;
}
