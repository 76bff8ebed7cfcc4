// Fiber-based SIMT emulator runtime (see hip/hip_runtime.h in this directory).  TEST INFRASTRUCTURE ONLY.
#include <hip/hip_runtime.h>

#include <chrono>
#include <vector>

extern "C" void hipemu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size hipemu_switch,.-hipemu_switch
)");

namespace hipemu {

enum State { READY, WAIT_WAVE, WAIT_BLOCK, DONE };
static const size_t kStack = 256 * 1024;

struct Wave {
    int nlanes = 0;      // lanes that exist in this wave
    int alive = 0;       // not DONE
    int arrived = 0;     // blocked at the current wave op
    int op = 0;          // op id of the pending wave op
    // per-lane operand / result slots
    float a[64], b[64];
    f32x16 c16[64], d16[64];
    f16x8 ha[64], hb[64];
    int isrc[64];
    unsigned long long mask;
};

struct Thread {
    void* sp = nullptr;
    char* stack = nullptr;
    State st = DONE;
    uint3 tid;
    int lane, wave;
};

Thread* cur = nullptr;
uint3 g_tid, g_bid;
dim3 g_bdim, g_gdim;

static void* sched_sp = nullptr;
static std::vector<Thread> threads;
static std::vector<Wave> waves;
static std::vector<char> dsmem;
static const std::function<void()>* g_body = nullptr;
static int block_alive = 0, block_arrived = 0;

void* dyn_smem() { return dsmem.data(); }
int lane_id() { return cur->lane; }

[[noreturn]] static void die(const char* msg) {
    fprintf(stderr, "hipemu: %s (block %u,%u,%u thread %u,%u,%u)\n", msg, g_bid.x, g_bid.y, g_bid.z, g_tid.x,
            g_tid.y, g_tid.z);
    abort();
}

static void yield_to_sched() { hipemu_switch(&cur->sp, sched_sp); }

static void fiber_entry() {
    (*g_body)();
    cur->st = DONE;
    yield_to_sched();
    die("resumed a finished fiber");
}

static void init_fiber(Thread& t) {
    if (!t.stack) t.stack = (char*)aligned_alloc(64, kStack);
    uintptr_t top = ((uintptr_t)t.stack + kStack) & ~(uintptr_t)15;
    void** sp = (void**)(top - 64);
    for (int i = 0; i < 6; i++) sp[i] = nullptr;
    sp[6] = (void*)&fiber_entry;
    sp[7] = nullptr;
    t.sp = sp;
}

static void release_wave(Wave& w, int wave_idx) {
    for (auto& t : threads)
        if (t.wave == wave_idx && t.st == WAIT_WAVE) t.st = READY;
    w.arrived = 0;
    w.op = 0;
}

// Block until every live lane of the wave has arrived with the same op; the last arriver
// runs `resolve` for the whole wave.
template <typename F>
static void wave_rendezvous(int op, F&& resolve) {
    Wave& w = waves[cur->wave];
    if (w.arrived == 0) w.op = op;
    else if (w.op != op) die("divergent wave-level operations (different ops pending in one wave)");
    w.arrived++;
    if (w.arrived == w.alive) {
        resolve(w);
        int me = cur->wave;
        Thread* self = cur;
        release_wave(w, me);
        self->st = READY;
        return;
    }
    cur->st = WAIT_WAVE;
    yield_to_sched();
}

void barrier() {
    block_arrived++;
    if (block_arrived == block_alive) {
        for (auto& t : threads)
            if (t.st == WAIT_BLOCK) t.st = READY;
        block_arrived = 0;
        return;
    }
    cur->st = WAIT_BLOCK;
    yield_to_sched();
}

float shfl(float v, int src, int width) {
    int l = cur->lane;
    Wave& w = waves[cur->wave];
    w.a[l] = v;
    int base = (l / width) * width;
    w.isrc[l] = base + (((src % width) + width) % width);
    wave_rendezvous(1, [](Wave& ww) {
        for (int i = 0; i < 64; i++) ww.b[i] = (i < ww.nlanes) ? ww.a[ww.isrc[i] < ww.nlanes ? ww.isrc[i] : i] : 0.f;
    });
    return w.b[l];
}

unsigned long long ballot(int pred) {
    int l = cur->lane;
    Wave& w = waves[cur->wave];
    w.isrc[l] = pred ? 1 : 0;
    wave_rendezvous(2, [](Wave& ww) {
        unsigned long long m = 0;
        for (int i = 0; i < ww.nlanes; i++)
            if (ww.isrc[i]) m |= 1ull << i;
        ww.mask = m;
    });
    return w.mask;
}

static void need_full_wave(Wave& w) {
    if (w.alive != 64 || w.nlanes != 64) die("MFMA issued with inactive lanes (EXEC is ignored by MFMA on hardware)");
}

f32x16 mfma_32x32x2f32(float a, float b, f32x16 c) {
    int l = cur->lane;
    Wave& w = waves[cur->wave];
    w.a[l] = a; w.b[l] = b; w.c16[l] = c;
    wave_rendezvous(10, [](Wave& ww) {
        need_full_wave(ww);
        for (int ln = 0; ln < 64; ln++) {
            int j = ln & 31;
            for (int r = 0; r < 16; r++) {
                int i = (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
                float acc = ww.c16[ln][r];
                for (int k = 0; k < 2; k++) acc = fmaf(ww.a[i + 32 * k], ww.b[j + 32 * k], acc);
                ww.d16[ln][r] = acc;
            }
        }
    });
    return w.d16[l];
}

f32x4 mfma_16x16x4f32(float a, float b, f32x4 c) {
    int l = cur->lane;
    Wave& w = waves[cur->wave];
    w.a[l] = a; w.b[l] = b;
    for (int r = 0; r < 4; r++) w.c16[l][r] = c[r];
    wave_rendezvous(11, [](Wave& ww) {
        need_full_wave(ww);
        for (int ln = 0; ln < 64; ln++) {
            int j = ln & 15;
            for (int r = 0; r < 4; r++) {
                int i = 4 * (ln >> 4) + r;
                float acc = ww.c16[ln][r];
                for (int k = 0; k < 4; k++) acc = fmaf(ww.a[i + 16 * k], ww.b[j + 16 * k], acc);
                ww.d16[ln][r] = acc;
            }
        }
    });
    f32x4 d;
    for (int r = 0; r < 4; r++) d[r] = w.d16[l][r];
    return d;
}

f32x16 mfma_32x32x16f16(f16x8 a, f16x8 b, f32x16 c) {
    int l = cur->lane;
    Wave& w = waves[cur->wave];
    w.ha[l] = a; w.hb[l] = b; w.c16[l] = c;
    wave_rendezvous(12, [](Wave& ww) {
        need_full_wave(ww);
        for (int ln = 0; ln < 64; ln++) {
            int j = ln & 31;
            for (int r = 0; r < 16; r++) {
                int i = (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
                float acc = ww.c16[ln][r];
                for (int kb = 0; kb < 2; kb++)
                    for (int e = 0; e < 8; e++) acc += (float)ww.ha[i + 32 * kb][e] * (float)ww.hb[j + 32 * kb][e];
                ww.d16[ln][r] = acc;
            }
        }
    });
    return w.d16[l];
}

f32x4 mfma_16x16x32f16(f16x8 a, f16x8 b, f32x4 c) {
    int l = cur->lane;
    Wave& w = waves[cur->wave];
    w.ha[l] = a; w.hb[l] = b;
    for (int r = 0; r < 4; r++) w.c16[l][r] = c[r];
    wave_rendezvous(13, [](Wave& ww) {
        need_full_wave(ww);
        for (int ln = 0; ln < 64; ln++) {
            int j = ln & 15;
            for (int r = 0; r < 4; r++) {
                int i = 4 * (ln >> 4) + r;
                float acc = ww.c16[ln][r];
                for (int kb = 0; kb < 4; kb++)
                    for (int e = 0; e < 8; e++) acc += (float)ww.ha[i + 16 * kb][e] * (float)ww.hb[j + 16 * kb][e];
                ww.d16[ln][r] = acc;
            }
        }
    });
    f32x4 d;
    for (int r = 0; r < 4; r++) d[r] = w.d16[l][r];
    return d;
}

static void run_block(dim3 block) {
    int n = (int)(block.x * block.y * block.z);
    int nw = (n + 63) / 64;
    if ((int)threads.size() < n) threads.resize(n);
    waves.assign(nw, Wave());
    for (int t = 0; t < n; t++) {
        Thread& th = threads[t];
        th.tid = uint3{(unsigned)(t % block.x), (unsigned)((t / block.x) % block.y), (unsigned)(t / (block.x * block.y))};
        th.lane = t & 63;
        th.wave = t >> 6;
        th.st = READY;
        init_fiber(th);
        waves[th.wave].nlanes++;
        waves[th.wave].alive++;
    }
    block_alive = n;
    block_arrived = 0;
    int done = 0;
    while (done < n) {
        bool progress = false;
        for (int t = 0; t < n; t++) {
            Thread& th = threads[t];
            if (th.st != READY) continue;
            progress = true;
            cur = &th;
            g_tid = th.tid;
            hipemu_switch(&sched_sp, th.sp);
            if (th.st == DONE) {
                done++;
                Wave& w = waves[th.wave];
                w.alive--;
                block_alive--;
                if (w.alive > 0 && w.arrived == w.alive)
                    die("a lane exited while the rest of its wave waits in a wave-level op");
                if (block_alive > 0 && block_arrived == block_alive) {
                    for (int u = 0; u < n; u++)
                        if (threads[u].st == WAIT_BLOCK) threads[u].st = READY;
                    block_arrived = 0;
                }
            }
        }
        if (!progress) die("deadlock: no runnable thread (mismatched __syncthreads or wave op)");
    }
    for (int t = (int)threads.size() - 1; t >= n; t--) threads[t].st = DONE;
}

void launch(const std::function<void()>& body, dim3 grid, dim3 block, size_t shmem) {
    if (cur != nullptr && sched_sp != nullptr && g_body != nullptr) die("nested kernel launch");
    if (block.x * block.y * block.z == 0 || block.x * block.y * block.z > 1024) die("bad block size");
    if (grid.x == 0 || grid.y == 0 || grid.z == 0) die("empty grid");
    dsmem.assign(shmem + 64, 0);
    g_body = &body;
    g_bdim = block;
    g_gdim = grid;
    for (unsigned z = 0; z < grid.z; z++)
        for (unsigned y = 0; y < grid.y; y++)
            for (unsigned x = 0; x < grid.x; x++) {
                g_bid = uint3{x, y, z};
                run_block(block);
            }
    g_body = nullptr;
    cur = nullptr;
}

}  // namespace hipemu

// ---- runtime API -------------------------------------------------------------------------
hipError_t hipMalloc(void** p, size_t n) {
    const size_t bytes = (n + 255) & ~(size_t)255;
    *p = aligned_alloc(256, bytes ? bytes : 256);
    // hipMalloc does not zero device memory: poison it (0xFF = NaN as fp32 / fp16, -1 as int) so that a kernel reading
    // something nobody wrote shows up in the CPU tier as it would -- randomly -- on the GPU
    if (*p) memset(*p, 0xFF, bytes ? bytes : 256);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipPeekAtLastError() { return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    memset(p, 0, sizeof(*p));
    snprintf(p->name, sizeof(p->name), "hipemu (host SIMT emulator)");
    snprintf(p->gcnArchName, sizeof(p->gcnArchName), "gfx950-emu");
    p->multiProcessorCount = 256;
    return hipSuccess;
}
struct ihipEvent_t { double t; };
static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
hipError_t hipEventCreate(hipEvent_t* e) { *e = new ihipEvent_t{0}; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = now_ms(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "hipemu error"; }
const char* hipGetErrorName(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipErrorEmu"; }
