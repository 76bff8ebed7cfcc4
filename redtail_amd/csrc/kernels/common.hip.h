// Shared helpers for the gfx950 kernels of the Stereo DNN hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kWave = 64;   // CDNA wavefront

// register budget of a kernel as waves per SIMD (the test emulator defines this away)
#ifndef RT_WAVES_PER_EU
#define RT_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n, n)))
#endif

__host__ __device__ static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
__host__ __device__ static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// ELU(alpha = 1), NaN-propagating like cudnnActivationForward(CUDNN_ACTIVATION_ELU, PROPAGATE_NAN)
// (reference lib/elu_plugin.cpp:93): x > 0 ? x : exp(x) - 1  (TF computes exp(x) - 1, not expm1).
__device__ static __forceinline__ float elu1(float x) { return x > 0.f ? x : expf(x) - 1.f; }
__device__ static __forceinline__ float sigmoid1(float x) { return 1.f / (1.f + expf(-x)); }

template <int ACT>
__device__ static __forceinline__ float apply_act(float v) {
    if (ACT == 1) return elu1(v);
    if (ACT == 2) return sigmoid1(v);
    return v;
}
// Fused-epilogue variants: exp through the hardware v_exp_f32 (2 instructions instead of ~25).  |error| of
// exp(x) is <= ~1e-7 * max(1, |x|) for x <= 0, far inside the 1e-4 .. 1e-5 tolerances the reference applies
// to convolution outputs (tests_main.cpp:419-877); the stand-alone ELU plugin keeps the accurate expf.
__device__ static __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
// ELU of the fused epilogues: compare and select, NaN-propagating like the reference's cudnnActivationForward(ELU, PROPAGATE_NAN)
// (lib/elu_plugin.cpp:93).  Round 6 measured the one-instruction form med3(v, exp(v) - 1, 0) -- exp(v) - 1 >= v everywhere, so the median
// IS the ELU -- at +3 % on C3 and +2 % on C5, and dropped it: v_med3_f32 returns the minimum of the other operands for a NaN, i.e. 0, and
// an out-of-range input (the fp16 split's |x| >= 65504 -> inf - inf) would no longer surface as NaN in the result
// (tests/test_net_parity.py::test_debug_mode_reports_the_fp16_split_domain).
__device__ static __forceinline__ float elu_fast(float v) { return v > 0.f ? v : fast_exp(v) - 1.f; }
__device__ static __forceinline__ float apply_act_fast(float v, int act) {
    if (act == 1) return elu_fast(v);
    if (act == 2) return __builtin_amdgcn_rcpf(1.f + fast_exp(-v));      // v_rcp_f32: 1 ulp
    return v;
}
__device__ static __forceinline__ float apply_act_rt(float v, int act) {
    return act == 1 ? elu1(v) : (act == 2 ? sigmoid1(v) : v);
}

// ---- buffer addressing ------------------------------------------------------------------------------------
// On gfx950 the fp32 MFMA shares the SIMD's issue slot with every other VALU instruction (measured,
// tools/micro/mfma_valu.hip: the two do not overlap), so per-element 64-bit address arithmetic and bounds
// selects are paid in MFMA time.  Raw buffer instructions take the address as
//   base (4 SGPRs) + soffset (SGPR: plane / channel offset, scalar ALU) + voffset (VGPR: in-plane offset),
// and the hardware range check turns out-of-image lanes into zeros / dropped stores: voffset = kBufOOB.
// Valid voffset + soffset stay below 2^31 (checked when a plan is created).
typedef __amdgpu_buffer_rsrc_t buf_rsrc;
constexpr unsigned kBufOOB = 0x80000000u;     // voffset of a lane that must read 0 / must not store
constexpr unsigned kBufSpan = 0x80000000u;    // num_records of every resource
__device__ static __forceinline__ buf_rsrc make_buf(const void* base, bool valid = true) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, valid ? kBufSpan : 0u, 0x00020000);
}
__device__ static __forceinline__ float buf_load(buf_rsrc r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ static __forceinline__ f32x4 buf_load4(buf_rsrc r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
// 8-byte forms: the address only has to be 4-byte aligned (odd image widths)
__device__ static __forceinline__ f32x2_t buf_load2(buf_rsrc r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x2_t, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ static __forceinline__ void buf_store2(f32x2_t v, buf_rsrc r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v), r, voff, soff, 0);
}
// 16-byte store.  The plane / channel offset is ADDED TO THE VECTOR OFFSET on purpose.  With a register in the soffset field the
// compiler's hazard recognizer assumes the hardware protects the store data (no wait state before the data registers are
// written again); on gfx950 it does not always: measured in round 2 (conv_rbs.hip.h, reproducible on every launch)
//     buffer_store_dwordx4 v[0:3], v116, s[20:23], s3 offen ;  v_pk_add_f32 v[0:1], ...      <- next instruction
// stores the NEW value of v1 in lanes 12-15 / 28-31 of each half wave.  With an immediate soffset the recognizer inserts the
// wait state of the documented ">64-bit store data followed by a write of the data registers" hazard itself.  Offsets stay
// below 2^31 and kBufOOB + soff is still out of range, so dropped lanes stay dropped.
__device__ static __forceinline__ void buf_store4(f32x4 v, buf_rsrc r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), r, voff + soff, 0u, 0);
}
__device__ static __forceinline__ void buf_store(float v, buf_rsrc r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}

// LDS destination of a direct global -> LDS load (buffer_load ... lds): wave-uniform base, lane l lands at + size*l
#ifndef RT_LDS_PTR
#define RT_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#endif

// ---- hand-placed waits around LDS-DMA (conv_f16dw.hip.h) -------------------------------------------------------------------------------
// __syncthreads() drains the vector-memory counter whenever an LDS-DMA is pending -- and with it every global STORE the wave has in flight
// (gfx9 counts stores on vmcnt): a kernel that stores its results in the middle of a phase would wait for their acknowledgement at the
// next barrier.  These two let it wait for its DMA pieces where that is cheap and cross the barrier with the stores still under way.
// wait_vmem_but<N>: all but the N newest vector-memory operations are complete (loads return in order among themselves).
// pack_f16: two fp32 values rounded to nearest even into one register (v_cvt_pk_f16_f32, new on gfx950: one instruction instead of three).
#ifdef HIPEMU
__device__ static __forceinline__ void wait_vmem() {}
template <int N> __device__ static __forceinline__ void wait_vmem_but() {}
static inline void sload2_i32(const void* base, int off0, int off1, int& a, int& b) {
    a = *reinterpret_cast<const int*>(static_cast<const char*>(base) + off0);
    b = *reinterpret_cast<const int*>(static_cast<const char*>(base) + off1);
}
static inline void sload2_i64(const void* base, int off0, int off1, long long& a, long long& b) {
    a = *reinterpret_cast<const long long*>(static_cast<const char*>(base) + off0);
    b = *reinterpret_cast<const long long*>(static_cast<const char*>(base) + off1);
}
__device__ static __forceinline__ void lds_barrier() { __syncthreads(); }
__device__ static __forceinline__ void wave_lds_sync() { (void)__shfl(0.f, 0); }      // (a rendezvous of the wave's fibers)
__device__ static __forceinline__ unsigned pack_f16(float lo, float hi) {
    return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)lo) | ((unsigned)__builtin_bit_cast(unsigned short, (_Float16)hi) << 16);
}
#else
// Two dwords / two qwords of a READ-ONLY table by scalar loads, whatever the compiler thinks of the surrounding code: inside a loop that
// stores it will not use s_load on its own (it cannot prove the table unclobbered) and falls back to vector loads -- a vmcnt wait in the
// middle of a counted LDS-DMA pipeline (deconv_f16pw_kernel).  base: wave-uniform pointer; byte offsets: uniform, multiples of 4 / 8.
__device__ static __forceinline__ void sload2_i32(const void* base, int off0, int off1, int& a, int& b) {
    asm volatile("s_load_dword %0, %2, %3\n\ts_load_dword %1, %2, %4\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b) : "s"(base), "s"(off0), "s"(off1) : "memory");
}
__device__ static __forceinline__ void sload2_i64(const void* base, int off0, int off1, long long& a, long long& b) {
    asm volatile("s_load_dwordx2 %0, %2, %3\n\ts_load_dwordx2 %1, %2, %4\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b) : "s"(base), "s"(off0), "s"(off1) : "memory");
}
__device__ static __forceinline__ void wait_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <int N> __device__ static __forceinline__ void wait_vmem_but() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ static __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// LDS written by some lanes of a wave and read by others of the SAME wave: its LDS operations complete in order, so only the compiler
// has to be kept from moving them across this point
__device__ static __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ static __forceinline__ unsigned pack_f16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
#endif

// ---- storage-type generic element access ---------------------------------------------------------------------
// Activations are stored as fp32 or (TensorRT "half2 mode": setHalf2Mode + fp16 weights, sample_app/main.cpp:256-262)
// as fp16; arithmetic is fp32 either way (what the reference's fp16 correlation kernel does too,
// lib/kernels.cu:219-221).  Offsets are BYTES, ES = element size; the 2-element forms need 2*ES alignment, which
// the executor guarantees through even row pitches.
template <typename T> struct Io;
template <> struct Io<float> {
    static constexpr unsigned ES = 4;
    __device__ static __forceinline__ float load(buf_rsrc r, unsigned vo, unsigned so) { return buf_load(r, vo, so); }
    __device__ static __forceinline__ f32x2_t load2(buf_rsrc r, unsigned vo, unsigned so) { return buf_load2(r, vo, so); }
    __device__ static __forceinline__ void store(float v, buf_rsrc r, unsigned vo, unsigned so) { buf_store(v, r, vo, so); }
    __device__ static __forceinline__ void store2(f32x2_t v, buf_rsrc r, unsigned vo, unsigned so) { buf_store2(v, r, vo, so); }
};
template <> struct Io<_Float16> {
    static constexpr unsigned ES = 2;
    __device__ static __forceinline__ float load(buf_rsrc r, unsigned vo, unsigned so) {
        const unsigned short u = __builtin_amdgcn_raw_buffer_load_b16(r, vo, so, 0);
        return (float)__builtin_bit_cast(_Float16, u);
    }
    __device__ static __forceinline__ f32x2_t load2(buf_rsrc r, unsigned vo, unsigned so) {
        const unsigned u = __builtin_amdgcn_raw_buffer_load_b32(r, vo, so, 0);
        return f32x2_t{(float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu)),
                       (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16))};
    }
    __device__ static __forceinline__ void store(float v, buf_rsrc r, unsigned vo, unsigned so) {
        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (_Float16)v), r, vo, so, 0);
    }
    __device__ static __forceinline__ void store2(f32x2_t v, buf_rsrc r, unsigned vo, unsigned so) {
        const unsigned a = __builtin_bit_cast(unsigned short, (_Float16)v[0]), b = __builtin_bit_cast(unsigned short, (_Float16)v[1]);
        __builtin_amdgcn_raw_buffer_store_b32(a | (b << 16), r, vo, so, 0);
    }
};
// base + n elements of size ES (tensor pointers travel as float* whatever the storage type)
__device__ static __forceinline__ const char* elem_ptr(const void* base, int64_t elems, unsigned es) {
    return static_cast<const char*>(base) + elems * (int64_t)es;
}

}  // namespace rt
