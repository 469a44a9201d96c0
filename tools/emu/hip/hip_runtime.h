// Host-side SIMT emulator standing in for <hip/hip_runtime.h>.  DEVELOPMENT/TEST TOOLING ONLY.
//
// Purpose: there is no GPU in the build container, so the product's device code
// (pinn_elastodynamics_amd/csrc/*.hpp, *.hip -- unmodified, no #ifdefs) is also compiled
// for x86 against this header and executed lane-by-lane on fibers, which lets the index
// math of the MFMA fragment layouts, spill panels and launch geometry be checked against
// the CPU oracle before any GPU minute is spent.  It is never part of the shipped library:
// only tests/ build it (tests/test_emulated_kernels.py) and nothing in the package imports it.
//
// Model: one workgroup at a time; every thread of the workgroup is a fiber; wave-collective
// builtins (MFMA, shuffles) and __syncthreads() are rendezvous points.  Lanes of a wave must
// reach collectives in the same order (wave-uniform control flow), as on the hardware.
// MFMA operand/result lane maps follow /opt/skills/guides/cdna_hip_programming.md section 3
// (16x16x32: A[i=l&15][k=8*(l>>4)+j], B[k=8*(l>>4)+j][n=l&15], D[row=4*(l>>4)+r][col=l&15]).
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>
#include <algorithm>

#define PINN_SIMT_EMULATOR 1      // (the x86 test build of the kernel sources; never defined for gfx950)
#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef int hipError_t;
typedef void* hipStream_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorLaunchFailure = 719 };
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "emulated"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
typedef void* hipEvent_t;
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
enum { hipMemcpyDeviceToDevice = 3, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2 };

namespace emu {
struct Lane {
    void* sp = nullptr;          // saved stack pointer of the fiber
    char* stack = nullptr;
    dim3 tid;
    int state = 0;               // 0 runnable, 1 at wave sync, 2 at block sync, 3 done, 4 polling (runnable again on the next round)
    const void* xchg[3] = {nullptr, nullptr, nullptr};
};
struct Block {
    std::vector<Lane> lanes;
    dim3 bid, bdim, gdim;
    std::function<void()> body;
    void* sched_sp = nullptr;
    int cur = 0;
};
extern Block* g_blk;
inline Lane& cur_lane() { return g_blk->lanes[g_blk->cur]; }
inline Lane& wave_lane(int l) { return g_blk->lanes[(g_blk->cur & ~63) + l]; }
inline int lane_id() { return g_blk->cur & 63; }
void yield_to_scheduler(int state);
inline void wave_sync() { yield_to_scheduler(1); }
inline void block_sync() { yield_to_scheduler(2); }
inline void spin_yield() { yield_to_scheduler(4); }      // a lane polling a flag another wave will set: let the other waves run
void run_grid(dim3 grid, dim3 block, const std::function<void()>& body);
extern long g_mfma_count;
}  // namespace emu

#define threadIdx (emu::cur_lane().tid)
#define blockIdx (emu::g_blk->bid)
#define blockDim (emu::g_blk->bdim)
#define gridDim (emu::g_blk->gdim)

namespace emu {
template <class K, class... Args>
inline void launch(K kernel, dim3 grid, dim3 block, Args... args) {
    run_grid(grid, block, [=]() { kernel(args...); });
}
}  // namespace emu
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    emu::launch(kernel, dim3(grid), dim3(block), __VA_ARGS__)

inline void __syncthreads() { emu::block_sync(); }
inline void __builtin_amdgcn_s_barrier() { emu::block_sync(); }

// ---- wave collectives -------------------------------------------------------------------------
template <class T>
inline T __shfl_xor(T v, int mask, int width = 64) {
    (void)width;
    emu::cur_lane().xchg[0] = &v;
    emu::wave_sync();
    T r = *static_cast<const T*>(emu::wave_lane(emu::lane_id() ^ mask).xchg[0]);
    emu::wave_sync();
    return r;
}
template <class T>
inline T __shfl(T v, int src, int width = 64) {
    (void)width;
    emu::cur_lane().xchg[0] = &v;
    emu::wave_sync();
    T r = *static_cast<const T*>(emu::wave_lane(src & 63).xchg[0]);
    emu::wave_sync();
    return r;
}
template <class T>
inline T __shfl_down(T v, unsigned d, int width = 64) {
    (void)width;
    emu::cur_lane().xchg[0] = &v;
    emu::wave_sync();
    int src = emu::lane_id() + (int)d;
    T r = src < 64 ? *static_cast<const T*>(emu::wave_lane(src).xchg[0]) : v;
    emu::wave_sync();
    return r;
}
inline int __builtin_amdgcn_readfirstlane(int v) { return __shfl(v, 0); }

typedef float emu_f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 emu_f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 emu_bf16x8 __attribute__((ext_vector_type(8)));

template <class V8>
inline emu_f32x4 emu_mfma_16x16x32(V8 a, V8 b, emu_f32x4 c) {
    emu::Lane& me = emu::cur_lane();
    me.xchg[1] = &a;
    me.xchg[2] = &b;
    emu::wave_sync();
    const int l = emu::lane_id(), col = l & 15, q = l >> 4;
    emu_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * q + r;
        float acc = c[r];
        for (int k = 0; k < 32; ++k) {
            const V8& av = *static_cast<const V8*>(emu::wave_lane(row + 16 * (k >> 3)).xchg[1]);
            const V8& bv = *static_cast<const V8*>(emu::wave_lane(col + 16 * (k >> 3)).xchg[2]);
            acc = fmaf((float)av[k & 7], (float)bv[k & 7], acc);
        }
        d[r] = acc;
    }
    emu::wave_sync();
    if (l == 0) ++emu::g_mfma_count;
    return d;
}
inline emu_f32x4 __builtin_amdgcn_mfma_f32_16x16x32_f16(emu_f16x8 a, emu_f16x8 b, emu_f32x4 c, int, int, int) {
    return emu_mfma_16x16x32(a, b, c);
}
inline emu_f32x4 __builtin_amdgcn_mfma_f32_16x16x32_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x4 c, int, int, int) {
    return emu_mfma_16x16x32(a, b, c);
}
// f32-input MFMA 16x16x4: A[i=l&15][k=l>>4], B[k=l>>4][n=l&15]
inline emu_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, emu_f32x4 c, int, int, int) {
    emu::Lane& me = emu::cur_lane();
    me.xchg[1] = &a;
    me.xchg[2] = &b;
    emu::wave_sync();
    const int l = emu::lane_id(), col = l & 15, q = l >> 4;
    emu_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        float acc = c[r];
        for (int k = 0; k < 4; ++k)
            acc = fmaf(*static_cast<const float*>(emu::wave_lane(4 * q + r + 16 * k).xchg[1]),
                       *static_cast<const float*>(emu::wave_lane(col + 16 * k).xchg[2]), acc);
        d[r] = acc;
    }
    emu::wave_sync();
    return d;
}

// LDS transpose read (gfx950 ds_read_b64_tr_b16), semantics measured with tools/probes/tr_read_probe.hip:
// within each 16-lane group, out[lane c][e] = in[lane 4e + (c>>2)][c & 3], in[i][.] = the four 16-bit values at lane i's address.
typedef short emu_v4i16 __attribute__((ext_vector_type(4)));
inline emu_v4i16 __builtin_amdgcn_ds_read_tr16_b64_v4i16(__attribute__((address_space(3))) emu_v4i16* p) {
    const short* mine = reinterpret_cast<const short*>(reinterpret_cast<uintptr_t>(p));
    emu::cur_lane().xchg[0] = mine;
    emu::wave_sync();
    const int l = emu::lane_id(), g = l & ~15, c = l & 15;
    emu_v4i16 r;
    for (int e = 0; e < 4; ++e) r[e] = static_cast<const short*>(emu::wave_lane(g + 4 * e + (c >> 2)).xchg[0])[c & 3];
    emu::wave_sync();
    return r;
}

// buffer descriptors (raw, byte-addressed): base + per-lane voffset + wave-uniform soffset
struct emu_buffer_rsrc { char* base; unsigned bytes; };
#define __amdgpu_buffer_rsrc_t emu_buffer_rsrc
typedef unsigned int emu_u32x4 __attribute__((ext_vector_type(4)));
inline emu_buffer_rsrc __builtin_amdgcn_make_buffer_rsrc(void* p, short, int num_bytes, int) { return emu_buffer_rsrc{static_cast<char*>(p), (unsigned)num_bytes}; }
inline emu_u32x4 __builtin_amdgcn_raw_buffer_load_b128(emu_buffer_rsrc r, unsigned voff, unsigned soff, int) {
    if (voff + soff + 16u > r.bytes) { fprintf(stderr, "emu: buffer load out of range (%u + %u > %u)\n", voff, soff, r.bytes); abort(); }
    emu_u32x4 v;
    memcpy(&v, r.base + voff + soff, 16);
    return v;
}
typedef unsigned int emu_u32x2 __attribute__((ext_vector_type(2)));
inline emu_u32x2 __builtin_amdgcn_raw_buffer_load_b64(emu_buffer_rsrc r, unsigned voff, unsigned soff, int) {
    if (voff + soff + 8u > r.bytes) { fprintf(stderr, "emu: buffer load out of range (%u + %u > %u)\n", voff, soff, r.bytes); abort(); }
    emu_u32x2 v;
    memcpy(&v, r.base + voff + soff, 8);
    return v;
}
inline void __builtin_amdgcn_raw_buffer_store_b128(emu_u32x4 v, emu_buffer_rsrc r, unsigned voff, unsigned soff, int) {
    if (voff + soff + 16u > r.bytes) { fprintf(stderr, "emu: buffer store out of range (%u + %u > %u)\n", voff, soff, r.bytes); abort(); }
    memcpy(r.base + voff + soff, &v, 16);
}

// LDS-DMA (buffer_load ... lds), semantics measured with tools/probes/lds_dma_probe.hip:
//   LDS dst = ldsptr (wave-uniform) + inst_offset + lane*size ;  src = base + voffset + soffset + inst_offset
// (executed synchronously here; on hardware it completes asynchronously under vmcnt)
inline void __builtin_amdgcn_raw_ptr_buffer_load_lds(emu_buffer_rsrc r, __attribute__((address_space(3))) void* ldsptr, unsigned size,
                                                     unsigned voff, unsigned soff, int inst_off, int) {
    char* mine = reinterpret_cast<char*>(reinterpret_cast<uintptr_t>(ldsptr));
    emu::cur_lane().xchg[0] = mine;
    emu::wave_sync();
    char* base = static_cast<char*>(const_cast<void*>(emu::wave_lane(0).xchg[0]));
    emu::wave_sync();
    if (voff + soff + inst_off + size > r.bytes) { fprintf(stderr, "emu: lds-dma out of range\n"); abort(); }
    memcpy(base + inst_off + emu::lane_id() * size, r.base + voff + soff + inst_off, size);
}
typedef unsigned int emu_u32x2 __attribute__((ext_vector_type(2)));
inline void __builtin_amdgcn_raw_buffer_store_b64(emu_u32x2 v, emu_buffer_rsrc r, unsigned voff, unsigned soff, int) {
    if (voff + soff + 8u > r.bytes) { fprintf(stderr, "emu: buffer store out of range (%u + %u > %u)\n", voff, soff, r.bytes); abort(); }
    memcpy(r.base + voff + soff, &v, 8);
}

inline unsigned __builtin_amdgcn_raw_buffer_load_b32(emu_buffer_rsrc r, unsigned voff, unsigned soff, int) {
    if (voff + soff + 4u > r.bytes) { fprintf(stderr, "emu: buffer load out of range (%u + %u > %u)\n", voff, soff, r.bytes); abort(); }
    unsigned v;
    memcpy(&v, r.base + voff + soff, 4);
    return v;
}
inline void __builtin_amdgcn_raw_buffer_store_b32(unsigned v, emu_buffer_rsrc r, unsigned voff, unsigned soff, int) {
    if (voff + soff + 4u > r.bytes) { fprintf(stderr, "emu: buffer store out of range (%u + %u > %u)\n", voff, soff, r.bytes); abort(); }
    memcpy(r.base + voff + soff, &v, 4);
}

inline unsigned long long emu_cycle_counter() { static unsigned long long t = 0; return t += 100; }
#define __builtin_readcyclecounter emu_cycle_counter

// ---- scalar math builtins ---------------------------------------------------------------------
inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
inline float __builtin_amdgcn_sqrtf(float x) { return sqrtf(x); }
inline float __builtin_amdgcn_rsqf(float x) { return 1.0f / sqrtf(x); }
inline void __builtin_amdgcn_s_setprio(int) {}
inline void __builtin_amdgcn_sched_barrier(int) {}
inline void __builtin_amdgcn_s_sleep(int) { emu::spin_yield(); }
inline void __builtin_amdgcn_sched_group_barrier(int, int, int) {}

template <class T>
inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T>
inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <class T>
inline T atomicMax(T* p, T v) { T o = *p; *p = o > v ? o : v; return o; }
