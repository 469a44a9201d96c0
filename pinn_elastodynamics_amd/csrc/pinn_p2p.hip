// One-shot all-reduce(sum) of the step's fused buffer [gradient | loss sums] over IPC-mapped peer memory, with the TF1 Adam update in the
// same kernel (include/pinn_hip.h: pinn_p2p_*; SURVEY section 5 / 8e: the message is ~119 KB -- latency-bound, so every rank WRITES its
// buffer straight into a slot of every peer's receive buffer over its point-to-point xGMI link and sums the world's slots locally; no ring,
// no tree, no dependence on RCCL's small-message protocol).  One process per GPU; the receive buffers are exchanged once as hipIpcMemHandles.
//
//   receive buffer of rank r (device memory of r, fine-grained so that peer writes are visible inside a running kernel):
//       slots [2 parities][world][max_floats]      slot (q, s) <- rank s's buffer of the calls with (seq & 1) == q
//       flags [2 parities][world][P2P_BLOCKS]      flag (q, s, b) = seq once rank s's chunk b of that call has landed
//       ctrl  {abort, arrive, fail}                abort: written by PEERS (the call number a peer failed at); arrive / fail: this rank's own grid agreement
//   call number `seq` (1, 2, ...; the same on every rank -- a collective; calls of one comm are made on ONE stream):
//       block b of rank s:  copies chunk b of its buffer into slot (seq & 1, s) of EVERY rank (its own included) with 16-byte stores,
//                           __threadfence_system(), release-stores seq into flag (seq & 1, s, b) of every rank;
//                           then acquire-polls its OWN flags (seq & 1, 0..world-1, b) until all equal seq -- bounded by the device's constant-rate
//                           wall clock (pinn_p2p_set_timeout_ms; default 30 s);
//       GRID AGREEMENT (round 6): every block reports "arrived" (and "failed") in this rank's ctrl words and waits for all P2P_BLOCKS blocks, so
//                           the call succeeds or fails AS A WHOLE on a rank: a failed call never applies Adam to some chunks and not to others;
//       success:            sums the world's chunk b in rank order (the same order on every rank: the same bits everywhere), writes it back
//                           to the buffer and, for the first n_params entries, applies Adam;
//       failure:            the buffer is POISONED with NaN (the model classes' finite-gradient checks trip on it), Adam is skipped, the status word
//                           (pinned host memory) becomes PINN_ERR_COLLECTIVE, and the call number is written into every peer's abort word: a peer
//                           that arrives later -- or had already completed the call -- fails its current or next call at once instead of waiting.
//                           Failure is sticky: every later call on the comm poisons its buffer and returns immediately.
//   Two parities: a rank can be at most one call ahead of a peer (it cannot finish call k+1 before the peer has STARTED k+1, i.e. finished
//   reading the slots of call k), so the slots of call k+1 never overwrite what a peer still reads of call k.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/pinn_hip.h"

namespace {

constexpr int P2P_BLOCKS = 64;           // chunks of a call = workgroups of its kernel
constexpr int P2P_MAX_WORLD = 16;
constexpr double P2P_DEFAULT_TIMEOUT_MS = 30000.0;

struct Ctrl {                            // one per rank, behind its flags (fine-grained device memory of that rank)
    unsigned abort_seq;                  // written by PEERS (system-scope atomic max): call number a peer failed at; 0 = nobody has
    unsigned fail_seq;                   // this rank's blocks: call number some block of this rank failed at
    unsigned long long arrive;           // this rank's blocks: P2P_BLOCKS per completed call
};

struct PeerTable {
    float* slots[P2P_MAX_WORLD];         // base of rank r's slots  [2][world][max_floats]
    unsigned* flags[P2P_MAX_WORLD];      // base of rank r's flags  [2][world][P2P_BLOCKS]
    Ctrl* ctrl[P2P_MAX_WORLD];
};

struct AdamArgs {
    float* theta;                        // nullptr: no optimizer step
    float* m;
    float* v;
    long n_params;
    float lr_t, beta1, beta2, eps;
};

__device__ __forceinline__ void poison(float* buf, long lo, long hi) {
    const float nan = __builtin_nanf("");
    for (long i = lo + threadIdx.x; i < hi; i += 256) buf[i] = nan;
}

// vec16: buf and every slot are 16-byte aligned and the chunk bounds are multiples of 4 floats (the host checks): 16-byte pushes
__global__ __launch_bounds__(256) void p2p_allreduce_kernel(PeerTable peers, int rank, int world, long max_floats, float* buf, long n, unsigned seq,
                                                            AdamArgs adam, int* status, long long timeout_ticks, int vec16) {
    const int b = blockIdx.x, par = (int)(seq & 1u);
    long chunk = (n + P2P_BLOCKS - 1) / P2P_BLOCKS;
    chunk = (chunk + 3) & ~3L;
    const long lo = (long)b * chunk < n ? (long)b * chunk : n, hi = lo + chunk < n ? lo + chunk : n;
    Ctrl* my = peers.ctrl[rank];
    __shared__ int failed, dead_s;
    // ---- a comm that has failed stays failed: no push, no wait -- but every block still takes part in the grid agreement below (a block that
    // returned here would leave the others of its launch waiting for it).  Read ONCE per block: the words can change under a running launch.
    if (threadIdx.x == 0) {
        failed = 0;
        dead_s = (__hip_atomic_load(&my->abort_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != 0u ||
                  __hip_atomic_load(&my->fail_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0u) ? 1 : 0;
    }
    __syncthreads();
    const bool dead = dead_s != 0;
    const long long t0 = (long long)wall_clock64();
    if (dead) {
        if (threadIdx.x == 0) failed = 1;
    } else {
        // ---- push: my chunk b into slot (par, rank) of every rank
        for (int r = 0; r < world; ++r) {
            float* dst = peers.slots[r] + ((long)par * world + rank) * max_floats;
            if (vec16) {                     // (lo, hi are multiples of 4 here: chunk is, and the host requires n % 4 == 0)
                const float4* s4 = reinterpret_cast<const float4*>(buf);
                float4* d4 = reinterpret_cast<float4*>(dst);
                for (long i = (lo >> 2) + threadIdx.x; i < (hi >> 2); i += 256) d4[i] = s4[i];
            } else {
                for (long i = lo + threadIdx.x; i < hi; i += 256) dst[i] = buf[i];
            }
        }
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x < world) {
            unsigned* f = peers.flags[threadIdx.x] + ((long)par * world + rank) * P2P_BLOCKS + b;
            __hip_atomic_store(f, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        // ---- wait for chunk b of every rank (bounded by the wall clock; a peer's abort ends the wait at once)
        if (threadIdx.x < world) {
            const unsigned* f = peers.flags[rank] + ((long)par * world + threadIdx.x) * P2P_BLOCKS + b;
            while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
                if ((long long)wall_clock64() - t0 > timeout_ticks ||
                    __hip_atomic_load(&my->abort_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) { failed = 1; break; }
                __builtin_amdgcn_s_sleep(8);
            }
        }
    }
    __syncthreads();
    // ---- grid agreement: the call succeeds or fails as a whole on this rank
    if (threadIdx.x == 0) {
        if (failed) __hip_atomic_fetch_max(&my->fail_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&my->arrive, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long target = (unsigned long long)P2P_BLOCKS * seq;
        while (__hip_atomic_load(&my->arrive, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if ((long long)wall_clock64() - t0 > 2 * timeout_ticks + 100000000ll) { failed = 1; break; }      // (a block of this launch that never ran: cannot happen on an idle device)
            __builtin_amdgcn_s_sleep(2);
        }
        if (__hip_atomic_load(&my->fail_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0u ||
            __hip_atomic_load(&my->abort_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != 0u)
            failed = 1;
    }
    __syncthreads();
    if (failed) {
        poison(buf, lo, hi);
        if (b == 0) {
            if (threadIdx.x == 0) __hip_atomic_store(status, (int)PINN_ERR_COLLECTIVE, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            if (threadIdx.x < world && threadIdx.x != rank) __hip_atomic_fetch_max(&peers.ctrl[threadIdx.x]->abort_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    __threadfence_system();
    // ---- sum in rank order, write back, Adam
    const float* mine = peers.slots[rank] + (long)par * world * max_floats;
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
        float s = mine[i];
        for (int r = 1; r < world; ++r) s += mine[(long)r * max_floats + i];
        buf[i] = s;
        if (adam.theta != nullptr && i < adam.n_params) {
            const float mi = adam.beta1 * adam.m[i] + (1.0f - adam.beta1) * s;
            const float vi = adam.beta2 * adam.v[i] + (1.0f - adam.beta2) * s * s;
            adam.m[i] = mi;
            adam.v[i] = vi;
            adam.theta[i] -= adam.lr_t * mi / (__builtin_amdgcn_sqrtf(vi) + adam.eps);
        }
    }
}

// what travels with the IPC handle (include/pinn_hip.h: PINN_IPC_HANDLE_BYTES = 128): the memory kind and the physical device of the buffer
struct HandleBlob {
    hipIpcMemHandle_t ipc;               // 64 bytes
    unsigned char fine_grained;
    char pci[47];                        // hipDeviceGetPCIBusId of the owning device ("0000:05:00.0"): the same string in every process of the node
    unsigned char reserved[16];
};
static_assert(sizeof(HandleBlob) == PINN_IPC_HANDLE_BYTES, "handle blob layout");

int g_force_coarse = 0;                  // pinn_p2p_debug_force_coarse (tests: the refusal of coarse-grained buffers across devices)

}  // namespace

struct pinn_p2p_comm {
    int rank, world;
    long max_floats;
    void* base;                          // this rank's receive buffer (slots, then flags, then the ctrl words)
    size_t bytes;
    void* peer_base[P2P_MAX_WORLD];      // opened IPC mappings (own entry = base)
    PeerTable table;
    int* status;                         // pinned host word (mapped): 0 or PINN_ERR_COLLECTIVE, written by the kernel
    int* status_dev;                     // its device address
    unsigned seq;
    int connected;
    int fine_grained;
    int all_same_device;                 // every rank's buffer lives on this rank's physical device (the one-GPU tests)
    char pci[48];
    double timeout_ms;
    int wall_khz;
};

static size_t slots_bytes(int world, long max_floats) { return (size_t)2 * world * max_floats * sizeof(float); }
static size_t flags_bytes(int world) { return (size_t)2 * world * P2P_BLOCKS * sizeof(unsigned); }

extern "C" {

int pinn_p2p_debug_force_coarse(int enable) {
    const int old = g_force_coarse;
    g_force_coarse = enable ? 1 : 0;
    return old;
}

int pinn_p2p_create(int rank, int world, int64_t max_floats, pinn_p2p_comm** comm_out, unsigned char handle_out[PINN_IPC_HANDLE_BYTES]) {
    if (!comm_out || !handle_out) return PINN_ERR_NULL;
    if (world < 1 || world > P2P_MAX_WORLD || rank < 0 || rank >= world || max_floats < 1) return PINN_ERR_SIZE;
    pinn_p2p_comm* c = new pinn_p2p_comm();
    c->rank = rank;
    c->world = world;
    c->max_floats = (long)((max_floats + 63) / 64 * 64);
    c->bytes = slots_bytes(world, c->max_floats) + flags_bytes(world) + 256;
    c->seq = 0;
    c->connected = 0;
    c->all_same_device = 0;
    c->status = nullptr;
    c->timeout_ms = P2P_DEFAULT_TIMEOUT_MS;
    if (const char* env = getenv("PINN_P2P_TIMEOUT_MS")) {
        const double v = atof(env);
        if (v > 0.0) c->timeout_ms = v;
    }
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) { delete c; return (int)e; }
    c->wall_khz = 0;
    if (hipDeviceGetAttribute(&c->wall_khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || c->wall_khz <= 0) {
        (void)hipGetLastError();
        c->wall_khz = 100000;            // gfx950: 100 MHz
    }
    memset(c->pci, 0, sizeof(c->pci));
    if (hipDeviceGetPCIBusId(c->pci, (int)sizeof(c->pci) - 1, dev) != hipSuccess) { (void)hipGetLastError(); c->pci[0] = '?'; }
    // fine-grained device memory: a peer's stores become visible to a kernel that is already running here (coarse-grained memory is only
    // coherent at kernel boundaries).  If the runtime refuses the flag the buffer is plain device memory -- correct ONLY for ranks that share
    // one GPU (the tests); pinn_p2p_connect refuses it as soon as a peer lives on another device.
    c->fine_grained = g_force_coarse ? 0 : 1;
    e = g_force_coarse ? hipErrorNotSupported : hipExtMallocWithFlags(&c->base, c->bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        c->fine_grained = 0;
        e = hipMalloc(&c->base, c->bytes);
    }
    if (e != hipSuccess) { delete c; return (int)e; }
    if ((e = hipMemset(c->base, 0, c->bytes)) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess) { hipFree(c->base); delete c; return (int)e; }
    // (zeroed and complete BEFORE the handle leaves this function: a peer can only write here after it has opened the handle)
    // status word: pinned, mapped host memory -- the kernel writes it with a system-scope store, the host reads it without a copy
    if ((e = hipHostMalloc(reinterpret_cast<void**>(&c->status), 64, hipHostMallocMapped)) != hipSuccess) { hipFree(c->base); delete c; return (int)e; }
    *c->status = 0;
    if ((e = hipHostGetDevicePointer(reinterpret_cast<void**>(&c->status_dev), c->status, 0)) != hipSuccess) { hipHostFree(c->status); hipFree(c->base); delete c; return (int)e; }
    HandleBlob hb;
    memset(&hb, 0, sizeof(hb));
    if ((e = hipIpcGetMemHandle(&hb.ipc, c->base)) != hipSuccess) { hipHostFree(c->status); hipFree(c->base); delete c; return (int)e; }
    hb.fine_grained = (unsigned char)c->fine_grained;
    strncpy(hb.pci, c->pci, sizeof(hb.pci) - 1);
    memcpy(handle_out, &hb, sizeof(hb));
    for (int r = 0; r < P2P_MAX_WORLD; ++r) c->peer_base[r] = nullptr;
    *comm_out = c;
    return PINN_OK;
}

int pinn_p2p_connect(pinn_p2p_comm* c, const unsigned char* all_handles) {
    if (!c || !all_handles) return PINN_ERR_NULL;
    if (c->connected) return PINN_OK;
    // ---- refuse BEFORE mapping anything: a coarse-grained buffer (this rank's or a peer's) is only coherent at kernel boundaries, which is
    // enough when all ranks share one physical device (their kernels serialise through the same L2) and NOT across devices
    int same = 1;
    for (int r = 0; r < c->world; ++r) {
        HandleBlob hb;
        memcpy(&hb, all_handles + (size_t)r * PINN_IPC_HANDLE_BYTES, sizeof(hb));
        hb.pci[sizeof(hb.pci) - 1] = 0;
        const bool other_device = strcmp(hb.pci, c->pci) != 0;
        if (other_device) same = 0;
        if (r != c->rank && other_device && (!hb.fine_grained || !c->fine_grained)) return PINN_ERR_COLLECTIVE;
    }
    c->all_same_device = same;
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank) {
            c->peer_base[r] = c->base;
        } else {
            HandleBlob hb;
            memcpy(&hb, all_handles + (size_t)r * PINN_IPC_HANDLE_BYTES, sizeof(hb));
            const hipError_t e = hipIpcOpenMemHandle(&c->peer_base[r], hb.ipc, hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) {
                (void)hipGetLastError();      // (reported here: must not surface again behind the caller's next kernel launch)
                c->peer_base[r] = nullptr;
                return (int)e;
            }
        }
        char* p = static_cast<char*>(c->peer_base[r]);
        c->table.slots[r] = reinterpret_cast<float*>(p);
        c->table.flags[r] = reinterpret_cast<unsigned*>(p + slots_bytes(c->world, c->max_floats));
        c->table.ctrl[r] = reinterpret_cast<Ctrl*>(p + slots_bytes(c->world, c->max_floats) + flags_bytes(c->world));
    }
    c->connected = 1;
    return PINN_OK;
}

int pinn_p2p_set_timeout_ms(pinn_p2p_comm* c, double timeout_ms) {
    if (!c) return PINN_ERR_NULL;
    if (!(timeout_ms > 0.0)) return PINN_ERR_SIZE;
    c->timeout_ms = timeout_ms;
    return PINN_OK;
}

int pinn_p2p_allreduce(pinn_p2p_comm* c, float* buf, int64_t n, float* params_flat, const pinn_adam_state* adam, int64_t n_params, void* stream) {
    if (!c || !buf) return PINN_ERR_NULL;
    if (!c->connected) return PINN_ERR_COLLECTIVE;
    if (n < 1 || n > c->max_floats) return PINN_ERR_SIZE;
    AdamArgs a = {nullptr, nullptr, nullptr, 0, 0.f, 0.f, 0.f, 0.f};
    if (adam) {
        if (!params_flat || !adam->m || !adam->v) return PINN_ERR_NULL;
        if (n_params < 1 || n_params > n || adam->step < 1) return PINN_ERR_SIZE;
        const double b1t = __builtin_pow(adam->beta1, (double)adam->step), b2t = __builtin_pow(adam->beta2, (double)adam->step);      // (pinn_adam_step's expression)
        const double lr_t = adam->lr * __builtin_sqrt(1.0 - b2t) / (1.0 - b1t);
        a = AdamArgs{params_flat, adam->m, adam->v, (long)n_params, (float)lr_t, (float)adam->beta1, (float)adam->beta2, (float)adam->eps};
    }
    ++c->seq;
    // 16-byte pushes where the caller's buffer allows them: aligned base and a length that is a multiple of 4 floats (the kernel reads the
    // buffer in float4 units up to n); otherwise 4-byte pushes
    const int vec16 = ((reinterpret_cast<uintptr_t>(buf) & 15u) == 0 && (n & 3) == 0) ? 1 : 0;
    const long long ticks = (long long)(c->timeout_ms * (double)c->wall_khz);
    hipLaunchKernelGGL(p2p_allreduce_kernel, dim3(P2P_BLOCKS), dim3(256), 0, (hipStream_t)stream, c->table, c->rank, c->world, c->max_floats, buf, (long)n,
                       c->seq, a, c->status_dev, ticks, vec16);
    return (int)hipGetLastError();
}

int pinn_p2p_status(pinn_p2p_comm* c, int* fine_grained_out) {
    if (!c || !c->connected) return PINN_ERR_COLLECTIVE;
    if (fine_grained_out) *fine_grained_out = c->fine_grained;
    const hipError_t e = hipDeviceSynchronize();      // (for tests / the end of a run; pinn_p2p_peek_status does not synchronise)
    if (e != hipSuccess) return (int)e;
    return __atomic_load_n(c->status, __ATOMIC_ACQUIRE);
}

int pinn_p2p_peek_status(pinn_p2p_comm* c) {
    if (!c || !c->connected) return PINN_ERR_COLLECTIVE;
    return __atomic_load_n(c->status, __ATOMIC_ACQUIRE);
}

int pinn_p2p_destroy(pinn_p2p_comm* c) {
    if (!c) return PINN_OK;
    (void)hipDeviceSynchronize();
    if (c->connected)
        for (int r = 0; r < c->world; ++r)
            if (r != c->rank && c->peer_base[r]) (void)hipIpcCloseMemHandle(c->peer_base[r]);
    if (c->status) (void)hipHostFree(c->status);
    (void)hipFree(c->base);
    delete c;
    return PINN_OK;
}

}  // extern "C"
