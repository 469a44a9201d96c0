// One-shot all-reduce(sum) of the step's fused buffer [gradient | loss sums] over IPC-mapped peer memory, with the TF1 Adam update in the
// same kernel (include/pinn_hip.h: pinn_p2p_*; SURVEY section 5 / 8e: the message is ~119 KB -- latency-bound, so every rank WRITES its
// buffer straight into a slot of every peer's receive buffer over its point-to-point xGMI link and sums the world's slots locally; no ring,
// no tree, no dependence on RCCL's small-message protocol).  One process per GPU; the receive buffers are exchanged once as hipIpcMemHandles.
//
//   receive buffer of rank r (device memory of r, fine-grained so that peer writes are visible inside a running kernel):
//       slots [2 parities][world][max_floats]      slot (q, s) <- rank s's buffer of the calls with (seq & 1) == q
//       flags [2 parities][world][P2P_BLOCKS]      flag (q, s, b) = seq once rank s's chunk b of that call has landed
//   call number `seq` (1, 2, ...; the same on every rank -- a collective):
//       block b of rank s:  copies chunk b of its buffer into slot (seq & 1, s) of EVERY rank (its own included), __threadfence_system(),
//                           release-stores seq into flag (seq & 1, s, b) of every rank;
//                           then acquire-polls its OWN flags (seq & 1, 0..world-1, b) until all equal seq, sums the world's chunk b in rank
//                           order (the same order on every rank: the results are the same bits everywhere), writes it back to the buffer and,
//                           for the first n_params entries, applies Adam.
//   Two parities: a rank can be at most one call ahead of a peer (it cannot finish call k+1 before the peer has STARTED k+1, i.e. finished
//   reading the slots of call k), so the slots of call k+1 never overwrite what a peer still reads of call k.
//   The polls are bounded (~2 s): a rank that never arrives makes the call fail with PINN_ERR_COLLECTIVE in the status word instead of hanging the GPU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/pinn_hip.h"

namespace {

constexpr int P2P_BLOCKS = 64;           // chunks of a call = workgroups of its kernel
constexpr int P2P_MAX_WORLD = 16;
constexpr long P2P_POLL_LIMIT = 4000000; // bounded wait: ~2 s of system-scope polls

struct PeerTable {
    float* slots[P2P_MAX_WORLD];         // base of rank r's slots  [2][world][max_floats]
    unsigned* flags[P2P_MAX_WORLD];      // base of rank r's flags  [2][world][P2P_BLOCKS]
};

struct AdamArgs {
    float* theta;                        // nullptr: no optimizer step
    float* m;
    float* v;
    long n_params;
    float lr_t, beta1, beta2, eps;
};

__global__ __launch_bounds__(256) void p2p_allreduce_kernel(PeerTable peers, int rank, int world, long max_floats, float* buf, long n, unsigned seq,
                                                            AdamArgs adam, int* status) {
    const int b = blockIdx.x, par = (int)(seq & 1u);
    const long chunk = (n + P2P_BLOCKS - 1) / P2P_BLOCKS, lo = (long)b * chunk, hi = lo + chunk < n ? lo + chunk : n;
    // ---- push: my chunk b into slot (par, rank) of every rank
    for (int r = 0; r < world; ++r) {
        float* dst = peers.slots[r] + ((long)par * world + rank) * max_floats;
        for (long i = lo + threadIdx.x; i < hi; i += 256) dst[i] = buf[i];
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x < world) {
        unsigned* f = peers.flags[threadIdx.x] + ((long)par * world + rank) * P2P_BLOCKS + b;
        __hip_atomic_store(f, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // ---- wait for chunk b of every rank (bounded)
    __shared__ int failed;
    if (threadIdx.x == 0) failed = 0;
    __syncthreads();
    if (threadIdx.x < world) {
        const unsigned* f = peers.flags[rank] + ((long)par * world + threadIdx.x) * P2P_BLOCKS + b;
        long polls = 0;
        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            if (++polls > P2P_POLL_LIMIT) { failed = 1; break; }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
    if (failed) {
        if (threadIdx.x == 0) *status = PINN_ERR_COLLECTIVE;
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

}  // namespace

struct pinn_p2p_comm {
    int rank, world;
    long max_floats;
    void* base;                          // this rank's receive buffer (slots, then flags, then the status word)
    size_t bytes;
    void* peer_base[P2P_MAX_WORLD];      // opened IPC mappings (own entry = base)
    PeerTable table;
    int* status;                         // device word: 0 or PINN_ERR_COLLECTIVE
    unsigned seq;
    int connected;
    int fine_grained;
};

static size_t slots_bytes(int world, long max_floats) { return (size_t)2 * world * max_floats * sizeof(float); }
static size_t flags_bytes(int world) { return (size_t)2 * world * P2P_BLOCKS * sizeof(unsigned); }

extern "C" {

int pinn_p2p_create(int rank, int world, int64_t max_floats, pinn_p2p_comm** comm_out, unsigned char handle_out[PINN_IPC_HANDLE_BYTES]) {
    if (!comm_out || !handle_out) return PINN_ERR_NULL;
    if (world < 1 || world > P2P_MAX_WORLD || rank < 0 || rank >= world || max_floats < 1) return PINN_ERR_SIZE;
    static_assert(sizeof(hipIpcMemHandle_t) <= PINN_IPC_HANDLE_BYTES, "IPC handle size");
    pinn_p2p_comm* c = new pinn_p2p_comm();
    c->rank = rank;
    c->world = world;
    c->max_floats = (long)((max_floats + 63) / 64 * 64);
    c->bytes = slots_bytes(world, c->max_floats) + flags_bytes(world) + 256;
    c->seq = 0;
    c->connected = 0;
    // fine-grained device memory: a peer's stores become visible to a kernel that is already running here (coarse-grained memory is only
    // coherent at kernel boundaries).  If the runtime refuses the flag the buffer is plain device memory -- still correct for ranks that share
    // one GPU (the tests), and reported by pinn_p2p_status.
    c->fine_grained = 1;
    hipError_t e = hipExtMallocWithFlags(&c->base, c->bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        c->fine_grained = 0;
        e = hipMalloc(&c->base, c->bytes);
    }
    if (e != hipSuccess) { delete c; return (int)e; }
    if ((e = hipMemset(c->base, 0, c->bytes)) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess) { hipFree(c->base); delete c; return (int)e; }
    // (zeroed and complete BEFORE the handle leaves this function: a peer can only write here after it has opened the handle)
    hipIpcMemHandle_t h;
    if ((e = hipIpcGetMemHandle(&h, c->base)) != hipSuccess) { hipFree(c->base); delete c; return (int)e; }
    memset(handle_out, 0, PINN_IPC_HANDLE_BYTES);
    memcpy(handle_out, &h, sizeof(h));
    for (int r = 0; r < P2P_MAX_WORLD; ++r) c->peer_base[r] = nullptr;
    *comm_out = c;
    return PINN_OK;
}

int pinn_p2p_connect(pinn_p2p_comm* c, const unsigned char* all_handles) {
    if (!c || !all_handles) return PINN_ERR_NULL;
    if (c->connected) return PINN_OK;
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank) {
            c->peer_base[r] = c->base;
        } else {
            hipIpcMemHandle_t h;
            memcpy(&h, all_handles + (size_t)r * PINN_IPC_HANDLE_BYTES, sizeof(h));
            const hipError_t e = hipIpcOpenMemHandle(&c->peer_base[r], h, hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) return (int)e;
        }
        char* p = static_cast<char*>(c->peer_base[r]);
        c->table.slots[r] = reinterpret_cast<float*>(p);
        c->table.flags[r] = reinterpret_cast<unsigned*>(p + slots_bytes(c->world, c->max_floats));
    }
    c->status = reinterpret_cast<int*>(static_cast<char*>(c->base) + slots_bytes(c->world, c->max_floats) + flags_bytes(c->world));
    c->connected = 1;
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
    hipLaunchKernelGGL(p2p_allreduce_kernel, dim3(P2P_BLOCKS), dim3(256), 0, (hipStream_t)stream, c->table, c->rank, c->world, c->max_floats, buf, (long)n,
                       c->seq, a, c->status);
    return (int)hipGetLastError();
}

int pinn_p2p_status(pinn_p2p_comm* c, int* fine_grained_out) {
    if (!c || !c->connected) return PINN_ERR_COLLECTIVE;
    if (fine_grained_out) *fine_grained_out = c->fine_grained;
    int st = 0;
    const hipError_t e = hipMemcpy(&st, c->status, sizeof(int), hipMemcpyDeviceToHost);      // (synchronises: for tests / the end of a run)
    return e != hipSuccess ? (int)e : st;
}

int pinn_p2p_destroy(pinn_p2p_comm* c) {
    if (!c) return PINN_OK;
    (void)hipDeviceSynchronize();
    if (c->connected)
        for (int r = 0; r < c->world; ++r)
            if (r != c->rank && c->peer_base[r]) (void)hipIpcCloseMemHandle(c->peer_base[r]);
    (void)hipFree(c->base);
    delete c;
    return PINN_OK;
}

}  // extern "C"
