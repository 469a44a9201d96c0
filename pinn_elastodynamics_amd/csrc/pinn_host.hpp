// Host-side launch logic of libpinn_hip.so, templated on the matrix-pipe operand type, the
// split factor and the padded hidden width.  One instantiation per line of pinn_variants.def
// (compiled as its own object by the build, see __graft_entry__.build()).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/pinn_hip.h"
#include "pinn_device.hpp"
#include "pinn_fused.hpp"

namespace pinn {

struct DataSet {                // one value-only point set of pinn_data_loss_grad_multi
    const float *x, *y, *t, *targets;
    long n;
    float tw[16];              // (3-input nets: 8 used; the 3-D data head: up to 16 outputs)
    const float* z = nullptr;  // 4-input nets (the 3-D side sets)
    float* loss_out;
    int head = 0;              // 0: sum_o w_o (Y_o - target_o)^2;  1: hole traction of the plate's composite fields (HEAD_TRACTION), `aux` = [12][n]
    const float* aux = nullptr;
};

// Asynchronous launch timing (pinn_debug_profile_ring_arm / _read): while armed, every launch of a fused kernel is bracketed by HIP
// events on the call's stream and NOTHING synchronises -- the launches keep their place in the stream order of a running step loop, so the
// durations are those of the warm, back-to-back launches the step time is made of.  Read once, after the loop.
struct ProfRing {
    static constexpr int CAP = 4096;
    hipEvent_t ev[CAP][2];
    int tag[CAP];              // streams of the recorded launch (4 / 5: a collocation set, 1: the value-only side sets)
    int created = 0, n = 0, limit = 0;
    int every = 1, steps_seen = 0;      // record every `every`-th step only (a collocation launch opens a step)
    bool armed = false, step_on = true;
    // slot for the next launch, or -1 (not armed / full / a step that is not recorded)
    int begin(hipStream_t st, int tag_) {
        if (!armed || n >= limit) return -1;
        if (tag_ >= 4) step_on = (steps_seen++ % every) == 0;
        if (!step_on) return -1;
        if (n >= created) {
            if (hipEventCreate(&ev[n][0]) != hipSuccess || hipEventCreate(&ev[n][1]) != hipSuccess) return -1;
            created = n + 1;
        }
        tag[n] = tag_;
        hipEventRecord(ev[n][0], st);
        return n++;
    }
    void end(int slot, hipStream_t st) { if (slot >= 0) hipEventRecord(ev[slot][1], st); }
};

struct Call {
    NetDesc net;
    const float* params;
    const float* x;
    const float* y;
    const float* t;
    const float* z;            // 4-input heads only (input order x, y, z, t)
    long n;
    float sx[4], ox[4];        // input map; 4-input heads: index 2 = z, 3 = t
    void* ws;
    size_t ws_bytes;
    hipStream_t stream;
    float* loss_out;
    float* grad_out;
    int accumulate;
    // wave residual head
    float c1, c2, G, rho;
    float tw[16];
    // data head
    const float* targets;
    int nsets;                 // > 0: pinn_data_loss_grad_multi -- the value-only sets of this call (else the single set x, y, t, n, targets, tw)
    DataSet sets[4];
    // fields head
    float* fields_out;
    // plate / traction / stream-target heads
    const float* aux;
    float w5[5][8];
    // optional per-kernel timing (host pointer, 4 floats: repack, chain, wgrad, reductions) -- makes the call synchronous
    float* prof_ms;
    ProfRing* ring;            // optional asynchronous launch timing (see ProfRing)
    unsigned long long* dbg_stamps;   // optional device buffer for the fused kernel's phase timestamps (128 x u64)
    int adj_shift;             // PINN_ADJOINT_SHIFT(k): adjoint seeds scaled by 2^-k inside the kernels, the gradient by 2^k at the reduction
    int weights_packed;        // skip the repack: the workspace already holds the packed form of `params` (same net / precision mode)
    int use_fused;             // 1: prefer the fused kernel where it applies (default), 0: force the two-kernel path
    int one_stream_head;       // single-set one-stream calls: 0 = data head, 1 = hole traction (fused one-stream kernel, see DataSet::head)
    int fast_state;            // PINN_FLAG_STATE_FP16: fused kernel parks its states as fp16 high parts only (faster, less accurate at trained weights)
};

// paths taken by the loss + gradient calls of this process (pinn_debug_path_counts); defined in pinn_capi.hip
extern long g_path_counts[5];
// extra steps of the even-XCD workgroups in 1/1000 (FusedArgs::n_plain; pinn_debug_set_xcd_bonus); 0 = off
extern int g_xcd_tail_permille;
bool xcd_tail_device_ok();      // (pinn_capi.hip) the tail's premises hold on the current device: gfx950, 256 compute units (SPX)
// testing hook (pinn_debug_set_fused_grid_cap): at most this many workgroups in a fused launch (0: no cap beyond FUSED_GRID)
extern int g_fused_grid_cap;

struct Impl {
    int (*path_for)(const NetDesc&, int head, size_t ws_bytes);
    int (*wave_step)(const Call&, const Call&, const AdamEpilogue&, int* rc);      // 1: ran (narrow fused layouts), 0: make the calls one by one
    int (*plate_step)(const Call&, const Call&, const AdamEpilogue&, int* rc);
    int (*wave_loss_grad)(const Call&);
    int (*data_loss_grad)(const Call&);
    int (*fields)(const Call&);
    size_t (*ws_bytes)(const NetDesc&, long n, int minimum);
    // 5-stream family (second time derivative), compiled for the split-precision variants only
    int (*plate_loss_grad)(const Call&);
    int (*traction_loss_grad)(const Call&);
    int (*stream_loss_grad)(const Call&);
    int (*streams)(const Call&);
    // 4-input family (3-D Navier-Cauchy extension: value + 4 first-order streams, 12 outputs), split-precision variants only
    int (*nc3d_loss_grad)(const Call&);
    int (*nc3d_data_loss_grad)(const Call&);
    int (*nc3d_fields)(const Call&);
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// pair of HIP events of the optional per-kernel timing, released on every exit path
struct EventPair {
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool on;
    explicit EventPair(bool enable) : on(enable) { if (on) { hipEventCreate(&ev[0]); hipEventCreate(&ev[1]); } }
    ~EventPair() { if (on) { hipEventDestroy(ev[0]); hipEventDestroy(ev[1]); } }
    EventPair(const EventPair&) = delete;
    EventPair& operator=(const EventPair&) = delete;
};

template <class Op, int SPLIT, int WIDTH>
struct Host {
    // points per wave tile = 16*NB: two blocks for narrow nets, one when the register budget is tight (wide nets, 5 streams)
    template <int NS>
    static constexpr int nb() { return (WIDTH <= 64 && NS != 5) ? 2 : 1; }
    static constexpr int NP = SPLIT == 3 ? 2 : 1;
    static constexpr int NPS = SPLIT == 3 ? 2 : 1;    // stored weight-fragment parts
    static constexpr int FUSED_PARTS = SPLIT == 3 ? 3 : 1;   // ... of the fused kernel's format (repack_kernel)
    static constexpr int NCHUNK = 128;        // split-K slices of the weight-gradient kernel
    static constexpr int FUSED_GRID = 256;    // persistent workgroups of the fused kernel (one per MI355X CU)
    // The fused kernel is a PERSISTENT launch: its grid is the number of per-workgroup scratch images the workspace holds.  A workspace
    // that holds only a handful (pinn_min_workspace_bytes is sized for the two-kernel path) would run the whole call on a few CUs -- far
    // slower than the two-kernel path, silently.  Below this many workgroups (unless the call has fewer steps anyway) the fused path
    // declines and the two-kernel path runs.  (The x86 test build keeps 1: its tests use small workspaces to get several steps per workgroup.)
#if defined(PINN_SIMT_EMULATOR)
    static constexpr long FUSED_MIN_GRID = 1;
#else
    static constexpr long FUSED_MIN_GRID = 64;
#endif
    static constexpr int FUSED_MAX_WIDTH = 160;     // widest padded net the fused kernel takes (160: 4 streams, 6 layers = CONF:891; 128: 4 and 1 streams (+ the 3-D head); 96 also 5 streams)
    static constexpr size_t FUSED_ACC_W64 = 32 * 1024;
    static constexpr size_t FUSED_ACC_BYTES = WIDTH <= 64 ? FUSED_ACC_W64 : (WIDTH <= 96 ? 72 * 1024 : 160 * 1024);     // per weight-gradient wave: in-memory accumulator blocks
    // fused_step_kernel (collocation set + side sets of a training step in one launch): the narrow layouts, and (round 6) every LDS-operand
    // layout that has both of its parts -- the reference's own nets pay 0.36 ms (8 x 80) / 0.61 ms (8 x 100) of a 6.6 / 10.6 ms step for their
    // side sets as a second launch (profiles/r06_wide_kernel_stats.csv)
    static constexpr bool step_has() { return true; }
    template <int NSC>
    static constexpr bool step_has_ns() {
        if (WIDTH <= 64) return NSC == 4 || SPLIT == 3;      // (the plate's five streams: split-precision families)
        // (padded width 160 keeps the separate calls: CONF's own step -- 185 k collocation + 90 k side points, tools/conf_step_time.py -- measured
        // 3.34 / 3.32 ms as one launch against 3.30 / 3.29 as two: its side part is eleven rounds of its own, nothing to hide in a tail)
        return WIDTH < 160 && fused_has<NSC>() && fused_has<1>();
    }
    // depth of the fused instantiations of this width (narrow: 4 or 8, by the net)
    static constexpr int WIDE_NL = WIDTH == 160 ? 6 : 8;
    template <int NS>
    static constexpr bool fused_has() { return WIDTH <= 64 || (SPLIT == 3 && ((WIDTH <= 96 && (NS == 4 || NS == 5 || NS == 1)) || (WIDTH <= 128 && (NS == 4 || NS == 1)) || (WIDTH == 160 && (NS == 4 || NS == 1)))); }
    static constexpr int MAX_BLOCKS = 2048;   // chain kernel grid cap (4 waves per block)
    static constexpr long MIN_TILES = 64;
    static constexpr int MAX_REPACK_BLOCKS = 2048;
    typedef FragIndex<WIDTH> FI;

    struct Plan {
        size_t w0p, bias_mid, bias_last, wflags, frags, frags_fused, loss_part, partial, wg_acc, panels, fixed_end;
        size_t loss_part_b, partial_b, wg_acc_b;      // fused_step_kernel: the side-set part's own partial sums (narrow nets only)
        long s_tile, z_tile;     // 16-bit elements per tile
        long ntiles;             // total tiles of the call (even)
        long chunk_tiles;        // tiles per workspace pass (even)
    };

    template <int NS>
    static void plan_fixed(const NetDesc& net, long n, Plan& p) {
        constexpr int NB = nb<NS>(), TP = 16 * NB;
        typedef PanelGeom<WIDTH, NB, NS, NP> PG;
        size_t o = 0;
        p.w0p = o;
        o = align_up(o + (size_t)WIDTH * 8 * sizeof(float), 256);
        p.bias_mid = o;
        o = align_up(o + (size_t)(net.nl > 1 ? net.nl - 1 : 1) * WIDTH * sizeof(float), 256);
        p.bias_last = o;
        o = align_up(o + NOUT_PAD * sizeof(float), 256);
        p.wflags = o;           // repack_kernel's per-block weight-range flags
        o = align_up(o + (size_t)MAX_REPACK_BLOCKS * sizeof(int), 256);
        p.frags = o;
        o = align_up(o + (size_t)FI::total(net.nl) * NPS * 64 * sizeof(u32x4), 256);
        p.frags_fused = o;      // second copy in the fused kernel's format (narrow nets only)
        if (WIDTH <= FUSED_MAX_WIDTH) o = align_up(o + (size_t)FI::total(net.nl) * FUSED_PARTS * 64 * sizeof(u32x4), 256);
        p.loss_part = o;
        o = align_up(o + (size_t)MAX_BLOCKS * 4 * LOSS_SLOTS_3D * sizeof(float), 256);
        p.partial = o;
        o = align_up(o + (size_t)(NCHUNK > FUSED_GRID ? NCHUNK : FUSED_GRID) * net.nparams * sizeof(float), 256);
        p.wg_acc = o;           // fused kernel: weight-gradient accumulator blocks kept in memory (<= 2 layers x 4 blocks x 1 KB per wave)
        if (WIDTH <= FUSED_MAX_WIDTH) o = align_up(o + (size_t)FUSED_GRID * 4 * FUSED_ACC_BYTES, 256);
        // (fused_step_kernel's second set of partial-sum areas -- the side-set part of the launch -- is NOT part of the fixed plan since round 6:
        // step() places it behind the scratch images of the call that needs it, sized for that call's side-set grid.  Round 5 reserved
        // FUSED_GRID x nparams x 4 + 32 MB here for every narrow-net engine, also the frozen 4 x 20 nets and inference-only use.)
        p.loss_part_b = p.partial_b = p.wg_acc_b = 0;
        p.panels = o;
        p.fixed_end = o;
        p.s_tile = PG::s_tile(net.nl);
        p.z_tile = PG::z_tile(net.nl);
        long nt = (n + TP - 1) / TP;
        p.ntiles = (nt + 1) & ~1L;
    }

    template <int NS>
    static size_t ws_bytes_ns(const NetDesc& net, long n, int minimum) {
        Plan p;
        plan_fixed<NS>(net, n, p);
        const long tiles = minimum ? (p.ntiles < MIN_TILES ? p.ntiles : MIN_TILES) : p.ntiles;
        return p.fixed_end + (size_t)tiles * (size_t)(p.s_tile + p.z_tile) * 2;
    }
    static size_t ws_bytes(const NetDesc& net, long n, int minimum) {
        const size_t a = ws_bytes_ns<4>(net, n, minimum);
        if constexpr (SPLIT == 3) {
            const size_t b = ws_bytes_ns<5>(net, n, minimum);
            return a > b ? a : b;
        }
        return a;
    }

    template <int NS>
    static int make_plan(const Call& c, Plan& p, bool panels) {
        if (((uintptr_t)c.ws & 255) != 0) return PINN_ERR_WORKSPACE;
        plan_fixed<NS>(c.net, c.n, p);
        if (c.ws_bytes < p.fixed_end) return PINN_ERR_WORKSPACE;
        if (!panels) { p.chunk_tiles = p.ntiles; return PINN_OK; }
        const size_t per_tile = (size_t)(p.s_tile + p.z_tile) * 2;
        long fit = (long)((c.ws_bytes - p.fixed_end) / per_tile) & ~1L;
        if (fit > p.ntiles) fit = p.ntiles;
        if (fit < 2) return PINN_ERR_WORKSPACE;
        p.chunk_tiles = fit;
        return PINN_OK;
    }

    static PackedWeights packed(const Call& c, const Plan& p) {
        char* b = static_cast<char*>(c.ws);
        PackedWeights pw;
        pw.w0p = reinterpret_cast<const float*>(b + p.w0p);
        pw.bias_mid = reinterpret_cast<const float*>(b + p.bias_mid);
        pw.bias_last = reinterpret_cast<const float*>(b + p.bias_last);
        pw.frags = reinterpret_cast<const u32x4*>(b + p.frags);
        return pw;
    }

    static int repack_blocks(const NetDesc& net) {
        const long items = (long)FI::total(net.nl) * 64;
        const long blocks = (items + 255) / 256;
        return (int)(blocks < MAX_REPACK_BLOCKS ? blocks : MAX_REPACK_BLOCKS);      // (the kernel strides: any grid covers all items)
    }
    static int repack(const Call& c, const Plan& p) {
        if (c.weights_packed) return 0;      // PINN_FLAG_WEIGHTS_PACKED: the previous call on this workspace left them in place
        char* b = static_cast<char*>(c.ws);
        RepackArgs ra;
        ra.net = c.net;
        ra.params = c.params;
        ra.w0p = reinterpret_cast<float*>(b + p.w0p);
        ra.bias_mid = reinterpret_cast<float*>(b + p.bias_mid);
        ra.bias_last = reinterpret_cast<float*>(b + p.bias_last);
        ra.frags = reinterpret_cast<u32x4*>(b + p.frags);
        ra.frags_fused = WIDTH <= FUSED_MAX_WIDTH ? reinterpret_cast<u32x4*>(b + p.frags_fused) : nullptr;
        ra.wflags = reinterpret_cast<int*>(b + p.wflags);
        const int blocks = repack_blocks(c.net);
        hipLaunchKernelGGL((repack_kernel<Op, SPLIT, WIDTH>), dim3(blocks), dim3(256), 0, c.stream, ra);
        return (int)hipGetLastError();
    }

    static void fill_common(const Call& c, const Plan& p, ChainArgs& a) {
        a.net = c.net;
        a.pw = packed(c, p);
        a.x = c.x;
        a.y = c.y;
        a.t = c.t;
        a.z = c.z;
        a.n = c.n;
        for (int k = 0; k < 4; ++k) { a.sx[k] = c.sx[k]; a.ox[k] = c.ox[k]; }
        a.c1 = c.c1;
        a.c2 = c.c2;
        a.G = c.G;
        a.rho = c.rho;
        a.targets = c.targets;
        a.fields_out = c.fields_out;
        a.aux = c.aux;
        for (int i = 0; i < 5; ++i)
            for (int o = 0; o < 8; ++o) a.w5[i][o] = 0.0f;
        a.S = nullptr;
        a.Z = nullptr;
        a.S_tile_stride = p.s_tile;
        a.Z_tile_stride = p.z_tile;
        a.loss_part = reinterpret_cast<float*>(static_cast<char*>(c.ws) + p.loss_part);
    }

    static int chain_blocks(long ntiles) {
        long b = (ntiles + 3) / 4;
        if (b > MAX_BLOCKS) b = MAX_BLOCKS;
        if (b < 1) b = 1;
        return (int)b;
    }

    // forward + reverse chain + weight gradient for one head (NS streams)
    template <int NS, int HEAD>
    static int loss_grad(const Call& c, int nterms) {
        constexpr int NB = nb<NS>();
        Plan p;
        int rc = make_plan<NS>(c, p, true);
        if (rc) return rc;
        ++g_path_counts[PINN_PATH_TWO_KERNEL];
        // optional HIP-event timing of each kernel class (bench.py's roofline leg)
        float acc_ms[4] = {0.f, 0.f, 0.f, 0.f};
        const bool prof = c.prof_ms != nullptr;
        EventPair evp(prof);
        hipEvent_t (&ev)[2] = evp.ev;
        auto tic = [&]() { if (prof) hipEventRecord(ev[0], c.stream); };
        auto toc = [&](int slot) {
            if (!prof) return;
            hipEventRecord(ev[1], c.stream);
            hipEventSynchronize(ev[1]);
            float ms = 0.f;
            hipEventElapsedTime(&ms, ev[0], ev[1]);
            acc_ms[slot] += ms;
        };
        tic();
        rc = repack(c, p);
        if (rc) return rc;
        toc(0);
        float twmax = 0.0f;
        for (int i = 0; i < 16; ++i) { const float v = c.tw[i] < 0 ? -c.tw[i] : c.tw[i]; if (v > twmax) twmax = v; }
        if (HEAD != HEAD_STREAMS) twmax *= (float)(1u << c.adj_shift);      // the weights are only ever used normalised: folds the shift in
        ChainArgs a;
        fill_common(c, p, a);
        for (int i = 0; i < 16; ++i) a.tw[i] = twmax > 0.0f ? c.tw[i] / twmax : 0.0f;
        if (HEAD == HEAD_STREAMS) {
            twmax = 0.0f;
            for (int i = 0; i < 5; ++i)
                for (int o = 0; o < 8; ++o) { const float v = c.w5[i][o] < 0 ? -c.w5[i][o] : c.w5[i][o]; if (v > twmax) twmax = v; }
            for (int i = 0; i < 5; ++i)
                for (int o = 0; o < 8; ++o) a.w5[i][o] = twmax > 0.0f ? c.w5[i][o] / twmax : 0.0f;
        }
        char* b = static_cast<char*>(c.ws);
        a.S = reinterpret_cast<uint16_t*>(b + p.panels);
        a.Z = a.S + p.chunk_tiles * p.s_tile;
        WgradArgs w;
        w.net = c.net;
        w.S = a.S;
        w.Z = a.Z;
        w.S_tile_stride = p.s_tile;
        w.Z_tile_stride = p.z_tile;
        w.partial = reinterpret_cast<float*>(b + p.partial);
        int pass = 0;
        for (long t0 = 0; t0 < p.ntiles; t0 += p.chunk_tiles, ++pass) {
            const long nt = (p.ntiles - t0) < p.chunk_tiles ? (p.ntiles - t0) : p.chunk_tiles;
            a.tile0 = t0;
            a.ntiles = nt;
            const int blocks = chain_blocks(nt);
            tic();
            hipLaunchKernelGGL((chain_kernel<Op, SPLIT, WIDTH, NB, NS, HEAD>), dim3(blocks), dim3(256), 0, c.stream, a);
            if ((rc = (int)hipGetLastError())) return rc;
            toc(1);
            tic();
            hipLaunchKernelGGL((reduce_loss_kernel<0>), dim3(1), dim3(256), 0, c.stream, (const float*)a.loss_part, (long)blocks * 4, nterms,
                               c.loss_out, pass > 0 ? 1 : 0, head_is_3d(HEAD) ? LOSS_SLOTS_3D : 8, (const int*)nullptr, 0);
            if ((rc = (int)hipGetLastError())) return rc;
            toc(3);
            w.ntiles = nt;
            w.first_pass = pass == 0 ? 1 : 0;
            tic();
            hipLaunchKernelGGL((wgrad_kernel<Op, SPLIT, WIDTH, NB, NS>), dim3(NCHUNK, c.net.nl + 1), dim3(64 * WgradCfg<WIDTH>::NW), 0, c.stream, w);
            if ((rc = (int)hipGetLastError())) return rc;
            toc(2);
        }
        tic();
        hipLaunchKernelGGL((reduce_grad_kernel<0>), dim3((c.net.nparams + 63) / 64), dim3(256), 0, c.stream, (const float*)w.partial,
                           (int)NCHUNK, c.net.nparams, twmax, c.grad_out, c.accumulate);
        rc = (int)hipGetLastError();
        toc(3);
        if (prof)
            for (int i = 0; i < 4; ++i) c.prof_ms[i] = acc_ms[i];
        return rc;
    }

    // fused path (pinn_fused.hpp): padded width <= 64 and a compiled depth; needs the per-wave state scratch in the workspace
    // The value-only sets of a call as a table (a single-set call becomes a table of one).
    static int data_sets(const Call& c, DataSet (&sets)[4]) {
        if (c.nsets > 0) {
            int m = 0;
            for (int k = 0; k < c.nsets; ++k)
                if (c.sets[k].n > 0) sets[m++] = c.sets[k];
            return m;
        }
        sets[0].x = c.x;
        sets[0].y = c.y;
        sets[0].t = c.t;
        sets[0].targets = c.targets;
        sets[0].n = c.n;
        sets[0].z = c.z;
        for (int i = 0; i < 16; ++i) sets[0].tw[i] = c.tw[i];
        sets[0].loss_out = c.loss_out;
        sets[0].head = c.one_stream_head;
        sets[0].aux = c.aux;
        return 1;
    }

    // the 3-D instantiation (4 inputs, five first-order streams, 10 x 128: BASELINE configs[4]) exists for the split-precision width-128 family
    static constexpr bool fused_has_3d() { return SPLIT == 3 && WIDTH == 128; }
    // Arguments of one part of a fused launch: `area` 0 = the call's own per-workgroup areas, 1 = the second set (the side-set part of
    // fused_step_kernel); `scratch_off` = byte offset of this part's scratch images behind p.panels; `block0` = its first workgroup.
    struct FusedSetup {
        FusedArgs a;
        float twmax;
        LossOuts lo;
        int nsets;
    };
    template <int NL, int NS, bool FS = false, int DIN = 3>
    static void fused_setup(const Call& c, const Plan& p, int grid, long nsteps, int area, size_t scratch_off, int block0, FusedSetup& su) {
        typedef Fused<Op, SPLIT, WIDTH, NL, NS, FS, DIN> F;
        char* b = static_cast<char*>(c.ws);
        FusedArgs& a = su.a;
        a.net = c.net;
        a.pw = packed(c, p);
        a.pw.frags = reinterpret_cast<const u32x4*>(b + p.frags_fused);
        a.frags_bytes = (unsigned)((size_t)FI::total(c.net.nl) * FUSED_PARTS * 64 * sizeof(u32x4));
        a.x = c.x;
        a.y = c.y;
        a.t = c.t;
        a.z = c.z;
        a.n = c.n;
        a.nsteps = nsteps;
        for (int k = 0; k < 4; ++k) { a.sx[k] = c.sx[k]; a.ox[k] = c.ox[k]; }
        a.c1 = c.c1;
        a.c2 = c.c2;
        a.G = c.G;
        a.rho = c.rho;
        a.targets = nullptr;
        a.aux = c.aux;
        float twmax = 0.0f;
        LossOuts lo = {{nullptr, nullptr, nullptr, nullptr}};
        int nsets = 1;
        if constexpr (NS == 1) {
            DataSet sets[4];
            nsets = data_sets(c, sets);
            constexpr int NTW = DIN == 4 ? 16 : 8;      // output weights of a set
            for (int k = 0; k < nsets; ++k)
                for (int i = 0; i < NTW; ++i) { const float v = sets[k].tw[i] < 0 ? -sets[k].tw[i] : sets[k].tw[i]; if (v > twmax) twmax = v; }
            twmax *= (float)(1u << c.adj_shift);
            if constexpr (F::WG_HI) twmax *= 1.0f / F::ZDB_SEED_SCALE;
            long s0 = 0;
            for (int k = 0; k < 4; ++k) {
                const bool on = k < nsets;
                a.set_step0[k] = s0;
                a.set_x[k] = on ? sets[k].x : nullptr;
                a.set_y[k] = on ? sets[k].y : nullptr;
                a.set_t[k] = on ? sets[k].t : nullptr;
                a.set_z[k] = on ? sets[k].z : nullptr;
                a.set_targets[k] = on ? sets[k].targets : nullptr;
                a.set_n[k] = on ? sets[k].n : 0;
                a.set_head[k] = on ? sets[k].head : 0;
                a.set_aux[k] = on ? sets[k].aux : nullptr;
                for (int i = 0; i < 16; ++i) a.set_tw[k][i] = on && i < NTW && twmax > 0.0f ? sets[k].tw[i] / twmax : 0.0f;
                if (on) { s0 += (sets[k].n + 16 * F::TILES - 1) / (16 * F::TILES); lo.p[k] = sets[k].loss_out; }
            }
            a.set_step0[4] = s0;
            a.nsets = nsets;
            for (int i = 0; i < 16; ++i) a.tw[i] = 0.0f;
        } else {
            for (int i = 0; i < 16; ++i) { const float v = c.tw[i] < 0 ? -c.tw[i] : c.tw[i]; if (v > twmax) twmax = v; }
            twmax *= (float)(1u << c.adj_shift);
            if constexpr (F::WG_HI) twmax *= 1.0f / F::ZDB_SEED_SCALE;      // adjoint seeds x 16: the weight gradient's fp16 adjoints in the normal range (Fused::ZDB)
            for (int i = 0; i < 16; ++i) a.tw[i] = twmax > 0.0f ? c.tw[i] / twmax : 0.0f;
            a.nsets = 1;
            lo.p[0] = c.loss_out;
        }
        a.scratch = reinterpret_cast<u32x4*>(b + p.panels + scratch_off);
        a.loss_part = reinterpret_cast<float*>(b + (area ? p.loss_part_b : p.loss_part));
        a.partial = reinterpret_cast<float*>(b + (area ? p.partial_b : p.partial));
        a.wg_acc = reinterpret_cast<u32x4*>(b + (area ? p.wg_acc_b : p.wg_acc));
        static_assert(F::WG_ACC_BYTES <= FUSED_ACC_BYTES, "accumulator area of the plan");
        a.dbg = c.dbg_stamps;
        a.block0 = block0;
        a.grid = grid;
        // XCD-aware step assignment (FusedArgs::n_plain): the collocation part of a full grid with enough rounds for the skew to be expressible in
        // whole steps: g_xcd_tail_permille / 1000 more steps for the even-XCD workgroups, taken as a tail behind R plain rounds with
        // 128 (R + e) + 128 R = nsteps, e = skew * R; not the side-set part (its workgroups start wherever a compute unit frees up)
        a.n_plain = 0x7fffffffffffffffL;
#if defined(PINN_SIMT_EMULATOR)
        const bool shape_ok = grid >= 8 && grid % 8 == 0 && nsteps >= 4L * grid;      // (the x86 test build: small grids, a few rounds -- the index arithmetic is the point)
#else
        const bool shape_ok = grid == FUSED_GRID && nsteps >= 64L * FUSED_GRID;
#endif
        if (NS >= 4 && DIN == 3 && block0 == 0 && shape_ok && g_xcd_tail_permille > 0 && xcd_tail_device_ok()) {
            const long R = (long)((double)nsteps / ((grid / 2) * (2.0 + 0.001 * g_xcd_tail_permille)));
            a.n_plain = R * grid;
        }
        su.twmax = twmax;
        su.lo = lo;
        su.nsets = nsets;
    }

    template <int NL, int NS, bool FS = false, int DIN = 3>
    static int fused_launch(const Call& c, const Plan& p, int grid, int nterms, long nsteps) {
        if constexpr (DIN == 4 ? fused_has_3d() : fused_has<NS>()) {
            typedef Fused<Op, SPLIT, WIDTH, NL, NS, FS, DIN> F;
            int rc = repack(c, p);
            if (rc) return rc;
            char* b = static_cast<char*>(c.ws);
            FusedSetup su;
            fused_setup<NL, NS, FS, DIN>(c, p, grid, nsteps, 0, 0, 0, su);
            const FusedArgs& a = su.a;
            const float twmax = su.twmax;
            const LossOuts lo = su.lo;
            const int nsets = su.nsets;
            EventPair evp(c.prof_ms != nullptr);
            hipEvent_t (&ev)[2] = evp.ev;
            if (c.prof_ms) hipEventRecord(ev[0], c.stream);
            const int ring_slot = c.ring ? c.ring->begin(c.stream, NS) : -1;
            hipLaunchKernelGGL((fused_wave_kernel<Op, SPLIT, WIDTH, NL, NS, FS, DIN>), dim3(grid), dim3(512), 0, c.stream, a);
            if (c.ring) c.ring->end(ring_slot, c.stream);
            if ((rc = (int)hipGetLastError())) return rc;
            if (c.prof_ms) {
                hipEventRecord(ev[1], c.stream);
                hipEventSynchronize(ev[1]);
                c.prof_ms[0] = c.prof_ms[2] = c.prof_ms[3] = 0.f;
                hipEventElapsedTime(&c.prof_ms[1], ev[0], ev[1]);
            }
            // loss partials are [wave][set][8] with set = FUSED_MAX_SETS slots for NS = 1 and one slot for NS = 4
            constexpr int SLOTS = NS == 1 ? FUSED_MAX_SETS : 1;
            if constexpr (DIN == 4) {
                // 12 terms in LOSS_SLOTS_3D slots per tile: the gradient blocks of the fused reduction, then the loss reduction of the two-kernel path
                hipLaunchKernelGGL((reduce_grad_loss_kernel<0>), dim3((c.net.nparams + 63) / 64), dim3(256), 0, c.stream, (const float*)a.partial,
                                   grid, c.net.nparams, twmax, c.grad_out, c.accumulate, (const float*)a.loss_part, (long)grid * F::TILES, 0, 0, SLOTS, lo,
                                   (const int*)(b + p.wflags), SPLIT == 3 ? repack_blocks(c.net) : 0);
                hipLaunchKernelGGL((reduce_loss_kernel<0>), dim3(1), dim3(256), 0, c.stream, (const float*)a.loss_part, (long)grid * F::TILES, nterms,
                                   c.loss_out, 0, LOSS_SLOTS_3D * SLOTS, (const int*)(b + p.wflags), SPLIT == 3 ? repack_blocks(c.net) : 0);      // (NS = 1: set 0's slots of [tile][FUSED_MAX_SETS][16])
                return (int)hipGetLastError();
            }
            hipLaunchKernelGGL((reduce_grad_loss_kernel<0>), dim3((c.net.nparams + 63) / 64 + nsets), dim3(256), 0, c.stream, (const float*)a.partial,
                               grid, c.net.nparams, twmax, c.grad_out, c.accumulate, (const float*)a.loss_part, (long)grid * F::TILES, nterms, nsets,
                               SLOTS, lo, (const int*)(b + p.wflags), SPLIT == 3 ? repack_blocks(c.net) : 0);
            return (int)hipGetLastError();
        } else {
            return PINN_ERR_LAYERS;
        }
    }

    // One training step's sets in one launch + one reduction (+ Adam): fused_step_kernel, reduce_step_kernel.  c = the collocation call
    // (pinn_wave2d_loss_grad's arguments), d = the side sets (pinn_data_loss_grad_multi's; d.nsets > 0).  Returns 1 if it ran (rc in *out), 0 if
    // this net / workspace / set sizes do not take it -- the caller then makes the two calls (+ pinn_adam_step) one after the other: same bits.
    template <int NL, int NSC, bool FS>
    static int step_launch(const Call& c, const Call& d, const AdamEpilogue& adam, const Plan& p, int grid4, long nsteps4, int grid1, long nsteps1, size_t off1,
                           int nterms_a, int nterms_b) {
        typedef Fused<Op, SPLIT, WIDTH, NL, NSC, FS, 3> F4;
        typedef Fused<Op, SPLIT, WIDTH, NL, 1, false, 3> F1;
        int rc = repack(c, p);
        if (rc) return rc;
        char* b = static_cast<char*>(c.ws);
        FusedSetup s4, s1;
        fused_setup<NL, NSC, FS, 3>(c, p, grid4, nsteps4, 0, 0, 0, s4);
        fused_setup<NL, 1, false, 3>(d, p, grid1, nsteps1, 1, off1, grid4, s1);
        const int ring_slot = c.ring ? c.ring->begin(c.stream, NSC) : -1;
        hipLaunchKernelGGL((fused_step_kernel<Op, SPLIT, WIDTH, NL, NSC, FS>), dim3(grid4 + grid1), dim3(512), 0, c.stream, s4.a, s1.a);
        if (c.ring) c.ring->end(ring_slot, c.stream);
        if ((rc = (int)hipGetLastError())) return rc;
        StepPart A = {(const float*)s4.a.partial, grid4, s4.twmax, (const float*)s4.a.loss_part, (long)grid4 * F4::TILES};
        StepPart B = {(const float*)s1.a.partial, grid1, s1.twmax, (const float*)s1.a.loss_part, (long)grid1 * F1::TILES};
        hipLaunchKernelGGL((reduce_step_kernel<0>), dim3((c.net.nparams + 63) / 64 + s1.nsets + 1), dim3(256), 0, c.stream, A, B, c.net.nparams, c.grad_out,
                           c.accumulate, nterms_a, c.loss_out, nterms_b, s1.nsets, (int)FUSED_MAX_SETS, s1.lo, adam, (const int*)(b + p.wflags),
                           SPLIT == 3 ? repack_blocks(c.net) : 0);
        return (int)hipGetLastError();
    }
    // NSC = 4: the wave step (c: pinn_wave2d_loss_grad's call, d: the value-only sets, nterms 7 / n_out); NSC = 5: the plate's (c: pinn_plate2d_loss_grad's
    // call, d: the hole-traction set as a one-set call with one_stream_head = 1, nterms 5 / 2)
    template <int NSC>
    static int step(const Call& c, const Call& d, const AdamEpilogue& adam, int* out, int nterms_a, int nterms_b) {
        if constexpr (step_has_ns<NSC>()) {
            if (!c.use_fused || c.prof_ms != nullptr || !fused_depth<NSC>(c.net) || !fused_depth<1>(c.net)) return 0;
            if (((uintptr_t)c.ws & 255) != 0 || c.n <= 0) return 0;
            Plan p;
            plan_fixed<4>(c.net, c.n, p);
            constexpr int T4 = Fused<Op, SPLIT, WIDTH, WIDE_NL, NSC>::TILES, T1 = Fused<Op, SPLIT, WIDTH, WIDE_NL, 1>::TILES;
            size_t per4, per1;
            if constexpr (WIDTH > 64) {
                per4 = (size_t)T4 * Fused<Op, SPLIT, WIDTH, WIDE_NL, NSC>::SCRATCH_BYTES;
                per1 = (size_t)T1 * Fused<Op, SPLIT, WIDTH, WIDE_NL, 1>::SCRATCH_BYTES;
            } else {
                per4 = (size_t)T4 * (c.net.nl == 4 ? Fused<Op, SPLIT, WIDTH, 4, NSC>::SCRATCH_BYTES : Fused<Op, SPLIT, WIDTH, 8, NSC>::SCRATCH_BYTES);
                per1 = (size_t)T1 * (c.net.nl == 4 ? Fused<Op, SPLIT, WIDTH, 4, 1>::SCRATCH_BYTES : Fused<Op, SPLIT, WIDTH, 8, 1>::SCRATCH_BYTES);
            }
            const long nsteps4 = (c.n + 16 * T4 - 1) / (16 * T4);
            long nsteps1 = 0;
            DataSet sets[4];
            const int m = data_sets(d, sets);
            for (int k = 0; k < m; ++k) nsteps1 += (sets[k].n + 16 * T1 - 1) / (16 * T1);
            if (nsteps1 == 0) return 0;
            const long grid4 = nsteps4 < FUSED_GRID ? nsteps4 : FUSED_GRID, grid1 = nsteps1 < FUSED_GRID ? nsteps1 : FUSED_GRID;
            const size_t off1 = align_up((size_t)grid4 * per4, 256);
            // the side-set part's own partial sums, behind its scratch images: [loss partials | gradient partials | in-memory weight-gradient sums]
            size_t bo = align_up(p.fixed_end + off1 + (size_t)grid1 * per1, 256);
            p.loss_part_b = bo;
            bo = align_up(bo + (size_t)grid1 * 4 * FUSED_MAX_SETS * 8 * sizeof(float), 256);
            p.partial_b = bo;
            bo = align_up(bo + (size_t)grid1 * c.net.nparams * sizeof(float), 256);
            p.wg_acc_b = bo;
            bo = align_up(bo + (size_t)grid1 * 4 * FUSED_ACC_BYTES, 256);
            if (c.ws_bytes < bo) return 0;      // (the two calls then size their grids to the workspace one by one)
            if constexpr (WIDTH > 64) {
                *out = step_launch<WIDE_NL, NSC, false>(c, d, adam, p, (int)grid4, nsteps4, (int)grid1, nsteps1, off1, nterms_a, nterms_b);
                g_path_counts[PINN_PATH_FUSED_LDS] += 2;
                return 1;
            } else {
            if (NSC == 4 && c.fast_state && SPLIT == 3 && c.net.nl == 8)
                *out = step_launch<8, NSC, NSC == 4>(c, d, adam, p, (int)grid4, nsteps4, (int)grid1, nsteps1, off1, nterms_a, nterms_b);
            else *out = c.net.nl == 4 ? step_launch<4, NSC, false>(c, d, adam, p, (int)grid4, nsteps4, (int)grid1, nsteps1, off1, nterms_a, nterms_b)
                                      : step_launch<8, NSC, false>(c, d, adam, p, (int)grid4, nsteps4, (int)grid1, nsteps1, off1, nterms_a, nterms_b);
            g_path_counts[PINN_PATH_FUSED_REGISTERS] += 2;      // (both families of the step)
            return 1;
            }
        } else {
            return 0;
        }
    }
    static int wave_step(const Call& c, const Call& d, const AdamEpilogue& adam, int* out) { return step<4>(c, d, adam, out, 7, c.net.nout); }
    static int plate_step(const Call& c, const Call& d, const AdamEpilogue& adam, int* out) { return step<5>(c, d, adam, out, 5, 2); }

    // The depths the fused kernel is compiled for (the ones the reference's scripts use): the rule of try_fused AND of pinn_path_for.
    template <int NS>
    static bool fused_depth(const NetDesc& net) {
        if constexpr (!fused_has<NS>()) return false;
        if constexpr (WIDTH == 160) return net.nl == 6;          // padded width 160: the reference's confined-domain net, 6 x 140 (CONF:891)
        if (net.nl != 4 && net.nl != 8) return false;
        if (WIDTH > 64 && net.nl != 8) return false;            // padded widths 96 / 128: the 8-layer instantiations only (INF:645 8 x 80, SEMI:679 8 x 100)
        return true;
    }
    // scratch images (= persistent workgroups) a workspace of ws_bytes holds for the NS-stream fused kernel of this net
    template <int NS>
    static long fused_images(const NetDesc& net, size_t ws_bytes) {
        if constexpr (fused_has<NS>()) {
            Plan p;
            plan_fixed<4>(net, 1, p);
            constexpr int TILES = Fused<Op, SPLIT, WIDTH, WIDTH == 160 ? 6 : 4, NS>::TILES;
            // (sized for the default layout, which parks more than the fp16-state one)
            size_t per_wg;
            if constexpr (WIDTH == 160) per_wg = (size_t)TILES * Fused<Op, SPLIT, WIDTH, 6, NS>::SCRATCH_BYTES;      // (one depth: other depths of a one-stream layout would not fit the LDS)
            else per_wg = (size_t)TILES * (net.nl == 4 ? Fused<Op, SPLIT, WIDTH, 4, NS>::SCRATCH_BYTES : Fused<Op, SPLIT, WIDTH, 8, NS>::SCRATCH_BYTES);
            if (ws_bytes < p.fixed_end + per_wg) return 0;
            return (long)((ws_bytes - p.fixed_end) / per_wg);
        } else {
            return 0;
        }
    }
    template <int NS = 5>
    static long fused_images_3d(const NetDesc& net, size_t ws_bytes) {
        if constexpr (fused_has_3d()) {
            typedef Fused<Op, SPLIT, WIDTH, 10, NS, false, 4> F;
            Plan p;
            plan_fixed<4>(net, 1, p);
            const size_t per_wg = (size_t)F::TILES * F::SCRATCH_BYTES;
            if (ws_bytes < p.fixed_end + per_wg) return 0;
            return (long)((ws_bytes - p.fixed_end) / per_wg);
        } else {
            return 0;
        }
    }
    // pinn_path_for: the path a call of this family takes for this net (ws_bytes == 0: a workspace that holds every scratch image)
    static int path_for(const NetDesc& net, int head, size_t ws_bytes) {
        const int fused_kind = WIDTH <= 64 ? PINN_PATH_FUSED_REGISTERS : PINN_PATH_FUSED_LDS;
        auto enough = [&](long images) { return ws_bytes == 0 || images >= FUSED_MIN_GRID; };
        switch (head) {
            case PINN_HEAD_WAVE: return fused_depth<4>(net) && enough(fused_images<4>(net, ws_bytes)) ? fused_kind : PINN_PATH_TWO_KERNEL;
            case PINN_HEAD_DATA: return fused_depth<1>(net) && enough(fused_images<1>(net, ws_bytes)) ? fused_kind : PINN_PATH_TWO_KERNEL;
            case PINN_HEAD_PLATE:
                if (SPLIT != 3) return PINN_ERR_PRECISION;
                return fused_depth<5>(net) && enough(fused_images<5>(net, ws_bytes)) ? fused_kind : PINN_PATH_TWO_KERNEL;
            case PINN_HEAD_NC3D:
                if (SPLIT != 3) return PINN_ERR_PRECISION;
                return fused_has_3d() && net.nl == 10 && net.din == 4 && net.nout == 12 && enough(fused_images_3d(net, ws_bytes)) ? PINN_PATH_FUSED_LDS : PINN_PATH_TWO_KERNEL;
            case PINN_HEAD_NC3D_DATA:
                if (SPLIT != 3) return PINN_ERR_PRECISION;
                return fused_has_3d() && net.nl == 10 && net.din == 4 && net.nout == 12 && enough(fused_images_3d<1>(net, ws_bytes)) ? PINN_PATH_FUSED_LDS : PINN_PATH_TWO_KERNEL;
            case PINN_HEAD_STREAMS: return SPLIT == 3 ? PINN_PATH_TWO_KERNEL : PINN_ERR_PRECISION;
            default: return PINN_ERR_LAYERS;
        }
    }

    // returns 1 if the fused path ran (rc in *out), 0 if it does not apply
    template <int NS>
    static int try_fused(const Call& c, int* out, int nterms) {
        if constexpr (fused_has<NS>()) {
            if (!fused_depth<NS>(c.net)) return 0;
            Plan p;
            if (((uintptr_t)c.ws & 255) != 0) return 0;
            plan_fixed<4>(c.net, c.n, p);
            constexpr int TILES = Fused<Op, SPLIT, WIDTH, WIDTH == 160 ? 6 : 4, NS>::TILES;
            long grid = fused_images<NS>(c.net, c.ws_bytes);
            if (grid == 0) return 0;
            if (grid > FUSED_GRID) grid = FUSED_GRID;
            if (g_fused_grid_cap > 0 && grid > g_fused_grid_cap) grid = g_fused_grid_cap;
            long nsteps = 0;
            if (NS == 1) {
                DataSet sets[4];
                const int m = data_sets(c, sets);
                for (int k = 0; k < m; ++k) nsteps += (sets[k].n + 16 * TILES - 1) / (16 * TILES);
            } else {
                nsteps = (c.n + 16 * TILES - 1) / (16 * TILES);
            }
            if (nsteps == 0) return 0;
            if (grid > nsteps) grid = nsteps;
            if (grid < FUSED_MIN_GRID && grid < nsteps) return 0;      // too few scratch images for a persistent launch: two-kernel path
            if constexpr (WIDTH == 160) *out = fused_launch<6, NS>(c, p, (int)grid, nterms, nsteps);
            else if constexpr (WIDTH > 64) *out = fused_launch<8, NS>(c, p, (int)grid, nterms, nsteps);
            else if (c.fast_state && SPLIT == 3 && NS == 4 && c.net.nl == 8) *out = fused_launch<8, NS, true>(c, p, (int)grid, nterms, nsteps);   // the collocation kernel of the 8-layer nets
            else *out = c.net.nl == 4 ? fused_launch<4, NS>(c, p, (int)grid, nterms, nsteps) : fused_launch<8, NS>(c, p, (int)grid, nterms, nsteps);
            ++g_path_counts[WIDTH <= 64 ? PINN_PATH_FUSED_REGISTERS : PINN_PATH_FUSED_LDS];
            return 1;
        } else {
            return 0;
        }
    }

    // the 3-D head through the fused LDS-operand kernel: 10 hidden layers of padded width 128 (BASELINE configs[4])
    // (NS = 5: the collocation head; NS = 1, round 6: a value-only side set -- source, initial or top-surface points -- through the one-stream
    // instantiation with the same parked one-slot layout, so that no set of a 3-D training step is left on the two-kernel path)
    template <int NS = 5>
    static int try_fused_3d(const Call& c, int* out) {
        if constexpr (fused_has_3d()) {
            if (c.net.nl != 10 || c.net.din != 4 || c.net.nout != 12) return 0;
            Plan p;
            if (((uintptr_t)c.ws & 255) != 0) return 0;
            plan_fixed<4>(c.net, c.n, p);
            typedef Fused<Op, SPLIT, WIDTH, 10, NS, false, 4> F;
            const size_t per_wg = (size_t)F::TILES * F::SCRATCH_BYTES;
            if (c.ws_bytes < p.fixed_end + per_wg) return 0;
            long grid = (long)((c.ws_bytes - p.fixed_end) / per_wg);
            if (grid > FUSED_GRID) grid = FUSED_GRID;
            if (g_fused_grid_cap > 0 && grid > g_fused_grid_cap) grid = g_fused_grid_cap;
            const long nsteps = (c.n + 16 * F::TILES - 1) / (16 * F::TILES);
            if (nsteps == 0) return 0;
            if (grid > nsteps) grid = nsteps;
            if (grid < FUSED_MIN_GRID && grid < nsteps) return 0;      // (see FUSED_MIN_GRID)
            *out = fused_launch<10, NS, false, 4>(c, p, (int)grid, 12, nsteps);
            ++g_path_counts[PINN_PATH_FUSED_LDS];
            return 1;
        } else {
            return 0;
        }
    }

    static int wave_loss_grad(const Call& c) {
        int rc = 0;
        if (c.use_fused && try_fused<4>(c, &rc, 7)) return rc;
        return loss_grad<4, HEAD_WAVE>(c, 7);
    }
    static int data_loss_grad(const Call& c) {
        int rc = 0;
        if (c.use_fused && try_fused<1>(c, &rc, c.net.nout)) return rc;
        if (c.nsets == 0) return loss_grad<1, HEAD_DATA>(c, c.net.nout);
        // several sets on the two-kernel path: one after the other, accumulating into the same gradient
        bool first = true;
        for (int k = 0; k < c.nsets; ++k) {
            if (c.sets[k].n <= 0) continue;
            Call s = c;
            s.nsets = 0;
            s.x = c.sets[k].x;
            s.y = c.sets[k].y;
            s.t = c.sets[k].t;
            s.n = c.sets[k].n;
            s.targets = c.sets[k].targets;
            for (int i = 0; i < 8; ++i) s.tw[i] = c.sets[k].tw[i];
            s.loss_out = c.sets[k].loss_out;
            s.accumulate = c.accumulate || !first;
            s.weights_packed = c.weights_packed || !first;
            if ((rc = loss_grad<1, HEAD_DATA>(s, c.net.nout))) return rc;
            first = false;
        }
        return 0;
    }

    template <int NS, int HEAD>
    static int fields_ns(const Call& c) {
        Plan p;
        int rc = make_plan<NS>(c, p, false);
        if (rc) return rc;
        rc = repack(c, p);
        if (rc) return rc;
        ChainArgs a;
        fill_common(c, p, a);
        for (int i = 0; i < 16; ++i) a.tw[i] = 0.0f;
        a.tile0 = 0;
        a.ntiles = p.ntiles;
        hipLaunchKernelGGL((chain_kernel<Op, SPLIT, WIDTH, nb<NS>(), NS, HEAD>), dim3(chain_blocks(p.ntiles)), dim3(256), 0, c.stream, a);
        return (int)hipGetLastError();
    }
    static int fields(const Call& c) { return fields_ns<4, HEAD_FIELDS>(c); }

    // 5-stream family (plate): split-precision variants only
    static int plate_loss_grad(const Call& c) {
        if constexpr (SPLIT == 3) {
            int rc = 0;
            if (c.use_fused && try_fused<5>(c, &rc, 5)) return rc;      // padded width <= 64: five-stream instantiation of the fused kernel
            return loss_grad<5, HEAD_PLATE>(c, 5);
        }
        return PINN_ERR_PRECISION;
    }
    static int traction_loss_grad(const Call& c) {
        if constexpr (SPLIT == 3) {
            // the hole-traction set through the one-stream instantiation of the fused kernel (its head takes the set's kind from the set
            // table: round 4; the plate's last user of the two-kernel path)
            int rc = 0;
            Call t = c;
            t.one_stream_head = 1;
            if (c.use_fused && try_fused<1>(t, &rc, 2)) return rc;
            return loss_grad<1, HEAD_TRACTION>(c, 2);
        }
        return PINN_ERR_PRECISION;
    }
    static int stream_loss_grad(const Call& c) {
        if constexpr (SPLIT == 3) return loss_grad<5, HEAD_STREAMS>(c, c.net.nout);
        return PINN_ERR_PRECISION;
    }
    static int streams(const Call& c) {
        if constexpr (SPLIT == 3) return fields_ns<5, HEAD_FIELDS>(c);
        return PINN_ERR_PRECISION;
    }
    // 4-input family (two-kernel path; the fused kernel covers the reference's 3-input nets only)
    static int nc3d_loss_grad(const Call& c) {
        if constexpr (SPLIT == 3) {
            int rc = 0;
            if (c.use_fused && try_fused_3d(c, &rc)) return rc;
            return loss_grad<5, HEAD_NC3D>(c, 12);
        }
        return PINN_ERR_PRECISION;
    }
    static int nc3d_data_loss_grad(const Call& c) {
        if constexpr (SPLIT == 3) {
            int rc = 0;
            if (c.use_fused && try_fused_3d<1>(c, &rc)) return rc;
            return loss_grad<1, HEAD_DATA3D>(c, c.net.nout);
        }
        return PINN_ERR_PRECISION;
    }
    static int nc3d_fields(const Call& c) {
        if constexpr (SPLIT == 3) return fields_ns<5, HEAD_FIELDS3D>(c);
        return PINN_ERR_PRECISION;
    }

    static const Impl* impl() {
        static const Impl I = {&path_for, &wave_step, &plate_step, &wave_loss_grad, &data_loss_grad, &fields, &ws_bytes,
                               &plate_loss_grad, &traction_loss_grad, &stream_loss_grad, &streams,
                               &nc3d_loss_grad, &nc3d_data_loss_grad, &nc3d_fields};
        return &I;
    }
};

}  // namespace pinn
