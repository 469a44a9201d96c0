// Fused loss+gradient kernel for narrow nets (padded width <= 64): forward chain, residual head,
// reverse chain AND the weight gradient in one persistent launch.
//
// Versus chain_kernel + wgrad_kernel (pinn_device.hpp) nothing per-point ever goes to HBM as a
// [feature][point] panel:
//   * a workgroup = 8 waves in two roles (2 waves per SIMD, <= 256 registers each): waves 0-3 are
//     CHAIN waves, each owning a 16-point tile (64 points per workgroup step); waves 4-7 are
//     WEIGHT-GRADIENT waves, each owning one quadrant of every Wbar_l in persistent accumulators.
//     The matrix pipe of a SIMD is shared by one wave of each role, so weight-gradient MFMAs fill
//     the gaps the chain wave leaves while it runs the tanh / tangent VALU chain;
//   * forward state S_l (fp16, see below) is parked in a per-wave scratch slot as the very
//     [stream][point][feature] image the LDS hand-off needs and comes back by asynchronous LDS-DMA
//     (buffer_load ... lds, probed in tools/probes/lds_dma_probe.hip) into a double-buffered LDS
//     tensor one layer ahead of its use, so the reload costs no registers and no exposed latency;
//   * for the weight gradient  Wbar_l = sum_points S_l^T Z_l  the contraction runs over points, so
//     both operands are needed "feature per lane, points in registers" -- the transpose of the
//     chain layout.  Every chain wave drops its S_l and Z_l tiles into LDS as [point][feature] rows
//     (ds_write_b64) and the weight-gradient waves rebuild MFMA fragments from all four tiles with
//     ds_read_b64_tr_b16 (semantics probed in tools/probes/tr_read_probe.hip);
//   * the weight-gradient accumulators are PERSISTENT MFMA accumulators, written once per launch
//     as per-workgroup partials (deterministic two-stage reduction, no atomics).
// Addressing discipline (this kernel is unrolled over 9 weight layers, so every loop-invariant
// address the compiler can hoist costs a VGPR for the whole launch): weights, biases and scratch
// go through buffer descriptors with ONE lane-offset VGPR and scalar (SGPR) offsets; LDS accesses
// use one lane-base VGPR per tensor plus compile-time immediates.
// Two workgroup barriers per weight layer.
#pragma once
#include "pinn_device.hpp"

namespace pinn {

typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16 lds_v4i16;
typedef __attribute__((address_space(3))) void lds_void;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int FUSED_MAX_SETS = 4;

struct FusedArgs {
    NetDesc net;
    PackedWeights pw;
    unsigned frags_bytes;      // size of pw.frags
    const float* x;
    const float* y;
    const float* t;
    long n;
    long nsteps;               // workgroup steps = ceil(n / (16 * TILES))
    float sx[3], ox[3];
    float c1, c2, G, rho;
    float tw[8];
    const float* targets;      // NS = 1 only: [nout][n] or nullptr (= 0)   (single-set form; the set table below supersedes it)
    // NS = 1: up to FUSED_MAX_SETS value-only point sets in ONE launch (loss_IC, loss_SRC, loss_NB, loss_FIX of a step): set k owns
    // the workgroup steps [set_step0[k], set_step0[k+1]) and has its own points, targets, output weights and loss slot
    int nsets;
    long set_step0[5];
    const float* set_x[4];
    const float* set_y[4];
    const float* set_t[4];
    const float* set_targets[4];
    long set_n[4];
    float set_tw[4][8];
    u32x4* scratch;            // [gridDim.x * TILES][SCRATCH_BYTES]: per-tile images of the parked states
    float* loss_part;          // [gridDim.x * TILES][8]
    float* partial;            // [gridDim.x][nparams]
    unsigned long long* dbg;   // optional phase timestamps (s_memtime) of workgroup 0, chain wave 0 / weight-gradient wave 0; nullptr = off
};

__device__ __forceinline__ void fused_stamp(const FusedArgs& a, bool who, int slot) {
    if (a.dbg != nullptr && who) a.dbg[slot] = __builtin_readcyclecounter();
}

// NS = 4: value + three tangent streams, residual head of net_f_sig (the collocation set).  NS = 1: value stream only, head
// sum_o w_o (Y_o - target_o)^2 -- the side sets loss_IC / loss_SRC / loss_NB / loss_FIX (INF:111-118, CONF:145-146).
template <class Op, int SPLIT, int WIDTH, int NL, int NS_ = 4>
struct Fused {
    static constexpr int NS = NS_, WB = WIDTH / 16, KS = WIDTH / 32, NP = SPLIT == 3 ? 2 : 1;
    static_assert(NS == 4 || NS == 1, "wave residual head (4 streams) or value-only data head (1 stream)");
    // weight fragments as stored by repack_kernel: [hi, lo, hi*LO_SCALE] when split; the fused kernel accumulates
    //   acc = (hi*LS).x_hi + hi.x_lo + lo.x_hi = LS * (W.x)   in ONE accumulator per stream (x_lo, lo carry the 2^11 scale)
    static constexpr int NPS = NP;                     // stored parts per weight fragment: [hi, lo] when split
    static_assert(WB == 2 || WB == 4, "fused kernel supports padded widths 32 and 64");
    static_assert(NL >= 2, "fused kernel needs at least two hidden layers");
    static constexpr int IBW = WB / 2, OBW = WB / 2;          // weight-gradient wave (i,o) owns IBW x OBW blocks of every mid Wbar
    static constexpr float INV_LS = 1.0f / Op::LO_SCALE;
    typedef Chain<Op, SPLIT, WIDTH, 1, NS, NS == 4 ? HEAD_WAVE : HEAD_DATA> CH;
    typedef FragIndex<WIDTH> FI;
    // LDS: per chain wave [Z tensor | S tensor]; tensor = NS*NP panels of [16 points][ROWB bytes]
    static constexpr int ROWB = WIDTH * 2 + 8;
    static constexpr int PANEL_B = 16 * ROWB;
    // The parked state S is kept in the operand type's precision only (no low part): measured in tools/precision_study2.py,
    // rounding S for the reverse pass changes the gradient error by < 10 % of itself as long as adjoints and weights stay split.
    static constexpr int TENSOR_Z_B = NS * NP * PANEL_B;
    static constexpr int TENSOR_S_B = NS * PANEL_B;
    // The S tensor is double buffered (layer parity) and filled by LDS-DMA straight from the scratch image, so it is padded
    // to whole 1 KB DMA chunks; the scratch holds the same [stream][point][feature] image per parked layer.
    static constexpr int SBUF_B = (TENSOR_S_B + 1023) / 1024 * 1024;
    // "SLDS": where S_0..S_NL of a tile fit in LDS (NL+1 slots: every 1-stream case, and the 4-stream 4x32 net) nothing is parked in
    // scratch and no LDS-DMA round trip sits between the layer phases; otherwise two slots (layer parity), filled by LDS-DMA
    static constexpr bool SLDS = 4 * (TENSOR_Z_B + (NL + 1) * SBUF_B) <= 160 * 1024;      // all 1-stream cases; 4 streams: 4x32 only
    static constexpr int S_SLOTS = SLDS ? NL + 1 : 2;
    static constexpr int WAVE_B = TENSOR_Z_B + S_SLOTS * SBUF_B;
    static constexpr int LDS_B = 4 * WAVE_B;
    static_assert(LDS_B <= 160 * 1024, "LDS budget");
    static constexpr int TILES = 4;                                           // 16-point tiles per workgroup step (one per chain wave)
    static constexpr unsigned SCRATCH_BYTES = (unsigned)((NL - 1) * SBUF_B);   // per tile: parked states S_1..S_{NL-1}

    struct Acc {                       // persistent across the whole launch, all statically indexed
        f32x4 mid[NL - 1][IBW][OBW];
        f32x4 first;                   // Wbar_0 block (in-block 0, out-block = quad) if quad < WB
        f32x4 last;                    // Wbar_NL block (in-block = quad, out-block 0) if quad < WB
        float bias[NL + 1];
    };

    // ---------------------------------------------------------------------------------------------
    // weight-gradient role
    // ---------------------------------------------------------------------------------------------
    // MFMA fragment "16-feature block at byte column `off` of stream/part panel, 32 points of one k-step" rebuilt from the
    // chain waves' [point][feature] rows.  k-slot (q, e) <-> chain wave 2j + (q>>1), local point 8(q&1) + e (same map for both
    // operands); the lane-dependent part of the address lives in `base`, everything else is an immediate.
    static __device__ __forceinline__ u32x4 get_frag(const char* base, int off) {
        const v4i16 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)(base + off));
        const v4i16 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)(base + off + 4 * ROWB));
        const u32x2 d0 = __builtin_bit_cast(u32x2, v0), d1 = __builtin_bit_cast(u32x2, v1);
        return u32x4{d0[0], d0[1], d1[0], d1[1]};
    }

    // NA x NBK blocks of one weight gradient:  acc[a][b] += sum over 64 points, 4 streams of  S(block fa+a)^T . Z(block fb+b).
    // Operand fragments are fetched once per (k-step, stream) and shared by the NA*NBK blocks; a scheduling fence after every
    // group keeps the compiler from hoisting all transpose-reads of a layer ahead of the MFMAs.
    // sbase/zbase: lane base of the S / Z tensor of chain wave 0 (+ 32 bytes per feature block already added by the caller).
    template <int NA, int NBK>
    static __device__ __forceinline__ void wg_blocks(const char* sbase, const char* zbase, f32x4 (&acc)[NA][NBK], float (&bias_out)[NBK]) {
        f32x4 cc[NA][NBK], bm[NBK], bc[NBK];
#pragma unroll
        for (int b = 0; b < NBK; ++b) {
            bm[b] = f32x4{0.f, 0.f, 0.f, 0.f};
            bc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < NA; ++a) cc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const uint32_t one2 = pack2<Op>(1.0f, 1.0f);
        const u32x4 ones = {one2, one2, one2, one2};
        // software pipeline over the 8 (k-step, stream) groups: the transpose-reads of group g+1 are issued before the MFMAs
        // of group g, so LDS latency hides behind matrix work; the fence after each group bounds how far the compiler may hoist
        struct Frags { u32x4 Ah[NA], Bh[NBK], Bl[NBK]; };
        auto fetch = [&](int g, Frags& f) {
            const int j = g / NS, st = g % NS;
#pragma unroll
            for (int a = 0; a < NA; ++a) f.Ah[a] = get_frag(sbase, 2 * j * WAVE_B + st * PANEL_B + 32 * a);
#pragma unroll
            for (int b = 0; b < NBK; ++b) {
                f.Bh[b] = get_frag(zbase, 2 * j * WAVE_B + (st * NP) * PANEL_B + 32 * b);
                if (NP == 2) f.Bl[b] = get_frag(zbase, 2 * j * WAVE_B + (st * NP + 1) * PANEL_B + 32 * b);
            }
        };
        Frags cur, nxt;
        fetch(0, cur);
#pragma unroll
        for (int g = 0; g < 2 * NS; ++g) {
            if (g + 1 < 2 * NS) fetch(g + 1, nxt);
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int b = 0; b < NBK; ++b) {
                    acc[a][b] = Op::mfma(cur.Ah[a], cur.Bh[b], acc[a][b]);
                    if (NP == 2) cc[a][b] = Op::mfma(cur.Ah[a], cur.Bl[b], cc[a][b]);
                }
            if (g % NS == 0) {                    // bias gradient = ones^T . Z (value stream)
#pragma unroll
                for (int b = 0; b < NBK; ++b) {
                    bm[b] = Op::mfma(ones, cur.Bh[b], bm[b]);
                    if (NP == 2) bc[b] = Op::mfma(ones, cur.Bl[b], bc[b]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < 2 * NS) cur = nxt;
        }
#pragma unroll
        for (int b = 0; b < NBK; ++b) {
            bias_out[b] = NP == 2 ? bm[b][0] + bc[b][0] * INV_LS : bm[b][0];
            if (NP == 2) {
#pragma unroll
                for (int a = 0; a < NA; ++a)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[a][b][r] += cc[a][b][r] * INV_LS;
            }
        }
    }

    // weight gradient of weight layer L (quad = weight-gradient wave index 0..3); lanebase = LDS base + lane part of the address
    template <int L>
    static __device__ __forceinline__ void wgrad(const char* lanebase, Acc& A, int quad) {
        const char* zl = lanebase;                  // Z tensor of chain wave 0
        const char* sl = lanebase + TENSOR_Z_B + (SLDS ? L : (L & 1)) * SBUF_B;     // S tensor (slot of layer L) of chain wave 0
        const int wi = quad >> 1, wo = quad & 1;
        if constexpr (L == 0) {
            if (quad < WB) {
                f32x4 t[1][1] = {{A.first}};
                float b[1];
                wg_blocks<1, 1>(sl, zl + 32 * quad, t, b);
                A.first = t[0][0];
                A.bias[0] += b[0];
            }
        } else if constexpr (L == NL) {
            if (quad < WB) {
                f32x4 t[1][1] = {{A.last}};
                float b[1];
                wg_blocks<1, 1>(sl + 32 * quad, zl, t, b);
                A.last = t[0][0];
                if (quad == 0) A.bias[NL] += b[0];
            }
        } else {
            float b[OBW];
            wg_blocks<IBW, OBW>(sl + 32 * (wi * IBW), zl + 32 * (wo * OBW), A.mid[L - 1], b);
            // one bias block per wave per layer: out-block wo*OBW + wi (OBW == 2) or wo (OBW == 1, waves with wi == 0)
            if (OBW == 1) { if (wi == 0) A.bias[L] += b[0]; }
            else A.bias[L] += wi ? b[OBW - 1] : b[0];
        }
    }

    template <int L>
    struct WgDown {     // same barrier sequence as the chain role's Down<>
        static __device__ __forceinline__ void run(const FusedArgs& a, bool tracer, const char* lanebase, Acc& A, int quad) {
            __syncthreads();                                   // (chain waves now overwrite the tensors)
            fused_stamp(a, tracer, 64 + 3 * (NL - L));
            __syncthreads();                                   // tensors of layer L are complete
            fused_stamp(a, tracer, 65 + 3 * (NL - L));
            wgrad<L>(lanebase, A, quad);
            fused_stamp(a, tracer, 66 + 3 * (NL - L));
            if constexpr (L >= 1) WgDown<L - 1>::run(a, tracer, lanebase, A, quad);
        }
    };

    static __device__ __forceinline__ void wgrad_role(const FusedArgs& a, const char* lds, int quad, int c, int q) {
        Acc A;
#pragma unroll
        for (int l = 0; l < NL - 1; ++l)
#pragma unroll
            for (int i = 0; i < IBW; ++i)
#pragma unroll
                for (int o = 0; o < OBW; ++o) A.mid[l][i][o] = f32x4{0.f, 0.f, 0.f, 0.f};
        A.first = f32x4{0.f, 0.f, 0.f, 0.f};
        A.last = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int l = 0; l <= NL; ++l) A.bias[l] = 0.0f;
        const char* lanebase = lds + (q >> 1) * WAVE_B + (8 * (q & 1) + (c >> 2)) * ROWB + 8 * (c & 3);
        for (long step = blockIdx.x; step < a.nsteps; step += gridDim.x) {
            WgDown<NL>::run(a, blockIdx.x == 0 && quad == 0 && c == 0 && q == 0 && step == 2 * (long)gridDim.x, lanebase, A, quad);
        }
        // ---- write this workgroup's partial gradient
        float* part = a.partial + (long)blockIdx.x * a.net.nparams;
        const int H = a.net.h, NO = a.net.nout, wi = quad >> 1, wo = quad & 1;
        auto put_block = [&](const f32x4& v, int l, int ib, int ob, int n_in, int n_out) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int in = 16 * ib + 4 * q + r, out = 16 * ob + c;
                if (in < n_in && out < n_out) part[a.net.w_off[l] + in * n_out + out] = v[r];
            }
        };
        {
            // Wbar_0 rows 4..6 hold the contribution of the inputs' low parts (see Down<0>): fold them into rows 0..2
            f32x4 lo;
#pragma unroll
            for (int r = 0; r < 4; ++r) lo[r] = __shfl_xor(A.first[r], 16);
#pragma unroll
            for (int r = 0; r < 4; ++r) A.first[r] += lo[r] * INV_LS;
        }
        if (quad < WB) {
            put_block(A.first, 0, 0, quad, 3, H);
            put_block(A.last, NL, quad, 0, H, NO);
            if (q == 0 && 16 * quad + c < H) part[a.net.b_off[0] + 16 * quad + c] = A.bias[0];
        }
        if (quad == 0 && q == 0 && c < NO) part[a.net.b_off[NL] + c] = A.bias[NL];
#pragma unroll
        for (int l = 1; l < NL; ++l) {
#pragma unroll
            for (int i = 0; i < IBW; ++i)
#pragma unroll
                for (int o = 0; o < OBW; ++o) put_block(A.mid[l - 1][i][o], l, wi * IBW + i, wo * OBW + o, H, H);
            const bool owner = OBW == 1 ? wi == 0 : true;
            const int ob = wo * OBW + (OBW == 1 ? 0 : wi);
            if (owner && q == 0 && 16 * ob + c < H) part[a.net.b_off[l] + 16 * ob + c] = A.bias[l];
        }
    }

    // ---------------------------------------------------------------------------------------------
    // chain role
    // ---------------------------------------------------------------------------------------------
    struct Ctx {                                   // wave-invariant addressing state of a chain wave
        __amdgpu_buffer_rsrc_t frags, scr, bias, w0p;
        unsigned lane16;                           // lane * 16: the only VGPR offset of the fragment / scratch traffic
        unsigned rowoff;                           // c*ROWB + 8q: this lane's row/column offset inside a tensor image
        char* tenZ;                                // wave's Z tensor (uniform); S buffers follow at +TENSOR_Z_B (+SBUF_B)
        int c, q;
        bool tracer;                               // workgroup 0, chain wave 0, lane 0
        __device__ __forceinline__ void set_tile(const FusedArgs& a, long gtile) {
            scr = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<char*>(a.scratch) + gtile * (long)SCRATCH_BYTES), 0, (int)SCRATCH_BYTES, 0x00020000);
        }
        __device__ __forceinline__ void init(const FusedArgs& a, char* lds, int slot, int lane, int c_, int q_) {
            frags = __builtin_amdgcn_make_buffer_rsrc((void*)a.pw.frags, 0, (int)a.frags_bytes, 0x00020000);
            bias = __builtin_amdgcn_make_buffer_rsrc((void*)a.pw.bias_mid, 0, (NL - 1) * WIDTH * 4, 0x00020000);
            w0p = __builtin_amdgcn_make_buffer_rsrc((void*)a.pw.w0p, 0, WIDTH * 16, 0x00020000);
            lane16 = (unsigned)lane * 16u;
            tenZ = lds + slot * WAVE_B;
            rowoff = (unsigned)(c_ * ROWB + 8 * q_);
            c = c_;
            q = q_;
            tracer = false;
        }
        __device__ __forceinline__ char* rowZ() const { return tenZ + rowoff; }
        __device__ __forceinline__ char* rowS(int L) const { return tenZ + TENSOR_Z_B + (SLDS ? L : (L & 1)) * SBUF_B + rowoff; }
    };

    // chain-layout fragments -> [point][feature] rows of this wave's LDS tensor (row = lane's point, 8 bytes per feature block)
    // NMB = 16-feature blocks actually present (WB for states/adjoints, 1 for inputs/outputs); NPW = parts written (NP for
    // adjoints, 1 for states)
    template <int KSF, int NMB, int NPW>
    static __device__ __forceinline__ void put_tensor(char* row, const u32x4 (&F)[NS][1][KSF][NP]) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int p = 0; p < NPW; ++p)
#pragma unroll
                for (int mb = 0; mb < NMB; ++mb) {
                    u32x2 v = {F[s][0][mb >> 1][p][(mb & 1) * 2 + 0], F[s][0][mb >> 1][p][(mb & 1) * 2 + 1]};
                    *reinterpret_cast<u32x2*>(row + (s * NPW + p) * PANEL_B + 32 * mb) = v;
                }
    }

    // state (h, hdot_k) of feature block MB of this wave's own points, read back from its LDS S rows
    template <int MB>
    static __device__ __forceinline__ void state_from_lds(const char* rowS, float (&st)[NS][1][4]) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const u32x2 h = *reinterpret_cast<const u32x2*>(rowS + s * PANEL_B + 32 * MB);
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                st[s][0][2 * d + 0] = cvt16<Op>((uint16_t)(h[d] & 0xffffu));
                st[s][0][2 * d + 1] = cvt16<Op>((uint16_t)(h[d] >> 16));
            }
        }
    }

    template <int KSB>
    static __device__ __forceinline__ void load_afrags(const Ctx& x, int frag0, u32x4 (&Af)[KSB][NP]) {
#pragma unroll
        for (int kk = 0; kk < KSB; ++kk)
#pragma unroll
            for (int p = 0; p < NP; ++p) Af[kk][p] = __builtin_amdgcn_raw_buffer_load_b128(x.frags, x.lane16, ((frag0 + kk) * NPS + p) * 1024, 0);
    }
    // acc[s] + accc[s]/LS = sum_k W[.,k] x_s[k]: eight independent MFMA chains (4 streams x {main, correction})
    template <int KSB>
    static __device__ __forceinline__ void gemm_pre(const u32x4 (&Af)[KSB][NP], const u32x4 (&B)[NS][1][KSB][NP], f32x4 (&acc)[NS],
                                                    f32x4 (&accc)[NS]) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            f32x4 m = {0.f, 0.f, 0.f, 0.f}, cc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KSB; ++kk) {
                m = Op::mfma(Af[kk][0], B[s][0][kk][0], m);
                if (NP == 2) {
                    cc = Op::mfma(Af[kk][0], B[s][0][kk][1], cc);
                    cc = Op::mfma(Af[kk][1], B[s][0][kk][0], cc);
                }
            }
            acc[s] = m;
            accc[s] = cc;
        }
    }
    static __device__ __forceinline__ float comb(const f32x4& m, const f32x4& cc, int r) { return NP == 2 ? m[r] + cc[r] * INV_LS : m[r]; }

    // forward first layer (K = 3, VALU): INF:191-195 with the tangent seeds e_k * sx_k
    template <int MB>
    static __device__ __forceinline__ void first_mb(const FusedArgs& a, const Ctx& x, const float (&xin)[3], u32x4 (&Bn)[NS][1][KS][NP]) {
        float vals[NS][1][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const u32x4 wu = __builtin_amdgcn_raw_buffer_load_b128(x.w0p, (unsigned)x.q * 64u, (16 * MB + r) * 16, 0);
            const f32x4 w = __builtin_bit_cast(f32x4, wu);
            float h, sd;
            tanh_act(w[3] + w[0] * xin[0] + w[1] * xin[1] + w[2] * xin[2], h, sd);
            vals[0][0][r] = h;
#pragma unroll
            for (int s = 1; s < NS; ++s) vals[s][0][r] = sd * (a.sx[s - 1] * w[s - 1]);
        }
        CH::template emit<KS, MB>(Bn, vals, nullptr, WIDTH, x.c, x.q);
        if constexpr (MB + 1 < WB) first_mb<MB + 1>(a, x, xin, Bn);
    }

    // forward hidden layer, one 16-feature block per step; the next block's weight fragments are in flight while this
    // block's MFMAs and tanh chain run
    template <int MB>
    static __device__ __forceinline__ void fwd_mb(const Ctx& x, int frag0, int bias_off, const u32x4 (&Af)[KS][NP],
                                                  const u32x4 (&B)[NS][1][KS][NP], u32x4 (&Bn)[NS][1][KS][NP]) {
        u32x4 An[KS][NP];
        if constexpr (MB + 1 < WB) load_afrags<KS>(x, frag0 + (MB + 1) * KS, An);
        const u32x4 bu = __builtin_amdgcn_raw_buffer_load_b128(x.bias, (unsigned)x.q * 16u, bias_off + 16 * MB * 4, 0);
        const f32x4 bias = __builtin_bit_cast(f32x4, bu);
        f32x4 acc[NS], accc[NS];
        gemm_pre<KS>(Af, B, acc, accc);
        float vals[NS][1][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float h, sd;
            tanh_act(comb(acc[0], accc[0], r) + bias[r], h, sd);
            vals[0][0][r] = h;
#pragma unroll
            for (int s = 1; s < NS; ++s) vals[s][0][r] = sd * comb(acc[s], accc[s], r);
        }
        CH::template emit<KS, MB>(Bn, vals, nullptr, WIDTH, x.c, x.q);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MB + 1 < WB) fwd_mb<MB + 1>(x, frag0, bias_off, An, B, Bn);
    }

    // reverse through a weight layer (KSB k-steps of its outputs) + the activation below; state from the wave's LDS rows
    template <int MB, int KSB>
    static __device__ __forceinline__ void bwd_mb(const Ctx& x, int frag0, const char* rowS, const u32x4 (&Af)[KSB][NP],
                                                  const u32x4 (&Zf)[NS][1][KSB][NP], u32x4 (&Zn)[NS][1][KS][NP]) {
        u32x4 An[KSB][NP];
        if constexpr (MB + 1 < WB) load_afrags<KSB>(x, frag0 + (MB + 1) * KSB, An);
        f32x4 acc[NS], accc[NS];
        gemm_pre<KSB>(Af, Zf, acc, accc);
        float vals[NS][1][4];
        // reverse of (h = tanh z, hdot_k = (1-h^2) zdot_k); state of this block from the wave's own LDS rows
        if constexpr (MixF16<Op>::value) {
            // fp16 state consumed in place by mixed-precision FMAs (no v_cvt_f32_f16: 64 per layer otherwise)
            u32x2 sp[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) sp[s] = *reinterpret_cast<const u32x2*>(rowS + s * PANEL_B + 32 * MB);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t hw = sp[0][r >> 1];
                const float sd = (r & 1) ? MixF16<Op>::template one_minus_sq<1>(hw) : MixF16<Op>::template one_minus_sq<0>(hw);
                float dot = 0.0f;
#pragma unroll
                for (int s = 1; s < NS; ++s) {
                    const float hdb = comb(acc[s], accc[s], r);
                    dot = (r & 1) ? MixF16<Op>::template fma<1>(sp[s][r >> 1], hdb, dot) : MixF16<Op>::template fma<0>(sp[s][r >> 1], hdb, dot);
                    vals[s][0][r] = sd * hdb;
                }
                const float hd = (r & 1) ? MixF16<Op>::template fma<1>(hw, dot, 0.0f) : MixF16<Op>::template fma<0>(hw, dot, 0.0f);
                vals[0][0][r] = sd * comb(acc[0], accc[0], r) - 2.0f * hd;
            }
        } else {
            float st[NS][1][4];
            state_from_lds<MB>(rowS, st);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float h = st[0][0][r];
                const float sd = 1.0f - h * h;
                float dot = 0.0f;
#pragma unroll
                for (int s = 1; s < NS; ++s) {
                    const float hdb = comb(acc[s], accc[s], r);
                    dot += hdb * st[s][0][r];
                    vals[s][0][r] = sd * hdb;
                }
                vals[0][0][r] = sd * comb(acc[0], accc[0], r) - 2.0f * h * dot;
            }
        }
        CH::template emit<KS, MB>(Zn, vals, nullptr, WIDTH, x.c, x.q);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MB + 1 < WB) bwd_mb<MB + 1, KSB>(x, frag0, rowS, An, Zf, Zn);
    }

    // parked state S_l: asynchronous LDS-DMA of the scratch image into the parity buffer of layer l (no registers involved;
    // completion is covered by the vmcnt(0) of the next workgroup barrier)
    static __device__ __forceinline__ void dma_state(const Ctx& x, int l /*1..NL-1*/) {
        if constexpr (SLDS) return;
        char* dst = x.tenZ + TENSOR_Z_B + (l & 1) * SBUF_B;
#pragma unroll
        for (int i = 0; i < SBUF_B / 1024; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x.scr, (lds_void*)(dst + i * 1024), 16, x.lane16, (l - 1) * SBUF_B + i * 1024, 0, 0);
    }
    // forward: park feature block MB of S_l (hi parts) as rows of the scratch image
    template <int MB>
    static __device__ __forceinline__ void park_block(const Ctx& x, int l, const u32x4 (&Sf)[NS][1][KS][NP]) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const u32x2 v = {Sf[s][0][MB >> 1][0][(MB & 1) * 2 + 0], Sf[s][0][MB >> 1][0][(MB & 1) * 2 + 1]};
            __builtin_amdgcn_raw_buffer_store_b64(v, x.scr, x.rowoff, (l - 1) * SBUF_B + s * PANEL_B + 32 * MB, 0);
        }
    }
    template <int MB>
    static __device__ __forceinline__ void park_state(const Ctx& x, int l, const u32x4 (&Sf)[NS][1][KS][NP]) {
        if constexpr (SLDS) {
            if constexpr (MB == 0) put_tensor<KS, WB, 1>(x.rowS(l), Sf);      // the tile's own rows of slot l; nobody else touches them in the forward
        } else {
            park_block<MB>(x, l, Sf);
            if constexpr (MB + 1 < WB) park_state<MB + 1>(x, l, Sf);
        }
    }

    // Force the fragments to be fully computed at this point: without it the compiler sinks the reverse elementwise work past
    // the next workgroup barrier, where the weight-gradient waves can no longer overlap with it.
    template <int KSF>
    static __device__ __forceinline__ void pin(const u32x4 (&F)[NS][1][KSF][NP]) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int kk = 0; kk < KSF; ++kk)
#pragma unroll
                for (int p = 0; p < NP; ++p) asm volatile("" ::"v"(F[s][0][kk][p]));
    }

    // Weight layers L = NL-1 .. 1 (hidden-to-hidden) and finally L = 0, fully unrolled (static fragment indices / offsets).
    template <int L>
    struct Down {
        // entry: Zc = Z_L (adjoint of weight layer L's pre-activation) in chain fragment order; S_L is (being) DMA'd into its parity buffer
        static __device__ __forceinline__ void run(const FusedArgs& a, const Ctx& x, const float (&xin)[3], const u32x4 (&Zc)[NS][1][KS][NP]) {
            __syncthreads();                                   // previous layer's fragment reads are done
            fused_stamp(a, x.tracer, 3 + 3 * (NL - L));
            put_tensor<KS, WB, NP>(x.rowZ(), Zc);
            if constexpr (L == 0) {
                // S_0: the inputs as a 16-feature tensor (rows 0..2 = x', tangent stream k carries sx_k in row k)
                float v0[NS][1][4];
#pragma unroll
                for (int s = 0; s < NS; ++s)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // rows 0..2: hi parts; rows 4..6 (lanes q == 1): the 2^11-scaled low parts of the same numbers, so that the
                        // first layer's weight gradient keeps full input precision (combined at write-out)
                        float v = 0.0f;
                        if (x.q < 2 && r < 3) {
                            const float full = (s == 0) ? xin[r] : (r == s - 1 ? a.sx[r] : 0.0f);
                            v = x.q == 0 ? full : (full - round16<Op>(full)) * Op::LO_SCALE;
                        }
                        v0[s][0][r] = v;
                    }
                u32x4 S0[NS][1][1][NP];
                CH::template emit<1, 0>(S0, v0, nullptr, 16, x.c, x.q);
                put_tensor<1, 1, 1>(x.rowS(0), S0);
            }
            __syncthreads();                                   // tensors visible to the weight-gradient waves (also drains the S_L DMA)
            fused_stamp(a, x.tracer, 4 + 3 * (NL - L));
            if constexpr (L >= 1) {
                // reverse through W_L and the activation that produced S_L -> Z_{L-1}; meanwhile S_{L-1} streams into the other buffer
                u32x4 Zn[NS][1][KS][NP];
                {
                    const int frag0 = FI::bwd_mid(NL, L, 0, 0);
                    u32x4 A0[KS][NP];
                    load_afrags<KS>(x, frag0, A0);
                    if constexpr (L >= 2) dma_state(x, L - 1);
                    __builtin_amdgcn_sched_barrier(0);
                    bwd_mb<0, KS>(x, frag0, x.rowS(L), A0, Zc, Zn);
                    pin<KS>(Zn);
                }
                fused_stamp(a, x.tracer, 5 + 3 * (NL - L));
                Down<L - 1>::run(a, x, xin, Zn);
            }
        }
    };

    // forward (same arithmetic as chain_kernel) + output layer + residual head (net_f_sig INF:221-265) of the tile addressed by x:
    // parks S_1..S_{NL-1} in the tile's scratch image, returns S_NL (fragments) and the head's adjoint Z_NL, adds the loss sums
    static __device__ __forceinline__ void forward_tile(const FusedArgs& a, const Ctx& x, const float (&xin)[3], bool valid, long pidx, int set, float (&lsum)[8],
                                                        u32x4 (&B)[NS][1][KS][NP], u32x4 (&ZL)[NS][1][1][NP]) {
        const int c = x.c, q = x.q;
        first_mb<0>(a, x, xin, B);
        park_state<0>(x, 1, B);
        // two layers per loop trip, ping-ponging between B and B2: a one-buffer loop has to copy the 64 fragment registers
        // back at the end of every layer (48 v_mov per layer in the ISA)
        u32x4 B2[NS][1][KS][NP];
        auto layer = [&](int l, const u32x4 (&in)[NS][1][KS][NP], u32x4 (&out)[NS][1][KS][NP]) {
            const int frag0 = FI::fwd_mid(l, 0, 0);
            u32x4 A0[KS][NP];
            load_afrags<KS>(x, frag0, A0);
            fwd_mb<0>(x, frag0, (l - 1) * WIDTH * 4, A0, in, out);
            if (l + 1 < NL) park_state<0>(x, l + 1, out);      // S_NL is handled by the caller
        };
        int l = 1;
        for (; l + 1 < NL; l += 2) {
            layer(l, B, B2);
            layer(l + 1, B2, B);
        }
        if (l < NL) {
            layer(l, B, B2);
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int kk = 0; kk < KS; ++kk)
#pragma unroll
                    for (int pp = 0; pp < NP; ++pp) B[s][0][kk][pp] = B2[s][0][kk][pp];
        }
        fused_stamp(a, x.tracer, 1);
        f32x4 yacc[NS], yaccc[NS];
        {
            u32x4 A0[KS][NP];
            load_afrags<KS>(x, FI::fwd_last(NL, 0), A0);
            gemm_pre<KS>(A0, B, yacc, yaccc);
        }
        const f32x4 bl = *reinterpret_cast<const f32x4*>(a.pw.bias_last + 4 * q);
        float Y[NS][8];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float own = comb(yacc[s], yaccc[s], r) + (s == 0 ? bl[r] : 0.0f);
                const float oth = __shfl_xor(own, 16);
                Y[s][r] = (q & 1) ? oth : own;
                Y[s][4 + r] = (q & 1) ? own : oth;
            }
        const float vm = valid ? 1.0f : 0.0f;
        float adj[NS][8];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int o = 0; o < 8; ++o) adj[s][o] = 0.0f;
        if constexpr (NS == 4) {
            const float e11 = Y[1][0], e22 = Y[2][1], e12 = Y[2][0] + Y[1][1];
            float f[7];
            f[0] = Y[1][4] + Y[2][6] - a.rho * Y[3][2];
            f[1] = Y[2][5] + Y[1][6] - a.rho * Y[3][3];
            f[2] = Y[3][0] - Y[0][2];
            f[3] = Y[3][1] - Y[0][3];
            f[4] = Y[0][4] - (a.c1 * e11 + a.c2 * e22);
            f[5] = Y[0][5] - (a.c2 * e11 + a.c1 * e22);
            f[6] = Y[0][6] - a.G * e12;
            float g[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                if (q == 0) lsum[i] += vm * f[i] * f[i];
                g[i] = 2.0f * a.tw[i] * f[i] * vm;
            }
            adj[0][2] = -g[2];
            adj[0][3] = -g[3];
            adj[0][4] = g[4];
            adj[0][5] = g[5];
            adj[0][6] = g[6];
            adj[1][0] = -a.c1 * g[4] - a.c2 * g[5];
            adj[1][1] = -a.G * g[6];
            adj[1][4] = g[0];
            adj[1][6] = g[1];
            adj[2][0] = -a.G * g[6];
            adj[2][1] = -a.c2 * g[4] - a.c1 * g[5];
            adj[2][5] = g[1];
            adj[2][6] = g[0];
            adj[3][0] = g[2];
            adj[3][1] = g[3];
            adj[3][2] = -a.rho * g[0];
            adj[3][3] = -a.rho * g[1];
        } else {
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                float d = 0.0f;
                const float* tg = a.set_targets[set];
                if (o < a.net.nout) d = Y[0][o] - (tg ? tg[(long)o * a.set_n[set] + pidx] : 0.0f);
                if (q == 0) lsum[o] += vm * d * d;
                adj[0][o] = 2.0f * a.set_tw[set][o] * d * vm;
            }
        }
        float vals[NS][1][4];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) vals[s][0][r] = q < 2 ? ((q & 1) ? adj[s][4 + r] : adj[s][r]) : 0.0f;
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int pp = 0; pp < NP; ++pp) ZL[s][0][0][pp] = u32x4{0u, 0u, 0u, 0u};
        CH::template emit<1, 0>(ZL, vals, nullptr, 16, c, q);
    }

    // reverse of one tile through all weight layers, in step with the weight-gradient waves; B holds S_NL
    static __device__ __forceinline__ void reverse_tile(const FusedArgs& a, const Ctx& x, const float (&xin)[3], const u32x4 (&B)[NS][1][KS][NP],
                                                        const u32x4 (&ZL)[NS][1][1][NP]) {
        // ---- top weight layer NL: hand Z_NL (16 outputs) and S_NL over, then reverse into the hidden chain
        fused_stamp(a, x.tracer, 2);
        __syncthreads();
        fused_stamp(a, x.tracer, 3);
        put_tensor<1, 1, NP>(x.rowZ(), ZL);
        put_tensor<KS, WB, 1>(x.rowS(NL), B);
        __syncthreads();
        fused_stamp(a, x.tracer, 4);
        u32x4 Zn[NS][1][KS][NP];
        {
            const int frag0 = FI::bwd_last(NL, 0);
            u32x4 A0[1][NP];
            load_afrags<1>(x, frag0, A0);
            dma_state(x, NL - 1);
            __builtin_amdgcn_sched_barrier(0);
            bwd_mb<0, 1>(x, frag0, x.rowS(NL), A0, ZL, Zn);
            pin<KS>(Zn);
        }
        fused_stamp(a, x.tracer, 5);
        Down<NL - 1>::run(a, x, xin, Zn);
    }

    static __device__ __forceinline__ void load_inputs(const FusedArgs& a, const float* px, const float* py, const float* pt, long n, long tile, int c,
                                                       float (&xin)[3], bool& valid, long& pidx) {
        const long p = tile * 16 + c;
        valid = p < n;
        pidx = valid ? p : n - 1;
        xin[0] = px[pidx] * a.sx[0] + a.ox[0];
        xin[1] = py[pidx] * a.sx[1] + a.ox[1];
        xin[2] = pt[pidx] * a.sx[2] + a.ox[2];
    }

    static __device__ __forceinline__ void chain_role(const FusedArgs& a, char* lds, int wave, int lane, int c, int q) {
        const long gwave = (long)blockIdx.x * TILES + wave;
        Ctx x;
        x.init(a, lds, wave, lane, c, q);
        x.set_tile(a, gwave);
        x.tracer = blockIdx.x == 0 && wave == 0 && lane == 0;
        constexpr int NSETS = NS == 1 ? FUSED_MAX_SETS : 1;
        float lsum[NSETS][8];
#pragma unroll
        for (int k = 0; k < NSETS; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) lsum[k][i] = 0.0f;

        for (long step = blockIdx.x; step < a.nsteps; step += gridDim.x) {
            float xin[3];
            bool valid;
            long pidx;
            int set = 0;
            if constexpr (NS == 1) {
#pragma unroll
                for (int k = 1; k < FUSED_MAX_SETS; ++k)
                    if (k < a.nsets && step >= a.set_step0[k]) set = k;
                load_inputs(a, a.set_x[set], a.set_y[set], a.set_t[set], a.set_n[set], (step - a.set_step0[set]) * TILES + wave, c, xin, valid, pidx);
            } else {
                load_inputs(a, a.x, a.y, a.t, a.n, step * TILES + wave, c, xin, valid, pidx);
            }
            x.tracer = blockIdx.x == 0 && wave == 0 && lane == 0 && step == 2 * (long)gridDim.x;      // a steady-state step
            fused_stamp(a, x.tracer, 0);
            {
                u32x4 B[NS][1][KS][NP], ZL[NS][1][1][NP];
                float ls[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) ls[i] = 0.0f;
                forward_tile(a, x, xin, valid, pidx, set, ls, B, ZL);
#pragma unroll
                for (int k = 0; k < NSETS; ++k)
#pragma unroll
                    for (int i = 0; i < 8; ++i) lsum[k][i] += (k == set) ? ls[i] : 0.0f;
                reverse_tile(a, x, xin, B, ZL);
            }
        }
#pragma unroll
        for (int k = 0; k < NSETS; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float v = lsum[k][i];
                v += __shfl_xor(v, 1);
                v += __shfl_xor(v, 2);
                v += __shfl_xor(v, 4);
                v += __shfl_xor(v, 8);
                if (lane == 0) a.loss_part[(gwave * NSETS + k) * 8 + i] = v;
            }
    }

    static __device__ void run(const FusedArgs& a) {
        __shared__ __attribute__((aligned(16))) char lds[LDS_B];
        const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
        const int wave8 = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // provably wave-uniform
        if (wave8 >= 4) {
            wgrad_role(a, lds, wave8 - 4, c, q);
        } else {
            __builtin_amdgcn_s_setprio(2);          // the chain wave is the critical path of its SIMD: issue it first
            chain_role(a, lds, wave8, lane, c, q);
        }
    }
};

template <class Op, int SPLIT, int WIDTH, int NL, int NS>
__global__ __launch_bounds__(512) void fused_wave_kernel(const FusedArgs a) {
    Fused<Op, SPLIT, WIDTH, NL, NS>::run(a);
}

}  // namespace pinn
