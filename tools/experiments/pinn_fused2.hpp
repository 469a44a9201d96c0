// SHELVED EXPERIMENT (not part of the build): measured 7.4 ms vs 6.5 ms for pinn_fused.hpp on the 2M-point 8x64 f16x3 launch
// (MI355X, tools/exp_run.py); kept for the record of what was tried -- see DESIGN.md section 6.  To build it again, include it from
// pinn_host.hpp and launch fused2_wave_kernel in Host::fused_launch.
// Fused loss+gradient kernel, second generation ("pair" layout) for narrow nets (padded width <= 64).
//
// Measured on the first fused kernel (pinn_fused.hpp, tools/phase_trace.py): a lone wave issues about one instruction per
// 4-5 cycles however the stream is scheduled, so the 16-point chain of ~650 instructions per layer is ISSUE-latency bound,
// and during the forward half its four weight-gradient waves had nothing to do.  Here all eight waves of a workgroup are
// symmetric and busy in every phase:
//   * a workgroup step still covers 4 tiles of 16 points, but each tile is shared by a PAIR of waves (w, w+4) that split
//     every layer's output features in halves.  The layer state therefore lives in LDS as [stream][point][feature] rows
//     (the very image the weight gradient needs); each wave rebuilds the MFMA B operand of a layer with ds_read_b64 and
//     writes its half of the next state with ds_write_b64.  Per wave and layer: half the MFMAs and half the tanh / split
//     arithmetic of the old chain wave, i.e. half the latency, with two such waves per SIMD to fill each other's gaps;
//   * the weight gradient is spread over all eight waves as well (16 blocks of a 64x64 Wbar -> 2 per wave, persistent MFMA
//     accumulators, 72 registers), read from the same LDS tensors with ds_read_b64_tr_b16;
//   * forward: ONE workgroup barrier per layer (state ping-pongs between two LDS regions); reverse: two (adjoint tensor is
//     single-buffered).  Parked forward state (fp16 hi parts) goes to the per-tile scratch image and returns by LDS-DMA one
//     layer ahead of its use, as before.
// LDS per tile: R0 = [part][stream] panels (adjoint tensor Z in reverse / state of odd distance from the top in forward),
//               R1 = two S buffers (reverse) / state of even distance from the top (forward; its hi panels ARE S buffer 0,
//               so the last hidden state needs no parking).
#pragma once
#include "pinn_fused.hpp"

#ifndef PINN_FUSED2_STAGGER
#define PINN_FUSED2_STAGGER 0
#endif

namespace pinn {

template <class Op, int SPLIT, int WIDTH, int NL>
struct Fused2 {
    static constexpr int NS = 4, WB = WIDTH / 16, KS = WIDTH / 32, NP = SPLIT == 3 ? 2 : 1, NPS = NP;
    static constexpr int HB = WB / 2;                          // feature blocks per wave of a pair
    static constexpr int NOBW = WB == 4 ? 2 : 1;               // weight-gradient out-blocks per wave (mid layers)
    static_assert(WB == 2 || WB == 4, "fused kernel supports padded widths 32 and 64");
    static_assert(NL >= 2 && (NL & 1) == 0, "compiled for an even number of hidden layers");
    static constexpr float INV_LS = 1.0f / Op::LO_SCALE;
    typedef FragIndex<WIDTH> FI;
    static constexpr int ROWB = WIDTH * 2 + 8;
    static constexpr int PANEL_B = 16 * ROWB;
    static constexpr int TENSOR_Z_B = NS * NP * PANEL_B;
    static constexpr int TENSOR_S_B = NS * PANEL_B;
    static constexpr int SBUF_B = (TENSOR_S_B + 1023) / 1024 * 1024;
    static constexpr int R1 = TENSOR_Z_B;
    static constexpr int TILE_B = TENSOR_Z_B + 2 * SBUF_B;
    static_assert(2 * SBUF_B >= TENSOR_Z_B, "forward state must fit the S double buffer");
    static constexpr int LDS_B = 4 * TILE_B;
    static_assert(LDS_B <= 160 * 1024, "LDS budget");
    static constexpr int NCHUNK_DMA = SBUF_B / 1024;
    static constexpr unsigned SCRATCH_BYTES = (unsigned)((NL - 1) * SBUF_B);   // per tile

    static constexpr int xreg(int l) { return ((NL - l) & 1) ? 0 : R1; }                 // forward state S_l (l = 1..NL)
    static constexpr int sbuf(int L) { return R1 + ((NL - L) & 1) * SBUF_B; }          // reverse: S_L (hi panels)

    struct Acc {                       // persistent across the whole launch, all statically indexed
        f32x4 mid[NL - 1][NOBW];       // Wbar_l (l = 1..NL-1) blocks (ib, ob0 + o)
        f32x4 edge;                    // waves 0..WB-1: Wbar_0 block (0, wave); waves 4..4+WB-1: Wbar_NL block (wave-4, 0)
        float bias[NL + 1][NOBW];
    };

    // ---------------------------------------------------------------------------------------------
    // weight gradient (every wave owns a few blocks of every Wbar)
    // ---------------------------------------------------------------------------------------------
    static __device__ __forceinline__ u32x4 get_frag(const char* base, int off) {
        const v4i16 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)(base + off));
        const v4i16 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)(base + off + 4 * ROWB));
        const u32x2 d0 = __builtin_bit_cast(u32x2, v0), d1 = __builtin_bit_cast(u32x2, v1);
        return u32x4{d0[0], d0[1], d1[0], d1[1]};
    }

    // acc[b] += sum over the 64 points of the step and the 4 streams of  S(block at sbase)^T . Z(block at zbase + 32 b)
    template <int NBK>
    static __device__ __forceinline__ void wg_blocks(const char* sbase, const char* zbase, f32x4 (&acc)[NBK], float (&bias_out)[NBK], bool want_bias) {
        f32x4 cc[NBK], bm[NBK], bc[NBK];
#pragma unroll
        for (int b = 0; b < NBK; ++b) {
            bm[b] = f32x4{0.f, 0.f, 0.f, 0.f};
            bc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
            cc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const uint32_t one2 = pack2<Op>(1.0f, 1.0f);
        const u32x4 ones = {one2, one2, one2, one2};
        struct Frags { u32x4 Ah, Bh[NBK], Bl[NBK]; };
        auto fetch = [&](int g, Frags& f) {
            const int j = g >> 2, st = g & 3;
            f.Ah = get_frag(sbase, 2 * j * TILE_B + st * PANEL_B);
#pragma unroll
            for (int b = 0; b < NBK; ++b) {
                f.Bh[b] = get_frag(zbase, 2 * j * TILE_B + st * PANEL_B + 32 * b);
                if (NP == 2) f.Bl[b] = get_frag(zbase, 2 * j * TILE_B + (NS + st) * PANEL_B + 32 * b);
            }
        };
        Frags cur, nxt;
        fetch(0, cur);
#pragma unroll
        for (int g = 0; g < 2 * NS; ++g) {
            if (g + 1 < 2 * NS) fetch(g + 1, nxt);
#pragma unroll
            for (int b = 0; b < NBK; ++b) {
                acc[b] = Op::mfma(cur.Ah, cur.Bh[b], acc[b]);
                if (NP == 2) cc[b] = Op::mfma(cur.Ah, cur.Bl[b], cc[b]);
            }
            if ((g & 3) == 0 && want_bias) {                   // bias gradient = ones^T . Z (value stream)
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
                for (int r = 0; r < 4; ++r) acc[b][r] += cc[b][r] * INV_LS;
            }
        }
    }

    static __device__ __forceinline__ bool has_mid(int wave) { return WB == 4 ? true : wave < 4; }
    static __device__ __forceinline__ int mid_ib(int wave) { return WB == 4 ? (wave & 3) : (wave & 1); }
    static __device__ __forceinline__ int mid_ob0(int wave) { return WB == 4 ? 2 * (wave >> 2) : ((wave >> 1) & 1); }

    template <int L>
    static __device__ __forceinline__ void wgrad(const char* lanebase, Acc& A, int wave) {
        const char* zl = lanebase;                          // Z tensor of tile 0 (R0)
        const char* sl = lanebase + sbuf(L);                // S_L of tile 0
        if constexpr (L == 0) {
            if (wave < WB) {
                f32x4 t[1] = {A.edge};
                float b[1];
                wg_blocks<1>(sl, zl + 32 * wave, t, b, true);
                A.edge = t[0];
                A.bias[0][0] += b[0];
            }
        } else if constexpr (L == NL) {
            if (wave >= 4 && wave - 4 < WB) {
                f32x4 t[1] = {A.edge};
                float b[1];
                wg_blocks<1>(sl + 32 * (wave - 4), zl, t, b, wave == 4);
                A.edge = t[0];
                if (wave == 4) A.bias[NL][0] += b[0];
            }
        } else {
            if (has_mid(wave)) {
                float b[NOBW];
                const int ib = mid_ib(wave);
                wg_blocks<NOBW>(sl + 32 * ib, zl + 32 * mid_ob0(wave), A.mid[L - 1], b, ib == 0);
                if (ib == 0) {
#pragma unroll
                    for (int o = 0; o < NOBW; ++o) A.bias[L][o] += b[o];
                }
            }
        }
    }

    // ---------------------------------------------------------------------------------------------
    // chain (each wave: half the feature blocks of its tile)
    // ---------------------------------------------------------------------------------------------
    struct Ctx {
        __amdgpu_buffer_rsrc_t frags, scr, bias, w0p;
        unsigned lane16;                           // lane * 16
        unsigned rowoff;                           // c*ROWB + 8q
        char* tile;                                // tile's LDS base (uniform)
        char* row;                                 // tile + rowoff
        int c, q, half, mb0;                       // mb0 = half * HB: first own feature block
        bool tracer;
    };

    // MFMA B operand (state / adjoint of all features, this lane's point) from [part][stream] panels
    template <int KSB>
    static __device__ __forceinline__ void read_frags(const char* row, u32x4 (&B)[NS][KSB][NP]) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int kk = 0; kk < KSB; ++kk) {
                    const char* a = row + (p * NS + s) * PANEL_B + 64 * kk;
                    const u32x2 d0 = *reinterpret_cast<const u32x2*>(a), d1 = *reinterpret_cast<const u32x2*>(a + 32);
                    B[s][kk][p] = u32x4{d0[0], d0[1], d1[0], d1[1]};
                }
    }
    template <int KSB>
    static __device__ __forceinline__ void load_afrags(const Ctx& x, int frag0, u32x4 (&Af)[KSB][NP]) {
#pragma unroll
        for (int kk = 0; kk < KSB; ++kk)
#pragma unroll
            for (int p = 0; p < NP; ++p) Af[kk][p] = __builtin_amdgcn_raw_buffer_load_b128(x.frags, x.lane16, ((frag0 + kk) * NPS + p) * 1024, 0);
    }
    template <int KSB>
    static __device__ __forceinline__ void gemm(const u32x4 (&Af)[KSB][NP], const u32x4 (&B)[NS][KSB][NP], f32x4 (&acc)[NS], f32x4 (&accc)[NS]) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            f32x4 m = {0.f, 0.f, 0.f, 0.f}, cc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KSB; ++kk) {
                m = Op::mfma(Af[kk][0], B[s][kk][0], m);
                if (NP == 2) {
                    cc = Op::mfma(Af[kk][0], B[s][kk][1], cc);
                    cc = Op::mfma(Af[kk][1], B[s][kk][0], cc);
                }
            }
            acc[s] = m;
            accc[s] = cc;
        }
    }
    static __device__ __forceinline__ float comb(const f32x4& m, const f32x4& cc, int r) { return NP == 2 ? m[r] + cc[r] * INV_LS : m[r]; }

    // four features x NS streams of this lane's point -> packed 16-bit rows (hi, and the scaled low part when split)
    static __device__ __forceinline__ void pack_block(const float (&vals)[NS][4], u32x2 (&hi)[NS], u32x2 (&lo)[NS]) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (NP == 2) {
                uint32_t h0, h1, l0, l1;
                split2<Op>(vals[s][0], vals[s][1], h0, l0);
                split2<Op>(vals[s][2], vals[s][3], h1, l1);
                hi[s] = u32x2{h0, h1};
                lo[s] = u32x2{l0, l1};
            } else {
                hi[s] = u32x2{pack2<Op>(vals[s][0], vals[s][1]), pack2<Op>(vals[s][2], vals[s][3])};
                lo[s] = u32x2{0u, 0u};
            }
        }
    }
    // rows of feature block `boff` (= 32 * block, bytes) in a [part][stream] tensor at `row`
    static __device__ __forceinline__ void write_block(char* row, int boff, const u32x2 (&hi)[NS], const u32x2 (&lo)[NS], bool with_lo) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            *reinterpret_cast<u32x2*>(row + s * PANEL_B + boff) = hi[s];
            if (NP == 2 && with_lo) *reinterpret_cast<u32x2*>(row + (NS + s) * PANEL_B + boff) = lo[s];
        }
    }
    static __device__ __forceinline__ void park_block(const Ctx& x, int l /*state S_l, 1..NL-1*/, int mb, const u32x2 (&hi)[NS]) {
#pragma unroll
        for (int s = 0; s < NS; ++s) __builtin_amdgcn_raw_buffer_store_b64(hi[s], x.scr, x.rowoff, (l - 1) * SBUF_B + 32 * mb + s * PANEL_B, 0);
    }
    static __device__ __forceinline__ void read_state(const char* rowS, int boff, float (&st)[NS][4]) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const u32x2 h = *reinterpret_cast<const u32x2*>(rowS + s * PANEL_B + boff);
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                st[s][2 * d + 0] = cvt16<Op>((uint16_t)(h[d] & 0xffffu));
                st[s][2 * d + 1] = cvt16<Op>((uint16_t)(h[d] >> 16));
            }
        }
    }
    // this wave's share of the LDS-DMA that brings parked S_l (l = 1..NL-1) back into its reverse buffer
    static __device__ __forceinline__ void dma_state(const Ctx& x, int l, int dst_off) {
#pragma unroll
        for (int i = 0; i < NCHUNK_DMA; i += 2) {
            const int ii = i + x.half;
            if (ii < NCHUNK_DMA)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x.scr, (lds_void*)(x.tile + dst_off + ii * 1024), 16, x.lane16, (l - 1) * SBUF_B + ii * 1024, 0, 0);
        }
    }

    // reverse of (h = tanh z, hdot_k = (1-h^2) zdot_k) for one feature block
    static __device__ __forceinline__ void act_bwd(const f32x4 (&acc)[NS], const f32x4 (&accc)[NS], const float (&st)[NS][4], float (&vals)[NS][4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float h = st[0][r];
            const float sd = 1.0f - h * h;
            float dot = 0.0f;
#pragma unroll
            for (int s = 1; s < NS; ++s) {
                const float hdb = comb(acc[s], accc[s], r);
                dot += hdb * st[s][r];
                vals[s][r] = sd * hdb;
            }
            vals[0][r] = sd * comb(acc[0], accc[0], r) - 2.0f * h * dot;
        }
    }

    // own feature blocks of Z_{L-1} from the B operand Zf of weight layer L (fragments at frag0 + block * KSB) and state S_L
    template <int KSB>
    static __device__ __forceinline__ void reverse_blocks(const Ctx& x, int frag_first, const char* rowS, const u32x4 (&Zf)[NS][KSB][NP],
                                                          u32x2 (&zh)[HB][NS], u32x2 (&zl)[HB][NS]) {
        u32x4 Af[KSB][NP];
        load_afrags<KSB>(x, frag_first, Af);
#pragma unroll
        for (int hb = 0; hb < HB; ++hb) {
            u32x4 An[KSB][NP];
            if (hb + 1 < HB) load_afrags<KSB>(x, frag_first + (hb + 1) * KSB, An);
            f32x4 acc[NS], accc[NS];
            gemm<KSB>(Af, Zf, acc, accc);
            float st[NS][4], vals[NS][4];
            read_state(rowS, 32 * hb, st);
            act_bwd(acc, accc, st, vals);
            pack_block(vals, zh[hb], zl[hb]);
            __builtin_amdgcn_sched_barrier(0);
            if (hb + 1 < HB) {
#pragma unroll
                for (int kk = 0; kk < KSB; ++kk)
#pragma unroll
                    for (int p = 0; p < NP; ++p) Af[kk][p] = An[kk][p];
            }
        }
    }

    // Weight layers L = NL-1 .. 0, fully unrolled.  Entry: own blocks of Z_L in registers (zh, zl); wgrad of layer L+1 done.
    template <int L>
    struct Down {
        static __device__ __forceinline__ void run(const FusedArgs& a, const Ctx& x, const char* lanebase, Acc& A, int wave, const float (&xin)[3],
                                                   const u32x2 (&zh)[HB][NS], const u32x2 (&zl)[HB][NS]) {
            __syncthreads();                                   // everybody is done with Z_{L+1} and S_{L+1}
            fused_stamp(a, x.tracer, 3 + 3 * (NL - L));
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) write_block(x.row + 32 * x.mb0, 32 * hb, zh[hb], zl[hb], true);
            if constexpr (L == 0) {
                if (x.half == 0) {
                    // S_0: the inputs as a 16-feature tensor (rows 0..2 = x', tangent stream k carries sx_k in row k); rows 4..6
                    // (lanes q == 1) hold the 2^11-scaled low parts so that Wbar_0 keeps full input precision (combined at write-out)
                    u32x2 h0[NS], l0[NS];
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float t = 0.0f;
                            if (x.q < 2 && r < 3) {
                                const float full = (s == 0) ? xin[r] : (r == s - 1 ? a.sx[r] : 0.0f);
                                t = x.q == 0 ? full : (full - round16<Op>(full)) * Op::LO_SCALE;
                            }
                            v[r] = t;
                        }
                        h0[s] = u32x2{pack2<Op>(v[0], v[1]), pack2<Op>(v[2], v[3])};
                        l0[s] = u32x2{0u, 0u};
                    }
                    write_block(x.row + sbuf(0), 0, h0, l0, false);
                }
            }
            __syncthreads();                                   // tensors of layer L complete (also drains the S_L DMA)
            fused_stamp(a, x.tracer, 4 + 3 * (NL - L));
            if constexpr (L >= 1) {
                if constexpr (L >= 2) dma_state(x, L - 1, sbuf(L - 1));
                u32x2 nh[HB][NS], nl[HB][NS];
#if PINN_FUSED2_STAGGER
                if (x.half) wgrad<L>(lanebase, A, wave);      // the two waves of a SIMD run their matrix-heavy and VALU-heavy parts out of phase
#endif
                {
                    u32x4 Zf[NS][KS][NP];
                    read_frags<KS>(x.row, Zf);
                    reverse_blocks<KS>(x, FI::bwd_mid(NL, L, 0, 0) + x.mb0 * KS, x.row + sbuf(L) + 32 * x.mb0, Zf, nh, nl);
                }
#if PINN_FUSED2_STAGGER
                if (!x.half) wgrad<L>(lanebase, A, wave);
#else
                wgrad<L>(lanebase, A, wave);
#endif
                fused_stamp(a, x.tracer, 5 + 3 * (NL - L));
                Down<L - 1>::run(a, x, lanebase, A, wave, xin, nh, nl);
            } else {
                wgrad<0>(lanebase, A, wave);
                fused_stamp(a, x.tracer, 5 + 3 * NL);
            }
        }
    };

    static __device__ void run(const FusedArgs& a) {
        __shared__ __attribute__((aligned(16))) char lds[LDS_B];
        const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
        const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        const int tile = wave & 3;
        const long gtile = (long)blockIdx.x * 4 + tile;
        Ctx x;
        x.frags = __builtin_amdgcn_make_buffer_rsrc((void*)a.pw.frags, 0, (int)a.frags_bytes, 0x00020000);
        x.scr = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<char*>(a.scratch) + gtile * (long)SCRATCH_BYTES), 0, (int)SCRATCH_BYTES,
                                                  0x00020000);
        x.bias = __builtin_amdgcn_make_buffer_rsrc((void*)a.pw.bias_mid, 0, (NL - 1) * WIDTH * 4, 0x00020000);
        x.w0p = __builtin_amdgcn_make_buffer_rsrc((void*)a.pw.w0p, 0, WIDTH * 16, 0x00020000);
        x.lane16 = (unsigned)lane * 16u;
        x.rowoff = (unsigned)(c * ROWB + 8 * q);
        x.tile = lds + tile * TILE_B;
        x.row = x.tile + x.rowoff;
        x.c = c;
        x.q = q;
        x.half = wave >> 2;
        x.mb0 = x.half * HB;
        x.tracer = blockIdx.x == 0 && wave == 0 && lane == 0;
        const char* lanebase = lds + (q >> 1) * TILE_B + (8 * (q & 1) + (c >> 2)) * ROWB + 8 * (c & 3);

        Acc A;
#pragma unroll
        for (int l = 0; l < NL - 1; ++l)
#pragma unroll
            for (int o = 0; o < NOBW; ++o) A.mid[l][o] = f32x4{0.f, 0.f, 0.f, 0.f};
        A.edge = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int l = 0; l <= NL; ++l)
#pragma unroll
            for (int o = 0; o < NOBW; ++o) A.bias[l][o] = 0.0f;
        float lsum[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) lsum[i] = 0.0f;

        for (long step = blockIdx.x; step < a.nsteps; step += gridDim.x) {
            float xin[3];
            const long p = (step * 4 + tile) * 16 + c;
            const bool valid = p < a.n;
            const long pidx = valid ? p : a.n - 1;
            xin[0] = a.x[pidx] * a.sx[0] + a.ox[0];
            xin[1] = a.y[pidx] * a.sx[1] + a.ox[1];
            xin[2] = a.t[pidx] * a.sx[2] + a.ox[2];
            __syncthreads();                                   // previous step's tensors are dead
            fused_stamp(a, x.tracer, 0);
            // ---- forward, first layer (K = 3, VALU): INF:191-195 with the tangent seeds e_k * sx_k; own blocks of S_1
            {
                char* dst = x.row + xreg(1) + 32 * x.mb0;
#pragma unroll
                for (int hb = 0; hb < HB; ++hb) {
                    float vals[NS][4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const u32x4 wu = __builtin_amdgcn_raw_buffer_load_b128(x.w0p, (unsigned)q * 64u, (16 * (x.mb0 + hb) + r) * 16, 0);
                        const f32x4 w = __builtin_bit_cast(f32x4, wu);
                        float h, sd;
                        tanh_act(w[3] + w[0] * xin[0] + w[1] * xin[1] + w[2] * xin[2], h, sd);
                        vals[0][r] = h;
#pragma unroll
                        for (int s = 1; s < NS; ++s) vals[s][r] = sd * (a.sx[s - 1] * w[s - 1]);
                    }
                    u32x2 hi[NS], lo[NS];
                    pack_block(vals, hi, lo);
                    write_block(dst, 32 * hb, hi, lo, true);
                    park_block(x, 1, x.mb0 + hb, hi);
                }
            }
            __syncthreads();
            // ---- forward, hidden layers: S_{l+1} = tanh(W_l S_l + b_l) with tangents; own blocks
            for (int l = 1; l < NL; ++l) {
                const int sreg = ((NL - l) & 1) ? 0 : R1, dreg = ((NL - l) & 1) ? R1 : 0;
                u32x4 B[NS][KS][NP];
                read_frags<KS>(x.row + sreg, B);
                char* dst = x.row + dreg + 32 * x.mb0;
                const int frag_first = FI::fwd_mid(l, 0, 0) + x.mb0 * KS;
                u32x4 Af[KS][NP];
                load_afrags<KS>(x, frag_first, Af);
#pragma unroll
                for (int hb = 0; hb < HB; ++hb) {
                    u32x4 An[KS][NP];
                    if (hb + 1 < HB) load_afrags<KS>(x, frag_first + (hb + 1) * KS, An);
                    const u32x4 bu = __builtin_amdgcn_raw_buffer_load_b128(x.bias, (unsigned)q * 16u, ((l - 1) * WIDTH + 16 * (x.mb0 + hb)) * 4, 0);
                    const f32x4 bias = __builtin_bit_cast(f32x4, bu);
                    f32x4 acc[NS], accc[NS];
                    gemm<KS>(Af, B, acc, accc);
                    float vals[NS][4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float h, sd;
                        tanh_act(comb(acc[0], accc[0], r) + bias[r], h, sd);
                        vals[0][r] = h;
#pragma unroll
                        for (int s = 1; s < NS; ++s) vals[s][r] = sd * comb(acc[s], accc[s], r);
                    }
                    u32x2 hi[NS], lo[NS];
                    pack_block(vals, hi, lo);
                    write_block(dst, 32 * hb, hi, lo, true);
                    if (l + 1 < NL) park_block(x, l + 1, x.mb0 + hb, hi);          // S_NL stays in LDS (R1 = S buffer 0)
                    __builtin_amdgcn_sched_barrier(0);
                    if (hb + 1 < HB) {
#pragma unroll
                        for (int kk = 0; kk < KS; ++kk)
#pragma unroll
                            for (int pp = 0; pp < NP; ++pp) Af[kk][pp] = An[kk][pp];
                    }
                }
                __syncthreads();
            }
            fused_stamp(a, x.tracer, 1);
            // ---- output layer + residual head (net_f_sig INF:221-265); both waves of the pair evaluate it
            u32x4 ZL[NS][1][NP];
            {
                f32x4 yacc[NS], yaccc[NS];
                {
                    u32x4 B[NS][KS][NP];
                    read_frags<KS>(x.row + xreg(NL), B);
                    u32x4 A0[KS][NP];
                    load_afrags<KS>(x, FI::fwd_last(NL, 0), A0);
                    gemm<KS>(A0, B, yacc, yaccc);
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
                    if (q == 0 && x.half == 0) lsum[i] += vm * f[i] * f[i];
                    g[i] = 2.0f * a.tw[i] * f[i] * vm;
                }
                float adj[NS][8];
#pragma unroll
                for (int s = 0; s < NS; ++s)
#pragma unroll
                    for (int o = 0; o < 8; ++o) adj[s][o] = 0.0f;
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
                float vals[NS][4];
#pragma unroll
                for (int s = 0; s < NS; ++s)
#pragma unroll
                    for (int r = 0; r < 4; ++r) vals[s][r] = q < 2 ? ((q & 1) ? adj[s][4 + r] : adj[s][r]) : 0.0f;
                u32x2 zh[NS], zl[NS];
                pack_block(vals, zh, zl);
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    ZL[s][0][0] = u32x4{zh[s][0], zh[s][1], 0u, 0u};
                    if (NP == 2) ZL[s][0][NP - 1] = u32x4{zl[s][0], zl[s][1], 0u, 0u};
                }
                // ---- top weight layer NL: publish Z_NL (16 outputs); S_NL already sits in S buffer 0
                fused_stamp(a, x.tracer, 2);
                __syncthreads();                               // all reads of the forward state (incl. S_NL low parts) are done
                fused_stamp(a, x.tracer, 3);
                if (x.half == 0) write_block(x.row, 0, zh, zl, true);
            }
            __syncthreads();
            fused_stamp(a, x.tracer, 4);
            u32x2 nh[HB][NS], nl[HB][NS];
            dma_state(x, NL - 1, sbuf(NL - 1));
            reverse_blocks<1>(x, FI::bwd_last(NL, 0) + x.mb0, x.row + sbuf(NL) + 32 * x.mb0, ZL, nh, nl);
            wgrad<NL>(lanebase, A, wave);
            fused_stamp(a, x.tracer, 5);
            Down<NL - 1>::run(a, x, lanebase, A, wave, xin, nh, nl);
        }
        if (x.half == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float v = lsum[i];
                v += __shfl_xor(v, 1);
                v += __shfl_xor(v, 2);
                v += __shfl_xor(v, 4);
                v += __shfl_xor(v, 8);
                if (lane == 0) a.loss_part[gtile * 8 + i] = v;
            }
        }
        // ---- this workgroup's partial gradient
        float* part = a.partial + (long)blockIdx.x * a.net.nparams;
        const int H = a.net.h, NO = a.net.nout;
        auto put_block = [&](const f32x4& v, int l, int ib, int ob, int n_in, int n_out) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int in = 16 * ib + 4 * q + r, out = 16 * ob + c;
                if (in < n_in && out < n_out) part[a.net.w_off[l] + in * n_out + out] = v[r];
            }
        };
        if (wave < WB) {
            // Wbar_0 rows 4..6 hold the contribution of the inputs' low parts (see Down<0>): fold them into rows 0..2
            f32x4 lo;
#pragma unroll
            for (int r = 0; r < 4; ++r) lo[r] = __shfl_xor(A.edge[r], 16);
#pragma unroll
            for (int r = 0; r < 4; ++r) A.edge[r] += lo[r] * INV_LS;
            put_block(A.edge, 0, 0, wave, 3, H);
            if (q == 0 && 16 * wave + c < H) part[a.net.b_off[0] + 16 * wave + c] = A.bias[0][0];
        }
        if (wave >= 4 && wave - 4 < WB) put_block(A.edge, NL, wave - 4, 0, H, NO);
        if (wave == 4 && q == 0 && c < NO) part[a.net.b_off[NL] + c] = A.bias[NL][0];
        if (has_mid(wave)) {
            const int ib = mid_ib(wave), ob0 = mid_ob0(wave);
#pragma unroll
            for (int l = 1; l < NL; ++l)
#pragma unroll
                for (int o = 0; o < NOBW; ++o) {
                    put_block(A.mid[l - 1][o], l, ib, ob0 + o, H, H);
                    if (ib == 0 && q == 0 && 16 * (ob0 + o) + c < H) part[a.net.b_off[l] + 16 * (ob0 + o) + c] = A.bias[l][o];
                }
        }
    }
};

template <class Op, int SPLIT, int WIDTH, int NL>
__global__ __launch_bounds__(512) void fused2_wave_kernel(const FusedArgs a) {
    Fused2<Op, SPLIT, WIDTH, NL>::run(a);
}

}  // namespace pinn
