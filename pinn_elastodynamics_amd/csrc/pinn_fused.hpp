// Fused loss+gradient kernel: forward chain, residual head, reverse chain AND the weight gradient in one persistent launch.
// Layouts: padded width <= 64 with the tile's state in registers (1, 4 or 5 streams: value-only sets, the wave head, the plate head),
// described first; padded width 96 (the reference's 8x80 net) with the state in the chain wave's LDS images ("LDSOP", further down).
//
// Versus chain_kernel + wgrad_kernel (pinn_device.hpp) nothing per-point ever goes to HBM as a
// [feature][point] panel:
//   * a workgroup = 8 waves in two roles (2 waves per SIMD, <= 256 registers each): waves 0-3 are
//     CHAIN waves, each owning a 16-point tile (64 points per workgroup step); waves 4-7 are
//     WEIGHT-GRADIENT waves, each owning one quadrant of every Wbar_l in persistent accumulators;
//   * what bounds a SIMD here is INSTRUCTION ISSUE, not a pipe (round-2 measurements, tools/probes/coexec_probe.hip: a plain
//     vector instruction costs ~4.5 cycles of the SIMD whichever wave issues it, a transcendental or packed-fp32 one 8-10, an MFMA
//     ~6 of issue on top of its 16 pipe cycles, and two waves of one SIMD do not issue concurrently).  So the chain is written for
//     FEW instructions: the weights carry a scale (repack_kernel's fused format, FUSED_WEIGHT_SCALE), which lets the three MFMAs of a
//     product accumulate in ONE chain (no per-value recombination), lets the bias ride in as the accumulator's initial value, and
//     lets the forward activations use an unscaled low part (1.5 instead of 2.5 instructions per value);
//   * and for OVERLAP INSIDE the wave: the MFMAs of feature block m+1 are issued between the vector instructions of block m
//     (software pipeline over the blocks of a layer and, in the forward, across layers);
//   * forward state S_l is parked per tile as its REGISTER IMAGE: one fully coalesced 1 KB store per (stream, k-step) fragment for
//     the fp16 high parts and one for the low parts (STATE_LO), issued behind the next weight-fragment loads (CDNA4 returns loads
//     and stores in order on one counter, so a store in front of a load is waited for with it).  In the reverse pass the
//     weight-gradient wave that shares the SIMD brings the high-part image back by LDS-DMA (no registers, and off the chain wave's
//     instruction stream); the chain wave reads its own lanes' records back from LDS -- and the low parts with plain loads, so that
//     the activation reverse sees the state in full precision -- and the weight-gradient waves rebuild MFMA fragments from the
//     same image with ds_read_b64_tr_b16 (the record order is rotated per 16-lane group so that those reads are at most 2-way
//     bank conflicted);
//   * for the weight gradient  Wbar_l = sum_points S_l^T Z_l  the contraction runs over points, so
//     both operands are needed "feature per lane, points in registers" -- the transpose of the
//     chain layout.  Every chain wave drops its Z_l tile into LDS as the same kind of register image (ds_write_b128);
//   * the weight-gradient accumulators are PERSISTENT MFMA accumulators, written once per launch
//     as per-workgroup partials (deterministic two-stage reduction, no atomics).
// Addressing discipline (this kernel is unrolled over 9 weight layers, so every loop-invariant
// address the compiler can hoist costs a VGPR for the whole launch): weights, biases and scratch
// go through buffer descriptors with ONE lane-offset VGPR and scalar (SGPR) offsets; LDS accesses
// use one lane-base VGPR per tensor plus compile-time immediates.
// Two workgroup barriers per weight layer -- ONE in the narrow four-stream layout since round 4 (ZDB below: the weight gradient takes the
// adjoints as fp16 high parts, the Z area holds two high-part images, and the chain wave writes Z_{L-1} into the other one while the
// weight-gradient waves read Z_L).
#pragma once
#include "pinn_device.hpp"
#include <type_traits>
#ifndef PINN_LO8_ENABLED
#define PINN_LO8_ENABLED 1
#endif
#ifndef PINN_FWD8_ENABLED
#define PINN_FWD8_ENABLED 1      // 0: the padded-width-96 layouts keep the forward on the four chain waves alone (A / B timing)
#endif
#ifndef PINN_QUAD_ENABLED
#define PINN_QUAD_ENABLED 1      // 0: the padded-width-128 layouts keep the round-3 chain (two waves per tile, half of the feature blocks each): for A / B timing
#endif

namespace pinn {


constexpr int FUSED_MAX_SETS = 4;

struct FusedArgs {
    NetDesc net;
    PackedWeights pw;          // pw.frags: the FUSED-format fragment store (repack_kernel: frags_fused)
    unsigned frags_bytes;      // size of pw.frags
    const float* x;
    const float* y;
    const float* t;
    const float* z;            // 4-input instantiations only (input order x, y, z, t)
    long n;
    long nsteps;               // workgroup steps = ceil(n / (16 * TILES))
    float sx[4], ox[4];        // input map; 4 inputs: index 2 = z, 3 = t
    float c1, c2, G, rho;
    float tw[16];
    const float* targets;      // NS = 1 only: [nout][n] or nullptr (= 0)   (single-set form; the set table below supersedes it)
    const float* aux;          // NS = 5 (plate head): the frozen nets' streams [2 nets (D,P)][5 streams][5 fields][n]
    // NS = 1: up to FUSED_MAX_SETS value-only point sets in ONE launch (loss_IC, loss_SRC, loss_NB, loss_FIX of a step): set k owns
    // the workgroup steps [set_step0[k], set_step0[k+1]) and has its own points, targets, output weights and loss slot
    int nsets;
    long set_step0[5];
    const float* set_x[4];
    const float* set_y[4];
    const float* set_t[4];
    const float* set_z[4];     // 4-input instantiations only (the 3-D side sets, round 6)
    const float* set_targets[4];
    long set_n[4];
    float set_tw[4][16];       // (3-input nets: 8 used; the 3-D data head: up to 16 outputs)
    int set_head[4];           // 0: data head (targets, output weights); 1: hole traction of the plate's composite fields (PLATE:452-461)
    const float* set_aux[4];   // traction sets: [12][n] = D0[5], P0[5], nx, ny of the set's points
    u32x4* scratch;            // [gridDim.x * TILES][SCRATCH_BYTES]: per-tile images of the parked states
    float* loss_part;          // [gridDim.x * TILES][8]
    float* partial;            // [gridDim.x][nparams]
    u32x4* wg_acc;             // [gridDim.x * 4 waves][WG_ACC_BYTES]: weight-gradient accumulators kept in memory (see Fused::NG)
    unsigned long long* dbg;   // optional phase timestamps (s_memtime) of workgroup 0, chain wave 0 / weight-gradient wave 0; nullptr = off
    // The persistent grid this launch part runs on: workgroups block0 .. block0 + grid - 1 of the launch (a plain launch: 0 and the launch's
    // grid).  fused_step_kernel (round 5) runs the collocation set AND the value-only side sets of a training step in ONE launch -- the side
    // sets' workgroups are the blocks behind the collocation set's and start on the compute units that run out of collocation steps first.
    int block0, grid;
    // XCD-aware step assignment (round 5).  Workgroups are dispatched round-robin over the 8 XCDs (XCD = blockIdx % 8), and the odd XCDs of an
    // MI355X run this kernel 2-3 % slower than the even ones -- measured with every workgroup's lifetime on the device wall clock, the same
    // pattern on two boxes and in every repetition (profiles/r05_workgroup_lifetimes.txt: us per step by XCD 36.7 37.6 35.9 37.2 36.5 37.7 36.2 37.3
    // and 35.3 35.9 35.0 35.7 35.1 36.1 34.9 35.6) --, so with one step per workgroup and round the launch ends with its slowest XCD, 2 % behind
    // the mean.  Steps [0, n_plain) go one per workgroup and round as ever; the TAIL [n_plain, nsteps) goes to the even-XCD workgroups only,
    // one per round of grid / 2.  Static, hence deterministic: which workgroup sums which points depends on the launch's shape alone.
    // n_plain >= nsteps: no tail.  (A first version interleaved bonus rounds and carried two more loop counters: +20 spilled registers in the
    // 8 x 64 kernel; this form keeps the loop's state at the step index.)
    long n_plain;
};
__device__ __forceinline__ int fused_bid(const FusedArgs& a) { return (int)blockIdx.x - a.block0; }
// the step of this workgroup behind `step` (>= nsteps: none)
__device__ __forceinline__ long fused_next_step(const FusedArgs& a, long step) {
    if (step >= a.n_plain) return step + (a.grid >> 1);
    const long nx = step + a.grid;
    if (nx < a.n_plain) return nx;
    const int b = fused_bid(a);                                   // leaving the plain rounds: the tail is the even XCDs'
    return (b & 1) ? a.nsteps : a.n_plain + ((b >> 3) * 4 + ((b & 7) >> 1));
}

// A launch constant, made opaque at its point of use inside the step loop.  Otherwise the compiler hoists whatever is computed from
// such constants alone (2 * term weight, the packed tangent seeds of the input state ...) out of the loop into registers it then has
// to spill -- and a spill reload in the step is a full memory round trip (round-2 phase stamps: 3.5 k cycles in the residual head).
__device__ __forceinline__ float in_loop(float v) {
#if defined(__AMDGCN__)
    asm volatile("" : "+v"(v));      // (a vector register: a scalar constraint is refused where the compiler holds the value in one)
#endif
    return v;
}

__device__ __forceinline__ unsigned in_loop(unsigned v) {
#if defined(__AMDGCN__)
    asm volatile("" : "+v"(v));
#endif
    return v;
}

__device__ __forceinline__ void fused_stamp(const FusedArgs& a, bool who, int slot) {
    if (a.dbg != nullptr && who) a.dbg[slot] = __builtin_readcyclecounter();
}
// the constant-rate wall clock of the device (hipDeviceAttributeWallClockRate; 100 MHz): with the shader-cycle stamp beside it, the launch's
// duration AND its clock come from the kernel itself -- no host event in the measurement (the two events around a bracketed launch cost it
// 0.02-0.03 ms)
__device__ __forceinline__ void fused_stamp_wall(const FusedArgs& a, bool who, int slot) {
#if defined(__AMDGCN__)
    if (a.dbg != nullptr && who) a.dbg[slot] = (unsigned long long)wall_clock64();
#else
    if (a.dbg != nullptr && who) a.dbg[slot] = 0ull;
#endif
}

// NS = 4: value + three tangent streams, residual head of net_f_sig (the collocation set).  NS = 1: value stream only, head
// sum_o w_o (Y_o - target_o)^2 -- the side sets loss_IC / loss_SRC / loss_NB / loss_FIX (INF:111-118, CONF:145-146).
// FASTSTATE (PINN_FLAG_STATE_FP16): the parked states keep their fp16 high parts only -- 17 % faster for the 8x64 net, at the price of
// a 2^-12 state rounding that cancellation amplifies at trained weights (see STATE_LO below).  Not the default.
// DIN_ = 4 (round 3): the 3-D Navier-Cauchy extension of BASELINE configs[4] -- inputs (x, y, z, t), five FIRST-order streams (value, x, y, z,
// t), 12 outputs and the 3-D residual head (oracle/nc3d_oracle.py); built for the LDS-operand layout of padded width 128.
template <class Op, int SPLIT, int WIDTH, int NL, int NS_ = 4, bool FASTSTATE = false, int DIN_ = 3>
struct Fused {
    static constexpr int NS = NS_, WB = WIDTH / 16, KS = WIDTH / 32, NP = SPLIT == 3 ? 2 : 1, DIN = DIN_;
    // the XCD-aware step assignment (FusedArgs::n_plain) -- not in the 3-D instantiation: the most register-starved kernel of the file (785 spilled
    // SGPRs) took the three extra scalar values of the loop as +10 % launch time (25.2 against 22.3-23.3 ms per 1 M points, with the tail on or off)
    static constexpr bool XCD_TAIL = DIN_ == 3;
    static_assert(NS == 4 || NS == 1 || NS == 5, "wave residual head (4 streams), value-only data head (1 stream) or plate / 3-D head (5 streams)");
    static_assert(DIN == 3 || (DIN == 4 && (NS == 5 || NS == 1)), "4 inputs: the five-stream 3-D head only");
    // NS = 5, 3 inputs: streams (value, x, y, t, tt) -- the fifth carries the second time derivative (PLATE:417-419) -- and the plate head:
    // composite F = P + D*N with the frozen nets' streams, plane-stress residuals (PLATE:358-439)
    static constexpr bool SECOND = NS == 5 && DIN == 3;
    static constexpr int NT = NS >= 4 ? (SECOND ? 3 : NS - 1) : 0;                 // first-order tangent streams 1..NT (stream s differentiates by input s - 1)
    static constexpr int HEAD = (DIN == 4 && NS == 5) ? HEAD_NC3D : (NS == 4 ? HEAD_WAVE : (NS == 1 ? HEAD_DATA : HEAD_PLATE));
    static constexpr int NOG = DIN == 4 ? 16 : 8;              // outputs a lane gathers for the head
    static constexpr int LT = DIN == 4 ? LOSS_SLOTS_3D : 8;    // loss partial slots per tile and set
    // weight fragments in the fused format of repack_kernel: [T(V), T(V - T(V)), T(T(V)/LO_SCALE)] with V = FUSED_WEIGHT_SCALE * w
    // when split, [T(w)] otherwise.  Every accumulator of this kernel holds WS * (W . x).
    static constexpr int P3 = NP == 2 ? 3 : 1;
    static constexpr float WS = NP == 2 ? FUSED_WEIGHT_SCALE : 1.0f;
    static constexpr float INV_WS = 1.0f / WS;
    static_assert(WB == 2 || WB == 4 || WB == 6 || WB == 8 || WB == 10, "fused kernel supports padded widths 32, 64, 96, 128 and 160");
    static_assert(NL >= 2, "fused kernel needs at least two hidden layers");
    // LDSOP (padded width 96, the reference's 70 / 80): a tile's state does not fit the register file next to its successor (2 x 96
    // registers), so the chain wave keeps it in its LDS image -- the register image IS the MFMA operand layout -- and reads one k-step
    // at a time (one ds_read_b128 per stream and part); every layer is "all output blocks accumulate, then the vector part block by
    // block".  The images are 24 KB (hi + lo) per tensor and tile: two tiles per workgroup.
    static constexpr bool LDSOP = WB > 4;
    static_assert(!LDSOP || NP == 2, "the LDS-operand layout is built for the split-precision cases");
    // One stream at these widths (the value-only side sets loss_IC / loss_SRC / loss_NB / loss_FIX of the reference's 8 x 80 / 8 x 100 nets,
    // round 3): images of 6 / 8 KB, so ALL layer states S_0..S_NL of both tiles stay in LDS (NL + 1 slots) -- nothing is parked, no LDS-DMA.
    // (the 3-D net's one-stream instantiation -- 10 layers: eleven 8 KB states per tile do not fit twice -- takes the PARKED one-slot layout of
    // the five-stream kernel instead: 16 KB of LDS per tile, states through the scratch images; round 6)
    static constexpr bool WSLDS = LDSOP && NS_ == 1 && DIN_ == 3;
    // Five streams at padded width 96 (the reference's plate net, 8 x 70: PLATE:885-887): images of 30 KB, and two tiles have room for
    // ONE state slot each beside the Z area (2 x 60 KB).  The LDS-DMA of S_L can then only start when the readers of S_{L+1} are done
    // -- in the hand-off window of layer L itself -- and that window waits for it.
    // Padded width 128 (the reference's semi-infinite net, 8 x 100: SEMI:679) with four streams: images of 32 KB, the same budget.
    static constexpr bool ONE_SLOT = LDSOP && !WSLDS && WB > 4 && (NS_ == 5 || WB >= 8);
    // (Five streams at padded width 128 -- the 3-D net of BASELINE configs[4]: 40 KB images, two tiles fill the 160 KB exactly and the
    // net constants come from memory.)
    static_assert(!(NS_ == 5 && WB == 8) || DIN_ == 4, "five streams at padded width 128: the 3-D instantiation only");
    static constexpr int TILES = LDSOP ? 2 : 4;               // 16-point tiles per workgroup step (one per chain wave)
    static constexpr int NJ = TILES / 2;                      // 32-point k-steps of the weight gradient per workgroup step
    static constexpr int IBW = WB / 2, OBW = WB / 2;          // weight-gradient wave (i,o) owns IBW x OBW blocks of every mid Wbar
    static constexpr float INV_LS = 1.0f / Op::LO_SCALE;
    typedef Chain<Op, SPLIT, WIDTH, 1, NS, HEAD> CH;
    typedef FragIndex<WIDTH> FI;
    // LDS per chain wave: [Z image | S images].  An image = the fragment records (1 KB each: 64 lanes x 16 B) of a chain-layout tensor,
    // lane records rotated (imgoff): record block (stream, k-step) of the state's high parts (S), (stream, k-step, part) of the
    // adjoints' high and scaled low parts (Z).  A chain lane writes / reads its own records with 16-byte LDS accesses; the
    // weight-gradient waves rebuild transposed MFMA fragments from the same bytes with ds_read_b64_tr_b16.
    // The state images in LDS hold the operand type's precision only (no low part): that is enough for the WEIGHT GRADIENT's operand
    // (a 2^-12 / sqrt(points) rounding noise), not for the chain wave's own activation reverse, which reads the low parts back from the
    // scratch image (STATE_LO below: at trained weights cancellation amplifies a 2^-12 state rounding ~20x).
    static constexpr int TENSOR_Z_B = NS * KS * NP * 1024;
    // LDSOP keeps the state images with BOTH parts (records ((s * KS + kk) * NP + p), the operand layout): the chain wave's reverse
    // reads its state in full precision.  (At the reference's trained weights the activation reverse amplifies a 2^-12 rounding of
    // the state ~20x by cancellation -- first-layer gradient blocks 5e-3 off against 2e-4 for fp32 --, while rounding the state
    // only as the weight gradient's operand costs nothing: tests/test_gpu_parity.py, DESIGN_HISTORY.md section 6.)
    static constexpr int SP = LDSOP ? NP : 1;                 // parts per state-image record group
    static constexpr int IMG_B = NS * KS * SP * 1024;
    // "SLDS": where S_0..S_NL of a tile fit in LDS (NL+1 slots: every 1-stream case, and the 4-stream 4x32 net) nothing is parked in
    // scratch and no LDS-DMA round trip sits between the layer phases; otherwise two slots (layer parity), filled by LDS-DMA
    // Net constants the forward reads block by block, staged in LDS once per launch: [(NL-1) x WIDTH hidden biases | 16 output biases]
    // (both pre-multiplied by WS: they are accumulator start values) and the first layer's [WIDTH][4] weight/bias rows.  From memory
    // each of these small loads was consumed right behind its issue -- a full L2 round trip per feature block, and behind the park
    // stores of a layer even the stores' acknowledgements (vector-memory operations return in order): round-2 block-level phase
    // trace, 1.6 k cycles for the first block step of a forward layer against 0.8 k for the others.
    static constexpr int CONST_BIAS_F = (NL - 1) * WIDTH + 16, CONST_F = CONST_BIAS_F + WIDTH * 4, CONST_B = CONST_F * 4;
    // (Five streams at width 64 fill the 160 KB with tensors alone: that instantiation reads the constants from memory.)
    static constexpr int BASE_SLOTS = WSLDS ? NL + 1 : (ONE_SLOT ? 1 : 2);
    // (Four inputs: the first layer's rows are 32 bytes and read from memory wherever they are used -- q_first, wide_first --: no LDS copy.)
    static constexpr bool CONST_LDS = DIN_ == 3 && TILES * (TENSOR_Z_B + BASE_SLOTS * IMG_B) + CONST_B <= 160 * 1024;
    static constexpr int CONST_USED = CONST_LDS ? CONST_B : 0;
    static constexpr bool SLDS = !LDSOP && 4 * (TENSOR_Z_B + (NL + 1) * IMG_B) + CONST_USED <= 160 * 1024;      // all 1-stream cases; 4 streams: 4x32 only
    // ZDB (round 4; the narrow four-stream layouts with parked states, i.e. the collocation kernel of the 8 x 64 / 4 x 64 nets): the WEIGHT
    // GRADIENT takes the adjoints as fp16 HIGH PARTS ONLY -- one MFMA per product instead of two.  tools/studies/wgrad_operand_study.py (the
    // kernel's arithmetic in numpy at the reference's trained nets, 4 k and 32 k points): per weight layer and bias the error against float64
    // is the same multiple of fp32's own error with Z (hi + lo), Z (hi) and even with S (hi + lo): the chain's error dominates, the weight
    // gradient's operand rounding (2^-12 per factor, random, averaging over the points) does not show -- PROVIDED the adjoints sit in fp16's
    // normal range, which the normalised term weights (host: tw / max tw) times ZDB_SEED_SCALE arrange (the "10x worse gradient" of the
    // round-2 study was the subnormal range of UNSCALED adjoints).  The reverse CHAIN keeps hi + scaled lo (three MFMAs per product).
    // What it buys: the Z area (NS * KS * NP records per tile) holds TWO high-part images instead of one two-part image, so the chain
    // wave writes Z_{L-1} into the other buffer WHILE the weight gradient of layer L reads Z_L -- fragment by fragment as the reverse step
    // produces them, not as a burst in a hand-off window -- and a reverse layer needs ONE workgroup barrier, not two.
    // (Also the plate's five-stream narrow layout, with seed scale 1: its residuals at fresh weights are thousands -- E = 20 -- and x 16 would put
    // the normalised seeds beyond fp16; the normalised scale alone has 8x of margin in the study.)
    static constexpr bool ZDB = !LDSOP && !SLDS && (NS_ == 4 || (NS_ == 5 && DIN_ == 3)) && KS == 2 && NP == 2;
    static constexpr float ZDB_SEED_SCALE = NS_ == 4 ? 16.0f : 1.0f;      // host side: adjoint seeds x 16, gradient / 16 at the reduction (fused_launch)
    static constexpr int ZNP = ZDB ? 1 : NP;                         // parts of an adjoint image in LDS
    // WGLO: the weight gradient also multiplies LOW parts (the adjoints' scaled low part; LDS-operand layouts: the states' too).  Off for ZDB
    // only.  (Round 4, measured and NOT adopted: high parts only in the LDS-operand layouts as well -- 8 x 80 6.2 -> 5.5 ms per 1 M points, but
    // 8 x 100 / 6 x 140 / the 3-D net within 1-4 %, and the reference's inf10s net then misses the per-layer fp32 bound of
    // tests/test_gpu_parity.py in its FIRST layer, whose gradient fp32 itself gets to 1e-6: these layouts keep three MFMAs per product.)
    static constexpr bool WG_HI = ZDB;
    static constexpr bool WGLO = NP == 2 && !WG_HI;
    // (Round 4, also measured and NOT adopted: high parts only in the MID layers of the padded-width-96 four-stream layout, first and last
    // layer as they are -- 8 x 80 6.24 -> 5.70 ms per 1 M points, and the reference's inf10s net then misses the same bound in a MID layer:
    // 1.8e-5 against fp32's 2.2e-7 on a block of norm 0.18, 1024 points.  At those weights the sum over points cancels, and the operand
    // rounding does not average out as 1 / sqrt(points) of the RESULT.)
    static constexpr int ZBUF_B = NS * KS * 1024;                    // ZDB: one high-part adjoint image
    static __device__ __forceinline__ constexpr int zbuf(int L) { return ZDB ? (L & 1) * ZBUF_B : 0; }      // Z_L lives in buffer L & 1
    static constexpr int S_SLOTS = SLDS ? NL + 1 : BASE_SLOTS;
    static constexpr int WAVE_B = TENSOR_Z_B + S_SLOTS * IMG_B;
    static constexpr int CONST_OFF = TILES * WAVE_B;
    // S1_BY_WG (round 4; the four-stream narrow kernel that recomputes S_1 in the reverse, RECOMP1 below): the recomputation is the
    // WEIGHT-GRADIENT wave's -- the wave that shares the tile's SIMD and has finished layer 2 long before the chain wave has.  It writes the high
    // parts into the layer's state slot (where the LDS-DMA would have put them) and the low parts, as LO8 records, into S1LO: 4 KB per tile
    // behind the constants.  The chain wave finds S_1 like any other kept state; 2.4 k cycles leave its critical path.  (Measured: 1.3 % of the
    // launch, not the 3.3 % those cycles are -- at its lower priority the weight-gradient wave needs 2-2.5 k cycles per QUARTER of the layer, and
    // the chain wave's own steps stretch by a few hundred cycles when its SIMD partner has vector work: the SIMD's vector issue is not idle
    // while the chain wave runs, which is also what the shared forward of tools/experiments/ found.)
    static constexpr int S1LO_OFF = (CONST_OFF + CONST_USED + 15) & ~15;
    static constexpr bool S1_BY_WG = !LDSOP && !SLDS && NS_ == 4 && NL >= 4 && KS == 2 && NP == 2 && !FASTSTATE && WB == 4 && PINN_LO8_ENABLED && Op::TOP_BYTE_IS_FLOAT && CONST_LDS &&
                                     S1LO_OFF + TILES * NS * 1024 <= 160 * 1024;
    static constexpr int LDS_B = S1_BY_WG ? S1LO_OFF + TILES * NS * 1024 : CONST_OFF + CONST_USED;
    // S1_HI_BY_WG (the plate's five-stream narrow kernel, whose LDS is full and whose chain wave has no registers to recompute anything): the
    // same wave recomputes the HIGH parts of S_1 into the slot; the low parts stay parked as LO8 records (the forward has them anyway).  The
    // chain wave's code does not change -- S_1 is a kept state with parked low parts, like S_{NL-1} -- and 10 KB per tile and direction
    // (of 86: high parts of six layers, low parts of seven) neither leave through L2 nor come back by LDS-DMA.
    // (3.19 -> 3.08 ms per 1 M points, -3.4 %; in two pieces instead of four -2.8 %; the role spills 31 registers more either way)
    static constexpr bool S1_HI_BY_WG = !LDSOP && !SLDS && NS_ == 5 && DIN_ == 3 && NL >= 6 && KS == 2 && NP == 2 && !FASTSTATE && WB == 4 && PINN_LO8_ENABLED &&
                                        Op::TOP_BYTE_IS_FLOAT;
    static constexpr bool S1_WG_ANY = S1_BY_WG || S1_HI_BY_WG;
    static_assert(LDS_B <= 160 * 1024, "LDS budget");
    // per tile: parked states S_1..S_{NL-1}.  Narrow layouts: [high-part images | low-part images]: the high parts return to LDS by
    // LDS-DMA (they are also the weight gradient's operand), the (unscaled) low parts are read back by the chain wave itself, block by
    // block, so that the activation reverse sees the state in full precision.  (With fp16-rounded states the gradient at the
    // reference's trained weights is 5e-3 off in the first-layer blocks -- fp32: 2e-4 --, amplified by cancellation; DESIGN_HISTORY.md section 6.)
    static constexpr bool STATE_LO = !LDSOP && NP == 2 && !FASTSTATE;
    static constexpr unsigned SCRATCH_LO = (unsigned)((NL - 1) * IMG_B);          // byte offset of the low-part images
    // LO8 (round 4, the narrow collocation kernels -- wave head and plate head -- of padded width 64): the parked LOW parts travel as ONE BYTE per value -- the top byte of the fp16 low
    // part, which is an e5m2 number (sign, the fp16 exponent, two mantissa bits) -- packed four to a dword by one v_perm_b32 and expanded back by
    // two.  What the activation reverse needs of the low parts was measured in round 2 (first-layer gradient blocks of the reference's inf10s net
    // against float64: exact states 8.5e-7, fp16 high parts only 7.7e-5, + 3 significant bits of low part 3.1e-6, + 8 bits 9.2e-7): a few bits
    // carry it.  Half the low-part bytes each way, one 16-byte store per stream and layer instead of two, four 16-byte loads per layer (requested
    // with the layer's first fragments) instead of sixteen 8-byte ones.  The launch is sensitive to its bytes through the vector-memory path more
    // than to the instructions that move them (DESIGN_HISTORY.md section 6 "Round 4"); the low image of a layer is then NS records: (stream) -> 16 bytes per
    // lane = the four blocks' four values each.
    static constexpr bool LO8 = PINN_LO8_ENABLED && Op::TOP_BYTE_IS_FLOAT && !LDSOP && !SLDS && WB == 4 && NP == 2 && !FASTSTATE && KS == 2;      // (four- and five-stream narrow layouts)
    // LO_FROM (round 5, the per-layer low-part policy): parked states S_2 .. S_{LO_FROM-1} travel WITHOUT a low-part record -- the activation reverse
    // of those layers sees fp16 states.  tools/studies/lo_policy_study.py (the kernel's arithmetic in numpy at the reference's trained nets and the
    // 8 x 64 fixture, per weight layer and bias, error against float64 as a multiple of host-fp32's: the GPU tests' bound is 6) prices every subset.
    // Which layers the reverse needs: the UPPER ones (at inf20s, the most sensitive net: without S_6's low part 3.9 -> 18, S_7 14, S_5 11.5, S_4 7.3,
    // S_3 5.4, S_2 4.6 on 4096 points; semi16s / conf14s / the 8 x 64 fixture do not move for any single layer).  Shipped: S_2 alone goes without
    // (four-stream kernel): worst multiple 3.9 -> 4.6 (4096 points) and 7.5 -> 7.4 (32 k points) at inf20s, unchanged elsewhere; dropping S_3 as well
    // reaches 6.3 and was not taken.  Measured (60 interleaved launches each, order shuffled per round): 4.531 -> 4.439 ms per 2 M points (-2.0 %; with
    // S_3 as well -2.5 %, all low parts gone -12 %).  The plate's five-stream kernel keeps every record: it has no trained-weight fixture of its own.
#ifdef PINN_LO_FROM
    static constexpr int LO_FROM = PINN_LO_FROM;
#else
    static constexpr int LO_FROM = NS_ == 4 ? 3 : 2;
#endif
    static __device__ __forceinline__ constexpr bool lo_layer(int l) { return l <= 1 || l >= LO_FROM; }
    static __device__ __forceinline__ uint32_t lo8_pack(uint32_t rows01, uint32_t rows23) {       // the high bytes of four fp16 values
#if defined(__AMDGCN__)
        return __builtin_amdgcn_perm(rows23, rows01, 0x07050301u);
#else
        return ((rows01 >> 8) & 0xffu) | (((rows01 >> 24) & 0xffu) << 8) | (((rows23 >> 8) & 0xffu) << 16) | (((rows23 >> 24) & 0xffu) << 24);
#endif
    }
    static __device__ __forceinline__ u32x2 lo8_unpack(uint32_t p) {                              // back to two fp16 pairs, low bytes zero
#if defined(__AMDGCN__)
        return u32x2{__builtin_amdgcn_perm(p, p, 0x010c000cu), __builtin_amdgcn_perm(p, p, 0x030c020cu)};
#else
        return u32x2{((p & 0xffu) << 8) | (((p >> 8) & 0xffu) << 24), (((p >> 16) & 0xffu) << 8) | (((p >> 24) & 0xffu) << 24)};
#endif
    }
    // LSUM_MEM (the 3-D kernel): the per-lane loss sums live in a 4 KB tail of the tile's scratch image instead of 16 registers per lane.
    // As registers they are live through the whole step and touched once: the compiler spilled them, and reloaded them in the head ONE BY
    // ONE behind a full vmcnt(0) each -- 20 memory round trips, 21 k cycles of a 419 k step (tools/phase_trace_3d.py).  In memory: 12 loads
    // in flight together, 12 adds, 12 stores per step (same wave, same addresses: program order).
    static constexpr bool LSUM_MEM = NS_ == 5 && DIN_ == 4;      // (the plate's five-stream kernels, 8 sums: measured no gain at width 64, slower at 96)
    static constexpr unsigned LSUM_OFF = (unsigned)((STATE_LO ? 2 : 1) * (NL - 1) * IMG_B);
    static constexpr unsigned SCRATCH_BYTES = LSUM_OFF + (LSUM_MEM ? 16u * 256u : 0u);
    static __device__ __forceinline__ constexpr int slot_of(int L) { return (SLDS || WSLDS) ? L : (ONE_SLOT ? 0 : (L & 1)); }

    // The NG mid weight layers that the reverse sweep reaches first (L = NL-1 .. NL-NG) keep their accumulator blocks in memory
    // (loaded in the layer's hand-off window, stored one layer later so that the write acknowledgement never sits in front of a
    // full drain): with all NL-1 layers in registers the compiler spilled several layers' worth anyway, reloaded and stored them
    // around the barriers, and the weight-gradient waves became the critical path of those layers (round-2 phase traces).
    // (padded width 96: layer 1's nine blocks stay in registers -- one layer less of the sums' round trip through L2: 8 x 80 6.39 -> 6.28 ms,
    // the plate's 8 x 70 8.07 -> 7.97 ms per 1 M points; a second layer spills 82 registers)
        static constexpr int NG = LDSOP ? ((WB == 6 && NS_ >= 4) ? NL - 2 : (WB == 8 ? NL - 2 : NL - 1)) : (NL >= 8 ? (ZDB ? 1 : 5) : (NL >= 4 ? 2 : 0));      // (NL = 8 before ZDB: NG = 2..7 within 1 %, 5 left the fewest spills; with ZDB the role has registers to spare -- no second accumulator set for the adjoints' low parts -- and every layer kept in registers is a round trip of sums less: NG = 5 / 4 / 3 / 2 / 1 / 0 -> 5.00 / 4.93 / 4.88 / 4.85 / 4.82 / 4.82 ms per 2 M points; the plate's five streams: NG = 5 / 3 / 1 -> 3.50 / 3.44 / 3.38 ms per 1 M points)
    static constexpr int NREG = NL - 1 - NG;                                   // mid layers 1..NREG accumulate in registers
    // Padded width 160: 5 x 5 blocks per wave do not fit the register file next to their running sums (100 + 100 registers), so the
    // weight gradient walks its out-blocks in three passes (2 + 2 + 1) and STREAMS the sums: a pass starts from its ten (five) records,
    // requested one pass ahead, and stores them behind its last MFMA.
    // Padded width 128 (round 4; the reference's 8 x 100 net and the 3-D net): the same streaming in two passes of 4 x 2 blocks.  Their 16 blocks
    // fit, but as one set they cost 2 x 68 registers (the sums requested for the next layer + the ones being accumulated) and 34 vector-memory
    // operations in front of the layer's LDS-DMA; streamed, the role has room to keep the FIRST mid layer's 16 blocks in registers -- one
    // layer's matrix less to read and write per 32-point step, 4.3 of 60 KB per point.  8 x 100: 11.70 -> 11.4 (streaming) -> 10.25 ms per
    // 1 M points (+ the register layer, at 38 spilled registers; a second one: 298 spilled, 15 ms); the 3-D kernel: 23.3 -> 23.1-23.3 ms
    // (its launch is bound by its 86 KB per point, of which this is 4).  Six blocks per side (8 x 80, plate 8 x 70) measured 1-3 % SLOWER streamed.
    static constexpr bool STREAM_SUMS = LDSOP && (WB == 10 || WB == 8);
    static constexpr int NBIASREC = (OBW + 3) / 4;                            // lane records holding the bias blocks (one float per block, four per record)
    static constexpr int NSUM = IBW * OBW + (LDSOP ? NBIASREC : 0);            // in-memory records per layer: the blocks (+ LDSOP: the bias blocks' lane records)
    static constexpr unsigned WG_ACC_BYTES = (unsigned)((NG > 0 ? NG : 1) * NSUM * 1024);
    static __device__ __forceinline__ constexpr bool in_memory(int L) { return L >= 1 && L <= NL - 1 && L > NREG; }
    // Cache policy of the two per-workgroup memory classes (round 6).  A persistent workgroup owns TILES scratch images (parked states: written
    // in the forward, read back once by LDS-DMA in the reverse) and four areas of running weight-gradient sums (read and written once per
    // step and in-memory layer).  Nothing of either is ever shared between workgroups, and both cycle once per 32-point step: whether the
    // traffic stops at the 256 MB Infinity Cache or goes on to HBM is a question of the launch's FOOTPRINT, 256 x (images + sums).
    // Measured (profiles/r06_footprint_and_cache_policy.txt): the 3-D kernel at 318 MB runs its 174 us step in 119 us with 176 workgroups
    // (224 MB) and in 139 us with the same traffic squeezed into half the addresses; marking ONE of the two classes non-temporal (`nt` on
    // its loads and stores: allocated to be replaced first) keeps the other one resident: 3-D 21.4 -> 18.8 ms per 1 M points, 6 x 140
    // (235 MB) 14.1 -> 12.8; both classes marked: 19.3; the layouts that fit (8 x 100: 214 MB, 8 x 80, plate 8 x 70, all narrow ones)
    // LOSE 2-10 % with either class marked.  So: only the layouts whose full grid exceeds the cache, and of the two classes the one that
    // moves fewer bytes per step.
#ifndef PINN_NT_POLICY
#define PINN_NT_POLICY 1           // A / B: 0 = no non-temporal hints anywhere, 2 = the sums of every LDS-operand layout, 3 = the images, 4 = both
#endif
    static constexpr size_t IMAGES_WG_BYTES = (size_t)TILES * SCRATCH_BYTES, SUMS_WG_BYTES = (size_t)4 * WG_ACC_BYTES;
    static constexpr bool OVER_MALL = PINN_NT_POLICY == 1 && LDSOP && !WSLDS && (size_t)256 * (IMAGES_WG_BYTES + SUMS_WG_BYTES) > ((size_t)224 << 20);
    static constexpr bool NT_FORCED = LDSOP && !WSLDS && PINN_NT_POLICY >= 2;
    static constexpr bool NT_SUMS = (OVER_MALL && SUMS_WG_BYTES <= IMAGES_WG_BYTES) || (NT_FORCED && PINN_NT_POLICY != 3);      // (each class is read and written once per step: bytes per step ~ size)
    static constexpr bool NT_IMAGES = (OVER_MALL && !NT_SUMS) || (NT_FORCED && PINN_NT_POLICY != 2);
    static constexpr int ACC_AUX = NT_SUMS ? 2 : 0, SCR_AUX = NT_IMAGES ? 2 : 0;        // aux operand of the buffer builtins: bit 1 = nt
    struct Sums {                      // running sums of one in-memory layer: this wave's blocks (+ LDSOP: its bias blocks, one float per lane and block)
        f32x4 blk[IBW][STREAM_SUMS ? 2 : OBW];      // (STREAM_SUMS: the first pass's two out-blocks only, requested ahead)
        f32x4 bias;
    };
    struct Acc {                       // persistent across the whole launch, all statically indexed
        f32x4 mid[NREG > 0 ? NREG : 1][NREG > 0 ? IBW : 1][NREG > 0 ? OBW : 1];
        f32x4 first;                   // Wbar_0 block (in-block 0, out-block = quad) if quad < WB
        f32x4 last;                    // Wbar_NL block (in-block = quad, out-block 0) if quad < WB
        float bias[NL + 1];
        // LDSOP (six blocks per side over four waves): a second first / last block (out- / in-block quad + 4 for quad < 2), and three bias
        // blocks per mid layer held by the waves with wi == 0
        f32x4 first2, last2;
        f32x4 first3, last3;           // (ten blocks per side: out- / in-block quad + 8 for quad < 2)
        float bias0b, bias0c;          // (LDSOP mid-layer bias blocks: one more in-memory record per layer, bias_record)
        float biasr[NREG > 0 ? NREG : 1][4];      // ... of the in-register layers (LDSOP)
    };

    // ---------------------------------------------------------------------------------------------
    // weight-gradient role
    // ---------------------------------------------------------------------------------------------
    // MFMA fragment "16-feature block, 32 points of one k-step" rebuilt from the chain waves' LDS data.  k-slot (q, e) <-> chain wave
    // 2j + (q>>1), local point 8(q&1) + e (same map for both operands); the lane-dependent part of the address lives in the base
    // pointer(s), everything else is an immediate.
    //   lane (c, q) reads the 8 bytes "features 4(c&3)..+3 of point 8(q&1)+(c>>2) [+4]" = one half of the record of chain lane
    //   (point, c&3).  The record rotation makes the +4-point address lane dependent: two base pointers.
    static __device__ __forceinline__ u32x4 sfrag(const char* b0, const char* b1, int off) {
        const v4i16 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)(b0 + off));
        const v4i16 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)(b1 + off));
        const u32x2 d0 = __builtin_bit_cast(u32x2, v0), d1 = __builtin_bit_cast(u32x2, v1);
        return u32x4{d0[0], d0[1], d1[0], d1[1]};
    }
    // byte offset of lane (c = point, q)'s record inside a 1 KB fragment record block: records of a 16-lane group rotated by 4q
    static __device__ __forceinline__ unsigned img_record(int point, int q) { return (unsigned)((((point + 4 * q) & 15) + 16 * q) * 16); }

    // NA x NBK blocks of one weight gradient:  acc[a][b] += sum over 64 points, NS streams of  S(block fa+a)^T . Z(block fb+b).
    // Operand fragments are fetched once per (k-step, stream) and shared by the NA*NBK blocks; a scheduling fence after every
    // group keeps the compiler from hoisting all transpose-reads of a layer ahead of the MFMAs.
    // s0/s1 (z0/z1): lane bases of the S (Z) image of chain wave 0 (+ the first block's offset); a second block steps 8 bytes (the
    // other half of the record: blocks 2k, 2k+1 share a record).
    // SLO (LDSOP, layers 1..NL): the state image also holds the (unscaled) low parts; a third MFMA per block pair brings the weight
    // gradient to the two-kernel path's accuracy (without it: a 1/sqrt(points) rounding noise, 7e-5 on 5 k points).
    // The LDS-DMA of S_{L-1} rides along, a slice behind every group: issued as one burst in the hand-off window, the 21 KB of reads
    // per wave (running sums + state image) took 3-7 k cycles to ISSUE (phase stamps: a compute unit's outstanding-request capacity
    // against the memory latency), with every wave of the workgroup waiting at the barrier behind it.
    struct DmaSrc;
    struct DmaJob {
        const DmaSrc* scr;
        unsigned lane16;
        char* tile_lds;
        int quad;
        __amdgpu_buffer_rsrc_t accr;      // this wave's in-memory running sums (STREAM_SUMS: the weight gradient loads / stores them pass by pass)
    };
    // DL >= 2: the LDS-DMA of S_{DL-1} rides along, a slice behind every group (see DmaJob)
    template <int NA, int NBK, bool SLO = false, int DL = 0, int SSTR = KS * SP * 1024>
    static __device__ __forceinline__ void wg_blocks(const char* s0, const char* s1, const char* z0, const char* z1, f32x4 (&acc)[NA][NBK],
                                                     float (&bias_out)[NBK], const DmaJob* job = nullptr) {
        auto dma_slice = [&](int g) {
            if constexpr (DL >= 2) dma_state(*job->scr, job->lane16, job->tile_lds, DL - 1, job->quad, g * N_DMA_ALL / (NJ * NS), (g + 1) * N_DMA_ALL / (NJ * NS));
        };
        static_assert(NA <= 2 && NBK <= 2, "blocks of one call share a fragment record");
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
        // software pipeline over the 2*NS (k-step, stream) groups: the transpose-reads of group g+1 are issued before the MFMAs
        // of group g, so LDS latency hides behind matrix work; the fence after each group bounds how far the compiler may hoist.
        // Two fragment sets used in strict alternation (no copies: a third set would not fit beside the persistent accumulators).
        struct Frags { u32x4 Ah[NA], Al[SLO ? NA : 1], Bh[NBK], Bl[NBK]; };
        auto fetch = [&](int g, Frags& f) {
            const int j = g / NS, st = g % NS;
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                f.Ah[a] = sfrag(s0, s1, 2 * j * WAVE_B + st * SSTR + 8 * a);
                if constexpr (SLO && WGLO) f.Al[a] = sfrag(s0, s1, 2 * j * WAVE_B + st * SSTR + 1024 + 8 * a);
            }
#pragma unroll
            for (int b = 0; b < NBK; ++b) {
                f.Bh[b] = sfrag(z0, z1, 2 * j * WAVE_B + (st * KS * ZNP) * 1024 + 8 * b);
                if constexpr (WGLO) f.Bl[b] = sfrag(z0, z1, 2 * j * WAVE_B + (st * KS * ZNP + 1) * 1024 + 8 * b);
            }
        };
        auto work = [&](int g, const Frags& f) {
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int b = 0; b < NBK; ++b) {
                    acc[a][b] = Op::mfma(f.Ah[a], f.Bh[b], acc[a][b]);
                    if constexpr (WGLO) cc[a][b] = Op::mfma(f.Ah[a], f.Bl[b], cc[a][b]);
                    if constexpr (SLO && WGLO) acc[a][b] = Op::mfma(f.Al[a], f.Bh[b], acc[a][b]);
                }
            if (g % NS == 0) {                    // bias gradient = ones^T . Z (value stream)
#pragma unroll
                for (int b = 0; b < NBK; ++b) {
                    bm[b] = Op::mfma(ones, f.Bh[b], bm[b]);
                    if constexpr (WGLO) bc[b] = Op::mfma(ones, f.Bl[b], bc[b]);
                }
            }
        };
        Frags fa, fb;
        fetch(0, fa);
#pragma unroll
        for (int g = 0; g < NJ * NS; g += 2) {
            if (g + 1 < NJ * NS) fetch(g + 1, fb);
            work(g, fa);
            dma_slice(g);
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < NJ * NS) {
                if (g + 2 < NJ * NS) fetch(g + 2, fa);
                work(g + 1, fb);
                dma_slice(g + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int b = 0; b < NBK; ++b) {
            bias_out[b] = WGLO ? bm[b][0] + bc[b][0] * INV_LS : bm[b][0];
            if constexpr (WGLO) {
#pragma unroll
                for (int a = 0; a < NA; ++a)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[a][b][r] += cc[a][b][r] * INV_LS;
            }
        }
    }

    // LDSOP mid layers: the wave's 3 x 3 blocks in ONE pass -- in-blocks (pair sp | single ss) x out-blocks (pair zp | single zs), both
    // state parts -- so that every operand fragment of a (k-step, stream) group is transpose-read once: 24 reads per group where four
    // calls of wg_blocks (2x2, 2x1, 1x2, 1x1) issue 48.  The round-2 phase trace had this role bound by its LDS reads (786 KB per
    // layer and workgroup).
    template <int L>
    static __device__ __forceinline__ void wg_blocks33(const char* sp0, const char* sp1, const char* ss0, const char* ss1, const char* zp0, const char* zp1,
                                                       const char* zs0, const char* zs1, f32x4 (&acc)[3][3], const DmaJob& job) {
        constexpr int NGRP = NJ * NS;
        auto dma_slice = [&](int g) {
            if constexpr (L >= 2 && !ONE_SLOT && !kept_in_lds(L - 1)) dma_state(*job.scr, job.lane16, job.tile_lds, L - 1, job.quad, g * N_DMA_ALL / NGRP, (g + 1) * N_DMA_ALL / NGRP);
        };
        static_assert(NP == 2 || !LDSOP, "split-precision layout");
        f32x4 cc[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) cc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        struct Frags { u32x4 Ah[3], Al[3], Bh[3], Bl[3]; };
        auto fetch = [&](int g, Frags& f) {
            const int j = g / NS, st = g % NS;
            const int os = 2 * j * WAVE_B + st * KS * SP * 1024, oz = 2 * j * WAVE_B + st * KS * NP * 1024;
            f.Ah[0] = sfrag(sp0, sp1, os);
            f.Ah[1] = sfrag(sp0, sp1, os + 8);
            f.Ah[2] = sfrag(ss0, ss1, os);
            f.Bh[0] = sfrag(zp0, zp1, oz);
            f.Bh[1] = sfrag(zp0, zp1, oz + 8);
            f.Bh[2] = sfrag(zs0, zs1, oz);
            if constexpr (WGLO) {
                f.Bl[0] = sfrag(zp0, zp1, oz + 1024);
                f.Bl[1] = sfrag(zp0, zp1, oz + 1024 + 8);
                f.Bl[2] = sfrag(zs0, zs1, oz + 1024);
                f.Al[0] = sfrag(sp0, sp1, os + 1024);
                f.Al[1] = sfrag(sp0, sp1, os + 1024 + 8);
                f.Al[2] = sfrag(ss0, ss1, os + 1024);
            }
        };
        auto work = [&](const Frags& f) {
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) acc[a][b] = Op::mfma(f.Ah[a], f.Bh[b], acc[a][b]);
            if constexpr (WGLO) {
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b) cc[a][b] = Op::mfma(f.Ah[a], f.Bl[b], cc[a][b]);
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b) acc[a][b] = Op::mfma(f.Al[a], f.Bh[b], acc[a][b]);
            }
        };
        Frags fa, fb;
        fetch(0, fa);
#pragma unroll
        for (int g = 0; g < NGRP; g += 2) {
            if (g + 1 < NGRP) fetch(g + 1, fb);
            work(fa);
            dma_slice(g);
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < NGRP) {
                if (g + 2 < NGRP) fetch(g + 2, fa);
                work(fb);
                dma_slice(g + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (WGLO) {
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[a][b][r] += cc[a][b][r] * INV_LS;
        }
    }
    // bias gradient of three out-blocks (pair zp | single zs):  ones^T . Z of the value stream, both parts
    static __device__ __forceinline__ void wg_bias3(const char* zp0, const char* zp1, const char* zs0, const char* zs1, float (&bias_out)[3]) {
        const uint32_t one2 = pack2<Op>(1.0f, 1.0f);
        const u32x4 ones = {one2, one2, one2, one2};
        f32x4 bm[3], bc[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) bm[b] = bc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int oz = 2 * j * WAVE_B;
            bm[0] = Op::mfma(ones, sfrag(zp0, zp1, oz), bm[0]);
            bm[1] = Op::mfma(ones, sfrag(zp0, zp1, oz + 8), bm[1]);
            bm[2] = Op::mfma(ones, sfrag(zs0, zs1, oz), bm[2]);
            if constexpr (WGLO) {
                bc[0] = Op::mfma(ones, sfrag(zp0, zp1, oz + 1024), bc[0]);
                bc[1] = Op::mfma(ones, sfrag(zp0, zp1, oz + 1024 + 8), bc[1]);
                bc[2] = Op::mfma(ones, sfrag(zs0, zs1, oz + 1024), bc[2]);
            }
        }
#pragma unroll
        for (int b = 0; b < 3; ++b) bias_out[b] = WGLO ? bm[b][0] + bc[b][0] * INV_LS : bm[b][0];
    }

    struct WgCtx {                     // lane bases of a weight-gradient wave (chain wave 0's tensors + the lane part)
        const char* z0;                // Z image, first four points of the k-slots
        const char* z1;                // Z image, the +4 points
        const char* s0;                // S image slot 0, first four points
        const char* s1;                // S image slot 0, the +4 points
    };
    // byte offset of 16-feature block mb inside an image: fragment record block (mb >> 1), half (mb & 1) of the 16-byte lane record
    static __device__ __forceinline__ int img_block(int mb) { return (mb >> 1) * SP * 1024 + 8 * (mb & 1); }
    static __device__ __forceinline__ int zimg_block(int mb) { return (mb >> 1) * ZNP * 1024 + 8 * (mb & 1); }

    // LDSOP: six 16-feature blocks per side.  Wave (wi, wo) owns in-blocks {2wi, 2wi+1, 4+wi} x out-blocks {2wo, 2wo+1, 4+wo}: a pair that
    // shares a fragment record and a single block, the same shape for every wave (offsets at run time, block counts at compile time).
    // (Eight blocks per side, padded width 128: wave (wi, wo) owns in-blocks 4wi..4wi+3 x out-blocks 4wo..4wo+3, two record pairs each.)
    // (Ten blocks per side, padded width 160: in-blocks 4wi..4wi+3 | 8+wi x out-blocks 4wo..4wo+3 | 8+wo, two record pairs and a single each.)
    static __device__ __forceinline__ int wide_block(int half, int i) { return WB == 4 ? 2 * half + i : WB == 8 ? 4 * half + i : (WB == 10 ? (i < 4 ? 4 * half + i : 8 + half) : (i < 2 ? 2 * half + i : 4 + half)); }
    // STREAM_SUMS (ten blocks per side): one pass = all IBW in-blocks x NBK out-blocks (O0, O0 + 1) of a mid layer over the step's 32 points.
    // `start`: the pass's running sums (requested a pass ahead); the next pass's records are requested into `next` before the first MFMA
    // and this pass's sums are stored behind the last one (loads ahead of stores: vector-memory operations complete in issue order).
    template <int L, int O0, int NBK, int NNEXT>
    static __device__ __forceinline__ void stream_pass(const char* s0, const char* s1, const char* z0, const char* z1, const int (&ia)[IBW], const int (&oz)[OBW],
                                                       const f32x4 (&start)[IBW][2], f32x4 (&next)[IBW][2], float (&bias_out)[2], const DmaJob& job) {
        f32x4 acc[IBW][NBK], cc[IBW][NBK], bm[NBK], bc[NBK];
#pragma unroll
        for (int i = 0; i < IBW; ++i)
#pragma unroll
            for (int b = 0; b < NBK; ++b) {
                acc[i][b] = start[i][b];
                cc[i][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
        for (int b = 0; b < NBK; ++b) bm[b] = bc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (NNEXT > 0) {
#pragma unroll
            for (int i = 0; i < IBW; ++i)
#pragma unroll
                for (int b = 0; b < NNEXT; ++b)
                    next[i][b] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(job.accr, job.lane16, acc_record(L, i, O0 + NBK + b), ACC_AUX));
        }
        const uint32_t one2 = pack2<Op>(1.0f, 1.0f);
        const u32x4 ones = {one2, one2, one2, one2};
        static_assert(NJ == 1, "one 32-point k-step per workgroup step");
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            u32x4 Ah[IBW], Al[IBW], Bh[NBK], Bl[NBK];
#pragma unroll
            for (int i = 0; i < IBW; ++i) {
                Ah[i] = sfrag(s0 + ia[i], s1 + ia[i], st * KS * SP * 1024);
                if constexpr (WGLO) Al[i] = sfrag(s0 + ia[i], s1 + ia[i], st * KS * SP * 1024 + 1024);
            }
#pragma unroll
            for (int b = 0; b < NBK; ++b) {
                Bh[b] = sfrag(z0 + oz[O0 + b], z1 + oz[O0 + b], st * KS * NP * 1024);
                if constexpr (WGLO) Bl[b] = sfrag(z0 + oz[O0 + b], z1 + oz[O0 + b], st * KS * NP * 1024 + 1024);
            }
#pragma unroll
            for (int i = 0; i < IBW; ++i)
#pragma unroll
                for (int b = 0; b < NBK; ++b) {
                    acc[i][b] = Op::mfma(Ah[i], Bh[b], acc[i][b]);
                    if constexpr (WGLO) {
                        cc[i][b] = Op::mfma(Ah[i], Bl[b], cc[i][b]);
                        acc[i][b] = Op::mfma(Al[i], Bh[b], acc[i][b]);
                    }
                }
            if (st == 0) {                    // bias gradient = ones^T . Z (value stream)
#pragma unroll
                for (int b = 0; b < NBK; ++b) {
                    bm[b] = Op::mfma(ones, Bh[b], bm[b]);
                    if constexpr (WGLO) bc[b] = Op::mfma(ones, Bl[b], bc[b]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < IBW; ++i)
#pragma unroll
            for (int b = 0; b < NBK; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][b][r] += WGLO ? cc[i][b][r] * INV_LS : 0.0f;
#pragma unroll
        for (int b = 0; b < NBK; ++b) bias_out[b] = WGLO ? bm[b][0] + bc[b][0] * INV_LS : bm[b][0];
        // A THIRD hazard hipcc does not see (found on the GPU, round 3; the x86 emulator cannot show it): a 16-byte buffer store whose data
        // registers a vector instruction overwrites in the NEXT issue slot stores the new value in its last dword.  The compiler pads that
        // hazard only for stores without a scalar offset register; ours have one.  Written as "store; fma into the same registers; store",
        // the per-block loop above produced exactly that sequence.  So: all sums are final first, then all stores, then wait states, and the
        // scheduler may not move vector work in between.
        store_fence(acc);
#pragma unroll
        for (int i = 0; i < IBW; ++i)
#pragma unroll
            for (int b = 0; b < NBK; ++b)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][b]), job.accr, job.lane16, acc_record(L, i, O0 + b), ACC_AUX);
        stores_issued();
    }
    // mid layer L of the ten-block layout: three passes over the out-blocks (pair, pair, single); lds_ = the first pass's sums (+ the bias
    // records), requested behind the previous layer's weight gradient (load_sums)
    template <int L>
    static __device__ __forceinline__ void wgrad_stream(const WgCtx& w, int quad, const Sums& lds_, const DmaJob& job) {
        static_assert((IBW == 5 && OBW == 5) || (IBW == 4 && OBW == 4), "ten or eight blocks per side");
        const int wi = quad >> 1, wo = quad & 1;
        const char* s0 = w.s0 + slot_of(L) * IMG_B;
        const char* s1 = w.s1 + slot_of(L) * IMG_B;
        int ia[IBW], oz[OBW];
#pragma unroll
        for (int i = 0; i < IBW; ++i) {
            ia[i] = img_block(wide_block(wi, i));
            oz[i] = zimg_block(wide_block(wo, i));
        }
        f32x4 n1[IBW][2], n2[IBW][2];
        float b01[2], b23[2], b4[2];
        if constexpr (OBW == 4) {              // eight blocks per side: two passes, one bias record
            stream_pass<L, 0, 2, 2>(s0, s1, w.z0, w.z1, ia, oz, lds_.blk, n1, b01, job);
            stream_pass<L, 2, 2, 0>(s0, s1, w.z0, w.z1, ia, oz, n1, n2, b23, job);
            if (wi == 0) {
                f32x4 ba = lds_.bias;
                ba[0] += b01[0];
                ba[1] += b01[1];
                ba[2] += b23[0];
                ba[3] += b23[1];
                f32x4 one[1][1] = {{ba}};
                store_fence(one);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, one[0][0]), job.accr, job.lane16, bias_record(L), ACC_AUX);
                stores_issued();
            }
            return;
        }
        stream_pass<L, 0, 2, 2>(s0, s1, w.z0, w.z1, ia, oz, lds_.blk, n1, b01, job);
        stream_pass<L, 2, 2, 1>(s0, s1, w.z0, w.z1, ia, oz, n1, n2, b23, job);
        stream_pass<L, 4, 1, 0>(s0, s1, w.z0, w.z1, ia, oz, n2, n1, b4, job);
        if (wi == 0) {                         // the bias blocks of out-half wo: two lane records (four + one floats)
            f32x4 ba = lds_.bias;
            ba[0] += b01[0];
            ba[1] += b01[1];
            ba[2] += b23[0];
            ba[3] += b23[1];
            f32x4 bb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(job.accr, job.lane16, bias_record(L) + 1024, ACC_AUX));
            bb[0] += b4[0];
            f32x4 both[1][2] = {{ba, bb}};
            store_fence(both);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, both[0][0]), job.accr, job.lane16, bias_record(L), ACC_AUX);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, both[0][1]), job.accr, job.lane16, bias_record(L) + 1024, ACC_AUX);
            stores_issued();
        }
    }

    template <int L>
    static __device__ __forceinline__ void wgrad_wide(const WgCtx& w, Acc& A, int quad, const Sums& lds_, Sums& pends_, const DmaJob& job) {
        const auto& ld = lds_.blk;
        auto& pend = pends_.blk;
        const char* s0 = w.s0 + slot_of(L) * IMG_B;
        const char* s1 = w.s1 + slot_of(L) * IMG_B;
        const int wi = quad >> 1, wo = quad & 1;
        if constexpr (L == 0 || L == NL) {
            // first layer: in-block 0 x out-blocks {quad, quad + 4, quad + 8}; last layer: in-blocks {quad, quad + 4, quad + 8} x out-block 0
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int blk = quad + 4 * k;
                if (blk < WB) {
                    f32x4 t[1][1] = {{L == 0 ? (k == 0 ? A.first : (k == 1 ? A.first2 : A.first3)) : (k == 0 ? A.last : (k == 1 ? A.last2 : A.last3))}};
                    float b[1];
                    if constexpr (L == 0) wg_blocks<1, 1>(s0, s1, w.z0 + zimg_block(blk), w.z1 + zimg_block(blk), t, b);
                    else wg_blocks<1, 1, true>(s0 + img_block(blk), s1 + img_block(blk), w.z0, w.z1, t, b);
                    if constexpr (L == 0) {
                        if (k == 0) { A.first = t[0][0]; A.bias[0] += b[0]; } else if (k == 1) { A.first2 = t[0][0]; A.bias0b += b[0]; } else { A.first3 = t[0][0]; A.bias0c += b[0]; }
                    } else {
                        if (k == 0) A.last = t[0][0]; else if (k == 1) A.last2 = t[0][0]; else A.last3 = t[0][0];
                        if (quad == 0 && k == 0) A.bias[NL] += b[0];
                    }
                }
            }
        } else if constexpr (STREAM_SUMS && in_memory(L)) {
            wgrad_stream<L>(w, quad, lds_, job);
        } else {
            if constexpr (!in_memory(L) && WB == 8) {    // an in-register layer of the eight-block layouts: four 2 x 2 passes in place
#pragma unroll
                for (int ip = 0; ip < 2; ++ip)
#pragma unroll
                    for (int op = 0; op < 2; ++op) {
                        f32x4 t[2][2] = {{A.mid[L - 1][2 * ip][2 * op], A.mid[L - 1][2 * ip][2 * op + 1]}, {A.mid[L - 1][2 * ip + 1][2 * op], A.mid[L - 1][2 * ip + 1][2 * op + 1]}};
                        float b2[2];
                        wg_blocks<2, 2, true, 0>(s0 + img_block(IBW * wi + 2 * ip), s1 + img_block(IBW * wi + 2 * ip), w.z0 + zimg_block(OBW * wo + 2 * op),
                                                 w.z1 + zimg_block(OBW * wo + 2 * op), t, b2, &job);
                        A.mid[L - 1][2 * ip][2 * op] = t[0][0]; A.mid[L - 1][2 * ip][2 * op + 1] = t[0][1];
                        A.mid[L - 1][2 * ip + 1][2 * op] = t[1][0]; A.mid[L - 1][2 * ip + 1][2 * op + 1] = t[1][1];
                        if (ip == 0 && wi == 0) {
                            A.biasr[L - 1][2 * op] += b2[0];
                            A.biasr[L - 1][2 * op + 1] += b2[1];
                        }
                    }
            } else if constexpr (!in_memory(L)) {        // an in-register layer (two-slot layout): the nine blocks accumulate in place
                static_assert(WB == 6, "in-register layers of the LDS-operand layouts: width 96 only");
                const char* sp0 = s0 + img_block(2 * wi);
                const char* sp1 = s1 + img_block(2 * wi);
                const char* ss0 = s0 + img_block(4 + wi);
                const char* ss1 = s1 + img_block(4 + wi);
                const char* zp0 = w.z0 + zimg_block(2 * wo);
                const char* zp1 = w.z1 + zimg_block(2 * wo);
                const char* zs0 = w.z0 + zimg_block(4 + wo);
                const char* zs1 = w.z1 + zimg_block(4 + wo);
                wg_blocks33<L>(sp0, sp1, ss0, ss1, zp0, zp1, zs0, zs1, A.mid[L - 1], job);
                if (wi == 0) {
                    float b3[3];
                    wg_bias3(zp0, zp1, zs0, zs1, b3);
                    A.biasr[L - 1][0] += b3[0];
                    A.biasr[L - 1][1] += b3[1];
                    A.biasr[L - 1][2] += b3[2];
                }
            } else {
            pends_.bias = lds_.bias;
            // pend starts from the layer's running sums
#pragma unroll
            for (int i = 0; i < IBW; ++i)
#pragma unroll
                for (int o = 0; o < OBW; ++o) pend[i][o] = ld[i][o];
            if constexpr (WB == 8 || WB == 4) {
                // 4 x 4 blocks as four 2 x 2 passes (pairs share a fragment record); bias blocks from the passes of in-pair 0
                // (padded width 64: 2 x 2 blocks, one pass)
#pragma unroll
                for (int ip = 0; ip < IBW / 2; ++ip)
#pragma unroll
                    for (int op = 0; op < OBW / 2; ++op) {
                        f32x4 t[2][2] = {{pend[2 * ip][2 * op], pend[2 * ip][2 * op + 1]}, {pend[2 * ip + 1][2 * op], pend[2 * ip + 1][2 * op + 1]}};
                        float b2[2];
                        // (two-slot layout, i.e. padded width 64: the LDS-DMA of S_{L-1} rides in the -- single -- pass, as in wg_blocks33)
                        constexpr int DMA_L = (WB == 4 && L >= 2 && !ONE_SLOT && !kept_in_lds(L - 1)) ? L : 0;
                        wg_blocks<2, 2, true, DMA_L>(s0 + img_block(IBW * wi + 2 * ip), s1 + img_block(IBW * wi + 2 * ip), w.z0 + zimg_block(OBW * wo + 2 * op),
                                                     w.z1 + zimg_block(OBW * wo + 2 * op), t, b2, &job);
                        pend[2 * ip][2 * op] = t[0][0]; pend[2 * ip][2 * op + 1] = t[0][1];
                        pend[2 * ip + 1][2 * op] = t[1][0]; pend[2 * ip + 1][2 * op + 1] = t[1][1];
                        if (ip == 0 && wi == 0) {
                            pends_.bias[2 * op] += b2[0];
                            pends_.bias[2 * op + 1] += b2[1];
                        }
                    }
            } else {
            const char* sp0 = s0 + img_block(2 * wi);
            const char* sp1 = s1 + img_block(2 * wi);
            const char* ss0 = s0 + img_block(4 + wi);
            const char* ss1 = s1 + img_block(4 + wi);
            const char* zp0 = w.z0 + zimg_block(2 * wo);
            const char* zp1 = w.z1 + zimg_block(2 * wo);
            const char* zs0 = w.z0 + zimg_block(4 + wo);
            const char* zs1 = w.z1 + zimg_block(4 + wo);
            wg_blocks33<L>(sp0, sp1, ss0, ss1, zp0, zp1, zs0, zs1, pend, job);
            if (wi == 0) {                 // the bias blocks of out-half wo
                float b3[3];
                wg_bias3(zp0, zp1, zs0, zs1, b3);
                pends_.bias[0] += b3[0];
                pends_.bias[1] += b3[1];
                pends_.bias[2] += b3[2];
            }
            }
            }
        }
    }

    // weight gradient of weight layer L (quad = weight-gradient wave index 0..3)
    template <int L>
    static __device__ __forceinline__ void wgrad(const WgCtx& w, Acc& A, int quad, const Sums& ld, Sums& pend, const DmaJob& job) {
        if constexpr (LDSOP) {
            wgrad_wide<L>(w, A, quad, ld, pend, job);
        } else {
            wgrad_narrow<L>(w, A, quad, ld.blk, pend.blk, job);
        }
    }
    template <int L>
    static __device__ __forceinline__ void wgrad_narrow(const WgCtx& w, Acc& A, int quad, const f32x4 (&ld)[IBW][OBW], f32x4 (&pend)[IBW][OBW], const DmaJob& job) {
        constexpr int DMA_L = DMA_IN_WGRAD && L >= 2 && L <= NL - 1 && !kept_in_lds(L - 1) ? L : 0;
        const char* s0 = w.s0 + slot_of(L) * IMG_B;
        const char* s1 = w.s1 + slot_of(L) * IMG_B;
        const char* z0 = w.z0 + zbuf(L);           // (ZDB: Z_L sits in buffer L & 1 of the Z area)
        const char* z1 = w.z1 + zbuf(L);
        const int wi = quad >> 1, wo = quad & 1;
        if constexpr (L == 0) {
            if (quad < WB) {
                f32x4 t[1][1] = {{A.first}};
                float b[1];
                wg_blocks<1, 1>(s0, s1, z0 + zimg_block(quad), z1 + zimg_block(quad), t, b);
                A.first = t[0][0];
                A.bias[0] += b[0];
            }
        } else if constexpr (L == NL) {
            if (quad < WB) {
                f32x4 t[1][1] = {{A.last}};
                float b[1];
                if constexpr (TOP_IN_Z) wg_blocks<1, 1, false, 0, TOPZ_STRIDE>(w.z0 + TOPZ_OFF + img_block(quad), w.z1 + TOPZ_OFF + img_block(quad), z0, z1, t, b);
                else wg_blocks<1, 1>(s0 + img_block(quad), s1 + img_block(quad), z0, z1, t, b);
                A.last = t[0][0];
                if (quad == 0) A.bias[NL] += b[0];
            }
        } else {
            float b[OBW];
            if constexpr (in_memory(L)) {
#pragma unroll
                for (int i = 0; i < IBW; ++i)
#pragma unroll
                    for (int o = 0; o < OBW; ++o) pend[i][o] = f32x4{0.f, 0.f, 0.f, 0.f};
                wg_blocks<IBW, OBW, false, DMA_L>(s0 + img_block(wi * IBW), s1 + img_block(wi * IBW), z0 + zimg_block(wo * OBW), z1 + zimg_block(wo * OBW), pend, b, &job);
#pragma unroll
                for (int i = 0; i < IBW; ++i)
#pragma unroll
                    for (int o = 0; o < OBW; ++o) pend[i][o] += ld[i][o];
            } else {
                wg_blocks<IBW, OBW, false, DMA_L>(s0 + img_block(wi * IBW), s1 + img_block(wi * IBW), z0 + zimg_block(wo * OBW), z1 + zimg_block(wo * OBW), A.mid[L - 1], b, &job);
            }
            // one bias block per wave per layer: out-block wo*OBW + wi (OBW == 2) or wo (OBW == 1, waves with wi == 0)
            if (OBW == 1) { if (wi == 0) A.bias[L] += b[0]; }
            else A.bias[L] += wi ? b[OBW - 1] : b[0];
        }
    }

    // parked state S_l of chain tile `quad`: asynchronous LDS-DMA of its scratch image into the layer's LDS slot, issued by the
    // weight-gradient wave that shares the SIMD.  Written as inline assembly on the GPU: through the builtin the compiler treats the
    // DMA as an LDS store it cannot disambiguate and drains it (s_waitcnt vmcnt(0)) in front of this wave's next transpose-read --
    // a full memory round trip per layer.  Completion is covered by the counted wait in front of the layer's second barrier.
    struct DmaSrc {
#if defined(__AMDGCN__)
        u32x4 desc;                    // buffer descriptor words (wave-uniform): base, base_hi, bytes, flags
#else
        __amdgpu_buffer_rsrc_t rsrc;
#endif
        __device__ __forceinline__ void init(char* base, unsigned bytes) {
#if defined(__AMDGCN__)
            const unsigned long long p = reinterpret_cast<unsigned long long>(base);
            desc = u32x4{(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)p), (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(p >> 32)) & 0xffffu,
                         bytes, 0x00020000u};
#else
            rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
#endif
        }
    };
    // Narrow layouts: the state slots are idle in the forward, so the high parts of the last parked states go straight into them instead of
    // to scratch and back by LDS-DMA (the kernel moves 17 KB per point through L2 at ~6 TB/s, profiles/README.md): S_{NL-1} always, and
    // S_{NL-2} where S_NL -- which takes the other slot at the top of the reverse -- can live in the Z area instead (TOP_IN_Z: the
    // 16-output adjoint Z_NL fills only the k-step-0 records of the Z area; with two k-steps and both parts the k-step-1 records are
    // exactly one high-part state image, stream stride 4 KB).  One LDS barrier in the forward, in front of the first such write, orders
    // it behind the previous step's last weight gradient (which finished a forward ago; the barrier makes that formal).
    static constexpr bool KEEP2 = !SLDS && !LDSOP;
    // ... and S_1 is not parked at all: the first layer is three multiply-adds and a tanh per feature, so the reverse recomputes it from
    // the inputs (first_mb, the forward's own function: the same bits) instead of moving 2 KB per point through L2.
    // (Four streams only: the five-stream instantiation has no registers for it -- 43 -> 63 spilled, 4.0 -> 4.25 ms.)
    static constexpr bool RECOMP1 = KEEP2 && NL >= 4 && NS_ == 4;
    // LDS-operand layouts: the chain halves write the recomputed S_1 (both parts) straight into the layer's slot in its hand-off window
    // (no registers involved: their states are read from the image anyway); neither parked nor brought back.
    // One-slot layouts only, where it also empties the layer-1 window of its LDS-DMA (8x100: 13.0 -> 11.7 ms; the two-slot width-96
    // layout measures the same with and without).
    static constexpr bool RECOMP_W = ONE_SLOT;
    // (LDS-operand layouts with two slots: slot 1 is idle in the forward too; S_{NL-1} is written there beside its ping-pong image and
    // neither parked nor brought back.)
    static constexpr bool KEEP_W = LDSOP && !ONE_SLOT && !WSLDS;
    static constexpr bool TOP_IN_Z = KEEP2 && KS == 2 && NP == 2;
    static constexpr int FIRST_KEPT = TOP_IN_Z ? NL - 2 : NL - 1;
    static __device__ __forceinline__ constexpr bool kept_in_lds(int l) { return WSLDS || (KEEP2 && l >= FIRST_KEPT && l <= NL - 1) || (KEEP_W && l == NL - 1) || ((RECOMP1 || RECOMP_W || S1_HI_BY_WG) && l == 1); }
    // S_NL inside the Z area: record (s * KS + 1) * NP + kk beside the two-part Z_NL; ZDB: the whole OTHER buffer as a plain high-part image
    static constexpr int TOPZ_OFF = ZDB ? zbuf(NL + 1) : NP * 1024, TOPZ_STRIDE = ZDB ? KS * 1024 : KS * NP * 1024;
    static constexpr int N_DMA_ALL = LDSOP ? IMG_B / 2048 : IMG_B / 1024;       // LDSOP: two waves share a tile's records
    // mid layers: the LDS-DMA of S_{L-1} is issued in slices inside the weight gradient of layer L, not as a burst in the hand-off window
    static constexpr bool DMA_IN_WGRAD = !SLDS && !ONE_SLOT && !WSLDS;
    static __device__ __forceinline__ void dma_state(const DmaSrc& src, unsigned lane16, char* tile_lds, int l /*1..NL-1*/, int quad, int ii0 = 0,
                                                     int ii1 = N_DMA_ALL) {
        if constexpr (SLDS || WSLDS) return;
        char* dst = tile_lds + TENSOR_Z_B + slot_of(l) * IMG_B;
        // LDSOP: two tiles, four waves: wave quad brings every second record of tile quad & 1 (records quad >> 1, +2, ...)
        const int i0 = LDSOP ? (quad >> 1) : 0;
#pragma unroll
        for (int ii = 0; ii < N_DMA_ALL; ++ii) {
            if (ii < ii0 || ii >= ii1) continue;
            const int i = LDSOP ? i0 + 2 * ii : ii;
#if defined(__AMDGCN__)
            const unsigned lds_addr = (unsigned)(uintptr_t)((lds_void*)(dst + i * 1024));
            if constexpr (NT_IMAGES)
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen nt lds"
                             :: "v"(lane16), "s"(src.desc), "s"(lds_addr), "s"((unsigned)((l - 1) * IMG_B + i * 1024)) : "memory");
            else
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds"
                             :: "v"(lane16), "s"(src.desc), "s"(lds_addr), "s"((unsigned)((l - 1) * IMG_B + i * 1024)) : "memory");
#else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(src.rsrc, (lds_void*)(dst + i * 1024), 16, lane16, (l - 1) * IMG_B + i * 1024, 0, 0);
#endif
        }
    }

    // LDSOP forward: S_k (k = 1..NL-1) sits in the tile's Z area (k odd) or S slot 0 (k even), complete behind the layer's barrier; copy
    // this wave's records (the ones dma_state brings back) to the scratch image.  The LDS reads are drained by the next lds_barrier.
    static __device__ __forceinline__ void park_image(__amdgpu_buffer_rsrc_t scr, unsigned lane16, const char* tile_lds, int k, int quad) {
        const char* src = tile_lds + ((k & 1) ? 0 : TENSOR_Z_B) + lane16;
        const int i0 = quad >> 1;
        u32x4 v[N_DMA_ALL];
#pragma unroll
        for (int ii = 0; ii < N_DMA_ALL; ++ii) v[ii] = *reinterpret_cast<const u32x4*>(src + (i0 + 2 * ii) * 1024);
#pragma unroll
        for (int ii = 0; ii < N_DMA_ALL; ++ii) __builtin_amdgcn_raw_buffer_store_b128(v[ii], scr, lane16, (k - 1) * IMG_B + (i0 + 2 * ii) * 1024, SCR_AUX);
    }

    // accumulator blocks of an in-memory layer: record (L - NREG - 1, i, o) of this wave's 1 KB-record area
    static __device__ __forceinline__ int acc_record(int L, int i, int o) { return ((L - NREG - 1) * NSUM + i * OBW + o) * 1024; }
    static __device__ __forceinline__ int bias_record(int L) { return ((L - NREG - 1) * NSUM + IBW * OBW) * 1024; }
    // Early sums (LDSOP): a layer's running sums go back to memory, and the next layer's are requested, right behind the layer's weight
    // gradient -- not in the next hand-off window, where their issue (a compute unit's request capacity against the memory latency)
    // kept every wave of the workgroup waiting.  The first barrier of a layer is then an LDS-only one for this role.
    static constexpr bool EARLY_SUMS = LDSOP;      // (the narrow layouts measure the same either way: 5.77 / 5.78 ms)
    static __device__ __forceinline__ void store_sums(__amdgpu_buffer_rsrc_t accr, unsigned lane16, int L, const Sums& p) {
        if constexpr (STREAM_SUMS) return;          // (stored pass by pass inside the weight gradient)
#pragma unroll
        for (int i = 0; i < IBW; ++i)
#pragma unroll
            for (int o = 0; o < OBW; ++o) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, p.blk[i][o]), accr, lane16, acc_record(L, i, o), ACC_AUX);
        if constexpr (LDSOP) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, p.bias), accr, lane16, bias_record(L), ACC_AUX);
        stores_issued();
    }
    static __device__ __forceinline__ void load_sums(__amdgpu_buffer_rsrc_t accr, unsigned lane16, int L, Sums& p) {
#pragma unroll
        for (int i = 0; i < IBW; ++i)
#pragma unroll
            for (int o = 0; o < (STREAM_SUMS ? 2 : OBW); ++o) p.blk[i][o] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(accr, lane16, acc_record(L, i, o), ACC_AUX));
        if constexpr (LDSOP) p.bias = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(accr, lane16, bias_record(L), ACC_AUX));
    }

    // fences around a group of 16-byte buffer stores (see stream_pass): the values are complete before the first store, and no vector
    // instruction follows the last one within eight wait states
    template <int NA, int NB>
    static __device__ __forceinline__ void store_fence(f32x4 (&v)[NA][NB]) {
#if defined(__AMDGCN__)
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b) asm volatile("" : "+v"(v[a][b]));
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
    static __device__ __forceinline__ void stores_issued() {
#if defined(__AMDGCN__)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 7" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
    // "wait until at most N of this wave's vector-memory operations are outstanding" (they complete in issue order)
    template <int N>
    static __device__ __forceinline__ void wait_vmcnt() {
#if defined(__AMDGCN__)
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
#endif
    }

    struct Ctx;
    struct S1Job {                    // S1_BY_WG: the chain tile this wave recomputes S_1 for, and the tile's inputs of this step
        const Ctx* x;
        float xin[4];
        mutable u32x4 S1[NS][1][KS][NP];      // k-step 0 between the two halves of recompute_s1
    };
    template <int L>
    struct WgDown {     // same barrier sequence as the chain role's Down<>
        // vector-memory operations this wave issues in the hand-off window of layer L (between the two barriers), in this order:
        // the stores of layer L+1's in-memory sums, the loads of layer L's, the LDS-DMA of S_{L-1}
        // (STREAM_SUMS: behind the last LDS-DMA slice of layer L+1's weight gradient stand the stores of its LAST pass -- the bias record, which
        // only two of the four waves store, is not counted: a smaller count only waits longer -- and the first pass's records of layer L)
        static constexpr int N_STORE = in_memory(L + 1) ? (STREAM_SUMS ? IBW * (OBW == 4 ? 2 : 1) : NSUM) : 0;
        static constexpr int N_LOAD = in_memory(L) ? (STREAM_SUMS ? IBW * 2 + 1 : NSUM) : 0;
        // LDSOP mid layers: the DMA of S_{L-1} is issued inside the weight gradient of layer L (wg_blocks33), not in the window
        // ONE_SLOT: the DMA of S_L itself, in layer L's own window, and the window waits for all of it
        static constexpr bool DMA_IN_WINDOW = !SLDS && !WSLDS && !ONE_SLOT && L >= 2 && (!DMA_IN_WGRAD || L == NL) && !kept_in_lds(L - 1);
        static constexpr bool DMA_OWN = ONE_SLOT && L >= 1 && L <= NL - 1 && !kept_in_lds(L);
        static constexpr int N_DMA = DMA_IN_WINDOW ? N_DMA_ALL : 0;
        // ZDB, layers NL-2 .. 0: the chain waves wrote Z_L into the other buffer of the Z area while layer L+1 was worked on, so ONE barrier
        // separates the layers.  This wave reaches it ahead of the chain waves (its layer is the shorter one): it uses the wait to send
        // layer L+1's sums back and to request layer L's; the counted wait covers what it issued before those -- the LDS-DMA of S_L.
        static constexpr bool ONE_BARRIER = ZDB && L <= NL - 2;
        static __device__ __forceinline__ void run(const FusedArgs& a, bool tracer, const WgCtx& w, const DmaSrc& scr, __amdgpu_buffer_rsrc_t accr,
                                                   unsigned lane16, char* tile_lds, Acc& A, int quad, Sums& pend, Sums& ld, const S1Job& sj) {
            if constexpr (ONE_BARRIER) {
                static_assert(!EARLY_SUMS && !DMA_IN_WINDOW && !DMA_OWN, "narrow parked layout");
                if constexpr (in_memory(L + 1)) store_sums(accr, lane16, L + 1, pend);
                if constexpr (in_memory(L)) load_sums(accr, lane16, L, ld);
                // S1_BY_WG: a piece of S_1 per idle window of this wave (a whole first layer in one window made the chain wave wait 1.9 k cycles
                // at layer 1's barrier, halves 0.8 k, quarters none): feature block 4 - L behind the weight gradient of layer L + 1
                if constexpr (S1_WG_ANY && NL >= 6) {
                    if constexpr (L >= 1 && L <= 4) recompute_s1<4 - L, 5 - L>(a, sj, sj.S1);
                } else if constexpr (S1_WG_ANY) {
                    if constexpr (L == 2) recompute_s1<0, 2>(a, sj, sj.S1);
                    if constexpr (L == 1) recompute_s1<2, 4>(a, sj, sj.S1);
                }
                // (the write in the last piece: slot 1 is free since the barrier of layer 2 -- S_3's readers were layer 3's)
                __builtin_amdgcn_sched_barrier(0);
                wait_vmcnt<N_STORE + N_LOAD>();
                fused_stamp(a, tracer, 64 + 3 * (NL - L));
                lds_barrier();
                fused_stamp(a, tracer, 65 + 3 * (NL - L));
                wgrad<L>(w, A, quad, ld, pend, DmaJob{&scr, lane16, tile_lds, quad, accr});
                fused_stamp(a, tracer, 66 + 3 * (NL - L));
                if constexpr (L >= 1) WgDown<L - 1>::run(a, tracer, w, scr, accr, lane16, tile_lds, A, quad, pend, ld, sj);
                return;
            }
            // (chain waves now overwrite the tensors; this wave's reads of them are done)
            if constexpr (EARLY_SUMS) lds_barrier();
            else __syncthreads();
            fused_stamp(a, tracer, 64 + 3 * (NL - L));
            // While the chain waves write their Z rows this wave has nothing to compute: it issues its memory traffic here, off
            // the critical phase.  The slot of S_{L-1} is free (its last readers finished before the barrier above).
            if constexpr (!EARLY_SUMS) {
                if constexpr (in_memory(L + 1)) store_sums(accr, lane16, L + 1, pend);      // a whole layer ahead of the next full drain
                if constexpr (in_memory(L)) load_sums(accr, lane16, L, ld);
            }
            if constexpr (DMA_IN_WINDOW) dma_state(scr, lane16, tile_lds, L - 1, quad);      // S_{L-1} streams in while layer L is worked on
            if constexpr (DMA_OWN) dma_state(scr, lane16, tile_lds, L, quad);
            // Second barrier: the tensors of layer L are complete.  This wave's contribution is the LDS-DMA of S_L, issued one layer
            // ago; everything issued since (the sums of layer L+1 stored, those of layer L requested, the window's DMA) may stay in flight,
            // so the drain is a COUNTED one.  (A surplus operation the compiler might add only makes the wait more conservative:
            // completion is in issue order.)
            __builtin_amdgcn_sched_barrier(0);
            wait_vmcnt<DMA_OWN ? 0 : N_STORE + N_LOAD + N_DMA>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            fused_stamp(a, tracer, 65 + 3 * (NL - L));
            wgrad<L>(w, A, quad, ld, pend, DmaJob{&scr, lane16, tile_lds, quad, accr});
            if constexpr (EARLY_SUMS) {
                if constexpr (in_memory(L)) store_sums(accr, lane16, L, pend);
                if constexpr (in_memory(L - 1)) load_sums(accr, lane16, L - 1, ld);
            }
            fused_stamp(a, tracer, 66 + 3 * (NL - L));
            if constexpr (L >= 1) WgDown<L - 1>::run(a, tracer, w, scr, accr, lane16, tile_lds, A, quad, pend, ld, sj);
        }
    };

    static __device__ __forceinline__ void wgrad_role(const FusedArgs& a, char* lds, int quad, int lane, int c, int q) {
        Acc A;
#pragma unroll
        for (int l = 0; l < NREG; ++l)
#pragma unroll
            for (int i = 0; i < IBW; ++i)
#pragma unroll
                for (int o = 0; o < OBW; ++o) A.mid[l][i][o] = f32x4{0.f, 0.f, 0.f, 0.f};
        A.first = A.first2 = A.first3 = f32x4{0.f, 0.f, 0.f, 0.f};
        A.last = A.last2 = A.last3 = f32x4{0.f, 0.f, 0.f, 0.f};
        A.bias0b = A.bias0c = 0.0f;
#pragma unroll
        for (int l = 0; l <= NL; ++l) A.bias[l] = 0.0f;
#pragma unroll
        for (int l = 0; l < (NREG > 0 ? NREG : 1); ++l) A.biasr[l][0] = A.biasr[l][1] = A.biasr[l][2] = A.biasr[l][3] = 0.0f;
        WgCtx w;
        {
            const char* wave0 = lds + (q >> 1) * WAVE_B;
            const int p0 = 8 * (q & 1) + (c >> 2), sub = c & 3;
            w.z0 = wave0 + img_record(p0, sub);
            w.z1 = wave0 + img_record(p0 + 4, sub);
            w.s0 = w.z0 + TENSOR_Z_B;
            w.s1 = w.z1 + TENSOR_Z_B;
        }
        // this wave feeds chain tile `quad` (if there is one: LDSOP has two tiles): descriptor of that tile's scratch image
        const int ftile = LDSOP ? (quad & 1) : quad;
        const long gtile = (long)fused_bid(a) * TILES + ftile;
        DmaSrc scr;
        scr.init(reinterpret_cast<char*>(a.scratch) + gtile * (long)SCRATCH_BYTES, SCRATCH_BYTES);
        const unsigned lane16 = (unsigned)lane * 16u;
        char* tile_lds = lds + ftile * WAVE_B;
        const __amdgpu_buffer_rsrc_t scr_st =
            __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<char*>(a.scratch) + gtile * (long)SCRATCH_BYTES), 0, (int)SCRATCH_BYTES, 0x00020000);
        // this wave's in-memory accumulator records, zeroed here (same-wave program order makes the first loads see the zeros)
        const __amdgpu_buffer_rsrc_t accr = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(reinterpret_cast<char*>(a.wg_acc) + ((long)fused_bid(a) * 4 + quad) * (long)WG_ACC_BYTES), 0, (int)WG_ACC_BYTES, 0x00020000);
        if constexpr (NG > 0) {
#pragma unroll
            for (int r = 0; r < NG * NSUM; ++r) __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, accr, lane16, r * 1024, ACC_AUX);
        }
        Sums pend, ld;
        Ctx xs1;                                          // S1_BY_WG: addressing of chain tile `quad` as far as the first layer needs it
        Ctx xf;                                   // FWD8: this wave's addressing state for its block of the forward (tile ftile)
        if constexpr (FWD8) xf.init(a, lds, ftile, lane, c, q);
        S1Job sj;
        sj.x = &xs1;
        if constexpr (S1_WG_ANY) {
            if constexpr (!CONST_LDS) xs1.w0p = __builtin_amdgcn_make_buffer_rsrc((void*)a.pw.w0p, 0, WIDTH * 16, 0x00020000);
            xs1.tenZ = lds + quad * WAVE_B;
            xs1.cw0 = lds + CONST_OFF + CONST_BIAS_F * 4 + q * 64;
            xs1.imgoff = img_record(c, q);
            xs1.s1lo = lds + S1LO_OFF + quad * NS * 1024 + lane * 16;
            xs1.c = c;
            xs1.q = q;
        }
        for (long step = fused_bid(a); step < a.nsteps; step = XCD_TAIL ? fused_next_step(a, step) : step + a.grid) {
            if constexpr (S1_WG_ANY) {                    // the tile's inputs: requested here, used a forward and six reverse layers later
                bool valid;
                long pidx;
                load_inputs(a, a.x, a.y, a.t, a.z, a.n, step * TILES + quad, c, sj.xin, valid, pidx);
            }
            if constexpr (KEEP2) lds_barrier();           // the forward's barrier in front of its first state-slot write (see KEEP2)
            if constexpr (FWD8) {
                float xf_in[4];
                bool valid;
                long pidx;
                load_inputs(a, a.x, a.y, a.t, a.z, a.n, step * TILES + ftile, c, xf_in, valid, pidx);
                xf.imgoff = in_loop(xf.imgoff);
                xf.lane16 = in_loop(xf.lane16);
                __syncthreads();                          // step barrier
                f8_wg_forward(a, xf, xf_in, quad >> 1, scr_st, lane16, tile_lds, quad);
            } else if constexpr (LDSOP) {
                __syncthreads();                          // step barrier (see the chain role): this wave's reads of the previous step are done
                // the forward's exchange barriers between the two halves of a tile; behind barrier l the image of S_{l+1} is complete and
                // this wave parks its share of it (the records it will bring back by LDS-DMA: same wave, same addresses, program order)
                for (int l = 0; l < NL; ++l) {
                    lds_barrier();
                    if (l + 1 <= NL - 1 && !kept_in_lds(l + 1)) park_image(scr_st, lane16, tile_lds, l + 1, quad);
                }
            }
            WgDown<NL>::run(a, fused_bid(a) == 0 && quad == 0 && c == 0 && q == 0 && step == 2 * (long)a.grid, w, scr, accr, lane16, tile_lds, A, quad, pend, ld, sj);
        }
        // ---- write this workgroup's partial gradient
        // (kept in the body of the role: as a separate function taking the accumulators by reference the same statements cost the
        // default kernel 17 more spilled registers and 6 % of its time)
        float* part = a.partial + (long)fused_bid(a) * a.net.nparams;
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
#pragma unroll
            for (int r = 0; r < 4; ++r) lo[r] = __shfl_xor(A.first2[r], 16);
#pragma unroll
            for (int r = 0; r < 4; ++r) A.first2[r] += lo[r] * INV_LS;
#pragma unroll
            for (int r = 0; r < 4; ++r) lo[r] = __shfl_xor(A.first3[r], 16);
#pragma unroll
            for (int r = 0; r < 4; ++r) A.first3[r] += lo[r] * INV_LS;
        }
        if constexpr (LDSOP) {
            // wave quad: first / last blocks quad and quad + 4; mid layers: in-blocks wide_block(wi, i) x out-blocks wide_block(wo, o)
            put_block(A.first, 0, 0, quad, DIN, H);
            put_block(A.last, NL, quad, 0, H, NO);
            if (q == 0 && 16 * quad + c < H) part[a.net.b_off[0] + 16 * quad + c] = A.bias[0];
            if (quad + 4 < WB) {
                put_block(A.first2, 0, 0, quad + 4, DIN, H);
                put_block(A.last2, NL, quad + 4, 0, H, NO);
                if (q == 0 && 16 * (quad + 4) + c < H) part[a.net.b_off[0] + 16 * (quad + 4) + c] = A.bias0b;
            }
            if (quad + 8 < WB) {
                put_block(A.first3, 0, 0, quad + 8, DIN, H);
                put_block(A.last3, NL, quad + 8, 0, H, NO);
                if (q == 0 && 16 * (quad + 8) + c < H) part[a.net.b_off[0] + 16 * (quad + 8) + c] = A.bias0c;
            }
            if (quad == 0 && q == 0 && c < NO) part[a.net.b_off[NL] + c] = A.bias[NL];
#pragma unroll
            for (int l = 1; l < NL; ++l) {
#pragma unroll
                for (int i = 0; i < IBW; ++i)
#pragma unroll
                    for (int o = 0; o < OBW; ++o) {
                        f32x4 v;
                        if (in_memory(l)) v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(accr, lane16, acc_record(l, i, o), ACC_AUX));
                        else v = A.mid[l <= NREG ? l - 1 : 0][NREG > 0 ? i : 0][NREG > 0 ? o : 0];
                        put_block(v, l, wide_block(wi, i), wide_block(wo, o), H, H);
                    }
                f32x4 bv, bv2 = f32x4{0.f, 0.f, 0.f, 0.f};
                if (in_memory(l)) {
                    bv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(accr, lane16, bias_record(l), ACC_AUX));
                    if constexpr (NBIASREC > 1) bv2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(accr, lane16, bias_record(l) + 1024, ACC_AUX));
                } else {
                    bv = f32x4{A.biasr[l <= NREG ? l - 1 : 0][0], A.biasr[l <= NREG ? l - 1 : 0][1], A.biasr[l <= NREG ? l - 1 : 0][2], A.biasr[l <= NREG ? l - 1 : 0][3]};
                }
                if (wi == 0 && q == 0) {
#pragma unroll
                    for (int o = 0; o < OBW; ++o) {
                        const int ob = wide_block(wo, o);
                        if (16 * ob + c < H) part[a.net.b_off[l] + 16 * ob + c] = o < 4 ? bv[o] : bv2[o - 4];
                    }
                }
            }
            return;
        }
        if (quad < WB) {
            put_block(A.first, 0, 0, quad, DIN, H);
            put_block(A.last, NL, quad, 0, H, NO);
            if (q == 0 && 16 * quad + c < H) part[a.net.b_off[0] + 16 * quad + c] = A.bias[0];
        }
        if (quad == 0 && q == 0 && c < NO) part[a.net.b_off[NL] + c] = A.bias[NL];
#pragma unroll
        for (int l = 1; l < NL; ++l) {
#pragma unroll
            for (int i = 0; i < IBW; ++i)
#pragma unroll
                for (int o = 0; o < OBW; ++o) {
                    f32x4 v;
                    if (in_memory(l)) v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(accr, lane16, acc_record(l, i, o), ACC_AUX));
                    else v = A.mid[l <= NREG ? l - 1 : 0][i][o];
                    put_block(v, l, wi * IBW + i, wo * OBW + o, H, H);
                }
            const bool owner = OBW == 1 ? wi == 0 : true;
            const int ob = wo * OBW + (OBW == 1 ? 0 : wi);
            if (owner && q == 0 && 16 * ob + c < H) part[a.net.b_off[l] + 16 * ob + c] = A.bias[l];
        }
    }

    // ---------------------------------------------------------------------------------------------
    // chain role
    // ---------------------------------------------------------------------------------------------
    static constexpr bool QUAD = LDSOP && !WSLDS && WB == 8 && PINN_QUAD_ENABLED != 0;      // (see "QUAD chain" below)
    // (a member the other layouts never read still moved their register allocation -- and put a vector write directly behind a 16-byte
    // park store of the width-32 kernel, tests/test_isa_hazards.py: it lives in a base only the QUAD layouts have)
    struct CtxQuad { char* lds0; };                // the workgroup's LDS (uniform): tile t's tensors at + t * WAVE_B (a QUAD chain wave works on both tiles)
    struct CtxPlain {};
    struct Ctx : std::conditional_t<QUAD, CtxQuad, CtxPlain> {       // wave-invariant addressing state of a chain wave
        __amdgpu_buffer_rsrc_t frags, scr, bias, w0p;  // bias / w0p: only where the constants are not in LDS
        unsigned lane16;                           // lane * 16: the only VGPR offset of the fragment traffic
        unsigned imgoff;                           // this lane's (rotated) 16-byte record inside a fragment record block of an S image
        char* tenZ;                                // wave's Z tensor (uniform); S images follow at +TENSOR_Z_B (+k*IMG_B)
        const char* cbias;                         // LDS constants: bias table + q * 16 (a lane holds accumulator rows 4q..4q+3)
        const char* cw0;                           // LDS constants: first-layer rows + q * 64
        const float* blast;                        // output-layer bias (constants not in LDS)
        char* s1lo;                                // S1_BY_WG: this lane's record of the tile's low-part image of S_1 (stream s at + s * 1024)
        int c, q;
        bool tracer;                               // workgroup 0, chain wave 0, lane 0
        __device__ __forceinline__ void set_tile(const FusedArgs& a, long gtile) {
            scr = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<char*>(a.scratch) + gtile * (long)SCRATCH_BYTES), 0, (int)SCRATCH_BYTES, 0x00020000);
        }
        __device__ __forceinline__ void init(const FusedArgs& a, char* lds, int slot, int lane, int c_, int q_) {
            frags = __builtin_amdgcn_make_buffer_rsrc((void*)a.pw.frags, 0, (int)a.frags_bytes, 0x00020000);
            if constexpr (!CONST_LDS) {
                bias = __builtin_amdgcn_make_buffer_rsrc((void*)a.pw.bias_mid, 0, (NL - 1) * WIDTH * 4, 0x00020000);
                w0p = __builtin_amdgcn_make_buffer_rsrc((void*)a.pw.w0p, 0, WIDTH * (DIN == 4 ? 32 : 16), 0x00020000);
                blast = a.pw.bias_last;
            }
            lane16 = (unsigned)lane * 16u;
            tenZ = lds + slot * WAVE_B;
            if constexpr (QUAD) this->lds0 = lds;
            cbias = lds + CONST_OFF + q_ * 16;
            cw0 = lds + CONST_OFF + CONST_BIAS_F * 4 + q_ * 64;
            imgoff = img_record(c_, q_);
            s1lo = lds + S1LO_OFF + slot * NS * 1024 + lane * 16;
            c = c_;
            q = q_;
            tracer = false;
        }
        __device__ __forceinline__ char* imgZ() const { return tenZ + imgoff; }
        __device__ __forceinline__ char* imgS(int L) const { return tenZ + TENSOR_Z_B + slot_of(L) * IMG_B + imgoff; }
    };

    // chain-layout adjoint fragments (hi and scaled lo) -> this lane's records of the wave's Z image
    template <int KSF>
    static __device__ __forceinline__ void put_zimage(char* img, const u32x4 (&F)[NS][1][KSF][NP]) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int kk = 0; kk < KSF; ++kk)
#pragma unroll
                for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(img + ((s * KS + kk) * NP + p) * 1024) = F[s][0][kk][p];
    }
    // ZDB: high parts only, into the buffer of the Z area that belongs to the layer (img = x.imgZ() + zbuf(L))
    template <int KSF>
    static __device__ __forceinline__ void put_zimage_hi(char* img, const u32x4 (&F)[NS][1][KSF][NP]) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int kk = 0; kk < KSF; ++kk) *reinterpret_cast<u32x4*>(img + (s * KS + kk) * 1024) = F[s][0][kk][0];
    }
    // high parts of chain-layout state fragments -> this lane's records of an LDS image
    template <int KSF>
    static __device__ __forceinline__ void put_image(char* img, const u32x4 (&F)[NS][1][KSF][NP]) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int kk = 0; kk < KSF; ++kk) *reinterpret_cast<u32x4*>(img + (s * KS + kk) * SP * 1024) = F[s][0][kk][0];
    }

    // Issue order of a forward block step: N groups of (one MFMA, V vector instructions).  A wave issues in order, and left to itself the
    // compiler puts the MFMAs of a block step in bursts of 8..24 (the matrix pipe then runs 16 cycles per instruction with the issue port
    // idle) and the vector part in runs of 40..80 (the pipe idle): tools/isa_model.py.  Two waves of a SIMD barely co-execute MFMA and
    // vector work (tools/probes/coexec_probe.hip: 0.83 of the serial time); inside ONE wave a 1 : 2 interleave runs at the pipe's rate.
    // Measured (round 3, same box, interleaved timing): forward 1 : 2 -> -1 % of the launch; the same in the reverse step +2 % (its
    // bursts time-share the pipe with the weight-gradient wave of the SIMD better than a fine interleave does), so the reverse keeps
    // the compiler's order.
    template <int N, int V>
    static __device__ __forceinline__ void interleave() {
#if defined(__AMDGCN__)
        if constexpr (N > 0) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, V, 0);
            interleave<N - 1, V>();
        }
#endif
    }
    // weight fragments of one feature block (NPARTS = 2: forward parts [V_hi, V_lo]; 3: reverse, plus [w_hi])
    template <int KSB, int NPARTS>
    static __device__ __forceinline__ void load_afrags(const Ctx& x, int frag0, u32x4 (&Af)[KSB][NPARTS]) {
#pragma unroll
        for (int kk = 0; kk < KSB; ++kk)
#pragma unroll
            for (int p = 0; p < NPARTS; ++p) Af[kk][p] = __builtin_amdgcn_raw_buffer_load_b128(x.frags, x.lane16, ((frag0 + kk) * P3 + p) * 1024, 0);
    }
    static constexpr int FP = NP == 2 ? 2 : 1;      // weight-fragment parts the forward uses
    static constexpr int RP = P3;                   // ... and the reverse

    // forward k-step of one feature block:  acc[s] += V_hi.x_hi + V_hi.x_lo + V_lo.x_hi  (= WS * W.x; x_lo unscaled)
    // MFMAs of different streams alternate, so that consecutive ones never share an accumulator
    template <int KK, int KSB>
    static __device__ __forceinline__ void fwd_kstep(const u32x4 (&Af)[KSB][FP], const u32x4 (&B)[NS][1][KSB][NP], f32x4 (&acc)[NS]) {
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[s] = Op::mfma(Af[KK][0], B[s][0][KK][0], acc[s]);
        if constexpr (NP == 2) {
#pragma unroll
            for (int s = 0; s < NS; ++s) acc[s] = Op::mfma(Af[KK][0], B[s][0][KK][1], acc[s]);
#pragma unroll
            for (int s = 0; s < NS; ++s) acc[s] = Op::mfma(Af[KK][1], B[s][0][KK][0], acc[s]);
        }
    }
    // reverse k-step:  acc[s] += V_hi.z_hi + (V_hi/LS).z_lo' + V_lo.z_hi  (= WS * W^T.z; z_lo' carries the LO_SCALE = LS)
    template <int KK, int KSB>
    static __device__ __forceinline__ void bwd_kstep(const u32x4 (&Af)[KSB][RP], const u32x4 (&Zf)[NS][1][KSB][NP], f32x4 (&acc)[NS]) {
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[s] = Op::mfma(Af[KK][0], Zf[s][0][KK][0], acc[s]);
        if constexpr (NP == 2) {
#pragma unroll
            for (int s = 0; s < NS; ++s) acc[s] = Op::mfma(Af[KK][2], Zf[s][0][KK][1], acc[s]);
#pragma unroll
            for (int s = 0; s < NS; ++s) acc[s] = Op::mfma(Af[KK][1], Zf[s][0][KK][0], acc[s]);
        }
    }
    template <int K0, int K1, int KSB>
    static __device__ __forceinline__ void fwd_ksteps(const u32x4 (&Af)[KSB][FP], const u32x4 (&B)[NS][1][KSB][NP], f32x4 (&acc)[NS]) {
        if constexpr (K0 < K1) {
            fwd_kstep<K0, KSB>(Af, B, acc);
            fwd_ksteps<K0 + 1, K1, KSB>(Af, B, acc);
        }
    }
    template <int K0, int K1, int KSB>
    static __device__ __forceinline__ void bwd_ksteps(const u32x4 (&Af)[KSB][RP], const u32x4 (&Zf)[NS][1][KSB][NP], f32x4 (&acc)[NS]) {
        if constexpr (K0 < K1) {
            bwd_kstep<K0, KSB>(Af, Zf, acc);
            bwd_ksteps<K0 + 1, K1, KSB>(Af, Zf, acc);
        }
    }
    // accumulators of a new forward block: the value stream starts at WS * bias (the LDS table holds the product)
    static __device__ __forceinline__ void acc_init(const f32x4& bias, f32x4 (&acc)[NS]) {
        acc[0] = bias;
#pragma unroll
        for (int s = 1; s < NS; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    static __device__ __forceinline__ void acc_zero(f32x4 (&acc)[NS]) {
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // WS * bias of feature block mb of weight layer l (1..NL-1; l = NL, mb = 0: the output layer), from the LDS table
    static __device__ __forceinline__ f32x4 load_bias(const Ctx& x, int l, int mb) {
        if constexpr (CONST_LDS) {
            return *reinterpret_cast<const f32x4*>(x.cbias + ((l - 1) * WIDTH + 16 * mb) * 4);
        } else {
            if (l == NL) return *reinterpret_cast<const f32x4*>(x.blast + 4 * x.q) * WS;
            return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x.bias, (unsigned)x.q * 16u, ((l - 1) * WIDTH + 16 * mb) * 4, 0)) * WS;
        }
    }

    // state fragments (hi, unscaled lo) of feature block MB from per-point values
    template <int MB, int KSF = KS>
    static __device__ __forceinline__ void emit_state(u32x4 (&Bn)[NS][1][KSF][NP], const float (&vals)[NS][4]) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            uint32_t h0, h1, l0 = 0, l1 = 0;
            if constexpr (NP == 2) {
                split4u<Op>(vals[s][0], vals[s][1], vals[s][2], vals[s][3], h0, h1, l0, l1);
                Bn[s][0][MB >> 1][1][(MB & 1) * 2 + 0] = l0;
                Bn[s][0][MB >> 1][1][(MB & 1) * 2 + 1] = l1;
            } else {
                h0 = pack2<Op>(vals[s][0], vals[s][1]);
                h1 = pack2<Op>(vals[s][2], vals[s][3]);
            }
            Bn[s][0][MB >> 1][0][(MB & 1) * 2 + 0] = h0;
            Bn[s][0][MB >> 1][0][(MB & 1) * 2 + 1] = h1;
        }
    }

    // tanh and the scaled derivative from a scaled pre-activation  zs = WS * z :
    //   e = 2^(zs * 2 log2(e) / WS),  r = 1/(1 + e),  h = 1 - 2r,  sds = (1 - h^2)/WS = (4/WS) r (1 - r)
    static __device__ __forceinline__ void tanh_scaled(float zs, float& h, float& sds) {
        const float e = __builtin_amdgcn_exp2f(zs * (2.8853900817779268f * INV_WS));
        const float r = __builtin_amdgcn_rcpf(1.0f + e);
        h = 1.0f - 2.0f * r;
        const float c4 = r * (4.0f * INV_WS);
        sds = c4 - c4 * r;
    }

    // The operand fragments are written by inline assembly (split2u / split2: mixed-precision conversions).  hipcc pads the
    // "vector write -> MFMA operand read" wait states only between instructions it knows; an MFMA scheduled right behind such a
    // statement reads stale registers (seen on the GPU as a wrong forward as soon as nothing else happened to sit in between).  This
    // fence names the fragments of k-step kk and opens with the two wait states; every MFMA reading them is ordered behind it.
    template <int KSF>
    static __device__ __forceinline__ void operands_ready(u32x4 (&F)[NS][1][KSF][NP], int kk) {
#if defined(__AMDGCN__)
        if constexpr (NS == 5 && NP == 2)
            asm volatile("s_nop 1" : "+v"(F[0][0][kk][0]), "+v"(F[0][0][kk][1]), "+v"(F[1][0][kk][0]), "+v"(F[1][0][kk][1]), "+v"(F[2][0][kk][0]),
                                     "+v"(F[2][0][kk][1]), "+v"(F[3][0][kk][0]), "+v"(F[3][0][kk][1]), "+v"(F[4][0][kk][0]), "+v"(F[4][0][kk][1]));
        else if constexpr (NS == 5)
            asm volatile("s_nop 1" : "+v"(F[0][0][kk][0]), "+v"(F[1][0][kk][0]), "+v"(F[2][0][kk][0]), "+v"(F[3][0][kk][0]), "+v"(F[4][0][kk][0]));
        else if constexpr (NS == 4 && NP == 2)
            asm volatile("s_nop 1" : "+v"(F[0][0][kk][0]), "+v"(F[0][0][kk][1]), "+v"(F[1][0][kk][0]), "+v"(F[1][0][kk][1]),
                                     "+v"(F[2][0][kk][0]), "+v"(F[2][0][kk][1]), "+v"(F[3][0][kk][0]), "+v"(F[3][0][kk][1]));
        else if constexpr (NS == 4) asm volatile("s_nop 1" : "+v"(F[0][0][kk][0]), "+v"(F[1][0][kk][0]), "+v"(F[2][0][kk][0]), "+v"(F[3][0][kk][0]));
        else if constexpr (NP == 2) asm volatile("s_nop 1" : "+v"(F[0][0][kk][0]), "+v"(F[0][0][kk][1]));
        else asm volatile("s_nop 1" : "+v"(F[0][0][kk][0]));
#endif
    }

    // (Round 4, measured and NOT adopted: the elementwise work on PAIRS of accumulator rows with v_pk_mul / v_pk_add / v_pk_fma_f32.  Standing
    // alone a packed instruction issues at the rate of its scalar form and does two values -- but next to an MFMA it does not issue in the
    // MFMA's shadow: a group "1 MFMA + 1 v_pk_*" takes 33 cycles where "1 MFMA + 2 scalar" takes 17 (tools/probes/opcode_cost_probe.hip,
    // profiles/r04_opcode_issue_costs.md).  10 % fewer vector instructions, the same launch time: tools/experiments/r4_chain_issue_experiments.patch.)
    // vector part of a forward block: activation of the value stream, tangent streams, split into the next layer's operand
    template <int MB, int KSF = KS>
    static __device__ __forceinline__ void fwd_valu(const f32x4 (&acc)[NS], u32x4 (&Bn)[NS][1][KSF][NP]) {
        float vals[NS][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float h, sds;
            tanh_scaled(acc[0][r], h, sds);
            vals[0][r] = h;
#pragma unroll
            for (int s = 1; s <= NT; ++s) vals[s][r] = sds * acc[s][r];
            if constexpr (SECOND)                        // h_tt = (1-h^2) z_tt - 2 h h_t z_t   (z_t = acc[3] / WS)
                vals[4][r] = sds * acc[4][r] - (2.0f * INV_WS) * h * vals[3][r] * acc[3][r];
        }
        emit_state<MB, KSF>(Bn, vals);
    }

    // forward first layer (K = 3, VALU): INF:191-195 with the tangent seeds e_k * sx_k
    template <int MB, int END = WB>
    static __device__ __forceinline__ void first_mb(const FusedArgs& a, const Ctx& x, const float (&xin)[4], u32x4 (&Bn)[NS][1][KS][NP]) {
        float vals[NS][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            f32x4 w;
            if constexpr (CONST_LDS) w = *reinterpret_cast<const f32x4*>(x.cw0 + (16 * MB + r) * 16);
            else w = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x.w0p, (unsigned)x.q * 64u, (16 * MB + r) * 16, 0));
            float h, sd;
            tanh_act(w[3] + w[0] * xin[0] + w[1] * xin[1] + w[2] * xin[2], h, sd);
            vals[0][r] = h;
#pragma unroll
            for (int s = 1; s <= NT; ++s) vals[s][r] = sd * (a.sx[s - 1] * w[s - 1]);
            if constexpr (SECOND) vals[4][r] = -2.0f * h * vals[3][r] * (a.sx[2] * w[2]);      // z_tt = 0 at the first layer
        }
        emit_state<MB>(Bn, vals);
        if constexpr (MB + 1 < END) first_mb<MB + 1, END>(a, x, xin, Bn);
    }

    // forward: park the high parts of state S_l as the tile's register image (scratch), or straight into its LDS slot
    // Called in every block step MB of the layer that consumes S_l (it stays in registers as that layer's MFMA operand): the scratch stores
    // are SPREAD over the layer's block steps -- stream MB's records in step MB (round 4).  Issued as one burst in step 0, the 16 stores of a
    // layer filled the wave's vector-memory queue and each cost it ~50 cycles of issue (ablation: all parks 4.7 k cycles of a 78 k step).
    // Every step issues its stores BEHIND its weight-fragment loads, and a load waited for in a later step was issued after at most one
    // step's stores, long acknowledged by then (vector-memory operations complete in order on one counter).
    static constexpr bool SPREAD_PARKS = !SLDS && !LDSOP && WB >= NS - 1;
    template <int MB>
    static __device__ __forceinline__ void park_state(const Ctx& x, int l, const u32x4 (&Sf)[NS][1][KS][NP]) {
        constexpr bool FIRST = MB == 0;
        // streams parked in this step: all in step 0 (no spreading), or stream MB (+ the streams beyond WB - 1 in the last step)
        auto mine = [&](int s) { return SPREAD_PARKS ? (s == MB || (MB == WB - 1 && s >= WB)) : FIRST; };
        if constexpr (SLDS) {
            if constexpr (FIRST) put_image<KS>(x.imgS(l), Sf);               // the tile's own records of slot l; nobody else touches them in the forward
        } else if (RECOMP1 && l == 1) {
            return;                                     // recomputed by the reverse (hi and lo)
        } else if (S1_HI_BY_WG && l == 1) {
            // (high parts recomputed by the weight-gradient wave; the low parts are parked below)
        } else if (kept_in_lds(l)) {
            if constexpr (FIRST) {
                if (l == FIRST_KEPT) lds_barrier();     // (see KEEP2)
                put_image<KS>(x.imgS(l), Sf);
            }
        } else {
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int kk = 0; kk < KS; ++kk)
                    if (mine(s)) __builtin_amdgcn_raw_buffer_store_b128(Sf[s][0][kk][0], x.scr, x.imgoff, (l - 1) * IMG_B + (s * KS + kk) * 1024, 0);
        }
        if constexpr (LO8) {
            if (lo_layer(l))
#pragma unroll
            for (int s = 0; s < NS; ++s)
                if (mine(s)) {
                    const u32x4 &k0 = Sf[s][0][0][NP - 1], &k1 = Sf[s][0][1][NP - 1];
                    const u32x4 rec = {lo8_pack(k0[0], k0[1]), lo8_pack(k0[2], k0[3]), lo8_pack(k1[0], k1[1]), lo8_pack(k1[2], k1[3])};
                    __builtin_amdgcn_raw_buffer_store_b128(rec, x.scr, x.imgoff, SCRATCH_LO + (l - 1) * IMG_B + s * 1024, 0);
                }
        } else if constexpr (STATE_LO) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int kk = 0; kk < KS; ++kk)
                    if (mine(s)) __builtin_amdgcn_raw_buffer_store_b128(Sf[s][0][kk][NP - 1], x.scr, x.imgoff, SCRATCH_LO + (l - 1) * IMG_B + (s * KS + kk) * 1024, 0);
        }
    }

    // One hidden layer of the forward pipeline (weight layer l: `in` = S_l -> `out` = S_{l+1}).
    //   entry: A[m] = fragments of block m of this layer, all loaded (during the previous layer); acc[0] = block 0, MFMAs issued
    //   step m: reload A[m] (consumed) with block m of the NEXT weight layer (nfrag0; the output layer has one block), issue the MFMAs
    //   of block m+1, run the vector part of block m in between.  In the last step the k-steps of the next layer's block 0 that only
    //   need finished parts of `out` are issued next to the last vector part -- the pipeline runs across layers.
    //   S_l is parked in step 0 BEHIND that step's fragment load, and every fragment load has a whole layer to arrive: CDNA4 returns
    //   loads and stores in order on one counter, so a load issued after the park stores waits for their write acknowledgements
    //   (measured: ~2 k cycles; with a one-block prefetch distance every layer stalled on them).
    //   exit: the same invariant for the next layer.
    static constexpr int KOVL = (WB - 1) / 2;      // k-steps of the next layer's block 0 that can start before this layer's last block
    template <int MB>
    static __device__ __forceinline__ void fwd_step(const Ctx& x, int l, int nfrag0, bool next_is_out, const u32x4 (&in)[NS][1][KS][NP],
                                                    u32x4 (&out)[NS][1][KS][NP], u32x4 (&A)[WB][KS][FP], f32x4 (&acca)[NS], f32x4 (&accb)[NS],
                                                    f32x4 (&bb)[WB]) {
        f32x4 (&acur)[NS] = (MB & 1) ? accb : acca;      // block MB, complete
        f32x4 (&anxt)[NS] = (MB & 1) ? acca : accb;      // block MB+1
        if (MB == 0 || !next_is_out) load_afrags<KS, FP>(x, nfrag0 + MB * KS, A[MB]);
        if constexpr (MB == 0 || SPREAD_PARKS) park_state<MB>(x, l, in);
        // accumulator start values (LDS table), requested a block step or more ahead of their use.  bb[m], m >= 1: block m of this
        // layer (bb[1] was requested during the previous layer); bb[0]: block 0 of the next layer.
        if constexpr (MB == 0) {
#pragma unroll
            for (int m = 2; m < WB; ++m) bb[m] = load_bias(x, l, m);
            bb[0] = load_bias(x, l + 1, 0);
        }
        if constexpr (MB + 1 < WB) {
            acc_init(bb[MB + 1], anxt);
            if constexpr (MB == 0) {
                if (!next_is_out) bb[1] = load_bias(x, l + 1, 1);       // behind its use just above
            }
            {
            fwd_ksteps<0, KS, KS>(A[MB + 1], in, anxt);
            fwd_valu<MB>(acur, out);
            }
            if constexpr (NS == 4 && NP == 2) interleave<NS * KS * P3, 2>();
        } else {
            // last block: the next layer's block 0 starts on the finished half of `out`
            acc_init(bb[0], anxt);
            {
            fwd_ksteps<0, KOVL, KS>(A[0], out, anxt);
            fwd_valu<MB>(acur, out);
            }
            __builtin_amdgcn_sched_barrier(0);
            operands_ready<KS>(out, KS - 1);
            fwd_ksteps<KOVL, KS, KS>(A[0], out, anxt);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MB + 1 < WB) fwd_step<MB + 1>(x, l, nfrag0, next_is_out, in, out, A, acca, accb, bb);
    }

    // reverse: vector part of block MB of weight layer L's transpose -- the activation below it.  acc = WS * (W_L Z_L) for the NS
    // streams; state (h, hdot_k) of this lane's point as packed 16-bit pairs sp[s] (fp16: consumed in place by mixed-precision FMAs)
    //   zbar = sd hbar - 2 h sum_k hdotbar_k hdot_k ,  zdotbar_k = sd hdotbar_k          (INF:131-133, gradient of TanhGrad)
    // sp: the state's high parts as packed 16-bit pairs; sl: its (unscaled) low parts where the layout carries them (STATE_LO).  The state
    // enters in full precision: v = hi + lo (one mixed-precision FMA per value on the fp16 path).
    template <int MB>
    static __device__ __forceinline__ void bwd_valu(f32x4 (&acc)[NS], const u32x2 (&sp)[NS], const u32x2 (&sl)[NS], u32x4 (&Zn)[NS][1][KS][NP], int c, int q) {
        float vals[NS][1][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float st[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if constexpr (STATE_LO && MixF16<Op>::value) {
                    st[s] = (r & 1) ? MixF16<Op>::template sum2<1>(sp[s][r >> 1], sl[s][r >> 1]) : MixF16<Op>::template sum2<0>(sp[s][r >> 1], sl[s][r >> 1]);
                } else {
                    st[s] = cvt16<Op>((uint16_t)((r & 1) ? (sp[s][r >> 1] >> 16) : (sp[s][r >> 1] & 0xffffu)));
                    if constexpr (STATE_LO) st[s] += cvt16<Op>((uint16_t)((r & 1) ? (sl[s][r >> 1] >> 16) : (sl[s][r >> 1] & 0xffffu)));
                }
            }
            const float h = st[0];
            const float sds = (1.0f - h * h) * INV_WS;
            float dot = 0.0f;
#pragma unroll
            for (int s = 1; s <= NT; ++s) {
                dot += acc[s][r] * st[s];
                vals[s][0][r] = sds * acc[s][r];
            }
            float zb = NS > 1 ? sds * acc[0][r] - (2.0f * INV_WS) * h * dot : sds * acc[0][r];
            if constexpr (SECOND) {
                // adjoint of h_tt = (1-h^2) z_tt - 2 h h_t z_t from post-activation state only:
                //   d h_tt / d z = -2 h h_tt - 2 h_t^2,   d h_tt / d z_t = -4 h h_t,   d h_tt / d z_tt = 1 - h^2        (PLATE:417-419)
                const float ht = st[3], htt = st[4];
                const float httb = acc[4][r] * INV_WS;
                vals[4][0][r] = sds * acc[4][r];
                vals[3][0][r] -= 4.0f * h * ht * httb;
                zb += httb * (-2.0f * h * htt - 2.0f * ht * ht);
            }
            vals[0][0][r] = zb;
        }
        CH::template emit<KS, MB>(Zn, vals, nullptr, WIDTH, c, q);
    }

    // this lane's state pairs of feature block MB from its records of an LDS image
    template <int MB>
    static __device__ __forceinline__ void state_from_image(const char* img, u32x2 (&sp)[NS]) {
#pragma unroll
        for (int s = 0; s < NS; ++s) sp[s] = *reinterpret_cast<const u32x2*>(img + (s * KS + (MB >> 1)) * 1024 + 8 * (MB & 1));
    }
    template <int MB>
    static __device__ __forceinline__ void state_from_frags(const u32x4 (&B)[NS][1][KS][NP], u32x2 (&sp)[NS]) {
#pragma unroll
        for (int s = 0; s < NS; ++s) sp[s] = u32x2{B[s][0][MB >> 1][0][(MB & 1) * 2 + 0], B[s][0][MB >> 1][0][(MB & 1) * 2 + 1]};
    }

    // the low parts of the same values: from the forward's registers (S_NL) or from the tile's parked low-part image (plain loads)
    // S1_BY_WG: feature blocks MB0 .. MB1 - 1 of S_1 of the job's tile from its inputs (the forward's own function: the same bits); they
    // wait in registers for the slot, which S_3 occupies until layer 2's barrier.  The last piece writes the high parts into the tile's
    // slot of S_1; the low parts go, as LO8 records (the parked layers' format), into the tile's S1LO image.
    template <int MB0, int MB1>
    static __device__ __forceinline__ void recompute_s1(const FusedArgs& a, const S1Job& sj, u32x4 (&S1)[NS][1][KS][NP]) {
        if constexpr (S1_HI_BY_WG) {
            const Ctx& x = *sj.x;
            first_mb<MB0, MB1>(a, x, sj.xin, S1);
            if constexpr (MB1 == WB) put_image<KS>(x.imgS(1), S1);
        }
        if constexpr (S1_BY_WG) {
            const Ctx& x = *sj.x;
            first_mb<MB0, MB1>(a, x, sj.xin, S1);
            // the low parts leave at once (dword MB of a stream's LO8 record is block MB's; S1LO was last read in the previous step's layer 1):
            // only the high parts wait in registers
#pragma unroll
            for (int mb = MB0; mb < MB1; ++mb)
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const u32x4& lo = S1[s][0][mb >> 1][NP - 1];
                    *reinterpret_cast<uint32_t*>(x.s1lo + s * 1024 + 4 * mb) = lo8_pack(lo[(mb & 1) * 2], lo[(mb & 1) * 2 + 1]);
                }
            if constexpr (MB1 == WB) put_image<KS>(x.imgS(1), S1);
        }
    }
    template <int MB>
    static __device__ __forceinline__ void lo_from_frags(const u32x4 (&B)[NS][1][KS][NP], u32x2 (&sl)[NS]) {
#pragma unroll
        for (int s = 0; s < NS; ++s) sl[s] = u32x2{B[s][0][MB >> 1][NP - 1][(MB & 1) * 2 + 0], B[s][0][MB >> 1][NP - 1][(MB & 1) * 2 + 1]};
    }
    template <int MB>
    static __device__ __forceinline__ void lo_from_scratch(const Ctx& x, int L /*1..NL-1*/, u32x2 (&sl)[NS]) {
        if constexpr (STATE_LO) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
                sl[s] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(x.scr, x.imgoff + 8u * (MB & 1),
                                                                                      SCRATCH_LO + (L - 1) * IMG_B + (s * KS + (MB >> 1)) * 1024, 0));
        }
    }

    // Reverse through weight layer L (KSB k-steps of its outputs; fragments frag0 + MB*KSB) and the activation that produced S_L:
    // Zf = Z_L  ->  Zn = Z_{L-1}.  Same software pipeline as the forward: MFMAs of block MB+1 between the vector part of block MB.
    // TOP: the state comes from the forward's registers (S_NL), otherwise from the wave's LDS image of S_L.
    //   entry: Aa = fragments of block 0, Ab = fragments of block 1, both loaded (issued before the hand-off barriers)
    template <int MB, int KSB, bool TOP, bool ZOUT = false>
    //   sla = low parts of the state for block 0 (requested with the first fragments); block MB+1's are requested in step MB
    //   ZOUT (ZDB): the high parts of Zn go to the adjoint image `zout` as soon as a fragment (two blocks) is complete
    static __device__ __forceinline__ void bwd_step(const Ctx& x, int frag0, int L, const char* img, const u32x4 (&Sreg)[NS][1][KS][NP], const u32x4 (&Zf)[NS][1][KSB][NP],
                                                    u32x4 (&Zn)[NS][1][KS][NP], u32x4 (&Aa)[KSB][RP], u32x4 (&Ab)[KSB][RP], f32x4 (&acca)[NS], f32x4 (&accb)[NS],
                                                    u32x2 (&sla)[NS], u32x2 (&slb)[NS], char* zout = nullptr, const u32x4 (*slq)[NS] = nullptr) {
        u32x2 (&slcur)[NS] = (MB & 1) ? slb : sla;
        u32x2 (&slnxt)[NS] = (MB & 1) ? sla : slb;
        if constexpr (!TOP && LO8) {                           // this block's four bytes of every stream -> two fp16 pairs
#pragma unroll
            for (int s = 0; s < NS; ++s) slcur[s] = slq != nullptr ? lo8_unpack((*slq)[s][MB]) : u32x2{0u, 0u};      // (nullptr: a layer without low-part record, LO_FROM)
        }
        u32x4 (&Acur)[KSB][RP] = (MB & 1) ? Ab : Aa;
        u32x4 (&Anxt)[KSB][RP] = (MB & 1) ? Aa : Ab;
        f32x4 (&acur)[NS] = (MB & 1) ? accb : acca;
        f32x4 (&anxt)[NS] = (MB & 1) ? acca : accb;
        if constexpr (MB + 2 < WB) load_afrags<KSB, RP>(x, frag0 + (MB + 2) * KSB, Acur);
        u32x2 sp[NS];
        if constexpr (TOP) {
            state_from_frags<MB>(Sreg, sp);
            if constexpr (STATE_LO) lo_from_frags<MB>(Sreg, slcur);
        } else {
            state_from_image<MB>(img, sp);
            if constexpr (MB + 1 < WB && !LO8) lo_from_scratch<MB + 1>(x, L, slnxt);
        }
        if constexpr (MB + 1 < WB) {
            acc_zero(anxt);
            bwd_ksteps<0, KSB, KSB>(Anxt, Zf, anxt);
        }
        bwd_valu<MB>(acur, sp, slcur, Zn, x.c, x.q);
        if constexpr (ZOUT && (MB & 1)) {
#pragma unroll
            for (int s = 0; s < NS; ++s) *reinterpret_cast<u32x4*>(zout + (s * KS + (MB >> 1)) * 1024) = Zn[s][0][MB >> 1][0];
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MB + 1 < WB) bwd_step<MB + 1, KSB, TOP, ZOUT>(x, frag0, L, img, Sreg, Zf, Zn, Aa, Ab, acca, accb, sla, slb, zout, slq);
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

    // S_0: the inputs as a 16-feature state (rows 0..2 = x', tangent stream k carries sx_k in row k) -> image slot 0
    static __device__ __forceinline__ void put_input_state(const FusedArgs& a, const Ctx& x, const float (&xin)[4]) {
        float v0[NS][1][4];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // rows 0..2: hi parts; rows 4..6 (lanes q == 1): the 2^11-scaled low parts of the same numbers, so that the
                // first layer's weight gradient keeps full input precision (combined at write-out)
                float v = 0.0f;
                if (x.q < 2 && r < DIN) {
                    const float full = (s == 0) ? xin[r] : ((s <= NT && r == s - 1) ? in_loop(a.sx[r]) : 0.0f);
                    v = x.q == 0 ? full : (full - round16<Op>(full)) * Op::LO_SCALE;
                }
                v0[s][0][r] = v;
            }
        u32x4 S0[NS][1][1][NP];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int p = 0; p < NP; ++p) S0[s][0][0][p] = u32x4{0u, 0u, 0u, 0u};
        CH::template emit<1, 0>(S0, v0, nullptr, 16, x.c, x.q);
        put_image<1>(x.imgS(0), S0);
    }

    // Weight layers L = NL-1 .. 1 (hidden-to-hidden) and finally L = 0, fully unrolled (static fragment indices / offsets).
    template <int L>
    struct Down {
        // entry: Zc = Z_L (adjoint of weight layer L's pre-activation) in chain fragment order; S_L is (being) DMA'd into its slot
        static __device__ __forceinline__ void run(const FusedArgs& a, const Ctx& x, const float (&xin)[4], const u32x4 (&Zc)[NS][1][KS][NP]) {
            u32x4 Aa[KS][RP], Ab[KS][RP];
            u32x2 sla[NS], slb[NS];
            u32x4 slq[NS];                                     // (LO8: the layer's low parts, one 16-byte record per stream)
            constexpr bool RECOMP = RECOMP1 && L == 1 && !S1_BY_WG;
            constexpr bool S1_LDS = S1_BY_WG && L == 1;        // S_1: high parts in its slot, low parts in S1LO, both from the weight-gradient wave
            u32x4 S1[NS][1][KS][NP];                           // (RECOMP only)
            if constexpr (L >= 1) {                            // this layer's first fragments travel during the hand-off
                load_afrags<KS, RP>(x, FI::bwd_mid(NL, L, 0, 0), Aa);
                load_afrags<KS, RP>(x, FI::bwd_mid(NL, L, 1, 0), Ab);
                if constexpr (S1_LDS) {
                    // (read behind the barrier below)
                } else if constexpr (!RECOMP && LO8 && !lo_layer(L)) {
                    // (no low-part record for this layer: LO_FROM)
                } else if constexpr (!RECOMP && LO8) {
#pragma unroll
                    for (int s = 0; s < NS; ++s) slq[s] = __builtin_amdgcn_raw_buffer_load_b128(x.scr, x.imgoff, SCRATCH_LO + (L - 1) * IMG_B + s * 1024, 0);
                } else if constexpr (!RECOMP) lo_from_scratch<0>(x, L, sla);
            }
            constexpr bool ONE_BARRIER = ZDB && L <= NL - 2;   // Z_L is in its buffer already: written by the reverse step that produced it
            if constexpr (!ONE_BARRIER) {
                hand_barrier();                                // previous layer's fragment reads are done
                fused_stamp(a, x.tracer, 3 + 3 * (NL - L));
                if constexpr (ZDB) put_zimage_hi<KS>(x.imgZ() + zbuf(L), Zc);
                else put_zimage<KS>(x.imgZ(), Zc);
            } else {
                fused_stamp(a, x.tracer, 3 + 3 * (NL - L));
            }
            // (ONE_BARRIER: the slots written here were last read a layer ago -- S_0 goes where S_2 was, S_1 where S_3 was)
            if constexpr (L == 0) put_input_state(a, x, xin);
            if constexpr (RECOMP) {                            // S_1 again from the inputs; its high parts into the layer's slot for the weight gradient
                first_mb<0>(a, x, xin, S1);
                put_image<KS>(x.imgS(1), S1);
            }
            hand_barrier();                                    // tensors visible to the weight-gradient waves; S_L has landed
            fused_stamp(a, x.tracer, 4 + 3 * (NL - L));
            if constexpr (S1_LDS) {
#pragma unroll
                for (int s = 0; s < NS; ++s) slq[s] = *reinterpret_cast<const u32x4*>(x.s1lo + s * 1024);
            }
            if constexpr (L >= 1) {
                // reverse through W_L and the activation that produced S_L -> Z_{L-1}
                u32x4 Zn[NS][1][KS][NP];
                f32x4 acca[NS], accb[NS];
                acc_zero(acca);
                bwd_ksteps<0, KS, KS>(Aa, Zc, acca);
                // (ZDB: Z_{L-1} goes to the other adjoint buffer while the weight-gradient waves read Z_L from this layer's)
                char* zout = x.imgZ() + zbuf(L - 1);
                if constexpr (RECOMP) bwd_step<0, KS, true, ZDB>(x, FI::bwd_mid(NL, L, 0, 0), L, nullptr, S1, Zc, Zn, Aa, Ab, acca, accb, sla, slb, zout);
                else bwd_step<0, KS, false, ZDB>(x, FI::bwd_mid(NL, L, 0, 0), L, x.imgS(L), Zc /*unused*/, Zc, Zn, Aa, Ab, acca, accb, sla, slb, zout, (LO8 && !lo_layer(L)) ? nullptr : &slq);
                pin<KS>(Zn);
                fused_stamp(a, x.tracer, 5 + 3 * (NL - L));
                Down<L - 1>::run(a, x, xin, Zn);
            }
        }
    };

    // ---- the forward in pieces ----
    // first layer + pipeline prologue: all fragments of layer 1 are requested BEFORE the first layer's vector work (they take a full
    // L2 round trip), then block 0's MFMAs
    static __device__ __forceinline__ void fwd_first(const FusedArgs& a, const Ctx& x, const float (&xin)[4], u32x4 (&B)[NS][1][KS][NP], u32x4 (&A)[WB][KS][FP],
                                                     f32x4 (&acca)[NS], f32x4 (&bb)[WB]) {
#pragma unroll
        for (int mb = 0; mb < WB; ++mb) load_afrags<KS, FP>(x, FI::fwd_mid(1, mb, 0), A[mb]);
        bb[0] = load_bias(x, 1, 0);
        bb[1] = load_bias(x, 1, 1);
        first_mb<0>(a, x, xin, B);
        acc_init(bb[0], acca);
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) operands_ready<KS>(B, kk);
        fwd_ksteps<0, KS, KS>(A[0], B, acca);
        __builtin_amdgcn_sched_barrier(0);
    }
    static __device__ __forceinline__ void fwd_layer(const Ctx& x, int l, const u32x4 (&in)[NS][1][KS][NP], u32x4 (&out)[NS][1][KS][NP], u32x4 (&A)[WB][KS][FP],
                                                     f32x4 (&acca)[NS], f32x4 (&accb)[NS], f32x4 (&bb)[WB]) {
        const bool last = l + 1 == NL;
        const int nfrag0 = last ? FI::fwd_last(NL, 0) : FI::fwd_mid(l + 1, 0, 0);
        fwd_step<0>(x, l, nfrag0, last, in, out, A, acca, accb, bb);
    }


    // ---------------------------------------------------------------------------------------------
    // LDSOP chain (padded width 96): state tensors live in the wave's LDS images and are read one k-step at a time
    // ---------------------------------------------------------------------------------------------
    static_assert(!LDSOP || (NL & 1) == 0, "LDSOP forward buffer parity: the last hidden layer (odd) writes slot 0 = slot_of(NL)");
    // operand image = records ((s * KS + kk) * NP + p) of 1 KB; this lane's 16 bytes sit at the (rotated) record offset inside each
    static __device__ __forceinline__ void op_load(const char* img, int kk, u32x4 (&Bk)[NS][1][1][NP]) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int p = 0; p < NP; ++p) Bk[s][0][0][p] = *reinterpret_cast<const u32x4*>(img + ((s * KS + kk) * NP + p) * 1024);
    }
    // A tile's chain is shared by two waves (tile = wave & 1, half = wave >> 1), three of the six feature blocks each: half 0 owns
    // blocks (0, 1 | 2), half 1 blocks (4, 5 | 3) -- a pair that fills one fragment record and a single block that fills half of
    // record 1.  The wave's results are local fragments Fl[s][0][0][p] = the pair's record, Fl[s][0][1][p].xy = the single block's
    // 8 bytes; block numbers enter as run-time offsets only, so both halves run the same code.
    static constexpr int HB = WB / 2;
    static constexpr int HR = (HB + 1) / 2;            // local fragment records of a half: its pairs, then (odd HB) the single block's half record
    // (Padded width 128: four blocks per half, 4h .. 4h+3 = the two records 2h, 2h+1.  Padded width 160 -- the reference's confined-domain
    // net, 6 x 140, CONF:891 --: five blocks per half, 4h .. 4h+3 = the records 2h, 2h+1, and block 8 + h = one half of record 4.)
    static __device__ __forceinline__ int half_block(int h, int j) {
        return WB == 4 ? 2 * h + j : WB == 8 ? 4 * h + j : (WB == 10 ? (j < 4 ? 4 * h + j : 8 + h) : (h ? (j < 2 ? 4 + j : 3) : j));
    }
    // workgroup barrier that orders LDS traffic only: the chain waves' park stores and fragment loads stay in flight across it
    // hand-off barriers of the chain waves: LDS traffic only, the fragment / low-part loads requested in front of them stay in flight
    static __device__ __forceinline__ void hand_barrier() {
        lds_barrier();
    }
    static __device__ __forceinline__ void lds_barrier() {
#if defined(__AMDGCN__)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#else
        __syncthreads();
#endif
    }
    static __device__ __forceinline__ void half_store(char* img, int h, const u32x4 (&Fl)[NS][1][HR][NP]) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                *reinterpret_cast<u32x4*>(img + ((s * KS + (WB == 4 ? h : 2 * h)) * NP + p) * 1024) = Fl[s][0][0][p];
                if constexpr (WB >= 8) *reinterpret_cast<u32x4*>(img + ((s * KS + 2 * h + 1) * NP + p) * 1024) = Fl[s][0][1][p];
                if constexpr (WB == 10) *reinterpret_cast<u32x2*>(img + ((s * KS + 4) * NP + p) * 1024 + 8 * h) = u32x2{Fl[s][0][2][p][0], Fl[s][0][2][p][1]};
                if constexpr (WB == 6) *reinterpret_cast<u32x2*>(img + ((s * KS + 1) * NP + p) * 1024 + 8 * h) = u32x2{Fl[s][0][1][p][0], Fl[s][0][1][p][1]};
            }
    }
    // The GEMM of a half: its HB blocks accumulate over the KS k-steps of the operand image, NIT = KS * HB items of (k-step, block).
    // Weight fragments come from memory (L2) and are requested far ahead: with a two-item distance every item stalled on its load
    // (round-2 phase stamps: 10 k cycles per forward layer for 3.5 k of matrix work).
    static constexpr int NIT = KS * HB;
    static __device__ __forceinline__ int item_frag(int frag_l0 /*fragment (block 0, k-step 0) of the layer*/, int h, int t) {
        return frag_l0 + half_block(h, t % HB) * KS + t / HB;
    }
    // A ring of RING slots runs over the items of all layers in sequence; the slot an item leaves is reloaded with the item RING places
    // ahead -- of this layer or the next.  (A whole layer of fragments in registers made the compiler spill them: a spill store waits
    // for its load, i.e. drains the queue.)  Forward: layer l's item 0 sits in slot ((l - 1) * NIT) % RING = 0 (l odd) or 3 (l even).
    // The chain waves issue no stores in the forward: vector-memory operations complete in order on one counter, and a park store in
    // the queue puts its write acknowledgement -- ~2 k cycles -- in front of the next fragment wait.  The weight-gradient waves, idle in
    // the forward, copy every finished state image from LDS to the scratch image instead (park_image).
    static constexpr int RING = (WB == 8 || WB == 4) ? 4 : (WB == 10 ? 5 : 6);
    static_assert(!LDSOP || (2 * NIT) % RING == 0, "ring phase repeats every two layers");
    template <int PAR /* l & 1 */>
    static __device__ __forceinline__ void fwd_request(const Ctx& x, int l, int h, int t /*item of layer l, may run past NIT*/, u32x4 (&Ar)[RING][1][FP]) {
        constexpr int T0 = PAR ? 0 : NIT % RING;
        if (t < NIT) load_afrags<1, FP>(x, item_frag(FI::fwd_mid(l, 0, 0), h, t), Ar[(T0 + t) % RING]);
        else if (l + 1 < NL) load_afrags<1, FP>(x, item_frag(FI::fwd_mid(l + 1, 0, 0), h, t - NIT), Ar[(T0 + t) % RING]);
        else if (t - NIT < KS) load_afrags<1, FP>(x, FI::fwd_last(NL, t - NIT), Ar[(T0 + t) % RING]);      // the output layer's k-steps
    }
    template <int PAR>
    static __device__ __forceinline__ void half_gemm_fwd(const Ctx& x, int l, int h, const char* in, u32x4 (&Ar)[RING][1][FP], f32x4 (&acc)[HB][NS]) {
        constexpr int T0 = PAR ? 0 : NIT % RING;
        u32x4 Bk[NS][1][1][NP];
#pragma unroll
        for (int t = 0; t < NIT; ++t) {
            if (t % HB == 0) op_load(in, t / HB, Bk);
            fwd_kstep<0, 1>(Ar[(T0 + t) % RING], Bk, acc[t % HB]);
            fwd_request<PAR>(x, l, h, t + RING, Ar);
        }
    }
    // reverse: the same kind of ring over the items of layers NL-1 .. 1 (T0 = global index of this layer's item 0)
    static constexpr int RINGB = (WB == 8 || WB == 4) ? 4 : (WB == 10 ? 5 : 6);      // (three fragment parts per item: 12 registers a slot)
    template <int L>
    static __device__ __forceinline__ void ring_request(const Ctx& x, int h, int t /*item of layer L, may run past NIT*/, u32x4 (&Ar)[RINGB][1][RP]) {
        constexpr int T0 = (NL - 1 - L) * NIT;
        if (t < NIT) load_afrags<1, RP>(x, item_frag(FI::bwd_mid(NL, L, 0, 0), h, t), Ar[(T0 + t) % RINGB]);
        else if (L >= 2) load_afrags<1, RP>(x, item_frag(FI::bwd_mid(NL, L >= 2 ? L - 1 : 1, 0, 0), h, t - NIT), Ar[(T0 + t) % RINGB]);
    }
    template <int L>
    static __device__ __forceinline__ void half_gemm_bwd(const Ctx& x, int h, const char* in, u32x4 (&Ar)[RINGB][1][RP], f32x4 (&acc)[HB][NS]) {
        constexpr int T0 = (NL - 1 - L) * NIT;
        u32x4 Bk[NS][1][1][NP];
#pragma unroll
        for (int t = 0; t < NIT; ++t) {
            if (t % HB == 0) op_load(in, t / HB, Bk);
            bwd_kstep<0, 1>(Ar[(T0 + t) % RINGB], Bk, acc[t % HB]);
            ring_request<L>(x, h, t + RINGB, Ar);
        }
    }
    template <int J>
    static __device__ __forceinline__ void wide_fwd_epilogue(const f32x4 (&acc)[HB][NS], u32x4 (&out)[NS][1][HR][NP]) {
        fwd_valu<J, HR>(acc[J], out);
        if constexpr (J + 1 < HB) wide_fwd_epilogue<J + 1>(acc, out);
    }
    // one hidden weight layer l (1..NL-1): S_l (image `in`) -> this half's blocks of S_{l+1} (image `out`, parked if a reverse layer will DMA it back)
    template <int PAR>
    static __device__ __forceinline__ void wide_fwd_layer(const Ctx& x, int l, int h, const char* in, char* outimg, u32x4 (&Af)[RING][1][FP]) {
        f32x4 acc[HB][NS];
        if constexpr (CONST_LDS) {
#pragma unroll
            for (int j = 0; j < HB; ++j) acc_init(load_bias(x, l, half_block(h, j)), acc[j]);
            half_gemm_fwd<PAR>(x, l, h, in, Af, acc);
        } else {
            // constants from memory: as the accumulators' start value the biases are consumed right behind their request -- a full L2 round
            // trip in front of every layer, behind everything the ring has in flight.  Requested here, added behind the layer's MFMAs.
            f32x4 bias[HB];
#pragma unroll
            for (int j = 0; j < HB; ++j) {
                bias[j] = load_bias(x, l, half_block(h, j));
                acc_zero(acc[j]);
            }
            half_gemm_fwd<PAR>(x, l, h, in, Af, acc);
#pragma unroll
            for (int j = 0; j < HB; ++j) acc[j][0] += bias[j];
        }
        u32x4 out[NS][1][HR][NP];
        wide_fwd_epilogue<0>(acc, out);
        half_store(outimg, h, out);
        if (kept_in_lds(l + 1)) half_store(x.imgS(l + 1), h, out);       // S_{NL-1} also into its reverse slot (KEEP_W)
    }
    // first layer (VALU) of local block J of half h
    template <int J>
    static __device__ __forceinline__ void first_block(const FusedArgs& a, const Ctx& x, const float (&xin)[4], int h, u32x4 (&Bn)[NS][1][HR][NP]) {
        const int mb = half_block(h, J);
        float vals[NS][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // first-layer row of feature 16 mb + 4 q + r: (W0[0..DIN-1][f], b0[f]); from the LDS constants or, where the tensors fill the
            // LDS (five streams at padded width 128), from memory
            float w[5];
            if constexpr (DIN == 4) {
                const f32x4 wa = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x.w0p, (unsigned)x.q * 128u, (16 * mb + r) * 32, 0));
                const f32x4 wb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x.w0p, (unsigned)x.q * 128u, (16 * mb + r) * 32 + 16, 0));
                w[0] = wa[0]; w[1] = wa[1]; w[2] = wa[2]; w[3] = wa[3]; w[4] = wb[0];
            } else {
                f32x4 wa;
                if constexpr (CONST_LDS) wa = *reinterpret_cast<const f32x4*>(x.cw0 + (16 * mb + r) * 16);
                else wa = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x.w0p, (unsigned)x.q * 64u, (16 * mb + r) * 16, 0));
                w[0] = wa[0]; w[1] = wa[1]; w[2] = wa[2]; w[3] = 0.0f; w[4] = wa[3];
            }
            float z0 = w[4];
#pragma unroll
            for (int k = 0; k < DIN; ++k) z0 += w[k] * xin[k];
            float hh, sd;
            tanh_act(z0, hh, sd);
            vals[0][r] = hh;
#pragma unroll
            for (int s = 1; s <= NT; ++s) vals[s][r] = sd * (a.sx[s - 1] * w[s - 1]);
            if constexpr (SECOND) vals[4][r] = -2.0f * hh * vals[3][r] * (a.sx[2] * w[2]);      // z_tt = 0 at the first layer
        }
        emit_state<J, HR>(Bn, vals);
    }
    template <int J>
    static __device__ __forceinline__ void wide_first(const FusedArgs& a, const Ctx& x, const float (&xin)[4], int h, u32x4 (&Bn)[NS][1][HR][NP]) {
        first_block<J>(a, x, xin, h, Bn);
        if constexpr (J + 1 < HB) wide_first<J + 1>(a, x, xin, h, Bn);
    }
    // ---------------------------------------------------------------------------------------------
    // FWD8 (round 6; padded width 96: the reference's 8 x 80 net, INF:645, and the plate's 8 x 70, PLATE:885)
    // ---------------------------------------------------------------------------------------------
    // These two layouts are not bound by bytes (ablations, profiles/r06_footprint_and_cache_policy.txt section 6: images -13 %, sums -4 %, fragments
    // -5 %); 37 % of their step is the forward (profiles/r06_phase_trace_wide.txt: 36.7 k of 99.5 k cycles), in which the four chain waves run at 38 %
    // of their matrix pipe while the four weight-gradient waves only park images.  The state lives in LDS images anyway, so the forward's feature
    // blocks are dealt to all EIGHT waves: a chain wave keeps the PAIR of its half (blocks 0, 1 | 4, 5: one fragment record), the weight-gradient
    // wave (tile = quad & 1, half = quad >> 1) computes the half's SINGLE block (2 | 3: half of record 1).  The reverse is unchanged (three blocks
    // per chain half).  Barriers: one per layer for all eight waves, as before; the weight-gradient wave parks S_l BEHIND its block of layer l.
    // Measured (profiles/r06_fwd8_ab.txt): 8 x 80 -1.9 ... -2.7 %, plate 8 x 70 -1.9 ... -4.8 % of the launch on two boxes -- a third of what the
    // block count promised: the forward's 4.4 k cycles per layer did NOT move with two blocks instead of three (nor with a second layer of fragment
    // distance, nor much without the parks: 3.7 k), i.e. a forward layer of these layouts is bound by its barrier-to-barrier latency chain (operand
    // reads, dependent MFMA chains, the tanh epilogue, the exchange barrier), not by the chain wave's issue; what the step gains (100 k -> 94 k
    // cycles) comes from the weight-gradient role starting its reverse with warm fragments and a shorter top of the reverse.
    static constexpr bool FWD8 = LDSOP && !WSLDS && WB == 6 && PINN_FWD8_ENABLED != 0;
    template <int J0, int NJ>            // a wave's forward block set: local blocks J0 .. J0 + NJ - 1 of half h
    struct F8 {
        static constexpr int NITJ = KS * NJ;                 // items (k-step, block) of a layer
#ifndef PINN_F8_DEPTH
#define PINN_F8_DEPTH 1
#endif
        // fragment slots in flight: PINN_F8_DEPTH whole layers of this wave's items.  (Two layers of distance -- the dealt forward leaves the
        // registers for it -- measured 0.5-4 % SLOWER than one: profiles/r06_fwd8_ab.txt; the forward layer is not waiting for its fragments.)
        static constexpr int DEPTH = PINN_F8_DEPTH;
        static constexpr int RINGJ = DEPTH * NITJ;
        static_assert(DEPTH == 1 || DEPTH == 2, "ring phase: item 0 of layer l in slot 0 (l odd) or NITJ (l even, DEPTH 2)");
        static __device__ __forceinline__ int frag(int frag_l0, int h, int t) { return frag_l0 + half_block(h, J0 + t % NJ) * KS + t / NJ; }
        // PAR = l & 1; t = item of layer l, may run DEPTH layers past it
        template <int PAR>
        static __device__ __forceinline__ void request(const Ctx& x, int l, int h, int t, u32x4 (&Ar)[RINGJ][1][FP]) {
            constexpr int T0 = (PAR || DEPTH == 1) ? 0 : NITJ;
            const int ll = l + t / NITJ, tt = t % NITJ;
            u32x4 (&slot)[1][FP] = Ar[(T0 + t) % RINGJ];
            if (ll < NL) load_afrags<1, FP>(x, frag(FI::fwd_mid(ll, 0, 0), h, tt), slot);
            else if (J0 == 0 && ll == NL && tt < KS) load_afrags<1, FP>(x, FI::fwd_last(NL, tt), slot);      // the output layer's k-steps (chain waves)
        }
        template <int PAR>
        static __device__ __forceinline__ void gemm(const Ctx& x, int l, int h, const char* in, u32x4 (&Ar)[RINGJ][1][FP], f32x4 (&acc)[NJ][NS]) {
            constexpr int T0 = (PAR || DEPTH == 1) ? 0 : NITJ;
            u32x4 Bk[NS][1][1][NP];
#pragma unroll
            for (int t = 0; t < NITJ; ++t) {
                if (t % NJ == 0) op_load(in, t / NJ, Bk);
                fwd_kstep<0, 1>(Ar[(T0 + t) % RINGJ], Bk, acc[t % NJ]);
                request<PAR>(x, l, h, t + RINGJ, Ar);
            }
        }
        // the set's part of the local fragments -> the image: the pair's record 2h, the single block's 8 bytes of record 1
        static __device__ __forceinline__ void store(char* img, int h, const u32x4 (&Fl)[NS][1][HR][NP]) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    if constexpr (J0 == 0) *reinterpret_cast<u32x4*>(img + ((s * KS + 2 * h) * NP + p) * 1024) = Fl[s][0][0][p];
                    if constexpr (J0 + NJ > 2) *reinterpret_cast<u32x2*>(img + ((s * KS + 1) * NP + p) * 1024 + 8 * h) = u32x2{Fl[s][0][1][p][0], Fl[s][0][1][p][1]};
                }
        }
        static __device__ __forceinline__ void first(const FusedArgs& a, const Ctx& x, const float (&xin)[4], int h, u32x4 (&Bn)[NS][1][HR][NP]) {
            first_block<J0>(a, x, xin, h, Bn);
            if constexpr (NJ == 2) first_block<J0 + 1>(a, x, xin, h, Bn);
        }
        // one hidden weight layer l: S_l (image `in`) -> this set's blocks of S_{l+1} (image `outimg`)
        template <int PAR>
        static __device__ __forceinline__ void layer(const Ctx& x, int l, int h, const char* in, char* outimg, u32x4 (&Af)[RINGJ][1][FP]) {
            f32x4 acc[NJ][NS];
            if constexpr (CONST_LDS) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc_init(load_bias(x, l, half_block(h, J0 + j)), acc[j]);
                gemm<PAR>(x, l, h, in, Af, acc);
            } else {
                f32x4 bias[NJ];          // (constants from memory: requested here, added behind the layer's MFMAs -- see wide_fwd_layer)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    bias[j] = load_bias(x, l, half_block(h, J0 + j));
                    acc_zero(acc[j]);
                }
                gemm<PAR>(x, l, h, in, Af, acc);
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[j][0] += bias[j];
            }
            u32x4 out[NS][1][HR][NP];
            fwd_valu<J0, HR>(acc[0], out);
            if constexpr (NJ == 2) fwd_valu<J0 + 1, HR>(acc[1], out);
            store(outimg, h, out);
            if (kept_in_lds(l + 1)) store(x.imgS(l + 1), h, out);       // S_{NL-1} also into its reverse slot (KEEP_W)
        }
    };
    typedef F8<0, 2> F8P;      // chain wave: the pair
    typedef F8<2, 1> F8S;      // weight-gradient wave: the single block
    // chain wave's forward under FWD8: wide_forward with the pair only
    static __device__ __forceinline__ void f8_chain_forward(const FusedArgs& a, const Ctx& x, const float (&xin)[4], int h, f32x4 (&acca)[NS]) {
        char* opa = x.tenZ + x.imgoff;
        char* opb = x.tenZ + TENSOR_Z_B + x.imgoff;
        u32x4 Af[F8P::RINGJ][1][FP];
#pragma unroll
        for (int t = 0; t < F8P::RINGJ; ++t) F8P::template request<1>(x, 1, h, t, Af);
        {
            u32x4 S1[NS][1][HR][NP];
            F8P::first(a, x, xin, h, S1);
            F8P::store(opa, h, S1);
        }
        lds_barrier();
        fused_stamp(a, x.tracer, 32);
        for (int l = 1; l < NL; ++l) {
            if (l & 1) F8P::template layer<1>(x, l, h, opa, opb, Af);
            else F8P::template layer<0>(x, l, h, opb, opa, Af);
            lds_barrier();
            fused_stamp(a, x.tracer, 32 + l);
        }
        acc_init(load_bias(x, NL, 0), acca);
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            u32x4 Bk[NS][1][1][NP];
            op_load(opb, kk, Bk);
            fwd_kstep<0, 1>(Af[((NL - 1) * F8P::NITJ + kk) % F8P::RINGJ], Bk, acca);      // (requested during the last hidden layers)
        }
    }
    // weight-gradient wave's forward under FWD8: its half's single block of every layer, and the park of the finished images as before
    static __device__ __forceinline__ void f8_wg_forward(const FusedArgs& a, const Ctx& x, const float (&xin)[4], int h, __amdgpu_buffer_rsrc_t scr_st, unsigned lane16,
                                                         const char* tile_lds, int quad) {
        char* opa = x.tenZ + x.imgoff;
        char* opb = x.tenZ + TENSOR_Z_B + x.imgoff;
        u32x4 Af[F8S::RINGJ][1][FP];
#pragma unroll
        for (int t = 0; t < F8S::RINGJ; ++t) F8S::template request<1>(x, 1, h, t, Af);
        {
            u32x4 S1[NS][1][HR][NP];
            F8S::first(a, x, xin, h, S1);
            F8S::store(opa, h, S1);
        }
        lds_barrier();                                       // S_1 complete
        for (int l = 1; l < NL; ++l) {
            if (l & 1) F8S::template layer<1>(x, l, h, opa, opb, Af);
            else F8S::template layer<0>(x, l, h, opb, opa, Af);
            if (!kept_in_lds(l)) park_image(scr_st, lane16, tile_lds, l, quad);      // S_l: complete since the previous barrier, read by this layer, overwritten behind the next one
            lds_barrier();                                   // S_{l+1} complete
        }
    }

    // forward of one tile (this wave's half): returns the output layer's products (acca, both halves compute them) for fwd_head; S_NL
    // ends in the second buffer.  One LDS barrier behind every layer: the halves exchange their blocks through the image.
    static __device__ __forceinline__ void wide_forward(const FusedArgs& a, const Ctx& x, const float (&xin)[4], int h, f32x4 (&acca)[NS]) {
        char* opa = x.tenZ + x.imgoff;                       // Z area
        char* opb = x.tenZ + TENSOR_Z_B + x.imgoff;          // S slot 0 = slot_of(NL): S_NL ends where the reverse expects it
        u32x4 Af[RING][1][FP];
#pragma unroll
        for (int t = 0; t < RING; ++t) fwd_request<1>(x, 1, h, t, Af);
        {
            u32x4 S1[NS][1][HR][NP];
            wide_first<0>(a, x, xin, h, S1);
            half_store(opa, h, S1);
            if constexpr (WSLDS) half_store(x.imgS(1), h, S1);       // every state keeps its own slot for the reverse
        }
        lds_barrier();
        fused_stamp(a, x.tracer, 32);
        for (int l = 1; l < NL; ++l) {                       // odd layers: first -> second buffer, even layers back
            if (l & 1) wide_fwd_layer<1>(x, l, h, opa, opb, Af);
            else wide_fwd_layer<0>(x, l, h, opb, opa, Af);
            lds_barrier();
            fused_stamp(a, x.tracer, 32 + l);
        }
        // output layer (16 padded outputs): one block, operand S_NL from slot 0 (NL - 1 is odd)
        acc_init(load_bias(x, NL, 0), acca);
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            u32x4 Bk[NS][1][1][NP];
            op_load(opb, kk, Bk);
            fwd_kstep<0, 1>(Af[(((NL - 1) * NIT) % RING + kk) % RING], Bk, acca);       // requested during the last hidden layer
        }
    }
    // reverse vector part with the state in full precision (hi + unscaled lo from the operand-layout image)
    template <int J>
    static __device__ __forceinline__ void wide_bwd_epilogue(f32x4 (&acc)[HB][NS], const char* simg, int h, u32x4 (&Zn)[NS][1][HR][NP], int c, int q) {
        const int mb = half_block(h, J);
        float st[NS][4];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const char* rec = simg + ((s * KS + (mb >> 1)) * NP) * 1024 + 8 * (mb & 1);
            const u32x2 hi = *reinterpret_cast<const u32x2*>(rec);
            const u32x2 lo = *reinterpret_cast<const u32x2*>(rec + 1024);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (MixF16<Op>::value)
                    st[s][r] = (r & 1) ? MixF16<Op>::template sum2<1>(hi[r >> 1], lo[r >> 1]) : MixF16<Op>::template sum2<0>(hi[r >> 1], lo[r >> 1]);
                else
                    st[s][r] = cvt16<Op>((uint16_t)((r & 1) ? (hi[r >> 1] >> 16) : (hi[r >> 1] & 0xffffu))) +
                               cvt16<Op>((uint16_t)((r & 1) ? (lo[r >> 1] >> 16) : (lo[r >> 1] & 0xffffu)));
            }
        }
        float vals[NS][1][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float hh = st[0][r];
            const float sds = (1.0f - hh * hh) * INV_WS;
            float dot = 0.0f;
#pragma unroll
            for (int s = 1; s <= NT; ++s) {
                dot += acc[J][s][r] * st[s][r];
                vals[s][0][r] = sds * acc[J][s][r];
            }
            float zb = sds * acc[J][0][r] - (2.0f * INV_WS) * hh * dot;
            if constexpr (SECOND) {                      // adjoint of h_tt = (1-h^2) z_tt - 2 h h_t z_t, as in bwd_valu (PLATE:417-419)
                const float ht = st[3][r], htt = st[4][r];
                const float httb = acc[J][4][r] * INV_WS;
                vals[4][0][r] = sds * acc[J][4][r];
                vals[3][0][r] -= 4.0f * hh * ht * httb;
                zb += httb * (-2.0f * hh * htt - 2.0f * ht * ht);
            }
            vals[0][0][r] = zb;
        }
        CH::template emit<HR, J>(Zn, vals, nullptr, WIDTH, c, q);
        if constexpr (J + 1 < HB) wide_bwd_epilogue<J + 1>(acc, simg, h, Zn, c, q);
    }
    // reverse of one tile; on entry the first barrier of the top layer has NOT been passed, S_NL (hi + lo) sits in S slot 0
    static __device__ __forceinline__ void wide_reverse(const FusedArgs& a, const Ctx& x, const float (&xin)[4], int h, const u32x4 (&ZL)[NS][1][1][NP]) {
        u32x4 Zn[NS][1][HR][NP];
        u32x4 Ar[RINGB][1][RP];
        u32x4 At[HB][1][RP];
        // (the fence keeps these requests behind the head's vector work: a spill reload in there would otherwise wait for all of them)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < HB; ++j) load_afrags<1, RP>(x, FI::bwd_last(NL, half_block(h, j)), At[j]);
#pragma unroll
        for (int t = 0; t < RINGB; ++t) ring_request<NL - 1>(x, h, t, Ar);
        {
            fused_stamp(a, x.tracer, 2);
            lds_barrier();                                      // A(NL); S_NL already sits in slot_of(NL) in the image layout
            fused_stamp(a, x.tracer, 3);
            if (h == 0) put_zimage<1>(x.imgZ(), ZL);           // (both halves hold the same Z_NL)
            lds_barrier();                                      // B(NL)
            fused_stamp(a, x.tracer, 4);
            f32x4 acc[HB][NS];
#pragma unroll
            for (int j = 0; j < HB; ++j) {
                acc_zero(acc[j]);
                bwd_kstep<0, 1>(At[j], ZL, acc[j]);
            }
            wide_bwd_epilogue<0>(acc, x.imgS(NL), h, Zn, x.c, x.q);
            fused_stamp(a, x.tracer, 5);
        }
        wide_down<NL - 1>(a, x, xin, h, Zn, Ar);
    }
    // entry: Zc = this half's blocks of Z_L in registers, first barrier of layer L not yet passed
    template <int L>
    static __device__ __forceinline__ void wide_down(const FusedArgs& a, const Ctx& x, const float (&xin)[4], int h, const u32x4 (&Zc)[NS][1][HR][NP],
                                                     u32x4 (&Ar)[RINGB][1][RP]) {
        lds_barrier();                                          // A(L): the weight-gradient waves are done with Z_{L+1}, S_{L+1}
        fused_stamp(a, x.tracer, 3 + 3 * (NL - L));
        half_store(x.imgZ(), h, Zc);
        if constexpr (L == 0) {
            if (h == 0) put_input_state(a, x, xin);
        }
        if constexpr (RECOMP_W && L == 1) {                     // S_1 again from the inputs, this half's blocks (the forward's own function)
            u32x4 S1[NS][1][HR][NP];
            wide_first<0>(a, x, xin, h, S1);
            half_store(x.imgS(1), h, S1);
        }
        lds_barrier();                                          // B(L)
        fused_stamp(a, x.tracer, 4 + 3 * (NL - L));
        if constexpr (L >= 1) {
            f32x4 acc[HB][NS];
#pragma unroll
            for (int j = 0; j < HB; ++j) acc_zero(acc[j]);
            half_gemm_bwd<L>(x, h, x.imgZ(), Ar, acc);           // operand: the Z_L image both halves have just written
            u32x4 Zn[NS][1][HR][NP];
            wide_bwd_epilogue<0>(acc, x.imgS(L), h, Zn, x.c, x.q);
            fused_stamp(a, x.tracer, 5 + 3 * (NL - L));
            wide_down<L - 1>(a, x, xin, h, Zn, Ar);
        }
    }

    // ---------------------------------------------------------------------------------------------
    // QUAD chain (round 6; padded width 128: the reference's 8 x 100 net, SEMI:679, and the 3-D net of BASELINE configs[4])
    // ---------------------------------------------------------------------------------------------
    // What bounds the chain of these layouts is its weight-fragment traffic: with the points as the MFMA's N dimension a 1 KB fragment feeds
    // 15 MFMAs (five streams x three parts) of ONE 16-point tile, and the two waves that own the same half of the feature blocks of the two
    // tiles load the same 96 KB of a layer's fragments twice -- 4 MB through L2 per 32-point workgroup step, against a ring of four slots in
    // flight per wave.  Ablations on the 3-D kernel (profiles/r06_lds_operand_ablations.txt): no fragment loads at all -20 % of the launch,
    // every second one skipped -9 %.  QUAD: a chain wave owns a QUARTER of the feature blocks (blocks 2 qt, 2 qt + 1 = fragment record qt of
    // an image) of BOTH tiles, so every fragment is loaded once per workgroup and feeds 30 MFMAs; the same ring then covers twice the matrix
    // work per byte in flight.  Accumulators, operand fragments and results per wave stay what they were (2 tiles x 2 blocks instead of 1 x 4);
    // LDS operand reads double (every wave reads both tiles' images: 80 of 1 KB per layer, 5 % of the LDS's time).  The head keeps its
    // assignment (wave w: tile w & 1, both waves of a tile compute it), so do the barriers: the weight-gradient role does not change.
    static_assert(!QUAD || (ONE_SLOT && TILES == 2), "QUAD: the one-slot layouts of padded width 128");
    static constexpr int QB = 2;                        // feature blocks per chain wave
    static constexpr int NITQ = KS * QB;                // items (k-step, block) of a layer; an item's fragment serves both tiles
    static constexpr int RINGQ = 4;                     // fragment slots in flight: two k-steps' pairs
    static_assert(!QUAD || NITQ % RINGQ == 0, "the ring's phase is the same in every layer");
    static __device__ __forceinline__ int q_frag(int frag_l0, int qt, int t) { return frag_l0 + (2 * qt + (t & 1)) * KS + (t >> 1); }
    static __device__ __forceinline__ void q_fwd_request(const Ctx& x, int l, int qt, int t /*item of layer l, may run past NITQ*/, u32x4 (&Ar)[RINGQ][1][FP]) {
        if (t < NITQ) load_afrags<1, FP>(x, q_frag(FI::fwd_mid(l, 0, 0), qt, t), Ar[t % RINGQ]);
        else if (l + 1 < NL) load_afrags<1, FP>(x, q_frag(FI::fwd_mid(l + 1, 0, 0), qt, t - NITQ), Ar[t % RINGQ]);
        else if (t - NITQ < KS) load_afrags<1, FP>(x, FI::fwd_last(NL, t - NITQ), Ar[t % RINGQ]);      // the output layer's k-steps
    }
    template <int L>
    static __device__ __forceinline__ void q_bwd_request(const Ctx& x, int qt, int t, u32x4 (&Ar)[RINGQ][1][RP]) {
        if (t < NITQ) load_afrags<1, RP>(x, q_frag(FI::bwd_mid(NL, L, 0, 0), qt, t), Ar[t % RINGQ]);
        else if (L >= 2) load_afrags<1, RP>(x, q_frag(FI::bwd_mid(NL, L >= 2 ? L - 1 : 1, 0, 0), qt, t - NITQ), Ar[t % RINGQ]);
    }
    // this lane's record of tile t's Z area / S slot of layer L
    static __device__ __forceinline__ char* q_imgZ(const Ctx& x, int t) { return x.lds0 + t * WAVE_B + x.imgoff; }
    static __device__ __forceinline__ char* q_imgS(const Ctx& x, int t, int L) { return x.lds0 + t * WAVE_B + TENSOR_Z_B + slot_of(L) * IMG_B + x.imgoff; }
    // fragment record qt (this wave's two blocks) of every stream and part -> the image
    static __device__ __forceinline__ void q_store(char* img, int qt, const u32x4 (&F)[NS][1][1][NP]) {
        char* rec = img + qt * NP * 1024;
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(rec + (s * KS * NP + p) * 1024) = F[s][0][0][p];
    }
    static __device__ __forceinline__ void q_gemm_fwd(const Ctx& x, int l, int qt, const char* in0, const char* in1, u32x4 (&Ar)[RINGQ][1][FP], f32x4 (&acc)[2][QB][NS]) {
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                u32x4 Bk[NS][1][1][NP];
                op_load(t ? in1 : in0, kk, Bk);
#pragma unroll
                for (int j = 0; j < QB; ++j) fwd_kstep<0, 1>(Ar[(QB * kk + j) % RINGQ], Bk, acc[t][j]);
            }
#pragma unroll
            for (int j = 0; j < QB; ++j) q_fwd_request(x, l, qt, QB * kk + j + RINGQ, Ar);
        }
    }
    template <int L>
    static __device__ __forceinline__ void q_gemm_bwd(const Ctx& x, int qt, const char* in0, const char* in1, u32x4 (&Ar)[RINGQ][1][RP], f32x4 (&acc)[2][QB][NS]) {
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                u32x4 Bk[NS][1][1][NP];
                op_load(t ? in1 : in0, kk, Bk);
#pragma unroll
                for (int j = 0; j < QB; ++j) bwd_kstep<0, 1>(Ar[(QB * kk + j) % RINGQ], Bk, acc[t][j]);
            }
#pragma unroll
            for (int j = 0; j < QB; ++j) q_bwd_request<L>(x, qt, QB * kk + j + RINGQ, Ar);
        }
    }
    // first layer (VALU) of this wave's blocks for one tile: wide_first with the quarter's block numbers
    template <int J>
    static __device__ __forceinline__ void q_first(const FusedArgs& a, const Ctx& x, const float (&xin)[4], int qt, u32x4 (&Bn)[NS][1][1][NP]) {
        const int mb = 2 * qt + J;
        float vals[NS][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float w[5];
            if constexpr (DIN == 4) {
                const f32x4 wa = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x.w0p, (unsigned)x.q * 128u, (16 * mb + r) * 32, 0));
                const f32x4 wb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x.w0p, (unsigned)x.q * 128u, (16 * mb + r) * 32 + 16, 0));
                w[0] = wa[0]; w[1] = wa[1]; w[2] = wa[2]; w[3] = wa[3]; w[4] = wb[0];
            } else {
                f32x4 wa;
                if constexpr (CONST_LDS) wa = *reinterpret_cast<const f32x4*>(x.cw0 + (16 * mb + r) * 16);
                else wa = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x.w0p, (unsigned)x.q * 64u, (16 * mb + r) * 16, 0));
                w[0] = wa[0]; w[1] = wa[1]; w[2] = wa[2]; w[3] = 0.0f; w[4] = wa[3];
            }
            float z0 = w[4];
#pragma unroll
            for (int k = 0; k < DIN; ++k) z0 += w[k] * xin[k];
            float hh, sd;
            tanh_act(z0, hh, sd);
            vals[0][r] = hh;
#pragma unroll
            for (int s = 1; s <= NT; ++s) vals[s][r] = sd * (a.sx[s - 1] * w[s - 1]);
            if constexpr (SECOND) vals[4][r] = -2.0f * hh * vals[3][r] * (a.sx[2] * w[2]);
        }
        emit_state<J, 1>(Bn, vals);
        if constexpr (J + 1 < QB) q_first<J + 1>(a, x, xin, qt, Bn);
    }
    // one hidden weight layer l: S_l (images in0 / in1 of the two tiles) -> this wave's blocks of S_{l+1} of both tiles (images out0 / out1)
    static __device__ __forceinline__ void q_fwd_layer(const Ctx& x, int l, int qt, const char* in0, const char* in1, char* out0, char* out1, u32x4 (&Af)[RINGQ][1][FP]) {
        f32x4 acc[2][QB][NS];
        f32x4 bias[QB];
#pragma unroll
        for (int j = 0; j < QB; ++j) {
            bias[j] = load_bias(x, l, 2 * qt + j);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if constexpr (CONST_LDS) acc_init(bias[j], acc[t][j]);
                else acc_zero(acc[t][j]);       // (constants from memory: requested here, added behind the layer's MFMAs -- see wide_fwd_layer)
            }
        }
        q_gemm_fwd(x, l, qt, in0, in1, Af, acc);
        if constexpr (!CONST_LDS) {
#pragma unroll
            for (int j = 0; j < QB; ++j)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t][j][0] += bias[j];
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            u32x4 out[NS][1][1][NP];
            fwd_valu<0, 1>(acc[t][0], out);
            fwd_valu<1, 1>(acc[t][1], out);
            q_store(t ? out1 : out0, qt, out);
        }
    }
    // forward of both tiles (this wave's quarter of the blocks); returns the output layer's products of the wave's HEAD tile (x) for fwd_head
    static __device__ __forceinline__ void quad_forward(const FusedArgs& a, const Ctx& x, const float (&xt)[2][4], int qt, f32x4 (&acca)[NS]) {
        char* opa0 = q_imgZ(x, 0);                            // Z areas
        char* opa1 = q_imgZ(x, 1);
        char* opb0 = opa0 + TENSOR_Z_B;                       // S slot 0 = slot_of(NL): S_NL ends where the reverse expects it
        char* opb1 = opa1 + TENSOR_Z_B;
        u32x4 Af[RINGQ][1][FP];
#pragma unroll
        for (int t = 0; t < RINGQ; ++t) q_fwd_request(x, 1, qt, t, Af);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            u32x4 S1[NS][1][1][NP];
            q_first<0>(a, x, xt[t], qt, S1);
            q_store(t ? opa1 : opa0, qt, S1);
        }
        lds_barrier();
        fused_stamp(a, x.tracer, 32);
        for (int l = 1; l < NL; ++l) {                        // odd layers: Z area -> slot 0, even layers back
            if (l & 1) q_fwd_layer(x, l, qt, opa0, opa1, opb0, opb1, Af);
            else q_fwd_layer(x, l, qt, opb0, opb1, opa0, opa1, Af);
            lds_barrier();
            fused_stamp(a, x.tracer, 32 + l);
        }
        // output layer (16 padded outputs) of the head tile: one block, operand S_NL from slot 0; fragments requested during the last hidden layer
        acc_init(load_bias(x, NL, 0), acca);
        const char* opb = x.tenZ + TENSOR_Z_B + x.imgoff;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            u32x4 Bk[NS][1][1][NP];
            op_load(opb, kk, Bk);
            fwd_kstep<0, 1>(Af[kk % RINGQ], Bk, acca);
        }
    }
    // reverse vector part of block J of this wave's quarter for one tile (state in full precision from the operand-layout image)
    template <int J>
    static __device__ __forceinline__ void q_bwd_epilogue(f32x4 (&acc)[QB][NS], const char* simg, int qt, u32x4 (&Zn)[NS][1][1][NP], int c, int q) {
        const int mb = 2 * qt + J;
        float st[NS][4];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const char* rec = simg + ((s * KS + (mb >> 1)) * NP) * 1024 + 8 * (mb & 1);
            const u32x2 hi = *reinterpret_cast<const u32x2*>(rec);
            const u32x2 lo = *reinterpret_cast<const u32x2*>(rec + 1024);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (MixF16<Op>::value)
                    st[s][r] = (r & 1) ? MixF16<Op>::template sum2<1>(hi[r >> 1], lo[r >> 1]) : MixF16<Op>::template sum2<0>(hi[r >> 1], lo[r >> 1]);
                else
                    st[s][r] = cvt16<Op>((uint16_t)((r & 1) ? (hi[r >> 1] >> 16) : (hi[r >> 1] & 0xffffu))) +
                               cvt16<Op>((uint16_t)((r & 1) ? (lo[r >> 1] >> 16) : (lo[r >> 1] & 0xffffu)));
            }
        }
        float vals[NS][1][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float hh = st[0][r];
            const float sds = (1.0f - hh * hh) * INV_WS;
            float dot = 0.0f;
#pragma unroll
            for (int s = 1; s <= NT; ++s) {
                dot += acc[J][s][r] * st[s][r];
                vals[s][0][r] = sds * acc[J][s][r];
            }
            float zb = sds * acc[J][0][r] - (2.0f * INV_WS) * hh * dot;
            if constexpr (SECOND) {
                const float ht = st[3][r], htt = st[4][r];
                const float httb = acc[J][4][r] * INV_WS;
                vals[4][0][r] = sds * acc[J][4][r];
                vals[3][0][r] -= 4.0f * hh * ht * httb;
                zb += httb * (-2.0f * hh * htt - 2.0f * ht * ht);
            }
            vals[0][0][r] = zb;
        }
        CH::template emit<1, J>(Zn, vals, nullptr, WIDTH, c, q);
        if constexpr (J + 1 < QB) q_bwd_epilogue<J + 1>(acc, simg, qt, Zn, c, q);
    }
    // entry: Zc[t] = this wave's blocks of Z_L of tile t in registers, first barrier of layer L not yet passed
    template <int L>
    static __device__ __forceinline__ void quad_down(const FusedArgs& a, const Ctx& x, const float (&xt)[2][4], int qt, int half, const u32x4 (&Zc)[2][NS][1][1][NP],
                                                     u32x4 (&Ar)[RINGQ][1][RP]) {
        lds_barrier();                                          // A(L): the weight-gradient waves are done with Z_{L+1}, S_{L+1}
        fused_stamp(a, x.tracer, 3 + 3 * (NL - L));
        q_store(q_imgZ(x, 0), qt, Zc[0]);
        q_store(q_imgZ(x, 1), qt, Zc[1]);
        if constexpr (L == 0) {
            if (half == 0) put_input_state(a, x, xt[(x.tenZ - x.lds0) / WAVE_B]);      // the head tile's inputs (x addresses the head tile)
        }
        if constexpr (RECOMP_W && L == 1) {                     // S_1 again from the inputs, this wave's blocks of both tiles
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                u32x4 S1[NS][1][1][NP];
                q_first<0>(a, x, xt[t], qt, S1);
                q_store(q_imgS(x, t, 1), qt, S1);
            }
        }
        lds_barrier();                                          // B(L)
        fused_stamp(a, x.tracer, 4 + 3 * (NL - L));
        if constexpr (L >= 1) {
            f32x4 acc[2][QB][NS];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int j = 0; j < QB; ++j) acc_zero(acc[t][j]);
            q_gemm_bwd<L>(x, qt, q_imgZ(x, 0), q_imgZ(x, 1), Ar, acc);       // operand: the Z_L images all four waves have just written
            u32x4 Zn[2][NS][1][1][NP];
#pragma unroll
            for (int t = 0; t < 2; ++t) q_bwd_epilogue<0>(acc[t], q_imgS(x, t, L), qt, Zn[t], x.c, x.q);
            fused_stamp(a, x.tracer, 5 + 3 * (NL - L));
            quad_down<L - 1>(a, x, xt, qt, half, Zn, Ar);
        }
    }
    // reverse of both tiles; on entry the first barrier of the top layer has NOT been passed, S_NL (hi + lo) sits in S slot 0, ZL = the HEAD tile's Z_NL
    static __device__ __forceinline__ void quad_reverse(const FusedArgs& a, const Ctx& x, const float (&xt)[2][4], int qt, int half, const u32x4 (&ZL)[NS][1][1][NP]) {
        u32x4 Ar[RINGQ][1][RP];
        u32x4 At[QB][1][RP];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < QB; ++j) load_afrags<1, RP>(x, FI::bwd_last(NL, 2 * qt + j), At[j]);
#pragma unroll
        for (int t = 0; t < RINGQ; ++t) q_bwd_request<NL - 1>(x, qt, t, Ar);
        u32x4 Zn[2][NS][1][1][NP];
        {
            fused_stamp(a, x.tracer, 2);
            lds_barrier();                                      // A(NL)
            fused_stamp(a, x.tracer, 3);
            if (half == 0) put_zimage<1>(x.imgZ(), ZL);         // (both waves of a tile hold the same Z_NL)
            lds_barrier();                                      // B(NL)
            fused_stamp(a, x.tracer, 4);
            f32x4 acc[2][QB][NS];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                u32x4 Bk[NS][1][1][NP];
                op_load(q_imgZ(x, t), 0, Bk);                   // Z_NL of tile t: the k-step-0 records of its Z area
#pragma unroll
                for (int j = 0; j < QB; ++j) {
                    acc_zero(acc[t][j]);
                    bwd_kstep<0, 1>(At[j], Bk, acc[t][j]);
                }
                q_bwd_epilogue<0>(acc[t], q_imgS(x, t, NL), qt, Zn[t], x.c, x.q);
            }
            fused_stamp(a, x.tracer, 5);
        }
        quad_down<NL - 1>(a, x, xt, qt, half, Zn, Ar);
    }

    // ---------------------------------------------------------------------------------------------
    // forward + output layer + residual head (net_f_sig INF:221-265) of the tile addressed by x:
    // parks S_1..S_{NL-1} (scratch image or LDS slots), returns S_NL (fragments) and the head's adjoint Z_NL, adds the loss sums
    static __device__ __forceinline__ void forward_tile(const FusedArgs& a, const Ctx& x, const float (&xin)[4], bool valid, long pidx, int set, float (&lsum)[LT],
                                                        u32x4 (&B)[NS][1][KS][NP], u32x4 (&ZL)[NS][1][1][NP]) {
        u32x4 A[WB][KS][FP];
        f32x4 acca[NS], accb[NS], bb[WB];
        fwd_first(a, x, xin, B, A, acca, bb);
        // two layers per loop trip, ping-ponging between B and B2: a one-buffer loop has to copy the fragment registers
        // back at the end of every layer
        u32x4 B2[NS][1][KS][NP];
        int l = 1;
        for (; l + 1 < NL; l += 2) {
            fwd_layer(x, l, B, B2, A, acca, accb, bb);
            fwd_layer(x, l + 1, B2, B, A, acca, accb, bb);
        }
        if (l < NL) {
            fwd_layer(x, l, B, B2, A, acca, accb, bb);
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int kk = 0; kk < KS; ++kk)
#pragma unroll
                    for (int pp = 0; pp < NP; ++pp) B[s][0][kk][pp] = B2[s][0][kk][pp];
        }
        fwd_head(a, x, valid, pidx, set, lsum, acca, ZL);
    }

    // output layer's products (acca) -> outputs, residual head, loss sums, adjoint seeds Z_NL
    static __device__ __forceinline__ void fwd_head(const FusedArgs& a, const Ctx& x, bool valid, long pidx, int set, float (&lsum)[LT], const f32x4 (&acca)[NS],
                                                    u32x4 (&ZL)[NS][1][1][NP]) {
        const int c = x.c, q = x.q;
        fused_stamp(a, x.tracer, 1);
        // acca = WS * (Y of the 16 padded outputs, bias included): lane holds outputs 4q+r of its point
        float Y[NS][NOG];
        if constexpr (NOG == 16) {
            // all 16 outputs of the point, from the four lanes (c, 0..3)
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float own = acca[s][r] * INV_WS;
                    const float o1 = __shfl_xor(own, 16), o2 = __shfl_xor(own, 32), o3 = __shfl_xor(own, 48);
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) Y[s][4 * qq + r] = q == qq ? own : ((q ^ 1) == qq ? o1 : ((q ^ 2) == qq ? o2 : o3));
                }
        } else {
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float own = acca[s][r] * INV_WS;
                    const float oth = __shfl_xor(own, 16);
                    Y[s][r] = (q & 1) ? oth : own;
                    Y[s][4 + r] = (q & 1) ? own : oth;
                }
        }
        const float vm = valid ? 1.0f : 0.0f;
        float adj[NS][NOG];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int o = 0; o < NOG; ++o) adj[s][o] = 0.0f;
        if constexpr (HEAD == HEAD_NC3D) {
            // 3-D Navier-Cauchy residuals (oracle/nc3d_oracle.py: the 3-D statement of INF:221-265); outputs (u,v,w, ut,vt,wt, s11,s22,s33,
            // s12,s13,s23); streams (value, d/dx, d/dy, d/dz, d/dt); c1 = lambda + 2G, c2 = lambda
            const float(&V)[NOG] = Y[0];
            const float(&X)[NOG] = Y[1];
            const float(&Yy)[NOG] = Y[2];
            const float(&Zz)[NOG] = Y[3];
            const float(&T)[NOG] = Y[4];
            const float e11 = X[0], e22 = Yy[1], e33 = Zz[2];
            const float e12 = Yy[0] + X[1], e13 = Zz[0] + X[2], e23 = Zz[1] + Yy[2];
            float f[12];
            f[0] = X[6] + Yy[9] + Zz[10] - a.rho * T[3];
            f[1] = X[9] + Yy[7] + Zz[11] - a.rho * T[4];
            f[2] = X[10] + Yy[11] + Zz[8] - a.rho * T[5];
            f[3] = T[0] - V[3];
            f[4] = T[1] - V[4];
            f[5] = T[2] - V[5];
            f[6] = V[6] - (a.c1 * e11 + a.c2 * (e22 + e33));
            f[7] = V[7] - (a.c1 * e22 + a.c2 * (e11 + e33));
            f[8] = V[8] - (a.c1 * e33 + a.c2 * (e11 + e22));
            f[9] = V[9] - a.G * e12;
            f[10] = V[10] - a.G * e13;
            f[11] = V[11] - a.G * e23;
            float g[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                if (q == 0) lsum[i] += vm * f[i] * f[i];
                g[i] = 2.0f * in_loop(a.tw[i]) * f[i] * vm;
            }
            adj[0][3] = -g[3];
            adj[0][4] = -g[4];
            adj[0][5] = -g[5];
#pragma unroll
            for (int i = 6; i < 12; ++i) adj[0][i] = g[i];
            adj[1][0] = -(a.c1 * g[6] + a.c2 * (g[7] + g[8]));
            adj[1][1] = -a.G * g[9];
            adj[1][2] = -a.G * g[10];
            adj[1][6] = g[0];
            adj[1][9] = g[1];
            adj[1][10] = g[2];
            adj[2][0] = -a.G * g[9];
            adj[2][1] = -(a.c1 * g[7] + a.c2 * (g[6] + g[8]));
            adj[2][2] = -a.G * g[11];
            adj[2][9] = g[0];
            adj[2][7] = g[1];
            adj[2][11] = g[2];
            adj[3][0] = -a.G * g[10];
            adj[3][1] = -a.G * g[11];
            adj[3][2] = -(a.c1 * g[8] + a.c2 * (g[6] + g[7]));
            adj[3][10] = g[0];
            adj[3][11] = g[1];
            adj[3][8] = g[2];
            adj[4][0] = g[3];
            adj[4][1] = g[4];
            adj[4][2] = g[5];
            adj[4][3] = -a.rho * g[0];
            adj[4][4] = -a.rho * g[1];
            adj[4][5] = -a.rho * g[2];
        } else if constexpr (HEAD == HEAD_PLATE) {
            // composite F = P + D*N (PLATE:383-387) with product-rule derivatives, then net_f_sig PLATE:404-439
            // outputs (u,v,s11,s22,s12); streams (value, x, y, t, tt); aux = [D|P][stream][field][n]
            float D[5][5], F[5][5];
#pragma unroll
            for (int st = 0; st < 5; ++st)
#pragma unroll
                for (int o = 0; o < 5; ++o) {
                    D[st][o] = a.aux[((long)(0 * 5 + st) * 5 + o) * a.n + pidx];
                    F[st][o] = a.aux[((long)(1 * 5 + st) * 5 + o) * a.n + pidx];      // start from P
                }
#pragma unroll
            for (int o = 0; o < 5; ++o) {
                const float n0 = Y[0][o];
                F[0][o] += D[0][o] * n0;
#pragma unroll
                for (int k = 1; k <= 3; ++k) F[k][o] += D[k][o] * n0 + D[0][o] * Y[k][o];
                F[4][o] += D[4][o] * n0 + 2.0f * D[3][o] * Y[3][o] + D[0][o] * Y[4][o];
            }
            const float e11 = F[1][0], e22 = F[2][1], e12 = F[2][0] + F[1][1];
            float f[5];
            f[0] = F[1][2] + F[2][4] - a.rho * F[4][0];                       // f_u   PLATE:436
            f[1] = F[2][3] + F[1][4] - a.rho * F[4][1];                       // f_v   PLATE:437
            f[2] = F[0][2] - (a.c1 * e11 + a.c2 * e22);                       // f_s11 PLATE:421
            f[3] = F[0][3] - (a.c2 * e11 + a.c1 * e22);                       // f_s22 PLATE:423
            f[4] = F[0][4] - a.G * e12;                                       // f_s12 PLATE:422
            float g[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                if (q == 0) lsum[i] += vm * f[i] * f[i];
                g[i] = 2.0f * in_loop(a.tw[i]) * f[i] * vm;
            }
            float Fb[5][5];
#pragma unroll
            for (int st = 0; st < 5; ++st)
#pragma unroll
                for (int o = 0; o < 5; ++o) Fb[st][o] = 0.0f;
            Fb[0][2] = g[2];
            Fb[0][3] = g[3];
            Fb[0][4] = g[4];
            Fb[1][0] = -a.c1 * g[2] - a.c2 * g[3];
            Fb[2][1] = -a.c2 * g[2] - a.c1 * g[3];
            Fb[2][0] = -a.G * g[4];
            Fb[1][1] = -a.G * g[4];
            Fb[1][2] = g[0];
            Fb[2][4] = g[0];
            Fb[4][0] = -a.rho * g[0];
            Fb[2][3] = g[1];
            Fb[1][4] = g[1];
            Fb[4][1] = -a.rho * g[1];
#pragma unroll
            for (int o = 0; o < 5; ++o) {
                adj[0][o] = Fb[0][o] * D[0][o] + Fb[1][o] * D[1][o] + Fb[2][o] * D[2][o] + Fb[3][o] * D[3][o] + Fb[4][o] * D[4][o];
                adj[1][o] = Fb[1][o] * D[0][o];
                adj[2][o] = Fb[2][o] * D[0][o];
                adj[3][o] = Fb[3][o] * D[0][o] + 2.0f * Fb[4][o] * D[3][o];
                adj[4][o] = Fb[4][o] * D[0][o];
            }
        } else if constexpr (NS == 4) {
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
                g[i] = 2.0f * in_loop(a.tw[i]) * f[i] * vm;
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
        } else if (a.set_head[set] == 1) {
            // net_t PLATE:452-461 on the composite values F = P + D N (value stream only); the set's aux rows: D0[0..4], P0[5..9], nx[10], ny[11]
            const float* aux = a.set_aux[set];
            const long nn = a.set_n[set];
            float Fv[5], D0[5];
#pragma unroll
            for (int o = 0; o < 5; ++o) {
                D0[o] = aux[(long)o * nn + pidx];
                Fv[o] = aux[(long)(5 + o) * nn + pidx] + D0[o] * Y[0][o];
            }
            const float nx = aux[10L * nn + pidx], ny = aux[11L * nn + pidx];
            const float tx = Fv[2] * nx + Fv[4] * ny, ty = Fv[4] * nx + Fv[3] * ny;
            if (q == 0) {
                lsum[0] += vm * tx * tx;
                lsum[1] += vm * ty * ty;
            }
            const float gx = 2.0f * in_loop(a.set_tw[set][0]) * tx * vm, gy = 2.0f * in_loop(a.set_tw[set][1]) * ty * vm;
            adj[0][2] = gx * nx * D0[2];
            adj[0][3] = gy * ny * D0[3];
            adj[0][4] = (gx * ny + gy * nx) * D0[4];
        } else {
#pragma unroll
            for (int o = 0; o < NOG; ++o) {
                float d = 0.0f;
                const float* tg = a.set_targets[set];
                if (o < a.net.nout) d = Y[0][o] - (tg ? tg[(long)o * a.set_n[set] + pidx] : 0.0f);
                if (q == 0) lsum[o] += vm * d * d;
                adj[0][o] = 2.0f * in_loop(a.set_tw[set][o]) * d * vm;
            }
        }
        float vals[NS][1][4];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (NOG == 16) vals[s][0][r] = q == 0 ? adj[s][r] : (q == 1 ? adj[s][4 + r] : (q == 2 ? adj[s][8 + r] : adj[s][12 + r]));
                else vals[s][0][r] = q < 2 ? ((q & 1) ? adj[s][4 + r] : adj[s][r]) : 0.0f;
            }
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int pp = 0; pp < NP; ++pp) ZL[s][0][0][pp] = u32x4{0u, 0u, 0u, 0u};
        CH::template emit<1, 0>(ZL, vals, nullptr, 16, c, q);
    }

    // reverse of one tile through all weight layers, in step with the weight-gradient waves; B holds S_NL
    static __device__ __forceinline__ void reverse_tile(const FusedArgs& a, const Ctx& x, const float (&xin)[4], const u32x4 (&B)[NS][1][KS][NP],
                                                        const u32x4 (&ZL)[NS][1][1][NP]) {
        // ---- top weight layer NL: hand Z_NL (16 outputs) and S_NL over, then reverse into the hidden chain
        fused_stamp(a, x.tracer, 2);
        u32x4 Aa[1][RP], Ab[1][RP];
        load_afrags<1, RP>(x, FI::bwd_last(NL, 0), Aa);
        load_afrags<1, RP>(x, FI::bwd_last(NL, 1), Ab);
        __syncthreads();              // the one full drain of the reverse: every park store of this forward has landed before an LDS-DMA reads it
        fused_stamp(a, x.tracer, 3);
        if constexpr (ZDB) put_zimage_hi<1>(x.imgZ() + zbuf(NL), ZL);
        else put_zimage<1>(x.imgZ(), ZL);
        if constexpr (TOP_IN_Z) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) *reinterpret_cast<u32x4*>(x.imgZ() + TOPZ_OFF + s * TOPZ_STRIDE + kk * 1024) = B[s][0][kk][0];
        } else {
            put_image<KS>(x.imgS(NL), B);
        }
        hand_barrier();
        fused_stamp(a, x.tracer, 4);
        u32x4 Zn[NS][1][KS][NP];
        {
            f32x4 acca[NS], accb[NS];
            acc_zero(acca);
            bwd_ksteps<0, 1, 1>(Aa, ZL, acca);
            u32x2 sla[NS], slb[NS];
            bwd_step<0, 1, true>(x, FI::bwd_last(NL, 0), NL, nullptr, B, ZL, Zn, Aa, Ab, acca, accb, sla, slb);
            pin<KS>(Zn);
        }
        fused_stamp(a, x.tracer, 5);
        Down<NL - 1>::run(a, x, xin, Zn);
    }

    static __device__ __forceinline__ void load_inputs(const FusedArgs& a, const float* px, const float* py, const float* pt, const float* pz, long n, long tile, int c,
                                                       float (&xin)[4], bool& valid, long& pidx) {
        const long p = tile * 16 + c;
        valid = p < n;
        pidx = valid ? p : n - 1;
        xin[0] = px[pidx] * a.sx[0] + a.ox[0];
        xin[1] = py[pidx] * a.sx[1] + a.ox[1];
        if constexpr (DIN == 4) {
            xin[2] = pz[pidx] * a.sx[2] + a.ox[2];
            xin[3] = pt[pidx] * a.sx[3] + a.ox[3];
        } else {
            xin[2] = pt[pidx] * a.sx[2] + a.ox[2];
            xin[3] = 0.0f;
        }
    }

    // this step's loss terms onto the tile's in-memory running sums (LSUM_MEM): all loads in flight together, then the adds, then the stores
    static __device__ __forceinline__ void lsum_add(const Ctx& x, const float (&ls)[LT]) {
        float run[LT];
#pragma unroll
        for (int i = 0; i < LT; ++i) run[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x.scr, x.lane16 >> 2, LSUM_OFF + i * 256, 0));
#pragma unroll
        for (int i = 0; i < LT; ++i) run[i] += ls[i];
#pragma unroll
        for (int i = 0; i < LT; ++i) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, run[i]), x.scr, x.lane16 >> 2, LSUM_OFF + i * 256, 0);
    }

    static __device__ __forceinline__ void chain_role(const FusedArgs& a, char* lds, int wave4, int lane, int c, int q) {
        // LDSOP: two waves per tile (tile = wave & 1, half = wave >> 1); otherwise one wave per tile
        const int wave = LDSOP ? (wave4 & 1) : wave4, half = LDSOP ? (wave4 >> 1) : 0;
        const long gwave = (long)fused_bid(a) * TILES + wave;
        Ctx x;
        x.init(a, lds, wave, lane, c, q);
        x.set_tile(a, gwave);
        x.tracer = fused_bid(a) == 0 && wave4 == 0 && lane == 0;
        constexpr int NSETS = NS == 1 ? FUSED_MAX_SETS : 1;
        float lsum[NSETS][LT];
#pragma unroll
        for (int k = 0; k < NSETS; ++k)
#pragma unroll
            for (int i = 0; i < LT; ++i) lsum[k][i] = 0.0f;

        if constexpr (LSUM_MEM) {
            if (half == 0) {
#pragma unroll
                for (int i = 0; i < LT; ++i) __builtin_amdgcn_raw_buffer_store_b32(0u, x.scr, x.lane16 >> 2, LSUM_OFF + i * 256, 0);
            }
        }
        // launch-long stamps of workgroup 0: shader cycles (slots 124 / 125; the one-stream part 122 / 123) and the device's constant-rate wall clock
        // (120 / 121; 118 / 119) around all of its steps: duration = wall ticks / rate, clock = cycles / duration (bench.py: shader_clock_ghz,
        // launch_ms_device_clock -- the kernel is power-limited, and a bench line without its clock cannot tell a code change from a box)
        const bool launch_tracer = x.tracer;
        fused_stamp(a, launch_tracer, NS == 1 ? 122 : 124);
        fused_stamp_wall(a, launch_tracer, NS == 1 ? 118 : 120);
        for (long step = fused_bid(a); step < a.nsteps; step = XCD_TAIL ? fused_next_step(a, step) : step + a.grid) {
            float xin[4];
            bool valid;
            long pidx;
            int set = 0;
            if constexpr (NS == 1) {
#pragma unroll
                for (int k = 1; k < FUSED_MAX_SETS; ++k)
                    if (k < a.nsets && step >= a.set_step0[k]) set = k;
                load_inputs(a, a.set_x[set], a.set_y[set], a.set_t[set], a.set_z[set], a.set_n[set], (step - a.set_step0[set]) * TILES + wave, c, xin, valid, pidx);
            } else {
                load_inputs(a, a.x, a.y, a.t, a.z, a.n, step * TILES + wave, c, xin, valid, pidx);
            }
            x.tracer = fused_bid(a) == 0 && wave4 == 0 && lane == 0 && step == 2 * (long)a.grid;      // a steady-state step
            if constexpr (LDSOP) {                 // lane addresses derived from these are then formed where they are used, not hoisted and spilled
                x.imgoff = in_loop(x.imgoff);
                x.lane16 = in_loop(x.lane16);
            }
            fused_stamp(a, x.tracer, 0);
            if constexpr (LDSOP) {
                u32x4 ZL[NS][1][1][NP];
                f32x4 acca[NS];
                // step barrier: this layout's forward writes the LDS images (Z area, S slot 0) that the previous step's last weight
                // gradient still reads (the narrow layouts' forward does not touch them).  Without it: a race that the x86 emulator cannot
                // show and that happened not to bite with two tiles.
                __syncthreads();
                float xt[2][4];                    // QUAD: the inputs of both tiles (this wave computes its quarter of the blocks for both)
                if constexpr (QUAD) {
                    float xo[4];
                    bool vo;
                    long po;
                    if constexpr (NS == 1) load_inputs(a, a.set_x[set], a.set_y[set], a.set_t[set], a.set_z[set], a.set_n[set], (step - a.set_step0[set]) * TILES + (wave ^ 1), c, xo, vo, po);
                    else load_inputs(a, a.x, a.y, a.t, a.z, a.n, step * TILES + (wave ^ 1), c, xo, vo, po);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        xt[0][k] = wave ? xo[k] : xin[k];
                        xt[1][k] = wave ? xin[k] : xo[k];
                    }
                    quad_forward(a, x, xt, wave4, acca);
                } else if constexpr (FWD8) {
                    f8_chain_forward(a, x, xin, half, acca);
                } else {
                    wide_forward(a, x, xin, half, acca);
                }
                float ls[LT];
#pragma unroll
                for (int i = 0; i < LT; ++i) ls[i] = 0.0f;
                fwd_head(a, x, valid, pidx, set, ls, acca, ZL);          // both halves: the same Z_NL; the loss sums count once
                if constexpr (LSUM_MEM) {
                    if (half == 0) lsum_add(x, ls);
                } else {
#pragma unroll
                    for (int k = 0; k < NSETS; ++k)
#pragma unroll
                        for (int i = 0; i < LT; ++i) lsum[k][i] += (k == set && half == 0) ? ls[i] : 0.0f;
                }
                if constexpr (QUAD) quad_reverse(a, x, xt, wave4, half, ZL);
                else wide_reverse(a, x, xin, half, ZL);
            } else {
                u32x4 B[NS][1][KS][NP], ZL[NS][1][1][NP];
                float ls[LT];
#pragma unroll
                for (int i = 0; i < LT; ++i) ls[i] = 0.0f;
                forward_tile(a, x, xin, valid, pidx, set, ls, B, ZL);
                if constexpr (LSUM_MEM) {
                    lsum_add(x, ls);
                } else {
#pragma unroll
                    for (int k = 0; k < NSETS; ++k)
#pragma unroll
                        for (int i = 0; i < LT; ++i) lsum[k][i] += (k == set) ? ls[i] : 0.0f;
                }
                reverse_tile(a, x, xin, B, ZL);
            }
        }
        fused_stamp(a, launch_tracer, NS == 1 ? 123 : 125);
        fused_stamp_wall(a, launch_tracer, NS == 1 ? 119 : 121);
        if constexpr (LSUM_MEM) {
            if (half == 0) {
#pragma unroll
                for (int i = 0; i < LT; ++i) lsum[0][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x.scr, x.lane16 >> 2, LSUM_OFF + i * 256, 0));
            }
        }
#pragma unroll
        for (int k = 0; k < NSETS; ++k)
#pragma unroll
            for (int i = 0; i < LT; ++i) {
                float v = lsum[k][i];
                v += __shfl_xor(v, 1);
                v += __shfl_xor(v, 2);
                v += __shfl_xor(v, 4);
                v += __shfl_xor(v, 8);
                if (lane == 0 && half == 0) a.loss_part[(gwave * NSETS + k) * LT + i] = v;
            }
    }

    static __device__ __forceinline__ void run(const FusedArgs& a, char* lds /* LDS_B bytes, 16-byte aligned: declared by the kernel */) {
        const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
        const int wave8 = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // provably wave-uniform
        if constexpr (CONST_LDS) {
            float* cst = reinterpret_cast<float*>(lds + CONST_OFF);
            for (int i = threadIdx.x; i < CONST_F; i += 512) {
                float v;
                if (i < (NL - 1) * WIDTH) v = a.pw.bias_mid[i] * WS;
                else if (i < CONST_BIAS_F) v = a.pw.bias_last[i - (NL - 1) * WIDTH] * WS;
                else v = a.pw.w0p[i - CONST_BIAS_F];
                cst[i] = v;
            }
            __syncthreads();
        }
        if (wave8 >= 4) {
            wgrad_role(a, lds, wave8 - 4, lane, c, q);
        } else {
            __builtin_amdgcn_s_setprio(2);          // the chain wave is the critical path of its SIMD: issue it first
            chain_role(a, lds, wave8, lane, c, q);
        }
    }
};

template <class Op, int SPLIT, int WIDTH, int NL, int NS, bool FASTSTATE, int DIN = 3>
__global__ __launch_bounds__(512) void fused_wave_kernel(const FusedArgs a) {
    typedef Fused<Op, SPLIT, WIDTH, NL, NS, FASTSTATE, DIN> F;
    __shared__ __attribute__((aligned(16))) char lds[F::LDS_B];
    F::run(a, lds);
}

// One training step's point sets in ONE persistent launch (round 5): workgroups [0, a4.grid) run the collocation set (four streams, the
// residual head), workgroups [a4.grid, a4.grid + a1.grid) the value-only side sets (one stream: loss_IC / loss_SRC / loss_NB / loss_FIX,
// INF:111-118).  Both parts are the roles of fused_wave_kernel, unchanged; they share nothing but the launch.  What it buys: the side sets'
// workgroups are dispatched as compute units run out of collocation steps -- a collocation set of N points is ceil(N / 64) steps over 256
// workgroups, so in its last step most compute units idle (250,000 points, one of 8 GPUs' share of BASELINE's 2 M: 67 workgroups have a
// 16th step, 189 do not) -- instead of waiting, as a second launch, for the whole first one; and a step is one launch less.
// (NSC = 5: the plate's step -- its five-stream collocation set and the hole-traction set, PLATE:187-217 -- the same way.)
template <class Op, int SPLIT, int WIDTH, int NL, int NSC, bool FASTSTATE>
__global__ __launch_bounds__(512) void fused_step_kernel(const FusedArgs a4, const FusedArgs a1) {
    typedef Fused<Op, SPLIT, WIDTH, NL, NSC, FASTSTATE, 3> F4;
    typedef Fused<Op, SPLIT, WIDTH, NL, 1, false, 3> F1;
    __shared__ __attribute__((aligned(16))) char lds[F4::LDS_B > F1::LDS_B ? F4::LDS_B : F1::LDS_B];
    if ((int)blockIdx.x < a4.grid) F4::run(a4, lds);
    else F1::run(a1, lds);
}

}  // namespace pinn
