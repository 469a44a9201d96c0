// Device code of the PINN-elastodynamics hot path for gfx950 (CDNA4, wave64, MFMA 16x16x32).
//
// What the kernels compute (reference: ElasticWaveInfinite/ElasticWave.py, "INF"):
//   neural_net INF:188-199, net_uv INF:201-211, net_e INF:213-219, net_f_sig INF:221-265,
//   the mean-square terms INF:104-118 and d(loss)/d(W,b) (what AdamOptimizer.minimize
//   differentiates, INF:131-133).
// How: the twelve tf.gradients reverse passes of the reference are replaced by three input
// tangents (d/dx, d/dy, d/dt) carried forward next to the value ("4 streams"), followed by one
// reverse pass over that computation.  See DESIGN.md for the derivation and the layouts.
//
// Lane maps used throughout (MFMA 16x16x32, lane l: c = l & 15, q = l >> 4):
//   A operand: row m = c, k-slot 8q+j (j = 0..7);  B operand: col n = c, k-slot 8q+j;
//   C/D: row 4q+r (r = 0..3), col c.
// "Swapped chain": every layer computes D'[feature, point] = W^T[feature, k] . H^T[k, point], so a
// lane ends up holding 4 consecutive output features (16 mb + 4q + r) of ITS point c, which is
// exactly what it must supply as B-operand k-slots of the next layer once the weight fragments
// are stored with the matching k permutation  kmap(kk,q,j) = 32kk + 16(j>>2) + 4q + (j&3).
// No LDS round trip or cross-lane traffic is needed between layers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pinn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// Scale carried by the weights in the fused kernel's fragment format (see repack_kernel / pinn_fused.hpp).  2^5 keeps the low
// part of the scaled weight in the 16-bit normal range for |w| > 4e-3 (measured on the reference's trained nets: same field /
// residual / gradient error as 2^11) and lets |w| reach 2047 before the high part overflows fp16.
constexpr float FUSED_WEIGHT_SCALE = 32.0f;
constexpr float FUSED_OPERAND_MAX = 65504.0f;      // largest finite 16-bit operand of the split formats (fp16): |w| <= 2047 in the fused weight format
constexpr int MAX_WLAYERS = 16;   // weight matrices per net (hidden layers + 1)
constexpr int NOUT_PAD = 16;      // network outputs are padded to one 16-row MFMA block

// ------------------------------------------------------------------------------------------
// 16-bit operand types of the matrix pipe
// ------------------------------------------------------------------------------------------
struct OpF16 {
    typedef _Float16 T;
    typedef f16x8 V8;
    typedef f16x2 V2;
    // low part of the hi/lo split is stored scaled by 2^11 so that it keeps fp16's normal range
    static constexpr float LO_SCALE = 2048.0f;
    static constexpr bool TOP_BYTE_IS_FLOAT = true;      // the top byte of a value is itself a float (e5m2): the fused kernel's one-byte low parts (LO8)
    static constexpr float OPERAND_MAX = 65504.0f;       // largest finite fp16: |w| <= 2047 in the fused weight format (FUSED_OPERAND_MAX)
    static __device__ __forceinline__ f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(V8, a), __builtin_bit_cast(V8, b), c, 0, 0, 0);
    }
};
struct OpBF16 {
    typedef __bf16 T;
    typedef bf16x8 V8;
    typedef bf16x2 V2;
    static constexpr float LO_SCALE = 256.0f;
    static constexpr bool TOP_BYTE_IS_FLOAT = false;     // (sign + seven of eight exponent bits)
    static constexpr float OPERAND_MAX = 3.3895314e38f;  // largest finite bf16: the fused weight format of bf16x3 has fp32's range
    static __device__ __forceinline__ f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(V8, a), __builtin_bit_cast(V8, b), c, 0, 0, 0);
    }
};

template <class Op>
__device__ __forceinline__ uint32_t pack2(float a, float b) {
    typename Op::V2 v;
    v[0] = (typename Op::T)a;
    v[1] = (typename Op::T)b;
    return __builtin_bit_cast(uint32_t, v);
}
// (a, b) -> packed hi pair and packed scaled-lo pair:  lo = T((x - float(T(x))) * LO_SCALE), written as one fused
// multiply-add on the converted hi so that fp16 compiles to v_fma_mixlo/mixhi_f16 (x*LO_SCALE and hi*LO_SCALE are exact).
template <class Op>
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
    typename Op::V2 h, l;
    h[0] = (typename Op::T)a;
    h[1] = (typename Op::T)b;
    l[0] = (typename Op::T)__builtin_fmaf((float)h[0], -Op::LO_SCALE, a * Op::LO_SCALE);
    l[1] = (typename Op::T)__builtin_fmaf((float)h[1], -Op::LO_SCALE, b * Op::LO_SCALE);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, l);
}
#if defined(__AMDGCN__) && !defined(PINN_GENERIC_SPLIT)
// fp16 on the GPU: the same arithmetic pinned to  v_cvt_pk_f16_f32 ; v_fma_mixlo_f16 ; v_fma_mixhi_f16  (the mixed-precision
// FMA reads the fp16 hi part directly and writes the rounded fp16 low part into its half of the destination).  Left to
// itself the compiler SLP-vectorises the two FMAs into v_pk_fma_f32 and pays two v_cvt_f32_f16 plus a second v_cvt_pk for
// them; the split is 18 % of the fused kernel's time (measured by deleting it, DESIGN_HISTORY.md section 6).
template <>
__device__ __forceinline__ void split2<OpF16>(float a, float b, uint32_t& hi, uint32_t& lo) {
    const uint32_t h = pack2<OpF16>(a, b);
    const float nls = -OpF16::LO_SCALE;
    const float ta = a * OpF16::LO_SCALE, tb = b * OpF16::LO_SCALE;      // (a packed v_pk_mul_f32 here measured +4 %)
    uint32_t l;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, %2, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(l)
        : "v"(h), "s"(nls), "v"(ta), "v"(tb));
    hi = h;
    lo = l;
}
#endif
// (a, b) -> packed hi pair and packed UNSCALED low pair  lo = T(x - float(T(x))).  Used for the forward activations of the fused
// kernel, whose weights carry the scale instead (FUSED_WEIGHT_SCALE = 2^5, see repack_kernel's fused format): measured in tools/precision_study3.py, an
// unscaled (possibly subnormal) low part of the ACTIVATIONS costs nothing in accuracy, an unscaled low part of the weights does.
template <class Op>
__device__ __forceinline__ void split2u(float a, float b, uint32_t& hi, uint32_t& lo) {
    typename Op::V2 h, l;
    h[0] = (typename Op::T)a;
    h[1] = (typename Op::T)b;
    l[0] = (typename Op::T)(a - (float)h[0]);
    l[1] = (typename Op::T)(b - (float)h[1]);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, l);
}
#if defined(__AMDGCN__) && !defined(PINN_GENERIC_SPLIT)
// fp16 on the GPU:  v_cvt_pk_f16_f32 ; v_fma_mixlo_f16 ; v_fma_mixhi_f16  -- 1.5 vector instructions per value
template <>
__device__ __forceinline__ void split2u<OpF16>(float a, float b, uint32_t& hi, uint32_t& lo) {
    const uint32_t h = pack2<OpF16>(a, b);
    uint32_t l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(l)
        : "v"(h), "v"(a), "v"(b));
    hi = h;
    lo = l;
}
#endif
// Two pairs at once.  v_fma_mixhi_f16 writes the other half of the register its v_fma_mixlo_f16 has just written: issued right behind it, it
// waits for that result -- a dependent vector instruction issues after 8.3 cycles, an independent one after 5 (tools/probes/
// valu_dep_probe.hip) -- and inside one asm statement the compiler cannot put anything in between.  With two pairs the statement
// alternates them: lo, lo, hi, hi.
template <class Op>
__device__ __forceinline__ void split4(float a, float b, float c, float d, uint32_t& h0, uint32_t& h1, uint32_t& l0, uint32_t& l1) {
    split2<Op>(a, b, h0, l0);
    split2<Op>(c, d, h1, l1);
}
template <class Op>
__device__ __forceinline__ void split4u(float a, float b, float c, float d, uint32_t& h0, uint32_t& h1, uint32_t& l0, uint32_t& l1) {
    split2u<Op>(a, b, h0, l0);
    split2u<Op>(c, d, h1, l1);
}
#if defined(__AMDGCN__) && !defined(PINN_GENERIC_SPLIT)
template <>
__device__ __forceinline__ void split4<OpF16>(float a, float b, float c, float d, uint32_t& h0, uint32_t& h1, uint32_t& l0, uint32_t& l1) {
    const uint32_t ha = pack2<OpF16>(a, b), hb = pack2<OpF16>(c, d);
    const float nls = -OpF16::LO_SCALE;
    const float ta = a * OpF16::LO_SCALE, tb = b * OpF16::LO_SCALE, tc = c * OpF16::LO_SCALE, td = d * OpF16::LO_SCALE;
    uint32_t la, lb;
    asm("v_fma_mixlo_f16 %0, %2, %4, %5 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mixlo_f16 %1, %3, %4, %7 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %2, %4, %6 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %1, %3, %4, %8 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(la), "=&v"(lb)
        : "v"(ha), "v"(hb), "s"(nls), "v"(ta), "v"(tb), "v"(tc), "v"(td));
    h0 = ha; h1 = hb; l0 = la; l1 = lb;
}
template <>
__device__ __forceinline__ void split4u<OpF16>(float a, float b, float c, float d, uint32_t& h0, uint32_t& h1, uint32_t& l0, uint32_t& l1) {
    const uint32_t ha = pack2<OpF16>(a, b), hb = pack2<OpF16>(c, d);
    uint32_t la, lb;
    asm("v_fma_mixlo_f16 %0, %2, -1.0, %4 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mixlo_f16 %1, %3, -1.0, %6 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %2, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %1, %3, -1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(la), "=&v"(lb)
        : "v"(ha), "v"(hb), "v"(a), "v"(b), "v"(c), "v"(d));
    h0 = ha; h1 = hb; l0 = la; l1 = lb;
}
#endif
// fp16 operands read straight out of a packed dword by the mixed-precision FMA (HALF = 0 / 1 selects the low / high half):
//   fma<HALF>(w, b, c) = float(half(w)) * b + c        one_minus_sq<HALF>(w) = 1 - float(half(w))^2        sum2<HALF>(hi, lo)
// Only the gfx950 build of the fp16 operand type has it; everything else converts first.
template <class Op>
struct MixF16 { static constexpr bool value = false; };
#if defined(__AMDGCN__) && !defined(PINN_GENERIC_SPLIT)
template <>
struct MixF16<OpF16> {
    static constexpr bool value = true;
    template <int HALF>
    static __device__ __forceinline__ float fma(uint32_t w, float b, float c) {
        float d;
        if (HALF) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(w), "v"(b), "v"(c));
        else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(w), "v"(b), "v"(c));
        return d;
    }
    template <int HALF>
    static __device__ __forceinline__ float one_minus_sq(uint32_t w) {
        float d;
        if (HALF) asm("v_fma_mix_f32 %0, -%1, %1, 1.0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(w));
        else asm("v_fma_mix_f32 %0, -%1, %1, 1.0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(w));
        return d;
    }
    // float(half(whi)) + float(half(wlo)): a value from its high and (unscaled) low part in one instruction
    template <int HALF>
    static __device__ __forceinline__ float sum2(uint32_t whi, uint32_t wlo) {
        float d;
        if (HALF) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(whi), "v"(wlo));
        else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(whi), "v"(wlo));
        return d;
    }
};
#endif
template <class Op>
__device__ __forceinline__ float cvt16(uint16_t bits) {
    return (float)__builtin_bit_cast(typename Op::T, bits);
}
template <class Op>
__device__ __forceinline__ float round16(float a) {
    return (float)(typename Op::T)a;
}

// ------------------------------------------------------------------------------------------
// kernel arguments (plain structs passed by value)
// ------------------------------------------------------------------------------------------
struct NetDesc {
    int nl;                    // hidden layers; weight matrices = nl + 1
    int h;                     // real hidden width (<= WIDTH template parameter)
    int nout;                  // real number of outputs (7 for the wave scripts, INF:204-210; 12 for the 3-D extension)
    int din;                   // inputs: 3 = (x, y, t) as in the reference, 4 = (x, y, z, t) for the 3-D extension
    int w_off[MAX_WLAYERS];    // offset of W_l in the flat parameter vector (W row-major [in,out])
    int b_off[MAX_WLAYERS];    // offset of b_l
    int nparams;
};

struct PackedWeights {         // produced by repack_kernel, consumed by chain_kernel
    const float* w0p;          // din = 3: [WIDTH][4] (W0[0][f], W0[1][f], W0[2][f], b0[f]);  din = 4: [WIDTH][8] (W0[0..3][f], b0[f], 0, 0, 0)
    const float* bias_mid;     // [nl-1][WIDTH] biases of weight layers 1..nl-1
    const float* bias_last;    // [16]
    const u32x4* frags;        // fragment store, see frag_index()
};

// HEAD_WAVE   : 4 streams, residuals of net_f_sig (INF:221-265)
// HEAD_DATA   : 1 stream, sum_o w_o (Y_o - target_o)^2 (INF:111-118)
// HEAD_FIELDS : forward only, writes all NS streams of the outputs
// HEAD_PLATE  : 5 streams (value, x, y, t, tt), composite P + D*N with frozen D/P streams in `aux`, plane-stress residuals (PLATE:358-439)
// HEAD_TRACTION: 1 stream, hole traction of the composite (PLATE:452-461); aux = D0[5], P0[5], nx, ny per point
// HEAD_STREAMS: 5 streams, sum_{s,o} w[s][o] (Y[s][o] - target[s][o])^2 -- the D / P pre-training losses (PLATE:194-215)
// HEAD_NC3D   : 5 first-order streams (value, x, y, z, t), 4 inputs, 12 outputs: 3-D Navier-Cauchy residuals (build-side extension of
//               INF:221-265, stated in oracle/nc3d_oracle.py; BASELINE.json configs[4])
// HEAD_DATA3D : HEAD_DATA for the 4-input net (up to 16 outputs);  HEAD_FIELDS3D : HEAD_FIELDS for it (5 streams)
enum { HEAD_WAVE = 0, HEAD_DATA = 1, HEAD_FIELDS = 2, HEAD_PLATE = 3, HEAD_TRACTION = 4, HEAD_STREAMS = 5, HEAD_NC3D = 6, HEAD_DATA3D = 7,
       HEAD_FIELDS3D = 8 };
__host__ __device__ constexpr bool head_is_3d(int head) { return head == HEAD_NC3D || head == HEAD_DATA3D || head == HEAD_FIELDS3D; }
constexpr int LOSS_SLOTS_3D = 16;     // per-wave loss partials: 8 slots for the reference's heads, 16 for the 3-D ones

struct ChainArgs {
    NetDesc net;
    PackedWeights pw;
    const float* x;            // SoA point coordinates
    const float* y;
    const float* t;
    const float* z;            // third space coordinate of the 4-input heads (input order x, y, z, t), else unused
    long n;                    // total points of this call
    long tile0;                // first tile of this workspace chunk
    long ntiles;               // tiles in this chunk
    float sx[4], ox[4];        // input map x' = x*sx + ox  (INF:191 when normalising, identity otherwise); 4-input heads: index 2 = z, 3 = t
    float c1, c2, G, rho;      // Hooke coefficients (INF:238-241 / PLATE:416-418) and density
    float tw[16];              // per-residual (HEAD_WAVE / HEAD_NC3D) or per-output (HEAD_DATA) weights, max-normalised
    const float* targets;      // HEAD_DATA: [nout][n] SoA targets or nullptr (= 0)
    uint16_t* S;               // forward-state panels of this chunk
    uint16_t* Z;               // adjoint panels of this chunk
    long S_tile_stride;        // in 16-bit elements
    long Z_tile_stride;
    float* loss_part;          // [total waves][8 or LOSS_SLOTS_3D] per-wave partial sums of squares
    float* fields_out;         // HEAD_FIELDS: [NS*nout][n]  (Y, dY/dx, dY/dy, dY/dt [, d2Y/dt2])
    const float* aux;          // HEAD_PLATE: [2 nets (D,P)][5 streams][5 fields][n]; HEAD_TRACTION: [12][n]; HEAD_STREAMS: targets [5][nout][n] or null
    float w5[5][8];            // HEAD_STREAMS: per (stream, output) weights, max-normalised
};

struct WgradArgs {
    NetDesc net;
    const uint16_t* S;
    const uint16_t* Z;
    long S_tile_stride, Z_tile_stride;
    long ntiles;               // tiles in this chunk (even when NB == 1)
    float* partial;            // [gridDim.x][nparams]
    int first_pass;            // 1: overwrite partials, 0: add to them
};

struct RepackArgs {
    NetDesc net;
    const float* params;
    float* w0p;
    float* bias_mid;
    float* bias_last;
    u32x4* frags;
    u32x4* frags_fused;        // second copy in the fused kernel's format (see repack_kernel), or nullptr
    int* wflags;               // [gridDim.x]: 1 where this block met a weight the fused format cannot hold (|FUSED_WEIGHT_SCALE * w| beyond fp16), or nullptr
};

typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16 lds_v4i16;
typedef __attribute__((address_space(3))) void lds_void;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// k-slot -> feature permutation shared by every A/B fragment pair of the chain
__host__ __device__ __forceinline__ int kmap(int kk, int q, int j) { return 32 * kk + 16 * (j >> 2) + 4 * q + (j & 3); }

// Fragment store layout: [frag][part][lane] of u32x4 with
//   fwd_mid (l,mb,kk), l = 1..nl-1 : ((l-1)*WB + mb)*KS + kk
//   fwd_last(kk)                   : FM + kk                      (FM = (nl-1)*WB*KS)
//   bwd_mid (l,mb,kk)              : FM + KS + ((l-1)*WB + mb)*KS + kk
//   bwd_last(mb)                   : 2*FM + KS + mb
template <int WIDTH>
struct FragIndex {
    static constexpr int WB = WIDTH / 16, KS = WIDTH / 32;
    __host__ __device__ static int fm(int nl) { return (nl - 1) * WB * KS; }
    __host__ __device__ static int fwd_mid(int l, int mb, int kk) { return ((l - 1) * WB + mb) * KS + kk; }
    __host__ __device__ static int fwd_last(int nl, int kk) { return fm(nl) + kk; }
    __host__ __device__ static int bwd_mid(int nl, int l, int mb, int kk) { return fm(nl) + KS + ((l - 1) * WB + mb) * KS + kk; }
    __host__ __device__ static int bwd_last(int nl, int mb) { return 2 * fm(nl) + KS + mb; }
    __host__ __device__ static int total(int nl) { return 2 * fm(nl) + KS + WB; }
};

// Panel geometry inside one tile record (units: 16-bit elements).  TP = points per tile.
//   S panels: layer 0 has 16 rows (the inputs), layers 1..nl have WIDTH rows (hidden states h_l)
//   Z panels: layers 0..nl-1 have WIDTH rows (adjoints of the pre-activations), layer nl has 16 rows
// Both keep the high and the scaled low part: this path is the accurate one (gradient 2e-7 from the float64 oracle at fresh weights;
// the fused kernel parks its state as high parts only and trades a 1/sqrt(points) rounding noise for LDS room, DESIGN_HISTORY.md section 6).
template <int WIDTH, int NB, int NS, int NP>
struct PanelGeom {
    static constexpr int TP = 16 * NB;
    __host__ __device__ static long s_off(int l) { return l == 0 ? 0 : (long)NS * NP * 16 * TP + (long)(l - 1) * NS * NP * WIDTH * TP; }
    __host__ __device__ static long z_off(int l) { return (long)l * NS * NP * WIDTH * TP; }
    __host__ __device__ static long s_tile(int nl) { return s_off(nl + 1); }
    __host__ __device__ static long z_tile(int nl) { return z_off(nl) + (long)NS * NP * 16 * TP; }
};

// ------------------------------------------------------------------------------------------
// repack: flat fp32 parameters -> MFMA A-operand fragments (hi [+ scaled lo]) + fp32 side tables
// ------------------------------------------------------------------------------------------
template <class Op, int SPLIT, int WIDTH>
__global__ __launch_bounds__(256) void repack_kernel(const RepackArgs a) {
    typedef FragIndex<WIDTH> FI;
    constexpr int NP = SPLIT == 3 ? 2 : 1;
    const int nl = a.net.nl, H = a.net.h, NO = a.net.nout;
    const int FM = FI::fm(nl), NF = FI::total(nl);
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long nthreads = (long)gridDim.x * blockDim.x;
    bool out_of_range = false;      // a weight whose scaled value leaves fp16's range: the fused kernel would compute with +-inf
    for (long i = gid; i < (long)NF * 64; i += nthreads) {
        const int frag = (int)(i >> 6), lane = (int)(i & 63), c = lane & 15, q = lane >> 4;
        // decode the fragment kind
        int kind, l = 0, mb = 0, kk = 0, rem = frag;
        if (rem < FM) { kind = 0; l = 1 + rem / (FI::WB * FI::KS); mb = (rem / FI::KS) % FI::WB; kk = rem % FI::KS; }
        else if ((rem -= FM) < FI::KS) { kind = 1; kk = rem; }
        else if ((rem -= FI::KS) < FM) { kind = 2; l = 1 + rem / (FI::WB * FI::KS); mb = (rem / FI::KS) % FI::WB; kk = rem % FI::KS; }
        else { kind = 3; mb = rem - FM; }
        uint32_t hi[4], lo[4];
        float wv[8];
        for (int d = 0; d < 4; ++d) {
            float w2[2];
            for (int e = 0; e < 2; ++e) {
                const int j = 2 * d + e;
                int in, out, n_in, n_out, wl;
                if (kind == 0) { wl = l; in = kmap(kk, q, j); out = 16 * mb + c; n_in = H; n_out = H; }
                else if (kind == 1) { wl = nl; in = kmap(kk, q, j); out = c; n_in = H; n_out = NO; }
                else if (kind == 2) { wl = l; in = 16 * mb + c; out = kmap(kk, q, j); n_in = H; n_out = H; }
                else { wl = nl; in = 16 * mb + c; out = kmap(0, q, j); n_in = H; n_out = NO; }
                w2[e] = (in < n_in && out < n_out) ? a.params[a.net.w_off[wl] + in * n_out + out] : 0.0f;
            }
            wv[2 * d] = w2[0];
            wv[2 * d + 1] = w2[1];
            hi[d] = pack2<Op>(w2[0], w2[1]);
            lo[d] = pack2<Op>((w2[0] - round16<Op>(w2[0])) * Op::LO_SCALE, (w2[1] - round16<Op>(w2[1])) * Op::LO_SCALE);
        }
        // stored parts: [hi, lo] (split) or [hi]
        constexpr int NPS = SPLIT == 3 ? 2 : 1;
        u32x4 vh = {hi[0], hi[1], hi[2], hi[3]};
        a.frags[((long)frag * NPS + 0) * 64 + lane] = vh;
        if (NP == 2) {
            u32x4 vl = {lo[0], lo[1], lo[2], lo[3]};
            a.frags[((long)frag * NPS + 1) * 64 + lane] = vl;
        }
        // fused-kernel format (pinn_fused.hpp): the WEIGHT carries a scale.  With V = FUSED_WEIGHT_SCALE * w:
        //   part 0 = T(V), part 1 = T(V - part0)  (the remainder stays in the normal range because V is large),  part 2 = T(part0 / LO_SCALE)
        // so that  part0.x_hi + part0.x_lo + part1.x_hi  = FUSED_WEIGHT_SCALE * (w.x)  accumulates in ONE MFMA chain (forward, unscaled x_lo)
        // and      part0.z_hi + part2.z_lo' + part1.z_hi = FUSED_WEIGHT_SCALE * (w.z)  (reverse, z_lo' = LO_SCALE-scaled low part).
        if (a.frags_fused != nullptr) {
            if (NP == 2) {
                uint32_t p0[4], p1[4], p2[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const float v0 = wv[2 * d] * FUSED_WEIGHT_SCALE, v1 = wv[2 * d + 1] * FUSED_WEIGHT_SCALE;
                    out_of_range |= !(__builtin_fabsf(v0) <= Op::OPERAND_MAX) || !(__builtin_fabsf(v1) <= Op::OPERAND_MAX);      // (also true for NaN; per operand type: bf16x3 holds fp32's range)
                    p0[d] = pack2<Op>(v0, v1);
                    p1[d] = pack2<Op>(v0 - round16<Op>(v0), v1 - round16<Op>(v1));
                    p2[d] = pack2<Op>(round16<Op>(v0) * (1.0f / Op::LO_SCALE), round16<Op>(v1) * (1.0f / Op::LO_SCALE));
                }
                a.frags_fused[((long)frag * 3 + 0) * 64 + lane] = u32x4{p0[0], p0[1], p0[2], p0[3]};
                a.frags_fused[((long)frag * 3 + 1) * 64 + lane] = u32x4{p1[0], p1[1], p1[2], p1[3]};
                a.frags_fused[((long)frag * 3 + 2) * 64 + lane] = u32x4{p2[0], p2[1], p2[2], p2[3]};
            } else {
                a.frags_fused[(long)frag * 64 + lane] = vh;
            }
        }
    }
    // fp32 side tables
    for (long i = gid; i < WIDTH; i += nthreads) {
        const int f = (int)i;
        const bool ok = f < H;
        if (a.net.din == 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) a.w0p[8 * f + k] = ok ? a.params[a.net.w_off[0] + k * H + f] : 0.0f;
            a.w0p[8 * f + 4] = ok ? a.params[a.net.b_off[0] + f] : 0.0f;
            a.w0p[8 * f + 5] = a.w0p[8 * f + 6] = a.w0p[8 * f + 7] = 0.0f;
        } else {
            a.w0p[4 * f + 0] = ok ? a.params[a.net.w_off[0] + 0 * H + f] : 0.0f;
            a.w0p[4 * f + 1] = ok ? a.params[a.net.w_off[0] + 1 * H + f] : 0.0f;
            a.w0p[4 * f + 2] = ok ? a.params[a.net.w_off[0] + 2 * H + f] : 0.0f;
            a.w0p[4 * f + 3] = ok ? a.params[a.net.b_off[0] + f] : 0.0f;
        }
    }
    for (long i = gid; i < (long)(nl - 1) * WIDTH; i += nthreads) {
        const int l = 1 + (int)(i / WIDTH), f = (int)(i % WIDTH);
        a.bias_mid[i] = f < H ? a.params[a.net.b_off[l] + f] : 0.0f;
    }
    for (long i = gid; i < NOUT_PAD; i += nthreads) a.bias_last[i] = i < NO ? a.params[a.net.b_off[nl] + i] : 0.0f;
    // one flag per block, rewritten by every repack (no reset, no atomics): read by the fused path's reduction (reduce_grad_loss_kernel)
    if (a.wflags != nullptr) {
        __shared__ int any;
        if (threadIdx.x == 0) any = 0;
        __syncthreads();
        if (out_of_range) any = 1;
        __syncthreads();
        if (threadIdx.x == 0) a.wflags[blockIdx.x] = any;
    }
}

// ------------------------------------------------------------------------------------------
// chain kernel: forward (value + tangents), residual/loss head, reverse chain; spills the
// per-layer states S and adjoints Z as [feature][point] panels for the weight-gradient kernel.
// One wave owns a tile of TP = 16*NB points; no LDS, no barriers.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void tanh_act(float z, float& h, float& sd) {
    // tanh(z) = 1 - 2/(1 + e^{2z});  exp2/rcp are the hardware transcendental ops
    const float e = __builtin_amdgcn_exp2f(z * 2.8853900817779268f);
    const float r = __builtin_amdgcn_rcpf(1.0f + e);
    h = 1.0f - 2.0f * r;
    sd = 1.0f - h * h;
}

template <class Op, int SPLIT, int WIDTH, int NB, int NS, int HEAD>
struct Chain {
    static constexpr int WB = WIDTH / 16, KS = WIDTH / 32, NP = SPLIT == 3 ? 2 : 1, TP = 16 * NB;
    static constexpr int NPS = SPLIT == 3 ? 2 : 1;     // stored weight-fragment parts (see repack_kernel)
    static constexpr int DIN = head_is_3d(HEAD) ? 4 : 3;                       // inputs (x, y, t) or (x, y, z, t)
    static constexpr bool SECOND = NS == 5 && DIN == 3;                         // stream 4 = second time derivative (plate, PLATE:427-433)
    static constexpr int NT = NS >= 4 ? (SECOND ? 3 : NS - 1) : 0;              // first-order tangent streams 1..NT (one per input)
    static constexpr int NOG = DIN == 4 ? 16 : 8;                               // outputs every lane gathers for the head
    static constexpr int LT = DIN == 4 ? LOSS_SLOTS_3D : 8;                     // loss partial slots per wave
    static constexpr float INV_LS = 1.0f / Op::LO_SCALE;
    typedef PanelGeom<WIDTH, NB, NS, NP> PG;
    typedef FragIndex<WIDTH> FI;

    // Take one 16-feature block of per-point values (vals[s][nb][r], feature 16*mb+4q+r), turn it
    // into k-slots of the next GEMM's B fragments (hi [+lo]) and, if `panel` is given, store it
    // into the [feature][point] panel of this tile.  KSN = k-steps of the destination fragments.
    template <int KSN, int MB>
    static __device__ __forceinline__ void emit(u32x4 (&Bn)[NS][NB][KSN][NP], const float (&vals)[NS][NB][4],
                                                uint16_t* panel, int rows, int c, int q) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const float* v = vals[s][nb];
                uint32_t h0, h1, l0 = 0, l1 = 0;
                if (NP == 2) {
                    split4<Op>(v[0], v[1], v[2], v[3], h0, h1, l0, l1);
                    Bn[s][nb][MB >> 1][NP - 1][(MB & 1) * 2 + 0] = l0;
                    Bn[s][nb][MB >> 1][NP - 1][(MB & 1) * 2 + 1] = l1;
                } else {
                    h0 = pack2<Op>(v[0], v[1]);
                    h1 = pack2<Op>(v[2], v[3]);
                }
                Bn[s][nb][MB >> 1][0][(MB & 1) * 2 + 0] = h0;
                Bn[s][nb][MB >> 1][0][(MB & 1) * 2 + 1] = h1;
                if (panel) {
                    uint16_t* p = panel + ((long)(s * NP) * rows + 16 * MB + 4 * q) * TP + 16 * nb + c;
                    p[0 * TP] = (uint16_t)(h0 & 0xffffu);
                    p[1 * TP] = (uint16_t)(h0 >> 16);
                    p[2 * TP] = (uint16_t)(h1 & 0xffffu);
                    p[3 * TP] = (uint16_t)(h1 >> 16);
                    if (NP == 2) {
                        uint16_t* pl = p + (long)rows * TP;
                        pl[0 * TP] = (uint16_t)(l0 & 0xffffu);
                        pl[1 * TP] = (uint16_t)(l0 >> 16);
                        pl[2 * TP] = (uint16_t)(l1 & 0xffffu);
                        pl[3 * TP] = (uint16_t)(l1 >> 16);
                    }
                }
            }
        }
    }

    // Spill of a chain-layout tensor to its [stream][part][feature row][point] panel with WIDE stores.  A lane holds, for its point,
    // four features of two 16-feature blocks per fragment register quad -- written element by element that is eight 2-byte stores
    // per lane and fragment, and the chain kernel was bound by its own store instructions (round 2: 30 % of its time).  Instead the
    // fragment goes through a 1 KB per-wave LDS record (one ds_write_b128 per lane, lane records rotated as in the fused kernel)
    // and comes back transposed by ds_read_b64_tr_b16: lane (c, q) then holds feature row c of block (q >> 1), points 8 (q & 1)..+7
    // = 16 contiguous bytes of the panel; the 64 lanes of one store cover 1 KB of consecutive rows.
    static __device__ __forceinline__ unsigned img_record(int point, int qq) { return (unsigned)((((point + 4 * qq) & 15) + 16 * qq) * 16); }
    template <int KSF, int PARTS>
    static __device__ __forceinline__ void spill_frags(char* rec, const u32x4 (&F)[NS][NB][KSF][NP], uint16_t* panel, int rows, int c, int q) {
        char* wr = rec + img_record(c, q);
        const int p0 = 8 * (q & 1) + (c >> 2), sub = c & 3;
        const char* r0 = rec + img_record(p0, sub) + 8 * (q >> 1);
        const char* r1 = rec + img_record(p0 + 4, sub) + 8 * (q >> 1);
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int kk = 0; kk < KSF; ++kk)
#pragma unroll
                    for (int p = 0; p < PARTS; ++p) {
                        *reinterpret_cast<u32x4*>(wr) = F[s][nb][kk][p];
                        const v4i16 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)r0);
                        const v4i16 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)r1);
                        const u32x2 d0 = __builtin_bit_cast(u32x2, v0), d1 = __builtin_bit_cast(u32x2, v1);
                        const int row = 32 * kk + 16 * (q >> 1) + c;
                        if (row < rows)
                            *reinterpret_cast<u32x4*>(panel + ((long)(s * PARTS + p) * rows + row) * TP + 16 * nb + 8 * (q & 1)) = u32x4{d0[0], d0[1], d1[0], d1[1]};
                    }
    }

    // acc[s][nb] (+ corr) = A(mb) . B over KSB k-steps
    // FENCE (first block of a layer): the operand fragments were just written by emit()'s inline assembly, and hipcc pads the
    // "vector write -> MFMA operand read" wait states only between instructions it knows.  The fragments written last (k-step
    // KSB-1) pass through a statement that opens with those wait states, so no MFMA reading them can be scheduled closer.
    template <int KSB, bool FENCE = false>
    static __device__ __forceinline__ void gemm_block(const u32x4* Afr, int lane, const u32x4 (&B)[NS][NB][KSB][NP],
                                                      f32x4 (&acc)[NS][NB], f32x4 (&accc)[NS][NB]) {
#if defined(__AMDGCN__)
        if constexpr (FENCE) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int p = 0; p < NP; ++p) asm volatile("s_nop 1" : "+v"(const_cast<u32x4&>(B[s][nb][KSB - 1][p])));
        }
#endif
        u32x4 A[KSB][NP];
#pragma unroll
        for (int kk = 0; kk < KSB; ++kk)
#pragma unroll
            for (int p = 0; p < NP; ++p) A[kk][p] = Afr[((long)kk * NPS + p) * 64 + lane];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                f32x4 m = {0.f, 0.f, 0.f, 0.f}, cc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < KSB; ++kk) {
                    m = Op::mfma(A[kk][0], B[s][nb][kk][0], m);
                    if (NP == 2) {
                        cc = Op::mfma(A[kk][0], B[s][nb][kk][1], cc);
                        cc = Op::mfma(A[kk][1], B[s][nb][kk][0], cc);
                    }
                }
                acc[s][nb] = m;
                accc[s][nb] = cc;
            }
    }

    static __device__ __forceinline__ float comb(const f32x4& m, const f32x4& cc, int r) {
        return NP == 2 ? m[r] + cc[r] * INV_LS : m[r];
    }

    // load the stored state (h, hdot_k) of one feature block from the S panel of layer l
    // this lane's state values (features 16 MB + 4q + r of its point) from the S panel.  (Measured in round 2: keeping a second copy
    // of the states as register images for this read -- one 8-byte load instead of four 2-byte ones -- is SLOWER, 24.0 against
    // 21.4 ms per 1 M points for the 8x80 net: this path is bound by the bytes it spills, not by its load instructions.)
    template <int MB>
    static __device__ __forceinline__ void load_state(const uint16_t* panel, int c, int q, float (&st)[NS][NB][4]) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const uint16_t* p = panel + ((long)(s * NP) * WIDTH + 16 * MB + 4 * q) * TP + 16 * nb + c;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = cvt16<Op>(p[r * TP]);
                    if (NP == 2) v += cvt16<Op>(p[(long)WIDTH * TP + r * TP]) * INV_LS;
                    st[s][nb][r] = v;
                }
            }
    }

    // reverse of (h = tanh z, h_k = (1-h^2) z_k [, h_tt = (1-h^2) z_tt - 2 h h_t z_t]):   INF:131-133 (gradient of TanhGrad).
    // Only post-activation state is needed:  d h_tt/d z = -2 h h_tt - 2 h_t^2 ,  d h_tt/d z_t = -4 h h_t.
    static __device__ __forceinline__ void act_bwd(const float (&st)[NS][NB][4], const f32x4 (&acc)[NS][NB],
                                                   const f32x4 (&accc)[NS][NB], float (&vals)[NS][NB][4]) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float h = st[0][nb][r];
                const float sd = 1.0f - h * h;
                const float hb = comb(acc[0][nb], accc[0][nb], r);
                float dot = 0.0f;
#pragma unroll
                for (int s = 1; s <= NT; ++s) {
                    const float hdb = comb(acc[s][nb], accc[s][nb], r);
                    dot += hdb * st[s][nb][r];
                    vals[s][nb][r] = sd * hdb;
                }
                float zb = sd * hb - 2.0f * h * dot;
                if constexpr (SECOND) {
                    const float httb = comb(acc[4][nb], accc[4][nb], r);
                    const float ht = st[3][nb][r], htt = st[4][nb][r];
                    vals[4][nb][r] = sd * httb;
                    vals[3][nb][r] -= 4.0f * h * ht * httb;
                    zb += httb * (-2.0f * h * htt - 2.0f * ht * ht);
                }
                vals[0][nb][r] = zb;
            }
    }

    template <int MB>
    struct MbLoop {
        // forward first layer (K = 3, plain VALU): INF:191-195 with the tangent seeds e_k * sx_k
        static __device__ __forceinline__ void first(const ChainArgs& a, const float (&xin)[NB][DIN], u32x4 (&Bn)[NS][NB][KS][NP],
                                                     uint16_t* panel, int c, int q) {
            float vals[NS][NB][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float w[5];              // input weights of feature 16 MB + 4q + r, then its bias
                if constexpr (DIN == 4) {
                    const f32x4 wa = *reinterpret_cast<const f32x4*>(a.pw.w0p + 8 * (16 * MB + 4 * q + r));
                    w[0] = wa[0]; w[1] = wa[1]; w[2] = wa[2]; w[3] = wa[3];
                    w[4] = a.pw.w0p[8 * (16 * MB + 4 * q + r) + 4];
                } else {
                    const f32x4 wa = *reinterpret_cast<const f32x4*>(a.pw.w0p + 4 * (16 * MB + 4 * q + r));
                    w[0] = wa[0]; w[1] = wa[1]; w[2] = wa[2]; w[3] = 0.0f;
                    w[4] = wa[3];
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    float z = w[4];
#pragma unroll
                    for (int k = 0; k < DIN; ++k) z += w[k] * xin[nb][k];
                    float h, sd;
                    tanh_act(z, h, sd);
                    vals[0][nb][r] = h;
#pragma unroll
                    for (int s = 1; s <= NT; ++s) vals[s][nb][r] = sd * (a.sx[s - 1] * w[s - 1]);
                    if constexpr (SECOND) {                  // z_tt = 0 at the first layer: h_tt = -2 h h_t z_t
                        const float zt = a.sx[2] * w[2];
                        vals[4][nb][r] = -2.0f * h * vals[3][nb][r] * zt;
                    }
                }
            }
            emit<KS, MB>(Bn, vals, panel, WIDTH, c, q);
            if constexpr (MB + 1 < WB) MbLoop<MB + 1>::first(a, xin, Bn, panel, c, q);
        }
        // forward hidden layer: z = W^T h + b, h' = tanh z, hdot' = (1-h'^2) W^T hdot     INF:192-195
        static __device__ __forceinline__ void fwd(const u32x4* Al, const float* bl, int lane, const u32x4 (&B)[NS][NB][KS][NP],
                                                   u32x4 (&Bn)[NS][NB][KS][NP], uint16_t* panel, int c, int q) {
            f32x4 acc[NS][NB], accc[NS][NB];
            gemm_block<KS, MB == 0>(Al + (long)MB * KS * NPS * 64, lane, B, acc, accc);
            const f32x4 bias = *reinterpret_cast<const f32x4*>(bl + 16 * MB + 4 * q);
            float vals[NS][NB][4];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float h, sd;
                    tanh_act(comb(acc[0][nb], accc[0][nb], r) + bias[r], h, sd);
                    vals[0][nb][r] = h;
#pragma unroll
                    for (int s = 1; s <= NT; ++s) vals[s][nb][r] = sd * comb(acc[s][nb], accc[s][nb], r);
                    if constexpr (SECOND) {                  // h_tt = (1-h^2) z_tt - 2 h h_t z_t
                        const float zt = comb(acc[3][nb], accc[3][nb], r);
                        vals[4][nb][r] = sd * comb(acc[4][nb], accc[4][nb], r) - 2.0f * h * vals[3][nb][r] * zt;
                    }
                }
            emit<KS, MB>(Bn, vals, panel, WIDTH, c, q);
            if constexpr (MB + 1 < WB) MbLoop<MB + 1>::fwd(Al, bl, lane, B, Bn, panel, c, q);
        }
        // reverse through one weight layer (KSB k-steps of its outputs) and the activation below it
        template <int KSB>
        static __device__ __forceinline__ void bwd(const u32x4* Al, int lane, const u32x4 (&Zf)[NS][NB][KSB][NP],
                                                   const uint16_t* spanel, u32x4 (&Zn)[NS][NB][KS][NP], uint16_t* zpanel, int c, int q) {
            f32x4 acc[NS][NB], accc[NS][NB];
            gemm_block<KSB, MB == 0>(Al + (long)MB * KSB * NPS * 64, lane, Zf, acc, accc);
            float st[NS][NB][4], vals[NS][NB][4];
            load_state<MB>(spanel, c, q, st);
            act_bwd(st, acc, accc, vals);
            emit<KS, MB>(Zn, vals, zpanel, WIDTH, c, q);
            if constexpr (MB + 1 < WB) MbLoop<MB + 1>::template bwd<KSB>(Al, lane, Zf, spanel, Zn, zpanel, c, q);
        }
    };

    static __device__ void run(const ChainArgs& a) {
        const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
        const int wpb = blockDim.x >> 6;
        const long gwave = (long)blockIdx.x * wpb + (threadIdx.x >> 6), nwaves = (long)gridDim.x * wpb;
        const int nl = a.net.nl;
        constexpr bool FWD_ONLY = HEAD == HEAD_FIELDS || HEAD == HEAD_FIELDS3D;
        constexpr bool SPILL = !FWD_ONLY;
        __shared__ __attribute__((aligned(16))) char spill_lds[4 * 1024];      // one 1 KB transpose record per wave (256-thread blocks)
        char* rec = spill_lds + (threadIdx.x >> 6) * 1024;
        float lsum[LT];
#pragma unroll
        for (int i = 0; i < LT; ++i) lsum[i] = 0.0f;

        for (long tile = gwave; tile < a.ntiles; tile += nwaves) {
            uint16_t* St = SPILL ? a.S + tile * a.S_tile_stride : nullptr;
            uint16_t* Zt = SPILL ? a.Z + tile * a.Z_tile_stride : nullptr;
            float xin[NB][DIN];
            bool valid[NB];
            long pidx[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const long p = (a.tile0 + tile) * TP + 16 * nb + c;
                valid[nb] = p < a.n;
                pidx[nb] = valid[nb] ? p : a.n - 1;
                xin[nb][0] = a.x[pidx[nb]] * a.sx[0] + a.ox[0];
                xin[nb][1] = a.y[pidx[nb]] * a.sx[1] + a.ox[1];
                if constexpr (DIN == 4) {
                    xin[nb][2] = a.z[pidx[nb]] * a.sx[2] + a.ox[2];
                    xin[nb][3] = a.t[pidx[nb]] * a.sx[3] + a.ox[3];
                } else {
                    xin[nb][2] = a.t[pidx[nb]] * a.sx[2] + a.ox[2];
                }
            }
            // ---- S_0: the inputs as a 16-row panel (rows 0..DIN-1 = x'; tangent stream k has sx_k in row k)
            if (SPILL) {
                float v0[NS][NB][4];
#pragma unroll
                for (int s = 0; s < NS; ++s)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float v = 0.0f;
                            if (q == 0 && r < DIN && s <= NT) v = (s == 0) ? xin[nb][r] : (r == s - 1 ? a.sx[r] : 0.0f);
                            v0[s][nb][r] = v;
                        }
                u32x4 S0[NS][NB][1][NP];
#pragma unroll
                for (int s = 0; s < NS; ++s)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                        for (int p = 0; p < NP; ++p) S0[s][nb][0][p] = u32x4{0u, 0u, 0u, 0u};
                emit<1, 0>(S0, v0, nullptr, 16, c, q);
                spill_frags<1, NP>(rec, S0, St + PG::s_off(0), 16, c, q);
            }
            // ---- forward
            u32x4 B[NS][NB][KS][NP];
            MbLoop<0>::first(a, xin, B, nullptr, c, q);
            if (SPILL) spill_frags<KS, NP>(rec, B, St + PG::s_off(1), WIDTH, c, q);
            for (int l = 1; l < nl; ++l) {
                u32x4 Bn[NS][NB][KS][NP];
                MbLoop<0>::fwd(a.pw.frags + (long)FI::fwd_mid(l, 0, 0) * NPS * 64, a.pw.bias_mid + (long)(l - 1) * WIDTH, lane, B, Bn, nullptr, c, q);
                if (SPILL) spill_frags<KS, NP>(rec, Bn, St + PG::s_off(l + 1), WIDTH, c, q);
#pragma unroll
                for (int s = 0; s < NS; ++s)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                        for (int kk = 0; kk < KS; ++kk)
#pragma unroll
                            for (int p = 0; p < NP; ++p) B[s][nb][kk][p] = Bn[s][nb][kk][p];
            }
            // ---- output layer  Y = h W_L + b_L  (INF:196-198): lane holds outputs 4q+r of its point
            f32x4 yacc[NS][NB], yaccc[NS][NB];
            gemm_block<KS, true>(a.pw.frags + (long)FI::fwd_last(nl, 0) * NPS * 64, lane, B, yacc, yaccc);
            const f32x4 bl = *reinterpret_cast<const f32x4*>(a.pw.bias_last + 4 * q);
            // Every lane gathers the NOG (padded) outputs of its point: own block + the q^1 partner's (8 outputs), or all four (16)
            float Y[NS][NB][NOG];
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float own = comb(yacc[s][nb], yaccc[s][nb], r) + (s == 0 ? bl[r] : 0.0f);
                        if constexpr (NOG == 16) {
#pragma unroll
                            for (int qq = 0; qq < 4; ++qq) Y[s][nb][4 * qq + r] = __shfl(own, c + 16 * qq);
                        } else {
                            const float oth = __shfl_xor(own, 16);
                            Y[s][nb][r] = (q & 1) ? oth : own;
                            Y[s][nb][4 + r] = (q & 1) ? own : oth;
                        }
                    }
            // ---- head
            float adj[NS][NB][NOG];     // dL/d(stream s of output o) for the padded outputs
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const float vm = valid[nb] ? 1.0f : 0.0f;
                if constexpr (HEAD == HEAD_WAVE) {
                    // net_f_sig INF:221-265; outputs (u,v,ut,vt,s11,s22,s12); streams (value, d/dx, d/dy, d/dt)
                    const float(&V)[8] = Y[0][nb];
                    const float(&X)[8] = Y[1][nb];
                    const float(&Yy)[8] = Y[2][nb];
                    const float(&T)[8] = Y[3][nb];
                    const float e11 = X[0], e22 = Yy[1], e12 = Yy[0] + X[1];          // INF:216-218
                    float f[7];
                    f[0] = X[4] + Yy[6] - a.rho * T[2];                               // f_u   INF:262
                    f[1] = Yy[5] + X[6] - a.rho * T[3];                               // f_v   INF:263
                    f[2] = T[0] - V[2];                                               // f_ut  INF:248
                    f[3] = T[1] - V[3];                                               // f_vt  INF:249
                    f[4] = V[4] - (a.c1 * e11 + a.c2 * e22);                          // f_s11 INF:244
                    f[5] = V[5] - (a.c2 * e11 + a.c1 * e22);                          // f_s22 INF:246
                    f[6] = V[6] - a.G * e12;                                          // f_s12 INF:245
                    float g[7];
#pragma unroll
                    for (int i = 0; i < 7; ++i) {
                        if (q == 0) lsum[i] += vm * f[i] * f[i];
                        g[i] = 2.0f * a.tw[i] * f[i] * vm;
                    }
#pragma unroll
                    for (int s = 0; s < NS; ++s)
#pragma unroll
                        for (int o = 0; o < 8; ++o) adj[s][nb][o] = 0.0f;
                    adj[0][nb][2] = -g[2];
                    adj[0][nb][3] = -g[3];
                    adj[0][nb][4] = g[4];
                    adj[0][nb][5] = g[5];
                    adj[0][nb][6] = g[6];
                    adj[1][nb][0] = -a.c1 * g[4] - a.c2 * g[5];
                    adj[1][nb][1] = -a.G * g[6];
                    adj[1][nb][4] = g[0];
                    adj[1][nb][6] = g[1];
                    adj[2][nb][0] = -a.G * g[6];
                    adj[2][nb][1] = -a.c2 * g[4] - a.c1 * g[5];
                    adj[2][nb][5] = g[1];
                    adj[2][nb][6] = g[0];
                    adj[3][nb][0] = g[2];
                    adj[3][nb][1] = g[3];
                    adj[3][nb][2] = -a.rho * g[0];
                    adj[3][nb][3] = -a.rho * g[1];
                } else if constexpr (HEAD == HEAD_NC3D) {
                    // 3-D Navier-Cauchy residuals (oracle/nc3d_oracle.py: the 3-D statement of INF:221-265); outputs
                    // (u,v,w, ut,vt,wt, s11,s22,s33, s12,s13,s23); streams (value, d/dx, d/dy, d/dz, d/dt); c1 = lambda+2G, c2 = lambda
                    const float(&V)[16] = Y[0][nb];
                    const float(&X)[16] = Y[1][nb];
                    const float(&Yy)[16] = Y[2][nb];
                    const float(&Zz)[16] = Y[3][nb];
                    const float(&T)[16] = Y[4][nb];
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
                        g[i] = 2.0f * a.tw[i] * f[i] * vm;
                    }
#pragma unroll
                    for (int s = 0; s < NS; ++s)
#pragma unroll
                        for (int o = 0; o < NOG; ++o) adj[s][nb][o] = 0.0f;
                    // value stream
                    adj[0][nb][3] = -g[3];
                    adj[0][nb][4] = -g[4];
                    adj[0][nb][5] = -g[5];
#pragma unroll
                    for (int i = 6; i < 12; ++i) adj[0][nb][i] = g[i];
                    // d/dx
                    adj[1][nb][0] = -(a.c1 * g[6] + a.c2 * (g[7] + g[8]));
                    adj[1][nb][1] = -a.G * g[9];
                    adj[1][nb][2] = -a.G * g[10];
                    adj[1][nb][6] = g[0];
                    adj[1][nb][9] = g[1];
                    adj[1][nb][10] = g[2];
                    // d/dy
                    adj[2][nb][0] = -a.G * g[9];
                    adj[2][nb][1] = -(a.c1 * g[7] + a.c2 * (g[6] + g[8]));
                    adj[2][nb][2] = -a.G * g[11];
                    adj[2][nb][9] = g[0];
                    adj[2][nb][7] = g[1];
                    adj[2][nb][11] = g[2];
                    // d/dz
                    adj[3][nb][0] = -a.G * g[10];
                    adj[3][nb][1] = -a.G * g[11];
                    adj[3][nb][2] = -(a.c1 * g[8] + a.c2 * (g[6] + g[7]));
                    adj[3][nb][10] = g[0];
                    adj[3][nb][11] = g[1];
                    adj[3][nb][8] = g[2];
                    // d/dt
                    adj[4][nb][0] = g[3];
                    adj[4][nb][1] = g[4];
                    adj[4][nb][2] = g[5];
                    adj[4][nb][3] = -a.rho * g[0];
                    adj[4][nb][4] = -a.rho * g[1];
                    adj[4][nb][5] = -a.rho * g[2];
                } else if constexpr (HEAD == HEAD_PLATE) {
                    // composite F = P + D*N (PLATE:383-387) with product-rule derivatives, then net_f_sig PLATE:404-439
                    // outputs (u,v,s11,s22,s12); streams (value, x, y, t, tt); aux = [D|P][stream][field][n]
                    float D[5][5], F[5][5];
#pragma unroll
                    for (int st = 0; st < 5; ++st)
#pragma unroll
                        for (int o = 0; o < 5; ++o) {
                            D[st][o] = a.aux[((long)(0 * 5 + st) * 5 + o) * a.n + pidx[nb]];
                            F[st][o] = a.aux[((long)(1 * 5 + st) * 5 + o) * a.n + pidx[nb]];      // start from P
                        }
#pragma unroll
                    for (int o = 0; o < 5; ++o) {
                        const float n0 = Y[0][nb][o];
                        F[0][o] += D[0][o] * n0;
#pragma unroll
                        for (int k = 1; k <= 3; ++k) F[k][o] += D[k][o] * n0 + D[0][o] * Y[k][nb][o];
                        F[4][o] += D[4][o] * n0 + 2.0f * D[3][o] * Y[3][nb][o] + D[0][o] * Y[4][nb][o];
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
                        g[i] = 2.0f * a.tw[i] * f[i] * vm;
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
                    for (int s = 0; s < NS; ++s)
#pragma unroll
                        for (int o = 0; o < 8; ++o) adj[s][nb][o] = 0.0f;
#pragma unroll
                    for (int o = 0; o < 5; ++o) {
                        adj[0][nb][o] = Fb[0][o] * D[0][o] + Fb[1][o] * D[1][o] + Fb[2][o] * D[2][o] + Fb[3][o] * D[3][o] + Fb[4][o] * D[4][o];
                        adj[1][nb][o] = Fb[1][o] * D[0][o];
                        adj[2][nb][o] = Fb[2][o] * D[0][o];
                        adj[3][nb][o] = Fb[3][o] * D[0][o] + 2.0f * Fb[4][o] * D[3][o];
                        adj[4][nb][o] = Fb[4][o] * D[0][o];
                    }
                } else if constexpr (HEAD == HEAD_TRACTION) {
                    // net_t PLATE:452-461 on the composite values; aux rows: D0[0..4], P0[5..9], nx[10], ny[11]
                    float Fv[5], D0[5];
#pragma unroll
                    for (int o = 0; o < 5; ++o) {
                        D0[o] = a.aux[(long)o * a.n + pidx[nb]];
                        Fv[o] = a.aux[(long)(5 + o) * a.n + pidx[nb]] + D0[o] * Y[0][nb][o];
                    }
                    const float nx = a.aux[10L * a.n + pidx[nb]], ny = a.aux[11L * a.n + pidx[nb]];
                    const float tx = Fv[2] * nx + Fv[4] * ny, ty = Fv[4] * nx + Fv[3] * ny;
                    if (q == 0) {
                        lsum[0] += vm * tx * tx;
                        lsum[1] += vm * ty * ty;
                    }
                    const float gx = 2.0f * a.tw[0] * tx * vm, gy = 2.0f * a.tw[1] * ty * vm;
#pragma unroll
                    for (int o = 0; o < 8; ++o) adj[0][nb][o] = 0.0f;
                    adj[0][nb][2] = gx * nx * D0[2];
                    adj[0][nb][3] = gy * ny * D0[3];
                    adj[0][nb][4] = (gx * ny + gy * nx) * D0[4];
                } else if constexpr (HEAD == HEAD_STREAMS) {
                    // sum_{s,o} w[s][o] (Y[s][o] - target[s][o])^2 ; lsum[o] accumulates the weight-normalised sum over streams
#pragma unroll
                    for (int s = 0; s < NS; ++s)
#pragma unroll
                        for (int o = 0; o < 8; ++o) {
                            float d = 0.0f;
                            if (o < a.net.nout) d = Y[s][nb][o] - (a.aux ? a.aux[((long)s * a.net.nout + o) * a.n + pidx[nb]] : 0.0f);
                            const float w = a.w5[s][o];
                            if (q == 0) lsum[o] += vm * w * d * d;
                            adj[s][nb][o] = 2.0f * w * d * vm;
                        }
                } else if constexpr (HEAD == HEAD_DATA || HEAD == HEAD_DATA3D) {
                    // loss_IC / loss_SRC / loss_NB / loss_FIX (INF:111-118, CONF:145-146): sum_o w_o (Y_o - target_o)^2
#pragma unroll
                    for (int o = 0; o < NOG; ++o) {
                        float d = 0.0f;
                        if (o < a.net.nout) d = Y[0][nb][o] - (a.targets ? a.targets[(long)o * a.n + pidx[nb]] : 0.0f);
                        if (q == 0) lsum[o] += vm * d * d;
                        adj[0][nb][o] = 2.0f * a.tw[o] * d * vm;
                    }
                } else {
                    // predict (INF:337-347): write Y and its tangent streams, [NS*nout][n]
                    if (q == 0 && valid[nb]) {
#pragma unroll
                        for (int s = 0; s < NS; ++s)
#pragma unroll
                            for (int o = 0; o < NOG; ++o)
                                if (o < a.net.nout) a.fields_out[((long)s * a.net.nout + o) * a.n + pidx[nb]] = Y[s][nb][o];
                    }
                }
            }
            if constexpr (!FWD_ONLY) {
                // ---- Z_nl: adjoint of the outputs, lane keeps outputs 4q+r (zeros beyond the gathered ones)
                u32x4 ZL[NS][NB][1][NP];
                {
                    float vals[NS][NB][4];
#pragma unroll
                    for (int s = 0; s < NS; ++s)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float v = 0.0f;
#pragma unroll
                                for (int qq = 0; qq < NOG / 4; ++qq) v = q == qq ? adj[s][nb][4 * qq + r] : v;
                                vals[s][nb][r] = v;
                            }
#pragma unroll
                    for (int s = 0; s < NS; ++s)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                            for (int p = 0; p < NP; ++p) ZL[s][nb][0][p] = u32x4{0u, 0u, 0u, 0u};
                    emit<1, 0>(ZL, vals, nullptr, 16, c, q);
                    spill_frags<1, NP>(rec, ZL, Zt + PG::z_off(nl), 16, c, q);
                }
                // ---- reverse chain
                u32x4 Zc[NS][NB][KS][NP];
                MbLoop<0>::template bwd<1>(a.pw.frags + (long)FI::bwd_last(nl, 0) * NPS * 64, lane, ZL, St + PG::s_off(nl), Zc, nullptr, c, q);
                spill_frags<KS, NP>(rec, Zc, Zt + PG::z_off(nl - 1), WIDTH, c, q);
                for (int l = nl - 1; l >= 1; --l) {
                    u32x4 Zn[NS][NB][KS][NP];
                    MbLoop<0>::template bwd<KS>(a.pw.frags + (long)FI::bwd_mid(nl, l, 0, 0) * NPS * 64, lane, Zc, St + PG::s_off(l), Zn, nullptr, c, q);
                    spill_frags<KS, NP>(rec, Zn, Zt + PG::z_off(l - 1), WIDTH, c, q);
#pragma unroll
                    for (int s = 0; s < NS; ++s)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                            for (int kk = 0; kk < KS; ++kk)
#pragma unroll
                                for (int p = 0; p < NP; ++p) Zc[s][nb][kk][p] = Zn[s][nb][kk][p];
                }
            }
        }
        if constexpr (!FWD_ONLY) {
            // per-wave partial sums of squares (only q == 0 lanes hold data): reduce over the 16 points
#pragma unroll
            for (int i = 0; i < LT; ++i) {
                float v = lsum[i];
                v += __shfl_xor(v, 1);
                v += __shfl_xor(v, 2);
                v += __shfl_xor(v, 4);
                v += __shfl_xor(v, 8);
                if (lane == 0) a.loss_part[gwave * LT + i] = v;
            }
        }
    }
};

template <class Op, int SPLIT, int WIDTH, int NB, int NS, int HEAD>
__global__ __launch_bounds__(256) void chain_kernel(const ChainArgs a) {
    Chain<Op, SPLIT, WIDTH, NB, NS, HEAD>::run(a);
}

// ------------------------------------------------------------------------------------------
// weight-gradient kernel:  Wbar_l[in,out] = sum_tiles sum_streams S_l[in, pts] . Z_l[out, pts]^T
// grid = (chunks, nl+1); wave w of a block owns in-blocks w, w+NW, ...; contraction over points
// (32 per MFMA k-step) read straight from the [feature][point] panels -- no LDS.
// ------------------------------------------------------------------------------------------
// waves per weight-gradient block: 8 for the 128-wide variant so that a wave owns ONE in-block (with 4 waves the
// 2 x 8 accumulator blocks, main + correction, spilled 629 VGPRs and the kernel ran 6x slower than the 160-wide one)
template <int WIDTH>
struct WgradCfg { static constexpr int NW = WIDTH == 128 ? 8 : 4; };

template <class Op, int SPLIT, int WIDTH, int NB, int NS>
__global__ __launch_bounds__(64 * WgradCfg<WIDTH>::NW) void wgrad_kernel(const WgradArgs a) {
    constexpr int WB = WIDTH / 16, NP = SPLIT == 3 ? 2 : 1, TP = 16 * NB, NW = WgradCfg<WIDTH>::NW, IBW = (WB + NW - 1) / NW;
    constexpr int OBN = WB, ob0 = 0;
    constexpr float INV_LS = 1.0f / Op::LO_SCALE;
    typedef PanelGeom<WIDTH, NB, NS, NP> PG;
    const int l = blockIdx.y, nl = a.net.nl;
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4, wave = threadIdx.x >> 6;
    const int IB = l == 0 ? 1 : WB, OB = l == nl ? 1 : WB;
    const int rowsS = l == 0 ? 16 : WIDTH, rowsZ = l == nl ? 16 : WIDTH;
    const int real_in = l == 0 ? a.net.din : a.net.h, real_out = l == nl ? a.net.nout : a.net.h;

    f32x4 acc[IBW][OBN], accc[IBW][OBN], bacc[OBN], baccc[OBN];
#pragma unroll
    for (int ob = 0; ob < OBN; ++ob) {
        bacc[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
        baccc[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < IBW; ++i) {
            acc[i][ob] = f32x4{0.f, 0.f, 0.f, 0.f};
            accc[i][ob] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const uint32_t one2 = pack2<Op>(1.0f, 1.0f);
    const u32x4 ones = {one2, one2, one2, one2};

    const long nksteps = a.ntiles * TP / 32;
    for (long ks = blockIdx.x; ks < nksteps; ks += gridDim.x) {
        const long tile = NB == 2 ? ks : 2 * ks + (q >> 1);
        const int col0 = NB == 2 ? 8 * q : 8 * (q & 1);
        for (int s = 0; s < NS; ++s) {
            const uint16_t* Sp = a.S + tile * a.S_tile_stride + PG::s_off(l) + (long)(s * NP) * rowsS * TP + col0;
            const uint16_t* Zp = a.Z + tile * a.Z_tile_stride + PG::z_off(l) + (long)(s * NP) * rowsZ * TP + col0;
            u32x4 Zh[OBN], Zl[OBN];
#pragma unroll
            for (int ob = 0; ob < OBN; ++ob)
                if (ob0 + ob < OB) {
                    Zh[ob] = *reinterpret_cast<const u32x4*>(Zp + (long)(16 * (ob0 + ob) + c) * TP);
                    if (NP == 2) Zl[ob] = *reinterpret_cast<const u32x4*>(Zp + (long)rowsZ * TP + (long)(16 * (ob0 + ob) + c) * TP);
                }
#pragma unroll
            for (int i = 0; i < IBW; ++i) {
                const int ib = wave + NW * i;
                if (ib < IB) {
                    const u32x4 Ah = *reinterpret_cast<const u32x4*>(Sp + (long)(16 * ib + c) * TP);
                    u32x4 Al = {0u, 0u, 0u, 0u};
                    if (NP == 2) Al = *reinterpret_cast<const u32x4*>(Sp + (long)rowsS * TP + (long)(16 * ib + c) * TP);
#pragma unroll
                    for (int ob = 0; ob < OBN; ++ob)
                        if (ob0 + ob < OB) {
                            acc[i][ob] = Op::mfma(Ah, Zh[ob], acc[i][ob]);
                            if (NP == 2) {
                                accc[i][ob] = Op::mfma(Ah, Zl[ob], accc[i][ob]);
                                accc[i][ob] = Op::mfma(Al, Zh[ob], accc[i][ob]);
                            }
                        }
                }
            }
            if (s == 0 && wave == 0) {   // bias gradient: ones^T . Z (value stream)
#pragma unroll
                for (int ob = 0; ob < OBN; ++ob)
                    if (ob0 + ob < OB) {
                        bacc[ob] = Op::mfma(ones, Zh[ob], bacc[ob]);
                        if (NP == 2) baccc[ob] = Op::mfma(ones, Zl[ob], baccc[ob]);
                    }
            }
        }
    }
    float* part = a.partial + (long)blockIdx.x * a.net.nparams;
#pragma unroll
    for (int i = 0; i < IBW; ++i) {
        const int ib = wave + NW * i;
        if (ib < IB) {
#pragma unroll
            for (int ob = 0; ob < OBN; ++ob)
                if (ob0 + ob < OB) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int in = 16 * ib + 4 * q + r, out = 16 * (ob0 + ob) + c;
                        if (in < real_in && out < real_out) {
                            const float v = NP == 2 ? acc[i][ob][r] + accc[i][ob][r] * INV_LS : acc[i][ob][r];
                            float* dst = part + a.net.w_off[l] + in * real_out + out;
                            *dst = a.first_pass ? v : *dst + v;
                        }
                    }
                }
        }
    }
    if (wave == 0 && q == 0) {
#pragma unroll
        for (int ob = 0; ob < OBN; ++ob)
            if (ob0 + ob < OB) {
                const int out = 16 * (ob0 + ob) + c;
                if (out < real_out) {
                    const float v = NP == 2 ? bacc[ob][0] + baccc[ob][0] * INV_LS : bacc[ob][0];
                    float* dst = part + a.net.b_off[l] + out;
                    *dst = a.first_pass ? v : *dst + v;
                }
            }
    }
}

// ------------------------------------------------------------------------------------------
// small reductions / optimizer
// ------------------------------------------------------------------------------------------
// grad[p] = (accumulate ? grad[p] : 0) + scale * sum_c partial[c][p]
// A block owns 64 parameters; its 4 waves each sum every 4th partial with four independent accumulators (the loop is
// latency-bound otherwise), then the 16 sub-sums are combined in a fixed order: deterministic, no atomics.
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void reduce_grad_kernel(const float* partial, int nchunks, int nparams, float scale,
                                                          float* grad, int accumulate) {
    __shared__ float sub[4][64];
    const int pl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const long p = (long)blockIdx.x * 64 + pl;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    if (p < nparams) {
        int cidx = sl;
        for (; cidx + 12 < nchunks; cidx += 16) {
            s0 += partial[(long)cidx * nparams + p];
            s1 += partial[(long)(cidx + 4) * nparams + p];
            s2 += partial[(long)(cidx + 8) * nparams + p];
            s3 += partial[(long)(cidx + 12) * nparams + p];
        }
        for (; cidx < nchunks; cidx += 4) s0 += partial[(long)cidx * nparams + p];
    }
    sub[sl][pl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && p < nparams) {
        const float s = (sub[0][pl] + sub[1][pl]) + (sub[2][pl] + sub[3][pl]);
        grad[p] = (accumulate ? grad[p] : 0.0f) + scale * s;
    }
}

// Fused-path epilogue: reduce_grad_kernel's blocks plus one extra block per point set (the last nsets blocks) that do
// reduce_loss_kernel's job, so that a call ends with one launch instead of two (the small configurations are bound by the chain
// of short launches).  loss_part is [wave][slots][8] (set k in slot k); set k's sums go to loss_out.p[k][0..nterms).
struct LossOuts { float* p[4]; };
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void reduce_grad_loss_kernel(const float* partial, int nchunks, int nparams, float scale, float* grad,
                                                               int accumulate, const float* loss_part, long nwaves, int nterms, int nsets,
                                                               int slots, LossOuts loss_out, const int* wflags = nullptr, int nflags = 0) {
    // Weights outside the fused format's range (repack_kernel's flags): the launch computed with infinities.  Its results are replaced by
    // NaN as a whole -- gradient and sums -- so that the condition is unmistakable (include/pinn_hip.h, PINN_FLAG_TWO_KERNEL).
    __shared__ int bad_weights;
    if (threadIdx.x == 0) bad_weights = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < nflags; i += 256)
        if (wflags[i]) bad_weights = 1;
    __syncthreads();
    const float poison = bad_weights ? __builtin_nanf("") : 0.0f;
    const int grad_blocks = (int)gridDim.x - nsets;
    if ((int)blockIdx.x >= grad_blocks) {
        const int k = (int)blockIdx.x - grad_blocks;
        const int term = threadIdx.x >> 5, sub = threadIdx.x & 31;
        double s = 0.0;
        if (term < nterms)
            for (long w = sub; w < nwaves; w += 32) s += (double)loss_part[(w * slots + k) * 8 + term];
        float v = (float)s;
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 8);
        v += __shfl_xor(v, 16);
        if (sub == 0 && term < nterms && loss_out.p[k] != nullptr) loss_out.p[k][term] = v + poison;
        return;
    }
    __shared__ float sub[4][64];
    const int pl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const long p = (long)blockIdx.x * 64 + pl;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    if (p < nparams) {
        int cidx = sl;
        for (; cidx + 12 < nchunks; cidx += 16) {
            s0 += partial[(long)cidx * nparams + p];
            s1 += partial[(long)(cidx + 4) * nparams + p];
            s2 += partial[(long)(cidx + 8) * nparams + p];
            s3 += partial[(long)(cidx + 12) * nparams + p];
        }
        for (; cidx < nchunks; cidx += 4) s0 += partial[(long)cidx * nparams + p];
    }
    sub[sl][pl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && p < nparams) {
        const float s = (sub[0][pl] + sub[1][pl]) + (sub[2][pl] + sub[3][pl]);
        grad[p] = (accumulate ? grad[p] : 0.0f) + scale * s + poison;
    }
}

// Epilogue of fused_step_kernel (round 5): ONE launch reduces the per-workgroup partials of BOTH parts of the step launch -- A: the collocation
// set (slot layout [wave][8]), B: the value-only side sets ([wave][FUSED_MAX_SETS][8]) -- in the order the two separate calls used to
// (grad = scaleA * sumA, then += scaleB * sumB: the same bits), writes the 1 + nsets loss sums, and, if `adam.theta` is set, applies the TF1 Adam
// rule of adam_tf1_kernel to the parameters in the same thread (pinn_wave2d_step with an optimizer state: a training step at world size 1 is
// repack + step launch + this; with a collective in between the gradient is left in `grad` and pinn_adam_step follows the all-reduce).
struct AdamEpilogue {
    float* theta;              // nullptr: no optimizer step
    float* m;
    float* v;
    float lr_t, beta1, beta2, eps;
};
struct StepPart {
    const float* partial;      // [nchunks][nparams]
    int nchunks;
    float scale;
    const float* loss_part;
    long nwaves;
};
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void reduce_step_kernel(StepPart A, StepPart B, int nparams, float* grad, int accumulate, int nterms_a, float* loss_a,
                                                          int nterms_b, int nsets, int slots_b, LossOuts loss_b, AdamEpilogue adam, const int* wflags,
                                                          int nflags) {
    __shared__ int bad_weights;
    if (threadIdx.x == 0) bad_weights = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < nflags; i += 256)
        if (wflags[i]) bad_weights = 1;
    __syncthreads();
    const float poison = bad_weights ? __builtin_nanf("") : 0.0f;
    const int grad_blocks = (int)gridDim.x - nsets - 1;
    if ((int)blockIdx.x >= grad_blocks) {
        const int k = (int)blockIdx.x - grad_blocks - 1;          // -1: the collocation set, 0..nsets-1: the side sets
        const float* lp = k < 0 ? A.loss_part : B.loss_part;
        const long nw = k < 0 ? A.nwaves : B.nwaves;
        const int slots = k < 0 ? 1 : slots_b, slot = k < 0 ? 0 : k, nterms = k < 0 ? nterms_a : nterms_b;
        float* out = k < 0 ? loss_a : loss_b.p[k];
        const int term = threadIdx.x >> 5, sub = threadIdx.x & 31;
        double s = 0.0;
        if (term < nterms)
            for (long w = sub; w < nw; w += 32) s += (double)lp[(w * slots + slot) * 8 + term];
        float v = (float)s;
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 8);
        v += __shfl_xor(v, 16);
        if (sub == 0 && term < nterms && out != nullptr) out[term] = v + poison;
        return;
    }
    __shared__ float sub[2][4][64];
    const int pl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const long p = (long)blockIdx.x * 64 + pl;
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        const StepPart& P = part ? B : A;
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
        if (p < nparams) {
            int cidx = sl;
            for (; cidx + 12 < P.nchunks; cidx += 16) {
                s0 += P.partial[(long)cidx * nparams + p];
                s1 += P.partial[(long)(cidx + 4) * nparams + p];
                s2 += P.partial[(long)(cidx + 8) * nparams + p];
                s3 += P.partial[(long)(cidx + 12) * nparams + p];
            }
            for (; cidx < P.nchunks; cidx += 4) s0 += P.partial[(long)cidx * nparams + p];
        }
        sub[part][sl][pl] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    if (sl == 0 && p < nparams) {
        const float sa = (sub[0][0][pl] + sub[0][1][pl]) + (sub[0][2][pl] + sub[0][3][pl]);
        const float sb = (sub[1][0][pl] + sub[1][1][pl]) + (sub[1][2][pl] + sub[1][3][pl]);
        float g = (accumulate ? grad[p] : 0.0f) + A.scale * sa + poison;      // what the collocation call's reduction wrote ...
        g = g + B.scale * sb + poison;                                         // ... and the side-set call's added
        grad[p] = g;
        if (adam.theta != nullptr) {
            const float mi = adam.beta1 * adam.m[p] + (1.0f - adam.beta1) * g;
            const float vi = adam.beta2 * adam.v[p] + (1.0f - adam.beta2) * g * g;
            adam.m[p] = mi;
            adam.v[p] = vi;
            adam.theta[p] -= adam.lr_t * mi / (__builtin_amdgcn_sqrtf(vi) + adam.eps);
        }
    }
}

// loss_terms[i] (+)= sum over waves of loss_part[w][i]   (one block of 256 threads: 32 lanes per term and pass, `slots` partial
// slots per wave: 8, or LOSS_SLOTS_3D for the 3-D heads; nterms <= slots)
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void reduce_loss_kernel(const float* loss_part, long nwaves, int nterms, float* loss_terms,
                                                          int accumulate, int slots, const int* wflags = nullptr, int nflags = 0) {
    // (fused 3-D path: the same NaN poisoning as reduce_grad_loss_kernel when a weight left the fused format)
    __shared__ int bad_weights;
    if (threadIdx.x == 0) bad_weights = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < nflags; i += 256)
        if (wflags[i]) bad_weights = 1;
    __syncthreads();
    const float poison = bad_weights ? __builtin_nanf("") : 0.0f;
    const int sub = threadIdx.x & 31;
    for (int term = threadIdx.x >> 5; term < slots; term += 8) {
        double s = 0.0;
        if (term < nterms)
            for (long w = sub; w < nwaves; w += 32) s += (double)loss_part[w * slots + term];
        float v = (float)s;
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 8);
        v += __shfl_xor(v, 16);
        if (sub == 0 && term < nterms) loss_terms[term] = (accumulate ? loss_terms[term] : 0.0f) + v + poison;
    }
}

// tf.train.AdamOptimizer (TF1 rule; INF:131-133): epsilon added to the UNCORRECTED sqrt(v); the bias
// correction lives in lr_t = lr * sqrt(1-beta2^t)/(1-beta1^t), computed by the caller in double.
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void adam_tf1_kernel(float* theta, float* m, float* v, const float* g, long n, float lr_t,
                                                       float beta1, float beta2, float eps) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    theta[i] -= lr_t * mi / (__builtin_amdgcn_sqrtf(vi) + eps);
}

}  // namespace pinn
