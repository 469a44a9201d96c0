// Probe (dev tool): how do the matrix pipe and the vector ALU of ONE SIMD share time on gfx950?
//   hipcc --offload-arch=gfx950 -O3 -o coexec_probe coexec_probe.hip && ./coexec_probe
// One workgroup per CU (LDS-limited), 4 or 8 waves (1 or 2 per SIMD).  Every wave runs `iters` times a body made of
//   NM MFMAs (16x16x32 f16, 8 independent accumulators) and NV vector instructions of a chosen kind,
// either in two bursts (MFMAs then VALU) or finely interleaved (1 MFMA : NV/NM VALU).  In the two-role mode waves 0-3 issue
// only the MFMAs and waves 4-7 (same SIMDs) only the vector instructions.  Prints shader cycles per body (s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

enum { V_FMA = 0, V_PKFMA = 1, V_MIX = 2, V_TRANS = 3, V_CVT = 4 };

template <int KIND>
__device__ __forceinline__ void valu1(float& a, float& b, float c) {
    if (KIND == V_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(c));
    else if (KIND == V_PKFMA) { typedef float f2 __attribute__((ext_vector_type(2))); f2 t = {a, b}; f2 cc = {c, c};
        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(t) : "v"(cc)); a = t[0]; b = t[1]; }
    else if (KIND == V_MIX) asm volatile("v_fma_mix_f32 %0, %0, %1, %1 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "+v"(a) : "v"(c));
    else if (KIND == V_TRANS) asm volatile("v_exp_f32 %0, %0" : "+v"(a));
    else asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a) : "v"(c));
}

// MODE 0: MFMA burst then VALU burst (all waves); 1: interleaved; 2: roles (waves<4 MFMA only, waves>=4 VALU only);
// 3: MFMA only; 4: VALU only; 5: burst, VALU depends on the MFMA results (reads acc)
template <int MODE, int KIND, int NM, int NV>
__global__ __launch_bounds__(512) void probe(long* out, int iters, float seed) {
    extern __shared__ char lds[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    f32x4 acc[8];
    f16x8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (_Float16)(seed + i); B[i] = (_Float16)(seed - i); }
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{seed, seed, seed, seed};
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed * i;
    const bool do_m = MODE == 0 || MODE == 1 || MODE == 3 || MODE == 5 || (MODE == 2 && wave < 4);
    const bool do_v = MODE == 0 || MODE == 1 || MODE == 4 || MODE == 5 || (MODE == 2 && wave >= 4);
    __syncthreads();
    const long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
            constexpr int R = NV / NM;
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(A), "v"(B));
#pragma unroll
                for (int k = 0; k < R; ++k) valu1<KIND>(v[(m * R + k) & 15], v[(m * R + k + 8) & 15], seed);
            }
        } else {
            if (do_m) {
#pragma unroll
                for (int m = 0; m < NM; ++m) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(A), "v"(B));
            }
            if (MODE == 5) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("s_nop 0\n\tv_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(acc[i][0]));
            }
            if (do_v) {
#pragma unroll
                for (int k = 0; k < NV; ++k) valu1<KIND>(v[k & 15], v[(k + 8) & 15], seed);
            }
        }
    }
    const long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += v[i];
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = (t1 - t0) + (s == 12345.f ? 1 : 0);
}

template <int MODE, int KIND, int NM, int NV>
static void run(const char* name, int waves, long* d_out) {
    const int iters = 2000;
    hipFuncSetAttribute((const void*)probe<MODE, KIND, NM, NV>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<MODE, KIND, NM, NV>), dim3(256), dim3(64 * waves), 100 * 1024, 0, d_out, iters, 1.0f);
    hipDeviceSynchronize();
    std::vector<long> h(256 * 8);
    hipMemcpy(h.data(), d_out, h.size() * sizeof(long), hipMemcpyDeviceToHost);
    double lo = 0, hi = 0;
    for (int b = 0; b < 256; ++b) {
        for (int w = 0; w < waves; ++w) (w < 4 ? lo : hi) += (double)h[b * 8 + w];
    }
    lo /= 256.0 * 4 * iters;
    hi /= 256.0 * 4 * iters;
    printf("%-44s waves/SIMD %d  NM %3d NV %3d : cycles/body  waves0-3 %8.1f", name, waves / 4, NM, NV, lo);
    if (waves == 8) printf("   waves4-7 %8.1f", hi);
    printf("\n");
}

#define RUNK(MODE, KIND, NM, NV, label) run<MODE, KIND, NM, NV>(label, 4, d_out); run<MODE, KIND, NM, NV>(label, 8, d_out);

int main() {
    long* d_out;
    hipMalloc(&d_out, 256 * 8 * sizeof(long));
    RUNK(3, V_FMA, 24, 96, "mfma only");
    RUNK(4, V_FMA, 24, 96, "v_fma_f32 only");
    RUNK(4, V_PKFMA, 24, 96, "v_pk_fma_f32 only");
    RUNK(4, V_MIX, 24, 96, "v_fma_mix_f32 only");
    RUNK(4, V_TRANS, 24, 96, "v_exp_f32 only");
    RUNK(4, V_CVT, 24, 96, "v_cvt_pk_f16_f32 only");
    RUNK(0, V_FMA, 24, 96, "burst mfma ; burst v_fma (independent)");
    RUNK(5, V_FMA, 24, 96, "burst mfma ; v_fma reading the results");
    RUNK(1, V_FMA, 24, 96, "interleaved 1 mfma : 4 v_fma");
    RUNK(1, V_FMA, 24, 72, "interleaved 1 mfma : 3 v_fma");
    RUNK(1, V_FMA, 24, 48, "interleaved 1 mfma : 2 v_fma");
    RUNK(1, V_PKFMA, 24, 48, "interleaved 1 mfma : 2 v_pk_fma");
    RUNK(1, V_MIX, 24, 48, "interleaved 1 mfma : 2 v_fma_mix");
    RUNK(1, V_TRANS, 24, 24, "interleaved 1 mfma : 1 v_exp");
    run<2, V_FMA, 24, 96>("roles: mfma waves 0-3 | v_fma waves 4-7", 8, d_out);
    run<2, V_FMA, 24, 192>("roles: mfma waves 0-3 | v_fma waves 4-7", 8, d_out);
    run<2, V_PKFMA, 24, 96>("roles: mfma | v_pk_fma", 8, d_out);
    run<2, V_MIX, 24, 96>("roles: mfma | v_fma_mix", 8, d_out);
    run<2, V_TRANS, 24, 48>("roles: mfma | v_exp", 8, d_out);
    return 0;
}
