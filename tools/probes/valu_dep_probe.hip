// Probe (dev tool): issue cost of DEPENDENT vector instructions on gfx950: NCH independent chains, each instruction reads the result of
// the previous instruction of its chain.  One wave per SIMD (256 threads) and two (512).  Prints shader cycles per instruction.
//   hipcc --offload-arch=gfx950 -O3 -o valu_dep_probe valu_dep_probe.hip && ./valu_dep_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
enum { K_FMA = 0, K_MIX = 1, K_CVT = 2, K_MUL = 3, K_EXP = 4, K_MIXLOHI = 5, K_PKMUL = 6 };
template <int KIND>
__device__ __forceinline__ void op(float& a, float c) {
    if (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(c));
    else if (KIND == K_MIX) asm volatile("v_fma_mix_f32 %0, %0, %1, %1 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "+v"(a) : "v"(c));
    else if (KIND == K_CVT) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a) : "v"(c));
    else if (KIND == K_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(c));
    else if (KIND == K_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(a));
    else if (KIND == K_MIXLOHI) asm volatile("v_fma_mixlo_f16 %0, %0, %1, %1 op_sel:[0,0,0] op_sel_hi:[0,0,0]\n\tv_fma_mixhi_f16 %0, %1, %1, %1 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(a) : "v"(c));
    else { typedef float f2 __attribute__((ext_vector_type(2))); f2 t = {a, a}; f2 cc = {c, c}; asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(t) : "v"(cc)); a = t[0]; }
}
template <int KIND, int NCH>
__global__ __launch_bounds__(512) void probe(long* out, int iters, float seed) {
    float v[NCH];
    for (int i = 0; i < NCH; ++i) v[i] = seed * (i + 1);
    __syncthreads();
    const long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 64; ++k) op<KIND>(v[k % NCH], seed);
    }
    const long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < NCH; ++i) s += v[i];
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = (t1 - t0) + (s == 12345.f ? 1 : 0);
}
template <int KIND, int NCH>
static void run(const char* name, long* d_out) {
    const int iters = 500;
    for (int waves : {4, 8}) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<KIND, NCH>), dim3(256), dim3(64 * waves), 0, 0, d_out, iters, 1.0f);
        hipDeviceSynchronize();
        std::vector<long> h(256 * 8);
        hipMemcpy(h.data(), d_out, h.size() * sizeof(long), hipMemcpyDeviceToHost);
        double t = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) t += (double)h[b * 8 + w];
        t /= 256.0 * waves * iters * 64 * (KIND == K_MIXLOHI ? 2 : 1);
        printf("%-22s chains %2d  waves/SIMD %d : %6.2f cycles per instruction\n", name, NCH, waves / 4, t);
    }
}
#define ALLCH(KIND, name) run<KIND, 1>(name, d); run<KIND, 2>(name, d); run<KIND, 4>(name, d); run<KIND, 8>(name, d); run<KIND, 16>(name, d);
int main() {
    long* d;
    hipMalloc(&d, 256 * 8 * sizeof(long));
    ALLCH(K_FMA, "v_fma_f32");
    ALLCH(K_MUL, "v_mul_f32");
    ALLCH(K_MIX, "v_fma_mix_f32");
    ALLCH(K_CVT, "v_cvt_pk_f16_f32");
    ALLCH(K_MIXLOHI, "v_fma_mixlo+hi_f16");
    ALLCH(K_EXP, "v_exp_f32");
    ALLCH(K_PKMUL, "v_pk_mul_f32");
    return 0;
}
