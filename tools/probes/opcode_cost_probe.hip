// Probe (dev tool, round 4): ISSUE-COST TABLE of every opcode class the fused kernel's chain wave retires, on gfx950.
// For each opcode: NCH independent chains (1 = every instruction reads its predecessor's result ... 8 = eight instructions apart), one wave per
// SIMD (256 threads per workgroup) and two (512).  Prints shader-clock cycles per instruction; tools/opcode_table.py turns the log into
// profiles/r04_opcode_issue_costs.md and multiplies it with the ISA histogram of a forward / reverse layer of the chain wave.
//   hipcc --offload-arch=gfx950 -O3 -o opcode_cost_probe opcode_cost_probe.hip && ./opcode_cost_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
enum { K_FMA, K_MUL, K_ADD, K_MOV, K_MIXF32, K_MIXLO, K_MIXLOHI, K_CVTPK, K_EXP, K_RCP, K_PKMUL, K_PKFMA, K_PERM, K_N };
static const char* NAMES[K_N] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_mov_b32", "v_fma_mix_f32", "v_fma_mixlo_f16", "v_fma_mixlo+mixhi_same_reg",
                                 "v_cvt_pk_f16_f32", "v_exp_f32", "v_rcp_f32", "v_pk_mul_f32", "v_pk_fma_f32", "v_perm_b32"};
template <int KIND>
__device__ __forceinline__ void op(float& a, f32x2& p, float c, f32x2 cc) {
    if (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(c));
    else if (KIND == K_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(c));
    else if (KIND == K_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(c));
    else if (KIND == K_MOV) asm volatile("v_mov_b32 %0, %0" : "+v"(a));
    else if (KIND == K_MIXF32) asm volatile("v_fma_mix_f32 %0, %0, %1, %1 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "+v"(a) : "v"(c));
    else if (KIND == K_MIXLO) asm volatile("v_fma_mixlo_f16 %0, %0, %1, %1 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(a) : "v"(c));
    else if (KIND == K_MIXLOHI) asm volatile("v_fma_mixlo_f16 %0, %0, %1, %1 op_sel:[0,0,0] op_sel_hi:[0,0,0]\n\tv_fma_mixhi_f16 %0, %1, %1, %1 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(a) : "v"(c));
    else if (KIND == K_CVTPK) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a) : "v"(c));
    else if (KIND == K_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(a));
    else if (KIND == K_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a));
    else if (KIND == K_PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(cc));
    else if (KIND == K_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p) : "v"(cc));
    else if (KIND == K_PERM) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a) : "v"(c));
}
template <int KIND, int NCH>
__global__ __launch_bounds__(512) void probe(long* out, int iters, float seed) {
    float v[NCH];
    f32x2 p[NCH];
    const f32x2 cc = {seed, seed};
    for (int i = 0; i < NCH; ++i) { v[i] = seed * (i + 1); p[i] = f32x2{seed, seed * i}; }
    __syncthreads();
    const long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 64; ++k) op<KIND>(v[k % NCH], p[k % NCH], seed, cc);
    }
    const long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < NCH; ++i) s += v[i] + p[i][0] + p[i][1];
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = (t1 - t0) + (s == 12345.f ? 1 : 0);
}
static double collect(long* d_out, int waves, double per) {
    hipDeviceSynchronize();
    std::vector<long> h(256 * 8);
    hipMemcpy(h.data(), d_out, h.size() * sizeof(long), hipMemcpyDeviceToHost);
    double t = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) t += (double)h[b * 8 + w];
    return t / (256.0 * waves * per);
}
template <int KIND, int NCH>
static void run(long* d_out) {
    const int iters = 300;
    for (int waves : {4, 8}) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<KIND, NCH>), dim3(256), dim3(64 * waves), 0, 0, d_out, iters, 1.0f);
        printf("VALU %-28s chains %d waves_per_simd %d cycles %.2f\n", NAMES[KIND], NCH, waves / 4, collect(d_out, waves, (double)iters * 64 * (KIND == K_MIXLOHI ? 2 : 1)));
    }
}
template <int KIND> static void all(long* d) { run<KIND, 1>(d); run<KIND, 2>(d); run<KIND, 4>(d); run<KIND, 8>(d); }

// ---- MFMA: NACC independent accumulators (1 = every MFMA accumulates onto its predecessor), NV vector instructions between two MFMAs
template <int NACC, int NV>
__global__ __launch_bounds__(512) void probe_mfma(long* out, int iters, float seed) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    f32x4 acc[NACC];
    float v[4] = {seed, seed, seed, seed};
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{seed, 0.f, 0.f, 0.f};
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed * 0.01f); b[i] = (_Float16)(seed * 0.02f); }
    __syncthreads();
    const long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[k % NACC]) : "v"(a), "v"(b));
#pragma unroll
            for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[j % 4]) : "v"(seed));
        }
    }
    const long t1 = __builtin_readcyclecounter();
    float s = v[0] + v[1] + v[2] + v[3];
    for (int i = 0; i < NACC; ++i) s += acc[i][0];
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = (t1 - t0) + (s == 12345.f ? 1 : 0);
}
template <int NACC, int NV>
static void run_mfma(long* d_out) {
    const int iters = 200;
    for (int waves : {4, 8}) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe_mfma<NACC, NV>), dim3(256), dim3(64 * waves), 0, 0, d_out, iters, 1.0f);
        printf("MFMA v_mfma_f32_16x16x32_f16 accumulators %d valu_behind_each %d waves_per_simd %d cycles_per_group %.2f\n", NACC, NV, waves / 4,
               collect(d_out, waves, (double)iters * 32));
    }
}

// ---- which vector opcodes issue in the shadow of an MFMA?  group = 1 MFMA (16 rotating accumulators) + NV instructions of KIND (8 chains)
template <int KIND, int NV>
__global__ __launch_bounds__(512) void probe_mfma_x(long* out, int iters, float seed) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    f32x4 acc[16];
    float v[8];
    f32x2 p[8];
    const f32x2 cc = {seed, seed};
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{seed, 0.f, 0.f, 0.f};
    for (int i = 0; i < 8; ++i) { v[i] = seed * (i + 1); p[i] = f32x2{seed, seed * i}; }
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed * 0.01f); b[i] = (_Float16)(seed * 0.02f); }
    __syncthreads();
    const long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[k % 16]) : "v"(a), "v"(b));
#pragma unroll
            for (int j = 0; j < NV; ++j) op<KIND>(v[(k * NV + j) % 8], p[(k * NV + j) % 8], seed, cc);
        }
    }
    const long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i] + p[i][0] + p[i][1];
    for (int i = 0; i < 16; ++i) s += acc[i][0];
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = (t1 - t0) + (s == 12345.f ? 1 : 0);
}
template <int KIND, int NV>
static void run_mfma_x(long* d_out) {
    const int iters = 200;
    for (int waves : {4, 8}) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe_mfma_x<KIND, NV>), dim3(256), dim3(64 * waves), 0, 0, d_out, iters, 1.0f);
        printf("MFMAX %-28s per_mfma %d waves_per_simd %d cycles_per_group %.2f\n", NAMES[KIND], NV * (KIND == K_MIXLOHI ? 2 : 1), waves / 4, collect(d_out, waves, (double)iters * 32));
    }
}
template <int KIND> static void all_x(long* d) { run_mfma_x<KIND, 1>(d); run_mfma_x<KIND, 2>(d); run_mfma_x<KIND, 3>(d); }

// ---- LDS: reads / writes of the kinds the chain and weight-gradient waves issue, NQ in flight before the wait
enum { L_READ_B64, L_READ_B128, L_WRITE_B128, L_READ2ST64, L_READ_TR, L_N };
static const char* LNAMES[L_N] = {"ds_read_b64", "ds_read_b128", "ds_write_b128", "ds_read2st64_b64", "ds_read_b64_tr_b16"};
template <int KIND, int NQ>
__global__ __launch_bounds__(512) void probe_lds(long* out, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[64 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = lane; i < 8192 / 4; i += 64) reinterpret_cast<unsigned*>(lds + wave * 8192)[i] = i;
    __syncthreads();
    u32x4 acc = {1u, 2u, 3u, 4u};
    const unsigned addr = (unsigned)(size_t)(lds + wave * 8192 + lane * 16);
    const long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        u32x4 r[NQ];
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            if (KIND == L_READ_B64) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(*reinterpret_cast<u32x2*>(&r[k])) : "v"(addr), "i"(k * 1024));
            else if (KIND == L_READ_B128) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[k]) : "v"(addr), "i"(k * 1024));
            else if (KIND == L_WRITE_B128) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(acc), "i"(k * 1024));
            else if (KIND == L_READ2ST64) asm volatile("ds_read2st64_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(r[k]) : "v"(addr), "i"(k), "i"(k + 8));
            else asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(*reinterpret_cast<u32x2*>(&r[k])) : "v"(addr & ~8u), "i"(k * 1024));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (KIND != L_WRITE_B128) {
#pragma unroll
            for (int k = 0; k < NQ; ++k) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(acc[0]) : "v"(r[k][0]));
        }
    }
    const long t1 = __builtin_readcyclecounter();
    if (lane == 0) out[blockIdx.x * 8 + wave] = (t1 - t0) + (acc[0] == 12345u ? 1 : 0);
}
template <int KIND, int NQ>
static void run_lds(long* d_out) {
    const int iters = 2000;
    for (int waves : {4, 8}) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe_lds<KIND, NQ>), dim3(256), dim3(64 * waves), 0, 0, d_out, iters);
        printf("LDS %-20s in_flight %d waves_per_cu %d cycles %.2f\n", LNAMES[KIND], NQ, waves, collect(d_out, waves, (double)iters * NQ));
    }
}
template <int KIND> static void all_lds(long* d) { run_lds<KIND, 1>(d); run_lds<KIND, 4>(d); run_lds<KIND, 8>(d); }

int main() {
    long* d;
    hipMalloc(&d, 256 * 8 * sizeof(long));
    all<K_FMA>(d); all<K_MUL>(d); all<K_ADD>(d); all<K_MOV>(d); all<K_MIXF32>(d); all<K_MIXLO>(d); all<K_MIXLOHI>(d); all<K_CVTPK>(d);
    all<K_EXP>(d); all<K_RCP>(d); all<K_PKMUL>(d); all<K_PKFMA>(d); all<K_PERM>(d);
    run_mfma<1, 0>(d); run_mfma<2, 0>(d); run_mfma<4, 0>(d); run_mfma<16, 0>(d);
    run_mfma<16, 1>(d); run_mfma<16, 2>(d); run_mfma<16, 3>(d); run_mfma<16, 4>(d); run_mfma<4, 2>(d); run_mfma<4, 3>(d);
    all_x<K_FMA>(d); all_x<K_MUL>(d); all_x<K_MOV>(d); all_x<K_MIXF32>(d); all_x<K_MIXLO>(d); all_x<K_MIXLOHI>(d); all_x<K_CVTPK>(d); all_x<K_EXP>(d);
    all_x<K_RCP>(d); all_x<K_PKMUL>(d); all_x<K_PKFMA>(d); all_x<K_PERM>(d);
    all_lds<L_READ_B64>(d); all_lds<L_READ_B128>(d); all_lds<L_WRITE_B128>(d); all_lds<L_READ2ST64>(d); all_lds<L_READ_TR>(d);
    return 0;
}
