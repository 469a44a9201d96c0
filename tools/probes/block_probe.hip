// Probe (dev tool, round 4): the fused kernel's OWN forward block step (Fused<>::fwd_block_staged: 24 MFMAs + the vector part of a block) in
// isolation -- registers only, no memory traffic, no barriers -- one wave per SIMD and two.  Cycles per block step.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPINN_X_STAGE -I pinn_elastodynamics_amd/csrc -o block_probe tools/probes/block_probe.hip
#include "pinn_fused.hpp"
#include <cstdio>
#include <vector>
using namespace pinn;
typedef Fused<OpF16, 3, 64, 8, 4, false> F;
template <int MODE>
__global__ __launch_bounds__(512) void probe(long* out, int iters, float seed) {
    f32x4 acca[4], accb[4], accx[4];
    u32x4 B[4][1][2][2], B2[4][1][2][2], A[2][2];
    for (int s = 0; s < 4; ++s) {
        acca[s] = f32x4{seed * 0.01f * (s + 1), seed * 0.02f, -seed * 0.015f, seed * 0.005f};
        accb[s] = acca[s];
        accx[s] = acca[s] * 0.5f;
        for (int kk = 0; kk < 2; ++kk)
            for (int p = 0; p < 2; ++p) B[s][0][kk][p] = B2[s][0][kk][p] = u32x4{0x2e662e66u + threadIdx.x, 0x2a002a00u, 0x2c002c00u, 0x29002900u};
    }
    for (int kk = 0; kk < 2; ++kk)
        for (int p = 0; p < 2; ++p) A[kk][p] = u32x4{0x28002800u, 0x24002400u + threadIdx.x, 0x26002600u, 0x22002200u};
    __syncthreads();
    const long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        // two block steps: vector part on acca while the MFMAs fill accb, then the other way round (as fwd_step alternates)
        if (MODE == 0) {
            F::fwd_block_staged<0, 24>(acca, B2, A, B, accb);
            F::fwd_block_staged<1, 24>(accb, B2, A, B, acca);
        } else if (MODE == 3) {       // as 0, but the vector part reads accumulators no MFMA of the loop writes (accx), and writes fragments no MFMA reads
            F::fwd_block_staged<0, 24>(accx, B2, A, B, accb);
            F::fwd_block_staged<1, 24>(accx, B2, A, B, acca);
        } else if (MODE == 1) {       // the MFMAs alone
            F::fwd_mfs<0, 24, 2>(A, B, accb);
            F::fwd_mfs<0, 24, 2>(A, B, acca);
        } else {                      // the vector part alone (compiler's order)
            F::fwd_valu<0>(acca, B2);
            F::fwd_valu<1>(accb, B2);
        }
        __builtin_amdgcn_sched_barrier(0);
        for (int s = 0; s < 4; ++s) { acca[s] = acca[s] * 1e-3f + 0.1f; accb[s] = accb[s] * 1e-3f + 0.1f; }
        if (MODE == 3) for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(accx[s]));
    }
    const long t1 = __builtin_readcyclecounter();
    float sum = 0.f;
    for (int s = 0; s < 4; ++s) sum += acca[s][0] + accb[s][1] + accx[s][2] + __builtin_bit_cast(float, B2[s][0][0][0][0]) + __builtin_bit_cast(float, B2[s][0][0][1][1]) + __builtin_bit_cast(float, B2[s][0][1][0][0]);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = (t1 - t0) + (sum == 12345.f ? 1 : 0);
}
template <int MODE>
static void run(const char* name, long* d_out) {
    const int iters = 2000;
    for (int waves : {4, 8}) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<MODE>), dim3(256), dim3(64 * waves), 0, 0, d_out, iters, 1.0f);
        (void)hipDeviceSynchronize();
        std::vector<long> h(256 * 8);
        (void)hipMemcpy(h.data(), d_out, h.size() * sizeof(long), hipMemcpyDeviceToHost);
        double t = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) t += (double)h[b * 8 + w];
        printf("%-44s waves/SIMD %d : %7.1f cycles per block step (24 MFMAs + one block's vector part)\n", name, waves / 4, t / (256.0 * waves * iters * 2));
    }
}
int main() {
    long* d;
    (void)hipMalloc(&d, 256 * 8 * sizeof(long));
    run<0>("staged block step (V V M order)", d);
    run<1>("24 MFMAs alone", d);
    run<2>("vector part alone", d);
    run<3>("staged block step, vector part on registers the MFMAs do not touch", d);
    return 0;
}
