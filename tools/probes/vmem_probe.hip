// Probe (dev tool): vector-memory throughput of ONE compute unit on gfx950 for the fused kernels' request patterns.
//   hipcc --offload-arch=gfx950 -O3 -o vmem_probe vmem_probe.hip && ./vmem_probe
// One workgroup per CU (LDS-limited), W waves; every wave issues 1 KB requests (64 lanes x 16 B, buffer_load_dwordx4):
//   mode 0: reads of a 320 KB table shared by all workgroups (the weight fragments: L2 hits)
//   mode 1: reads of a private 256 KB region per wave (parked states: L2 misses, streaming)
//   mode 2: stores to a private region per wave
//   mode 3: LDS-DMA (buffer_load ... lds) of the shared table
// Prints bytes per clock per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void probe(const u32x4* shared_tab, u32x4* priv, long* out, int iters, int nwaves) {
    extern __shared__ char lds[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    u32x4 acc = {0, 0, 0, 0};
    const u32x4* tab = shared_tab + lane;
    u32x4* mine = priv + ((long)blockIdx.x * 8 + wave) * (256 * 1024 / 16) + lane;
    __syncthreads();
    const long t0 = __builtin_readcyclecounter();
    if (wave < nwaves) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int rec = ((it * 16 + k) * 5 + wave * 37) % 320;      // 1 KB records of the 320 KB table
                const int prec = (it * 16 + k) % 256;
                if (MODE == 0) { u32x4 v = __builtin_nontemporal_load(tab + rec * 64); acc ^= v; }
                else if (MODE == 1) { u32x4 v = mine[prec * 64]; acc ^= v; }
                else if (MODE == 2) { mine[prec * 64] = acc; acc[0] += k; }
            }
        }
    }
    const long t1 = __builtin_readcyclecounter();
    if (lane == 0) out[blockIdx.x * 8 + wave] = (t1 - t0) + ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345 ? 1 : 0);
}
template <int MODE>
static void run(const char* name, int nwaves, const u32x4* tab, u32x4* priv, long* d_out) {
    const int iters = 64;
    hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<MODE>), dim3(256), dim3(512), 100 * 1024, 0, tab, priv, d_out, iters, nwaves);
    hipDeviceSynchronize();
    std::vector<long> h(256 * 8);
    hipMemcpy(h.data(), d_out, h.size() * sizeof(long), hipMemcpyDeviceToHost);
    double cyc = 0;
    for (int b = 0; b < 256; ++b) { long m = 0; for (int w = 0; w < nwaves; ++w) m = h[b * 8 + w] > m ? h[b * 8 + w] : m; cyc += (double)m; }
    cyc /= 256.0;
    const double bytes = (double)nwaves * iters * 16 * 1024;
    printf("%-40s waves %d: %8.0f cycles, %6.1f B/clk/CU  (%.2f TB/s chip at 2.4 GHz)\n", name, nwaves, cyc, bytes / cyc, bytes / cyc * 256 * 2.4e9 / 1e12);
}
int main() {
    u32x4 *tab, *priv; long* d_out;
    hipMalloc(&tab, 320 * 1024);
    hipMalloc(&priv, 256L * 8 * 256 * 1024);
    hipMalloc(&d_out, 256 * 8 * sizeof(long));
    hipMemset(tab, 1, 320 * 1024);
    hipMemset(priv, 1, 256L * 8 * 256 * 1024);
    for (int nw : {1, 2, 4, 8}) run<0>("shared 320 KB table (L2 hits)", nw, tab, priv, d_out);
    for (int nw : {1, 2, 4, 8}) run<1>("private streaming reads", nw, tab, priv, d_out);
    for (int nw : {1, 2, 4, 8}) run<2>("private streaming stores", nw, tab, priv, d_out);
    return 0;
}
