// Probe (dev tool): where do the waves of small workgroups land on gfx950?
//   hipcc --offload-arch=gfx950 -O3 -o place_probe place_probe.hip && ./place_probe
// 512 workgroups of 4 waves with 67 KB of LDS each (two per CU by the LDS limit): per wave HW_ID (SIMD, CU, SE), XCC_ID and start time.
// Question: do the four waves of a workgroup sit on four different SIMDs, and do the two workgroups of a CU use the same wave -> SIMD map?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void probe(unsigned* out, int spin) {
    extern __shared__ char lds[];
    const int wave = threadIdx.x >> 6;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const long t0 = __builtin_readcyclecounter();
    lds[threadIdx.x] = (char)hw;
    long t = t0;
    while (t - t0 < spin) t = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + wave) * 4 + 0] = hw;
        out[(blockIdx.x * 4 + wave) * 4 + 1] = xcc;
        out[(blockIdx.x * 4 + wave) * 4 + 2] = (unsigned)(t0 & 0xffffffffu);
        out[(blockIdx.x * 4 + wave) * 4 + 3] = lds[threadIdx.x ^ 1];
    }
}
int main() {
    const int G = 512;
    unsigned* d;
    hipMalloc(&d, G * 16 * sizeof(unsigned));
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 67 * 1024);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(G), dim3(256), 67 * 1024, 0, d, 2000000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(G * 16);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cus;
    int distinct4 = 0, same_map = 0, pairs = 0;
    for (int b = 0; b < G; ++b) {
        unsigned hw = h[b * 16], xcc = h[b * 16 + 1] & 0xf;
        unsigned key = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf);
        cus[key].push_back(b);
        unsigned m = 0;
        for (int w = 0; w < 4; ++w) m |= 1u << ((h[(b * 4 + w) * 4] >> 4) & 3);
        distinct4 += m == 0xf;
    }
    std::map<size_t, int> hist;
    for (auto& kv : cus) {
        hist[kv.second.size()]++;
        if (kv.second.size() == 2) {
            ++pairs;
            bool same = true;
            for (int w = 0; w < 4; ++w) same &= ((h[(kv.second[0] * 4 + w) * 4] >> 4) & 3) == ((h[(kv.second[1] * 4 + w) * 4] >> 4) & 3);
            same_map += same;
        }
    }
    printf("workgroups %d, distinct CUs %zu; workgroups per CU histogram:", G, cus.size());
    for (auto& kv : hist) printf("  %zu wg: %d CUs", kv.first, kv.second);
    printf("\nworkgroups whose 4 waves sit on 4 different SIMDs: %d of %d\n", distinct4, G);
    printf("CUs with two workgroups: %d, of which both use the same wave->SIMD map: %d\n", pairs, same_map);
    for (int b = 0; b < 6; ++b) {
        printf("wg %3d: xcc %u se %u cu %2u  simd of waves:", b, h[b * 16 + 1] & 0xf, (h[b * 16] >> 13) & 7, (h[b * 16] >> 8) & 0xf);
        for (int w = 0; w < 4; ++w) printf(" %u", (h[(b * 4 + w) * 4] >> 4) & 3);
        printf("   t0 %u\n", h[b * 16 + 2]);
    }
    // which workgroup indices share a CU?
    int shown = 0;
    for (auto& kv : cus) if (kv.second.size() == 2 && shown++ < 6) printf("CU key %06x: workgroups %d and %d\n", kv.first, kv.second[0], kv.second[1]);
    return 0;
}
