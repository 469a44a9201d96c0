// Probe: semantics of ds_read_b64_tr_b16 on gfx950 (used to design the LDS hand-off of the fused kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16 lds_v4i16;
__global__ void probe(const int* lane_off_bytes, short* out) {
    __shared__ __attribute__((aligned(16))) short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    char* p = reinterpret_cast<char*>(lds) + lane_off_bytes[l];
    v4i16 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)p);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
    int *doff; short* dout;
    hipMalloc(&doff, 64 * 4); hipMalloc(&dout, 64 * 4 * 2);
    std::vector<int> off(64); std::vector<short> out(256);
    // test 1: lane-linear addresses (lane i -> 8*i bytes)
    for (int t = 0; t < 2; ++t) {
        for (int l = 0; l < 64; ++l) off[l] = t == 0 ? 8 * l : ((l >> 4) * 16 * 136 + ((l & 15) >> 2) * 136 + (l & 3) * 8);
        hipMemcpy(doff, off.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, doff, dout);
        hipMemcpy(out.data(), dout, 512, hipMemcpyDeviceToHost);
        printf("test %d\n", t);
        for (int l = 0; l < 64; ++l) { printf("lane %2d off %5d:", l, off[l]); for (int j = 0; j < 4; ++j) printf(" %5d", out[l * 4 + j]); printf("\n"); }
    }
    return 0;
}
