// Probe: LDS addressing of buffer_load ... lds (raw_buffer_load_lds builtin) on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
__global__ void probe(const unsigned* src, unsigned* out, int inst_case) {
    __shared__ __attribute__((aligned(16))) unsigned lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = 0xdead0000u + i;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1 << 20, 0x00020000);
    const unsigned lane = threadIdx.x;
    // voffset = lane*16, soffset = 4096 bytes, inst offset variants
    if (inst_case == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(lds + 256), 16, lane * 16, 4096, 0, 0);
    if (inst_case == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(lds + 256), 16, lane * 16, 4096, 64, 0);
    if (inst_case == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(lds + 256), 16, (63 - lane) * 16, 0, 0, 0);
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) out[i] = lds[i];
}
int main() {
    unsigned *dsrc, *dout;
    hipMalloc(&dsrc, 1 << 20); hipMalloc(&dout, 8192);
    std::vector<unsigned> src(1 << 18), out(2048);
    for (size_t i = 0; i < src.size(); ++i) src[i] = (unsigned)i;     // dword index
    hipMemcpy(dsrc, src.data(), 1 << 20, hipMemcpyHostToDevice);
    for (int c = 0; c < 3; ++c) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dsrc, dout, c);
        hipMemcpy(out.data(), dout, 8192, hipMemcpyDeviceToHost);
        printf("case %d: first changed lds dword:", c);
        int first = -1, last = -1;
        for (int i = 0; i < 2048; ++i) if (out[i] != 0xdead0000u + i) { if (first < 0) first = i; last = i; }
        printf(" %d last %d; lds[first..+8] =", first, last);
        for (int i = first; i < first + 8 && first >= 0; ++i) printf(" %u", out[i]);
        printf(" ... lds[last-3..] =");
        for (int i = last - 3; i <= last && first >= 0; ++i) printf(" %u", out[i]);
        printf("\n");
    }
    return 0;
}
