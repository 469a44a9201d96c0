// Probe (dev tool): do v_fma_mixlo_f16 and v_mfma_f32_16x16x32_f16 keep fp16 subnormals on gfx950 (default kernel float mode)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void k(float* out, float x) {
    // lo = fp16(x - fp16(x)) by the mixed-precision FMA, x chosen so that lo is an fp16 subnormal
    uint32_t h, l;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(h) : "v"(x));
    asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=&v"(l) : "v"(h), "v"(x));
    const int lane = threadIdx.x & 63;
    if (lane == 0) { out[0] = __builtin_bit_cast(float, l); out[1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(l & 0xffff)); out[2] = x - (float)__builtin_bit_cast(_Float16, (uint16_t)(h & 0xffff)); }
    // MFMA with a subnormal A operand: A[i][k] = 2^-20 (subnormal in fp16) for k = 0, B[k][j] = 1024 -> D = 2^-10 if not flushed
    f16x8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (_Float16)0.0f; B[i] = (_Float16)0.0f; }
    if ((lane >> 4) == 0) { A[0] = __builtin_bit_cast(_Float16, (uint16_t)0x0010); B[0] = (_Float16)1024.0f; }   // 0x0010 = 16 * 2^-24 = 2^-20
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, c, 0, 0, 0);
    if (lane == 0) out[3] = c[0];
    // subnormal B operand
    if ((lane >> 4) == 0) { B[0] = __builtin_bit_cast(_Float16, (uint16_t)0x0010); A[0] = (_Float16)1024.0f; }
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, d, 0, 0, 0);
    if (lane == 0) out[4] = d[0];
}
int main() {
    float* d; hipMalloc(&d, 64);
    const float x = 0.0123456789f;   // fp16 ulp here is 2^-17: lo ~ 2^-18..2^-19 (subnormal: below 2^-14)
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, x);
    float h[5]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("lo bits as float %g ; lo (mixlo, fp16) = %.10g ; exact x - hi = %.10g\n", h[0], h[1], h[2]);
    printf("mfma subnormal A * 1024 = %.10g, subnormal B * 1024 = %.10g (expect %.10g)\n", h[3], h[4], 1.0 / 1024);
    return 0;
}
