// Probe (dev tool): what do no-return fp32 atomic adds into an L2-resident buffer cost on gfx950, against plain 16-byte stores and
// against the load + store round trip of private per-workgroup sums (what the fused kernels do today)?
//   hipcc --offload-arch=gfx950 -O3 -o atomic_probe atomic_probe.hip && ./atomic_probe
// 256 workgroups x 256 threads (4 waves = the weight-gradient waves of a fused workgroup).  Per "layer-step" a wave updates NREC records
// of 1 KB (64 lanes x 16 B); the workgroup's 4 waves own 4 x NREC records; 7 layers.  MODE 0: private sums, load + add + store (b128).
// MODE 1: shared sums per XCD (blockIdx % 8), 4 x buffer atomic add f32 per record.  MODE 2: private, store only (no load).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NREC>
__global__ __launch_bounds__(256) void probe(float* buf, int steps, float v) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int LAYERS = 7;
    const long per_owner = (long)LAYERS * 4 * NREC * 256;                                   // floats
    float* base = buf + (MODE == 1 ? (long)(blockIdx.x % 8) : (long)blockIdx.x) * per_owner;
    for (int s = 0; s < steps; ++s) {
        for (int l = 0; l < LAYERS; ++l) {
            float* rec = base + ((long)(l * 4 + wave) * NREC) * 256 + lane * 4;
#pragma unroll
            for (int r = 0; r < NREC; ++r) {
                f32x4* p = reinterpret_cast<f32x4*>(rec + r * 256);
                if (MODE == 0) {
                    f32x4 t = *p;
                    t += f32x4{v, v, v, v};
                    *p = t;
                } else if (MODE == 2) {
                    *p = f32x4{v, v, v, v};
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) __builtin_amdgcn_global_atomic_fadd_f32(rec + r * 256 + k, v);
                }
            }
            // a little ALU work between layers, like the MFMA phase (keeps the requests from being one dense burst)
            for (int k = 0; k < 64; ++k) v = __builtin_fmaf(v, 1.0000001f, 1e-9f);
        }
    }
    if (v == 123.0f) buf[0] = v;
}

template <int MODE, int NREC>
static void run(const char* name, float* buf, int steps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((probe<MODE, NREC>), dim3(256), dim3(256), 0, 0, buf, 2, 1e-6f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((probe<MODE, NREC>), dim3(256), dim3(256), 0, 0, buf, steps, 1e-6f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double recs = 256.0 * 4 * NREC * 7 * steps;
    printf("%-44s NREC=%2d: %8.3f ms for %d steps  -> %6.1f ns per workgroup-step, %6.2f G records/s (%5.2f TB/s of 1 KB records)\n", name, NREC, ms, steps,
           1e6 * ms / steps, recs / ms * 1e-6, recs * 1024 / ms * 1e-9);
}

int main() {
    float* buf;
    const size_t bytes = 256ull * 7 * 4 * 16 * 1024 + (1 << 20);
    hipMalloc(&buf, bytes);
    hipMemset(buf, 0, bytes);
    const int steps = 200;
    run<0, 9>("private sums: load + add + store", buf, steps);
    run<2, 9>("private sums: store only", buf, steps);
    run<1, 9>("shared per XCD: 4 atomic fadd per record", buf, steps);
    run<0, 16>("private sums: load + add + store", buf, steps);
    run<1, 16>("shared per XCD: 4 atomic fadd per record", buf, steps);
    hipFree(buf);
    return 0;
}
