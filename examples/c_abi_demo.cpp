// Minimal host program on the C-ABI of libpinn_hip.so -- no PyTorch, no Python: HIP runtime for the device buffers, the library
// for everything else.  It is what a non-Python host (or the reference, through ctypes) sees.
//
//   hipcc -O2 -Iinclude examples/c_abi_demo.cpp -Lpinn_elastodynamics_amd/lib -lpinn_hip -Wl,-rpath,$PWD/pinn_elastodynamics_amd/lib -o build/c_abi_demo
//   build/c_abi_demo [n_points]
//
// Prints the seven residual mean squares of net_f_sig (INF:221-265,104-110) of a fresh Xavier 8x64 net on random collocation
// points, the gradient norm, and the loss after 20 Adam steps (TF1 rule, INF:131-133).  Every evaluation goes through
// pinn_wave2d_loss_grad_checked: the finite-gradient ladder (weight range of the fused format, fp16 range of the reverse pass) that the
// Python classes run around their calls, here as a library call; the last lines provoke both rungs on purpose.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "pinn_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define CHECK_PINN(x) do { int r_ = (x); if (r_ != 0) { std::fprintf(stderr, "pinn error %d (%s) at %s:%d\n", r_, pinn_error_string(r_), __FILE__, __LINE__); return 3; } } while (0)

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? std::atoll(argv[1]) : 200000;
    const int layers[] = {3, 64, 64, 64, 64, 64, 64, 64, 64, 7};
    const int n_layers = 10;
    const double lb[3] = {0, 0, 0}, ub[3] = {30, 30, 20};
    int nparams = 0;
    for (int l = 0; l + 1 < n_layers; ++l) nparams += layers[l] * layers[l + 1] + layers[l + 1];

    std::mt19937 gen(1111);
    std::vector<float> theta(nparams), x(n), y(n), t(n);
    {   // Xavier normal, zero biases (INF:141-156); flat order W0,b0,W1,b1,... with W row-major [in,out]
        int o = 0;
        for (int l = 0; l + 1 < n_layers; ++l) {
            std::normal_distribution<float> d(0.0f, std::sqrt(2.0f / (layers[l] + layers[l + 1])));
            for (int i = 0; i < layers[l] * layers[l + 1]; ++i) theta[o++] = d(gen);
            for (int i = 0; i < layers[l + 1]; ++i) theta[o++] = 0.0f;
        }
        std::uniform_real_distribution<float> u(0.0f, 1.0f);
        for (int64_t i = 0; i < n; ++i) { x[i] = 30 * u(gen); y[i] = 30 * u(gen); t[i] = 20 * u(gen); }
    }
    const size_t ws_bytes = pinn_workspace_bytes(layers, n_layers, n < (1 << 18) ? n : (1 << 18), PINN_PREC_F16X3);
    if (!ws_bytes) { std::fprintf(stderr, "no kernel variant for this net\n"); return 1; }

    float *d_theta, *d_m, *d_v, *d_x, *d_y, *d_t, *d_loss, *d_grad;
    void* d_ws;
    CHECK_HIP(hipMalloc(&d_theta, nparams * 4)); CHECK_HIP(hipMalloc(&d_m, nparams * 4)); CHECK_HIP(hipMalloc(&d_v, nparams * 4));
    CHECK_HIP(hipMalloc(&d_grad, nparams * 4)); CHECK_HIP(hipMalloc(&d_loss, 8 * 4));
    CHECK_HIP(hipMalloc(&d_x, n * 4)); CHECK_HIP(hipMalloc(&d_y, n * 4)); CHECK_HIP(hipMalloc(&d_t, n * 4));
    CHECK_HIP(hipMalloc(&d_ws, ws_bytes));          // hipMalloc is 256-byte aligned
    CHECK_HIP(hipMemcpy(d_theta, theta.data(), nparams * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemset(d_m, 0, nparams * 4)); CHECK_HIP(hipMemset(d_v, 0, nparams * 4));
    CHECK_HIP(hipMemcpy(d_x, x.data(), n * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_y, y.data(), n * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_t, t.data(), n * 4, hipMemcpyHostToDevice));

    float tw[7];
    for (float& w : tw) w = 1.0f / (float)n;        // loss = loss_f_uv + loss_f_s with mean squares (INF:104-110,119)
    float loss[8];
    std::vector<float> grad(nparams);
    pinn_range_state range = {0, 0, 0, 0};          // kept across the run: adjoint shift / two-kernel flag the ladder has settled on
    for (int step = 0; step <= 20; ++step) {
        CHECK_PINN(pinn_wave2d_loss_grad_checked(d_theta, layers, n_layers, d_x, d_y, d_t, n, lb, ub, 1, 2.5, 0.25, 1.0, 1, tw, d_loss, d_grad, PINN_PREC_F16X3,
                                                 d_ws, ws_bytes, nullptr, &range));
        if (step == 0 || step == 20) {
            CHECK_HIP(hipMemcpy(loss, d_loss, 7 * 4, hipMemcpyDeviceToHost));
            CHECK_HIP(hipMemcpy(grad.data(), d_grad, nparams * 4, hipMemcpyDeviceToHost));
            double total = 0, g2 = 0;
            for (int i = 0; i < 7; ++i) total += loss[i] / (double)n;
            for (float g : grad) g2 += (double)g * g;
            std::printf("step %2d  loss %.6e  |grad| %.6e  terms", step, total, std::sqrt(g2));
            for (int i = 0; i < 7; ++i) std::printf(" %.3e", loss[i] / (double)n);
            std::printf("\n");
        }
        if (step < 20) CHECK_PINN(pinn_adam_step(d_theta, d_m, d_v, d_grad, nparams, 1e-3, 0.9, 0.999, 1e-8, step + 1, nullptr));
    }
    CHECK_HIP(hipDeviceSynchronize());
    std::printf("ladder after training: shift %d two_kernel %d\n", range.adjoint_shift, range.two_kernel);
    {   // both rungs on purpose.  (a) output layer x 3000: residuals ~1e3 x their trained size, the fp16 reverse pass overflows -> adjoint shift
        CHECK_HIP(hipMemcpy(theta.data(), d_theta, nparams * 4, hipMemcpyDeviceToHost));
        std::vector<float> big(theta);
        const int last_w = nparams - (layers[n_layers - 2] * layers[n_layers - 1] + layers[n_layers - 1]);
        for (int i = last_w; i < nparams - layers[n_layers - 1]; ++i) big[i] *= 3000.0f;
        CHECK_HIP(hipMemcpy(d_theta, big.data(), nparams * 4, hipMemcpyHostToDevice));
        pinn_range_state r1 = {0, 0, 0, 0};
        CHECK_PINN(pinn_wave2d_loss_grad_checked(d_theta, layers, n_layers, d_x, d_y, d_t, n, lb, ub, 1, 2.5, 0.25, 1.0, 1, tw, d_loss, d_grad, PINN_PREC_F16X3,
                                                 d_ws, ws_bytes, nullptr, &r1));
        int finite = 0;
        float wmax = 0;
        CHECK_PINN(pinn_probe_ranges(d_theta, d_grad, nparams, d_ws, ws_bytes, nullptr, &finite, &wmax));
        std::printf("ladder overflow: shift %d two_kernel %d attempts %d finite %d\n", r1.adjoint_shift, r1.two_kernel, r1.attempts, finite);
        // (b) one hidden weight beyond the fused kernels' format (|w| <= %g): the fused path returns NaN throughout -> two-kernel path
        big = theta;
        big[layers[0] * layers[1] + layers[1] + 5] = 2100.0f;
        CHECK_HIP(hipMemcpy(d_theta, big.data(), nparams * 4, hipMemcpyHostToDevice));
        pinn_range_state r2 = {0, 0, 0, 0};
        CHECK_PINN(pinn_wave2d_loss_grad_checked(d_theta, layers, n_layers, d_x, d_y, d_t, n, lb, ub, 1, 2.5, 0.25, 1.0, 1, tw, d_loss, d_grad, PINN_PREC_F16X3,
                                                 d_ws, ws_bytes, nullptr, &r2));
        CHECK_PINN(pinn_probe_ranges(d_theta, d_grad, nparams, d_ws, ws_bytes, nullptr, &finite, &wmax));
        std::printf("ladder range: shift %d two_kernel %d attempts %d finite %d wmax %g limit %g\n", r2.adjoint_shift, r2.two_kernel, r2.attempts, finite, wmax,
                    pinn_fused_weight_limit());
    }
    std::printf("abi %d ok\n", pinn_abi_version());
    return 0;
}
