// Exact-fp32 mode (PINN_PREC_FP32): the same algorithm -- forward tangents, residual head, one reverse pass, weight gradient --
// in plain fp32 FMA arithmetic: no matrix pipe, no 16-bit operand anywhere.  It is what the reference's own arithmetic is (TF1
// float32, INF:71-92) and exists as the on-device third leg of the parity tests (float64 oracle / f16x3 product / fp32 device) and
// for users who want the reference's precision bit-class regardless of speed.  One thread per point, every per-point intermediate
// kept in the workspace as [layer][stream][feature][point] fp32 (point fastest: coalesced); a second kernel with one thread per
// parameter contracts state and adjoint over the points.  Slow by design (~100x the f16x3 path): tests and spot checks only.
//
// Heads: the wave residual head (INF:221-265; 4 streams), the value-only data head (INF:111-118; 1 stream) and the forward-only
// fields head.  The plate's second-order stream and the 3-D extension are not offered in this mode.
#pragma once
#include "pinn_device.hpp"

namespace pinn {

constexpr int FP32_MAX_NS = 4;

struct Fp32Args {
    NetDesc net;
    const float* params;       // flat fp32 parameters, reference order
    const float* x;
    const float* y;
    const float* t;
    long n;                    // points of the call
    long p0;                   // first point of this workspace pass
    long m;                    // points of this pass
    float sx[3], ox[3];
    float c1, c2, G, rho;
    float tw[8];
    const float* targets;      // data head: [nout][n] or nullptr
    float* S;                  // [nl+1][ns][hr][m]: S_0 = inputs and tangent seeds (first din rows), S_l = state behind hidden layer l
    float* Z;                  // [nl+1][ns][hr][m]: Z_l = adjoint of the pre-activations of weight layer l (Z_nl: nout rows)
    float* fsq;                // [8][m]: squared residuals of the pass
    float* fields_out;         // fields head: [ns][nout][n]
    int ns;                    // streams: 4 (value, x, y, t) or 1
    int hr;                    // row stride of S / Z: max(h, 16)
    int head;                  // HEAD_WAVE, HEAD_DATA, HEAD_FIELDS
};

__device__ __forceinline__ long fp32_idx(const Fp32Args& a, int l, int s, int f, long p) { return (((long)l * a.ns + s) * a.hr + f) * a.m + p; }

__global__ __launch_bounds__(256) void fp32_chain_kernel(const Fp32Args a) {
    const int nl = a.net.nl, H = a.net.h, NO = a.net.nout, ns = a.ns;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < a.m; p += (long)gridDim.x * blockDim.x) {
        const long gp = a.p0 + p;
        // ---- S_0: normalised inputs (INF:191) and the tangent seeds
        const float xin[3] = {a.x[gp] * a.sx[0] + a.ox[0], a.y[gp] * a.sx[1] + a.ox[1], a.t[gp] * a.sx[2] + a.ox[2]};
        for (int s = 0; s < ns; ++s)
            for (int k = 0; k < 3; ++k) a.S[fp32_idx(a, 0, s, k, p)] = s == 0 ? xin[k] : (k == s - 1 ? a.sx[k] : 0.0f);
        // ---- hidden layers (INF:192-195):  z = h W + b,  h' = tanh z,  hdot'_k = (1 - h'^2) (hdot_k W)
        for (int l = 0; l < nl; ++l) {
            const int nin = l == 0 ? 3 : H;
            const float* W = a.params + a.net.w_off[l];
            const float* b = a.params + a.net.b_off[l];
            for (int f = 0; f < H; ++f) {
                float acc[FP32_MAX_NS] = {b[f], 0.0f, 0.0f, 0.0f};
                for (int i = 0; i < nin; ++i) {
                    const float w = W[i * H + f];
                    for (int s = 0; s < ns; ++s) acc[s] = fmaf(w, a.S[fp32_idx(a, l, s, i, p)], acc[s]);
                }
                const float h = tanhf(acc[0]);
                const float sd = 1.0f - h * h;
                a.S[fp32_idx(a, l + 1, 0, f, p)] = h;
                for (int s = 1; s < ns; ++s) a.S[fp32_idx(a, l + 1, s, f, p)] = sd * acc[s];
            }
        }
        // ---- output layer (INF:196-198)
        float Y[FP32_MAX_NS][8];
        {
            const float* W = a.params + a.net.w_off[nl];
            const float* b = a.params + a.net.b_off[nl];
            for (int o = 0; o < 8; ++o)
                for (int s = 0; s < FP32_MAX_NS; ++s) Y[s][o] = 0.0f;
            for (int o = 0; o < NO; ++o) {
                float acc[FP32_MAX_NS] = {b[o], 0.0f, 0.0f, 0.0f};
                for (int i = 0; i < H; ++i) {
                    const float w = W[i * NO + o];
                    for (int s = 0; s < ns; ++s) acc[s] = fmaf(w, a.S[fp32_idx(a, nl, s, i, p)], acc[s]);
                }
                for (int s = 0; s < ns; ++s) Y[s][o] = acc[s];
            }
        }
        if (a.head == HEAD_FIELDS) {
            for (int s = 0; s < ns; ++s)
                for (int o = 0; o < NO; ++o) a.fields_out[((long)s * NO + o) * a.n + gp] = Y[s][o];
            continue;
        }
        // ---- head: residuals, their squares, adjoint seeds dL/dY
        float adj[FP32_MAX_NS][8];
        for (int s = 0; s < FP32_MAX_NS; ++s)
            for (int o = 0; o < 8; ++o) adj[s][o] = 0.0f;
        if (a.head == HEAD_WAVE) {
            // outputs (u,v,ut,vt,s11,s22,s12); streams (value, d/dx, d/dy, d/dt)          net_f_sig INF:221-265
            const float e11 = Y[1][0], e22 = Y[2][1], e12 = Y[2][0] + Y[1][1];        // INF:216-218
            float f[7];
            f[0] = Y[1][4] + Y[2][6] - a.rho * Y[3][2];                               // f_u   INF:262
            f[1] = Y[2][5] + Y[1][6] - a.rho * Y[3][3];                               // f_v   INF:263
            f[2] = Y[3][0] - Y[0][2];                                                 // f_ut  INF:248
            f[3] = Y[3][1] - Y[0][3];                                                 // f_vt  INF:249
            f[4] = Y[0][4] - (a.c1 * e11 + a.c2 * e22);                               // f_s11 INF:244
            f[5] = Y[0][5] - (a.c2 * e11 + a.c1 * e22);                               // f_s22 INF:246
            f[6] = Y[0][6] - a.G * e12;                                               // f_s12 INF:245
            float g[7];
            for (int i = 0; i < 7; ++i) {
                a.fsq[(long)i * a.m + p] = f[i] * f[i];
                g[i] = 2.0f * a.tw[i] * f[i];
            }
            adj[0][2] = -g[2];
            adj[0][3] = -g[3];
            adj[0][4] = g[4];
            adj[0][5] = g[5];
            adj[0][6] = g[6];
            adj[1][0] = -a.c1 * g[4] - a.c2 * g[5];
            adj[1][1] = -a.G * g[6];
            adj[1][4] = g[0];
            adj[1][6] = g[1];
            adj[2][0] = -a.G * g[6];
            adj[2][1] = -a.c2 * g[4] - a.c1 * g[5];
            adj[2][5] = g[1];
            adj[2][6] = g[0];
            adj[3][0] = g[2];
            adj[3][1] = g[3];
            adj[3][2] = -a.rho * g[0];
            adj[3][3] = -a.rho * g[1];
        } else {
            for (int o = 0; o < NO; ++o) {
                const float d = Y[0][o] - (a.targets ? a.targets[(long)o * a.n + gp] : 0.0f);
                a.fsq[(long)o * a.m + p] = d * d;
                adj[0][o] = 2.0f * a.tw[o] * d;
            }
        }
        for (int s = 0; s < ns; ++s)
            for (int o = 0; o < NO; ++o) a.Z[fp32_idx(a, nl, s, o, p)] = adj[s][o];
        // ---- reverse (INF:131-133; gradient of TanhGrad):  hbar = Z_l W_l^T,  zbar = sd hbar - 2 h sum_k hdotbar_k hdot_k,  zdotbar_k = sd hdotbar_k
        for (int l = nl; l >= 1; --l) {
            const int nout_l = l == nl ? NO : H;
            const float* W = a.params + a.net.w_off[l];
            for (int i = 0; i < H; ++i) {
                float hb[FP32_MAX_NS] = {0.0f, 0.0f, 0.0f, 0.0f};
                for (int o = 0; o < nout_l; ++o) {
                    const float w = W[i * nout_l + o];
                    for (int s = 0; s < ns; ++s) hb[s] = fmaf(w, a.Z[fp32_idx(a, l, s, o, p)], hb[s]);
                }
                const float h = a.S[fp32_idx(a, l, 0, i, p)];
                const float sd = 1.0f - h * h;
                float dot = 0.0f;
                for (int s = 1; s < ns; ++s) {
                    dot = fmaf(hb[s], a.S[fp32_idx(a, l, s, i, p)], dot);
                    a.Z[fp32_idx(a, l - 1, s, i, p)] = sd * hb[s];
                }
                a.Z[fp32_idx(a, l - 1, 0, i, p)] = sd * hb[0] - 2.0f * h * dot;
            }
        }
    }
}

// one thread per parameter:  Wbar_l[i][o] = sum_points sum_streams S_l[s][i] Z_l[s][o],  bbar_l[o] = sum_points Z_l[0][o]
__global__ __launch_bounds__(256) void fp32_wgrad_kernel(const Fp32Args a, float* grad, int add) {
    const int nl = a.net.nl, H = a.net.h, NO = a.net.nout;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < a.net.nparams; k += gridDim.x * blockDim.x) {
        int l = 0;
        while (l < nl && k >= a.net.w_off[l + 1]) ++l;
        const int nin = l == 0 ? 3 : H, nout_l = l == nl ? NO : H;
        float v = 0.0f;
        if (k < a.net.b_off[l]) {
            const int i = (k - a.net.w_off[l]) / nout_l, o = (k - a.net.w_off[l]) % nout_l;
            (void)nin;
            for (int s = 0; s < a.ns; ++s) {
                const float* Sp = a.S + fp32_idx(a, l, s, i, 0);
                const float* Zp = a.Z + fp32_idx(a, l, s, o, 0);
                for (long p = 0; p < a.m; ++p) v = fmaf(Sp[p], Zp[p], v);
            }
        } else {
            const int o = k - a.net.b_off[l];
            const float* Zp = a.Z + fp32_idx(a, l, 0, o, 0);
            for (long p = 0; p < a.m; ++p) v += Zp[p];
        }
        grad[k] = add ? grad[k] + v : v;
    }
}

// one block per loss term: sum of the pass's squared residuals, added to (or written into) the call's loss sums
__global__ __launch_bounds__(256) void fp32_sum_kernel(const float* fsq, long m, float* loss_out, int add) {
    __shared__ float red[256];
    const float* src = fsq + (long)blockIdx.x * m;
    float v = 0.0f;
    for (long p = threadIdx.x; p < m; p += blockDim.x) v += src[p];
    red[threadIdx.x] = v;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_out[blockIdx.x] = add ? loss_out[blockIdx.x] + red[0] : red[0];
}

// bytes of workspace per point of a pass
inline size_t fp32_bytes_per_point(const NetDesc& net, int ns) {
    const int hr = net.h > 16 ? net.h : 16;
    return ((size_t)2 * (net.nl + 1) * ns * hr + 8) * sizeof(float);
}

}  // namespace pinn
