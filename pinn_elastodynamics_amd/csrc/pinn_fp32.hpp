// Exact-fp32 mode (PINN_PREC_FP32): the same algorithm -- forward tangents, residual head, one reverse pass, weight gradient --
// in plain fp32 FMA arithmetic: no matrix pipe, no 16-bit operand anywhere.  It is what the reference's own arithmetic is (TF1
// float32, INF:71-92) and exists as the on-device third leg of the parity tests (float64 oracle / f16x3 product / fp32 device) and
// for users who want the reference's precision bit-class regardless of speed.  One thread per point, every per-point intermediate
// kept in the workspace as [layer][stream][feature][point] fp32 (point fastest: coalesced); a second kernel with one thread per
// parameter contracts state and adjoint over the points.  Slow by design (~100x the f16x3 path): tests and spot checks only.
//
// Every entry point of include/pinn_hip.h is offered in this mode (round 3): the wave residual head (INF:221-265; 4 streams), the
// value-only data head (INF:111-118; 1 stream), the forward-only fields / streams heads, the plate family -- five streams with the
// second time derivative (PLATE:417-419), composite head P + D*N with plane-stress residuals (PLATE:358-439), hole traction
// (PLATE:452-461), the stream-wise data head of the pre-training losses (PLATE:194-215) -- and the 4-input 3-D extension (five
// first-order streams, oracle/nc3d_oracle.py).
#pragma once
#include "pinn_device.hpp"

namespace pinn {

constexpr int FP32_MAX_NS = 5;
constexpr int FP32_NOUT = 16;          // outputs gathered per point (7 wave, 5 plate, 12 3-D)
constexpr int FP32_TERMS = 16;         // rows of the squared-residual buffer

struct Fp32Args {
    NetDesc net;
    const float* params;       // flat fp32 parameters, reference order
    const float* x;
    const float* y;
    const float* z;            // din = 4 only
    const float* t;
    long n;                    // points of the call
    long p0;                   // first point of this workspace pass
    long m;                    // points of this pass
    float sx[4], ox[4];        // input map per input (x, y, t) or (x, y, z, t)
    float c1, c2, G, rho;
    float tw[16];
    float w5[5][8];            // HEAD_STREAMS: weight per (stream, output)
    float w5n[5][8];           // ... divided by the largest: the loss sums of that head are reported weight-normalised
    const float* targets;      // data heads: [nout][n] or nullptr
    const float* aux;          // HEAD_PLATE: frozen streams [2][5][5][n]; HEAD_TRACTION: [12][n]; HEAD_STREAMS: targets [5][nout][n] or nullptr
    float* S;                  // [nl+1][ns][hr][m]: S_0 = inputs and tangent seeds (first din rows), S_l = state behind hidden layer l
    float* Z;                  // [nl+1][ns][hr][m]: Z_l = adjoint of the pre-activations of weight layer l (Z_nl: nout rows)
    float* fsq;                // [FP32_TERMS][m]: squared residuals of the pass
    float* fields_out;         // fields heads: [ns][nout][n]
    int ns;                    // streams: 1, 4 (value, x, y, t) or 5 (+ tt for din = 3; value, x, y, z, t for din = 4)
    int hr;                    // row stride of S / Z: max(h, 16)
    int head;                  // HEAD_*
    int din;                   // 3 or 4
    int second;                // stream 4 carries the second time derivative (din = 3, ns = 5)
};

__device__ __forceinline__ long fp32_idx(const Fp32Args& a, int l, int s, int f, long p) { return (((long)l * a.ns + s) * a.hr + f) * a.m + p; }

__global__ __launch_bounds__(256) void fp32_chain_kernel(const Fp32Args a) {
    const int nl = a.net.nl, H = a.net.h, NO = a.net.nout, ns = a.ns, din = a.din;
    const int nt = a.second ? 3 : ns - 1;      // first-order tangent streams 1..nt (stream s differentiates by input s - 1; din = 3: the last input is t)
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < a.m; p += (long)gridDim.x * blockDim.x) {
        const long gp = a.p0 + p;
        // ---- S_0: (normalised, INF:191) inputs and the tangent seeds; the second-order stream starts at zero (the input map is affine)
        float xin[4];
        xin[0] = a.x[gp] * a.sx[0] + a.ox[0];
        xin[1] = a.y[gp] * a.sx[1] + a.ox[1];
        if (din == 4) {
            xin[2] = a.z[gp] * a.sx[2] + a.ox[2];
            xin[3] = a.t[gp] * a.sx[3] + a.ox[3];
        } else {
            xin[2] = a.t[gp] * a.sx[2] + a.ox[2];
            xin[3] = 0.0f;
        }
        for (int s = 0; s < ns; ++s)
            for (int k = 0; k < din; ++k) a.S[fp32_idx(a, 0, s, k, p)] = s == 0 ? xin[k] : ((s <= nt && k == s - 1) ? a.sx[k] : 0.0f);
        // ---- hidden layers (INF:192-195):  z = h W + b,  h' = tanh z,  hdot'_k = (1 - h'^2) (hdot_k W),
        //      h_tt' = (1 - h'^2) z_tt - 2 h' hdot'_t z_t  (PLATE:417-419 by the chain rule)
        for (int l = 0; l < nl; ++l) {
            const int nin = l == 0 ? din : H;
            const float* W = a.params + a.net.w_off[l];
            const float* b = a.params + a.net.b_off[l];
            for (int f = 0; f < H; ++f) {
                float acc[FP32_MAX_NS] = {b[f], 0.0f, 0.0f, 0.0f, 0.0f};
                for (int i = 0; i < nin; ++i) {
                    const float w = W[i * H + f];
                    for (int s = 0; s < ns; ++s) acc[s] = fmaf(w, a.S[fp32_idx(a, l, s, i, p)], acc[s]);
                }
                const float h = tanhf(acc[0]);
                const float sd = 1.0f - h * h;
                a.S[fp32_idx(a, l + 1, 0, f, p)] = h;
                for (int s = 1; s <= nt; ++s) a.S[fp32_idx(a, l + 1, s, f, p)] = sd * acc[s];
                if (a.second) a.S[fp32_idx(a, l + 1, 4, f, p)] = sd * acc[4] - 2.0f * h * (sd * acc[3]) * acc[3];
            }
        }
        // ---- output layer (INF:196-198)
        float Y[FP32_MAX_NS][FP32_NOUT];
        {
            const float* W = a.params + a.net.w_off[nl];
            const float* b = a.params + a.net.b_off[nl];
            for (int o = 0; o < FP32_NOUT; ++o)
                for (int s = 0; s < FP32_MAX_NS; ++s) Y[s][o] = 0.0f;
            for (int o = 0; o < NO; ++o) {
                float acc[FP32_MAX_NS] = {b[o], 0.0f, 0.0f, 0.0f, 0.0f};
                for (int i = 0; i < H; ++i) {
                    const float w = W[i * NO + o];
                    for (int s = 0; s < ns; ++s) acc[s] = fmaf(w, a.S[fp32_idx(a, nl, s, i, p)], acc[s]);
                }
                for (int s = 0; s < ns; ++s) Y[s][o] = acc[s];
            }
        }
        if (a.head == HEAD_FIELDS || a.head == HEAD_FIELDS3D) {
            for (int s = 0; s < ns; ++s)
                for (int o = 0; o < NO; ++o) a.fields_out[((long)s * NO + o) * a.n + gp] = Y[s][o];
            continue;
        }
        // ---- head: residuals, their squares, adjoint seeds dL/dY
        float adj[FP32_MAX_NS][FP32_NOUT];
        for (int s = 0; s < FP32_MAX_NS; ++s)
            for (int o = 0; o < FP32_NOUT; ++o) adj[s][o] = 0.0f;
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
        } else if (a.head == HEAD_NC3D) {
            // outputs (u,v,w, ut,vt,wt, s11,s22,s33, s12,s13,s23); streams (value, d/dx, d/dy, d/dz, d/dt); c1 = lambda + 2G, c2 = lambda
            // (oracle/nc3d_oracle.py: the 3-D statement of INF:221-265)
            const float* V = Y[0];
            const float* X = Y[1];
            const float* Yy = Y[2];
            const float* Zz = Y[3];
            const float* T = Y[4];
            const float e11 = X[0], e22 = Yy[1], e33 = Zz[2];
            const float e12 = Yy[0] + X[1], e13 = Zz[0] + X[2], e23 = Zz[1] + Yy[2];
            float f[12];
            f[0] = X[6] + Yy[9] + Zz[10] - a.rho * T[3];
            f[1] = X[9] + Yy[7] + Zz[11] - a.rho * T[4];
            f[2] = X[10] + Yy[11] + Zz[8] - a.rho * T[5];
            f[3] = T[0] - V[3];
            f[4] = T[1] - V[4];
            f[5] = T[2] - V[5];
            f[6] = V[6] - (a.c1 * e11 + a.c2 * (e22 + e33));
            f[7] = V[7] - (a.c1 * e22 + a.c2 * (e11 + e33));
            f[8] = V[8] - (a.c1 * e33 + a.c2 * (e11 + e22));
            f[9] = V[9] - a.G * e12;
            f[10] = V[10] - a.G * e13;
            f[11] = V[11] - a.G * e23;
            float g[12];
            for (int i = 0; i < 12; ++i) {
                a.fsq[(long)i * a.m + p] = f[i] * f[i];
                g[i] = 2.0f * a.tw[i] * f[i];
            }
            adj[0][3] = -g[3];
            adj[0][4] = -g[4];
            adj[0][5] = -g[5];
            for (int i = 6; i < 12; ++i) adj[0][i] = g[i];
            adj[1][0] = -(a.c1 * g[6] + a.c2 * (g[7] + g[8]));
            adj[1][1] = -a.G * g[9];
            adj[1][2] = -a.G * g[10];
            adj[1][6] = g[0];
            adj[1][9] = g[1];
            adj[1][10] = g[2];
            adj[2][0] = -a.G * g[9];
            adj[2][1] = -(a.c1 * g[7] + a.c2 * (g[6] + g[8]));
            adj[2][2] = -a.G * g[11];
            adj[2][9] = g[0];
            adj[2][7] = g[1];
            adj[2][11] = g[2];
            adj[3][0] = -a.G * g[10];
            adj[3][1] = -a.G * g[11];
            adj[3][2] = -(a.c1 * g[8] + a.c2 * (g[6] + g[7]));
            adj[3][10] = g[0];
            adj[3][11] = g[1];
            adj[3][8] = g[2];
            adj[4][0] = g[3];
            adj[4][1] = g[4];
            adj[4][2] = g[5];
            adj[4][3] = -a.rho * g[0];
            adj[4][4] = -a.rho * g[1];
            adj[4][5] = -a.rho * g[2];
        } else if (a.head == HEAD_PLATE) {
            // composite F = P + D*N (PLATE:383-387) with product-rule derivatives, then net_f_sig PLATE:404-439
            // outputs (u,v,s11,s22,s12); streams (value, x, y, t, tt); aux = [D|P][stream][field][n]
            float D[5][5], F[5][5];
            for (int st = 0; st < 5; ++st)
                for (int o = 0; o < 5; ++o) {
                    D[st][o] = a.aux[((long)(0 * 5 + st) * 5 + o) * a.n + gp];
                    F[st][o] = a.aux[((long)(1 * 5 + st) * 5 + o) * a.n + gp];      // start from P
                }
            for (int o = 0; o < 5; ++o) {
                const float n0 = Y[0][o];
                F[0][o] += D[0][o] * n0;
                for (int k = 1; k <= 3; ++k) F[k][o] += D[k][o] * n0 + D[0][o] * Y[k][o];
                F[4][o] += D[4][o] * n0 + 2.0f * D[3][o] * Y[3][o] + D[0][o] * Y[4][o];
            }
            const float e11 = F[1][0], e22 = F[2][1], e12 = F[2][0] + F[1][1];
            float f[5];
            f[0] = F[1][2] + F[2][4] - a.rho * F[4][0];                       // f_u   PLATE:436
            f[1] = F[2][3] + F[1][4] - a.rho * F[4][1];                       // f_v   PLATE:437
            f[2] = F[0][2] - (a.c1 * e11 + a.c2 * e22);                       // f_s11 PLATE:421
            f[3] = F[0][3] - (a.c2 * e11 + a.c1 * e22);                       // f_s22 PLATE:423
            f[4] = F[0][4] - a.G * e12;                                       // f_s12 PLATE:422
            float g[5];
            for (int i = 0; i < 5; ++i) {
                a.fsq[(long)i * a.m + p] = f[i] * f[i];
                g[i] = 2.0f * a.tw[i] * f[i];
            }
            float Fb[5][5];
            for (int st = 0; st < 5; ++st)
                for (int o = 0; o < 5; ++o) Fb[st][o] = 0.0f;
            Fb[0][2] = g[2];
            Fb[0][3] = g[3];
            Fb[0][4] = g[4];
            Fb[1][0] = -a.c1 * g[2] - a.c2 * g[3];
            Fb[2][1] = -a.c2 * g[2] - a.c1 * g[3];
            Fb[2][0] = -a.G * g[4];
            Fb[1][1] = -a.G * g[4];
            Fb[1][2] = g[0];
            Fb[2][4] = g[0];
            Fb[4][0] = -a.rho * g[0];
            Fb[2][3] = g[1];
            Fb[1][4] = g[1];
            Fb[4][1] = -a.rho * g[1];
            for (int o = 0; o < 5; ++o) {
                adj[0][o] = Fb[0][o] * D[0][o] + Fb[1][o] * D[1][o] + Fb[2][o] * D[2][o] + Fb[3][o] * D[3][o] + Fb[4][o] * D[4][o];
                adj[1][o] = Fb[1][o] * D[0][o];
                adj[2][o] = Fb[2][o] * D[0][o];
                adj[3][o] = Fb[3][o] * D[0][o] + 2.0f * Fb[4][o] * D[3][o];
                adj[4][o] = Fb[4][o] * D[0][o];
            }
        } else if (a.head == HEAD_TRACTION) {
            // net_t PLATE:452-461 on the composite values; aux rows: D0[0..4], P0[5..9], nx[10], ny[11]
            float Fv[5], D0[5];
            for (int o = 0; o < 5; ++o) {
                D0[o] = a.aux[(long)o * a.n + gp];
                Fv[o] = a.aux[(long)(5 + o) * a.n + gp] + D0[o] * Y[0][o];
            }
            const float nx = a.aux[10L * a.n + gp], ny = a.aux[11L * a.n + gp];
            const float tx = Fv[2] * nx + Fv[4] * ny, ty = Fv[4] * nx + Fv[3] * ny;
            a.fsq[0L * a.m + p] = tx * tx;
            a.fsq[1L * a.m + p] = ty * ty;
            const float gx = 2.0f * a.tw[0] * tx, gy = 2.0f * a.tw[1] * ty;
            adj[0][2] = gx * nx * D0[2];
            adj[0][3] = gy * ny * D0[3];
            adj[0][4] = (gx * ny + gy * nx) * D0[4];
        } else if (a.head == HEAD_STREAMS) {
            // sum_{s,o} w[s][o] (Y[s][o] - target[s][o])^2 : the loss term of output o collects its streams (PLATE:194-215)
            for (int o = 0; o < NO; ++o) {
                float acc = 0.0f;
                for (int s = 0; s < ns; ++s) {
                    const float d = Y[s][o] - (a.aux ? a.aux[((long)s * NO + o) * a.n + gp] : 0.0f);
                    acc = fmaf(a.w5n[s][o] * d, d, acc);
                    adj[s][o] = 2.0f * a.w5[s][o] * d;
                }
                a.fsq[(long)o * a.m + p] = acc;
            }
        } else {      // HEAD_DATA / HEAD_DATA3D
            for (int o = 0; o < NO; ++o) {
                const float d = Y[0][o] - (a.targets ? a.targets[(long)o * a.n + gp] : 0.0f);
                a.fsq[(long)o * a.m + p] = d * d;
                adj[0][o] = 2.0f * a.tw[o] * d;
            }
        }
        for (int s = 0; s < ns; ++s)
            for (int o = 0; o < NO; ++o) a.Z[fp32_idx(a, nl, s, o, p)] = adj[s][o];
        // ---- reverse (INF:131-133; gradient of TanhGrad):  hbar = Z_l W_l^T,  zbar = sd hbar - 2 h sum_k hdotbar_k hdot_k,  zdotbar_k = sd hdotbar_k
        //      second-order stream, from post-activation state only:  d h_tt/dz = -2 h h_tt - 2 h_t^2,  d h_tt/dz_t = -4 h h_t,  d h_tt/dz_tt = sd
        for (int l = nl; l >= 1; --l) {
            const int nout_l = l == nl ? NO : H;
            const float* W = a.params + a.net.w_off[l];
            for (int i = 0; i < H; ++i) {
                float hb[FP32_MAX_NS] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                for (int o = 0; o < nout_l; ++o) {
                    const float w = W[i * nout_l + o];
                    for (int s = 0; s < ns; ++s) hb[s] = fmaf(w, a.Z[fp32_idx(a, l, s, o, p)], hb[s]);
                }
                const float h = a.S[fp32_idx(a, l, 0, i, p)];
                const float sd = 1.0f - h * h;
                float dot = 0.0f;
                float zd[FP32_MAX_NS] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                for (int s = 1; s <= nt; ++s) {
                    dot = fmaf(hb[s], a.S[fp32_idx(a, l, s, i, p)], dot);
                    zd[s] = sd * hb[s];
                }
                float zb = sd * hb[0] - 2.0f * h * dot;
                if (a.second) {
                    const float ht = a.S[fp32_idx(a, l, 3, i, p)], htt = a.S[fp32_idx(a, l, 4, i, p)];
                    zd[4] = sd * hb[4];
                    zd[3] -= 4.0f * h * ht * hb[4];
                    zb += hb[4] * (-2.0f * h * htt - 2.0f * ht * ht);
                }
                a.Z[fp32_idx(a, l - 1, 0, i, p)] = zb;
                for (int s = 1; s < ns; ++s) a.Z[fp32_idx(a, l - 1, s, i, p)] = zd[s];
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
        const int nout_l = l == nl ? NO : H;
        float v = 0.0f;
        if (k < a.net.b_off[l]) {
            const int i = (k - a.net.w_off[l]) / nout_l, o = (k - a.net.w_off[l]) % nout_l;
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
    return ((size_t)2 * (net.nl + 1) * ns * hr + FP32_TERMS) * sizeof(float);
}

}  // namespace pinn
