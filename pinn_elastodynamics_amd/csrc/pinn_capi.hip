// extern "C" entry points of libpinn_hip.so (declared in include/pinn_hip.h): argument checking,
// decoding into pinn::Call, and dispatch to the kernel family for (precision_mode, hidden width).
#include "pinn_host.hpp"
#include "pinn_fp32.hpp"

#ifndef PINN_VARIANTS_DEF
#define PINN_VARIANTS_DEF "pinn_variants.def"     // experiments (tools/exp_build.sh) build a one-variant library
#endif

namespace pinn {
#define PINN_VARIANT(op, split, width) const Impl* impl_##op##_##split##_##width();
#include PINN_VARIANTS_DEF
#undef PINN_VARIANT

static const Impl* find_impl(int precision_mode, int width) {
    const char* op;
    int split;
    switch (precision_mode) {
        case PINN_PREC_BF16: op = "BF16"; split = 1; break;
        case PINN_PREC_F16X3: op = "F16"; split = 3; break;
        case PINN_PREC_F16: op = "F16"; split = 1; break;
        case PINN_PREC_BF16X3: op = "BF16"; split = 3; break;
        default: return nullptr;
    }
    auto same = [](const char* a, const char* b) { while (*a && *a == *b) { ++a; ++b; } return *a == *b; };
#define PINN_VARIANT(o, s, w) if (same(op, #o) && split == s && width == w) return impl_##o##_##s##_##w();
#include PINN_VARIANTS_DEF
#undef PINN_VARIANT
    return nullptr;
}

// din: 3 for the reference's (x, y, t) nets (up to 8 outputs), 4 for the (x, y, z, t) nets of the 3-D entry points (up to 16)
static int decode_net(const int* layers, int n_layers, NetDesc& net, int& width, int din = 3) {
    if (!layers) return PINN_ERR_NULL;
    if (n_layers < 3 || n_layers - 1 > MAX_WLAYERS) return PINN_ERR_LAYERS;
    if (layers[0] != din) return PINN_ERR_LAYERS;
    const int h = layers[1], nout = layers[n_layers - 1];
    for (int i = 1; i < n_layers - 1; ++i) if (layers[i] != h) return PINN_ERR_LAYERS;
    if (nout < 1 || nout > (din == 4 ? 16 : 8)) return PINN_ERR_LAYERS;
    width = pinn_supported_width(h);
    if (!width) return PINN_ERR_LAYERS;
    net.nl = n_layers - 2;
    net.h = h;
    net.nout = nout;
    net.din = din;
    int o = 0;
    for (int l = 0; l <= net.nl; ++l) {
        const int n_in = layers[l], n_out = layers[l + 1];
        net.w_off[l] = o;
        o += n_in * n_out;
        net.b_off[l] = o;
        o += n_out;
    }
    for (int l = net.nl + 1; l < MAX_WLAYERS; ++l) net.w_off[l] = net.b_off[l] = 0;
    net.nparams = o;
    return PINN_OK;
}

static void input_map(const double* lb, const double* ub, int normalize, Call& c, int din = 3) {
    c.sx[3] = 1.0f;
    c.ox[3] = 0.0f;
    for (int k = 0; k < din; ++k) {
        if (normalize) {   // INF:191  H = 2 (X - lb)/(ub - lb) - 1
            const double s = 2.0 / (ub[k] - lb[k]);
            c.sx[k] = (float)s;
            c.ox[k] = (float)(-lb[k] * s - 1.0);
        } else {
            c.sx[k] = 1.0f;
            c.ox[k] = 0.0f;
        }
    }
}
}  // namespace pinn

using namespace pinn;

static int g_use_fused = 1;
static unsigned long long* g_dbg_stamps = nullptr;
static float* g_prof_ms = nullptr;
static ProfRing g_ring;
static int g_ring_every = 1;
#ifndef PINN_XCD_TAIL_DEFAULT
#define PINN_XCD_TAIL_DEFAULT 16      // 1.6 % more steps for the even XCDs (A / B on one box, 96 interleaved launches each: 0 / 12 / 20 permille -> 4.492 / 4.437 / 4.432 ms per 2 M points; profiles/r05_xcd_bonus_ab.txt)
#endif
namespace pinn {
long g_path_counts[5] = {0, 0, 0, 0, 0};
int g_xcd_tail_permille = PINN_XCD_TAIL_DEFAULT;
int g_fused_grid_cap = 0;
// The XCD-aware tail assumes what was measured on an MI355X in SPX mode: 256 compute units in 8 XCDs, workgroup b on XCD b % 8, the even
// XCDs 2-3 % faster on this kernel.  Any other device (another part, a partitioned one: CPX / DPX show fewer compute units) keeps the plain
// round-robin loop.  Asked once per device.
bool xcd_tail_device_ok() {
#if defined(PINN_SIMT_EMULATOR)
    return true;       // (the x86 test build exercises the index arithmetic on small grids)
#else
    static int cached[64];      // 0 unknown, 1 yes, 2 no
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (cached[dev] == 0) {
        hipDeviceProp_t pr;
        bool ok = hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount == 256;
        if (ok) {
            const char* a = pr.gcnArchName;
            ok = a[0] == 'g' && a[1] == 'f' && a[2] == 'x' && a[3] == '9' && a[4] == '5' && a[5] == '0';
        }
        cached[dev] = ok ? 1 : 2;
    }
    return cached[dev] == 1;
#endif
}
}

template <class F>
static int cache_policy_of(size_t* images, size_t* sums) {
    if (images) *images = F::IMAGES_WG_BYTES;
    if (sums) *sums = F::SUMS_WG_BYTES;
    return F::NT_SUMS ? 1 : (F::NT_IMAGES ? 2 : 0);
}

extern "C" {

int pinn_abi_version(void) { return 2; }      // 2 (round 6): PINN_IPC_HANDLE_BYTES 64 -> 128, pinn_p2p_set_timeout_ms / _peek_status, pinn_wave2d_step_checked

float pinn_fused_weight_limit(void) { return FUSED_OPERAND_MAX / FUSED_WEIGHT_SCALE; }

int pinn_debug_wall_clock_khz(void) {
#if defined(PINN_SIMT_EMULATOR)
    return 0;
#else
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) return 0;
    return khz;
#endif
}
void pinn_debug_set_stamp_buffer(void* device_u64x128) { g_dbg_stamps = static_cast<unsigned long long*>(device_u64x128); }
void pinn_debug_set_profile_buffer(float* host_ms4) { g_prof_ms = host_ms4; }

int pinn_debug_profile_ring_arm(int max_launches) {
    if (max_launches < 0) max_launches = 0;
    g_ring.limit = max_launches > ProfRing::CAP ? ProfRing::CAP : max_launches;
    g_ring.n = 0;
    g_ring.every = g_ring_every;
    g_ring.steps_seen = 0;
    g_ring.step_on = true;
    g_ring.armed = g_ring.limit > 0;
    return g_ring.limit;
}

void pinn_debug_profile_ring_stride(int every) { g_ring_every = every < 1 ? 1 : every; }

void pinn_debug_path_counts(int64_t counts[5], int reset) {
    for (int i = 0; i < 5; ++i) {
        if (counts) counts[i] = (int64_t)g_path_counts[i];
        if (reset) g_path_counts[i] = 0;
    }
}

int pinn_debug_profile_ring_read(float* ms_out, int* streams_out, int capacity) {
    g_ring.armed = false;
    int m = 0;
    for (int i = 0; i < g_ring.n && m < capacity; ++i) {
        if (hipEventSynchronize(g_ring.ev[i][1]) != hipSuccess) break;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_ring.ev[i][0], g_ring.ev[i][1]) != hipSuccess) break;
        if (ms_out) ms_out[m] = ms;
        if (streams_out) streams_out[m] = g_ring.tag[i];
        ++m;
    }
    g_ring.n = 0;
    return m;
}

int pinn_debug_set_xcd_bonus(int permille) {
    const int old = g_xcd_tail_permille;
    g_xcd_tail_permille = permille < 0 ? 0 : (permille > 200 ? 200 : permille);
    return old;
}

int pinn_debug_cache_policy(const int* layers, int n_layers, int head, size_t* images_bytes, size_t* sums_bytes) {
    const int din = head == PINN_HEAD_NC3D ? 4 : 3;
    NetDesc net;
    int width = 0;
    const int rc = decode_net(layers, n_layers, net, width, din);
    if (rc) return rc;
    // (the instantiations the f16x3 families ship: pinn_host.hpp, Host::fused_depth / fused_has_3d)
    if (head == PINN_HEAD_NC3D) {
        if (width == 128 && net.nl == 10 && net.nout == 12) return cache_policy_of<Fused<OpF16, 3, 128, 10, 5, false, 4>>(images_bytes, sums_bytes);
        return PINN_ERR_LAYERS;
    }
    if (head == PINN_HEAD_PLATE) {
        if (width == 32 && net.nl == 4) return cache_policy_of<Fused<OpF16, 3, 32, 4, 5>>(images_bytes, sums_bytes);
        if (width == 64 && net.nl == 4) return cache_policy_of<Fused<OpF16, 3, 64, 4, 5>>(images_bytes, sums_bytes);
        if (width == 64 && net.nl == 8) return cache_policy_of<Fused<OpF16, 3, 64, 8, 5>>(images_bytes, sums_bytes);
        if (width == 96 && net.nl == 8) return cache_policy_of<Fused<OpF16, 3, 96, 8, 5>>(images_bytes, sums_bytes);
        return PINN_ERR_LAYERS;
    }
    if (head != PINN_HEAD_WAVE) return PINN_ERR_LAYERS;
    if (width == 32 && net.nl == 4) return cache_policy_of<Fused<OpF16, 3, 32, 4, 4>>(images_bytes, sums_bytes);
    if (width == 32 && net.nl == 8) return cache_policy_of<Fused<OpF16, 3, 32, 8, 4>>(images_bytes, sums_bytes);
    if (width == 64 && net.nl == 4) return cache_policy_of<Fused<OpF16, 3, 64, 4, 4>>(images_bytes, sums_bytes);
    if (width == 64 && net.nl == 8) return cache_policy_of<Fused<OpF16, 3, 64, 8, 4>>(images_bytes, sums_bytes);
    if (width == 96 && net.nl == 8) return cache_policy_of<Fused<OpF16, 3, 96, 8, 4>>(images_bytes, sums_bytes);
    if (width == 128 && net.nl == 8) return cache_policy_of<Fused<OpF16, 3, 128, 8, 4>>(images_bytes, sums_bytes);
    if (width == 160 && net.nl == 6) return cache_policy_of<Fused<OpF16, 3, 160, 6, 4>>(images_bytes, sums_bytes);
    return PINN_ERR_LAYERS;
}

int pinn_debug_set_fused_grid_cap(int cap) { const int old = g_fused_grid_cap; g_fused_grid_cap = cap < 0 ? 0 : cap; return old; }

int pinn_debug_set_fused(int enable) {
    const int old = g_use_fused;
    g_use_fused = enable ? 1 : 0;
    return old;
}

int pinn_supported_width(int h) {
    if (h < 1) return 0;
    static const int widths[] = {32, 64, 96, 128, 160};
    for (int w : widths) if (h <= w) return w;
    return 0;
}

const char* pinn_error_string(int code) {
    switch (code) {
        case PINN_OK: return "ok";
        case PINN_ERR_NULL: return "required pointer is NULL";
        case PINN_ERR_LAYERS: return "unsupported layer list (need {3, H x k, n_out<=8} -- {4, H x k, 12} for the 3-D entry points --, H<=160, <=16 weight layers, and a compiled variant)";
        case PINN_ERR_PRECISION: return "unknown precision_mode";
        case PINN_ERR_WORKSPACE: return "workspace too small or not 256-byte aligned";
        case PINN_ERR_SIZE: return "n must not be negative";
        case PINN_ERR_COLLECTIVE: return "p2p collective: not connected, a coarse-grained buffer across devices, or a rank did not arrive within the bounded wait (the call failed as a whole: buffer NaN, no Adam update)";
        case PINN_ERR_RANGE: return "gradient non-finite even on the two-kernel path with the reverse pass scaled by 2^-24 (pinn_wave2d_loss_grad_checked)";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}

// An empty point set contributes nothing: zero sums, and a zero gradient unless the call accumulates.  (The reference would
// produce NaN means there; a data-parallel rank that gets no rows of a small set must be able to make the same call sequence.)
static int empty_batch(const Call& c, int nterms) {
    hipStream_t st = static_cast<hipStream_t>(c.stream);
    int rc = (int)hipMemsetAsync(c.loss_out, 0, (size_t)nterms * sizeof(float), st);
    if (!rc && !c.accumulate) rc = (int)hipMemsetAsync(c.grad_out, 0, (size_t)c.net.nparams * sizeof(float), st);
    return rc;
}

static int prepare(const float* params, const int* layers, int n_layers, const float* x, const float* y, const float* t, int64_t n,
                   const double* lb, const double* ub, int normalize, int precision_mode, void* ws, size_t ws_bytes, void* stream,
                   Call& c, const Impl*& impl, int din = 3, const float* z = nullptr) {
    if (!params || !ws) return PINN_ERR_NULL;
    if (n < 0) return PINN_ERR_SIZE;
    if (n > 0 && (!x || !y || !t || (din == 4 && !z))) return PINN_ERR_NULL;          // n == 0 is a valid empty batch (see empty_batch)
    if (normalize && (!lb || !ub)) return PINN_ERR_NULL;
    c.weights_packed = (precision_mode & PINN_FLAG_WEIGHTS_PACKED) ? 1 : 0;
    c.fast_state = (precision_mode & PINN_FLAG_STATE_FP16) ? 1 : 0;
    const bool two_kernel = (precision_mode & PINN_FLAG_TWO_KERNEL) != 0;
    c.adj_shift = (precision_mode >> 16) & 0x1f;
    precision_mode &= ~(PINN_FLAG_WEIGHTS_PACKED | PINN_FLAG_STATE_FP16 | PINN_FLAG_TWO_KERNEL | (0x1f << 16));
    if (precision_mode < 0 || precision_mode > PINN_PREC_FP32) return PINN_ERR_PRECISION;
    int width = 0;
    const int rc = decode_net(layers, n_layers, c.net, width, din);
    if (rc) return rc;
    // PINN_PREC_FP32 has no kernel family: impl stays NULL and the entry points that offer the mode branch to fp32_call()
    impl = precision_mode == PINN_PREC_FP32 ? nullptr : find_impl(precision_mode, width);
    if (!impl && precision_mode != PINN_PREC_FP32) return PINN_ERR_LAYERS;
    c.params = params;
    c.z = z;
    c.nsets = 0;
    c.targets = nullptr;
    c.x = x;
    c.y = y;
    c.t = t;
    c.n = (long)n;
    input_map(lb, ub, normalize, c, din);
    c.ws = ws;
    c.ws_bytes = ws_bytes;
    c.stream = (hipStream_t)stream;
    c.loss_out = nullptr;
    c.grad_out = nullptr;
    c.accumulate = 0;
    c.c1 = c.c2 = c.G = c.rho = 0.0f;
    for (int i = 0; i < 16; ++i) c.tw[i] = 0.0f;
    c.targets = nullptr;
    c.fields_out = nullptr;
    c.aux = nullptr;
    for (int i = 0; i < 5; ++i)
        for (int o = 0; o < 8; ++o) c.w5[i][o] = 0.0f;
    c.prof_ms = g_prof_ms;
    c.ring = g_ring.armed ? &g_ring : nullptr;
    c.use_fused = two_kernel ? 0 : g_use_fused;
    c.one_stream_head = 0;
    c.dbg_stamps = g_dbg_stamps;
    return PINN_OK;
}

// the sizing entry points accept the same precision_mode word as the calls (flag and shift bits are ignored) and both input counts
static int mode_only(int precision_mode) { return precision_mode & ~(PINN_FLAG_WEIGHTS_PACKED | PINN_FLAG_STATE_FP16 | PINN_FLAG_TWO_KERNEL | (0x1f << 16)); }

size_t pinn_workspace_bytes(const int* layers, int n_layers, int64_t n, int precision_mode) {
    NetDesc net;
    int width = 0;
    if (n <= 0 || !layers || decode_net(layers, n_layers, net, width, layers[0] == 4 ? 4 : 3)) return 0;
    if (mode_only(precision_mode) == PINN_PREC_FP32) return align_up(fp32_bytes_per_point(net, FP32_MAX_NS) * (size_t)(n < 256 ? 256 : (n < 65536 ? n : 65536)), 256);
    const Impl* impl = find_impl(mode_only(precision_mode), width);
    return impl ? impl->ws_bytes(net, (long)n, 0) : 0;
}

int pinn_path_for(const int* layers, int n_layers, int precision_mode, int head, size_t ws_bytes) {
    if (!layers) return PINN_ERR_NULL;
    if (head < PINN_HEAD_WAVE || head > PINN_HEAD_STREAMS) return PINN_ERR_LAYERS;
    const int din = (head == PINN_HEAD_NC3D || head == PINN_HEAD_NC3D_DATA) ? 4 : 3;
    NetDesc net;
    int width = 0;
    const int rc = decode_net(layers, n_layers, net, width, din);
    if (rc) return rc;
    const int mode = mode_only(precision_mode);
    if (mode < 0 || mode > PINN_PREC_FP32) return PINN_ERR_PRECISION;
    if (mode == PINN_PREC_FP32) return PINN_PATH_FP32;
    const Impl* impl = find_impl(mode, width);
    if (!impl) return PINN_ERR_LAYERS;
    const int path = impl->path_for(net, head, ws_bytes);
    if (path > 0 && ((precision_mode & PINN_FLAG_TWO_KERNEL) || !g_use_fused)) return PINN_PATH_TWO_KERNEL;
    return path;
}

size_t pinn_min_workspace_bytes(const int* layers, int n_layers, int precision_mode) {
    NetDesc net;
    int width = 0;
    if (!layers || decode_net(layers, n_layers, net, width, layers[0] == 4 ? 4 : 3)) return 0;
    if (mode_only(precision_mode) == PINN_PREC_FP32) return align_up(fp32_bytes_per_point(net, FP32_MAX_NS) * (size_t)256, 256);
    const Impl* impl = find_impl(mode_only(precision_mode), width);
    return impl ? impl->ws_bytes(net, 1L << 40, 1) : 0;
}

// PINN_PREC_FP32: plain fp32 arithmetic (pinn_fp32.hpp), the points walked in as many passes as the workspace holds.
// ns = streams of the head (1: data / traction heads, 4: wave / fields, 5: plate family and the 4-input heads), din = inputs.
static int fp32_call(const Call& c, int head, int nterms, int ns, int din = 3) {
    hipStream_t st = c.stream;
    const size_t per_point = fp32_bytes_per_point(c.net, ns);
    if (((uintptr_t)c.ws & 255) != 0 || c.ws_bytes < per_point * 256) return PINN_ERR_WORKSPACE;
    ++g_path_counts[PINN_PATH_FP32];
    long mmax = (long)(c.ws_bytes / per_point);
    if (mmax > (1L << 20)) mmax = 1L << 20;
    Fp32Args a;
    a.net = c.net;
    a.params = c.params;
    a.x = c.x;
    a.y = c.y;
    a.z = c.z;
    a.t = c.t;
    a.n = c.n;
    for (int k = 0; k < 4; ++k) { a.sx[k] = c.sx[k]; a.ox[k] = c.ox[k]; }
    a.c1 = c.c1;
    a.c2 = c.c2;
    a.G = c.G;
    a.rho = c.rho;
    for (int i = 0; i < 16; ++i) a.tw[i] = c.tw[i];
    // stream-wise data head: the loss sums are reported weight-normalised like the 16-bit kernels do (sum_s w[s][o] / max|w| d^2)
    float wmax = 0.0f;
    for (int i = 0; i < 5; ++i)
        for (int o = 0; o < 8; ++o) { const float v = c.w5[i][o] < 0 ? -c.w5[i][o] : c.w5[i][o]; if (v > wmax) wmax = v; }
    for (int i = 0; i < 5; ++i)
        for (int o = 0; o < 8; ++o) { a.w5[i][o] = c.w5[i][o]; a.w5n[i][o] = wmax > 0.0f ? c.w5[i][o] / wmax : 0.0f; }
    a.targets = c.targets;
    a.aux = c.aux;
    a.fields_out = c.fields_out;
    a.ns = ns;
    a.hr = c.net.h > 16 ? c.net.h : 16;
    a.head = head;
    a.din = din;
    a.second = (din == 3 && ns == 5) ? 1 : 0;
    const bool forward_only = head == HEAD_FIELDS || head == HEAD_FIELDS3D;
    int pass = 0;
    for (long p0 = 0; p0 < c.n; p0 += mmax, ++pass) {
        a.p0 = p0;
        a.m = c.n - p0 < mmax ? c.n - p0 : mmax;
        const size_t tensor = (size_t)(c.net.nl + 1) * ns * a.hr * a.m;
        a.S = static_cast<float*>(c.ws);
        a.Z = a.S + tensor;
        a.fsq = a.Z + tensor;
        int blocks = (int)((a.m + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(fp32_chain_kernel, dim3(blocks), dim3(256), 0, st, a);
        if (!forward_only) {
            hipLaunchKernelGGL(fp32_sum_kernel, dim3(nterms), dim3(256), 0, st, (const float*)a.fsq, a.m, c.loss_out, pass > 0 ? 1 : 0);
            hipLaunchKernelGGL(fp32_wgrad_kernel, dim3((c.net.nparams + 255) / 256), dim3(256), 0, st, a, c.grad_out, (c.accumulate || pass > 0) ? 1 : 0);
        }
        const int rc = (int)hipGetLastError();
        if (rc) return rc;
    }
    return PINN_OK;
}

// lr_t of the TF1 rule (bias correction folded into the step size, computed in double)
static float adam_lr_t(double lr, double beta1, double beta2, int64_t step) {
    const double b1t = __builtin_pow(beta1, (double)step), b2t = __builtin_pow(beta2, (double)step);      // (pinn_adam_step's own expression: the same bits)
    const double lr_t = lr * __builtin_sqrt(1.0 - b2t) / (1.0 - b1t);
    return (float)lr_t;
}

// Hooke coefficients: plane strain INF:238-241, plane stress PLATE:416-418
static void set_hooke(Call& c, double E, double mu, double rho, int plane_strain) {
    double c1, c2;
    if (plane_strain) {
        const double coef = E / ((1.0 + mu) * (1.0 - 2.0 * mu));
        c1 = coef * (1.0 - mu);
        c2 = coef * mu;
    } else {
        c1 = E / (1.0 - mu * mu);
        c2 = E * mu / (1.0 - mu * mu);
    }
    c.c1 = (float)c1;
    c.c2 = (float)c2;
    c.G = (float)(E / (2.0 * (1.0 + mu)));
    c.rho = (float)rho;
}

static int wave2d_loss_grad_impl(const float* params_flat, const int* layers, int n_layers, const float* x, const float* y, const float* t,
                          int64_t n, const double lb[3], const double ub[3], int normalize, double E, double mu, double rho,
                          int plane_strain, const float term_weights[7], float* loss_terms_out, float* grad_flat_out, int accumulate,
                          int precision_mode, void* workspace, size_t ws_bytes, void* stream, float* prof_ms) {
    Call c;
    const Impl* impl = nullptr;
    int rc = prepare(params_flat, layers, n_layers, x, y, t, n, lb, ub, normalize, precision_mode, workspace, ws_bytes, stream, c, impl);
    if (rc) return rc;
    if (!term_weights || !loss_terms_out || !grad_flat_out) return PINN_ERR_NULL;
    if (c.net.nout != 7) return PINN_ERR_LAYERS;
    set_hooke(c, E, mu, rho, plane_strain);
    for (int i = 0; i < 7; ++i) c.tw[i] = term_weights[i];
    c.loss_out = loss_terms_out;
    c.grad_out = grad_flat_out;
    c.accumulate = accumulate;
    if (prof_ms) c.prof_ms = prof_ms;
    if (n == 0) return empty_batch(c, 7);
    if (!impl) return fp32_call(c, HEAD_WAVE, 7, 4);
    return impl->wave_loss_grad(c);
}

int pinn_wave2d_loss_grad(const float* params_flat, const int* layers, int n_layers, const float* x, const float* y, const float* t,
                          int64_t n, const double lb[3], const double ub[3], int normalize, double E, double mu, double rho,
                          int plane_strain, const float term_weights[7], float* loss_terms_out, float* grad_flat_out, int accumulate,
                          int precision_mode, void* workspace, size_t ws_bytes, void* stream) {
    return wave2d_loss_grad_impl(params_flat, layers, n_layers, x, y, t, n, lb, ub, normalize, E, mu, rho, plane_strain, term_weights,
                                 loss_terms_out, grad_flat_out, accumulate, precision_mode, workspace, ws_bytes, stream, nullptr);
}

int pinn_wave2d_loss_grad_profile(const float* params_flat, const int* layers, int n_layers, const float* x, const float* y, const float* t,
                                  int64_t n, const double lb[3], const double ub[3], int normalize, double E, double mu, double rho,
                                  int plane_strain, const float term_weights[7], float* loss_terms_out, float* grad_flat_out,
                                  int accumulate, int precision_mode, void* workspace, size_t ws_bytes, void* stream, float kernel_ms[4]) {
    if (!kernel_ms) return PINN_ERR_NULL;
    return wave2d_loss_grad_impl(params_flat, layers, n_layers, x, y, t, n, lb, ub, normalize, E, mu, rho, plane_strain, term_weights,
                                 loss_terms_out, grad_flat_out, accumulate, precision_mode, workspace, ws_bytes, stream, kernel_ms);
}

int pinn_data_loss_grad(const float* params_flat, const int* layers, int n_layers, const float* x, const float* y, const float* t,
                        int64_t n, const double lb[3], const double ub[3], int normalize, const float* targets,
                        const float* out_weights, float* loss_terms_out, float* grad_flat_out, int accumulate, int precision_mode,
                        void* workspace, size_t ws_bytes, void* stream) {
    Call c;
    const Impl* impl = nullptr;
    int rc = prepare(params_flat, layers, n_layers, x, y, t, n, lb, ub, normalize, precision_mode, workspace, ws_bytes, stream, c, impl);
    if (rc) return rc;
    if (!out_weights || !loss_terms_out || !grad_flat_out) return PINN_ERR_NULL;
    for (int i = 0; i < c.net.nout; ++i) c.tw[i] = out_weights[i];
    c.targets = targets;
    c.loss_out = loss_terms_out;
    c.grad_out = grad_flat_out;
    c.accumulate = accumulate;
    if (n == 0) return empty_batch(c, c.net.nout);
    if (!impl) return fp32_call(c, HEAD_DATA, c.net.nout, 1);
    return impl->data_loss_grad(c);
}

int pinn_data_loss_grad_multi(const float* params_flat, const int* layers, int n_layers, const pinn_point_set* sets, int n_sets,
                              const double lb[3], const double ub[3], int normalize, float* grad_flat_out, int accumulate, int precision_mode,
                              void* workspace, size_t ws_bytes, void* stream) {
    if (!sets || n_sets < 1 || n_sets > PINN_MAX_SETS || !grad_flat_out) return n_sets < 1 || n_sets > PINN_MAX_SETS ? PINN_ERR_SIZE : PINN_ERR_NULL;
    int64_t nmax = 0, ntot = 0;
    const pinn_point_set* first = nullptr;
    for (int k = 0; k < n_sets; ++k) {
        if (sets[k].n < 0) return PINN_ERR_SIZE;
        if (!sets[k].loss_terms_out || (sets[k].n > 0 && (!sets[k].x || !sets[k].y || !sets[k].t))) return PINN_ERR_NULL;
        if (sets[k].n > 0 && !first) first = &sets[k];
        if (sets[k].n > nmax) nmax = sets[k].n;
        ntot += sets[k].n;
    }
    Call c;
    const Impl* impl = nullptr;
    int rc = prepare(params_flat, layers, n_layers, first ? first->x : nullptr, first ? first->y : nullptr, first ? first->t : nullptr, nmax, lb, ub,
                     normalize, precision_mode, workspace, ws_bytes, stream, c, impl);
    if (rc) return rc;
    c.grad_out = grad_flat_out;
    c.accumulate = accumulate;
    c.targets = nullptr;
    c.nsets = n_sets;
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int k = 0; k < n_sets; ++k) {
        c.sets[k].x = sets[k].x;
        c.sets[k].y = sets[k].y;
        c.sets[k].t = sets[k].t;
        c.sets[k].targets = sets[k].targets;
        c.sets[k].n = (long)sets[k].n;
        for (int i = 0; i < 8; ++i) c.sets[k].tw[i] = i < c.net.nout ? sets[k].out_weights[i] : 0.0f;
        c.sets[k].loss_out = sets[k].loss_terms_out;
        if (sets[k].n == 0 && (rc = (int)hipMemsetAsync(sets[k].loss_terms_out, 0, (size_t)c.net.nout * sizeof(float), st))) return rc;
    }
    if (ntot == 0) {
        if (!accumulate) return (int)hipMemsetAsync(grad_flat_out, 0, (size_t)c.net.nparams * sizeof(float), st);
        return 0;
    }
    if (!impl) {            // PINN_PREC_FP32: one set after the other into the same gradient
        bool first_set = true;
        for (int k = 0; k < n_sets; ++k) {
            if (sets[k].n <= 0) continue;
            Call s1 = c;
            s1.nsets = 0;
            s1.x = sets[k].x;
            s1.y = sets[k].y;
            s1.t = sets[k].t;
            s1.n = (long)sets[k].n;
            s1.targets = sets[k].targets;
            for (int i = 0; i < 8; ++i) s1.tw[i] = c.sets[k].tw[i];
            s1.loss_out = sets[k].loss_terms_out;
            s1.accumulate = accumulate || !first_set;
            if ((rc = fp32_call(s1, HEAD_DATA, c.net.nout, 1))) return rc;
            first_set = false;
        }
        return PINN_OK;
    }
    return impl->data_loss_grad(c);
}

int pinn_wave2d_step(float* params_flat, const int* layers, int n_layers, const float* x, const float* y, const float* t, int64_t n,
                     const double lb[3], const double ub[3], int normalize, double E, double mu, double rho, int plane_strain,
                     const float term_weights[7], float* loss_terms_out, const pinn_point_set* sets, int n_sets, float* grad_flat_out,
                     int accumulate, const pinn_adam_state* adam, int precision_mode, void* workspace, size_t ws_bytes, void* stream) {
    if (n_sets < 0 || n_sets > PINN_MAX_SETS) return PINN_ERR_SIZE;
    if (n_sets > 0 && !sets) return PINN_ERR_NULL;
    if (adam && (!adam->m || !adam->v || adam->step < 1)) return adam->step < 1 ? PINN_ERR_SIZE : PINN_ERR_NULL;
    int64_t side_total = 0;
    for (int k = 0; k < n_sets; ++k) {
        if (sets[k].n < 0) return PINN_ERR_SIZE;
        if (!sets[k].loss_terms_out || (sets[k].n > 0 && (!sets[k].x || !sets[k].y || !sets[k].t))) return PINN_ERR_NULL;
        side_total += sets[k].n;
    }
    // ---- the one-launch form: collocation set and side sets through fused_step_kernel, one reduction (+ Adam) behind it
    if (n > 0 && side_total > 0 && term_weights && loss_terms_out && grad_flat_out) {
        Call c;
        const Impl* impl = nullptr;
        int rc = prepare(params_flat, layers, n_layers, x, y, t, n, lb, ub, normalize, precision_mode, workspace, ws_bytes, stream, c, impl);
        if (rc) return rc;
        if (c.net.nout != 7) return PINN_ERR_LAYERS;
        if (impl) {
            set_hooke(c, E, mu, rho, plane_strain);
            for (int i = 0; i < 7; ++i) c.tw[i] = term_weights[i];
            c.loss_out = loss_terms_out;
            c.grad_out = grad_flat_out;
            c.accumulate = accumulate;
            Call d = c;
            for (int i = 0; i < 16; ++i) d.tw[i] = 0.0f;
            d.loss_out = nullptr;
            d.nsets = n_sets;
            hipStream_t st = static_cast<hipStream_t>(stream);
            for (int k = 0; k < n_sets; ++k) {
                d.sets[k].x = sets[k].x;
                d.sets[k].y = sets[k].y;
                d.sets[k].t = sets[k].t;
                d.sets[k].targets = sets[k].targets;
                d.sets[k].n = (long)sets[k].n;
                for (int i = 0; i < 8; ++i) d.sets[k].tw[i] = i < c.net.nout ? sets[k].out_weights[i] : 0.0f;
                d.sets[k].loss_out = sets[k].loss_terms_out;
            }
            AdamEpilogue ep = {nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, 0.f};
            if (adam) ep = AdamEpilogue{params_flat, adam->m, adam->v, adam_lr_t(adam->lr, adam->beta1, adam->beta2, adam->step), (float)adam->beta1, (float)adam->beta2, (float)adam->eps};
            if (impl->wave_step(c, d, ep, &rc)) {
                if (rc) return rc;
                for (int k = 0; k < n_sets; ++k)      // (empty sets report zeros, as in pinn_data_loss_grad_multi)
                    if (sets[k].n == 0 && (rc = (int)hipMemsetAsync(sets[k].loss_terms_out, 0, (size_t)c.net.nout * sizeof(float), st))) return rc;
                return PINN_OK;
            }
        }
    }
    // ---- every other case (other widths / depths, PINN_PREC_FP32, an empty set, a small workspace): the same results from the calls one by one
    int rc = pinn_wave2d_loss_grad(params_flat, layers, n_layers, x, y, t, n, lb, ub, normalize, E, mu, rho, plane_strain, term_weights, loss_terms_out,
                                   grad_flat_out, accumulate, precision_mode, workspace, ws_bytes, stream);
    if (rc) return rc;
    if (n_sets > 0) {
        const int packed = n > 0 ? PINN_FLAG_WEIGHTS_PACKED : 0;      // (an empty collocation batch packed nothing)
        rc = pinn_data_loss_grad_multi(params_flat, layers, n_layers, sets, n_sets, lb, ub, normalize, grad_flat_out, 1, precision_mode | packed, workspace,
                                       ws_bytes, stream);
        if (rc) return rc;
    }
    if (adam) {
        NetDesc net;
        int width = 0;
        if ((rc = decode_net(layers, n_layers, net, width, 3))) return rc;
        return pinn_adam_step(params_flat, adam->m, adam->v, grad_flat_out, net.nparams, adam->lr, adam->beta1, adam->beta2, adam->eps, adam->step, stream);
    }
    return PINN_OK;
}

int pinn_wave2d_fields(const float* params_flat, const int* layers, int n_layers, const float* x, const float* y, const float* t,
                       int64_t n, const double lb[3], const double ub[3], int normalize, float* fields_out, int precision_mode,
                       void* workspace, size_t ws_bytes, void* stream) {
    Call c;
    const Impl* impl = nullptr;
    int rc = prepare(params_flat, layers, n_layers, x, y, t, n, lb, ub, normalize, precision_mode, workspace, ws_bytes, stream, c, impl);
    if (rc) return rc;
    if (n > 0 && !fields_out) return PINN_ERR_NULL;
    c.fields_out = fields_out;
    if (n == 0) return 0;
    if (!impl) return fp32_call(c, HEAD_FIELDS, 0, 4);
    return impl->fields(c);
}

int pinn_net_streams(const float* params_flat, const int* layers, int n_layers, const float* x, const float* y, const float* t, int64_t n,
                     const double lb[3], const double ub[3], int normalize, float* streams_out, int precision_mode, void* workspace,
                     size_t ws_bytes, void* stream) {
    Call c;
    const Impl* impl = nullptr;
    int rc = prepare(params_flat, layers, n_layers, x, y, t, n, lb, ub, normalize, precision_mode, workspace, ws_bytes, stream, c, impl);
    if (rc) return rc;
    if (n > 0 && !streams_out) return PINN_ERR_NULL;
    c.fields_out = streams_out;
    if (n == 0) return 0;
    if (!impl) return fp32_call(c, HEAD_FIELDS, 0, 5);
    return impl->streams(c);
}

int pinn_plate2d_loss_grad(const float* params_flat, const int* layers, int n_layers, const float* x, const float* y, const float* t,
                           int64_t n, const double lb[3], const double ub[3], int normalize, const float* frozen_streams, double E,
                           double mu, double rho, const float term_weights[5], float* loss_terms_out, float* grad_flat_out,
                           int accumulate, int precision_mode, void* workspace, size_t ws_bytes, void* stream) {
    Call c;
    const Impl* impl = nullptr;
    int rc = prepare(params_flat, layers, n_layers, x, y, t, n, lb, ub, normalize, precision_mode, workspace, ws_bytes, stream, c, impl);
    if (rc) return rc;
    if ((n > 0 && !frozen_streams) || !term_weights || !loss_terms_out || !grad_flat_out) return PINN_ERR_NULL;
    if (c.net.nout != 5) return PINN_ERR_LAYERS;
    c.c1 = (float)(E / (1.0 - mu * mu));               // plane stress, PLATE:416-418
    c.c2 = (float)(E * mu / (1.0 - mu * mu));
    c.G = (float)(E / (2.0 * (1.0 + mu)));
    c.rho = (float)rho;
    for (int i = 0; i < 5; ++i) c.tw[i] = term_weights[i];
    c.aux = frozen_streams;
    c.loss_out = loss_terms_out;
    c.grad_out = grad_flat_out;
    c.accumulate = accumulate;
    if (n == 0) return empty_batch(c, 5);
    if (!impl) return fp32_call(c, HEAD_PLATE, 5, 5);
    return impl->plate_loss_grad(c);
}

int pinn_plate2d_traction_loss_grad(const float* params_flat, const int* layers, int n_layers, const float* x, const float* y,
                                    const float* t, int64_t n, const double lb[3], const double ub[3], int normalize,
                                    const float* frozen_and_normals, const float weights[2], float* loss_terms_out,
                                    float* grad_flat_out, int accumulate, int precision_mode, void* workspace, size_t ws_bytes,
                                    void* stream) {
    Call c;
    const Impl* impl = nullptr;
    int rc = prepare(params_flat, layers, n_layers, x, y, t, n, lb, ub, normalize, precision_mode, workspace, ws_bytes, stream, c, impl);
    if (rc) return rc;
    if ((n > 0 && !frozen_and_normals) || !weights || !loss_terms_out || !grad_flat_out) return PINN_ERR_NULL;
    if (c.net.nout != 5) return PINN_ERR_LAYERS;
    c.tw[0] = weights[0];
    c.tw[1] = weights[1];
    c.aux = frozen_and_normals;
    c.loss_out = loss_terms_out;
    c.grad_out = grad_flat_out;
    c.accumulate = accumulate;
    if (n == 0) return empty_batch(c, 2);
    if (!impl) return fp32_call(c, HEAD_TRACTION, 2, 1);
    return impl->traction_loss_grad(c);
}

int pinn_plate2d_step(float* params_flat, const int* layers, int n_layers, const float* x, const float* y, const float* t, int64_t n,
                      const double lb[3], const double ub[3], int normalize, const float* frozen_streams, double E, double mu, double rho,
                      const float term_weights[5], float* loss_terms_out, const float* hole_x, const float* hole_y, const float* hole_t, int64_t hole_n,
                      const float* hole_frozen_and_normals, const float hole_weights[2], float* hole_loss_terms_out, float* grad_flat_out, int accumulate,
                      const pinn_adam_state* adam, int precision_mode, void* workspace, size_t ws_bytes, void* stream) {
    if (hole_n < 0) return PINN_ERR_SIZE;
    if (adam && (!adam->m || !adam->v || adam->step < 1)) return adam->step < 1 ? PINN_ERR_SIZE : PINN_ERR_NULL;
    if (hole_n > 0 && (!hole_x || !hole_y || !hole_t || !hole_frozen_and_normals || !hole_weights || !hole_loss_terms_out)) return PINN_ERR_NULL;
    // ---- the one-launch form (fused_step_kernel<..., NSC = 5>): the five-stream collocation set and the hole-traction set
    if (n > 0 && hole_n > 0 && frozen_streams && term_weights && loss_terms_out && grad_flat_out) {
        Call c;
        const Impl* impl = nullptr;
        int rc = prepare(params_flat, layers, n_layers, x, y, t, n, lb, ub, normalize, precision_mode, workspace, ws_bytes, stream, c, impl);
        if (rc) return rc;
        if (c.net.nout != 5) return PINN_ERR_LAYERS;
        if (impl) {
            c.c1 = (float)(E / (1.0 - mu * mu));               // plane stress, PLATE:416-418
            c.c2 = (float)(E * mu / (1.0 - mu * mu));
            c.G = (float)(E / (2.0 * (1.0 + mu)));
            c.rho = (float)rho;
            for (int i = 0; i < 5; ++i) c.tw[i] = term_weights[i];
            c.aux = frozen_streams;
            c.loss_out = loss_terms_out;
            c.grad_out = grad_flat_out;
            c.accumulate = accumulate;
            Call d = c;                                        // the traction set as a one-set call of the one-stream kernel (head kind 1)
            d.x = hole_x;
            d.y = hole_y;
            d.t = hole_t;
            d.n = (long)hole_n;
            for (int i = 0; i < 16; ++i) d.tw[i] = 0.0f;
            d.tw[0] = hole_weights[0];
            d.tw[1] = hole_weights[1];
            d.aux = hole_frozen_and_normals;
            d.loss_out = hole_loss_terms_out;
            d.one_stream_head = 1;
            d.nsets = 0;
            AdamEpilogue ep = {nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, 0.f};
            if (adam) ep = AdamEpilogue{params_flat, adam->m, adam->v, adam_lr_t(adam->lr, adam->beta1, adam->beta2, adam->step), (float)adam->beta1, (float)adam->beta2, (float)adam->eps};
            if (impl->plate_step(c, d, ep, &rc)) return rc;
        }
    }
    // ---- every other case: the calls one after the other, the same bits
    int rc = pinn_plate2d_loss_grad(params_flat, layers, n_layers, x, y, t, n, lb, ub, normalize, frozen_streams, E, mu, rho, term_weights, loss_terms_out,
                                    grad_flat_out, accumulate, precision_mode, workspace, ws_bytes, stream);
    if (rc) return rc;
    if (hole_loss_terms_out && hole_weights) {             // (an empty hole set reports zeros; no hole set at all: NULL outputs)
        const int packed = n > 0 ? PINN_FLAG_WEIGHTS_PACKED : 0;
        rc = pinn_plate2d_traction_loss_grad(params_flat, layers, n_layers, hole_x, hole_y, hole_t, hole_n, lb, ub, normalize, hole_frozen_and_normals,
                                             hole_weights, hole_loss_terms_out, grad_flat_out, 1, precision_mode | packed, workspace, ws_bytes, stream);
        if (rc) return rc;
    }
    if (adam) {
        NetDesc net;
        int width = 0;
        if ((rc = decode_net(layers, n_layers, net, width, 3))) return rc;
        return pinn_adam_step(params_flat, adam->m, adam->v, grad_flat_out, net.nparams, adam->lr, adam->beta1, adam->beta2, adam->eps, adam->step, stream);
    }
    return PINN_OK;
}

int pinn_stream_loss_grad(const float* params_flat, const int* layers, int n_layers, const float* x, const float* y, const float* t,
                          int64_t n, const double lb[3], const double ub[3], int normalize, const float* targets,
                          const float* weights, float* loss_terms_out, float* grad_flat_out, int accumulate, int precision_mode,
                          void* workspace, size_t ws_bytes, void* stream) {
    Call c;
    const Impl* impl = nullptr;
    int rc = prepare(params_flat, layers, n_layers, x, y, t, n, lb, ub, normalize, precision_mode, workspace, ws_bytes, stream, c, impl);
    if (rc) return rc;
    if (!weights || !loss_terms_out || !grad_flat_out) return PINN_ERR_NULL;
    for (int s = 0; s < 5; ++s)
        for (int o = 0; o < c.net.nout; ++o) c.w5[s][o] = weights[s * c.net.nout + o];
    c.aux = targets;
    c.loss_out = loss_terms_out;
    c.grad_out = grad_flat_out;
    c.accumulate = accumulate;
    if (n == 0) return empty_batch(c, c.net.nout);
    if (!impl) return fp32_call(c, HEAD_STREAMS, c.net.nout, 5);
    return impl->stream_loss_grad(c);
}

// ---- 4-input family: the 3-D Navier-Cauchy extension (BASELINE.json configs[4]; not in the reference -- oracle/nc3d_oracle.py) ----
int pinn_nc3d_loss_grad(const float* params_flat, const int* layers, int n_layers, const float* x, const float* y, const float* z,
                        const float* t, int64_t n, const double lb[4], const double ub[4], int normalize, double E, double mu, double rho,
                        const float term_weights[12], float* loss_terms_out, float* grad_flat_out, int accumulate, int precision_mode,
                        void* workspace, size_t ws_bytes, void* stream) {
    Call c;
    const Impl* impl = nullptr;
    int rc = prepare(params_flat, layers, n_layers, x, y, t, n, lb, ub, normalize, precision_mode, workspace, ws_bytes, stream, c, impl, 4, z);
    if (rc) return rc;
    if (!term_weights || !loss_terms_out || !grad_flat_out) return PINN_ERR_NULL;
    if (c.net.nout != 12) return PINN_ERR_LAYERS;
    const double coef = E / ((1.0 + mu) * (1.0 - 2.0 * mu));       // isotropic law: c1 = lambda + 2G, c2 = lambda
    c.c1 = (float)(coef * (1.0 - mu));
    c.c2 = (float)(coef * mu);
    c.G = (float)(E / (2.0 * (1.0 + mu)));
    c.rho = (float)rho;
    for (int i = 0; i < 12; ++i) c.tw[i] = term_weights[i];
    c.loss_out = loss_terms_out;
    c.grad_out = grad_flat_out;
    c.accumulate = accumulate;
    if (n == 0) return empty_batch(c, 12);
    if (!impl) return fp32_call(c, HEAD_NC3D, 12, 5, 4);
    return impl->nc3d_loss_grad(c);
}

int pinn_nc3d_data_loss_grad(const float* params_flat, const int* layers, int n_layers, const float* x, const float* y, const float* z,
                             const float* t, int64_t n, const double lb[4], const double ub[4], int normalize, const float* targets,
                             const float* out_weights, float* loss_terms_out, float* grad_flat_out, int accumulate, int precision_mode,
                             void* workspace, size_t ws_bytes, void* stream) {
    Call c;
    const Impl* impl = nullptr;
    int rc = prepare(params_flat, layers, n_layers, x, y, t, n, lb, ub, normalize, precision_mode, workspace, ws_bytes, stream, c, impl, 4, z);
    if (rc) return rc;
    if (!out_weights || !loss_terms_out || !grad_flat_out) return PINN_ERR_NULL;
    for (int i = 0; i < c.net.nout; ++i) c.tw[i] = out_weights[i];
    c.targets = targets;
    c.loss_out = loss_terms_out;
    c.grad_out = grad_flat_out;
    c.accumulate = accumulate;
    if (n == 0) return empty_batch(c, c.net.nout);
    if (!impl) return fp32_call(c, HEAD_DATA3D, c.net.nout, 1, 4);
    return impl->nc3d_data_loss_grad(c);
}

int pinn_nc3d_fields(const float* params_flat, const int* layers, int n_layers, const float* x, const float* y, const float* z,
                     const float* t, int64_t n, const double lb[4], const double ub[4], int normalize, float* fields_out, int precision_mode,
                     void* workspace, size_t ws_bytes, void* stream) {
    Call c;
    const Impl* impl = nullptr;
    int rc = prepare(params_flat, layers, n_layers, x, y, t, n, lb, ub, normalize, precision_mode, workspace, ws_bytes, stream, c, impl, 4, z);
    if (rc) return rc;
    if (n > 0 && !fields_out) return PINN_ERR_NULL;
    c.fields_out = fields_out;
    if (n == 0) return 0;
    if (!impl) return fp32_call(c, HEAD_FIELDS3D, 0, 5, 4);
    return impl->nc3d_fields(c);
}

int pinn_adam_step(float* params_flat, float* m, float* v, const float* grad_flat, int64_t n_params, double lr, double beta1,
                   double beta2, double eps, int64_t step, void* stream) {
    if (!params_flat || !m || !v || !grad_flat) return PINN_ERR_NULL;
    if (n_params <= 0 || step < 1) return PINN_ERR_SIZE;
    // lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t)   (TF1 AdamOptimizer)
    const double b1t = __builtin_pow(beta1, (double)step), b2t = __builtin_pow(beta2, (double)step);
    const double lr_t = lr * __builtin_sqrt(1.0 - b2t) / (1.0 - b1t);
    hipLaunchKernelGGL((adam_tf1_kernel<0>), dim3((unsigned)((n_params + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params_flat, m, v,
                       grad_flat, (long)n_params, (float)lr_t, (float)beta1, (float)beta2, (float)eps);
    return (int)hipGetLastError();
}


// ---- the finite-gradient ladder for callers that are not Python (round 6; elastic_wave.py: evaluate_with_finite_gradient) ----------------
namespace {
__global__ __launch_bounds__(256) void probe_ranges_kernel(const float* params, const float* grad, long n, unsigned* out /* {non-finite count (saturating flag), bits of max |w|} */) {
    unsigned bad = 0u, wmax = 0u;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        if (grad != nullptr) {
            const unsigned g = __builtin_bit_cast(unsigned, grad[i]);
            if ((g & 0x7f800000u) == 0x7f800000u) bad = 1u;              // Inf or NaN
        }
        if (params != nullptr) {
            const unsigned w = __builtin_bit_cast(unsigned, params[i]) & 0x7fffffffu;      // |w|: non-negative floats order like their bits (NaN sorts above Inf)
            wmax = w > wmax ? w : wmax;
        }
    }
    if (bad) atomicOr(&out[0], 1u);
    if (wmax) atomicMax(&out[1], wmax);
}
}  // namespace

int pinn_probe_ranges(const float* params_flat, const float* grad_flat, int64_t n_params, void* workspace, size_t ws_bytes, void* stream,
                      int* grad_finite_out, float* max_abs_weight_out) {
    if (!workspace || (!params_flat && !grad_flat)) return PINN_ERR_NULL;
    if (n_params < 1) return PINN_ERR_SIZE;
    if (ws_bytes < 256) return PINN_ERR_WORKSPACE;
    // two words at the very end of the workspace: behind a finished call that region is dead data (scratch images / panels are written before
    // they are read in every call), and nothing a later PINN_FLAG_WEIGHTS_PACKED call relies on lives there
    unsigned* out = reinterpret_cast<unsigned*>(static_cast<char*>(workspace) + ((ws_bytes - 16) & ~(size_t)15));
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc = (int)hipMemsetAsync(out, 0, 8, st);
    if (rc) return rc;
    const long blocks = (n_params + 255) / 256;
    hipLaunchKernelGGL(probe_ranges_kernel, dim3((unsigned)(blocks < 256 ? blocks : 256)), dim3(256), 0, st, params_flat, grad_flat, (long)n_params, out);
    if ((rc = (int)hipGetLastError())) return rc;
    unsigned host[2] = {0u, 0u};
    if ((rc = (int)hipMemcpyAsync(host, out, 8, hipMemcpyDeviceToHost, st))) return rc;
    if ((rc = (int)hipStreamSynchronize(st))) return rc;
    if (grad_finite_out) *grad_finite_out = host[0] ? 0 : 1;
    if (max_abs_weight_out) *max_abs_weight_out = __builtin_bit_cast(float, host[1]);
    return PINN_OK;
}

int pinn_wave2d_loss_grad_checked(const float* params_flat, const int* layers, int n_layers, const float* x, const float* y, const float* t,
                                  int64_t n, const double lb[3], const double ub[3], int normalize, double E, double mu, double rho,
                                  int plane_strain, const float term_weights[7], float* loss_terms_out, float* grad_flat_out,
                                  int precision_mode, void* workspace, size_t ws_bytes, void* stream, pinn_range_state* state) {
    if (!state) return PINN_ERR_NULL;
    if (state->adjoint_shift < 0 || state->adjoint_shift > 24) return PINN_ERR_SIZE;
    if (n_layers < 2 || !layers) return PINN_ERR_LAYERS;
    long np_ = 0;
    for (int i = 0; i + 1 < n_layers; ++i) np_ += (long)layers[i] * layers[i + 1] + layers[i + 1];
    // the caller's mode word without the two things the ladder owns
    const int base = precision_mode & ~(PINN_FLAG_TWO_KERNEL | PINN_ADJOINT_SHIFT(31));
    const bool f16x3 = (base & 0xff) == PINN_PREC_F16X3;
    state->attempts = 0;
    for (int attempt = 0; attempt < 9; ++attempt) {
        const int mode = base | (state->two_kernel ? PINN_FLAG_TWO_KERNEL : 0) | PINN_ADJOINT_SHIFT(state->adjoint_shift);
        int rc = pinn_wave2d_loss_grad(params_flat, layers, n_layers, x, y, t, n, lb, ub, normalize, E, mu, rho, plane_strain, term_weights, loss_terms_out,
                                       grad_flat_out, 0, attempt == 0 ? mode : (mode & ~PINN_FLAG_WEIGHTS_PACKED), workspace, ws_bytes, stream);
        ++state->attempts;
        if (rc) return rc;
        int finite = 1;
        float wmax = 0.0f;
        if ((rc = pinn_probe_ranges(params_flat, grad_flat_out, np_, workspace, ws_bytes, stream, &finite, &wmax))) return rc;
        if (finite) return PINN_OK;
        // 1. a weight beyond the fused kernels' format poisons the whole result with NaN: leave the fused path for good and repeat
        if (f16x3 && !state->two_kernel && !(wmax <= pinn_fused_weight_limit())) {
            state->two_kernel = 1;
            continue;
        }
        // 2. the 16-bit reverse pass overflowed (residuals orders of magnitude above their trained size): scale it by another 2^-4
        if (state->adjoint_shift >= 24) return PINN_ERR_RANGE;
        state->adjoint_shift = state->adjoint_shift + 4 > 24 ? 24 : state->adjoint_shift + 4;
    }
    return PINN_ERR_RANGE;
}

}  // extern "C"
