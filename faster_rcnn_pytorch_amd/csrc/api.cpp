// api.cpp -- host-only part of libfrcnn_hip.so: status strings, workspace sizing,
// host-side anchor bases, and the per-kernel HIP-event timing facility.
#include "frcnn_common.h"
#include "frcnn_layout.h"
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cctype>
#include <mutex>
#include <string>
#include <vector>

static thread_local char g_err[512] = "";

int frcnn_set_error(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// ---- layout stamps (frcnn_layout.h): every object of the library registers the stamp it was compiled with ----
#define FRCNN_N_OBJECTS 15             // the fourteen .hip objects + api.o (csrc/Makefile: SRCS_HIP)
struct LayoutReg { const char *object; uint64_t stamp; };
static std::vector<LayoutReg> &layout_registry() { static std::vector<LayoutReg> r; return r; }     // function-local: static initialisers of other objects may run first
void frcnn_layout_register(const char *object, uint64_t stamp) { layout_registry().push_back({object, stamp}); }
FRCNN_LAYOUT_STAMP(api);

int frcnn_layout_check_impl(void)
{
    const uint64_t mine = frcnn_layout_stamp_value();
    const std::vector<LayoutReg> &r = layout_registry();
    for (const LayoutReg &e : r)
        if (e.stamp != mine)
            return frcnn_set_error(FRCNN_ERR_UNSUPPORTED, "layout stamp mismatch: object '%s' was compiled against other versions of the shared headers "
                                   "(%016llx, api.o has %016llx): stale object -- rebuild the library from clean (make -C faster_rcnn_pytorch_amd/csrc clean all)",
                                   e.object, (unsigned long long)e.stamp, (unsigned long long)mine);
    if ((int)r.size() != FRCNN_N_OBJECTS)
        return frcnn_set_error(FRCNN_ERR_UNSUPPORTED, "layout stamp: %d objects registered, %d expected (an object was linked without its stamp)",
                               (int)r.size(), FRCNN_N_OBJECTS);
    return FRCNN_OK;
}
FRCNN_EXPORT int frcnn_layout_check(void) { return frcnn_layout_check_impl(); }
FRCNN_EXPORT uint64_t frcnn_layout_stamp(void) { return frcnn_layout_stamp_value(); }

// the ABI version, or FRCNN_ERR_UNSUPPORTED (negative; message in frcnn_last_error) when the library's objects disagree about a layout
FRCNN_EXPORT int frcnn_abi_version(void) { const int rc = frcnn_layout_check_impl(); return rc ? rc : FRCNN_ABI_VERSION; }
FRCNN_EXPORT const char *frcnn_last_error(void) { return g_err; }

// ---- workspace layout sizes (must agree with the carving in topk.hip / nms.hip / targets.hip) ----
size_t frcnn_ws_topk(int64_t N);
size_t frcnn_ws_nms(int64_t K);
size_t frcnn_ws_rpn_targets(int64_t N, int64_t G);
size_t frcnn_ws_head_targets(int64_t n);
size_t frcnn_ws_region_proposal(int64_t N, int64_t K, int64_t P);
size_t frcnn_ws_preprocess(int64_t in_hw, int64_t out_hw);
size_t frcnn_ws_head_bwd(int64_t C);
size_t frcnn_ws_rpn_conv(void);
size_t frcnn_ws_rpn_conv_wgrad(void);
size_t frcnn_ws_rpn_conv_f32(int64_t C);

FRCNN_EXPORT size_t frcnn_workspace_bytes(int op, int64_t n1, int64_t n2)
{
    if (n1 < 0 || n2 < 0) return 0;
    switch (op) {
    case FRCNN_OP_TOPK: return frcnn_ws_topk(n1);
    case FRCNN_OP_NMS: return frcnn_ws_nms(n1);
    case FRCNN_OP_REGION_PROPOSAL: { const int64_t k = n2 < n1 ? n2 : n1; return frcnn_ws_region_proposal(n1, k, k); }
    case FRCNN_OP_RPN_TARGETS: return frcnn_ws_rpn_targets(n1, n2);
    case FRCNN_OP_HEAD_TARGETS: return frcnn_ws_head_targets(n1);
    case FRCNN_OP_PREPROCESS: return frcnn_ws_preprocess(n1, n2);
    case FRCNN_OP_HEAD_BWD: return frcnn_ws_head_bwd(n1);
    case FRCNN_OP_RPN_CONV: return frcnn_ws_rpn_conv();
    case FRCNN_OP_RPN_CONV_WGRAD: return frcnn_ws_rpn_conv_wgrad();
    case FRCNN_OP_RPN_CONV_F32: return frcnn_ws_rpn_conv_f32(n1);
    default: return 0;
    }
}

// ---- A1: FRCNNAnchorMaker.generate_anchor_base (anchor.py:15-32): float64 math, float32 store ----
FRCNN_EXPORT int frcnn_anchor_base_host(int base_size, const double *ratios, int n_ratios,
                                        const double *scales, int n_scales, float *out)
{
    FRCNN_REQUIRE(ratios && scales && out && n_ratios > 0 && n_scales > 0 && base_size > 0, "anchor_base: bad argument");
    const double px = base_size / 2.0, py = base_size / 2.0;
    for (int i = 0; i < n_ratios; ++i)
        for (int j = 0; j < n_scales; ++j) {
            const double w = base_size * scales[j] * std::sqrt(ratios[i]);
            const double h = base_size * scales[j] * std::sqrt(1.0 / ratios[i]);
            float *o = out + 4 * (i * n_scales + j);
            o[0] = (float)(px - w / 2.0);
            o[1] = (float)(py - h / 2.0);
            o[2] = (float)(px + w / 2.0);
            o[3] = (float)(py + h / 2.0);
        }
    return FRCNN_OK;
}

// ---- B-AG: torchvision AnchorGenerator.generate_anchors, one size (models/new_model.py:23-25) ----
FRCNN_EXPORT int frcnn_tv_base_anchors_host(float size, const float *ratios, int n_ratios, float *out)
{
    FRCNN_REQUIRE(ratios && out && n_ratios > 0, "tv_base_anchors: bad argument");
    for (int i = 0; i < n_ratios; ++i) {
        const float hr = std::sqrt(ratios[i]);
        const float wr = 1.0f / hr;
        const float ws = wr * size, hs = hr * size;
        out[4 * i + 0] = std::nearbyint(-ws / 2.0f);
        out[4 * i + 1] = std::nearbyint(-hs / 2.0f);
        out[4 * i + 2] = std::nearbyint(ws / 2.0f);
        out[4 * i + 3] = std::nearbyint(hs / 2.0f);
    }
    return FRCNN_OK;
}

// ---- per-kernel timing with HIP events on the launch stream ----
#define KID_COUNT FRCNN_PROF_MAX_KERNELS
struct ProfRec { int kid; hipEvent_t a, b; };
static std::mutex g_prof_mu;
static bool g_prof_enabled = false;
static std::vector<ProfRec> g_prof_pending;
static std::vector<hipEvent_t> g_prof_free;
static double g_prof_ms[KID_COUNT];
static int64_t g_prof_n[KID_COUNT];
static std::vector<float> g_prof_samples[KID_COUNT];      // per-launch durations (ms): median / percentiles for bench.py
static std::string g_kernel_names[KID_COUNT];
static int g_n_kernels = 0;

// "(roi_pool_fwd_lds_kernel<7, 7, AT>)" -> "roi_pool_fwd_lds_kernel": the name rocprofv3 prints without its template arguments
int frcnn_prof_register(const char *expr)
{
    std::string n(expr ? expr : "");
    size_t b = 0;
    while (b < n.size() && (n[b] == '(' || n[b] == ' ')) ++b;
    size_t e = b;
    while (e < n.size() && (isalnum((unsigned char)n[e]) || n[e] == '_')) ++e;
    n = n.substr(b, e - b);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int i = 0; i < g_n_kernels; ++i)
        if (g_kernel_names[i] == n) return i;
    if (g_n_kernels >= KID_COUNT) return KID_COUNT - 1;
    g_kernel_names[g_n_kernels] = n;
    return g_n_kernels++;
}

bool frcnn_prof_on() { return g_prof_enabled; }

static hipEvent_t prof_get_event()
{
    if (!g_prof_free.empty()) { hipEvent_t e = g_prof_free.back(); g_prof_free.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

void frcnn_prof_begin(int kid, hipStream_t s)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r{kid, prof_get_event(), prof_get_event()};
    (void)hipEventRecord(r.a, s);
    g_prof_pending.push_back(r);
}

void frcnn_prof_end(int kid, hipStream_t s)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (size_t i = g_prof_pending.size(); i-- > 0;)
        if (g_prof_pending[i].kid == kid) { (void)hipEventRecord(g_prof_pending[i].b, s); break; }
}

FRCNN_EXPORT int frcnn_prof_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_enabled = on != 0;
    return FRCNN_OK;
}

FRCNN_EXPORT int frcnn_prof_collect(void)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto &r : g_prof_pending) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            g_prof_ms[r.kid] += ms;
            g_prof_n[r.kid] += 1;
            if (g_prof_samples[r.kid].size() < (size_t)1 << 20) g_prof_samples[r.kid].push_back(ms);
        }
        g_prof_free.push_back(r.a);
        g_prof_free.push_back(r.b);
    }
    g_prof_pending.clear();
    return FRCNN_OK;
}

FRCNN_EXPORT int frcnn_prof_reset(void)
{
    frcnn_prof_collect();
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int i = 0; i < KID_COUNT; ++i) { g_prof_ms[i] = 0.0; g_prof_n[i] = 0; g_prof_samples[i].clear(); }
    return FRCNN_OK;
}

FRCNN_EXPORT int frcnn_prof_num_kernels(void) { std::lock_guard<std::mutex> lk(g_prof_mu); return g_n_kernels; }
FRCNN_EXPORT const char *frcnn_prof_kernel_name(int kid) { return (kid >= 0 && kid < KID_COUNT) ? g_kernel_names[kid].c_str() : ""; }
FRCNN_EXPORT int frcnn_prof_get(int kid, double *total_ms, int64_t *launches)
{
    FRCNN_REQUIRE(kid >= 0 && kid < KID_COUNT && total_ms && launches, "prof_get: bad argument");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    *total_ms = g_prof_ms[kid];
    *launches = g_prof_n[kid];
    return FRCNN_OK;
}

// per-launch samples of one kernel id, in launch order: copies min(cap, n) values (ms) and returns n (negative = error)
FRCNN_EXPORT int64_t frcnn_prof_get_samples(int kid, float *out_ms, int64_t cap)
{
    if (kid < 0 || kid >= KID_COUNT || cap < 0 || (cap > 0 && !out_ms)) return frcnn_set_error(FRCNN_ERR_INVALID_ARG, "prof_get_samples: bad argument");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    const std::vector<float> &v = g_prof_samples[kid];
    const int64_t n = (int64_t)v.size();
    for (int64_t i = 0; i < n && i < cap; ++i) out_ms[i] = v[(size_t)i];
    return n;
}
