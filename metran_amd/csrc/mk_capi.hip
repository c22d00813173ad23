// mk_capi.hip -- C ABI of libmetran_hip.so (declared in include/metran_hip.h).
// Thin, exception-free layer: argument validation, kernel dispatch by (N,K), HIP error ->
// mk_status translation, optional hipEvent timing of the two hot kernels.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <vector>
#include <new>

#include "mk_generic.h"
#include "mk_internal.h"
#include "mk_lbfgs.h"

struct mk_context {
    int device;
    hipStream_t stream;
    int timing;       // 0 off, 1 the most recent launch of each kind, 2 every launch until mk_kernel_ms_totals
    hipEvent_t ev[4]; // filter start/stop, smoother start/stop (mode 1)
    bool have_filter_time, have_smooth_time;
    std::vector<hipEvent_t> ev_pool;                              // mode 2: recycled events
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pending[2]; // mode 2: one (start, stop) pair per launch; 0 filter, 1 smoother
    int *tlist;      // workspace of the sparse objective (observed-step list), grown on demand
    long tlist_cap;
    // the list is rebuilt only when the record it was built from changes: same pointer, shape and layout, and no
    // mk_observations_changed() since (the solver evaluates the objective ~80 times on one uploaded record)
    const double *tlist_obs;
    long tlist_T, tlist_N, tlist_ostep;
    int variant[MK_VARIANT_COUNT]; // mk_set_kernel_variant: which of two equivalent (tested) kernels serves a shape class
    double *gws;       // workspace of the size-generic smoother (mk_generic.hip), grown on demand
    size_t gws_cap;    // ... in doubles
    int *lb_counters;  // device int[4] of the L-BFGS kernels (mk_lbfgs.hip)
    double *adj_upd;   // update tape of the wide adjoint gradient (mk_set_adjoint_updates; caller-owned), nullptr = recompute
    size_t adj_upd_cap; // ... its capacity in doubles
    double *cur_upd;   // the tape of the recording forward pass in progress (set around do_filter by mk_loglik_grad_phases)
    void *comm;        // RCCL communicator of mk_allreduce_sum (ncclComm_t), nullptr = none
    bool comm_owned;   // created by mk_comm_init_rank (destroyed with the context) or handed in by mk_set_communicator
};

static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define MK_HIP(call)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) return fail(MK_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

#define MK_CTX(ctx)                                              \
    if (!(ctx)) return fail(MK_ERR_INVALID, "null mk_context"); \
    MK_HIP(hipSetDevice((ctx)->device))

// ---- hipEvent timing of the hot kernels: kind 0 = filter / objective, 1 = smoother ----
static hipError_t timing_start(mk_context *ctx, int kind)
{
    if (!ctx->timing) return hipSuccess;
    if (ctx->timing == 2) {
        hipEvent_t e[2];
        for (auto &x : e) {
            if (!ctx->ev_pool.empty()) {
                x = ctx->ev_pool.back();
                ctx->ev_pool.pop_back();
            } else {
                const hipError_t err = hipEventCreate(&x);
                if (err != hipSuccess) return err;
            }
        }
        ctx->ev_pending[kind].push_back({e[0], e[1]});
        return hipEventRecord(e[0], ctx->stream);
    }
    return hipEventRecord(ctx->ev[2 * kind], ctx->stream);
}
static hipError_t timing_stop(mk_context *ctx, int kind)
{
    if (!ctx->timing) return hipSuccess;
    if (ctx->timing == 2) return hipEventRecord(ctx->ev_pending[kind].back().second, ctx->stream);
    (kind ? ctx->have_smooth_time : ctx->have_filter_time) = true;
    return hipEventRecord(ctx->ev[2 * kind + 1], ctx->stream);
}

// ---- run-time shape modules (see the MK_SHAPE_MODULE block of mk_kernels.hip and metran_amd/jit.py) ----
namespace {
struct ShapeModule {
    int N, K;
    void *handle;
    int (*launch_filter)(const mk::FilterArgs *, void *);
    int (*launch_smoother)(const mk::SmootherArgs *, void *);
    int (*launch_adjoint)(const mk::AdjointArgs *, void *);
    int (*launch_sparse)(const mk::SparseArgs *, void *);
};
std::vector<ShapeModule> g_modules;
std::mutex g_modules_mutex;

const ShapeModule *find_module(int64_t N, int64_t K, bool by_n_only = false)
{
    std::lock_guard<std::mutex> lock(g_modules_mutex);
    for (const auto &m : g_modules)
        if ((m.N == N && m.K == K) || (by_n_only && m.N + m.K == N + K)) return &m;
    return nullptr;
}
bool aot_shape(int64_t N, int64_t K)
{
    for (int i = 0; i < mk::num_shapes(); ++i) {
        int n_, k_;
        mk::get_shape(i, &n_, &k_);
        if (n_ == N && k_ == K) return true;
    }
    return false;
}
bool aot_state_dim(int64_t n)
{
    for (int i = 0; i < mk::num_shapes(); ++i) {
        int n_, k_;
        mk::get_shape(i, &n_, &k_);
        if (n_ + k_ == n) return true;
    }
    return false;
}
bool specialised(int64_t N, int64_t K) { return aot_shape(N, K) || find_module(N, K) != nullptr; }
bool generic_shape(int64_t N, int64_t K) { return N >= 1 && K >= 1 && N + K <= MK_GENERIC_MAX_STATES; }
hipError_t dispatch_filter(int N, int K, const mk::FilterArgs &a, hipStream_t s, bool force_generic = false)
{
    if (force_generic) return generic_shape(N, K) ? mk::launch_filter_generic(N, K, a, s) : hipErrorInvalidValue;
    if (aot_shape(N, K)) return mk::launch_filter(N, K, a, s);
    if (const ShapeModule *m = find_module(N, K)) return (hipError_t)m->launch_filter(&a, (void *)s);
    if (generic_shape(N, K)) return mk::launch_filter_generic(N, K, a, s); // any shape, not specialised (mk_generic.hip)
    return hipErrorInvalidValue;
}
hipError_t dispatch_sparse(int N, int K, const mk::SparseArgs &a, hipStream_t s)
{
    if (aot_shape(N, K)) return mk::launch_sparse(N, K, a, s);
    if (const ShapeModule *m = find_module(N, K)) return (hipError_t)m->launch_sparse(&a, (void *)s);
    return hipErrorInvalidValue;
}
hipError_t dispatch_adjoint(int N, int K, const mk::AdjointArgs &a, hipStream_t s)
{
    if (aot_shape(N, K)) return mk::launch_adjoint(N, K, a, s);
    if (const ShapeModule *m = find_module(N, K)) return (hipError_t)m->launch_adjoint(&a, (void *)s);
    return hipErrorInvalidValue;
}
// workspace of the generic smoother: grown on demand, stream-ordered reuse (every launch of a context is on its stream)
hipError_t generic_workspace(mk_context *ctx, long B, int n, double **ws)
{
    const size_t need = mk::generic_smoother_ws_doubles(B, n);
    if (ctx->gws_cap < need) {
        if (ctx->gws) {
            hipError_t e = hipStreamSynchronize(ctx->stream); // a launch still reading the old buffer
            if (e != hipSuccess) return e;
            e = hipFree(ctx->gws);
            if (e != hipSuccess) return e;
        }
        ctx->gws = nullptr;
        ctx->gws_cap = 0;
        const hipError_t e = hipMalloc((void **)&ctx->gws, need * sizeof(double));
        if (e != hipSuccess) return e;
        ctx->gws_cap = need;
    }
    *ws = ctx->gws;
    return hipSuccess;
}
hipError_t dispatch_smoother(mk_context *ctx, int N, int K, const mk::SmootherArgs &a, hipStream_t s)
{
    // the plain smoother depends on n = N + K only; the projecting one and the tape's backward pass (the state tape without a
    // projection included: a (16,2) tape walked by the (17,1) module's kernel was the defect the (16,2) state-tape test found)
    // need the exact (N, K)
    const bool proj = a.sim_means || a.sim_vars || a.tape != 0;
    const bool force_generic = ctx->variant[MK_VARIANT_KERNEL_FAMILY] == 1;
    if (!force_generic) {
        if (proj ? aot_shape(N, K) : aot_state_dim(N + K)) return mk::launch_smoother(N, K, a, s);
        if (const ShapeModule *m = find_module(N, K, !proj)) return (hipError_t)m->launch_smoother(&a, (void *)s);
    }
    if (generic_shape(N, K)) {
        mk::GenericSmootherArgs g;
        g.a = a;
        g.N = N;
        g.K = K;
        g.Xp = g.Pp = nullptr;
        const hipError_t e = generic_workspace(ctx, a.B, N + K, &g.ws);
        if (e != hipSuccess) return e;
        return mk::launch_smoother_generic(g, s);
    }
    return hipErrorInvalidValue;
}
} // namespace

extern "C" {

MK_API int mk_register_shape_module(const char *path)
{
    if (!path) return fail(MK_ERR_INVALID, "null module path");
    void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(MK_ERR_INVALID, "dlopen(%s): %s", path, dlerror());
    auto abi = (int (*)(void))dlsym(h, "mkmod_abi");
    auto shape = (int (*)(int *, int *))dlsym(h, "mkmod_shape");
    auto lf = (int (*)(const mk::FilterArgs *, void *))dlsym(h, "mkmod_launch_filter");
    auto ls = (int (*)(const mk::SmootherArgs *, void *))dlsym(h, "mkmod_launch_smoother");
    auto la = (int (*)(const mk::AdjointArgs *, void *))dlsym(h, "mkmod_launch_adjoint");
    auto lsp = (int (*)(const mk::SparseArgs *, void *))dlsym(h, "mkmod_launch_sparse");
    if (!abi || !shape || !lf || !ls || !la || !lsp) {
        dlclose(h);
        return fail(MK_ERR_INVALID, "%s is not a metran_hip shape module", path);
    }
    if (abi() != (int)(sizeof(mk::FilterArgs) * 1000 + sizeof(mk::SmootherArgs) + sizeof(mk::AdjointArgs) +
                       sizeof(mk::SparseArgs))) {
        dlclose(h);
        return fail(MK_ERR_INVALID, "%s was built against different kernel-argument structs (stale cache)", path);
    }
    int N = 0, K = 0;
    if (shape(&N, &K) != 1) {
        dlclose(h);
        return fail(MK_ERR_INVALID, "%s must contain exactly one (N,K) shape", path);
    }
    std::lock_guard<std::mutex> lock(g_modules_mutex);
    for (const auto &m : g_modules)
        if (m.N == N && m.K == K) {
            dlclose(h);
            return MK_OK; // already registered
        }
    g_modules.push_back(ShapeModule{N, K, h, lf, ls, la, lsp});
    return MK_OK;
}

MK_API int mk_abi_version(void) { return MK_ABI_VERSION; }
MK_API const char *mk_last_error(void) { return g_err; }

MK_API int mk_device_count(int *count)
{
    if (!count) return fail(MK_ERR_INVALID, "null count");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *count = 0;
        return fail(MK_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = c;
    return MK_OK;
}

MK_API int mk_create(int device, mk_context **out)
{
    if (!out) return fail(MK_ERR_INVALID, "null ctx out-pointer");
    *out = nullptr;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess || c <= 0)
        return fail(MK_ERR_NO_DEVICE, "no HIP device visible (libmetran_hip needs an MI355X / gfx950)");
    if (device < 0 || device >= c) return fail(MK_ERR_INVALID, "device %d out of range [0,%d)", device, c);
    MK_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    MK_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(MK_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device,
                    prop.gcnArchName);
    mk_context *ctx = new (std::nothrow) mk_context();
    if (!ctx) return fail(MK_ERR_ALLOC, "out of host memory");
    ctx->device = device;
    ctx->stream = nullptr;
    ctx->timing = 0;
    ctx->have_filter_time = ctx->have_smooth_time = false;
    ctx->tlist = nullptr;
    ctx->tlist_cap = 0;
    ctx->tlist_obs = nullptr;
    ctx->tlist_T = ctx->tlist_N = ctx->tlist_ostep = 0;
    ctx->gws = nullptr;
    ctx->gws_cap = 0;
    ctx->lb_counters = nullptr;
    ctx->comm = nullptr;
    ctx->comm_owned = false;
    ctx->adj_upd = ctx->cur_upd = nullptr;
    ctx->adj_upd_cap = 0;
    for (int &v : ctx->variant) v = 0;
    for (auto &e : ctx->ev) {
        if (hipEventCreate(&e) != hipSuccess) {
            delete ctx;
            return fail(MK_ERR_HIP, "hipEventCreate failed");
        }
    }
    *out = ctx;
    return MK_OK;
}

// ---- the one collective of the design: all-reduce(sum, f64) of the summed objective over the ranks ----
// librccl is bound at first use with dlopen -- the library has no link-time dependency on it, so a single-GPU caller never
// loads it -- preferring a copy the process has already mapped (PyTorch-ROCm ships its own librccl.so: two RCCL runtimes in
// one process would each bootstrap their own network state).  Only the five entry points below are used; their
// signatures are RCCL's public C API (rccl/rccl.h: ncclGetUniqueId :187, ncclCommInitRank :220, ncclCommDestroy :260,
// ncclAllReduce, ncclGetErrorString :339), restated here so that the header is not needed to build.
namespace {
struct RcclUniqueId {
    char internal[128]; // NCCL_UNIQUE_ID_BYTES
};
struct RcclApi {
    void *handle;
    int (*GetUniqueId)(RcclUniqueId *);
    int (*CommInitRank)(void **, int, RcclUniqueId, int);
    int (*CommDestroy)(void *);
    int (*AllReduce)(const void *, void *, size_t, int /*ncclDataType_t*/, int /*ncclRedOp_t*/, void *, hipStream_t);
    const char *(*GetErrorString)(int);
    char why[256];
};
RcclApi g_rccl = {};
char g_rccl_path[1024] = "";
std::mutex g_rccl_mutex;
constexpr int kNcclFloat64 = 8, kNcclSum = 0; // rccl.h: ncclFloat64 = 8, ncclSum = 0

const RcclApi *rccl_api()
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return &g_rccl;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    if (g_rccl_path[0]) h = dlopen(g_rccl_path, RTLD_NOW | RTLD_GLOBAL);
    for (int pass = 0; pass < 2 && !h; ++pass) // pass 0: a copy that is already mapped; pass 1: load one
        for (const char *nm : names) {
            h = dlopen(nm, (pass == 0 ? RTLD_NOLOAD : 0) | RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
    if (!h) {
        snprintf(g_rccl.why, sizeof(g_rccl.why), "librccl.so could not be loaded (%s); name it with mk_comm_set_library", dlerror());
        return nullptr;
    }
    RcclApi a = {};
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
    a.AllReduce = (decltype(a.AllReduce))dlsym(h, "ncclAllReduce");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce || !a.GetErrorString) {
        snprintf(g_rccl.why, sizeof(g_rccl.why), "the loaded librccl lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / "
                                                 "ncclAllReduce / ncclGetErrorString");
        return nullptr;
    }
    a.handle = h;
    g_rccl = a;
    return &g_rccl;
}
} // namespace

#define MK_RCCL(api, call)                                                                          \
    do {                                                                                            \
        const int r_ = (call);                                                                      \
        if (r_ != 0) return fail(MK_ERR_HIP, "%s: RCCL error %d: %s", #call, r_, (api)->GetErrorString(r_)); \
    } while (0)

MK_API int mk_comm_set_library(const char *path)
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return fail(MK_ERR_INVALID, "mk_comm_set_library: librccl is already bound in this process");
    if (!path || strlen(path) >= sizeof(g_rccl_path)) return fail(MK_ERR_INVALID, "mk_comm_set_library: bad path");
    strcpy(g_rccl_path, path);
    return MK_OK;
}

MK_API int mk_comm_unique_id(void *id128)
{
    if (!id128) return fail(MK_ERR_INVALID, "mk_comm_unique_id: null buffer (128 bytes)");
    const RcclApi *api = rccl_api();
    if (!api) return fail(MK_ERR_HIP, "mk_comm_unique_id: %s", g_rccl.why);
    RcclUniqueId id;
    MK_RCCL(api, api->GetUniqueId(&id));
    memcpy(id128, id.internal, sizeof(id.internal));
    return MK_OK;
}

MK_API int mk_comm_destroy(mk_context *ctx)
{
    if (!ctx) return fail(MK_ERR_INVALID, "null mk_context");
    if (ctx->comm && ctx->comm_owned) {
        const RcclApi *api = rccl_api();
        if (api) {
            (void)hipSetDevice(ctx->device);
            (void)api->CommDestroy(ctx->comm);
        }
    }
    ctx->comm = nullptr;
    ctx->comm_owned = false;
    return MK_OK;
}

MK_API int mk_comm_init_rank(mk_context *ctx, int nranks, int rank, const void *id128)
{
    MK_CTX(ctx);
    if (nranks < 1 || rank < 0 || rank >= nranks || !id128)
        return fail(MK_ERR_INVALID, "mk_comm_init_rank: bad argument (nranks %d, rank %d)", nranks, rank);
    const RcclApi *api = rccl_api();
    if (!api) return fail(MK_ERR_HIP, "mk_comm_init_rank: %s", g_rccl.why);
    (void)mk_comm_destroy(ctx);
    RcclUniqueId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    void *comm = nullptr;
    MK_RCCL(api, api->CommInitRank(&comm, nranks, id, rank)); // collective: every rank of the job calls it
    ctx->comm = comm;
    ctx->comm_owned = true;
    return MK_OK;
}

MK_API int mk_set_communicator(mk_context *ctx, void *nccl_comm)
{
    if (!ctx) return fail(MK_ERR_INVALID, "null mk_context");
    (void)mk_comm_destroy(ctx); // an owned communicator is released; a borrowed one is just forgotten
    ctx->comm = nccl_comm;      // the caller keeps ownership (ncclCommDestroy is the caller's to call, after mk_destroy / a detach)
    ctx->comm_owned = false;
    return MK_OK;
}

MK_API int mk_allreduce_sum(mk_context *ctx, double *d_buf, int64_t count)
{
    if (!ctx) return fail(MK_ERR_INVALID, "null mk_context");
    if (!ctx->comm)
        return fail(MK_ERR_INVALID, "mk_allreduce_sum: this context has no communicator -- call mk_comm_init_rank (every rank, with "
                                    "the id of mk_comm_unique_id) or mk_set_communicator first; there is no single-rank shortcut");
    if (!d_buf || count <= 0) return fail(MK_ERR_INVALID, "mk_allreduce_sum: bad argument");
    MK_HIP(hipSetDevice(ctx->device));
    const RcclApi *api = rccl_api();
    if (!api) return fail(MK_ERR_HIP, "mk_allreduce_sum: %s", g_rccl.why);
    MK_RCCL(api, api->AllReduce(d_buf, d_buf, (size_t)count, kNcclFloat64, kNcclSum, ctx->comm, ctx->stream)); // in place, stream-ordered
    return MK_OK;
}

MK_API int mk_destroy(mk_context *ctx)
{
    if (!ctx) return MK_OK;
    (void)hipSetDevice(ctx->device);
    for (auto &e : ctx->ev) (void)hipEventDestroy(e);
    for (auto &e : ctx->ev_pool) (void)hipEventDestroy(e);
    for (auto &v : ctx->ev_pending)
        for (auto &pr : v) {
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
    if (ctx->tlist) (void)hipFree(ctx->tlist);
    if (ctx->gws) (void)hipFree(ctx->gws);
    if (ctx->lb_counters) (void)hipFree(ctx->lb_counters);
    (void)mk_comm_destroy(ctx);
    delete ctx;
    return MK_OK;
}

MK_API int mk_set_stream(mk_context *ctx, void *s)
{
    MK_CTX(ctx);
    if (ctx->stream != (hipStream_t)s) ctx->tlist_obs = nullptr; // the cached observed-step list was built in the OLD stream's order
    ctx->stream = (hipStream_t)s;
    return MK_OK;
}

MK_API int mk_set_kernel_variant(mk_context *ctx, int which, int value)
{
    MK_CTX(ctx);
    if (which < 0 || which >= MK_VARIANT_COUNT) return fail(MK_ERR_INVALID, "mk_set_kernel_variant: unknown selector %d", which);
    if (value < 0 || value > ((which == MK_VARIANT_SMOOTHER16 || which == MK_VARIANT_SINGLE_RECORD || which == MK_VARIANT_KERNEL_FAMILY ||
                               which == MK_VARIANT_TAPE_FILTER) ? 1 : 2))
        return fail(MK_ERR_INVALID, "mk_set_kernel_variant: value must be 0 or 1 (0, 1 or 2 for the two wide selectors)");
    ctx->variant[which] = value;
    return MK_OK;
}

MK_API int mk_get_kernel_variant(mk_context *ctx, int which, int *value)
{
    MK_CTX(ctx);
    if (which < 0 || which >= MK_VARIANT_COUNT || !value) return fail(MK_ERR_INVALID, "mk_get_kernel_variant: bad argument");
    *value = ctx->variant[which];
    return MK_OK;
}

MK_API int mk_observations_changed(mk_context *ctx)
{
    MK_CTX(ctx);
    ctx->tlist_obs = nullptr;
    return MK_OK;
}

MK_API int mk_sync(mk_context *ctx)
{
    MK_CTX(ctx);
    MK_HIP(hipStreamSynchronize(ctx->stream));
    return MK_OK;
}

MK_API int mk_shape_supported(int64_t N, int64_t K)
{
    return (specialised(N, K) || generic_shape(N, K)) ? 1 : 0;
}
MK_API int mk_shape_specialised(int64_t N, int64_t K) { return specialised(N, K) ? 1 : 0; }
MK_API int64_t mk_generic_max_states(void) { return MK_GENERIC_MAX_STATES; }

MK_API int64_t mk_record_stride(int64_t n) { return mk::record_stride((int)n); }
MK_API int64_t mk_record_stride_sym(int64_t n) { return mk::record_stride_sym((int)n); }
MK_API int64_t mk_tape_stride(int64_t N, int64_t K) { return mk::tape_stride_c((int)N, (int)K); }
MK_API int64_t mk_state_tape_stride(int64_t N, int64_t K) { return mk::state_tape_stride_c((int)N, (int)K); }
MK_API int mk_tape_supported(int64_t N, int64_t K)
{
    return (N + K > 16 && K <= 16 && N + K + 1 <= 64 && specialised(N, K)) ? 1 : 0;
}
// MK_OUT_TAPE (mk_outputs.flags): 0 = not asked for, 1 = asked for and consistent, 2 = the STATE tape (with MK_OUT_VAR_ONLY:
// d_S / d_Ps are the smoothed state means / variances [B,T,n]), < 0 = an inconsistent description
static int tape_outputs(const mk_problem *p, const mk_outputs *o)
{
    if (!(o->flags & MK_OUT_TAPE)) return 0;
    if (o->flags & MK_OUT_PACKED_SYM) return fail(MK_ERR_INVALID, "MK_OUT_TAPE excludes MK_OUT_PACKED_SYM");
    if (!mk_tape_supported(p->N, p->K))
        return fail(MK_ERR_SHAPE, "MK_OUT_TAPE serves specialised shapes with 16 < N + K <= 63 (got N=%lld, K=%lld)", (long long)p->N, (long long)p->K);
    if (!p->d_loadings) return fail(MK_ERR_INVALID, "MK_OUT_TAPE needs d_loadings");
    if (o->flags & MK_OUT_VAR_ONLY) {
        if (!o->d_F || o->d_Pf || o->d_Xp || o->d_Pp || !o->d_S || !o->d_Ps)
            return fail(MK_ERR_INVALID, "MK_OUT_TAPE | MK_OUT_VAR_ONLY: d_F is the state tape, d_S / d_Ps the smoothed state means / "
                                        "variances [B,T,n]; d_Pf / d_Xp / d_Pp must be NULL");
        if (o->record_stride != mk_state_tape_stride(p->N, p->K))
            return fail(MK_ERR_INVALID, "MK_OUT_TAPE | MK_OUT_VAR_ONLY: record_stride must be mk_state_tape_stride(N, K) = %lld doubles",
                        (long long)mk_state_tape_stride(p->N, p->K));
        if (p->d_obsvar)
            return fail(MK_ERR_INVALID, "MK_OUT_TAPE | MK_OUT_VAR_ONLY serves zero observation variances (d_obsvar = NULL, Metran's R); "
                                        "use filtered records + MK_OUT_VAR_ONLY otherwise");
        return 2;
    }
    if (!o->d_F || o->d_Pf || o->d_Xp || o->d_Pp || o->d_S || o->d_Ps)
        return fail(MK_ERR_INVALID, "MK_OUT_TAPE: d_F is the tape, d_Pf / d_Xp / d_Pp / d_S / d_Ps must be NULL");
    if (o->record_stride != mk_tape_stride(p->N, p->K))
        return fail(MK_ERR_INVALID, "MK_OUT_TAPE: record_stride must be mk_tape_stride(N, K) = %lld doubles",
                    (long long)mk_tape_stride(p->N, p->K));
    return 1;
}

MK_API int mk_supported_shapes(int64_t *shapes, int cap)
{
    int cnt = mk::num_shapes();
    for (int i = 0; i < cnt && i < cap && shapes; ++i) {
        int n_, k_;
        mk::get_shape(i, &n_, &k_);
        shapes[2 * i] = n_;
        shapes[2 * i + 1] = k_;
    }
    std::lock_guard<std::mutex> lock(g_modules_mutex);
    for (const auto &m : g_modules) {
        if (cnt < cap && shapes) {
            shapes[2 * cnt] = m.N;
            shapes[2 * cnt + 1] = m.K;
        }
        ++cnt;
    }
    return cnt;
}

MK_API int mk_malloc(mk_context *ctx, size_t bytes, void **p)
{
    MK_CTX(ctx);
    if (!p) return fail(MK_ERR_INVALID, "null out-pointer");
    *p = nullptr;
    if (bytes == 0) return MK_OK;
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) return fail(MK_ERR_ALLOC, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    return MK_OK;
}

MK_API int mk_free(mk_context *ctx, void *p)
{
    MK_CTX(ctx);
    if (p) MK_HIP(hipFree(p));
    return MK_OK;
}

MK_API int mk_memcpy_h2d(mk_context *ctx, void *d, const void *h, size_t bytes)
{
    MK_CTX(ctx);
    if (bytes && (!d || !h)) return fail(MK_ERR_INVALID, "null pointer in memcpy_h2d");
    if (bytes) {
        MK_HIP(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, ctx->stream));
        MK_HIP(hipStreamSynchronize(ctx->stream)); // h may be pageable / freed by the caller
    }
    return MK_OK;
}

MK_API int mk_memcpy_d2h(mk_context *ctx, void *h, const void *d, size_t bytes)
{
    MK_CTX(ctx);
    if (bytes && (!d || !h)) return fail(MK_ERR_INVALID, "null pointer in memcpy_d2h");
    if (bytes) {
        MK_HIP(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, ctx->stream));
        MK_HIP(hipStreamSynchronize(ctx->stream));
    }
    return MK_OK;
}

MK_API int mk_memset(mk_context *ctx, void *d, int value, size_t bytes)
{
    MK_CTX(ctx);
    if (bytes && !d) return fail(MK_ERR_INVALID, "null pointer in memset");
    if (bytes) MK_HIP(hipMemsetAsync(d, value, bytes, ctx->stream));
    return MK_OK;
}

MK_API int mk_params_from_alpha(mk_context *ctx, int64_t B, int64_t R, int64_t N, int64_t K, const double *alpha,
                                const double *loadings, double dt, double *phi, double *q)
{
    MK_CTX(ctx);
    if (B <= 0 || R <= 0 || N <= 0 || K < 0 || !alpha || !phi || !q || (K > 0 && !loadings))
        return fail(MK_ERR_INVALID, "mk_params_from_alpha: bad argument");
    MK_HIP(mk::launch_params(B, R, (int)N, (int)K, alpha, loadings, dt, phi, q, ctx->stream));
    return MK_OK;
}

static int check_problem(const mk_problem *p)
{
    if (!p) return fail(MK_ERR_INVALID, "null mk_problem");
    if (p->n_instances <= 0 || p->n_records <= 0 || p->n_records > p->n_instances)
        return fail(MK_ERR_INVALID, "need 1 <= n_records <= n_instances (got R=%lld, B=%lld)",
                    (long long)p->n_records, (long long)p->n_instances);
    if (p->T <= 0 || p->N <= 0 || p->K <= 0)
        return fail(MK_ERR_INVALID, "need T, N, K >= 1 (got T=%lld N=%lld K=%lld)", (long long)p->T,
                    (long long)p->N, (long long)p->K);
    if (p->warmup < 0) return fail(MK_ERR_INVALID, "warmup must be >= 0");
    if (!mk_shape_supported(p->N, p->K))
        return fail(MK_ERR_SHAPE,
                    "no kernel for (N=%lld series, K=%lld factors): the size-generic kernels serve N + K <= %d states, specialised "
                    "ones (metran_amd.jit.ensure_shape / mk_register_shape_module / MK_SHAPES in mk_internal.h) N + K <= 64",
                    (long long)p->N, (long long)p->K, MK_GENERIC_MAX_STATES);
    if (!p->d_phi || !p->d_q) return fail(MK_ERR_INVALID, "d_phi and d_q are required");
    return MK_OK;
}

// Packed-record convention of mk_outputs.record_stride: returns 0 if `o` does not use records,
// 1 if it does (and is consistent), < 0 on an inconsistent description.
static int records_filter(const mk_problem *p, const mk_outputs *o)
{
    const bool sym = (o->flags & MK_OUT_PACKED_SYM) != 0;
    if (o->record_stride == 0) {
        if (sym) return fail(MK_ERR_INVALID, "MK_OUT_PACKED_SYM needs the record layout (record_stride = mk_record_stride_sym(n))");
        return 0;
    }
    const int64_t n = p->N + p->K, nc = sym ? n * (n + 1) / 2 : n * n, nv = n + nc;
    const int64_t want = sym ? mk::record_stride_sym((int)n) : mk::record_stride((int)n);
    if (o->record_stride != want)
        return fail(MK_ERR_INVALID, "record_stride must be %s(n) = %lld doubles", sym ? "mk_record_stride_sym" : "mk_record_stride",
                    (long long)want);
    const bool any = o->d_F || o->d_Pf || o->d_Xp || o->d_Pp;
    if (!any) return 0; // bookkeeping-only / loglik launches do not touch state arrays
    if (!(o->d_F && o->d_Pf == o->d_F + n))
        return fail(MK_ERR_INVALID, "record layout needs a filtered record array d_F with d_Pf = d_F + n");
    if ((o->d_Xp || o->d_Pp) && !(o->d_Xp && o->d_Pp == o->d_Xp + n))
        return fail(MK_ERR_INVALID, "record layout needs d_Pp = d_Xp + n (or both NULL: filtered record only)");
    if (o->d_sigmas || o->d_detfs)
        if (o->d_sigmas != o->d_F + nv || (o->d_detfs && o->d_detfs != o->d_sigmas + 1))
            return fail(MK_ERR_INVALID, "record layout needs d_sigmas = d_F + n + %s and d_detfs = d_sigmas + 1",
                        sym ? "n*(n+1)/2" : "n*n");
    return 1;
}

// The single-record routes (loglik_sparse_kernel): every instance shares ONE uploaded record, whose observed steps are
// listed once per record on the device; the objective (mk_loglik) and -- round 5 -- the record-writing filter of a few
// instances walk that list, the runs of empty steps in closed form.  d_F / d_Xp NULL: objective only.
static int sparse_route(mk_context *ctx, const mk_problem *p, double *d_mle, double *d_F, double *d_Xp, int64_t rs, int time_major,
                        int64_t *d_sigmacount, uint32_t *d_status)
{
    if (ctx->tlist_cap < p->T + 1) {
        if (ctx->tlist) MK_HIP(hipFree(ctx->tlist));
        ctx->tlist = nullptr;
        ctx->tlist_cap = 0;
        MK_HIP(hipMalloc((void **)&ctx->tlist, sizeof(int) * (size_t)(p->T + 1)));
        ctx->tlist_cap = p->T + 1;
        ctx->tlist_obs = nullptr;
    }
    mk::SparseArgs a;
    a.B = p->n_instances;
    a.T = p->T;
    a.warmup = p->warmup;
    a.ostep = (p->obs_time_major ? p->n_records : 1) * p->N; // one record: both layouts coincide
    a.obs = p->d_obs;
    a.phi = p->d_phi;
    a.q = p->d_q;
    a.loadings = p->d_loadings;
    a.obsvar = p->d_obsvar;
    a.x0 = p->d_x0;
    a.P0 = p->d_P0;
    a.tlist = ctx->tlist;
    a.rebuild = !(ctx->tlist_obs == p->d_obs && ctx->tlist_T == p->T && ctx->tlist_N == p->N && ctx->tlist_ostep == a.ostep);
    // the list is only known to describe this record once the launch that (re)builds it has been accepted: the key
    // is dropped first and committed after a successful dispatch, so a failed call leaves no stale key behind.
    // (Stream order: the list is built and read on ctx->stream; mk_set_stream drops the key when the stream changes.)
    ctx->tlist_obs = nullptr;
    a.mle = d_mle;
    a.status = d_status;
    a.F = d_F;
    a.Xp = d_Xp;
    a.rs = rs;
    a.bs = time_major ? 1 : p->T;
    a.ts = time_major ? p->n_instances : 1;
    a.sigmacount = (long long *)d_sigmacount;
    MK_HIP(timing_start(ctx, 0));
    MK_HIP(dispatch_sparse((int)p->N, (int)p->K, a, ctx->stream));
    ctx->tlist_obs = p->d_obs;
    ctx->tlist_T = p->T;
    ctx->tlist_N = p->N;
    ctx->tlist_ostep = a.ostep;
    MK_HIP(timing_stop(ctx, 0));
    return MK_OK;
}

static int do_filter(mk_context *ctx, const mk_problem *p, const mk_outputs *o)
{
    if (!p->d_obs || !p->d_loadings) return fail(MK_ERR_INVALID, "d_obs and d_loadings are required");
    const int tape = tape_outputs(p, o);
    if (tape < 0) return tape;
    const int rec = tape ? 0 : records_filter(p, o);
    if (rec < 0) return rec;
    // ONE record, a handful of instances, both record sets (the 7-tuple of seqkalmanfilter: what Metran.solve() asks of the
    // engine ~80 times): a launch of the batched filter is T sequential steps for at most a few wavefronts -- latency, not
    // throughput.  Real Metran records are sparse (examples/data: 343 observed of 6255 daily steps, kalmanfilter.py:335 skips
    // the update on the others), so the observed steps are walked one after the other and every empty step's records are
    // written in closed form by a second, fully parallel kernel (mk_kernels.hip: loglik_sparse_kernel<.., REC>, fill_gaps_kernel).
    if (rec && p->n_records == 1 && p->n_instances <= MK_SPARSE_RECORD_MAX_INSTANCES && p->N + p->K <= 16 && o->d_Xp &&
        !(o->flags & MK_OUT_PACKED_SYM) && specialised(p->N, p->K) && !ctx->variant[MK_VARIANT_SINGLE_RECORD] &&
        !ctx->variant[MK_VARIANT_KERNEL_FAMILY])
        return sparse_route(ctx, p, o->d_mle, o->d_F, o->d_Xp, o->record_stride, (int)o->time_major, o->d_sigmacount, o->d_status);
    mk::FilterArgs a;
    a.variant = ctx->variant[MK_VARIANT_WIDE_FILTER]; // 0 auto, 1 lane per state, 2 split
    a.tape = tape;
    a.tape_basis = ctx->variant[MK_VARIANT_TAPE_FILTER]; // 0 observable basis (filter_obs_kernel), 1 state basis (filter_split_kernel OUT = 4)
    a.upd = ctx->cur_upd;                                // recording pass of the wide adjoint gradient: the update tape (or NULL)
    a.us = a.upd ? mk::adjoint_update_stride_c((int)p->N, (int)p->K) : 0;
    if (a.upd) a.variant = 1;                            // ... is written by the one-model-per-wavefront filter, whatever the batch size
    a.rs = (rec || tape) ? o->record_stride : 0;
    a.sym = (rec && (o->flags & MK_OUT_PACKED_SYM)) ? 1 : 0;
    // dense sigmas/detfs are [B,T] (stride 1); inside filtered records they are RS doubles apart
    a.sig_stride = (o->record_stride && !tape) ? o->record_stride : 1;
    a.B = p->n_instances;
    a.R = p->n_records;
    a.T = p->T;
    a.warmup = p->warmup;
    a.bs = o->time_major ? 1 : p->T;
    a.ts = o->time_major ? p->n_instances : 1;
    a.obs_bs = p->obs_time_major ? 1 : p->T;
    a.obs_ts = p->obs_time_major ? p->n_records : 1;
    a.obs = p->d_obs;
    a.phi = p->d_phi;
    a.q = p->d_q;
    a.loadings = p->d_loadings;
    a.obsvar = p->d_obsvar;
    a.x0 = p->d_x0;
    a.P0 = p->d_P0;
    a.mle = o->d_mle;
    a.sigmas = o->d_sigmas;
    a.detfs = o->d_detfs;
    a.sigmacount = (long long *)o->d_sigmacount;
    a.F = o->d_F;
    a.Pf = o->d_Pf;
    a.Xp = o->d_Xp;
    a.Pp = o->d_Pp;
    a.status = o->d_status;
    if ((a.sym || a.tape) && (!specialised(p->N, p->K) || ctx->variant[MK_VARIANT_KERNEL_FAMILY] == 1))
        return fail(MK_ERR_SHAPE, "packed-symmetric records and the tape exist for specialised shapes only (N=%lld, K=%lld runs the size-generic "
                                  "kernels: mk_shape_specialised)", (long long)p->N, (long long)p->K);
    MK_HIP(timing_start(ctx, 0));
    {
        const hipError_t e = dispatch_filter((int)p->N, (int)p->K, a, ctx->stream, ctx->variant[MK_VARIANT_KERNEL_FAMILY] == 1);
        if (e == hipErrorNotSupported)
            return fail(MK_ERR_SHAPE, "a model of N=%lld series and K=%lld factors is not served on this device in this mode (the size-generic "
                                      "filter keeps the %lld x %lld covariance in LDS: %zu bytes)", (long long)p->N, (long long)p->K,
                        (long long)(p->N + p->K), (long long)(p->N + p->K), mk::generic_filter_lds_bytes((int)p->N, (int)p->K));
        MK_HIP(e);
    }
    MK_HIP(timing_stop(ctx, 0));
    return MK_OK;
}

static int do_smooth(mk_context *ctx, const mk_problem *p, const mk_outputs *o)
{
    const int tape = tape_outputs(p, o);
    if (tape < 0) return tape;
    if (!o->d_F || (!o->d_Pf && !tape))
        return fail(MK_ERR_INVALID, "the smoother reads d_F and d_Pf (filtered moments); both must be non-NULL");
    if (tape == 1 && !(o->d_sim_means || o->d_sim_vars))
        return fail(MK_ERR_INVALID, "MK_OUT_TAPE: nothing to write, give d_sim_means / d_sim_vars");
    mk::SmootherArgs a;
    a.tape = tape;
    a.obsvar = tape ? p->d_obsvar : nullptr;
    a.variant = (ctx->variant[MK_VARIANT_SMOOTHER16] ? 1 : 0) | (ctx->variant[MK_VARIANT_WIDE_SMOOTHER] == 1 ? 2 : 0) |
                (ctx->variant[MK_VARIANT_WIDE_SMOOTHER] == 2 ? 4 : 0);
    a.rs = 0;
    a.sym = 0;
    a.state_means = a.state_vars = nullptr;
    const bool sym = (o->flags & MK_OUT_PACKED_SYM) != 0, var = (o->flags & MK_OUT_VAR_ONLY) != 0;
    if (tape) {
        a.rs = o->record_stride;
    } else if (o->record_stride) {
        const int64_t n = p->N + p->K;
        const int64_t want = sym ? mk::record_stride_sym((int)n) : mk::record_stride((int)n);
        if (o->record_stride != want)
            return fail(MK_ERR_INVALID, "record_stride must be %s(n) = %lld doubles", sym ? "mk_record_stride_sym" : "mk_record_stride",
                        (long long)want);
        const bool proj = o->d_sim_means || o->d_sim_vars;
        if (o->d_Pf != o->d_F + n) return fail(MK_ERR_INVALID, "record layout needs d_Pf = d_F + n");
        if (var) {
            if (!o->d_S || !o->d_Ps || proj)
                return fail(MK_ERR_INVALID, "MK_OUT_VAR_ONLY needs d_S [B,T,n] (means) and d_Ps [B,T,n] (variances), no projection outputs");
        } else {
            if ((o->d_S || o->d_Ps) && !(o->d_S && o->d_Ps == o->d_S + n))
                return fail(MK_ERR_INVALID, "record layout needs a smoothed record array d_S with d_Ps = d_S + n");
            if (!o->d_S && !proj)
                return fail(MK_ERR_INVALID, "nothing to write: give d_S/d_Ps records or d_sim_means/d_sim_vars");
        }
        if (proj && !p->d_loadings) return fail(MK_ERR_INVALID, "the projection outputs need d_loadings");
        a.rs = o->record_stride;
        a.sym = sym ? 1 : 0;
    } else if (o->d_sim_means || o->d_sim_vars || sym || var) {
        return fail(MK_ERR_INVALID, "d_sim_means / d_sim_vars, MK_OUT_PACKED_SYM and MK_OUT_VAR_ONLY need the record layout "
                                    "(record_stride = mk_record_stride[_sym](n))");
    }
    a.R = p->n_records;
    a.loadings = p->d_loadings;
    a.scale = p->d_scale;
    a.offset = p->d_offset;
    a.sim_means = o->d_sim_means;
    a.sim_vars = o->d_sim_vars;
    a.B = p->n_instances;
    a.T = p->T;
    a.bs = o->time_major ? 1 : p->T;
    a.ts = o->time_major ? p->n_instances : 1;
    a.phi = p->d_phi;
    a.q = p->d_q;
    a.F = o->d_F;
    a.Pf = o->d_Pf;
    if (tape == 2 && !p->d_loadings) return fail(MK_ERR_INVALID, "the state tape needs d_loadings");
    a.S = var ? nullptr : o->d_S;
    a.Ps = var ? nullptr : o->d_Ps;
    if (var) {
        a.state_means = o->d_S;
        a.state_vars = o->d_Ps;
    }
    a.status = o->d_status;
    MK_HIP(timing_start(ctx, 1));
    if ((sym || tape) && (!specialised(p->N, p->K) || ctx->variant[MK_VARIANT_KERNEL_FAMILY] == 1))
        return fail(MK_ERR_SHAPE, "packed-symmetric records and the tape exist for specialised shapes only (N=%lld, K=%lld runs the "
                                  "size-generic kernels: mk_shape_specialised)", (long long)p->N, (long long)p->K);
    MK_HIP(dispatch_smoother(ctx, (int)p->N, (int)p->K, a, ctx->stream));
    MK_HIP(timing_stop(ctx, 1));
    return MK_OK;
}

MK_API int mk_filter(mk_context *ctx, const mk_problem *p, const mk_outputs *o)
{
    MK_CTX(ctx);
    if (int rc = check_problem(p)) return rc;
    if (!o) return fail(MK_ERR_INVALID, "null mk_outputs");
    return do_filter(ctx, p, o);
}

MK_API int mk_loglik(mk_context *ctx, const mk_problem *p, double *d_mle)
{
    MK_CTX(ctx);
    if (int rc = check_problem(p)) return rc;
    if (!d_mle) return fail(MK_ERR_INVALID, "d_mle is required");
    if (p->n_records == 1 && p->N + p->K <= 16 && p->d_obs && p->d_loadings && specialised(p->N, p->K) && !ctx->variant[MK_VARIANT_KERNEL_FAMILY]) {
        // every instance shares the one record (the solver's finite-difference points): walk only its
        // observed steps, the runs of empty steps in closed form (loglik_sparse_kernel)
        return sparse_route(ctx, p, d_mle, nullptr, nullptr, 0, 0, nullptr, nullptr);
    }
    mk_outputs o;
    memset(&o, 0, sizeof(o));
    o.d_mle = d_mle;
    return do_filter(ctx, p, &o);
}

MK_API int mk_smooth(mk_context *ctx, const mk_problem *p, const mk_outputs *o)
{
    MK_CTX(ctx);
    if (int rc = check_problem(p)) return rc;
    if (!o) return fail(MK_ERR_INVALID, "null mk_outputs");
    return do_smooth(ctx, p, o);
}

MK_API int mk_smooth_dense(mk_context *ctx, int64_t B, int64_t T, int64_t n, const double *d_phi, const double *d_F,
                           const double *d_Pf, const double *d_Xp, const double *d_Pp, double *d_S, double *d_Ps, uint32_t *d_status)
{
    MK_CTX(ctx);
    if (B <= 0 || T <= 0 || n <= 0 || !d_phi || !d_F || !d_Pf || !d_Xp || !d_Pp || !(d_S || d_Ps))
        return fail(MK_ERR_INVALID, "mk_smooth_dense: bad argument (all five inputs and one of d_S / d_Ps are required)");
    if (n > MK_GENERIC_MAX_STATES)
        return fail(MK_ERR_SHAPE, "mk_smooth_dense serves n <= %d states (got %lld)", MK_GENERIC_MAX_STATES, (long long)n);
    mk::GenericSmootherArgs g;
    memset(&g, 0, sizeof(g));
    g.a.B = B;
    g.a.T = T;
    g.a.bs = T; // dense [B,T,...] arrays, the reference's layout with a leading batch axis
    g.a.ts = 1;
    g.a.R = 1;
    g.a.phi = d_phi;
    g.a.q = nullptr; // not needed: the predicted covariances are the caller's
    g.a.F = d_F;
    g.a.Pf = d_Pf;
    g.a.S = d_S;
    g.a.Ps = d_Ps;
    g.a.status = d_status;
    g.N = (int)n; // no projection outputs: only n = N + K matters
    g.K = 0;
    g.Xp = d_Xp;
    g.Pp = d_Pp;
    MK_HIP(generic_workspace(ctx, B, (int)n, &g.ws));
    MK_HIP(timing_start(ctx, 1));
    MK_HIP(mk::launch_smoother_generic(g, ctx->stream));
    MK_HIP(timing_stop(ctx, 1));
    return MK_OK;
}

MK_API int mk_filter_smooth(mk_context *ctx, const mk_problem *p, const mk_outputs *o)
{
    MK_CTX(ctx);
    if (int rc = check_problem(p)) return rc;
    if (!o) return fail(MK_ERR_INVALID, "null mk_outputs");
    if ((o->flags & MK_OUT_VAR_ONLY) && !(o->flags & MK_OUT_TAPE) && (o->d_Xp || o->d_Pp))
        return fail(MK_ERR_INVALID, "MK_OUT_VAR_ONLY: d_Xp / d_Pp must be NULL (the filter writes the filtered record only)");
    if (int rc = do_filter(ctx, p, o)) return rc;
    return do_smooth(ctx, p, o);
}

MK_API int mk_simulate(mk_context *ctx, int64_t B, int64_t RZ, int64_t T, int64_t N, int64_t n, const double *Z,
                       const double *means, const double *covs, double *sm, double *sv)
{
    MK_CTX(ctx);
    if (B <= 0 || RZ <= 0 || T <= 0 || N <= 0 || n < N || !Z || !means || (sv && !covs))
        return fail(MK_ERR_INVALID, "mk_simulate: bad argument");
    MK_HIP(mk::launch_simulate(B, RZ, T, (int)N, (int)n, Z, means, covs, sm, sv, ctx->stream));
    return MK_OK;
}

MK_API int mk_decompose(mk_context *ctx, int64_t B, int64_t RZ, int64_t T, int64_t N, int64_t n, const double *Z,
                        const double *means, double *sdf, double *cdf)
{
    MK_CTX(ctx);
    if (B <= 0 || RZ <= 0 || T <= 0 || N <= 0 || n < N || !Z || !means)
        return fail(MK_ERR_INVALID, "mk_decompose: bad argument");
    MK_HIP(mk::launch_decompose(B, RZ, T, (int)N, (int)n, Z, means, sdf, cdf, ctx->stream));
    return MK_OK;
}

MK_API int mk_sum(mk_context *ctx, int64_t count, const double *v, double *out)
{
    MK_CTX(ctx);
    if (count <= 0 || !v || !out) return fail(MK_ERR_INVALID, "mk_sum: bad argument");
    MK_HIP(mk::launch_sum(count, v, out, ctx->stream));
    return MK_OK;
}

MK_API int64_t mk_adjoint_update_stride(int64_t N, int64_t K)
{
    if (N < 1 || K < 1 || N + K <= 16 || N + K > 64) return 0; // the 16-lane adjoint kernel recomputes (n <= 16)
    return mk::adjoint_update_stride_c((int)N, (int)K);
}

MK_API int mk_set_adjoint_updates(mk_context *ctx, double *d_buf, int64_t capacity_doubles)
{
    if (!ctx) return fail(MK_ERR_INVALID, "null mk_context");
    if ((d_buf && capacity_doubles <= 0) || ((uintptr_t)d_buf & 15)) return fail(MK_ERR_INVALID, "mk_set_adjoint_updates: need a 16-byte aligned buffer and its capacity");
    ctx->adj_upd = d_buf;
    ctx->adj_upd_cap = d_buf ? (size_t)capacity_doubles : 0;
    return MK_OK;
}

MK_API int mk_loglik_grad(mk_context *ctx, const mk_problem *p, double *d_work, int time_major, double *d_mle,
                          int64_t *d_sigmacount, double *d_gphi, double *d_gq, uint32_t *d_status)
{
    return mk_loglik_grad_phases(ctx, p, d_work, time_major, d_mle, d_sigmacount, d_gphi, d_gq, d_status,
                                 MK_GRAD_FORWARD | MK_GRAD_BACKWARD);
}

MK_API int mk_loglik_grad_phases(mk_context *ctx, const mk_problem *p, double *d_work, int time_major, double *d_mle,
                                 int64_t *d_sigmacount, double *d_gphi, double *d_gq, uint32_t *d_status, int phases)
{
    MK_CTX(ctx);
    if (int rc = check_problem(p)) return rc;
    if (!(phases & (MK_GRAD_FORWARD | MK_GRAD_BACKWARD)) || (phases & ~(MK_GRAD_FORWARD | MK_GRAD_BACKWARD)))
        return fail(MK_ERR_INVALID, "mk_loglik_grad_phases: phases must be MK_GRAD_FORWARD, MK_GRAD_BACKWARD or both");
    if (!d_work || !d_sigmacount || ((phases & MK_GRAD_FORWARD) && !d_mle) || ((phases & MK_GRAD_BACKWARD) && (!d_gphi || !d_gq)))
        return fail(MK_ERR_INVALID, "mk_loglik_grad: d_work and d_sigmacount are required; d_mle by the forward pass, d_gphi and d_gq by the backward pass");
    const int64_t n = p->N + p->K;
    if (!specialised(p->N, p->K) || ctx->variant[MK_VARIANT_KERNEL_FAMILY] == 1)
        return fail(MK_ERR_SHAPE, "the adjoint gradient exists for specialised shapes (N + K <= 64: ahead-of-time list or a shape "
                                  "module); N=%lld, K=%lld runs the size-generic kernels, difference mk_loglik instead",
                    (long long)p->N, (long long)p->K);
    // forward pass: filtered records only (+ per-step bookkeeping in the record pads), objective, step count
    mk_outputs o;
    memset(&o, 0, sizeof(o));
    o.d_mle = d_mle;
    o.d_sigmacount = d_sigmacount;
    o.d_status = d_status;
    o.d_F = d_work;
    o.d_Pf = d_work + n;
    o.d_sigmas = d_work + n + n * n;
    o.d_detfs = o.d_sigmas + 1;
    o.time_major = time_major;
    o.record_stride = mk::record_stride((int)n);
    // wide models (16 < N + K): with an update tape on the context that holds this call (mk_set_adjoint_updates), the forward pass
    // records (d, 1/f, v) of every scalar update and the backward walk reads them instead of recomputing the step (round 6).
    // The decision depends on the tape, the shape and the sizes only: the two phases of one gradient agree.
    double *upd = nullptr;
    const int64_t us = mk_adjoint_update_stride(p->N, p->K);
    if (us > 0 && ctx->adj_upd && (size_t)(p->n_instances * p->T * us) <= ctx->adj_upd_cap) upd = ctx->adj_upd;
    if (phases & MK_GRAD_FORWARD) {
        ctx->cur_upd = upd;
        const int rc = do_filter(ctx, p, &o);
        ctx->cur_upd = nullptr;
        if (rc) return rc;
    }
    if (!(phases & MK_GRAD_BACKWARD)) return MK_OK;
    mk::AdjointArgs a;
    a.upd = upd;
    a.us = upd ? us : 0;
    a.B = p->n_instances;
    a.R = p->n_records;
    a.T = p->T;
    a.warmup = p->warmup;
    a.bs = time_major ? 1 : p->T;
    a.ts = time_major ? p->n_instances : 1;
    a.rs = o.record_stride;
    a.obs_bs = p->obs_time_major ? 1 : p->T;
    a.obs_ts = p->obs_time_major ? p->n_records : 1;
    a.obs = p->d_obs;
    a.phi = p->d_phi;
    a.q = p->d_q;
    a.loadings = p->d_loadings;
    a.obsvar = p->d_obsvar;
    a.x0 = p->d_x0;
    a.P0 = p->d_P0;
    a.F = d_work;
    a.sigmacount = (const long long *)d_sigmacount;
    a.gphi = d_gphi;
    a.gq = d_gq;
    MK_HIP(timing_start(ctx, 1));
    MK_HIP(dispatch_adjoint((int)p->N, (int)p->K, a, ctx->stream));
    MK_HIP(timing_stop(ctx, 1)); // reported in the smoother slot of mk_last_kernel_ms / mk_kernel_ms_totals
    return MK_OK;
}

// ---- lock-step L-BFGS of the batched calibration (mk_lbfgs.hip) ----
static int lbfgs_run(mk_context *ctx, int which, mk::LbfgsArgs &a, int counter, int *h_count)
{
    if (a.R <= 0 || a.n <= 0 || a.n > MK_LBFGS_MAX_N || a.H < 1 || a.H > MK_LBFGS_MAX_H)
        return fail(MK_ERR_INVALID, "mk_lbfgs: need R >= 1, 1 <= n <= %d, 1 <= ring slots <= %d", MK_LBFGS_MAX_N, MK_LBFGS_MAX_H);
    if (!ctx->lb_counters) MK_HIP(hipMalloc((void **)&ctx->lb_counters, 4 * sizeof(int)));
    a.counters = ctx->lb_counters;
    if (counter >= 0) MK_HIP(hipMemsetAsync(ctx->lb_counters + counter, 0, sizeof(int), ctx->stream));
    MK_HIP(mk::launch_lbfgs(which, a, ctx->stream));
    if (counter >= 0 && h_count) {
        MK_HIP(hipMemcpyAsync(h_count, ctx->lb_counters + counter, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        MK_HIP(hipStreamSynchronize(ctx->stream));
    }
    return MK_OK;
}

MK_API int mk_lbfgs_direction(mk_context *ctx, int64_t R, int64_t n, int64_t history, const double *d_x, const double *d_g, const double *d_lo,
                              uint8_t *d_active, const double *d_Sh, const double *d_Yh, const double *d_rho, const int *d_hlen,
                              const int *d_hpos, double gtol, double *d_pg, double *d_d, uint8_t *d_phase, double *d_step, int *d_nback,
                              int *h_nactive)
{
    MK_CTX(ctx);
    if (!d_x || !d_g || !d_lo || !d_active || !d_pg || !d_d || !d_Sh || !d_Yh || !d_rho || !d_hlen || !d_hpos)
        return fail(MK_ERR_INVALID, "mk_lbfgs_direction: null pointer");
    mk::LbfgsArgs a;
    memset(&a, 0, sizeof(a));
    a.R = R;
    a.n = (int)n;
    a.H = (int)history;
    a.gtol = gtol;
    a.x = const_cast<double *>(d_x);
    a.g = const_cast<double *>(d_g);
    a.lo = d_lo;
    a.active = d_active;
    a.Sh = const_cast<double *>(d_Sh);
    a.Yh = const_cast<double *>(d_Yh);
    a.rho = const_cast<double *>(d_rho);
    a.hlen = const_cast<int *>(d_hlen);
    a.hpos = const_cast<int *>(d_hpos);
    a.pg = d_pg;
    a.d = d_d;
    a.phase = d_phase;
    a.step = d_step;
    a.nback = d_nback;
    return lbfgs_run(ctx, 0, a, 0, h_nactive);
}

MK_API int mk_lbfgs_trial(mk_context *ctx, int64_t R, int64_t n, const double *d_x, const double *d_d, const double *d_step, const double *d_lo,
                          const uint8_t *d_searching, const double *d_x_new, double *d_xt, double *d_xe)
{
    MK_CTX(ctx);
    if (!d_x || !d_d || !d_step || !d_lo || !d_searching || !d_x_new || !d_xt || !d_xe) return fail(MK_ERR_INVALID, "mk_lbfgs_trial: null pointer");
    mk::LbfgsArgs a;
    memset(&a, 0, sizeof(a));
    a.R = R;
    a.n = (int)n;
    a.H = 1;
    a.x = const_cast<double *>(d_x);
    a.d = const_cast<double *>(d_d);
    a.step = const_cast<double *>(d_step);
    a.lo = d_lo;
    a.searching = const_cast<uint8_t *>(d_searching);
    a.x_new = const_cast<double *>(d_x_new);
    a.xt = d_xt;
    a.xe = d_xe;
    return lbfgs_run(ctx, 1, a, -1, nullptr);
}

MK_API int mk_lbfgs_armijo(mk_context *ctx, int64_t R, int64_t n, const double *d_ft, const double *d_f, const double *d_pg, const double *d_xt,
                           const double *d_x, uint8_t *d_searching, double *d_step, double *d_x_new, double *d_f_new, int *d_nback,
                           int64_t max_backtracks, uint8_t *d_accepted, int *h_nsearching, int *h_naccepted)
{
    MK_CTX(ctx);
    if (!d_ft || !d_f || !d_pg || !d_xt || !d_x || !d_searching || !d_step || !d_x_new || !d_f_new || (d_nback && !d_accepted))
        return fail(MK_ERR_INVALID, "mk_lbfgs_armijo: null pointer");
    mk::LbfgsArgs a;
    memset(&a, 0, sizeof(a));
    a.R = R;
    a.n = (int)n;
    a.H = 1;
    a.ft = d_ft;
    a.f = const_cast<double *>(d_f);
    a.pg = const_cast<double *>(d_pg);
    a.xt = const_cast<double *>(d_xt);
    a.x = const_cast<double *>(d_x);
    a.searching = d_searching;
    a.step = d_step;
    a.x_new = d_x_new;
    a.f_new = d_f_new;
    a.nback = d_nback;
    a.max_backtracks = (int)max_backtracks;
    a.mask = d_accepted;
    if (!ctx->lb_counters) MK_HIP(hipMalloc((void **)&ctx->lb_counters, 4 * sizeof(int)));
    MK_HIP(hipMemsetAsync(ctx->lb_counters + 3, 0, sizeof(int), ctx->stream));
    if (int rc = lbfgs_run(ctx, 2, a, 1, h_naccepted ? nullptr : h_nsearching)) return rc;
    if (h_naccepted) { // both counts with one synchronisation
        int two[4];
        MK_HIP(hipMemcpyAsync(two, ctx->lb_counters, 4 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        MK_HIP(hipStreamSynchronize(ctx->stream));
        if (h_nsearching) *h_nsearching = two[1];
        *h_naccepted = two[3];
    }
    return MK_OK;
}

MK_API int mk_lbfgs_update(mk_context *ctx, int64_t R, int64_t n, int64_t history, double *d_x, double *d_f, double *d_g, const double *d_x_new,
                           const double *d_f_new, const double *d_g_new, int keep_old_gradient_if_searching, const uint8_t *d_searching,
                           const uint8_t *d_mask, uint8_t *d_active, double ftol, double *d_Sh, double *d_Yh, double *d_rho, int *d_hlen,
                           int *d_hpos, uint8_t *d_phase, int *d_nit, int64_t maxiter, int *h_ngood)
{
    MK_CTX(ctx);
    if (!d_x || !d_f || !d_g || !d_x_new || !d_f_new || !d_g_new || !(d_searching || d_mask) || !d_active || !d_Sh || !d_Yh || !d_rho || !d_hlen ||
        !d_hpos)
        return fail(MK_ERR_INVALID, "mk_lbfgs_update: null pointer");
    mk::LbfgsArgs a;
    memset(&a, 0, sizeof(a));
    a.R = R;
    a.n = (int)n;
    a.H = (int)history;
    a.keep_old = keep_old_gradient_if_searching;
    a.ftol = ftol;
    a.x = d_x;
    a.f = d_f;
    a.g = d_g;
    a.x_new = const_cast<double *>(d_x_new);
    a.f_new = const_cast<double *>(d_f_new);
    a.g_new = d_g_new;
    a.searching = const_cast<uint8_t *>(d_searching);
    a.mask = const_cast<uint8_t *>(d_mask);
    a.active = d_active;
    a.Sh = d_Sh;
    a.Yh = d_Yh;
    a.rho = d_rho;
    a.hlen = d_hlen;
    a.hpos = d_hpos;
    a.phase = d_phase;
    a.nit = d_nit;
    a.maxiter = (int)maxiter;
    return lbfgs_run(ctx, 3, a, 2, h_ngood);
}

MK_API int mk_alpha_grad(mk_context *ctx, int64_t B, int64_t R, int64_t N, int64_t K, const double *alpha,
                         const double *loadings, double dt, const double *gphi, const double *gq, double *galpha)
{
    MK_CTX(ctx);
    if (B <= 0 || R <= 0 || N <= 0 || K < 0 || !alpha || !loadings || !gphi || !gq || !galpha)
        return fail(MK_ERR_INVALID, "mk_alpha_grad: bad argument");
    MK_HIP(mk::launch_alpha_grad(B, R, (int)N, (int)K, alpha, loadings, dt, gphi, gq, galpha, ctx->stream));
    return MK_OK;
}

MK_API int mk_standardize(mk_context *ctx, int64_t R, int64_t T, int64_t N, int time_major, const double *in,
                          double *out, double *mean, double *stdev)
{
    MK_CTX(ctx);
    if (R <= 0 || T <= 0 || N <= 0 || N > 64 || !in) return fail(MK_ERR_INVALID, "mk_standardize: bad argument (1 <= N <= 64)");
    MK_HIP(mk::launch_standardize(R, T, (int)N, time_major, in, out, mean, stdev, ctx->stream));
    return MK_OK;
}

MK_API int mk_mask_observations(mk_context *ctx, int64_t count, const double *obs, const unsigned char *mask,
                                double *out)
{
    MK_CTX(ctx);
    if (count <= 0 || !obs || !mask || !out) return fail(MK_ERR_INVALID, "mk_mask_observations: bad argument");
    MK_HIP(mk::launch_mask(count, obs, mask, out, ctx->stream));
    return MK_OK;
}

MK_API int mk_pack_observations(mk_context *ctx, int64_t R, int64_t T, int64_t N, const double *obs,
                                double *observations, double *indices, int64_t *count)
{
    MK_CTX(ctx);
    if (R <= 0 || T <= 0 || N <= 0 || !obs) return fail(MK_ERR_INVALID, "mk_pack_observations: bad argument");
    static_assert(sizeof(long) == sizeof(int64_t), "LP64");
    MK_HIP(mk::launch_pack(R * T, (int)N, obs, observations, indices, reinterpret_cast<long *>(count), ctx->stream));
    return MK_OK;
}

MK_API int mk_fa_correlation(mk_context *ctx, int64_t R, int64_t T, int64_t N, int time_major, const double *obs,
                             double *corr)
{
    MK_CTX(ctx);
    if (R <= 0 || T <= 0 || N <= 0 || N > 64 || !obs || !corr)
        return fail(MK_ERR_INVALID, "mk_fa_correlation: bad argument (1 <= N <= 64)");
    MK_HIP(mk::launch_fa_corr(R, T, (int)N, time_major, obs, corr, ctx->stream));
    return MK_OK;
}

MK_API int mk_fa_analyse(mk_context *ctx, int64_t B, int64_t N, int64_t maxfactors, const double *corr, double *eigval,
                         int64_t *nfactors, int64_t *nfactors_map, int64_t *nfactors_map4, double *psi0, uint32_t *status)
{
    MK_CTX(ctx);
    if (B <= 0 || N < 2 || N > 64 || !corr) return fail(MK_ERR_INVALID, "mk_fa_analyse: bad argument (2 <= N <= 64)");
    MK_HIP(mk::launch_fa_analyse(B, (int)N, maxfactors, corr, eigval, (long long *)nfactors, (long long *)nfactors_map,
                                 (long long *)nfactors_map4, psi0, status, ctx->stream));
    return MK_OK;
}

MK_API int mk_fa_minres(mk_context *ctx, int64_t B, int64_t R, int64_t N, int64_t KMAX, const double *corr,
                        const int64_t *nfactors, const double *psi, const int64_t *order, double *fval, double *grad,
                        double *loadings)
{
    MK_CTX(ctx);
    if (B <= 0 || R <= 0 || R > B || N < 2 || N > 64 || KMAX < 1 || KMAX > N || !corr || !nfactors || !psi)
        return fail(MK_ERR_INVALID, "mk_fa_minres: bad argument (2 <= N <= 64, 1 <= KMAX <= N)");
    MK_HIP(mk::launch_fa_minres(B, R, (int)N, (int)KMAX, corr, (const long long *)nfactors, psi, (const long long *)order,
                                fval, grad, loadings, ctx->stream));
    return MK_OK;
}

MK_API int mk_fa_rotate(mk_context *ctx, int64_t B, int64_t N, int64_t KMAX, const int64_t *nfactors, double *loadings,
                        double gamma, int maxiter, double tol)
{
    MK_CTX(ctx);
    if (B <= 0 || N < 1 || N > 64 || KMAX < 1 || KMAX > N || !nfactors || !loadings)
        return fail(MK_ERR_INVALID, "mk_fa_rotate: bad argument");
    MK_HIP(mk::launch_fa_rotate(B, (int)N, (int)KMAX, (const long long *)nfactors, loadings, gamma, maxiter, tol, ctx->stream));
    return MK_OK;
}

MK_API int mk_fa_eigh(mk_context *ctx, int64_t B, int64_t N, const double *sym, double *val, double *vec)
{
    MK_CTX(ctx);
    if (B <= 0 || N < 1 || N > 64 || !sym || !val) return fail(MK_ERR_INVALID, "mk_fa_eigh: bad argument (1 <= N <= 64)");
    MK_HIP(mk::launch_fa_eigh(B, (int)N, sym, val, vec, ctx->stream));
    return MK_OK;
}

MK_API int mk_enable_timing(mk_context *ctx, int enable)
{
    MK_CTX(ctx);
    ctx->timing = enable == 2 ? 2 : (enable != 0 ? 1 : 0);
    ctx->have_filter_time = ctx->have_smooth_time = false;
    for (auto &v : ctx->ev_pending) { // pairs nobody asked for: back to the pool
        for (auto &pr : v) {
            ctx->ev_pool.push_back(pr.first);
            ctx->ev_pool.push_back(pr.second);
        }
        v.clear();
    }
    return MK_OK;
}

MK_API int mk_kernel_ms_totals(mk_context *ctx, double *filter_ms, int64_t *filter_launches, double *smoother_ms,
                               int64_t *smoother_launches)
{
    MK_CTX(ctx);
    double tot[2] = {0.0, 0.0};
    int64_t cnt[2] = {0, 0};
    for (int kind = 0; kind < 2; ++kind) {
        for (auto &pr : ctx->ev_pending[kind]) {
            float ms = 0.f;
            MK_HIP(hipEventSynchronize(pr.second));
            MK_HIP(hipEventElapsedTime(&ms, pr.first, pr.second));
            tot[kind] += ms;
            ++cnt[kind];
            ctx->ev_pool.push_back(pr.first);
            ctx->ev_pool.push_back(pr.second);
        }
        ctx->ev_pending[kind].clear();
    }
    if (filter_ms) *filter_ms = tot[0];
    if (filter_launches) *filter_launches = cnt[0];
    if (smoother_ms) *smoother_ms = tot[1];
    if (smoother_launches) *smoother_launches = cnt[1];
    return MK_OK;
}

MK_API int mk_last_kernel_ms(mk_context *ctx, float *filter_ms, float *smoother_ms)
{
    MK_CTX(ctx);
    if (filter_ms) {
        *filter_ms = -1.f;
        if (ctx->have_filter_time) {
            MK_HIP(hipEventSynchronize(ctx->ev[1]));
            MK_HIP(hipEventElapsedTime(filter_ms, ctx->ev[0], ctx->ev[1]));
        }
    }
    if (smoother_ms) {
        *smoother_ms = -1.f;
        if (ctx->have_smooth_time) {
            MK_HIP(hipEventSynchronize(ctx->ev[3]));
            MK_HIP(hipEventElapsedTime(smoother_ms, ctx->ev[2], ctx->ev[3]));
        }
    }
    return MK_OK;
}

} // extern "C"
