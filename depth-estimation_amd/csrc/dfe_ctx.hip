// dfe_ctx.hip -- context, error text, device memory helpers of the C ABI (include/dfe.h).
#include "dfe_internal.h"
#include <cstring>
#include <cstdlib>

static thread_local char g_create_err[512] = "";

// key (dfe_set_option) | environment variable read once by dfe_ctx_create | the variable's mere presence means 0 (the old DFE_NO_* switches)
const DfeOptName dfe_opt_names[DFE_NOPT] = {
    {"cascade_px", "DFE_CASCADE_PX", false},   {"fine_fuse", "DFE_FINE_FUSE", false},         {"mid_fuse", "DFE_MID_FUSE", false},
    {"fine_nq", "DFE_FINE_NQ", false},         {"mid_nq", "DFE_MID_NQ", false},               {"prep_tiles", "DFE_NO_PREP_TILES", true},
    {"xpose", "DFE_NO_XPOSE", true},           {"xpose_nt", "DFE_XPOSE_NT", false},           {"soft_epilogue", "DFE_SOFT_EPILOGUE", false},
    {"conv_batch", "DFE_NO_CONV_BATCH", true}, {"conv_nt10", "DFE_CONV_NT5", true},           {"fm64", "DFE_NO_FM64", true},
    {"fm_rows", "DFE_FM_ROWS", false},         {"sweep_ovh", "DFE_SWEEP_OVH", false},         {"sweep_blocks", "DFE_SWEEP_BLOCKS", false},
    {"debug_arena", "DFE_DEBUG_ARENA", false}, {"fm_flat", "DFE_FM_FLAT", false},             {"fm_split", "DFE_FM_SPLIT", false},
    {"conv_narrow", "DFE_CONV_NARROW", false}, {"conv_mfma", "DFE_CONV_MFMA", false},         {"fm_mfma", "DFE_FM_MFMA", false},
    {"arena_contig", "DFE_ARENA_CONTIG", false},
};

int dfe_fail(dfe_ctx *ctx, int code, const char *fmt, ...) {
    char *dst = ctx ? ctx->err : g_create_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
    return code;
}

int dfe_scratch(dfe_ctx *ctx, size_t bytes, void **out, bool plain) {
    void *&arena = plain ? ctx->scratch_plain : ctx->scratch;
    size_t &arena_bytes = plain ? ctx->scratch_plain_bytes : ctx->scratch_bytes;
    if (bytes > arena_bytes) {
        // grow-only; a free/realloc here is a stream-ordered hazard only if a previous op still
        // uses the arena, so drain the stream first (rare: sizes settle after the first call).
        DFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (arena) DFE_HIP(ctx, hipFree(arena));
        arena = nullptr;
        arena_bytes = 0;
        hipError_t e = hipErrorUnknown;
        bool contig = false;
        // Physically contiguous memory first.  A plain hipMalloc of this size is assembled from scattered pieces, and how they happen to
        // lie decides up to 10 % of a sweep's time: vga-luma 0.250 against 0.224 ms per pair, the same in EVERY call on a given arena, one
        // plain arena in three to six fast, every contiguous one fast; vga-pyramid 0.077 -> 0.073, 720p-radial 0.180 -> 0.172
        // (tools/mode_probe.py, profiles/r05_q_*, r05_r_*).  The per-CU translation counters are the same for both kinds
        // (profiles/r05_v_*): the difference is behind the L2, in how the physical layout spreads the sweep's row-sized write streams
        // over DRAM channels and banks.  Falls back to hipMalloc when no contiguous range is free.
        // `plain` callers: the batched convolution writes 4..32 feature planes side by side from every block; in contiguous memory the
        // planes' fixed distance puts those streams on the same channels (conv 4 -> 10 planes: 45.6 against 36.1 us for a VGA pair,
        // vga-learned 0.211 against 0.196 ms, tools/conv_place_probe.py), which scattered pages break up.
        if (!plain && ctx->opt_bool(DFE_OPT_ARENA_CONTIG, true)) {
            e = hipExtMallocWithFlags(&arena, bytes, hipDeviceMallocContiguous);
            contig = e == hipSuccess;
            if (!contig) { (void)hipGetLastError(); arena = nullptr; }
        }
        if (!contig) e = hipMalloc(&arena, bytes);
        if (e != hipSuccess) { arena = nullptr; return dfe_fail(ctx, DFE_E_ALLOC, "scratch hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); }
        arena_bytes = bytes;
        // DFE_DEBUG_ARENA=1: where the arena landed and what it is made of
        if (ctx->opt[DFE_OPT_DEBUG_ARENA] > 0)
            fprintf(stderr, "[dfe] scratch arena %p .. %p (%zu bytes, %s2 MiB-aligned, %s)\n", arena, (char *)arena + bytes, bytes,
                    ((uintptr_t)arena & ((1u << 21) - 1)) ? "not " : "", contig ? "physically contiguous" : "plain hipMalloc");
    }
    *out = arena;
    return DFE_OK;
}

int dfe_device_alloc(dfe_ctx *ctx, size_t bytes, void **ptr, int *contiguous) {
    DFE_REQUIRE(ctx, ptr && bytes > 0, DFE_E_ARG, "dfe_device_alloc: ptr = %p, bytes = %zu", (void *)ptr, bytes);
    DFE_HIP(ctx, hipSetDevice(ctx->device));
    *ptr = nullptr;
    bool contig = false;
    if (ctx->opt_bool(DFE_OPT_ARENA_CONTIG, true)) {
        contig = hipExtMallocWithFlags(ptr, bytes, hipDeviceMallocContiguous) == hipSuccess;
        if (!contig) { (void)hipGetLastError(); *ptr = nullptr; }
    }
    if (!contig) {
        hipError_t e = hipMalloc(ptr, bytes);
        if (e != hipSuccess) { *ptr = nullptr; return dfe_fail(ctx, DFE_E_ALLOC, "dfe_device_alloc hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); }
    }
    if (contiguous) *contiguous = contig ? 1 : 0;
    return DFE_OK;
}

int dfe_device_free(dfe_ctx *ctx, void *ptr) {
    if (!ptr) return DFE_OK;
    DFE_HIP(ctx, hipStreamSynchronize(ctx->stream));   // nothing of this ctx may still be writing there
    DFE_HIP(ctx, hipFree(ptr));
    return DFE_OK;
}

int dfe_aux_scratch(dfe_ctx *ctx, size_t bytes, void **out) {
    if (bytes > ctx->aux_bytes) {
        DFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->aux) DFE_HIP(ctx, hipFree(ctx->aux));
        ctx->aux = nullptr;
        ctx->aux_bytes = 0;
        hipError_t e = hipMalloc(&ctx->aux, bytes);
        if (e != hipSuccess) return dfe_fail(ctx, DFE_E_ALLOC, "aux hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
        ctx->aux_bytes = bytes;
    }
    *out = ctx->aux;
    return DFE_OK;
}

int dfe_graph_lookup(dfe_ctx *ctx, dfe_ctx::GraphSlot &slot, const void *key, size_t bytes) {
    if (!ctx->graphs || !ctx->stream || ctx->profile || ctx->stage_timers) return 0;   // (event records must not become graph nodes)
    const unsigned char *k = (const unsigned char *)key;
    if (slot.key.size() == bytes && memcmp(slot.key.data(), k, bytes) == 0) {
        if (slot.exec) return 2;
        if (++slot.hits < 1) return 0;
        // capture: everything the launcher enqueues on the ctx stream until dfe_graph_finish
        if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            (void)hipGetLastError();
            ctx->graphs = false;   // a stream that cannot be captured (e.g. a legacy default stream handed in by the caller)
            return 0;
        }
        return 1;
    }
    if (slot.exec) { (void)hipGraphExecDestroy(slot.exec); slot.exec = nullptr; }
    slot.key.assign(k, k + bytes);
    slot.hits = 0;
    return 0;
}

int dfe_graph_finish(dfe_ctx *ctx, dfe_ctx::GraphSlot &slot, int rc) {
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(ctx->stream, &g);
    if (rc != DFE_OK) {   // the launcher failed half way: nothing has run, the caller sees its error
        if (g) (void)hipGraphDestroy(g);
        return rc;
    }
    if (e != hipSuccess || !g) return dfe_fail(ctx, DFE_E_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
    e = hipGraphInstantiate(&slot.exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) { slot.exec = nullptr; return dfe_fail(ctx, DFE_E_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e)); }
    DFE_HIP(ctx, hipGraphLaunch(slot.exec, ctx->stream));
    return DFE_OK;
}

extern "C" {

int dfe_version(void) { return 100; }

int dfe_ctx_create(int device, void *stream, int own_stream, dfe_ctx **out) {
    if (!out) return dfe_fail(nullptr, DFE_E_ARG, "dfe_ctx_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return dfe_fail(nullptr, DFE_E_HIP, "dfe_ctx_create: no HIP device (%s); libdfe has no CPU fallback",
                        e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device < 0 || device >= n) return dfe_fail(nullptr, DFE_E_ARG, "dfe_ctx_create: device %d of %d", device, n);
    int prev_device = -1;
    (void)hipGetDevice(&prev_device);
    struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore{prev_device};   // the caller's current device is not ours to change
    e = hipSetDevice(device);
    if (e != hipSuccess) return dfe_fail(nullptr, DFE_E_HIP, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return dfe_fail(nullptr, DFE_E_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return dfe_fail(nullptr, DFE_E_UNSUPPORTED, "device %d is %s; libdfe ships gfx950 code only", device, prop.gcnArchName);
    dfe_ctx *ctx = new dfe_ctx();
    ctx->device = device;
    ctx->ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // tuning switches: the environment is read here, once; afterwards only dfe_set_option changes them
    for (int i = 0; i < DFE_NOPT; ++i)
        if (const char *e = getenv(dfe_opt_names[i].env)) ctx->opt[i] = dfe_opt_names[i].env_presence_means_zero ? 0 : atoi(e);
    if (const char *e = getenv("DFE_GRAPHS")) ctx->graphs = atoi(e) != 0;
    if (const char *e = getenv("DFE_CV_MODE")) { int m = atoi(e); if (m >= 0 && m <= 3) ctx->cv_mode = m; }   // tuning: initial kernel mode
    if (const char *e = getenv("DFE_CV_TILE")) { int t = atoi(e); if ((t >= 0 && t <= 7) || (t > 100 && t <= 164)) ctx->cv_tyq = t; }   // tuning: initial tile code
    if (!own_stream) {
        ctx->stream = (hipStream_t)stream;   // NULL = the default stream
    } else {
        e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete ctx; return dfe_fail(nullptr, DFE_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
        ctx->own_stream = true;
    }
    e = hipMalloc((void **)&ctx->dflag, sizeof(int));
    if (e != hipSuccess) { dfe_ctx_destroy(ctx); return dfe_fail(nullptr, DFE_E_ALLOC, "hipMalloc flag: %s", hipGetErrorString(e)); }
    *out = ctx;
    return DFE_OK;
}

void dfe_ctx_destroy(dfe_ctx *ctx) {
    if (!ctx) return;
    DfeDeviceGuard guard(ctx);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->ms_graph.exec) (void)hipGraphExecDestroy(ctx->ms_graph.exec);
    for (hipEvent_t e : ctx->prof_events) (void)hipEventDestroy(e);
    for (const dfe_ctx::StageEvent &e : ctx->stage_events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->scratch_plain) (void)hipFree(ctx->scratch_plain);
    if (ctx->cn_coef) (void)hipFree(ctx->cn_coef);
    if (ctx->ingest) (void)hipFree(ctx->ingest);
    if (ctx->aux) (void)hipFree(ctx->aux);
    for (int i = 0; i < DFE_NSLOT; ++i) {
        if (ctx->slot[i]) (void)hipFree(ctx->slot[i]);
        if (ctx->copied[i]) (void)hipEventDestroy(ctx->copied[i]);
        if (ctx->consumed[i]) (void)hipEventDestroy(ctx->consumed[i]);
    }
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->dflag) (void)hipFree(ctx->dflag);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *dfe_last_error(const dfe_ctx *ctx) { return ctx ? ctx->err : g_create_err; }

int dfe_ctx_synchronize(dfe_ctx *ctx) {
    DFE_ENTER(ctx);
    DFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return DFE_OK;
}

void *dfe_ctx_stream(dfe_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int dfe_malloc(dfe_ctx *ctx, size_t bytes, void **dptr) {
    DFE_REQUIRE(ctx, ctx && dptr, DFE_E_ARG, "dfe_malloc: NULL argument");
    DfeDeviceGuard guard(ctx);
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
    if (e != hipSuccess) return dfe_fail(ctx, DFE_E_ALLOC, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    return DFE_OK;
}

int dfe_free(dfe_ctx *ctx, void *dptr) {
    DFE_ENTER(ctx);
    if (dptr) DFE_HIP(ctx, hipFree(dptr));
    return DFE_OK;
}

int dfe_memcpy_h2d(dfe_ctx *ctx, void *dst, const void *src, size_t bytes) {
    DFE_REQUIRE(ctx, ctx && (bytes == 0 || (dst && src)), DFE_E_ARG, "dfe_memcpy_h2d: NULL argument");
    DfeDeviceGuard guard(ctx);
    {
        DfeStageScope st(ctx, DFE_STAGE_LOAD);
        DFE_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    DFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return DFE_OK;
}

int dfe_memcpy_d2h(dfe_ctx *ctx, void *dst, const void *src, size_t bytes) {
    DFE_REQUIRE(ctx, ctx && (bytes == 0 || (dst && src)), DFE_E_ARG, "dfe_memcpy_d2h: NULL argument");
    DfeDeviceGuard guard(ctx);
    DFE_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    DFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return DFE_OK;
}

int dfe_host_register(dfe_ctx *ctx, void *ptr, size_t bytes) {
    DFE_REQUIRE(ctx, ctx && ptr && bytes > 0, DFE_E_ARG, "dfe_host_register: NULL / empty range");
    DfeDeviceGuard guard(ctx);
    hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterDefault);
    if (e == hipErrorHostMemoryAlreadyRegistered) { (void)hipGetLastError(); return DFE_OK; }
    if (e != hipSuccess) return dfe_fail(ctx, DFE_E_HIP, "hipHostRegister(%p, %zu): %s", ptr, bytes, hipGetErrorString(e));
    return DFE_OK;
}

int dfe_host_unregister(dfe_ctx *ctx, void *ptr) {
    DFE_REQUIRE(ctx, ctx && ptr, DFE_E_ARG, "dfe_host_unregister: NULL argument");
    DfeDeviceGuard guard(ctx);
    hipError_t e = hipHostUnregister(ptr);
    if (e == hipErrorHostMemoryNotRegistered) { (void)hipGetLastError(); return DFE_OK; }
    if (e != hipSuccess) return dfe_fail(ctx, DFE_E_HIP, "hipHostUnregister(%p): %s", ptr, hipGetErrorString(e));
    return DFE_OK;
}

int dfe_host_alloc(dfe_ctx *ctx, size_t bytes, void **hptr) {
    DFE_REQUIRE(ctx, ctx && hptr && bytes > 0, DFE_E_ARG, "dfe_host_alloc: bad argument");
    DfeDeviceGuard guard(ctx);
    *hptr = nullptr;
    hipError_t e = hipHostMalloc(hptr, bytes, hipHostMallocDefault);
    if (e != hipSuccess) return dfe_fail(ctx, DFE_E_ALLOC, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e));
    return DFE_OK;
}

int dfe_host_free(dfe_ctx *ctx, void *hptr) {
    DFE_REQUIRE(ctx, ctx, DFE_E_ARG, "dfe_host_free: ctx is NULL");
    if (!hptr) return DFE_OK;
    DfeDeviceGuard guard(ctx);
    DFE_HIP(ctx, hipHostFree(hptr));
    return DFE_OK;
}

int dfe_set_cost_volume_kernel(dfe_ctx *ctx, int mode) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, mode >= 0 && mode <= 3, DFE_E_ARG, "cost-volume kernel mode %d not in 0..3", mode);
    ctx->cv_mode = mode;
    return DFE_OK;
}

int dfe_set_cost_volume_tile(dfe_ctx *ctx, int tyq) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, (tyq >= 0 && tyq <= 7) || (tyq > 100 && tyq <= 164), DFE_E_ARG, "tile height code %d not in 0..7 / 101..164", tyq);
    ctx->cv_tyq = tyq;
    return DFE_OK;
}

const char *dfe_last_kernel(const dfe_ctx *ctx) { return ctx ? ctx->last_kernel : ""; }

static int dfe_opt_find(const char *key) {
    if (!key) return -1;
    for (int i = 0; i < DFE_NOPT; ++i)
        if (strcmp(key, dfe_opt_names[i].key) == 0) return i;
    return -1;
}

int dfe_set_option(dfe_ctx *ctx, const char *key, int value) {
    DFE_ENTER(ctx);
    if (key && strcmp(key, "graphs") == 0) { ctx->graphs = value > 0; return DFE_OK; }
    const int o = dfe_opt_find(key);
    DFE_REQUIRE(ctx, o >= 0, DFE_E_ARG, "dfe_set_option: unknown key '%s'", key ? key : "(null)");
    DFE_REQUIRE(ctx, value >= -1, DFE_E_ARG, "dfe_set_option: %s = %d (>= 0, or -1 for automatic)", key, value);
    ctx->opt[o] = value;
    return DFE_OK;
}

int dfe_get_option(dfe_ctx *ctx, const char *key, int *value) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, value, DFE_E_ARG, "dfe_get_option: value is NULL");
    if (key && strcmp(key, "graphs") == 0) { *value = ctx->graphs ? 1 : 0; return DFE_OK; }
    const int o = dfe_opt_find(key);
    DFE_REQUIRE(ctx, o >= 0, DFE_E_ARG, "dfe_get_option: unknown key '%s'", key ? key : "(null)");
    *value = ctx->opt[o];
    return DFE_OK;
}

int dfe_set_scratch_limit(dfe_ctx *ctx, size_t bytes) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, bytes >= ((size_t)1 << 20), DFE_E_ARG, "scratch limit %zu below 1 MiB", bytes);
    ctx->scratch_limit = bytes;
    return DFE_OK;
}

int dfe_stage_timers_enable(dfe_ctx *ctx, int on) {
    DFE_ENTER(ctx);
    ctx->stage_timers = on != 0;
    ctx->stage_depth = 0;
    return DFE_OK;
}

int dfe_stage_timers_read(dfe_ctx *ctx, double *ms, int *regions) {
    DFE_REQUIRE(ctx, ctx && ms && regions, DFE_E_ARG, "dfe_stage_timers_read: NULL argument");
    DfeDeviceGuard guard(ctx);
    DFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int s = 0; s < DFE_NSTAGES; ++s) { ms[s] = 0; regions[s] = 0; }
    for (const dfe_ctx::StageEvent &e : ctx->stage_events) {
        float t = 0.f;
        if (e.stage >= 0 && e.stage < DFE_NSTAGES && hipEventElapsedTime(&t, e.a, e.b) == hipSuccess) { ms[e.stage] += t; ++regions[e.stage]; }
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    ctx->stage_events.clear();
    return DFE_OK;
}

int dfe_profile_enable(dfe_ctx *ctx, int on) {
    DFE_ENTER(ctx);
    ctx->profile = on != 0;
    return DFE_OK;
}

int dfe_profile_read(dfe_ctx *ctx, double *total_ms, int *launches) { return dfe_profile_read_each(ctx, total_ms, launches, nullptr, 0); }

int dfe_profile_read_each(dfe_ctx *ctx, double *total_ms, int *launches, float *each_ms, int cap) {
    DFE_REQUIRE(ctx, ctx && total_ms && launches && (each_ms || cap <= 0), DFE_E_ARG, "dfe_profile_read: NULL argument");
    DfeDeviceGuard guard(ctx);
    DFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double sum = 0;
    int n = 0;
    for (size_t i = 0; i + 1 < ctx->prof_events.size(); i += 2) {
        float ms = 0.f;
        DFE_HIP(ctx, hipEventElapsedTime(&ms, ctx->prof_events[i], ctx->prof_events[i + 1]));
        sum += ms;
        if (n < cap) each_ms[n] = ms;
        ++n;
    }
    for (hipEvent_t e : ctx->prof_events) (void)hipEventDestroy(e);
    ctx->prof_events.clear();
    *total_ms = sum;
    *launches = n;
    return DFE_OK;
}

}  // extern "C"
