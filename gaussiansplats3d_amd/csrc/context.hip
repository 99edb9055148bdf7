// context.hip — error reporting and the per-GPU context of libgsplat_hip.so.
#include <stdarg.h>
#include <stdlib.h>

#include "gs_internal.hpp"

static thread_local char g_err[512] = "";

void gs_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {

const char* gs_last_error(void) { return g_err; }
int gs_abi_version(void) { return GS_ABI_VERSION; }

int gs_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        gs_set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return GS_ERR_HIP;
    }
    return n;
}

int gs_context_create(int device, void* hip_stream, gs_context** out) {
    const char* serial = getenv("GSPLAT_SERIAL");               // "1" (anything but empty / "0"): one stream for everything
    const bool single = serial && serial[0] != '\0' && !(serial[0] == '0' && serial[1] == '\0');
    return gs_context_create_ex(device, hip_stream, single ? GS_CTX_SINGLE_STREAM : 0u, out);
}

int gs_context_create_ex(int device, void* hip_stream, uint32_t flags, gs_context** out) {
    GS_REQUIRE(out != nullptr, "out == NULL");
    GS_REQUIRE((flags & ~(GS_CTX_SINGLE_STREAM | GS_CTX_STAGE_TIMING | GS_CTX_FORK_JOIN)) == 0, "unknown context flags");
    GS_REQUIRE(!((flags & GS_CTX_SINGLE_STREAM) && (flags & GS_CTX_FORK_JOIN)), "GS_CTX_FORK_JOIN needs the streams GS_CTX_SINGLE_STREAM removes");
    *out = nullptr;
    int n = gs_device_count();
    if (n < 0) return n;
    if (n == 0) {
        gs_set_error("no HIP device visible: libgsplat_hip has no CPU fallback");
        return GS_ERR_HIP;
    }
    GS_REQUIRE(device >= 0 && device < n, "device index out of range");
    GS_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    GS_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        gs_set_error("device %d is %s; this library carries gfx950 (MI355X) code objects only", device, prop.gcnArchName);
        return GS_ERR_HIP;
    }
    gs_context* ctx = new (std::nothrow) gs_context();
    if (!ctx) return GS_ERR_NOMEM;
    ctx->device = device;
    ctx->cu_count = prop.multiProcessorCount;
    if (hip_stream) {
        ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
    } else {
        hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            gs_set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
            delete ctx;
            return GS_ERR_HIP;
        }
        ctx->own_stream = true;
    }
    const char* wide = getenv("GSPLAT_WIDE_ENTRY_KEYS");
    ctx->wide_entry_keys = wide && wide[0] == '1';
    ctx->serial = (flags & GS_CTX_SINGLE_STREAM) != 0;
    if (const char* ks = getenv("GSPLAT_KERNEL_SAMPLE")) ctx->kernel_sample = (uint32_t)atoi(ks);
    const char* se = getenv("GSPLAT_STAGE_EVENTS");
    ctx->stage_events = (flags & GS_CTX_STAGE_TIMING) != 0;
    if (se && se[0] && !se[1]) ctx->stage_events = se[0] != '0';
    if (!ctx->serial) {
        hipError_t e = hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking);
        if (e != hipSuccess) {
            gs_set_error("hipStreamCreate(aux) failed: %s", hipGetErrorString(e));
            gs_context_destroy(ctx);
            return GS_ERR_HIP;
        }
    } else {
        ctx->aux = ctx->stream;
    }
    ctx->fork_join = (flags & GS_CTX_FORK_JOIN) != 0;
    if (ctx->fork_join && hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess) {
        gs_set_error("hipEventCreate(fork) failed");
        gs_context_destroy(ctx);
        return GS_ERR_HIP;
    }
    int st = ctx->radix.init();
    if (st >= 0) {
        bool ok = false;
        st = gs_selftest_lds_atomic_order(ctx, &ok);
        ctx->lds_atomic_lane_order = ok && !getenv("GSPLAT_NO_LDS_ATOMIC_RANK");
    }
    if (st < 0) {
        gs_context_destroy(ctx);
        return st;
    }
    *out = ctx;
    return GS_OK;
}

void gs_context_destroy(gs_context* ctx) {
    if (!ctx) return;
    ScopedDevice sd(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->aux && ctx->aux != ctx->stream) {
        (void)hipStreamSynchronize(ctx->aux);
        (void)hipStreamDestroy(ctx->aux);
    }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    ctx->radix.block_hist.release();
    ctx->radix.digit_total.release();
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int gs_context_synchronize(gs_context* ctx) {
    GS_REQUIRE(ctx != nullptr, "ctx == NULL");
    ScopedDevice sd(ctx->device);
    GS_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->aux != ctx->stream) GS_HIP(hipStreamSynchronize(ctx->aux));
    return GS_OK;
}

int gs_context_set_stage_timing(gs_context* ctx, int enable) {
    GS_REQUIRE(ctx != nullptr, "ctx == NULL");
    ctx->stage_events = enable != 0;
    return GS_OK;
}

}  // extern "C"
