// group.hip — the one exchange step of a multi-GPU draw: row strips of the framebuffer gathered to one rank over RCCL
// (xGMI inside a node).  The reference has no counterpart (one WebGL context, /root/reference/src/Viewer.js:1616).
//
// One rank per GPU (a process or a thread, each with its own gs_context).  Every rank keys all splats but sorts, bins and
// blends only what reaches its strip of 16-px tile rows (gs_sorter_set_visibility_cull + gs_camera.tile_row_begin/end), so the
// only bytes that cross GPUs per frame are the RGBA8 strips: W*H*4 in total, received by the root over up to 7 distinct
// point-to-point links at once - one grouped ncclSend / ncclRecv (a gatherv), enqueued on the context's stream like any
// kernel.  librccl is loaded on first use, so single-GPU users never need it.
#include <dlfcn.h>

#include <mutex>

#include "gs_internal.hpp"

namespace {

typedef void* comm_t;
struct unique_id { char internal[128]; };
static_assert(sizeof(unique_id) == GS_GROUP_ID_BYTES, "ncclUniqueId is 128 bytes");
constexpr int kUint8 = 1;                                 // ncclUint8

struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(unique_id*) = nullptr;
    int (*CommInitRank)(comm_t*, int, unique_id, int) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

// thread-per-rank callers (one gs_context per thread) may reach this concurrently: the table is filled exactly once
Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (r.lib) {
            auto sym = [&](const char* n) { return dlsym(r.lib, n); };
            r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
            r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
            r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
            r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
            r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
            r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
            r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
            r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
            if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.Send || !r.Recv) {
                dlclose(r.lib);
                r.lib = nullptr;
            }
        }
    });
    return r.lib ? &r : nullptr;
}

}  // namespace

struct gs_group {
    gs_context* ctx = nullptr;
    comm_t comm = nullptr;
    uint32_t world = 1, rank = 0;
    DevBuf full;                       // the root's gathered frame (gs_group_render_gather)
    // overlapped gathers (gs_group_set_overlap): the transfers run on `coll`; call k records `ready` on the context's stream
    // (what it has to wait for) and `done[k & 1]` behind the transfer (what call k + 1 makes the context's stream wait for)
    bool overlap = false;
    hipStream_t coll = nullptr;
    hipEvent_t ready = nullptr, done[2] = {nullptr, nullptr};
    uint32_t calls = 0;
    DevBuf full_alt, strip_alt[2];     // gs_group_render_gather's second frame buffer and its two draw targets
};

#define GS_NCCL(expr)                                                                              \
    do {                                                                                           \
        const int _r = (expr);                                                                     \
        if (_r != 0) {                                                                             \
            gs_set_error("%s failed: %s", #expr, R->GetErrorString ? R->GetErrorString(_r) : "RCCL error"); \
            return GS_ERR_HIP;                                                                     \
        }                                                                                          \
    } while (0)

extern "C" void gs_group_destroy(gs_group* g);

// rank r sends SELFTEST_BASE + 4096 * r bytes, byte i = (uint8_t)(i * 131 + r * 29 + 7); the root receives the strips
// back to back and compares them on the host
static int group_selftest(gs_group* g) {
    Rccl* R = rccl();
    GS_REQUIRE(R != nullptr, "RCCL is gone");
    constexpr size_t SELFTEST_BASE = 64 * 1024;
    ScopedDevice sd(g->ctx->device);
    hipStream_t st = g->ctx->stream;
    auto len = [&](uint32_t r) { return SELFTEST_BASE + 4096 * (size_t)r; };
    auto val = [](uint32_t r, size_t i) { return (uint8_t)(i * 131u + r * 29u + 7u); };
    size_t total = 0;
    for (uint32_t r = 0; r < g->world; r++) total += len(r);
    DevBuf mine, all;
    GS_TRY(mine.alloc(len(g->rank)));
    std::vector<uint8_t> host(len(g->rank));
    for (size_t i = 0; i < host.size(); i++) host[i] = val(g->rank, i);
    GS_HIP(hipMemcpyAsync(mine.p, host.data(), host.size(), hipMemcpyHostToDevice, st));
    if (g->rank == 0) {
        GS_TRY(all.alloc(total));
        GS_HIP(hipMemsetAsync(all.p, 0, total, st));
        GS_HIP(hipMemcpyAsync(all.p, mine.p, len(0), hipMemcpyDeviceToDevice, st));
    }
    GS_NCCL(R->GroupStart());
    if (g->rank == 0) {
        size_t off = len(0);
        for (uint32_t r = 1; r < g->world; r++) {
            GS_NCCL(R->Recv(all.as<char>() + off, len(r), kUint8, (int)r, g->comm, st));
            off += len(r);
        }
    } else {
        GS_NCCL(R->Send(mine.p, len(g->rank), kUint8, 0, g->comm, st));
    }
    GS_NCCL(R->GroupEnd());
    GS_HIP(hipStreamSynchronize(st));
    if (g->rank == 0) {
        std::vector<uint8_t> got(total);
        GS_HIP(hipMemcpy(got.data(), all.p, total, hipMemcpyDeviceToHost));
        size_t off = 0;
        for (uint32_t r = 0; r < g->world; r++) {
            for (size_t i = 0; i < len(r); i++)
                if (got[off + i] != val(r, i)) {
                    gs_set_error("gs_group_create self-test: byte %zu of rank %u's %zu-byte pattern arrived as %u, expected %u - the RCCL "
                                 "gather does not deliver what the ranks send", i, r, len(r), (unsigned)got[off + i], (unsigned)val(r, i));
                    return GS_ERR_HIP;
                }
            off += len(r);
        }
    }
    return GS_OK;
}

extern "C" {

int gs_group_unique_id(uint8_t* id_out) {
    GS_REQUIRE(id_out != nullptr, "id_out == NULL");
    Rccl* R = rccl();
    if (!R) {
        gs_set_error("librccl.so could not be loaded: %s", dlerror());
        return GS_ERR_UNSUPPORTED;
    }
    unique_id id;
    GS_NCCL(R->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return GS_OK;
}

int gs_group_create(gs_context* ctx, const uint8_t* id_bytes, uint32_t world_size, uint32_t rank, gs_group** out) {
    GS_REQUIRE(ctx && out, "ctx / out == NULL");
    *out = nullptr;
    GS_REQUIRE(world_size >= 1 && rank < world_size, "rank outside [0, world_size)");
    GS_REQUIRE(world_size == 1 || id_bytes, "a group of more than one rank needs the id from gs_group_unique_id");
    gs_group* g = new (std::nothrow) gs_group();
    if (!g) return GS_ERR_NOMEM;
    g->ctx = ctx;
    g->world = world_size;
    g->rank = rank;
    if (world_size > 1) {                                  // a group of one never touches RCCL
        Rccl* R = rccl();
        if (!R) {
            gs_set_error("librccl.so could not be loaded: %s", dlerror());
            delete g;
            return GS_ERR_UNSUPPORTED;
        }
        ScopedDevice sd(ctx->device);
        unique_id id;
        memcpy(&id, id_bytes, sizeof(id));
        const int r = R->CommInitRank(&g->comm, (int)world_size, id, (int)rank);
        if (r != 0) {
            gs_set_error("ncclCommInitRank failed: %s", R->GetErrorString ? R->GetErrorString(r) : "RCCL error");
            delete g;
            return GS_ERR_HIP;
        }
    }
    // First use of the RCCL branch: prove it.  Every rank sends a known pattern to rank 0 through the very calls the strip
    // gather makes (one grouped ncclSend / ncclRecv on the context's stream, ragged lengths), the root checks every byte.
    // This builder never had two GPUs: the first run on a real node either proves the path or fails HERE, loudly, instead of
    // handing back a plausible-looking frame.  GS_GROUP_SELFTEST=0 skips it.
    if (world_size > 1) {
        const char* e = getenv("GS_GROUP_SELFTEST");
        if (!e || strcmp(e, "0") != 0) {
            const int st = group_selftest(g);
            if (st != GS_OK) {
                gs_group_destroy(g);
                return st;
            }
        }
    }
    *out = g;
    return GS_OK;
}

void gs_group_destroy(gs_group* g) {
    if (!g) return;
    ScopedDevice sd(g->ctx->device);
    if (g->coll) (void)hipStreamSynchronize(g->coll);
    if (g->comm) {
        (void)hipStreamSynchronize(g->ctx->stream);
        if (Rccl* R = rccl()) (void)R->CommDestroy(g->comm);
    }
    if (g->ready) (void)hipEventDestroy(g->ready);
    for (hipEvent_t e : g->done) if (e) (void)hipEventDestroy(e);
    if (g->coll) (void)hipStreamDestroy(g->coll);
    delete g;
}

int gs_group_set_overlap(gs_group* g, int enabled) {
    GS_REQUIRE(g, "group == NULL");
    ScopedDevice sd(g->ctx->device);
    if (enabled && !g->coll) {
        // created into locals and committed together: a failure half-way must not leave a stream without its events behind
        // (a later call would skip creation and record / wait on null events)
        hipStream_t coll = nullptr;
        hipEvent_t ready = nullptr;
        constexpr size_t ND = sizeof(g->done) / sizeof(g->done[0]);
        hipEvent_t done[ND] = {};
        bool ok = hipStreamCreateWithFlags(&coll, hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&ready, hipEventDisableTiming) == hipSuccess;
        for (size_t k = 0; ok && k < ND; k++) ok = hipEventCreateWithFlags(&done[k], hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            for (hipEvent_t e : done) if (e) (void)hipEventDestroy(e);
            if (ready) (void)hipEventDestroy(ready);
            if (coll) (void)hipStreamDestroy(coll);
            gs_set_error("gs_group_set_overlap: creating the gather stream / events failed");
            return GS_ERR_HIP;
        }
        g->coll = coll;
        g->ready = ready;
        for (size_t k = 0; k < ND; k++) g->done[k] = done[k];
    }
    if (!enabled && g->overlap) {                          // back on the context's stream: nothing may still be in flight
        GS_HIP(hipStreamSynchronize(g->coll));
        g->calls = 0;
    }
    g->overlap = enabled != 0;
    return GS_OK;
}

int gs_group_wait(gs_group* g) {
    GS_REQUIRE(g, "group == NULL");
    ScopedDevice sd(g->ctx->device);
    if (g->coll) GS_HIP(hipStreamSynchronize(g->coll));
    return GS_OK;
}

int gs_group_gather_strips(gs_group* g, const void* strip_dev, void* full_dev, uint32_t width, const uint32_t* row_begin,
                           const uint32_t* row_end, uint32_t root) {
    GS_REQUIRE(g && row_begin && row_end, "group / row tables == NULL");
    GS_REQUIRE(root < g->world && width > 0, "root outside the group or width == 0");
    for (uint32_t r = 0; r < g->world; r++) GS_REQUIRE(row_begin[r] <= row_end[r], "strip rows: begin > end");
    GS_REQUIRE(g->rank != root || full_dev, "the root needs full_dev");
    const size_t row_bytes = (size_t)width * 4;
    const size_t mine = (size_t)(row_end[g->rank] - row_begin[g->rank]) * row_bytes;
    GS_REQUIRE(mine == 0 || strip_dev, "strip_dev == NULL");
    ScopedDevice sd(g->ctx->device);
    hipStream_t st = g->ctx->stream;
    // records the end of this call's transfer (overlap mode), whichever way the function leaves
    struct Done {
        gs_group* g; bool armed;
        ~Done() { if (armed) { (void)hipEventRecord(g->done[g->calls & 1u], g->coll); g->calls++; } }
    } done_guard{g, false};
    if (g->overlap) {
        // the transfer starts when the context's stream has finished what it holds now (this frame's draw) and runs beside
        // whatever the caller enqueues next; the context's stream in turn waits for the PREVIOUS call's transfer, so that the
        // draw after this call may reuse the buffers of the call before that one (the caller alternates two sets)
        GS_HIP(hipEventRecord(g->ready, st));
        GS_HIP(hipStreamWaitEvent(g->coll, g->ready, 0));
        if (g->calls) GS_HIP(hipStreamWaitEvent(st, g->done[(g->calls - 1u) & 1u], 0));
        st = g->coll;
        done_guard.armed = true;
    }
    if (g->rank == root && mine)
        GS_HIP(hipMemcpyAsync(static_cast<char*>(full_dev) + (size_t)row_begin[root] * row_bytes, strip_dev, mine, hipMemcpyDeviceToDevice, st));
    if (g->world == 1) return GS_OK;
    Rccl* R = rccl();
    GS_REQUIRE(R != nullptr, "RCCL is gone");
    GS_NCCL(R->GroupStart());
    if (g->rank == root) {
        for (uint32_t r = 0; r < g->world; r++) {
            const size_t bytes = (size_t)(row_end[r] - row_begin[r]) * row_bytes;
            if (r == root || bytes == 0) continue;
            GS_NCCL(R->Recv(static_cast<char*>(full_dev) + (size_t)row_begin[r] * row_bytes, bytes, kUint8, (int)r, g->comm, st));
        }
    } else if (mine) {
        GS_NCCL(R->Send(strip_dev, mine, kUint8, (int)root, g->comm, st));
    }
    GS_NCCL(R->GroupEnd());
    return GS_OK;
}

int gs_group_render_gather(gs_group* g, gs_mesh* m, const gs_camera* cam, const uint32_t* sorted_host, gs_sorter* sorter, uint32_t render_count,
                           const uint32_t* row_begin, const uint32_t* row_end, uint32_t root, uint8_t* rgba_out_host) {
    // Argument errors every rank sees alike (same tables, same root) may return early: no rank enters the collective.
    GS_REQUIRE(g && m && cam && row_begin && row_end, "group / mesh / camera / row tables == NULL");
    GS_REQUIRE(m->ctx == g->ctx, "mesh lives on another context");
    GS_REQUIRE(root < g->world, "root outside the group");
    for (uint32_t r = 0; r < g->world; r++)
        GS_REQUIRE(row_begin[r] % GS_TILE == 0 && (row_end[r] % GS_TILE == 0 || row_end[r] == cam->height) &&
                       row_begin[r] <= row_end[r] && row_end[r] <= cam->height,
                   "every rank's pixel rows must be whole 16-px tile rows of the viewport");
    const uint32_t y0 = row_begin[g->rank], y1 = row_end[g->rank];
    gs_camera c = *cam;
    c.tile_row_begin = y0 / GS_TILE;
    c.tile_row_end = (y1 + GS_TILE - 1) / GS_TILE;
    ScopedDevice sd(g->ctx->device);
    const size_t frame_bytes = (size_t)cam->width * cam->height * 4, strip_bytes = (size_t)(y1 - y0) * cam->width * 4;
    // From here on a failure is LOCAL (this rank's draw, this rank's allocation).  The other ranks are already on their way
    // into the gather, and the root would wait for this rank's strip forever, so this rank always takes part: it sends
    // whatever its strip buffer holds (zeros if the draw never ran) and reports its own error afterwards.
    int status = GS_OK;
    // overlap mode: the draw targets and the root's frame alternate between two buffers (the transfer of the previous call may
    // still be reading / writing the other set)
    const uint32_t flip = g->overlap ? (g->calls & 1u) : 0u;
    DevBuf& full = flip ? g->full_alt : g->full;
    DevBuf& target = g->overlap ? g->strip_alt[flip] : m->fb;      // the draw's target; also what a failed draw sends
    if (g->rank == root) status = full.ensure(frame_bytes + 16);
    int st_fb = target.ensure(strip_bytes + 16);
    if (status >= 0 && st_fb < 0) status = st_fb;
    if (status >= 0 && y1 > y0) {
        const int st_draw = gs_mesh_render(m, &c, sorted_host, sorter, render_count, nullptr, g->overlap ? target.p : nullptr, nullptr);
        if (st_draw < 0 && target.p && strip_bytes) (void)hipMemsetAsync(target.p, 0, strip_bytes, g->ctx->stream);
        status = st_draw;
    }
    if (st_fb < 0 || (g->rank == root && !full.p)) {
        // no buffer to send from / receive into: the only case that cannot take part; the peers' watchdog (bench.py) or the
        // caller's own timeout has to end the collective
        return status < 0 ? status : GS_ERR_NOMEM;
    }
    const int st_gather = gs_group_gather_strips(g, y1 > y0 ? target.p : nullptr, full.p, cam->width, row_begin, row_end, root);
    if (status >= 0 && st_gather < 0) status = st_gather;
    if (status >= 0 && g->rank == root && rgba_out_host) {
        hipStream_t rs = g->overlap ? g->coll : g->ctx->stream;    // (behind the transfer)
        GS_HIP(hipMemcpyAsync(rgba_out_host, full.p, frame_bytes, hipMemcpyDeviceToHost, rs));
        GS_HIP(hipStreamSynchronize(rs));
    }
    return status;
}

}  // extern "C"
