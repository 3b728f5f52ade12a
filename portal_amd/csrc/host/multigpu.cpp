// multigpu.cpp -- layer 3 of the C ABI: ONE frame across the GPUs of one node, from one host process, without Python or
// torch (the draw it shards is SceneRenderer::draw_texture, src/main.rs:1411-1428).
//
// Pixels are independent, so the frame shards with no exchange while tracing: rank g of G renders the 8-row blocks b with
// b % G == g (interleaved: cost is spatially clustered -- portal interiors take many trips, walls one).  What is left is
// getting every rank's rows into ONE frame buffer on devices[0].  Three transports, same bytes:
//
//   PTL_GROUP_PEER_STORES   rank g's kernel stores its rows straight into the frame in devices[0]'s HBM (peer access inside
//                           the process, hipDeviceEnablePeerAccess; ptl_frame.in_place = 1).  The 128-byte row stores of
//                           the trace kernel travel over the direct xGMI link g -> 0 while the kernel is still tracing:
//                           no staging shard, no gather, no de-interleave pass.
//   PTL_GROUP_COPY_GATHER   every rank renders a packed shard in its own HBM; ONE strided peer copy per rank
//                           (hipMemcpy2DAsync: source pitch = one 8-row block, destination pitch = G blocks) moves it over
//                           the same link with the SDMA engine and puts every block where it belongs -- the gather and the
//                           de-interleave in one transfer.  This is what an RCCL gather to one root decomposes into
//                           (G - 1 point-to-point transfers into rank 0, no ring), minus the collective launch.
//   PTL_GROUP_RCCL_GATHER   the transport BASELINE.json's north star names, spelled with the collective library itself: packed shards
//                           as above, then ONE RCCL group -- ncclSend(shard -> rank 0) on every rank's communicator and stream,
//                           ncclRecv x G on rank 0's -- i.e. ncclGather as RCCL's own documentation composes it, and one strided
//                           device-local copy per shard on rank 0 to put the blocks where they belong.  Communicators come from
//                           ncclCommInitAll (one process, G devices); librccl is bound with dlopen like the HIP runtime (hip_api.cpp),
//                           so nothing links it and a box without it still has the other two transports.  Devices must be distinct
//                           (RCCL refuses a device listed twice), G = 1 is a self send / receive.
//
// One thread drives all ranks: launches are asynchronous, each rank has its own non-blocking stream on its own device, and
// the call returns after every rank's event has completed.  The same device may be listed more than once (rehearsal of the
// N-rank control flow on a 1-GPU box; tests/test_gpu_parity.py).  bench.py's multi-process variant (torch.distributed, one
// process per GPU, RCCL) lives in portal_amd/parallel.py; the CLI's `render-frame --gpus N` uses this file.
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../../include/portal_amd.h"
#include "hip_api.h"
#include "internal.h"

using namespace ptl;

struct ptl_frame_group {
    std::vector<int> devices;
    std::vector<ptl_renderer*> renderers;
    std::vector<hip::hipStream_t> streams;       // per rank: the trace kernels
    std::vector<hip::hipStream_t> comm_streams;  // per rank: what moves its rows to devices[0] (peer copy, RCCL send / receive, de-interleave)
    // Two frames may be in flight (ptl_frame_group_submit / _wait): the transfer of frame n runs on the comm streams behind its kernels
    // while the compute streams already trace frame n + 1 into the other slot's buffers ("gather of frame n overlaps tracing
    // of frame n+1 on a second stream").
    struct Slot {
        std::vector<hip::hipEvent_t> begin, end;  // around rank g's trace kernel (compute stream)
        std::vector<hip::hipEvent_t> sent;        // rank g's rows have left its shard (comm stream; peer stores: = end)
        hip::hipEvent_t assembled = nullptr;      // RCCL gather: the de-interleave copies on devices[0] have finished
        std::vector<void*> shards;                // PTL_GROUP_COPY_GATHER / RCCL_GATHER: packed rows of rank g in devices[g]'s memory
        void* frame = nullptr;                    // the assembled RGBA8 frame, devices[0]
        void* gathered = nullptr;                 // RCCL gather: the G packed shards side by side on devices[0], before the de-interleave copies
        bool in_flight = false, used = false;
        int ticket = -1;
    } slots[2];
    size_t shard_bytes = 0;
    void* frame = nullptr;      // the frame handed out last (one of the slots'), for ptl_frame_group_download
    size_t frame_bytes = 0;
    int width = 0, height = 0;
    int transport = PTL_GROUP_PEER_STORES;
    int next_ticket = 0;
    std::vector<hip::ncclComm_t> comms;  // PTL_GROUP_RCCL_GATHER: one communicator per rank (ncclCommInitAll)
};

namespace {

int hip_fail(const hip::Runtime* rt, int e, const char* what) {
    if (e == 0) return PTL_OK;
    set_last_error(std::string(what) + ": " + rt->hipGetErrorString(e) + " (" + std::to_string(e) + ")");
    rt->hipGetLastError();
    return PTL_ERR_HIP;
}

void release_buffers(ptl_frame_group* g, const hip::Runtime* rt) {
    for (auto& slot : g->slots) {
        if (slot.frame) {
            rt->hipSetDevice(g->devices[0]);
            rt->hipFree(slot.frame);
            slot.frame = nullptr;
        }
        for (size_t k = 0; k < slot.shards.size(); ++k)
            if (slot.shards[k]) {
                rt->hipSetDevice(g->devices[k]);
                rt->hipFree(slot.shards[k]);
                slot.shards[k] = nullptr;
            }
        if (slot.gathered) {
            rt->hipSetDevice(g->devices[0]);
            rt->hipFree(slot.gathered);
            slot.gathered = nullptr;
        }
        slot.used = false;
    }
    g->frame = nullptr;
    g->frame_bytes = g->shard_bytes = 0;
}

// everything that has been enqueued for every rank has finished (before buffers are freed or re-sized, and on errors)
void drain_all(ptl_frame_group* g, const hip::Runtime* rt) {
    for (size_t k = 0; k < g->streams.size(); ++k) {
        rt->hipSetDevice(g->devices[k]);
        if (g->streams[k]) rt->hipStreamSynchronize(g->streams[k]);
        if (k < g->comm_streams.size() && g->comm_streams[k]) rt->hipStreamSynchronize(g->comm_streams[k]);
    }
    for (auto& slot : g->slots) slot.in_flight = false;
}

int nccl_fail(const hip::Rccl* nc, int e, const char* what) {
    if (e == hip::kNcclSuccess) return PTL_OK;
    set_last_error(std::string(what) + ": " + nc->ncclGetErrorString(e) + " (" + std::to_string(e) + ")");
    return PTL_ERR_HIP;
}

}  // namespace

extern "C" int ptl_frame_group_create(ptl_scene* scene, const int* devices, int n_devices, const char* asset_root, unsigned flags, int transport,
                                      ptl_frame_group** out, char* log, size_t log_cap) {
    if (!scene || !devices || n_devices < 1 || n_devices > 64 || !out) return PTL_ERR_INVALID;
    if (transport != PTL_GROUP_PEER_STORES && transport != PTL_GROUP_COPY_GATHER && transport != PTL_GROUP_RCCL_GATHER) return PTL_ERR_INVALID;
    *out = nullptr;
    std::string err;
    const hip::Runtime* rt = hip::runtime(&err);
    if (!rt) {
        set_last_error(err);
        return PTL_ERR_NO_DEVICE;
    }
    auto g = std::make_unique<ptl_frame_group>();
    g->transport = transport;
    g->devices.assign(devices, devices + n_devices);
    int rc = PTL_OK;
    // every rank must be able to address devices[0]'s memory (stores, or the peer copy's destination)
    for (int k = 1; k < n_devices && rc == PTL_OK; ++k) {
        if (devices[k] == devices[0]) continue;
        int can = 0;
        rc = hip_fail(rt, rt->hipDeviceCanAccessPeer(&can, devices[k], devices[0]), "hipDeviceCanAccessPeer");
        if (rc == PTL_OK && !can) {
            set_last_error("device " + std::to_string(devices[k]) + " cannot access the memory of device " + std::to_string(devices[0]));
            rc = PTL_ERR_HIP;
        }
        if (rc == PTL_OK) rc = hip_fail(rt, rt->hipSetDevice(devices[k]), "hipSetDevice");
        if (rc == PTL_OK) {
            int e = rt->hipDeviceEnablePeerAccess(devices[0], 0);
            if (e == hip::kErrorPeerAccessAlreadyEnabled) {
                rt->hipGetLastError();
                e = 0;
            }
            rc = hip_fail(rt, e, "hipDeviceEnablePeerAccess");
        }
    }
    for (int k = 0; k < n_devices && rc == PTL_OK; ++k) {
        ptl_renderer* r = nullptr;
        rc = ptl_renderer_create(scene, devices[k], asset_root, flags, &r, log, log_cap);  // the code-object cache makes ranks 1.. a module load
        if (rc != PTL_OK) break;
        g->renderers.push_back(r);
        hip::hipStream_t s = nullptr, c = nullptr;
        rc = hip_fail(rt, rt->hipSetDevice(devices[k]), "hipSetDevice");
        if (rc == PTL_OK) rc = hip_fail(rt, rt->hipStreamCreateWithFlags(&s, hip::kStreamNonBlocking), "hipStreamCreateWithFlags");
        if (rc == PTL_OK) rc = hip_fail(rt, rt->hipStreamCreateWithFlags(&c, hip::kStreamNonBlocking), "hipStreamCreateWithFlags");
        g->streams.push_back(s);
        g->comm_streams.push_back(c);
        for (auto& slot : g->slots) {
            hip::hipEvent_t b = nullptr, e = nullptr, t = nullptr;
            if (rc == PTL_OK) rc = hip_fail(rt, rt->hipEventCreate(&b), "hipEventCreate");
            if (rc == PTL_OK) rc = hip_fail(rt, rt->hipEventCreate(&e), "hipEventCreate");
            if (rc == PTL_OK) rc = hip_fail(rt, rt->hipEventCreateWithFlags(&t, hip::kEventDisableTiming), "hipEventCreateWithFlags");
            slot.begin.push_back(b);
            slot.end.push_back(e);
            slot.sent.push_back(t);
        }
    }
    for (auto& slot : g->slots) slot.shards.assign(n_devices, nullptr);
    if (rc == PTL_OK) {
        rc = hip_fail(rt, rt->hipSetDevice(devices[0]), "hipSetDevice");
        for (auto& slot : g->slots)
            if (rc == PTL_OK) rc = hip_fail(rt, rt->hipEventCreateWithFlags(&slot.assembled, hip::kEventDisableTiming), "hipEventCreateWithFlags");
    }
    if (rc == PTL_OK && transport == PTL_GROUP_RCCL_GATHER) {
        const hip::Rccl* nc = hip::rccl(&err);
        if (!nc) {
            set_last_error(err);
            rc = PTL_ERR_NO_DEVICE;
        } else {
            for (int a = 0; a < n_devices && rc == PTL_OK; ++a)
                for (int b = a + 1; b < n_devices; ++b)
                    if (devices[a] == devices[b]) {
                        set_last_error("PTL_GROUP_RCCL_GATHER: device " + std::to_string(devices[a]) + " is listed twice; an RCCL communicator has one rank per device");
                        rc = PTL_ERR_INVALID;
                        break;
                    }
            if (rc == PTL_OK) {
                g->comms.assign(n_devices, nullptr);
                rc = nccl_fail(nc, nc->ncclCommInitAll(g->comms.data(), n_devices, devices), "ncclCommInitAll");
                if (rc != PTL_OK) g->comms.clear();
            }
        }
    }
    if (rc != PTL_OK) {
        ptl_frame_group_destroy(g.release());
        return rc;
    }
    *out = g.release();
    return PTL_OK;
}

extern "C" int ptl_frame_group_size(const ptl_frame_group* g) { return g ? (int)g->renderers.size() : 0; }
extern "C" ptl_renderer* ptl_frame_group_renderer(ptl_frame_group* g, int rank) {
    return (g && rank >= 0 && rank < (int)g->renderers.size()) ? g->renderers[rank] : nullptr;
}

extern "C" int ptl_frame_group_set_option(ptl_frame_group* g, const char* name, double value) {
    if (!g) return PTL_ERR_INVALID;
    for (ptl_renderer* r : g->renderers)
        if (int rc = ptl_renderer_set_option(r, name, value); rc != PTL_OK) return rc;
    return PTL_OK;
}
extern "C" int ptl_frame_group_use_camera(ptl_frame_group* g, const char* camera) {
    if (!g) return PTL_ERR_INVALID;
    for (ptl_renderer* r : g->renderers)
        if (int rc = ptl_renderer_use_camera(r, camera); rc != PTL_OK) return rc;
    return PTL_OK;
}
extern "C" int ptl_frame_group_set_camera(ptl_frame_group* g, const double look_at[3], double alpha, double beta, double radius) {
    if (!g) return PTL_ERR_INVALID;
    for (ptl_renderer* r : g->renderers)
        if (int rc = ptl_renderer_set_camera(r, look_at, alpha, beta, radius); rc != PTL_OK) return rc;
    return PTL_OK;
}
// SceneRenderer::update on every rank: each keeps its own camera state and runs its own (identical) portal-crossing query,
// so all ranks agree on the camera without exchanging anything.
extern "C" int ptl_frame_group_update(ptl_frame_group* g, double seconds) {
    if (!g) return PTL_ERR_INVALID;
    for (ptl_renderer* r : g->renderers)
        if (int rc = ptl_renderer_update(r, seconds, nullptr, nullptr); rc != PTL_OK) return rc;
    return PTL_OK;
}

extern "C" int ptl_frame_group_submit(ptl_frame_group* g, int width, int height, int* ticket) {
    if (!g || width <= 0 || height <= 0 || !ticket) return PTL_ERR_INVALID;
    const hip::Runtime* rt = hip::runtime(nullptr);
    if (!rt) return PTL_ERR_NO_DEVICE;
    const int n = (int)g->renderers.size();
    const size_t pitch = (size_t)width * 4;
    const int blocks = (height + 7) / 8;
    const size_t frame_bytes = (size_t)blocks * 8 * pitch;  // whole blocks: the strided copy of a ragged last block stays inside
    const size_t shard_bytes = (size_t)((blocks + n - 1) / n) * 8 * pitch;
    const bool packed = g->transport != PTL_GROUP_PEER_STORES;  // ranks render packed shards in their own memory
    ptl_frame_group::Slot& slot = g->slots[g->next_ticket & 1];
    if (slot.in_flight) {
        set_last_error("ptl_frame_group_submit: two frames are in flight already; ptl_frame_group_wait for ticket " + std::to_string(slot.ticket) + " first");
        return PTL_ERR_INVALID;
    }
    if (frame_bytes != g->frame_bytes || (packed && shard_bytes != g->shard_bytes)) {
        for (auto& sl : g->slots)
            if (sl.in_flight) {  // its buffers are about to be re-allocated
                set_last_error("ptl_frame_group_submit: another frame size while ticket " + std::to_string(sl.ticket) + " is in flight; ptl_frame_group_wait for it first");
                return PTL_ERR_INVALID;
            }
        drain_all(g, rt);
        release_buffers(g, rt);
        int rc = PTL_OK;
        for (auto& sl : g->slots) {
            if (rc == PTL_OK) rc = hip_fail(rt, rt->hipSetDevice(g->devices[0]), "hipSetDevice");
            if (rc == PTL_OK) rc = hip_fail(rt, rt->hipMalloc(&sl.frame, frame_bytes), "hipMalloc(frame)");
            if (rc == PTL_OK && g->transport == PTL_GROUP_RCCL_GATHER) rc = hip_fail(rt, rt->hipMalloc(&sl.gathered, shard_bytes * (size_t)n), "hipMalloc(gathered shards)");
            for (int k = 0; k < n && rc == PTL_OK && packed; ++k) {
                rc = hip_fail(rt, rt->hipSetDevice(g->devices[k]), "hipSetDevice");
                if (rc == PTL_OK) rc = hip_fail(rt, rt->hipMalloc(&sl.shards[k], shard_bytes), "hipMalloc(shard)");
            }
        }
        if (rc != PTL_OK) {
            release_buffers(g, rt);
            return rc;
        }
        g->frame_bytes = frame_bytes;
        g->shard_bytes = packed ? shard_bytes : 0;
    }
    g->width = width;
    g->height = height;
    // A failure on rank k must not leave launches in flight into buffers the next call may free: every early return drains first.
    auto fail = [&](int rc) {
        drain_all(g, rt);
        return rc;
    };
    for (int k = 0; k < n; ++k) {
        if (int rc = hip_fail(rt, rt->hipSetDevice(g->devices[k]), "hipSetDevice"); rc != PTL_OK) return fail(rc);
        // this slot's shard of two frames ago must have left before the kernel overwrites it (a no-op wait in the steady state)
        if (slot.used && packed)
            if (int rc = hip_fail(rt, rt->hipStreamWaitEvent(g->streams[k], slot.sent[k], 0), "hipStreamWaitEvent(shard sent)"); rc != PTL_OK) return fail(rc);
        rt->hipEventRecord(slot.begin[k], g->streams[k]);
        if (g->transport == PTL_GROUP_PEER_STORES) {
            ptl_frame f{width, height, k, n, 1};
            if (int rc = ptl_renderer_draw(g->renderers[k], &f, slot.frame, nullptr, nullptr, g->streams[k], nullptr); rc != PTL_OK) return fail(rc);
            rt->hipEventRecord(slot.end[k], g->streams[k]);
            rt->hipEventRecord(slot.sent[k], g->streams[k]);  // the stores ARE the transfer
        } else {
            ptl_frame f{width, height, k, n, 0};
            if (int rc = ptl_renderer_draw(g->renderers[k], &f, slot.shards[k], nullptr, nullptr, g->streams[k], nullptr); rc != PTL_OK) return fail(rc);
            rt->hipEventRecord(slot.end[k], g->streams[k]);
            // the transfer runs on the rank's comm stream, behind this kernel -- the compute stream is free for the next frame's trace
            if (int rc = hip_fail(rt, rt->hipStreamWaitEvent(g->comm_streams[k], slot.end[k], 0), "hipStreamWaitEvent(trace done)"); rc != PTL_OK) return fail(rc);
            const int my_blocks = blocks > k ? (blocks - k + n - 1) / n : 0;
            if (my_blocks > 0 && g->transport == PTL_GROUP_COPY_GATHER) {
                // shard block j -> frame block j * n + k: source pitch one block, destination pitch n blocks, `my_blocks` rows of 8 * pitch bytes
                char* dst = static_cast<char*>(slot.frame) + (size_t)k * 8 * pitch;
                if (int rc = hip_fail(rt, rt->hipMemcpy2DAsync(dst, (size_t)n * 8 * pitch, slot.shards[k], 8 * pitch, 8 * pitch, (size_t)my_blocks,
                                                               hip::kMemcpyDefault, g->comm_streams[k]),
                                      "hipMemcpy2DAsync(shard -> frame)");
                    rc != PTL_OK)
                    return fail(rc);
            }
            if (g->transport == PTL_GROUP_COPY_GATHER) rt->hipEventRecord(slot.sent[k], g->comm_streams[k]);
        }
    }
    if (g->transport == PTL_GROUP_RCCL_GATHER) {
        // ONE collective: every rank sends its packed shard to rank 0 (behind its kernel, on its comm stream), rank 0 receives them side by
        // side -- a gather as RCCL composes it from point-to-point calls in a group; over xGMI that is G - 1 direct transfers into rank 0.
        const hip::Rccl* nc = hip::rccl(nullptr);
        int rc = nccl_fail(nc, nc->ncclGroupStart(), "ncclGroupStart");
        for (int k = 0; k < n && rc == PTL_OK; ++k) rc = nccl_fail(nc, nc->ncclSend(slot.shards[k], shard_bytes, hip::kNcclUint8, 0, g->comms[k], g->comm_streams[k]), "ncclSend(shard)");
        for (int k = 0; k < n && rc == PTL_OK; ++k)
            rc = nccl_fail(nc, nc->ncclRecv(static_cast<char*>(slot.gathered) + (size_t)k * shard_bytes, shard_bytes, hip::kNcclUint8, k, g->comms[0], g->comm_streams[0]), "ncclRecv(shard)");
        int end = nc->ncclGroupEnd();  // always closed, also after a failed call inside the group
        if (rc == PTL_OK) rc = nccl_fail(nc, end, "ncclGroupEnd");
        if (rc != PTL_OK) return fail(rc);
        for (int k = 0; k < n; ++k) {
            rt->hipSetDevice(g->devices[k]);
            rt->hipEventRecord(slot.sent[k], g->comm_streams[k]);
        }
        // de-interleave on rank 0, behind the receives on its comm stream: shard k's block j -> frame block j * n + k
        if (int rc2 = hip_fail(rt, rt->hipSetDevice(g->devices[0]), "hipSetDevice"); rc2 != PTL_OK) return fail(rc2);
        for (int k = 0; k < n; ++k) {
            const int my_blocks = blocks > k ? (blocks - k + n - 1) / n : 0;
            if (my_blocks == 0) continue;
            char* dst = static_cast<char*>(slot.frame) + (size_t)k * 8 * pitch;
            const char* src = static_cast<const char*>(slot.gathered) + (size_t)k * shard_bytes;
            if (int rc2 = hip_fail(rt, rt->hipMemcpy2DAsync(dst, (size_t)n * 8 * pitch, src, 8 * pitch, 8 * pitch, (size_t)my_blocks, hip::kMemcpyDefault, g->comm_streams[0]),
                                   "hipMemcpy2DAsync(gathered shard -> frame)");
                rc2 != PTL_OK)
                return fail(rc2);
        }
        rt->hipEventRecord(slot.assembled, g->comm_streams[0]);
    }
    slot.in_flight = slot.used = true;
    slot.ticket = g->next_ticket++;
    *ticket = slot.ticket;
    return PTL_OK;
}

extern "C" int ptl_frame_group_wait(ptl_frame_group* g, int ticket, void** device_rgba8, float* kernel_ms) {
    if (!g || ticket < 0) return PTL_ERR_INVALID;
    const hip::Runtime* rt = hip::runtime(nullptr);
    if (!rt) return PTL_ERR_NO_DEVICE;
    ptl_frame_group::Slot& slot = g->slots[ticket & 1];
    if (!slot.in_flight || slot.ticket != ticket) {
        set_last_error("ptl_frame_group_wait: no frame with ticket " + std::to_string(ticket) + " is in flight");
        return PTL_ERR_INVALID;
    }
    const int n = (int)g->renderers.size();
    for (int k = 0; k < n; ++k) {
        if (int rc = hip_fail(rt, rt->hipSetDevice(g->devices[k]), "hipSetDevice"); rc != PTL_OK) return rc;
        if (int rc = hip_fail(rt, rt->hipEventSynchronize(slot.sent[k]), "hipEventSynchronize(rank)"); rc != PTL_OK) return rc;
        if (kernel_ms) rt->hipEventElapsedTime(&kernel_ms[k], slot.begin[k], slot.end[k]);
    }
    if (g->transport == PTL_GROUP_RCCL_GATHER) {
        rt->hipSetDevice(g->devices[0]);
        if (int rc = hip_fail(rt, rt->hipEventSynchronize(slot.assembled), "hipEventSynchronize(assembled)"); rc != PTL_OK) return rc;
    }
    slot.in_flight = false;
    g->frame = slot.frame;
    if (device_rgba8) *device_rgba8 = slot.frame;
    return PTL_OK;
}

extern "C" int ptl_frame_group_draw(ptl_frame_group* g, int width, int height, void** device_rgba8, float* kernel_ms) {
    if (!g) return PTL_ERR_INVALID;
    // (a caller that mixes the two forms gets the frames in order: whatever is in flight is finished first)
    for (int older = g->next_ticket - 2; older < g->next_ticket; ++older)
        if (older >= 0 && g->slots[older & 1].in_flight && g->slots[older & 1].ticket == older)
            if (int rc = ptl_frame_group_wait(g, older, nullptr, nullptr); rc != PTL_OK) return rc;
    int ticket = -1;
    if (int rc = ptl_frame_group_submit(g, width, height, &ticket); rc != PTL_OK) return rc;
    return ptl_frame_group_wait(g, ticket, device_rgba8, kernel_ms);
}

extern "C" int ptl_frame_group_download(ptl_frame_group* g, uint8_t* host_rgba8) {
    if (!g || !host_rgba8 || !g->frame) return PTL_ERR_INVALID;
    const hip::Runtime* rt = hip::runtime(nullptr);
    if (int rc = hip_fail(rt, rt->hipSetDevice(g->devices[0]), "hipSetDevice"); rc != PTL_OK) return rc;
    return hip_fail(rt, rt->hipMemcpy(host_rgba8, g->frame, (size_t)g->width * g->height * 4, hip::kMemcpyDeviceToHost), "hipMemcpy(frame)");
}

extern "C" void ptl_frame_group_destroy(ptl_frame_group* g) {
    if (!g) return;
    const hip::Runtime* rt = hip::runtime(nullptr);
    if (rt) drain_all(g, rt);  // before the renderers go: frames may still be in flight
    for (ptl_renderer* r : g->renderers) ptl_renderer_destroy(r);  // a kernel handle waits for its last launch's stream
    g->renderers.clear();
    if (rt && !g->comms.empty()) {  // before the streams the collectives ran on
        if (const hip::Rccl* nc = hip::rccl(nullptr))
            for (hip::ncclComm_t c : g->comms)
                if (c) nc->ncclCommDestroy(c);
        g->comms.clear();
    }
    if (rt) {
        for (size_t k = 0; k < g->streams.size(); ++k) {
            rt->hipSetDevice(g->devices[k]);
            if (g->streams[k]) rt->hipStreamDestroy(g->streams[k]);
            if (k < g->comm_streams.size() && g->comm_streams[k]) rt->hipStreamDestroy(g->comm_streams[k]);
            for (auto& slot : g->slots) {
                if (k < slot.begin.size() && slot.begin[k]) rt->hipEventDestroy(slot.begin[k]);
                if (k < slot.end.size() && slot.end[k]) rt->hipEventDestroy(slot.end[k]);
                if (k < slot.sent.size() && slot.sent[k]) rt->hipEventDestroy(slot.sent[k]);
            }
        }
        for (auto& slot : g->slots)
            if (slot.assembled) rt->hipEventDestroy(slot.assembled);
        release_buffers(g, rt);
    }
    delete g;
}
