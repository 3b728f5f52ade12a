// postprocess.cpp -- C ABI for the fixed (ahead-of-time compiled) gfx950 helper kernels.
//
// ptl_average_images: the GPU form of the reference's average_images (src/main.rs:645-722), the
// motion-blur step of the video pipeline (src/main.rs:1787-1817).  The kernel is built by
// `make kernels` (hipcc --genco --offload-arch=gfx950, portal_amd/csrc/kernels/average_images.hip)
// into portal_amd/kernels/average_images.hsaco next to this library and loaded with hipModuleLoadData.
#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../../include/portal_amd.h"
#include "hip_api.h"
#include "internal.h"

using namespace ptl;

namespace {

constexpr int kMaxSubframes = 64;        // PTL_MAX_SUBFRAMES in average_images.hip: pointers in the kernel arguments
constexpr int kMaxSubframesTable = 256;  // beyond: a pointer table in device memory; 256 is where the exact multiply-high mean ends

struct LoadedKernel {
    hip::hipModule_t module = nullptr;
    hip::hipFunction_t fn = nullptr;
    hip::hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

std::string library_dir() {
    Dl_info info;
    if (dladdr(reinterpret_cast<void*>(&library_dir), &info) == 0 || !info.dli_fname) return ".";
    std::string path = info.dli_fname;
    size_t p = path.rfind('/');
    return p == std::string::npos ? "." : path.substr(0, p);
}

// one module per (device, kernel file)
int load_kernel(int device, const char* file, const char* entry, LoadedKernel** out) {
    static std::mutex mu;
    static std::map<std::string, LoadedKernel> cache;
    std::lock_guard<std::mutex> lock(mu);
    std::string key = std::to_string(device) + ":" + file + ":" + entry;
    auto it = cache.find(key);
    if (it != cache.end()) {
        *out = &it->second;
        return PTL_OK;
    }
    std::string err;
    const hip::Runtime* rt = hip::runtime(&err);
    if (!rt) {
        set_last_error(err);
        return PTL_ERR_NO_DEVICE;
    }
    std::string path = library_dir() + "/kernels/" + file;
    std::ifstream f(path, std::ios::binary);
    if (!f) {
        set_last_error("missing gfx950 code object `" + path + "`: run `make kernels` (no CPU fallback)");
        return PTL_ERR_INVALID;
    }
    std::vector<char> code((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    LoadedKernel k;
    if (rt->hipSetDevice(device) != 0 || rt->hipModuleLoadData(&k.module, code.data()) != 0 ||
        rt->hipModuleGetFunction(&k.fn, k.module, entry) != 0) {
        set_last_error("cannot load `" + path + "` on device " + std::to_string(device));
        return PTL_ERR_HIP;
    }
    rt->hipEventCreate(&k.ev0);
    rt->hipEventCreate(&k.ev1);
    *out = &cache.emplace(key, k).first->second;
    return PTL_OK;
}

}  // namespace

extern "C" int ptl_average_images(int device, const void* const* frames_rgba8, int n_frames, void* out_rgba8, int width, int height,
                                  void* stream, float* elapsed_ms) {
    if (!frames_rgba8 || !out_rgba8 || n_frames < 1 || n_frames > kMaxSubframesTable || width <= 0 || height <= 0) return PTL_ERR_INVALID;
    for (int k = 0; k < n_frames; ++k)
        if (!frames_rgba8[k] || (reinterpret_cast<uintptr_t>(frames_rgba8[k]) & 15u)) return PTL_ERR_INVALID;
    if (reinterpret_cast<uintptr_t>(out_rgba8) & 15u) return PTL_ERR_INVALID;
    LoadedKernel* k = nullptr;
    const char* variant = std::getenv("PTL_AVERAGE_IMAGES_HSACO");  // tuning only: another build of the same kernel (tools/average_variants.py)
    const bool table = n_frames > kMaxSubframes;
    int rc = load_kernel(device, variant && *variant ? variant : "average_images.hsaco", table ? "ptl_average_images_table_kernel" : "ptl_average_images_kernel", &k);
    if (rc != PTL_OK) return rc;
    const hip::Runtime* rt = hip::runtime(nullptr);
    rt->hipSetDevice(device);
    struct {
        const void* frame[kMaxSubframes];
    } list{};
    for (int i = 0; i < n_frames && !table; ++i) list.frame[i] = frames_rgba8[i];
    long n_px = (long)width * height, n_vec = n_px / 4;
    int n = n_frames;
    void* dev_table = nullptr;
    if (table) {  // rare (the reference's clips use <= 16): a pointer table per call, freed once the launch has gone through the stream
        if (rt->hipMalloc(&dev_table, sizeof(void*) * (size_t)n_frames) != 0 ||
            rt->hipMemcpyAsync(dev_table, frames_rgba8, sizeof(void*) * (size_t)n_frames, hip::kMemcpyHostToDevice, stream) != 0) {
            if (dev_table) rt->hipFree(dev_table);
            set_last_error("average_images: cannot stage the sub-frame pointer table");
            return PTL_ERR_HIP;
        }
    }
    void* args_list[] = {&list, &n, &out_rgba8, &n_px};
    void* args_table[] = {&dev_table, &n, &out_rgba8, &n_px};
    void** args = table ? args_table : args_list;
    long blocks = std::max(1L, (n_vec + 255) / 256);
    long cap = 256 * 16;  // grid-stride beyond 16 workgroups per CU
    if (const char* c = std::getenv("PTL_AVERAGE_IMAGES_GRID_CAP")) cap = std::atol(c) > 0 ? std::atol(c) : cap;
    if (blocks > cap) blocks = cap;
    if (elapsed_ms) rt->hipEventRecord(k->ev0, stream);
    int err = rt->hipModuleLaunchKernel(k->fn, (unsigned)blocks, 1, 1, 256, 1, 1, 0, stream, args, nullptr);
    if (err != 0) {
        if (dev_table) rt->hipFree(dev_table);
        set_last_error(std::string("hipModuleLaunchKernel(average_images): ") + rt->hipGetErrorString(err));
        return PTL_ERR_HIP;
    }
    if (elapsed_ms) {
        rt->hipEventRecord(k->ev1, stream);
        rt->hipEventSynchronize(k->ev1);
        rt->hipEventElapsedTime(elapsed_ms, k->ev0, k->ev1);
    }
    if (dev_table) {
        rt->hipStreamSynchronize(stream);  // hipFree would wait for the device anyway
        rt->hipFree(dev_table);
    }
    return PTL_OK;
}

// Device frame buffers for callers that keep frames on the GPU between kernels (the video pipeline: sub-frames ->
// ptl_average_images -> one download).  The reference's counterpart is the macroquad render target and
// Texture2D::get_texture_data() (src/main.rs:1041-1042,1803-1816).
extern "C" int ptl_device_alloc(int device, size_t bytes, void** out) {
    if (!out || bytes == 0) return PTL_ERR_INVALID;
    std::string err;
    const hip::Runtime* rt = hip::runtime(&err);
    if (!rt) {
        set_last_error(err);
        return PTL_ERR_NO_DEVICE;
    }
    int e = rt->hipSetDevice(device);
    if (e == 0) e = rt->hipMalloc(out, bytes);
    if (e != 0) {
        set_last_error(std::string("hipMalloc: ") + rt->hipGetErrorString(e));
        return PTL_ERR_HIP;
    }
    return PTL_OK;
}

extern "C" int ptl_device_free(void* p) {
    const hip::Runtime* rt = hip::runtime(nullptr);
    if (!rt) return PTL_ERR_NO_DEVICE;
    return rt->hipFree(p) == 0 ? PTL_OK : PTL_ERR_HIP;
}

extern "C" int ptl_device_download(void* host_dst, const void* device_src, size_t bytes, void* stream) {
    if (!host_dst || !device_src) return PTL_ERR_INVALID;
    const hip::Runtime* rt = hip::runtime(nullptr);
    if (!rt) return PTL_ERR_NO_DEVICE;
    int e = rt->hipMemcpyAsync(host_dst, device_src, bytes, hip::kMemcpyDeviceToHost, stream);
    if (e == 0) e = rt->hipStreamSynchronize(stream);
    if (e != 0) {
        set_last_error(std::string("hipMemcpy(D2H): ") + rt->hipGetErrorString(e));
        return PTL_ERR_HIP;
    }
    return PTL_OK;
}

// Page-locked host memory: a download into it runs at PCIe speed (a 4K RGBA8 frame in < 1 ms instead of ~10 ms pageable).
extern "C" int ptl_host_alloc(size_t bytes, void** out) {
    if (!out || bytes == 0) return PTL_ERR_INVALID;
    std::string err;
    const hip::Runtime* rt = hip::runtime(&err);
    if (!rt) {
        set_last_error(err);
        return PTL_ERR_NO_DEVICE;
    }
    int e = rt->hipHostMalloc(out, bytes, 0);
    if (e != 0) {
        set_last_error(std::string("hipHostMalloc: ") + rt->hipGetErrorString(e));
        rt->hipGetLastError();
        return PTL_ERR_HIP;
    }
    return PTL_OK;
}

extern "C" int ptl_host_free(void* p) {
    const hip::Runtime* rt = hip::runtime(nullptr);
    if (!rt) return PTL_ERR_NO_DEVICE;
    return rt->hipHostFree(p) == 0 ? PTL_OK : PTL_ERR_HIP;
}

// Streams and events for callers that overlap downloads with tracing (the video pipeline: the copy of frame i runs on its
// own stream while frame i+1 is traced).  Thin, 1:1 over HIP; a stream is created non-blocking (no implicit ordering
// against the default stream), an event without timing.
namespace {
int hip_status(const hip::Runtime* rt, int e, const char* what) {
    if (e == 0) return PTL_OK;
    set_last_error(std::string(what) + ": " + rt->hipGetErrorString(e));
    rt->hipGetLastError();
    return PTL_ERR_HIP;
}
}  // namespace

extern "C" int ptl_stream_create(int device, void** stream) {
    if (!stream) return PTL_ERR_INVALID;
    std::string err;
    const hip::Runtime* rt = hip::runtime(&err);
    if (!rt) {
        set_last_error(err);
        return PTL_ERR_NO_DEVICE;
    }
    int e = rt->hipSetDevice(device);
    if (e == 0) e = rt->hipStreamCreateWithFlags(stream, hip::kStreamNonBlocking);
    return hip_status(rt, e, "hipStreamCreate");
}
extern "C" int ptl_stream_destroy(void* stream) {
    const hip::Runtime* rt = hip::runtime(nullptr);
    return rt ? hip_status(rt, rt->hipStreamDestroy(stream), "hipStreamDestroy") : PTL_ERR_NO_DEVICE;
}
extern "C" int ptl_event_create(int device, void** event) {
    if (!event) return PTL_ERR_INVALID;
    const hip::Runtime* rt = hip::runtime(nullptr);
    if (!rt) return PTL_ERR_NO_DEVICE;
    int e = rt->hipSetDevice(device);
    if (e == 0) e = rt->hipEventCreateWithFlags(event, hip::kEventDisableTiming);
    return hip_status(rt, e, "hipEventCreate");
}
extern "C" int ptl_event_destroy(void* event) {
    const hip::Runtime* rt = hip::runtime(nullptr);
    return rt ? hip_status(rt, rt->hipEventDestroy(event), "hipEventDestroy") : PTL_ERR_NO_DEVICE;
}
extern "C" int ptl_event_record(void* event, void* stream) {
    const hip::Runtime* rt = hip::runtime(nullptr);
    return rt ? hip_status(rt, rt->hipEventRecord(event, stream), "hipEventRecord") : PTL_ERR_NO_DEVICE;
}
extern "C" int ptl_event_synchronize(void* event) {
    const hip::Runtime* rt = hip::runtime(nullptr);
    return rt ? hip_status(rt, rt->hipEventSynchronize(event), "hipEventSynchronize") : PTL_ERR_NO_DEVICE;
}
extern "C" int ptl_stream_wait_event(void* stream, void* event) {
    const hip::Runtime* rt = hip::runtime(nullptr);
    return rt ? hip_status(rt, rt->hipStreamWaitEvent(stream, event, 0), "hipStreamWaitEvent") : PTL_ERR_NO_DEVICE;
}
extern "C" int ptl_device_download_async(void* host_dst, const void* device_src, size_t bytes, void* stream) {
    if (!host_dst || !device_src) return PTL_ERR_INVALID;
    const hip::Runtime* rt = hip::runtime(nullptr);
    if (!rt) return PTL_ERR_NO_DEVICE;
    return hip_status(rt, rt->hipMemcpyAsync(host_dst, device_src, bytes, hip::kMemcpyDeviceToHost, stream), "hipMemcpyAsync(D2H)");
}

// A strided device-to-device copy on `stream` (hipMemcpy2DAsync, direction from the pointers): `rows` rows of `width_bytes`, source
// and destination pitches in bytes.  With a packed shard as the source (pitch = one 8-row block) and another GPU's frame as the
// destination (pitch = G blocks, mapped through ptl_ipc_open or peer access) ONE such copy gathers a rank's shard over its xGMI link
// with the SDMA engine and de-interleaves it in the same transfer.
extern "C" int ptl_device_copy2d_async(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width_bytes, size_t rows, void* stream) {
    if (!dst || !src || width_bytes == 0 || dst_pitch < width_bytes || src_pitch < width_bytes) return PTL_ERR_INVALID;
    if (rows == 0) return PTL_OK;
    const hip::Runtime* rt = hip::runtime(nullptr);
    if (!rt) return PTL_ERR_NO_DEVICE;
    return hip_status(rt, rt->hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, width_bytes, rows, hip::kMemcpyDefault, stream), "hipMemcpy2DAsync");
}

// Frame buffers shared between the processes of a node (one process per GPU): the destination rank allocates the full frame
// with ptl_device_alloc, exports it, and every other rank maps it into its own address space; their render kernels then store
// their row blocks straight into the destination GPU's HBM over xGMI (ptl_frame.in_place), no gather, no de-interleave copy.
// 1:1 over hipIpcGetMemHandle / hipIpcOpenMemHandle / hipIpcCloseMemHandle.  The pointer must come from ptl_device_alloc
// (a whole hipMalloc allocation), and a handle cannot be opened by the process that exported it.
extern "C" int ptl_ipc_export(void* device_ptr, unsigned char handle[PTL_IPC_HANDLE_BYTES]) {
    if (!device_ptr || !handle) return PTL_ERR_INVALID;
    std::string err;
    const hip::Runtime* rt = hip::runtime(&err);
    if (!rt) {
        set_last_error(err);
        return PTL_ERR_NO_DEVICE;
    }
    hip::IpcMemHandle h;
    static_assert(sizeof h == PTL_IPC_HANDLE_BYTES, "hipIpcMemHandle_t is 64 bytes");
    int e = rt->hipIpcGetMemHandle(&h, device_ptr);
    if (e == 0) std::memcpy(handle, &h, sizeof h);
    return hip_status(rt, e, "hipIpcGetMemHandle");
}

extern "C" int ptl_ipc_open(int device, const unsigned char handle[PTL_IPC_HANDLE_BYTES], void** device_ptr) {
    if (!handle || !device_ptr) return PTL_ERR_INVALID;
    std::string err;
    const hip::Runtime* rt = hip::runtime(&err);
    if (!rt) {
        set_last_error(err);
        return PTL_ERR_NO_DEVICE;
    }
    hip::IpcMemHandle h;
    std::memcpy(&h, handle, sizeof h);
    int e = rt->hipSetDevice(device);
    if (e == 0) e = rt->hipIpcOpenMemHandle(device_ptr, h, hip::kIpcMemLazyEnablePeerAccess);
    return hip_status(rt, e, "hipIpcOpenMemHandle");
}

extern "C" int ptl_ipc_close(void* device_ptr) {
    const hip::Runtime* rt = hip::runtime(nullptr);
    return rt ? hip_status(rt, rt->hipIpcCloseMemHandle(device_ptr), "hipIpcCloseMemHandle") : PTL_ERR_NO_DEVICE;
}
