#ifndef _GNU_SOURCE
#define _GNU_SOURCE  // dlmopen
#endif
// hip_api.cpp -- see hip_api.h.
#include "hip_api.h"

#include <stdlib.h>

#include <dlfcn.h>
#include <link.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace ptl::hip {
namespace {

// Path of an already-loaded shared object whose file name starts with `stem` ("" if none).
std::string loaded_library_path(const char* stem) {
    struct Ctx {
        const char* stem;
        std::string found;
    } ctx{stem, ""};
    dl_iterate_phdr(
        [](struct dl_phdr_info* info, size_t, void* data) -> int {
            auto* c = static_cast<Ctx*>(data);
            if (!info->dlpi_name || !*info->dlpi_name) return 0;
            const char* base = std::strrchr(info->dlpi_name, '/');
            base = base ? base + 1 : info->dlpi_name;
            if (std::strncmp(base, c->stem, std::strlen(c->stem)) == 0) {
                c->found = info->dlpi_name;
                return 1;
            }
            return 0;
        },
        &ctx);
    return ctx.found;
}

// `isolated`: load into a link-map namespace of its own (dlmopen).  For the COMPILER: hiprtc finds its code generator
// (libamd_comgr, i.e. LLVM) by soname, so inside a PyTorch process -- which has loaded its own, older comgr -- /opt/rocm's hiprtc
// would silently compile with PyTorch's LLVM, and the same source would give different code objects with and without
// `import torch` (measured: 109 800 vs 110 568 bytes for scenes/basics.ron).  In a fresh namespace hiprtc resolves its
// dependencies from its own directory: one compiler, whatever else the process has loaded.
// A namespace created by dlmopen gets its own copy of libc, whose `environ` is initialised with the address of the process's
// environment ARRAY at that moment.  The array belongs to the main libc, which reallocates it on setenv/putenv (Python's
// os.environ[...] = ...): the copy is then left pointing at freed memory and the next getenv inside the namespace (comgr and LLVM
// read a dozen variables per compile) crashes.  Give the namespace a snapshot of its own, allocated with ITS malloc, that nobody
// else touches.  (Reproduced and fixed: 400 assignments to os.environ between two compiles segfaulted every time.)
extern "C" char** environ;
void give_namespace_its_own_environ(void* handle) {
    char*** ns_environ = reinterpret_cast<char***>(dlsym(handle, "environ"));
    auto ns_malloc = reinterpret_cast<void* (*)(size_t)>(dlsym(handle, "malloc"));
    if (!ns_environ || !ns_malloc || ns_environ == &environ) return;  // not a separate libc after all
    size_t n = 0;
    while (environ && environ[n]) ++n;
    char** copy = static_cast<char**>(ns_malloc((n + 1) * sizeof(char*)));
    if (!copy) return;
    for (size_t k = 0; k < n; ++k) {
        size_t len = std::strlen(environ[k]) + 1;
        copy[k] = static_cast<char*>(ns_malloc(len));
        if (!copy[k]) {
            copy[k] = nullptr;
            n = k;
            break;
        }
        std::memcpy(copy[k], environ[k], len);
    }
    copy[n] = nullptr;
    *ns_environ = copy;
}

void* open_first(const std::vector<std::string>& candidates, std::string* chosen, std::string* error, bool isolated = false) {
    std::string errs;
    for (const std::string& c : candidates) {
        if (c.empty()) continue;
        void* h = nullptr;
        if (isolated && c[0] == '/') {
            h = dlmopen(LM_ID_NEWLM, c.c_str(), RTLD_NOW);
            if (h) give_namespace_its_own_environ(h);
        }
        if (!h) h = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (h) {
            *chosen = c;
            return h;
        }
        errs += std::string(dlerror()) + "; ";
    }
    if (error) *error = errs;
    return nullptr;
}

template <class F>
bool bind(void* h, const char* name, F& fn, std::string* error) {
    fn = reinterpret_cast<F>(dlsym(h, name));
    if (!fn && error) *error = std::string("missing symbol ") + name;
    return fn != nullptr;
}

std::string dir_of(const std::string& path) {
    size_t p = path.rfind('/');
    return p == std::string::npos ? "" : path.substr(0, p + 1);
}

}  // namespace

const Runtime* runtime(std::string* error) {
    static std::mutex mu;
    static Runtime rt;
    static bool tried = false, ok = false;
    static std::string err;
    std::lock_guard<std::mutex> lock(mu);
    if (!tried) {
        tried = true;
        const char* env = std::getenv("PTL_HIP_LIB");
        std::vector<std::string> candidates = {env ? env : "", loaded_library_path("libamdhip64.so"), "libamdhip64.so",
                                               "/opt/rocm/lib/libamdhip64.so"};
        void* h = open_first(candidates, &rt.path, &err);
        if (h) {
            ok = bind(h, "hipInit", rt.hipInit, &err) && bind(h, "hipGetDeviceCount", rt.hipGetDeviceCount, &err) &&
                 bind(h, "hipSetDevice", rt.hipSetDevice, &err) && bind(h, "hipGetDevice", rt.hipGetDevice, &err) &&
                 bind(h, "hipMalloc", rt.hipMalloc, &err) && bind(h, "hipFree", rt.hipFree, &err) &&
                 bind(h, "hipHostMalloc", rt.hipHostMalloc, &err) && bind(h, "hipHostFree", rt.hipHostFree, &err) &&
                 bind(h, "hipMemcpy", rt.hipMemcpy, &err) && bind(h, "hipMemcpyAsync", rt.hipMemcpyAsync, &err) &&
                 bind(h, "hipMemsetAsync", rt.hipMemsetAsync, &err) && bind(h, "hipStreamSynchronize", rt.hipStreamSynchronize, &err) &&
                 bind(h, "hipDeviceSynchronize", rt.hipDeviceSynchronize, &err) &&
                 bind(h, "hipStreamCreateWithFlags", rt.hipStreamCreateWithFlags, &err) && bind(h, "hipStreamDestroy", rt.hipStreamDestroy, &err) &&
                 bind(h, "hipStreamWaitEvent", rt.hipStreamWaitEvent, &err) && bind(h, "hipEventCreateWithFlags", rt.hipEventCreateWithFlags, &err) && bind(h, "hipEventCreate", rt.hipEventCreate, &err) &&
                 bind(h, "hipEventDestroy", rt.hipEventDestroy, &err) && bind(h, "hipEventRecord", rt.hipEventRecord, &err) &&
                 bind(h, "hipEventSynchronize", rt.hipEventSynchronize, &err) &&
                 bind(h, "hipEventElapsedTime", rt.hipEventElapsedTime, &err) && bind(h, "hipModuleLoadData", rt.hipModuleLoadData, &err) &&
                 bind(h, "hipModuleUnload", rt.hipModuleUnload, &err) && bind(h, "hipModuleGetFunction", rt.hipModuleGetFunction, &err) &&
                 bind(h, "hipModuleGetGlobal", rt.hipModuleGetGlobal, &err) && bind(h, "hipFuncGetAttribute", rt.hipFuncGetAttribute, &err) &&
                 bind(h, "hipModuleLaunchKernel", rt.hipModuleLaunchKernel, &err) && bind(h, "hipGetErrorString", rt.hipGetErrorString, &err) &&
                 bind(h, "hipIpcGetMemHandle", rt.hipIpcGetMemHandle, &err) && bind(h, "hipIpcOpenMemHandle", rt.hipIpcOpenMemHandle, &err) &&
                 bind(h, "hipIpcCloseMemHandle", rt.hipIpcCloseMemHandle, &err) && bind(h, "hipGetLastError", rt.hipGetLastError, &err) &&
                 bind(h, "hipDeviceCanAccessPeer", rt.hipDeviceCanAccessPeer, &err) && bind(h, "hipDeviceEnablePeerAccess", rt.hipDeviceEnablePeerAccess, &err) &&
                 bind(h, "hipMemcpy2DAsync", rt.hipMemcpy2DAsync, &err);
        }
    }
    if (!ok && error) *error = "HIP runtime unavailable: " + err;
    return ok ? &rt : nullptr;
}

const Rtc* rtc(std::string* error) {
    static std::mutex mu;
    static Rtc r;
    static bool tried = false, ok = false;
    static std::string err;
    std::lock_guard<std::mutex> lock(mu);
    if (!tried) {
        tried = true;
        const char* env = std::getenv("PTL_HIPRTC_LIB");
        std::string loaded_rt = loaded_library_path("libamdhip64.so");
        // The compiler is NOT shared state (unlike the runtime above, whose streams and pointers must be the process's own): always
        // take the system toolchain when there is one, so that a kernel is built by the same compiler whether or not PyTorch -- which
        // bundles an older hiprtc -- happens to be loaded.  The code-object cache key names the library that did the work.
        std::vector<std::string> candidates = {env ? env : "", "/opt/rocm/lib/libhiprtc.so", loaded_library_path("libhiprtc.so"),
                                               loaded_rt.empty() ? "" : dir_of(loaded_rt) + "libhiprtc.so", "libhiprtc.so"};
        void* h = open_first(candidates, &r.path, &err, /*isolated=*/std::getenv("PTL_HIPRTC_SHARED") == nullptr);
        if (h) {
            ok = bind(h, "hiprtcCreateProgram", r.hiprtcCreateProgram, &err) && bind(h, "hiprtcCompileProgram", r.hiprtcCompileProgram, &err) &&
                 bind(h, "hiprtcGetProgramLogSize", r.hiprtcGetProgramLogSize, &err) && bind(h, "hiprtcGetProgramLog", r.hiprtcGetProgramLog, &err) &&
                 bind(h, "hiprtcGetCodeSize", r.hiprtcGetCodeSize, &err) && bind(h, "hiprtcGetCode", r.hiprtcGetCode, &err) &&
                 bind(h, "hiprtcDestroyProgram", r.hiprtcDestroyProgram, &err) && bind(h, "hiprtcGetErrorString", r.hiprtcGetErrorString, &err) &&
                 bind(h, "hiprtcVersion", r.hiprtcVersion, &err);
            char resolved[4096];
            r.real_path = ::realpath(r.path.c_str(), resolved) ? resolved : r.path;
        }
    }
    if (!ok && error) *error = "hiprtc unavailable: " + err;
    return ok ? &r : nullptr;
}

const Rccl* rccl(std::string* error) {
    static std::mutex mu;
    static Rccl r;
    static bool tried = false, ok = false;
    static std::string err;
    std::lock_guard<std::mutex> lock(mu);
    if (!tried) {
        tried = true;
        // The collectives run on streams and pointers of the HIP runtime this library is bound to: take the RCCL that goes with it
        const Runtime* rt = runtime(&err);
        if (rt) {
            const char* env = std::getenv("PTL_RCCL_LIB");
            std::string beside = dir_of(loaded_library_path("libamdhip64.so"));
            std::vector<std::string> candidates = {env ? env : "", loaded_library_path("librccl.so"), beside.empty() ? "" : beside + "librccl.so",
                                                   beside.empty() ? "" : beside + "librccl.so.1", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
            void* h = open_first(candidates, &r.path, &err);
            if (h)
                ok = bind(h, "ncclGetVersion", r.ncclGetVersion, &err) && bind(h, "ncclCommInitAll", r.ncclCommInitAll, &err) &&
                     bind(h, "ncclCommDestroy", r.ncclCommDestroy, &err) && bind(h, "ncclGroupStart", r.ncclGroupStart, &err) &&
                     bind(h, "ncclGroupEnd", r.ncclGroupEnd, &err) && bind(h, "ncclSend", r.ncclSend, &err) && bind(h, "ncclRecv", r.ncclRecv, &err) &&
                     bind(h, "ncclGetErrorString", r.ncclGetErrorString, &err);
        }
    }
    if (!ok && error) *error = "RCCL unavailable: " + err;
    return ok ? &r : nullptr;
}

}  // namespace ptl::hip
