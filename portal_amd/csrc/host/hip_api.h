// hip_api.h -- the slice of the HIP runtime and hiprtc this library uses, bound at run time.
//
// libportal_amd.so does not link libamdhip64: inside a PyTorch process the runtime that is
// already loaded (torch ships its own copy) must be the one whose streams and device pointers
// we are handed, and in a stand-alone process (the CLI) /opt/rocm's is used.  Binding through
// dlopen/dlsym keeps one library working in both, and lets `ptl_kernel_compile(device=-1)`
// run hiprtc on a machine without any GPU (the build container).
#pragma once
#include <cstddef>
#include <string>

namespace ptl::hip {

using hipError_t = int;
using hipStream_t = void*;
using hipEvent_t = void*;
using hipModule_t = void*;
using hipFunction_t = void*;
using hipDeviceptr_t = void*;
using hiprtcProgram = void*;
using hiprtcResult = int;

struct IpcMemHandle {  // hipIpcMemHandle_t: 64 opaque bytes, passed BY VALUE to hipIpcOpenMemHandle
    char reserved[64];
};

struct Runtime {
    std::string path;
    hipError_t (*hipInit)(unsigned);
    hipError_t (*hipGetDeviceCount)(int*);
    hipError_t (*hipSetDevice)(int);
    hipError_t (*hipGetDevice)(int*);
    hipError_t (*hipMalloc)(void**, size_t);
    hipError_t (*hipFree)(void*);
    hipError_t (*hipHostMalloc)(void**, size_t, unsigned);
    hipError_t (*hipHostFree)(void*);
    hipError_t (*hipMemcpy)(void*, const void*, size_t, int);
    hipError_t (*hipMemcpyAsync)(void*, const void*, size_t, int, hipStream_t);
    hipError_t (*hipMemsetAsync)(void*, int, size_t, hipStream_t);
    hipError_t (*hipStreamSynchronize)(hipStream_t);
    hipError_t (*hipStreamCreateWithFlags)(hipStream_t*, unsigned);
    hipError_t (*hipStreamDestroy)(hipStream_t);
    hipError_t (*hipStreamWaitEvent)(hipStream_t, hipEvent_t, unsigned);
    hipError_t (*hipEventCreateWithFlags)(hipEvent_t*, unsigned);
    hipError_t (*hipDeviceSynchronize)();
    hipError_t (*hipEventCreate)(hipEvent_t*);
    hipError_t (*hipEventDestroy)(hipEvent_t);
    hipError_t (*hipEventRecord)(hipEvent_t, hipStream_t);
    hipError_t (*hipEventSynchronize)(hipEvent_t);
    hipError_t (*hipEventElapsedTime)(float*, hipEvent_t, hipEvent_t);
    hipError_t (*hipModuleLoadData)(hipModule_t*, const void*);
    hipError_t (*hipModuleUnload)(hipModule_t);
    hipError_t (*hipModuleGetFunction)(hipFunction_t*, hipModule_t, const char*);
    hipError_t (*hipModuleGetGlobal)(hipDeviceptr_t*, size_t*, hipModule_t, const char*);
    hipError_t (*hipFuncGetAttribute)(int*, int, hipFunction_t);
    hipError_t (*hipModuleLaunchKernel)(hipFunction_t, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned,
                                        hipStream_t, void**, void**);
    const char* (*hipGetErrorString)(hipError_t);
    hipError_t (*hipIpcGetMemHandle)(IpcMemHandle*, void*);
    hipError_t (*hipIpcOpenMemHandle)(void**, IpcMemHandle, unsigned);
    hipError_t (*hipIpcCloseMemHandle)(void*);
    hipError_t (*hipGetLastError)();  // also CLEARS the thread's sticky error: call after a failure that was expected
    hipError_t (*hipDeviceCanAccessPeer)(int*, int, int);
    hipError_t (*hipDeviceEnablePeerAccess)(int, unsigned);
    hipError_t (*hipMemcpy2DAsync)(void*, size_t, const void*, size_t, size_t, size_t, int, hipStream_t);
};
constexpr int kFuncAttrSharedSizeBytes = 1, kFuncAttrLocalSizeBytes = 3, kFuncAttrNumRegs = 4;  // hipFunction_attribute
constexpr unsigned kStreamNonBlocking = 1, kEventDisableTiming = 2, kIpcMemLazyEnablePeerAccess = 1;
constexpr int kMemcpyHostToDevice = 1;
constexpr int kMemcpyDeviceToHost = 2;
constexpr int kMemcpyDefault = 4;  // direction (and peer routing) from the pointers' own attributes
constexpr int kErrorPeerAccessAlreadyEnabled = 704;

struct Rtc {
    std::string path;
    std::string real_path;  // symlinks resolved: ".../libhiprtc.so.7.2.70200" -- the toolchain's identity (hiprtcVersion() is a constant 9.0)
    hiprtcResult (*hiprtcCreateProgram)(hiprtcProgram*, const char*, const char*, int, const char**, const char**);
    hiprtcResult (*hiprtcCompileProgram)(hiprtcProgram, int, const char**);
    hiprtcResult (*hiprtcGetProgramLogSize)(hiprtcProgram, size_t*);
    hiprtcResult (*hiprtcGetProgramLog)(hiprtcProgram, char*);
    hiprtcResult (*hiprtcGetCodeSize)(hiprtcProgram, size_t*);
    hiprtcResult (*hiprtcGetCode)(hiprtcProgram, char*);
    hiprtcResult (*hiprtcDestroyProgram)(hiprtcProgram*);
    const char* (*hiprtcGetErrorString)(hiprtcResult);
    hiprtcResult (*hiprtcVersion)(int*, int*);
};

// RCCL (the NCCL API on ROCm), for the collective transport of layer 3 (multigpu.cpp).  Bound like the runtime: the copy the process has
// already loaded (PyTorch ships one next to its HIP runtime), else the one beside the HIP runtime in use, else the system's.
struct NcclUniqueId {
    char internal[128];
};
using ncclComm_t = void*;
constexpr int kNcclSuccess = 0, kNcclUint8 = 1;
struct Rccl {
    std::string path;
    int (*ncclGetVersion)(int*);
    int (*ncclCommInitAll)(ncclComm_t*, int, const int*);
    int (*ncclCommDestroy)(ncclComm_t);
    int (*ncclGroupStart)();
    int (*ncclGroupEnd)();
    int (*ncclSend)(const void*, size_t, int, int, ncclComm_t, hipStream_t);
    int (*ncclRecv)(void*, size_t, int, int, ncclComm_t, hipStream_t);
    const char* (*ncclGetErrorString)(int);
};

// nullptr (and *error filled) if the library cannot be found / lacks a symbol.
const Runtime* runtime(std::string* error = nullptr);
const Rtc* rtc(std::string* error = nullptr);
const Rccl* rccl(std::string* error = nullptr);

}  // namespace ptl::hip
