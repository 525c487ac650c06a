// hip/hip_runtime.h -- HOST-ONLY stand-in for the HIP runtime, for sanitizer builds of the library's own host code
// (tests/host_sanitize/Makefile; SURVEY.md section 5: "host tests under -fsanitize=address,undefined").
//
// This is NOT a compatibility layer of the product: libdsp_gpu.so is built by hipcc against the real runtime.  Here
// capi.hip and host_path.hip -- the tap cache, the per-stream workspace sets, the staging ring and its copy threads --
// are compiled by g++ as plain C++ so that AddressSanitizer / UBSan / ThreadSanitizer can watch them on a machine with no
// GPU.  "Device" memory is the host heap, every stream runs its work at once (so a stream is always idle and an event
// has always happened), and the kernels are host functions that touch exactly the bytes the real ones read and write
// (stub_launchers.cpp).
#pragma once

#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>

typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
constexpr hipError_t hipErrorInvalidValue = 1;
constexpr hipError_t hipErrorOutOfMemory = 2;

struct hipStubStream { int id; };
struct hipStubEvent { int recorded; };
typedef hipStubStream *hipStream_t;
typedef hipStubEvent *hipEvent_t;

struct float2 { float x, y; };

enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
enum hipMemoryType { hipMemoryTypeUnregistered, hipMemoryTypeHost, hipMemoryTypeDevice };
struct hipPointerAttribute_t { hipMemoryType type; };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize };
constexpr unsigned hipEventDisableTiming = 2, hipStreamNonBlocking = 1, hipHostMallocDefault = 0, hipHostRegisterDefault = 0;

namespace hipstub {
inline int &device() { static thread_local int d = 0; return d; }
inline std::mutex &pinned_mutex() { static std::mutex m; return m; }
inline std::set<const void *> &pinned() { static std::set<const void *> s; return s; }
}  // namespace hipstub

inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "stub error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDevice(int *d) { *d = hipstub::device(); return hipSuccess; }
inline hipError_t hipSetDevice(int d) { if (d < 0 || d >= 8) return hipErrorInvalidValue; hipstub::device() = d; return hipSuccess; }
inline hipError_t hipGetDeviceCount(int *n) { *n = 8; return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }

inline hipError_t hipMalloc(void **p, size_t bytes) { *p = std::malloc(bytes ? bytes : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <class T> inline hipError_t hipMalloc(T **p, size_t bytes) { return hipMalloc(reinterpret_cast<void **>(p), bytes); }
inline hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void **p, size_t bytes, unsigned) {
    *p = std::malloc(bytes ? bytes : 1);
    if (!*p) return hipErrorOutOfMemory;
    std::lock_guard<std::mutex> lock(hipstub::pinned_mutex());
    hipstub::pinned().insert(*p);
    return hipSuccess;
}
inline hipError_t hipHostFree(void *p) {
    {
        std::lock_guard<std::mutex> lock(hipstub::pinned_mutex());
        hipstub::pinned().erase(p);
    }
    std::free(p);
    return hipSuccess;
}
inline hipError_t hipHostRegister(void *, size_t, unsigned) { return hipSuccess; }
inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *p) {
    std::lock_guard<std::mutex> lock(hipstub::pinned_mutex());
    if (!hipstub::pinned().count(p)) return hipErrorInvalidValue;  // pageable memory, as the real runtime reports it
    a->type = hipMemoryTypeHost;
    return hipSuccess;
}
inline hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind, hipStream_t) {
    std::memmove(dst, src, bytes);
    return hipSuccess;
}
inline hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind) { std::memmove(dst, src, bytes); return hipSuccess; }
inline hipError_t hipMemset(void *p, int v, size_t bytes) { std::memset(p, v, bytes); return hipSuccess; }
inline hipError_t hipMemset2DAsync(void *p, size_t pitch, int v, size_t width, size_t height, hipStream_t) {
    for (size_t r = 0; r < height; ++r) std::memset(static_cast<char *>(p) + r * pitch, v, width);
    return hipSuccess;
}
inline hipError_t hipMemsetAsync(void *p, int v, size_t bytes, hipStream_t) { std::memset(p, v, bytes); return hipSuccess; }

inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = new hipStubStream{1}; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t e, unsigned) { return e ? hipSuccess : hipErrorInvalidValue; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new hipStubEvent{0}; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { if (!e) return hipErrorInvalidValue; e->recorded = 1; return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t e) { return e ? hipSuccess : hipErrorInvalidValue; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 1.0f; return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
