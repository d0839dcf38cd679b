// HOST-ONLY STAND-IN for <hip/hip_runtime.h> - TEST INFRASTRUCTURE (tests/test_host_asan.py), never part of the product.
//
// SURVEY section 5 asks for sanitizer coverage of the host code; GPU AddressSanitizer is not available on this pool.  The host
// half of libkarios_hip.so (api.hip: argument validation, workspace slots, upload tickets, the frame ring; staging.hip: the
// page-locked staging ring and landing arena) is compiled a second time with g++ -fsanitize=address,undefined against THIS
// header: "device" memory is host memory, streams execute at once, events are always complete, kernels (stub_kernels.cpp) do
// nothing but fill their outputs deterministically.  What is exercised is the bookkeeping and every memcpy of the staging paths.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600 };
typedef struct stub_stream *hipStream_t;
typedef struct stub_event *hipEvent_t;
struct stub_stream { int priority; };
struct stub_event { int recorded; };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipHostRegisterDefault = 0 };
enum hipMemoryType { hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2, hipMemoryTypeUnregistered = 0 };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 16 };
struct hipPointerAttribute_t { hipMemoryType type; int device; void *devicePointer, *hostPointer; };
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };

// registry of the page-locked ranges (hipHostMalloc): hipPointerGetAttributes tells them from pageable memory
struct stub_state {
    std::mutex m;
    std::map<const char *, size_t> pinned;
    long device_allocs = 0, host_allocs = 0, pageable_async_copies = 0;
    int fail_malloc_after = -1;      // test knob: the n-th hipMalloc from now fails
};
stub_state &stub();

static inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : e == hipErrorOutOfMemory ? "hipErrorOutOfMemory" : "hipError(stub)"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = getenv("KARIOS_STUB_NO_DEVICE") ? 0 : 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0; *hi = -1; return hipSuccess; }

hipError_t hipMalloc(void **p, size_t n);
hipError_t hipFree(void *p);
hipError_t hipHostMalloc(void **p, size_t n, unsigned flags);
hipError_t hipHostFree(void *p);
hipError_t hipPointerGetAttributes(hipPointerAttribute_t *at, const void *p);
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind kind, hipStream_t s);
hipError_t hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind, hipStream_t s);
static inline hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind k) { return hipMemcpyAsync(dst, src, n, k, nullptr); }
static inline hipError_t hipMemsetAsync(void *dst, int v, size_t n, hipStream_t) { memset(dst, v, n); return hipSuccess; }
static inline hipError_t hipHostRegister(void *, size_t, unsigned) { return hipSuccess; }

static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = new stub_stream{0}; return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int p) { *s = new stub_stream{p}; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t e, unsigned) { return e ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new stub_event{0}; return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new stub_event{0}; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { if (!e) return hipErrorInvalidValue; e->recorded++; return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t e) { return e ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipEventQuery(hipEvent_t e) { return e ? hipSuccess : hipErrorInvalidValue; }   // (the stand-in executes everything at once: always complete)
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = 0.f; return a && b ? hipSuccess : hipErrorInvalidValue; }
