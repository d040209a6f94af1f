// A stand-in HIP runtime for the HOST-SIDE sanitizer build of libhicdiff_hip (tools/sanitize/Makefile): "device" memory is host memory,
// kernel launches do nothing, streams / events / graphs are counters.  With it the engine's host code -- planners, the activation pool, the
// lanes and their graph caches, the weight loaders, the trainers' slot tables -- runs end to end on a machine without a GPU under
// AddressSanitizer and UBSan (GPU ASan is not available on the pool).  Only the runtime calls the library makes are provided.
// Test infrastructure: nothing here is part of the product.
#include <hip/hip_runtime_api.h>

#include <cstdlib>
#include <cstring>

namespace {
int g_device = 0;
bool g_capturing = false;
long g_launches = 0, g_graph_launches = 0;
}

extern "C" {
long hipstub_launches() { return g_launches; }
long hipstub_graph_launches() { return g_graph_launches; }

hipError_t hipSetDevice(int d) { g_device = d; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = g_device; return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyFromSymbol(void* d, const void*, size_t n, size_t, hipMemcpyKind) { memset(d, 0, n); return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "hip_stub"; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }

hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)malloc(8); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return hipSuccess; }

hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { g_capturing = true; return hipSuccess; }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { g_capturing = false; *g = (hipGraph_t)malloc(8); return hipSuccess; }
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t, hipGraphNode_t*, char*, size_t) { *e = (hipGraphExec_t)malloc(8); return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t g) { free(g); return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t e) { free(e); return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { ++g_graph_launches; return hipSuccess; }

// what hipcc's host stubs of the kernels call
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t) { ++g_launches; return hipSuccess; }
hipError_t __hipPushCallConfiguration(dim3, dim3, size_t, hipStream_t) { return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* sh, hipStream_t* st) { *g = dim3(1); *b = dim3(1); *sh = 0; *st = nullptr; return hipSuccess; }
void** __hipRegisterFatBinary(const void*) { static void* h; return &h; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, char*, int, size_t, int, int) {}
}
