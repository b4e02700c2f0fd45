// runtime.hip — HIP runtime shims behind include/vksift_hip.h.
// Replaces the reference's Vulkan plumbing (src/vulkansift/vkenv/vulkan_device.c, vulkan_utils.c):
// device enumeration, device/pinned memory, streams (== queues), events (== fences/semaphores).
#include <hip/hip_runtime.h>
#include <roctracer/roctx.h>

#include <cstdio>
#include <cstring>

#include "vksift_hip.h"

__global__ void k_post_words(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, unsigned long long n)
{
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    dst[i] = src[i];
}

extern "C"
{

  uint32_t vksift_hip_abi_version(void) { return VKSIFT_HIP_ABI_VERSION; }

  /* development / test knobs of the launch shims (vksift_hip.h: VKSIFT_TUNE_*): 0 = the built-in choice, WIDE_MASK -1 */
  static int g_tune[VKSIFT_TUNE_COUNT] = {0, -1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  int vksift_hip_tune(int knob, int value)
  {
    if (knob < 0 || knob >= VKSIFT_TUNE_COUNT)
      return -1;
    g_tune[knob] = value;
    return 0;
  }
  int vksift_hip_tune_get(int knob) { return (knob >= 0 && knob < VKSIFT_TUNE_COUNT) ? g_tune[knob] : 0; }

  int vksift_hip_init(void)
  {
    hipError_t e = hipInit(0);
    if (e != hipSuccess)
      return (int)e;
    int n = 0;
    e = hipGetDeviceCount(&n);
    if (e != hipSuccess)
      return (int)e;
    return n > 0 ? 0 : (int)hipErrorNoDevice;
  }

  int vksift_hip_device_count(void)
  {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
      return 0;
    return n;
  }

  int vksift_hip_device_name(int idx, char *out256)
  {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, idx);
    if (e != hipSuccess)
      return (int)e;
    std::memset(out256, 0, 256);
    std::snprintf(out256, 256, "%s (%s)", prop.name, prop.gcnArchName);
    return 0;
  }

  int vksift_hip_set_device(int idx) { return (int)hipSetDevice(idx); }

  size_t vksift_hip_device_free_mem(void)
  {
    size_t f = 0, t = 0;
    if (hipMemGetInfo(&f, &t) != hipSuccess)
      return 0;
    return f;
  }

  void *vksift_hip_malloc(size_t bytes)
  {
    void *p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess)
      return nullptr;
    return p;
  }
  void vksift_hip_free(void *p)
  {
    if (p)
      (void)hipFree(p);
  }
  void *vksift_hip_host_malloc(size_t bytes)
  {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess)
      return nullptr;
    return p;
  }
  /* page-lock caller memory (hipHostRegister) so that device-to-host copies into it are real DMA transfers; 1 / 0 / <0 */
  int vksift_hip_host_register(void *p, size_t bytes) { return (int)hipHostRegister(p, bytes, hipHostRegisterDefault); }
  int vksift_hip_host_unregister(void *p) { return (int)hipHostUnregister(p); }
  int vksift_hip_is_pinned(const void *p)
  {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess)
    {
      (void)hipGetLastError(); /* plain pageable memory: the query fails, and the sticky error must not surface in a later launch check */
      return 0;
    }
    return a.type == hipMemoryTypeHost ? 1 : 0;
  }

  void vksift_hip_host_free(void *p)
  {
    if (p)
      (void)hipHostFree(p);
  }

  vksift_hip_stream vksift_hip_stream_create(void)
  {
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess)
      return nullptr;
    return (vksift_hip_stream)s;
  }
  void vksift_hip_stream_destroy(vksift_hip_stream s)
  {
    if (s)
      (void)hipStreamDestroy((hipStream_t)s);
  }
  int vksift_hip_stream_sync(vksift_hip_stream s) { return (int)hipStreamSynchronize((hipStream_t)s); }
  int vksift_hip_stream_busy(vksift_hip_stream s)
  {
    hipError_t e = hipStreamQuery((hipStream_t)s);
    if (e == hipSuccess)
      return 0;
    if (e == hipErrorNotReady)
      return 1;
    return -(int)e;
  }

  vksift_hip_event vksift_hip_event_create(void)
  {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess)
      return nullptr;
    return (vksift_hip_event)e;
  }
  void vksift_hip_event_destroy(vksift_hip_event e)
  {
    if (e)
      (void)hipEventDestroy((hipEvent_t)e);
  }
  int vksift_hip_event_record(vksift_hip_event e, vksift_hip_stream s) { return (int)hipEventRecord((hipEvent_t)e, (hipStream_t)s); }
  int vksift_hip_event_sync(vksift_hip_event e) { return (int)hipEventSynchronize((hipEvent_t)e); }
  int vksift_hip_event_busy(vksift_hip_event e)
  {
    hipError_t r = hipEventQuery((hipEvent_t)e);
    if (r == hipSuccess)
      return 0;
    if (r == hipErrorNotReady)
      return 1;
    return -(int)r;
  }
  float vksift_hip_event_elapsed_ms(vksift_hip_event a, vksift_hip_event b)
  {
    float ms = -1.f;
    if (hipEventElapsedTime(&ms, (hipEvent_t)a, (hipEvent_t)b) != hipSuccess)
      return -1.f;
    return ms;
  }
  int vksift_hip_stream_wait_event(vksift_hip_stream s, vksift_hip_event e) { return (int)hipStreamWaitEvent((hipStream_t)s, (hipEvent_t)e, 0); }

  /* ---- hipGraph capture of a launch sequence issued to `s` (and to the streams it forks through events) ---- */
  int vksift_hip_capture_begin(vksift_hip_stream s) { return (int)hipStreamBeginCapture((hipStream_t)s, hipStreamCaptureModeRelaxed); }
  int vksift_hip_capture_end(vksift_hip_stream s, vksift_hip_graph *out)
  {
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    *out = nullptr;
    hipError_t e = hipStreamEndCapture((hipStream_t)s, &g);
    if (e != hipSuccess)
      return (int)e;
    e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess)
      return (int)e;
    *out = (vksift_hip_graph)ge;
    return 0;
  }
  int vksift_hip_graph_launch(vksift_hip_graph g, vksift_hip_stream s) { return (int)hipGraphLaunch((hipGraphExec_t)g, (hipStream_t)s); }
  void vksift_hip_graph_destroy(vksift_hip_graph g)
  {
    if (g)
      (void)hipGraphExecDestroy((hipGraphExec_t)g);
  }

  int vksift_hip_memcpy_h2d(void *dst, const void *src, size_t n, vksift_hip_stream s)
  {
    return n ? (int)hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, (hipStream_t)s) : 0;
  }
  int vksift_hip_memcpy_d2h(void *dst, const void *src, size_t n, vksift_hip_stream s)
  {
    return n ? (int)hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, (hipStream_t)s) : 0;
  }
  int vksift_hip_memcpy_d2d(void *dst, const void *src, size_t n, vksift_hip_stream s)
  {
    return n ? (int)hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, (hipStream_t)s) : 0;
  }
  int vksift_hip_memcpy2d_d2h(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width_bytes, size_t height, vksift_hip_stream s)
  {
    return (int)hipMemcpy2DAsync(dst, dpitch, src, spitch, width_bytes, height, hipMemcpyDeviceToHost, (hipStream_t)s);
  }
  /* Small read-backs that DEPEND on queued kernels (feature counts, match counts): a copy-engine transfer with an unmet
   * dependency parks in the engine's in-order ring and every later device-to-host copy of the process — the download of the
   * previous buffer, on whatever stream — waits behind it until those kernels have finished (measured: a 20 MB pinned copy
   * on a fresh stream takes 0.4 ms on an idle GPU and 22 ms behind a queued 512-frame detection).  A kernel storing into the
   * mapped pinned allocation keeps the dependency in the compute queue where it belongs. */
  int vksift_hip_post_words(uint32_t *host_mapped_dst, const uint32_t *src, size_t n_words, vksift_hip_stream s)
  {
    if (!n_words)
      return 0;
    const unsigned blocks = (unsigned)((n_words + 255) / 256);
    hipLaunchKernelGGL(k_post_words, dim3(blocks), dim3(256), 0, (hipStream_t)s, host_mapped_dst, src, (unsigned long long)n_words);
    return (int)hipGetLastError();
  }
  int vksift_hip_memset(void *dst, int value, size_t n, vksift_hip_stream s) { return n ? (int)hipMemsetAsync(dst, value, n, (hipStream_t)s) : 0; }

  const char *vksift_hip_error_string(int err) { return hipGetErrorString((hipError_t)(err < 0 ? -err : err)); }

  void vksift_hip_range_push(const char *name) { roctxRangePushA(name); }
  void vksift_hip_range_pop(void) { roctxRangePop(); }
}
