// csrc/dss_async.cpp -- Part 0 of include/dss_hip.h: the plumbing a host needs to keep several calls of this library in
// flight at once -- streams, events and page-locked host memory as plain handles -- without binding a HIP runtime itself.
// (Python hosts use PyTorch-ROCm's streams and events for the same purpose; these entry points are for hosts without it
// and for memory kinds torch does not hand out.)
#include <algorithm>

#include "dss_host.h"

extern "C" void *dss_stream_create(void)
{
    if (dss_ensure_device()) return nullptr;
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { dss_set_error("hipStreamCreate failed"); return nullptr; }
    return (void *)s;
}

extern "C" void dss_stream_destroy(void *hip_stream)
{
    if (hip_stream) hipStreamDestroy((hipStream_t)hip_stream);
}

extern "C" int dss_stream_synchronize(void *hip_stream)
{
    DSS_HIP_CHECK(hipStreamSynchronize((hipStream_t)hip_stream));
    return DSS_OK;
}

extern "C" void *dss_event_create(void)
{
    if (dss_ensure_device()) return nullptr;
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { dss_set_error("hipEventCreate failed"); return nullptr; }
    return (void *)e;
}

extern "C" void dss_event_destroy(void *event)
{
    if (event) hipEventDestroy((hipEvent_t)event);
}

extern "C" int dss_event_record(void *event, void *hip_stream)
{
    if (!event) { dss_set_error("null event"); return DSS_EINVAL; }
    DSS_HIP_CHECK(hipEventRecord((hipEvent_t)event, (hipStream_t)hip_stream));
    return DSS_OK;
}

// 1 = everything recorded before the event has finished, 0 = not yet
extern "C" int dss_event_query(void *event)
{
    if (!event) { dss_set_error("null event"); return DSS_EINVAL; }
    hipError_t e = hipEventQuery((hipEvent_t)event);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) return 0;
    dss_set_error("hipEventQuery failed: %s", hipGetErrorString(e));
    return DSS_ENODEV;
}

extern "C" int dss_event_synchronize(void *event)
{
    if (!event) { dss_set_error("null event"); return DSS_EINVAL; }
    DSS_HIP_CHECK(hipEventSynchronize((hipEvent_t)event));
    return DSS_OK;
}

extern "C" int dss_stream_wait_event(void *hip_stream, void *event)
{
    if (!event) { dss_set_error("null event"); return DSS_EINVAL; }
    DSS_HIP_CHECK(hipStreamWaitEvent((hipStream_t)hip_stream, (hipEvent_t)event, 0));
    return DSS_OK;
}

// Page-locked host memory for results that come back asynchronously (PCM of a finished segment).  cached != 0: ordinary
// cacheable pages (hipHostMallocNonCoherent) -- the CPU reads them at memory speed; the device's writes are visible once the
// copy's event has completed, which is the only time the host looks.  cached == 0: coherent (fine-grained) pages.
extern "C" void *dss_host_alloc(size_t bytes, int cached)
{
    if (dss_ensure_device()) return nullptr;
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, cached ? hipHostMallocNonCoherent : hipHostMallocDefault) != hipSuccess) {
        dss_set_error("hipHostMalloc(%zu) failed", bytes);
        return nullptr;
    }
    return p;
}

extern "C" void dss_host_free(void *p)
{
    if (p) hipHostFree(p);
}

// Device -> page-locked host memory as a KERNEL (16-byte stores straight into the mapped host pages), not as a DMA copy.
// A hipMemcpyAsync queued behind a long kernel parks in a DMA-engine queue until that kernel has finished -- and every other
// copy the process issues meanwhile, the tick's blocking packet upload included, queues up behind it there: measured, the
// tick's 0.1 ms host-to-device copy took 137 ms while a vocoder launch with a PCM copy behind it was in flight.  A kernel
// waits in its own stream only.
__global__ void __launch_bounds__(256) dss_copy_out_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16,
                                                           const unsigned char *__restrict__ src_tail, unsigned char *__restrict__ dst_tail,
                                                           int tail)
{
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n16; k += (size_t)gridDim.x * 256) dst[k] = src[k];
    if (blockIdx.x == 0 && (int)threadIdx.x < tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}

extern "C" int dss_memcpy_d2h_async(void *host_dst, const void *d_src, size_t bytes, void *hip_stream)
{
    if (!host_dst || !d_src) { dss_set_error("null argument"); return DSS_EINVAL; }
    if (((uintptr_t)host_dst | (uintptr_t)d_src) & 15) { dss_set_error("dss_memcpy_d2h_async: both pointers must be 16-byte aligned"); return DSS_EINVAL; }
    if (!bytes) return DSS_OK;
    void *dev_view = nullptr;                       // the device's address of the page-locked block (fails for pageable memory)
    DSS_HIP_CHECK(hipHostGetDevicePointer(&dev_view, host_dst, 0));
    const size_t n16 = bytes / 16;
    const int tail = (int)(bytes - n16 * 16);
    const unsigned grid = (unsigned)std::min<size_t>(std::max<size_t>((n16 + 255) / 256, 1), 512);
    hipLaunchKernelGGL(dss_copy_out_kernel, dim3(grid), dim3(256), 0, (hipStream_t)hip_stream, (const uint4 *)d_src, (uint4 *)dev_view, n16,
                       (const unsigned char *)d_src + n16 * 16, (unsigned char *)dev_view + n16 * 16, tail);
    DSS_HIP_CHECK(hipGetLastError());
    return DSS_OK;
}
