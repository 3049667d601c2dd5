// csrc/dss_host.h -- host-side helpers shared by the C ABI's translation units (dss_capi.cpp, dss_async.cpp).
#pragma once

#include "dss_common.h"

// Selects (and on first use picks: LOCAL_RANK, else 0) this thread's device; DSS_ENODEV without one.
int dss_ensure_device(void);

// Small host -> device uploads that must not stall, and must not be overwritten, while earlier calls are still queued.
//
// hipMemcpyAsync from pageable memory may wait for the stream's earlier work (the runtime stages it), which would hold the
// host for the length of a queued vocoder launch; from pinned memory it is asynchronous, but then the pinned words must
// stay untouched until the copy has run.  A ring of K pinned slots, each guarded by an event recorded behind the copies
// issued from it: acquire() hands out the next slot and waits only if the call K calls ago has not reached its copies yet.
struct DssPinnedRing {
    static const int K = 8;
    int *host[K] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev[K] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool pending[K] = {false, false, false, false, false, false, false, false};
    size_t ints = 0;
    int next = 0, cur = -1;

    int init(size_t n_ints)
    {
        ints = n_ints;
        for (int k = 0; k < K; ++k) {
            if (hipHostMalloc((void **)&host[k], n_ints * sizeof(int), hipHostMallocDefault) != hipSuccess) return DSS_ENOMEM;
            if (hipEventCreateWithFlags(&ev[k], hipEventDisableTiming) != hipSuccess) return DSS_ENOMEM;
        }
        return DSS_OK;
    }
    void destroy()
    {
        for (int k = 0; k < K; ++k) {
            if (pending[k] && ev[k]) hipEventSynchronize(ev[k]);
            if (host[k]) hipHostFree(host[k]);
            if (ev[k]) hipEventDestroy(ev[k]);
            host[k] = nullptr; ev[k] = nullptr; pending[k] = false;
        }
    }
    // the slot this call may fill (nullptr on a HIP error)
    int *acquire()
    {
        cur = next;
        next = (next + 1) % K;
        if (pending[cur]) {
            if (hipEventSynchronize(ev[cur]) != hipSuccess) return nullptr;
            pending[cur] = false;
        }
        return host[cur];
    }
    // call after the copies out of the acquired slot have been issued on s
    int commit(hipStream_t s)
    {
        if (cur < 0) return DSS_EINVAL;
        DSS_HIP_CHECK(hipEventRecord(ev[cur], s));
        pending[cur] = true;
        return DSS_OK;
    }
};
