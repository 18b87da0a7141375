// Device-to-device exchange without a collective call (gfx950, xGMI): the buffers, handles and flag words behind
// sharded.py's exchange mode "p2p".
//
// The table-sharded forward has ONE exchange step (dlrm_s_pytorch.py:543-570, extend_distributed.py:389-465: every rank pools
// its tables for the full batch, one all-to-all turns "local tables x full batch" into "all tables x local batch").  Through
// RCCL that step costs ~24 us of stream time per call before a byte is on the wire (tools/a2a_cost.py), more than the 20 us
// interaction it feeds.  Here the pooling kernel writes each peer's block STRAIGHT INTO THAT PEER'S RECEIVE BUFFER (IPC-mapped
// over xGMI: point-to-point links, one hop) and the hand-over is two flag words per (peer, slot):
//   ready[src][slot] = k   written by the source behind its pooling kernel: "use k of your slot holds my block"
//   free [dst][slot] = k   written by the consumer behind its interaction kernel: "I am through with use k: overwrite it"
// Buffers and flags are fine-grained device memory (coherent at system scope: a peer's writes are visible to this device's
// loads without a cache flush), handles cross the process group once at start-up.  The kernels below only signal (system-
// scope release stores) and wait (system-scope acquire loads, bounded); the data path is evs_embedding_bag_sum_p2p
// (evs_gather.hip: the sharded pooling launch with a per-peer address table).
#include "evs_common.h"
#include <hip/hip_runtime.h>

namespace evs {

struct P2pSyncArgs {
    unsigned *sig[64];        // words to write (peers' flag blocks), n_sig of them
    const unsigned *wait[64]; // words to wait for (this rank's own flag block), n_wait of them
    unsigned sig_value, wait_value;
    int n_sig, n_wait;
    int *err;                 // sticky: a wait ran out of patience
    long long spins;
};

// one wave: lane i signals word i, then waits for word i.  Everything queued on the stream in front of this kernel is
// complete and visible device-wide at its start (kernel boundary); the release store makes it visible system-wide before
// the flag is, the acquire fence behind the wait keeps the next kernel's loads behind the peers' data.  The wait is bounded (~2 s): a peer that died must not hang the GPU.
__global__ void __launch_bounds__(64) p2p_sync_kernel(const P2pSyncArgs a) {
    const int i = (int)threadIdx.x;
    // (the release store carries the write-back of what this device wrote; no separate fence in front of it -- every
    //  agent- / system-scope fence is an L2 write-back on this part, tools/atomic_probe.hip)
    if (i < a.n_sig) __hip_atomic_store(a.sig[i], a.sig_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (i < a.n_wait) {
        long long n = 0;
        // relaxed polls (an acquire load per poll would invalidate the caches under whatever else runs on the device), ONE
        // acquire fence when the word is there; wrap-safe comparison: the flags count uses of a slot
        while ((int)(__hip_atomic_load(a.wait[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - a.wait_value) < 0) {
            __builtin_amdgcn_s_sleep(8);
            if (++n > a.spins) { atomicOr(a.err, 2); break; }
        }
    }
    if (a.n_wait > 0) __atomic_thread_fence(__ATOMIC_ACQUIRE);   // (system scope: the default of the builtin)
}

}  // namespace evs

extern "C" int evs_p2p_alloc(void **out, int64_t bytes) {
    using namespace evs;
    EVS_REQUIRE(out && bytes > 0, "evs_p2p_alloc: bad argument");
    void *p = nullptr;
    if (hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        set_error("evs_p2p_alloc: hipExtMallocWithFlags(%lld bytes, fine-grained) failed", (long long)bytes);
        return EVS_ENOMEM;
    }
    EVS_HIP_CHECK(hipMemset(p, 0, (size_t)bytes));
    *out = p;
    return EVS_OK;
}

extern "C" int evs_p2p_free(void *p) {
    if (p) (void)hipFree(p);
    return EVS_OK;
}

extern "C" int evs_p2p_ipc_export(void *p, void *handle64) {
    using namespace evs;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the handle travels as 64 bytes");
    EVS_REQUIRE(p && handle64, "evs_p2p_ipc_export: NULL argument");
    EVS_HIP_CHECK(hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t *>(handle64), p));
    return EVS_OK;
}

extern "C" int evs_p2p_ipc_open(const void *handle64, void **out) {
    using namespace evs;
    EVS_REQUIRE(handle64 && out, "evs_p2p_ipc_open: NULL argument");
    hipIpcMemHandle_t h;
    __builtin_memcpy(&h, handle64, sizeof h);
    EVS_HIP_CHECK(hipIpcOpenMemHandle(out, h, hipIpcMemLazyEnablePeerAccess));
    return EVS_OK;
}

extern "C" int evs_p2p_ipc_close(void *p) {
    using namespace evs;
    if (p) EVS_HIP_CHECK(hipIpcCloseMemHandle(p));
    return EVS_OK;
}

extern "C" int evs_p2p_sync(int n_sig, uint32_t *const *sig, uint32_t sig_value, int n_wait, const uint32_t *const *wait,
                            uint32_t wait_value, void *stream) {
    using namespace evs;
    EVS_REQUIRE(n_sig >= 0 && n_sig <= 64 && n_wait >= 0 && n_wait <= 64, "evs_p2p_sync: at most 64 words each way");
    EVS_REQUIRE((n_sig == 0 || sig) && (n_wait == 0 || wait), "evs_p2p_sync: NULL argument");
    if (n_sig == 0 && n_wait == 0) return EVS_OK;
    P2pSyncArgs a;
    for (int i = 0; i < 64; i++) { a.sig[i] = i < n_sig ? sig[i] : nullptr; a.wait[i] = i < n_wait ? wait[i] : nullptr; }
    a.sig_value = sig_value; a.wait_value = wait_value; a.n_sig = n_sig; a.n_wait = n_wait;
    a.err = index_error_flag();
    if (!a.err) return EVS_EHIP;
    static const long long spins = getenv("EVS_P2P_SPINS") ? atoll(getenv("EVS_P2P_SPINS")) : 4000000ll;   // x ~0.5 us per look
    a.spins = spins;
    hipLaunchKernelGGL(p2p_sync_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), a);
    EVS_HIP_CHECK(hipGetLastError());
    return EVS_OK;
}

// ---- stream-ordered signal words: hand-overs between two streams of ONE device without an event --------------------------
// An event wait between two streams goes through the host-side scheduler on this stack (0.175-0.18 ms per batch in the
// prefetcher's loop against 0.160 with signal words; far more when the issuing core has been idle).  A signal word is written and waited for
// by the command processors themselves (hipStreamWriteValue32 / hipStreamWaitValue32, ">= value" wait): the copy stream of
// inference_loop.Prefetcher tells the compute stream "batch k is in its slot", the compute stream tells it "slot free".
extern "C" int evs_signal_alloc(void **out) {
    using namespace evs;
    EVS_REQUIRE(out, "evs_signal_alloc: NULL argument");
    int dev = 0, ok = 0;
    EVS_HIP_CHECK(hipGetDevice(&dev));
    if (hipDeviceGetAttribute(&ok, hipDeviceAttributeCanUseStreamWaitValue, dev) != hipSuccess || !ok) {
        (void)hipGetLastError();
        set_error("evs_signal_alloc: this device has no stream wait-value operations");
        return EVS_ESTATE;
    }
    void *p = nullptr;
    if (hipExtMallocWithFlags(&p, 8, hipMallocSignalMemory) != hipSuccess) {
        (void)hipGetLastError();
        set_error("evs_signal_alloc: hipExtMallocWithFlags(hipMallocSignalMemory) failed");
        return EVS_ENOMEM;
    }
    EVS_HIP_CHECK(hipMemset(p, 0, 8));
    *out = p;
    return EVS_OK;
}

extern "C" int evs_signal_free(void *p) {
    if (p) (void)hipFree(p);
    return EVS_OK;
}

extern "C" int evs_stream_write_value(void *stream, void *signal, uint32_t value) {
    using namespace evs;
    EVS_REQUIRE(signal, "evs_stream_write_value: NULL signal");
    EVS_HIP_CHECK(hipStreamWriteValue32(reinterpret_cast<hipStream_t>(stream), signal, value, 0));
    return EVS_OK;
}

extern "C" int evs_stream_wait_value(void *stream, void *signal, uint32_t value) {
    using namespace evs;
    EVS_REQUIRE(signal, "evs_stream_wait_value: NULL signal");
    EVS_HIP_CHECK(hipStreamWaitValue32(reinterpret_cast<hipStream_t>(stream), signal, value, hipStreamWaitValueGte, 0xffffffffu));
    return EVS_OK;
}
