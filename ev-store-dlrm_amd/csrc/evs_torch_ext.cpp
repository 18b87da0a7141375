// PyTorch-ROCm C++ extension over the C ABI of libevstore_hip.so (include/evstore_hip.h): the call path of the plugin
// surface -- apply_emb / interact_features / apply_emb_interact (dlrm_s_pytorch.py:407-461, 483-516, 588-601), the cache
// tier's request / lookup_interact, and the batch-1 EVStore request (dlrm_s_pytorch_C1.py:227-275 -> cache modules) --
// without the Python + ctypes marshalling in front of every launch (9-11 us per call: profiles/r02_sweep.md reads
// 10.6-11.5 us for every batch up to 2 048 samples, where the kernel itself is 4-6 us).
//
// What lives here: argument checks, output allocation, the pointer tables, torch's current HIP stream, ONE call into the
// library.  No arithmetic: every kernel is in libevstore_hip.so and is reached through the same extern "C" entry points a
// C caller uses.  Built in-tree by ev-store-dlrm_amd/_ext_build.py -> lib/_evs_torch_ext.so (g++, host only).
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>
#include <c10/hip/HIPGuard.h>

#include <cstdint>
#include <cstring>
#include <optional>
#include <vector>

#include "../../include/evstore_hip.h"

namespace py = pybind11;

namespace {

// evstore_dlrm_amd._lib.EvsError, handed over by the Python side at import.  A raw, deliberately leaked reference: a static
// py::object would be released by a destructor that runs after the interpreter has been finalised.
PyObject *g_error_class = nullptr;

[[noreturn]] void raise_evs(int code) {
    const char *msg = evs_last_error();
    if (g_error_class) {
        py::object cls = py::reinterpret_borrow<py::object>(g_error_class);
        py::object exc = cls(code, std::string(msg ? msg : ""));
        PyErr_SetObject(g_error_class, exc.ptr());
        throw py::error_already_set();
    }
    throw std::runtime_error(std::string("libevstore_hip: ") + (msg ? msg : "") + " (code " + std::to_string(code) + ")");
}
inline void check(int rc) { if (rc != 0) raise_evs(rc); }

inline void *stream_of(const c10::Device &dev) {
    return static_cast<void *>(c10::hip::getCurrentHIPStream(dev.index()).stream());
}

// The tables of one model (dlrm_ops.EVTables): raw byte tensors kept alive, their device-side addresses and row counts
struct Tables {
    std::vector<at::Tensor> keep;
    std::vector<const void *> ptrs;
    std::vector<int64_t> n_rows;
    int T = 0, d = 0, codec = 32;
    c10::Device device{c10::kCUDA, 0};

    Tables(std::vector<at::Tensor> raws, std::vector<int64_t> dev_ptrs, int d_, int codec_, int device_index)
        : keep(std::move(raws)), d(d_), codec(codec_), device(c10::kCUDA, static_cast<c10::DeviceIndex>(device_index)) {
        TORCH_CHECK(keep.size() == dev_ptrs.size(), "Tables: one address per table");
        T = static_cast<int>(keep.size());
        for (int k = 0; k < T; k++) {
            ptrs.push_back(reinterpret_cast<const void *>(dev_ptrs[k]));
            n_rows.push_back(keep[k].size(0));
        }
    }
};

inline int64_t pairs(int F, bool itself) { return itself ? (int64_t)F * (F + 1) / 2 : (int64_t)F * (F - 1) / 2; }

inline void check_x(const at::Tensor &x, int64_t B, int d) {
    TORCH_CHECK(x.is_cuda() && x.scalar_type() == at::kFloat && x.dim() == 2 && x.size(0) == B && x.size(1) == d &&
                (d == 1 || x.stride(1) == 1), "x must be a (B, d) fp32 device tensor with unit inner stride");
}
inline void check_idx(const at::Tensor &t, const char *name) {
    TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kLong && t.dim() == 2 && t.stride(1) == 1, name,
                " must be a (T, B) int64 device tensor with unit inner stride (dlrm_wrap moves it to the GPU)");
}

// R = interact_features(x, apply_emb(lS_o, lS_i, emb_l)) -- evs_emb_interact_dot_stacked
at::Tensor apply_emb_interact(const Tables &ev, const at::Tensor &x, const std::optional<at::Tensor> &lS_o, const at::Tensor &lS_i,
                              bool itself, std::optional<at::Tensor> out, bool one_index_per_bag, bool check_indices) {
    const int64_t B = x.size(0);
    const int T = ev.T, d = ev.d, F = T + 1;
    check_x(x, B, d);
    check_idx(lS_i, "lS_i");
    TORCH_CHECK(lS_i.size(0) == T, "lS_i has ", lS_i.size(0), " rows, the model has ", T, " tables");
    const bool no_off = one_index_per_bag && lS_i.size(1) == B;
    if (!no_off) {
        TORCH_CHECK(lS_o.has_value(), "lS_o is required unless one_index_per_bag is declared");
        check_idx(*lS_o, "lS_o");
        TORCH_CHECK(lS_o->size(0) == T && lS_o->size(1) == B, "lS_o must be (T, B)");
    }
    const int64_t K = d + pairs(F, itself);
    at::Tensor R = out.has_value() ? *out : at::empty({B, K}, x.options());
    TORCH_CHECK(R.is_cuda() && R.scalar_type() == at::kFloat && R.dim() == 2 && R.size(0) == B && R.size(1) == K && R.is_contiguous(),
                "out must be a contiguous (B, d + P) fp32 device tensor");
    void *st = stream_of(ev.device);
    check(evs_emb_interact_dot_stacked(B, T, d, ev.codec, ev.ptrs.data(), ev.n_rows.data(), x.data_ptr<float>(),
                                       B > 1 ? x.stride(0) : d, lS_i.data_ptr<int64_t>(), lS_i.stride(0), lS_i.size(1),
                                       no_off ? nullptr : lS_o->data_ptr<int64_t>(), no_off ? 0 : lS_o->stride(0), nullptr,
                                       itself ? 1 : 0, R.data_ptr<float>(), st));
    if (check_indices) check(evs_check_index_errors(st));
    return R;
}

// K independent batches in one call (evs_emb_interact_dot_stacked_multi): lists of K tensors, same shapes
std::vector<at::Tensor> apply_emb_interact_multi(const Tables &ev, const std::vector<at::Tensor> &xs,
                                                 const std::optional<std::vector<at::Tensor>> &lS_os, const std::vector<at::Tensor> &lS_is,
                                                 bool itself, std::optional<std::vector<at::Tensor>> outs, bool one_index_per_bag) {
    const int K = static_cast<int>(xs.size());
    TORCH_CHECK(K >= 1 && (int)lS_is.size() == K, "apply_emb_interact_multi: K tensors of each kind");
    const int64_t B = xs[0].size(0);
    const int T = ev.T, d = ev.d, F = T + 1;
    const int64_t Kc = d + pairs(F, itself);
    const bool no_off = one_index_per_bag && lS_is[0].size(1) == B;
    TORCH_CHECK(no_off || (lS_os.has_value() && (int)lS_os->size() == K), "lS_o is required unless one_index_per_bag is declared");
    std::vector<at::Tensor> R;
    std::vector<const float *> xp(K);
    std::vector<const int64_t *> ip(K), op(K);
    std::vector<float *> rp(K);
    for (int k = 0; k < K; k++) {
        check_x(xs[k], B, d);
        check_idx(lS_is[k], "lS_i");
        TORCH_CHECK(lS_is[k].size(0) == T && lS_is[k].size(1) == lS_is[0].size(1) && lS_is[k].stride(0) == lS_is[0].stride(0) &&
                    xs[k].stride(0) == xs[0].stride(0), "apply_emb_interact_multi: every batch must have the same shape and strides");
        if (!no_off) {
            check_idx((*lS_os)[k], "lS_o");
            TORCH_CHECK((*lS_os)[k].size(0) == T && (*lS_os)[k].size(1) == B && (*lS_os)[k].stride(0) == (*lS_os)[0].stride(0), "lS_o must be (T, B)");
            op[k] = (*lS_os)[k].data_ptr<int64_t>();
        }
        TORCH_CHECK(!outs.has_value() || (int)outs->size() == K, "apply_emb_interact_multi: K output tensors");
        at::Tensor r = outs.has_value() ? (*outs)[k] : at::empty({B, Kc}, xs[k].options());
        TORCH_CHECK(r.is_cuda() && r.scalar_type() == at::kFloat && r.dim() == 2 && r.size(0) == B && r.size(1) == Kc && r.is_contiguous(),
                    "out must be a contiguous (B, d + P) fp32 device tensor");
        xp[k] = xs[k].data_ptr<float>(); ip[k] = lS_is[k].data_ptr<int64_t>(); rp[k] = r.data_ptr<float>();
        R.push_back(std::move(r));
    }
    check(evs_emb_interact_dot_stacked_multi(K, B, T, d, ev.codec, ev.ptrs.data(), ev.n_rows.data(), xp.data(), B > 1 ? xs[0].stride(0) : d,
                                             ip.data(), lS_is[0].stride(0), lS_is[0].size(1), no_off ? nullptr : op.data(),
                                             no_off ? 0 : (*lS_os)[0].stride(0), itself ? 1 : 0, rp.data(), stream_of(ev.device)));
    return R;
}

// apply_emb, stacked Criteo layout -> the (T, B, d) buffer whose T slices are the list the reference returns
at::Tensor apply_emb(const Tables &ev, const std::optional<at::Tensor> &lS_o, const at::Tensor &lS_i, bool one_index_per_bag,
                     bool check_indices) {
    const int T = ev.T, d = ev.d;
    check_idx(lS_i, "lS_i");
    TORCH_CHECK(lS_i.size(0) == T, "lS_i has ", lS_i.size(0), " rows, the model has ", T, " tables");
    const int64_t B = lS_o.has_value() ? lS_o->size(1) : lS_i.size(1);
    const bool no_off = one_index_per_bag && lS_i.size(1) == B;
    if (!no_off) {
        TORCH_CHECK(lS_o.has_value(), "lS_o is required unless one_index_per_bag is declared");
        check_idx(*lS_o, "lS_o");
        TORCH_CHECK(lS_o->size(0) == T, "lS_o must be (T, B)");
    }
    at::Tensor buf = at::empty({T, B, d}, at::TensorOptions().dtype(at::kFloat).device(ev.device));
    void *st = stream_of(ev.device);
    check(evs_embedding_bag_sum_stacked(T, B, d, ev.codec, ev.ptrs.data(), ev.n_rows.data(), lS_i.data_ptr<int64_t>(), lS_i.stride(0),
                                        lS_i.size(1), no_off ? nullptr : lS_o->data_ptr<int64_t>(), no_off ? 0 : lS_o->stride(0), nullptr,
                                        buf.data_ptr<float>(), B * d, d, st));
    if (check_indices) check(evs_check_index_errors(st));
    return buf;
}

// apply_emb, list form (the reference's random-data loader: T offsets vectors of B entries, T index vectors of any
// length): the pointer tables built here instead of in Python + ctypes (52 tensors: ~39 us per call there)
at::Tensor apply_emb_list(const Tables &ev, const std::vector<at::Tensor> &lS_o, const std::vector<at::Tensor> &lS_i, bool one_index_per_bag,
                          bool check_indices) {
    const int T = ev.T, d = ev.d;
    TORCH_CHECK((int)lS_o.size() == T && (int)lS_i.size() == T, "apply_emb: one offsets and one index tensor per table");
    const int64_t B = lS_o[0].size(0);
    const int64_t *ip[64], *op[64];
    int64_t nnz[64];
    TORCH_CHECK(T <= 64, "at most 64 tables");
    bool no_off = one_index_per_bag;
    for (int k = 0; k < T; k++) {
        const at::Tensor &i = lS_i[k], &o = lS_o[k];
        TORCH_CHECK(i.is_cuda() && o.is_cuda() && i.scalar_type() == at::kLong && o.scalar_type() == at::kLong && i.dim() == 1 && o.dim() == 1 &&
                    o.size(0) == B && (i.numel() == 0 || i.stride(0) == 1) && (o.numel() == 0 || o.stride(0) == 1),
                    "apply_emb: lS_o[k] (B,) and lS_i[k] (nnz,) must be contiguous int64 device vectors");
        nnz[k] = i.numel();
        ip[k] = i.numel() ? i.data_ptr<int64_t>() : o.data_ptr<int64_t>();   // (an empty tensor has no address: never dereferenced at nnz = 0)
        op[k] = o.data_ptr<int64_t>();
        no_off = no_off && nnz[k] == B;
    }
    at::Tensor buf = at::empty({T, B, d}, at::TensorOptions().dtype(at::kFloat).device(ev.device));
    void *st = stream_of(ev.device);
    check(evs_embedding_bag_sum(T, B, d, ev.codec, ev.ptrs.data(), ev.n_rows.data(), ip, no_off ? nullptr : op, nnz, nullptr,
                                buf.data_ptr<float>(), B * d, d, st));
    if (check_indices) check(evs_check_index_errors(st));
    return buf;
}

// interact_features(x, ly) for "dot": ly as any list of (B, d) fp32 views
at::Tensor interact_dot(const at::Tensor &x, const std::vector<at::Tensor> &ly, bool itself) {
    const int64_t B = x.size(0);
    const int d = static_cast<int>(x.size(1));
    const int F = static_cast<int>(ly.size()) + 1;
    check_x(x, B, d);
    TORCH_CHECK(F <= EVS_MAX_FEATURES, "interact_features: at most ", EVS_MAX_FEATURES, " features");
    const float *ptrs[EVS_MAX_FEATURES];
    int64_t strides[EVS_MAX_FEATURES];
    ptrs[0] = x.data_ptr<float>(); strides[0] = B > 1 ? x.stride(0) : d;
    for (int f = 1; f < F; f++) {
        const at::Tensor &t = ly[f - 1];
        TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kFloat && t.dim() == 2 && t.size(0) == B && t.size(1) == d &&
                    (d == 1 || t.stride(1) == 1), "ly[", f - 1, "] must be a (B, d) fp32 device tensor with unit inner stride");
        ptrs[f] = t.data_ptr<float>(); strides[f] = B > 1 ? t.stride(0) : d;
    }
    at::Tensor R = at::empty({B, d + pairs(F, itself)}, x.options());
    check(evs_interact_dot(B, F, d, ptrs, strides, itself ? 1 : 0, R.data_ptr<float>(), stream_of(x.device())));
    return R;
}

// ... and the common case: ly is still the list apply_emb built = the T slices of ONE (T, B, d) buffer (or strided views
// of a (B, F, d) tile): addressed arithmetically, no per-tensor work
at::Tensor interact_dot_pooled(const at::Tensor &x, int64_t base, int64_t tstride, int64_t bstride, int T, bool itself) {
    const int64_t B = x.size(0);
    const int d = static_cast<int>(x.size(1));
    const int F = T + 1;
    check_x(x, B, d);
    TORCH_CHECK(F <= EVS_MAX_FEATURES, "interact_features: at most ", EVS_MAX_FEATURES, " features");
    const float *ptrs[EVS_MAX_FEATURES];
    int64_t strides[EVS_MAX_FEATURES];
    ptrs[0] = x.data_ptr<float>(); strides[0] = B > 1 ? x.stride(0) : d;
    for (int k = 0; k < T; k++) {
        ptrs[k + 1] = reinterpret_cast<const float *>(base) + tstride * k;
        strides[k + 1] = B > 1 ? bstride : d;
    }
    at::Tensor R = at::empty({B, d + pairs(F, itself)}, x.options());
    check(evs_interact_dot(B, F, d, ptrs, strides, itself ? 1 : 0, R.data_ptr<float>(), stream_of(x.device())));
    return R;
}

// ---- GPU cache tier --------------------------------------------------------------------------------------------------------
inline void *dev_ptr(const at::Tensor &t) {   // device tensor, or a pinned host tensor the kernel reads / writes itself
    if (t.is_cuda()) return t.data_ptr();
    TORCH_CHECK(t.is_pinned() && t.is_contiguous(), "host tensors must be pinned (torch.Tensor.pin_memory) and contiguous");
    void *p = evs_host_device_pointer(t.data_ptr());
    TORCH_CHECK(p, "pinned tensor is not device-accessible");
    return p;
}

void cache_request(int64_t handle, const at::Tensor &rows, const at::Tensor &out, const at::Tensor &hit, int approx_thres, int device_index) {
    TORCH_CHECK(rows.scalar_type() == at::kInt && rows.is_contiguous() && rows.dim() == 2, "rows must be a contiguous (B, T) int32 tensor");
    TORCH_CHECK(out.scalar_type() == at::kFloat && out.is_contiguous() && hit.scalar_type() == at::kByte && hit.is_contiguous() &&
                hit.numel() == rows.numel() && out.numel() % std::max<int64_t>(rows.numel(), 1) == 0,
                "out must be a contiguous (B, T, d) fp32 tensor, hit a contiguous (B, T) uint8 one");
    check(evs_cache_request(reinterpret_cast<evs_cache *>(handle), rows.size(0), static_cast<const int32_t *>(dev_ptr(rows)),
                            static_cast<float *>(dev_ptr(out)), static_cast<uint8_t *>(dev_ptr(hit)), approx_thres,
                            stream_of(c10::Device(c10::kCUDA, static_cast<c10::DeviceIndex>(device_index)))));
}

void cache_lookup_interact(int64_t handle, const at::Tensor &rows, const at::Tensor &x, bool itself, const at::Tensor &out,
                           const at::Tensor &hit) {
    const int64_t B = rows.size(0);
    TORCH_CHECK(rows.is_cuda() && rows.scalar_type() == at::kInt && rows.is_contiguous() && rows.dim() == 2, "rows must be a contiguous (B, T) int32 device tensor");
    check_x(x, B, static_cast<int>(x.size(1)));
    const int64_t T = rows.size(1);
    TORCH_CHECK(out.is_cuda() && out.is_contiguous() && out.scalar_type() == at::kFloat && out.numel() == B * (x.size(1) + pairs((int)T + 1, itself)) &&
                hit.is_cuda() && hit.is_contiguous() && hit.scalar_type() == at::kByte && hit.numel() == B * T,
                "out must be a contiguous (B, d + P) fp32 device tensor, hit a contiguous (B, T) uint8 one");
    check(evs_cache_lookup_interact(reinterpret_cast<evs_cache *>(handle), B, rows.data_ptr<int32_t>(), x.data_ptr<float>(),
                                    B > 1 ? x.stride(0) : x.size(1), itself ? 1 : 0, out.data_ptr<float>(), hit.data_ptr<uint8_t>(),
                                    stream_of(x.device())));
}

// ---- batch-1 EVStore request on the host engine ------------------------------------------------------------------------------
// apply_emb_evstore -> request_to_ev_lfu / _lru / _lfu (dlrm_s_pytorch_C1.py:236-253, EvLFU_C1.py:97-166): element 0 of each
// table's index row -> the exact policy (evs_hostcache_request) -> (hit flags, 26 x Tensor(1, d) with requires_grad) -- the
// rows land in ONE fresh (T, 1, d) tensor, its T views are the list (use_gpu: one host-to-device copy, EvLFU_C1.py:157-161
// makes 26).
// The slices block[0], block[1], ... as independent tensors aliasing the block's storage -- optionally LEAF tensors with
// requires_grad, what the reference builds one torch.FloatTensor([val]) at a time (EvLFU_C1.py:157-159).  unbind() goes
// through the dispatcher and the autograd view machinery once per slice (0.7 us each, 18 us for 26 -- more than the policy
// or the launch they follow); these are plain TensorImpls over the same storage (0.3 us each, Python wrapper included).
std::vector<at::Tensor> slices(const at::Tensor &block, bool requires_grad) {
    TORCH_CHECK(block.dim() >= 2, "slices: at least 2 dimensions");
    const int64_t n = block.size(0), step = block.stride(0), off0 = block.storage_offset();
    const c10::IntArrayRef sizes = block.sizes().slice(1), strides = block.strides().slice(1);
    std::vector<at::Tensor> rows;
    rows.reserve(n);
    const c10::Storage &st = block.storage();
    for (int64_t k = 0; k < n; k++) {
        at::Tensor t = at::detail::make_tensor<c10::TensorImpl>(c10::Storage(st), block.key_set(), block.dtype());
        c10::TensorImpl *impl = t.unsafeGetTensorImpl();
        impl->set_storage_offset(off0 + k * step);
        impl->set_sizes_and_strides(sizes, strides);
        if (requires_grad) t.set_requires_grad(true);
        rows.push_back(std::move(t));
    }
    return rows;
}

// lS_i may still be on the device (dlrm_wrap moved it there, dlrm_s_pytorch_C1.py:233-234 brings it back with .cpu(): a
// pageable device-to-host copy, 20-25 us): then element 0 of each row comes over through a PINNED staging buffer (one
// asynchronous copy + one stream synchronise), and with use_gpu the rows go back through pinned staging too (asynchronous:
// the consumers are stream-ordered behind it; a staging slot is re-used four requests later, and every request
// synchronises the stream first).
static void first_ids(const at::Tensor &lS_i_in, int T, int32_t (&ids)[64]) {
    TORCH_CHECK(lS_i_in.dim() >= 1 && lS_i_in.size(0) == T && lS_i_in.numel() >= T, "lS_i must have one non-empty row per table");
    TORCH_CHECK(T <= 64, "at most 64 tables");
    at::Tensor lS_i = lS_i_in;
    if (lS_i_in.is_cuda()) {
        static at::Tensor stage_ids;
        if (!stage_ids.defined()) stage_ids = at::empty({64}, at::TensorOptions().dtype(at::kLong)).pin_memory();
        TORCH_CHECK(lS_i_in.scalar_type() == at::kLong, "a device lS_i must be int64");
        at::Tensor col = lS_i_in.dim() >= 2 ? lS_i_in.select(1, 0) : lS_i_in;
        lS_i = stage_ids.narrow(0, 0, T);
        lS_i.copy_(col, /*non_blocking=*/true);
        c10::hip::getCurrentHIPStream(lS_i_in.device().index()).synchronize();
    }
    if (lS_i.scalar_type() == at::kLong) {
        const int64_t *p = lS_i.data_ptr<int64_t>();
        const int64_t s0 = lS_i.stride(0);
        for (int k = 0; k < T; k++) ids[k] = static_cast<int32_t>(p[k * s0]);   // element 0 of each row
    } else {
        TORCH_CHECK(lS_i.scalar_type() == at::kInt, "lS_i must be int64 or int32");
        const int32_t *p = lS_i.data_ptr<int32_t>();
        const int64_t s0 = lS_i.stride(0);
        for (int k = 0; k < T; k++) ids[k] = p[k * s0];
    }
}

// the same body for the GPU engine's resident server (evs_cache_serve_*): the ids as above, the request through the mailbox, the
// rows of the answer's ring slot (HBM) copied once into a fresh block and handed out as its 26 views
py::tuple serve_request_list(int64_t handle, const at::Tensor &lS_i_in, const at::Tensor &ring, int T, int d) {
    uint8_t hit[64];
    int slot = 0;
    evs_cache *c = reinterpret_cast<evs_cache *>(handle);
    const bool by_address = lS_i_in.is_cuda() && lS_i_in.scalar_type() == at::kLong && lS_i_in.dim() >= 1 && lS_i_in.size(0) == T &&
                            lS_i_in.numel() >= T && lS_i_in.device() == ring.device();
    c10::hip::HIPGuard guard(ring.device().index());
    hipStream_t st = c10::hip::getCurrentHIPStream(ring.device().index()).stream();
    // Whatever wrote lS_i ran on this stream, and whatever last used the block allocated below was ordered on it: both have to
    // have finished before the server -- on a stream of its own, ordered by nothing -- reads the one and writes the other
    // (the reference's own copies from pageable memory have)
    int32_t ids[64];
    if (!by_address) first_ids(lS_i_in, T, ids);
    bool idle = hipStreamQuery(st) == hipSuccess;
    if (!idle) {
        (void)hipGetLastError();
        if (by_address) { (void)hipStreamSynchronize(st); idle = true; }
    }
    at::Tensor block;
    if (T <= 26 && idle) {
        // round 6: the server writes the rows INTO the fresh block (no copy out of a ring slot, no event behind it)
        block = at::empty({T, 1, d}, ring.options());
        check(evs_cache_serve_request_to(c, by_address ? nullptr : ids, by_address ? lS_i_in.data_ptr<int64_t>() : nullptr,
                                         by_address ? lS_i_in.stride(0) : 0, block.data_ptr<float>(), hit));
    } else {
        if (by_address) check(evs_cache_serve_request_dev(c, lS_i_in.data_ptr<int64_t>(), lS_i_in.stride(0), hit, &slot));
        else check(evs_cache_serve_request(c, ids, hit, &slot));
        TORCH_CHECK(ring.dim() == 3 && slot >= 0 && slot < ring.size(0) && ring.size(1) == T && ring.size(2) == d, "ring slot out of range");
        block = ring.select(0, slot).clone().unsqueeze(1);
        // the copy above is only ENQUEUED: the slot is handed out again (and overwritten by the server, in host order) only after it has run
        check(evs_cache_serve_consumed(c, slot, st));
    }
    py::list flags;
    bool all = true;
    for (int k = 0; k < T; k++) { flags.append(py::bool_(hit[k] != 0)); all = all && hit[k]; }
    return py::make_tuple(flags, slices(block, true), all);
}

py::tuple hostcache_request_list(int64_t handle, const at::Tensor &lS_i_in, int T, int d, int approx_thres, bool use_gpu, int device_index) {
    int32_t ids[64];
    first_ids(lS_i_in, T, ids);
    at::Tensor block;
    uint8_t hit[64];
    if (use_gpu) {
        // four pinned staging slots in turn; a slot is rewritten only after the copy out of it has finished: an event recorded
        // behind that copy ON THE STREAM IT RAN ON (the rows' device, which need not be lS_i's), waited for before the host
        // touches the slot again -- a CPU lS_i with use_gpu, or a backed-up stream, would otherwise overwrite rows in flight
        static at::Tensor stage_rows[4];
        static hipEvent_t stage_done[4] = {nullptr, nullptr, nullptr, nullptr};
        static bool stage_busy[4] = {false, false, false, false};
        static unsigned next_slot = 0;
        const unsigned slot = next_slot++ & 3u;
        at::Tensor &st = stage_rows[slot];
        if (stage_busy[slot]) { (void)hipEventSynchronize(stage_done[slot]); stage_busy[slot] = false; }
        if (!st.defined() || st.size(0) != T || st.size(2) != d) st = at::empty({T, 1, d}, at::TensorOptions().dtype(at::kFloat)).pin_memory();
        check(evs_hostcache_request(reinterpret_cast<evs_hostcache *>(handle), 1, ids, st.data_ptr<float>(), hit, approx_thres));
        const c10::DeviceIndex dev_i = static_cast<c10::DeviceIndex>(device_index);
        c10::hip::HIPGuard guard(dev_i);
        block = at::empty({T, 1, d}, at::TensorOptions().dtype(at::kFloat).device(c10::Device(c10::kCUDA, dev_i)));
        block.copy_(st, /*non_blocking=*/true);
        if (!stage_done[slot]) (void)hipEventCreateWithFlags(&stage_done[slot], hipEventDisableTiming);
        if (stage_done[slot] && hipEventRecord(stage_done[slot], c10::hip::getCurrentHIPStream(dev_i).stream()) == hipSuccess) stage_busy[slot] = true;
        else c10::hip::getCurrentHIPStream(dev_i).synchronize();
    } else {
        block = at::empty({T, 1, d}, at::TensorOptions().dtype(at::kFloat));
        check(evs_hostcache_request(reinterpret_cast<evs_hostcache *>(handle), 1, ids, block.data_ptr<float>(), hit, approx_thres));
    }
    py::list flags;
    bool all = true;
    for (int k = 0; k < T; k++) { flags.append(py::bool_(hit[k] != 0)); all = all && hit[k]; }
    return py::make_tuple(flags, slices(block, true), all);
}

// ---- the resident dispatcher of the fused launch (evs_emb_interact_serve_*): post one batch / wait for a ticket ----------------
// (checks and the descriptor write without Python in between: a post is ~1 us of host time)
uint64_t serve_post(int64_t handle, const at::Tensor &x, const at::Tensor &lS_o, const at::Tensor &lS_i, const at::Tensor &R, int T, int d, int64_t K) {
    const int64_t B = x.size(0);
    check_x(x, B, d);
    check_idx(lS_i, "lS_i");
    check_idx(lS_o, "lS_o");
    TORCH_CHECK(lS_i.size(0) == T && lS_i.size(1) == B && lS_o.size(0) == T && lS_o.size(1) == B, "lS_o / lS_i must be (T, B)");
    TORCH_CHECK(R.is_cuda() && R.scalar_type() == at::kFloat && R.dim() == 2 && R.size(0) == B && R.size(1) == K && R.is_contiguous(),
                "out must be a contiguous (B, d + P) fp32 device tensor");
    uint64_t ticket = 0;
    check(evs_emb_interact_serve_post(reinterpret_cast<evs_rf_server *>(handle), B, x.data_ptr<float>(), B > 1 ? x.stride(0) : d,
                                      lS_i.data_ptr<int64_t>(), lS_i.stride(0), lS_o.data_ptr<int64_t>(), lS_o.stride(0), R.data_ptr<float>(), &ticket));
    return ticket;
}
// post + wait in one call (the latency path: one batch posted and waited for)
void serve_run(int64_t handle, const at::Tensor &x, const at::Tensor &lS_o, const at::Tensor &lS_i, const at::Tensor &R, int T, int d, int64_t K) {
    const uint64_t ticket = serve_post(handle, x, lS_o, lS_i, R, T, d, K);
    py::gil_scoped_release nogil;
    const int rc = evs_emb_interact_serve_wait(reinterpret_cast<evs_rf_server *>(handle), ticket);
    if (rc) { py::gil_scoped_acquire gil; raise_evs(rc); }
}
void serve_wait(int64_t handle, uint64_t ticket) {
    py::gil_scoped_release nogil;
    const int rc = evs_emb_interact_serve_wait(reinterpret_cast<evs_rf_server *>(handle), ticket);
    if (rc) { py::gil_scoped_acquire gil; raise_evs(rc); }
}

// ---- the all-to-all of the sharded step, issued from here (round 6) -----------------------------------------------------------
// dlrm_s_pytorch.py:543-570 / extend_distributed.py:389-465 exchange "local tables x full batch" for "all tables x local
// batch" with one all_to_all_single; through torch.distributed that call costs 14-38 us of HOST time per step on this stack
// (ProcessGroupNCCL: work objects, event pairs, stream hand-overs), more than the two kernels of the step together.  Here the
// same exchange is ONE call into RCCL -- ncclAllToAllv (RCCL's own entry point; grouped ncclSend / ncclRecv where a build
// lacks it) -- on torch's CURRENT stream, in stream order between the pooling launch and the interaction, over a communicator
// of this extension's own (unique id made on rank 0 and handed round through the process group by the Python side).
// RCCL is reached through the copy torch has already loaded (dlsym on the global scope; no link-time dependency: the
// extension loads where RCCL is absent, and sharded.py then keeps all_to_all_single).
}  // namespace
#include <dlfcn.h>
#include <rccl/rccl.h>
namespace {

struct RcclApi {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllToAllv) AllToAllv = nullptr;   // may stay NULL
    bool ok = false;
};
const RcclApi &rccl() {
    static RcclApi api = [] {
        RcclApi a;
        void *h = dlopen(nullptr, RTLD_NOW);        // the global scope: torch's librccl.so is in it once torch has been imported
        auto sym = [&](const char *name) -> void * {
            void *p = h ? dlsym(h, name) : nullptr;
            if (!p) { void *l = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL); if (l) p = dlsym(l, name); }
            return p;
        };
#define EVS_RCCL_SYM(field, name) a.field = reinterpret_cast<decltype(a.field)>(sym(name))
        EVS_RCCL_SYM(GetUniqueId, "ncclGetUniqueId"); EVS_RCCL_SYM(CommInitRank, "ncclCommInitRank");
        EVS_RCCL_SYM(CommDestroy, "ncclCommDestroy"); EVS_RCCL_SYM(CommAbort, "ncclCommAbort");
        EVS_RCCL_SYM(GetErrorString, "ncclGetErrorString");
        EVS_RCCL_SYM(GroupStart, "ncclGroupStart"); EVS_RCCL_SYM(GroupEnd, "ncclGroupEnd");
        EVS_RCCL_SYM(Send, "ncclSend"); EVS_RCCL_SYM(Recv, "ncclRecv"); EVS_RCCL_SYM(AllToAllv, "ncclAllToAllv");
#undef EVS_RCCL_SYM
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.GroupStart && a.GroupEnd && a.Send && a.Recv;
        return a;
    }();
    return api;
}
inline void rccl_check(ncclResult_t r, const char *what) {
    if (r != ncclSuccess) {
        const char *m = rccl().GetErrorString ? rccl().GetErrorString(r) : "?";
        throw std::runtime_error(std::string("RCCL ") + what + ": " + m);
    }
}
bool rccl_available() { return rccl().ok; }
py::bytes rccl_unique_id() {
    TORCH_CHECK(rccl().ok, "RCCL is not loaded in this process");
    ncclUniqueId id;
    rccl_check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
    return py::bytes(reinterpret_cast<const char *>(&id), sizeof id);
}

// one communicator + the exchanges planned over it (fixed buffers and split lists: a serving loop reuses its staged buffers)
struct DirectA2A {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    c10::DeviceIndex device = 0;
    bool use_v = true;
    struct Plan { const float *send; float *recv; std::vector<size_t> scnt, sdis, rcnt, rdis; };
    std::vector<Plan> plans;

    DirectA2A(const std::string &id_bytes, int rank_, int world_, int device_index, bool allow_alltoallv)
        : rank(rank_), world(world_), device(static_cast<c10::DeviceIndex>(device_index)) {
        TORCH_CHECK(rccl().ok, "RCCL is not loaded in this process");
        ncclUniqueId id;
        TORCH_CHECK(id_bytes.size() == sizeof id, "the unique id has ", id_bytes.size(), " bytes, expected ", sizeof id);
        std::memcpy(&id, id_bytes.data(), sizeof id);
        c10::hip::HIPGuard guard(device);
        {
            py::gil_scoped_release nogil;     // (blocks until every rank has arrived)
            rccl_check(rccl().CommInitRank(&comm, world, id, rank), "ncclCommInitRank");
        }
        use_v = allow_alltoallv && rccl().AllToAllv != nullptr;
    }
    ~DirectA2A() { close(); }
    void close() {
        if (comm) { (void)rccl().CommDestroy(comm); comm = nullptr; }
    }
    // recv / send: flat fp32 device tensors; out_splits[p] = elements received FROM rank p, in_splits[p] = elements sent TO rank p
    // (all_to_all_single's argument order, extend_distributed.py:394-405).  -> the plan's number
    int64_t plan(const at::Tensor &recv, const at::Tensor &send, const std::vector<int64_t> &out_splits, const std::vector<int64_t> &in_splits) {
        TORCH_CHECK((int)out_splits.size() == world && (int)in_splits.size() == world, "one split per rank");
        TORCH_CHECK(recv.is_cuda() && send.is_cuda() && recv.scalar_type() == at::kFloat && send.scalar_type() == at::kFloat &&
                    recv.is_contiguous() && send.is_contiguous(), "recv / send must be contiguous fp32 device tensors");
        Plan p;
        p.send = send.data_ptr<float>(); p.recv = recv.data_ptr<float>();
        size_t so = 0, ro = 0;
        for (int r = 0; r < world; r++) {
            TORCH_CHECK(out_splits[r] >= 0 && in_splits[r] >= 0, "negative split");
            p.scnt.push_back((size_t)in_splits[r]); p.sdis.push_back(so); so += (size_t)in_splits[r];
            p.rcnt.push_back((size_t)out_splits[r]); p.rdis.push_back(ro); ro += (size_t)out_splits[r];
        }
        TORCH_CHECK((int64_t)so <= send.numel() && (int64_t)ro <= recv.numel(), "the splits exceed the buffers");
        TORCH_CHECK(p.send != p.recv, "the exchange needs a receive buffer of its own");
        plans.push_back(std::move(p));
        return (int64_t)plans.size() - 1;
    }
    void run(int64_t k) {
        TORCH_CHECK(comm && k >= 0 && k < (int64_t)plans.size(), "no such plan");
        const Plan &p = plans[(size_t)k];
        hipStream_t st = c10::hip::getCurrentHIPStream(device).stream();
        if (use_v) {
            rccl_check(rccl().AllToAllv(p.send, p.scnt.data(), p.sdis.data(), p.recv, p.rcnt.data(), p.rdis.data(), ncclFloat, comm, st), "ncclAllToAllv");
            return;
        }
        rccl_check(rccl().GroupStart(), "ncclGroupStart");
        for (int r = 0; r < world; r++) {
            if (p.scnt[r]) rccl_check(rccl().Send(p.send + p.sdis[r], p.scnt[r], ncclFloat, r, comm, st), "ncclSend");
            if (p.rcnt[r]) rccl_check(rccl().Recv(p.recv + p.rdis[r], p.rcnt[r], ncclFloat, r, comm, st), "ncclRecv");
        }
        rccl_check(rccl().GroupEnd(), "ncclGroupEnd");
    }
};

void set_error_class(py::object cls) {
    Py_XDECREF(g_error_class);
    g_error_class = cls.release().ptr();
}

}  // namespace

PYBIND11_MODULE(_evs_torch_ext, m) {
    m.doc() = "PyTorch-ROCm C++ extension over libevstore_hip.so (call path of the DLRM plugin surface)";
    py::class_<Tables>(m, "Tables")
        .def(py::init<std::vector<at::Tensor>, std::vector<int64_t>, int, int, int>())
        .def_readonly("T", &Tables::T)
        .def_readonly("d", &Tables::d)
        .def_readonly("codec", &Tables::codec);
    m.def("set_error_class", &set_error_class);
    m.def("abi_version", []() { return evs_abi_version(); });
    m.def("apply_emb_interact", &apply_emb_interact, py::arg("ev"), py::arg("x"), py::arg("lS_o"), py::arg("lS_i"), py::arg("itself") = false,
          py::arg("out") = py::none(), py::arg("one_index_per_bag") = false, py::arg("check_indices") = false);
    m.def("apply_emb_interact_multi", &apply_emb_interact_multi, py::arg("ev"), py::arg("xs"), py::arg("lS_os"), py::arg("lS_is"),
          py::arg("itself") = false, py::arg("outs") = py::none(), py::arg("one_index_per_bag") = false);
    m.def("apply_emb", &apply_emb, py::arg("ev"), py::arg("lS_o"), py::arg("lS_i"), py::arg("one_index_per_bag") = false,
          py::arg("check_indices") = false);
    m.def("apply_emb_list", &apply_emb_list, py::arg("ev"), py::arg("lS_o"), py::arg("lS_i"), py::arg("one_index_per_bag") = false,
          py::arg("check_indices") = false);
    m.def("interact_dot", &interact_dot);
    m.def("interact_dot_pooled", &interact_dot_pooled);
    m.def("cache_request", &cache_request);
    m.def("cache_lookup_interact", &cache_lookup_interact);
    m.def("hostcache_request_list", &hostcache_request_list);
    m.def("serve_request_list", &serve_request_list);
    m.def("slices", &slices);
    m.def("serve_post", &serve_post);
    m.def("serve_wait", &serve_wait);
    m.def("serve_run", &serve_run);
    m.def("rccl_available", &rccl_available);
    m.def("rccl_unique_id", &rccl_unique_id);
    py::class_<DirectA2A>(m, "DirectA2A")
        .def(py::init<const std::string &, int, int, int, bool>())
        .def("plan", &DirectA2A::plan)
        .def("run", &DirectA2A::run)
        .def("close", &DirectA2A::close)
        .def_readonly("use_alltoallv", &DirectA2A::use_v);
}
