// hs_capture.h - record the device work of a run of library calls, replay it with ONE call.
//
// A query that the engine has already run once with every buffer in place repeats exactly the same launches with
// exactly the same arguments (minispark_amd/device.py Recording).  Instead of walking the library's entry points
// again from Python - argument marshalling, validation, JIT cache keys - the launches themselves are captured here:
// between hs_capture_begin() and hs_capture_end() every kernel launch, event record and memset of THIS thread is
// executed as usual and also appended to a list (function, geometry, a private copy of the arguments);
// hs_capture_replay() issues the list on a stream.  The host-side counterpart of a hipGraph, built from the launches
// the library really makes, with no stream-capture restrictions on the code in between.
//
// Every launch in the library goes through hs_launch (the hipLaunchKernelGGL macro is pointed at it below), the few
// module launches / event records / memsets through their hs_* wrappers.
#pragma once
#ifndef HS_JIT_BUILD

#include <hip/hip_runtime.h>

#include <memory>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

struct HsCapOp {
    enum Kind { KERNEL, MODULE, EVENT, MEMSET } kind = KERNEL;
    const void* fn = nullptr;     // KERNEL: host stub address
    hipFunction_t mfn = nullptr;  // MODULE
    dim3 grid, block;
    size_t lds = 0;
    std::shared_ptr<void> hold;  // KERNEL: the argument tuple
    std::vector<void*> argv;     // KERNEL: addresses of its elements
    std::vector<char> blob;      // MODULE: the argument buffer
    hipEvent_t ev = nullptr;     // EVENT
    void* ptr = nullptr;         // MEMSET
    int value = 0;
    size_t bytes = 0;
};
struct HsCapture {
    std::vector<HsCapOp> ops;
};
inline thread_local HsCapture* g_hs_capture = nullptr;

// ---- per-launch GPU slices for the query trace (hs_trace_begin / hs_trace_end; reference utils.py:85-135) ----------
struct HsTraceLaunch {
    const void* fn = nullptr;     // host stub (name via hipKernelNameRefByPtr) ...
    hipFunction_t mfn = nullptr;  // ... or module function (hipKernelNameRef)
    hipEvent_t begin = nullptr, end = nullptr;
};
struct HsTrace {
    hipEvent_t base = nullptr;
    std::vector<HsTraceLaunch> launches;
};
inline thread_local HsTrace* g_hs_trace = nullptr;

inline HsTraceLaunch* hs_trace_open(const void* fn, hipFunction_t mfn, hipStream_t stream) {
    if (!g_hs_trace) return nullptr;
    HsTraceLaunch t;
    t.fn = fn;
    t.mfn = mfn;
    if (hipEventCreate(&t.begin) != hipSuccess || hipEventCreate(&t.end) != hipSuccess) return nullptr;
    (void)hipEventRecord(t.begin, stream);
    g_hs_trace->launches.push_back(t);
    return &g_hs_trace->launches.back();
}
inline void hs_trace_close(HsTraceLaunch* t, hipStream_t stream) {
    if (t) (void)hipEventRecord(t->end, stream);
}

template <typename Tuple, size_t... I>
inline void hs_tuple_addresses(Tuple& t, std::vector<void*>& out, std::index_sequence<I...>) {
    (out.push_back((void*)&std::get<I>(t)), ...);
}

template <typename... K, typename... A>
inline void hs_launch(void (*kernel)(K...), dim3 grid, dim3 block, size_t lds, hipStream_t stream, A&&... a) {
    using Tuple = std::tuple<std::decay_t<K>...>;
    if (g_hs_capture) {
        auto held = std::make_shared<Tuple>(std::forward<A>(a)...);
        HsCapOp op;
        op.kind = HsCapOp::KERNEL;
        op.fn = (const void*)kernel;
        op.grid = grid;
        op.block = block;
        op.lds = lds;
        hs_tuple_addresses(*held, op.argv, std::index_sequence_for<K...>{});
        op.hold = held;
        HsTraceLaunch* tr = hs_trace_open(op.fn, nullptr, stream);
        (void)hipLaunchKernel(op.fn, grid, block, op.argv.data(), lds, stream);
        hs_trace_close(tr, stream);
        g_hs_capture->ops.push_back(std::move(op));
        return;
    }
    Tuple args(std::forward<A>(a)...);
    void* argv[sizeof...(K) > 0 ? sizeof...(K) : 1];
    {
        std::vector<void*> tmp;
        hs_tuple_addresses(args, tmp, std::index_sequence_for<K...>{});
        for (size_t i = 0; i < tmp.size(); ++i) argv[i] = tmp[i];
    }
    HsTraceLaunch* tr = hs_trace_open((const void*)kernel, nullptr, stream);
    (void)hipLaunchKernel((const void*)kernel, grid, block, argv, lds, stream);
    hs_trace_close(tr, stream);
}

inline hipError_t hs_module_launch(hipFunction_t fn, unsigned grid, unsigned block, size_t lds, hipStream_t stream,
                                   const void* args, size_t size) {
    std::vector<char> blob((const char*)args, (const char*)args + size);
    size_t sz = size;
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, blob.data(), HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    HsTraceLaunch* tr = hs_trace_open(nullptr, fn, stream);
    const hipError_t rc = hipModuleLaunchKernel(fn, grid, 1, 1, block, 1, 1, (unsigned)lds, stream, nullptr, extra);
    hs_trace_close(tr, stream);
    if (rc == hipSuccess && g_hs_capture) {
        HsCapOp op;
        op.kind = HsCapOp::MODULE;
        op.mfn = fn;
        op.grid = dim3(grid);
        op.block = dim3(block);
        op.lds = lds;
        op.blob = std::move(blob);
        g_hs_capture->ops.push_back(std::move(op));
    }
    return rc;
}

inline void hs_event_record(hipEvent_t ev, hipStream_t stream) {
    (void)hipEventRecord(ev, stream);
    if (g_hs_capture) {
        HsCapOp op;
        op.kind = HsCapOp::EVENT;
        op.ev = ev;
        g_hs_capture->ops.push_back(std::move(op));
    }
}

inline void hs_memset_async(void* ptr, int value, size_t bytes, hipStream_t stream) {
    (void)hipMemsetAsync(ptr, value, bytes, stream);
    if (g_hs_capture) {
        HsCapOp op;
        op.kind = HsCapOp::MEMSET;
        op.ptr = ptr;
        op.value = value;
        op.bytes = bytes;
        g_hs_capture->ops.push_back(std::move(op));
    }
}

// every hipLaunchKernelGGL of the library is a capturable launch
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
    hs_launch((kernel), dim3(grid), dim3(block), (size_t)(lds), (hipStream_t)(stream), __VA_ARGS__)

#endif  // HS_JIT_BUILD
