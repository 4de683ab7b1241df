// Host-side glue between torch's C++ autograd engine and the C ABI of libepipolar_hip (include/epipolar_hip.h).
//
// Not a kernel and not a second implementation: every function here only allocates outputs, forwards raw device
// pointers to an `epi_*` entry point on the current HIP stream, and tells autograd what to save.  It exists because the
// network calls a fused op ~60 times per step in each direction, and a Python autograd.Function + ctypes call costs
// ~35 us of host time per call -- at batch 32 the step had become host-bound (tools/host_profile.py).  The Python classes
// in models/fused.py keep the module interface (parameters, buffers, state_dict) and call into this extension.
//
//   bn_act        BatchNorm (+ residual) (+ ReLU)                                   one autograd node
//   conv_bn_act   Conv2d -> BatchNorm (+ residual) (+ ReLU)  (a whole conv-bn-relu    one autograd node, 2-3 launches forward,
//                 stage of BasicBlock / Bottleneck, pose3d_resnet.py:31-47,68-88)   4 backward
//   adam_prepare  the per-step pointer table of FusedAdam (170 parameters) without a Python loop
#include <torch/extension.h>
#include <torch/custom_class.h>
#include <torch/csrc/autograd/engine.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/epipolar_hip.h"

namespace {

using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

inline epi_stream_t current_stream(const Tensor& t) {
    return reinterpret_cast<epi_stream_t>(c10::hip::getCurrentHIPStream(t.device().index()).stream());
}

inline void check(int status, const char* what) {
    TORCH_CHECK(status == EPI_OK, what, " failed: ", epi_status_string(status));
}

inline bool nhwc_bf16(const Tensor& t) {
    return t.scalar_type() == at::kBFloat16 && t.dim() == 4 && t.is_contiguous(at::MemoryFormat::ChannelsLast);
}

// ---- optional per-entry-point timing with HIP events on the launch stream (bench.py's live roofline figures; off by default) ----
struct TimingRec { const char* name; double flops, bytes; hipEvent_t a, b; };
bool g_timing = false;
std::vector<TimingRec> g_recs;
std::vector<hipEvent_t> g_event_pool;         // events are created once and reused: hipEventCreate per launch made the step host-bound
hipEvent_t pooled_event() {
    if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    return hipEventCreate(&e) == hipSuccess ? e : nullptr;
}
struct ScopedTimer {
    TimingRec r;
    hipStream_t st;
    bool on;
    ScopedTimer(const char* name, double flops, double bytes, epi_stream_t stream) : st((hipStream_t)stream), on(g_timing) {
        if (!on) return;
        r.name = name; r.flops = flops; r.bytes = bytes;
        r.a = pooled_event();
        r.b = pooled_event();
        on = r.a && r.b && hipEventRecord(r.a, st) == hipSuccess;
    }
    ~ScopedTimer() {
        if (on && hipEventRecord(r.b, st) == hipSuccess) g_recs.push_back(r);
    }
};
void timing_enable(bool on) {
    if (on && g_event_pool.size() < 16384) {       // warm the pool outside the timed region
        g_event_pool.reserve(16384);
        while (g_event_pool.size() < 16384) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) break;
            g_event_pool.push_back(e);
        }
    }
    g_timing = on;
}
// {name: (launches, total ms, total algorithmic FLOPs, total algorithmic bytes)}; synchronises; clears the records
std::map<std::string, std::tuple<int64_t, double, double, double>> timing_collect() {
    std::map<std::string, std::tuple<int64_t, double, double, double>> out;
    for (auto& r : g_recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            auto& e = out[r.name];
            std::get<0>(e) += 1; std::get<1>(e) += ms; std::get<2>(e) += r.flops; std::get<3>(e) += r.bytes;
        }
        g_event_pool.push_back(r.a);
        g_event_pool.push_back(r.b);
    }
    g_recs.clear();
    return out;
}

// split-K / slab scratch shared by every GEMM-type launch of this process (one process per GPU; launches on one stream are
// ordered, so consecutive kernels may reuse it); grown on demand, never shrunk
Tensor& workspace(size_t bytes, const Tensor& like, int slot = 0) {      // slot 1: launches on the weight-gradient stream
    static std::vector<Tensor> per_device(128);
    Tensor& ws = per_device[2 * like.device().index() + slot];
    if (!ws.defined() || (size_t)ws.numel() < bytes)
        ws = at::empty({(int64_t)std::max<size_t>(bytes, (size_t)1 << 16)}, like.options().dtype(at::kByte).memory_format(at::MemoryFormat::Contiguous));
    return ws;
}

// ---- weight gradients on a second stream ---------------------------------------------------------------------------
// Nothing inside the backward pass waits for a weight gradient: the chain is BatchNorm backward -> backward-data -> the next layer's
// BatchNorm backward, and the weight-gradient GEMMs (1.7 of the 8 ms step on ResNet-50 / batch 32, each one under-filling the chip)
// only have to be complete when the optimizer runs.  With EPI_WGRAD_STREAM != 0 they are launched on a second HIP stream: fork = ONE
// event per autograd node, recorded on the main stream when the node's own work has been enqueued (side_run_jobs), join = the main
// stream waits once, at the end of the pass (the engine's final callback, together with the deferred slab sums) or earlier when
// somebody consumes a gradient inside the pass.  State is per process (one process drives one GPU; nodes of a pass run one at a time).
// The operands (x, dy) are kept alive until the join instead of being registered with the caching allocator's per-stream use list
// (recordStream costs an event per tensor when it is freed; holding ~2x the activations for one backward pass costs nothing at 288 GB).
struct SideStream {
    int mode = -1;                   // 0 off, 1 same priority as the main stream, 2 lowest priority
    int device = -1;
    hipStream_t stream = nullptr;
    std::vector<hipEvent_t> fork;    // ring of events recorded on the main stream
    size_t next = 0;
    hipEvent_t joined = nullptr;
    hipStream_t main_stream = nullptr;
    bool dirty = false, low_priority = false;
    std::vector<Tensor> keep;
    // weight-gradient launches of the residual unit whose backward is being enqueued: they go out together behind ONE fork event when
    // the unit is done (an event recorded on the main stream costs it a ~6 us bubble -- measured launch by launch with
    // tools/trace_step_sequence.py --, 52 of them per step ate half of what the second stream gained)
    struct Job {
        Tensor x, dy, dw;                // operands (kept alive until the join) and the gradient (released before autograd sees it)
        // launches the weight-gradient kernel(s) on the given stream; returns the pending slab sum (nsplit == 0: `dw` is complete)
        std::function<EpiSlabReduce(epi_stream_t)> launch;
        const char* name;
        double flops, bytes;
    };
    std::vector<Job> jobs;
    // Grouped weight gradients (round 3): the backward-weight GEMMs of SEVERAL autograd nodes -- every residual unit of a ResNet stage --
    // wait here as plain descriptors and leave in ONE epi_wgrad_group launch when the stage's first unit (the last one the backward pass
    // reaches) is done: together their output tiles fill the chip with unsplit reductions (866 MB of fp32 slabs per ResNet-50 step
    // before, < 100 MB now) and the main stream records one fork event per stage instead of one per unit.  The gradient itself is held
    // WEAKLY (through its storage), exactly as PendingReduces does: AccumulateGrad must find itself the only owner to adopt the tensor.
    struct GroupItem {
        EpiWgradItem it;
        Tensor x, dy;                                            // operands, alive until the launch (second stream: until the join)
        c10::weak_intrusive_ptr<c10::StorageImpl> dw;
        c10::Device dev;
        double flops, bytes;
        Tensor x_stats;                                          // EpiWgradItem::x_scale_shift points into it: alive as long as x
    };
    std::vector<GroupItem> group;
};
SideStream g_side;
// EPI_WGRAD_GROUP: 0 one launch (+ its own reduction split) per layer as in round 2; 1 one grouped launch per autograd node (residual
// unit); 2 (default) one per ResNet stage -- flushed behind the unit that carries the stage's downsample projection
int g_group_mode = -1;
int group_mode() {
    if (g_group_mode < 0) g_group_mode = 2;
    return g_group_mode;
}
int wgrad_group_mode(int mode) {               // test / measurement hook: returns the previous setting; a negative mode only queries
    const int before = group_mode();
    if (mode >= 0 && mode <= 2) g_group_mode = mode;
    return before;
}
void side_group_flush();
// scratch of the launches on the weight-gradient stream.  Grown on demand like workspace(); the outgrown buffer may still be in use by
// a launch in flight on that stream while the caching allocator would hand it to the main stream at once -- it is parked until the join.
Tensor& side_workspace(size_t bytes, const Tensor& like) {
    static std::vector<Tensor> per_device(64);
    Tensor& ws = per_device[like.device().index()];
    if (!ws.defined() || (size_t)ws.numel() < bytes) {
        if (ws.defined()) g_side.keep.push_back(ws);
        ws = at::empty({(int64_t)std::max<size_t>(bytes, (size_t)1 << 16)}, like.options().dtype(at::kByte).memory_format(at::MemoryFormat::Contiguous));
    }
    return ws;
}

int side_mode() {
    if (g_side.mode < 0) g_side.mode = 1;
    return g_side.mode;
}
int wgrad_stream_mode(int mode) {             // test / measurement hook: returns the previous setting; a negative mode only queries
    const int before = side_mode();
    if (mode >= 0) g_side.mode = mode;
    return before;
}

// the main stream waits for everything launched on the weight-gradient stream so far
void side_join() {
    SideStream& S = g_side;
    if (!S.dirty) return;
    TORCH_CHECK(hipEventRecord(S.joined, S.stream) == hipSuccess, "weight-gradient stream: join event");
    hipStream_t main_stream = c10::hip::getCurrentHIPStream(S.device).stream();
    TORCH_CHECK(hipStreamWaitEvent(main_stream, S.joined, 0) == hipSuccess, "weight-gradient stream: join");
    if (S.main_stream != main_stream)         // forked from another stream than the one that joins: that one has to wait as well
        TORCH_CHECK(hipStreamWaitEvent(S.main_stream, S.joined, 0) == hipSuccess, "weight-gradient stream: join");
    S.keep.clear();
    S.dirty = false;
}

// returns the weight-gradient stream, ordered behind everything enqueued on `main_stream` so far
hipStream_t side_fork(int dev, hipStream_t main_stream) {
    SideStream& S = g_side;
    const bool low = S.mode == 2;
    if (S.stream == nullptr || S.device != dev || S.low_priority != low) {
        c10::hip::HIPGuard device_guard((c10::DeviceIndex)dev);          // the stream and its events belong to `dev`, whatever the caller's current device
        if (S.stream != nullptr && S.device == dev) side_join();           // priority class changed: drain, then replace the stream
        TORCH_CHECK(!S.dirty, "weight-gradient stream: pending work on another device");
        if (S.stream != nullptr) (void)hipStreamDestroy(S.stream);        // (destruction waits for the stream's work)
        int least = 0, greatest = 0;
        TORCH_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess, "weight-gradient stream: priority range");
        // (a CU-masked stream -- hipExtStreamCreateWithCUMask, 25 .. 75 % of the units -- was measured at 12.8 .. 15.9 ms/step against 7.56:
        // rejected)
        TORCH_CHECK(hipStreamCreateWithPriority(&S.stream, hipStreamNonBlocking, low ? least : 0) == hipSuccess,
                    "weight-gradient stream: create");
        if (S.fork.empty()) {
            // These events carry DATA dependencies (x / dy produced on the main stream, dw / slabs produced on the second one), so they keep
            // the fence an event adds by default.  EPI_EVENT_FENCE=0 drops it (hipEventDisableSystemFence: both streams are on one device
            // and every kernel ends / starts with its own agent-scope release / acquire; measured 7.26 -> 7.13 ms/step with 20 fork events
            // per step in round 2 -- with one fork per ResNet stage the difference is gone, and HIP documents the flag for timing-only events)
            const bool fence = true;
            const unsigned flags = hipEventDisableTiming | (fence ? 0u : (unsigned)hipEventDisableSystemFence);
            S.fork.resize(256);
            for (auto& e : S.fork) TORCH_CHECK(hipEventCreateWithFlags(&e, flags) == hipSuccess, "weight-gradient stream: event");
            TORCH_CHECK(hipEventCreateWithFlags(&S.joined, flags) == hipSuccess, "weight-gradient stream: event");
        }
        S.device = dev;
        S.low_priority = low;
    }
    hipEvent_t e = S.fork[S.next++ % S.fork.size()];
    TORCH_CHECK(hipEventRecord(e, main_stream) == hipSuccess, "weight-gradient stream: fork event");
    TORCH_CHECK(hipStreamWaitEvent(S.stream, e, 0) == hipSuccess, "weight-gradient stream: fork");
    S.main_stream = main_stream;
    S.dirty = true;
    return S.stream;
}

// ---- deferred weight-gradient reductions ---------------------------------------------------------------------------
// A split weight gradient leaves fp32 slabs that one more launch per layer would sum (55 launches per ResNet-50 backward, 4 .. 10 us
// each and mostly latency).  Instead every backward-weight launch of a backward pass parks its slabs in ONE arena and registers the
// outstanding sum; a final callback of the autograd engine (it runs on the caller's stream once the whole graph task is done,
// before backward() returns) sums them all with a single epi_slab_reduce_multi launch.  Anything that consumes a gradient earlier --
// the bucketed all-reduce hooks -- calls flush_pending_reduces() first.  EPI_DEFER_WGRAD_REDUCE=0 restores one reduce per layer.
struct PendingReduces {
    Tensor arena;                    // bytes; slabs of the current backward pass, bump-allocated
    size_t used = 0, wanted = 0;     // wanted: what this pass would have needed (the arena grows after the flush)
    size_t target = (size_t)64 << 20;   // capacity of the next allocation: the largest `wanted` seen so far
    std::vector<EpiSlabReduce> rows;
    // the gradients the rows point to, held WEAKLY through their storage: a strong reference would keep AccumulateGrad from adopting
    // the tensor as .grad (it would clone the not-yet-reduced memory instead); a storage that died before the flush is skipped
    std::vector<c10::weak_intrusive_ptr<c10::StorageImpl>> keep;
    c10::Device dev = c10::Device(c10::kCPU);
    int device = -1;
};
PendingReduces g_pend;
int g_defer = -1;
bool defer_enabled() {
    if (g_defer < 0) g_defer = 1;
    return g_defer != 0;
}
// test / measurement hook: returns the previous setting
bool defer_wgrad_reduce(bool on) {
    const bool before = defer_enabled();
    g_defer = on ? 1 : 0;
    return before;
}

void end_of_pass_callback();

// one cached device table per flush position inside a pass (0: an early flush from begin_early_step(), 1: the end of the pass): the
// tables repeat from step to step, so each slot is uploaded only when its content changed
struct ReduceTable {
    Tensor dev, host;                // device copy of the rows / pinned staging
    std::vector<EpiSlabReduce> uploaded;
    hipEvent_t upload_done = nullptr;
};
constexpr int REDUCE_TABLE_SLOTS = 10;   // 0: begin_early_step, 1: the end of the pass, 2 ..: the asynchronous flushes of a pass, in order
ReduceTable g_reduce_tables[REDUCE_TABLE_SLOTS];

// sum the registered slabs with one launch on `stream`; leaves the arena bookkeeping alone
void launch_pending_rows(hipStream_t stream, int slot) {
    PendingReduces& P = g_pend;
    if (P.rows.empty()) return;
    ReduceTable& T = g_reduce_tables[slot];
    const int nrows = (int)P.rows.size();
    long long chunks = 0;
    std::vector<c10::intrusive_ptr<c10::StorageImpl>> alive(P.keep.size());
    for (size_t i = 0; i < P.rows.size(); ++i) {
        alive[i] = P.keep[i].lock();
        if (!alive[i]) { P.rows[i].n = 0; P.rows[i].nsplit = 0; }                    // nobody holds this gradient any more
    }
    for (auto& r : P.rows) { r.chunk_begin = chunks; chunks += epi_slab_reduce_chunks(r.n); }
    const size_t bytes = sizeof(EpiSlabReduce) * (size_t)nrows;
    const auto byte_opts = at::TensorOptions().dtype(at::kByte).device(P.dev);
    // the table is the same from step to step (same layers, same arena offsets, gradients from the caching allocator usually at
    // the same addresses): upload only when it changed
    const bool same = T.uploaded.size() == P.rows.size() && std::memcmp(T.uploaded.data(), P.rows.data(), bytes) == 0;
    if (!same) {
        if (!T.dev.defined() || (size_t)T.dev.numel() < bytes || T.dev.device() != P.dev) {
            T.dev = at::empty({(int64_t)std::max<size_t>(bytes, 8192)}, byte_opts);
            T.host = at::empty({T.dev.numel()}, at::TensorOptions().dtype(at::kByte).pinned_memory(true));
        }
        if (T.upload_done) {                             // the previous copy must have read the staging buffer
            TORCH_CHECK(hipEventSynchronize(T.upload_done) == hipSuccess, "deferred reduce: event");
        } else {
            TORCH_CHECK(hipEventCreateWithFlags(&T.upload_done, hipEventDisableTiming) == hipSuccess, "deferred reduce: event");
        }
        std::memcpy(T.host.data_ptr(), P.rows.data(), bytes);
        TORCH_CHECK(hipMemcpyAsync(T.dev.data_ptr(), T.host.data_ptr(), bytes, hipMemcpyHostToDevice, stream) == hipSuccess,
                    "deferred reduce: table upload");
        TORCH_CHECK(hipEventRecord(T.upload_done, stream) == hipSuccess, "deferred reduce: event");
        T.uploaded = P.rows;
    }
    if (chunks > 0) {
        ScopedTimer timer("conv_bwd_weight", 0.0, 0.0, reinterpret_cast<epi_stream_t>(stream));    // (the reduce belongs to the family's time)
        check(epi_slab_reduce_multi(reinterpret_cast<const EpiSlabReduce*>(T.dev.data_ptr()), nrows, chunks,
                                    reinterpret_cast<epi_stream_t>(stream)), "epi_slab_reduce_multi");
    }
    P.rows.clear();
    P.keep.clear();
}

std::vector<const void*> g_seen_weights;     // weights whose gradient this backward pass has already produced once (shared weights)
struct FlushTickets;
void reset_flush_ordinal();

void flush_pending_reduces() {
    side_group_flush();              // weight gradients still waiting for their grouped launch
    PendingReduces& P = g_pend;
    if (!P.rows.empty()) {
        // Which stream?  Optionally the second one, behind the weight gradients in flight there (and behind the main stream's own split
        // launches: one more fork event), overlapping what the main stream still has to do -- at the end of a ResNet backward the stem's
        // max-pool / BatchNorm / convolution backward.
        hipStream_t stream = c10::hip::getCurrentHIPStream(P.dev.index()).stream();
        // (EPI_REDUCE_STREAM=1; measured on MI355X: 7.48 vs 7.46 ms/step on the main stream -- the sum then competes with the stem's
        // backward instead of following it -- so it stays on the main stream by default)
        const bool reduce_on_side = false;
        if (reduce_on_side && g_side.dirty && g_side.device == P.dev.index()) stream = side_fork(P.dev.index(), stream);
        else side_join();            // the slabs may still be in flight on the weight-gradient stream
        launch_pending_rows(stream, 1);
    }
    side_join();                     // the main stream waits for everything on the weight-gradient stream: unsplit gradients, slabs, the sum
    reset_flush_ordinal();
    P.target = std::max(P.target, P.wanted);
    if (P.arena.defined() && P.target > (size_t)P.arena.numel()) P.arena = Tensor();            // regrown by the next pass (stream-ordered free)
    P.used = 0;
    P.wanted = 0;
}

// ---- asynchronous flush (the bucketed gradient path, distributed.BucketedGradSync) --------------------------------------------------
// A bucket hook used to call flush_pending_reduces(): the grouped weight gradients went out and the MAIN stream joined the second one --
// four stalls of ~100 us per ResNet-50 backward pass, 5.4 % of the step (profiles/r03_bucket_path_overhead_a_*.txt).  Here nothing
// waits: the pending group and the slab sums registered so far are enqueued on the second stream and an event marks their end; the
// caller lets the main stream wait for that event when it LAUNCHES the bucket -- one bucket of backward later, when the event has
// normally fired long ago.  Returns a ticket for wait_flush_ticket(), or -1 when the flush was synchronous (second stream off).
struct FlushTickets {
    std::vector<hipEvent_t> events;
    size_t next = 0;
    int ordinal = 0;                 // asynchronous flushes of the current pass so far (selects the cached reduce table)
};
FlushTickets g_tickets;
void reset_flush_ordinal() { g_tickets.ordinal = 0; }

int64_t flush_pending_async() {
    if (side_mode() == 0) { flush_pending_reduces(); return -1; }
    side_group_flush();
    PendingReduces& P = g_pend;
    SideStream& S = g_side;
    if (!S.dirty && P.rows.empty()) return -1;                 // nothing in flight anywhere: the gradients are final already
    const int dev = P.rows.empty() ? S.device : (int)P.dev.index();
    hipStream_t main_stream = c10::hip::getCurrentHIPStream((c10::DeviceIndex)dev).stream();
    hipStream_t side = side_fork(dev, main_stream);             // behind everything enqueued so far on either stream
    if (!P.rows.empty()) launch_pending_rows(side, std::min(2 + g_tickets.ordinal, REDUCE_TABLE_SLOTS - 1));
    g_tickets.ordinal += 1;
    if (g_tickets.events.empty()) {
        g_tickets.events.resize(32);
        for (auto& e : g_tickets.events) TORCH_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess, "flush ticket: event");
    }
    const size_t idx = g_tickets.next++ % g_tickets.events.size();
    TORCH_CHECK(hipEventRecord(g_tickets.events[idx], side) == hipSuccess, "flush ticket: record");
    end_of_pass_callback();                                     // the join (and the arena bookkeeping) still happen at the end of the pass
    return (int64_t)idx;
}

void wait_flush_ticket(int64_t ticket, int64_t device_index) {
    if (ticket < 0) return;
    TORCH_CHECK((size_t)ticket < g_tickets.events.size(), "flush ticket: unknown ticket");
    hipStream_t main_stream = c10::hip::getCurrentHIPStream((c10::DeviceIndex)device_index).stream();
    TORCH_CHECK(hipStreamWaitEvent(main_stream, g_tickets.events[(size_t)ticket], 0) == hipSuccess, "flush ticket: wait");
}

// Optimizer work INSIDE the backward pass (optim.FusedAdam.enable_step_in_backward): called from a tensor hook at a point of the
// pass where the gradients of everything downstream are final except for their pending slab sums.  Sums those slabs on the
// weight-gradient stream (behind the launches that write them) and returns that stream: what the caller enqueues on it next -- the
// Adam update and the weight re-packing of those parameters -- runs beside the rest of the backward pass instead of after it.  The
// main stream joins at the end of the pass as usual.  Returns 0 when the second stream is off (the caller then does nothing early).
int64_t begin_early_step(int64_t device_index) {
    if (side_mode() == 0) return 0;
    side_group_flush();
    PendingReduces& P = g_pend;
    hipStream_t main_stream = c10::hip::getCurrentHIPStream((c10::DeviceIndex)device_index).stream();
    hipStream_t side = side_fork((int)device_index, main_stream);          // behind everything enqueued so far on either stream
    if (!P.rows.empty() && P.dev.index() == device_index) launch_pending_rows(side, 0);
    end_of_pass_callback();                                                // somebody has to join, even if no weight gradient follows
    return reinterpret_cast<int64_t>(side);
}

// slab memory for one backward-weight launch of `bytes`, or nullptr when the arena is full (the caller then reduces at once)
void* pending_slab_alloc(size_t bytes, const Tensor& like) {
    PendingReduces& P = g_pend;
    bytes = (bytes + 255) & ~(size_t)255;
    if (P.device != like.device().index()) {            // one process drives one GPU; a device change starts over
        TORCH_CHECK(P.rows.empty(), "deferred reduce: pending work on another device");
        P.arena = Tensor(); P.device = like.device().index();
    }
    P.wanted += bytes;
    if (!P.arena.defined()) {
        if (P.used != 0) return nullptr;                 // dropped mid-pass: wait for the next pass
        const size_t cap = std::max(P.wanted, P.target);
        P.arena = at::empty({(int64_t)cap}, like.options().dtype(at::kByte).memory_format(at::MemoryFormat::Contiguous));
    }
    if (P.used + bytes > (size_t)P.arena.numel()) return nullptr;
    void* p = static_cast<char*>(P.arena.data_ptr()) + P.used;
    P.used += bytes;
    return p;
}

// May the gradient of weight `w` stay unreduced until the end of the backward pass?  Only when nothing looks at it earlier: `w` is a
// leaf (no upstream node reads the gradient), has no gradient yet (AccumulateGrad will adopt the tensor, not add to it), and carries
// no tensor hooks / post-accumulate hooks (those run inside the pass; the bucketed all-reduce, which hooks the LAST gradient of each
// bucket, flushes explicitly for the others).
// Parameters whose post-accumulate hooks were all REMOVED again: torch keeps the (now empty) hook object on the tensor for good, so
// post_acc_grad_hooks(w) stays non-null and the rule below would keep every such gradient on the main stream, unsplit-only and ungrouped.
// distributed.BucketedGradSync hooks every parameter during its first step (to learn which gradient completes each bucket) and
// declares the others hook-free afterwards.  Held weakly; an entry whose tensor died is dropped.
std::vector<c10::weak_intrusive_ptr<c10::TensorImpl>> g_hook_free;
void declare_hook_free(const std::vector<Tensor>& tensors) {
    g_hook_free.clear();
    for (const Tensor& t : tensors)
        if (t.defined()) g_hook_free.emplace_back(c10::weak_intrusive_ptr<c10::TensorImpl>(t.getIntrusivePtr()));
}
bool declared_hook_free(const Tensor& w) {
    const c10::TensorImpl* key = w.unsafeGetTensorImpl();
    for (const auto& e : g_hook_free)
        if (e._unsafe_get_target() == key && !e.expired()) return true;
    return false;
}

// ---- column sums of a gradient, delivered by the kernel that wrote it ----
// The criterion's backward kernel (epi_softargmax3d_bwd_colsums, core/integral_loss.py) writes the logits' gradient AND its per-channel sums -- the bias
// gradient of the final 1x1 convolution, which otherwise re-reads the 285 MB gradient on the second stream (epi_column_sums_bf16, 92 us beside the
// head's backward chain).  The producer OFFERS the sums for the tensor it returns; conv1x1_bias's backward TAKES them when the gradient it receives is
// that very memory, unmodified (same live storage, offset, element count and version counter: a view made by autograd's reshape shares all of them; a
// sum of two gradients or an in-place edit by a hook does not) -- otherwise it computes them as before.  One offer at a time, consumed once.
// MEASURED AND OFF BY DEFAULT (EPI_BIAS_GRAD_FUSE=1 / bias_grad_fuse_mode(1) turn it on): the kernel pays 2 .. 5 us for the sums and the 92 us pass
// with its 285 MB disappear from the second stream, yet the step gets 0.4 % SLOWER (6.172 / 6.151 / 6.158 -> 6.196 / 6.183 / 6.189 ms, same box,
// profiles/r04_ab_bias_grad_from_criterion.txt): the column-sum pass was a light neighbour of the head's backward chain, and without it the final
// layer's weight-gradient GEMM (247 us, reading the same 285 MB) starts 92 us earlier, beside the chain's two largest GEMMs instead of behind them.
struct ColumnSumsOffer {
    c10::weak_intrusive_ptr<c10::StorageImpl> storage{c10::intrusive_ptr<c10::StorageImpl>()};   // of the offered gradient: while it lives, nobody else owns that memory
    int64_t numel = 0, offset = 0, version = -1;
    Tensor sums;
};
ColumnSumsOffer g_colsum_offer;
long long g_bias_sums_taken = 0;         // test hook: how many bias gradients came from an offer
int g_colsum_mode = -1;
bool column_sums_wanted() {
    if (g_colsum_mode < 0) g_colsum_mode = 0;
    return g_colsum_mode != 0 && epi_set_deterministic(-1) == 0;
}
int bias_grad_fuse_mode(int mode) {              // test / measurement hook: returns the previous setting; a negative mode only queries
    (void)column_sums_wanted();
    const int before = g_colsum_mode;
    if (mode == 0 || mode == 1) g_colsum_mode = mode;
    return before;
}
void offer_column_sums(const Tensor& grad, const Tensor& sums) {
    TORCH_CHECK(grad.is_cuda() && sums.is_cuda() && sums.scalar_type() == at::kFloat && sums.is_contiguous(), "offer_column_sums: a float32 device tensor of sums");
    ColumnSumsOffer o;
    o.storage = grad.storage().getWeakStorageImpl();
    o.numel = grad.numel();
    o.offset = grad.storage_offset();
    o.version = (int64_t)grad._version();
    o.sums = sums;
    g_colsum_offer = std::move(o);
}
Tensor take_column_sums(const Tensor& grad, int64_t channels) {
    ColumnSumsOffer o = std::move(g_colsum_offer);      // consumed (or dropped) either way: an offer never outlives the next backward of a bias
    g_colsum_offer = ColumnSumsOffer();
    if (!o.sums.defined() || !grad.defined()) return Tensor();
    const auto alive = o.storage.lock();                // the offered tensor's storage, if anybody still holds it
    if (!alive || alive.get() != grad.storage().unsafeGetStorageImpl() || grad.numel() != o.numel || grad.storage_offset() != o.offset ||
        (int64_t)grad._version() != o.version || o.sums.numel() != channels || o.sums.device() != grad.device())
        return Tensor();
    return o.sums;
}

bool gradient_consumed_after_backward(const Tensor& w) {
    if (!w.defined() || !w.is_leaf() || w.grad().defined()) return false;
    if (torch::autograd::impl::post_acc_grad_hooks(w) != nullptr && !declared_hook_free(w)) return false;
    if (!torch::autograd::impl::hooks(w).empty()) return false;
    auto acc = torch::autograd::impl::try_get_grad_accumulator(w);
    // (post hooks of the accumulator: torch.nn.parallel.DistributedDataParallel registers its reducer there)
    if (acc && (!acc->tensor_pre_hooks().empty() || !acc->pre_hooks().empty() || !acc->retains_grad_hooks().empty() || !acc->post_hooks().empty()))
        return false;
    return true;
}

// A weight used twice in one graph: the engine ADDS the two gradients on the main stream when the second one arrives, so the first must
// be complete by then (w.grad() is still undefined at that point: gradient_consumed_after_backward cannot see it).  Returns true the
// first time `w` is seen in this backward pass; the list is cleared by the end-of-pass flush.
bool first_gradient_of_pass(const Tensor& w) {
    const void* key = w.unsafeGetTensorImpl();
    if (std::find(g_seen_weights.begin(), g_seen_weights.end(), key) != g_seen_weights.end()) return false;
    if (g_seen_weights.empty()) end_of_pass_callback();         // somebody has to clear the list when this pass ends
    g_seen_weights.push_back(key);
    return true;
}

// inside a backward pass: sum everything (and join the weight-gradient stream) when the pass ends -- one callback per registration,
// the first one to run does the work, the rest find nothing (a flag instead would be left stale by a pass that aborts); outside a
// pass: at once
void end_of_pass_callback() {
    bool queued = false;
    try {
        torch::autograd::Engine::get_default_engine().queue_callback([] { flush_pending_reduces(); g_seen_weights.clear(); });
        queued = true;
    } catch (const c10::Error&) {
    }
    if (!queued) { flush_pending_reduces(); g_seen_weights.clear(); }
}

void pending_register(const EpiSlabReduce& r, const Tensor& grad) {
    PendingReduces& P = g_pend;
    P.rows.push_back(r);
    P.keep.push_back(grad.storage().getWeakStorageImpl());
    P.dev = grad.device();
    end_of_pass_callback();
}

void pending_register_weak(const EpiSlabReduce& r, const c10::weak_intrusive_ptr<c10::StorageImpl>& grad, c10::Device dev) {
    PendingReduces& P = g_pend;
    P.rows.push_back(r);
    P.keep.push_back(grad);
    P.dev = dev;
}

// every waiting grouped weight gradient: one epi_wgrad_group call per <= epi_wgrad_group_max() items, on the second stream behind ONE
// fork event (or on the main stream when the second stream is off)
void side_group_flush() {
    SideStream& S = g_side;
    if (S.group.empty()) return;
    std::vector<SideStream::GroupItem> items;
    items.swap(S.group);
    const Tensor first = items.front().x;
    const int dev = first.device().index();
    const bool on_side = side_mode() != 0;
    const epi_stream_t main_stream = current_stream(first);
    const epi_stream_t st = on_side ? reinterpret_cast<epi_stream_t>(side_fork(dev, reinterpret_cast<hipStream_t>(main_stream))) : main_stream;
    const size_t cap = (size_t)epi_wgrad_group_max();
    for (size_t begin = 0; begin < items.size(); begin += cap) {
        const size_t end = std::min(items.size(), begin + cap);
        std::vector<EpiWgradItem> its;
        std::vector<size_t> idx;
        std::vector<c10::intrusive_ptr<c10::StorageImpl>> alive;         // the gradients stay alive while we enqueue
        double flops = 0, bytes = 0;
        for (size_t i = begin; i < end; ++i) {
            auto strong = items[i].dw.lock();
            if (!strong) continue;                                         // nobody holds this gradient any more: skip its GEMM
            alive.push_back(strong);
            its.push_back(items[i].it);
            idx.push_back(i);
            flops += items[i].flops; bytes += items[i].bytes;
        }
        if (its.empty()) continue;
        size_t slab_bytes = 0;
        check(epi_wgrad_group_plan(its.data(), (int)its.size(), &slab_bytes, nullptr), "epi_wgrad_group_plan");
        void* slabs = slab_bytes ? pending_slab_alloc(slab_bytes, first) : nullptr;
        if (slab_bytes && !slabs) {
            // the arena is full (first pass, or a pass larger than any before): one launch per layer with its own reduction, as in round 2
            for (size_t k = 0; k < its.size(); ++k) {
                const EpiWgradItem& it = its[k];
                ScopedTimer timer("conv_bwd_weight", items[idx[k]].flops, items[idx[k]].bytes, st);
                if (it.kind == EPI_WGRAD_DECONV4X4S2) {
                    Tensor& ws = on_side ? side_workspace(epi_gemm_tn_workspace_bytes(it.B * it.H * it.W, it.Cin, it.Cout, 16), first)
                                         : workspace(epi_gemm_tn_workspace_bytes(it.B * it.H * it.W, it.Cin, it.Cout, 16), first);
                    check(epi_deconv4x4s2_bwd_weight(it.x, it.dy, it.dw, it.dw_dtype, it.B, it.H, it.W, it.Cin, it.Cout, ws.data_ptr(), (size_t)ws.numel(), st),
                          "epi_deconv4x4s2_bwd_weight");
                } else {
                    const int Ho = (it.H + 2 * it.pad - it.KH) / it.stride + 1, Wo = (it.W + 2 * it.pad - it.KW) / it.stride + 1;
                    const size_t need = epi_gemm_tn_workspace_bytes(it.B * Ho * Wo, it.Cout, it.Cin, it.KH * it.KW);
                    Tensor& ws = on_side ? side_workspace(need, first) : workspace(need, first);
                    check(epi_wgrad_item(&it, ws.data_ptr(), (size_t)ws.numel(), nullptr, st), "epi_wgrad_item");
                }
            }
        } else {
            std::vector<EpiSlabReduce> pend(its.size());
            {
                ScopedTimer timer("conv_bwd_weight", flops, bytes, st);
                check(epi_wgrad_group(its.data(), (int)its.size(), slabs, slab_bytes, pend.data(), st), "epi_wgrad_group");
            }
            for (size_t k = 0; k < its.size(); ++k)
                if (pend[k].nsplit > 0) pending_register_weak(pend[k], items[idx[k]].dw, items[idx[k]].dev);
        }
    }
    if (on_side)
        for (auto& g : items) {
            S.keep.push_back(std::move(g.x));
            S.keep.push_back(std::move(g.dy));
            if (g.x_stats.defined()) S.keep.push_back(std::move(g.x_stats));
        }
    end_of_pass_callback();
}

// the queued weight-gradient launches of one autograd node, on the second stream behind one fork event
void side_run_jobs() {
    SideStream& S = g_side;
    if (S.jobs.empty()) return;
    const Tensor& first = S.jobs.front().x;
    const epi_stream_t st = reinterpret_cast<epi_stream_t>(side_fork(first.device().index(), reinterpret_cast<hipStream_t>(current_stream(first))));
    bool unsplit = false;
    for (SideStream::Job& j : S.jobs) {
        EpiSlabReduce pend = {};
        {
            ScopedTimer timer(j.name, j.flops, j.bytes, st);
            pend = j.launch(st);
        }
        S.keep.push_back(std::move(j.x));
        S.keep.push_back(std::move(j.dy));
        if (pend.nsplit > 0) pending_register(pend, j.dw);
        else unsplit = true;
    }
    S.jobs.clear();                 // (drops the strong references to the gradients before autograd sees them)
    if (unsplit) end_of_pass_callback();
}

// ---- BatchNorm halves shared by bn_act and conv_bn_act ------------------------------------------------------------
// flags (CPU int32[2], owned by the module): [0] sums_ws still holds a forward's sums, [1] bwd_sums still holds a
// backward's sums -- the accumulator hand-over protocol of epi_bn_act_fwd / epi_bn_act_bwd (see include/epipolar_hip.h)
struct BnBuffers {
    Tensor weight, bias, running_mean, running_var, num_batches, sums_ws, bwd_sums, flags;
};

// x: raw input [B,C,H,W] NHWC bf16.  Returns y; `stats` receives (mean | rstd | scale | shift).
// sums_ready: the producer already accumulated the batch sums into b.sums_ws (and ran the flag protocol): no statistics pass
Tensor bn_forward(const Tensor& x, const Tensor& residual, const BnBuffers& b, bool training, double momentum, double eps, bool relu,
                  Tensor* stats, bool sums_ready = false) {
    const bool has_res = residual.defined();
    if (has_res) TORCH_CHECK(nhwc_bf16(residual) && residual.sizes() == x.sizes(), "FusedBatchNormAct: residual layout");
    const int64_t B = x.size(0), C = x.size(1), H = x.size(2), W = x.size(3);
    TORCH_CHECK(b.weight.numel() == C, "FusedBatchNormAct: channel count mismatch");
    TORCH_CHECK(!training || b.sums_ws.numel() == 2 * C * epi_bn_sum_copies((int)C), "FusedBatchNormAct: sums_ws must be [epi_bn_sum_copies(C)][2C]");
    int* fl = b.flags.data_ptr<int>();
    Tensor sums_ws = b.sums_ws;
    if (training && !sums_ready) {
        if (fl[0]) sums_ws.zero_();
        fl[0] = 1;
        fl[1] = 0;
    }
    Tensor y = at::empty_like(x);
    *stats = at::empty({4 * C}, b.weight.options().dtype(at::kFloat));
    float* sp = stats->data_ptr<float>();
    const double tbytes = 2.0 * (double)x.numel();
    ScopedTimer timer(training ? (sums_ready ? "bn_fwd_apply" : "bn_fwd_stats+apply") : "bn_fwd_eval", 0.0,
                      tbytes * ((training && !sums_ready ? 1 : 0) + 2 + (has_res ? 1 : 0)), current_stream(x));
    check(epi_bn_act_fwd(x.data_ptr(), has_res ? residual.data_ptr() : nullptr, B * H * W, (int)C, b.weight.data_ptr<float>(),
                         b.bias.data_ptr<float>(), (float)eps, (float)momentum, training ? (sums_ready ? 2 : 1) : 0, relu ? 1 : 0,
                         b.running_mean.data_ptr<float>(), b.running_var.data_ptr<float>(),
                         reinterpret_cast<long long*>(b.num_batches.data_ptr<int64_t>()), sp, sp + C, sp + 2 * C, sums_ws.data_ptr<float>(),
                         training ? b.bwd_sums.data_ptr<float>() : nullptr, y.data_ptr(), current_stream(x)),
          "epi_bn_act_fwd");
    return y;
}

struct BnGrads { Tensor dx, dres, dgamma, dbeta; };

// dy: gradient of the fused output; x: the raw BatchNorm input saved by the forward; y: saved output (relu && residual) or undefined
// ---- The BatchNorm-backward reduction done by the backward-data GEMM that PRODUCES the layer's dy (EpiBnReduce, include/epipolar_hip.h) ----
// A node that ends in a BatchNorm publishes a link for its output tensor at forward time; the node that consumes that tensor claims it.
// In the backward pass the consumer runs first: if it is the only consumer, its backward-data epilogue masks the gradient, accumulates
// sum(dz) and sum(dz * xhat) into the producer's accumulator and records WHICH tensor it handed to autograd (TensorImpl + version).
// The producer skips its reduction pass only when exactly that tensor arrives unchanged -- a gradient that autograd summed with another
// consumer's (a new tensor, or an in-place add that bumps the version) takes the ordinary path on a fresh accumulator, which is still
// right because masking is idempotent: mask * (mask * g1 + g2) = mask * (g1 + g2).
struct BnLink : torch::CustomClassHolder {
    Tensor raw, ymask, stats, bwd_sums, flags;      // ymask: the saved output of a residual + ReLU layer (mask = y > 0), else undefined
    bool relu = false;
    c10::weak_intrusive_ptr<c10::TensorImpl> y{c10::intrusive_ptr<c10::TensorImpl>()};   // identity of the forward output
    int consumers = 0;
    bool reduced = false;
    c10::TensorImpl* dz_impl = nullptr;
    uint32_t dz_version = 0;
    void* key = nullptr;                            // its entry in g_links
};
// by the data pointer of the forward output; WEAK: a link (and the activations it references) lives exactly as long as the producing
// node's saved state -- a training-mode forward under no_grad leaves nothing behind
std::unordered_map<void*, c10::weak_intrusive_ptr<BnLink>> g_links;
int g_bn_fuse = -1;
bool bn_fuse_enabled() {
    if (g_bn_fuse < 0) g_bn_fuse = 1;
    return g_bn_fuse != 0;
}
int64_t g_bn_fused = 0, g_bn_refused = 0;         // BatchNorm backward passes that found their reduction done / done but unusable
std::vector<int64_t> bn_bwd_fuse_counts(bool reset) {
    std::vector<int64_t> out{g_bn_fused, g_bn_refused};
    if (reset) g_bn_fused = g_bn_refused = 0;
    return out;
}
int bn_bwd_fuse_mode(int mode) {                // test / measurement hook: returns the previous setting; a negative mode only queries
    const int prev = bn_fuse_enabled() ? 1 : 0;
    if (mode >= 0) g_bn_fuse = mode ? 1 : 0;
    return prev;
}
c10::intrusive_ptr<BnLink> link_publish(const Tensor& y, const Tensor& raw, const Tensor& ymask, const Tensor& stats, const Tensor& bwd_sums,
                                        const Tensor& flags, bool relu) {
    if (!bn_fuse_enabled()) return c10::intrusive_ptr<BnLink>();
    if (g_links.size() > 4096)                                     // (entries of forward passes whose backward never ran)
        for (auto it = g_links.begin(); it != g_links.end();) it = it->second.expired() ? g_links.erase(it) : std::next(it);
    auto link = c10::make_intrusive<BnLink>();
    link->raw = raw; link->ymask = ymask; link->stats = stats; link->bwd_sums = bwd_sums; link->flags = flags; link->relu = relu;
    link->y = c10::weak_intrusive_ptr<c10::TensorImpl>(y.getIntrusivePtr());
    link->key = y.data_ptr();
    g_links.insert_or_assign(link->key, c10::weak_intrusive_ptr<BnLink>(link));
    return link;
}
c10::intrusive_ptr<BnLink> link_claim(const Tensor& x) {
    if (!bn_fuse_enabled() || g_links.empty()) return c10::intrusive_ptr<BnLink>();
    auto it = g_links.find(x.data_ptr());
    if (it == g_links.end()) return c10::intrusive_ptr<BnLink>();
    auto link = it->second.lock();
    if (!link) { g_links.erase(it); return c10::intrusive_ptr<BnLink>(); }
    auto alive = link->y.lock();
    if (!alive || alive.get() != x.unsafeGetTensorImpl() || link->raw.sizes() != x.sizes()) return c10::intrusive_ptr<BnLink>();
    link->consumers += 1;
    return link;
}
void link_retire(const c10::intrusive_ptr<BnLink>& link) {          // the producer's backward has run (or the pass is over)
    if (!link) return;
    auto it = g_links.find(link->key);
    if (it != g_links.end() && it->second._unsafe_get_target() == link.get()) g_links.erase(it);
}
// the consumer's side: fills `red` when the reduction may be fused into its backward-data launch
bool link_reduce_args(const c10::intrusive_ptr<BnLink>& link, EpiBnReduce* red) {
    if (!link || !bn_fuse_enabled() || link->consumers != 1 || link->reduced || !link->raw.defined() || link->flags.data_ptr<int>()[1] != 0)
        return false;
    red->z = link->raw.data_ptr();
    red->y = (link->relu && link->ymask.defined()) ? link->ymask.data_ptr() : nullptr;
    red->bn = link->stats.data_ptr<float>();
    red->sums = link->bwd_sums.data_ptr<float>();
    red->relu = link->relu ? 1 : 0;
    return true;
}
void link_mark_reduced(const c10::intrusive_ptr<BnLink>& link, const Tensor& dz) {
    link->reduced = true;
    link->dz_impl = dz.unsafeGetTensorImpl();
    link->dz_version = dz._version();
}
// the producer's side.  0: nothing was fused; 1: `dy` is the dz a consumer produced and the accumulator holds its sums;
// 2: a consumer accumulated into bwd_sums but `dy` is not its tensor any more -- reduce again, on a fresh accumulator
int link_state(const c10::intrusive_ptr<BnLink>& link, const Tensor& dy) {
    if (!link || !link->reduced) return 0;
    return (dy.unsafeGetTensorImpl() == link->dz_impl && dy._version() == link->dz_version && nhwc_bf16(dy)) ? 1 : 2;
}

// pre: 0 / 1 / 2 as link_state
BnGrads bn_backward(Tensor dy, const Tensor& x, const Tensor& y, const Tensor& stats, const Tensor& weight, Tensor sums_ws, Tensor bwd_sums,
                    Tensor flags, bool relu, bool has_res, int pre = 0) {
    if (!nhwc_bf16(dy)) dy = dy.to(at::kBFloat16).contiguous(at::MemoryFormat::ChannelsLast);
    const int64_t B = x.size(0), C = x.size(1), H = x.size(2), W = x.size(3);
    int* fl = flags.data_ptr<int>();
    if (pre == 2) g_bn_refused += 1;
    if (pre == 1) {         // dy is dz, the sums are in: the apply pass alone (3 passes over the tensor instead of 5 .. 8)
        g_bn_fused += 1;
        BnGrads g;
        g.dx = at::empty_like(x);
        if (has_res) g.dres = dy;                       // the gradient of the shortcut input IS dz
        Tensor pg = at::empty({2 * C}, stats.options());
        const float* sp = stats.data_ptr<float>();
        ScopedTimer timer("bn_bwd_apply", 0.0, 2.0 * (double)x.numel() * 3, current_stream(x));
        check(epi_bn_act_bwd_reduced(dy.data_ptr(), x.data_ptr(), B * H * W, (int)C, weight.data_ptr<float>(), sp, sp + C, sp + 2 * C,
                                     bwd_sums.data_ptr<float>(), g.dx.data_ptr(), sums_ws.data_ptr<float>(), pg.data_ptr<float>(), current_stream(x)),
              "epi_bn_act_bwd_reduced");
        fl[0] = 0;
        fl[1] = 1;
        g.dbeta = pg.slice(0, 0, C);
        g.dgamma = pg.slice(0, C, 2 * C);
        return g;
    }
    // bwd_sums was cleared by this layer's forward pass; a second backward without a forward in between gets a fresh accumulator
    Tensor sums = (fl[1] || pre == 2) ? at::zeros({2 * C}, stats.options()) : bwd_sums;
    BnGrads g;
    g.dx = at::empty_like(x);
    if (has_res) g.dres = at::empty_like(x);
    // the parameter gradients leave in their OWN memory: the accumulator is cleared by the next forward, which must not wipe
    // a gradient that is still waiting for the optimizer (gradient accumulation, a forward between backward and step)
    Tensor pg = at::empty({2 * C}, stats.options());
    const float* sp = stats.data_ptr<float>();
    const double tbytes = 2.0 * (double)x.numel();
    const int reads = 2 + (y.defined() ? 1 : 0);
    ScopedTimer timer("bn_bwd_reduce+apply", 0.0, tbytes * (2 * reads + 1 + (has_res ? 1 : 0)), current_stream(x));
    check(epi_bn_act_bwd(dy.data_ptr(), x.data_ptr(), y.defined() ? y.data_ptr() : nullptr, B * H * W, (int)C, weight.data_ptr<float>(), sp,
                         sp + C, sp + 2 * C, relu ? 1 : 0, sums.data_ptr<float>(), g.dx.data_ptr(), has_res ? g.dres.data_ptr() : nullptr,
                         sums_ws.data_ptr<float>(), pg.data_ptr<float>(), current_stream(x)),
          "epi_bn_act_bwd");
    fl[0] = 0;
    fl[1] = 1;
    g.dbeta = pg.slice(0, 0, C);
    g.dgamma = pg.slice(0, C, 2 * C);
    return g;
}

struct BnAct : public torch::autograd::Function<BnAct> {
    static Tensor forward(AutogradContext* ctx, Tensor x, Tensor weight, Tensor bias, c10::optional<Tensor> residual_opt, Tensor running_mean,
                          Tensor running_var, Tensor num_batches, Tensor sums_ws, Tensor bwd_sums, Tensor flags, bool training,
                          double momentum, double eps, bool relu) {
        TORCH_CHECK(x.is_cuda(), "FusedBatchNormAct: input must live on the GPU (no CPU fallback in epipolarpose_amd)");
        TORCH_CHECK(nhwc_bf16(x), "FusedBatchNormAct: x must be channels_last bf16");
        const bool has_res = residual_opt.has_value() && residual_opt->defined();
        const Tensor residual = has_res ? *residual_opt : Tensor();
        BnBuffers b{weight, bias, running_mean, running_var, num_batches, sums_ws, bwd_sums, flags};
        Tensor stats;
        Tensor y = bn_forward(x, residual, b, training, momentum, eps, relu, &stats);
        ctx->saved_data["training"] = training;
        if (training) {
            ctx->saved_data["relu"] = relu;
            ctx->saved_data["has_res"] = has_res;
            ctx->saved_data["sums_ws"] = sums_ws;
            ctx->saved_data["bwd_sums"] = bwd_sums;
            ctx->saved_data["flags"] = flags;
            ctx->save_for_backward({x, (relu && has_res) ? y : Tensor(), stats, weight});
        }
        return y;
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        TORCH_CHECK(ctx->saved_data["training"].toBool(),
                    "FusedBatchNormAct: backward through inference-mode statistics is not supported");
        const auto saved = ctx->get_saved_variables();
        const bool relu = ctx->saved_data["relu"].toBool(), has_res = ctx->saved_data["has_res"].toBool();
        BnGrads g = bn_backward(grads[0], saved[0], saved[1], saved[2], saved[3], ctx->saved_data["sums_ws"].toTensor(),
                                ctx->saved_data["bwd_sums"].toTensor(), ctx->saved_data["flags"].toTensor(), relu, has_res);
        return {g.dx, g.dgamma, g.dbeta, g.dres, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(),
                Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

Tensor bn_act(Tensor x, Tensor weight, Tensor bias, c10::optional<Tensor> residual, Tensor running_mean, Tensor running_var,
              Tensor num_batches, Tensor sums_ws, Tensor bwd_sums, Tensor flags, bool training, double momentum, double eps, bool relu) {
    return BnAct::apply(x, weight, bias, residual, running_mean, running_var, num_batches, sums_ws,
                        bwd_sums, flags, training, momentum, eps, relu);
}

// ---- Conv2d -> BatchNorm (+ residual) (+ ReLU): the stage shared by conv_bn_act (one stage = one autograd node) and residual_unit
//      (a whole BasicBlock / Bottleneck = one node) ------------------------------------------------------------------------
// w: [Cout, Cin, k, k], bf16 training copy (gradient returned in bf16, channels_last order) or the fp32 master (converted per
// call, gradient returned in fp32); w_bwd: the packed backward-data operand kept up to date by the optimizer
// (epi_conv2d_pack_weight_bwd_multi after every step), or undefined / empty -> packed here.
inline Tensor channels_last_bf16_weight(const Tensor& w) {
    Tensor w16 = w.scalar_type() == at::kBFloat16 ? w : w.to(at::kBFloat16);
    if (w16.is_contiguous(at::MemoryFormat::ChannelsLast)) return w16;
    if (w16.size(2) == 1 && w16.size(3) == 1 && w16.is_contiguous()) return w16;          // 1x1: the same memory either way
    return w16.contiguous(at::MemoryFormat::ChannelsLast);
}

struct StageParams {      // what the Python side hands over per conv/bn pair
    Tensor w, w_bwd, gamma, beta, running_mean, running_var, num_batches, sums_ws, bwd_sums, flags;
    int64_t stride, pad;
    bool relu;
};
constexpr int STAGE_TENSORS = 10;

struct StageSaved {       // what a stage's backward needs
    Tensor x, raw, y, stats, gamma, wb, sums_ws, bwd_sums, flags;
    std::vector<int64_t> w_sizes, w_strides;
    int K, S, P;
    bool relu, has_res, w_f32, need_dx;
    Tensor w;                           // the weight itself (a parameter or its training copy): consulted by the deferred-reduce rule
    Tensor x_bn_stats;                  // defined: `x` is the RAW output of the stage in front and this stage's input was relu(bn(x)), never written
                                        // (BnInput); [4C] mean | rstd | scale | shift of that BatchNorm, filled by this stage's convolution launch
};

// The BatchNorm + ReLU still pending on a stage's input: the stage in front ran its convolution only (defer_bn) and delivered its batch sums; this
// stage's 1x1 convolution normalises its A operand on the fly (epi_conv1x1_fwd_bn_in) -- one apply launch and one tensor write + read less per
// bottleneck (round 4).  MEASURED AND NOT ADOPTED (profiles/r04_ab_bn_in_fuse.txt, one box, interleaved arms): 16 apply launches and 0.13 ms of
// BatchNorm time disappear, but the forward convolutions gain 0.10 ms (every workgroup derives the K-channel table, every wave transforms the
// fragments it reads) and the weight gradients 0.085 ms (the TN kernel re-applies the affine to the same activation in every output tile: 4 .. 32x
// redundant work beside its MFMAs) -- 6.363 / 6.380 ms per step with it against 6.317 / 6.345 ms without.  Off by default; EPI_BN_IN_FUSE=1 /
// bn_in_fuse_mode(1) turns it on (tests/test_hip_conv.py keeps it correct).
struct BnInput {
    const StageParams* sp = nullptr;    // the stage in front
    Tensor stats;                       // [4C], becomes that stage's saved statistics
};
int g_bn_in_fuse = -1;
bool bn_in_fuse_enabled() {
    if (g_bn_in_fuse < 0) g_bn_in_fuse = 0;
    return g_bn_in_fuse != 0;
}
int bn_in_fuse_mode(int mode) {                 // test / measurement hook: returns the previous setting; a negative mode only queries
    const int prev = bn_in_fuse_enabled() ? 1 : 0;
    if (mode >= 0) g_bn_in_fuse = mode ? 1 : 0;
    return prev;
}
int64_t g_bn_in_fused = 0;                       // convolutions that normalised their input themselves (tests count them)
int64_t bn_in_fuse_count(bool reset) {
    const int64_t v = g_bn_in_fused;
    if (reset) g_bn_in_fused = 0;
    return v;
}

// defer_bn: run the convolution (and its epilogue statistics) only and return the RAW output -- the caller normalises it together with another
// stage's (dual_bn_forward); *sums_done_out then says whether the batch sums are already in sums_ws
Tensor stage_forward(const Tensor& x_in, const StageParams& sp, const Tensor& residual, bool training, double momentum, double eps, bool need_dx,
                     StageSaved* save, bool defer_bn = false, int* sums_done_out = nullptr, const BnInput* bn_input = nullptr) {
    Tensor x = x_in;
    TORCH_CHECK(x.is_cuda() && sp.w.is_cuda(), "conv_bn_act: tensors must live on the GPU (no CPU fallback in epipolarpose_amd)");
    if (!nhwc_bf16(x)) x = x.to(at::kBFloat16).contiguous(at::MemoryFormat::ChannelsLast);
    const Tensor& w = sp.w;
    TORCH_CHECK(w.dim() == 4 && w.size(1) == x.size(1) && w.size(2) == w.size(3), "conv_bn_act: weight shape");
    const bool has_res = residual.defined();
    const int B = (int)x.size(0), Cin = (int)x.size(1), H = (int)x.size(2), W = (int)x.size(3);
    const int Cout = (int)w.size(0), K = (int)w.size(2), S = (int)sp.stride, P = (int)sp.pad;
    const int Ho = (H + 2 * P - K) / S + 1, Wo = (W + 2 * P - K) / S + 1;
    const Tensor w16 = channels_last_bf16_weight(w.detach());
    Tensor raw = at::empty({B, Cout, Ho, Wo}, x.options().memory_format(at::MemoryFormat::ChannelsLast));
    Tensor& ws = workspace(epi_conv2d_workspace_bytes(B, H, W, Cin, Cout, K, K, S, P), x);
    int sums_done = 0;
    Tensor sums_ws = sp.sums_ws;
    TORCH_CHECK(!training || sums_ws.numel() == 2 * (int64_t)Cout * epi_bn_sum_copies(Cout), "conv_bn_act: sums_ws must be [epi_bn_sum_copies(C)][2C]");
    if (training) {             // the accumulator hand-over of bn_forward, done here because the GEMM epilogue may fill sums_ws
        int* fl = sp.flags.data_ptr<int>();
        if (fl[0]) sums_ws.zero_();
        fl[0] = 1;
        fl[1] = 0;
    }
    const double conv_flops = 2.0 * B * Ho * Wo * (double)Cout * Cin * K * K;
    const double conv_bytes = 2.0 * ((double)x.numel() + (double)raw.numel() + (double)w.numel());
    if (bn_input) {             // x is the raw output of the stage in front: normalised inside this launch (the caller checked the geometry)
        const StageParams& ip = *bn_input->sp;
        float* st = bn_input->stats.data_ptr<float>();
        EpiBnLayer l;
        l.gamma = ip.gamma.data_ptr<float>(); l.beta = ip.beta.data_ptr<float>();
        l.running_mean = ip.running_mean.data_ptr<float>(); l.running_var = ip.running_var.data_ptr<float>();
        l.num_batches_tracked = reinterpret_cast<long long*>(ip.num_batches.data_ptr<int64_t>());
        l.mean = st; l.rstd = st + Cin; l.scale_shift = st + 2 * Cin;
        l.sums_ws = ip.sums_ws.data_ptr<float>(); l.bwd_sums = ip.bwd_sums.data_ptr<float>();
        ScopedTimer timer("conv_fwd", conv_flops, conv_bytes, current_stream(x));
        check(epi_conv1x1_fwd_bn_in(x.data_ptr(), &l, training ? 2 : 0, (float)eps, (float)momentum, w16.data_ptr(), raw.data_ptr(), B, H, W, Cin, Cout,
                                    training ? sums_ws.data_ptr<float>() : nullptr, training ? &sums_done : nullptr, ws.data_ptr(), (size_t)ws.numel(),
                                    current_stream(x)), "epi_conv1x1_fwd_bn_in");
        g_bn_in_fused += 1;
    } else {
        ScopedTimer timer("conv_fwd", conv_flops, conv_bytes, current_stream(x));
        check(epi_conv2d_fwd(x.data_ptr(), w16.data_ptr(), raw.data_ptr(), B, H, W, Cin, Cout, K, K, S, P,
                             training ? sums_ws.data_ptr<float>() : nullptr, training ? &sums_done : nullptr, ws.data_ptr(),
                             (size_t)ws.numel(), current_stream(x)),
              "epi_conv2d_fwd");
    }
    BnBuffers b{sp.gamma, sp.beta, sp.running_mean, sp.running_var, sp.num_batches, sp.sums_ws, sp.bwd_sums, sp.flags};
    Tensor stats;
    Tensor y;
    if (sums_done_out) *sums_done_out = sums_done;
    if (defer_bn) {
        y = raw;                        // (statistics, y and the saved mask source are filled in by dual_bn_forward)
    } else if (training && !sums_done) {       // the statistics pass runs separately; re-arm the flag protocol for bn_forward
        sp.flags.data_ptr<int>()[0] = 0;
        y = bn_forward(raw, residual, b, training, momentum, eps, sp.relu, &stats, false);
    } else {
        y = bn_forward(raw, residual, b, training, momentum, eps, sp.relu, &stats, training);
    }
    if (training && save) {
        Tensor wb;
        if (need_dx) {                              // backward-data operand: the optimizer's packed copy, or packed now
            if (sp.w_bwd.defined() && sp.w_bwd.numel() == w.numel()) {
                wb = sp.w_bwd;
                TORCH_CHECK(wb.scalar_type() == at::kBFloat16, "conv_bn_act: packed weight dtype");
            } else {
                wb = at::empty({w.numel()}, w16.options().memory_format(at::MemoryFormat::Contiguous));
                check(epi_conv2d_pack_weight_bwd(w16.data_ptr(), Cout, Cin, K, K, S, P, wb.data_ptr(), current_stream(x)),
                      "epi_conv2d_pack_weight_bwd");
            }
        }
        save->x = x; save->raw = raw;
        // a detached alias: the returned tensor itself will carry this node as grad_fn -- holding it would close a reference cycle
        save->y = (sp.relu && has_res && !defer_bn) ? y.detach() : Tensor(); save->stats = stats; save->gamma = sp.gamma; save->wb = wb;
        save->sums_ws = sp.sums_ws; save->bwd_sums = sp.bwd_sums; save->flags = sp.flags;
        save->w_sizes = w.sizes().vec();
        save->w_strides = (w.is_contiguous(at::MemoryFormat::ChannelsLast) || (K == 1 && w.is_contiguous())) ? w.strides().vec() : w16.strides().vec();
        save->K = K; save->S = S; save->P = P; save->relu = sp.relu; save->has_res = has_res; save->w_f32 = w.scalar_type() != at::kBFloat16;
        save->w = w;
        save->need_dx = need_dx;
        save->x_bn_stats = bn_input ? bn_input->stats : Tensor();
    }
    return y;
}

// may the convolution of `next` normalise its own input (the raw output of a stage with `c_in` channels)?  (epi_conv1x1_fwd_bn_in's conditions)
bool bn_input_eligible(const StageParams& next, int64_t c_in, bool training) {
    return bn_in_fuse_enabled() && training && next.w.dim() == 4 && next.w.size(2) == 1 && next.w.size(3) == 1 && next.stride == 1 && next.pad == 0 &&
           c_in % 64 == 0 && c_in <= 2048 && next.w.size(0) % 8 == 0 && epi_set_deterministic(-1) == 0;
}
// the separate pass after all: normalise `raw` of a stage that ran with defer_bn (its consumer turned out not to take it raw)
Tensor finish_deferred_stage(const Tensor& raw, const StageParams& p, const Tensor& res, int sums_done, bool training, double momentum, double eps, StageSaved* sv) {
    BnBuffers b{p.gamma, p.beta, p.running_mean, p.running_var, p.num_batches, p.sums_ws, p.bwd_sums, p.flags};
    Tensor stats;
    if (training && !sums_done) p.flags.data_ptr<int>()[0] = 0;
    Tensor out_y = bn_forward(raw, res, b, training, momentum, eps, p.relu, &stats, training && sums_done != 0);
    if (sv) {
        sv->stats = stats;
        sv->has_res = res.defined();
        sv->y = (p.relu && res.defined()) ? out_y.detach() : Tensor();
    }
    return out_y;
}

// y = relu(bn_main(raw_main) + bn_proj(raw_proj)) in one pass (epi_bn_act_fwd_dual): the last stage of a residual unit with a projection shortcut.
// Both stages ran with defer_bn; their saved state gets the statistics (and the main stage the mask source) here.
int g_bn_dual = -1;
bool bn_dual_enabled() {
    if (g_bn_dual < 0) g_bn_dual = 1;
    return g_bn_dual != 0;
}
int bn_dual_mode(int mode) {                    // test / measurement hook: returns the previous setting; a negative mode only queries
    const int prev = bn_dual_enabled() ? 1 : 0;
    if (mode >= 0) g_bn_dual = mode ? 1 : 0;
    return prev;
}
Tensor dual_bn_forward(const Tensor& raw_main, const StageParams& pm, const Tensor& raw_proj, const StageParams& pp, bool training, bool sums_ready,
                       double momentum, double eps, StageSaved* save_main, StageSaved* save_proj) {
    const int64_t B = raw_main.size(0), C = raw_main.size(1), H = raw_main.size(2), W = raw_main.size(3);
    TORCH_CHECK(raw_proj.sizes() == raw_main.sizes() && pm.gamma.numel() == C && pp.gamma.numel() == C, "residual_unit: projection shape");
    Tensor y = at::empty_like(raw_main);
    Tensor stats_m = at::empty({4 * C}, pm.gamma.options().dtype(at::kFloat)), stats_p = at::empty({4 * C}, pm.gamma.options().dtype(at::kFloat));
    auto layer = [&](const StageParams& p, Tensor& st) {
        EpiBnLayer l;
        float* sp = st.data_ptr<float>();
        l.gamma = p.gamma.data_ptr<float>(); l.beta = p.beta.data_ptr<float>();
        l.running_mean = p.running_mean.data_ptr<float>(); l.running_var = p.running_var.data_ptr<float>();
        l.num_batches_tracked = reinterpret_cast<long long*>(p.num_batches.data_ptr<int64_t>());
        l.mean = sp; l.rstd = sp + C; l.scale_shift = sp + 2 * C;
        l.sums_ws = p.sums_ws.data_ptr<float>(); l.bwd_sums = p.bwd_sums.data_ptr<float>();
        return l;
    };
    const EpiBnLayer lm = layer(pm, stats_m), lp = layer(pp, stats_p);
    const double tbytes = 2.0 * (double)raw_main.numel();
    ScopedTimer timer(training ? (sums_ready ? "bn_fwd_apply" : "bn_fwd_stats+apply") : "bn_fwd_eval", 0.0, tbytes * ((training && !sums_ready ? 2 : 0) + 3),
                      current_stream(raw_main));
    check(epi_bn_act_fwd_dual(raw_main.data_ptr(), raw_proj.data_ptr(), B * H * W, (int)C, &lm, &lp, (float)eps, (float)momentum,
                              training ? (sums_ready ? 2 : 1) : 0, y.data_ptr(), current_stream(raw_main)), "epi_bn_act_fwd_dual");
    if (training) {
        if (save_main) { save_main->stats = stats_m; save_main->y = y.detach(); save_main->has_res = true; }
        if (save_proj) { save_proj->stats = stats_p; save_proj->has_res = false; }
    }
    return y;
}

struct StageGrads { Tensor dx, dw, dgamma, dbeta, dres; };

// dy: gradient of the stage output; addend: added to dx in the backward-data epilogue (the other branch of a residual junction).
// pre: state of dy (link_state: 1 = already dz with the sums in the accumulator).  feeds / feeds_link: the BatchNorm stage whose output
// this stage consumed -- inside the same node (feeds) or in the producing node (feeds_link): its reduction is fused into this stage's
// backward-data launch where the library can (*fed = true: dx is that layer's dz).
// addend_step 2: `addend` is the half-resolution gradient of the even pixels (a 1x1 stride-2 projection's backward-data left at half resolution).
// half_dx: this stage IS such a projection -- return its input gradient at half resolution (a plain 1x1 backward-data on the output grid).
StageGrads stage_backward(const Tensor& dy, const StageSaved& sv, bool need_dx, bool need_dw, const Tensor& addend, int pre = 0,
                          const StageSaved* feeds = nullptr, const c10::intrusive_ptr<BnLink>& feeds_link = c10::intrusive_ptr<BnLink>(),
                          bool* fed = nullptr, int addend_step = 1, bool half_dx = false) {
    if (fed) *fed = false;
    BnGrads g = bn_backward(dy, sv.raw, sv.y, sv.stats, sv.gamma, sv.sums_ws, sv.bwd_sums, sv.flags, sv.relu, sv.has_res, pre);
    const Tensor& x = sv.x;
    const int K = sv.K, S = sv.S, P = sv.P;
    const int B = (int)x.size(0), Cin = (int)x.size(1), H = (int)x.size(2), W = (int)x.size(3), Cout = (int)sv.raw.size(1);
    const int Ho = (int)sv.raw.size(2), Wo = (int)sv.raw.size(3);
    StageGrads out;
    out.dgamma = g.dgamma; out.dbeta = g.dbeta; out.dres = g.dres;
    if (need_dx && half_dx) {
        // 1x1 / stride 2 / pad 0: dx is non-zero at the even pixels only, where it is dz * W^T of the matching output pixel -- computed on the OUTPUT grid
        // (the packed backward weight of a 1x1 stride-2 layer is [Cin][Cout], the stride-1 layout) and handed on at half resolution
        TORCH_CHECK(sv.wb.defined() && K == 1 && S == 2 && P == 0 && !(H & 1) && !(W & 1), "conv_bn_act: half-resolution input gradient: geometry");
        out.dx = at::empty({B, Cin, Ho, Wo}, x.options().memory_format(at::MemoryFormat::ChannelsLast));
        Tensor& ws = workspace(epi_conv2d_workspace_bytes(B, Ho, Wo, Cin, Cout, 1, 1, 1, 0), x);
        ScopedTimer timer("conv_bwd_data", 2.0 * B * Ho * Wo * (double)Cout * Cin, 2.0 * ((double)out.dx.numel() + (double)sv.raw.numel() + (double)sv.wb.numel()),
                          current_stream(x));
        check(epi_conv2d_bwd_data(g.dx.data_ptr(), sv.wb.data_ptr(), out.dx.data_ptr(), B, Ho, Wo, Cin, Cout, 1, 1, 1, 0, nullptr, ws.data_ptr(), (size_t)ws.numel(),
                                  current_stream(x)), "epi_conv2d_bwd_data");
    } else if (need_dx) {
        TORCH_CHECK(sv.wb.defined(), "conv_bn_act: input gradient requested but no backward-data weight was prepared");
        if (addend.defined() && addend_step == 2) {
            TORCH_CHECK(nhwc_bf16(addend) && addend.size(0) == B && addend.size(1) == Cin && addend.size(2) * 2 == H && addend.size(3) * 2 == W,
                        "conv_bn_act: half-resolution addend layout");
        } else if (addend.defined()) {
            TORCH_CHECK(nhwc_bf16(addend) && addend.sizes() == x.sizes(), "conv_bn_act: residual-junction addend layout");
        }
        out.dx = at::empty_like(x);
        Tensor& ws = workspace(epi_conv2d_workspace_bytes(B, H, W, Cin, Cout, K, K, S, P), x);
        ScopedTimer timer("conv_bwd_data", 2.0 * B * Ho * Wo * (double)Cout * Cin * K * K,
                          2.0 * ((double)x.numel() + (double)sv.raw.numel() + (double)sv.wb.numel()), current_stream(x));
        EpiBnReduce red = {};
        bool want = false;
        if (fed && bn_fuse_enabled()) {
            if (feeds && feeds->raw.defined() && feeds->raw.sizes() == x.sizes() && !feeds->has_res && feeds->flags.data_ptr<int>()[1] == 0) {
                red.z = feeds->raw.data_ptr(); red.y = nullptr; red.bn = feeds->stats.data_ptr<float>();
                red.sums = feeds->bwd_sums.data_ptr<float>(); red.relu = feeds->relu ? 1 : 0;
                want = true;
            } else if (feeds_link) {
                want = link_reduce_args(feeds_link, &red);
            }
        }
        int red_done = 0;
        if (want || (addend.defined() && addend_step == 2))
            check(epi_conv2d_bwd_data_bnred(g.dx.data_ptr(), sv.wb.data_ptr(), out.dx.data_ptr(), B, H, W, Cin, Cout, K, K, S, P,
                                            addend.defined() ? addend.data_ptr() : nullptr, addend.defined() ? addend_step : 1, want ? &red : nullptr, &red_done,
                                            ws.data_ptr(), (size_t)ws.numel(), current_stream(x)), "epi_conv2d_bwd_data_bnred");
        else
            check(epi_conv2d_bwd_data(g.dx.data_ptr(), sv.wb.data_ptr(), out.dx.data_ptr(), B, H, W, Cin, Cout, K, K, S, P,
                                      addend.defined() ? addend.data_ptr() : nullptr, ws.data_ptr(), (size_t)ws.numel(), current_stream(x)),
                  "epi_conv2d_bwd_data");
        if (red_done) {
            *fed = true;
            if (!feeds && feeds_link) link_mark_reduced(feeds_link, out.dx);
        }
    }
    if (need_dw) {
        out.dw = at::empty_strided(sv.w_sizes, sv.w_strides, x.options().dtype(sv.w_f32 ? at::kFloat : at::kBFloat16));
        const size_t slab_bytes = epi_gemm_tn_workspace_bytes(B * Ho * Wo, Cout, Cin, K * K);
        const bool first_use = first_gradient_of_pass(sv.w);
        if (!first_use) flush_pending_reduces();       // a second use of a shared weight: the engine adds it to the first on the main stream
        const bool after_pass = first_use && gradient_consumed_after_backward(sv.w);        // nobody reads this gradient before backward() returns
        const double wflops_g = 2.0 * B * Ho * Wo * (double)Cout * Cin * K * K;
        // (x_bn_stats: x is the raw output of the stage in front -- the kernel re-applies that stage's BatchNorm + ReLU to its operand)
        const float* x_ss = sv.x_bn_stats.defined() ? sv.x_bn_stats.data_ptr<float>() + 2 * Cin : nullptr;
        if (after_pass && defer_enabled() && group_mode() != 0) {      // leaves with its stage's other weight gradients (side_group_flush)
            EpiWgradItem it = {};
            it.x = x.data_ptr(); it.dy = g.dx.data_ptr(); it.dw = out.dw.data_ptr(); it.dw_dtype = sv.w_f32 ? EPI_F32 : EPI_BF16; it.kind = EPI_WGRAD_CONV2D;
            it.B = B; it.H = H; it.W = W; it.Cin = Cin; it.Cout = Cout; it.KH = K; it.KW = K; it.stride = S; it.pad = P;
            it.x_scale_shift = x_ss;
            g_side.group.push_back(SideStream::GroupItem{it, x, g.dx, out.dw.storage().getWeakStorageImpl(), out.dw.device(), wflops_g,
                                                         2.0 * ((double)x.numel() + (double)sv.raw.numel() + (double)out.dw.numel()), sv.x_bn_stats});
            end_of_pass_callback();
            return out;
        }
        // (a gradient somebody reads inside the pass -- a hooked parameter -- is computed right here on the main stream with its own
        //  reduction; whoever reads it flushes what ELSE it needs: the bucket hooks do, asynchronously.  Round 2 joined the streams here.)
        const bool may_defer = slab_bytes && defer_enabled() && after_pass;
        void* slabs = may_defer ? pending_slab_alloc(slab_bytes, x) : nullptr;
        // second stream: only a launch whose result is complete by the end-of-pass join (an arena-less split would reduce right away)
        const bool on_side = side_mode() != 0 && after_pass && (slab_bytes == 0 || slabs != nullptr);
        const double wflops = 2.0 * B * Ho * Wo * (double)Cout * Cin * K * K;
        const double wbytes = 2.0 * ((double)x.numel() + (double)sv.raw.numel() + (double)out.dw.numel());
        EpiWgradItem item = {};
        item.x = x.data_ptr(); item.dy = g.dx.data_ptr(); item.dw = out.dw.data_ptr(); item.dw_dtype = sv.w_f32 ? EPI_F32 : EPI_BF16; item.kind = EPI_WGRAD_CONV2D;
        item.B = B; item.H = H; item.W = W; item.Cin = Cin; item.Cout = Cout; item.KH = K; item.KW = K; item.stride = S; item.pad = P;
        item.x_scale_shift = x_ss;
        if (on_side) {           // launched by side_run_jobs() when the node's main-stream work has been enqueued
            const Tensor xin = x, dyin = g.dx, dwout = out.dw, stats_keep = sv.x_bn_stats;
            g_side.jobs.push_back(SideStream::Job{x, g.dx, out.dw, [=](epi_stream_t st) {
                EpiSlabReduce pend = {};
                Tensor* ws = slabs ? nullptr : &side_workspace(slab_bytes, xin);
                if (stats_keep.defined()) g_side.keep.push_back(stats_keep);
                check(epi_wgrad_item(&item, slabs ? slabs : ws->data_ptr(), slabs ? slab_bytes : (size_t)ws->numel(), slabs ? &pend : nullptr, st),
                      "epi_wgrad_item");
                return pend;
            }, "conv_bwd_weight", wflops, wbytes});
        } else {
            EpiSlabReduce pend = {};
            Tensor* ws = slabs ? nullptr : &workspace(slab_bytes, x);
            {
                ScopedTimer timer("conv_bwd_weight", wflops, wbytes, current_stream(x));
                check(epi_wgrad_item(&item, slabs ? slabs : ws->data_ptr(), slabs ? slab_bytes : (size_t)ws->numel(), slabs ? &pend : nullptr,
                                     current_stream(x)), "epi_wgrad_item");
            }
            if (pend.nsplit > 0) pending_register(pend, out.dw);
        }
    }
    return out;
}

// The saved state of a node lives in a small holder object kept alive by the autograd context (an IValue capsule, released with the
// graph): tensors inside
// are plain references (the differentiable inputs among them -- x of the first stage -- are recorded through save_for_backward).
struct SavedHolder : torch::CustomClassHolder {
    std::vector<StageSaved> stages;
    bool has_downsample = false;
    int n_main = 0;
    c10::intrusive_ptr<BnLink> in_link, out_link;      // the BatchNorm that produced this node's input / this node's own last BatchNorm
};
// publish the node's output (the last main-path stage of `holder`)
void publish_output(const c10::intrusive_ptr<SavedHolder>& holder, const Tensor& y, const StageSaved& sv) {
    holder->out_link = link_publish(y, sv.raw, sv.y, sv.stats, sv.bwd_sums, sv.flags, sv.relu);
}

struct ConvBnAct : public torch::autograd::Function<ConvBnAct> {
    static Tensor forward(AutogradContext* ctx, Tensor x, Tensor w, c10::optional<Tensor> w_bwd_opt, int64_t stride, int64_t pad, Tensor gamma,
                          Tensor beta, c10::optional<Tensor> residual_opt, Tensor running_mean, Tensor running_var, Tensor num_batches,
                          Tensor sums_ws, Tensor bwd_sums, Tensor flags, bool training, double momentum, double eps, bool relu) {
        StageParams sp{w, (w_bwd_opt.has_value() && w_bwd_opt->defined()) ? *w_bwd_opt : Tensor(), gamma, beta, running_mean, running_var,
                       num_batches, sums_ws, bwd_sums, flags, stride, pad, relu};
        const Tensor residual = (residual_opt.has_value() && residual_opt->defined()) ? *residual_opt : Tensor();
        auto holder = c10::make_intrusive<SavedHolder>();
        holder->stages.resize(1);
        if (training && x.requires_grad() && nhwc_bf16(x)) holder->in_link = link_claim(x);
        Tensor y = stage_forward(x, sp, residual, training, momentum, eps, x.requires_grad(), &holder->stages[0]);
        ctx->saved_data["training"] = training;
        if (training) {
            publish_output(holder, y, holder->stages[0]);
            ctx->saved_data["holder"] = c10::IValue::make_capsule(holder);
        }
        return y;
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        TORCH_CHECK(ctx->saved_data["training"].toBool(), "conv_bn_act: backward through inference-mode statistics is not supported");
        auto holder = c10::static_intrusive_pointer_cast<SavedHolder>(ctx->saved_data["holder"].toCapsule());
        TORCH_CHECK(!holder->stages.empty(), "conv_bn_act: backward called twice (the fused nodes free their activations in backward)");
        g_side.jobs.clear();                         // (left-overs of a pass that aborted inside a node)
        bool fed = false;
        StageGrads g = stage_backward(grads[0], holder->stages[0], ctx->needs_input_grad(0), ctx->needs_input_grad(1), Tensor(),
                                      link_state(holder->out_link, grads[0]), nullptr, holder->in_link, &fed);
        link_retire(holder->out_link);
        side_run_jobs();
        side_group_flush();
        holder->stages.clear();                      // release the saved activations now, not when the graph is torn down
        holder->in_link.reset();
        holder->out_link.reset();
        return {g.dx, g.dw, Tensor(), Tensor(), Tensor(), g.dgamma, g.dbeta, g.dres, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(),
                Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

Tensor conv_bn_act(Tensor x, Tensor w, c10::optional<Tensor> w_bwd, int64_t stride, int64_t pad, Tensor gamma, Tensor beta,
                   c10::optional<Tensor> residual, Tensor running_mean, Tensor running_var, Tensor num_batches, Tensor sums_ws, Tensor bwd_sums,
                   Tensor flags, bool training, double momentum, double eps, bool relu) {
    return ConvBnAct::apply(x, w, w_bwd, stride, pad, gamma, beta, residual, running_mean, running_var, num_batches, sums_ws, bwd_sums, flags,
                            training, momentum, eps, relu);
}

// ---- The stem: Conv2d(3, C, 7, stride 2, pad 3) -> BatchNorm -> ReLU (pose3d_resnet.py:99-103,186-188) as ONE autograd node ------------
// The convolution runs on the implicit-GEMM kernels through the space-to-depth form of the image (epi_stem7x7s2_*: csrc/head_gemm.hip);
// rounds 1-2 left it to MIOpen.  x: [B, 3, H, W] f32 or bf16, NCHW or channels_last; needs no gradient.  w: [C, 3, 7, 7] bf16 training copy
// or fp32 master.  The weight gradient is a TN GEMM on the second stream followed by a 9 408-element re-ordering into the parameter's layout.
struct StemConvBnAct : public torch::autograd::Function<StemConvBnAct> {
    // pool: also the stem's MaxPool2d(3, 2, 1) (pose3d_resnet.py:104,186), applied to bf16(relu(bn(conv))) straight from the RAW convolution output
    // (epi_bn_finalize + epi_maxpool3x3s2_bn_relu_fwd): the normalised 67 MB tensor, whose only reader is the pool, is never written
    static Tensor forward(AutogradContext* ctx, Tensor x, Tensor w, Tensor gamma, Tensor beta, Tensor running_mean, Tensor running_var,
                          Tensor num_batches, Tensor sums_ws, Tensor bwd_sums, Tensor flags, bool training, double momentum, double eps, bool relu,
                          bool pool) {
        TORCH_CHECK(x.is_cuda() && w.is_cuda(), "stem_conv_bn_act: tensors must live on the GPU (no CPU fallback in epipolarpose_amd)");
        TORCH_CHECK(x.dim() == 4 && x.size(1) == 3 && w.dim() == 4 && w.size(1) == 3 && w.size(2) == 7 && w.size(3) == 7, "stem_conv_bn_act: shapes");
        TORCH_CHECK(x.scalar_type() == at::kFloat || x.scalar_type() == at::kBFloat16, "stem_conv_bn_act: image dtype");
        const int B = (int)x.size(0), H = (int)x.size(2), W = (int)x.size(3), Cout = (int)w.size(0);
        const bool nhwc = x.is_contiguous(at::MemoryFormat::ChannelsLast);
        if (!nhwc && !x.is_contiguous()) x = x.contiguous();
        Tensor w16 = w.detach();
        if (w16.scalar_type() != at::kBFloat16) w16 = w16.to(at::kBFloat16);
        const bool w_cl = w16.is_contiguous(at::MemoryFormat::ChannelsLast);
        if (!w_cl && !w16.is_contiguous()) w16 = w16.contiguous();
        const auto bf = x.options().dtype(at::kBFloat16).memory_format(at::MemoryFormat::Contiguous);
        const size_t s2d_bytes = epi_stem7x7s2_s2d_bytes(B, H, W);
        TORCH_CHECK(s2d_bytes > 0, "stem_conv_bn_act: image extent must be even");
        Tensor s2d = at::empty({(int64_t)(s2d_bytes / 2)}, bf);
        Tensor wp = at::empty({(int64_t)Cout * 256}, bf);
        const epi_stream_t st = current_stream(x);
        check(epi_stem7x7s2_s2d(x.data_ptr(), x.scalar_type() == at::kFloat ? EPI_F32 : EPI_BF16, nhwc ? EPI_NHWC : EPI_NCHW, B, H, W, s2d.data_ptr(), st),
              "epi_stem7x7s2_s2d");
        check(epi_stem7x7s2_pack_weight(w16.data_ptr(), w_cl ? 1 : 0, Cout, wp.data_ptr(), st), "epi_stem7x7s2_pack_weight");
        Tensor raw = at::empty({B, Cout, H / 2, W / 2}, x.options().dtype(at::kBFloat16).memory_format(at::MemoryFormat::ChannelsLast));
        TORCH_CHECK(!training || sums_ws.numel() == 2 * (int64_t)Cout * epi_bn_sum_copies(Cout), "stem_conv_bn_act: sums_ws must be [epi_bn_sum_copies(C)][2C]");
        if (training) {
            int* fl = flags.data_ptr<int>();
            if (fl[0]) sums_ws.zero_();
            fl[0] = 1;
            fl[1] = 0;
        }
        int sums_done = 0;
        const double flops = 2.0 * B * (H / 2) * (W / 2) * (double)Cout * 147.0;
        {
            Tensor& ws = workspace(epi_stem7x7s2_workspace_bytes(B, H, W, Cout), raw);
            ScopedTimer timer("stem_conv_fwd", flops, 2.0 * ((double)x.numel() + (double)raw.numel()), st);
            check(epi_stem7x7s2_fwd(s2d.data_ptr(), wp.data_ptr(), raw.data_ptr(), B, H, W, Cout, training ? sums_ws.data_ptr<float>() : nullptr,
                                    training ? &sums_done : nullptr, ws.data_ptr(), (size_t)ws.numel(), st), "epi_stem7x7s2_fwd");
        }
        BnBuffers b{gamma, beta, running_mean, running_var, num_batches, sums_ws, bwd_sums, flags};
        Tensor stats, y, pos;
        const bool pooled = pool && relu && (!training || sums_done) && Cout % 8 == 0;
        if (pooled) {
            const int Hc = H / 2, Wc = W / 2, Hp = (Hc - 1) / 2 + 1, Wp = (Wc - 1) / 2 + 1;
            stats = at::empty({4 * (int64_t)Cout}, gamma.options().dtype(at::kFloat));
            float* sp = stats.data_ptr<float>();
            EpiBnLayer l;
            l.gamma = gamma.data_ptr<float>(); l.beta = beta.data_ptr<float>(); l.running_mean = running_mean.data_ptr<float>();
            l.running_var = running_var.data_ptr<float>(); l.num_batches_tracked = reinterpret_cast<long long*>(num_batches.data_ptr<int64_t>());
            l.mean = sp; l.rstd = sp + Cout; l.scale_shift = sp + 2 * Cout; l.sums_ws = sums_ws.data_ptr<float>(); l.bwd_sums = bwd_sums.data_ptr<float>();
            check(epi_bn_finalize(&l, (long long)B * Hc * Wc, Cout, (float)eps, (float)momentum, training ? 2 : 0, st), "epi_bn_finalize");
            y = at::empty({B, Cout, Hp, Wp}, raw.options().memory_format(at::MemoryFormat::ChannelsLast));
            pos = at::empty({B, Hp, Wp, Cout}, raw.options().dtype(at::kByte).memory_format(at::MemoryFormat::Contiguous));
            ScopedTimer timer("maxpool_fwd", 0.0, 2.0 * (double)raw.numel() + 3.0 * (double)y.numel(), st);
            check(epi_maxpool3x3s2_bn_relu_fwd(raw.data_ptr(), sp + 2 * Cout, y.data_ptr(), pos.data_ptr(), B, Hc, Wc, Cout, st), "epi_maxpool3x3s2_bn_relu_fwd");
        } else {
            if (training && !sums_done) {
                flags.data_ptr<int>()[0] = 0;
                y = bn_forward(raw, Tensor(), b, training, momentum, eps, relu, &stats, false);
            } else {
                y = bn_forward(raw, Tensor(), b, training, momentum, eps, relu, &stats, training);
            }
            if (pool) {                 // (the fused form was not applicable: the two passes)
                const int64_t Hc = y.size(2), Wc = y.size(3), Hp = (Hc - 1) / 2 + 1, Wp = (Wc - 1) / 2 + 1;
                Tensor yp = at::empty({B, Cout, Hp, Wp}, y.options().memory_format(at::MemoryFormat::ChannelsLast));
                pos = at::empty({B, Hp, Wp, Cout}, y.options().dtype(at::kByte).memory_format(at::MemoryFormat::Contiguous));
                ScopedTimer timer("maxpool_fwd", 0.0, 2.0 * (double)y.numel() + 3.0 * (double)yp.numel(), st);
                check(epi_maxpool3x3s2_fwd(y.data_ptr(), yp.data_ptr(), pos.data_ptr(), B, (int)Hc, (int)Wc, Cout, st), "epi_maxpool3x3s2_fwd");
                y = yp;
            }
        }
        ctx->saved_data["training"] = training;
        ctx->saved_data["pool"] = pool;
        if (pool && training) ctx->saved_data["pos"] = pos;
        if (training) {
            auto holder = c10::make_intrusive<SavedHolder>();
            holder->stages.resize(1);
            StageSaved& sv = holder->stages[0];
            sv.x = s2d; sv.raw = raw; sv.stats = stats; sv.gamma = gamma; sv.sums_ws = sums_ws; sv.bwd_sums = bwd_sums; sv.flags = flags;
            sv.relu = relu; sv.has_res = false; sv.w_f32 = w.scalar_type() != at::kBFloat16; sv.w = w;
            sv.w_sizes = w.sizes().vec();
            sv.w_strides = w.strides().vec();
            sv.K = H; sv.S = W; sv.P = B;                    // (geometry of the image, reusing the integer slots)
            ctx->saved_data["holder"] = c10::IValue::make_capsule(holder);
        }
        return y;
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        TORCH_CHECK(ctx->saved_data["training"].toBool(), "stem_conv_bn_act: backward through inference-mode statistics is not supported");
        auto holder = c10::static_intrusive_pointer_cast<SavedHolder>(ctx->saved_data["holder"].toCapsule());
        TORCH_CHECK(!holder->stages.empty(), "stem_conv_bn_act: backward called twice (the fused nodes free their activations in backward)");
        g_side.jobs.clear();
        const StageSaved& sv = holder->stages[0];
        Tensor dy = grads[0];
        if (ctx->saved_data["pool"].toBool()) {         // the pool's backward first: dy of the BatchNorm output from the pooled gradient and the window positions
            const Tensor pos = ctx->saved_data["pos"].toTensor();
            if (!nhwc_bf16(dy)) dy = dy.to(at::kBFloat16).contiguous(at::MemoryFormat::ChannelsLast);
            Tensor dfull = at::empty_like(sv.raw);
            ScopedTimer timer("maxpool_bwd", 0.0, 3.0 * (double)dy.numel() + 2.0 * (double)dfull.numel(), current_stream(dy));
            check(epi_maxpool3x3s2_bwd(dy.data_ptr(), pos.data_ptr(), dfull.data_ptr(), (int)sv.raw.size(0), (int)sv.raw.size(2), (int)sv.raw.size(3),
                                       (int)sv.raw.size(1), current_stream(dy)), "epi_maxpool3x3s2_bwd");
            dy = dfull;
        }
        BnGrads g = bn_backward(dy, sv.raw, Tensor(), sv.stats, sv.gamma, sv.sums_ws, sv.bwd_sums, sv.flags, sv.relu, false);
        const int H = sv.K, W = sv.S, B = sv.P, Cout = (int)sv.raw.size(1);
        Tensor dw;
        if (ctx->needs_input_grad(1)) {
            dw = at::empty_strided(sv.w_sizes, sv.w_strides, sv.raw.options().dtype(sv.w_f32 ? at::kFloat : at::kBFloat16));
            const bool w_cl = dw.is_contiguous(at::MemoryFormat::ChannelsLast) && !dw.is_contiguous();
            Tensor dwp = at::empty({(int64_t)Cout * 256}, sv.raw.options().dtype(at::kFloat).memory_format(at::MemoryFormat::Contiguous));
            const double flops = 2.0 * B * (H / 2) * (W / 2) * (double)Cout * 147.0;
            const size_t ws_bytes = epi_stem7x7s2_workspace_bytes(B, H, W, Cout);
            const Tensor s2d = sv.x, dyin = g.dx, dwout = dw;
            const bool w_f32 = sv.w_f32;
            auto launch = [=](epi_stream_t st, Tensor& ws) {
                check(epi_stem7x7s2_bwd_weight(s2d.data_ptr(), dyin.data_ptr(), dwp.data_ptr<float>(), B, H, W, Cout, ws.data_ptr(), (size_t)ws.numel(), st),
                      "epi_stem7x7s2_bwd_weight");
                check(epi_stem7x7s2_unpack_weight_grad(dwp.data_ptr<float>(), Cout, w_cl ? 1 : 0, dwout.data_ptr(), w_f32 ? EPI_F32 : EPI_BF16, st),
                      "epi_stem7x7s2_unpack_weight_grad");
            };
            const bool first_use = first_gradient_of_pass(sv.w);
            if (!first_use) flush_pending_reduces();                  // (the model used twice in one graph: the first gradient must be final before the engine adds this one)
            // The stem is the LAST node of the backward chain: on the second stream its weight gradient queues behind layer 1's grouped launches, which
            // are still running when the chain ends, and the step's tail waits for both one after the other.  On the main stream it runs BESIDE them.
            // EPI_STEM_WGRAD_SIDE=1: the second stream as before (A/B)
            const bool stem_side = false;
            if (stem_side && side_mode() != 0 && first_use && gradient_consumed_after_backward(sv.w)) {
                g_side.jobs.push_back(SideStream::Job{sv.x, g.dx, dw, [=](epi_stream_t st) {
                    launch(st, side_workspace(ws_bytes, s2d));
                    g_side.keep.push_back(dwp);
                    return EpiSlabReduce();
                }, "stem_conv_bwd_weight", flops, 0.0});
            } else {
                ScopedTimer timer("stem_conv_bwd_weight", flops, 0.0, current_stream(sv.raw));
                launch(current_stream(sv.raw), workspace(ws_bytes, sv.raw));
            }
        }
        side_run_jobs();
        holder->stages.clear();
        return {Tensor(), dw, g.dgamma, g.dbeta, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

Tensor stem_conv_bn_act(Tensor x, Tensor w, Tensor gamma, Tensor beta, Tensor running_mean, Tensor running_var, Tensor num_batches, Tensor sums_ws,
                        Tensor bwd_sums, Tensor flags, bool training, double momentum, double eps, bool relu, bool pool) {
    return StemConvBnAct::apply(x, w, gamma, beta, running_mean, running_var, num_batches, sums_ws, bwd_sums, flags, training, momentum, eps, relu, pool);
}

// ---- A whole residual unit (BasicBlock / Bottleneck, pose3d_resnet.py:18-88) as ONE autograd node ------------------------------
// tensors: STAGE_TENSORS per stage, main-path stages first, the downsample projection (if any) last; geometry: (stride, pad) per stage.
// Every main-path stage has ReLU; the last one adds the shortcut before it; the projection has none.  In the backward pass the
// gradient that reaches the unit input through the shortcut (identity: dres of the last stage; projection: its backward-data result)
// is added in the epilogue of the first stage's backward-data GEMM -- autograd's separate accumulation launch per unit disappears.
struct ResidualUnitFn : public torch::autograd::Function<ResidualUnitFn> {
    static Tensor forward(AutogradContext* ctx, Tensor x, at::TensorList tensors, std::vector<int64_t> geometry, bool has_downsample, bool training,
                          double momentum, double eps) {
        const int n_total = (int)(tensors.size() / STAGE_TENSORS), n_main = n_total - (has_downsample ? 1 : 0);
        TORCH_CHECK((int)tensors.size() == n_total * STAGE_TENSORS && (int)geometry.size() == 2 * n_total && n_main >= 2, "residual_unit: arguments");
        auto stage = [&](int i, bool relu) {
            const at::Tensor* t = &tensors[i * STAGE_TENSORS];
            return StageParams{t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8], t[9], geometry[2 * i], geometry[2 * i + 1], relu};
        };
        if (!nhwc_bf16(x)) x = x.to(at::kBFloat16).contiguous(at::MemoryFormat::ChannelsLast);
        auto holder = c10::make_intrusive<SavedHolder>();
        holder->stages.resize(n_total);
        holder->has_downsample = has_downsample;
        holder->n_main = n_main;
        const bool x_grad = x.requires_grad();
        if (training && x_grad) holder->in_link = link_claim(x);
        Tensor out = x;
        // BatchNorm + ReLU of a stage whose consumer is a 1x1 / stride-1 convolution (the Bottleneck's bn2 -> conv3): left to that convolution's
        // launch (BnInput) -- `pending` then travels with the raw tensor
        std::vector<StageParams> sps;
        sps.reserve(n_total);
        for (int i = 0; i < n_total; ++i) sps.push_back(stage(i, i + 1 < n_main || i == n_main - 1));
        sps[n_main - 1].relu = true;
        if (has_downsample) sps[n_total - 1].relu = false;
        BnInput pending;
        for (int i = 0; i + 1 < n_main; ++i) {
            const BnInput* in = pending.sp ? &pending : nullptr;
            BnInput next_pending;
            const int64_t c_out = sps[i].w.size(0);
            if (bn_input_eligible(sps[i + 1], c_out, training)) {
                int done = 0;
                Tensor raw = stage_forward(out, sps[i], Tensor(), training, momentum, eps, i > 0 || x_grad, &holder->stages[i], true, &done, in);
                if (done) {             // the batch sums are in: the next convolution normalises its own input
                    next_pending.sp = &sps[i];
                    next_pending.stats = at::empty({4 * c_out}, sps[i].gamma.options().dtype(at::kFloat));
                    holder->stages[i].stats = next_pending.stats;
                    holder->stages[i].has_res = false;
                    out = raw;
                } else {
                    out = finish_deferred_stage(raw, sps[i], Tensor(), done, training, momentum, eps, &holder->stages[i]);
                }
            } else {
                out = stage_forward(out, sps[i], Tensor(), training, momentum, eps, i > 0 || x_grad, &holder->stages[i], false, nullptr, in);
            }
            pending = next_pending;
        }
        const BnInput* last_in = pending.sp ? &pending : nullptr;
        Tensor shortcut = x, y;
        if (has_downsample && bn_dual_enabled()) {
            // projection shortcut: both convolutions first, then ONE pass normalises both raw outputs, adds them and applies the ReLU -- the
            // projection's BatchNorm output is never written (EPI_BN_DUAL=0: the two separate passes of round 2)
            int done_p = 0, done_m = 0;
            const StageParams pp = stage(n_total - 1, false), pm = stage(n_main - 1, true);
            Tensor raw_p = stage_forward(x, pp, Tensor(), training, momentum, eps, x_grad, &holder->stages[n_total - 1], true, &done_p);
            Tensor raw_m = stage_forward(out, pm, Tensor(), training, momentum, eps, true, &holder->stages[n_main - 1], true, &done_m, last_in);
            if (!training || done_p == done_m) {
                y = dual_bn_forward(raw_m, pm, raw_p, pp, training, training && done_m, momentum, eps, training ? &holder->stages[n_main - 1] : nullptr,
                                    training ? &holder->stages[n_total - 1] : nullptr);
            } else {                    // one producer delivered its batch sums, the other did not: the two separate passes
                auto finish = [&](const Tensor& raw, const StageParams& p, const Tensor& res, int done, StageSaved* sv) {
                    BnBuffers b{p.gamma, p.beta, p.running_mean, p.running_var, p.num_batches, p.sums_ws, p.bwd_sums, p.flags};
                    Tensor stats;
                    if (!done) p.flags.data_ptr<int>()[0] = 0;
                    Tensor out_y = bn_forward(raw, res, b, training, momentum, eps, p.relu, &stats, done != 0);
                    sv->stats = stats;
                    sv->has_res = res.defined();
                    sv->y = (p.relu && res.defined()) ? out_y.detach() : Tensor();
                    return out_y;
                };
                shortcut = finish(raw_p, pp, Tensor(), done_p, &holder->stages[n_total - 1]);
                y = finish(raw_m, pm, shortcut, done_m, &holder->stages[n_main - 1]);
            }
        } else {
            if (has_downsample) shortcut = stage_forward(x, stage(n_total - 1, false), Tensor(), training, momentum, eps, x_grad, &holder->stages[n_total - 1]);
            y = stage_forward(out, stage(n_main - 1, true), shortcut, training, momentum, eps, true, &holder->stages[n_main - 1], false, nullptr, last_in);
        }
        ctx->saved_data["training"] = training;
        if (training) {
            publish_output(holder, y, holder->stages[n_main - 1]);
            ctx->saved_data["holder"] = c10::IValue::make_capsule(holder);
        }
        return y;
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        TORCH_CHECK(ctx->saved_data["training"].toBool(), "residual_unit: backward through inference-mode statistics is not supported");
        auto holder = c10::static_intrusive_pointer_cast<SavedHolder>(ctx->saved_data["holder"].toCapsule());
        TORCH_CHECK(!holder->stages.empty(), "residual_unit: backward called twice (the fused nodes free their activations in backward)");
        g_side.jobs.clear();                         // (left-overs of a pass that aborted inside a node)
        const int n_total = (int)holder->stages.size(), n_main = holder->n_main;
        const bool need_x = ctx->needs_input_grad(0);
        std::vector<StageGrads> g(n_total);
        // input slots: 0 = x, then STAGE_TENSORS per stage (w at +0, gamma at +2, beta at +3)
        auto need_w = [&](int i) { return ctx->needs_input_grad(1 + i * STAGE_TENSORS); };
        // every stage's backward-data launch also does the BatchNorm-backward reduction of the stage in front of it (the first stage's: of
        // the node that produced x) where the library can -- `fed` says whether the next call receives dz with the sums in place
        bool fed = false;
        g[n_main - 1] = stage_backward(grads[0], holder->stages[n_main - 1], true, need_w(n_main - 1), Tensor(), link_state(holder->out_link, grads[0]),
                                       &holder->stages[n_main - 2], c10::intrusive_ptr<BnLink>(), &fed);
        link_retire(holder->out_link);
        Tensor shortcut_grad = g[n_main - 1].dres;                     // gradient of the shortcut input
        int shortcut_step = 1;
        if (holder->has_downsample) {
            // a 1x1 stride-2 projection in front of a 1x1 stride-1 first stage (every stride-2 Bottleneck): the projection's input gradient is zero at
            // three pixels of four -- it stays at half resolution and the first stage's backward-data epilogue adds it at the even pixels
            // (EPI_HALF_SHORTCUT=0: expanded with zeros by the four-phase launch, as in round 2)
            const bool half_ok = true;
            const StageSaved& ds = holder->stages[n_total - 1];
            const StageSaved& s0 = holder->stages[0];
            const bool half = half_ok && need_x && ds.K == 1 && ds.S == 2 && ds.P == 0 && s0.K == 1 && s0.S == 1 && s0.P == 0 && ds.x.defined() &&
                              !(ds.x.size(2) & 1) && !(ds.x.size(3) & 1) &&
                              epi_conv2d_bwd_data_half_addend_ok((int)s0.x.size(0), (int)s0.x.size(2), (int)s0.x.size(3), (int)s0.x.size(1), (int)s0.raw.size(1)) != 0;
            g[n_total - 1] = stage_backward(shortcut_grad, ds, need_x, need_w(n_total - 1), Tensor(), 0, nullptr, c10::intrusive_ptr<BnLink>(), nullptr, 1, half);
            shortcut_grad = g[n_total - 1].dx;                          // undefined when x needs no gradient
            if (half) shortcut_step = 2;
        }
        Tensor flow = g[n_main - 1].dx;
        for (int i = n_main - 2; i >= 1; --i) {
            const bool was_fed = fed;
            g[i] = stage_backward(flow, holder->stages[i], true, need_w(i), Tensor(), was_fed ? 1 : 0, &holder->stages[i - 1], c10::intrusive_ptr<BnLink>(), &fed);
            flow = g[i].dx;
        }
        {
            const bool was_fed = fed;
            g[0] = stage_backward(flow, holder->stages[0], need_x, need_w(0), need_x ? shortcut_grad : Tensor(), was_fed ? 1 : 0, nullptr, holder->in_link, &fed,
                                  shortcut_step);
        }
        side_run_jobs();                                              // the unit's weight gradients: second stream, one fork event
        // grouped weight gradients leave when the stage is complete (its first unit carries the downsample projection), when one more
        // unit would not fit into a launch, or per unit (EPI_WGRAD_GROUP=1)
        // (EPI_WGRAD_GROUP_ROWS = N: also flush a unit by itself when it reduces over >= N rows.  Measured with N = 65 536 -- layer 1 at the bench
        //  shape, whose stage group otherwise leaves at the very end of the backward chain: 6.606 vs 6.600 ms/step, 7 launches instead of 5; the
        //  earlier launches take from the main stream what the shorter tail gives back.  Off.)
        const long long unit_rows = 0;
        const StageSaved& s0 = holder->stages.empty() ? StageSaved() : holder->stages[0];
        const long long rows = s0.x.defined() ? (long long)s0.x.size(0) * s0.x.size(2) * s0.x.size(3) : 0;
        if (group_mode() == 1 || holder->has_downsample || g_side.group.size() + 4 > (size_t)epi_wgrad_group_max() ||
            (unit_rows > 0 && rows >= unit_rows))
            side_group_flush();
        variable_list out;
        out.reserve(1 + n_total * STAGE_TENSORS + 5);
        out.push_back(g[0].dx);
        for (int i = 0; i < n_total; ++i) {
            out.push_back(g[i].dw);
            out.push_back(Tensor());
            out.push_back(g[i].dgamma);
            out.push_back(g[i].dbeta);
            for (int k = 4; k < STAGE_TENSORS; ++k) out.push_back(Tensor());
        }
        for (int k = 0; k < 5; ++k) out.push_back(Tensor());          // geometry, has_downsample, training, momentum, eps
        holder->stages.clear();                                       // release the saved activations now
        holder->in_link.reset();
        holder->out_link.reset();
        return out;
    }
};

Tensor residual_unit(Tensor x, std::vector<Tensor> tensors, std::vector<int64_t> geometry, bool has_downsample, bool training, double momentum,
                     double eps) {
    return ResidualUnitFn::apply(x, at::TensorList(tensors), geometry, has_downsample, training, momentum, eps);
}

// ---- Deconvolution head (pose3d_resnet.py:158-183, 116-122) ---------------------------------------------------------------------
// ConvTranspose2d(k4, s2, p1, no bias) -> BatchNorm -> ReLU as ONE autograd node, and the final 1x1 convolution (+ bias) as one node.
// Same kernels as the Python autograd.Functions they replace (models/fused.py round 1): what changes is the host cost per call and
// that their weight gradients -- the largest of the network: 0.35 ms per step -- join the backbone's on the second stream.
// w: [Cin, Cout, 4, 4] bf16 training copy or fp32 master, channels_last memory ([Cin][kh][kw][Cout] = the backward-data operand as it
// stands); w_phase: the packed forward operand [4][Cout][4 Cin] kept by the optimizer, or undefined / empty -> packed here.
struct DeconvBnAct : public torch::autograd::Function<DeconvBnAct> {
    static Tensor forward(AutogradContext* ctx, Tensor x, Tensor w, c10::optional<Tensor> w_phase_opt, Tensor gamma, Tensor beta, Tensor running_mean,
                          Tensor running_var, Tensor num_batches, Tensor sums_ws, Tensor bwd_sums, Tensor flags, bool training, double momentum,
                          double eps, bool relu) {
        TORCH_CHECK(x.is_cuda() && w.is_cuda(), "deconv_bn_act: tensors must live on the GPU (no CPU fallback in epipolarpose_amd)");
        if (!nhwc_bf16(x)) x = x.to(at::kBFloat16).contiguous(at::MemoryFormat::ChannelsLast);
        TORCH_CHECK(w.dim() == 4 && w.size(0) == x.size(1) && w.size(2) == 4 && w.size(3) == 4, "deconv_bn_act: weight must be [Cin, Cout, 4, 4]");
        const int B = (int)x.size(0), Cin = (int)x.size(1), H = (int)x.size(2), W = (int)x.size(3), Cout = (int)w.size(1);
        Tensor w16 = w.detach();
        if (w16.scalar_type() != at::kBFloat16) w16 = w16.to(at::kBFloat16);
        if (!w16.is_contiguous(at::MemoryFormat::ChannelsLast)) w16 = w16.contiguous(at::MemoryFormat::ChannelsLast);
        Tensor w_phase = (w_phase_opt.has_value() && w_phase_opt->defined() && w_phase_opt->numel() == w.numel()) ? *w_phase_opt : Tensor();
        if (!w_phase.defined()) {
            w_phase = at::empty({4, Cout, 4 * Cin}, w16.options().memory_format(at::MemoryFormat::Contiguous));
            check(epi_deconv4x4s2_pack_phase_cl(w16.data_ptr(), Cin, Cout, w_phase.data_ptr(), current_stream(x)), "epi_deconv4x4s2_pack_phase_cl");
        }
        Tensor raw = at::empty({B, Cout, 2 * H, 2 * W}, x.options().memory_format(at::MemoryFormat::ChannelsLast));
        const double flops = 2.0 * B * H * W * 16.0 * Cin * Cout;
        int sums_done = 0;
        TORCH_CHECK(!training || sums_ws.numel() == 2 * (int64_t)Cout * epi_bn_sum_copies(Cout), "deconv_bn_act: sums_ws must be [epi_bn_sum_copies(C)][2C]");
        if (training) {             // the accumulator hand-over of bn_forward, done here because the GEMM epilogue may fill sums_ws
            int* fl = flags.data_ptr<int>();
            if (fl[0]) sums_ws.zero_();
            fl[0] = 1;
            fl[1] = 0;
        }
        {
            Tensor& ws = workspace(epi_gemm_workspace_bytes(B * H * W, Cout, 4 * Cin, 4), x);
            ScopedTimer timer("head_deconv4x4s2_fwd", flops, 0.0, current_stream(x));
            check(epi_deconv4x4s2_fwd_stats(x.data_ptr(), w_phase.data_ptr(), raw.data_ptr(), B, H, W, Cin, Cout,
                                            training ? sums_ws.data_ptr<float>() : nullptr, training ? &sums_done : nullptr, ws.data_ptr(),
                                            (size_t)ws.numel(), current_stream(x)), "epi_deconv4x4s2_fwd");
        }
        BnBuffers b{gamma, beta, running_mean, running_var, num_batches, sums_ws, bwd_sums, flags};
        Tensor stats;
        Tensor y;
        if (training && !sums_done) {       // the statistics pass runs separately; re-arm the flag protocol for bn_forward
            flags.data_ptr<int>()[0] = 0;
            y = bn_forward(raw, Tensor(), b, training, momentum, eps, relu, &stats, false);
        } else {
            y = bn_forward(raw, Tensor(), b, training, momentum, eps, relu, &stats, training);
        }
        ctx->saved_data["training"] = training;
        if (training) {
            auto holder = c10::make_intrusive<SavedHolder>();
            holder->stages.resize(1);
            StageSaved& sv = holder->stages[0];
            sv.x = x; sv.raw = raw; sv.stats = stats; sv.gamma = gamma; sv.wb = w16; sv.sums_ws = sums_ws; sv.bwd_sums = bwd_sums; sv.flags = flags;
            sv.relu = relu; sv.has_res = false; sv.w_f32 = w.scalar_type() != at::kBFloat16; sv.w = w; sv.need_dx = x.requires_grad();
            if (x.requires_grad()) holder->in_link = link_claim(x);
            publish_output(holder, y, sv);
            ctx->saved_data["holder"] = c10::IValue::make_capsule(holder);
        }
        return y;
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        TORCH_CHECK(ctx->saved_data["training"].toBool(), "deconv_bn_act: backward through inference-mode statistics is not supported");
        auto holder = c10::static_intrusive_pointer_cast<SavedHolder>(ctx->saved_data["holder"].toCapsule());
        TORCH_CHECK(!holder->stages.empty(), "deconv_bn_act: backward called twice (the fused nodes free their activations in backward)");
        g_side.jobs.clear();
        const StageSaved& sv = holder->stages[0];
        BnGrads g = bn_backward(grads[0], sv.raw, Tensor(), sv.stats, sv.gamma, sv.sums_ws, sv.bwd_sums, sv.flags, sv.relu, false,
                                link_state(holder->out_link, grads[0]));
        link_retire(holder->out_link);
        const Tensor& x = sv.x;
        const int B = (int)x.size(0), Cin = (int)x.size(1), H = (int)x.size(2), W = (int)x.size(3), Cout = (int)sv.raw.size(1);
        const double flops = 2.0 * B * H * W * 16.0 * Cin * Cout;
        Tensor dx, dw;
        if (ctx->needs_input_grad(0)) {
            dx = at::empty_like(x);
            Tensor& ws = workspace(epi_gemm_workspace_bytes(B * H * W, Cin, 16 * Cout, 1), x);
            ScopedTimer timer("head_deconv4x4s2_bwd_data", flops, 0.0, current_stream(x));
            EpiBnReduce red = {};
            int red_done = 0;
            if (link_reduce_args(holder->in_link, &red))       // the BatchNorm-backward reduction of the layer that produced x, in this epilogue
                check(epi_deconv4x4s2_bwd_data_bnred(g.dx.data_ptr(), sv.wb.data_ptr(), dx.data_ptr(), B, H, W, Cin, Cout, &red, &red_done, ws.data_ptr(),
                                                     (size_t)ws.numel(), current_stream(x)), "epi_deconv4x4s2_bwd_data_bnred");
            else
                check(epi_deconv4x4s2_bwd_data(g.dx.data_ptr(), sv.wb.data_ptr(), dx.data_ptr(), B, H, W, Cin, Cout, ws.data_ptr(), (size_t)ws.numel(),
                                               current_stream(x)), "epi_deconv4x4s2_bwd_data");
            if (red_done) link_mark_reduced(holder->in_link, dx);
        }
        if (ctx->needs_input_grad(1)) {
            // [Cin, Cout, 4, 4] in channels_last strides = memory [Cin][16 taps][Cout]: the kernel's own output order
            dw = at::empty_strided({Cin, Cout, 4, 4}, {16 * (int64_t)Cout, 1, 4 * (int64_t)Cout, (int64_t)Cout},
                                   x.options().dtype(sv.w_f32 ? at::kFloat : at::kBFloat16));
            const size_t slab_bytes = epi_gemm_tn_workspace_bytes(B * H * W, Cin, Cout, 16);
            // (a layer used twice in one graph: the engine adds the second gradient to the first on the main stream, so the first must be final
            //  by then -- everything pending is joined and this one stays on the main stream, as in stage_backward)
            const bool first_use = first_gradient_of_pass(sv.w);
            if (!first_use) flush_pending_reduces();
            const bool on_side = side_mode() != 0 && first_use && gradient_consumed_after_backward(sv.w);
            const Tensor xin = x, dyin = g.dx, dwout = dw;
            const bool w_f32 = sv.w_f32;
            auto launch = [=](epi_stream_t st, Tensor& ws) {
                check(epi_deconv4x4s2_bwd_weight(xin.data_ptr(), dyin.data_ptr(), dwout.data_ptr(), w_f32 ? EPI_F32 : EPI_BF16, B, H, W, Cin, Cout,
                                                 ws.data_ptr(), (size_t)ws.numel(), st), "epi_deconv4x4s2_bwd_weight");
            };
            if (on_side) {
                g_side.jobs.push_back(SideStream::Job{x, g.dx, dw, [=](epi_stream_t st) {
                    launch(st, side_workspace(slab_bytes, xin));
                    return EpiSlabReduce();
                }, "head_deconv4x4s2_bwd_weight", flops, 0.0});
            } else {
                ScopedTimer timer("head_deconv4x4s2_bwd_weight", flops, 0.0, current_stream(x));
                launch(current_stream(x), workspace(slab_bytes, x));
            }
        }
        side_run_jobs();
        holder->stages.clear();
        holder->in_link.reset();
        holder->out_link.reset();
        return {dx, dw, Tensor(), g.dgamma, g.dbeta, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

Tensor deconv_bn_act(Tensor x, Tensor w, c10::optional<Tensor> w_phase, Tensor gamma, Tensor beta, Tensor running_mean, Tensor running_var,
                     Tensor num_batches, Tensor sums_ws, Tensor bwd_sums, Tensor flags, bool training, double momentum, double eps, bool relu) {
    return DeconvBnAct::apply(x, w, w_phase, gamma, beta, running_mean, running_var, num_batches, sums_ws, bwd_sums, flags, training, momentum, eps, relu);
}

// Final 1x1 convolution with bias: y[m][co] = sum_ci x[m][ci] w[co][ci] + b[co] on the [B*H*W, C] views of the NHWC tensors.
// w: [Cout, Cin, 1, 1] bf16 training copy or fp32 master; bias fp32 or undefined.
struct Conv1x1Bias : public torch::autograd::Function<Conv1x1Bias> {
    static Tensor forward(AutogradContext* ctx, Tensor x, Tensor w, c10::optional<Tensor> bias_opt) {
        TORCH_CHECK(x.is_cuda() && w.is_cuda(), "conv1x1: tensors must live on the GPU (no CPU fallback in epipolarpose_amd)");
        if (!nhwc_bf16(x)) x = x.to(at::kBFloat16).contiguous(at::MemoryFormat::ChannelsLast);
        TORCH_CHECK(w.dim() == 4 && w.size(1) == x.size(1) && w.size(2) == 1 && w.size(3) == 1, "conv1x1: weight must be [Cout, Cin, 1, 1]");
        const int B = (int)x.size(0), Cin = (int)x.size(1), H = (int)x.size(2), W = (int)x.size(3), Cout = (int)w.size(0);
        const int M = B * H * W;
        Tensor w16 = w.detach().reshape({Cout, Cin});
        if (w16.scalar_type() != at::kBFloat16) w16 = w16.to(at::kBFloat16);
        if (!w16.is_contiguous()) w16 = w16.contiguous();
        const bool has_bias = bias_opt.has_value() && bias_opt->defined();
        Tensor bias;
        if (has_bias) bias = bias_opt->detach().to(at::kFloat).contiguous();
        Tensor y = at::empty({B, Cout, H, W}, x.options().memory_format(at::MemoryFormat::ChannelsLast));
        {
            Tensor& ws = workspace(epi_gemm_workspace_bytes(M, Cout, Cin, 1), x);
            ScopedTimer timer("head_gemm_bf16", 2.0 * M * (double)Cout * Cin, 0.0, current_stream(x));
            check(epi_gemm_bf16(x.data_ptr(), Cin, w16.data_ptr(), Cin, y.data_ptr(), Cout, EPI_BF16, M, Cout, Cin, has_bias ? bias.data_ptr<float>() : nullptr,
                                ws.data_ptr(), (size_t)ws.numel(), current_stream(x)), "epi_gemm_bf16");
        }
        ctx->saved_data["has_bias"] = has_bias;
        ctx->saved_data["w_f32"] = w.scalar_type() != at::kBFloat16;
        if (x.requires_grad()) {
            auto link = link_claim(x);
            if (link) ctx->saved_data["in_link"] = c10::IValue::make_capsule(link);
        }
        ctx->save_for_backward({x, w16, w, has_bias ? *bias_opt : Tensor()});
        return y;
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        const auto saved = ctx->get_saved_variables();
        const Tensor x = saved[0], w16 = saved[1], w = saved[2];
        Tensor dy = grads[0];
        const int B = (int)x.size(0), Cin = (int)x.size(1), H = (int)x.size(2), W = (int)x.size(3), Cout = (int)w16.size(0);
        // the per-channel sums of this gradient, when the kernel that wrote it delivered them (only for the tensor as written: no conversion below)
        Tensor offered_sums = take_column_sums(dy, Cout);                          // (consumes or drops whatever offer is pending)
        if (offered_sums.defined() && !(nhwc_bf16(dy) && dy.dim() == 4 && dy.size(1) == Cout)) offered_sums = Tensor();
        if (!nhwc_bf16(dy)) dy = dy.to(at::kBFloat16).contiguous(at::MemoryFormat::ChannelsLast);
        const int M = B * H * W;
        const bool w_f32 = ctx->saved_data["w_f32"].toBool();
        const double flops = 2.0 * M * (double)Cout * Cin;
        g_side.jobs.clear();
        Tensor dx, dw, db;
        if (ctx->needs_input_grad(0)) {
            const Tensor wt = w16.t().contiguous();                  // [Cin][Cout]: the backward-data operand (0.5 MB)
            dx = at::empty_like(x);
            Tensor& ws = workspace(epi_gemm_workspace_bytes(M, Cin, Cout, 1), x);
            ScopedTimer timer("head_gemm_bf16", flops, 0.0, current_stream(x));
            c10::intrusive_ptr<BnLink> link;
            if (ctx->saved_data.count("in_link")) link = c10::static_intrusive_pointer_cast<BnLink>(ctx->saved_data["in_link"].toCapsule());
            EpiBnReduce red = {};
            int red_done = 0;
            if (link_reduce_args(link, &red))                  // the last deconvolution's BatchNorm-backward reduction, in this epilogue
                check(epi_gemm_bf16_bnred(dy.data_ptr(), Cout, wt.data_ptr(), Cout, dx.data_ptr(), Cin, M, Cin, Cout, &red, &red_done, ws.data_ptr(),
                                          (size_t)ws.numel(), current_stream(x)), "epi_gemm_bf16_bnred");
            else
                check(epi_gemm_bf16(dy.data_ptr(), Cout, wt.data_ptr(), Cout, dx.data_ptr(), Cin, EPI_BF16, M, Cin, Cout, nullptr, ws.data_ptr(),
                                    (size_t)ws.numel(), current_stream(x)), "epi_gemm_bf16");
            if (red_done) link_mark_reduced(link, dx);
        }
        if (ctx->saved_data["has_bias"].toBool() && ctx->needs_input_grad(2)) {
            // bias gradient = column sums of dy (one more read of the 285 MB logits gradient): second stream as well when nobody reads it early --
            // unless the criterion's backward kernel delivered them with the gradient (take_column_sums): then they are complete on the main stream
            const Tensor bias_leaf = saved[3];
            const bool first_bias = bias_leaf.defined() && first_gradient_of_pass(bias_leaf);
            if (bias_leaf.defined() && !first_bias) flush_pending_reduces();          // (second use of the layer in this graph, see the weight below)
            Tensor sums;
            if (offered_sums.defined()) {
                db = offered_sums;
                g_bias_sums_taken += 1;
            } else {
                sums = at::zeros({2 * (int64_t)Cout}, x.options().dtype(at::kFloat));
                db = sums.slice(0, 0, Cout);
            }
            if (offered_sums.defined()) {
            } else if (side_mode() != 0 && first_bias && gradient_consumed_after_backward(bias_leaf)) {
                const Tensor dyin = dy, out = sums;
                g_side.jobs.push_back(SideStream::Job{dy, sums, db, [=](epi_stream_t st) {
                    check(epi_column_sums_bf16(dyin.data_ptr(), (long long)M, Cout, out.data_ptr<float>(), st), "epi_column_sums_bf16");
                    return EpiSlabReduce();
                }, "bias_column_sums", 0.0, 2.0 * (double)M * Cout});
            } else {
                check(epi_column_sums_bf16(dy.data_ptr(), (long long)M, Cout, sums.data_ptr<float>(), current_stream(x)), "epi_column_sums_bf16");
            }
        }
        if (ctx->needs_input_grad(1)) {
            // the weight gradient of a 1x1 convolution: dW[co][ci] = sum_m dy[m][co] x[m][ci] (bf16 or fp32 result, split sums deferrable)
            dw = at::empty({Cout, Cin, 1, 1}, x.options().dtype(w_f32 ? at::kFloat : at::kBFloat16).memory_format(at::MemoryFormat::Contiguous));
            const size_t slab_bytes = epi_gemm_tn_workspace_bytes(M, Cout, Cin, 1);
            // a second use of this layer in one graph (the model called on two inputs before one backward): w.grad() is still undefined, yet the
            // engine adds this gradient to the first one on the main stream -- join what is pending, no second stream, no deferred sum
            const bool first_use = first_gradient_of_pass(w);
            if (!first_use) flush_pending_reduces();
            const bool after_pass = first_use && gradient_consumed_after_backward(w);
            const bool may_defer = slab_bytes && defer_enabled() && after_pass;
            void* slabs = may_defer ? pending_slab_alloc(slab_bytes, x) : nullptr;
            const bool on_side = side_mode() != 0 && after_pass && (slab_bytes == 0 || slabs != nullptr);
            const Tensor xin = x, dyin = dy, dwout = dw;
            auto launch = [=](epi_stream_t st, Tensor* ws) {
                EpiSlabReduce pend = {};
                check(epi_conv2d_bwd_weight_deferred(xin.data_ptr(), dyin.data_ptr(), dwout.data_ptr(), w_f32 ? EPI_F32 : EPI_BF16, B, H, W, Cin, Cout, 1, 1, 1, 0,
                                                     slabs ? slabs : ws->data_ptr(), slabs ? slab_bytes : (size_t)ws->numel(), slabs ? &pend : nullptr, st),
                      "epi_conv2d_bwd_weight");
                return pend;
            };
            if (on_side) {
                g_side.jobs.push_back(SideStream::Job{x, dy, dw, [=](epi_stream_t st) { return launch(st, slabs ? nullptr : &side_workspace(slab_bytes, xin)); },
                                                      "head_gemm_tn_bf16", flops, 0.0});
            } else {
                EpiSlabReduce pend = {};
                {
                    ScopedTimer timer("head_gemm_tn_bf16", flops, 0.0, current_stream(x));
                    pend = launch(current_stream(x), slabs ? nullptr : &workspace(slab_bytes, x));
                }
                if (pend.nsplit > 0) pending_register(pend, dw);
            }
        }
        side_run_jobs();
        return {dx, dw, db};
    }
};

Tensor conv1x1_bias(Tensor x, Tensor w, c10::optional<Tensor> bias) { return Conv1x1Bias::apply(x, w, bias); }

// ---- MaxPool2d(3, 2, 1) of the stem (pose3d_resnet.py:104,186) on epi_maxpool3x3s2_* --------------------------------------------
struct MaxPool3x3s2 : public torch::autograd::Function<MaxPool3x3s2> {
    static Tensor forward(AutogradContext* ctx, Tensor x) {
        TORCH_CHECK(x.is_cuda(), "maxpool3x3s2: input must live on the GPU (no CPU fallback in epipolarpose_amd)");
        if (!nhwc_bf16(x)) x = x.to(at::kBFloat16).contiguous(at::MemoryFormat::ChannelsLast);
        const int64_t B = x.size(0), C = x.size(1), H = x.size(2), W = x.size(3);
        const int64_t Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
        Tensor y = at::empty({B, C, Ho, Wo}, x.options().memory_format(at::MemoryFormat::ChannelsLast));
        Tensor pos = at::empty({B, Ho, Wo, C}, x.options().dtype(at::kByte).memory_format(at::MemoryFormat::Contiguous));
        {
            ScopedTimer timer("maxpool_fwd", 0.0, 2.0 * (double)x.numel() + 3.0 * (double)y.numel(), current_stream(x));
            check(epi_maxpool3x3s2_fwd(x.data_ptr(), y.data_ptr(), pos.data_ptr(), (int)B, (int)H, (int)W, (int)C, current_stream(x)), "epi_maxpool3x3s2_fwd");
        }
        ctx->saved_data["H"] = H;
        ctx->saved_data["W"] = W;
        ctx->save_for_backward({pos});
        return y;
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        const Tensor pos = ctx->get_saved_variables()[0];
        Tensor dy = grads[0];
        if (!nhwc_bf16(dy)) dy = dy.to(at::kBFloat16).contiguous(at::MemoryFormat::ChannelsLast);
        const int64_t B = dy.size(0), C = dy.size(1), H = ctx->saved_data["H"].toInt(), W = ctx->saved_data["W"].toInt();
        Tensor dx = at::empty({B, C, H, W}, dy.options().memory_format(at::MemoryFormat::ChannelsLast));
        ScopedTimer timer("maxpool_bwd", 0.0, 3.0 * (double)dy.numel() + 2.0 * (double)dx.numel(), current_stream(dy));
        check(epi_maxpool3x3s2_bwd(dy.data_ptr(), pos.data_ptr(), dx.data_ptr(), (int)B, (int)H, (int)W, (int)C, current_stream(dy)), "epi_maxpool3x3s2_bwd");
        return {dx};
    }
};
Tensor maxpool3x3s2(Tensor x) { return MaxPool3x3s2::apply(x); }

// ---- FusedAdam's per-step pointer table ----------------------------------------------------------------------------
// Row layout (int64 slots): p, g, m, v, shadow, n, flags (bit 0: gradient is bf16).  For every parameter: take the gradient of its
// bf16 training copy (when it has one) or its own, check dtype and memory order against the parameter (same strides on every
// dimension of extent > 1; otherwise the gradient is re-laid-out into a temporary returned to the caller, who keeps it alive until
// the Adam kernel has been enqueued), and write parameter / shadow / gradient addresses into the pinned host table.  Returns
// (changed, temporaries): `changed` tells the caller to upload the table again.
std::tuple<bool, std::vector<Tensor>> adam_prepare(const std::vector<Tensor>& params, const std::vector<c10::optional<Tensor>>& copies,
                                                  Tensor table_host, int64_t row, const std::vector<int64_t>& only) {
    TORCH_CHECK(params.size() == copies.size() && table_host.scalar_type() == at::kLong && table_host.numel() >= (int64_t)params.size() * row,
                "adam_prepare: table size");
    int64_t* t = table_host.data_ptr<int64_t>();
    bool changed = false;
    std::vector<Tensor> keep;
    // `only`: the parameter indices to refresh (a partial step, optim.FusedAdam.enable_step_in_backward); empty = all
    const size_t count = only.empty() ? params.size() : only.size();
    for (size_t k = 0; k < count; ++k) {
        const size_t i = only.empty() ? k : (size_t)only[k];
        TORCH_CHECK(i < params.size(), "adam_prepare: parameter index");
        const Tensor& p = params[i];
        const bool has_copy = copies[i].has_value() && copies[i]->defined();
        Tensor g = has_copy ? copies[i]->grad() : p.grad();
        TORCH_CHECK(g.defined(), "FusedAdam: parameter ", i, " received no gradient (partial updates are not supported)");
        TORCH_CHECK(g.scalar_type() == at::kFloat || g.scalar_type() == at::kBFloat16, "FusedAdam: gradient dtype not supported");
        bool same = g.sizes() == p.sizes();
        if (same)
            for (int64_t d = 0; d < p.dim(); ++d)
                if (p.size(d) > 1 && g.stride(d) != p.stride(d)) { same = false; break; }
        if (!same) {                                   // e.g. an NCHW-strided gradient for a channels_last weight
            Tensor g2 = at::empty_strided(p.sizes(), p.strides(), g.options());
            g2.copy_(g);
            keep.push_back(g2);
            g = g2;
        }
        const int64_t gp = reinterpret_cast<int64_t>(g.data_ptr()), pp = reinterpret_cast<int64_t>(p.data_ptr());
        const int64_t sh = has_copy ? reinterpret_cast<int64_t>(copies[i]->data_ptr()) : 0;
        const int64_t fl = g.scalar_type() == at::kBFloat16 ? 1 : 0;
        int64_t* r = t + (int64_t)i * row;
        if (r[0] != pp || r[1] != gp || r[4] != sh || r[6] != fl) {
            r[0] = pp; r[1] = gp; r[4] = sh; r[6] = fl;
            changed = true;
        }
    }
    return std::make_tuple(changed, keep);
}

// distributed.BucketedGradSync._launch without its Python loop (~170 parameters per step: 0.3 ms of host time inside finish(), during
// which the GPU had nothing queued): move every gradient of a bucket into its slice of the flat buffer with ONE multi-tensor copy and
// re-point .grad at the slice.  views[i]: the parameter-strided view of params[i] inside the flat buffer.
void pack_bucket(const std::vector<Tensor>& params, const std::vector<Tensor>& views) {
    TORCH_CHECK(params.size() == views.size(), "pack_bucket: list sizes");
    std::vector<Tensor> dst, src;
    dst.reserve(params.size());
    src.reserve(params.size());
    for (size_t i = 0; i < params.size(); ++i) {
        Tensor g = params[i].grad();
        const Tensor& v = views[i];
        if (!g.defined()) {
            v.zero_();                                   // parameter without a gradient this step
        } else if (g.data_ptr() != v.data_ptr()) {
            bool fast = g.strides() == v.strides();
            if (!fast && g.sizes() == v.sizes()) {
                // same memory order, only the strides of extent-1 dimensions differ (a 1x1 convolution weight's gradient comes back
                // "contiguous", the parameter is "channels_last"): re-stride the alias so that the multi-tensor copy keeps its fast path
                bool same_order = true;
                for (int64_t d = 0; d < g.dim(); ++d)
                    if (g.size(d) > 1 && g.stride(d) != v.stride(d)) { same_order = false; break; }
                if (same_order) { g = g.as_strided(g.sizes(), v.strides(), g.storage_offset()); fast = true; }
            }
            if (fast) { dst.push_back(v); src.push_back(g); }
            else v.copy_(g);                             // genuinely different layout: its own strided copy
        }
    }
    if (!dst.empty()) at::_foreach_copy_(dst, src);
    for (size_t i = 0; i < params.size(); ++i) params[i].mutable_grad() = views[i];
}

// optimizer.zero_grad(set_to_none=True) for a parameter list without a Python loop (~170 parameters + their bf16 training copies)
void clear_grads(const std::vector<Tensor>& tensors) {
    for (const Tensor& t : tensors)
        if (t.defined()) t.mutable_grad().reset();
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "torch-autograd glue over the libepipolar_hip C ABI (no compute of its own)";
    m.def("bn_act", &bn_act, "fused BatchNorm (+residual) (+ReLU), NHWC bf16, autograd-aware");
    m.def("conv_bn_act", &conv_bn_act, "Conv2d -> BatchNorm (+residual) (+ReLU) as one autograd node, NHWC bf16");
    m.def("stem_conv_bn_act", &stem_conv_bn_act, "Conv2d(3, C, 7, stride 2, pad 3) -> BatchNorm -> ReLU of the stem as one autograd node (space-to-depth implicit GEMM)");
    m.def("deconv_bn_act", &deconv_bn_act, "ConvTranspose2d(k4, s2, p1) -> BatchNorm -> ReLU as one autograd node, NHWC bf16");
    m.def("conv1x1_bias", &conv1x1_bias, "1x1 convolution (+ bias) on the NHWC view as one autograd node");
    m.def("maxpool3x3s2", &maxpool3x3s2, "MaxPool2d(kernel 3, stride 2, padding 1), NHWC bf16, autograd-aware");
    m.def("residual_unit", &residual_unit, "a whole BasicBlock / Bottleneck (conv-bn-relu stages + shortcut) as one autograd node");
    m.def("begin_early_step", &begin_early_step,
          "inside a backward pass: sum the pending weight-gradient slabs on the second stream and return that stream (0: second stream off)");
    m.def("flush_pending_async", &flush_pending_async,
          "enqueue the pending grouped weight gradients and slab sums on the second stream without joining; returns a ticket (-1: already final)");
    m.def("wait_flush_ticket", &wait_flush_ticket, "the main stream of the device waits for the event behind a flush_pending_async ticket");
    m.def("flush_pending_reduces", &flush_pending_reduces,
          "sum the weight-gradient slabs parked by this backward pass now (the engine's final callback does it at the end of backward())");
    m.def("bn_in_fuse_mode", &bn_in_fuse_mode, "BatchNorm + ReLU of a bottleneck interior inside the consuming 1x1 convolution's launch (1) or as its own pass (0, default); "
          "returns the previous setting, a negative mode only queries");
    m.def("bn_in_fuse_count", &bn_in_fuse_count, "convolutions that normalised their own input since the last reset");
    m.def("wgrad_stream_mode", &wgrad_stream_mode,
          "weight gradients on a second HIP stream: 0 off, 1 on, 2 on with the lowest stream priority; returns the previous setting");
    m.def("declare_hook_free", &declare_hook_free,
          "parameters whose post-accumulate hooks were all removed again (torch keeps the empty hook object): their gradients may stay on the "
          "second stream / in a grouped launch / unreduced until the end of the pass; replaces the previous declaration");
    m.def("bn_dual_mode", &bn_dual_mode,
          "projection shortcut: the unit's last BatchNorm and the projection's BatchNorm in one pass (1, default; EPI_BN_DUAL=0 turns it off) or "
          "two passes (0); returns the previous setting, a negative argument only queries");
    m.def("bn_bwd_fuse_counts", &bn_bwd_fuse_counts, "(fused, refused) BatchNorm backward passes since the last reset", py::arg("reset") = false);
    m.def("bn_bwd_fuse_mode", &bn_bwd_fuse_mode,
          "BatchNorm-backward reduction fused into the backward-data GEMM that produces the layer's gradient (EpiBnReduce): 1 on (default; "
          "EPI_BN_BWD_FUSE=0 turns it off), 0 off; returns the previous setting, a negative argument only queries");
    m.def("column_sums_wanted", &column_sums_wanted, "whether a gradient producer should deliver per-channel sums with its gradient (off in deterministic mode / EPI_BIAS_GRAD_FUSE=0)");
    m.def("offer_column_sums", &offer_column_sums, "per-channel sums (float32 [C]) of the gradient tensor the caller is about to return from its backward");
    m.def("bias_grad_fuse_mode", &bias_grad_fuse_mode, "the final layer's bias gradient from the criterion's backward kernel: 0 off (default, measured slower in the step), 1 on; returns the previous setting");
    m.def("bias_sums_taken", [](bool reset) { const long long v = g_bias_sums_taken; if (reset) g_bias_sums_taken = 0; return v; },
          "test hook: bias gradients that came from an offered column-sum tensor since the last reset");
    m.def("wgrad_group_mode", &wgrad_group_mode,
          "grouped weight-gradient launches: 0 one launch per layer, 1 one per autograd node, 2 one per ResNet stage; returns the previous setting");
    m.def("defer_wgrad_reduce", &defer_wgrad_reduce, "enable / disable the deferred weight-gradient reduction; returns the previous setting");
    m.def("pack_bucket", &pack_bucket, "gradient bucket packing of distributed.BucketedGradSync in one call (multi-tensor copy + .grad re-pointing)");
    m.def("clear_grads", &clear_grads, "drop the .grad of every tensor in the list (zero_grad(set_to_none=True))");
    m.def("adam_prepare", &adam_prepare, "FusedAdam pointer table refresh (no Python loop over the parameters)");
    m.def("timing_enable", &timing_enable, "record HIP events around every epi_* launch made by this extension");
    m.def("timing_collect", &timing_collect, "{name: (launches, total ms, algorithmic FLOPs, algorithmic bytes)}; clears the records");
    m.def("abi_version", []() { return epi_version(); });
}
