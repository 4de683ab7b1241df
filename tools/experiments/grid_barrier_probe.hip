// What does a device-wide hand-over INSIDE a kernel cost on MI355X (8 XCDs, L2s not coherent with each other)?  The number that prices every design
// which replaces launch boundaries by grid barriers (persistent kernels that span layers, stream-K fix-ups).
//   barrier:   G workgroups (one per CU and multiples), K rounds of: arrive on an agent-scope counter, spin until everybody has arrived
//   hand-over: the same, and around each barrier every workgroup writes BYTES with agent-scope (write-through) stores and afterwards reads the BYTES its
//              neighbour (+ G/2, i.e. another XCD) wrote, with agent-scope (L2-bypassing) loads -- the cheapest correct way to pass data between XCDs
//   launches:  K empty launches of G workgroups on one stream, for comparison (the kernel boundary)
// Every spin is bounded: a workgroup that waits longer than ~20 ms sets an error flag and leaves (no hang if the grid is not co-resident).
// hipcc --offload-arch=gfx950 -O3 -o tools/grid_barrier_probe_bin tools/grid_barrier_probe.hip ; tools/grid_barrier_probe_bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ bool grid_arrive_and_wait(int* counter, int target, int* error) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long long polls = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++polls > 4000000) { __hip_atomic_store(error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; break; }
        }
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(256) void barrier_kernel(int* counter, int rounds, int* error) {
    for (int k = 0; k < rounds; ++k)
        if (!grid_arrive_and_wait(counter, (k + 1) * (int)gridDim.x, error)) return;
}

// words: 8-byte words per thread and round
__global__ __launch_bounds__(256) void handover_kernel(int* counter, int rounds, int* error, unsigned long long* buf, int words, unsigned long long* sink) {
    const int g = gridDim.x, me = blockIdx.x, other = (me + g / 2) % g;
    unsigned long long acc = 0;
    for (int k = 0; k < rounds; ++k) {
        unsigned long long* mine = buf + ((size_t)(k & 1) * g + me) * 256 * words;
        for (int w = 0; w < words; ++w)
            __hip_atomic_store(mine + (size_t)w * 256 + threadIdx.x, (unsigned long long)(k * 131 + me + w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!grid_arrive_and_wait(counter, (k + 1) * g, error)) return;
        const unsigned long long* theirs = buf + ((size_t)(k & 1) * g + other) * 256 * words;
        for (int w = 0; w < words; ++w)
            acc += __hip_atomic_load(theirs + (size_t)w * 256 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (acc == 0x123456789abcdefULL) sink[0] = acc;
}

// XCD-local variant: workgroups that run on the same XCD (HW_REG_XCC_ID) form a group with its own counter; data moves between workgroups of ONE XCD
// through that XCD's L2: plain stores are written through the per-CU cache and stay in the L2; the reader either uses sc1 loads (L2-served) or one
// agent-scope acquire (buffer_inv sc1 = drop this CU's cached lines) followed by plain loads.  No L2 is crossed; workgroup-scope accesses / buffer_inv sc0
// do NOT work (stale data: first version of this probe, and MI355X_MICROARCH.md).
// state: [0..7] members per XCD (filled in round 0), [8..15] barrier counters per XCD, [16] device-wide counter for the set-up round
__device__ __forceinline__ bool xcd_arrive_and_wait(int* counter, int target, int* error, bool acquire) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long long polls = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++polls > 4000000) { __hip_atomic_store(error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; break; }
        }
        if (acquire) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // buffer_inv sc1: drops THIS CU's cached lines (MI355X_MICROARCH.md: the L2 keeps its own)
    }
    __syncthreads();
    return ok;
}
__global__ __launch_bounds__(256) void xcd_kernel(int* state, int rounds, int* error, unsigned long long* buf, int words, unsigned long long* sink, int max_per_xcd, int inv_l1) {
    __shared__ int s_rank, s_members;
    const int xcc = (int)__builtin_amdgcn_s_getreg((31 << 11) | 20) & 7;        // HW_REG_XCC_ID
    if (threadIdx.x == 0) s_rank = __hip_atomic_fetch_add(state + xcc, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!grid_arrive_and_wait(state + 16, (int)gridDim.x, error)) return;       // everybody has registered
    if (threadIdx.x == 0) s_members = __hip_atomic_load(state + xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int rank = s_rank, members = s_members, partner = (rank + 1) % members;
    if (rank >= max_per_xcd) { if (threadIdx.x == 0) __hip_atomic_store(error, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
    unsigned long long acc = 0;
    for (int k = 0; k < rounds; ++k) {
        unsigned long long* mine = buf + (((size_t)(k & 1) * 8 + xcc) * max_per_xcd + rank) * 256 * words;
        for (int w = 0; w < words; ++w) mine[(size_t)w * 256 + threadIdx.x] = (unsigned long long)(k * 131 + rank + w);       // plain stores: written through to the XCD's L2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!xcd_arrive_and_wait(state + 8 + xcc, (k + 1) * members, error, inv_l1 != 0)) return;
        const unsigned long long* theirs = buf + (((size_t)(k & 1) * 8 + xcc) * max_per_xcd + partner) * 256 * words;
        for (int w = 0; w < words; ++w) {
            const unsigned long long v = inv_l1 ? theirs[(size_t)w * 256 + threadIdx.x]
                                                : __hip_atomic_load(theirs + (size_t)w * 256 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v != (unsigned long long)(k * 131 + partner + w)) __hip_atomic_store(error, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // stale data: the hand-over is NOT coherent
            acc += v;
        }
    }
    if (acc == 0x123456789abcdefULL) sink[0] = acc;
}

__global__ __launch_bounds__(256) void empty_kernel(int* sink) { if (sink && threadIdx.x == 1024) sink[0] = 1; }

static float elapsed_us(hipEvent_t a, hipEvent_t b) { float ms = 0; CK(hipEventElapsedTime(&ms, a, b)); return ms * 1e3f; }

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# %s, %d CUs\n", prop.gcnArchName, cus);
    int *counter, *error;
    unsigned long long *buf, *sink;
    CK(hipMalloc(&counter, 64)); CK(hipMalloc(&error, 64)); CK(hipMalloc(&sink, 64));
    const int max_words = 32, max_g = 4 * cus;
    CK(hipMalloc(&buf, (size_t)2 * 8 * max_g * 256 * max_words * 8));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int rounds = 200;
    printf("# grid barrier (agent-scope counter, one poller per workgroup), %d rounds per launch\n", rounds);
    for (int g : {cus / 2, cus, 2 * cus}) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemset(counter, 0, 64)); CK(hipMemset(error, 0, 64));
            CK(hipEventRecord(a, 0));
            hipLaunchKernelGGL(barrier_kernel, dim3(g), dim3(256), 0, 0, counter, rounds, error);
            CK(hipEventRecord(b, 0));
            CK(hipEventSynchronize(b));
            int err = 0; CK(hipMemcpy(&err, error, 4, hipMemcpyDeviceToHost));
            if (err) { printf("workgroups %4d: a workgroup gave up waiting (grid not co-resident?)\n", g); best = -1; break; }
            best = std::min(best, elapsed_us(a, b));
        }
        if (best > 0) printf("workgroups %4d: %7.2f us per barrier\n", g, best / rounds);
    }
    printf("# hand-over: write BYTES per workgroup (agent-scope stores), barrier, read the BYTES a workgroup on another XCD wrote (agent-scope loads)\n");
    for (int g : {cus, 2 * cus}) {
        for (int words : {1, 4, 16, 32}) {
            float best = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipMemset(counter, 0, 64)); CK(hipMemset(error, 0, 64));
                CK(hipEventRecord(a, 0));
                hipLaunchKernelGGL(handover_kernel, dim3(g), dim3(256), 0, 0, counter, rounds, error, buf, words, sink);
                CK(hipEventRecord(b, 0));
                CK(hipEventSynchronize(b));
                int err = 0; CK(hipMemcpy(&err, error, 4, hipMemcpyDeviceToHost));
                if (err) { best = -1; break; }
                best = std::min(best, elapsed_us(a, b));
            }
            if (best > 0) printf("workgroups %4d, %6d B per workgroup (%5.1f MB per round): %7.2f us per round\n", g, words * 256 * 8, (double)g * words * 2048 / 1e6, best / rounds);
            else printf("workgroups %4d, %6d B: gave up\n", g, words * 256 * 8);
        }
    }
    printf("# XCD-local hand-over: groups = the workgroups on one XCD (HW_REG_XCC_ID), per-XCD counter, plain stores (they stay in that XCD's L2), read back either with sc1 loads or with plain loads behind an acquire\n");
    {
        int* state;
        CK(hipMalloc(&state, 32 * sizeof(int)));
        for (int g : {cus, 2 * cus}) {
            const int max_per_xcd = g;       // (placement is the hardware's: size for the worst case)
            for (int inv_l1 = 0; inv_l1 < 2; ++inv_l1)
            for (int words : {0, 4, 16, 32}) {
                float best = 1e30f;
                int err = 0, members[8];
                for (int rep = 0; rep < 5; ++rep) {
                    CK(hipMemset(state, 0, 32 * sizeof(int))); CK(hipMemset(error, 0, 64));
                    CK(hipEventRecord(a, 0));
                    hipLaunchKernelGGL(xcd_kernel, dim3(g), dim3(256), 0, 0, state, rounds, error, buf, words, sink, max_per_xcd, inv_l1);
                    CK(hipEventRecord(b, 0));
                    CK(hipEventSynchronize(b));
                    CK(hipMemcpy(&err, error, 4, hipMemcpyDeviceToHost));
                    CK(hipMemcpy(members, state, 32, hipMemcpyDeviceToHost));
                    if (err) break;
                    best = std::min(best, elapsed_us(a, b));
                }
                if (err) printf("workgroups %4d, %6d B, %s: error %d (1 gave up waiting, 2 too many on one XCD, 3 STALE DATA)\n", g, words * 2048, inv_l1 ? "plain loads after one agent-scope acquire per workgroup" : "agent-scope loads (sc1: per-CU cache bypassed, L2-served)", err);
                else printf("workgroups %4d (per XCD %d %d %d %d %d %d %d %d), %6d B per workgroup, %s: %7.2f us per round\n", g, members[0], members[1],
                            members[2], members[3], members[4], members[5], members[6], members[7], words * 2048, inv_l1 ? "plain loads after one agent-scope acquire per workgroup" : "agent-scope loads (sc1: per-CU cache bypassed, L2-served)", best / rounds);
            }
        }
    }
    printf("# kernel boundary: back-to-back empty launches on one stream\n");
    for (int g : {cus, 2 * cus}) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(a, 0));
            for (int k = 0; k < rounds; ++k) hipLaunchKernelGGL(empty_kernel, dim3(g), dim3(256), 0, 0, (int*)nullptr);
            CK(hipEventRecord(b, 0));
            CK(hipEventSynchronize(b));
            best = std::min(best, elapsed_us(a, b));
        }
        printf("workgroups %4d: %7.2f us per empty launch\n", g, best / rounds);
    }
    return 0;
}
