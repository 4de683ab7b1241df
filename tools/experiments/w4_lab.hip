// Laboratory for a ONE-WAVE-PER-SIMD 256 x 256 bf16 GEMM loop (round 5, after tools/stag_lab.hip located the bound of the 8-wave loop in its LDS traffic):
// 4 waves, each owns a 128 x 128 quarter of the tile (4 x 4 MFMA 32x32x16 fragments, 256 accumulator registers), so a K tile costs
//   LDS reads   4 waves x (128 + 128) rows x 128 B = 128 KB   (8 waves of 128 x 64: 192 KB)
//   LDS-DMA     64 KB
// per 2048 MFMA cycles -- 94 B/clk against the 8-wave loop's 125 B/clk.  There is no second wave on a SIMD to cover anything: every wave interleaves its own
// fragment reads (next 16-wide K slice, double-buffered registers) and its DMA pieces (next K tile, other LDS stage) BETWEEN its MFMAs; one barrier per K tile.
//
//   C[M][N] (bf16) = A[M][K] * Bt[N][K]^T, M, N multiples of 256, K a multiple of 64.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/w4_lab_bin tools/w4_lab.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));

enum {
    F_NODMA = 1,        // ablation: no DMA inside the loop (stale operands)
    F_NOREAD = 2,       // ablation: no fragment reads inside the loop
    F_NOMFMA = 4,       // ablation: no MFMAs
    F_PRIO = 8,         // s_setprio 1 for the whole loop (nothing to prioritise against on the SIMD; other CU clients?)
    F_READS_FIRST = 16, // all 8 fragment reads of a slice at the top of the slot (default: one behind each of the first 8 MFMAs)
    F_DMA_LATE = 32,    // DMA pieces behind MFMAs 8.. of the slot (default: behind MFMAs 1, 3, 5, ... interleaved with the reads)
    F_NOFENCE = 64,     // no sched_barrier fences inside a slot: the compiler orders the slot
    F_GM4 = 256,        // tile order inside an XCD's range: groups of 4 tile rows, column-major inside a group (the 32 tiles an XCD runs at once: 4 x 8, 12 operand panels
                        // per K tile instead of the 33 of a 1 x 32 strip)
    F_GM8 = 512,        // groups of 8 tile rows (8 x 4)
    F_GM16 = 1024,
    F_BUF = 2048,       // (w4) buffer_load ... lds with an SGPR piece offset and a 32-bit lane offset instead of global_load_lds with a 64-bit lane address
    F_SWZ4 = 4096,      // (w4, feed experiment, WRONG results) source-side swizzle restricted to bit 2: whole quads of lanes swap, the order inside a quad ascends
    F_SWZ0 = 8192,      // (w4, feed experiment, WRONG results) no source-side swizzle: every lane group of 8 reads its row in ascending address order
    F_ROWMAP = 16384,   // (w4) piece ps of wave w stages rows ps*32 + w*8 .. +7 (the four waves' pieces of one index form one 32-row block) instead of w*64 + ps*8 .. +7
    F_CLOCK = 128,      // wave 0 of every workgroup: s_memtime (shader clock) and s_memrealtime (100 MHz) across the K loop -> cycles per K tile and the clock the loop ran at
};

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int xcd_remap(int lin, int total) {
    const int q = total >> 3, r = total & 7, xcd = lin & 7, idx = lin >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 v; v.x = lo; v.y = hi;
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, b2));
}

constexpr int STAGE = 65536, A_BYTES = 32768;

// P3, P0, P1, P2: DMA pieces (of this wave's 16 per K tile: 8 of A, 8 of B) issued in slice slot 3 of the PREVIOUS tile (behind the barrier) and in slots 0, 1, 2.
// BARPOS: MFMAs of slot 3 issued before the vmcnt(0) + barrier.
template <int F, int P3, int P0, int P1, int P2, int BARPOS>
__global__ __launch_bounds__(256) void w4_kernel(const unsigned short* __restrict__ A, const unsigned short* __restrict__ Bt, unsigned short* __restrict__ C, int M, int N,
                                                 int K, int ld, unsigned long long* __restrict__ prof) {
    static_assert(P3 + P0 + P1 + P2 == 16, "16 pieces per wave and K tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int tiles_n = N / 256;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    int tile_m = lid / tiles_n, tile_n = lid - tile_m * tiles_n;
    if (F & (F_GM4 | F_GM8 | F_GM16)) {
        constexpr int GM = (F & F_GM4) ? 4 : (F & F_GM8) ? 8 : 16;
        const int tiles_m = M / 256, per_group = GM * tiles_n, g = lid / per_group, i = lid - g * per_group;
        const int rows = tiles_m - g * GM < GM ? tiles_m - g * GM : GM;
        tile_m = g * GM + i % rows;
        tile_n = i / rows;
    }
    const int m0 = tile_m * 256, n0 = tile_n * 256;
    const int nk = K / 64;

    // DMA piece q of this wave (q < 8: A, else B): 8 rows x 128 B, rows (wid*8 + (q&7))*8 + (lane>>3); the lane fetches the 16-byte chunk that lds_off() expects
    // in its slot: chunk (lane&7) ^ ((row>>1)&7), and (row>>1)&7 = 4*(q&1) + (lane>>4) -- two lane offsets, everything else is uniform
    int voff[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) voff[par] = (lane >> 3) * ld + (((lane & 7) ^ ((4 * par + (lane >> 4)) & ((F & F_SWZ0) ? 0 : (F & F_SWZ4) ? 4 : 7))) << 3);
    const unsigned short* a_w = A + (long long)(m0 + wid * 64) * ld;
    const unsigned short* b_w = Bt + (long long)(n0 + wid * 64) * ld;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(A), 0, 0x7fffffff, 0x00027000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Bt), 0, 0x7fffffff, 0x00027000);
    const int a_s = (m0 + wid * 64) * ld * 2, b_s = (n0 + wid * 64) * ld * 2;          // (bytes; this laboratory's matrices stay below 2 GB)
    const unsigned short* a_r = A + (long long)(m0 + wid * 8) * ld;
    const unsigned short* b_r = Bt + (long long)(n0 + wid * 8) * ld;
    auto issue_piece = [&](int q, int buf, int k_run) {
        const bool is_b = q >= 8;
        const int ps = q & 7;
        if (F & F_ROWMAP) {
            char* dst = smem + buf * STAGE + (is_b ? A_BYTES : 0) + ps * 4096 + wid * 1024;
            const unsigned short* src = (is_b ? b_r : a_r) + (long long)(ps * 32) * ld + k_run;
            glds16(src + voff[wid & 1], dst);
            return;
        }
        char* dst = smem + buf * STAGE + (is_b ? A_BYTES : 0) + wid * 8192 + ps * 1024;
        if (F & F_BUF) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(is_b ? rs_b : rs_a, (__attribute__((address_space(3))) void*)dst, 16, voff[ps & 1] * 2,
                                                     (is_b ? b_s : a_s) + ps * 16 * ld + k_run * 2, 0, 0);
        } else {
            const unsigned short* src = (is_b ? b_w : a_w) + (long long)(ps * 8) * ld + k_run;
            glds16(src + voff[ps & 1], dst);
        }
    };

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31, fhalf = lane >> 5;
    const int a_frag = lds_off(wm * 128 + frow, fhalf), b_frag = A_BYTES + lds_off(wn * 128 + frow, fhalf);

    bf16x8 fa[2][4], fb[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            fa[s][t] = __builtin_bit_cast(bf16x8, uint4v{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});
            fb[s][t] = fa[s][t];
        }
    auto read_frag = [&](int i, int set, const char* stage, int ks) {          // i < 4: A fragment i, else B fragment i - 4
        if (i < 4) fa[set][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4v*>(stage + ((a_frag ^ (ks << 5)) + i * 4096)));
        else fb[set][i - 4] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4v*>(stage + ((b_frag ^ (ks << 5)) + (i - 4) * 4096)));
    };

    // prologue: tile 0 whole, the first P3 pieces of tile 1
#pragma unroll
    for (int q = 0; q < 16; ++q) issue_piece(q, 0, 0);
    if (nk > 1) {
#pragma unroll
        for (int q = 0; q < P3; ++q) issue_piece(q, 1, 64);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P3) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) read_frag(i, 0, smem, 0);
    if (F & F_NOREAD) {          // the ablation multiplies REAL operand values (tile 0's), not constants: what the matrix pipes toggle is part of what is measured
#pragma unroll
        for (int i = 0; i < 8; ++i) read_frag(i, 1, smem, 1);
    }
    if (F & F_PRIO) __builtin_amdgcn_s_setprio(1);

    constexpr bool FENCE = !(F & F_NOFENCE);
    unsigned long long c0 = 0, r0 = 0;
    if (F & F_CLOCK) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const char* st = smem + buf * STAGE;
        const char* st_next = smem + (buf ^ 1) * STAGE;
        const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            // pieces of this slot: slots 0..2 stage tile kt+1 into the other stage, slot 3 (behind the barrier) starts tile kt+2 into THIS stage
            const int q_lo = ks == 0 ? P3 : ks == 1 ? P3 + P0 : ks == 2 ? P3 + P0 + P1 : 0;
            const int q_n = ks == 0 ? P0 : ks == 1 ? P1 : ks == 2 ? P2 : P3;
            // (no branch inside the MFMA stream: past the last tile the pieces re-fetch K offset 0 into a stage nobody reads any more)
            const bool dma_on = !(F & F_NODMA);
            const int dma_buf = ks == 3 ? buf : buf ^ 1;
            const int dma_k = (ks == 3 ? (more2 ? kt + 2 : 0) : (more1 ? kt + 1 : 0)) * 64;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the fragments of this slice (read one slot ago)
            if (FENCE) __builtin_amdgcn_sched_barrier(0);
            int q_done = 0, r_done = 0;
            if (ks < 3 && (F & F_READS_FIRST) && !(F & F_NOREAD)) {
#pragma unroll
                for (int i = 0; i < 8; ++i) read_frag(i, nxt, st, ks + 1);
                r_done = 8;
                if (FENCE) __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (ks == 3 && i == BARPOS) {
                    // every piece of tile kt+1 of this wave has landed; behind the barrier: of every wave, and every wave is done reading tile kt's stage
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    if (FENCE) __builtin_amdgcn_sched_barrier(0);
                }
                const int ti = i >> 2, tj = i & 3;
                if (!(F & F_NOMFMA)) acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cur][tj], fa[cur][ti], acc[ti][tj], 0, 0, 0);
                const bool reads_open = ks < 3 || i >= BARPOS;
                // (r_done, q_done stay compile-time: the run-time conditions guard the instruction only.  Past the last tile slot 3 reads the other stage
                // into registers nobody uses.)
                if (!(F & F_NOREAD) && reads_open && r_done < 8) {
                    if (ks < 3) read_frag(r_done, nxt, st, ks + 1); else read_frag(r_done, nxt, st_next, 0);
                    ++r_done;
                }
                const bool dma_slot = (F & F_DMA_LATE) ? (i >= 8) : ((i & 1) == 1);
                if (reads_open && dma_slot && q_done < q_n) {
                    if (dma_on) issue_piece(q_lo + q_done, dma_buf, dma_k);
                    ++q_done;
                }
                if (i == 15) {          // whatever the pattern left over
                    for (; !(F & F_NOREAD) && r_done < 8; ++r_done) {
                        if (ks < 3) read_frag(r_done, nxt, st, ks + 1); else read_frag(r_done, nxt, st_next, 0);
                    }
                    for (; q_done < q_n; ++q_done)
                        if (dma_on) issue_piece(q_lo + q_done, dma_buf, dma_k);
                }
                if (FENCE) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (F & F_PRIO) __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((F & F_CLOCK) && tid == 0) {
        prof[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0;
        prof[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    // epilogue: lane holds rows wm*128 + ti*32 + frow, columns wn*128 + tj*32 + 8*q + 4*fhalf + e (reg 4*q + e): 8-byte stores
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
        const int m = m0 + wm * 128 + ti * 32 + frow;
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * 128 + tj * 32 + 8 * q + 4 * fhalf;
                uint2 t;
                t.x = pack_bf16x2(acc[ti][tj][4 * q], acc[ti][tj][4 * q + 1]);
                t.y = pack_bf16x2(acc[ti][tj][4 * q + 2], acc[ti][tj][4 * q + 3]);
                *reinterpret_cast<uint2*>(C + (long long)m * N + n) = t;
            }
    }
}

// ---- the same wave tile on a FOUR-STAGE ring of 32-wide K stages (4 x 32 KB): the operand of stage s+4 is requested while stage s is multiplied -- three stages
// (3 072 MFMA cycles) of cover for the L2 / fabric round trip instead of the two-stage loop's one tile at best.  Per stage and wave: 8 DMA pieces (1 KB = 16 rows x 64 B;
// 4 of A, 4 of B), two 16-wide slices of 16 MFMAs, one barrier (in slice 1, behind BARPOS MFMAs).  After barrier X_s every wave has all of stage s in registers, so the
// slot of stage s is free: H1 pieces of stage s+4 go out behind the barrier in slice 1, the other 8 - H1 in slice 0 of stage s+1.  At X_s the pieces of stages s+2 and
// s+3 may still be in flight: vmcnt(16) retires stage s+1 (in-order completion).
__device__ __forceinline__ int lds_off32(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }
constexpr int RSTAGE = 32768, RA_BYTES = 16384;

template <int F, int H1, int BARPOS>
__global__ __launch_bounds__(256) void w4r_kernel(const unsigned short* __restrict__ A, const unsigned short* __restrict__ Bt, unsigned short* __restrict__ C, int M, int N,
                                                  int K, int ld, unsigned long long* __restrict__ prof) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int tiles_n = N / 256;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    int tile_m = lid / tiles_n, tile_n = lid - tile_m * tiles_n;
    if (F & (F_GM4 | F_GM8 | F_GM16)) {
        constexpr int GM = (F & F_GM4) ? 4 : (F & F_GM8) ? 8 : 16;
        const int tiles_m = M / 256, per_group = GM * tiles_n, g = lid / per_group, i = lid - g * per_group;
        const int rows = tiles_m - g * GM < GM ? tiles_m - g * GM : GM;
        tile_m = g * GM + i % rows;
        tile_n = i / rows;
    }
    const int m0 = tile_m * 256, n0 = tile_n * 256;
    const int ns = K / 32;

    // piece q of this wave (q < 4: A, else B): rows (wid*4 + (q&3))*16 + (lane>>2), physical chunk lane&3 <- logical chunk (lane&3) ^ ((row>>2)&3) = (lane&3) ^ ((lane>>4)&3)
    const int voff = (lane >> 2) * ld + (((lane & 3) ^ ((lane >> 4) & 3)) << 3);
    const unsigned short* a_w = A + (long long)(m0 + wid * 64) * ld;
    const unsigned short* b_w = Bt + (long long)(n0 + wid * 64) * ld;
    auto issue_piece = [&](int q, int slot, int k_run) {
        const bool is_b = q >= 4;
        const int ps = q & 3;
        char* dst = smem + slot * RSTAGE + (is_b ? RA_BYTES : 0) + wid * 4096 + ps * 1024;
        const unsigned short* src = (is_b ? b_w : a_w) + (long long)(ps * 16) * ld + k_run;
        glds16(src + voff, dst);
    };

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31, fhalf = lane >> 5;
    const int a_frag = lds_off32(wm * 128 + frow, fhalf), b_frag = RA_BYTES + lds_off32(wn * 128 + frow, fhalf);
    bf16x8 fa[2][4], fb[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            fa[s][t] = __builtin_bit_cast(bf16x8, uint4v{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});
            fb[s][t] = fa[s][t];
        }
    auto read_frag = [&](int i, int set, const char* stage, int ks) {          // i < 4: A fragment i, else B fragment i - 4
        if (i < 4) fa[set][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4v*>(stage + ((a_frag ^ (ks << 5)) + i * 2048)));
        else fb[set][i - 4] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4v*>(stage + ((b_frag ^ (ks << 5)) + (i - 4) * 2048)));
    };
    auto k_of = [&](int stage) { return stage < ns ? stage * 32 : 0; };          // (past the end: a harmless re-fetch keeps the vmcnt arithmetic uniform)

    // prologue: stages 0, 1, 2 whole; stage 0 landed -> barrier -> H1 pieces of stage 3, fragments (0, 0)
#pragma unroll
    for (int st = 0; st < 3; ++st)
#pragma unroll
        for (int q = 0; q < 8; ++q) issue_piece(q, st, k_of(st));
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int q = 0; q < H1; ++q) issue_piece(q, 3, k_of(3));
#pragma unroll
    for (int i = 0; i < 8; ++i) read_frag(i, 0, smem, 0);
    if (F & F_PRIO) __builtin_amdgcn_s_setprio(1);

    constexpr bool FENCE = !(F & F_NOFENCE);
    unsigned long long c0 = 0, r0 = 0;
    if (F & F_CLOCK) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int s = 0; s < ns; ++s) {
        const char* st = smem + (s & 3) * RSTAGE;
        const char* st_next = smem + ((s + 1) & 3) * RSTAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // slice 0: the second part of stage s+3 (its first H1 pieces went out behind X_{s-1}); slice 1, behind X_s: the first H1 pieces of stage s+4 into the slot of stage s
            const int q_lo = ks == 0 ? H1 : 0, q_n = ks == 0 ? 8 - H1 : H1;
            const int dma_stage = ks == 0 ? s + 3 : s + 4;
            const int dma_slot = dma_stage & 3, dma_k = k_of(dma_stage);
            const bool dma_on = !(F & F_NODMA);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (FENCE) __builtin_amdgcn_sched_barrier(0);
            int q_done = 0, r_done = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (ks == 1 && i == BARPOS) {
                    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    if (FENCE) __builtin_amdgcn_sched_barrier(0);
                }
                const int ti = i >> 2, tj = i & 3;
                if (!(F & F_NOMFMA)) acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks][tj], fa[ks][ti], acc[ti][tj], 0, 0, 0);
                const bool open = ks == 0 || i >= BARPOS;
                if (!(F & F_NOREAD) && open && r_done < 8) {
                    if (ks == 0) read_frag(r_done, 1, st, 1); else read_frag(r_done, 0, st_next, 0);
                    ++r_done;
                }
                const bool dma_here = (F & F_DMA_LATE) ? (i >= 8) : ((i & 1) == 1);
                if (open && dma_here && q_done < q_n) {
                    if (dma_on) issue_piece(q_lo + q_done, dma_slot, dma_k);
                    ++q_done;
                }
                if (i == 15) {
                    for (; !(F & F_NOREAD) && r_done < 8; ++r_done) {
                        if (ks == 0) read_frag(r_done, 1, st, 1); else read_frag(r_done, 0, st_next, 0);
                    }
                    for (; q_done < q_n; ++q_done)
                        if (dma_on) issue_piece(q_lo + q_done, dma_slot, dma_k);
                }
                if (FENCE) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (F & F_PRIO) __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((F & F_CLOCK) && tid == 0) {
        prof[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0;
        prof[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
        const int m = m0 + wm * 128 + ti * 32 + frow;
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * 128 + tj * 32 + 8 * q + 4 * fhalf;
                uint2 t;
                t.x = pack_bf16x2(acc[ti][tj][4 * q], acc[ti][tj][4 * q + 1]);
                t.y = pack_bf16x2(acc[ti][tj][4 * q + 2], acc[ti][tj][4 * q + 3]);
                *reinterpret_cast<uint2*>(C + (long long)m * N + n) = t;
            }
    }
}


// ---- the same wave tile, operands staged THROUGH REGISTERS: global_load_dwordx4 -> 64 VGPRs -> ds_write_b128 (swizzle on the LDS side) instead of global_load_lds.
// A global_load_lds costs its wave ~60-100 issue cycles (M0 set-up, the TA hand-shake) and one wave per SIMD has nobody to cover them; a plain load is issued and
// forgotten, and its data waits in registers for a whole K tile: tile t+2 is requested while tile t is multiplied (two tiles of cover, two LDS stages).  Per K tile and
// wave: 16 loads + 16 ds_write + 32 ds_read + 64 MFMAs.  Order per wave: ... W(t+1, j), L(t+2, j), W(t+1, j+1), ... -- before W(t+1, j) exactly 15 younger loads may be
// in flight (the compiler derives the same vmcnt(15)); the writes of tile t+1 start behind barrier X_{t-1} (everyone has tile t-1 in registers) and are retired
// (lgkmcnt(0)) before X_t.
template <int F, int P3, int P0, int P1, int P2, int BARPOS>
__global__ __launch_bounds__(256) void w4v_kernel(const unsigned short* __restrict__ A, const unsigned short* __restrict__ Bt, unsigned short* __restrict__ C, int M, int N,
                                                  int K, int ld, unsigned long long* __restrict__ prof) {
    static_assert(P3 + P0 + P1 + P2 == 16, "16 pieces per wave and K tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int tiles_n = N / 256;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    int tile_m = lid / tiles_n, tile_n = lid - tile_m * tiles_n;
    if (F & (F_GM4 | F_GM8 | F_GM16)) {
        constexpr int GM = (F & F_GM4) ? 4 : (F & F_GM8) ? 8 : 16;
        const int tiles_m = M / 256, per_group = GM * tiles_n, g = lid / per_group, i = lid - g * per_group;
        const int rows = tiles_m - g * GM < GM ? tiles_m - g * GM : GM;
        tile_m = g * GM + i % rows;
        tile_n = i / rows;
    }
    const int m0 = tile_m * 256, n0 = tile_n * 256;
    const int nk = K / 64;

    // piece q (q < 8: A, else B): rows (wid*8 + (q&7))*8 + (lane>>3), the lane loads chunk lane&7 of its row (a row = 8 lanes = 128 contiguous bytes) and stores it
    // at lds_off(row, lane&7): (row>>1)&7 = 4*(q&1) + (lane>>4)
    const int goff = (lane >> 3) * ld + ((lane & 7) << 3);
    int soff[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) soff[par] = (lane >> 3) * 128 + (((lane & 7) ^ (4 * par + (lane >> 4))) << 4);
    const unsigned short* a_w = A + (long long)(m0 + wid * 64) * ld + goff;
    const unsigned short* b_w = Bt + (long long)(n0 + wid * 64) * ld + goff;
    uint4v stg[16];
    auto load_piece = [&](int q, int k_run) {
        const unsigned short* src = (q >= 8 ? b_w : a_w) + (long long)((q & 7) * 8) * ld + k_run;
        stg[q] = *reinterpret_cast<const uint4v*>(src);
    };
    auto store_piece = [&](int q, int buf) {
        char* dst = smem + buf * STAGE + (q >= 8 ? A_BYTES : 0) + wid * 8192 + (q & 7) * 1024 + soff[q & 1];
        *reinterpret_cast<uint4v*>(dst) = stg[q];
    };

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31, fhalf = lane >> 5;
    const int a_frag = lds_off(wm * 128 + frow, fhalf), b_frag = A_BYTES + lds_off(wn * 128 + frow, fhalf);
    bf16x8 fa[2][4], fb[2][4];
    auto read_frag = [&](int i, int set, const char* stage, int ks) {
        if (i < 4) fa[set][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4v*>(stage + ((a_frag ^ (ks << 5)) + i * 4096)));
        else fb[set][i - 4] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4v*>(stage + ((b_frag ^ (ks << 5)) + (i - 4) * 4096)));
    };
    auto k_of = [&](int t) { return t < nk ? t * 64 : 0; };

    // prologue: tile 0 through the registers into stage 0, tile 1 requested, the first P3 writes of tile 1 belong to "slot 3 of tile -1"
#pragma unroll
    for (int q = 0; q < 16; ++q) load_piece(q, 0);
#pragma unroll
    for (int q = 0; q < 16; ++q) { store_piece(q, 0); load_piece(q, k_of(1)); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < P3; ++q) { store_piece(q, 1); load_piece(q, k_of(2)); }
#pragma unroll
    for (int i = 0; i < 8; ++i) read_frag(i, 0, smem, 0);
    if (F & F_PRIO) __builtin_amdgcn_s_setprio(1);

    constexpr bool FENCE = !(F & F_NOFENCE);
    unsigned long long c0 = 0, r0 = 0;
    if (F & F_CLOCK) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const char* st = smem + buf * STAGE;
        const char* st_next = smem + (buf ^ 1) * STAGE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            // slots 0..2: writes of tile kt+1 into the other stage + loads of tile kt+2; slot 3 (behind the barrier): the first writes of tile kt+2 into THIS stage + loads of kt+3
            const int q_lo = ks == 0 ? P3 : ks == 1 ? P3 + P0 : ks == 2 ? P3 + P0 + P1 : 0;
            const int q_n = ks == 0 ? P0 : ks == 1 ? P1 : ks == 2 ? P2 : P3;
            const int w_buf = ks == 3 ? buf : buf ^ 1;
            const int l_k = k_of(ks == 3 ? kt + 3 : kt + 2);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the fragments of this slice; in slot 3 also: every write of tile kt+1 of this wave
            if (FENCE) __builtin_amdgcn_sched_barrier(0);
            int q_done = 0, r_done = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (ks == 3 && i == BARPOS) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    if (FENCE) __builtin_amdgcn_sched_barrier(0);
                }
                const int ti = i >> 2, tj = i & 3;
                if (!(F & F_NOMFMA)) acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cur][tj], fa[cur][ti], acc[ti][tj], 0, 0, 0);
                const bool open = ks < 3 || i >= BARPOS;
                if (!(F & F_NOREAD) && open && r_done < 8) {
                    if (ks < 3) read_frag(r_done, nxt, st, ks + 1); else read_frag(r_done, nxt, st_next, 0);
                    ++r_done;
                }
                const bool here = (F & F_DMA_LATE) ? (i >= 8) : ((i & 1) == 1);
                if (open && here && q_done < q_n) {
                    if (!(F & F_NODMA)) { store_piece(q_lo + q_done, w_buf); load_piece(q_lo + q_done, l_k); }
                    ++q_done;
                }
                if (i == 15) {
                    for (; !(F & F_NOREAD) && r_done < 8; ++r_done) {
                        if (ks < 3) read_frag(r_done, nxt, st, ks + 1); else read_frag(r_done, nxt, st_next, 0);
                    }
                    for (; q_done < q_n; ++q_done)
                        if (!(F & F_NODMA)) { store_piece(q_lo + q_done, w_buf); load_piece(q_lo + q_done, l_k); }
                }
                if (FENCE) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (F & F_PRIO) __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if ((F & F_CLOCK) && tid == 0) {
        prof[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0;
        prof[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
        const int m = m0 + wm * 128 + ti * 32 + frow;
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * 128 + tj * 32 + 8 * q + 4 * fhalf;
                uint2 t;
                t.x = pack_bf16x2(acc[ti][tj][4 * q], acc[ti][tj][4 * q + 1]);
                t.y = pack_bf16x2(acc[ti][tj][4 * q + 2], acc[ti][tj][4 * q + 3]);
                *reinterpret_cast<uint2*>(C + (long long)m * N + n) = t;
            }
    }
}


// ---- what the matrix pipes alone sustain: one wave per SIMD issuing MFMAs back to back on operand values loaded ONCE from memory (so they toggle like real data), 256 accumulator
// registers, no LDS, no loads in the loop.  SHAPE 32: v_mfma_f32_32x32x16_bf16 (64 per "K tile" of 2 048 cycles); SHAPE 16: v_mfma_f32_16x16x32_bf16 (128 per tile) -- the shape
// hipBLASLt's kernel uses.  Same flops per cycle; the question is the clock the chip holds on each.
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_power_kernel(const unsigned short* __restrict__ A, const unsigned short* __restrict__ Bt, float* __restrict__ out, int tiles,
                                                         unsigned long long* __restrict__ prof) {
    const int tid = threadIdx.x;
    bf16x8 fa[8], fb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        fa[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4v*>(A + ((size_t)blockIdx.x * 2048 + i * 256 + tid) * 8));
        fb[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4v*>(Bt + ((size_t)blockIdx.x * 2048 + i * 256 + tid) * 8));
    }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float keep = 0.f;
    if (SHAPE == 32) {
        f32x16 acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int t = 0; t < tiles; ++t) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[(i + ks) & 7], fa[(i >> 2) + 2 * (ks & 1) + (ks >> 1)], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) keep += acc[i][i];
    } else {
        f32x4 acc[64];
#pragma unroll
        for (int i = 0; i < 64; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        for (int t = 0; t < tiles; ++t) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 64; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[(i + ks) & 7], fa[((i >> 3) + 4 * ks) & 7], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 64; ++i) keep += acc[i][i & 3];
    }
    if (tid == 0) {
        prof[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0;
        prof[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    out[(size_t)blockIdx.x * 256 + tid] = keep;
}

template <int SHAPE>
static void run_power(const char* name, const unsigned short* A, const unsigned short* Bt, float* out) {
    static unsigned long long* prof = nullptr;
    if (!prof) CK(hipMalloc(&prof, 65536 * 16));
    const int grid = 1024, tiles = 128, reps = 8;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(mfma_power_kernel<SHAPE>, dim3(grid), dim3(256), 0, 0, A, Bt, out, tiles, prof);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(mfma_power_kernel<SHAPE>, dim3(grid), dim3(256), 0, 0, A, Bt, out, tiles, prof);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> h((size_t)grid * 2);
    CK(hipMemcpy(h.data(), prof, h.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0, real = 0;
    for (int g = 0; g < grid; ++g) { cyc += (double)h[2 * g]; real += (double)h[2 * g + 1]; }
    const double flops = 2.0 * 256 * 256 * 64 * (double)tiles * grid;
    printf("%-40s %9.2f us  %7.1f TF   | %.0f cycles per 2048-cycle tile, clock %.0f MHz\n", name, ms * 1e3 / reps, flops / (ms * 1e-3 / reps) * 1e-12, cyc / grid / tiles,
           cyc / real * 100.0);
    fflush(stdout);
}


// mode 0: uniform in [-0.5, 0.5) (every mantissa bit toggles); 1: values from {-1, 0, 1}; 2: all 1.0; 3: what a convolution of the network multiplies -- seed 1 (A):
// relu(normal), half zeros; seed 2 (B): 0.05 * normal -- the same instruction stream at four switching activities
__global__ void fill_bf16(unsigned short* p, size_t n, unsigned seed, int mode) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        float v = ((int)(h & 0xffff) - 32768) * (1.0f / 65536.0f);
        if (mode == 1) v = (float)((int)(h % 3u) - 1);
        if (mode == 2) v = 1.0f;
        if (mode == 3) {
            unsigned h2 = h * 747796405u + 2891336453u;
            h2 ^= h2 >> 16;
            const float u1 = ((h & 0xffffff) + 1) * (1.0f / 16777217.0f), u2 = (h2 & 0xffffff) * (1.0f / 16777216.0f);
            const float g = sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
            v = seed == 1u ? fmaxf(g, 0.0f) : 0.05f * g;
        }
        p[i] = (unsigned short)(__float_as_uint(v) >> 16);
    }
}
__global__ void ref_gemm_kernel(const unsigned short* __restrict__ A, const unsigned short* __restrict__ Bt, float* __restrict__ R, int N, int K, int ld, int m_lo, int n_lo) {
    const int n = n_lo + blockIdx.x * blockDim.x + threadIdx.x, m = m_lo + blockIdx.y;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc += __uint_as_float((unsigned)A[(size_t)m * ld + k] << 16) * __uint_as_float((unsigned)Bt[(size_t)n * ld + k] << 16);
    R[(size_t)blockIdx.y * 256 + blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void cmp_kernel(const unsigned short* __restrict__ C, int N, const float* __restrict__ R, int m_lo, int n_lo, unsigned int* err) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    const float c = __uint_as_float((unsigned)C[(size_t)(m_lo + m) * N + n_lo + n] << 16), r = R[(size_t)m * 256 + n];
    const float d = fabsf(c - r);
    atomicMax(err, __float_as_uint(d));
    if (d > 0.01f * fabsf(r) + 0.02f) atomicAdd(err + 1, 1u);
}
__global__ void checksum_kernel(const unsigned int* __restrict__ C, size_t n_words, unsigned long long* out) {
    unsigned long long s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x) s += (unsigned long long)C[i] * (unsigned)(i | 1);
    atomicAdd(out, s);
}

typedef void (*KernFn)(const unsigned short*, const unsigned short*, unsigned short*, int, int, int, int, unsigned long long*);
static int g_ld_pad = 0;          // row pitch of A and Bt = K + g_ld_pad elements (a power-of-two pitch puts a tile's rows 16 KB apart)
static void run_kernel(const char* name, KernFn kern, int F, int k_unit, const unsigned short* A, const unsigned short* Bt, unsigned short* C, int M, int N, int K, float* ref,
                       unsigned int* err, unsigned long long* sum, int reps) {
    static unsigned long long* prof = nullptr;
    if (!prof) CK(hipMalloc(&prof, 65536 * 16));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    const int grid = (M / 256) * (N / 256);
    CK(hipMemsetAsync(C, 0xff, (size_t)M * N * 2, 0));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 131072, 0, A, Bt, C, M, N, K, K + g_ld_pad, prof);
    CK(hipGetLastError());
    unsigned int herr[2] = {0, 0};
    int races = 0;
    if (!(F & (F_NODMA | F_NOREAD | F_NOMFMA))) {
        unsigned int tot_bad = 0; float max_err = 0;
        for (int blk = 0; blk < 2; ++blk) {
            const int m_lo = blk ? M - 256 : 0, n_lo = blk ? N - 256 : 0;
            CK(hipMemsetAsync(err, 0, 8, 0));
            hipLaunchKernelGGL(ref_gemm_kernel, dim3(4, 256), dim3(64), 0, 0, A, Bt, ref, N, K, g_ld_pad + K, m_lo, n_lo);
            hipLaunchKernelGGL(cmp_kernel, dim3(4, 256), dim3(64), 0, 0, C, N, ref, m_lo, n_lo, err);
            CK(hipMemcpy(herr, err, 8, hipMemcpyDeviceToHost));
            float e; memcpy(&e, &herr[0], 4);
            if (e > max_err) max_err = e;
            tot_bad += herr[1];
        }
        herr[1] = tot_bad; memcpy(&herr[0], &max_err, 4);
        // race screen: the whole output must checksum the same on every rerun
        unsigned long long first = 0;
        for (int r = 0; r < 5; ++r) {
            CK(hipMemsetAsync(sum, 0, 8, 0));
            if (r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 131072, 0, A, Bt, C, M, N, K, K + g_ld_pad, prof);
            hipLaunchKernelGGL(checksum_kernel, dim3(2048), dim3(256), 0, 0, reinterpret_cast<const unsigned int*>(C), (size_t)M * N / 2, sum);
            unsigned long long h; CK(hipMemcpy(&h, sum, 8, hipMemcpyDeviceToHost));
            if (!r) first = h; else if (h != first) ++races;
        }
    }
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 131072, 0, A, Bt, C, M, N, K, K + g_ld_pad, prof);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 131072, 0, A, Bt, C, M, N, K, K + g_ld_pad, prof);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / reps;
    float e; memcpy(&e, &herr[0], 4);
    printf("%-58s %9.2f us  %7.1f TF   max err %.4f bad %u reruns differing %d", name, us, 2.0 * M * N * K / us * 1e-6, e, herr[1], races);
    if (F & F_CLOCK) {
        std::vector<unsigned long long> h((size_t)grid * 2);
        CK(hipMemcpy(h.data(), prof, h.size() * 8, hipMemcpyDeviceToHost));
        double cyc = 0, real = 0;
        for (int g = 0; g < grid; ++g) { cyc += (double)h[2 * g]; real += (double)h[2 * g + 1]; }
        printf("   | K loop: %.0f shader cycles per K tile, %.0f ns per K tile, clock %.0f MHz", cyc / grid / (K / 64), real / grid / (K / 64) * 10.0, cyc / real * 100.0);
        (void)k_unit;
    }
    printf("\n");
    fflush(stdout);
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
}

template <int F, int P3, int P0, int P1, int P2, int BARPOS>
static void run_variant(const char* name, const unsigned short* A, const unsigned short* Bt, unsigned short* C, int M, int N, int K, float* ref, unsigned int* err,
                        unsigned long long* sum, int reps) {
    run_kernel(name, &w4_kernel<F, P3, P0, P1, P2, BARPOS>, F, 64, A, Bt, C, M, N, K, ref, err, sum, reps);
}
template <int F, int P3, int P0, int P1, int P2, int BARPOS>
static void run_vgpr(const char* name, const unsigned short* A, const unsigned short* Bt, unsigned short* C, int M, int N, int K, float* ref, unsigned int* err,
                     unsigned long long* sum, int reps) {
    run_kernel(name, &w4v_kernel<F, P3, P0, P1, P2, BARPOS>, F, 64, A, Bt, C, M, N, K, ref, err, sum, reps);
}
template <int F, int H1, int BARPOS>
static void run_ring(const char* name, const unsigned short* A, const unsigned short* Bt, unsigned short* C, int M, int N, int K, float* ref, unsigned int* err,
                     unsigned long long* sum, int reps) {
    run_kernel(name, &w4r_kernel<F, H1, BARPOS>, F, 32, A, Bt, C, M, N, K, ref, err, sum, reps);
}

int main(int argc, char** argv) {
    const int sizes[][3] = {{8192, 8192, 8192}, {4096, 4096, 4096}, {131072, 256, 1088 - 1088 % 64}, {32768, 256, 1024}};
    unsigned short *A, *B, *C;
    const size_t max_el = (size_t)131072 * (1088 + 192);
    CK(hipMalloc(&A, max_el * 2)); CK(hipMalloc(&B, max_el * 2)); CK(hipMalloc(&C, max_el * 2));
    float* ref; CK(hipMalloc(&ref, 256 * 256 * 4));
    unsigned int* err; CK(hipMalloc(&err, 8));
    unsigned long long* sum; CK(hipMalloc(&sum, 8));
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    const int mode = argc > 2 ? atoi(argv[2]) : 0;
    hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, A, max_el, 1u, mode);
    hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, B, max_el, 2u, mode);
    CK(hipDeviceSynchronize());
    if (argc > 1 && !strcmp(argv[1], "pmc")) {          // one launch set per variant at 8192^3 for a rocprofv3 --pmc pass (kernel names carry the variant)
        const int M = 8192, N = 8192, K = 8192, reps = 1;
#define RUN(F, P3, P0, P1, P2, BP) run_variant<(F), P3, P0, P1, P2, BP>(#F " P " #P3 " " #P0 " " #P1 " " #P2 " bar " #BP, A, B, C, M, N, K, ref, err, sum, reps)
        RUN(F_GM4, 6, 6, 4, 0, 4);
        RUN(0, 6, 6, 4, 0, 4);
        RUN(F_GM4 | F_NOMFMA, 6, 6, 4, 0, 4);
        RUN(F_GM4 | F_NOMFMA | F_SWZ0, 6, 6, 4, 0, 4);
        RUN(F_GM4 | F_SWZ0 | F_NODMA, 6, 6, 4, 0, 4);
        run_vgpr<F_GM4, 4, 4, 4, 4, 4>("vgpr F_GM4", A, B, C, M, N, K, ref, err, sum, reps);
#undef RUN
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "power")) {          // the matrix pipes alone, two MFMA shapes, the operand fill of argv[2]
        float* out; CK(hipMalloc(&out, (size_t)1024 * 256 * 4));
        for (int rep = 0; rep < 3; ++rep) {
            run_power<32>("mfma only 32x32x16", A, B, out);
            run_power<16>("mfma only 16x16x32", A, B, out);
        }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "swz")) {          // source-side swizzle experiment (feed only; results of the SWZ variants are wrong by construction)
        const int M = 8192, N = 8192, K = 8192, reps = 8;
#define RUN(F, P3, P0, P1, P2, BP) run_variant<(F), P3, P0, P1, P2, BP>(#F " P " #P3 " " #P0 " " #P1 " " #P2 " bar " #BP, A, B, C, M, N, K, ref, err, sum, reps)
        for (int rep = 0; rep < 3; ++rep) {
            RUN(F_CLOCK | F_GM4, 6, 6, 4, 0, 4);
            RUN(F_CLOCK | F_GM4 | F_ROWMAP, 6, 6, 4, 0, 4);
            RUN(F_CLOCK | F_GM4 | F_ROWMAP, 8, 8, 0, 0, 4);
            RUN(F_CLOCK | F_GM4 | F_ROWMAP | F_NOMFMA, 6, 6, 4, 0, 4);
            RUN(F_CLOCK | F_GM4 | F_ROWMAP | F_NOMFMA | F_NOREAD, 6, 6, 4, 0, 4);
            RUN(F_CLOCK | F_GM4, 12, 4, 0, 0, 4);
            RUN(F_CLOCK | F_GM4, 8, 8, 0, 0, 4);
            RUN(F_CLOCK | F_GM4, 8, 8, 0, 0, 0);
            RUN(F_CLOCK | F_GM4 | F_DMA_LATE, 8, 8, 0, 0, 4);
            RUN(F_CLOCK | F_GM4 | F_NOMFMA, 6, 6, 4, 0, 4);
            RUN(F_CLOCK | F_GM4 | F_NOMFMA | F_SWZ4, 6, 6, 4, 0, 4);
            RUN(F_CLOCK | F_GM4 | F_NOMFMA | F_SWZ0, 6, 6, 4, 0, 4);
            RUN(F_CLOCK | F_GM4 | F_NOMFMA | F_NOREAD, 6, 6, 4, 0, 4);
            RUN(F_CLOCK | F_GM4 | F_NOMFMA | F_NOREAD | F_SWZ4, 6, 6, 4, 0, 4);
            RUN(F_CLOCK | F_GM4 | F_NOMFMA | F_NOREAD | F_SWZ0, 6, 6, 4, 0, 4);
            RUN(F_CLOCK | F_GM4 | F_NOREAD, 6, 6, 4, 0, 4);
            RUN(F_CLOCK | F_GM4 | F_NOREAD | F_SWZ0, 6, 6, 4, 0, 4);
        }
#undef RUN
        return 0;
    }
    if (argc > 3) {          // row-pitch experiment: K + pad elements per row
        for (int rep = 0; rep < 2; ++rep)
            for (int pad : {0, 64, 192, 0, 64, 192}) {
                g_ld_pad = pad;
                const int M = 8192, N = 8192, K = 8192, reps = 8;
                printf("# row pitch K + %d  M %d N %d K %d\n", pad, M, N, K);
#define RUN(F, P3, P0, P1, P2, BP) run_variant<(F), P3, P0, P1, P2, BP>(#F " P " #P3 " " #P0 " " #P1 " " #P2 " bar " #BP, A, B, C, M, N, K, ref, err, sum, reps)
                RUN(F_CLOCK | F_GM4, 6, 6, 4, 0, 4);
                RUN(F_CLOCK | F_GM4 | F_NOMFMA, 6, 6, 4, 0, 4);
                run_ring<F_CLOCK | F_GM4, 4, 4>("ring F_CLOCK | F_GM4 H1 4 bar 4", A, B, C, M, N, K, ref, err, sum, reps);
#undef RUN
            }
        return 0;
    }
    if (mode) {          // operand-data experiment: the full loop only
        for (int rep = 0; rep < 3; ++rep) {
            const int M = 8192, N = 8192, K = 8192, reps = 8;
            printf("# data mode %d M %d N %d K %d\n", mode, M, N, K);
#define RUN(F, P3, P0, P1, P2, BP) run_variant<(F), P3, P0, P1, P2, BP>(#F " P " #P3 " " #P0 " " #P1 " " #P2 " bar " #BP, A, B, C, M, N, K, ref, err, sum, reps)
            RUN(F_CLOCK | F_GM4, 6, 6, 4, 0, 4);
            run_ring<F_CLOCK | F_GM4, 4, 4>("ring F_CLOCK | F_GM4 H1 4 bar 4", A, B, C, M, N, K, ref, err, sum, reps);
            run_vgpr<F_CLOCK | F_GM4, 4, 4, 4, 4, 4>("vgpr F_CLOCK | F_GM4 P 4 4 4 4 bar 4", A, B, C, M, N, K, ref, err, sum, reps);
            RUN(F_CLOCK | F_NODMA | F_NOREAD, 6, 6, 4, 0, 4);
#undef RUN
        }
        return 0;
    }
    for (const auto& s : sizes) {
        const int M = s[0], N = s[1], K = s[2];
        const int reps = (M >= 8192 && N >= 8192) ? 8 : 20;
        printf("# M %d N %d K %d\n", M, N, K);
#define RUN(F, P3, P0, P1, P2, BP) run_variant<(F), P3, P0, P1, P2, BP>(#F " P " #P3 " " #P0 " " #P1 " " #P2 " bar " #BP, A, B, C, M, N, K, ref, err, sum, reps)
#define RUNV(F, P3, P0, P1, P2, BP) run_vgpr<(F), P3, P0, P1, P2, BP>("vgpr " #F " P " #P3 " " #P0 " " #P1 " " #P2 " bar " #BP, A, B, C, M, N, K, ref, err, sum, reps)
#define RUNR(F, H1, BP) run_ring<(F), H1, BP>("ring " #F " H1 " #H1 " bar " #BP, A, B, C, M, N, K, ref, err, sum, reps)
        for (int rep = 0; rep < (quick ? 1 : 2); ++rep) {
            RUN(F_CLOCK | F_GM4 | F_BUF, 6, 6, 4, 0, 4);
            RUN(F_CLOCK | F_GM4 | F_BUF, 4, 4, 4, 4, 4);
            RUN(F_CLOCK | F_GM4 | F_BUF, 4, 6, 6, 0, 8);
            RUN(F_CLOCK | F_GM4 | F_BUF | F_DMA_LATE, 6, 6, 4, 0, 4);
            RUN(F_CLOCK | F_GM4 | F_BUF | F_NOREAD, 6, 6, 4, 0, 4);
            RUN(F_CLOCK | F_GM4 | F_BUF | F_NOMFMA, 6, 6, 4, 0, 4);
            RUNV(F_CLOCK | F_GM4, 4, 4, 4, 4, 4);
            RUNV(F_CLOCK | F_GM4, 4, 4, 4, 4, 0);
            RUNV(F_CLOCK | F_GM4, 4, 4, 4, 4, 8);
            RUNV(F_CLOCK | F_GM4, 2, 6, 4, 4, 4);
            RUNV(F_CLOCK | F_GM4, 6, 6, 4, 0, 4);
            RUNV(F_CLOCK | F_GM4 | F_DMA_LATE, 4, 4, 4, 4, 4);
            RUNV(F_CLOCK | F_GM4 | F_NOFENCE, 4, 4, 4, 4, 4);
            RUNV(F_CLOCK | F_GM4 | F_NOREAD, 4, 4, 4, 4, 4);
            RUNV(F_CLOCK | F_GM4 | F_NOMFMA, 4, 4, 4, 4, 4);
            RUNV(F_CLOCK | F_GM4 | F_NODMA, 4, 4, 4, 4, 4);
            RUNR(F_CLOCK | F_GM4, 4, 4);
            RUNR(F_CLOCK | F_GM4, 4, 0);
            RUNR(F_CLOCK | F_GM4, 4, 8);
            RUNR(F_CLOCK | F_GM4, 6, 4);
            RUNR(F_CLOCK | F_GM4, 2, 4);
            RUNR(F_CLOCK | F_GM4 | F_DMA_LATE, 4, 4);
            RUNR(F_CLOCK | F_GM4 | F_NOREAD, 4, 4);
            RUNR(F_CLOCK | F_GM4 | F_NOMFMA, 4, 4);
            RUNR(F_CLOCK | F_GM4 | F_NODMA, 4, 4);
            RUN(F_CLOCK, 6, 6, 4, 0, 4);
            RUN(F_CLOCK | F_GM4, 6, 6, 4, 0, 4);
            RUN(F_CLOCK | F_GM8, 6, 6, 4, 0, 4);
            RUN(F_CLOCK | F_GM16, 6, 6, 4, 0, 4);
            RUN(F_CLOCK | F_GM8 | F_NOREAD, 6, 6, 4, 0, 4);
            RUN(F_CLOCK | F_GM8 | F_NOMFMA, 6, 6, 4, 0, 4);
            RUN(F_CLOCK | F_GM8 | F_PRIO, 6, 6, 4, 0, 4);
            RUN(F_CLOCK | F_GM8, 4, 4, 4, 4, 0);
            RUN(F_CLOCK | F_GM8, 6, 6, 4, 0, 0);
            RUN(F_CLOCK | F_GM8, 4, 6, 6, 0, 8);
            RUN(F_CLOCK | F_NODMA, 6, 6, 4, 0, 4);
            RUN(F_CLOCK | F_NOREAD, 6, 6, 4, 0, 4);
            RUN(F_CLOCK | F_NODMA | F_NOREAD, 6, 6, 4, 0, 4);
            RUN(F_CLOCK | F_NOMFMA, 6, 6, 4, 0, 4);
            RUN(0, 4, 4, 4, 4, 0);
            RUN(0, 6, 6, 4, 0, 0);
            RUN(0, 6, 6, 4, 0, 4);
            RUN(0, 4, 6, 6, 0, 8);
            RUN(F_DMA_LATE, 6, 6, 4, 0, 4);
            RUN(F_READS_FIRST, 6, 6, 4, 0, 4);
            RUN(F_READS_FIRST | F_DMA_LATE, 6, 6, 4, 0, 4);
            RUN(F_NOFENCE, 6, 6, 4, 0, 4);
            RUN(F_PRIO, 6, 6, 4, 0, 4);
            RUN(F_NODMA, 6, 6, 4, 0, 4);
            RUN(F_NOREAD, 6, 6, 4, 0, 4);
            RUN(F_NODMA | F_NOREAD, 6, 6, 4, 0, 4);
            RUN(F_NOMFMA, 6, 6, 4, 0, 4);
        }
#undef RUN
    }
    return 0;
}
