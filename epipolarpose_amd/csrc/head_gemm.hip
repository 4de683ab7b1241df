// bf16 MFMA implicit-GEMM for the deconvolution head (gfx950).
//
// One kernel template computes  C[m][n] = sum_k A(m,k) * Bt[n][k]  with fp32 accumulation on
// v_mfma_f32_32x32x16_bf16, where the A operand is either a plain row-major matrix or an implicit
// (never materialised) convolution gather over an NHWC activation tensor:
//   * ConvTranspose2d(k=4,s=2,p=1) forward  (pose3d_resnet.py:158-183) = 4 output-parity phases, each a
//     2x2-tap gather GEMM with K = 4*Cin;
//   * its backward-data = a 4x4 stride-2 gather GEMM with K = 16*Cout;
//   * the final 1x1 convolution (pose3d_resnet.py:116-122) forward / backward-data = plain GEMMs (the forward, short K and
//     wide N, through the A-stationary kernel further down).
// Tile 128x128x64, 256 threads (2x2 waves, each 2x2 MFMA 32x32 tiles), global->LDS by direct DMA
// (global_load_lds_dwordx4: no VGPR round trip, no ds_write pass) with the next K tile's DMA in flight during the
// MFMAs, double-buffered LDS (64 KiB -> 2 workgroups / CU), 16-byte chunks XOR-swizzled on the SOURCE address so the
// lane-linear DMA image is conflict-free for the 16-lane ds_read_b128 groups.
#include "common.h"
#include "bn_affine.h"
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <unordered_map>
#include <utility>
#include <vector>

namespace epi {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GBK = 64;                            // K (reduction) depth of one staged tile, every kernel

// NT kernel tile configurations: WM x WN waves, each wave TM x 2 MFMA tiles (32x32): tile (WM*TM*32) x (WN*64).
//   small: 128 x 128, 4 waves (2 x 2), 64 KiB LDS (2 workgroups / CU)  -- short K, few rows, ragged N
//   tall:  256 x  64, 4 waves (4 x 1), 80 KiB LDS (2 workgroups / CU)  -- N <= 64 (the 64-channel layers): no half-empty MFMA tiles
//   big:   256 x 256, 8 waves (2 x 4), 128 KiB LDS (1 workgroup / CU)  -- twice the MFMA work per staged byte and per DMA
//          instruction, 1.5x less LDS read traffic per MFMA
template <int TM_, int WM_, int WN_, int MIN_WAVES_, int NSTAGE_, bool STAGGER_ = false> struct GemmCfg {
    static constexpr int TM = TM_, WM = WM_, WN = WN_, NW = WM_ * WN_, THREADS = 64 * NW, MIN_WAVES = MIN_WAVES_, NSTAGE = NSTAGE_;
    static constexpr bool STAGGER = STAGGER_;                      // the two-wave-group K loop (see head_gemm_kernel)
    static constexpr int BM = WM_ * TM_ * 32, BN = WN_ * 64;
    static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int AP = BM / 8 / NW, BP = BN / 8 / NW;       // 1 KiB DMA pieces (8 rows x 128 B) per wave and K tile
    static constexpr bool EARLY = NW == 4 && NSTAGE_ == 2;         // 2-stage 4-wave tiles: issue the next K tile's DMA at the top of the current one
    static_assert(AP >= 1 && BP >= 1 && BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "whole pieces per wave");
    static_assert(NSTAGE_ >= 2 && NSTAGE_ <= 4, "LDS ring depth");
};
typedef GemmCfg<2, 2, 2, 2, 2> CfgSmall;
typedef GemmCfg<2, 4, 1, 2, 2> CfgTall;
typedef GemmCfg<4, 2, 4, 1, 2> CfgBigLock;                        // the lock-step two-stage loop of rounds 1-4 (epi_gemm_tune tile 5: A/B only)
typedef GemmCfg<4, 2, 4, 1, 2, true> CfgBig;                      // round 5: two wave groups half a K-tile phase apart
// Under-filled launches (the deep layers at batch 32: 8192 or 2048 rows, 64 .. 512 tiles of 128 x 128): a workgroup that owns a CU alone
// fills LDS at ~30 B/clk (tools/probe_fill.hip: 4 waves issuing 1 KiB loads), two per CU at ~51 B/clk, and a CU without a workgroup at
// nothing -- so the same problem cut into twice or four times as many, smaller tiles finishes sooner although it stages more bytes.
//   half:    64 x 128, 4 waves (2 x 2, one MFMA tile row each), 48 KiB LDS (3 workgroups / CU)
//   quarter: 64 x  64, 2 waves (2 x 1),                         32 KiB LDS (5 workgroups / CU)
typedef GemmCfg<1, 2, 2, 3, 2> CfgHalf;
typedef GemmCfg<1, 2, 1, 3, 2> CfgQuarter;
// Pipelined variants: a ring of 4 (3) LDS stages owned by ONE workgroup per CU; the DMA of K tile kt+3 (kt+2) is issued while tile
// kt computes and is waited for with a COUNTED vmcnt, so a load has three (two) tile times to land instead of one.  For the
// convolution GEMMs of this network (a few hundred workgroups, 9 .. 72 K tiles each) the 2-stage loop spends most of every K
// tile waiting for its DMA at the closing barrier (~1.0-1.4 us per tile against 0.21 us of MFMA work).
typedef GemmCfg<2, 2, 2, 1, 4> CfgSmallP;
typedef GemmCfg<2, 4, 1, 1, 3> CfgTallP;

struct GemmGather {        // maps GEMM row m / K tile to an NHWC source pixel
    int enabled;           // 0: plain A[m*lda + k]
    int Hg, Wg;            // rows enumerate (n, i, j) over an Hg x Wg grid
    int Hs, Ws, Cs;        // source tensor [n][Hs][Ws][Cs]
    int stride;            // source pixel = (i*stride + dy[tap], j*stride + dx[tap]); k = tap*Cs + c
    int dy[16], dx[16];
    int pitch;             // elements between consecutive source pixels; 0 = Cs.  pitch < Cs: a tap reads Cs CONTIGUOUS elements that span
                           // several pixels (the space-to-depth stem: 4 pixels x 16 channels per tap)
};
struct GemmScatter {       // maps GEMM row m to an output row
    int enabled;           // 0: row m
    int Hg, Wg;            // same (n, i, j) enumeration
    int Ho, Wo;            // output tensor [n][Ho][Wo][ldc]
    int so, oy, ox;        // output pixel = (i*so + oy, j*so + ox)
};
// Output-parity phases of a stride-2 transposed convolution (ConvTranspose2d forward, or the backward-data of a stride-2
// Conv2d): blockIdx.z = phase 2*py + px owns the output pixels (2i + py, 2j + px); only the taps whose stride lands on that
// parity contribute, so every phase is its own gather GEMM with its own tap list, K extent and weight block.
struct GemmPhases {
    int enabled;
    int ntap[4];           // taps of phase p (0 .. 4); K of the phase = ntap * ga.Cs
    int dy[4][4], dx[4][4];   // source pixel of tap t = (i + dy, j + dx)
    long long bt_off[4];   // element offset of the phase's weight block [N][ntap * Cs] inside Bt
};
// BatchNorm-backward reduction fused into a backward-data epilogue.  The gradient this GEMM produces is the dy of a training-mode
// BatchNorm (+ ReLU) whose backward pass starts with sum(dz) and sum(dz * xhat) over all rows, dz = dy * [output > 0]: the epilogue
// has the bf16-rounded dy tile in registers, so it fetches the same tile of z (the raw convolution output that layer normalised) and
// of the mask source, writes dz INSTEAD of dy and adds the two column sums of its tile to `sums` -- the separate reduction pass over
// (dy, z, y) disappears and the apply pass needs neither the mask source nor a second output for the shortcut branch (dz is it).
struct GemmBnRed {
    const unsigned short* z;   // [rows / stride as C] raw forward output of the normalised layer; null: off
    const unsigned short* y;   // mask source: the gradient passes where y > 0 (residual + ReLU layers); null: where scale*z + shift > 0
    const float* bn;           // [mean | rstd | scale | shift], N floats each
    float* sums;               // [2N] += (sum dz | sum dz * xhat), xhat = (z - mean) * rstd
    int relu;                  // 0: no mask at all
};

struct GemmArgs {
    const unsigned short* A;
    const unsigned short* Bt;
    void* C;
    const float* bias;     // per output column n, or null
    int M, N, K, lda, ldb, ldc;
    GemmGather ga;
    GemmScatter sc;
    GemmPhases ph;         // enabled: K = the LARGEST phase's extent (split-K planning); ldb is per phase ntap * Cs
    int k_per_split;       // split-K: blockIdx.y handles K range [y*k_per_split, ...) and writes an fp32 slab (plain rows)
    float* slabs;          // [nsplit][nphase][M][N] when gridDim.y > 1
    const unsigned short* addend;   // bf16 [same rows / stride as C] or null: C = A*B + addend (the other branch's gradient at a residual junction)
    float* stats;          // [stats_copies][2N] or null: += per-column (sum, sum of squares) of the bf16 result (unsplit bf16 launches only);
    int stats_copies;      //   M tile t adds into copy t % stats_copies
    int coalesce;          // bf16 result with N % 8 == 0, ldc % 8 == 0: LDS-staged 128-byte row segments
    int patch_px;          // conv_patch_kernel: pixels of the staged patch (256 + 2W + 2, rounded up to 8)
    int chunks_per_split;  // conv_patch_kernel: 64-channel chunks per blockIdx.y
    GemmBnRed br;          // unsplit bf16 launches on the coalesced epilogue only (launch_gemm decides and reports)
    int addend_step;       // 2: `addend` is [B][add_h / 2][add_w / 2][ldc] and holds the contribution of the EVEN pixels of the [B][add_h][add_w] output only
    int add_h, add_w;      //    (the shortcut gradient through a 1x1 stride-2 projection: every other pixel receives none); plain rows, coalesced epilogue only
    int store_policy;      // coalesced epilogue's result stores: 0 plain, 1 non-temporal (nt) -- epi_gemm_store_policy
    // BatchNorm (+ ReLU) of the A operand, applied on the fly (BNIN instantiation, plain rows): A holds the RAW output z of the convolution in front
    // and this launch computes C = relu(z * scale[k] + shift[k]) x Bt -- the normalised tensor is never written (round 4: one apply launch per
    // bottleneck interior less).  Every workgroup derives (scale, shift) of all K channels from the batch sums into LDS (bn_derive); workgroup 0 also
    // writes the saved statistics, updates the running estimates and clears the layer's backward accumulator, as the apply kernel's lead does.
    int bn_in_on;
    BnAffine bn_in;
};

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// XCD-aware workgroup order (8 XCDs, each with a private 4 MiB L2; the dispatcher is observed to place linear workgroup
// id b on XCD b % 8).  The bijective remap gives every XCD a CONTIGUOUS range of logical ids, so tiles that share an
// operand panel (consecutive logical ids) hit the same L2 instead of re-fetching the panel from HBM on 8 different XCDs
// (PMC: the final 1x1 conv read A 8.8x before this remap).  Placement only affects speed, never results.
__device__ __forceinline__ int xcd_remap(int lin, int total) {
    const int q = total >> 3, r = total & 7, xcd = lin & 7, idx = lin >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// 16 zero bytes in global memory: the source of every out-of-range / padding chunk of a direct-to-LDS load
__device__ uint4v epi_zero_chunk[1];

// global -> LDS DMA (global_load_lds_dwordx4): each lane supplies its own 16-byte global source; the 64 lanes of the wave
// land contiguously at lds_base + 16*lane (lds_base must be wave-uniform).  No VGPR round trip, no ds_write pass.
__device__ __forceinline__ void glds16(const void* gsrc, char* lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}

// eight bf16 values + eight bf16 values, in fp32, rounded back (the residual-junction add of the backward pass)
__device__ __forceinline__ uint4v add_bf16x8(uint4v a, uint4v b) {
    const unsigned int aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
    unsigned int o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        o[k] = pack_bf16x2(__uint_as_float(aw[k] << 16) + __uint_as_float(bw[k] << 16),
                           __uint_as_float(aw[k] & 0xffff0000u) + __uint_as_float(bw[k] & 0xffff0000u));
    uint4v r; r.x = o[0]; r.y = o[1]; r.z = o[2]; r.w = o[3];
    return r;
}

// ---- GemmBnRed: per-lane state of the eight columns n .. n + 7 a lane copies in the coalesced epilogues ----
struct BnRedCols { float mu[8], rs[8], sc[8], sh[8]; };
__device__ __forceinline__ void bnred_cols(const GemmBnRed& br, int N, int n, BnRedCols& c) {
    const bool ok = n < N;                       // (N % 8 == 0 on this path: all eight columns or none)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float4v zero = float4v{0.f, 0.f, 0.f, 0.f};
        const float4v mu = ok ? *reinterpret_cast<const float4v*>(br.bn + n + 4 * h) : zero;
        const float4v rs = ok ? *reinterpret_cast<const float4v*>(br.bn + N + n + 4 * h) : zero;
        const float4v sc = ok ? *reinterpret_cast<const float4v*>(br.bn + 2 * N + n + 4 * h) : zero;
        const float4v sh = ok ? *reinterpret_cast<const float4v*>(br.bn + 3 * N + n + 4 * h) : zero;
        c.mu[4 * h] = mu.x; c.mu[4 * h + 1] = mu.y; c.mu[4 * h + 2] = mu.z; c.mu[4 * h + 3] = mu.w;
        c.rs[4 * h] = rs.x; c.rs[4 * h + 1] = rs.y; c.rs[4 * h + 2] = rs.z; c.rs[4 * h + 3] = rs.w;
        c.sc[4 * h] = sc.x; c.sc[4 * h + 1] = sc.y; c.sc[4 * h + 2] = sc.z; c.sc[4 * h + 3] = sc.w;
        c.sh[4 * h] = sh.x; c.sh[4 * h + 1] = sh.y; c.sh[4 * h + 2] = sh.z; c.sh[4 * h + 3] = sh.w;
    }
}
// o: eight bf16 gradients of one row (element `off` of the output onwards).  Returns dz (o with the masked elements cleared) and
// accumulates s += dz, q += dz * xhat.  Same arithmetic as bn_bwd_reduce_kernel (csrc/bn_nhwc.hip) on the same bf16-rounded values.
__device__ __forceinline__ uint4v bnred_apply(const GemmBnRed& br, uint4v o, uint4v zv, uint4v yv, const BnRedCols& c, float (&s)[8], float (&q)[8]);
__device__ __forceinline__ uint4v bnred_row(const GemmBnRed& br, uint4v o, long long off, const BnRedCols& c, float (&s)[8], float (&q)[8]) {
    const uint4v zv = *reinterpret_cast<const uint4v*>(br.z + off);
    uint4v yv = uint4v{0u, 0u, 0u, 0u};
    if (br.relu && br.y) yv = *reinterpret_cast<const uint4v*>(br.y + off);
    return bnred_apply(br, o, zv, yv, c, s, q);
}
// (zv, yv: the z / y elements of the same positions, fetched by the caller -- ahead of time where it has the registers)
__device__ __forceinline__ uint4v bnred_apply(const GemmBnRed& br, uint4v o, uint4v zv, uint4v yv, const BnRedCols& c, float (&s)[8], float (&q)[8]) {
    unsigned int ow[4] = {o.x, o.y, o.z, o.w};
    const unsigned int zw[4] = {zv.x, zv.y, zv.z, zv.w}, yw[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int e = 2 * k + h;
            const unsigned int keep = h ? 0xffff0000u : 0x0000ffffu;
            const float g = __uint_as_float(h ? (ow[k] & 0xffff0000u) : (ow[k] << 16));
            const float xv = __uint_as_float(h ? (zw[k] & 0xffff0000u) : (zw[k] << 16));
            bool on = true;
            if (br.relu) on = br.y ? (__uint_as_float(h ? (yw[k] & 0xffff0000u) : (yw[k] << 16)) > 0.f) : (xv * c.sc[e] + c.sh[e] > 0.f);
            const float dz = on ? g : 0.f;
            s[e] += dz;
            q[e] = fmaf(dz, (xv - c.mu[e]) * c.rs[e], q[e]);
            if (!on) ow[k] &= ~keep;
        }
    }
    uint4v r; r.x = ow[0]; r.y = ow[1]; r.z = ow[2]; r.w = ow[3];
    return r;
}

// -DEPI_GEMM_TRACE (tools/gemm_lab.hip's trace build only): wave 0 of every workgroup stamps s_memtime at its phase boundaries
//   0 entry | 1 s_memrealtime at entry | 2 staging roles computed | 3 first K tile landed | 4 K loop done | 5 tile parked in LDS |
//   6 stores issued | 7 stores drained (+ statistics) | 8 s_memrealtime at exit | 9 HW_ID | 10 XCC_ID | 11 K tiles
#ifdef EPI_GEMM_TRACE
constexpr int GEMM_TRACE_WGS = 8192, GEMM_TRACE_SLOTS = 16;
__device__ unsigned long long epi_gemm_trace[GEMM_TRACE_WGS * GEMM_TRACE_SLOTS];
#define GEMM_STAMP_V(i, v)                                                                                               \
    do {                                                                                                                 \
        const int wg_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                                  \
        if (threadIdx.x == 0 && wg_ < GEMM_TRACE_WGS) epi_gemm_trace[wg_ * GEMM_TRACE_SLOTS + (i)] = (v);               \
    } while (0)
#define GEMM_STAMP(i) GEMM_STAMP_V(i, __builtin_amdgcn_s_memtime())
#else
#define GEMM_STAMP_V(i, v) do {} while (0)
#define GEMM_STAMP(i) do {} while (0)
#endif

// ---------------------------------------------------------------------------------------------------------------
// Coalesced bf16 epilogue shared by head_gemm_kernel and conv_patch_kernel (N % 8 == 0, 16-byte aligned rows).
// The wave parks its (TM*32) x 64 bf16 sub-tile in its own slice of the (now idle) staging LDS -- 8-byte units XOR-swizzled by (row & 15) so
// the 16 rows of a ds_write_b64 lane group hit 16 different bank pairs -- and writes it out as whole 128-byte row segments (8 lanes x 16 B per
// row, 8 rows per instruction) instead of 8-byte pieces of 32 different rows.  The caller's last barrier separated the final fragment reads from
// these writes; afterwards each wave only touches its own region (LDS ops of a wave are ordered).
// Optional, all decided by WAVE-UNIFORM flags that are tested ONCE (round 4: the round-3 form tested p.bias per element and the row mapping /
// addend / statistics flags per copied row -- ~300 branches per wave, 3.3 us of a workgroup's life on the phase trace, tools/gemm_trace.hip):
//   * bias: added to the accumulators before the park;
//   * residual-junction addend: requested before the park, consumed after the LDS round trip;
//   * BatchNorm statistics of the bf16-ROUNDED outputs (p.stats: the convolution feeds a training-mode BatchNorm): 8 columns per lane over the
//     rows it copies, reduced over the 8 row lanes with three shuffles, combined over the workgroup's wave rows through LDS, then ONE contiguous
//     fp32 atomic per column and workgroup (what serialises in L2 is the number of (instruction, 128-byte line) pairs per line);
//   * RED (compile time): the fused BatchNorm-backward reduction (GemmBnRed), whose two sums take the statistics' route.
// ---------------------------------------------------------------------------------------------------------------
// (the fields of GemmArgs the epilogue reads, BY VALUE: a reference to the kernel argument itself would make the compiler copy all of it to
//  scratch in the instantiations that also index its tap tables dynamically)
struct EpiArgs {
    void* C;
    const float* bias;
    const unsigned short* addend;
    float* stats;
    int M, N, ldc, stats_copies, addend_step, add_h, add_w, store_policy;
    GemmScatter sc;
    GemmBnRed br;
};
#define EPI_ARGS_OF(p) EpiArgs{(p).C, (p).bias, (p).addend, (p).stats, (p).M, (p).N, (p).ldc, (p).stats_copies, (p).addend_step, (p).add_h, (p).add_w, (p).store_policy, (p).sc, (p).br}
template <int TM, int WM, int WN, int THREADS, bool RED, bool RED_PREFETCH = true>
__device__ __forceinline__ void coalesced_epilogue(const EpiArgs p, f32x16 (&acc)[TM][2], char* smem, int m0, int n0, int tile_m, int wm, int wn, int wid,
                                                   int lane, int tid, int sc_oy, int sc_ox) {
    constexpr int GBN = WN * 64, ROWS = TM * 4;
    const int frow = lane & 31, fhalf = lane >> 5, c8 = lane & 7, r8 = lane >> 3;
    char* mine = smem + wid * (TM * 32 * 128);
    const int n = n0 + wn * 64 + c8 * 8;
    const int m_first = m0 + wm * (TM * 32) + r8;          // the row this lane copies in read-back iteration `it` is m_first + 8 * it
    const bool col_ok = n < p.N;
    const bool scatter = p.sc.enabled != 0, has_add = p.addend != nullptr, half_add = p.addend_step == 2, nt_stores = p.store_policy == 1;
    // element offset of (copied row of iteration it, column n) in C (and in the addend / z / y, which share C's row mapping); ok: inside the matrix
    auto row_off = [&](int it, bool& ok, auto sc_c) __attribute__((always_inline)) -> long long {
        const int m = m_first + 8 * it;
        ok = col_ok && m < p.M;
        if (!decltype(sc_c)::value) return (long long)m * p.ldc + n;
        const int hw = p.sc.Hg * p.sc.Wg;
        const int b = m / hw, rem = m - b * hw;
        const int i = rem / p.sc.Wg, j = rem - i * p.sc.Wg;
        return (((long long)b * p.sc.Ho + i * p.sc.so + sc_oy) * p.sc.Wo + j * p.sc.so + sc_ox) * p.ldc + n;
    };
    // the residual-junction addend of a plain row m (addend_step 2: the half-resolution tensor that holds the even pixels only)
    auto addend_at = [&](int m, long long off) __attribute__((always_inline)) -> uint4v {
        if (!half_add) return *reinterpret_cast<const uint4v*>(p.addend + off);
        const int hw = p.add_h * p.add_w;
        const int b = m / hw, rem = m - b * hw;
        const int yy = rem / p.add_w, xx = rem - yy * p.add_w;
        if ((yy | xx) & 1) return uint4v{0u, 0u, 0u, 0u};
        const long long ar = ((long long)b * (p.add_h >> 1) + (yy >> 1)) * (p.add_w >> 1) + (xx >> 1);
        return *reinterpret_cast<const uint4v*>(p.addend + ar * p.ldc + n);
    };
    constexpr bool AD_EARLY = !(RED && (TM >= 4 || !RED_PREFETCH));   // (the 256-row tile / the patch kernel with the fused reduction: no registers for the addend rows)
    constexpr int RED_GROUP = 4;        // rows' z / y tiles in flight per lane where they are fetched inside the copy loop (8 for the 256-row tile: no gain, 6.355 vs 6.339 ms)
    constexpr bool RED_EARLY = RED && RED_PREFETCH && TM <= 2;   // z / y of all rows requested before the park too: their latency hides behind it
                                                                 // (RED_PREFETCH = false: the patch kernel's register budget has no room for them)
    uint4v ad[AD_EARLY ? ROWS : 1];
    uint4v zpre[RED_EARLY ? ROWS : 1], ypre[RED_EARLY ? ROWS : 1];
    auto prefetch = [&](auto sc_c) __attribute__((always_inline)) {
        if (AD_EARLY && has_add) {
#pragma unroll
            for (int it = 0; it < ROWS; ++it) {
                bool ok;
                const long long off = row_off(it, ok, sc_c);
                ad[it] = ok ? addend_at(m_first + 8 * it, off) : uint4v{0u, 0u, 0u, 0u};
            }
        }
        if (RED_EARLY) {
            const bool use_y = p.br.relu && p.br.y;
#pragma unroll
            for (int it = 0; it < ROWS; ++it) {
                bool ok;
                const long long off = row_off(it, ok, sc_c);
                zpre[it] = ok ? *reinterpret_cast<const uint4v*>(p.br.z + off) : uint4v{0u, 0u, 0u, 0u};
                ypre[it] = (ok && use_y) ? *reinterpret_cast<const uint4v*>(p.br.y + off) : uint4v{0u, 0u, 0u, 0u};
            }
        }
    };
    if (RED_EARLY || (AD_EARLY && has_add)) { if (scatter) prefetch(std::true_type{}); else prefetch(std::false_type{}); }
    // ---- park: lane holds, for tile (ti, tj): row ti*32 + (lane & 31) of the wave's sub-tile, columns tj*32 + 8*q + 4*(lane >> 5) + e (reg 4*q + e).
    //      Two compile-time copies (with / without the bias add: rare, the head's final layer when it leaves the A-stationary kernel); the
    //      accumulators are only READ -- a conditional in-place update would keep a second copy of all of them live ----
    auto park = [&](auto bias_c) __attribute__((always_inline)) {
        constexpr bool BIAS = decltype(bias_c)::value;
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float b[4] = {0.f, 0.f, 0.f, 0.f};
                if (BIAS) {
                    const int nq = n0 + wn * 64 + tj * 32 + 8 * q + 4 * fhalf;
#pragma unroll
                    for (int e = 0; e < 4; ++e) b[e] = nq + e < p.N ? p.bias[nq + e] : 0.f;
                }
#pragma unroll
                for (int ti = 0; ti < TM; ++ti) {
                    const int row = ti * 32 + frow;
                    uint2 t;
                    if (BIAS) {
                        t.x = pack_bf16x2(acc[ti][tj][4 * q] + b[0], acc[ti][tj][4 * q + 1] + b[1]);
                        t.y = pack_bf16x2(acc[ti][tj][4 * q + 2] + b[2], acc[ti][tj][4 * q + 3] + b[3]);
                    } else {
                        t.x = pack_bf16x2(acc[ti][tj][4 * q], acc[ti][tj][4 * q + 1]);
                        t.y = pack_bf16x2(acc[ti][tj][4 * q + 2], acc[ti][tj][4 * q + 3]);
                    }
                    const int unit = tj * 8 + 2 * q + fhalf;
                    *reinterpret_cast<uint2*>(mine + row * 128 + ((unit ^ (row & 15)) << 3)) = t;
                }
            }
    };
    if (p.bias) park(std::true_type{}); else park(std::false_type{});
    GEMM_STAMP(5);
    float ssum[8], ssq[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { ssum[k] = 0.f; ssq[k] = 0.f; }
    BnRedCols rcols;
    if (RED) bnred_cols(p.br, p.N, n, rcols);
    // ---- read back + store; STATS / ADD are compile-time copies of the loop, selected once ----
    auto copy_rows = [&](auto stats_c, auto add_c, auto sc_c) __attribute__((always_inline)) {
        constexpr bool STATS = decltype(stats_c)::value, ADD = decltype(add_c)::value;
#pragma unroll
        for (int it = 0; it < ROWS; ++it) {
            // (RED: at most four rows' z / y tiles in flight per lane -- the scheduler would otherwise hoist all TM*4 loads to the top)
            if (RED && !RED_EARLY && it % RED_GROUP == 0 && it) __builtin_amdgcn_sched_barrier(0);
            const int row = it * 8 + r8, sw = row & 15;
            const uint2 lo = *reinterpret_cast<const uint2*>(mine + row * 128 + (((2 * c8) ^ sw) << 3));
            const uint2 hi = *reinterpret_cast<const uint2*>(mine + row * 128 + (((2 * c8 + 1) ^ sw) << 3));
            bool ok;
            const long long off = row_off(it, ok, sc_c);
            if (!ok) continue;
            if (STATS) {
                const unsigned int w4[4] = {lo.x, lo.y, hi.x, hi.y};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float a = __uint_as_float(w4[k] << 16), b = __uint_as_float(w4[k] & 0xffff0000u);
                    ssum[2 * k] += a; ssq[2 * k] = fmaf(a, a, ssq[2 * k]);
                    ssum[2 * k + 1] += b; ssq[2 * k + 1] = fmaf(b, b, ssq[2 * k + 1]);
                }
            }
            uint4v o; o.x = lo.x; o.y = lo.y; o.z = hi.x; o.w = hi.y;
            if (ADD) o = add_bf16x8(o, AD_EARLY ? ad[it] : addend_at(m_first + 8 * it, off));
            if (RED) o = RED_EARLY ? bnred_apply(p.br, o, zpre[it], ypre[it], rcols, ssum, ssq) : bnred_row(p.br, o, off, rcols, ssum, ssq);
            uint4v* dst = reinterpret_cast<uint4v*>(reinterpret_cast<unsigned short*>(p.C) + off);
            if (nt_stores) __builtin_nontemporal_store(o, dst);
            else *dst = o;
        }
    };
    const bool has_stats = !RED && p.stats != nullptr;
    auto copy_sc = [&](auto sc_c) __attribute__((always_inline)) {
        if (has_stats) { if (has_add) copy_rows(std::true_type{}, std::true_type{}, sc_c); else copy_rows(std::true_type{}, std::false_type{}, sc_c); }
        else if (has_add) copy_rows(std::false_type{}, std::true_type{}, sc_c);
        else copy_rows(std::false_type{}, std::false_type{}, sc_c);
    };
    if (scatter) copy_sc(std::true_type{}); else copy_sc(std::false_type{});
    GEMM_STAMP(6);
    if (has_stats || RED) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) { ssum[k] += __shfl_xor(ssum[k], o, 64); ssq[k] += __shfl_xor(ssq[k], o, 64); }
        }
        float* st = reinterpret_cast<float*>(smem + WM * WN * (TM * 32 * 128));    // [WM][sum | sq][GBN], behind the waves' park slices
        if (lane < 8) {
            float* row = st + wm * (2 * GBN) + wn * 64 + c8 * 8;
            *reinterpret_cast<float4v*>(row) = float4v{ssum[0], ssum[1], ssum[2], ssum[3]};
            *reinterpret_cast<float4v*>(row + 4) = float4v{ssum[4], ssum[5], ssum[6], ssum[7]};
            *reinterpret_cast<float4v*>(row + GBN) = float4v{ssq[0], ssq[1], ssq[2], ssq[3]};
            *reinterpret_cast<float4v*>(row + GBN + 4) = float4v{ssq[4], ssq[5], ssq[6], ssq[7]};
        }
        __syncthreads();
        float* dst = RED ? p.br.sums : p.stats + (long long)(tile_m % p.stats_copies) * 2 * p.N;
        for (int t = tid; t < 2 * GBN; t += THREADS) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) v += st[w * (2 * GBN) + t];
            const int which = t / GBN, col = n0 + (t % GBN);
            if (col < p.N) atomicAdd(dst + which * p.N + col, v);
        }
    }
#ifdef EPI_GEMM_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GEMM_STAMP(7);
    GEMM_STAMP_V(8, __builtin_amdgcn_s_memrealtime());
#endif
}

enum { A_PLAIN = 0, A_GATHER = 1, A_PHASED = 2 };     // how the A operand's rows are addressed (compile-time: keeps the K loop branch-free)


// RED: the instantiation with the fused BatchNorm-backward reduction (GemmBnRed) in the coalesced epilogue -- a separate one, so that the
// registers its epilogue needs (column parameters, the z / y tiles in flight) never enter the allocation of the other launches
// the A fragment of eight consecutive k (one lane's operand of a 32x32x16 MFMA) through relu(z * scale + shift); tab: (scale, shift) pairs of those k
__device__ __forceinline__ bf16x8 bn_in_apply(bf16x8 a, const float4v (&t)[4]) {
    const uint4v u = __builtin_bit_cast(uint4v, a);
    const unsigned int w[4] = {u.x, u.y, u.z, u.w};
    unsigned int o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float lo = __uint_as_float(w[k] << 16), hi = __uint_as_float(w[k] & 0xffff0000u);
        o[k] = pack_bf16x2(fmaxf(lo * t[k].x + t[k].y, 0.f), fmaxf(hi * t[k].z + t[k].w, 0.f));      // (the arithmetic of bn_apply2d_kernel)
    }
    uint4v r; r.x = o[0]; r.y = o[1]; r.z = o[2]; r.w = o[3];
    return __builtin_bit_cast(bf16x8, r);
}

template <bool OUT_F32, typename Cfg, int MODE, bool RED = false, bool BNIN = false>
__global__ __launch_bounds__(Cfg::THREADS, Cfg::MIN_WAVES) void head_gemm_kernel(GemmArgs p) {
    static_assert(!BNIN || (MODE == A_PLAIN && Cfg::NSTAGE == 2), "BatchNorm on the A operand: plain rows, two-stage loop");
    constexpr int GBM = Cfg::BM, GBN = Cfg::BN, TM = Cfg::TM, AP = Cfg::AP, BP = Cfg::BP;
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [2 buffers][A tile | B tile]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    GEMM_STAMP(0);
    GEMM_STAMP_V(1, __builtin_amdgcn_s_memrealtime());
    GEMM_STAMP_V(9, (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4));
    GEMM_STAMP_V(10, (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20));
    const int wm = wid / Cfg::WN, wn = wid % Cfg::WN;
    const int tiles_n = (p.N + GBN - 1) / GBN;
    // logical id: tile_n fastest, then tile_m, then split, then phase -> neighbours share the A panel (and B)
    const int total_wg = gridDim.x * gridDim.y * gridDim.z;
    int lid = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), total_wg);
    const int tile_id = lid % (int)gridDim.x;
    lid /= (int)gridDim.x;
    const int split_id = lid % (int)gridDim.y, phase = lid / (int)gridDim.y;
    int tile_m = tile_id / tiles_n, tile_n = tile_id - tile_m * tiles_n;
    if constexpr (GBM == 256 && GBN == 256) {
        // One workgroup per CU: the 32 tiles an XCD runs at once are 32 CONSECUTIVE ids.  As a 1 x 32 strip they fetch 1 + 32 operand panels per K tile
        // through that XCD's L2 (hit rate 48 %); in groups of 4 tile rows, column-major inside a group, they are a 4 x 8 block: 12 panels (81 %).
        // Measured (tools/w4_lab.hip, 8192^3): the operand feed alone 1.53 -> 1.18 us per K tile, the GEMM +10 .. 13 %.
        constexpr int GM = 4;
        const int tiles_m = (p.M + GBM - 1) / GBM, per_group = GM * tiles_n, g = tile_id / per_group, i = tile_id - g * per_group;
        const int rows = min(GM, tiles_m - g * GM);
        tile_m = g * GM + i % rows;
        tile_n = i / rows;
    }
    const int m0 = tile_m * GBM, n0 = tile_n * GBN;
    const int ph = phase >> 1, pw = phase & 1;
    const unsigned short* Bt = p.Bt + (MODE == A_PHASED ? p.ph.bt_off[phase] : 0);
    const int ldb = MODE == A_PHASED ? p.ph.ntap[phase] * p.ga.Cs : p.ldb;
    const int sc_oy = MODE == A_PHASED ? ph : p.sc.oy, sc_ox = MODE == A_PHASED ? pw : p.sc.ox;
    const int k_total = MODE == A_PHASED ? p.ph.ntap[phase] * p.ga.Cs : p.K;
    const int k_begin = split_id * p.k_per_split;
    const int k_end = min(k_total, k_begin + p.k_per_split);
    const int* tap_dy = MODE == A_PHASED ? p.ph.dy[phase] : p.ga.dy;
    const int* tap_dx = MODE == A_PHASED ? p.ph.dx[phase] : p.ga.dx;

    // ---- staging roles (direct-to-LDS): wave w, piece ps fills LDS rows (w*AP + ps)*8 .. +7 of the A tile (1 KiB, lane-linear) and
    //      rows (w*BP + ps)*8 .. +7 of the B tile; lane l lands at row base + (l >> 3), PHYSICAL chunk l & 7, so it fetches the
    //      LOGICAL chunk (l & 7) ^ f(row) of that row from global memory (the XOR swizzle is applied on the source side).
    //      Gather modes: everything about the lane's output pixel is folded ONCE into a pixel index and a bit mask of the filter
    //      taps that fall inside the image; per K tile only a (scalar) tap offset is added -- no bounds compares, no 64-bit
    //      multiplies in the loop. ----
    const int schunk_phys = lane & 7;
    int a_pix[AP], a_mask[AP], a_chunk[AP], b_chunk[BP];
    const unsigned short* a_ptr[AP];
    const unsigned short* b_ptr[BP];
    bool b_ok[BP];
    const int ntaps = MODE == A_PLAIN ? 0 : k_total / p.ga.Cs;
#pragma unroll
    for (int ps = 0; ps < AP; ++ps) {
        const int trow = (wid * AP + ps) * 8 + (lane >> 3);
        a_chunk[ps] = (schunk_phys ^ ((trow >> 1) & 7)) * 8;
        const int m = m0 + trow;
        const bool ok = m < p.M;
        if (MODE != A_PLAIN) {
            const int hw = p.ga.Hg * p.ga.Wg;
            const int n = m / hw, rem = m - n * hw;
            const int i = rem / p.ga.Wg, j = rem - i * p.ga.Wg;
            const int iy = i * p.ga.stride, jx = j * p.ga.stride;
            a_pix[ps] = (n * p.ga.Hs + iy) * p.ga.Ws + jx;
            int mask = 0;
            for (int t = 0; t < ntaps; ++t) {
                const int y = iy + tap_dy[t], x = jx + tap_dx[t];
                if (ok && (unsigned)y < (unsigned)p.ga.Hs && (unsigned)x < (unsigned)p.ga.Ws) mask |= 1 << t;
            }
            a_mask[ps] = mask;
            a_ptr[ps] = p.A + a_chunk[ps];
        } else {
            a_pix[ps] = 0;
            a_mask[ps] = ok ? 1 : 0;
            a_ptr[ps] = p.A + (long long)(ok ? m : 0) * p.lda + a_chunk[ps];
        }
    }
#pragma unroll
    for (int ps = 0; ps < BP; ++ps) {
        const int trow = (wid * BP + ps) * 8 + (lane >> 3);
        b_chunk[ps] = (schunk_phys ^ ((trow >> 1) & 7)) * 8;
        const int n = n0 + trow;
        b_ok[ps] = n < p.N;
        b_ptr[ps] = Bt + (long long)(b_ok[ps] ? n : 0) * ldb + b_chunk[ps];
    }
    const char* zero_src = reinterpret_cast<const char*>(epi_zero_chunk);
    // K tiles are issued strictly in sequence, so the position of the NEXT tile to be issued is running state: k_run (the k0 arguments below
    // are kept for readability only).  (Round 4 measured a per-M-tile ROTATION of that sequence and padded row pitches against the idea that
    // the power-of-two row pitches make all workgroups hit the same few L2 / memory channels at once: no effect at any pitch --
    // profiles/r04_probe_pitch_rotation.txt -- so every workgroup starts at k_begin.)
    const int nk = k_end > k_begin ? (k_end - k_begin + GBK - 1) / GBK : 0;     // 0: a phase without taps / an empty split (zeros)
    int k_run = k_begin;
    // gather state of the NEXT K tile to be issued (wave-uniform): filter tap, channel offset inside the tap, pixel shift
    int g_tap = MODE == A_PLAIN ? 0 : k_run / p.ga.Cs;
    int g_c0 = MODE == A_PLAIN ? 0 : k_run - g_tap * p.ga.Cs;
    int g_shift = (MODE == A_PLAIN || ntaps == 0) ? 0 : tap_dy[min(g_tap, ntaps - 1)] * p.ga.Ws + tap_dx[min(g_tap, ntaps - 1)];
    auto issue_a = [&](int ps, int, int buf) {
        const int k0 = k_run;
        char* a_s = smem + buf * Cfg::STAGE_BYTES + __builtin_amdgcn_readfirstlane(wid) * (AP * 1024);
        const void* src;
        if (MODE != A_PLAIN) {       // K tiles never straddle a tap (Cs % 64 == 0) and K has no tail in the gather modes
            const unsigned eoff = (unsigned)(a_pix[ps] + g_shift) * (unsigned)(p.ga.pitch ? p.ga.pitch : p.ga.Cs) + (unsigned)g_c0;
            // (the staggered loop issues one or two tiles past the end of its K range: zero chunks, never read)
            const bool in_k = !Cfg::STAGGER || k0 < k_end;
            src = (in_k && ((a_mask[ps] >> (g_tap & 31)) & 1)) ? reinterpret_cast<const void*>(a_ptr[ps] + eoff) : reinterpret_cast<const void*>(zero_src);
        } else {
            const bool ok = a_mask[ps] && (k0 + a_chunk[ps] < k_end);          // K tail (K % 8 == 0): zero-filled chunks
            src = ok ? reinterpret_cast<const void*>(a_ptr[ps] + k0) : reinterpret_cast<const void*>(zero_src);
        }
        glds16(src, a_s + ps * 1024);
    };
    auto issue_b = [&](int ps, int, int buf) {
        const int k0 = k_run;
        char* b_s = smem + buf * Cfg::STAGE_BYTES + Cfg::A_BYTES + __builtin_amdgcn_readfirstlane(wid) * (BP * 1024);
        const bool ok = b_ok[ps] && (MODE != A_PLAIN || k0 + b_chunk[ps] < k_end);
        glds16(ok ? reinterpret_cast<const void*>(b_ptr[ps] + k0) : reinterpret_cast<const void*>(zero_src), b_s + ps * 1024);
    };
    auto advance_gather = [&]() {    // after all pieces of one K tile were issued
        k_run += GBK;
        if (MODE != A_PLAIN) {
            g_c0 += GBK;
            if (g_c0 >= p.ga.Cs) {
                g_c0 = 0;
                g_tap += 1;
                const int t = min(g_tap, ntaps - 1);
                g_shift = tap_dy[t] * p.ga.Ws + tap_dx[t];
            }
        }
    };
    auto issue_tile = [&](int k0, int buf) {
#pragma unroll
        for (int ps = 0; ps < AP; ++ps) issue_a(ps, k0, buf);
#pragma unroll
        for (int ps = 0; ps < BP; ++ps) issue_b(ps, k0, buf);
        advance_gather();
    };

    f32x16 acc[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31, fhalf = lane >> 5;
    // BNIN: (scale, shift) of all K channels of the A operand, behind the staging ring (the epilogue's park and combine areas end below it)
    float2* bn_tab = reinterpret_cast<float2*>(smem + Cfg::NSTAGE * Cfg::STAGE_BYTES);
    if (BNIN) {
        const bool lead = (blockIdx.x | blockIdx.y | blockIdx.z) == 0;
        for (int k = tid; k < p.K; k += Cfg::THREADS) {
            float sc, sh;
            bn_derive(p.bn_in, p.K, k, lead, sc, sh);
            bn_tab[k] = float2{sc, sh};
        }
        if (lead && tid == 0 && p.bn_in.num_batches && p.bn_in.sums) p.bn_in.num_batches[0] += 1;
    }
    const int a_frag = lds_off(wm * (TM * 32) + frow, fhalf), b_frag = lds_off(wn * 64 + frow, fhalf);
    // fragment address of (row + 32*t, k step ks): rows 32 apart share the swizzle term -> + t*4096; the k step flips
    // chunk bits 1..2 of the XOR-swizzled chunk index -> ^ (ks << 5)
    auto read_frags = [&](const char* a_s, const char* b_s, int ks, bf16x8 (&af)[TM], bf16x8 (&bfr)[2]) {
#pragma unroll
        for (int t = 0; t < TM; ++t)
            af[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4v*>(a_s + ((a_frag ^ (ks << 5)) + t * 4096)));
#pragma unroll
        for (int t = 0; t < 2; ++t)
            bfr[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4v*>(b_s + ((b_frag ^ (ks << 5)) + t * 4096)));
    };
    // the share of one K tile's DMA that goes between the MFMA groups of k step ks (a quarter of the A and of the B pieces)
    auto issue_quarter = [&](int ks, int k0, int buf) {
#pragma unroll
        for (int ps = 0; ps < AP; ++ps)
            if (ps * 4 / AP == ks) issue_a(ps, k0, buf);
#pragma unroll
        for (int ps = 0; ps < BP; ++ps)
            if ((BP >= 4 ? ps * 4 / BP : ps) == ks) issue_b(ps, k0, buf);
        if (ks == 3) advance_gather();
    };
    // One K tile from buffer `buf`: the fragments of k step ks+1 are requested before the MFMAs of step ks issue (LDS latency
    // hidden behind the matrix pipe); `spread_next` interleaves a later tile's DMA, a quarter per k step, between the MFMA groups.
    auto compute_tile = [&](int buf, bool spread_next, int k_next, int buf_next, int k_tile = 0) {
        const char* a_s = smem + buf * Cfg::STAGE_BYTES;
        const char* b_s = a_s + Cfg::A_BYTES;
        bf16x8 af[2][TM], bfr[2][2];
        float4v bt[2][4];                           // BNIN: (scale, shift) of the lane's eight k of a step, fetched with the fragments
        auto read_tab = [&](int ks, float4v (&t)[4]) {
            const float4v* src = reinterpret_cast<const float4v*>(bn_tab + k_tile + ks * 16 + fhalf * 8);
#pragma unroll
            for (int q = 0; q < 4; ++q) t[q] = src[q];
        };
        read_frags(a_s, b_s, 0, af[0], bfr[0]);
        if (BNIN) read_tab(0, bt[0]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < 3) read_frags(a_s, b_s, ks + 1, af[(ks + 1) & 1], bfr[(ks + 1) & 1]);
            if (BNIN && ks < 3) read_tab(ks + 1, bt[(ks + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);      // keep the prefetch ABOVE this step's MFMAs (the scheduler would sink it)
            if (BNIN) {
#pragma unroll
                for (int ti = 0; ti < TM; ++ti) af[ks & 1][ti] = bn_in_apply(af[ks & 1][ti], bt[ks & 1]);
            }
            if (spread_next) issue_quarter(ks, k_next, buf_next);
            // operands swapped: D[i][j] with i = output column n (register rows), j = output row m (lane & 31),
            // so that a lane ends up holding 4 consecutive columns of one row
#pragma unroll
            for (int ti = 0; ti < TM; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj)
                    acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[ks & 1][tj], af[ks & 1][ti], acc[ti][tj], 0, 0, 0);
        }
    };
    GEMM_STAMP(2);
    GEMM_STAMP_V(11, (unsigned long long)nk);
    if constexpr (Cfg::STAGGER) {
        // ---- staggered loop (round 5; 8 waves, 256 x 256 tile) ----
        // The lock-step loop below lets all eight waves read fragments, issue DMA and wait at the same moments, so the matrix pipe of every
        // SIMD idles whenever its two waves do anything else (ISA + phase trace: < 50 % MFMA duty).  Here the workgroup is two GROUPS of four
        // waves -- group g = wave row wm owns output rows g*128 .. +127 -- that run the same sequence of SLOTS one barrier apart:
        //     READ(t,0)  fragments of B (all 64 k) and of the group's first 64 A rows -> 16 ds_read_b128; DMA of the A tile t+1     | barrier
        //     MFMA(t,0)  16 MFMAs (rows 0..63 of the wave's 128 x 64, 4 k steps); counted vmcnt                                   | barrier
        //     READ(t,1)  fragments of the second 64 A rows -> 8 ds_read_b128; DMA of the B tile t+2                                  | barrier
        //     MFMA(t,1)  16 MFMAs with the B fragments still in registers; counted vmcnt                                           | barrier
        // Group 1 passes one extra barrier first, so whenever group 0 is in a READ slot group 1 is in an MFMA slot and vice versa: the two
        // waves that share a SIMD alternate between feeding its matrix pipe (512 cycles per slot) and everything else (reads, DMA issue, waits).
        // Hazards, in slots (group 0's READ(t,0) is slot 4t, group 1's is 4t+1; a barrier ends every slot; waves 4g .. 4g+3 stage A rows of group g
        // and all waves a share of B):
        //   WAR  A(t+1) lands in buffer (t+1)&1 whose A rows were last read in READ(t-1,1) (slots 4t-2 / 4t-1, lgkmcnt(0) before the barrier);
        //        B(t+2) lands in buffer t&1 whose B rows were last read in READ(t,0) (slots 4t / 4t+1) and is issued in slots 4t+2 / 4t+3.
        //   RAW  DMA completes in issue order per wave, batches of 4 pieces: ... A(t+1), B(t+2), A(t+2), B(t+3) ...; `vmcnt(4)` at the end of
        //        every MFMA slot leaves only the newest batch in flight, so before the barrier that opens READ(t,0) every wave has retired its
        //        pieces of A(t) and B(t), and the barrier publishes them.  A tile has 3.5 slots (~1800 cycles) to land, a B tile 5.
        //   Tiles past the end of the K range are issued as zero chunks (same counts, no branches) into buffers nobody reads again.
        static_assert(AP == 4 && BP == 4 && TM == 4 && Cfg::NW == 8 && Cfg::NSTAGE == 2 && !BNIN, "staggered loop: 256 x 256 tile, 8 waves");
        const int grp = __builtin_amdgcn_readfirstlane(wid) >> 2;          // == wm
        int kb_run = k_begin;                                               // position of the next B tile to be issued (A's is k_run)
        auto issue_a_tile = [&](int buf) {
#pragma unroll
            for (int ps = 0; ps < AP; ++ps) issue_a(ps, 0, buf);
            advance_gather();
        };
        auto issue_b_tile = [&](int buf) {
            char* b_s = smem + buf * Cfg::STAGE_BYTES + Cfg::A_BYTES + __builtin_amdgcn_readfirstlane(wid) * (BP * 1024);
#pragma unroll
            for (int ps = 0; ps < BP; ++ps) {
                const bool ok = b_ok[ps] && kb_run + b_chunk[ps] < k_end;
                glds16(ok ? reinterpret_cast<const void*>(b_ptr[ps] + kb_run) : reinterpret_cast<const void*>(zero_src), b_s + ps * 1024);
            }
            kb_run += GBK;
        };
        issue_b_tile(0);
        issue_a_tile(0);
        issue_b_tile(1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        GEMM_STAMP(3);
        if (grp) __builtin_amdgcn_s_barrier();
        bf16x8 af[4][2], bfr[4][2];
        auto mfma_half = [&](int h) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                    for (int tj = 0; tj < 2; ++tj)
                        acc[2 * h + ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[ks][tj], af[ks][ti], acc[2 * h + ti][tj], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        };
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            const char* a_s = smem + buf * Cfg::STAGE_BYTES;
            const char* b_s = a_s + Cfg::A_BYTES;
            // READ(t,0)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj)
                    bfr[ks][tj] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4v*>(b_s + ((b_frag ^ (ks << 5)) + tj * 4096)));
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
                    af[ks][ti] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4v*>(a_s + ((a_frag ^ (ks << 5)) + ti * 4096)));
            __builtin_amdgcn_sched_barrier(0);
            issue_a_tile(buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            mfma_half(0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // READ(t,1)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
                    af[ks][ti] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4v*>(a_s + ((a_frag ^ (ks << 5)) + (2 + ti) * 4096)));
            __builtin_amdgcn_sched_barrier(0);
            issue_b_tile(buf);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            mfma_half(1);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!grp) __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                // the epilogue reuses the staging LDS
    } else if constexpr (Cfg::NSTAGE > 2) {
        // ---- pipelined loop: tiles kt+1 .. kt+S-2 stay in flight across the barrier of tile kt ----
        // RAW: a wave's `s_waitcnt vmcnt(N)` retires ITS pieces of tile kt (DMA completes in issue order), the barrier that
        // follows makes every wave's pieces visible to every reader.  WAR: tile kt+S-1 goes into the buffer of tile kt-1, whose
        // last fragment reads were consumed by MFMAs that precede this barrier in every wave.
        constexpr int S = Cfg::NSTAGE, DPT = AP + BP;
#pragma unroll
        for (int t = 0; t < S - 1; ++t)
            if (t < nk) issue_tile(k_begin + t * GBK, t);
        int kt = 0;
        for (; kt + (S - 2) < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * DPT) : "memory");
            __builtin_amdgcn_s_barrier();
#ifdef EPI_GEMM_TRACE
            if (kt == 0) GEMM_STAMP(3);
#endif
            const int nxt = kt + S - 1;
            compute_tile(kt % S, nxt < nk, k_begin + nxt * GBK, nxt % S);
        }
        for (; kt < nk; ++kt) {                      // drain: fewer tiles behind this one
            if (S > 3 && nk - 1 - kt == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            compute_tile(kt % S, false, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                // the epilogue reuses the staging LDS
    } else {
        if (nk > 0) issue_tile(k_begin, 0);
        __syncthreads();                               // drains the DMA (vmcnt(0)) before the first fragment reads
        GEMM_STAMP(3);
        // 2-stage loop.  The next tile's DMA: 4-wave tiles (two workgroups per CU) issue ALL of it right at the top, so that it
        // has the whole tile's MFMA time to land before the closing barrier; the 8-wave 256^2 tile (one workgroup owns the CU)
        // spreads it a quarter per k step between the MFMA groups, so that its waves are never all in a load-only phase.
        for (int kt = 0; kt < nk; ++kt) {
            const bool has_next = kt + 1 < nk;
            const int k_next = k_begin + (kt + 1) * GBK, buf = kt & 1;
            if (Cfg::EARLY && has_next) issue_tile(k_next, buf ^ 1);
            compute_tile(buf, !Cfg::EARLY && has_next, k_next, buf ^ 1, k_begin + kt * GBK);
            __syncthreads();
        }
    }

    GEMM_STAMP(4);
    // ---- epilogue: lane holds, for tile (ti, tj): row m = wm*TM*32 + ti*32 + (lane & 31),
    //      columns n = wn*64 + tj*32 + 8*q + 4*(lane >> 5) + e   for reg = 4*q + e ----
    if (gridDim.y > 1) {        // split-K partial: fp32, plain rows, finished by splitk_finish_kernel
        float* slab = p.slabs + ((long long)split_id * gridDim.z + phase) * p.M * p.N;
#pragma unroll
        for (int ti = 0; ti < TM; ++ti) {
            const int m = m0 + wm * (TM * 32) + ti * 32 + frow;
            if (m >= p.M) continue;
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = n0 + wn * 64 + tj * 32 + 8 * q + 4 * fhalf;
                    if (n + 3 < p.N) {
                        float4v t; t.x = acc[ti][tj][4 * q]; t.y = acc[ti][tj][4 * q + 1]; t.z = acc[ti][tj][4 * q + 2]; t.w = acc[ti][tj][4 * q + 3];
                        *reinterpret_cast<float4v*>(slab + (long long)m * p.N + n) = t;
                    } else {
                        for (int e = 0; e < 4 && n + e < p.N; ++e) slab[(long long)m * p.N + n + e] = acc[ti][tj][4 * q + e];
                    }
                }
        }
#ifdef EPI_GEMM_TRACE
        GEMM_STAMP(6);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        GEMM_STAMP(7);
        GEMM_STAMP_V(8, __builtin_amdgcn_s_memrealtime());
#endif
        return;
    }
    if constexpr (!OUT_F32) {
        if (p.coalesce) {           // bf16 result, N % 8 == 0, 16-byte aligned rows: whole 128-byte row segments through LDS
            coalesced_epilogue<TM, Cfg::WM, Cfg::WN, Cfg::THREADS, RED>(EPI_ARGS_OF(p), acc, smem, m0, n0, tile_m, wm, wn, wid, lane, tid, sc_oy, sc_ox);
            return;
        }
    }
#pragma unroll
    for (int ti = 0; ti < TM; ++ti) {
        const int m = m0 + wm * (TM * 32) + ti * 32 + frow;
        const bool row_ok = m < p.M;
        long long orow = m;
        if (p.sc.enabled) {
            const int hw = p.sc.Hg * p.sc.Wg;
            const int n = m / hw, rem = m - n * hw;
            const int i = rem / p.sc.Wg, j = rem - i * p.sc.Wg;
            orow = ((long long)n * p.sc.Ho + i * p.sc.so + sc_oy) * p.sc.Wo + j * p.sc.so + sc_ox;
        }
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * 64 + tj * 32 + 8 * q + 4 * fhalf;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[ti][tj][4 * q + e];
                    if (p.bias && n + e < p.N) v[e] += p.bias[n + e];
                }
                if (row_ok && n < p.N) {
                    if (OUT_F32) {
                        float* c = reinterpret_cast<float*>(p.C) + orow * p.ldc + n;
                        if (n + 3 < p.N) { float4v t; t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3]; *reinterpret_cast<float4v*>(c) = t; }
                        else for (int e = 0; e < 4 && n + e < p.N; ++e) c[e] = v[e];
                    } else {
                        unsigned short* c = reinterpret_cast<unsigned short*>(p.C) + orow * p.ldc + n;
                        if (p.addend)
                            for (int e = 0; e < 4 && n + e < p.N; ++e) v[e] = bf16_to_f32(f32_to_bf16(v[e])) + bf16_to_f32(p.addend[orow * p.ldc + n + e]);
                        if (n + 3 < p.N) {
                            uint2 t;
                            t.x = pack_bf16x2(v[0], v[1]);
                            t.y = pack_bf16x2(v[2], v[3]);
                            *reinterpret_cast<uint2*>(c) = t;
                        } else for (int e = 0; e < 4 && n + e < p.N; ++e) c[e] = f32_to_bf16(v[e]);
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// A-stationary variant for a SHORT K and a WIDE N (the final 1x1 convolution forward: M = B*H*W = 131 072 rows,
// K = 256 input channels, N = J*D = 1088 output channels).  With only K/64 = 4 K-tiles per output tile the generic
// kernel is a chain of exposed DMA latencies (310 TF), while the operation itself is bound by writing C (285 MB).
// Here a workgroup owns 256 rows of A for ALL N: every wave keeps its 32 rows x K of A in registers as MFMA
// fragments (64 VGPRs at K = 256, loaded once), the workgroup streams [64 n][K] slices of Bt through a double-buffered
// LDS tile (DMA, source-side XOR swizzle over the 32 16-byte chunks of a row), and each wave writes its 32 x 64 result
// of every slice as whole 128-byte row segments through a private 4 KiB LDS staging area.
// ---------------------------------------------------------------------------------------------------------------
constexpr int AS_WAVES = 8, AS_LOADERS = 2, AS_RING = 3;                 // 8 MFMA waves + 2 loader waves, 3-deep ring of B slices
constexpr int AS_THREADS = 64 * (AS_WAVES + AS_LOADERS), AS_BM = 32 * AS_WAVES, AS_BN = 64;

template <int KSTEPS>          // K = 16 * KSTEPS (64, 128 or 256)
__global__ __launch_bounds__(AS_THREADS, 1) void head_gemm_astat_kernel(GemmArgs p) {
    constexpr int K = 16 * KSTEPS, ROW_BYTES = 2 * K, CHUNKS = ROW_BYTES / 16;      // one Bt row in LDS
    constexpr int BTILE = AS_BN * ROW_BYTES;                                           // <= 32 KiB
    constexpr int ROWS_PER_PIECE = 1024 / ROW_BYTES, PIECES = BTILE / 1024 / AS_LOADERS;  // 1 KiB DMA pieces per slice per loader wave
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [AS_RING][BTILE] | [AS_WAVES][4 KiB] staging | bias [N] f32
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const bool loader = wid >= AS_WAVES;
    const int frow = lane & 31, fhalf = lane >> 5;
    const int m0 = xcd_remap(blockIdx.x, gridDim.x) * AS_BM + wid * 32;
    const int n_tiles = (p.N + AS_BN - 1) / AS_BN;
    const char* zero_src = reinterpret_cast<const char*>(epi_zero_chunk);
    float* bias_s = reinterpret_cast<float*>(smem + AS_RING * BTILE + AS_WAVES * 4096);
    float* stats_s = bias_s + n_tiles * AS_BN;                 // [2][n_tiles * 64] per-column (sum | sum of squares) of this workgroup
    for (int n = tid; n < n_tiles * AS_BN; n += AS_THREADS) bias_s[n] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
    if (p.stats)
        for (int n = tid; n < 2 * n_tiles * AS_BN; n += AS_THREADS) stats_s[n] = 0.f;

    // ---- B slice DMA (loader waves only): piece q of loader w covers LDS rows (w*PIECES + q)*ROWS_PER_PIECE ..; lane l
    //      lands at row r = base + l / CHUNKS, physical chunk l % CHUNKS, and fetches logical chunk (l % CHUNKS) ^ (r % CHUNKS).
    //      Dedicated loader waves keep vmcnt of the MFMA waves free of loads: those never wait for their result stores,
    //      which stay in flight across the slice barriers (a wave that issues both must drain both), and the loaders run
    //      one slice ahead of the slice being consumed (counted vmcnt: loads return in order). ----
    auto issue_b = [&](int nt) {
        const int lw = wid - AS_WAVES;
        char* dst = smem + (nt % AS_RING) * BTILE + lw * (PIECES * 1024);
#pragma unroll 8
        for (int q = 0; q < PIECES; ++q) {
            const int r = (lw * PIECES + q) * ROWS_PER_PIECE + lane / CHUNKS;
            const int chunk = (lane % CHUNKS) ^ (r % CHUNKS);
            const int n = nt * AS_BN + r;
            const void* src = n < p.N ? reinterpret_cast<const void*>(p.Bt + (long long)n * p.ldb + chunk * 8)
                                      : reinterpret_cast<const void*>(zero_src);
            glds16(src, dst + q * 1024);
        }
    };
    // ---- A fragments (MFMA waves): row m0 + frow, k = 16*ks + 8*fhalf .. +7, resident for the whole kernel ----
    bf16x8 af[KSTEPS];
    if (loader) {
        issue_b(0);
        if (n_tiles > 1) issue_b(1);
    } else {
        const int m = m0 + frow;
        const uint4v* arow = reinterpret_cast<const uint4v*>(p.A + (long long)(m < p.M ? m : 0) * p.lda) + fhalf;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            uint4v v = arow[2 * ks];
            if (m >= p.M) v.x = v.y = v.z = v.w = 0u;
            af[ks] = __builtin_bit_cast(bf16x8, v);
        }
    }
    __syncthreads();                         // slices 0 and 1 landed, bias staged, A fragments loaded
    if (loader) {
        for (int nt = 0; nt < n_tiles; ++nt) {
            // slice nt+2 goes where slice nt-1 was: its readers passed barrier nt-1.  Before barrier nt, slice nt+1 must have
            // landed: everything but the PIECES just issued.
            if (nt + 2 < n_tiles) {
                issue_b(nt + 2);
                asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PIECES) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            }
        }
        if (p.stats) __builtin_amdgcn_s_barrier();      // pairs with the MFMA waves' barrier before the statistics flush
        return;
    }
    char* stage = smem + AS_RING * BTILE + wid * 4096;
    const int c8 = lane & 7;
    // the slice loop exists in four compile-time copies (residual addend x BatchNorm statistics), selected ONCE: tested per slice and per
    // copied row these wave-uniform flags were ~15 branches per slice beside 8 .. 32 MFMAs
    auto slices = [&](auto add_c, auto stats_c) __attribute__((always_inline)) {
    constexpr bool ADD = decltype(add_c)::value, STATS = decltype(stats_c)::value;
    for (int nt = 0; nt < n_tiles; ++nt) {
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        // B fragments are requested two k steps (four ds_read_b128) ahead of the MFMAs that consume them.  Address of
        // (row r, k step ks) = slice base + r*ROW_BYTES + (((2ks + fhalf) ^ r) % CHUNKS) * 16 = (ks = 0 address) ^ (ks << 5):
        // one XOR with a constant per read, on an offset that changes with nt (so that the 32 addresses are not hoisted
        // out of the slice loop into 32 live registers)
        const unsigned slice_off = (unsigned)(nt % AS_RING) * BTILE;
        auto read_b = [&](int ks, uint4v (&f)[2]) {
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) {
                const int r = tj * 32 + frow;
                const unsigned off = (slice_off + r * ROW_BYTES + (((fhalf ^ r) % CHUNKS) << 4)) ^ (unsigned)(((2 * ks) % CHUNKS) << 4);
                f[tj] = *reinterpret_cast<const uint4v*>(smem + off);
            }
        };
        // residual-junction addend of this slice: requested before the MFMAs and consumed after the LDS round trip where the
        // registers allow it (K <= 128: the stationary A fragments leave room); loaded at its use otherwise
        uint4v ad[4];
        auto load_addend = [&]() {
            if (!ADD) return;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int m = m0 + it * 8 + (lane >> 3), nn = nt * AS_BN + (lane & 7) * 8;
                ad[it] = (m < p.M && nn < p.N) ? *reinterpret_cast<const uint4v*>(p.addend + (long long)m * p.ldc + nn) : uint4v{0u, 0u, 0u, 0u};
            }
        };
        if (KSTEPS <= 8) load_addend();
        uint4v fb[3][2];
        read_b(0, fb[0]);
        if (KSTEPS > 1) read_b(1, fb[1]);
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            if (ks + 2 < KSTEPS) read_b(ks + 2, fb[(ks + 2) % 3]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
                acc[tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[ks % 3][tj]), af[ks], acc[tj], 0, 0, 0);
        }
        // every LDS read of this slice is consumed: release the buffer (the loader refills it two slices ahead)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // ---- lane holds row m0 + frow, columns nt*64 + tj*32 + 8*q + 4*fhalf + e (reg 4*q + e): park as bf16 in the wave's
        //      staging rows (8-byte units XOR (row & 15)), then whole 128-byte row segments to global ----
        const int n0 = nt * AS_BN;
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nl = tj * 32 + 8 * q + 4 * fhalf;
                const float4v bv = *reinterpret_cast<const float4v*>(bias_s + n0 + nl);
                const float v0 = acc[tj][4 * q] + bv.x, v1 = acc[tj][4 * q + 1] + bv.y, v2 = acc[tj][4 * q + 2] + bv.z,
                            v3 = acc[tj][4 * q + 3] + bv.w;
                uint2 t;
                t.x = pack_bf16x2(v0, v1);
                t.y = pack_bf16x2(v2, v3);
                const int unit = tj * 8 + 2 * q + fhalf;
                *reinterpret_cast<uint2*>(stage + frow * 128 + ((unit ^ (frow & 15)) << 3)) = t;
            }
        const int n = n0 + c8 * 8;
        float ssum[8], ssq[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { ssum[k] = 0.f; ssq[k] = 0.f; }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + (lane >> 3), sw = row & 15;
            const uint2 lo = *reinterpret_cast<const uint2*>(stage + row * 128 + (((2 * c8) ^ sw) << 3));
            const uint2 hi = *reinterpret_cast<const uint2*>(stage + row * 128 + (((2 * c8 + 1) ^ sw) << 3));
            const int m = m0 + row;
            if (m < p.M && n < p.N) {
                uint4v o; o.x = lo.x; o.y = lo.y; o.z = hi.x; o.w = hi.y;
                if (ADD) o = add_bf16x8(o, KSTEPS <= 8 ? ad[it] : *reinterpret_cast<const uint4v*>(p.addend + (long long)m * p.ldc + n));
                *reinterpret_cast<uint4v*>(reinterpret_cast<unsigned short*>(p.C) + (long long)m * p.ldc + n) = o;
                if (STATS) {      // BatchNorm statistics of the bf16-rounded outputs (see coalesced_epilogue)
                    const unsigned int w4[4] = {lo.x, lo.y, hi.x, hi.y};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float a = __uint_as_float(w4[k] << 16), b = __uint_as_float(w4[k] & 0xffff0000u);
                        ssum[2 * k] += a; ssq[2 * k] = fmaf(a, a, ssq[2 * k]);
                        ssum[2 * k + 1] += b; ssq[2 * k + 1] = fmaf(b, b, ssq[2 * k + 1]);
                    }
                }
            }
        }
        if (STATS) {          // reduce over the 8 row lanes, then one LDS atomic per column and wave
#pragma unroll
            for (int k = 0; k < 8; ++k) {
#pragma unroll
                for (int o = 8; o < 64; o <<= 1) { ssum[k] += __shfl_xor(ssum[k], o, 64); ssq[k] += __shfl_xor(ssq[k], o, 64); }
            }
            if (lane < 8) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    atomicAdd(stats_s + n + k, ssum[k]);
                    atomicAdd(stats_s + n_tiles * AS_BN + n + k, ssq[k]);
                }
            }
        }
    }
    };
    if (p.addend) { if (p.stats) slices(std::true_type{}, std::true_type{}); else slices(std::true_type{}, std::false_type{}); }
    else if (p.stats) slices(std::false_type{}, std::true_type{});
    else slices(std::false_type{}, std::false_type{});
    if (p.stats) {              // every wave's LDS atomics are in: one global atomic per column and workgroup
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        float* dst = p.stats + (long long)(blockIdx.x % p.stats_copies) * 2 * p.N;
        for (int n = tid; n < p.N; n += 64 * AS_WAVES) {
            atomicAdd(dst + n, stats_s[n]);
            atomicAdd(dst + p.N + n, stats_s[n_tiles * AS_BN + n]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 3x3 / stride-1 / pad-1 convolutions (forward, and backward-data = the same convolution with the taps mirrored): patch-stationary
// implicit GEMM.  The generic gather kernel stages a fresh 256 x 64 A tile for EVERY filter tap -- nine L2 -> LDS fills of (almost)
// the same pixels per 64-channel chunk.  PMC on MI355X (profiles/r02_pmc_l2_conv_gemms.txt): L2 hit rate 0.77 .. 0.84, fabric
// reads ~ the algorithmic bytes, yet only ~9 TB/s of L2 -> LDS fill for 15 % MFMA utilisation -- the K loop waits for the slowest
// line of each tile, and the bytes in flight are capped by the LDS that receives them.  So: fewer bytes per MFMA.
//   * rows of the GEMM are consecutive NHWC pixels, so tap (dy, dx) of output pixel m is input pixel m + dy*W + dx of the SAME
//     linear pixel array: a workgroup stages the pixels [m0 - W - 1, m0 + 256 + W + 1) of one 64-channel chunk ONCE (the "patch",
//     128 bytes per pixel) and runs all nine taps from it with a row shift; taps that cross an image border are masked per
//     (row, tap) by pointing the lane at a 128-byte block of zeros in LDS (one v_cndmask on the address, not four on the data)
//   * the weights stream through a ring of [BN][64] tiles, one per tap, three deep where LDS allows
//   * A traffic per chunk: one patch (256 + 2W + 2 pixels) instead of 9 x 256 pixels; with the B tiles 193 KB instead of 432 KB
//     per 256 x 128 x 576 MACs.
// DMA schedule: every step (tap) issues exactly BP + 1 instructions per wave -- the B tile two steps ahead and one piece of the
// NEXT chunk's patch (a dummy piece when there is none), so one counted `s_waitcnt vmcnt(BP + 1)` + barrier per step retires
// everything but the group just issued.
// ---------------------------------------------------------------------------------------------------------------
template <int TM_, int WM_, int WN_> struct PatchCfg {
    static constexpr int TM = TM_, WM = WM_, WN = WN_, NW = WM_ * WN_, THREADS = 64 * NW;
    static constexpr int BM = WM_ * TM_ * 32, BN = WN_ * 64, B_BYTES = BN * 128, BP = BN / 8 / NW;
    static_assert(NW == 8 && BP >= 1, "8 waves");
};
typedef PatchCfg<2, 4, 2> PatchWide;      // 256 x 128
typedef PatchCfg<1, 8, 1> PatchNarrow;    // 256 x 64 (the 64-channel layers)
typedef PatchCfg<1, 4, 2> PatchHalf;      // 128 x 128 (mid layers: enough tiles to fill the chip without a channel split)
constexpr int PATCH_TAPS = 9, PATCH_APW = 8;     // patch pieces (8 pixels, 1 KiB) per wave: one per tap step, the ninth step issues a dummy

// -DEPI_PATCH_TRACE (tools/patch_trace.cpp only): s_memtime stamps of every 16th workgroup's wave 0 -- start, prologue issued,
// prologue landed, after every step's barrier, loop drained, results stored
#ifdef EPI_PATCH_TRACE
__device__ unsigned long long epi_patch_trace[64 * 96];
#define PATCH_STAMP(i)                                                                                                   \
    do {                                                                                                                 \
        const int wg_ = blockIdx.x + gridDim.x * blockIdx.y;                                                             \
        if (lane == 0 && wid == 0 && wg_ % 16 == 0 && wg_ / 16 < 64 && (i) < 96) epi_patch_trace[(wg_ / 16) * 96 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define PATCH_STAMP(i) do {} while (0)
#endif

// SINGLE: the workgroup covers ONE 64-channel chunk (64-channel layers, or one chunk per split) -- no second patch buffer, no
// patch pieces in the per-step DMA group, and (narrow tiles) two workgroups per CU: one's prologue / epilogue under the other's loop
template <typename Cfg, int NB, bool SINGLE, bool RED = false>
__global__ __launch_bounds__(Cfg::THREADS, SINGLE ? 4 : 2) void conv_patch_kernel(GemmArgs p) {      // (waves per SIMD)
    constexpr int TM = Cfg::TM, GBN = Cfg::BN, BP = Cfg::BP, NW = Cfg::NW, BB = Cfg::B_BYTES;
    constexpr int NPATCH = SINGLE ? 1 : 2, GROUP = BP + (SINGLE ? 0 : 1);
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [2 patches][NB weight tiles][1 KiB dummy][128 B zeros]
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / Cfg::WN, wn = wid % Cfg::WN;
    const int tiles_n = (p.N + GBN - 1) / GBN;
    const int lid = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
    const int tile_id = lid % (int)gridDim.x, split_id = lid / (int)gridDim.x;
    const int tile_m = tile_id / tiles_n, tile_n = tile_id - tile_m * tiles_n;
    const int m0 = tile_m * Cfg::BM, n0 = tile_n * GBN;
    const int W = p.ga.Wg, H = p.ga.Hg, Cs = p.ga.Cs;
    const int c_begin = split_id * p.chunks_per_split, c_end = min(Cs / GBK, c_begin + p.chunks_per_split);
    const int PB = p.patch_px * 128;
    const int ring_off = NPATCH * PB, dummy_off = ring_off + NB * BB, zero_off = dummy_off + 1024;
    PATCH_STAMP(0);
    if (tid < 32) reinterpret_cast<float*>(smem + zero_off)[tid] = 0.f;
    const char* zero_src = reinterpret_cast<const char*>(epi_zero_chunk);

    // ---- staging roles: wave w owns patch pieces q = j*8 + w (patch pixels 8q .. 8q+7) and weight-tile pieces w*BP + ps ----
    // patch piece j of this wave: source element offset of the lane's 16 bytes (chunk c adds c*64), or -1 outside the pixel array
    auto a_off = [&](int j) -> int {
        const int prow = (j * NW + wid) * 8 + (lane >> 3);
        const long long pin = (long long)m0 - (W + 1) + prow;           // linear input pixel behind patch pixel prow
        const bool ok = prow < p.patch_px && pin >= 0 && pin < p.M;
        return ok ? (int)pin * Cs + (((lane & 7) ^ ((prow >> 1) & 7)) << 3) : -1;
    };
    const unsigned short* b_ptr[BP];
    bool b_ok[BP];
#pragma unroll
    for (int ps = 0; ps < BP; ++ps) {
        const int trow = (wid * BP + ps) * 8 + (lane >> 3);
        const int n = n0 + trow;
        b_ok[ps] = n < p.N;
        b_ptr[ps] = p.Bt + (long long)(b_ok[ps] ? n : 0) * p.ldb + (((lane & 7) ^ ((trow >> 1) & 7)) << 3);
    }
    auto issue_a = [&](int j, int c, int buf) {
        const bool real = j < PATCH_APW && (j * NW + wid) * 8 < p.patch_px && c < c_end;
        char* dst = real ? smem + buf * PB + (j * NW + wid) * 1024 : smem + dummy_off;
        const int off = a_off(j);
        glds16((real && off >= 0) ? reinterpret_cast<const void*>(p.A + off + c * GBK) : reinterpret_cast<const void*>(zero_src), dst);
    };
    auto issue_b = [&](int t, int c, int slot, bool real) {
#pragma unroll
        for (int ps = 0; ps < BP; ++ps) {
            char* dst = real ? smem + ring_off + slot * BB + (wid * BP + ps) * 1024 : smem + dummy_off;
            const void* src = (real && b_ok[ps]) ? reinterpret_cast<const void*>(b_ptr[ps] + t * Cs + c * GBK) : reinterpret_cast<const void*>(zero_src);
            glds16(src, dst);
        }
    };

    const int frow = lane & 31, fhalf = lane >> 5;
    const int b_frag = ring_off + lds_off(wn * 64 + frow, fhalf);

    f32x16 acc[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- prologue: the first chunk's whole patch and the first NB - 1 weight tiles ----
    const int nsteps = (c_end - c_begin) * PATCH_TAPS;
    if (nsteps > 0) {
#pragma unroll
        for (int j = 0; j < PATCH_APW; ++j) issue_a(j, c_begin, 0);
        issue_b(0, c_begin, 0, true);
        if (NB > 2) issue_b(1, c_begin, 1, true);
    }
    __builtin_amdgcn_sched_barrier(0);             // the index math below runs behind the DMA, not in front of it
    // ---- per (row, tap) validity: the tap must stay inside the row's own image (computed while the prologue's DMA is in flight) ----
    int amask[TM], arow[TM];
#pragma unroll
    for (int ti = 0; ti < TM; ++ti) {
        const int r = wm * (TM * 32) + ti * 32 + frow, m = m0 + r;
        arow[ti] = r;
        const int j = m % W, i = (m / W) % H;
        int mask = 0;
#pragma unroll
        for (int t = 0; t < PATCH_TAPS; ++t) {
            const int y = i + p.ga.dy[t], x = j + p.ga.dx[t];
            if (m < p.M && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) mask |= 1 << t;
        }
        amask[ti] = mask;
    }
    PATCH_STAMP(1);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    PATCH_STAMP(2);
    int step_no = 0;

    int slot = 0;                                                         // ring slot of the current step
    for (int c = c_begin; c < c_end; ++c) {
        const int pbuf = SINGLE ? 0 : (c - c_begin) & 1;
#pragma unroll
        for (int t = 0; t < PATCH_TAPS; ++t) {
            // -- this step's DMA group: weight tile NB - 1 steps ahead, one piece of the next chunk's patch --
            {
                int ta = t + NB - 1, ca = c;
                if (ta >= PATCH_TAPS) { ta -= PATCH_TAPS; ca += 1; }
                int sa = slot + NB - 1; if (sa >= NB) sa -= NB;
                issue_b(ta, ca, sa, ca < c_end);
                if (!SINGLE) issue_a(t, c + 1, pbuf ^ 1);
            }
            // -- MFMAs of tap t from the resident patch --
            const int sh = (p.ga.dy[t] + 1) * W + p.ga.dx[t] + 1;
            int a_addr[TM];
#pragma unroll
            for (int ti = 0; ti < TM; ++ti) {
                const int pr = arow[ti] + sh;
                const int in = pbuf * PB + pr * 128 + ((fhalf ^ ((pr >> 1) & 7)) << 4);
                a_addr[ti] = ((amask[ti] >> t) & 1) ? in : zero_off;
            }
            const int b_addr = b_frag + slot * BB;
            bf16x8 af[2][TM], bfr[2][2];
            auto read_frags = [&](int ks, bf16x8 (&a)[TM], bf16x8 (&b)[2]) {
#pragma unroll
                for (int ti = 0; ti < TM; ++ti) a[ti] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4v*>(smem + (a_addr[ti] ^ (ks << 5))));
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) b[tj] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4v*>(smem + ((b_addr ^ (ks << 5)) + tj * 4096)));
            };
            read_frags(0, af[0], bfr[0]);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks < 3) read_frags(ks + 1, af[(ks + 1) & 1], bfr[(ks + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ti = 0; ti < TM; ++ti)
#pragma unroll
                    for (int tj = 0; tj < 2; ++tj)
                        acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[ks & 1][tj], af[ks & 1][ti], acc[ti][tj], 0, 0, 0);
            }
            // -- everything but the group just issued has landed (next step's weight tile; at t = 8 the next patch) --
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 2) * GROUP) : "memory");
            __builtin_amdgcn_s_barrier();
            slot = slot + 1 == NB ? 0 : slot + 1;
            PATCH_STAMP(3 + step_no);
            ++step_no;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                  // dummy pieces and fragment reads are done: the epilogue reuses the LDS
    PATCH_STAMP(3 + step_no);

    // ---- epilogue: lane holds, for tile (ti, tj): row wm*TM*32 + ti*32 + (lane & 31), columns wn*64 + tj*32 + 8*q + 4*(lane >> 5) + e ----
    if (gridDim.y > 1) {        // split over channel chunks: fp32 partials, finished by splitk_finish_kernel
        float* slab = p.slabs + (long long)split_id * p.M * p.N;
#pragma unroll
        for (int ti = 0; ti < TM; ++ti) {
            const int m = m0 + arow[ti];
            if (m >= p.M) continue;
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = n0 + wn * 64 + tj * 32 + 8 * q + 4 * fhalf;
                    if (n + 3 < p.N) {
                        float4v t; t.x = acc[ti][tj][4 * q]; t.y = acc[ti][tj][4 * q + 1]; t.z = acc[ti][tj][4 * q + 2]; t.w = acc[ti][tj][4 * q + 3];
                        *reinterpret_cast<float4v*>(slab + (long long)m * p.N + n) = t;
                    }
                }
        }
#ifdef EPI_PATCH_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PATCH_STAMP(4 + step_no);
#endif
        return;
    }
    // bf16 result through the wave's own LDS slice as whole 128-byte row segments; residual addend, BatchNorm statistics and the fused
    // BatchNorm-backward reduction as in head_gemm_kernel (plain rows: p.sc is off on this kernel)
    coalesced_epilogue<TM, Cfg::WM, Cfg::WN, Cfg::THREADS, RED, false>(EPI_ARGS_OF(p), acc, smem, m0, n0, tile_m, wm, wn, wid, lane, tid, 0, 0);
#ifdef EPI_PATCH_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PATCH_STAMP(4 + step_no);
#endif
}

// sum the split-K slabs, add the bias, convert and write through the row scatter (N % 4 == 0)
template <bool OUT_F32>
__global__ void splitk_finish_kernel(const float* __restrict__ slabs, int nsplit, int nphase, GemmArgs p) {
    const long long nq = (long long)p.N >> 2;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)nphase * p.M * nq) return;
    const int n = (int)(t % nq) * 4;
    const long long pm = t / nq;
    const int m = (int)(pm % p.M), phase = (int)(pm / p.M);
    float4v a; a.x = a.y = a.z = a.w = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float4v v = *reinterpret_cast<const float4v*>(slabs + (((long long)s * nphase + phase) * p.M + m) * p.N + n);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    if (p.bias) { a.x += p.bias[n]; a.y += p.bias[n + 1]; a.z += p.bias[n + 2]; a.w += p.bias[n + 3]; }
    long long orow = m;
    if (p.sc.enabled) {
        const int hw = p.sc.Hg * p.sc.Wg;
        const int b = m / hw, rem = m - b * hw;
        const int i = rem / p.sc.Wg, j = rem - i * p.sc.Wg;
        const int oy = p.ph.enabled ? (phase >> 1) : p.sc.oy, ox = p.ph.enabled ? (phase & 1) : p.sc.ox;
        orow = ((long long)b * p.sc.Ho + i * p.sc.so + oy) * p.sc.Wo + j * p.sc.so + ox;
    }
    if (OUT_F32) {
        *reinterpret_cast<float4v*>(reinterpret_cast<float*>(p.C) + orow * p.ldc + n) = a;
    } else {
        if (p.addend) {
            const uint2 r = *reinterpret_cast<const uint2*>(p.addend + orow * p.ldc + n);
            a.x = bf16_to_f32(f32_to_bf16(a.x)) + __uint_as_float(r.x << 16); a.y = bf16_to_f32(f32_to_bf16(a.y)) + __uint_as_float(r.x & 0xffff0000u);
            a.z = bf16_to_f32(f32_to_bf16(a.z)) + __uint_as_float(r.y << 16); a.w = bf16_to_f32(f32_to_bf16(a.w)) + __uint_as_float(r.y & 0xffff0000u);
        }
        uint2 o;
        o.x = pack_bf16x2(a.x, a.y);
        o.y = pack_bf16x2(a.z, a.w);
        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.C) + orow * p.ldc + n) = o;
    }
}

// The same finish for a convolution that feeds a training-mode BatchNorm (plain rows, bf16 result, no bias / addend / phases): it also
// accumulates the per-column sum and sum of squares of the bf16-ROUNDED outputs into p.stats, as the unsplit launches do from their GEMM
// epilogue -- the deep layers at batch 32 (2048 rows) are the ones that split K, and each of them paid a separate statistics launch.
// Block: 256 threads = 64 column quads x 4 row lanes over a tile of FS_ROWS rows x 256 columns; registers -> LDS combine of the 4 row
// lanes -> one contiguous atomic per column and block into copy (row tile % stats_copies).
constexpr int FS_ROWS = 32;
__global__ __launch_bounds__(256) void splitk_finish_stats_kernel(const float* __restrict__ slabs, int nsplit, GemmArgs p) {
    __shared__ float red[4][2][256];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int col_tiles = (p.N + 255) / 256;
    const int tile_m = blockIdx.x / col_tiles, tile_n = blockIdx.x - tile_m * col_tiles;
    const int n = tile_n * 256 + tx * 4;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if (n < p.N) {
        for (int r = ty; r < FS_ROWS; r += 4) {
            const int m = tile_m * FS_ROWS + r;
            if (m >= p.M) break;
            float4v a; a.x = a.y = a.z = a.w = 0.f;
            for (int sp = 0; sp < nsplit; ++sp) {
                const float4v v = *reinterpret_cast<const float4v*>(slabs + ((long long)sp * p.M + m) * p.N + n);
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
            uint2 o;
            o.x = pack_bf16x2(a.x, a.y);
            o.y = pack_bf16x2(a.z, a.w);
            *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.C) + (long long)m * p.ldc + n) = o;
            const float v[4] = {__uint_as_float(o.x << 16), __uint_as_float(o.x & 0xffff0000u), __uint_as_float(o.y << 16), __uint_as_float(o.y & 0xffff0000u)};
#pragma unroll
            for (int e = 0; e < 4; ++e) { s1[e] += v[e]; s2[e] = fmaf(v[e], v[e], s2[e]); }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[ty][0][tx * 4 + e] = s1[e]; red[ty][1][tx * 4 + e] = s2[e]; }
    __syncthreads();
    float* dst = p.stats + (long long)(tile_m % p.stats_copies) * 2 * p.N;
    for (int t = threadIdx.x; t < 512; t += 256) {
        const int which = t >> 8, c = t & 255, col = tile_n * 256 + c;
        if (col < p.N) atomicAdd(dst + which * p.N + col, red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c]);
    }
}

}  // namespace epi

using namespace epi;

enum { CFG_SMALL = 0, CFG_TALL = 1, CFG_BIG = 2, CFG_HALF = 3, CFG_QUARTER = 4 };
struct GemmPlan { int cfg; int nsplit, kps; long long tiles; int pipe; };

// EPI_GEMM_TILE=small|big forces a tile configuration (benchmarking); default: by shape
static int g_tile_force_fwd();
static int gemm_tile_override() {
    const int v = 0;
    const int f = g_tile_force_fwd();
    return f == 1 || f == 2 ? f : (f == 5 ? 2 : v);
}

// Split-K factor for one tile configuration: split until every CU has a workgroup (4-wave tiles: two), each split
// keeping >= 512 of K.
// EPI_GEMM_PIPE: 0 (default) never, 1 when a workgroup's K loop has >= 6 tiles, 2 always (where a pipelined variant exists), 3 as 1 but only
// for launches of <= 256 workgroups (see gemm_plan: round 4 measured 3 per kernel and in the step).
// Measured on MI355X (profiles/r02_conv_layers_c_pipelined_ab.txt): the ring removes the DMA-latency stall but these GEMMs are bound by
// the global->LDS fill itself (32 KB per 128x128x64 tile step, ~23 GB/s per CU = ~6 TB/s over the chip), so it gains nothing here
static int g_pipe_force_fwd();
static int gemm_pipe_mode() {
    const int v = 0;
    const int f = g_pipe_force_fwd();
    return f >= 0 ? f : v;
}

// tile / pipeline overrides at run time (tuning hook; -1 only queries): tile 0 by shape, 1 small, 2 big, 3 half, 4 quarter; pipe as EPI_GEMM_PIPE
static int g_store_policy = -1;
static int gemm_store_policy() {
    if (g_store_policy < 0) g_store_policy = 0;
    return g_store_policy;
}
extern "C" int epi_gemm_store_policy(int v) {
    const int before = gemm_store_policy();
    if (v >= 0) g_store_policy = v;
    return before;
}
static int g_tile_force = 0, g_pipe_force = -1;
extern "C" int epi_gemm_tune(int tile, int pipe) {
    if (tile >= 0) g_tile_force = tile;
    if (pipe >= -1) g_pipe_force = pipe;
    return g_tile_force;
}

static int g_tile_force_fwd() { return g_tile_force; }
static int g_pipe_force_fwd() { return g_pipe_force; }

static GemmPlan gemm_plan_cfg(int cfg, int M, int N, int K, int nphase, bool pipe = false) {
    GemmPlan pl;
    pl.cfg = cfg;
    pl.pipe = pipe ? 1 : 0;
    const int bm = cfg == CFG_SMALL ? 128 : (cfg == CFG_HALF || cfg == CFG_QUARTER ? 64 : 256);
    const int bn = cfg == CFG_BIG ? 256 : (cfg == CFG_TALL || cfg == CFG_QUARTER ? 64 : 128);
    pl.tiles = (long long)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
    const long long wgs = pl.tiles * nphase;
    const bool one_per_cu = cfg == CFG_BIG || pipe;
    // no split once about half the CUs have a workgroup: measured per layer (tools/bench_conv.py), an unsplit 256-workgroup launch of
    // 16 K tiles beats two splits + the finish kernel (ResNet-50 layer4.conv1: 16 us vs 27 us)
    // (EPI_GEMM_ENOUGH: measurement switch.  120 vs 200 on MI355X, profiles/r02_conv_layers_e_*: the 128-tile layers -- ResNet-50
    // layer3 1x1 at batch 32 -- run unsplit in 14.6 us instead of 21.9 us with four splits + finish; 60 loses again)
    const long long enough_env = 120;
    const long long enough = enough_env, target = one_per_cu ? 256 : 512;
    int nsplit = 1;
    if (wgs < enough) {
        nsplit = (int)(target / wgs);               // floor: one workgroup over the resident capacity costs a whole extra round
        const int max_split = K / 512 > 0 ? K / 512 : 1;
        if (nsplit > max_split) nsplit = max_split;
        if (nsplit > 16) nsplit = 16;
        if (nsplit < 1) nsplit = 1;
    }
    if (N % 4) nsplit = 1;
    pl.kps = K;
    if (nsplit > 1) {
        pl.kps = ((K + nsplit - 1) / nsplit + GBK - 1) / GBK * GBK;
        nsplit = (K + pl.kps - 1) / pl.kps;
    }
    pl.nsplit = nsplit;
    return pl;
}

// Tile configuration.  big (256^2, 1 workgroup / CU) needs 16-byte output row segments, columns that fill 256-wide
// tiles, and a deep K loop per workgroup: either enough tiles to fill the chip unsplit (K >= 512), or >= 1024 of K
// left per split (measured on MI355X: below that the 128^2 tile at 2 workgroups / CU hides the pipeline prologue better).
// tall (256 x 64) serves N <= 64 with enough rows to fill the chip.
static GemmPlan gemm_plan(int M, int N, int K, int ldc, int nphase, bool out_f32, bool allow_pipe = true) {
    const int ov = gemm_tile_override();
    const bool can_big = !out_f32 && N % 8 == 0 && ldc % 8 == 0;
    if (can_big && ov != 1) {
        const GemmPlan pb = gemm_plan_cfg(CFG_BIG, M, N, K, nphase);
        const int tiles_n = (N + 255) / 256;
        const bool fills = M >= 256 && tiles_n * 256 <= N + N / 4;
        const bool deep = pb.nsplit == 1 ? (K >= 512 && pb.tiles * nphase >= 200) : pb.kps >= 1024;
        // (round 4) ... unless the 256 x 256 tile has to split K while 128 x 128 tiles fill the chip UNSPLIT: the unsplit launch needs no finish launch and
        // can carry the fused column sums.  The first deconvolution's backward-data (2048 x 2048 x 4096): 49 us + 18 us finish + 11 us BatchNorm-backward
        // reduction on four splits of 64 tiles, 54 us alone on 256 unsplit tiles (tools/bench_nt_tiles.py).  EPI_GEMM_BIG_SPLIT=1: the round-3 choice.
        const bool big_split_ok = false;
        const bool small_fills_unsplit = (long long)((M + 127) / 128) * ((N + 127) / 128) * nphase >= 200;
        if (ov == 2 || (fills && deep && (pb.nsplit == 1 || big_split_ok || !small_fills_unsplit))) return pb;
    }
    // (fp32 results run on the 128 x 128 instantiation only: planning them for the tall tile launched that kernel on the tall tile's grid --
    //  wrong results for N <= 64 at M >= 65536, i.e. the fp32-grade mode's 64-channel layers from batch 16 on; found in round 4 by the
    //  bench-shape trained-state test, tests/test_hip_precise.py)
    int cfg = (ov == 0 && !out_f32 && N <= 64 && (long long)M * nphase >= 256 * 256) ? CFG_TALL : CFG_SMALL;
    // smaller tiles while the launch leaves CUs without two workgroups (EPI_GEMM_FILL: the workgroup count below which the next
    // smaller tile is taken; 0 = never, the default -- measured per layer and in the step (profiles/r02_conv_layers_g_*): a few layers
    // gain 1 .. 5 us, the stride-2 3x3 layers lose 30 us, the step 7.64 (off) / 7.70 (384) / 7.81 ms (640))
    const long long fill_env = 0;
    if (cfg == CFG_SMALL && !out_f32 && (g_tile_force == 3 || g_tile_force == 4)) cfg = g_tile_force == 3 ? CFG_HALF : CFG_QUARTER;
    else if (cfg == CFG_SMALL && ov == 0 && !out_f32 && fill_env > 0) {
        const long long t128 = (long long)((M + 127) / 128) * ((N + 127) / 128) * nphase;
        const long long t64x128 = (long long)((M + 63) / 64) * ((N + 127) / 128) * nphase;
        if (t128 < fill_env) cfg = t64x128 < fill_env ? CFG_QUARTER : CFG_HALF;
    }
    const int pm = (out_f32 || !allow_pipe || cfg == CFG_HALF || cfg == CFG_QUARTER) ? 0 : gemm_pipe_mode();
    if (pm) {       // pipelined ring (one workgroup per CU): pays when every workgroup still has a K loop of >= 6 tiles
        const GemmPlan pp = gemm_plan_cfg(cfg, M, N, K, nphase, true);
        // mode 3 (round 4): only launches that fit the chip in ONE round of one workgroup per CU anyway -- there the two-stage loop exposes a
        // full DMA latency per K tile (0.9 us on operands that come from HBM, tools/gemm_trace.hip) and the ring hides two thirds of it
        // (profiles/r04_gemm_lab_*: 1024->256 at 16 x 16 19.9 -> 14.9 us, 512->128 at 32 x 32 15.8 -> 13.4, 512->2048 at 8 x 8 12.8 -> 10.8);
        // launches with more tiles lose the second resident workgroup per CU to the ring's LDS and run slower (512->256 at 32 x 32 18.6 -> 21.2).
        // In the STEP the operands were written by the previous launch and come from L2 / MALL, where the ring gains nothing (the "warm" column of
        // the same tables): 6.204 / 6.212 ms with mode 3 against 6.194 / 6.187 ms without, fused-reduction launches included
        // (profiles/r04_ab_pipe_mode3.txt) -- so the default stays 0
        const bool one_round = pp.tiles * nphase * pp.nsplit <= 256;
        // mode 4: only the long under-filled launches (<= 128 workgroups with >= 32 K tiles each: the deconvolution backward-data at 16 x 16 / 8 x 8):
        // GEMM family 4.96 -> 4.94 ms on one stream, the two-stream step 6.26 -> 6.30 ms (the ring's 128 KB per CU leave no LDS for the weight gradients beside it)
        const bool long_underfilled = pp.tiles * nphase * pp.nsplit <= 128 && pp.kps >= 32 * GBK;
        if (pm == 2 || (pm == 1 && pp.kps >= 6 * GBK) || (pm == 3 && one_round && pp.kps >= 6 * GBK) || (pm == 4 && long_underfilled)) return pp;
    }
    return gemm_plan_cfg(cfg, M, N, K, nphase);
}

extern "C" size_t epi_gemm_workspace_bytes(int M, int N, int K, int nphase) {
    if (M <= 0 || N <= 0 || K <= 0 || nphase <= 0) return 0;
    // the larger of the two configurations' needs (the output stride / dtype are not known here)
    size_t need = 0;
    for (int pipe = 0; pipe < 2; ++pipe) {             // (with / without the pipelined ring: the split factors differ)
        for (int f32 = 0; f32 < 2; ++f32) {
            const GemmPlan pl = gemm_plan(M, N, K, 8, nphase, f32 != 0, pipe != 0);
            if (pl.nsplit > 1) need = std::max(need, (size_t)pl.nsplit * nphase * M * N * sizeof(float));
        }
        const GemmPlan ps = gemm_plan(M, N, K, 4, nphase, false, pipe != 0);
        if (ps.nsplit > 1) need = std::max(need, (size_t)ps.nsplit * nphase * M * N * sizeof(float));
    }
    return need;
}

template <bool OUT_F32, typename Cfg, int MODE, bool RED = false, bool BNIN = false>
static int launch_gemm_mode(const GemmArgs& a, const GemmPlan& pl, int nphase, hipStream_t st) {
    // staging ring; the epilogue reuses it for the waves' output slices and, behind them, the BatchNorm-statistics combine area
    // (BNIN: the (scale, shift) table of the A operand's K channels behind the ring)
    constexpr size_t ring = (size_t)Cfg::NSTAGE * Cfg::STAGE_BYTES;
    // (round 5) a launch whose workgroups run ONE K tile each (K = 64: the 64-channel layers of ResNet layer 1, forward and backward-data) never touches the
    // second stage of the two-stage loop: it asks for one stage, and three workgroups instead of two fit a CU (registers allow three 4-wave workgroups) --
    // these launches are a chain of latencies per workgroup (operands from HBM, 16 MFMAs, the epilogue), so residency is what hides them
    const bool one_tile = !BNIN && !Cfg::STAGGER && Cfg::NSTAGE == 2 && pl.kps <= GBK;
    const size_t staged = one_tile ? (size_t)Cfg::STAGE_BYTES : ring;
    const size_t lds = BNIN ? ring + (size_t)a.K * sizeof(float2)
                            : std::max(staged, (size_t)Cfg::WM * Cfg::WN * (Cfg::TM * 32 * 128) + (size_t)Cfg::WM * 2 * Cfg::BN * sizeof(float));
    static_assert(!BNIN || ring >= (size_t)Cfg::WM * Cfg::WN * (Cfg::TM * 32 * 128) + (size_t)Cfg::WM * 2 * Cfg::BN * sizeof(float), "epilogue areas end below the table");
    if (lds > 65536) {
        constexpr size_t full = std::max(ring, (size_t)Cfg::WM * Cfg::WN * (Cfg::TM * 32 * 128) + (size_t)Cfg::WM * 2 * Cfg::BN * sizeof(float));
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&head_gemm_kernel<OUT_F32, Cfg, MODE, RED, BNIN>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, BNIN ? (int)(ring + 2048 * sizeof(float2)) : (int)full);
        if (attr != hipSuccess) return EPI_ERR_LAUNCH;
    }
    const dim3 grid((unsigned)pl.tiles, (unsigned)pl.nsplit, (unsigned)nphase);
    hipLaunchKernelGGL((head_gemm_kernel<OUT_F32, Cfg, MODE, RED, BNIN>), grid, dim3(Cfg::THREADS), lds, st, a);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}
template <bool OUT_F32, typename Cfg, bool RED = false>
static int launch_gemm_cfg(const GemmArgs& a, const GemmPlan& pl, int nphase, hipStream_t st) {
    if (a.ph.enabled) return launch_gemm_mode<OUT_F32, Cfg, A_PHASED, RED>(a, pl, nphase, st);
    if (a.ga.enabled) return launch_gemm_mode<OUT_F32, Cfg, A_GATHER, RED>(a, pl, nphase, st);
    return launch_gemm_mode<OUT_F32, Cfg, A_PLAIN, RED>(a, pl, nphase, st);
}

#ifdef EPI_GEMM_TRACE
extern "C" int epi_gemm_trace_read(unsigned long long* out, int clear) {
    const size_t bytes = sizeof(unsigned long long) * epi::GEMM_TRACE_WGS * epi::GEMM_TRACE_SLOTS;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(epi::epi_gemm_trace), bytes) != hipSuccess) return 1;
    if (clear) {
        static unsigned long long zeros[epi::GEMM_TRACE_WGS * epi::GEMM_TRACE_SLOTS];
        if (hipMemcpyToSymbol(HIP_SYMBOL(epi::epi_gemm_trace), zeros, bytes) != hipSuccess) return 1;
    }
    return 0;
}
#endif

#ifdef EPI_PATCH_TRACE
extern "C" int epi_patch_trace_read(unsigned long long* out, int clear) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(epi::epi_patch_trace), sizeof(unsigned long long) * 64 * 96) != hipSuccess) return 1;
    if (clear) {
        static unsigned long long zeros[64 * 96];
        if (hipMemcpyToSymbol(HIP_SYMBOL(epi::epi_patch_trace), zeros, sizeof(zeros)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

// ---- conv_patch_kernel: eligibility, plan, launch ----
enum { PATCH_WIDE = 0, PATCH_NARROW = 1, PATCH_HALF = 2 };
struct PatchPlan { bool ok; int cfg, nb, patch_px, nsplit, cps; long long tiles; size_t lds; bool single; };
// EPI_CONV3X3_PATCH: 0 never (the generic gather kernel), 1 only where the tiles fill the chip without a channel split, 2 (default) always
static int g_patch_mode = -1;
static int patch_mode() {
    if (g_patch_mode < 0) g_patch_mode = 2;
    return g_patch_mode;
}
// tuning / test hook: set the mode (0 .. 2; anything else only queries); returns the mode in force before the call
extern "C" int epi_conv3x3_patch_mode(int mode) {
    const int before = patch_mode();
    if (mode >= 0 && mode <= 2) g_patch_mode = mode;
    return before;
}
static PatchPlan patch_plan(int M, int N, int Cs, int W, size_t workspace_bytes) {
    PatchPlan pl = {};
    const int nchunks = Cs / GBK;
    // tile: 256 x 64 for the 64-channel layers; 256 x 128 when that already fills the chip, else 128 x 128 (twice the tiles)
    pl.cfg = N <= 64 ? PATCH_NARROW : PATCH_WIDE;
    if (pl.cfg == PATCH_WIDE && (long long)((M + 255) / 256) * ((N + 127) / 128) < 200) pl.cfg = PATCH_HALF;
    const int bm = pl.cfg == PATCH_HALF ? 128 : 256, bn = pl.cfg == PATCH_NARROW ? 64 : 128;
    const int tm = pl.cfg == PATCH_WIDE ? 2 : 1, wmv = pl.cfg == PATCH_NARROW ? 8 : 4;
    pl.patch_px = (bm + 2 * W + 2 + 7) / 8 * 8;
    if (pl.patch_px > 8 * 8 * PATCH_APW) return pl;                 // one patch piece per wave and tap step
    const size_t epi = (size_t)8 * tm * 32 * 128 + (size_t)wmv * 2 * bn * sizeof(float);
    pl.single = pl.cfg == PATCH_NARROW && nchunks == 1;             // (a wide tile's registers do not allow two workgroups per CU)
    for (pl.nb = 3; pl.nb >= 2; --pl.nb) {
        pl.lds = std::max((size_t)(pl.single ? 1 : 2) * pl.patch_px * 128 + (size_t)pl.nb * bn * 128 + 1024 + 128, epi);
        if (pl.lds <= 160 * 1024) break;
    }
    if (pl.nb < 2) return pl;
    pl.tiles = (long long)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
    pl.nsplit = 1;
    if (pl.tiles < 120) {                                           // (measured: 128 unsplit workgroups of 36 steps beat 4 splits + finish)
        if (patch_mode() < 2) return pl;                            // would need a channel split: left to the generic kernel
        pl.nsplit = (int)std::min<long long>(nchunks, std::max<long long>(1, 256 / pl.tiles));
    }
    while (pl.nsplit > 1 && (size_t)pl.nsplit * M * N * sizeof(float) > workspace_bytes) --pl.nsplit;
    pl.cps = (nchunks + pl.nsplit - 1) / pl.nsplit;
    pl.nsplit = (nchunks + pl.cps - 1) / pl.cps;
    pl.ok = pl.tiles <= 0x7fffffffLL;
    return pl;
}
static bool patch_eligible(const GemmArgs& a, bool out_f32, int nphase) {
    if (patch_mode() == 0 || out_f32 || nphase != 1 || !a.ga.enabled || a.ph.enabled || a.sc.enabled || a.bias || !a.coalesce) return false;
    if (a.ga.stride != 1 || a.ga.Hg != a.ga.Hs || a.ga.Wg != a.ga.Ws || a.ga.Cs % GBK || a.K != PATCH_TAPS * a.ga.Cs || a.ga.pitch) return false;
    for (int t = 0; t < PATCH_TAPS; ++t)
        if (a.ga.dy[t] < -1 || a.ga.dy[t] > 1 || a.ga.dx[t] < -1 || a.ga.dx[t] > 1) return false;
    return (long long)a.M * a.ga.Cs < (1LL << 31) && gemm_tile_override() == 0;
}
template <typename Cfg, int NB, bool SINGLE = false, bool RED = false>
static int launch_patch_red(const GemmArgs& a, const PatchPlan& pl, hipStream_t st) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_patch_kernel<Cfg, NB, SINGLE, RED>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr != hipSuccess) return EPI_ERR_LAUNCH;
    hipLaunchKernelGGL((conv_patch_kernel<Cfg, NB, SINGLE, RED>), dim3((unsigned)pl.tiles, (unsigned)pl.nsplit), dim3(Cfg::THREADS), pl.lds, st, a);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}
template <typename Cfg, int NB, bool SINGLE = false>
static int launch_patch(const GemmArgs& a, const PatchPlan& pl, hipStream_t st) {
    return a.br.z ? launch_patch_red<Cfg, NB, SINGLE, true>(a, pl, st) : launch_patch_red<Cfg, NB, SINGLE, false>(a, pl, st);
}

// stats_done (may be null): set to 1 when a.stats was accumulated by the GEMM launch itself (unsplit bf16 result), else 0 --
// the caller then computes the statistics with its own pass
// red_done (may be null): likewise for a.br (GemmBnRed): 1 when the launch masked the gradient and accumulated the two BatchNorm-backward
// sums, 0 when the result is the plain gradient (split launches, fp32 results, uncoalesced rows) and the caller runs its reduction pass
static int launch_gemm(GemmArgs a, bool out_f32, int nphase, void* workspace, size_t workspace_bytes, hipStream_t st, int* stats_done = nullptr,
                       int* red_done = nullptr) {
    if (stats_done) *stats_done = 0;
    if (red_done) *red_done = 0;
    a.store_policy = gemm_store_policy();
    // EPI_BN_BWD_FUSE=0: never (A/B measurements)
    const bool fuse_red = true;
    GemmBnRed want_red = a.br;
    a.br = GemmBnRed{};
    // EPI_BN_BWD_FUSE_MAX_ROWS / _MIN_ROWS: fuse only for outputs of at most / at least that many rows (all phases together) -- measurement switches
    const long long red_max_rows = 1LL << 40;
    const long long red_min_rows = 0;
    if (deterministic() || !fuse_red || !red_done || !want_red.z || !want_red.bn || !want_red.sums || a.stats || a.bias ||
        ((reinterpret_cast<uintptr_t>(want_red.z) | reinterpret_cast<uintptr_t>(want_red.y)) & 15u) || (long long)a.M * nphase > red_max_rows ||
        (long long)a.M * nphase < red_min_rows)
        want_red.z = nullptr;
    if (a.addend && (out_f32 || (reinterpret_cast<uintptr_t>(a.addend) & 15u))) return EPI_ERR_UNSUPPORTED;
    a.coalesce = (!out_f32 && a.N % 8 == 0 && a.ldc % 8 == 0) ? 1 : 0;
    const bool half_addend = a.addend && a.addend_step == 2;      // only the generic kernel's coalesced epilogue reads it: plain rows, unsplit
    if (half_addend && (!a.coalesce || a.ga.enabled || a.sc.enabled || a.ph.enabled || nphase != 1 || a.add_h <= 0 || a.add_w <= 0 || ((a.add_h | a.add_w) & 1) ||
                        (long long)a.M % ((long long)a.add_h * a.add_w)))
        return EPI_ERR_UNSUPPORTED;
    // BatchNorm statistics from the GEMM epilogue (on; EPI_FUSE_BN_STATS=0 keeps the separate statistics pass: 8.46 vs 8.49 ms/step,
    // 471 vs 507 launches, profiles/r02_fs_steady_state_d_*).  First version, measured and rejected: per-WAVE
    // atomics, 8 lanes x 8 strided columns per instruction -- 16 (instruction, line) pairs per wave and line, which L2 serialises at
    // ~9 ns each (layer-1 convolutions 26 us -> 230 .. 430 us, profiles/r02_fs_steady_state_x_*).  Now: one contiguous atomic
    // instruction per workgroup after an LDS combine (as the standalone statistics kernel does).
    const bool fuse_stats = true;
    float* const want_stats = (fuse_stats && !deterministic()) ? a.stats : nullptr;      // (deterministic mode: no column sums by atomics, csrc/capi.hip)
    a.stats = nullptr;
    if (!a.A || !a.Bt || !a.C || a.M <= 0 || a.N <= 0 || a.K <= 0) return EPI_ERR_INVALID_ARGUMENT;
    if (a.K % 8 || a.ldb % 8 || a.ldc % 4 || (!a.ga.enabled && a.lda % 8)) return EPI_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(a.A) | reinterpret_cast<uintptr_t>(a.Bt) | reinterpret_cast<uintptr_t>(a.C)) & 15u) return EPI_ERR_UNSUPPORTED;
    if (a.ga.enabled && (a.ga.Cs % GBK)) return EPI_ERR_UNSUPPORTED;     // a K tile must not straddle two taps
    if (a.bn_in_on) {
        // BatchNorm + ReLU on the A operand (plain rows, bf16 result): the 128 x 128 two-stage instantiation, statistics of the result from its epilogue
        if (out_f32 || a.ga.enabled || a.sc.enabled || a.ph.enabled || nphase != 1 || a.K % GBK || a.K > 2048 || want_red.z || a.bias || a.addend || !a.coalesce ||
            !a.bn_in.gamma || !a.bn_in.beta || !a.bn_in.scale || !a.bn_in.shift)
            return EPI_ERR_UNSUPPORTED;
        const GemmPlan pl = gemm_plan_cfg(CFG_SMALL, a.M, a.N, a.K, 1);
        if (pl.tiles > 0x7fffffffLL) return EPI_ERR_UNSUPPORTED;
        if (pl.nsplit > 1) {
            if (!workspace || (size_t)pl.nsplit * a.M * a.N * sizeof(float) > workspace_bytes) return EPI_ERR_WORKSPACE;
            a.slabs = (float*)workspace;
        }
        a.k_per_split = pl.kps;
        if (want_stats && pl.nsplit == 1) { a.stats = want_stats; if (stats_done) *stats_done = 1; }
        const int rc = launch_gemm_mode<false, CfgSmall, A_PLAIN, false, true>(a, pl, 1, st);
        if (rc != EPI_OK) return rc;
        if (pl.nsplit > 1) {
            if (want_stats && a.N % 4 == 0) {
                a.stats = want_stats;
                if (stats_done) *stats_done = 1;
                const unsigned blocks = (unsigned)(((a.M + FS_ROWS - 1) / FS_ROWS) * ((a.N + 255) / 256));
                hipLaunchKernelGGL(splitk_finish_stats_kernel, dim3(blocks), dim3(256), 0, st, a.slabs, pl.nsplit, a);
            } else {
                const long long n = (long long)a.M * (a.N >> 2);
                hipLaunchKernelGGL(splitk_finish_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a.slabs, pl.nsplit, 1, a);
            }
            EPI_CHECK_LAUNCH();
        }
        return EPI_OK;
    }
    // short K, wide N, plain operands, bf16 result: the A-stationary kernel (the final 1x1 convolution forward)
    // (a launch that carries the fused BatchNorm-backward reduction stays on the generic kernel: the A-stationary kernel has no registers left
    //  to request a slice's z / y ahead of its MFMAs, and with them requested at their use it ran 84 .. 92 us instead of 19 .. 20 us --
    //  profiles/r03_bn_bwd_fused_reduction.txt)
    // A launch that wants BatchNorm statistics leaves the A-stationary kernel for the generic one, whose epilogue delivers them: 6.530 vs 6.562
    // ms/step against the A-stationary kernel + a separate statistics pass over its output (forward convolutions +46 us, BatchNorm -64 us and 8
    // launches; EPI_STATS_OFF_ASTAT=0 restores the round-2 arrangement; statistics from the A-stationary epilogue itself cost more than either)
    const bool stats_off_astat = true;
    if (!want_red.z && !(want_stats && stats_off_astat && !a.bias) && !half_addend &&
        !out_f32 && !a.ga.enabled && !a.sc.enabled && nphase == 1 && (a.K == 64 || a.K == 128 || a.K == 256) && a.N % 8 == 0 &&
        a.ldc % 8 == 0 && a.N >= 4 * AS_BN && a.N <= 8192 && a.M >= 64 * AS_BM && gemm_tile_override() == 0) {
        const unsigned grid = (unsigned)((a.M + AS_BM - 1) / AS_BM);
        // BatchNorm statistics from THIS kernel's epilogue are off by default (EPI_FUSE_BN_STATS_ASTAT=1 turns them on): in the step trace the
        // A-stationary launches with statistics take 39 / 37 us (K = 64 / 128) against 21 / 20 us without -- more than the separate statistics
        // pass over their output costs (14 / 8 us); measured in the step, same box: 7.06 ms without, 7.13 ms with
        const bool astat_stats = false;
        if (want_stats && !a.bias && astat_stats) { a.stats = want_stats; if (stats_done) *stats_done = 1; }
#define EPI_ASTAT(KS)                                                                                                      \
        do {                                                                                                               \
            const size_t lds = AS_RING * (size_t)AS_BN * 32 * KS + AS_WAVES * 4096 + (size_t)((a.N + AS_BN - 1) / AS_BN) * AS_BN * 4 * 3;                                          \
            static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&head_gemm_astat_kernel<KS>),  \
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      \
            if (attr != hipSuccess) return EPI_ERR_LAUNCH;                                                                 \
            hipLaunchKernelGGL((head_gemm_astat_kernel<KS>), dim3(grid), dim3(AS_THREADS), lds, st, a);                     \
        } while (0)
        if (a.K == 256) EPI_ASTAT(16);
        else if (a.K == 128) EPI_ASTAT(8);
        else EPI_ASTAT(4);
#undef EPI_ASTAT
        EPI_CHECK_LAUNCH();
        return EPI_OK;
    }
    if (patch_eligible(a, out_f32, nphase)) {           // 3x3 / stride 1: the patch-stationary kernel
        const PatchPlan pp = patch_plan(a.M, a.N, a.ga.Cs, a.ga.Wg, workspace ? workspace_bytes : 0);
        if (pp.ok) {
            a.patch_px = pp.patch_px;
            a.chunks_per_split = pp.cps;
            if (pp.nsplit > 1) a.slabs = (float*)workspace;
            else if (want_stats) { a.stats = want_stats; if (stats_done) *stats_done = 1; }
            else if (want_red.z) { a.br = want_red; *red_done = 1; }
            int rc;
            if (pp.cfg == PATCH_NARROW && pp.single) rc = pp.nb == 3 ? launch_patch<PatchNarrow, 3, true>(a, pp, st) : launch_patch<PatchNarrow, 2, true>(a, pp, st);
            else if (pp.cfg == PATCH_NARROW) rc = pp.nb == 3 ? launch_patch<PatchNarrow, 3>(a, pp, st) : launch_patch<PatchNarrow, 2>(a, pp, st);
            else if (pp.cfg == PATCH_HALF) rc = pp.nb == 3 ? launch_patch<PatchHalf, 3>(a, pp, st) : launch_patch<PatchHalf, 2>(a, pp, st);
            else rc = pp.nb == 3 ? launch_patch<PatchWide, 3>(a, pp, st) : launch_patch<PatchWide, 2>(a, pp, st);
            if (rc != EPI_OK) return rc;
            if (pp.nsplit > 1) {
                if (want_stats && !a.addend) {            // the finish also accumulates the BatchNorm batch sums
                    a.stats = want_stats;
                    if (stats_done) *stats_done = 1;
                    const unsigned blocks = (unsigned)(((a.M + FS_ROWS - 1) / FS_ROWS) * ((a.N + 255) / 256));
                    hipLaunchKernelGGL(splitk_finish_stats_kernel, dim3(blocks), dim3(256), 0, st, a.slabs, pp.nsplit, a);
                } else {
                    const long long n = (long long)a.M * (a.N >> 2);
                    hipLaunchKernelGGL(splitk_finish_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a.slabs, pp.nsplit, 1, a);
                }
                EPI_CHECK_LAUNCH();
            }
            return EPI_OK;
        }
    }
    // (round 4, measured and not kept: split launches finished by the workgroup that completes a tile -- arrival counter per tile, partials through
    //  agent-scope (L2-bypassing) accesses, the last arriver sums them in split order and runs the whole epilogue with the fused column sums: 12 finish
    //  and 6 reduction launches per step gone, the step 6.17 -> 6.75 ms (7.15 ms with __threadfence() around ordinary accesses: a whole-L2 write-back
    //  and invalidate per workgroup).  A finish launch spreads over the whole chip and reads the partials from L2; the last arrivers are 64 .. 256
    //  workgroups reading them past the L2, serially behind their own K loop.  profiles/r04_ab_split_finished_in_kernel.txt)
    // (round 4, measured and not kept: under-filled launches with fused column sums -- 64 .. 128 tiles of 128 x 128 in layers 3 / 4 at batch 32 -- unsplit on
    //  64 x 128 tiles with the reduction in their epilogue: 3 .. 5 separate reduction launches go, the step gets 2 % SLOWER (6.25 -> 6.39 ms; with the 3x3
    //  launches moved off the patch kernel as well 6.62 ms), profiles/r04_ab_underfill_half_tiles.txt)
    // (a launch that carries the fused BatchNorm-backward reduction has a pipelined instantiation on the 128 x 128 tile only: a plan that comes
    //  back with another tile AND the ring is redone without the ring)
    GemmPlan pl = gemm_plan(a.M, a.N, a.K, a.ldc, nphase, out_f32);
    if (want_red.z && pl.pipe && pl.cfg != CFG_SMALL) pl = gemm_plan(a.M, a.N, a.K, a.ldc, nphase, out_f32, false);
    if (pl.tiles > 0x7fffffffLL) return EPI_ERR_UNSUPPORTED;
    if (half_addend && pl.nsplit > 1) return EPI_ERR_UNSUPPORTED;          // (epi_conv2d_bwd_data_half_addend_ok tells the caller beforehand)
    if (pl.nsplit > 1) {
        if (!workspace || (size_t)pl.nsplit * nphase * a.M * a.N * sizeof(float) > workspace_bytes) return EPI_ERR_WORKSPACE;
        a.slabs = (float*)workspace;
    }
    a.k_per_split = pl.kps;
    if (want_stats && pl.nsplit == 1 && a.coalesce && !a.bias) { a.stats = want_stats; if (stats_done) *stats_done = 1; }
    if (want_red.z && pl.nsplit == 1 && a.coalesce && (!pl.pipe || pl.cfg == CFG_SMALL) && (pl.cfg == CFG_BIG || pl.cfg == CFG_SMALL || pl.cfg == CFG_TALL)) {
        a.br = want_red;
        *red_done = 1;
    }
    int rc;
    if (a.br.z) rc = pl.cfg == CFG_BIG ? launch_gemm_cfg<false, CfgBig, true>(a, pl, nphase, st)
                   : (pl.cfg == CFG_TALL ? launch_gemm_cfg<false, CfgTall, true>(a, pl, nphase, st)
                      : (pl.pipe ? launch_gemm_cfg<false, CfgSmallP, true>(a, pl, nphase, st) : launch_gemm_cfg<false, CfgSmall, true>(a, pl, nphase, st)));
    else if (pl.cfg == CFG_BIG) rc = g_tile_force == 5 ? launch_gemm_cfg<false, CfgBigLock>(a, pl, nphase, st) : launch_gemm_cfg<false, CfgBig>(a, pl, nphase, st);
    else if (out_f32) rc = launch_gemm_cfg<true, CfgSmall>(a, pl, nphase, st);
    else if (pl.cfg == CFG_HALF) rc = launch_gemm_cfg<false, CfgHalf>(a, pl, nphase, st);
    else if (pl.cfg == CFG_QUARTER) rc = launch_gemm_cfg<false, CfgQuarter>(a, pl, nphase, st);
    else if (pl.cfg == CFG_TALL) rc = pl.pipe ? launch_gemm_cfg<false, CfgTallP>(a, pl, nphase, st) : launch_gemm_cfg<false, CfgTall>(a, pl, nphase, st);
    else rc = pl.pipe ? launch_gemm_cfg<false, CfgSmallP>(a, pl, nphase, st) : launch_gemm_cfg<false, CfgSmall>(a, pl, nphase, st);
    if (rc != EPI_OK) return rc;
    if (pl.nsplit > 1) {
        const long long n = (long long)nphase * a.M * (a.N >> 2);
        const dim3 fg((unsigned)((n + 255) / 256));
        if (!out_f32 && want_stats && nphase == 1 && !a.bias && !a.addend && !a.sc.enabled && !a.ph.enabled && a.N % 4 == 0) {
            a.stats = want_stats;                         // split-K convolution in front of a BatchNorm: statistics from the finish kernel
            if (stats_done) *stats_done = 1;
            const unsigned blocks = (unsigned)(((a.M + FS_ROWS - 1) / FS_ROWS) * ((a.N + 255) / 256));
            hipLaunchKernelGGL(splitk_finish_stats_kernel, dim3(blocks), dim3(256), 0, st, a.slabs, pl.nsplit, a);
        } else if (out_f32) hipLaunchKernelGGL(splitk_finish_kernel<true>, fg, dim3(256), 0, st, a.slabs, pl.nsplit, nphase, a);
        else hipLaunchKernelGGL(splitk_finish_kernel<false>, fg, dim3(256), 0, st, a.slabs, pl.nsplit, nphase, a);
        EPI_CHECK_LAUNCH();
    }
    return EPI_OK;
}

static GemmBnRed bnred_args(const EpiBnReduce* r) {
    GemmBnRed b = {};
    if (r && r->z && r->bn && r->sums) {
        b.z = (const unsigned short*)r->z; b.y = (const unsigned short*)r->y; b.bn = r->bn; b.sums = r->sums; b.relu = r->relu ? 1 : 0;
    }
    return b;
}

// epi_gemm_bf16 with bf16 C whose rows are the gradient of a BatchNorm(+ReLU) output (the final 1x1 convolution's backward-data):
// see EpiBnReduce (include/epipolar_hip.h).  *red_done = 1: C holds dz and red->sums the two column sums; 0: C is the plain product.
extern "C" int epi_gemm_bf16_bnred(const void* A, int lda, const void* Bt, int ldb, void* C, int ldc, int M, int N, int K, const EpiBnReduce* red,
                                   int* red_done, void* workspace, size_t workspace_bytes, epi_stream_t stream) {
    if (!red_done) return EPI_ERR_INVALID_ARGUMENT;
    GemmArgs a = {};
    a.A = (const unsigned short*)A; a.Bt = (const unsigned short*)Bt; a.C = C;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.br = bnred_args(red);
    return launch_gemm(a, false, 1, workspace, workspace_bytes, (hipStream_t)stream, nullptr, red_done);
}

extern "C" int epi_gemm_bf16(const void* A, int lda, const void* Bt, int ldb, void* C, int ldc, int c_dtype, int M, int N, int K,
                             const float* bias, void* workspace, size_t workspace_bytes, epi_stream_t stream) {
    GemmArgs a = {};
    a.A = (const unsigned short*)A; a.Bt = (const unsigned short*)Bt; a.C = C; a.bias = bias;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    if (c_dtype != EPI_BF16 && c_dtype != EPI_F32) return EPI_ERR_UNSUPPORTED;
    return launch_gemm(a, c_dtype == EPI_F32, 1, workspace, workspace_bytes, (hipStream_t)stream);
}

// ConvTranspose2d(k=4, s=2, p=1), NHWC bf16:  x [B][H][W][Cin]  ->  y [B][2H][2W][Cout]  (raw, pre-BatchNorm).
// w_phase: [4 phases][Cout][4 taps * Cin] packed by epi_deconv4x4s2_pack_weight (phase = 2*(oh&1) + (ow&1)).
// All four output-parity phases run in ONE launch (blockIdx.z); workspace: epi_gemm_workspace_bytes(B*H*W, Cout, 4*Cin, 4).
extern "C" int epi_deconv4x4s2_fwd(const void* x, const void* w_phase, void* y, int B, int H, int W, int Cin, int Cout,
                                   void* workspace, size_t workspace_bytes, epi_stream_t stream) {
    return epi_deconv4x4s2_fwd_stats(x, w_phase, y, B, H, W, Cin, Cout, nullptr, nullptr, workspace, workspace_bytes, stream);
}

// The same with the BatchNorm batch sums of the result (as epi_conv2d_fwd): bn_sums [epi_bn_sum_copies(Cout)][2 Cout] f32 += per-channel
// (sum, sum of squares) of the bf16 outputs when the launch can do it from its epilogue (*stats_done = 1), else untouched (*stats_done = 0).
static int deconv4x4s2_fwd_impl(const void* x, const void* w_phase, void* y, int B, int H, int W, int Cin, int Cout, float* bn_sums,
                               int* stats_done, void* workspace, size_t workspace_bytes, epi_stream_t stream, bool out_f32) {
    if (stats_done) *stats_done = 0;
    if (!x || !w_phase || !y || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return EPI_ERR_INVALID_ARGUMENT;
    if (Cin % GBK || Cout % 4) return EPI_ERR_UNSUPPORTED;
    GemmArgs a = {};
    a.A = (const unsigned short*)x; a.Bt = (const unsigned short*)w_phase; a.C = y;
    a.M = B * H * W; a.N = Cout; a.K = 4 * Cin; a.lda = 0; a.ldb = 4 * Cin; a.ldc = Cout;
    a.ga.enabled = 1; a.ga.Hg = H; a.ga.Wg = W; a.ga.Hs = H; a.ga.Ws = W; a.ga.Cs = Cin; a.ga.stride = 1;
    a.sc.enabled = 1; a.sc.Hg = H; a.sc.Wg = W; a.sc.Ho = 2 * H; a.sc.Wo = 2 * W; a.sc.so = 2;
    a.ph.enabled = 1;
    // oh = 2*ih - 1 + kh: tap (ty, tx) of phase (py, px) reads the input pixel (i + py - ty, j + px - tx)
    for (int phase = 0; phase < 4; ++phase) {
        a.ph.ntap[phase] = 4;
        a.ph.bt_off[phase] = (long long)phase * Cout * 4 * Cin;
        for (int t = 0; t < 4; ++t) { a.ph.dy[phase][t] = (phase >> 1) - (t >> 1); a.ph.dx[phase][t] = (phase & 1) - (t & 1); }
    }
    a.stats = out_f32 ? nullptr : bn_sums;
    a.stats_copies = epi_bn_sum_copies(Cout);
    return launch_gemm(a, out_f32, 4, workspace, workspace_bytes, (hipStream_t)stream, stats_done);
}
extern "C" int epi_deconv4x4s2_fwd_stats(const void* x, const void* w_phase, void* y, int B, int H, int W, int Cin, int Cout, float* bn_sums,
                                         int* stats_done, void* workspace, size_t workspace_bytes, epi_stream_t stream) {
    return deconv4x4s2_fwd_impl(x, w_phase, y, B, H, W, Cin, Cout, bn_sums, stats_done, workspace, workspace_bytes, stream, false);
}
// fp32 result (y float [B][2H][2W][Cout]): the fp32-grade verification mode (csrc/precise.hip), operands bf16 as always
extern "C" int epi_deconv4x4s2_fwd_f32(const void* x, const void* w_phase, void* y, int B, int H, int W, int Cin, int Cout,
                                       void* workspace, size_t workspace_bytes, epi_stream_t stream) {
    return deconv4x4s2_fwd_impl(x, w_phase, y, B, H, W, Cin, Cout, nullptr, nullptr, workspace, workspace_bytes, stream, true);
}

// Backward-data of the same layer:  dy [B][2H][2W][Cout] -> dx [B][H][W][Cin];
// w_bwd: [Cin][16 taps * Cout] packed by epi_deconv4x4s2_pack_weight (tap = kh*4 + kw).
// workspace: epi_gemm_workspace_bytes(B*H*W, Cin, 16*Cout, 1).
static int deconv4x4s2_bwd_data_impl(const void* dy, const void* w_bwd, void* dx, int B, int H, int W, int Cin, int Cout,
                                    void* workspace, size_t workspace_bytes, epi_stream_t stream, bool out_f32,
                                    const EpiBnReduce* red = nullptr, int* red_done = nullptr) {
    if (red_done) *red_done = 0;
    if (!dy || !w_bwd || !dx || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return EPI_ERR_INVALID_ARGUMENT;
    if (Cout % GBK || Cin % 4) return EPI_ERR_UNSUPPORTED;
    GemmArgs a = {};
    a.A = (const unsigned short*)dy; a.Bt = (const unsigned short*)w_bwd; a.C = dx;
    a.M = B * H * W; a.N = Cin; a.K = 16 * Cout; a.lda = 0; a.ldb = 16 * Cout; a.ldc = Cin;
    a.ga.enabled = 1; a.ga.Hg = H; a.ga.Wg = W; a.ga.Hs = 2 * H; a.ga.Ws = 2 * W; a.ga.Cs = Cout; a.ga.stride = 2;
    for (int kh = 0; kh < 4; ++kh)
        for (int kw = 0; kw < 4; ++kw) { a.ga.dy[4 * kh + kw] = kh - 1; a.ga.dx[4 * kh + kw] = kw - 1; }   // oh = 2*ih - 1 + kh
    a.br = bnred_args(red);
    return launch_gemm(a, out_f32, 1, workspace, workspace_bytes, (hipStream_t)stream, nullptr, red_done);
}
extern "C" int epi_deconv4x4s2_bwd_data(const void* dy, const void* w_bwd, void* dx, int B, int H, int W, int Cin, int Cout,
                                        void* workspace, size_t workspace_bytes, epi_stream_t stream) {
    return deconv4x4s2_bwd_data_impl(dy, w_bwd, dx, B, H, W, Cin, Cout, workspace, workspace_bytes, stream, false);
}
// the same with the fused BatchNorm-backward reduction of the layer that produced this deconvolution's input (EpiBnReduce)
extern "C" int epi_deconv4x4s2_bwd_data_bnred(const void* dy, const void* w_bwd, void* dx, int B, int H, int W, int Cin, int Cout,
                                              const EpiBnReduce* red, int* red_done, void* workspace, size_t workspace_bytes, epi_stream_t stream) {
    if (!red_done) return EPI_ERR_INVALID_ARGUMENT;
    return deconv4x4s2_bwd_data_impl(dy, w_bwd, dx, B, H, W, Cin, Cout, workspace, workspace_bytes, stream, false, red, red_done);
}
extern "C" int epi_deconv4x4s2_bwd_data_f32(const void* dy, const void* w_bwd, void* dx, int B, int H, int W, int Cin, int Cout,
                                            void* workspace, size_t workspace_bytes, epi_stream_t stream) {
    return deconv4x4s2_bwd_data_impl(dy, w_bwd, dx, B, H, W, Cin, Cout, workspace, workspace_bytes, stream, true);
}

namespace epi {
// weight [Cin][Cout][4][4] (any float dtype converted by the caller to bf16) -> both packed forms
__global__ void pack_deconv_weight_kernel(const unsigned short* __restrict__ w, int Cin, int Cout,
                                          unsigned short* __restrict__ w_phase, unsigned short* __restrict__ w_bwd) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)Cin * Cout * 16;
    if (t >= total) return;
    const int kw = (int)(t & 3), kh = (int)((t >> 2) & 3);
    const long long r = t >> 4;
    const int co = (int)(r % Cout), ci = (int)(r / Cout);
    const unsigned short v = w[t];
    if (w_bwd) w_bwd[(long long)ci * 16 * Cout + (kh * 4 + kw) * Cout + co] = v;
    if (w_phase) {
        // kh odd <-> output row parity 0 (kh=1: dy 0 -> tap row 0, kh=3: dy -1 -> tap row 1); kh even <-> parity 1
        const int ph = (kh & 1) ? 0 : 1, ty = ph ? (kh == 0 ? 0 : 1) : (kh == 1 ? 0 : 1);
        const int pw = (kw & 1) ? 0 : 1, tx = pw ? (kw == 0 ? 0 : 1) : (kw == 1 ? 0 : 1);
        w_phase[((long long)(2 * ph + pw) * Cout + co) * 4 * Cin + (2 * ty + tx) * Cin + ci] = v;
    }
}
}  // namespace epi

extern "C" int epi_deconv4x4s2_pack_weight(const void* w_bf16, int Cin, int Cout, void* w_phase, void* w_bwd, epi_stream_t stream) {
    if (!w_bf16 || Cin <= 0 || Cout <= 0 || (!w_phase && !w_bwd)) return EPI_ERR_INVALID_ARGUMENT;
    const long long total = (long long)Cin * Cout * 16;
    hipLaunchKernelGGL(epi::pack_deconv_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)w_bf16, Cin, Cout, (unsigned short*)w_phase, (unsigned short*)w_bwd);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// "TN" GEMM for the weight gradients:  C[i][j] = sum_r A[r][i] * B(r)[j]   (reduction over the ROW index of both
// operands: r enumerates batch*pixels).  Both tiles are row-major [r][cols] images of global memory filled by DMA, and
// the MFMA operands (8 consecutive r per lane) are produced by the LDS transpose read ds_read_b64_tr_b16
// (a 16-lane group reads a 4(r) x 16(col) block; lane c receives column c).
// Convolution weight gradients are ONE such GEMM with the filter taps laid along j:  column j = tap*Cs + c reads the
// activation pixel shifted by that tap (an implicit gather: every lane of a DMA piece supplies its own source address), so
// the result [I][ntap*Cs] IS the channels_last weight gradient [Cout][KH][KW][Cin] and the dy tile is staged once for all
// taps.  The reduction is split across blockIdx.y only as far as needed to give every CU a workgroup; a split writes an fp32
// slab (summed by slab_reduce_kernel), an unsplit launch writes the result directly.
// ---------------------------------------------------------------------------------------------------------------
namespace epi {

typedef short s16x4 __attribute__((ext_vector_type(4)));

struct GemmTnArgs {
    const unsigned short* A;      // [R][lda]
    const unsigned short* B;      // plain [R][ldb], or the gather source [n][Hs][Ws][ldb]
    void* C;                      // result [I][J] (f32 or bf16) when nsplit == 1, else fp32 slabs [nsplit][I][J]
    int R, I, J, lda, ldb;        // J = ntap * Cs for a gather
    int rows_per_split, nsplit;
    int out_bf16;                 // dtype of the direct (unsplit) result
    int gather;                   // 0: B row r, column j.  1: rows enumerate (n, ih, iw) over Hg x Wg; column j = tap*Cs + c reads
    int Hg, Wg, Hs, Ws, Cs;       //    B[n][ih*stride + kh - pad][iw*stride + kw - pad][c],  tap = kh*KW + kw
    int stride, pad, KW;
    const float* b_bn;            // plain B only (grouped launches): B holds the RAW output z of the convolution in front and the operand is
                                  // relu(z * b_bn[j] + b_bn[J + j]) -- the normalised activation the forward pass never wrote (GemmArgs::bn_in)
};

// Tile configurations (64 reduction rows per K tile; UNPADDED row-major tiles of COLS*2 bytes per row, filled by DMA):
//   small:  128(i) x 128(j), 4 waves (2 x 2, each 64 x 64),  64 KiB LDS, 2 workgroups / CU
//   narrow:  64(i) x 128(j), 4 waves (2 x 2, each 32 x 64),  48 KiB LDS, 3 workgroups / CU   (64 output channels: layer1)
//   big:    256(i) x 256(j), 8 waves (2 x 4, each 128 x 64), 128 KiB LDS, 1 workgroup / CU  (half the operand traffic per flop)
// Bank conflicts of the transpose reads are avoided by XOR-ing the 16-byte chunk index with a function of the row -- applied on
// the SOURCE address of the DMA, like the NT kernel: a 32-lane ds_read_b64_tr_b16 group touches rows r..r+3 at two adjacent
// 32-byte column blocks, which then fall into disjoint 64-byte bank windows.
template <int TI_, int WJ_> struct TnCfg {
    static constexpr int TI = TI_, WJ = WJ_, NW = 2 * WJ_, THREADS = 64 * NW;
    static constexpr int BI = 2 * TI_ * 32, BJ = WJ_ * 64;
    static constexpr int A_ROWB = BI * 2, B_ROWB = BJ * 2;                      // bytes per tile row
    static constexpr int A_BYTES = GBK * A_ROWB, B_BYTES = GBK * B_ROWB, STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int A_PIECES = A_BYTES / 1024 / NW, B_PIECES = B_BYTES / 1024 / NW;   // 1 KiB DMA pieces per wave and K tile
    static constexpr int WG_PER_CU = (2 * STAGE_BYTES <= 49152) ? 3 : (2 * STAGE_BYTES <= 65536 ? 2 : 1);
    static_assert(A_PIECES >= 1 && B_PIECES >= 1 && A_PIECES <= 4 && B_PIECES <= 4, "pieces per wave");
};
typedef TnCfg<2, 2> TnSmall;
typedef TnCfg<1, 2> TnNarrow;
typedef TnCfg<4, 4> TnBig;

// chunk swizzle of a tile row: rows that share banks (256-byte bank row) get different 64-byte windows
template <int ROWB> __device__ __forceinline__ int tn_swz(int row) { return ROWB >= 256 ? 4 * (row & 3) : 4 * ((row >> 1) & 1); }

// MFMA operand (8 consecutive r per lane) by two transpose reads: a 16-lane group reads a 4(r) x 16(col) block, lane c
// receives column c.  Lane c addresses row row0 + (c >> 2) (and +4), 4 columns at col + 4*(c & 3); col % 32 == 0, row0 % 8 == 0.
template <int ROWB>
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int row0, int col, int lane_c) {
    const int row = row0 + (lane_c >> 2);
    const int chunk = (col >> 3) + ((lane_c & 3) >> 1);
    const char* p = tile + row * ROWB + ((chunk ^ tn_swz<ROWB>(row)) << 4) + 8 * (lane_c & 1);
    struct { s16x4 lo, hi; } v;
    v.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
    v.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * ROWB));     // row + 4: same swizzle term
    return __builtin_bit_cast(bf16x8, v);
}

// One workgroup's share of a TN GEMM: output tile `tile_id` (tile_j fastest), reduction slice `split`.  Shared by the single-GEMM
// kernel and the grouped kernel (many weight gradients in one launch).
// eight consecutive r of ONE column (a lane's B operand) through relu(z * sc + sh): the arithmetic of bn_apply2d_kernel
__device__ __forceinline__ bf16x8 tn_bn_apply(bf16x8 b, float sc, float sh) {
    const uint4v u = __builtin_bit_cast(uint4v, b);
    const unsigned int w[4] = {u.x, u.y, u.z, u.w};
    unsigned int o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float lo = __uint_as_float(w[k] << 16), hi = __uint_as_float(w[k] & 0xffff0000u);
        o[k] = pack_bf16x2(fmaxf(lo * sc + sh, 0.f), fmaxf(hi * sc + sh, 0.f));
    }
    uint4v r; r.x = o[0]; r.y = o[1]; r.z = o[2]; r.w = o[3];
    return __builtin_bit_cast(bf16x8, r);
}

template <typename Cfg, bool GATHER, bool BNB = false>
__device__ __forceinline__ void tn_body(const GemmTnArgs& p, const int tile_id, const int split, char* smem) {
    static_assert(!(GATHER && BNB), "BatchNorm on the B operand: plain rows only");
    constexpr int TI = Cfg::TI, BI = Cfg::BI, BJ = Cfg::BJ, AP = Cfg::A_PIECES, BP = Cfg::B_PIECES;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / Cfg::WJ, wn = wid % Cfg::WJ;
    const int tiles_j = (p.J + BJ - 1) / BJ;
    const int tile_i = tile_id / tiles_j, tile_j = tile_id - tile_i * tiles_j;
    const int i0 = tile_i * BI, j0 = tile_j * BJ;
    const int r_begin = split * p.rows_per_split;
    const int r_end = min(p.R, r_begin + p.rows_per_split);
    const char* zero_src = reinterpret_cast<const char*>(epi_zero_chunk);

    // ---- DMA roles: piece q of wave w covers tile rows (w*PIECES + q) * RPP .. + RPP-1 (RPP = 1024 / row bytes); lane l lands at
    //      row base + l / CH, physical chunk l % CH and fetches logical chunk (l % CH) ^ swz(row).  Everything that does not change
    //      from K tile to K tile is computed once: the column (and with it the filter tap and its pixel shift), and the (n, ih, iw)
    //      of the lane's row, which then ADVANCES by 64 rows per K tile with two conditional carries -- no division in the loop. ----
    constexpr int A_CH = Cfg::A_ROWB / 16, A_RPP = 1024 / Cfg::A_ROWB, B_CH = Cfg::B_ROWB / 16, B_RPP = 1024 / Cfg::B_ROWB;
    int a_row[AP], b_row[BP];
    const unsigned short* a_src[AP];
    bool a_colok[AP], b_colok[BP];
    const unsigned short* b_src[BP];          // plain: B + col;  gather: B + channel within the tap
    int b_n[BP];                              // gather: image index of the lane's row in the current K tile (b_ih / b_iw below: its pixel)
#pragma unroll
    for (int q = 0; q < AP; ++q) {
        a_row[q] = (wid * AP + q) * A_RPP + lane / A_CH;
        const int col = i0 + (((lane % A_CH) ^ tn_swz<Cfg::A_ROWB>(a_row[q])) << 3);
        a_colok[q] = col < p.I;
        a_src[q] = p.A + (long long)(r_begin + a_row[q]) * p.lda + col;
    }
    // advance of (iw, ih, n) per 64 rows
    const int adv_w = GATHER ? GBK % p.Wg : 0, adv_h = GATHER ? (GBK / p.Wg) % p.Hg : 0, adv_n = GATHER ? GBK / (p.Wg * p.Hg) : 0;
    int b_ih[BP], b_iw[BP], b_sy[BP], b_sx[BP];
#pragma unroll
    for (int q = 0; q < BP; ++q) {
        b_row[q] = (wid * BP + q) * B_RPP + lane / B_CH;
        const int col = j0 + (((lane % B_CH) ^ tn_swz<Cfg::B_ROWB>(b_row[q])) << 3);
        b_colok[q] = col < p.J;
        if (GATHER) {
            const int tap = col / p.Cs, c = col - tap * p.Cs;
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
            b_sy[q] = kh - p.pad;
            b_sx[q] = kw - p.pad;
            const int r = r_begin + b_row[q], hw = p.Hg * p.Wg;
            b_n[q] = r / hw;
            const int rem = r - b_n[q] * hw;
            b_ih[q] = rem / p.Wg;
            b_iw[q] = rem - b_ih[q] * p.Wg;
            b_src[q] = p.B + c;
        } else {
            b_src[q] = p.B + (long long)(r_begin + b_row[q]) * p.ldb + col;
            b_n[q] = b_ih[q] = b_iw[q] = b_sy[q] = b_sx[q] = 0;
        }
    }
    // issue the DMA of one K tile (rows r0 .. r0+63) into buffer buf and step the row state to the next K tile
    // (pointers stay immutable -- per-tile offsets are wave-uniform scalars -- so that the arrays live in registers, not scratch)
    auto issue_tile_piece_a = [&](int q, int r0, int buf) {
        char* a_s = smem + buf * Cfg::STAGE_BYTES + __builtin_amdgcn_readfirstlane(wid) * (AP * 1024);
        const bool ok = a_colok[q] && r0 + a_row[q] < r_end;
        const unsigned short* src = a_src[q] + (long long)(r0 - r_begin) * p.lda;
        glds16(ok ? reinterpret_cast<const void*>(src) : reinterpret_cast<const void*>(zero_src), a_s + q * 1024);
    };
    auto issue_tile_piece_b = [&](int q, int r0, int buf) {
        char* b_s = smem + buf * Cfg::STAGE_BYTES + Cfg::A_BYTES + __builtin_amdgcn_readfirstlane(wid) * (BP * 1024);
        bool ok = b_colok[q] && r0 + b_row[q] < r_end;
        const unsigned short* src;
        if (GATHER) {                // 32-bit element offsets (the launcher refuses sources of 2^31 elements or more)
            const int y = b_ih[q] * p.stride + b_sy[q], x = b_iw[q] * p.stride + b_sx[q];
            ok = ok && (unsigned)y < (unsigned)p.Hs && (unsigned)x < (unsigned)p.Ws;
            const unsigned pix = (unsigned)((b_n[q] * p.Hs + y) * p.Ws + x);
            src = b_src[q] + pix * (unsigned)p.ldb;
            b_iw[q] += adv_w;
            if (b_iw[q] >= p.Wg) { b_iw[q] -= p.Wg; b_ih[q] += 1; }
            b_ih[q] += adv_h;
            if (b_ih[q] >= p.Hg) { b_ih[q] -= p.Hg; b_n[q] += 1; }
            b_n[q] += adv_n;
        } else {
            src = b_src[q] + (long long)(r0 - r_begin) * p.ldb;
        }
        glds16(ok ? reinterpret_cast<const void*>(src) : reinterpret_cast<const void*>(zero_src), b_s + q * 1024);
    };
    auto issue_tile = [&](int r0, int buf) {
#pragma unroll
        for (int q = 0; q < AP; ++q) issue_tile_piece_a(q, r0, buf);
#pragma unroll
        for (int q = 0; q < BP; ++q) issue_tile_piece_b(q, r0, buf);
    };

    f32x16 acc[TI][2];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = r_end > r_begin ? (r_end - r_begin + GBK - 1) / GBK : 0;
    if (nk > 0) issue_tile(r_begin, 0);
    __syncthreads();
    const int lane_c = lane & 15, grp = lane >> 4;
    const int cblk = 16 * (grp & 1), khalf = grp >> 1;
    float bsc[2] = {0.f, 0.f}, bsh[2] = {0.f, 0.f};     // BNB: the lane's B column (one channel per tj) keeps its (scale, shift) for the whole reduction
    if (BNB) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int col = j0 + wn * 64 + t * 32 + cblk + lane_c;
            if (col < p.J) { bsc[t] = p.b_bn[col]; bsh[t] = p.b_bn[p.J + col]; }
        }
    }
    auto read_frags = [&](const char* a_s, const char* b_s, int ks, bf16x8 (&af)[TI], bf16x8 (&bfr)[2]) {
#pragma unroll
        for (int t = 0; t < TI; ++t) af[t] = tr_frag<Cfg::A_ROWB>(a_s, ks * 16 + 8 * khalf, wm * (TI * 32) + t * 32 + cblk, lane_c);
#pragma unroll
        for (int t = 0; t < 2; ++t) bfr[t] = tr_frag<Cfg::B_ROWB>(b_s, ks * 16 + 8 * khalf, wn * 64 + t * 32 + cblk, lane_c);
    };
    // one K tile (64 reduction rows): the NEXT tile's DMA is issued first (the whole tile's MFMA time hides its latency; its
    // buffer was released by the barrier that ended the previous tile), fragments of k step ks+1 are requested before the MFMAs
    // of step ks
    auto k_tile = [&](int kt, auto has_next_tag) {
        constexpr bool HAS_NEXT = decltype(has_next_tag)::value;
        const int buf = kt & 1;
        const char* a_s = smem + buf * Cfg::STAGE_BYTES;
        const char* b_s = a_s + Cfg::A_BYTES;
        bf16x8 af[2][TI], bfr[2][2];
        read_frags(a_s, b_s, 0, af[0], bfr[0]);
        if (HAS_NEXT) issue_tile(r_begin + (kt + 1) * GBK, buf ^ 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < 3) read_frags(a_s, b_s, ks + 1, af[(ks + 1) & 1], bfr[(ks + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            if (BNB) {
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) bfr[ks & 1][tj] = tn_bn_apply(bfr[ks & 1][tj], bsc[tj], bsh[tj]);
            }
#pragma unroll
            for (int ti = 0; ti < TI; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj)
                    acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][ti], bfr[ks & 1][tj], acc[ti][tj], 0, 0, 0);
        }
        __syncthreads();
    };
    for (int kt = 0; kt + 1 < nk; ++kt) k_tile(kt, std::true_type());
    if (nk > 0) k_tile(nk - 1, std::false_type());
    // D[i][j]: lane holds column j = lane & 31, rows i = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
    const int fcol = lane & 31, fhalf = lane >> 5;
    const bool direct = p.nsplit == 1;
    float* slab = reinterpret_cast<float*>(p.C) + (direct ? 0 : (long long)split * p.I * p.J);
    unsigned short* out16 = reinterpret_cast<unsigned short*>(p.C);
    // (the output type is wave-uniform: decided once, not per element)
    auto store_tile = [&](auto bf16_c) __attribute__((always_inline)) {
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) {
                const int j = j0 + wn * 64 + tj * 32 + fcol;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int i = i0 + wm * (TI * 32) + ti * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * fhalf;
                    if (i < p.I && j < p.J) {
                        if (decltype(bf16_c)::value) out16[(long long)i * p.J + j] = f32_to_bf16(acc[ti][tj][reg]);
                        else slab[(long long)i * p.J + j] = acc[ti][tj][reg];
                    }
                }
            }
    };
    if (direct && p.out_bf16) store_tile(std::true_type{}); else store_tile(std::false_type{});
}

template <typename Cfg, bool GATHER, bool BNB = false>
__global__ __launch_bounds__(Cfg::THREADS, Cfg::WG_PER_CU * Cfg::THREADS / 256) void head_gemm_tn_kernel(GemmTnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [2 buffers][A tile | B tile]
    // logical id: tile fastest, then split: the tiles of one split (same rows r) stay on one XCD
    const int total_wg = gridDim.x * gridDim.y;
    const int lid = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, total_wg);
    tn_body<Cfg, GATHER, BNB>(p, lid % (int)gridDim.x, lid / (int)gridDim.x, smem);
}

// MANY weight gradients in one launch (a whole ResNet stage's backward-weight GEMMs, csrc/torch_glue.cpp): the deep layers at batch 32 have
// 16 .. 144 output tiles and 2048 .. 8192 reduction rows each -- alone, each of them had to cut its reduction into 4 .. 16 slices to give
// every CU a workgroup, and every slice wrote an fp32 slab of the whole result (866 MB of slabs per ResNet-50 step,
// profiles/r02_wgrad_plan_slabs.txt).  Together the tiles of ~10 .. 20 layers fill the chip with UNSPLIT reductions; the workgroups of
// row r are [wg_begin, wg_begin + tiles * nsplit), tile fastest.  Rows live in the kernel-argument segment (scalar loads).
struct TnGroupRow { GemmTnArgs a; int wg_begin, tiles; };
constexpr int TN_GROUP_MAX = 32;
struct TnGroupArgs { int nrows, reserved[3]; TnGroupRow rows[TN_GROUP_MAX]; };
static_assert(sizeof(TnGroupArgs) <= 4096, "kernel-argument segment");

template <typename Cfg>
__global__ __launch_bounds__(Cfg::THREADS, Cfg::WG_PER_CU * Cfg::THREADS / 256) void head_gemm_tn_group_kernel(TnGroupArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    int row = 0;
    for (int r = 1; r < g.nrows; ++r) row = g.rows[r].wg_begin <= lid ? r : row;        // (uniform: scalar loads and compares)
    const TnGroupRow& gr = g.rows[row];
    const int local = lid - gr.wg_begin;
    const int split = local / gr.tiles, tile_id = local - split * gr.tiles;
    if (gr.a.gather) tn_body<Cfg, true>(gr.a, tile_id, split, smem);
    else if (gr.a.b_bn) tn_body<Cfg, false, true>(gr.a, tile_id, split, smem);
    else tn_body<Cfg, false>(gr.a, tile_id, split, smem);
}

// out[e] = sum over the splits; out_bf16: bf16 result.  256 threads = 32 element pairs x 8 split lanes: every lane sums
// nsplit / 8 slabs (independent loads in flight), the 8 partial sums meet through shuffles (lanes 8 apart hold the same pair).
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, int nsplit, long long n, void* __restrict__ out,
                                                          int out_bf16) {
    const int pair = threadIdx.x & 7, sl = threadIdx.x >> 3;            // 8 pairs per 64-byte line x 32 split lanes
    const long long e = ((long long)blockIdx.x * 8 + pair) * 2;
    float s0 = 0.f, s1 = 0.f;
    if (e < n) {
        for (int k = sl; k < nsplit; k += 32) {
            const float2 v = *reinterpret_cast<const float2*>(slabs + (long long)k * n + e);
            s0 += v.x; s1 += v.y;
        }
    }
    __shared__ float2 part[32][8];
    part[sl][pair].x = s0;
    part[sl][pair].y = s1;
    __syncthreads();
    if (sl == 0 && e < n) {
#pragma unroll 8
        for (int k = 1; k < 32; ++k) { s0 += part[k][pair].x; s1 += part[k][pair].y; }
        if (out_bf16) reinterpret_cast<unsigned int*>(out)[e >> 1] = pack_bf16x2(s0, s1);
        else { float2 o; o.x = s0; o.y = s1; *reinterpret_cast<float2*>(reinterpret_cast<float*>(out) + e) = o; }
    }
}

// few splits: one thread per element pair, every slab read in turn (coalesced 8-byte loads)
__global__ void slab_reduce_few_kernel(const float* __restrict__ slabs, int nsplit, long long n, void* __restrict__ out, int out_bf16) {
    const long long e = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (e >= n) return;
    float s0 = 0.f, s1 = 0.f;
    for (int k = 0; k < nsplit; ++k) {
        const float2 v = *reinterpret_cast<const float2*>(slabs + (long long)k * n + e);
        s0 += v.x; s1 += v.y;
    }
    if (out_bf16) reinterpret_cast<unsigned int*>(out)[e >> 1] = pack_bf16x2(s0, s1);
    else { float2 o; o.x = s0; o.y = s1; *reinterpret_cast<float2*>(reinterpret_cast<float*>(out) + e) = o; }
}

// MANY pending reductions in one launch (every split weight gradient of a backward pass): blockIdx.x = chunk of 512 elements; the
// row (result tensor) a chunk belongs to is found by bisection over the rows' first chunks
__global__ __launch_bounds__(256) void slab_reduce_multi_kernel(const EpiSlabReduce* __restrict__ rows, int nrows) {
    int lo = 0, hi = nrows - 1;
    while (lo < hi) {                                   // last row with chunk_begin <= blockIdx.x (uniform: scalar loads)
        const int mid = (lo + hi + 1) >> 1;
        if (rows[mid].chunk_begin <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const EpiSlabReduce r = rows[lo];
    const long long e = (((long long)blockIdx.x - r.chunk_begin) * 256 + threadIdx.x) * 2;
    if (e >= r.n) return;
    float s0 = 0.f, s1 = 0.f;
    const float* src = r.slabs + e;
    int k = 0;
    for (; k + 3 < r.nsplit; k += 4) {                  // four slabs in flight
        const float2 a = *reinterpret_cast<const float2*>(src + (long long)k * r.n), b = *reinterpret_cast<const float2*>(src + (long long)(k + 1) * r.n);
        const float2 c = *reinterpret_cast<const float2*>(src + (long long)(k + 2) * r.n), d = *reinterpret_cast<const float2*>(src + (long long)(k + 3) * r.n);
        s0 += (a.x + b.x) + (c.x + d.x); s1 += (a.y + b.y) + (c.y + d.y);
    }
    for (; k < r.nsplit; ++k) {
        const float2 v = *reinterpret_cast<const float2*>(src + (long long)k * r.n);
        s0 += v.x; s1 += v.y;
    }
    if (r.out_bf16) reinterpret_cast<unsigned int*>(r.out)[e >> 1] = pack_bf16x2(s0, s1);
    else { float2 o; o.x = s0; o.y = s1; *reinterpret_cast<float2*>(reinterpret_cast<float*>(r.out) + e) = o; }
}

}  // namespace epi

struct TnPlan { int cfg; long long tiles; int nsplit, rps; };      // cfg: 0 small, 1 narrow, 2 big

// Tile configuration and reduction split by a small cost model (microseconds, calibrated on MI355X with tools/bench_conv.py):
// a workgroup spends ~t_tile per 64-row K tile plus a fixed prologue / epilogue; the chip holds `slots` workgroups at once; every
// split writes an fp32 slab of the whole result that the reduce kernel reads back (~3 TB/s through L2 / MALL, + one launch).
static TnPlan tn_plan(int R, int I, int J) {
    struct Cfg { int id, bi, bj, slots; double t_tile, t_fixed; };
    static const Cfg cfgs[3] = {{0, 128, 128, 512, 1.0, 4.0}, {1, 64, 128, 768, 0.7, 3.0}, {2, 256, 256, 256, 2.0, 8.0}};
    static const int cand[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256};
    auto fills = [](int n) { const int t = (n + 255) / 256; return n >= 192 && t * 256 <= n + n / 4; };
    const int ov = gemm_tile_override();
    // EPI_TN_SLOTS=<percent>: plan as if the chip held that share of its workgroup slots -- a smaller footprint for the per-layer weight-gradient
    // launches (deconvolution head, final convolution, stem), which run beside the backward chain on the second stream.  Default 70 since round 3:
    // 6.534 -> 6.462 ms/step over three boxes (100: 6.529 / 6.539 / 6.474, 70: 6.440 / 6.484 / 6.415; 60: 6.424, 50: 6.463, 40: 6.527).  Round 2
    // measured the opposite on its per-layer backbone launches (7.67 / 7.73 vs 7.65 ms for 50 / 70): those are grouped now (group_plan).
    const int slot_percent = 70;
    // Long reductions over large operands (the final layer: 131 072 rows x (1088 + 256) columns; the last deconvolution: 32 768 x (256 + 16 x 256)): the
    // launch is bound by the LDS fill traffic, which the 256 x 256 tile halves (every operand row is staged once per 256 instead of 128 output columns /
    // rows) -- measured alone with HBM-fresh operands (tools/bench_tn_tiles.py): 195 -> 165 us and 138 -> 125 us; shorter reductions lose 25 .. 35 % on
    // that tile (2048 / 8192 rows), and the cost model below, calibrated on those, never picks it.  EPI_TN_BIG_LONG=0: the model alone.
    const bool big_long_on = true;
    const bool big_long = big_long_on && ov == 0 && R >= 32768 && fills(I) && fills(J) && (long long)I * J >= 256 * 1024;
    TnPlan best = {0, 0, 1, 0};
    double best_t = 1e30;
    for (const Cfg& c : cfgs) {
        if (big_long && c.id != 2) continue;
        if (c.id == 1 && I > 64) continue;
        if (c.id == 0 && I <= 64 && ov == 0) continue;
        if (c.id == 2 && !((fills(I) && fills(J)) || ov == 2)) continue;
        if ((ov == 1 && c.id == 2) || (ov == 2 && c.id != 2)) continue;
        const long long tiles = (long long)((I + c.bi - 1) / c.bi) * ((J + c.bj - 1) / c.bj);
        for (int ns : cand) {
            if (ns > 1 && (long long)ns * 128 > R) break;
            int rps = ((R + ns - 1) / ns + GBK - 1) / GBK * GBK;
            const int nsplit = (R + rps - 1) / rps;
            const long long slots = (long long)c.slots * slot_percent / 100;
            const double rounds = (double)((tiles * nsplit + slots - 1) / slots);
            const double t_main = rounds * ((rps / GBK) * c.t_tile + c.t_fixed);
            const double t_slab = nsplit > 1 ? (double)nsplit * I * J * 8.0 / 3.0e6 + 3.0 : 0.0;
            if (t_main + t_slab < best_t) { best_t = t_main + t_slab; best = {c.id, tiles, nsplit, rps}; }
        }
    }
    return best;
}

template <typename Cfg, bool GATHER, bool BNB = false>
static int launch_tn_cfg(const GemmTnArgs& a, const TnPlan& pl, hipStream_t st) {
    const size_t lds = 2 * Cfg::STAGE_BYTES;
    if (lds > 65536) {
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&head_gemm_tn_kernel<Cfg, GATHER, BNB>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (attr != hipSuccess) return EPI_ERR_LAUNCH;
    }
    hipLaunchKernelGGL((head_gemm_tn_kernel<Cfg, GATHER, BNB>), dim3((unsigned)pl.tiles, (unsigned)pl.nsplit), dim3(Cfg::THREADS), lds, st, a);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}

// defer (may be null): when the reduction is split, leave the slabs in slab_ws and describe the pending reduce instead of launching it
// (nsplit = 0 in *defer: nothing pending, `out` is complete)
static int launch_tn(GemmTnArgs a, void* out, int out_bf16, float* slab_ws, size_t slab_bytes, hipStream_t st, EpiSlabReduce* defer = nullptr) {
    if (defer) { *defer = EpiSlabReduce(); }
    if (!a.A || !a.B || !out || a.R <= 0 || a.I <= 0 || a.J <= 0) return EPI_ERR_INVALID_ARGUMENT;
    if (a.I % 8 || a.J % 8 || a.lda % 8 || a.ldb % 8 || (a.gather && a.Cs % 8)) return EPI_ERR_UNSUPPORTED;
    if (a.gather && (long long)((a.R + a.Hg * a.Wg - 1) / (a.Hg * a.Wg)) * a.Hs * a.Ws * a.ldb >= (1LL << 31)) return EPI_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(a.A) | reinterpret_cast<uintptr_t>(a.B)) & 15u) return EPI_ERR_UNSUPPORTED;
    const TnPlan pl = tn_plan(a.R, a.I, a.J);
    const long long n = (long long)a.I * a.J;
    a.rows_per_split = pl.rps;
    a.nsplit = pl.nsplit;
    a.out_bf16 = out_bf16;
    if (pl.nsplit > 1) {
        if (!slab_ws || (size_t)pl.nsplit * n * sizeof(float) > slab_bytes) return EPI_ERR_WORKSPACE;
        a.C = slab_ws;
    } else {
        a.C = out;
    }
    int rc;
    if (a.b_bn && a.gather) return EPI_ERR_UNSUPPORTED;
    if (a.b_bn) rc = pl.cfg == 2 ? launch_tn_cfg<TnBig, false, true>(a, pl, st)
                     : (pl.cfg == 1 ? launch_tn_cfg<TnNarrow, false, true>(a, pl, st) : launch_tn_cfg<TnSmall, false, true>(a, pl, st));
    else if (pl.cfg == 2) rc = a.gather ? launch_tn_cfg<TnBig, true>(a, pl, st) : launch_tn_cfg<TnBig, false>(a, pl, st);
    else if (pl.cfg == 1) rc = a.gather ? launch_tn_cfg<TnNarrow, true>(a, pl, st) : launch_tn_cfg<TnNarrow, false>(a, pl, st);
    else rc = a.gather ? launch_tn_cfg<TnSmall, true>(a, pl, st) : launch_tn_cfg<TnSmall, false>(a, pl, st);
    if (rc != EPI_OK) return rc;
    if (pl.nsplit > 1 && defer) {
        defer->slabs = slab_ws; defer->out = out; defer->n = n; defer->nsplit = pl.nsplit; defer->out_bf16 = out_bf16;
        return EPI_OK;
    }
    if (pl.nsplit > 1) {
        if (pl.nsplit >= 16 && n / 2 < 262144)       // many slabs of a small result: spread the split dimension over the threads
            hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((n / 2 + 7) / 8)), dim3(256), 0, st, slab_ws, pl.nsplit, n, out, out_bf16);
        else
            hipLaunchKernelGGL(slab_reduce_few_kernel, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, st, slab_ws, pl.nsplit, n, out, out_bf16);
        EPI_CHECK_LAUNCH();
    }
    return EPI_OK;
}

// The plan a weight-gradient launch of this shape would use (host only, no device work): plan[0] tile configuration (0: 128 x 128,
// 1: 64 x 128, 2: 256 x 256), plan[1] output tiles, plan[2] reduction splits (= fp32 slabs), plan[3] rows per split.
extern "C" int epi_gemm_tn_plan(int R, int I, int J, int ntap, long long* plan) {
    if (!plan || R <= 0 || I <= 0 || J <= 0 || ntap <= 0) return EPI_ERR_INVALID_ARGUMENT;
    const TnPlan pl = tn_plan(R, I, J * ntap);
    plan[0] = pl.cfg; plan[1] = pl.tiles; plan[2] = pl.nsplit; plan[3] = pl.rps;
    return EPI_OK;
}

extern "C" size_t epi_gemm_tn_workspace_bytes(int R, int I, int J, int ntap) {
    if (R <= 0 || I <= 0 || J <= 0 || ntap <= 0) return 0;
    const TnPlan pl = tn_plan(R, I, J * ntap);
    return pl.nsplit > 1 ? (size_t)pl.nsplit * ntap * I * J * sizeof(float) : 0;
}

// C[I][J] (f32) = A[R][I]^T * B[R][J]   (weight gradient of the 1x1 convolution: A = dlogits, B = activations)
extern "C" int epi_gemm_tn_bf16(const void* A, int lda, const void* B, int ldb, float* C, int R, int I, int J, void* workspace,
                                size_t workspace_bytes, epi_stream_t stream) {
    GemmTnArgs a = {};
    a.A = (const unsigned short*)A; a.B = (const unsigned short*)B; a.R = R; a.I = I; a.J = J; a.lda = lda; a.ldb = ldb;
    return launch_tn(a, C, 0, (float*)workspace, workspace_bytes, (hipStream_t)stream);
}

// dw[Cin][16 taps][Cout] (f32 or bf16, tap = kh*4 + kw: the memory order of a channels_last [Cin, Cout, 4, 4] weight) of
// ConvTranspose2d(k4 s2 p1): x [B][H][W][Cin], dy [B][2H][2W][Cout]
extern "C" int epi_deconv4x4s2_bwd_weight(const void* x, const void* dy, void* dw_taps, int dw_dtype, int B, int H, int W, int Cin, int Cout,
                                          void* workspace, size_t workspace_bytes, epi_stream_t stream) {
    if (B <= 0 || H <= 0 || W <= 0) return EPI_ERR_INVALID_ARGUMENT;
    if (dw_dtype != EPI_F32 && dw_dtype != EPI_BF16) return EPI_ERR_UNSUPPORTED;
    GemmTnArgs a = {};
    a.A = (const unsigned short*)x; a.B = (const unsigned short*)dy; a.R = B * H * W; a.I = Cin; a.J = 16 * Cout; a.lda = Cin; a.ldb = Cout;
    // the input pixel (ih, iw) reaches the output pixels (2*ih - 1 + kh, 2*iw - 1 + kw)
    a.gather = 1; a.Hg = H; a.Wg = W; a.Hs = 2 * H; a.Ws = 2 * W; a.Cs = Cout; a.stride = 2; a.pad = 1; a.KW = 4;
    return launch_tn(a, dw_taps, dw_dtype == EPI_BF16, (float*)workspace, workspace_bytes, (hipStream_t)stream);
}

// Weight gradient of a Conv2d (groups = 1, dilation = 1), NHWC bf16:  x [B][H][W][Cin], dy [B][Ho][Wo][Cout]  ->
// dw [Cout][KH][KW][Cin] (the memory order of a channels_last weight; for KH = KW = 1 also the contiguous one), f32 or bf16.
//   dW[co][kh][kw][ci] = sum over (n, oh, ow) of dy[n][oh][ow][co] * x[n][oh*stride + kh - pad][ow*stride + kw - pad][ci]
// = the TN GEMM with A = dy (plain rows) and B = x gathered per tap along the columns.  Replaces MIOpen's split-K wrw kernels
// and the memset / zero-fill / cast launches around them.  workspace: epi_gemm_tn_workspace_bytes(B*Ho*Wo, Cout, Cin, KH*KW).
extern "C" int epi_conv2d_bwd_weight_deferred(const void* x, const void* dy, void* dw, int dw_dtype, int B, int H, int W, int Cin, int Cout, int KH,
                                              int KW, int stride, int pad, void* workspace, size_t workspace_bytes, EpiSlabReduce* pending,
                                              epi_stream_t stream) {
    if (pending) *pending = EpiSlabReduce();
    if (B <= 0 || H <= 0 || W <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0) return EPI_ERR_INVALID_ARGUMENT;
    if (dw_dtype != EPI_F32 && dw_dtype != EPI_BF16) return EPI_ERR_UNSUPPORTED;
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return EPI_ERR_INVALID_ARGUMENT;
    GemmTnArgs a = {};
    a.A = (const unsigned short*)dy; a.B = (const unsigned short*)x; a.R = B * Ho * Wo; a.I = Cout; a.J = KH * KW * Cin; a.lda = Cout; a.ldb = Cin;
    if (!(KH == 1 && KW == 1 && stride == 1 && pad == 0)) {
        a.gather = 1; a.Hg = Ho; a.Wg = Wo; a.Hs = H; a.Ws = W; a.Cs = Cin; a.stride = stride; a.pad = pad; a.KW = KW;
    }
    return launch_tn(a, dw, dw_dtype == EPI_BF16, (float*)workspace, workspace_bytes, (hipStream_t)stream, pending);
}

extern "C" int epi_conv2d_bwd_weight(const void* x, const void* dy, void* dw, int dw_dtype, int B, int H, int W, int Cin, int Cout, int KH,
                                     int KW, int stride, int pad, void* workspace, size_t workspace_bytes, epi_stream_t stream) {
    return epi_conv2d_bwd_weight_deferred(x, dy, dw, dw_dtype, B, H, W, Cin, Cout, KH, KW, stride, pad, workspace, workspace_bytes, nullptr, stream);
}

// ---- grouped weight gradients: many convolution / deconvolution backward-weight GEMMs in ONE launch per tile class ----
namespace {

struct GroupRowPlan { GemmTnArgs a; int cls; long long tiles; int ktiles, nsplit, rps; long long n; };
struct GroupPlan { GroupRowPlan rows[epi::TN_GROUP_MAX]; int nrows; size_t slab_bytes; };

// the TN operands of one item (what epi_conv2d_bwd_weight / epi_deconv4x4s2_bwd_weight build)
int group_item_args(const EpiWgradItem& it, GemmTnArgs* out) {
    if (it.B <= 0 || it.H <= 0 || it.W <= 0 || it.Cin <= 0 || it.Cout <= 0) return EPI_ERR_INVALID_ARGUMENT;
    if (it.dw_dtype != EPI_F32 && it.dw_dtype != EPI_BF16) return EPI_ERR_UNSUPPORTED;
    GemmTnArgs a = {};
    if (it.kind == EPI_WGRAD_DECONV4X4S2) {
        a.A = (const unsigned short*)it.x; a.B = (const unsigned short*)it.dy; a.R = it.B * it.H * it.W; a.I = it.Cin; a.J = 16 * it.Cout;
        a.lda = it.Cin; a.ldb = it.Cout;
        a.gather = 1; a.Hg = it.H; a.Wg = it.W; a.Hs = 2 * it.H; a.Ws = 2 * it.W; a.Cs = it.Cout; a.stride = 2; a.pad = 1; a.KW = 4;
    } else if (it.kind == EPI_WGRAD_CONV2D) {
        if (it.KH <= 0 || it.KW <= 0 || it.stride <= 0 || it.pad < 0) return EPI_ERR_INVALID_ARGUMENT;
        const int Ho = (it.H + 2 * it.pad - it.KH) / it.stride + 1, Wo = (it.W + 2 * it.pad - it.KW) / it.stride + 1;
        if (Ho <= 0 || Wo <= 0) return EPI_ERR_INVALID_ARGUMENT;
        a.A = (const unsigned short*)it.dy; a.B = (const unsigned short*)it.x; a.R = it.B * Ho * Wo; a.I = it.Cout; a.J = it.KH * it.KW * it.Cin;
        a.lda = it.Cout; a.ldb = it.Cin;
        if (!(it.KH == 1 && it.KW == 1 && it.stride == 1 && it.pad == 0)) {
            if (it.x_scale_shift) return EPI_ERR_UNSUPPORTED;          // (an un-normalised input: 1x1 / stride-1 layers only)
            a.gather = 1; a.Hg = Ho; a.Wg = Wo; a.Hs = it.H; a.Ws = it.W; a.Cs = it.Cin; a.stride = it.stride; a.pad = it.pad; a.KW = it.KW;
        }
        a.b_bn = it.x_scale_shift;
    } else {
        return EPI_ERR_UNSUPPORTED;
    }
    if (it.kind != EPI_WGRAD_CONV2D && it.x_scale_shift) return EPI_ERR_UNSUPPORTED;
    a.out_bf16 = it.dw_dtype == EPI_BF16;
    if (a.I % 8 || a.J % 8 || a.lda % 8 || a.ldb % 8 || (a.gather && a.Cs % 8)) return EPI_ERR_UNSUPPORTED;
    if (a.gather && (long long)((a.R + a.Hg * a.Wg - 1) / (a.Hg * a.Wg)) * a.Hs * a.Ws * a.ldb >= (1LL << 31)) return EPI_ERR_UNSUPPORTED;
    *out = a;
    return EPI_OK;
}

// Reduction slices per row.  Model (microseconds, the constants of tn_plan): a workgroup costs t_fixed + t_tile per 64-row K tile; the chip
// runs `slots` workgroups of a class at once; a launch takes max(total workgroup time / resident workgroups, the longest workgroup); every
// slice of a split row writes an fp32 slab that the deferred sum reads back.  One knob for the whole group: T = K tiles per workgroup
// (rows with fewer K tiles stay unsplit), chosen per tile class because each class is its own launch.
int group_plan(const EpiWgradItem* items, int n, GroupPlan* gp) {
    if (!items || n <= 0 || n > epi::TN_GROUP_MAX) return EPI_ERR_INVALID_ARGUMENT;
    gp->nrows = n;
    gp->slab_bytes = 0;
    for (int r = 0; r < n; ++r) {
        GroupRowPlan& row = gp->rows[r];
        const int rc = group_item_args(items[r], &row.a);
        if (rc != EPI_OK) return rc;
        row.cls = row.a.I <= 64 ? 1 : 0;                     // 1: 64 x 128 tiles (64 output channels), 0: 128 x 128
        const int bi = row.cls ? 64 : 128;
        row.tiles = (long long)((row.a.I + bi - 1) / bi) * ((row.a.J + 127) / 128);
        row.ktiles = (row.a.R + GBK - 1) / GBK;
        row.n = (long long)row.a.I * row.a.J;
    }
    static const int cand[] = {2, 3, 4, 6, 8, 12, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 256, 384, 512, 768, 1024, 1536, 2048, 4096, 1 << 30};
    // the choice depends on the shapes only: remembered per (class, shapes) -- a training step asks for the same four groups every pass
    static std::mutex memo_lock;
    struct MemoEntry { std::vector<long long> shapes; int T; };     // the shapes the choice was made for: checked on a hit (a key collision recomputes)
    static std::unordered_map<unsigned long long, MemoEntry> memo;
    static const size_t MEMO_MAX = 256;                             // a training step asks for four groups; evaluation at many batch sizes must not grow it
    std::vector<long long> shapes;
    for (int cls = 0; cls < 2; ++cls) {
        const double t_tile = cls ? 0.7 : 1.0, t_fixed = cls ? 3.0 : 4.0;
        // EPI_TN_GROUP_SLOTS=<percent>: the same knob for the grouped launches (measured: 100 / 70 / 50 / 35 -> 6.440 / 6.441 / 6.466 / 6.509 ms: off)
        const int group_percent = 100;
        // EPI_TN_GROUP_MODEL=0: the round-3 estimate, total workgroup time / resident workgroups -- blind to the SECOND ROUND a launch of 860
        // workgroups needs on 768 slots (layer 1 of ResNet-50 at batch 32: the launch took 144 us for 302 MB, 2.1 TB/s, alone on the chip).
        // 1 (default, round 4): the makespan of the launch's workgroups on `slots` slots, longest first (the order they are launched in)
        const bool lpt_model = true;
        const long long slots = (cls ? 768 : 512) * group_percent / 100;
        unsigned long long key = 1469598103934665603ULL ^ (unsigned long long)(cls + 2 * group_percent + 1000 * (lpt_model ? 1 : 0));
        int members = 0;
        shapes.assign(1, cls);
        for (int r = 0; r < n; ++r) {
            const GroupRowPlan& row = gp->rows[r];
            if (row.cls != cls) continue;
            ++members;
            for (long long v : {row.tiles, (long long)row.ktiles, row.n}) { key ^= (unsigned long long)v; key *= 1099511628211ULL; shapes.push_back(v); }
        }
        if (members == 0) continue;
        int best_T = -1;
        {
            std::lock_guard<std::mutex> g(memo_lock);
            auto it = memo.find(key);
            if (it != memo.end() && it->second.shapes == shapes) best_T = it->second.T;
        }
        if (best_T < 0) {
            double best_t = 1e30;
            best_T = 1 << 30;
            std::vector<std::pair<double, long long>> jobs;
            std::vector<double> heap;
            for (int T : cand) {
                double work = 0, longest = 0, slab = 0;
                long long wgs = 0;
                jobs.clear();
                for (int r = 0; r < n; ++r) {
                    const GroupRowPlan& row = gp->rows[r];
                    if (row.cls != cls) continue;
                    const int ns = std::max(1, (row.ktiles + T - 1) / T), kt = (row.ktiles + ns - 1) / ns;
                    const double t_wg = kt * t_tile + t_fixed;
                    work += (double)row.tiles * ns * t_wg;
                    wgs += row.tiles * ns;
                    longest = std::max(longest, t_wg);
                    jobs.emplace_back(t_wg, row.tiles * ns);
                    if (ns > 1) slab += (double)ns * row.n * 8.0 / 3.0e6;
                }
                double span = std::max(work / (double)std::min(slots, wgs), longest);
                if (lpt_model && wgs > slots && wgs <= 16 * slots) {
                    std::sort(jobs.begin(), jobs.end(), [](const std::pair<double, long long>& x, const std::pair<double, long long>& y) { return x.first > y.first; });
                    heap.assign((size_t)slots, 0.0);                       // min-heap of the slots' free times
                    auto cmp = [](double x, double y) { return x > y; };
                    for (const auto& j : jobs)
                        for (long long c = 0; c < j.second; ++c) {
                            std::pop_heap(heap.begin(), heap.end(), cmp);
                            heap.back() += j.first;
                            std::push_heap(heap.begin(), heap.end(), cmp);
                        }
                    span = *std::max_element(heap.begin(), heap.end());
                }
                const double t = span + slab + (slab > 0 ? 3.0 : 0.0);
                if (t < best_t) { best_t = t; best_T = T; }
            }
            std::lock_guard<std::mutex> g(memo_lock);
            if (memo.size() >= MEMO_MAX) memo.clear();
            memo[key] = MemoEntry{shapes, best_T};
        }
        for (int r = 0; r < n; ++r) {
            GroupRowPlan& row = gp->rows[r];
            if (row.cls != cls) continue;
            int ns = std::max(1, (row.ktiles + best_T - 1) / best_T);
            row.rps = ((row.ktiles + ns - 1) / ns) * GBK;
            ns = (row.a.R + row.rps - 1) / row.rps;
            row.nsplit = ns;
            if (ns > 1) gp->slab_bytes += ((size_t)ns * row.n * sizeof(float) + 255) & ~(size_t)255;
        }
    }
    return EPI_OK;
}

}  // namespace

extern "C" int epi_wgrad_group_max(void) { return epi::TN_GROUP_MAX; }

extern "C" int epi_wgrad_group_plan(const EpiWgradItem* items, int n, size_t* slab_bytes, int* nsplit) {
    GroupPlan gp;
    const int rc = group_plan(items, n, &gp);
    if (rc != EPI_OK) return rc;
    if (slab_bytes) *slab_bytes = gp.slab_bytes;
    if (nsplit) for (int r = 0; r < n; ++r) nsplit[r] = gp.rows[r].nsplit;
    return EPI_OK;
}

template <typename Cfg>
static int launch_tn_group(const epi::TnGroupArgs& g, long long total_wg, hipStream_t st) {
    const size_t lds = 2 * Cfg::STAGE_BYTES;
    if (total_wg <= 0 || total_wg > 0x7fffffffLL) return EPI_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((epi::head_gemm_tn_group_kernel<Cfg>), dim3((unsigned)total_wg), dim3(Cfg::THREADS), lds, st, g);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}

// items[0 .. n): every weight gradient of the group (n <= epi_wgrad_group_max()); at most two launches (128 x 128 tiles; 64 x 128 tiles for
// the 64-output-channel layers).  slab_ws: epi_wgrad_group_plan's slab_bytes (256-byte aligned pieces, device memory that stays valid
// until the sums have run).  pending[r]: the outstanding slab sum of item r (nsplit == 0: dw is complete when the launch is) -- the
// caller runs them with epi_slab_reduce_multi.
extern "C" int epi_wgrad_group(const EpiWgradItem* items, int n, void* slab_ws, size_t slab_bytes, EpiSlabReduce* pending, epi_stream_t stream) {
    if (!pending) return EPI_ERR_INVALID_ARGUMENT;
    GroupPlan gp;
    const int rc = group_plan(items, n, &gp);
    if (rc != EPI_OK) return rc;
    if (gp.slab_bytes > 0 && (!slab_ws || gp.slab_bytes > slab_bytes)) return EPI_ERR_WORKSPACE;
    for (int r = 0; r < n; ++r) {
        if (!items[r].x || !items[r].dy || !items[r].dw) return EPI_ERR_INVALID_ARGUMENT;
        if ((reinterpret_cast<uintptr_t>(items[r].x) | reinterpret_cast<uintptr_t>(items[r].dy)) & 15u) return EPI_ERR_UNSUPPORTED;
    }
    size_t off = 0;
    for (int cls = 0; cls < 2; ++cls) {
        epi::TnGroupArgs g = {};
        long long wg = 0;
        // longest workgroups first: the tail of the launch is made of short ones
        int order[epi::TN_GROUP_MAX], m = 0;
        for (int r = 0; r < n; ++r) if (gp.rows[r].cls == cls) order[m++] = r;
        std::stable_sort(order, order + m, [&](int x, int y) { return gp.rows[x].rps > gp.rows[y].rps; });
        for (int k = 0; k < m; ++k) {
            const int r = order[k];
            GroupRowPlan& row = gp.rows[r];
            epi::TnGroupRow& gr = g.rows[g.nrows++];
            gr.a = row.a;
            gr.a.rows_per_split = row.rps;
            gr.a.nsplit = row.nsplit;
            pending[r] = EpiSlabReduce();
            if (row.nsplit > 1) {
                gr.a.C = static_cast<char*>(slab_ws) + off;
                pending[r].slabs = reinterpret_cast<const float*>(gr.a.C); pending[r].out = items[r].dw; pending[r].n = row.n;
                pending[r].nsplit = row.nsplit; pending[r].out_bf16 = row.a.out_bf16;
                off += ((size_t)row.nsplit * row.n * sizeof(float) + 255) & ~(size_t)255;
            } else {
                gr.a.C = items[r].dw;
            }
            gr.wg_begin = (int)wg;
            gr.tiles = (int)row.tiles;
            wg += row.tiles * row.nsplit;
        }
        if (g.nrows == 0) continue;
        const int lrc = cls ? launch_tn_group<epi::TnNarrow>(g, wg, (hipStream_t)stream) : launch_tn_group<epi::TnSmall>(g, wg, (hipStream_t)stream);
        if (lrc != EPI_OK) return lrc;
    }
    return EPI_OK;
}

// ONE item by itself (its own reduction split; the per-layer path of a gradient somebody reads inside the backward pass, and the fallback of a
// group whose slab arena is full): everything an item can describe, EpiWgradItem::x_scale_shift included.  workspace: epi_gemm_tn_workspace_bytes of
// the item's GEMM; pending as epi_conv2d_bwd_weight_deferred (NULL: the slab sum runs right behind the GEMM).
extern "C" int epi_wgrad_item(const EpiWgradItem* item, void* workspace, size_t workspace_bytes, EpiSlabReduce* pending, epi_stream_t stream) {
    if (pending) *pending = EpiSlabReduce();
    if (!item || !item->x || !item->dy || !item->dw) return EPI_ERR_INVALID_ARGUMENT;
    GemmTnArgs a = {};
    const int rc = group_item_args(*item, &a);
    if (rc != EPI_OK) return rc;
    return launch_tn(a, item->dw, item->dw_dtype == EPI_BF16, (float*)workspace, workspace_bytes, (hipStream_t)stream, pending);
}

extern "C" long long epi_slab_reduce_chunks(long long n) { return n > 0 ? (n / 2 + 255) / 256 : 0; }

extern "C" int epi_slab_reduce_multi(const EpiSlabReduce* rows_dev, int nrows, long long total_chunks, epi_stream_t stream) {
    if (!rows_dev || nrows <= 0 || total_chunks <= 0 || total_chunks > 0x7fffffffLL) return EPI_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(epi::slab_reduce_multi_kernel, dim3((unsigned)total_chunks), dim3(256), 0, (hipStream_t)stream, rows_dev, nrows);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Backbone convolutions (lib/models/pose3d_resnet.py:21-88: BasicBlock / Bottleneck conv1..conv3 and the downsample
// projections) on the same implicit-GEMM kernel: NHWC bf16 activations, channels_last weights.
//   forward        y[n][oh][ow][co] = sum_{kh,kw,ci} x[n][oh*s + kh - pad][ow*s + kw - pad][ci] * w[co][kh][kw][ci]
//                  = gather GEMM, M = B*Ho*Wo, N = Cout, K = KH*KW*Cin, Bt = the weight as it lies in memory
//   backward-data  stride 1: gather GEMM over dy with taps (pad - kh, pad - kw) and the transposed weight [Cin][KH][KW][Cout];
//                  stride 2: four output-parity phases (GemmPhases), each with the taps whose stride lands on that parity
//   backward-weight epi_conv2d_bwd_weight above (TN kernel).
// The transposed / phase-ordered weight is produced by epi_conv2d_pack_weight_bwd once per optimizer step.
// ---------------------------------------------------------------------------------------------------------------
namespace {

struct ConvBwdLayout {      // where tap (kh, kw) of the ORIGINAL weight goes in the packed backward-data weight
    int nphase;             // 1 (stride 1) or 4 (stride 2)
    int ntap[4];
    int dy[4][4], dx[4][4];
    long long bt_off[4];    // element offset of phase block [Cin][ntap * Cout]
    int tap_phase[16], tap_slot[16];   // per original tap: phase and position inside the phase (-1: tap unused)
};

// returns false when the geometry is not covered (more than 16 taps, more than 4 taps in a stride-2 phase, stride > 2)
bool conv_bwd_layout(int KH, int KW, int stride, int pad, int Cin, int Cout, ConvBwdLayout* L) {
    if (KH * KW > 16 || stride < 1 || stride > 2) return false;
    *L = ConvBwdLayout();
    if (stride == 1) {
        L->nphase = 1;
        for (int t = 0; t < KH * KW; ++t) { L->tap_phase[t] = 0; L->tap_slot[t] = t; }
        L->ntap[0] = KH * KW;
        return true;
    }
    L->nphase = 4;
    long long off = 0;
    for (int phase = 0; phase < 4; ++phase) {
        const int py = phase >> 1, px = phase & 1;
        int n = 0;
        for (int kh = 0; kh < KH; ++kh) {
            if ((py + pad - kh) & 1) continue;
            for (int kw = 0; kw < KW; ++kw) {
                if ((px + pad - kw) & 1) continue;
                if (n == 4) return false;
                L->dy[phase][n] = (py + pad - kh) / 2;          // exact: the numerator is even
                L->dx[phase][n] = (px + pad - kw) / 2;
                L->tap_phase[kh * KW + kw] = phase;
                L->tap_slot[kh * KW + kw] = n;
                ++n;
            }
        }
        L->ntap[phase] = n;
        L->bt_off[phase] = off;
        off += (long long)Cin * n * Cout;
    }
    return true;
}

struct PackArgs {
    const unsigned short* w;     // [Cout][ntap][Cin]
    unsigned short* out;
    int Cout, Cin, ntap;
    int tiles_ci, tiles_co;      // 64 x 64 tiles along Cin / Cout
    int reserved;
    long long tile_begin;        // first global tile of this row      (multi-layer table only)
    long long dst_base[16];      // per original tap: element offset of (ci = 0, co = 0) in the packed buffer
    int dst_ci_stride[16];       // per original tap: elements between consecutive ci
};

}  // namespace

namespace epi {
// 32 x 32 (co x ci) tile transpose through LDS: coalesced reads along ci, coalesced writes along co
__device__ __forceinline__ void conv_pack_tile(const PackArgs& p, int tap, int ci0, int co0);

__global__ __launch_bounds__(256) void conv_pack_weight_bwd_kernel(PackArgs p) {
    conv_pack_tile(p, blockIdx.z, blockIdx.x * 64, blockIdx.y * 64);
}

// every layer of a table in one launch: blockIdx.x = global tile; the owning row is found by bisection on tile_begin
__global__ __launch_bounds__(256) void conv_pack_weight_bwd_multi_kernel(const PackArgs* __restrict__ rows, int nrows) {
    const long long t = blockIdx.x;
    int lo = 0, hi = nrows - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (rows[mid].tile_begin <= t) lo = mid; else hi = mid - 1;
    }
    const PackArgs& p = rows[lo];
    int local = (int)(t - p.tile_begin);
    const int tci = local % p.tiles_ci;
    local /= p.tiles_ci;
    const int tco = local % p.tiles_co, tap = local / p.tiles_co;
    conv_pack_tile(p, tap, tci * 64, tco * 64);
}

// One 64 (co) x 64 (ci) tile of tap `tap`: 16-byte loads along ci, transposed through LDS, 16-byte stores along co.
// (round 4: the 32 x 32 tile with 2-byte accesses moved the 94 MB of a ResNet-50's weights in 75 us -- 1.25 TB/s, on the main stream behind Adam)
__device__ __forceinline__ void conv_pack_tile(const PackArgs& p, int tap, int ci0, int co0) {
    constexpr int PITCH = 66;                                   // elements: 33 words -> the transposed reads of 8 lanes fall 8 banks apart
    __shared__ unsigned short tile[64 * PITCH];
    const int chunk = threadIdx.x & 7, r0 = threadIdx.x >> 3;   // 8 chunks of 8 elements per 64-element row; 32 rows per round
    const bool vec = (p.Cin % 8 == 0) && (p.Cout % 8 == 0);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int row = r0 + 32 * r, co = co0 + row, ci = ci0 + chunk * 8;
        unsigned int v[4] = {0u, 0u, 0u, 0u};
        if (co < p.Cout && ci < p.Cin) {
            const unsigned short* src = p.w + ((long long)co * p.ntap + tap) * p.Cin + ci;
            if (vec) {
                const uint4v q = *reinterpret_cast<const uint4v*>(src);
                v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (ci + e < p.Cin) v[e >> 1] |= (unsigned int)src[e] << (16 * (e & 1));
            }
        }
        unsigned int* dst = reinterpret_cast<unsigned int*>(tile + row * PITCH + chunk * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[e] = v[e];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int row = r0 + 32 * r, ci = ci0 + row, co = co0 + chunk * 8;
        if (ci >= p.Cin || co >= p.Cout) continue;
        unsigned short e8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) e8[e] = tile[(chunk * 8 + e) * PITCH + row];
        unsigned short* out = p.out + p.dst_base[tap] + (long long)ci * p.dst_ci_stride[tap] + co;
        if (vec) {
            uint4v q;
            q.x = e8[0] | ((unsigned int)e8[1] << 16); q.y = e8[2] | ((unsigned int)e8[3] << 16);
            q.z = e8[4] | ((unsigned int)e8[5] << 16); q.w = e8[6] | ((unsigned int)e8[7] << 16);
            *reinterpret_cast<uint4v*>(out) = q;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (co + e < p.Cout) out[e] = e8[e];
        }
    }
}
}  // namespace epi

static int conv_pack_args(PackArgs* a, const void* w, void* w_bwd, int Cout, int Cin, int KH, int KW, int stride, int pad) {
    if (!w || !w_bwd || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0) return EPI_ERR_INVALID_ARGUMENT;
    ConvBwdLayout L;
    if (!conv_bwd_layout(KH, KW, stride, pad, Cin, Cout, &L)) return EPI_ERR_UNSUPPORTED;
    *a = PackArgs();
    a->w = (const unsigned short*)w; a->out = (unsigned short*)w_bwd; a->Cout = Cout; a->Cin = Cin; a->ntap = KH * KW;
    a->tiles_ci = (Cin + 63) / 64; a->tiles_co = (Cout + 63) / 64;
    for (int t = 0; t < KH * KW; ++t) {
        const int ph = L.tap_phase[t];
        a->dst_base[t] = L.bt_off[ph] + (long long)L.tap_slot[t] * Cout;
        a->dst_ci_stride[t] = L.ntap[ph] * Cout;
    }
    return EPI_OK;
}

// ConvTranspose2d(k4 s2 p1) weight in channels_last memory [Cin][kh][kw][Cout] (which IS the backward-data operand [Cin][16*Cout])
// -> the forward operand w_phase [4][Cout][4*Cin]: per tap a (Cin x Cout) -> (Cout x Cin) transpose, i.e. the same tile kernel with
// the roles of the two channel counts exchanged.  Tap (kh, kw) belongs to output parity ph = kh even, position ty (see
// epi_deconv4x4s2_pack_weight).
static int deconv_phase_pack_args(PackArgs* a, const void* w_cl, void* w_phase, int Cin, int Cout) {
    if (!w_cl || !w_phase || Cin <= 0 || Cout <= 0) return EPI_ERR_INVALID_ARGUMENT;
    *a = PackArgs();
    a->w = (const unsigned short*)w_cl; a->out = (unsigned short*)w_phase; a->Cout = Cin; a->Cin = Cout; a->ntap = 16;
    a->tiles_ci = (Cout + 63) / 64; a->tiles_co = (Cin + 63) / 64;
    for (int kh = 0; kh < 4; ++kh)
        for (int kw = 0; kw < 4; ++kw) {
            const int ph = (kh & 1) ? 0 : 1, ty = ph ? (kh == 0 ? 0 : 1) : (kh == 1 ? 0 : 1);
            const int pw = (kw & 1) ? 0 : 1, tx = pw ? (kw == 0 ? 0 : 1) : (kw == 1 ? 0 : 1);
            a->dst_base[kh * 4 + kw] = ((long long)(2 * ph + pw) * Cout * 4 + (2 * ty + tx)) * Cin;
            a->dst_ci_stride[kh * 4 + kw] = 4 * Cin;
        }
    return EPI_OK;
}

extern "C" int epi_deconv4x4s2_pack_phase_cl(const void* w_cl, int Cin, int Cout, void* w_phase, epi_stream_t stream) {
    PackArgs a;
    const int rc = deconv_phase_pack_args(&a, w_cl, w_phase, Cin, Cout);
    if (rc != EPI_OK) return rc;
    hipLaunchKernelGGL(epi::conv_pack_weight_bwd_kernel, dim3((unsigned)a.tiles_ci, (unsigned)a.tiles_co, 16u), dim3(256), 0, (hipStream_t)stream, a);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}

extern "C" int epi_deconv4x4s2_pack_fill_row(void* row_host, const void* w_cl, void* w_phase, int Cin, int Cout, long long tile_begin,
                                             long long* ntiles) {
    if (!row_host || !ntiles) return EPI_ERR_INVALID_ARGUMENT;
    PackArgs a;
    const int rc = deconv_phase_pack_args(&a, w_cl, w_phase, Cin, Cout);
    if (rc != EPI_OK) return rc;
    a.tile_begin = tile_begin;
    *reinterpret_cast<PackArgs*>(row_host) = a;
    *ntiles = (long long)a.tiles_ci * a.tiles_co * a.ntap;
    return EPI_OK;
}

extern "C" size_t epi_conv2d_pack_row_bytes(void) { return sizeof(PackArgs); }

extern "C" int epi_conv2d_pack_fill_row(void* row_host, const void* w, void* w_bwd, int Cout, int Cin, int KH, int KW, int stride,
                                        int pad, long long tile_begin, long long* ntiles) {
    if (!row_host || !ntiles) return EPI_ERR_INVALID_ARGUMENT;
    PackArgs a;
    const int rc = conv_pack_args(&a, w, w_bwd, Cout, Cin, KH, KW, stride, pad);
    if (rc != EPI_OK) return rc;
    a.tile_begin = tile_begin;
    *reinterpret_cast<PackArgs*>(row_host) = a;
    *ntiles = (long long)a.tiles_ci * a.tiles_co * a.ntap;
    return EPI_OK;
}

extern "C" int epi_conv2d_pack_weight_bwd_multi(const void* rows, int nrows, long long total_tiles, epi_stream_t stream) {
    if (!rows || nrows <= 0 || total_tiles <= 0 || total_tiles > 0x7fffffffLL) return EPI_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(epi::conv_pack_weight_bwd_multi_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream,
                       (const PackArgs*)rows, nrows);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}

extern "C" int epi_conv2d_pack_weight_bwd(const void* w, int Cout, int Cin, int KH, int KW, int stride, int pad, void* w_bwd,
                                          epi_stream_t stream) {
    PackArgs a;
    const int rc = conv_pack_args(&a, w, w_bwd, Cout, Cin, KH, KW, stride, pad);
    if (rc != EPI_OK) return rc;
    hipLaunchKernelGGL(epi::conv_pack_weight_bwd_kernel, dim3((unsigned)a.tiles_ci, (unsigned)a.tiles_co, (unsigned)(KH * KW)),
                       dim3(256), 0, (hipStream_t)stream, a);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}

static inline int conv_out_dim(int H, int K, int stride, int pad) { return (H + 2 * pad - K) / stride + 1; }

extern "C" size_t epi_conv2d_workspace_bytes(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0) return 0;
    const int Ho = conv_out_dim(H, KH, stride, pad), Wo = conv_out_dim(W, KW, stride, pad);
    if (Ho <= 0 || Wo <= 0) return 0;
    size_t need = epi_gemm_workspace_bytes(B * Ho * Wo, Cout, KH * KW * Cin, 1);                       // forward
    if (KH == 3 && KW == 3 && stride == 1 && pad == 1 && Cin % GBK == 0 && Cout % GBK == 0) {          // patch kernel: split over channel chunks
        const PatchPlan pf = patch_plan(B * H * W, Cout, Cin, W, (size_t)-1), pb = patch_plan(B * H * W, Cin, Cout, W, (size_t)-1);
        if (pf.ok && pf.nsplit > 1) need = std::max(need, (size_t)pf.nsplit * B * H * W * Cout * sizeof(float));
        if (pb.ok && pb.nsplit > 1) need = std::max(need, (size_t)pb.nsplit * B * H * W * Cin * sizeof(float));
    }
    if (stride == 1) need = std::max(need, epi_gemm_workspace_bytes(B * H * W, Cin, KH * KW * Cout, 1));   // backward-data
    else need = std::max(need, epi_gemm_workspace_bytes(B * ((H + 1) / 2) * ((W + 1) / 2), Cin, 4 * Cout, 4));
    return need;
}

static int conv2d_fwd_impl(const void* x, const void* w, void* y, int B, int H, int W, int Cin, int Cout, int KH, int KW,
                          int stride, int pad, float* bn_sums, int* bn_sums_done, void* workspace, size_t workspace_bytes,
                          epi_stream_t stream, bool out_f32) {
    if (bn_sums_done) *bn_sums_done = 0;
    if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0)
        return EPI_ERR_INVALID_ARGUMENT;
    const int Ho = conv_out_dim(H, KH, stride, pad), Wo = conv_out_dim(W, KW, stride, pad);
    if (Ho <= 0 || Wo <= 0) return EPI_ERR_INVALID_ARGUMENT;
    if (KH * KW > 16 || Cout % 4) return EPI_ERR_UNSUPPORTED;
    GemmArgs a = {};
    a.A = (const unsigned short*)x; a.Bt = (const unsigned short*)w; a.C = y;
    a.M = B * Ho * Wo; a.N = Cout; a.K = KH * KW * Cin; a.ldb = KH * KW * Cin; a.ldc = Cout;
    if (KH == 1 && KW == 1 && stride == 1 && pad == 0) {
        a.lda = Cin;                                     // plain GEMM on the [B*H*W][Cin] view
    } else {
        if (Cin % GBK) return EPI_ERR_UNSUPPORTED;
        a.ga.enabled = 1; a.ga.Hg = Ho; a.ga.Wg = Wo; a.ga.Hs = H; a.ga.Ws = W; a.ga.Cs = Cin; a.ga.stride = stride;
        for (int kh = 0; kh < KH; ++kh)
            for (int kw = 0; kw < KW; ++kw) { a.ga.dy[kh * KW + kw] = kh - pad; a.ga.dx[kh * KW + kw] = kw - pad; }
        if ((long long)B * H * W * Cin >= (1LL << 31)) return EPI_ERR_UNSUPPORTED;      // 32-bit element offsets in the gather
    }
    a.stats = out_f32 ? nullptr : bn_sums;
    a.stats_copies = epi_bn_sum_copies(Cout);
    return launch_gemm(a, out_f32, 1, workspace, workspace_bytes, (hipStream_t)stream, bn_sums_done);
}
extern "C" int epi_conv2d_fwd(const void* x, const void* w, void* y, int B, int H, int W, int Cin, int Cout, int KH, int KW,
                              int stride, int pad, float* bn_sums, int* bn_sums_done, void* workspace, size_t workspace_bytes,
                              epi_stream_t stream) {
    return conv2d_fwd_impl(x, w, y, B, H, W, Cin, Cout, KH, KW, stride, pad, bn_sums, bn_sums_done, workspace, workspace_bytes, stream, false);
}
// 1x1 / stride-1 convolution whose INPUT is a raw convolution output still waiting for its BatchNorm (+ ReLU): y = conv(relu(bn(z))), the
// normalised tensor is never written (GemmArgs::bn_in).  in_bn: the layer in front (its batch sums already in in_bn->sums_ws when in_training = 2;
// in_training = 0: running statistics); the launch fills in_bn->mean / rstd / scale_shift, updates its running estimates and clears in_bn->bwd_sums
// exactly as epi_bn_act_fwd(training = 2) would.  Anything it cannot do: EPI_ERR_UNSUPPORTED (the caller normalises first, as before).
extern "C" int epi_conv1x1_fwd_bn_in(const void* z, const EpiBnLayer* in_bn, int in_training, float eps, float momentum, const void* w, void* y, int B, int H,
                                     int W, int Cin, int Cout, float* bn_sums, int* bn_sums_done, void* workspace, size_t workspace_bytes,
                                     epi_stream_t stream) {
    if (bn_sums_done) *bn_sums_done = 0;
    if (!z || !in_bn || !w || !y || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (in_training != 0 && in_training != 2)) return EPI_ERR_INVALID_ARGUMENT;
    if (!in_bn->gamma || !in_bn->beta || !in_bn->scale_shift || (in_training && (!in_bn->sums_ws || !in_bn->mean || !in_bn->rstd)) ||
        (!in_training && (!in_bn->running_mean || !in_bn->running_var)))
        return EPI_ERR_INVALID_ARGUMENT;
    if (Cin % GBK || Cin > 2048 || Cout % 8 || deterministic()) return EPI_ERR_UNSUPPORTED;
    GemmArgs a = {};
    a.A = (const unsigned short*)z; a.Bt = (const unsigned short*)w; a.C = y;
    a.M = B * H * W; a.N = Cout; a.K = Cin; a.lda = Cin; a.ldb = Cin; a.ldc = Cout;
    a.stats = bn_sums;
    a.stats_copies = epi_bn_sum_copies(Cout);
    a.bn_in_on = 1;
    BnAffine& n = a.bn_in;
    const long long R = (long long)B * H * W;
    n.sums = in_training ? in_bn->sums_ws : nullptr; n.ncopies = epi_bn_sum_copies(Cin); n.R = R; n.inv_r = 1.0 / (double)R;
    n.gamma = in_bn->gamma; n.beta = in_bn->beta; n.eps = eps; n.momentum = momentum;
    n.running_mean = in_bn->running_mean; n.running_var = in_bn->running_var; n.num_batches = in_bn->num_batches_tracked;
    n.mean = in_bn->mean; n.rstd = in_bn->rstd; n.scale = in_bn->scale_shift; n.shift = in_bn->scale_shift + Cin;
    n.bwd_sums = in_training ? in_bn->bwd_sums : nullptr;
    return launch_gemm(a, false, 1, workspace, workspace_bytes, (hipStream_t)stream, bn_sums_done);
}
// fp32 result (y float [B][Ho][Wo][Cout]; operands bf16): the fp32-grade verification mode feeds split operands (csrc/precise.hip)
extern "C" int epi_conv2d_fwd_f32(const void* x, const void* w, void* y, int B, int H, int W, int Cin, int Cout, int KH, int KW,
                                  int stride, int pad, void* workspace, size_t workspace_bytes, epi_stream_t stream) {
    return conv2d_fwd_impl(x, w, y, B, H, W, Cin, Cout, KH, KW, stride, pad, nullptr, nullptr, workspace, workspace_bytes, stream, true);
}

static int conv2d_bwd_data_impl(const void* dy, const void* w_bwd, void* dx, int B, int H, int W, int Cin, int Cout, int KH,
                               int KW, int stride, int pad, const void* addend, void* workspace, size_t workspace_bytes,
                               epi_stream_t stream, bool out_f32, const EpiBnReduce* red = nullptr, int* red_done = nullptr, int addend_step = 1) {
    if (red_done) *red_done = 0;
    if (addend_step != 1 && addend_step != 2) return EPI_ERR_INVALID_ARGUMENT;
    if (!dy || !w_bwd || !dx || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0)
        return EPI_ERR_INVALID_ARGUMENT;
    const int Ho = conv_out_dim(H, KH, stride, pad), Wo = conv_out_dim(W, KW, stride, pad);
    if (Ho <= 0 || Wo <= 0) return EPI_ERR_INVALID_ARGUMENT;
    ConvBwdLayout L;
    if (!conv_bwd_layout(KH, KW, stride, pad, Cin, Cout, &L) || Cin % 4) return EPI_ERR_UNSUPPORTED;
    GemmArgs a = {};
    a.A = (const unsigned short*)dy; a.Bt = (const unsigned short*)w_bwd; a.C = dx; a.N = Cin; a.ldc = Cin;
    a.addend = (const unsigned short*)addend;
    a.addend_step = addend ? addend_step : 1; a.add_h = H; a.add_w = W;
    a.br = bnred_args(red);
    if (stride == 1) {
        a.M = B * H * W; a.K = KH * KW * Cout; a.ldb = KH * KW * Cout;
        if (KH == 1 && KW == 1 && pad == 0) {
            a.lda = Cout;
        } else {
            if (Cout % GBK) return EPI_ERR_UNSUPPORTED;
            a.ga.enabled = 1; a.ga.Hg = H; a.ga.Wg = W; a.ga.Hs = Ho; a.ga.Ws = Wo; a.ga.Cs = Cout; a.ga.stride = 1;
            for (int kh = 0; kh < KH; ++kh)
                for (int kw = 0; kw < KW; ++kw) { a.ga.dy[kh * KW + kw] = pad - kh; a.ga.dx[kh * KW + kw] = pad - kw; }
        }
        return launch_gemm(a, out_f32, 1, workspace, workspace_bytes, (hipStream_t)stream, nullptr, red_done);
    }
    // stride 2: dx pixel (2i + py, 2j + px) gathers dy pixels (i + dy_t, j + dx_t) over the taps of its parity phase
    if ((H & 1) || (W & 1) || Cout % GBK) return EPI_ERR_UNSUPPORTED;
    const int Hh = H / 2, Wh = W / 2;
    int kmax = 0;
    for (int p = 0; p < 4; ++p) kmax = std::max(kmax, L.ntap[p] * Cout);
    a.M = B * Hh * Wh; a.K = kmax; a.ldb = kmax;
    a.ga.enabled = 1; a.ga.Hg = Hh; a.ga.Wg = Wh; a.ga.Hs = Ho; a.ga.Ws = Wo; a.ga.Cs = Cout; a.ga.stride = 1;
    a.sc.enabled = 1; a.sc.Hg = Hh; a.sc.Wg = Wh; a.sc.Ho = H; a.sc.Wo = W; a.sc.so = 2;
    a.ph.enabled = 1;
    for (int p = 0; p < 4; ++p) {
        a.ph.ntap[p] = L.ntap[p];
        a.ph.bt_off[p] = L.bt_off[p];
        for (int t = 0; t < 4; ++t) { a.ph.dy[p][t] = L.dy[p][t]; a.ph.dx[p][t] = L.dx[p][t]; }
    }
    return launch_gemm(a, out_f32, 4, workspace, workspace_bytes, (hipStream_t)stream, nullptr, red_done);
}
extern "C" int epi_conv2d_bwd_data(const void* dy, const void* w_bwd, void* dx, int B, int H, int W, int Cin, int Cout, int KH,
                                   int KW, int stride, int pad, const void* addend, void* workspace, size_t workspace_bytes,
                                   epi_stream_t stream) {
    return conv2d_bwd_data_impl(dy, w_bwd, dx, B, H, W, Cin, Cout, KH, KW, stride, pad, addend, workspace, workspace_bytes, stream, false);
}
// the same with the fused BatchNorm-backward reduction of the layer that produced this convolution's input (EpiBnReduce): dx becomes dz
extern "C" int epi_conv2d_bwd_data_bnred(const void* dy, const void* w_bwd, void* dx, int B, int H, int W, int Cin, int Cout, int KH,
                                         int KW, int stride, int pad, const void* addend, int addend_step, const EpiBnReduce* red, int* red_done,
                                         void* workspace, size_t workspace_bytes, epi_stream_t stream) {
    if (red && !red_done) return EPI_ERR_INVALID_ARGUMENT;
    return conv2d_bwd_data_impl(dy, w_bwd, dx, B, H, W, Cin, Cout, KH, KW, stride, pad, addend, workspace, workspace_bytes, stream, false, red, red_done,
                                addend_step);
}
// 1 when epi_conv2d_bwd_data_bnred takes a half-resolution addend (addend_step 2) for this 1x1 / stride-1 backward-data problem: an unsplit launch
// on 16-byte rows (the plan decides; a caller that gets 0 materialises the full-resolution addend as before)
extern "C" int epi_conv2d_bwd_data_half_addend_ok(int B, int H, int W, int Cin, int Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || ((H | W) & 1) || Cin % 8 || Cout % 8) return 0;
    const GemmPlan pl = gemm_plan(B * H * W, Cin, Cout, Cin, 1, false);
    return pl.nsplit == 1 ? 1 : 0;
}
extern "C" int epi_conv2d_bwd_data_f32(const void* dy, const void* w_bwd, void* dx, int B, int H, int W, int Cin, int Cout, int KH,
                                       int KW, int stride, int pad, void* workspace, size_t workspace_bytes, epi_stream_t stream) {
    return conv2d_bwd_data_impl(dy, w_bwd, dx, B, H, W, Cin, Cout, KH, KW, stride, pad, nullptr, workspace, workspace_bytes, stream, true);
}


// ---------------------------------------------------------------------------------------------------------------
// The 7x7 / stride-2 / pad-3 stem convolution on 3 input channels (pose3d_resnet.py:99,186) on the SAME implicit-GEMM kernels.
// Three channels cannot feed a 16-byte DMA chunk, so the image is first rewritten space-to-depth: with P = the image zero-padded
// by 3 (top / left), s2d[a][b][(dy*2 + dx)*3 + c] = P[2a + dy][2b + dx][c] (12 channels, padded to 16 = 32 bytes per pixel), and
//     y[oh][ow][co] = sum_{kh,kw,c} P[2oh + kh][2ow + kw][c] w[co][c][kh][kw]
//                   = sum_{a,b < 4} sum_{ch < 16} s2d[oh + a][ow + b][ch] w8[co][a][b][ch]           (kh = 2a + dy, kw = 2b + dx; w8 = 0 for kh, kw = 7)
// -- a 4 x 4 stride-1 convolution whose four horizontal taps of one row are 64 CONTIGUOUS elements starting at pixel (oh + a, ow):
// the gather GEMM with 4 vertical taps, 64 elements per tap and a pixel pitch of 16 (GemmGather::pitch), K = 256 (147 real products,
// 1.7x the MFMA work of a kernel that is bound by writing its 67 MB of output anyway).  The weight gradient is the TN kernel's
// column gather with the same geometry (ldb = the pixel pitch).  No input gradient (the image needs none).
// ---------------------------------------------------------------------------------------------------------------
namespace epi {

// x [B][3][H][W] (NCHW) or [B][H][W][3] (NHWC), f32 or bf16 -> s2d [B][H/2 + 3][W/2 + 3][16] bf16.  One thread = one s2d pixel.
template <typename T, bool NHWC>
__global__ __launch_bounds__(256) void stem_s2d_kernel(const T* __restrict__ x, int B, int H, int W, unsigned short* __restrict__ out) {
    const int Hs = H / 2 + 3, Ws = W / 2 + 3;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)B * Hs * Ws) return;
    const int b = (int)(t % Ws);
    const long long r = t / Ws;
    const int a = (int)(r % Hs), n = (int)(r / Hs);
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = 0.f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int ih = 2 * a + dy - 3, iw = 2 * b + dx - 3;
            if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const long long src = NHWC ? (((long long)n * H + ih) * W + iw) * 3 + c : (((long long)n * 3 + c) * H + ih) * W + iw;
                    v[(dy * 2 + dx) * 3 + c] = Elem<T>::load1(x + src);
                }
            }
        }
    uint4v o0, o1;
    o0.x = pack_bf16x2(v[0], v[1]); o0.y = pack_bf16x2(v[2], v[3]); o0.z = pack_bf16x2(v[4], v[5]); o0.w = pack_bf16x2(v[6], v[7]);
    o1.x = pack_bf16x2(v[8], v[9]); o1.y = pack_bf16x2(v[10], v[11]); o1.z = 0u; o1.w = 0u;
    uint4v* q = reinterpret_cast<uint4v*>(out + t * 16);
    q[0] = o0;
    q[1] = o1;
}

// w [Cout][3][7][7] (contiguous) or channels_last memory [Cout][7][7][3], bf16 -> wp [Cout][4][4][16] bf16 (zero where kh or kw = 7, ch >= 12)
__global__ void stem_pack_weight_kernel(const unsigned short* __restrict__ w, int Cout, int channels_last, unsigned short* __restrict__ wp) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Cout * 256) return;
    const int ch = t & 15, b = (t >> 4) & 3, a = (t >> 6) & 3, co = t >> 8;
    unsigned short v = 0;
    if (ch < 12) {
        const int c = ch % 3, dx = (ch / 3) & 1, dy = ch / 6, kh = 2 * a + dy, kw = 2 * b + dx;
        if (kh < 7 && kw < 7) v = channels_last ? w[((co * 7 + kh) * 7 + kw) * 3 + c] : w[((co * 3 + c) * 7 + kh) * 7 + kw];
    }
    wp[t] = v;
}

// dwp [Cout][4][4][16] f32 -> dw [Cout][3][7][7] (contiguous or channels_last memory), f32 or bf16
__global__ void stem_unpack_weight_grad_kernel(const float* __restrict__ dwp, int Cout, int channels_last, int out_bf16, void* __restrict__ dw) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Cout * 147) return;
    int co, c, kh, kw;
    if (channels_last) { c = t % 3; kw = (t / 3) % 7; kh = (t / 21) % 7; co = t / 147; }
    else { kw = t % 7; kh = (t / 7) % 7; c = (t / 49) % 3; co = t / 147; }
    const float v = dwp[co * 256 + (kh >> 1) * 64 + (kw >> 1) * 16 + ((kh & 1) * 2 + (kw & 1)) * 3 + c];
    if (out_bf16) reinterpret_cast<unsigned short*>(dw)[t] = f32_to_bf16(v);
    else reinterpret_cast<float*>(dw)[t] = v;
}

}  // namespace epi

extern "C" size_t epi_stem7x7s2_s2d_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return 0;
    return (size_t)B * (H / 2 + 3) * (W / 2 + 3) * 16 * sizeof(unsigned short);
}

extern "C" int epi_stem7x7s2_s2d(const void* x, int x_dtype, int x_layout, int B, int H, int W, void* s2d, epi_stream_t stream) {
    if (!x || !s2d || B <= 0 || H <= 0 || W <= 0) return EPI_ERR_INVALID_ARGUMENT;
    if ((H & 1) || (W & 1) || (x_dtype != EPI_F32 && x_dtype != EPI_BF16) || (x_layout != EPI_NCHW && x_layout != EPI_NHWC)) return EPI_ERR_UNSUPPORTED;
    if ((long long)B * (H / 2 + 3) * (W / 2 + 3) * 16 >= (1LL << 31)) return EPI_ERR_UNSUPPORTED;      // 32-bit element offsets in the gather
    const long long total = (long long)B * (H / 2 + 3) * (W / 2 + 3);
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (x_dtype == EPI_F32) {
        if (x_layout == EPI_NHWC) hipLaunchKernelGGL((epi::stem_s2d_kernel<float, true>), grid, block, 0, st, (const float*)x, B, H, W, (unsigned short*)s2d);
        else hipLaunchKernelGGL((epi::stem_s2d_kernel<float, false>), grid, block, 0, st, (const float*)x, B, H, W, (unsigned short*)s2d);
    } else {
        if (x_layout == EPI_NHWC) hipLaunchKernelGGL((epi::stem_s2d_kernel<unsigned short, true>), grid, block, 0, st, (const unsigned short*)x, B, H, W, (unsigned short*)s2d);
        else hipLaunchKernelGGL((epi::stem_s2d_kernel<unsigned short, false>), grid, block, 0, st, (const unsigned short*)x, B, H, W, (unsigned short*)s2d);
    }
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}

extern "C" int epi_stem7x7s2_pack_weight(const void* w_bf16, int channels_last, int Cout, void* wp, epi_stream_t stream) {
    if (!w_bf16 || !wp || Cout <= 0) return EPI_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(epi::stem_pack_weight_kernel, dim3((unsigned)((Cout * 256 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)w_bf16, Cout, channels_last ? 1 : 0, (unsigned short*)wp);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}

extern "C" int epi_stem7x7s2_unpack_weight_grad(const float* dwp, int Cout, int channels_last, void* dw, int dw_dtype, epi_stream_t stream) {
    if (!dwp || !dw || Cout <= 0) return EPI_ERR_INVALID_ARGUMENT;
    if (dw_dtype != EPI_F32 && dw_dtype != EPI_BF16) return EPI_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(epi::stem_unpack_weight_grad_kernel, dim3((unsigned)((Cout * 147 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dwp, Cout,
                       channels_last ? 1 : 0, dw_dtype == EPI_BF16 ? 1 : 0, dw);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}

extern "C" size_t epi_stem7x7s2_workspace_bytes(int B, int H, int W, int Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 0;
    return std::max(epi_gemm_workspace_bytes(B * (H / 2) * (W / 2), Cout, 256, 1), epi_gemm_tn_workspace_bytes(B * (H / 2) * (W / 2), Cout, 256, 1));
}

// y [B][H/2][W/2][Cout] bf16 (raw, pre-BatchNorm) from s2d (epi_stem7x7s2_s2d) and wp (epi_stem7x7s2_pack_weight); bn_sums / bn_sums_done as
// epi_conv2d_fwd.  Cout % 8 == 0.
extern "C" int epi_stem7x7s2_fwd(const void* s2d, const void* wp, void* y, int B, int H, int W, int Cout, float* bn_sums, int* bn_sums_done,
                                 void* workspace, size_t workspace_bytes, epi_stream_t stream) {
    if (bn_sums_done) *bn_sums_done = 0;
    if (!s2d || !wp || !y || B <= 0 || H <= 0 || W <= 0 || Cout <= 0) return EPI_ERR_INVALID_ARGUMENT;
    if ((H & 1) || (W & 1) || Cout % 8) return EPI_ERR_UNSUPPORTED;
    GemmArgs a = {};
    a.A = (const unsigned short*)s2d; a.Bt = (const unsigned short*)wp; a.C = y;
    a.M = B * (H / 2) * (W / 2); a.N = Cout; a.K = 256; a.ldb = 256; a.ldc = Cout;
    a.ga.enabled = 1; a.ga.Hg = H / 2; a.ga.Wg = W / 2; a.ga.Hs = H / 2 + 3; a.ga.Ws = W / 2 + 3; a.ga.Cs = 64; a.ga.pitch = 16; a.ga.stride = 1;
    for (int t = 0; t < 4; ++t) { a.ga.dy[t] = t; a.ga.dx[t] = 0; }
    a.stats = bn_sums;
    a.stats_copies = epi_bn_sum_copies(Cout);
    return launch_gemm(a, false, 1, workspace, workspace_bytes, (hipStream_t)stream, bn_sums_done);
}

// dwp [Cout][4][4][16] f32 = the packed weight gradient (epi_stem7x7s2_unpack_weight_grad turns it into the parameter's layout)
extern "C" int epi_stem7x7s2_bwd_weight(const void* s2d, const void* dy, float* dwp, int B, int H, int W, int Cout, void* workspace,
                                        size_t workspace_bytes, epi_stream_t stream) {
    if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0) return EPI_ERR_INVALID_ARGUMENT;
    if ((H & 1) || (W & 1)) return EPI_ERR_UNSUPPORTED;
    GemmTnArgs a = {};
    a.A = (const unsigned short*)dy; a.B = (const unsigned short*)s2d; a.R = B * (H / 2) * (W / 2); a.I = Cout; a.J = 256; a.lda = Cout; a.ldb = 16;
    a.gather = 1; a.Hg = H / 2; a.Wg = W / 2; a.Hs = H / 2 + 3; a.Ws = W / 2 + 3; a.Cs = 64; a.stride = 1; a.pad = 0; a.KW = 1;
    return launch_tn(a, dwp, 0, (float*)workspace, workspace_bytes, (hipStream_t)stream);
}
