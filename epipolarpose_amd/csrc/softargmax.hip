// Integral regression (soft-argmax) forward / backward and row arg-max for gfx950.
//
// Replaces lib/core/integral_loss.py:49-86 of the reference (softmax over D*H*W + three marginal
// reductions + dot with arange = >= 5 full passes over the logits) with ONE streaming pass:
// every workgroup owns a 16 KiB-vector chunk of one (batch, joint) row, keeps an online
// (max, sum e, sum e*x, sum e*y, sum e*z) in registers, reduces it with wave64 shuffles + LDS, and
// writes one 32-byte partial; a tiny second kernel merges the partials of a row.
// HBM-bound: algorithmic traffic = 1 read of the logits (fwd), 1 read + 1 write (bwd).
#include "common.h"

namespace epi {

constexpr int SA_THREADS = 256;
constexpr int SA_ITERS = 16;       // vectors per thread per chunk
constexpr int SA_UNROLL = 4;       // independent 16-byte loads in flight per thread

struct alignas(32) SoftPartial { float m, s, a0, a1, a2, pad0, pad1, pad2; };

// online-softmax merge of two partial states
__device__ __forceinline__ void merge(float& m, float& s, float& a0, float& a1, float& a2,
                                      float m2, float s2, float b0, float b1, float b2) {
    const float M = fmaxf(m, m2);
    const float f1 = fast_exp2((m - M) * EPI_LOG2E);
    const float f2 = fast_exp2((m2 - M) * EPI_LOG2E);
    s = s * f1 + s2 * f2;
    a0 = a0 * f1 + b0 * f2;
    a1 = a1 * f1 + b1 * f2;
    a2 = a2 * f1 + b2 * f2;
    m = M;
}

template <typename T, int VEC> struct VecIO {
    __device__ static __forceinline__ void load(const T* p, float (&v)[VEC]) { Elem<T>::load(p, v); }
    __device__ static __forceinline__ void store(T* p, const float (&v)[VEC]) { Elem<T>::store(p, v); }
};
template <typename T> struct VecIO<T, 1> {
    __device__ static __forceinline__ void load(const T* p, float (&v)[1]) { v[0] = Elem<T>::load1(p); }
    __device__ static __forceinline__ void store(T* p, const float (&v)[1]) { Elem<T>::store1(p, v[0]); }
};

// Row geometry shared by forward and backward.  A row (b, j) is viewed as an abstract [E2][E1][E0]
// array whose fastest axis E0 is the one that is contiguous in memory:
//   NCHW: (E0,E1,E2) = (W,H,D), memory offset = row*N + e
//   NHWC: (E0,E1,E2) = (D,W,H), memory offset = (b*H*W + e/E0)*C + j*D + e%E0
struct RowGeom {
    int E0, E1, E2;        // extents
    int N;                 // E0*E1*E2
    int C;                 // channels J*D (NHWC only)
    int J;
};

// Per-thread cursor: coordinates of the first element of the current vector, kept as floats
// (exact small integers) and advanced by a fixed stride with carries -- no divisions in the loop.
template <bool NHWC> struct Cursor {
    float c0, c1, c2;      // coordinates along E0, E1, E2
    float s0, s1, s2;      // stride decomposition
    float e0, e1;          // extents as float
    long long mem;         // element offset of the vector in memory
    long long mem_step;    // NCHW: stride ; NHWC: (stride / E0) * C + (stride % E0)
    long long mem_carry;   // NHWC: C - E0 (extra when c0 wraps) ; NCHW: 0
    __device__ __forceinline__ void init(const RowGeom& g, long long row_base, int e, int stride) {
        c0 = (float)(e % g.E0);
        const int t = e / g.E0;
        c1 = (float)(t % g.E1);
        c2 = (float)(t / g.E1);
        s0 = (float)(stride % g.E0);
        const int ts = stride / g.E0;
        s1 = (float)(ts % g.E1);
        s2 = (float)(ts / g.E1);
        e0 = (float)g.E0;
        e1 = (float)g.E1;
        if (NHWC) {
            mem = row_base + (long long)t * g.C + (e % g.E0);
            mem_step = (long long)ts * g.C + (stride % g.E0);
            mem_carry = (long long)g.C - g.E0;
        } else {
            mem = row_base + e;
            mem_step = stride;
            mem_carry = 0;
        }
    }
    __device__ __forceinline__ void advance() {
        mem += mem_step;
        c0 += s0;
        if (c0 >= e0) { c0 -= e0; c1 += 1.f; if (NHWC) mem += mem_carry; }
        c1 += s1;
        if (c1 >= e1) { c1 -= e1; c2 += 1.f; }
        c2 += s2;
    }
};

template <bool NHWC>
__device__ __forceinline__ long long row_base_of(const RowGeom& g, int row) {
    if (NHWC) {
        const int b = row / g.J, j = row - b * g.J;
        return (long long)b * g.N * g.J + (long long)j * g.E0;   // b*H*W*C + j*D   (N*J == H*W*C)
    }
    return (long long)row * g.N;
}

// raw (unconverted) 16-byte vector held in registers while the next block's loads are in flight
template <typename T, int VEC> struct RawVec {
    uint4v r;
    __device__ __forceinline__ void load(const T* p) { r = *reinterpret_cast<const uint4v*>(p); }
    __device__ __forceinline__ void fill_low() { r.x = r.y = r.z = r.w = (sizeof(T) == 4) ? 0xff7fffffu : 0xff7fff7fu; }   // -FLT_MAX / -bf16 max
    __device__ __forceinline__ void unpack(float (&v)[VEC]) const {
        if (sizeof(T) == 4) {
            v[0] = __uint_as_float(r.x); v[1] = __uint_as_float(r.y); v[2] = __uint_as_float(r.z); v[VEC - 1] = __uint_as_float(r.w);
        } else {
            const unsigned int w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) { v[(2 * i) % VEC] = __uint_as_float(w[i] << 16); v[(2 * i + 1) % VEC] = __uint_as_float(w[i] & 0xffff0000u); }
        }
    }
};
template <typename T> struct RawVec<T, 1> {
    float r;
    __device__ __forceinline__ void load(const T* p) { r = Elem<T>::load1(p); }
    __device__ __forceinline__ void fill_low() { r = -FLT_MAX; }
    __device__ __forceinline__ void unpack(float (&v)[1]) const { v[0] = r; }
};

template <typename T, int VEC, bool NHWC, int ITERS>
__global__ __launch_bounds__(SA_THREADS) void softargmax_partial_kernel(const T* __restrict__ logits, RowGeom g,
                                                                        int nchunk, SoftPartial* __restrict__ part) {
    const int row = blockIdx.x / nchunk;
    const int chunk = blockIdx.x - row * nchunk;
    const int tid = threadIdx.x;
    constexpr int STRIDE = SA_THREADS * VEC;
    constexpr int NBLK = ITERS / SA_UNROLL;
    const int e_begin = chunk * (STRIDE * ITERS) + tid * VEC;

    Cursor<NHWC> cur;
    cur.init(g, row_base_of<NHWC>(g, row), e_begin, STRIDE);

    // running state: mx = exact running max; exp(x - mx) = exp2(fma(x, log2e, -mx*log2e)): one FMA + v_exp_f32
    float mx = -FLT_MAX, s = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f;
    int e = e_begin;
    RawVec<T, VEC> cur_raw[SA_UNROLL], nxt_raw[SA_UNROLL];
    float k0[SA_UNROLL], k1[SA_UNROLL], k2[SA_UNROLL], n0[SA_UNROLL], n1[SA_UNROLL], n2[SA_UNROLL];
    bool ok[SA_UNROLL], nok[SA_UNROLL];
    auto issue = [&](RawVec<T, VEC> (&dst)[SA_UNROLL], float (&c0)[SA_UNROLL], float (&c1)[SA_UNROLL], float (&c2)[SA_UNROLL],
                     bool (&valid)[SA_UNROLL]) {
#pragma unroll
        for (int u = 0; u < SA_UNROLL; ++u) {
            valid[u] = e < g.N;
            if (valid[u]) dst[u].load(logits + cur.mem); else dst[u].fill_low();
            c0[u] = cur.c0; c1[u] = cur.c1; c2[u] = cur.c2;
            cur.advance();
            e += STRIDE;
        }
    };
    issue(cur_raw, k0, k1, k2, ok);
#pragma unroll 1
    for (int blk = 0; blk < NBLK; ++blk) {
        if (blk + 1 < NBLK) issue(nxt_raw, n0, n1, n2, nok);        // next block's loads fly during this block's math
        float x[SA_UNROLL][VEC];
        float bm = -FLT_MAX;
#pragma unroll
        for (int u = 0; u < SA_UNROLL; ++u) {
            cur_raw[u].unpack(x[u]);
#pragma unroll
            for (int k = 0; k < VEC; ++k) bm = fmaxf(bm, x[u][k]);
        }
        const float mn = fmaxf(mx, bm);
        const float sc = fast_exp2((mx - mn) * EPI_LOG2E);
        s *= sc; a0 *= sc; a1 *= sc; a2 *= sc;
        mx = mn;
        const float ml = mx * EPI_LOG2E;
#pragma unroll
        for (int u = 0; u < SA_UNROLL; ++u) {
            float es = 0.f, ek = 0.f;
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                float ex = fast_exp2(fmaf(x[u][k], EPI_LOG2E, -ml));
                ex = ok[u] ? ex : 0.f;
                es += ex;
                if (k) ek = fmaf(ex, (float)k, ek);
            }
            s += es;
            a0 += fmaf(es, k0[u], ek);
            a1 = fmaf(es, k1[u], a1);
            a2 = fmaf(es, k2[u], a2);
        }
#pragma unroll
        for (int u = 0; u < SA_UNROLL; ++u) {
            cur_raw[u] = nxt_raw[u]; k0[u] = n0[u]; k1[u] = n1[u]; k2[u] = n2[u]; ok[u] = nok[u];
        }
    }
    // wave reduce: one max reduction, one rescale, then plain sums (no exp inside the shuffle tree)
    float m = wave_max(mx);
    const float f = fast_exp2((mx - m) * EPI_LOG2E);
    s = wave_sum(s * f); a0 = wave_sum(a0 * f); a1 = wave_sum(a1 * f); a2 = wave_sum(a2 * f);
    __shared__ float red[SA_THREADS / 64][5];
    const int lane = tid & 63, wid = tid >> 6;
    if (lane == 0) { red[wid][0] = m; red[wid][1] = s; red[wid][2] = a0; red[wid][3] = a1; red[wid][4] = a2; }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int w = 1; w < SA_THREADS / 64; ++w) merge(m, s, a0, a1, a2, red[w][0], red[w][1], red[w][2], red[w][3], red[w][4]);
        SoftPartial p; p.m = m; p.s = s; p.a0 = a0; p.a1 = a1; p.a2 = a2; p.pad0 = p.pad1 = p.pad2 = 0.f;
        part[(long long)row * nchunk + chunk] = p;
    }
}

template <bool NHWC>
__global__ void softargmax_combine_kernel(const SoftPartial* __restrict__ part, int rows, int nchunk, RowGeom g,
                                          float* __restrict__ xyz, float* __restrict__ row_max, float* __restrict__ row_sum) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    const SoftPartial* p = part + (long long)row * nchunk;
    float m = p[0].m, s = p[0].s, a0 = p[0].a0, a1 = p[0].a1, a2 = p[0].a2;
    for (int c = 1; c < nchunk; ++c) merge(m, s, a0, a1, a2, p[c].m, p[c].s, p[c].a0, p[c].a1, p[c].a2);
    const float inv = 1.f / s;
    float ex, ey, ez, W, H, D;
    if (NHWC) { ez = a0; ex = a1; ey = a2; D = (float)g.E0; W = (float)g.E1; H = (float)g.E2; }
    else      { ex = a0; ey = a1; ez = a2; W = (float)g.E0; H = (float)g.E1; D = (float)g.E2; }
    row_max[row] = m;
    row_sum[row] = s;
    xyz[3 * (long long)row + 0] = ex * inv / W - 0.5f;      // integral_loss.py:81
    xyz[3 * (long long)row + 1] = ey * inv / H - 0.5f;      // :82
    xyz[3 * (long long)row + 2] = ez * inv / D - 0.5f;      // :83
}

// COLSUM (NHWC only): also accumulate, per channel c = j*D + d, the sum over batch and pixels of the gradient AS STORED (rounded to T) into
// col_sums[C] -- the bias gradient of the 1x1 convolution that produced the logits (pose3d_resnet.py:116-122), which otherwise costs one more read of
// the 285 MB gradient (epi_column_sums_bf16).  A thread's vector covers VEC consecutive depth bins d0 .. d0+VEC-1 of its row's joint and keeps them
// for the whole chunk (the launcher checks SA_THREADS * VEC % D == 0); threads that share d0 are D / VEC apart.
template <typename T, int VEC, bool NHWC, bool COLSUM = false>
__global__ __launch_bounds__(SA_THREADS) void softargmax_bwd_kernel(const T* __restrict__ logits, RowGeom g, int nchunk,
                                                                    const float* __restrict__ row_max, const float* __restrict__ row_sum,
                                                                    const float* __restrict__ xyz, const float* __restrict__ gxyz,
                                                                    const float* __restrict__ gscale, T* __restrict__ dlogits, float* __restrict__ col_sums) {
    static_assert(!COLSUM || NHWC, "column sums: channels-last gradients only");
    const int row = blockIdx.x / nchunk;
    const int chunk = blockIdx.x - row * nchunk;
    const int tid = threadIdx.x;
    constexpr int STRIDE = SA_THREADS * VEC;
    const int e_begin = chunk * (STRIDE * SA_ITERS) + tid * VEC;

    Cursor<NHWC> cur;
    cur.init(g, row_base_of<NHWC>(g, row), e_begin, STRIDE);

    const float m = row_max[row];
    const float gs = gscale ? gscale[0] : 1.f;
    const float inv_s = gs / row_sum[row];
    const float gx = gxyz[3 * (long long)row], gy = gxyz[3 * (long long)row + 1], gz = gxyz[3 * (long long)row + 2];
    const float px = xyz[3 * (long long)row] + 0.5f, py = xyz[3 * (long long)row + 1] + 0.5f, pz = xyz[3 * (long long)row + 2] + 0.5f;
    const float kk = gx * px + gy * py + gz * pz;
    float q0, q1, q2;     // coefficient of each abstract coordinate
    if (NHWC) { q0 = gz / (float)g.E0; q1 = gx / (float)g.E1; q2 = gy / (float)g.E2; }
    else      { q0 = gx / (float)g.E0; q1 = gy / (float)g.E1; q2 = gz / (float)g.E2; }
    float cs[COLSUM ? VEC : 1];
#pragma unroll
    for (int k = 0; k < (COLSUM ? VEC : 1); ++k) cs[k] = 0.f;

    int e = e_begin;
#pragma unroll 1
    for (int it = 0; it < SA_ITERS; it += SA_UNROLL) {
        float x[SA_UNROLL][VEC];
        float t0[SA_UNROLL];
        long long off[SA_UNROLL];
        bool ok[SA_UNROLL];
#pragma unroll
        for (int u = 0; u < SA_UNROLL; ++u) {
            ok[u] = (e < g.N);
            off[u] = cur.mem;
            if (ok[u]) VecIO<T, VEC>::load(logits + cur.mem, x[u]);
            t0[u] = q0 * cur.c0 + q1 * cur.c1 + q2 * cur.c2 - kk;
            cur.advance();
            e += STRIDE;
        }
#pragma unroll
        for (int u = 0; u < SA_UNROLL; ++u) {
            if (!ok[u]) continue;
            float d[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const float p = fast_exp2((x[u][k] - m) * EPI_LOG2E) * inv_s;
                d[k] = p * (t0[u] + q0 * (float)k);
            }
            VecIO<T, VEC>::store(dlogits + off[u], d);
            if (COLSUM) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) cs[k] += sizeof(T) == 2 ? bf16_to_f32(f32_to_bf16(d[k])) : d[k];
            }
        }
    }
    if (COLSUM) {
        // threads tid, tid + P, tid + 2P, ... (P = D / VEC, a power of two <= 64) hold the same depth bins: butterfly over the lanes of a wave that
        // differ in the bits above log2(P), then one LDS row per wave, then one atomic per channel of the joint
        __shared__ float wsum[SA_THREADS / 64][64 * VEC];
        const int P = g.E0 / VEC, lane = tid & 63, wid = tid >> 6;
#pragma unroll
        for (int k = 0; k < VEC; ++k)
            for (int o = 32; o >= P; o >>= 1) cs[k] += __shfl_xor(cs[k], o, 64);
        if (lane < P) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) wsum[wid][lane * VEC + k] = cs[k];
        }
        __syncthreads();
        if (tid < g.E0) {                     // (depth bin tid of the row's joint: lane (tid / VEC) of every wave holds it at slot tid)
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < SA_THREADS / 64; ++w) v += wsum[w][tid];
            const int j = row % g.J;
            atomicAdd(col_sums + (long long)j * g.E0 + tid, v);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// row arg-max (inference.py:25-26): first maximum, NaN counts as the maximum (NumPy semantics)
// ---------------------------------------------------------------------------------------------
struct alignas(16) ArgPartial { float v; int pad; long long i; };

__device__ __forceinline__ bool arg_better(float v, long long i, float bv, long long bi) {
    const bool vn = (v != v), bn = (bv != bv);
    if (vn || bn) return vn && (!bn || i < bi);
    return (v > bv) || (v == bv && i < bi);
}

template <typename T, int VEC>
__global__ __launch_bounds__(SA_THREADS) void argmax_partial_kernel(const T* __restrict__ x, int n, int nchunk,
                                                                    ArgPartial* __restrict__ part) {
    const int row = blockIdx.x / nchunk;
    const int chunk = blockIdx.x - row * nchunk;
    const int tid = threadIdx.x;
    constexpr int STRIDE = SA_THREADS * VEC;
    const T* p = x + (long long)row * n;
    float bv = -INFINITY;
    long long bi = 0x7fffffffffffffffLL;
    int e = chunk * (STRIDE * SA_ITERS) + tid * VEC;
    for (int it = 0; it < SA_ITERS; ++it, e += STRIDE) {
        if (e >= n) break;
        float v[VEC];
        VecIO<T, VEC>::load(p + e, v);
#pragma unroll
        for (int k = 0; k < VEC; ++k)
            if (arg_better(v[k], e + k, bv, bi)) { bv = v[k]; bi = e + k; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float v2 = __shfl_xor(bv, o, 64);
        const long long i2 = __shfl_xor(bi, o, 64);
        if (arg_better(v2, i2, bv, bi)) { bv = v2; bi = i2; }
    }
    __shared__ float rv[SA_THREADS / 64];
    __shared__ long long ri[SA_THREADS / 64];
    if ((tid & 63) == 0) { rv[tid >> 6] = bv; ri[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < SA_THREADS / 64; ++w)
            if (arg_better(rv[w], ri[w], bv, bi)) { bv = rv[w]; bi = ri[w]; }
        ArgPartial a; a.v = bv; a.pad = 0; a.i = bi;
        part[(long long)row * nchunk + chunk] = a;
    }
}

__global__ void argmax_combine_kernel(const ArgPartial* __restrict__ part, int rows, int nchunk,
                                      long long* __restrict__ idx, float* __restrict__ val) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    const ArgPartial* p = part + (long long)row * nchunk;
    float bv = p[0].v;
    long long bi = p[0].i;
    for (int c = 1; c < nchunk; ++c)
        if (arg_better(p[c].v, p[c].i, bv, bi)) { bv = p[c].v; bi = p[c].i; }
    idx[row] = bi;
    val[row] = bv;
}

static inline int chunks_for(long long n, int vec, int iters = SA_ITERS) {
    const long long per = (long long)SA_THREADS * vec * iters;
    return (int)((n + per - 1) / per);
}
// forward chunk length (vectors per thread): long chunks for the 8-wide bf16 path halve the number of partials
template <typename T> struct FwdIters { static constexpr int value = SA_ITERS; };
template <> struct FwdIters<unsigned short> { static constexpr int value = 2 * SA_ITERS; };

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace epi

using namespace epi;

extern "C" size_t epi_softargmax3d_workspace_bytes(int B, int J, int D, int H, int W) {
    if (B <= 0 || J <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    // sized for the scalar fallback (most chunks)
    return (size_t)B * J * chunks_for((long long)D * H * W, 1) * sizeof(SoftPartial);
}

template <typename T, bool NHWC>
static int launch_fwd(const T* logits, RowGeom g, int rows, bool vec, float* xyz, float* row_max, float* row_sum,
                      SoftPartial* part, size_t ws_bytes, hipStream_t st) {
    const int VECW = Elem<T>::VEC;
    const int nchunk = vec ? chunks_for(g.N, VECW, FwdIters<T>::value) : chunks_for(g.N, 1);
    if ((size_t)rows * nchunk * sizeof(SoftPartial) > ws_bytes) return EPI_ERR_WORKSPACE;
    const unsigned grid = (unsigned)((long long)rows * nchunk);
    if (vec)
        hipLaunchKernelGGL((softargmax_partial_kernel<T, Elem<T>::VEC, NHWC, FwdIters<T>::value>), dim3(grid), dim3(SA_THREADS), 0, st, logits, g, nchunk, part);
    else
        hipLaunchKernelGGL((softargmax_partial_kernel<T, 1, NHWC, SA_ITERS>), dim3(grid), dim3(SA_THREADS), 0, st, logits, g, nchunk, part);
    EPI_CHECK_LAUNCH();
    hipLaunchKernelGGL((softargmax_combine_kernel<NHWC>), dim3((rows + 127) / 128), dim3(128), 0, st, part, rows, nchunk, g, xyz, row_max, row_sum);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}

static bool make_geom(int layout, int B, int J, int D, int H, int W, RowGeom* g) {
    if (B <= 0 || J <= 0 || D <= 0 || H <= 0 || W <= 0) return false;
    const long long n = (long long)D * H * W;
    if (n > 0x3fffffffLL || (long long)B * J > 0x7fffffffLL) return false;
    g->N = (int)n; g->J = J; g->C = J * D;
    if (layout == EPI_NCHW) { g->E0 = W; g->E1 = H; g->E2 = D; }
    else if (layout == EPI_NHWC) { g->E0 = D; g->E1 = W; g->E2 = H; }
    else return false;
    return true;
}

extern "C" int epi_softargmax3d_fwd(const void* logits, int dtype, int layout, int B, int J, int D, int H, int W,
                                    float* xyz, float* row_max, float* row_sum, void* workspace, size_t workspace_bytes,
                                    epi_stream_t stream) {
    if (!logits || !xyz || !row_max || !row_sum || !workspace) return EPI_ERR_INVALID_ARGUMENT;
    RowGeom g;
    if (!make_geom(layout, B, J, D, H, W, &g)) return EPI_ERR_INVALID_ARGUMENT;
    hipStream_t st = (hipStream_t)stream;
    const int rows = B * J;
    SoftPartial* part = (SoftPartial*)workspace;
    if (dtype == EPI_F32) {
        // a vector never straddles the contiguous axis; NHWC additionally needs C % VEC == 0 for alignment
        const bool vec = aligned16(logits) && (g.E0 % 4 == 0) && (layout == EPI_NCHW || g.C % 4 == 0);
        return layout == EPI_NCHW ? launch_fwd<float, false>((const float*)logits, g, rows, vec, xyz, row_max, row_sum, part, workspace_bytes, st)
                                  : launch_fwd<float, true>((const float*)logits, g, rows, vec, xyz, row_max, row_sum, part, workspace_bytes, st);
    } else if (dtype == EPI_BF16) {
        const bool vec = aligned16(logits) && (g.E0 % 8 == 0) && (layout == EPI_NCHW || g.C % 8 == 0);
        return layout == EPI_NCHW ? launch_fwd<unsigned short, false>((const unsigned short*)logits, g, rows, vec, xyz, row_max, row_sum, part, workspace_bytes, st)
                                  : launch_fwd<unsigned short, true>((const unsigned short*)logits, g, rows, vec, xyz, row_max, row_sum, part, workspace_bytes, st);
    }
    return EPI_ERR_UNSUPPORTED;
}

template <typename T, bool NHWC>
static int launch_bwd(const T* logits, RowGeom g, int rows, bool vec, const float* row_max, const float* row_sum,
                      const float* xyz, const float* gxyz, const float* gscale, T* dlogits, hipStream_t st, float* col_sums = nullptr, int* col_sums_done = nullptr) {
    const int nchunk = chunks_for(g.N, vec ? Elem<T>::VEC : 1);
    const unsigned grid = (unsigned)((long long)rows * nchunk);
    if (col_sums_done) *col_sums_done = 0;
    if (vec) {
        constexpr int V = Elem<T>::VEC;
        // the fused column sums: a thread must keep its depth bins from vector to vector, and the threads that share them must be a power of two apart
        const int P = g.E0 / V;
        const bool colsum_ok = NHWC && col_sums && col_sums_done && !deterministic() && g.E0 % V == 0 && (SA_THREADS * V) % g.E0 == 0 && P >= 1 && P <= 64 &&
                               (P & (P - 1)) == 0 && g.E0 <= SA_THREADS;
        if (colsum_ok) {
            if constexpr (NHWC) {
                hipLaunchKernelGGL((softargmax_bwd_kernel<T, V, true, true>), dim3(grid), dim3(SA_THREADS), 0, st, logits, g, nchunk, row_max, row_sum, xyz, gxyz, gscale,
                                   dlogits, col_sums);
                *col_sums_done = 1;
            }
        } else {
            hipLaunchKernelGGL((softargmax_bwd_kernel<T, V, NHWC>), dim3(grid), dim3(SA_THREADS), 0, st, logits, g, nchunk, row_max, row_sum, xyz, gxyz, gscale, dlogits,
                               (float*)nullptr);
        }
    } else {
        hipLaunchKernelGGL((softargmax_bwd_kernel<T, 1, NHWC>), dim3(grid), dim3(SA_THREADS), 0, st, logits, g, nchunk, row_max, row_sum, xyz, gxyz, gscale, dlogits,
                           (float*)nullptr);
    }
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}

static int softargmax3d_bwd_impl(const void* logits, int dtype, int layout, int B, int J, int D, int H, int W,
                                 const float* row_max, const float* row_sum, const float* xyz, const float* grad_xyz,
                                 const float* grad_scale, void* dlogits, float* col_sums, int* col_sums_done, epi_stream_t stream) {
    if (col_sums_done) *col_sums_done = 0;
    if (!logits || !row_max || !row_sum || !xyz || !grad_xyz || !dlogits) return EPI_ERR_INVALID_ARGUMENT;
    RowGeom g;
    if (!make_geom(layout, B, J, D, H, W, &g)) return EPI_ERR_INVALID_ARGUMENT;
    hipStream_t st = (hipStream_t)stream;
    const int rows = B * J;
    if (dtype == EPI_F32) {
        const bool vec = aligned16(logits) && aligned16(dlogits) && (g.E0 % 4 == 0) && (layout == EPI_NCHW || g.C % 4 == 0);
        return layout == EPI_NCHW ? launch_bwd<float, false>((const float*)logits, g, rows, vec, row_max, row_sum, xyz, grad_xyz, grad_scale, (float*)dlogits, st)
                                  : launch_bwd<float, true>((const float*)logits, g, rows, vec, row_max, row_sum, xyz, grad_xyz, grad_scale, (float*)dlogits, st,
                                                            col_sums, col_sums_done);
    } else if (dtype == EPI_BF16) {
        const bool vec = aligned16(logits) && aligned16(dlogits) && (g.E0 % 8 == 0) && (layout == EPI_NCHW || g.C % 8 == 0);
        return layout == EPI_NCHW ? launch_bwd<unsigned short, false>((const unsigned short*)logits, g, rows, vec, row_max, row_sum, xyz, grad_xyz, grad_scale, (unsigned short*)dlogits, st)
                                  : launch_bwd<unsigned short, true>((const unsigned short*)logits, g, rows, vec, row_max, row_sum, xyz, grad_xyz, grad_scale, (unsigned short*)dlogits, st,
                                                                     col_sums, col_sums_done);
    }
    return EPI_ERR_UNSUPPORTED;
}

extern "C" int epi_softargmax3d_bwd(const void* logits, int dtype, int layout, int B, int J, int D, int H, int W,
                                    const float* row_max, const float* row_sum, const float* xyz, const float* grad_xyz,
                                    const float* grad_scale, void* dlogits, epi_stream_t stream) {
    return softargmax3d_bwd_impl(logits, dtype, layout, B, J, D, H, W, row_max, row_sum, xyz, grad_xyz, grad_scale, dlogits, nullptr, nullptr, stream);
}

// epi_softargmax3d_bwd that also delivers the per-channel sums of the gradient it writes (channels-last logits): col_sums [J*D] f32, ZERO on entry.
// *col_sums_done = 1: col_sums holds, per channel, the sum over batch and pixels of dlogits as stored -- the bias gradient of the 1x1 convolution
// that produced the logits; 0 (NCHW, an unsuitable depth extent, deterministic mode: the sums are fp32 atomics): col_sums untouched.
extern "C" int epi_softargmax3d_bwd_colsums(const void* logits, int dtype, int layout, int B, int J, int D, int H, int W,
                                            const float* row_max, const float* row_sum, const float* xyz, const float* grad_xyz,
                                            const float* grad_scale, void* dlogits, float* col_sums, int* col_sums_done, epi_stream_t stream) {
    if (!col_sums || !col_sums_done) return EPI_ERR_INVALID_ARGUMENT;
    return softargmax3d_bwd_impl(logits, dtype, layout, B, J, D, H, W, row_max, row_sum, xyz, grad_xyz, grad_scale, dlogits, col_sums, col_sums_done, stream);
}

extern "C" size_t epi_argmax_workspace_bytes(int rows, int n) {
    if (rows <= 0 || n <= 0) return 0;
    return (size_t)rows * chunks_for(n, 1) * sizeof(ArgPartial);
}

template <typename T>
static int launch_argmax(const T* x, int rows, int n, bool vec, long long* idx, float* val, ArgPartial* part, size_t ws, hipStream_t st) {
    const int nchunk = chunks_for(n, vec ? Elem<T>::VEC : 1);
    if ((size_t)rows * nchunk * sizeof(ArgPartial) > ws) return EPI_ERR_WORKSPACE;
    const unsigned grid = (unsigned)((long long)rows * nchunk);
    if (vec)
        hipLaunchKernelGGL((argmax_partial_kernel<T, Elem<T>::VEC>), dim3(grid), dim3(SA_THREADS), 0, st, x, n, nchunk, part);
    else
        hipLaunchKernelGGL((argmax_partial_kernel<T, 1>), dim3(grid), dim3(SA_THREADS), 0, st, x, n, nchunk, part);
    EPI_CHECK_LAUNCH();
    hipLaunchKernelGGL(argmax_combine_kernel, dim3((rows + 127) / 128), dim3(128), 0, st, part, rows, nchunk, idx, val);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}

extern "C" int epi_argmax_rows(const void* x, int dtype, int rows, int n, int64_t* idx, float* val, void* workspace,
                               size_t workspace_bytes, epi_stream_t stream) {
    if (!x || !idx || !val || !workspace || rows <= 0 || n <= 0) return EPI_ERR_INVALID_ARGUMENT;
    if ((long long)rows * ((n + 255) / 256) > 0x7fffffffLL) return EPI_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == EPI_F32)
        return launch_argmax<float>((const float*)x, rows, n, aligned16(x) && n % 4 == 0, (long long*)idx, val, (ArgPartial*)workspace, workspace_bytes, st);
    if (dtype == EPI_BF16)
        return launch_argmax<unsigned short>((const unsigned short*)x, rows, n, aligned16(x) && n % 8 == 0, (long long*)idx, val, (ArgPartial*)workspace, workspace_bytes, st);
    return EPI_ERR_UNSUPPORTED;
}
