// bf16 MFMA implicit-GEMM for the deconvolution head (gfx950).
//
// One kernel template computes  C[m][n] = sum_k A(m,k) * Bt[n][k]  with fp32 accumulation on
// v_mfma_f32_32x32x16_bf16, where the A operand is either a plain row-major matrix or an implicit
// (never materialised) convolution gather over an NHWC activation tensor:
//   * ConvTranspose2d(k=4,s=2,p=1) forward  (pose3d_resnet.py:158-183) = 4 output-parity phases, each a
//     2x2-tap gather GEMM with K = 4*Cin;
//   * its backward-data = a 4x4 stride-2 gather GEMM with K = 16*Cout;
//   * the final 1x1 convolution (pose3d_resnet.py:116-122) forward / backward-data = plain GEMMs.
// Tile 128x128x64, 256 threads (2x2 waves, each 2x2 MFMA 32x32 tiles), register-staged global->LDS with the next
// K tile's loads in flight during the MFMAs, double-buffered LDS (64 KiB -> 2 workgroups / CU), XOR-swizzled
// 16-byte chunks so both the 8-lane ds_write_b128 groups and the 16-lane ds_read_b128 groups are conflict-free.
#include "common.h"

namespace epi {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GBM = 128, GBN = 128, GBK = 64, GTHREADS = 256;
constexpr int TILE_BYTES = GBM * GBK * 2;          // 16 KiB per operand tile

struct GemmGather {        // maps GEMM row m / K tile to an NHWC source pixel
    int enabled;           // 0: plain A[m*lda + k]
    int Hg, Wg;            // rows enumerate (n, i, j) over an Hg x Wg grid
    int Hs, Ws, Cs;        // source tensor [n][Hs][Ws][Cs]
    int stride;            // source pixel = (i*stride + dy[tap], j*stride + dx[tap]); k = tap*Cs + c
    int dy[16], dx[16];
};
struct GemmScatter {       // maps GEMM row m to an output row
    int enabled;           // 0: row m
    int Hg, Wg;            // same (n, i, j) enumeration
    int Ho, Wo;            // output tensor [n][Ho][Wo][ldc]
    int so, oy, ox;        // output pixel = (i*so + oy, j*so + ox)
};
struct GemmArgs {
    const unsigned short* A;
    const unsigned short* Bt;
    void* C;
    const float* bias;     // per output column n, or null
    int M, N, K, lda, ldb, ldc;
    GemmGather ga;
    GemmScatter sc;
};

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <bool OUT_F32>
__global__ __launch_bounds__(GTHREADS, 2) void head_gemm_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [2 buffers][A tile | B tile]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int tiles_n = (p.N + GBN - 1) / GBN;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
    const int m0 = tile_m * GBM, n0 = tile_n * GBN;

    // ---- staging roles: thread t moves chunk (t & 7) of rows (t >> 3) + 32*pass of both tiles ----
    const int srow = tid >> 3, schunk = tid & 7;
    long long a_base[4];
    int a_iy[4], a_jx[4];
    bool a_ok[4];
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int m = m0 + ps * 32 + srow;
        a_ok[ps] = m < p.M;
        if (p.ga.enabled) {
            const int hw = p.ga.Hg * p.ga.Wg;
            const int n = m / hw, rem = m - n * hw;
            const int i = rem / p.ga.Wg, j = rem - i * p.ga.Wg;
            a_iy[ps] = i * p.ga.stride;
            a_jx[ps] = j * p.ga.stride;
            a_base[ps] = (long long)n * p.ga.Hs;
        } else {
            a_iy[ps] = a_jx[ps] = 0;
            a_base[ps] = (long long)m * p.lda;
        }
    }
    uint4v ra[4], rb[4];
    auto load_tiles = [&](int k0) {
        int tap = 0, c0 = k0;
        if (p.ga.enabled) { tap = k0 / p.ga.Cs; c0 = k0 - tap * p.ga.Cs; }
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            bool ok = a_ok[ps];
            long long off;
            if (p.ga.enabled) {
                const int y = a_iy[ps] + p.ga.dy[tap], x = a_jx[ps] + p.ga.dx[tap];
                ok = ok && (unsigned)y < (unsigned)p.ga.Hs && (unsigned)x < (unsigned)p.ga.Ws;
                off = ((a_base[ps] + y) * p.ga.Ws + x) * p.ga.Cs + c0 + schunk * 8;
            } else {
                off = a_base[ps] + k0 + schunk * 8;
            }
            uint4v z; z.x = z.y = z.z = z.w = 0u;
            ra[ps] = ok ? *reinterpret_cast<const uint4v*>(p.A + off) : z;
            const int n = n0 + ps * 32 + srow;
            rb[ps] = (n < p.N) ? *reinterpret_cast<const uint4v*>(p.Bt + (long long)n * p.ldb + k0 + schunk * 8) : z;
        }
    };
    auto store_tiles = [&](int buf) {
        char* a_s = smem + buf * 2 * TILE_BYTES;
        char* b_s = a_s + TILE_BYTES;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int o = lds_off(ps * 32 + srow, schunk);
            *reinterpret_cast<uint4v*>(a_s + o) = ra[ps];
            *reinterpret_cast<uint4v*>(b_s + o) = rb[ps];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.K / GBK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    const int frow = lane & 31, fhalf = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tiles((kt + 1) * GBK);
        const char* a_s = smem + buf * 2 * TILE_BYTES;
        const char* b_s = a_s + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 af[2], bfr[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const uint4v va = *reinterpret_cast<const uint4v*>(a_s + lds_off(wm * 64 + t * 32 + frow, ks * 2 + fhalf));
                const uint4v vb = *reinterpret_cast<const uint4v*>(b_s + lds_off(wn * 64 + t * 32 + frow, ks * 2 + fhalf));
                af[t] = __builtin_bit_cast(bf16x8, va);
                bfr[t] = __builtin_bit_cast(bf16x8, vb);
            }
            // operands swapped: D[i][j] with i = output column n (register rows), j = output row m (lane & 31),
            // so that a lane ends up holding 4 consecutive columns of one row -> 8-byte bf16 stores.
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj)
                    acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[tj], af[ti], acc[ti][tj], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds, for tile (ti, tj): row m = wm*64 + ti*32 + (lane & 31),
    //      columns n = wn*64 + tj*32 + 8*q + 4*(lane >> 5) + e   for reg = 4*q + e ----
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
        const int m = m0 + wm * 64 + ti * 32 + frow;
        const bool row_ok = m < p.M;
        long long orow = m;
        if (p.sc.enabled) {
            const int hw = p.sc.Hg * p.sc.Wg;
            const int n = m / hw, rem = m - n * hw;
            const int i = rem / p.sc.Wg, j = rem - i * p.sc.Wg;
            orow = ((long long)n * p.sc.Ho + i * p.sc.so + p.sc.oy) * p.sc.Wo + j * p.sc.so + p.sc.ox;
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
                        if (n + 3 < p.N) {
                            uint2 t;
                            t.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                            t.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                            *reinterpret_cast<uint2*>(c) = t;
                        } else for (int e = 0; e < 4 && n + e < p.N; ++e) c[e] = f32_to_bf16(v[e]);
                    }
                }
            }
        }
    }
}

}  // namespace epi

using namespace epi;

static int launch_gemm(const GemmArgs& a, bool out_f32, hipStream_t st) {
    if (!a.A || !a.Bt || !a.C || a.M <= 0 || a.N <= 0 || a.K <= 0) return EPI_ERR_INVALID_ARGUMENT;
    if (a.K % GBK || a.lda % 8 || a.ldb % 8 || a.ldc % 4) return EPI_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(a.A) | reinterpret_cast<uintptr_t>(a.Bt) | reinterpret_cast<uintptr_t>(a.C)) & 15u) return EPI_ERR_UNSUPPORTED;
    if (a.ga.enabled && (a.ga.Cs % GBK)) return EPI_ERR_UNSUPPORTED;     // a K tile must not straddle two taps
    const long long tiles = (long long)((a.M + GBM - 1) / GBM) * ((a.N + GBN - 1) / GBN);
    if (tiles > 0x7fffffffLL) return EPI_ERR_UNSUPPORTED;
    const size_t lds = 4 * TILE_BYTES;
    if (out_f32) hipLaunchKernelGGL(head_gemm_kernel<true>, dim3((unsigned)tiles), dim3(GTHREADS), lds, st, a);
    else hipLaunchKernelGGL(head_gemm_kernel<false>, dim3((unsigned)tiles), dim3(GTHREADS), lds, st, a);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}

extern "C" int epi_gemm_bf16(const void* A, int lda, const void* Bt, int ldb, void* C, int ldc, int c_dtype, int M, int N, int K,
                             const float* bias, epi_stream_t stream) {
    GemmArgs a = {};
    a.A = (const unsigned short*)A; a.Bt = (const unsigned short*)Bt; a.C = C; a.bias = bias;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    if (c_dtype != EPI_BF16 && c_dtype != EPI_F32) return EPI_ERR_UNSUPPORTED;
    return launch_gemm(a, c_dtype == EPI_F32, (hipStream_t)stream);
}

// ConvTranspose2d(k=4, s=2, p=1), NHWC bf16:  x [B][H][W][Cin]  ->  y [B][2H][2W][Cout]  (raw, pre-BatchNorm).
// w_phase: [4 phases][Cout][4 taps * Cin] packed by epi_deconv4x4s2_pack_weight (phase = 2*(oh&1) + (ow&1)).
extern "C" int epi_deconv4x4s2_fwd(const void* x, const void* w_phase, void* y, int B, int H, int W, int Cin, int Cout,
                                   epi_stream_t stream) {
    if (!x || !w_phase || !y || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return EPI_ERR_INVALID_ARGUMENT;
    if (Cin % GBK || Cout % 4) return EPI_ERR_UNSUPPORTED;
    for (int ph = 0; ph < 2; ++ph) {
        for (int pw = 0; pw < 2; ++pw) {
            GemmArgs a = {};
            a.A = (const unsigned short*)x;
            a.Bt = (const unsigned short*)w_phase + (size_t)(2 * ph + pw) * Cout * 4 * Cin;
            a.C = y;
            a.M = B * H * W; a.N = Cout; a.K = 4 * Cin; a.lda = 0; a.ldb = 4 * Cin; a.ldc = Cout;
            a.ga.enabled = 1; a.ga.Hg = H; a.ga.Wg = W; a.ga.Hs = H; a.ga.Ws = W; a.ga.Cs = Cin; a.ga.stride = 1;
            // oh = 2*ih - 1 + kh.  oh even: kh in {1,3} -> ih = i, i-1;  oh odd: kh in {0,2} -> ih = i+1, i
            const int dyv[2] = {ph ? 1 : 0, ph ? 0 : -1}, dxv[2] = {pw ? 1 : 0, pw ? 0 : -1};
            for (int ty = 0; ty < 2; ++ty)
                for (int tx = 0; tx < 2; ++tx) { a.ga.dy[2 * ty + tx] = dyv[ty]; a.ga.dx[2 * ty + tx] = dxv[tx]; }
            a.sc.enabled = 1; a.sc.Hg = H; a.sc.Wg = W; a.sc.Ho = 2 * H; a.sc.Wo = 2 * W; a.sc.so = 2; a.sc.oy = ph; a.sc.ox = pw;
            const int st = launch_gemm(a, false, (hipStream_t)stream);
            if (st != EPI_OK) return st;
        }
    }
    return EPI_OK;
}

// Backward-data of the same layer:  dy [B][2H][2W][Cout] -> dx [B][H][W][Cin];
// w_bwd: [Cin][16 taps * Cout] packed by epi_deconv4x4s2_pack_weight (tap = kh*4 + kw).
extern "C" int epi_deconv4x4s2_bwd_data(const void* dy, const void* w_bwd, void* dx, int B, int H, int W, int Cin, int Cout,
                                        epi_stream_t stream) {
    if (!dy || !w_bwd || !dx || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return EPI_ERR_INVALID_ARGUMENT;
    if (Cout % GBK || Cin % 4) return EPI_ERR_UNSUPPORTED;
    GemmArgs a = {};
    a.A = (const unsigned short*)dy; a.Bt = (const unsigned short*)w_bwd; a.C = dx;
    a.M = B * H * W; a.N = Cin; a.K = 16 * Cout; a.lda = 0; a.ldb = 16 * Cout; a.ldc = Cin;
    a.ga.enabled = 1; a.ga.Hg = H; a.ga.Wg = W; a.ga.Hs = 2 * H; a.ga.Ws = 2 * W; a.ga.Cs = Cout; a.ga.stride = 2;
    for (int kh = 0; kh < 4; ++kh)
        for (int kw = 0; kw < 4; ++kw) { a.ga.dy[4 * kh + kw] = kh - 1; a.ga.dx[4 * kh + kw] = kw - 1; }   // oh = 2*ih - 1 + kh
    return launch_gemm(a, false, (hipStream_t)stream);
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
