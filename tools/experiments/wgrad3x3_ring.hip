// EXPERIMENT, NOT COMPILED INTO THE LIBRARY (round 4): the weight gradient of a 3x3 / stride-1 / pad-1 convolution with the
// reduction over zero-padded positions and the input rows in an LDS ring.  Bit-for-bit parity with the TN kernel's results on tests/test_hip_conv.py /
// tests/test_hip_head.py (-k weight), but 51 .. 60 us per ResNet-50 layer at batch 32 against the TN kernel's 37 .. 40 us (~3 us per 64-position tile).  Not the DMA
// (three tiles in flight changed nothing); the operand reads (20 ds_read_b64_tr_b16 per 9 MFMAs, 3.1 ns each per CU: tools/lds_tr_probe.hip) explain ~1 us per tile;
// the rest -- 147 KB of fp32 result per workgroup, 19 MB of slabs per layer, only 128 workgroups -- was not separated.  Kept as a starting point: the device part below dropped into csrc/head_gemm.hip behind
// tn_body (it uses glds16, tn_swz, tr_frag's layout, xcd_remap, GemmTnArgs), the host part in front of tn_plan; launch_tn called launch_w3 when
// w3_eligible(a), epi_gemm_tn_workspace_bytes covered w3_nsplit's slabs.

// ======== device part ========
// ---- weight gradient of a 3x3 / stride-1 / pad-1 convolution with ring-staged input rows (round 4) -------------------------------------------------
// The TN kernel above treats the nine taps as columns: every 128-column tile (two taps at 64 input channels) re-stages its dy tile and its own
// shifted x rows -- 24 KB per K tile for 32 MFMAs, and the launches are bound by that fill traffic through L2.  Here the reduction runs over the
// positions g of the ZERO-PADDED images ((H + 2) x (W + 2) per image; halo positions carry a zero dy row and a zero x row), where a tap is a uniform
// shift:   dW[co][kh][kw][ci] = sum_g dy_pad[g][co] * x_pad[g + (kh - 1)(W + 2) + (kw - 1)][ci].
// A workgroup owns 64 output channels x 9 taps x 64 input channels (4 waves: output-channel half x input-channel half, 9 accumulator tiles each)
// and walks 64 positions per K tile: the dy rows go through two 8 KB stages, the x rows live in an LDS RING indexed by position (mod ring size)
// that receives only the 64 NEW rows per K tile -- 16 KB staged per K tile for 144 MFMAs.  Row pieces are 8 positions (1 KB DMA), so the ring's
// leading edge runs HALO_A = roundup(W + 3, 8) positions ahead, and DEPTH = 3 K tiles are in flight behind the one being consumed (counted vmcnt waits):
// 512 ring rows cover 64 DEPTH + 127 + 2 HALO_A positions up to W = 93.  LDS: 4 dy stages (32 KB) + ring (64 KB), one workgroup per CU.
// Padding costs (H + 2)(W + 2) / (H W) more MFMA work: 6 % at 64 x 64, 27 % at 16 x 16, 56 % at 8 x 8.
constexpr int W3_THREADS = 256, W3_DEPTH = 3, W3_DY_BYTES = (W3_DEPTH + 1) * 8192;      // DEPTH K tiles in flight behind the one being consumed
__host__ __device__ inline int w3_halo(int W) { return (W + 3 + 7) / 8 * 8; }
__host__ __device__ inline int w3_ring_rows(int W) { return 64 * W3_DEPTH + 127 + 2 * w3_halo(W) < 512 ? 512 : 1024; }    // a power of two above the live span (W <= 93: 512)

__device__ __forceinline__ void wgrad3x3_body(const GemmTnArgs& p, const int tile_id, const int split, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int H = p.Hs, W = p.Ws, Cin = p.Cs;
    const int WP = W + 2, HP = H + 2, P = HP * WP;
    const int nimg = p.R / (H * W);
    const int G = nimg * P;
    const int n_ci = Cin >> 6;
    const int co0 = (tile_id / n_ci) * 64, ci0 = (tile_id % n_ci) * 64;
    const int kt_total = (G + 63) >> 6, kt_per = (kt_total + p.nsplit - 1) / p.nsplit;
    const int kt_begin = split * kt_per, kt_end = min(kt_total, kt_begin + kt_per);
    const int halo = w3_halo(W), mask = w3_ring_rows(W) - 1;
    char* ring = smem + W3_DY_BYTES;
    const char* zero_src = reinterpret_cast<const char*>(epi_zero_chunk);
    const int lrow = lane >> 3, lchunk = lane & 7;       // a 1 KB piece = 8 rows x 8 chunks of 16 bytes

    // ---- DMA sources.  A padded position decomposes into (image n, padded row r, padded column c); real pixels are 1 <= r <= H, 1 <= c <= W.
    //      The four pieces a lane feeds per K tile (two of the dy tile, two of the ring's leading edge) keep their (n, r, c) and ADVANCE by 64
    //      positions per tile with two conditional carries -- no division in the loop (the prologue's ring fill divides once per piece). ----
    auto source = [&](int n, int r, int c, int lds_row, const unsigned short* base, int ld, int c0) -> const void* {
        if (n < 0 || n >= nimg || r < 1 || r > H || c < 1 || c > W) return zero_src;
        const long long pix = ((long long)n * H + (r - 1)) * W + (c - 1);
        return base + pix * ld + c0 + ((lchunk ^ tn_swz<128>(lds_row)) << 3);
    };
    auto decompose = [&](int pos, int& n, int& r, int& c) {
        if (pos < 0) { n = -1; r = c = 0; return; }
        n = pos / P;
        const int q = pos - n * P;
        r = q / WP;
        c = q - r * WP;
    };
    const int adv_c = 64 % WP, adv_r = (64 / WP) % HP, adv_n = 64 / P;
    auto advance = [&](int& n, int& r, int& c) {
        c += adv_c;
        if (c >= WP) { c -= WP; r += 1; }
        r += adv_r;
        if (r >= HP) { r -= HP; n += 1; }
        n += adv_n;
    };
    int dn[2], dr[2], dc[2], xn[2], xr[2], xc[2];          // state of the lane's two dy pieces / two ring pieces for the NEXT issue
    const int uwid = __builtin_amdgcn_readfirstlane(wid);
    auto issue_dy = [&](int buf) {                           // dy tile -> stage buf: 8 pieces, two per wave
        char* dst = smem + buf * 8192 + uwid * 2048;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int row = (wid * 2 + q) * 8 + lrow;
            glds16(source(dn[q], dr[q], dc[q], row, p.A, p.lda, co0), dst + q * 1024);
            advance(dn[q], dr[q], dc[q]);
        }
    };
    auto issue_edge = [&](int pos0) {                        // the 64 positions the ring's leading edge advances by: 8 pieces, two per wave
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int row0 = (pos0 + (uwid * 2 + q) * 8) & mask;
            glds16(source(xn[q], xr[q], xc[q], row0 + lrow, p.B, p.ldb, ci0), ring + row0 * 128);
            advance(xn[q], xr[q], xc[q]);
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int lane_c = lane & 15, grp = lane >> 4;
    const int cblk = 16 * (grp & 1), khalf = grp >> 1;
    const int cohalf = wid & 1, cihalf = wid >> 1;
    // ---- ring reads: a tap's operand rows are (tile base + ks * 16) + rel[t] + {0, 4}, rel[t] = 8 * khalf + (lane_c >> 2) + shift of the tap.  The tile base
    //      is a multiple of 16, so the swizzle term (bit 1 of the row) and the column part of the address are constants of (lane, tap). ----
    int rel[9], colpart[9];
    {
        const int chunk = ((cihalf * 32 + cblk) >> 3) + ((lane_c & 3) >> 1);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            rel[t] = 8 * khalf + (lane_c >> 2) + (t / 3 - 1) * WP + (t % 3 - 1);
            colpart[t] = ((chunk ^ tn_swz<128>(rel[t] & 3)) << 4) + 8 * (lane_c & 1);
        }
    }
    // LDS operand reads as inline assembly: hipcc puts `s_waitcnt vmcnt(0)` in front of every LDS load it can see while LDS-DMA is outstanding (it cannot
    // prove that a ds_read_b64_tr_b16 does not alias a pending global_load_lds), which would serialise the DEPTH tiles in flight.  The reads of a k step
    // are issued together, their completion is awaited by ONE counted s_waitcnt that names the fragments (so that no consumer moves above it).
    struct Frag { s16x4 lo, hi; };
    const unsigned ring_lds = (unsigned)(size_t)((__attribute__((address_space(3))) char*)ring);
    const unsigned dy_lds = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
    auto lds_tr = [&](unsigned addr) -> s16x4 {
        s16x4 v;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
        return v;
    };
    auto ring_frag = [&](int base, int t) -> Frag {
        const int r0 = (base + rel[t]) & mask, r1 = (base + rel[t] + 4) & mask;
        Frag f;
        f.lo = lds_tr(ring_lds + r0 * 128 + colpart[t]);
        f.hi = lds_tr(ring_lds + r1 * 128 + colpart[t]);
        return f;
    };
    // the dy operand of k step ks from stage `slot`: rows ks * 16 + 8 * khalf + (lane_c >> 2) (+ 4), output channels cohalf * 32 + cblk ..
    int a_off;
    {
        const int row = 8 * khalf + (lane_c >> 2);
        const int chunk = ((cohalf * 32 + cblk) >> 3) + ((lane_c & 3) >> 1);
        a_off = row * 128 + ((chunk ^ tn_swz<128>(row)) << 4) + 8 * (lane_c & 1);          // (+ ks * 16 rows and + 4 rows keep the swizzle term)
    }
    auto dy_frag = [&](int slot, int ks) -> Frag {
        Frag f;
        f.lo = lds_tr(dy_lds + slot * 8192 + ks * 2048 + a_off);
        f.hi = lds_tr(dy_lds + slot * 8192 + ks * 2048 + a_off + 512);
        return f;
    };
    // A k step's 20 operand reads go out as two groups of 10 (lgkmcnt counts to 15): X = the dy fragment + taps 0 .. 3, Y = taps 4 .. 8.
    struct GroupX { Frag a, b[4]; };
    struct GroupY { Frag b[5]; };
    auto read_x = [&](int slot, int kt, int ks) -> GroupX {
        GroupX g;
        g.a = dy_frag(slot, ks);
#pragma unroll
        for (int t = 0; t < 4; ++t) g.b[t] = ring_frag(kt * 64 + ks * 16, t);
        return g;
    };
    auto read_y = [&](int kt, int ks) -> GroupY {
        GroupY g;
#pragma unroll
        for (int t = 0; t < 5; ++t) g.b[t] = ring_frag(kt * 64 + ks * 16, 4 + t);
        return g;
    };
    // wait until at most N LDS reads issued after this group's are outstanding (LDS reads return in order), naming the group's registers so that
    // no consumer moves above the wait
#define W3_WAIT_X(g, N)                                                                                                                              \
    asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                                                         \
                 : "+v"((g).a.lo), "+v"((g).a.hi), "+v"((g).b[0].lo), "+v"((g).b[0].hi), "+v"((g).b[1].lo), "+v"((g).b[1].hi), "+v"((g).b[2].lo),   \
                   "+v"((g).b[2].hi), "+v"((g).b[3].lo), "+v"((g).b[3].hi))
#define W3_WAIT_Y(g, N)                                                                                                                              \
    asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                                                         \
                 : "+v"((g).b[0].lo), "+v"((g).b[0].hi), "+v"((g).b[1].lo), "+v"((g).b[1].hi), "+v"((g).b[2].lo), "+v"((g).b[2].hi), "+v"((g).b[3].lo), \
                   "+v"((g).b[3].hi), "+v"((g).b[4].lo), "+v"((g).b[4].hi))
    auto as_bf16x8 = [](const Frag& f) -> bf16x8 {
        struct { s16x4 lo, hi; } v = {f.lo, f.hi};
        return __builtin_bit_cast(bf16x8, v);
    };
    auto mfma_x = [&](const GroupX& g) {
        const bf16x8 af = as_bf16x8(g.a);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, as_bf16x8(g.b[t]), acc[t], 0, 0, 0);
    };
    auto mfma_y = [&](const Frag& a, const GroupY& g) {
        const bf16x8 af = as_bf16x8(a);
#pragma unroll
        for (int t = 0; t < 5; ++t) acc[4 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, as_bf16x8(g.b[t]), acc[4 + t], 0, 0, 0);
    };

    constexpr int D = W3_DEPTH, DPT = 4;              // DMA instructions per wave and K tile: two dy pieces + two ring pieces
    if (kt_begin < kt_end) {
        // prologue: the ring from kt_begin * 64 - halo up to the first tile's leading edge (pieces round-robin over the waves), then the dy tiles and
        // leading edges of the first D tiles
        const int pos0 = kt_begin * 64 - halo, npieces = (64 + 2 * halo) >> 3;
        for (int q = uwid; q < npieces; q += 4) {
            const int pos = pos0 + q * 8, row0 = pos & mask;
            int n, r, c;
            decompose(pos + lrow, n, r, c);
            glds16(source(n, r, c, row0 + lrow, p.B, p.ldb, ci0), ring + row0 * 128);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            decompose(kt_begin * 64 + (wid * 2 + q) * 8 + lrow, dn[q], dr[q], dc[q]);
            decompose(kt_begin * 64 + 64 + halo + (wid * 2 + q) * 8 + lrow, xn[q], xr[q], xc[q]);
        }
        issue_dy(0);
        issue_edge(kt_begin * 64 + 64 + halo);       // (the edge of tile kt_begin + 1; tile kt_begin's own rows came with the fill)
#pragma unroll
        for (int d = 1; d < D; ++d)
            if (kt_begin + d < kt_end) { issue_dy(d); issue_edge((kt_begin + d) * 64 + 64 + halo); }
    }
    // Per tile: dy(kt) + edge(kt + 1) are one DMA group of DPT instructions per wave (the first tile's group also carries the ring fill, in front).
    // Tile kt needs group kt - 1's edge and group kt's dy: waiting until at most the LATER groups are outstanding covers both.
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int later = min(D - 1, kt_end - 1 - kt);
        if (later >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DPT) : "memory");
        else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();               // every wave's pieces of tile kt have landed; every wave is done reading tile kt - 1
        const int slot = (kt - kt_begin) % (D + 1);
        if (kt + D < kt_end) { issue_dy((slot + D) % (D + 1)); issue_edge((kt + D) * 64 + 64 + halo); }
        // one group (10 reads) is in flight while the MFMAs of the group before it run (lgkmcnt is also bumped by scalar loads: an extra count only
        // makes the wait longer, never shorter than the in-order LDS reads it is meant for)
        GroupX x0 = read_x(slot, kt, 0);
        GroupY y0 = read_y(kt, 0);
        W3_WAIT_X(x0, 10);
        mfma_x(x0);
        GroupX x1 = read_x(slot, kt, 1);
        W3_WAIT_Y(y0, 10);
        mfma_y(x0.a, y0);
        GroupY y1 = read_y(kt, 1);
        W3_WAIT_X(x1, 10);
        mfma_x(x1);
        GroupX x2 = read_x(slot, kt, 2);
        W3_WAIT_Y(y1, 10);
        mfma_y(x1.a, y1);
        GroupY y2 = read_y(kt, 2);
        W3_WAIT_X(x2, 10);
        mfma_x(x2);
        GroupX x3 = read_x(slot, kt, 3);
        W3_WAIT_Y(y2, 10);
        mfma_y(x2.a, y2);
        GroupY y3 = read_y(kt, 3);
        W3_WAIT_X(x3, 10);
        mfma_x(x3);
        W3_WAIT_Y(y3, 0);
        mfma_y(x3.a, y3);
    }
#undef W3_WAIT_X
#undef W3_WAIT_Y
    // D[i][j]: lane holds column j = lane & 31 (input channel), rows i = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (output channel)
    const int fcol = lane & 31, fhalf = lane >> 5;
    const bool direct = p.nsplit == 1;
    float* slab = reinterpret_cast<float*>(p.C) + (direct ? 0 : (long long)split * p.I * p.J);
    unsigned short* out16 = reinterpret_cast<unsigned short*>(p.C);
    auto store_all = [&](auto bf16_c) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int j = t * Cin + ci0 + cihalf * 32 + fcol;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int i = co0 + cohalf * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * fhalf;
                if (decltype(bf16_c)::value) out16[(long long)i * p.J + j] = f32_to_bf16(acc[t][reg]);
                else slab[(long long)i * p.J + j] = acc[t][reg];
            }
        }
    };
    if (direct && p.out_bf16) store_all(std::true_type{}); else store_all(std::false_type{});
}

__global__ __launch_bounds__(W3_THREADS, 1) void wgrad3x3_kernel(GemmTnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles = (p.I >> 6) * (p.Cs >> 6);
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    wgrad3x3_body(p, lid % tiles, lid / tiles, smem);
}

// ======== host part ========
// ---- the ring-staged 3x3 weight-gradient kernel (wgrad3x3_kernel): eligibility and reduction split ----
// EPI_WGRAD3X3=0: never (the TN kernel with the taps as columns)
static bool w3_enabled() {
    static const bool on = [] { const char* e = getenv("EPI_WGRAD3X3"); return !(e && e[0] == '0'); }();
    return on;
}
static bool w3_eligible(const GemmTnArgs& a) {
    return w3_enabled() && a.gather && !a.b_bn && a.KW == 3 && a.J == 9 * a.Cs && a.stride == 1 && a.pad == 1 && a.Hg == a.Hs && a.Wg == a.Ws && a.Cs % 64 == 0 &&
           a.I % 64 == 0 && a.lda == a.I && a.ldb == a.Cs && a.Ws >= 1 && epi::w3_ring_rows(a.Ws) == 512 && a.Hs >= 1 && a.R % (a.Hs * a.Ws) == 0 &&
           (long long)(a.R / (a.Hs * a.Ws)) * (a.Hs + 2) * (a.Ws + 2) < (1LL << 30) && gemm_tile_override() == 0;
}
// slices of the position range: up to ~256 workgroups per launch, at least ~16 K tiles (of 64 positions; R real pixels stand for ~1.1 .. 1.6 R positions) each
// -- a workgroup's fixed costs are its ring fill (64 + 2 halo rows) and its 64 x 576 fp32 result.
// A function of (R, I, Cin) alone, so that epi_gemm_tn_workspace_bytes can size the slabs without the image extents.
static int w3_nsplit(int R, int I, int Cin) {
    const long long tiles = (long long)(I / 64) * (Cin / 64);
    long long ns = (256 + tiles - 1) / tiles;
    const long long cap = std::max(1LL, (long long)R / (64 * 16));
    if (ns > cap) ns = cap;
    return (int)std::max(1LL, std::min(ns, 512LL));
}
static int launch_w3(const GemmTnArgs& a, hipStream_t st) {
    const size_t lds = epi::W3_DY_BYTES + (size_t)epi::w3_ring_rows(a.Ws) * 128;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&epi::wgrad3x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       (int)(epi::W3_DY_BYTES + 512 * 128));
    if (attr != hipSuccess) return EPI_ERR_LAUNCH;
    const long long wgs = (long long)(a.I / 64) * (a.Cs / 64) * a.nsplit;
    if (wgs <= 0 || wgs > 0x7fffffffLL) return EPI_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(epi::wgrad3x3_kernel, dim3((unsigned)wgs), dim3(epi::W3_THREADS), lds, st, a);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}

