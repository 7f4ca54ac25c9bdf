// conv_tap4_halo_kernel: the 4x4 / stride-2 convolution and its input gradient on the halo-resident form (see igemm.hip for the
// family, igemm_dev.h for the shared epilogue).  Its own translation unit for the build time.
#include "igemm_dev.h"

namespace dmk {

// =================================================================================================
// v6: halo-resident kernel for FOUR-tap layers: the 4x4 stride-2 convolution and its input gradient
// =================================================================================================
// A 4x4 / stride 2 / pad 1 convolution (new_scripy.py:229, UnetDown.down[4]) re-reads every input pixel for 4 of its 16 taps; on the
// gather kernel that is 4x the L2->LDS fill of the 3x3 layers per MFMA and the layers ran at 340-490 TFLOP/s.  Split the input into
// its four pixel-parity sub-images V_g(y, x) = X(2y + py, 2x + px): with (ky - 1) = 2a + py every tap reads ONE sub-image at offset
// a in {-1, 0, +1} — py = 0: (ky, a) = (1, 0), (3, +1);  py = 1: (0, -1), (2, 0) — so the layer is a sum over (sub-image, 64-channel
// chunk) of FOUR-tap contributions out of that sub-image's halo, and the halo can stay resident exactly as in the 3x3 kernel (same
// tile geometry over the OUTPUT image, same LDS image, same fragment addresses).  The sub-images are never materialised: the halo
// DMA's per-lane pixel offsets simply step by two pixels, the parity is a scalar offset per chunk.  S2 = true is that forward form
// (weights straight from the [N][16][C] pack).  S2 = false is the input gradient of one output-parity class: a plain 2x2-tap
// convolution over dy with tap offsets from (ty, tx, oy0, ox0) and a strided output (osy = osx = 2; the epilogue maps it), weights
// from the per-class transposed pack [c][4][n].
//   * 4 k-steps per chunk on a 3-stage weight ring: the stage of a step is (4 * chunk + j) % 3, not a compile-time constant as with
//     9 taps.  It is a scalar: the DMA destination takes it as such, the fragment reads add it to their 8 addresses (8 VALU per
//     32 MFMAs; three rotating address sets instead cost 16 more VGPRs and pushed the kernel into scratch).
//   * the tap offset (dy, dx) of a step is a scalar too: a 9-way switch picks the tap body with the offset as an immediate.
//   * next chunk's halo (<= 54 pieces, 7 per wave) is fetched 3 + 2 + 2 pieces during steps 0..2; waits are compile-time vmcnt.
template <typename T, int TW, bool S2>
__global__ __launch_bounds__(512) void conv_tap4_halo_kernel(const ConvP pp) {
    constexpr int TH = TW == 8 ? 8 : 256 / TW, HS = TW == 8 ? 40 : TW + 8, HR = TH + 2, NP = HR * HS / 8;
    static_assert(NP <= HALO_PIECES, "halo does not fit");
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [halo 0][halo 1][w 0][w 1][w 2]
    char* const sW = smem + 2 * HALO_BYTES;

    // parity class of this workgroup (S2 = false with npar = 4): its weight pack, tap offsets and output offsets replace the descriptor's
    ConvP p = pp;
    const int gpar = (int)gridDim.x / pp.npar;             // workgroups per parity class
    const int par = (int)blockIdx.x / gpar;
    const int blk = (int)blockIdx.x - par * gpar;
    if (!S2 && pp.npar > 1) {
        p.w = pp.w4[par];
        p.oy0 = pp.oy4[par]; p.ox0 = pp.ox4[par];
        p.ooy = pp.ooy4[par]; p.oox = pp.oox4[par];
    }

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm4 = wave & 3, wn = wave >> 2;
    const int fr = lane & 15, fg = lane >> 4;
    const int nb_n = (p.N + 127) >> 7;
    const int ntiles = gpar / p.splits;
    const int split = blk / ntiles;
    const int bid = remap_xcd(blk - split * ntiles, ntiles);
    const int mb = bid / nb_n, nb = bid - mb * nb_n;
    const int m0 = mb * 256, n0 = nb * 128;
    const int CK = p.C1;                                   // reduction channels per tap (single source)
    const int cpg = CK >> 6;                               // 64-channel chunks per sub-image
    const int nch_total = S2 ? 4 * cpg : cpg;
    const int c_lo = split * p.kper;
    const int nchunks = min(nch_total, c_lo + p.kper);
    const int tiles_img = TW == 8 ? 1 : (p.Hq * TW) >> 8;
    const int b = TW == 8 ? mb * 4 : mb / tiles_img;
    const int y0 = TW == 8 ? 0 : (mb - b * tiles_img) * TH;

    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.N * p.ldw * 2, SRD_FLAGS);
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)p.in1, 0, p.B * p.Hi * p.Wi * CK * 2, SRD_FLAGS);

    const int lrow = lane >> 3;
    const int slotb = ((lane & 7) ^ lrow) << 4;
    unsigned hv[7];                                        // halo pieces wave + 8 i: byte offset of the pixel (sub-image 0,0)
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int q = min(wave + 8 * i, NP - 1);
        const int hp = q * 8 + lrow;
        const int hy = hp / HS, hx = hp - hy * HS;
        const int img = TW == 8 ? hx / 10 : 0;
        const int y = y0 + hy - 1, x = TW == 8 ? hx - img * 10 - 1 : hx - 1;        // position in the output-sized (sub-)image
        const bool ok = (unsigned)y < (unsigned)p.Hq && (unsigned)x < (unsigned)p.Wq && img < 4 && hx < TW + 2 + (TW == 8 ? 30 : 0);
        const int pix = S2 ? ((b + img) * p.Hi + 2 * y) * p.Wi + 2 * x : ((b + img) * p.Hi + y) * p.Wi + x;
        hv[i] = ok ? (unsigned)(pix * CK * 2 + slotb) : OOB;
    }
    unsigned wv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + (wave * 2 + j) * 8 + lrow;
        wv[j] = n < p.N ? (unsigned)(n * p.ldw * 2 + slotb) : OOB;
    }
    int hoff[3][2][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        int ly, col;
        if constexpr (TW == 8) {
            ly = mt * 2 + (fr >> 3);
            col = wm4 * 10 + (fr & 7);
        } else {
            const int g = wm4 * 4 + mt;
            ly = g / (TW / 16);
            col = (g - ly * (TW / 16)) * 16 + fr;
        }
        const int base = (ly * HS + col) * ROWB;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) hoff[kx][sub][mt] = base + (((sub * 4 + fg) ^ ((col + kx) & 7)) << 4);
    }
    int woff[2][4];                                        // weight fragment addresses within stage 0; the stage offset of a step is a scalar
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) woff[sub][nt] = 2 * HALO_BYTES + lds_off(wn * 64 + nt * 16 + fr, sub * 4 + fg);

    // (dy, dx) in {-1, 0, 1}^2 and the weight-row offset of k-step (chunk, j)
    auto tap_of = [&](int chunk, int j, int& dy, int& dx, int& soff) {
        if constexpr (S2) {
            const int g = chunk / cpg, cc = chunk - g * cpg;
            const int py = g >> 1, px = g & 1;
            dy = (j >> 1) - py;
            dx = (j & 1) - px;
            const int ky = 2 * dy + py + 1, kx = 2 * dx + px + 1;
            soff = ((ky * 4 + kx) * CK + (cc << 6)) * 2;
        } else {
            dy = (j >> 1) * p.ty + p.oy0;
            dx = (j & 1) * p.tx + p.ox0;
            soff = (j * CK + (chunk << 6)) * 2;
        }
    };
    auto issue_w = [&](int chunk, int j, int stage) {
        const bool live = chunk < nchunks;
        int dy, dx, soff;
        tap_of(live ? chunk : c_lo, j, dy, dx, soff);
#pragma unroll
        for (int k = 0; k < 2; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_dst)(sW + stage * WSTAGE + (wave * 2 + k) * 1024), 16, live ? wv[k] : OOB, soff, 0, 0);
    };
    auto issue_h = [&](int i, int chunk, int buf) {
        const bool live = chunk < nchunks;
        int soff;
        if constexpr (S2) {
            const int ch = live ? chunk : c_lo;
            const int g = ch / cpg, cc = ch - g * cpg;
            soff = (((g >> 1) * p.Wi + (g & 1)) * CK + (cc << 6)) * 2;
        } else {
            soff = (chunk << 6) * 2;
        }
        const int q = min(wave + 8 * i, NP - 1);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lds_dst)(smem + buf * HALO_BYTES + q * 1024), 16, live ? hv[i] : OOB, soff, 0, 0);
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int i = 0; i < 7; ++i) issue_h(i, c_lo, 0);
    issue_w(c_lo, 0, 0);
    issue_w(c_lo, 1, 1);
    int hdelta = HALO_BYTES;
    int s0 = 0;                                            // ring stage of step 0 of the current chunk
    for (int chunk = c_lo; chunk < nchunks; ++chunk) {
        const int nbuf = (chunk - c_lo + 1) & 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // everything older than the previous step's group has landed (groups: step 0 = 2 w + 3 h, steps 1, 2 = 2 w + 2 h, step 3 = 2 w)
            if (j == 0) wait_vmcnt<2>();
            else if (j == 1) wait_vmcnt<5>();
            else wait_vmcnt<4>();
            __builtin_amdgcn_s_barrier();
            {
                const int j2 = (j + 2) & 3;
                int st = s0 + j + 2;
                st -= st >= 3 ? 3 : 0;
                st -= st >= 3 ? 3 : 0;
                issue_w(chunk + (j + 2 >= 4 ? 1 : 0), j2, st);
            }
            if (j == 0) { issue_h(0, chunk + 1, nbuf); issue_h(1, chunk + 1, nbuf); issue_h(2, chunk + 1, nbuf); }
            else if (j == 1) { issue_h(3, chunk + 1, nbuf); issue_h(4, chunk + 1, nbuf); }
            else if (j == 2) { issue_h(5, chunk + 1, nbuf); issue_h(6, chunk + 1, nbuf); }
            int dy, dx, soff_unused;
            tap_of(chunk, j, dy, dx, soff_unused);
            const int code = (dy + 1) * 3 + (dx + 1);
            int wst = s0 + j;                                  // ring stage of this step (scalar)
            wst -= wst >= 3 ? 3 : 0;
            wst -= wst >= 3 ? 3 : 0;
            wst *= WSTAGE;
            auto body = [&](auto KY, auto KX) {
                constexpr int ky = decltype(KY)::value, kx = decltype(KX)::value;
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
                    u32x4 fb[4], fa[4];
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) fb[mt] = *(const u32x4*)(smem + hoff[kx][sub][mt] + (ky * HS + kx) * ROWB);
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) fa[nt] = *(const u32x4*)(smem + woff[sub][nt] + wst);
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) Mma<T>::run(fa[nt], fb[mt], acc[nt][mt]);
                }
            };
            using I0 = std::integral_constant<int, 0>;
            using I1 = std::integral_constant<int, 1>;
            using I2 = std::integral_constant<int, 2>;
            switch (code) {
                case 0: body(I0{}, I0{}); break;
                case 1: body(I0{}, I1{}); break;
                case 2: body(I0{}, I2{}); break;
                case 3: body(I1{}, I0{}); break;
                case 4: body(I1{}, I1{}); break;
                case 5: body(I1{}, I2{}); break;
                case 6: body(I2{}, I0{}); break;
                case 7: body(I2{}, I1{}); break;
                default: body(I2{}, I2{}); break;
            }
        }
        // 4 steps = one turn of the 3-stage ring plus one
        s0 = s0 == 2 ? 0 : s0 + 1;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) hoff[kx][sub][mt] += hdelta;
        hdelta = -hdelta;
    }
    wait_vmcnt<0>();
    __syncthreads();
    const int half = wm4 >> 1;
    if (p.splits > 1) {                  // the two 128-row halves are half-tiles 2 mb, 2 mb + 1 of the partial layout
        // (never with npar > 1: the launcher takes all four classes in one launch only when that fills the chip without a split)
        if (p.counters == nullptr) {     // two-launch form: splitk_epilogue_kernel folds the partials
            store_partial<128>(p, acc, split, ntiles * 2, (mb * 2 + half) * nb_n + nb, (wm4 & 1) + 2 * wn, lane);
            return;
        }
        if (!splitk_last_arriver(p, acc, smem, split, ntiles * 2, bid, (mb * 2 + half) * nb_n + nb, (wm4 & 1) + 2 * wn, tid, lane)) return;
    }
    const int tidh = (((wave & 1) + 2 * wn) << 6) + lane;
    conv_epilogue<T, 128>(p, acc, smem + half * 4096, tidh, wm4 & 1, wn, fr, fg, mb * 2 + half, m0 + wm4 * 64 - (wm4 & 1) * 64, n0);
}

// halo kernels with split channel chunks: 0 (default) = separate splitk_epilogue_kernel launch, 1 = the last split to arrive finishes the
// tile in the same launch.  The in-kernel form is bit-exact and 2.0 ms per train step SLOWER (17.6 vs 15.5 ms, same box, r02): the
// agent-scope release / acquire fences it needs write back and invalidate the XCD's whole L2 (the eight L2s are not coherent with
// each other), once per workgroup — far more than the 34 epilogue launches of ~13 us it removes.  The same holds for an in-kernel
// reduction of the weight-gradient splits; cross-workgroup hand-offs stay on kernel boundaries.
template <typename T, int TW, bool S2>
int launch_tap4(const ConvP& p, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_tap4_halo_kernel<T, TW, S2>, hipFuncAttributeMaxDynamicSharedMemorySize, HALO_LDS);
        if (e != hipSuccess) { dm_set_error("hipFuncSetAttribute(%d B LDS) failed: %s", HALO_LDS, hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    const int tiles = (p.M / 256) * cdiv(p.N, 128) * p.npar;
    ConvP q = p;
    const int nchunks = (S2 ? 4 : 1) * (p.C1 / 64);
    if (p.npar == 1 && tiles <= 128 && nchunks >= 4 && dm_g_ws != nullptr) {       // few tiles, deep K: split the chunks over workgroups
        int splits = 256 / tiles;
        if (splits > nchunks / 2) splits = nchunks / 2;
        if (splits >= 2 && (int64_t)splits * tiles * 2 * (128 * 128 * 4) <= dm_g_ws_bytes) {
            q.kper = cdiv(nchunks, splits);
            q.splits = cdiv(nchunks, q.kper);
            q.ws = dm_g_ws;
            q.counters = g_splitk_inkernel ? dm_g_counters : nullptr;
        }
    }
    hipLaunchKernelGGL((conv_tap4_halo_kernel<T, TW, S2>), dim3((unsigned)(tiles * q.splits)), dim3(512), HALO_LDS, st, q);
    DM_LAUNCH_CHECK();
    g_last_path = 2;
    if (q.splits > 1 && q.counters == nullptr) {
        return launch_splitk_epilogue128(q, std::is_same<T, f16>::value, (unsigned)(tiles * 2), st);
    }
    return DM_OK;
}

template <typename T, bool S2>
int launch_tap4_tw(const ConvP& p, hipStream_t st) {
    if (p.Wq == 64) return launch_tap4<T, 64, S2>(p, st);
    if (p.Wq == 32) return launch_tap4<T, 32, S2>(p, st);
    if (p.Wq == 16) return launch_tap4<T, 16, S2>(p, st);
    return launch_tap4<T, 8, S2>(p, st);
}

int launch_tap4_any(const ConvP& p, bool is_f16, bool s2, hipStream_t st) {
    if (is_f16) return s2 ? launch_tap4_tw<f16, true>(p, st) : launch_tap4_tw<f16, false>(p, st);
    return s2 ? launch_tap4_tw<bf16, true>(p, st) : launch_tap4_tw<bf16, false>(p, st);
}

}  // namespace dmk
