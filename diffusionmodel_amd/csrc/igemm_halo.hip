// conv3x3_halo_kernel: the 3x3 stride-1 layers with the input halo resident in LDS (see igemm.hip for the family, igemm_dev.h for
// the shared epilogue).  Its own translation unit: the kernel's 16 instantiations dominate the build time.
#include "igemm_dev.h"

namespace dmk {

// =================================================================================================
// v5: halo-resident 3x3 stride-1 kernel (bf16)
// =================================================================================================
// The gather kernels above re-fetch every input pixel once per tap: a 128x128 tile pulls 32 KiB from L2
// into LDS per k-step, and the L2->LDS fill rate of a CU (~70 GB/s, MI355X_MICROARCH.md "Indexed rows:
// gather into LDS"), not the MFMA, then bounds the 64x64 / 32x32 / 16x16 layers at < 50 % of peak.  Here a
// 512-thread workgroup owns 256 output pixels = TH full image rows (W == TW in {64, 32, 16}) x 128 output
// channels and keeps the INPUT HALO of one 64-channel chunk — (TH+2) x (TW+2) pixels x 128 B — resident
// in LDS: all 9 taps are MFMA'd out of it through shifted fragment addresses, only the weights stream
// (16 KiB per k-step, 3-stage ring).  L2->LDS bytes per k-step drop from 2 x 32 KiB (two 128x128
// workgroups) to ~21.6 KiB for the same MFMA work.
//   * halo image: row hp = hy*HS + hx (HS = TW+8, a multiple of 8), 128 B per row, 16-B slot v stored at
//     v ^ (hp & 7).  As HS % 8 == 0 a tap shift (ky, kx) changes hp & 7 only through kx: the fragment
//     addresses are 3 (kx) x 2 (sub-step) x 4 (pixel group) precomputed VGPRs plus an immediate.
//   * staging: `buffer_load_dwordx4 ... lds` with a per-lane 32-bit offset that never changes (pixel /
//     weight-row offset, or 0x80000000 = out of range -> the DMA writes zeros: image border, n >= N) and a
//     scalar offset per k-step (channel chunk / tap).  No per-step address VALU.
//   * the 9 taps are unrolled: ring stage = tap % 3, the next chunk's halo (<= 54 pieces of 8 px) is
//     fetched one piece per wave per tap during taps 0..6 into the other halo buffer, all waits are
//     compile-time `s_waitcnt vmcnt(N)` + one raw s_barrier per k-step.
//   * 8 waves = 4 (pixel rows of 64) x 2 (64 channels): the wave tile, accumulator layout and epilogue are
//     those of the 128x128 kernels (each half of the tile emits its own 128-row statistics partial).


template <typename T, int TW, bool FLIP>
__global__ __launch_bounds__(512) void conv3x3_halo_kernel(const ConvP p) {
    // TW = 8 (8x8 images): the tile is FOUR whole images laid side by side in the halo, each with its own zero columns
    // ([0 A 0][0 B 0][0 C 0][0 D 0], 10 columns apiece): rows stay multiples of 8 pixels, a wave (64 pixels) is one image
    constexpr int TH = TW == 8 ? 8 : 256 / TW, HS = TW == 8 ? 40 : TW + 8, HR = TH + 2, NP = HR * HS / 8;
    static_assert(NP <= HALO_PIECES, "halo does not fit");
    constexpr bool REUSE = DM_HALO_REUSE && TW == 16 && DM_HALO_DMA_POS != 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [halo 0][halo 1][w 0][w 1][w 2]
    char* const sW = smem + 2 * HALO_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm4 = wave & 3, wn = wave >> 2;
    const int fr = lane & 15, fg = lane >> 4;
    const int nb_n = (p.N + 127) >> 7;
    const int ntiles = gridDim.x / p.splits;
    const int split = blockIdx.x / ntiles;
    const int bid = remap_xcd(blockIdx.x - split * ntiles, ntiles);
    const int mb = bid / nb_n, nb = bid - mb * nb_n;
    const int m0 = mb * 256, n0 = nb * 128;
    const int C = p.C1 + p.C2;
    const int c_lo = split * p.kper;                       // this split's channel chunks [c_lo, nchunks)
    const int nchunks = min((C + 63) >> 6, c_lo + p.kper);    // a lone partial chunk (C < 64, single source) reads zeros past C
    // TW = 64 also serves wider images (Wi a multiple of 64): the tile is then 4 rows x 64 COLUMNS x0 .. x0+63 and its
    // left / right halo columns are real pixels of the neighbouring tile
    const int tcols = TW == 64 ? p.Wi >> 6 : 1;
    const int tiles_img = TW == 8 ? 1 : ((p.Hi * TW) >> 8) * tcols;
    const int b = TW == 8 ? mb * 4 : mb / tiles_img;
    const int trem = mb - b * tiles_img;
    const int y0 = TW == 8 ? 0 : (trem / tcols) * TH, x0 = (trem % tcols) * 64;

    const int pix_img = p.B * p.Hi * p.Wi;
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.N * p.ldw * 2, SRD_FLAGS);

    // ---- per-lane constants (nothing below changes inside the loop) ----
    const int lrow = lane >> 3;
    const int slotb = ((lane & 7) ^ lrow) << 4;          // byte offset of the logical vector this lane fetches
    unsigned hv1[7], hv2[7];                               // halo pieces wave + 8 i: byte offset of the pixel in source 1 / 2
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int q = min(wave + 8 * i, NP - 1);           // surplus pieces re-fetch the last one (same bytes, same place)
        const int hp = q * 8 + lrow;
        const int hy = hp / HS, hx = hp - hy * HS;
        const int img = TW == 8 ? hx / 10 : 0;                     // TW = 8: image of the tile this halo column belongs to
        const int y = y0 + hy - 1, x = TW == 8 ? hx - img * 10 - 1 : x0 + hx - 1;
        const bool ok = (unsigned)y < (unsigned)p.Hi && (unsigned)x < (unsigned)p.Wi && img < 4 && hx < TW + 2 + (TW == 8 ? 30 : 0);
        const int pix = ((b + img) * p.Hi + y) * p.Wi + x;
        const int b2 = p.B2 >= p.B ? b + img : (b + img) % p.B2;             // CFG sampler: the skip tensor of n samples feeds 2n (no division otherwise)
        const int pix2 = (b2 * p.Hi + y) * p.Wi + x;
        hv1[i] = ok && slotb < p.C1 * 2 ? (unsigned)(pix * p.C1 * 2 + slotb) : OOB;
        hv2[i] = ok ? (unsigned)(pix2 * p.C2 * 2 + slotb) : OOB;
    }
    unsigned wv[2];                                        // weight pieces 2 wave + j: 8 rows (n) x 128 B
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + (wave * 2 + j) * 8 + lrow;
        wv[j] = n < p.N && slotb < C * 2 ? (unsigned)(n * p.ldw * 2 + slotb) : OOB;
    }
    int hoff[3][2][4];                                     // pixel-operand fragment addresses in the current halo buffer
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        int ly, col;                                        // halo row / column of this lane's pixel of the 16-pixel group (tap 0,0)
        if constexpr (TW == 8) {                            // group = rows 2 mt, 2 mt + 1 of image wm4
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
    int woff[2][4];                                        // weight-operand fragment addresses within a stage
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) woff[sub][nt] = lds_off(wn * 64 + nt * 16 + fr, sub * 4 + fg);

    auto issue_w = [&](int tap, int chunk, int stage) {    // weights of k-step (chunk, tap) -> ring stage
        const bool live = chunk < nchunks;
        const int soff = (tap * C + (chunk << 6)) * 2;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_dst)(sW + stage * WSTAGE + (wave * 2 + j) * 1024), 16, live ? wv[j] : OOB,
                                                     soff, 0, 0);
    };
    auto issue_w1 = [&](int j, int tap, int chunk, int stage) {   // one of the two weight pieces of issue_w
        const bool live = chunk < nchunks;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_dst)(sW + stage * WSTAGE + (wave * 2 + j) * 1024), 16, live ? wv[j] : OOB,
                                                 (tap * C + (chunk << 6)) * 2, 0, 0);
    };
    auto issue_h = [&](int i, int chunk, int buf) {        // halo piece wave + 8 i of `chunk` -> halo buffer
        const bool live = chunk < nchunks;
        const int c0 = chunk << 6;
        const bool first = c0 < p.C1;
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(first ? p.in1 : p.in2), 0,
                                                                           first ? pix_img * p.C1 * 2 : p.B2 * p.Hi * p.Wi * p.C2 * 2, SRD_FLAGS);
        const unsigned v = first ? hv1[i] : hv2[i];
        const int q = min(wave + 8 * i, NP - 1);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_dst)(smem + buf * HALO_BYTES + q * 1024), 16, live ? v : OOB,
                                                 (first ? c0 : c0 - p.C1) * 2, 0, 0);
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int i = 0; i < 7; ++i) issue_h(i, c_lo, 0);
    issue_w(REUSE ? (FLIP ? 8 : 0) : 0, c_lo, 0);          // the weights of steps 0 and 1 (REUSE: column-major steps, see the loop)
    issue_w(REUSE ? (FLIP ? 5 : 3) : 1, c_lo, 1);
    int hdelta = HALO_BYTES;
    if constexpr (REUSE) {
        // r04: pixel fragments kept in registers across the three taps of a halo COLUMN offset.  The wave's pixels are 4 image rows x 16
        // columns (TW = 16: rows 4 wm4 .. 4 wm4 + 3), so for a fixed column offset hx the taps hy = 0, 1, 2 read halo rows j = hy .. hy + 3
        // of ONE six-row strip: step (hx, 0) reads rows 0..3, (hx, 1) adds row 4, (hx, 2) adds row 5 — 6 + 6 pixel-fragment reads per
        // column offset instead of 24 (weight fragments unchanged: 8 per tap): 108 instead of 144 ds_read_b128 per chunk and wave.
        // Steps walk the taps column-major (hx = s / 3, hy = s % 3); the weight of halo offset (hy, hx) is tap hy * 3 + hx, or its mirror
        // image when FLIP (input gradient).  Same DMA schedule, ring stage = step % 3.  The summation order over the taps differs from
        // the row-major loop: results agree to fp32 rounding, integer data exactly.
        u32x4 fbr[2][6];
        auto wtap = [](int st) { const int hx = st / 3, hy = st % 3; return FLIP ? (2 - hy) * 3 + (2 - hx) : hy * 3 + hx; };
        for (int chunk = c_lo; chunk < nchunks; ++chunk) {
            const int nbuf = (chunk - c_lo + 1) & 1;
#pragma unroll
            for (int st = 0; st < 9; ++st) {
                if (st >= 1 && st <= 7) wait_vmcnt<3>();
                else wait_vmcnt<2>();
                __builtin_amdgcn_s_barrier();
                const int s2 = (st + 2) % 9;
                const int hx = st / 3, hy = st % 3;
                const char* sWs = sW + (st % 3) * WSTAGE;
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
                    u32x4 fa[4];
                    if (hy == 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) fbr[sub][j] = *(const u32x4*)(smem + hoff[hx][sub][0] + (j * HS + hx) * ROWB);
                    } else {
                        fbr[sub][3 + hy] = *(const u32x4*)(smem + hoff[hx][sub][0] + ((3 + hy) * HS + hx) * ROWB);
                    }
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) fa[nt] = *(const u32x4*)(sWs + woff[sub][nt]);
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        {
                            constexpr int G0 = DM_HALO_DMA_POS / 100, G1 = DM_HALO_DMA_POS / 10 % 10, G2 = DM_HALO_DMA_POS % 10;
                            const int g = sub * 4 + nt;
                            const int cw = chunk + (st + 2 >= 9 ? 1 : 0);
                            if (g == G0 || g == G1 || g == G2) __builtin_amdgcn_sched_barrier(0);
                            if (g == G0) issue_w1(0, wtap(s2), cw, s2 % 3);
                            if (g == G1) issue_w1(1, wtap(s2), cw, s2 % 3);
                            if (g == G2 && st < 7) issue_h(st, chunk + 1, nbuf);
                            if (g == G0 || g == G1 || g == G2) __builtin_amdgcn_sched_barrier(0);
                        }
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) Mma<T>::run(fa[nt], fbr[sub][mt + hy], acc[nt][mt]);
                    }
                }
            }
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) hoff[kx][sub][0] += hdelta;
            hdelta = -hdelta;
        }
    } else {
        for (int chunk = c_lo; chunk < nchunks; ++chunk) {
            const int nbuf = (chunk - c_lo + 1) & 1;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                // everything older than the previous step's group (2 weight pieces + its halo piece) has landed
                if (tap >= 1 && tap <= 7) wait_vmcnt<3>();
                else wait_vmcnt<2>();
                __builtin_amdgcn_s_barrier();
                const int t2 = (tap + 2) % 9;
#if DM_HALO_DMA_POS == 0
                issue_w(t2, chunk + (tap + 2 >= 9 ? 1 : 0), t2 % 3);
                if (tap < 7) issue_h(tap, chunk + 1, nbuf);
#endif
                const int ky = FLIP ? 2 - tap / 3 : tap / 3, kx = FLIP ? 2 - tap % 3 : tap % 3;   // halo offset of this tap
                const char* sWs = sW + (tap % 3) * WSTAGE;
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
                    u32x4 fb[4], fa[4];
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) fb[mt] = *(const u32x4*)(smem + hoff[kx][sub][mt] + (ky * HS + kx) * ROWB);
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) fa[nt] = *(const u32x4*)(sWs + woff[sub][nt]);
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
#if DM_HALO_DMA_POS != 0
                        // the k-step's three LDS-DMA pieces go between MFMA groups (4 MFMAs each), away from the fragment reads: an
                        // LDS-DMA piece costs 100-185 issue cycles next to ds_read_b128s and 25-60 among MFMAs (MI355X_MICROARCH.md)
                        {
                            constexpr int G0 = DM_HALO_DMA_POS / 100, G1 = DM_HALO_DMA_POS / 10 % 10, G2 = DM_HALO_DMA_POS % 10;
                            const int g = sub * 4 + nt;
                            const int cw = chunk + (tap + 2 >= 9 ? 1 : 0);
                            if (g == G0 || g == G1 || g == G2) __builtin_amdgcn_sched_barrier(0);
                            if (g == G0) issue_w1(0, t2, cw, t2 % 3);
                            if (g == G1) issue_w1(1, t2, cw, t2 % 3);
                            if (g == G2 && tap < 7) issue_h(tap, chunk + 1, nbuf);
                            if (g == G0 || g == G1 || g == G2) __builtin_amdgcn_sched_barrier(0);
                        }
#endif
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) Mma<T>::run(fa[nt], fb[mt], acc[nt][mt]);
                    }
                }
            }
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) hoff[kx][sub][mt] += hdelta;
            hdelta = -hdelta;
        }
    }
    wait_vmcnt<0>();                     // the surplus (out-of-range) pieces of the last steps
    __syncthreads();                     // the epilogue reuses the LDS for the statistics fold
    const int half = wm4 >> 1;
    if (p.splits > 1) {                  // the two 128-row halves are half-tiles 2 mb, 2 mb + 1 of the partial layout
        if (p.counters == nullptr) {     // two-launch form: splitk_epilogue_kernel folds the partials
            store_partial<128>(p, acc, split, ntiles * 2, (mb * 2 + half) * nb_n + nb, (wm4 & 1) + 2 * wn, lane);
            return;
        }
        if (!splitk_last_arriver(p, acc, smem, split, ntiles * 2, bid, (mb * 2 + half) * nb_n + nb, (wm4 & 1) + 2 * wn, tid, lane)) return;
    }
    const int tidh = (((wave & 1) + 2 * wn) << 6) + lane;
    // first output pixel of this wave's 64: linear in the tile for whole-row tiles, its own image row for column tiles
    const int mw = (TW == 64 && tcols > 1) ? ((b * p.Hi + y0 + wm4) * p.Wi + x0) : m0 + wm4 * 64;
    conv_epilogue<T, 128>(p, acc, smem + half * 4096, tidh, wm4 & 1, wn, fr, fg, mb * 2 + half, mw - (wm4 & 1) * 64, n0);
}

template <typename T, int TW, bool FLIP>
int launch_halo(const ConvP& p, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3x3_halo_kernel<T, TW, FLIP>, hipFuncAttributeMaxDynamicSharedMemorySize, HALO_LDS);
        if (e != hipSuccess) { dm_set_error("hipFuncSetAttribute(%d B LDS) failed: %s", HALO_LDS, hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    const int tiles = (p.M / 256) * cdiv(p.N, 128);
    ConvP q = p;
    // few tiles, deep K (the 8x8 / 16x16 layers): split the channel chunks over workgroups until the chip is full
    const int nchunks = (p.C1 + p.C2 + 63) / 64;
    if (tiles <= 128 && nchunks >= 4 && dm_g_ws != nullptr && p.Wi <= 64) {     // (the split epilogue assumes whole-row tiles)
        int splits = 256 / tiles;
        if (splits > nchunks / 2) splits = nchunks / 2;
        if (splits >= 2 && (int64_t)splits * tiles * 2 * (128 * 128 * 4) <= dm_g_ws_bytes) {
            q.kper = cdiv(nchunks, splits);
            q.splits = cdiv(nchunks, q.kper);
            q.ws = dm_g_ws;
            q.counters = g_splitk_inkernel ? dm_g_counters : nullptr;   // the last split to arrive runs the epilogue in the same launch
        }
    }
    if (TW == 64 && halo4_ok(q)) return launch_halo4_any(q, std::is_same<T, f16>::value, st);      // (experiment, off by default)
    if (q.splits == 1 && g_conv_persist) {          // persistent workgroups with cross-tile prefetch (igemm_halo_p.hip) where they apply
        static int ncu = 0;
        if (ncu == 0) {
            int dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
            if (ncu <= 0) ncu = 256;
        }
        if (halo_persist_ok(q, tiles, ncu)) return launch_halo_persist_any(q, std::is_same<T, f16>::value, tiles, ncu, st);
    }
    hipLaunchKernelGGL((conv3x3_halo_kernel<T, TW, FLIP>), dim3((unsigned)(tiles * q.splits)), dim3(512), HALO_LDS, st, q);
    DM_LAUNCH_CHECK();
    g_last_path = 1;
    if (q.splits > 1 && q.counters == nullptr) {
        return launch_splitk_epilogue128(q, std::is_same<T, f16>::value, (unsigned)(tiles * 2), st);
    }
    return DM_OK;
}

int launch_halo_any(const ConvP& p, bool is_f16, hipStream_t st) {
    const bool flip = p.ty < 0;
#define DM_HALO(T)                                                                                                   \
    do {                                                                                                             \
        if (p.Wi >= 64) return flip ? launch_halo<T, 64, true>(p, st) : launch_halo<T, 64, false>(p, st);            \
        if (p.Wi == 32) return flip ? launch_halo<T, 32, true>(p, st) : launch_halo<T, 32, false>(p, st);            \
        if (p.Wi == 16) return flip ? launch_halo<T, 16, true>(p, st) : launch_halo<T, 16, false>(p, st);            \
        return flip ? launch_halo<T, 8, true>(p, st) : launch_halo<T, 8, false>(p, st);                              \
    } while (0)
    if (is_f16) DM_HALO(f16);
    DM_HALO(bf16);
#undef DM_HALO
}

}  // namespace dmk
