// conv3x3_halo4_kernel: the halo-resident 3x3 kernel of igemm_halo.hip with ONE wave per SIMD (r04 experiment).
//
// Same tile (256 output pixels = 4 image rows x 64 columns, x 128 output channels), same LDS image (two 54-KiB halo buffers, a 3-stage
// ring of 16-KiB weight stages), same DMA staging and the same order of the fp32 sums (chunk-major, tap by tap, two k = 32 sub-steps per
// tap) — results are bit-identical to conv3x3_halo_kernel<T, 64, FLIP>.  What differs: FOUR waves, each with a 128-pixel x 64-channel
// wave tile (two image rows x four 16-pixel groups, four 16-channel blocks: 8 x 4 MFMA tiles, 128 accumulator registers), so that a
// k = 32 sub-step is 12 fragment reads (8 pixel + 4 weight) for 32 MFMAs instead of 2 x 8 for 2 x 16: a quarter fewer LDS read bytes,
// and no second wave competing for a SIMD's matrix pipe and VALU issue.  The price: no partner wave covers a wait, so every latency the
// loop exposes is paid in full — the schedule has to hide it itself (SCHED below).
//   SCHED = 0: the compiler's schedule.
//   SCHED = 1: fragments double-buffered one sub-step ahead (the reads of sub-step s + 1 interleaved one per two MFMAs of sub-step s,
//              pinned with sched_group_barrier); the first sub-step of a tap is read right after the tap's barrier.
//   SCHED = 2: (-DDM_HALO4_XTAP=1 only, RESULTS WRONG) sub-step 0 of tap t + 1 read during sub-step 1 of tap t: the upper bound of what a
//              legal cross-tap prefetch (six 8-KiB weight half-stages instead of three 16-KiB stages) could gain.
// Only the geometry the 64x64 layers need: rows of >= 64 pixels in whole 64-column tiles, one source, no split over channel chunks.
// Measured (profiles/r04_ab_same_box.txt; 64x64 128 -> 128 forward, B = 64, one box, three alternations; PMC in the same call):
//   persistent eight-wave 76.5 - 80.6 us (MFMA pipe busy 0.54) | per-tile eight-wave 84.2 - 85.5 us (0.475) | four waves, compiler schedule
//   85.8 - 87.8 us (0.43) | + pinned prefetch of the second sub-step 87.4 - 88.6 us (0.42) | cross-tap upper bound 84.8 - 86.4 us (0.45).
// One wave per SIMD is 2 - 4 % SLOWER than two and even the illegal upper bound does not reach the eight-wave kernel: a quarter fewer
// fragment reads do not pay for losing the partner wave that covers the barrier, the DMA issue and the read latency.  Off by default.
#include "igemm_dev.h"
#ifndef DM_HALO4_XTAP
#define DM_HALO4_XTAP 0
#endif

namespace dmk {

int g_conv_wave4 = 0;          // 0: off; 1: SCHED 0; 2: SCHED 1; 3: SCHED 2 = timing experiment with wrong results (dm_set_conv_wave4 / DM_CONV_WAVE4)

template <typename T, bool FLIP, int SCHED>
__global__ __launch_bounds__(256) void conv3x3_halo4_kernel(const ConvP p) {
    constexpr int TH = 4, HS = 72, NP = 54;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [halo 0][halo 1][w 0][w 1][w 2]
    char* const sW = smem + 2 * HALO_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm2 = wave & 1, wn = wave >> 1;
    const int fr = lane & 15, fg = lane >> 4;
    const int nb_n = (p.N + 127) >> 7;
    const int ntiles = gridDim.x;
    const int bid = remap_xcd(blockIdx.x, ntiles);
    const int mb = bid / nb_n, nb = bid - mb * nb_n;
    const int m0 = mb * 256, n0 = nb * 128;
    const int C = p.C1 + p.C2;
    const int nchunks = (C + 63) >> 6;
    const int tcols = p.Wi >> 6;
    const int tiles_img = ((p.Hi * 64) >> 8) * tcols;
    const int b = mb / tiles_img;
    const int trem = mb - b * tiles_img;
    const int y0 = (trem / tcols) * TH, x0 = (trem % tcols) * 64;

    const int pix_img = p.B * p.Hi * p.Wi;
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.N * p.ldw * 2, SRD_FLAGS);
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)p.in1, 0, pix_img * p.C1 * 2, SRD_FLAGS);

    // ---- per-lane constants
    const int lrow = lane >> 3;
    const int slotb = ((lane & 7) ^ lrow) << 4;
    unsigned hv1[14];                                      // halo pieces wave + 4 i (one source: the second set of 14 offsets spills into the loop)
#pragma unroll
    for (int i = 0; i < 14; ++i) {
        const int q = min(wave + 4 * i, NP - 1);           // surplus pieces re-fetch the last one (same bytes, same place)
        const int hp = q * 8 + lrow;
        const int hy = hp / HS, hx = hp - hy * HS;
        const int y = y0 + hy - 1, x = x0 + hx - 1;
        const bool ok = (unsigned)y < (unsigned)p.Hi && (unsigned)x < (unsigned)p.Wi && hx < 66;
        const int pix = (b * p.Hi + y) * p.Wi + x;
        hv1[i] = ok && slotb < p.C1 * 2 ? (unsigned)(pix * p.C1 * 2 + slotb) : OOB;
    }
    unsigned wv[4];                                        // weight pieces 4 wave + j: 8 rows (n) x 128 B
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + (wave * 4 + j) * 8 + lrow;
        wv[j] = n < p.N && slotb < C * 2 ? (unsigned)(n * p.ldw * 2 + slotb) : OOB;
    }
    // pixel fragment (mt, tap): halo row 2 wm2 + (mt >> 2) + ky, column 16 (mt & 3) + fr + kx; the swizzle follows (fr + kx) & 7 only
    int hoff[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) hoff[kx][sub] = ((2 * wm2) * HS + fr) * ROWB + (((sub * 4 + fg) ^ ((fr + kx) & 7)) << 4);
    int woff[2][4];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) woff[sub][nt] = lds_off(wn * 64 + nt * 16 + fr, sub * 4 + fg);

    auto issue_w1 = [&](int j, int tap, int chunk, int stage) {
        const bool live = chunk < nchunks;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_dst)(sW + stage * WSTAGE + (wave * 4 + j) * 1024), 16, live ? wv[j] : OOB,
                                                 (tap * C + (chunk << 6)) * 2, 0, 0);
    };
    auto issue_h = [&](int i, int chunk, int buf) {
        const bool live = chunk < nchunks;
        const int q = min(wave + 4 * i, NP - 1);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lds_dst)(smem + buf * HALO_BYTES + q * 1024), 16, live ? hv1[i] : OOB, (chunk << 6) * 2, 0, 0);
    };

    f32x4 accA[4][4], accB[4][4];                          // [nt][mt]: image row 2 wm2 / 2 wm2 + 1 of the tile
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) accA[i][j] = accB[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int i = 0; i < 14; ++i) issue_h(i, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) issue_w1(j, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) issue_w1(j, 1, 0, 1);
    int hdelta = HALO_BYTES;
    u32x4 pfa[4], pfb[8];                                  // SCHED 2 (timing experiment): the NEXT tap's first sub-step, read one tap early
#pragma unroll
    for (int i = 0; i < 4; ++i) pfa[i] = (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 8; ++i) pfb[i] = (u32x4){0u, 0u, 0u, 0u};
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int nbuf = (chunk + 1) & 1;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // everything older than the previous tap's group (4 weight pieces + its 2 halo pieces) has landed
            if (tap >= 1 && tap <= 7) wait_vmcnt<6>();
            else wait_vmcnt<4>();
            __builtin_amdgcn_s_barrier();
            const int t2 = (tap + 2) % 9;
            const int cw = chunk + (tap + 2 >= 9 ? 1 : 0);
            const int ky = FLIP ? 2 - tap / 3 : tap / 3, kx = FLIP ? 2 - tap % 3 : tap % 3;
            const char* sWs = sW + (tap % 3) * WSTAGE;
            auto rd_b = [&](int sub, int mt) {
                return *(const u32x4*)(smem + hoff[kx][sub] + (((mt >> 2) + ky) * HS + (mt & 3) * 16 + kx) * ROWB);
            };
            auto rd_a = [&](int sub, int nt) { return *(const u32x4*)(sWs + woff[sub][nt]); };
            auto dma = [&](int g) {                         // the tap's six LDS-DMA pieces, between MFMA groups 3 .. 6 of its eight
                if (g == 3) issue_w1(0, t2, cw, t2 % 3);
                if (g == 4) issue_w1(1, t2, cw, t2 % 3);
                if (g == 5) { issue_w1(2, t2, cw, t2 % 3); if (tap < 7) issue_h(2 * tap, chunk + 1, nbuf); }
                if (g == 6) { issue_w1(3, t2, cw, t2 % 3); if (tap < 7) issue_h(2 * tap + 1, chunk + 1, nbuf); }
            };
            if constexpr (SCHED == 0) {
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
                    u32x4 fb[8], fa[4];
#pragma unroll
                    for (int mt = 0; mt < 8; ++mt) fb[mt] = rd_b(sub, mt);
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) fa[nt] = rd_a(sub, nt);
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const int g = sub * 4 + nt;
                        if (g >= 3 && g <= 6) {
                            __builtin_amdgcn_sched_barrier(0);
                            dma(g);
                            __builtin_amdgcn_sched_barrier(0);
                        }
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) Mma<T>::run(fa[nt], fb[mt], accA[nt][mt]);
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) Mma<T>::run(fa[nt], fb[4 + mt], accB[nt][mt]);
                    }
                }
            } else if constexpr (SCHED == 2) {
                // UPPER-BOUND TIMING EXPERIMENT, RESULTS WRONG: sub-step 0 of tap t + 1 is read during sub-step 1 of tap t — one barrier
                // early, i.e. before the other waves' pieces of that weight stage are known to have landed (a ring of 3 x 16 KiB cannot
                // guarantee them; six 8-KiB half-stages could).  What this measures: what a legal cross-tap prefetch could gain at most.
                const int tn = (tap + 1) % 9;
                const int kyn = FLIP ? 2 - tn / 3 : tn / 3, kxn = FLIP ? 2 - tn % 3 : tn % 3;
                const char* sWn = sW + (tn % 3) * WSTAGE;
                const int hnext = tap == 8 ? hdelta : 0;                 // tap 8 prefetches from the next chunk's halo buffer
                auto rd_bn = [&](int mt) { return *(const u32x4*)(smem + hoff[kxn][0] + hnext + (((mt >> 2) + kyn) * HS + (mt & 3) * 16 + kxn) * ROWB); };
                auto rd_an = [&](int nt) { return *(const u32x4*)(sWn + woff[0][nt]); };
                u32x4 fb1[8], fa1[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    if (nt == 3) {
                        __builtin_amdgcn_sched_barrier(0);
                        dma(3);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (nt == 0) { fa1[0] = rd_a(1, 0); fa1[1] = rd_a(1, 1); fb1[0] = rd_b(1, 0); }
                    if (nt == 1) { fb1[1] = rd_b(1, 1); fb1[2] = rd_b(1, 2); fb1[3] = rd_b(1, 3); }
                    if (nt == 2) { fa1[2] = rd_a(1, 2); fa1[3] = rd_a(1, 3); fb1[4] = rd_b(1, 4); }
                    if (nt == 3) { fb1[5] = rd_b(1, 5); fb1[6] = rd_b(1, 6); fb1[7] = rd_b(1, 7); }
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) Mma<T>::run(pfa[nt], pfb[mt], accA[nt][mt]);
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) Mma<T>::run(pfa[nt], pfb[4 + mt], accB[nt][mt]);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                u32x4 nfa[4], nfb[8];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int g = 4 + nt;
                    if (g <= 6) {
                        __builtin_amdgcn_sched_barrier(0);
                        dma(g);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (nt == 0) { nfa[0] = rd_an(0); nfa[1] = rd_an(1); nfb[0] = rd_bn(0); }
                    if (nt == 1) { nfb[1] = rd_bn(1); nfb[2] = rd_bn(2); nfb[3] = rd_bn(3); }
                    if (nt == 2) { nfa[2] = rd_an(2); nfa[3] = rd_an(3); nfb[4] = rd_bn(4); }
                    if (nt == 3) { nfb[5] = rd_bn(5); nfb[6] = rd_bn(6); nfb[7] = rd_bn(7); }
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) Mma<T>::run(fa1[nt], fb1[mt], accA[nt][mt]);
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) Mma<T>::run(fa1[nt], fb1[4 + mt], accB[nt][mt]);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) pfa[i] = nfa[i];
#pragma unroll
                for (int i = 0; i < 8; ++i) pfb[i] = nfb[i];
            } else {
                // sub-step 0: read right behind the barrier; sub-step 1: its 12 fragments are read one per two MFMAs of sub-step 0
                u32x4 fb0[8], fa0[4], fb1[8], fa1[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) fa0[nt] = rd_a(0, nt);
#pragma unroll
                for (int mt = 0; mt < 8; ++mt) fb0[mt] = rd_b(0, mt);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    if (nt == 3) {
                        __builtin_amdgcn_sched_barrier(0);
                        dma(3);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // three reads of the next sub-step per MFMA group of eight
                    if (nt == 0) { fa1[0] = rd_a(1, 0); fa1[1] = rd_a(1, 1); fb1[0] = rd_b(1, 0); }
                    if (nt == 1) { fb1[1] = rd_b(1, 1); fb1[2] = rd_b(1, 2); fb1[3] = rd_b(1, 3); }
                    if (nt == 2) { fa1[2] = rd_a(1, 2); fa1[3] = rd_a(1, 3); fb1[4] = rd_b(1, 4); }
                    if (nt == 3) { fb1[5] = rd_b(1, 5); fb1[6] = rd_b(1, 6); fb1[7] = rd_b(1, 7); }
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) Mma<T>::run(fa0[nt], fb0[mt], accA[nt][mt]);
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) Mma<T>::run(fa0[nt], fb0[4 + mt], accB[nt][mt]);
                    // 8 MFMA, 3 DS reads spread one per two to three MFMAs
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int g = 4 + nt;
                    if (g <= 6) {
                        __builtin_amdgcn_sched_barrier(0);
                        dma(g);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) Mma<T>::run(fa1[nt], fb1[mt], accA[nt][mt]);
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) Mma<T>::run(fa1[nt], fb1[4 + mt], accB[nt][mt]);
                }
            }
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) hoff[kx][sub] += hdelta;
        hdelta = -hdelta;
    }
    wait_vmcnt<0>();                     // the surplus (out-of-range) pieces of the last steps
    __syncthreads();                     // the epilogue reuses the LDS for the statistics fold
    // the two image rows of this wave are what waves (wm4 = 2 wm2, wn) and (2 wm2 + 1, wn) of the eight-wave kernel hold: the shared
    // epilogue once per row — the first call leaves its column sums in LDS, the second publishes both (one statistics partial per 128 rows)
    const int tidp = (wn << 6) + lane;
    const int mwA = tcols > 1 ? ((b * p.Hi + y0 + 2 * wm2) * p.Wi + x0) : m0 + (2 * wm2) * 64;
    const int mwB = tcols > 1 ? ((b * p.Hi + y0 + 2 * wm2 + 1) * p.Wi + x0) : m0 + (2 * wm2 + 1) * 64;
    conv_epilogue<T, 128, 1>(p, accA, smem + wm2 * 4096, tidp, 0, wn, fr, fg, mb * 2 + wm2, mwA, n0);
    conv_epilogue<T, 128, 3>(p, accB, smem + wm2 * 4096, tidp, 1, wn, fr, fg, mb * 2 + wm2, mwB - 64, n0);
}

bool halo4_ok(const ConvP& p) {
    if (!g_conv_wave4 || p.Wi < 64 || (p.Wi & 63) || (p.Hi & 3) || p.splits != 1) return false;
    return p.C2 == 0 && (p.C1 % 64 == 0 || p.C1 < 64);            // one source
}

template <typename T, bool FLIP, int SCHED>
static int launch_halo4(const ConvP& p, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3x3_halo4_kernel<T, FLIP, SCHED>, hipFuncAttributeMaxDynamicSharedMemorySize, HALO_LDS);
        if (e != hipSuccess) { dm_set_error("hipFuncSetAttribute(%d B LDS) failed: %s", HALO_LDS, hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    const int tiles = (p.M / 256) * cdiv(p.N, 128);
    hipLaunchKernelGGL((conv3x3_halo4_kernel<T, FLIP, SCHED>), dim3((unsigned)tiles), dim3(256), HALO_LDS, st, p);
    DM_LAUNCH_CHECK();
    g_last_path = 1;
    return DM_OK;
}

int launch_halo4_any(const ConvP& p, bool is_f16, hipStream_t st) {
    const bool flip = p.ty < 0;
#define DM_H4(T, S) (flip ? launch_halo4<T, true, S>(p, st) : launch_halo4<T, false, S>(p, st))
#if DM_HALO4_XTAP        // diagnostic builds only (-DDM_HALO4_XTAP=1): mode 3 = the cross-tap prefetch TIMING experiment, results wrong
    if (g_conv_wave4 == 3) return is_f16 ? DM_H4(f16, 2) : DM_H4(bf16, 2);
#endif
    if (g_conv_wave4 == 2) return is_f16 ? DM_H4(f16, 1) : DM_H4(bf16, 1);
    return is_f16 ? DM_H4(f16, 0) : DM_H4(bf16, 0);
#undef DM_H4
}

}  // namespace dmk

extern "C" int dm_set_conv_wave4(int mode) {
    DM_CHECK_ARG(mode >= 0 && mode <= (DM_HALO4_XTAP ? 3 : 2), "dm_set_conv_wave4: 0 (off), 1 (four waves, compiler schedule) or 2 (pinned prefetch of the second sub-step)");
    dmk::g_conv_wave4 = mode;
    return DM_OK;
}
