// conv3x3_halo_pkernel: the halo-resident 3x3 kernel of igemm_halo.hip as PERSISTENT workgroups (one per CU) that walk over
// output tiles, with the NEXT tile's first halo chunk and first two weight stages fetched during the LAST chunk of the current
// tile and awaited with compile-time `s_waitcnt vmcnt(N)` values that leave the epilogue's stores outstanding.
//
// Why: on the 64x64 and 32x32 layers a tile is 18 .. 36 k-steps (~1 us each) plus ~5 us that no MFMA
// covers — workgroup start, the first halo fetch (54 KiB from L2 / HBM), the epilogue.  r01 / r02 built two persistent forms and
// both lost: (1) prefetch under the epilogue, but `vmcnt` retires in order, so the next tile's first wait sat behind the
// epilogue's 64 KiB of stores; (2) the tiles' chunks as one stream — the per-tap choice between "next chunk of this tile" and
// "first chunk of the next tile" put two tile descriptors and selects into the k-loop (134 SGPRs spilled).  This form:
//   * the LAST chunk of a tile is peeled: its nine taps fetch the next tile's halo (taps 0..6) and weights of taps 0 / 1
//     (taps 7 / 8) from precomputed per-lane offsets of the next tile — straight-line code, no selects in the generic loop;
//   * no `vmcnt(0)` before the epilogue.  The epilogue of a whole-tile 16-bit launch issues exactly 8 stores per wave AFTER the
//     prefetch, so the next tile's tap 0 / tap 1 wait with vmcnt(2 + 8) / vmcnt(3 + 8): in-order retirement then guarantees that
//     the prefetched halo and weights have landed while the stores may still be in flight;
//   * the statistics fold of the epilogue lives in ring stage 2 (free after the last tap) instead of halo buffer 0 (a prefetch
//     target now).
// Launched only where it can help and where the 8-store epilogue is guaranteed (persist_ok): more tiles than CUs, no split-K,
// N % 128 == 0, 16-byte-aligned NHWC output.  Everything else stays on conv3x3_halo_kernel.
#include "igemm_dev.h"

namespace dmk {

// The descriptor as the kernel sees it in its kernarg segment, behind a pointer the optimiser cannot see through: fields read
// through it are loaded (s_load) where they are used instead of being hoisted out of the tile loop — hoisted, the ~60 scalars of
// the descriptor plus the per-tap offsets do not fit the 102 SGPRs, uniform values end up in VGPRs and every LDS-DMA issue turns
// into a readfirstlane "waterfall" loop (what sank r02's persistent form).
typedef const __attribute__((address_space(4))) ConvP* KernargP;     // constant address space: scalar loads (s_load_dword)
__device__ __forceinline__ KernargP kernarg_descriptor() {
    const __attribute__((address_space(4))) char* kp = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    return (KernargP)kp;
}

template <typename T, int TW, bool FLIP>
__global__ __launch_bounds__(512) void conv3x3_halo_pkernel(const ConvP p, const int ntiles) {
    constexpr int TH = TW == 8 ? 8 : 256 / TW, HS = TW == 8 ? 40 : TW + 8, HR = TH + 2, NP = HR * HS / 8;
    static_assert(NP <= HALO_PIECES, "halo does not fit");
    constexpr int E_STORES = 8;                            // vector-memory operations a wave issues in the epilogue after the prefetch (wide path)
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [halo 0][halo 1][w 0][w 1][w 2]
    char* const sW = smem + 2 * HALO_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm4 = wave & 3, wn = wave >> 2;
    const int fr = lane & 15, fg = lane >> 4;
    const int C2 = p.C1 * 2;                               // bytes of one pixel's / one tap's channel vector (single source)
    const int nchunks = (p.C1 + 63) >> 6;
    const int pix_img = p.B * p.Hi * p.Wi;
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.N * p.ldw * 2, SRD_FLAGS);

    const int lrow = lane >> 3;
    const int slotb = ((lane & 7) ^ lrow) << 4;

    struct Tile { int mb, nb, b, y0, x0; };
    auto tile_of = [&](const ConvP& p, int vt) -> Tile {
        const int nb_n = (p.N + 127) >> 7;
        const int tcols = TW == 64 ? p.Wi >> 6 : 1;
        const int tiles_img = TW == 8 ? 1 : ((p.Hi * TW) >> 8) * tcols;
        Tile t;
        const int bid = remap_xcd(vt, ntiles);
        t.mb = bid / nb_n;
        t.nb = bid - t.mb * nb_n;
        t.b = TW == 8 ? t.mb * 4 : t.mb / tiles_img;
        const int trem = t.mb - t.b * tiles_img;
        t.y0 = TW == 8 ? 0 : (trem / tcols) * TH;
        t.x0 = (trem % tcols) * 64;
        return t;
    };
    // per-lane DMA source offsets of a tile: 7 halo pieces per wave, 2 weight pieces
    auto lane_offsets = [&](const ConvP& p, const Tile& t, bool live, unsigned (&h1)[7], unsigned (&wv)[2]) {
        const int C = p.C1;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int q = min(wave + 8 * i, NP - 1);
            const int hp = q * 8 + lrow;
            const int hy = hp / HS, hx = hp - hy * HS;
            const int img = TW == 8 ? hx / 10 : 0;
            const int y = t.y0 + hy - 1, x = TW == 8 ? hx - img * 10 - 1 : t.x0 + hx - 1;
            const bool ok = live && (unsigned)y < (unsigned)p.Hi && (unsigned)x < (unsigned)p.Wi && img < 4 && hx < TW + 2 + (TW == 8 ? 30 : 0);
            const int pix = ((t.b + img) * p.Hi + y) * p.Wi + x;
            h1[i] = ok ? (unsigned)(pix * p.C1 * 2 + slotb) : OOB;          // single source, whole 64-channel chunks (halo_persist_ok)
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = t.nb * 128 + (wave * 2 + j) * 8 + lrow;
            wv[j] = live && n < p.N && slotb < C * 2 ? (unsigned)(n * p.ldw * 2 + slotb) : OOB;
        }
    };

    int hoff[3][2][4];                                     // pixel-operand fragment addresses in the CURRENT halo buffer
    int woff[2][4];
    // (re)computed at the top of every tile: nothing of this is live across the epilogue, which needs the registers
    auto frag_addresses = [&](int buf) {
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
        const int base = (ly * HS + col) * ROWB + buf * HALO_BYTES;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) hoff[kx][sub][mt] = base + (((sub * 4 + fg) ^ ((col + kx) & 7)) << 4);
    }
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) woff[sub][nt] = lds_off(wn * 64 + nt * 16 + fr, sub * 4 + fg);
    };

    auto issue_w = [&](const unsigned (&wv)[2], int soff, int stage) {     // soff = (tap * C + 64 chunk) * 2 bytes
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_dst)(sW + stage * WSTAGE + (wave * 2 + j) * 1024), 16, wv[j], soff, 0, 0);
    };
    auto issue_w1 = [&](const unsigned (&wv)[2], int j, int soff, int stage) {   // one of issue_w's two pieces
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_dst)(sW + stage * WSTAGE + (wave * 2 + j) * 1024), 16, wv[j], soff, 0, 0);
    };
    const __amdgpu_buffer_rsrc_t rI = __builtin_amdgcn_make_buffer_rsrc((void*)p.in1, 0, pix_img * p.C1 * 2, SRD_FLAGS);
    auto issue_h = [&](const unsigned (&h1)[7], int i, int chunk, int buf) {
        const int q = min(wave + 8 * i, NP - 1);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rI, (lds_dst)(smem + buf * HALO_BYTES + q * 1024), 16, h1[i], (chunk << 6) * 2, 0, 0);
    };

    f32x4 acc[4][4];
    // `dma(g)` runs before MFMA group g = 4 sub + nt (4 MFMAs each): the k-step's LDS-DMA pieces are issued there, among MFMAs and
    // away from the fragment reads (DM_HALO_DMA_POS, igemm_dev.h)
    auto mma_tap = [&](int tap, auto&& dma) {
        const int ky = FLIP ? 2 - tap / 3 : tap / 3, kx = FLIP ? 2 - tap % 3 : tap % 3;
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
                dma(sub * 4 + nt);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) Mma<T>::run(fa[nt], fb[mt], acc[nt][mt]);
            }
        }
    };
    constexpr int G0 = DM_HALO_DMA_POS ? DM_HALO_DMA_POS / 100 : -1, G1 = DM_HALO_DMA_POS ? DM_HALO_DMA_POS / 10 % 10 : -1,
                  G2 = DM_HALO_DMA_POS ? DM_HALO_DMA_POS % 10 : -1;
    // wait before tap `tap`: everything older than the previous step's group has landed.  `after_epilogue`: chunk 0 of a tile that
    // follows another one in this workgroup — the previous tile's epilogue stores sit between the prefetch and this tap
    auto wait_tap = [&](int tap, bool after_epilogue) {
        if (tap == 0) { if (after_epilogue) wait_vmcnt<2 + E_STORES>(); else wait_vmcnt<2>(); }
        else if (tap == 1) { if (after_epilogue) wait_vmcnt<3 + E_STORES>(); else wait_vmcnt<3>(); }
        else if (tap <= 7) wait_vmcnt<3>();
        else wait_vmcnt<2>();
    };
    auto swap_halo = [&](int& hdelta) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) hoff[kx][sub][mt] += hdelta;
        hdelta = -hdelta;
    };

    unsigned hc1[7], wc[2], hn1[7], wn_[2];
    int vt = blockIdx.x;
    Tile cur = tile_of(p, vt);
    lane_offsets(p, cur, true, hc1, wc);
#pragma unroll
    for (int i = 0; i < 7; ++i) issue_h(hc1, i, 0, 0);
    issue_w(wc, 0, 0);
    issue_w(wc, C2, 1);
    // byte offset of the NEXT weight stage to fetch, (tap * C + 64 chunk) * 2, advanced by scalar adds the optimiser cannot
    // re-derive into nine hoisted multiples of C (asm volatile).  The adds write SCC: declared, or the compiler keeps the chunk
    // loop's exit compare live across them (first build: an endless chunk loop whenever a tile had >= 3 chunks)
    int wso = 2 * C2;
    auto wso_step = [&](int tap) {                         // after the issue of tap + 2: on to tap + 3 (or tap 0 of the next chunk)
        if ((tap + 3) % 9 == 0) asm volatile("s_sub_u32 %0, %0, %1\n\ts_add_u32 %0, %0, 0x80" : "+s"(wso) : "s"(8 * C2) : "scc");
        else asm volatile("s_add_u32 %0, %0, %1" : "+s"(wso) : "s"(C2) : "scc");
    };
    int cbuf = 0;
    bool first_tile = true;
    for (;;) {
        const int vnext = vt + gridDim.x;
        const bool has_next = vnext < ntiles;
        KernargP kq = kernarg_descriptor();
        ConvP pk;                                          // the few scalars the tile bookkeeping needs, re-read per tile
        pk.N = kq->N; pk.Hi = kq->Hi; pk.Wi = kq->Wi; pk.C1 = kq->C1; pk.ldw = kq->ldw;
        const Tile nxt = tile_of(pk, has_next ? vnext : vt);
        lane_offsets(pk, nxt, has_next, hn1, wn_);            // (no next tile: every offset out of range -> the DMA writes zeros)
        frag_addresses(cbuf);
        int hdelta = cbuf ? -HALO_BYTES : HALO_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // ---- chunks 0 .. nchunks - 2: the next chunk of THIS tile is fetched behind the taps
        for (int chunk = 0; chunk < nchunks - 1; ++chunk) {
            const bool ae = chunk == 0 && !first_tile;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                wait_tap(tap, ae);
                __builtin_amdgcn_s_barrier();
                const int t2 = (tap + 2) % 9;
#if DM_HALO_DMA_POS == 0
                issue_w(wc, wso, t2 % 3);
                wso_step(tap);
                if (tap < 7) issue_h(hc1, tap, chunk + 1, cbuf ^ 1);
                mma_tap(tap, [](int) {});
#else
                mma_tap(tap, [&](int g) {
                    if (g != G0 && g != G1 && g != G2) return;
                    __builtin_amdgcn_sched_barrier(0);
                    if (g == G0) issue_w1(wc, 0, wso, t2 % 3);
                    if (g == G1) { issue_w1(wc, 1, wso, t2 % 3); wso_step(tap); }
                    if (g == G2 && tap < 7) issue_h(hc1, tap, chunk + 1, cbuf ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
                });
#endif
            }
            swap_halo(hdelta);
            cbuf ^= 1;
        }
        // ---- last chunk: the FIRST chunk of the next tile is fetched behind the taps (halo: taps 0..6; weights of its taps 0 / 1: taps 7 / 8)
        {
            const int chunk = nchunks - 1;
            const bool ae = chunk == 0 && !first_tile;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                wait_tap(tap, ae);
                __builtin_amdgcn_s_barrier();
                const int t2 = (tap + 2) % 9;
                if (tap == 7) wso = 0;                               // the next tile starts at chunk 0, tap 0
#if DM_HALO_DMA_POS == 0
                if (tap + 2 >= 9) issue_w(wn_, wso, t2 % 3);
                else issue_w(wc, wso, t2 % 3);
                if (tap + 2 >= 9) asm volatile("s_add_u32 %0, %0, %1" : "+s"(wso) : "s"(C2) : "scc");
                else wso_step(tap);
                if (tap < 7) issue_h(hn1, tap, 0, cbuf ^ 1);
                mma_tap(tap, [](int) {});
#else
                mma_tap(tap, [&](int g) {
                    if (g != G0 && g != G1 && g != G2) return;
                    __builtin_amdgcn_sched_barrier(0);
                    if (g == G0) { if (tap + 2 >= 9) issue_w1(wn_, 0, wso, t2 % 3); else issue_w1(wc, 0, wso, t2 % 3); }
                    if (g == G1) {
                        if (tap + 2 >= 9) issue_w1(wn_, 1, wso, t2 % 3); else issue_w1(wc, 1, wso, t2 % 3);
                        if (tap + 2 >= 9) asm volatile("s_add_u32 %0, %0, %1" : "+s"(wso) : "s"(C2) : "scc");
                        else wso_step(tap);
                    }
                    if (g == G2 && tap < 7) issue_h(hn1, tap, 0, cbuf ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
                });
#endif
            }
            swap_halo(hdelta);
            cbuf ^= 1;
        }
        // every wave is past its last LDS operand read; ring stage 2 (last tap's weights) is free: the statistics fold goes there
        __builtin_amdgcn_s_barrier();
        {
            ConvP pe;                                       // scalars re-read here; the POINTERS come from the real kernel argument
            {
                KernargP kq = kernarg_descriptor();
                pe.act = kq->act; pe.out_nchw = kq->out_nchw; pe.B = kq->B; pe.Hi = kq->Hi; pe.Wi = kq->Wi; pe.C1 = kq->C1; pe.C2 = kq->C2;
                pe.Hq = kq->Hq; pe.Wq = kq->Wq; pe.Ho = kq->Ho; pe.Wo = kq->Wo; pe.osy = kq->osy; pe.osx = kq->osx; pe.ooy = kq->ooy;
                pe.oox = kq->oox; pe.N = kq->N; pe.ldc = kq->ldc; pe.coff = kq->coff; pe.M = kq->M; pe.stat_slots = kq->stat_slots;
                pe.splits = 1; pe.kper = 0;
            }
            pe.out = p.out; pe.scale = p.scale; pe.shift = p.shift; pe.psum = p.psum; pe.psq = p.psq; pe.addend = p.addend;   // (address space: global, not flat)
            pe.in1 = p.in1; pe.in2 = p.in2; pe.w = p.w; pe.ws = p.ws; pe.counters = p.counters;
            const int half = wm4 >> 1;
            const int m0 = cur.mb * 256, n0 = cur.nb * 128;
            const int tidh = (((wave & 1) + 2 * wn) << 6) + lane;
            const int tcols = TW == 64 ? pe.Wi >> 6 : 1;
            const int mw = (TW == 64 && tcols > 1) ? ((cur.b * pe.Hi + cur.y0 + wm4) * pe.Wi + cur.x0) : m0 + wm4 * 64;
            conv_epilogue<T, 128>(pe, acc, sW + 2 * WSTAGE + half * 4096, tidh, wm4 & 1, wn, fr, fg, cur.mb * 2 + half, mw - (wm4 & 1) * 64, n0);
        }
        if (!has_next) break;
        cur = nxt;
        vt = vnext;
        first_tile = false;
#pragma unroll
        for (int i = 0; i < 7; ++i) hc1[i] = hn1[i];
        wc[0] = wn_[0];
        wc[1] = wn_[1];
    }
    wait_vmcnt<0>();                                       // the zero-fill "prefetch" of the last tile
}

// the 8-stores-per-wave epilogue (igemm_dev.h conv_store, `wide`) and a reason to be persistent at all
bool halo_persist_ok(const ConvP& p, int tiles, int ncu) {
    // at least three tiles per workgroup: with two (the 32x32 layers at B = 64: 36 k-steps per tile) the cross-tile prefetch gains
    // less than the longer code path costs (same-box A/B: 64x64 128->128 -4 .. -6 %, 64x64 128->256 -6.6 %, 32x32 256->256 +1.5 .. +2.7 %)
    if (tiles < 3 * ncu || p.N % 128 != 0 || p.out_nchw) return false;
    if (((p.ldc | p.coff) & 7) != 0 || ((uintptr_t)p.out & 15) != 0 || ((uintptr_t)p.addend & 15) != 0) return false;
    if (p.C2 != 0 || p.C1 % 64 != 0 || p.B2 != p.B) return false;     // single source, whole chunks
    return true;
}

template <typename T, int TW, bool FLIP>
static int launch_halo_p(const ConvP& p, int tiles, int ncu, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3x3_halo_pkernel<T, TW, FLIP>, hipFuncAttributeMaxDynamicSharedMemorySize, HALO_LDS);
        if (e != hipSuccess) { dm_set_error("hipFuncSetAttribute(%d B LDS) failed: %s", HALO_LDS, hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    ConvP q = p;
    q.splits = 1;
    hipLaunchKernelGGL((conv3x3_halo_pkernel<T, TW, FLIP>), dim3((unsigned)ncu), dim3(512), HALO_LDS, st, q, tiles);
    DM_LAUNCH_CHECK();
    g_last_path = 1;
    g_last_persist = 1;
    return DM_OK;
}

int launch_halo_persist_any(const ConvP& p, bool is_f16, int tiles, int ncu, hipStream_t st) {
    const bool flip = p.ty < 0;
#define DM_HALOP(T)                                                                                                    \
    do {                                                                                                               \
        if (p.Wi >= 64) return flip ? launch_halo_p<T, 64, true>(p, tiles, ncu, st) : launch_halo_p<T, 64, false>(p, tiles, ncu, st); \
        if (p.Wi == 32) return flip ? launch_halo_p<T, 32, true>(p, tiles, ncu, st) : launch_halo_p<T, 32, false>(p, tiles, ncu, st); \
        if (p.Wi == 16) return flip ? launch_halo_p<T, 16, true>(p, tiles, ncu, st) : launch_halo_p<T, 16, false>(p, tiles, ncu, st); \
        return flip ? launch_halo_p<T, 8, true>(p, tiles, ncu, st) : launch_halo_p<T, 8, false>(p, tiles, ncu, st);    \
    } while (0)
    if (is_f16) DM_HALOP(f16);
    DM_HALOP(bf16);
#undef DM_HALOP
}

}  // namespace dmk
