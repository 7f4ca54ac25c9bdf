// conv3x3_packtap_kernel: 3x3 stride-1 layers whose INPUT has 8 (padded) channels — the stem (3 -> F, new_scripy.py:184 through
// ResConvBlock 3 -> F at :381) and the input gradient of the head (F -> 3, :314, run as a conv of the 3-channel gradient with the
// transposed weights).  The halo kernel gives such a layer one 64-channel chunk of which 56 channels are zeros: 9 taps x 2 MFMA
// sub-steps per tile for the work of 3, and 9 x 16 KiB of (mostly zero) weights through the ring.  Here the K dimension packs the
// TAPS: an MFMA 16x16x32 operand row is 4 taps x 8 channels, a lane's 16-byte fragment is one halo pixel at the tap its lane group
// stands for (k-group fg of k-step s = tap 4 s + fg), so the 9 taps are 3 k-steps (the last one three-quarters zero weights) and the
// im2col never exists: the same pixel vectors are read at shifted addresses.
//   * tile = 4 image rows x 64 columns (256 output pixels) x 128 output channels, 8 waves = 4 rows x 2 channel halves: wave tile,
//     accumulator layout and epilogue (statistics, scale / shift, activation, addend, stores) are the halo kernel's.
//   * halo: 6 x 72 pixels x 16 B, linear, filled by one LDS-DMA piece per wave (out-of-image lanes carry the out-of-range offset ->
//     zeros); weights: [12 taps][128 n] x 16 B (taps 9..11 zero) = 24 KiB, three pieces per wave.  One fill, one barrier, 48 MFMAs per
//     wave, epilogue: the kernel is bound by its 64 KiB of output per tile, several workgroups per CU overlap fill and stores.
#include "igemm_dev.h"

namespace dmk {

constexpr int PT_HS = 72;                                  // halo row pitch in pixels
constexpr int PT_HALO = 7 * 1024;                          // 6 x 72 x 16 B = 6912 B, filled as 7 pieces
constexpr int PT_W = 12 * 128 * 16;                        // weights
constexpr int PT_LDS = PT_HALO + PT_W + 2 * 4096;          // + the epilogue's statistics scratch

template <typename T, bool FLIP>
__global__ __launch_bounds__(512) void conv3x3_packtap_kernel(const ConvP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [halo][weights][statistics]
    char* const sH = smem;
    char* const sW = smem + PT_HALO;
    char* const sS = sW + PT_W;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm4 = wave & 3, wn = wave >> 2;
    const int fr = lane & 15, fg = lane >> 4;
    const int nb_n = (p.N + 127) >> 7;
    const int ntiles = gridDim.x;
    const int bid = remap_xcd(blockIdx.x, ntiles);
    const int mb = bid / nb_n, nb = bid - mb * nb_n;
    const int m0 = mb * 256, n0 = nb * 128;
    const int tcols = p.Wi >> 6;
    const int tiles_img = (p.Hi >> 2) * tcols;
    const int b = mb / tiles_img;
    const int trem = mb - b * tiles_img;
    const int y0 = (trem / tcols) * 4, x0 = (trem % tcols) * 64;

    // ---- fill: halo piece `wave` (waves 0..6), weight pieces 3 wave .. 3 wave + 2 (piece = tap t, 64 rows n)
    {
        const __amdgpu_buffer_rsrc_t rI = __builtin_amdgcn_make_buffer_rsrc((void*)p.in1, 0, p.B * p.Hi * p.Wi * 16, SRD_FLAGS);
        const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.N * p.ldw * 2, SRD_FLAGS);
        if (wave < 7) {
            const int hp = wave * 64 + lane;
            const int hy = hp / PT_HS, hx = hp - hy * PT_HS;
            const int y = y0 + hy - 1, x = x0 + hx - 1;
            const bool ok = hy < 6 && hx < 66 && (unsigned)y < (unsigned)p.Hi && (unsigned)x < (unsigned)p.Wi;
            const unsigned v = ok ? (unsigned)(((b * p.Hi + y) * p.Wi + x) * 16) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rI, (lds_dst)(sH + wave * 1024), 16, v, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int piece = wave * 3 + j;                // 0 .. 23
            const int t = piece >> 1, n = n0 + (piece & 1) * 64 + lane;
            const unsigned v = (t < 9 && n < p.N) ? (unsigned)((n * p.ldw + t * 8) * 2) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_dst)(sW + piece * 1024), 16, v, 0, 0, 0);
        }
    }

    // ---- per-lane fragment addresses: k-step s, lane group fg -> tap 4 s + fg (taps 9..11: zero weights; the pixel of tap 8 is read)
    int hoff[3], woff[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int t = 4 * s + fg, tt = t < 9 ? t : 8;
        const int ky = FLIP ? 2 - tt / 3 : tt / 3, kx = FLIP ? 2 - tt % 3 : tt % 3;
        hoff[s] = ((wm4 + ky) * PT_HS + fr + kx) * 16;
        woff[s] = (t * 128 + wn * 64 + fr) * 16;
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    wait_vmcnt<0>();
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        u32x4 fb[4], fa[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) fb[mt] = *(const u32x4*)(sH + hoff[s] + mt * 256);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) fa[nt] = *(const u32x4*)(sW + woff[s] + nt * 256);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) Mma<T>::run(fa[nt], fb[mt], acc[nt][mt]);
    }

    const int half = wm4 >> 1;
    const int tidh = (((wave & 1) + 2 * wn) << 6) + lane;
    const int mw = tcols > 1 ? ((b * p.Hi + y0 + wm4) * p.Wi + x0) : m0 + wm4 * 64;   // first output pixel of this wave's 64
    conv_epilogue<T, 128>(p, acc, sS + half * 4096, tidh, wm4 & 1, wn, fr, fg, mb * 2 + half, mw - (wm4 & 1) * 64, n0);
}

// 16-bit input of exactly 8 (padded) channels, one source, full-resolution geometry of the halo kernel's TW = 64 form
bool packtap_ok(const ConvP& p) {
    return p.C1 == 8 && p.C2 == 0 && p.splits <= 1 && p.T == 9 && p.ldw == 72 && p.Wi >= 64 && (p.Wi & 63) == 0 && (p.Hi & 3) == 0 && p.B2 >= p.B &&
           p.M == p.B * p.Hi * p.Wi && (((uintptr_t)p.in1 | (uintptr_t)p.w) & 15) == 0 && (int64_t)p.B * p.Hi * p.Wi * 16 < (int64_t)1 << 31;
}

template <typename T, bool FLIP>
static int launch_packtap(const ConvP& p, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3x3_packtap_kernel<T, FLIP>, hipFuncAttributeMaxDynamicSharedMemorySize, PT_LDS);
        if (e != hipSuccess) { dm_set_error("hipFuncSetAttribute(%d B LDS) failed: %s", PT_LDS, hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    const int tiles = (p.M / 256) * cdiv(p.N, 128);
    ConvP q = p;
    q.splits = 1;
    hipLaunchKernelGGL((conv3x3_packtap_kernel<T, FLIP>), dim3((unsigned)tiles), dim3(512), PT_LDS, st, q);
    DM_LAUNCH_CHECK();
    g_last_path = 4;
    return DM_OK;
}

// =================================================================================================
// conv3x3_narrow_kernel: 3x3 stride-1 layers with at most 16 OUTPUT channels — the head (F -> 3, new_scripy.py:314)
// =================================================================================================
// The gather kernel's 32-wide tiles re-fetch every input pixel once per tap (9 x 67 MB through L2 -> LDS at B = 64); a 128-wide
// halo tile would be 97 % padding in the MFMA and in the weight stream.  Here a 256-thread workgroup keeps the input halo of a
// 64-channel chunk resident (the halo kernel's image: 6 x 72 pixels x 128 B, 16-byte slot v of pixel hp at v ^ (hp & 7)) together
// with ALL nine taps of the chunk's weights for 16 output channels (9 x 16 rows x 128 B = 18 KiB); each of the four waves owns one
// image row (64 pixels) x 16 channels: 4 MFMAs per 32-channel sub-step, 5 fragment reads.  One fill and two barriers per chunk, no
// ring: at 72 KiB of LDS and < 64 VGPRs two workgroups share a CU and overlap each other's fills.  The kernel is bound by reading
// its input once.
constexpr int NR_HALO = 54 * 1024;                         // 6 x 72 pixels x 128 B
constexpr int NR_W = 9 * 16 * 128;                         // 18 KiB
constexpr int NR_LDS = NR_HALO + NR_W;

template <typename T>
__global__ __launch_bounds__(256) void conv3x3_narrow_kernel(const ConvP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [halo][weights]
    char* const sH = smem;
    char* const sW = smem + NR_HALO;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int nb_n = (p.N + 15) >> 4;
    const int bid = remap_xcd(blockIdx.x, gridDim.x);
    const int mb = bid / nb_n, nb = bid - mb * nb_n;
    const int n0 = nb * 16;
    const int tcols = p.Wi >> 6;
    const int tiles_img = (p.Hi >> 2) * tcols;
    const int b = mb / tiles_img;
    const int trem = mb - b * tiles_img;
    const int y0 = (trem / tcols) * 4, x0 = (trem % tcols) * 64;
    const int C = p.C1, nchunks = C >> 6;

    const __amdgpu_buffer_rsrc_t rI = __builtin_amdgcn_make_buffer_rsrc((void*)p.in1, 0, p.B * p.Hi * p.Wi * C * 2, SRD_FLAGS);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.N * p.ldw * 2, SRD_FLAGS);
    const int lrow = lane >> 3;
    const int slotb = ((lane & 7) ^ lrow) << 4;
    unsigned hv[14], wv[5];                                 // halo pieces wave + 4 i (54), weight pieces wave + 4 j (18)
#pragma unroll
    for (int i = 0; i < 14; ++i) {
        const int q = min(wave + 4 * i, 53);                // (surplus pieces re-fetch the last one: same bytes, same place)
        const int hp = q * 8 + lrow;
        const int hy = hp / 72, hx = hp - hy * 72;
        const int y = y0 + hy - 1, x = x0 + hx - 1;
        const bool ok = hx < 66 && (unsigned)y < (unsigned)p.Hi && (unsigned)x < (unsigned)p.Wi;
        hv[i] = ok ? (unsigned)(((b * p.Hi + y) * p.Wi + x) * C * 2 + slotb) : OOB;
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int q = min(wave + 4 * j, 17);
        const int r = q * 8 + lrow;                         // row = tap * 16 + n
        const int t = r >> 4, n = n0 + (r & 15);
        wv[j] = n < p.N ? (unsigned)((n * p.ldw + t * C) * 2 + slotb) : OOB;
    }
    int hoff[3][2], woff[2];                                // this wave's row `wave` of the tile
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) hoff[kx][sub] = (wave * 72 + fr + kx) * 128 + (((sub * 4 + fg) ^ ((fr + kx) & 7)) << 4);
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) woff[sub] = fr * 128 + (((sub * 4 + fg) ^ (fr & 7)) << 4);

    f32x4 acc[1][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[0][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        if (chunk > 0) __syncthreads();                     // every wave is done reading the previous chunk
#pragma unroll
        for (int i = 0; i < 14; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rI, (lds_dst)(sH + min(wave + 4 * i, 53) * 1024), 16, hv[i], chunk * 128, 0, 0);
#pragma unroll
        for (int j = 0; j < 5; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_dst)(sW + min(wave + 4 * j, 17) * 1024), 16, wv[j], chunk * 128, 0, 0);
        wait_vmcnt<0>();
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                u32x4 fb[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) fb[mt] = *(const u32x4*)(sH + hoff[kx][sub] + (ky * 72 + mt * 16) * 128);
                const u32x4 fa = *(const u32x4*)(sW + woff[sub] + tap * 16 * 128);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) Mma<T>::run(fa, fb[mt], acc[0][mt]);
            }
        }
    }
    // each wave finishes its own 64 pixels x 16 channels: the 32-wide epilogue with (wm, wn) = (0, 0) and the wave's first pixel
    const int mw = (b * p.Hi + y0 + wave) * p.Wi + x0;
    conv_epilogue<T, 32>(p, acc, smem, tid, 0, 0, fr, fg, mb, mw, n0);
}

// forward taps only, whole 64-channel chunks of one source, no statistics (the head has no BatchNorm), full-resolution geometry
bool narrow_ok(const ConvP& p) {
    if (p.T != 9 || p.KW != 3 || p.sy != 1 || p.sx != 1 || p.N > 16 || p.psum != nullptr || p.splits > 1) return false;
    if (!(p.ty == 1 && p.tx == 1 && p.oy0 == -1 && p.ox0 == -1)) return false;
    if (p.Hq != p.Hi || p.Wq != p.Wi || p.Ho != p.Hi || p.Wo != p.Wi || p.osy != 1 || p.osx != 1 || p.ooy != 0 || p.oox != 0) return false;
    if (p.Wi < 64 || (p.Wi & 63) || (p.Hi & 3) || p.C2 != 0 || p.C1 < 64 || (p.C1 & 63) || p.B2 < p.B || p.M != p.B * p.Hi * p.Wi) return false;
    if (((uintptr_t)p.in1 | (uintptr_t)p.w) & 15) return false;
    return (int64_t)p.B * p.Hi * p.Wi * p.C1 * 2 < (1ll << 31) && (int64_t)p.N * p.ldw * 2 < (1ll << 31);
}

template <typename T>
static int launch_narrow(const ConvP& p, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3x3_narrow_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, NR_LDS);
        if (e != hipSuccess) { dm_set_error("hipFuncSetAttribute(%d B LDS) failed: %s", NR_LDS, hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    const int tiles = (p.M / 256) * cdiv(p.N, 16);
    ConvP q = p;
    q.splits = 1;
    hipLaunchKernelGGL((conv3x3_narrow_kernel<T>), dim3((unsigned)tiles), dim3(256), NR_LDS, st, q);
    DM_LAUNCH_CHECK();
    g_last_path = 5;
    return DM_OK;
}

int launch_narrow_any(const ConvP& p, bool is_f16, hipStream_t st) {
    return is_f16 ? launch_narrow<f16>(p, st) : launch_narrow<bf16>(p, st);
}

int launch_packtap_any(const ConvP& p, bool is_f16, hipStream_t st) {
    const bool flip = p.ty < 0;
    if (is_f16) return flip ? launch_packtap<f16, true>(p, st) : launch_packtap<f16, false>(p, st);
    return flip ? launch_packtap<bf16, true>(p, st) : launch_packtap<bf16, false>(p, st);
}

}  // namespace dmk
