// dm_conv: the convolutions of the denoiser on the CDNA4 matrix cores (forward, input-gradient, ConvTranspose,
// and — as a 1x1 "image" — the dense fp32 layers).  Two kernel families:
//
//   * conv3x3_halo_kernel (further down): the 3x3 stride-1 layers — 93 % of the FLOPs — with the input halo of a
//     256-pixel tile resident in LDS and only the weights streaming; split-K through the workspace for few-tile /
//     deep-K layers.  Its header comment explains why the gather form is L2->LDS-fill-bound on these layers.
//   * the gather kernels described here: everything else (4x4 stride 2, 1x1, ConvTranspose, dense layers, odd sizes).
//
// Gather kernels (include/dm_amd.h, dm_conv): the
// output tile is 128 output pixels x BN output channels per 256-thread workgroup (4 waves as 2(m) x 2(n)),
// K runs over taps x input channels in steps of 128 bytes per row (64 bf16 / 32 f32 channels).  Both
// operands are staged in LDS as [row][128 B] images whose 16-byte vectors are XOR-swizzled with the row
// index (vec ^= row & 7): the ds_read_b128 fragment reads are conflict-free (every 16-lane group of the
// instruction hits 16 distinct 16-B slots of the 256-B bank row).
//
// Two staging pipelines:
//   v2 LDS-DMA (2-stage ring; 4 stages when the grid does not fill the chip; split-K through the workspace when a handful
//      of tiles share a long reduction): `global_load_lds_dwordx4` writes the stage directly (no VGPR round trip, no
//      ds_write, almost no per-step address arithmetic: per-lane row offsets and a per-row tap-validity
//      bitmask are computed once, a k-step costs one add + one select per 1-KiB piece).  The LDS image
//      is lane-linear per wave-instruction (8 rows x 8 vectors), so the swizzle sits on the SOURCE
//      address: lane L fetches logical vector (L&7) ^ (L>>3) of row L>>3.  Zero padding (3x3 halo, ragged
//      tiles, channel tails) = lanes pointed at a small zero page.  NS-stage ring, one raw s_barrier per
//      k-step, counted `s_waitcnt vmcnt(N)` so NS-2 stages stay in flight across the barrier.
//   v1 register staging (global_load -> VGPR -> ds_write), one step ahead, LDS double buffer.  Kept as the
//      fallback for tensors beyond 2^31 elements and for A/B measurements (dm_set_conv_variant).
//
// MFMA orientation is "swapped": A operand = packed weights (row i = output channel), B operand =
// gathered input pixels (column j = output pixel), so every lane ends up with 4 consecutive output
// channels of one pixel -> one 8-byte (bf16) / 16-byte (f32) NHWC store per 16x16 tile.
//   bf16: v_mfma_f32_16x16x32_bf16, a lane's fragment = 8 consecutive k of its row.
//   f32 : v_mfma_f32_16x16x4_f32 x 4 per 16-B fragment; element j of lane group g is k = 4g + j on both
//         operands, so the four MFMAs together cover the 16 k of the sub-step exactly once.
//         (f32 MFMA is a k-ordered fmaf chain: exact fp32, which is what the 1e-4 parity mode needs.)
//
// Workgroup -> tile mapping is XCD-aware: the dispatcher deals consecutive workgroups round-robin over
// the 8 XCDs, so tile = (bid % 8) * (ntiles / 8) + bid / 8 (bijective form) gives every XCD a contiguous
// run of tiles — vertically adjacent image rows (shared 3x3 halo) and the n-tiles of one pixel block
// then meet in the same 4 MiB L2.  Placement only affects speed, never results.
#include <cstdlib>
#include "igemm_dev.h"

namespace dmk {

// =================================================================================================
// v1: register staging
// =================================================================================================
template <typename T, int BN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvP p) {
    constexpr int VE = Elem<T>::VE;
    constexpr int BK = 8 * VE;                // channels per k-step
    constexpr int NT = BN / 32;               // 16-wide channel tiles per wave
    constexpr int BV = (BN * 8 + 255) / 256;  // weight vectors per thread per k-step
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    __shared__ __attribute__((aligned(16))) char smem[2 * (A_BYTES + B_BYTES)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int fr = lane & 15, fg = lane >> 4;
    const int nb_n = (p.N + BN - 1) / BN;
    const int bid = remap_xcd(blockIdx.x, gridDim.x);
    const int mb = bid / nb_n, nb = bid - mb * nb_n;
    const int m0 = mb * BM, n0 = nb * BN;
    const int C = p.C1 + p.C2;
    const T* in1 = (const T*)p.in1;
    const T* in2 = (const T*)p.in2;
    const T* wgt = (const T*)p.w;

    // per-thread staging assignment: A rows (tid>>3) + 32*i, vector tid&7
    const int sv = tid & 7, srow = tid >> 3;
    int a_b[4], a_y[4], a_x[4];
    bool a_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + srow + 32 * i;
        a_ok[i] = m < p.M;
        const int mm = a_ok[i] ? m : 0;
        const int qx = mm % p.Wq, tq = mm / p.Wq;
        const int qy = tq % p.Hq;
        a_b[i] = tq / p.Hq;
        a_y[i] = qy * p.sy + p.oy0;
        a_x[i] = qx * p.sx + p.ox0;
    }

    // one register set: global loads run one k-step ahead of the MFMAs (in registers on their way to LDS);
    // a second workgroup on the CU covers the rest of the HBM/L2 latency.  (Two register sets — loads two
    // k-steps ahead — were measured: 198 VGPRs drop the kernel to one workgroup per CU, 30 % slower.)
    u32x4 ra[4], rb[BV];
    int ld_t = 0, ld_ky = 0, ld_kx = 0, ld_c0 = 0;   // tap state of the NEXT load (no division in the loop)
    auto gload = [&]() {
        const int dy = ld_ky * p.ty, dx = ld_kx * p.tx;
        const int c = ld_c0 + sv * VE;
        const bool c_ok = c < C;
        const bool first = c < p.C1;
        const T* src_base = first ? in1 : in2;
        const int cs = first ? p.C1 : p.C2, cc = first ? c : c - p.C1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int iy = a_y[i] + dy, ix = a_x[i] + dx;
            const bool ok = a_ok[i] && c_ok && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (ok) {
                const size_t pix = ((size_t)a_b[i] * p.Hi + iy) * p.Wi + ix;
                v = *(const u32x4*)(src_base + pix * cs + cc);
            }
            ra[i] = v;
        }
#pragma unroll
        for (int j = 0; j < BV; ++j) {
            const int row = srow + 32 * j;
            const int n = n0 + row;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (row < BN && n < p.N && c_ok) v = *(const u32x4*)(wgt + (size_t)n * p.ldw + (size_t)ld_t * C + c);
            rb[j] = v;
        }
        ld_c0 += BK;
        if (ld_c0 >= C) {
            ld_c0 = 0;
            ++ld_t;
            if (++ld_kx == p.KW) { ld_kx = 0; ++ld_ky; }
        }
    };
    auto sstore = [&](int buf) {
        char* sA = smem + buf * (A_BYTES + B_BYTES);
        char* sB = sA + A_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) *(u32x4*)(sA + lds_off(srow + 32 * i, sv)) = ra[i];
#pragma unroll
        for (int j = 0; j < BV; ++j) {
            const int row = srow + 32 * j;
            if (row < BN) *(u32x4*)(sB + lds_off(row, sv)) = rb[j];
        }
    };

    f32x4 acc[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nsteps = p.T * ((C + BK - 1) / BK);
    gload();
    sstore(0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int cur = s & 1;
        const bool more = s + 1 < nsteps;
        if (more) gload();                         // step s+1 in flight while LDS[cur] (step s) is consumed
        const char* sA = smem + cur * (A_BYTES + B_BYTES);
        mma_stage<T, BN>(sA, sA + A_BYTES, wm, wn, fr, fg, acc);
        if (more) sstore(cur ^ 1);
        __syncthreads();
    }
    conv_epilogue<T, BN>(p, acc, smem, tid, wm, wn, fr, fg, mb, m0, n0);
}

// =================================================================================================
// v2: LDS-DMA staging, NS-stage ring
// =================================================================================================
typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* gbl_vptr;


template <typename T, int BN, int NS>
__global__ __launch_bounds__(256) void conv_igemm2_kernel(const ConvP p) {
    constexpr int VE = Elem<T>::VE;
    constexpr int BK = 8 * VE;
    constexpr int NT = BN / 32;
    constexpr int GB = BN / 32;                 // weight pieces (8 rows x 128 B) per wave per k-step
    constexpr int G = 4 + GB;                   // LDS-DMA instructions per wave per k-step
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // NS * STAGE; the ONLY LDS object of the kernel

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int fr = lane & 15, fg = lane >> 4;
    const int nb_n = (p.N + BN - 1) / BN;
    const int ntiles = gridDim.x / p.splits;
    const int split = blockIdx.x / ntiles;
    const int bid = remap_xcd(blockIdx.x - split * ntiles, ntiles);
    const int mb = bid / nb_n, nb = bid - mb * nb_n;
    const int m0 = mb * BM, n0 = nb * BN;
    const int C = p.C1 + p.C2;
    // base addresses as laundered scalars: stops the compiler from turning `cond ? p.in1 : p.in2` into a
    // per-lane LOAD from the kernarg segment (an ordinary global load in the loop would drain the LDS-DMA queue)
    unsigned long long a_in1 = (unsigned long long)p.in1, a_in2 = (unsigned long long)(p.in2 ? p.in2 : p.in1);
    unsigned long long a_w = (unsigned long long)p.w, a_zero = (unsigned long long)g_zero_page;
    asm volatile("" : "+s"(a_in1), "+s"(a_in2), "+s"(a_w), "+s"(a_zero));

    // ---- per-lane constants: the piece a wave-instruction writes is 8 rows x 8 slots, lane L -> (row L>>3, slot L&7)
    const int lrow = lane >> 3;
    const int cl = (((lane & 7) ^ lrow)) * VE;          // channel offset of this lane's logical vector within a k-step
    int offA1[4], offA2[4];                              // element offset of tap (0,0) of the lane's 4 pixel rows
    unsigned long long tapmask[4];                       // bit t: tap t of that pixel is inside the image
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wave * 32 + i * 8 + lrow;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int qx = mm % p.Wq, tq = mm / p.Wq;
        const int qy = tq % p.Hq, b = tq / p.Hq;
        const int by = qy * p.sy + p.oy0, bx = qx * p.sx + p.ox0;
        const int pix = (b * p.Hi + by) * p.Wi + bx;
        offA1[i] = pix * p.C1;
        offA2[i] = pix * p.C2 - p.C1;                    // second source is indexed with (c - C1)
        unsigned long long mk = 0ull;
        int ky = 0, kx = 0;
        for (int t = 0; t < p.T; ++t) {
            const int iy = by + ky * p.ty, ix = bx + kx * p.tx;
            if (ok && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi) mk |= 1ull << t;
            if (++kx == p.KW) { kx = 0; ++ky; }
        }
        tapmask[i] = mk;
    }
    int offB[GB];
    bool okB[GB];
#pragma unroll
    for (int j = 0; j < GB; ++j) {
        const int n = n0 + wave * (BN / 4) + j * 8 + lrow;
        okB[j] = n < p.N;
        offB[j] = (okB[j] ? n : 0) * p.ldw;
    }

    const int spt = (C + BK - 1) / BK;                   // k-steps per tap
    const int ks_lo = split * p.kper;                    // this split's k-steps [ks_lo, ks_lo + nsteps)
    int ld_t = ks_lo / spt, ld_c0 = (ks_lo - ld_t * spt) * BK;   // state of the NEXT k-step to be issued (wave-uniform)
    int ld_ky = ld_t / p.KW, ld_kx = ld_t - ld_ky * p.KW;
    auto issue = [&](int stage) {
        char* sA = smem + stage * STAGE;
        char* sB = sA + A_BYTES;
        const int c = ld_c0 + cl;
        const bool c_ok = c < C;
        const bool first = c < p.C1;
        const int tap_pix = ld_ky * p.ty * p.Wi + ld_kx * p.tx;                 // uniform
        const int tapA = first ? tap_pix * p.C1 + c : tap_pix * p.C2 + c;
        const unsigned long long srcA = first ? a_in1 : a_in2;                  // branch-free selects on integers:
#pragma unroll                                                                   // no loads, no divergence in the loop
        for (int i = 0; i < 4; ++i) {
            const bool v = c_ok && ((tapmask[i] >> ld_t) & 1ull);
            const int off = (first ? offA1[i] : offA2[i]) + tapA;
            unsigned long long g = srcA + (unsigned long long)(unsigned)off * sizeof(T);
            g = v ? g : a_zero;
            __builtin_amdgcn_global_load_lds((gbl_vptr)g, (lds_vptr)(sA + (wave * 32 + i * 8) * ROWB), 16, 0, 0);
        }
        const int kB = ld_t * C + c;
#pragma unroll
        for (int j = 0; j < GB; ++j) {
            unsigned long long g = a_w + (unsigned long long)(unsigned)(offB[j] + kB) * sizeof(T);
            g = (okB[j] && c_ok) ? g : a_zero;
            __builtin_amdgcn_global_load_lds((gbl_vptr)g, (lds_vptr)(sB + (wave * (BN / 4) + j * 8) * ROWB), 16, 0, 0);
        }
        ld_c0 += BK;
        if (ld_c0 >= C) {
            ld_c0 = 0;
            ++ld_t;
            if (++ld_kx == p.KW) { ld_kx = 0; ++ld_ky; }
        }
    };

    f32x4 acc[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nsteps = min(p.kper, p.T * spt - ks_lo);
    // prologue: NS-1 stages in flight
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nsteps) issue(s);
    int cons = 0;                        // stage consumed this iteration
    int prod = NS - 1;                   // stage the next issue writes (the one consumed last iteration)
    for (int s = 0; s < nsteps; ++s) {
        // step s has landed once at most NS-2 younger groups are outstanding (exact while the ring is full)
        if (s + NS - 1 <= nsteps) wait_vmcnt<G * (NS - 2)>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();    // publishes step s of every wave; every wave is done reading stage `prod`
        if (s + NS - 1 < nsteps) issue(prod);
        const char* sA = smem + cons * STAGE;
        mma_stage<T, BN>(sA, sA + A_BYTES, wm, wn, fr, fg, acc);
        prod = cons;
        cons = cons + 1 == NS ? 0 : cons + 1;
    }
    if (p.splits > 1) {
        store_partial<BN>(p, acc, split, ntiles, bid, wave, lane);
        return;
    }
    __syncthreads();                     // the epilogue reuses the LDS for the statistics fold
    conv_epilogue<T, BN>(p, acc, smem, tid, wm, wn, fr, fg, mb, m0, n0);
}

// sums the split partials of a 128 x BN tile back into the accumulator registers and runs the ordinary epilogue
template <typename T, int BN>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const ConvP p) {
    constexpr int NT = BN / 32;
    __shared__ __attribute__((aligned(16))) char smem[4 * BN * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1, fr = lane & 15, fg = lane >> 4;
    const int nb_n = (p.N + BN - 1) / BN;
    const int tiles = gridDim.x, bid = blockIdx.x;
    const int mb = bid / nb_n, nb = bid - mb * nb_n;
    f32x4 acc[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < p.splits; ++k) {
        const f32x4* src = (const f32x4*)p.ws + ((((size_t)k * tiles + bid) * 4 + wave) * (NT * 4)) * 64 + lane;
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] += src[(i * 4 + j) * 64];
    }
    conv_epilogue<T, BN>(p, acc, smem, tid, wm, wn, fr, fg, mb, mb * BM, nb * BN);
}

// =================================================================================================
// v7: pointwise (1x1) convolution with a short reduction — operands straight from global memory
// =================================================================================================
// The 1x1 layers of UnetDown (channel_compress C -> C/4, ch_adjust C/4 -> Cout: new_scripy.py:217-222) and their input gradients reduce
// over 32 ... 128 channels only: one or two k-steps per 128-pixel tile.  On the gather kernel a workgroup then spends its life in
// latency (DMA -> wait -> 1-2 k-steps -> epilogue), two workgroups per CU keep < 32 KiB in flight, and the layers ran at 1.1-2.9 TB/s
// of the ~4.5 TB/s they are worth (they are pure HBM streaming: 84 MB per launch at 64x64).  Here nothing is staged: every lane loads
// its MFMA fragments directly — a pixel fragment is 16 pixels x 64 contiguous bytes of their channel rows, a weight fragment
// 16 rows of the (L2-resident) weight matrix — all of a tile's loads are issued up front (4 waves x up to 32 KiB in flight per
// workgroup, 2-4 workgroups per CU: no LDS, 110-200 VGPRs), the compiler's own s_waitcnt orders them, and the tile goes through the
// ordinary epilogue (bias / BatchNorm statistics / addend / 16-byte stores).  K in {32, 64, 128} (KS = K / 32 sub-steps).
template <typename T, int BN, int KS>
__global__ __launch_bounds__(256) void conv_pw_kernel(const ConvP p) {
    constexpr int NT = BN / 32;
    __shared__ __attribute__((aligned(16))) char smem[4 * BN * 4];       // statistics fold of the epilogue
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1, fr = lane & 15, fg = lane >> 4;
    const int nb_n = (p.N + BN - 1) / BN;
    const int bid = remap_xcd(blockIdx.x, gridDim.x);
    const int mb = bid / nb_n, nb = bid - mb * nb_n;
    const int m0 = mb * BM, n0 = nb * BN;
    const int K = p.C1;
    const T* x = (const T*)p.in1;
    const T* w = (const T*)p.w;
    u32x4 fb[KS][4], fa[KS][NT];
    const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = m0 + wm * 64 + mt * 16 + fr;
        const T* row = x + (size_t)(m < p.M ? m : 0) * K + fg * 8;
#pragma unroll
        for (int sk = 0; sk < KS; ++sk) fb[sk][mt] = m < p.M ? *(const u32x4*)(row + sk * 32) : zero;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + wn * (BN / 2) + nt * 16 + fr;
        const T* row = w + (size_t)(n < p.N ? n : 0) * p.ldw + fg * 8;
#pragma unroll
        for (int sk = 0; sk < KS; ++sk) fa[sk][nt] = n < p.N ? *(const u32x4*)(row + sk * 32) : zero;
    }
    f32x4 acc[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sk = 0; sk < KS; ++sk)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) Mma<T>::run(fa[sk][nt], fb[sk][mt], acc[nt][mt]);
    conv_epilogue<T, BN>(p, acc, smem, tid, wm, wn, fr, fg, mb, m0, n0);
}

int g_splitk_inkernel = 1;   // r03: on (the sc1 hand-off of igemm_dev.h costs no fence); 0: separate split-K epilogue launches
int g_last_path = 0;  // kernel family the last dm_conv launch used: 0 = gather (conv_igemm*), 1 = conv3x3_halo_kernel, 3 = pointwise, 4 = packed-tap, 5 = narrow (<= 16 output channels)
int g_packtap = 1;    // 8-channel inputs of 3x3 layers on conv3x3_packtap_kernel (igemm_skinny.hip)

int g_variant = 5;    // 1 = register staging, 2..4 = LDS-DMA with that many ring stages (2 workgroups/CU at 2),
                      // 5 (default) = halo-resident kernel for eligible 3x3 layers, else LDS-DMA with 2 stages (4 on small grids)

template <typename T, int BN, int NS>
int launch2(const ConvP& p, int64_t grid, hipStream_t st) {
    constexpr int bytes = NS * (BM + BN) * ROWB;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_igemm2_kernel<T, BN, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) { dm_set_error("hipFuncSetAttribute(%d B LDS) failed: %s", bytes, hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_igemm2_kernel<T, BN, NS>), dim3((unsigned)grid), dim3(256), bytes, st, p);
    return DM_OK;
}

template <typename T, int BN>
int launch_bn(const ConvP& p, int64_t grid, int variant, hipStream_t st) {
    int rc = DM_OK;
    switch (variant) {
        case 1: hipLaunchKernelGGL((conv_igemm_kernel<T, BN>), dim3((unsigned)grid), dim3(256), 0, st, p); break;
        case 3: rc = launch2<T, BN, 3>(p, grid, st); break;
        case 4: rc = launch2<T, BN, 4>(p, grid, st); break;
        default: rc = launch2<T, BN, 2>(p, grid, st); break;
    }
    return rc;
}

// 3x3, stride 1, pad 1, whole image rows of 16/32/64 pixels, 64-channel chunks, byte offsets below 2^31
bool halo_eligible(const ConvP& p) {
    if (p.T != 9 || p.KW != 3 || p.sy != 1 || p.sx != 1) return false;
    if (p.N <= 32) return false;                                       // the 3-channel head: a 128-wide tile would be 97 % padding; the gather kernel has 32-wide tiles
    // forward taps (y-1+ky, x-1+kx), or the input-gradient's mirrored traversal (y+1-ky, x+1-kx)
    if (!((p.ty == 1 && p.tx == 1 && p.oy0 == -1 && p.ox0 == -1) || (p.ty == -1 && p.tx == -1 && p.oy0 == 1 && p.ox0 == 1))) return false;
    if (p.Hq != p.Hi || p.Wq != p.Wi || p.Ho != p.Hi || p.Wo != p.Wi || p.osy != 1 || p.osx != 1 || p.ooy != 0 || p.oox != 0) return false;
    if (p.Wi == 8) {                                                   // four whole 8x8 images per tile
        if (p.Hi != 8 || p.B % 4 != 0) return false;
    } else if (p.Wi > 64) {                                            // column tiles of 4 rows x 64 pixels
        if (p.Wi % 64 != 0 || p.Hi % 4 != 0) return false;
    } else if ((p.Wi != 16 && p.Wi != 32 && p.Wi != 64) || (p.Hi * p.Wi) % 256 != 0) {
        return false;
    }
    // whole 64-channel chunks, or one partial chunk of a single source (the 8-channel stem / head gradients)
    if ((p.C1 % 64 != 0 || p.C2 % 64 != 0) && !(p.C2 == 0 && p.C1 < 64)) return false;
    const int64_t pix = (int64_t)p.B * p.Hi * p.Wi;
    const int64_t cmax = p.C1 > p.C2 ? p.C1 : p.C2;
    return pix * cmax * 2 < (1ll << 31) && (int64_t)p.N * p.ldw * 2 < (1ll << 31);
}

// 0: not eligible; 1: the 4x4 / stride-2 / pad-1 forward form (S2); 2: a 2x2-tap stride-1 gather with offsets in {-1, 0, 1}
// (the input gradient of one output-parity class of that layer).  Output image rows of 8 (four images per tile), 16, 32 or 64 pixels.
int g_conv_persist = 1;
int g_last_persist = 0;          // 1: the last dm_conv launch was the persistent form of the halo kernel
int g_tap4 = 1;
int tap4_mode(const ConvP& p) {
    if (!g_tap4 || p.C2 != 0 || p.C1 % 64 != 0 || p.N < 64 || p.B2 != p.B) return 0;
    if (p.Wq == 8) { if (p.Hq != 8 || p.B % 4 != 0) return 0; }
    else if ((p.Wq != 16 && p.Wq != 32 && p.Wq != 64) || (p.Hq * p.Wq) % 256 != 0) return 0;
    const int64_t in_bytes = (int64_t)p.B * p.Hi * p.Wi * p.C1 * 2;
    if (in_bytes >= (1ll << 31) || (int64_t)p.N * p.ldw * 2 >= (1ll << 31)) return 0;
    if (p.T == 16 && p.KW == 4 && p.sy == 2 && p.sx == 2 && p.ty == 1 && p.tx == 1 && p.oy0 == -1 && p.ox0 == -1 && p.Hi == 2 * p.Hq &&
        p.Wi == 2 * p.Wq && p.Ho == p.Hq && p.Wo == p.Wq && p.osy == 1 && p.osx == 1 && p.ooy == 0 && p.oox == 0)
        return 1;
    if (p.T == 4 && p.KW == 2 && p.sy == 1 && p.sx == 1 && p.Hi == p.Hq && p.Wi == p.Wq && (p.ty == 1 || p.ty == -1) && (p.tx == 1 || p.tx == -1)) {
        const int dy0 = p.oy0, dy1 = p.ty + p.oy0, dx0 = p.ox0, dx1 = p.tx + p.ox0;
        if (dy0 < -1 || dy0 > 1 || dy1 < -1 || dy1 > 1 || dx0 < -1 || dx0 > 1 || dx1 < -1 || dx1 > 1) return 0;
        return 2;
    }
    return 0;
}

int g_pw = 1;
// a 1x1 stride-1 convolution over one source with 32 / 64 / 128 reduction channels
bool pw_eligible(const ConvP& p) {
    if (!g_pw || p.T != 1 || p.sy != 1 || p.sx != 1 || p.C2 != 0 || p.B2 != p.B) return false;
    if (p.C1 != 32 && p.C1 != 64 && p.C1 != 128) return false;
    if (p.Hi != p.Hq || p.Wi != p.Wq || p.oy0 != 0 || p.ox0 != 0 || p.N < 32) return false;
    if (((uintptr_t)p.in1 & 15) || ((uintptr_t)p.w & 15) || (p.ldw & 7)) return false;
    return true;
}

template <typename T, int BN>
int launch_pw(const ConvP& p, hipStream_t st) {
    const int64_t grid = (int64_t)cdiv(p.M, BM) * cdiv(p.N, BN);
    if (p.C1 == 32) hipLaunchKernelGGL((conv_pw_kernel<T, BN, 1>), dim3((unsigned)grid), dim3(256), 0, st, p);
    else if (p.C1 == 64) hipLaunchKernelGGL((conv_pw_kernel<T, BN, 2>), dim3((unsigned)grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv_pw_kernel<T, BN, 4>), dim3((unsigned)grid), dim3(256), 0, st, p);
    DM_LAUNCH_CHECK();
    g_last_path = 3;
    return DM_OK;
}

template <typename T>
int launch_conv(const ConvP& p, bool small_offsets, hipStream_t st) {
    if constexpr (sizeof(T) == 2) {
        if (g_variant == 5 && small_offsets && pw_eligible(p)) {
            if (p.N <= 32) return launch_pw<T, 32>(p, st);
            if (p.N <= 64) return launch_pw<T, 64>(p, st);
            return launch_pw<T, 128>(p, st);
        }
        if (g_variant == 5 && small_offsets) {
            const int m4 = tap4_mode(p);
            if (m4 == 1) return launch_tap4_any(p, std::is_same<T, f16>::value, true, st);
            if (m4 == 2) return launch_tap4_any(p, std::is_same<T, f16>::value, false, st);
        }
        if (g_variant == 5 && g_packtap && halo_eligible(p) && packtap_ok(p)) return launch_packtap_any(p, std::is_same<T, f16>::value, st);
        if (g_variant == 5 && g_packtap && small_offsets && narrow_ok(p)) return launch_narrow_any(p, std::is_same<T, f16>::value, st);
        if (g_variant == 5 && halo_eligible(p)) return launch_halo_any(p, std::is_same<T, f16>::value, st);
    }
    const int mblocks = cdiv(p.M, BM);
    int bn = 128;
    if (p.N <= 32) bn = 32;
    else if (p.N <= 64) bn = 64;
    else if ((int64_t)mblocks * cdiv(p.N, 128) < 256) {            // small problems: more, smaller tiles ...
        bn = 64;
        // ... unless the reduction is deep enough for split-K to fill the chip with 128-wide tiles (the 4x4 / stride-2 layer on 8x8
        // maps, K = 16 x 1024): the fill per MFMA is what bounds this kernel, and a 128 x 128 tile needs 1.5x less of it
        static const int deep = getenv("DM_DEEP_BN128") ? atoi(getenv("DM_DEEP_BN128")) : 1;   // (0: A/B measurements; 97 -> 73 us on that layer)
        const int64_t t128 = (int64_t)mblocks * cdiv(p.N, 128);
        const int ks = p.T * cdiv(p.C1 + p.C2, 128 / (int)sizeof(T));
        if (deep && small_offsets && t128 <= 128 && dm_g_ws != nullptr && ks / 4 >= 256 / t128 && (256 / t128) * t128 >= 192) bn = 128;
    }
    const int64_t tiles = (int64_t)mblocks * cdiv(p.N, bn);
    int variant = small_offsets ? g_variant : 1;
    // fewer workgroups than CUs: nothing else hides the load latency, so run the ring 3 steps ahead instead of 1
    if (variant == 5 && tiles <= 256) variant = 4;
    ConvP q = p;
    // a handful of tiles with a long reduction (dense layers on pooled vectors, the 4x4 / 8x8 bottleneck convolutions):
    // split the k-steps over workgroups, partial tiles to the workspace, splitk_epilogue_kernel finishes
    if (variant >= 2 && tiles <= 128 && dm_g_ws != nullptr) {
        const int bk = 128 / (int)sizeof(T);
        const int ksteps = p.T * cdiv(p.C1 + p.C2, bk);
        int splits = (int)(256 / tiles);
        if (splits > ksteps / 4) splits = ksteps / 4;
        if (splits >= 2 && (int64_t)splits * tiles * (128 * bn * 4) <= dm_g_ws_bytes) {
            q.kper = cdiv(ksteps, splits);
            q.splits = cdiv(ksteps, q.kper);
            q.ws = dm_g_ws;
        }
    }
    const int64_t grid = tiles * q.splits;
    int rc;
    if (bn == 128) rc = launch_bn<T, 128>(q, grid, variant, st);
    else if (bn == 64) rc = launch_bn<T, 64>(q, grid, variant, st);
    else rc = launch_bn<T, 32>(q, grid, variant, st);
    if (rc) return rc;
    DM_LAUNCH_CHECK();
    if (q.splits > 1) {
        if (bn == 128) hipLaunchKernelGGL((splitk_epilogue_kernel<T, 128>), dim3((unsigned)tiles), dim3(256), 0, st, q);
        else if (bn == 64) hipLaunchKernelGGL((splitk_epilogue_kernel<T, 64>), dim3((unsigned)tiles), dim3(256), 0, st, q);
        else hipLaunchKernelGGL((splitk_epilogue_kernel<T, 32>), dim3((unsigned)tiles), dim3(256), 0, st, q);
        DM_LAUNCH_CHECK();
    }
    return DM_OK;
}

int launch_splitk_epilogue128(const ConvP& q, bool is_f16, unsigned grid, hipStream_t st) {
    if (is_f16) hipLaunchKernelGGL((splitk_epilogue_kernel<f16, 128>), dim3(grid), dim3(256), 0, st, q);
    else hipLaunchKernelGGL((splitk_epilogue_kernel<bf16, 128>), dim3(grid), dim3(256), 0, st, q);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

}  // namespace dmk
using namespace dmk;

/* 1 (default): 3x3 launches with more tiles than CUs run as persistent workgroups with cross-tile prefetch (igemm_halo_p.hip) */
extern "C" int dm_set_conv_persist(int on) { dmk::g_conv_persist = on ? 1 : 0; return DM_OK; }

extern "C" int dm_set_conv_tap4(int on) { g_tap4 = (on & 1) != 0; g_pw = (on & 2) == 0; return DM_OK; }   // bit 1 set: pointwise kernel off too
extern "C" int dm_set_splitk_inkernel(int on) { g_splitk_inkernel = on != 0; return DM_OK; }

extern "C" int dm_set_conv_variant(int variant) {
    DM_CHECK_ARG(variant >= 1 && variant <= 5, "dm_set_conv_variant: 1 (register staging), 2..4 (LDS-DMA ring stages) or 5 (2 + halo-resident 3x3)");
    g_variant = variant;
    return DM_OK;
}

extern "C" int dm_last_conv_path(void) { return g_last_path; }
extern "C" int dm_set_conv_packtap(int on) { g_packtap = on ? 1 : 0; return DM_OK; }
extern "C" int dm_last_conv_persistent(void) { return dmk::g_last_persist; }
extern "C" int dm_get_conv_variant(void) { return g_variant; }

static int conv_fill(const DmConv* d, ConvP& p, bool& small);

// Four descriptors that differ only in w, oy0, ox0, ooy, oox — the input-gradient launches of the four output-parity classes of a
// stride-2 layer: ONE launch of the four-tap halo kernel when that is eligible and fills the chip without a K split, otherwise four
// dm_conv calls.
extern "C" int dm_conv_parity4(const DmConv* d4, dm_stream_t stream) {
    DM_CHECK_ARG(d4 != nullptr, "dm_conv_parity4: null descriptors");
    ConvP p[4];
    bool small[4];
    for (int i = 0; i < 4; ++i) {
        const int rc = conv_fill(d4 + i, p[i], small[i]);
        if (rc != DM_OK) return rc;
    }
    bool same = d4[0].dtype != DM_F32 && g_variant == 5 && small[0];
    for (int i = 1; i < 4 && same; ++i) {
        const DmConv &a = d4[0], &b = d4[i];
        same = a.in1 == b.in1 && a.in2 == b.in2 && a.out == b.out && a.addend == b.addend && a.scale == b.scale && a.shift == b.shift &&
               a.psum == nullptr && b.psum == nullptr && a.dtype == b.dtype && a.act == b.act && a.out_nchw_f32 == 0 && b.out_nchw_f32 == 0 &&
               a.B == b.B && a.Hi == b.Hi && a.Wi == b.Wi && a.C1 == b.C1 && a.C2 == b.C2 && a.Hq == b.Hq && a.Wq == b.Wq && a.sy == b.sy &&
               a.sx == b.sx && a.T == b.T && a.KW == b.KW && a.ty == b.ty && a.tx == b.tx && a.Ho == b.Ho && a.Wo == b.Wo && a.osy == b.osy &&
               a.osx == b.osx && a.N == b.N && a.ldw == b.ldw && a.ldc == b.ldc && a.coff == b.coff && a.in2_batch == b.in2_batch;
    }
    if (same) for (int i = 0; i < 4 && same; ++i) same = tap4_mode(p[i]) == 2;
    const int tiles1 = (int)((p[0].M / 256) * cdiv(p[0].N, 128));
    if (same && 4 * tiles1 >= 192) {               // the four classes together fill the chip: no K split, one launch
        ConvP q = p[0];
        q.npar = 4;
        for (int i = 0; i < 4; ++i) {
            q.w4[i] = p[i].w; q.oy4[i] = p[i].oy0; q.ox4[i] = p[i].ox0; q.ooy4[i] = p[i].ooy; q.oox4[i] = p[i].oox;
        }
        hipStream_t st = (hipStream_t)stream;
        return launch_tap4_any(q, d4[0].dtype == DM_F16, false, st);
    }
    for (int i = 0; i < 4; ++i) {
        const int rc = dm_conv(d4 + i, stream);
        if (rc != DM_OK) return rc;
    }
    return DM_OK;
}

static int conv_fill(const DmConv* d, ConvP& p_out, bool& small_out) {
    DM_CHECK_ARG(d != nullptr, "dm_conv: null descriptor");
    const int ve = d->dtype == DM_F32 ? 4 : 8;
    DM_CHECK_ARG(d->dtype == DM_F32 || d->dtype == DM_BF16 || d->dtype == DM_F16, "dm_conv: bad dtype %d", d->dtype);
    DM_CHECK_ARG(d->in1 && d->w && d->out, "dm_conv: null tensor pointer");
    DM_CHECK_ARG(d->C1 > 0 && d->C1 % ve == 0 && d->C2 >= 0 && d->C2 % ve == 0, "dm_conv: C1=%d C2=%d must be multiples of %d", d->C1, d->C2, ve);
    DM_CHECK_ARG(d->C2 == 0 || d->in2, "dm_conv: C2 > 0 but in2 is null");
    DM_CHECK_ARG(d->ldw % ve == 0 && d->ldw >= d->T * (d->C1 + d->C2), "dm_conv: ldw=%d invalid for T=%d C=%d", d->ldw, d->T, d->C1 + d->C2);
    DM_CHECK_ARG(d->B > 0 && d->Hi > 0 && d->Wi > 0 && d->Hq > 0 && d->Wq > 0 && d->T > 0 && d->T <= 64 && d->KW > 0 && d->N > 0, "dm_conv: bad extent (T must be 1..64)");
    DM_CHECK_ARG((d->Hq - 1) * d->osy + d->ooy < d->Ho && (d->Wq - 1) * d->osx + d->oox < d->Wo && d->ooy >= 0 && d->oox >= 0, "dm_conv: output mapping exceeds Ho/Wo");
    DM_CHECK_ARG(d->out_nchw_f32 || (d->ldc >= d->coff + d->N && d->coff >= 0), "dm_conv: ldc=%d < coff+N=%d", d->ldc, d->coff + d->N);
    DM_CHECK_ARG((d->psum == nullptr) == (d->psq == nullptr), "dm_conv: psum/psq must both be set or both null");
    DM_CHECK_ARG(((uintptr_t)d->in1 & 15) == 0 && ((uintptr_t)d->in2 & 15) == 0 && ((uintptr_t)d->w & 15) == 0, "dm_conv: tensors must be 16-byte aligned");
    const int64_t M = (int64_t)d->B * d->Hq * d->Wq;
    DM_CHECK_ARG(M < (1ll << 31), "dm_conv: M too large");
    ConvP& p = p_out;
    p.in1 = (const char*)d->in1; p.in2 = (const char*)d->in2; p.w = (const char*)d->w;
    p.scale = d->scale; p.shift = d->shift; p.out = (char*)d->out; p.psum = d->psum; p.psq = d->psq;
    p.act = d->act; p.out_nchw = d->out_nchw_f32;
    p.B = d->B; p.Hi = d->Hi; p.Wi = d->Wi; p.C1 = d->C1; p.C2 = d->C2; p.Hq = d->Hq; p.Wq = d->Wq; p.sy = d->sy; p.sx = d->sx;
    p.T = d->T; p.KW = d->KW; p.ty = d->ty; p.tx = d->tx; p.oy0 = d->oy0; p.ox0 = d->ox0;
    p.Ho = d->Ho; p.Wo = d->Wo; p.osy = d->osy; p.osx = d->osx; p.ooy = d->ooy; p.oox = d->oox;
    p.N = d->N; p.ldw = d->ldw; p.ldc = d->ldc; p.coff = d->coff; p.M = (int)M;
    p.splits = 1; p.kper = 1 << 24; p.ws = nullptr; p.counters = nullptr; p.npar = 1;
    p.B2 = d->in2_batch > 0 ? d->in2_batch : d->B;
    p.addend = (const char*)d->addend;
    p.stat_slots = d->stat_slots;
    DM_CHECK_ARG(d->stat_slots >= 0 && d->stat_slots <= 64 && (d->stat_slots == 0 || (d->psum && (((uintptr_t)d->psum | (uintptr_t)d->psq) & 7) == 0)),
                 "dm_conv: stat_slots=%d needs 8-byte aligned psum / psq (and must be <= 64)", d->stat_slots);
    DM_CHECK_ARG(d->addend == nullptr || !d->out_nchw_f32, "dm_conv: addend is not supported with the NCHW fp32 output");
    if (p.B2 != p.B) {                           // broadcast second source: the halo-resident kernel only
        DM_CHECK_ARG(d->C2 > 0 && d->B % p.B2 == 0, "dm_conv: in2_batch=%d must divide B=%d", p.B2, d->B);
        DM_CHECK_ARG(d->dtype != DM_F32 && g_variant >= 5 && halo_eligible(p),
                     "dm_conv: a broadcast second source (in2_batch) needs the halo-resident 3x3 kernel (bf16, stride 1, 16/32/64-pixel rows)");
    }
    // the LDS-DMA kernel indexes with unsigned 32-bit element offsets (with a margin for the halo arithmetic)
    const int64_t in_elems = (int64_t)d->B * d->Hi * d->Wi * (d->C1 > d->C2 ? d->C1 : d->C2);
    const int64_t w_elems = (int64_t)d->N * d->ldw;
    const bool small = in_elems < (1ll << 31) - (1ll << 24) && w_elems < (1ll << 31);
    small_out = small;
    return DM_OK;
}

extern "C" int dm_conv(const DmConv* d, dm_stream_t stream) {
    g_last_path = 0;
    g_last_persist = 0;
    ConvP p;
    bool small;
    const int rc = conv_fill(d, p, small);
    if (rc != DM_OK) return rc;
    if (d->dtype == DM_BF16) return launch_conv<bf16>(p, small, (hipStream_t)stream);
    if (d->dtype == DM_F16) return launch_conv<f16>(p, small, (hipStream_t)stream);
    return launch_conv<float>(p, small, (hipStream_t)stream);
}
