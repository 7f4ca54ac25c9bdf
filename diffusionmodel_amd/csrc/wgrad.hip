// Weight gradient of the gather convolution on the matrix cores.
//
//   dw[n][t][c] += sum_m dy[orow(m)][n] * in(pix(m,t))[c]
//
// Two kernel families: wgrad3x3_halo_kernel (further down) for the 3x3 stride-1 layers — all nine taps from one staged
// input halo, split partials through the workspace — and the per-tap kernels described here for everything else.
//
// GEMM view per tap t: D[i = n][j = c], reduction index k = output pixel m.  In NHWC memory the
// reduction index is the SLOW index of both operands (rows are pixels), while an MFMA fragment wants
// consecutive k per lane, so both operands are staged as [pixel][channel] images in LDS and read
// TRANSPOSED:  bf16 with ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group, delivered
// channel-per-lane), f32 with plain ds_read_b32 (one k per lane, no contiguity needed).
// k assignment inside a 32-pixel step (bf16): lane group g holds pixels {4g..4g+3} and {16+4g..16+4g+3};
// A and B use the same assignment, so the MFMA sums every pixel exactly once.  Row stride is padded
// by 32 B (bf16) / 64 B (f32): the 8 rows a 32-lane half touches per instruction then fall on
// disjoint banks.
//
// Grid: x = 128-wide n tiles, y = taps x 128-wide c tiles, z = split over pixels (fp32 atomics
// combine the splits — and accumulate into whatever dw already holds, which is how gradient
// accumulation over micro-batches comes for free).  Blocks with y == 0 also produce dbias.
#include <cstdlib>
#include "common.h"
// Where the halo weight-gradient kernel issues the next tile's LDS-DMA pieces: 0 = all right after the tile's barrier; 0xAABB = the dy
// pieces at slot AA, the halo pieces at slot BB, slot = 16 k-step + tap (issued before that tap's MFMAs).
#ifndef DM_WGRAD_DMA_POS
#define DM_WGRAD_DMA_POS 0x1111
#endif

#ifndef DM_WGRAD_NOFLUSH
#define DM_WGRAD_NOFLUSH 0
#endif

namespace {

struct WgP {
    const char* dy; const char* in1; const char* in2; float* dw; float* dbias;
    int B, Hi, Wi, C1, C2, Hq, Wq, sy, sx, T, KW, ty, tx, oy0, ox0;
    int Ho, Wo, osy, osx, ooy, oox, N, ldy, ldw, M, csteps_c, steps_per_split, total_steps;
    int gx, gy, gz;   // logical grid of the v2 kernel (launched 1-D, see the XCD mapping there)
    int rmw;          // v2: 1 = no pixel split -> plain read-modify-write of dw instead of atomics
};

template <typename T> struct WgCfg;
template <> struct WgCfg<bf16> { static constexpr int KD = 32, STRIDE = 288; };
template <> struct WgCfg<f16> { static constexpr int KD = 32, STRIDE = 288; };
template <> struct WgCfg<float> { static constexpr int KD = 16, STRIDE = 576; };

// 16-bit MFMA on 128-bit fragments (kept as bf16x8 containers; fp16 reinterprets the same bits)
template <typename T> struct WMma;
template <> struct WMma<bf16> {
    __device__ static __forceinline__ f32x4 run(const bf16x8& a, const bf16x8& b, const f32x4& c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct WMma<f16> {
    __device__ static __forceinline__ f32x4 run(const bf16x8& a, const bf16x8& b, const f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};
template <> struct WMma<float> {      // never called: the fp32 kernels take the v_mfma_f32_16x16x4_f32 branch
    __device__ static __forceinline__ f32x4 run(const bf16x8&, const bf16x8&, const f32x4& c) { return c; }
};

__device__ inline bf16x8 tr_frag(const char* tile, int stride, int col_base, int lane) {
    // 16x16x32 operand fragment for 16 channels starting at col_base, transposed read.
    const int g = lane >> 4, il = lane & 15, q = il >> 2, pp = il & 3;
    const char* a0 = tile + (4 * g + q) * stride + (col_base + 4 * pp) * 2;
    const char* a1 = a0 + 16 * stride;
    typedef __attribute__((address_space(3))) s16x4* lds_p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a1));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

template <typename T>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgP p) {
    constexpr int VE = Elem<T>::VE;
    constexpr int KD = WgCfg<T>::KD, STRIDE = WgCfg<T>::STRIDE;
    constexpr int VPR = 128 / VE;         // vectors per row
    constexpr int RPP = 256 / VPR;        // rows covered per pass
    constexpr int TILE = KD * STRIDE;
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave & 1, wc = wave >> 1;
    const int C = p.C1 + p.C2;
    const int n0 = blockIdx.x * 128;
    const int t = blockIdx.y / p.csteps_c, c0 = (blockIdx.y - t * p.csteps_c) * 128;
    const int ky = t / p.KW, kx = t - ky * p.KW;
    const int dyo = ky * p.ty + p.oy0, dxo = kx * p.tx + p.ox0;
    const T* dy = (const T*)p.dy;
    const T* in1 = (const T*)p.in1;
    const T* in2 = (const T*)p.in2;

    const int svec = tid % VPR, srow = tid / VPR;
    const int step_lo = blockIdx.z * p.steps_per_split;
    const int step_hi = min(step_lo + p.steps_per_split, p.total_steps);

    u32x4 ra[2], rb[2];
    auto gload = [&](int step) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = srow + RPP * i;
            const int m = step * KD + row;
            u32x4 va = {0u, 0u, 0u, 0u}, vb = {0u, 0u, 0u, 0u};
            if (m < p.M) {
                const int qx = m % p.Wq, tq = m / p.Wq;
                const int qy = tq % p.Hq, b = tq / p.Hq;
                const int n = n0 + svec * VE;
                if (n + VE <= p.ldy) {
                    const size_t orow = ((size_t)b * p.Ho + qy * p.osy + p.ooy) * p.Wo + qx * p.osx + p.oox;
                    va = *(const u32x4*)(dy + orow * p.ldy + n);
                }
                const int iy = qy * p.sy + dyo, ix = qx * p.sx + dxo;
                const int c = c0 + svec * VE;
                if ((unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi && c < C) {
                    const size_t pix = ((size_t)b * p.Hi + iy) * p.Wi + ix;
                    const T* src = (c < p.C1) ? (in1 + pix * p.C1 + c) : (in2 + pix * p.C2 + (c - p.C1));
                    vb = *(const u32x4*)src;
                }
            }
            ra[i] = va;
            rb[i] = vb;
        }
    };
    auto sstore = [&](int buf) {
        char* sA = smem + buf * 2 * TILE;
        char* sB = sA + TILE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int off = (srow + RPP * i) * STRIDE + svec * 16;
            *(u32x4*)(sA + off) = ra[i];
            *(u32x4*)(sB + off) = rb[i];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bias_acc = 0.f;
    const bool do_bias = p.dbias != nullptr && blockIdx.y == 0 && tid < 128;

    if (step_lo < step_hi) {
        gload(step_lo);
        sstore(0);
    }
    __syncthreads();
    for (int s = step_lo; s < step_hi; ++s) {
        const int cur = (s - step_lo) & 1;
        const bool more = s + 1 < step_hi;
        if (more) gload(s + 1);
        const char* sA = smem + cur * 2 * TILE;
        const char* sB = sA + TILE;
        if constexpr (sizeof(T) == 2) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) fa[it] = tr_frag(sA, STRIDE, wn * 64 + it * 16, lane);
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) fb[jt] = tr_frag(sB, STRIDE, wc * 64 + jt * 16, lane);
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int jt = 0; jt < 4; ++jt)
                    acc[it][jt] = WMma<T>::run(fa[it], fb[jt], acc[it][jt]);
        } else {
            const int g = lane >> 4, il = lane & 15;
#pragma unroll
            for (int ss = 0; ss < KD / 4; ++ss) {
                float fa[4], fb[4];
                const int roff = (4 * ss + g) * STRIDE;
#pragma unroll
                for (int it = 0; it < 4; ++it) fa[it] = *(const float*)(sA + roff + (wn * 64 + it * 16 + il) * 4);
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) fb[jt] = *(const float*)(sB + roff + (wc * 64 + jt * 16 + il) * 4);
#pragma unroll
                for (int it = 0; it < 4; ++it)
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt)
                        acc[it][jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[it], fb[jt], acc[it][jt], 0, 0, 0);
            }
        }
        if (do_bias) {
#pragma unroll 8
            for (int r = 0; r < KD; ++r) bias_acc += Elem<T>::ld((const T*)(sA + r * STRIDE) + tid);
        }
        if (more) sstore(cur ^ 1);
        __syncthreads();
    }

    const int g = lane >> 4, il = lane & 15;
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + wn * 64 + it * 16 + 4 * g + r;
            if (n >= p.N) continue;
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                const int c = c0 + wc * 64 + jt * 16 + il;
                if (c < C) atomicAdd(p.dw + (size_t)n * p.ldw + (size_t)t * C + c, acc[it][jt][r]);
            }
        }
    if (do_bias && n0 + tid < p.N) atomicAdd(p.dbias + n0 + tid, bias_acc);
}

// =================================================================================================
// v2: LDS-DMA staging.  Stage = [KD pixels][128 channels] images of dy and of the tap-shifted input,
// rows unpadded (so a 1-KiB DMA piece = 4 (bf16) / 2 (f32) whole rows) and XOR-swizzled in chunks of
// 32 B (bf16) / 64 B (f32): chunk ^= row & 7 — the 8 pixel rows a 32-lane half reads in one transposed
// read then sit in 8 disjoint bank groups.  As in the forward kernel the swizzle is applied on the DMA
// SOURCE side (lane -> logical vector) and padding comes from a zero page.  KD = 64 (bf16) / 32 (f32)
// pixels per barrier, 2-stage ring, 2 workgroups per CU.  Requires the identity output mapping
// (dy row of pixel m is row m), which is every call the model makes.
// =================================================================================================
__device__ __attribute__((aligned(128))) unsigned int g_wg_zero_page[64];
typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* gbl_vptr;

template <typename T> struct Wg2Cfg;
template <> struct Wg2Cfg<bf16> { static constexpr int KD = 64, RB = 256, SPC = 2, RPP = 4; };   // rows per piece
template <> struct Wg2Cfg<f16> { static constexpr int KD = 64, RB = 256, SPC = 2, RPP = 4; };
template <> struct Wg2Cfg<float> { static constexpr int KD = 32, RB = 512, SPC = 4, RPP = 2; };

template <typename T>
__device__ __forceinline__ int wg2_off(int row, int elem) {   // byte offset of element `elem` (0..127) of pixel row `row`
    constexpr int RB = Wg2Cfg<T>::RB, CH = Wg2Cfg<T>::SPC * 16, EPC = CH / (int)sizeof(T);   // elements per swizzle chunk
    return row * RB + (((elem / EPC) ^ (row & 7)) * CH) + (elem % EPC) * (int)sizeof(T);
}

__device__ __forceinline__ bf16x8 tr_frag2(const char* tile, int row0, int col_base, int lane) {
    // 16x16x32 operand fragment: lane group g holds pixels row0 + {4g..4g+3} and row0 + 16 + {4g..4g+3}
    const int g = lane >> 4, il = lane & 15, q = il >> 2, pp = il & 3;
    const int r0 = row0 + 4 * g + q, r1 = r0 + 16;
    typedef __attribute__((address_space(3))) s16x4* lds_p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(tile + wg2_off<bf16>(r0, col_base) + 8 * pp));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(tile + wg2_off<bf16>(r1, col_base) + 8 * pp));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

template <typename T>
__global__ __launch_bounds__(256) void conv_wgrad2_kernel(const WgP p) {
    constexpr int VE = Elem<T>::VE;
    constexpr int KD = Wg2Cfg<T>::KD, RB = Wg2Cfg<T>::RB, SPC = Wg2Cfg<T>::SPC, RPP = Wg2Cfg<T>::RPP;
    constexpr int TILE = KD * RB, STAGE = 2 * TILE;          // 16 KiB per operand tile, 32 KiB per stage
    constexpr int SLOTS = RB / 16;                            // 16-B slots per row
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 * STAGE, the only LDS object

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wc = wave >> 1;
    const int C = p.C1 + p.C2;
    // XCD-aware decode of the 1-D grid: workgroups are dealt round-robin over the 8 XCDs, so with
    // id = xcd + 8*(tile + ntiles*zhi) and pixel-split z = 8*zhi + xcd, ALL tap / channel tiles of one pixel
    // split run on ONE XCD back to back: the 9 taps re-read the same dy / input rows out of that L2.
    const int ntile = p.gx * p.gy;
    // (only when there are >= 8 pixel splits; with fewer, a plain decode keeps all XCDs busy)
    const int lin = blockIdx.x;
    int tile, bz;
    if (p.gz >= 8) {
        const int xcd = lin & 7, rest = lin >> 3;
        tile = rest % ntile;
        bz = (rest / ntile) * 8 + xcd;
    } else {
        tile = lin % ntile;
        bz = lin / ntile;
    }
    if (bz >= p.gz) return;
    const int bx = tile % p.gx, by = tile / p.gx;
    const int n0 = bx * 128;
    const int t = by / p.csteps_c, c0 = (by - t * p.csteps_c) * 128;
    const int ky = t / p.KW, kx = t - ky * p.KW;
    const int dyo = ky * p.ty + p.oy0, dxo = kx * p.tx + p.ox0;
    unsigned long long a_dy = (unsigned long long)p.dy, a_in1 = (unsigned long long)p.in1;
    unsigned long long a_in2 = (unsigned long long)(p.in2 ? p.in2 : p.in1), a_zero = (unsigned long long)g_wg_zero_page;
    asm volatile("" : "+s"(a_dy), "+s"(a_in1), "+s"(a_in2), "+s"(a_zero));

    const int step_lo = bz * p.steps_per_split;
    const int step_hi = min(step_lo + p.steps_per_split, p.total_steps);

    // ---- per-lane DMA geometry: a piece = RPP rows; lane -> (row in piece, physical slot) -> logical vector
    const int lr = lane / SLOTS, ps = lane % SLOTS;
    int prow[4];                 // the lane's pixel row (within the stage) for each of the wave's 4 pieces per tile
    int lvec[4];                 // logical 16-B vector index it must fetch for that row (swizzle on the source side)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        prow[j] = (wave * 4 + j) * RPP + lr;
        lvec[j] = (((ps / SPC) ^ (prow[j] & 7)) * SPC) | (ps % SPC);
    }
    // tracked pixel coordinates of the 4 rows (advanced by KD pixels per step without divisions)
    int tb[4], tyq[4], txq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = step_lo * KD + prow[j];
        txq[j] = m % p.Wq;
        const int tq = m / p.Wq;
        tyq[j] = tq % p.Hq;
        tb[j] = tq / p.Hq;
    }
    const int dqx = KD % p.Wq, dq1 = KD / p.Wq, dqy = dq1 % p.Hq, dqb = dq1 / p.Hq;

    int next_step = step_lo;
    auto issue = [&](int stage) {
        char* sA = smem + stage * STAGE;
        char* sB = sA + TILE;
        const int mbase = next_step * KD;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = mbase + prow[j];
            const bool m_ok = m < p.M;
            // dy piece: row m, channels n0 + lvec*VE ...
            const int n = n0 + lvec[j] * VE;
            unsigned long long ga = a_dy + ((unsigned long long)(unsigned)m * (unsigned)p.ldy + (unsigned)n) * sizeof(T);
            ga = (m_ok && n + VE <= p.ldy) ? ga : a_zero;
            __builtin_amdgcn_global_load_lds((gbl_vptr)ga, (lds_vptr)(sA + (wave * 4 + j) * 1024), 16, 0, 0);
            // input piece: tap-shifted pixel, channels c0 + lvec*VE ...
            const int iy = tyq[j] * p.sy + dyo, ix = txq[j] * p.sx + dxo;
            const int c = c0 + lvec[j] * VE;
            const bool ok = m_ok && c < C && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            const bool first = c < p.C1;
            const unsigned pix = (unsigned)((tb[j] * p.Hi + iy) * p.Wi + ix);
            unsigned long long gb = (first ? a_in1 : a_in2) +
                                    ((unsigned long long)pix * (unsigned)(first ? p.C1 : p.C2) + (unsigned)(first ? c : c - p.C1)) * sizeof(T);
            gb = ok ? gb : a_zero;
            __builtin_amdgcn_global_load_lds((gbl_vptr)gb, (lds_vptr)(sB + (wave * 4 + j) * 1024), 16, 0, 0);
            // advance this row by KD pixels
            int x = txq[j] + dqx, y = tyq[j] + dqy, b = tb[j] + dqb;
            if (x >= p.Wq) { x -= p.Wq; ++y; }
            if (y >= p.Hq) { y -= p.Hq; ++b; }
            txq[j] = x; tyq[j] = y; tb[j] = b;
        }
        ++next_step;
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bias_acc = 0.f;
    const bool do_bias = p.dbias != nullptr && by == 0 && tid < 128;

    const int nsteps = step_hi - step_lo;
    if (nsteps > 0) issue(0);
    for (int s = 0; s < nsteps; ++s) {
        const int cur = s & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();            // step s of every wave has landed; stage cur^1 is free again
        if (s + 1 < nsteps) issue(cur ^ 1);
        const char* sA = smem + cur * STAGE;
        const char* sB = sA + TILE;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                bf16x8 fa[4], fb[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) fa[it] = tr_frag2(sA, sub * 32, wn * 64 + it * 16, lane);
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) fb[jt] = tr_frag2(sB, sub * 32, wc * 64 + jt * 16, lane);
#pragma unroll
                for (int it = 0; it < 4; ++it)
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt)
                        acc[it][jt] = WMma<T>::run(fa[it], fb[jt], acc[it][jt]);
            }
        } else {
            const int g = lane >> 4, il = lane & 15;
#pragma unroll
            for (int ss = 0; ss < KD / 4; ++ss) {
                float fa[4], fb[4];
                const int row = 4 * ss + g;
#pragma unroll
                for (int it = 0; it < 4; ++it) fa[it] = *(const float*)(sA + wg2_off<float>(row, wn * 64 + it * 16 + il));
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) fb[jt] = *(const float*)(sB + wg2_off<float>(row, wc * 64 + jt * 16 + il));
#pragma unroll
                for (int it = 0; it < 4; ++it)
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt)
                        acc[it][jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[it], fb[jt], acc[it][jt], 0, 0, 0);
            }
        }
        if (do_bias) {
#pragma unroll 8
            for (int r = 0; r < KD; ++r) bias_acc += Elem<T>::ld((const T*)(sA + wg2_off<T>(r, tid)));
        }
    }

    const int g = lane >> 4, il = lane & 15;
    if (p.rmw) {
        // no pixel split (the two 16.7 M-parameter layers of the bottleneck: 1024 tiles already): this workgroup is the only writer
        // of its dw elements within the launch — plain read-modify-write instead of 16.7 M fp32 atomics
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * 64 + it * 16 + 4 * g + r;
                if (n >= p.N) continue;
                float* row = p.dw + (size_t)n * p.ldw + (size_t)t * C + c0 + wc * 64 + il;
                float old[4];
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) old[jt] = c0 + wc * 64 + jt * 16 + il < C ? row[jt * 16] : 0.f;
#pragma unroll
                for (int jt = 0; jt < 4; ++jt)
                    if (c0 + wc * 64 + jt * 16 + il < C) row[jt * 16] = old[jt] + acc[it][jt][r];
            }
    } else {
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * 64 + it * 16 + 4 * g + r;
                if (n >= p.N) continue;
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) {
                    const int c = c0 + wc * 64 + jt * 16 + il;
                    if (c < C) atomicAdd(p.dw + (size_t)n * p.ldw + (size_t)t * C + c, acc[it][jt][r]);
                }
            }
    }
    if (do_bias && n0 + tid < p.N) atomicAdd(p.dbias + n0 + tid, bias_acc);
}

// =================================================================================================
// v4: weight gradient of the 1x1 layers (bf16 / fp16)
// =================================================================================================
// dw[n][c] = sum_m dy[m][n] * in[m][c] with one of N, C at most 128: 1-4 MFLOP per KiB read, so the launch is a stream of
// dy and the input through the chip (17 us of HBM time for the 64^2 layers at n_feat = 128) and the per-tap kernel above
// spends it on 128 x 128 tiles that are 75 % padding at C = 32.  Here a 256-thread workgroup owns an output block
// [BIG = 64*NTW channels of the wider operand] x [SMALL = 16*CTW = ALL channels of the narrower one] and walks over PX-pixel
// chunks of its pixel range: 16-byte global loads into registers (8-10 per thread, issued one chunk ahead), ds_write_b128
// into two [pixel][channel] images whose rows are padded by 32 B (the 8 rows a 32-lane half of ds_read_b64_tr_b16 touches
// then fall on disjoint banks for any row width), transposed fragment reads as in the kernels above, NTW x CTW MFMAs per
// wave per 32 pixels.  <= 49 KiB of LDS: up to three workgroups per CU.  Wave w takes the wider operand's tiles
// [w*NTW, (w+1)*NTW).  SWAP says which operand is the wide one: false = dy (wide N, rows of dw), true = the input (wide C,
// columns of dw) — then the MFMA operands trade places, so that the 16 lanes of a result row always hold 16 consecutive
// floats of a dw row.  The pixel splits meet in dw through fp32 atomics like the per-tap kernel (which also makes the
// launch accumulate into dw); measured: ~400 G atomics/s when a wave instruction covers whole 64-byte runs, 65 G/s when
// its lanes are a row apart (the first version of the SWAP case), so the split count is what the launch pays for: the
// workgroup count is chosen for a fixed volume of atomics (384 workgroups of 4096 floats, 96 of 16384).
struct WgPwP {
    const char* big; const char* small; float* dw; float* dbias;
    int M, ld_big, ld_small, ldw;     // row strides in elements
    int nblocks, splits, chunks_per_split, nchunks, swap;
};

template <typename T, int NTW, int CTW, bool SWAP>
__global__ __launch_bounds__(256) void wgrad_pw_kernel(const WgPwP p) {
    constexpr int BIG = 64 * NTW, SMALL = 16 * CTW;
    constexpr int PX = (BIG + SMALL) <= 160 ? 128 : 64;
    constexpr int SB = BIG * 2 + 32, SS = SMALL * 2 + 32;       // padded row strides in bytes
    constexpr int VB = BIG / 8, VS = SMALL / 8;                 // 16-byte vectors per row
    constexpr int NVB = PX * VB / 256, NVS = PX * VS / 256;     // vectors per thread per chunk
    static_assert(PX * VB % 256 == 0 && PX * VS % 256 == 0, "whole passes of the 256 threads");
    __shared__ __attribute__((aligned(16))) char smem[PX * (SB + SS)];
    char* const img_b = smem;
    char* const img_s = smem + PX * SB;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int blk = blockIdx.x % p.nblocks, sp = blockIdx.x / p.nblocks;
    const char* gb = p.big + (size_t)blk * BIG * 2;
    const char* gs = p.small;
    const int ch_lo = sp * p.chunks_per_split;
    const int ch_hi = min(ch_lo + p.chunks_per_split, p.nchunks);

    uint4 rb[NVB], rs[NVS];
    auto fetch = [&](int chunk) {
        const int m0 = chunk * PX;
#pragma unroll
        for (int i = 0; i < NVB; ++i) {
            const int v = tid + i * 256, row = v / VB, col = v % VB, m = m0 + row;
            rb[i] = m < p.M ? *(const uint4*)(gb + ((size_t)m * p.ld_big + col * 8) * 2) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NVS; ++i) {
            const int v = tid + i * 256, row = v / VS, col = v % VS, m = m0 + row;
            rs[i] = m < p.M ? *(const uint4*)(gs + ((size_t)m * p.ld_small + col * 8) * 2) : make_uint4(0, 0, 0, 0);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < NVB; ++i) {
            const int v = tid + i * 256, row = v / VB, col = v % VB;
            *(uint4*)(img_b + row * SB + col * 16) = rb[i];
        }
#pragma unroll
        for (int i = 0; i < NVS; ++i) {
            const int v = tid + i * 256, row = v / VS, col = v % VS;
            *(uint4*)(img_s + row * SS + col * 16) = rs[i];
        }
    };

    f32x4 acc[NTW][CTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int j = 0; j < CTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // dbias = column sums of the dy image: every block when dy is the wide operand, block 0 when it is the narrow one
    constexpr int dy_ch = SWAP ? SMALL : BIG;
    const bool do_bias = p.dbias != nullptr && tid < dy_ch && (!SWAP || blk == 0);
    const char* bias_col = (SWAP ? img_s : img_b) + tid * 2;
    constexpr int bias_stride = SWAP ? SS : SB;
    float bias_acc = 0.f;

    if (ch_lo < ch_hi) fetch(ch_lo);
    for (int ch = ch_lo; ch < ch_hi; ++ch) {
        stash();                                   // waits for the chunk's loads
        __syncthreads();
        if (ch + 1 < ch_hi) fetch(ch + 1);         // in flight under the MFMAs and the next barrier
#pragma unroll
        for (int ks = 0; ks < PX / 32; ++ks) {
            bf16x8 fa[NTW], fb[CTW];
#pragma unroll
            for (int i = 0; i < NTW; ++i) fa[i] = tr_frag(img_b + ks * 32 * SB, SB, (wave * NTW + i) * 16, lane);
#pragma unroll
            for (int j = 0; j < CTW; ++j) fb[j] = tr_frag(img_s + ks * 32 * SS, SS, j * 16, lane);
#pragma unroll
            for (int i = 0; i < NTW; ++i)
#pragma unroll
                for (int j = 0; j < CTW; ++j)      // rows of the MFMA result = dy channels n, columns = input channels c, either way
                    acc[i][j] = SWAP ? WMma<T>::run(fb[j], fa[i], acc[i][j]) : WMma<T>::run(fa[i], fb[j], acc[i][j]);
        }
        if (do_bias) {
#pragma unroll 8
            for (int r = 0; r < PX; ++r) bias_acc += Elem<T>::ld((const T*)(bias_col + r * bias_stride));
        }
        __syncthreads();                           // every wave is done with the images
    }

    const int g = lane >> 4, il = lane & 15;
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int j = 0; j < CTW; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {          // 16 lanes -> 16 consecutive floats of one dw row
                const int bt = blk * BIG + (wave * NTW + i) * 16, st = j * 16;
                const size_t at = SWAP ? (size_t)(st + 4 * g + r) * p.ldw + bt + il : (size_t)(bt + 4 * g + r) * p.ldw + st + il;
                atomicAdd(p.dw + at, acc[i][j][r]);
            }
    if (do_bias) atomicAdd(p.dbias + (SWAP ? tid : blk * BIG + tid), bias_acc);
}

// =================================================================================================
// v3: halo-resident weight gradient of the 3x3 stride-1 layers (bf16)
// =================================================================================================
// The kernels above give every (n tile, tap, channel tile) its own workgroups, so the nine taps re-stage
// the same dy and input rows nine times and the L2->LDS fill (32 KiB per 64-pixel step for 2.1 MFLOP)
// bounds them.  Here a 512-thread workgroup owns the output block dw[128 n][9 taps][64 c] — 144 fp32
// accumulators per lane — and walks over 128-pixel tiles (whole image rows): per tile it stages the
// dy rows (32 KiB) and the INPUT HALO of one 64-channel chunk (<= 36 KiB) once and feeds all nine
// taps from the halo through shifted transposed reads: 68 KiB of fill per 18.9 MFLOP.
//   * both operands are read with ds_read_b64_tr_b16 (k = pixel is the row index of both images);
//     lane group g takes pixels {4g..4g+3} and {16+4g..} of a 32-pixel k-step on both operands.
//   * dy image: 256-B rows, 32-B chunk ch stored at ch ^ (row & 7); halo image: 128-B rows, chunk ch at
//     ch ^ ((row >> 1) & 3): the 8 rows a 32-lane half touches fall on disjoint banks for ANY first row,
//     so the tap shifts (whole rows: multiples of HS = TW + 8; columns: +0/1/2) stay conflict-free and
//     every read address is one of 4 (dy) + 3 (halo, per kx) VGPRs plus an immediate.
//   * staging: `buffer_load_dwordx4 ... lds`, swizzle on the source address, out-of-image lanes read
//     zeros through the buffer range check; two stages, one barrier per 128-pixel tile (144 MFMAs/wave).
//   * waves: 2 (64 n each) x 4 (16 of the chunk's 64 channels each, all 9 taps).
//   * the pixel range is split over the workgroups that share an output block; the partials go to the
//     workspace with plain stores (fp32 atomics issue at ~1 wave-instruction / 50 ns / CU: 56 us for a
//     workgroup's 288 KiB) and wgrad_reduce_kernel adds them into dw.  dbias comes from the dy image.
typedef __attribute__((address_space(3))) void* lds_dst3;
constexpr unsigned WG_OOB = 0x80000000u;
constexpr int WG_SRD = 0x00020000;
constexpr int WGH_DY = 128 * 256;                 // dy image of a stage
constexpr int WGH_STAGE = WGH_DY + 36 * 1024;     // + halo image (36 pieces of 8 pixels x 128 B at most)

struct WgHP {
    const char* dy; const char* in1; const char* in2; float* dw; float* dbias; float* ws;
    int B, Hi, Wi, C1, C2, N, ldy, ldw;
    int ntiles, nchunks, blocks, splits, tiles_per_split;
};

// S2 = true: the 4x4 / stride-2 / pad-1 layer, one input-parity class (py, px) per workgroup.  With V(y, x) = X(2y + py, 2x + px) the
// class's four taps (ky, kx) = (2 jy + 1 - py, 2 jx + 1 - px), j in {0, 1}^2, read V at (qy + jy - py, qx + jx - px): a 2x2-tap layer on a
// sub-image of the output's size.  The halo is staged from row y0 - py / column -px on, so the taps sit at halo offsets (jy, jx) for
// every class (compile-time fragment addresses); only the source addresses (pixels two apart, rows two apart) and the edge tests know
// the class.  p.Hi / p.Wi are the OUTPUT (= dy = sub-image) extents.  The splits meet in dw through fp32 atomics.
template <typename T, int TW, bool S2 = false>
__global__ __launch_bounds__(512) void wgrad3x3_halo_kernel(const WgHP p) {
    constexpr int NTAP = S2 ? 4 : 9;
    // TW = 8 (8x8 images): a tile is TWO whole images side by side in the halo ([0 A 0][0 B 0], 10 columns apiece, HS = 24)
    constexpr int R = TW == 8 ? 8 : 128 / TW, HS = TW == 8 ? 24 : TW + 8, HR = R + 2, XP = HR * HS / 8;    // 36 / 30 / 30 / 30 halo pieces
    constexpr int HI = (TW == 8 ? 2 * HS : TW == 16 ? HS : 16) * 128;         // byte offset of the k-step's second 16 pixels in the halo
    static_assert(XP <= 36, "halo does not fit");
    extern __shared__ __attribute__((aligned(16))) char smem[];               // 2 stages

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wc = wave >> 1;
    const int g = lane >> 4, il = lane & 15, q = il >> 2, pp = il & 3;

    // workgroup -> (output block, pixel split); consecutive ids of one XCD share a pixel range
    const int G = gridDim.x;
    int idx = blockIdx.x;
    if ((G & 7) == 0) idx = (idx & 7) * (G >> 3) + (idx >> 3);
    const int block = idx % p.blocks, split = idx / p.blocks;
    const int cls = S2 ? block & 3 : 0, py = cls >> 1, px = cls & 1;      // S2: blocks = 4 classes x n tiles x chunks
    const int blk = S2 ? block >> 2 : block;
    const int nbk = blk / p.nchunks, chunk = blk - nbk * p.nchunks;
    const int n0 = nbk * 128, c0 = chunk * 64;
    const int C = p.C1 + p.C2;
    const bool first = c0 < p.C1;
    const int Cs = first ? p.C1 : p.C2;
    const int cin0 = first ? c0 : c0 - p.C1;
    const int t_lo = split * p.tiles_per_split, t_hi = min(t_lo + p.tiles_per_split, p.ntiles);
    // TW = 64 also serves wider images (Wi a multiple of 64): the tile is then 2 rows x 64 COLUMNS x0 .. x0+63
    const int tcols = TW == 64 ? p.Wi >> 6 : 1;
    const int RW = TW == 64 ? p.Wi : TW;                   // pixels per image row in memory
    const int tiles_img = TW == 8 ? 1 : ((p.Hi * TW) >> 7) * tcols;

    const __amdgpu_buffer_rsrc_t rDY = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.ntiles * 128 * p.ldy * 2, WG_SRD);
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)(first ? p.in1 : p.in2), 0, p.ntiles * 128 * Cs * 2 * (S2 ? 4 : 1), WG_SRD);

    // ---- staging: every per-lane quantity is ONE register per image; the piece index moves through scalar offsets
    // dy pieces 8 j + wave (4 rows x 256 B each): row & 7 does not depend on j, so neither does the lane's source column
    unsigned dyv;
    {
        const int row = wave * 4 + (lane >> 4), s = lane & 15;
        const int lslot = ((((s >> 1) ^ (row & 7)) << 1) | (s & 1));
        const int ncol = n0 + lslot * 8;
        dyv = ncol + 8 <= p.ldy ? (unsigned)((row * p.ldy + ncol) * 2) : WG_OOB;
    }
    // halo pieces (8 pixels x 128 B): a wave takes one 8-pixel column strip of the halo and walks down its rows, so the
    // lane's column, swizzle and x-validity are fixed and a row is a scalar step.  NC strips x HR rows:
    //   TW=64: 9 x 4 — waves 0..7 take strips 0..7, waves 0..3 also row `wave` of strip 8;   TW=32: 5 x 6 — waves 0..4;
    //   TW=16: 3 x 10 — waves 0..5 as 3 strips x 2 groups of 5 rows.
    constexpr int NC = HS / 8, RG = TW <= 16 ? 2 : 1, RPG = HR / RG;
    const int strip = wave % (NC < 8 ? NC : 8), rgroup = wave / (NC < 8 ? NC : 8);
    const bool halo_wave = rgroup < RG;
    const int hy0 = rgroup * RPG;
    int xv0, xv8 = 0, xedge0 = 0;
    bool xok0, xok8 = false, xright8 = false;
    {
        const int hx = strip * 8 + (lane >> 3), s = lane & 7;
        const int lslot = ((((s >> 1) ^ ((hx >> 1) & 3)) << 1) | (s & 1));       // HS % 8 == 0: the swizzle follows hx only
        const int img = TW == 8 ? hx / 10 : 0;                                   // TW = 8: which of the tile's two images
        const int x = (TW == 8 ? hx - img * 10 : hx) - (S2 ? px : 1);
        if constexpr (S2)    // sub-image pixel (y, x) = input pixel (2 y + py, 2 x + px) of a (2 Hi) x (2 Wi) image
            xv0 = (((hy0 - py) * 2 + py) * (2 * p.Wi) + img * (4 * 64) + 2 * x + px) * Cs * 2 + lslot * 16 + cin0 * 2;
        else
        xv0 = ((hy0 - 1) * RW + img * 64 + x) * Cs * 2 + lslot * 16 + cin0 * 2;
        // column inside the tile; columns -1 and TW belong to the neighbouring column tile when there is one (xedge0: 1 = left, 2 = right)
        xok0 = (unsigned)x < (unsigned)TW && img < 2 && lslot * 8 < Cs;      // (a lone partial chunk, C < 64, reads zeros past C)
        xedge0 = (TW == 64 && lslot * 8 < Cs) ? (x == -1 ? 1 : x == TW ? 2 : 0) : 0;
        if constexpr (TW == 64) {
            const int hx8 = 64 + (lane >> 3);
            xv8 = ((wave - 1) * RW + hx8 - 1) * Cs * 2 + lslot * 16 + cin0 * 2;      // row `wave` of strip 8 (64 % 8 == 0: same lslot)
            xok8 = hx8 - 1 < TW && lslot * 8 < Cs;
            xright8 = hx8 - 1 == TW && lslot * 8 < Cs;
        }
    }
    // `parts`: bit 0 = dy pieces 0, 1; bit 1 = dy pieces 2, 3; bit 2 = the first half of the wave's halo rows; bit 3 = the rest
    auto issue = [&](int stage, int tile, int parts = 15) {
        char* sA = smem + stage * WGH_STAGE;
        char* sX = sA + WGH_DY;
        const int bimg = tile / tiles_img, trem = tile - bimg * tiles_img;
        const int y0 = TW == 8 ? 0 : (trem / tcols) * R, x0 = (trem % tcols) * 64;
        const int pix0 = TW == 8 ? tile * 128 : (bimg * p.Hi + y0) * RW + x0;       // first pixel of the tile
#pragma unroll
        for (int j = 0; j < 4; ++j) {                  // tile pixels 32 j .. 32 j + 31: (part of) one image row, or whole rows
            if (!(parts & (j < 2 ? 1 : 2))) continue;
            const int prow = (TW == 64 && tcols > 1) ? pix0 + (j >> 1) * RW + (j & 1) * 32 : pix0 + j * 32;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rDY, (lds_dst3)(sA + (j * 8 + wave) * 1024), 16, dyv, prow * p.ldy * 2, 0, 0);
        }
        const bool top = y0 == 0, bottom = y0 + R == p.Hi;                 // halo rows outside the image
        const bool has_left = x0 > 0, has_right = x0 + 64 < RW;            // neighbouring column tiles (TW = 64 only)
        // S2: the tile's first sub-image pixel (bimg, y0, 0) is input pixel (bimg, 2 y0, 0) — 4 input pixels per output pixel
        const int tbase = S2 ? pix0 * 4 * Cs * 2 : pix0 * Cs * 2;
        const int rstep = S2 ? 4 * p.Wi * Cs * 2 : RW * Cs * 2;            // one halo row down: two input rows of 2 Wi pixels
        if (halo_wave) {
#pragma unroll
            for (int i = 0; i < RPG; ++i) {
                if (!(parts & (i < RPG / 2 ? 4 : 8))) continue;
                const int hy = hy0 + i;
                // S2: halo row hy is sub-image row y0 + hy - py; rows 0 .. R are used
                const bool row_ok = S2 ? !((py == 1 && hy == 0 && top) || (py == 0 && hy == R && bottom) || hy > R)
                                       : !((hy == 0 && top) || (hy == HR - 1 && bottom));
                const bool col_ok = xok0 || (xedge0 == 1 && has_left) || (xedge0 == 2 && has_right);
                const unsigned v = (col_ok && row_ok) ? (unsigned)(tbase + i * rstep + xv0) : WG_OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lds_dst3)(sX + (hy * NC + strip) * 1024), 16, v, 0, 0, 0);
            }
        }
        if constexpr (TW == 64) {
            if (wave < 4 && (parts & 8)) {
                const bool row_ok = !((wave == 0 && top) || (wave == HR - 1 && bottom));
                const unsigned v = ((xok8 || (xright8 && has_right)) && row_ok) ? (unsigned)(tbase + xv8) : WG_OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lds_dst3)(sX + (wave * NC + 8) * 1024), 16, v, 0, 0, 0);
            }
        }
    };

    // ---- fragment read addresses
    int addrA[4], addrB[3][2];                         // [kx][image of the tile (TW = 8 only)]
    {
        const int row = 4 * g + q;
#pragma unroll
        for (int it = 0; it < 4; ++it) addrA[it] = row * 256 + (((wn * 4 + it) ^ (row & 7)) << 5) + 8 * pp;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int im = 0; im < 2; ++im) {
                // halo row of this lane's first pixel of a k-step: 32 consecutive pixels of an image row (TW >= 32), two rows
                // (TW = 16), or four rows of one 8x8 image (TW = 8: pixel kp -> row kp >> 3, column kp & 7 of image `im`)
                const int hp = TW == 8 ? (row >> 3) * HS + (row & 7) + im * 10 + kx : row + kx;
                addrB[kx][im] = WGH_DY + hp * 128 + ((wc ^ ((hp >> 1) & 3)) << 5) + 8 * pp;
            }
    }
    typedef __attribute__((address_space(3))) s16x4* lds_p;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    auto tr_pair = [&](const char* a, int hi) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a));
        const s16x4 h = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a + hi));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], h[0], h[1], h[2], h[3]};
        return __builtin_bit_cast(bf16x8, v);
    };

    f32x4 acc[4][NTAP];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < NTAP; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // dbias[n] = sum over pixels of dy, straight from the dy image.  Every chunk of an n tile stages the same dy rows,
    // so the tiles are dealt round-robin over the chunks; a thread sums 4 columns x 8 rows of its tile.
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    const bool want_bias = p.dbias != nullptr && cls == 0;      // (S2: the four classes stage the same dy rows)
    const int bcol = (tid & 31) * 4, brow = tid >> 5;
    int bphase = t_lo % p.nchunks;

    if (t_lo < t_hi) issue(0, t_lo);
    for (int tile = t_lo; tile < t_hi; ++tile) {
        const int cur = (tile - t_lo) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();            // this tile has landed for every wave; the other stage is free again
#if DM_WGRAD_DMA_POS == 0
        if (tile + 1 < t_hi) issue(cur ^ 1, tile + 1);
#endif
        const char* sS = smem + cur * WGH_STAGE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int koffA = ks * 32 * 256;
            const int koffB = (TW == 64 ? (ks >> 1) * HS + (ks & 1) * 32 : TW == 32 ? ks * HS : TW == 16 ? 2 * ks * HS : (ks & 1) * 4 * HS) * 128;
            const int im = TW == 8 ? ks >> 1 : 0;
            bf16x8 fa[4], fb[NTAP];
#pragma unroll
            for (int it = 0; it < 4; ++it) fa[it] = tr_pair(sS + addrA[it] + koffA, 16 * 256);
#pragma unroll
            for (int t = 0; t < NTAP; ++t) {      // tap t = (row, column) offset in the halo: 3x3, or 2x2 for a stride-2 class
                constexpr int TPR = S2 ? 2 : 3;
                fb[t] = tr_pair(sS + addrB[t % TPR][im] + koffB + ((t / TPR) * HS) * 128, HI);
            }
#pragma unroll
            for (int t = 0; t < NTAP; ++t) {
#if DM_WGRAD_DMA_POS != 0
                // the next tile's LDS-DMA pieces go between the MFMAs, after the k-step's fragment reads (an LDS-DMA piece costs 100-185
                // issue cycles next to ds_reads, 25-60 among MFMAs: MI355X_MICROARCH.md): the dy pieces before tap (DY_SLOT & 15) of
                // k-step (DY_SLOT >> 4), the halo pieces at X_SLOT
                {
                    constexpr int SD = DM_WGRAD_DMA_POS >> 8, SX = DM_WGRAD_DMA_POS & 255;
                    constexpr int TD = (SD & 15) < NTAP ? (SD & 15) : NTAP - 1, TX = (SX & 15) < NTAP ? (SX & 15) : NTAP - 1;   // (stride-2 classes: 4 taps)
                    const int pm = (ks == (SD >> 4) && t == TD ? 3 : 0) | (ks == (SX >> 4) && t == TX ? 12 : 0);
                    if (pm != 0) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (tile + 1 < t_hi) issue(cur ^ 1, tile + 1, pm);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#endif
#pragma unroll
                for (int it = 0; it < 4; ++it) acc[it][t] = WMma<T>::run(fa[it], fb[t], acc[it][t]);
            }
            // keep the k-steps apart: merged / hoisted halo reads push the kernel past 256 VGPRs, and a scratch reload in the
            // loop waits on vmcnt — i.e. on the next tile's DMA — which serialises staging and compute
            __builtin_amdgcn_sched_barrier(0);
        }
        if (want_bias && bphase == chunk) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = brow + 16 * j;
                const typename V16<T>::x4 v = *(const typename V16<T>::x4*)(sS + r * 256 + (((bcol >> 4) ^ (r & 7)) << 5) + (bcol & 15) * 2);
#pragma unroll
                for (int e = 0; e < 4; ++e) bsum[e] += (float)v[e];
            }
        }
        bphase = bphase + 1 == p.nchunks ? 0 : bphase + 1;
    }

    // ---- flush: D[i = n][j = c] per tap; lane holds rows 4g..4g+3 (n), column il (c)
    if constexpr (S2) {                          // tap (jy, jx) of class (py, px) is weight tap (2 jy + 1 - py, 2 jx + 1 - px) of the 4x4 kernel
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * 64 + it * 16 + 4 * g + r;
                if (n >= p.N || c0 + wc * 16 + il >= C) continue;
                float* row = p.dw + (size_t)n * p.ldw + c0 + wc * 16 + il;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int ky = 2 * (t >> 1) + 1 - py, kx = 2 * (t & 1) + 1 - px;
                    atomicAdd(row + (ky * 4 + kx) * C, acc[it][t][r]);
                }
            }
    } else if (p.splits == 1) {                  // sole owner of its dw elements within the launch
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * 64 + it * 16 + 4 * g + r;
                if (n >= p.N) continue;
                float* row = p.dw + (size_t)n * p.ldw + c0 + wc * 16 + il;
#pragma unroll
                for (int t = 0; t < 9; ++t)
                    if (c0 + wc * 16 + il < C) row[t * C] += acc[it][t][r];
            }
    } else {                                     // register image as it stands: 1 KiB contiguous per wave-instruction
        // r04, two measurements behind keeping this two-launch form (profiles/r04_ab_same_box.txt):
        //  * what the fold costs: with these stores compiled out (DM_WGRAD_NOFLUSH) dm_conv_wgrad takes 72 - 74 us instead of 82 - 84 on
        //    the 64x64 / 32x32 / 16x16 layers — a ~10-us store tail (all 256 workgroups write their 288 KiB at once) + a ~17-us reduce launch;
        //  * an in-kernel pairwise tree fold instead of the reduce launch (commit d33a092: of the two workgroups of a tree node the first
        //    to arrive parks its image with sc1 stores and raises a ready word, the second polls it, loads the image with sc1 loads, adds
        //    and moves up; no co-residency needed, fixed tree = fixed sums; bit-exact on every test): a level costs ~20 us — the store
        //    drain of one workgroup followed by the dependent load batches of another, across XCDs, is a latency chain where the reduce
        //    launch is a bandwidth-bound sweep by 9000 workgroups: 8x8 layers (2 splits, 1 level) 86 -> 108 us, 16x16 (8 splits, 3 levels)
        //    81 -> 142 us, two / three levels ahead of a smaller reduce launch on the 32x32 / 64x64 layers 81 -> 96 / 105 us.  Every partial
        //    has to leave its CU once whatever the scheme (256 x 288 KiB per launch), so a fold can only ever save the second pass.
#if DM_WGRAD_NOFLUSH                             // diagnostic build only (results garbage)
        if (p.N > 0) return;
#endif
        f32x4* dst = (f32x4*)p.ws + ((((size_t)split * p.blocks + block) * 8 + wave) * 36) * 64 + lane;
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int t = 0; t < NTAP; ++t) dst[(it * 9 + t) * 64] = acc[it][t];
    }
    if (want_bias) {                             // fold the 16 row groups in LDS, then one value per column and workgroup
        __syncthreads();
        float* sb = (float*)smem;                // [16][128]
#pragma unroll
        for (int e = 0; e < 4; ++e) sb[brow * 128 + bcol + e] = bsum[e];
        __syncthreads();
        if (tid < 128) {
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) v += sb[j * 128 + tid];
            if (p.splits > 1 && !S2) p.ws[(size_t)p.splits * p.blocks * (128 * 9 * 64) + ((size_t)split * p.blocks + block) * 128 + tid] = v;
            else if (n0 + tid < p.N) atomicAdd(p.dbias + n0 + tid, v);
        }
    }
}

// dw += sum over the pixel splits of the register images in ws: slot e = ((block*8 + wave)*36 + it*9 + t)*64 + lane holds
// rows n = 4g..4g+3 of the 16x16 tile (it, t) of that wave, column c = il
// SL = 8: eight threads share an element and take every 8th split (128 splits on the 64x64 layers: one thread per element ran
// a 16-deep chain of load batches on 144 workgroups — 18 us for 75 MB); the eight partial sums fold through LDS in a fixed order.
template <int SL>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const f32x4* __restrict__ ws, float* __restrict__ dw, float* __restrict__ dbias, int N,
                                                            int C, int nchunks, int blocks, int ldw, int splits, int overwrite) {
    constexpr int EPB = 256 / SL;                // elements per workgroup
    __shared__ f32x4 red[SL > 1 ? 256 : 1];
    const int per_split = blocks * 8 * 36 * 64;
    const int nmain = per_split / EPB;
    if ((int)blockIdx.x >= nmain) {              // trailing workgroups, one wave per n: dbias[n] += the per-workgroup column sums of dy
        const int t = ((int)blockIdx.x - nmain) * 256 + threadIdx.x;
        const int n = t >> 6, l = t & 63;
        if (dbias == nullptr || n >= N) return;
        const float* wb = (const float*)(ws + (size_t)splits * per_split);
        float v = 0.f;
        for (int i = l; i < splits * nchunks; i += 64) {
            const int k = i / nchunks, ch = i - k * nchunks;
            v += wb[((size_t)k * blocks + (n >> 7) * nchunks + ch) * 128 + (n & 127)];
        }
        v = wave_sum(v);
        if (l == 0) dbias[n] += v;
        return;
    }
    const int el = threadIdx.x % EPB, sub = threadIdx.x / EPB;
    const int e = blockIdx.x * EPB + el;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int k = sub; k < splits; k += SL) s += ws[(size_t)k * per_split + e];
    if constexpr (SL > 1) {
        red[threadIdx.x] = s;
        __syncthreads();
        if (sub != 0) return;
#pragma unroll
        for (int l = 1; l < SL; ++l) s += red[l * EPB + el];
    }
    const int lane = e & 63, tile = (e >> 6) % 36, wave = (e / (64 * 36)) & 7, block = e / (64 * 36 * 8);
    const int it = tile / 9, t = tile - it * 9;
    const int nbk = block / nchunks, chunk = block - nbk * nchunks;
    const int n = nbk * 128 + (wave & 1) * 64 + it * 16 + 4 * (lane >> 4);
    float* o = dw + (size_t)n * ldw + t * C + chunk * 64 + (wave >> 1) * 16 + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (n + r < N && chunk * 64 + (wave >> 1) * 16 + (lane & 15) < C) {
            if (overwrite) o[(size_t)r * ldw] = s[r];           // DmWgrad.overwrite: dw was not zeroed (no read of it either)
            else o[(size_t)r * ldw] += s[r];
        }
}

// =================================================================================================
// wgrad3x3_skinny_kernel: weight gradient of the 3x3 stride-1 layers with 8 (padded) channels on ONE side — the stem (3 -> F,
// new_scripy.py:381 -> :184) and the head (F -> 3, :314) — at full resolution
// =================================================================================================
// dw[n][tap][c] = sum over pixels of dy[p][n] * x[p + tap][c].  The halo kernel above gives the 8-channel side a 64- (input) or
// 128-wide (output) tile of zeros.  Here the WIDE side (128 channels: dy for the stem, x for the head) is the MFMA's row operand,
// unshifted, and the NARROW side's 8 channels x 9 taps are PACKED into the column operand: a transposed read delivers 4 pixels x 16
// columns from four 8-byte pieces per pixel row, and a lane's address may point each piece anywhere — columns 0..7 at the narrow
// pixel shifted by tap 2 jt, columns 8..15 at tap 2 jt + 1.  D[wide 128][(tap, narrow) 72 of 80] per k-step of 32 pixels: 20 MFMAs
// per wave instead of 144.  SWAP = false (stem): narrow = x shifted by +tap, dw[n = wide][tap][c = narrow], and the spare column 72
// reads a constant-one pixel, i.e. it is dbias.  SWAP = true (head): narrow = dy shifted by -tap, dw[n = narrow][tap][c = wide]; dbias
// = column sums of the narrow tile (wave 7).
//   * tile = 2 image rows x 64 columns (128 pixels); 8 waves = 2 (wide halves of 64 channels) x 4 (k-steps of the tile): every wave
//     owns ONE k-step of every tile, so all fragment addresses are per-lane constants.
//   * wide image: the dy image of the kernel above (256-B rows, 32-B chunk ch at ch ^ (row & 7)); narrow image: 4 x 72 pixels x 16 B,
//     linear, out-of-image pixels zero (DMA range check); two stages, one barrier per tile.
//   * the four k-step waves fold through LDS at the end; the two remaining register images go to the workspace and
//     wgrad_skinny_reduce_kernel adds the workgroups' partials into dw / dbias.
constexpr int WGS_WIDE = 128 * 256;                        // wide image of a stage
constexpr int WGS_NARROW = 5 * 1024;                       // 4 x 72 x 16 B = 4608 B, filled as 5 pieces
constexpr int WGS_STAGE = WGS_WIDE + WGS_NARROW;
constexpr int WGS_LDS = 2 * WGS_STAGE + 64;                // + the constant-one pixel

// One LDS-DMA piece (64 lanes x 16 B -> 1 KiB at `lds`) as inline asm.  hipcc orders every ds_read_tr INTRINSIC after all LDS-DMA it
// knows to be pending (s_waitcnt vmcnt(0) before the first transposed read that follows an issue), which would serialise this
// kernel's two stages; it does not see these.  M0 = LDS base, one wait state before the DMA reads it; no other user of M0 here.
__device__ __forceinline__ void wgs_dma16(__amdgpu_buffer_rsrc_t r, char* lds, unsigned voff, int soff) {
    const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_dst3)lds);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(la), "v"(voff), "s"(r), "s"(soff) : "memory");
}

struct WgSP {
    const char* wide; const char* narrow; float* dw; float* dbias; float* ws;
    int B, Hi, Wi, CW, ldwide, N, ldw, C;                  // CW: channels of the wide tensor; N / C: the layer's output / input channels
    int ntiles, tiles_per_wg, nblk;                        // nblk: 128-channel blocks of the wide side
};

template <typename T, bool SWAP>
__global__ __launch_bounds__(512) void wgrad3x3_skinny_kernel(const WgSP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wh = wave & 1, wq = wave >> 1;               // wide half, k-step of the tile
    const int g = lane >> 4, il = lane & 15, q = il >> 2, pp = il & 3;
    const int blk = blockIdx.x % p.nblk, wgi = blockIdx.x / p.nblk;
    const int c0 = blk * 128;
    const int t_lo = wgi * p.tiles_per_wg, t_hi = min(t_lo + p.tiles_per_wg, p.ntiles);
    const int tcols = p.Wi >> 6, tiles_img = (p.Hi >> 1) * tcols;

    const __amdgpu_buffer_rsrc_t rWd = __builtin_amdgcn_make_buffer_rsrc((void*)p.wide, 0, p.ntiles * 128 * p.ldwide * 2, WG_SRD);
    const __amdgpu_buffer_rsrc_t rNr = __builtin_amdgcn_make_buffer_rsrc((void*)p.narrow, 0, p.ntiles * 128 * 16, WG_SRD);
    if (tid < 8) ((unsigned short*)(smem + 2 * WGS_STAGE))[tid] = tid == 0 ? (std::is_same<T, f16>::value ? 0x3C00 : 0x3F80) : 0;     // 1.0 in the tensor type
    __syncthreads();

    // ---- staging constants: wide pieces 8 j + wave (4 pixel rows x 256 B), narrow piece `wave` (waves 0..4, 64 pixels x 16 B)
    unsigned wdv;
    {
        const int row = wave * 4 + (lane >> 4), s_ = lane & 15;
        const int lslot = ((((s_ >> 1) ^ (row & 7)) << 1) | (s_ & 1));
        const int col = c0 + lslot * 8;
        wdv = col + 8 <= p.CW ? (unsigned)(((lane >> 4) * p.ldwide + col) * 2) : WG_OOB;     // + the piece's first pixel row as a scalar offset
    }
    const int nhp = wave * 64 + lane, nhy = nhp / 72, nhx = nhp - nhy * 72;
    auto issue = [&](int stage, int tile) {
        char* sA = smem + stage * WGS_STAGE;
        const int bimg = tile / tiles_img, trem = tile - bimg * tiles_img;
        const int y0 = (trem / tcols) * 2, x0 = (trem % tcols) * 64;
        const int pix0 = (bimg * p.Hi + y0) * p.Wi + x0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {                      // pieces 8 j + wave: tile pixels 4 (8 j + wave) .. + 3 — (part of) one image row
            const int tp = (8 * j + wave) * 4;              // first tile pixel of the piece: row tp >> 6 of the tile, column tp & 63
            const int prow = pix0 + (tp >> 6) * p.Wi + (tp & 63);
            wgs_dma16(rWd, sA + (j * 8 + wave) * 1024, wdv, prow * p.ldwide * 2);
        }
        if (wave < 5) {
            const int y = y0 + nhy - 1, x = x0 + nhx - 1;
            const bool ok = nhy < 4 && nhx < 66 && (unsigned)y < (unsigned)p.Hi && (unsigned)x < (unsigned)p.Wi;
            const unsigned v = ok ? (unsigned)(((bimg * p.Hi + y) * p.Wi + x) * 16) : WG_OOB;
            wgs_dma16(rNr, sA + WGS_WIDE + wave * 1024, v, 0);
        }
    };

    // ---- fragment addresses (this wave's k-step wq: tile pixels 32 wq .. 32 wq + 31 = tile row wq >> 1, columns 32 (wq & 1) ..)
    int addrA[4], addrB[5];
    {
        const int row = 4 * g + q;                          // pixel of the k-step (the second half of a pair: + 16)
#pragma unroll
        for (int it = 0; it < 4; ++it) addrA[it] = (wq * 32 + row) * 256 + (((wh * 4 + it) ^ (row & 7)) << 5) + 8 * pp;
#pragma unroll
        for (int jt = 0; jt < 5; ++jt) {
            const int t = 2 * jt + (pp >> 1), tt = t < 9 ? t : 8;
            const int sy = SWAP ? 1 - tt / 3 : tt / 3 - 1, sx = SWAP ? 1 - tt % 3 : tt % 3 - 1;
            const int hp = (1 + (wq >> 1) + sy) * 72 + 1 + (wq & 1) * 32 + row + sx;
            addrB[jt] = WGS_WIDE + hp * 16 + (pp & 1) * 8;
        }
    }
    typedef __attribute__((address_space(3))) s16x4* lds_p;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    auto tr_pair = [&](const char* a, int hi) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a));
        const s16x4 h = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a + hi));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], h[0], h[1], h[2], h[3]};
        return __builtin_bit_cast(bf16x8, v);
    };
    // stem: the spare taps of column tile 4 (t = 9) read the constant-one pixel: column 72 accumulates sum(dy) = dbias
    const bool ones = !SWAP && (pp >> 1) == 1;

    f32x4 acc[4][5];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // head: column sums of the narrow tile (wave 7)

    if (t_lo < t_hi) issue(0, t_lo);
    for (int tile = t_lo; tile < t_hi; ++tile) {
        const int cur = (tile - t_lo) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tile + 1 < t_hi) issue(cur ^ 1, tile + 1);
        const char* sS = smem + cur * WGS_STAGE;
        bf16x8 fa[4], fb[5];
#pragma unroll
        for (int it = 0; it < 4; ++it) fa[it] = tr_pair(sS + addrA[it], 16 * 256);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) fb[jt] = tr_pair(sS + addrB[jt], 16 * 16);
        fb[4] = tr_pair(ones ? smem + 2 * WGS_STAGE + (pp & 1) * 8 : sS + addrB[4], ones ? 0 : 16 * 16);      // (address select: every lane takes part in the read)
#pragma unroll
        for (int jt = 0; jt < 5; ++jt)
#pragma unroll
            for (int it = 0; it < 4; ++it) acc[it][jt] = WMma<T>::run(fa[it], fb[jt], acc[it][jt]);
        if (SWAP && p.dbias != nullptr && wave == 7 && blk == 0) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {                   // tile pixels lane, lane + 64: narrow rows 1, 2, columns 1 .. 64
                const typename V16<T>::x8 v = *(const typename V16<T>::x8*)(sS + WGS_WIDE + ((1 + h) * 72 + 1 + lane) * 16);
#pragma unroll
                for (int e = 0; e < 8; ++e) bsum[e] += (float)v[e];
            }
        }
    }

    // ---- fold the four k-step waves of each wide half (k-steps 1..3 into 0), then park the register images
    __syncthreads();
    f32x4* fold = (f32x4*)smem;                            // [wh][20][64] f32x4 = 40 KiB
#pragma unroll 1
    for (int r = 1; r < 4; ++r) {
        if (wq == r) {
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int jt = 0; jt < 5; ++jt) fold[(wh * 20 + it * 5 + jt) * 64 + lane] = acc[it][jt];
        }
        __syncthreads();
        if (wq == 0) {
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int jt = 0; jt < 5; ++jt) {
                    const f32x4 v = fold[(wh * 20 + it * 5 + jt) * 64 + lane];
                    acc[it][jt] += v;
                }
        }
        __syncthreads();
    }
    if (wq == 0) {
        f32x4* dst = (f32x4*)p.ws + ((size_t)blockIdx.x * 2 + wh) * 20 * 64 + lane;
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int jt = 0; jt < 5; ++jt) dst[(it * 5 + jt) * 64] = acc[it][jt];
    }
    if (SWAP && p.dbias != nullptr && wave == 7 && blk == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = bsum[e];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            // parked next to the images: the reduce kernel folds the workgroups' sums in a fixed order
            if (lane == 0) p.ws[(size_t)gridDim.x * 10240 + wgi * 8 + e] = v;
        }
    }
}

// dw (+ dbias for the stem) += sum over the workgroups' register images: element (wh, it, jt, lane, r) is row i = 64 wh + 16 it +
// 4 (lane >> 4) + r of the wide block, column j = 16 jt + (lane & 15)
template <bool SWAP>
__global__ __launch_bounds__(256) void wgrad_skinny_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, float* __restrict__ dbias,
                                                                 int nwg, int nblk, int N, int C, int ldw) {
    const int e = blockIdx.x * 256 + threadIdx.x;          // float index within a (block, workgroup) image: 2 * 20 * 64 * 4 = 10240
    const int blk = blockIdx.y;
    if (SWAP && dbias != nullptr && blockIdx.x == 0 && blk == 0 && blockIdx.z == 0) {    // head: the workgroups' column sums of dy, a wave per channel
        const int lane = threadIdx.x & 63;
        for (int n = threadIdx.x >> 6; n < N; n += 4) {
            float b = 0.f;
            for (int w = lane; w < nwg; w += 64) b += ws[(size_t)nwg * nblk * 10240 + w * 8 + n];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) b += __shfl_xor(b, o, 64);
            if (lane == 0) dbias[n] += b;
        }
    }
    if (e >= 10240) return;
    // gridDim.z slices of the workgroup list, eight independent loads in flight per thread; the slices meet in dw through atomics
    const int per = (nwg + gridDim.z - 1) / gridDim.z, w0 = blockIdx.z * per, w1 = min(w0 + per, nwg);
    float s = 0.f;
    int w = w0;
    for (; w + 8 <= w1; w += 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = ws[((size_t)((w + k) * nblk + blk)) * 10240 + e];
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[k];
    }
    for (; w < w1; ++w) s += ws[((size_t)(w * nblk + blk)) * 10240 + e];
    if (w0 >= w1) return;
    const int r = e & 3, lane = (e >> 2) & 63, tile = (e >> 8) % 20, wh = e / (20 * 256);
    const int it = tile / 5, jt = tile - it * 5;
    const int i = blk * 128 + wh * 64 + it * 16 + 4 * (lane >> 4) + r, j = jt * 16 + (lane & 15);
    if (SWAP) {
        const int t = j >> 3, n = j & 7;
        if (j < 72 && n < N && i < C) atomicAdd(dw + (size_t)n * ldw + t * C + i, s);
    } else {
        if (i >= N) return;
        if (j < 72) { if ((j & 7) < C) atomicAdd(dw + (size_t)i * ldw + (j >> 3) * C + (j & 7), s); }
        else if (j == 72 && dbias != nullptr) atomicAdd(dbias + i, s);
    }
}

int g_wgrad_skinny = 1;
int g_wgrad_halo = 1;

int g_wgrad_variant = 2;
int g_wgrad_blocks = 512;   // workgroups the pixel split aims at (more splits = more parallelism but more fp32 atomic traffic: 16.8 M atomics per launch on the 64x64 4x4 layer at 1024; 512 measured -0.08 ms per step)

template <typename T>
int launch_wgrad(WgP& p, int splitk_req, bool v2_ok, hipStream_t st) {
    const bool v2 = v2_ok && g_wgrad_variant == 2;
    const int KD = v2 ? Wg2Cfg<T>::KD : WgCfg<T>::KD;
    const int C = p.C1 + p.C2;
    p.csteps_c = cdiv(C, 128);
    p.total_steps = cdiv(p.M, KD);
    const int tiles = cdiv(p.N, 128) * p.T * p.csteps_c;
    int splitk = splitk_req;
    if (splitk <= 0) {  // aim at ~g_wgrad_blocks (512) workgroups, at least 4 k-steps (v1) / 2 k-steps (v2) per split
        splitk = cdiv(g_wgrad_blocks, tiles);
        const int min_steps = v2 ? 2 : 4;
        const int maxsplit = p.total_steps / min_steps > 0 ? p.total_steps / min_steps : 1;
        if (splitk > maxsplit) splitk = maxsplit;
        if (splitk < 1) splitk = 1;
    }
    p.steps_per_split = cdiv(p.total_steps, splitk);
    splitk = cdiv(p.total_steps, p.steps_per_split);
    dim3 grid(cdiv(p.N, 128), p.T * p.csteps_c, splitk);
    if (v2) {
        constexpr int bytes = 4 * Wg2Cfg<T>::KD * Wg2Cfg<T>::RB;
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute((const void*)conv_wgrad2_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e != hipSuccess) { dm_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
            attr_set = true;
        }
        p.gx = grid.x; p.gy = grid.y; p.gz = grid.z;
        static const int rmw_on = getenv("DM_WGRAD_RMW") ? atoi(getenv("DM_WGRAD_RMW")) : 1;     // (0: A/B measurements)
        p.rmw = rmw_on && grid.z == 1;
        const unsigned total = grid.x * grid.y * (grid.z >= 8 ? (grid.z + 7) / 8 * 8 : grid.z);
        hipLaunchKernelGGL((conv_wgrad2_kernel<T>), dim3(total), dim3(256), bytes, st, p);
    } else {
        hipLaunchKernelGGL((conv_wgrad_kernel<T>), grid, dim3(256), 0, st, p);
    }
    DM_LAUNCH_CHECK();
    return DM_OK;
}

}  // namespace

template <typename T, int TW>
int launch_wgrad_halo(const WgHP& p, hipStream_t st, int overwrite) {
    if (overwrite && p.splits == 1) {             // the kernel adds into dw itself: zero its rows first (rare: >= 256 output blocks)
        hipError_t e = hipMemsetAsync(p.dw, 0, (size_t)p.N * p.ldw * sizeof(float), st);
        if (e != hipSuccess) { dm_set_error("dm_conv_wgrad: hipMemsetAsync failed: %s", hipGetErrorString(e)); return (int)e; }
    }
    constexpr int bytes = 2 * WGH_STAGE;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)wgrad3x3_halo_kernel<T, TW>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) { dm_set_error("hipFuncSetAttribute(%d B LDS) failed: %s", bytes, hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    hipLaunchKernelGGL((wgrad3x3_halo_kernel<T, TW>), dim3((unsigned)(p.blocks * p.splits)), dim3(512), bytes, st, p);
    DM_LAUNCH_CHECK();
    if (p.splits > 1) {
        const int per_split = p.blocks * 8 * 36 * 64;
        if (p.splits >= 32)
            hipLaunchKernelGGL(wgrad_reduce_kernel<8>, dim3((unsigned)(per_split / 32 + cdiv(p.N, 4))), dim3(256), 0, st, (const f32x4*)p.ws, p.dw, p.dbias,
                               p.N, p.C1 + p.C2, p.nchunks, p.blocks, p.ldw, p.splits, overwrite);
        else if (p.splits >= 8)
            hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3((unsigned)(per_split / 64 + cdiv(p.N, 4))), dim3(256), 0, st, (const f32x4*)p.ws, p.dw, p.dbias,
                               p.N, p.C1 + p.C2, p.nchunks, p.blocks, p.ldw, p.splits, overwrite);
        else
            hipLaunchKernelGGL(wgrad_reduce_kernel<1>, dim3((unsigned)(per_split / 256 + cdiv(p.N, 4))), dim3(256), 0, st, (const f32x4*)p.ws, p.dw, p.dbias,
                               p.N, p.C1 + p.C2, p.nchunks, p.blocks, p.ldw, p.splits, overwrite);
        DM_LAUNCH_CHECK();
    }
    return DM_OK;
}

template <typename T, int TW>
int launch_wgrad_tap4(const WgHP& p, hipStream_t st) {
    constexpr int bytes = 2 * WGH_STAGE;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)wgrad3x3_halo_kernel<T, TW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) { dm_set_error("hipFuncSetAttribute(%d B LDS) failed: %s", bytes, hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    hipLaunchKernelGGL((wgrad3x3_halo_kernel<T, TW, true>), dim3((unsigned)(p.blocks * p.splits)), dim3(512), bytes, st, p);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

// the 4x4 / stride-2 / pad-1 layer (new_scripy.py:229) on output rows of 32 / 16 pixels or 8x8 output images, one source, whole
// 64-channel chunks: the halo-resident kernel in its S2 form (one input-parity class per workgroup, atomics into dw)
int g_wgrad_tap4 = 1;
int g_wgrad_tap4_blocks = 256;     // workgroups the pixel split aims at (each adds 32768 floats to dw through atomics)
bool wgrad_tap4_plan(const DmWgrad* d, int64_t M, WgHP& hp) {
    if (!g_wgrad_tap4 || !g_wgrad_halo || d->dtype == DM_F32 || d->T != 16 || d->KW != 4 || d->sy != 2 || d->sx != 2 || d->splitk > 0) return false;
    if (d->ty != 1 || d->tx != 1 || d->oy0 != -1 || d->ox0 != -1 || d->C2 != 0 || d->C1 % 64 != 0) return false;
    if (d->Hi != 2 * d->Hq || d->Wi != 2 * d->Wq || d->Ho != d->Hq || d->Wo != d->Wq || d->osy != 1 || d->osx != 1 || d->ooy != 0 || d->oox != 0) return false;
    if (d->Wq == 8) {
        if (d->Hq != 8 || d->B % 2 != 0) return false;
    } else if ((d->Wq != 16 && d->Wq != 32) || (d->Hq * d->Wq) % 128 != 0) {
        return false;
    }
    if (d->ldy % 8 != 0 || d->ldw != 16 * d->C1) return false;
    if (M * 4 * d->C1 * 2 >= (1ll << 31) || M * d->ldy * 2 >= (1ll << 31)) return false;
    if (((uintptr_t)d->dy & 15) || ((uintptr_t)d->in1 & 15)) return false;
    hp.dy = (const char*)d->dy; hp.in1 = (const char*)d->in1; hp.in2 = nullptr; hp.dw = d->dw; hp.dbias = d->dbias; hp.ws = nullptr;
    hp.B = d->B; hp.Hi = d->Hq; hp.Wi = d->Wq; hp.C1 = d->C1; hp.C2 = 0; hp.N = d->N; hp.ldy = d->ldy; hp.ldw = d->ldw;
    hp.ntiles = (int)(M / 128);
    hp.nchunks = d->C1 / 64;
    hp.blocks = cdiv(d->N, 128) * hp.nchunks * 4;
    int splits = g_wgrad_tap4_blocks / hp.blocks;
    if (splits < 1) splits = 1;
    if (splits > hp.ntiles) splits = hp.ntiles;
    hp.tiles_per_split = cdiv(hp.ntiles, splits);
    hp.splits = cdiv(hp.ntiles, hp.tiles_per_split);
    return true;
}

// bf16 3x3 stride-1 pad-1 layer on whole 16/32/64-pixel rows, 64-channel chunks, identity dy mapping, 31-bit byte offsets.
// Returns true (and fills hp) when the halo kernel can take the launch with the registered workspace.
bool wgrad_halo_plan(const DmWgrad* d, int64_t M, WgHP& hp) {
    if (!g_wgrad_halo || d->dtype == DM_F32 || d->T != 9 || d->KW != 3 || d->sy != 1 || d->sx != 1) return false;
    if (d->ty != 1 || d->tx != 1 || d->oy0 != -1 || d->ox0 != -1) return false;
    if (d->Hq != d->Hi || d->Wq != d->Wi || d->Ho != d->Hq || d->Wo != d->Wq || d->osy != 1 || d->osx != 1 || d->ooy != 0 || d->oox != 0) return false;
    if (d->Wi == 8) {                                                  // two whole 8x8 images per tile
        if (d->Hi != 8 || d->B % 2 != 0) return false;
    } else if (d->Wi > 64) {                                           // column tiles of 2 rows x 64 pixels
        if (d->Wi % 64 != 0 || d->Hi % 2 != 0) return false;
    } else if ((d->Wi != 16 && d->Wi != 32 && d->Wi != 64) || (d->Hi * d->Wi) % 128 != 0) {
        return false;
    }
    if (d->ldy % 8 != 0) return false;
    if ((d->C1 % 64 != 0 || d->C2 % 64 != 0) && !(d->C2 == 0 && d->C1 < 64)) return false;     // whole chunks, or one partial chunk
    const int64_t cmax = d->C1 > d->C2 ? d->C1 : d->C2;
    if (M * cmax * 2 >= (1ll << 31) || M * d->ldy * 2 >= (1ll << 31)) return false;
    if (((uintptr_t)d->dy & 15) || ((uintptr_t)d->in1 & 15) || ((uintptr_t)d->in2 & 15)) return false;
    const int C = d->C1 + d->C2;
    hp.dy = (const char*)d->dy; hp.in1 = (const char*)d->in1; hp.in2 = (const char*)d->in2; hp.dw = d->dw; hp.dbias = d->dbias; hp.ws = dm_g_ws;
    hp.B = d->B; hp.Hi = d->Hi; hp.Wi = d->Wi; hp.C1 = d->C1; hp.C2 = d->C2; hp.N = d->N; hp.ldy = d->ldy; hp.ldw = d->ldw;
    hp.ntiles = (int)(M / 128);
    hp.nchunks = (C + 63) / 64;
    hp.blocks = cdiv(d->N, 128) * hp.nchunks;
    int splits = 256 / hp.blocks;
    if (splits < 1) splits = 1;
    if (splits > hp.ntiles) splits = hp.ntiles;
    hp.tiles_per_split = cdiv(hp.ntiles, splits);
    hp.splits = cdiv(hp.ntiles, hp.tiles_per_split);
    if (hp.splits > 1 && (dm_g_ws == nullptr || (int64_t)hp.splits * hp.blocks * (128 * 9 * 64 * 4 + 128 * 4) > dm_g_ws_bytes)) return false;
    return true;
}

// 3x3 stride-1 pad-1 layer at full resolution (rows a multiple of 64 pixels, an even number of them) with 8 channels on one side
// and whole 128-channel blocks on the other: 1 = stem form (8 input channels), 2 = head form (<= 8 output channels, ldy == 8)
int wgrad_skinny_plan(const DmWgrad* d, int64_t M, WgSP& sp) {
    if (!g_wgrad_skinny || !g_wgrad_halo || d->dtype == DM_F32 || d->T != 9 || d->KW != 3 || d->sy != 1 || d->sx != 1) return 0;
    if (d->ty != 1 || d->tx != 1 || d->oy0 != -1 || d->ox0 != -1 || d->C2 != 0 || d->splitk > 0) return 0;
    if (d->Hq != d->Hi || d->Wq != d->Wi || d->Ho != d->Hq || d->Wo != d->Wq || d->osy != 1 || d->osx != 1 || d->ooy != 0 || d->oox != 0) return 0;
    if (d->Wi < 64 || d->Wi % 64 != 0 || d->Hi % 2 != 0) return 0;
    if (((uintptr_t)d->dy & 15) || ((uintptr_t)d->in1 & 15) || dm_g_ws == nullptr) return 0;
    int form = 0;
    if (d->C1 == 8 && d->N % 128 == 0 && d->ldy == d->N && d->ldw == 72) form = 1;
    else if (d->N <= 8 && d->ldy == 8 && d->C1 % 128 == 0 && d->ldw == 9 * d->C1) form = 2;
    if (form == 0) return 0;
    const int CW = form == 1 ? d->N : d->C1;
    if (M * CW * 2 >= (1ll << 31)) return 0;
    sp.wide = (const char*)(form == 1 ? d->dy : d->in1);
    sp.narrow = (const char*)(form == 1 ? d->in1 : d->dy);
    sp.dw = d->dw; sp.dbias = d->dbias; sp.ws = dm_g_ws;
    sp.B = d->B; sp.Hi = d->Hi; sp.Wi = d->Wi; sp.CW = CW; sp.ldwide = CW; sp.N = d->N; sp.ldw = d->ldw; sp.C = d->C1;
    sp.ntiles = (int)(M / 128);
    sp.nblk = CW / 128;
    int wgs = 256 / sp.nblk;                                // one workgroup per CU
    if (wgs < 1) wgs = 1;
    if (wgs > sp.ntiles) wgs = sp.ntiles;
    sp.tiles_per_wg = cdiv(sp.ntiles, wgs);
    const int nwg = cdiv(sp.ntiles, sp.tiles_per_wg);
    if (((int64_t)nwg * sp.nblk * 10240 + nwg * 8) * 4 > dm_g_ws_bytes) return 0;
    return form;
}

template <typename T, bool SWAP>
int launch_wgrad_skinny_t(const WgSP& p, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)wgrad3x3_skinny_kernel<T, SWAP>, hipFuncAttributeMaxDynamicSharedMemorySize, WGS_LDS);
        if (e != hipSuccess) { dm_set_error("hipFuncSetAttribute(%d B LDS) failed: %s", WGS_LDS, hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    const int nwg = cdiv(p.ntiles, p.tiles_per_wg);
    hipLaunchKernelGGL((wgrad3x3_skinny_kernel<T, SWAP>), dim3((unsigned)(nwg * p.nblk)), dim3(512), WGS_LDS, st, p);
    DM_LAUNCH_CHECK();
    hipLaunchKernelGGL(wgrad_skinny_reduce_kernel<SWAP>, dim3(40, (unsigned)p.nblk, 16), dim3(256), 0, st, (const float*)p.ws, p.dw, p.dbias, nwg, p.nblk, p.N, p.C, p.ldw);
    DM_LAUNCH_CHECK();
    return DM_OK;
}
int launch_wgrad_skinny(const WgSP& p, bool swap, bool is_f16, hipStream_t st) {
    if (is_f16) return swap ? launch_wgrad_skinny_t<f16, true>(p, st) : launch_wgrad_skinny_t<f16, false>(p, st);
    return swap ? launch_wgrad_skinny_t<bf16, true>(p, st) : launch_wgrad_skinny_t<bf16, false>(p, st);
}

int g_wgrad_pw = 1;            // 1x1 layers on wgrad_pw_kernel
int g_wgrad_pw_blocks = 384;   // workgroups its pixel split aims at when a workgroup's block is 4096 floats; fewer for larger blocks (same atomic traffic)
int g_wgrad_pw_min_m = 0;      // pixels below which the per-tap kernel keeps the launch

// 16-bit 1x1 stride-1 layer, one source, identity pixel mapping, min(N, C) in {32, 64, 128} and the wider side a whole number of blocks
bool wgrad_pw_plan(const DmWgrad* d, int64_t M, WgPwP& q, int& ntw, int& ctw) {
    if (!g_wgrad_pw || d->dtype == DM_F32 || d->T != 1 || d->KW != 1 || d->sy != 1 || d->sx != 1 || d->oy0 != 0 || d->ox0 != 0 || d->C2 != 0) return false;
    if (d->Hq != d->Hi || d->Wq != d->Wi || d->Ho != d->Hq || d->Wo != d->Wq || d->osy != 1 || d->osx != 1 || d->ooy != 0 || d->oox != 0) return false;
    if (((uintptr_t)d->dy & 15) || ((uintptr_t)d->in1 & 15) || d->ldy % 8 != 0 || d->splitk > 0) return false;
    const int N = d->N, C = d->C1;
    const int small = N < C ? N : C, big = N < C ? C : N;
    if (small == 32 && big % 256 == 0) { ntw = 4; ctw = 2; }
    else if (small == 32 && big % 128 == 0) { ntw = 2; ctw = 2; }
    else if (small == 64 && big % 256 == 0) { ntw = 4; ctw = 4; }
    else if (small == 128 && big % 128 == 0) { ntw = 2; ctw = 8; }
    else return false;
    if (M * (int64_t)(d->ldy > C ? d->ldy : C) * 2 >= (1ll << 32) || M < g_wgrad_pw_min_m) return false;
    const int BIG = 64 * ntw, PX = (BIG + 16 * ctw) <= 160 ? 128 : 64;
    q.swap = N < C ? 1 : 0;                       // N == C (128 x 128): dy is the wide operand
    q.big = (const char*)(q.swap ? d->in1 : d->dy);
    q.small = (const char*)(q.swap ? d->dy : d->in1);
    q.ld_big = q.swap ? C : d->ldy;
    q.ld_small = q.swap ? d->ldy : C;
    q.dw = d->dw; q.dbias = d->dbias; q.M = (int)M; q.ldw = d->ldw;
    q.nblocks = big / BIG;
    q.nchunks = (int)((M + PX - 1) / PX);
    int splits = (int)((int64_t)g_wgrad_pw_blocks * 4096 / (BIG * 16 * ctw)) / q.nblocks;
    if (splits > q.nchunks / 2) splits = q.nchunks / 2;      // at least two chunks per workgroup: the second is fetched under the first
    if (splits < 1) splits = 1;
    q.chunks_per_split = cdiv(q.nchunks, splits);
    q.splits = cdiv(q.nchunks, q.chunks_per_split);
    return true;
}

template <typename T>
int launch_wgrad_pw(const WgPwP& q, int ntw, int ctw, hipStream_t st) {
    const dim3 grid((unsigned)(q.nblocks * q.splits));
#define DM_WGPW(A, B_) \
    if (q.swap) hipLaunchKernelGGL((wgrad_pw_kernel<T, A, B_, true>), grid, dim3(256), 0, st, q); \
    else hipLaunchKernelGGL((wgrad_pw_kernel<T, A, B_, false>), grid, dim3(256), 0, st, q)
    if (ntw == 2 && ctw == 2) { DM_WGPW(2, 2); }
    else if (ntw == 4 && ctw == 2) { DM_WGPW(4, 2); }
    else if (ntw == 4 && ctw == 4) { DM_WGPW(4, 4); }
    else { DM_WGPW(2, 8); }
#undef DM_WGPW
    DM_LAUNCH_CHECK();
    return DM_OK;
}

static int g_last_wgrad_path = 0;   // 1: the last dm_conv_wgrad launch went to wgrad3x3_halo_kernel, 2: to wgrad_pw_kernel, 3: to the four-tap (S2) form of 1
extern "C" int dm_last_wgrad_path(void) { return g_last_wgrad_path; }
extern "C" int dm_set_wgrad_skinny(int on) { g_wgrad_skinny = on ? 1 : 0; return DM_OK; }

extern "C" int dm_set_wgrad_tap4(int on) {      // on > 1 also sets the workgroup target
    g_wgrad_tap4 = on != 0;
    if (on > 1) g_wgrad_tap4_blocks = on;
    return DM_OK;
}

extern "C" int dm_set_wgrad_pw(int enable, int target_blocks, int min_pixels) {
    DM_CHECK_ARG(target_blocks >= 0 && target_blocks <= 65536, "dm_set_wgrad_pw: target_blocks %d out of range", target_blocks);
    g_wgrad_pw = enable != 0;
    if (target_blocks > 0) g_wgrad_pw_blocks = target_blocks;
    if (min_pixels >= 0) g_wgrad_pw_min_m = min_pixels;
    return DM_OK;
}

extern "C" int dm_set_wgrad_variant(int variant) {
    // variant >= 64 sets the workgroup target of the pixel split instead (tuning)
    if (variant >= 64) { g_wgrad_blocks = variant; return DM_OK; }
    DM_CHECK_ARG(variant >= 1 && variant <= 3, "dm_set_wgrad_variant: 1 (register staging), 2 (LDS-DMA) or 3 (2 + halo-resident 3x3, default)");
    g_wgrad_halo = variant == 3;
    g_wgrad_variant = variant == 3 ? 2 : variant;
    return DM_OK;
}

extern "C" int dm_conv_wgrad(const DmWgrad* d, dm_stream_t stream) {
    DM_CHECK_ARG(d != nullptr, "dm_conv_wgrad: null descriptor");
    DM_CHECK_ARG(d->dtype == DM_F32 || d->dtype == DM_BF16 || d->dtype == DM_F16, "dm_conv_wgrad: bad dtype %d", d->dtype);
    const int ve = d->dtype == DM_F32 ? 4 : 8;
    DM_CHECK_ARG(d->dy && d->in1 && d->dw, "dm_conv_wgrad: null tensor pointer");
    DM_CHECK_ARG(d->C1 > 0 && d->C1 % ve == 0 && d->C2 >= 0 && d->C2 % ve == 0, "dm_conv_wgrad: C1=%d C2=%d must be multiples of %d", d->C1, d->C2, ve);
    DM_CHECK_ARG(d->C2 == 0 || d->in2, "dm_conv_wgrad: C2 > 0 but in2 is null");
    DM_CHECK_ARG(d->ldy % ve == 0 && d->ldy >= d->N, "dm_conv_wgrad: ldy=%d invalid for N=%d", d->ldy, d->N);
    DM_CHECK_ARG(d->ldw >= d->T * (d->C1 + d->C2), "dm_conv_wgrad: ldw=%d < T*C", d->ldw);
    DM_CHECK_ARG(d->B > 0 && d->Hi > 0 && d->Wi > 0 && d->Hq > 0 && d->Wq > 0 && d->T > 0 && d->KW > 0 && d->N > 0, "dm_conv_wgrad: non-positive extent");
    DM_CHECK_ARG((d->Hq - 1) * d->osy + d->ooy < d->Ho && (d->Wq - 1) * d->osx + d->oox < d->Wo && d->ooy >= 0 && d->oox >= 0, "dm_conv_wgrad: output mapping exceeds Ho/Wo");
    const int64_t M = (int64_t)d->B * d->Hq * d->Wq;
    DM_CHECK_ARG(M < (1ll << 31), "dm_conv_wgrad: M too large");
    DM_CHECK_ARG((int64_t)d->T * cdiv(d->C1 + d->C2, 128) < 65536, "dm_conv_wgrad: grid.y too large");
    WgHP hp;
    g_last_wgrad_path = 0;
    // DmWgrad.overwrite on a path that accumulates through atomics: zero dw here (the halo path below handles it in its reduce launch)
    auto zero_dw = [&]() -> int {
        if (!d->overwrite) return DM_OK;
        hipError_t e = hipMemsetAsync(d->dw, 0, (size_t)d->N * d->ldw * sizeof(float), (hipStream_t)stream);
        if (e != hipSuccess) { dm_set_error("dm_conv_wgrad: hipMemsetAsync failed: %s", hipGetErrorString(e)); return (int)e; }
        return DM_OK;
    };
    {
        WgSP sp;
        const int sk = wgrad_skinny_plan(d, M, sp);         // 1: stem form, 2: head form
        if (sk != 0) {
            if (int rc = zero_dw()) return rc;
            g_last_wgrad_path = 4;
            return launch_wgrad_skinny(sp, sk == 2, d->dtype == DM_F16, (hipStream_t)stream);
        }
    }
    if (wgrad_halo_plan(d, M, hp)) {
        g_last_wgrad_path = 1;
        hipStream_t hst = (hipStream_t)stream;
        if (d->dtype == DM_F16) {
            if (d->Wi >= 64) return launch_wgrad_halo<f16, 64>(hp, hst, d->overwrite);
            if (d->Wi == 32) return launch_wgrad_halo<f16, 32>(hp, hst, d->overwrite);
            if (d->Wi == 16) return launch_wgrad_halo<f16, 16>(hp, hst, d->overwrite);
            return launch_wgrad_halo<f16, 8>(hp, hst, d->overwrite);
        }
        if (d->Wi >= 64) return launch_wgrad_halo<bf16, 64>(hp, hst, d->overwrite);
        if (d->Wi == 32) return launch_wgrad_halo<bf16, 32>(hp, hst, d->overwrite);
        if (d->Wi == 16) return launch_wgrad_halo<bf16, 16>(hp, hst, d->overwrite);
        return launch_wgrad_halo<bf16, 8>(hp, hst, d->overwrite);
    }
    if (int rc = zero_dw()) return rc;
    if (wgrad_tap4_plan(d, M, hp)) {
        g_last_wgrad_path = 3;
        hipStream_t hst = (hipStream_t)stream;
        if (d->dtype == DM_F16) {
            if (d->Wq == 32) return launch_wgrad_tap4<f16, 32>(hp, hst);
            if (d->Wq == 16) return launch_wgrad_tap4<f16, 16>(hp, hst);
            return launch_wgrad_tap4<f16, 8>(hp, hst);
        }
        if (d->Wq == 32) return launch_wgrad_tap4<bf16, 32>(hp, hst);
        if (d->Wq == 16) return launch_wgrad_tap4<bf16, 16>(hp, hst);
        return launch_wgrad_tap4<bf16, 8>(hp, hst);
    }
    WgPwP q;
    int ntw = 0, ctw = 0;
    if (wgrad_pw_plan(d, M, q, ntw, ctw)) {
        g_last_wgrad_path = 2;
        return d->dtype == DM_F16 ? launch_wgrad_pw<f16>(q, ntw, ctw, (hipStream_t)stream) : launch_wgrad_pw<bf16>(q, ntw, ctw, (hipStream_t)stream);
    }
    WgP p;
    p.dy = (const char*)d->dy; p.in1 = (const char*)d->in1; p.in2 = (const char*)d->in2; p.dw = d->dw; p.dbias = d->dbias;
    p.B = d->B; p.Hi = d->Hi; p.Wi = d->Wi; p.C1 = d->C1; p.C2 = d->C2; p.Hq = d->Hq; p.Wq = d->Wq; p.sy = d->sy; p.sx = d->sx;
    p.T = d->T; p.KW = d->KW; p.ty = d->ty; p.tx = d->tx; p.oy0 = d->oy0; p.ox0 = d->ox0;
    p.Ho = d->Ho; p.Wo = d->Wo; p.osy = d->osy; p.osx = d->osx; p.ooy = d->ooy; p.oox = d->oox;
    p.N = d->N; p.ldy = d->ldy; p.ldw = d->ldw; p.M = (int)M;
    // v2 (LDS-DMA) needs the identity dy mapping and 32-bit pixel / element arithmetic
    const int64_t in_elems = (int64_t)d->B * d->Hi * d->Wi * (d->C1 > d->C2 ? d->C1 : d->C2);
    const bool v2_ok = d->osy == 1 && d->osx == 1 && d->ooy == 0 && d->oox == 0 && d->Ho == d->Hq && d->Wo == d->Wq &&
                       in_elems < (1ll << 31) && M * d->ldy < (1ll << 32) && ((uintptr_t)d->dy & 15) == 0 &&
                       ((uintptr_t)d->in1 & 15) == 0 && ((uintptr_t)d->in2 & 15) == 0;
    if (d->dtype == DM_BF16) return launch_wgrad<bf16>(p, d->splitk, v2_ok, (hipStream_t)stream);
    if (d->dtype == DM_F16) return launch_wgrad<f16>(p, d->splitk, v2_ok, (hipStream_t)stream);
    return launch_wgrad<float>(p, d->splitk, v2_ok, (hipStream_t)stream);
}
