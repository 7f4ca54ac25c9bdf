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
#include <type_traits>
#include "common.h"

namespace {

constexpr int BM = 128;
constexpr int ROWB = 128;  // bytes per LDS row per k-step (8 x 16-B vectors)

struct ConvP {
    const char* in1; const char* in2; const char* w;
    const float* scale; const float* shift;
    char* out; float* psum; float* psq;
    int act, out_nchw;
    int B, Hi, Wi, C1, C2, Hq, Wq, sy, sx, T, KW, ty, tx, oy0, ox0;
    int Ho, Wo, osy, osx, ooy, oox, N, ldw, ldc, coff, M;
    // split-K (LDS-DMA kernels): `splits` workgroups share an output tile, each takes `kper` k-steps (gather kernel) or
    // channel chunks (halo kernel) and leaves its accumulators in ws; splitk_epilogue_kernel adds them up and finishes
    int splits, kper;
    float* ws;
    int* counters;                               // halo kernels: arrival counter per output tile (the last split to arrive finishes the tile)
    // four-tap kernel, S2 = false: `npar` (1 or 4) output-parity classes in ONE launch — class q takes workgroups [q, q + 1) * grid / npar,
    // its own transposed weight pack, tap offsets and output offsets (the four input-gradient launches of a 4x4 / stride-2 layer)
    int npar;
    const char* w4[4];
    int oy4[4], ox4[4], ooy4[4], oox4[4];
    int B2;                                      // batch of the second source (in2 is read at sample b % B2); == B unless broadcast
    int stat_slots;                              // 0: psum / psq are [tiles][N] partial rows; S > 0: [S][N] accumulators, tile mb adds into slot mb % S
    const char* addend;                          // optional tensor of the output's layout / dtype added after the activation (gradient of a forked tensor)
};

// raw accumulators of one 128 x BN tile as they sit in the registers: [tile][wave][nt][mt][lane] float4 (1 KiB per store)
template <int BN>
__device__ __forceinline__ void store_partial(const ConvP& p, const f32x4 (&acc)[BN / 32][4], int split, int tiles, int tile, int wave, int lane) {
    constexpr int NT = BN / 32;
    f32x4* dst = (f32x4*)p.ws + ((((size_t)split * tiles + tile) * 4 + wave) * (NT * 4)) * 64 + lane;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) dst[(nt * 4 + mt) * 64] = acc[nt][mt];
}

// Split-K without a second launch (halo kernels): every split parks its raw accumulators in ws (store_partial), publishes them
// (release fence) and counts itself in; the split that arrives LAST (no spinning: whoever it is) adds the others' partials to
// the accumulators it still holds in registers and runs the ordinary epilogue.  Returns false for the splits that are done.
__device__ __forceinline__ bool splitk_last_arriver(const ConvP& p, f32x4 (&acc)[4][4], char* smem, int split, int ntiles2, int tile_id,
                                                    int half_tile, int wave4, int tid, int lane) {
    store_partial<128>(p, acc, split, ntiles2, half_tile, wave4, lane);
    __threadfence();                                   // the partials are visible device-wide before the count says so
    __syncthreads();
    int* flag = (int*)(smem + 16384);
    if (tid == 0) {
        const int old = atomicAdd(p.counters + tile_id, 1);
        const int last = old == p.splits - 1;
        if (last) p.counters[tile_id] = 0;             // everyone has arrived: ready for the next launch on this stream
        *flag = last;
    }
    __syncthreads();
    if (!*flag) return false;
    __threadfence();                                   // acquire: the other splits' partials (written on other XCDs) are read from memory
    for (int k = 0; k < p.splits; ++k) {
        if (k == split) continue;
        const f32x4* src = (const f32x4*)p.ws + ((((size_t)k * ntiles2 + half_tile) * 4 + wave4) * 16) * 64 + lane;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const f32x4 v = __builtin_nontemporal_load(src + (nt * 4 + mt) * 64);
                acc[nt][mt] += v;
            }
    }
    __syncthreads();                                   // the flag word is LDS the epilogue reuses
    return true;
}

__device__ __attribute__((aligned(128))) unsigned int g_zero_page[64];  // source of every padded 16-B vector (v2)

__device__ inline int lds_off(int row, int vec) { return row * ROWB + ((vec ^ (row & 7)) << 4); }

__device__ inline int remap_xcd(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

template <typename T> struct Mma;
template <> struct Mma<bf16> {
    __device__ static inline void run(const u32x4& a, const u32x4& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma<f16> {
    __device__ static inline void run(const u32x4& a, const u32x4& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    __device__ static inline void run(const u32x4& a, const u32x4& b, f32x4& c) {
        const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
#pragma unroll
        for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[j], fb[j], c, 0, 0, 0);
    }
};

// one k-step (two MFMA sub-steps) of a wave's 64 x (BN/2) sub-tile out of the stage at sA / sB
template <typename T, int BN>
__device__ __forceinline__ void mma_stage(const char* sA, const char* sB, int wm, int wn, int fr, int fg, f32x4 (&acc)[BN / 32][4]) {
    constexpr int NT = BN / 32;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
        u32x4 fb[4], fa[NT];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) fb[mt] = *(const u32x4*)(sA + lds_off(wm * 64 + mt * 16 + fr, sub * 4 + fg));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) fa[nt] = *(const u32x4*)(sB + lds_off(wn * (BN / 2) + nt * 16 + fr, sub * 4 + fg));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) Mma<T>::run(fa[nt], fb[mt], acc[nt][mt]);
    }
}

// epilogue: z = acc*scale + shift ; optional per-block column statistics ; activation ; store
// sum over the 16 lanes of a DPP row, result in every lane: quad butterflies, then the two mirrors (v_add_f32 with DPP operands —
// the __shfl_xor form compiled to 128 ds_bpermute_b32 per lane in the statistics epilogue)
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    return v;
}

// activation with a compile-time selector (ACTC >= 0) or the runtime value (ACTC < 0)
template <typename T, int ACTC>
__device__ __forceinline__ float act_c(float x, int act) {
    if constexpr (ACTC == DM_ACT_NONE) return x;
    else if constexpr (ACTC >= 0) return act_apply_t<T>(x, ACTC);
    else return act_apply_t<T>(x, act);
}

// second half of the epilogue: activation, addend, stores
template <typename T, int BN, int ACTC>
__device__ __forceinline__ void conv_store(const ConvP& p, f32x4 (&acc)[BN / 32][4], const bool (&m_ok)[4], const size_t (&orow)[4], const int (&ob)[4],
                                           const int (&oy)[4], const int (&ox)[4], int wn, int fg, int n0) {
    constexpr int NT = BN / 32;
    const bool vec_ok = ((p.ldc | p.coff) & 3) == 0 && !p.out_nchw;
    const int act = ACTC >= 0 ? ACTC : p.act;
    bool applied = false;
    // out = act(z) + addend: the other consumer's gradient of a tensor used twice (ops.GradFork)
    if constexpr (sizeof(T) == 2 && NT % 2 == 0) {
        // 16-bit NHWC, both 16-channel blocks of a pair inside N and 16-byte aligned: a lane holds 4 channels (8 B) of blocks a and b;
        // one v_permlane16_swap per dword trades block b of the even 16-lane rows for block a of the odd rows, so that every lane
        // ends up with 8 consecutive channels of ONE block -> one 16-byte store per pair instead of two 8-byte ones (half the store
        // instructions, 64-byte instead of 32-byte segments per pixel).  The addend is fetched the same way — one 16-byte load at the
        // lane's store address — and taken back to the accumulator layout by the same swap (it is its own inverse), so the sum is
        // formed in fp32 and rounded once (8-byte loads at the accumulator positions cost +39 us on the 64x64 1x1 input gradient).
        const bool wide = vec_ok && ((p.ldc | p.coff) & 7) == 0 && n0 + wn * (BN / 2) + NT * 16 <= p.N && ((uintptr_t)p.out & 15) == 0 &&
                          ((uintptr_t)p.addend & 15) == 0;
        if (wide) {                                        // workgroup-uniform
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
                for (int pr = 0; pr < NT / 2; ++pr) {
                    typedef typename V16<T>::x2 t2;
                    const int blk = 2 * pr + (fg & 1);     // even 16-lane rows store block a, odd rows block b
                    const int nb8 = n0 + wn * (BN / 2) + blk * 16 + (fg >> 1) * 8;
                    float ada[4] = {0.f, 0.f, 0.f, 0.f}, adb[4] = {0.f, 0.f, 0.f, 0.f};
                    if (p.addend) {                        // uniform
                        u32x4 qd = {0u, 0u, 0u, 0u};
                        if (m_ok[mt]) qd = *(const u32x4*)((const T*)p.addend + orow[mt] * p.ldc + p.coff + nb8);
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const auto sw = __builtin_amdgcn_permlane16_swap(qd[h], qd[2 + h], false, false);
                            const t2 a2 = __builtin_bit_cast(t2, (unsigned)sw[0]), b2 = __builtin_bit_cast(t2, (unsigned)sw[1]);
                            ada[2 * h] = (float)a2[0]; ada[2 * h + 1] = (float)a2[1];
                            adb[2 * h] = (float)b2[0]; adb[2 * h + 1] = (float)b2[1];
                        }
                    }
                    unsigned qa[2], qb[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const t2 a2 = {(T)(act_c<T, ACTC>(acc[2 * pr][mt][2 * h], act) + ada[2 * h]), (T)(act_c<T, ACTC>(acc[2 * pr][mt][2 * h + 1], act) + ada[2 * h + 1])};
                        const t2 b2 = {(T)(act_c<T, ACTC>(acc[2 * pr + 1][mt][2 * h], act) + adb[2 * h]), (T)(act_c<T, ACTC>(acc[2 * pr + 1][mt][2 * h + 1], act) + adb[2 * h + 1])};
                        const auto sw = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a2), __builtin_bit_cast(unsigned, b2), false, false);
                        qa[h] = sw[0];                     // even rows: own a      | odd rows: partner's b
                        qb[h] = sw[1];                     // even rows: partner's a | odd rows: own b
                    }
                    if (!m_ok[mt]) continue;
                    T* o = (T*)p.out + orow[mt] * p.ldc + p.coff + nb8;
                    *(u32x4*)o = (u32x4){qa[0], qa[1], qb[0], qb[1]};
                }
            }
            return;
        }
    }
    if (p.addend) {                                        // narrow paths: the addend at the accumulator positions
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            if (!m_ok[mt]) continue;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int nb4 = n0 + wn * (BN / 2) + nt * 16 + fg * 4;
                const T* a = (const T*)p.addend + orow[mt] * p.ldc + p.coff + nb4;
                if (vec_ok && nb4 + 3 < p.N) {
                    if constexpr (sizeof(T) == 4) {
                        const f32x4 q = *(const f32x4*)a;
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[nt][mt][r] = act_c<T, ACTC>(acc[nt][mt][r], act) + q[r];
                    } else {
                        const typename V16<T>::x4 q = *(const typename V16<T>::x4*)a;
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[nt][mt][r] = act_c<T, ACTC>(acc[nt][mt][r], act) + (float)q[r];
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[nt][mt][r] = act_c<T, ACTC>(acc[nt][mt][r], act) + (nb4 + r < p.N ? Elem<T>::ld(a + r) : 0.f);
                }
            }
        }
        applied = true;                                    // the activation went in before the addend
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        if (!m_ok[mt]) continue;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int nb4 = n0 + wn * (BN / 2) + nt * 16 + fg * 4;
            if (nb4 >= p.N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = applied ? acc[nt][mt][r] : act_c<T, ACTC>(acc[nt][mt][r], act);
            if (p.out_nchw) {
                float* o = (float*)p.out;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (nb4 + r < p.N) o[(((size_t)ob[mt] * p.N + nb4 + r) * p.Ho + oy[mt]) * p.Wo + ox[mt]] = v[r];
            } else {
                T* o = (T*)p.out + orow[mt] * p.ldc + p.coff + nb4;
                if (vec_ok && nb4 + 3 < p.N) {
                    if constexpr (sizeof(T) == 4) {
                        *(f32x4*)o = (f32x4){v[0], v[1], v[2], v[3]};
                    } else {
                        typedef typename V16<T>::x4 t4;
                        t4 q = {(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
                        *(t4*)o = q;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (nb4 + r < p.N) Elem<T>::st(o + r, v[r]);
                }
            }
        }
    }
}

template <typename T, int BN>
__device__ __forceinline__ void conv_epilogue(const ConvP& p, f32x4 (&acc)[BN / 32][4], char* smem, int tid, int wm, int wn, int fr,
                                              int fg, int mb, int m0, int n0) {
    constexpr int NT = BN / 32;
    float sc[NT][4], sh[NT][4];
    // the lane's 4 consecutive channels of a block in one 16-byte load when the vectors are aligned and inside N (16 + 16 dword loads
    // per lane otherwise, and the first thing the epilogue waits for)
    const bool sv4 = (((uintptr_t)p.scale | (uintptr_t)p.shift) & 15) == 0 && (n0 & 3) == 0;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int nb4 = n0 + wn * (BN / 2) + nt * 16 + fg * 4;
        if (sv4 && nb4 + 3 < p.N) {
            const f32x4 a = p.scale ? *(const f32x4*)(p.scale + nb4) : (f32x4){1.f, 1.f, 1.f, 1.f};
            const f32x4 b = p.shift ? *(const f32x4*)(p.shift + nb4) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) { sc[nt][r] = a[r]; sh[nt][r] = b[r]; }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = nb4 + r;
                sc[nt][r] = (p.scale && n < p.N) ? p.scale[n] : 1.f;
                sh[nt][r] = (p.shift && n < p.N) ? p.shift[n] : 0.f;
            }
        }
    }
    bool m_ok[4];
    size_t orow[4];
    int ob[4], oy[4], ox[4];
    // output pixel = GEMM row for the stride-1 NHWC layers (most launches): no divisions by the image extents (8 per lane otherwise)
    const bool ident = p.osy == 1 && p.osx == 1 && p.ooy == 0 && p.oox == 0 && p.Ho == p.Hq && p.Wo == p.Wq && !p.out_nchw;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = m0 + wm * 64 + mt * 16 + fr;
        m_ok[mt] = m < p.M;
        const int mm = m_ok[mt] ? m : 0;
        if (ident) {
            ob[mt] = oy[mt] = ox[mt] = 0;
            orow[mt] = (size_t)mm;
        } else {
            const int qx = mm % p.Wq, tq = mm / p.Wq;
            const int qy = tq % p.Hq;
            ob[mt] = tq / p.Hq;
            oy[mt] = qy * p.osy + p.ooy;
            ox[mt] = qx * p.osx + p.oox;
            orow[mt] = ((size_t)ob[mt] * p.Ho + oy[mt]) * p.Wo + ox[mt];
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[nt][mt][r] = acc[nt][mt][r] * sc[nt][r] + sh[nt][r];

    if (p.psum) {
        float* sred = (float*)smem;  // [2 (wm)][2 (sum,sq)][BN]
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    if (m_ok[mt]) { const float z = acc[nt][mt][r]; s1 += z; s2 += z * z; }
                s1 = row16_sum(s1);
                s2 = row16_sum(s2);
                if (fr == 0) {
                    const int nl = wn * (BN / 2) + nt * 16 + fg * 4 + r;
                    sred[(wm * 2 + 0) * BN + nl] = s1;
                    sred[(wm * 2 + 1) * BN + nl] = s2;
                }
            }
    }

    // the activation is dispatched ONCE per workgroup: with a runtime `act` inside the 64 per-value expressions the epilogue carried
    // ~2100 scalar branches (three per value, each skipping a GELU / sigmoid body) through 160 KB of code
    if (p.act == DM_ACT_NONE) conv_store<T, BN, DM_ACT_NONE>(p, acc, m_ok, orow, ob, oy, ox, wn, fg, n0);
    else if (p.act == DM_ACT_GELU) conv_store<T, BN, DM_ACT_GELU>(p, acc, m_ok, orow, ob, oy, ox, wn, fg, n0);
    else conv_store<T, BN, -1>(p, acc, m_ok, orow, ob, oy, ox, wn, fg, n0);

    if (p.psum) {                                          // after the stores: the barrier's wait for the slowest wave hides behind them
        const float* sred = (const float*)smem;
        __syncthreads();
        if (tid < BN && n0 + tid < p.N) {
            if (p.stat_slots > 0) {                        // a few slots the consuming kernel folds itself (no finalize launch)
                // fp64 accumulators: precision of the old fp64 fold over the partial rows, and the arrival order moves nothing visible
                atomicAdd((double*)p.psum + (size_t)(mb % p.stat_slots) * p.N + n0 + tid, (double)(sred[0 * BN + tid] + sred[2 * BN + tid]));
                atomicAdd((double*)p.psq + (size_t)(mb % p.stat_slots) * p.N + n0 + tid, (double)(sred[1 * BN + tid] + sred[3 * BN + tid]));
            } else {
                p.psum[(size_t)mb * p.N + n0 + tid] = sred[0 * BN + tid] + sred[2 * BN + tid];
                p.psq[(size_t)mb * p.N + n0 + tid] = sred[1 * BN + tid] + sred[3 * BN + tid];
            }
        }
    }
}

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

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

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
typedef __attribute__((address_space(3))) void* lds_dst;

constexpr int HALO_PIECES = 54;                       // 1-KiB pieces per halo buffer (TW=64: 6x72/8, TW=16: 18x24/8)
constexpr int HALO_BYTES = HALO_PIECES * 1024;
constexpr int WSTAGE = 128 * ROWB;                    // one weight stage: 128 rows (n) x 128 B
constexpr int HALO_LDS = 2 * HALO_BYTES + 3 * WSTAGE; // 159,744 B
constexpr unsigned OOB = 0x80000000u;
constexpr int SRD_FLAGS = 0x00020000;

template <typename T, int TW, bool FLIP>
__global__ __launch_bounds__(512) void conv3x3_halo_kernel(const ConvP p) {
    // TW = 8 (8x8 images): the tile is FOUR whole images laid side by side in the halo, each with its own zero columns
    // ([0 A 0][0 B 0][0 C 0][0 D 0], 10 columns apiece): rows stay multiples of 8 pixels, a wave (64 pixels) is one image
    constexpr int TH = TW == 8 ? 8 : 256 / TW, HS = TW == 8 ? 40 : TW + 8, HR = TH + 2, NP = HR * HS / 8;
    static_assert(NP <= HALO_PIECES, "halo does not fit");
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
    issue_w(0, c_lo, 0);
    issue_w(1, c_lo, 1);
    int hdelta = HALO_BYTES;
    {
        for (int chunk = c_lo; chunk < nchunks; ++chunk) {
            const int nbuf = (chunk - c_lo + 1) & 1;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                // everything older than the previous step's group (2 weight pieces + its halo piece) has landed
                if (tap >= 1 && tap <= 7) wait_vmcnt<3>();
                else wait_vmcnt<2>();
                __builtin_amdgcn_s_barrier();
                const int t2 = (tap + 2) % 9;
                issue_w(t2, chunk + (tap + 2 >= 9 ? 1 : 0), t2 % 3);
                if (tap < 7) issue_h(tap, chunk + 1, nbuf);
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
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) Mma<T>::run(fa[nt], fb[mt], acc[nt][mt]);
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
int g_splitk_inkernel = 0;
int g_last_path = 0;  // kernel family the last dm_conv launch used: 0 = gather (conv_igemm*), 1 = conv3x3_halo_kernel

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
    hipLaunchKernelGGL((conv3x3_halo_kernel<T, TW, FLIP>), dim3((unsigned)(tiles * q.splits)), dim3(512), HALO_LDS, st, q);
    DM_LAUNCH_CHECK();
    g_last_path = 1;
    if (q.splits > 1 && q.counters == nullptr) {
        hipLaunchKernelGGL((splitk_epilogue_kernel<T, 128>), dim3((unsigned)(tiles * 2)), dim3(256), 0, st, q);
        DM_LAUNCH_CHECK();
    }
    return DM_OK;
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
        hipLaunchKernelGGL((splitk_epilogue_kernel<T, 128>), dim3((unsigned)(tiles * 2)), dim3(256), 0, st, q);
        DM_LAUNCH_CHECK();
    }
    return DM_OK;
}

// 0: not eligible; 1: the 4x4 / stride-2 / pad-1 forward form (S2); 2: a 2x2-tap stride-1 gather with offsets in {-1, 0, 1}
// (the input gradient of one output-parity class of that layer).  Output image rows of 8 (four images per tile), 16, 32 or 64 pixels.
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

template <typename T, bool S2>
int launch_tap4_tw(const ConvP& p, hipStream_t st) {
    if (p.Wq == 64) return launch_tap4<T, 64, S2>(p, st);
    if (p.Wq == 32) return launch_tap4<T, 32, S2>(p, st);
    if (p.Wq == 16) return launch_tap4<T, 16, S2>(p, st);
    return launch_tap4<T, 8, S2>(p, st);
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
            if (m4 == 1) return launch_tap4_tw<T, true>(p, st);
            if (m4 == 2) return launch_tap4_tw<T, false>(p, st);
        }
        if (g_variant == 5 && halo_eligible(p)) {
            const bool flip = p.ty < 0;
            if (p.Wi >= 64) return flip ? launch_halo<T, 64, true>(p, st) : launch_halo<T, 64, false>(p, st);
            if (p.Wi == 32) return flip ? launch_halo<T, 32, true>(p, st) : launch_halo<T, 32, false>(p, st);
            if (p.Wi == 16) return flip ? launch_halo<T, 16, true>(p, st) : launch_halo<T, 16, false>(p, st);
            return flip ? launch_halo<T, 8, true>(p, st) : launch_halo<T, 8, false>(p, st);
        }
    }
    const int mblocks = cdiv(p.M, BM);
    int bn = 128;
    if (p.N <= 32) bn = 32;
    else if (p.N <= 64) bn = 64;
    else if ((int64_t)mblocks * cdiv(p.N, 128) < 256) bn = 64;  // small problems: more, smaller tiles
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

}  // namespace

extern "C" int dm_set_conv_tap4(int on) { g_tap4 = (on & 1) != 0; g_pw = (on & 2) == 0; return DM_OK; }   // bit 1 set: pointwise kernel off too
extern "C" int dm_set_splitk_inkernel(int on) { g_splitk_inkernel = on != 0; return DM_OK; }

extern "C" int dm_set_conv_variant(int variant) {
    DM_CHECK_ARG(variant >= 1 && variant <= 5, "dm_set_conv_variant: 1 (register staging), 2..4 (LDS-DMA ring stages) or 5 (2 + halo-resident 3x3)");
    g_variant = variant;
    return DM_OK;
}

extern "C" int dm_last_conv_path(void) { return g_last_path; }
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
        if (d4[0].dtype == DM_BF16) return launch_tap4_tw<bf16, false>(q, st);
        return launch_tap4_tw<f16, false>(q, st);
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
    ConvP p;
    bool small;
    const int rc = conv_fill(d, p, small);
    if (rc != DM_OK) return rc;
    if (d->dtype == DM_BF16) return launch_conv<bf16>(p, small, (hipStream_t)stream);
    if (d->dtype == DM_F16) return launch_conv<f16>(p, small, (hipStream_t)stream);
    return launch_conv<float>(p, small, (hipStream_t)stream);
}
