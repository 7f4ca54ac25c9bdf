// Shared device code of the dm_conv kernels: descriptor, MFMA wrappers, LDS swizzle, the common epilogue.  Included by the three
// translation units of the family (igemm.hip: gather / pointwise kernels + dispatch; igemm_halo.hip: conv3x3_halo_kernel;
// igemm_tap4.hip: conv_tap4_halo_kernel) so that they compile in parallel.
#pragma once
// Where the halo kernels issue the three LDS-DMA pieces of a k-step (2 weight pieces, 1 halo piece): 0 = right after the barrier,
// next to the fragment reads; abc = before MFMA groups a, b, c of the k-step's 8 (4 MFMAs each; a <= b <= c keeps the vmcnt order).
#ifndef DM_HALO_DMA_POS
#define DM_HALO_DMA_POS 567
#endif
// 1: the per-tile halo kernel keeps pixel fragments in registers across the three taps of a halo column offset where the wave's
// pixels are 4 rows x 16 columns (TW = 16; igemm_halo.hip, REUSE): 108 instead of 144 ds_read_b128 per chunk and wave.  Measured r04
// (profiles/r04_ab_same_box.txt): 16x16 512->512 forward 65.0 / 66.2 / 65.7 us without, 63.7 / 65.5 / 64.1 us with — a quarter of the
// fragment reads buys 1.8 %, i.e. the LDS reads are not what the k-loop waits for.  Off: the tap order changes the fp32 summation
// order (the persistent form would have to follow for its bit-identity test) for 0.015 ms per train step.
#ifndef DM_HALO_REUSE
#define DM_HALO_REUSE 0
#endif
#include <type_traits>
#include "common.h"

namespace dmk {


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

// Split-K without a second launch (halo kernels): every split parks its raw accumulators in ws, publishes them and counts itself
// in; the split that arrives LAST (no spinning: whoever it is) adds the others' partials to the accumulators it still holds in
// registers and runs the ordinary epilogue.  Returns false for the splits that are done.
// r02 published through `__threadfence()` on both sides (buffer_wbl2 + buffer_inv: the XCD's whole L2 written back and the L1
// invalidated per workgroup): the train step got 2 ms SLOWER.  r03: the cross-XCD hand-off of MI355X_MICROARCH.md (hand-off table,
// first row): the partials are stored `sc1` (write-through, the line is dropped from the storing XCD's L2), every storing wave waits
// for its stores, ONE lane adds to the tile's counter behind a workgroup barrier (agent-scope atomic), the workgroup whose add
// came last — told by the value the add returned — loads the partials with `sc1` loads behind a workgroup barrier that lane joins.
// No fence, no L2 write-back.  (The workspace lines cannot sit stale in the reader's L2: nothing on that XCD reads them in this
// launch before the hand-off, and a launch boundary invalidates what earlier launches left.)
// r04: the partials travel as BUFFER stores / loads with the sc1 cache policy (aux = 16) — compiler-visible, so the
// `s_waitcnt` in front of each use is the compiler's own, and no address registers (lane offset in one VGPR, image offset scalar).  r03 used
// inline-asm `global_load/store … sc1`, whose memory operations the compiler's waitcnt tracking cannot see: correct only as long as no
// copy or spill of a loaded register lands between the asm load and the hand-written wait.
constexpr int SC1_POLICY = 16;

__device__ __forceinline__ bool splitk_last_arriver(const ConvP& p, f32x4 (&acc)[4][4], char* smem, int split, int ntiles2, int tile_id,
                                                    int half_tile, int wave4, int tid, int lane) {
    const __amdgpu_buffer_rsrc_t rws = __builtin_amdgcn_make_buffer_rsrc((void*)p.ws, 0, 0x7fffffff, 0x00020000);
    {
        const int soff = (((split * ntiles2 + half_tile) * 4 + wave4) * 16) * 1024;       // bytes; the workspace is < 2^31
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[nt][mt]), rws, lane * 16, soff + (nt * 4 + mt) * 1024, SC1_POLICY);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's partials have left for memory
    __syncthreads();                                   // ... and every other wave's of this workgroup
    int* flag = (int*)(smem + 16384);
    if (tid == 0) {
        const int old = __hip_atomic_fetch_add(p.counters + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old == p.splits - 1;
        // everyone has arrived: ready for the next launch on this stream (an agent-scope atomic store, like the adds that will meet it:
        // a plain store would sit in this XCD's L2 until the launch boundary's write-back — sufficient today, not by contract)
        if (last) __hip_atomic_store(p.counters + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = last;
    }
    __syncthreads();
    if (!*flag) return false;
    // fold in the FIXED order k = 0 .. splits - 1 (this workgroup's own partial is read back like the others): the result does not
    // depend on which split arrived last, and equals the two-launch epilogue's bit for bit (run-to-run reproducibility of the step)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < p.splits; ++k) {
        const int soff = (((k * ntiles2 + half_tile) * 4 + wave4) * 16) * 1024;
        u32x4 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rws, lane * 16, soff + i * 1024, SC1_POLICY);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[nt][mt] += __builtin_bit_cast(f32x4, v[nt * 4 + mt]);
    }
    __syncthreads();                                   // the flag word is LDS the epilogue reuses
    return true;
}

static __device__ __attribute__((aligned(128))) unsigned int g_zero_page[64];  // source of every padded 16-B vector (v2)

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

// PHASE: bit 0 = this call's column sums go to the LDS fold, bit 1 = barrier + publish the fold (both `wm` slots).  3 = the whole
// epilogue in one call (every kernel but the four-wave halo form, whose waves hold two `wm` rows and call it once per row: 1, then 3)
template <typename T, int BN, int PHASE = 3>
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

    if (p.psum && (PHASE & 2)) {                           // after the stores: the barrier's wait for the slowest wave hides behind them
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

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---- halo-resident kernels (igemm_halo.hip, igemm_tap4.hip)
typedef __attribute__((address_space(3))) void* lds_dst;
constexpr int HALO_PIECES = 54;                       // 1-KiB pieces per halo buffer (TW=64: 6x72/8, TW=16: 18x24/8)
constexpr int HALO_BYTES = HALO_PIECES * 1024;
constexpr int WSTAGE = 128 * ROWB;                    // one weight stage: 128 rows (n) x 128 B
constexpr int HALO_LDS = 2 * HALO_BYTES + 3 * WSTAGE; // 159,744 B
constexpr unsigned OOB = 0x80000000u;
constexpr int SRD_FLAGS = 0x00020000;

// ---- host entry points across the translation units
extern int g_splitk_inkernel;        // igemm.hip
extern int g_last_path;              // igemm.hip: kernel family of the last dm_conv launch
int launch_halo_any(const ConvP& p, bool is_f16, hipStream_t st);                          // igemm_halo.hip
bool halo4_ok(const ConvP& p);                                                             // igemm_halo4.hip (r04 experiment: one wave per SIMD)
int launch_halo4_any(const ConvP& p, bool is_f16, hipStream_t st);
extern int g_last_persist;
extern int g_conv_persist;           // igemm.hip: 1 = persistent halo kernel where halo_persist_ok()
bool halo_persist_ok(const ConvP& p, int tiles, int ncu);                                  // igemm_halo_p.hip
int launch_halo_persist_any(const ConvP& p, bool is_f16, int tiles, int ncu, hipStream_t st);
int launch_tap4_any(const ConvP& p, bool is_f16, bool s2, hipStream_t st);                 // igemm_tap4.hip
extern int g_packtap;                                                                      // igemm.hip
bool packtap_ok(const ConvP& p);                                                           // igemm_skinny.hip
int launch_packtap_any(const ConvP& p, bool is_f16, hipStream_t st);
bool narrow_ok(const ConvP& p);
int launch_narrow_any(const ConvP& p, bool is_f16, hipStream_t st);
int launch_splitk_epilogue128(const ConvP& q, bool is_f16, unsigned grid, hipStream_t st);  // igemm.hip

}  // namespace dmk
