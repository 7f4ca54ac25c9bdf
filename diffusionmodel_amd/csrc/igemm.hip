// Implicit-GEMM gather convolution on the CDNA4 matrix cores (forward, input-gradient, ConvTranspose,
// and — as a 1x1 "image" — the dense fp32 layers).
//
// One kernel serves every dense contraction of the denoiser (include/dm_amd.h, dm_conv): the output
// tile is 128 output pixels x BN output channels per 256-thread workgroup (4 waves as 2(m) x 2(n)),
// K runs over taps x input channels in steps of 128 bytes per row (64 bf16 / 32 f32 channels).  Both
// operands are staged through LDS as [row][128 B] images whose 16-byte vectors are XOR-swizzled with
// the row index (vec ^= row & 7), which makes the ds_read_b128 fragment reads conflict-free (every
// 16-lane group of the instruction hits 16 distinct 16-B slots of the 256-B bank row) while the
// staging writes stay whole-row contiguous.  Global loads are register-prefetched one k-step ahead and
// the LDS image is double buffered: one barrier per k-step, two MFMA k-sub-steps between barriers.
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
};

__device__ inline int lds_off(int row, int vec) { return row * ROWB + ((vec ^ (row & 7)) << 4); }

template <typename T> struct Mma;
template <> struct Mma<bf16> {
    __device__ static inline void run(const u32x4& a, const u32x4& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    __device__ static inline void run(const u32x4& a, const u32x4& b, f32x4& c) {
        const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
#pragma unroll
        for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[j], fb[j], c, 0, 0, 0);
    }
};

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
    // XCD-aware remap of the linear workgroup id (bijective for any grid size)
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int mb = bid / nb_n, nb = bid - mb * nb_n;
    const int m0 = mb * BM, n0 = nb * BN;
    const int C = p.C1 + p.C2;
    const T* in1 = (const T*)p.in1;
    const T* in2 = (const T*)p.in2;
    const T* wgt = (const T*)p.w;

    // ---- per-thread staging assignment: A rows (tid>>3) + 32*i, vector tid&7
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
    // a second workgroup on the CU covers the rest of the HBM/L2 latency
    u32x4 ra0[4], rb0[BV];
    // tap state of the NEXT load (advanced incrementally: no division in the loop)
    int ld_t = 0, ld_ky = 0, ld_kx = 0, ld_c0 = 0;
    auto gload = [&](u32x4 (&ra)[4], u32x4 (&rb)[BV]) {
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
    auto sstore = [&](int buf, const u32x4 (&ra)[4], const u32x4 (&rb)[BV]) {
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

    auto compute = [&](int buf) {
        const char* sA = smem + buf * (A_BYTES + B_BYTES);
        const char* sB = sA + A_BYTES;
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
    };

    const int csteps = (C + BK - 1) / BK;
    const int nsteps = p.T * csteps;
    // (A variant with two register sets — loads two k-steps ahead — was measured on MI355X: 198 VGPRs drop the
    //  kernel to one workgroup per CU and the conv runs 30 % slower; one set + two workgroups per CU wins.)
    gload(ra0, rb0);
    sstore(0, ra0, rb0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int cur = s & 1;
        const bool more = s + 1 < nsteps;
        if (more) gload(ra0, rb0);                 // step s+1 in flight while LDS[cur] (step s) is consumed
        compute(cur);
        if (more) sstore(cur ^ 1, ra0, rb0);
        __syncthreads();
    }

    // ---- epilogue: z = acc*scale + shift ; optional column statistics ; activation ; store
    float sc[NT][4], sh[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + wn * (BN / 2) + nt * 16 + fg * 4 + r;
            sc[nt][r] = (p.scale && n < p.N) ? p.scale[n] : 1.f;
            sh[nt][r] = (p.shift && n < p.N) ? p.shift[n] : 0.f;
        }
    bool m_ok[4];
    size_t orow[4];
    int ob[4], oy[4], ox[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = m0 + wm * 64 + mt * 16 + fr;
        m_ok[mt] = m < p.M;
        const int mm = m_ok[mt] ? m : 0;
        const int qx = mm % p.Wq, tq = mm / p.Wq;
        const int qy = tq % p.Hq;
        ob[mt] = tq / p.Hq;
        oy[mt] = qy * p.osy + p.ooy;
        ox[mt] = qx * p.osx + p.oox;
        orow[mt] = ((size_t)ob[mt] * p.Ho + oy[mt]) * p.Wo + ox[mt];
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
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
                if (fr == 0) {
                    const int nl = wn * (BN / 2) + nt * 16 + fg * 4 + r;
                    sred[(wm * 2 + 0) * BN + nl] = s1;
                    sred[(wm * 2 + 1) * BN + nl] = s2;
                }
            }
        __syncthreads();
        if (tid < BN && n0 + tid < p.N) {
            p.psum[(size_t)mb * p.N + n0 + tid] = sred[0 * BN + tid] + sred[2 * BN + tid];
            p.psq[(size_t)mb * p.N + n0 + tid] = sred[1 * BN + tid] + sred[3 * BN + tid];
        }
    }

    const bool vec_ok = ((p.ldc | p.coff) & 3) == 0 && !p.out_nchw;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        if (!m_ok[mt]) continue;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int nb4 = n0 + wn * (BN / 2) + nt * 16 + fg * 4;
            if (nb4 >= p.N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = act_apply(acc[nt][mt][r], p.act);
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
                        bf16x4 q = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                        *(bf16x4*)o = q;
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

template <typename T>
int launch_conv(const ConvP& p, hipStream_t st) {
    const int mblocks = cdiv(p.M, BM);
    int bn = 128;
    if (p.N <= 32) bn = 32;
    else if (p.N <= 64) bn = 64;
    else if ((int64_t)mblocks * cdiv(p.N, 128) < 256) bn = 64;  // small problems: more, smaller tiles
    const int64_t grid = (int64_t)mblocks * cdiv(p.N, bn);
    if (bn == 128) hipLaunchKernelGGL((conv_igemm_kernel<T, 128>), dim3((unsigned)grid), dim3(256), 0, st, p);
    else if (bn == 64) hipLaunchKernelGGL((conv_igemm_kernel<T, 64>), dim3((unsigned)grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv_igemm_kernel<T, 32>), dim3((unsigned)grid), dim3(256), 0, st, p);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

}  // namespace

extern "C" int dm_conv(const DmConv* d, dm_stream_t stream) {
    DM_CHECK_ARG(d != nullptr, "dm_conv: null descriptor");
    const int ve = d->dtype == DM_BF16 ? 8 : 4;
    DM_CHECK_ARG(d->dtype == DM_F32 || d->dtype == DM_BF16, "dm_conv: bad dtype %d", d->dtype);
    DM_CHECK_ARG(d->in1 && d->w && d->out, "dm_conv: null tensor pointer");
    DM_CHECK_ARG(d->C1 > 0 && d->C1 % ve == 0 && d->C2 >= 0 && d->C2 % ve == 0, "dm_conv: C1=%d C2=%d must be multiples of %d", d->C1, d->C2, ve);
    DM_CHECK_ARG(d->C2 == 0 || d->in2, "dm_conv: C2 > 0 but in2 is null");
    DM_CHECK_ARG(d->ldw % ve == 0 && d->ldw >= d->T * (d->C1 + d->C2), "dm_conv: ldw=%d invalid for T=%d C=%d", d->ldw, d->T, d->C1 + d->C2);
    DM_CHECK_ARG(d->B > 0 && d->Hi > 0 && d->Wi > 0 && d->Hq > 0 && d->Wq > 0 && d->T > 0 && d->KW > 0 && d->N > 0, "dm_conv: non-positive extent");
    DM_CHECK_ARG((d->Hq - 1) * d->osy + d->ooy < d->Ho && (d->Wq - 1) * d->osx + d->oox < d->Wo && d->ooy >= 0 && d->oox >= 0, "dm_conv: output mapping exceeds Ho/Wo");
    DM_CHECK_ARG(d->out_nchw_f32 || (d->ldc >= d->coff + d->N && d->coff >= 0), "dm_conv: ldc=%d < coff+N=%d", d->ldc, d->coff + d->N);
    DM_CHECK_ARG((d->psum == nullptr) == (d->psq == nullptr), "dm_conv: psum/psq must both be set or both null");
    DM_CHECK_ARG(((uintptr_t)d->in1 & 15) == 0 && ((uintptr_t)d->in2 & 15) == 0 && ((uintptr_t)d->w & 15) == 0, "dm_conv: tensors must be 16-byte aligned");
    const int64_t M = (int64_t)d->B * d->Hq * d->Wq;
    DM_CHECK_ARG(M < (1ll << 31), "dm_conv: M too large");
    ConvP p;
    p.in1 = (const char*)d->in1; p.in2 = (const char*)d->in2; p.w = (const char*)d->w;
    p.scale = d->scale; p.shift = d->shift; p.out = (char*)d->out; p.psum = d->psum; p.psq = d->psq;
    p.act = d->act; p.out_nchw = d->out_nchw_f32;
    p.B = d->B; p.Hi = d->Hi; p.Wi = d->Wi; p.C1 = d->C1; p.C2 = d->C2; p.Hq = d->Hq; p.Wq = d->Wq; p.sy = d->sy; p.sx = d->sx;
    p.T = d->T; p.KW = d->KW; p.ty = d->ty; p.tx = d->tx; p.oy0 = d->oy0; p.ox0 = d->ox0;
    p.Ho = d->Ho; p.Wo = d->Wo; p.osy = d->osy; p.osx = d->osx; p.ooy = d->ooy; p.oox = d->oox;
    p.N = d->N; p.ldw = d->ldw; p.ldc = d->ldc; p.coff = d->coff; p.M = (int)M;
    if (d->dtype == DM_BF16) return launch_conv<bf16>(p, (hipStream_t)stream);
    return launch_conv<float>(p, (hipStream_t)stream);
}
