// Implicit-GEMM gather convolution on the CDNA4 matrix cores (forward, input-gradient, ConvTranspose).
//
// One kernel serves every dense contraction of the denoiser (include/dm_amd.h, dm_conv): the output
// tile is 128 output pixels x BN output channels per 256-thread workgroup (4 waves as 2(m) x 2(n)),
// K runs over taps x input channels in steps of 64 bytes per row.  Both operands are staged through
// LDS as [row][64 B] images with an XOR swizzle that makes the ds_read_b128 fragment reads
// conflict-free (16-lane groups hit 16 distinct 16-B slots), register-prefetched one k-step ahead
// and double buffered (one barrier per k-step).
//
// MFMA orientation is "swapped": A operand = packed weights (row i = output channel), B operand =
// gathered input pixels (column j = output pixel), so every lane ends up with 4 consecutive output
// channels of one pixel -> one 8-byte (bf16) / 16-byte (f32) NHWC store per 16x16 tile.
//   bf16: v_mfma_f32_16x16x32_bf16, a lane's fragment = 8 consecutive k of its row.
//   f32 : v_mfma_f32_16x16x4_f32 x 4 per 16-B fragment; element j of the lane group g is k = 4g + j on
//         both operands, so the four MFMAs together cover the 16 k of the step exactly once.
//         (f32 MFMA is a k-ordered fmaf chain: exact fp32, which is what the 1e-4 parity mode needs.)
#include "common.h"

namespace {

constexpr int BM = 128;
constexpr int ROWB = 64;  // bytes per LDS row per k-step (4 x 16-B vectors)

struct ConvP {
    const char* in1; const char* in2; const char* w;
    const float* scale; const float* shift;
    char* out; float* psum; float* psq;
    int act, out_nchw;
    int B, Hi, Wi, C1, C2, Hq, Wq, sy, sx, T, KW, ty, tx, oy0, ox0;
    int Ho, Wo, osy, osx, ooy, oox, N, ldw, ldc, coff, M;
};

__device__ inline int lds_off(int row, int vec) { return row * ROWB + ((vec ^ ((row >> 1) & 3)) << 4); }

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
    constexpr int BK = 4 * VE;
    constexpr int NT = BN / 32;              // 16-wide channel tiles per wave
    constexpr int BV = (BN * 4 + 255) / 256;  // weight vectors per thread per k-step
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    __shared__ __attribute__((aligned(16))) char smem[2 * (A_BYTES + B_BYTES)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int fr = lane & 15, fg = lane >> 4;
    const int nb_n = (p.N + BN - 1) / BN;
    const int mb = blockIdx.x / nb_n, nb = blockIdx.x - mb * nb_n;
    const int m0 = mb * BM, n0 = nb * BN;
    const int C = p.C1 + p.C2;
    const T* in1 = (const T*)p.in1;
    const T* in2 = (const T*)p.in2;
    const T* wgt = (const T*)p.w;

    // ---- per-thread staging assignment: A rows (tid>>2) and +64, vector tid&3
    const int sv = tid & 3;
    int a_b[2], a_y[2], a_x[2];
    bool a_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + (tid >> 2) + 64 * i;
        a_ok[i] = m < p.M;
        const int mm = a_ok[i] ? m : 0;
        const int qx = mm % p.Wq, tq = mm / p.Wq;
        const int qy = tq % p.Hq;
        a_b[i] = tq / p.Hq;
        a_y[i] = qy * p.sy + p.oy0;
        a_x[i] = qx * p.sx + p.ox0;
    }

    u32x4 ra[2], rb[BV];
    auto gload = [&](int t, int c0) {
        const int ky = t / p.KW, kx = t - ky * p.KW;
        const int dy = ky * p.ty, dx = kx * p.tx;
        const int c = c0 + sv * VE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int iy = a_y[i] + dy, ix = a_x[i] + dx;
            const bool ok = a_ok[i] && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi && c < C;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (ok) {
                const size_t pix = ((size_t)a_b[i] * p.Hi + iy) * p.Wi + ix;
                const T* src = (c < p.C1) ? (in1 + pix * p.C1 + c) : (in2 + pix * p.C2 + (c - p.C1));
                v = *(const u32x4*)src;
            }
            ra[i] = v;
        }
#pragma unroll
        for (int j = 0; j < BV; ++j) {
            const int idx = tid + 256 * j;
            const int row = idx >> 2;
            const int n = n0 + row;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (row < BN && n < p.N && c < C) v = *(const u32x4*)(wgt + (size_t)n * p.ldw + (size_t)t * C + c);
            rb[j] = v;
        }
    };
    auto sstore = [&](int buf) {
        char* sA = smem + buf * (A_BYTES + B_BYTES);
        char* sB = sA + A_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) *(u32x4*)(sA + lds_off((tid >> 2) + 64 * i, sv)) = ra[i];
#pragma unroll
        for (int j = 0; j < BV; ++j) {
            const int row = (tid + 256 * j) >> 2;
            if (row < BN) *(u32x4*)(sB + lds_off(row, sv)) = rb[j];
        }
    };

    f32x4 acc[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int csteps = (C + BK - 1) / BK;
    const int nsteps = p.T * csteps;
    int t_next = 0, c_next = 0;
    gload(0, 0);
    sstore(0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int cur = s & 1;
        const bool more = s + 1 < nsteps;
        if (more) {
            c_next += BK;
            if (c_next >= C) { c_next = 0; ++t_next; }
            gload(t_next, c_next);
        }
        const char* sA = smem + cur * (A_BYTES + B_BYTES);
        const char* sB = sA + A_BYTES;
        u32x4 fb[4], fa[NT];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) fb[mt] = *(const u32x4*)(sA + lds_off(wm * 64 + mt * 16 + fr, fg));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) fa[nt] = *(const u32x4*)(sB + lds_off(wn * (BN / 2) + nt * 16 + fr, fg));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) Mma<T>::run(fa[nt], fb[mt], acc[nt][mt]);
        if (more) sstore(cur ^ 1);
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
