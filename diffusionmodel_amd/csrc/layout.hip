// Weight repacks, layout conversion and resampling glue (all HBM-bound, NHWC).
#include "common.h"

namespace {

template <typename T>
__global__ void pack_w_kernel(const float* src, T* dst, int64_t rows /*N*T*/, int C, int Cp) {
    const int64_t total = rows * Cp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / Cp;
        const int c = (int)(i - r * Cp);
        Elem<T>::st(dst + i, c < C ? src[r * C + c] : 0.f);
    }
}

struct TapList { int32_t t[64]; };

// dst[c][tt][np] = src[n][taps[tt]][c]; 32x32 tile transpose through LDS, grid (Np/32, C/32, Tt)
template <typename T>
__global__ __launch_bounds__(256) void pack_wT_kernel(const float* src, T* dst, int N, int Tn, int C, int Tt, int Np, TapList taps) {
    __shared__ float tile[32][33];
    const int n0 = blockIdx.x * 32, c0 = blockIdx.y * 32, tt = blockIdx.z;
    const int ts = taps.t[tt];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + ty + 8 * j, c = c0 + tx;
        tile[ty + 8 * j][tx] = (n < N && c < C) ? src[((size_t)n * Tn + ts) * C + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + ty + 8 * j, n = n0 + tx;
        if (c < C && n < Np) Elem<T>::st(dst + ((size_t)c * Tt + tt) * Np + n, tile[tx][ty + 8 * j]);
    }
}

// table-driven form of pack_wT_kernel: blocks[b] = {entry, n tile, c tile, tt}, 64 x 64 tiles (16 loads in flight per thread: the
// 32 x 32 form moved 2 KB per workgroup between two barriers and ran the 213 MB re-pack at 2.3 TB/s)
__global__ __launch_bounds__(256) void pack_multi_kernel(const int64_t* __restrict__ entries, const int32_t* __restrict__ taps,
                                                         const int32_t* __restrict__ blocks, int n_blocks) {
    __shared__ float tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
    for (int b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        const int e = blocks[4 * b], n0 = blocks[4 * b + 1] * 64, c0 = blocks[4 * b + 2] * 64, tt = blocks[4 * b + 3];
        const int64_t* ent = entries + 8 * (int64_t)e;
        const float* src = (const float*)ent[0];
        const int N = (int)ent[2], Tn = (int)ent[3], C = (int)ent[4], Tt = (int)ent[5], Np = (int)ent[6], dtype = (int)ent[7] & 0xff;
        const int src16 = (int)ent[7] >> 8;                // 1 / 2: the source is the bf16 / fp16 shadow of the parameter (same element layout): half the read
        const int ts = taps[16 * e + tt];
        if (src16 != 0 && dtype != DM_F32 && (C & 7) == 0 && (Np & 7) == 0 && (N & 7) == 0 && ((ent[0] | ent[1]) & 15) == 0) {
            // 16-bit source and destination (the optimiser's shadow -> the transposed pack), rows in whole 16-byte vectors: the tile
            // moves as 16-byte loads and stores with the 16-bit transpose inside LDS (the element-wise form below issues 2-byte
            // accesses: 2.6 TB/s for the 2 x 213 MB of the step)
            unsigned short* t16 = (unsigned short*)&tile[0][0];           // [64 n][72] 16-bit (row stride 144 B: 16-byte aligned rows)
            const unsigned short* s16 = (const unsigned short*)src;
            unsigned short* d16 = (unsigned short*)ent[1];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int idx = threadIdx.x + 256 * h, n = idx >> 3, cv = idx & 7;         // n 0..63, vector 0..7 of the 64 channels
                u32x4 q = {0u, 0u, 0u, 0u};
                if (n0 + n < N && c0 + cv * 8 < C) q = *(const u32x4*)(s16 + ((size_t)(n0 + n) * Tn + ts) * C + c0 + cv * 8);
                *(u32x4*)(t16 + n * 72 + cv * 8) = q;
            }
            __syncthreads();
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int idx = threadIdx.x + 256 * h, c = idx >> 3, nv = idx & 7;         // c 0..63, vector 0..7 of the 64 n
                if (c0 + c < C && n0 + nv * 8 < Np) {
                    unsigned short e8[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) e8[k] = t16[(nv * 8 + k) * 72 + c];
                    u32x4 q = {(unsigned)e8[0] | ((unsigned)e8[1] << 16), (unsigned)e8[2] | ((unsigned)e8[3] << 16), (unsigned)e8[4] | ((unsigned)e8[5] << 16),
                               (unsigned)e8[6] | ((unsigned)e8[7] << 16)};
                    *(u32x4*)(d16 + ((size_t)(c0 + c) * Tt + tt) * Np + n0 + nv * 8) = q;
                }
            }
            __syncthreads();
            continue;
        }
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int n = n0 + ty + 4 * j, c = c0 + tx;
            const size_t i = ((size_t)n * Tn + ts) * C + c;
            v[j] = (n < N && c < C) ? (src16 == 1 ? (float)((const bf16*)src)[i] : src16 == 2 ? (float)((const f16*)src)[i] : src[i]) : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) tile[ty + 4 * j][tx] = v[j];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int c = c0 + ty + 4 * j, n = n0 + tx;
            if (c < C && n < Np) {
                const size_t o = ((size_t)c * Tt + tt) * Np + n;
                if (dtype == DM_BF16) ((bf16*)ent[1])[o] = (bf16)tile[tx][ty + 4 * j];
                else if (dtype == DM_F16) ((f16*)ent[1])[o] = (f16)tile[tx][ty + 4 * j];
                else ((float*)ent[1])[o] = tile[tx][ty + 4 * j];
            }
        }
        __syncthreads();
    }
}

__global__ void unpad_dw_kernel(const float* src, float* dst, int64_t rows, int C, int Cp, int accumulate) {
    const int64_t total = rows * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / C;
        const int c = (int)(i - r * C);
        const float v = src[r * Cp + c];
        dst[i] = accumulate ? dst[i] + v : v;
    }
}

template <typename TI, typename TO>
__global__ void cast_kernel(const TI* x, TO* y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) Elem<TO>::st(y + i, Elem<TI>::ld(x + i));
}

// NCHW fp32 (B,C,H,W) -> NHWC T (B*repeat, H, W, Cp)
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* x, T* y, int B, int C, int HW, int Cp, int repeat) {
    const int64_t total = (int64_t)B * repeat * HW * Cp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % Cp);
        const int64_t pix = i / Cp;
        const int p = (int)(pix % HW);
        const int b = (int)((pix / HW) % B);
        Elem<T>::st(y + i, c < C ? x[((size_t)b * C + c) * HW + p] : 0.f);
    }
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* x, float* y, int B, int C, int HW, int Cp) {
    const int64_t total = (int64_t)B * C * HW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int p = (int)(i % HW);
        const int c = (int)((i / HW) % C);
        const int b = (int)(i / ((int64_t)HW * C));
        y[i] = Elem<T>::ld(x + ((size_t)b * HW + p) * Cp + c);
    }
}

template <typename T, int V>
__device__ inline void ldv(const T* p, float* f) { if constexpr (V == 1) f[0] = Elem<T>::ld(p); else load_vec<T>(p, f); }
template <typename T, int V>
__device__ inline void stv(T* p, const float* f) { if constexpr (V == 1) Elem<T>::st(p, f[0]); else store_vec<T>(p, f); }

// ---- FiLM ------------------------------------------------------------------------------------
template <typename T>
__global__ void film_fwd_kernel(const T* x, const float* ce, const float* te, T* y, int B, int HW, int C) {
    const int64_t total = (int64_t)B * HW * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int b = (int)(i / ((int64_t)HW * C));
        Elem<T>::st(y + i, ce[b * C + c] * Elem<T>::ld(x + i) + te[b * C + c]);
    }
}
// grid (B, ceil(C/256)): thread owns one channel of one sample, loops over pixels
template <typename T>
__global__ void film_bwd_kernel(const T* x, const T* dy, const float* ce, T* dx, float* dce, float* dte, int HW, int C) {
    const int b = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
    if (c >= C) return;
    const float cev = ce[b * C + c];
    float s1 = 0.f, s2 = 0.f;
    for (int p = 0; p < HW; ++p) {
        const size_t i = ((size_t)b * HW + p) * C + c;
        const float g = Elem<T>::ld(dy + i);
        s1 += g * Elem<T>::ld(x + i);
        s2 += g;
        Elem<T>::st(dx + i, g * cev);
    }
    dce[b * C + c] = s1;
    dte[b * C + c] = s2;
}

// ---- concat + bilinear x2 (align_corners=True) ---------------------------------------------------
// PyTorch upsample_bilinear2d (align_corners): scale = (in-1)/(out-1); src = scale*dst; i0 = (int)src;
// i1 = i0 + (i0 < in-1); l1 = src - i0; l0 = 1 - l1.
__device__ inline void bil_coef(int o, float scale, int in, int& i0, int& i1, float& l0, float& l1) {
    const float src = scale * (float)o;
    i0 = (int)src;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = src - (float)i0;
    l0 = 1.f - l1;
}

template <typename T, int V>
__global__ void upcat_fwd_kernel(const T* x1, const T* x2, T* y, int B, int H, int W, int C1, int C2, int B2) {
    const int C = C1 + C2, CV = C / V, Ho = 2 * H, Wo = 2 * W;
    const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const int64_t total = (int64_t)B * Ho * Wo * CV;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        int64_t r = i / CV;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        int y0, y1, x0, x1i; float h0, h1, w0, w1;
        bil_coef(oy, sy, H, y0, y1, h0, h1);
        bil_coef(ox, sx, W, x0, x1i, w0, w1);
        const int c = cv * V;
        const T* src; int Cs, cc;
        int bs = b;
        if (c < C1) { src = x1; Cs = C1; cc = c; } else { src = x2; Cs = C2; cc = c - C1; bs = b % B2; }   // x2 may hold B2 < B samples (CFG)
        const T* sb = src + (size_t)bs * H * W * Cs + cc;
        float a[V], bq[V], cq[V], d[V], o[V];
        ldv<T, V>(sb + ((size_t)y0 * W + x0) * Cs, a);
        ldv<T, V>(sb + ((size_t)y0 * W + x1i) * Cs, bq);
        ldv<T, V>(sb + ((size_t)y1 * W + x0) * Cs, cq);
        ldv<T, V>(sb + ((size_t)y1 * W + x1i) * Cs, d);
#pragma unroll
        for (int k = 0; k < V; ++k) o[k] = h0 * (w0 * a[k] + w1 * bq[k]) + h1 * (w0 * cq[k] + w1 * d[k]);
        stv<T, V>(y + i * V, o);
    }
}

struct AxisTaps { int idx[6]; float w[6]; int n; };
// all output positions o (0..out-1) whose interpolation touches input position i, with weights
__device__ inline void gather_taps(int i, float scale, int in, int out, AxisTaps& t) {
    t.n = 0;
    int lo, hi;
    if (scale <= 0.f) { lo = 0; hi = out - 1; }
    else {
        lo = (int)floorf((float)(i - 1) / scale) - 1;
        hi = (int)ceilf((float)(i + 1) / scale) + 1;
        lo = lo < 0 ? 0 : lo;
        hi = hi > out - 1 ? out - 1 : hi;
    }
    for (int o = lo; o <= hi && t.n < 6; ++o) {
        int i0, i1; float l0, l1;
        bil_coef(o, scale, in, i0, i1, l0, l1);
        float w = 0.f;
        if (i0 == i) w += l0;
        if (i1 == i) w += l1;
        if (i0 == i || i1 == i) { t.idx[t.n] = o; t.w[t.n] = w; ++t.n; }
    }
}

template <typename T, int V>
__global__ void upcat_bwd_kernel(const T* dy, T* dx1, T* dx2, int B, int H, int W, int C1, int C2) {
    const int C = C1 + C2, CV = C / V, Ho = 2 * H, Wo = 2 * W;
    const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const int64_t total = (int64_t)B * H * W * CV;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        int64_t r = i / CV;
        const int ix = (int)(r % W); r /= W;
        const int iy = (int)(r % H);
        const int b = (int)(r / H);
        AxisTaps ty, tx;
        gather_taps(iy, sy, H, Ho, ty);
        gather_taps(ix, sx, W, Wo, tx);
        float acc[V];
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] = 0.f;
        const int c = cv * V;
        for (int a = 0; a < ty.n; ++a)
            for (int q = 0; q < tx.n; ++q) {
                float g[V];
                ldv<T, V>(dy + (((size_t)b * Ho + ty.idx[a]) * Wo + tx.idx[q]) * C + c, g);
                const float w = ty.w[a] * tx.w[q];
#pragma unroll
                for (int k = 0; k < V; ++k) acc[k] += w * g[k];
            }
        if (c < C1) stv<T, V>(dx1 + (((size_t)b * H + iy) * W + ix) * C1 + c, acc);
        else stv<T, V>(dx2 + (((size_t)b * H + iy) * W + ix) * C2 + (c - C1), acc);
    }
}

// The same gather for H, W >= 4 with the taps in registers: the outputs whose interpolation touches input row i lie in an
// 8-wide window starting at floor((i - 1) / scale) - 1 (2 / scale <= 4.7 there), so each axis is 8 statically indexed
// (weight, index) pairs — the AxisTaps form above indexes its arrays dynamically, which puts them in scratch memory and ran
// the 32x32 -> 64x64 gradient at 180 us for 170 MB of traffic.
template <int NT>
__device__ __forceinline__ void window_taps(int i, float scale, int in, int out, int (&idx)[NT], float (&w)[NT]) {
    int lo = (int)floorf((float)(i - 1) / scale) - 1;
    lo = lo < 0 ? 0 : lo;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int o = lo + j;
        float ww = 0.f;
        if (o < out) {
            int i0, i1; float l0, l1;
            bil_coef(o, scale, in, i0, i1, l0, l1);
            if (i0 == i) ww += l0;
            if (i1 == i) ww += l1;
        }
        idx[j] = o < out ? o : out - 1;
        w[j] = ww;
    }
}

template <typename T, int V>
__global__ void upcat_bwd_win_kernel(const T* dy, T* dx1, T* dx2, int B, int H, int W, int C1, int C2) {
    const int C = C1 + C2, CV = C / V, Ho = 2 * H, Wo = 2 * W;
    const float sy = (float)(H - 1) / (float)(Ho - 1), sx = (float)(W - 1) / (float)(Wo - 1);
    const int64_t total = (int64_t)B * H * W * CV;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        int64_t r = i / CV;
        const int ix = (int)(r % W); r /= W;
        const int iy = (int)(r % H);
        const int b = (int)(r / H);
        int oy[8], ox[8];
        float wy[8], wx[8];
        window_taps<8>(iy, sy, H, Ho, oy, wy);
        window_taps<8>(ix, sx, W, Wo, ox, wx);
        float acc[V];
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] = 0.f;
        const int c = cv * V;
        const T* base = dy + (size_t)b * Ho * Wo * C + c;
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            if (wy[a] == 0.f) continue;
            const T* row = base + (size_t)oy[a] * Wo * C;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (wx[q] == 0.f) continue;
                float g[V];
                ldv<T, V>(row + (size_t)ox[q] * C, g);
                const float w = wy[a] * wx[q];
#pragma unroll
                for (int k = 0; k < V; ++k) acc[k] += w * g[k];
            }
        }
        if (c < C1) stv<T, V>(dx1 + (((size_t)b * H + iy) * W + ix) * C1 + c, acc);
        else stv<T, V>(dx2 + (((size_t)b * H + iy) * W + ix) * C2 + (c - C1), acc);
    }
}

template <typename T, int V>
__global__ void cat_kernel(const T* x1, const T* x2, T* y, int64_t M, int C1, int C2, int bwd) {
    // fwd: y[m] = [x1[m] | x2[m]];  bwd: x1[m], x2[m] <- y[m]  (x pointers are then outputs)
    const int C = C1 + C2, CV = C / V;
    const int64_t total = M * CV;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % CV) * V;
        const int64_t m = i / CV;
        T* p = (c < C1) ? (T*)x1 + m * C1 + c : (T*)x2 + m * C2 + (c - C1);
        float f[V];
        if (!bwd) { ldv<T, V>(p, f); stv<T, V>(y + i * V, f); }
        else { ldv<T, V>(y + i * V, f); stv<T, V>(p, f); }
    }
}

// ---- AvgPool k (stride k) + GELU -> fp32 -------------------------------------------------------
template <typename T>
__global__ void avgpool_gelu_fwd_kernel(const T* x, float* y, int B, int H, int W, int C, int k) {
    const int Ho = H / k, Wo = W / k;
    const int64_t total = (int64_t)B * Ho * Wo * C;
    const float inv = 1.f / (float)(k * k);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        int64_t r = i / C;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        float s = 0.f;
        for (int dy = 0; dy < k; ++dy)
            for (int dx = 0; dx < k; ++dx) s += Elem<T>::ld(x + (((size_t)b * H + oy * k + dy) * W + ox * k + dx) * C + c);
        y[i] = gelu_f(s * inv);
    }
}
template <typename T>
__global__ void avgpool_gelu_bwd_kernel(const T* x, const float* dy, T* dx, int B, int H, int W, int C, int k) {
    const int Ho = H / k, Wo = W / k;
    const int64_t total = (int64_t)B * Ho * Wo * C;
    const float inv = 1.f / (float)(k * k);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        int64_t r = i / C;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        float s = 0.f;
        for (int dyy = 0; dyy < k; ++dyy)
            for (int dxx = 0; dxx < k; ++dxx) s += Elem<T>::ld(x + (((size_t)b * H + oy * k + dyy) * W + ox * k + dxx) * C + c);
        const float g = dy[i] * gelu_grad_f(s * inv) * inv;
        for (int dyy = 0; dyy < k; ++dyy)
            for (int dxx = 0; dxx < k; ++dxx) Elem<T>::st(dx + (((size_t)b * H + oy * k + dyy) * W + ox * k + dxx) * C + c, g);
    }
}

// ---- MaxPool2d(2) --------------------------------------------------------------------------------
template <typename T>
__global__ void maxpool2_kernel(const T* x, const T* dy, T* out, int B, int H, int W, int C, int bwd) {
    const int Ho = H / 2, Wo = W / 2;
    const int64_t total = (int64_t)B * Ho * Wo * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        int64_t r = i / C;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        float best = -INFINITY; int arg = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float v = Elem<T>::ld(x + (((size_t)b * H + oy * 2 + (q >> 1)) * W + ox * 2 + (q & 1)) * C + c);
            if (v > best || v != v) { best = v; arg = q; }
        }
        if (!bwd) Elem<T>::st(out + i, best);
        else {
            const float g = Elem<T>::ld(dy + i);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                Elem<T>::st(out + (((size_t)b * H + oy * 2 + (q >> 1)) * W + ox * 2 + (q & 1)) * C + c, q == arg ? g : 0.f);
        }
    }
}

template <typename T, int V>
__global__ void add_kernel(const T* a, const T* b, T* y, int64_t nvec) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        float fa[V], fb[V];
        ldv<T, V>(a + i * V, fa);
        ldv<T, V>(b + i * V, fb);
#pragma unroll
        for (int k = 0; k < V; ++k) fa[k] += fb[k];
        stv<T, V>(y + i * V, fa);
    }
}

template <typename T>
bool al16(int c1, int c2, const void* a, const void* b = nullptr, const void* c = nullptr) {
    const uintptr_t m = (uintptr_t)a | (uintptr_t)b | (uintptr_t)c;
    return (c1 % Elem<T>::VE) == 0 && (c2 % Elem<T>::VE) == 0 && (m & 15) == 0;
}

}  // namespace

#define ST ((hipStream_t)s)

extern "C" int dm_pack_w(const float* src, void* dst, int dtype, int N, int T_, int C, int Cp, dm_stream_t s) {
    DM_CHECK_ARG(src && dst && N > 0 && T_ > 0 && C > 0 && Cp >= C, "dm_pack_w: bad arguments");
    const int64_t rows = (int64_t)N * T_;
    DM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((pack_w_kernel<T>), dim3(grid_for(rows * Cp, 256)), dim3(256), 0, ST, src, (T*)dst, rows, C, Cp));
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_pack_wT(const float* src, void* dst, int dtype, int N, int T_, int C, int Tt, const int32_t* taps, int Np, dm_stream_t s) {
    DM_CHECK_ARG(src && dst && N > 0 && T_ > 0 && C > 0 && Tt > 0 && Tt <= 64 && Np >= N, "dm_pack_wT: bad arguments");
    TapList tl;
    for (int i = 0; i < Tt; ++i) {
        tl.t[i] = taps ? taps[i] : i;
        DM_CHECK_ARG(tl.t[i] >= 0 && tl.t[i] < T_, "dm_pack_wT: tap %d out of range", tl.t[i]);
    }
    dim3 grid(cdiv(Np, 32), cdiv(C, 32), Tt);
    DM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((pack_wT_kernel<T>), grid, dim3(256), 0, ST, src, (T*)dst, N, T_, C, Tt, Np, tl));
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_pack_multi(const int64_t* entries, const int32_t* taps, const int32_t* blocks, int n_blocks, dm_stream_t s) {
    DM_CHECK_ARG(entries && taps && blocks && n_blocks > 0, "dm_pack_multi: bad arguments");
    hipLaunchKernelGGL(pack_multi_kernel, dim3(n_blocks < 16384 ? n_blocks : 16384), dim3(256), 0, ST, entries, taps, blocks, n_blocks);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_unpad_dw(const float* src, float* dst, int N, int T_, int C, int Cp, int accumulate, dm_stream_t s) {
    DM_CHECK_ARG(src && dst && N > 0 && T_ > 0 && C > 0 && Cp >= C, "dm_unpad_dw: bad arguments");
    const int64_t rows = (int64_t)N * T_;
    hipLaunchKernelGGL(unpad_dw_kernel, dim3(grid_for(rows * C, 256)), dim3(256), 0, ST, src, dst, rows, C, Cp, accumulate);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_cast(const void* x, void* y, int from_dtype, int to_dtype, int64_t n, dm_stream_t s) {
    DM_CHECK_ARG(x && y && n > 0, "dm_cast: bad arguments");
    const dim3 g(grid_for(n, 256));
#define DM_CAST_CASE(FD, FT, TD, TT) \
    if (from_dtype == FD && to_dtype == TD) { hipLaunchKernelGGL((cast_kernel<FT, TT>), g, dim3(256), 0, ST, (const FT*)x, (TT*)y, n); } else
    DM_CAST_CASE(DM_F32, float, DM_BF16, bf16)
    DM_CAST_CASE(DM_BF16, bf16, DM_F32, float)
    DM_CAST_CASE(DM_F32, float, DM_F32, float)
    DM_CAST_CASE(DM_BF16, bf16, DM_BF16, bf16)
    DM_CAST_CASE(DM_F32, float, DM_F16, f16)
    DM_CAST_CASE(DM_F16, f16, DM_F32, float)
    DM_CAST_CASE(DM_F16, f16, DM_F16, f16)
#undef DM_CAST_CASE
    { dm_set_error("dm_cast: bad dtypes %d -> %d", from_dtype, to_dtype); return DM_EINVAL; }
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_nchw_to_nhwc(const float* x, void* y, int dtype, int B, int C, int H, int W, int Cp, int repeat, dm_stream_t s) {
    DM_CHECK_ARG(x && y && B > 0 && C > 0 && H > 0 && W > 0 && Cp >= C && repeat >= 1, "dm_nchw_to_nhwc: bad arguments");
    const int64_t total = (int64_t)B * repeat * H * W * Cp;
    DM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((nchw_to_nhwc_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, ST, x, (T*)y, B, C, H * W, Cp, repeat));
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_nhwc_to_nchw(const void* x, float* y, int dtype, int B, int C, int H, int W, int Cp, dm_stream_t s) {
    DM_CHECK_ARG(x && y && B > 0 && C > 0 && H > 0 && W > 0 && Cp >= C, "dm_nhwc_to_nchw: bad arguments");
    const int64_t total = (int64_t)B * C * H * W;
    DM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((nhwc_to_nchw_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, ST, (const T*)x, y, B, C, H * W, Cp));
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_film_fwd(const void* x, const float* cemb, const float* temb, void* y, int dtype, int B, int HW, int C, dm_stream_t s) {
    DM_CHECK_ARG(x && cemb && temb && y && B > 0 && HW > 0 && C > 0, "dm_film_fwd: bad arguments");
    DM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((film_fwd_kernel<T>), dim3(grid_for((int64_t)B * HW * C, 256)), dim3(256), 0, ST, (const T*)x, cemb, temb, (T*)y, B, HW, C));
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_film_bwd(const void* x, const void* dy, const float* cemb, void* dx, float* dcemb, float* dtemb, int dtype, int B,
                           int HW, int C, dm_stream_t s) {
    DM_CHECK_ARG(x && dy && cemb && dx && dcemb && dtemb && B > 0 && HW > 0 && C > 0, "dm_film_bwd: bad arguments");
    DM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((film_bwd_kernel<T>), dim3(B, cdiv(C, 256)), dim3(256), 0, ST, (const T*)x, (const T*)dy, cemb, (T*)dx, dcemb, dtemb, HW, C));
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_upcat_fwd_bcast(const void* x1, const void* x2, void* y, int dtype, int B, int B2, int H, int W, int C1, int C2, dm_stream_t s) {
    DM_CHECK_ARG(x1 && y && (C2 == 0 || x2) && B > 0 && H > 0 && W > 0 && C1 > 0 && C2 >= 0, "dm_upcat_fwd: bad arguments");
    DM_CHECK_ARG(B2 > 0 && B % B2 == 0, "dm_upcat_fwd_bcast: the batch of x2 (%d) must divide B (%d)", B2, B);
    const int C = C1 + C2;
    DM_DISPATCH_DTYPE(dtype, {
        if (al16<T>(C1, C2, x1, x2, y)) hipLaunchKernelGGL((upcat_fwd_kernel<T, Elem<T>::VE>), dim3(grid_for((int64_t)B * 4 * H * W * C / Elem<T>::VE, 256)), dim3(256), 0, ST, (const T*)x1, (const T*)x2, (T*)y, B, H, W, C1, C2, B2);
        else hipLaunchKernelGGL((upcat_fwd_kernel<T, 1>), dim3(grid_for((int64_t)B * 4 * H * W * C, 256)), dim3(256), 0, ST, (const T*)x1, (const T*)x2, (T*)y, B, H, W, C1, C2, B2);
    });
    DM_LAUNCH_CHECK();
    return DM_OK;
}
extern "C" int dm_upcat_fwd(const void* x1, const void* x2, void* y, int dtype, int B, int H, int W, int C1, int C2, dm_stream_t s) {
    return dm_upcat_fwd_bcast(x1, x2, y, dtype, B, B, H, W, C1, C2, s);
}

extern "C" int dm_upcat_bwd(const void* dy, void* dx1, void* dx2, int dtype, int B, int H, int W, int C1, int C2, dm_stream_t s) {
    DM_CHECK_ARG(dy && dx1 && (C2 == 0 || dx2) && B > 0 && H > 0 && W > 0 && C1 > 0 && C2 >= 0, "dm_upcat_bwd: bad arguments");
    const int C = C1 + C2;
    DM_DISPATCH_DTYPE(dtype, {
        if (al16<T>(C1, C2, dy, dx1, dx2) && H >= 4 && W >= 4) hipLaunchKernelGGL((upcat_bwd_win_kernel<T, Elem<T>::VE>), dim3(grid_for((int64_t)B * H * W * C / Elem<T>::VE, 256, 1 << 16)), dim3(256), 0, ST, (const T*)dy, (T*)dx1, (T*)dx2, B, H, W, C1, C2);
        else if (al16<T>(C1, C2, dy, dx1, dx2)) hipLaunchKernelGGL((upcat_bwd_kernel<T, Elem<T>::VE>), dim3(grid_for((int64_t)B * H * W * C / Elem<T>::VE, 256)), dim3(256), 0, ST, (const T*)dy, (T*)dx1, (T*)dx2, B, H, W, C1, C2);
        else hipLaunchKernelGGL((upcat_bwd_kernel<T, 1>), dim3(grid_for((int64_t)B * H * W * C, 256)), dim3(256), 0, ST, (const T*)dy, (T*)dx1, (T*)dx2, B, H, W, C1, C2);
    });
    DM_LAUNCH_CHECK();
    return DM_OK;
}

static int cat_impl(const void* x1, const void* x2, void* y, int dtype, int M, int C1, int C2, int bwd, dm_stream_t s) {
    DM_CHECK_ARG(x1 && x2 && y && M > 0 && C1 > 0 && C2 > 0, "dm_cat: bad arguments");
    const int C = C1 + C2;
    DM_DISPATCH_DTYPE(dtype, {
        if (al16<T>(C1, C2, x1, x2, y)) hipLaunchKernelGGL((cat_kernel<T, Elem<T>::VE>), dim3(grid_for((int64_t)M * C / Elem<T>::VE, 256)), dim3(256), 0, ST, (const T*)x1, (const T*)x2, (T*)y, (int64_t)M, C1, C2, bwd);
        else hipLaunchKernelGGL((cat_kernel<T, 1>), dim3(grid_for((int64_t)M * C, 256)), dim3(256), 0, ST, (const T*)x1, (const T*)x2, (T*)y, (int64_t)M, C1, C2, bwd);
    });
    DM_LAUNCH_CHECK();
    return DM_OK;
}
extern "C" int dm_cat_fwd(const void* x1, const void* x2, void* y, int dtype, int M, int C1, int C2, dm_stream_t s) { return cat_impl(x1, x2, y, dtype, M, C1, C2, 0, s); }
extern "C" int dm_cat_bwd(const void* dy, void* dx1, void* dx2, int dtype, int M, int C1, int C2, dm_stream_t s) { return cat_impl(dx1, dx2, (void*)dy, dtype, M, C1, C2, 1, s); }

extern "C" int dm_avgpool_gelu_fwd(const void* x, float* y, int dtype, int B, int H, int W, int C, int k, dm_stream_t s) {
    DM_CHECK_ARG(x && y && B > 0 && k > 0 && H >= k && W >= k && C > 0, "dm_avgpool_gelu_fwd: bad arguments (H=%d W=%d k=%d)", H, W, k);
    DM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((avgpool_gelu_fwd_kernel<T>), dim3(grid_for((int64_t)B * (H / k) * (W / k) * C, 256)), dim3(256), 0, ST, (const T*)x, y, B, H, W, C, k));
    DM_LAUNCH_CHECK();
    return DM_OK;
}
extern "C" int dm_avgpool_gelu_bwd(const void* x, const float* dy, void* dx, int dtype, int B, int H, int W, int C, int k, dm_stream_t s) {
    DM_CHECK_ARG(x && dy && dx && B > 0 && k > 0 && H >= k && W >= k && C > 0 && H % k == 0 && W % k == 0, "dm_avgpool_gelu_bwd: bad arguments");
    DM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((avgpool_gelu_bwd_kernel<T>), dim3(grid_for((int64_t)B * (H / k) * (W / k) * C, 256)), dim3(256), 0, ST, (const T*)x, dy, (T*)dx, B, H, W, C, k));
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_maxpool2_fwd(const void* x, void* y, int dtype, int B, int H, int W, int C, dm_stream_t s) {
    DM_CHECK_ARG(x && y && B > 0 && H >= 2 && W >= 2 && C > 0, "dm_maxpool2_fwd: bad arguments");
    DM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((maxpool2_kernel<T>), dim3(grid_for((int64_t)B * (H / 2) * (W / 2) * C, 256)), dim3(256), 0, ST, (const T*)x, (const T*)nullptr, (T*)y, B, H, W, C, 0));
    DM_LAUNCH_CHECK();
    return DM_OK;
}
extern "C" int dm_maxpool2_bwd(const void* x, const void* dy, void* dx, int dtype, int B, int H, int W, int C, dm_stream_t s) {
    DM_CHECK_ARG(x && dy && dx && B > 0 && H >= 2 && W >= 2 && C > 0 && H % 2 == 0 && W % 2 == 0, "dm_maxpool2_bwd: bad arguments (odd sizes unsupported)");
    DM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((maxpool2_kernel<T>), dim3(grid_for((int64_t)B * (H / 2) * (W / 2) * C, 256)), dim3(256), 0, ST, (const T*)x, (const T*)dy, (T*)dx, B, H, W, C, 1));
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_add(const void* a, const void* b, void* y, int dtype, int64_t n, dm_stream_t s) {
    DM_CHECK_ARG(a && b && y && n > 0, "dm_add: bad arguments");
    DM_DISPATCH_DTYPE(dtype, {
        constexpr int V = Elem<T>::VE;
        const uintptr_t m = (uintptr_t)a | (uintptr_t)b | (uintptr_t)y;
        if (n % V == 0 && (m & 15) == 0) hipLaunchKernelGGL((add_kernel<T, V>), dim3(grid_for(n / V, 256)), dim3(256), 0, ST, (const T*)a, (const T*)b, (T*)y, n / V);
        else hipLaunchKernelGGL((add_kernel<T, 1>), dim3(grid_for(n, 256)), dim3(256), 0, ST, (const T*)a, (const T*)b, (T*)y, n);
    });
    DM_LAUNCH_CHECK();
    return DM_OK;
}

// out = (x ? x : 0) + y * [mask[b,pix] > thresh]   (LocalEnhancer, new_scripy.py:172-174)
namespace {
template <typename T>
__global__ void mask_axpy_kernel(const T* x, const T* y, const float* mask, float thresh, T* out, int64_t npix, int C) {
    const int64_t total = npix * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const float h = mask[i / C] > thresh ? 1.f : 0.f;
        Elem<T>::st(out + i, (x ? Elem<T>::ld(x + i) : 0.f) + Elem<T>::ld(y + i) * h);
    }
}
}  // namespace
extern "C" int dm_mask_axpy(const void* x, const void* y, const float* mask, float thresh, void* out, int dtype, int64_t npix, int C, dm_stream_t s) {
    DM_CHECK_ARG(y && mask && out && npix > 0 && C > 0, "dm_mask_axpy: bad arguments");
    DM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((mask_axpy_kernel<T>), dim3(grid_for(npix * C, 256)), dim3(256), 0, ST, (const T*)x, (const T*)y, mask, thresh, (T*)out, npix, C));
    DM_LAUNCH_CHECK();
    return DM_OK;
}
