// Channel-attention pieces: SE squeeze/scale + residual combine, coordinate-attention strip pooling and
// gated multiply, and the small fp32 dense layers (EmbedFC, SE fc, CoordAttn 1x1 convs on strips).
// Activation tensors are NHWC `dtype`; strips, gates and embeddings are fp32.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// strip mean / product-sum: out[b,a,c] = scale * sum_r f(x[b, a*sa + r*sr, c])   (pixel-unit strides)
// grid: B*A blocks; thread (cv, rl) with V channels; LDS folds the row lanes.
// MODE 0: sum x ; MODE 1: sum x*y (second tensor same layout)
// ---------------------------------------------------------------------------------------------
template <typename T, int V, int MODE>
__global__ __launch_bounds__(256) void strip_reduce_kernel(const T* x, const T* y, float* out, int A, int R, int sa, int sr,
                                                           int P /*pixels per sample*/, int C, float scale, int RS, int rps) {
    // RS > 1: the R positions are split over RS workgroups (rps each); workgroup (strip, rs) writes its partial sums to
    // out[rs][strip][c] (a scratch array, scale 1) and strip_fold_kernel adds them up — a whole-image pool (A = 1, R = H*W) on
    // B workgroups left 3/4 of the chip idle and ran a 256-deep serial load chain per thread (142 us for 67 MB)
    __shared__ float red[2048];
    const int strip = blockIdx.x / RS, rs = blockIdx.x - strip * RS;
    const int b = strip / A, a = strip - b * A;
    const int r_lo = rs * rps, r_hi = min(R, r_lo + rps);
    const int CVt = (C + V - 1) / V;
    const size_t base = ((size_t)b * P + (size_t)a * sa) * C;
    out += (size_t)rs * (gridDim.x / RS) * C;
    for (int cbase = 0; cbase < CVt; cbase += 256) {
        const int ncv = min(256, CVt - cbase);
        const int RL = 256 / ncv;
        const int cv = threadIdx.x % ncv, rl = threadIdx.x / ncv;
        float acc[V];
#pragma unroll
        for (int i = 0; i < V; ++i) acc[i] = 0.f;
        if (rl < RL) {
            const int c0 = (cbase + cv) * V;
            for (int r = r_lo + rl; r < r_hi; r += RL) {
                float v[V];
                const size_t off = base + (size_t)r * sr * C + c0;
                if constexpr (V == 1) v[0] = Elem<T>::ld(x + off); else load_vec<T>(x + off, v);
                if constexpr (MODE == 1) {
                    float w[V];
                    if constexpr (V == 1) w[0] = Elem<T>::ld(y + off); else load_vec<T>(y + off, w);
#pragma unroll
                    for (int i = 0; i < V; ++i) v[i] *= w[i];
                }
#pragma unroll
                for (int i = 0; i < V; ++i) acc[i] += v[i];
            }
        }
        __syncthreads();
        if (rl < RL) {
#pragma unroll
            for (int i = 0; i < V; ++i) red[(rl * ncv + cv) * V + i] = acc[i];
        }
        __syncthreads();
        for (int j = threadIdx.x; j < ncv * V; j += 256) {
            float s = 0.f;
            for (int l = 0; l < RL; ++l) s += red[l * ncv * V + j];
            const int col = cbase * V + j;
            if (col < C) out[((size_t)b * A + a) * C + col] = s * scale;
        }
        __syncthreads();
    }
}

// out[j] = scale * sum_rs part[rs][j]
__global__ void strip_fold_kernel(const float* part, float* out, int n, int RS, float scale) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    float s = 0.f;
    for (int k = 0; k < RS; ++k) s += part[(size_t)k * n + j];
    out[j] = s * scale;
}

template <int MODE>
int launch_strip(const void* x, const void* y, float* out, int dtype, int B, int A, int R, int sa, int sr, int P, int C, float scale,
                 hipStream_t st, const float** parts_out = nullptr, int* rs_out = nullptr) {
    // few long strips (the whole-image pools of the SE block): split the positions until ~1024 workgroups share the read
    // parts_out != nullptr: no fold launch — the consumer (chain.hip) adds the RS partial arrays [RS][B * A][C] itself (scale 1);
    // they always live in the workspace then, also when RS == 1
    const bool keep = parts_out != nullptr;
    int RS = 1, rps = R;
    if (B * A < 512 && R >= 256 && dm_g_ws != nullptr) {
        RS = 1024 / (B * A);
        if (RS > R / 64) RS = R / 64;
        if (RS < 1 || (int64_t)RS * B * A * C * (int64_t)sizeof(float) > dm_g_ws_bytes) RS = 1;
        rps = cdiv(R, RS);
        RS = cdiv(R, rps);
    }
    if (keep && (dm_g_ws == nullptr || (int64_t)RS * B * A * C * (int64_t)sizeof(float) > dm_g_ws_bytes)) {
        dm_set_error("strip sums: the workspace (dm_set_workspace) is missing or too small for %d x %d x %d partial sums", RS, B * A, C);
        return DM_EINVAL;
    }
    float* dst = (RS > 1 || keep) ? dm_g_ws : out;
    const float sc = (RS > 1 || keep) ? 1.f : scale;
    DM_DISPATCH_DTYPE(dtype, {
        const uintptr_t m = (uintptr_t)x | (uintptr_t)y;
        if (C % Elem<T>::VE == 0 && (m & 15) == 0)
            hipLaunchKernelGGL((strip_reduce_kernel<T, Elem<T>::VE, MODE>), dim3(B * A * RS), dim3(256), 0, st, (const T*)x, (const T*)y, dst, A, R, sa, sr, P, C, sc, RS, rps);
        else
            hipLaunchKernelGGL((strip_reduce_kernel<T, 1, MODE>), dim3(B * A * RS), dim3(256), 0, st, (const T*)x, (const T*)y, dst, A, R, sa, sr, P, C, sc, RS, rps);
    });
    if (keep) {
        *parts_out = dm_g_ws;
        *rs_out = RS;
    } else if (RS > 1) {
        hipLaunchKernelGGL(strip_fold_kernel, dim3(cdiv(B * A * C, 256)), dim3(256), 0, st, dm_g_ws, out, B * A * C, RS, scale);
    }
    DM_LAUNCH_CHECK();
    return DM_OK;
}

// out = (res + x2 * s[b,c]) * inv
template <typename T>
__global__ void scale_res_fwd_kernel(const T* x2, const T* res, const float* sg, T* out, int B, int HW, int C, float inv) {
    const int64_t total = (int64_t)B * HW * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int b = (int)(i / ((int64_t)HW * C));
        const float s = sg ? sg[b * C + c] : 1.f;
        Elem<T>::st(out + i, (Elem<T>::ld(res + i) + Elem<T>::ld(x2 + i) * s) * inv);
    }
}
// dx2 = dout*inv*s + dym[b,c]/HW ; dres = dout*inv
template <typename T>
__global__ void scale_res_bwd_kernel(const T* dout, const float* sg, const float* dym, T* dx2, T* dres, int B, int HW, int C, float inv) {
    const int64_t total = (int64_t)B * HW * C;
    const float ihw = 1.f / (float)HW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int b = (int)(i / ((int64_t)HW * C));
        const float g = Elem<T>::ld(dout + i) * inv;
        const float s = sg ? sg[b * C + c] : 1.f;
        Elem<T>::st(dx2 + i, g * s + (dym ? dym[b * C + c] * ihw : 0.f));
        Elem<T>::st(dres + i, g);
    }
}

// 16-byte-vector forms (C % VE == 0, aligned): grid.y = sample, a thread walks over the sample's pixels with ONE channel
// vector, whose gate / mean-gradient values it loads once
template <typename T>
__global__ void scale_res_fwd_vec_kernel(const T* x2, const T* res, const float* sg, T* out, int HW, int C, float inv) {
    constexpr int V = Elem<T>::VE;
    const int CV = C / V, b = blockIdx.y;
    const int n = HW * CV;                                  // vectors per sample
    const int step = (gridDim.x * 256 / CV) * CV;           // a multiple of CV: the channel vector of a thread never changes
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= step) return;
    const int c = (i % CV) * V;
    float s[V];
#pragma unroll
    for (int k = 0; k < V; ++k) s[k] = sg ? sg[b * C + c + k] : 1.f;
    const size_t base = (size_t)b * n * V;
    for (; i < n; i += step) {
        float a[V], r[V];
        load_vec<T>(x2 + base + (size_t)i * V, a);
        load_vec<T>(res + base + (size_t)i * V, r);
#pragma unroll
        for (int k = 0; k < V; ++k) a[k] = (r[k] + a[k] * s[k]) * inv;
        store_vec<T>(out + base + (size_t)i * V, a);
    }
}
template <typename T>
__global__ void scale_res_bwd_vec_kernel(const T* dout, const float* sg, const float* dym, T* dx2, T* dres, int HW, int C, float inv) {
    constexpr int V = Elem<T>::VE;
    const int CV = C / V, b = blockIdx.y;
    const int n = HW * CV;
    const int step = (gridDim.x * 256 / CV) * CV;
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= step) return;
    const int c = (i % CV) * V;
    const float ihw = 1.f / (float)HW;
    float s[V], m[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { s[k] = sg ? sg[b * C + c + k] : 1.f; m[k] = dym ? dym[b * C + c + k] * ihw : 0.f; }
    const size_t base = (size_t)b * n * V;
    for (; i < n; i += step) {
        float g[V], d[V];
        load_vec<T>(dout + base + (size_t)i * V, g);
#pragma unroll
        for (int k = 0; k < V; ++k) { g[k] *= inv; d[k] = g[k] * s[k] + m[k]; }
        store_vec<T>(dx2 + base + (size_t)i * V, d);
        store_vec<T>(dres + base + (size_t)i * V, g);
    }
}

// ---- coordinate attention gate ------------------------------------------------------------------
__device__ inline void ca_mix(const float* alpha, const float* beta, float& al, float& be) {
    const float sa = sigmoid_f(alpha[0]), sb = sigmoid_f(beta[0]);
    const float S = sa + sb + 1e-8f;
    al = sa / S;
    be = sb / S;
}

template <typename T>
__global__ void ca_gate_fwd_kernel(const T* x, const float* lh, const float* lw, const float* alpha, const float* beta, T* out,
                                   int B, int H, int W, int C) {
    float al, be;
    ca_mix(alpha, beta, al, be);
    const int64_t total = (int64_t)B * H * W * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        int64_t r = i / C;
        const int xx = (int)(r % W); r /= W;
        const int yy = (int)(r % H);
        const int b = (int)(r / H);
        const float att = al * sigmoid_f(lh[((size_t)b * H + yy) * C + c]) + be * sigmoid_f(lw[((size_t)b * W + xx) * C + c]);
        Elem<T>::st(out + i, Elem<T>::ld(x + i) * att);
    }
}

// dx_gate = dout * att
template <typename T>
__global__ void ca_gate_bwd_dx_kernel(const T* dout, const float* lh, const float* lw, const float* alpha, const float* beta,
                                      T* dx, int B, int H, int W, int C) {
    float al, be;
    ca_mix(alpha, beta, al, be);
    const int64_t total = (int64_t)B * H * W * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        int64_t r = i / C;
        const int xx = (int)(r % W); r /= W;
        const int yy = (int)(r % H);
        const int b = (int)(r / H);
        const float att = al * sigmoid_f(lh[((size_t)b * H + yy) * C + c]) + be * sigmoid_f(lw[((size_t)b * W + xx) * C + c]);
        Elem<T>::st(dx + i, Elem<T>::ld(dout + i) * att);
    }
}

// 16-byte-vector form of the three per-pixel CoordAttn kernels (C % VE == 0, aligned): grid.y = sample, a thread keeps one channel
// vector and walks over the pixels (32-bit index math; the scalar forms above divide 64-bit indices per element: 18 us for 33 MB).
//   GATE: out = in * (al sigmoid(lh[b,y,c]) + be sigmoid(lw[b,x,c]))        (forward, and dx of the gate with in = dout)
//   POOL: out = in + sh[b,y,c] / W + sw[b,x,c] / H                          (backward of the strip means; in may be NULL)
template <typename T, bool GATE>
__global__ void ca_pix_vec_kernel(const T* in, const float* sh, const float* sw, const float* alpha, const float* beta, T* out, int H, int W,
                                  int C) {
    constexpr int V = Elem<T>::VE;
    const int CV = C / V, b = blockIdx.y, n = H * W * CV;
    const int step = (gridDim.x * 256 / CV) * CV;
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= step) return;
    const int c = (i % CV) * V;
    float al = 0.f, be = 0.f;
    if constexpr (GATE) ca_mix(alpha, beta, al, be);
    else { al = 1.f / (float)W; be = 1.f / (float)H; }
    const size_t base = (size_t)b * n * V;
    for (; i < n; i += step) {
        const int pix = i / CV, yy = pix / W, xx = pix - yy * W;
        const float* ph = sh + ((size_t)b * H + yy) * C + c;
        const float* pw = sw + ((size_t)b * W + xx) * C + c;
        float v[V];
        if (in) load_vec<T>(in + base + (size_t)i * V, v);
        else {
#pragma unroll
            for (int k = 0; k < V; ++k) v[k] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < V; ++k) {
            if constexpr (GATE) v[k] *= al * sigmoid_f(ph[k]) + be * sigmoid_f(pw[k]);
            else v[k] = v[k] + ph[k] * al + pw[k] * be;
        }
        store_vec<T>(out + base + (size_t)i * V, v);
    }
}
template <typename T, bool GATE>
static bool launch_ca_pix(const T* in, const float* sh, const float* sw, const float* alpha, const float* beta, T* out, int B, int H, int W, int C,
                          hipStream_t st) {
    constexpr int V = Elem<T>::VE;
    if (C % V != 0 || C / V > 256 || ((((uintptr_t)in | (uintptr_t)out)) & 15) != 0 || (int64_t)H * W * (C / V) >= (1ll << 31)) return false;
    const int CV = C / V;
    int64_t g = ((int64_t)H * W * CV + 256 * 4 - 1) / (256 * 4);
    const int64_t need = (CV + 255) / 256, cap = 8192 / B + 1;
    if (g > cap) g = cap;
    if (g < need) g = need;
    hipLaunchKernelGGL((ca_pix_vec_kernel<T, GATE>), dim3((unsigned)g, B), dim3(256), 0, st, in, sh, sw, alpha, beta, out, H, W, C);
    return true;
}

// strip sums S[b,a,c] = sum_r dout*x  ->  dl = mix * sig'(l) * S ;  dmix += sum sig(l) * S
__global__ void ca_gate_bwd_strip_kernel(const float* S, const float* l, const float* alpha, const float* beta, int which,
                                         float* dl, float* dmix, int64_t n) {
    __shared__ float red[16];
    float al, be;
    ca_mix(alpha, beta, al, be);
    const float mix = which == 0 ? al : be;
    float part = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float sg = sigmoid_f(l[i]);
        const float sv = S[i];            // S and dl alias (in-place): read before the store
        dl[i] = mix * sg * (1.f - sg) * sv;
        part += sg * sv;
    }
    part = block_sum(part, red);
    if (threadIdx.x == 0) atomicAdd(dmix + which, part);
}

// dmix = {d_al, d_be} -> out[0] = dalpha, out[1] = dbeta  (out += )
__global__ void ca_mix_bwd_kernel(const float* alpha, const float* beta, const float* dmix, float* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float sa = sigmoid_f(alpha[0]), sb = sigmoid_f(beta[0]);
    const float S = sa + sb + 1e-8f;
    const float dsa = (dmix[0] * (S - sa) - dmix[1] * sb) / (S * S);
    const float dsb = (dmix[1] * (S - sb) - dmix[0] * sa) / (S * S);
    out[0] += dsa * sa * (1.f - sa);
    out[1] += dsb * sb * (1.f - sb);
}

// dx = dgate + dxh[b,y,c]/W + dxw[b,x,c]/H
template <typename T>
__global__ void ca_pool_bwd_kernel(const float* dxh, const float* dxw, const T* dg, T* dx, int B, int H, int W, int C) {
    const int64_t total = (int64_t)B * H * W * C;
    const float iw = 1.f / (float)W, ih = 1.f / (float)H;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        int64_t r = i / C;
        const int xx = (int)(r % W); r /= W;
        const int yy = (int)(r % H);
        const int b = (int)(r / H);
        Elem<T>::st(dx + i, (dg ? Elem<T>::ld(dg + i) : 0.f) + dxh[((size_t)b * H + yy) * C + c] * iw + dxw[((size_t)b * W + xx) * C + c] * ih);
    }
}

__global__ void sigmix_fwd_kernel(const float* x, const float* y, const float* gamma, float* xo, int n) {
    const float sg = sigmoid_f(gamma[0]);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) xo[i] = x[i] + sg * y[i];
}
__global__ void sigmix_bwd_kernel(const float* dxo, const float* y, const float* gamma, float* dy, float* dgamma, int n) {
    __shared__ float red[16];
    const float sg = sigmoid_f(gamma[0]);
    float part = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        dy[i] = sg * dxo[i];
        part += dxo[i] * y[i];
    }
    part = block_sum(part, red);
    if (threadIdx.x == 0) atomicAdd(dgamma, part * sg * (1.f - sg));
}

// ---------------------------------------------------------------------------------------------
// Small fp32 GEMM with arbitrary strides: Cm[m][n] (+)= act(sum_k A(m,k) * Bm(k,n) + bias[n])
// 64x64 tile, BK 16, 256 threads, 4x4 outputs per thread.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sgemm_kernel(const float* A, const float* Bm, const float* bias, float* Cm, int M, int N, int K,
                                                    int64_t sam, int64_t sak, int64_t sbk, int64_t sbn, int ldc, int act, int accumulate) {
    __shared__ float sA[16][65], sB[16][65];
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = threadIdx.x + 256 * q;  // 0..1023 over 64 x 16
            // choose the faster-varying index to follow the unit stride
            int mi, ki;
            if (sak == 1) { ki = idx & 15; mi = idx >> 4; } else { mi = idx & 63; ki = idx >> 6; }
            const int m = m0 + mi, k = k0 + ki;
            sA[ki][mi] = (m < M && k < K) ? A[m * sam + k * sak] : 0.f;
            int ni, kj;
            if (sbk == 1) { kj = idx & 15; ni = idx >> 4; } else { ni = idx & 63; kj = idx >> 6; }
            const int n = n0 + ni, kk = k0 + kj;
            sB[kj][ni] = (n < N && kk < K) ? Bm[kk * sbk + n * sbn] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = sA[k][ty * 4 + i]; b[i] = sB[k][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n >= N) continue;
            float v = acc[i][j] + (bias ? bias[n] : 0.f);
            v = act_apply(v, act);
            float* o = Cm + (size_t)m * ldc + n;
            *o = accumulate ? *o + v : v;
        }
    }
}

// db[n] += sum_m dy[m][n]: block = 64 columns x 4 row lanes over a 256-row slab (grid.y), atomics across slabs
__global__ __launch_bounds__(256) void colsum_small_kernel(const float* dy, float* db, int M, int N) {
    __shared__ float red[4][64];
    const int n = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const int r0 = blockIdx.y * 256, r1 = min(r0 + 256, M);
    float s = 0.f;
    if (n < N)
        for (int m = r0 + rl; m < r1; m += 4) s += dy[(size_t)m * N + n];
    red[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && n < N) atomicAdd(db + n, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// K = 1 dense layer (EmbedFC of the timestep, new_scripy.py:255-268 with input_dim 1): an outer product — the strided sgemm above spent
// 6 + 11 us on it.  y[m][n] = act(x[m] w[n] + b[n]);  dw[n] += sum_m dy[m][n] x[m], db[n] += sum_m dy[m][n], dx[m] = sum_n dy[m][n] w[n]
__global__ __launch_bounds__(256) void lin_k1_fwd_kernel(const float* x, const float* w, const float* b, float* y, int M, int N, int act) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M * N) return;
    const int m = i / N, n = i - m * N;
    y[i] = act_apply(x[m] * w[n] + (b ? b[n] : 0.f), act);
}
// block = 64 columns x 4 row lanes (the whole of M: M is the batch), fixed fold order
__global__ __launch_bounds__(256) void lin_k1_bwd_kernel(const float* x, const float* dy, float* dw, float* db, int M, int N) {
    __shared__ float r1[4][64], r2[4][64];
    const int n = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    float s1 = 0.f, s2 = 0.f;
    if (n < N)
        for (int m = rl; m < M; m += 4) { const float g = dy[(size_t)m * N + n]; s1 += g * x[m]; s2 += g; }
    r1[rl][threadIdx.x & 63] = s1; r2[rl][threadIdx.x & 63] = s2;
    __syncthreads();
    if (rl == 0 && n < N) {
        if (dw) dw[n] += r1[0][threadIdx.x] + r1[1][threadIdx.x] + r1[2][threadIdx.x] + r1[3][threadIdx.x];
        if (db) db[n] += r2[0][threadIdx.x] + r2[1][threadIdx.x] + r2[2][threadIdx.x] + r2[3][threadIdx.x];
    }
}

__global__ void act_fwd_kernel(const float* x, float* y, int n, int act) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) y[i] = act_apply(x[i], act);
}
__global__ void act_bwd_kernel(const float* x, const float* dy, float* dx, int n, int act) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) dx[i] = dy[i] * act_grad(x[i], act);
}

__global__ void onehot_mask_kernel(const int64_t* c, const float* mask, float* out, int B, int ncls, int flip) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * ncls) return;
    const int b = i / ncls, k = i - b * ncls;
    const float m = flip ? -(1.f - mask[b]) : mask[b];
    out[i] = (c[b] == (int64_t)k) ? m : 0.f * m;
}

}  // namespace

#define ST ((hipStream_t)s)

// chain.hip: per-sample sums over the pixels (mode 0: x; mode 1: x * y) left as partial arrays in the workspace, no fold launch
int dm_strip_partials(int mode, const void* x, const void* y, int dtype, int B, int HW, int C, float* out, const float** parts, int* rs_out,
                      hipStream_t st) {
    (void)out;
    if (mode == 0) return launch_strip<0>(x, nullptr, nullptr, dtype, B, 1, HW, 0, 1, HW, C, 1.f, st, parts, rs_out);
    return launch_strip<1>(x, y, nullptr, dtype, B, 1, HW, 0, 1, HW, C, 1.f, st, parts, rs_out);
}

extern "C" int dm_pool_hw(const void* x, int dtype, int B, int HW, int C, float* mean_bc, dm_stream_t s) {
    DM_CHECK_ARG(x && mean_bc && B > 0 && HW > 0 && C > 0, "dm_pool_hw: bad arguments");
    return launch_strip<0>(x, nullptr, mean_bc, dtype, B, 1, HW, 0, 1, HW, C, 1.f / (float)HW, ST);
}

// workgroups per sample for the vector kernels above: ~4 vectors per thread, at least one whole pixel row of vectors per pass
static int sr_grid(int HW, int CV, int B) {
    int64_t g = ((int64_t)HW * CV + 256 * 4 - 1) / (256 * 4);
    const int64_t need = (CV + 255) / 256;
    if (g < need) g = need;
    const int64_t cap = 8192 / (B > 0 ? B : 1) + 1;
    if (g > cap) g = cap;
    return (int)(g < need ? need : g);
}

extern "C" int dm_scale_residual_fwd(const void* x2, const void* res, const float* sgate, void* out, int dtype, int B, int HW, int C,
                                     float inv, dm_stream_t s) {
    DM_CHECK_ARG(x2 && res && out && B > 0 && HW > 0 && C > 0, "dm_scale_residual_fwd: bad arguments");
    DM_DISPATCH_DTYPE(dtype, {
        if (C % Elem<T>::VE == 0 && (((uintptr_t)x2 | (uintptr_t)res | (uintptr_t)out) & 15) == 0)
            hipLaunchKernelGGL((scale_res_fwd_vec_kernel<T>), dim3(sr_grid(HW, C / Elem<T>::VE, B), B), dim3(256), 0, ST, (const T*)x2, (const T*)res, sgate, (T*)out, HW, C, inv);
        else
            hipLaunchKernelGGL((scale_res_fwd_kernel<T>), dim3(grid_for((int64_t)B * HW * C, 256)), dim3(256), 0, ST, (const T*)x2, (const T*)res, sgate, (T*)out, B, HW, C, inv);
    });
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_scale_residual_bwd_reduce(const void* dout, const void* x2, int dtype, int B, int HW, int C, float inv, float* dsgate,
                                            dm_stream_t s) {
    DM_CHECK_ARG(dout && x2 && dsgate && B > 0 && HW > 0 && C > 0, "dm_scale_residual_bwd_reduce: bad arguments");
    return launch_strip<1>(dout, x2, dsgate, dtype, B, 1, HW, 0, 1, HW, C, inv, ST);
}

extern "C" int dm_scale_residual_bwd_apply(const void* dout, const float* sgate, const float* dy_mean, void* dx2, void* dres, int dtype,
                                           int B, int HW, int C, float inv, dm_stream_t s) {
    DM_CHECK_ARG(dout && dx2 && dres && B > 0 && HW > 0 && C > 0, "dm_scale_residual_bwd_apply: bad arguments");
    DM_DISPATCH_DTYPE(dtype, {
        if (C % Elem<T>::VE == 0 && (((uintptr_t)dout | (uintptr_t)dx2 | (uintptr_t)dres) & 15) == 0)
            hipLaunchKernelGGL((scale_res_bwd_vec_kernel<T>), dim3(sr_grid(HW, C / Elem<T>::VE, B), B), dim3(256), 0, ST, (const T*)dout, sgate, dy_mean, (T*)dx2, (T*)dres, HW, C, inv);
        else
            hipLaunchKernelGGL((scale_res_bwd_kernel<T>), dim3(grid_for((int64_t)B * HW * C, 256)), dim3(256), 0, ST, (const T*)dout, sgate, dy_mean, (T*)dx2, (T*)dres, B, HW, C, inv);
    });
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_ca_pool_fwd(const void* x, int dtype, int B, int H, int W, int C, float* xh, float* xw, dm_stream_t s) {
    DM_CHECK_ARG(x && xh && xw && B > 0 && H > 0 && W > 0 && C > 0, "dm_ca_pool_fwd: bad arguments");
    int rc = launch_strip<0>(x, nullptr, xh, dtype, B, H, W, W, 1, H * W, C, 1.f / (float)W, ST);
    if (rc) return rc;
    return launch_strip<0>(x, nullptr, xw, dtype, B, W, H, 1, W, H * W, C, 1.f / (float)H, ST);
}

extern "C" int dm_ca_pool_bwd(const float* dxh, const float* dxw, const void* dout_gate, void* dx, int dtype, int B, int H, int W, int C,
                              dm_stream_t s) {
    DM_CHECK_ARG(dxh && dxw && dx && B > 0 && H > 0 && W > 0 && C > 0, "dm_ca_pool_bwd: bad arguments");
    DM_DISPATCH_DTYPE(dtype, {
        if (!launch_ca_pix<T, false>((const T*)dout_gate, dxh, dxw, nullptr, nullptr, (T*)dx, B, H, W, C, ST))
            hipLaunchKernelGGL((ca_pool_bwd_kernel<T>), dim3(grid_for((int64_t)B * H * W * C, 256)), dim3(256), 0, ST, dxh, dxw, (const T*)dout_gate, (T*)dx, B, H, W, C);
    });
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_ca_gate_fwd(const void* x, const float* lh, const float* lw, const float* alpha, const float* beta, void* out, int dtype,
                              int B, int H, int W, int C, dm_stream_t s) {
    DM_CHECK_ARG(x && lh && lw && alpha && beta && out && B > 0 && H > 0 && W > 0 && C > 0, "dm_ca_gate_fwd: bad arguments");
    DM_DISPATCH_DTYPE(dtype, {
        if (!launch_ca_pix<T, true>((const T*)x, lh, lw, alpha, beta, (T*)out, B, H, W, C, ST))
            hipLaunchKernelGGL((ca_gate_fwd_kernel<T>), dim3(grid_for((int64_t)B * H * W * C, 256)), dim3(256), 0, ST, (const T*)x, lh, lw, alpha, beta, (T*)out, B, H, W, C);
    });
    DM_LAUNCH_CHECK();
    return DM_OK;
}

// dlh / dlw double as scratch for the strip sums (written in place); dalpha_dbeta: [0..1] += grads, [2..3] scratch (zeroed here)
extern "C" int dm_ca_gate_bwd(const void* x, const void* dout, const float* lh, const float* lw, const float* alpha, const float* beta,
                              void* dx_gate, float* dlh, float* dlw, float* dalpha_dbeta, int dtype, int B, int H, int W, int C,
                              dm_stream_t s) {
    DM_CHECK_ARG(x && dout && lh && lw && alpha && beta && dx_gate && dlh && dlw && dalpha_dbeta && B > 0 && H > 0 && W > 0 && C > 0,
                 "dm_ca_gate_bwd: bad arguments");
    DM_DISPATCH_DTYPE(dtype, {
        if (!launch_ca_pix<T, true>((const T*)dout, lh, lw, alpha, beta, (T*)dx_gate, B, H, W, C, ST))
            hipLaunchKernelGGL((ca_gate_bwd_dx_kernel<T>), dim3(grid_for((int64_t)B * H * W * C, 256)), dim3(256), 0, ST, (const T*)dout, lh, lw, alpha, beta, (T*)dx_gate, B, H, W, C);
    });
    DM_LAUNCH_CHECK();
    hipError_t e = hipMemsetAsync(dalpha_dbeta + 2, 0, 2 * sizeof(float), ST);
    if (e != hipSuccess) { dm_set_error("dm_ca_gate_bwd: memset failed"); return (int)e; }
    int rc = launch_strip<1>(dout, x, dlh, dtype, B, H, W, W, 1, H * W, C, 1.f, ST);
    if (rc) return rc;
    rc = launch_strip<1>(dout, x, dlw, dtype, B, W, H, 1, W, H * W, C, 1.f, ST);
    if (rc) return rc;
    const int64_t nh = (int64_t)B * H * C, nw = (int64_t)B * W * C;
    hipLaunchKernelGGL(ca_gate_bwd_strip_kernel, dim3(grid_for(nh, 256, 64)), dim3(256), 0, ST, dlh, lh, alpha, beta, 0, dlh, dalpha_dbeta + 2, nh);
    hipLaunchKernelGGL(ca_gate_bwd_strip_kernel, dim3(grid_for(nw, 256, 64)), dim3(256), 0, ST, dlw, lw, alpha, beta, 1, dlw, dalpha_dbeta + 2, nw);
    hipLaunchKernelGGL(ca_mix_bwd_kernel, dim3(1), dim3(64), 0, ST, alpha, beta, dalpha_dbeta + 2, dalpha_dbeta);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_sigmix_fwd(const float* x, const float* y, const float* gamma, float* xo, int n, dm_stream_t s) {
    DM_CHECK_ARG(x && y && gamma && xo && n > 0, "dm_sigmix_fwd: bad arguments");
    hipLaunchKernelGGL(sigmix_fwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ST, x, y, gamma, xo, n);
    DM_LAUNCH_CHECK();
    return DM_OK;
}
extern "C" int dm_sigmix_bwd(const float* dxo, const float* y, const float* gamma, float* dy, float* dgamma, int n, dm_stream_t s) {
    DM_CHECK_ARG(dxo && y && gamma && dy && dgamma && n > 0, "dm_sigmix_bwd: bad arguments");
    hipLaunchKernelGGL(sigmix_bwd_kernel, dim3(grid_for(n, 256, 64)), dim3(256), 0, ST, dxo, y, gamma, dy, dgamma, n);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

// dense.hip: the MFMA kernels for vector-width shapes
int dm_dense_fwd(const float* x, const float* w, const float* b, float* y, int M, int K, int N, int act, hipStream_t st);
int dm_dense_bwd(const float* x, const float* w, const float* gy, float* dx, float* dw, float* db, int M, int K, int N, hipStream_t st);
static inline bool dense_ok(int K, int N, const void* a, const void* b, const void* c, const void* d) {
    return K % 4 == 0 && N % 4 == 0 && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d) & 15) == 0;
}

extern "C" int dm_linear_fwd(const float* x, const float* w, const float* b, float* y, int M, int K, int N, int act, dm_stream_t s) {
    DM_CHECK_ARG(x && w && y && M > 0 && K > 0 && N > 0, "dm_linear_fwd: bad arguments");
    if (dense_ok(K, N, x, w, y, nullptr)) {
        dm_dense_fwd(x, w, b, y, M, K, N, act, ST);
        DM_LAUNCH_CHECK();
        return DM_OK;
    }
    if (K == 1) {
        hipLaunchKernelGGL(lin_k1_fwd_kernel, dim3(cdiv(M * N, 256)), dim3(256), 0, ST, x, w, b, y, M, N, act);
        DM_LAUNCH_CHECK();
        return DM_OK;
    }
    hipLaunchKernelGGL(sgemm_kernel, dim3(cdiv(N, 64), cdiv(M, 64)), dim3(256), 0, ST, x, w, b, y, M, N, K, (int64_t)K, (int64_t)1, (int64_t)1, (int64_t)K, N, act, 0);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_linear_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int M, int K, int N,
                             dm_stream_t s) {
    DM_CHECK_ARG(x && w && dy && M > 0 && K > 0 && N > 0, "dm_linear_bwd: bad arguments");
    if (dense_ok(K, N, x, w, dy, dx) && (((uintptr_t)dw) & 15) == 0 && (dw || !db)) {
        dm_dense_bwd(x, w, dy, dx, dw, db, M, K, N, ST);
        DM_LAUNCH_CHECK();
        return DM_OK;
    }
    if (dx)  // dx[m][k] = sum_n dy[m][n] w[n][k]
        hipLaunchKernelGGL(sgemm_kernel, dim3(cdiv(K, 64), cdiv(M, 64)), dim3(256), 0, ST, dy, w, (const float*)nullptr, dx, M, K, N, (int64_t)N, (int64_t)1, (int64_t)K, (int64_t)1, K, 0, 0);
    if (K == 1 && (dw || db)) {          // the timestep embedding's first layer: one launch for dw and db
        hipLaunchKernelGGL(lin_k1_bwd_kernel, dim3(cdiv(N, 64)), dim3(256), 0, ST, x, dy, dw, db, M, N);
        DM_LAUNCH_CHECK();
        return DM_OK;
    }
    if (dw)  // dw[n][k] += sum_m dy[m][n] x[m][k]
        hipLaunchKernelGGL(sgemm_kernel, dim3(cdiv(K, 64), cdiv(N, 64)), dim3(256), 0, ST, dy, x, (const float*)nullptr, dw, N, K, M, (int64_t)1, (int64_t)N, (int64_t)K, (int64_t)1, K, 0, 1);
    if (db) hipLaunchKernelGGL(colsum_small_kernel, dim3(cdiv(N, 64), cdiv(M, 256)), dim3(256), 0, ST, dy, db, M, N);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_act_fwd(const float* x, float* y, int n, int act, dm_stream_t s) {
    DM_CHECK_ARG(x && y && n > 0, "dm_act_fwd: bad arguments");
    hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ST, x, y, n, act);
    DM_LAUNCH_CHECK();
    return DM_OK;
}
extern "C" int dm_act_bwd(const float* x, const float* dy, float* dx, int n, int act, dm_stream_t s) {
    DM_CHECK_ARG(x && dy && dx && n > 0, "dm_act_bwd: bad arguments");
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ST, x, dy, dx, n, act);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_onehot_mask(const int64_t* c, const float* mask, float* out, int B, int ncls, int flip, dm_stream_t s) {
    DM_CHECK_ARG(c && mask && out && B > 0 && ncls > 0, "dm_onehot_mask: bad arguments");
    hipLaunchKernelGGL(onehot_mask_kernel, dim3(cdiv(B * ncls, 256)), dim3(256), 0, ST, c, mask, out, B, ncls, flip);
    DM_LAUNCH_CHECK();
    return DM_OK;
}
