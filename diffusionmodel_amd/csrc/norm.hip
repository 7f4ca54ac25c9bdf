// BatchNorm (training-mode statistics, fused activation, backward) and GroupNorm on NHWC tensors.
// All of these are HBM-bound streaming kernels: 16-byte vector accesses along the contiguous channel
// axis, per-channel reductions done as column sums of the [M][C] matrix (rows = B*H*W) with
// block-level partials that a second tiny kernel folds (deterministic, no atomics on the hot path).
#include "common.h"

namespace {


// ---------------------------------------------------------------------------------------------
// Column partial reduction skeleton: block handles rows [r0, r0+rows_per_block) of an [M][C] matrix.
// Thread (cv, rl): column vector cv (V elements), row lane rl; LDS folds the row lanes.
// F: functor object: init(col0) loads whatever is per-column (kept in registers for the whole row loop),
//    row(r, col0, v1[V], v2[V]) produces the two quantities to sum.
// ---------------------------------------------------------------------------------------------
template <int V, typename F>
__device__ inline void col_partial(int M, int C, int rows_per_block, float* p1, float* p2, F& f, int slots = 0) {
    __shared__ float red[2][2048];
    const int CVt = (C + V - 1) / V;  // column vectors in total
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(r0 + rows_per_block, M);
    for (int cbase = 0; cbase < CVt; cbase += 256) {
        const int ncv = min(256, CVt - cbase);     // column vectors in this chunk
        const int RL = 256 / ncv;                  // row lanes (>= 1)
        const int cv = threadIdx.x % ncv, rl = threadIdx.x / ncv;
        float a1[V], a2[V];
#pragma unroll
        for (int i = 0; i < V; ++i) a1[i] = a2[i] = 0.f;
        if (rl < RL) {
            const int col0 = (cbase + cv) * V;
            f.init(col0);
#pragma unroll 4
            for (int r = r0 + rl; r < r1; r += RL) {
                float v1[V], v2[V];
                f.row(r, col0, v1, v2);
#pragma unroll
                for (int i = 0; i < V; ++i) { a1[i] += v1[i]; a2[i] += v2[i]; }
            }
        }
        // fold row lanes: stage to LDS in chunks so that ncv*V*RL floats <= 2048
        __syncthreads();
        if (rl < RL) {
#pragma unroll
            for (int i = 0; i < V; ++i) {
                red[0][(rl * ncv + cv) * V + i] = a1[i];
                red[1][(rl * ncv + cv) * V + i] = a2[i];
            }
        }
        __syncthreads();
        for (int j = threadIdx.x; j < ncv * V; j += 256) {
            float s1 = 0.f, s2 = 0.f;
            for (int l = 0; l < RL; ++l) { s1 += red[0][l * ncv * V + j]; s2 += red[1][l * ncv * V + j]; }
            const int col = cbase * V + j;
            if (col < C) {
                if (slots > 0) {                       // [slots][C] fp64 accumulators (zeroed by the caller): the consumer folds the few slots itself;
                                                       // fp64 atomics: the order of arrival then moves the sums by ~1e-16, i.e. not at all in fp32
                    atomicAdd((double*)p1 + (size_t)(blockIdx.x % slots) * C + col, (double)s1);
                    atomicAdd((double*)p2 + (size_t)(blockIdx.x % slots) * C + col, (double)s2);
                } else {
                    p1[(size_t)blockIdx.x * C + col] = s1;
                    p2[(size_t)blockIdx.x * C + col] = s2;
                }
            }
        }
        __syncthreads();
    }
}

template <typename T, int V>
__device__ inline void load_cols(const T* p, float* f) {
    if constexpr (V == 1) f[0] = Elem<T>::ld(p);
    else load_vec<T>(p, f);
}
template <typename T, int V>
__device__ inline void store_cols(T* p, const float* f) {
    if constexpr (V == 1) Elem<T>::st(p, f[0]);
    else store_vec<T>(p, f);
}

template <typename T, int V>
struct StatsF {
    const T* z; int C;
    __device__ inline void init(int) {}
    __device__ inline void row(int r, int c0, float* v1, float* v2) const {
        load_cols<T, V>(z + (size_t)r * C + c0, v1);
#pragma unroll
        for (int i = 0; i < V; ++i) v2[i] = v1[i] * v1[i];
    }
};
template <typename T, int V>
__global__ __launch_bounds__(256) void col_stats_kernel(const T* z, int M, int C, int rpb, float* psum, float* psq) {
    StatsF<T, V> f{z, C};
    col_partial<V>(M, C, rpb, psum, psq, f);
}

// per-thread channel parameters held in registers by the streaming BN kernels
template <int V>
struct BnParams {
    float mu[V], rs[V], gm[V], bt[V];
    __device__ inline void load(int c0, int C, const float* mean, const float* rstd, const float* gamma, const float* beta) {
        if constexpr (V % 4 == 0) {       // vector path: c0 is a multiple of V, C a multiple of V -> aligned float4 loads
#pragma unroll
            for (int k = 0; k < V; k += 4) {
                const f32x4 m4 = *(const f32x4*)(mean + c0 + k), r4 = *(const f32x4*)(rstd + c0 + k);
                f32x4 g4 = {1.f, 1.f, 1.f, 1.f}, b4 = {0.f, 0.f, 0.f, 0.f};
                if (gamma) g4 = *(const f32x4*)(gamma + c0 + k);
                if (beta) b4 = *(const f32x4*)(beta + c0 + k);
#pragma unroll
                for (int j = 0; j < 4; ++j) { mu[k + j] = m4[j]; rs[k + j] = r4[j]; gm[k + j] = g4[j]; bt[k + j] = b4[j]; }
            }
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const int c = c0 + k < C ? c0 + k : C - 1;
                mu[k] = mean[c]; rs[k] = rstd[c];
                gm[k] = gamma ? gamma[c] : 1.f; bt[k] = beta ? beta[c] : 0.f;
            }
        }
    }
};

#ifndef SLOT_THREADS
#define SLOT_THREADS 512      // workgroup size of the slot-folding kernels: every workgroup folds slots x C doubles, so fewer, larger workgroups
#endif
// The same constants for the packed 16-bit GELU path (gelu_parts_fast2): xhat = z * rs + nmr, u = xhat * gm + bt on pairs of channels
struct BnPair {
    f32x2 rs[4], nmr[4], gm[4], bt[4];
    __device__ inline void set(const BnParams<8>& P) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            rs[j] = (f32x2){P.rs[2 * j], P.rs[2 * j + 1]};
            nmr[j] = (f32x2){-P.mu[2 * j] * P.rs[2 * j], -P.mu[2 * j + 1] * P.rs[2 * j + 1]};
            gm[j] = (f32x2){P.gm[2 * j], P.gm[2 * j + 1]};
            bt[j] = (f32x2){P.bt[2 * j], P.bt[2 * j + 1]};
        }
    }
    // y = gelu(bn(z)) for the 8 values of a vector
    __device__ inline void fwd_gelu(float* v) const {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x2 u = ((f32x2){v[2 * j], v[2 * j + 1]} * rs[j] + nmr[j]) * gm[j] + bt[j];
            f32x2 cdf, e;
            gelu_parts_fast2(u, cdf, e);
            const f32x2 y = u * cdf;
            v[2 * j] = y[0]; v[2 * j + 1] = y[1];
        }
    }
    // g = dy * gelu'(bn(z)) and xhat for channel pair j
    __device__ inline void bwd_gelu(int j, const float* zz, const float* dd, f32x2& g, f32x2& xh) const {
        xh = (f32x2){zz[2 * j], zz[2 * j + 1]} * rs[j] + nmr[j];
        const f32x2 u = xh * gm[j] + bt[j];
        f32x2 cdf, e;
        gelu_parts_fast2(u, cdf, e);
        g = (f32x2){dd[2 * j], dd[2 * j + 1]} * ((u * e) * 0.39894228040143267794f + cdf);
    }
};

// one wave per channel: lanes stride over the partial blocks
// Column sums over the per-block partial rows (nblk x C, up to 2048 rows on the 64x64 layers).  A 1024-thread workgroup owns
// 16 columns: thread = (column, one of 64 row lanes), so a 16-lane group reads one 64-byte line per row (a wave-load touches 4
// lines, not 64 as with a wave per column: that form spent 21 us on a 2048 x 128 array in the texture path alone) and every
// thread has 16 independent loads in flight.  Row lanes fold in a fixed order (shuffles, then 16 per-wave values through LDS):
// deterministic, fp64 accumulation.
__device__ __forceinline__ double colsum16(const float* __restrict__ part, int nblk, int C, int c, bool ok, double* red) {
    const int rl = threadIdx.x >> 4;
    double s = 0.0;
    if (ok) {
        int b = rl;
        for (; b + 64 * 7 < nblk; b += 64 * 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = part[(size_t)(b + 64 * j) * C + c];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (double)v[j];
        }
        for (; b < nblk; b += 64) s += (double)part[(size_t)b * C + c];
    }
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const int wave = threadIdx.x >> 6, col = threadIdx.x & 15;
    __syncthreads();                                   // `red` may still be read from a previous call
    if ((threadIdx.x & 63) < 16) red[wave * 16 + col] = s;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x < 16) {
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[w * 16 + col];
    }
    return t;                                          // valid in threads 0..15 (column = threadIdx.x)
}

__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* psum, const float* psq, int nblk, int M, int C,
                                                           float eps, float mom, float* mean, float* rstd, float* rmean,
                                                           float* rvar) {
    __shared__ double red[256];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15);
    const double s1 = colsum16(psum, nblk, C, c, c < C, red);
    const double s2 = colsum16(psq, nblk, C, c, c < C, red);
    if (threadIdx.x < 16 && c < C) {
        const double mu = s1 / M;
        double var = s2 / M - mu * mu;
        if (var < 0.0) var = 0.0;
        mean[c] = (float)mu;
        rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (rmean) {
            const double unb = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
            rmean[c] = (float)((1.0 - mom) * (double)rmean[c] + mom * mu);
            rvar[c] = (float)((1.0 - mom) * (double)rvar[c] + mom * unb);
        }
    }
}

__global__ __launch_bounds__(1024) void col_reduce_kernel(const float* part, int nblk, int C, float* out, int accumulate) {
    __shared__ double red[256];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15);
    const double s = colsum16(part, nblk, C, c, c < C, red);
    if (threadIdx.x < 16 && c < C) out[c] = accumulate ? out[c] + (float)s : (float)s;
}

// two partial arrays of the same shape in one launch (dbeta / dgamma of a BatchNorm backward)
__global__ __launch_bounds__(1024) void col_reduce2_kernel(const float* part1, const float* part2, int nblk, int C, float* out1, float* out2) {
    __shared__ double red[256];
    int c = blockIdx.x * 16 + (threadIdx.x & 15);
    const bool ok = c < 2 * C;
    const float* part = c < C ? part1 : part2;
    float* out = c < C ? out1 : out2;
    if (c >= C) c -= C;
    const double s = colsum16(part, nblk, C, c, ok, red);
    if (threadIdx.x < 16 && ok) out[c] = (float)s;
}

// Streaming kernels: a thread keeps ONE column vector (its BN parameters live in registers) and strides over
// rows; the global stride is rounded down to a multiple of the vectors per row so the column never changes.
template <typename T, int V>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const T* z, T* y, int64_t nvec, int CV, const float* mean,
                                                         const float* rstd, const float* gamma, const float* beta, int act) {
    const int64_t G = ((int64_t)gridDim.x * 256 / CV) * CV;
    const int64_t g0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g0 >= G) return;
    BnParams<V> P;
    P.load((int)(g0 % CV) * V, CV * V, mean, rstd, gamma, beta);
    if constexpr (sizeof(T) == 2 && V == 8) {
        if (act == DM_ACT_GELU) {                    // packed fp32 math, see gelu_parts_fast2
            BnPair Q;
            Q.set(P);
#pragma unroll 2
            for (int64_t i = g0; i < nvec; i += G) {
                float v[V];
                load_cols<T, V>(z + i * V, v);
                Q.fwd_gelu(v);
                store_cols<T, V>(y + i * V, v);
            }
            return;
        }
    }
#pragma unroll 2
    for (int64_t i = g0; i < nvec; i += G) {
        float v[V];
        load_cols<T, V>(z + i * V, v);
#pragma unroll
        for (int k = 0; k < V; ++k) v[k] = act_apply_t<T>((v[k] - P.mu[k]) * P.rs[k] * P.gm[k] + P.bt[k], act);
        store_cols<T, V>(y + i * V, v);
    }
}

// bn_act_fwd with the statistics folded in: psum / psq are [S][C] slot accumulators the producing convolution's epilogue added its
// per-tile column sums into (DmConv.stat_slots).  Every workgroup folds the S slots of all C channels into LDS (S * C * 8 bytes from
// L2 — S = 8: 8 KiB at C = 128) instead of a separate finalize launch; workgroup 0 also publishes mean / rstd (the backward pass
// needs them) and updates the running statistics (nn.BatchNorm2d: momentum, unbiased variance).
template <typename T, int V>
__global__ __launch_bounds__(SLOT_THREADS) void bn_act_fwd_slots_kernel(const T* z, T* y, int64_t nvec, int CV, const double* psum, const double* psq,
                                                               int S, int M, float eps, float mom, const float* gamma, const float* beta,
                                                               int act, float* mean_out, float* rstd_out, float* rmean, float* rvar) {
    extern __shared__ float sstat[];                   // [2][C]
    const int C = CV * V;
    for (int c = threadIdx.x; c < C; c += SLOT_THREADS) {
        double s1 = 0.0, s2 = 0.0;
        for (int k = 0; k < S; ++k) { s1 += psum[(size_t)k * C + c]; s2 += psq[(size_t)k * C + c]; }
        const double mu = s1 / M;
        double var = s2 / M - mu * mu;
        if (var < 0.0) var = 0.0;
        const float rs = (float)(1.0 / sqrt(var + (double)eps));
        sstat[c] = (float)mu;
        sstat[C + c] = rs;
        if (blockIdx.x == 0) {
            mean_out[c] = (float)mu;
            rstd_out[c] = rs;
            if (rmean) {
                const double unb = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
                rmean[c] = (float)((1.0 - mom) * (double)rmean[c] + mom * mu);
                rvar[c] = (float)((1.0 - mom) * (double)rvar[c] + mom * unb);
            }
        }
    }
    __syncthreads();
    const int64_t G = ((int64_t)gridDim.x * SLOT_THREADS / CV) * CV;
    const int64_t g0 = (int64_t)blockIdx.x * SLOT_THREADS + threadIdx.x;
    if (g0 >= G) return;
    BnParams<V> P;
    P.load((int)(g0 % CV) * V, C, sstat, sstat + C, gamma, beta);
    if constexpr (sizeof(T) == 2 && V == 8) {
        if (act == DM_ACT_GELU) {                    // packed fp32 math, see gelu_parts_fast2
            BnPair Q;
            Q.set(P);
#pragma unroll 2
            for (int64_t i = g0; i < nvec; i += G) {
                float v[V];
                load_cols<T, V>(z + i * V, v);
                Q.fwd_gelu(v);
                store_cols<T, V>(y + i * V, v);
            }
            return;
        }
    }
#pragma unroll 2
    for (int64_t i = g0; i < nvec; i += G) {
        float v[V];
        load_cols<T, V>(z + i * V, v);
#pragma unroll
        for (int k = 0; k < V; ++k) v[k] = act_apply_t<T>((v[k] - P.mu[k]) * P.rs[k] * P.gm[k] + P.bt[k], act);
        store_cols<T, V>(y + i * V, v);
    }
}

// PK: the packed 16-bit GELU path (gelu_parts_fast2) as its own instantiation, so that only one set of channel constants is live
// in the row loop (with both, the kernel went from < 128 to 136 VGPRs and ran 3 us slower per launch in the train step)
template <typename T, int V, bool PK>
struct BwdRedF {
    const T* z; const T* dy; int C; const float* mean; const float* rstd; const float* gamma; const float* beta; int act;
    BnParams<PK ? 1 : V> P;
    BnPair Q;
    __device__ inline void init(int c0) {
        if constexpr (PK) {
            BnParams<V> P8;
            P8.load(c0, C, mean, rstd, gamma, beta);
            Q.set(P8);
        } else {
            P.load(c0, C, mean, rstd, gamma, beta);
        }
    }
    __device__ inline void row(int r, int c0, float* v1, float* v2) const {
        float zz[V], dd[V];
        load_cols<T, V>(z + (size_t)r * C + c0, zz);
        load_cols<T, V>(dy + (size_t)r * C + c0, dd);
        if constexpr (PK) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x2 g, xh;
                Q.bwd_gelu(j, zz, dd, g, xh);
                const f32x2 gx = g * xh;
                v1[2 * j] = g[0]; v1[2 * j + 1] = g[1];
                v2[2 * j] = gx[0]; v2[2 * j + 1] = gx[1];
            }
        } else {
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const float xh = (zz[i] - P.mu[i]) * P.rs[i];
                const float g = dd[i] * act_grad_t<T>(xh * P.gm[i] + P.bt[i], act);
                v1[i] = g;
                v2[i] = g * xh;
            }
        }
    }
};
template <typename T, int V, bool PK>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* z, const T* dy, int M, int C, const float* mean,
                                                            const float* rstd, const float* gamma, const float* beta,
                                                            int act, int rpb, float* p1, float* p2, int slots) {
    BwdRedF<T, V, PK> f{z, dy, C, mean, rstd, gamma, beta, act, {}, {}};
    col_partial<V>(M, C, rpb, p1, p2, f, slots);
}

template <typename T, int V>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* z, const T* dy, T* dz, int64_t nvec, int CV, float invM,
                                                           const float* mean, const float* rstd, const float* gamma,
                                                           const float* beta, int act, const float* s1, const float* s2) {
    const int64_t G = ((int64_t)gridDim.x * 256 / CV) * CV;
    const int64_t g0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g0 >= G) return;
    const int c0 = (int)(g0 % CV) * V;
    BnParams<V> P;
    P.load(c0, CV * V, mean, rstd, gamma, beta);
    float m1[V], m2[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { m1[k] = s1[c0 + k] * invM; m2[k] = s2[c0 + k] * invM; }
    if constexpr (sizeof(T) == 2 && V == 8) {
        if (act == DM_ACT_GELU) {                    // packed fp32 math, see gelu_parts_fast2
            BnPair Q;
            Q.set(P);
            f32x2 nm1[4], nm2[4], gr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                nm1[j] = (f32x2){-m1[2 * j], -m1[2 * j + 1]};
                nm2[j] = (f32x2){-m2[2 * j], -m2[2 * j + 1]};
                gr[j] = Q.gm[j] * Q.rs[j];
            }
#pragma unroll 2
            for (int64_t i = g0; i < nvec; i += G) {
                float zz[V], dd[V];
                load_cols<T, V>(z + i * V, zz);
                load_cols<T, V>(dy + i * V, dd);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x2 g, xh;
                    Q.bwd_gelu(j, zz, dd, g, xh);
                    const f32x2 r = (xh * nm2[j] + (g + nm1[j])) * gr[j];
                    zz[2 * j] = r[0]; zz[2 * j + 1] = r[1];
                }
                store_cols<T, V>(dz + i * V, zz);
            }
            return;
        }
    }
#pragma unroll 2
    for (int64_t i = g0; i < nvec; i += G) {
        float zz[V], dd[V];
        load_cols<T, V>(z + i * V, zz);
        load_cols<T, V>(dy + i * V, dd);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float xh = (zz[k] - P.mu[k]) * P.rs[k];
            const float g = dd[k] * act_grad_t<T>(xh * P.gm[k] + P.bt[k], act);
            zz[k] = P.gm[k] * P.rs[k] * (g - m1[k] - xh * m2[k]);
        }
        store_cols<T, V>(dz + i * V, zz);
    }
}

// bn_bwd_apply with the column sums folded in: p1 / p2 are the [S][C] slot accumulators of bn_bwd_reduce_kernel (slots mode);
// workgroup 0 publishes dbeta = sum(g), dgamma = sum(g * xhat) — the parameter gradients — instead of a separate reduce launch.
template <typename T, int V>
__global__ __launch_bounds__(SLOT_THREADS) void bn_bwd_apply_slots_kernel(const T* z, const T* dy, T* dz, int64_t nvec, int CV, float invM,
                                                                 const float* mean, const float* rstd, const float* gamma, const float* beta,
                                                                 int act, const double* p1, const double* p2, int S, float* dbeta, float* dgamma) {
    extern __shared__ float ssum[];                    // [2][C]
    const int C = CV * V;
    for (int c = threadIdx.x; c < C; c += SLOT_THREADS) {
        double a = 0.0, b = 0.0;
        for (int k = 0; k < S; ++k) { a += p1[(size_t)k * C + c]; b += p2[(size_t)k * C + c]; }
        ssum[c] = (float)a;
        ssum[C + c] = (float)b;
        if (blockIdx.x == 0) { dbeta[c] = (float)a; dgamma[c] = (float)b; }
    }
    __syncthreads();
    const int64_t G = ((int64_t)gridDim.x * SLOT_THREADS / CV) * CV;
    const int64_t g0 = (int64_t)blockIdx.x * SLOT_THREADS + threadIdx.x;
    if (g0 >= G) return;
    const int c0 = (int)(g0 % CV) * V;
    BnParams<V> P;
    P.load(c0, C, mean, rstd, gamma, beta);
    float m1[V], m2[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { m1[k] = ssum[c0 + k] * invM; m2[k] = ssum[C + c0 + k] * invM; }
    if constexpr (sizeof(T) == 2 && V == 8) {
        if (act == DM_ACT_GELU) {                    // packed fp32 math, see gelu_parts_fast2
            BnPair Q;
            Q.set(P);
            f32x2 nm1[4], nm2[4], gr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                nm1[j] = (f32x2){-m1[2 * j], -m1[2 * j + 1]};
                nm2[j] = (f32x2){-m2[2 * j], -m2[2 * j + 1]};
                gr[j] = Q.gm[j] * Q.rs[j];
            }
#pragma unroll 2
            for (int64_t i = g0; i < nvec; i += G) {
                float zz[V], dd[V];
                load_cols<T, V>(z + i * V, zz);
                load_cols<T, V>(dy + i * V, dd);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x2 g, xh;
                    Q.bwd_gelu(j, zz, dd, g, xh);
                    const f32x2 r = (xh * nm2[j] + (g + nm1[j])) * gr[j];
                    zz[2 * j] = r[0]; zz[2 * j + 1] = r[1];
                }
                store_cols<T, V>(dz + i * V, zz);
            }
            return;
        }
    }
#pragma unroll 2
    for (int64_t i = g0; i < nvec; i += G) {
        float zz[V], dd[V];
        load_cols<T, V>(z + i * V, zz);
        load_cols<T, V>(dy + i * V, dd);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float xh = (zz[k] - P.mu[k]) * P.rs[k];
            const float g = dd[k] * act_grad_t<T>(xh * P.gm[k] + P.bt[k], act);
            zz[k] = P.gm[k] * P.rs[k] * (g - m1[k] - xh * m2[k]);
        }
        store_cols<T, V>(dz + i * V, zz);
    }
}

__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* rmean, const float* rvar,
                               const float* cbias, float eps, int C, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float s = gamma[c] / sqrtf(rvar[c] + eps);
    scale[c] = s;
    shift[c] = ((cbias ? cbias[c] : 0.f) - rmean[c]) * s + beta[c];
}

// ---------------------------------------------------------------------------------------------
// GroupNorm: one block per (b, group). Elements: HW pixels x cg channels (contiguous chunk per pixel).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gn_fwd_kernel(const T* x, T* y, int HW, int C, int G, float eps, const float* gamma,
                                                     const float* beta, int act, float* mean_o, float* rstd_o) {
    __shared__ float red[16];
    const int b = blockIdx.x / G, g = blockIdx.x - b * G;
    const int cg = C / G;
    const T* xb = x + (size_t)b * HW * C + g * cg;
    T* yb = y + (size_t)b * HW * C + g * cg;
    const int n = HW * cg;
    float s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int pix = i / cg, c = i - pix * cg;
        const float v = Elem<T>::ld(xb + (size_t)pix * C + c);
        s1 += v; s2 += v * v;
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    const float mu = s1 / n;
    float var = s2 / n - mu * mu;
    var = var < 0.f ? 0.f : var;
    const float rs = rsqrtf(var + eps);
    if (threadIdx.x == 0) { mean_o[blockIdx.x] = mu; rstd_o[blockIdx.x] = rs; }
    for (int i = threadIdx.x; i < n; i += 256) {
        const int pix = i / cg, c = i - pix * cg;
        const float v = Elem<T>::ld(xb + (size_t)pix * C + c);
        const int ch = g * cg + c;
        Elem<T>::st(yb + (size_t)pix * C + c, act_apply((v - mu) * rs * gamma[ch] + beta[ch], act));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_kernel(const T* x, const T* dy, T* dx, int HW, int C, int G, const float* gamma,
                                                     const float* beta, int act, const float* mean_i, const float* rstd_i,
                                                     float* dgamma, float* dbeta) {
    __shared__ float red[16];
    extern __shared__ float chacc[];  // [2][cg]
    const int b = blockIdx.x / G, g = blockIdx.x - b * G;
    const int cg = C / G;
    const size_t base = (size_t)b * HW * C + g * cg;
    const int n = HW * cg;
    const float mu = mean_i[blockIdx.x], rs = rstd_i[blockIdx.x];
    for (int i = threadIdx.x; i < 2 * cg; i += 256) chacc[i] = 0.f;
    __syncthreads();
    float s1 = 0.f, s2 = 0.f;
    // thread's channel is i % cg; with 256 % cg == 0 it is fixed per thread, otherwise use LDS atomics per element
    const bool fixed = (256 % cg) == 0;
    float cg1 = 0.f, cg2 = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int pix = i / cg, c = i - pix * cg;
        const int ch = g * cg + c;
        const float xh = (Elem<T>::ld(x + base + (size_t)pix * C + c) - mu) * rs;
        const float u = xh * gamma[ch] + beta[ch];
        const float gg = Elem<T>::ld(dy + base + (size_t)pix * C + c) * act_grad(u, act);
        s1 += gg * gamma[ch];
        s2 += gg * gamma[ch] * xh;
        if (fixed) { cg1 += gg; cg2 += gg * xh; }
        else { atomicAdd(&chacc[c], gg); atomicAdd(&chacc[cg + c], gg * xh); }
    }
    if (fixed && threadIdx.x < n) { const int c = threadIdx.x % cg; atomicAdd(&chacc[c], cg1); atomicAdd(&chacc[cg + c], cg2); }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    __syncthreads();
    for (int i = threadIdx.x; i < cg; i += 256) {
        atomicAdd(dbeta + g * cg + i, chacc[i]);
        atomicAdd(dgamma + g * cg + i, chacc[cg + i]);
    }
    const float m1 = s1 / n, m2 = s2 / n;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int pix = i / cg, c = i - pix * cg;
        const int ch = g * cg + c;
        const float xh = (Elem<T>::ld(x + base + (size_t)pix * C + c) - mu) * rs;
        const float u = xh * gamma[ch] + beta[ch];
        const float gg = Elem<T>::ld(dy + base + (size_t)pix * C + c) * act_grad(u, act);
        Elem<T>::st(dx + base + (size_t)pix * C + c, rs * (gg * gamma[ch] - m1 - xh * m2));
    }
}

// 16-byte-vector forms of the two kernels above (cg % VE == 0, 256 % (cg / VE) == 0, aligned): a thread keeps ONE channel
// vector of the group (its gamma / beta live in registers) and strides over pixels; the second pass re-reads from L2 / MALL.
template <typename T, int V>
__global__ __launch_bounds__(256) void gn_fwd_vec_kernel(const T* x, T* y, int HW, int C, int G, float eps, const float* gamma,
                                                         const float* beta, int act, float* mean_o, float* rstd_o) {
    __shared__ float red[16];
    const int b = blockIdx.x / G, g = blockIdx.x - b * G;
    const int cg = C / G, cv = cg / V;                 // vectors per pixel of this group
    const int v0 = threadIdx.x % cv, p0 = threadIdx.x / cv, pstep = 256 / cv;
    const size_t base = (size_t)b * HW * C + g * cg + v0 * V;
    float gm[V], bt[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { gm[k] = gamma[g * cg + v0 * V + k]; bt[k] = beta[g * cg + v0 * V + k]; }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll 4
    for (int pix = p0; pix < HW; pix += pstep) {
        float v[V];
        load_vec<T>(x + base + (size_t)pix * C, v);
#pragma unroll
        for (int k = 0; k < V; ++k) { s1 += v[k]; s2 += v[k] * v[k]; }
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    const float n = (float)HW * cg;
    const float mu = s1 / n;
    float var = s2 / n - mu * mu;
    var = var < 0.f ? 0.f : var;
    const float rs = rsqrtf(var + eps);
    if (threadIdx.x == 0) { mean_o[blockIdx.x] = mu; rstd_o[blockIdx.x] = rs; }
#pragma unroll 4
    for (int pix = p0; pix < HW; pix += pstep) {
        float v[V];
        load_vec<T>(x + base + (size_t)pix * C, v);
#pragma unroll
        for (int k = 0; k < V; ++k) v[k] = act_apply_t<T>((v[k] - mu) * rs * gm[k] + bt[k], act);
        store_vec<T>(y + base + (size_t)pix * C, v);
    }
}

template <typename T, int V>
__global__ __launch_bounds__(256) void gn_bwd_vec_kernel(const T* x, const T* dy, T* dx, int HW, int C, int G, const float* gamma,
                                                         const float* beta, int act, const float* mean_i, const float* rstd_i,
                                                         float* dgamma, float* dbeta) {
    __shared__ float red[16];
    extern __shared__ float chacc[];  // [2][cg]
    const int b = blockIdx.x / G, g = blockIdx.x - b * G;
    const int cg = C / G, cv = cg / V;
    const int v0 = threadIdx.x % cv, p0 = threadIdx.x / cv, pstep = 256 / cv;
    const size_t base = (size_t)b * HW * C + g * cg + v0 * V;
    const float mu = mean_i[blockIdx.x], rs = rstd_i[blockIdx.x];
    float gm[V], bt[V], a1[V], a2[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { gm[k] = gamma[g * cg + v0 * V + k]; bt[k] = beta[g * cg + v0 * V + k]; a1[k] = 0.f; a2[k] = 0.f; }
    for (int i = threadIdx.x; i < 2 * cg; i += 256) chacc[i] = 0.f;
    __syncthreads();
    float s1 = 0.f, s2 = 0.f;
#pragma unroll 2
    for (int pix = p0; pix < HW; pix += pstep) {
        float xv[V], gv[V];
        load_vec<T>(x + base + (size_t)pix * C, xv);
        load_vec<T>(dy + base + (size_t)pix * C, gv);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float xh = (xv[k] - mu) * rs;
            const float gg = gv[k] * act_grad_t<T>(xh * gm[k] + bt[k], act);
            s1 += gg * gm[k];
            s2 += gg * gm[k] * xh;
            a1[k] += gg;
            a2[k] += gg * xh;
        }
    }
#pragma unroll
    for (int k = 0; k < V; ++k) { atomicAdd(&chacc[v0 * V + k], a1[k]); atomicAdd(&chacc[cg + v0 * V + k], a2[k]); }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    __syncthreads();
    for (int i = threadIdx.x; i < cg; i += 256) {
        atomicAdd(dbeta + g * cg + i, chacc[i]);
        atomicAdd(dgamma + g * cg + i, chacc[cg + i]);
    }
    const float n = (float)HW * cg;
    const float m1 = s1 / n, m2 = s2 / n;
#pragma unroll 2
    for (int pix = p0; pix < HW; pix += pstep) {
        float xv[V], gv[V];
        load_vec<T>(x + base + (size_t)pix * C, xv);
        load_vec<T>(dy + base + (size_t)pix * C, gv);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float xh = (xv[k] - mu) * rs;
            const float gg = gv[k] * act_grad_t<T>(xh * gm[k] + bt[k], act);
            xv[k] = rs * (gg * gm[k] - m1 - xh * m2);
        }
        store_vec<T>(dx + base + (size_t)pix * C, xv);
    }
}

// ---- GroupNorm on large images: pixel slabs instead of one workgroup per (sample, group) ----------------------------------------
// The kernels above give a workgroup the cg channels of ONE group: 32 bytes out of every 256-byte pixel row at C = 128, G = 8 —
// strided 32-byte reads, 512 workgroups, two serial passes: 1.1 TB/s on the 64x64 `out` head (180 us backward).  Here a workgroup
// takes a slab of whole pixel rows of one sample (fully coalesced), a thread keeps one channel vector:
//   stats kernel : per-slab partial sums of every group -> scratch[b][slab][g][2]      (backward: also dgamma / dbeta)
//   apply kernel : every workgroup adds up its sample's slab partials in a fixed order (deterministic), then streams.
template <typename T, int V, bool BWD>
__global__ __launch_bounds__(256) void gn_slab_stats_kernel(const T* x, const T* dy, int HW, int C, int G, int pps, const float* gamma,
                                                            const float* beta, int act, const float* mean_i, const float* rstd_i,
                                                            float* part, float* part_db, float* part_dg) {
    extern __shared__ float sm[];                        // backward: [256][2 V] per-thread channel sums
    const int b = blockIdx.y, slab = blockIdx.x, S = gridDim.x;
    const int CVt = C / V, cg = C / G;
    const int v0 = threadIdx.x % CVt, p0 = threadIdx.x / CVt, pstep = 256 / CVt;
    const int c0 = v0 * V, g = c0 / cg;
    float gm[V], bt[V], a1[V], a2[V];
    float mu = 0.f, rs = 1.f;
    if constexpr (BWD) {
        mu = mean_i[b * G + g];
        rs = rstd_i[b * G + g];
#pragma unroll
        for (int k = 0; k < V; ++k) { gm[k] = gamma[c0 + k]; bt[k] = beta[c0 + k]; a1[k] = 0.f; a2[k] = 0.f; }
    }
    const size_t base = (size_t)b * HW * C + c0;
    const int lo = slab * pps, hi = min(HW, lo + pps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll 4
    for (int pix = lo + p0; pix < hi; pix += pstep) {
        float xv[V];
        load_vec<T>(x + base + (size_t)pix * C, xv);
        if constexpr (!BWD) {
#pragma unroll
            for (int k = 0; k < V; ++k) { s1 += xv[k]; s2 += xv[k] * xv[k]; }
        } else {
            float gv[V];
            load_vec<T>(dy + base + (size_t)pix * C, gv);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float xh = (xv[k] - mu) * rs;
                const float gg = gv[k] * act_grad_t<T>(xh * gm[k] + bt[k], act);
                s1 += gg * gm[k];
                s2 += gg * gm[k] * xh;
                a1[k] += gg;
                a2[k] += gg * xh;
            }
        }
    }
    // group sums in a fixed order (the forward result must not depend on the run: graph replay == eager, bit for bit)
    __shared__ float thr[512];
    thr[2 * threadIdx.x] = s1;
    thr[2 * threadIdx.x + 1] = s2;
    if constexpr (BWD) {
#pragma unroll
        for (int k = 0; k < V; ++k) { sm[threadIdx.x * 2 * V + k] = a1[k]; sm[threadIdx.x * 2 * V + V + k] = a2[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * G) {
        const int gg = threadIdx.x >> 1, which = threadIdx.x & 1, vpg = cg / V;      // vectors per group and pixel
        float t = 0.f;
        for (int p = 0; p < pstep; ++p)
            for (int v = 0; v < vpg; ++v) t += thr[2 * (p * CVt + gg * vpg + v) + which];
        part[((size_t)b * S + slab) * 2 * G + threadIdx.x] = t;
    }
    if constexpr (BWD) {                                 // per-channel sums of this slab -> one row of the dbeta / dgamma partial arrays
        for (int i = threadIdx.x; i < 2 * C; i += 256) {
            const int which = i / C, c = i - which * C, v = c / V, k = c - v * V;
            float t = 0.f;
            for (int p = 0; p < pstep; ++p) t += sm[(p * CVt + v) * 2 * V + which * V + k];
            (which ? part_dg : part_db)[((size_t)b * S + slab) * C + c] = t;
        }
    }
}

template <typename T, int V, bool BWD>
__global__ __launch_bounds__(256) void gn_slab_apply_kernel(const T* x, const T* dy, T* out, int HW, int C, int G, int pps, int S, float eps,
                                                            const float* gamma, const float* beta, int act, const float* part,
                                                            float* mean_io, float* rstd_io) {
    const int b = blockIdx.y, slab = blockIdx.x;
    const int CVt = C / V, cg = C / G;
    const int v0 = threadIdx.x % CVt, p0 = threadIdx.x / CVt, pstep = 256 / CVt;
    const int c0 = v0 * V, g = c0 / cg;
    float t1 = 0.f, t2 = 0.f;
    for (int k = 0; k < S; ++k) {                         // the same order in every workgroup of the sample
        t1 += part[((size_t)b * S + k) * 2 * G + 2 * g];
        t2 += part[((size_t)b * S + k) * 2 * G + 2 * g + 1];
    }
    const float n = (float)HW * cg;
    float gm[V], bt[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { gm[k] = gamma[c0 + k]; bt[k] = beta[c0 + k]; }
    float mu, rs, m1 = 0.f, m2 = 0.f;
    if constexpr (!BWD) {
        mu = t1 / n;
        float var = t2 / n - mu * mu;
        var = var < 0.f ? 0.f : var;
        rs = rsqrtf(var + eps);
        if (slab == 0 && p0 == 0 && c0 == g * cg) { mean_io[b * G + g] = mu; rstd_io[b * G + g] = rs; }
    } else {
        mu = mean_io[b * G + g];
        rs = rstd_io[b * G + g];
        m1 = t1 / n;
        m2 = t2 / n;
    }
    const size_t base = (size_t)b * HW * C + c0;
    const int lo = slab * pps, hi = min(HW, lo + pps);
#pragma unroll 4
    for (int pix = lo + p0; pix < hi; pix += pstep) {
        float xv[V];
        load_vec<T>(x + base + (size_t)pix * C, xv);
        if constexpr (!BWD) {
#pragma unroll
            for (int k = 0; k < V; ++k) xv[k] = act_apply_t<T>((xv[k] - mu) * rs * gm[k] + bt[k], act);
        } else {
            float gv[V];
            load_vec<T>(dy + base + (size_t)pix * C, gv);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float xh = (xv[k] - mu) * rs;
                const float gg = gv[k] * act_grad_t<T>(xh * gm[k] + bt[k], act);
                xv[k] = rs * (gg * gm[k] - m1 - xh * m2);
            }
        }
        store_vec<T>(out + base + (size_t)pix * C, xv);
    }
}

// slab geometry for a (B, HW) problem: ~2048 workgroups, >= 64 pixels per slab; 0 = use the per-group kernels
static int gn_slabs(int B, int HW, int C, int G, int V, const void* a, const void* b2, const void* c2, int* pps) {
    const int cg = C / G;
    if (C % V != 0 || cg % V != 0 || 256 % (C / V) != 0 || C / V > 256 || HW < 1024 || dm_g_ws == nullptr) return 0;
    if ((((uintptr_t)a | (uintptr_t)b2 | (uintptr_t)c2) & 15) != 0) return 0;
    int S = 2048 / B;
    if (S > HW / 64) S = HW / 64;
    if (S < 2) return 0;
    *pps = cdiv(HW, S);
    S = cdiv(HW, *pps);
    if ((int64_t)B * S * (2 * G + 2 * C + 64) * (int64_t)sizeof(float) > dm_g_ws_bytes) return 0;
    return S;
}

// grid of the per-thread-column streaming kernels: >= 8 vectors per thread (the 4 x V channel parameters a
// thread keeps in registers are then amortised), at most 8 workgroups per CU, at least one pass over a row
int stream_grid(int64_t nvec, int cv) {
    // 8 vectors per thread on large tensors; small ones (the 8x8 / 16x16 levels: 0.5 M vectors) would then get one workgroup per
    // CU and run at 1.6 TB/s: at least ~1024 workgroups as long as a thread keeps 2 vectors
    int per = 8;
    while (per > 2 && nvec / (256 * per) < 1024) per /= 2;
    int64_t g = (nvec + 256 * per - 1) / (256 * per);
    if (g > 2048) g = 2048;
    const int64_t need = (cv + 255) / 256;
    if (g < need) g = need;
    return (int)(g < 1 ? 1 : g);
}

// rows per workgroup of the column-reduction kernels: small matrices get thinner slabs so that the grid
// still covers the chip
int g_bn_packed = getenv("DM_BN_REDUCE_PACKED") ? atoi(getenv("DM_BN_REDUCE_PACKED")) : 1;   // A/B knob for the packed reduce kernel
int cs_rows(int M) {                                  // ~1024 workgroups: 8 ... 256 rows each (a power of two)
    int r = 8;
    while (r < 256 && (int64_t)r * 2048 <= M) r *= 2;
    return r;
}

template <typename T>
bool vec_ok(int C, const void* a, const void* b = nullptr, const void* c = nullptr) {
    const uintptr_t m = (uintptr_t)a | (uintptr_t)b | (uintptr_t)c;
    return (C % Elem<T>::VE) == 0 && (m & 15) == 0;
}

}  // namespace

extern "C" int dm_colstat_blocks(int M) { return cdiv(M, cs_rows(M)); }

extern "C" int dm_col_stats(const void* z, int dtype, int M, int C, float* psum, float* psq, dm_stream_t s) {
    DM_CHECK_ARG(z && psum && psq && M > 0 && C > 0, "dm_col_stats: bad arguments");
    const int rpb = cs_rows(M), grid = cdiv(M, rpb);
    DM_DISPATCH_DTYPE(dtype, {
        if (vec_ok<T>(C, z)) hipLaunchKernelGGL((col_stats_kernel<T, Elem<T>::VE>), dim3(grid), dim3(256), 0, (hipStream_t)s, (const T*)z, M, C, rpb, psum, psq);
        else hipLaunchKernelGGL((col_stats_kernel<T, 1>), dim3(grid), dim3(256), 0, (hipStream_t)s, (const T*)z, M, C, rpb, psum, psq);
    });
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_bn_finalize(const float* psum, const float* psq, int nblk, int M, int C, float eps, float momentum,
                              float* mean, float* rstd, float* running_mean, float* running_var, dm_stream_t s) {
    DM_CHECK_ARG(psum && psq && mean && rstd && nblk > 0 && M > 0 && C > 0, "dm_bn_finalize: bad arguments");
    DM_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "dm_bn_finalize: running_mean/var must come together");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 16)), dim3(1024), 0, (hipStream_t)s, psum, psq, nblk, M, C, eps, momentum,
                       mean, rstd, running_mean, running_var);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_col_reduce(const float* part, int nblk, int C, float* out, int accumulate, dm_stream_t s) {
    DM_CHECK_ARG(part && out && nblk > 0 && C > 0, "dm_col_reduce: bad arguments");
    hipLaunchKernelGGL(col_reduce_kernel, dim3(cdiv(C, 16)), dim3(1024), 0, (hipStream_t)s, part, nblk, C, out, accumulate);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_col_reduce2(const float* part1, const float* part2, int nblk, int C, float* out1, float* out2, dm_stream_t s) {
    DM_CHECK_ARG(part1 && part2 && out1 && out2 && nblk > 0 && C > 0, "dm_col_reduce2: bad arguments");
    hipLaunchKernelGGL(col_reduce2_kernel, dim3(cdiv(2 * C, 16)), dim3(1024), 0, (hipStream_t)s, part1, part2, nblk, C, out1, out2);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_bn_act_fwd(const void* z, void* y, int dtype, int M, int C, const float* mean, const float* rstd,
                             const float* gamma, const float* beta, int act, dm_stream_t s) {
    DM_CHECK_ARG(z && y && mean && rstd && M > 0 && C > 0, "dm_bn_act_fwd: bad arguments");
    DM_DISPATCH_DTYPE(dtype, {
        if (vec_ok<T>(C, z, y)) {
            constexpr int V = Elem<T>::VE;
            const int64_t nvec = (int64_t)M * C / V;
            hipLaunchKernelGGL((bn_act_fwd_kernel<T, V>), dim3(stream_grid(nvec, C / V)), dim3(256), 0, (hipStream_t)s, (const T*)z, (T*)y, nvec, C / V, mean, rstd, gamma, beta, act);
        } else {
            const int64_t nvec = (int64_t)M * C;
            hipLaunchKernelGGL((bn_act_fwd_kernel<T, 1>), dim3(stream_grid(nvec, C)), dim3(256), 0, (hipStream_t)s, (const T*)z, (T*)y, nvec, C, mean, rstd, gamma, beta, act);
        }
    });
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_bn_act_bwd_reduce(const void* z, const void* dy, int dtype, int M, int C, const float* mean, const float* rstd,
                                    const float* gamma, const float* beta, int act, float* p1, float* p2, dm_stream_t s) {
    DM_CHECK_ARG(z && dy && mean && rstd && p1 && p2 && M > 0 && C > 0, "dm_bn_act_bwd_reduce: bad arguments");
    const int rpb = cs_rows(M), grid = cdiv(M, rpb);
    DM_DISPATCH_DTYPE(dtype, {
        if (vec_ok<T>(C, z, dy)) {
            if constexpr (sizeof(T) == 2) {
                if (act == DM_ACT_GELU && g_bn_packed) hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, 8, true>), dim3(grid), dim3(256), 0, (hipStream_t)s, (const T*)z, (const T*)dy, M, C, mean, rstd, gamma, beta, act, rpb, p1, p2, 0);
                else hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, 8, false>), dim3(grid), dim3(256), 0, (hipStream_t)s, (const T*)z, (const T*)dy, M, C, mean, rstd, gamma, beta, act, rpb, p1, p2, 0);
            } else {
                hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, Elem<T>::VE, false>), dim3(grid), dim3(256), 0, (hipStream_t)s, (const T*)z, (const T*)dy, M, C, mean, rstd, gamma, beta, act, rpb, p1, p2, 0);
            }
        } else {
            hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, 1, false>), dim3(grid), dim3(256), 0, (hipStream_t)s, (const T*)z, (const T*)dy, M, C, mean, rstd, gamma, beta, act, rpb, p1, p2, 0);
        }
    });
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_bn_act_bwd_apply(const void* z, const void* dy, void* dz, int dtype, int M, int C, const float* mean,
                                   const float* rstd, const float* gamma, const float* beta, int act, const float* s1,
                                   const float* s2, dm_stream_t s) {
    DM_CHECK_ARG(z && dy && dz && mean && rstd && s1 && s2 && M > 0 && C > 0, "dm_bn_act_bwd_apply: bad arguments");
    const float invM = 1.0f / (float)M;
    DM_DISPATCH_DTYPE(dtype, {
        if (vec_ok<T>(C, z, dy, dz)) {
            constexpr int V = Elem<T>::VE;
            const int64_t nvec = (int64_t)M * C / V;
            hipLaunchKernelGGL((bn_bwd_apply_kernel<T, V>), dim3(stream_grid(nvec, C / V)), dim3(256), 0, (hipStream_t)s, (const T*)z, (const T*)dy, (T*)dz, nvec, C / V, invM, mean, rstd, gamma, beta, act, s1, s2);
        } else {
            const int64_t nvec = (int64_t)M * C;
            hipLaunchKernelGGL((bn_bwd_apply_kernel<T, 1>), dim3(stream_grid(nvec, C)), dim3(256), 0, (hipStream_t)s, (const T*)z, (const T*)dy, (T*)dz, nvec, C, invM, mean, rstd, gamma, beta, act, s1, s2);
        }
    });
    DM_LAUNCH_CHECK();
    return DM_OK;
}

/* ---- slot-folded statistics: no finalize / column-reduce launches between the producing and the consuming kernel ---- */
extern "C" int dm_bn_act_fwd_slots(const void* z, void* y, int dtype, int M, int C, const void* psum, const void* psq, int slots, float eps,
                                   float momentum, const float* gamma, const float* beta, int act, float* mean, float* rstd,
                                   float* running_mean, float* running_var, dm_stream_t s) {
    DM_CHECK_ARG(z && y && psum && psq && mean && rstd && M > 0 && C > 0 && slots > 0 && slots <= 64, "dm_bn_act_fwd_slots: bad arguments");
    DM_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "dm_bn_act_fwd_slots: running_mean/var must come together");
    DM_CHECK_ARG(C <= 8192, "dm_bn_act_fwd_slots: C=%d > 8192 (the statistics of all channels sit in LDS)", C);
    DM_DISPATCH_DTYPE(dtype, {
        DM_CHECK_ARG(vec_ok<T>(C, z, y), "dm_bn_act_fwd_slots: C=%d must be a multiple of %d and the tensors 16-byte aligned", C, Elem<T>::VE);
        constexpr int V = Elem<T>::VE;
        const int64_t nvec = (int64_t)M * C / V;
        hipLaunchKernelGGL((bn_act_fwd_slots_kernel<T, V>), dim3(cdiv(stream_grid(nvec, C / V) * 256, SLOT_THREADS)), dim3(SLOT_THREADS), 2 * C * sizeof(float), (hipStream_t)s, (const T*)z, (T*)y,
                           nvec, C / V, (const double*)psum, (const double*)psq, slots, M, eps, momentum, gamma, beta, act, mean, rstd, running_mean, running_var);
    });
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_bn_act_bwd_reduce_slots(const void* z, const void* dy, int dtype, int M, int C, const float* mean, const float* rstd,
                                          const float* gamma, const float* beta, int act, void* p1v, void* p2v, int slots, dm_stream_t s) {
    float* p1 = (float*)p1v;
    float* p2 = (float*)p2v;
    DM_CHECK_ARG(z && dy && mean && rstd && p1 && p2 && M > 0 && C > 0 && slots > 0 && slots <= 64 && (((uintptr_t)p1 | (uintptr_t)p2) & 7) == 0,
                 "dm_bn_act_bwd_reduce_slots: bad arguments");
    const int rpb = cs_rows(M), grid = cdiv(M, rpb);
    DM_DISPATCH_DTYPE(dtype, {
        if (vec_ok<T>(C, z, dy)) {
            if constexpr (sizeof(T) == 2) {
                if (act == DM_ACT_GELU && g_bn_packed) hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, 8, true>), dim3(grid), dim3(256), 0, (hipStream_t)s, (const T*)z, (const T*)dy, M, C, mean, rstd, gamma, beta, act, rpb, p1, p2, slots);
                else hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, 8, false>), dim3(grid), dim3(256), 0, (hipStream_t)s, (const T*)z, (const T*)dy, M, C, mean, rstd, gamma, beta, act, rpb, p1, p2, slots);
            } else {
                hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, Elem<T>::VE, false>), dim3(grid), dim3(256), 0, (hipStream_t)s, (const T*)z, (const T*)dy, M, C, mean, rstd, gamma, beta, act, rpb, p1, p2, slots);
            }
        } else {
            hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, 1, false>), dim3(grid), dim3(256), 0, (hipStream_t)s, (const T*)z, (const T*)dy, M, C, mean, rstd, gamma, beta, act, rpb, p1, p2, slots);
        }
    });
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_bn_act_bwd_apply_slots(const void* z, const void* dy, void* dz, int dtype, int M, int C, const float* mean, const float* rstd,
                                         const float* gamma, const float* beta, int act, const void* p1, const void* p2, int slots,
                                         float* dbeta, float* dgamma, dm_stream_t s) {
    DM_CHECK_ARG(z && dy && dz && mean && rstd && p1 && p2 && dbeta && dgamma && M > 0 && C > 0 && slots > 0 && slots <= 64,
                 "dm_bn_act_bwd_apply_slots: bad arguments");
    DM_CHECK_ARG(C <= 8192, "dm_bn_act_bwd_apply_slots: C=%d > 8192", C);
    const float invM = 1.0f / (float)M;
    DM_DISPATCH_DTYPE(dtype, {
        DM_CHECK_ARG(vec_ok<T>(C, z, dy, dz), "dm_bn_act_bwd_apply_slots: C=%d must be a multiple of %d and the tensors 16-byte aligned", C, Elem<T>::VE);
        constexpr int V = Elem<T>::VE;
        const int64_t nvec = (int64_t)M * C / V;
        hipLaunchKernelGGL((bn_bwd_apply_slots_kernel<T, V>), dim3(cdiv(stream_grid(nvec, C / V) * 256, SLOT_THREADS)), dim3(SLOT_THREADS), 2 * C * sizeof(float), (hipStream_t)s, (const T*)z,
                           (const T*)dy, (T*)dz, nvec, C / V, invM, mean, rstd, gamma, beta, act, (const double*)p1, (const double*)p2, slots, dbeta, dgamma);
    });
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_bn_fold(const float* gamma, const float* beta, const float* rmean, const float* rvar, const float* conv_bias,
                          float eps, int C, float* scale, float* shift, dm_stream_t s) {
    DM_CHECK_ARG(gamma && beta && rmean && rvar && scale && shift && C > 0, "dm_bn_fold: bad arguments");
    hipLaunchKernelGGL(bn_fold_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)s, gamma, beta, rmean, rvar, conv_bias, eps, C, scale, shift);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_gn_act_fwd(const void* x, void* y, int dtype, int B, int HW, int C, int G, float eps, const float* gamma,
                             const float* beta, int act, float* mean, float* rstd, dm_stream_t s) {
    DM_CHECK_ARG(x && y && gamma && beta && mean && rstd && B > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0, "dm_gn_act_fwd: bad arguments");
    DM_DISPATCH_DTYPE(dtype, {
        constexpr int V = Elem<T>::VE;
        const int cg = C / G;
        int pps = 0;
        const int S = gn_slabs(B, HW, C, G, V, x, y, nullptr, &pps);
        if (S > 0) {
            hipLaunchKernelGGL((gn_slab_stats_kernel<T, V, false>), dim3(S, B), dim3(256), 0, (hipStream_t)s, (const T*)x, (const T*)nullptr, HW, C, G, pps, gamma, beta, act, (const float*)nullptr, (const float*)nullptr, dm_g_ws, (float*)nullptr, (float*)nullptr);
            hipLaunchKernelGGL((gn_slab_apply_kernel<T, V, false>), dim3(S, B), dim3(256), 0, (hipStream_t)s, (const T*)x, (const T*)nullptr, (T*)y, HW, C, G, pps, S, eps, gamma, beta, act, (const float*)dm_g_ws, mean, rstd);
        } else if (cg % V == 0 && 256 % (cg / V) == 0 && C % V == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0)
            hipLaunchKernelGGL((gn_fwd_vec_kernel<T, V>), dim3(B * G), dim3(256), 0, (hipStream_t)s, (const T*)x, (T*)y, HW, C, G, eps, gamma, beta, act, mean, rstd);
        else
            hipLaunchKernelGGL((gn_fwd_kernel<T>), dim3(B * G), dim3(256), 0, (hipStream_t)s, (const T*)x, (T*)y, HW, C, G, eps, gamma, beta, act, mean, rstd);
    });
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_gn_act_bwd(const void* x, const void* dy, void* dx, int dtype, int B, int HW, int C, int G, const float* gamma,
                             const float* beta, int act, const float* mean, const float* rstd, float* dgamma, float* dbeta,
                             dm_stream_t s) {
    DM_CHECK_ARG(x && dy && dx && gamma && beta && mean && rstd && dgamma && dbeta && B > 0 && HW > 0 && C % G == 0, "dm_gn_act_bwd: bad arguments");
    const size_t shm = 2 * (C / G) * sizeof(float);
    DM_DISPATCH_DTYPE(dtype, {
        constexpr int V = Elem<T>::VE;
        const int cg = C / G;
        int pps = 0;
        const int S = gn_slabs(B, HW, C, G, V, x, dy, dx, &pps);
        if (S > 0) {
            float* pdb = dm_g_ws + (((size_t)B * S * 2 * G + 63) & ~(size_t)63);
            float* pdg = pdb + (size_t)B * S * C;
            hipLaunchKernelGGL((gn_slab_stats_kernel<T, V, true>), dim3(S, B), dim3(256), 256 * 2 * V * sizeof(float), (hipStream_t)s, (const T*)x, (const T*)dy, HW, C, G, pps, gamma, beta, act, mean, rstd, dm_g_ws, pdb, pdg);
            hipLaunchKernelGGL(col_reduce_kernel, dim3(cdiv(C, 16)), dim3(1024), 0, (hipStream_t)s, (const float*)pdb, B * S, C, dbeta, 1);
            hipLaunchKernelGGL(col_reduce_kernel, dim3(cdiv(C, 16)), dim3(1024), 0, (hipStream_t)s, (const float*)pdg, B * S, C, dgamma, 1);
            hipLaunchKernelGGL((gn_slab_apply_kernel<T, V, true>), dim3(S, B), dim3(256), 0, (hipStream_t)s, (const T*)x, (const T*)dy, (T*)dx, HW, C, G, pps, S, 0.f, gamma, beta, act, (const float*)dm_g_ws, (float*)mean, (float*)rstd);
        } else if (cg % V == 0 && 256 % (cg / V) == 0 && C % V == 0 && (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0)
            hipLaunchKernelGGL((gn_bwd_vec_kernel<T, V>), dim3(B * G), dim3(256), shm, (hipStream_t)s, (const T*)x, (const T*)dy, (T*)dx, HW, C, G, gamma, beta, act, mean, rstd, dgamma, dbeta);
        else
            hipLaunchKernelGGL((gn_bwd_kernel<T>), dim3(B * G), dim3(256), shm, (hipStream_t)s, (const T*)x, (const T*)dy, (T*)dx, HW, C, G, gamma, beta, act, mean, rstd, dgamma, dbeta);
    });
    DM_LAUNCH_CHECK();
    return DM_OK;
}
