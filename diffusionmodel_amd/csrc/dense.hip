// Dense fp32 layers on small matrices (new_scripy.py:148-157 SE MLP, :104-127 CoordAttn 1x1 convolutions on strips,
// :255-268 EmbedFC): y = x W^T + b and its two gradients, M = 64 ... 2048 rows, K, N = 4 ... 1024.
//
// These are ~120 launches per train step of a few MFLOP each: nothing to do with a roofline, everything with latency.  The
// implicit-GEMM convolution kernels (igemm.hip) ran them at 8-11 us apiece (a 32-deep serial k-loop of DMA -> barrier -> MFMA
// steps per workgroup, a split-K epilogue launch and a transposed weight pack on top); here a launch is one wave-level pass:
//   * operands go global -> registers directly (16-byte loads along the contiguous dimension where there is one, 64-byte
//     segments otherwise), no LDS staging and no barrier inside the reduction loop, loads of four 16-deep chunks in flight;
//   * the four waves of a workgroup split the REDUCTION dimension and fold their 64 x 16 partial tiles through LDS once;
//   * v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulation) with the weights as the A operand, so a lane ends up
//     with four consecutive output columns of one row: 16-byte stores;
//   * the input gradient reads W in its natural [N][K] layout (no transposed pack), the weight gradient splits the rows over
//     workgroups and accumulates with fp32 atomics (dw is an accumulator by contract), the bias gradient rides along.
// Element j of lane group g is reduction index 4 g + j on BOTH operands, so the four MFMAs of a 16-deep chunk cover it once.
#include "common.h"

namespace {

__device__ __forceinline__ f32x4 ld4(const float* p, bool ok) { return ok ? *(const f32x4*)p : (f32x4){0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ float ld1(const float* p, bool ok) { return ok ? *p : 0.f; }

// sum the four waves' partial tiles acc[4 frags] through LDS; wave w returns the total of fragment w
__device__ __forceinline__ f32x4 fold_waves(f32x4 (&acc)[4], f32x4* red, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) red[(wave * 4 + i) * 64 + lane] = acc[i];
    __syncthreads();
    f32x4 s = red[(0 * 4 + wave) * 64 + lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) s += red[(w * 4 + wave) * 64 + lane];
    return s;
}

// ---- y[M][N] = act(x[M][K] w[N][K]^T + b) -----------------------------------------------------------------------------
// tile: 16 FR rows (FR fragments) x 16 columns; grid (N/16, M/(16 FR)).  FR = 4: 64-row tiles, every wave finishes one fragment.
// FR = 1 (r04): 16-row tiles for the launches whose 64-row grid leaves three quarters of the chip idle — the second layer of an
// EmbedFC at B = 64 is (N/16, 1) = 64 workgroups walking a 1024-deep reduction in four dependent load batches (18 us for 134 MFLOP);
// with 16-row tiles it is 256 workgroups and two batches of 16 loads (the weights are re-read from L2 by four row tiles).
template <int FR>
__global__ __launch_bounds__(256) void dense_nt_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                       float* __restrict__ y, int M, int K, int N, int act) {
    __shared__ f32x4 red[16 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16 * FR;
    const int nck = (K + 15) >> 4;
    constexpr int U = FR == 4 ? 4 : 8;                      // chunks per batch: 4 x (1 + 4) = 20 or 8 x (1 + 1) = 16 loads in flight
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* wp = w + (size_t)(n0 + r) * K + 4 * g;
    const bool wok = n0 + r < N;
    const float* xp[FR];
    bool xok[FR];
#pragma unroll
    for (int i = 0; i < FR; ++i) {
        xok[i] = m0 + i * 16 + r < M;
        xp[i] = x + (size_t)(m0 + i * 16 + r) * K + 4 * g;
    }
    for (int c0 = wave; c0 < nck; c0 += 4 * U) {            // this wave's chunks c0, c0 + 4, ...
        f32x4 fw[U], fx[U][FR];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kb = (c0 + 4 * u) * 16;
            const bool kok = kb + 4 * g < K;
            fw[u] = ld4(wp + kb, wok && kok);
#pragma unroll
            for (int i = 0; i < FR; ++i) fx[u][i] = ld4(xp[i] + kb, xok[i] && kok);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < FR; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[u][j], fx[u][i][j], acc[i], 0, 0, 0);
    }
    f32x4 s;
    if constexpr (FR == 4) {
        s = fold_waves(acc, red, wave, lane);
    } else {                                                 // one fragment: the four waves' reduction parts meet in wave 0
        red[wave * 64 + lane] = acc[0];
        __syncthreads();
        if (wave != 0) return;
        s = red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane];
    }
    const int m = m0 + wave * 16 + r, n = n0 + 4 * g;
    if (m < M && n < N) {                                    // N % 4 == 0: the four columns are in range together
        f32x4 o;
#pragma unroll
        for (int v = 0; v < 4; ++v) o[v] = act_apply(s[v] + (b ? b[n + v] : 0.f), act);
        *(f32x4*)(y + (size_t)m * N + n) = o;
    }
}

// ---- dx[M][K] = g[M][N] w[N][K] -----------------------------------------------------------------------------------------
// tile: 16 FR rows x 16 columns (k); the waves split the reduction over n; grid (K/16, M/(16 FR)); FR as in dense_nt_kernel
template <int FR>
__global__ __launch_bounds__(256) void dense_nn_kernel(const float* __restrict__ gy, const float* __restrict__ w, float* __restrict__ dx, int M,
                                                       int K, int N) {
    __shared__ f32x4 red[16 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const int k0 = blockIdx.x * 16, m0 = blockIdx.y * 16 * FR;
    const int nck = (N + 15) >> 4;
    constexpr int U = FR == 4 ? 2 : 4;                      // chunks per batch: 2 x (4 + 4) = 16 or 4 x (4 + 1) = 20 loads in flight
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool wok = k0 + r < K;
    const float* gp[FR];
    bool gok[FR];
#pragma unroll
    for (int i = 0; i < FR; ++i) {
        gok[i] = m0 + i * 16 + r < M;
        gp[i] = gy + (size_t)(m0 + i * 16 + r) * N + 4 * g;
    }
    for (int c0 = wave; c0 < nck; c0 += 4 * U) {             // chunks c0, c0 + 4, ...
        float fw[U][4];
        f32x4 fg[U][FR];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int nb = (c0 + 4 * u) * 16 + 4 * g;
            const bool nok = nb < N;                         // N % 4 == 0
#pragma unroll
            for (int j = 0; j < 4; ++j) fw[u][j] = ld1(w + (size_t)(nb + j) * K + k0 + r, wok && nok);
#pragma unroll
            for (int i = 0; i < FR; ++i) fg[u][i] = ld4(gp[i] + (c0 + 4 * u) * 16, gok[i] && nok);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < FR; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[u][j], fg[u][i][j], acc[i], 0, 0, 0);
    }
    f32x4 s;
    if constexpr (FR == 4) {
        s = fold_waves(acc, red, wave, lane);
    } else {
        red[wave * 64 + lane] = acc[0];
        __syncthreads();
        if (wave != 0) return;
        s = red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane];
    }
    const int m = m0 + wave * 16 + r, k = k0 + 4 * g;
    if (m < M && k < K) *(f32x4*)(dx + (size_t)m * K + k) = s;       // K % 4 == 0
}

// ---- dw[N][K] += g[M][N]^T x[M][K];  db[N] += column sums of g ---------------------------------------------------------------
// tile: 16 rows (n) x 64 columns (k, 4 fragments); the reduction over the M rows is split over the waves and over grid.z;
// grid (K/64, N/16, splits)
__global__ __launch_bounds__(256) void dense_tn_kernel(const float* __restrict__ gy, const float* __restrict__ x, float* __restrict__ dw,
                                                       float* __restrict__ db, int M, int K, int N, int rows_per_split) {
    __shared__ f32x4 red[16 * 64];
    __shared__ float redb[4 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const int k0 = blockIdx.x * 64, n0 = blockIdx.y * 16;
    const int mlo = blockIdx.z * rows_per_split, mhi = min(M, mlo + rows_per_split);
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    const bool nok = n0 + r < N;
    bool kok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) kok[i] = k0 + i * 16 + r < K;
    for (int mb = mlo + wave * 16; mb < mhi; mb += 64) {     // one 16-row chunk per wave per pass: 4 + 16 loads in flight
        float fg[4], fx[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = mb + 4 * g + j;
            const bool mok = m < mhi;
            fg[j] = ld1(gy + (size_t)m * N + n0 + r, mok && nok);
#pragma unroll
            for (int i = 0; i < 4; ++i) fx[i][j] = ld1(x + (size_t)m * K + k0 + i * 16 + r, mok && kok[i]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bsum += fg[j];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fx[i][j], fg[j], acc[i], 0, 0, 0);
        }
    }
    // D[i = k][j = n]: the lane holds k = k0 + 16 frag + 4 g .. + 3 of row n = n0 + r
    const f32x4 s = fold_waves(acc, red, wave, lane);
    // r04: the totals go through LDS once more so that a wave instruction of atomics covers 64 CONSECUTIVE floats of one dw row (the
    // register layout has a wave's lanes 16 rows apart: fp32 atomics run at ~65 G/s that way and at ~400 G/s on whole 64-byte runs —
    // wgrad_pw_kernel's measurement — which was 16 of the 30 us of the 1024 x 1024 EmbedFC layers)
    __syncthreads();                                         // (every wave has read its fold operands)
    float* tile = (float*)red;                               // [16 n][64 k]
    *(f32x4*)(tile + r * 64 + wave * 16 + 4 * g) = s;
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int nn = wave * 4 + v;
        if (n0 + nn < N && k0 + lane < K) unsafeAtomicAdd(dw + (size_t)(n0 + nn) * K + k0 + lane, tile[nn * 64 + lane]);
    }
    if (db != nullptr && blockIdx.x == 0) {                  // column sums of g: lanes r, r + 16, r + 32, r + 48 of every wave hold parts of n0 + r
        redb[wave * 64 + lane] = bsum;
        __syncthreads();
        if (threadIdx.x < 16 && n0 + threadIdx.x < N) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += redb[(q >> 2) * 64 + (q & 3) * 16 + threadIdx.x];
            unsafeAtomicAdd(db + n0 + threadIdx.x, t);
        }
    }
}

}  // namespace

// used by dm_linear_fwd / dm_linear_bwd (attn.hip) when K % 4 == 0 and N % 4 == 0 and the tensors are 16-byte aligned
int dm_dense_fwd(const float* x, const float* w, const float* b, float* y, int M, int K, int N, int act, hipStream_t st) {
    // few 64-row tiles and a deep reduction: 16-row tiles fill the chip (see dense_nt_kernel)
    if (cdiv(N, 16) * cdiv(M, 64) <= 96 && M > 16 && K >= 256)
        hipLaunchKernelGGL(dense_nt_kernel<1>, dim3(cdiv(N, 16), cdiv(M, 16)), dim3(256), 0, st, x, w, b, y, M, K, N, act);
    else
        hipLaunchKernelGGL(dense_nt_kernel<4>, dim3(cdiv(N, 16), cdiv(M, 64)), dim3(256), 0, st, x, w, b, y, M, K, N, act);
    return DM_OK;
}

int dm_dense_bwd(const float* x, const float* w, const float* gy, float* dx, float* dw, float* db, int M, int K, int N, hipStream_t st) {
    if (dx) {
        if (cdiv(K, 16) * cdiv(M, 64) <= 96 && M > 16 && N >= 256)
            hipLaunchKernelGGL(dense_nn_kernel<1>, dim3(cdiv(K, 16), cdiv(M, 16)), dim3(256), 0, st, gy, w, dx, M, K, N);
        else
            hipLaunchKernelGGL(dense_nn_kernel<4>, dim3(cdiv(K, 16), cdiv(M, 64)), dim3(256), 0, st, gy, w, dx, M, K, N);
    }
    if (dw) {
        // rows per workgroup: at most 4 passes of 64 rows, fewer workgroups than ~1024 in total
        int splits = cdiv(M, 256);
        const int tiles = cdiv(K, 64) * cdiv(N, 16);
        while (splits > 1 && (int64_t)splits * tiles > 2048) splits = (splits + 1) / 2;
        const int rps = cdiv(cdiv(M, splits), 16) * 16;
        hipLaunchKernelGGL(dense_tn_kernel, dim3(cdiv(K, 64), cdiv(N, 16), cdiv(M, rps)), dim3(256), 0, st, gy, x, dw, db, M, K, N, rps);
    }
    return DM_OK;
}
