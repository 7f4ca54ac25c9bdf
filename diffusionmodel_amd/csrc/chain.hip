// Fused small-matrix chains of the attention blocks: the SE excitation MLP (new_scripy.py:143-158) and the strip chain of
// CoordAttn between its pooling pass and its gate pass (new_scripy.py:105-129), forward and backward.
//
// r02 ran these as one launch per torch.nn op: 10 launches per SE block, 43 per CoordAttn block, each a few MFLOP on fp32
// matrices of <= 2048 x 128 — ~300 dependent launches per train step at 2.6 us of device wall time apiece, 1.9 ms of
// profiler-reported kernel time for < 0.1 % of the FLOPs.  Here a chain is 1-2 launches in each direction: a workgroup owns 16
// rows (SE: samples; CoordAttn: strip positions of ONE sample, both strips) and walks through the chain with the intermediate
// matrices in LDS; only the reductions over ALL rows (BatchNorm batch statistics, weight gradients) end a launch.
//
// Arithmetic: v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulation) like dense.hip; the weights are the A operand, so a
// lane (r = lane & 15, g = lane >> 4) ends up with out[row r][4 g .. 4 g + 3] of a 16 x 16 fragment.  A "chunk" is 16
// consecutive k; element j of lane group g is k = kb + 4 g + j on both operands.
#include "common.h"

namespace {

typedef f32x4 F4;
__device__ __forceinline__ F4 z4() { return (F4){0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ F4 ld4c(const float* p, bool ok) { return ok ? *(const F4*)p : z4(); }
#define DM_MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// weights of output column n for k = kb4 .. kb4 + 3.  WT = false: W[n][k] (y = x W^T); WT = true: W[k][n] (y = x W)
template <bool WT>
__device__ __forceinline__ F4 wfrag(const float* __restrict__ W, int ldw, int n, int kb4, int N, int K) {
    if (n >= N || kb4 >= K) return z4();                 // K % 4 == 0
    if constexpr (!WT) return *(const F4*)(W + (size_t)n * ldw + kb4);
    else return (F4){W[(size_t)kb4 * ldw + n], W[(size_t)(kb4 + 1) * ldw + n], W[(size_t)(kb4 + 2) * ldw + n], W[(size_t)(kb4 + 3) * ldw + n]};
}

// sum the four waves' partial tiles through LDS (`red`: 16 * 64 F4); wave w returns the total of fragment w
__device__ __forceinline__ F4 fold4(F4 (&acc)[4], F4* red, int wave, int lane) {
    __syncthreads();                                     // `red` may still be read by a slower wave of the previous fold
#pragma unroll
    for (int i = 0; i < 4; ++i) red[(wave * 4 + i) * 64 + lane] = acc[i];
    __syncthreads();
    F4 s = red[(0 * 4 + wave) * 64 + lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) s += red[(w * 4 + wave) * 64 + lane];
    return s;
}

// out[16][64 columns n_lo ..] = X[16][K] W^T, long K: the waves split the k-chunks.  xf(kb4) returns this lane's four x values
// (row r, k = kb4 .. kb4 + 3; zeros outside).  Wave w returns columns n_lo + 16 w + 4 g .. + 3 of row r.
template <bool WT, typename XF>
__device__ __forceinline__ F4 mm_bigK(XF xf, const float* __restrict__ W, int ldw, int n_lo, int N, int K, F4* red, int wave, int lane) {
    const int r = lane & 15, g = lane >> 4;
    F4 acc[4] = {z4(), z4(), z4(), z4()};
    const int nck = (K + 15) >> 4;
    // chunks c0, c0 + 4 per pass: 10 loads in flight (four chunks per pass — 20 loads — measured SLOWER on every kernel that
    // uses this: 47 vs 29 us on ca_bwd_mix, 17 vs 9 on ca_z; the W^T form holds 64 scalar loads and their addresses then)
    for (int c0 = wave; c0 < nck; c0 += 8) {
        F4 fx[2], fw[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int kb4 = (c0 + 4 * u) * 16 + 4 * g;
            fx[u] = kb4 < K ? xf(kb4) : z4();
#pragma unroll
            for (int i = 0; i < 4; ++i) fw[u][i] = wfrag<WT>(W, ldw, n_lo + 16 * i + r, kb4, N, K);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = DM_MFMA4(fw[u][i][j], fx[u][j], acc[i]);
    }
    return fold4(acc, red, wave, lane);
}

// out[16][N] = X[16][K] W^T, short K (<= 128), X in LDS (row stride ldx floats): wave w takes the 16-column fragments
// f = w, w + 4, ...; epi(n4, acc): acc[v] = out[r][n4 + v], n4 = 16 f + 4 g
template <bool WT, typename EPI>
__device__ __forceinline__ void mm_bigN(const float* xl, int ldx, const float* __restrict__ W, int ldw, int N, int K, int wave, int lane, EPI epi,
                                        int part = 0, int nparts = 1) {
    // `part` of `nparts`: the fragments are dealt out over workgroups too (f = 4 part + wave, step 4 nparts)
    const int r = lane & 15, g = lane >> 4;
    const int nf = (N + 15) >> 4, nck = (K + 15) >> 4;
    const int fstep = 4 * nparts;
    for (int f0 = 4 * part + wave; f0 < nf; f0 += 2 * fstep) {     // two fragments per pass: their weight loads are in flight together
        F4 acc[2] = {z4(), z4()};
        const bool two = f0 + fstep < nf;
        for (int c0 = 0; c0 < nck; c0 += 4) {
            F4 fw[2][4], fx[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kb4 = (c0 + u) * 16 + 4 * g;
                fw[0][u] = wfrag<WT>(W, ldw, 16 * f0 + r, kb4, N, K);
                fw[1][u] = two ? wfrag<WT>(W, ldw, 16 * (f0 + fstep) + r, kb4, N, K) : z4();
                fx[u] = kb4 < K ? *(const F4*)(xl + r * ldx + kb4) : z4();
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[0] = DM_MFMA4(fw[0][u][j], fx[u][j], acc[0]);
                    acc[1] = DM_MFMA4(fw[1][u][j], fx[u][j], acc[1]);
                }
        }
        epi(16 * f0 + 4 * g, acc[0]);
        if (two) epi(16 * (f0 + fstep) + 4 * g, acc[1]);
    }
}

// dw[n0 .. +16][k0 .. +64] += g[mlo .. mhi][n]^T x[m][k] (fp32 atomics: dw is an accumulator); db[n] += column sums of g when db
// is given and `with_db`.  gf(m, n) / xf(m, k) return the operand values (0 outside).  The waves split the rows.
template <typename GF, typename XF>
__device__ __forceinline__ void tn_tile(GF gf, XF xf, float* __restrict__ dw, float* __restrict__ db, bool with_db, int mlo, int mhi, int K, int N,
                                        int k0, int n0, F4* red, float* redb, int wave, int lane) {
    const int r = lane & 15, g = lane >> 4;
    F4 acc[4] = {z4(), z4(), z4(), z4()};
    float bsum = 0.f;
    for (int mb = mlo + wave * 16; mb < mhi; mb += 64) {
        float fg[4], fx[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = mb + 4 * g + j;
            const bool mok = m < mhi;
            fg[j] = (mok && n0 + r < N) ? gf(m, n0 + r) : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) fx[i][j] = (mok && k0 + i * 16 + r < K) ? xf(m, k0 + i * 16 + r) : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bsum += fg[j];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = DM_MFMA4(fx[i][j], fg[j], acc[i]);
        }
    }
    const F4 s = fold4(acc, red, wave, lane);             // the lane holds k = k0 + 16 wave + 4 g .. + 3 of row n = n0 + r
    const int n = n0 + r, k = k0 + wave * 16 + 4 * g;
    if (n < N && k < K) {
#pragma unroll
        for (int v = 0; v < 4; ++v) unsafeAtomicAdd(dw + (size_t)n * K + k + v, s[v]);
    }
    if (db != nullptr && with_db) {
        redb[wave * 64 + lane] = bsum;
        __syncthreads();
        if (threadIdx.x < 16 && n0 + threadIdx.x < N) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += redb[(q >> 2) * 64 + (q & 3) * 16 + threadIdx.x];
            unsafeAtomicAdd(db + n0 + threadIdx.x, t);
        }
        __syncthreads();
    }
}

// =================================================================================================================================
// SE excitation MLP: sg = sigmoid(W2 gelu(W1 y)), y = mean_hw(x2)  (new_scripy.py:148-157; no biases)
// =================================================================================================================================
struct SeP {
    const float* parts; int RS; float scale;             // pooled input: y[m][c] = scale * sum_rs parts[rs][m][c]  (RS = 1: parts is the sum)
    const float* w1; const float* w2;                    // [R][C], [C][R]
    float *y, *hid, *gh, *sg;                            // outputs [B][C], [B][R], [B][R], [B][C]
    int B, C, R, save;                                   // save = 0 (inference): only sg is written
};

constexpr int SE_LD = 132;                               // LDS row stride of the [16][R <= 128] block

__global__ __launch_bounds__(256) void se_fwd_kernel(const SeP p) {
    __shared__ F4 red[16 * 64];
    __shared__ __attribute__((aligned(16))) float ghl[16 * SE_LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * 16, m = m0 + r;
    const bool mok = m < p.B;
    const size_t bc = (size_t)p.B * p.C;
    for (int n_lo = 0; n_lo < p.R; n_lo += 64) {
        auto xf = [&](int kb4) -> F4 {                   // fold of the pooling partials; every (row, k) passes through exactly one lane
            if (!mok) return z4();
            F4 s = *(const F4*)(p.parts + (size_t)m * p.C + kb4);
            for (int q = 1; q < p.RS; ++q) s += *(const F4*)(p.parts + q * bc + (size_t)m * p.C + kb4);
            s *= p.scale;
            if (p.save && n_lo == 0 && blockIdx.y == 0) *(F4*)(p.y + (size_t)m * p.C + kb4) = s;
            return s;
        };
        const F4 h = mm_bigK<false>(xf, p.w1, p.C, n_lo, p.R, p.C, red, wave, lane);
        const int n4 = n_lo + 16 * wave + 4 * g;
        if (n4 < p.R) {
            F4 gv;
#pragma unroll
            for (int v = 0; v < 4; ++v) gv[v] = gelu_f(h[v]);
            *(F4*)(ghl + r * SE_LD + n4) = gv;
            if (p.save && mok && blockIdx.y == 0) {
                *(F4*)(p.hid + (size_t)m * p.R + n4) = h;
                *(F4*)(p.gh + (size_t)m * p.R + n4) = gv;
            }
        }
    }
    __syncthreads();
    mm_bigN<false>(ghl, SE_LD, p.w2, p.R, p.C, p.R, wave, lane, [&](int n4, F4 acc) {
        if (mok && n4 < p.C) {
            F4 s;
#pragma unroll
            for (int v = 0; v < 4; ++v) s[v] = sigmoid_f(acc[v]);
            *(F4*)(p.sg + (size_t)m * p.C + n4) = s;
        }
    }, blockIdx.y, gridDim.y);
}

struct SeBP {
    const float* parts; int RS; float scale;             // dsg[m][c] = scale * sum_rs parts[rs][m][c]
    const float *sg, *hid, *gh, *y, *w1, *w2;
    float *dlogit, *dhid, *dy, *dw1, *dw2;               // scratch [B][C], [B][R]; outputs dy [B][C]; accumulators dw1 [R][C], dw2 [C][R]
    int B, C, R;
};

// rows: dlogit = dsg sg (1 - sg);  dgh = dlogit W2;  dhid = dgh gelu'(hid);  dy = dhid W1
__global__ __launch_bounds__(256) void se_bwd_rows_kernel(const SeBP p) {
    __shared__ F4 red[16 * 64];
    __shared__ __attribute__((aligned(16))) float dhl[16 * SE_LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * 16, m = m0 + r;
    const bool mok = m < p.B;
    const size_t bc = (size_t)p.B * p.C;
    for (int n_lo = 0; n_lo < p.R; n_lo += 64) {
        auto xf = [&](int kb4) -> F4 {
            if (!mok) return z4();
            F4 s = *(const F4*)(p.parts + (size_t)m * p.C + kb4);
            for (int q = 1; q < p.RS; ++q) s += *(const F4*)(p.parts + q * bc + (size_t)m * p.C + kb4);
            const F4 sg = *(const F4*)(p.sg + (size_t)m * p.C + kb4);
            F4 d;
#pragma unroll
            for (int v = 0; v < 4; ++v) d[v] = s[v] * p.scale * sg[v] * (1.f - sg[v]);
            if (n_lo == 0 && blockIdx.y == 0) *(F4*)(p.dlogit + (size_t)m * p.C + kb4) = d;
            return d;
        };
        const F4 dg = mm_bigK<true>(xf, p.w2, p.R, n_lo, p.R, p.C, red, wave, lane);
        const int n4 = n_lo + 16 * wave + 4 * g;
        if (n4 < p.R) {
            F4 dh = z4();
            if (mok) {
                const F4 h = *(const F4*)(p.hid + (size_t)m * p.R + n4);
#pragma unroll
                for (int v = 0; v < 4; ++v) dh[v] = dg[v] * gelu_grad_f(h[v]);
                if (blockIdx.y == 0) *(F4*)(p.dhid + (size_t)m * p.R + n4) = dh;
            }
            *(F4*)(dhl + r * SE_LD + n4) = dh;
        }
    }
    __syncthreads();
    mm_bigN<true>(dhl, SE_LD, p.w1, p.C, p.C, p.R, wave, lane, [&](int n4, F4 acc) {
        if (mok && n4 < p.C) *(F4*)(p.dy + (size_t)m * p.C + n4) = acc;
    }, blockIdx.y, gridDim.y);
}

// weights: dw2[c][r] += sum_b dlogit[b][c] gh[b][r];  dw1[r][c] += sum_b dhid[b][r] y[b][c]   (blocks [0, nA): dw2 tiles, then dw1 tiles)
__global__ __launch_bounds__(256) void se_bwd_w_kernel(const SeBP p, int nA, int kA, int kB) {
    __shared__ F4 red[16 * 64];
    __shared__ float redb[4 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int bid = blockIdx.x;
    if (bid < nA) {                                       // dw2: N = C, K = R
        const int kt = bid % kA, nt = bid / kA;
        tn_tile([&](int m, int n) { return p.dlogit[(size_t)m * p.C + n]; }, [&](int m, int k) { return p.gh[(size_t)m * p.R + k]; }, p.dw2, nullptr,
                false, 0, p.B, p.R, p.C, kt * 64, nt * 16, red, redb, wave, lane);
    } else {                                              // dw1: N = R, K = C
        bid -= nA;
        const int kt = bid % kB, nt = bid / kB;
        tn_tile([&](int m, int n) { return p.dhid[(size_t)m * p.R + n]; }, [&](int m, int k) { return p.y[(size_t)m * p.C + k]; }, p.dw1, nullptr,
                false, 0, p.B, p.C, p.R, kt * 64, nt * 16, red, redb, wave, lane);
    }
}


// =================================================================================================================================
// CoordAttn strip chain (new_scripy.py:105-129) between the pooling pass (x_h, x_w) and the gate pass (logits l_h, l_w):
//   z = conv1(x) ; a = gelu(bn1(z)) ; h2w = h2w_proj(a_h) ; w2h = w2h_proj(a_w) ;
//   x_h' = a_h + sigmoid(gamma_h) adapt_H(w2h) ; x_w' = a_w + sigmoid(gamma_w) adapt_W(h2w) ; l_h = conv_h(x_h') ; l_w = conv_w(x_w')
// adapt_L = F.adaptive_avg_pool2d along the strip axis (:119-120; the identity when H == W).
// Forward = 2 launches: ca_z_kernel (conv1 on both strips + per-block column statistics) and ca_mix_kernel (one workgroup per
// sample: BatchNorm from the folded statistics, GELU, both projections, the mix, both output convolutions).
// Backward = 2 launches: ca_bwd_mix_kernel (per sample: back through conv_h / conv_w, the mix, the projections and the GELU, the
// BatchNorm partial sums; + weight-gradient tiles of conv_h / conv_w) and ca_bwd_z_kernel (BatchNorm backward, back through
// conv1; + weight-gradient tiles of conv1 and of the projections).
// =================================================================================================================================
typedef DmCaChain CaP;

__device__ __forceinline__ void adapt_range(int i, int Lin, int Lout, int& s, int& e) {      // bin i of adaptive_avg_pool (Lin -> Lout)
    s = (i * Lin) / Lout;
    e = ((i + 1) * Lin + Lout - 1) / Lout;
}

// column sums of `nb` partial rows of `cols` floats each, folded in double in a fixed order (every workgroup gets the same bits);
// out[c] for c < cols.  `dred` holds >= 256 doubles.  All 256 threads must call.
__device__ __forceinline__ void fold_cols(const float* part, int nb, int cols, double* dred, double* out) {
    for (int c0 = 0; c0 < cols; c0 += 256) {
        const int nc = min(256, cols - c0), L = 256 / nc;
        const int col = threadIdx.x % nc, li = threadIdx.x / nc;
        double s = 0.0;
        if (li < L)
            for (int k = li; k < nb; k += L) s += (double)part[(size_t)k * cols + c0 + col];
        __syncthreads();
        if (li < L) dred[li * nc + col] = s;
        __syncthreads();
        if (threadIdx.x < nc) {
            double t = 0.0;
            for (int l = 0; l < L; ++l) t += dred[l * nc + threadIdx.x];
            out[c0 + threadIdx.x] = t;
        }
        __syncthreads();
    }
}

// ---- forward 1: z = x W1^T + b1 on 16-row blocks of both strips; per-block column sums of z and z^2 -> stat[block][2 R]
__global__ __launch_bounds__(256) void ca_z_kernel(const CaP p, int nbh) {
    __shared__ F4 red[16 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const bool isw = (int)blockIdx.x >= nbh;
    const int blk = isw ? blockIdx.x - nbh : blockIdx.x;
    const int M = p.B * (isw ? p.W : p.H);
    const float* x = isw ? p.xw : p.xh;
    const float* w1 = isw ? p.w1w : p.w1h;
    const float* b1 = isw ? p.b1w : p.b1h;
    float* z = isw ? p.zw : p.zh;
    const int m = blk * 16 + r;
    const bool mok = m < M;
    for (int n_lo = 0; n_lo < p.R; n_lo += 64) {
        auto xf = [&](int kb4) -> F4 { return mok ? *(const F4*)(x + (size_t)m * p.C + kb4) : z4(); };
        F4 v = mm_bigK<false>(xf, w1, p.C, n_lo, p.R, p.C, red, wave, lane);
        const int n4 = n_lo + 16 * wave + 4 * g;
        if (n4 < p.R) {
            if (b1) v += *(const F4*)(b1 + n4);
            if (mok) *(F4*)(z + (size_t)m * p.R + n4) = v;
            if (p.train) {
                F4 s1 = mok ? v : z4(), s2 = mok ? v * v : z4();
#pragma unroll
                for (int o = 8; o > 0; o >>= 1)
#pragma unroll
                    for (int q = 0; q < 4; ++q) { s1[q] += __shfl_xor(s1[q], o, 16); s2[q] += __shfl_xor(s2[q], o, 16); }
                if (r == 0) {
                    float* st = p.stat + (size_t)blockIdx.x * 2 * p.R;
                    *(F4*)(st + n4) = s1;
                    *(F4*)(st + p.R + n4) = s2;
                }
            }
        }
    }
}

// ---- forward 2: one workgroup per sample
// dynamic LDS: 6 matrices [Lp][LD] (a_h, a_w, h2w, w2h, x_h', x_w') + BatchNorm vectors
__global__ __launch_bounds__(256) void ca_mix_kernel(const CaP p, int nbh, int nbw, int Hp, int Wp, int LD) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ double dred[256];
    __shared__ double dsum[512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15;
    const int b = blockIdx.x, R = p.R, H = p.H, W = p.W;
    const bool lead = blockIdx.y == 0;                      // grid.y deals out the C logits; one workgroup per sample owns the side effects
    float* Ah = sm;
    float* Aw = Ah + Hp * LD;
    float* H2W = Aw + Wp * LD;
    float* W2H = H2W + Hp * LD;
    float* XH = W2H + Wp * LD;
    float* XW = XH + Hp * LD;
    float* bnv = XW + Wp * LD;                              // [2 strips][4][R]: mean, rstd, gamma, beta
    // BatchNorm statistics: batch (folded from the per-block sums of ca_z_kernel) or running
    for (int sIdx = 0; sIdx < 2; ++sIdx) {
        float* mean = bnv + sIdx * 4 * R;
        float* rstd = mean + R;
        const float* gm = sIdx ? p.bn_w_g : p.bn_h_g;
        const float* bt = sIdx ? p.bn_w_b : p.bn_h_b;
        float* rm = sIdx ? p.rm_w : p.rm_h;
        float* rv = sIdx ? p.rv_w : p.rv_h;
        const int M = p.B * (sIdx ? W : H);
        if (p.train) {
            fold_cols(p.stat + (size_t)(sIdx ? nbh : 0) * 2 * R, sIdx ? nbw : nbh, 2 * R, dred, dsum);
            for (int n = threadIdx.x; n < R; n += 256) {
                const double mu = dsum[n] / M;
                double var = dsum[R + n] / M - mu * mu;
                if (var < 0.0) var = 0.0;
                mean[n] = (float)mu;
                rstd[n] = (float)(1.0 / sqrt(var + (double)p.eps));
                if (b == 0 && lead) {                       // one workgroup owns the side effects
                    const double unb = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
                    rm[n] = (float)((1.0 - p.momentum) * (double)rm[n] + p.momentum * mu);
                    rv[n] = (float)((1.0 - p.momentum) * (double)rv[n] + p.momentum * unb);
                }
            }
        } else {
            for (int n = threadIdx.x; n < R; n += 256) {
                mean[n] = rm[n];
                rstd[n] = (float)(1.0 / sqrt((double)rv[n] + (double)p.eps));
            }
        }
        for (int n = threadIdx.x; n < R; n += 256) {
            mean[2 * R + n] = gm[n];
            mean[3 * R + n] = bt[n];
            if (b == 0 && lead && p.save) {
                (sIdx ? p.mean_w : p.mean_h)[n] = mean[n];
                (sIdx ? p.rstd_w : p.rstd_h)[n] = rstd[n];
            }
        }
        __syncthreads();
    }
    // a = gelu(bn(z)); pad rows are zero
    for (int sIdx = 0; sIdx < 2; ++sIdx) {
        const int L = sIdx ? W : H, Lp = sIdx ? Wp : Hp;
        float* A = sIdx ? Aw : Ah;
        const float* z = (sIdx ? p.zw : p.zh) + (size_t)b * L * R;
        float* ag = sIdx ? p.aw : p.ah;
        const float* mean = bnv + sIdx * 4 * R;
        for (int i = threadIdx.x; i < Lp * R; i += 256) {
            const int l = i / R, n = i - l * R;
            float v = 0.f;
            if (l < L) {
                v = gelu_f((z[i] - mean[n]) * mean[R + n] * mean[2 * R + n] + mean[3 * R + n]);
                if (p.save && lead) ag[(size_t)b * L * R + i] = v;
            }
            A[l * LD + n] = v;
        }
    }
    __syncthreads();
    // projections: h2w = a_h Whw^T + bhw ; w2h = a_w Wwh^T + bwh
    for (int sIdx = 0; sIdx < 2; ++sIdx) {
        const int L = sIdx ? W : H, Lp = sIdx ? Wp : Hp;
        const float* A = sIdx ? Aw : Ah;
        float* O = sIdx ? W2H : H2W;
        const float* wp = sIdx ? p.wwh : p.whw;
        const float* bp = sIdx ? p.bwh : p.bhw;
        for (int rb = 0; rb < Lp; rb += 16)
            mm_bigN<false>(A + rb * LD, LD, wp, R, R, R, wave, lane, [&](int n4, F4 acc) {
                if (n4 < R) {
                    if (bp) acc += *(const F4*)(bp + n4);
                    if (rb + r >= L) acc = z4();
                    *(F4*)(O + (rb + r) * LD + n4) = acc;
                }
            });
    }
    __syncthreads();
    // mix: x_h'[i] = a_h[i] + sig(gamma_h) mean_{j in bin_i(W -> H)} w2h[j]   (and the mirror image)
    const float sgh = sigmoid_f(p.gam_h[0]), sgw = sigmoid_f(p.gam_w[0]);
    for (int sIdx = 0; sIdx < 2; ++sIdx) {
        const int L = sIdx ? W : H, Lp = sIdx ? Wp : Hp, Lo = sIdx ? H : W;
        const float* A = sIdx ? Aw : Ah;
        const float* Oth = sIdx ? H2W : W2H;                // the OTHER strip's projection (length Lo), pooled to L bins
        float* X = sIdx ? XW : XH;
        float* xg = sIdx ? p.xwp : p.xhp;
        const float sg = sIdx ? sgw : sgh;
        for (int i = threadIdx.x; i < Lp * R; i += 256) {
            const int l = i / R, n = i - l * R;
            float v = 0.f;
            if (l < L) {
                int s0, e0;
                adapt_range(l, Lo, L, s0, e0);
                float t = 0.f;
                for (int j = s0; j < e0; ++j) t += Oth[j * LD + n];
                v = A[l * LD + n] + sg * (t / (float)(e0 - s0));
                if (p.save && lead) xg[(size_t)b * L * R + i] = v;
            }
            X[l * LD + n] = v;
        }
    }
    __syncthreads();
    // logits: l = x' Wc^T + bc -> [B][L][C]
    for (int sIdx = 0; sIdx < 2; ++sIdx) {
        const int L = sIdx ? W : H, Lp = sIdx ? Wp : Hp;
        const float* X = sIdx ? XW : XH;
        const float* wc = sIdx ? p.wcw : p.wch;
        const float* bc = sIdx ? p.bcw : p.bch;
        float* lo = (sIdx ? p.lw : p.lh) + (size_t)b * L * p.C;
        for (int rb = 0; rb < Lp; rb += 16)
            mm_bigN<false>(X + rb * LD, LD, wc, R, p.C, R, wave, lane, [&](int n4, F4 acc) {
                if (n4 < p.C && rb + r < L) {
                    if (bc) acc += *(const F4*)(bc + n4);
                    *(F4*)(lo + (size_t)(rb + r) * p.C + n4) = acc;
                }
            }, blockIdx.y, gridDim.y);
    }
}

// a weight-gradient problem of a mixed grid: dw[N][K] += g^T x over M rows, tiles (kt x nt x splits) from block `first`
struct TnJob { int first, kt, nt, splits, rps; };
__device__ __forceinline__ bool tn_decode(const TnJob& j, int bid, int& k0, int& n0, int& mlo, int& mhi, int M) {
    const int per = j.kt * j.nt, q = bid - j.first;
    if (q < 0 || q >= per * j.splits) return false;
    const int sp = q / per, t = q - sp * per;
    k0 = (t % j.kt) * 64;
    n0 = (t / j.kt) * 16;
    mlo = sp * j.rps;
    mhi = min(M, mlo + j.rps);
    return true;
}

// ---- backward 1: one workgroup per sample (blocks [0, B)), then the weight-gradient tiles of conv_h / conv_w
// dynamic LDS: 7 matrices [Lp][LD]: a_h, a_w, d x_h', d x_w', h2w / d h2w, w2h / d w2h, scratch
__global__ __launch_bounds__(256) void ca_bwd_mix_kernel(const CaP p, int Hp, int Wp, int LD, TnJob jh, TnJob jw) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ F4 red[16 * 64];
    __shared__ float redb[4 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const int R = p.R, H = p.H, W = p.W, C = p.C;
    if ((int)blockIdx.x >= p.B) {                          // d Wc[c][n] += sum_rows dl[row][c] x'[row][n] ; d bc[c] += sum_rows dl[row][c]
        int k0, n0, mlo, mhi;
        if (tn_decode(jh, blockIdx.x, k0, n0, mlo, mhi, p.B * H))
            tn_tile([&](int m, int n) { return p.dlh[(size_t)m * C + n]; }, [&](int m, int k) { return p.xhp[(size_t)m * R + k]; }, p.d_wch, p.d_bch,
                    k0 == 0, mlo, mhi, R, C, k0, n0, red, redb, wave, lane);
        else if (tn_decode(jw, blockIdx.x, k0, n0, mlo, mhi, p.B * W))
            tn_tile([&](int m, int n) { return p.dlw[(size_t)m * C + n]; }, [&](int m, int k) { return p.xwp[(size_t)m * R + k]; }, p.d_wcw, p.d_bcw,
                    k0 == 0, mlo, mhi, R, C, k0, n0, red, redb, wave, lane);
        return;
    }
    const int b = blockIdx.x;
    float* Ah = sm;
    float* Aw = Ah + Hp * LD;
    float* DXH = Aw + Wp * LD;
    float* DXW = DXH + Hp * LD;
    float* PH = DXW + Wp * LD;                              // h2w, then d h2w   [Hp]
    float* PW = PH + Hp * LD;                               // w2h, then d w2h   [Wp]
    for (int sIdx = 0; sIdx < 2; ++sIdx) {                  // a from the forward pass; pad rows zero
        const int L = sIdx ? W : H, Lp = sIdx ? Wp : Hp;
        float* A = sIdx ? Aw : Ah;
        const float* ag = (sIdx ? p.aw : p.ah) + (size_t)b * L * R;
        for (int i = threadIdx.x; i < Lp * R; i += 256) {
            const int l = i / R, n = i - l * R;
            A[l * LD + n] = l < L ? ag[i] : 0.f;
        }
    }
    // d x' = dl Wc   (reduction over the C channels: the waves split it)
    for (int sIdx = 0; sIdx < 2; ++sIdx) {
        const int L = sIdx ? W : H, Lp = sIdx ? Wp : Hp;
        float* DX = sIdx ? DXW : DXH;
        const float* dl = (sIdx ? p.dlw : p.dlh) + (size_t)b * L * C;
        const float* wc = sIdx ? p.wcw : p.wch;
        for (int rb = 0; rb < Lp; rb += 16) {
            const bool mok = rb + r < L;
            for (int n_lo = 0; n_lo < R; n_lo += 64) {
                auto xf = [&](int kb4) -> F4 { return mok ? *(const F4*)(dl + (size_t)(rb + r) * C + kb4) : z4(); };
                const F4 v = mm_bigK<true>(xf, wc, R, n_lo, R, C, red, wave, lane);
                const int n4 = n_lo + 16 * wave + 4 * g;
                if (n4 < R) *(F4*)(DX + (rb + r) * LD + n4) = v;
            }
        }
    }
    __syncthreads();
    // the projections again (only d gamma needs them): PH = h2w, PW = w2h
    for (int sIdx = 0; sIdx < 2; ++sIdx) {
        const int L = sIdx ? W : H, Lp = sIdx ? Wp : Hp;
        const float* A = sIdx ? Aw : Ah;
        float* O = sIdx ? PW : PH;
        const float* wp = sIdx ? p.wwh : p.whw;
        const float* bp = sIdx ? p.bwh : p.bhw;
        for (int rb = 0; rb < Lp; rb += 16)
            mm_bigN<false>(A + rb * LD, LD, wp, R, R, R, wave, lane, [&](int n4, F4 acc) {
                if (n4 < R) {
                    if (bp) acc += *(const F4*)(bp + n4);
                    if (rb + r >= L) acc = z4();
                    *(F4*)(O + (rb + r) * LD + n4) = acc;
                }
            });
    }
    __syncthreads();
    // d gamma_h = sig'(gamma_h) sum_{i,n} d x_h'[i][n] adapt_H(w2h)[i][n]   (and the mirror image)
    const float sgh = sigmoid_f(p.gam_h[0]), sgw = sigmoid_f(p.gam_w[0]);
    {
        float part[2] = {0.f, 0.f};
        for (int sIdx = 0; sIdx < 2; ++sIdx) {
            const int L = sIdx ? W : H, Lo = sIdx ? H : W;
            const float* DX = sIdx ? DXW : DXH;
            const float* Oth = sIdx ? PH : PW;
            for (int i = threadIdx.x; i < L * R; i += 256) {
                const int l = i / R, n = i - l * R;
                int s0, e0;
                adapt_range(l, Lo, L, s0, e0);
                float t = 0.f;
                for (int j = s0; j < e0; ++j) t += Oth[j * LD + n];
                part[sIdx] += DX[l * LD + n] * (t / (float)(e0 - s0));
            }
        }
        const float t0 = block_sum(part[0], redb), t1 = block_sum(part[1], redb + 32);
        if (threadIdx.x == 0) {
            unsafeAtomicAdd(p.d_gam + 0, t0 * sgh * (1.f - sgh));
            unsafeAtomicAdd(p.d_gam + 1, t1 * sgw * (1.f - sgw));
        }
    }
    __syncthreads();
    // d w2h[j] = sig(gamma_h) sum_{i: j in bin_i(W -> H)} d x_h'[i] / |bin_i|  -> PW (and global, for the weight gradients of w2h_proj)
    for (int sIdx = 0; sIdx < 2; ++sIdx) {
        // sIdx = 0: target d w2h (length W) from d x_h' (length H);  sIdx = 1: target d h2w (length H) from d x_w' (length W)
        const int Lt = sIdx ? H : W, Ltp = sIdx ? Hp : Wp, Ls = sIdx ? W : H;
        const float* DX = sIdx ? DXW : DXH;
        float* O = sIdx ? PH : PW;
        float* og = (sIdx ? p.dh2w : p.dw2h) + (size_t)b * Lt * R;
        const float sg = sIdx ? sgw : sgh;
        for (int i = threadIdx.x; i < Ltp * R; i += 256) {
            const int j = i / R, n = i - j * R;
            float t = 0.f;
            if (j < Lt) {
                for (int q = 0; q < Ls; ++q) {
                    int s0, e0;
                    adapt_range(q, Lt, Ls, s0, e0);
                    if (j >= s0 && j < e0) t += DX[q * LD + n] / (float)(e0 - s0);
                }
                t *= sg;
                og[i] = t;
            }
            O[j * LD + n] = t;
        }
    }
    __syncthreads();
    // d a_h = d x_h' + d h2w Whw ; g = d a gelu'(y), y = bn(z) ; per-sample BatchNorm sums
    for (int sIdx = 0; sIdx < 2; ++sIdx) {
        const int L = sIdx ? W : H, Lp = sIdx ? Wp : Hp;
        float* DX = sIdx ? DXW : DXH;
        const float* DP = sIdx ? PW : PH;                   // gradient of THIS strip's projection output
        const float* wp = sIdx ? p.wwh : p.whw;
        for (int rb = 0; rb < Lp; rb += 16)
            mm_bigN<true>(DP + rb * LD, LD, wp, R, R, R, wave, lane, [&](int n4, F4 acc) {
                if (n4 < R) *(F4*)(DX + (rb + r) * LD + n4) += acc;
            });
    }
    __syncthreads();
    for (int sIdx = 0; sIdx < 2; ++sIdx) {
        const int L = sIdx ? W : H;
        const float* DX = sIdx ? DXW : DXH;
        const float* z = (sIdx ? p.zw : p.zh) + (size_t)b * L * R;
        float* gg = (sIdx ? p.gw : p.gh) + (size_t)b * L * R;
        const float* mean = sIdx ? p.mean_w : p.mean_h;
        const float* rstd = sIdx ? p.rstd_w : p.rstd_h;
        const float* gm = sIdx ? p.bn_w_g : p.bn_h_g;
        const float* bt = sIdx ? p.bn_w_b : p.bn_h_b;
        float* part = p.bnpart + ((size_t)sIdx * p.B + b) * 2 * R;
        // thread (n, li): column n, rows li, li + Lr, ... ; LDS fold over the row lanes in a fixed order
        for (int n0 = 0; n0 < R; n0 += 256) {
            const int nc = min(256, R - n0), Lr = 256 / nc;
            const int n = n0 + threadIdx.x % nc, li = threadIdx.x / nc;
            float s1 = 0.f, s2 = 0.f;
            if (li < Lr) {
                const float mu = mean[n], rs = rstd[n], ga = gm[n], be = bt[n];
                for (int l = li; l < L; l += Lr) {
                    const float xh = (z[l * R + n] - mu) * rs;
                    const float gv = DX[l * LD + n] * gelu_grad_f(xh * ga + be);
                    gg[l * R + n] = gv;
                    s1 += gv;
                    s2 += gv * xh;
                }
            }
            float* fr = (float*)red;                        // [Lr][2][nc]
            __syncthreads();
            if (li < Lr) { fr[(li * 2 + 0) * nc + threadIdx.x % nc] = s1; fr[(li * 2 + 1) * nc + threadIdx.x % nc] = s2; }
            __syncthreads();
            if (threadIdx.x < nc) {
                float t1 = 0.f, t2 = 0.f;
                for (int l = 0; l < Lr; ++l) { t1 += fr[(l * 2 + 0) * nc + threadIdx.x]; t2 += fr[(l * 2 + 1) * nc + threadIdx.x]; }
                part[n0 + threadIdx.x] = t1;
                part[R + n0 + threadIdx.x] = t2;
            }
            __syncthreads();
        }
    }
}

// ---- backward 2: blocks [0, nbh + nbw): 16-row blocks — d z = BatchNorm backward of g, d x = d z W1 ; then weight-gradient tiles
// of conv1 (d z recomputed on the fly) and of the two projections
__global__ __launch_bounds__(256) void ca_bwd_z_kernel(const CaP p, int nbh, int nbw, TnJob j1h, TnJob j1w, TnJob jph, TnJob jpw) {
    __shared__ F4 red[16 * 64];
    __shared__ float redb[4 * 64];
    __shared__ double dred[256];
    __shared__ double dsum[256];
    __shared__ __attribute__((aligned(16))) float dzl[16 * SE_LD];
    __shared__ float cs[6 * 128];                           // per column: mean, rstd, gamma, beta, s1 / M, s2 / M
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15;
    const int R = p.R, C = p.C;
    int kind = -1, k0 = 0, n0 = 0, mlo = 0, mhi = 0;        // 0 / 1: row block of strip h / w; 2 / 3: conv1 tiles; 4 / 5: projection tiles
    const int bid = blockIdx.x;
    if (bid < nbh) kind = 0;
    else if (bid < nbh + nbw) kind = 1;
    else if (tn_decode(j1h, bid, k0, n0, mlo, mhi, p.B * p.H)) kind = 2;
    else if (tn_decode(j1w, bid, k0, n0, mlo, mhi, p.B * p.W)) kind = 3;
    else if (tn_decode(jph, bid, k0, n0, mlo, mhi, p.B * p.H)) kind = 4;
    else if (tn_decode(jpw, bid, k0, n0, mlo, mhi, p.B * p.W)) kind = 5;
    if (kind < 0) return;
    if (kind >= 4) {                                        // d Whw[n][k] += sum_rows d h2w[row][n] a_h[row][k] ; d bhw += column sums
        const bool w = kind == 5;
        const float* dp = w ? p.dw2h : p.dh2w;
        const float* a = w ? p.aw : p.ah;
        tn_tile([&](int m, int n) { return dp[(size_t)m * R + n]; }, [&](int m, int k) { return a[(size_t)m * R + k]; }, w ? p.d_wwh : p.d_whw,
                w ? p.d_bwh : p.d_bhw, k0 == 0, mlo, mhi, R, R, k0, n0, red, redb, wave, lane);
        return;
    }
    const bool w = kind & 1;
    const int M = p.B * (w ? p.W : p.H);
    // BatchNorm sums over all rows: fold of the per-sample partials (fixed order)
    fold_cols(p.bnpart + (size_t)(w ? 1 : 0) * p.B * 2 * R, p.B, 2 * R, dred, dsum);
    {
        const float* mean = w ? p.mean_w : p.mean_h;
        const float* rstd = w ? p.rstd_w : p.rstd_h;
        const float* gm = w ? p.bn_w_g : p.bn_h_g;
        const float* bt = w ? p.bn_w_b : p.bn_h_b;
        for (int n = threadIdx.x; n < R; n += 256) {
            cs[n] = mean[n]; cs[128 + n] = rstd[n]; cs[256 + n] = gm[n]; cs[384 + n] = bt[n];
            cs[512 + n] = p.train ? (float)(dsum[n] / M) : 0.f;
            cs[640 + n] = p.train ? (float)(dsum[R + n] / M) : 0.f;
            if (bid == (w ? nbh : 0)) {                     // d beta = sum g, d gamma = sum g xhat
                (w ? p.d_bn_w_b : p.d_bn_h_b)[n] = (float)dsum[n];
                (w ? p.d_bn_w_g : p.d_bn_h_g)[n] = (float)dsum[R + n];
            }
        }
    }
    __syncthreads();
    const float* gg = w ? p.gw : p.gh;
    const float* z = w ? p.zw : p.zh;
    auto dz_at = [&](int m, int n) -> float {               // d z = gamma rstd (g - s1/M - xhat s2/M)
        const float xh = (z[(size_t)m * R + n] - cs[n]) * cs[128 + n];
        return cs[256 + n] * cs[128 + n] * (gg[(size_t)m * R + n] - cs[512 + n] - xh * cs[640 + n]);
    };
    if (kind >= 2) {                                        // d W1[n][k] += sum_rows d z[row][n] x[row][k] ; d b1 += column sums of d z
        const float* x = w ? p.xw : p.xh;
        tn_tile(dz_at, [&](int m, int k) { return x[(size_t)m * C + k]; }, w ? p.d_w1w : p.d_w1h, w ? p.d_b1w : p.d_b1h, k0 == 0, mlo, mhi, C, R,
                k0, n0, red, redb, wave, lane);
        return;
    }
    const int blk = w ? bid - nbh : bid;
    for (int i = threadIdx.x; i < 16 * R; i += 256) {
        const int l = i / R, n = i - l * R, m = blk * 16 + l;
        dzl[l * SE_LD + n] = m < M ? dz_at(m, n) : 0.f;
    }
    __syncthreads();
    float* dx = w ? p.dxw : p.dxh;
    mm_bigN<true>(dzl, SE_LD, w ? p.w1w : p.w1h, C, C, R, wave, lane, [&](int n4, F4 acc) {
        const int m = blk * 16 + r;
        if (n4 < C && m < M) *(F4*)(dx + (size_t)m * C + n4) = acc;
    });
}

static TnJob tn_job(int& next, int M, int K, int N) {
    TnJob j;
    j.first = next;
    j.kt = cdiv(K, 64);
    j.nt = cdiv(N, 16);
    int splits = cdiv(M, 256);
    while (splits > 1 && (int64_t)splits * j.kt * j.nt > 1024) splits = (splits + 1) / 2;
    j.rps = cdiv(cdiv(M, splits), 16) * 16;
    j.splits = cdiv(M, j.rps);
    next += j.kt * j.nt * j.splits;
    return j;
}

constexpr size_t DM_CA_CHAIN_SMALL_LDS = (64 - 22) * 1024;
static int ca_geometry(const DmCaChain* d, const char* who, int nmat, int& Hp, int& Wp, int& LD, size_t& lds) {
    if (!(d && d->B > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->R > 0)) { dm_set_error("%s: bad geometry", who); return DM_EINVAL; }
    if (d->C % 4 || d->R % 4 || d->R > 128) { dm_set_error("%s: C %% 4 == 0, R %% 4 == 0, R <= 128 (got C=%d R=%d)", who, d->C, d->R); return DM_EINVAL; }
    Hp = cdiv(d->H, 16) * 16;
    Wp = cdiv(d->W, 16) * 16;
    LD = d->R + 4;
    lds = ((size_t)nmat / 2 * (Hp + Wp) * LD + 8 * d->R) * sizeof(float);
    if (lds > 120 * 1024) { dm_set_error("%s: strips of %d + %d positions x %d channels need %zu B of LDS", who, d->H, d->W, d->R, lds); return DM_EINVAL; }
    // The <= 64-KiB rule of a GPU shared between processes (conv variant 2, selected by the device guard: DESIGN.md section 6) binds the
    // chain kernels too: dynamic strips + <= 22 KiB of static scratch must stay inside 64 KiB.  The host then falls back to
    // the one-launch-per-op path (ops.ca_chain_ok) — this refusal is the backstop for callers of the C ABI.
    if (dm_get_conv_variant() == 2 && lds > DM_CA_CHAIN_SMALL_LDS) {
        dm_set_error("%s: %zu B of strip LDS (+ 22 KiB static) exceed the 64-KiB workgroup limit of a shared device (conv variant 2)", who, lds);
        return DM_EUNSUPPORTED;
    }
    return DM_OK;
}

}  // namespace

#define ST ((hipStream_t)s)

extern "C" int dm_ca_chain_fwd(const DmCaChain* d, dm_stream_t s) {
    int Hp, Wp, LD;
    size_t lds;
    int rc = ca_geometry(d, "dm_ca_chain_fwd", 6, Hp, Wp, LD, lds);
    if (rc) return rc;
    DM_CHECK_ARG(d->xh && d->xw && d->w1h && d->w1w && d->zh && d->zw && d->whw && d->wwh && d->gam_h && d->gam_w && d->wch && d->wcw && d->lh && d->lw &&
                 d->bn_h_g && d->bn_h_b && d->bn_w_g && d->bn_w_b && d->rm_h && d->rv_h && d->rm_w && d->rv_w, "dm_ca_chain_fwd: missing tensors");
    DM_CHECK_ARG(!d->train || d->stat, "dm_ca_chain_fwd: batch statistics need the `stat` scratch");
    DM_CHECK_ARG(!d->save || (d->mean_h && d->rstd_h && d->mean_w && d->rstd_w && d->ah && d->aw && d->xhp && d->xwp), "dm_ca_chain_fwd: save needs the saved-tensor slots");
    const int nbh = cdiv(d->B * d->H, 16), nbw = cdiv(d->B * d->W, 16);
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)ca_mix_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ca_bwd_mix_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
        if (e != hipSuccess) { dm_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
        attr = true;
    }
    hipLaunchKernelGGL(ca_z_kernel, dim3(nbh + nbw), dim3(256), 0, ST, *d, nbh);
    const int ny = d->C >= 1024 ? 4 : (d->C >= 512 ? 2 : 1);      // wide layers: the C logits of a sample over several workgroups
    hipLaunchKernelGGL(ca_mix_kernel, dim3(d->B, ny), dim3(256), lds, ST, *d, nbh, nbw, Hp, Wp, LD);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_ca_chain_bwd(const DmCaChain* d, dm_stream_t s) {
    int Hp, Wp, LD;
    size_t lds;
    int rc = ca_geometry(d, "dm_ca_chain_bwd", 6, Hp, Wp, LD, lds);
    if (rc) return rc;
    DM_CHECK_ARG(d->dlh && d->dlw && d->xh && d->xw && d->zh && d->zw && d->mean_h && d->rstd_h && d->mean_w && d->rstd_w && d->ah && d->aw && d->xhp &&
                 d->xwp && d->gh && d->gw && d->bnpart && d->dh2w && d->dw2h && d->dxh && d->dxw && d->d_w1h && d->d_w1w && d->d_bn_h_g && d->d_bn_h_b &&
                 d->d_bn_w_g && d->d_bn_w_b && d->d_whw && d->d_wwh && d->d_gam && d->d_wch && d->d_wcw, "dm_ca_chain_bwd: missing tensors");
    const int nbh = cdiv(d->B * d->H, 16), nbw = cdiv(d->B * d->W, 16);
    int next = d->B;
    const TnJob jh = tn_job(next, d->B * d->H, d->R, d->C), jw = tn_job(next, d->B * d->W, d->R, d->C);
    hipLaunchKernelGGL(ca_bwd_mix_kernel, dim3(next), dim3(256), lds, ST, *d, Hp, Wp, LD, jh, jw);
    int next2 = nbh + nbw;
    const TnJob j1h = tn_job(next2, d->B * d->H, d->C, d->R), j1w = tn_job(next2, d->B * d->W, d->C, d->R);
    const TnJob jph = tn_job(next2, d->B * d->H, d->R, d->R), jpw = tn_job(next2, d->B * d->W, d->R, d->R);
    hipLaunchKernelGGL(ca_bwd_z_kernel, dim3(next2), dim3(256), 0, ST, *d, nbh, nbw, j1h, j1w, jph, jpw);
    DM_LAUNCH_CHECK();
    return DM_OK;
}


// attn.hip: sums over the pixels of a sample, optionally left as `*rs_out` partial arrays in the workspace (no fold launch)
int dm_strip_partials(int mode, const void* x, const void* y, int dtype, int B, int HW, int C, float* out, const float** parts, int* rs_out,
                      hipStream_t st);

extern "C" int dm_se_fwd(const void* x2, int dtype, int B, int HW, int C, const float* w1, const float* w2, int R, float* y, float* hid,
                         float* gh, float* sg, dm_stream_t s) {
    DM_CHECK_ARG(x2 && w1 && w2 && sg && B > 0 && HW > 0 && C > 0 && R > 0, "dm_se_fwd: bad arguments");
    DM_CHECK_ARG(C % 4 == 0 && R % 4 == 0 && R <= 128, "dm_se_fwd: C %% 4 == 0, R %% 4 == 0, R <= 128 (got C=%d R=%d)", C, R);
    DM_CHECK_ARG((y && hid && gh) || (!y && !hid && !gh), "dm_se_fwd: y / hid / gh are saved together or not at all");
    SeP p;
    p.save = y != nullptr;
    // the pooling pass leaves its partial sums in the workspace (or the whole sums in a scratch row block of it)
    int rc = dm_strip_partials(0, x2, nullptr, dtype, B, HW, C, nullptr, &p.parts, &p.RS, ST);
    if (rc) return rc;
    p.scale = 1.f / (float)HW;
    p.w1 = w1; p.w2 = w2; p.y = y; p.hid = hid; p.gh = gh; p.sg = sg;
    p.B = B; p.C = C; p.R = R;
    // the hidden vector (a reduction over C) is recomputed by every workgroup of a row block; the C logits are dealt out over grid.y
    const int ny = C >= 1024 ? 4 : (C >= 512 ? 2 : 1);
    hipLaunchKernelGGL(se_fwd_kernel, dim3(cdiv(B, 16), ny), dim3(256), 0, ST, p);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_se_bwd(const void* dout, const void* x2, int dtype, int B, int HW, int C, float inv, const float* sg, const float* hid,
                         const float* gh, const float* y, const float* w1, const float* w2, int R, float* dlogit, float* dhid, float* dy,
                         float* dw1, float* dw2, dm_stream_t s) {
    DM_CHECK_ARG(dout && x2 && sg && hid && gh && y && w1 && w2 && dlogit && dhid && dy && dw1 && dw2 && B > 0 && HW > 0 && C > 0 && R > 0,
                 "dm_se_bwd: bad arguments");
    DM_CHECK_ARG(C % 4 == 0 && R % 4 == 0 && R <= 128, "dm_se_bwd: C %% 4 == 0, R %% 4 == 0, R <= 128 (got C=%d R=%d)", C, R);
    SeBP p;
    int rc = dm_strip_partials(1, dout, x2, dtype, B, HW, C, nullptr, &p.parts, &p.RS, ST);
    if (rc) return rc;
    p.scale = inv;
    p.sg = sg; p.hid = hid; p.gh = gh; p.y = y; p.w1 = w1; p.w2 = w2;
    p.dlogit = dlogit; p.dhid = dhid; p.dy = dy; p.dw1 = dw1; p.dw2 = dw2;
    p.B = B; p.C = C; p.R = R;
    const int ny = C >= 1024 ? 4 : (C >= 512 ? 2 : 1);
    hipLaunchKernelGGL(se_bwd_rows_kernel, dim3(cdiv(B, 16), ny), dim3(256), 0, ST, p);
    const int kA = cdiv(R, 64), nA = kA * cdiv(C, 16), kB = cdiv(C, 64), nB = kB * cdiv(R, 16);
    hipLaunchKernelGGL(se_bwd_w_kernel, dim3(nA + nB), dim3(256), 0, ST, p, nA, kA, kB);
    DM_LAUNCH_CHECK();
    return DM_OK;
}
