// DDPM wrapper kernels: q-sample, weighted loss (+ gradient), CFG combine + ancestral update with a
// device-resident step counter (hipGraph replay needs no host-side arguments), Philox N(0,1),
// and the fused clip + AdamW optimiser step on flat fp32 buffers.  All HBM-bound.
#include "common.h"

namespace {

// ---- Philox4x32-10 ---------------------------------------------------------------------------
struct U4 { uint32_t x, y, z, w; };
__device__ inline U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = U4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}
__device__ inline void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
    const float u1 = ((float)a + 1.0f) * 2.3283064365386963e-10f;  // (0, 1]
    const float u2 = (float)b * 2.3283064365386963e-10f;
    const float r = sqrtf(-2.0f * __logf(u1));
    float sn, cs;
    __sincosf(6.283185307179586f * u2, &sn, &cs);
    z0 = r * cs;
    z1 = r * sn;
}
__device__ inline void normal4(uint64_t seed, uint64_t ctr_hi, uint64_t ctr_lo, float* z) {
    const U4 r = philox4x32_10(U4{(uint32_t)ctr_lo, (uint32_t)(ctr_lo >> 32), (uint32_t)ctr_hi, (uint32_t)(ctr_hi >> 32)},
                               (uint32_t)seed, (uint32_t)(seed >> 32));
    box_muller(r.x, r.y, z[0], z[1]);
    box_muller(r.z, r.w, z[2], z[3]);
}

__global__ void randn_kernel(float* out, int64_t n, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, int64_t first_quad) {
    if (offset_dev) offset = *offset_dev;              // stream offset kept on the device (hipGraph replays advance it)
    const int64_t n4 = (n + 3) / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float z[4];
        normal4(seed, offset, (uint64_t)(first_quad + i), z);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i * 4 + k < n) out[i * 4 + k] = z[k];
    }
}

// The per-sample draws of DDPM.forward (new_scripy.py:405, 413): t ~ U{1..n_T} and the classifier-free-guidance keep mask
// ~ Bernoulli(keep_prob), from the same Philox key as the noise with a counter range of their own (bit 62 of the low word), at the
// stream offset held on the device — which this launch ADVANCES by one (one workgroup: every thread reads the offset, then thread 0
// writes it back), so a replayed step (hipGraph or launch plan) draws fresh timesteps with no host involvement.  Also emits
// t / n_T as the float the network is fed (:415, an fp32 true division like torch's).
__global__ __launch_bounds__(256) void draw_ts_keep_kernel(int64_t* ts, float* tfrac, float* keep, int B, int n_T, float keep_prob, uint64_t seed,
                                                           uint64_t* offset_dev) {
    const uint64_t offset = *offset_dev + 1;
    for (int i = threadIdx.x; i < B; i += 256) {
        const uint64_t lo = (1ull << 62) + (uint64_t)i;
        const U4 r = philox4x32_10(U4{(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)}, (uint32_t)seed, (uint32_t)(seed >> 32));
        const int64_t t = 1 + (int64_t)(((uint64_t)r.x * (uint64_t)n_T) >> 32);
        ts[i] = t;
        tfrac[i] = (float)t / (float)n_T;
        keep[i] = ((float)r.y * 2.3283064365386963e-10f) < keep_prob ? 1.f : 0.f;
    }
    __syncthreads();
    if (threadIdx.x == 0) *offset_dev = offset;
}

// ---- q-sample: NCHW fp32 -> NHWC T (Cp) ------------------------------------------------------
template <typename T>
__global__ void qsample_kernel(const float* x, const float* noise, const int64_t* ts, const float* sqrtab, const float* sqrtmab,
                               T* xt, int B, int C, int HW, int Cp) {
    const int64_t total = (int64_t)B * HW * Cp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % Cp);
        const int64_t pix = i / Cp;
        const int p = (int)(pix % HW);
        const int b = (int)(pix / HW);
        float v = 0.f;
        if (c < C) {
            const size_t j = ((size_t)b * C + c) * HW + p;
            const int64_t t = ts[b];
            v = sqrtab[t] * x[j] + sqrtmab[t] * noise[j];
        }
        Elem<T>::st(xt + i, v);
    }
}

// ---- loss -------------------------------------------------------------------------------------
__global__ void loss_fwd_kernel(const float* pred, const float* noise, const float* mask, const float* cfg, float* loss, int B, int C,
                                int HW) {
    __shared__ float red[16];
    const int64_t total = (int64_t)B * C * HW;
    const float inv = 1.f / (float)total;
    float hi_t = 0, mid_t = 0, hi_w = 1, mid_w = 1, lo_w = 1, feat_w = 0;
    if (mask) { hi_t = cfg[0]; mid_t = cfg[1]; hi_w = cfg[2]; mid_w = cfg[3]; lo_w = cfg[4]; feat_w = cfg[5]; }
    float s1 = 0.f, s2 = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const float d = noise[i] - pred[i];
        float w = 1.f, h = 0.f;
        if (mask) {
            const int p = (int)(i % HW);
            const int b = (int)(i / ((int64_t)C * HW));
            const float m = mask[(size_t)b * HW + p];
            w = m > hi_t ? hi_w : (m > mid_t ? mid_w : lo_w);
            h = m > hi_t ? 1.f : 0.f;
        }
        s1 += d * d * w;
        s2 += fabsf(pred[i] * h - noise[i] * h);
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) atomicAdd(loss, s1 * inv + s2 * inv * feat_w);
}

template <typename T>
__global__ void loss_bwd_kernel(const float* pred, const float* noise, const float* mask, const float* cfg, const float* gscale,
                                T* dpred, int B, int C, int HW, int Cp) {
    const bool nchw = Cp == 0;  // Cp == 0: write NCHW (same indexing as pred), T must be float
    if (nchw) Cp = C;
    const int64_t total = (int64_t)B * HW * Cp;
    const float inv = gscale[0] / (float)((int64_t)B * C * HW);
    float hi_t = 0, mid_t = 0, hi_w = 1, mid_w = 1, lo_w = 1, feat_w = 0;
    if (mask) { hi_t = cfg[0]; mid_t = cfg[1]; hi_w = cfg[2]; mid_w = cfg[3]; lo_w = cfg[4]; feat_w = cfg[5]; }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % Cp);
        const int64_t pix = i / Cp;
        const int p = (int)(pix % HW);
        const int b = (int)(pix / HW);
        float g = 0.f;
        if (c < C) {
            const size_t j = ((size_t)b * C + c) * HW + p;
            float w = 1.f, h = 0.f;
            if (mask) {
                const float m = mask[(size_t)b * HW + p];
                w = m > hi_t ? hi_w : (m > mid_t ? mid_w : lo_w);
                h = m > hi_t ? 1.f : 0.f;
            }
            const float d = pred[j] - noise[j];
            const float e = pred[j] * h - noise[j] * h;
            const float sg = e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f);
            g = inv * (2.f * w * d + feat_w * h * sg);
        }
        if (nchw) { if (c < C) Elem<T>::st(dpred + ((size_t)b * C + c) * HW + p, g); }
        else Elem<T>::st(dpred + i, g);
    }
}

// ---- CFG combine + ancestral update ------------------------------------------------------------
__global__ void cfg_update_kernel(float* x, const float* eps2n, const float* z, float gw, const float* a_t, const float* b_t,
                                  const float* s_t, const int32_t* step, uint64_t seed, int64_t n, int64_t first_quad) {
    const int i_t = step[0];
    const float a = a_t[i_t], bb = b_t[i_t], sg = i_t > 1 ? s_t[i_t] : 0.f;
    const int64_t n4 = (n + 3) / 4;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += (int64_t)gridDim.x * 256) {
        float zz[4] = {0.f, 0.f, 0.f, 0.f};
        if (i_t > 1 && !z) normal4(seed, (uint64_t)i_t, (uint64_t)(first_quad + q), zz);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t i = q * 4 + k;
            if (i >= n) break;
            const float e1 = eps2n[i], e2 = eps2n[n + i];
            const float e = (1.f + gw) * e1 - gw * e2;
            const float zk = (i_t > 1 && z) ? z[i] : zz[k];
            x[i] = a * (x[i] - e * bb) + sg * zk;
        }
    }
}
__global__ void dec_step_kernel(int32_t* step) {
    if (threadIdx.x == 0 && blockIdx.x == 0) step[0] -= 1;
}
__global__ void fill_t_kernel(float* t, const int32_t* step, int n_T, int B) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < B) t[i] = (float)step[0] / (float)n_T;
}

// ---- optimiser -----------------------------------------------------------------------------------
__global__ void sumsq_kernel(const float* g, int64_t n, float* out) {
    __shared__ float red[16];
    float s = 0.f;
    const int64_t n4 = n / 4;
    // (r04: four loads in flight per thread instead of one measured the same 86 - 92 us for 426 MB — the pass already reads at the ~4.7 TB/s a
    //  pure read stream reaches on this part)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = ((const f32x4*)g)[i];
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - n4 * 4)) { const float v = g[n4 * 4 + threadIdx.x]; s += v * v; }
    s = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(out, s);
}

// 4 parameters per lane-iteration (16-byte loads / stores of p, g, m, v: 28 B of traffic per parameter, HBM-bound);
// optionally refreshes the bf16 shadow of the parameters that the conv kernels read (p16[i] = bf16(p[i])).
// `scaler` (may be null): dynamic loss-scaling state {scale, growth tracker, found_inf of this step, 1 / scale the gradients of this step
// were produced under} maintained by scaler_update_kernel: a step whose gradients hold inf / nan is SKIPPED (torch.amp.GradScaler.step)
template <typename T16>
__global__ __launch_bounds__(256) void adamw_kernel(float* p, const float* g, float* m, float* v, int64_t n, const float* sumsq, const float* hy,
                                                    T16* p16, const int64_t* step_dev, const float* scaler) {
    if (scaler && scaler[2] != 0.f) return;
    const float lr = hy[0], b1 = hy[1], b2 = hy[2], eps = hy[3], wd = hy[4], max_norm = hy[5], gscale = hy[6] * (scaler ? scaler[3] : 1.f);
    float bc1 = hy[7], bc2 = hy[8];
    if (step_dev) {                                    // step count kept on the device (hipGraph replays advance it)
        const double t = (double)*step_dev;
        bc1 = (float)(1.0 - pow((double)b1, t));
        bc2 = (float)(1.0 - pow((double)b2, t));
    }
    float coef = gscale;
    if (max_norm > 0.f) {
        const float total = sqrtf(sumsq[0]) * gscale;
        const float cc = max_norm / (total + 1e-6f);
        coef *= cc < 1.f ? cc : 1.f;
    }
    const float step_size = lr / bc1, rbc2 = 1.f / sqrtf(bc2), decay = 1.f - lr * wd;
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 gv = ((const f32x4*)g)[i], mv = ((const f32x4*)m)[i], vv4 = ((const f32x4*)v)[i], pv = ((const f32x4*)p)[i];
        f32x4 mo, vo, po;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gg = gv[e] * coef;
            const float mm = b1 * mv[e] + (1.f - b1) * gg;
            const float vv = b2 * vv4[e] + (1.f - b2) * gg * gg;
            mo[e] = mm;
            vo[e] = vv;
            po[e] = pv[e] * decay - step_size * mm / (sqrtf(vv) * rbc2 + eps);
        }
        ((f32x4*)m)[i] = mo;
        ((f32x4*)v)[i] = vo;
        ((f32x4*)p)[i] = po;
        if (p16) ((typename V16<T16>::x4*)p16)[i] = (typename V16<T16>::x4){(T16)po[0], (T16)po[1], (T16)po[2], (T16)po[3]};
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {   // tail (< 4 elements)
        const float gg = g[i] * coef;
        const float mm = b1 * m[i] + (1.f - b1) * gg;
        const float vv = b2 * v[i] + (1.f - b2) * gg * gg;
        m[i] = mm;
        v[i] = vv;
        const float po = p[i] * decay - step_size * mm / (sqrtf(vv) * rbc2 + eps);
        p[i] = po;
        if (p16) p16[i] = (T16)po;
    }
}

// torch.amp.GradScaler.update() + the found-inf test of .step(), on the device (one thread): runs after dm_sumsq, before dm_adamw.
// st = {scale, growth tracker, found_inf, inv_scale}: found_inf = the sum of squares of the (scaled) gradients is not finite;
// then the optimiser step is skipped (the step count taken back), scale *= backoff, tracker = 0; otherwise tracker += 1 and
// every `interval` clean steps scale *= growth.  inv_scale = 1 / (the scale this step's gradients carry).
__global__ void scaler_update_kernel(float* st, const float* sumsq, int64_t* step_dev, float growth, float backoff, int interval) {
    const float s = st[0];
    const bool bad = !(sumsq[0] < __builtin_huge_valf());      // inf or nan
    st[3] = 1.f / s;
    st[2] = bad ? 1.f : 0.f;
    if (bad) {
        st[0] = s * backoff;
        st[1] = 0.f;
        if (step_dev) *step_dev -= 1;
    } else {
        const float t = st[1] + 1.f;
        if ((int)t >= interval) { st[0] = s * growth; st[1] = 0.f; }
        else st[1] = t;
    }
}

}  // namespace

#define ST ((hipStream_t)s)

extern "C" int dm_randn(float* out, int64_t n, uint64_t seed, uint64_t offset, dm_stream_t s) {
    DM_CHECK_ARG(out && n > 0, "dm_randn: bad arguments");
    hipLaunchKernelGGL(randn_kernel, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, ST, out, n, seed, offset, (const uint64_t*)nullptr, (int64_t)0);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_randn_slice(float* out, int64_t n, uint64_t seed, uint64_t offset, int64_t first_elem, dm_stream_t s) {
    DM_CHECK_ARG(out && n > 0 && first_elem >= 0 && first_elem % 4 == 0, "dm_randn_slice: bad arguments (first_elem must be a multiple of 4)");
    hipLaunchKernelGGL(randn_kernel, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, ST, out, n, seed, offset, (const uint64_t*)nullptr, first_elem / 4);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_randn_dev(float* out, int64_t n, uint64_t seed, const uint64_t* offset_dev, dm_stream_t s) {
    DM_CHECK_ARG(out && n > 0 && offset_dev, "dm_randn_dev: bad arguments");
    hipLaunchKernelGGL(randn_kernel, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, ST, out, n, seed, (uint64_t)0, offset_dev, (int64_t)0);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_draw_ts_keep(int64_t* ts, float* t_frac, float* keep, int B, int n_T, float keep_prob, uint64_t seed, uint64_t* offset_dev,
                               dm_stream_t s) {
    DM_CHECK_ARG(ts && t_frac && keep && offset_dev && B > 0 && n_T > 0 && keep_prob >= 0.f && keep_prob <= 1.f, "dm_draw_ts_keep: bad arguments");
    hipLaunchKernelGGL(draw_ts_keep_kernel, dim3(1), dim3(256), 0, ST, ts, t_frac, keep, B, n_T, keep_prob, seed, offset_dev);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_qsample(const float* x, const float* noise, const int64_t* ts, const float* sqrtab, const float* sqrtmab, void* xt,
                          int dtype, int B, int C, int H, int W, int Cp, dm_stream_t s) {
    DM_CHECK_ARG(x && noise && ts && sqrtab && sqrtmab && xt && B > 0 && C > 0 && H > 0 && W > 0 && Cp >= C, "dm_qsample: bad arguments");
    DM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((qsample_kernel<T>), dim3(grid_for((int64_t)B * H * W * Cp, 256)), dim3(256), 0, ST, x, noise, ts, sqrtab, sqrtmab, (T*)xt, B, C, H * W, Cp));
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_loss_fwd(const float* pred, const float* noise, const float* mask, const float* cfg6, float* loss, int B, int C, int H,
                           int W, dm_stream_t s) {
    DM_CHECK_ARG(pred && noise && loss && (!mask || cfg6) && B > 0 && C > 0 && H > 0 && W > 0, "dm_loss_fwd: bad arguments");
    hipLaunchKernelGGL(loss_fwd_kernel, dim3(grid_for((int64_t)B * C * H * W, 256, 512)), dim3(256), 0, ST, pred, noise, mask, cfg6, loss, B, C, H * W);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_loss_bwd(const float* pred, const float* noise, const float* mask, const float* cfg6, const float* gscale, void* dpred,
                           int dtype, int B, int C, int H, int W, int Cp, dm_stream_t s) {
    DM_CHECK_ARG(pred && noise && gscale && dpred && (!mask || cfg6) && B > 0 && C > 0 && H > 0 && W > 0 && (Cp >= C || (Cp == 0 && dtype == DM_F32)), "dm_loss_bwd: bad arguments");
    DM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((loss_bwd_kernel<T>), dim3(grid_for((int64_t)B * H * W * (Cp ? Cp : C), 256)), dim3(256), 0, ST, pred, noise, mask, cfg6, gscale, (T*)dpred, B, C, H * W, Cp));
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_cfg_update_slice(float* x, const float* eps2n, const float* z, float guide_w, const float* oneover_sqrta,
                                   const float* mab_over_sqrtmab, const float* sqrt_beta_t, int32_t* step, uint64_t seed, int64_t n_elems,
                                   int64_t first_elem, int dec_step, dm_stream_t s) {
    DM_CHECK_ARG(x && eps2n && oneover_sqrta && mab_over_sqrtmab && sqrt_beta_t && step && n_elems > 0, "dm_cfg_update: bad arguments");
    DM_CHECK_ARG(first_elem >= 0 && first_elem % 4 == 0, "dm_cfg_update_slice: first_elem must be a non-negative multiple of 4");
    hipLaunchKernelGGL(cfg_update_kernel, dim3(grid_for((n_elems + 3) / 4, 256)), dim3(256), 0, ST, x, eps2n, z, guide_w, oneover_sqrta, mab_over_sqrtmab, sqrt_beta_t, step, seed, n_elems, first_elem / 4);
    if (dec_step) hipLaunchKernelGGL(dec_step_kernel, dim3(1), dim3(64), 0, ST, step);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_cfg_update(float* x, const float* eps2n, const float* z, float guide_w, const float* oneover_sqrta,
                             const float* mab_over_sqrtmab, const float* sqrt_beta_t, int32_t* step, uint64_t seed, int64_t n_elems,
                             int dec_step, dm_stream_t s) {
    return dm_cfg_update_slice(x, eps2n, z, guide_w, oneover_sqrta, mab_over_sqrtmab, sqrt_beta_t, step, seed, n_elems, 0, dec_step, s);
}

extern "C" int dm_fill_t(float* t, const int32_t* step, int n_T, int B, dm_stream_t s) {
    DM_CHECK_ARG(t && step && n_T > 0 && B > 0, "dm_fill_t: bad arguments");
    hipLaunchKernelGGL(fill_t_kernel, dim3(cdiv(B, 256)), dim3(256), 0, ST, t, step, n_T, B);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_sumsq(const float* g, int64_t n, float* out, dm_stream_t s) {
    DM_CHECK_ARG(g && out && n > 0 && ((uintptr_t)g & 15) == 0, "dm_sumsq: bad arguments (g must be 16-byte aligned)");
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for(n / 4 + 1, 256, 1024)), dim3(256), 0, ST, g, n, out);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_adamw(float* p, const float* g, float* m, float* v, int64_t n, const float* sumsq, const float* hyper9, void* p_bf16,
                        const int64_t* step_dev, dm_stream_t s) {
    DM_CHECK_ARG(p && g && m && v && sumsq && hyper9 && n > 0, "dm_adamw: bad arguments");
    DM_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0 && ((uintptr_t)p_bf16 & 7) == 0,
                 "dm_adamw: the flat buffers must be 16-byte aligned (the bf16 shadow 8-byte)");
    static int blocks_env = -1;
    if (blocks_env < 0) { const char* e = getenv("DM_ADAMW_BLOCKS"); blocks_env = e ? atoi(e) : 0; }
    const int blocks = blocks_env > 0 ? blocks_env : grid_for(n / 4 + 1, 256, 16384);
    hipLaunchKernelGGL(adamw_kernel<bf16>, dim3(blocks), dim3(256), 0, ST, p, g, m, v, n, sumsq, hyper9, (bf16*)p_bf16, step_dev, (const float*)nullptr);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_adamw_scaled(float* p, const float* g, float* m, float* v, int64_t n, const float* sumsq, const float* hyper9, void* p16,
                               int p16_dtype, const int64_t* step_dev, const float* scaler4, dm_stream_t s) {
    DM_CHECK_ARG(p && g && m && v && sumsq && hyper9 && n > 0, "dm_adamw_scaled: bad arguments");
    DM_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0 && ((uintptr_t)p16 & 7) == 0,
                 "dm_adamw_scaled: the flat buffers must be 16-byte aligned (the 16-bit shadow 8-byte)");
    DM_CHECK_ARG(p16 == nullptr || p16_dtype == DM_BF16 || p16_dtype == DM_F16, "dm_adamw_scaled: the shadow is bf16 or fp16");
    const int blocks = grid_for(n / 4 + 1, 256, 16384);
    if (p16_dtype == DM_F16) hipLaunchKernelGGL(adamw_kernel<f16>, dim3(blocks), dim3(256), 0, ST, p, g, m, v, n, sumsq, hyper9, (f16*)p16, step_dev, scaler4);
    else hipLaunchKernelGGL(adamw_kernel<bf16>, dim3(blocks), dim3(256), 0, ST, p, g, m, v, n, sumsq, hyper9, (bf16*)p16, step_dev, scaler4);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_scaler_update(float* state4, const float* sumsq, int64_t* step_dev, float growth, float backoff, int interval, dm_stream_t s) {
    DM_CHECK_ARG(state4 && sumsq && growth >= 1.f && backoff > 0.f && backoff <= 1.f && interval > 0, "dm_scaler_update: bad arguments");
    hipLaunchKernelGGL(scaler_update_kernel, dim3(1), dim3(1), 0, ST, state4, sumsq, step_dev, growth, backoff, interval);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

// Multi-tensor copy: table[e] = {src_ptr, dst_ptr, count} (fp32 elements), one workgroup per entry chunk.
namespace {
__global__ void scatter_copy_kernel(const int64_t* table, int n_entries, int add) {
    for (int e = blockIdx.x; e < n_entries; e += gridDim.x) {
        const float* src = (const float*)table[3 * e + 0];
        float* dst = (float*)table[3 * e + 1];
        const int64_t n = table[3 * e + 2];
        if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {      // 16-byte path (r04: the gradient gather moved ~15 MB in 39 us with scalar copies)
            const int64_t n4 = n >> 2;
            for (int64_t i = threadIdx.x; i < n4; i += blockDim.x) {
                const f32x4 v = ((const f32x4*)src)[i];
                ((f32x4*)dst)[i] = add ? ((f32x4*)dst)[i] + v : v;
            }
            for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) dst[i] = add ? dst[i] + src[i] : src[i];
            continue;
        }
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = add ? dst[i] + src[i] : src[i];
    }
}
}  // namespace
extern "C" int dm_scatter_copy(const int64_t* table_dev, int n_entries, int add, dm_stream_t s) {
    DM_CHECK_ARG(table_dev && n_entries > 0, "dm_scatter_copy: bad arguments");
    hipLaunchKernelGGL(scatter_copy_kernel, dim3(n_entries < 1024 ? n_entries : 1024), dim3(256), 0, ST, table_dev, n_entries, add);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

namespace {
__global__ void zero_ranges_kernel(const int64_t* table, int n_entries) {
    for (int e = blockIdx.x; e < n_entries; e += gridDim.x) {
        f32x4* dst = (f32x4*)table[2 * e + 0];
        const int64_t n4 = table[2 * e + 1] >> 2;
        for (int64_t i = threadIdx.x; i < n4; i += blockDim.x) dst[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}
}  // namespace
extern "C" int dm_zero_ranges(const int64_t* table_dev, int n_entries, dm_stream_t s) {
    DM_CHECK_ARG(table_dev && n_entries > 0, "dm_zero_ranges: bad arguments");
    hipLaunchKernelGGL(zero_ranges_kernel, dim3(n_entries < 4096 ? n_entries : 4096), dim3(256), 0, ST, table_dev, n_entries);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

// ---- evaluation helpers (new_scripy.py:1188-1250) and attention-mask rasterisation (:533-546) ------------------
namespace {
// one workgroup per image pair: {sum a, sum b, sum a^2, sum b^2, sum ab, min a, min b, n} in double
__global__ __launch_bounds__(256) void image_moments_kernel(const float* a, const float* b, double* out, int64_t n) {
    __shared__ double red[8][4];
    const float* pa = a + (size_t)blockIdx.x * n;
    const float* pb = b + (size_t)blockIdx.x * n;
    double s[5] = {0, 0, 0, 0, 0};
    float mna = INFINITY, mnb = INFINITY;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const double x = pa[i], y = pb[i];
        s[0] += x; s[1] += y; s[2] += x * x; s[3] += y * y; s[4] += x * y;
        mna = fminf(mna, pa[i]); mnb = fminf(mnb, pb[i]);
    }
    double v[7] = {s[0], s[1], s[2], s[3], s[4], (double)mna, (double)mnb};
#pragma unroll
    for (int k = 0; k < 7; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double other = __shfl_xor(v[k], o, 64);
            v[k] = k < 5 ? v[k] + other : fmin(v[k], other);
        }
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 7; ++k) red[k][w] = v[k];
    __syncthreads();
    if (threadIdx.x < 7) {
        const int k = threadIdx.x;
        double r = red[k][0];
        for (int j = 1; j < 4; ++j) r = k < 5 ? r + red[k][j] : fmin(r, red[k][j]);
        out[(size_t)blockIdx.x * 8 + k] = r;
    }
    if (threadIdx.x == 7) out[(size_t)blockIdx.x * 8 + 7] = (double)n;
}

__global__ void attn_mask_kernel(const int32_t* boxes, float* out, int B, int S, float lo, float mid, float hi) {
    const int64_t total = (int64_t)B * S * S;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % S), y = (int)((i / S) % S), b = (int)(i / ((int64_t)S * S));
        const int32_t* q = boxes + 4 * b;      // x0, y0, x1, y1 (already scaled and clamped), half-open like the slice
        float v = y >= S / 2 ? mid : lo;
        if (y >= q[1] && y < q[3] && x >= q[0] && x < q[2]) v = hi;
        out[i] = v;
    }
}
}  // namespace

extern "C" int dm_image_moments(const float* a, const float* b, double* out, int n_images, int64_t n_per_image, dm_stream_t s) {
    DM_CHECK_ARG(a && b && out && n_images > 0 && n_per_image > 0, "dm_image_moments: bad arguments");
    hipLaunchKernelGGL(image_moments_kernel, dim3(n_images), dim3(256), 0, ST, a, b, out, n_per_image);
    DM_LAUNCH_CHECK();
    return DM_OK;
}

extern "C" int dm_attn_mask(const int32_t* boxes, float* out, int B, int S, float lo, float mid, float hi, dm_stream_t s) {
    DM_CHECK_ARG(boxes && out && B > 0 && S > 0, "dm_attn_mask: bad arguments");
    hipLaunchKernelGGL(attn_mask_kernel, dim3(grid_for((int64_t)B * S * S, 256)), dim3(256), 0, ST, boxes, out, B, S, lo, mid, hi);
    DM_LAUNCH_CHECK();
    return DM_OK;
}
