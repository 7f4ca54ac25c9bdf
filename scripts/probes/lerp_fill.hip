// Probe (r04): what would it cost the 3x3 halo kernel to fill a halo chunk THROUGH REGISTERS with the align-corners
// bilinear x2 upsample of a low-resolution source (new_scripy.py:242-243) instead of by LDS-DMA from a materialised upsampled tensor?
// One 512-thread workgroup per CU fills the halo of a 64-channel chunk of a 4 x 64-pixel tile — 54 pieces of 8 pixels x 128 B, the lane
// mapping of the kernel's DMA (piece = wave + 8 i, lane -> pixel lane >> 3, 16-byte slot lane & 7) — per halo pixel: four 16-byte
// loads of the low-resolution neighbours, 8 channels lerped in fp32 (bf16 -> f32, two horizontal + one vertical lerp, f32 -> bf16),
// one ds_write_b128 into the swizzled slot.  Reports microseconds per chunk per workgroup with every CU busy; the halo kernel spends
// 9 k-steps ~ 8 us of MFMA time on a chunk (64x64, 128 -> 128: 88 us for 4 tiles x 2 chunks + epilogues).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ inline float bf_lo(unsigned v) { return __uint_as_float(v << 16); }
__device__ inline float bf_hi(unsigned v) { return __uint_as_float(v & 0xffff0000u); }
__device__ inline unsigned pack_bf(float a, float b) {          // round-to-nearest-even, as v_cvt_pk_bf16_f32
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__global__ __launch_bounds__(512) void lerp_fill(const u32x4* __restrict__ src, int Hl, int Wl, int C, int nchunks, int ntiles, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];       // 2 x 54 KiB halo buffers
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = 2 * Hl, W = 2 * Wl;
    const float ry = (float)(Hl - 1) / (float)(H - 1), rx = (float)(Wl - 1) / (float)(W - 1);
    unsigned acc = 0;
    for (int t = 0; t < ntiles; ++t) {
        const int tile = blockIdx.x * ntiles + t;
        const int b = tile / (H / 4), y0 = (tile % (H / 4)) * 4;
        for (int ch = 0; ch < nchunks; ++ch) {
            char* buf = smem + ((t * nchunks + ch) & 1) * 54 * 1024;
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const int q = min(wave + 8 * i, 53);
                const int hp = q * 8 + (lane >> 3);
                const int hy = hp / 72, hx = hp - hy * 72;
                const int Y = y0 + hy - 1, X = hx - 1;
                u32x4 o = {0u, 0u, 0u, 0u};
                if ((unsigned)Y < (unsigned)H && (unsigned)X < (unsigned)W) {
                    const float sy = Y * ry, sx = X * rx;
                    const int iy = (int)sy, ix = (int)sx;
                    const float wy = sy - iy, wx = sx - ix;
                    const int iy1 = min(iy + 1, Hl - 1), ix1 = min(ix + 1, Wl - 1);
                    const size_t base = ((size_t)b * Hl * Wl) * (C / 8) + ch * 8 + (lane & 7);
                    const u32x4 a00 = src[base + ((size_t)iy * Wl + ix) * (C / 8)], a01 = src[base + ((size_t)iy * Wl + ix1) * (C / 8)];
                    const u32x4 a10 = src[base + ((size_t)iy1 * Wl + ix) * (C / 8)], a11 = src[base + ((size_t)iy1 * Wl + ix1) * (C / 8)];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float t0 = bf_lo(a00[k]) + wx * (bf_lo(a01[k]) - bf_lo(a00[k])), t1 = bf_hi(a00[k]) + wx * (bf_hi(a01[k]) - bf_hi(a00[k]));
                        const float u0 = bf_lo(a10[k]) + wx * (bf_lo(a11[k]) - bf_lo(a10[k])), u1 = bf_hi(a10[k]) + wx * (bf_hi(a11[k]) - bf_hi(a10[k]));
                        o[k] = pack_bf(t0 + wy * (u0 - t0), t1 + wy * (u1 - t1));
                    }
                }
                *(u32x4*)(buf + q * 1024 + (lane >> 3) * 128 + (((lane & 7) ^ (hp & 7)) << 4)) = o;
            }
            __syncthreads();
            acc += *(const unsigned*)(buf + ((tid * 68) & (54 * 1024 - 4)));      // keep the stores alive
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
int main() {
    const int B = 64, Hl = 32, Wl = 32, C = 256, nchunks = C / 64;        // the 64x64 level of the sampler's UnetUp: 256 channels at 32x32
    const size_t n = (size_t)B * Hl * Wl * C / 8;
    std::vector<u32x4> h(n);
    unsigned s = 12345;
    for (size_t i = 0; i < n; ++i) for (int k = 0; k < 4; ++k) { s = s * 1664525u + 1013904223u; h[i][k] = (s & 0x3fff3fffu) | 0x3c003c00u; }
    u32x4* d; unsigned* sink;
    hipMalloc(&d, n * 16); hipMalloc(&sink, 4);
    hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)lerp_fill, hipFuncAttributeMaxDynamicSharedMemorySize, 108 * 1024);
    const int ntiles_total = B * (2 * Hl) / 4, ncu = 256, per = ntiles_total / ncu;   // 1024 tiles of 4 rows x 64 pixels: 4 per CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int it = 0; it < 20; ++it) lerp_fill<<<ncu, 512, 108 * 1024>>>(d, Hl, Wl, C, nchunks, per, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("lerp halo fill: %.1f us per launch (4 tiles x %d chunks per CU) = %.2f us per chunk per workgroup\n", ms * 1e3 / 20, nchunks, ms * 1e3 / 20 / (per * nchunks));
    }
    return 0;
}
