// Probe: what does an out-of-range lane of `buffer_load_dwordx4 ... lds` leave in LDS on gfx950 — zeros or the old bytes?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr;
__global__ void k(const unsigned* a, unsigned* o, int bytes) {
    __shared__ __attribute__((aligned(16))) unsigned smem[256];
    const int lane = threadIdx.x;
    for (int i = 0; i < 4; ++i) smem[lane * 4 + i] = 0xDEAD0000u + lane;      // poison
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, bytes, 0x00020000);
    unsigned off = lane * 16;
    if (lane & 1) off = 0x7FFFFFF0u;                                          // far out of range
    if (lane == 2) off = bytes - 8;                                           // straddles the end
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)smem, 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) o[lane * 4 + i] = smem[lane * 4 + i];
}
int main() {
    const int n = 256;
    std::vector<unsigned> h(n);
    for (int i = 0; i < n; ++i) h[i] = 0x1000 + i;
    unsigned *a, *o;
    hipMalloc(&a, n * 4); hipMalloc(&o, n * 4);
    hipMemcpy(a, h.data(), n * 4, hipMemcpyHostToDevice);
    k<<<1, 64>>>(a, o, n * 4);
    std::vector<unsigned> r(n);
    hipMemcpy(r.data(), o, n * 4, hipMemcpyDeviceToHost);
    for (int lane = 0; lane < 6; ++lane) printf("lane %d: %08x %08x %08x %08x\n", lane, r[lane * 4], r[lane * 4 + 1], r[lane * 4 + 2], r[lane * 4 + 3]);
    int zeros = 0, stale = 0;
    for (int lane = 1; lane < 64; lane += 2) for (int i = 0; i < 4; ++i) { zeros += r[lane * 4 + i] == 0; stale += (r[lane * 4 + i] >> 16) == 0xDEAD; }
    printf("odd (out-of-range) lanes: %d words zero, %d words stale of %d\n", zeros, stale, 32 * 4);
    return 0;
}
