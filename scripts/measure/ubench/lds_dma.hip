// Where does `buffer_load_dword{,x3,x4} ... lds` put a lane's data on gfx950?  Prints, per size, the LDS word index at which lane l's
// first dword landed (source dword = 1000 l + j).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void_t;
template <int SIZE>
__global__ void k(const unsigned* p, unsigned* o, int n) {
    __shared__ __attribute__((aligned(16))) unsigned buf[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) buf[i] = 0xdeadbeefu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(p), 0, n * 4, 0x00020000);
    if (SIZE == 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)buf, 4, (int)(threadIdx.x * 64), 0, 0, 0);
    if (SIZE == 12) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)buf, 12, (int)(threadIdx.x * 64), 0, 0, 0);
    if (SIZE == 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)buf, 16, (int)(threadIdx.x * 64), 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) o[i] = buf[i];
}
int main() {
    const int n = 64 * 16;
    std::vector<unsigned> h(n);
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 16; ++j) h[l * 16 + j] = 1000 * l + j;
    unsigned *d, *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, 2048 * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    std::vector<unsigned> r(2048);
    for (int sz : {4, 12, 16}) {
        if (sz == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, d, o, n);
        if (sz == 12) hipLaunchKernelGGL(k<12>, dim3(1), dim3(64), 0, 0, d, o, n);
        if (sz == 16) hipLaunchKernelGGL(k<16>, dim3(1), dim3(64), 0, 0, d, o, n);
        hipMemcpy(r.data(), o, 2048 * 4, hipMemcpyDeviceToHost);
        printf("size %d: first 24 LDS words:", sz);
        for (int i = 0; i < 24; ++i) printf(" %u", r[i]);
        printf("\n  lane 1 dword 0 (1000) at word:");
        for (int i = 0; i < 2048; ++i) if (r[i] == 1000) printf(" %d", i);
        printf("; lane 1 dword 1 (1001) at:");
        for (int i = 0; i < 2048; ++i) if (r[i] == 1001) printf(" %d", i);
        printf("; lane 63 dword 0 at:");
        for (int i = 0; i < 2048; ++i) if (r[i] == 63000) printf(" %d", i);
        printf("\n");
    }
    return 0;
}
