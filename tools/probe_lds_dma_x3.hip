// Where does global_load_lds_dwordx3 put a lane's 12 bytes?  (dev probe: hipcc --offload-arch=gfx950 -O2 -o probe tools/probe_lds_dma_x3.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned char *p, unsigned *o)
{
    __shared__ unsigned l[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) l[i] = 0xDEAD0000u + i;
    __syncthreads();
    unsigned off = threadIdx.x * 12;
    unsigned lds = (unsigned)(uintptr_t)&l[0];
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx3 %0, %1" ::"v"(off), "s"(p), "s"(lds) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) o[i] = l[i];
}
int main()
{
    std::vector<unsigned> h(64 * 3);
    for (int i = 0; i < 64 * 3; ++i) h[i] = (i / 3) * 256 + (i % 3); // lane * 256 + dword
    unsigned char *d; unsigned *o;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o, 4096);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    std::vector<unsigned> r(1024);
    hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
    for (int i = 0; i < 260; ++i) printf("%s%08x", i % 8 ? " " : "\n", r[i]);
    printf("\n");
    return 0;
}
