// Micro-benchmark: integer VALU issue rate per SIMD on gfx950 at 1, 2, 4, 8 waves per SIMD,
// with dependent and independent instruction streams.  Build: hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP>
__global__ void k(unsigned *out, unsigned iters)
{
    unsigned a[ILP];
    for (int i = 0; i < ILP; ++i) a[i] = threadIdx.x + i;
    for (unsigned it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < ILP; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(it));
    }
    unsigned s = 0;
    for (int i = 0; i < ILP; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP>
void run(int waves_per_simd)
{
    unsigned *d; hipMalloc(&d, 256 * 1024 * 64 * 4);
    const unsigned iters = 20000;
    const int threads = 64 * 4 * waves_per_simd;   // one block per CU, waves spread over 4 SIMDs
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<ILP><<<256, threads>>>(d, 10);
    hipEventRecord(e0);
    k<ILP><<<256, threads>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 16 * ILP * waves_per_simd;
    printf("ILP=%d waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", ILP, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / instr_per_simd);
    hipFree(d);
}
int main()
{
    for (int w : {1, 2, 4, 8}) { run<1>(w); run<4>(w); }
    return 0;
}
