// Micro-benchmark: issue cost of the integer / bit-field vector instructions the decode loop is made of,
// on gfx950, 4 waves per SIMD, four independent chains per wave.  Reports cycles per wave-instruction per
// SIMD relative to the measured time (clock taken as 2.4 GHz; compare rows, not absolute values).
// Build: hipcc -O3 --offload-arch=gfx950 op_rates.hip -o op_rates
#include <hip/hip_runtime.h>
#include <cstdio>

#define KERNEL(NAME, ASM)                                                                       \
    __global__ void NAME(unsigned *out, unsigned iters)                                         \
    {                                                                                           \
        unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = threadIdx.x | 1, c = 5; \
        unsigned long long w0 = a0, w1 = a1, w2 = a2, w3 = a3;                                   \
        __shared__ unsigned lds[1024];                                                          \
        lds[threadIdx.x & 1023] = a0;                                                           \
        unsigned addr = (threadIdx.x & 255) * 4;                                                \
        (void)w0; (void)w1; (void)w2; (void)w3; (void)addr; (void)lds;                          \
        for (unsigned it = 0; it < iters; ++it) {                                               \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) { ASM(a0, w0) ASM(a1, w1) ASM(a2, w2) ASM(a3, w3) } \
        }                                                                                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (unsigned)(w0 + w1 + w2 + w3); \
    }

#define OP_ADD(a, w) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
#define OP_MAD24(a, w) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));
#define OP_MULLO(a, w) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(b));
#define OP_MADU64(a, w) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w) : "v"(b), "v"(c) : "vcc");
#define OP_LSHL(a, w) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(a) : "v"(c));
#define OP_LSHL64(a, w) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(w) : "v"(c));
#define OP_LSHR64(a, w) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(w) : "v"(c));
#define OP_AND(a, w) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(b));
#define OP_CNDMASK(a, w) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");
#define OP_CNDMASK64(a, w) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(a) : "v"(b) : "s10", "s11");
#define OP_BFEU(a, w) asm volatile("v_bfe_u32 %0, %0, %1, %2" : "+v"(a) : "v"(c), "v"(b));
#define OP_BFEI(a, w) asm volatile("v_bfe_i32 %0, %0, 0, 1" : "+v"(a));
#define OP_FFBH(a, w) asm volatile("v_ffbh_u32 %0, %0" : "+v"(a));
#define OP_XAD(a, w) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define OP_ADD3(a, w) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define OP_ALIGNBIT(a, w) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define OP_LSHLOR(a, w) asm volatile("v_lshl_or_b32 %0, %0, %1, %2" : "+v"(a) : "v"(c), "v"(b));
#define OP_LSHLADD(a, w) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a) : "v"(b));
#define OP_ASHR(a, w) asm volatile("v_ashrrev_i32 %0, %1, %0" : "+v"(a) : "v"(c));
#define OP_MAX(a, w) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a) : "v"(b));
#define OP_CMP(a, w) asm volatile("v_cmp_lt_u32_e64 s[10:11], %0, %1" : : "v"(a), "v"(b) : "s10", "s11");
#define OP_DPP(a, w) asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a));
#define OP_DOT2(a, w) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));
#define OP_PERM(a, w) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define OP_DSREAD(a, w) asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(a) : "v"(addr) : "memory");
#define OP_DSREAD_NW(a, w) asm volatile("ds_read_b32 %0, %1" : "=v"(a) : "v"(addr) : "memory");
#define OP_DSWRITE(a, w) asm volatile("ds_write_b32 %0, %1" : : "v"(addr), "v"(a) : "memory");
#define OP_SDWA(a, w) asm volatile("v_cmp_eq_u32_sdwa s[10:11], %0, %1 src0_sel:WORD_1 src1_sel:DWORD" : : "v"(a), "v"(b) : "s10", "s11");
#define OP_ADDC(a, w) asm volatile("v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(a) : : "vcc");
#define OP_SUBREV(a, w) asm volatile("v_subrev_u32 %0, %1, %0" : "+v"(a) : "v"(b));
#define OP_OR3(a, w) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define OP_ANDOR(a, w) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define OP_SNOP(a, w) asm volatile("s_nop 0");
#define OP_SADD(a, w) asm volatile("s_add_u32 s10, s10, 1" : : : "s10");

#define ALL(X) X(ADD) X(MAD24) X(MULLO) X(MADU64) X(LSHL) X(LSHL64) X(LSHR64) X(AND) X(CNDMASK) X(CNDMASK64) X(BFEU) X(BFEI) \
    X(FFBH) X(XAD) X(ADD3) X(ALIGNBIT) X(LSHLOR) X(LSHLADD) X(ASHR) X(MAX) X(CMP) X(DPP) X(DOT2) X(PERM) X(DSREAD) X(DSREAD_NW) \
    X(DSWRITE) X(SDWA) X(ADDC) X(SUBREV) X(OR3) X(ANDOR) X(SNOP) X(SADD)

#define DEF(N) KERNEL(k_##N, OP_##N)
ALL(DEF)

typedef void (*kern_t)(unsigned *, unsigned);
int main()
{
    unsigned *d; hipMalloc(&d, 256 * 1024 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct { const char *name; kern_t k; } tests[] = {
#define ROW(N) {#N, k_##N},
        ALL(ROW)
    };
    for (int waves : {1, 4})
        for (auto &t : tests) {
            const unsigned iters = 4000;
            const int threads = 64 * 4 * waves;
            t.k<<<256, threads>>>(d, 10);
            hipEventRecord(e0);
            t.k<<<256, threads>>>(d, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double instr_per_simd = (double)iters * 64 * waves;
            printf("%-10s waves/SIMD=%d  %.3f ms  %.2f cycles per wave-instruction per SIMD\n", t.name, waves, ms,
                   ms * 1e-3 * 2.4e9 / instr_per_simd);
        }
    return 0;
}
