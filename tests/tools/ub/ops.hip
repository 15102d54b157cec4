// per-instruction VALU throughput on MI355X: 1 / 2 / 8 waves per SIMD, 8 independent chains, inline asm.
// Each line: wall time per wave-instruction per SIMD (HIP events), the in-kernel shader clock over the run
// (s_memtime ticks / s_memrealtime ticks x 100 MHz, MI355X_MICROARCH.md "DVFS give-back" item 6) and hence
// SHADER CYCLES per wave-instruction per SIMD -- the unit DESIGN.md's VALU-issue roofline is priced in.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X(a,b,c) X(b,c,d) X(c,d,e) X(d,e,f) X(e,f,g) X(f,g,h) X(g,h,a) X(h,a,b)
#define DEFK(NAME, ASM) \
__global__ __launch_bounds__(64) void k_##NAME(int *out, int iters, unsigned long long *clk) { \
    int a = threadIdx.x, b = a * 3 + 1, c = a ^ 5, d = a + 7, e = 1, f = 2, g = 3, h = 4; \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
    for (int i = 0; i < iters; ++i) { \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) { REP8(ASM) } } \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; } \
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + e + f + g + h; }
#define A_ADD(x,y,z)  asm volatile("v_add_u32 %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
#define A_SUB(x,y,z)  asm volatile("v_sub_u32 %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
#define A_AND(x,y,z)  asm volatile("v_and_b32 %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
#define A_MAX(x,y,z)  asm volatile("v_max_i32 %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
#define A_MAX3(x,y,z) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
#define A_ANDOR(x,y,z) asm volatile("v_and_or_b32 %0, %1, -4, %2" : "+v"(x) : "v"(y), "v"(z));
#define A_ADD3(x,y,z) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
#define A_LSHLADD(x,y,z) asm volatile("v_lshl_add_u32 %0, %1, 2, %2" : "+v"(x) : "v"(y), "v"(z));
#define A_ALIGN(x,y,z) asm volatile("v_alignbit_b32 %0, %1, %2, 2" : "+v"(x) : "v"(y), "v"(z));
#define A_DOT2(x,y,z) asm volatile("v_dot2c_i32_i16 %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
#define A_DOT4(x,y,z) asm volatile("v_dot4c_i32_i8 %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
#define A_DOT2V3(x,y,z) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(z));
#define A_MAD24(x,y,z) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(z));
#define A_MULLO(x,y,z) asm volatile("v_mul_lo_u32 %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
#define A_MUL24(x,y,z) asm volatile("v_mul_i32_i24 %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
#define A_MOVDPP(x,y,z) asm volatile("v_mov_b32_dpp %0, %1 wave_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(y));
#define A_MAXDPP(x,y,z) asm volatile("v_max_i32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(y), "v"(z));
#define A_ADDDPP(x,y,z) asm volatile("v_add_u32_dpp %0, %1, %2 wave_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(y), "v"(z));
#define A_CND(x,y,z)  asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "+v"(x) : "v"(y), "v"(z));
#define A_CMP(x,y,z)  asm volatile("v_cmp_gt_i32 vcc, %1, %2" : "+v"(x) : "v"(y), "v"(z) : "vcc");
#define A_CMPE64(x,y,z)  asm volatile("v_cmp_gt_i32 s[10:11], %1, %2" : "+v"(x) : "v"(y), "v"(z) : "s10", "s11");
#define A_PKADD(x,y,z) asm volatile("v_pk_add_i16 %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
#define A_PKMAX(x,y,z) asm volatile("v_pk_max_i16 %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
#define A_PKMAD(x,y,z) asm volatile("v_pk_mad_i16 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(z));
#define A_PERM(x,y,z) asm volatile("v_perm_b32 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(z));
#define A_BFE(x,y,z) asm volatile("v_bfe_u32 %0, %1, 2, 6" : "+v"(x) : "v"(y));
#define A_READLANE(x,y,z) asm volatile("v_readlane_b32 s10, %1, 63\n v_add_u32 %0, s10, %0" : "+v"(x) : "v"(y) : "s10");
#define A_SADD(x,y,z) asm volatile("s_add_u32 s10, s10, 1" ::: "s10");
#define A_SNOP(x,y,z) asm volatile("s_nop 1");
#define A_LDS(x,y,z) asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(y));
#define A_XOR3(x,y,z) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(x) : "v"(y), "v"(z));
#define LIST(X) X(ADD) X(SUB) X(AND) X(MAX) X(MAX3) X(ANDOR) X(ADD3) X(LSHLADD) X(ALIGN) X(DOT2) X(DOT4) X(DOT2V3) X(MAD24) X(MULLO) X(MUL24) \
    X(MOVDPP) X(MAXDPP) X(ADDDPP) X(CND) X(CMP) X(CMPE64) X(PKADD) X(PKMAX) X(PKMAD) X(PERM) X(BFE) X(READLANE) X(SADD) X(SNOP)
#define MK(N) DEFK(N, A_##N)
LIST(MK)
template <typename K> void run(const char *name, K kern, int blocks, int iters)
{
    int *out; hipMalloc(&out, blocks * 64 * 4);
    unsigned long long *clk, hclk[2]; hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, 10, clk); hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost);
    double instr = (double)iters * 64 * (blocks / 1024.0);     // wave-instructions per SIMD
    double ghz = hclk[1] ? (double)hclk[0] / (double)hclk[1] * 0.1 : 0.0;
    printf("%-9s %5.1f w/SIMD: %7.3f ns per wave-instr per SIMD, shader clock %.2f GHz -> %.2f cycles per wave-instr\n",
           name, blocks / 1024.0, ms * 1e6 / instr, ghz, ms * 1e6 / instr * ghz);
    hipFree(out); hipFree(clk);
}
int main()
{
    for (int blocks : {1024, 2048, 8192}) {
#define RUN(N) run(#N, k_##N, blocks, 20000);
        LIST(RUN)
        printf("\n");
    }
    return 0;
}
