// micro-benchmark: VALU issue rate per SIMD and effective clock on MI355X
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef short short2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int D2(int a, int b, int acc) { return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, a), __builtin_bit_cast(short2_t, b), acc, false); }
template <int KIND>
__global__ __launch_bounds__(64) void k(int *out, int iters, long long *cyc)
{
    int a = threadIdx.x, b = a * 3 + 1, c = a ^ 5, d = a + 7, e = 1, f = 2, g = 3, h = 4;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) {        // plain adds (8 independent chains)
                a += b; b += c; c += d; d += e; e += f; f += g; g += h; h += a;
            } else if (KIND == 1) { // dot2c
                a = D2(b, c, a); b = D2(c, d, b);
                c = D2(d, e, c); d = D2(e, f, d);
                e = D2(f, g, e); f = D2(g, h, f);
                g = D2(h, a, g); h = D2(a, b, h);
            } else if (KIND == 2) { // max3
                a = max(max(a, b), c); b = max(max(b, c), d); c = max(max(c, d), e); d = max(max(d, e), f);
                e = max(max(e, f), g); f = max(max(f, g), h); g = max(max(g, h), a); h = max(max(h, a), b);
            } else if (KIND == 3) { // dpp mov + add
                a += __builtin_amdgcn_mov_dpp(b, 0x13C, 0xF, 0xF, false); b += __builtin_amdgcn_mov_dpp(c, 0x13C, 0xF, 0xF, false);
                c += __builtin_amdgcn_mov_dpp(d, 0x13C, 0xF, 0xF, false); d += __builtin_amdgcn_mov_dpp(e, 0x13C, 0xF, 0xF, false);
                e += __builtin_amdgcn_mov_dpp(f, 0x13C, 0xF, 0xF, false); f += __builtin_amdgcn_mov_dpp(g, 0x13C, 0xF, 0xF, false);
                g += __builtin_amdgcn_mov_dpp(h, 0x13C, 0xF, 0xF, false); h += __builtin_amdgcn_mov_dpp(a, 0x13C, 0xF, 0xF, false);
            } else if (KIND == 4) { // cndmask + alignbit mix
                a = (b > c) ? a : d; b = __builtin_amdgcn_alignbit(c, b, 2); c = (d > e) ? c : f; d = __builtin_amdgcn_alignbit(e, d, 2);
                e = (f > g) ? e : h; f = __builtin_amdgcn_alignbit(g, f, 2); g = (h > a) ? g : b; h = __builtin_amdgcn_alignbit(a, h, 2);
            }
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + e + f + g + h;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int KIND>
void run(const char *name, int blocks, int iters, int per_iter)
{
    int *out; long long *cyc, hc;
    hipMalloc(&out, blocks * 64 * 4); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, 10, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    double instr = (double)iters * per_iter;
    // waves per SIMD = blocks / 1024
    printf("%-10s blocks %6d (%.1f waves/SIMD): %.3f ms, %.2f ns per wave-instr, clock64 %lld ticks (%.2f ticks/instr), %.1f Ginstr/s/SIMD-equivalent\n",
           name, blocks, blocks / 1024.0, ms, ms * 1e6 / instr, hc, hc / instr, instr * (blocks / 1024.0) / (ms * 1e6));
    hipFree(out); hipFree(cyc);
}
int main()
{
    for (int blocks : {1024, 2048, 4096, 8192}) {
        run<0>("add", blocks, 20000, 64);
        run<1>("dot2", blocks, 20000, 64);
        run<2>("max3", blocks, 20000, 64);
        run<3>("dpp+add", blocks, 20000, 128);
        run<4>("cnd/align", blocks, 20000, 96);
    }
    return 0;
}
