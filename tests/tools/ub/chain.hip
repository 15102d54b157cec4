// micro-benchmark: what a chain of small dependent kernels costs on one stream while another stream keeps the GPU full of long-lived
// waves (the situation of a chunk's front -- copy, expansion, plan, scans -- beside the DPs of the chunks before it).
//   hipcc -O3 --offload-arch=gfx950 chain.hip -o chain && ./chain
// Prints microseconds per kernel of the chain: GPU idle / beside a hog that leaves room (4 waves per SIMD, 6 KB of LDS each) / beside
// one that takes all of a CU's LDS (5 x 8 KB per SIMD) / the same with the hog's blocks arriving as a stream of short-lived waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// (88 live registers: with the loop's own ~96 VGPRs, as k_dp_row -- five waves take 480 of a SIMD's 512)
__global__ __launch_bounds__(64) void hog(int *out, long long cycles, int lds_dw, int *dirty, long long dirty_dw)
{
    extern __shared__ int sh[];
    int v[88];
#pragma unroll
    for (int i = 0; i < 88; ++i) v[i] = threadIdx.x * (i + 3) + i;
    if (lds_dw) sh[threadIdx.x % lds_dw] = v[0];
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 88; ++i) v[i] = max(max(v[i], v[(i + 1) % 88]), v[(i + 7) % 88]) + u;
        if (dirty) {                                         // a 256-byte row per wave per round, as the DP's traceback stores
            static_assert(true, "");
            const long long at = (((long long)blockIdx.x * 977 + (wall_clock64() & 0xffff)) * 64 + threadIdx.x) % dirty_dw;
            dirty[at] = v[3];
        }
    }
    int s = 0;
#pragma unroll
    for (int i = 0; i < 88; ++i) s += v[i];
    if (s == 12345 && lds_dw) out[0] = sh[0];
}
template <int VG, int PRIO>
__global__ __launch_bounds__(64) void tiny(int *out, int n)
{
    if (PRIO) __builtin_amdgcn_s_setprio(3);
    int v[VG];
#pragma unroll
    for (int i = 0; i < VG; ++i) v[i] = out[(threadIdx.x + i * 64) % n];
    int s = 0;
#pragma unroll
    for (int i = 0; i < VG; ++i) s += v[i] * (i + 1);
    out[(blockIdx.x * 64 + threadIdx.x) % n] = s;
}

static int g_prio;
static double chain_us(hipStream_t s, int *buf, int n, int blocks, int reps, int vg)
{
    CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) {
        if (vg <= 8) { if (g_prio) hipLaunchKernelGGL((tiny<8, 1>), dim3(blocks), dim3(64), 0, s, buf, n); else hipLaunchKernelGGL((tiny<8, 0>), dim3(blocks), dim3(64), 0, s, buf, n); }
        else { if (g_prio) hipLaunchKernelGGL((tiny<40, 1>), dim3(blocks), dim3(64), 0, s, buf, n); else hipLaunchKernelGGL((tiny<40, 0>), dim3(blocks), dim3(64), 0, s, buf, n); }
    }
    CK(hipStreamSynchronize(s));
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
}

int main(int argc, char **argv)
{
    const bool only_multi = argc > 1;
    hipStream_t a, b;
    int *buf, *hb;
    const int n = 1 << 20;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    CK(hipMalloc(&buf, n * 4)); CK(hipMalloc(&hb, 4096)); CK(hipMemset(buf, 0, n * 4));
    const long long ms20 = 20LL * 100000;                  // 20 ms of the 100 MHz wall clock
    int *dirty = nullptr;
    if (only_multi) CK(hipMalloc(&dirty, (1LL << 28) * 4));
    const long long dirty_dw = 1LL << 28;                   // 1 GB
    for (int with_stores = 0; with_stores < 2 && !only_multi; ++with_stores) {
    if (with_stores) { CK(hipMalloc(&dirty, dirty_dw * 4)); printf("-- the hog now also stores a 256-byte row per wave per round into 1 GB\n"); }
    for (g_prio = 0; g_prio < 2; ++g_prio)
    for (int vg : { 8, 40 })
        for (int blocks : { 1, 64, 4096 }) {
            const double idle = chain_us(b, buf, n, blocks, 200, vg);
            // hog A: 4 waves per SIMD (16 per CU), 6 KB LDS each: a slot, LDS and VGPRs left on every SIMD
            hipLaunchKernelGGL(hog, dim3(256 * 16), dim3(64), 6 * 1024, a, hb, ms20, 1536, dirty, dirty_dw);
            const double roomy = chain_us(b, buf, n, blocks, 200, vg);
            CK(hipStreamSynchronize(a));
            // hog B: 5 waves per SIMD, 8 KB each: all of the LDS
            hipLaunchKernelGGL(hog, dim3(256 * 20), dim3(64), 8 * 1024, a, hb, ms20, 2048, dirty, dirty_dw);
            const double full = chain_us(b, buf, n, blocks, 200, vg);
            CK(hipStreamSynchronize(a));
            // hog C: the same slots, but as twice as many blocks as fit, each living ~0.4 ms (waves retire and are replaced all the time)
            hipLaunchKernelGGL(hog, dim3(256 * 20 * 40), dim3(64), 8 * 1024, a, hb, ms20 / 50, 2048, dirty, dirty_dw);
            const double churn = chain_us(b, buf, n, blocks, 200, vg);
            CK(hipStreamSynchronize(a));
            printf("%s tiny kernel of %4d blocks, %2d VGPR loads: %6.1f us idle, %6.1f beside 4 waves/SIMD, %6.1f beside 5 waves/SIMD + all LDS, %6.1f beside the same as a stream of 0.4 ms waves\n",
                   g_prio ? "s_setprio 3:" : "            ", blocks, vg, idle, roomy, full, churn);
        }
    }
    // ---- the chunk pipelines' situation: NH hogs side by side (streams of their own) and NC chains of tiny dependent kernels (with
    // s_setprio) on NC more streams at once.  argv: NH NC waves-per-CU-per-hog persistent(0/1) tiny-blocks
    {
        const int NH = argc > 1 ? atoi(argv[1]) : 2, NC = argc > 2 ? atoi(argv[2]) : 4, wpc = argc > 3 ? atoi(argv[3]) : 10;
        const int persistent = argc > 4 ? atoi(argv[4]) : 1, tb = argc > 5 ? atoi(argv[5]) : 64;
        hipStream_t hs[8], cs[8];
        for (int i = 0; i < NH; ++i) CK(hipStreamCreateWithFlags(&hs[i], hipStreamNonBlocking));
        for (int i = 0; i < NC; ++i) CK(hipStreamCreateWithFlags(&cs[i], hipStreamNonBlocking));
        g_prio = 1;
        for (int mode = 0; mode < 2; ++mode) {
            if (mode == 1) for (int i = 0; i < NH; ++i) {
                if (persistent) hipLaunchKernelGGL(hog, dim3(256 * wpc), dim3(64), 8 * 1024, hs[i], hb, ms20 * 40 / 50, 2048, dirty, dirty_dw);
                else hipLaunchKernelGGL(hog, dim3(256 * wpc * 40), dim3(64), 8 * 1024, hs[i], hb, ms20 / 50, 2048, dirty, dirty_dw);
            }
            auto t0 = std::chrono::steady_clock::now();
            const int reps = 100;
            for (int r = 0; r < reps; ++r) for (int i = 0; i < NC; ++i) hipLaunchKernelGGL((tiny<8, 1>), dim3(tb), dim3(64), 0, cs[i], buf, n);
            double each[8];
            for (int i = 0; i < NC; ++i) { CK(hipStreamSynchronize(cs[i])); each[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps; }
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
            if (mode == 1) { printf("   per chain (in the order the streams were created; a chain cannot be seen to end before the ones before it):"); for (int i = 0; i < NC; ++i) printf(" %.1f", each[i]); printf("\n"); }
            for (int i = 0; i < NH; ++i) CK(hipStreamSynchronize(hs[i]));
            printf("%d chains of tiny kernels (%d blocks) at once, %s: %.1f us per kernel of a chain\n", NC, tb,
                   mode == 0 ? "GPU idle" : persistent ? "beside PERSISTENT hogs" : "beside hogs of 40 x the slots' blocks", us);
        }
        printf("   (%d hogs of %d waves per CU each)\n", NH, wpc);
    }
    return 0;
}
