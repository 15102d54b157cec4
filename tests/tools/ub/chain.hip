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

int main()
{
    hipStream_t a, b;
    int *buf, *hb;
    const int n = 1 << 20;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    CK(hipMalloc(&buf, n * 4)); CK(hipMalloc(&hb, 4096)); CK(hipMemset(buf, 0, n * 4));
    const long long ms20 = 20LL * 100000;                  // 20 ms of the 100 MHz wall clock
    int *dirty = nullptr;
    const long long dirty_dw = 1LL << 28;                   // 1 GB
    for (int with_stores = 0; with_stores < 2; ++with_stores) {
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
    // ---- the chunk pipelines' situation: two hogs side by side, each a stream of 0.4 ms blocks twice as many as fit (two chunks' DPs
    // abreast), and FOUR chains of tiny dependent kernels (with s_setprio) on four more streams at once
    {
        hipStream_t hs[2], cs[4];
        for (auto &x : hs) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
        for (auto &x : cs) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
        g_prio = 1;
        for (int mode = 0; mode < 3; ++mode) {
            // mode 0: no hog; 1: hogs as one block per work item (40 x the slots); 2: hogs as PERSISTENT grids (exactly the slots, each wave working 40 items' time)
            if (mode == 1) for (auto &x : hs) hipLaunchKernelGGL(hog, dim3(256 * 10 * 40), dim3(64), 8 * 1024, x, hb, ms20 / 50, 2048, dirty, dirty_dw);
            if (mode == 2) for (auto &x : hs) hipLaunchKernelGGL(hog, dim3(256 * 10), dim3(64), 8 * 1024, x, hb, ms20 * 40 / 50, 2048, dirty, dirty_dw);
            auto t0 = std::chrono::steady_clock::now();
            const int reps = 100;
            for (int r = 0; r < reps; ++r) for (auto &x : cs) hipLaunchKernelGGL((tiny<8, 1>), dim3(64), dim3(64), 0, x, buf, n);
            for (auto &x : cs) CK(hipStreamSynchronize(x));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
            for (auto &x : hs) CK(hipStreamSynchronize(x));
            printf("four chains of tiny kernels at once, %s: %.1f us per kernel of a chain\n",
                   mode == 0 ? "GPU idle" : mode == 1 ? "beside two hogs of 40 x the slots' blocks (0.4 ms each)" : "beside two PERSISTENT hogs (a block per slot)", us);
        }
    }
    return 0;
}
