// mz_device.hip -- gfx950 (MI355X / CDNA4) kernels for the multiz block-pair merge DP.
//
// What is computed: exactly the recurrence of reference mz_yama.c:83-255 (three-state
// C/D/I affine-gap sum-of-pairs DP over a band LB[]..RB[]), its traceback (mz_yama.c:257-291)
// and the merged-column emit (mz_yama.c:293-313), bit for bit, for a whole batch of
// independent block pairs.  How it is computed is not the reference's:
//
//  * per-cell sums over (row of A) x (row of B) are evaluated as small integer bilinear
//    forms of per-column class/gap counts (oracle/yama_profile_oracle.c is the executable
//    specification; integer-exact).  The counts are packed as int8/int16 vectors so that a
//    cell costs a handful of v_dot4_i32_i8 / v_dot2_i32_i16 instead of 4*K*L table look-ups.
//  * one 64-lane wave owns one block pair.  The main kernels (kernels/row.inc) are ROW-parallel:
//    a lane owns a band column (a ring over the columns), the wave computes one whole band row
//    per iteration, C and D come from the previous row (one DPP rotate for the diagonal) and the
//    along-the-row I recurrence is a DPP prefix maximum.  Bands that fit neither that form nor
//    its transposed twin because some rows are wider than 64 columns run the same scheme with the column periods
//    lagged against one another (kernels/lag.inc); what is left falls back to anti-diagonal wavefronts (kernels/wavefront_*.inc: lane =
//    DP row mod 64, step t handles cells (r, t-r)) or to 64-row strips (kernels/strip.inc).
//    Column records sit in an LDS ring, row records in an LDS block, both built in-kernel.
//  * traceback entries are 2-bit pick tags in three streams (reference bytes in the exact
//    kernels), laid out so that every store is one fully coalesced 256-byte row per wave.
//
// No MFMA anywhere: this is an integer max-plus recurrence, not a contraction.

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdlib>
#include <mutex>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "mz_device.h"

#define WAVE   64
#define SIMDS_TOTAL 1024        // 256 CUs x 4 SIMDs
#define ROLL_VMAX 64            // kernels/roll.inc: the longest wait (in steps) of a row the rolling form with late starts admits (its rings hold 2 V + 128: static_assert there)
#define MZ_ROWS_SUM(t11)     ((t11) & ((1LL << 44) - 1))      // totals[11]: rows (K+L) of all valid pairs ...
#define MZ_TSTRIP_PAIRS(t11) ((t11) >> 44)                    // ... and the number of MZ_MODE_TSTRIP pairs (kernels/plan.inc, scan_load)
#define BRING  128          // B-profile ring entries (columns) in LDS, 16 B each
#define REC_DW 16           // dwords per staged row record
// row-parallel kernel: prep layout of a pair (dwords): row records of rows 1..M+2 (dead beyond M), then (COL
// mode) the transposed band bounds
#define RREC 16             // dwords per row record in b.prep
#define RCOL 8              // dwords per column record: uA uB c01 c23 | c45 xI+P P Q
#define ROW_NROWS(M) ((M) + 2)
#define ROW_NCOLS(N) ((((N) + WAVE + WAVE - 1) / WAVE) * WAVE + WAVE)
#define COL_PREP_DWORDS(N) (2LL * ((N) + 1))           // transposed band bounds of a COL pair
// row-parallel family (lane = column / transposed: lane = row) and its transposed members
#define MODE_IS_ROWFAM(m) ((m) >= MZ_MODE_ROW && (m) <= MZ_MODE_COLR)
#define MODE_IS_COLFAM(m) ((m) == MZ_MODE_COL || (m) == MZ_MODE_COLR)
#define MODE_IS_TAGGED(m) ((m) == MZ_MODE_FASTT || MODE_IS_ROWFAM(m) || (m) == MZ_MODE_LAG || (m) == MZ_MODE_DUO)

// Helper kernels (everything but the DP) raise their waves' issue priority: in the chunk pipelines they run beside five DP waves per
// SIMD that issue VALU four cycles in five, and a helper wave that only gets what those leave takes 10-30x its time alone -- the plan of
// chunk k+2 then is not through when the DP of chunk k ends, and the DP stream idles (round 5; -DMZ_HELPER_PRIO=0 for A/B builds).
#ifndef MZ_HELPER_PRIO
#define MZ_HELPER_PRIO 3
#endif
// ... and take no more than the 32 VGPRs five row-parallel DP waves leave of a SIMD's 512 (no LDS either: those take all 160 KB of a CU), so
// that a helper wave is placed at once instead of waiting for a DP wave to retire and then taking the place of the next one
#ifndef MZ_HELPER_VGPRS
#define MZ_HELPER_VGPRS 32
#endif
#define HELPER_VGPRS
#define HELPER_PRIO() do { if (MZ_HELPER_PRIO) __builtin_amdgcn_s_setprio(MZ_HELPER_PRIO); } while (0)
// ... when their batch asks for it (MZ_DP_HELPERS_FIRST, include/mz_amd.h: the chunk pipelines do; a device-resident pipeline's helpers have a whole DP's time and
// cost it 1.5-3 % at the raised priority: C2 571 against 579 GCUPS, C4 501 / 517, c4i 399 / 408, same box, alternating)
#define HELPER_PRIO_OF(b) do { if (MZ_HELPER_PRIO && ((b).dp_hint & MZ_DP_HELPERS_FIRST)) __builtin_amdgcn_s_setprio(MZ_HELPER_PRIO); } while (0)

struct ScoreConst { int S6[36]; int go; int ge; int g1, g2; int tag_ok; int maxS; int row_on; int lag_on; int tstrip_on; };
__constant__ ScoreConst c_sc;

// byte -> class {A/a:0, C/c:1, G/g:2, T/t:3, '-':4, other:5}; the six classes on which the
// reference score table is constant (mz_scores.c:39-54)
__device__ __forceinline__ int byte_class(unsigned ch)
{
    const unsigned u = ch | 0x20u;           // fold case: only 'X' and 'x' map onto 'x'
    int c = 5;
    c = (u == 'a') ? 0 : c;
    c = (u == 'c') ? 1 : c;
    c = (u == 'g') ? 2 : c;
    c = (u == 't') ? 3 : c;
    c = (ch == '-') ? 4 : c;
    return c;
}

struct __attribute__((packed, aligned(1))) u32_any { uint32_t v; };     // a dword at any byte address (gfx950 global memory takes it)

// What the kernels need of one column of a block (n bytes at p; q: the column before, or null): the counts of A, C,
// G, T (four byte counters), of dashes and of anything else, and the rows that have a base here and before (n00) or
// a dash here and before (n11).  Four bytes at a time: a byte equals a code where the XOR with the code repeated is
// zero, (x & 0x7f..) + 0x7f.. carries into bit 7 of every non-zero byte without crossing bytes, and the counts are
// population counts of the 0x80 marks -- ~11 instructions per byte against ~33 for compare-and-select byte by byte,
// which made the staging of blocks of 20-30 rows cost as much as two thirds of the recurrence it feeds.
struct ByteCounts { unsigned acgt; int dash, other, n00, n11; };
__device__ __forceinline__ unsigned zero_bytes(unsigned x)          // 0x80 in every byte of x that is zero
{
    return ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu);
}
__device__ __forceinline__ ByteCounts count_bytes(const uint8_t *p, const uint8_t *q, int n)
{
    unsigned acgt = 0;                                               // four byte counters (n <= 255)
    int dash = 0, n00 = 0, n11 = 0;
    int i = 0;
#pragma unroll 1
    for (; i + 4 <= n; i += 4) {
        const unsigned w = ((const u32_any *)(p + i))->v;
        const unsigned u = w | 0x20202020u;                          // (case folded as byte_class does)
        const unsigned zD = zero_bytes(w ^ 0x2d2d2d2du);
        const unsigned zP = q ? zero_bytes(((const u32_any *)(q + i))->v ^ 0x2d2d2d2du) : 0u;
        acgt += __popc(zero_bytes(u ^ 0x61616161u));
        acgt += __popc(zero_bytes(u ^ 0x63636363u)) << 8;
        acgt += __popc(zero_bytes(u ^ 0x67676767u)) << 16;
        acgt += __popc(zero_bytes(u ^ 0x74747474u)) << 24;
        dash += __popc(zD);
        n00 += __popc(~(zD | zP) & 0x80808080u);
        n11 += __popc(zD & zP);
    }
    // (of the dwords, what is neither a base nor a dash: by difference)
    int other = (n & ~3) - (int)((acgt & 0xff) + ((acgt >> 8) & 0xff) + ((acgt >> 16) & 0xff) + (acgt >> 24)) - dash;
    for (; i < n; ++i) {                                             // the last n & 3 bytes, one at a time
        const unsigned ch = p[i];
        const bool d = ch == '-', pd = q ? (q[i] == '-') : false;
        const int cl = byte_class(ch);
        acgt += (cl < 4) ? (1u << (cl << 3)) : 0u;
        other += cl == 5;
        dash += d;
        n00 += (!d) & (!pd);
        n11 += d & pd;
    }
    ByteCounts r;
    r.acgt = acgt;
    r.dash = dash; r.n00 = n00; r.n11 = n11;
    r.other = other;
    return r;
}

__device__ __forceinline__ int pack4(int b0, int b1, int b2, int b3)
{
    return (b0 & 0xff) | ((b1 & 0xff) << 8) | ((b2 & 0xff) << 16) | ((b3 & 0xff) << 24);
}
__device__ __forceinline__ int pack2(int lo, int hi) { return (lo & 0xffff) | (hi << 16); }
__device__ __forceinline__ int m24(int a, int b) { return __mul24(a, b); }      // full-rate multiply of factors below 2^23

typedef short short2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int dot4(int a, int b, int acc) { return __builtin_amdgcn_sdot4(a, b, acc, false); }
__device__ __forceinline__ int dot2(int a, int b, int acc)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, a), __builtin_bit_cast(short2_t, b), acc, false);
}
// acc + a.b with the result in a new register (VOP3P form): the compiler's choice, v_dot2c, accumulates
// in place and costs a v_mov when acc must survive
__device__ __forceinline__ int dot2_keep(int a, int b, int acc)
{
    int d;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(acc));
    return d;
}
// the same with the row vector in a scalar register
__device__ __forceinline__ int dot2_keep_s(int a, int b, int acc)
{
    int d;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "s"(a), "v"(b), "v"(acc));
    return d;
}
// lane i <- lane i-1, lane 0 <- lane 63 (DPP wave_ror:1)
__device__ __forceinline__ int ror1(int v) { return __builtin_amdgcn_mov_dpp(v, 0x13C, 0xF, 0xF, false); }

__device__ __forceinline__ int wave_min(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ long long wave_sum64(long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// transposed band (the band seen column by column): rows of column c are t_lo(c) .. t_hi(c)
__device__ __forceinline__ int t_lo(const int *RB, int M, int c)        // first r with RB[r] >= c
{
    int lo = 0, hi = M;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (RB[mid] >= c) hi = mid; else lo = mid + 1; }
    return lo;
}
__device__ __forceinline__ int t_hi(const int *LB, int M, int c)        // last r with LB[r] <= c
{
    int lo = 0, hi = M;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (LB[mid] <= c) lo = mid; else hi = mid - 1; }
    return lo;
}

// The kernels that take their pairs from a shared counter (k_dp, k_dp_wide, k_dp_lag).  The plan's prefix sums leave, in
// b.packList, the indices of the pairs of each of these kernels one list after the other (wavefront / strip, blocks of
// 128+ rows, lagged; k_scan3): a wave's atomic increment IS its next pair -- no walking through the batch's indices
// (a handful of wavefront pairs in 50 000 would cost a few waves 50 000 atomics), and the launch needs no more waves
// than the list has pairs (mz_dp_grid) while the pairs still go to whichever wave is free.  A launch over a PART of
// the batch (first, count) walks the indices of its range one by one instead.
template <class Pred, class Body>
__device__ __forceinline__ void for_my_pairs(const mz_dev_batch &b, unsigned long long *next, int list_first, int list_count,
                                             int first, int count, int lane, Pred mine_is, Body run)
{
    const bool whole = first == 0 && count == b.n;
    const int limit = whole ? list_count : count;
    for (;;) {
        int k = 0;
        if (lane == 0) k = (int)atomicAdd(next, 1ULL);
        k = __builtin_amdgcn_readfirstlane(k);
        if (k >= limit) break;
        int p;
        if (whole) { p = __builtin_amdgcn_readfirstlane(b.packList[list_first + k]); if (b.status[p] != MZ_OK || !mine_is(p)) continue; }   // (k_fit may have failed it since; two kernels share the second list)
        else { p = first + k; if (!mine_is(p)) continue; }
        run(p);
        __syncthreads();                               // the next pair restages the same LDS
    }
}

// A block per pair, whole batch: block k takes the k-th entry of the plan's lists (every valid pair is on one of the four,
// most cells first within each: the blocks the dispatcher hands out last are short ones); a part of a batch: its pairs in order.
__device__ __forceinline__ int listed_pair(const mz_dev_batch &b, int first, int count, int k)     // -1: no pair for this block
{
    if (first != 0 || count != b.n) return k < count ? first + k : -1;
    return k < b.n - (int)b.totals[3] ? b.packList[k] : -1;
}

#include "kernels/plan.inc"
#include "kernels/wavefront_exact.inc"
#include "kernels/wavefront_fast.inc"
#include "kernels/row.inc"
#include "kernels/lag.inc"
#include "kernels/strip.inc"
#include "kernels/roll.inc"
#include "kernels/duo.inc"
#include "kernels/dispatch.inc"
#include "kernels/walk.inc"
#include "kernels/emit.inc"
#include "kernels/prepost.inc"
// ------------------------------------------------------------------------------------------
// C-ABI launchers
// ------------------------------------------------------------------------------------------
static thread_local char g_err[256];   // (one host thread per GPU in multi-GPU batches)
static int fail(hipError_t e, const char *what)
{
    snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
    return -1;
}
#define CK(call, what) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(e_, what); } while (0)

extern "C" const char *mzk_last_error(void) { return g_err; }

extern "C" int mzk_upload_scores(const mz_score_model *m, void *stream)
{
    ScoreConst h;
    for (int i = 0; i < 36; ++i) h.S6[i] = m->S6[i];
    h.go = m->gap_open;
    h.ge = m->gap_extend;
    h.g1 = m->g1;
    h.g2 = m->g2;
    h.maxS = 0;
    for (int i = 0; i < 36; ++i) { const int a = m->S6[i] < 0 ? -m->S6[i] : m->S6[i]; if (a > h.maxS) h.maxS = a; }
    h.row_on = m->row;
    { const char *e = getenv("MZ_NO_LAG"); h.lag_on = !(e && e[0] == '1'); }
    // MZ_NO_TSTRIP=1: wide-and-high bands stay on the exact strips (A/B measurements); MZ_TROLL=1: the rolling form with late starts
    // (kernels/roll.inc) where its rings hold the band -- off by default: measured no faster than the tagged strips
    { const char *e = getenv("MZ_NO_TSTRIP"), *r = getenv("MZ_TROLL"); const char *d = getenv("MZ_NO_DUO");      // (bit 0: tagged strips; bit 1: the rolling form; bit 2: the two-wave wavefront, kernels/duo.inc -- MZ_NO_DUO=1: A/B measurements)
      h.tstrip_on = (e && e[0] == '1') ? 0 : ((r && r[0] == '1') ? 3 : 1) | ((d && d[0] == '1') ? 0 : 4); }       // MZ_NO_TSTRIP=1: wide-and-high bands stay on the exact strips (A/B measurements)      // MZ_NO_LAG=1: bands with wide rows stay on the wavefront kernels (A/B measurements)
    h.tag_ok = (m->g1 > 0 && 2 * m->g1 * 127 <= 32767 && 2 * m->g2 * 127 <= 32767) ? 1 : 0;
    CK(hipMemcpyToSymbolAsync(HIP_SYMBOL(c_sc), &h, sizeof h, 0, hipMemcpyHostToDevice, (hipStream_t)stream), "upload scores");
    CK(hipStreamSynchronize((hipStream_t)stream), "upload scores sync");
    return 0;
}

// ------------------------------------------------------------------------------------------
// What the host path (mz_yama_batch, mz_host.c / mz_pack.c) sends instead of the pools themselves.
//
// Band bounds: LB and RB are monotone and move by a column or two per row, so instead of 8 bytes per row (two thirds
// of a C2 pair's input) the staging block carries, per pair,
//   format 2:  M bytes  (LB[i]-LB[i-1]) | (RB[i]-RB[i-1]) << 4; LB[0], RB[0] in the header     (all steps 0..15)
//   format 1:  LB[0], RB[0] (int32), then M bytes LB[i]-LB[i-1], then M bytes RB[i]-RB[i-1]   (all steps 0..255)
//   format 0:  LB[0..M], RB[0..M] as int32                                                    (anything else)
// at byte offset offC[p] (4-byte aligned) -- format 2 in the staging block itself (every pair has its slot there), the
// other two in the exception block.  One wave per pair turns that back into the int32 pools every kernel reads.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WAVE) void k_unband(int n, const int32_t *bandLen, const int32_t *lb0, const int32_t *rb0, const int64_t *offBand, const int64_t *offC,
                                                 const uint8_t *fmt, const uint8_t *packed, const uint8_t *exceptions,
                                                 int32_t *poolLB, int32_t *poolRB)
{
    HELPER_PRIO();
    const int p = blockIdx.x, lane = threadIdx.x;
    if (p >= n) return;
    const int M = bandLen[p] - 1;                       // entries 0..M (an invalid job carries one dummy entry)
    int32_t *LB = poolLB + offBand[p], *RB = poolRB + offBand[p];
    const int f = fmt[p];
    const uint8_t *src = (f == 2 ? packed : exceptions) + offC[p];
    if (f == 0) {
        const int32_t *s = (const int32_t *)src;
        for (int i = lane; i <= M; i += WAVE) { LB[i] = s[i]; RB[i] = s[M + 1 + i]; }
        return;
    }
    int baseL = f == 2 ? lb0[p] : ((const int32_t *)src)[0], baseR = f == 2 ? rb0[p] : ((const int32_t *)src)[1];
    const uint8_t *dL = f == 2 ? src : src + 8, *dR = src + 8 + M;
    if (lane == 0) { LB[0] = baseL; RB[0] = baseR; }
    for (int i0 = 1; i0 <= M; i0 += WAVE) {
        const int i = i0 + lane;
        int x = 0, y = 0;
        if (i <= M) {
            if (f == 2) { const int v = dL[i - 1]; x = v & 15; y = v >> 4; }
            else { x = dL[i - 1]; y = dR[i - 1]; }
        }
#pragma unroll
        for (int o = 1; o < WAVE; o <<= 1) {
            const int a = __shfl_up(x, o), c = __shfl_up(y, o);
            if (lane >= o) { x += a; y += c; }
        }
        if (i <= M) { LB[i] = baseL + x; RB[i] = baseR + y; }
        baseL += __builtin_amdgcn_readlane(x, WAVE - 1);
        baseR += __builtin_amdgcn_readlane(y, WAVE - 1);
    }
}

// Columns: class nibbles (mz_pack.c) back to one canonical byte per class -- A, C, G, T, '-', N: the six classes the
// score tables distinguish (mz_scores.c:39-54), so the DP sees what it would see on the caller's own bytes; the merged
// columns are assembled on the host from the caller's bytes and the edit script.  A thread per dword: 8 nibbles in,
// 8 bytes out; pair boundaries do not matter (every pair's slice is a whole number of 32-byte lines).
__global__ __launch_bounds__(256) void k_unnib(const uint32_t *src, uint2 *dst, long long ndw)
{
    HELPER_PRIO();
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= ndw) return;
    const unsigned long long tab = 0x00004e2d54474341ULL;           // "ACGT-N", a byte per class
    const uint32_t w = src[i];
    uint2 o;
    o.x = (uint32_t)((tab >> (8 * (w & 15))) & 0xff) | (uint32_t)((tab >> (8 * ((w >> 4) & 15))) & 0xff) << 8 |
          (uint32_t)((tab >> (8 * ((w >> 8) & 15))) & 0xff) << 16 | (uint32_t)((tab >> (8 * ((w >> 12) & 15))) & 0xff) << 24;
    o.y = (uint32_t)((tab >> (8 * ((w >> 16) & 15))) & 0xff) | (uint32_t)((tab >> (8 * ((w >> 20) & 15))) & 0xff) << 8 |
          (uint32_t)((tab >> (8 * ((w >> 24) & 15))) & 0xff) << 16 | (uint32_t)((tab >> (8 * (w >> 28))) & 0xff) << 24;
    dst[i] = o;
}

extern "C" int mzk_unband(int n, const int32_t *bandLen, const int32_t *lb0, const int32_t *rb0, const int64_t *offBand, const int64_t *offC, const uint8_t *fmt,
                          const uint8_t *packed, const uint8_t *exceptions, int32_t *poolLB, int32_t *poolRB, void *stream)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_unband, dim3(n), dim3(WAVE), 0, (hipStream_t)stream, n, bandLen, lb0, rb0, offBand, offC, fmt, packed, exceptions, poolLB, poolRB);
    CK(hipGetLastError(), "unband launch");
    return 0;
}

extern "C" int mzk_unnib(const void *nibbles, void *bytes, long long nbytes_out, void *stream)
{
    const long long ndw = nbytes_out / 8;               // (the caller pads every slice to 8 bytes)
    if (ndw <= 0) return 0;
    hipLaunchKernelGGL(k_unnib, dim3((unsigned)((ndw + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint32_t *)nibbles, (uint2 *)bytes, ndw);
    CK(hipGetLastError(), "unnib launch");
    return 0;
}

extern "C" int mzk_plan(const mz_dev_batch *b, void *stream)
{
    if (b->n <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    // few pairs: long pairs as likely as not -- their rows in PLAN_SEGS stretches first (kernels/plan.inc)
    const int folded = b->n <= MZ_PLAN_FOLD_MAX;
    if (folded) hipLaunchKernelGGL(k_plan_seg, dim3(b->n, PLAN_SEGS), dim3(WAVE), 0, s, *b);
    hipLaunchKernelGGL(k_plan, dim3(b->n), dim3(WAVE), 0, s, *b, folded);
    {
        const int nblk = (b->n + SCAN_B - 1) / SCAN_B;
        hipLaunchKernelGGL(k_scan1, dim3(nblk), dim3(SCAN_B), 0, s, *b);
        hipLaunchKernelGGL(k_scan2, dim3(SCAN_Q + ORD_KEYS), dim3(64), 0, s, *b, nblk);
        hipLaunchKernelGGL(k_scan3, dim3(nblk), dim3(SCAN_B), 0, s, *b);
    }
    hipLaunchKernelGGL(k_fit, dim3((b->n + 255) / 256), dim3(256), 0, s, *b);
    CK(hipGetLastError(), "plan launch");
    return 0;
}

// row / column records of the MZ_MODE_ROW pairs (needs the plan's offsets and the caller's prep buffer)
extern "C" int mzk_prep(const mz_dev_batch *b, void *stream)
{
    if (b->n <= 0) return 0;
    // few pairs: long pairs as likely as not -- their rows in 16 segments (blockIdx.y)
    hipLaunchKernelGGL(k_rowprep, dim3(b->n, b->n <= 4096 ? 16 : 1), dim3(WAVE), 0, (hipStream_t)stream, *b, 0, b->n);
    CK(hipGetLastError(), "prep launch");
    return 0;
}

// Side streams of mzk_dp_range, one set per device (created on first use; a device's calls come from one host thread)
struct DpSide { int ready; hipStream_t s[4]; hipEvent_t fork, join[4]; std::mutex mu; };
static DpSide g_side[16];
static std::mutex g_side_mu;
static DpSide *dp_side(void)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    DpSide *S = &g_side[dev];
    std::lock_guard<std::mutex> lock(g_side_mu);
    if (!S->ready) {
        for (int i = 0; i < 4; ++i)
            if (hipStreamCreateWithFlags(&S->s[i], hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&S->join[i], hipEventDisableTiming) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&S->fork, hipEventDisableTiming) != hipSuccess) return nullptr;
        S->ready = 1;
    }
    return S;
}

// mz_finalize(): the side streams and events of one device go with its context
extern "C" void mzk_release_device(int dev)
{
    if (dev < 0 || dev >= 16) return;
    DpSide *S = &g_side[dev];
    std::lock_guard<std::mutex> lock(g_side_mu);
    if (!S->ready) return;
    int cur = 0;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    if (hipSetDevice(dev) == hipSuccess) {
        for (int i = 0; i < 4; ++i) { (void)hipStreamSynchronize(S->s[i]); (void)hipStreamDestroy(S->s[i]); (void)hipEventDestroy(S->join[i]); }
        (void)hipEventDestroy(S->fork);
    }
    S->ready = 0;
    if (have) (void)hipSetDevice(cur);
}

// how many batches' DPs the caller runs side by side (mz_dev_run_async): what counts for the choice of the
// latency-tolerant row kernel is the waves on the GPU, not the blocks of one launch
static int g_abreast = 1;
extern "C" void mzk_set_abreast(int k) { g_abreast = k < 1 ? 1 : k; }

extern "C" int mz_dp_hint(int n, const int64_t *totals)
{
    const long long failed = totals[3], wf = totals[5] & 0xffffffffLL, rowbig = totals[5] >> 32, wide = totals[8] & 0xffffffffLL, lag = totals[8] >> 32;
    const long long row = (long long)n - failed - wf - wide - lag - rowbig;         // row-parallel pairs of blocks of one to three rows
    // bits 8..15: the kind with the most pairs (launched last: the others' long-running pairs start first and it fills the GPU)
    // (a batch with blocks of four rows or more gives ALL its row-parallel pairs to k_dp_row_big: MZ_DP_ROW stays clear)
    const long long cnt[5] = { rowbig > 0 ? 0 : row, wf, wide, lag, rowbig > 0 ? row + rowbig : 0 };
    const int bit[5] = { MZ_DP_ROW, MZ_DP_WAVEFRONT, MZ_DP_WIDE, MZ_DP_LAG, MZ_DP_ROWBIG };
    int most = 0;
    for (int i = 1; i < 5; ++i) if (cnt[i] > cnt[most]) most = i;
    return MZ_DP_KNOWN | (cnt[0] > 0 ? MZ_DP_ROW : 0) | (cnt[4] > 0 ? MZ_DP_ROWBIG : 0) | (wf > 0 ? MZ_DP_WAVEFRONT : 0) | (wide > 0 ? MZ_DP_WIDE : 0) |
           (lag > 0 ? MZ_DP_LAG : 0) | (bit[most] << 8);
}

extern "C" int mz_dp_kinds(int dp_hint)                 // how many DP kernels a batch with this hint launches
{
    int k = 0;
    if (dp_hint & MZ_DP_ROWBIG) dp_hint &= ~MZ_DP_ROW;
    for (int bit : { MZ_DP_ROW, MZ_DP_ROWBIG, MZ_DP_WAVEFRONT, MZ_DP_WIDE, MZ_DP_LAG }) k += (dp_hint & bit) != 0;
    return k;
}

extern "C" int mz_dp_rows(int n, const int64_t *totals)
{
    const long long row = (long long)n - totals[3] - (totals[5] & 0xffffffffLL) - (totals[8] & 0xffffffffLL) - (totals[8] >> 32);
    return row > 0 ? (int)row : 0;
}
extern "C" int mz_dp_grid(int n, const int64_t *totals)
{
    const long long wf = totals[5] & 0xffffffffLL, wide = totals[8] & 0xffffffffLL, lag = totals[8] >> 32;
    auto f = [](long long c) { const long long g = (c + 7) / 8; return (int)(g > 1023 ? 1023 : g); };
    return f(wf) | (f(wide) << 10) | (f(lag) << 20);
}

// The DP of pairs [first, first+count): k_dp_row / k_dp_row_big (a block per pair; pairs of other modes leave at once), and the three
// kernels that take their pairs from a counter (wavefront / strip, blocks of 128..255 rows, lagged row-parallel).
// Without a hint all four go onto `stream`, one after the other.  With the plan's totals in hand (b->dp_hint) only the
// kernels that have pairs are launched -- and side by side, on side streams forked from and joined back into `stream`,
// when several do: a kernel that got a few hundred pairs takes as long as its longest pair, and back to back those
// tails cost a mixed batch a quarter of its DP time (20 000 pairs with indels: 0.29 + 0.95 + 3.0 ms).
static int grid_of(int count, int most, int field)       // waves for a counter kernel: pairs (8 x field, when known), at most `most`
{
    static int use = -1;                          // MZ_DP_GRID=0: the full grids (measurements)
    if (use < 0) { const char *e = getenv("MZ_DP_GRID"); use = !(e && e[0] == '0'); }
    if (!use) field = 0;
    int g = count < most ? count : most;
    if (field > 0 && 8 * field < g) g = 8 * field;
    return g < 1 ? 1 : g;
}
// `done` (may be null): an event to stand for "these DP kernels are through".  Where the batch takes ONE row-parallel kernel the event
// rides on that kernel's own dispatch packet (hipExtLaunchKernel's stop event) and *done_set = 1; otherwise the caller records it.  Why
// it matters: on a queue whose pipe-mate has a kernel running every PACKET starts 60-200 us late (mz_flow.c) -- a DP stream's next DP
// behind an event-record packet and a wait packet started three such delays after the DP before it ended (the chunk pipelines' DP
// kernels: 70-340 us apart on their stream, profiles/r6_timeline_host_c2.txt before / after).
static thread_local hipEvent_t t_dp_done = nullptr;
static thread_local int t_dp_done_set = 0;
extern "C" int mzk_dp_range_on(const mz_dev_batch *b, int first, int count, void *stream, const mz_dp_lanes *lanes);
extern "C" int mzk_dp_range_ev(const mz_dev_batch *b, int first, int count, void *stream, const mz_dp_lanes *lanes, void *done, int *done_set)
{
    static int on = -1;
    if (on < 0) { const char *e = getenv("MZ_DP_EVENT_RIDES"); on = !(e && e[0] == '0'); }
    t_dp_done = on ? (hipEvent_t)done : nullptr; t_dp_done_set = 0;
    const int rc = mzk_dp_range_on(b, first, count, stream, lanes);
    if (done_set) *done_set = t_dp_done_set;
    t_dp_done = nullptr;
    return rc;
}
#define ROW_LAUNCH(kernel, ...) do { \
        if (t_dp_done && nk == 1) { hipExtLaunchKernelGGL(kernel, dim3(row_blocks), dim3(WAVE), dyn_lds, s, nullptr, t_dp_done, 0, __VA_ARGS__); t_dp_done_set = 1; } \
        else hipLaunchKernelGGL(kernel, dim3(row_blocks), dim3(WAVE), dyn_lds, s, __VA_ARGS__); } while (0)
extern "C" int mzk_dp_range_on(const mz_dev_batch *b, int first, int count, void *stream, const mz_dp_lanes *lanes)
{
    if (count <= 0) return 0;
    static int dyn_lds = -1;                      // MZ_DYN_LDS=<bytes>: occupancy experiments (extra, unused LDS per wave)
    if (dyn_lds < 0) { const char *e = getenv("MZ_DYN_LDS"); dyn_lds = e ? atoi(e) : 0; }
    static int dyn_lds_lag = -1;                  // MZ_DYN_LDS_LAG=<bytes>: the same for k_dp_lag
    if (dyn_lds_lag < 0) { const char *e = getenv("MZ_DYN_LDS_LAG"); dyn_lds_lag = e ? atoi(e) : 0; }
    static int lat_max = -1;                      // MZ_LAT_MAX=<waves>: launches that leave the GPU at most that many row-parallel waves take k_dp_row_lat (0: never)
    if (lat_max < 0) { const char *e = getenv("MZ_LAT_MAX"); lat_max = e ? atoi(e) : 2048; }
    static int solo_on = -1;                      // MZ_SOLO=0: never the one-wave-a-SIMD build (measurements)
    if (solo_on < 0) { const char *e = getenv("MZ_SOLO"); solo_on = !(e && e[0] == '0'); }
    static int serial = -1;                       // MZ_DP_SERIAL=1: never side by side (measurements)
    if (serial < 0) { const char *e = getenv("MZ_DP_SERIAL"); serial = e && e[0] == '1'; }
    hipStream_t main_s = (hipStream_t)stream;
    const bool known = (b->dp_hint & MZ_DP_KNOWN) != 0;
    int hint = known ? b->dp_hint : (MZ_DP_ROW | MZ_DP_ROWBIG | MZ_DP_WAVEFRONT | MZ_DP_WIDE | MZ_DP_LAG);
    if (known && (hint & MZ_DP_ROWBIG)) hint &= ~MZ_DP_ROW;      // blocks of four rows or more in the batch: k_dp_row_big takes every row-parallel pair
    // launch order: the kernels whose pairs take longest first (a wavefront pair has twice the steps, a lagged one 10 %
    // more than a row-parallel one), and the kind with the most pairs last of all, on `stream` itself: the others get
    // their CUs before its blocks have taken all the LDS, and it fills what their last pairs leave idle (20 000 pairs,
    // row kernel last / lagged kernel last: 2.45 / 2.71 ms at 2 indel events per 1 000 columns, 3.15 / 3.02 ms at 10)
    int kinds[5] = { MZ_DP_WAVEFRONT, MZ_DP_WIDE, MZ_DP_LAG, MZ_DP_ROWBIG, MZ_DP_ROW };
    {
        const int most = (b->dp_hint >> 8) & 0xff;
        for (int i = 0; known && i < 4; ++i)                         // move it to the end, the rest keep their order
            if (kinds[i] == most) { for (int j = i; j < 4; ++j) kinds[j] = kinds[j + 1]; kinds[4] = most; break; }
    }
    // (whole batch, counts known: the row kernels' blocks take the plan's list too -- it is ordered, most cells first)
    const bool rows_listed = known && b->dp_rows > 0 && first == 0 && count == b->n;
    int nk = 0, last = 0;
    for (int i = 0; i < 5; ++i) if (hint & kinds[i]) { ++nk; last = i; }
    if (hint & (MZ_DP_WAVEFRONT | MZ_DP_WIDE | MZ_DP_LAG))
        CK(hipMemsetAsync(&b->totals[16], 0, 6 * sizeof(int64_t), main_s), "dp counters");
    // The side streams: the caller's own (`lanes`: the chunk pipelines give every chunk stream a set of its own -- one set per
    // device made the few long wavefront pairs of chunk k+1 wait for those of chunk k on the shared stream, a chain of 12 x 0.55 ms
    // that WAS a 20 000-pair call with indel bands), else the device's.  Fewer lanes than kinds: several kinds share a lane, in
    // launch order; none (lanes->n == 0): everything on `stream`.
    DpSide *S = (!lanes && (b->dp_hint & MZ_DP_KNOWN) && nk > 1 && !serial) ? dp_side() : nullptr;
    DpSide L;                                         // (the caller's lanes in the same shape)
    if (lanes && lanes->n > 0 && (b->dp_hint & MZ_DP_KNOWN) && nk > 1 && !serial) {
        L.ready = 1;
        for (int i = 0; i < 4; ++i) { L.s[i] = (hipStream_t)lanes->stream[i % lanes->n]; L.join[i] = (hipEvent_t)lanes->join[i % lanes->n]; }
        L.fork = (hipEvent_t)lanes->fork;
    }
    const int nlane = lanes ? (lanes->n < 4 ? lanes->n : 4) : 4;
    // (one set of side streams and events per DEVICE: two host threads on one device -- the tests' MZ_ALLOW_DUP_DEVICES --
    // must not interleave their record / wait sequences; what a wait refers to is fixed when it is enqueued)
    std::unique_lock<std::mutex> side_lock;
    if (S) side_lock = std::unique_lock<std::mutex>(S->mu);
    if (!S && lanes && lanes->n > 0 && (b->dp_hint & MZ_DP_KNOWN) && nk > 1 && !serial) S = &L;
    if (S) CK(hipEventRecord(S->fork, main_s), "dp fork");
    int used = 0, sides = 0;
    for (int i = 0; i < 5; ++i) {
        if (!(hint & kinds[i])) continue;
        const bool side = S && i != last;
        hipStream_t s = main_s;
        if (side) { s = S->s[sides % nlane]; if (sides < nlane) CK(hipStreamWaitEvent(s, S->fork, 0), "dp fork wait"); }
        // (the row kernels: a block per pair of the batch, or -- whole batch, counts known -- per entry of the plan's list)
        const int row_items = rows_listed ? b->dp_rows : count, row_blocks = row_items;
        if (kinds[i] == MZ_DP_ROW) {
            // few pairs -- a wave or two per SIMD, nothing to hide a latency behind: the latency-tolerant build (kernels/row.inc)
            // (a wave per SIMD at most, counting what the caller runs beside this launch -- MZ_DP_SOLO: the chunks of a call of at most 1 024
            //  long pairs; mz_dev_run_async's batches abreast: g_abreast --: the build whose waves cannot share a SIMD, kernels/row.inc)
            if (solo_on && ((b->dp_hint & MZ_DP_SOLO) || (!(b->dp_hint & MZ_DP_HELPERS_FIRST) && (long long)row_blocks * g_abreast <= SIMDS_TOTAL)) && (long long)row_blocks * g_abreast <= lat_max)
                ROW_LAUNCH(k_dp_row_solo, *b, first, row_items, rows_listed ? 3 : known ? 0 : 1);
            else if ((long long)row_blocks * g_abreast <= lat_max)   // (the DPs of g_abreast consecutive batches run side by side: mz_dev_run_async)
                ROW_LAUNCH(k_dp_row_lat, *b, first, row_items, rows_listed ? 3 : known ? 0 : 1);
            else
                ROW_LAUNCH(k_dp_row, *b, first, row_items, rows_listed ? 3 : known ? 0 : 1);
        }
        else if (kinds[i] == MZ_DP_ROWBIG)
            ROW_LAUNCH(k_dp_row_big, *b, first, row_items, rows_listed ? 3 : known ? 0 : 2);
        // (the counter kernels: no more waves than pairs -- a wave that finds the counter exhausted still had to wait for
        //  its 9-13 KB of LDS beside the other kernels' waves, and the launch is over only when the last one has)
        else if (kinds[i] == MZ_DP_WAVEFRONT)
            hipLaunchKernelGGL(k_dp, dim3(grid_of(count, 6144, b->dp_grid & 1023)), dim3(WAVE), dyn_lds, s, *b, first, count);
        else if (kinds[i] == MZ_DP_WIDE) {                // the second list: blocks of 128+ rows, and the rolling form with late starts
            // (three kernels share the list and follow one another on this stream; one without pairs of its own leaves at once -- the
            //  tagged strips first: theirs are the usual ones)
            hipLaunchKernelGGL(k_dp_duo, dim3(grid_of(count, 1792, (b->dp_grid >> 10) & 1023)), dim3(DUO_T), 0, s, *b, first, count);     // (21 KB of LDS: seven workgroups a CU)
            hipLaunchKernelGGL(k_dp_tstrip, dim3(grid_of(count, 5120, (b->dp_grid >> 10) & 1023)), dim3(WAVE), 0, s, *b, first, count);
            hipLaunchKernelGGL(k_dp_wide, dim3(grid_of(count, 2048, (b->dp_grid >> 10) & 1023)), dim3(WAVE), 0, s, *b, first, count);
            hipLaunchKernelGGL(k_dp_roll, dim3(grid_of(count, 2048, (b->dp_grid >> 10) & 1023)), dim3(WAVE), 0, s, *b, first, count);
        }
        else
            hipLaunchKernelGGL(k_dp_lag, dim3(grid_of(count, 4096, (b->dp_grid >> 20) & 1023)), dim3(WAVE), dyn_lds_lag, s, *b, first, count);
        if (side) { ++sides; used = sides < nlane ? sides : nlane; }
    }
    // (a lane's join is recorded once, behind the last kind it got)
    for (int i = 0; i < used; ++i) { CK(hipEventRecord(S->join[i], S->s[i]), "dp join"); CK(hipStreamWaitEvent(main_s, S->join[i], 0), "dp join wait"); }
    CK(hipGetLastError(), "dp launch");
    return 0;
}
extern "C" int mzk_dp_range(const mz_dev_batch *b, int first, int count, void *stream) { return mzk_dp_range_on(b, first, count, stream, nullptr); }

// ------------------------------------------------------------------------------------------
// The chunk pipelines' link traffic as KERNELS on the chunk's own stream (round 5).  The staging block lies in pinned host memory,
// which the GPU reads and writes over PCIe like any other address; a copy kernel in the chunk's stream is ordered with the kernels
// around it by the stream itself, where a hipMemcpyAsync() goes to an SDMA engine that takes its commands in order across ALL
// streams -- and on this platform was seen to stand still for 3-8 ms at a time (both directions at once, kernels running on:
// the "one call in five takes twice as long" of rounds 3 and 4).  16 bytes per lane, four loads in flight per lane.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) HELPER_VGPRS void k_link_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, long long n16)
{
    HELPER_PRIO();
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + stride < n16; i += 2 * stride) {           // (two loads in flight per lane: 24 VGPRs -- room beside five DP waves)
        const uint4 a = src[i], b = src[i + stride];
        dst[i] = a; dst[i + stride] = b;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}
extern "C" int mzk_link_copy(void *dst, const void *src, size_t bytes, void *stream)
{
    if (!bytes) return 0;
    if (((uintptr_t)dst | (uintptr_t)src | bytes) & 15) { snprintf(g_err, sizeof g_err, "link copy: not 16-byte aligned"); return -1; }
    // Few waves, each with 2 KB in flight: the link holds ~57 GB/s x a few microseconds = a few hundred KB at a time, and a copy wave
    // waiting for it sits where a DP wave could (MZ_COPY_BLOCKS overrides; a 50 000-pair C2 call: 2 / 4 / 8 / 16 / 32 / 64 / 2 048 blocks 13.9 / 8.7 / 8.2 / 8.15 / 8.6 / 9.2 / 11.5 ms)
    static int cap = -1;
    if (cap < 0) { const char *e = getenv("MZ_COPY_BLOCKS"); cap = e && atoi(e) > 0 ? atoi(e) : 16; }
    const long long n16 = (long long)(bytes / 16);
    long long blocks = (n16 + 511) / 512;                // two 16-byte pieces per lane at least
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_link_copy, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint4 *)src, (uint4 *)dst, n16);
    CK(hipGetLastError(), "link copy launch");
    return 0;
}
extern "C" int mzk_walk_range(const mz_dev_batch *b, int first, int count, void *stream, int beside_dp)
{
    if (count <= 0) return 0;
    // Serial form: the run-following walk, a wave per pair.  Beside another batch's DP (mz_dev_run_async, the chunks
    // of mz_yama_batch) the choice between it and the lane-per-pair chase depends on the batch (walk_by_chase(),
    // kernels/walk.inc) and is made on the device from the plan's totals: both kernels are launched and the one not
    // chosen returns at once.  MZ_WALK=wave|direct forces one (tests, measurements).
    static int force = -1;
    if (force < 0) { const char *e = getenv("MZ_WALK"); force = !e ? 0 : e[0] == 'w' ? 1 : e[0] == 'd' ? 2 : e[0] == 'r' ? 3 : 0; }
    const int hint = (force || !beside_dp || count <= 16384) ? 0 : b->walk_hint;
    const bool both = !force && beside_dp && count > 16384 && hint == MZ_WALK_AUTO;
    // Beside DP kernels the row-parallel pairs of a batch whose caller says so (walk_hint MZ_WALK_REG, from mz_walk_choice(): a launch of at
    // most 4 096 pairs, or thin blocks -- and no long pairs) take the walk whose window lies in registers (k_walk_reg, kernels/walk.inc: no
    // LDS, 30 VGPRs, runs of at most 17-32 steps an iteration); k_walk_wave takes what is left, when the plan's totals say there is
    // anything.  Measured, one mz_yama_batch() call, reg / LDS window: C3 4.9 / 5.45 ms, c2i 5.6 / 5.85, C2 8.42 / 8.42 -- and C4, whose paths
    // run straight for 64 steps and more, 22 / 18; C5's 100 000-row pairs 11.9 / 7 ms a walk (a lone wave per pair: the LDS form's window
    // fetched a whole window ahead is what counts there): hence the rule.  MZ_WALK=reg: everywhere; MZ_WALK_REG=0: nowhere.
    static int reg_on = -1;
    if (reg_on < 0) { const char *e = getenv("MZ_WALK_REG"); reg_on = !(e && e[0] == '0'); }
    const bool known = (b->dp_hint & MZ_DP_KNOWN) != 0;
    const bool reg = force == 3 || (!force && reg_on && beside_dp && count <= 16384 && b->walk_hint == MZ_WALK_REG &&
                                    (!known || (b->dp_hint & (MZ_DP_ROW | MZ_DP_ROWBIG))));
    const bool rest = !reg || !known || (b->dp_hint & (MZ_DP_WAVEFRONT | MZ_DP_WIDE | MZ_DP_LAG));
    if (reg) hipLaunchKernelGGL(k_walk_reg, dim3(count), dim3(WAVE), 0, (hipStream_t)stream, *b, first, count);
    if ((force == 3 || !force) && reg ? rest : force ? force == 1 : (both || hint != MZ_WALK_CHASE)) {
        const int pol = (both ? 1 : 0) | (reg ? 2 : 0);
        if (count <= 4096)      // a launch of few pairs: long ones as likely as not -- the variant that fetches its next window ahead
            hipLaunchKernelGGL(k_walk_wave_ahead, dim3(count), dim3(WAVE), 0, (hipStream_t)stream, *b, first, count, pol);
        else
            hipLaunchKernelGGL(k_walk_wave, dim3(count), dim3(WAVE), 0, (hipStream_t)stream, *b, first, count, pol);
    }
    if (force ? force == 2 : (both || hint == MZ_WALK_CHASE)) {
        const int waves = (count + WALK_LANES - 1) / WALK_LANES;
        CK(hipMemsetAsync(&b->totals[10], 0, sizeof(int64_t), (hipStream_t)stream), "walk counter");      // the chase's pair counter
        hipLaunchKernelGGL(k_walk, dim3(waves < WALK_GRID ? waves : WALK_GRID), dim3(WAVE), 0, (hipStream_t)stream, *b, first, count, both ? 1 : 0);
        // (the exact kernels' pairs, if the batch has any: the plan's totals know -- wavefront and 128+-row lists)
        if (!(b->dp_hint & MZ_DP_KNOWN) || (b->dp_hint & (MZ_DP_WAVEFRONT | MZ_DP_WIDE)))
            hipLaunchKernelGGL(k_walk_exact, dim3(waves), dim3(WAVE), 0, (hipStream_t)stream, *b, first, count, both ? 1 : 0);
    }
    CK(hipGetLastError(), "walk launch");
    return 0;
}
// the host-side twin of walk_by_chase() (kernels/walk.inc), for callers that hold a copy of the plan's totals
extern "C" int mz_walk_choice(int n, const int64_t *totals)
{
    const long long ok = (long long)n - totals[3];
    const bool thin = MZ_ROWS_SUM(totals[11]) < 8 * (ok > 0 ? ok : 1);       // fewer than 8 rows a pair: the paths turn every few steps
    const bool lng = totals[0] > (long long)(ok > 0 ? ok : 1) * (1 << 18);      // a quarter of a million traceback dwords a pair: ~20 000 rows and more
    return (n > 16384 && thin) ? MZ_WALK_CHASE : ((thin || n <= 4096) && !lng) ? MZ_WALK_REG : MZ_WALK_RUNS;
}

extern "C" int mzk_emit_range(const mz_dev_batch *b, int first, int count, void *stream)
{
    if (count <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    // a launch of few pairs: long thin pairs by several waves each (kernels/emit.inc; the count kernel reuses the head of the
    // pair's traceback slice, which the walk -- earlier on this stream -- has finished with)
    const int split = count <= 4096;
    hipLaunchKernelGGL(k_emit, dim3(count), dim3(WAVE), 0, s, *b, first, count, split);
    if (split) {
        hipLaunchKernelGGL(k_emit_long_count, dim3(count, EMIT_MAXSEG), dim3(WAVE), 0, s, *b, first, count);
        hipLaunchKernelGGL(k_emit_long, dim3(count, EMIT_MAXSEG), dim3(WAVE), 0, s, *b, first, count);
    }
    hipLaunchKernelGGL(k_emit_wide, dim3(count), dim3(WAVE), 0, s, *b, first, count);
    CK(hipGetLastError(), "emit launch");
    return 0;
}
// The host path's result: per pair a 32-byte record and the edit script at two bits per merged column, in column order
// (the walk leaves a byte per column, last column first).  Also the reference's closing check of the emit
// (mz_yama.c:310-312): the script must take exactly M columns of A and N of B.  hdr[0] += band cells of the pair.
extern "C" int mzk_script_pack(const mz_dev_batch *b, void *hdr, void *recs, void *packed, void *stream)
{
    if (b->n <= 0) return 0;
    const int split = b->n <= 4096;                 // long thin pairs by several waves each (kernels/emit.inc)
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_script_pack, dim3(b->n), dim3(WAVE), 0, s, *b, (long long *)hdr, (mz_res_rec *)recs, (uint8_t *)packed, split);
    if (split) {
        hipLaunchKernelGGL(k_script_pack_long, dim3(b->n, EMIT_MAXSEG), dim3(WAVE), 0, s, *b, (uint8_t *)packed);
        hipLaunchKernelGGL(k_script_fin, dim3((b->n + 255) / 256), dim3(256), 0, s, *b, (long long *)hdr, (mz_res_rec *)recs);
    }
    CK(hipGetLastError(), "script pack launch");
    return 0;
}
static int helper_grid(void)              // waves of k_pre_lds / k_fin (each takes its pairs in turn); MZ_HELPER_GRID overrides
{
    static int v = -1;
    if (v < 0) { const char *e = getenv("MZ_HELPER_GRID"); v = e && atoi(e) > 0 ? atoi(e) : 65536; }
    return v;
}
extern "C" int mzk_pre(const mz_pre_batch *q, const mz_fin_batch *f, const mz_dev_batch *b, void *stream)
{
    if (q->n <= 0) return 0;
    if (q->nib && !q->txt && q->lds_bytes > 0 && q->lds_bytes <= 64 * 1024) {
        const int grid = q->n < helper_grid() ? q->n : helper_grid();
        if (q->lds16) hipLaunchKernelGGL(k_pre_lds<int16_t>, dim3(grid), dim3(WAVE), (size_t)q->lds_bytes, (hipStream_t)stream, *q, *f, *b);
        else hipLaunchKernelGGL(k_pre_lds<int32_t>, dim3(grid), dim3(WAVE), (size_t)q->lds_bytes, (hipStream_t)stream, *q, *f, *b);
    } else hipLaunchKernelGGL(k_pre, dim3(q->n), dim3(WAVE), 0, (hipStream_t)stream, *q, *f, *b);
    CK(hipGetLastError(), "pre launch");
    return 0;
}
extern "C" int mzk_mid(const mz_pre_batch *q, const mz_dev_batch *b1, const mz_dev_batch *b2, void *stream)
{
    if (q->n <= 0) return 0;
    hipLaunchKernelGGL(k_mid, dim3(q->n), dim3(WAVE), 0, (hipStream_t)stream, *q, *b1, *b2);
    CK(hipGetLastError(), "mid launch");
    return 0;
}
extern "C" int mzk_fin(const mz_pre_batch *q, const mz_fin_batch *f, const mz_dev_batch *b1, const mz_dev_batch *b2, void *stream)
{
    if (b1->n <= 0) return 0;
    hipLaunchKernelGGL(k_fin, dim3(b1->n < helper_grid() ? b1->n : helper_grid()), dim3(WAVE), (size_t)f->lds_bytes, (hipStream_t)stream, *q, *f, *b1, f->any0 ? *b2 : *b1);
    CK(hipGetLastError(), "fin launch");
    return 0;
}
extern "C" int mzk_dp(const mz_dev_batch *b, void *stream)   { return mzk_dp_range(b, 0, b->n, stream); }
extern "C" int mzk_walk(const mz_dev_batch *b, void *stream, int beside_dp) { return mzk_walk_range(b, 0, b->n, stream, beside_dp); }
extern "C" int mzk_emit(const mz_dev_batch *b, void *stream) { return mzk_emit_range(b, 0, b->n, stream); }
