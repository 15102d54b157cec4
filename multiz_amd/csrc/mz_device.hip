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
//  * one 64-lane wave owns one block pair and sweeps anti-diagonals: lane = DP row mod 64,
//    step t handles cells (r, t-r).  (r-1,c) and (r-1,c-1) come from the neighbouring lane
//    through DPP wave rotates (no LDS traffic for the recurrence); (r,c-1) is the lane's own
//    previous value.  Column profiles of B sit in an LDS ring, row records of A in LDS.
//  * traceback bytes are stored "diagonal-major", four steps per dword, so that every store
//    is one fully coalesced 256-byte row per wave.
//
// No MFMA anywhere: this is an integer max-plus recurrence, not a contraction.

#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "mz_device.h"

#define WAVE   64
#define BRING  128          // B-profile ring entries (columns) in LDS, 16 B each
#define REC_DW 16           // dwords per staged row record
// row-parallel kernel: prep layout of a pair (dwords): row records of rows 1..M+2 (dead beyond M), then (COL
// mode) the transposed band bounds
#define RREC 16             // dwords per row record in b.prep
#define RCOL 8              // dwords per column record: uA uB c01 c23 | c45 xI+P P Q
#define ROW_NROWS(M) ((M) + 2)
#define ROW_NCOLS(N) ((((N) + WAVE + WAVE - 1) / WAVE) * WAVE + WAVE)
#define COL_PREP_DWORDS(N) (2LL * ((N) + 1))           // transposed band bounds of a COL pair

struct ScoreConst { int S6[36]; int go; int ge; int g1, g2; int tag_ok; int maxS; int row_on; };
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

__device__ __forceinline__ int pack4(int b0, int b1, int b2, int b3)
{
    return (b0 & 0xff) | ((b1 & 0xff) << 8) | ((b2 & 0xff) << 16) | ((b3 & 0xff) << 24);
}
__device__ __forceinline__ int pack2(int lo, int hi) { return (lo & 0xffff) | (hi << 16); }

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

// ------------------------------------------------------------------------------------------
// plan: validity prologue of yama (reference mz_yama.c:58-71), work sizes, kernel mode
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WAVE) void k_plan(mz_dev_batch b)
{
    const int p = blockIdx.x, lane = threadIdx.x;
    const int K = b.K[p], L = b.L[p], M = b.M[p], N = b.N[p];
    int status = MZ_OK, badrow = -1, mode = MZ_MODE_WF64, edgeLo = 0, edgeHi = 0;
    long long cells = 0, szTb = 0, szPrep = 0;

    if (K < 1 || K > 127 || L < 1 || L > 127) status = MZ_E_ROWS;
    else if (M < 1 || N < 1) status = MZ_E_SHAPE;
    else if ((long long)K * L * (c_sc.go + c_sc.ge) * ((long long)M + N + 2) >= (1LL << 30)) status = MZ_E_RANGE;

    if (status == MZ_OK) {
        const int *LB = b.poolLB + b.offBand[p], *RB = b.poolRB + b.offBand[p];
        if (LB[0] != 0 || RB[M] != N) status = MZ_E_TERMINATION;
        else {
            const int need = N < 10 ? N : 10;
            int key = 0x7fffffff;           // (row << 2 | kind), kind in the reference's test order
            int wf_ok = 1, conn = 1, row_ok = 1, col_ok = 1;
            int rL = 0, rN = M;             // last row with LB[r] <= 1, first row with RB[r] == N
#pragma unroll 4                     // (the loads of four chunks in flight: the loop is pure memory latency)
            for (int r = lane; r <= M; r += WAVE) {
                const int lo = LB[r], hi = RB[r];
                if (hi - lo < need) key = min(key, (r << 2) | 0);
                if (r > 0 && lo < LB[r-1]) key = min(key, (r << 2) | 1);
                if (r > 0 && hi < RB[r-1]) key = min(key, (r << 2) | 2);
                cells += hi - lo + 1;
                // a lane must have left row r before row r+64 (same lane) and its right
                // neighbour's row r+65 need it: RB[r] - LB[r+64] <= 62
                if (r + WAVE <= M && hi - LB[r + WAVE] > 62) wf_ok = 0;
                if (r > 0 && lo > RB[r-1]) conn = 0;          // row r would not touch row r-1's band
                if (hi - lo > 62) row_ok = 0;                  // row-parallel kernel: one row of the band per wave
                if (r + 63 <= M && LB[r + 63] <= hi) col_ok = 0;   // transposed: a column would span 64 rows
                if (lo <= 1) rL = max(rL, r);
                if (hi == N) rN = min(rN, r);
            }
            conn = wave_min(conn);
            row_ok = wave_min(row_ok);
            col_ok = wave_min(col_ok);
            rL = -wave_min(-rL);
            rN = wave_min(rN);
            key = wave_min(key);
            wf_ok = wave_min(wf_ok);
            cells = wave_sum64(cells);
            if (key != 0x7fffffff) {
                badrow = key >> 2;
                status = (key & 3) == 0 ? MZ_E_NARROW : (key & 3) == 1 ? MZ_E_LB_MONO : MZ_E_RB_MONO;
            } else if (wf_ok) {
                // fast kernel: connected band + every reachable score above -2^29 (so that sentinel
                // states, which sit at about -2^30, can never win a comparison) + factorable gap_open
                // |score| of a reachable state: a path has at most M+N steps and a step moves the score by at most
                // K*L*(go + ge) (gap step: every row pair opens and extends) or K*L*(go + max|sigma|) (aligned step)
                const long long reach = (long long)K * L * (c_sc.go + max(c_sc.ge, c_sc.maxS)) * ((long long)M + N + 2);
                const bool small = reach < (1LL << 29);
                mode = (conn && small && c_sc.g1 > 0) ? MZ_MODE_FAST : MZ_MODE_WF64;
                // tagged variant: one more bit of headroom, doubled int16 vectors must still fit
                if (mode == MZ_MODE_FAST && c_sc.tag_ok && reach < (1LL << 28) &&
                    2 * K * (c_sc.maxS + c_sc.go) <= 32767)
                    mode = MZ_MODE_FASTT;
                edgeLo = rL + 1;
                edgeHi = min(rN + N, M + LB[M]);          // first step that can touch column N or row M
                szTb = max((long long)(((M + N) >> 2) + 1) * WAVE, (long long)(((M + N) >> 4) + 1) * 3 * WAVE);
                // Row-parallel kernels.  Their scores are re-based every 32 rows and their running sums every 64
                // columns, so what must fit is one window: ~240 steps (band width + 2 x 32 rows + band width) of at
                // most K*L*(go + max(ge, max|sigma|)) each, times 4 for the tags, within 2^27 (the ring lift is 2^30)
                // -- whatever M and N are.
                const bool fam = conn && c_sc.g1 > 0 && c_sc.tag_ok && c_sc.row_on;
                if (fam && (long long)K * L * (c_sc.go + max(c_sc.ge, c_sc.maxS)) * 960 < (1LL << 27)) {
                    if (row_ok && 2 * K * (c_sc.maxS + c_sc.go) <= 32767) {
                        mode = MZ_MODE_ROW;
                        edgeLo = rL;                            // rows <= rL can hold column 0 or 1
                        edgeHi = rN;                            // rows >= rN hold column N
                        szTb = (long long)((M >> 4) + 1) * 3 * WAVE;
                    } else if (col_ok && 2 * L * (c_sc.maxS + c_sc.go) <= 32767) {
                        // the same kernel on the transposed problem (A and B, D and I exchanged): the band
                        // column by column must be at most 63 rows high
                        // (column c lies in rows r and r+63 iff LB[r+63] <= c <= RB[r]: checked in the row loop)
                        mode = MZ_MODE_COL;
                        edgeLo = RB[1 < M ? 1 : M];             // last column whose first row is 0 or 1
                        edgeHi = LB[M];                         // first column that reaches row M
                        szTb = (long long)((N >> 4) + 1) * 3 * WAVE;
                        szPrep = COL_PREP_DWORDS(N);
                    }
                }
                // more rows (K*L above ~200): one window no longer fits beside the 2^30 lift of the ring periods, so
                // the prefix maximum runs on lanes rotated to the band start instead (same re-basing).
                if (mode != MZ_MODE_ROW && mode != MZ_MODE_COL && fam &&
                    (long long)K * L * (c_sc.go + max(c_sc.ge, c_sc.maxS)) * 960 < (1LL << 29)) {
                    if (row_ok && 2 * K * (c_sc.maxS + c_sc.go) <= 32767) {
                        mode = MZ_MODE_ROWR;
                        edgeLo = rL; edgeHi = rN;
                        szTb = (long long)((M >> 4) + 1) * 3 * WAVE;
                    } else if (col_ok && 2 * L * (c_sc.maxS + c_sc.go) <= 32767) {
                        mode = MZ_MODE_COLR;
                        edgeLo = RB[1 < M ? 1 : M]; edgeHi = LB[M];
                        szTb = (long long)((N >> 4) + 1) * 3 * WAVE;
                        szPrep = COL_PREP_DWORDS(N);
                    }
                }
            } else {
                mode = MZ_MODE_STRIP;
                // strips of 64 rows; strip s sweeps columns LB[first]..RB[last] with a 64-step skew.
                // layout: [2*S header dwords rounded up to 64] [2 boundary rows of 3*(N+1) dwords, rounded]
                //         + per strip ceil(steps/4)*64 dwords
                const int S = (M + WAVE - 1) / WAVE;
                long long acc = 0;
                for (int s = lane; s < S; s += WAVE) {
                    const int first = s * WAVE + 1, last = min(first + WAVE - 1, M);
                    const int steps = RB[last] - LB[first] + 1 + (last - first);
                    acc += (long long)((steps + 3) >> 2) * WAVE;
                }
                acc = wave_sum64(acc);
                szTb = acc + (((2LL * S) + WAVE - 1) / WAVE) * WAVE + ((6LL * (N + 1) + WAVE - 1) / WAVE) * WAVE;
            }
        }
    }
    if (lane == 0) {
        const bool ok = status == MZ_OK;
        b.status[p] = status;
        b.badrow[p] = badrow;
        b.mode[p] = mode;
        b.edgeLo[p] = edgeLo;
        b.edgeHi[p] = edgeHi;
        b.cells[p] = ok ? cells : 0;
        b.szTb[p] = ok ? szTb : 0;
        b.szPrep[p] = ok ? ((szPrep + 63) & ~63LL) : 0;
        b.szScript[p] = ok ? (((long long)M + N + 3) & ~3LL) : 0;
        b.szOut[p] = ok ? (((long long)(M + N) * (K + L) + 15) & ~15LL) : 0;
        b.om[p] = 0;
    }
}

// Exclusive prefix sums of the per-pair sizes and the failure count.
// Three small launches: per-block totals, scan of the block totals, per-block scan + write.
#define SCAN_Q 6
#define SCAN_B 256          // small blocks: these kernels run beside the DP and must fit into whatever slots are free
__device__ __forceinline__ void scan_load(const mz_dev_batch &b, int i, long long v[SCAN_Q])
{
    if (i < b.n) {
        const bool ok = b.status[i] == MZ_OK;
        v[0] = b.szTb[i]; v[1] = b.szScript[i]; v[2] = b.szOut[i]; v[3] = b.szPrep[i];
        v[4] = !ok; v[5] = ok && b.mode[i] < MZ_MODE_ROW;       // pairs left to the wavefront kernels (k_dp)
    } else {
#pragma unroll
        for (int q = 0; q < SCAN_Q; ++q) v[q] = 0;
    }
}
// block-wide exclusive scan of SCAN_Q interleaved sequences; tot = block totals
__device__ __forceinline__ void scan_block(long long v[SCAN_Q], long long ex[SCAN_Q], long long tot[SCAN_Q],
                                           long long (*sm)[SCAN_Q])
{
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    long long inc[SCAN_Q];
#pragma unroll
    for (int q = 0; q < SCAN_Q; ++q) {
        long long x = v[q];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const long long y = __shfl_up(x, o); if (lane >= o) x += y; }
        inc[q] = x;
        if (lane == 63) sm[w][q] = x;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < SCAN_Q; ++q) {
        long long base = 0, all = 0;
        for (int k = 0; k < SCAN_B / 64; ++k) { const long long y = sm[k][q]; if (k < w) base += y; all += y; }
        ex[q] = base + inc[q] - v[q];
        tot[q] = all;
    }
    __syncthreads();
}
__global__ __launch_bounds__(SCAN_B) void k_scan1(mz_dev_batch b)
{
    __shared__ long long sm[SCAN_B / 64][SCAN_Q];
    long long v[SCAN_Q], ex[SCAN_Q], tot[SCAN_Q];
    scan_load(b, blockIdx.x * SCAN_B + threadIdx.x, v);
    scan_block(v, ex, tot, sm);
    if (threadIdx.x < SCAN_Q) b.scanAux[(long long)blockIdx.x * SCAN_Q + threadIdx.x] = tot[threadIdx.x];
}
__global__ __launch_bounds__(64) void k_scan2(mz_dev_batch b, int nblk)
{
    const int q = threadIdx.x;
    if (q >= SCAN_Q) return;
    long long run = 0;
    for (int k = 0; k < nblk; ++k) { const long long y = b.scanAux[(long long)k * SCAN_Q + q]; b.scanAux[(long long)k * SCAN_Q + q] = run; run += y; }
    b.totals[q == 4 ? 3 : q == 5 ? 5 : q == 3 ? 4 : q] = run;      // [0..2] tb/script/out, [3] failed, [4] prep, [5] wavefront-kernel pairs
}
__global__ __launch_bounds__(SCAN_B) void k_scan3(mz_dev_batch b)
{
    __shared__ long long sm[SCAN_B / 64][SCAN_Q];
    long long v[SCAN_Q], ex[SCAN_Q], tot[SCAN_Q];
    const int i = blockIdx.x * SCAN_B + threadIdx.x;
    scan_load(b, i, v);
    scan_block(v, ex, tot, sm);
    if (i >= b.n) return;
    const int64_t *base = b.scanAux + (long long)blockIdx.x * SCAN_Q;
    b.offTb[i] = base[0] + ex[0]; b.offScript[i] = base[1] + ex[1]; b.offOut[i] = base[2] + ex[2]; b.offPrep[i] = base[3] + ex[3];
}

// after the scan: a pair whose slices do not fit the caller's workspace is failed, loudly
__global__ void k_fit(mz_dev_batch b)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.n || b.status[i] != MZ_OK) return;
    if (b.offTb[i] + b.szTb[i] > b.capTb || b.offScript[i] + b.szScript[i] > b.capScript ||
        b.offOut[i] + b.szOut[i] > b.capOut || b.offPrep[i] + b.szPrep[i] > b.capPrep)
        b.status[i] = MZ_E_WORKSPACE;
}

// ------------------------------------------------------------------------------------------
// DP
// ------------------------------------------------------------------------------------------
struct Tri { int C, D, I; };

// everything a lane needs to know about its current DP row; built once per row by
// stage_rows() (64 rows in parallel) and read back from LDS when the lane re-arms
struct RowRegs {
    int lo, hi;        // LB[r], RB[r]
    int lb1;           // LB[r-1]
    int tC;            // r>1 ? LB[r-2] : BIG   (guards of mz_yama.c:177-179,215-216)
    int tY;            // r>1 ? 0 : BIG         (the bare "row > 1" guards, :181-182,217-218)
    int mI;            // r<M ? gap_open : 0    (no open for trailing end-gaps, :123)
    int rxC, ryC, rzC; // int8x4 row vectors for the three C-state gap sums
    int rxI;           // ... for the I-state x sum
    int rxD;           // ... for the D-state x sum (accumulator accD)
    int accD;          // nA * L
    int penDy;         // gap_open * L * (nA - PA00)
    int penDz;         // gap_open * L * nA
    int extD;          // gap_extend * L * nA
    int w01, w23, w45; // int16x2 substitution row vector  cntA^T * S6
};

struct PairCtx {
    int K, L, M, N;
    const uint8_t *A, *B;
    const int *LB, *RB;
};

__device__ __forceinline__ void rec_dead(int *dst)
{
    int4 *d = (int4 *)dst;
    d[0] = make_int4(MZ_BIG, -1, MZ_BIG, MZ_BIG);
    d[1] = make_int4(MZ_BIG, 0, 0, 0);
    d[2] = make_int4(0, 0, 0, 0);
    d[3] = make_int4(0, 0, 0, 0);
}

// Build the records of rows 64*blk+1 .. 64*blk+64 (lane <-> row) into LDS slot blk&1.
// Column profile of A column r (appendix A.4 of SURVEY.md): class counts, non-dash / dash
// counts and the two "same as previous column" counts PA00, PA11.
__device__ __forceinline__ void stage_rows(int blk, int lane, const PairCtx &J, int *recs)
{
    const int rr = blk * WAVE + lane + 1;
    int *dst = recs + (((blk & 1) * WAVE) + lane) * REC_DW;
    if (rr > J.M) { rec_dead(dst); return; }

    const int K = J.K, L = J.L;
    const uint8_t *col = J.A + (long long)(rr - 1) * K;
    unsigned cnt = 0;              // four 8-bit counters: classes 0..3
    int dA = 0, a00 = 0, a11 = 0, other = 0;
    for (int i = 0; i < K; ++i) {
        const unsigned ch = col[i];
        const bool dash = ch == '-';
        const bool pdash = (rr > 1) ? (col[i - K] == '-') : false;
        const int cl = byte_class(ch);
        cnt += (cl < 4) ? (1u << (cl << 3)) : 0u;
        other += cl == 5;
        dA += dash;
        a00 += (!dash) & (!pdash);
        a11 += dash & pdash;
    }
    const int nA = K - dA;
    int cn[6] = { (int)(cnt & 0xff), (int)((cnt >> 8) & 0xff), (int)((cnt >> 16) & 0xff), (int)(cnt >> 24), dA, other };
    int w[6];
#pragma unroll
    for (int l = 0; l < 6; ++l) {
        int s = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) s += cn[k] * c_sc.S6[k * 6 + l];
        w[l] = s;
    }
    const int go = c_sc.go;
    const int lb1 = J.LB[rr - 1];
    const int lb2 = rr > 1 ? J.LB[rr - 2] : 0;
    int4 *d = (int4 *)dst;
    d[0] = make_int4(J.LB[rr], J.RB[rr], lb1, rr > 1 ? lb2 : MZ_BIG);
    d[1] = make_int4(rr > 1 ? 0 : MZ_BIG, rr < J.M ? go : 0,
                     pack4(nA, dA, -a00, -a11), pack4(nA - a00, dA, 0, 0));
    d[2] = make_int4(pack4(nA, dA, 0, -dA), pack4(0, K, 0, -dA), pack4(-a00, 0, 0, 0), nA * L);
    d[3] = make_int4(go * L * (nA - a00), pack2(w[0], w[1]), pack2(w[2], w[3]), pack2(w[4], w[5]));
}

__device__ __forceinline__ void load_rec(RowRegs &R, const int *src)
{
    const int4 *s = (const int4 *)src;
    const int4 a = s[0], b = s[1], c = s[2], d = s[3];
    R.lo = a.x; R.hi = a.y; R.lb1 = a.z; R.tC = a.w;
    R.tY = b.x; R.mI = b.y; R.rxC = b.z; R.ryC = b.w;
    R.rzC = c.x; R.rxI = c.y; R.rxD = c.z; R.accD = c.w;
    R.penDy = d.x; R.w01 = d.y; R.w23 = d.z; R.w45 = d.w;
    R.penDz = c_sc.go * c.w;
    R.extD = c_sc.ge * c.w;
}

// Stage the profiles of B columns first..first+63 (lane <-> column) into the LDS ring.
// entry = { int8x4(dB, nB, PB11, PB00), int16x2(cnt0,cnt1), (cnt2,cnt3), (cnt4,cnt5) }
__device__ __forceinline__ void stage_bcols(int first, int lane, const PairCtx &J, int4 *ring)
{
    const int cc = first + lane;
    int4 e = make_int4(0, 0, 0, 0);
    if (cc >= 1 && cc <= J.N) {
        const int L = J.L;
        const uint8_t *col = J.B + (long long)(cc - 1) * L;
        unsigned cnt = 0;
        int dB = 0, b00 = 0, b11 = 0, other = 0;
        for (int j = 0; j < L; ++j) {
            const unsigned ch = col[j];
            const bool dash = ch == '-';
            const bool pdash = (cc > 1) ? (col[j - L] == '-') : false;
            const int cl = byte_class(ch);
            cnt += (cl < 4) ? (1u << (cl << 3)) : 0u;
            other += cl == 5;
            dB += dash;
            b00 += (!dash) & (!pdash);
            b11 += dash & pdash;
        }
        e.x = pack4(dB, L - dB, b11, b00);
        e.y = pack2(cnt & 0xff, (cnt >> 8) & 0xff);
        e.z = pack2((cnt >> 16) & 0xff, cnt >> 24);
        e.w = pack2(dB, other);
    }
    ring[cc & (BRING - 1)] = e;
}

// interior tie order of mz_yama.c:138-154: the C-predecessor wins ties, then D only if
// strictly greater than I.  m == x  <=>  x >= y && x >= z.
__device__ __forceinline__ int pick(int x, int y, int z, int fD, int fI, int &flag)
{
    const int m = max(max(x, y), z);
    const int f = (y > z) ? fD : fI;
    flag = (x == m) ? 0 : f;
    return m;
}

// One DP cell (r, c): the three updates of mz_yama.c:113-242 in profile form.
//   left = (C,D,I)(r, c-1)   up = P(r-1, c)   dg = P(r-1, c-1)
// q is the B-column profile.  Returns the new triple and the traceback byte (:253).
__device__ __forceinline__ Tri cell(const RowRegs &R, int c, int N, int4 q, Tri left, Tri up, Tri dg,
                                    int pkKy, int pkKz, int go, int ge, int &tbyte)
{
    const int cm1 = c - 1;
    const bool g1  = cm1 > R.lb1;            // c > LB[r-1]+1
    const bool gIz = cm1 > R.lo;             // c > LB[r]+1
    const bool vI  = c > R.lo;               // I exists (c != LB[r])
    const bool vC  = c > R.lb1;              // C exists
    const bool gCx = cm1 > R.tC;             // r>1 && c > LB[r-2]+1
    const bool gCy = cm1 > R.tY;             // r>1 && c > 1
    const bool inN = c < N;
    const bool gDx = (c > R.tC) & inN;       // r>1 && c > LB[r-2] && c < N   (c > LB[r-2] >= 0 implies c > 0)
    const bool gDy = (c > R.tY) & inN;       // r>1 && 0 < c < N
    const bool gDz = vC & inN;               // c > LB[r-1] && c < N
    Tri o;
    int fi, fc, fd, x, y, z, t;

    // ---- I  (from the same row, previous column)
    const int KnB = dot4(pkKy, q.x, 0);                 // K*nB
    t = left.C - __mul24(dot4(R.rxI, q.x, 0), R.mI);    // K*nB - dA*PB00
    x = g1 ? t : left.C;
    y = left.D - __mul24(KnB, R.mI);
    t = left.I - __mul24(dot4(pkKz, q.x, 0), R.mI);     // K*(nB - PB00)
    z = gIz ? t : left.I;
    o.I = pick(x, y, z, MZ_FD << 4, MZ_FI << 4, fi) - __mul24(KnB, ge);
    o.I = vI ? o.I : MZ_NEG;

    // ---- C  (diagonal)
    t = dg.C - __mul24(dot4(R.rxC, q.x, 0), go);
    x = gCx ? t : dg.C;
    t = dg.D - __mul24(dot4(R.ryC, q.x, 0), go);
    y = gCy ? t : dg.D;
    t = dg.I - __mul24(dot4(R.rzC, q.x, 0), go);
    z = g1 ? t : dg.I;                                   // c > 1 is implied by c > LB[r-1]+1
    t = pick(x, y, z, MZ_FD, MZ_FI, fc);
    t = dot2(R.w01, q.y, dot2(R.w23, q.z, dot2(R.w45, q.w, t)));
    o.C = vC ? t : MZ_NEG;

    // ---- D  (from the row above)
    t = up.C - __mul24(dot4(R.rxD, q.x, R.accD), go);    // nA*L - PA00*dB
    x = gDx ? t : up.C;
    t = up.D - R.penDy;
    y = gDy ? t : up.D;
    t = up.I - R.penDz;
    z = gDz ? t : up.I;
    o.D = pick(x, y, z, MZ_FD << 2, MZ_FI << 2, fd) - R.extD;

    tbyte = fc | fd | fi;
    return o;
}

// wf64: 64 DP rows in flight, row r on lane (r-1)&63 (row 0 on lane 63), cell (r, t-r) at step t.
__device__ __forceinline__ void dp_wf64_body(const mz_dev_batch &b, int p, int lane, int *s_rec, int4 *s_ring)
{

    PairCtx J;
    J.K = b.K[p]; J.L = b.L[p]; J.M = b.M[p]; J.N = b.N[p];
    J.A = b.poolA + b.offA[p]; J.B = b.poolB + b.offB[p];
    J.LB = b.poolLB + b.offBand[p]; J.RB = b.poolRB + b.offBand[p];
    const int M = J.M, N = J.N;
    const int go = c_sc.go, ge = c_sc.ge;
    const int pkKy = pack4(0, J.K, 0, 0), pkKz = pack4(0, J.K, 0, -J.K);
    uint32_t *tbw = b.tbw + b.offTb[p];

    // ---- prologue: rows 1..128, B columns 1..64, arm the lanes
    stage_rows(0, lane, J, s_rec);
    stage_rows(1, lane, J, s_rec);
    s_ring[0] = make_int4(0, 0, 0, 0);       // column 0 has no profile (only D is computed there)
    stage_bcols(1, lane, J, s_ring);
    int cst = WAVE;                           // highest staged B column
    __builtin_amdgcn_s_waitcnt(0);            // LDS writes of this wave are in order; keep the compiler honest
    __syncthreads();

    RowRegs R;
    int r;                                    // this lane's current row
    Tri st = { MZ_NEG, MZ_NEG, MZ_NEG };      // (C,D,I) of this lane's latest cell; NEG while idle
    if (lane == WAVE - 1) {
        // row 0 (mz_yama.c:83-94): C = D = NEG, I(0,c) = I(0,c-1) - nB(c)*K*gap_extend.
        // Expressed with the general cell by zeroing every penalty; C and D are forced to NEG
        // after each step while row 0 is live.
        r = 0;
        R.lo = 0; R.hi = J.RB[0]; R.lb1 = MZ_BIG; R.tC = MZ_BIG; R.tY = MZ_BIG; R.mI = 0;
        R.rxC = R.ryC = R.rzC = R.rxI = R.rxD = 0; R.accD = R.penDy = R.penDz = R.extD = 0;
        R.w01 = R.w23 = R.w45 = 0;
        st.C = st.D = st.I = 0;               // grid point (0,0)
    } else {
        r = lane + 1;
        load_rec(R, s_rec + lane * REC_DW);
    }

    // wave-uniform bookkeeping of the oldest live row: rows finish strictly in order, one per step at most
    int rlo = 0;                              // oldest row not yet finished
    int lfin = WAVE - 1;                      // its lane
    int tfin = __builtin_amdgcn_readlane(R.hi, WAVE - 1);   // step at which it computes its last cell
    Tri up = { MZ_NEG, MZ_NEG, MZ_NEG }, dg;
    unsigned tbword = 0;
    const int Tend = M + N;

    for (int t = 1; t <= Tend; ++t) {
        // ---- predecessors from the neighbouring lane (row r-1): its cell of step t-1 is (r-1, c),
        //      the one it had a step earlier is (r-1, c-1).  Read BEFORE a finished lane is re-armed.
        dg = up;
        up.C = ror1(st.C); up.D = ror1(st.D); up.I = ror1(st.I);

        // ---- a finished row hands its lane to row+64
        if (t > tfin) {
            const int rn = rlo + WAVE;
            if (lane == lfin) {
                r = rn;
                load_rec(R, s_rec + ((((rn - 1) >> 6) & 1) * WAVE + lane) * REC_DW);
                st.C = st.D = st.I = MZ_NEG;
            }
            if (lfin == 0) {                  // lane 0 entered block k: every lane has left block k-1
                stage_rows(((rn - 1) >> 6) + 1, lane, J, s_rec);
                __syncthreads();
            }
            rlo += 1;
            lfin = (lfin + 1) & (WAVE - 1);
            tfin = rlo > M ? MZ_BIG : rlo + __builtin_amdgcn_readlane(R.hi, lfin);
        }
        // ---- keep the B-profile ring ahead of the leading column (t - rlo)
        if (t - rlo > cst) {
            stage_bcols(cst + 1, lane, J, s_ring);
            cst += WAVE;
            __syncthreads();
        }

        const int c = t - r;
        const int4 q = s_ring[c & (BRING - 1)];
        int tbyte;
        Tri nw = cell(R, c, N, q, st, up, dg, pkKy, pkKz, go, ge, tbyte);
        const bool active = (c >= R.lo) & (c <= R.hi);
        st.C = active ? nw.C : MZ_NEG;
        st.D = active ? nw.D : MZ_NEG;
        st.I = active ? nw.I : MZ_NEG;
        if (rlo == 0 && lane == WAVE - 1) { st.C = MZ_NEG; st.D = MZ_NEG; }   // row 0

        // ---- traceback byte: four steps per dword, one coalesced row per store
        tbword = __builtin_amdgcn_alignbyte(tbyte, tbword, 1);
        if ((t & 3) == 3) tbw[(t >> 2) * WAVE + lane] = tbword;
    }
    if ((Tend & 3) != 3)
        tbw[(Tend >> 2) * WAVE + lane] = tbword >> (8 * (3 - (Tend & 3)));

    if (lane == ((M - 1) & (WAVE - 1))) {
        b.final3[3 * p + 0] = st.C;
        b.final3[3 * p + 1] = st.D;
        b.final3[3 * p + 2] = st.I;
    }
}

// ------------------------------------------------------------------------------------------
// fast DP kernels (MZ_MODE_FAST, MZ_MODE_FASTT)
//
// Same wavefront as k_dp_wf64, for pairs the plan proved well-formed:
//   (1) the band is connected: LB[r] <= RB[r-1] for every row, so every in-band grid point has at
//       least one state reachable from (0,0);
//   (2) K*L*(open+extend+258)*(M+N+2) < 2^29, so every reachable state scores above -2^29.
// A state is unreachable only when its predecessor POINT lies outside the band (or is a row-0
// C/D state); the reference gives those states NEG or NEG minus/plus one step's terms, i.e. a
// value near -2^30, and no chain of them can form because a predecessor point inside the band
// always contributes a reachable state.  Every guard of mz_yama.c that is false only when the
// predecessor state is such a sentinel (all LB[r-1]/LB[r-2]/LB[r] tests and the bare "row > 1"
// tests) therefore only perturbs a value that loses every comparison it takes part in, and
// reachable values -- hence the traceback along the optimal path and the merged columns -- are
// unchanged.  The guards that do touch reachable states are kept: no gap-open when entering
// column 1 (C), none in column 0 or N (D), none on row M (I) -- the first two in the "edge"
// phases of the step loop only, the last through zeroed row vectors.
//
// Arithmetic: gap_open = g1*g2; row vectors carry g1, column vectors carry -g2, both as int16
// pairs, so that "x -= gap_open * (bilinear count form)" is one or two v_dot2c_i32_i16 with the
// running value as accumulator.
//
// TAG variant (MZ_MODE_FASTT, needs scores below 2^28 and K*max|score|*2 < 2^15): every state is
// held as 4*value + tag with tag(C)=2, tag(I)=1, tag(D)=0.  A candidate inherits the tag of the
// state it comes from, so m = max3(x,y,z) resolves the reference's tie order by itself (C wins
// ties; D beats I only if strictly greater, mz_yama.c:138-154) and m&3 IS the traceback flag:
// three instructions per pick instead of five, and no flag merge.  All increments are
// multiples of 4 (row vectors carry 2*g1, column vectors -2*g2, score vectors and counts 2x).
// The traceback byte then holds tags, which k_walk maps back (node = 2 - tag).
// ------------------------------------------------------------------------------------------
#define FRING 128                 // ring entries; six dword arrays (structure of arrays: conflict-free reads)
#define NEGT (-1610612736)        // TAG variant sentinel: -(3 << 29), a multiple of 4

struct FastRow {
    int lo, hi;
    int pC1x, pC2x, pC1y, pC2z;   // C-state row vectors
    int pI1, pI2x, pI2z;          // I-state row vectors (zero on row M)
    int pD1;                      // D-state x row vector
    int cD;                       // gap_open * nA * L      (D.x constant part and D.z penalty)
    int penDy;                    // gap_open * L * (nA - PA00)
    int extD;                     // gap_extend * L * nA
    int cDe, penDye;              // cD + extD, penDy + extD (interior steps fold the extension in)
    int w01, w23, w45;
};

template <bool TAG>
__device__ __forceinline__ void fast_stage_rows(int blk, int lane, const PairCtx &J, int *recs)
{
    const int rr = blk * WAVE + lane + 1;
    int4 *d = (int4 *)(recs + (((blk & 1) * WAVE) + lane) * REC_DW);
    if (rr > J.M) {
        d[0] = make_int4(MZ_BIG, MZ_BIG, 0, 0); d[1] = make_int4(0, 0, 0, 0);   // never active: lo == hi == BIG
        d[2] = make_int4(0, 0, 0, 0);           d[3] = make_int4(0, 0, 0, 0);
        return;
    }
    const int K = J.K, L = J.L;
    const uint8_t *col = J.A + (long long)(rr - 1) * K;
    unsigned cnt = 0;
    int dA = 0, a00 = 0, a11 = 0, other = 0;
    for (int i = 0; i < K; ++i) {
        const unsigned ch = col[i];
        const bool dash = ch == '-';
        const bool pdash = (rr > 1) ? (col[i - K] == '-') : false;
        const int cl = byte_class(ch);
        cnt += (cl < 4) ? (1u << (cl << 3)) : 0u;
        other += cl == 5;
        dA += dash;
        a00 += (!dash) & (!pdash);
        a11 += dash & pdash;
    }
    const int nA = K - dA;
    int cn[6] = { (int)(cnt & 0xff), (int)((cnt >> 8) & 0xff), (int)((cnt >> 16) & 0xff), (int)(cnt >> 24), dA, other };
    int w[6];
#pragma unroll
    for (int l = 0; l < 6; ++l) {
        int acc = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) acc += cn[k] * c_sc.S6[k * 6 + l];
        w[l] = TAG ? 2 * acc : acc;
    }
    const int V = TAG ? 4 : 1;                        // value scale
    const int go = c_sc.go, g1 = TAG ? 2 * c_sc.g1 : c_sc.g1;
    const bool last = rr >= J.M;                      // row M: trailing end-gaps open for free
    d[0] = make_int4(J.LB[rr], J.RB[rr], pack2(nA * g1, dA * g1), pack2(-a00 * g1, -a11 * g1));
    d[1] = make_int4(pack2((nA - a00) * g1, dA * g1), pack2(0, -dA * g1),
                     last ? 0 : pack2(0, K * g1), last ? 0 : pack2(0, -dA * g1));
    d[2] = make_int4(last ? 0 : pack2(0, -K * g1), pack2(-a00 * g1, 0), V * go * nA * L, V * go * L * (nA - a00));
    d[3] = make_int4(V * c_sc.ge * L * nA, pack2(w[0], w[1]), pack2(w[2], w[3]), pack2(w[4], w[5]));
}

__device__ __forceinline__ void fast_load_rec(FastRow &R, const int *src)
{
    const int4 *s = (const int4 *)src;
    const int4 a = s[0], b = s[1], c = s[2], d = s[3];
    R.lo = a.x; R.hi = a.y; R.pC1x = a.z; R.pC2x = a.w;
    R.pC1y = b.x; R.pC2z = b.y; R.pI1 = b.z; R.pI2x = b.w;
    R.pI2z = c.x; R.pD1 = c.y; R.cD = c.z; R.penDy = c.w;
    R.extD = d.x; R.w01 = d.y; R.w23 = d.z; R.w45 = d.w;
    R.cDe = c.z + d.x; R.penDye = c.w + d.x;
}

// ring[f * FRING + (col & (FRING-1))], f = 0..5: v1=(-g2*dB,-g2*nB)  v2=(-g2*PB11,-g2*PB00)
// cnt01 cnt23 cnt45  extI = gap_extend*K*nB          (TAG: g2, counts doubled; extI times 4)
template <bool TAG>
__device__ __forceinline__ void fast_stage_bcols(int first, int lane, const PairCtx &J, int *ring)
{
    const int cc = first + lane;
    int e0 = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0, e5 = 0;
    if (cc >= 1 && cc <= J.N) {
        const int L = J.L, g2 = TAG ? 2 * c_sc.g2 : c_sc.g2, cm = TAG ? 2 : 1;
        const uint8_t *col = J.B + (long long)(cc - 1) * L;
        unsigned cnt = 0;
        int dB = 0, b00 = 0, b11 = 0, other = 0;
        for (int j = 0; j < L; ++j) {
            const unsigned ch = col[j];
            const bool dash = ch == '-';
            const bool pdash = (cc > 1) ? (col[j - L] == '-') : false;
            const int cl = byte_class(ch);
            cnt += (cl < 4) ? (1u << (cl << 3)) : 0u;
            other += cl == 5;
            dB += dash;
            b00 += (!dash) & (!pdash);
            b11 += dash & pdash;
        }
        const int nB = L - dB;
        e0 = pack2(-g2 * dB, -g2 * nB);
        e1 = pack2(-g2 * b11, -g2 * b00);
        e2 = pack2(cm * (cnt & 0xff), cm * ((cnt >> 8) & 0xff));
        e3 = pack2(cm * ((cnt >> 16) & 0xff), cm * (cnt >> 24));
        e4 = pack2(cm * dB, cm * other);
        e5 = (TAG ? 4 : 1) * c_sc.ge * J.K * nB;
    }
    const int i = cc & (FRING - 1);
    ring[i] = e0; ring[FRING + i] = e1; ring[2 * FRING + i] = e2;
    ring[3 * FRING + i] = e3; ring[4 * FRING + i] = e4; ring[5 * FRING + i] = e5;
}

// one cell; EDGE = this step may contain cells in column 0, 1 or N
template <bool EDGE, bool TAG>
__device__ __forceinline__ Tri fast_cell(const FastRow &R, int c, int N, const int *ring, Tri left, Tri up, Tri dg, int &tbyte)
{
    const int i = c & (FRING - 1);
    const int v1 = ring[i], v2 = ring[FRING + i];
    const int c01 = ring[2 * FRING + i], c23 = ring[3 * FRING + i], c45 = ring[4 * FRING + i], extI = ring[5 * FRING + i];
    Tri o;
    int x, y, z;

    // I: x -= go*(K*nB - dA*PB00), y -= go*K*nB, z -= go*K*(nB - PB00)   (row vectors zero on row M)
    x = dot2(R.pI2x, v2, dot2(R.pI1, v1, left.C));
    y = dot2(R.pI1, v1, left.D);
    z = dot2(R.pI2z, v2, dot2(R.pI1, v1, left.I));
    int fi;
    if (TAG) { fi = max(max(x, y), z); o.I = ((fi & ~3) | 1) - extI; }
    else       o.I = pick(x, y, z, MZ_FD << 4, MZ_FI << 4, fi) - extI;

    // C
    x = dot2(R.pC2x, v2, dot2(R.pC1x, v1, dg.C));
    y = dot2(R.pC1y, v1, dg.D);
    z = dot2(R.pC2z, v2, dot2(R.pC1x, v1, dg.I));
    if (EDGE) {                                   // no gap-open when entering column 1 (mz_yama.c:173)
        const bool g = c > 1;
        x = g ? x : dg.C; y = g ? y : dg.D; z = g ? z : dg.I;
    }
    int fc;
    if (TAG) { fc = max(max(x, y), z); x = (fc & ~3) | 2; }
    else       x = pick(x, y, z, MZ_FD, MZ_FI, fc);
    o.C = dot2(R.w01, c01, dot2(R.w23, c23, dot2(R.w45, c45, x)));

    // D: gap_extend*L*nA is folded into the three penalties away from the edges
    int fd;
    if (EDGE) {                                   // no gap-open in the first or last column (mz_yama.c:211)
        const bool g = (c > 0) & (c < N);
        x = dot2(R.pD1, v1, up.C - R.cD);
        y = up.D - R.penDy;
        z = up.I - R.cD;
        x = g ? x : up.C; y = g ? y : up.D; z = g ? z : up.I;
        if (TAG) { fd = max(max(x, y), z); o.D = (fd & ~3) - R.extD; }
        else       o.D = pick(x, y, z, MZ_FD << 2, MZ_FI << 2, fd) - R.extD;
    } else {
        x = dot2(R.pD1, v1, up.C - R.cDe);
        y = up.D - R.penDye;
        z = up.I - R.cDe;
        if (TAG) { fd = max(max(x, y), z); o.D = fd & ~3; }
        else       o.D = pick(x, y, z, MZ_FD << 2, MZ_FI << 2, fd);
    }

    if (TAG) {
        // bits 0-1 tag(C pick), 2-3 tag(D pick), 4-5 tag(I pick); bits 6-7 are don't-care
        tbyte = (fc & 3) | ((fd & 3) << 2) | (fi << 4);
    } else {
        tbyte = fc | fd | fi;
    }
    return o;
}

struct FastState {
    FastRow R;
    Tri st, up, dg;
    int r;
    unsigned tbword;
    int rlo, lfin, tfin, cst;
};

template <bool EDGE, bool TAG>
__device__ __forceinline__ void fast_steps(FastState &S, int t0, int t1, int lane, const PairCtx &J,
                                           int *s_rec, int *s_ring, uint32_t *tbw)
{
    const int M = J.M, N = J.N;
    const int negC = TAG ? NEGT + 2 : MZ_NEG, negD = TAG ? NEGT : MZ_NEG, negI = TAG ? NEGT + 1 : MZ_NEG;
    // wave-uniform control state lives in SGPRs
    t0 = __builtin_amdgcn_readfirstlane(t0);
    t1 = __builtin_amdgcn_readfirstlane(t1);
    S.rlo = __builtin_amdgcn_readfirstlane(S.rlo);
    S.lfin = __builtin_amdgcn_readfirstlane(S.lfin);
    S.tfin = __builtin_amdgcn_readfirstlane(S.tfin);
    S.cst = __builtin_amdgcn_readfirstlane(S.cst);
    for (int t = t0; t <= t1; ++t) {
        S.dg = S.up;
        S.up.C = ror1(S.st.C); S.up.D = ror1(S.st.D); S.up.I = ror1(S.st.I);

        if (t > S.tfin) {                              // oldest row finished at step t-1
            const int rn = S.rlo + WAVE;
            if (lane == S.lfin) {
                S.r = rn;
                fast_load_rec(S.R, s_rec + ((((rn - 1) >> 6) & 1) * WAVE + lane) * REC_DW);
                S.st.C = negC; S.st.D = negD; S.st.I = negI;
            }
            if (S.lfin == 0) {
                fast_stage_rows<TAG>(((rn - 1) >> 6) + 1, lane, J, s_rec);
                __syncthreads();
            }
            S.rlo += 1;
            S.lfin = (S.lfin + 1) & (WAVE - 1);
            S.tfin = S.rlo > M ? MZ_BIG : S.rlo + __builtin_amdgcn_readlane(S.R.hi, S.lfin);
        }
        if (t - S.rlo > S.cst) {
            fast_stage_bcols<TAG>(S.cst + 1, lane, J, s_ring);
            S.cst += WAVE;
            __syncthreads();
        }

        const int c = t - S.r;
        int tbyte;
        const Tri nw = fast_cell<EDGE, TAG>(S.R, c, N, s_ring, S.st, S.up, S.dg, tbyte);
        const bool active = (unsigned)(c - S.R.lo) <= (unsigned)(S.R.hi - S.R.lo);
        S.st.C = active ? nw.C : negC;
        S.st.D = active ? nw.D : negD;
        S.st.I = active ? nw.I : negI;
        if (EDGE && S.rlo == 0 && lane == WAVE - 1) { S.st.C = negC; S.st.D = negD; }   // row 0

        S.tbword = __builtin_amdgcn_alignbyte(tbyte, S.tbword, 1);
        if ((t & 3) == 3) tbw[(t >> 2) * WAVE + lane] = S.tbword;
    }
}

template <bool TAG>
__device__ __forceinline__ void dp_fast_body(const mz_dev_batch &b, int p, int lane, int *s_rec, int *s_ring)
{
    PairCtx J;
    J.K = b.K[p]; J.L = b.L[p]; J.M = b.M[p]; J.N = b.N[p];
    J.A = b.poolA + b.offA[p]; J.B = b.poolB + b.offB[p];
    J.LB = b.poolLB + b.offBand[p]; J.RB = b.poolRB + b.offBand[p];
    const int M = J.M, N = J.N;
    uint32_t *tbw = b.tbw + b.offTb[p];

    fast_stage_rows<TAG>(0, lane, J, s_rec);
    fast_stage_rows<TAG>(1, lane, J, s_rec);
    fast_stage_bcols<TAG>(0, lane, J, s_ring);       // columns 0..63 (column 0 = zero entry)
    fast_stage_bcols<TAG>(WAVE, lane, J, s_ring);    // columns 64..127
    __syncthreads();

    FastState S;
    S.cst = 2 * WAVE - 1;
    S.st.C = TAG ? NEGT + 2 : MZ_NEG; S.st.D = TAG ? NEGT : MZ_NEG; S.st.I = TAG ? NEGT + 1 : MZ_NEG;
    if (lane == WAVE - 1) {                           // row 0 (mz_yama.c:83-94)
        S.r = 0;
        S.R.lo = 0; S.R.hi = J.RB[0];
        S.R.pC1x = S.R.pC2x = S.R.pC1y = S.R.pC2z = S.R.pI1 = S.R.pI2x = S.R.pI2z = S.R.pD1 = 0;
        S.R.cD = S.R.penDy = S.R.extD = S.R.cDe = S.R.penDye = 0; S.R.w01 = S.R.w23 = S.R.w45 = 0;
        S.st.C = TAG ? 2 : 0; S.st.D = 0; S.st.I = TAG ? 1 : 0;      // grid point (0,0): value 0
    } else {
        S.r = lane + 1;
        fast_load_rec(S.R, s_rec + lane * REC_DW);
    }
    S.rlo = 0; S.lfin = WAVE - 1;
    S.tfin = __builtin_amdgcn_readlane(S.R.hi, WAVE - 1);
    S.up.C = TAG ? NEGT + 2 : MZ_NEG; S.up.D = TAG ? NEGT : MZ_NEG; S.up.I = TAG ? NEGT + 1 : MZ_NEG;
    S.dg = S.up;
    S.tbword = 0;

    // three phases: steps that may touch column 0/1 (and row 0), the interior, steps that may touch column N
    const int Tend = M + N;
    int eLo = max(b.edgeLo[p], J.RB[0] + 1);          // row 0 is live until step RB[0]
    int eHi = b.edgeHi[p];
    eLo = min(eLo, Tend);
    eHi = max(eHi, eLo + 1);
    fast_steps<true, TAG>(S, 1, eLo, lane, J, s_rec, s_ring, tbw);
    fast_steps<false, TAG>(S, eLo + 1, min(eHi - 1, Tend), lane, J, s_rec, s_ring, tbw);
    fast_steps<true, TAG>(S, max(eHi, eLo + 1), Tend, lane, J, s_rec, s_ring, tbw);

    if ((Tend & 3) != 3)
        tbw[(Tend >> 2) * WAVE + lane] = S.tbword >> (8 * (3 - (Tend & 3)));
    if (lane == ((M - 1) & (WAVE - 1))) {             // (C,D,I) at (M,N), unscaled
        b.final3[3 * p + 0] = TAG ? S.st.C >> 2 : S.st.C;
        b.final3[3 * p + 1] = TAG ? S.st.D >> 2 : S.st.D;
        b.final3[3 * p + 2] = TAG ? S.st.I >> 2 : S.st.I;
    }
}

// ------------------------------------------------------------------------------------------
// tagged fast kernel body (MZ_MODE_FASTT) -- the tuned form of the fast kernel.
//
// On gfx950 every integer / dot / select VALU instruction of this loop occupies its SIMD for 4
// cycles and one wave per SIMD already saturates that pipe (measured: identical kernel time at
// 1, 2 and 3.5 waves per SIMD), so the step is built to minimise VALU instructions:
//  * states are 4*value + tag (tag C=2, I=1, D=0): one v_max3 resolves the reference's tie
//    order and its low two bits are the traceback flag;
//  * the flags of the three picks go into three 2-bit streams with one v_alignbit each and are
//    stored every 16 steps (layout: tbw[((t>>4)*3 + s)*64 + lane], s = 0 C, 1 D, 2 I picks);
//  * nB = L - dB is eliminated: column vectors uA = -2*g2*(dB, PB00), uB = -2*g2*(PB11, 0); the
//    constant parts of the penalties (4*go*K*L for I, 4*go*dA*L for C) are folded into the ring's
//    extension term and into the score row vector (sum of the class counts is L);
//  * gap_extend*L*nA is folded into the D penalties; the column counter is kept times 4 so that
//    it is the ring byte offset; the step loop is unrolled by two so that the "row above, one
//    step ago" registers alternate roles instead of being copied.
// Steps that can touch column 0/1/N, row 0 or row M run the EDGE form, which undoes the folds
// where the reference charges no gap-open.
// ------------------------------------------------------------------------------------------
#define TREC 20                   // dwords per staged row record

struct TagRow {
    int lo4, wid4;                // 4*LB[r], 4*(RB[r]-LB[r])
    int rIx, rIy, rIz;            // I-state row vectors against uA (zero on row M)
    int rCxA, rCxB, rCy, rCz;     // C-state row vectors
    int rDx;                      // D-state x row vector against uA
    int cDe, penDye;              // 4*(go*nA*L + ge*L*nA), 4*(go*L*(nA-PA00) + ge*L*nA)
    int w01, w23, w45;            // 2*(w[k] - go*dA)
    int kC, extD, kIfix;          // EDGE only: 4*go*dA*L, 4*ge*L*nA, (row 0 or M) ? 4*go*K*L : 0
};

__device__ __forceinline__ void tag_stage_rows(int blk, int lane, const PairCtx &J, int *recs)
{
    const int rr = blk * WAVE + lane + 1;
    int4 *d = (int4 *)(recs + (((blk & 1) * WAVE) + lane) * TREC);
    if (rr > J.M) {
        d[0] = make_int4(MZ_BIG, 0, 0, 0);            // never active: column counter never reaches BIG
        d[1] = d[2] = d[3] = d[4] = make_int4(0, 0, 0, 0);
        return;
    }
    const int K = J.K, L = J.L;
    const uint8_t *col = J.A + (long long)(rr - 1) * K;
    unsigned cnt = 0;
    int dA = 0, a00 = 0, a11 = 0, other = 0;
    for (int i = 0; i < K; ++i) {
        const unsigned ch = col[i];
        const bool dash = ch == '-';
        const bool pdash = (rr > 1) ? (col[i - K] == '-') : false;
        const int cl = byte_class(ch);
        cnt += (cl < 4) ? (1u << (cl << 3)) : 0u;
        other += cl == 5;
        dA += dash;
        a00 += (!dash) & (!pdash);
        a11 += dash & pdash;
    }
    const int nA = K - dA;
    const int go = c_sc.go, ge = c_sc.ge, g1 = 2 * c_sc.g1;
    int cn[6] = { (int)(cnt & 0xff), (int)((cnt >> 8) & 0xff), (int)((cnt >> 16) & 0xff), (int)(cnt >> 24), dA, other };
    int w[6];
#pragma unroll
    for (int l = 0; l < 6; ++l) {
        int acc = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) acc += cn[k] * c_sc.S6[k * 6 + l];
        w[l] = 2 * (acc - go * dA);                   // -go*dA per unit count: sums to -go*dA*L over a column
    }
    const bool last = rr >= J.M;
    const int lo = J.LB[rr], hi = J.RB[rr];
    d[0] = make_int4(4 * lo, 4 * (hi - lo),
                     last ? 0 : pack2(-K * g1, -dA * g1), last ? 0 : pack2(-K * g1, 0));
    d[1] = make_int4(last ? 0 : pack2(-K * g1, -K * g1),
                     pack2((nA - dA) * g1, -a11 * g1), pack2(-a00 * g1, 0), pack2((nA - a00 - dA) * g1, 0));
    d[2] = make_int4(pack2((nA - dA) * g1, -dA * g1), pack2(-a00 * g1, 0),
                     4 * (go + ge) * nA * L, 4 * (go * L * (nA - a00) + ge * L * nA));
    d[3] = make_int4(pack2(w[0], w[1]), pack2(w[2], w[3]), pack2(w[4], w[5]), 4 * go * dA * L);
    d[4] = make_int4(4 * ge * L * nA, last ? 4 * go * K * L : 0, 0, 0);
}

__device__ __forceinline__ void tag_load_rec(TagRow &R, const int *src)
{
    const int4 *s = (const int4 *)src;
    const int4 a = s[0], b = s[1], c = s[2], d = s[3], e = s[4];
    R.lo4 = a.x; R.wid4 = a.y; R.rIx = a.z; R.rIy = a.w;
    R.rIz = b.x; R.rCxA = b.y; R.rCxB = b.z; R.rCy = b.w;
    R.rCz = c.x; R.rDx = c.y; R.cDe = c.z; R.penDye = c.w;
    R.w01 = d.x; R.w23 = d.y; R.w45 = d.z; R.kC = d.w;
    R.extD = e.x; R.kIfix = e.y;
}

// ring[f*FRING + (col & 127)]: uA=-2g2*(dB,PB00)  uB=-2g2*(PB11,0)  2*cnt01 2*cnt23 2*cnt45
//                              xI = 4*ge*K*nB + 4*go*K*L - 1
__device__ __forceinline__ void tag_stage_bcols(int first, int lane, const PairCtx &J, int *ring)
{
    const int cc = first + lane;
    int e0 = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0, e5 = 4 * c_sc.go * J.K * J.L - 1;
    if (cc >= 1 && cc <= J.N) {
        const int L = J.L, g2 = 2 * c_sc.g2;
        const uint8_t *col = J.B + (long long)(cc - 1) * L;
        unsigned cnt = 0;
        int dB = 0, b00 = 0, b11 = 0, other = 0;
        for (int j = 0; j < L; ++j) {
            const unsigned ch = col[j];
            const bool dash = ch == '-';
            const bool pdash = (cc > 1) ? (col[j - L] == '-') : false;
            const int cl = byte_class(ch);
            cnt += (cl < 4) ? (1u << (cl << 3)) : 0u;
            other += cl == 5;
            dB += dash;
            b00 += (!dash) & (!pdash);
            b11 += dash & pdash;
        }
        e0 = pack2(-g2 * dB, -g2 * b00);
        e1 = pack2(-g2 * b11, 0);
        e2 = pack2(2 * (cnt & 0xff), 2 * ((cnt >> 8) & 0xff));
        e3 = pack2(2 * ((cnt >> 16) & 0xff), 2 * (cnt >> 24));
        e4 = pack2(2 * dB, 2 * other);
        e5 += 4 * c_sc.ge * J.K * (L - dB);
    }
    const int i = cc & (FRING - 1);
    ring[i] = e0; ring[FRING + i] = e1; ring[2 * FRING + i] = e2;
    ring[3 * FRING + i] = e3; ring[4 * FRING + i] = e4; ring[5 * FRING + i] = e5;
}

struct TagState {
    TagRow R;
    Tri st;                       // this lane's latest cell (tagged), sentinel while idle
    Tri u0, u1;                   // row above: alternately "one step ago" / "two steps ago"
    int r4;                       // 4 * current row
    unsigned wC, wD, wI;          // 2-bit flag streams
    int rlo, lfin, tfin, cst;
};

// one step; Un receives the row-above triple of this step, Uo holds the one of the step before
template <bool EDGE>
__device__ __forceinline__ void tag_step(TagState &S, Tri &Un, const Tri &Uo, int t, int lane, const PairCtx &J,
                                         int *s_rec, const int *s_ring, int *s_ring_w, uint32_t *tbw)
{
    Un.C = ror1(S.st.C); Un.D = ror1(S.st.D); Un.I = ror1(S.st.I);

    if (t > S.tfin) {                                  // oldest row finished at step t-1
        const int rn = S.rlo + WAVE;
        if (lane == S.lfin) {
            S.r4 = 4 * rn;
            tag_load_rec(S.R, s_rec + ((((rn - 1) >> 6) & 1) * WAVE + lane) * TREC);
            S.st.C = NEGT + 2; S.st.D = NEGT; S.st.I = NEGT + 1;
        }
        if (S.lfin == 0) {
            tag_stage_rows(((rn - 1) >> 6) + 1, lane, J, s_rec);
            __syncthreads();
        }
        S.rlo += 1;
        S.lfin = (S.lfin + 1) & (WAVE - 1);
        const int l4 = __builtin_amdgcn_readlane(S.R.lo4, S.lfin), w4 = __builtin_amdgcn_readlane(S.R.wid4, S.lfin);
        S.tfin = S.rlo > J.M ? MZ_BIG : S.rlo + ((l4 + w4) >> 2);
    }
    if (t - S.rlo > S.cst) {
        tag_stage_bcols(S.cst + 1, lane, J, s_ring_w);
        S.cst += WAVE;
        __syncthreads();
    }

    const TagRow &R = S.R;
    const int c4 = 4 * t - S.r4;                       // 4 * column; also the ring byte offset
    const int *e = (const int *)((const char *)s_ring + (c4 & (4 * FRING - 4)));
    const int uA = e[0], uB = e[FRING], c01 = e[2 * FRING], c23 = e[3 * FRING], c45 = e[4 * FRING], xI = e[5 * FRING];
    const Tri left = S.st, &up = Un, &dg = Uo;
    int x, y, z, mI, mC, mD, nI, nC, nD;

    // I: candidates inherit tags 2 / 0 / 1 from C / D / I of (r, c-1)
    x = dot2(R.rIx, uA, left.C);
    y = dot2(R.rIy, uA, left.D);
    z = dot2(R.rIz, uA, left.I);
    mI = max(max(x, y), z);
    nI = (mI & ~3) - xI;
    if (EDGE) nI += R.kIfix;                           // rows 0 and M pay no gap-open (mz_yama.c:123)

    // C
    x = dot2(R.rCxB, uB, dot2(R.rCxA, uA, dg.C));
    y = dot2(R.rCy, uA, dg.D);
    z = dot2(R.rCz, uA, dg.I);
    if (EDGE) {                                        // no gap-open entering column 1 (mz_yama.c:173)
        const bool g = c4 > 4;
        x = g ? x : dg.C; y = g ? y : dg.D; z = g ? z : dg.I;
        mC = max(max(x, y), z);
        nC = dot2(R.w01, c01, dot2(R.w23, c23, dot2(R.w45, c45, (mC & ~3) | 2))) + (g ? 0 : R.kC);
    } else {
        mC = max(max(x, y), z);
        nC = dot2(R.w01, c01, dot2(R.w23, c23, dot2(R.w45, c45, (mC & ~3) | 2)));
    }

    // D
    if (EDGE) {                                        // none in column 0 or N (mz_yama.c:211)
        const bool g = (c4 > 0) & (c4 < 4 * J.N);
        x = dot2(R.rDx, uA, up.C - (R.cDe - R.extD));
        y = up.D - (R.penDye - R.extD);
        z = up.I - (R.cDe - R.extD);
        x = g ? x : up.C; y = g ? y : up.D; z = g ? z : up.I;
        mD = max(max(x, y), z);
        nD = (mD & ~3) - R.extD;
    } else {
        x = dot2(R.rDx, uA, up.C - R.cDe);
        y = up.D - R.penDye;
        z = up.I - R.cDe;
        mD = max(max(x, y), z);
        nD = mD & ~3;
    }

    // idle lanes keep publishing the sentinel (the reference's "unwritten dp[col] is still MININT")
    const bool active = (unsigned)(c4 - R.lo4) <= (unsigned)R.wid4;
    S.st.C = active ? nC : NEGT + 2;
    S.st.D = active ? nD : NEGT;
    S.st.I = active ? nI : NEGT + 1;
    if (EDGE && S.rlo == 0 && lane == WAVE - 1) { S.st.C = NEGT + 2; S.st.D = NEGT; }   // row 0: C = D = NEG

    S.wC = __builtin_amdgcn_alignbit(mC, S.wC, 2);
    S.wD = __builtin_amdgcn_alignbit(mD, S.wD, 2);
    S.wI = __builtin_amdgcn_alignbit(mI, S.wI, 2);
    if ((t & 15) == 15) {
        uint32_t *g = tbw + (t >> 4) * (3 * WAVE) + lane;
        g[0] = S.wC; g[WAVE] = S.wD; g[2 * WAVE] = S.wI;
    }
}

template <bool EDGE>
__device__ __forceinline__ void tag_steps(TagState &S, int t0, int t1, int lane, const PairCtx &J,
                                          int *s_rec, int *s_ring, uint32_t *tbw)
{
    t0 = __builtin_amdgcn_readfirstlane(t0);
    t1 = __builtin_amdgcn_readfirstlane(t1);
    S.rlo = __builtin_amdgcn_readfirstlane(S.rlo);
    S.lfin = __builtin_amdgcn_readfirstlane(S.lfin);
    S.tfin = __builtin_amdgcn_readfirstlane(S.tfin);
    S.cst = __builtin_amdgcn_readfirstlane(S.cst);
    int t = t0;
    if ((t & 1) && t <= t1) {                          // odd steps write u1, even steps write u0
        tag_step<EDGE>(S, S.u1, S.u0, t, lane, J, s_rec, s_ring, s_ring, tbw);
        ++t;
    }
    for (; t + 1 <= t1; t += 2) {
        tag_step<EDGE>(S, S.u0, S.u1, t, lane, J, s_rec, s_ring, s_ring, tbw);
        tag_step<EDGE>(S, S.u1, S.u0, t + 1, lane, J, s_rec, s_ring, s_ring, tbw);
    }
    if (t <= t1)
        tag_step<EDGE>(S, S.u0, S.u1, t, lane, J, s_rec, s_ring, s_ring, tbw);
}

__device__ __forceinline__ void dp_tag_body(const mz_dev_batch &b, int p, int lane, int *s_rec, int *s_ring)
{
    PairCtx J;
    J.K = b.K[p]; J.L = b.L[p]; J.M = b.M[p]; J.N = b.N[p];
    J.A = b.poolA + b.offA[p]; J.B = b.poolB + b.offB[p];
    J.LB = b.poolLB + b.offBand[p]; J.RB = b.poolRB + b.offBand[p];
    const int M = J.M, N = J.N;
    uint32_t *tbw = b.tbw + b.offTb[p];

    tag_stage_rows(0, lane, J, s_rec);
    tag_stage_rows(1, lane, J, s_rec);
    tag_stage_bcols(0, lane, J, s_ring);
    tag_stage_bcols(WAVE, lane, J, s_ring);
    __syncthreads();

    TagState S;
    S.cst = 2 * WAVE - 1;
    S.st.C = NEGT + 2; S.st.D = NEGT; S.st.I = NEGT + 1;
    if (lane == WAVE - 1) {                           // row 0 (mz_yama.c:83-94): every vector zero, no opens
        S.r4 = 0;
        S.R.lo4 = 0; S.R.wid4 = 4 * J.RB[0];
        S.R.rIx = S.R.rIy = S.R.rIz = S.R.rCxA = S.R.rCxB = S.R.rCy = S.R.rCz = S.R.rDx = 0;
        S.R.cDe = S.R.penDye = 0; S.R.w01 = S.R.w23 = S.R.w45 = 0;
        S.R.kC = S.R.extD = 0; S.R.kIfix = 4 * c_sc.go * J.K * J.L;
        S.st.C = 2; S.st.D = 0; S.st.I = 1;           // grid point (0,0): value 0 in all three states
    } else {
        S.r4 = 4 * (lane + 1);
        tag_load_rec(S.R, s_rec + lane * TREC);
    }
    S.rlo = 0; S.lfin = WAVE - 1;
    S.tfin = J.RB[0];
    S.u0.C = NEGT + 2; S.u0.D = NEGT; S.u0.I = NEGT + 1;
    S.u1 = S.u0;
    S.wC = S.wD = S.wI = 0;

    const int Tend = M + N;
    int eLo = max(b.edgeLo[p], J.RB[0] + 1);          // row 0 is live until step RB[0]
    int eHi = b.edgeHi[p];                            // first step that can touch column N or row M
    eLo = min(eLo, Tend);
    eHi = max(eHi, eLo + 1);
    tag_steps<true>(S, 1, eLo, lane, J, s_rec, s_ring, tbw);
    tag_steps<false>(S, eLo + 1, min(eHi - 1, Tend), lane, J, s_rec, s_ring, tbw);
    tag_steps<true>(S, max(eHi, eLo + 1), Tend, lane, J, s_rec, s_ring, tbw);

    if ((Tend & 15) != 15) {                          // flush the partial group
        const int sh = 2 * (15 - (Tend & 15));
        uint32_t *g = tbw + (Tend >> 4) * (3 * WAVE) + lane;
        g[0] = S.wC >> sh; g[WAVE] = S.wD >> sh; g[2 * WAVE] = S.wI >> sh;
    }
    if (lane == ((M - 1) & (WAVE - 1))) {             // (C,D,I) at (M,N), unscaled
        b.final3[3 * p + 0] = S.st.C >> 2;
        b.final3[3 * p + 1] = S.st.D >> 2;
        b.final3[3 * p + 2] = S.st.I >> 2;
    }
}

// ------------------------------------------------------------------------------------------
// row-parallel kernel (MZ_MODE_ROW) -- for bands at most 63 columns wide.
//
// The anti-diagonal wavefront keeps only about half of the 64 lanes busy on a band of width 2R+1
// (an anti-diagonal crosses it in R+1 cells).  Here a lane owns a COLUMN (lane = column & 63, a ring
// over the columns; a lane moves on to column c+64 when c drops out of the band on the left) and the
// wave computes one whole ROW of the band per iteration, so ~(2R+1)/64 of the lanes work.  C(r,c)
// and D(r,c) depend on row r-1 only (one DPP rotate for the diagonal).  I(r,c) depends on
// I(r,c-1) -- a max-plus recurrence along the row:
//     nI_c = max(J0_c, nI_{c-1} + e_c),   J0_c = open from C/D of (r,c-1),
//     e_c  = -4K(go*(nB_c - PB00_c) + ge*nB_c)   (tag_step's  z-candidate minus xI, tag kept at 1).
// e_c is a function of the column alone (rows 1..M-1; row M and row 0 pay no gap-open: e_c =
// -4K*ge*nB_c), so with the prefix sums P_c = e_1 + ... + e_c staged once per column,
//     nI_c = P_c + max_{k<=c} (J0_k - P_k),
// a plain prefix MAXIMUM over the lanes: six DPP steps (row_shr 1,2,4,8, row_bcast 15, 31).  The ring
// makes the band start at lane s = LB[r] & 63: lanes >= s (the lower ring period) are lifted by 2^30
// so that they ignore lanes < s, and lanes < s (the wrapped tail) take the lower period's total from
// lane 63.  Values are exact integers, so nI and all three pick flags (computed afterwards from the
// final neighbours, as tag_step does) equal the wavefront kernel's.
// Scores and running sums are re-based as the kernel advances (row_rebase, row_pre), so the tagged int32
// states hold alignments of any length; the conditions (k_plan) are a connected band, int16 row vectors,
// RB[r]-LB[r] <= 62 for every row and one 64-row WINDOW of scores within 2^27 beside the 2^30 lift.
// ROT variants (MZ_MODE_ROWR/COLR, more rows per block): no lift; the candidates are rotated with
// ds_bpermute so that the band starts at lane 0, scanned and rotated back.
// Traceback: 2-bit tag streams per lane as in FASTT, but one entry per ROW:
//     tbw[((r>>4)*3 + s)*64 + (c & 63)], bits 2*(r&15).
// Everything is built inside the kernel from the raw column bytes: row records (16 dwords) 64 rows at a
// time into a 4 KB LDS block that every lane reads back (broadcast) one row ahead; column records (8 dwords)
// 64 columns at a time into a 128-entry LDS ring from which a lane re-arms.  k_rowprep only scatters the
// transposed band bounds of the COL pairs.  (Earlier versions read prep records from HBM with scalar loads:
// see DESIGN.md section 4.2 for why that lost.)
// ------------------------------------------------------------------------------------------
#define ROW_LIFT (1 << 30)

typedef int int8v __attribute__((ext_vector_type(8)));


// COL (the transposed problem): the band column by column -- tlo[c] = first row with RB[r] >= c, thi[c] =
// last row with LB[r] <= c for c = 0..N, scattered from the rows (both arrays are monotone, so the work is
// M + N) into the pair's prep slice.  ROW pairs need no prep.
__global__ __launch_bounds__(WAVE) void k_rowprep(mz_dev_batch b, int first, int count)
{
    const int p = first + blockIdx.x, lane = threadIdx.x;
    if (b.status[p] != MZ_OK || (b.mode[p] != MZ_MODE_COL && b.mode[p] != MZ_MODE_COLR)) return;
    const int M = b.M[p], N = b.N[p];
    const int *LB = b.poolLB + b.offBand[p], *RB = b.poolRB + b.offBand[p];
    int *tlo = (int *)(b.prep + b.offPrep[p]), *thi = tlo + (N + 1);
    for (int r = lane; r <= M; r += WAVE) {
        const int c1 = RB[r], c0 = r > 0 ? RB[r - 1] + 1 : 0;
        for (int cc = c0; cc <= c1; ++cc) tlo[cc] = r;
        const int d0 = LB[r], d1 = r < M ? LB[r + 1] - 1 : N;
        for (int cc = d0; cc <= d1; ++cc) thi[cc] = r;
    }
}

// The row record: 16 dwords, built for 64 rows at a time by the 64 lanes (row_stage_rows) in LDS and read back
// by every lane of the wave (broadcast reads) one row ahead of its use; interior rows read 12 dwords.
//   a = {lo32, wid32, rIx, rCxA}  b = {rCxB (= rDx), rCy, rCz, cDe}  c = {penDye - cDe, w01, w23, w45}
//   d = {rIy, rIz, dA | nA << 8 | last << 16, 4*(LB[r]&63)}    (rIy, rIz are pair constants except on the last row)
// History: the records were first read with scalar loads from a prep pass in HBM (SGPR operands, no LDS).  That
// made the loop wait ~0.8 us per scalar-cache miss, cost a 3 GB prep kernel per batch competing with the DP, and
// -- with compact records derived by scalar code -- saturated the CU's shared scalar unit.
struct RowRec { int4 a, b, c, d; };
#define R_lo32(R)   ((R).a.x)       /* 32 * LB[r]: the column counter is kept times 32, the ring byte offset */
#define R_wid32(R)  ((R).a.y)       /* 32 * (RB[r] - LB[r]) */
#define R_rIx(R)    ((R).a.z)
#define R_rCxA(R)   ((R).a.w)
#define R_rCxB(R)   ((R).b.x)
#define R_rDx(R)    ((R).b.x)
#define R_rCy(R)    ((R).b.y)
#define R_rCz(R)    ((R).b.z)
#define R_cDe(R)    ((R).b.w)
#define R_dDy(R)    ((R).c.x)       /* penDye - cDe */
#define R_w01(R)    ((R).c.y)
#define R_w23(R)    ((R).c.z)
#define R_w45(R)    ((R).c.w)
#define R_dA(R)     ((R).d.z & 0xff)
#define R_nA(R)     (((R).d.z >> 8) & 0xff)
#define R_last(R)   ((R).d.z >> 16)

struct RowSrc { const uint8_t *A; const int *lo, *hi; int K, L, M, go, ge, g1; };   // lo/hi: band of row r (ROW: LB/RB; COL: tlo/thi)

// records of rows blk*64+1 .. blk*64+64 (dead beyond M) into s_rec[lane]
__device__ __forceinline__ void row_stage_rows(int blk, int lane, const RowSrc &Z, int4 *s_rec)
{
    const int rr = blk * WAVE + lane + 1;
    int4 *d = s_rec + lane * (RREC / 4);
    if (rr > Z.M) {
        d[0] = make_int4(MZ_BIG, 0, 0, 0);
        d[1] = d[2] = d[3] = make_int4(0, 0, 0, 0);
        return;
    }
    const int K = Z.K, L = Z.L, go = Z.go, ge = Z.ge, g1 = Z.g1;
    const uint8_t *col = Z.A + (long long)(rr - 1) * K;
    unsigned cnt = 0;
    int dA = 0, a00 = 0, a11 = 0, other = 0;
    for (int i = 0; i < K; ++i) {
        const unsigned ch = col[i];
        const bool dash = ch == '-';
        const bool pdash = (rr > 1) ? (col[i - K] == '-') : false;
        const int cl = byte_class(ch);
        cnt += (cl < 4) ? (1u << (cl << 3)) : 0u;
        other += cl == 5;
        dA += dash;
        a00 += (!dash) & (!pdash);
        a11 += dash & pdash;
    }
    const int nA = K - dA;
    int cn[6] = { (int)(cnt & 0xff), (int)((cnt >> 8) & 0xff), (int)((cnt >> 16) & 0xff), (int)(cnt >> 24), dA, other };
    int w[6];
#pragma unroll
    for (int l = 0; l < 6; ++l) {
        int acc = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) acc += cn[k] * c_sc.S6[k * 6 + l];
        w[l] = 2 * (acc - go * dA);                    // -go*dA per unit count: sums to -go*dA*L over a column
    }
    const bool last = rr >= Z.M;
    const int lo = Z.lo[rr], hi = Z.hi[rr];
    const int cDe = 4 * (go + ge) * nA * L;
    d[0] = make_int4(32 * lo, 32 * (hi - lo), last ? 0 : pack2(-K * g1, -dA * g1), pack2((nA - dA) * g1, -a11 * g1));
    d[1] = make_int4(pack2(-a00 * g1, 0), pack2((nA - a00 - dA) * g1, 0), pack2((nA - dA) * g1, -dA * g1), cDe);
    d[2] = make_int4(4 * (go * L * (nA - a00) + ge * L * nA) - cDe, pack2(w[0], w[1]), pack2(w[2], w[3]), pack2(w[4], w[5]));
    d[3] = make_int4(last ? 0 : pack2(-K * g1, 0), last ? 0 : pack2(-K * g1, -K * g1), dA | (nA << 8) | ((int)last << 16), 4 * (lo & (WAVE - 1)));
}

// the record of row r from the staged block (every lane reads the same address: LDS broadcast)
template <bool FULL>
__device__ __forceinline__ void row_rec_read(RowRec &R, const int4 *s_rec, int r)
{
    const int4 *s = s_rec + ((r - 1) & (WAVE - 1)) * (RREC / 4);
    R.a = s[0]; R.b = s[1]; R.c = s[2];
    if (FULL) R.d = s[3];
}
// ROT kernels also need the band's start lane in interior rows
__device__ __forceinline__ int row_rec_s4(const int4 *s_rec, int r) { return s_rec[((r - 1) & (WAVE - 1)) * (RREC / 4) + 3].w; }

struct RowState {
    int c32;                      // 32 * the column this lane holds
    int uA, uB, c01, c23, c45;
    int xIPl, Pl;                 // xI + P and P (running sum), both minus 2^30 while the lane is in the lower ring period
    int Q;                        // running sum for rows 0 and M
    Tri p;                        // row r-1 at this column (tagged), sentinel outside the band
    Tri l;                        // the same, one lane to the left (= ror1(p) before any re-arm)
    unsigned wC, wD, wI;
};

// Column records of columns first .. first+63 into the LDS ring (lift = 2^30 if they belong to the ring period the
// band's left edge is in, see row_pre), built from the raw bytes of B (column vectors,
// class counts, and the running sums P, Q of the max-plus recurrence, carried from chunk to chunk): ~100
// instructions per 64 columns, cheaper than a round trip of 32 bytes per column through HBM.
struct ColSrc { const uint8_t *B; int L, N, K4go, K4ge, g2, xI0, carryP, carryQ; };   // K4go = 4*K*go, xI0 = 4*go*K*L - TI;
                                                  // carryP/Q: running sums at the last staged column, relative to the current base (row_pre)
__device__ __forceinline__ void row_stage_cols(int first, int lift, int lane, ColSrc &Z, int4 *ring)
{
    const int cc = first + lane;
    int e0 = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0, e5 = Z.xI0, eP = 0, eQ = 0;
    if (cc >= 1 && cc <= Z.N) {
        const int L = Z.L;
        const uint8_t *col = Z.B + (long long)(cc - 1) * L;
        unsigned cnt = 0;
        int dB = 0, b00 = 0, b11 = 0, other = 0;
        for (int j = 0; j < L; ++j) {
            const unsigned ch = col[j];
            const bool dash = ch == '-';
            const bool pdash = (cc > 1) ? (col[j - L] == '-') : false;
            const int cl = byte_class(ch);
            cnt += (cl < 4) ? (1u << (cl << 3)) : 0u;
            other += cl == 5;
            dB += dash;
            b00 += (!dash) & (!pdash);
            b11 += dash & pdash;
        }
        const int nB = L - dB;
        e0 = pack2(-Z.g2 * dB, -Z.g2 * b00);
        e1 = pack2(-Z.g2 * b11, 0);
        e2 = pack2(2 * (cnt & 0xff), 2 * ((cnt >> 8) & 0xff));
        e3 = pack2(2 * ((cnt >> 16) & 0xff), 2 * (cnt >> 24));
        e4 = pack2(2 * dB, 2 * other);
        e5 += Z.K4ge * nB;
        eP = -(Z.K4go * (nB - b00) + Z.K4ge * nB);
        eQ = -Z.K4ge * nB;
    }
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        const int y = __shfl_up(eP, o), z = __shfl_up(eQ, o);
        if (lane >= o) { eP += y; eQ += z; }
    }
    eP += Z.carryP; eQ += Z.carryQ;
    Z.carryP = __builtin_amdgcn_readlane(eP, WAVE - 1);
    Z.carryQ = __builtin_amdgcn_readlane(eQ, WAVE - 1);
    ring[2 * (cc & (FRING - 1))] = make_int4(e0, e1, e2, e3);
    ring[2 * (cc & (FRING - 1)) + 1] = make_int4(e5 + eP - lift, eP - lift, e4, eQ);   // xI + P, P (both lifted or not), c45, Q
}

// column data of column c32/32 from the ring (its xI+P and P come lifted or not, as the ring holds them now)
__device__ __forceinline__ void row_load_col(RowState &S, const int4 *ring)
{
    const int4 *e = (const int4 *)((const char *)ring + (S.c32 & (32 * FRING - 32)));
    const int4 x = e[0], y = e[1];
    S.uA = x.x; S.uB = x.y; S.c01 = x.z; S.c23 = x.w;
    S.xIPl = y.x; S.Pl = y.y; S.c45 = y.z; S.Q = y.w;
}

// inclusive prefix maximum over the 64 lanes (lane order)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dpp_max_step(int g)
{
    // old = INT_MIN is the identity of max: lanes without a source keep g, and the DPP folds into v_max_i32
    const int t = __builtin_amdgcn_update_dpp((int)0x80000000, g, CTRL, ROWMASK, 0xF, false);
    return max(g, t);
}
__device__ __forceinline__ int prefix_max64(int g)
{
    g = dpp_max_step<0x111, 0xF>(g);                  // row_shr:1
    g = dpp_max_step<0x112, 0xF>(g);                  // row_shr:2
    g = dpp_max_step<0x114, 0xF>(g);                  // row_shr:4
    g = dpp_max_step<0x118, 0xF>(g);                  // row_shr:8
    g = dpp_max_step<0x142, 0xA>(g);                  // row_bcast:15 into rows 1 and 3
    g = dpp_max_step<0x143, 0xC>(g);                  // row_bcast:31 into rows 2 and 3
    return g;
}

struct RowCtx { int K, L, N32, KL4go, rIy, rIz; };     // rIy, rIz: the I-state vectors of every row but the last

// one row of the band; EDGE = the row can hold column 0, 1 or N, or is row M (COL: or is row 1).
// COL = transposed problem: the D slot holds the reference's I state (tag 1) and the I slot its D state
// (tag 0), so that one max still resolves the reference's tie order C > I > D.
// ROT: no ring lift; the open candidates are rotated so that the band starts at lane 0 (s4 = 4 * start lane),
// scanned, and rotated back
template <bool EDGE, bool COL, bool ROT>
__device__ __forceinline__ void row_step(RowState &S, const RowRec &R, int s4, int r, const RowCtx &J, const int4 *s_ring, uint32_t *tbw, int lane)
{
    constexpr int TD = COL ? 1 : 0, TI = COL ? 0 : 1;
    const Tri dg = S.l;                                // (r-1, c-1), rotated at the end of row r-1
    // column left the band: take column c+64.  Neither (r-1, c+64) nor (r-1, c+63) was in row r-1's band
    // (its width is at most 63), so a re-armed lane's C and D are sentinels in this row: instead of resetting
    // its "up" and "diagonal" registers, the lane is masked out of C and D below.
    const bool stays = S.c32 >= R_lo32(R);             // (one compare: the re-arm below runs on its complement)
    if (!stays) {
        S.c32 += 32 * WAVE;
        row_load_col(S, s_ring);
    }
    const int c32 = S.c32, uA = S.uA, uB = S.uB;
    const Tri &up = S.p;
    int x, y, z, mI, mC, mD, nI, nC, nD;

    // C
    x = dot2(R_rCxB(R), uB, dot2(R_rCxA(R), uA, dg.C));
    y = dot2(R_rCy(R), uA, dg.D);
    z = dot2(R_rCz(R), uA, dg.I);
    if (EDGE) {                                        // no gap-open entering column 1 (mz_yama.c:173)
        const bool g = COL ? (r > 1) : (c32 > 32);     // (transposed: the reference's column 1 is row 1)
        const int kC = 4 * c_sc.go * R_dA(R) * J.L;
        x = g ? x : dg.C; y = g ? y : dg.D; z = g ? z : dg.I;
        mC = max(max(x, y), z);
        nC = dot2(R_w01(R), S.c01, dot2(R_w23(R), S.c23, dot2(R_w45(R), S.c45, (mC & ~3) | 2))) + (g ? 0 : kC);
    } else {
        mC = max(max(x, y), z);
        nC = dot2(R_w01(R), S.c01, dot2(R_w23(R), S.c23, dot2(R_w45(R), S.c45, (mC & ~3) | 2)));
    }
    // D: the penalty cDe common to the three candidates is taken off after the pick
    if (EDGE) {                                        // none in column 0 or N (mz_yama.c:211)
        const bool g = (c32 > 0) & (c32 < J.N32);
        const int extD = 4 * c_sc.ge * J.L * R_nA(R);
        x = dot2_keep(R_rDx(R), uA, up.C);
        y = up.D - R_dDy(R);
        x = g ? x : up.C + (R_cDe(R) - extD); y = g ? y : up.D + (R_cDe(R) - extD);
        z = g ? up.I : up.I + (R_cDe(R) - extD);
        mD = max(max(x, y), z);
        nD = ((mD & ~3) | TD) - R_cDe(R);
    } else {
        x = dot2_keep(R_rDx(R), uA, up.C);
        y = up.D - R_dDy(R);
        mD = max(max(x, y), up.I);
        nD = ((mD & ~3) | TD) - R_cDe(R);
    }
    const bool active = (unsigned)(c32 - R_lo32(R)) <= (unsigned)R_wid32(R);
    const bool activeCD = active && stays;
    nC = activeCD ? nC : NEGT + 2;
    nD = activeCD ? nD : NEGT + TD;

    // I: open candidates from the finished C / D of (r, c-1), then the prefix maximum along the row.
    // Lanes right of the band need no masking here: they come last in ring order.
    const int lC = ror1(nC), lD = ror1(nD);
    x = dot2_keep(R_rIx(R), uA, lC);                   // lC, lD, lI stay live: they are row r+1's diagonal
    y = EDGE ? dot2_keep(R.d.x, uA, lD) : dot2_keep_s(J.rIy, uA, lD);
    const int base = max(x, y);
    int g, Pl;
    if (EDGE) {                                        // row M pays no gap-open (mz_yama.c:123): Q instead of P
        const bool last = R_last(R) != 0;
        const int lift = (!ROT && ((c32 ^ R_lo32(R)) >> 11) == 0) ? ROW_LIFT : 0;   // lower ring period
        Pl = last ? S.Q - lift : S.Pl;
        g = (base & ~3) - (S.xIPl - S.Pl) + (last ? J.KL4go : 0) - Pl;
    } else {
        Pl = S.Pl;
        g = (base & ~3) - S.xIPl;
    }
    if (ROT) {
        g = __builtin_amdgcn_ds_bpermute((4 * lane + s4) & (4 * WAVE - 4), g);      // lane i <- lane (i + s) & 63
        g = prefix_max64(g);
        g = __builtin_amdgcn_ds_bpermute((4 * lane - s4) & (4 * WAVE - 4), g);      // and back
    } else {
        g = prefix_max64(g);
        g = max(g, __builtin_amdgcn_readlane(g, WAVE - 1) - ROW_LIFT);   // the wrapped tail continues the lower period
    }
    nI = active ? g + Pl : NEGT + TI;
    const int lI = ror1(nI);
    z = EDGE ? dot2_keep(R.d.y, uA, lI) : dot2_keep_s(J.rIz, uA, lI);
    mI = max(base, z);

    S.p.C = nC; S.p.D = nD; S.p.I = nI;
    S.l.C = lC; S.l.D = lD; S.l.I = lI;
    S.wC = __builtin_amdgcn_alignbit(mC, S.wC, 2);
    S.wD = __builtin_amdgcn_alignbit(mD, S.wD, 2);
    S.wI = __builtin_amdgcn_alignbit(mI, S.wI, 2);
}

struct RowLoop { int next32, rcross; long long offset; };   // 32 * first column of the next 64-column period; first staged
                                                            // row at or beyond it; what re-basing has taken off the scores (x4)

// bookkeeping before a row (record R), one scalar compare per row.  When the band's left edge enters the next
// 64-column period k: the ring's copy of period k (still "upper", unlifted) is lifted in place, period k+1 is
// staged unlifted (the ring then holds k and k+1: everything a re-arming lane can ask for, with the right
// lift already applied), and the lanes that already hold a column of period k are lifted too.
// first row of the staged block (rows blk*64+1 ..) whose band starts at or beyond column next32/32
__device__ __forceinline__ int row_find_cross(const int4 *s_rec, int blk, int lane, int next32)
{
    const unsigned long long m = __builtin_amdgcn_ballot_w64(s_rec[lane * (RREC / 4)].x >= next32 &&
                                                             s_rec[lane * (RREC / 4)].x != MZ_BIG);
    return m ? blk * WAVE + 1 + (int)__builtin_ctzll(m) : MZ_BIG;
}

template <bool ROT>
__device__ __forceinline__ void row_pre(RowState &S, RowLoop &Q, int r, int lane, ColSrc &cols, const int4 *s_rec, int4 *s_ring)
{
    if (r == Q.rcross) {                               // scalar compare: the row was located when the block was staged
        // The running sums P, Q only ever enter as differences between columns of the same row, so they are
        // re-based here: the sums at the first column of the new period k become the new zero (ring copy of
        // period k, the lanes already holding its columns, the staging carry).  They stay within three periods'
        // worth of steps whatever N is.
        const int4 y0 = s_ring[2 * ((Q.next32 >> 5) & (FRING - 1)) + 1];     // {xI+P, P, c45, Q} of column 64k
        const int dP = y0.y, dQ = y0.w, sub = dP + (ROT ? 0 : ROW_LIFT);
        int4 *e = s_ring + 2 * (((Q.next32 >> 5) + lane) & (FRING - 1)) + 1;
        int4 v = *e;
        v.x -= sub; v.y -= sub; v.w -= dQ;
        __syncthreads();                               // every lane has read column 64k's entry before it changes
        *e = v;
        cols.carryP -= __builtin_amdgcn_readfirstlane(dP);
        cols.carryQ -= __builtin_amdgcn_readfirstlane(dQ);
        row_stage_cols((Q.next32 >> 5) + WAVE, 0, lane, cols, s_ring);
        __syncthreads();
        if (((S.c32 ^ Q.next32) >> 11) == 0) {         // lanes that already hold a column of period k
            S.Pl -= sub; S.xIPl -= sub; S.Q -= dQ;
        }
        Q.next32 += 32 * WAVE;
        Q.rcross = row_find_cross(s_rec, (r - 1) >> 6, lane, Q.next32);
    }
}

// Scores only ever enter as differences too (every pick compares candidates that carry the same history), so
// the frontier is re-based every 32 rows: the best state of the wave becomes the new zero and the
// amount goes into a 64-bit running offset.  Tagged int32 states then hold any length of alignment; what bounds
// them is the score range of one window of rows (k_plan).  Sentinels are left alone.
#define ROW_REBASE_FLOOR (-(1 << 30))
__device__ __forceinline__ int rebase1(int v, int d) { return v > ROW_REBASE_FLOOR ? v - d : v; }
__device__ __forceinline__ void row_rebase(RowState &S, RowLoop &Q)
{
    int m = max(max(S.p.C, S.p.D), S.p.I);
    m = prefix_max64(m);
    const int d = __builtin_amdgcn_readlane(m, WAVE - 1) & ~3;
    if (d <= ROW_REBASE_FLOOR) return;                  // (no reachable state in the frontier: cannot happen in a connected band)
    S.p.C = rebase1(S.p.C, d); S.p.D = rebase1(S.p.D, d); S.p.I = rebase1(S.p.I, d);
    S.l.C = rebase1(S.l.C, d); S.l.D = rebase1(S.l.D, d); S.l.I = rebase1(S.l.I, d);
    Q.offset += d;
}

__device__ __forceinline__ void row_store(const RowState &S, uint32_t *tbw, int r, int lane)
{
    uint32_t *o = tbw + (r >> 4) * (3 * WAVE) + lane;
    o[0] = S.wC; o[WAVE] = S.wD; o[2 * WAVE] = S.wI;
}

// rows r0..r1 of one phase, block of 64 staged rows by block (in halves: scores are re-based every 32 rows);
// within a half two rows per iteration so that
// the record registers alternate (the next row's record is read from LDS while the current row is computed).
// Pairs start on even rows, so only the second row of a pair can close a 16-row traceback group.
template <bool EDGE, bool COL, bool ROT>
__device__ __forceinline__ void row_rows(RowState &S, RowLoop &Q, int r0, int r1, int lane, const RowCtx &J,
                                         const RowSrc &src, ColSrc &cols, int4 *s_rec, int4 *s_ring, uint32_t *tbw)
{
    r0 = __builtin_amdgcn_readfirstlane(r0);
    r1 = __builtin_amdgcn_readfirstlane(r1);
    for (int r = r0; r <= r1; ) {
        if (((r - 1) & 31) == 0 && r > 1) row_rebase(S, Q);          // every 32 rows (the score window of k_plan)
        if (((r - 1) & (WAVE - 1)) == 0 && r > 1) {    // first row of a block: every record of the block before is consumed
            __syncthreads();
            row_stage_rows((r - 1) >> 6, lane, src, s_rec);
            __syncthreads();
            Q.rcross = row_find_cross(s_rec, (r - 1) >> 6, lane, Q.next32);
        }
        const int last = min(r1, ((r - 1) | 31) + 1);                // last row of this half block within [r0, r1]
        RowRec Ra, Rb;
        row_rec_read<EDGE>(Ra, s_rec, r);
        if (r & 1) {                                   // odd first row on its own
            row_pre<ROT>(S, Q, r, lane, cols, s_rec, s_ring);
            row_step<EDGE, COL, ROT>(S, Ra, ROT ? row_rec_s4(s_rec, r) : 0, r, J, s_ring, tbw, lane);
            if ((r & 15) == 15) row_store(S, tbw, r, lane);
            ++r;
            if (r > last) continue;
            row_rec_read<EDGE>(Ra, s_rec, r);
        }
        for (; r + 1 <= last; r += 2) {
            row_rec_read<EDGE>(Rb, s_rec, r + 1);
            row_pre<ROT>(S, Q, r, lane, cols, s_rec, s_ring);
            row_step<EDGE, COL, ROT>(S, Ra, ROT ? row_rec_s4(s_rec, r) : 0, r, J, s_ring, tbw, lane);
            if (r + 2 <= last) row_rec_read<EDGE>(Ra, s_rec, r + 2);
            row_pre<ROT>(S, Q, r + 1, lane, cols, s_rec, s_ring);
            row_step<EDGE, COL, ROT>(S, Rb, ROT ? row_rec_s4(s_rec, r + 1) : 0, r + 1, J, s_ring, tbw, lane);
            if (((r + 1) & 15) == 15) row_store(S, tbw, r + 1, lane);
        }
        if (r <= last) {
            row_pre<ROT>(S, Q, r, lane, cols, s_rec, s_ring);
            row_step<EDGE, COL, ROT>(S, Ra, ROT ? row_rec_s4(s_rec, r) : 0, r, J, s_ring, tbw, lane);
            if ((r & 15) == 15) row_store(S, tbw, r, lane);
            ++r;
        }
    }
}

// a uniform 64-bit value in scalar registers
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v)
{
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

template <bool COL, bool ROT>
__device__ __forceinline__ void dp_row_body(const mz_dev_batch &b, int p, int lane, int4 *s_rec, int4 *s_ring)
{
    constexpr int TD = COL ? 1 : 0, TI = COL ? 0 : 1;
    // everything per-pair is uniform: keep it in scalar registers (values the compiler fetched with vector
    // loads would otherwise sit in VGPRs and cost occupancy)
#define UNI(x) __builtin_amdgcn_readfirstlane(x)
    const int M = UNI(COL ? b.N[p] : b.M[p]), N = UNI(COL ? b.M[p] : b.N[p]);      // rows / columns of this run
    const int go = c_sc.go, ge = c_sc.ge, g1 = c_sc.g1, g2 = c_sc.g2;
    RowCtx J;
    J.K = UNI(COL ? b.L[p] : b.K[p]); J.L = UNI(COL ? b.K[p] : b.L[p]); J.N32 = 32 * N; J.KL4go = 4 * go * J.K * J.L;
    J.rIy = pack2(-J.K * 2 * g1, 0); J.rIz = pack2(-J.K * 2 * g1, -J.K * 2 * g1);
    // (offsets made uniform, not pointers: a pointer rebuilt from integers loses its address space and every
    // access through it becomes a flat_ instruction)
    const long long oBand = (long long)uniform64((unsigned long long)b.offBand[p]);
    const int *LB = b.poolLB + oBand, *RB = b.poolRB + oBand;
    const uint8_t *pA = b.poolA + (long long)uniform64((unsigned long long)b.offA[p]);
    const uint8_t *pB = b.poolB + (long long)uniform64((unsigned long long)b.offB[p]);
    RowSrc src;
    src.A = COL ? pB : pA;
    src.lo = COL ? (const int *)(b.prep + (long long)uniform64((unsigned long long)b.offPrep[p])) : LB;
    src.hi = COL ? src.lo + (M + 1) : RB;              // (COL: M is the reference's N)
    src.K = J.K; src.L = J.L; src.M = M; src.go = go; src.ge = ge; src.g1 = 2 * g1;
    ColSrc cols;
    cols.B = COL ? pA : pB;
    cols.L = J.L; cols.N = N; cols.K4go = 4 * J.K * go; cols.K4ge = 4 * J.K * ge; cols.g2 = 2 * g2;
    cols.xI0 = J.KL4go - TI; cols.carryP = cols.carryQ = 0;
    uint32_t *tbw = b.tbw + (long long)uniform64((unsigned long long)b.offTb[p]);
    const int rL = UNI(b.edgeLo[p]), rN = UNI(b.edgeHi[p]);   // rows <= rL hold column 0/1, rows >= rN column N
#undef UNI

    RowLoop Q;
    Q.next32 = 32 * WAVE; Q.offset = 0;
    row_stage_cols(0, ROT ? 0 : ROW_LIFT, lane, cols, s_ring);
    row_stage_cols(WAVE, 0, lane, cols, s_ring);
    row_stage_rows(0, lane, src, s_rec);
    __syncthreads();
    Q.rcross = row_find_cross(s_rec, 0, lane, Q.next32);

    // row 0 in closed form (mz_yama.c:83-94): C = D = NEG beyond (0,0); I(0,c) = -ge*K*(nB_1+..+nB_c)
    RowState S;
    S.c32 = 32 * lane;
    row_load_col(S, s_ring);
    const int rb0 = COL ? t_hi(LB, N, 0) : RB[0];
    S.p.C = lane == 0 ? 2 : NEGT + 2;
    S.p.D = lane == 0 ? TD : NEGT + TD;
    S.p.I = lane <= rb0 ? TI + S.Q : NEGT + TI;
    S.l.C = ror1(S.p.C); S.l.D = ror1(S.p.D); S.l.I = ror1(S.p.I);
    S.wC = S.wD = S.wI = 0;

    // rows 1..e1 edge, e1+1..e2-1 interior, e2..M edge (transposed: row 1 is always an edge row)
    const int e1 = min(max(rL, COL ? 1 : 0), M), e2 = max(rN, e1 + 1);
    row_rows<true, COL, ROT>(S, Q, 1, e1, lane, J, src, cols, s_rec, s_ring, tbw);
    row_rows<false, COL, ROT>(S, Q, e1 + 1, e2 - 1, lane, J, src, cols, s_rec, s_ring, tbw);
    row_rows<true, COL, ROT>(S, Q, e2, M, lane, J, src, cols, s_rec, s_ring, tbw);

    if ((M & 15) != 15) {                              // flush the partial group
        const int sh = 2 * (15 - (M & 15));
        uint32_t *o = tbw + (M >> 4) * (3 * WAVE) + lane;
        o[0] = S.wC >> sh; o[WAVE] = S.wD >> sh; o[2 * WAVE] = S.wI >> sh;
    }
    if (lane == (N & (WAVE - 1))) {                    // the reference's (C,D,I) at (M,N), unscaled
        b.final3[3 * p + 0] = (int)((S.p.C + Q.offset) >> 2);
        b.final3[3 * p + 1] = (int)(((COL ? S.p.I : S.p.D) + Q.offset) >> 2);
        b.final3[3 * p + 2] = (int)(((COL ? S.p.D : S.p.I) + Q.offset) >> 2);
    }
}

__global__ __launch_bounds__(WAVE, 5) void k_dp_row(mz_dev_batch b, int first, int count)
{
    __shared__ int4 s_ring[2 * FRING];                 // 4 KB: column records, 128-entry ring
    __shared__ int4 s_rec[WAVE * (RREC / 4)];          // 4 KB: row records of the current block of 64 rows
    const int p = first + blockIdx.x, lane = threadIdx.x;
    if (blockIdx.x == 0 && lane == 0) b.totals[6] = 0;       // the pair counter of k_dp, launched next on this stream
    if (b.status[p] != MZ_OK) return;
    const int mode = b.mode[p];
    if (mode == MZ_MODE_ROW)       dp_row_body<false, false>(b, p, lane, s_rec, s_ring);
    else if (mode == MZ_MODE_COL)  dp_row_body<true, false>(b, p, lane, s_rec, s_ring);
    else if (mode == MZ_MODE_ROWR) dp_row_body<false, true>(b, p, lane, s_rec, s_ring);
    else if (mode == MZ_MODE_COLR) dp_row_body<true, true>(b, p, lane, s_rec, s_ring);
}

// ------------------------------------------------------------------------------------------
// strip-mined DP kernel (MZ_MODE_STRIP): any legal band.
//
// Rows are processed 64 at a time (lane <-> row of the strip); a strip sweeps the columns
// LB[first row] .. RB[last row] with the usual one-step skew between lanes.  The row above the
// strip is read from a boundary row in global memory (64 columns per coalesced load, handed to
// lane 0 one column per step), the strip's last row is collected the same way and written back
// for the next strip.  Arithmetic is the exact cell() of k_dp_wf64.  Slower than the rolling
// wavefront for narrow bands (fill/drain per strip), efficient for wide ones.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void dp_strip_body(const mz_dev_batch &b, int p, int lane, int *s_rec, int4 *s_ring)
{

    PairCtx J;
    J.K = b.K[p]; J.L = b.L[p]; J.M = b.M[p]; J.N = b.N[p];
    J.A = b.poolA + b.offA[p]; J.B = b.poolB + b.offB[p];
    J.LB = b.poolLB + b.offBand[p]; J.RB = b.poolRB + b.offBand[p];
    const int M = J.M, N = J.N;
    const int go = c_sc.go, ge = c_sc.ge;
    const int pkKy = pack4(0, J.K, 0, 0), pkKz = pack4(0, J.K, 0, -J.K);
    const int S = (M + WAVE - 1) / WAVE;
    uint32_t *tbw = b.tbw + b.offTb[p];
    const long long hdrDw = ((2LL * S + WAVE - 1) / WAVE) * WAVE;
    const long long bndDw = ((6LL * (N + 1) + WAVE - 1) / WAVE) * WAVE;
    int *bnd = (int *)(tbw + hdrDw);                 // two rows x {C[N+1], D[N+1], I[N+1]}
    long long dataOff = hdrDw + bndDw;
    const int NP = N + 1;

    // ---- row 0 (mz_yama.c:83-94) into boundary row 0: C = D = NEG, I = running sum of -nB(c)*K*ge
    {
        int *r0 = bnd;
        int carry = 0;
        const int hi0 = J.RB[0];
        for (int c0 = 0; c0 <= hi0; c0 += WAVE) {
            const int c = c0 + lane;
            int v = 0;
            if (c >= 1 && c <= hi0) {
                const uint8_t *col = J.B + (long long)(c - 1) * J.L;
                int nb = 0;
                for (int j = 0; j < J.L; ++j) nb += col[j] != '-';
                v = -nb * J.K * ge;
            }
#pragma unroll
            for (int o = 1; o < WAVE; o <<= 1) { const int u = __shfl_up(v, o); if (lane >= o) v += u; }
            v += carry;
            if (c <= hi0) {
                r0[c] = c == 0 ? 0 : MZ_NEG; r0[NP + c] = c == 0 ? 0 : MZ_NEG; r0[2 * NP + c] = v;
            }
            carry = __shfl(v, WAVE - 1);
        }
    }
    __threadfence();

    Tri st = { MZ_NEG, MZ_NEG, MZ_NEG };
    for (int s = 0; s < S; ++s) {
        const int first = s * WAVE + 1, last = min(first + WAVE - 1, M), nr = last - first + 1;
        const int clo = J.LB[first], chi = J.RB[last];
        const int pLB = J.LB[first - 1], pRB = J.RB[first - 1];
        const int *prev = bnd + (s & 1) * 3 * NP;
        int *next = bnd + ((s + 1) & 1) * 3 * NP;
        const int nsteps = chi - clo + nr;            // tau = 0 .. nsteps-1
        if (lane == 0) { tbw[2 * s] = (uint32_t)clo; tbw[2 * s + 1] = (uint32_t)dataOff; }
        uint32_t *tbs = tbw + dataOff;
        dataOff += (long long)((nsteps + 3) >> 2) * WAVE;

        stage_rows(s, lane, J, s_rec);
        int staged = (clo >> 6) << 6;                 // B columns are staged in aligned blocks of 64
        stage_bcols(staged, lane, J, s_ring);         // (column 0 and columns > N get the zero entry)
        staged += WAVE;
        __syncthreads();
        RowRegs R;
        load_rec(R, s_rec + (((s & 1) * WAVE) + lane) * REC_DW);
        const int r = first + lane;

        st.C = st.D = st.I = MZ_NEG;
        Tri up = { MZ_NEG, MZ_NEG, MZ_NEG }, dg;
        // lane 0's diagonal predecessor of its first cell: P(first-1, clo-1)
        if (lane == 0 && clo - 1 >= pLB && clo - 1 <= pRB && clo >= 1) {
            up.C = prev[clo - 1]; up.D = prev[NP + clo - 1]; up.I = prev[2 * NP + clo - 1];
        }
        // hand that value over as "state of the lane before lane 0": emulate by keeping it in `up`
        // and skipping the first rotate for lane 0 (see below).
        int pc = MZ_NEG, pd = MZ_NEG, pi = MZ_NEG;    // 64 columns of the row above the strip
        int cc = MZ_NEG, cd = MZ_NEG, ci = MZ_NEG;    // 64 collected columns of the strip's last row
        unsigned tbword = 0;

        for (int tau = 0; tau < nsteps; ++tau) {
            if ((tau & (WAVE - 1)) == 0) {            // next 64 columns of the boundary row
                const int c = clo + tau + lane;
                const bool ok = c >= pLB && c <= pRB && c <= N;
                pc = ok ? prev[c] : MZ_NEG; pd = ok ? prev[NP + c] : MZ_NEG; pi = ok ? prev[2 * NP + c] : MZ_NEG;
            }
            if (clo + tau >= staged) {                // lane 0 is about to need column `staged`
                stage_bcols(staged, lane, J, s_ring);
                staged += WAVE;
                __syncthreads();
            }
            dg = up;
            up.C = ror1(st.C); up.D = ror1(st.D); up.I = ror1(st.I);
            {
                const int k = tau & (WAVE - 1);
                const int uc = __builtin_amdgcn_readlane(pc, k), ud = __builtin_amdgcn_readlane(pd, k),
                          ui = __builtin_amdgcn_readlane(pi, k);
                if (lane == 0) { up.C = uc; up.D = ud; up.I = ui; }
            }
            const int c = clo + tau - lane;
            const int4 q = s_ring[c & (BRING - 1)];
            int tbyte;
            const Tri nw = cell(R, c, N, q, st, up, dg, pkKy, pkKz, go, ge, tbyte);
            const bool active = (r <= M) & (c >= R.lo) & (c <= R.hi);
            st.C = active ? nw.C : MZ_NEG;
            st.D = active ? nw.D : MZ_NEG;
            st.I = active ? nw.I : MZ_NEG;

            tbword = __builtin_amdgcn_alignbyte(tbyte, tbword, 1);
            if ((tau & 3) == 3) tbs[(tau >> 2) * WAVE + lane] = tbword;

            // collect the last row of the strip: at step tau it is at column clo + tau - (nr-1)
            const int kcol = tau - (nr - 1);
            if (kcol >= 0) {
                const int lc = __builtin_amdgcn_readlane(st.C, nr - 1), ld = __builtin_amdgcn_readlane(st.D, nr - 1),
                          li = __builtin_amdgcn_readlane(st.I, nr - 1);
                if (lane == (kcol & (WAVE - 1))) { cc = lc; cd = ld; ci = li; }
                if ((kcol & (WAVE - 1)) == WAVE - 1 || tau == nsteps - 1) {
                    const int cbase = clo + (kcol & ~(WAVE - 1));
                    const int c2 = cbase + lane;
                    if (lane <= (kcol & (WAVE - 1)) && c2 <= N) { next[c2] = cc; next[NP + c2] = cd; next[2 * NP + c2] = ci; }
                }
            }
        }
        if ((nsteps & 3) != 0)
            tbs[((nsteps - 1) >> 2) * WAVE + lane] = tbword >> (8 * (4 - (nsteps & 3)));
        __threadfence();                              // boundary row visible to the next strip's loads
        __syncthreads();
    }
    if (lane == ((M - 1) & (WAVE - 1))) {
        b.final3[3 * p + 0] = st.C;
        b.final3[3 * p + 1] = st.D;
        b.final3[3 * p + 2] = st.I;
    }
}

// ------------------------------------------------------------------------------------------
// the DP kernel: one wave per pair, dispatching on the mode the plan chose.  [first, first+count) is the
// slice of the batch this launch covers (the host runs a batch in a few slices so that the traceback walk
// of one slice overlaps the DP of the next).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WAVE) void k_dp(mz_dev_batch b, int first, int count)
{
    __shared__ __attribute__((aligned(16))) int4 smem[(2 * WAVE * TREC + 6 * FRING) / 4];     // 13 KB, carved per mode
#ifdef MZ_LDS_PAD
    __shared__ int s_pad[MZ_LDS_PAD];
    if (b.n < 0) s_pad[threadIdx.x] = 1;
#endif
    // a fixed grid of waves takes pairs from a shared counter (totals[6], zeroed by k_dp_row, which always runs
    // just before on the same stream): when the
    // row-parallel kernels took every pair (the usual case) this launch costs a few microseconds instead of one
    // dispatched-and-exited wave per pair, and when they did not the waves still balance dynamically
    const int lane = threadIdx.x;
    int *s_rec = (int *)smem;
    unsigned long long *next = (unsigned long long *)&b.totals[6];
    if (b.totals[5] == 0) return;                       // nothing for these kernels in this batch (no atomics either)
    for (;;) {
        int p = 0;
        if (lane == 0) p = first + (int)atomicAdd(next, 1ULL);
        p = __builtin_amdgcn_readfirstlane(p);
        if (p >= first + count) break;
        if (b.status[p] != MZ_OK) continue;
        const int mode = b.mode[p];
        if (mode == MZ_MODE_FASTT)      dp_tag_body(b, p, lane, s_rec, s_rec + 2 * WAVE * TREC);
        else if (mode == MZ_MODE_FAST)  dp_fast_body<false>(b, p, lane, s_rec, s_rec + 2 * WAVE * TREC);
        else if (mode == MZ_MODE_WF64)  dp_wf64_body(b, p, lane, s_rec, (int4 *)(s_rec + 2 * WAVE * REC_DW));
        else if (mode == MZ_MODE_STRIP) dp_strip_body(b, p, lane, s_rec, (int4 *)(s_rec + 2 * WAVE * REC_DW));
        else continue;
        __syncthreads();                               // the next pair restages the same LDS
    }
}

// ------------------------------------------------------------------------------------------
// traceback walk (mz_yama.c:257-291): one lane per pair, serial pointer chase over the
// traceback bytes; writes the edit script in reverse order.
// ------------------------------------------------------------------------------------------
#define WALK_LANES 64            // pairs per wave (measured: 16 per wave is no faster -- the chase is bound by its own dependent loads -- and costs four times the instruction issue beside the DP)
// (An LDS-tile variant of this chase -- each pair caching the entries around its position -- was superseded by the
// run-following walk below and removed.)
__device__ __forceinline__ void walk_pair(const mz_dev_batch &b, int p)
{
    const int M = b.M[p], N = b.N[p];
    const uint32_t *tbw = b.tbw + b.offTb[p];
    uint8_t *ops = b.script + b.offScript[p];
    const int mode = b.mode[p];
    const bool tagged = mode == MZ_MODE_FASTT || mode >= MZ_MODE_ROW;
    const bool rowfam = mode >= MZ_MODE_ROW, colfam = mode == MZ_MODE_COL || mode == MZ_MODE_COLR;
    const int fC = b.final3[3 * p], fD = b.final3[3 * p + 1], fI = b.final3[3 * p + 2];

    // final-cell tie order C, D, I (mz_yama.c:262-267): D wins a D/I tie here
    int node = (fC >= fD && fC >= fI) ? MZ_FC : (fD >= fI) ? MZ_FD : MZ_FI;
    int r = M, c = N, n = 0, status = MZ_OK;
    const int limit = M + N;
    while (r > 0 || c > 0) {
        if (r < 0 || c < 0 || n >= limit) { status = MZ_E_TRACEBACK; break; }
        unsigned stb;
        const int r_was = r;
        if (r == 0) {
            stb = MZ_FI << 4;                              // row 0 bytes, mz_yama.c:92
        } else if (rowfam) {
            // row-parallel kernels: entry of (r,c) = bits 2*(u&15) of word ((u>>4)*3 + s)*64 + (w & 63), with
            // (u,w) = (r,c) and streams C,D,I for ROW; (u,w) = (c,r) and streams C,I,D for the transposed COL
            // (its D slot holds the picks of the reference's I state).  COL does not store column 0: only D
            // is reachable there (mz_yama.c:211), and it comes from D.
            const int u = colfam ? c : r, w = colfam ? r : c;
            const int g = u >> 4, l = w & (WAVE - 1);
            const int sidx = node == MZ_FC ? 0 : node == (colfam ? MZ_FI : MZ_FD) ? 1 : 2;
            unsigned tg = 0;
            if (!(colfam && c == 0)) tg = (tbw[(g * 3 + sidx) * WAVE + l] >> (2 * (u & 15))) & 3;
            stb = tg | (tg << 2) | (tg << 4);
        } else if (tagged) {
            // 2-bit tag streams: word ((t>>4)*3 + s)*64 + lane, s = 0/1/2 for the C/D/I pick; read only
            // the stream of the node we stand on and put its tag where the decode below expects it
            const int t = r + c, l = (r - 1) & (WAVE - 1);
            const int sidx = node == MZ_FC ? 0 : node == MZ_FD ? 1 : 2;
            const unsigned tg = (tbw[((t >> 4) * 3 + sidx) * WAVE + l] >> (2 * (t & 15))) & 3;
            stb = tg | (tg << 2) | (tg << 4);
        } else if (mode != MZ_MODE_STRIP) {
            const int t = r + c, g = t >> 2, l = (r - 1) & (WAVE - 1);
            stb = (tbw[g * WAVE + l] >> (8 * (t & 3))) & 0xff;
        } else {
            const int s = (r - 1) >> 6, l = (r - 1) & (WAVE - 1);
            const int clo = (int)tbw[2 * s];
            const long long base = (long long)tbw[2 * s + 1];
            const int tau = c - clo + l;
            stb = (tbw[base + (long long)(tau >> 2) * WAVE + l] >> (8 * (tau & 3))) & 0xff;
        }
        ops[n++] = (uint8_t)node;
        int nx;
        if (node == MZ_FI)      { c -= 1;         nx = (stb >> 4) & 3; }
        else if (node == MZ_FD) { r -= 1;         nx = (stb >> 2) & 3; }
        else if (node == MZ_FC) { r -= 1; c -= 1; nx = stb & 3; }
        else { status = MZ_E_TRACEBACK; break; }
        node = (tagged && r_was > 0) ? 2 - nx : nx;     // tagged kernels store tie-break tags: C=2, I=1, D=0
    }
    if (status == MZ_OK && (r != 0 || c != 0)) status = MZ_E_TRACEBACK;
    b.om[p] = n;
    if (status != MZ_OK) b.status[p] = status;
}
__global__ __launch_bounds__(WAVE) void k_walk(mz_dev_batch b, int first, int count)
{
    if (threadIdx.x >= WALK_LANES) return;
    const int p = first + blockIdx.x * WALK_LANES + threadIdx.x;
    if (p >= first + count || b.status[p] != MZ_OK) return;
    walk_pair(b, p);
}

// One WAVE per pair, following RUNS instead of steps.  The 64 lanes fetch a window of WIN_G groups of the
// traceback -- all three streams, all 64 ring lanes, 12 KB contiguous -- into LDS with coalesced 16-byte loads.
// Then, standing on node s at (r,c), lane k looks at the entry of the same state k moves further along (k cells up
// the diagonal for C, up the column for D, along the row for I): as long as an entry says "came from the same
// state" the path goes straight on, so one ballot finds the end of the run and the wave advances the whole run --
// up to 64 steps, and 64 bytes of edit script in one coalesced store -- for the price of one step.  Alignments
// are mostly long diagonal runs; a path that turns at every step costs what a step-by-step chase costs.
// Layouts (2-bit tie-break tags C=2, I=1, D=0, so node = 2 - tag with FC=0, FI=1, FD=2; word = (group*3 + stream)
// * 64 + lane, bits 2*(index & 15)):
//     ROW / ROWR   group r>>4,     lane c & 63,     index r,   streams C,D,I
//     COL / COLR   group c>>4,     lane r & 63,     index c,   streams C,I,D   (column 0 not stored)
//     FASTT        group (r+c)>>4, lane (r-1) & 63, index r+c, streams C,D,I
// Row 0 is stored by none of them (all I, mz_yama.c:92).  Pairs of the untagged kernels are chased by lane 0.
#define WIN_G 16
template <int LAYOUT>       // 0 ROW, 1 COL, 2 FASTT
__device__ __forceinline__ void walk_runs(const mz_dev_batch &b, int p, int lane, int *s_win)
{
    const int M = b.M[p], N = b.N[p];
    const uint32_t *tbw = b.tbw + b.offTb[p];
    uint8_t *ops = b.script + b.offScript[p];
    const int fC = b.final3[3 * p], fD = b.final3[3 * p + 1], fI = b.final3[3 * p + 2];
    int node = (fC >= fD && fC >= fI) ? MZ_FC : (fD >= fI) ? MZ_FD : MZ_FI;     // mz_yama.c:262-267
    const int cmin = LAYOUT == 1 ? 1 : 0;
    int r = M, c = N, n = 0, status = MZ_OK;
    int gb = 1 << 30;                                   // first group of the window in LDS (none yet)
    while (r > 0 && c >= cmin) {
        const int g = (LAYOUT == 0 ? r : LAYOUT == 1 ? c : r + c) >> 4;
        if ((unsigned)(g - gb) >= (unsigned)WIN_G) {    // wave-uniform: every lane follows the same chain
            gb = max(g - (WIN_G - 1), 0);
            const int4 *src = (const int4 *)(tbw + (long long)gb * (3 * WAVE));
            const int n4 = (g - gb + 1) * (3 * WAVE / 4);
            __syncthreads();
            if (n4 == WIN_G * 3 * WAVE / 4) {           // a full window: twelve loads in flight, then twelve LDS writes
                int4 v[WIN_G * 3 / 4];
#pragma unroll
                for (int j = 0; j < WIN_G * 3 / 4; ++j) v[j] = src[lane + j * WAVE];
#pragma unroll
                for (int j = 0; j < WIN_G * 3 / 4; ++j) ((int4 *)s_win)[lane + j * WAVE] = v[j];
            } else {                                    // the top of the pair (fewer than WIN_G groups left)
                for (int k = lane; k < n4; k += WAVE) ((int4 *)s_win)[k] = src[k];
            }
            __syncthreads();
        }
        const int dr = node != MZ_FI, dc = node != MZ_FD;
        const int rk = r - lane * dr, ck = c - lane * dc;
        const int ix = LAYOUT == 0 ? rk : LAYOUT == 1 ? ck : rk + ck;
        const int ln = LAYOUT == 0 ? ck : LAYOUT == 1 ? rk : rk - 1;
        const int sidx = node == MZ_FC ? 0 : (node == MZ_FD) == (LAYOUT != 1) ? 1 : 2;
        const bool valid = rk >= 1 && ck >= cmin && (ix >> 4) >= gb;
        const unsigned word = valid ? (unsigned)s_win[(((ix >> 4) - gb) * 3 + sidx) * WAVE + (ln & (WAVE - 1))] : 0u;
        const int tag = (word >> (2 * (ix & 15))) & 3;
        const unsigned long long vmask = __builtin_amdgcn_ballot_w64(valid);
        const unsigned long long stop = ~__builtin_amdgcn_ballot_w64(valid && tag == 2 - node);
        const int L = stop ? __builtin_ctzll(stop) : WAVE;
        const bool turn = L < WAVE && ((vmask >> L) & 1);       // lane L stands on a stored entry: the path turns there
        const int E = L + (turn ? 1 : 0);                        // steps taken in this state
        if (lane < E) ops[n + lane] = (uint8_t)node;
        n += E; r -= E * dr; c -= E * dc;
        if (turn) {
            node = 2 - __builtin_amdgcn_readlane(tag, L);
            if (node < 0) { status = MZ_E_TRACEBACK; break; }
        }
    }
    if (status == MZ_OK) {
        // the edges: row 0 is all I (mz_yama.c:92), column 0 all D (mz_yama.c:211) -- and the path must arrive there
        // in that state, or the reference's walk would step outside the grid (mz_yama.c:274-276)
        if (r < 0 || c < 0) status = MZ_E_TRACEBACK;
        else if (r == 0 && c > 0) {
            if (node != MZ_FI) status = MZ_E_TRACEBACK;
            else { for (int k = lane; k < c; k += WAVE) ops[n + k] = MZ_FI; n += c; }
        } else if (c == 0 && r > 0) {                   // (COL layouts only: the others store column 0)
            if (node != MZ_FD) status = MZ_E_TRACEBACK;
            else { for (int k = lane; k < r; k += WAVE) ops[n + k] = MZ_FD; n += r; }
        }
    }
    if (lane == 0) {
        b.om[p] = n;
        if (status != MZ_OK) b.status[p] = status;
    }
}

__global__ __launch_bounds__(WAVE) void k_walk_wave(mz_dev_batch b, int first, int count)
{
    __shared__ __attribute__((aligned(16))) int s_win[WIN_G * 3 * WAVE];
    const int p = first + blockIdx.x, lane = threadIdx.x;
    if (p >= first + count || b.status[p] != MZ_OK) return;
    const int mode = b.mode[p];
    if (mode == MZ_MODE_ROW || mode == MZ_MODE_ROWR)      walk_runs<0>(b, p, lane, s_win);
    else if (mode == MZ_MODE_COL || mode == MZ_MODE_COLR) walk_runs<1>(b, p, lane, s_win);
    else if (mode == MZ_MODE_FASTT)                       walk_runs<2>(b, p, lane, s_win);
    else if (lane == 0)                                   walk_pair(b, p);
}

// ------------------------------------------------------------------------------------------
// emit (mz_yama.c:293-313 + new_col :39-47): one wave per pair, 64 output columns per
// iteration; column m takes A[i] or dashes on top of B[j] or dashes, where (i, j) are the
// running counts of A- and B-advancing ops up to m.
// ------------------------------------------------------------------------------------------
// WIDE: blocks of more than 8 rows in all go through LDS (below); narrow ones write their few bytes directly --
// the 16 KB of LDS would only cost them occupancy.
template <bool WIDE>
__device__ __forceinline__ void emit_body(const mz_dev_batch &b, int p, uint8_t *s_cols)
{
    const int lane = threadIdx.x;
    if (b.status[p] != MZ_OK) return;
    const int K = b.K[p], L = b.L[p], M = b.M[p], N = b.N[p], n = b.om[p];
    if ((K + L > 8) != WIDE) return;
    const uint8_t *A = b.poolA + b.offA[p], *B = b.poolB + b.offB[p];
    const uint8_t *ops = b.script + b.offScript[p];
    uint8_t *out = b.out + b.offOut[p];
    const unsigned long long below = (lane == 63) ? ~0ULL : ((1ULL << (lane + 1)) - 1ULL);
    const int W = K + L;
    int ia = 0, jb = 0;                                     // columns of A / B consumed so far
    for (int base = 0; base < n; base += WAVE) {
        const int m = base + lane;
        const bool live = m < n;
        const int op = live ? ops[n - 1 - m] : MZ_FI;
        const bool adA = live && op != MZ_FI, adB = live && op != MZ_FD;
        const unsigned long long mA = __ballot(adA), mB = __ballot(adB);
        const int i = ia + __popcll(mA & below);            // 1-based column of A (if adA)
        const int j = jb + __popcll(mB & below);
        if (!WIDE) {
            if (live) {
                uint8_t *col = out + (long long)m * W;
                const uint8_t *ca = A + (long long)(i - 1) * K, *cb = B + (long long)(j - 1) * L;
                for (int k = 0; k < K; ++k) col[k] = adA ? ca[k] : (uint8_t)'-';
                for (int k = 0; k < L; ++k) col[K + k] = adB ? cb[k] : (uint8_t)'-';
            }
        } else {
            // the 64 columns of this round are assembled in LDS (byte stores) and leave as consecutive dwords:
            // a lane writing its own K+L bytes straight to HBM costs one scattered byte store per row of the block
            if (live) {
                uint8_t *col = s_cols + lane * W;
                const uint8_t *ca = A + (long long)(i - 1) * K, *cb = B + (long long)(j - 1) * L;
                for (int k = 0; k < K; ++k) col[k] = adA ? ca[k] : (uint8_t)'-';
                for (int k = 0; k < L; ++k) col[K + k] = adB ? cb[k] : (uint8_t)'-';
            }
            __syncthreads();
            {
                const int nbytes = min(WAVE, n - base) * W;                // base * W is a multiple of 64: dword aligned
                uint32_t *g = (uint32_t *)(out + (long long)base * W);
                const uint32_t *s = (const uint32_t *)s_cols;
                for (int w = lane; w < (nbytes >> 2); w += WAVE) g[w] = s[w];
                if (lane < (nbytes & 3)) out[(long long)base * W + (nbytes & ~3) + lane] = s_cols[(nbytes & ~3) + lane];
            }
            __syncthreads();
        }
        ia += __popcll(mA);
        jb += __popcll(mB);
    }
    if (lane == 0 && (ia != M || jb != N)) b.status[p] = MZ_E_EMIT;   // mz_yama.c:310-312
}

__global__ __launch_bounds__(WAVE) void k_emit(mz_dev_batch b, int first, int count)
{
    emit_body<false>(b, first + blockIdx.x, NULL);
}
// (a fixed grid striding over the batch: a batch without wide blocks then costs a few thousand waves that exit at
// once, not one 16 KB LDS allocation per pair queued behind the DP)
__global__ __launch_bounds__(WAVE) void k_emit_wide(mz_dev_batch b, int first, int count)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_cols[WAVE * 254];   // 64 output columns of up to 127 + 127 rows
    for (int p = first + blockIdx.x; p < first + count; p += gridDim.x) emit_body<true>(b, p, s_cols);
}

// ------------------------------------------------------------------------------------------
// C-ABI launchers
// ------------------------------------------------------------------------------------------
static char g_err[256];
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
    h.tag_ok = (m->g1 > 0 && 2 * m->g1 * 127 <= 32767 && 2 * m->g2 * 127 <= 32767) ? 1 : 0;
    CK(hipMemcpyToSymbolAsync(HIP_SYMBOL(c_sc), &h, sizeof h, 0, hipMemcpyHostToDevice, (hipStream_t)stream), "upload scores");
    CK(hipStreamSynchronize((hipStream_t)stream), "upload scores sync");
    return 0;
}

// ------------------------------------------------------------------------------------------
// Band bounds as the host path ships them (mz_yama_batch): LB and RB are monotone and move by a column or two per
// row, so instead of 8 bytes per row (two thirds of a C2 pair's input) the staging block carries, per pair,
//   format 1:  LB[0], RB[0] (int32), then M bytes LB[i]-LB[i-1], then M bytes RB[i]-RB[i-1]   (all steps 0..255)
//   format 0:  LB[0..M], RB[0..M] as int32                                                    (anything else)
// at byte offset offC[p] (4-byte aligned).  One wave per pair turns that back into the int32 pools every kernel reads.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WAVE) void k_unband(int n, const int32_t *bandLen, const int64_t *offBand, const int64_t *offC,
                                                 const uint8_t *fmt, const uint8_t *packed, int32_t *poolLB, int32_t *poolRB)
{
    const int p = blockIdx.x, lane = threadIdx.x;
    if (p >= n) return;
    const int M = bandLen[p] - 1;                       // entries 0..M (an invalid job carries one dummy entry)
    int32_t *LB = poolLB + offBand[p], *RB = poolRB + offBand[p];
    const uint8_t *src = packed + offC[p];
    if (fmt[p] == 0) {
        const int32_t *s = (const int32_t *)src;
        for (int i = lane; i <= M; i += WAVE) { LB[i] = s[i]; RB[i] = s[M + 1 + i]; }
        return;
    }
    int baseL = ((const int32_t *)src)[0], baseR = ((const int32_t *)src)[1];
    const uint8_t *dL = src + 8, *dR = src + 8 + M;
    if (lane == 0) { LB[0] = baseL; RB[0] = baseR; }
    for (int i0 = 1; i0 <= M; i0 += WAVE) {
        const int i = i0 + lane;
        int x = i <= M ? dL[i - 1] : 0, y = i <= M ? dR[i - 1] : 0;
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

extern "C" int mzk_unband(int n, const int32_t *bandLen, const int64_t *offBand, const int64_t *offC, const uint8_t *fmt,
                          const uint8_t *packed, int32_t *poolLB, int32_t *poolRB, void *stream)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_unband, dim3(n), dim3(WAVE), 0, (hipStream_t)stream, n, bandLen, offBand, offC, fmt, packed, poolLB, poolRB);
    CK(hipGetLastError(), "unband launch");
    return 0;
}

extern "C" int mzk_plan(const mz_dev_batch *b, void *stream)
{
    if (b->n <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_plan, dim3(b->n), dim3(WAVE), 0, s, *b);
    {
        const int nblk = (b->n + SCAN_B - 1) / SCAN_B;
        hipLaunchKernelGGL(k_scan1, dim3(nblk), dim3(SCAN_B), 0, s, *b);
        hipLaunchKernelGGL(k_scan2, dim3(1), dim3(64), 0, s, *b, nblk);
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
    hipLaunchKernelGGL(k_rowprep, dim3(b->n), dim3(WAVE), 0, (hipStream_t)stream, *b, 0, b->n);
    CK(hipGetLastError(), "prep launch");
    return 0;
}

extern "C" int mzk_dp_range(const mz_dev_batch *b, int first, int count, void *stream)
{
    if (count <= 0) return 0;
    static int dyn_lds = -1;                      // MZ_DYN_LDS=<bytes>: occupancy experiments (extra, unused LDS per wave)
    if (dyn_lds < 0) { const char *e = getenv("MZ_DYN_LDS"); dyn_lds = e ? atoi(e) : 0; }
    hipLaunchKernelGGL(k_dp_row, dim3(count), dim3(WAVE), dyn_lds, (hipStream_t)stream, *b, first, count);
    hipLaunchKernelGGL(k_dp, dim3(count < 6144 ? count : 6144), dim3(WAVE), dyn_lds, (hipStream_t)stream, *b, first, count);
    CK(hipGetLastError(), "dp launch");
    return 0;
}
extern "C" int mzk_walk_range(const mz_dev_batch *b, int first, int count, void *stream, int beside_dp)
{
    if (count <= 0) return 0;
    // The run-following walk, a wave per pair -- except for batches above 16 Ki pairs that run BESIDE the DP of a
    // neighbouring batch (mz_dev_run_async, the chunks of mz_yama_batch): there the step-by-step chase with a lane
    // per pair is used, latency-bound but almost free in instruction issue, which is what that DP needs; on
    // alignments whose paths turn every few steps (the synthetic C2 pairs drift out of their band) 50 000
    // run-following waves cost the DP beside them 10 %.
    // MZ_WALK=wave|direct forces one (tests, measurements).
    static int force = -1;
    if (force < 0) { const char *e = getenv("MZ_WALK"); force = !e ? 0 : e[0] == 'w' ? 1 : e[0] == 'd' ? 2 : 0; }
    if (force ? force == 1 : (!beside_dp || count <= 16384))
        hipLaunchKernelGGL(k_walk_wave, dim3(count), dim3(WAVE), 0, (hipStream_t)stream, *b, first, count);
    else
        hipLaunchKernelGGL(k_walk, dim3((count + WALK_LANES - 1) / WALK_LANES), dim3(WAVE), 0, (hipStream_t)stream, *b, first, count);
    CK(hipGetLastError(), "walk launch");
    return 0;
}
extern "C" int mzk_emit_range(const mz_dev_batch *b, int first, int count, void *stream)
{
    if (count <= 0) return 0;
    hipLaunchKernelGGL(k_emit, dim3(count), dim3(WAVE), 0, (hipStream_t)stream, *b, first, count);
    hipLaunchKernelGGL(k_emit_wide, dim3(count < 4096 ? count : 4096), dim3(WAVE), 0, (hipStream_t)stream, *b, first, count);
    CK(hipGetLastError(), "emit launch");
    return 0;
}
extern "C" int mzk_dp(const mz_dev_batch *b, void *stream)   { return mzk_dp_range(b, 0, b->n, stream); }
extern "C" int mzk_walk(const mz_dev_batch *b, void *stream, int beside_dp) { return mzk_walk_range(b, 0, b->n, stream, beside_dp); }
extern "C" int mzk_emit(const mz_dev_batch *b, void *stream) { return mzk_emit_range(b, 0, b->n, stream); }
