/* oracle/oracle.h -- CPU restatement of the multiz yama()/smooth() hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under multiz_amd/ may include, link or call this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
 * checker / reported CPU baseline -- never as the thing measured or shipped.
 *
 * Parity status: PINNED.  The reference has no golden vectors of its own (SURVEY.md
 * section 4: no tests exist), so the oracle is pinned against outputs of the reference
 * itself: oracle/_ref/libref.so (built from /root/reference by oracle/Makefile) in
 * tests/test_oracle_vs_reference.py, and the committed fixtures under tests/golden/
 * that tests/golden/make_golden.py generated from that same libref.so.
 *
 * Conventions here are 0-based and flat (no 1-based pointer arrays):
 *   A : M columns of K bytes, column r (1-based, as in the reference) at A + (r-1)*K
 *   B : N columns of L bytes, column c at B + (c-1)*L
 *   LB, RB : int[M+1], index = DP row 0..M            (reference mz_yama.h:10-13)
 *   out : merged block, OM columns of K+L bytes        (reference mz_yama.c:293-313)
 */
#ifndef MZ_ORACLE_H
#define MZ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference mz_yama.c:29  (#define MININT INT_MIN/2) */
#define MZO_NEG (-1073741824)

/* traceback flags, reference mz_yama.c:24-26 */
#define MZO_FC 0
#define MZO_FI 1
#define MZO_FD 2

/* bump arena for per-thread scratch (NULL = plain malloc/free) */
typedef struct mzo_arena { char *base; size_t cap, used; } mzo_arena;

typedef struct mzo_scores {
    int ss[128][128];   /* reference mz_scores.c:34-54  */
    int gop[16];        /* reference mz_scores.c:56-80  */
    int gap_open;
    int gap_extend;
} mzo_scores;

/* reference mz_scores.c:94-107 (HOXD70, open 400, extend 30) / :109-122 (HOXD85, 600/50) */
void mzo_scores_hoxd70(mzo_scores *sc);
void mzo_scores_hoxd85(mzo_scores *sc);

/* error codes = which reference fatal() the input would have hit (mz_yama.c:58-71) */
enum {
    MZO_OK = 0,
    MZO_E_TERMINATION = 1,   /* "LB and RB not terminated properly"  mz_yama.c:58-59 */
    MZO_E_NARROW      = 2,   /* "RB[%d] - LB[%d] < %d"               mz_yama.c:63-65 */
    MZO_E_LB_MONO     = 3,   /* "LB not monotonic"                    mz_yama.c:67-68 */
    MZO_E_RB_MONO     = 4,   /* "RB not monotonic"                    mz_yama.c:69-70 */
    MZO_E_TRACEBACK   = 5,   /* traceback left the grid               mz_yama.c:274-276,290 */
    MZO_E_EMIT        = 6    /* i!=M || j!=N after emit               mz_yama.c:310-312 */
};

/* band validity prologue; returns MZO_OK or the error, *cells = tback_size (mz_yama.c:60-66) */
int mzo_yama_check(int M, int N, const int *LB, const int *RB, int64_t *cells, int *bad_row);

/* Faithful restatement: O(K*L) table look-ups per cell, as mz_yama.c:97-255 does it.
 * out must hold (M+N)*(K+L) bytes.  final3 (optional) receives C,D,I at (M,N).
 * tb (optional) receives the traceback bytes in the reference's band-packed row-major
 * order (tback_size bytes). Returns MZO_OK or an error code. */
int mzo_yama_faithful(const uint8_t *A, int K, int M, const uint8_t *B, int L, int N,
                      const int *LB, const int *RB, const mzo_scores *sc,
                      uint8_t *out, int *OM, int32_t *final3, uint8_t *tb);

/* Integer-exact O(1)-per-cell restatement via per-column class/gap profiles
 * (SURVEY.md appendix A.4).  This is the executable specification of the GPU arithmetic.
 * Same contract as mzo_yama_faithful. */
int mzo_yama_profile(const uint8_t *A, int K, int M, const uint8_t *B, int L, int N,
                     const int *LB, const int *RB, const mzo_scores *sc,
                     uint8_t *out, int *OM, int32_t *final3, uint8_t *tb);

/* reference mz_preyama.c:17-35 */
void mzo_smooth(int *LB, int *RB, int M, int N, int radius);

/* Batch driver used for the CPU baseline: runs pairs [0,n) with `threads` OpenMP threads
 * (one pair per thread at a time).  All arrays are packed pools with per-pair offsets.
 * variant 0 = faithful, 1 = profile.  om[n], hash[n] (FNV-1a 64 over OM + out bytes).
 * Returns the number of pairs that failed validation. */
int mzo_yama_batch(int n, const int *K, const int *L, const int *M, const int *N,
                   const int64_t *offA, const int64_t *offB, const int64_t *offBand,
                   const uint8_t *poolA, const uint8_t *poolB, const int *poolLB, const int *poolRB,
                   const mzo_scores *sc, int variant, int threads,
                   int *om, uint64_t *hash, int64_t *cells_done);

uint64_t mzo_fnv1a(const uint8_t *p, int64_t n, uint64_t h);

#ifdef __cplusplus
}
#endif
#endif
