/* include/mz_amd.h -- C ABI of libmzamd.so: the MI355X (gfx950) implementation of the multiz
 * block-pair merge DP.  Plain pointers and sizes only; no C++/torch types.
 *
 * What each group of entry points replaces in the reference (multiz/multiz @ /root/reference):
 *
 *   mz_yama_batch()            N independent calls of   yama()        mz_yama.h:22 / mz_yama.c:50-320
 *   mz_dev_plan/dp/walk/emit   the phases of yama():    validity      mz_yama.c:58-71
 *                                                       DP            mz_yama.c:83-255
 *                                                       traceback     mz_yama.c:257-291
 *                                                       emit          mz_yama.c:293-313
 *   mz_set_scores()            init_scores70()/85()     mz_scores.c:94-122 (tables :34-81)
 *
 * The reference-signature drop-ins themselves -- yama(), pre_yama(), smooth(), mafBuild(),
 * rmColDash(), mapping(), init_scores70(), init_scores85(), mafScoreRange() -- are declared in
 * the sibling headers mz_yama.h, mz_preyama.h, mz_scores.h with the reference's own prototypes;
 * they are thin C wrappers over the batch entry points below (a batch of one).
 *
 * A "batch" is a set of independent block pairs.  Device-side data layout (all in HBM):
 *
 *   poolA  : for pair p, M*K bytes at offA[p]; column r (1..M) = K contiguous bytes at
 *            offA[p] + (r-1)*K  -- the reference's own column-major packing
 *            (mz_preyama.c:203-214) minus the 1-based pointer array.
 *   poolB  : same for the L x N block (mz_preyama.c:174-180).
 *   poolLB/poolRB : int32, M+1 entries per pair at offBand[p]  (LB[]/RB[] of mz_yama.h:10-13).
 *   tbw    : traceback workspace: one byte per (anti-diagonal step, lane), four steps per
 *            dword, so every store is one coalesced 256-byte row per wave.
 *   script : edit script per pair, reverse order, one byte per output column.
 *   out    : merged blocks; pair p's OM columns of K+L bytes at offOut[p] (*OAL of yama()).
 */
#ifndef MZ_AMD_H
#define MZ_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Layout / ownership version of the additive batch API below (mz_job, mz_out, mz_prejob, mz_preout): bumped whenever a
 * struct of this header changes size or the owner of a result changes, so that a caller built against another
 * revision fails to compile (#if MZ_AMD_ABI != ...) instead of corrupting its heap.
 *   2 (round 3): mz_out gained `block`; `cols` points into it (free with mz_free_outs(), not free(cols), unless n == 1)
 *   3 (round 4): mz_preout gained `block`; `rows` / `size` point into it (free with mz_free_preouts(), not free(rows)) */
#define MZ_AMD_ABI 3
int mz_abi_version(void);                  /* MZ_AMD_ABI of the library that is loaded */

#define MZ_NEG      (-1073741824)          /* reference mz_yama.c:29, INT_MIN/2 */
#define MZ_BIG      (0x3fffffff)
#define MZ_FC 0                            /* reference mz_yama.c:24-26 */
#define MZ_FI 1
#define MZ_FD 2

/* per-pair status: 0 ok, 1..4 = the reference fatal() the input would hit (mz_yama.c:58-71),
 * 5/6 traceback/emit invariants (mz_yama.c:274-276,290,308-312), >=16 = limits of this build */
enum {
    MZ_OK = 0,
    MZ_E_TERMINATION = 1,
    MZ_E_NARROW = 2,
    MZ_E_LB_MONO = 3,
    MZ_E_RB_MONO = 4,
    MZ_E_TRACEBACK = 5,
    MZ_E_EMIT = 6,
    MZ_E_ROWS = 16,        /* K or L outside 1..255 (per-column class counters are bytes, the substitution row
                              vector is int16) */
    MZ_E_SHAPE = 17,       /* M < 1 or N < 1 */
    MZ_E_RANGE = 18,       /* M + N >= 2^30 (step counters are int32).  There is no bound on K*L*(M+N): like the
                              reference (mz_yama.c:50-71) the exact kernels compute in plain int32, the row-parallel
                              kernels re-base their scores, and the guard-dropping kernels are only chosen where the
                              plan proves every reachable score stays above -2^29 */
    MZ_E_WORKSPACE = 19,   /* workspace too small for this batch (caller must re-plan) */
    MZ_E_DEVICE = 20,      /* not computed: a device error (or malloc failure) ended the call before this pair's result */
    MZ_E_SENTINEL = 21     /* the traceback left the band of a pair whose scores can fall below the reference's "unreachable" sentinel
                              MININT = INT_MIN / 2 (mz_yama.c:29) -- K * L * (gap_open + max(gap_extend, max |ss|)) * (M + N + 2) >= 2^30: two
                              blocks of 100+ rows EACH over ~700 mismatching columns.  Down there sentinel states win comparisons in the
                              reference too; its walk then steps onto cells outside the band, reads whatever bytes of its one traceback array
                              lie there (mz_yama.c:98,277) and still arrives.  That is not reproduced: the pair is reported, never mis-aligned.
                              (The same walk failure on a pair whose scores cannot get there stays MZ_E_TRACEBACK: the reference's own error.) */
};

/* which DP kernel the plan picked for a pair:
 *   EXACT : anti-diagonal wavefront, every guard of mz_yama.c evaluated literally (any legal band
 *           whose rows leave their lane in time: RB[r] - LB[r+64] <= 62)
 *   STRIP : strip-mined fallback for wider bands
 *   FAST  : same wavefront for well-formed pairs -- connected band (LB[r] <= RB[r-1]) and scores
 *           below 2^29 -- where guards that can only touch unreachable (sentinel) states are
 *           dropped; outputs are identical ("fast path equivalence") */
enum { MZ_MODE_WF64 = 0, MZ_MODE_STRIP = 1, MZ_MODE_FAST = 2, MZ_MODE_FASTT = 3 /* FAST with tag-encoded picks */,
       MZ_MODE_TSTRIP = 4 /* STRIP for well-formed pairs: 64-row strips with FASTT's guard-dropping tagged arithmetic (bands wider AND
                               higher than the rolling wavefront and the row-parallel kernels take; kernels/strip.inc) */,
       MZ_MODE_ROW = 5 /* FASTT arithmetic, lane = column, one band row per iteration (bands <= 63 wide) */,
       MZ_MODE_COL = 6 /* MZ_MODE_ROW on the transposed problem (bands <= 63 high) */,
       MZ_MODE_ROWR = 7, MZ_MODE_COLR = 8 /* ROW / COL for scores too large for the 2^30 ring lift: the prefix
                                             maximum runs on lanes rotated to the band start (two ds_bpermute) */,
       MZ_MODE_WIDE = 9, MZ_MODE_WIDESTRIP = 10 /* blocks of 128..255 rows: WF64 / STRIP with int16 gap vectors */,
       MZ_MODE_LAG = 11 /* ROW for bands with rows of 65..127 columns: the 64-column periods run a few rows apart (kernels/lag.inc) */,
       MZ_MODE_TROLL = 12 /* FASTT for bands wide AND high: a lane that is still busy when its next row is due starts it late, and every
                              later row with it (kernels/roll.inc); the strips' traceback layout */,
       MZ_MODE_DUO = 13 /* FASTT on a workgroup of TWO waves, 128 rows in flight: bands wide AND high whose anti-diagonals cross at most 128 rows
                            (RB[r+1] - LB[r+128] <= 126); the rows above across the wave boundaries through LDS, one barrier per step
                            (kernels/duo.inc); FASTT's traceback layout on 128 lanes */ };

#define MZ_PLAN_FOLD_MAX 4096         /* batches of at most this many pairs: the plan shares a pair's rows out over 16 waves */
#define MZ_SCAN_AUX_BYTES(n) ((size_t)576 * ((size_t)(n) / 64 + 2) + ((n) <= MZ_PLAN_FOLD_MAX ? (size_t)768 * (size_t)(n) : 0))   /* mz_dev_batch.scanAux */
#define MZ_TOTALS 96        /* int64 entries of mz_dev_batch.totals */
typedef struct mz_dev_batch {
    int32_t n;
    int32_t dp_hint;       /* 0: unknown -- every DP kernel is launched, one after the other; else mz_dp_hint() of the plan's totals:
                              only the kernels that have pairs are launched, side by side when there are several */
    int32_t dp_grid;       /* 0: unknown; else mz_dp_grid() of the plan's totals: how many waves the DP kernels that take their pairs from a
                              counter need (three 10-bit fields, pairs / 8 rounded up: wavefront, blocks of 128+ rows, lagged) */
    int32_t dp_rows;       /* 0: unknown; else mz_dp_rows() of the plan's totals: the batch's row-parallel pairs (k_dp_row then launches
                              that many blocks, each taking its pair from the plan's list, instead of one per pair of the batch) */
    int32_t hint_gen;      /* mz_hint_generation() at the time dp_hint / dp_grid / dp_rows / walk_hint were derived.  The hints describe
                              the plan made under ONE kernel selection and score model; every run re-plans on the device under the
                              CURRENT ones (mz_enable_row / mz_enable_fast / mz_set_scores / init_scores85 may have been called since),
                              so hints stamped with another generation -- or not stamped, 0 -- are ignored: every DP kernel is launched
                              over the whole batch and the device chooses the walk, as if dp_hint were 0 */
    int32_t walk_hint;     /* 0: let the device choose the traceback-walk kernel of a batch that runs beside another batch's
                              DP (both kernels are launched, one returns at once); MZ_WALK_RUNS / MZ_WALK_CHASE: the caller
                              has read the plan's totals and chosen (mz_walk_choice()) -- one launch */
    /* inputs (device pointers) */
    const int32_t *K, *L, *M, *N;
    const int64_t *offA, *offB, *offBand;
    const uint8_t *poolA, *poolB;
    const int32_t *poolLB, *poolRB;
    /* plan outputs (device, n entries each) */
    int32_t *status;       /* MZ_OK or error                      */
    int32_t *badrow;       /* row index for the error message     */
    int32_t *mode;         /* MZ_MODE_*                           */
    int64_t *cells;        /* tback_size, reference mz_yama.c:60-66 */
    int32_t *edgeLo;       /* last step with a cell in column <= 1   (fast kernel phases) */
    int32_t *edgeHi;       /* first step with a cell in column N                          */
    int64_t *szTb, *szScript, *szOut, *szPrep;     /* per-pair sizes (dwords, bytes, bytes, dwords) */
    int64_t *offTb, *offScript, *offOut, *offPrep; /* exclusive prefix sums of the above            */
    int64_t *totals;       /* [0..2] totals of tb/script/out, [3] failed pairs, [4] prep total, [5] pairs on the wavefront kernels (lower half) and row-parallel pairs of blocks of four rows or more (upper half), [6], [7] spare,
                              [8] pairs of more than 127 rows (lower half) and pairs on the lagged kernel (upper half), [9] spare, [10] the batch chase's pair counter,
                              [11] rows (K+L) of all valid pairs in bits 0..43 and the number of MZ_MODE_TSTRIP and MZ_MODE_DUO pairs from bit 44 up, [12] bytes of the packed outputs (host path), [16..21] work counters of k_dp / k_dp_wide / k_dp_lag / k_dp_roll / k_dp_tstrip / k_dp_duo, [32..95] the pair lists' totals per (kind, size class): 128 ints; MZ_TOTALS entries in all */
    int32_t *packList;     /* n entries: the pairs of the wavefront kernels, of blocks of 128+ rows, of the lagged kernel and of the row-parallel kernels, one list after the other, each with the pairs of most cells first */
    int64_t *scanAux;      /* scratch of the prefix-sum kernels and the plan: MZ_SCAN_AUX_BYTES(n) bytes (8 sums, then 128 ints of list counts, per 64 pairs; 16 x 12 ints per pair for batches of at most MZ_PLAN_FOLD_MAX pairs) */
    /* workspaces + results (device) */
    uint32_t *tbw;
    uint8_t  *script;
    uint8_t  *out;
    uint32_t *prep;        /* MZ_MODE_COL / COLR pairs: the band column by column (2 ints per column)   */
    int64_t capTb, capScript, capOut, capPrep;     /* capacities in the same units as the sizes */
    int32_t *om;           /* OM per pair                          */
    int32_t *final3;       /* C,D,I at (M,N), 3 per pair           */
} mz_dev_batch;

enum { MZ_WALK_AUTO = 0, MZ_WALK_RUNS = 1, MZ_WALK_CHASE = 2, MZ_WALK_REG = 3 };     /* REG: runs, the window in registers (beside DP kernels, thin blocks) */
/* the choice the device would make, from the plan's totals (host copy) of an n-pair batch */
int mz_walk_choice(int n, const int64_t *totals);
/* which DP kernels have pairs in this batch (mz_dev_batch.dp_hint; bits 8..15: the one with the most), from the plan's
 * totals (host copy) */
enum { MZ_DP_ROW = 1, MZ_DP_WAVEFRONT = 2, MZ_DP_WIDE = 4, MZ_DP_LAG = 8, MZ_DP_KNOWN = 16, MZ_DP_ROWBIG = 32,
       MZ_DP_STAMP = 0x20000,           /* a request too (measurements: tests/tools/c5_where.py): k_dp_row_lat leaves per pair p, in scanAux[3p .. 3p+2] (batches of
                                           at most MZ_PLAN_FOLD_MAX pairs), where its wave ran -- HW_ID | XCC_ID << 32 -- and when: s_memrealtime at its start and end */
       MZ_DP_SOLO = 0x40000,            /* a request: this batch is one of several whose row-parallel pairs TOGETHER leave the GPU's SIMDs at most a wave each (the chunks
                                           of a call of few long pairs): their DP takes the build whose waves cannot share a SIMD (kernels/row.inc: k_dp_row_solo) */
       MZ_DP_HELPERS_FIRST = 0x10000,
       MZ_DP_REQUESTS = 0x10000 | 0x20000 | 0x40000 };   /* the bits of dp_hint that are requests, not hints: kept whatever hint_gen says */   /* not a hint, a request (kept whatever hint_gen says): the batch's plan / prep / walk / emit kernels raise their waves'
                                             issue priority.  For batches whose small kernels must get through beside OTHER batches' DP waves at once (the chunk
                                             pipelines of mz_yama_batch / mz_preyama_batch: a late plan is an idle DP stream); a loss of 1.5-3 % for device-resident
                                             batches, whose helpers only have to be done a whole DP later */
int mz_dp_hint(int n, const int64_t *totals);
int mz_dp_grid(int n, const int64_t *totals);
int mz_dp_rows(int n, const int64_t *totals);
/* generation of the kernel selection + score model the devices hold now (>= 1 once initialised; it changes whenever a
 * new model is uploaded): the stamp for mz_dev_batch.hint_gen */
int mz_hint_generation(void);

typedef struct mz_score_model {
    int32_t S6[36];        /* 6x6 class matrix {A,C,G,T,-,other}, from ss[][] (mz_scores.c:34-54) */
    int32_t gap_open;      /* the single non-zero value of gop[] (mz_scores.c:78-79)              */
    int32_t gap_extend;
    int32_t g1, g2;        /* gap_open = g1*g2 with both <= 258 (int16 dot-product operands); 0 = none */
    int32_t pack;          /* spare                                                                       */
    int32_t row;           /* 1 (default): let the plan pick the row-parallel kernel (MZ_MODE_ROW)        */
} mz_score_model;


/* ---------------------------------------------------------------- library life cycle */

/* Select the GPU, create the stream, upload the HOXD70 tables.  Returns 0, or -1 with
 * mz_last_error() set.  There is NO CPU fallback: without a usable HIP device every entry
 * point fails. */
int  mz_init(int device);
/* The same on several GPUs of the node (devices == NULL: GPUs first..first+ngpu-1, first = MZ_DEVICE in the
 * environment, default 0 -- the GPUs an environment-driven start opens); the first is the primary device, where
 * the device-resident API (mz_dev_*) runs.  mz_yama_batch() then deals a large batch out over all of them -- one host
 * thread, one set of streams and staging buffers per GPU, the jobs dealt by weight in a snake so that every GPU gets the same mix
 * of long and short pairs (mz_preyama_batch() deals its merges the same way, by text volume);
 * block pairs are independent, so there is no exchange between GPUs.  Without an explicit call the first use of the
 * library reads MZ_DEVICE (first GPU, default 0) and MZ_NGPU (count, default 1) from the environment, which is how
 * the drivers mz_multiz / mz_multic use a whole node. */
int  mz_init_multi(int ngpu, const int *devices);
int  mz_device_count(void);               /* GPUs the library is running on (0 before initialisation) */
/* "<PCI bus id> <device name>" of the GPU context ctx (0 .. mz_device_count()-1) runs on; 0, or -1 */
int  mz_device_identity(int ctx, char *buf, int len);
void mz_finalize(void);
const char *mz_last_error(void);
/* the hipStream_t the library launches on (for callers that time with HIP events) */
void *mz_stream(void);

/* Score tables as the reference holds them (int ss[128][128] flattened row-major, int gop[16],
 * gap_extend).  Fails (-1) unless ss is constant on the six byte classes and gop has the
 * quasi-natural structure -- true of both reference tables. */
int  mz_set_scores(const int *ss_flat, const int *gop16, int gap_extend);

/* 0: run every pair on the exact kernels (all guards of mz_yama.c evaluated literally); 1 (default):
 * let the plan pick the fast kernel for well-formed pairs.  Outputs are identical either way. */
void mz_enable_fast(int on);
/* 0: keep the plan from choosing the row-parallel kernel (MZ_MODE_ROW; default on; MZ_NO_ROW=1 in the
 * environment also disables it).  Outputs are identical either way. */
void mz_enable_row(int on);

/* ---------------------------------------------------------------- host-buffer batch API */

typedef struct mz_job {
    int K, L, M, N;
    const unsigned char *A;   /* M columns of K bytes, contiguous, column r at A + (r-1)*K */
    const unsigned char *B;   /* N columns of L bytes                                      */
    const int *LB, *RB;       /* int[M+1]                                                  */
} mz_job;

typedef struct mz_out {
    int status;               /* MZ_OK or MZ_E_*                                           */
    int badrow;               /* row named by the reference's message for status 2         */
    int OM;                   /* number of merged columns                                  */
    int score[3];             /* C, D, I at grid point (M, N)                              */
    unsigned char *cols;      /* OM*(K+L) bytes: the merged columns; NULL on error.  Inside `block` of this or an
                                 earlier entry -- not a heap pointer of its own, except in a call with n == 1           */
    void *block;              /* non-NULL on the first pair of each chunk of the call: ONE malloc()ed block holding the
                                 merged columns of that chunk's pairs.  Release a call's results with mz_free_outs()
                                 (or, for n == 1: free(outs[0].cols), which is the block -- what yama() hands on)      */
} mz_out;

/* Align n independent block pairs on the GPU(s).  Returns the number of failed pairs, or -1 on a device error;
 * in either case every outs[i] is defined: status MZ_E_DEVICE and cols == NULL for pairs without a result -- and in either
 * case the caller releases outs with mz_free_outs(): after -1 the chunks that did come back still own their blocks.
 * What crosses the PCIe link is the byte CLASSES of the columns (two per byte), the band bounds as steps and, back,
 * a 32-byte record and a 2-bit edit script per pair; the merged columns (reference mz_yama.c:293-313) are assembled
 * on the host from the caller's own A and B, which must stay valid until the call returns.
 * Calls are serialised (the library state is process-wide).  MZ_TIMING=1 in the environment: one JSON line per call on
 * stderr (pairs, band cells, seconds, GCUPS, bytes over the link each way); MZ_TIMING=2: and one per chunk. */
int mz_yama_batch(int n, const mz_job *jobs, mz_out *outs);

/* Start the GPU in the background (HIP runtime, context, streams, code object: ~0.3 s of a process's first batch call)
 * and return at once; the first batch call waits for what is left of it.  For drivers that read files first; set the
 * score tables before the call.  Without it the first batch call does the same work itself. */
void mz_warm_start(void);
/* Wait for that thread (a no-op without one).  A program that may end without a single batch call calls this before it
 * returns from main(): the HIP runtime must not be torn down under a thread that is still starting it. */
void mz_warm_wait(void);
/* bytes the last mz_yama_batch() call moved over the PCIe link: to the device(s), and back */
void mz_link_bytes(int64_t *up, int64_t *down);
/* free the result blocks of a finished call (all n entries of it) and reset cols / block to NULL */
void mz_free_outs(int n, mz_out *outs);

/* ---------------------------------------------------------------- link images: the same traffic, moved by the caller
 *
 * What mz_yama_batch() moves over PCIe, for a caller that moves it itself -- multiz_amd/shard.py: the list exists on rank 0,
 * every other rank's share reaches ITS GPU over RCCL / xGMI (ncclSend / ncclRecv of independent shards).
 * The sender packs its jobs into an IMAGE (byte classes two per byte, band bounds as steps: DESIGN.md section 3), the rank that
 * aligns runs the image where it lies in HBM and fills a RESULT image (a record per pair + the edit scripts at two bits per
 * merged column), the sender assembles the merged columns from its own A and B.  Per C2 pair 3.1 KB out and 0.53 KB back instead
 * of the pools' 6.0 + 4.3 KB.
 *   mz_link_pack      host only (no GPU is touched): image and exception block from the library (256-byte aligned,
 *                     release with mz_link_free()); d describes them and travels beside them as it is (8 x int64)
 *   mz_link_plan      dev_image / dev_exc: the same bytes in device memory (dev_exc may be NULL when d->exc_bytes is 0).
 *                     Expands and plans on `stream`, waits for the plan, sets d->res_bytes: the result image's size
 *   mz_link_finish    DP, walk, script packing on `stream` into dev_result (d->res_bytes bytes of device memory); returns
 *                     without waiting.  One image at a time per process: plan and finish use the primary context's buffers
 *                     (not beside a running mz_yama_batch()); `stream` NULL = the library's own
 *   mz_link_assemble  host only: outs[] from the result image (in host memory) and the jobs it was packed from, exactly
 *                     what mz_yama_batch() would have returned; release with mz_free_outs()
 *   mz_link_parts     byte offsets of the image's parts: K L M N (int32 x n) offA offB offBand (int64 x n: offsets into the
 *                     EXPANDED pools) bandLen LB0 RB0 offC fmt, band steps, nibbles of A, nibbles of B, total -- at[16]
 *   mz_link_expand    only the expansion, into the caller's device buffers: dev_cols takes the byte pools (A at 0, B at
 *                     2 * al256(colsA / 2); 2 * (al256(colsA / 2) + al256(colsB / 2)) bytes, al256 = up to a multiple of 256), dev_LB / dev_RB
 *                     `band` int32 each -- with the image's own K..offBand a device-resident batch (mz_dev_batch) of its own
 * All return 0, or -1 with mz_last_error() (mz_link_assemble: the number of failed pairs, or -1 for an image that does not
 * belong to these jobs). */
typedef struct mz_link_desc { int64_t n, image_bytes, exc_bytes, colsA, colsB, band, steps, res_bytes; } mz_link_desc;
int mz_link_pack(int n, const mz_job *jobs, mz_link_desc *d, void **image, void **exc);
void mz_link_free(void *p);
int mz_link_plan(mz_link_desc *d, const void *dev_image, const void *dev_exc, void *stream);
int mz_link_finish(const mz_link_desc *d, void *dev_result, void *stream);
int mz_link_assemble(int n, const mz_job *jobs, const void *result, int64_t res_bytes, mz_out *outs);
int mz_link_parts(const mz_link_desc *d, int64_t at[16]);
int mz_link_expand(const mz_link_desc *d, const void *dev_image, const void *dev_exc, void *dev_cols, void *dev_LB, void *dev_RB, void *stream);

/* ---------------------------------------------------------------- pre_yama() batches: block text in, block text out
 *
 * N independent merges -- pre_yama(a1, a2, beg, end, radius, v, ...) of reference mz_preyama.c:152-359: one stage
 * (v = 1, :152-262) or two (v = 0: the first block's top row sits the first yama() out and is aligned against its
 * result by a second one whose band comes from mapping(), :265-336, the reference's two defects there included) --
 * given as the TEXT of the two blocks over their overlap.  Everything between the text and the text that needs the
 * alignment happens on the GPU (kernels/prepost.inc around the DP): column packing, removal of all-dash columns
 * (rmColDash), the band from the shared reference row and smooth(), yama() itself -- both of them for v = 0 --, the base
 * counts and mafScoreRange() of the block that mafBuild() would assemble (rows without a base left out).  What crosses
 * the PCIe link is the byte CLASSES of the text (two per byte: the DP and the score only distinguish A/a C/c G/g T/t,
 * '-' and "other") and, back, a record, the base counts and a few bits per merged column; the merged block's ROWS are
 * put together on the host from the caller's own text (which must stay valid until the call returns), on the
 * library's host threads, a chunk of the call at a time while the GPU works on the next ones.  The caller keeps what
 * only it knows: names, strands and start coordinates of the rows. */
typedef struct mz_prejob {
    int K;                        /* rows of the first block, its top row included (v = 0 needs K >= 2: with
                                     nothing below the top row pre_yama() returns NULL after writing the second
                                     block's slice to fpw2, mz_preyama.c:193-196 -- that stays with the caller)    */
    int L1;                       /* rows of the second block INCLUDING its top (reference) row: L = L1 - 1 */
    int M_all, N_all;             /* columns of the two slices (cend - cbeg + 1 of mz_preyama.c:167-172)    */
    int radius;
    const char *const *rows1;     /* K pointers: row r of the first slice is rows1[r][0 .. M_all)           */
    const char *const *rows2;     /* L1 pointers, the reference row first                                   */
    int v;                        /* 1: one-stage merge; 0: two stages                                      */
} mz_prejob;

typedef struct mz_preout {
    int status, badrow;           /* as in mz_out (yama()'s own refusals, limits of this build)             */
    int null_result;              /* 1: pre_yama() returns NULL -- no column of the second block (v = 0: or of the
                                     first block's lower rows) is left; 2: v = 0 with K = 1 (see mz_prejob.K);
                                     3: v = 0 and the two top rows hold different numbers of bases over the
                                     overlap -- the reference dies of "M3 not equals N3!!" (mz_preyama.c:330)  */
    int stage;                    /* which yama() call `status` is about (1 or 2)                            */
    int M, N;                     /* yama()'s M and N (after the dash columns went)                         */
    int OM;                       /* columns of the merged block                                            */
    double score;                 /* mafScoreRange(block, 0, OM) over the rows that keep a base             */
    const int *size;              /* bases per row, K + L entries (inside `block`, behind the rows)         */
    unsigned char *rows;          /* K + L rows of OM bytes, row after row; NULL without a block.  Inside `block` of
                                     this or an earlier entry of the call -- not a heap pointer of its own   */
    void *block;                  /* non-NULL on the first merge of each chunk of the call: ONE malloc()ed block holding
                                     the rows and base counts of that chunk's merges (mz_free_preouts())     */
} mz_preout;

/* Returns the number of pairs without a block (refused or NULL), -1 on a device error (mz_free_preouts() all the same: the
 * chunks that did come back own their blocks), -2 when the current score
 * tables lack the structure the device form of mafScoreRange() needs (symmetric classes): use the host path then.
 * MZ_TIMING=1: one JSON line per call on stderr (merges, band cells, seconds, GCUPS, link bytes); 2: and one per chunk. */
int mz_preyama_batch(int n, const mz_prejob *jobs, mz_preout *outs);
/* free the result blocks of a finished call (all n entries of it) and reset rows / size / block to NULL */
void mz_free_preouts(int n, mz_preout *outs);
/* bytes the last mz_preyama_batch() call moved over the PCIe link: to the device(s), and back; band cells it computed */
void mz_pre_link_bytes(int64_t *up, int64_t *down, int64_t *cells);

/* ---------------------------------------------------------------- device-resident API */

/* bytes of device memory needed for the per-pair plan/result arrays of an n-pair batch, and a
 * helper that carves them out of one allocation (256-byte aligned slices) */
size_t mz_dev_plan_bytes(int n);
void   mz_dev_carve(mz_dev_batch *b, void *plan_mem);

/* phases; all asynchronous on `stream` (hipStream_t; NULL = the library's stream) */
int mz_dev_plan(const mz_dev_batch *b, void *stream);   /* validity + sizes + offsets (+ the COL pairs' prep records).
                                                         * Sizing pass: capacities = INT64_MAX, prep = NULL; then read
                                                         * totals[0,1,2,4], allocate tbw/script/out/prep and run.      */
int mz_dev_dp(const mz_dev_batch *b, void *stream);     /* DP + traceback bytes             */
int mz_dev_walk(const mz_dev_batch *b, void *stream);   /* traceback -> edit script, OM     */
int mz_dev_emit(const mz_dev_batch *b, void *stream);   /* merged columns                   */
/* The whole path for one batch, phases one after the other on `stream`.  If ms != NULL it receives the
 * HIP-event time of {plan, dp, walk, emit} in milliseconds (synchronises the stream). */
int mz_dev_run(const mz_dev_batch *b, void *stream, float ms[4]);
/* Pipelined form for a stream of batches.  The DPs of successive batches run back to back on `stream`; plan
 * and prep of a batch run on one library stream and walk + emit on another, beside the DPs of the neighbouring
 * batches.  Calls in flight at the same time must use different workspaces (tbw/script/out/prep and the plan
 * arrays); the library keys a workspace by its tbw pointer and orders its reuse after the walk/emit of the
 * batch that used it last (at most 8 workspaces in rotation).  Three rotating workspaces keep everything but
 * the DP off the critical path; two work, with less overlap (small batches: mz_dev_pipeline_depth() + 1).  The batch's inputs must be complete when the
 * call is made, or ready_event (a hipEvent_t recorded after their producer) must be given; NULL otherwise.
 * mz_dev_wait() makes `stream` wait for everything issued so far; results of a batch are valid after it. */
int mz_dev_run_async(const mz_dev_batch *b, void *stream, void *ready_event);
int mz_dev_wait(void *stream);
/* Batches of at most 16 Ki pairs do not fill the GPU on their own: mz_dev_run_async() runs the DPs of up to this many
 * consecutive batches side by side (on its own streams beside `stream`).  Rotate one workspace more than this. */
int mz_dev_pipeline_depth(int n);

#ifdef __cplusplus
}
#endif
#endif
