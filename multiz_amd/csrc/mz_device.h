/* mz_device.h -- internal: launchers implemented in mz_device.hip, called by the C host shim. */
#ifndef MZ_DEVICE_H
#define MZ_DEVICE_H
#include "../../include/mz_amd.h"
#ifdef __cplusplus
extern "C" {
#endif
/* all asynchronous on `stream` (a hipStream_t) */
int mzk_upload_scores(const mz_score_model *m, void *stream);
int mzk_plan(const mz_dev_batch *b, void *stream);
/* expand the host path's delta-coded band bounds into the int32 pools (see k_unband) */
int mzk_unband(int n, const int32_t *bandLen, const int32_t *lb0, const int32_t *rb0, const int64_t *offBand, const int64_t *offC, const uint8_t *fmt,
               const uint8_t *packed, const uint8_t *exceptions, int32_t *poolLB, int32_t *poolRB, void *stream);
/* class nibbles (mz_pack.c) -> one canonical byte per class; nbytes_out a multiple of 8 */
int mzk_unnib(const void *nibbles, void *bytes, long long nbytes_out, void *stream);
/* what mz_yama_batch() copies back: hdr (64 B, spare), one record
 * per pair, the edit scripts at two bits per merged column (pair p at byte recs[p].off of `packed`) */
typedef struct mz_res_rec { int32_t status, badrow, om, f[3]; int64_t off; int64_t cells; /* band cells of the pair (0 unless status is MZ_OK): the host adds them up -- 50 000 atomic adds on one address were 0.7 ms of a batch */ } mz_res_rec;
int mzk_script_pack(const mz_dev_batch *b, void *hdr, void *recs, void *packed, void *stream);
int mzk_prep(const mz_dev_batch *b, void *stream);
int mzk_dp(const mz_dev_batch *b, void *stream);
int mzk_walk(const mz_dev_batch *b, void *stream, int beside_dp);   /* beside_dp: another batch's DP runs at the same time */
int mzk_emit(const mz_dev_batch *b, void *stream);
/* the same three phases on the slice [first, first+count) of the batch */
int mzk_dp_range(const mz_dev_batch *b, int first, int count, void *stream);
/* ... with the caller's own side streams for the DP kernels of a batch that has several kinds of pairs (n = 0: all on `stream`;
 * lanes == NULL: the device's own set, one per device) -- hipStream_t / hipEvent_t as void * */
typedef struct mz_dp_lanes { int n; void *stream[4]; void *fork; void *join[4]; } mz_dp_lanes;
int mz_dp_kinds(int dp_hint);          /* DP kernels a batch with this mz_dp_hint() launches */
int mzk_dp_range_on(const mz_dev_batch *b, int first, int count, void *stream, const mz_dp_lanes *lanes);
/* ... and an event (hipEvent_t as void *) for "these DP kernels are through": *done_set = 1 when it rode on the one kernel's own dispatch
 * packet (a batch of one row-parallel kind), 0 when the caller has to record it on `stream` */
int mzk_dp_range_ev(const mz_dev_batch *b, int first, int count, void *stream, const mz_dp_lanes *lanes, void *done, int *done_set);
int mzk_walk_range(const mz_dev_batch *b, int first, int count, void *stream, int beside_dp);
/* bytes (a multiple of 16, both addresses 16-byte aligned) from src to dst by a kernel on `stream`: either may be pinned host memory */
int mzk_link_copy(void *dst, const void *src, size_t bytes, void *stream);
int mzk_emit_range(const mz_dev_batch *b, int first, int count, void *stream);
/* device side of pre_yama() around the DP (kernels/prepost.inc): all device pointers */
typedef struct mz_pre_batch {
    int n;
    const int32_t *K, *L, *Ma, *Na, *rad;      /* K: rows of the first block, its top row included; L: rows of the second below its top row */
    const int32_t *v;                          /* 1: one-stage merge (all K rows align); 0: two stages (mz_preyama.c:265-336) */
    int stride64;                              /* 1: the text rows of a slice lie (columns rounded up to 64) bytes apart -- the host-buffer path,
                                                  whose rows travel as whole 32-byte lines of class nibbles; 0: columns bytes apart */
    const int64_t *offT1;                      /* the first block's K rows; the second block's L + 1 rows follow them */
    const uint8_t *txt;                        /* the text as bytes, or NULL: then ... */
    const uint8_t *nib;                        /* ... as class nibbles, two per byte (mz_pack.c; text offset i in nibble i & 1 of byte i >> 1) */
    int lds_bytes;                             /* > 0 (with nib): k_pre keeps a pair's nibbles and scratch in this much dynamic LDS -- the host sizes it
                                                  for the batch's largest pair (mz_pre_lds_bytes()) -- ... */
    int lds16;                                 /* ... with int16 scratch (no slice of the batch has 32 768 columns) */
    const int64_t *offScr;
    int32_t *scr;
    int32_t *nullres;
} mz_pre_batch;
/* what mz_preyama_batch() copies back instead of rows (k_fin, kernels/prepost.inc): a record per merge, bases per row,
 * and per merged column / slice column the bits from which the host puts the rows together out of the caller's own text */
typedef struct mz_pre_rec { int32_t status, badrow, nullres, stage, M, N, om, om1; int64_t score; int64_t cells; /* band cells of the yama() calls that ran for this merge: the host adds them up */ } mz_pre_rec;
typedef struct mz_fin_batch {
    int any0;                                  /* the batch has two-stage merges (b2 is valid) */
    const int64_t *offRow;                     /* first of merge p's K + L entries of `size` */
    int32_t *size;
    const int64_t *offMask;                    /* byte offset (a multiple of 8) of merge p's mask block in `masks` */
    uint8_t *masks;
    mz_pre_rec *recs;
    int cols;                                  /* merged columns k_fin stages per round: MZ_FIN_COLS(rows of the batch's widest merged block) */
    int lds_bytes;                             /* its dynamic LDS: MZ_FIN_LDS(those rows) */
} mz_fin_batch;
#define MZ_FIN_COLS(W) ((W) >= 512 ? 64 : (W) <= 128 ? 256 : (32768 / (W)) & ~63)
#define MZ_FIN_LDS(W) (((((MZ_FIN_COLS(W) + 1) * (W) + 20 + 15) & ~15) + (W) + 15) & ~15)
/* k_pre's dynamic LDS for one pair: its nibbles, five scratch arrays of `elem`-byte integers, a counter per row */
#define MZ_PRE_LDS(text_bytes, Ma, Na, W, elem) (((text_bytes) / 2 + (size_t)(elem) * (2 * ((size_t)(Na) + 2) + 3 * ((size_t)(Ma) + 2)) + 4 + 4 * (size_t)(W) + 15) & ~(size_t)15)
/* k_pre: A, B and the band of every pair where the DP kernels read them; bases per row and rmColDash's verdicts into f */
int mzk_pre(const mz_pre_batch *q, const mz_fin_batch *f, const mz_dev_batch *b, void *stream);
/* between the two stages of the v == 0 merges: the second yama() job of every such pair of b1 into b2 (same indices;
 * the other pairs of b2 get M = 0, which the plan refuses) */
int mzk_mid(const mz_pre_batch *q, const mz_dev_batch *b1, const mz_dev_batch *b2, void *stream);
/* b2: the second stage of the two-stage merges (ignored unless f->any0) */
int mzk_fin(const mz_pre_batch *q, const mz_fin_batch *f, const mz_dev_batch *b1, const mz_dev_batch *b2, void *stream);
const char *mzk_last_error(void);
void mzk_set_abreast(int k);           /* DPs of k consecutive batches run side by side (kernel choice of mzk_dp_range) */
void mzk_release_device(int dev);      /* destroy the launchers' side streams and events of one GPU (mz_finalize) */
#ifdef __cplusplus
}
#endif
#endif
