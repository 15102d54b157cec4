/* mz_pack.h -- internal: host side of mz_yama_batch()'s link formats (mz_pack.c) */
#ifndef MZAMD_MZ_PACK_H
#define MZAMD_MZ_PACK_H
#include <stddef.h>
#include <stdint.h>

/* n bytes of column text -> (n + 1) / 2 bytes of class nibbles */
void mz_pack_classes(const uint8_t *src, size_t n, uint8_t *dst);
/* the same through streaming stores to a 32-byte aligned slot of `slot` bytes (a multiple of 32), padded */
void mz_pack_classes_stream(const uint8_t *src, size_t n, uint8_t *dst, size_t slot);
/* the M step bytes alone, streaming, to a 32-byte aligned slot (a multiple of 32 bytes, >= M); the OR of all steps */
uint32_t mz_pack_band_nib_stream(const int *LB, const int *RB, int M, uint8_t *dst, size_t slot);
/* LB[0], RB[0], M bytes (LB step | RB step << 4) to dst (8 + M bytes); the OR of all steps: valid iff < 16 */
uint32_t mz_pack_band_nib(const int *LB, const int *RB, int M, uint8_t *dst);
/* LB[0], RB[0], M bytes of LB steps, M bytes of RB steps (8 + 2M bytes; every step 0..255) */
void mz_pack_band_bytes(const int *LB, const int *RB, int M, uint8_t *dst);
/* merged columns of one pair from its 2-bit edit script (reference mz_yama.c:293-313) */
void mz_assemble_cols(int K, int L, int M, int N, const uint8_t *A, const uint8_t *B, const uint8_t *script, int om, uint8_t *out);
/* one row of a merged block (mz_preyama_batch): n source bytes; squeeze 0: all of them are its bytes, 1: those whose bit
 * in `keep` is set, 2: those that are not dashes (tmp: n + 16 bytes of scratch for 1 and 2); `ops`: one bit per merged
 * column -- the row's next byte, or a dash */
typedef struct mz_rowspec { const uint8_t *src; int n, squeeze; const uint64_t *keep, *ops; uint8_t *tmp; } mz_rowspec;
void mz_assemble_rows(int nrows, const mz_rowspec *rows, int om, uint8_t *out);
#endif
