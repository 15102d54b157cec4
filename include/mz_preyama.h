/* include/mz_preyama.h -- drop-in for reference mz_preyama.h:25,27 and the non-static helpers
 * of mz_preyama.c that other objects may link against (:17,38,87,111).
 *
 * pre_yama(a1, a2, beg, end, radius, v, fpw2): a1 and a2 share their top (reference) row over
 * positions beg..end; slice both, derive the DP band from the shared row, widen it by `radius`,
 * align with yama() (once for v == 1 "a1's top row is fixed", twice for v == 0) and return the
 * merged block, or NULL when nothing aligns (mz_preyama.c:152-359).
 */
#ifndef MZAMD_MZ_PREYAMA_H
#define MZAMD_MZ_PREYAMA_H

#include <stdio.h>
#include "maf.h"
#include "mz_yama.h"

struct pwuAliFiles;       /* reference align_util.h:51-54; opaque here */

struct mafAli *pre_yama(struct mafAli *a1, struct mafAli *a2, int beg, int end, int radius, int reference, FILE *fpw2);
/* reference mz_preyama.h:27 / mz_preyama.c:386-521 (no caller in the reference tree): merge the slice of a2 over
 * X[beg1..end1] with the slice of a3 over Y[begN..endN], the band taken from the pairwise alignment a1 of X and Y.
 * Calls the caller's connectionAgreement2() (reference align_util.c; a weak reference in the library). */
struct mafAli *pre_yama2(struct mafAli *a1, struct mafAli *a2, struct mafAli *a3, int beg1, int end1,
                         int begN, int endN, int radius, struct pwuAliFiles *pws);

void smooth(int *LB, int *RB, int M, int N, int radius);                                   /* :17-35  */
struct mafAli *mafBuild(uchar **A_new, int nrow, int ncol, struct mafAli *a2, int cbeg2,
                        struct mafAli *a3, int cbeg3, int top);                            /* :38-81  */
int *rmColDash(uchar **X, int *N, int row);                                                /* :87-108 */
int *mapping(uchar **A, int a_row1, int a_row2, int a_col1, int a_col2,
             uchar **B, int b_row1, int b_row2, int b_col1, int b_col2);                   /* :111-148 */

#endif
