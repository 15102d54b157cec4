/* include/mz_yama.h -- drop-in for reference mz_yama.h:22.
 *
 * void yama(A, K, M, B, L, N, LB, RB, &OAL, &OM): global alignment of two alignment blocks
 * under sum-of-pairs scores with quasi-natural gap costs inside the band LB[]..RB[].
 *   A  : 1-based array of M column pointers, column i = K bytes (rows of block 1)
 *   B  : 1-based array of N column pointers, column j = L bytes
 *   LB, RB : int[M+1], bounds of DP row i (0 <= i <= M); must satisfy mz_yama.c:58-71
 *   *OAL : receives a 1-based array of *OM column pointers over one contiguous buffer of
 *          *OM * (K+L) bytes; both blocks malloc()ed; free with free(OAL[1]); free(OAL+1);
 * Same error behaviour as the reference: a violated precondition prints the reference's
 * message to stderr and exit(1)s (util.c:21-30).  This implementation runs the DP on the GPU
 * (a batch of one through mz_yama_batch(), include/mz_amd.h); a missing or failing HIP device
 * is fatal, there is no CPU path.
 */
#ifndef MZAMD_MZ_YAMA_H
#define MZAMD_MZ_YAMA_H

#ifndef MZAMD_UCHAR
#define MZAMD_UCHAR
typedef unsigned char uchar;     /* reference util.h:20 */
#endif

void yama(uchar **A, int K, int M, uchar **B, int L, int N, int *LB, int *RB, uchar ***OAL, int *OM);

#endif
