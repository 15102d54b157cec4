/* mz_preyama.c -- the block-pair adapter around yama(): band construction and (next commit)
 * pre_yama() itself, with the reference's names so that stock drivers link unchanged
 * (reference mz_preyama.c:17-35 here).
 */
#include <stdlib.h>
#include "../../include/mz_preyama.h"

/* Turn the raw per-row column range implied by the shared reference row into a legal DP band
 * (reference mz_preyama.c:17-35): make LB a running maximum and RB a running minimum from the
 * right, then widen both by rad = min(M, radius) rows/columns. */
void smooth(int *LB, int *RB, int M, int N, int radius)
{
    const int rad = M < radius ? M : radius;
    int i, run;

    for (i = 0, run = 0; i <= M; ++i) {
        if (LB[i] > run) run = LB[i];
        LB[i] = run;
    }
    for (i = M, run = N; i >= 0; --i) {
        if (RB[i] < run) run = RB[i];
        RB[i] = run;
    }
    /* descending i: LB[i-rad] is still un-widened when it is read */
    for (i = M; i > rad; --i) {
        int a = LB[i] - rad;
        if (a < 0) a = 0;
        LB[i] = a < LB[i - rad] ? a : LB[i - rad];
    }
    for (; i >= 0; --i) LB[i] = 0;
    /* ascending i: RB[i+rad] is still un-widened when it is read */
    for (i = 0; i < M - rad; ++i) {
        int a = RB[i] + rad;
        if (a > N) a = N;
        RB[i] = a > RB[i + rad] ? a : RB[i + rad];
    }
    for (; i <= M; ++i) RB[i] = N;
}
