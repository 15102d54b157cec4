/* include/mz_scores.h -- drop-in for reference mz_scores.h:6-19.
 *
 * The reference DEFINES ss/gop/gap_open/gap_extend in its header (tentative definitions,
 * mz_scores.h:8-11); here they are declared extern and defined once in the shim.  The GPU path
 * reads them at call time (mz_sync_scores in mz_host.c), so a caller that switches tables with
 * init_scores85() -- or fills its own class-structured table -- is honoured.
 */
#ifndef MZAMD_MZ_SCORES_H
#define MZAMD_MZ_SCORES_H

#include "maf.h"

extern int **ss;          /* 128 x 128 substitution scores            */
extern int *gop;          /* 16 quasi-natural gap-open penalties      */
extern int gap_open, gap_extend;

#define SS(c,d) ss[c][d]
#define GAP(s,t,u,v) gop[((s)<<3)+((t)<<2)+((u)<<1)+(v)]
#define GAP2(s,t,u,v) GAP(((s) == '-'), ((t) == '-'), ((u) == '-'), ((v) == '-'))

void init_scores70(void);                                           /* reference mz_scores.c:94-107  */
void init_scores85(void);                                           /* reference mz_scores.c:109-122 */
double mafScoreRange(struct mafAli *maf, int start, int size);      /* reference mz_scores.c:124-152 */
/* (ours) mafScoreRange() works from per-column class counts when ss / gop have the structure of the reference
 * tables; the model is derived when the ss / gop POINTERS change.  Call this after pointing them at new tables
 * and before scoring from several threads. */
void mz_score_profile_sync(void);

#endif
