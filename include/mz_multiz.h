/* include/mz_multiz.h -- the multiz driver with all of its pairwise merges run as GPU batches
 * (SURVEY.md section 8, row f1).  Replaces the main loop of reference multiz.c:60-177 and its command
 * line (:180-294); output is byte-identical to the stock binary's.
 */
#ifndef MZAMD_MZ_MULTIZ_H
#define MZAMD_MZ_MULTIZ_H

#include <stdio.h>
#include "maf.h"

/* All blocks of a MAF file, in file order (reference maf.c mafReadAll(); `verbose` echoes comment lines to
 * stdout as the stock reader does).  Errors end the program with the reader's messages. */
struct mafAli *mz_maf_read_all(const char *path, int verbose);

/* Merge two position-sorted block lists that share their top (reference) row, contig by contig in the order
 * of list1 (reference multiz.c:267-275 + multiz()).  Merged blocks go to `out`, unused parts of the inputs to
 * out1 / out2 (NULL: dropped); blocks of contigs present in one list only stay in *list1 / *list2.  Every
 * pre_yama() of the run is enumerated first and aligned in one (v == 1) or two (v == 0) GPU batches. */
int mz_multiz_lists(struct mafAli **list1, struct mafAli **list2, int v, int radius, int min_output_wid,
                    FILE *out, FILE *out1, FILE *out2);

/* The same run in three steps, so that several independent runs -- the merges at sibling nodes of a guide tree --
 * share their GPU batches: prepare() walks the lists and records events and merges (has_out1/2: whether unused
 * parts are wanted), align() runs the pending merges of ALL given runs as one batch per wave, finish() replays a
 * run into its sinks and frees it. */
struct mz_mzrun;
struct mz_mzrun *mz_multiz_prepare(struct mafAli **list1, struct mafAli **list2, int v, int radius, int min_output_wid,
                                   int has_out1, int has_out2);
void mz_multiz_align(struct mz_mzrun **runs, int n);
void mz_multiz_finish(struct mz_mzrun *run, FILE *out, FILE *out1, FILE *out2);
/* A run whose output goes on to another step of the same process ends in LISTS instead: keep_blocks() between prepare() and align()
 * (nothing is rendered as text then), finish_lists() instead of finish().  Every block is what a reader of the text would hold: the
 * score with one decimal, a source "x.x" as "x" (what mafWrite() prints and the MAF reader makes of it). */
void mz_multiz_keep_blocks(struct mz_mzrun *run);
void mz_multiz_finish_lists(struct mz_mzrun *run, struct mafAli **out, struct mafAli **out1, struct mafAli **out2);

/* the multiz command line: [R=?] [M=?] file1 file2 v [out1 out2] [nohead] [all] */
int mz_multiz_main(int argc, char **argv);

/* the multic command line (reference multic.c): [s=?] [R=?] [M=?] [C=?] file1 file2 v [out1 out2] [nohead] [all].
 * Two block lists topped by the same reference, single coverage not required: every overlapping pair of blocks
 * with no species in common is merged over its overlap (all of them enumerated first and aligned as GPU batches),
 * then the stretches no merge used are printed.  Output is byte-identical to the stock binary's. */
int mz_multic_main(int argc, char **argv);

/* The reference-guided tree driver in one process (reference auto_mz.c "roast", speciesTree.c, and the
 * maf_project / multiz / multic chain it spawns), same command line:
 *     [+-] [R=?] [M=?] [P=?] [T=?] [X=?] [C=?] E=reference-species species-guid-tree maf-source... destination
 * No temporary files, no child processes, one GPU start-up; the merges of sibling subtrees share GPU batches.
 * The destination holds, block for block, what the stock roast writes. */
int mz_roast_main(int argc, char **argv);

#endif
