/* tests/tools/hostprof/fake_align.c -- a stand-in for the library's GPU entry points, for PROFILING THE HOST SIDE of the
 * tree driver on a machine without a GPU (tests/tools/hostprof/run.sh).  NOT an aligner: every merge comes back as the two
 * slices side by side, padded with '-' to the longer one (rows, base counts, a made-up score) -- blocks of the right
 * shape and size for the list walks, projections, replay and rendering above it to chew on, nothing more.  Test
 * infrastructure: never linked into libmzamd.so. */
#include "mz_amd.h"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

char *argv0 = (char *)"roast_prof";
int mz_roast_main(int argc, char **argv);
int mz_multiz_main(int argc, char **argv);
/* HOSTPROF_MAIN=multiz: the multiz command line instead of the tree driver's */
int main(int argc, char **argv) { const char *e = getenv("HOSTPROF_MAIN"); return e && strcmp(e, "multiz") == 0 ? mz_multiz_main(argc, argv) : mz_roast_main(argc, argv); }
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
void mz_fatalf(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); exit(1); }
void mz_fatal_status(const void *job, const void *out) { (void)job; (void)out; mz_fatalf("fake aligner: a refused job"); }
const char *mz_last_error(void) { return "fake aligner"; }
int mz_scores_explicit;
void mz_warm_start(void) {}
void mz_warm_wait(void) {}
int mzi_timing(void) { const char *e = getenv("MZ_TIMING"); return e ? atoi(e) : 0; }
int mz_yama_batch(int n, const mz_job *jobs, mz_out *outs) { (void)n; (void)jobs; (void)outs; mz_fatalf("fake aligner: mz_yama_batch() is not stood in for"); return -1; }
void mz_py_run_one(mz_job *job, unsigned char **flat, int *om) { (void)job; (void)flat; (void)om; mz_fatalf("fake aligner: mz_py_run_one() is not stood in for"); }

int mz_preyama_batch(int n, const mz_prejob *jobs, mz_preout *outs)
{
    size_t total = 0, at = 0;
    unsigned char *blk;
    int i;
    const double t0 = now_s();
    for (i = 0; i < n; ++i) {
        const int rows = jobs[i].K + jobs[i].L1 - 1, OM = jobs[i].M_all > jobs[i].N_all ? jobs[i].M_all : jobs[i].N_all;
        total += (size_t)rows * (size_t)OM + (size_t)rows * sizeof(int) + 8;
    }
    blk = (unsigned char *)malloc(total + 16);
    memset(outs, 0, (size_t)n * sizeof *outs);
#pragma omp parallel for schedule(static)
    for (i = 0; i < n; ++i) { (void)i; }
    for (i = 0; i < n; ++i) {
        const mz_prejob *j = &jobs[i];
        const int rows = j->K + j->L1 - 1, OM = j->M_all > j->N_all ? j->M_all : j->N_all;
        int *size, r, c;
        mz_preout *o = &outs[i];
        at = (at + 7) & ~(size_t)7;
        size = (int *)(blk + at); at += (size_t)rows * sizeof(int);
        o->rows = blk + at; at += (size_t)rows * (size_t)OM;
        o->size = size; o->OM = OM; o->status = MZ_OK; o->null_result = 0; o->score = 100.0 * OM;
        for (r = 0; r < rows; ++r) {
            const char *src = r < j->K ? j->rows1[r] : j->rows2[r - j->K + 1];
            const int len = r < j->K ? j->M_all : j->N_all;
            unsigned char *dst = o->rows + (size_t)r * (size_t)OM;
            int bases = 0;
            memcpy(dst, src, (size_t)len);
            memset(dst + len, '-', (size_t)(OM - len));
            for (c = 0; c < len; ++c) bases += src[c] != '-';
            size[r] = bases;
        }
    }
    outs[0].block = blk;
    if (mzi_timing()) fprintf(stderr, "{\"fake_preyama_batch\": {\"merges\": %d, \"seconds\": %.3f}}\n", n, now_s() - t0);
    return 0;
}
void mz_free_preouts(int n, mz_preout *outs)
{
    int i;
    for (i = 0; i < n; ++i) { free(outs[i].block); outs[i].block = NULL; outs[i].rows = NULL; outs[i].size = NULL; }
}
