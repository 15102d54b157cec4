/* tests/tools/hostprof/sampler.c -- a sampling profiler small enough to link into the profiling build: SIGPROF on the
 * process's CPU time (every thread that burns CPU gets its share of the signals), a backtrace per sample, and at exit the samples in a file
 * for tests/tools/hostprof/report.py: per function of the program, the samples in it AND in the library code it called
 * (malloc, memcpy, vfprintf ...), and which library functions were on top.  MZ_SAMPLER=0 turns it off. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <pthread.h>
#include <time.h>
#include <unistd.h>
#include <sys/syscall.h>

#define DEPTH 24
#define MAXS (1 << 20)
static void *g_pc[MAXS][DEPTH];
static unsigned char g_n[MAXS];
static volatile int g_count;
static pid_t g_main_tid;
static volatile int g_stop;

static void on_prof(int sig)
{
    const int i = __sync_fetch_and_add(&g_count, 1);
    (void)sig;
    if (i < MAXS) g_n[i] = (unsigned char)backtrace(g_pc[i], DEPTH);
}

/* one line per sample into MZ_SAMPLER_OUT (default sampler.out): the frames from the top, "b<hex address>" for the program's own
 * code (static functions have no dynamic symbol: tests/tools/hostprof/report.py names them with addr2line), "l<name>" for library code */
static void dump(void)
{
    struct itimerval off = { { 0, 0 }, { 0, 0 } };
    const char *path = getenv("MZ_SAMPLER_OUT"), *match = getenv("MZ_SAMPLER_MATCH");
    const int exec = match == NULL;                        /* linked into the (non-PIE) profiling program: absolute addresses; preloaded beside a
                                                              shared library named by MZ_SAMPLER_MATCH (libmzamd): offsets into that library */
    FILE *f;
    int n, i, k;
    g_stop = 1;
    setitimer(ITIMER_PROF, &off, NULL);
    if (!match) match = "roast_prof";
    f = fopen(path ? path : "sampler.out", "w");
    if (!f) return;
    n = g_count < MAXS ? g_count : MAXS;
    for (i = 0; i < n; ++i) {
        for (k = 2; k < g_n[i]; ++k) {                     /* frames 0, 1: the handler and the signal trampoline */
            Dl_info d;
            if (!dladdr(g_pc[i][k], &d)) { fprintf(f, "l? "); continue; }
            if (d.dli_fname && strstr(d.dli_fname, match)) fprintf(f, "b%lx ", (unsigned long)((char *)g_pc[i][k] - (exec ? (char *)0 : (char *)d.dli_fbase)));
            else fprintf(f, "l%s ", d.dli_sname ? d.dli_sname : "?");
        }
        fputc('\n', f);
    }
    fclose(f);
}

/* MZ_SAMPLER_WALL=<microseconds>: the MAIN thread only, every so many microseconds of wall time (what the serial parts of a run are
 * doing, waits included), instead of every thread by CPU time (whose clock ticks at the kernel's 100 Hz on these boxes) */
static void *wall_thread(void *arg)
{
    const long us = (long)(size_t)arg;
    struct timespec d = { 0, us * 1000 };
    sigset_t all;
    sigfillset(&all); pthread_sigmask(SIG_BLOCK, &all, NULL);
    while (!g_stop) { nanosleep(&d, NULL); syscall(SYS_tgkill, getpid(), g_main_tid, SIGPROF); }
    return NULL;
}

__attribute__((constructor)) static void start(void)
{
    struct itimerval it = { { 0, 1000 }, { 0, 1000 } };
    struct sigaction sa;
    void *prime[4];
    const char *e = getenv("MZ_SAMPLER");
    if (e && atoi(e) == 0) return;
    backtrace(prime, 4);                                   /* (loads libgcc's unwinder outside the handler) */
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_prof; sa.sa_flags = SA_RESTART;
    sigaction(SIGPROF, &sa, NULL);
    e = getenv("MZ_SAMPLER_WALL");
    if (e && atoi(e) > 0) {
        pthread_t t;
        g_main_tid = (pid_t)syscall(SYS_gettid);
        pthread_create(&t, NULL, wall_thread, (void *)(size_t)atoi(e));
    } else setitimer(ITIMER_PROF, &it, NULL);
    atexit(dump);
}
