/* mz_host.c -- C host side of libmzamd.so: device context, score-model hand-over, the
 * host-buffer batch entry point and the reference-signature yama() built on it.
 *
 * Host code stays C (as the reference is); the HIP kernels are reached through the mzk_*
 * launchers of mz_device.hip.  There is no CPU implementation of the DP in this library: when
 * no HIP device is usable every entry point fails (yama(): prints and exit(1)s, like any other
 * fatal condition of the reference, util.c:21-30).
 */
#include <limits.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "mz_ctx.h"
#include "../../include/mz_scores.h"
#include "../../include/mz_yama.h"

/* ------------------------------------------------------------------ error plumbing */

static __thread char g_err[512];             /* per thread: the GPUs of a multi-GPU batch are driven by one host thread each */
char *argv0;                               /* reference util.c:4; drivers set it in main() */

const char *mz_last_error(void) { return g_err; }

int mzi_set_err(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return -1;
}


/* same shape as the reference's fatalf(): "<argv0 basename>: message\n", exit(1) */
int mz_scores_explicit;                  /* set by mz_set_scores(), cleared by init_scores70/85() */

void mz_warm_wait(void);

/* MZ_TIMING in the environment, parsed once: unset or "0" = quiet, 1 = one JSON line per call, 2 = and one per chunk;
 * a value that is not a number (MZ_TIMING=yes) counts as 1 */
int mzi_timing(void)
{
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("MZ_TIMING");
        char *end = NULL;
        long x = e ? strtol(e, &end, 10) : 0;
        v = !e ? 0 : (end == e ? 1 : x < 0 ? 0 : x > 9 ? 9 : (int)x);
    }
    return v;
}

/* One caller at a time, and only one ever gets to exit(): the drivers' worker threads (OpenMP regions of mz_roast.c /
 * mz_multiz.c) can fail together; exit() runs the atexit handlers once, the later callers wait here for the process to
 * end. */
static pthread_mutex_t g_fatal_mu = PTHREAD_MUTEX_INITIALIZER;
__attribute__((noreturn)) void mz_fatalf(const char *fmt, ...)
{
    va_list ap;
    pthread_mutex_lock(&g_fatal_mu);                   /* (never released: the winner exits) */
    va_start(ap, fmt);
    mz_warm_wait();                                    /* no thread of ours inside the HIP runtime while the process exits */
    fflush(stdout);
    if (argv0) {
        const char *p = strrchr(argv0, '/');
        fprintf(stderr, "%s: ", p ? p + 1 : argv0);
    }
    vfprintf(stderr, fmt, ap);
    fputc('\n', stderr);
    va_end(ap);
    exit(1);
}

/* ------------------------------------------------------------------ context (struct and helpers: mz_ctx.h) */

mz_ctx g_dev[MZ_MAX_DEV];
int g_ndev;                                /* contexts in use (0 before mz_init) */
static unsigned long long g_score_sum;     /* checksum of the score tables last handed to the devices */
static int g_score_have;
pthread_mutex_t g_big = PTHREAD_MUTEX_INITIALIZER;   /* one host-path call at a time (the library state is process-wide) */

int mzi_dev_reserve(gbuf *b, size_t need)
{
    if (need <= b->cap) return 0;
    if (b->p) { HIPCK(hipFree(b->p)); b->p = NULL; b->cap = 0; }
    need = need + need / 4 + 4096;
    HIPCK(hipMalloc(&b->p, need));
    b->cap = need;
    return 0;
}
int mzi_host_reserve(gbuf *b, size_t need)
{
    if (need <= b->cap) return 0;
    if (b->p) { HIPCK(hipHostFree(b->p)); b->p = NULL; b->cap = 0; }
    need = need + need / 4 + 4096;
    HIPCK(hipHostMalloc(&b->p, need, hipHostMallocDefault));
    b->cap = need;
    return 0;
}

/* Events that only order streams -- or tell the host that results lie in pinned HOST memory, which the GPU writes through -- need no
 * system-scope release: hipEventRecord() otherwise makes the GPU write its L2 back to memory at every record, a dozen times per chunk
 * of a call beside DP kernels that dirty 0.5 TB/s of it (MZ_EVENT_SYSTEM=1: the runtime's default, for A/B measurements). */
unsigned mzi_event_flags(void)
{
    static int sys = -1;
    if (sys < 0) { const char *e = getenv("MZ_EVENT_SYSTEM"); sys = e && e[0] == '1'; }
    return hipEventDisableTiming | (sys ? 0u : hipEventReleaseToDevice);
}

void *mz_stream(void) { return G.ready ? (void *)G.stream : NULL; }

static int ctx_open(mz_ctx *X, int device)
{
    int i;
    memset(X, 0, sizeof *X);
    HIPCK(hipSetDevice(device));
    HIPCK(hipStreamCreateWithFlags(&X->stream, hipStreamNonBlocking));
    /* the helper streams (mz_dev_run_async) and the further chunk streams are created on first use: a stream costs
     * ~9 ms of start-up and a short run -- one yama() call, one chunk -- needs none of them */
    X->bstream[0] = X->stream;
    for (i = 0; i < MZ_SETS; ++i) {
        HIPCK(hipEventCreateWithFlags(&X->bdone[i], mzi_event_flags()));
        HIPCK(hipEventCreateWithFlags(&X->bplan[i], mzi_event_flags()));
        HIPCK(hipEventCreateWithFlags(&X->bdp[i], mzi_event_flags()));
        HIPCK(hipEventCreateWithFlags(&X->bprep[i], mzi_event_flags()));
        HIPCK(hipEventCreateWithFlags(&X->bcopy[i], mzi_event_flags()));
        HIPCK(hipEventCreateWithFlags(&X->pplan2[i], mzi_event_flags()));
    }
    for (i = 0; i < 5; ++i) HIPCK(hipEventCreate(&X->ev[i]));
    for (i = 0; i <= MZ_SLICES; ++i) HIPCK(hipEventCreateWithFlags(&X->evs[i], hipEventDisableTiming));
    X->device = device;
    X->copy_threads = MZ_COPY_THREADS;
    X->ready = 1;
    return 0;
}

static void ctx_close(mz_ctx *X)
{
    int i, s;
    if (!X->ready) return;
    mzi_workers_stop(X);
    hipSetDevice(X->device);
    hipStreamSynchronize(X->stream);
    for (s = 0; s < MZ_SETS; ++s) {
        gbuf *d[] = { &X->d_in[s], &X->d_exc[s], &X->d_cols[s], &X->d_band[s], &X->d_plan[s], &X->d_tb[s], &X->d_script[s], &X->d_prep[s], &X->d_res[s] };
        gbuf *h[] = { &X->h_in[s], &X->h_exc[s], &X->h_tot[s], &X->h_res[s] };
        if (X->bstream[s]) hipStreamSynchronize(X->bstream[s]);
        for (i = 0; i < (int)(sizeof d / sizeof d[0]); ++i) if (d[i]->p) { hipFree(d[i]->p); d[i]->p = NULL; d[i]->cap = 0; }
        for (i = 0; i < (int)(sizeof h / sizeof h[0]); ++i) if (h[i]->p) { hipHostFree(h[i]->p); h[i]->p = NULL; h[i]->cap = 0; }
        if (s >= 1 && X->bstream[s]) hipStreamDestroy(X->bstream[s]);
        if (X->btime_ready) for (i = 0; i < 6; ++i) hipEventDestroy(X->btime[s][i]);
        hipEventDestroy(X->bdone[s]);
        hipEventDestroy(X->bplan[s]);
        hipEventDestroy(X->bdp[s]);
        hipEventDestroy(X->bprep[s]);
        hipEventDestroy(X->bcopy[s]);
    }
    for (s = 0; s < MZ_SETS; ++s) {
        for (i = 0; i < MZ_PD_N; ++i) if (X->pd[s][i].p) { hipFree(X->pd[s][i].p); X->pd[s][i].p = NULL; X->pd[s][i].cap = 0; }
        for (i = 0; i < MZ_PH_N; ++i) if (X->ph[s][i].p) { hipHostFree(X->ph[s][i].p); X->ph[s][i].p = NULL; X->ph[s][i].cap = 0; }
        hipEventDestroy(X->pplan2[s]);
        if (X->ptime_ready) for (i = 0; i < 8; ++i) hipEventDestroy(X->ptime[s][i]);
    }
    for (s = 0; s < (X->nq_wide ? X->nq_wide : X->nq); ++s) {    /* the chunk pipelines' streams (mz_flow.c) */
        for (i = 0; i < X->qlane[s].n; ++i) hipEventDestroy((hipEvent_t)X->qlane[s].join[i]);
        hipEventDestroy((hipEvent_t)X->qlane[s].fork);
        X->qlane[s].n = 0;
    }
    X->nqall = X->lanes_made = X->nq_wide = 0;
    for (s = 0; s < MZ_QALL; ++s) if (X->qall[s]) { hipStreamSynchronize(X->qall[s]); hipStreamDestroy(X->qall[s]); X->qall[s] = NULL; }
    memset(X->qd, 0, sizeof X->qd); memset(X->qf, 0, sizeof X->qf); memset(X->qt, 0, sizeof X->qt); X->qc = NULL;
    X->nq = X->nf = X->nt = 0;
    for (i = 0; i < 5; ++i) hipEventDestroy(X->ev[i]);
    for (i = 0; i <= MZ_SLICES; ++i) hipEventDestroy(X->evs[i]);
    for (i = 0; i < MZ_WS_MAX; ++i) if (X->ws[i].used) { hipEventDestroy(X->ws[i].done); X->ws[i].used = 0; }
    if (X->stream2) { hipStreamSynchronize(X->stream2); hipStreamDestroy(X->stream2); }
    if (X->stream3) { hipStreamSynchronize(X->stream3); hipStreamDestroy(X->stream3); }
    for (i = 0; i < 4; ++i) if (X->stream_dp[i]) { hipStreamSynchronize(X->stream_dp[i]); hipStreamDestroy(X->stream_dp[i]); X->stream_dp[i] = NULL; }
    hipStreamDestroy(X->stream);
    X->ready = 0;
}

/* ngpu contexts on the given devices (devices == NULL: first, first+1, ...).  The first one is the primary
 * context.  Returns 0, or -1 with mz_last_error() set (nothing is left open then). */
static int init_devices(int ngpu, const int *devices, int first)
{
    int count = 0, i, j;
    /* the pipelined form keeps up to nine streams busy; the HIP runtime multiplexes streams onto 4 hardware queues
     * unless told otherwise, and streams that share a queue run one after the other (C5: 313 -> 381 GCUPS with 8).
     * Effective only if the runtime is not up yet; an application that starts it first sets the variable itself. */
    setenv("GPU_MAX_HW_QUEUES", "24", 0);
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return mzi_set_err("no HIP device available (this library has no CPU path)");
    if (ngpu < 1 || ngpu > MZ_MAX_DEV) return mzi_set_err("mz_init_multi: %d GPUs requested (1..%d supported)", ngpu, MZ_MAX_DEV);
    for (i = 0; i < ngpu; ++i) {
        const int d = devices ? devices[i] : first + i;
        if (d < 0 || d >= count) return mzi_set_err("HIP device %d out of range (%d present)", d, count);
        for (j = 0; j < i; ++j)       /* (MZ_ALLOW_DUP_DEVICES=1: several contexts on one GPU, to exercise the dealing on a one-GPU box) */
            if ((devices ? devices[j] : first + j) == d && !(getenv("MZ_ALLOW_DUP_DEVICES") && atoi(getenv("MZ_ALLOW_DUP_DEVICES"))))
                return mzi_set_err("mz_init_multi: device %d listed twice", d);
    }
    if (g_ndev) mz_finalize();
    for (i = 0; i < ngpu; ++i)
        if (ctx_open(&g_dev[i], devices ? devices[i] : first + i)) {
            for (j = 0; j <= i; ++j) ctx_close(&g_dev[j]);
            return -1;
        }
    g_ndev = ngpu;
    g_score_have = 0;
    HIPCK(hipSetDevice(g_dev[0].device));
    return 0;
}

int mz_init(int device)
{
    if (g_ndev == 1 && G.ready && G.device == device) return 0;
    return init_devices(1, NULL, device);
}

/* devices == NULL: first, first+1, ... with first = MZ_DEVICE (default 0) -- the same GPUs the environment-driven
 * start (ensure_init) opens, so an explicit call after one is a no-op instead of a tear-down under running batches */
static int env_first_device(void)
{
    const char *e = getenv("MZ_DEVICE");
    return e ? atoi(e) : 0;
}

int mz_init_multi(int ngpu, const int *devices)
{
    const int first = env_first_device();
    int i, same = g_ndev == ngpu;
    for (i = 0; same && i < ngpu; ++i) same = g_dev[i].ready && g_dev[i].device == (devices ? devices[i] : first + i);
    if (same) return 0;
    return init_devices(ngpu, devices, first);
}

int mz_device_count(void) { return g_ndev; }
int mz_abi_version(void) { return MZ_AMD_ABI; }

/* which physical GPU context `ctx` (0 .. mz_device_count()-1) runs on: "<PCI bus id> <device name>" -- what a multi-GPU
 * bench line prints per rank so that N ranks provably sat on N distinct devices */
int mz_device_identity(int ctx, char *buf, int len)
{
    char pci[64] = "?";
    hipDeviceProp_t prop;
    if (ctx < 0 || ctx >= g_ndev || !buf || len < 2) return -1;
    if (hipDeviceGetPCIBusId(pci, (int)sizeof pci, g_dev[ctx].device) != hipSuccess) snprintf(pci, sizeof pci, "device%d", g_dev[ctx].device);
    if (hipGetDeviceProperties(&prop, g_dev[ctx].device) != hipSuccess) prop.name[0] = 0;
    snprintf(buf, (size_t)len, "%s %s", pci, prop.name);
    return 0;
}

/* helper streams at normal priority (measured: lowest priority starves them behind the DP and costs 4 % of the
 * pipelined rate, highest gains nothing); MZ_HELPER_PRIO overrides for experiments */
int mzi_lazy_stream(hipStream_t *s)
{
    if (*s) return 0;
    HIPCK(hipStreamCreateWithPriority(s, hipStreamNonBlocking, getenv("MZ_HELPER_PRIO") ? atoi(getenv("MZ_HELPER_PRIO")) : 0));
    return 0;
}

void mz_finalize(void)
{
    int i;
    for (i = 0; i < g_ndev; ++i) {
        ctx_close(&g_dev[i]);
        mzk_release_device(g_dev[i].device);   /* the launchers' side streams and events on that GPU */
    }
    g_ndev = 0;
    g_score_have = 0;
    mzi_pool_stop();
    mzi_blocks_drop();
}

/* first use without mz_init(): MZ_DEVICE = first GPU (default 0), MZ_NGPU = how many (default 1) */
int mzi_ensure_init(void)
{
    const char *n;
    int ngpu;
    if (g_ndev) return 0;
    n = getenv("MZ_NGPU");
    ngpu = n ? atoi(n) : 1;
    if (ngpu <= 1) return mz_init(env_first_device());
    return init_devices(ngpu, NULL, env_first_device());
}

/* ------------------------------------------------------------------ start-up in the background
 * The first GPU call of a process pays for the HIP runtime, the context, the streams and the code object (0.3 s on
 * the boxes of this project; mz_start_up in the MZ_TIMING lines).  A driver that has files to read first calls
 * mz_warm_start(): a thread runs a one-pair batch, the caller parses its inputs meanwhile, and its first real batch
 * waits on the library's call lock for whatever of the start-up is left. */
__thread int mzi_warm_thread;
static pthread_t g_warm_th;
static int g_warm_on;
static void *warm_main(void *arg)
{
    static const unsigned char a[4] = { 'A', 'C', 'G', 'T' }, b[4] = { 'A', 'C', 'G', 'T' };
    static const int lb[5] = { 0, 0, 0, 0, 0 }, rb[5] = { 4, 4, 4, 4, 4 };
    mz_job job;
    mz_out out;
    (void)arg;
    mzi_warm_thread = 1;
    job.K = 1; job.L = 1; job.M = 4; job.N = 4; job.A = a; job.B = b; job.LB = lb; job.RB = rb;
    if (mz_yama_batch(1, &job, &out) >= 0) mz_free_outs(1, &out);
    return NULL;
}
void mz_warm_wait(void)
{
    /* one joiner: whoever takes the flag down (mz_fatalf() from a worker thread and the atexit handler may both come by) */
    if (!mzi_warm_thread && __atomic_exchange_n(&g_warm_on, 0, __ATOMIC_ACQ_REL)) pthread_join(g_warm_th, NULL);
}
void mz_warm_start(void)
{
    if (__atomic_load_n(&g_warm_on, __ATOMIC_ACQUIRE) || g_ndev) return;
    if (pthread_create(&g_warm_th, NULL, warm_main, NULL) != 0) return;      /* (no thread: the first call starts the GPU as ever) */
    __atomic_store_n(&g_warm_on, 1, __ATOMIC_RELEASE);
    atexit(mz_warm_wait);                                /* a run that never gets to a batch must not exit under the thread (the
                                                          * drivers also wait before they return from main) */
}

/* ------------------------------------------------------------------ scores */

static int g_no_fast;                      /* mz_enable_fast(0): exact kernels only */
static int g_no_row;                       /* mz_enable_row(0): no row-parallel kernel */

static int class_of(int ch)
{
    switch (ch) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    case '-': return 4;
    default:  return 5;
    }
}

/* ss as 128 row pointers or flat; returns 0 and fills the model when the tables have the
 * class structure the kernels rely on */
static int model_from_tables(int **rows, const int *flat, const int *g16, int ext, mz_score_model *m)
{
    static const unsigned char rep[6] = { 'A', 'C', 'G', 'T', '-', 'N' };
    int a, b, x;
#define SSAT(i, j) (rows ? rows[i][j] : flat[(i) * 128 + (j)])
    for (a = 0; a < 6; ++a)
        for (b = 0; b < 6; ++b)
            m->S6[a * 6 + b] = SSAT(rep[a], rep[b]);
    for (a = 0; a < 128; ++a)
        for (b = 0; b < 128; ++b)
            if (SSAT(a, b) != m->S6[class_of(a) * 6 + class_of(b)])
                return mzi_set_err("substitution table is not constant on the byte classes {A,C,G,T,-,other} at (%d,%d)", a, b);
#undef SSAT
    m->gap_open = g16[1];
    for (x = 0; x < 16; ++x) {
        int s = (x >> 3) & 1, t = (x >> 2) & 1, u = (x >> 1) & 1, v = x & 1;
        int want = (u != v && !(s == u && t == v)) ? m->gap_open : 0;
        if (g16[x] != want)
            return mzi_set_err("gap-open table entry %d is %d, expected %d (quasi-natural structure)", x, g16[x], want);
    }
    m->gap_extend = ext;
    /* gap_open = g1*g2, both small enough that 127*g fits an int16 dot-product operand; the fast
     * kernel needs it (MZ_NO_FAST=1 in the environment disables that kernel: exact kernel only) */
    m->g1 = m->g2 = 0;
    m->pack = 0;
    m->row = !g_no_row && !(getenv("MZ_NO_ROW") && atoi(getenv("MZ_NO_ROW")) != 0);
    if (!g_no_fast && (!getenv("MZ_NO_FAST") || atoi(getenv("MZ_NO_FAST")) == 0)) {
        if (m->gap_open == 0) { m->g1 = 1; m->g2 = 0; }
        else for (x = 1; x * x <= m->gap_open; ++x)       /* the most balanced factorisation */
            if (m->gap_open % x == 0 && m->gap_open / x <= 258) { m->g1 = m->gap_open / x; m->g2 = x; }
    }
    for (a = 0; a < 36; ++a)
        if (m->S6[a] < -258 || m->S6[a] > 258)      /* 127 rows * |score| must fit the int16 dot-product operand */
            return mzi_set_err("substitution score %d too large for the packed int16 row vector", m->S6[a]);
    if (m->gap_open < 0 || m->gap_open >= (1 << 15) || ext < 0 || ext >= (1 << 15))
        return mzi_set_err("gap penalties out of range (open %d, extend %d)", m->gap_open, ext);
    return 0;
}

/* the model goes to every context's __constant__ copy.  Kernels of earlier calls may still be reading it
 * (mz_dev_run_async, another caller's stream), so each device is drained first; score changes are rare. */
int g_hint_gen;                     /* bumped with every upload: hints of a plan made under another model are void */
int mz_hint_generation(void) { return g_hint_gen; }

/* the batch as the launchers may see it: hints (include/mz_amd.h) only when they were derived under the selection and
 * scores in force now -- a stale dp_hint would skip the kernels of modes the re-plan assigns */
static const mz_dev_batch *checked_hints(const mz_dev_batch *b, mz_dev_batch *tmp)
{
    if (!((b->dp_hint & ~MZ_DP_REQUESTS) | b->dp_grid | b->dp_rows | b->walk_hint) || b->hint_gen == g_hint_gen) return b;
    *tmp = *b;
    tmp->dp_hint = b->dp_hint & MZ_DP_REQUESTS;         /* (a request, not a hint) */
    tmp->dp_grid = tmp->dp_rows = tmp->walk_hint = 0;
    return tmp;
}

static int upload_everywhere(const mz_score_model *m)
{
    int i;
    g_hint_gen = g_hint_gen >= 0x7ffffff0 ? 1 : g_hint_gen + 1;
    for (i = 0; i < g_ndev; ++i) {
        HIPCK(hipSetDevice(g_dev[i].device));
        HIPCK(hipDeviceSynchronize());
        if (mzk_upload_scores(m, g_dev[i].stream)) return mzi_set_err("%s", mzk_last_error());
        g_dev[i].scores_ok = 1;
    }
    HIPCK(hipSetDevice(G.device));
    return 0;
}

int mz_set_scores(const int *ss_flat, const int *gop16, int ext)
{
    mz_score_model m;
    if (mzi_ensure_init()) return -1;
    if (model_from_tables(NULL, ss_flat, gop16, ext, &m)) return -1;
    if (upload_everywhere(&m)) return -1;
    g_score_have = 0;
    mz_scores_explicit = 1;                /* keep them until init_scores70/85() is called again */
    return 0;
}

static void scores_stale(void)
{
    int i;
    for (i = 0; i < MZ_MAX_DEV; ++i) g_dev[i].scores_ok = 0;
    g_score_have = 0;
}

/* Switch the fast DP kernel off (exact kernels only) or back on; used by the parity tests to
 * exercise both kernels on the same inputs.  Takes effect with the next call. */
void mz_enable_fast(int on)
{
    g_no_fast = !on;
    scores_stale();                        /* force the score model (which carries g1,g2) to be re-sent */
    mz_scores_explicit = 0;
}

void mz_enable_row(int on)
{
    g_no_row = !on;
    scores_stale();
    mz_scores_explicit = 0;
}

/* Checksum of what the kernels take from the caller's tables: the scores of the ten bytes that stand for the six
 * classes in both cases, gop[16], gap_extend.  A caller that edits its tables IN PLACE (same pointers) is
 * noticed too; a changed sum triggers the full structure check of model_from_tables() and a new upload. */
static unsigned long long score_checksum(void)
{
    static const unsigned char rep[10] = { 'A', 'C', 'G', 'T', 'a', 'c', 'g', 't', '-', 'N' };
    unsigned long long h = 1469598103934665603ULL;
    int a, b;
#define MIX(v) do { h ^= (unsigned long long)(unsigned)(v); h *= 1099511628211ULL; } while (0)
    for (a = 0; a < 10; ++a)
        for (b = 0; b < 10; ++b) MIX(ss[rep[a]][rep[b]]);
    for (a = 0; a < 16; ++a) MIX(gop[a]);
    MIX(gap_extend);
    MIX((g_no_fast << 1) | g_no_row);
#undef MIX
    return h;
}

/* hand the reference-style globals (ss, gop, gap_extend) to the devices if they changed */
int mzi_sync_scores(void)
{
    mz_score_model m;
    unsigned long long sum;
    int i, all = 1;
    for (i = 0; i < g_ndev; ++i) all &= g_dev[i].scores_ok;
    if (mz_scores_explicit && all) return 0;   /* tables given through mz_set_scores() */
    if (ss == NULL || gop == NULL)
        init_scores70();                   /* batch API default: HOXD70, as multiz.c:257 */
    sum = score_checksum();
    if (all && g_score_have && sum == g_score_sum) return 0;
    if (model_from_tables(ss, NULL, gop, gap_extend, &m)) return -1;
    if (upload_everywhere(&m)) return -1;
    g_score_sum = sum; g_score_have = 1;
    return 0;
}

/* ------------------------------------------------------------------ device-resident API */


size_t mz_dev_plan_bytes(int n)
{
    size_t s = 0, N = (size_t)(n > 0 ? n : 1);
    s += 5 * mzi_al256(4 * N);                 /* status, badrow, mode, edgeLo, edgeHi */
    s += 9 * mzi_al256(8 * N);                 /* cells, 4 sizes, 4 offsets */
    s += mzi_al256(8 * MZ_TOTALS);             /* totals */
    s += mzi_al256(4 * N) + mzi_al256(MZ_SCAN_AUX_BYTES(N));    /* packList, scanAux (8 sums and 128 list keys per 64 pairs; the plan's segment results of small batches) */
    s += mzi_al256(4 * N) + mzi_al256(12 * N);     /* om, final3 */
    return s;
}

void mz_dev_carve(mz_dev_batch *b, void *mem)
{
    char *p = (char *)mem;
    size_t N = (size_t)(b->n > 0 ? b->n : 1);
#define TAKE(field, type, bytes) do { b->field = (type)p; p += mzi_al256(bytes); } while (0)
    TAKE(status, int32_t *, 4 * N); TAKE(badrow, int32_t *, 4 * N); TAKE(mode, int32_t *, 4 * N);
    TAKE(edgeLo, int32_t *, 4 * N); TAKE(edgeHi, int32_t *, 4 * N);
    TAKE(cells, int64_t *, 8 * N);
    TAKE(szTb, int64_t *, 8 * N); TAKE(szScript, int64_t *, 8 * N); TAKE(szOut, int64_t *, 8 * N); TAKE(szPrep, int64_t *, 8 * N);
    TAKE(offTb, int64_t *, 8 * N); TAKE(offScript, int64_t *, 8 * N); TAKE(offOut, int64_t *, 8 * N); TAKE(offPrep, int64_t *, 8 * N);
    TAKE(totals, int64_t *, 8 * MZ_TOTALS);
    TAKE(packList, int32_t *, 4 * N); TAKE(scanAux, int64_t *, MZ_SCAN_AUX_BYTES(N));
    TAKE(om, int32_t *, 4 * N); TAKE(final3, int32_t *, 12 * N);
#undef TAKE
}

static void *pick_stream(void *s) { return s ? s : (void *)G.stream; }

int mz_dev_plan(const mz_dev_batch *b, void *stream)
{
    if (mzi_ensure_init() || mzi_sync_scores()) return -1;
    /* prep == NULL: a sizing pass (validity, modes, sizes, offsets, totals) before the workspaces exist */
    return (mzk_plan(b, pick_stream(stream)) || (b->prep && mzk_prep(b, pick_stream(stream)))) ? mzi_set_err("%s", mzk_last_error()) : 0;
}
int mz_dev_dp(const mz_dev_batch *b, void *stream)
{
    mz_dev_batch tmp;
    if (mzi_ensure_init() || mzi_sync_scores()) return -1;
    b = checked_hints(b, &tmp);
    return mzk_dp(b, pick_stream(stream)) ? mzi_set_err("%s", mzk_last_error()) : 0;
}
int mz_dev_walk(const mz_dev_batch *b, void *stream)
{
    mz_dev_batch tmp;
    if (mzi_ensure_init()) return -1;
    b = checked_hints(b, &tmp);
    return mzk_walk(b, pick_stream(stream), 0) ? mzi_set_err("%s", mzk_last_error()) : 0;
}
int mz_dev_emit(const mz_dev_batch *b, void *stream)
{
    if (mzi_ensure_init()) return -1;
    return mzk_emit(b, pick_stream(stream)) ? mzi_set_err("%s", mzk_last_error()) : 0;
}

int mz_dev_run(const mz_dev_batch *b, void *stream, float ms[4])
{
    hipStream_t s;
    mz_dev_batch tmp;
    int i;
    if (mzi_ensure_init() || mzi_sync_scores()) return -1;
    b = checked_hints(b, &tmp);
    s = (hipStream_t)pick_stream(stream);
    if (ms) {                              /* serial, one HIP event pair per phase */
        HIPCK(hipEventRecord(G.ev[0], s));
        if (mzk_plan(b, s) || mzk_prep(b, s)) return mzi_set_err("%s", mzk_last_error());
        HIPCK(hipEventRecord(G.ev[1], s));
        if (mzk_dp(b, s)) return mzi_set_err("%s", mzk_last_error());
        HIPCK(hipEventRecord(G.ev[2], s));
        if (mzk_walk(b, s, 0)) return mzi_set_err("%s", mzk_last_error());
        HIPCK(hipEventRecord(G.ev[3], s));
        if (mzk_emit(b, s)) return mzi_set_err("%s", mzk_last_error());
        HIPCK(hipEventRecord(G.ev[4], s));
        HIPCK(hipEventSynchronize(G.ev[4]));
        for (i = 0; i < 4; ++i) HIPCK(hipEventElapsedTime(&ms[i], G.ev[i], G.ev[i + 1]));
        return 0;
    }
    if (mzk_plan(b, s) || mzk_prep(b, s) || mzk_dp(b, s) || mzk_walk(b, s, 0) || mzk_emit(b, s)) return mzi_set_err("%s", mzk_last_error());
    return 0;
}

/* Pipelined form for a stream of batches.  The DPs of successive batches run back to back on `stream`; plan
 * and prep of a batch run on a third stream and walk + emit on a second, beside the DPs of the neighbouring
 * batches (the walk is a latency-bound pointer chase, plan and prep are short bandwidth bursts; the DP is
 * issue-bound).  Calls in flight at the same time MUST use different workspaces (tbw/script/out/prep and the
 * plan arrays).  The library keys a workspace by its tbw pointer and orders "reuse of workspace w" after
 * "walk/emit of the batch that used w last"; with three rotating workspaces the plan of batch k+1 does not
 * wait for the walk of batch k-1 and everything but the DP is off the critical path.  The plan does not run
 * on `stream`, so the batch's inputs must be complete when the call is made, or `ready_event` (a hipEvent_t
 * recorded after their producer) must be given.  mz_dev_wait() makes `stream` wait for everything issued. */

/* how many batches of n pairs should be in flight for the GPU to be busy: their DPs run side by side on that many
 * streams (mz_dev_run_async); a caller rotates one workspace more than that.  MZ_DP_STREAMS=<k> overrides. */
int mz_dev_pipeline_depth(int n)
{
    static int forced = -1;
    int d;
    if (forced < 0) { const char *e = getenv("MZ_DP_STREAMS"); forced = e ? atoi(e) : 0; }
    if (forced > 0) return forced > 5 ? 5 : forced;
    if (n <= 0) return 1;
    if (n > 16384) return 2;                         /* the next batch's DP fills the tail of this one (C2: 585 -> 603 GCUPS, C4 497 -> 508) */
    d = (10 * 1024 + n - 1) / n;                     /* waves wanted: twice five per SIMD, 1024 SIMDs -- a batch that just fills the GPU still
                                                      * tails off (C3, 5 000 pairs: 506 GCUPS two abreast, 515 three abreast, 459 four) */
    return d < 2 ? 2 : d > 5 ? 5 : d;
}

int mz_dev_run_async(const mz_dev_batch *b, void *stream, void *ready_event)
{
    hipStream_t s;
    mz_dev_batch tmp;
    int w, slot = -1, rc_dp;
    if (mzi_ensure_init() || mzi_sync_scores() || mzi_lazy_stream(&G.stream2) || mzi_lazy_stream(&G.stream3)) return -1;
    b = checked_hints(b, &tmp);
    s = (hipStream_t)pick_stream(stream);
    for (w = 0; w < MZ_WS_MAX; ++w) if (G.ws[w].used && G.ws[w].key == (const void *)b->tbw) slot = w;
    if (slot < 0) {
        for (w = 0; w < MZ_WS_MAX; ++w) if (!G.ws[w].used) { slot = w; break; }
        if (slot < 0) {                                   /* table full: forget the oldest entry once it is idle */
            slot = G.ws_victim;
            G.ws_victim = (G.ws_victim + 1) % MZ_WS_MAX;
            HIPCK(hipEventSynchronize(G.ws[slot].done));
        } else {
            HIPCK(hipEventCreateWithFlags(&G.ws[slot].done, hipEventDisableTiming));
        }
        G.ws[slot].key = (const void *)b->tbw;
        G.ws[slot].used = 1;
    } else {
        HIPCK(hipStreamWaitEvent(G.stream3, G.ws[slot].done, 0));      /* its previous batch has been walked and emitted */
    }
    if (ready_event) HIPCK(hipStreamWaitEvent(G.stream3, (hipEvent_t)ready_event, 0));
    if (mzk_plan(b, G.stream3) || mzk_prep(b, G.stream3)) return mzi_set_err("%s", mzk_last_error());
    HIPCK(hipEventRecord(G.evs[3], G.stream3));
    /* Small batches -- at most a few waves per SIMD: the 1 000 long pairs of C5, the 5 000 of C3 -- leave the GPU
     * half empty while their last waves finish, and a lone wave per SIMD is latency-bound throughout (section 4.1 of
     * 14.9 / 8.3 / 6.6 / 5.35 / 5.27 ms per C2 batch at 1..5 waves per SIMD).  Their DPs therefore go round
     * up to five streams (the caller's and four of the library's), so that the DPs of consecutive batches run side by
     * side, as many as it takes to put about five waves on a SIMD (mz_dev_pipeline_depth()); larger batches
     * go two abreast (the second fills the first's tail). */
    {
        hipStream_t sd = s;
        int depth = mz_dev_pipeline_depth(b->n);
        /* a large batch whose pairs went to several DP kernels runs those side by side (mzk_dp_range) and fills its own
         * tails; a second such batch beside it only adds waves that cannot become resident (the indel mix of 20 000
         * pairs: 390 GCUPS back to back, 360 two abreast) */
        if (b->n > 16384 && !getenv("MZ_DP_STREAMS") && (b->dp_hint & MZ_DP_KNOWN)) {
            const int kinds = b->dp_hint & (MZ_DP_ROW | MZ_DP_ROWBIG | MZ_DP_WAVEFRONT | MZ_DP_WIDE | MZ_DP_LAG);
            if (kinds & (kinds - 1)) depth = 1;
        }
        if (depth > 1) {
            const unsigned turn = G.dp_turn++ % (unsigned)depth;
            if (turn > 0) {
                if (mzi_lazy_stream(&G.stream_dp[turn - 1])) return -1;
                sd = G.stream_dp[turn - 1];
            }
        }
        HIPCK(hipStreamWaitEvent(sd, G.evs[3], 0));
        mzk_set_abreast(depth);
        rc_dp = mzk_dp(b, sd);
        mzk_set_abreast(1);
        if (rc_dp) return mzi_set_err("%s", mzk_last_error());
        HIPCK(hipEventRecord(G.evs[2], sd));
    }
    HIPCK(hipStreamWaitEvent(G.stream2, G.evs[2], 0));
    if (mzk_walk(b, G.stream2, 1) || mzk_emit(b, G.stream2)) return mzi_set_err("%s", mzk_last_error());
    HIPCK(hipEventRecord(G.ws[slot].done, G.stream2));
    return 0;
}

int mz_dev_wait(void *stream)
{
    hipStream_t s;
    int w;
    if (mzi_ensure_init()) return -1;
    s = (hipStream_t)pick_stream(stream);
    for (w = 0; w < MZ_WS_MAX; ++w)
        if (G.ws[w].used) HIPCK(hipStreamWaitEvent(s, G.ws[w].done, 0));
    return 0;
}

/* ------------------------------------------------------------------ yama(): a batch of one */

/* messages of reference mz_yama.c:59,64,68,70,275,308 */
__attribute__((noreturn)) void mz_fatal_status(const mz_job *j, const mz_out *o)
{
    int need = j->N < 10 ? j->N : 10, r = o->badrow;
    switch (o->status) {
    case MZ_E_TERMINATION:
        mz_fatalf("LB and RB not terminated properly: %d %d %d", j->LB[0], j->RB[j->M], j->N);
    case MZ_E_NARROW:
        mz_fatalf("RB[%d] - LB[%d] < %d, %d %d %d", r, r, need, j->RB[r], j->LB[r], j->N);
    case MZ_E_LB_MONO:  mz_fatalf("LB not monotonic");
    case MZ_E_RB_MONO:  mz_fatalf("RB not monotonic");
    case MZ_E_TRACEBACK: mz_fatalf("Error generating edit script.");
    case MZ_E_EMIT:     /* mz_yama.c:311-312 (the reference prints j twice) */
        mz_fatalf("new_align: i=%d, j=%d, m=%d, M=%d, N=%d, M_new=%d\n", o->score[0], o->score[1], o->score[1], j->M, j->N, o->OM);
    case MZ_E_ROWS:     mz_fatalf("yama(gfx950): K=%d, L=%d outside the supported 1..255 rows per block", j->K, j->L);
    case MZ_E_SHAPE:    mz_fatalf("yama(gfx950): empty block (M=%d, N=%d)", j->M, j->N);
    case MZ_E_RANGE:    mz_fatalf("yama(gfx950): M + N = %lld columns exceed the 2^30 steps of this build", (long long)j->M + j->N);
    case MZ_E_DEVICE:   mz_fatalf("yama(gfx950): not computed (device error or out of memory)");
    case MZ_E_SENTINEL: mz_fatalf("yama(gfx950): K=%d, L=%d, M=%d, N=%d: scores below the reference's MININT sentinel, where its traceback leaves the band (not reproduced by this build)", j->K, j->L, j->M, j->N);
    default:            mz_fatalf("yama(gfx950): device status %d", o->status);
    }
}

/* one job, contiguous columns in and out (the staged pre_yama of mz_preyama.c) */
void mz_py_run_one(mz_job *job, uchar **flat, int *om)
{
    mz_out o;
    int rc = mz_yama_batch(1, job, &o);
    if (rc < 0) mz_fatalf("yama(gfx950): %s", mz_last_error());
    if (o.status != MZ_OK) mz_fatal_status(job, &o);
    *flat = o.cols;
    *om = o.OM;
}

void yama(uchar **A, int K, int M, uchar **B, int L, int N, int *LB, int *RB, uchar ***OAL, int *OM)
{
    mz_job j;
    mz_out o;
    uchar *ca, *cb, **al;
    int i, rc;

    if (K < 1 || L < 1 || M < 1 || N < 1)
        mz_fatalf("yama(gfx950): empty block (K=%d, L=%d, M=%d, N=%d)", K, L, M, N);
    /* (a limit of this build -- the reference sizes dashes[MAX(K,L)] at run time, mz_yama.c:73-75 -- said here, before the columns are
     *  gathered and before the GPU is touched: the plan would refuse the pair from K and L alone) */
    if (K > 255 || L > 255) mz_fatalf("yama(gfx950): K=%d, L=%d outside the supported 1..255 rows per block", K, L);
    /* the reference only promises 1-based column pointers (mz_yama.h:6-9); gather them */
    ca = (uchar *)malloc((size_t)K * M);
    cb = (uchar *)malloc((size_t)L * N);
    if (!ca || !cb) mz_fatalf("Ran out of memory trying to allocate %lu.", (unsigned long)((size_t)K * M + (size_t)L * N));
    for (i = 1; i <= M; ++i) memcpy(ca + (size_t)(i - 1) * K, A[i], (size_t)K);
    for (i = 1; i <= N; ++i) memcpy(cb + (size_t)(i - 1) * L, B[i], (size_t)L);
    j.K = K; j.L = L; j.M = M; j.N = N; j.A = ca; j.B = cb; j.LB = LB; j.RB = RB;

    rc = mz_yama_batch(1, &j, &o);
    if (rc < 0) mz_fatalf("yama(gfx950): %s", mz_last_error());
    if (o.status != MZ_OK) mz_fatal_status(&j, &o);
    free(ca); free(cb);

    /* two malloc blocks, freed by the caller as free(OAL[1]); free(OAL+1); (mz_yama.h:17-18) */
    al = (uchar **)malloc((size_t)(o.OM > 0 ? o.OM : 1) * sizeof(uchar *));
    if (!al) mz_fatalf("Ran out of memory trying to allocate %lu.", (unsigned long)(o.OM * sizeof(uchar *)));
    al -= 1;
    al[1] = o.cols;
    for (i = 2; i <= o.OM; ++i) al[i] = al[i - 1] + (K + L);
    *OAL = al;
    *OM = o.OM;
}
