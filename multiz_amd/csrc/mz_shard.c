/* mz_shard.c -- include/mz_shard.h: a list of independent yama() jobs (reference mz_yama.h:22; the tree drivers' independent
 * multiz runs, /root/reference/tba.c:177-255, auto_mz.c:101,113) that exists on one rank of a process-per-GPU job, dealt out,
 * aligned on every rank's GPU and put together again on the root -- in C, over a table of transport functions (mz_comm):
 * RCCL's grouped ncclSend / ncclRecv over xGMI, a mailbox inside the process, or the caller's own.
 *
 * The exchange (every "group" is one group_start .. group_end of the transport -- with RCCL ONE
 * ncclGroupStart / ncclGroupEnd, all peers' links at once):
 *   scatter   root: deal by cost (mzi_deal_snake: the rule mz_yama_batch() deals a batch over the GPUs of one process with),
 *             pack every share as a link image on the host threads (mz_link_pack), group 1: a header per peer (the image's
 *             descriptor, the share's pairs), group 2: image, exception block, the pairs' indices in the list;
 *   align     every rank: mz_link_plan + mz_link_finish on the image where it landed in HBM;
 *   gather    group 3: a header per peer (its result image's size), group 4: the result images; the root assembles the merged columns
 *             of every share from ITS OWN A and B (mz_link_assemble) and puts every pair's result at its place in the list.
 * Nothing here touches the oracle; the CPU tests put their own result images where `align` would (mz_shard_set_result).
 */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "mz_ctx.h"
#include "../../include/mz_shard.h"

static int64_t g_sent, g_received;         /* (ranks may be threads of one process -- the loop-back tests: atomic adds) */
#define SENT(b) __atomic_fetch_add(&g_sent, (int64_t)(b), __ATOMIC_RELAXED)
#define RECEIVED(b) __atomic_fetch_add(&g_received, (int64_t)(b), __ATOMIC_RELAXED)
void mz_shard_traffic(int64_t *sent, int64_t *received)
{
    if (sent) *sent = __atomic_load_n(&g_sent, __ATOMIC_RELAXED);
    if (received) *received = __atomic_load_n(&g_received, __ATOMIC_RELAXED);
}

/* ------------------------------------------------------------------------------------------------ host buffers (loop-back, custom) */

static void *host_alloc(mz_comm *c, size_t bytes) { void *p = NULL; (void)c; return posix_memalign(&p, 256, bytes ? mzi_al256(bytes) : 256) ? NULL : p; }
static void host_release(mz_comm *c, void *p) { (void)c; free(p); }
static int host_put(mz_comm *c, void *buf, const void *host, size_t bytes) { (void)c; memcpy(buf, host, bytes); return 0; }
static int host_get(mz_comm *c, void *host, const void *buf, size_t bytes) { (void)c; memcpy(host, buf, bytes); return 0; }
static int no_group(mz_comm *c) { (void)c; return 0; }

/* ------------------------------------------------------------------------------------------------ loop-back: every rank in this process */

typedef struct lb_msg { struct lb_msg *next; int src, dst; size_t bytes; } lb_msg;      /* (the bytes follow) */
typedef struct lb_world { pthread_mutex_t mu; pthread_cond_t cv; lb_msg *head, *tail; int refs; } lb_world;

static int lb_send(mz_comm *c, const void *buf, size_t bytes, int peer)
{
    lb_world *w = (lb_world *)c->self;
    lb_msg *m = (lb_msg *)malloc(sizeof *m + bytes);
    if (peer < 0 || peer >= c->size) return mzi_set_err("loop-back send: no rank %d", peer);
    if (!m) return mzi_set_err("out of memory");
    m->next = NULL; m->src = c->rank; m->dst = peer; m->bytes = bytes;
    memcpy(m + 1, buf, bytes);
    pthread_mutex_lock(&w->mu);
    if (w->tail) w->tail->next = m; else w->head = m;
    w->tail = m;
    pthread_cond_broadcast(&w->cv);
    pthread_mutex_unlock(&w->mu);
    return 0;
}
static int lb_recv(mz_comm *c, void *buf, size_t bytes, int peer)
{
    lb_world *w = (lb_world *)c->self;
    struct timespec until;
    clock_gettime(CLOCK_MONOTONIC, &until);                  /* (ten seconds of the monotonic clock, whatever the wall clock does meanwhile) */
    until.tv_sec += 10;
    pthread_mutex_lock(&w->mu);
    for (;;) {
        lb_msg **pp, *m;
        for (pp = &w->head; *pp && !((*pp)->src == peer && (*pp)->dst == c->rank); pp = &(*pp)->next) ;
        if ((m = *pp) != NULL) {
            const size_t got = m->bytes;
            if (got == bytes) memcpy(buf, m + 1, bytes);
            *pp = m->next;
            if (w->tail == m) { lb_msg *t = w->head; while (t && t->next) t = t->next; w->tail = t; }
            pthread_mutex_unlock(&w->mu);
            free(m);
            return got == bytes ? 0 : mzi_set_err("loop-back recv on rank %d: %zu bytes from rank %d where %zu were expected", c->rank, got, peer, bytes);
        }
        if (pthread_cond_timedwait(&w->cv, &w->mu, &until) != 0) {
            pthread_mutex_unlock(&w->mu);
            return mzi_set_err("loop-back recv on rank %d: nothing from rank %d", c->rank, peer);
        }
    }
}
static void lb_destroy(mz_comm *c)
{
    lb_world *w = (lb_world *)c->self;
    int last;
    pthread_mutex_lock(&w->mu);
    last = --w->refs == 0;
    pthread_mutex_unlock(&w->mu);
    if (last) {
        while (w->head) { lb_msg *m = w->head; w->head = m->next; free(m); }
        pthread_mutex_destroy(&w->mu); pthread_cond_destroy(&w->cv);
        free(w);
    }
    free(c);
}
int mz_comm_loopback(int size, mz_comm **ranks)
{
    lb_world *w;
    int r;
    if (size < 1 || !ranks) return mzi_set_err("mz_comm_loopback: bad arguments");
    w = (lb_world *)calloc(1, sizeof *w);
    if (!w) return mzi_set_err("out of memory");
    pthread_mutex_init(&w->mu, NULL);
    { pthread_condattr_t a; pthread_condattr_init(&a); pthread_condattr_setclock(&a, CLOCK_MONOTONIC); pthread_cond_init(&w->cv, &a); pthread_condattr_destroy(&a); }   /* (lb_recv's deadline) */
    for (r = 0; r < size; ++r) {
        mz_comm *c = (mz_comm *)calloc(1, sizeof *c);
        if (!c) { while (r-- > 0) free(ranks[r]); free(w); return mzi_set_err("out of memory"); }
        c->rank = r; c->size = size; c->device_buffers = 0; c->self = w;
        c->group_start = no_group; c->group_end = no_group; c->send = lb_send; c->recv = lb_recv;
        c->alloc = host_alloc; c->release = host_release; c->put = host_put; c->get = host_get; c->destroy = lb_destroy;
        ranks[r] = c;
    }
    w->refs = size;
    return 0;
}

/* ------------------------------------------------------------------------------------------------ the caller's transport */

typedef struct cu_self {
    void *user;
    int (*send)(void *, const void *, size_t, int);
    int (*recv)(void *, void *, size_t, int);
    int (*gs)(void *);
    int (*ge)(void *);
} cu_self;
static int cu_send(mz_comm *c, const void *b, size_t n, int p) { cu_self *s = (cu_self *)c->self; return s->send(s->user, b, n, p) ? mzi_set_err("the caller's send to rank %d failed", p) : 0; }
static int cu_recv(mz_comm *c, void *b, size_t n, int p) { cu_self *s = (cu_self *)c->self; return s->recv(s->user, b, n, p) ? mzi_set_err("the caller's recv from rank %d failed", p) : 0; }
static int cu_gs(mz_comm *c) { cu_self *s = (cu_self *)c->self; return s->gs && s->gs(s->user) ? mzi_set_err("the caller's group start failed") : 0; }
static int cu_ge(mz_comm *c) { cu_self *s = (cu_self *)c->self; return s->ge && s->ge(s->user) ? mzi_set_err("the caller's group end failed") : 0; }
static void cu_destroy(mz_comm *c) { free(c->self); free(c); }
int mz_comm_custom(int rank, int size, void *user, int (*send)(void *, const void *, size_t, int), int (*recv)(void *, void *, size_t, int),
                   int (*group_start)(void *), int (*group_end)(void *), mz_comm **comm)
{
    mz_comm *c;
    cu_self *s;
    if (!comm || !send || !recv || size < 1 || rank < 0 || rank >= size) return mzi_set_err("mz_comm_custom: bad arguments");
    c = (mz_comm *)calloc(1, sizeof *c); s = (cu_self *)calloc(1, sizeof *s);
    if (!c || !s) { free(c); free(s); return mzi_set_err("out of memory"); }
    s->user = user; s->send = send; s->recv = recv; s->gs = group_start; s->ge = group_end;
    c->rank = rank; c->size = size; c->device_buffers = 0; c->self = s;
    c->group_start = cu_gs; c->group_end = cu_ge; c->send = cu_send; c->recv = cu_recv;
    c->alloc = host_alloc; c->release = host_release; c->put = host_put; c->get = host_get; c->destroy = cu_destroy;
    *comm = c;
    return 0;
}

/* ------------------------------------------------------------------------------------------------ RCCL
 * librccl.so is loaded on first use (the library does not link it: a process that never shards needs none of it).  Prototypes:
 * /opt/rocm/include/rccl/rccl.h -- ncclUniqueId is 128 bytes and travels BY VALUE into ncclCommInitRank; ncclUint8 = 1. */
typedef struct { char internal[128]; } rccl_id;
static struct {
    void *lib;
    int (*GetUniqueId)(rccl_id *);
    int (*CommInitRank)(void **, int, rccl_id, int);
    int (*CommDestroy)(void *);
    int (*GroupStart)(void);
    int (*GroupEnd)(void);
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t);
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t);
    const char *(*GetErrorString)(int);
} R;
static pthread_mutex_t g_rccl_mu = PTHREAD_MUTEX_INITIALIZER;

static int rccl_load(void)
{
    char why[300] = "";
    int ok;
    pthread_mutex_lock(&g_rccl_mu);
    if (!R.lib) {
        const char *e;
        void *h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) { e = dlerror(); snprintf(why, sizeof why, "%s", e ? e : "dlopen failed"); }      /* (dlerror() clears what it returns: asked once) */
        if (h) {
#define SYM(field, name) *(void **)&R.field = dlsym(h, name)
            SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
            SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv");
            SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
            if (R.GetUniqueId && R.CommInitRank && R.CommDestroy && R.GroupStart && R.GroupEnd && R.Send && R.Recv) R.lib = h;
            else { snprintf(why, sizeof why, "it lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclGroupStart / ncclGroupEnd / ncclSend / ncclRecv"); memset(&R, 0, sizeof R); dlclose(h); }
        }
    }
    ok = R.lib != NULL;
    pthread_mutex_unlock(&g_rccl_mu);
    return ok ? 0 : mzi_set_err("librccl.so cannot be loaded: %s", why);
}
#define NCK(call) do { const int r_ = (call); if (r_ != 0) return mzi_set_err("%s failed: %s", #call, R.GetErrorString ? R.GetErrorString(r_) : "?"); } while (0)

typedef struct rc_self { void *comm; hipStream_t stream; int device; } rc_self;
static int rc_gs(mz_comm *c) { (void)c; NCK(R.GroupStart()); return 0; }
static int rc_ge(mz_comm *c) { rc_self *s = (rc_self *)c->self; NCK(R.GroupEnd()); HIPCK(hipStreamSynchronize(s->stream)); return 0; }
static int rc_send(mz_comm *c, const void *b, size_t n, int p) { rc_self *s = (rc_self *)c->self; NCK(R.Send(b, n, 1 /* ncclUint8 */, p, s->comm, s->stream)); return 0; }
static int rc_recv(mz_comm *c, void *b, size_t n, int p) { rc_self *s = (rc_self *)c->self; NCK(R.Recv(b, n, 1, p, s->comm, s->stream)); return 0; }
static void *rc_alloc(mz_comm *c, size_t bytes) { void *p = NULL; (void)c; return hipMalloc(&p, bytes ? mzi_al256(bytes) : 256) == hipSuccess ? p : NULL; }
static void rc_release(mz_comm *c, void *p) { (void)c; if (p) hipFree(p); }
static int rc_put(mz_comm *c, void *buf, const void *host, size_t bytes) { (void)c; if (bytes) HIPCK(hipMemcpy(buf, host, bytes, hipMemcpyHostToDevice)); return 0; }
static int rc_get(mz_comm *c, void *host, const void *buf, size_t bytes) { (void)c; if (bytes) HIPCK(hipMemcpy(host, buf, bytes, hipMemcpyDeviceToHost)); return 0; }
static void rc_destroy(mz_comm *c)
{
    rc_self *s = (rc_self *)c->self;
    if (s->comm) R.CommDestroy(s->comm);
    if (s->stream) hipStreamDestroy(s->stream);
    free(s); free(c);
}
int mz_comm_rccl_unique_id(void *id128)
{
    if (!id128) return mzi_set_err("mz_comm_rccl_unique_id: NULL");
    if (rccl_load()) return -1;
    NCK(R.GetUniqueId((rccl_id *)id128));
    return 0;
}
int mz_comm_rccl_create(const void *id128, int rank, int size, mz_comm **comm)
{
    mz_comm *c;
    rc_self *s;
    rccl_id id;
    if (!id128 || !comm || size < 1 || rank < 0 || rank >= size) return mzi_set_err("mz_comm_rccl_create: bad arguments");
    if (rccl_load()) return -1;
    pthread_mutex_lock(&g_big);
    if (mzi_ensure_init()) { pthread_mutex_unlock(&g_big); return -1; }
    pthread_mutex_unlock(&g_big);
    c = (mz_comm *)calloc(1, sizeof *c); s = (rc_self *)calloc(1, sizeof *s);
    if (!c || !s) { free(c); free(s); return mzi_set_err("out of memory"); }
    c->self = s; c->rank = rank; c->size = size; c->device_buffers = 1;
    c->group_start = rc_gs; c->group_end = rc_ge; c->send = rc_send; c->recv = rc_recv;
    c->alloc = rc_alloc; c->release = rc_release; c->put = rc_put; c->get = rc_get; c->destroy = rc_destroy;
    s->device = G.device;
    memcpy(&id, id128, sizeof id);
    if (hipSetDevice(s->device) != hipSuccess || hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) { rc_destroy(c); return mzi_set_err("mz_comm_rccl_create: no stream on GPU %d", s->device); }
    { const int r_ = R.CommInitRank(&s->comm, size, id, rank); if (r_ != 0) { s->comm = NULL; rc_destroy(c); return mzi_set_err("ncclCommInitRank failed: %s", R.GetErrorString ? R.GetErrorString(r_) : "?"); } }
    *comm = c;
    return 0;
}

void mz_comm_free(mz_comm *comm) { if (comm) comm->destroy(comm); }

int mz_comm_echo(mz_comm *c, size_t bytes)
{
    unsigned char *h0, *h1;
    void *b0, *b1;
    size_t i;
    int rc = -1;
    if (!c || !bytes) return mzi_set_err("mz_comm_echo: bad arguments");
    h0 = (unsigned char *)malloc(bytes); h1 = (unsigned char *)calloc(bytes, 1);
    b0 = c->alloc(c, bytes); b1 = c->alloc(c, bytes);
    if (!h0 || !h1 || !b0 || !b1) { mzi_set_err("out of memory"); goto out; }
    for (i = 0; i < bytes; ++i) h0[i] = (unsigned char)(i * 131u + (i >> 8) * 7u + 3u);
    if (c->put(c, b0, h0, bytes) || c->put(c, b1, h1, bytes)) goto out;
    if (c->group_start(c) || c->send(c, b0, bytes, c->rank) || c->recv(c, b1, bytes, c->rank) || c->group_end(c)) goto out;
    if (c->get(c, h1, b1, bytes)) goto out;
    rc = memcmp(h0, h1, bytes) == 0 ? 0 : mzi_set_err("mz_comm_echo: what came back differs");
    SENT(bytes); RECEIVED(bytes);
out:
    if (b0) c->release(c, b0);
    if (b1) c->release(c, b1);
    free(h0); free(h1);
    return rc;
}

/* ------------------------------------------------------------------------------------------------ scatter / align / gather */

struct mz_shard {
    mz_link_desc desc;
    int64_t n_share, n_total, res_bytes;
    int is_root, size, on_device;
    int64_t *index;                        /* host: where in the list every pair of this share belongs */
    void *image, *exc, *result;            /* transport buffers (device memory with RCCL) */
    void *h_image, *h_exc;                 /* host copies, on demand (mz_shard_host_image) */
    /* the root: every rank's share as a job list of its own, and the way back */
    int *cnt, *start, *order;              /* rank r's pairs: order[start[r] .. start[r] + cnt[r]) = their places in the list */
    mz_job *jbuf;
};

const mz_link_desc *mz_shard_desc(const mz_shard *sh) { return sh ? &sh->desc : NULL; }
const int64_t *mz_shard_index(const mz_shard *sh) { return sh ? sh->index : NULL; }

void mz_shard_free(mz_comm *c, mz_shard *sh)
{
    if (!sh) return;
    if (c) { if (sh->image) c->release(c, sh->image); if (sh->exc) c->release(c, sh->exc); if (sh->result) c->release(c, sh->result); }
    free(sh->index); free(sh->h_image); free(sh->h_exc); free(sh->cnt); free(sh->start); free(sh->order); free(sh->jbuf);
    free(sh);
}

/* what a pair costs the GPU, roughly (as mz_yama_batch() weighs it when it deals a batch over the GPUs of one process) */
static double shard_weight(const mz_job *j)
{
    if (j->K < 1 || j->L < 1 || j->M < 1 || j->N < 1 || !j->LB || !j->RB) return 1.0;
    return ((double)j->M + 1.0) * (double)(j->RB[j->M / 2] - j->LB[j->M / 2] + 1) + 64.0 * (j->K + j->L);
}

#define HDR_I64 10                         /* a scatter header: the share's mz_link_desc (8 x int64), its pairs, spare */

int mz_shard_scatter(mz_comm *c, int root, int n, const mz_job *jobs, mz_shard **out)
{
    mz_shard *sh;
    int r, p, rc = -1;
    int64_t hdr[HDR_I64];
    void *hbuf = NULL;
    /* the root's per-peer send buffers */
    void **s_hdr = NULL, **s_img = NULL, **s_exc = NULL, **s_idx = NULL;
    mz_link_desc *descs = NULL;

    if (out) *out = NULL;
    if (!c || !out || root < 0 || root >= c->size || (c->rank == root && (n < 0 || (n && !jobs)))) return mzi_set_err("mz_shard_scatter: bad arguments");
    sh = (mz_shard *)calloc(1, sizeof *sh);
    if (!sh) return mzi_set_err("out of memory");
    sh->is_root = c->rank == root; sh->size = c->size; sh->on_device = c->device_buffers;

    if (sh->is_root) {
        const int W = c->size;
        int *owner = (int *)malloc(((size_t)n + 1) * sizeof *owner), *where = (int *)malloc(((size_t)n + 1) * sizeof *where);
        double *wt = (double *)calloc((size_t)n + 1, sizeof *wt);
        sh->cnt = (int *)calloc((size_t)W, sizeof *sh->cnt); sh->start = (int *)calloc((size_t)W, sizeof *sh->start);
        sh->order = (int *)malloc(((size_t)n + 1) * sizeof *sh->order); sh->jbuf = (mz_job *)malloc(((size_t)n + 1) * sizeof *sh->jbuf);
        s_hdr = (void **)calloc((size_t)W, sizeof *s_hdr); s_img = (void **)calloc((size_t)W, sizeof *s_img);
        s_exc = (void **)calloc((size_t)W, sizeof *s_exc); s_idx = (void **)calloc((size_t)W, sizeof *s_idx);
        descs = (mz_link_desc *)calloc((size_t)W, sizeof *descs);
        if (!owner || !where || !wt || !sh->cnt || !sh->start || !sh->order || !sh->jbuf || !s_hdr || !s_img || !s_exc || !s_idx || !descs) {
            free(owner); free(where); free(wt); mzi_set_err("out of memory"); goto done;
        }
        for (p = 0; p < n; ++p) wt[p] = shard_weight(&jobs[p]);
        if (mzi_deal_snake(n, wt, W, owner, where, sh->cnt, sh->start)) { free(owner); free(where); free(wt); goto done; }
        for (p = 0; p < n; ++p) { sh->jbuf[where[p]] = jobs[p]; sh->order[where[p]] = p; }
        free(owner); free(where); free(wt);
        sh->n_total = n;
        /* every share as a link image (packed on the library's host threads), into a buffer the transport can move */
        for (r = 0; r < W; ++r) {
            void *img = NULL, *exc = NULL;
            int64_t *idx;
            int i;
            if (mz_link_pack(sh->cnt[r], sh->jbuf + sh->start[r], &descs[r], &img, &exc)) goto done;
            idx = (int64_t *)malloc(((size_t)sh->cnt[r] + 1) * sizeof *idx);
            if (!idx) { mz_link_free(img); mz_link_free(exc); mzi_set_err("out of memory"); goto done; }
            for (i = 0; i < sh->cnt[r]; ++i) idx[i] = sh->order[sh->start[r] + i];
            memcpy(hdr, &descs[r], 8 * sizeof(int64_t)); hdr[8] = sh->cnt[r]; hdr[9] = 0;
            if (r == root) {
                sh->desc = descs[r]; sh->n_share = sh->cnt[r]; sh->index = idx; idx = NULL;
                sh->image = c->alloc(c, (size_t)descs[r].image_bytes); sh->exc = c->alloc(c, (size_t)descs[r].exc_bytes);
                if (!sh->image || !sh->exc || c->put(c, sh->image, img, (size_t)descs[r].image_bytes) || (descs[r].exc_bytes && c->put(c, sh->exc, exc, (size_t)descs[r].exc_bytes)))
                    { mz_link_free(img); mz_link_free(exc); if (!sh->image || !sh->exc) mzi_set_err("out of memory for the share's image"); goto done; }
            } else {
                s_hdr[r] = c->alloc(c, sizeof hdr); s_img[r] = c->alloc(c, (size_t)descs[r].image_bytes);
                s_exc[r] = c->alloc(c, (size_t)descs[r].exc_bytes); s_idx[r] = c->alloc(c, 8 * (size_t)sh->cnt[r]);
                if (!s_hdr[r] || !s_img[r] || !s_exc[r] || !s_idx[r]) { mz_link_free(img); mz_link_free(exc); free(idx); mzi_set_err("out of memory for rank %d's image", r); goto done; }
                if (c->put(c, s_hdr[r], hdr, sizeof hdr) || c->put(c, s_img[r], img, (size_t)descs[r].image_bytes) ||
                    (descs[r].exc_bytes && c->put(c, s_exc[r], exc, (size_t)descs[r].exc_bytes)) || (sh->cnt[r] && c->put(c, s_idx[r], idx, 8 * (size_t)sh->cnt[r])))
                    { mz_link_free(img); mz_link_free(exc); free(idx); goto done; }
            }
            mz_link_free(img); mz_link_free(exc); free(idx);
        }
        if (c->group_start(c)) goto done;                    /* group 1: the headers */
        for (r = 0; r < W; ++r) if (r != root && c->send(c, s_hdr[r], sizeof hdr, r)) goto done;
        if (c->group_end(c)) goto done;
        if (c->group_start(c)) goto done;                    /* group 2: image, exceptions, indices -- every peer's at once */
        for (r = 0; r < W; ++r) {
            if (r == root) continue;
            if (c->send(c, s_img[r], (size_t)descs[r].image_bytes, r)) goto done;
            if (descs[r].exc_bytes && c->send(c, s_exc[r], (size_t)descs[r].exc_bytes, r)) goto done;
            if (sh->cnt[r] && c->send(c, s_idx[r], 8 * (size_t)sh->cnt[r], r)) goto done;
            SENT((int64_t)sizeof hdr + descs[r].image_bytes + descs[r].exc_bytes + 8 * (int64_t)sh->cnt[r]);
        }
        if (c->group_end(c)) goto done;
    } else {
        void *ibuf;
        hbuf = c->alloc(c, sizeof hdr);
        if (!hbuf) { mzi_set_err("out of memory"); goto done; }
        if (c->group_start(c) || c->recv(c, hbuf, sizeof hdr, root) || c->group_end(c) || c->get(c, hdr, hbuf, sizeof hdr)) goto done;
        memcpy(&sh->desc, hdr, 8 * sizeof(int64_t));
        sh->n_share = hdr[8];
        if (sh->n_share < 0 || sh->n_share != sh->desc.n || sh->desc.image_bytes < 0 || sh->desc.exc_bytes < 0) { mzi_set_err("mz_shard_scatter: rank %d got a header that is none", c->rank); goto done; }
        sh->image = c->alloc(c, (size_t)sh->desc.image_bytes); sh->exc = c->alloc(c, (size_t)sh->desc.exc_bytes);
        ibuf = c->alloc(c, 8 * (size_t)sh->n_share);
        sh->index = (int64_t *)malloc(((size_t)sh->n_share + 1) * sizeof *sh->index);
        if (!sh->image || !sh->exc || !ibuf || !sh->index) { if (ibuf) c->release(c, ibuf); mzi_set_err("out of memory for the share's image"); goto done; }
        if (c->group_start(c) || c->recv(c, sh->image, (size_t)sh->desc.image_bytes, root) ||
            (sh->desc.exc_bytes && c->recv(c, sh->exc, (size_t)sh->desc.exc_bytes, root)) ||
            (sh->n_share && c->recv(c, ibuf, 8 * (size_t)sh->n_share, root)) || c->group_end(c) ||
            (sh->n_share && c->get(c, sh->index, ibuf, 8 * (size_t)sh->n_share))) { c->release(c, ibuf); goto done; }
        c->release(c, ibuf);
        RECEIVED((int64_t)sizeof hdr + sh->desc.image_bytes + sh->desc.exc_bytes + 8 * sh->n_share);
    }
    rc = 0;
done:
    if (hbuf) c->release(c, hbuf);
    if (s_hdr) for (r = 0; r < c->size; ++r) { if (s_hdr[r]) c->release(c, s_hdr[r]); if (s_img && s_img[r]) c->release(c, s_img[r]); if (s_exc && s_exc[r]) c->release(c, s_exc[r]); if (s_idx && s_idx[r]) c->release(c, s_idx[r]); }
    free(s_hdr); free(s_img); free(s_exc); free(s_idx); free(descs);
    if (rc) { mz_shard_free(c, sh); return -1; }
    *out = sh;
    return 0;
}

int mz_shard_host_image(mz_comm *c, mz_shard *sh, const void **image, const void **exc)
{
    if (!c || !sh || !image || !exc) return mzi_set_err("mz_shard_host_image: bad arguments");
    free(sh->h_image); free(sh->h_exc);
    sh->h_image = malloc((size_t)sh->desc.image_bytes + 1); sh->h_exc = malloc((size_t)sh->desc.exc_bytes + 1);
    if (!sh->h_image || !sh->h_exc) return mzi_set_err("out of memory");
    if (c->get(c, sh->h_image, sh->image, (size_t)sh->desc.image_bytes) || (sh->desc.exc_bytes && c->get(c, sh->h_exc, sh->exc, (size_t)sh->desc.exc_bytes))) return -1;
    *image = sh->h_image; *exc = sh->h_exc;
    return 0;
}

int mz_shard_set_result(mz_comm *c, mz_shard *sh, const void *result, int64_t bytes)
{
    if (!c || !sh || !result || bytes < 0) return mzi_set_err("mz_shard_set_result: bad arguments");
    if (bytes < mzi_result_image_min((int)sh->n_share))
        return mzi_set_err("mz_shard_set_result: %lld bytes cannot be the result image of %lld pairs (header and records alone: %lld)", (long long)bytes, (long long)sh->n_share, (long long)mzi_result_image_min((int)sh->n_share));
    if (sh->result) c->release(c, sh->result);
    sh->result = c->alloc(c, (size_t)bytes);
    if (!sh->result) return mzi_set_err("out of memory");
    sh->res_bytes = bytes;
    return c->put(c, sh->result, result, (size_t)bytes);
}

/* band cells and pairs without a result of this share, from its result image's records */
int mz_shard_totals(mz_comm *c, const mz_shard *sh, int64_t *cells, int64_t *failed)
{
    mz_res_rec *rec;
    int64_t p, n;
    if (!c || !sh || !cells || !failed) return mzi_set_err("mz_shard_totals: bad arguments");
    *cells = *failed = 0;
    n = sh->n_share;
    if (!n) return 0;
    if (!sh->result || sh->res_bytes < 64 + (int64_t)sizeof *rec * n) return mzi_set_err("mz_shard_totals: the share has no result image");
    rec = (mz_res_rec *)malloc(sizeof *rec * (size_t)n);
    if (!rec) return mzi_set_err("out of memory");
    if (c->get(c, rec, (const char *)sh->result + 64, sizeof *rec * (size_t)n)) { free(rec); return -1; }
    for (p = 0; p < n; ++p) { *cells += rec[p].cells; *failed += rec[p].status != MZ_OK; }
    free(rec);
    return 0;
}

/* (needs the transport's buffers in device memory: the image is aligned where it lies) */
int mz_shard_align(mz_shard *sh)
{
    void *res = NULL;
    if (!sh) return mzi_set_err("mz_shard_align: NULL");
    if (!sh->on_device) return mzi_set_err("mz_shard_align: this share arrived in host memory (a transport without device buffers): move its image (mz_shard_host_image, mz_link_plan / mz_link_finish) and hand the result over with mz_shard_set_result()");
    if (mz_link_plan(&sh->desc, sh->image, sh->exc, NULL)) return -1;
    if (hipMalloc(&res, (size_t)(sh->desc.res_bytes > 0 ? sh->desc.res_bytes : 256)) != hipSuccess) return mzi_set_err("out of device memory for the result image (%lld bytes)", (long long)sh->desc.res_bytes);
    if (mz_link_finish(&sh->desc, res, NULL)) { hipFree(res); return -1; }
    if (hipStreamSynchronize((hipStream_t)mz_stream()) != hipSuccess) { hipFree(res); return mzi_set_err("mz_shard_align: the stream did not come back"); }
    if (sh->result) hipFree(sh->result);
    sh->result = res; sh->res_bytes = sh->desc.res_bytes;
    return 0;
}

int mz_shard_gather(mz_comm *c, int root, mz_shard *sh, const mz_job *jobs, mz_out *outs)
{
    int64_t hdr[2];
    int r, p, rc = -1, failed = 0;
    if (!c || !sh || root < 0 || root >= c->size || (c->rank == root) != sh->is_root) return mzi_set_err("mz_shard_gather: bad arguments");
    if (!sh->result && sh->n_share) return mzi_set_err("mz_shard_gather: rank %d's share has not been aligned", c->rank);
    if (!sh->is_root) {
        void *hb = c->alloc(c, sizeof hdr);
        hdr[0] = sh->n_share; hdr[1] = sh->res_bytes;
        if (!hb) return mzi_set_err("out of memory");
        if (c->put(c, hb, hdr, sizeof hdr) || c->group_start(c) || c->send(c, hb, sizeof hdr, root) || c->group_end(c)) { c->release(c, hb); return -1; }
        c->release(c, hb);
        if (sh->res_bytes && (c->group_start(c) || c->send(c, sh->result, (size_t)sh->res_bytes, root) || c->group_end(c))) return -1;
        SENT((int64_t)sizeof hdr + sh->res_bytes);
        return 0;
    }
    if (sh->n_total && (!jobs || !outs)) return mzi_set_err("mz_shard_gather: the root needs the jobs it scattered and outs");
    {
        const int W = c->size;
        void **hb = (void **)calloc((size_t)W, sizeof *hb), **rb = (void **)calloc((size_t)W, sizeof *rb);
        int64_t *rbytes = (int64_t *)calloc((size_t)W, sizeof *rbytes);
        mz_out *obuf = (mz_out *)malloc(((size_t)sh->n_total + 1) * sizeof *obuf);
        if (!hb || !rb || !rbytes || !obuf) { mzi_set_err("out of memory"); goto rdone; }
        for (p = 0; p < sh->n_total; ++p) { outs[p].status = MZ_E_DEVICE; outs[p].badrow = -1; outs[p].OM = 0; outs[p].cols = NULL; outs[p].block = NULL; outs[p].score[0] = outs[p].score[1] = outs[p].score[2] = 0; }
        for (r = 0; r < W; ++r) if (r != root && !(hb[r] = c->alloc(c, sizeof hdr))) { mzi_set_err("out of memory"); goto rdone; }
        if (c->group_start(c)) goto rdone;                   /* group 3: how large every result image is */
        for (r = 0; r < W; ++r) if (r != root && c->recv(c, hb[r], sizeof hdr, r)) goto rdone;
        if (c->group_end(c)) goto rdone;
        for (r = 0; r < W; ++r) {
            if (r == root) { rbytes[r] = sh->res_bytes; continue; }
            if (c->get(c, hdr, hb[r], sizeof hdr)) goto rdone;
            if (hdr[0] != sh->cnt[r] || hdr[1] < 0) { mzi_set_err("mz_shard_gather: rank %d answers for %lld pairs, it was given %d", r, (long long)hdr[0], sh->cnt[r]); goto rdone; }
            /* (an image too short for its own records -- none at all, say -- is not read: the records would come from whatever lies behind it) */
            if (hdr[1] < mzi_result_image_min(sh->cnt[r])) { mzi_set_err("mz_shard_gather: rank %d sends a result image of %lld bytes for %d pairs (their records alone take %lld)", r, (long long)hdr[1], sh->cnt[r], (long long)mzi_result_image_min(sh->cnt[r])); goto rdone; }
            rbytes[r] = hdr[1];
            if (rbytes[r] && !(rb[r] = c->alloc(c, (size_t)rbytes[r]))) { mzi_set_err("out of memory for rank %d's results", r); goto rdone; }
        }
        if (c->group_start(c)) goto rdone;                   /* group 4: the result images, every peer's at once */
        for (r = 0; r < W; ++r) if (r != root && rbytes[r] && c->recv(c, rb[r], (size_t)rbytes[r], r)) goto rdone;
        if (c->group_end(c)) goto rdone;
        for (r = 0; r < W; ++r) if (r != root) RECEIVED((int64_t)sizeof hdr + rbytes[r]);
        /* every share's merged columns from the root's own A and B; every pair's result to its place in the list */
        for (r = 0; r < W; ++r) {
            void *host;
            int f;
            if (!sh->cnt[r]) continue;
            host = malloc((size_t)rbytes[r] + 1);
            if (!host) { mzi_set_err("out of memory"); goto rdone; }
            if (c->get(c, host, r == root ? sh->result : rb[r], (size_t)rbytes[r])) { free(host); goto rdone; }
            f = mz_link_assemble(sh->cnt[r], sh->jbuf + sh->start[r], host, rbytes[r], obuf + sh->start[r]);
            free(host);
            if (f < 0) { char why[400]; snprintf(why, sizeof why, "%s", mz_last_error()); mzi_set_err("rank %d's result image: %s", r, why); goto rdone; }
            failed += f;
            for (p = 0; p < sh->cnt[r]; ++p) outs[sh->order[sh->start[r] + p]] = obuf[sh->start[r] + p];
        }
        rc = failed;
rdone:
        if (hb) for (r = 0; r < W; ++r) { if (hb[r]) c->release(c, hb[r]); if (rb && rb[r]) c->release(c, rb[r]); }
        free(hb); free(rb); free(rbytes); free(obuf);
    }
    return rc;
}

/* ------------------------------------------------------------------------------------------------ the exchange in chunks (round 6)
 * mz_shard_scatter / _align / _gather above are three phases one after the other: the root packs EVERY share, then everybody
 * aligns, then the root assembles everything -- scatter 0.14 s + align 0.08 s + gather 0.16 s for 60 000 pairs on two ranks, nothing
 * hidden behind anything.  mz_shard_run() deals the list over ranks x chunks (the same snake, so every chunk of every
 * rank is the same mix) and moves chunk by chunk; in step t, side by side:
 *
 *      the root's host threads   pack chunk t+1 of every rank's share (mz_link_pack), assemble chunk t-3's results (mz_link_assemble)
 *      the transport             group 1: header of chunk t down, header of chunk t-2's result image up
 *                                group 2: chunk t's image + exception block down, chunk t-2's result image up -- every peer's at once
 *      every rank's GPU          aligns chunk t-1 of its share (mz_link_plan + mz_link_finish), the root's included
 *
 * so a run of C chunks takes C + 4 steps of max(pack + assemble, move, align) each instead of the sum of the three.  A rank's chunk
 * may be empty (fewer pairs than ranks x chunks): its header says so and nothing else travels.  A rank whose GPU fails a chunk says
 * so in that chunk's result header (-1 bytes): the root leaves those pairs MZ_E_DEVICE and the exchange runs to its end -- nobody
 * is left waiting in a receive.  ("overlap chunked scatter with compute anyway"; the independent jobs are the tree
 * drivers' per-node multiz runs, /root/reference/tba.c:177-255, auto_mz.c:101,113.) */
#define RUN_SLOTS 4                        /* a chunk's buffers live from step c (received) to step c+3 (assembled) */
#define RUN_HDR 12                         /* int64: the chunk's mz_link_desc (8), its pairs, chunks of the run, this chunk, a magic */
#define RUN_MAGIC 0x6d7a5f72756eLL

typedef struct rslot { void *p; size_t cap; } rslot;
static int rslot_need(mz_comm *c, rslot *s, size_t need)
{
    if (s->p && s->cap >= need) return 0;
    if (s->p) c->release(c, s->p);
    s->cap = need + need / 4 + 256;                          /* (chunks of one run are about the same size: the first few allocate) */
    s->p = c->alloc(c, s->cap);
    if (!s->p) { s->cap = 0; return mzi_set_err("out of memory for a transport buffer of %zu bytes", need); }
    return 0;
}
static void rslot_drop(mz_comm *c, rslot *s) { if (s && s->p) c->release(c, s->p); if (s) { s->p = NULL; s->cap = 0; } }

typedef struct shard_run {
    mz_comm *c;
    int root, is_root, W, C;
    /* this rank's chunk in slot c & 3: descriptor, pairs, image / exceptions / result where the transport can move them */
    mz_link_desc desc[RUN_SLOTS];
    int64_t cn[RUN_SLOTS], res_bytes[RUN_SLOTS];            /* res_bytes: -1 = the chunk failed here */
    rslot img[RUN_SLOTS], exc[RUN_SLOTS], res[RUN_SLOTS], hdr_s, hdr_r;
    /* the GPU side of a transport that moves host memory (the image goes up, the result comes down) */
    void *d_img, *d_exc, *d_res; size_t d_img_cap, d_exc_cap, d_res_cap;
    hipEvent_t ev0, ev1; int ev_ready, aligning;
    /* the caller's align (CPU tests: the oracle), and what it hands back */
    mz_shard_align_fn align; void *user;
    void *cb_result; int64_t cb_bytes; int cb_set;
    /* the root */
    const mz_job *jobs; mz_out *outs; int n;
    int *cnt, *start, *order;              /* bin b = r * C + c: the pairs order[start[b] .. start[b] + cnt[b]) of the list */
    mz_job *jbuf; mz_out *obuf;
    rslot *s_img, *s_exc, *s_hdr, *r_hdr;  /* per peer: send buffers [2 * W] (chunk parity), header buffers [W] */
    rslot *r_res;                          /* per peer and slot [RUN_SLOTS * W]: its result images */
    mz_link_desc *pdesc;                   /* [RUN_SLOTS * W]: the packed chunks' descriptors */
    int64_t *rbytes;                       /* [RUN_SLOTS * W]: bytes of the result images (-1: failed there) */
    int failed;
    /* the root's host thread of a step */
    pthread_t th; int th_on, w_step, w_rc; char w_err[500];
    int64_t my_pairs, my_cells, my_failed;
    double t_pack, t_comm, t_align, t_asm;
    int rc; char err[500];                 /* the first error of this rank (the exchange still runs to its end) */
} shard_run;

static void run_note(shard_run *S) { if (!S->rc) { S->rc = -1; snprintf(S->err, sizeof S->err, "%s", mz_last_error()); } }

int mz_shard_chunk_result(void *handle, const void *result, int64_t bytes)
{
    shard_run *S = (shard_run *)handle;
    if (!S || !result || bytes < 0) return mzi_set_err("mz_shard_chunk_result: bad arguments");
    free(S->cb_result);
    S->cb_result = malloc((size_t)bytes + 1);
    if (!S->cb_result) return mzi_set_err("out of memory");
    memcpy(S->cb_result, result, (size_t)bytes);
    S->cb_bytes = bytes; S->cb_set = 1;
    return 0;
}

/* ---- the root's host side of step t: assemble chunk t-3, pack chunk t+1 */
static int run_assemble(shard_run *S, int ch)
{
    mz_comm *c = S->c;
    const int W = S->W, C = S->C, k = ch & (RUN_SLOTS - 1);
    int r, p;
    const double t0 = mzi_now_s();
    for (r = 0; r < W; ++r) {
        const int b = r * C + ch, cnt = S->cnt[b];
        const int64_t bytes = r == S->root ? S->res_bytes[k] : S->rbytes[k * W + r];
        void *host;
        int f;
        if (!cnt) continue;
        if (bytes < 0) { S->failed += cnt; continue; }      /* (that rank's GPU failed the chunk: its pairs stay MZ_E_DEVICE) */
        host = malloc((size_t)bytes + 1);
        if (!host) return mzi_set_err("out of memory");
        if (c->get(c, host, r == S->root ? S->res[k].p : S->r_res[k * W + r].p, (size_t)bytes)) { free(host); return -1; }
        f = mz_link_assemble(cnt, S->jbuf + S->start[b], host, bytes, S->obuf + S->start[b]);
        free(host);
        if (f < 0) { char why[400]; snprintf(why, sizeof why, "%s", mz_last_error()); return mzi_set_err("rank %d's result image of chunk %d: %s", r, ch, why); }
        S->failed += f;
        for (p = 0; p < cnt; ++p) S->outs[S->order[S->start[b] + p]] = S->obuf[S->start[b] + p];
    }
    S->t_asm += mzi_now_s() - t0;
    return 0;
}

static int run_pack(shard_run *S, int ch)
{
    mz_comm *c = S->c;
    const int W = S->W, C = S->C, k = ch & (RUN_SLOTS - 1), par = ch & 1;
    int r;
    const double t0 = mzi_now_s();
    for (r = 0; r < W; ++r) {
        const int b = r * C + ch;
        mz_link_desc *d = &S->pdesc[k * W + r];
        void *img = NULL, *exc = NULL;
        rslot *si = r == S->root ? &S->img[k] : &S->s_img[par * W + r], *se = r == S->root ? &S->exc[k] : &S->s_exc[par * W + r];
        int bad;
        if (mz_link_pack(S->cnt[b], S->jbuf + S->start[b], d, &img, &exc)) return -1;
        bad = rslot_need(c, si, (size_t)d->image_bytes) || rslot_need(c, se, (size_t)d->exc_bytes) ||
              (d->image_bytes && c->put(c, si->p, img, (size_t)d->image_bytes)) || (d->exc_bytes && c->put(c, se->p, exc, (size_t)d->exc_bytes));
        mz_link_free(img); mz_link_free(exc);
        if (bad) return -1;
        if (r == S->root) { S->desc[k] = *d; S->cn[k] = S->cnt[b]; }
    }
    S->t_pack += mzi_now_s() - t0;
    return 0;
}

static void *run_host_thread(void *arg)
{
    shard_run *S = (shard_run *)arg;
    const int t = S->w_step;
    S->w_rc = 0;
    if (S->c->device_buffers) hipSetDevice(G.device);       /* (put / get are device copies with such a transport) */
    if (t - 3 >= 0 && t - 3 < S->C && run_assemble(S, t - 3)) S->w_rc = -1;
    if (!S->w_rc && t + 1 < S->C && run_pack(S, t + 1)) S->w_rc = -1;
    if (S->w_rc) snprintf(S->w_err, sizeof S->w_err, "%s", mz_last_error());
    return NULL;
}

/* ---- this rank's GPU: chunk ch of its share (slot k), started here, waited for in run_align_wait() */
static int run_align_start(shard_run *S, int ch)
{
    mz_comm *c = S->c;
    const int k = ch & (RUN_SLOTS - 1);
    mz_link_desc *d = &S->desc[k];
    S->aligning = 0; S->res_bytes[k] = 0;
    if (!S->cn[k]) return 0;
    S->my_pairs += S->cn[k];
    if (S->align) {                                          /* the caller's: on host copies, waited for on the spot */
        void *hi = malloc((size_t)d->image_bytes + 1), *he = malloc((size_t)d->exc_bytes + 1);
        const double t0 = mzi_now_s();
        int bad = !hi || !he || c->get(c, hi, S->img[k].p, (size_t)d->image_bytes) || (d->exc_bytes && c->get(c, he, S->exc[k].p, (size_t)d->exc_bytes));
        if (!hi || !he) mzi_set_err("out of memory");
        S->cb_set = 0;
        if (!bad && S->align(S->user, ch, d, hi, he, S)) { bad = 1; mzi_set_err("the caller's align of chunk %d failed", ch); }
        if (!bad && !S->cb_set) { bad = 1; mzi_set_err("the caller's align of chunk %d handed no result image over (mz_shard_chunk_result)", ch); }
        if (!bad && S->cb_bytes < mzi_result_image_min((int)S->cn[k])) { bad = 1; mzi_set_err("the caller's align of chunk %d: %lld bytes cannot be the result image of %lld pairs", ch, (long long)S->cb_bytes, (long long)S->cn[k]); }
        if (!bad) bad = rslot_need(c, &S->res[k], (size_t)S->cb_bytes) || c->put(c, S->res[k].p, S->cb_result, (size_t)S->cb_bytes);
        free(hi); free(he);
        S->t_align += mzi_now_s() - t0;
        if (bad) { S->res_bytes[k] = -1; return -1; }
        S->res_bytes[k] = S->cb_bytes;
        return 0;
    }
    {
        const void *di = S->img[k].p, *de = S->exc[k].p;
        void *dr;
        if (!S->ev_ready) { if (hipEventCreate(&S->ev0) != hipSuccess || hipEventCreate(&S->ev1) != hipSuccess) { S->res_bytes[k] = -1; return mzi_set_err("no HIP events"); } S->ev_ready = 1; }
        if (!c->device_buffers) {                            /* the image arrived in host memory: up it goes */
            if (S->d_img_cap < (size_t)d->image_bytes + 256) { if (S->d_img) hipFree(S->d_img); S->d_img = NULL; S->d_img_cap = (size_t)d->image_bytes * 5 / 4 + 512; if (hipMalloc(&S->d_img, S->d_img_cap) != hipSuccess) { S->d_img_cap = 0; S->res_bytes[k] = -1; return mzi_set_err("out of device memory for a chunk's image"); } }
            if (S->d_exc_cap < (size_t)d->exc_bytes + 256) { if (S->d_exc) hipFree(S->d_exc); S->d_exc = NULL; S->d_exc_cap = (size_t)d->exc_bytes * 5 / 4 + 512; if (hipMalloc(&S->d_exc, S->d_exc_cap) != hipSuccess) { S->d_exc_cap = 0; S->res_bytes[k] = -1; return mzi_set_err("out of device memory for a chunk's exception block"); } }
            if (hipMemcpy(S->d_img, di, (size_t)d->image_bytes, hipMemcpyHostToDevice) != hipSuccess ||
                (d->exc_bytes && hipMemcpy(S->d_exc, de, (size_t)d->exc_bytes, hipMemcpyHostToDevice) != hipSuccess)) { S->res_bytes[k] = -1; return mzi_set_err("mz_shard_run: a chunk's image did not reach the GPU"); }
            di = S->d_img; de = S->d_exc;
        }
        if (mz_link_plan(d, di, de, NULL)) { S->res_bytes[k] = -1; return -1; }
        if (c->device_buffers) { if (rslot_need(c, &S->res[k], (size_t)d->res_bytes)) { S->res_bytes[k] = -1; return -1; } dr = S->res[k].p; }
        else {
            if (S->d_res_cap < (size_t)d->res_bytes + 256) { if (S->d_res) hipFree(S->d_res); S->d_res = NULL; S->d_res_cap = (size_t)d->res_bytes * 5 / 4 + 512; if (hipMalloc(&S->d_res, S->d_res_cap) != hipSuccess) { S->d_res_cap = 0; S->res_bytes[k] = -1; return mzi_set_err("out of device memory for a chunk's result image"); } }
            dr = S->d_res;
        }
        hipEventRecord(S->ev0, (hipStream_t)mz_stream());
        if (mz_link_finish(d, dr, NULL)) { S->res_bytes[k] = -1; return -1; }
        hipEventRecord(S->ev1, (hipStream_t)mz_stream());
        S->aligning = 1;
    }
    return 0;
}

static int run_align_wait(shard_run *S, int ch)
{
    mz_comm *c = S->c;
    const int k = ch & (RUN_SLOTS - 1);
    mz_link_desc *d = &S->desc[k];
    float ms = 0;
    if (!S->aligning) return 0;
    S->aligning = 0;
    if (hipStreamSynchronize((hipStream_t)mz_stream()) != hipSuccess) { S->res_bytes[k] = -1; return mzi_set_err("mz_shard_run: the GPU did not come back from chunk %d", ch); }
    if (hipEventElapsedTime(&ms, S->ev0, S->ev1) == hipSuccess) S->t_align += 1e-3 * ms;
    if (!c->device_buffers) {                                /* ... and the result comes down, to where the transport can move it */
        void *h = malloc((size_t)d->res_bytes + 1);
        int bad = !h || hipMemcpy(h, S->d_res, (size_t)d->res_bytes, hipMemcpyDeviceToHost) != hipSuccess ||
                  rslot_need(c, &S->res[k], (size_t)d->res_bytes) || c->put(c, S->res[k].p, h, (size_t)d->res_bytes);
        free(h);
        if (bad) { S->res_bytes[k] = -1; return mzi_set_err("mz_shard_run: chunk %d's result image did not come back from the GPU", ch); }
    }
    S->res_bytes[k] = d->res_bytes;
    return 0;
}

/* cells and failures of this rank's own chunk (from its records, wherever they lie) */
static void run_count(shard_run *S, int ch)
{
    mz_comm *c = S->c;
    const int k = ch & (RUN_SLOTS - 1);
    const int64_t n = S->cn[k];
    mz_res_rec *rec;
    int64_t p;
    if (!n) return;
    if (S->res_bytes[k] < 0) { S->my_failed += n; return; }
    rec = (mz_res_rec *)malloc(sizeof *rec * (size_t)n);
    if (!rec) return;
    if (c->get(c, rec, (const char *)S->res[k].p + 64, sizeof *rec * (size_t)n) == 0)
        for (p = 0; p < n; ++p) { S->my_cells += rec[p].cells; S->my_failed += rec[p].status != MZ_OK; }
    free(rec);
}

/* ---- the transport's two groups of step t */
static int run_comm(shard_run *S, int t)
{
    mz_comm *c = S->c;
    const int W = S->W, root = S->root;
    int64_t hdr[RUN_HDR], rh[2];
    int r;
    const double t0 = mzi_now_s();
    if (S->is_root) {
        const int C = S->C, down = t < C, up = t - 2 >= 0 && t - 2 < C, kd = t & (RUN_SLOTS - 1), ku = (t - 2) & (RUN_SLOTS - 1), par = t & 1;
        if (W == 1 || (!down && !up)) return 0;
        for (r = 0; r < W && down; ++r) {                   /* group 1: chunk t's header down, chunk t-2's result header up */
            if (r == root) continue;
            memcpy(hdr, &S->pdesc[kd * W + r], 8 * sizeof(int64_t));
            hdr[8] = S->cnt[r * C + t]; hdr[9] = C; hdr[10] = t; hdr[11] = RUN_MAGIC;
            if (rslot_need(c, &S->s_hdr[r], sizeof hdr) || c->put(c, S->s_hdr[r].p, hdr, sizeof hdr)) return -1;
        }
        for (r = 0; r < W && up; ++r) if (r != root && rslot_need(c, &S->r_hdr[r], sizeof rh)) return -1;
        if (c->group_start(c)) return -1;
        for (r = 0; r < W; ++r) {
            if (r == root) continue;
            if (down && c->send(c, S->s_hdr[r].p, sizeof hdr, r)) return -1;
            if (up && c->recv(c, S->r_hdr[r].p, sizeof rh, r)) return -1;
        }
        if (c->group_end(c)) return -1;
        for (r = 0; r < W && up; ++r) {
            const int64_t cnt = S->cnt[r * C + (t - 2)];
            if (r == root) continue;
            if (c->get(c, rh, S->r_hdr[r].p, sizeof rh)) return -1;
            if (rh[0] != cnt) return mzi_set_err("mz_shard_run: rank %d answers for %lld pairs of chunk %d, it was given %lld", r, (long long)rh[0], t - 2, (long long)cnt);
            if (rh[1] >= 0 && cnt && rh[1] < mzi_result_image_min((int)cnt)) return mzi_set_err("mz_shard_run: rank %d sends a result image of %lld bytes for the %lld pairs of chunk %d (their records alone take %lld)", r, (long long)rh[1], (long long)cnt, t - 2, (long long)mzi_result_image_min((int)cnt));
            S->rbytes[ku * W + r] = cnt ? rh[1] : 0;
            if (rh[1] > 0 && rslot_need(c, &S->r_res[ku * W + r], (size_t)rh[1])) return -1;
            RECEIVED(sizeof rh);
        }
        if (c->group_start(c)) return -1;                    /* group 2: the images down, the result images up -- every peer's at once */
        for (r = 0; r < W; ++r) {
            const mz_link_desc *d = &S->pdesc[kd * W + r];
            if (r == root) continue;
            if (down) {
                if (d->image_bytes && c->send(c, S->s_img[par * W + r].p, (size_t)d->image_bytes, r)) return -1;
                if (d->exc_bytes && c->send(c, S->s_exc[par * W + r].p, (size_t)d->exc_bytes, r)) return -1;
                SENT((int64_t)sizeof hdr + d->image_bytes + d->exc_bytes);
            }
            if (up && S->rbytes[ku * W + r] > 0) { if (c->recv(c, S->r_res[ku * W + r].p, (size_t)S->rbytes[ku * W + r], r)) return -1; RECEIVED(S->rbytes[ku * W + r]); }
        }
        if (c->group_end(c)) return -1;
    } else {
        const int down = S->C == 0 || t < S->C, up = S->C > 0 && t - 2 >= 0 && t - 2 < S->C, kd = t & (RUN_SLOTS - 1), ku = (t - 2) & (RUN_SLOTS - 1);
        if (!down && !up) return 0;
        if ((down && rslot_need(c, &S->hdr_r, sizeof hdr)) || (up && rslot_need(c, &S->hdr_s, sizeof rh))) return -1;
        if (up) { rh[0] = S->cn[ku]; rh[1] = S->cn[ku] ? S->res_bytes[ku] : 0; if (c->put(c, S->hdr_s.p, rh, sizeof rh)) return -1; }
        if (c->group_start(c)) return -1;
        if (down && c->recv(c, S->hdr_r.p, sizeof hdr, root)) return -1;
        if (up && c->send(c, S->hdr_s.p, sizeof rh, root)) return -1;
        if (c->group_end(c)) return -1;
        if (up) SENT(sizeof rh);
        if (down) {
            if (c->get(c, hdr, S->hdr_r.p, sizeof hdr)) return -1;
            memcpy(&S->desc[kd], hdr, 8 * sizeof(int64_t));
            S->cn[kd] = hdr[8];
            if (hdr[11] != RUN_MAGIC || hdr[10] != t || hdr[9] < 1 || (S->C && hdr[9] != S->C) || S->cn[kd] < 0 || S->cn[kd] != S->desc[kd].n ||
                S->desc[kd].image_bytes < 0 || S->desc[kd].exc_bytes < 0)
                return mzi_set_err("mz_shard_run: rank %d got a header of chunk %d that is none", c->rank, t);
            S->C = (int)hdr[9];
            if (rslot_need(c, &S->img[kd], (size_t)S->desc[kd].image_bytes) || rslot_need(c, &S->exc[kd], (size_t)S->desc[kd].exc_bytes)) return -1;
        }
        if (c->group_start(c)) return -1;
        if (down) {
            if (S->desc[kd].image_bytes && c->recv(c, S->img[kd].p, (size_t)S->desc[kd].image_bytes, root)) return -1;
            if (S->desc[kd].exc_bytes && c->recv(c, S->exc[kd].p, (size_t)S->desc[kd].exc_bytes, root)) return -1;
            RECEIVED((int64_t)sizeof hdr + S->desc[kd].image_bytes + S->desc[kd].exc_bytes);
        }
        if (up && S->cn[ku] && S->res_bytes[ku] > 0) { if (c->send(c, S->res[ku].p, (size_t)S->res_bytes[ku], root)) return -1; SENT(S->res_bytes[ku]); }
        if (c->group_end(c)) return -1;
    }
    S->t_comm += mzi_now_s() - t0;
    return 0;
}

static int run_chunks_for(int share)
{
    static int v = -1;
    int C;
    if (v < 0) { const char *e = getenv("MZ_SHARD_CHUNKS"); v = e && atoi(e) > 0 ? atoi(e) : 0; }
    if (v) return v > 64 ? 64 : v;
    C = (share + 2047) / 4096;                               /* chunks of about 4 Ki pairs: a full round of DP waves and more */
    return C < 1 ? 1 : C > 32 ? 32 : C;
}

int mz_shard_run(mz_comm *c, int root, int n, const mz_job *jobs, mz_out *outs, int chunks, mz_shard_align_fn align, void *user, mz_shard_times *times)
{
    shard_run *S;
    int t, r, p, rc;
    const double t_start = mzi_now_s();
    if (times) memset(times, 0, sizeof *times);
    if (!c || root < 0 || root >= c->size || chunks < 0 || chunks > 64 || (c->rank == root && (n < 0 || (n && (!jobs || !outs)))))
        return mzi_set_err("mz_shard_run: bad arguments");
    S = (shard_run *)calloc(1, sizeof *S);
    if (!S) return mzi_set_err("out of memory");
    S->c = c; S->root = root; S->is_root = c->rank == root; S->W = c->size; S->align = align; S->user = user;
    if (!align && !c->device_buffers) {                      /* (the GPU is this rank's own business: started before the first chunk is waited for) */
        pthread_mutex_lock(&g_big);
        if (mzi_ensure_init()) { pthread_mutex_unlock(&g_big); free(S); return -1; }
        pthread_mutex_unlock(&g_big);
    }
    if (S->is_root) {
        const int W = S->W;
        int *owner, *where;
        double *wt;
        S->C = chunks ? chunks : run_chunks_for((n + W - 1) / W);
        S->jobs = jobs; S->outs = outs; S->n = n;
        owner = (int *)malloc(((size_t)n + 1) * sizeof *owner); where = (int *)malloc(((size_t)n + 1) * sizeof *where);
        wt = (double *)calloc((size_t)n + 1, sizeof *wt);
        S->cnt = (int *)calloc((size_t)W * S->C, sizeof *S->cnt); S->start = (int *)calloc((size_t)W * S->C, sizeof *S->start);
        S->order = (int *)malloc(((size_t)n + 1) * sizeof *S->order); S->jbuf = (mz_job *)malloc(((size_t)n + 1) * sizeof *S->jbuf);
        S->obuf = (mz_out *)malloc(((size_t)n + 1) * sizeof *S->obuf);
        S->s_img = (rslot *)calloc(2 * (size_t)W, sizeof(rslot)); S->s_exc = (rslot *)calloc(2 * (size_t)W, sizeof(rslot));
        S->s_hdr = (rslot *)calloc((size_t)W, sizeof(rslot)); S->r_hdr = (rslot *)calloc((size_t)W, sizeof(rslot));
        S->r_res = (rslot *)calloc(RUN_SLOTS * (size_t)W, sizeof(rslot));
        S->pdesc = (mz_link_desc *)calloc(RUN_SLOTS * (size_t)W, sizeof *S->pdesc); S->rbytes = (int64_t *)calloc(RUN_SLOTS * (size_t)W, sizeof *S->rbytes);
        if (!owner || !where || !wt || !S->cnt || !S->start || !S->order || !S->jbuf || !S->obuf || !S->s_img || !S->s_exc || !S->s_hdr || !S->r_hdr || !S->r_res || !S->pdesc || !S->rbytes) {
            free(owner); free(where); free(wt); mzi_set_err("out of memory"); run_note(S); goto out;
        }
        for (p = 0; p < n; ++p) wt[p] = shard_weight(&jobs[p]);
        /* ranks x chunks bins in the snake's order c * W + r -- the heaviest pairs go round the RANKS first -- kept as b = r * C + c */
        {
            int *cnt2 = (int *)calloc((size_t)W * S->C, sizeof *cnt2), *start2 = (int *)calloc((size_t)W * S->C, sizeof *start2);
            int bad = !cnt2 || !start2 || mzi_deal_snake(n, wt, W * S->C, owner, where, cnt2, start2), b, pos = 0;
            if (!bad) {
                for (r = 0; r < W; ++r) for (t = 0; t < S->C; ++t) { b = r * S->C + t; S->cnt[b] = cnt2[t * W + r]; S->start[b] = pos; pos += S->cnt[b]; }
                for (p = 0; p < n; ++p) {
                    const int sb = owner[p], rr = sb % W, cc = sb / W, at = S->start[rr * S->C + cc] + (where[p] - start2[sb]);
                    S->jbuf[at] = jobs[p]; S->order[at] = p;
                }
            } else if (!cnt2 || !start2) mzi_set_err("out of memory");
            free(cnt2); free(start2);
            if (bad) { free(owner); free(where); free(wt); run_note(S); goto out; }
        }
        free(owner); free(where); free(wt);
        for (p = 0; p < n; ++p) { outs[p].status = MZ_E_DEVICE; outs[p].badrow = -1; outs[p].OM = 0; outs[p].cols = NULL; outs[p].block = NULL; outs[p].score[0] = outs[p].score[1] = outs[p].score[2] = 0; }
        if (run_pack(S, 0)) { run_note(S); goto out; }       /* (the first chunk: nothing to hide it behind) */
    }
    /* the steps.  An error on this rank is noted and the exchange goes on where it can (a failed chunk travels as such); only a
     * transport that fails ends it -- there is nothing to go on with */
    for (t = 0; S->C == 0 || t < S->C + 4; ++t) {
        if (S->is_root) {
            S->w_step = t; S->th_on = pthread_create(&S->th, NULL, run_host_thread, S) == 0;
            if (!S->th_on) run_host_thread(S);
        }
        if (t - 1 >= 0 && t - 1 < S->C && run_align_start(S, t - 1)) run_note(S);
        rc = run_comm(S, t);
        if (S->is_root) {
            if (S->th_on) pthread_join(S->th, NULL);
            S->th_on = 0;
            if (S->w_rc) { mzi_set_err("%s", S->w_err); run_note(S); }
        }
        if (rc) { run_note(S); break; }
        if (t - 1 >= 0 && t - 1 < S->C) { if (run_align_wait(S, t - 1)) run_note(S); run_count(S, t - 1); }
    }
out:
    if (S->aligning) hipStreamSynchronize((hipStream_t)mz_stream());
    for (t = 0; t < RUN_SLOTS; ++t) { rslot_drop(c, &S->img[t]); rslot_drop(c, &S->exc[t]); rslot_drop(c, &S->res[t]); }
    rslot_drop(c, &S->hdr_s); rslot_drop(c, &S->hdr_r);
    for (r = 0; S->s_img && r < 2 * S->W; ++r) { rslot_drop(c, &S->s_img[r]); if (S->s_exc) rslot_drop(c, &S->s_exc[r]); }
    for (r = 0; S->s_hdr && r < S->W; ++r) { rslot_drop(c, &S->s_hdr[r]); if (S->r_hdr) rslot_drop(c, &S->r_hdr[r]); }
    for (r = 0; S->r_res && r < RUN_SLOTS * S->W; ++r) rslot_drop(c, &S->r_res[r]);
    if (S->d_img) hipFree(S->d_img);
    if (S->d_exc) hipFree(S->d_exc);
    if (S->d_res) hipFree(S->d_res);
    if (S->ev_ready) { hipEventDestroy(S->ev0); hipEventDestroy(S->ev1); }
    if (times) {
        times->chunks = S->C; times->steps = S->C + 4; times->pack_s = S->t_pack; times->comm_s = S->t_comm; times->align_s = S->t_align; times->assemble_s = S->t_asm;
        times->wall_s = mzi_now_s() - t_start; times->pairs = S->my_pairs; times->cells = S->my_cells; times->failed = S->my_failed;
    }
    rc = S->rc ? -1 : S->is_root ? S->failed : 0;
    if (S->rc) mzi_set_err("%s", S->err);
    free(S->cnt); free(S->start); free(S->order); free(S->jbuf); free(S->obuf); free(S->s_img); free(S->s_exc); free(S->s_hdr); free(S->r_hdr);
    free(S->r_res); free(S->pdesc); free(S->rbytes); free(S->cb_result);
    free(S);
    return rc;
}
