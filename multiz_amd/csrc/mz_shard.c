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
