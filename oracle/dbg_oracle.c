/*
 * CPU oracle, C restatement of the graph-construction half of the hot path.
 * TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg; the product never links or loads it.
 *
 * Restates, for the DNA alphabet, what debruijn.py:98-147 (get_graph_from_reads) and
 * debruijn.py:213-222 (edge_count_table) compute:
 *   - for every read with len > k (debruijn.py:126), every window pos in [0, len-k]
 *     is a vertex occurrence; every pos < len-k is an occurrence of the edge
 *     (k-mer, next base);
 *   - per distinct k-mer: the 4 successor counts (the edge_count_table entries
 *     k-mer + base) and the first occurrence, kept as
 *     stamp = (byte offset << 1) | (pos != 0)   [indegree, debruijn.py:134,141-142].
 * Output order is first-occurrence order == the reference's dict order.
 *
 * Parity status: pinned -- tests/test_oracle_c.py checks it against
 * oracle/dbg_oracle.py, which is itself pinned by the reference's vectors.
 *
 * Base code = (ascii >> 1) & 3 (A=0 C=1 T=2 G=3), same packing as include/dbg.h.
 * Single-threaded, plain C, open-addressing table; k <= 31 in one 64-bit word, 32..63 in unsigned __int128.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 orc_u128; /* k in 32..63: a k-mer is up to 126 bits */

typedef struct {
    uint64_t key;
    uint64_t stamp;
    uint32_t cnt[4];
} orc_slot;

typedef struct {
    orc_u128 key;
    uint64_t stamp;
    uint32_t cnt[4];
    uint64_t pad_;
} orc_wslot;

typedef struct {
    orc_slot *tab;   /* k <= 31 */
    orc_wslot *wtab; /* k >= 32 */
    uint64_t cap, n_nodes, n_kmer_inst, n_edge_inst;
    int k;
} orc_t;

static uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}
static uint64_t hash_narrow(uint64_t key) { return mix64(key); }
static uint64_t hash_wide(orc_u128 key) { return mix64((uint64_t)key ^ mix64((uint64_t)(key >> 64) + 0x9E3779B97F4A7C15ULL)); }

void orc_free(orc_t *o) {
    if (!o) return;
    free(o->tab);
    free(o->wtab);
    free(o);
}

/* The scan of debruijn.py:126-143 over one key width.  Returns 0, or -1 on a byte outside ACGT. */
#define ORC_DEFINE_SCAN(NAME, KEY_T, SLOT_T, TAB, HASH)                                                     \
    static int NAME(orc_t *o, const char *bases, const uint64_t *offsets, uint64_t n_reads) {               \
        const int k = o->k;                                                                                  \
        const uint64_t mask = o->cap - 1;                                                                    \
        const KEY_T empty = ~(KEY_T)0, kmask = (((KEY_T)1) << (2 * k)) - 1;                                  \
        SLOT_T *tab = o->TAB;                                                                                \
        for (uint64_t r = 0; r < n_reads; ++r) {                                                             \
            const uint64_t beg = offsets[r], len = offsets[r + 1] - beg;                                     \
            if (len <= (uint64_t)k) continue; /* debruijn.py:126 */                                          \
            const unsigned char *s = (const unsigned char *)bases + beg;                                     \
            KEY_T key = 0;                                                                                   \
            for (uint64_t i = 0; i <= len; ++i) {                                                            \
                const unsigned char c = i < len ? s[i] : 0;                                                  \
                if (i < len && c != 'A' && c != 'C' && c != 'G' && c != 'T') return -1;                      \
                if (i >= (uint64_t)k) {                                                                      \
                    /* window [i-k, i) is complete in `key`; s[i] is its successor base (none at i == len) */\
                    const uint64_t pos = i - k;                                                              \
                    uint64_t h = HASH(key) & mask;                                                           \
                    while (tab[h].key != empty && tab[h].key != key) h = (h + 1) & mask;                     \
                    SLOT_T *e = &tab[h];                                                                     \
                    if (e->key == empty) {                                                                   \
                        e->key = key;                                                                        \
                        e->stamp = ((beg + pos) << 1) | (pos != 0);                                          \
                        e->cnt[0] = e->cnt[1] = e->cnt[2] = e->cnt[3] = 0;                                   \
                        o->n_nodes++;                                                                        \
                    }                                                                                        \
                    o->n_kmer_inst++;                                                                        \
                    if (i < len) { e->cnt[(c >> 1) & 3]++; o->n_edge_inst++; }                               \
                }                                                                                            \
                key = ((key << 2) | ((c >> 1) & 3)) & kmask;                                                 \
            }                                                                                                \
        }                                                                                                    \
        return 0;                                                                                            \
    }

ORC_DEFINE_SCAN(scan_narrow, uint64_t, orc_slot, tab, hash_narrow)
ORC_DEFINE_SCAN(scan_wide, orc_u128, orc_wslot, wtab, hash_wide)

/* returns NULL on bad input (k outside 1..63, byte outside ACGT, allocation failure) */
orc_t *orc_build(const char *bases, const uint64_t *offsets, uint64_t n_reads, int k) {
    if (k < 1 || k > 63) return NULL;
    uint64_t windows = 0;
    for (uint64_t r = 0; r < n_reads; ++r) {
        uint64_t len = offsets[r + 1] - offsets[r];
        if (len > (uint64_t)k) windows += len - k + 1;
    }
    orc_t *o = (orc_t *)calloc(1, sizeof(orc_t));
    if (!o) return NULL;
    o->k = k;
    o->cap = 1024;
    while (o->cap < windows * 2) o->cap <<= 1;
    int rc;
    if (k <= 31) {
        o->tab = (orc_slot *)malloc(o->cap * sizeof(orc_slot));
        if (!o->tab) { free(o); return NULL; }
        memset(o->tab, 0xFF, o->cap * sizeof(orc_slot));
        rc = scan_narrow(o, bases, offsets, n_reads);
    } else {
        o->wtab = (orc_wslot *)malloc(o->cap * sizeof(orc_wslot));
        if (!o->wtab) { free(o); return NULL; }
        memset(o->wtab, 0xFF, o->cap * sizeof(orc_wslot));
        rc = scan_wide(o, bases, offsets, n_reads);
    }
    if (rc) { orc_free(o); return NULL; }
    return o;
}

uint64_t orc_n_nodes(const orc_t *o) { return o->n_nodes; }
uint64_t orc_n_kmer_instances(const orc_t *o) { return o->n_kmer_inst; }
uint64_t orc_n_edge_instances(const orc_t *o) { return o->n_edge_inst; }

static int cmp_stamp(const void *a, const void *b) {
    const uint64_t x = ((const orc_wslot *)a)->stamp, y = ((const orc_wslot *)b)->stamp;
    return x < y ? -1 : x > y;
}

/* nodes in first-occurrence (dict) order; arrays sized orc_n_nodes (counts: n*4, by base code).
 * keys: low 64 bits of the k-mer; keys_hi (may be NULL): bits 64.. (zero for k <= 32). */
int orc_export2(const orc_t *o, uint64_t *keys, uint64_t *keys_hi, uint64_t *stamps, uint32_t *counts) {
    orc_wslot *tmp = (orc_wslot *)malloc((o->n_nodes ? o->n_nodes : 1) * sizeof(orc_wslot));
    if (!tmp) return -1;
    uint64_t n = 0;
    for (uint64_t i = 0; i < o->cap; ++i) {
        if (o->tab && o->tab[i].key != ~0ULL) {
            tmp[n].key = o->tab[i].key;
            tmp[n].stamp = o->tab[i].stamp;
            memcpy(tmp[n].cnt, o->tab[i].cnt, 16);
            ++n;
        } else if (o->wtab && o->wtab[i].key != ~(orc_u128)0) {
            tmp[n++] = o->wtab[i];
        }
    }
    qsort(tmp, n, sizeof(orc_wslot), cmp_stamp);
    for (uint64_t i = 0; i < n; ++i) {
        if (keys) keys[i] = (uint64_t)tmp[i].key;
        if (keys_hi) keys_hi[i] = (uint64_t)(tmp[i].key >> 64);
        if (stamps) stamps[i] = tmp[i].stamp;
        if (counts) memcpy(counts + 4 * i, tmp[i].cnt, 16);
    }
    free(tmp);
    return 0;
}

int orc_export(const orc_t *o, uint64_t *keys, uint64_t *stamps, uint32_t *counts) {
    return orc_export2(o, keys, NULL, stamps, counts);
}

/* ------------------------------------------------------------------------------------------------------------
 * Multi-threaded variant of the same scan (k <= 31): the CPU baseline of bench.py on ALL host cores.
 * Hash-partitioned: thread t owns the k-mers whose hash falls into its slice, scans every read, rolls every window
 * (a shift and an or) and inserts only its own -- no locks, no atomics, nothing shared but the read-only input; each
 * thread's table grows by doubling.  Same per-node results as orc_build (tests/test_oracle_c.py compares the digest).
 * out[0] nodes, out[1] distinct edges, out[2] k-mer instances, out[3] edge instances,
 * out[4] digest = sum over nodes of mix64(key ^ mix64(stamp) ^ mix64(cnt0 + 3 cnt1 + 5 cnt2 + 7 cnt3 + 1)).
 * ------------------------------------------------------------------------------------------------------------ */
#include <pthread.h>

typedef struct {
    const char *bases;
    const uint64_t *offsets;
    uint64_t n_reads;
    int k, t, n_threads, rc;
    uint64_t out[5];
} orc_mt_job;

static uint64_t orc_node_digest(uint64_t key, uint64_t stamp, const uint32_t *cnt) {
    return mix64(key ^ mix64(stamp) ^ mix64((uint64_t)cnt[0] + 3ull * cnt[1] + 5ull * cnt[2] + 7ull * cnt[3] + 1));
}

static orc_slot *mt_find(orc_slot **ptab, uint64_t *pcap, uint64_t *pn, uint64_t key, uint64_t h) {
    if ((*pn + 1) * 2 > *pcap) { /* grow: rehash into twice the slots */
        const uint64_t ncap = *pcap * 2;
        orc_slot *nt = (orc_slot *)malloc(ncap * sizeof(orc_slot));
        if (!nt) return NULL;
        memset(nt, 0xFF, ncap * sizeof(orc_slot));
        for (uint64_t i = 0; i < *pcap; ++i) {
            if ((*ptab)[i].key == ~0ULL) continue;
            uint64_t j = hash_narrow((*ptab)[i].key) & (ncap - 1);
            while (nt[j].key != ~0ULL) j = (j + 1) & (ncap - 1);
            nt[j] = (*ptab)[i];
        }
        free(*ptab);
        *ptab = nt;
        *pcap = ncap;
    }
    const uint64_t mask = *pcap - 1;
    uint64_t j = h & mask;
    while ((*ptab)[j].key != ~0ULL && (*ptab)[j].key != key) j = (j + 1) & mask;
    return &(*ptab)[j];
}

static void *mt_worker(void *arg) {
    orc_mt_job *jb = (orc_mt_job *)arg;
    const int k = jb->k;
    const uint64_t kmask = (k == 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
    uint64_t cap = 1 << 16, n = 0, n_inst = 0, n_einst = 0;
    orc_slot *tab = (orc_slot *)malloc(cap * sizeof(orc_slot));
    if (!tab) { jb->rc = -2; return NULL; }
    memset(tab, 0xFF, cap * sizeof(orc_slot));
    const uint64_t T = (uint64_t)jb->n_threads, me = (uint64_t)jb->t;
    for (uint64_t r = 0; r < jb->n_reads; ++r) {
        const uint64_t beg = jb->offsets[r], len = jb->offsets[r + 1] - beg;
        if (len <= (uint64_t)k) continue; /* debruijn.py:126 */
        const unsigned char *s = (const unsigned char *)jb->bases + beg;
        uint64_t key = 0;
        for (uint64_t i = 0; i <= len; ++i) {
            const unsigned char c = i < len ? s[i] : 0;
            if (i < len && c != 'A' && c != 'C' && c != 'G' && c != 'T') { jb->rc = -1; free(tab); return NULL; }
            if (i >= (uint64_t)k) {
                const uint64_t h = hash_narrow(key);
                if ((((h >> 32) * T) >> 32) == me) { /* this thread's slice of the key space */
                    const uint64_t pos = i - k;
                    orc_slot *e = mt_find(&tab, &cap, &n, key, h);
                    if (!e) { jb->rc = -2; free(tab); return NULL; }
                    if (e->key == ~0ULL) {
                        e->key = key;
                        e->stamp = ((beg + pos) << 1) | (pos != 0);
                        e->cnt[0] = e->cnt[1] = e->cnt[2] = e->cnt[3] = 0;
                        ++n;
                    }
                    ++n_inst;
                    if (i < len) { e->cnt[(c >> 1) & 3]++; ++n_einst; }
                }
            }
            key = ((key << 2) | ((c >> 1) & 3)) & kmask;
        }
    }
    uint64_t edges = 0, dig = 0;
    for (uint64_t i = 0; i < cap; ++i) {
        if (tab[i].key == ~0ULL) continue;
        for (int b = 0; b < 4; ++b) edges += tab[i].cnt[b] != 0;
        dig += orc_node_digest(tab[i].key, tab[i].stamp, tab[i].cnt);
    }
    free(tab);
    jb->out[0] = n; jb->out[1] = edges; jb->out[2] = n_inst; jb->out[3] = n_einst; jb->out[4] = dig;
    return NULL;
}

/* 0 on success, -1 byte outside ACGT, -2 allocation or thread failure, -3 bad argument */
int orc_build_mt(const char *bases, const uint64_t *offsets, uint64_t n_reads, int k, int n_threads, uint64_t *out5) {
    if (k < 1 || k > 31 || n_threads < 1 || n_threads > 4096 || !out5) return -3;
    orc_mt_job *jobs = (orc_mt_job *)calloc((size_t)n_threads, sizeof(orc_mt_job));
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    if (!jobs || !th) { free(jobs); free(th); return -2; }
    int rc = 0, started = 0;
    for (int t = 0; t < n_threads; ++t) {
        jobs[t].bases = bases; jobs[t].offsets = offsets; jobs[t].n_reads = n_reads;
        jobs[t].k = k; jobs[t].t = t; jobs[t].n_threads = n_threads;
        if (pthread_create(&th[t], NULL, mt_worker, &jobs[t])) { rc = -2; break; }
        ++started;
    }
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
    for (int i = 0; i < 5; ++i) out5[i] = 0;
    for (int t = 0; t < started; ++t) {
        if (jobs[t].rc && !rc) rc = jobs[t].rc;
        for (int i = 0; i < 5; ++i) out5[i] += jobs[t].out[i];
    }
    free(jobs);
    free(th);
    return rc;
}

/* ------------------------------------------------------------------------------------------------------------
 * The same multi-threaded build with the k-mers partitioned ONCE (what a tuned CPU builder does; orc_build_mt above
 * lets every thread roll and hash every window and keep 1/T of them).  Phase 1: thread t scans ITS share of the reads
 * and appends every window as (key, stamp | next base) to the list of the thread that owns the key's hash slice.
 * Phase 2: thread d builds its table from the T lists addressed to it, in reader order (ascending positions: the first
 * instance inserts the first-occurrence stamp).  Same slices, same per-node results, same digest as orc_build_mt.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct { uint64_t key, meta; } orc_tuple;   /* meta: stamp | (next base code + 1) << 61, 0 in the top bits = no successor */
typedef struct { orc_tuple *p; uint64_t n, cap; } orc_vec;
typedef struct {
    const char *bases;
    const uint64_t *offsets;
    uint64_t r_beg, r_end;
    int k, t, n_threads, rc;
    orc_vec *lists;   /* [n_threads][n_threads]: lists[t * T + d] = what reader t found for owner d */
    uint64_t out[5];
} orc_p_job;

static int vec_push(orc_vec *v, uint64_t key, uint64_t meta) {
    if (v->n == v->cap) {
        const uint64_t nc = v->cap ? v->cap * 2 : 4096;
        orc_tuple *np = (orc_tuple *)realloc(v->p, nc * sizeof(orc_tuple));
        if (!np) return -1;
        v->p = np; v->cap = nc;
    }
    v->p[v->n].key = key; v->p[v->n].meta = meta; ++v->n;
    return 0;
}

static void *p_scan(void *arg) {
    orc_p_job *jb = (orc_p_job *)arg;
    const int k = jb->k;
    const uint64_t kmask = (k == 32) ? ~0ULL : ((1ULL << (2 * k)) - 1), T = (uint64_t)jb->n_threads;
    orc_vec *mine = jb->lists + (uint64_t)jb->t * T;
    for (uint64_t r = jb->r_beg; r < jb->r_end; ++r) {
        const uint64_t beg = jb->offsets[r], len = jb->offsets[r + 1] - beg;
        if (len <= (uint64_t)k) continue; /* debruijn.py:126 */
        const unsigned char *s = (const unsigned char *)jb->bases + beg;
        uint64_t key = 0;
        for (uint64_t i = 0; i <= len; ++i) {
            const unsigned char c = i < len ? s[i] : 0;
            if (i < len && c != 'A' && c != 'C' && c != 'G' && c != 'T') { jb->rc = -1; return NULL; }
            if (i >= (uint64_t)k) {
                const uint64_t h = hash_narrow(key), pos = i - k;
                const uint64_t d = ((h >> 32) * T) >> 32;
                const uint64_t stamp = ((beg + pos) << 1) | (pos != 0);
                if (stamp >> 61) { jb->rc = -3; return NULL; }
                if (vec_push(&mine[d], key, stamp | (i < len ? (uint64_t)(((c >> 1) & 3) + 1) << 61 : 0))) { jb->rc = -2; return NULL; }
            }
            key = ((key << 2) | ((c >> 1) & 3)) & kmask;
        }
    }
    return NULL;
}

static void *p_build(void *arg) {
    orc_p_job *jb = (orc_p_job *)arg;
    const uint64_t T = (uint64_t)jb->n_threads, me = (uint64_t)jb->t;
    uint64_t cap = 1 << 16, n = 0, n_inst = 0, n_einst = 0;
    orc_slot *tab = (orc_slot *)malloc(cap * sizeof(orc_slot));
    if (!tab) { jb->rc = -2; return NULL; }
    memset(tab, 0xFF, cap * sizeof(orc_slot));
    for (uint64_t t = 0; t < T; ++t) {
        orc_vec *v = jb->lists + t * T + me;
        for (uint64_t q = 0; q < v->n; ++q) {
            const uint64_t key = v->p[q].key, meta = v->p[q].meta;
            orc_slot *e = mt_find(&tab, &cap, &n, key, hash_narrow(key));
            if (!e) { jb->rc = -2; free(tab); return NULL; }
            if (e->key == ~0ULL) {
                e->key = key;
                e->stamp = meta & ((1ULL << 61) - 1);
                e->cnt[0] = e->cnt[1] = e->cnt[2] = e->cnt[3] = 0;
                ++n;
            }
            ++n_inst;
            if (meta >> 61) { e->cnt[(meta >> 61) - 1]++; ++n_einst; }
        }
        free(v->p); v->p = NULL; v->n = v->cap = 0;   /* consumed: only this thread reads list (t, me) */
    }
    uint64_t edges = 0, dig = 0;
    for (uint64_t i = 0; i < cap; ++i) {
        if (tab[i].key == ~0ULL) continue;
        for (int b = 0; b < 4; ++b) edges += tab[i].cnt[b] != 0;
        dig += orc_node_digest(tab[i].key, tab[i].stamp, tab[i].cnt);
    }
    free(tab);
    jb->out[0] = n; jb->out[1] = edges; jb->out[2] = n_inst; jb->out[3] = n_einst; jb->out[4] = dig;
    return NULL;
}

/* as orc_build_mt; 16 bytes of list per k-mer instance are held between the phases */
int orc_build_mt_partitioned(const char *bases, const uint64_t *offsets, uint64_t n_reads, int k, int n_threads, uint64_t *out5) {
    if (k < 1 || k > 31 || n_threads < 1 || n_threads > 1024 || !out5) return -3;
    const uint64_t T = (uint64_t)n_threads;
    orc_p_job *jobs = (orc_p_job *)calloc(T, sizeof(orc_p_job));
    pthread_t *th = (pthread_t *)calloc(T, sizeof(pthread_t));
    orc_vec *lists = (orc_vec *)calloc(T * T, sizeof(orc_vec));
    int rc = 0;
    if (!jobs || !th || !lists) rc = -2;
    for (int phase = 0; phase < 2 && !rc; ++phase) {
        int started = 0;
        for (uint64_t t = 0; t < T; ++t) {
            jobs[t].bases = bases; jobs[t].offsets = offsets; jobs[t].k = k; jobs[t].t = (int)t; jobs[t].n_threads = n_threads;
            jobs[t].r_beg = n_reads * t / T; jobs[t].r_end = n_reads * (t + 1) / T; jobs[t].lists = lists;
            if (pthread_create(&th[t], NULL, phase ? p_build : p_scan, &jobs[t])) { rc = -2; break; }
            ++started;
        }
        for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
        for (uint64_t t = 0; t < T && !rc; ++t) rc = jobs[t].rc;
    }
    for (int i = 0; i < 5; ++i) out5[i] = 0;
    if (!rc) for (uint64_t t = 0; t < T; ++t) for (int i = 0; i < 5; ++i) out5[i] += jobs[t].out[i];
    if (lists) for (uint64_t i = 0; i < T * T; ++i) free(lists[i].p);
    free(lists); free(jobs); free(th);
    return rc;
}
