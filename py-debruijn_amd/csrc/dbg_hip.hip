// MI355X (gfx950) implementation of the C ABI in include/dbg.h.
//
// Hot path of py-debruijn (reference lines in brackets):
//   k_count        encode + hash insert + edge counters      [debruijn.py:98-147, :213-222]
//   k_gather/k_succ/k_csr  compaction to node arrays + CSR   [the (vertices, edges) dicts]
//   k_prune        pruningEdges + branch detection            [debruijn.py:150-166, :230-236]
//   k_tip_*        tip removal by deterministic reservations  [debruijn.py:169-186, :241-254]
//   k_pull_reads   reads containing a branch k-mer            [debruijn.py:274-278]
//   k_walk_*       contig walk / DFS + getScore               [debruijn.py:288-347, II:14-18]
//
// Integer/indexing work only: no MFMA.  Bound by HBM traffic of the hash table.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>  // library sorts off the timed path: dbg_export_dict_order, dbg_export_marked, dbg_support_read_scores, dbg_export_sorted_fasta

#include "../../include/dbg.h"
#include "dbg_device.h"
#include "dbg_sk.h"
#include "dbg_sk2.h"
#include "dbg_sk3.h"
#include "dbg_wsk2.h"
#include "dbg_generic.h"
#include "dbg_genref.h"
#include "dbg_wide.h"
#include "dbg_support.h"
#include "dbg_wsk.h"

using namespace dbgk;

// ------------------------------------------------------------------------------------------
// handle
// ------------------------------------------------------------------------------------------
struct dbg {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    // reads
    char *d_bases = nullptr;
    bool own_bases = false;
    uint64_t n_bytes = 0;
    uint64_t *d_offsets = nullptr;
    bool own_offsets = false;
    uint64_t n_reads = 0;
    uint32_t *d_startbits = nullptr;  // bit p set: a read starts at byte p (bit n_bytes = sentinel)
    uint64_t startbits_words = 0;

    // build
    int k = 0;
    Slot *d_tab = nullptr;
    uint64_t cap = 0;
    int cap_log2 = 0;
    uint32_t *d_occ = nullptr;  // occupancy bitmap over slots
    uint64_t *d_scalars = nullptr;  // [0] error flags [1] N_k [2] N_e [3..63] scratch [64..127] kernel descriptors
    uint64_t n_kmer_inst = 0, n_edge_inst = 0;
    uint64_t n_nodes = 0, n_edges = 0;
    uint64_t *d_keys = nullptr, *d_stamps = nullptr;
    uint64_t *d_keys_hi = nullptr;  // k > 31 only: bits 2k-1..64 of every node's k-mer (dbg_wide.h)
    uint32_t *d_cnt = nullptr;
    uint8_t *d_flags = nullptr;
    uint8_t *d_order = nullptr;  // successor codes ranked by (count desc, ascii asc), 2 bits each
    uint8_t *d_deg = nullptr;    // distinct successors (pre-pruning outdegree) -- feeds the CSR row scan
    uint8_t *d_fsorder = nullptr; // successor codes in first-seen order (dbg_refine_edge_order)
    bool order_exact = false;
    uint32_t *d_succ = nullptr;
    bool nodes_in_arena = false;  // node arrays borrowed from ar_node (super-k-mer engine)
    bool csr_built = false;       // rowptr/col/ecnt already written by k_sk_count
    uint64_t *d_rowptr = nullptr;
    uint32_t *d_col = nullptr, *d_ecnt = nullptr;
    // engine 0 writes ONE representation in the build: keys, stamps (32-bit while the reads stay below 2 GiB), one byte
    // per node (indegree | present bases << 1) and the CSR with 32-bit row pointers.  The dense per-base views above
    // (d_cnt, d_succ, d_order, 64-bit d_stamps / d_rowptr, plain d_flags) are derived by ensure_dense() on first use.
    bool dense_pending = false;
    uint32_t *d_rowptr32 = nullptr;
    void *d_stamps_st = nullptr;
    int stamps_st_bytes = 0;

    // prune / tips
    bool pruned = false, tipped = false;
    uint64_t n_branch = 0, n_pulled = 0, tip_rounds = 0;
    uint64_t *d_pull_rank = nullptr;

    // pull reads
    uint8_t *d_read_flags = nullptr;
    uint64_t n_pull_reads = 0;
    bool pull_reads_done = false;

    // walk
    uint64_t n_starts = 0, n_contigs = 0, contig_chars = 0;
    bool starts_known = false;    // n_starts counted for the current graph
    // engine 0, single GPU: the bucketed super-k-mer records of the last build (arena-owned, valid until the next build /
    // extraction): dbg_refine_edge_order re-reads them bucket by bucket instead of streaming the reads against a global set
    struct { const uint64_t *w0 = nullptr, *w1 = nullptr; const void *st = nullptr; int st_bytes = 0;
             const uint64_t *b_start = nullptr, *b_cnt = nullptr; bool valid = false; } sk_src;
    uint64_t *d_ctg_off = nullptr;
    char *d_ctg_chars = nullptr;
    uint64_t *d_ctg_score = nullptr, *d_ctg_stamp = nullptr;
    uint32_t *d_ctg_seq = nullptr;
    uint32_t *d_ctg_start = nullptr;  // list-ranking walk: start node of every contig (text on demand, dbg_export_contig_text)
    uint32_t *d_lift = nullptr;       // binary-lifting tables of the last walk, built on the first text request
    int lift_levels = 0;
    bool walked = false;          // contig text materialised
    bool walk_indexed = false;    // contig index (offsets, scores, start stamps) valid
    uint64_t walk_jump_min = 1ull << 14;  // non-final walk: list ranking from this many nodes on (below: one thread per start)

    // alphabet / node layout: 2 bits and 4 successors for DNA, 5 bits and 32 for the generic engine
    int D = 4, sym_bits = 2, n_sym = 0;
    bool alpha_known = false, is_dna = true;
    uint8_t alphabet[32] = {0};       // generic: code -> byte (codes are assigned in byte order)
    uint8_t *d_lut = nullptr;         // generic: byte -> code (256 entries)
    char *d_alpha = nullptr;          // generic: code -> byte (32 entries, device)
    uint32_t *d_keepmask = nullptr;   // generic: surviving successors per node
    uint8_t *d_rank_mc = nullptr, *d_rank_fs = nullptr;  // generic: [n][32] successor codes by rank

    // options (dbg_set_option)
    int engine = 0;          // 0 = super-k-mer partitioned build, 1 = single global hash table
    int bucket_bits = 0;     // 0 = auto (super-k-mer engine)
    int lds_slots = 4096;    // LDS table slots per bucket workgroup (2048 or 4096)
    int phase_limit = 0;     // ablation of k_sk_count (timing only; the build then fails on purpose)
    int est_scale_pct = 100; // test hook: scales the distinct-k-mer estimate (a low one exercises the capacity retry)
    int wide_engine = 1;     // k = 32..63: 1 = super-k-mer / LDS engine (dbg_wsk.h), 0 = global reference-keyed table (dbg_wide.h)
    int count_kernel_u64 = 3; // option "count_kernel_u64": the count kernel of 64-bit stamps (shards, reads of 2 GiB and more): 3 = k_sk_count3 (one hint per slot; 12.2 ms on a 10 M-read shard), 1 = k_sk_count (13.4), 2 = k_sk_count2 (15.5: 320 staged records)
    int stamp64 = 0;         // option "stamp64" 1: dbg_build keeps 64-bit stamps even for reads below 2 GiB (what reads of 2 GiB and more get by themselves; tests)
    int resolve_sorted = 0;  // option "resolve_sorted" 1: cross-bucket queries grouped by the 512 level-1 groups of their target before k_succ_resolve.  Measured (tools/resolve_ab.py, 10 M reads): 3.09 ms with the grouping against 2.05 ms in the askers' order -- off
    int wcount_kernel = 2;   // 32 <= k <= 63, 32-bit stamps: 2 = k_wsk_count2 (one successor hint per slot, deferred lookups), 1 = k_wsk_count
    int count_kernel = 2;    // k <= 31, 4096 slots: 2 = k_sk_count2 (successor hints, 16-bit counters; falls back to 1 on counter overflow), 1 = k_sk_count

    // grow-only device arena of the super-k-mer engine: hipMalloc of tens of GB costs seconds,
    // so buffers survive across dbg_build calls on the same handle
    struct Buf {
        void *p = nullptr;
        uint64_t bytes = 0;
    };
    Buf ar_rec[2][3], ar_q[2][3], ar_node[8], ar_misc[9], ar_csr[4], ar_dir, ar_l2, ar_scan, ar_shard[6], ar_walk[3], ar_wide[6], ar_refine[4], ar_tips[4], ar_part[4][3] /* records of dbg_shard_extract_part, per part */;
    int sk_T = 0, sk_l1 = 0, sk_l2 = 0, sk_nb2 = 0 /* scaled second level, 0 = power of two */, sk_cap = 0;
    bool refine_streaming = false;  // option (tests): dbg_refine_edge_order always takes the pass over the reads
    int target_distinct = 0;  // option: mean distinct k-mers per bucket the auto geometry aims at (0 = default)
    int shard_stamp64 = 0;    // option: dbg_shard_extract hands out 64-bit rank-local stamps even below 2 GiB of reads (tests)
    int extract_generic = 0;  // option: 1 = the window-minimum-through-LDS extraction kernels instead of the per-window register ones (A/B, tests)
    uint64_t sk_n_ranges = 0, sk_n_buckets = 0;
    SkGeom sk_geom{};            // hash -> bucket mapping of the last partitioned build (k_succ_resolve)
    void *shard_state = nullptr;  // ShardState (multi-GPU builds)
    void *multipass = nullptr;    // MultiPass (dbg_build_multipass): the parts of the graph, parked in HBM
    bool dropping_parts = false;  // dbg_destroy / a build of another kind: nothing is parked
    dbg *spare_part = nullptr;    // the part handle of the last ONE-pass sharded build: its arenas serve the next one (bench loops)
    bool wide_owner = false;      // a part of a multi-pass build: successor ids are (owner byte, 32-bit local id), not tagged
    bool borrowed_stream = false; // sub-handle of a multi-pass build: the stream belongs to the parent
    bool arena_freed = false;     // an arena buffer went back to the memory pool since the last trim
    std::vector<uint64_t> host_seg_cnt;  // sk_extract: records per extraction segment (host copy)
    std::vector<uint64_t> host_scpre;    // multisplit_level: staging that must outlive an asynchronous upload
    uint64_t shard_node_limit = 0;  // option (tests): lowers the 2^29 - 16 node ids a shard may hand out
    bool partial_graph = false;   // the node table is one shard of several: successor ids point into other handles
    // branch k-mer lookup (pull-out reads)
    uint64_t *d_btab = nullptr;
    uint64_t btab_cap = 0;

    dbg_stats_t stats{};
};

// What a sharded build leaves behind for the exchange of successors owned by other ranks.
struct ShardState {
    int n_shards = 0, my_shard = 0, shard_bits = 0, k = 0;
    std::vector<uint64_t> q_start, q_cnt;   // remote queries grouped by owner (own group: count 0)
    uint64_t n_remote = 0;
    uint64_t n_kmer_inst_local = 0;          // k-mer instances of the reads this rank extracted
    std::vector<uint64_t> l1_counts;         // records per level-1 bucket (512) of the last dbg_shard_extract
    int rec_words = 1, rec_stamp_bytes = 4;  // what dbg_shard_extract handed out: words per record in d_w0, bytes per stamp
};
static ShardState &shard_of(dbg *h);
static void multipass_free(dbg *h);
static void drop_spare_part(dbg *h);

// Arena buffers come from the device's stream-ordered memory pool (hipMallocAsync on the handle's stream) with the
// pool told to keep what is freed: a hipMalloc / hipFree of several GB costs ~0.1 s each, and a multi-pass build
// allocates, trims and frees a few hundred GB in pieces of similar sizes -- 14 s of allocator calls around 0.4 s of
// kernels before this.  Everything a handle does is on its one stream, so stream order is program order.
static int buf_ensure(dbg *h, dbg::Buf &b, uint64_t bytes) {
    if (bytes == 0) bytes = 16;
    if (b.bytes >= bytes) return DBG_OK;
    if (b.p) { (void)hipFreeAsync(b.p, h->stream); h->arena_freed = true; }
    b.p = nullptr;
    b.bytes = 0;
    static const bool trace = getenv("DBG_TRACE_ALLOC") != nullptr;  // diagnostic: arena allocations that take longer than 5 ms
    const auto t_alloc = std::chrono::steady_clock::now();
    hipError_t e = hipMallocAsync(&b.p, bytes, h->stream);
    if (trace) {
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_alloc).count();
        if (ms > 5.0) fprintf(stderr, "[dbg] hipMallocAsync(%.3f GB) took %.1f ms (buffer %p of handle %p)\n", bytes / 1e9, ms, (void *)&b, (void *)h);
    }
    if (e != hipSuccess) {
        // the pool may be holding freed blocks of the wrong sizes: give them back to the device and try once more
        (void)hipGetLastError();
        (void)hipStreamSynchronize(h->stream);
        hipMemPool_t pool;
        if (hipDeviceGetDefaultMemPool(&pool, h->device) == hipSuccess) (void)hipMemPoolTrimTo(pool, 0);
        e = hipMallocAsync(&b.p, bytes, h->stream);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        h->err = std::string("hipMallocAsync(") + std::to_string(bytes) + "): " + hipGetErrorString(e);
        b.p = nullptr;
        return DBG_E_NOMEM;
    }
    b.bytes = bytes;
    return DBG_OK;
}
static void buf_free(dbg *h, dbg::Buf &b) {
    if (b.p) { (void)hipFreeAsync(b.p, h->stream); h->arena_freed = true; }
    b.p = nullptr;
    b.bytes = 0;
}
// Blocks the pool kept for reuse go back to the device: called when a build is complete, so that what the library
// holds is what it uses (other allocators of the process -- torch -- see the rest) and a later build does not find the
// pool full of blocks of the wrong sizes.
static void pool_trim(dbg *h) {
    if (!h->arena_freed) return;
    (void)hipStreamSynchronize(h->stream);
    hipMemPool_t pool;
    if (hipDeviceGetDefaultMemPool(&pool, h->device) == hipSuccess) (void)hipMemPoolTrimTo(pool, 0);
    (void)hipGetLastError();
    h->arena_freed = false;
}

#define HIPCHK(h, call)                                                                      \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                    \
            return e_ == hipErrorOutOfMemory ? DBG_E_NOMEM : DBG_E_HIP;                      \
        }                                                                                    \
    } while (0)

#define CHK(expr)                 \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != DBG_OK) return rc_; \
    } while (0)

template <class T>
static int dev_alloc(dbg *h, T **p, uint64_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    HIPCHK(h, hipMalloc((void **)p, count * sizeof(T)));
    return DBG_OK;
}
template <class T>
static void dev_free(T *&p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

static inline unsigned grid_for(uint64_t n, unsigned block) { return (unsigned)((n + block - 1) / block); }

struct Timer {
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t s;
    explicit Timer(hipStream_t st) : s(st) {
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        (void)hipEventRecord(a, s);
    }
    double stop() {
        (void)hipEventRecord(b, s);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        return ms;
    }
    ~Timer() {
        (void)hipEventDestroy(a);
        (void)hipEventDestroy(b);
    }
};

// ------------------------------------------------------------------------------------------
// generic exclusive scan: out[i] = sum_{j<i} f(j); 256 threads x 16 items per block
// ------------------------------------------------------------------------------------------
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_TILE = 256 * SCAN_ITEMS;

template <class F>
__global__ __launch_bounds__(256) void k_scan_reduce(uint64_t n, F f, uint64_t *partial) {
    uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < n) s += f(base + i);
    uint64_t tot;
    (void)block_exscan_256(s, &tot);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// single block: partial[] -> exclusive prefix in place, grand total to *total
__global__ __launch_bounds__(256) void k_scan_partials(uint64_t *partial, uint64_t nblk, uint64_t *total) {
    uint64_t carry = 0;
    for (uint64_t base = 0; base < nblk; base += 256) {
        uint64_t i = base + threadIdx.x;
        uint64_t v = i < nblk ? partial[i] : 0;
        uint64_t tot;
        uint64_t ex = block_exscan_256(v, &tot);
        if (i < nblk) partial[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *total = carry;
}

template <class F, class OutT>
__global__ __launch_bounds__(256) void k_scan_write(uint64_t n, F f, const uint64_t *partial, OutT *out) {
    uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t v[SCAN_ITEMS];
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        v[i] = (base + i < n) ? f(base + i) : 0;
        s += v[i];
    }
    uint64_t tot;
    uint64_t run = partial[blockIdx.x] + block_exscan_256(s, &tot);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        if (base + i < n) out[base + i] = (OutT)run;
        run += v[i];
    }
}

// out must hold n entries; *d_total (device) receives the total; returns total on host too
template <class F, class OutT>
static int exclusive_scan(dbg *h, uint64_t n, F f, OutT *out, uint64_t *h_total) {
    uint64_t nblk = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (nblk == 0) nblk = 1;
    CHK(buf_ensure(h, h->ar_scan, (nblk + 1) * 8));
    uint64_t *partial = (uint64_t *)h->ar_scan.p;
    uint64_t *d_total = partial + nblk;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_reduce<F>), dim3((unsigned)nblk), dim3(256), 0, h->stream, n, f, partial);
    hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(256), 0, h->stream, partial, nblk, d_total);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_write<F, OutT>), dim3((unsigned)nblk), dim3(256), 0, h->stream, n, f,
                       partial, out);
    if (!h_total) return DBG_OK;  // the caller does not need the total on the host: no synchronisation
    hipError_t e = hipMemcpyAsync(h_total, d_total, 8, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) {
        h->err = std::string("exclusive_scan: ") + hipGetErrorString(e);
        return DBG_E_HIP;
    }
    return DBG_OK;
}

template <class F>
static int reduce_sum(dbg *h, uint64_t n, F f, uint64_t *h_total) {
    uint64_t nblk = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (nblk == 0) nblk = 1;
    CHK(buf_ensure(h, h->ar_scan, (nblk + 1) * 8));
    uint64_t *partial = (uint64_t *)h->ar_scan.p;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_reduce<F>), dim3((unsigned)nblk), dim3(256), 0, h->stream, n, f, partial);
    hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(256), 0, h->stream, partial, nblk, partial + nblk);
    hipError_t e = hipMemcpyAsync(h_total, partial + nblk, 8, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) {
        h->err = std::string("reduce_sum: ") + hipGetErrorString(e);
        return DBG_E_HIP;
    }
    return DBG_OK;
}

struct OccBytes32 {  // occupied slots among 32 one-byte flags (each 0 or 1)
    const uint8_t *occ;
    __device__ uint64_t operator()(uint64_t w) const {
        const uint4 a = reinterpret_cast<const uint4 *>(occ)[2 * w], b = reinterpret_cast<const uint4 *>(occ)[2 * w + 1];
        return __popc(a.x) + __popc(a.y) + __popc(a.z) + __popc(a.w) + __popc(b.x) + __popc(b.y) + __popc(b.z) + __popc(b.w);
    }
};
struct PopcWords {
    const uint32_t *w;
    __device__ uint64_t operator()(uint64_t i) const { return __popc(w[i]); }
};
struct DegOf {
    const uint8_t *deg;
    __device__ uint64_t operator()(uint64_t i) const { return deg[i]; }
};
struct U64At {
    const uint64_t *p;
    __device__ uint64_t operator()(uint64_t i) const { return p[i]; }
};
struct KmerInstances {  // k-mer instances of read i
    const uint64_t *off;
    uint64_t k;
    __device__ uint64_t operator()(uint64_t i) const { const uint64_t len = off[i + 1] - off[i]; return len >= k ? len - k + 1 : 0; }
};
struct WRecLenAt {   // k-mers of wide record i
    const uint64_t *w1;
    __device__ uint64_t operator()(uint64_t i) const { return (uint64_t)wrec_len(w1[i]); }
};
struct WRecSuccAt {  // 1 if the last k-mer of wide record i has a successor
    const uint64_t *w1;
    __device__ uint64_t operator()(uint64_t i) const { return (uint64_t)wrec_has_succ(w1[i]); }
};
struct FlagSet {
    const uint8_t *f;
    uint8_t mask, want;
    __device__ uint64_t operator()(uint64_t i) const { return (f[i] & mask) == want; }
};

// ------------------------------------------------------------------------------------------
// synthetic reads (device twin of py-debruijn_amd/synth.py) + checksum
// ------------------------------------------------------------------------------------------
__device__ inline uint64_t synth_draw(uint64_t key, uint64_t ctr) { return mix64(key ^ (ctr * 0xD6E8FEB86659FD93ull)); }

__global__ __launch_bounds__(256) void k_synth(char *bases, uint64_t *offsets, uint64_t kg, uint64_t ks, uint64_t ke,
                                               uint64_t genome_len, uint64_t first_read, uint64_t n_reads,
                                               uint32_t read_len, uint32_t err_thr) {
    const uint64_t total = n_reads * read_len;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / read_len, j = i - r * read_len, gr = first_read + r;
        const uint64_t start = synth_draw(ks, gr) % (genome_len - read_len + 1);
        uint32_t code = (uint32_t)(synth_draw(kg, start + j) & 3);
        if (err_thr) {
            const uint64_t e = synth_draw(ke, gr * read_len + j);
            if ((uint32_t)(e & 0xFFFFFFu) < err_thr) code = (code + 1 + (uint32_t)((e >> 24) % 3)) & 3;
        }
        bases[i] = "ACGT"[code];
        if (j == 0) offsets[r] = i;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) offsets[n_reads] = total;
}

__global__ __launch_bounds__(256) void k_checksum(const char *bases, uint64_t n, unsigned long long *out) {
    uint64_t s = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        s += mix64((uint64_t)(uint8_t)bases[i] + 256ull * i);
    s = wave_sum_u64(s);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, (unsigned long long)s);
}

// ------------------------------------------------------------------------------------------
// read-start bitmap
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_startbits(const uint64_t *offsets, uint64_t n_reads, uint32_t *bits) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n_reads) {
        uint64_t p = offsets[i];
        atomicOr(&bits[p >> 5], 1u << (p & 31));
    }
}

// ------------------------------------------------------------------------------------------
// a3 + a4: encode, hash insert, successor counters, first-occurrence stamp
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_table_init(Slot *tab, uint64_t cap) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cap) {
        uint4 *p = reinterpret_cast<uint4 *>(tab + i);
        p[0] = make_uint4(~0u, ~0u, ~0u, ~0u);
        p[1] = make_uint4(0, 0, 0, 0);
    }
}

__global__ __launch_bounds__(256) void k_count(const char *__restrict__ bases, uint64_t n_bytes,
                                               const uint32_t *__restrict__ startbits, int k, Slot *tab,
                                               uint64_t cap_mask, int hash_shift, uint32_t *occ,
                                               unsigned long long *scalars) {
    __shared__ TileLds t;
    const uint64_t tile0 = (uint64_t)blockIdx.x * TILE;
    const uint32_t bad = load_tile(t, bases, n_bytes, startbits, tile0);
    if (bad) atomicOr(&scalars[0], 1ull);
    __syncthreads();

    const uint32_t mid_mask = (k >= 2) ? ((1u << (k - 1)) - 1u) : 0u;
    uint64_t n_k = 0, n_e = 0;
    for (int j = threadIdx.x; j < TILE; j += 256) {
        const uint64_t p = tile0 + j;
        if (p >= n_bytes) break;
        const uint32_t sw = startwin32(t, j);
        if ((sw >> 1) & mid_mask) continue;        // a read boundary inside the k-mer
        const uint32_t s0 = sw & 1u;                // k-mer sits at position 0 of its read
        const uint32_t sk = (sw >> k) & 1u;         // position p+k starts another read (or is the end)
        if (sk && s0) continue;                     // read of length exactly k: contributes nothing [:126]
        const uint64_t win = window32(t, j);
        const uint64_t kmer = win >> (64 - 2 * k);
        const uint32_t b = (uint32_t)(win >> (62 - 2 * k)) & 3u;
        const uint64_t stamp = (p << 1) | (s0 ^ 1u);
        n_k += 1;
        n_e += sk ^ 1u;

        uint64_t slot = kmer_hash(kmer) >> hash_shift;
        bool found = false;
        for (uint64_t probe = 0; probe <= cap_mask; ++probe) {
            unsigned long long cur =
                __hip_atomic_load(&tab[slot].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == EMPTY_KEY) {
                cur = atomicCAS(&tab[slot].key, EMPTY_KEY, (unsigned long long)kmer);
                if (cur == EMPTY_KEY) {
                    atomicOr(&occ[slot >> 5], 1u << (slot & 31));
                    cur = kmer;
                }
            }
            if (cur == kmer) { found = true; break; }
            slot = (slot + 1) & cap_mask;
        }
        if (!found) { atomicOr(&scalars[0], 2ull); continue; }  // table full
        if (!sk) atomicAdd(&tab[slot].cnt[b], 1u);
        atomicMin(&tab[slot].stamp, (unsigned long long)stamp);
    }
    // one update per workgroup and counter: these are same-address atomics (serialised chip-wide, ~12 ns each)
    uint64_t tot_k, tot_e;
    (void)block_exscan_256(n_k, &tot_k);
    (void)block_exscan_256(n_e, &tot_e);
    if (threadIdx.x == 0) {
        if (tot_k) atomicAdd(&scalars[1], (unsigned long long)tot_k);
        if (tot_e) atomicAdd(&scalars[2], (unsigned long long)tot_e);
    }
}

// occupied slots -> node arrays (table order); the slot's stamp field is re-used for the node id
__global__ __launch_bounds__(256) void k_gather(Slot *tab, const uint32_t *occ, const uint32_t *word_rank,
                                                uint64_t n_words, uint64_t *keys, uint64_t *stamps, uint32_t *cnt,
                                                uint8_t *flags) {
    uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    uint32_t bits = occ[w];
    uint32_t node = word_rank[w];
    while (bits) {
        const int b = __ffs(bits) - 1;
        bits &= bits - 1;
        Slot *s = tab + (w * 32 + b);
        const uint4 lo = reinterpret_cast<const uint4 *>(s)[0];
        const uint4 hi = reinterpret_cast<const uint4 *>(s)[1];
        keys[node] = ((uint64_t)lo.y << 32) | lo.x;
        const uint64_t st = ((uint64_t)lo.w << 32) | lo.z;
        stamps[node] = st;
        reinterpret_cast<uint4 *>(cnt)[node] = hi;
        flags[node] = (uint8_t)(st & 1);
        s->stamp = node;
        ++node;
    }
}

__device__ inline uint32_t tab_find(const Slot *tab, uint64_t cap_mask, int hash_shift, uint64_t key) {
    uint64_t slot = kmer_hash(key) >> hash_shift;
    for (uint64_t probe = 0; probe <= cap_mask; ++probe) {
        const uint64_t cur = tab[slot].key;
        if (cur == key) return (uint32_t)tab[slot].stamp;
        if (cur == EMPTY_KEY) return NO_NODE;
        slot = (slot + 1) & cap_mask;
    }
    return NO_NODE;
}

// successor ids (4-way) + successor rank order byte
__global__ __launch_bounds__(256) void k_succ(const Slot *tab, uint64_t cap_mask, int hash_shift, int k, uint64_t n_nodes,
                                              const uint64_t *keys, const uint32_t *cnt, uint32_t *succ,
                                              uint8_t *order, uint8_t *deg) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const uint64_t kmask = (k == 32) ? ~0ull : ((1ull << (2 * k)) - 1);
    const uint64_t key = keys[i];
    const uint4 c4 = reinterpret_cast<const uint4 *>(cnt)[i];
    const uint32_t c[4] = {c4.x, c4.y, c4.z, c4.w};
    uint32_t s[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
        s[b] = c[b] ? tab_find(tab, cap_mask, hash_shift, ((key << 2) | (uint64_t)b) & kmask) : NO_NODE;
    reinterpret_cast<uint4 *>(succ)[i] = make_uint4(s[0], s[1], s[2], s[3]);
    deg[i] = (uint8_t)((c[0] != 0) + (c[1] != 0) + (c[2] != 0) + (c[3] != 0));
    // rank codes by (count desc, ascii order asc): insertion sort of 4
    uint32_t code[4] = {0, 1, 3, 2};  // ascii order A, C, G, T as codes
#pragma unroll
    for (int a = 1; a < 4; ++a) {
#pragma unroll
        for (int b = a; b > 0; --b) {
            if (c[code[b]] > c[code[b - 1]]) {
                uint32_t tmp = code[b]; code[b] = code[b - 1]; code[b - 1] = tmp;
            }
        }
    }
    order[i] = (uint8_t)(code[0] | (code[1] << 2) | (code[2] << 4) | (code[3] << 6));
}

__global__ __launch_bounds__(256) void k_csr_fill(uint64_t n_nodes, const uint64_t *rowptr, const uint32_t *cnt,
                                                  const uint32_t *succ, uint32_t *col, uint32_t *ecnt) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const uint4 c4 = reinterpret_cast<const uint4 *>(cnt)[i];
    const uint4 s4 = reinterpret_cast<const uint4 *>(succ)[i];
    const uint32_t c[4] = {c4.x, c4.y, c4.z, c4.w}, s[4] = {s4.x, s4.y, s4.z, s4.w};
    uint64_t o = rowptr[i];
#pragma unroll
    for (int b = 0; b < 4; ++b)
        if (c[b]) { col[o] = s[b]; ecnt[o] = c[b]; ++o; }
}

// ------------------------------------------------------------------------------------------
// a5 + a6: pruningEdges + branch flag
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prune(uint64_t n_nodes, const uint32_t *cnt, const uint8_t *order, double threshold,
                                               uint8_t *flags, unsigned long long *n_branch) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t br = 0;
    if (i < n_nodes) {
        const uint4 c4 = reinterpret_cast<const uint4 *>(cnt)[i];
        const uint32_t c[4] = {c4.x, c4.y, c4.z, c4.w};
        const uint32_t mx = max(max(c[0], c[1]), max(c[2], c[3]));
        const int distinct = (c[0] != 0) + (c[1] != 0) + (c[2] != 0) + (c[3] != 0);
        uint32_t keep = 0;
        if (distinct == 1) {
            for (int b = 0; b < 4; ++b) keep |= (c[b] != 0) << b;  // debruijn.py:156-157
        } else if (distinct > 1) {
            const double lim = (double)mx / threshold;  // debruijn.py:163 (true division)
            bool top_taken = false;
            for (int r = 0; r < 4; ++r) {
                // the arg-max is kept unconditionally (:159); any other count is tested against lim.
                // Among equal maxima the reference keeps the first-seen one (first in the rank byte).
                const int b = (order[i] >> (2 * r)) & 3;  // most_common order: the first maximum is the arg-max
                if (!c[b]) continue;
                if (c[b] == mx && !top_taken) { keep |= 1u << b; top_taken = true; continue; }
                if ((double)c[b] >= lim) keep |= 1u << b;
            }
        }
        const bool branch = __popc(keep) > 1;
        br = branch;
        flags[i] = (uint8_t)((flags[i] & DBG_F_INDEG) | (keep << DBG_F_KEEP_SHIFT) | (branch ? DBG_F_BRANCH : 0));
    }
    br = wave_sum_u64(br);
    if ((threadIdx.x & 63) == 0 && br) atomicAdd(n_branch, (unsigned long long)br);
}

// ------------------------------------------------------------------------------------------
// a7: tip removal.  The reference processes branch nodes in dict order and every DFS reads
// the pulled set left by earlier ones, so the result is order dependent.  Deterministic
// reservations reproduce the sequential result in parallel: every pending branch claims
// (atomicMin of its stamp) each node within 4 steps; a branch whose claims all hold runs its
// DFS against the committed state; the others retry next round.  The lowest pending stamp
// always wins its claims, so every round makes progress.
// ------------------------------------------------------------------------------------------
constexpr int TIP_DEPTH = 5;  // debruijn.py:246

// Graph accessors: the tip / walk / jump kernels are written against this interface so that the 4-way DNA
// layout (2-bit codes, ranks and keep mask packed in bytes) and the generic layout (5-bit codes, up
// to 32 successors, dbg_generic.h) share them.
struct GDna {
    const uint64_t *keys;
    const uint64_t *keys_hi;  // k > 32 bases do not fit one word; nullptr for k <= 31
    const uint8_t *flags;
    const uint8_t *order;
    const uint32_t *succ;
    const uint32_t *cnt;
    int k;
    __device__ int n_ranks(uint32_t) const { return 4; }
    __device__ uint32_t code_at(uint32_t x, int r) const { return (order[x] >> (2 * r)) & 3u; }
    __device__ bool kept(uint32_t x, uint32_t code) const { return (flags[x] >> (DBG_F_KEEP_SHIFT + code)) & 1u; }
    __device__ uint32_t keep_count(uint32_t x) const { return __popc((uint32_t)(flags[x] & DBG_F_KEEP_MASK)); }
    __device__ uint32_t first_kept(uint32_t x) const { return __ffs((uint32_t)(flags[x] & DBG_F_KEEP_MASK) >> DBG_F_KEEP_SHIFT) - 1; }
    __device__ uint32_t succ_of(uint32_t x, uint32_t code) const { return succ[(uint64_t)x * 4 + code]; }
    __device__ uint32_t cnt_of(uint32_t x, uint32_t code) const { return cnt[(uint64_t)x * 4 + code]; }
    __device__ bool terminal(uint32_t x) const {  // pre-pruning outdegree == 0 [:173]
        const uint4 c = reinterpret_cast<const uint4 *>(cnt)[x];
        return (c.x | c.y | c.z | c.w) == 0;
    }
    __device__ uint32_t last_code(uint32_t x) const { return (uint32_t)(keys[x] & 3u); }
    __device__ char sym_char(uint32_t code) const { return code_to_ascii(code); }
    __device__ char char_at(uint32_t x, int q) const {  // character q of node x's k-mer
        const int sh = 2 * (k - 1 - q);
        return code_to_ascii((uint32_t)(sh >= 64 ? keys_hi[x] >> (sh - 64) : keys[x] >> sh) & 3u);
    }
    __device__ void spell(uint32_t x, char *out) const {
        const uint64_t key = keys[x], hi = keys_hi ? keys_hi[x] : 0ull;
        for (int q = 0; q < k; ++q) {
            const int sh = 2 * (k - 1 - q);
            out[q] = code_to_ascii((uint32_t)(sh >= 64 ? hi >> (sh - 64) : key >> sh) & 3u);
        }
    }
};

// DFS below `root` over surviving edges, depth-limited like debruijn.py:169-186.
// visit(x): every node entered at levels 1..4; term(path, len): a path root..t ending in a terminal node.
template <bool CHECK_PULLED, class G, class Visit, class Term>
__device__ inline void tip_dfs(const G &g, uint32_t root, Visit visit, Term term) {
    uint32_t path[TIP_DEPTH];
    int idx[TIP_DEPTH];
    path[0] = root;
    idx[0] = 0;
    int d = 0;
    while (d >= 0) {
        const uint32_t cur = path[d];
        if (idx[d] >= g.n_ranks(cur)) { --d; continue; }
        const int r = idx[d]++;
        const uint32_t code = g.code_at(cur, r);
        if (!g.kept(cur, code)) continue;
        const uint32_t s = g.succ_of(cur, code);
        if (CHECK_PULLED && (g.flags[s] & DBG_F_PULLED)) continue;
        if (d + 1 >= TIP_DEPTH) continue;  // child would be entered with depth == 0
        visit(s);
        path[d + 1] = s;
        if (g.terminal(s)) { term(path, d + 2); continue; }
        idx[d + 1] = 0;
        ++d;
    }
}

template <class G>
__global__ __launch_bounds__(256) void k_tip_reset(G g, const uint32_t *pending, uint64_t n_pending,
                                                   unsigned long long *owner) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pending) return;
    tip_dfs<false>(g, pending[i], [&](uint32_t x) { owner[x] = ~0ull; }, [](const uint32_t *, int) {});
}

template <class G>
__global__ __launch_bounds__(256) void k_tip_claim(G g, const uint32_t *pending, uint64_t n_pending,
                                                   const uint64_t *stamps, unsigned long long *owner) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pending) return;
    const uint32_t b = pending[i];
    const unsigned long long me = stamps[b];
    tip_dfs<false>(g, b, [&](uint32_t x) { atomicMin(&owner[x], me); }, [](const uint32_t *, int) {});
}

template <class G>
__global__ __launch_bounds__(256) void k_tip_commit(G g, uint8_t *flags_rw, const uint32_t *pending,
                                                    uint64_t n_pending, const uint64_t *stamps,
                                                    const unsigned long long *owner, unsigned long long *pull_rank,
                                                    uint32_t *next_pending, unsigned long long *counters) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pending) return;
    const uint32_t b = pending[i];
    const unsigned long long me = stamps[b];
    bool mine = true;
    tip_dfs<false>(g, b, [&](uint32_t x) { mine = mine && (owner[x] == me); }, [](const uint32_t *, int) {});
    if (!mine) {
        const unsigned long long slot = atomicAdd(&counters[0], 1ull);
        next_pending[slot] = b;
        return;
    }
    // exclusive owner of everything within reach: run the reference DFS against the committed state
    unsigned long long j = 0;
    tip_dfs<true>(
        g, b, [](uint32_t) {},
        [&](const uint32_t *path, int len) {
            for (int q = 1; q < len; ++q) {  // path[0] is the branch node itself, never pulled [:251]
                const uint32_t x = path[q];
                if ((g.flags[x] & (DBG_F_PULLED | DBG_F_BRANCH)) == 0 && pull_rank[x] == ~0ull) {
                    pull_rank[x] = (me << 22) + j;  // pull order: branch (dict order), then discovery order
                    ++j;
                }
            }
        });
    if (j) {
        tip_dfs<false>(
            g, b,
            [&](uint32_t x) {
                if (pull_rank[x] != ~0ull) flags_rw[x] |= DBG_F_PULLED;
            },
            [](const uint32_t *, int) {});
        atomicAdd(&counters[1], j);
    }
}

// ---- ordered compaction: ids i < n with pred(i), ascending.  Two streaming passes over contiguous chunks (count, then
//      write at the chunk's offset) -- one shared cursor would cost a same-address atomic (~12 ns, serialised chip-wide)
//      per wave: 65 ms for the 12 M splitters of a 3.6e8-node graph.
constexpr unsigned COMPACT_WGS = 4096;
template <class Pred>
__global__ __launch_bounds__(256) void k_compact_count(uint64_t n, Pred pred, uint64_t *wg_count) {
    const uint64_t beg = n * blockIdx.x / gridDim.x, end = n * (blockIdx.x + 1) / gridDim.x;
    uint64_t c = 0;
    for (uint64_t i = beg + threadIdx.x; i < end; i += 256) c += pred(i) ? 1 : 0;
    uint64_t tot;
    (void)block_exscan_256(c, &tot);
    if (threadIdx.x == 0) wg_count[blockIdx.x] = tot;
}
template <class Pred>
__global__ __launch_bounds__(256) void k_compact_write(uint64_t n, Pred pred, const uint64_t *__restrict__ wg_offset, uint32_t *out) {
    const uint64_t beg = n * blockIdx.x / gridDim.x, end = n * (blockIdx.x + 1) / gridDim.x;
    uint64_t base = wg_offset[blockIdx.x];
    for (uint64_t i0 = beg; i0 < end; i0 += 256) {  // uniform trip count
        const uint64_t i = i0 + threadIdx.x;
        const bool p = i < end && pred(i);
        uint64_t tot;
        const uint64_t ex = block_exscan_256(p ? 1 : 0, &tot);
        if (p) out[base + ex] = (uint32_t)i;
        base += tot;
    }
}
struct PredFlags {
    const uint8_t *flags;
    uint8_t mask, want;
    __device__ bool operator()(uint64_t i) const { return (flags[i] & mask) == want; }
};
// out (capacity >= the number of hits; nullptr: count only) receives the ids in ascending order
template <class Pred>
static int compact_ids(dbg *h, uint64_t n, Pred pred, uint32_t *out, uint64_t *h_total) {
    const unsigned g = (unsigned)std::min<uint64_t>(COMPACT_WGS, std::max<uint64_t>(1, (n + 255) / 256));
    CHK(buf_ensure(h, h->ar_scan, (uint64_t)(g + 1) * 8));
    uint64_t *cnt = (uint64_t *)h->ar_scan.p, *d_total = cnt + g;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_compact_count<Pred>), dim3(g), dim3(256), 0, h->stream, n, pred, cnt);
    hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(256), 0, h->stream, cnt, (uint64_t)g, d_total);
    if (out) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_compact_write<Pred>), dim3(g), dim3(256), 0, h->stream, n, pred, cnt, out);
    hipError_t e = hipMemcpyAsync(h_total, d_total, 8, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) { h->err = std::string("compact_ids: ") + hipGetErrorString(e); return DBG_E_HIP; }
    return DBG_OK;
}

template <class T>
__global__ __launch_bounds__(256) void k_fill(T *p, uint64_t n, T v) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ------------------------------------------------------------------------------------------
// a9: reads that contain a branch k-mer as a substring (reads of length == k included)
// ------------------------------------------------------------------------------------------
// open-address set of the branch k-mers (few): the only lookup structure a9 needs
__global__ __launch_bounds__(256) void k_branch_insert(uint64_t n_nodes, const uint8_t *flags, const uint64_t *keys,
                                                       unsigned long long *btab, uint64_t cap_mask) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes || !(flags[i] & DBG_F_BRANCH)) return;
    const unsigned long long key = keys[i];
    uint64_t slot = kmer_hash(key) & cap_mask;
    for (uint64_t probe = 0; probe <= cap_mask; ++probe) {
        const unsigned long long cur = atomicCAS(&btab[slot], EMPTY_KEY, key);
        if (cur == EMPTY_KEY || cur == key) return;
        slot = (slot + 1) & cap_mask;
    }
}

__global__ __launch_bounds__(256) void k_pull_reads(const char *__restrict__ bases, uint64_t n_bytes,
                                                    const uint32_t *__restrict__ startbits, int k,
                                                    const uint64_t *__restrict__ btab, uint64_t cap_mask,
                                                    const uint64_t *offsets, uint64_t n_reads, uint8_t *read_flags) {
    __shared__ TileLds t;
    const uint64_t tile0 = (uint64_t)blockIdx.x * TILE;
    (void)load_tile(t, bases, n_bytes, startbits, tile0);
    __syncthreads();
    const uint32_t mid_mask = (k >= 2) ? ((1u << (k - 1)) - 1u) : 0u;
    for (int j = threadIdx.x; j < TILE; j += 256) {
        const uint64_t p = tile0 + j;
        if (p + k > n_bytes) break;
        const uint32_t sw = startwin32(t, j);
        if ((sw >> 1) & mid_mask) continue;
        const uint64_t kmer = window32(t, j) >> (64 - 2 * k);
        uint64_t slot = kmer_hash(kmer) & cap_mask;
        bool hit = false;
        for (uint64_t probe = 0; probe <= cap_mask; ++probe) {
            const uint64_t cur = btab[slot];
            if (cur == kmer) { hit = true; break; }
            if (cur == EMPTY_KEY) break;
            slot = (slot + 1) & cap_mask;
        }
        if (!hit) continue;
        // read index: last r with offsets[r] <= p (skipping empty reads that share the offset)
        uint64_t lo = 0, hi = n_reads;  // offsets[lo] <= p < offsets[hi]
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (offsets[mid] <= p) lo = mid; else hi = mid;
        }
        read_flags[lo] = 1;
    }
}

constexpr int RF_SLOTS = 1024;  // LDS set of the per-range kernels (k_sk_pull, k_sk_refine): at most half full
// a9 for the partitioned build: per range, the range's branch nodes (usually none or one) go into an LDS set and the
// bucket's own super-k-mer records are expanded against it; a hit marks the read that holds the instance.  Buckets
// without a branch node are skipped, nothing streams the reads.  Reads of length exactly k have no record (they
// contribute nothing to the graph, debruijn.py:126) but can still contain a branch k-mer: k_pull_len_k.
template <class ST>
__global__ __launch_bounds__(256) void k_sk_pull(const SkRange *__restrict__ ranges, const uint64_t *__restrict__ b_start,
                                                 const uint64_t *__restrict__ b_cnt, const uint64_t *__restrict__ rec_w0,
                                                 const uint64_t *__restrict__ rec_w1, const ST *__restrict__ rec_st, int k,
                                                 const uint64_t *__restrict__ keys, const uint8_t *__restrict__ flags,
                                                 const uint64_t *__restrict__ offsets, uint64_t n_reads, uint8_t *read_flags,
                                                 unsigned long long *flag) {
    __shared__ unsigned long long skey[RF_SLOTS];
    __shared__ uint32_t n_branch;
    const SkRange rg = ranges[blockIdx.x];
    if (rg.node_cnt == 0) return;
    for (int i = threadIdx.x; i < RF_SLOTS; i += 256) skey[i] = EMPTY_KEY;
    if (threadIdx.x == 0) n_branch = 0;
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < rg.node_cnt; j += 256) {
        const uint64_t node = rg.node_base + j;
        if (!(flags[node] & DBG_F_BRANCH)) continue;
        if (atomicAdd(&n_branch, 1u) >= RF_SLOTS / 2) continue;  // reported below
        const unsigned long long key = keys[node];
        uint32_t slot = slot_hash(key) >> 22;
        while (atomicCAS(&skey[slot], EMPTY_KEY, key) != EMPTY_KEY) slot = (slot + 1) & (RF_SLOTS - 1);
    }
    __syncthreads();
    const uint32_t nb = n_branch;
    if (nb == 0) return;
    if (nb > RF_SLOTS / 2) { if (threadIdx.x == 0) atomicOr(flag, 1ull); return; }
    const uint64_t r_beg = b_start[rg.bucket], r_n = b_cnt[rg.bucket];
    for (uint64_t r = threadIdx.x; r < r_n; r += 256) {
        const uint64_t w0 = rec_w0[r_beg + r], w1 = rec_w1[r_beg + r];
        const int len = (int)((w1 >> 1) & 31) + 1;
        const uint64_t hi = w1 & (~0ull << SK_META_BITS);
        for (int i = 0; i < len; ++i) {
            const unsigned long long kmer = rec_window(w0, hi, i) >> (64 - 2 * k);
            uint32_t slot = slot_hash(kmer) >> 22;
            bool hit = false;
            for (;;) {
                const unsigned long long cur = skey[slot];
                if (cur == EMPTY_KEY) break;
                if (cur == kmer) { hit = true; break; }
                slot = (slot + 1) & (RF_SLOTS - 1);
            }
            if (!hit) continue;
            const uint64_t p = (uint64_t)(rec_st[r_beg + r] >> 1) + (uint64_t)i;  // byte position of the instance
            uint64_t lo = 0, up = n_reads;  // offsets[lo] <= p < offsets[up]
            while (up - lo > 1) {
                const uint64_t mid = (lo + up) >> 1;
                if (offsets[mid] <= p) lo = mid; else up = mid;
            }
            read_flags[lo] = 1;
        }
    }
}

// reads of length exactly k against the global branch set
__global__ __launch_bounds__(256) void k_pull_len_k(const char *__restrict__ bases, const uint64_t *__restrict__ offsets,
                                                    uint64_t n_reads, int k, const uint64_t *__restrict__ btab, uint64_t cap_mask,
                                                    uint8_t *read_flags) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads || offsets[r + 1] - offsets[r] != (uint64_t)k) return;
    uint64_t kmer = 0;
    for (int i = 0; i < k; ++i) kmer = (kmer << 2) | (((uint64_t)(uint8_t)bases[offsets[r] + i] >> 1) & 3ull);
    uint64_t slot = kmer_hash(kmer) & cap_mask;
    for (uint64_t probe = 0; probe <= cap_mask; ++probe) {
        const uint64_t cur = btab[slot];
        if (cur == kmer) { read_flags[r] = 1; return; }
        if (cur == EMPTY_KEY) return;
        slot = (slot + 1) & cap_mask;
    }
}

struct ByteAt {
    const uint8_t *p;
    __device__ uint64_t operator()(uint64_t i) const { return p[i] != 0; }
};

// ------------------------------------------------------------------------------------------
// a11 + a12: contig walk.  PASS 0 counts (contigs, chars) per start, PASS 1 writes them.
// ------------------------------------------------------------------------------------------
// non-final mode: each start yields at most one contig, a chain walk that stops at a branch
// node / dead end (inclusive) or before a pulled node; a chain that closes on itself yields
// nothing (debruijn.py:289-290).  Cycle detection by Brent's algorithm (no per-start memory).
template <class G>
__global__ __launch_bounds__(256) void k_walk_chain(G g, const uint32_t *starts, uint64_t n_starts, int pass,
                                                    uint64_t *ctg_per_start, uint64_t *chars_per_start,
                                                    const uint64_t *ctg_base, const uint64_t *char_base,
                                                    uint64_t *ctg_off, char *chars, uint64_t *score_out,
                                                    uint64_t *stamp_out, uint32_t *seq_out, const uint64_t *stamps) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_starts) return;
    const uint32_t s = starts[i];
    if (pass == 0) { ctg_per_start[i] = 0; chars_per_start[i] = 0; }
    if (g.flags[s] & DBG_F_PULLED) return;  // [:292-295]
    if (pass == 1 && ctg_per_start[i] == 0) return;
    char *out = nullptr;
    if (pass == 1) {
        out = chars + char_base[i];
        g.spell(s, out);
        out += g.k;
    }
    uint32_t cur = s, tortoise = s;
    uint64_t len = 1, score = 0, power = 1, lam = 0;
    bool emit = false;
    while (true) {
        const uint8_t f = g.flags[cur];
        if ((f & DBG_F_BRANCH) || g.keep_count(cur) == 0) { emit = true; break; }  // [:304-313]
        const uint32_t code = g.first_kept(cur);
        const uint32_t nxt = g.succ_of(cur, code);
        if (nxt == tortoise) break;  // revisit: nothing emitted [:289-290]
        if (g.flags[nxt] & DBG_F_PULLED) { emit = true; break; }  // path before the pulled node [:292-303]
        score += g.cnt_of(cur, code);
        if (pass == 1) *out++ = g.sym_char(code);
        cur = nxt;
        ++len;
        if (++lam == power) { tortoise = cur; power <<= 1; lam = 0; }
    }
    if (pass == 0) {
        if (emit) { ctg_per_start[i] = 1; chars_per_start[i] = g.k + len - 1; }
    } else {
        const uint64_t c = ctg_base[i];
        ctg_off[c] = char_base[i];
        score_out[c] = score;
        stamp_out[c] = stamps[s];
        seq_out[c] = 0;
    }
}

// The tortoise check above only catches a revisit of the tortoise itself; a walk that ends
// (emit) before the hare meets it is a plain chain.  A rho-shaped walk never ends and is caught
// once the tortoise sits on the cycle.  But a start that lies ON the path again (cycle through
// the start) is the same case.  Nothing else can revisit: chain nodes have one successor.

// final mode: every simple path from a start (debruijn.py:288-316 with branch_kmer == []).
// One walker per thread; walker w owns stack slices of `stride` entries and a bitmap slice.
template <class G>
__global__ __launch_bounds__(64) void k_walk_dfs(G g, const uint32_t *starts, uint64_t n_starts, int pass,
                                                 uint64_t n_walkers, uint64_t stride, uint32_t *st_node,
                                                 uint8_t *st_next, uint32_t *onpath, uint64_t bm_words,
                                                 uint64_t *ctg_per_start, uint64_t *chars_per_start,
                                                 const uint64_t *ctg_base, const uint64_t *char_base, uint64_t *ctg_off,
                                                 char *chars, uint64_t *score_out, uint64_t *stamp_out,
                                                 uint32_t *seq_out, const uint64_t *stamps, bool stop_at_branch) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_walkers) return;
    uint32_t *path = st_node + w * stride;
    uint8_t *nxt = st_next + w * stride;  // low 6 bits: next rank to try, bit 7: prefix already emitted
    uint32_t *bm = onpath + w * bm_words;
    for (uint64_t i = w; i < n_starts; i += n_walkers) {
        uint64_t n_ctg = 0, n_chr = 0;
        uint64_t c_out = pass ? ctg_base[i] : 0, ch_out = pass ? char_base[i] : 0;
        const uint32_t s0 = starts[i];
        int64_t depth = 0;  // nodes on the path
        auto emit = [&](int64_t len, uint32_t extra, bool has_extra) {
            const uint64_t nn = (uint64_t)len + (has_extra ? 1 : 0);
            const uint64_t nch = g.k + nn - 1;
            if (pass == 1) {
                char *out = chars + ch_out;
                g.spell(len > 0 ? path[0] : extra, out);
                out += g.k;
                uint64_t score = 0;
                uint32_t prev = len > 0 ? path[0] : extra;
                for (uint64_t q = 1; q < nn; ++q) {
                    const uint32_t x = (q < (uint64_t)len) ? path[q] : extra;
                    const uint32_t code = g.last_code(x);
                    *out++ = g.sym_char(code);
                    score += g.cnt_of(prev, code);
                    prev = x;
                }
                ctg_off[c_out] = ch_out;
                score_out[c_out] = score;
                stamp_out[c_out] = stamps[s0];
                seq_out[c_out] = (uint32_t)n_ctg;
                ++c_out;
                ch_out += nch;
            }
            ++n_ctg;
            n_chr += nch;
        };
        // enter(node): returns true when pushed
        auto enter = [&](uint32_t x) -> bool {
            if ((bm[x >> 5] >> (x & 31)) & 1u) return false;  // on the current path [:289]
            const uint8_t f = g.flags[x];
            if (f & DBG_F_PULLED) {  // [:292-303]
                if (depth > 0 && !(nxt[depth - 1] & 0x80)) {
                    nxt[depth - 1] |= 0x80;
                    emit(depth, 0, false);
                }
                return false;
            }
            if ((stop_at_branch && (f & DBG_F_BRANCH)) || g.keep_count(x) == 0) {  // [:304-313]
                emit(depth, x, true);
                return false;
            }
            path[depth] = x;
            nxt[depth] = 0;
            bm[x >> 5] |= 1u << (x & 31);
            ++depth;
            return true;
        };
        enter(s0);
        while (depth > 0) {
            const uint32_t cur = path[depth - 1];
            const int r = nxt[depth - 1] & 0x3F;
            if (r >= g.n_ranks(cur)) {
                bm[cur >> 5] &= ~(1u << (cur & 31));
                --depth;
                continue;
            }
            nxt[depth - 1] = (uint8_t)((nxt[depth - 1] & 0x80) | (r + 1));
            const uint32_t code = g.code_at(cur, r);
            if (!g.kept(cur, code)) continue;
            enter(g.succ_of(cur, code));
        }
        if (pass == 0) { ctg_per_start[i] = n_ctg; chars_per_start[i] = n_chr; }
    }
}

// ------------------------------------------------------------------------------------------
// FASTA ingest on the device (read_reads, debruijn.py:22-32): every line that does not start with
// '>' is one read, rstrip'ed; multi-line records are separate reads; a blank line is an empty read.
// Line terminators follow Python's universal-newline text mode ('\n', '\r\n' and a lone '\r').
// ------------------------------------------------------------------------------------------
__device__ inline bool fa_line_start(const char *text, uint64_t i) {
    if (i == 0) return true;
    const char prev = text[i - 1];
    return prev == '\n' || (prev == '\r' && text[i] != '\n');
}
__device__ inline bool fa_space(char c) {  // str.isspace() over the ASCII range
    return c == ' ' || (c >= 9 && c <= 13) || (c >= 28 && c <= 31);
}

__global__ __launch_bounds__(256) void k_fa_mark(const char *__restrict__ text, uint64_t n, uint32_t *bits) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 32-bit word of the bitmap per thread
    const uint64_t base = w * 32;
    if (base >= n) return;
    uint32_t m = 0;
    for (int b = 0; b < 32 && base + b < n; ++b) m |= (uint32_t)fa_line_start(text, base + b) << b;
    bits[w] = m;
}

__global__ __launch_bounds__(256) void k_fa_line_starts(const uint32_t *bits, const uint32_t *word_rank, uint64_t n_words,
                                                        uint64_t *line_start) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    uint32_t m = bits[w], r = word_rank[w];
    while (m) {
        const int b = __ffs(m) - 1;
        m &= m - 1;
        line_start[r++] = w * 32 + b;
    }
}

// per line: is it a read (not a header), and how long after rstrip
__global__ __launch_bounds__(256) void k_fa_lines(const char *__restrict__ text, uint64_t n, const uint64_t *line_start,
                                                  uint64_t n_lines, uint8_t *is_read, uint32_t *len) {
    const uint64_t l = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= n_lines) return;
    const uint64_t s0 = line_start[l];
    uint64_t e = (l + 1 < n_lines) ? line_start[l + 1] : n;
    const bool rd = text[s0] != '>';
    while (e > s0 && fa_space(text[e - 1])) --e;  // terminator + trailing white space
    is_read[l] = rd;
    len[l] = rd ? (uint32_t)(e - s0) : 0;
}

struct ByteSet {
    const uint8_t *p;
    __device__ uint64_t operator()(uint64_t i) const { return p[i]; }
};
struct U32At {
    const uint32_t *p;
    __device__ uint64_t operator()(uint64_t i) const { return p[i]; }
};

__global__ __launch_bounds__(256) void k_fa_copy(const char *__restrict__ text, const uint64_t *line_start,
                                                 uint64_t n_lines, const uint8_t *is_read, const uint32_t *len,
                                                 const uint64_t *read_idx, const uint64_t *byte_off, char *bases,
                                                 uint64_t *offsets) {
    // 16 lanes per line: consecutive lanes copy consecutive bytes (a thread per line wrote 150 bytes one by one,
    // each a separate uncoalesced access: 29 ms for 1.5 GB)
    const uint64_t l = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const uint32_t sub = threadIdx.x & 15u;
    if (l >= n_lines || !is_read[l]) return;
    const uint64_t src = line_start[l], dst = byte_off[l];
    if (sub == 0) offsets[read_idx[l]] = dst;
    const uint32_t n = len[l];
    for (uint32_t i = sub; i < n; i += 16) bases[dst + i] = text[src + i];
}

// ------------------------------------------------------------------------------------------
// exact successor order.  The reference lists the successors of a vertex in Counter order:
// first-seen order for the edge-count table (debruijn.py:215-216) and count-descending with
// first-seen ties after pruning (Counter.most_common, :159-165).  The build only keeps the first
// occurrence of NODES; for the few nodes with two or more distinct successors this pass finds the
// first occurrence of every out-edge by streaming the k-mer instances once more, and rewrites the
// rank bytes: order[] = (count desc, first-seen asc), fsorder[] = first-seen asc.
// ------------------------------------------------------------------------------------------
// `filter`: one bit per top-`fbits` hash value of a member.  The streaming pass below tests it first: the set itself
// (16 bytes per slot, > 100 MB at scale) is a random HBM access per k-mer instance, the filter (16 MB) stays in cache
// and rejects ~95 % of the instances.
__global__ __launch_bounds__(256) void k_multi_insert(uint64_t n_nodes, const uint32_t *cnt, const uint64_t *keys,
                                                      unsigned long long *set_keys, uint32_t *set_node, uint64_t cap_mask,
                                                      uint32_t *filter, int fbits) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const uint4 c = reinterpret_cast<const uint4 *>(cnt)[i];
    if ((c.x != 0) + (c.y != 0) + (c.z != 0) + (c.w != 0) < 2) return;
    const unsigned long long key = keys[i];
    const uint64_t hv = kmer_hash(key);
    const uint64_t fb = hv >> (64 - fbits);
    atomicOr(&filter[fb >> 5], 1u << (fb & 31));
    uint64_t slot = hv & cap_mask;
    for (uint64_t probe = 0; probe <= cap_mask; ++probe) {
        const unsigned long long cur = atomicCAS(&set_keys[slot], EMPTY_KEY, key);
        if (cur == EMPTY_KEY) { set_node[slot] = (uint32_t)i; return; }
        slot = (slot + 1) & cap_mask;
    }
}

__global__ __launch_bounds__(256) void k_edge_first_seen(const char *__restrict__ bases, uint64_t n_bytes,
                                                         const uint32_t *__restrict__ startbits, int k,
                                                         const uint64_t *__restrict__ set_keys, uint64_t cap_mask,
                                                         const uint32_t *__restrict__ filter, int fbits,
                                                         unsigned long long *estamp /* [slot * 4 + code] */) {
    __shared__ TileLds t;
    const uint64_t tile0 = (uint64_t)blockIdx.x * TILE;
    (void)load_tile(t, bases, n_bytes, startbits, tile0);
    __syncthreads();
    const uint32_t mid_mask = (k >= 2) ? ((1u << (k - 1)) - 1u) : 0u;
    for (int j = threadIdx.x; j < TILE; j += 256) {
        const uint64_t p = tile0 + j;
        if (p + k >= n_bytes) break;  // needs a successor base
        const uint32_t sw = startwin32(t, j);
        if (((sw >> 1) & mid_mask) || ((sw >> k) & 1u)) continue;  // k-mer or its successor crosses a read boundary
        const uint64_t win = window32(t, j);
        const uint64_t kmer = win >> (64 - 2 * k);
        const uint32_t b = (uint32_t)(win >> (62 - 2 * k)) & 3u;
        const uint64_t hv = kmer_hash(kmer);
        const uint64_t fb = hv >> (64 - fbits);
        if (!((filter[fb >> 5] >> (fb & 31)) & 1u)) continue;  // not a multi-successor node
        uint64_t slot = hv & cap_mask;
        for (uint64_t probe = 0; probe <= cap_mask; ++probe) {
            const uint64_t cur = set_keys[slot];
            if (cur == kmer) { atomicMin(&estamp[slot * 4 + b], (unsigned long long)p); break; }
            if (cur == EMPTY_KEY) break;
            slot = (slot + 1) & cap_mask;
        }
    }
}

__device__ inline void refine_node(uint64_t slot, uint32_t node, const unsigned long long *estamp, const uint32_t *cnt,
                                   uint8_t *order, uint8_t *fsorder) {
    const uint4 c4 = reinterpret_cast<const uint4 *>(cnt)[node];
    const uint32_t c[4] = {c4.x, c4.y, c4.z, c4.w};
    unsigned long long st[4];
    for (int b = 0; b < 4; ++b) st[b] = c[b] ? estamp[slot * 4 + b] : ~0ull;
    uint32_t a[4] = {0, 1, 2, 3}, f[4] = {0, 1, 2, 3};
    for (int x = 1; x < 4; ++x)
        for (int y = x; y > 0; --y) {
            // most_common order: count descending, ties by first appearance
            const bool swap_a = c[a[y]] > c[a[y - 1]] || (c[a[y]] == c[a[y - 1]] && st[a[y]] < st[a[y - 1]]);
            if (swap_a) { uint32_t tmp = a[y]; a[y] = a[y - 1]; a[y - 1] = tmp; }
            if (st[f[y]] < st[f[y - 1]]) { uint32_t tmp = f[y]; f[y] = f[y - 1]; f[y - 1] = tmp; }
        }
    order[node] = (uint8_t)(a[0] | (a[1] << 2) | (a[2] << 4) | (a[3] << 6));
    fsorder[node] = (uint8_t)(f[0] | (f[1] << 2) | (f[2] << 4) | (f[3] << 6));
}

// The same per range of the partitioned build: the multi-successor nodes of one range (a few dozen) go into an LDS set,
// the bucket's own super-k-mer records are expanded against it, LDS atomicMin keeps the first stamp of every out-edge.
// No pass over the reads, no global set: 30 ms -> a few ms at 10 M reads.  A range with more than RF_SLOTS / 2 such nodes
// (low-complexity input) reports it and the caller takes the streaming path for the whole graph.
template <class ST>
__global__ __launch_bounds__(256) void k_sk_refine(const SkRange *__restrict__ ranges, const uint64_t *__restrict__ b_start,
                                                   const uint64_t *__restrict__ b_cnt, const uint64_t *__restrict__ rec_w0,
                                                   const uint64_t *__restrict__ rec_w1, const ST *__restrict__ rec_st, int k,
                                                   const uint64_t *__restrict__ keys, const uint32_t *__restrict__ cnt,
                                                   uint8_t *order, uint8_t *fsorder, unsigned long long *flag) {
    __shared__ unsigned long long skey[RF_SLOTS];
    __shared__ uint32_t snode[RF_SLOTS];
    __shared__ ST sest[RF_SLOTS * 4];
    __shared__ uint32_t n_multi;
    const SkRange rg = ranges[blockIdx.x];
    if (rg.node_cnt == 0) return;
    for (int i = threadIdx.x; i < RF_SLOTS; i += 256) skey[i] = EMPTY_KEY;
    for (int i = threadIdx.x; i < RF_SLOTS * 4; i += 256) sest[i] = (ST)~(ST)0;
    if (threadIdx.x == 0) n_multi = 0;
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < rg.node_cnt; j += 256) {
        const uint64_t node = rg.node_base + j;
        const uint4 c = reinterpret_cast<const uint4 *>(cnt)[node];
        if ((c.x != 0) + (c.y != 0) + (c.z != 0) + (c.w != 0) < 2) continue;
        if (atomicAdd(&n_multi, 1u) >= RF_SLOTS / 2) continue;  // reported below
        const unsigned long long key = keys[node];
        uint32_t slot = slot_hash(key) >> 22;
        for (;;) {  // at most half full: a free slot exists
            if (atomicCAS(&skey[slot], EMPTY_KEY, key) == EMPTY_KEY) { snode[slot] = (uint32_t)node; break; }
            slot = (slot + 1) & (RF_SLOTS - 1);
        }
    }
    __syncthreads();
    const uint32_t nm = n_multi;
    if (nm == 0) return;
    if (nm > RF_SLOTS / 2) { if (threadIdx.x == 0) atomicOr(flag, 1ull); return; }
    const uint64_t r_beg = b_start[rg.bucket], r_n = b_cnt[rg.bucket];
    for (uint64_t r = threadIdx.x; r < r_n; r += 256) {
        const uint64_t w0 = rec_w0[r_beg + r], w1 = rec_w1[r_beg + r];
        const ST st0 = rec_st[r_beg + r];
        const int len = (int)((w1 >> 1) & 31) + 1;
        const uint64_t hi = w1 & (~0ull << SK_META_BITS);
        const int n_edges = len - 1 + (int)(w1 & 1);  // the last k-mer of the record has a successor only if the flag says so
        for (int i = 0; i < n_edges; ++i) {
            const uint64_t win = rec_window(w0, hi, i);
            const unsigned long long kmer = win >> (64 - 2 * k);
            uint32_t slot = slot_hash(kmer) >> 22;
            for (;;) {
                const unsigned long long cur = skey[slot];
                if (cur == EMPTY_KEY) break;
                if (cur == kmer) {
                    const uint32_t b = (uint32_t)(win >> (62 - 2 * k)) & 3u;
                    atomicMin(&sest[slot * 4 + b], i ? (ST)((st0 | (ST)1) + (ST)(2 * i)) : st0);
                    break;
                }
                slot = (slot + 1) & (RF_SLOTS - 1);
            }
        }
    }
    __syncthreads();
    for (int slot = threadIdx.x; slot < RF_SLOTS; slot += 256) {
        if (skey[slot] == EMPTY_KEY) continue;
        unsigned long long est[4];
        for (int b = 0; b < 4; ++b) est[b] = (unsigned long long)sest[slot * 4 + b];
        refine_node(0, snode[slot], est, cnt, order, fsorder);
    }
}

// set_keys == nullptr: the set holds node ids only (two-word k-mers), NO_NODE marks a free slot
__global__ __launch_bounds__(256) void k_order_refine(uint64_t cap, const unsigned long long *set_keys,
                                                      const uint32_t *set_node, const unsigned long long *estamp,
                                                      const uint32_t *cnt, uint8_t *order, uint8_t *fsorder) {
    uint64_t slot = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= cap) return;
    if (set_keys ? set_keys[slot] == EMPTY_KEY : set_node[slot] == NO_NODE) return;
    refine_node(slot, set_node[slot], estamp, cnt, order, fsorder);
}

__global__ __launch_bounds__(256) void k_fsorder_default(uint64_t n_nodes, const uint8_t *order, uint8_t *fsorder) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_nodes) fsorder[i] = order[i];  // at most one successor: any order of the empty codes will do
}

struct MultiSucc {
    const uint32_t *cnt;
    __device__ uint64_t operator()(uint64_t i) const {
        const uint4 c = reinterpret_cast<const uint4 *>(cnt)[i];
        return (c.x != 0) + (c.y != 0) + (c.z != 0) + (c.w != 0) >= 2;
    }
};

// ------------------------------------------------------------------------------------------
// non-final walk at scale: pointer jumping.  Every start walks its chain to the next branch node /
// dead end / pulled node, so contigs overlap massively (10M x 150 bp: 3.0e6 starts, 1.9e12 chain
// steps in total).  Doubling resolves (end node, hops, score) of EVERY node's chain in
// ceil(log2 n) rounds; a chain that never terminates is a cycle and emits nothing (debruijn.py:289-290).
// ------------------------------------------------------------------------------------------
struct alignas(16) Jump {      // 16 bytes: one 128-bit load per (random) lookup, one 32-byte sector
    uint32_t target;
    uint32_t hops;             // <= number of nodes
    unsigned long long score;  // bit 63: chain is resolved (target is its last node); below: sum of edge counts over the
                               // hops (getScore, II_assembleFromReads.py:14-18)
};
constexpr unsigned long long JUMP_TERM = 1ull << 63;
constexpr uint32_t JUMP_START = 1u << 31;  // in hops of the ONE-STEP table (k_jump_init) only: the node has indegree 0
static_assert(sizeof(Jump) == 16, "jump table entry");

template <class G>
__global__ __launch_bounds__(256) void k_jump_init(uint64_t n_nodes, G g, Jump *J) {
    uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_nodes) return;
    const uint8_t f = g.flags[v];
    Jump j{(uint32_t)v, 0u, JUMP_TERM};
    if (!(f & (DBG_F_PULLED | DBG_F_BRANCH)) && g.keep_count((uint32_t)v)) {  // chain node: exactly one surviving successor
        const uint32_t code = g.first_kept((uint32_t)v);
        const uint32_t nxt = g.succ_of((uint32_t)v, code);
        if (!(g.flags[nxt] & DBG_F_PULLED)) { j.target = nxt; j.hops = 1; j.score = g.cnt_of((uint32_t)v, code); }
    }
    if (!(f & DBG_F_INDEG)) j.hops |= JUMP_START;  // one-step table only: lets the splitter walk test "is a start" without flags[]
    J[v] = j;
}

// ---- list ranking by splitters: O(n) chain steps instead of O(n log n).  Splitters = every node whose chain ends
// in itself (branch node, dead end, before a pulled node), every start, and a pseudo-random 1/64 of the rest.
// Each splitter walks its chain to the next splitter (64 steps expected; a walk that closes on itself without
// meeting one is a cycle: Brent's test, exact); doubling then runs over the splitters only.
__device__ inline bool jump_is_splitter(uint32_t v, const Jump &j /* one-step entry */) {
    return (j.score & JUMP_TERM) || (j.hops & JUMP_START) || ((v * 0x9E3779B1u) >> 26) == 0u;  // 1/64 by hash
}

struct PredSplitter {
    const Jump *J0;
    __device__ bool operator()(uint64_t v) const { return jump_is_splitter((uint32_t)v, J0[v]); }
};

// position of node x in the ascending splitter list (x is a splitter)
__device__ inline uint32_t jump_rank(const uint32_t *__restrict__ list, uint64_t n, uint32_t x) {
    uint64_t lo = 0, hi = n;  // list[lo] <= x < list[hi]
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (list[mid] <= x) lo = mid; else hi = mid;
    }
    return (uint32_t)lo;
}

// R[i] for splitter i = list[i]: (RANK of the next splitter on its chain, hops and score up to it), unresolved; a
// splitter whose own chain ends in itself keeps its resolved entry (target unused).  A chain that cycles without a
// splitter points at itself: never resolved.  R is DENSE over the splitters -- 1/30 of the nodes, 190 MB at the
// BASELINE size: the doubling rounds below then run inside the last-level cache instead of touching one 64-byte line
// of a per-node table per splitter (17 rounds x 0.8 ms before, x 0.1 ms now).
__global__ __launch_bounds__(256) void k_jump_walk(const uint32_t *__restrict__ list, uint64_t n_list, const Jump *__restrict__ J0,
                                                   Jump *R) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_list) return;
    const uint32_t s = list[i];
    Jump acc = J0[s];
    acc.hops &= ~JUMP_START;
    if (!(acc.score & JUMP_TERM)) {
        uint32_t x = acc.target, tortoise = s;
        uint32_t power = 1, lam = 1;
        bool cycle = false;
        for (;;) {
            const Jump jx = J0[x];
            if (jump_is_splitter(x, jx)) break;
            if (x == tortoise) { cycle = true; break; }  // closed on itself: a cycle
            if (lam == power) { tortoise = x; power <<= 1; lam = 0; }
            ++lam;
            acc.hops += jx.hops;  // not a start: the flag bit is clear
            acc.score += jx.score;
            x = jx.target;
        }
        if (cycle) acc = Jump{(uint32_t)i, 0u, 0ull};
        else acc.target = jump_rank(list, n_list, x);
    }
    R[i] = acc;
}

__global__ __launch_bounds__(256) void k_jump_step_dense(uint64_t n, const Jump *__restrict__ in, Jump *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Jump a = in[i];
    if (!(a.score & JUMP_TERM)) {
        const Jump b = in[a.target];
        a.target = b.target;
        a.hops += b.hops;
        a.score += b.score;
    }
    out[i] = a;
}

// every start is a splitter: its position in the splitter list
__global__ __launch_bounds__(256) void k_jump_start_ranks(const uint32_t *__restrict__ starts, uint64_t n_starts,
                                                          const uint32_t *__restrict__ list, uint64_t n_list, uint32_t *rank,
                                                          unsigned long long *scalars) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_starts) return;
    const uint32_t r = n_list ? jump_rank(list, n_list, starts[i]) : 0u;
    if (!n_list || list[r] != starts[i]) atomicOr(&scalars[0], 1ull);
    rank[i] = r;
}

struct UnresolvedStart {
    const uint32_t *starts, *rank;
    const Jump *R;
    const uint8_t *flags;
    __device__ uint64_t operator()(uint64_t i) const {
        return !(flags[starts[i]] & DBG_F_PULLED) && !(R[rank[i]].score & JUMP_TERM);
    }
};

__global__ __launch_bounds__(256) void k_jump_starts(const uint32_t *starts, const uint32_t *rank, uint64_t n_starts, const Jump *R,
                                                     const uint8_t *flags, int k, uint64_t *per_ctg, uint64_t *per_chr,
                                                     uint64_t *per_score) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_starts) return;
    const uint32_t s = starts[i];
    const Jump a = R[rank[i]];
    const bool emit = !(flags[s] & DBG_F_PULLED) && (a.score & JUMP_TERM);
    per_ctg[i] = emit;
    per_chr[i] = emit ? (uint64_t)k + a.hops : 0;
    per_score[i] = emit ? (a.score & ~JUMP_TERM) : 0;
}

__global__ __launch_bounds__(256) void k_walk_desc(const uint32_t *starts, uint64_t n_starts, const uint64_t *per_ctg,
                                                   const uint64_t *ctg_base, const uint64_t *char_base,
                                                   const uint64_t *per_score, const uint64_t *stamps, uint64_t *ctg_off,
                                                   uint64_t *score_out, uint64_t *stamp_out, uint32_t *seq_out,
                                                   uint32_t *ctg_start /* may be null */) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_starts || !per_ctg[i]) return;
    const uint64_t c = ctg_base[i];
    if (ctg_start) ctg_start[c] = starts[i];
    ctg_off[c] = char_base[i];
    score_out[c] = per_score[i];
    stamp_out[c] = stamps[starts[i]];
    seq_out[c] = 0;
}

// ---- contig text in parallel.  One thread per start writing its chain character by character is a chain of
// dependent global loads per character (2 s for the 1 261 contigs / 6.3e7 characters of BASELINE configs[0] without
// errors: every start re-walks the shared genome path).  With the 2^j-th chain successor of every node tabulated
// (binary lifting, ceil(log2(longest contig)) levels of n x 4 bytes) any character of any contig is an independent
// O(log) lookup: character j >= k of the contig that starts at s is the last character of s's (j - k + 1)-th successor.
template <class G>
__global__ __launch_bounds__(256) void k_lift_init(uint64_t n_nodes, G g, uint32_t *up0) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_nodes) return;
    uint32_t nxt = (uint32_t)v;  // chain ends: the node is its own successor (never followed inside a contig)
    const uint8_t f = g.flags[v];
    if (!(f & (DBG_F_PULLED | DBG_F_BRANCH)) && g.keep_count((uint32_t)v)) {
        const uint32_t s = g.succ_of((uint32_t)v, g.first_kept((uint32_t)v));
        if (!(g.flags[s] & DBG_F_PULLED)) nxt = s;
    }
    up0[v] = nxt;
}
__global__ __launch_bounds__(256) void k_lift_step(uint64_t n_nodes, const uint32_t *__restrict__ prev, uint32_t *next) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n_nodes) next[v] = prev[prev[v]];
}
template <class G>
__global__ __launch_bounds__(256) void k_text_fill(uint64_t n_chars, G g, const uint64_t *__restrict__ ctg_off, uint64_t n_ctg,
                                                   const uint32_t *__restrict__ ctg_start, const uint32_t *__restrict__ up,
                                                   uint64_t n_nodes, int levels, char *chars) {
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chars) return;
    uint64_t lo = 0, hi = n_ctg;  // ctg_off[lo] <= c < ctg_off[hi]
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (ctg_off[mid] <= c) lo = mid; else hi = mid;
    }
    const uint64_t j = c - ctg_off[lo];
    uint32_t x = ctg_start[lo];
    if (j < (uint64_t)g.k) { chars[c] = g.char_at(x, (int)j); return; }
    uint64_t d = j - (uint64_t)g.k + 1;
    for (int l = 0; l < levels && d; ++l, d >>= 1)
        if (d & 1) x = up[(uint64_t)l * n_nodes + x];
    chars[c] = g.sym_char(g.last_code(x));
}

// ==========================================================================================
// C ABI
// ==========================================================================================
// a shard's successor ids (owner << 29 | id) index other handles' node tables: traversal kernels must not follow them
static const char *const kPartialGraph =
    "this handle holds one shard of a multi-GPU build: gather the shards first (multi_gpu.gather_graph / dbg_import_graph)";
static void free_build(dbg *h) {
    multipass_free(h);
    dev_free(h->d_tab); dev_free(h->d_occ);
    if (h->nodes_in_arena) {
        h->d_keys = nullptr; h->d_stamps = nullptr; h->d_cnt = nullptr; h->d_flags = nullptr;
        h->d_order = nullptr; h->d_succ = nullptr; h->d_deg = nullptr; h->d_keys_hi = nullptr;
        h->nodes_in_arena = false;
    }
    dev_free(h->d_btab);
    h->btab_cap = 0;
    dev_free(h->d_keys_hi);
    dev_free(h->d_fsorder);
    h->order_exact = false;
    dev_free(h->d_keepmask); dev_free(h->d_rank_mc); dev_free(h->d_rank_fs);
    h->D = 4; h->sym_bits = 2;
    dev_free(h->d_keys); dev_free(h->d_stamps); dev_free(h->d_cnt); dev_free(h->d_flags);
    dev_free(h->d_order); dev_free(h->d_succ); dev_free(h->d_deg);
    h->d_rowptr = nullptr; h->d_col = nullptr; h->d_ecnt = nullptr;  // arena-owned (ar_csr)
    h->d_rowptr32 = nullptr; h->d_stamps_st = nullptr; h->stamps_st_bytes = 0;
    h->dense_pending = false;
    h->csr_built = false;
    h->d_pull_rank = nullptr;  // arena-owned (ar_tips)
    dev_free(h->d_read_flags);
    dev_free(h->d_ctg_off); dev_free(h->d_ctg_chars); dev_free(h->d_ctg_score); dev_free(h->d_ctg_stamp);
    dev_free(h->d_ctg_seq); dev_free(h->d_ctg_start); dev_free(h->d_lift);
    h->lift_levels = 0;
    h->partial_graph = false;
    h->sk_src.valid = false;
    h->k = 0; h->cap = 0; h->n_nodes = h->n_edges = 0;
    h->pruned = h->tipped = h->pull_reads_done = h->walked = h->walk_indexed = false;
    h->n_branch = h->n_pulled = h->tip_rounds = h->n_pull_reads = 0;
    h->n_starts = h->n_contigs = h->contig_chars = 0;
    h->starts_known = false;
    h->n_kmer_inst = h->n_edge_inst = 0;
}

static void free_reads(dbg *h) {
    if (h->own_bases) dev_free(h->d_bases);
    if (h->own_offsets) dev_free(h->d_offsets);
    h->d_bases = nullptr; h->d_offsets = nullptr;
    h->own_bases = h->own_offsets = false;
    dev_free(h->d_startbits);
    dev_free(h->d_lut); dev_free(h->d_alpha);
    h->alpha_known = false;
    h->n_bytes = h->n_reads = 0;
}

extern "C" int dbg_abi_version(void) { return DBG_ABI_VERSION; }

extern "C" int dbg_create(int device, dbg_t **out) {
    if (!out) return DBG_E_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return DBG_E_HIP;
    dbg *h = new (std::nothrow) dbg();
    if (!h) return DBG_E_NOMEM;
    h->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&h->stream) != hipSuccess ||
        hipMalloc((void **)&h->d_scalars, 128 * sizeof(uint64_t)) != hipSuccess) {
        delete h;
        return DBG_E_HIP;
    }
    {  // keep freed arena memory in the pool instead of returning it to the device at every synchronisation
        hipMemPool_t pool;
        if (hipDeviceGetDefaultMemPool(&pool, device) == hipSuccess) {
            uint64_t keep = ~0ull;
            (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
        }
        (void)hipGetLastError();
    }
    *out = h;
    return DBG_OK;
}

extern "C" void dbg_destroy(dbg_t *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    h->dropping_parts = true;
    free_build(h);
    if (h->spare_part) { dbg_destroy(h->spare_part); h->spare_part = nullptr; }
    free_reads(h);
    for (auto &lvl : h->ar_rec) for (auto &b : lvl) buf_free(h, b);
    for (auto &lvl : h->ar_q) for (auto &b : lvl) buf_free(h, b);
    for (auto &b : h->ar_node) buf_free(h, b);
    for (auto &b : h->ar_misc) buf_free(h, b);
    for (auto &b : h->ar_csr) buf_free(h, b);
    buf_free(h, h->ar_dir);
    buf_free(h, h->ar_l2);
    for (auto &b : h->ar_shard) buf_free(h, b);
    for (auto &b : h->ar_walk) buf_free(h, b);
    for (auto &b : h->ar_wide) buf_free(h, b);
    for (auto &b : h->ar_refine) buf_free(h, b);
    for (auto &b : h->ar_tips) buf_free(h, b);
    for (auto &lvl : h->ar_part) for (auto &b : lvl) buf_free(h, b);
    buf_free(h, h->ar_scan);
    dev_free(h->d_scalars);
    delete (ShardState *)h->shard_state;
    (void)hipStreamSynchronize(h->stream);  // the arena buffers were freed in stream order
    if (!h->borrowed_stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" const char *dbg_last_error(const dbg_t *h) { return h ? h->err.c_str() : "null handle"; }

static int make_startbits(dbg *h) {
    Timer t(h->stream);
    dev_free(h->d_startbits);
    h->startbits_words = (h->n_bytes + 1 + 31) / 32 + SB_WORDS + 2;
    CHK(dev_alloc(h, &h->d_startbits, h->startbits_words));
    HIPCHK(h, hipMemsetAsync(h->d_startbits, 0, h->startbits_words * 4, h->stream));
    hipLaunchKernelGGL(k_startbits, dim3(grid_for(h->n_reads + 1, 256)), dim3(256), 0, h->stream, h->d_offsets,
                       h->n_reads, h->d_startbits);
    HIPCHK(h, hipGetLastError());
    h->stats.ms_startbits = t.stop();
    return DBG_OK;
}

extern "C" int dbg_set_reads(dbg_t *h, const char *bases, const uint64_t *offsets, uint64_t n_reads) {
    if (!h || !offsets || (n_reads && !bases && offsets[n_reads] != 0)) return DBG_E_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    if (offsets[0] != 0) { h->err = "offsets[0] must be 0"; return DBG_E_ARG; }
    for (uint64_t i = 0; i < n_reads; ++i)
        if (offsets[i + 1] < offsets[i]) { h->err = "offsets must be non-decreasing"; return DBG_E_ARG; }
    free_build(h);
    free_reads(h);
    const uint64_t nb = offsets[n_reads];
    Timer t(h->stream);
    CHK(dev_alloc(h, &h->d_bases, nb + 64));
    h->own_bases = true;
    CHK(dev_alloc(h, &h->d_offsets, n_reads + 1));
    h->own_offsets = true;
    if (nb) HIPCHK(h, hipMemcpyAsync(h->d_bases, bases, nb, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_offsets, offsets, (n_reads + 1) * 8, hipMemcpyHostToDevice, h->stream));
    h->stats.ms_h2d = t.stop();
    h->n_bytes = nb;
    h->n_reads = n_reads;
    return make_startbits(h);
}

extern "C" int dbg_set_reads_fasta(dbg_t *h, const char *text, uint64_t n_text) {
    if (!h || (n_text && !text)) return DBG_E_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    free_build(h);
    free_reads(h);
    char *d_text = nullptr;
    uint32_t *bits = nullptr, *word_rank = nullptr, *len = nullptr;
    uint64_t *line_start = nullptr, *read_idx = nullptr, *byte_off = nullptr;
    uint8_t *is_read = nullptr;
    auto cleanup = [&]() {
        dev_free(d_text); dev_free(bits); dev_free(word_rank); dev_free(len); dev_free(line_start); dev_free(read_idx);
        dev_free(byte_off); dev_free(is_read);
    };
    int rc = DBG_OK;
    uint64_t n_lines = 0, n_reads = 0, n_bases = 0;
    do {
        Timer t(h->stream);
        if ((rc = dev_alloc(h, &d_text, n_text + 64)) != DBG_OK) break;
        if (n_text && hipMemcpyAsync(d_text, text, n_text, hipMemcpyHostToDevice, h->stream) != hipSuccess) { h->err = "H2D copy failed"; rc = DBG_E_HIP; break; }
        h->stats.ms_h2d = t.stop();
        const uint64_t n_words = (n_text + 31) / 32;
        if ((rc = dev_alloc(h, &bits, n_words)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &word_rank, n_words)) != DBG_OK) break;
        if (n_words) {
            hipLaunchKernelGGL(k_fa_mark, dim3(grid_for(n_words, 256)), dim3(256), 0, h->stream, d_text, n_text, bits);
            if ((rc = exclusive_scan(h, n_words, PopcWords{bits}, word_rank, &n_lines)) != DBG_OK) break;
        }
        if ((rc = dev_alloc(h, &line_start, n_lines)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &is_read, n_lines)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &len, n_lines)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &read_idx, n_lines)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &byte_off, n_lines)) != DBG_OK) break;
        if (n_lines) {
            hipLaunchKernelGGL(k_fa_line_starts, dim3(grid_for(n_words, 256)), dim3(256), 0, h->stream, bits, word_rank, n_words,
                               line_start);
            hipLaunchKernelGGL(k_fa_lines, dim3(grid_for(n_lines, 256)), dim3(256), 0, h->stream, d_text, n_text, line_start,
                               n_lines, is_read, len);
            if ((rc = exclusive_scan(h, n_lines, ByteSet{is_read}, read_idx, &n_reads)) != DBG_OK) break;
            if ((rc = exclusive_scan(h, n_lines, U32At{len}, byte_off, &n_bases)) != DBG_OK) break;
        }
        if ((rc = dev_alloc(h, &h->d_bases, n_bases + 64)) != DBG_OK) break;
        h->own_bases = true;
        if ((rc = dev_alloc(h, &h->d_offsets, n_reads + 1)) != DBG_OK) break;
        h->own_offsets = true;
        if (n_lines)
            hipLaunchKernelGGL(k_fa_copy, dim3(grid_for(n_lines * 16, 256)), dim3(256), 0, h->stream, d_text, line_start, n_lines,
                               is_read, len, read_idx, byte_off, h->d_bases, h->d_offsets);
        if (hipMemcpyAsync(h->d_offsets + n_reads, &n_bases, 8, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess) { h->err = "FASTA ingest failed on the device"; rc = DBG_E_HIP; break; }
        h->n_bytes = n_bases;
        h->n_reads = n_reads;
    } while (0);
    cleanup();
    if (rc != DBG_OK) { free_reads(h); return rc; }
    return make_startbits(h);
}

extern "C" int dbg_set_reads_device(dbg_t *h, const void *d_bases, uint64_t n_bytes, const void *d_offsets,
                                    uint64_t n_reads) {
    if (!h || !d_offsets || (n_bytes && !d_bases)) return DBG_E_ARG;
    if (((uintptr_t)d_bases & 15) || ((uintptr_t)d_offsets & 7)) { h->err = "device buffers must be 16/8-byte aligned"; return DBG_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    free_build(h);
    free_reads(h);
    h->d_bases = (char *)d_bases;
    h->d_offsets = (uint64_t *)d_offsets;
    h->n_bytes = n_bytes;
    h->n_reads = n_reads;
    return make_startbits(h);
}

extern "C" int dbg_synth_reads(dbg_t *h, uint64_t seed, uint64_t genome_len, uint64_t first_read, uint64_t n_reads,
                               uint32_t read_len, uint32_t err_thr24) {
    if (!h || read_len == 0 || genome_len < read_len) return DBG_E_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    free_build(h);
    free_reads(h);
    const uint64_t nb = n_reads * read_len;
    CHK(dev_alloc(h, &h->d_bases, nb + 64));
    h->own_bases = true;
    CHK(dev_alloc(h, &h->d_offsets, n_reads + 1));
    h->own_offsets = true;
    const uint64_t golden = 0x9E3779B97F4A7C15ull;
    const uint64_t kg = mix64(seed + 1 * golden), ks = mix64(seed + 2 * golden), ke = mix64(seed + 3 * golden);
    const unsigned grid = (unsigned)std::min<uint64_t>(std::max<uint64_t>(1, (nb + 255) / 256), 1u << 20);
    hipLaunchKernelGGL(k_synth, dim3(grid), dim3(256), 0, h->stream, h->d_bases, h->d_offsets, kg, ks, ke, genome_len,
                       first_read, n_reads, read_len, err_thr24);
    HIPCHK(h, hipGetLastError());
    h->n_bytes = nb;
    h->n_reads = n_reads;
    return make_startbits(h);
}

extern "C" int dbg_reads_checksum(dbg_t *h, uint64_t *out) {
    if (!h || !out) return DBG_E_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemsetAsync(h->d_scalars + 8, 0, 8, h->stream));
    if (h->n_bytes) {
        const unsigned grid = (unsigned)std::min<uint64_t>((h->n_bytes + 255) / 256, 1u << 16);
        hipLaunchKernelGGL(k_checksum, dim3(grid), dim3(256), 0, h->stream, h->d_bases, h->n_bytes,
                           (unsigned long long *)(h->d_scalars + 8));
    }
    HIPCHK(h, hipMemcpyAsync(out, h->d_scalars + 8, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return DBG_OK;
}

extern "C" int dbg_copy_reads(dbg_t *h, char *bases, uint64_t *offsets) {
    if (!h) return DBG_E_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    if (bases && h->n_bytes) HIPCHK(h, hipMemcpyAsync(bases, h->d_bases, h->n_bytes, hipMemcpyDeviceToHost, h->stream));
    if (offsets && h->d_offsets)
        HIPCHK(h, hipMemcpyAsync(offsets, h->d_offsets, (h->n_reads + 1) * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return DBG_OK;
}

// ---- a selection of the resident reads, gathered on the device (pull_out_read of construct_graph, debruijn.py:274-278:
//      a few per cent of the reads -- the others never leave the GPU)
struct TakeLen {
    const uint64_t *idx, *offsets;
    uint64_t n_reads;
    __device__ uint64_t operator()(uint64_t i) const { const uint64_t r = idx[i]; return r < n_reads ? offsets[r + 1] - offsets[r] : 0; }
};
__global__ __launch_bounds__(256) void k_take_copy(const uint64_t *__restrict__ idx, uint64_t n, const uint64_t *__restrict__ offsets,
                                                   uint64_t n_reads, const char *__restrict__ bases, const uint64_t *__restrict__ out_off,
                                                   char *out) {
    const uint64_t i = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // one wave per selected read
    if (i >= n) return;
    const uint64_t r = idx[i];
    if (r >= n_reads) return;
    const uint64_t beg = offsets[r], len = offsets[r + 1] - beg, o = out_off[i];
    for (uint64_t j = threadIdx.x & 63; j < len; j += 64) out[o + j] = bases[beg + j];
}

extern "C" int dbg_take_reads(dbg_t *h, const uint64_t *indices, uint64_t n, uint64_t *out_offsets, char *out_chars,
                              uint64_t capacity, uint64_t *n_chars) {
    if (!h || (n && !indices) || !n_chars) return DBG_E_ARG;
    if (!h->d_offsets) { h->err = "no reads set"; return DBG_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    *n_chars = 0;
    if (out_offsets) out_offsets[0] = 0;
    if (!n) return DBG_OK;
    for (uint64_t i = 0; i < n; ++i)
        if (indices[i] >= h->n_reads) { h->err = "read index out of range"; return DBG_E_ARG; }
    uint64_t *d_idx = nullptr, *d_off = nullptr;
    char *d_out = nullptr;
    CHK(dev_alloc(h, &d_idx, n));
    int rc = dev_alloc(h, &d_off, n + 1);
    if (rc != DBG_OK) { dev_free(d_idx); return rc; }
    auto done = [&](int code) { dev_free(d_idx); dev_free(d_off); dev_free(d_out); return code; };
    if (hipMemcpyAsync(d_idx, indices, n * 8, hipMemcpyHostToDevice, h->stream) != hipSuccess) { h->err = "dbg_take_reads: copy of the indices failed"; return done(DBG_E_HIP); }
    uint64_t total = 0;
    rc = exclusive_scan(h, n, TakeLen{d_idx, h->d_offsets, h->n_reads}, d_off, &total);
    if (rc != DBG_OK) return done(rc);
    *n_chars = total;
    if (out_offsets) {
        if (hipMemcpyAsync(out_offsets, d_off, n * 8, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess) { h->err = "dbg_take_reads: copy of the offsets failed"; return done(DBG_E_HIP); }
        out_offsets[n] = total;
    }
    if (!out_chars) return done(DBG_OK);  // first call of the two-call pattern: sizes only
    if (capacity < total) { h->err = "dbg_take_reads: capacity below the selected reads' total length"; return done(DBG_E_CAPACITY); }
    if (total) {
        rc = dev_alloc(h, &d_out, total);
        if (rc != DBG_OK) return done(rc);
        hipLaunchKernelGGL(k_take_copy, dim3(grid_for(n, 4)), dim3(256), 0, h->stream, d_idx, n, h->d_offsets, h->n_reads,
                           h->d_bases, d_off, d_out);
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(out_chars, d_out, total, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess) { h->err = "dbg_take_reads: gather failed"; return done(DBG_E_HIP); }
    }
    return done(DBG_OK);
}

// ------------------------------------------------------------------------------------------
static int build_sk(dbg *h, int k, uint64_t node_capacity_hint);
static int build_wsk(dbg *h, int k, bool *fallback);
// CSR over distinct edges + start count; shared by both engines
static int finish_graph(dbg *h) {
    if (!h->csr_built) {
        Timer t(h->stream);
        CHK(buf_ensure(h, h->ar_csr[0], (h->n_nodes + 1) * 8));
        h->d_rowptr = (uint64_t *)h->ar_csr[0].p;
        uint64_t total = 0;
        CHK(exclusive_scan(h, h->n_nodes, DegOf{h->d_deg}, h->d_rowptr, &total));
        h->n_edges = total;
        HIPCHK(h, hipMemcpyAsync(h->d_rowptr + h->n_nodes, &h->n_edges, 8, hipMemcpyHostToDevice, h->stream));
        CHK(buf_ensure(h, h->ar_csr[1], total * 4));
        CHK(buf_ensure(h, h->ar_csr[2], total * 4));
        h->d_col = (uint32_t *)h->ar_csr[1].p;
        h->d_ecnt = (uint32_t *)h->ar_csr[2].p;
        if (h->n_nodes) {
            if (h->D == GEN_D)
                hipLaunchKernelGGL(k_g_csr_fill, dim3(grid_for(h->n_nodes, 256)), dim3(256), 0, h->stream, h->n_nodes,
                                   h->d_rowptr, h->d_cnt, h->d_succ, h->d_col, h->d_ecnt);
            else
                hipLaunchKernelGGL(k_csr_fill, dim3(grid_for(h->n_nodes, 256)), dim3(256), 0, h->stream, h->n_nodes,
                                   h->d_rowptr, h->d_cnt, h->d_succ, h->d_col, h->d_ecnt);
            HIPCHK(h, hipGetLastError());
        }
        h->stats.ms_csr = t.stop();
    }
    h->starts_known = false;  // counted when first asked for (dbg_get_sizes, dbg_walk): the walk's input, not the build's
    return DBG_OK;
}

// Engine 0 builds keys + stamps + CSR only (SkCountOut); the dense per-base views the traversal kernels and the
// exports read are derived here, once, the first time something asks for them.
static const char *const kMultipassGraph =
    "this handle holds a multi-pass graph (more nodes than one id space): read it part by part (dbg_part_sizes / dbg_export_part)";
static int ensure_dense(dbg *h) {
    if (h->multipass) { h->err = kMultipassGraph; return DBG_E_ARG; }
    if (!h->dense_pending) return DBG_OK;
    const uint64_t n = h->n_nodes, cap = n ? n : 1;
    CHK(buf_ensure(h, h->ar_node[2], cap * 16));
    CHK(buf_ensure(h, h->ar_node[4], cap));
    CHK(buf_ensure(h, h->ar_node[5], cap * 16));
    CHK(buf_ensure(h, h->ar_csr[0], (cap + 1) * 8));
    const bool widen = h->stamps_st_bytes == 4;
    if (widen) CHK(buf_ensure(h, h->ar_node[1], cap * 8));
    h->d_cnt = (uint32_t *)h->ar_node[2].p;
    h->d_order = (uint8_t *)h->ar_node[4].p;
    h->d_succ = (uint32_t *)h->ar_node[5].p;
    h->d_rowptr = (uint64_t *)h->ar_csr[0].p;
    if (widen) h->d_stamps = (uint64_t *)h->ar_node[1].p;
    if (n) {
        if (widen)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_dense_from_csr<uint32_t>), dim3(grid_for(n, 256)), dim3(256), 0, h->stream, n,
                               (const uint32_t *)h->d_stamps_st, h->d_rowptr32, h->d_col, h->d_ecnt, h->d_flags, h->d_stamps,
                               h->d_rowptr, h->d_cnt, h->d_succ, h->d_order);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_dense_from_csr<uint64_t>), dim3(grid_for(n, 256)), dim3(256), 0, h->stream, n,
                               (const uint64_t *)h->d_stamps_st, h->d_rowptr32, h->d_col, h->d_ecnt, h->d_flags,
                               (uint64_t *)nullptr, h->d_rowptr, h->d_cnt, h->d_succ, h->d_order);
        HIPCHK(h, hipGetLastError());
    }
    HIPCHK(h, hipMemcpyAsync(h->d_rowptr + n, &h->n_edges, 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->dense_pending = false;
    return DBG_OK;
}

// starts = nodes with indegree 0 (debruijn.py:334-336)
static int multipass_starts(dbg *h, uint64_t *total);
static int ensure_starts(dbg *h) {
    if (h->starts_known || !h->k) return DBG_OK;
    uint64_t total = 0;
    if (h->multipass) CHK(multipass_starts(h, &total));
    else CHK(reduce_sum(h, h->n_nodes, FlagSet{h->d_flags, DBG_F_INDEG, 0}, &total));
    h->n_starts = total;
    h->starts_known = true;
    return DBG_OK;
}


// ------------------------------------------------------------------------------------------
// alphabet of the current read set; generic engine host side
// ------------------------------------------------------------------------------------------
static int compute_alphabet(dbg *h) {
    if (h->alpha_known) return DBG_OK;
    uint64_t hist[256];
    memset(hist, 0, sizeof(hist));
    if (h->n_bytes) {
        unsigned long long *d_hist = nullptr;
        CHK(dev_alloc(h, &d_hist, 256));
        (void)hipMemsetAsync(d_hist, 0, 256 * 8, h->stream);
        const unsigned grid = (unsigned)std::min<uint64_t>((h->n_bytes + 255) / 256, 4096);
        hipLaunchKernelGGL(k_g_hist, dim3(grid), dim3(256), 0, h->stream, h->d_bases, h->n_bytes, d_hist);
        hipError_t e = hipMemcpyAsync(hist, d_hist, sizeof(hist), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        (void)hipFree(d_hist);
        if (e != hipSuccess) { h->err = std::string("alphabet scan: ") + hipGetErrorString(e); return DBG_E_HIP; }
    }
    h->n_sym = 0;
    h->is_dna = true;
    uint8_t lut[256];
    memset(lut, 0xFF, sizeof(lut));
    for (int c = 0; c < 256; ++c) {
        if (!hist[c]) continue;
        if (c != 'A' && c != 'C' && c != 'G' && c != 'T') h->is_dna = false;
        if (h->n_sym < 32) { h->alphabet[h->n_sym] = (uint8_t)c; lut[c] = (uint8_t)h->n_sym; }
        ++h->n_sym;
    }
    dev_free(h->d_lut);
    dev_free(h->d_alpha);
    CHK(dev_alloc(h, &h->d_lut, 256));
    CHK(dev_alloc(h, &h->d_alpha, 32));
    HIPCHK(h, hipMemcpyAsync(h->d_lut, lut, 256, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_alpha, h->alphabet, 32, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->alpha_known = true;
    return DBG_OK;
}

static GGen gen_view(const dbg *h) {
    return GGen{h->d_keys, h->d_bases, h->d_stamps, h->d_lut, h->d_flags, h->d_keepmask, h->d_rank_mc, h->d_deg, h->d_succ,
                h->d_cnt, h->d_alpha, h->k};
}

// k-mers longer than one packed word: tables keyed by reference into the reads (dbg_genref.h); d_keys stays null
static int build_genref(dbg *h, int k) {
    h->D = GEN_D;
    h->sym_bits = GEN_BITS;
    if (h->n_bytes >= (1ull << 46)) { h->err = "reads too large for 48-bit references"; return DBG_E_CAPACITY; }
    Timer t_count(h->stream);
    uint64_t cap = 1024;
    int lg = 10;
    while (cap < (h->n_bytes + 1) * 2) { cap <<= 1; ++lg; }
    unsigned long long *ntab = nullptr, *etab = nullptr, *estamp = nullptr;
    uint32_t *nocc = nullptr, *eocc = nullptr, *ecount = nullptr, *word_rank = nullptr;
    auto cleanup = [&]() { dev_free(ntab); dev_free(etab); dev_free(nocc); dev_free(eocc); dev_free(ecount); dev_free(word_rank); dev_free(estamp); };
    int rc = DBG_OK;
    do {
        if ((rc = dev_alloc(h, &ntab, cap)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &etab, cap)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &ecount, cap)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &nocc, cap / 32)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &eocc, cap / 32)) != DBG_OK) break;
        (void)hipMemsetAsync(ntab, 0xFF, cap * 8, h->stream);
        (void)hipMemsetAsync(etab, 0xFF, cap * 8, h->stream);
        (void)hipMemsetAsync(ecount, 0, cap * 4, h->stream);
        (void)hipMemsetAsync(nocc, 0, cap / 8, h->stream);
        (void)hipMemsetAsync(eocc, 0, cap / 8, h->stream);
        (void)hipMemsetAsync(h->d_scalars, 0, 64 * 8, h->stream);
        if (h->n_bytes) {
            const unsigned grid = (unsigned)std::min<uint64_t>((h->n_bytes + 255) / 256, 1u << 16);
            hipLaunchKernelGGL(k_gr_insert, dim3(grid), dim3(256), 0, h->stream, h->d_bases, h->n_bytes, h->d_startbits, k, ntab,
                               nocc, etab, eocc, ecount, cap - 1, 64 - lg, (unsigned long long *)h->d_scalars);
        }
        uint64_t sc[4] = {0, 0, 0, 0};
        if (hipMemcpyAsync(sc, h->d_scalars, 32, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess) { h->err = "generic count failed"; rc = DBG_E_HIP; break; }
        if (sc[0] & 2) { h->err = "generic engine: hash table full"; rc = DBG_E_CAPACITY; break; }
        h->n_kmer_inst = sc[1];
        h->n_edge_inst = sc[2];
        h->stats.ms_count = t_count.stop();
        h->stats.count_launches = 1;
        Timer t_c(h->stream);
        const uint64_t n_words = cap / 32;
        if ((rc = dev_alloc(h, &word_rank, n_words)) != DBG_OK) break;
        uint64_t n = 0;
        if ((rc = exclusive_scan(h, n_words, PopcWords{nocc}, word_rank, &n)) != DBG_OK) break;
        if (n >= 0xFFFFFFF0ull) { h->err = "more than 2^32-16 nodes"; rc = DBG_E_CAPACITY; break; }
        h->n_nodes = n;
        if ((rc = dev_alloc(h, &h->d_stamps, n)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &h->d_flags, n)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &h->d_cnt, n * GEN_D)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &h->d_succ, n * GEN_D)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &h->d_deg, n)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &h->d_keepmask, n)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &h->d_rank_mc, n * GEN_D)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &h->d_rank_fs, n * GEN_D)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &estamp, n * GEN_D)) != DBG_OK) break;
        if (n) {
            (void)hipMemsetAsync(h->d_cnt, 0, n * GEN_D * 4, h->stream);
            (void)hipMemsetAsync(h->d_succ, 0xFF, n * GEN_D * 4, h->stream);
            (void)hipMemsetAsync(estamp, 0xFF, n * GEN_D * 8, h->stream);
            (void)hipMemsetAsync(h->d_keepmask, 0, n * 4, h->stream);
            hipLaunchKernelGGL(k_gr_gather, dim3(grid_for(n_words, 256)), dim3(256), 0, h->stream, ntab, nocc, word_rank, n_words,
                               h->d_stamps, h->d_flags);
            hipLaunchKernelGGL(k_gr_edges, dim3(grid_for(cap, 256)), dim3(256), 0, h->stream, etab, ecount, cap, ntab, cap - 1,
                               64 - lg, h->d_bases, h->d_stamps, h->d_lut, k, h->d_cnt, h->d_succ, estamp,
                               (unsigned long long *)h->d_scalars);
            hipLaunchKernelGGL(k_g_rank, dim3(grid_for(n, 256)), dim3(256), 0, h->stream, n, h->d_cnt, estamp, h->d_rank_mc,
                               h->d_rank_fs, h->d_deg);
        }
        if (hipMemcpyAsync(sc, h->d_scalars, 8, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess) { h->err = "generic graph assembly failed"; rc = DBG_E_HIP; break; }
        if (sc[0] & 128) { h->err = "internal: edge endpoint missing from the node table"; rc = DBG_E_HIP; break; }
        h->stats.ms_succ = t_c.stop();
        h->order_exact = true;  // ranks come from per-edge first-seen positions already
    } while (0);
    cleanup();
    return rc;
}

static int build_generic(dbg *h, int k) {
    if (h->n_sym > GEN_D) { h->err = "reads hold more than 32 distinct characters"; return DBG_E_ALPHABET; }
    if (GEN_BITS * (k + 1) > 64) return build_genref(h, k);  // k >= 12: by-reference tables
    h->D = GEN_D;
    h->sym_bits = GEN_BITS;
    Timer t_count(h->stream);
    uint64_t cap = 1024;
    while (cap < (h->n_bytes + 1) * 2) cap <<= 1;
    GenNodeSlot *nodes = nullptr;
    GenEdgeSlot *edges = nullptr;
    uint32_t *occ = nullptr, *eocc = nullptr, *word_rank = nullptr;
    unsigned long long *estamp = nullptr;
    auto cleanup = [&]() { dev_free(nodes); dev_free(edges); dev_free(occ); dev_free(eocc); dev_free(word_rank); dev_free(estamp); };
    int rc = DBG_OK;
    do {
        if ((rc = dev_alloc(h, &nodes, cap)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &edges, cap)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &occ, cap / 32)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &eocc, cap / 32)) != DBG_OK) break;
        (void)hipMemsetAsync(occ, 0, cap / 8, h->stream);
        (void)hipMemsetAsync(eocc, 0, cap / 8, h->stream);
        (void)hipMemsetAsync(h->d_scalars, 0, 64 * 8, h->stream);
        hipLaunchKernelGGL(k_g_init_tables, dim3(grid_for(cap, 256)), dim3(256), 0, h->stream, nodes, edges, cap);
        if (h->n_bytes) {
            const unsigned grid = (unsigned)std::min<uint64_t>((h->n_bytes + 255) / 256, 1u << 16);
            hipLaunchKernelGGL(k_g_insert, dim3(grid), dim3(256), 0, h->stream, h->d_bases, h->n_bytes, h->d_startbits, k,
                               h->d_lut, nodes, occ, edges, eocc, cap - 1, (unsigned long long *)h->d_scalars);
        }
        uint64_t sc[4] = {0, 0, 0, 0};
        if (hipMemcpyAsync(sc, h->d_scalars, 32, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess) { h->err = "generic count failed"; rc = DBG_E_HIP; break; }
        if (sc[0] & 2) { h->err = "generic engine: hash table full"; rc = DBG_E_CAPACITY; break; }
        h->n_kmer_inst = sc[1];
        h->n_edge_inst = sc[2];
        h->stats.ms_count = t_count.stop();
        h->stats.count_launches = 1;
        // nodes in table order
        Timer t_c(h->stream);
        const uint64_t n_words = cap / 32;
        if ((rc = dev_alloc(h, &word_rank, n_words)) != DBG_OK) break;
        uint64_t n = 0;
        if ((rc = exclusive_scan(h, n_words, PopcWords{occ}, word_rank, &n)) != DBG_OK) break;
        h->n_nodes = n;
        if ((rc = dev_alloc(h, &h->d_keys, n)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &h->d_stamps, n)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &h->d_flags, n)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &h->d_cnt, n * GEN_D)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &h->d_succ, n * GEN_D)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &h->d_deg, n)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &h->d_keepmask, n)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &h->d_rank_mc, n * GEN_D)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &h->d_rank_fs, n * GEN_D)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &estamp, n * GEN_D)) != DBG_OK) break;
        if (n) {
            (void)hipMemsetAsync(h->d_cnt, 0, n * GEN_D * 4, h->stream);
            (void)hipMemsetAsync(h->d_succ, 0xFF, n * GEN_D * 4, h->stream);
            (void)hipMemsetAsync(estamp, 0xFF, n * GEN_D * 8, h->stream);
            (void)hipMemsetAsync(h->d_keepmask, 0, n * 4, h->stream);
            hipLaunchKernelGGL(k_g_gather, dim3(grid_for(n_words, 256)), dim3(256), 0, h->stream, nodes, occ, word_rank, n_words,
                               h->d_keys, h->d_stamps, h->d_flags);
            hipLaunchKernelGGL(k_g_edges, dim3(grid_for(cap, 256)), dim3(256), 0, h->stream, edges, cap, nodes, cap - 1, k,
                               h->d_cnt, h->d_succ, estamp, (unsigned long long *)h->d_scalars);
            hipLaunchKernelGGL(k_g_rank, dim3(grid_for(n, 256)), dim3(256), 0, h->stream, n, h->d_cnt, estamp, h->d_rank_mc,
                               h->d_rank_fs, h->d_deg);
        }
        if (hipMemcpyAsync(sc, h->d_scalars, 8, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess) { h->err = "generic graph assembly failed"; rc = DBG_E_HIP; break; }
        if (sc[0] & 128) { h->err = "internal: edge endpoint missing from the node table"; rc = DBG_E_HIP; break; }
        h->stats.ms_succ = t_c.stop();
        h->order_exact = true;  // ranks come from per-edge first-seen positions already
    } while (0);
    cleanup();
    return rc;
}

extern "C" int dbg_set_option(dbg_t *h, const char *name, int64_t value) {
    if (!h || !name) return DBG_E_ARG;
    const std::string n(name);
    if (n == "engine" && (value == 0 || value == 1)) { h->engine = (int)value; return DBG_OK; }
    if (n == "bucket_bits" && value >= 0 && value <= SK_BUCKET_BITS) { h->bucket_bits = (int)value; return DBG_OK; }
    if (n == "lds_slots" && (value == 2048 || value == 4096)) { h->lds_slots = (int)value; return DBG_OK; }
    if (n == "phase_limit" && value >= 0 && value <= 5) { h->phase_limit = (int)value; return DBG_OK; }
    if (n == "estimate_scale_pct" && value >= 1 && value <= 1000) { h->est_scale_pct = (int)value; return DBG_OK; }
    if (n == "target_distinct" && value >= 0 && value <= 4096) { h->target_distinct = (int)value; return DBG_OK; }
    if (n == "shard_stamp64" && (value == 0 || value == 1)) { h->shard_stamp64 = (int)value; return DBG_OK; }
    if (n == "extract_generic" && (value == 0 || value == 1)) { h->extract_generic = (int)value; return DBG_OK; }
    if (n == "refine_streaming" && (value == 0 || value == 1)) { h->refine_streaming = value != 0; return DBG_OK; }
    if (n == "walk_jump_min_nodes" && value >= 0) { h->walk_jump_min = (uint64_t)value; return DBG_OK; }
    if (n == "wide_engine" && (value == 0 || value == 1)) { h->wide_engine = (int)value; return DBG_OK; }
    if (n == "count_kernel" && value >= 1 && value <= 3) { h->count_kernel = (int)value; return DBG_OK; }
    if (n == "stamp64" && (value == 0 || value == 1)) { h->stamp64 = (int)value; return DBG_OK; }
    if (n == "resolve_sorted" && value >= 0 && value <= 2) { h->resolve_sorted = (int)value; return DBG_OK; }
    if (n == "wcount_kernel" && (value == 1 || value == 2)) { h->wcount_kernel = (int)value; return DBG_OK; }
    if (n == "count_kernel_u64" && value >= 1 && value <= 3) { h->count_kernel_u64 = (int)value; return DBG_OK; }
    if (n == "shard_node_limit" && value >= 0 && value < (1ll << 29)) { h->shard_node_limit = (uint64_t)value; return DBG_OK; }
    h->err = "unknown option or value out of range: " + n;
    return DBG_E_ARG;
}

// ---- 32 <= k <= 63 over ACGT: two-word k-mers (see dbg_wide.h for the table protocol)
// packed: the four successor counters of a slot are 16-bit fields of its spare word; a counter that overflows reports it
// and the caller builds again with the separate 32-bit counters
static int build_wide_once(dbg *h, int k, uint64_t table_capacity_hint, bool packed, bool *overflow) {
    if (h->n_bytes >= (1ull << 43)) { h->err = "reads too large for the 44-bit positions of the table's protocol word"; return DBG_E_CAPACITY; }
    // table slots: the k-mer instances (an upper bound of the distinct k-mers) / 0.7, any multiple of 1024 -- memset and
    // compaction scan are proportional to the table
    uint64_t want = table_capacity_hint;
    if (!want) {
        uint64_t inst = 0;
        CHK(reduce_sum(h, h->n_reads, KmerInstances{h->d_offsets, (uint64_t)k}, &inst));
        want = (uint64_t)((double)(inst + 1) / 0.7);
    }
    const uint64_t cap = std::max<uint64_t>(1024, (want + 1023) / 1024 * 1024);
    const uint64_t pk_words = (h->n_bytes + 31) / 32, n_occ = cap / 32;
    uint64_t *pk = nullptr;
    WSlot *tab = nullptr;
    uint32_t *tcnt = nullptr, *word_rank = nullptr;
    uint8_t *occ = nullptr;  // one byte per slot
    auto cleanup = [&]() {};  // everything lives in the grow-only arena (hipMalloc of tens of GB costs seconds)
    int rc = DBG_OK;
    uint64_t sc[4] = {0, 0, 0, 0};
    do {
        Timer t(h->stream);
        if ((rc = buf_ensure(h, h->ar_wide[0], (pk_words + 3) * 8)) != DBG_OK) break;
        if ((rc = buf_ensure(h, h->ar_wide[1], cap * sizeof(WSlot))) != DBG_OK) break;
        if (!packed && (rc = buf_ensure(h, h->ar_wide[2], cap * 16)) != DBG_OK) break;
        if ((rc = buf_ensure(h, h->ar_wide[3], cap)) != DBG_OK) break;
        if ((rc = buf_ensure(h, h->ar_wide[4], n_occ * 4)) != DBG_OK) break;
        pk = (uint64_t *)h->ar_wide[0].p;
        tab = (WSlot *)h->ar_wide[1].p;
        tcnt = packed ? nullptr : (uint32_t *)h->ar_wide[2].p;
        occ = (uint8_t *)h->ar_wide[3].p;
        word_rank = (uint32_t *)h->ar_wide[4].p;
        (void)hipMemsetAsync(pk + pk_words, 0, 3 * 8, h->stream);
        hipLaunchKernelGGL(k_wtab_init, dim3(grid_for(cap, 256)), dim3(256), 0, h->stream, tab, cap);
        if (tcnt) (void)hipMemsetAsync(tcnt, 0, cap * 16, h->stream);
        (void)hipMemsetAsync(occ, 0, cap, h->stream);
        (void)hipMemsetAsync(h->d_scalars, 0, 64 * 8, h->stream);
        if (pk_words)
            hipLaunchKernelGGL(k_wpack, dim3(grid_for(pk_words, 256)), dim3(256), 0, h->stream, h->d_bases, h->n_bytes,
                               pk_words, pk);
        h->stats.ms_table_init = t.stop();
        Timer tc(h->stream);
        const uint64_t tiles = (h->n_bytes + TILE - 1) / TILE;
        if (tiles)
            hipLaunchKernelGGL(k_wcount, dim3((unsigned)tiles), dim3(256), 0, h->stream, h->d_bases, h->n_bytes,
                               h->d_startbits, k, pk, tab, tcnt, cap, occ, (unsigned long long *)h->d_scalars);
        h->stats.count_launches = tiles ? 1 : 0;
        hipError_t e = hipMemcpyAsync(sc, h->d_scalars, 32, hipMemcpyDeviceToHost, h->stream);
        h->stats.ms_count = tc.stop();
        if (e != hipSuccess || hipGetLastError() != hipSuccess) { h->err = "k_wcount failed"; rc = DBG_E_HIP; break; }
        if (sc[0] & 1) { h->err = "reads hold a byte outside ACGT"; rc = DBG_E_ALPHABET; break; }
        if (sc[0] & 2) { h->err = "hash table capacity exceeded"; rc = DBG_E_CAPACITY; break; }
        if (sc[0] & 32) { *overflow = true; rc = DBG_E_CAPACITY; h->err = "successor counter overflow"; break; }
        h->n_kmer_inst = sc[1];
        h->n_edge_inst = sc[2];

        Timer tg(h->stream);
        uint64_t total = 0;
        if ((rc = exclusive_scan(h, n_occ, OccBytes32{occ}, word_rank, &total)) != DBG_OK) break;
        if (total >= 0xFFFFFFFFull) { h->err = "more than 2^32-1 nodes"; rc = DBG_E_CAPACITY; break; }
        h->n_nodes = total;
        if ((rc = buf_ensure(h, h->ar_node[0], total * 8)) != DBG_OK) break;
        if ((rc = buf_ensure(h, h->ar_node[1], total * 8)) != DBG_OK) break;
        if ((rc = buf_ensure(h, h->ar_node[2], total * 16)) != DBG_OK) break;
        if ((rc = buf_ensure(h, h->ar_node[3], total)) != DBG_OK) break;
        if ((rc = buf_ensure(h, h->ar_node[4], total)) != DBG_OK) break;
        if ((rc = buf_ensure(h, h->ar_node[5], total * 16)) != DBG_OK) break;
        if ((rc = buf_ensure(h, h->ar_node[6], total)) != DBG_OK) break;
        if ((rc = buf_ensure(h, h->ar_wide[5], total * 8)) != DBG_OK) break;
        h->d_keys = (uint64_t *)h->ar_node[0].p;
        h->d_stamps = (uint64_t *)h->ar_node[1].p;
        h->d_cnt = (uint32_t *)h->ar_node[2].p;
        h->d_flags = (uint8_t *)h->ar_node[3].p;
        h->d_order = (uint8_t *)h->ar_node[4].p;
        h->d_succ = (uint32_t *)h->ar_node[5].p;
        h->d_deg = (uint8_t *)h->ar_node[6].p;
        h->d_keys_hi = (uint64_t *)h->ar_wide[5].p;
        h->nodes_in_arena = true;
        hipLaunchKernelGGL(k_wgather, dim3(grid_for(n_occ * 32, 256)), dim3(256), 0, h->stream, tab, tcnt, occ, word_rank, n_occ,
                           h->d_keys, h->d_keys_hi, h->d_stamps, h->d_cnt, h->d_flags, 1);
        h->stats.ms_compact = tg.stop();
        Timer ts(h->stream);
        if (total)
            hipLaunchKernelGGL(k_wsucc, dim3(grid_for(total, 256)), dim3(256), 0, h->stream, tab, cap, k, total,
                               h->d_keys, h->d_keys_hi, h->d_cnt, h->d_succ, h->d_order, h->d_deg);
        e = hipStreamSynchronize(h->stream);
        h->stats.ms_succ = ts.stop();
        if (e != hipSuccess || hipGetLastError() != hipSuccess) { h->err = std::string("wide build: ") + hipGetErrorString(e); rc = DBG_E_HIP; break; }
    } while (false);
    cleanup();
    return rc;
}

static int build_wide(dbg *h, int k, uint64_t table_capacity_hint) {
    bool overflow = false;
    int rc = build_wide_once(h, k, table_capacity_hint, true, &overflow);
    if (rc != DBG_OK && overflow) {
        rc = build_wide_once(h, k, table_capacity_hint, false, &overflow);
        if (rc == DBG_OK) h->err.clear();
    }
    return rc;
}

extern "C" int dbg_build(dbg_t *h, int k, uint64_t table_capacity_hint) {
    if (!h) return DBG_E_ARG;
    if (k < 1 || k > 63) { h->err = "k must be in 1..63 (k-mers of at most two 64-bit words)"; return DBG_E_ARG; }
    if (!h->d_offsets) { h->err = "no reads set"; return DBG_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    h->dropping_parts = true;
    free_build(h);
    h->dropping_parts = false;
    drop_spare_part(h);
    h->k = k;
    h->stats = dbg_stats_t{};
    // The first build on a read set has to know its alphabet.  Reads over ACGT -- the hot path -- are taken at their word:
    // the extraction kernels check every byte anyway, so the separate pass over the reads (k_g_hist, 1.9 ms per 1.5 GB: 8 %
    // of a build, paid again after every `sequences.extend(...)` of the multi-k driver) runs only when they object.
    const bool tentative = !h->alpha_known && h->engine == 0 && h->n_bytes > 0;
    if (tentative) h->is_dna = true;
    else CHK(compute_alphabet(h));
    if (tentative) {
        Timer t_total(h->stream);
        int rc;
        if (k > 31) {
            bool fallback = true;
            rc = h->wide_engine == 1 ? build_wsk(h, k, &fallback) : DBG_E_CAPACITY;
            if (fallback && rc != DBG_E_ALPHABET) {
                free_build(h);
                h->k = k;
                h->stats = dbg_stats_t{};
                CHK(compute_alphabet(h));  // the global-table engine does not validate the bytes itself
                rc = h->is_dna ? build_wide(h, k, table_capacity_hint) : DBG_E_ALPHABET;
            }
        } else {
            rc = build_sk(h, k, table_capacity_hint);
        }
        if (rc == DBG_OK) {
            rc = finish_graph(h);
            if (rc != DBG_OK) { const std::string keep = h->err; free_build(h); h->err = keep; return rc; }
            h->alpha_known = true;  // ACGT, confirmed byte by byte
            h->is_dna = true;
            h->stats.ms_build_total = t_total.stop();
            if (k <= 31) pool_trim(h);
            return DBG_OK;
        }
        if (rc != DBG_E_ALPHABET) { const std::string keep = h->err; free_build(h); h->err = keep; return rc; }
        free_build(h);  // another alphabet: find out which, then the generic engine below
        h->k = k;
        h->stats = dbg_stats_t{};
        h->err.clear();
        CHK(compute_alphabet(h));
        if (h->is_dna) { h->err = "internal: the extraction refused reads the alphabet scan calls ACGT"; return DBG_E_HIP; }
    }
    if (!h->is_dna) {  // any other alphabet: generic 5-bit engine (peptides: the reference's real inputs)
        Timer t_total(h->stream);
        int rc = build_generic(h, k);
        if (rc == DBG_OK) rc = finish_graph(h);
        if (rc != DBG_OK) { const std::string keep = h->err; free_build(h); h->err = keep; return rc; }
        h->stats.ms_build_total = t_total.stop();
        return DBG_OK;
    }
    if (k > 31) {  // two-word k-mers: the super-k-mer / LDS engine (dbg_wsk.h), or the global reference-keyed table (dbg_wide.h)
        Timer t_total(h->stream);
        int rc;
        bool fallback = true;
        if (h->wide_engine == 1 && h->engine == 0) rc = build_wsk(h, k, &fallback);
        if (fallback) {
            free_build(h);
            h->k = k;
            h->stats = dbg_stats_t{};
            rc = build_wide(h, k, table_capacity_hint);
        }
        if (rc == DBG_OK) rc = finish_graph(h);
        if (rc != DBG_OK) { const std::string keep = h->err; free_build(h); h->err = keep; return rc; }
        h->stats.ms_build_total = t_total.stop();
        return DBG_OK;
    }
    if (h->engine == 0) {
        Timer t_total(h->stream);
        int rc = build_sk(h, k, table_capacity_hint);
        if (rc == DBG_OK) rc = finish_graph(h);
        if (rc != DBG_OK) { const std::string keep = h->err; free_build(h); h->err = keep; return rc; }
        h->stats.ms_build_total = t_total.stop();
        pool_trim(h);  // no-op in the steady state: the arenas only grow
        return DBG_OK;
    }
    Timer t_total(h->stream);

    // table sizing: worst case every window is a distinct k-mer
    uint64_t want = table_capacity_hint ? table_capacity_hint : (uint64_t)((double)(h->n_bytes + 1) / 0.7);
    uint64_t cap = 1024;
    int lg = 10;
    while (cap < want) { cap <<= 1; ++lg; }
    h->cap = cap;
    h->cap_log2 = lg;
    {
        Timer t(h->stream);
        CHK(dev_alloc(h, &h->d_tab, cap));
        CHK(dev_alloc(h, &h->d_occ, cap / 32));
        HIPCHK(h, hipMemsetAsync(h->d_occ, 0, cap / 8, h->stream));
        HIPCHK(h, hipMemsetAsync(h->d_scalars, 0, 64 * 8, h->stream));
        hipLaunchKernelGGL(k_table_init, dim3(grid_for(cap, 256)), dim3(256), 0, h->stream, h->d_tab, cap);
        HIPCHK(h, hipGetLastError());
        h->stats.ms_table_init = t.stop();
    }
    uint64_t sc[4] = {0, 0, 0, 0};
    {
        Timer t(h->stream);
        const uint64_t tiles = (h->n_bytes + TILE - 1) / TILE;
        if (tiles) {
            hipLaunchKernelGGL(k_count, dim3((unsigned)tiles), dim3(256), 0, h->stream, h->d_bases, h->n_bytes,
                               h->d_startbits, k, h->d_tab, cap - 1, 64 - lg, h->d_occ,
                               (unsigned long long *)h->d_scalars);
            HIPCHK(h, hipGetLastError());
        }
        h->stats.count_launches = tiles ? 1 : 0;
        HIPCHK(h, hipMemcpyAsync(sc, h->d_scalars, 32, hipMemcpyDeviceToHost, h->stream));
        h->stats.ms_count = t.stop();
    }
    if (sc[0] & 1) { h->err = "reads hold a byte outside ACGT"; free_build(h); return DBG_E_ALPHABET; }
    if (sc[0] & 2) { h->err = "hash table capacity exceeded"; free_build(h); return DBG_E_CAPACITY; }
    h->n_kmer_inst = sc[1];
    h->n_edge_inst = sc[2];

    // compaction
    {
        Timer t(h->stream);
        uint32_t *word_rank = nullptr;
        const uint64_t n_words = cap / 32;
        CHK(dev_alloc(h, &word_rank, n_words));
        uint64_t total = 0;
        int rc = exclusive_scan(h, n_words, PopcWords{h->d_occ}, word_rank, &total);
        if (rc != DBG_OK) { (void)hipFree(word_rank); return rc; }
        if (total >= 0xFFFFFFFFull) { (void)hipFree(word_rank); h->err = "more than 2^32-1 nodes"; return DBG_E_CAPACITY; }
        h->n_nodes = total;
        CHK(dev_alloc(h, &h->d_keys, total));
        CHK(dev_alloc(h, &h->d_stamps, total));
        CHK(dev_alloc(h, &h->d_cnt, total * 4));
        CHK(dev_alloc(h, &h->d_flags, total));
        CHK(dev_alloc(h, &h->d_order, total));
        CHK(dev_alloc(h, &h->d_succ, total * 4));
        CHK(dev_alloc(h, &h->d_deg, total));
        hipLaunchKernelGGL(k_gather, dim3(grid_for(n_words, 256)), dim3(256), 0, h->stream, h->d_tab, h->d_occ,
                           word_rank, n_words, h->d_keys, h->d_stamps, h->d_cnt, h->d_flags);
        HIPCHK(h, hipGetLastError());
        h->stats.ms_compact = t.stop();
        (void)hipFree(word_rank);
    }
    {
        Timer t(h->stream);
        if (h->n_nodes) {
            hipLaunchKernelGGL(k_succ, dim3(grid_for(h->n_nodes, 256)), dim3(256), 0, h->stream, h->d_tab, cap - 1,
                               64 - lg, k, h->n_nodes, h->d_keys, h->d_cnt, h->d_succ, h->d_order, h->d_deg);
            HIPCHK(h, hipGetLastError());
        }
        h->stats.ms_succ = t.stop();
    }
    CHK(finish_graph(h));
    h->stats.ms_build_total = t_total.stop();
    return DBG_OK;
}

extern "C" int dbg_refine_edge_order(dbg_t *h) {
    if (!h || !h->k) return DBG_E_ARG;
    if (h->partial_graph) { h->err = kPartialGraph; return DBG_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    CHK(ensure_dense(h));
    if (h->D == GEN_D) return DBG_OK;  // the generic engine ranks by per-edge first-seen positions at build time
    dev_free(h->d_fsorder);
    CHK(dev_alloc(h, &h->d_fsorder, h->n_nodes));
    if (!h->n_nodes) { h->order_exact = true; return DBG_OK; }
    hipLaunchKernelGGL(k_fsorder_default, dim3(grid_for(h->n_nodes, 256)), dim3(256), 0, h->stream, h->n_nodes, h->d_order,
                       h->d_fsorder);
    if (h->sk_src.valid && !h->d_keys_hi && h->sk_n_ranges && !h->refine_streaming) {  // partitioned build: per range, from the bucket's own records
        unsigned long long *flag = (unsigned long long *)(h->d_scalars + 48);
        HIPCHK(h, hipMemsetAsync(flag, 0, 8, h->stream));
        const SkRange *ranges = (const SkRange *)h->ar_misc[6].p;
        if (h->sk_src.st_bytes == 4)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sk_refine<uint32_t>), dim3((unsigned)h->sk_n_ranges), dim3(256), 0, h->stream, ranges,
                               h->sk_src.b_start, h->sk_src.b_cnt, h->sk_src.w0, h->sk_src.w1, (const uint32_t *)h->sk_src.st,
                               h->k, h->d_keys, h->d_cnt, h->d_order, h->d_fsorder, flag);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sk_refine<uint64_t>), dim3((unsigned)h->sk_n_ranges), dim3(256), 0, h->stream, ranges,
                               h->sk_src.b_start, h->sk_src.b_cnt, h->sk_src.w0, h->sk_src.w1, (const uint64_t *)h->sk_src.st,
                               h->k, h->d_keys, h->d_cnt, h->d_order, h->d_fsorder, flag);
        HIPCHK(h, hipGetLastError());
        unsigned long long crowded = 0;
        HIPCHK(h, hipMemcpyAsync(&crowded, flag, 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (!crowded) { h->order_exact = true; return DBG_OK; }
        // a range with hundreds of multi-successor nodes: the streaming pass below recomputes every node
    }
    uint64_t n_multi = 0;
    CHK(reduce_sum(h, h->n_nodes, MultiSucc{h->d_cnt}, &n_multi));
    if (n_multi && h->d_keys_hi) {  // two-word k-mers: the set holds node ids, keys are compared by reference
        uint64_t cap = 1024;
        while (cap < n_multi * 2) cap <<= 1;
        unsigned long long *estamp = nullptr;
        uint32_t *set_node = nullptr;
        int rc = dev_alloc(h, &set_node, cap);
        if (rc == DBG_OK) rc = dev_alloc(h, &estamp, cap * 4);
        if (rc == DBG_OK) {
            (void)hipMemsetAsync(set_node, 0xFF, cap * 4, h->stream);
            (void)hipMemsetAsync(estamp, 0xFF, cap * 32, h->stream);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wset_insert<WSelMulti>), dim3(grid_for(h->n_nodes, 256)), dim3(256), 0, h->stream,
                               h->n_nodes, WSelMulti{h->d_cnt}, h->d_keys, h->d_keys_hi, set_node, cap - 1);
            const uint64_t tiles = (h->n_bytes + TILE - 1) / TILE;
            hipLaunchKernelGGL(k_wedge_first_seen, dim3((unsigned)tiles), dim3(256), 0, h->stream, h->d_bases, h->n_bytes,
                               h->d_startbits, h->k, set_node, cap - 1, h->d_keys, h->d_keys_hi, estamp);
            hipLaunchKernelGGL(k_order_refine, dim3(grid_for(cap, 256)), dim3(256), 0, h->stream, cap,
                               (const unsigned long long *)nullptr, set_node, estamp, h->d_cnt, h->d_order, h->d_fsorder);
            hipError_t e = hipStreamSynchronize(h->stream);
            if (e != hipSuccess) { h->err = std::string("refine: ") + hipGetErrorString(e); rc = DBG_E_HIP; }
        }
        dev_free(set_node); dev_free(estamp);
        if (rc != DBG_OK) return rc;
    } else if (n_multi) {
        uint64_t cap = 1024;
        while (cap < n_multi * 2) cap <<= 1;
        unsigned long long *set_keys = nullptr, *estamp = nullptr;
        uint32_t *set_node = nullptr, *filter = nullptr;
        int fbits = 20;
        while (fbits < 32 && (1ull << fbits) < n_multi * 16) ++fbits;
        // grow-only arena (a multi-k driver calls this once per k: ~700 MB of hipMalloc + hipFree each time otherwise)
        int rc = buf_ensure(h, h->ar_refine[0], cap * 8);
        if (rc == DBG_OK) rc = buf_ensure(h, h->ar_refine[1], cap * 4);
        if (rc == DBG_OK) rc = buf_ensure(h, h->ar_refine[2], cap * 32);
        if (rc == DBG_OK) rc = buf_ensure(h, h->ar_refine[3], (1ull << fbits) / 8);
        if (rc == DBG_OK) {
            set_keys = (unsigned long long *)h->ar_refine[0].p;
            set_node = (uint32_t *)h->ar_refine[1].p;
            estamp = (unsigned long long *)h->ar_refine[2].p;
            filter = (uint32_t *)h->ar_refine[3].p;
            (void)hipMemsetAsync(set_keys, 0xFF, cap * 8, h->stream);
            (void)hipMemsetAsync(estamp, 0xFF, cap * 32, h->stream);
            (void)hipMemsetAsync(filter, 0, (1ull << fbits) / 8, h->stream);
            hipLaunchKernelGGL(k_multi_insert, dim3(grid_for(h->n_nodes, 256)), dim3(256), 0, h->stream, h->n_nodes, h->d_cnt,
                               h->d_keys, set_keys, set_node, cap - 1, filter, fbits);
            const uint64_t tiles = (h->n_bytes + TILE - 1) / TILE;
            hipLaunchKernelGGL(k_edge_first_seen, dim3((unsigned)tiles), dim3(256), 0, h->stream, h->d_bases, h->n_bytes,
                               h->d_startbits, h->k, (const uint64_t *)set_keys, cap - 1, filter, fbits, estamp);
            hipLaunchKernelGGL(k_order_refine, dim3(grid_for(cap, 256)), dim3(256), 0, h->stream, cap, set_keys, set_node,
                               estamp, h->d_cnt, h->d_order, h->d_fsorder);
            hipError_t e = hipStreamSynchronize(h->stream);
            if (e != hipSuccess) { h->err = std::string("refine: ") + hipGetErrorString(e); rc = DBG_E_HIP; }
        }
        if (rc != DBG_OK) return rc;
    }
    h->order_exact = true;
    return DBG_OK;
}

extern "C" int dbg_export_orders(dbg_t *h, uint8_t *order, uint8_t *fsorder) {
    if (!h || !h->k) return DBG_E_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    CHK(ensure_dense(h));
    if (h->D == GEN_D) {  // one byte per rank: [n_nodes][32] successor codes, 0xFF beyond the out-degree
        if (order && h->n_nodes) HIPCHK(h, hipMemcpyAsync(order, h->d_rank_mc, h->n_nodes * GEN_D, hipMemcpyDeviceToHost, h->stream));
        if (fsorder && h->n_nodes) HIPCHK(h, hipMemcpyAsync(fsorder, h->d_rank_fs, h->n_nodes * GEN_D, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return DBG_OK;
    }
    if (fsorder && !h->d_fsorder) { h->err = "dbg_refine_edge_order must run first"; return DBG_E_ARG; }
    if (order && h->n_nodes) HIPCHK(h, hipMemcpyAsync(order, h->d_order, h->n_nodes, hipMemcpyDeviceToHost, h->stream));
    if (fsorder && h->n_nodes) HIPCHK(h, hipMemcpyAsync(fsorder, h->d_fsorder, h->n_nodes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return DBG_OK;
}

__global__ __launch_bounds__(256) void k_iota32(uint64_t n, uint32_t *out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (uint32_t)i;
}

// Node ids in the reference's dict order (first occurrence of the k-mer in the reads, debruijn.py:120-133): one device
// radix sort of the stamps instead of a host argsort of 10^7..10^8 elements.
extern "C" int dbg_export_dict_order(dbg_t *h, uint32_t *order) {
    if (!h || !h->k || !order) { if (h) h->err = "dbg_build must run first"; return DBG_E_ARG; }
    // a shard's stamps are global positions: the order of one shard alone is not the dict order of anything
    if (h->partial_graph) { h->err = kPartialGraph; return DBG_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    CHK(ensure_dense(h));
    const uint64_t n = h->n_nodes;
    if (!n) return DBG_OK;
    if (n > 0x7FFFFFFFull) { h->err = "too many nodes for one sort"; return DBG_E_CAPACITY; }
    uint64_t *keys_out = nullptr;
    uint32_t *ids = nullptr, *ids_out = nullptr;
    void *tmp = nullptr;
    int rc = DBG_OK;
    do {
        if ((rc = dev_alloc(h, &keys_out, n)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &ids, n)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &ids_out, n)) != DBG_OK) break;
        hipLaunchKernelGGL(k_iota32, dim3(grid_for(n, 256)), dim3(256), 0, h->stream, n, ids);
        int end_bit = 2;  // stamp = position << 1 | flag, position < n_bytes
        while (end_bit < 64 && (h->n_bytes >> (end_bit - 1))) ++end_bit;
        size_t tmp_bytes = 0;
        hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, h->d_stamps, keys_out, ids, ids_out, (size_t)n, 0u,
                                                 (unsigned)end_bit, h->stream);
        if (e == hipSuccess) e = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 1);
        if (e == hipSuccess)
            e = rocprim::radix_sort_pairs(tmp, tmp_bytes, h->d_stamps, keys_out, ids, ids_out, (size_t)n, 0u, (unsigned)end_bit,
                                          h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(order, ids_out, n * 4, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) { h->err = std::string("dict order: ") + hipGetErrorString(e); rc = DBG_E_HIP; }
    } while (0);
    if (tmp) (void)hipFree(tmp);
    dev_free(keys_out); dev_free(ids); dev_free(ids_out);
    return rc;
}

// rows of the nodes that carry `flag`, in the reference's list order, with their k-mers: the pulled nodes in pull order
// (already_pull_out, debruijn.py:253) or the branch nodes in dict order (branch_kmer, debruijn.py:230-236) -- compacted,
// sorted and gathered on the device, so a caller that only needs these two short lists moves nothing of size n_nodes
__global__ __launch_bounds__(256) void k_gather_u64(const uint32_t *__restrict__ ids, uint64_t n, const uint64_t *__restrict__ src,
                                                    uint64_t *dst) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[ids[i]];
}

extern "C" int dbg_export_marked(dbg_t *h, uint32_t flag, uint64_t capacity, uint64_t *n_out, uint32_t *rows, uint64_t *keys,
                                 uint64_t *keys_hi) {
    if (!h || !h->k || !n_out) { if (h) h->err = "dbg_build must run first"; return DBG_E_ARG; }
    if (flag != DBG_F_PULLED && flag != DBG_F_BRANCH) { h->err = "flag must be DBG_F_PULLED or DBG_F_BRANCH"; return DBG_E_ARG; }
    if (flag == DBG_F_PULLED && !h->tipped) { h->err = "dbg_remove_tips must run first"; return DBG_E_ARG; }
    if (flag == DBG_F_BRANCH && !h->pruned) { h->err = "dbg_prune must run first"; return DBG_E_ARG; }
    if (h->partial_graph) { h->err = kPartialGraph; return DBG_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    CHK(ensure_dense(h));
    *n_out = 0;
    const uint64_t n = h->n_nodes;
    if (!n) return DBG_OK;
    if ((keys || keys_hi) && !h->d_keys) {  // tables keyed by reference into the reads (generic alphabet, k >= 12) hold no packed k-mers
        h->err = "dbg_export_marked: this graph has no packed keys (pass NULL for keys and spell the rows from the reads)";
        return DBG_E_ARG;
    }
    if (!h->d_flags || !h->d_stamps || (flag == DBG_F_PULLED && !h->d_pull_rank)) { h->err = "dbg_export_marked: node arrays missing"; return DBG_E_ARG; }
    uint32_t *ids = nullptr, *ids_sorted = nullptr;
    uint64_t *skey = nullptr, *skey_sorted = nullptr, *gathered = nullptr;
    void *tmp = nullptr;
    int rc = DBG_OK;
    do {
        uint64_t found = 0;
        if ((rc = compact_ids(h, n, PredFlags{h->d_flags, (uint8_t)flag, (uint8_t)flag}, (uint32_t *)nullptr, &found)) != DBG_OK) break;
        *n_out = found;
        if (!found || !rows) break;
        if (found > capacity) { h->err = "dbg_export_marked: capacity too small"; rc = DBG_E_CAPACITY; break; }
        if ((rc = dev_alloc(h, &ids, found)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &ids_sorted, found)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &skey, found)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &skey_sorted, found)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &gathered, found)) != DBG_OK) break;
        uint64_t again = 0;
        if ((rc = compact_ids(h, n, PredFlags{h->d_flags, (uint8_t)flag, (uint8_t)flag}, ids, &again)) != DBG_OK) break;
        const dim3 grid(grid_for(found, 256));
        hipLaunchKernelGGL(k_gather_u64, grid, dim3(256), 0, h->stream, ids, found, flag == DBG_F_PULLED ? h->d_pull_rank : h->d_stamps, skey);
        size_t tmp_bytes = 0;
        hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, skey, skey_sorted, ids, ids_sorted, (size_t)found, 0u, 64u, h->stream);
        if (e == hipSuccess) e = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 1);
        if (e == hipSuccess) e = rocprim::radix_sort_pairs(tmp, tmp_bytes, skey, skey_sorted, ids, ids_sorted, (size_t)found, 0u, 64u, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(rows, ids_sorted, found * 4, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess && keys) {
            hipLaunchKernelGGL(k_gather_u64, grid, dim3(256), 0, h->stream, ids_sorted, found, h->d_keys, gathered);
            e = hipMemcpyAsync(keys, gathered, found * 8, hipMemcpyDeviceToHost, h->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e == hipSuccess && keys_hi) {
            if (h->d_keys_hi) {
                hipLaunchKernelGGL(k_gather_u64, grid, dim3(256), 0, h->stream, ids_sorted, found, h->d_keys_hi, gathered);
                e = hipMemcpyAsync(keys_hi, gathered, found * 8, hipMemcpyDeviceToHost, h->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
            } else {
                memset(keys_hi, 0, found * 8);
            }
        }
        if (e != hipSuccess) { h->err = std::string("dbg_export_marked: ") + hipGetErrorString(e); rc = DBG_E_HIP; }
    } while (0);
    if (tmp) (void)hipFree(tmp);
    dev_free(ids); dev_free(ids_sorted); dev_free(skey); dev_free(skey_sorted); dev_free(gathered);
    return rc;
}

__global__ __launch_bounds__(256) void k_keepmask_from_flags(uint64_t n, const uint8_t *flags, uint32_t *out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (uint32_t)(flags[i] & DBG_F_KEEP_MASK) >> DBG_F_KEEP_SHIFT;
}

extern "C" int dbg_export_keepmask(dbg_t *h, uint32_t *keepmask) {
    if (!h || !h->pruned || !keepmask) { if (h) h->err = "dbg_prune must run first"; return DBG_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    CHK(ensure_dense(h));
    if (!h->n_nodes) return DBG_OK;
    if (h->D == GEN_D) {
        HIPCHK(h, hipMemcpyAsync(keepmask, h->d_keepmask, h->n_nodes * 4, hipMemcpyDeviceToHost, h->stream));
    } else {
        uint32_t *tmp = nullptr;
        CHK(dev_alloc(h, &tmp, h->n_nodes));
        hipLaunchKernelGGL(k_keepmask_from_flags, dim3(grid_for(h->n_nodes, 256)), dim3(256), 0, h->stream, h->n_nodes,
                           h->d_flags, tmp);
        hipError_t e = hipMemcpyAsync(keepmask, tmp, h->n_nodes * 4, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        (void)hipFree(tmp);
        if (e != hipSuccess) { h->err = hipGetErrorString(e); return DBG_E_HIP; }
        return DBG_OK;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return DBG_OK;
}

extern "C" int dbg_get_alphabet(dbg_t *h, char *codes32, int *n_symbols, int *bits_per_symbol) {
    if (!h || !h->k) return DBG_E_ARG;
    if (codes32) {
        memset(codes32, 0, 32);
        if (h->D == GEN_D) memcpy(codes32, h->alphabet, 32);
        else memcpy(codes32, "ACTG", 4);  // code = (ascii >> 1) & 3
    }
    if (n_symbols) *n_symbols = h->D == GEN_D ? h->n_sym : 4;
    if (bits_per_symbol) *bits_per_symbol = h->sym_bits;
    return DBG_OK;
}

extern "C" int dbg_prune(dbg_t *h, double threshold) {
    if (!h || !h->k) return DBG_E_ARG;
    if (threshold == 0.0) { h->err = "threshold must be non-zero (the reference divides by it)"; return DBG_E_ARG; }
    if (h->partial_graph) { h->err = kPartialGraph; return DBG_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    CHK(ensure_dense(h));
    Timer t(h->stream);
    HIPCHK(h, hipMemsetAsync(h->d_scalars + 16, 0, 8, h->stream));
    if (h->n_nodes && h->D == GEN_D) {
        hipLaunchKernelGGL(k_g_prune, dim3(grid_for(h->n_nodes, 256)), dim3(256), 0, h->stream, h->n_nodes, h->d_cnt,
                           h->d_rank_mc, h->d_deg, threshold, h->d_keepmask, h->d_flags,
                           (unsigned long long *)(h->d_scalars + 16));
        HIPCHK(h, hipGetLastError());
    } else if (h->n_nodes) {
        hipLaunchKernelGGL(k_prune, dim3(grid_for(h->n_nodes, 256)), dim3(256), 0, h->stream, h->n_nodes, h->d_cnt,
                           h->d_order, threshold, h->d_flags, (unsigned long long *)(h->d_scalars + 16));
        HIPCHK(h, hipGetLastError());
    }
    HIPCHK(h, hipMemcpyAsync(&h->n_branch, h->d_scalars + 16, 8, hipMemcpyDeviceToHost, h->stream));
    h->stats.ms_prune = t.stop();
    h->pruned = true;
    h->tipped = false;
    h->n_pulled = 0;
    h->pull_reads_done = false;
    h->walked = h->walk_indexed = false;
    return DBG_OK;
}

template <class G>
static int remove_tips_impl(dbg *h, const G &g) {
    Timer t(h->stream);
    h->tip_rounds = 0;
    h->n_pulled = 0;
    // grow-only arena: a hipFree + hipMalloc of the two n_nodes-sized arrays per call cost up to 150 ms at 3.6e8 nodes
    CHK(buf_ensure(h, h->ar_tips[0], (h->n_nodes ? h->n_nodes : 1) * 8));
    h->d_pull_rank = (uint64_t *)h->ar_tips[0].p;
    if (h->n_nodes)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fill<uint64_t>), dim3(grid_for(h->n_nodes, 256)), dim3(256), 0, h->stream,
                           h->d_pull_rank, h->n_nodes, ~0ull);
    if (h->n_branch) {
        CHK(buf_ensure(h, h->ar_tips[1], h->n_branch * 4));
        CHK(buf_ensure(h, h->ar_tips[2], h->n_branch * 4));
        CHK(buf_ensure(h, h->ar_tips[3], h->n_nodes * 8));
        uint32_t *pend[2] = {(uint32_t *)h->ar_tips[1].p, (uint32_t *)h->ar_tips[2].p};
        unsigned long long *owner = (unsigned long long *)h->ar_tips[3].p;
        unsigned long long *ctr = (unsigned long long *)(h->d_scalars + 24);
        HIPCHK(h, hipMemsetAsync(ctr, 0, 16, h->stream));
        uint64_t n_pending = 0;  // branch nodes, ascending id
        CHK(compact_ids(h, h->n_nodes, PredFlags{h->d_flags, (uint8_t)DBG_F_BRANCH, (uint8_t)DBG_F_BRANCH}, pend[0], &n_pending));
        if (n_pending != h->n_branch) { h->err = "internal: branch count changed"; return DBG_E_HIP; }
        int cur = 0;
        while (n_pending) {
            ++h->tip_rounds;
            HIPCHK(h, hipMemsetAsync(ctr, 0, 8, h->stream));
            const dim3 grid(grid_for(n_pending, 256));
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tip_reset<G>), grid, dim3(256), 0, h->stream, g, pend[cur], n_pending, owner);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tip_claim<G>), grid, dim3(256), 0, h->stream, g, pend[cur], n_pending, h->d_stamps, owner);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tip_commit<G>), grid, dim3(256), 0, h->stream, g, h->d_flags, pend[cur], n_pending,
                               h->d_stamps, owner, (unsigned long long *)h->d_pull_rank, pend[cur ^ 1], ctr);
            uint64_t c2[2];
            HIPCHK(h, hipMemcpyAsync(c2, ctr, 16, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            if (c2[0] >= n_pending) { h->err = "tip removal made no progress"; return DBG_E_HIP; }
            n_pending = c2[0];
            h->n_pulled = c2[1];
            cur ^= 1;
        }
    }
    h->stats.ms_tips = t.stop();
    h->tipped = true;
    h->walked = h->walk_indexed = false;
    return DBG_OK;
}

static GDna dna_view(const dbg *h) { return GDna{h->d_keys, h->d_keys_hi, h->d_flags, h->d_order, h->d_succ, h->d_cnt, h->k}; }

extern "C" int dbg_remove_tips(dbg_t *h) {
    if (!h || !h->pruned) { if (h) h->err = "dbg_prune must run first"; return DBG_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    CHK(ensure_dense(h));
    if (h->D == GEN_D) return remove_tips_impl(h, gen_view(h));
    return remove_tips_impl(h, dna_view(h));
}

extern "C" int dbg_mark_pull_reads(dbg_t *h) {
    if (!h || !h->pruned) { if (h) h->err = "dbg_prune must run first"; return DBG_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    CHK(ensure_dense(h));
    Timer t(h->stream);
    dev_free(h->d_read_flags);
    CHK(dev_alloc(h, &h->d_read_flags, h->n_reads));
    HIPCHK(h, hipMemsetAsync(h->d_read_flags, 0, h->n_reads ? h->n_reads : 1, h->stream));
    h->n_pull_reads = 0;
    if (h->n_branch && h->n_bytes) {
        const uint64_t tiles = (h->n_bytes + TILE - 1) / TILE;
        uint64_t bcap = 1024;
        while (bcap < h->n_branch * 2) bcap <<= 1;
        dev_free(h->d_btab);
        CHK(dev_alloc(h, &h->d_btab, bcap));
        h->btab_cap = bcap;
        HIPCHK(h, hipMemsetAsync(h->d_btab, 0xFF, bcap * 8, h->stream));
        if (h->d_keys_hi) {  // two-word k-mers: set of branch node ids (32-bit entries in the same buffer)
            uint32_t *set = (uint32_t *)h->d_btab;
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wset_insert<WSelBranch>), dim3(grid_for(h->n_nodes, 256)), dim3(256), 0, h->stream,
                               h->n_nodes, WSelBranch{h->d_flags}, h->d_keys, h->d_keys_hi, set, bcap - 1);
            hipLaunchKernelGGL(k_wpull_reads, dim3((unsigned)tiles), dim3(256), 0, h->stream, h->d_bases, h->n_bytes,
                               h->d_startbits, h->k, set, bcap - 1, h->d_keys, h->d_keys_hi, h->d_offsets, h->n_reads,
                               h->d_read_flags);
        } else if (h->D == GEN_D && !h->d_keys) {  // generic alphabet, k >= 12: set of branch node ids, keys by reference
            uint32_t *set = (uint32_t *)h->d_btab;
            hipLaunchKernelGGL(k_gr_set_insert, dim3(grid_for(h->n_nodes, 256)), dim3(256), 0, h->stream, h->n_nodes, h->d_flags,
                               h->d_bases, h->d_stamps, h->k, set, bcap - 1);
            hipLaunchKernelGGL(k_gr_pull_reads, dim3((unsigned)std::min<uint64_t>(tiles * 32, 1u << 16)), dim3(256), 0, h->stream,
                               h->d_bases, h->n_bytes, h->d_startbits, h->k, set, bcap - 1, h->d_stamps, h->d_offsets,
                               h->n_reads, h->d_read_flags);
        } else {
        hipLaunchKernelGGL(k_branch_insert, dim3(grid_for(h->n_nodes, 256)), dim3(256), 0, h->stream, h->n_nodes,
                           h->d_flags, h->d_keys, (unsigned long long *)h->d_btab, bcap - 1);
        bool by_range = false;
        if (h->D != GEN_D && h->sk_src.valid && h->sk_n_ranges && !h->refine_streaming) {  // partitioned build: per range
            unsigned long long *flag = (unsigned long long *)(h->d_scalars + 48);
            HIPCHK(h, hipMemsetAsync(flag, 0, 8, h->stream));
            const SkRange *ranges = (const SkRange *)h->ar_misc[6].p;
            if (h->sk_src.st_bytes == 4)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sk_pull<uint32_t>), dim3((unsigned)h->sk_n_ranges), dim3(256), 0, h->stream,
                                   ranges, h->sk_src.b_start, h->sk_src.b_cnt, h->sk_src.w0, h->sk_src.w1,
                                   (const uint32_t *)h->sk_src.st, h->k, h->d_keys, h->d_flags, h->d_offsets, h->n_reads,
                                   h->d_read_flags, flag);
            else
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sk_pull<uint64_t>), dim3((unsigned)h->sk_n_ranges), dim3(256), 0, h->stream,
                                   ranges, h->sk_src.b_start, h->sk_src.b_cnt, h->sk_src.w0, h->sk_src.w1,
                                   (const uint64_t *)h->sk_src.st, h->k, h->d_keys, h->d_flags, h->d_offsets, h->n_reads,
                                   h->d_read_flags, flag);
            hipLaunchKernelGGL(k_pull_len_k, dim3(grid_for(h->n_reads, 256)), dim3(256), 0, h->stream, h->d_bases, h->d_offsets,
                               h->n_reads, h->k, h->d_btab, bcap - 1, h->d_read_flags);
            HIPCHK(h, hipGetLastError());
            unsigned long long crowded = 0;
            HIPCHK(h, hipMemcpyAsync(&crowded, flag, 8, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            by_range = !crowded;  // else: the pass over the reads below marks every read again (flags only ever go to 1)
        }
        if (by_range) {
        } else if (h->D == GEN_D)
            hipLaunchKernelGGL(k_g_pull_reads, dim3((unsigned)std::min<uint64_t>(tiles * 32, 1u << 16)), dim3(256), 0, h->stream,
                               h->d_bases, h->n_bytes, h->d_startbits, h->k, h->d_lut, h->d_btab, bcap - 1, h->d_offsets,
                               h->n_reads, h->d_read_flags);
        else
            hipLaunchKernelGGL(k_pull_reads, dim3((unsigned)tiles), dim3(256), 0, h->stream, h->d_bases, h->n_bytes,
                               h->d_startbits, h->k, h->d_btab, bcap - 1, h->d_offsets, h->n_reads, h->d_read_flags);
        }
        HIPCHK(h, hipGetLastError());
        uint64_t total = 0;
        CHK(reduce_sum(h, h->n_reads, ByteAt{h->d_read_flags}, &total));
        h->n_pull_reads = total;
    }
    h->stats.ms_pull_reads = t.stop();
    h->pull_reads_done = true;
    return DBG_OK;
}

template <class G>
static int walk_impl(dbg *h, const G &g, int final_mode, uint64_t max_chars) {
    if (!max_chars) max_chars = 1ull << 30;
    // without a branch node every node has at most one surviving successor: "all simple paths" of the final mode ARE the
    // chain walks, and those have the parallel path (the per-start DFS took 2 s on BASELINE configs[0] without errors)
    if (final_mode && h->n_branch == 0) final_mode = 0;
    Timer t(h->stream);
    dev_free(h->d_ctg_off); dev_free(h->d_ctg_chars); dev_free(h->d_ctg_score); dev_free(h->d_ctg_stamp);
    dev_free(h->d_ctg_seq); dev_free(h->d_ctg_start); dev_free(h->d_lift);
    h->lift_levels = 0;
    h->n_contigs = h->contig_chars = 0;
    h->walked = false;
    h->walk_indexed = false;
    { const int rs = ensure_starts(h); if (rs != DBG_OK) return rs; }
    const uint64_t ns = h->n_starts;
    const bool use_jump = !final_mode && h->n_nodes >= h->walk_jump_min && ns;
    uint32_t *starts = nullptr;
    uint64_t *per_ctg = nullptr, *per_chr = nullptr, *base_ctg = nullptr, *base_chr = nullptr, *per_score = nullptr;
    uint32_t *st_node = nullptr, *onpath = nullptr, *ctg_start = nullptr, *lift = nullptr;
    uint8_t *st_next = nullptr;
    Jump *jump[2] = {nullptr, nullptr};
    auto cleanup = [&]() {
        dev_free(starts); dev_free(per_ctg); dev_free(per_chr); dev_free(base_ctg); dev_free(base_chr);
        dev_free(st_node); dev_free(onpath); dev_free(st_next); dev_free(per_score); dev_free(lift);
        jump[0] = jump[1] = nullptr;  // arena-owned
    };
    int rc = DBG_OK;
    do {
        if ((rc = dev_alloc(h, &starts, ns)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &per_ctg, ns)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &per_chr, ns)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &base_ctg, ns)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &base_chr, ns)) != DBG_OK) break;
        if (h->n_nodes) {  // nodes with indegree 0, ascending id
            uint64_t found = 0;
            if ((rc = compact_ids(h, h->n_nodes, PredFlags{h->d_flags, (uint8_t)DBG_F_INDEG, (uint8_t)0}, starts, &found)) != DBG_OK) break;
            if (found != ns) { h->err = "internal: start count changed"; rc = DBG_E_HIP; break; }
        }
        uint64_t n_walkers = 0, stride = 0, bm_words = 0;
        if (final_mode && ns) {
            stride = h->n_nodes + 1;
            bm_words = (h->n_nodes + 31) / 32;
            const uint64_t per_walker = stride * 5 + bm_words * 4;
            n_walkers = std::min<uint64_t>(ns, std::max<uint64_t>(1, (4ull << 30) / per_walker));
            n_walkers = std::min<uint64_t>(n_walkers, 1u << 16);
            if ((rc = dev_alloc(h, &st_node, n_walkers * stride)) != DBG_OK) break;
            if ((rc = dev_alloc(h, &st_next, n_walkers * stride)) != DBG_OK) break;
            if ((rc = dev_alloc(h, &onpath, n_walkers * bm_words)) != DBG_OK) break;
            (void)hipMemsetAsync(onpath, 0, n_walkers * bm_words * 4, h->stream);
        }
        auto launch = [&](int pass) {
            if (!ns) return;
            if (final_mode)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_walk_dfs<G>), dim3(grid_for(n_walkers, 64)), dim3(64), 0, h->stream, g, starts, ns, pass,
                                   n_walkers, stride, st_node, st_next, onpath, bm_words, per_ctg, per_chr, base_ctg,
                                   base_chr, h->d_ctg_off, h->d_ctg_chars, h->d_ctg_score, h->d_ctg_stamp,
                                   h->d_ctg_seq, h->d_stamps, false);
            else
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_walk_chain<G>), dim3(grid_for(ns, 256)), dim3(256), 0, h->stream, g, starts, ns, pass,
                                   per_ctg, per_chr, base_ctg, base_chr, h->d_ctg_off, h->d_ctg_chars,
                                   h->d_ctg_score, h->d_ctg_stamp, h->d_ctg_seq, h->d_stamps);
        };
        if (use_jump) {
            if ((rc = dev_alloc(h, &per_score, ns)) != DBG_OK) break;
            // the one-step table, 16 B per node: kept in the arena (a fresh 6 GB hipMalloc costs ~0.2 s at the BASELINE size)
            if ((rc = buf_ensure(h, h->ar_walk[0], h->n_nodes * sizeof(Jump))) != DBG_OK) break;
            jump[0] = (Jump *)h->ar_walk[0].p;
            const dim3 grid(grid_for(h->n_nodes, 256));
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_jump_init<G>), grid, dim3(256), 0, h->stream, h->n_nodes, g, jump[0]);
            // splitters (see k_jump_walk): their list, their walks into a table that is dense over the splitters,
            // then doubling over that table
            if ((rc = buf_ensure(h, h->ar_walk[2], h->n_nodes * 4)) != DBG_OK) break;
            uint32_t *slist = (uint32_t *)h->ar_walk[2].p;
            uint64_t n_split = 0;
            if ((rc = compact_ids(h, h->n_nodes, PredSplitter{jump[0]}, slist, &n_split)) != DBG_OK) break;
            if ((rc = buf_ensure(h, h->ar_walk[1], (2 * n_split + 1) * sizeof(Jump) + (ns + 4) * 4)) != DBG_OK) break;
            Jump *dense[2] = {(Jump *)h->ar_walk[1].p, (Jump *)h->ar_walk[1].p + n_split};
            uint32_t *start_rank = (uint32_t *)((Jump *)h->ar_walk[1].p + 2 * n_split);
            const dim3 sgrid(grid_for(n_split, 256));
            if (n_split)
                hipLaunchKernelGGL(k_jump_walk, sgrid, dim3(256), 0, h->stream, slist, n_split, jump[0], dense[0]);
            (void)hipMemsetAsync(h->d_scalars, 0, 8, h->stream);
            hipLaunchKernelGGL(k_jump_start_ranks, dim3(grid_for(ns, 256)), dim3(256), 0, h->stream, starts, ns, slist, n_split,
                               start_rank, (unsigned long long *)h->d_scalars);
            int cur = 0, max_rounds = 2;
            while ((1ull << (max_rounds - 1)) < n_split) ++max_rounds;  // a chain holds fewer splitters than there are
            for (int round = 0; round < max_rounds; ++round) {
                uint64_t open = 0;
                if ((round & 1) == 0) {  // every other round: the test costs as much as a round
                    if ((rc = reduce_sum(h, ns, UnresolvedStart{starts, start_rank, dense[cur], h->d_flags}, &open)) != DBG_OK) break;
                    if (!open) break;
                }
                hipLaunchKernelGGL(k_jump_step_dense, sgrid, dim3(256), 0, h->stream, n_split, dense[cur], dense[cur ^ 1]);
                cur ^= 1;
            }
            if (rc != DBG_OK) break;
            hipLaunchKernelGGL(k_jump_starts, dim3(grid_for(ns, 256)), dim3(256), 0, h->stream, starts, start_rank, ns, dense[cur],
                               h->d_flags, h->k, per_ctg, per_chr, per_score);
            {
                uint64_t sc0 = 0;
                if (hipMemcpyAsync(&sc0, h->d_scalars, 8, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
                    hipStreamSynchronize(h->stream) != hipSuccess) { h->err = "walk: list ranking failed on the device"; rc = DBG_E_HIP; break; }
                if (sc0 & 1) { h->err = "internal: a start node is not a splitter"; rc = DBG_E_HIP; break; }
            }
            jump[0] = jump[1] = nullptr;  // arena-owned
        } else {
            launch(0);
        }
        if (hipGetLastError() != hipSuccess) { h->err = "walk pass 0 launch failed"; rc = DBG_E_HIP; break; }
        uint64_t n_ctg = 0, n_chr = 0;
        if ((rc = exclusive_scan(h, ns, U64At{per_ctg}, base_ctg, &n_ctg)) != DBG_OK) break;
        if ((rc = exclusive_scan(h, ns, U64At{per_chr}, base_chr, &n_chr)) != DBG_OK) break;
        h->n_contigs = n_ctg;
        h->contig_chars = n_chr;
        if (n_chr > max_chars && !use_jump) {
            h->err = "contig text exceeds max_chars (sizes are valid, text not materialised)";
            rc = DBG_E_CAPACITY;
            break;
        }
        if ((rc = dev_alloc(h, &h->d_ctg_off, n_ctg + 1)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &h->d_ctg_score, n_ctg)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &h->d_ctg_stamp, n_ctg)) != DBG_OK) break;
        if ((rc = dev_alloc(h, &h->d_ctg_seq, n_ctg)) != DBG_OK) break;
        (void)hipMemcpyAsync(h->d_ctg_off + n_ctg, &h->contig_chars, 8, hipMemcpyHostToDevice, h->stream);
        bool text_done = false;
        if (use_jump) {  // the index (offsets, scores, start stamps) comes straight from the jump table
            if ((rc = dev_alloc(h, &h->d_ctg_start, n_ctg)) != DBG_OK) break;
            ctg_start = h->d_ctg_start;
            hipLaunchKernelGGL(k_walk_desc, dim3(grid_for(ns, 256)), dim3(256), 0, h->stream, starts, ns, per_ctg, base_ctg,
                               base_chr, per_score, h->d_stamps, h->d_ctg_off, h->d_ctg_score, h->d_ctg_stamp, h->d_ctg_seq,
                               ctg_start);
            hipError_t e = hipStreamSynchronize(h->stream);
            if (e != hipSuccess) { h->err = std::string("walk index: ") + hipGetErrorString(e); rc = DBG_E_HIP; break; }
            h->walk_indexed = true;
            if (n_chr > max_chars) break;  // index only: the text would not fit (rc stays DBG_OK)
            // text by binary lifting if its tables fit next to everything else (else: one thread per start below)
            const uint64_t longest = h->n_nodes + (uint64_t)h->k;  // no contig is longer: it visits a node at most once
            int levels = 1;
            while ((1ull << levels) <= longest) ++levels;
            if (n_ctg && (uint64_t)levels * h->n_nodes * 4 <= (8ull << 30)) {
                if ((rc = dev_alloc(h, &h->d_ctg_chars, n_chr)) != DBG_OK) break;
                if ((rc = dev_alloc(h, &lift, (uint64_t)levels * h->n_nodes)) != DBG_OK) break;
                const dim3 ngrid(grid_for(h->n_nodes, 256));
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lift_init<G>), ngrid, dim3(256), 0, h->stream, h->n_nodes, g, lift);
                for (int l = 1; l < levels; ++l)
                    hipLaunchKernelGGL(k_lift_step, ngrid, dim3(256), 0, h->stream, h->n_nodes, lift + (uint64_t)(l - 1) * h->n_nodes,
                                       lift + (uint64_t)l * h->n_nodes);
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_text_fill<G>), dim3(grid_for(n_chr, 256)), dim3(256), 0, h->stream, n_chr, g,
                                   h->d_ctg_off, n_ctg, ctg_start, lift, h->n_nodes, levels, h->d_ctg_chars);
                text_done = true;
            }
        }
        if (!text_done) {
            if ((rc = dev_alloc(h, &h->d_ctg_chars, n_chr)) != DBG_OK) break;
            launch(1);
        }
        if (hipGetLastError() != hipSuccess) { h->err = "walk pass 1 launch failed"; rc = DBG_E_HIP; break; }
        hipError_t e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) { h->err = std::string("walk: ") + hipGetErrorString(e); rc = DBG_E_HIP; break; }
        h->walked = true;
        h->walk_indexed = true;
    } while (0);
    cleanup();
    h->stats.ms_walk = t.stop();
    return rc;
}

extern "C" int dbg_walk(dbg_t *h, int final_mode, uint64_t max_chars) {
    if (!h || !h->pruned) { if (h) h->err = "dbg_prune must run first"; return DBG_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    if (h->D == GEN_D) return walk_impl(h, gen_view(h), final_mode, max_chars);
    return walk_impl(h, dna_view(h), final_mode, max_chars);
}

extern "C" int dbg_get_sizes(dbg_t *h, dbg_sizes_t *o) {
    if (!h || !o) return DBG_E_ARG;
    if (h->k && h->n_nodes && !h->starts_known) {
        HIPCHK(h, hipSetDevice(h->device));
        CHK(ensure_starts(h));
    }
    memset(o, 0, sizeof(*o));
    o->k = h->k;
    o->abi_version = DBG_ABI_VERSION;
    o->n_reads = h->n_reads;
    o->n_bytes = h->n_bytes;
    o->n_kmer_instances = h->n_kmer_inst;
    o->n_edge_instances = h->n_edge_inst;
    o->table_capacity = h->cap;
    o->n_nodes = h->n_nodes;
    o->n_edges = h->n_edges;
    o->n_branch = h->n_branch;
    o->n_pulled = h->n_pulled;
    o->n_pull_reads = h->n_pull_reads;
    o->n_starts = h->n_starts;
    o->n_contigs = h->n_contigs;
    o->contig_chars = h->contig_chars;
    o->tip_rounds = h->tip_rounds;
    o->contigs_materialised = h->walked ? 1 : 0;
    o->max_degree = (uint64_t)h->D;
    return DBG_OK;
}

extern "C" int dbg_get_stats(dbg_t *h, dbg_stats_t *o) {
    if (!h || !o) return DBG_E_ARG;
    *o = h->stats;
    return DBG_OK;
}

#define D2H(h, dst, src, bytes)                                                                       do {                                                                                                  if ((dst) && (bytes)) HIPCHK(h, hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDeviceToHost, (h)->stream));     } while (0)

extern "C" int dbg_export_nodes(dbg_t *h, uint64_t *keys, uint64_t *stamps, uint32_t *counts, uint8_t *flags) {
    if (!h || !h->k) return DBG_E_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    CHK(ensure_dense(h));
    if (keys && !h->d_keys) memset(keys, 0, h->n_nodes * 8);  // k-mers kept by reference (generic alphabet, k >= 12): see stamps
    else D2H(h, keys, h->d_keys, h->n_nodes * 8);
    D2H(h, stamps, h->d_stamps, h->n_nodes * 8);
    D2H(h, counts, h->d_cnt, h->n_nodes * 4 * h->D);
    D2H(h, flags, h->d_flags, h->n_nodes);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return DBG_OK;
}

extern "C" int dbg_export_keys_hi(dbg_t *h, uint64_t *keys_hi) {
    if (!h || !h->k || !keys_hi) return DBG_E_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->d_keys_hi) {  // k <= 31: one word per k-mer
        memset(keys_hi, 0, h->n_nodes * 8);
        return DBG_OK;
    }
    D2H(h, keys_hi, h->d_keys_hi, h->n_nodes * 8);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return DBG_OK;
}

extern "C" int dbg_export_succ(dbg_t *h, uint32_t *succ) {
    if (!h || !h->k) return DBG_E_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    CHK(ensure_dense(h));
    D2H(h, succ, h->d_succ, h->n_nodes * 4 * h->D);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return DBG_OK;
}

extern "C" int dbg_export_csr(dbg_t *h, uint64_t *row_ptr, uint32_t *col, uint32_t *cnt) {
    if (!h || !h->k) return DBG_E_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    CHK(ensure_dense(h));
    D2H(h, row_ptr, h->d_rowptr, (h->n_nodes + 1) * 8);
    D2H(h, col, h->d_col, h->n_edges * 4);
    D2H(h, cnt, h->d_ecnt, h->n_edges * 4);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return DBG_OK;
}

extern "C" int dbg_export_pull_ranks(dbg_t *h, uint64_t *ranks) {
    if (!h || !h->tipped) { if (h) h->err = "dbg_remove_tips must run first"; return DBG_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    D2H(h, ranks, h->d_pull_rank, h->n_nodes * 8);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return DBG_OK;
}

extern "C" int dbg_export_pull_reads(dbg_t *h, uint8_t *read_flags) {
    if (!h || !h->pull_reads_done) { if (h) h->err = "dbg_mark_pull_reads must run first"; return DBG_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    D2H(h, read_flags, h->d_read_flags, h->n_reads);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return DBG_OK;
}

extern "C" int dbg_export_contig_index(dbg_t *h, uint64_t *offsets, uint64_t *scores, uint64_t *start_stamp,
                                       uint32_t *seq_in_start) {
    if (!h || !h->walk_indexed) { if (h) h->err = "dbg_walk must run first"; return DBG_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    D2H(h, offsets, h->d_ctg_off, (h->n_contigs + 1) * 8);
    D2H(h, scores, h->d_ctg_score, h->n_contigs * 8);
    D2H(h, start_stamp, h->d_ctg_stamp, h->n_contigs * 8);
    D2H(h, seq_in_start, h->d_ctg_seq, h->n_contigs * 4);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return DBG_OK;
}

extern "C" int dbg_export_contigs(dbg_t *h, uint64_t *offsets, char *chars, uint64_t *scores, uint64_t *start_stamp,
                                  uint32_t *seq_in_start) {
    if (!h || !h->walked) {
        if (h) h->err = h->walk_indexed ? "contig text was not materialised (larger than max_chars): use dbg_export_contig_index"
                                        : "dbg_walk must run first";
        return DBG_E_ARG;
    }
    HIPCHK(h, hipSetDevice(h->device));
    D2H(h, offsets, h->d_ctg_off, (h->n_contigs + 1) * 8);
    D2H(h, chars, h->d_ctg_chars, h->contig_chars);
    D2H(h, scores, h->d_ctg_score, h->n_contigs * 8);
    D2H(h, start_stamp, h->d_ctg_stamp, h->n_contigs * 8);
    D2H(h, seq_in_start, h->d_ctg_seq, h->n_contigs * 4);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return DBG_OK;
}

template <class G>
__global__ __launch_bounds__(256) void k_text_one(G g, uint32_t start, uint64_t len, const uint32_t *__restrict__ up,
                                                  uint64_t n_nodes, int levels, char *chars) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= len) return;
    uint32_t x = start;
    if (j < (uint64_t)g.k) { chars[j] = g.char_at(x, (int)j); return; }
    uint64_t d = j - (uint64_t)g.k + 1;
    for (int l = 0; l < levels && d; ++l, d >>= 1)
        if (d & 1) x = up[(uint64_t)l * n_nodes + x];
    chars[j] = g.sym_char(g.last_code(x));
}

template <class G>
static int contig_text_impl(dbg *h, const G &g, uint64_t index, char *buf, uint64_t buf_len) {
    uint64_t off[2] = {0, 0};
    uint32_t start = 0;
    HIPCHK(h, hipMemcpyAsync(off, h->d_ctg_off + index, 16, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(&start, h->d_ctg_start + index, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const uint64_t len = off[1] - off[0];
    if (buf_len < len) { h->err = "buffer smaller than the contig"; return DBG_E_ARG; }
    if (!h->d_lift) {  // tables of the 2^j-th chain successors, built once per walk
        int levels = 1;
        while ((1ull << levels) <= h->n_nodes + (uint64_t)h->k) ++levels;
        if ((uint64_t)levels * h->n_nodes * 4 > (64ull << 30)) { h->err = "graph too large for on-demand contig text"; return DBG_E_CAPACITY; }
        CHK(dev_alloc(h, &h->d_lift, (uint64_t)levels * h->n_nodes));
        const dim3 ngrid(grid_for(h->n_nodes, 256));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lift_init<G>), ngrid, dim3(256), 0, h->stream, h->n_nodes, g, h->d_lift);
        for (int l = 1; l < levels; ++l)
            hipLaunchKernelGGL(k_lift_step, ngrid, dim3(256), 0, h->stream, h->n_nodes, h->d_lift + (uint64_t)(l - 1) * h->n_nodes,
                               h->d_lift + (uint64_t)l * h->n_nodes);
        h->lift_levels = levels;
    }
    char *tmp = nullptr;
    CHK(dev_alloc(h, &tmp, len));
    if (len) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_text_one<G>), dim3(grid_for(len, 256)), dim3(256), 0, h->stream, g, start, len,
                                h->d_lift, h->n_nodes, h->lift_levels, tmp);
    hipError_t e = len ? hipMemcpyAsync(buf, tmp, len, hipMemcpyDeviceToHost, h->stream) : hipSuccess;
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    dev_free(tmp);
    if (e != hipSuccess) { h->err = std::string("contig text: ") + hipGetErrorString(e); return DBG_E_HIP; }
    return DBG_OK;
}

extern "C" int dbg_export_contig_text(dbg_t *h, uint64_t index, char *buf, uint64_t buf_len) {
    if (!h || !h->walk_indexed || !buf) { if (h) h->err = "dbg_walk must run first"; return DBG_E_ARG; }
    if (index >= h->n_contigs) { h->err = "contig index out of range"; return DBG_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    if (h->walked) {  // the whole text is on the device already
        uint64_t off[2];
        HIPCHK(h, hipMemcpyAsync(off, h->d_ctg_off + index, 16, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (buf_len < off[1] - off[0]) { h->err = "buffer smaller than the contig"; return DBG_E_ARG; }
        if (off[1] > off[0]) HIPCHK(h, hipMemcpyAsync(buf, h->d_ctg_chars + off[0], off[1] - off[0], hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return DBG_OK;
    }
    if (!h->d_ctg_start) { h->err = "no per-contig start nodes (walk did not take the list-ranking path)"; return DBG_E_ARG; }
    if (h->D == GEN_D) return contig_text_impl(h, gen_view(h), index, buf, buf_len);
    return contig_text_impl(h, dna_view(h), index, buf, buf_len);
}

#ifdef DBG_MS_PROF
extern "C" int dbg_debug_ms_prof(unsigned long long *out8, int reset) {  // experiment builds only (tools/ms_prof.py)
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(dbgk::g_ms_prof), 64) != hipSuccess) return DBG_E_HIP;
    if (reset) { unsigned long long z[8] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(dbgk::g_ms_prof), z, 64) != hipSuccess) return DBG_E_HIP; }
    return DBG_OK;
}
#endif

#ifdef DBG_CNT_PROF
extern "C" int dbg_debug_cnt_prof(unsigned long long *out32, int reset) {  // experiment builds only (tools/cnt_prof.py)
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(dbgk::g_cnt_prof), 512) != hipSuccess) return DBG_E_HIP;
    if (reset) { unsigned long long z[64] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(dbgk::g_cnt_prof), z, 512) != hipSuccess) return DBG_E_HIP; }
    return DBG_OK;
}
#endif

extern "C" int dbg_device_keys_hi(dbg_t *h, const void **d_keys_hi) {
    if (!h || !h->k || !d_keys_hi) return DBG_E_ARG;
    *d_keys_hi = h->d_keys_hi;
    return DBG_OK;
}

extern "C" int dbg_reads_device(dbg_t *h, const void **d_bases, uint64_t *n_bytes, const void **d_offsets, uint64_t *n_reads) {
    if (!h || !h->d_offsets) { if (h) h->err = "no reads set"; return DBG_E_ARG; }
    if (d_bases) *d_bases = h->d_bases;
    if (n_bytes) *n_bytes = h->n_bytes;
    if (d_offsets) *d_offsets = h->d_offsets;
    if (n_reads) *n_reads = h->n_reads;
    return DBG_OK;
}

extern "C" int dbg_device_views(dbg_t *h, const void **d_keys, const void **d_counts, const void **d_stamps,
                                const void **d_flags, const void **d_succ) {
    if (!h || !h->k) return DBG_E_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    CHK(ensure_dense(h));
    if (d_keys) *d_keys = h->d_keys;
    if (d_counts) *d_counts = h->d_cnt;
    if (d_stamps) *d_stamps = h->d_stamps;
    if (d_flags) *d_flags = h->d_flags;
    if (d_succ) *d_succ = h->d_succ;
    return DBG_OK;
}

// ==========================================================================================
// super-k-mer engine: host orchestration
// ==========================================================================================
struct CeilDiv {
    const uint64_t *cnt;
    uint64_t d;
    __device__ uint64_t operator()(uint64_t i) const { return (cnt[i] + d - 1) / d; }
};

// One multisplit level: segments (p_start/p_cnt, device) -> children (c_start/c_cnt, device,
// n_groups * nb entries), records moved from in_* to out_*.
template <class ST, bool HAS_ST, class STI = ST>
static int multisplit_level(dbg *h, const uint64_t *p_start, const uint64_t *p_cnt, uint32_t n_seg, uint32_t spg,
                            uint64_t total, const uint64_t *in_w0, const uint64_t *in_w1, const STI *in_st,
                            uint64_t *out_w0, uint64_t *out_w1, ST *out_st, int shift, int nb, uint64_t *c_start,
                            uint64_t *c_cnt, dbg::Buf &b_scpre, dbg::Buf &b_cmat, dbg::Buf &b_offs, int fbits = 0,
                            const uint64_t *seg_add = nullptr, unsigned long long *d_sums = nullptr,
                            const uint64_t *host_cnt = nullptr) {
    // spg: segments per group (n_seg: all segments form one group; 1: every segment is its own group)
    // host_cnt: the segment counts on the host, when the caller has them: the super-chunk prefix then needs no device scan
    CHK(buf_ensure(h, b_scpre, (uint64_t)(n_seg + 1) * 8));
    uint64_t *sc_pre = (uint64_t *)b_scpre.p;
    uint64_t nsc = 0;
    if (host_cnt) {
        std::vector<uint64_t> &pre = h->host_scpre;  // lives in the handle: the upload is asynchronous
        pre.resize((size_t)n_seg + 1);
        for (uint32_t i = 0; i < n_seg; ++i) { pre[i] = nsc; nsc += (host_cnt[i] + MS_SC - 1) / MS_SC; }
        pre[n_seg] = nsc;
        HIPCHK(h, hipMemcpyAsync(sc_pre, pre.data(), pre.size() * 8, hipMemcpyHostToDevice, h->stream));
    } else {
        CHK(exclusive_scan(h, n_seg, CeilDiv{p_cnt, (uint64_t)MS_SC}, sc_pre, &nsc));
        HIPCHK(h, hipMemcpyAsync(sc_pre + n_seg, &nsc, 8, hipMemcpyHostToDevice, h->stream));
    }
    MsParents P{p_start, p_cnt, sc_pre, n_seg, spg};
    const uint64_t n_log = nsc * (uint64_t)nb;
    CHK(buf_ensure(h, b_cmat, n_log * 4));
    CHK(buf_ensure(h, b_offs, n_log * 8));
    uint32_t *cmat = (uint32_t *)b_cmat.p;
    uint64_t *offs = (uint64_t *)b_offs.p;
    if (nsc) {
        if (d_sums)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ms_hist<true>), dim3((unsigned)nsc), dim3(256), 0, h->stream, P, in_w1, shift, nb,
                               fbits, cmat, d_sums);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ms_hist<false>), dim3((unsigned)nsc), dim3(256), 0, h->stream, P, in_w1, shift, nb,
                               fbits, cmat, (unsigned long long *)nullptr);
        HIPCHK(h, hipGetLastError());
        // (the total of the histograms equals `total` by construction; reading it back here would cost a host
        //  synchronisation per level -- k_ms_children takes the ends from `total` itself)
        CHK(exclusive_scan(h, n_log, MsLogical{P, cmat, nb}, offs, (uint64_t *)nullptr));
    }
    const uint64_t n_child = (uint64_t)(n_seg / spg) * nb;
    hipLaunchKernelGGL(k_ms_children, dim3(grid_for(n_child, 256)), dim3(256), 0, h->stream, P, offs, nb, total, c_start,
                       c_cnt);
    if (nsc) {
        auto kern = k_ms_scatter<ST, HAS_ST, STI>;
        const size_t lds = sizeof(MsLds<ST>);
        HIPCHK(h, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)nsc), dim3(MS_NT), lds, h->stream, P, in_w0, in_w1, in_st, seg_add, shift, nb,
                           fbits, offs, out_w0, out_w1, out_st);
    }
    HIPCHK(h, hipGetLastError());
    return DBG_OK;
}

// ---- stage 1: K1 extraction into one private segment per persistent workgroup (arena set 0)
template <class ST>
static int sk_extract(dbg *h, int k, uint64_t *w0[2], uint64_t *w1[2], ST *st[2], uint64_t **seg_start_out,
                      uint64_t **seg_cnt_out, uint32_t *n_seg_out, uint64_t *n_rec_out, int part = 0, int n_parts = 1) {
    const int m = sk_m_for_k(k), w = k - m + 1;
    unsigned long long *sc_dev = (unsigned long long *)h->d_scalars;
    // part p of n_parts: the k-mers whose first base lies in the tiles [T p / n, T (p + 1) / n) (dbg_shard_extract_part)
    const uint64_t all_tiles = (h->n_bytes + TILE - 1) / TILE;
    const uint64_t tile_first = all_tiles * (uint64_t)part / (uint64_t)n_parts;
    const uint64_t tiles = all_tiles * (uint64_t)(part + 1) / (uint64_t)n_parts - tile_first;
    uint64_t sc[8] = {0};
    const uint32_t n_wg = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(tiles, 1), 2048);
    CHK(buf_ensure(h, h->ar_misc[0], (uint64_t)n_wg * 4 * 8));
    uint64_t *seg_start = (uint64_t *)h->ar_misc[0].p, *seg_cnt = seg_start + n_wg, *seg_nk = seg_cnt + n_wg,
             *seg_ne = seg_nk + n_wg;
    std::vector<uint64_t> hseg((size_t)n_wg * 4);
    uint64_t n_rec = 0;
    Timer t(h->stream);
    const uint64_t tiles_per_wg = (tiles + n_wg - 1) / n_wg;
    // records per position: a window of w k-mers changes its minimizer about every (w + 1) / 2 positions, plus one
    // record per read; sized at 1.3x that, and the second attempt (one record per position) cannot overflow
    const double density = std::min(1.0, 2.6 / (double)(w + 1) + 1.3 * (double)(h->n_reads + 1) / (double)(h->n_bytes + 1) + 0.01);
    uint64_t seg_cap = (w == 1) ? tiles_per_wg * TILE : (uint64_t)((double)(tiles_per_wg * TILE) * density) + 256;
    for (int attempt = 0; attempt < 2; ++attempt) {
        const uint64_t rec_cap = seg_cap * n_wg;
        for (int set = 0; set < 2; ++set) {
            CHK(buf_ensure(h, h->ar_rec[set][0], rec_cap * 8));
            CHK(buf_ensure(h, h->ar_rec[set][1], rec_cap * 8));
            CHK(buf_ensure(h, h->ar_rec[set][2], rec_cap * sizeof(ST)));
            w0[set] = (uint64_t *)h->ar_rec[set][0].p;
            w1[set] = (uint64_t *)h->ar_rec[set][1].p;
            st[set] = (ST *)h->ar_rec[set][2].p;
        }
        for (uint32_t g = 0; g < n_wg; ++g) hseg[g] = (uint64_t)g * seg_cap;
        HIPCHK(h, hipMemcpyAsync(seg_start, hseg.data(), (size_t)n_wg * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemsetAsync(seg_cnt, 0, (size_t)n_wg * 3 * 8, h->stream));
        HIPCHK(h, hipMemsetAsync(h->d_scalars, 0, 64 * 8, h->stream));
        if (tiles) {
            // m = 13 (k >= 13): the register kernel, one instantiation per window w = k - 12 in 1..19 -- the kernel that
            // looks the w hashes of every position up in LDS takes 12.5-15.6 ms at k = 23..30 where these take 3.3-3.7
            bool launched = false;
            if (m == SK_MAX_M && !h->extract_generic) {
#define DBG_EXW_CASE(W_) case W_: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sk_extract_w<ST, W_>), dim3(n_wg), dim3(256), 0, h->stream, \
                                                      h->d_bases, h->n_bytes, h->d_startbits, tiles, w0[0], w1[0], st[0], seg_cap, seg_cnt, \
                                                      seg_nk, seg_ne, sc_dev, tile_first); launched = true; break;
                switch (w) {
                    DBG_EXW_CASE(1) DBG_EXW_CASE(2) DBG_EXW_CASE(3) DBG_EXW_CASE(4) DBG_EXW_CASE(5) DBG_EXW_CASE(6) DBG_EXW_CASE(7)
                    DBG_EXW_CASE(8) DBG_EXW_CASE(9) DBG_EXW_CASE(10) DBG_EXW_CASE(11) DBG_EXW_CASE(12) DBG_EXW_CASE(13)
                    DBG_EXW_CASE(14) DBG_EXW_CASE(15) DBG_EXW_CASE(16) DBG_EXW_CASE(17) DBG_EXW_CASE(18) DBG_EXW_CASE(19)
                    default: break;
                }
#undef DBG_EXW_CASE
            }
            if (!launched)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sk_extract<ST>), dim3(n_wg), dim3(256), 0, h->stream, h->d_bases,
                                   h->n_bytes, h->d_startbits, k, m, tiles, w0[0], w1[0], st[0], seg_cap, seg_cnt, seg_nk,
                                   seg_ne, sc_dev, tile_first);
            HIPCHK(h, hipGetLastError());
        }
        HIPCHK(h, hipMemcpyAsync(sc, h->d_scalars, 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(hseg.data(), seg_start, (size_t)n_wg * 4 * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (sc[0] & 1) { h->err = "reads hold a byte outside ACGT"; return DBG_E_ALPHABET; }
        if (!(sc[0] & 4)) break;
        if (attempt == 1) { h->err = "super-k-mer record buffer overflow"; return DBG_E_CAPACITY; }
        seg_cap = tiles_per_wg * TILE;  // one record per position: cannot overflow
    }
    h->n_kmer_inst = h->n_edge_inst = 0;
    for (uint32_t g = 0; g < n_wg; ++g) {
        n_rec += hseg[n_wg + g];
        h->n_kmer_inst += hseg[2 * n_wg + g];
        h->n_edge_inst += hseg[3 * n_wg + g];
    }
    h->stats.ms_extract = t.stop();
    h->stats.n_records = n_rec;
    h->host_seg_cnt.assign(hseg.begin() + n_wg, hseg.begin() + 2 * (size_t)n_wg);
    *seg_start_out = seg_start;
    *seg_cnt_out = seg_cnt;
    *n_seg_out = n_wg;
    *n_rec_out = n_rec;
    return DBG_OK;
}


// What a sharded build knows about the records it received when every sender split its records by the 512
// level-1 buckets before the exchange (dbg_shard_extract does): the receiver then starts at level 2.
struct Presplit {
    int n_senders = 0;
    const uint64_t *counts = nullptr;     // host, [n_senders][512 / n_shards]: records of (sender, owned level-1 bucket)
    const uint64_t *recv_off = nullptr;   // host, [n_senders]: first record of every sender in the received arrays
    const uint64_t *stamp_add = nullptr;  // host, [n_senders]: 2 x byte offset of the sender's reads in the concatenation
    const void *in_st = nullptr;          // device: stamps of the received records (STI: rank-local 32-bit in a sharded build)
};

// ---- stages 2..: records given as segments of (in_w0, in_w1, in_st) -> node arrays + successors.
// The ping-pong sets w0/w1/st (arena) must hold n_rec records; n_inst bounds the distinct k-mers.
// shard_bits > 0: only buckets whose top shard_bits equal my_shard hold records (the caller made
// sure); successors owned by other shards are left as remote queries in ar_shard[0..1].
template <class ST, int CAP, class STI = ST>
static int sk_count_from_segments(dbg *h, int k, const uint64_t *seg_start, const uint64_t *seg_cnt, uint32_t n_seg,
                                  uint64_t n_rec, uint64_t n_inst, uint64_t n_edge_inst, const uint64_t *in_w0,
                                  const uint64_t *in_w1, const ST *in_st, uint64_t *w0[2], uint64_t *w1[2], ST *st[2],
                                  uint64_t node_capacity_hint, int shard_bits, int my_shard,
                                  const Presplit *pre = nullptr) {
    // pre: n_inst / n_edge_inst come in as upper bounds (the senders did not count per owner) and are replaced by the
    // exact sums of the level-2 histogram pass before anything is sized from them
    const int m = sk_m_for_k(k);
    unsigned long long *sc_dev = (unsigned long long *)h->d_scalars;
    uint64_t sc[8] = {0};
    // ---- bucket geometry.  Level 1 takes up to 9 bits of the bucket hash; the remaining bits are
    //      chosen after level 1 from a distinct-k-mer estimate on one level-1 bucket (auto mode).
    constexpr double TARGET_DISTINCT = CAP * 0.36;  // mean distinct k-mers per final bucket (table ~1/3 full: measured optimum)
    constexpr int T_MAX = 20;                       // provisional geometry: up to 10 + 10 bits; the estimate may add a third level
    const double own = shard_bits ? (double)(1 << shard_bits) : 1.0;  // buckets are spread over `own` shards
    int T = h->bucket_bits;
    const bool auto_T = (T == 0);
    if (auto_T) {  // provisional: assume 40 % of the instances are distinct
        const double want = (double)n_inst * own * 0.4 / TARGET_DISTINCT;
        while (T < T_MAX && (double)(1ull << T) < want) ++T;
    }
    if (T < shard_bits) T = shard_bits;
    if (pre) T = std::max(9, T);  // the senders split by 9 bits
    // level 1 is fixed before the estimate refines the rest; forced geometries take plain bit fields: up to 10 bits at
    // level 2, what is left (the bucket hash has 22 bits) at level 3
    int l1 = T < 9 ? T : (T >= 20 && !pre ? 10 : 9), l2 = std::min(10, T - l1);
    int nb2 = 0;                                            // children of the second level (0: not decided yet)
    int nb3 = (T - l1 - l2) > 0 ? 1 << (T - l1 - l2) : 1;   // children of the third level
    const int nb1 = 1 << l1;
    CHK(buf_ensure(h, h->ar_misc[1], (uint64_t)nb1 * 16));
    uint64_t *c1_start = (uint64_t *)h->ar_misc[1].p, *c1_cnt = c1_start + nb1;
    int where = 0;
    double est_distinct = 0.0;  // distinct k-mers of this shard, from the level-1 sample (0 = unknown)
    const int top = 6 + SK_BUCKET_BITS;
    Timer t_part(h->stream);
    // presplit: the level-2 input segments, bucket-major: segment (bucket b, sender r) = index (b - b_lo) * n_senders + r
    const int bps = pre ? nb1 >> shard_bits : 0;  // level-1 buckets this shard owns
    const uint64_t b_lo = pre ? (uint64_t)my_shard * bps : 0;
    uint64_t *ps_start = nullptr, *ps_cnt = nullptr, *ps_add = nullptr;
    uint32_t ps_n = 0;
    if (pre) {
        ps_n = (uint32_t)(bps * pre->n_senders);
        std::vector<uint64_t> hs((size_t)ps_n * 3);
        for (int r = 0; r < pre->n_senders; ++r) {
            uint64_t at = pre->recv_off[r];
            for (int b = 0; b < bps; ++b) {
                const size_t i = (size_t)b * pre->n_senders + r;
                hs[i] = at;
                hs[ps_n + i] = pre->counts[(size_t)r * bps + b];
                hs[2 * (size_t)ps_n + i] = pre->stamp_add[r];
                at += pre->counts[(size_t)r * bps + b];
            }
        }
        CHK(buf_ensure(h, h->ar_misc[0], (uint64_t)ps_n * 3 * 8));
        ps_start = (uint64_t *)h->ar_misc[0].p; ps_cnt = ps_start + ps_n; ps_add = ps_cnt + ps_n;
        HIPCHK(h, hipMemcpyAsync(ps_start, hs.data(), hs.size() * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));  // hs goes out of scope
    } else {
        const uint64_t *host_cnt = (h->host_seg_cnt.size() == n_seg && seg_cnt == (const uint64_t *)h->ar_misc[0].p + n_seg)
                                       ? h->host_seg_cnt.data() : nullptr;  // the segments sk_extract just wrote
        CHK((multisplit_level<ST, true>(h, seg_start, seg_cnt, n_seg, n_seg, n_rec, in_w0, in_w1, in_st, w0[1], w1[1], st[1],
                                        top - l1, nb1, c1_start, c1_cnt, h->ar_misc[2], h->ar_misc[3], h->ar_misc[4], 0, nullptr,
                                        nullptr, host_cnt)));
        where = 1;
    }
    if (auto_T && l1 >= 9 && n_rec) {  // refine T from a sample: the first level-1 bucket this shard owns
        const uint32_t probe_bucket = shard_bits ? (uint32_t)my_shard << (l1 - shard_bits) : 0u;
        const uint64_t inst_bucket = (uint64_t)((double)n_inst * own / nb1) * 2 + 1024;
        uint64_t set_cap = 1024;
        while (set_cap < inst_bucket * 2) set_cap <<= 1;
        CHK(buf_ensure(h, h->ar_misc[8], set_cap * 8));
        HIPCHK(h, hipMemsetAsync(h->ar_misc[8].p, 0xFF, set_cap * 8, h->stream));
        HIPCHK(h, hipMemsetAsync(h->d_scalars + 40, 0, 16, h->stream));
        if (pre) {  // the probe bucket's records sit in one segment per sender
            for (int r = 0; r < pre->n_senders; ++r)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_estimate_distinct<ST>), dim3(128), dim3(256), 0, h->stream, ps_start, ps_cnt,
                                   (uint32_t)r, in_w0, in_w1, k, (unsigned long long *)h->ar_misc[8].p, set_cap - 1,
                                   (unsigned long long *)(h->d_scalars + 40));
        } else {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_estimate_distinct<ST>), dim3(512), dim3(256), 0, h->stream, c1_start, c1_cnt,
                               probe_bucket, w0[1], w1[1], k, (unsigned long long *)h->ar_misc[8].p, set_cap - 1,
                               (unsigned long long *)(h->d_scalars + 40));
        }
        uint64_t est[2] = {0, 0};
        HIPCHK(h, hipMemcpyAsync(est, h->d_scalars + 40, 16, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (est[0]) {
            double distinct = (double)n_inst * own * (double)est[1] / (double)est[0];
            if (pre) {  // scale the sample by records: the instance total is not known yet
                uint64_t probe_recs = 0;
                for (int r = 0; r < pre->n_senders; ++r) probe_recs += pre->counts[(size_t)r * bps];
                distinct = probe_recs ? (double)est[1] * (double)n_rec / (double)probe_recs * own : 0.0;
            }
            est_distinct = distinct / own;
            // the second level takes any number of children up to 1024 (all hash bits below level 1, scaled): the
            // bucket count follows the estimate instead of jumping by powers of two
            const double want = distinct / ((double)h->target_distinct > 0 ? (double)h->target_distinct : TARGET_DISTINCT);
            const double want2 = std::max<double>(1.0, std::ceil(want / nb1));
            if (want2 <= 1024.0) {
                nb2 = (int)want2;
                nb3 = 1;
            } else {  // more than 1024 children per level-1 group: a plain 10-bit second level and a third one below it
                nb2 = 1024;
                nb3 = (int)std::min<double>((double)(1 << (SK_BUCKET_BITS - l1 - 10)), std::ceil(want2 / 1024.0));
            }
            l2 = nb2 > 1 ? 1 : 0;  // "there is a second level"
        }
    }
    if (nb2 == 0 && l2 > 0) nb2 = 1 << l2;  // forced or small geometries: a power of two, plain bit fields
    if (pre && l2 == 0) { l2 = 1; nb2 = 1; }  // the received records still have to be gathered bucket by bucket
    if (l2 == 0) nb3 = 1;
    // the second level is a scaled field over ALL hash bits below level 1 unless a third level needs the low ones
    const int fb2 = (nb3 == 1 && ((nb2 > 0 && (nb2 & (nb2 - 1)) != 0) || (auto_T && l1 >= 9 && nb2 > 1))) ? SK_BUCKET_BITS - l1 : 0;
    const uint64_t n_l2 = l2 > 0 ? (uint64_t)nb1 * (uint64_t)nb2 : (uint64_t)nb1;  // buckets after level 2
    const uint64_t n_buckets = n_l2 * (uint64_t)nb3;
    T = 0;
    while ((1ull << T) < n_buckets) ++T;  // only for reporting
    const int l2_pow = (fb2 || l2 == 0) ? 0 : (int)std::lround(std::log2((double)nb2));  // second level as plain bits
    const int fb3 = nb3 > 1 ? SK_BUCKET_BITS - l1 - l2_pow : 0;  // level 3: the hash bits below levels 1 and 2, scaled into [0, nb3)
    CHK(buf_ensure(h, h->ar_misc[5], n_buckets * 16));
    uint64_t *b_start = (uint64_t *)h->ar_misc[5].p, *b_cnt = b_start + n_buckets;
    uint64_t *l2_start = b_start, *l2_cnt = b_cnt;  // children of level 2: the final buckets unless a third level follows
    if (nb3 > 1) {
        CHK(buf_ensure(h, h->ar_l2, n_l2 * 16));
        l2_start = (uint64_t *)h->ar_l2.p;
        l2_cnt = l2_start + n_l2;
    }
    if (l2 > 0) {
        const int sh2 = fb2 ? top - SK_BUCKET_BITS : top - l1 - l2_pow;
        if (pre) {
            // only the level-1 buckets this shard owns have records: their children sit at [b_lo * nb2, ...)
            HIPCHK(h, hipMemsetAsync(b_start, 0, n_buckets * 16, h->stream));
            if (nb3 > 1) HIPCHK(h, hipMemsetAsync(l2_start, 0, n_l2 * 16, h->stream));
            HIPCHK(h, hipMemsetAsync(h->d_scalars + 56, 0, 16, h->stream));
            CHK((multisplit_level<ST, true, STI>(h, ps_start, ps_cnt, ps_n, (uint32_t)pre->n_senders, n_rec, in_w0, in_w1,
                                                      (const STI *)pre->in_st, w0[0], w1[0], st[0], sh2, nb2, l2_start + b_lo * nb2,
                                                      l2_cnt + b_lo * nb2, h->ar_misc[2], h->ar_misc[3], h->ar_misc[4], fb2,
                                                      ps_add, (unsigned long long *)(h->d_scalars + 56))));
            uint64_t sums[2] = {0, 0};
            HIPCHK(h, hipMemcpyAsync(sums, h->d_scalars + 56, 16, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            n_inst = sums[0];
            n_edge_inst = sums[1];
            h->n_kmer_inst = n_inst;
            h->n_edge_inst = n_edge_inst;
        } else {
            CHK((multisplit_level<ST, true>(h, c1_start, c1_cnt, (uint32_t)nb1, 1, n_rec, w0[1], w1[1], st[1], w0[0], w1[0],
                                            st[0], sh2, nb2, l2_start, l2_cnt, h->ar_misc[2], h->ar_misc[3], h->ar_misc[4], fb2)));
        }
        where = 0;
        if (nb3 > 1) {  // third level: every level-2 child (of the groups this build owns) is split once more
            if (!w0[1]) {  // callers that start at level 2 bring one record set only
                CHK(buf_ensure(h, h->ar_rec[1][0], (n_rec + 16) * 8));
                CHK(buf_ensure(h, h->ar_rec[1][1], (n_rec + 16) * 8));
                CHK(buf_ensure(h, h->ar_rec[1][2], (n_rec + 16) * sizeof(ST)));
                w0[1] = (uint64_t *)h->ar_rec[1][0].p; w1[1] = (uint64_t *)h->ar_rec[1][1].p; st[1] = (ST *)h->ar_rec[1][2].p;
            }
            const uint64_t p_lo = pre ? b_lo * nb2 : 0, p_n = pre ? (uint64_t)bps * nb2 : n_l2;
            CHK((multisplit_level<ST, true>(h, l2_start + p_lo, l2_cnt + p_lo, (uint32_t)p_n, 1, n_rec, w0[0], w1[0], st[0], w0[1],
                                            w1[1], st[1], top - SK_BUCKET_BITS, nb3, b_start + p_lo * nb3, b_cnt + p_lo * nb3,
                                            h->ar_misc[2], h->ar_misc[3], h->ar_misc[4], fb3)));
            where = 1;
        }
    } else {
        HIPCHK(h, hipMemcpyAsync(b_start, c1_start, (size_t)nb1 * 8, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(b_cnt, c1_cnt, (size_t)nb1 * 8, hipMemcpyDeviceToDevice, h->stream));
    }
    h->stats.ms_partition = t_part.stop();
    h->stats.n_buckets = n_buckets;

    // ---- K5: per-bucket counting.  Node and edge arrays are sized from the distinct-k-mer estimate (58 B per node:
    //      the worst case "every instance distinct" would not fit the HBM beyond ~3e9 instances); if the estimate
    //      was low the kernel reports it and the second attempt takes the worst case.
    // sharded ids carry the owner in bits 31:29; the parts of a multi-pass build keep it in a byte of its own
    const uint64_t id_limit = (shard_bits && !h->wide_owner) ? (h->shard_node_limit ? h->shard_node_limit : (1ull << 29) - 16)
                                                             : 0xFFFFFFF0ull;
    const uint32_t id_tag = (shard_bits && !h->wide_owner) ? ((uint32_t)my_shard << 29) : 0u;
    const uint64_t node_cap_max = std::min<uint64_t>(n_inst, id_limit);
    const uint64_t edge_cap_max = std::min<uint64_t>(n_edge_inst + 16, 0xFFFFFFF0ull);
    uint64_t node_cap = node_capacity_hint ? std::min<uint64_t>(node_capacity_hint, id_limit) : node_cap_max;
    if (!node_capacity_hint && est_distinct > 0.0)
        node_cap = std::min<uint64_t>(node_cap_max, (uint64_t)(est_distinct * 1.2 * h->est_scale_pct / 100.0) +
                                                        (h->est_scale_pct == 100 ? (1u << 20) : 1024u));
    uint64_t edge_cap = std::min<uint64_t>(edge_cap_max, node_cap + node_cap / 4 + 16);
    auto ensure_node_arrays = [&]() -> int {
        CHK(buf_ensure(h, h->ar_node[0], node_cap * 8));
        CHK(buf_ensure(h, h->ar_node[sizeof(ST) == 8 ? 1 : 7], node_cap * sizeof(ST)));
        CHK(buf_ensure(h, h->ar_node[3], node_cap));
        h->d_keys = (uint64_t *)h->ar_node[0].p;
        h->d_stamps_st = h->ar_node[sizeof(ST) == 8 ? 1 : 7].p;
        h->stamps_st_bytes = (int)sizeof(ST);
        h->d_stamps = sizeof(ST) == 8 ? (uint64_t *)h->d_stamps_st : nullptr;
        h->d_flags = (uint8_t *)h->ar_node[3].p;
        h->nodes_in_arena = true;
        CHK(buf_ensure(h, h->ar_csr[3], (node_cap + 1) * 4));
        CHK(buf_ensure(h, h->ar_csr[1], edge_cap * 4));
        CHK(buf_ensure(h, h->ar_csr[2], edge_cap * 4));
        h->d_rowptr32 = (uint32_t *)h->ar_csr[3].p;
        h->d_col = (uint32_t *)h->ar_csr[1].p;
        h->d_ecnt = (uint32_t *)h->ar_csr[2].p;
        return DBG_OK;
    };
    uint64_t q_cap = n_rec + 1024;
    uint64_t *qk[2], *qm[2];
    uint32_t *qc[2];
    const uint64_t range_cap = n_buckets + 4096 + n_inst / (CAP / 4);
    CHK(buf_ensure(h, h->ar_misc[6], range_cap * sizeof(SkRange)));
    SkRange *ranges = (SkRange *)h->ar_misc[6].p;
    // buckets this build owns (a shard or a pass owns 1 / 2^shard_bits of the level-1 groups): only they get a directory
    const uint64_t own_cnt = n_buckets >> shard_bits, own_lo = (uint64_t)my_shard * own_cnt;
    CHK(buf_ensure(h, h->ar_dir, (own_cnt + (range_cap - n_buckets)) * (CAP / 64) * sizeof(SkDirEnt)));
    SkDirEnt *dirs = (SkDirEnt *)h->ar_dir.p;
    // k_sk_count2 for 32-bit stamps; with 64-bit stamps (sharded builds, reads of 2 GiB and more) its LDS leaves room for
    // 320 staged records where the first kernel stages 640, and it loses: 15.4 vs 13.9 ms on a 10 M-read shard
    bool use_count2 = CAP == 4096 && h->count_kernel >= 2 && !h->phase_limit && (sizeof(ST) == 4 || h->count_kernel_u64 >= 2);
    // 64-bit stamps: k_sk_count3 (one hint per slot, 768 staged records) where k_sk_count2's layout leaves room for 320;
    // "count_kernel" 3 runs it for 32-bit stamps too (A/B: tools/sweep.py)
    const bool use_count3 = use_count2 && ((sizeof(ST) == 8 && h->count_kernel_u64 == 3) || h->count_kernel == 3);
    int extra_attempts = 0;
    for (int attempt = 0; attempt < 3 + extra_attempts; ++attempt) {
        CHK(ensure_node_arrays());
        for (int set = 0; set < 2; ++set) {
            CHK(buf_ensure(h, h->ar_q[set][0], q_cap * 8));
            CHK(buf_ensure(h, h->ar_q[set][2], q_cap * 4));
            if (shard_bits || h->resolve_sorted) CHK(buf_ensure(h, h->ar_q[set][1], q_cap * 8));
            qk[set] = (uint64_t *)h->ar_q[set][0].p;
            qm[set] = (uint64_t *)h->ar_q[set][1].p;
            qc[set] = (uint32_t *)h->ar_q[set][2].p;
            if (!shard_bits && !h->resolve_sorted) break;  // the second set is the output of the owner split / the target grouping
        }
        Timer t(h->stream);
        HIPCHK(h, hipMemsetAsync(h->d_scalars, 0, 64 * 8, h->stream));
        HIPCHK(h, hipMemsetAsync(ranges, 0, n_buckets * sizeof(SkRange), h->stream));
        // directory of the owned buckets: entries of a bucket that stays empty or is counted in sub-ranges must not
        // read as "whole bucket" (pad == 1) from an earlier build
        HIPCHK(h, hipMemsetAsync(dirs, 0, own_cnt * (CAP / 64) * sizeof(SkDirEnt), h->stream));
        SkCountOut out{h->d_keys, h->d_stamps_st, h->d_flags, node_cap, h->d_rowptr32, h->d_col, h->d_ecnt, edge_cap,
                       qk[0], qc[0], q_cap, ranges, n_buckets, range_cap, dirs, own_lo, own_cnt, id_tag, sc_dev};
        // split_recs: records beyond which a bucket starts in hash sub-ranges: ~3300 distinct k-mers (80 % of the table: the
        // mean is 36 %, so this is the far tail -- a bucket counted in sub-ranges turns in-bucket successors into queries)
        uint32_t split_recs = 0;
        if (est_distinct > 0.0 && n_rec)
            split_recs = (uint32_t)std::min<double>(1e9, std::max<double>(64.0, (CAP * 0.80) / (est_distinct / (double)n_rec)));
        if (use_count2 && n_rec) {
            auto kern2 = use_count3 ? k_sk_count3<ST> : k_sk_count2<ST>;
            const size_t lds2 = use_count3 ? sizeof(Cnt3Lds<ST>) : sizeof(Cnt2Lds<ST>);
            HIPCHK(h, hipFuncSetAttribute((const void *)kern2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
            int n_cu = 256;
            (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, h->device);
            const unsigned grid = (unsigned)std::min<uint64_t>(n_buckets, (uint64_t)n_cu);  // persistent, one workgroup per CU
            SkCount2Args a2{out, b_start, b_cnt, w0[where], w1[where], (const void *)st[where], n_buckets, split_recs, k};
            SkCount2Args *d_a2 = (SkCount2Args *)(h->d_scalars + 64);
            static_assert(sizeof(SkCount2Args) <= (SK2_QUERY_CURSOR - 64) * 8, "the query cursor sits behind the descriptor");
            HIPCHK(h, hipMemcpyAsync(d_a2, &a2, sizeof(a2), hipMemcpyHostToDevice, h->stream));
            HIPCHK(h, hipMemsetAsync(h->d_scalars + SK2_QUERY_CURSOR, 0, 8, h->stream));
            hipLaunchKernelGGL(kern2, dim3(grid), dim3(Cnt2Cfg<ST>::NT), lds2, h->stream, (const SkCount2Args *)d_a2);
            HIPCHK(h, hipGetLastError());
        }
        auto kern = k_sk_count<ST, CAP>;
        const size_t lds = sizeof(CntLds<ST, CAP>);
        HIPCHK(h, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (n_rec && !use_count2) {
            int n_cu = 256;
            (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, h->device);
            constexpr int NT = CntCfg<ST, CAP>::NT;
            const uint64_t per_cu = std::max<uint64_t>(1, std::min<uint64_t>(2048 / NT, (160 * 1024) / lds));
            const unsigned grid = (unsigned)std::min<uint64_t>(n_buckets, (uint64_t)n_cu * per_cu);  // persistent
            // descriptor in device memory (see fresh_args): words 64.. of the scalar block are reserved for it
            static_assert(sizeof(SkCountOut) <= 64 * 8, "descriptor slot");
            SkCountOut *d_out = (SkCountOut *)(h->d_scalars + 64);
            HIPCHK(h, hipMemcpyAsync(d_out, &out, sizeof(out), hipMemcpyHostToDevice, h->stream));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, h->stream, b_start, b_cnt, w0[where], w1[where],
                               st[where], k, m, n_buckets, (const SkCountOut *)d_out, split_recs, h->phase_limit);
            HIPCHK(h, hipGetLastError());
        }
        h->stats.count_launches = n_rec ? (uint64_t)(attempt + 1) : 0;
        HIPCHK(h, hipMemcpyAsync(sc, h->d_scalars, 64, hipMemcpyDeviceToHost, h->stream));
        uint64_t q2 = 0;
        if (use_count2 && n_rec) HIPCHK(h, hipMemcpyAsync(&q2, h->d_scalars + SK2_QUERY_CURSOR, 8, hipMemcpyDeviceToHost, h->stream));
        h->stats.ms_count = t.stop();
        if (use_count2 && n_rec) sc[5] = q2;  // k_sk_count2 keeps its query cursor on a cache line of its own
        if (h->phase_limit) { h->err = "ablation run (phase_limit set): timing only"; return DBG_E_ARG; }
        // buckets that had to be split by hash sub-range turn in-bucket successors into queries:
        // the usual bound (one query per record) no longer holds, retry with the safe one
        if (sc[0] & (8 | 32)) break;  // not a sizing problem
        bool again = false;
        if (use_count2 && (sc[0] & (512 | 2048))) {
            // (a wrapped 16-bit counter reads 0 twice: k_sk_count3 then reports the overflow AND a mismatch of its two counts)
            if ((sc[0] & 2048) && !(sc[0] & 512)) { h->err = "internal: k_sk_count2 / k_sk_count3 counted a bucket's nodes, edges or queries inconsistently"; return DBG_E_HIP; }
            use_count2 = false;  // an edge seen more than 65 535 times: the kernel with 32-bit counters
            ++extra_attempts;
            again = true;
        }
        if ((sc[0] & 16) && (node_cap < node_cap_max || edge_cap < edge_cap_max) && !node_capacity_hint) {
            node_cap = node_cap_max;  // the estimate was low
            edge_cap = edge_cap_max;
            again = true;
        }
        if ((sc[0] & 64) && q_cap < n_edge_inst + 1024) { q_cap = n_edge_inst + 1024; again = true; }
        if (!again || attempt == 2 + extra_attempts) break;
    }
    if (sc[0] & 8) { h->err = "a bucket could not be split to fit the LDS table"; return DBG_E_CAPACITY; }
    if (sc[0] & 16) { h->err = "node/edge capacity exceeded"; return DBG_E_CAPACITY; }
    if (sc[0] & (32 | 64)) { h->err = "range/query list overflow"; return DBG_E_CAPACITY; }
    h->n_nodes = sc[4] & 0xFFFFFFFFull;
    h->n_edges = sc[4] >> 32;
    {
        const uint32_t ne32 = (uint32_t)h->n_edges;
        HIPCHK(h, hipMemcpyAsync(h->d_rowptr32 + h->n_nodes, &ne32, 4, hipMemcpyHostToDevice, h->stream));
    }
    h->csr_built = true;
    h->dense_pending = true;
    uint64_t n_q = sc[5];
    const uint64_t n_ranges = n_buckets + sc[6];
    h->stats.n_queries = n_q;
    SkGeom geom{k, m, l1, l2 > 0 ? nb2 : 1, fb2, l2_pow, nb3, fb3, shard_bits, my_shard, own_lo, own_cnt};

    // ---- K6: successors that live in another bucket.  Of this shard: the asker looks them up through the target
    //      range's directory (k_succ_resolve).  Of another shard: grouped by owner and parked for the exchange.
    {
        Timer t(h->stream);
        uint64_t root[2] = {0, n_q};
        int qset = 0;
        if (shard_bits && n_q) {  // group by owner shard = top shard_bits of the bucket hash
            hipLaunchKernelGGL(k_q_bucket, dim3(grid_for(n_q, 256)), dim3(256), 0, h->stream, qk[0], qm[0], n_q, k, m);
            HIPCHK(h, hipGetLastError());
            CHK(buf_ensure(h, h->ar_misc[7], 512 * 16 + 16));
            uint64_t *q_seg = (uint64_t *)h->ar_misc[7].p;  // [0..1]: one input segment; [2..]: per-owner children
            const int nsh = 1 << shard_bits;
            HIPCHK(h, hipMemcpyAsync(q_seg, root, 16, hipMemcpyHostToDevice, h->stream));
            uint64_t *o_start = q_seg + 2, *o_cnt = o_start + nsh;
            CHK((multisplit_level<uint32_t, true>(h, q_seg, q_seg + 1, 1, 1, n_q, qk[0], qm[0], qc[0], qk[1], qm[1], qc[1],
                                                  40 + SK_BUCKET_BITS - shard_bits, nsh, o_start, o_cnt, h->ar_misc[2],
                                                  h->ar_misc[3], h->ar_misc[4])));
            ShardState &sh = shard_of(h);
            sh.q_start.assign(nsh, 0);
            sh.q_cnt.assign(nsh, 0);
            HIPCHK(h, hipMemcpyAsync(sh.q_start.data(), o_start, (size_t)nsh * 8, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipMemcpyAsync(sh.q_cnt.data(), o_cnt, (size_t)nsh * 8, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            // park keys and CSR positions (grouped by owner): the exchange reads them, the next build reuses ar_q
            CHK(buf_ensure(h, h->ar_shard[0], n_q * 8));
            CHK(buf_ensure(h, h->ar_shard[3], n_q * 4));
            HIPCHK(h, hipMemcpyAsync(h->ar_shard[0].p, qk[1], n_q * 8, hipMemcpyDeviceToDevice, h->stream));
            HIPCHK(h, hipMemcpyAsync(h->ar_shard[3].p, qc[1], n_q * 4, hipMemcpyDeviceToDevice, h->stream));
            root[0] = sh.q_start[my_shard];
            root[1] = sh.q_cnt[my_shard];
            sh.n_remote = n_q - root[1];
            sh.q_cnt[my_shard] = 0;  // what is left in the lists is remote
            n_q = root[1];
            qset = 1;
        }
        const uint64_t *q_meta = nullptr;
        if (!shard_bits && h->resolve_sorted && (n_q >= (1u << 20) || (h->resolve_sorted == 2 && n_q))) {  // 2: at any size (tests)
            // (experiment, off by default) The queries leave the count kernel in the ASKERS' bucket order and their targets are
            // anywhere: two dependent random lines of HBM each (directory entry, key run).  Grouped by the 512 level-1 groups of
            // the TARGET, the queries in flight at any time look into a few groups' directories and keys (6 MB a group).  The
            // bucket hash is computed once, for the split, and handed to the resolver.  The split (20 B per query read and
            // written, a histogram pass) costs more than the resolver gains: 3.09 vs 2.05 ms at 3.7e7 queries.
            hipLaunchKernelGGL(k_q_bucket, dim3(grid_for(n_q, 256)), dim3(256), 0, h->stream, qk[0], qm[0], n_q, k, m);
            HIPCHK(h, hipGetLastError());
            CHK(buf_ensure(h, h->ar_misc[7], 1024 * 16 + 16));
            uint64_t *q_seg = (uint64_t *)h->ar_misc[7].p;
            HIPCHK(h, hipMemcpyAsync(q_seg, root, 16, hipMemcpyHostToDevice, h->stream));
            uint64_t *o_start = q_seg + 2, *o_cnt = o_start + 512;
            CHK((multisplit_level<uint32_t, true>(h, q_seg, q_seg + 1, 1, 1, n_q, qk[0], qm[0], qc[0], qk[1], qm[1], qc[1],
                                                  40 + SK_BUCKET_BITS - 9, 512, o_start, o_cnt, h->ar_misc[2], h->ar_misc[3],
                                                  h->ar_misc[4])));
            qset = 1;
            q_meta = qm[1];
        }
        if (n_q) {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_succ_resolve<CAP>), dim3(grid_for(n_q, 256)), dim3(256), 0, h->stream,
                               qk[qset] + root[0], qc[qset] + root[0], n_q, geom, ranges, n_buckets, n_ranges, dirs, h->d_keys,
                               h->n_nodes, h->d_col, id_tag, sc_dev, q_meta);
            HIPCHK(h, hipGetLastError());
            HIPCHK(h, hipMemcpyAsync(sc, h->d_scalars, 8, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            if (sc[0] & 128) { h->err = "internal: a successor k-mer was not found in its bucket"; return DBG_E_HIP; }
        }
        h->stats.ms_succ = t.stop();
    }
    // geometry the answer stage of a sharded build needs again
    h->sk_T = T; h->sk_l1 = l1; h->sk_l2 = l2_pow; h->sk_nb2 = fb2 ? nb2 : 0; h->sk_n_ranges = n_ranges; h->sk_cap = CAP;
    h->sk_geom = geom; h->sk_n_buckets = n_buckets;
    if (!shard_bits) {
        h->sk_src.w0 = w0[where]; h->sk_src.w1 = w1[where]; h->sk_src.st = st[where]; h->sk_src.st_bytes = (int)sizeof(ST);
        h->sk_src.b_start = b_start; h->sk_src.b_cnt = b_cnt;
        h->sk_src.valid = n_rec != 0;
    }
    return DBG_OK;
}

// ---- two-word k-mers on the super-k-mer engine (dbg_wsk.h): extraction -> two or three multisplit levels -> k_wsk_gather
//      -> k_wsk_count -> k_wsucc_resolve.  The geometry logic is that of sk_count_from_segments.

__global__ __launch_bounds__(256) void k_wsk_iota128(uint64_t n, uint64_t *w0) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) w0[i] = i * 128;  // received record i keeps its bases in words [4 i, 4 i + 4) of the "packed reads"
}

// part 1: 2-bit packed reads + records (position, meta, stamp) in one private segment per persistent workgroup (set 0)
template <class ST>
static int wsk_extract(dbg *h, int k, uint64_t **pk_out, uint64_t *w0[2], uint64_t *w1[2], ST *st[2], uint64_t **seg_start_out,
                       uint64_t **seg_cnt_out, uint32_t *n_seg_out, uint64_t *n_rec_out, int part = 0, int n_parts = 1) {
    const int m = SK_MAX_M, w = k - m + 1;
    unsigned long long *sc_dev = (unsigned long long *)h->d_scalars;
    uint64_t sc[8] = {0};
    const uint64_t pk_words = (h->n_bytes + 31) / 32;
    CHK(buf_ensure(h, h->ar_wide[0], (pk_words + 8) * 8));
    uint64_t *pk = (uint64_t *)h->ar_wide[0].p;
    HIPCHK(h, hipMemsetAsync(pk + pk_words, 0, 8 * 8, h->stream));
    if (pk_words)
        hipLaunchKernelGGL(k_wpack, dim3(grid_for(pk_words, 256)), dim3(256), 0, h->stream, h->d_bases, h->n_bytes, pk_words, pk);
    // part p of n_parts: the k-mers whose first base lies in the tiles [T p / n, T (p + 1) / n) (dbg_shard_extract_part); the
    // 2-bit packed reads above are whole either way (a record's bases may lie beyond its slice)
    const uint64_t all_tiles = (h->n_bytes + TILE - 1) / TILE;
    const uint64_t tile_first = all_tiles * (uint64_t)part / (uint64_t)n_parts;
    const uint64_t tiles = all_tiles * (uint64_t)(part + 1) / (uint64_t)n_parts - tile_first;
    const bool reg_kernel = (w >= 20 && w <= 51) && !h->extract_generic;  // k = 32..63: the register kernel (256 threads, several workgroups per CU)
    const uint32_t n_wg = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(tiles, 1), reg_kernel ? 2048 : 1024);
    CHK(buf_ensure(h, h->ar_misc[0], (uint64_t)n_wg * 4 * 8));
    uint64_t *seg_start = (uint64_t *)h->ar_misc[0].p, *seg_cnt = seg_start + n_wg, *seg_nk = seg_cnt + n_wg, *seg_ne = seg_nk + n_wg;
    std::vector<uint64_t> hseg((size_t)n_wg * 4);
    uint64_t n_rec = 0;
    Timer t(h->stream);
    const uint64_t tiles_per_wg = (tiles + n_wg - 1) / n_wg;
    const double density = std::min(1.0, 2.6 / (double)(w + 1) + 1.3 * (double)(h->n_reads + 1) / (double)(h->n_bytes + 1) + 0.01);
    uint64_t seg_cap = (uint64_t)((double)(tiles_per_wg * TILE) * density) + 256;
    for (int attempt = 0; attempt < 2; ++attempt) {
        const uint64_t rec_cap = seg_cap * n_wg;
        for (int set = 0; set < 2; ++set) {
            CHK(buf_ensure(h, h->ar_rec[set][0], rec_cap * 8));
            CHK(buf_ensure(h, h->ar_rec[set][1], rec_cap * 8));
            CHK(buf_ensure(h, h->ar_rec[set][2], rec_cap * sizeof(ST)));
            w0[set] = (uint64_t *)h->ar_rec[set][0].p;
            w1[set] = (uint64_t *)h->ar_rec[set][1].p;
            st[set] = (ST *)h->ar_rec[set][2].p;
        }
        for (uint32_t g = 0; g < n_wg; ++g) hseg[g] = (uint64_t)g * seg_cap;
        HIPCHK(h, hipMemcpyAsync(seg_start, hseg.data(), (size_t)n_wg * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemsetAsync(seg_cnt, 0, (size_t)n_wg * 3 * 8, h->stream));
        HIPCHK(h, hipMemsetAsync(h->d_scalars, 0, 64 * 8, h->stream));
        if (tiles && reg_kernel) {  // one instantiation per window w = k - 12 (the doubling-table kernel stays as the generic fallback)
#define DBG_WEXW_CASE(W_) case W_: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wsk_extract_w<ST, W_>), dim3(n_wg), dim3(256), 0, h->stream, \
                                                       h->d_bases, h->n_bytes, h->d_startbits, tiles, w0[0], w1[0], st[0], seg_cap, seg_cnt, \
                                                       seg_nk, seg_ne, sc_dev, tile_first); break;
            switch (w) {
                DBG_WEXW_CASE(20) DBG_WEXW_CASE(21) DBG_WEXW_CASE(22) DBG_WEXW_CASE(23) DBG_WEXW_CASE(24) DBG_WEXW_CASE(25)
                DBG_WEXW_CASE(26) DBG_WEXW_CASE(27) DBG_WEXW_CASE(28) DBG_WEXW_CASE(29) DBG_WEXW_CASE(30) DBG_WEXW_CASE(31)
                DBG_WEXW_CASE(32) DBG_WEXW_CASE(33) DBG_WEXW_CASE(34) DBG_WEXW_CASE(35) DBG_WEXW_CASE(36) DBG_WEXW_CASE(37)
                DBG_WEXW_CASE(38) DBG_WEXW_CASE(39) DBG_WEXW_CASE(40) DBG_WEXW_CASE(41) DBG_WEXW_CASE(42) DBG_WEXW_CASE(43)
                DBG_WEXW_CASE(44) DBG_WEXW_CASE(45) DBG_WEXW_CASE(46) DBG_WEXW_CASE(47) DBG_WEXW_CASE(48) DBG_WEXW_CASE(49)
                DBG_WEXW_CASE(50) DBG_WEXW_CASE(51)
                default: break;
            }
#undef DBG_WEXW_CASE
        } else if (tiles) {
            auto ekern = k_wsk_extract<ST>;
            HIPCHK(h, hipFuncSetAttribute((const void *)ekern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(WSkLds)));
            hipLaunchKernelGGL(ekern, dim3(n_wg), dim3(WSK_NT), sizeof(WSkLds), h->stream, h->d_bases, h->n_bytes,
                               h->d_startbits, k, tiles, w0[0], w1[0], st[0], seg_cap, seg_cnt, seg_nk, seg_ne, sc_dev, tile_first);
        }
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipMemcpyAsync(sc, h->d_scalars, 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(hseg.data(), seg_start, (size_t)n_wg * 4 * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (sc[0] & 1) { h->err = "reads hold a byte outside ACGT"; return DBG_E_ALPHABET; }
        if (!(sc[0] & 4)) break;
        if (attempt == 1) { h->err = "super-k-mer record buffer overflow"; return DBG_E_CAPACITY; }
        seg_cap = tiles_per_wg * TILE;
    }
    h->n_kmer_inst = h->n_edge_inst = 0;
    for (uint32_t g = 0; g < n_wg; ++g) {
        n_rec += hseg[n_wg + g];
        h->n_kmer_inst += hseg[2 * n_wg + g];
        h->n_edge_inst += hseg[3 * n_wg + g];
    }
    h->host_seg_cnt.assign(hseg.begin() + n_wg, hseg.begin() + 2 * (size_t)n_wg);
    h->stats.ms_extract = t.stop();
    h->stats.n_records = n_rec;
    *pk_out = pk; *seg_start_out = seg_start; *seg_cnt_out = seg_cnt; *n_seg_out = n_wg; *n_rec_out = n_rec;
    return DBG_OK;
}

// part 2: records -> node arrays + CSR.  Either the segments wsk_extract wrote (single GPU: level 1 runs here), or -- a
// shard of a multi-GPU build -- records received from every sender, split by the 512 level-1 groups already (`pre`):
// the build starts at level 2, which rebases the 32-bit stamps, and a successor whose bucket another shard owns is
// left unresolved (the gathered graph resolves every successor anyway, dbg_import_graph).
template <class ST, class STI>
static int wsk_count(dbg *h, int k, const uint64_t *pk, const uint64_t *seg_start, const uint64_t *seg_cnt, uint32_t n_seg,
                     uint64_t n_rec, uint64_t n_inst, uint64_t n_edge_inst, const uint64_t *in_w0, const uint64_t *in_w1,
                     const ST *in_st, uint64_t *w0[2], uint64_t *w1[2], ST *st[2], int shard_bits, int my_shard,
                     const Presplit *pre) {
    const int m = SK_MAX_M, w = k - m + 1;
    unsigned long long *sc_dev = (unsigned long long *)h->d_scalars;
    uint64_t sc[8] = {0};
    const double own = shard_bits ? (double)(1 << shard_bits) : 1.0;
    // ---- bucket geometry
    // mean distinct k-mers per final bucket.  Lower than the one-word engine's 0.36 * slots: one minimizer occurrence brings
    // ~320 nodes at k = 63, so bucket sizes vary like a Poisson count of a handful of loci, and a bucket that outgrows the
    // table is counted in hash sub-ranges (measured, 10 M x 150 bp: k = 63 30.2 / 30.5 / 31.0 / 32.9 / 35.5 ms of count
    // kernel at 1100 / 900 / 1000 / 1475 / 1700; k = 47 and k = 32 best at 1000 as well)
    constexpr double TARGET_DISTINCT = 1000.0;
    int T = h->bucket_bits;
    const bool auto_T = (T == 0);
    if (auto_T) {
        const double want = (double)n_inst * own * 0.4 / TARGET_DISTINCT;
        while (T < 20 && (double)(1ull << T) < want) ++T;
    }
    if (T < shard_bits) T = shard_bits;
    if (pre) T = std::max(9, T);
    // 10 bits at level 1 from 2^18 buckets on (single GPU): a third level over half a million tiny segments costs more
    // than it splits; the senders of a sharded build split by 9 bits, there the third level takes what is left
    int l1 = pre ? 9 : (T < 9 ? T : (T >= 18 ? 10 : 9)), l2 = std::min(10, T - l1);
    int nb2 = 0, nb3 = (T - l1 - l2) > 0 ? 1 << (T - l1 - l2) : 1;
    const int nb1 = 1 << l1;
    CHK(buf_ensure(h, h->ar_misc[1], (uint64_t)nb1 * 16));
    uint64_t *c1_start = (uint64_t *)h->ar_misc[1].p, *c1_cnt = c1_start + nb1;
    int where = 1;
    double est_distinct = 0.0;
    const int top = 6 + SK_BUCKET_BITS;
    Timer t_part(h->stream);
    const int bps = pre ? nb1 >> shard_bits : 0;
    const uint64_t b_lo = pre ? (uint64_t)my_shard * bps : 0;
    uint64_t *ps_start = nullptr, *ps_cnt = nullptr, *ps_add = nullptr;
    uint32_t ps_n = 0;
    if (pre) {
        ps_n = (uint32_t)(bps * pre->n_senders);
        std::vector<uint64_t> &hs = h->host_scpre;  // lives in the handle: the upload is asynchronous
        hs.assign((size_t)ps_n * 3, 0);
        for (int r = 0; r < pre->n_senders; ++r) {
            uint64_t at = pre->recv_off[r];
            for (int b = 0; b < bps; ++b) {
                const size_t i = (size_t)b * pre->n_senders + r;
                hs[i] = at;
                hs[ps_n + i] = pre->counts[(size_t)r * bps + b];
                hs[2 * (size_t)ps_n + i] = pre->stamp_add[r];
                at += pre->counts[(size_t)r * bps + b];
            }
        }
        CHK(buf_ensure(h, h->ar_misc[0], (uint64_t)ps_n * 3 * 8));
        ps_start = (uint64_t *)h->ar_misc[0].p; ps_cnt = ps_start + ps_n; ps_add = ps_cnt + ps_n;
        HIPCHK(h, hipMemcpyAsync(ps_start, hs.data(), hs.size() * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    } else {
        CHK((multisplit_level<ST, true>(h, seg_start, seg_cnt, n_seg, n_seg, n_rec, in_w0, in_w1, in_st, w0[1], w1[1], st[1], top - l1,
                                        nb1, c1_start, c1_cnt, h->ar_misc[2], h->ar_misc[3], h->ar_misc[4], 0, nullptr, nullptr,
                                        h->host_seg_cnt.size() == n_seg ? h->host_seg_cnt.data() : nullptr)));
    }
    if (auto_T && l1 >= 9 && n_rec) {
        const uint64_t inst_bucket = (uint64_t)((double)n_inst * own / nb1) * 2 + 1024;
        uint64_t set_cap = 1024;
        while (set_cap < inst_bucket * 2) set_cap <<= 1;
        CHK(buf_ensure(h, h->ar_misc[8], set_cap * 8));
        HIPCHK(h, hipMemsetAsync(h->ar_misc[8].p, 0xFF, set_cap * 8, h->stream));
        HIPCHK(h, hipMemsetAsync(h->d_scalars + 40, 0, 16, h->stream));
        if (pre) {
            for (int r = 0; r < pre->n_senders; ++r)
                hipLaunchKernelGGL(k_wsk_estimate, dim3(128), dim3(256), 0, h->stream, ps_start, ps_cnt, (uint32_t)r, in_w0, in_w1, k, pk,
                                   (unsigned long long *)h->ar_misc[8].p, set_cap - 1, (unsigned long long *)(h->d_scalars + 40));
        } else {
            hipLaunchKernelGGL(k_wsk_estimate, dim3(512), dim3(256), 0, h->stream, c1_start, c1_cnt, 0u, w0[1], w1[1], k, pk,
                               (unsigned long long *)h->ar_misc[8].p, set_cap - 1, (unsigned long long *)(h->d_scalars + 40));
        }
        uint64_t est[2] = {0, 0};
        HIPCHK(h, hipMemcpyAsync(est, h->d_scalars + 40, 16, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (est[0]) {
            double distinct = (double)n_inst * own * (double)est[1] / (double)est[0];
            if (pre) {  // scale the sample by records: the instance total is not known yet
                uint64_t probe_recs = 0;
                for (int r = 0; r < pre->n_senders; ++r) probe_recs += pre->counts[(size_t)r * bps];
                distinct = probe_recs ? (double)est[1] * (double)n_rec / (double)probe_recs * own : 0.0;
            }
            est_distinct = distinct / own;
            const double want = distinct / ((double)h->target_distinct > 0 ? (double)h->target_distinct : TARGET_DISTINCT);
            const double want2 = std::max<double>(1.0, std::ceil(want / nb1));
            if (want2 <= 1024.0) { nb2 = (int)want2; nb3 = 1; }
            else { nb2 = 1024; nb3 = (int)std::min<double>((double)(1 << (SK_BUCKET_BITS - l1 - 10)), std::ceil(want2 / 1024.0)); }
            l2 = nb2 > 1 ? 1 : 0;
        }
    }
    if (nb2 == 0 && l2 > 0) nb2 = 1 << l2;
    if (pre && l2 == 0) { l2 = 1; nb2 = 1; }
    if (l2 == 0) nb3 = 1;
    const int fb2 = (nb3 == 1 && ((nb2 > 0 && (nb2 & (nb2 - 1)) != 0) || (auto_T && l1 >= 9 && nb2 > 1))) ? SK_BUCKET_BITS - l1 : 0;
    const uint64_t n_l2 = l2 > 0 ? (uint64_t)nb1 * (uint64_t)nb2 : (uint64_t)nb1;
    const uint64_t n_buckets = n_l2 * (uint64_t)nb3;
    const int l2_pow = (fb2 || l2 == 0) ? 0 : (int)std::lround(std::log2((double)nb2));
    const int fb3 = nb3 > 1 ? SK_BUCKET_BITS - l1 - l2_pow : 0;
    CHK(buf_ensure(h, h->ar_misc[5], n_buckets * 16));
    uint64_t *b_start = (uint64_t *)h->ar_misc[5].p, *b_cnt = b_start + n_buckets;
    uint64_t *l2_start = b_start, *l2_cnt = b_cnt;
    if (nb3 > 1) {
        CHK(buf_ensure(h, h->ar_l2, n_l2 * 16));
        l2_start = (uint64_t *)h->ar_l2.p;
        l2_cnt = l2_start + n_l2;
    }
    if (l2 > 0) {
        const int sh2 = fb2 ? top - SK_BUCKET_BITS : top - l1 - l2_pow;
        if (pre) {
            HIPCHK(h, hipMemsetAsync(b_start, 0, n_buckets * 16, h->stream));
            if (nb3 > 1) HIPCHK(h, hipMemsetAsync(l2_start, 0, n_l2 * 16, h->stream));
            HIPCHK(h, hipMemsetAsync(h->d_scalars + 56, 0, 16, h->stream));
            CHK((multisplit_level<ST, true, STI>(h, ps_start, ps_cnt, ps_n, (uint32_t)pre->n_senders, n_rec, in_w0, in_w1,
                                                 (const STI *)pre->in_st, w0[0], w1[0], st[0], sh2, nb2, l2_start + b_lo * nb2,
                                                 l2_cnt + b_lo * nb2, h->ar_misc[2], h->ar_misc[3], h->ar_misc[4], fb2, ps_add,
                                                 (unsigned long long *)nullptr)));
        } else {
            CHK((multisplit_level<ST, true>(h, c1_start, c1_cnt, (uint32_t)nb1, 1, n_rec, w0[1], w1[1], st[1], w0[0], w1[0], st[0], sh2,
                                            nb2, l2_start, l2_cnt, h->ar_misc[2], h->ar_misc[3], h->ar_misc[4], fb2)));
        }
        where = 0;
        if (nb3 > 1) {
            if (!w0[1]) {  // callers that start at level 2 may bring one record set only
                CHK(buf_ensure(h, h->ar_rec[1][0], (n_rec + 16) * 8));
                CHK(buf_ensure(h, h->ar_rec[1][1], (n_rec + 16) * 8));
                CHK(buf_ensure(h, h->ar_rec[1][2], (n_rec + 16) * sizeof(ST)));
                w0[1] = (uint64_t *)h->ar_rec[1][0].p; w1[1] = (uint64_t *)h->ar_rec[1][1].p; st[1] = (ST *)h->ar_rec[1][2].p;
            }
            const uint64_t p_lo = pre ? b_lo * nb2 : 0, p_n = pre ? (uint64_t)bps * nb2 : n_l2;
            CHK((multisplit_level<ST, true>(h, l2_start + p_lo, l2_cnt + p_lo, (uint32_t)p_n, 1, n_rec, w0[0], w1[0], st[0], w0[1], w1[1],
                                            st[1], top - SK_BUCKET_BITS, nb3, b_start + p_lo * nb3, b_cnt + p_lo * nb3, h->ar_misc[2],
                                            h->ar_misc[3], h->ar_misc[4], fb3)));
            where = 1;
        }
    } else {
        HIPCHK(h, hipMemcpyAsync(b_start, c1_start, (size_t)nb1 * 8, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(b_cnt, c1_cnt, (size_t)nb1 * 8, hipMemcpyDeviceToDevice, h->stream));
    }
    h->stats.ms_partition = t_part.stop();
    h->stats.n_buckets = n_buckets;
    // the records' bases, aligned, in bucket order (k_wsk_gather)
    CHK(buf_ensure(h, h->ar_wide[2], (n_rec + 1) * 32));
    uint4 *rec_b = (uint4 *)h->ar_wide[2].p;
    {
        Timer t(h->stream);
        if (n_rec)
            hipLaunchKernelGGL(k_wsk_gather, dim3(grid_for(n_rec, 256)), dim3(256), 0, h->stream, w0[where], w1[where], n_rec, pk, k, rec_b);
        HIPCHK(h, hipGetLastError());
        h->stats.ms_compact = t.stop();  // reported in the "compact" slot of dbg_stats_t: this engine has no compaction pass
    }
    if (pre) {  // the instance totals of what was received: a reduction over the meta words (the senders did not count per owner)
        uint64_t tot_len = 0, tot_succ = 0;
        CHK(reduce_sum(h, n_rec, WRecLenAt{w1[where]}, &tot_len));
        CHK(reduce_sum(h, n_rec, WRecSuccAt{w1[where]}, &tot_succ));
        n_inst = tot_len;
        n_edge_inst = tot_len - n_rec + tot_succ;
        h->n_kmer_inst = n_inst;
        h->n_edge_inst = n_edge_inst;
    }
    // ---- per-bucket counting
    // sharded ids carry the owner in bits 31:29; the parts of a multi-pass build keep it in a byte of its own
    const uint64_t id_limit = (shard_bits && !h->wide_owner) ? (h->shard_node_limit ? h->shard_node_limit : (1ull << 29) - 16)
                                                             : 0xFFFFFFF0ull;
    const uint32_t id_tag = (shard_bits && !h->wide_owner) ? ((uint32_t)my_shard << 29) : 0u;
    const uint64_t node_cap_max = std::min<uint64_t>(n_inst, id_limit);
    const uint64_t edge_cap_max = std::min<uint64_t>(n_edge_inst + 16, 0xFFFFFFF0ull);
    uint64_t node_cap = node_cap_max;
    if (est_distinct > 0.0)
        node_cap = std::min<uint64_t>(node_cap_max, (uint64_t)(est_distinct * 1.2 * h->est_scale_pct / 100.0) +
                                                        (h->est_scale_pct == 100 ? (1u << 20) : 1024u));
    uint64_t edge_cap = std::min<uint64_t>(edge_cap_max, node_cap + node_cap / 4 + 16);
    uint64_t q_cap = 2 * n_rec + 1024;  // one per record, more where a bucket is counted in hash sub-ranges
    const uint64_t own_cnt = n_buckets >> shard_bits, own_lo = (uint64_t)my_shard * own_cnt;
    const uint64_t range_cap = n_buckets + 4096 + n_inst / (WCAP / 4);
    CHK(buf_ensure(h, h->ar_misc[6], range_cap * sizeof(SkRange)));
    SkRange *ranges = (SkRange *)h->ar_misc[6].p;
    CHK(buf_ensure(h, h->ar_dir, (own_cnt + (range_cap - n_buckets)) * (WCAP / 64) * sizeof(SkDirEnt)));
    SkDirEnt *dirs = (SkDirEnt *)h->ar_dir.p;
    uint64_t *q_lo = nullptr, *q_hi = nullptr;
    uint32_t *q_col = nullptr;
    for (int attempt = 0; attempt < 3; ++attempt) {
        CHK(buf_ensure(h, h->ar_node[0], node_cap * 8));
        CHK(buf_ensure(h, h->ar_wide[5], node_cap * 8));
        CHK(buf_ensure(h, h->ar_node[sizeof(ST) == 8 ? 1 : 7], node_cap * sizeof(ST)));
        CHK(buf_ensure(h, h->ar_node[3], node_cap));
        CHK(buf_ensure(h, h->ar_csr[3], (node_cap + 1) * 4));
        CHK(buf_ensure(h, h->ar_csr[1], edge_cap * 4));
        CHK(buf_ensure(h, h->ar_csr[2], edge_cap * 4));
        CHK(buf_ensure(h, h->ar_q[0][0], q_cap * 8));
        CHK(buf_ensure(h, h->ar_q[0][1], q_cap * 8));
        CHK(buf_ensure(h, h->ar_q[0][2], q_cap * 4));
        h->d_keys = (uint64_t *)h->ar_node[0].p;
        h->d_keys_hi = (uint64_t *)h->ar_wide[5].p;
        h->d_stamps_st = h->ar_node[sizeof(ST) == 8 ? 1 : 7].p;
        h->stamps_st_bytes = (int)sizeof(ST);
        h->d_stamps = sizeof(ST) == 8 ? (uint64_t *)h->d_stamps_st : nullptr;
        h->d_flags = (uint8_t *)h->ar_node[3].p;
        h->nodes_in_arena = true;
        h->d_rowptr32 = (uint32_t *)h->ar_csr[3].p;
        h->d_col = (uint32_t *)h->ar_csr[1].p;
        h->d_ecnt = (uint32_t *)h->ar_csr[2].p;
        q_lo = (uint64_t *)h->ar_q[0][0].p; q_hi = (uint64_t *)h->ar_q[0][1].p; q_col = (uint32_t *)h->ar_q[0][2].p;
        Timer t(h->stream);
        HIPCHK(h, hipMemsetAsync(h->d_scalars, 0, 64 * 8, h->stream));
        HIPCHK(h, hipMemsetAsync(ranges, 0, n_buckets * sizeof(SkRange), h->stream));
        HIPCHK(h, hipMemsetAsync(dirs, 0, own_cnt * (WCAP / 64) * sizeof(SkDirEnt), h->stream));
        WSkCountOut out{h->d_keys, h->d_keys_hi, h->d_stamps_st, h->d_flags, node_cap, h->d_rowptr32, h->d_col, h->d_ecnt, edge_cap,
                        q_lo, q_hi, q_col, q_cap, ranges, n_buckets, range_cap, dirs, own_lo, own_cnt, sc_dev, id_tag};
        auto kern = k_wsk_count<ST>;
        size_t lds = sizeof(WCntLds<ST>);
        if constexpr (sizeof(ST) == 4) {
            if (h->wcount_kernel == 2) { kern = k_wsk_count2<ST>; lds = sizeof(WCnt2Lds); }
        }
        HIPCHK(h, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (n_rec) {
            int n_cu = 256;
            (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, h->device);
            const unsigned grid = (unsigned)std::min<uint64_t>(n_buckets, (uint64_t)n_cu);
            uint32_t split_recs = 0;
            if (est_distinct > 0.0) split_recs = (uint32_t)std::min<double>(1e9, std::max<double>(16.0, (WCAP * 0.80) / (est_distinct / (double)n_rec)));
            // descriptor in device memory (see fresh_args): words 64.. of the scalar block are reserved for it
            static_assert(sizeof(WSkCountOut) <= 64 * 8, "descriptor slot");
            WSkCountOut *d_out = (WSkCountOut *)(h->d_scalars + 64);
            HIPCHK(h, hipMemcpyAsync(d_out, &out, sizeof(out), hipMemcpyHostToDevice, h->stream));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(WCNT_NT), lds, h->stream, b_start, b_cnt, w0[where], w1[where], st[where],
                               (const uint4 *)rec_b, k, n_buckets, (const WSkCountOut *)d_out, split_recs);
            HIPCHK(h, hipGetLastError());
        }
        h->stats.count_launches = n_rec ? (uint64_t)(attempt + 1) : 0;
        HIPCHK(h, hipMemcpyAsync(sc, h->d_scalars, 64, hipMemcpyDeviceToHost, h->stream));
        h->stats.ms_count = t.stop();
        if (sc[0] & (8 | 32 | 512 | 2048)) break;
        bool again = false;
        if ((sc[0] & 16) && (node_cap < node_cap_max || edge_cap < edge_cap_max)) { node_cap = node_cap_max; edge_cap = edge_cap_max; again = true; }
        if ((sc[0] & 64) && q_cap < n_edge_inst + 1024) { q_cap = n_edge_inst + 1024; again = true; }
        if (!again || attempt == 2) break;
    }
    if (sc[0] & 1024) { h->err = "two-word count kernel: an LDS slot claim never completed"; return DBG_E_HIP; }
    if (sc[0] & 512) { h->err = "16-bit successor counter overflow"; return DBG_E_CAPACITY; }
    if (sc[0] & 2048) { h->err = "internal: k_wsk_count2 counted a bucket's nodes or edges inconsistently"; return DBG_E_HIP; }  // (after 512: a wrapped counter reads 0 twice)
    if (sc[0] & 8) { h->err = "a bucket could not be split to fit the LDS table"; return DBG_E_CAPACITY; }
    if (sc[0] & 16) { h->err = "node/edge capacity exceeded"; return DBG_E_CAPACITY; }
    if (sc[0] & (32 | 64)) { h->err = "range/query list overflow"; return DBG_E_CAPACITY; }
    h->n_nodes = sc[4] & 0xFFFFFFFFull;
    h->n_edges = sc[4] >> 32;
    {
        const uint32_t ne32 = (uint32_t)h->n_edges;
        HIPCHK(h, hipMemcpyAsync(h->d_rowptr32 + h->n_nodes, &ne32, 4, hipMemcpyHostToDevice, h->stream));
    }
    h->csr_built = true;
    h->dense_pending = true;
    uint64_t n_q = sc[5];
    const uint64_t n_ranges = n_buckets + sc[6];
    h->stats.n_queries = n_q;
    SkGeom geom{k, m, l1, l2 > 0 ? nb2 : 1, fb2, l2_pow, nb3, fb3, shard_bits, my_shard, own_lo, own_cnt};
    {
        // successors in another bucket.  Of this shard: through the target range's directory.  Of another shard: grouped
        // by owner and parked as (lo, hi) pairs for the exchange (dbg_shard_answer / dbg_shard_apply), as for k <= 31.
        Timer t(h->stream);
        const uint64_t *r_lo = q_lo, *r_hi = q_hi;
        const uint32_t *r_col = q_col;
        int stride = 1;
        if (shard_bits && n_q) {
            CHK(buf_ensure(h, h->ar_shard[4], n_q * 8));
            CHK(buf_ensure(h, h->ar_q[1][0], n_q * 8));
            CHK(buf_ensure(h, h->ar_q[1][1], n_q * 8));
            CHK(buf_ensure(h, h->ar_q[1][2], n_q * 4));
            uint64_t *q_meta = (uint64_t *)h->ar_shard[4].p;
            uint64_t *s_lo = (uint64_t *)h->ar_q[1][0].p, *s_meta = (uint64_t *)h->ar_q[1][1].p;
            uint32_t *s_col = (uint32_t *)h->ar_q[1][2].p;
            hipLaunchKernelGGL(k_wq_bucket, dim3(grid_for(n_q, 256)), dim3(256), 0, h->stream, q_lo, q_hi, q_meta, n_q, k, m);
            HIPCHK(h, hipGetLastError());
            CHK(buf_ensure(h, h->ar_misc[7], 512 * 16 + 16));
            uint64_t *q_seg = (uint64_t *)h->ar_misc[7].p;
            const int nsh = 1 << shard_bits;
            const uint64_t root[2] = {0, n_q};
            HIPCHK(h, hipMemcpyAsync(q_seg, root, 16, hipMemcpyHostToDevice, h->stream));
            uint64_t *o_start = q_seg + 2, *o_cnt = o_start + nsh;
            CHK((multisplit_level<uint32_t, true>(h, q_seg, q_seg + 1, 1, 1, n_q, q_lo, q_meta, q_col, s_lo, s_meta, s_col,
                                                  40 + SK_BUCKET_BITS - shard_bits, nsh, o_start, o_cnt, h->ar_misc[2],
                                                  h->ar_misc[3], h->ar_misc[4])));
            ShardState &sh = shard_of(h);
            sh.q_start.assign(nsh, 0);
            sh.q_cnt.assign(nsh, 0);
            HIPCHK(h, hipMemcpyAsync(sh.q_start.data(), o_start, (size_t)nsh * 8, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipMemcpyAsync(sh.q_cnt.data(), o_cnt, (size_t)nsh * 8, hipMemcpyDeviceToHost, h->stream));
            CHK(buf_ensure(h, h->ar_shard[0], n_q * 16));
            CHK(buf_ensure(h, h->ar_shard[3], n_q * 4));
            uint64_t *pairs = (uint64_t *)h->ar_shard[0].p;
            hipLaunchKernelGGL(k_wq_park, dim3(grid_for(n_q, 256)), dim3(256), 0, h->stream, n_q, s_lo, s_meta, q_hi, pairs);
            HIPCHK(h, hipGetLastError());
            HIPCHK(h, hipMemcpyAsync(h->ar_shard[3].p, s_col, n_q * 4, hipMemcpyDeviceToDevice, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            const uint64_t mine_at = sh.q_start[my_shard], mine_n = sh.q_cnt[my_shard];
            sh.n_remote = n_q - mine_n;
            sh.q_cnt[my_shard] = 0;  // what is left in the lists is remote
            r_lo = pairs + 2 * mine_at;
            r_hi = r_lo + 1;
            r_col = (const uint32_t *)h->ar_shard[3].p + mine_at;
            stride = 2;
            n_q = mine_n;
        }
        if (n_q) {
            hipLaunchKernelGGL(k_wsucc_resolve, dim3(grid_for(n_q, 256)), dim3(256), 0, h->stream, r_lo, r_hi, stride, r_col, n_q, geom,
                               ranges, n_buckets, n_ranges, dirs, h->d_keys, h->d_keys_hi, h->n_nodes, h->d_col, id_tag, sc_dev);
            HIPCHK(h, hipGetLastError());
            HIPCHK(h, hipMemcpyAsync(sc, h->d_scalars, 8, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            if (sc[0] & 128) { h->err = "internal: a successor k-mer was not found in its bucket"; return DBG_E_HIP; }
        }
        h->stats.ms_succ = t.stop();
    }
    h->sk_geom = geom; h->sk_n_buckets = n_buckets; h->sk_n_ranges = n_ranges; h->sk_cap = WCAP;
    h->sk_src.valid = false;  // the per-range kernels of dbg_refine_edge_order / dbg_mark_pull_reads read one-word records
    (void)w;
    return DBG_OK;
}

template <class ST>
static int build_wsk_t(dbg *h, int k) {
    uint64_t *pk = nullptr, *w0[2], *w1[2], *seg_start = nullptr, *seg_cnt = nullptr, n_rec = 0;
    ST *st[2];
    uint32_t n_seg = 0;
    CHK(wsk_extract<ST>(h, k, &pk, w0, w1, st, &seg_start, &seg_cnt, &n_seg, &n_rec));
    return wsk_count<ST, ST>(h, k, pk, seg_start, seg_cnt, n_seg, n_rec, h->n_kmer_inst, h->n_edge_inst, w0[0], w1[0], st[0], w0, w1, st,
                             0, 0, nullptr);
}

// Reads of 2 GiB and more keep 64-bit stamps (k_wsk_count<uint64_t>: 128 staged records per round beside the wider stamp
// array; round 2 sent them to the global-table engine).  Graphs with an edge seen more than 65 535 times (16-bit LDS
// counters) take the global-table engine.
static int build_wsk(dbg *h, int k, bool *fallback) {
    *fallback = false;
    int rc = (h->n_bytes >= (1ull << 31) || h->stamp64) ? build_wsk_t<uint64_t>(h, k) : build_wsk_t<uint32_t>(h, k);
    if (rc == DBG_E_CAPACITY && h->err == "16-bit successor counter overflow") *fallback = true;
    return rc;
}

template <class ST, int CAP>
static int build_sk_t(dbg *h, int k, uint64_t node_capacity_hint) {
    uint64_t *w0[2], *w1[2], *seg_start = nullptr, *seg_cnt = nullptr, n_rec = 0;
    ST *st[2];
    uint32_t n_seg = 0;
    CHK(sk_extract<ST>(h, k, w0, w1, st, &seg_start, &seg_cnt, &n_seg, &n_rec));
    return sk_count_from_segments<ST, CAP>(h, k, seg_start, seg_cnt, n_seg, n_rec, h->n_kmer_inst, h->n_edge_inst, w0[0],
                                           w1[0], st[0], w0, w1, st, node_capacity_hint, 0, 0);
}

static int build_sk(dbg *h, int k, uint64_t node_capacity_hint) {
    const bool small = (h->n_bytes < (1ull << 31)) && !h->stamp64;
    // 64-bit stamps (inputs of 2 GiB and more, or a sharded build) use the smaller record staging
    if (!small) return build_sk_t<uint64_t, 4096>(h, k, node_capacity_hint);
    if (h->lds_slots == 2048) return build_sk_t<uint32_t, 2048>(h, k, node_capacity_hint);
    return build_sk_t<uint32_t, 4096>(h, k, node_capacity_hint);
}

// ==========================================================================================
// multi-GPU: hash-prefix sharding of the super-k-mer records.
// One handle per rank.  The caller (py-debruijn_amd/multi_gpu.py) moves the buffers between
// ranks with RCCL all-to-all; nothing here talks to another device.
//   dbg_shard_extract : reads of this rank -> records grouped by owner shard (top bits of the
//                       minimizer bucket hash), counts per owner
//   dbg_shard_build   : records received from all ranks (32-bit rank-local stamps are rebased to
//                       global 64-bit ones) -> node table of the buckets this shard owns;
//                       successors owned by other shards come back as queries grouped by owner
//   dbg_shard_answer  : node ids for the successor k-mers other ranks asked about
//   dbg_shard_apply   : answers -> successor arrays; node ids become (owner << 29) | local id
// ==========================================================================================
static ShardState &shard_of(dbg *h) {
    if (!h->shard_state) h->shard_state = new ShardState();
    return *(ShardState *)h->shard_state;
}

// rank-local 32-bit stamps -> global 64-bit ones, and in the same pass the k-mer / edge instance totals of the
// received records (the shard's table is sized from them)
__global__ __launch_bounds__(256) void k_stamp_globalize(const uint32_t *__restrict__ st32, const uint64_t *__restrict__ w1,
                                                         uint64_t n, uint64_t base2, uint64_t *st64,
                                                         unsigned long long *sums /* [0] instances [1] edges */) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t n_inst = 0, n_edge = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        st64[i] = base2 + st32[i];  // ((base + p) << 1) | flag == (base << 1) + ((p << 1) | flag)
        const uint64_t x = w1[i];
        n_inst += ((x >> 1) & 31) + 1;
        n_edge += ((x >> 1) & 31) + (x & 1);
    }
    n_inst = wave_sum_u64(n_inst);
    n_edge = wave_sum_u64(n_edge);
    if ((threadIdx.x & 63) == 0) {
        if (n_inst) atomicAdd(&sums[0], (unsigned long long)n_inst);
        if (n_edge) atomicAdd(&sums[1], (unsigned long long)n_edge);
    }
}

__global__ __launch_bounds__(256) void k_apply_remote(const uint32_t *__restrict__ qcol, const uint32_t *__restrict__ ans,
                                                      uint64_t n, uint32_t tag, uint32_t *col, unsigned long long *scalars) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t a = ans[i];
    if (a == NO_NODE || a >= (1u << 29)) { atomicOr(&scalars[0], 256ull); return; }
    col[qcol[i]] = tag | a;
}

static int shard_args_ok(dbg *h, int k, int n_shards) {
    if (!h) return DBG_E_ARG;
    if (k < 1 || k > 63) { h->err = "k must be in 1..63"; return DBG_E_ARG; }
    if (n_shards < 1 || n_shards > 8 || (n_shards & (n_shards - 1))) {
        h->err = "n_shards must be 1, 2, 4 or 8 (owner = top bits of the bucket hash; node ids keep 29 bits)";
        return DBG_E_ARG;
    }
    if (h->engine != 0) { h->err = "sharded builds use the super-k-mer engine"; return DBG_E_ARG; }
    return DBG_OK;
}

// ---- two-word k-mers (32 <= k <= 63): the instances travel, see dbg_wide.h
struct HasSuccBit {
    const uint64_t *t_st;
    __device__ uint64_t operator()(uint64_t i) const { return (t_st[i] >> 32) & 1ull; }
};

static int shard_extract_wide(dbg *h, int k, int n_shards, uint64_t *send_counts, const void **d_lo, const void **d_hi,
                              const void **d_st) {
    free_build(h);
    h->stats = dbg_stats_t{};
    CHK(compute_alphabet(h));
    if (!h->is_dna) { h->err = "sharded builds take ACGT reads"; return DBG_E_ALPHABET; }
    int shard_bits = 0;
    while ((1 << shard_bits) < n_shards) ++shard_bits;
    Timer t(h->stream);
    unsigned long long *cursor = (unsigned long long *)(h->d_scalars + 48);
    HIPCHK(h, hipMemsetAsync(h->d_scalars, 0, 64 * 8, h->stream));
    const uint64_t tiles = (h->n_bytes + TILE - 1) / TILE;
    if (tiles)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ws_extract<false>), dim3((unsigned)tiles), dim3(256), 0, h->stream, h->d_bases,
                           h->n_bytes, h->d_startbits, k, shard_bits, cursor, (uint64_t *)nullptr, (uint64_t *)nullptr,
                           (uint64_t *)nullptr, (unsigned long long *)h->d_scalars);
    uint64_t sc[64];
    HIPCHK(h, hipMemcpyAsync(sc, h->d_scalars, 64 * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    if (sc[0] & 1) { h->err = "reads hold a byte outside ACGT"; return DBG_E_ALPHABET; }
    h->n_kmer_inst = sc[1];
    h->n_edge_inst = sc[2];
    uint64_t offs[8] = {0}, total = 0;
    for (int d = 0; d < 8; ++d) {
        offs[d] = total;
        if (d < n_shards) send_counts[d] = sc[48 + d];
        total += sc[48 + d];
    }
    for (int a = 0; a < 3; ++a) CHK(buf_ensure(h, h->ar_rec[0][a], (total + 16) * 8));
    uint64_t *lo = (uint64_t *)h->ar_rec[0][0].p, *hi = (uint64_t *)h->ar_rec[0][1].p, *st = (uint64_t *)h->ar_rec[0][2].p;
    HIPCHK(h, hipMemcpyAsync(cursor, offs, sizeof(offs), hipMemcpyHostToDevice, h->stream));
    if (tiles)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ws_extract<true>), dim3((unsigned)tiles), dim3(256), 0, h->stream, h->d_bases,
                           h->n_bytes, h->d_startbits, k, shard_bits, cursor, lo, hi, st, (unsigned long long *)h->d_scalars);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    h->stats.ms_extract = t.stop();
    h->stats.n_records = total;
    *d_lo = lo; *d_hi = hi; *d_st = st;
    ShardState &sh = shard_of(h);
    sh.n_shards = n_shards;
    sh.k = k;
    sh.n_kmer_inst_local = h->n_kmer_inst;
    return DBG_OK;
}

static int shard_build_wide(dbg *h, int k, int n_shards, int my_shard, const uint64_t *t_lo, const uint64_t *t_hi,
                            const uint64_t *t_st, const uint64_t *recv_counts, const uint64_t *stamp_base,
                            uint64_t *q_starts, uint64_t *q_counts, const void **d_q_keys) {
    uint64_t n_rec = 0;
    std::vector<uint64_t> seg(n_shards);
    for (int r = 0; r < n_shards; ++r) { seg[r] = n_rec; n_rec += recv_counts[r]; }
    for (int r = 0; r < n_shards; ++r)
        if (stamp_base[r] >= (1ull << 42)) { h->err = "reads too large for the 44-bit positions of the table's protocol word"; return DBG_E_CAPACITY; }
    h->k = k;
    h->stats.n_records = n_rec;
    Timer t_total(h->stream);
    const uint64_t cap = std::max<uint64_t>(1024, ((uint64_t)((double)(n_rec + 1) / 0.7) + 1023) / 1024 * 1024);
    const uint64_t n_occ = cap / 32;
    CHK(buf_ensure(h, h->ar_wide[1], cap * sizeof(WSlot)));
    CHK(buf_ensure(h, h->ar_wide[2], cap * 16));
    CHK(buf_ensure(h, h->ar_wide[3], cap));
    CHK(buf_ensure(h, h->ar_wide[4], n_occ * 4));
    WSlot *tab = (WSlot *)h->ar_wide[1].p;
    uint32_t *tcnt = (uint32_t *)h->ar_wide[2].p, *word_rank = (uint32_t *)h->ar_wide[4].p;
    uint8_t *occ = (uint8_t *)h->ar_wide[3].p;
    HIPCHK(h, hipMemsetAsync(tab, 0xFF, cap * sizeof(WSlot), h->stream));
    HIPCHK(h, hipMemsetAsync(tcnt, 0, cap * 16, h->stream));
    HIPCHK(h, hipMemsetAsync(occ, 0, cap, h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_scalars, 0, 64 * 8, h->stream));
    Timer tc(h->stream);
    for (int r = 0; r < n_shards; ++r) {
        if (!recv_counts[r]) continue;
        hipLaunchKernelGGL(k_ws_insert, dim3(grid_for(recv_counts[r], 256)), dim3(256), 0, h->stream, t_lo, t_hi, t_st, seg[r],
                           recv_counts[r], stamp_base[r] << 1, tab, tcnt, cap, occ,
                           (unsigned long long *)h->d_scalars);
    }
    uint64_t sc0 = 0;
    HIPCHK(h, hipMemcpyAsync(&sc0, h->d_scalars, 8, hipMemcpyDeviceToHost, h->stream));
    h->stats.ms_count = tc.stop();
    HIPCHK(h, hipGetLastError());
    if (sc0 & 2) { h->err = "hash table capacity exceeded"; return DBG_E_CAPACITY; }
    h->stats.count_launches = (uint64_t)n_shards;
    uint64_t n_edge = 0;
    CHK(reduce_sum(h, n_rec, HasSuccBit{t_st}, &n_edge));
    h->n_kmer_inst = n_rec;
    h->n_edge_inst = n_edge;
    uint64_t total = 0;
    CHK(exclusive_scan(h, n_occ, OccBytes32{occ}, word_rank, &total));
    if (total >= (1ull << 29) - 16) { h->err = "a shard holds at most 2^29 nodes"; return DBG_E_CAPACITY; }
    h->n_nodes = total;
    CHK(buf_ensure(h, h->ar_node[0], total * 8));
    CHK(buf_ensure(h, h->ar_node[1], total * 8));
    CHK(buf_ensure(h, h->ar_node[2], total * 16));
    CHK(buf_ensure(h, h->ar_node[3], total));
    CHK(buf_ensure(h, h->ar_node[4], total));
    CHK(buf_ensure(h, h->ar_node[5], total * 16));
    CHK(buf_ensure(h, h->ar_node[6], total));
    CHK(buf_ensure(h, h->ar_wide[5], total * 8));
    h->d_keys = (uint64_t *)h->ar_node[0].p;
    h->d_stamps = (uint64_t *)h->ar_node[1].p;
    h->d_cnt = (uint32_t *)h->ar_node[2].p;
    h->d_flags = (uint8_t *)h->ar_node[3].p;
    h->d_order = (uint8_t *)h->ar_node[4].p;
    h->d_succ = (uint32_t *)h->ar_node[5].p;
    h->d_deg = (uint8_t *)h->ar_node[6].p;
    h->d_keys_hi = (uint64_t *)h->ar_wide[5].p;
    h->nodes_in_arena = true;
    hipLaunchKernelGGL(k_wgather, dim3(grid_for(n_occ * 32, 256)), dim3(256), 0, h->stream, tab, tcnt, occ, word_rank, n_occ,
                       h->d_keys, h->d_keys_hi, h->d_stamps, h->d_cnt, h->d_flags, 1);
    // successors inside this shard only (ids local, untagged): the gathered graph resolves all of them (dbg_import_graph)
    if (total)
        hipLaunchKernelGGL(k_wsucc, dim3(grid_for(total, 256)), dim3(256), 0, h->stream, tab, cap, k, total,
                           h->d_keys, h->d_keys_hi, h->d_cnt, h->d_succ, h->d_order, h->d_deg);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    ShardState &sh = shard_of(h);
    int shard_bits = 0;
    while ((1 << shard_bits) < n_shards) ++shard_bits;
    sh.n_shards = n_shards; sh.my_shard = my_shard; sh.shard_bits = shard_bits; sh.k = k; sh.n_remote = 0;
    sh.q_start.assign(n_shards, 0);
    sh.q_cnt.assign(n_shards, 0);
    for (int d = 0; d < n_shards; ++d) { q_starts[d] = 0; q_counts[d] = 0; }
    *d_q_keys = nullptr;
    h->partial_graph = n_shards > 1;
    h->stats.ms_build_total = t_total.stop();
    return DBG_OK;
}

// ---- two-word k-mers, sharded, on the LDS engine: the unit on the wire is the super-k-mer record BY VALUE -- its bases
//      (k_wsk_gather: four aligned words), meta word and rank-local stamp, 44 bytes for ~26 k-mer instances at k = 63
//      (the instance tuples of shard_extract_wide are 24 bytes each).  The receiver takes the received bases as its
//      packed reads: record i sits at base position 128 i.
static int shard_extract_wsk(dbg *h, int k, int n_shards, uint64_t *send_counts, const void **d_rb, const void **d_w1,
                             const void **d_st, int part = 0, int n_parts = 1) {
    free_build(h);
    h->stats = dbg_stats_t{};
    CHK(compute_alphabet(h));
    if (!h->is_dna) { h->err = "sharded builds take ACGT reads"; return DBG_E_ALPHABET; }
    uint64_t *pk = nullptr, *w0[2], *w1[2], *seg_start = nullptr, *seg_cnt = nullptr, n_rec = 0;
    uint32_t *st[2];
    uint32_t n_seg = 0;
    CHK(wsk_extract<uint32_t>(h, k, &pk, w0, w1, st, &seg_start, &seg_cnt, &n_seg, &n_rec, part, n_parts));
    if (n_parts > 1) {  // what travels (meta words, stamps; the gathered bases below) keeps buffers of its own per part
        CHK(buf_ensure(h, h->ar_part[part][1], (n_rec + 16) * 8));
        CHK(buf_ensure(h, h->ar_part[part][2], (n_rec + 16) * 4));
        w1[1] = (uint64_t *)h->ar_part[part][1].p;
        st[1] = (uint32_t *)h->ar_part[part][2].p;
    }
    const int nb1 = 512;
    CHK(buf_ensure(h, h->ar_misc[1], (uint64_t)nb1 * 16));
    uint64_t *c1_start = (uint64_t *)h->ar_misc[1].p, *c1_cnt = c1_start + nb1;
    Timer t(h->stream);
    CHK((multisplit_level<uint32_t, true>(h, seg_start, seg_cnt, n_seg, n_seg, n_rec, w0[0], w1[0], st[0], w0[1], w1[1], st[1],
                                          6 + SK_BUCKET_BITS - 9, nb1, c1_start, c1_cnt, h->ar_misc[2], h->ar_misc[3],
                                          h->ar_misc[4], 0, nullptr, nullptr, h->host_seg_cnt.data())));
    std::vector<uint64_t> cnt(nb1);
    HIPCHK(h, hipMemcpyAsync(cnt.data(), c1_cnt, nb1 * 8, hipMemcpyDeviceToHost, h->stream));
    dbg::Buf &b_rb = n_parts > 1 ? h->ar_part[part][0] : h->ar_wide[2];
    CHK(buf_ensure(h, b_rb, (n_rec + 1) * 32));
    uint4 *rec_b = (uint4 *)b_rb.p;
    if (n_rec)
        hipLaunchKernelGGL(k_wsk_gather, dim3(grid_for(n_rec, 256)), dim3(256), 0, h->stream, w0[1], w1[1], n_rec, pk, k, rec_b);
    HIPCHK(h, hipGetLastError());
    h->stats.ms_partition = t.stop();
    for (int d = 0; d < n_shards; ++d) {
        send_counts[d] = 0;
        for (int b = d * nb1 / n_shards; b < (d + 1) * nb1 / n_shards; ++b) send_counts[d] += cnt[b];
    }
    *d_rb = rec_b;
    *d_w1 = w1[1];
    *d_st = st[1];
    ShardState &sh = shard_of(h);
    sh.n_shards = n_shards;
    sh.k = k;
    sh.n_kmer_inst_local = (part ? sh.n_kmer_inst_local : 0) + h->n_kmer_inst;
    sh.l1_counts = cnt;
    sh.rec_words = 4;
    sh.rec_stamp_bytes = 4;
    return DBG_OK;
}

static int shard_build_wsk(dbg *h, int k, int n_shards, int my_shard, const uint64_t *d_rb, const uint64_t *d_w1,
                           const uint32_t *d_st32, const uint64_t *recv_counts, const uint64_t *stamp_base,
                           const uint64_t *sender_bucket_counts, uint64_t *q_starts, uint64_t *q_counts, const void **d_q_keys) {
    int shard_bits = 0;
    while ((1 << shard_bits) < n_shards) ++shard_bits;
    uint64_t n_rec = 0;
    std::vector<uint64_t> seg(n_shards), add(n_shards);
    const int bps = 512 >> shard_bits;
    for (int r = 0; r < n_shards; ++r) {
        seg[r] = n_rec;
        n_rec += recv_counts[r];
        add[r] = stamp_base[r] << 1;
        uint64_t tot = 0;
        for (int b = 0; b < bps; ++b) tot += sender_bucket_counts[(size_t)r * bps + b];
        if (tot != recv_counts[r]) { h->err = "sender_bucket_counts do not add up to recv_counts"; return DBG_E_ARG; }
    }
    free_build(h);
    uint64_t *w0[2], *w1[2], *st[2];
    for (int set = 0; set < 2; ++set) {
        CHK(buf_ensure(h, h->ar_rec[set][0], (n_rec + 16) * 8));
        CHK(buf_ensure(h, h->ar_rec[set][1], (n_rec + 16) * 8));
        CHK(buf_ensure(h, h->ar_rec[set][2], (n_rec + 16) * 8));
        w0[set] = (uint64_t *)h->ar_rec[set][0].p;
        w1[set] = (uint64_t *)h->ar_rec[set][1].p;
        st[set] = (uint64_t *)h->ar_rec[set][2].p;
    }
    CHK(buf_ensure(h, h->ar_shard[2], (n_rec + 16) * 8));
    uint64_t *in_w0 = (uint64_t *)h->ar_shard[2].p;
    if (n_rec) hipLaunchKernelGGL(k_wsk_iota128, dim3(grid_for(n_rec, 256)), dim3(256), 0, h->stream, n_rec, in_w0);
    h->k = k;
    h->stats.n_records = n_rec;
    ShardState &sh = shard_of(h);
    sh.n_shards = n_shards; sh.my_shard = my_shard; sh.shard_bits = shard_bits; sh.k = k; sh.n_remote = 0;
    sh.q_start.assign(n_shards, 0);
    sh.q_cnt.assign(n_shards, 0);
    Presplit pre;
    pre.n_senders = n_shards;
    pre.counts = sender_bucket_counts;
    pre.recv_off = seg.data();
    pre.stamp_add = add.data();
    pre.in_st = d_st32;
    const int w = k - SK_MAX_M + 1;
    Timer t_total(h->stream);
    int rc = wsk_count<uint64_t, uint32_t>(h, k, d_rb, nullptr, nullptr, 0, n_rec, n_rec * (uint64_t)w, n_rec * (uint64_t)w, in_w0, d_w1,
                                           (const uint64_t *)nullptr, w0, w1, st, shard_bits, my_shard, &pre);
    if (rc != DBG_OK) { const std::string keep = h->err; free_build(h); h->err = keep; return rc; }
    // successors owned by other shards: (lo, hi) pairs grouped by owner (two words per query)
    for (int d = 0; d < n_shards; ++d) { q_starts[d] = sh.q_start[d]; q_counts[d] = sh.q_cnt[d]; }
    *d_q_keys = h->ar_shard[0].p;
    h->partial_graph = n_shards > 1;
    h->stats.ms_build_total = t_total.stop();
    return DBG_OK;
}

extern "C" int dbg_shard_record_layout(dbg_t *h, int *w0_words, int *stamp_bytes) {
    if (!h || !h->shard_state) { if (h) h->err = "dbg_shard_extract must run first"; return DBG_E_ARG; }
    ShardState &sh = shard_of(h);
    if (w0_words) *w0_words = sh.rec_words;
    if (stamp_bytes) *stamp_bytes = sh.rec_stamp_bytes;
    return DBG_OK;
}

// one-word k-mers: records split by the 512 level-1 groups; ST = width of the rank-local stamps
template <class ST>
static int shard_extract_sk(dbg *h, int k, int n_shards, uint64_t *send_counts, const void **d_w0, const void **d_w1,
                            const void **d_st, int part = 0, int n_parts = 1) {
    free_build(h);
    h->stats = dbg_stats_t{};
    uint64_t *w0[2], *w1[2], *seg_start = nullptr, *seg_cnt = nullptr, n_rec = 0;
    ST *st[2];
    uint32_t n_seg = 0;
    CHK(sk_extract<ST>(h, k, w0, w1, st, &seg_start, &seg_cnt, &n_seg, &n_rec, part, n_parts));
    if (n_parts > 1) {  // the grouped records of every part keep buffers of their own: part p travels while part p + 1 is cut
        CHK(buf_ensure(h, h->ar_part[part][0], (n_rec + 16) * 8));
        CHK(buf_ensure(h, h->ar_part[part][1], (n_rec + 16) * 8));
        CHK(buf_ensure(h, h->ar_part[part][2], (n_rec + 16) * sizeof(ST)));
        w0[1] = (uint64_t *)h->ar_part[part][0].p;
        w1[1] = (uint64_t *)h->ar_part[part][1].p;
        st[1] = (ST *)h->ar_part[part][2].p;
    }
    // group by the 9 top bits of the bucket hash: owners are contiguous ranges of those 512 groups
    const int nb1 = 512;
    CHK(buf_ensure(h, h->ar_misc[1], (uint64_t)nb1 * 16));
    uint64_t *c1_start = (uint64_t *)h->ar_misc[1].p, *c1_cnt = c1_start + nb1;
    Timer t(h->stream);
    CHK((multisplit_level<ST, true>(h, seg_start, seg_cnt, n_seg, n_seg, n_rec, w0[0], w1[0], st[0], w0[1], w1[1],
                                    st[1], 6 + SK_BUCKET_BITS - 9, nb1, c1_start, c1_cnt, h->ar_misc[2],
                                    h->ar_misc[3], h->ar_misc[4])));
    std::vector<uint64_t> cnt(nb1);
    HIPCHK(h, hipMemcpyAsync(cnt.data(), c1_cnt, nb1 * 8, hipMemcpyDeviceToHost, h->stream));
    h->stats.ms_partition = t.stop();
    for (int d = 0; d < n_shards; ++d) {
        send_counts[d] = 0;
        for (int b = d * nb1 / n_shards; b < (d + 1) * nb1 / n_shards; ++b) send_counts[d] += cnt[b];
    }
    *d_w0 = w0[1];
    *d_w1 = w1[1];
    *d_st = st[1];
    ShardState &sh = shard_of(h);
    sh.n_shards = n_shards;
    sh.k = k;
    sh.n_kmer_inst_local = (part ? sh.n_kmer_inst_local : 0) + h->n_kmer_inst;
    sh.l1_counts = cnt;
    sh.rec_words = 1;
    sh.rec_stamp_bytes = (int)sizeof(ST);
    return DBG_OK;
}

// dbg_shard_extract on one of n_parts slices of this rank's reads (k <= 31): slices of the position space cut at tile
// borders -- a k-mer belongs to the slice its first base lies in, so the parts' records together are exactly the records of
// dbg_shard_extract.  The arrays of part p are its own until the next extraction of part p.
extern "C" int dbg_shard_extract_part(dbg_t *h, int k, int n_shards, int part, int n_parts, uint64_t *send_counts,
                                      const void **d_w0, const void **d_w1, const void **d_st) {
    CHK(shard_args_ok(h, k, n_shards));
    if (!send_counts || !d_w0 || !d_w1 || !d_st || !h->d_offsets) { h->err = "bad argument / no reads"; return DBG_E_ARG; }
    if (n_parts < 1 || n_parts > 4 || part < 0 || part >= n_parts) { h->err = "dbg_shard_extract_part: 1 <= n_parts <= 4, 0 <= part < n_parts"; return DBG_E_ARG; }
    const bool st64 = h->n_bytes >= (1ull << 31) || h->shard_stamp64;
    HIPCHK(h, hipSetDevice(h->device));
    if (k > 31) {  // two-word k-mers: records by value (the LDS engine), 32-bit rank-local stamps
        if (h->wide_engine != 1) { h->err = "dbg_shard_extract_part: two-word k-mers on the LDS engine (\"wide_engine\" 1)"; return DBG_E_ARG; }
        if (h->n_bytes >= (1ull << 31)) { h->err = "a shard's reads must stay below 2 GiB for k > 31 (32-bit local stamps)"; return DBG_E_ARG; }
        return shard_extract_wsk(h, k, n_shards, send_counts, d_w0, d_w1, d_st, part, n_parts);
    }
    return st64 ? shard_extract_sk<uint64_t>(h, k, n_shards, send_counts, d_w0, d_w1, d_st, part, n_parts)
                : shard_extract_sk<uint32_t>(h, k, n_shards, send_counts, d_w0, d_w1, d_st, part, n_parts);
}

extern "C" int dbg_shard_extract(dbg_t *h, int k, int n_shards, uint64_t *send_counts, const void **d_w0,
                                 const void **d_w1, const void **d_st) {
    CHK(shard_args_ok(h, k, n_shards));
    if (!send_counts || !d_w0 || !d_w1 || !d_st || !h->d_offsets) { h->err = "bad argument / no reads"; return DBG_E_ARG; }
    // rank-local stamps: 32 bits while this rank's reads stay below 2 GiB (or "shard_stamp64" asks for the wide ones)
    const bool st64 = h->n_bytes >= (1ull << 31) || h->shard_stamp64;
    HIPCHK(h, hipSetDevice(h->device));
    if (k > 31) {
        if (h->n_bytes >= (1ull << 31)) { h->err = "a shard's reads must stay below 2 GiB for k > 31 (32-bit local stamps)"; return DBG_E_ARG; }
        if (h->wide_engine == 1) return shard_extract_wsk(h, k, n_shards, send_counts, d_w0, d_w1, d_st);
        int rcw = shard_extract_wide(h, k, n_shards, send_counts, d_w0, d_w1, d_st);
        if (rcw == DBG_OK) { shard_of(h).rec_words = 1; shard_of(h).rec_stamp_bytes = 8; }
        return rcw;
    }
    return st64 ? shard_extract_sk<uint64_t>(h, k, n_shards, send_counts, d_w0, d_w1, d_st)
                : shard_extract_sk<uint32_t>(h, k, n_shards, send_counts, d_w0, d_w1, d_st);
}

extern "C" int dbg_shard_bucket_counts(dbg_t *h, uint64_t *counts512) {
    if (!h || !counts512 || !h->shard_state) return DBG_E_ARG;
    ShardState &sh = shard_of(h);
    if (sh.l1_counts.size() != 512) { h->err = "dbg_shard_extract (k <= 31) must run first"; return DBG_E_ARG; }
    memcpy(counts512, sh.l1_counts.data(), 512 * 8);
    return DBG_OK;
}

extern "C" int dbg_shard_build(dbg_t *h, int k, int n_shards, int my_shard, const void *d_w0, const void *d_w1,
                               const void *d_st32, const uint64_t *recv_counts, const uint64_t *stamp_base,
                               uint64_t *q_starts, uint64_t *q_counts, const void **d_q_keys,
                               const uint64_t *sender_bucket_counts, int stamp_bytes) {
    CHK(shard_args_ok(h, k, n_shards));
    if (my_shard < 0 || my_shard >= n_shards || !recv_counts || !stamp_base || !q_starts || !q_counts || !d_q_keys)
        return DBG_E_ARG;
    if (stamp_bytes != 0 && stamp_bytes != 4 && stamp_bytes != 8) { h->err = "stamp_bytes must be 0 (the layout's default), 4 or 8"; return DBG_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    if (k > 31) {
        const bool lds_engine = h->wide_engine == 1 && sender_bucket_counts;
        if (stamp_bytes && stamp_bytes != (lds_engine ? 4 : 8)) { h->err = "two-word sharded builds: stamps are 4 bytes (records) or 8 (k-mer instances)"; return DBG_E_ARG; }
        if (h->wide_engine == 1 && sender_bucket_counts)
            return shard_build_wsk(h, k, n_shards, my_shard, (const uint64_t *)d_w0, (const uint64_t *)d_w1, (const uint32_t *)d_st32,
                                   recv_counts, stamp_base, sender_bucket_counts, q_starts, q_counts, d_q_keys);
        return shard_build_wide(h, k, n_shards, my_shard, (const uint64_t *)d_w0, (const uint64_t *)d_w1, (const uint64_t *)d_st32,
                                recv_counts, stamp_base, q_starts, q_counts, d_q_keys);
    }
    int shard_bits = 0;
    while ((1 << shard_bits) < n_shards) ++shard_bits;
    uint64_t n_rec = 0;
    std::vector<uint64_t> seg((size_t)n_shards * 2);
    for (int r = 0; r < n_shards; ++r) { seg[r] = n_rec; seg[n_shards + r] = recv_counts[r]; n_rec += recv_counts[r]; }
    free_build(h);  // node arrays of an earlier build on this handle (the record arenas stay)
    // arena sets for 64-bit stamps
    uint64_t *w0[2], *w1[2], *st[2];
    for (int set = 0; set < 2; ++set) {
        CHK(buf_ensure(h, h->ar_rec[set][0], (n_rec + 16) * 8));
        CHK(buf_ensure(h, h->ar_rec[set][1], (n_rec + 16) * 8));
        CHK(buf_ensure(h, h->ar_rec[set][2], (n_rec + 16) * 8));
        w0[set] = (uint64_t *)h->ar_rec[set][0].p;
        w1[set] = (uint64_t *)h->ar_rec[set][1].p;
        st[set] = (uint64_t *)h->ar_rec[set][2].p;
    }
    h->k = k;
    h->stats.n_records = n_rec;
    ShardState &sh = shard_of(h);
    sh.n_shards = n_shards; sh.my_shard = my_shard; sh.shard_bits = shard_bits; sh.k = k; sh.n_remote = 0;
    sh.q_start.assign(n_shards, 0);
    sh.q_cnt.assign(n_shards, 0);
    int rc;
    uint64_t n_inst = 0, n_edge = 0;
    const bool presplit = sender_bucket_counts && (h->bucket_bits == 0 || h->bucket_bits >= 9);
    Timer t_total(h->stream);
    if (presplit) {
        // every sender split its records by the 512 level-1 buckets (dbg_shard_extract) and says how many of each it
        // sent: the receiver starts at level 2, which also rebases the rank-local stamps and counts the instances
        const int bps = 512 >> shard_bits;
        for (int r = 0; r < n_shards; ++r) {
            uint64_t tot = 0;
            for (int b = 0; b < bps; ++b) tot += sender_bucket_counts[(size_t)r * bps + b];
            if (tot != recv_counts[r]) { h->err = "sender_bucket_counts do not add up to recv_counts"; return DBG_E_ARG; }
        }
        std::vector<uint64_t> add(n_shards);
        for (int r = 0; r < n_shards; ++r) add[r] = stamp_base[r] << 1;
        Presplit pre;
        pre.n_senders = n_shards;
        pre.counts = sender_bucket_counts;
        pre.recv_off = seg.data();
        pre.stamp_add = add.data();
        pre.in_st = d_st32;
        const int w = k - sk_m_for_k(k) + 1;  // a record holds at most w k-mers
        if (stamp_bytes == 8)  // senders that hold 2 GiB of reads or more
            rc = sk_count_from_segments<uint64_t, 4096, uint64_t>(h, k, nullptr, nullptr, 0, n_rec, n_rec * (uint64_t)w,
                                                                  n_rec * (uint64_t)w, (const uint64_t *)d_w0, (const uint64_t *)d_w1,
                                                                  (const uint64_t *)nullptr, w0, w1, st, 0, shard_bits, my_shard, &pre);
        else
            rc = sk_count_from_segments<uint64_t, 4096, uint32_t>(h, k, nullptr, nullptr, 0, n_rec, n_rec * (uint64_t)w,
                                                                  n_rec * (uint64_t)w, (const uint64_t *)d_w0, (const uint64_t *)d_w1,
                                                                  (const uint64_t *)nullptr, w0, w1, st, 0, shard_bits, my_shard, &pre);
        n_inst = h->n_kmer_inst;
        n_edge = h->n_edge_inst;
    } else {
        if (stamp_bytes == 8) { h->err = "64-bit rank-local stamps need sender_bucket_counts (the receiver starts at level 2)"; return DBG_E_ARG; }
        CHK(buf_ensure(h, h->ar_misc[0], (uint64_t)n_shards * 16));
        uint64_t *seg_start = (uint64_t *)h->ar_misc[0].p, *seg_cnt = seg_start + n_shards;
        HIPCHK(h, hipMemcpyAsync(seg_start, seg.data(), seg.size() * 8, hipMemcpyHostToDevice, h->stream));
        // the rebased stamps of the received records go to ar_shard[2]
        CHK(buf_ensure(h, h->ar_shard[2], (n_rec + 16) * 8));
        uint64_t *st64 = (uint64_t *)h->ar_shard[2].p;
        HIPCHK(h, hipMemsetAsync(h->d_scalars + 56, 0, 16, h->stream));
        for (int r = 0; r < n_shards; ++r) {
            if (!recv_counts[r]) continue;
            const unsigned grid = (unsigned)std::min<uint64_t>(grid_for(recv_counts[r], 256), 8192);
            hipLaunchKernelGGL(k_stamp_globalize, dim3(grid), dim3(256), 0, h->stream, (const uint32_t *)d_st32 + seg[r],
                               (const uint64_t *)d_w1 + seg[r], recv_counts[r], stamp_base[r] << 1, st64 + seg[r],
                               (unsigned long long *)(h->d_scalars + 56));
        }
        HIPCHK(h, hipGetLastError());
        uint64_t sums[2] = {0, 0};
        HIPCHK(h, hipMemcpyAsync(sums, h->d_scalars + 56, 16, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        n_inst = sums[0];
        n_edge = sums[1];
        rc = sk_count_from_segments<uint64_t, 4096>(h, k, seg_start, seg_cnt, (uint32_t)n_shards, n_rec, n_inst, n_edge,
                                                    (const uint64_t *)d_w0, (const uint64_t *)d_w1, st64, w0, w1, st, 0,
                                                    shard_bits, my_shard);
    }
    if (rc != DBG_OK) { const std::string keep = h->err; free_build(h); h->err = keep; return rc; }
    // the instance counters describe this shard's nodes from here on
    h->n_kmer_inst = n_inst;
    h->n_edge_inst = n_edge;
    for (int d = 0; d < n_shards; ++d) { q_starts[d] = sh.q_start[d]; q_counts[d] = sh.q_cnt[d]; }
    *d_q_keys = h->ar_shard[0].p;
    h->partial_graph = n_shards > 1;
    h->stats.ms_build_total = t_total.stop();
    return DBG_OK;
}

extern "C" int dbg_shard_answer(dbg_t *h, const void *d_q_keys, uint64_t n, void *d_answers) {
    if (!h || !h->k || (n && (!d_q_keys || !d_answers))) return DBG_E_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    if (!n) return DBG_OK;
    unsigned long long *sc_dev = (unsigned long long *)h->d_scalars;
    HIPCHK(h, hipMemsetAsync(d_answers, 0xFF, n * 4, h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_scalars, 0, 8, h->stream));
    const SkRange *ranges = (const SkRange *)h->ar_misc[6].p;
    const SkDirEnt *dirs = (const SkDirEnt *)h->ar_dir.p;
    if (!ranges || !dirs || h->sk_cap != 4096) { h->err = "dbg_shard_build must run first"; return DBG_E_ARG; }
    if (h->k > 31) {  // two-word k-mers: the keys come as (lo, hi) pairs
        if (!h->d_keys_hi) { h->err = "dbg_shard_build (LDS engine) must run first"; return DBG_E_ARG; }
        hipLaunchKernelGGL(k_wsucc_resolve, dim3(grid_for(n, 256)), dim3(256), 0, h->stream, (const uint64_t *)d_q_keys,
                           (const uint64_t *)d_q_keys + 1, 2, (const uint32_t *)nullptr, n, h->sk_geom, ranges, h->sk_n_buckets,
                           h->sk_n_ranges, dirs, h->d_keys, h->d_keys_hi, h->n_nodes, (uint32_t *)d_answers, 0u, sc_dev);
    } else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_succ_resolve<4096>), dim3(grid_for(n, 256)), dim3(256), 0, h->stream,
                       (const uint64_t *)d_q_keys, (const uint32_t *)nullptr, n, h->sk_geom, ranges, h->sk_n_buckets,
                       h->sk_n_ranges, dirs, h->d_keys, h->n_nodes, (uint32_t *)d_answers, 0u, sc_dev, (const uint64_t *)nullptr);
    HIPCHK(h, hipGetLastError());
    uint64_t sc0 = 0;
    HIPCHK(h, hipMemcpyAsync(&sc0, h->d_scalars, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (sc0 & 128) { h->err = "a queried successor k-mer is not a node of this shard"; return DBG_E_HIP; }
    return DBG_OK;
}

// ---- gathered graph: node arrays of all shards, concatenated in shard order, become this handle's graph
__global__ __launch_bounds__(256) void k_import_fix(uint64_t n_nodes, const uint64_t *__restrict__ stamps,
                                                    const uint32_t *__restrict__ cnt, uint32_t *succ,
                                                    const uint64_t *__restrict__ shard_base /* [8] */, uint8_t *flags,
                                                    uint8_t *order, uint8_t *deg, unsigned long long *scalars) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const uint4 c4 = reinterpret_cast<const uint4 *>(cnt)[i];
    const uint32_t c[4] = {c4.x, c4.y, c4.z, c4.w};
    uint4 s4 = reinterpret_cast<const uint4 *>(succ)[i];
    uint32_t s[4] = {s4.x, s4.y, s4.z, s4.w};
    unsigned long long edges = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        edges += c[b];
        if (s[b] == NO_NODE) {
            if (c[b]) atomicOr(&scalars[0], 512ull);  // a counted successor without a node
            continue;
        }
        const uint64_t id = shard_base[s[b] >> 29] + (s[b] & ((1u << 29) - 1));
        if (id >= n_nodes) { atomicOr(&scalars[0], 512ull); s[b] = NO_NODE; continue; }
        s[b] = (uint32_t)id;
    }
    reinterpret_cast<uint4 *>(succ)[i] = make_uint4(s[0], s[1], s[2], s[3]);
    flags[i] = (uint8_t)(stamps[i] & 1);
    deg[i] = (uint8_t)((c[0] != 0) + (c[1] != 0) + (c[2] != 0) + (c[3] != 0));
    uint32_t r1 = 0, r2 = 0, r3 = 0;  // rank by (count desc, ASCII order A C G T = codes 0 1 3 2), as in k_sk_count
    { const uint32_t f = c[0] >= c[1]; r1 += f; }
    { const uint32_t f = c[0] >= c[3]; r3 += f; }
    { const uint32_t f = c[0] >= c[2]; r2 += f; }
    { const uint32_t f = c[1] >= c[3]; r3 += f; r1 += 1u - f; }
    { const uint32_t f = c[1] >= c[2]; r2 += f; r1 += 1u - f; }
    { const uint32_t f = c[3] >= c[2]; r2 += f; r3 += 1u - f; }
    order[i] = (uint8_t)((1u << (2 * r1)) | (2u << (2 * r2)) | (3u << (2 * r3)));
    edges = wave_sum_u64(edges);
    if ((threadIdx.x & 63) == 0 && edges) atomicAdd(&scalars[2], edges);
}

__global__ __launch_bounds__(256) void k_flags_from_stamps(uint64_t n, const uint64_t *__restrict__ stamps, uint8_t *flags) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flags[i] = (uint8_t)(stamps[i] & 1);
}
struct CountSum {
    const uint32_t *cnt;
    __device__ uint64_t operator()(uint64_t i) const { return cnt[i]; }
};

extern "C" int dbg_import_graph(dbg_t *h, int k, int n_shards, const uint64_t *shard_nodes, const void *d_keys,
                                const void *d_keys_hi, const void *d_stamps, const void *d_counts, const void *d_succ) {
    if (!h || !shard_nodes || n_shards < 1 || n_shards > 8) return DBG_E_ARG;
    if (k < 1 || k > 63) { h->err = "k must be in 1..63"; return DBG_E_ARG; }
    const bool wide = k > 31;
    if (!h->d_offsets) { h->err = "set the gathered reads first"; return DBG_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    free_build(h);
    CHK(compute_alphabet(h));
    if (!h->is_dna) { h->err = "gathered graphs are ACGT graphs"; return DBG_E_ALPHABET; }
    uint64_t base[8] = {0}, n = 0;
    for (int s = 0; s < n_shards; ++s) {
        base[s] = n;
        if (k <= 31 && shard_nodes[s] >= (1ull << 29)) { h->err = "a shard holds at most 2^29 nodes"; return DBG_E_ARG; }
        n += shard_nodes[s];
    }
    if (n >= 0xFFFFFFF0ull) { h->err = "more than 2^32-16 nodes"; return DBG_E_CAPACITY; }
    if (n && (!d_keys || !d_stamps || !d_counts || (!wide && !d_succ) || (wide && !d_keys_hi))) return DBG_E_ARG;
    h->k = k;
    h->stats = dbg_stats_t{};
    Timer t(h->stream);
    // The big arrays are BORROWED, not copied (eight BASELINE-size shards are 140 GB of them: a second copy would
    // not fit): the caller keeps them alive and leaves them alone while this graph is in use.  Flags, rank bytes and
    // degrees are new (arena); successor ids are rewritten in place.
    CHK(buf_ensure(h, h->ar_node[3], n));
    CHK(buf_ensure(h, h->ar_node[4], n));
    CHK(buf_ensure(h, h->ar_node[6], n));
    h->d_keys = (uint64_t *)d_keys;
    h->d_stamps = (uint64_t *)d_stamps;
    h->d_cnt = (uint32_t *)d_counts;
    h->d_flags = (uint8_t *)h->ar_node[3].p;
    h->d_order = (uint8_t *)h->ar_node[4].p;
    h->d_deg = (uint8_t *)h->ar_node[6].p;
    if (wide) {
        CHK(buf_ensure(h, h->ar_node[5], n * 16));
        h->d_succ = (uint32_t *)h->ar_node[5].p;
    } else {
        h->d_succ = (uint32_t *)d_succ;
    }
    h->nodes_in_arena = true;  // free_build must not free any of these
    h->n_nodes = n;
    if (wide) {
        // two-word k-mers: the shards did not resolve successors (dbg_wide.h).  One table over all nodes, then the
        // same successor kernel as the single-GPU build (it also writes the rank bytes and the degrees).
        h->d_keys_hi = (uint64_t *)d_keys_hi;
        const uint64_t cap = (2 * n + 1024) / 1024 * 1024;  // half full
        CHK(buf_ensure(h, h->ar_wide[1], cap * sizeof(WSlot)));
        WSlot *tab = (WSlot *)h->ar_wide[1].p;
        HIPCHK(h, hipMemsetAsync(tab, 0xFF, cap * sizeof(WSlot), h->stream));
        if (n) {
            hipLaunchKernelGGL(k_wnode_insert, dim3(grid_for(n, 256)), dim3(256), 0, h->stream, n, h->d_keys, h->d_keys_hi, tab,
                               cap);
            hipLaunchKernelGGL(k_wsucc, dim3(grid_for(n, 256)), dim3(256), 0, h->stream, tab, cap, k, n, h->d_keys,
                               h->d_keys_hi, h->d_cnt, h->d_succ, h->d_order, h->d_deg);
            hipLaunchKernelGGL(k_flags_from_stamps, dim3(grid_for(n, 256)), dim3(256), 0, h->stream, n, h->d_stamps, h->d_flags);
            HIPCHK(h, hipGetLastError());
        }
        uint64_t edges = 0;
        CHK(reduce_sum(h, n * 4, CountSum{h->d_cnt}, &edges));
        h->n_edge_inst = edges;
        h->n_kmer_inst = 0;
        int rcw = finish_graph(h);
        if (rcw != DBG_OK) { const std::string keep = h->err; free_build(h); h->err = keep; return rcw; }
        h->stats.ms_build_total = t.stop();
        return DBG_OK;
    }
    HIPCHK(h, hipMemsetAsync(h->d_scalars, 0, 64 * 8, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_scalars + 48, base, sizeof(base), hipMemcpyHostToDevice, h->stream));
    if (n) {
        hipLaunchKernelGGL(k_import_fix, dim3(grid_for(n, 256)), dim3(256), 0, h->stream, n, h->d_stamps, h->d_cnt, h->d_succ,
                           h->d_scalars + 48, h->d_flags, h->d_order, h->d_deg, (unsigned long long *)h->d_scalars);
        HIPCHK(h, hipGetLastError());
    }
    uint64_t sc[4] = {0, 0, 0, 0};
    HIPCHK(h, hipMemcpyAsync(sc, h->d_scalars, 32, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (sc[0] & 512) { h->err = "gathered successor ids do not fit the shard sizes"; free_build(h); return DBG_E_ARG; }
    h->n_edge_inst = sc[2];
    h->n_kmer_inst = 0;  // not carried by the shards
    int rc = finish_graph(h);
    if (rc != DBG_OK) { const std::string keep = h->err; free_build(h); h->err = keep; return rc; }
    h->stats.ms_build_total = t.stop();
    return DBG_OK;
}

extern "C" int dbg_shard_apply(dbg_t *h, const void *d_answers) {
    if (!h || !h->k || !h->shard_state) return DBG_E_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    ShardState &sh = shard_of(h);
    Timer t(h->stream);
    HIPCHK(h, hipMemsetAsync(h->d_scalars, 0, 8, h->stream));
    // local successor ids already carry this shard's tag (k_sk_count / k_succ_resolve wrote them tagged)
    const uint32_t *qcol = (const uint32_t *)h->ar_shard[3].p;
    for (int d = 0; d < sh.n_shards; ++d) {
        if (d == sh.my_shard || !sh.q_cnt[d]) continue;
        if (!d_answers) { h->err = "answers missing"; return DBG_E_ARG; }
        hipLaunchKernelGGL(k_apply_remote, dim3(grid_for(sh.q_cnt[d], 256)), dim3(256), 0, h->stream, qcol + sh.q_start[d],
                           (const uint32_t *)d_answers + sh.q_start[d], sh.q_cnt[d], (uint32_t)d << 29, h->d_col,
                           (unsigned long long *)h->d_scalars);
    }
    HIPCHK(h, hipGetLastError());
    uint64_t sc0 = 0;
    HIPCHK(h, hipMemcpyAsync(&sc0, h->d_scalars, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (sc0 & 256) { h->err = "a remote successor came back unresolved"; return DBG_E_HIP; }
    h->stats.ms_succ += t.stop();
    return finish_graph(h);
}

// ==========================================================================================
// Multi-pass build (BASELINE.json configs[3]: "multi-pass radix buckets spilled to HBM"): graphs with more nodes
// than one 32-bit id space, or than the working set of one pass should hold.
// The reads are cut into super-k-mer records ONCE and split by the 512 level-1 buckets; the records stay parked
// in HBM.  Pass p then builds the node table of the level-1 buckets [p * 512 / P, (p + 1) * 512 / P) exactly like
// the owner of a shard does (second multisplit level -> per-bucket LDS count -> CSR), into arrays of its own that
// stay parked as well: part p of the graph.  A node id is (part, local id); a CSR column holds the local id and a
// byte beside it the part, so the graph may hold up to 256 x (2^32 - 16) nodes.  Successors that live in another
// part are resolved after the last pass through that part's directories (k_succ_resolve).
// ==========================================================================================
struct MultiPass {
    int n_passes = 0;                  // parts on this handle
    int n_virtual = 0, v_first = 0;    // part p is virtual shard v_first + p of n_virtual (ranks x passes; one GPU: n_passes, 0)
    std::vector<dbg *> part;
    std::vector<uint8_t *> col_owner;  // per part: [n_edges of the part] owner part of every CSR column
    std::vector<uint8_t *> pflags;     // per part: [n_nodes] traversal flags (DBG_F_* layout + DBG_PF_*), dbg_part_prune allocates them
    std::vector<uint64_t> base;        // global id of the part's first node (prefix sums of the part sizes)
};

// A finished part keeps only what later reads need -- node arrays, CSR, ranges and directory -- cut to their real
// sizes (they were sized from an estimate, x 1.2): the parts are parked in HBM side by side.
static int buf_shrink(dbg *h, dbg::Buf &b, uint64_t bytes) {
    if (bytes == 0) bytes = 16;
    if (!b.p || b.bytes <= bytes + (16ull << 20)) return DBG_OK;
    void *p = nullptr;
    HIPCHK(h, hipMallocAsync(&p, bytes, h->stream));
    hipError_t e = hipMemcpyAsync(p, b.p, bytes, hipMemcpyDeviceToDevice, h->stream);
    if (e != hipSuccess) { (void)hipFreeAsync(p, h->stream); h->err = std::string("buf_shrink: ") + hipGetErrorString(e); return DBG_E_HIP; }
    (void)hipFreeAsync(b.p, h->stream);  // stream order: after the copy
    h->arena_freed = true;
    b.p = p;
    b.bytes = bytes;
    return DBG_OK;
}

static int part_compact(dbg *sub, uint64_t n_dir_entries) {
    const uint64_t n = sub->n_nodes, ne = sub->n_edges;
    const int st_slot = sub->stamps_st_bytes == 8 ? 1 : 7;
    CHK(buf_shrink(sub, sub->ar_node[0], n * 8));
    if (sub->k > 31) {  // two-word k-mers: the high key words stay, the gathered record bases go
        CHK(buf_shrink(sub, sub->ar_wide[5], n * 8));
        sub->d_keys_hi = (uint64_t *)sub->ar_wide[5].p;
        for (int i : {0, 1, 2, 3, 4}) buf_free(sub, sub->ar_wide[i]);
        buf_free(sub, sub->ar_shard[4]);
    }
    CHK(buf_shrink(sub, sub->ar_node[st_slot], n * (uint64_t)sub->stamps_st_bytes));
    CHK(buf_shrink(sub, sub->ar_node[3], n));
    CHK(buf_shrink(sub, sub->ar_csr[3], (n + 1) * 4));
    CHK(buf_shrink(sub, sub->ar_csr[1], ne * 4));
    CHK(buf_shrink(sub, sub->ar_csr[2], ne * 4));
    CHK(buf_shrink(sub, sub->ar_dir, n_dir_entries * sizeof(SkDirEnt)));
    CHK(buf_shrink(sub, sub->ar_misc[6], sub->sk_n_ranges * sizeof(SkRange)));
    sub->d_keys = (uint64_t *)sub->ar_node[0].p;
    sub->d_stamps_st = sub->ar_node[st_slot].p;
    if (sub->stamps_st_bytes == 8) sub->d_stamps = (uint64_t *)sub->d_stamps_st;
    sub->d_flags = (uint8_t *)sub->ar_node[3].p;
    sub->d_rowptr32 = (uint32_t *)sub->ar_csr[3].p;
    sub->d_col = (uint32_t *)sub->ar_csr[1].p;
    sub->d_ecnt = (uint32_t *)sub->ar_csr[2].p;
    for (auto &lvl : sub->ar_rec) for (auto &b : lvl) buf_free(sub, b);
    for (auto &lvl : sub->ar_q) for (auto &b : lvl) buf_free(sub, b);
    for (int i : {0, 1, 2, 3, 4, 5, 7, 8}) buf_free(sub, sub->ar_misc[i]);
    buf_free(sub, sub->ar_scan);
    sub->sk_src.valid = false;
    return DBG_OK;
}

static void multipass_free(dbg *h) {
    MultiPass *mp = (MultiPass *)h->multipass;
    if (!mp) return;
    for (auto *o : mp->col_owner) if (o) (void)hipFreeAsync(o, h->stream);
    for (auto *o : mp->pflags) if (o) (void)hipFree(o);
    for (dbg *sub : mp->part) {
        if (!sub) continue;
        // A one-pass sharded build repeats on one handle (bench.py): its part handle is parked with its grow-only arenas,
        // like the handle of a single-GPU build keeps its own.  (Freeing ~100 GB into the pool and asking for it again cost a
        // 2 s hipMallocAsync per step at 40 M reads.)
        if (mp->n_passes == 1 && mp->n_virtual <= 8 && !h->spare_part && !h->dropping_parts) { free_build(sub); h->spare_part = sub; }
        else dbg_destroy(sub);
    }
    delete mp;
    h->multipass = nullptr;
}

// the parked part handle of an earlier one-pass sharded build goes when the handle builds anything else
static void drop_spare_part(dbg *h) {
    if (!h->spare_part) return;
    dbg_destroy(h->spare_part);
    h->spare_part = nullptr;
    h->arena_freed = true;
}

static int multipass_starts(dbg *h, uint64_t *total) {
    MultiPass *mp = (MultiPass *)h->multipass;
    *total = 0;
    for (dbg *sub : mp->part) {
        uint64_t t = 0;
        if (sub->n_nodes) CHK(reduce_sum(h, sub->n_nodes, FlagSet{sub->d_flags, DBG_F_INDEG, 0}, &t));
        *total += t;
    }
    return DBG_OK;
}

__global__ __launch_bounds__(256) void k_apply_part(const uint32_t *__restrict__ qcol, const uint32_t *__restrict__ ans,
                                                    uint64_t n, uint8_t owner, uint32_t *col, uint8_t *col_owner,
                                                    unsigned long long *scalars) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t a = ans[i];
    if (a == NO_NODE) { atomicOr(&scalars[0], 256ull); return; }
    col[qcol[i]] = a;
    col_owner[qcol[i]] = owner;
}

// The pass loop of a multi-pass build.  The records (set by the caller: in_w0 / in_w1 / in_st) are split by the 512
// level-1 groups already -- by this GPU itself (dbg_build_multipass: one "sender") or by every rank of a sharded build
// before the exchange (dbg_shard_build_multipass: n_senders segments, each in group order).  This handle owns the
// n_passes * bps groups of the VIRTUAL shards [v_first, v_first + n_passes) out of n_virtual = ranks x passes; pass p
// builds virtual shard v_first + p exactly as a rank of a sharded build does (level 2 (+3) -> count -> CSR) and parks
// it as part p.  grp_cnt[r][g]: records of sender r in the g-th group this handle owns; sender_off[r]: where sender r's
// records start; stamp_add[r]: added to its stamps.  Afterwards successors in another part OF THIS HANDLE are resolved
// through that part's directory; successors owned by another rank stay listed per part (dbg_part_queries).
template <class ST, class STI>
static int multipass_parts(dbg *h, int k, int n_virtual, int v_first, int n_passes, int n_senders, const uint64_t *grp_cnt,
                           const uint64_t *sender_off, const uint64_t *stamp_add, const uint64_t *in_w0,
                           const uint64_t *in_w1, const STI *in_st, const uint64_t *pk = nullptr) {
    // pk: two-word k-mers (k > 31): the packed bases the records' positions (in_w0) point into -- the packed reads of a
    // single-GPU build, or the bases received with the records (record i at position 128 i); the passes then run wsk_count
    int shard_bits = 0;
    while ((1 << shard_bits) < n_virtual) ++shard_bits;
    const int bps = 512 / n_virtual;            // level-1 groups per virtual shard
    const int own_groups = bps * n_passes;      // ... and of this handle
    MultiPass *mp = new MultiPass();
    h->multipass = mp;
    mp->n_passes = n_passes;
    mp->n_virtual = n_virtual;
    mp->v_first = v_first;
    mp->part.assign(n_passes, nullptr);
    mp->col_owner.assign(n_passes, nullptr);
    mp->pflags.assign(n_passes, nullptr);
    mp->base.assign(n_passes + 1, 0);
    const int w = k - sk_m_for_k(k) + 1;
    double ms_count = 0, ms_part = h->stats.ms_partition, ms_succ = 0;
    uint64_t n_buckets = 0, n_queries = 0;
    std::vector<uint64_t> p_cnt((size_t)n_senders * bps), p_off(n_senders);
    for (int p = 0; p < n_passes; ++p) {
        dbg *sub = nullptr;
        const bool reused = n_passes == 1 && h->spare_part;
        if (reused) { sub = h->spare_part; h->spare_part = nullptr; }
        else sub = new (std::nothrow) dbg();
        if (!sub) return DBG_E_NOMEM;
        mp->part[p] = sub;
        sub->device = h->device;
        sub->stream = h->stream;
        sub->borrowed_stream = true;
        sub->wide_owner = true;
        sub->bucket_bits = h->bucket_bits;
        sub->count_kernel = h->count_kernel;
        sub->count_kernel_u64 = h->count_kernel_u64;
        sub->wcount_kernel = h->wcount_kernel;
        sub->resolve_sorted = h->resolve_sorted;
        sub->target_distinct = h->target_distinct;
        sub->est_scale_pct = h->est_scale_pct;
        if (!sub->d_scalars) HIPCHK(h, hipMalloc((void **)&sub->d_scalars, 128 * sizeof(uint64_t)));
        sub->k = k;
        uint64_t n_rec_p = 0;
        for (int r = 0; r < n_senders; ++r) {
            uint64_t before = 0;
            for (int g = 0; g < p * bps; ++g) before += grp_cnt[(size_t)r * own_groups + g];
            p_off[r] = sender_off[r] + before;
            for (int b = 0; b < bps; ++b) {
                p_cnt[(size_t)r * bps + b] = grp_cnt[(size_t)r * own_groups + (size_t)p * bps + b];
                n_rec_p += p_cnt[(size_t)r * bps + b];
            }
        }
        // level-2 output of this pass (one set: the presplit path writes set 0 only); freed again below
        uint64_t *pw0[2] = {nullptr, nullptr}, *pw1[2] = {nullptr, nullptr};
        ST *pst[2] = {nullptr, nullptr};
        CHK(buf_ensure(sub, sub->ar_rec[0][0], (n_rec_p + 16) * 8));
        CHK(buf_ensure(sub, sub->ar_rec[0][1], (n_rec_p + 16) * 8));
        CHK(buf_ensure(sub, sub->ar_rec[0][2], (n_rec_p + 16) * sizeof(ST)));
        pw0[0] = (uint64_t *)sub->ar_rec[0][0].p; pw1[0] = (uint64_t *)sub->ar_rec[0][1].p; pst[0] = (ST *)sub->ar_rec[0][2].p;
        ShardState &sh = shard_of(sub);
        sh.n_shards = n_virtual; sh.my_shard = v_first + p; sh.shard_bits = shard_bits; sh.k = k; sh.n_remote = 0;
        sh.q_start.assign(n_virtual, 0);
        sh.q_cnt.assign(n_virtual, 0);
        Presplit pre;
        pre.n_senders = n_senders;
        pre.counts = p_cnt.data();
        pre.recv_off = p_off.data();
        pre.stamp_add = stamp_add;
        pre.in_st = in_st;
        int rc = DBG_OK;
        if (n_rec_p && pk) {
            rc = wsk_count<ST, STI>(sub, k, pk, nullptr, nullptr, 0, n_rec_p, n_rec_p * (uint64_t)w, n_rec_p * (uint64_t)w, in_w0,
                                    in_w1, (const ST *)nullptr, pw0, pw1, pst, shard_bits, v_first + p, &pre);
        } else if (n_rec_p)
            rc = sk_count_from_segments<ST, 4096, STI>(sub, k, nullptr, nullptr, 0, n_rec_p, n_rec_p * (uint64_t)w, n_rec_p * (uint64_t)w,
                                                       in_w0, in_w1, (const ST *)nullptr, pw0, pw1, pst, 0, shard_bits, v_first + p, &pre);
        if (rc != DBG_OK) { h->err = "pass " + std::to_string(p) + ": " + sub->err; return rc; }
        if (n_rec_p && n_passes > 1) {  // parts parked side by side: cut to size (one part: the estimate's slack may stay)
            const uint64_t own_cnt = sub->sk_n_buckets >> shard_bits;
            rc = part_compact(sub, (own_cnt + (sub->sk_n_ranges - sub->sk_n_buckets)) * (4096 / 64));
            if (rc != DBG_OK) { h->err = "pass " + std::to_string(p) + ": " + sub->err; return rc; }
        }
        mp->base[p + 1] = mp->base[p] + sub->n_nodes;
        if (sub->n_edges) {
            HIPCHK(h, hipMallocAsync((void **)&mp->col_owner[p], sub->n_edges, h->stream));
            HIPCHK(h, hipMemsetAsync(mp->col_owner[p], v_first + p, sub->n_edges, h->stream));
        }
        (void)reused;
        ms_count += sub->stats.ms_count; ms_part += sub->stats.ms_partition; ms_succ += sub->stats.ms_succ;
        n_buckets += sub->stats.n_buckets >> shard_bits; n_queries += sub->stats.n_queries;
        h->stats.count_launches += sub->stats.count_launches;
    }
    // ---- successors owned by another part of this handle: the asker's queries are grouped by owner (ShardState), the
    //      owner's directory answers them
    {
        Timer t(h->stream);
        uint64_t max_q = 0;
        for (int p = 0; p < n_passes; ++p) {
            ShardState &sh = shard_of(mp->part[p]);
            for (int q = 0; q < n_passes; ++q) if (q != p) max_q = std::max(max_q, sh.q_cnt[v_first + q]);
        }
        CHK(buf_ensure(h, h->ar_shard[2], (max_q + 16) * 4));
        uint32_t *ans = (uint32_t *)h->ar_shard[2].p;
        HIPCHK(h, hipMemsetAsync(h->d_scalars, 0, 8, h->stream));
        for (int p = 0; p < n_passes; ++p) {
            dbg *sub = mp->part[p];
            ShardState &sh = shard_of(sub);
            const uint64_t *keys = (const uint64_t *)sub->ar_shard[0].p;
            const uint32_t *qcol = (const uint32_t *)sub->ar_shard[3].p;
            const uint64_t qw = pk ? 2 : 1;  // two-word k-mers: a query is a (lo, hi) pair
            for (int q = 0; q < n_passes; ++q) {
                const int vq = v_first + q;
                if (q == p || !sh.q_cnt[vq]) continue;
                if (!mp->part[q]->n_nodes) { h->err = "part " + std::to_string(q) + " is empty and cannot own a successor"; return DBG_E_HIP; }
                int rc = dbg_shard_answer(mp->part[q], keys + qw * sh.q_start[vq], sh.q_cnt[vq], ans);
                if (rc != DBG_OK) { h->err = "part " + std::to_string(q) + ": " + mp->part[q]->err; return rc; }
                hipLaunchKernelGGL(k_apply_part, dim3(grid_for(sh.q_cnt[vq], 256)), dim3(256), 0, h->stream, qcol + sh.q_start[vq],
                                   ans, sh.q_cnt[vq], (uint8_t)vq, sub->d_col, mp->col_owner[p], (unsigned long long *)h->d_scalars);
            }
            if (n_virtual == n_passes) {  // nothing is owned elsewhere: the query lists are done with
                buf_free(sub, sub->ar_shard[0]);
                buf_free(sub, sub->ar_shard[3]);
            }
        }
        HIPCHK(h, hipGetLastError());
        uint64_t sc0 = 0;
        HIPCHK(h, hipMemcpyAsync(&sc0, h->d_scalars, 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (sc0 & 256) { h->err = "a successor in another part came back unresolved"; return DBG_E_HIP; }
        ms_succ += t.stop();
    }
    h->n_nodes = mp->base[n_passes];
    h->n_edges = 0;
    for (dbg *sub : mp->part) h->n_edges += sub->n_edges;
    h->stats.ms_count = ms_count; h->stats.ms_partition = ms_part; h->stats.ms_succ = ms_succ;
    h->stats.n_buckets = n_buckets; h->stats.n_queries = n_queries;
    h->starts_known = false;
    return DBG_OK;
}

template <class ST>
static int build_multipass_t(dbg *h, int k, int n_passes) {
    uint64_t *w0[2], *w1[2], *seg_start = nullptr, *seg_cnt = nullptr, n_rec = 0;
    ST *st[2];
    uint32_t n_seg = 0;
    CHK(sk_extract<ST>(h, k, w0, w1, st, &seg_start, &seg_cnt, &n_seg, &n_rec));
    const int nb1 = 512;
    CHK(buf_ensure(h, h->ar_misc[1], (uint64_t)nb1 * 16));
    uint64_t *c1_start = (uint64_t *)h->ar_misc[1].p, *c1_cnt = c1_start + nb1;
    std::vector<uint64_t> cnt(nb1), start(nb1);
    {
        Timer t(h->stream);
        CHK((multisplit_level<ST, true>(h, seg_start, seg_cnt, n_seg, n_seg, n_rec, w0[0], w1[0], st[0], w0[1], w1[1], st[1],
                                        6 + SK_BUCKET_BITS - 9, nb1, c1_start, c1_cnt, h->ar_misc[2], h->ar_misc[3],
                                        h->ar_misc[4])));
        HIPCHK(h, hipMemcpyAsync(cnt.data(), c1_cnt, nb1 * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(start.data(), c1_start, nb1 * 8, hipMemcpyDeviceToHost, h->stream));
        h->stats.ms_partition = t.stop();
        for (auto &b : h->ar_rec[0]) buf_free(h, b);  // the unsplit records; the split ones (set 1) stay parked for the passes
    }
    const uint64_t off0 = start[0], add0 = 0;
    int rc = multipass_parts<ST, ST>(h, k, n_passes, 0, n_passes, 1, cnt.data(), &off0, &add0, w0[1], w1[1], st[1]);
    for (auto &b : h->ar_rec[1]) buf_free(h, b);  // every pass has read its slice of the parked records
    return rc;
}

// two-word k-mers (k = 32..63, reads below 2 GiB): records by position into the packed reads, split by the 512 level-1 groups
template <class ST>  // uint32_t; uint64_t for reads of 2 GiB and more
static int build_multipass_wsk(dbg *h, int k, int n_passes) {
    uint64_t *pk = nullptr, *w0[2], *w1[2], *seg_start = nullptr, *seg_cnt = nullptr, n_rec = 0;
    ST *st[2];
    uint32_t n_seg = 0;
    CHK(wsk_extract<ST>(h, k, &pk, w0, w1, st, &seg_start, &seg_cnt, &n_seg, &n_rec));
    const int nb1 = 512;
    CHK(buf_ensure(h, h->ar_misc[1], (uint64_t)nb1 * 16));
    uint64_t *c1_start = (uint64_t *)h->ar_misc[1].p, *c1_cnt = c1_start + nb1;
    std::vector<uint64_t> cnt(nb1), start(nb1);
    {
        Timer t(h->stream);
        CHK((multisplit_level<ST, true>(h, seg_start, seg_cnt, n_seg, n_seg, n_rec, w0[0], w1[0], st[0], w0[1], w1[1], st[1],
                                        6 + SK_BUCKET_BITS - 9, nb1, c1_start, c1_cnt, h->ar_misc[2], h->ar_misc[3],
                                        h->ar_misc[4], 0, nullptr, nullptr, h->host_seg_cnt.data())));
        HIPCHK(h, hipMemcpyAsync(cnt.data(), c1_cnt, nb1 * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(start.data(), c1_start, nb1 * 8, hipMemcpyDeviceToHost, h->stream));
        h->stats.ms_partition = t.stop();
        for (auto &b : h->ar_rec[0]) buf_free(h, b);
    }
    const uint64_t off0 = start[0], add0 = 0;
    int rc = multipass_parts<ST, ST>(h, k, n_passes, 0, n_passes, 1, cnt.data(), &off0, &add0, w0[1], w1[1], st[1], pk);
    for (auto &b : h->ar_rec[1]) buf_free(h, b);
    buf_free(h, h->ar_wide[0]);
    return rc;
}

extern "C" int dbg_build_multipass(dbg_t *h, int k, int n_passes) {
    if (!h) return DBG_E_ARG;
    if (k < 1 || k > 63) { h->err = "multi-pass builds take k in 1..63"; return DBG_E_ARG; }
    if (k > 31 && h->wide_engine != 1) {
        h->err = "multi-pass builds of two-word k-mers use the LDS engine (\"wide_engine\" 1)";
        return DBG_E_ARG;
    }
    if (n_passes < 1 || n_passes > 64 || (n_passes & (n_passes - 1))) { h->err = "n_passes must be a power of two up to 64"; return DBG_E_ARG; }
    if (!h->d_offsets) { h->err = "no reads set"; return DBG_E_ARG; }
    if (h->engine != 0) { h->err = "multi-pass builds use the super-k-mer engine"; return DBG_E_ARG; }
    if (h->bucket_bits && h->bucket_bits < 9) { h->err = "multi-pass builds split by 9 bits first: bucket_bits must be 0 or >= 9"; return DBG_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    h->dropping_parts = true;
    free_build(h);
    h->dropping_parts = false;
    drop_spare_part(h);
    h->arena_freed = true;
    pool_trim(h);  // the parts of an earlier multi-pass build
    h->stats = dbg_stats_t{};
    CHK(compute_alphabet(h));
    if (!h->is_dna) { h->err = "multi-pass builds take ACGT reads"; return DBG_E_ALPHABET; }
    h->k = k;
    Timer t_total(h->stream);
    const bool narrow = h->n_bytes < (1ull << 31) && !h->stamp64;
    int rc = k > 31 ? (narrow ? build_multipass_wsk<uint32_t>(h, k, n_passes) : build_multipass_wsk<uint64_t>(h, k, n_passes))
                    : narrow ? build_multipass_t<uint32_t>(h, k, n_passes) : build_multipass_t<uint64_t>(h, k, n_passes);
    if (rc != DBG_OK) { const std::string keep = h->err; free_build(h); h->arena_freed = true; pool_trim(h); h->err = keep; return rc; }
    h->stats.ms_build_total = t_total.stop();
    h->arena_freed = true;
    pool_trim(h);
    return DBG_OK;
}

// configs[3] on several GPUs: ranks x passes.  A rank of a sharded build whose shard outgrows one 32-bit id space (or
// one piece of memory) builds it as n_passes parts; part p of rank r is VIRTUAL shard r * n_passes + p of
// n_shards * n_passes, and a CSR column's owner byte holds the virtual shard.  Input as for dbg_shard_build with
// sender_bucket_counts.  Afterwards the successors owned by other RANKS are still open: dbg_part_queries lists them per
// part and virtual shard, the owner answers with dbg_part_answer, dbg_part_apply patches them in, dbg_multipass_finish
// checks that nothing stayed open (multi_gpu.sharded_build_multipass runs the exchange).
extern "C" int dbg_shard_build_multipass(dbg_t *h, int k, int n_shards, int my_shard, int n_passes, const void *d_w0,
                                         const void *d_w1, const void *d_st, int stamp_bytes, const uint64_t *recv_counts,
                                         const uint64_t *stamp_base, const uint64_t *sender_bucket_counts) {
    return dbg_shard_build_multipass_from(h, k, n_shards, my_shard, n_passes, n_shards, d_w0, d_w1, d_st, stamp_bytes, recv_counts,
                                          stamp_base, sender_bucket_counts);
}

// the same with n_senders messages in the received arrays (a rank that sends its records in parts -- dbg_shard_extract_part --
// is several senders with one stamp base)
extern "C" int dbg_shard_build_multipass_from(dbg_t *h, int k, int n_shards, int my_shard, int n_passes, int n_senders,
                                              const void *d_w0, const void *d_w1, const void *d_st, int stamp_bytes,
                                              const uint64_t *recv_counts, const uint64_t *stamp_base,
                                              const uint64_t *sender_bucket_counts) {
    CHK(shard_args_ok(h, k, n_shards));
    if (n_senders < 1 || n_senders > 64) { h->err = "1 <= n_senders <= 64"; return DBG_E_ARG; }
    if (k > 31 && (h->wide_engine != 1 || stamp_bytes != 4)) {
        h->err = "ranks x passes of two-word k-mers: the LDS engine's records by value (\"wide_engine\" 1), 4-byte stamps";
        return DBG_E_ARG;
    }
    if (my_shard < 0 || my_shard >= n_shards || !recv_counts || !stamp_base || !sender_bucket_counts) return DBG_E_ARG;
    if (stamp_bytes != 4 && stamp_bytes != 8) { h->err = "stamp_bytes must be 4 or 8"; return DBG_E_ARG; }
    if (n_passes < 1 || (n_passes & (n_passes - 1)) || n_shards * n_passes > 64) {
        h->err = "n_passes must be a power of two with n_shards * n_passes <= 64";
        return DBG_E_ARG;
    }
    if (h->bucket_bits && h->bucket_bits < 9) { h->err = "multi-pass builds split by 9 bits first: bucket_bits must be 0 or >= 9"; return DBG_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    free_build(h);
    // One pass: a plain sharded build with (owner byte, 32-bit local id) successors.  Such builds repeat on one handle
    // (bench.py), and the freed blocks of the last one are exactly what this one asks for: they stay in the pool.
    // Several passes are the memory-tight case: everything goes back to the device first.
    h->arena_freed = true;
    if (n_passes > 1) { drop_spare_part(h); pool_trim(h); }
    h->stats = dbg_stats_t{};
    const int own_groups = 512 / n_shards;
    std::vector<uint64_t> off(n_senders), add(n_senders);
    uint64_t n_rec = 0;
    for (int r = 0; r < n_senders; ++r) {
        uint64_t tot = 0;
        for (int g = 0; g < own_groups; ++g) tot += sender_bucket_counts[(size_t)r * own_groups + g];
        if (tot != recv_counts[r]) { h->err = "sender_bucket_counts do not add up to recv_counts"; return DBG_E_ARG; }
        off[r] = n_rec;
        add[r] = stamp_base[r] << 1;
        n_rec += recv_counts[r];
    }
    h->k = k;
    h->stats.n_records = n_rec;
    Timer t_total(h->stream);
    int rc;
    if (k > 31) {  // d_w0 holds the received bases, four words per record: the receiver's "packed reads", record i at position 128 i
        CHK(buf_ensure(h, h->ar_shard[2], (n_rec + 16) * 8));
        uint64_t *pos = (uint64_t *)h->ar_shard[2].p;
        if (n_rec) hipLaunchKernelGGL(k_wsk_iota128, dim3(grid_for(n_rec, 256)), dim3(256), 0, h->stream, n_rec, pos);
        HIPCHK(h, hipGetLastError());
        rc = multipass_parts<uint64_t, uint32_t>(h, k, n_shards * n_passes, my_shard * n_passes, n_passes, n_senders, sender_bucket_counts,
                                                 off.data(), add.data(), pos, (const uint64_t *)d_w1, (const uint32_t *)d_st,
                                                 (const uint64_t *)d_w0);
    } else
    rc = stamp_bytes == 8
                 ? multipass_parts<uint64_t, uint64_t>(h, k, n_shards * n_passes, my_shard * n_passes, n_passes, n_senders, sender_bucket_counts,
                                                       off.data(), add.data(), (const uint64_t *)d_w0, (const uint64_t *)d_w1, (const uint64_t *)d_st)
                 : multipass_parts<uint64_t, uint32_t>(h, k, n_shards * n_passes, my_shard * n_passes, n_passes, n_senders, sender_bucket_counts,
                                                       off.data(), add.data(), (const uint64_t *)d_w0, (const uint64_t *)d_w1, (const uint32_t *)d_st);
    if (rc != DBG_OK) { const std::string keep = h->err; free_build(h); h->arena_freed = true; pool_trim(h); h->err = keep; return rc; }
    // the instance counters describe this rank's parts
    MultiPass *mp = (MultiPass *)h->multipass;
    h->n_kmer_inst = h->n_edge_inst = 0;
    for (dbg *sub : mp->part) { h->n_kmer_inst += sub->n_kmer_inst; h->n_edge_inst += sub->n_edge_inst; }
    HIPCHK(h, hipMemsetAsync(h->d_scalars, 0, 8, h->stream));
    h->stats.ms_build_total = t_total.stop();
    return DBG_OK;
}

static dbg *part_of(dbg *h, int part) {
    MultiPass *mp = h ? (MultiPass *)h->multipass : nullptr;
    if (!mp || part < 0 || part >= mp->n_passes) { if (h) h->err = "no such part (a multi-pass build must run first)"; return nullptr; }
    return mp->part[part];
}

// successor k-mers of `part` grouped by the virtual shard that owns them: group v at [q_starts[v], q_starts[v] + q_counts[v])
// of *d_q_keys (uint64, device), v in [0, n_shards * n_passes); the groups of this rank's own parts are answered already
extern "C" int dbg_part_queries(dbg_t *h, int part, uint64_t *q_starts, uint64_t *q_counts, const void **d_q_keys) {
    dbg *sub = part_of(h, part);
    if (!sub || !q_starts || !q_counts || !d_q_keys) return DBG_E_ARG;
    MultiPass *mp = (MultiPass *)h->multipass;
    ShardState &sh = shard_of(sub);
    for (int v = 0; v < mp->n_virtual; ++v) {
        const bool mine = v >= mp->v_first && v < mp->v_first + mp->n_passes;
        q_starts[v] = v < (int)sh.q_start.size() ? sh.q_start[v] : 0;
        q_counts[v] = (mine || v >= (int)sh.q_cnt.size()) ? 0 : sh.q_cnt[v];
    }
    *d_q_keys = sub->ar_shard[0].p;
    return DBG_OK;
}

// node ids (uint32, device) of n successor k-mers another rank asked this rank's `part` about
extern "C" int dbg_part_answer(dbg_t *h, int part, const void *d_q_keys, uint64_t n, void *d_answers) {
    dbg *sub = part_of(h, part);
    if (!sub) return DBG_E_ARG;
    if (n && !sub->n_nodes) { h->err = "part " + std::to_string(part) + " is empty and cannot own a successor"; return DBG_E_ARG; }
    int rc = dbg_shard_answer(sub, d_q_keys, n, d_answers);
    if (rc != DBG_OK) h->err = "part " + std::to_string(part) + ": " + sub->err;
    return rc;
}

// d_answers (uint32, device): the answers of virtual shard `owner` to this part's group of queries, in query order
extern "C" int dbg_part_apply(dbg_t *h, int part, int owner, const void *d_answers) {
    dbg *sub = part_of(h, part);
    if (!sub) return DBG_E_ARG;
    MultiPass *mp = (MultiPass *)h->multipass;
    ShardState &sh = shard_of(sub);
    if (owner < 0 || owner >= mp->n_virtual || owner >= (int)sh.q_cnt.size()) { h->err = "no such owner"; return DBG_E_ARG; }
    const uint64_t n = sh.q_cnt[owner];
    if (!n) return DBG_OK;
    if (!d_answers || !sub->ar_shard[3].p) { h->err = "no answers / the part's queries are gone (dbg_multipass_finish ran)"; return DBG_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    const uint32_t *qcol = (const uint32_t *)sub->ar_shard[3].p;
    hipLaunchKernelGGL(k_apply_part, dim3(grid_for(n, 256)), dim3(256), 0, h->stream, qcol + sh.q_start[owner],
                       (const uint32_t *)d_answers, n, (uint8_t)owner, sub->d_col, mp->col_owner[part],
                       (unsigned long long *)h->d_scalars);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));  // d_answers is the caller's
    return DBG_OK;
}

extern "C" int dbg_multipass_finish(dbg_t *h) {
    MultiPass *mp = h ? (MultiPass *)h->multipass : nullptr;
    if (!mp) { if (h) h->err = "a multi-pass build must run first"; return DBG_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    uint64_t sc0 = 0;
    HIPCHK(h, hipMemcpyAsync(&sc0, h->d_scalars, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (dbg *sub : mp->part) { buf_free(sub, sub->ar_shard[0]); buf_free(sub, sub->ar_shard[3]); }
    h->arena_freed = true;
    if (mp->n_passes > 1) pool_trim(h);
    if (sc0 & 256) { h->err = "a successor owned by another rank came back unresolved"; return DBG_E_HIP; }
    return DBG_OK;
}

extern "C" int dbg_part_count(dbg_t *h, int *n_parts) {
    if (!h || !n_parts) return DBG_E_ARG;
    *n_parts = h->multipass ? ((MultiPass *)h->multipass)->n_passes : 0;
    return DBG_OK;
}

extern "C" int dbg_part_sizes(dbg_t *h, int part, uint64_t *n_nodes, uint64_t *n_edges, uint64_t *first_node_id) {
    MultiPass *mp = h ? (MultiPass *)h->multipass : nullptr;
    if (!mp || part < 0 || part >= mp->n_passes) { if (h) h->err = "no such part (dbg_build_multipass must run first)"; return DBG_E_ARG; }
    if (n_nodes) *n_nodes = mp->part[part]->n_nodes;
    if (n_edges) *n_edges = mp->part[part]->n_edges;
    if (first_node_id) *first_node_id = mp->base[part];
    return DBG_OK;
}

template <class ST>
__global__ __launch_bounds__(256) void k_widen(const ST *__restrict__ in, uint64_t n, uint64_t *out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (uint64_t)in[i];
}

extern "C" int dbg_part_device_views(dbg_t *h, int part, const void **d_keys, const void **d_stamps, int *stamp_bytes,
                                     const void **d_flags, const void **d_row_ptr32, const void **d_col, const void **d_col_part,
                                     const void **d_cnt) {
    MultiPass *mp = h ? (MultiPass *)h->multipass : nullptr;
    if (!mp || part < 0 || part >= mp->n_passes) { if (h) h->err = "no such part (dbg_build_multipass must run first)"; return DBG_E_ARG; }
    dbg *sub = mp->part[part];
    if (d_keys) *d_keys = sub->d_keys;
    if (d_stamps) *d_stamps = sub->d_stamps_st;
    if (stamp_bytes) *stamp_bytes = sub->stamps_st_bytes;
    if (d_flags) *d_flags = sub->d_flags;
    if (d_row_ptr32) *d_row_ptr32 = sub->d_rowptr32;
    if (d_col) *d_col = sub->d_col;
    if (d_col_part) *d_col_part = mp->col_owner[part];
    if (d_cnt) *d_cnt = sub->d_ecnt;
    return DBG_OK;
}

// upper key words of a part's nodes (two-word k-mers, k > 31; all zero below): keys_hi[n] to the host and / or the
// device pointer (valid until the next build; NULL for one-word k-mers).  Either output may be NULL.
extern "C" int dbg_part_keys_hi(dbg_t *h, int part, uint64_t *keys_hi, const void **d_keys_hi) {
    MultiPass *mp = h ? (MultiPass *)h->multipass : nullptr;
    if (!mp || part < 0 || part >= mp->n_passes) { if (h) h->err = "no such part (dbg_build_multipass must run first)"; return DBG_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    dbg *sub = mp->part[part];
    const bool wide = sub->k > 31 && sub->d_keys_hi && sub->n_nodes;
    if (d_keys_hi) *d_keys_hi = wide ? sub->d_keys_hi : nullptr;
    if (keys_hi && sub->n_nodes) {
        if (wide) { D2H(h, keys_hi, sub->d_keys_hi, sub->n_nodes * 8); HIPCHK(h, hipStreamSynchronize(h->stream)); }
        else memset(keys_hi, 0, sub->n_nodes * 8);
    }
    return DBG_OK;
}

extern "C" int dbg_export_part(dbg_t *h, int part, uint64_t *keys, uint64_t *stamps, uint8_t *flags, uint64_t *row_ptr,
                               uint32_t *col, uint8_t *col_part, uint32_t *cnt) {
    MultiPass *mp = h ? (MultiPass *)h->multipass : nullptr;
    if (!mp || part < 0 || part >= mp->n_passes) { if (h) h->err = "no such part (dbg_build_multipass must run first)"; return DBG_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    dbg *sub = mp->part[part];
    const uint64_t n = sub->n_nodes, ne = sub->n_edges;
    if (n == 0) {  // a part whose level-1 groups hold no record has no arrays at all
        if (row_ptr) row_ptr[0] = 0;
        return DBG_OK;
    }
    D2H(h, keys, sub->d_keys, n * 8);
    D2H(h, flags, sub->d_flags, n);
    D2H(h, col, sub->d_col, ne * 4);
    D2H(h, col_part, mp->col_owner[part], ne);
    D2H(h, cnt, sub->d_ecnt, ne * 4);
    uint64_t *tmp = nullptr;
    if ((stamps && n) || row_ptr) {
        CHK(dev_alloc(h, &tmp, n + 1));
        if (stamps && n) {
            if (sub->stamps_st_bytes == 4)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_widen<uint32_t>), dim3(grid_for(n, 256)), dim3(256), 0, h->stream,
                                   (const uint32_t *)sub->d_stamps_st, n, tmp);
            else
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_widen<uint64_t>), dim3(grid_for(n, 256)), dim3(256), 0, h->stream,
                                   (const uint64_t *)sub->d_stamps_st, n, tmp);
            hipError_t e = hipMemcpyAsync(stamps, tmp, n * 8, hipMemcpyDeviceToHost, h->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
            if (e != hipSuccess) { (void)hipFree(tmp); h->err = hipGetErrorString(e); return DBG_E_HIP; }
        }
        if (row_ptr) {
            if (n || true)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_widen<uint32_t>), dim3(grid_for(n + 1, 256)), dim3(256), 0, h->stream,
                                   (const uint32_t *)sub->d_rowptr32, n + 1, tmp);
            hipError_t e = hipMemcpyAsync(row_ptr, tmp, (n + 1) * 8, hipMemcpyDeviceToHost, h->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
            if (e != hipSuccess) { (void)hipFree(tmp); h->err = hipGetErrorString(e); return DBG_E_HIP; }
        }
    }
    hipError_t e = hipStreamSynchronize(h->stream);
    if (tmp) (void)hipFree(tmp);
    if (e != hipSuccess) { h->err = hipGetErrorString(e); return DBG_E_HIP; }
    return DBG_OK;
}


// ==========================================================================================
// Traversal of a graph that lives in PARTS (a multi-pass build on one GPU, a rank of a sharded build, ranks x passes):
// prune, branch list, tip removal, pull-out reads and the contig walk without any GPU ever holding the whole graph.
// A node is (virtual shard, local id); the per-part primitives below work on one part and hand small id lists and
// attribute rows to the caller (py-debruijn_amd/part_traversal.py), which moves them between parts and ranks:
//   dbg_part_prune          pruningEdges + branch flag on the part's CSR                  [debruijn.py:150-166, :230-236]
//   dbg_part_select / _gather / _mark   nodes by flag; their rows; set flag bits          (frontiers, tip neighbourhoods)
//   dbg_part_cross_targets  kept edges of chain nodes that leave the part                 (entries of the other parts' chains)
//   dbg_part_segments       from every entry node along the part's own chain nodes        [debruijn.py:288-316, one segment]
//   dbg_scan_reads_for_keys reads that hold one of a few k-mers + first appearance of their edges   [:274-278, Counter order]
//   dbg_set_orders          successor ranks of an imported (tip neighbourhood) graph
// Chains leave a part at almost every change of minimizer (the parts split the minimizer-hash space), so a segment is a
// dozen nodes and one thread per entry walks it; the caller ranks the skeleton of segments (a few per cent of the nodes).
// ==========================================================================================
static int part_ctx(dbg_t *h, int part, MultiPass **mp_out, dbg **sub_out) {
    MultiPass *mp = h ? (MultiPass *)h->multipass : nullptr;
    if (!mp || part < 0 || part >= mp->n_passes) { if (h) h->err = "no such part (a multi-pass build must run first)"; return DBG_E_ARG; }
    *mp_out = mp;
    *sub_out = mp->part[part];
    return DBG_OK;
}

// keep mask of every node from the counts of its CSR row (threshold >= 1: the kept SET does not depend on the Counter order)
__global__ __launch_bounds__(256) void k_part_prune(uint64_t n, const uint8_t *__restrict__ flags, const uint32_t *__restrict__ rowptr,
                                                    const uint32_t *__restrict__ ecnt, double threshold, uint8_t *pflags,
                                                    unsigned long long *n_branch) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t br = 0;
    if (i < n) {
        const uint32_t f = flags[i], pres = (f >> 1) & 15u;
        uint32_t e = rowptr[i], c[4] = {0, 0, 0, 0}, mx = 0;
        for (int b = 0; b < 4; ++b)
            if ((pres >> b) & 1u) { c[b] = ecnt[e++]; mx = max(mx, c[b]); }
        uint32_t keep = 0;
        if (__popc(pres) == 1) keep = pres;
        else if (pres) {
            const double lim = (double)mx / threshold;
            for (int b = 0; b < 4; ++b)
                if (c[b] && (c[b] == mx || (double)c[b] >= lim)) keep |= 1u << b;
        }
        const bool branch = __popc(keep) > 1;
        br = branch;
        pflags[i] = (uint8_t)((f & DBG_F_INDEG) | (keep << DBG_F_KEEP_SHIFT) | (branch ? DBG_F_BRANCH : 0));
    }
    br = wave_sum_u64(br);
    if ((threadIdx.x & 63) == 0 && br) atomicAdd(n_branch, (unsigned long long)br);
}

extern "C" int dbg_part_prune(dbg_t *h, int part, double threshold, uint64_t *n_branch) {
    MultiPass *mp; dbg *sub;
    CHK(part_ctx(h, part, &mp, &sub));
    if (!(threshold >= 1.0)) { h->err = "traversal in parts takes threshold >= 1 (below, the kept successor depends on the Counter order)"; return DBG_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    if (n_branch) *n_branch = 0;
    if (!sub->n_nodes) return DBG_OK;
    if (!mp->pflags[part]) HIPCHK(h, hipMalloc((void **)&mp->pflags[part], (sub->n_nodes + 3) / 4 * 4));
    HIPCHK(h, hipMemsetAsync(h->d_scalars + 8, 0, 8, h->stream));
    hipLaunchKernelGGL(k_part_prune, dim3(grid_for(sub->n_nodes, 256)), dim3(256), 0, h->stream, sub->n_nodes, sub->d_flags,
                       sub->d_rowptr32, sub->d_ecnt, threshold, mp->pflags[part], (unsigned long long *)(h->d_scalars + 8));
    HIPCHK(h, hipGetLastError());
    uint64_t nb = 0;
    HIPCHK(h, hipMemcpyAsync(&nb, h->d_scalars + 8, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (n_branch) *n_branch = nb;
    return DBG_OK;
}

struct PredPflags {
    const uint8_t *f;
    uint8_t mask, want;
    __device__ bool operator()(uint64_t i) const { return (f[i] & mask) == want; }
};

// local ids (uint32, device, ascending) of the nodes with (pflags & mask) == want; *n is always set (d_ids NULL: count only)
extern "C" int dbg_part_select(dbg_t *h, int part, uint32_t mask, uint32_t want, void *d_ids, uint64_t capacity, uint64_t *n) {
    MultiPass *mp; dbg *sub;
    CHK(part_ctx(h, part, &mp, &sub));
    if (!n) return DBG_E_ARG;
    *n = 0;
    if (!sub->n_nodes) return DBG_OK;
    if (!mp->pflags[part]) { h->err = "dbg_part_prune must run first"; return DBG_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    const PredPflags pred{mp->pflags[part], (uint8_t)mask, (uint8_t)want};
    uint64_t total = 0;
    if (!d_ids) { CHK(reduce_sum(h, sub->n_nodes, FlagSet{mp->pflags[part], (uint8_t)mask, (uint8_t)want}, &total)); *n = total; return DBG_OK; }
    CHK(reduce_sum(h, sub->n_nodes, FlagSet{mp->pflags[part], (uint8_t)mask, (uint8_t)want}, &total));
    *n = total;
    if (total > capacity) { h->err = "dbg_part_select: capacity below the number of selected nodes"; return DBG_E_CAPACITY; }
    uint64_t got = 0;
    CHK(compact_ids(h, sub->n_nodes, pred, (uint32_t *)d_ids, &got));
    return DBG_OK;
}

template <class ST>
__global__ __launch_bounds__(256) void k_part_gather(const uint32_t *__restrict__ ids, uint64_t n, uint64_t n_nodes,
                                                     const uint64_t *__restrict__ keys, const uint64_t *__restrict__ keys_hi,
                                                     const ST *__restrict__ stamps, const uint8_t *__restrict__ flags,
                                                     const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ col,
                                                     const uint8_t *__restrict__ col_owner, const uint32_t *__restrict__ ecnt,
                                                     const uint8_t *__restrict__ pflags, uint64_t *o_keys, uint64_t *o_hi,
                                                     uint64_t *o_stamps, uint32_t *o_cnt, uint8_t *o_owner, uint32_t *o_local,
                                                     uint8_t *o_pf, unsigned long long *err) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t i = ids[j];
    if (i >= n_nodes) { atomicOr(err, 1ull); return; }
    if (o_keys) o_keys[j] = keys[i];
    if (o_hi) o_hi[j] = keys_hi ? keys_hi[i] : 0ull;
    if (o_stamps) o_stamps[j] = (uint64_t)stamps[i];
    if (o_pf) o_pf[j] = pflags ? pflags[i] : (uint8_t)(flags[i] & DBG_F_INDEG);
    const uint32_t pres = ((uint32_t)flags[i] >> 1) & 15u;
    uint32_t e = rowptr[i];
    for (int b = 0; b < 4; ++b) {
        uint32_t c = 0, loc = NO_NODE;
        uint8_t ow = 0xFF;
        if ((pres >> b) & 1u) { c = ecnt[e]; loc = col[e]; ow = col_owner[e]; ++e; }
        if (o_cnt) o_cnt[j * 4 + b] = c;
        if (o_local) o_local[j * 4 + b] = loc;
        if (o_owner) o_owner[j * 4 + b] = ow;
    }
}

// rows of n nodes of a part (d_ids: uint32 local ids, device); every output is a device array and may be NULL:
// keys / keys_hi / stamps u64[n], counts u32[n][4] by base code, succ_owner u8[n][4] (virtual shard, 0xFF: no successor),
// succ_local u32[n][4], pflags u8[n] (DBG_F_INDEG alone before dbg_part_prune)
extern "C" int dbg_part_gather(dbg_t *h, int part, const void *d_ids, uint64_t n, void *d_keys, void *d_keys_hi, void *d_stamps,
                               void *d_counts, void *d_succ_owner, void *d_succ_local, void *d_pflags) {
    MultiPass *mp; dbg *sub;
    CHK(part_ctx(h, part, &mp, &sub));
    if (!n) return DBG_OK;
    if (!d_ids || !sub->n_nodes) { h->err = "dbg_part_gather: no ids / the part is empty"; return DBG_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemsetAsync(h->d_scalars + 8, 0, 8, h->stream));
    unsigned long long *err = (unsigned long long *)(h->d_scalars + 8);
#define DBG_PG_ARGS (const uint32_t *)d_ids, n, sub->n_nodes, sub->d_keys, sub->k > 31 ? sub->d_keys_hi : (const uint64_t *)nullptr
#define DBG_PG_REST sub->d_flags, sub->d_rowptr32, sub->d_col, mp->col_owner[part], sub->d_ecnt, mp->pflags[part], (uint64_t *)d_keys, \
                    (uint64_t *)d_keys_hi, (uint64_t *)d_stamps, (uint32_t *)d_counts, (uint8_t *)d_succ_owner, (uint32_t *)d_succ_local, (uint8_t *)d_pflags, err
    if (sub->stamps_st_bytes == 4)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_part_gather<uint32_t>), dim3(grid_for(n, 256)), dim3(256), 0, h->stream, DBG_PG_ARGS,
                           (const uint32_t *)sub->d_stamps_st, DBG_PG_REST);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_part_gather<uint64_t>), dim3(grid_for(n, 256)), dim3(256), 0, h->stream, DBG_PG_ARGS,
                           (const uint64_t *)sub->d_stamps_st, DBG_PG_REST);
#undef DBG_PG_ARGS
#undef DBG_PG_REST
    HIPCHK(h, hipGetLastError());
    uint64_t e = 0;
    HIPCHK(h, hipMemcpyAsync(&e, h->d_scalars + 8, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (e) { h->err = "dbg_part_gather: a node id is out of range"; return DBG_E_ARG; }
    return DBG_OK;
}

__global__ __launch_bounds__(256) void k_part_mark(const uint32_t *__restrict__ ids, uint64_t n, uint64_t n_nodes, uint8_t bits,
                                                   uint8_t *pflags, uint8_t *newly, unsigned long long *err) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t i = ids[j];
    if (i >= n_nodes) { atomicOr(err, 1ull); return; }
    // byte-wide read-modify-write through the enclosing word (several ids may share it)
    uint32_t *w = reinterpret_cast<uint32_t *>(pflags + (i & ~3ull));
    const uint32_t sh = 8u * (i & 3u);
    const uint32_t old = atomicOr(w, (uint32_t)bits << sh);
    if (newly) newly[j] = (uint8_t)((((old >> sh) & bits) != bits) ? 1 : 0);
}

// sets flag bits (DBG_F_PULLED, DBG_PF_*) on n nodes; d_newly (u8[n], device, may be NULL): 1 where not all bits were set
// before.  The same id twice in one call: exactly one of the two reports "newly".
extern "C" int dbg_part_mark(dbg_t *h, int part, const void *d_ids, uint64_t n, uint32_t bits, void *d_newly) {
    MultiPass *mp; dbg *sub;
    CHK(part_ctx(h, part, &mp, &sub));
    if (!n) return DBG_OK;
    if (!d_ids || !sub->n_nodes || !mp->pflags[part]) { h->err = "dbg_part_mark: dbg_part_prune must run first"; return DBG_E_ARG; }
    if (bits & ~0xFFu) return DBG_E_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemsetAsync(h->d_scalars + 8, 0, 8, h->stream));
    hipLaunchKernelGGL(k_part_mark, dim3(grid_for(n, 256)), dim3(256), 0, h->stream, (const uint32_t *)d_ids, n, sub->n_nodes,
                       (uint8_t)bits, mp->pflags[part], (uint8_t *)d_newly, (unsigned long long *)(h->d_scalars + 8));
    HIPCHK(h, hipGetLastError());
    uint64_t e = 0;
    HIPCHK(h, hipMemcpyAsync(&e, h->d_scalars + 8, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (e) { h->err = "dbg_part_mark: a node id is out of range"; return DBG_E_ARG; }
    return DBG_OK;
}

// A chain node: not a branch node, not pulled, exactly one kept successor (debruijn.py:288-316 walks through it).
__device__ inline bool pf_chain(uint32_t pf) {
    return !(pf & (DBG_F_BRANCH | DBG_F_PULLED)) && __popc(pf & DBG_F_KEEP_MASK) == 1;
}
// CSR column of the kept successor of a chain node
__device__ inline uint32_t pf_kept_col(uint32_t flags_present, uint32_t pf, uint32_t row) {
    const uint32_t pres = (flags_present >> 1) & 15u, b = __ffs((pf & DBG_F_KEEP_MASK) >> DBG_F_KEEP_SHIFT) - 1;
    return row + __popc(pres & ((1u << b) - 1u));
}

// pass 0: counts per owner; pass 1: the targets, grouped by owner (cursor[v] starts at the group's offset).  A workgroup
// counts its edges per owner in LDS and takes ONE global atomic per owner it met: per-edge atomics on 64 cursors are
// same-address atomics, ~12 ns each and serialised chip-wide (0.35 s per pass for the 2.9e7 cross edges of a 10 M-read graph).
constexpr int PART_CROSS_TILES = 64;  // tiles of 256 nodes per workgroup
__global__ __launch_bounds__(256) void k_part_cross(uint64_t n, int me, const uint8_t *__restrict__ flags, const uint8_t *__restrict__ pflags,
                                                    const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ col,
                                                    const uint8_t *__restrict__ col_owner, unsigned long long *cursor /* [64] */,
                                                    uint32_t *targets) {
    // A workgroup owns 64 tiles: it counts their cross edges per owner in LDS, reserves ONE range per owner it met, and (fill
    // pass) walks the tiles again with LDS cursors.  One reservation per tile was 3.5e5 workgroups x 3 owners of same-address
    // atomics per pass and part: 6 ms of a 13 ms call at 9e7 nodes.
    __shared__ uint32_t bin[64];
    __shared__ unsigned long long base[64];
    if (threadIdx.x < 64) bin[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t i0 = (uint64_t)blockIdx.x * (256 * PART_CROSS_TILES);
    auto cross = [&](uint64_t i, uint32_t *tgt) -> int {
        if (i >= n) return -1;
        const uint32_t pf = pflags[i];
        if (!pf_chain(pf)) return -1;
        const uint32_t e = pf_kept_col(flags[i], pf, rowptr[i]);
        const int o = col_owner[e];
        if (o == me) return -1;
        *tgt = col[e];
        return o & 63;
    };
    for (int t = 0; t < PART_CROSS_TILES; ++t) {
        uint32_t tgt;
        const int ow = cross(i0 + (uint64_t)t * 256 + threadIdx.x, &tgt);
        if (ow >= 0) atomicAdd(&bin[ow], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        if (bin[threadIdx.x]) base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], (unsigned long long)bin[threadIdx.x]);
        bin[threadIdx.x] = 0;  // the fill pass's running cursor
    }
    if (!targets) return;
    __syncthreads();
    for (int t = 0; t < PART_CROSS_TILES; ++t) {
        uint32_t tgt = 0;
        const int ow = cross(i0 + (uint64_t)t * 256 + threadIdx.x, &tgt);
        if (ow >= 0) targets[base[ow] + atomicAdd(&bin[ow], 1u)] = tgt;
    }
}

// Kept edges of this part's chain nodes that leave the part: the local ids of their targets IN the target's part, grouped by
// target virtual shard (counts[n_virtual], group v at the prefix sum of the counts before it).  d_targets NULL: counts only.
extern "C" int dbg_part_cross_targets(dbg_t *h, int part, uint64_t *counts, void *d_targets, uint64_t capacity) {
    MultiPass *mp; dbg *sub;
    CHK(part_ctx(h, part, &mp, &sub));
    if (!counts) return DBG_E_ARG;
    for (int v = 0; v < mp->n_virtual; ++v) counts[v] = 0;
    if (!sub->n_nodes) return DBG_OK;
    if (!mp->pflags[part]) { h->err = "dbg_part_prune must run first"; return DBG_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    unsigned long long *cur = (unsigned long long *)(h->d_scalars + 64);  // 64 words: the descriptor slot is free outside a build
    HIPCHK(h, hipMemsetAsync(cur, 0, 64 * 8, h->stream));
    const int me = mp->v_first + part;
    hipLaunchKernelGGL(k_part_cross, dim3(grid_for(sub->n_nodes, 256 * PART_CROSS_TILES)), dim3(256), 0, h->stream, sub->n_nodes, me, sub->d_flags,
                       mp->pflags[part], sub->d_rowptr32, sub->d_col, mp->col_owner[part], cur, (uint32_t *)nullptr);
    HIPCHK(h, hipGetLastError());
    uint64_t hc[64];
    HIPCHK(h, hipMemcpyAsync(hc, cur, 64 * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    uint64_t total = 0, off[64];
    for (int v = 0; v < 64; ++v) { off[v] = total; total += hc[v]; if (v < mp->n_virtual) counts[v] = hc[v]; }
    if (!d_targets || !total) return DBG_OK;
    if (total > capacity) { h->err = "dbg_part_cross_targets: capacity below the number of cross edges"; return DBG_E_CAPACITY; }
    HIPCHK(h, hipMemcpyAsync(cur, off, 64 * 8, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_part_cross, dim3(grid_for(sub->n_nodes, 256 * PART_CROSS_TILES)), dim3(256), 0, h->stream, sub->n_nodes, me, sub->d_flags,
                       mp->pflags[part], sub->d_rowptr32, sub->d_col, mp->col_owner[part], cur, (uint32_t *)d_targets);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return DBG_OK;
}

// One segment of a contig walk (debruijn.py:288-316) inside a part.  Entry e is "entered" like DFS(current = e):
//   kind 0 EMIT_HERE   the path ends AT node `last` (a branch node, or no kept successor): emit, last included
//   kind 1 PULLED      e itself is pulled: the path ended at the previous node (nothing, if e is the start)
//   kind 2 REMOTE      the chain leaves the part: next = (owner, local) of the node entered next
//   kind 3 CYCLE       the chain came back to a node of this walk (`current in vec`): the start emits nothing
//   kind 4 NEXT_PULLED the kept successor of `last` is pulled and lives in this part: emit, last included
// hops = edges walked inside the part, score = sum of their counts.
__global__ __launch_bounds__(256) void k_part_segments(const uint32_t *__restrict__ entries, uint64_t n, uint64_t n_nodes, int me,
                                                       const uint8_t *__restrict__ flags, const uint8_t *__restrict__ pflags,
                                                       const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ col,
                                                       const uint8_t *__restrict__ col_owner, const uint32_t *__restrict__ ecnt,
                                                       uint8_t *o_kind, uint8_t *o_owner, uint32_t *o_local, uint32_t *o_hops,
                                                       uint64_t *o_score, uint32_t *o_last) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    uint32_t cur = entries[j], hops = 0, kind = 0, nloc = NO_NODE, nown = 0xFF;
    uint64_t score = 0;
    if (cur >= n_nodes) { o_kind[j] = 3; o_owner[j] = 0xFF; o_local[j] = NO_NODE; o_hops[j] = 0; o_score[j] = 0; o_last[j] = NO_NODE; return; }
    if (pflags[cur] & DBG_F_PULLED) kind = 1;
    else {
        // Brent's cycle test: a chain that stays inside the part for ever visits a node twice
        uint32_t tortoise = cur, power = 1, lam = 0;
        for (;;) {
            const uint32_t pf = pflags[cur];
            if (!pf_chain(pf)) { kind = 0; break; }   // branch node or no kept successor (a pulled node is never entered here)
            const uint32_t e = pf_kept_col(flags[cur], pf, rowptr[cur]);
            const uint32_t ow = col_owner[e], nx = col[e];
            if ((int)ow != me) { kind = 2; nown = ow; nloc = nx; cur = ecnt[e]; break; }  // `last` carries the leaving edge's count
            if (pflags[nx] & DBG_F_PULLED) { kind = 4; break; }
            score += ecnt[e];
            ++hops;
            cur = nx;
            if (cur == tortoise) { kind = 3; break; }
            if (++lam == power) { tortoise = cur; power <<= 1; lam = 0; }
        }
    }
    o_kind[j] = (uint8_t)kind; o_owner[j] = (uint8_t)nown; o_local[j] = nloc; o_hops[j] = hops; o_score[j] = score; o_last[j] = cur;
}

// segments from n entry nodes (d_entries: uint32 local ids, device).  Outputs (device): kind u8[n], next_owner u8[n],
// next_local u32[n], hops u32[n], score u64[n], last u32[n] (the node the segment ended at; kind REMOTE: the count of the edge
// that leaves the part -- the caller adds it when the node it enters is not pulled).
extern "C" int dbg_part_segments(dbg_t *h, int part, const void *d_entries, uint64_t n, void *d_kind, void *d_next_owner,
                                 void *d_next_local, void *d_hops, void *d_score, void *d_last) {
    MultiPass *mp; dbg *sub;
    CHK(part_ctx(h, part, &mp, &sub));
    if (!n) return DBG_OK;
    if (!d_entries || !d_kind || !d_next_owner || !d_next_local || !d_hops || !d_score || !d_last) return DBG_E_ARG;
    if (!sub->n_nodes || !mp->pflags[part]) { h->err = "dbg_part_segments: dbg_part_prune must run first"; return DBG_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    const int me = mp->v_first + part;
    hipLaunchKernelGGL(k_part_segments, dim3(grid_for(n, 256)), dim3(256), 0, h->stream, (const uint32_t *)d_entries, n, sub->n_nodes, me,
                       sub->d_flags, mp->pflags[part], sub->d_rowptr32, sub->d_col, mp->col_owner[part], sub->d_ecnt, (uint8_t *)d_kind,
                       (uint8_t *)d_next_owner, (uint32_t *)d_next_local, (uint32_t *)d_hops, (uint64_t *)d_score, (uint32_t *)d_last);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return DBG_OK;
}

// the characters one segment contributes to a contig: the last base of its entry node, then of every node it appends
__global__ __launch_bounds__(256) void k_part_segment_text(const uint32_t *__restrict__ entries, uint64_t n, uint64_t n_nodes, int me,
                                                          const uint64_t *__restrict__ keys, const uint8_t *__restrict__ flags,
                                                          const uint8_t *__restrict__ pflags, const uint32_t *__restrict__ rowptr,
                                                          const uint32_t *__restrict__ col, const uint8_t *__restrict__ col_owner,
                                                          const uint64_t *__restrict__ off, char *out, uint64_t cap,
                                                          unsigned long long *err) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    uint32_t cur = entries[j];
    uint64_t at = off[j];
    const uint64_t end = off[j + 1];
    if (cur >= n_nodes || end > cap) { atomicOr(err, 1ull); return; }
    if (at < end) out[at++] = code_to_ascii((uint32_t)(keys[cur] & 3u));
    while (at < end) {   // the caller sized the piece from dbg_part_segments' hops: the same walk, inside the part
        const uint32_t pf = pflags[cur];
        if (!pf_chain(pf)) { atomicOr(err, 2ull); return; }
        const uint32_t e = pf_kept_col(flags[cur], pf, rowptr[cur]);
        if ((int)col_owner[e] != me) { atomicOr(err, 2ull); return; }
        cur = col[e];
        out[at++] = code_to_ascii((uint32_t)(keys[cur] & 3u));
    }
}

// Text of n segments (dbg_part_segments): for entry j the characters at [d_off[j], d_off[j + 1]) of d_chars -- the last base
// of the entry node followed by the last base of each of the `hops` nodes the segment appends, so d_off[j + 1] - d_off[j] must
// be 1 + hops[j] (or 0 to skip the segment).  All pointers are device pointers; d_off is uint64[n + 1].
extern "C" int dbg_part_segment_text(dbg_t *h, int part, const void *d_entries, uint64_t n, const void *d_off, void *d_chars,
                                     uint64_t capacity) {
    MultiPass *mp; dbg *sub;
    CHK(part_ctx(h, part, &mp, &sub));
    if (!n) return DBG_OK;
    if (!d_entries || !d_off || !d_chars) return DBG_E_ARG;
    if (!sub->n_nodes || !mp->pflags[part]) { h->err = "dbg_part_segment_text: dbg_part_prune must run first"; return DBG_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemsetAsync(h->d_scalars + 8, 0, 8, h->stream));
    hipLaunchKernelGGL(k_part_segment_text, dim3(grid_for(n, 256)), dim3(256), 0, h->stream, (const uint32_t *)d_entries, n, sub->n_nodes,
                       mp->v_first + part, sub->d_keys, sub->d_flags, mp->pflags[part], sub->d_rowptr32, sub->d_col, mp->col_owner[part],
                       (const uint64_t *)d_off, (char *)d_chars, capacity, (unsigned long long *)(h->d_scalars + 8));
    HIPCHK(h, hipGetLastError());
    uint64_t e = 0;
    HIPCHK(h, hipMemcpyAsync(&e, h->d_scalars + 8, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (e) { h->err = "dbg_part_segment_text: an entry, a piece size or the capacity does not fit the segments of this part"; return DBG_E_ARG; }
    return DBG_OK;
}

__global__ __launch_bounds__(256) void k_part_clear(uint64_t n_words, uint32_t keep4, uint32_t *pflags_words, uint32_t rest) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_words) pflags_words[i] &= keep4;
    else if (i == n_words) {  // the last partial word, byte by byte
        uint8_t *tail = reinterpret_cast<uint8_t *>(pflags_words + n_words);
        for (uint32_t q = 0; q < rest; ++q) tail[q] &= (uint8_t)keep4;
    }
}

// clears flag bits on every node of the part (a mark of one phase, before the next phase uses the bit)
extern "C" int dbg_part_clear(dbg_t *h, int part, uint32_t bits) {
    MultiPass *mp; dbg *sub;
    CHK(part_ctx(h, part, &mp, &sub));
    if (bits & ~0xFFu) return DBG_E_ARG;
    if (!sub->n_nodes || !mp->pflags[part]) return DBG_OK;
    HIPCHK(h, hipSetDevice(h->device));
    const uint32_t keep = ~bits & 0xFFu, keep4 = keep * 0x01010101u;
    const uint64_t n_words = sub->n_nodes / 4;
    hipLaunchKernelGGL(k_part_clear, dim3(grid_for(n_words + 1, 256)), dim3(256), 0, h->stream, n_words, keep4, (uint32_t *)mp->pflags[part],
                       (uint32_t)(sub->n_nodes - n_words * 4));
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return DBG_OK;
}

// device pointer of a part's traversal flags (u8[n_nodes]; NULL before dbg_part_prune)
extern "C" int dbg_part_pflags(dbg_t *h, int part, const void **d_pflags) {
    MultiPass *mp; dbg *sub;
    CHK(part_ctx(h, part, &mp, &sub));
    if (!d_pflags) return DBG_E_ARG;
    *d_pflags = mp->pflags[part];
    return DBG_OK;
}

// ---- reads against a small set of k-mers (the branch k-mers of the WHOLE graph, gathered from all parts and ranks)
struct ScanKey { unsigned long long lo, hi; };
__device__ inline uint64_t scan_hash(uint64_t lo, uint64_t hi) { return mix64(lo * 0x9E3779B97F4A7C15ull ^ mix64(hi + 0x632BE59BD9B4E019ull)); }

__global__ __launch_bounds__(256) void k_scan_insert(const uint64_t *__restrict__ lo, const uint64_t *__restrict__ hi, uint64_t n,
                                                     unsigned long long *tab_lo, unsigned long long *tab_hi, uint32_t *tab_idx,
                                                     uint64_t cap_mask) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long kl = lo[i], kh = hi ? hi[i] : 0ull;
    uint64_t slot = scan_hash(kl, kh) & cap_mask;
    for (uint64_t probe = 0; probe <= cap_mask; ++probe) {
        // claim by index (the keys are distinct: no two threads insert the same one)
        if (atomicCAS(&tab_idx[slot], 0xFFFFFFFFu, (uint32_t)i) == 0xFFFFFFFFu) { tab_lo[slot] = kl; tab_hi[slot] = kh; return; }
        slot = (slot + 1) & cap_mask;
    }
}

// every k-mer window of the reads (two words: any k <= 63) against the set; a hit marks its read and lowers the first
// appearance of the edge (k-mer, next base) -- position of the k-mer in THIS handle's reads -- when the window has a successor.
// A thread owns SCAN_SPAN consecutive positions: one binary search for its first read, then the window rolls base by base
// (the set is small and almost every window misses on its first probe).
constexpr int SCAN_SPAN = 32;
__global__ __launch_bounds__(256) void k_scan_reads(const char *__restrict__ bases, uint64_t n_bytes, const uint64_t *__restrict__ offsets,
                                                    uint64_t n_reads, int k, const unsigned long long *__restrict__ tab_lo,
                                                    const unsigned long long *__restrict__ tab_hi, const uint32_t *__restrict__ tab_idx,
                                                    uint64_t cap_mask, uint8_t *read_flags, unsigned long long *first_seen) {
    const uint64_t p0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * SCAN_SPAN;
    if (p0 >= n_bytes) return;
    uint64_t rlo = 0, rhi = n_reads;  // read of position p0: last r with offsets[r] <= p0
    while (rhi - rlo > 1) {
        const uint64_t mid = (rlo + rhi) >> 1;
        if (offsets[mid] <= p0) rlo = mid; else rhi = mid;
    }
    uint64_t r = rlo, r_end = offsets[r + 1];
    const unsigned long long lo_mask = k < 32 ? (1ull << (2 * k)) - 1 : ~0ull;
    const unsigned long long hi_mask = k <= 32 ? 0ull : (1ull << (2 * k - 64)) - 1;
    unsigned long long lo = 0, hi = 0;
    int have = 0;  // bases of the current read in the window (the window ends at position p + k - 1)
    const uint64_t p_end = min(p0 + SCAN_SPAN, n_bytes);
    for (uint64_t q = p0; q < min(p_end + (uint64_t)k - 1, n_bytes); ++q) {  // q: the base that enters the window
        while (q >= r_end && r + 1 < n_reads) { ++r; r_end = offsets[r + 1]; have = 0; }
        if (q >= r_end) break;
        const uint32_t code = ((uint32_t)(uint8_t)bases[q] >> 1) & 3u;
        hi = ((hi << 2) | (lo >> 62)) & hi_mask;
        lo = ((lo << 2) | code) & lo_mask;
        if (++have < k) continue;
        const uint64_t p = q + 1 - k;  // start of the window: inside this read by construction
        if (p < p0) continue;          // (cannot happen: the window started at or after p0)
        if (p >= p_end) break;
        uint64_t slot = scan_hash(lo, hi) & cap_mask;
        for (uint64_t probe = 0; probe <= cap_mask; ++probe) {
            const uint32_t idx = tab_idx[slot];
            if (idx == 0xFFFFFFFFu) break;
            if (tab_lo[slot] == lo && tab_hi[slot] == hi) {
                read_flags[r] = 1;
                if (first_seen && q + 1 < r_end) {
                    const uint32_t nb = ((uint32_t)(uint8_t)bases[q + 1] >> 1) & 3u;
                    atomicMin(&first_seen[(uint64_t)idx * 4 + nb], (unsigned long long)p);
                }
                break;
            }
            slot = (slot + 1) & cap_mask;
        }
    }
}

// pull_out_read (debruijn.py:274-278) for a graph in parts: read_flags[n_reads] (host) = 1 where this handle's read holds
// one of the n_keys k-mers (host arrays; keys_hi NULL for k <= 32); first_seen[n_keys][4] (host, may be NULL): smallest
// byte offset in this handle's reads of the window k-mer + next base, UINT64_MAX if it never occurs here (the caller takes
// the minimum over ranks after adding each rank's byte base: Counter order of the successors of the branch nodes).
extern "C" int dbg_scan_reads_for_keys(dbg_t *h, int k, const uint64_t *keys, const uint64_t *keys_hi, uint64_t n_keys,
                                       uint8_t *read_flags, uint64_t *first_seen) {
    if (!h || k < 1 || k > 63 || (n_keys && !keys)) return DBG_E_ARG;
    if (!h->d_offsets) { h->err = "no reads set"; return DBG_E_ARG; }
    if (n_keys >= 0xFFFFFFF0ull) { h->err = "too many keys"; return DBG_E_CAPACITY; }
    HIPCHK(h, hipSetDevice(h->device));
    if (read_flags && h->n_reads) memset(read_flags, 0, h->n_reads);
    if (first_seen && n_keys) memset(first_seen, 0xFF, n_keys * 32);
    if (!n_keys || !h->n_reads || h->n_bytes < (uint64_t)k) return DBG_OK;
    uint64_t cap = 1024;
    while (cap < n_keys * 2) cap <<= 1;
    unsigned long long *t_lo = nullptr, *t_hi = nullptr, *d_fs = nullptr;
    uint64_t *d_lo = nullptr, *d_hi = nullptr;
    uint32_t *t_idx = nullptr;
    uint8_t *d_rf = nullptr;
    auto done = [&](int code) { dev_free(t_lo); dev_free(t_hi); dev_free(d_fs); dev_free(d_lo); dev_free(d_hi); dev_free(t_idx); dev_free(d_rf); return code; };
    int rc = dev_alloc(h, &t_lo, cap);
    if (rc == DBG_OK) rc = dev_alloc(h, &t_hi, cap);
    if (rc == DBG_OK) rc = dev_alloc(h, &t_idx, cap);
    if (rc == DBG_OK) rc = dev_alloc(h, &d_lo, n_keys);
    if (rc == DBG_OK) rc = dev_alloc(h, &d_hi, n_keys);
    if (rc == DBG_OK) rc = dev_alloc(h, &d_rf, h->n_reads);
    if (rc == DBG_OK && first_seen) rc = dev_alloc(h, &d_fs, n_keys * 4);
    if (rc != DBG_OK) return done(rc);
    hipError_t e = hipMemsetAsync(t_idx, 0xFF, cap * 4, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_rf, 0, h->n_reads, h->stream);
    if (e == hipSuccess && d_fs) e = hipMemsetAsync(d_fs, 0xFF, n_keys * 32, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_lo, keys, n_keys * 8, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = keys_hi ? hipMemcpyAsync(d_hi, keys_hi, n_keys * 8, hipMemcpyHostToDevice, h->stream) : hipMemsetAsync(d_hi, 0, n_keys * 8, h->stream);
    if (e != hipSuccess) { h->err = std::string("dbg_scan_reads_for_keys: ") + hipGetErrorString(e); return done(DBG_E_HIP); }
    hipLaunchKernelGGL(k_scan_insert, dim3(grid_for(n_keys, 256)), dim3(256), 0, h->stream, d_lo, d_hi, n_keys, t_lo, t_hi, t_idx, cap - 1);
    hipLaunchKernelGGL(k_scan_reads, dim3(grid_for((h->n_bytes + SCAN_SPAN - 1) / SCAN_SPAN, 256)), dim3(256), 0, h->stream, h->d_bases, h->n_bytes, h->d_offsets,
                       h->n_reads, k, t_lo, t_hi, t_idx, cap - 1, d_rf, d_fs);
    e = hipGetLastError();
    if (e == hipSuccess && read_flags) e = hipMemcpyAsync(read_flags, d_rf, h->n_reads, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess && first_seen) e = hipMemcpyAsync(first_seen, d_fs, n_keys * 32, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) { h->err = std::string("dbg_scan_reads_for_keys: ") + hipGetErrorString(e); return done(DBG_E_HIP); }
    return done(DBG_OK);
}

// successor ranks of the current (single-piece) graph: order[n_nodes] bytes in the layout of dbg_export_orders -- what
// dbg_refine_edge_order would compute from the reads, handed in by a caller that knows it (an imported neighbourhood graph)
extern "C" int dbg_set_orders(dbg_t *h, const uint8_t *order) {
    if (!h || !h->k || !order) return DBG_E_ARG;
    if (h->multipass) { h->err = kMultipassGraph; return DBG_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    CHK(ensure_dense(h));
    if (h->D == GEN_D || !h->d_order) { h->err = "dbg_set_orders takes a graph over ACGT"; return DBG_E_ARG; }
    if (h->n_nodes) HIPCHK(h, hipMemcpyAsync(h->d_order, order, h->n_nodes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->order_exact = true;
    return DBG_OK;
}


// ==========================================================================================
// Read-support scores (IV_sortOutputs.py:10-15), see dbg_support.h.  Uses the handle's device and stream only:
// the graph and the reads of the handle are not touched.
// ==========================================================================================
extern "C" int dbg_support_read_scores(dbg_t *h, const char *read_chars, const uint64_t *read_off, uint64_t n_reads,
                                       const double *read_scores, const uint8_t *read_is_float, const char *contig_chars,
                                       const uint64_t *contig_off, uint64_t n_contigs, double *out_scores,
                                       uint32_t *out_float_hits) {
    if (!h || !read_off || !contig_off || (n_reads && !read_scores) || (n_contigs && !out_scores)) return DBG_E_ARG;
    if (n_reads >= SUP_NONE || n_contigs >= SUP_NONE) { h->err = "at most 2^32 - 2 reads and contigs"; return DBG_E_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    if (!n_contigs) return DBG_OK;
    const uint64_t rbytes = read_off[n_reads], cbytes = contig_off[n_contigs];
    if ((rbytes && !read_chars) || (cbytes && !contig_chars)) return DBG_E_ARG;
    int a = 7;
    std::vector<uint32_t> empties;
    for (uint64_t r = 0; r < n_reads; ++r) {
        if (read_off[r + 1] < read_off[r]) { h->err = "offsets must be non-decreasing"; return DBG_E_ARG; }
        const uint64_t len = read_off[r + 1] - read_off[r];
        if (len == 0) empties.push_back((uint32_t)r);
        else a = (int)std::min<uint64_t>((uint64_t)a, len);
    }
    uint64_t cap = 1024;
    while (cap < 2 * (n_reads + 1)) cap <<= 1;
    char *d_r = nullptr, *d_c = nullptr;
    uint64_t *d_roff = nullptr, *d_coff = nullptr;
    double *d_sc = nullptr, *d_out = nullptr;
    uint8_t *d_isf = nullptr;
    unsigned long long *d_keys = nullptr, *d_hits = nullptr, *d_hits2 = nullptr, *d_cnt = nullptr;
    uint32_t *d_head = nullptr, *d_next = nullptr, *d_empty = nullptr, *d_fh = nullptr;
    void *d_tmp = nullptr;
    int rc = DBG_OK;
    auto fail = [&](hipError_t e, const char *what) { h->err = std::string(what) + ": " + hipGetErrorString(e); rc = DBG_E_HIP; };
    do {
        if ((rc = dev_alloc(h, &d_r, rbytes + 8)) || (rc = dev_alloc(h, &d_c, cbytes + 8)) || (rc = dev_alloc(h, &d_roff, n_reads + 1)) ||
            (rc = dev_alloc(h, &d_coff, n_contigs + 1)) || (rc = dev_alloc(h, &d_sc, n_reads)) || (rc = dev_alloc(h, &d_out, n_contigs)) ||
            (rc = dev_alloc(h, &d_isf, n_reads)) || (rc = dev_alloc(h, &d_keys, cap)) || (rc = dev_alloc(h, &d_head, cap)) ||
            (rc = dev_alloc(h, &d_next, n_reads)) || (rc = dev_alloc(h, &d_empty, empties.size())) ||
            (rc = dev_alloc(h, &d_fh, n_contigs)) || (rc = dev_alloc(h, &d_cnt, 1)))
            break;
        hipError_t e = hipSuccess;
        if (rbytes) e = hipMemcpyAsync(d_r, read_chars, rbytes, hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess && cbytes) e = hipMemcpyAsync(d_c, contig_chars, cbytes, hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_roff, read_off, (n_reads + 1) * 8, hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_coff, contig_off, (n_contigs + 1) * 8, hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess && n_reads) e = hipMemcpyAsync(d_sc, read_scores, n_reads * 8, hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess && n_reads && read_is_float) e = hipMemcpyAsync(d_isf, read_is_float, n_reads, hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess && !empties.empty())
            e = hipMemcpyAsync(d_empty, empties.data(), empties.size() * 4, hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_keys, 0xFF, cap * 8, h->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_head, 0xFF, cap * 4, h->stream);
        if (e != hipSuccess) { fail(e, "support scores: upload"); break; }
        if (n_reads)
            hipLaunchKernelGGL(k_sup_insert, dim3(grid_for(n_reads, 256)), dim3(256), 0, h->stream, d_r, d_roff, n_reads, a, d_keys,
                               d_head, d_next, cap - 1);
        uint64_t hit_cap = std::max<uint64_t>(1024, cbytes + empties.size() * n_contigs + 1024), n_hits = 0;
        for (int attempt = 0; attempt < 2 && rc == DBG_OK; ++attempt) {
            if (d_hits) { (void)hipFree(d_hits); d_hits = nullptr; }
            if ((rc = dev_alloc(h, &d_hits, hit_cap)) != DBG_OK) break;
            (void)hipMemsetAsync(d_cnt, 0, 8, h->stream);
            if (cbytes && n_reads > empties.size())
                hipLaunchKernelGGL(k_sup_scan, dim3(grid_for(cbytes, 256)), dim3(256), 0, h->stream, d_c, d_coff, n_contigs, cbytes, a,
                                   d_keys, d_head, d_next, cap - 1, d_r, d_roff, d_hits, hit_cap, d_cnt);
            if (!empties.empty())
                hipLaunchKernelGGL(k_sup_empty, dim3(grid_for(empties.size() * n_contigs, 256)), dim3(256), 0, h->stream, d_empty,
                                   (uint64_t)empties.size(), n_contigs, d_hits, hit_cap, d_cnt);
            e = hipMemcpyAsync(&n_hits, d_cnt, 8, hipMemcpyDeviceToHost, h->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
            if (e != hipSuccess) { fail(e, "support scores: scan"); break; }
            if (n_hits <= hit_cap) break;
            if (attempt == 1) { h->err = "support scores: hit list overflow"; rc = DBG_E_CAPACITY; break; }
            hit_cap = n_hits + 1024;  // one a-mer shared by many reads: take the exact size
        }
        if (rc != DBG_OK) break;
        const unsigned long long *sorted = d_hits;
        if (n_hits > 1) {
            if ((rc = dev_alloc(h, &d_hits2, n_hits)) != DBG_OK) break;
            size_t tmp_bytes = 0;
            e = rocprim::radix_sort_keys(nullptr, tmp_bytes, d_hits, d_hits2, (size_t)n_hits, 0u, 64u, h->stream);
            if (e == hipSuccess) e = hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 1);
            if (e == hipSuccess) e = rocprim::radix_sort_keys(d_tmp, tmp_bytes, d_hits, d_hits2, (size_t)n_hits, 0u, 64u, h->stream);
            if (e != hipSuccess) { fail(e, "support scores: sort"); break; }
            sorted = d_hits2;
        }
        hipLaunchKernelGGL(k_sup_sum, dim3(grid_for(n_contigs, 256)), dim3(256), 0, h->stream, sorted, n_hits, n_contigs, d_sc,
                           read_is_float ? d_isf : (const uint8_t *)nullptr, d_out, d_fh);
        e = hipMemcpyAsync(out_scores, d_out, n_contigs * 8, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess && out_float_hits) e = hipMemcpyAsync(out_float_hits, d_fh, n_contigs * 4, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) { fail(e, "support scores"); break; }
    } while (0);
    dev_free(d_r); dev_free(d_c); dev_free(d_roff); dev_free(d_coff); dev_free(d_sc); dev_free(d_out); dev_free(d_isf);
    dev_free(d_keys); dev_free(d_hits); dev_free(d_hits2); dev_free(d_cnt); dev_free(d_head); dev_free(d_next);
    dev_free(d_empty); dev_free(d_fh);
    if (d_tmp) (void)hipFree(d_tmp);
    return rc;
}

// ==========================================================================================
// Contig sort + FASTA text on the device (II_assembleFromReads.py:64-69): the contigs of the last
// (materialised) walk in the order `sequences.sort(key=getScore, reverse=True)` leaves them in -- stable, so ties keep
// the emission order (start nodes in dict order, emission order inside a start) -- as the text the driver writes:
// ">SEQUENCE_{i}_{k}mer\n{contig}\n".
// ==========================================================================================
struct FaRecLen {  // bytes of record i: header + contig + newline
    const uint32_t *order;
    const uint64_t *off;
    uint32_t k_digits;
    __host__ __device__ static uint32_t digits(uint64_t v) { uint32_t d = 1; while (v >= 10) { v /= 10; ++d; } return d; }
    __device__ uint64_t operator()(uint64_t i) const {
        const uint32_t c = order[i];
        return 10 + digits(i) + 1 + k_digits + 4 + (off[c + 1] - off[c]) + 1;
    }
};

__global__ __launch_bounds__(256) void k_fa_keys(uint64_t n, const uint32_t *__restrict__ ids, const uint64_t *__restrict__ src64,
                                                 const uint32_t *__restrict__ src32, int invert, uint64_t *keys) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = ids ? ids[i] : (uint32_t)i;
    const uint64_t v = src64 ? src64[c] : (uint64_t)src32[c];
    keys[i] = invert ? ~v : v;
}

__global__ __launch_bounds__(256) void k_fa_text(uint64_t n_bytes, uint64_t n_ctg, const uint64_t *__restrict__ pos,
                                                 const uint32_t *__restrict__ order, const uint64_t *__restrict__ off,
                                                 const char *__restrict__ chars, uint32_t k, char *out) {
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_bytes) return;
    uint64_t lo = 0, hi = n_ctg;  // pos[lo] <= b < pos[hi]
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (pos[mid] <= b) lo = mid; else hi = mid;
    }
    const uint64_t i = lo, rel = b - pos[i];
    const uint32_t c = order[i];
    const uint32_t di = FaRecLen::digits(i), dk = FaRecLen::digits(k);
    const uint64_t hdr = 10 + di + 1 + dk + 4, len = off[c + 1] - off[c];
    char ch;
    if (rel < 10) ch = ">SEQUENCE_"[rel];
    else if (rel < 10 + di) { uint64_t v = i; for (uint32_t s = (uint32_t)(10 + di - 1 - rel); s; --s) v /= 10; ch = (char)('0' + v % 10); }
    else if (rel == 10 + di) ch = '_';
    else if (rel < 11 + di + dk) { uint32_t v = k; for (uint32_t s = (uint32_t)(11 + di + dk - 1 - rel); s; --s) v /= 10; ch = (char)('0' + v % 10); }
    else if (rel < hdr) ch = "mer\n"[rel - (11 + di + dk)];
    else if (rel < hdr + len) ch = chars[off[c] + (rel - hdr)];
    else ch = '\n';
    out[b] = ch;
}

extern "C" int dbg_export_sorted_fasta(dbg_t *h, uint32_t *order_out, char *buf, uint64_t buf_len, uint64_t *bytes) {
    if (!h || !bytes) return DBG_E_ARG;
    if (!h->walked) {
        h->err = h->walk_indexed ? "contig text was not materialised (larger than max_chars)" : "dbg_walk must run first";
        return DBG_E_ARG;
    }
    HIPCHK(h, hipSetDevice(h->device));
    const uint64_t n = h->n_contigs;
    *bytes = 0;
    if (!n) return DBG_OK;
    if (n >= 0xFFFFFFF0ull) { h->err = "too many contigs for one sort"; return DBG_E_CAPACITY; }
    uint64_t *ka = nullptr, *kb = nullptr, *pos = nullptr;
    uint32_t *ia = nullptr, *ib = nullptr;
    char *d_out = nullptr;
    void *tmp = nullptr;
    size_t tmp_bytes = 0;
    int rc = DBG_OK;
    do {
        if ((rc = dev_alloc(h, &ka, n)) || (rc = dev_alloc(h, &kb, n)) || (rc = dev_alloc(h, &ia, n)) || (rc = dev_alloc(h, &ib, n)) ||
            (rc = dev_alloc(h, &pos, n + 1)))
            break;
        hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, ka, kb, ia, ib, (size_t)n, 0u, 64u, h->stream);
        if (e == hipSuccess) e = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 1);
        if (e != hipSuccess) { h->err = std::string("sorted fasta: ") + hipGetErrorString(e); rc = DBG_E_HIP; break; }
        const dim3 grid(grid_for(n, 256));
        // emission order = (start stamp, emission index inside the start): two stable passes, minor key first
        hipLaunchKernelGGL(k_iota32, grid, dim3(256), 0, h->stream, n, ia);
        hipLaunchKernelGGL(k_fa_keys, grid, dim3(256), 0, h->stream, n, (const uint32_t *)nullptr, (const uint64_t *)nullptr,
                           (const uint32_t *)h->d_ctg_seq, 0, ka);
        e = rocprim::radix_sort_pairs(tmp, tmp_bytes, ka, kb, ia, ib, (size_t)n, 0u, 32u, h->stream);
        hipLaunchKernelGGL(k_fa_keys, grid, dim3(256), 0, h->stream, n, (const uint32_t *)ib, (const uint64_t *)h->d_ctg_stamp,
                           (const uint32_t *)nullptr, 0, ka);
        if (e == hipSuccess) e = rocprim::radix_sort_pairs(tmp, tmp_bytes, ka, kb, ib, ia, (size_t)n, 0u, 64u, h->stream);
        // ... then the driver's sort: by score, descending, stable
        hipLaunchKernelGGL(k_fa_keys, grid, dim3(256), 0, h->stream, n, (const uint32_t *)ia, (const uint64_t *)h->d_ctg_score,
                           (const uint32_t *)nullptr, 1, ka);
        if (e == hipSuccess) e = rocprim::radix_sort_pairs(tmp, tmp_bytes, ka, kb, ia, ib, (size_t)n, 0u, 64u, h->stream);
        if (e != hipSuccess) { h->err = std::string("sorted fasta: ") + hipGetErrorString(e); rc = DBG_E_HIP; break; }
        uint64_t total = 0;
        if ((rc = exclusive_scan(h, n, FaRecLen{ib, h->d_ctg_off, FaRecLen::digits((uint64_t)h->k)}, pos, &total)) != DBG_OK) break;
        *bytes = total;
        if (order_out) {
            e = hipMemcpyAsync(order_out, ib, n * 4, hipMemcpyDeviceToHost, h->stream);
            if (e != hipSuccess) { h->err = hipGetErrorString(e); rc = DBG_E_HIP; break; }
        }
        if (buf) {
            if (buf_len < total) { h->err = "buffer smaller than the FASTA text"; rc = DBG_E_ARG; break; }
            if ((rc = dev_alloc(h, &d_out, total)) != DBG_OK) break;
            hipLaunchKernelGGL(k_fa_text, dim3(grid_for(total, 256)), dim3(256), 0, h->stream, total, n, pos, ib, h->d_ctg_off,
                               h->d_ctg_chars, (uint32_t)h->k, d_out);
            e = hipMemcpyAsync(buf, d_out, total, hipMemcpyDeviceToHost, h->stream);
            if (e != hipSuccess) { h->err = hipGetErrorString(e); rc = DBG_E_HIP; break; }
        }
        e = hipStreamSynchronize(h->stream);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) { h->err = std::string("sorted fasta: ") + hipGetErrorString(e); rc = DBG_E_HIP; break; }
    } while (0);
    if (tmp) (void)hipFree(tmp);
    dev_free(ka); dev_free(kb); dev_free(ia); dev_free(ib); dev_free(pos); dev_free(d_out);
    return rc;
}
