/*
 * dbg.h -- C ABI of the MI355X (gfx950) de Bruijn graph hot path.
 *
 * The reference (Zhijian-Mei/py-debruijn) has no FFI or plugin interface; the
 * boundary it offers is the Python module surface of debruijn.py as consumed by
 * II_assembleFromReads.py:11-12,58,61,63.  This header is the C-ABI a binding
 * for that surface sits on (ctypes stub: INTEGRATION.md; the shipped binding is
 * py-debruijn_amd/_dbg.py).  Each entry point names the reference lines it
 * replaces.  All pointers are plain host pointers unless the name says
 * "device"; every buffer is owned by the caller; the library owns only the
 * opaque handle and the device memory behind it.
 *
 * Conventions
 *   - every function returns 0 (DBG_OK) or a negative DBG_E_* code; the text of
 *     the last failure is dbg_last_error(h).  Nothing throws or aborts.
 *   - a handle is bound to one GPU and is not thread-safe; calls are
 *     stream-synchronous on return.
 *   - k-mers travel as 2-bit packed keys: base code = (ascii >> 1) & 3
 *     (A=0, C=1, T=2, G=3), first base in the most significant used bits,
 *     key = sum(code[i] << 2*(k-1-i)); 1 <= k <= 31 is one 64-bit word, 32 <= k <= 63 two (the upper word:
 *     dbg_export_keys_hi / dbg_device_keys_hi / dbg_part_keys_hi).
 *   - "stamp" of a node = (byte offset of its first occurrence in the
 *     concatenated read buffer << 1) | (1 if that occurrence is NOT at position
 *     0 of its read).  Ascending stamp == the reference's dict insertion order
 *     (debruijn.py:121-147); stamp & 1 == the reference's Node.indegree.
 *   - reads made of upper-case A/C/G/T only take the 2-bit path above (k <= 63).  Any other
 *     alphabet (the reference is alphabet-agnostic; its real inputs are peptides) takes the
 *     generic path: up to 32 distinct bytes, codes in byte order (dbg_get_alphabet), 32 successor
 *     slots per node, k <= 63 (5 bits per character in one word up to k = 11, tables keyed by
 *     reference into the reads above); DBG_E_ALPHABET for more symbols or k >= 64.
 */
#ifndef DBG_H
#define DBG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dbg dbg_t;

#define DBG_OK 0
#define DBG_E_ARG (-1)      /* bad argument or call order */
#define DBG_E_HIP (-2)      /* HIP runtime failure (text in dbg_last_error) */
#define DBG_E_ALPHABET (-3) /* a read holds a byte outside "ACGT" */
#define DBG_E_CAPACITY (-4) /* hash table or an output limit was exceeded */
#define DBG_E_NOMEM (-5)    /* host or device allocation failed */

#define DBG_ABI_VERSION 5

/* node flag bits (dbg_export_nodes: flags[]) */
#define DBG_F_INDEG 0x01u    /* Node.indegree (0 or 1), debruijn.py:134,141-142 */
#define DBG_F_KEEP_MASK 0x1Eu /* bit (1 + code) set: successor `code` survived pruningEdges */
#define DBG_F_KEEP_SHIFT 1
#define DBG_F_BRANCH 0x20u   /* > 1 surviving successor, debruijn.py:230-235 */
#define DBG_F_PULLED 0x40u   /* in already_pull_out, debruijn.py:249-254 */

#define DBG_NO_NODE 0xFFFFFFFFu

typedef struct dbg_sizes {
    int32_t k;
    int32_t abi_version;
    uint64_t n_reads;
    uint64_t n_bytes;          /* bases in the concatenated read buffer */
    uint64_t n_kmer_instances; /* N_k: k-mer windows of reads with len > k */
    uint64_t n_edge_instances; /* N_e: (k+1)-mer windows */
    uint64_t table_capacity;   /* hash slots */
    uint64_t n_nodes;          /* distinct k-mers == len(vertices) */
    uint64_t n_edges;          /* distinct (k+1)-mers == len(edge_count_table) */
    uint64_t n_branch;
    uint64_t n_pulled;
    uint64_t n_pull_reads;
    uint64_t n_starts;         /* nodes with indegree 0 */
    uint64_t n_contigs;
    uint64_t contig_chars;     /* total characters over all contigs */
    uint64_t tip_rounds;       /* reservation rounds the tip removal needed */
    uint64_t contigs_materialised; /* 1: contig text available (dbg_export_contigs); 0: index only */
    uint64_t max_degree;       /* successor slots per node: 4 (ACGT reads) or 32 (any other alphabet) */
} dbg_sizes_t;

typedef struct dbg_stats {
    /* device time of each phase of the last call, milliseconds (HIP events on the handle's stream) */
    double ms_startbits;  /* read-start bitmap */
    double ms_table_init; /* engine 1: hash table clear */
    double ms_count;      /* encode + hash insert + edge counters (dominant kernel) */
    double ms_compact;    /* engine 1: occupied slots -> node arrays */
    double ms_succ;       /* successor lookup -> 4-way adjacency */
    double ms_csr;        /* degree scan + CSR fill */
    double ms_build_total;
    double ms_prune;
    double ms_tips;
    double ms_pull_reads;
    double ms_walk;
    double ms_h2d;        /* dbg_set_reads copy */
    double ms_extract;    /* engine 0: reads -> super-k-mer records (k_sk_extract) */
    double ms_partition;  /* engine 0: two-level multisplit of the records */
    uint64_t count_launches; /* launches of the dominant kernel in the last dbg_build */
    uint64_t n_records;      /* engine 0: super-k-mer records */
    uint64_t n_buckets;      /* engine 0: final buckets */
    uint64_t n_queries;      /* engine 0: successors resolved across buckets */
} dbg_stats_t;

/* ---- lifetime ---------------------------------------------------------- */
int dbg_create(int device, dbg_t **out);
void dbg_destroy(dbg_t *h);
const char *dbg_last_error(const dbg_t *h);
int dbg_abi_version(void);
/* Tunables (no reference counterpart): "engine" 0 = partitioned super-k-mer build (default),
 * 1 = single global hash table; "bucket_bits" 0 = auto, else log2 of the bucket count (<= 20);
 * "lds_slots" 2048 or 4096 slots of the per-bucket LDS table; "walk_jump_min_nodes" see dbg_walk;
 * "phase_limit" timing ablation of the count kernel (the build then fails on purpose);
 * "estimate_scale_pct" test hook: scales the distinct-k-mer estimate that sizes the node arrays (a low value
 * makes the first count launch run out of room and exercises the retry; dbg_stats_t.count_launches);
 * "refine_streaming" 1: dbg_refine_edge_order takes its pass over the reads even when the bucketed records of the build
 * are there (test hook: both ways must agree).
 * Count kernels (A/B and tests; the defaults are the measured winners): "count_kernel" k <= 31, 32-bit
 * stamps: 2 = k_sk_count2 (default), 1 = k_sk_count, 3 = k_sk_count3; "count_kernel_u64" k <= 31, 64-bit stamps (shards, reads
 * of 2 GiB and more): 3 = k_sk_count3 (default), 1, 2; "wcount_kernel" k > 31, 32-bit stamps: 2 = k_wsk_count2 (default),
 * 1 = k_wsk_count; "stamp64" 1: dbg_build / dbg_build_multipass keep 64-bit stamps below 2 GiB of reads too (what larger
 * inputs get by themselves); "resolve_sorted" 1 (2: at any size): cross-bucket successor queries grouped by their target
 * before the resolver (measured slower: off); "wide_engine" 0: k > 31 on the global-table engine of round 1;
 * "extract_generic" 1: the window-minimum-through-LDS extraction kernels; "shard_stamp64" 1: dbg_shard_extract hands out
 * 64-bit rank-local stamps; "target_distinct": mean distinct k-mers per bucket the geometry aims at (0 = default). */
int dbg_set_option(dbg_t *h, const char *name, int64_t value);

/* ---- reads (replaces the `reads` list argument, debruijn.py:206; FASTA
 *      ingest debruijn.py:22-32 stays on the host side of the boundary) ---- */
/* bases: all reads concatenated without separators; offsets[n_reads+1], offsets[0]==0. Copies H2D. */
int dbg_set_reads(dbg_t *h, const char *bases, const uint64_t *offsets, uint64_t n_reads);
/* read_reads (debruijn.py:22-32) on the device: `text` is the raw FASTA file (host buffer); every line that
 * does not start with '>' becomes one read, rstrip'ed (universal newlines, multi-line records are separate
 * reads, blank lines are empty reads).  Sizes afterwards: dbg_get_sizes; the reads: dbg_copy_reads. */
int dbg_set_reads_fasta(dbg_t *h, const char *text, uint64_t n_text);
/* Zero-copy: device pointers the caller keeps alive (16-byte aligned bases, u64 offsets[n_reads+1]). */
int dbg_set_reads_device(dbg_t *h, const void *d_bases, uint64_t n_bytes, const void *d_offsets, uint64_t n_reads);
/* Generate reads [first_read, first_read+n_reads) of the synthetic set on the device
 * (bit-identical to py-debruijn_amd/synth.py).  err_thr24 = round(err_rate * 2^24). */
int dbg_synth_reads(dbg_t *h, uint64_t seed, uint64_t genome_len, uint64_t first_read, uint64_t n_reads,
                    uint32_t read_len, uint32_t err_thr24);
int dbg_reads_checksum(dbg_t *h, uint64_t *out);              /* synth.checksum twin */
int dbg_copy_reads(dbg_t *h, char *bases, uint64_t *offsets); /* D2H of the current read set */
/* [self[i] for i in indices] of the reference's `reads` list without bringing every read to the host (pull_out_read,
 * debruijn.py:274-278, is a few per cent of the reads): the selected reads, concatenated, gathered on the device.
 * out_offsets[n + 1] (may be NULL) and *n_chars are always set; out_chars may be NULL (first call: sizes only),
 * else capacity >= *n_chars. */
int dbg_take_reads(dbg_t *h, const uint64_t *indices, uint64_t n, uint64_t *out_offsets, char *out_chars, uint64_t capacity,
                   uint64_t *n_chars);
/* device pointers of the current read set (bases: n_bytes chars; offsets: u64[n_reads + 1]); valid until the reads change */
int dbg_reads_device(dbg_t *h, const void **d_bases, uint64_t *n_bytes, const void **d_offsets, uint64_t *n_reads);

/* ---- a3 + a4: get_graph_from_reads (debruijn.py:98-147) + edge-count table (:213-222)
 *      -> node table, 4-way successor edges, CSR ------------------------------------ */
/* table_capacity_hint: 0 = size for the worst case (every k-mer instance distinct);
 * otherwise a slot count (rounded up to a power of two); DBG_E_CAPACITY if too small.
 * k: 1..63 for reads over ACGT (k <= 31: one 64-bit word per k-mer, the partitioned engine;
 * 32..63: two words per k-mer, a reference-keyed global table -- BASELINE.json configs[4]);
 * 1..63 for any other alphabet of at most 32 distinct bytes (k <= 11: 5 bits per character in one word;
 * above: tables keyed by reference into the reads -- such nodes have no packed key, dbg_export_nodes returns
 * zeros for keys and the k-mer of node i is the k bytes at offset stamps[i] >> 1 of the reads). */
int dbg_build(dbg_t *h, int k, uint64_t table_capacity_hint);

/* ---- the same graph in several passes (BASELINE.json configs[3]: more nodes than one 32-bit id space, or than one pass
 *      should hold in flight).  The reads are cut into records once and split by the 512 top-9-bit groups of the
 *      bucket hash; pass p builds the groups [p * 512 / n_passes, (p + 1) * 512 / n_passes) into PART p, whose arrays
 *      stay parked in HBM.  A node id is (part, local id < 2^32 - 16): up to 64 x (2^32 - 16) nodes.  n_passes: a power of two
 *      up to 64; k <= 31; ACGT reads.  Afterwards dbg_get_sizes reports the totals; the graph is read part by part
 *      (below); the traversal entry points refuse it (their node ids are 32-bit). */
int dbg_build_multipass(dbg_t *h, int k, int n_passes);
int dbg_part_count(dbg_t *h, int *n_parts);
/* first_node_id: global id of the part's node 0 when the parts are concatenated in order */
int dbg_part_sizes(dbg_t *h, int part, uint64_t *n_nodes, uint64_t *n_edges, uint64_t *first_node_id);
/* keys[n], stamps[n], flags[n] (DBG_F_INDEG | bit (1 + code): base `code` is a successor), CSR row_ptr[n + 1] and per
 * column: col[e] = local id of the successor in part col_part[e], cnt[e] = count of the (k+1)-mer.  The columns of a
 * row follow the base codes set in flags, ascending.  NULL pointers are skipped. */
int dbg_export_part(dbg_t *h, int part, uint64_t *keys, uint64_t *stamps, uint8_t *flags, uint64_t *row_ptr, uint32_t *col,
                    uint8_t *col_part, uint32_t *cnt);
/* the same arrays on the device (valid until the next build): stamps are uint32 or uint64 (*stamp_bytes), row_ptr uint32 */
int dbg_part_device_views(dbg_t *h, int part, const void **d_keys, const void **d_stamps, int *stamp_bytes,
                          const void **d_flags, const void **d_row_ptr32, const void **d_col, const void **d_col_part,
                          const void **d_cnt);
/* upper key words of a part's nodes (k > 31: the k-mer is keys_hi * 2^64 + keys; all zero for k <= 31): keys_hi[n_nodes]
 * on the host and / or the device pointer (NULL for k <= 31); either output may be NULL */
int dbg_part_keys_hi(dbg_t *h, int part, uint64_t *keys_hi, const void **d_keys_hi);
/* ---- ranks x passes (BASELINE.json configs[3] on several GPUs): a rank of a sharded build that builds its shard as
 *      n_passes parts.  Input as for dbg_shard_build with sender_bucket_counts (stamp_bytes 4 or 8); part p of rank r is
 *      VIRTUAL shard r * n_passes + p of n_shards * n_passes (<= 64), and col_part of a column holds the virtual shard
 *      of the successor.  Successors owned by other RANKS are open afterwards:
 *        dbg_part_queries  the successor k-mers of `part`, group of virtual shard v at [q_starts[v], + q_counts[v]) of
 *                          *d_q_keys (uint64, device); arrays of n_shards * n_passes entries, own rank's groups count 0
 *        dbg_part_answer   the owner: node ids (uint32, device; local to `part`) of n k-mers it was asked about
 *        dbg_part_apply    the asker: d_answers (uint32, device) = answers of virtual shard `owner` to that whole group
 *        dbg_multipass_finish  frees the query lists; fails if any successor stayed unresolved
 *      (multi_gpu.sharded_build_multipass runs these around two all-to-alls per pass). */
int dbg_shard_build_multipass(dbg_t *h, int k, int n_shards, int my_shard, int n_passes, const void *d_w0, const void *d_w1,
                              const void *d_st, int stamp_bytes, const uint64_t *recv_counts, const uint64_t *stamp_base,
                              const uint64_t *sender_bucket_counts);
/* The same with n_senders (1..64) messages in the received arrays: recv_counts[n_senders], stamp_base[n_senders],
 * sender_bucket_counts[n_senders][512 / n_shards].  A rank that sends its records in parts (dbg_shard_extract_part, so that
 * the exchange of one part runs while the next is cut) counts as several senders with one stamp base. */
int dbg_shard_build_multipass_from(dbg_t *h, int k, int n_shards, int my_shard, int n_passes, int n_senders, const void *d_w0,
                                   const void *d_w1, const void *d_st, int stamp_bytes, const uint64_t *recv_counts,
                                   const uint64_t *stamp_base, const uint64_t *sender_bucket_counts);
int dbg_part_queries(dbg_t *h, int part, uint64_t *q_starts, uint64_t *q_counts, const void **d_q_keys);
int dbg_part_answer(dbg_t *h, int part, const void *d_q_keys, uint64_t n, void *d_answers);
int dbg_part_apply(dbg_t *h, int part, int owner, const void *d_answers);
int dbg_multipass_finish(dbg_t *h);

/* ---- traversal of a graph in parts (no reference counterpart for the partition; the steps are debruijn.py:150-186, :230-254,
 *      :274-278, :288-347): the per-part primitives of py-debruijn_amd/part_traversal.py, which moves the small id lists and
 *      rows they produce between parts and ranks.  A node is (virtual shard, local id).  Per part a byte of traversal flags
 *      in the DBG_F_* layout (INDEG, keep mask, BRANCH, PULLED) plus DBG_PF_MARK for the caller's marks. */
#define DBG_PF_MARK 0x80u
/* pruningEdges + branch flag on the part's CSR rows; threshold >= 1 */
int dbg_part_prune(dbg_t *h, int part, double threshold, uint64_t *n_branch);
/* local ids (uint32, device, ascending) of the nodes with (flags & mask) == want; d_ids NULL: *n only */
int dbg_part_select(dbg_t *h, int part, uint32_t mask, uint32_t want, void *d_ids, uint64_t capacity, uint64_t *n);
/* rows of n nodes (d_ids uint32, device); outputs are device arrays, any may be NULL: keys, keys_hi, stamps u64[n],
 * counts u32[n][4] by base code, succ_owner u8[n][4] (virtual shard; 0xFF none), succ_local u32[n][4], pflags u8[n] */
int dbg_part_gather(dbg_t *h, int part, const void *d_ids, uint64_t n, void *d_keys, void *d_keys_hi, void *d_stamps,
                    void *d_counts, void *d_succ_owner, void *d_succ_local, void *d_pflags);
/* sets flag bits on n nodes; d_newly (u8[n], device, may be NULL) = 1 where the bits were not all set before */
int dbg_part_mark(dbg_t *h, int part, const void *d_ids, uint64_t n, uint32_t bits, void *d_newly);
int dbg_part_clear(dbg_t *h, int part, uint32_t bits); /* ... and clears them on every node */
/* kept edges of the part's chain nodes (not branch, not pulled, one kept successor) that leave the part: the targets'
 * local ids grouped by target virtual shard; counts[n_virtual]; d_targets (uint32, device) NULL: counts only */
int dbg_part_cross_targets(dbg_t *h, int part, uint64_t *counts, void *d_targets, uint64_t capacity);
/* one segment of the contig walk per entry node (d_entries uint32, device), inside the part.  kind u8[n]: 0 the path ends
 * at a branch node / a node without kept successor (emitted, that node included), 1 the entry itself is pulled, 2 the chain
 * leaves the part for (next_owner, next_local) -- then `last` holds the count of the leaving edge --, 3 a cycle inside the
 * part, 4 the next node is pulled (emitted up to the current one).  hops u32[n] = edges walked inside the part, score u64[n] =
 * sum of their counts, last u32[n] = node the segment ended at (kind 2: the leaving edge's count). */
int dbg_part_segments(dbg_t *h, int part, const void *d_entries, uint64_t n, void *d_kind, void *d_next_owner, void *d_next_local,
                      void *d_hops, void *d_score, void *d_last);
/* the characters n segments contribute to their contigs: for entry j, at [d_off[j], d_off[j + 1]) of d_chars, the last base of
 * the entry node and of each of the hops[j] nodes the segment appends (d_off[j + 1] - d_off[j] = 1 + hops[j], or 0 to skip);
 * device pointers, d_off uint64[n + 1] */
int dbg_part_segment_text(dbg_t *h, int part, const void *d_entries, uint64_t n, const void *d_off, void *d_chars, uint64_t capacity);
int dbg_part_pflags(dbg_t *h, int part, const void **d_pflags); /* device pointer of the flags, NULL before dbg_part_prune */
/* pull_out_read (debruijn.py:274-278) against an explicit list of k-mers (the branch k-mers of all parts and ranks): host
 * arrays in and out; read_flags[n_reads] = 1 where the read holds one; first_seen[n_keys][4] (may be NULL) = smallest byte
 * offset in this handle's reads of (k-mer, next base by code), UINT64_MAX if absent -- the Counter order of a branch node's
 * successors is count descending, then the minimum of these over all ranks */
int dbg_scan_reads_for_keys(dbg_t *h, int k, const uint64_t *keys, const uint64_t *keys_hi, uint64_t n_keys, uint8_t *read_flags,
                            uint64_t *first_seen);
/* successor ranks (the order bytes of dbg_export_orders) of the current graph, handed in instead of dbg_refine_edge_order */
int dbg_set_orders(dbg_t *h, const uint8_t *order);

/* ---- exact successor order (Counter semantics of debruijn.py:159-165 and :215-216): finds the first
 *      occurrence of every out-edge of the nodes with >= 2 distinct successors (one more pass over the
 *      reads) and rewrites the per-node rank bytes.  Without it equal-count successors are ranked
 *      A<C<G<T; sets, counts and degrees do not depend on it.  Call after dbg_build, before dbg_prune. */
int dbg_refine_edge_order(dbg_t *h);
/* order[n_nodes]: successor base codes, 2 bits each, rank 0 in bits 1:0: count descending, ties by first
 * appearance (== Counter.most_common); fsorder[n_nodes]: by first appearance (== Counter key order). */
int dbg_export_orders(dbg_t *h, uint8_t *order, uint8_t *fsorder);
/* Generic alphabet (max_degree == 32): order / fsorder are [n_nodes][32] bytes, one successor code per
 * rank, 0xFF beyond the out-degree; counts and succ of dbg_export_nodes / dbg_export_succ are [n_nodes][32]. */
/* codes32[code] = the byte a code stands for; bits per symbol 2 (ACGT) or 5 */
int dbg_get_alphabet(dbg_t *h, char *codes32, int *n_symbols, int *bits_per_symbol);
/* surviving successors after dbg_prune, bit `code` per node (both layouts) */
int dbg_export_keepmask(dbg_t *h, uint32_t *keepmask);
/* order[n_nodes]: node ids in the reference's dict order (ascending first-occurrence stamp, debruijn.py:120-133) --
 * the argsort of dbg_export_nodes' stamps, done by a device radix sort. */
int dbg_export_dict_order(dbg_t *h, uint32_t *order);

/* ---- a5 + a6: pruningEdges (debruijn.py:150-166) + branch detection (:230-236) */
int dbg_prune(dbg_t *h, double threshold);
/* ---- a7: tip removal (debruijn.py:169-186 driven by :241-254) */
int dbg_remove_tips(dbg_t *h);
/* ---- a9: reads that contain a branch k-mer (debruijn.py:274-278) */
int dbg_mark_pull_reads(dbg_t *h);
/* ---- a11 + a12: output_contigs / DFS (debruijn.py:288-347); scores are getScore
 *      (II_assembleFromReads.py:14-18).  final_mode != 0 walks with branch_kmer == []
 *      (debruijn.py:281-283).  max_chars bounds the materialised contig text (0 = 1 GiB). */
int dbg_walk(dbg_t *h, int final_mode, uint64_t max_chars);
/* Non-final walks of graphs with >= "walk_jump_min_nodes" nodes (dbg_set_option, default 2^14) resolve every
 * chain by pointer jumping (contigs overlap massively at scale); if the text would exceed max_chars
 * dbg_walk still returns DBG_OK with the contig index only (dbg_get_sizes: contigs_materialised == 0). */

int dbg_get_sizes(dbg_t *h, dbg_sizes_t *out);
int dbg_get_stats(dbg_t *h, dbg_stats_t *out);

/* ---- exports: two-call pattern, sizes from dbg_get_sizes; NULL pointers are skipped */
/* keys[n_nodes], stamps[n_nodes], counts[n_nodes*4] (by base code), flags[n_nodes]; table order */
int dbg_export_nodes(dbg_t *h, uint64_t *keys, uint64_t *stamps, uint32_t *counts, uint8_t *flags);
/* k > 31: a k-mer is the 2k-bit number keys_hi[i] * 2^64 + keys[i] (first base in the top bit pair);
 * keys_hi[n_nodes] is all zero for k <= 31. */
int dbg_export_keys_hi(dbg_t *h, uint64_t *keys_hi);
/* succ[n_nodes*4]: node id of successor by base code or DBG_NO_NODE */
int dbg_export_succ(dbg_t *h, uint32_t *succ);
/* CSR over distinct edges: row_ptr[n_nodes+1], col[n_edges], cnt[n_edges] */
int dbg_export_csr(dbg_t *h, uint64_t *row_ptr, uint32_t *col, uint32_t *cnt);
/* ranks[n_nodes]: pull order key of pulled nodes (ascending == append order of
 * already_pull_out, debruijn.py:253), UINT64_MAX for the rest */
int dbg_export_pull_ranks(dbg_t *h, uint64_t *ranks);
/* The two short label lists of construct_graph without anything of size n_nodes leaving the device: the nodes that
 * carry `flag` -- DBG_F_PULLED in pull order (already_pull_out, debruijn.py:253) or DBG_F_BRANCH in dict order
 * (branch_kmer, debruijn.py:230-236) -- as rows[*n_out] (node ids), keys[*n_out], keys_hi[*n_out] (zeros for
 * k <= 32).  *n_out is always set (rows == NULL: count only); DBG_E_CAPACITY if capacity < *n_out. */
int dbg_export_marked(dbg_t *h, uint32_t flag, uint64_t capacity, uint64_t *n_out, uint32_t *rows, uint64_t *keys,
                      uint64_t *keys_hi);
int dbg_export_pull_reads(dbg_t *h, uint8_t *read_flags /* [n_reads] */);
/* contigs in emission order grouped by start: offsets[n_contigs+1] into chars[contig_chars],
 * scores[n_contigs], start_stamp[n_contigs] (stamp of the start node: sort key for dict order),
 * seq_in_start[n_contigs] (emission index within its start) */
int dbg_export_contigs(dbg_t *h, uint64_t *offsets, char *chars, uint64_t *scores, uint64_t *start_stamp,
                       uint32_t *seq_in_start);
/* the same without the text: offsets[n_contigs+1] (contig i has offsets[i+1]-offsets[i] characters) */
int dbg_export_contig_index(dbg_t *h, uint64_t *offsets, uint64_t *scores, uint64_t *start_stamp, uint32_t *seq_in_start);
/* f2: contig sort + FASTA text on the device (II_assembleFromReads.py:64-69).  The contigs of the last materialised walk,
 * sorted by score descending and stable like `sequences.sort(key=getScore, reverse=True)`, as the text the driver
 * writes: ">SEQUENCE_{i}_{k}mer\n{contig}\n" per contig.  *bytes receives the size of the text (call with buf == NULL to
 * ask for it); order_out (may be NULL): [n_contigs] index of the contig at every output position, indices as in
 * dbg_export_contigs. */
int dbg_export_sorted_fasta(dbg_t *h, uint32_t *order_out, char *buf, uint64_t buf_len, uint64_t *bytes);
/* Text of ONE contig (buf_len >= its length from the index): what a caller uses when dbg_walk kept the index only
 * because the whole text exceeds max_chars (contigs overlap massively at scale).  The first call after a walk builds
 * the binary-lifting tables of the chain successors (n_nodes x ceil(log2 n_nodes) x 4 bytes), later calls are one
 * small kernel each. */
int dbg_export_contig_text(dbg_t *h, uint64_t index, char *buf, uint64_t buf_len);
/* device-side views for callers that stay on the GPU (valid until the next build/destroy) */
int dbg_device_views(dbg_t *h, const void **d_keys, const void **d_counts, const void **d_stamps,
                     const void **d_flags, const void **d_succ);

/* ---- multi-GPU: hash-prefix sharding (no reference counterpart: the reference is one process).
 *      One handle per rank; the caller moves the buffers between ranks (RCCL all-to-all).  Owner
 *      shard of a k-mer = top log2(n_shards) bits of its minimizer bucket hash; n_shards in {1,2,4,8}.
 *      After dbg_shard_apply the handle holds the shard's nodes; successor ids are
 *      (owner shard << 29) | node id on that shard; stamps are global ((byte offset in the
 *      rank-major concatenation of all reads) << 1 | pos != 0). */
/* step 1: this rank's reads -> super-k-mer records grouped by owner (device arrays: w0, w1 uint64,
 * st = rank-local stamps: uint32 while this rank's reads stay below 2 GiB, else uint64 (k <= 31 only; option
 * "shard_stamp64" 1 forces the wide ones) -- dbg_shard_record_layout says which); send_counts[n_shards] records go to
 * each owner, contiguous and in owner order */
int dbg_shard_extract(dbg_t *h, int k, int n_shards, uint64_t *send_counts, const void **d_w0, const void **d_w1,
                      const void **d_st);
/* dbg_shard_extract for part `part` of n_parts (1..4) slices of this rank's reads (k <= 31: super-k-mer records; k > 31: the
 * records by value of the LDS engine, reads below 2 GiB): slices of the position space cut at
 * tile borders -- a k-mer belongs to the slice its first base lies in, so the parts' records together are exactly those of
 * dbg_shard_extract; the stamps are positions in ALL of the rank's reads either way.  The arrays of part p stay valid until
 * part p is extracted again (own buffers per part): multi_gpu.sharded_build_multipass(chunks=...) has part p on the wire
 * while part p + 1 is cut.  dbg_shard_bucket_counts / dbg_shard_record_layout describe the last part extracted. */
int dbg_shard_extract_part(dbg_t *h, int k, int n_shards, int part, int n_parts, uint64_t *send_counts, const void **d_w0,
                           const void **d_w1, const void **d_st);
/* What the last dbg_shard_extract handed out: *w0_words = 64-bit words per record in d_w0 (1: super-k-mer records of
 * k <= 31, or the low key word of the k-mer instances of k > 31 with "wide_engine" 0; 4: the aligned bases of a
 * super-k-mer record of k > 31), *stamp_bytes = bytes per entry of d_st (4; 8 for the instance tuples and for
 * k <= 31 records of a rank that holds 2 GiB of reads or more).  The exchange moves counts[d] x words elements of d_w0;
 * all senders of one dbg_shard_build must use ONE stamp width (a rank with 4-byte stamps zero-extends them when another
 * rank has 8: multi_gpu.sharded_build). */
int dbg_shard_record_layout(dbg_t *h, int *w0_words, int *stamp_bytes);
/* records per level-1 bucket (the 512 top-9-bit groups of the bucket hash) of the last dbg_shard_extract (k <= 31, and k > 31 on the LDS engine):
 * owner d holds the buckets [d * 512 / n_shards, (d + 1) * 512 / n_shards), in order, so send_counts[d] is their sum */
int dbg_shard_bucket_counts(dbg_t *h, uint64_t *counts512);
/* step 2: records received from rank r are recv_counts[r] consecutive entries (rank order);
 * stamp_base[r] = bytes of reads held by ranks < r.  Builds the shard's node table.  Successor
 * k-mers owned by other shards: device array *d_q_keys (uint64), group of owner d at
 * [q_starts[d], q_starts[d] + q_counts[d]).
 * sender_bucket_counts: NULL, or [n_shards][512 / n_shards] -- for every sender its dbg_shard_bucket_counts entries of
 * the buckets THIS shard owns.  With it the receiver skips the first multisplit level (the senders did it before the
 * exchange) and rebases the stamps inside the second one.
 * stamp_bytes: width of the entries of d_st32 -- 0 = the layout's default (4; 8 for k-mer instance tuples), 4, or 8
 * (k <= 31 with sender_bucket_counts: senders that hold 2 GiB of reads or more). */
int dbg_shard_build(dbg_t *h, int k, int n_shards, int my_shard, const void *d_w0, const void *d_w1, const void *d_st32,
                    const uint64_t *recv_counts, const uint64_t *stamp_base, uint64_t *q_starts, uint64_t *q_counts,
                    const void **d_q_keys, const uint64_t *sender_bucket_counts, int stamp_bytes);
/* step 3: node ids (uint32, device) of n successor k-mers other ranks asked this shard about */
int dbg_shard_answer(dbg_t *h, const void *d_q_keys, uint64_t n, void *d_answers);
/* step 4: d_answers (uint32, device) laid out like *d_q_keys of step 2; completes successors + CSR */
int dbg_shard_apply(dbg_t *h, const void *d_answers);
/* Traversal after a sharded build ("gather first"): the shards' node arrays, concatenated in shard
 * order on one GPU (device pointers: keys u64[n], stamps u64[n], counts u32[n][4], succ u32[n][4] with ids
 * (owner << 29) | id), become the graph of handle `h`, whose reads must be the rank-major concatenation of all
 * ranks' reads (the global stamps index into it).  Successor ids are rewritten to positions in the concatenation;
 * the arrays are BORROWED, not copied (they are most of the memory of the gathered graph): keep them alive and do
 * not touch them while this handle uses the graph; d_succ is rewritten in place.
 * Afterwards the handle is what dbg_build would have produced on the whole read set (node order aside):
 * dbg_refine_edge_order, dbg_prune, dbg_remove_tips, dbg_mark_pull_reads, dbg_walk and the exports apply. */
int dbg_import_graph(dbg_t *h, int k, int n_shards, const uint64_t *shard_nodes, const void *d_keys,
                     const void *d_keys_hi, const void *d_stamps, const void *d_counts, const void *d_succ);
/* k > 31 (two-word k-mers): d_keys_hi is required and d_succ is ignored -- the shards of such a build carry keys,
 * stamps and counts only (the k-mer INSTANCES travel between ranks, dbg_shard_extract / dbg_shard_build take and
 * return (lo, hi | next base << 62, local stamp | has-successor << 32) u64 triples and there are no successor
 * queries); dbg_import_graph resolves every successor with one table over the gathered nodes.
 * k <= 31: d_keys_hi may be NULL. */
/* device pointer of the upper key words (NULL for k <= 31) */
int dbg_device_keys_hi(dbg_t *h, const void **d_keys_hi);

/* ---- f4: read-support scores of contigs (findSupportReadScore, IV_sortOutputs.py:10-15): out_scores[c] = sum of
 *      read_scores[r] over the reads r (distinct strings: the reference's dict keys) that occur in contig c as a
 *      substring, added in ascending r from 0.0 -- the reference's order, so double sums equal the reference's bit for
 *      bit; an empty read occurs in every contig.  read_is_float (may be NULL = all): 1 where the reference's score
 *      is a Python float; out_float_hits[c] (may be NULL) counts the float-typed reads found in c (0: the reference's
 *      result is an int).  All pointers are host pointers; reads / contigs are concatenated characters + offsets[n + 1].
 *      Uses the handle's device only; its reads and graph are untouched. */
int dbg_support_read_scores(dbg_t *h, const char *read_chars, const uint64_t *read_off, uint64_t n_reads,
                            const double *read_scores, const uint8_t *read_is_float, const char *contig_chars,
                            const uint64_t *contig_off, uint64_t n_contigs, double *out_scores, uint32_t *out_float_hits);

#ifdef __cplusplus
}
#endif
#endif /* DBG_H */
