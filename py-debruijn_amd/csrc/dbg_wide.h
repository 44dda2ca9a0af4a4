// Two-word k-mers: 32 <= k <= 63 over ACGT (BASELINE.json configs[4]: k = 63, 128-bit keys).
//
// gfx950 has no 128-bit atomic, so the global table does not hold keys at all: a slot holds the
// STAMP (byte offset << 1 | pos != 0) of the earliest instance of its k-mer seen so far, tagged
// with a 16-bit fingerprint, and the key of a slot is "the k-mer at that offset of the reads".
// Claiming a slot is one 64-bit CAS, keeping the first occurrence is one 64-bit atomicMin (every
// value a slot ever holds points at an instance of the same k-mer, so its key never changes), and
// a probe compares against the 2-bit packed copy of the reads -- immutable, hence race-free.  Each slot also
// carries a write-once cache of its key next to the protocol word (struct WSlot): a hit there costs one sector.
// After compaction the protocol word holds the node id.
//
//   k_wpack        ASCII reads -> 2 bits per base, 32 bases per word        [debruijn.py:129-143 input]
//   k_wcount       every k-mer instance: insert / first-occurrence stamp / successor counter
//   k_wgather      occupied slots -> node arrays (keys lo/hi, stamp, counts, indegree flag)
//   k_wsucc        4-way successor ids + count-ranked successor order        [debruijn.py:159-165]
//   k_wset_insert / k_wpull_reads / k_wedge_first_seen   the wide twins of the 64-bit lookups
//                  used by pull_out_read [debruijn.py:248-263] and the Counter tie order
#pragma once
#include "dbg_device.h"

namespace dbgk {

struct K128 {
    uint64_t hi, lo;  // value = hi * 2^64 + lo, 2k bits, first base in the top bit pair
};
__device__ inline bool k128_eq(K128 a, K128 b) { return a.lo == b.lo && a.hi == b.hi; }
__device__ inline uint64_t k128_hash(K128 a) { return mix64(a.lo ^ (mix64(a.hi + 0x9E3779B97F4A7C15ull) * 0xD6E8FEB86659FD93ull)); }

// A = bases 0..31, B = bases 32..63 of a 64-base window (first base in bits 63:62): the first k bases
__device__ inline K128 k128_from_windows(uint64_t A, uint64_t B, int k) {
    const int s = 128 - 2 * k;  // 2 .. 64
    if (s == 64) return K128{0ull, A};
    return K128{A >> s, (A << (64 - s)) | (B >> s)};
}
// base k (0-based) of the window, 32 <= k <= 63
__device__ inline uint32_t base_after_kmer(uint64_t B, int k) { return (uint32_t)(B >> (62 - 2 * (k - 32))) & 3u; }
// successor k-mer: drop the first base, append b
__device__ inline K128 k128_append(K128 key, uint32_t b, int k) {
    K128 r{(key.hi << 2) | (key.lo >> 62), (key.lo << 2) | (uint64_t)b};
    const int hb = 2 * k - 64;  // 0 .. 62
    r.hi = hb ? (r.hi & ((1ull << hb) - 1)) : 0ull;
    return r;
}

constexpr uint64_t W_STAMP_MASK = (1ull << 48) - 1;  // ref = fingerprint << 48 | stamp (later: node id)
constexpr uint64_t W_EMPTY = ~0ull;
// One 32-byte sector per slot.  `ref` is the protocol word (see the header of this file); lo/hi are a CACHE of the slot's
// key, written once by the lane that claimed the slot, all-ones until then (no k-mer has hi == ~0).  A probe that
// finds its own key there is done with one sector; anything else (not written yet, or another key with the same
// fingerprint) falls back to the compare by reference, which is what makes the table exact.
struct alignas(32) WSlot {
    unsigned long long ref, lo, hi, pad;
};

// 64 bases starting at base p of the packed reads (the array is padded with three zero words)
__device__ inline void packed_windows(const uint64_t *__restrict__ pk, uint64_t p, uint64_t &A, uint64_t &B) {
    const uint64_t w = p >> 5;
    const int s = (int)(p & 31) * 2;
    const uint64_t a = pk[w], b = pk[w + 1], c = pk[w + 2];
    A = s ? (a << s) | (b >> (64 - s)) : a;
    B = s ? (b << s) | (c >> (64 - s)) : b;
}
__device__ inline K128 packed_kmer(const uint64_t *__restrict__ pk, uint64_t p, int k) {
    uint64_t A, B;
    packed_windows(pk, p, A, B);
    return k128_from_windows(A, B, k);
}

// read-start bits of positions j .. j+63 (bit 0 = position j)
template <class T>
__device__ inline uint64_t startwin64(const T &t, int j) {
    const int w = j >> 5, sh = j & 31;
    const uint64_t lo = ((uint64_t)t.sb[w + 1] << 32) | t.sb[w];
    const uint64_t hi = t.sb[w + 2];
    return sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
}

__global__ __launch_bounds__(256) void k_wpack(const char *__restrict__ bases, uint64_t n_bytes, uint64_t n_words,
                                               uint64_t *__restrict__ pk) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    uint64_t word = 0;
    const uint64_t off = w * 32;
    if (off + 32 <= n_bytes) {
        const uint4 q0 = *reinterpret_cast<const uint4 *>(bases + off);
        const uint4 q1 = *reinterpret_cast<const uint4 *>(bases + off + 16);
        const uint32_t h = (pack4(q0.x) << 24) | (pack4(q0.y) << 16) | (pack4(q0.z) << 8) | pack4(q0.w);
        const uint32_t l = (pack4(q1.x) << 24) | (pack4(q1.y) << 16) | (pack4(q1.z) << 8) | pack4(q1.w);
        word = ((uint64_t)h << 32) | l;
    } else {
        for (int b = 0; b < 32 && off + b < n_bytes; ++b)
            word |= (uint64_t)(((uint8_t)bases[off + b] >> 1) & 3u) << (62 - 2 * b);
    }
    pk[w] = word;
}

// the tile-side view of one k-mer instance, shared by every kernel that streams the reads
struct WInst {
    K128 key;
    uint32_t next;   // base after the k-mer (valid when !at_end)
    uint32_t s0;     // the k-mer sits at position 0 of its read
    uint32_t at_end; // position p + k starts another read (or is the end of the data)
};
// false: no k-mer of one read starts at tile position j
__device__ inline bool tile_inst(const TileLds &t, int j, int k, WInst &o) {
    const uint64_t sw = startwin64(t, j);
    if ((sw >> 1) & ((1ull << (k - 1)) - 1)) return false;  // a read boundary inside the k-mer
    o.s0 = (uint32_t)(sw & 1ull);
    o.at_end = (uint32_t)(sw >> k) & 1u;
    const uint64_t A = window32(t, j), B = window32(t, j + 32);
    o.key = k128_from_windows(A, B, k);
    o.next = base_after_kmer(B, k);
    return true;
}

// empty table: protocol word and key cache all ones (no k-mer has the top bits of `hi` set), spare word 0 (packed counters)
__global__ __launch_bounds__(256) void k_wtab_init(WSlot *tab, uint64_t cap) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    uint4 *q = reinterpret_cast<uint4 *>(tab + i);
    q[0] = make_uint4(~0u, ~0u, ~0u, ~0u);
    q[1] = make_uint4(~0u, ~0u, 0u, 0u);
}

// The table takes any number of slots (a multiple of 1024), not only powers of two: its memset and the compaction scan are
// proportional to it (2^31 slots x 48 B for 1.26e9 wanted = 16.8 ms of memset per build).  Home slot = high product.
__device__ inline uint64_t whome(uint64_t hv, uint64_t cap) { return __umul64hi(hv, cap); }
__device__ inline uint64_t wnext(uint64_t slot, uint64_t cap) { return slot + 1 == cap ? 0 : slot + 1; }

__global__ __launch_bounds__(256) void k_wcount(const char *__restrict__ bases, uint64_t n_bytes,
                                                const uint32_t *__restrict__ startbits, int k,
                                                const uint64_t *__restrict__ pk, WSlot *tab, uint32_t *tcnt,
                                                uint64_t cap, uint8_t *occ,
                                                unsigned long long *scalars) {
    __shared__ TileLds t;
    const uint64_t tile0 = (uint64_t)blockIdx.x * TILE;
    const uint32_t bad = load_tile(t, bases, n_bytes, startbits, tile0);
    if (bad) atomicOr(&scalars[0], 1ull);
    __syncthreads();
    uint64_t n_k = 0, n_e = 0;
    for (int j = threadIdx.x; j < TILE; j += 256) {
        const uint64_t p = tile0 + j;
        if (p >= n_bytes) break;
        WInst in;
        if (!tile_inst(t, j, k, in)) continue;
        if (in.at_end && in.s0) continue;  // read of length exactly k: contributes nothing [:126]
        // protocol word: position, indegree flag, and this instance's own out-edge (has one, which base).  The edge of
        // the instance a slot ends up pointing at is implicit -- k_wgather adds it -- so a k-mer seen once (most of them,
        // with sequencing errors) costs one atomic, the claim, and no counter update.
        const uint64_t stamp = (p << 4) | ((uint64_t)(in.s0 ^ 1u) << 3) | ((uint64_t)(in.at_end ^ 1u) << 2) |
                               (in.at_end ? 0u : in.next);
        n_k += 1;
        n_e += in.at_end ^ 1u;
        const uint64_t hv = k128_hash(in.key);
        const unsigned long long mine = ((hv & 0xFFFFull) << 48) | stamp;
        unsigned long long explicit_edge = mine;  // whose out-edge this thread adds to the counters (W_EMPTY: nobody's)
        uint64_t slot = whome(hv, cap);
        bool found = false;
        for (uint64_t probe = 0; probe < cap; ++probe) {
            WSlot *s = tab + slot;
            unsigned long long cur = __hip_atomic_load(&s->ref, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == W_EMPTY) {
                cur = atomicCAS(&s->ref, W_EMPTY, mine);
                if (cur == W_EMPTY) {
                    __hip_atomic_store(&s->lo, (unsigned long long)in.key.lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&s->hi, (unsigned long long)in.key.hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    occ[slot] = 1;  // one byte per slot: a plain store (an atomicOr per new node was 4.6e8 more global atomics)
                    explicit_edge = W_EMPTY;  // the slot points at this instance: its edge is implicit
                    found = true;
                    break;
                }
            }
            if ((cur >> 48) == (mine >> 48)) {
                const unsigned long long clo = __hip_atomic_load(&s->lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long chi = __hip_atomic_load(&s->hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((clo == in.key.lo && chi == in.key.hi) ||
                    k128_eq(packed_kmer(pk, (cur & W_STAMP_MASK) >> 4, k), in.key)) {
                    // the smaller value is the earlier instance.  Most instances are not the first of their k-mer
                    // (30x coverage) and skip the atomic: the slot only ever decreases.  An instance that takes the
                    // slot over adds the edge of the one it displaced (each value is displaced exactly once).
                    if (mine < cur) {
                        const unsigned long long old = atomicMin(&s->ref, mine);
                        if (old > mine) explicit_edge = old;
                    }
                    found = true;
                    break;
                }
            }
            slot = wnext(slot, cap);
        }
        if (!found) { atomicOr(&scalars[0], 2ull); continue; }  // table full
        if (explicit_edge != W_EMPTY && (explicit_edge & 4ull)) {
            const uint32_t nx = (uint32_t)(explicit_edge & 3ull);
            if (tcnt) {
                atomicAdd(&tcnt[slot * 4 + nx], 1u);
            } else {  // four 16-bit counters in the slot's spare word: same sector as the probe, no second random access
                const unsigned long long old = atomicAdd(&tab[slot].pad, 1ull << (16 * nx));
                if (((old >> (16 * nx)) & 0xFFFFull) == 0xFFFFull) atomicOr(&scalars[0], 32ull);  // rebuild with 32-bit counters
            }
        }
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

// occupied slots -> node arrays (table order); the slot keeps its fingerprint and takes the node id
__global__ __launch_bounds__(256) void k_wgather(WSlot *tab, const uint32_t *__restrict__ tcnt,
                                                 const uint8_t *occ, const uint32_t *word_rank, uint64_t n_words,
                                                 uint64_t *keys_lo, uint64_t *keys_hi,
                                                 uint64_t *stamps, uint32_t *cnt, uint8_t *flags, int implicit_edge) {
    // one thread per SLOT: neighbouring lanes read neighbouring sectors and write neighbouring nodes (one thread per
    // 32 slots walked them alone, every access a sector of its own: 41 ms for 4.6e8 nodes).  word_rank[w] = occupied
    // slots before slot 32 w; the slot count is a multiple of 1024, so a wave is inside the table as a whole.
    const uint64_t slot = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t w = slot >> 5;
    if (w >= n_words) return;
    const bool used = occ[slot] != 0;
    const uint64_t wave_mask = __ballot(used);
    if (!used) return;
    const uint32_t lane = threadIdx.x & 63, word = (uint32_t)(wave_mask >> (lane & 32)), bit = lane & 31;
    const uint32_t node = word_rank[w] + __popc(word & ((1u << bit) - 1u));
    const WSlot sl = tab[slot];  // the key cache is complete: the count kernel has finished
    uint64_t st = sl.ref & W_STAMP_MASK;
    uint4 c = (tcnt || !implicit_edge)
                  ? reinterpret_cast<const uint4 *>(tcnt)[slot]
                  : make_uint4((uint32_t)(sl.pad & 0xFFFF), (uint32_t)((sl.pad >> 16) & 0xFFFF),
                               (uint32_t)((sl.pad >> 32) & 0xFFFF), (uint32_t)(sl.pad >> 48));
    if (implicit_edge) {  // k_wcount's protocol word: position << 4 | indegree flag << 3 | has edge << 2 | base
        if (st & 4) {
            const uint32_t nx = (uint32_t)(st & 3);
            c.x += nx == 0; c.y += nx == 1; c.z += nx == 2; c.w += nx == 3;
        }
        st = ((st >> 4) << 1) | ((st >> 3) & 1);
    }
    keys_lo[node] = sl.lo;
    keys_hi[node] = sl.hi;
    stamps[node] = st;
    reinterpret_cast<uint4 *>(cnt)[node] = c;
    flags[node] = (uint8_t)(st & 1);
    tab[slot].ref = (sl.ref & ~W_STAMP_MASK) | node;
}

__device__ inline uint32_t wtab_find(const WSlot *__restrict__ tab, uint64_t cap, K128 key) {
    uint64_t slot = whome(k128_hash(key), cap);
    for (uint64_t probe = 0; probe < cap; ++probe) {
        const WSlot sl = tab[slot];  // one sector: reference word and key
        if (sl.ref == W_EMPTY) return NO_NODE;
        if (sl.lo == key.lo && sl.hi == key.hi) return (uint32_t)(sl.ref & W_STAMP_MASK);
        slot = wnext(slot, cap);
    }
    return NO_NODE;
}

__global__ __launch_bounds__(256) void k_wsucc(const WSlot *__restrict__ tab, uint64_t cap, int k,
                                               uint64_t n_nodes, const uint64_t *__restrict__ keys_lo,
                                               const uint64_t *__restrict__ keys_hi, const uint32_t *__restrict__ cnt,
                                               uint32_t *succ, uint8_t *order, uint8_t *deg) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const K128 key{keys_hi[i], keys_lo[i]};
    const uint4 c4 = reinterpret_cast<const uint4 *>(cnt)[i];
    const uint32_t c[4] = {c4.x, c4.y, c4.z, c4.w};
    uint32_t s[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
        s[b] = c[b] ? wtab_find(tab, cap, k128_append(key, (uint32_t)b, k)) : NO_NODE;
    reinterpret_cast<uint4 *>(succ)[i] = make_uint4(s[0], s[1], s[2], s[3]);
    deg[i] = (uint8_t)((c[0] != 0) + (c[1] != 0) + (c[2] != 0) + (c[3] != 0));
    uint32_t code[4] = {0, 1, 3, 2};  // ascii order A, C, G, T as codes; rank by (count desc, ascii asc)
#pragma unroll
    for (int a = 1; a < 4; ++a) {
#pragma unroll
        for (int b = a; b > 0; --b) {
            if (c[code[b]] > c[code[b - 1]]) { uint32_t tmp = code[b]; code[b] = code[b - 1]; code[b - 1] = tmp; }
        }
    }
    order[i] = (uint8_t)(code[0] | (code[1] << 2) | (code[2] << 4) | (code[3] << 6));
}

// ---- small sets of node ids (branch nodes; nodes with two or more successors), keys by reference ----
struct WSelBranch {
    const uint8_t *flags;
    __device__ bool operator()(uint64_t i) const { return (flags[i] & DBG_F_BRANCH) != 0; }
};
struct WSelMulti {
    const uint32_t *cnt;
    __device__ bool operator()(uint64_t i) const {
        const uint4 c = reinterpret_cast<const uint4 *>(cnt)[i];
        return (c.x != 0) + (c.y != 0) + (c.z != 0) + (c.w != 0) >= 2;
    }
};
template <class Sel>
__global__ __launch_bounds__(256) void k_wset_insert(uint64_t n_nodes, Sel sel, const uint64_t *__restrict__ keys_lo,
                                                     const uint64_t *__restrict__ keys_hi, uint32_t *set, uint64_t cap_mask) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes || !sel(i)) return;
    uint64_t slot = k128_hash(K128{keys_hi[i], keys_lo[i]}) & cap_mask;
    for (uint64_t probe = 0; probe <= cap_mask; ++probe) {  // node keys are distinct: claim the first free slot
        if (atomicCAS(&set[slot], NO_NODE, (uint32_t)i) == NO_NODE) return;
        slot = (slot + 1) & cap_mask;
    }
}
// slot of `key` in the set, or -1
__device__ inline int64_t wset_find(const uint32_t *__restrict__ set, uint64_t cap_mask, K128 key,
                                    const uint64_t *__restrict__ keys_lo, const uint64_t *__restrict__ keys_hi) {
    uint64_t slot = k128_hash(key) & cap_mask;
    for (uint64_t probe = 0; probe <= cap_mask; ++probe) {
        const uint32_t node = set[slot];
        if (node == NO_NODE) return -1;
        if (keys_lo[node] == key.lo && keys_hi[node] == key.hi) return (int64_t)slot;
        slot = (slot + 1) & cap_mask;
    }
    return -1;
}

// reads that contain a branch k-mer as a substring (reads of length == k included) [debruijn.py:248-263]
__global__ __launch_bounds__(256) void k_wpull_reads(const char *__restrict__ bases, uint64_t n_bytes,
                                                     const uint32_t *__restrict__ startbits, int k,
                                                     const uint32_t *__restrict__ set, uint64_t cap_mask,
                                                     const uint64_t *__restrict__ keys_lo, const uint64_t *__restrict__ keys_hi,
                                                     const uint64_t *offsets, uint64_t n_reads, uint8_t *read_flags) {
    __shared__ TileLds t;
    const uint64_t tile0 = (uint64_t)blockIdx.x * TILE;
    (void)load_tile(t, bases, n_bytes, startbits, tile0);
    __syncthreads();
    for (int j = threadIdx.x; j < TILE; j += 256) {
        const uint64_t p = tile0 + j;
        if (p + k > n_bytes) break;
        WInst in;
        if (!tile_inst(t, j, k, in)) continue;
        if (wset_find(set, cap_mask, in.key, keys_lo, keys_hi) < 0) continue;
        uint64_t lo = 0, hi = n_reads;  // offsets[lo] <= p < offsets[hi]
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (offsets[mid] <= p) lo = mid; else hi = mid;
        }
        read_flags[lo] = 1;
    }
}

// first occurrence of every out-edge of the nodes in the set (Counter order, debruijn.py:159-165, :215-216)
__global__ __launch_bounds__(256) void k_wedge_first_seen(const char *__restrict__ bases, uint64_t n_bytes,
                                                          const uint32_t *__restrict__ startbits, int k,
                                                          const uint32_t *__restrict__ set, uint64_t cap_mask,
                                                          const uint64_t *__restrict__ keys_lo,
                                                          const uint64_t *__restrict__ keys_hi,
                                                          unsigned long long *estamp /* [slot * 4 + code] */) {
    __shared__ TileLds t;
    const uint64_t tile0 = (uint64_t)blockIdx.x * TILE;
    (void)load_tile(t, bases, n_bytes, startbits, tile0);
    __syncthreads();
    for (int j = threadIdx.x; j < TILE; j += 256) {
        const uint64_t p = tile0 + j;
        if (p + k >= n_bytes) break;  // needs a successor base
        WInst in;
        if (!tile_inst(t, j, k, in) || in.at_end) continue;
        const int64_t slot = wset_find(set, cap_mask, in.key, keys_lo, keys_hi);
        if (slot >= 0) atomicMin(&estamp[(uint64_t)slot * 4 + in.next], (unsigned long long)p);
    }
}

// ------------------------------------------------------------------------------------------------
// Multi-GPU build of two-word k-mers.  The unit on the wire is the k-mer INSTANCE (the super-k-mer records of
// the 64-bit path hold 50 bases at most): (lo, hi | next base << 62, rank-local stamp | has-successor << 32),
// owner = top bits of the k-mer hash.  The owner counts in a table keyed by reference into the received tuples
// (claim = tuple index, one 32-bit-wide CAS; key cache and first-occurrence stamp beside it, as in WSlot).
// Successors are NOT resolved on the shards: traversal needs the gathered graph anyway, and
// dbg_import_graph resolves them there with one table over all nodes -- no query exchange for this path.
// ------------------------------------------------------------------------------------------------
__device__ inline uint32_t ws_owner(K128 key, int shard_bits) {
    return shard_bits ? (uint32_t)(k128_hash(key) >> (64 - shard_bits)) : 0u;
}

// pass 1: instances per owner (8 counters); pass 2: the tuples, each owner's into its own contiguous range
template <bool EMIT>
__global__ __launch_bounds__(256) void k_ws_extract(const char *__restrict__ bases, uint64_t n_bytes,
                                                    const uint32_t *__restrict__ startbits, int k, int shard_bits,
                                                    unsigned long long *owner_cursor /* [8]: counts (pass 1) / write cursors */,
                                                    uint64_t *t_lo, uint64_t *t_hi, uint64_t *t_st,
                                                    unsigned long long *scalars /* [0] err [1] N_k [2] N_e */) {
    __shared__ TileLds t;
    __shared__ uint32_t cnt[8];
    __shared__ unsigned long long base[8];
    const uint64_t tile0 = (uint64_t)blockIdx.x * TILE;
    const uint32_t bad = load_tile(t, bases, n_bytes, startbits, tile0);
    if (bad && !EMIT) atomicOr(&scalars[0], 1ull);
    if (threadIdx.x < 8) cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t slot_of_round[TILE / 256];  // owner | rank << 3, 0xFFFFFFFF: no instance
    uint64_t n_k = 0, n_e = 0;
#pragma unroll
    for (int r = 0; r < TILE / 256; ++r) {
        const int j = r * 256 + threadIdx.x;
        slot_of_round[r] = 0xFFFFFFFFu;
        WInst in;
        if (tile0 + j >= n_bytes || !tile_inst(t, j, k, in) || (in.at_end && in.s0)) continue;
        const uint32_t o = ws_owner(in.key, shard_bits);
        slot_of_round[r] = o | (atomicAdd(&cnt[o], 1u) << 3);
        n_k += 1;
        n_e += in.at_end ^ 1u;
    }
    __syncthreads();
    if (threadIdx.x < 8 && cnt[threadIdx.x])
        base[threadIdx.x] = atomicAdd(&owner_cursor[threadIdx.x], (unsigned long long)cnt[threadIdx.x]);
    if (!EMIT) {
        uint64_t tot_k, tot_e;
        (void)block_exscan_256(n_k, &tot_k);
        (void)block_exscan_256(n_e, &tot_e);
        if (threadIdx.x == 0) {
            if (tot_k) atomicAdd(&scalars[1], (unsigned long long)tot_k);
            if (tot_e) atomicAdd(&scalars[2], (unsigned long long)tot_e);
        }
        return;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < TILE / 256; ++r) {
        if (slot_of_round[r] == 0xFFFFFFFFu) continue;
        const int j = r * 256 + threadIdx.x;
        WInst in;
        (void)tile_inst(t, j, k, in);
        const uint64_t dst = base[slot_of_round[r] & 7u] + (slot_of_round[r] >> 3);
        const uint64_t stamp = ((tile0 + j) << 1) | (in.s0 ^ 1u);  // rank-local, < 2^32
        t_lo[dst] = in.key.lo;
        t_hi[dst] = in.key.hi | ((uint64_t)in.next << 62);
        t_st[dst] = stamp | ((uint64_t)(in.at_end ^ 1u) << 32);
    }
}

constexpr uint64_t WS_HI_MASK = (1ull << 62) - 1;

// received tuples of one source rank -> the shard's table.  WSlot: ref = first-occurrence stamp (global, atomicMin),
// lo/hi = key cache, pad = the claim: index of the tuple that created the slot (~0 free)
__global__ __launch_bounds__(256) void k_ws_insert(const uint64_t *__restrict__ t_lo, const uint64_t *__restrict__ t_hi,
                                                   const uint64_t *__restrict__ t_st, uint64_t first, uint64_t n,
                                                   uint64_t stamp_base2, WSlot *tab, uint32_t *tcnt, uint64_t cap,
                                                   uint8_t *occ, unsigned long long *scalars) {
    const uint64_t i = first + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= first + n) return;
    const K128 key{t_hi[i] & WS_HI_MASK, t_lo[i]};
    const uint32_t next = (uint32_t)(t_hi[i] >> 62);
    const unsigned long long gstamp = stamp_base2 + (t_st[i] & 0xFFFFFFFFull);
    const bool has_succ = (t_st[i] >> 32) & 1ull;
    const unsigned long long word = (gstamp << 3) | ((unsigned long long)has_succ << 2) | (has_succ ? next : 0u);
    uint64_t slot = whome(k128_hash(key), cap);
    for (uint64_t probe = 0; probe < cap; ++probe) {
        WSlot *s = tab + slot;
        unsigned long long c = __hip_atomic_load(&s->pad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool mine_now = false;
        if (c == W_EMPTY) {
            c = atomicCAS(&s->pad, W_EMPTY, (unsigned long long)i);
            if (c == W_EMPTY) {
                __hip_atomic_store(&s->lo, (unsigned long long)key.lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&s->hi, (unsigned long long)key.hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                occ[slot] = 1;  // one byte per slot: a plain store (an atomicOr per new node was 4.6e8 more global atomics)
                mine_now = true;
            }
        }
        if (!mine_now) {
            const unsigned long long clo = __hip_atomic_load(&s->lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long chi = __hip_atomic_load(&s->hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // the cache may not be written yet: the claiming tuple itself is the authority
            mine_now = (clo == key.lo && chi == key.hi) || (t_lo[c] == key.lo && (t_hi[c] & WS_HI_MASK) == key.hi);
        }
        if (mine_now) {
            // same protocol word as k_wcount (position << 4 | indegree flag << 3 | has edge << 2 | base): the edge of the
            // instance the slot ends up pointing at is implicit (k_wgather adds it), a taker adds the displaced one's
            unsigned long long explicit_edge = word;
            if (word < __hip_atomic_load(&s->ref, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                const unsigned long long old = atomicMin(&s->ref, word);
                if (old > word) explicit_edge = old;  // W_EMPTY (first taker): bit 2 set, but nobody's edge
            }
            if (explicit_edge != W_EMPTY && (explicit_edge & 4ull)) atomicAdd(&tcnt[slot * 4 + (explicit_edge & 3ull)], 1u);
            return;
        }
        slot = wnext(slot, cap);
    }
    atomicOr(&scalars[0], 2ull);  // table full
}

// gathered graph: every node into a fresh table (keys are distinct: claim the first free slot); k_wsucc then resolves
__global__ __launch_bounds__(256) void k_wnode_insert(uint64_t n_nodes, const uint64_t *__restrict__ keys_lo,
                                                      const uint64_t *__restrict__ keys_hi, WSlot *tab, uint64_t cap) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const K128 key{keys_hi[i], keys_lo[i]};
    uint64_t slot = whome(k128_hash(key), cap);
    for (uint64_t probe = 0; probe < cap; ++probe) {
        if (atomicCAS(&tab[slot].ref, W_EMPTY, (unsigned long long)i) == W_EMPTY) {
            tab[slot].lo = key.lo;
            tab[slot].hi = key.hi;
            return;
        }
        slot = wnext(slot, cap);
    }
}

}  // namespace dbgk
