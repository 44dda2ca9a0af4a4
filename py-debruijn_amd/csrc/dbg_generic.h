// Generic-alphabet engine (included by dbg_hip.hip).  The reference is alphabet-agnostic (k-mers are str
// slices, debruijn.py:127-128) and its real inputs are peptides: 20 amino-acid letters, k = 5..10.  This
// engine takes any alphabet of up to 32 distinct bytes at 5 bits per symbol ((k+1) * 5 <= 64 -> k <= 11)
// and up to 32 successors per node.  Its inputs are small (thousands of peptides), so it is written
// for exactness, not speed: one global hash table of nodes and one of edges (global atomics), dense
// [n][32] successor arrays; pruning, tips, pull-out reads and the walks run through the same kernels as the
// DNA path (accessor GGen).
#pragma once
#include "dbg_device.h"

namespace dbgk {

constexpr int GEN_BITS = 5;
constexpr int GEN_D = 32;

struct GenNodeSlot { unsigned long long key, stamp; };                         // stamp -> node id after compaction
struct GenEdgeSlot { unsigned long long key, stamp; unsigned int count, pad; };  // key = k-mer << 5 | successor code

// read-start bits of positions p .. p+31 straight from the global bitmap (no tile staging here)
__device__ inline uint32_t gen_startwin(const uint32_t *bits, uint64_t p) {
    const uint64_t w = p >> 5;
    const uint64_t both = ((uint64_t)bits[w + 1] << 32) | bits[w];
    return (uint32_t)(both >> (p & 31));
}

__device__ inline uint64_t gen_encode(const char *bases, uint64_t p, int n, const uint8_t *lut) {
    uint64_t key = 0;
    for (int i = 0; i < n; ++i) key = (key << GEN_BITS) | lut[(uint8_t)bases[p + i]];
    return key;
}

__global__ __launch_bounds__(256) void k_g_hist(const char *bases, uint64_t n, unsigned long long *hist256) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        atomicAdd(&h[(uint8_t)bases[i]], 1u);
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist256[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

template <class SlotT>
__device__ inline uint64_t gen_insert(SlotT *tab, uint64_t mask, uint64_t key, uint32_t *occ, bool *full) {
    uint64_t slot = kmer_hash(key) & mask;
    for (uint64_t probe = 0; probe <= mask; ++probe) {
        unsigned long long cur = __hip_atomic_load(&tab[slot].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == EMPTY_KEY) {
            cur = atomicCAS(&tab[slot].key, EMPTY_KEY, (unsigned long long)key);
            if (cur == EMPTY_KEY) {
                if (occ) atomicOr(&occ[slot >> 5], 1u << (slot & 31));
                cur = key;
            }
        }
        if (cur == key) return slot;
        slot = (slot + 1) & mask;
    }
    *full = true;
    return 0;
}

// a3 + a4 for any alphabet: vertex occurrences (first-occurrence stamp) and edge occurrences (count, first seen)
__global__ __launch_bounds__(256) void k_g_insert(const char *__restrict__ bases, uint64_t n_bytes,
                                                  const uint32_t *__restrict__ startbits, int k,
                                                  const uint8_t *__restrict__ lut, GenNodeSlot *nodes, uint32_t *occ,
                                                  GenEdgeSlot *edges, uint32_t *eocc, uint64_t mask,
                                                  unsigned long long *scalars /* [0] err [1] N_k [2] N_e */) {
    const uint32_t mid_mask = (k >= 2) ? ((1u << (k - 1)) - 1u) : 0u;
    uint64_t n_k = 0, n_e = 0;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_bytes; p += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t sw = gen_startwin(startbits, p);
        if ((sw >> 1) & mid_mask) continue;       // a read boundary inside the k-mer
        const uint32_t s0 = sw & 1u, sk = (sw >> k) & 1u;
        if (sk && s0) continue;                   // read of length exactly k [debruijn.py:126]
        const uint64_t key = gen_encode(bases, p, k, lut);
        const unsigned long long stamp = (p << 1) | (s0 ^ 1u);
        bool full = false;
        const uint64_t slot = gen_insert(nodes, mask, key, occ, &full);
        if (full) { atomicOr(&scalars[0], 2ull); continue; }
        atomicMin(&nodes[slot].stamp, stamp);
        ++n_k;
        if (!sk) {
            const uint64_t ekey = (key << GEN_BITS) | lut[(uint8_t)bases[p + k]];
            const uint64_t es = gen_insert(edges, mask, ekey, eocc, &full);
            if (full) { atomicOr(&scalars[0], 2ull); continue; }
            atomicAdd(&edges[es].count, 1u);
            atomicMin(&edges[es].stamp, (unsigned long long)p);
            ++n_e;
        }
    }
    n_k = wave_sum_u64(n_k);
    n_e = wave_sum_u64(n_e);
    if ((threadIdx.x & 63) == 0) {
        if (n_k) atomicAdd(&scalars[1], (unsigned long long)n_k);
        if (n_e) atomicAdd(&scalars[2], (unsigned long long)n_e);
    }
}

__global__ __launch_bounds__(256) void k_g_init_tables(GenNodeSlot *nodes, GenEdgeSlot *edges, uint64_t cap) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    nodes[i].key = EMPTY_KEY; nodes[i].stamp = ~0ull;
    edges[i].key = EMPTY_KEY; edges[i].stamp = ~0ull; edges[i].count = 0; edges[i].pad = 0;
}

__global__ __launch_bounds__(256) void k_g_gather(GenNodeSlot *nodes, const uint32_t *occ, const uint32_t *word_rank,
                                                  uint64_t n_words, uint64_t *keys, uint64_t *stamps, uint8_t *flags) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    uint32_t bits = occ[w], node = word_rank[w];
    while (bits) {
        const int b = __ffs(bits) - 1;
        bits &= bits - 1;
        GenNodeSlot *s = nodes + (w * 32 + b);
        keys[node] = s->key;
        stamps[node] = s->stamp;
        flags[node] = (uint8_t)(s->stamp & 1);
        s->stamp = node;
        ++node;
    }
}

__device__ inline uint32_t gen_find(const GenNodeSlot *nodes, uint64_t mask, uint64_t key) {
    uint64_t slot = kmer_hash(key) & mask;
    for (uint64_t probe = 0; probe <= mask; ++probe) {
        const uint64_t cur = nodes[slot].key;
        if (cur == key) return (uint32_t)nodes[slot].stamp;
        if (cur == EMPTY_KEY) return NO_NODE;
        slot = (slot + 1) & mask;
    }
    return NO_NODE;
}

// every distinct edge -> the dense [n][32] arrays of its source node
__global__ __launch_bounds__(256) void k_g_edges(const GenEdgeSlot *edges, uint64_t cap, const GenNodeSlot *nodes,
                                                 uint64_t mask, int k, uint32_t *cnt, uint32_t *succ,
                                                 unsigned long long *estamp, unsigned long long *scalars) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap || edges[i].key == EMPTY_KEY) return;
    const uint64_t ekey = edges[i].key;
    const uint64_t kmask = (1ull << (GEN_BITS * k)) - 1;
    const uint32_t code = (uint32_t)(ekey & (GEN_D - 1));
    const uint32_t src = gen_find(nodes, mask, ekey >> GEN_BITS), dst = gen_find(nodes, mask, ekey & kmask);
    if (src == NO_NODE || dst == NO_NODE) { atomicOr(&scalars[0], 128ull); return; }
    const uint64_t o = (uint64_t)src * GEN_D + code;
    cnt[o] = edges[i].count;
    succ[o] = dst;
    estamp[o] = edges[i].stamp;
}

// per node: successor codes ranked Counter.most_common style (count desc, first seen) and by first appearance
__global__ __launch_bounds__(256) void k_g_rank(uint64_t n_nodes, const uint32_t *cnt, const unsigned long long *estamp,
                                                uint8_t *rank_mc, uint8_t *rank_fs, uint8_t *deg) {
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n_nodes) return;
    uint8_t *mc = rank_mc + x * GEN_D, *fs = rank_fs + x * GEN_D;
    int n = 0;
    for (int c = 0; c < GEN_D; ++c) {
        mc[c] = 0xFF;
        fs[c] = 0xFF;
    }
    for (int c = 0; c < GEN_D; ++c) {
        const uint32_t cc = cnt[x * GEN_D + c];
        if (!cc) continue;
        const unsigned long long st = estamp[x * GEN_D + c];
        int a = n, f = n;  // insertion into both orders
        while (a > 0) {
            const uint32_t pc = cnt[x * GEN_D + mc[a - 1]];
            const unsigned long long ps = estamp[x * GEN_D + mc[a - 1]];
            if (pc > cc || (pc == cc && ps < st)) break;
            mc[a] = mc[a - 1];
            --a;
        }
        mc[a] = (uint8_t)c;
        while (f > 0 && estamp[x * GEN_D + fs[f - 1]] > st) { fs[f] = fs[f - 1]; --f; }
        fs[f] = (uint8_t)c;
        ++n;
    }
    deg[x] = (uint8_t)n;
}

__global__ __launch_bounds__(256) void k_g_csr_fill(uint64_t n_nodes, const uint64_t *rowptr, const uint32_t *cnt,
                                                    const uint32_t *succ, uint32_t *col, uint32_t *ecnt) {
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n_nodes) return;
    uint64_t o = rowptr[x];
    for (int c = 0; c < GEN_D; ++c)
        if (cnt[x * GEN_D + c]) { col[o] = succ[x * GEN_D + c]; ecnt[o] = cnt[x * GEN_D + c]; ++o; }
}

// a5 + a6 for up to 32 successors
__global__ __launch_bounds__(256) void k_g_prune(uint64_t n_nodes, const uint32_t *cnt, const uint8_t *rank_mc,
                                                 const uint8_t *deg, double threshold, uint32_t *keepmask, uint8_t *flags,
                                                 unsigned long long *n_branch) {
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t br = 0;
    if (x < n_nodes) {
        const int d = deg[x];
        uint32_t keep = 0;
        if (d >= 1) {
            const uint8_t *mc = rank_mc + x * GEN_D;
            keep = 1u << mc[0];  // the arg-max (first in most_common order) always survives [:156-159]
            if (d > 1) {
                const double lim = (double)cnt[x * GEN_D + mc[0]] / threshold;  // [:163]
                for (int r = 1; r < d; ++r)
                    if ((double)cnt[x * GEN_D + mc[r]] >= lim) keep |= 1u << mc[r];
            }
        }
        keepmask[x] = keep;
        const bool branch = __popc(keep) > 1;
        br = branch;
        flags[x] = (uint8_t)((flags[x] & DBG_F_INDEG) | (branch ? DBG_F_BRANCH : 0));
    }
    br = wave_sum_u64(br);
    if ((threadIdx.x & 63) == 0 && br) atomicAdd(n_branch, (unsigned long long)br);
}

__global__ __launch_bounds__(256) void k_g_pull_reads(const char *__restrict__ bases, uint64_t n_bytes,
                                                      const uint32_t *__restrict__ startbits, int k,
                                                      const uint8_t *__restrict__ lut, const uint64_t *__restrict__ btab,
                                                      uint64_t cap_mask, const uint64_t *offsets, uint64_t n_reads,
                                                      uint8_t *read_flags) {
    const uint32_t mid_mask = (k >= 2) ? ((1u << (k - 1)) - 1u) : 0u;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p + k <= n_bytes; p += (uint64_t)gridDim.x * blockDim.x) {
        if ((gen_startwin(startbits, p) >> 1) & mid_mask) continue;
        const uint64_t key = gen_encode(bases, p, k, lut);
        uint64_t slot = kmer_hash(key) & cap_mask;
        bool hit = false;
        for (uint64_t probe = 0; probe <= cap_mask; ++probe) {
            const uint64_t cur = btab[slot];
            if (cur == key) { hit = true; break; }
            if (cur == EMPTY_KEY) break;
            slot = (slot + 1) & cap_mask;
        }
        if (!hit) continue;
        uint64_t lo = 0, hi = n_reads;  // offsets[lo] <= p < offsets[hi]
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (offsets[mid] <= p) lo = mid; else hi = mid;
        }
        read_flags[lo] = 1;
    }
}

// accessor of the generic layout for the shared tip / walk / jump kernels
struct GGen {
    const uint64_t *keys;     // packed k-mers (k <= 11), or nullptr: k-mers are read from the reads at the node's stamp
    const char *bases;        // ... (dbg_genref.h)
    const uint64_t *stamps;
    const uint8_t *lut;       // byte -> code
    const uint8_t *flags;
    const uint32_t *keepmask;
    const uint8_t *rank_mc;
    const uint8_t *deg;
    const uint32_t *succ;
    const uint32_t *cnt;
    const char *alpha;  // code -> byte
    int k;
    __device__ int n_ranks(uint32_t x) const { return deg[x]; }
    __device__ uint32_t code_at(uint32_t x, int r) const { return rank_mc[(uint64_t)x * GEN_D + r]; }
    __device__ bool kept(uint32_t x, uint32_t code) const { return (keepmask[x] >> code) & 1u; }
    __device__ uint32_t keep_count(uint32_t x) const { return __popc(keepmask[x]); }
    __device__ uint32_t first_kept(uint32_t x) const { return __ffs(keepmask[x]) - 1; }
    __device__ uint32_t succ_of(uint32_t x, uint32_t code) const { return succ[(uint64_t)x * GEN_D + code]; }
    __device__ uint32_t cnt_of(uint32_t x, uint32_t code) const { return cnt[(uint64_t)x * GEN_D + code]; }
    __device__ bool terminal(uint32_t x) const { return deg[x] == 0; }
    __device__ uint32_t last_code(uint32_t x) const {
        return keys ? (uint32_t)(keys[x] & (GEN_D - 1)) : (uint32_t)lut[(uint8_t)bases[(stamps[x] >> 1) + k - 1]];
    }
    __device__ char sym_char(uint32_t code) const { return alpha[code]; }
    __device__ char char_at(uint32_t x, int q) const {  // character q of node x's k-mer
        if (!keys) return bases[(stamps[x] >> 1) + q];
        return alpha[(keys[x] >> (GEN_BITS * (k - 1 - q))) & (GEN_D - 1)];
    }
    __device__ void spell(uint32_t x, char *out) const {
        if (!keys) {
            const uint64_t p = stamps[x] >> 1;
            for (int q = 0; q < k; ++q) out[q] = bases[p + q];
            return;
        }
        const uint64_t key = keys[x];
        for (int q = 0; q < k; ++q) out[q] = alpha[(key >> (GEN_BITS * (k - 1 - q))) & (GEN_D - 1)];
    }
};

}  // namespace dbgk
