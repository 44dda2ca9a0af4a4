// Generic-alphabet engine for k-mers that do not fit one packed word: any alphabet of up to 32 distinct bytes,
// 12 <= k <= 63 (the reference is alphabet-agnostic, its inputs are peptides).  Same structure as
// dbg_generic.h (one global table of nodes, one of edges, dense [n][32] successor arrays, shared downstream
// kernels), but the tables are keyed BY REFERENCE like dbg_wide.h: a slot holds (16-bit fingerprint | position of
// the earliest instance seen so far) and its key is "the k (or k+1) bytes at that position of the reads" --
// one 64-bit CAS claims, one 64-bit atomicMin keeps the first occurrence, compares read immutable bytes.
// Written for exactness on small inputs (thousands of peptides), not for speed.
#pragma once
#include "dbg_device.h"
#include "dbg_generic.h"

namespace dbgk {

constexpr uint64_t GR_REF_MASK = (1ull << 48) - 1;
constexpr unsigned long long GR_EMPTY = ~0ull;

__device__ inline uint64_t gr_hash(const char *__restrict__ bases, uint64_t p, int n) {
    uint64_t hv = 0xCBF29CE484222325ull;  // FNV-1a over the bytes, then a finaliser
    for (int i = 0; i < n; ++i) hv = (hv ^ (uint8_t)bases[p + i]) * 0x100000001B3ull;
    return mix64(hv);
}
__device__ inline bool gr_bytes_eq(const char *__restrict__ bases, uint64_t p, uint64_t q, int n) {
    for (int i = 0; i < n; ++i)
        if (bases[p + i] != bases[q + i]) return false;
    return true;
}
// read-start bits of positions p .. p+63 from the global bitmap (bit 0 = position p)
__device__ inline uint64_t gr_startwin64(const uint32_t *bits, uint64_t p) {
    const uint64_t w = p >> 5;
    const int sh = (int)(p & 31);
    const uint64_t lo = ((uint64_t)bits[w + 1] << 32) | bits[w];
    return sh ? (lo >> sh) | ((uint64_t)bits[w + 2] << (64 - sh)) : lo;
}

// slot of the n-byte window at p, inserting it if new; `mine` = fingerprint << 48 | reference (the reference is a
// stamp (position << 1 | flag) for nodes: REF_SHIFT 1, a plain position for edges: REF_SHIFT 0).  -1: table full.
template <int REF_SHIFT>
__device__ inline int64_t gr_insert(unsigned long long *tab, uint64_t mask, int hash_shift, uint64_t hv,
                                    unsigned long long mine, const char *__restrict__ bases, uint64_t p, int n, uint32_t *occ) {
    uint64_t slot = hv >> hash_shift;
    for (uint64_t probe = 0; probe <= mask; ++probe) {
        unsigned long long cur = __hip_atomic_load(&tab[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == GR_EMPTY) {
            cur = atomicCAS(&tab[slot], GR_EMPTY, mine);
            if (cur == GR_EMPTY) {
                atomicOr(&occ[slot >> 5], 1u << (slot & 31));
                return (int64_t)slot;
            }
        }
        if ((cur >> 48) == (mine >> 48) && gr_bytes_eq(bases, p, (cur & GR_REF_MASK) >> REF_SHIFT, n)) {
            if (mine < cur) atomicMin(&tab[slot], mine);  // same fingerprint: the smaller value is the earlier instance
            return (int64_t)slot;
        }
        slot = (slot + 1) & mask;
    }
    return -1;
}

// a3 + a4: vertex occurrences (first-occurrence stamp) and edge occurrences (count, first seen)
__global__ __launch_bounds__(256) void k_gr_insert(const char *__restrict__ bases, uint64_t n_bytes,
                                                   const uint32_t *__restrict__ startbits, int k, unsigned long long *ntab,
                                                   uint32_t *nocc, unsigned long long *etab, uint32_t *eocc, uint32_t *ecount,
                                                   uint64_t mask, int hash_shift,
                                                   unsigned long long *scalars /* [0] err [1] N_k [2] N_e */) {
    uint64_t n_k = 0, n_e = 0;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_bytes; p += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t sw = gr_startwin64(startbits, p);
        if ((sw >> 1) & ((1ull << (k - 1)) - 1)) continue;  // a read boundary inside the k-mer
        const uint32_t s0 = (uint32_t)(sw & 1ull), sk = (uint32_t)(sw >> k) & 1u;
        if (sk && s0) continue;                              // read of length exactly k [debruijn.py:126]
        const uint64_t hv = gr_hash(bases, p, k);
        const unsigned long long stamp = (p << 1) | (s0 ^ 1u);
        if (gr_insert<1>(ntab, mask, hash_shift, hv, ((hv & 0xFFFFull) << 48) | stamp, bases, p, k, nocc) < 0) {
            atomicOr(&scalars[0], 2ull);
            continue;
        }
        ++n_k;
        if (!sk) {
            const uint64_t he = gr_hash(bases, p, k + 1);
            const int64_t es = gr_insert<0>(etab, mask, hash_shift, he, ((he & 0xFFFFull) << 48) | p, bases, p, k + 1, eocc);
            if (es < 0) { atomicOr(&scalars[0], 2ull); continue; }
            atomicAdd(&ecount[es], 1u);
            ++n_e;
        }
    }
    uint64_t tot_k, tot_e;
    (void)block_exscan_256(n_k, &tot_k);
    (void)block_exscan_256(n_e, &tot_e);
    if (threadIdx.x == 0) {
        if (tot_k) atomicAdd(&scalars[1], (unsigned long long)tot_k);
        if (tot_e) atomicAdd(&scalars[2], (unsigned long long)tot_e);
    }
}

// occupied node slots -> node arrays (table order); the slot keeps its fingerprint and takes the node id
__global__ __launch_bounds__(256) void k_gr_gather(unsigned long long *ntab, const uint32_t *occ, const uint32_t *word_rank,
                                                   uint64_t n_words, uint64_t *stamps, uint8_t *flags) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    uint32_t bits = occ[w], node = word_rank[w];
    while (bits) {
        const int b = __ffs(bits) - 1;
        bits &= bits - 1;
        const uint64_t slot = w * 32 + b;
        const unsigned long long cur = ntab[slot];
        const uint64_t st = cur & GR_REF_MASK;
        stamps[node] = st;
        flags[node] = (uint8_t)(st & 1);
        ntab[slot] = (cur & ~GR_REF_MASK) | node;
        ++node;
    }
}

// node id of the k-byte window at p (after k_gr_gather), or NO_NODE
__device__ inline uint32_t gr_find(const unsigned long long *__restrict__ ntab, uint64_t mask, int hash_shift,
                                   const char *__restrict__ bases, const uint64_t *__restrict__ stamps, uint64_t p, int k) {
    const uint64_t hv = gr_hash(bases, p, k);
    uint64_t slot = hv >> hash_shift;
    for (uint64_t probe = 0; probe <= mask; ++probe) {
        const unsigned long long cur = ntab[slot];
        if (cur == GR_EMPTY) return NO_NODE;
        if ((cur >> 48) == (hv & 0xFFFFull)) {
            const uint32_t node = (uint32_t)(cur & GR_REF_MASK);
            if (gr_bytes_eq(bases, p, stamps[node] >> 1, k)) return node;
        }
        slot = (slot + 1) & mask;
    }
    return NO_NODE;
}

// every distinct edge -> the dense [n][32] arrays of its source node
__global__ __launch_bounds__(256) void k_gr_edges(const unsigned long long *__restrict__ etab, const uint32_t *__restrict__ ecount,
                                                  uint64_t cap, const unsigned long long *__restrict__ ntab, uint64_t mask,
                                                  int hash_shift, const char *__restrict__ bases,
                                                  const uint64_t *__restrict__ stamps, const uint8_t *__restrict__ lut, int k,
                                                  uint32_t *cnt, uint32_t *succ, unsigned long long *estamp,
                                                  unsigned long long *scalars) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap || etab[i] == GR_EMPTY) return;
    const uint64_t q = etab[i] & GR_REF_MASK;  // first instance of the (k+1)-mer
    const uint32_t code = lut[(uint8_t)bases[q + k]];
    const uint32_t src = gr_find(ntab, mask, hash_shift, bases, stamps, q, k);
    const uint32_t dst = gr_find(ntab, mask, hash_shift, bases, stamps, q + 1, k);
    if (src == NO_NODE || dst == NO_NODE || code >= (uint32_t)GEN_D) { atomicOr(&scalars[0], 128ull); return; }
    const uint64_t o = (uint64_t)src * GEN_D + code;
    cnt[o] = ecount[i];
    succ[o] = dst;
    estamp[o] = q;
}

// set of branch nodes (ids), keys by reference; and the reads that contain one of them [debruijn.py:274-278]
__global__ __launch_bounds__(256) void k_gr_set_insert(uint64_t n_nodes, const uint8_t *__restrict__ flags,
                                                       const char *__restrict__ bases, const uint64_t *__restrict__ stamps, int k,
                                                       uint32_t *set, uint64_t cap_mask) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes || !(flags[i] & DBG_F_BRANCH)) return;
    uint64_t slot = gr_hash(bases, stamps[i] >> 1, k) & cap_mask;
    for (uint64_t probe = 0; probe <= cap_mask; ++probe) {  // node keys are distinct: claim the first free slot
        if (atomicCAS(&set[slot], NO_NODE, (uint32_t)i) == NO_NODE) return;
        slot = (slot + 1) & cap_mask;
    }
}

__global__ __launch_bounds__(256) void k_gr_pull_reads(const char *__restrict__ bases, uint64_t n_bytes,
                                                       const uint32_t *__restrict__ startbits, int k,
                                                       const uint32_t *__restrict__ set, uint64_t cap_mask,
                                                       const uint64_t *__restrict__ stamps, const uint64_t *offsets,
                                                       uint64_t n_reads, uint8_t *read_flags) {
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p + k <= n_bytes; p += (uint64_t)gridDim.x * blockDim.x) {
        if ((gr_startwin64(startbits, p) >> 1) & ((1ull << (k - 1)) - 1)) continue;
        uint64_t slot = gr_hash(bases, p, k) & cap_mask;
        bool hit = false;
        for (uint64_t probe = 0; probe <= cap_mask; ++probe) {
            const uint32_t node = set[slot];
            if (node == NO_NODE) break;
            if (gr_bytes_eq(bases, p, stamps[node] >> 1, k)) { hit = true; break; }
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

}  // namespace dbgk
