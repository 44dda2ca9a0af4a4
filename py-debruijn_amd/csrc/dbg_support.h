// Read-support scores of contigs (reference: findSupportReadScore, IV_sortOutputs.py:10-15).
//
// score(contig) = sum of score[read] over the reads that occur in the contig as a substring, added in read order.
// The reference tests every (read, contig) pair with `read in contig`: O(reads x contigs x length).  Here the reads
// are indexed by their first `a` bytes (a = shortest read, at most 7): every contig position looks its a-mer up,
// verifies the few reads that start with it, and emits (contig, read) hits; the hits are sorted, duplicates (a read
// that occurs twice in one contig counts once) drop out, and one thread per contig adds its reads' scores in
// ascending read order -- the reference's order, so floating-point sums equal the reference's bit for bit.
#pragma once
#include "dbg_device.h"

namespace dbgk {

constexpr uint32_t SUP_NONE = 0xFFFFFFFFu;

__device__ inline uint64_t sup_pack(const char *p, int a) {  // a <= 7: the top byte stays 0, never EMPTY_KEY
    uint64_t v = 0;
    for (int i = 0; i < a; ++i) v = (v << 8) | (uint8_t)p[i];
    return v;
}

__global__ __launch_bounds__(256) void k_sup_insert(const char *__restrict__ rchars, const uint64_t *__restrict__ roff,
                                                    uint64_t n_reads, int a, unsigned long long *keys, uint32_t *head,
                                                    uint32_t *next, uint64_t mask) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const uint64_t len = roff[r + 1] - roff[r];
    if (len == 0) return;  // occurs in every contig: k_sup_empty
    const uint64_t key = sup_pack(rchars + roff[r], a);
    uint64_t slot = mix64(key) & mask;
    for (;;) {
        unsigned long long cur = keys[slot];
        if (cur == EMPTY_KEY) {
            cur = atomicCAS(&keys[slot], EMPTY_KEY, (unsigned long long)key);
            if (cur == EMPTY_KEY) cur = key;
        }
        if (cur == key) break;
        slot = (slot + 1) & mask;
    }
    next[r] = atomicExch(&head[slot], (uint32_t)r);
}

// one thread per contig character position
__global__ __launch_bounds__(256) void k_sup_scan(const char *__restrict__ cchars, const uint64_t *__restrict__ coff,
                                                  uint64_t n_contigs, uint64_t n_chars, int a,
                                                  const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ head,
                                                  const uint32_t *__restrict__ next, uint64_t mask,
                                                  const char *__restrict__ rchars, const uint64_t *__restrict__ roff,
                                                  unsigned long long *hits, uint64_t hit_cap, unsigned long long *counter) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_chars) return;
    uint64_t lo = 0, hi = n_contigs;  // coff[lo] <= p < coff[hi]
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (coff[mid] <= p) lo = mid; else hi = mid;
    }
    const uint64_t c = lo, cend = coff[c + 1];
    if (p + (uint64_t)a > cend) return;
    const uint64_t key = sup_pack(cchars + p, a);
    uint64_t slot = mix64(key) & mask;
    for (;;) {
        const unsigned long long cur = keys[slot];
        if (cur == EMPTY_KEY) return;
        if (cur == key) break;
        slot = (slot + 1) & mask;
    }
    for (uint32_t r = head[slot]; r != SUP_NONE; r = next[r]) {
        const uint64_t rb = roff[r], rl = roff[r + 1] - rb;
        if (p + rl > cend) continue;
        bool same = true;
        for (uint64_t i = (uint64_t)a; i < rl && same; ++i) same = rchars[rb + i] == cchars[p + i];
        if (!same) continue;
        const unsigned long long at = atomicAdd(counter, 1ull);
        if (at < hit_cap) hits[at] = ((unsigned long long)c << 32) | r;
    }
}

// empty reads occur in every contig (Python: '' in s is True)
__global__ __launch_bounds__(256) void k_sup_empty(const uint32_t *__restrict__ empties, uint64_t n_empty, uint64_t n_contigs,
                                                   unsigned long long *hits, uint64_t hit_cap, unsigned long long *counter) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_empty * n_contigs) return;
    const uint64_t c = i / n_empty, e = i % n_empty;
    const unsigned long long at = atomicAdd(counter, 1ull);
    if (at < hit_cap) hits[at] = ((unsigned long long)c << 32) | empties[e];
}

// hits sorted ascending: contig-major, read index inside -- the order the reference adds in
__global__ __launch_bounds__(256) void k_sup_sum(const unsigned long long *__restrict__ hits, uint64_t n_hits, uint64_t n_contigs,
                                                 const double *__restrict__ scores, const uint8_t *__restrict__ is_float,
                                                 double *out, uint32_t *out_float_hits) {
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_contigs) return;
    uint64_t lo = 0, hi = n_hits;  // first hit with contig >= c
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if ((hits[mid] >> 32) < c) lo = mid + 1; else hi = mid;
    }
    double acc = 0.0;
    uint32_t nf = 0, prev = SUP_NONE;
    for (uint64_t i = lo; i < n_hits && (hits[i] >> 32) == c; ++i) {
        const uint32_t r = (uint32_t)hits[i];
        if (r == prev) continue;  // found at another position of the same contig
        prev = r;
        acc += scores[r];
        nf += is_float ? is_float[r] : 1u;
    }
    out[c] = acc;
    if (out_float_hits) out_float_hits[c] = nf;
}

}  // namespace dbgk
