// Pieces that k_wsk_count2 (dbg_wsk2.h) and k_sk_count3 (dbg_sk3.h) share word for word: both keep four 16-bit counters and one
// successor hint per table slot, list the occupied slots with ballots, take one packed global reservation per pass and describe
// the pass to the resolver (range record + directory).  The LDS structs of the two kernels name these fields alike.
#pragma once
#include "dbg_sk.h"

namespace dbgk {

// Dense list of occupied slots + CSR edge offsets of one pass: a wave takes its CAP / NT blocks of 64 slots together (reads back
// to back, ONE packed LDS atomic for all of them: nodes in the low half, edges in the high half of s.n_local), writes the
// directory of its blocks (mask + base) and, per occupied slot, list[] (local node index -> slot) and eoff[] (-> first CSR
// edge).  `occupied(key word)`: the table's emptiness test on the word in keys[] (the high word of a two-word key).
template <int CAP, int NT, class LDS>
__device__ inline void cnt_dense_list(LDS &s, const unsigned long long *keys) {
    constexpr int NB = CAP / NT;
    const uint32_t lane = threadIdx.x & 63;
    unsigned long long kk[NB];
    uint2 cc[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) {
        const int i = threadIdx.x + t * NT;
        kk[t] = keys[i];
        cc[t] = reinterpret_cast<const uint2 *>(s.cnt2)[i];
    }
    unsigned long long mask[NB];
    uint32_t below[NB], eexc[NB], nn[NB], ne[NB], tot = 0;
#pragma unroll
    for (int t = 0; t < NB; ++t) {
        const bool occ = kk[t] != EMPTY_KEY;
        const uint32_t deg = occ ? ((cc[t].x & 0xFFFFu) != 0) + ((cc[t].x >> 16) != 0) + ((cc[t].y & 0xFFFFu) != 0) + ((cc[t].y >> 16) != 0) : 0u;
        mask[t] = __ballot(occ);
        below[t] = lanes_below(mask[t]);
        eexc[t] = 0;
        ne[t] = 0;
#pragma unroll
        for (int j = 1; j <= 4; ++j) {
            const unsigned long long mj = __ballot(deg >= (uint32_t)j);
            eexc[t] += lanes_below(mj);
            ne[t] += (uint32_t)__popcll(mj);
        }
        nn[t] = (uint32_t)__popcll(mask[t]);
        tot += nn[t] | (ne[t] << 16);
    }
    uint32_t base = 0;
    if (tot && lane == 0) base = atomicAdd(&s.n_local, tot);
    base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
    for (int t = 0; t < NB; ++t) {
        const int i = threadIdx.x + t * NT;
        if (lane == 0) { s.dir_mask[i >> 6] = mask[t]; s.dir_base[i >> 6] = (uint16_t)base; }
        if (kk[t] != EMPTY_KEY) {
            const uint32_t li = (base & 0xFFFFu) + below[t];
            s.list[li] = (uint16_t)i;
            s.eoff[li] = (uint16_t)((base >> 16) + eexc[t]);
        }
        base += nn[t] | (ne[t] << 16);
    }
}

// Thread 0, once the packed reservation `got` (nodes low 32 | edges high 32) is back: capacity checks, the range record of this
// pass (a bucket counted whole: ranges[bucket]; a hash sub-range: a new record chained to the bucket's) and gbase / ebase / ri
// in LDS for everybody after the next barrier.  OUT: the output descriptor (SkCountOut / WSkCountOut: same field names).
template <class LDS, class OUT>
__device__ inline void cnt_take_reservation(LDS &s, const OUT &orr, unsigned long long got, uint32_t n_new, uint64_t bucket,
                                            uint32_t cur_mask, uint32_t cur_val) {
    const uint32_t n_local = n_new & 0xFFFFu, n_edges_local = n_new >> 16;
    const unsigned long long base = got & 0xFFFFFFFFull, eb = got >> 32;
    s.gbase = base;
    s.ebase = eb;
    if (base + n_local > orr.node_cap || base + n_local > 0xFFFFFFF0ull) { atomicOr(&orr.scalars[0], 16ull); s.fail = 1; }
    if (eb + n_edges_local > orr.edge_cap || eb + n_edges_local > 0xFFFFFFF0ull) { atomicOr(&orr.scalars[0], 16ull); s.fail = 1; }
    uint64_t ri = bucket;
    if (cur_mask) {
        ri = orr.n_buckets + atomicAdd(&orr.scalars[6], 1ull);
        if (ri >= orr.range_cap || ri >= 0xFFFFFFF0ull) { atomicOr(&orr.scalars[0], 32ull); s.fail = 1; }
    }
    s.ri = ri;
    if (!s.fail) {
        SkRange rg;
        rg.bucket = (uint32_t)bucket; rg.mask = cur_mask; rg.val = cur_val; rg.node_cnt = n_local; rg.node_base = base;
        rg.next = 0; rg.pad = 0;
        if (cur_mask) {
            rg.next = orr.ranges[bucket].next;
            orr.ranges[bucket].next = (uint32_t)ri;
        }
        orr.ranges[ri] = rg;
    }
}

// The directory of the pass (threads 0 .. CAP / 64 - 1; k_succ_resolve / k_wsucc_resolve replay the probing on it).  The thread
// index goes through an opaque move: &s.dir_mask[tid] is then computed here -- hoisted out of the bucket loop it lived across all
// phases and, at the 128-register limit of 1024 threads, was spilled to scratch.
template <int CAP, class LDS, class OUT>
__device__ inline void cnt_write_directory(const LDS &s, const OUT &ow, uint64_t gbase) {
    if (threadIdx.x < CAP / 64) {
        uint32_t td = threadIdx.x;
        asm volatile("" : "+v"(td));
        SkDirEnt de;
        de.mask = s.dir_mask[td];
        de.base = (uint32_t)(gbase + s.dir_base[td]);
        de.pad = s.ri < ow.n_buckets ? 1u : 0u;
        const uint64_t di = s.ri < ow.n_buckets ? s.ri - ow.own_lo : ow.own_cnt + (s.ri - ow.n_buckets);
        ow.dirs[di * (CAP / 64) + td] = de;
    }
}

// node id of the successor a hint or a lookup found in slot f of this pass's table
template <class LDS>
__device__ inline uint32_t cnt_local_index(const LDS &s, uint32_t f) {
    return (uint32_t)s.dir_base[f >> 6] + (uint32_t)__popcll(s.dir_mask[f >> 6] & ((1ull << (f & 63)) - 1ull));
}

}  // namespace dbgk
