// Count kernel, second generation (k <= 31, 4096-slot LDS table): k_sk_count2.
//
// Same contract as k_sk_count (dbg_sk.h): per bucket of super-k-mer records -> node keys, first-occurrence stamps, one
// flag byte per node, CSR rows (rowptr, col, ecnt), the range's directory and the list of successors that live in
// another bucket                                                                [debruijn.py:129-143, :213-222]
//
// What is different, and why (profiles/r02_*: the first kernel spends a quarter of its time probing the table a SECOND
// time for the successor of every node, in waves that run as long as their slowest lane and loop once per base):
//   * The successor of k-mer i of a record is k-mer i + 1 of the same record, and that k-mer is inserted by the
//     neighbouring lane in the same instruction stream.  Every lane takes the slot its neighbour ended up in (one DPP
//     wave shift, no LDS) and writes it into the upper half of the edge's counter word: one 32-bit LDS word per
//     (slot, base) = 16-bit count | 16-bit successor hint.
//   * Only edges whose successor is NOT in the neighbouring lane (the last k-mer of a record, the last lane of a wave,
//     a successor filtered into another hash sub-range) go on a pending list -- a private segment per wave, no atomics --
//     and only those are looked up afterwards, one lane per edge: ~300 per bucket instead of ~1500 node slots times
//     their bases.
//   * Node and edge totals of the bucket are counted while inserting, per lane (compare-and-swap winners; counter adds
//     that return 0, looked at one iteration later, when the value has long arrived) and summed once per wave, so the
//     one global atomic that reserves node ids and CSR rows goes out right after the insert phase and has the list
//     phase and the pending lookups to come back (same-address global atomics queue up chip-wide: ~1-2 us under load).
//   * One descriptor in device memory for inputs and outputs, re-read through scalar loads where a phase needs it: no
//     base pointers or 64-bit ranges live across the persistent loop; the next bucket's record range travels in vector
//     registers so that nothing waits for it at the load.
// 16-bit counters: a bucket whose records cannot add up to 65 536 instances of one edge needs no check; the others
// check the value the add returns, raise flag 512 and the host repeats the build with the 32-bit counters of k_sk_count.
#pragma once
#include "dbg_sk.h"

#ifndef DBG_SK2_LANEQUAD
#define DBG_SK2_LANEQUAD 0   // 0: four lanes per quad, one k-mer each; 1: a lane walks the four k-mers of its quad (measured: 12.9 vs 12.0 ms,
                             // and MORE instructions -- 3.95e9 vs 3.78e9 vector: what a lane saves on decoding the record it spends on the
                             // bookkeeping of four steps)
#endif
#ifndef DBG_SK2_UNROLL
#define DBG_SK2_UNROLL 8   // probe steps per unrolled body of the insert loop (4: 12.13, 8: 12.06, 16: 12.22 ms)
#endif
#ifndef DBG_SK2_PROBE
#define DBG_SK2_PROBE 0   // 0: per-lane exit, unrolled 16-fold; 1: wave-uniform probe loop with a ballot exit (measured: 14.3 vs 12.4 ms)
#endif

namespace dbgk {

struct SkCount2Args {
    SkCountOut out;
    const uint64_t *b_start, *b_cnt, *rec_w0, *rec_w1;
    const void *rec_st;
    uint64_t n_buckets;
    uint32_t split_recs;
    int k;
};
static_assert(sizeof(SkCount2Args) <= 64 * 8, "descriptor slot of the scalar block");

typedef const SkCount2Args __attribute__((address_space(4))) *SkArgs2ConstPtr;
__device__ inline SkArgs2ConstPtr fresh_args2(const SkCount2Args *p) {
    unsigned long long v = (unsigned long long)p;
    asm volatile("" : "+s"(v));
    return (SkArgs2ConstPtr)v;
}

template <class ST>
struct Cnt2Cfg {
    static constexpr int CAP = 4096;
    static constexpr int NT = 1024;
#ifdef DBG_CNT_PROF
    static constexpr int QBUF = sizeof(ST) == 8 ? 296 : 736;
#else
    static constexpr int QBUF = sizeof(ST) == 8 ? 320 : 768;   // staged records per round
#endif
    static constexpr int QS = sizeof(ST) == 8 ? 256 : 512;     // staged queries per bucket pass (more go out one by one)
    static constexpr int PEND = sizeof(ST) == 8 ? 2048 : 4096;  // edges waiting for a successor lookup
};

// the query cursor of this kernel: far from the node / edge cursor (word 4) -- atomics on one cache line queue up behind each other
constexpr int SK2_QUERY_CURSOR = 120;
constexpr uint32_t HINT_VALID = 0x8000u;   // hint half of a counter word: the successor's slot is in bits 11:0
constexpr uint32_t HINT_QUERY = 0x4000u;   // ... the successor is not in this table: a query (counted once)
constexpr int PEND_SEG = 256;              // pending edges a wave can list per bucket pass (PEND = 16 waves x PEND_SEG)

template <class ST>
struct Cnt2Lds {
    static constexpr int CAP = Cnt2Cfg<ST>::CAP;
    static constexpr int QBUF = Cnt2Cfg<ST>::QBUF;
    unsigned long long keys[CAP];
    uint32_t ch[CAP * 4];      // per (slot, base): count (low 16) | hint (high 16; 0 = not known)
    ST stamp[CAP];
    uint16_t list[CAP];        // insert: quad list; afterwards: local node index -> slot
    uint16_t eoff[CAP];        // insert: dedupe set (uint32[CAP / 2]); afterwards: local node index -> first CSR edge
    uint16_t pend[Cnt2Cfg<ST>::PEND];  // per wave a segment: slot * 4 + base of the edges whose successor slot no lane handed over
    unsigned long long q_key[QBUF];    // staged w0
    unsigned long long q_meta[QBUF];   // staged w1 (bucket-hash field = multiplicity)
    ST st_stage[QBUF];
    unsigned long long qs_key[Cnt2Cfg<ST>::QS];  // queries of the pass being written; they leave after the NEXT pass's first barrier
    uint32_t qs_col[Cnt2Cfg<ST>::QS];
    unsigned long long dir_mask[CAP / 64];
    uint16_t dir_base[CAP / 64];
    uint16_t pend_cnt[16];
    uint32_t dummy[64];        // where the lanes without a hint to write store theirs (no branch in the insert loop)
    uint32_t stk_mask[CNT_STACK], stk_val[CNT_STACK];
    uint32_t overflow, n_new /* nodes | edges << 16, counted by the insert */, n_list /* the same from the list phase */, n_q, n_q2, n_q3, n_flat, fail, pend_over;
    unsigned long long gbase, ebase, qbase, ri;
#ifdef DBG_CNT_PROF
    unsigned long long prof[64];
#endif
};

// lane i receives the value of lane i + 1 of its wave; lane 63 receives `last` (DPP wave shift: one VALU move, no LDS)
__device__ inline uint32_t from_next_lane(uint32_t v, uint32_t last) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)last, (int)v, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
}

// sum over the wave, valid in lane 63: four shifts inside the rows of 16 lanes, then the row totals are handed on
// (row_bcast:15 -> rows 1 and 3, row_bcast:31 -> rows 2 and 3); VALU only
__device__ inline uint32_t wave_sum_dpp(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /* row_shr:1 */, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112 /* row_shr:2 */, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114 /* row_shr:4 */, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118 /* row_shr:8 */, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142 /* row_bcast:15 */, 0xA, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143 /* row_bcast:31 */, 0xC, 0xF, false);
    return v;
}

template <class ST>
__global__ __launch_bounds__(1024) void k_sk_count2(const SkCount2Args *__restrict__ argp) {
    constexpr int CAP = Cnt2Cfg<ST>::CAP, NT = Cnt2Cfg<ST>::NT, QBUF = Cnt2Cfg<ST>::QBUF, PEND = Cnt2Cfg<ST>::PEND, QS = Cnt2Cfg<ST>::QS;
    constexpr int NPT = CAP / NT;
    constexpr int PSEG = PEND / 16;
    constexpr uint32_t STAGE = QBUF;
    extern __shared__ __attribute__((aligned(16))) unsigned char cnt_raw[];
    using LdsT = Cnt2Lds<ST>;
    LdsT &s = *reinterpret_cast<LdsT *>(cnt_raw);
    static_assert(sizeof(LdsT) <= 160 * 1024, "LDS");
    static_assert(QBUF * 5 <= CAP && STAGE <= (uint32_t)NT && CAP / 2 >= 2 * QBUF, "quad list / dedupe scratch");
    static_assert(PSEG <= PEND_SEG && NT == 1024, "one pending segment per wave");
    constexpr uint64_t BH_FIELD = ((1ull << SK_BUCKET_BITS) - 1) << 6;
    uint16_t *flat = s.list;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = fresh_args2(argp)->k;
    const uint32_t n_buckets = (uint32_t)fresh_args2(argp)->n_buckets;
    const uint32_t split_recs = fresh_args2(argp)->split_recs;

#ifdef DBG_CNT_PROF
    unsigned long long clast_ = clock64(), csub_ = clast_;
    if (threadIdx.x < 64) s.prof[threadIdx.x] = 0;
    __syncthreads();
#endif
    bool clean = false;
    uint64_t pf_w0 = 0, pf_w1 = 0;
    ST pf_st = 0;
    uint64_t nx_beg = 0;
    uint32_t nx_n = 0;
    // The record range of the bucket after the next one travels in vector registers: a wave-uniform value that the
    // compiler knows to be uniform is loaded into scalar registers, and the wait for a scalar register is placed at the
    // load -- one exposed memory latency per bucket.  The index goes through an opaque vector move, the range comes
    // back per lane and becomes scalar (readfirstlane) a whole bucket later, when it has long arrived.
    uint64_t r2_beg_v = 0, r2_n_v = 0;
    auto load_range = [&](uint32_t bucket) {
        uint32_t vb = bucket;
        asm volatile("" : "+v"(vb));
        r2_beg_v = 0;
        r2_n_v = 0;
        if (vb < n_buckets) {
            const auto &a = *fresh_args2(argp);
            r2_beg_v = a.b_start[vb];
            r2_n_v = a.b_cnt[vb];
        }
    };
    auto prefetch = [&](uint32_t bucket) {
        nx_beg = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(r2_beg_v >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)r2_beg_v);
        nx_n = __builtin_amdgcn_readfirstlane((uint32_t)min(r2_n_v, (uint64_t)0xFFFFFFF0u));
        if (threadIdx.x < min(nx_n, STAGE)) {
            const auto &a = *fresh_args2(argp);
            pf_w0 = a.rec_w0[nx_beg + threadIdx.x];
            pf_w1 = a.rec_w1[nx_beg + threadIdx.x];
            pf_st = reinterpret_cast<const ST *>(a.rec_st)[nx_beg + threadIdx.x];
        }
        load_range(bucket + gridDim.x);
    };
    // The queries of a pass (successors that are not in its table) are staged in LDS while its nodes are written and go
    // out after the first barrier of the NEXT pass: no barrier and no phase of their own.
    uint32_t fl_nq = 0, fl_unst = 0;  // uniform: queries waiting; per thread: those that found no room in the staging, bit (u * 4 + base)
    uint64_t fl_gbase = 0;
    auto flush_queries = [&]() {  // all threads, after a barrier that follows the pass's write phase
        if (!fl_nq || s.fail) return;  // (a failed pass -- no room in the query list -- ends the kernel at the next check)
        const auto &oq = fresh_args2(argp)->out;
        const uint64_t qbase = s.qbase;
        if (s.n_q2 != fl_nq && threadIdx.x == 0) { atomicOr(&oq.scalars[0], 2048ull); s.fail = 1; }  // internal: query totals disagree
        for (uint32_t i = threadIdx.x; i < min(fl_nq, (uint32_t)QS); i += NT) {
            oq.q_key[qbase + i] = s.qs_key[i];
            oq.q_col[qbase + i] = s.qs_col[i];
        }
        if (fl_nq > (uint32_t)QS) {  // rare: queries beyond the staging go out one by one, key and row re-read from the node arrays
            const uint64_t kmask = (1ull << (2 * k)) - 1;
#pragma unroll
            for (int u = 0; u < NPT; ++u) {
                const uint32_t mine = (fl_unst >> (u * 4)) & 15u;
                uint32_t qi = wave_alloc_n<4>(&s.n_q3, (uint32_t)__popc(mine));
                if (mine) {
                    const uint64_t node = fl_gbase + threadIdx.x + u * NT;
                    const uint64_t key = oq.keys[node];
                    const uint32_t pres = (uint32_t)oq.flags[node] >> 1;
                    const uint32_t e0 = oq.rowptr[node];
                    uint32_t m = mine;
                    while (m) {
                        const uint32_t b = __ffs(m) - 1;
                        m &= m - 1;
                        oq.q_key[qbase + QS + qi] = ((key << 2) | (uint64_t)b) & kmask;
                        oq.q_col[qbase + QS + qi] = e0 + __popc(pres & ((1u << b) - 1u));
                        ++qi;
                    }
                }
            }
        }
        fl_nq = 0;
    };
    if (threadIdx.x == 0) s.fail = 0;
    load_range(blockIdx.x);
    prefetch(blockIdx.x);

    for (uint32_t bucket = blockIdx.x; bucket < n_buckets; bucket += gridDim.x) {
        const uint64_t r_beg = nx_beg;
        const uint32_t r_n = nx_n;
        if (r_n == 0) { prefetch(bucket + gridDim.x); continue; }
        bool have_prefetch = true;
        // can one edge of this bucket be seen 65 536 times?  (a record holds at most 19 k-mers)
        const bool check16 = r_n >= 65536u / 20u;
        uint32_t stk_n = 1;
        bool root = true, failed = false;
        if (split_recs && r_n > split_recs) {
            uint32_t parts = 2;
            while (parts < 16 && (uint64_t)parts * split_recs < r_n) parts <<= 1;
            if (threadIdx.x < parts) { s.stk_mask[threadIdx.x] = parts - 1; s.stk_val[threadIdx.x] = threadIdx.x; }
            stk_n = parts;
            root = false;
            __syncthreads();
        }
        while (stk_n) {
            uint32_t cur_mask = 0, cur_val = 0;
            --stk_n;
            if (!root) { cur_mask = s.stk_mask[stk_n]; cur_val = s.stk_val[stk_n]; }
            root = false;
            __syncthreads();  // the previous pass (or bucket) is done with the staging arrays and the table
            CNT_TICK(0);
            flush_queries();
            if (!clean) {
                for (int i = threadIdx.x; i < CAP; i += NT) {
                    s.keys[i] = EMPTY_KEY;
                    s.stamp[i] = (ST)~(ST)0;
                    reinterpret_cast<uint4 *>(s.ch)[i] = make_uint4(0, 0, 0, 0);
                }
            }
            clean = false;
            if (threadIdx.x == 0) { s.overflow = 0; s.n_new = 0; s.n_list = 0; s.n_q = 0; s.pend_over = 0; }
            uint32_t pcur = 0;  // wave-uniform: entries in this wave's pending segment
            uint32_t my_new = 0;  // this lane's new nodes (low half) and new edges (high half)
            // ---- insert
            for (uint32_t c0 = 0; c0 < r_n; c0 += STAGE) {
                const uint32_t n_st = min(STAGE, r_n - c0);
                if (c0) __syncthreads();
                if (threadIdx.x == 0) s.n_flat = 0;
                if (c0 == 0 && have_prefetch) {
                    if (threadIdx.x < n_st) { s.q_key[threadIdx.x] = pf_w0; s.q_meta[threadIdx.x] = pf_w1 & ~BH_FIELD; s.st_stage[threadIdx.x] = pf_st; }
                } else {
                    const auto &a = *fresh_args2(argp);
                    for (uint32_t r = threadIdx.x; r < n_st; r += NT) {
                        s.q_key[r] = a.rec_w0[r_beg + c0 + r];
                        s.q_meta[r] = a.rec_w1[r_beg + c0 + r] & ~BH_FIELD;
                        s.st_stage[r] = reinterpret_cast<const ST *>(a.rec_st)[r_beg + c0 + r];
                    }
                }
                uint32_t *dd_tab = reinterpret_cast<uint32_t *>(s.eoff);
                constexpr uint32_t DD_SLOTS = CAP / 2;
                for (uint32_t i = threadIdx.x; i < DD_SLOTS; i += NT) dd_tab[i] = 0xFFFFFFFFu;
                __syncthreads();
                CNT_TICK(1);
                if (s.overflow) break;
                {   // identical records collapse to one representative with a multiplicity and the smallest stamp
                    const uint32_t r = threadIdx.x;
                    uint32_t nquad = 0;
                    if (r < n_st) {
                        const unsigned long long w0 = s.q_key[r], w1 = s.q_meta[r];
                        uint32_t hslot = fmix32(fold32(w0) ^ (fold32(w1) * 0x9E3779B1u)) & (DD_SLOTS - 1);
                        uint32_t rep = r;
                        for (uint32_t probe = 0; probe < DD_SLOTS; ++probe) {
                            uint32_t cur = dd_tab[hslot];
                            if (cur == 0xFFFFFFFFu) {
                                cur = atomicCAS(&dd_tab[hslot], 0xFFFFFFFFu, r);
                                if (cur == 0xFFFFFFFFu) break;
                            }
                            if (s.q_key[cur] == w0 && ((s.q_meta[cur] ^ w1) & ~BH_FIELD) == 0) { rep = cur; break; }
                            hslot = (hslot + 1) & (DD_SLOTS - 1);
                        }
                        atomicAdd(reinterpret_cast<uint32_t *>(&s.q_meta[rep]), 1u << 6);
                        if (rep != r) atomicMin(&s.st_stage[rep], s.st_stage[r]);
                        else nquad = ((uint32_t)((w1 >> 1) & 31) + 4) >> 2;
                    }
                    const uint32_t base = wave_alloc_n<5>(&s.n_flat, nquad);
                    for (uint32_t q = 0; q < nquad; ++q) flat[base + q] = (uint16_t)((r << 3) | q);
                }
                __syncthreads();
                CNT_TICK(3);
                const uint32_t n_flat = s.n_flat;
                uint32_t p_old = 1, p_mult = 0;  // what the previous counter add returned (looked at one step later)
#if DBG_SK2_LANEQUAD
                // One LANE per quad: it reads the record once and walks the quad's (up to) four k-mers one after the other, so the
                // slot of k-mer i + 1 is in the same lane a step later, and only the quad's last k-mer asks the next lane (the
                // record's next quad).  A wave-step still probes 64 k-mers, but record decode, list and staging reads are paid
                // once per four k-mers: ~190 instead of ~300 instructions per 64 k-mers.
                for (uint32_t f0 = 0; f0 < n_flat; f0 += NT) {
                    const uint32_t f = f0 + threadIdx.x;
                    const bool act = f < n_flat;
                    const uint32_t e = act ? flat[f] : 0u;
                    const uint32_t r = e >> 3;
                    const int i0 = (int)(e & 7) * 4;
                    const uint64_t w0 = s.q_key[r], w1 = s.q_meta[r];
                    const int len = (int)((w1 >> 1) & 31) + 1;
                    const ST st0 = s.st_stage[r];
                    const uint32_t mult = (uint32_t)(w1 >> 6) & ((1u << SK_BUCKET_BITS) - 1u);
                    const uint64_t hi = w1 & (~0ull << SK_META_BITS);
                    const int nk = act ? min(4, len - i0) : 0;
                    bool prev_edge = false;      // the previous k-mer of this lane has a successor and waits for its slot
                    uint32_t prev_sb = 0, first_slot = 0xFFFFu;
                    unsigned long long pq = 0;   // this lane's pending edges, 16 bits each
                    uint32_t pc = 0;
#pragma unroll 1
                    for (int j = 0; j < 4; ++j) {
                        const int i = i0 + j;
                        bool actj = j < nk;
                        const uint64_t win = rec_window(w0, hi, i & 31);
                        const uint64_t kmer = win >> (64 - 2 * k);
                        if (cur_mask) actj = actj && (sub_hash(kmer) & cur_mask) == cur_val;
                        const uint32_t b = (uint32_t)(win >> (62 - 2 * k)) & 3u;
                        uint32_t slot = slot_of<CAP>(kmer);
                        bool ok = false;
                        uint32_t won = 0;
                        if (actj) {
#pragma unroll DBG_SK2_UNROLL
                            for (int probe = 0; probe < CNT_PROBE_LIMIT; ++probe) {
                                unsigned long long cur = s.keys[slot];
                                if (cur == EMPTY_KEY) {
                                    cur = atomicCAS(&s.keys[slot], EMPTY_KEY, (unsigned long long)kmer);
                                    if (cur == EMPTY_KEY) { cur = kmer; won = 1; }
                                }
                                if (cur == kmer) { ok = true; break; }
                                slot = (slot + 1) & (CAP - 1);
                            }
                            if (!ok) s.overflow = 1;
                        }
                        my_new += won + (((p_old & 0xFFFFu) == 0) ? 0x10000u : 0u);
                        if (check16 && (p_old & 0xFFFFu) + p_mult > 0xFFFFu) s.fail = 2;
                        const bool good = actj && ok;
                        {   // the previous k-mer's edge ends here: hint, or (successor filtered out / record over) pending
                            const bool in_lane = prev_edge && good;
                            uint16_t *hp = in_lane ? reinterpret_cast<uint16_t *>(&s.ch[prev_sb]) + 1 : reinterpret_cast<uint16_t *>(&s.dummy[lane]);
                            *hp = (uint16_t)(HINT_VALID | slot);
                            if (prev_edge && !good) { pq |= (unsigned long long)prev_sb << (16 * pc); ++pc; }
                        }
                        if (j == 0) first_slot = good ? slot : 0xFFFFu;
                        const bool edge = good && ((i < len - 1) || (w1 & 1));
                        p_old = 1;
                        p_mult = 0;
                        if (edge) { p_old = atomicAdd(&s.ch[slot * 4 + b], mult); p_mult = mult; }
                        if (good) atomicMin(&s.stamp[slot], i ? (ST)((st0 | (ST)1) + (ST)(2 * i)) : st0);
                        prev_edge = edge;
                        prev_sb = slot * 4 + b;
                    }
                    {   // the quad's fourth k-mer: its successor is the first k-mer of the record's next quad -- the next lane's
                        const uint32_t nxt = from_next_lane(first_slot, 0xFFFFu);
                        const bool in_wave = prev_edge && (i0 + 3 < len - 1) && nxt != 0xFFFFu;
                        uint16_t *hp = in_wave ? reinterpret_cast<uint16_t *>(&s.ch[prev_sb]) + 1 : reinterpret_cast<uint16_t *>(&s.dummy[lane]);
                        *hp = (uint16_t)(HINT_VALID | (nxt & (CAP - 1)));
                        if (prev_edge && !in_wave) { pq |= (unsigned long long)prev_sb << (16 * pc); ++pc; }
                    }
                    // this lane's pending edges (at most four) go on the wave's own segment: no atomic, nothing to wait for
                    {
                        uint32_t at = pcur, total = 0;
#pragma unroll
                        for (int q = 1; q <= 4; ++q) {
                            const unsigned long long mq = __ballot(pc >= (uint32_t)q);
                            at += lanes_below(mq);
                            total += (uint32_t)__popcll(mq);
                        }
                        for (uint32_t q = 0; q < pc; ++q)
                            if (at + q < (uint32_t)PSEG) s.pend[wave * PSEG + at + q] = (uint16_t)(pq >> (16 * q));
                        pcur += total;
                    }
                }
#else
                // every lane stays in the loop (predicated): the wave hands successor slots from lane to lane
                for (uint32_t f0 = 0; f0 < n_flat; f0 += NT / 4) {
                    if (f0 + wave * 16 >= n_flat) break;  // wave-uniform: nothing left for this wave (the list is dealt out in lane order)
                    const uint32_t f = f0 + (threadIdx.x >> 2);
                    bool act = f < n_flat;
                    const uint32_t e = act ? flat[f] : 0u;
                    const uint32_t r = e >> 3;
                    const int i = (int)((e & 7) * 4 + (threadIdx.x & 3));
                    const uint64_t w0 = s.q_key[r], w1 = s.q_meta[r];
                    const int len = (int)((w1 >> 1) & 31) + 1;
                    act = act && i < len;
                    const ST st0 = s.st_stage[r];
                    const uint32_t mult = (uint32_t)(w1 >> 6) & ((1u << SK_BUCKET_BITS) - 1u);
                    const uint64_t hi = w1 & (~0ull << SK_META_BITS);
                    const uint64_t win = rec_window(w0, hi, i & 31);
                    const uint64_t kmer = win >> (64 - 2 * k);
                    if (cur_mask) act = act && (sub_hash(kmer) & cur_mask) == cur_val;
                    const bool has_succ = (i < len - 1) || (w1 & 1);
                    const uint32_t b = (uint32_t)(win >> (62 - 2 * k)) & 3u;
                    const ST stamp = i ? (ST)((st0 | (ST)1) + (ST)(2 * i)) : st0;
                    uint32_t slot = slot_of<CAP>(kmer);
                    bool ok = false;
                    uint32_t won = 0;
#if DBG_SK2_PROBE == 1
                    {   // The probe as a wave-uniform loop: the exit test is a ballot (one scalar branch per step), the lanes that are
                        // done stay in step under a predicate.  A loop whose lanes leave one by one is unrolled by the compiler
                        // into nested regions, and every iteration pays three scalar instructions per level to close them.
                        bool todo = act;
                        for (int probe = 0; __ballot(todo) != 0 && probe < CNT_PROBE_LIMIT; ++probe) {
                            unsigned long long cur = s.keys[slot];
                            if (todo && cur == EMPTY_KEY) {
                                cur = atomicCAS(&s.keys[slot], EMPTY_KEY, (unsigned long long)kmer);
                                if (cur == EMPTY_KEY) { cur = kmer; won = 1; }
                            }
                            const bool hit = todo && cur == kmer;
                            ok = ok || hit;
                            todo = todo && !hit;
                            slot = todo ? ((slot + 1) & (CAP - 1)) : slot;
                        }
                        if (todo) s.overflow = 1;
                    }
#else
                    if (act) {
#pragma unroll DBG_SK2_UNROLL
                        for (int probe = 0; probe < CNT_PROBE_LIMIT; ++probe) {
                            unsigned long long cur = s.keys[slot];
                            if (cur == EMPTY_KEY) {
                                cur = atomicCAS(&s.keys[slot], EMPTY_KEY, (unsigned long long)kmer);
                                if (cur == EMPTY_KEY) { cur = kmer; won = 1; }
                            }
                            if (cur == kmer) { ok = true; break; }
                            slot = (slot + 1) & (CAP - 1);
                        }
                        if (!ok) s.overflow = 1;
                    }
#endif
                    // the previous iteration's counter add: a return of 0 was the first instance of that edge
                    my_new += won + (((p_old & 0xFFFFu) == 0) ? 0x10000u : 0u);
                    if (check16 && (p_old & 0xFFFFu) + p_mult > 0xFFFFu) s.fail = 2;
                    const bool good = act && ok;
                    // slot of the next k-mer of the record: the neighbouring lane's (quads of one record are consecutive)
                    const uint32_t nxt = from_next_lane(good ? slot : 0xFFFFu, 0xFFFFu);
                    const bool edge = good && has_succ;
                    const bool in_wave = edge && (i < len - 1) && nxt != 0xFFFFu;
                    p_old = 1;
                    p_mult = 0;
                    if (edge) { p_old = atomicAdd(&s.ch[slot * 4 + b], mult); p_mult = mult; }
                    if (good) atomicMin(&s.stamp[slot], stamp);
                    // the hint: every instance of an edge that knows the successor's slot writes it (the same value each
                    // time); the others write to a dummy word -- an address select instead of a branch
                    {
                        uint16_t *hp = in_wave ? reinterpret_cast<uint16_t *>(&s.ch[slot * 4 + b]) + 1
                                               : reinterpret_cast<uint16_t *>(&s.dummy[lane]);
                        *hp = (uint16_t)(HINT_VALID | (nxt & (CAP - 1)));
                    }
                    // the edges that got no hint here go on the wave's own pending segment (no atomic, nothing to wait for)
                    const bool need_pend = edge && !in_wave;
                    const unsigned long long m_pend = __ballot(need_pend);
                    if (m_pend) {
                        const uint32_t pb = pcur + lanes_below(m_pend);
                        if (need_pend && pb < (uint32_t)PSEG) s.pend[wave * PSEG + pb] = (uint16_t)(slot * 4 + b);
                        pcur += (uint32_t)__popcll(m_pend);
                    }
                }
#endif
                my_new += ((p_old & 0xFFFFu) == 0) ? 0x10000u : 0u;  // the last iteration's add
                if (check16 && (p_old & 0xFFFFu) + p_mult > 0xFFFFu) s.fail = 2;
            }
            {   // this wave's share of the bucket's node and edge totals
                const uint32_t tot = wave_sum_dpp(my_new);
                if (lane == 63 && tot) atomicAdd(&s.n_new, tot);
            }
            if (lane == 0) {
                s.pend_cnt[wave] = (uint16_t)min(pcur, (uint32_t)PSEG);
                if (pcur > (uint32_t)PSEG) s.pend_over = 1;
            }
            CNT_TICK(4);
            __syncthreads();
            CNT_TICK(5);
            if (threadIdx.x == 0) { s.n_q2 = 0; s.n_q3 = 0; }  // used from the write phase on; the previous pass's queries are out
            if (have_prefetch) {
                have_prefetch = false;
                prefetch(bucket + gridDim.x);
            }
            CNT_TICK(13);
            if (s.fail) { failed = true; break; }
            if (s.overflow) {  // split this hash sub-range in two and retry (nothing was written out)
                const uint32_t bit = cur_mask + 1;
                if (stk_n + 2 > CNT_STACK || bit >= (1u << 20)) {
                    if (threadIdx.x == 0) atomicOr(&fresh_args2(argp)->out.scalars[0], 8ull);
                    failed = true;
                    break;
                }
                if (threadIdx.x == 0) {
                    s.stk_mask[stk_n] = cur_mask | bit; s.stk_val[stk_n] = cur_val;
                    s.stk_mask[stk_n + 1] = cur_mask | bit; s.stk_val[stk_n + 1] = cur_val | bit;
                }
                stk_n += 2;
                __syncthreads();
                continue;
            }
            // ---- the insert counted the bucket's nodes and edges: reserve node ids and CSR rows now; the pending lookups and
            //      the list phase run while the answer is on its way
            const uint32_t n_local = s.n_new & 0xFFFFu, n_edges_local = s.n_new >> 16;
            unsigned long long got = 0;
            if (threadIdx.x == 0)
                got = atomicAdd(&fresh_args2(argp)->out.scalars[4], (unsigned long long)n_local | ((unsigned long long)n_edges_local << 32));
            CNT_TICK(14);
            // ---- successor lookups of the pending edges: every wave its own segment (all (slot, base) words of the table
            //      if any segment overflowed).  A miss marks the edge as a query; the lane that sets the mark counts it.
            {
                const uint64_t kmask = (1ull << (2 * k)) - 1;
                const bool over = s.pend_over != 0;  // some wave's segment overflowed
                const uint32_t n_items = over ? (uint32_t)CAP * 4u / 16u : (uint32_t)s.pend_cnt[wave];  // per wave
                for (uint32_t e0 = 0; e0 < n_items; e0 += 64) {
                    const uint32_t e = e0 + lane;
                    bool miss = false;
                    if (e < n_items) {
                        const uint32_t sb = over ? (uint32_t)wave * ((uint32_t)CAP * 4u / 16u) + e : (uint32_t)s.pend[wave * PSEG + e];
                        const uint32_t c = s.ch[sb];
                        if ((c & 0xFFFFu) && (c >> 16) == 0) {
                            const unsigned long long key = s.keys[sb >> 2];
                            const int fnd = lds_find<CAP>(s.keys, ((key << 2) | (uint64_t)(sb & 3u)) & kmask);
                            if (fnd >= 0) reinterpret_cast<uint16_t *>(&s.ch[sb])[1] = (uint16_t)(HINT_VALID | (uint32_t)fnd);
                            else miss = !(atomicOr(&s.ch[sb], HINT_QUERY << 16) & (HINT_QUERY << 16));
                        }
                    }
                    const unsigned long long mm = __ballot(miss);
                    if (mm && lane == 0) atomicAdd(&s.n_q, (uint32_t)__popcll(mm));
                }
            }
            CNT_TICK(6);
            // ---- dense list of occupied slots + CSR edge offsets
            {
                unsigned long long kk[NPT];
                uint4 cc[NPT];
#pragma unroll
                for (int t = 0; t < NPT; ++t) {
                    const int i = threadIdx.x + t * NT;
                    kk[t] = s.keys[i];
                    cc[t] = reinterpret_cast<const uint4 *>(s.ch)[i];
                }
                unsigned long long mask[NPT];
                uint32_t below[NPT], eexc[NPT], nn[NPT], ne[NPT], tot = 0;
#pragma unroll
                for (int t = 0; t < NPT; ++t) {
                    const bool occ = kk[t] != EMPTY_KEY;
                    const uint32_t deg = occ ? ((cc[t].x & 0xFFFFu) != 0) + ((cc[t].y & 0xFFFFu) != 0) + ((cc[t].z & 0xFFFFu) != 0) + ((cc[t].w & 0xFFFFu) != 0) : 0u;
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
                if (tot && lane == 0) base = atomicAdd(&s.n_list, tot);
                base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
                for (int t = 0; t < NPT; ++t) {
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
            CNT_TICK(7);
            if (threadIdx.x == 0) {
                const auto &orr = fresh_args2(argp)->out;
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
                    rg.bucket = bucket; rg.mask = cur_mask; rg.val = cur_val; rg.node_cnt = n_local; rg.node_base = base;
                    rg.next = 0; rg.pad = 0;
                    if (cur_mask) {
                        rg.next = orr.ranges[bucket].next;
                        orr.ranges[bucket].next = (uint32_t)ri;
                    }
                    orr.ranges[ri] = rg;
                }
            }
            CNT_TICK(9);
            __syncthreads();
            CNT_TICK(10);
            const uint32_t nq = s.n_q;
            unsigned long long qgot = 0;
            if (threadIdx.x == 64 && nq) qgot = atomicAdd(&fresh_args2(argp)->out.scalars[SK2_QUERY_CURSOR], (unsigned long long)nq);
            if (s.n_list != s.n_new && threadIdx.x == 0) { atomicOr(&fresh_args2(argp)->out.scalars[0], 2048ull); s.fail = 1; }  // internal: the insert's totals are off
            if (s.fail) { failed = true; break; }  // written before the barrier above
            const uint64_t gbase = s.gbase, ebase = s.ebase;
            uint32_t unst = 0;  // queries of this thread that found no room in the staging: bit (u * 4 + base)
            {
                const auto &ow = fresh_args2(argp)->out;
                const uint64_t kmask = (1ull << (2 * k)) - 1;
                if (threadIdx.x < CAP / 64) {  // the range's directory (k_succ_resolve)
                    SkDirEnt de;
                    de.mask = s.dir_mask[threadIdx.x];
                    de.base = (uint32_t)(gbase + s.dir_base[threadIdx.x]);
                    de.pad = s.ri < ow.n_buckets ? 1u : 0u;
                    const uint64_t di = s.ri < ow.n_buckets ? s.ri - ow.own_lo : ow.own_cnt + (s.ri - ow.n_buckets);
                    ow.dirs[di * (CAP / 64) + threadIdx.x] = de;
                }
#pragma unroll
                for (int u = 0; u < NPT; ++u) {
                    if ((uint32_t)(u * NT) >= n_local) break;  // uniform
                    // The thread index goes through an opaque move: &s.list[li] and &s.eoff[li] are then computed here.  Hoisted
                    // out of the bucket loop as induction pointers they lived across all phases and, at the 128-register limit of
                    // 1024 threads, were spilled to scratch: two memory round trips per round of this loop to reload two additions.
                    uint32_t tw = threadIdx.x;
                    asm volatile("" : "+v"(tw));
                    const uint32_t li = tw + u * NT;
                    uint32_t qmask = 0, e0 = 0, pres = 0, slot_i = 0;
                    unsigned long long key = 0;
                    if (li < n_local) {
                        slot_i = s.list[li];
                        key = s.keys[slot_i];
                        const uint4 c4 = reinterpret_cast<const uint4 *>(s.ch)[slot_i];
                        const ST stamp = s.stamp[slot_i];
                        s.keys[slot_i] = EMPTY_KEY;
                        s.stamp[slot_i] = (ST)~(ST)0;
                        reinterpret_cast<uint4 *>(s.ch)[slot_i] = make_uint4(0, 0, 0, 0);
                        const uint64_t node = gbase + li;
                        pres = ((c4.x & 0xFFFFu) != 0) | (((c4.y & 0xFFFFu) != 0) << 1) | (((c4.z & 0xFFFFu) != 0) << 2) | (((c4.w & 0xFFFFu) != 0) << 3);
                        ow.keys[node] = key;
                        reinterpret_cast<ST *>(ow.stamps)[node] = stamp;
                        ow.flags[node] = (uint8_t)((uint32_t)(stamp & 1) | (pres << 1));
                        e0 = (uint32_t)(ebase + s.eoff[li]);
                        ow.rowptr[node] = e0;
                        // the node's edges by rank: the wave loops as often as its largest out-degree (mostly twice)
                        uint32_t todo = pres, e = e0;
                        while (todo) {
                            const uint32_t b = __ffs(todo) - 1;
                            todo &= todo - 1;
                            const uint32_t cw = b == 0 ? c4.x : b == 1 ? c4.y : b == 2 ? c4.z : c4.w;
                            const uint32_t hnt = cw >> 16;
                            uint32_t v = NO_NODE;
                            if (hnt & HINT_VALID) {
                                const uint32_t fs = hnt & (CAP - 1);
                                const uint32_t ix = (uint32_t)s.dir_base[fs >> 6] + (uint32_t)__popcll(s.dir_mask[fs >> 6] & ((1ull << (fs & 63)) - 1ull));
                                v = (uint32_t)(gbase + ix) | ow.id_tag;
                            } else {
                                qmask |= 1u << b;
                            }
                            ow.col[e] = v;
                            ow.ecnt[e] = cw & 0xFFFFu;
                            ++e;
                        }
                    }
                    // successors that are not in this table: staged as queries (key, CSR column)
                    uint32_t qi = wave_alloc_n<4>(&s.n_q2, (uint32_t)__popc(qmask));
                    while (qmask) {
                        const uint32_t b = __ffs(qmask) - 1;
                        qmask &= qmask - 1;
                        if (qi < (uint32_t)QS) {
                            s.qs_key[qi] = ((key << 2) | (uint64_t)b) & kmask;
                            s.qs_col[qi] = e0 + __popc(pres & ((1u << b) - 1u));
                        } else {
                            unst |= 1u << (u * 4 + b);
                        }
                        ++qi;
                    }
                }
            }
            CNT_TICK(11);
            if (threadIdx.x == 64 && nq) {
                s.qbase = qgot;
                const auto &oq = fresh_args2(argp)->out;
                if (qgot + nq > oq.q_cap || qgot + nq > 0xFFFFFFF0ull) { atomicOr(&oq.scalars[0], 64ull); s.fail = 1; }
            }
            fl_nq = nq;        // the staged queries leave after the next barrier every thread passes (flush_queries)
            fl_unst = unst;
            fl_gbase = gbase;
            clean = true;
        }
        if (failed) {
            if (s.fail == 2 && threadIdx.x == 0) atomicOr(&fresh_args2(argp)->out.scalars[0], 512ull);  // 16-bit counter overflow
            return;
        }
    }
    __syncthreads();
    flush_queries();  // the last pass's
#ifdef DBG_CNT_PROF
    if (threadIdx.x == 0) {
        for (int i = 0; i < 31; ++i) atomicAdd(&g_cnt_prof[i], s.prof[i]);
        atomicAdd(&g_cnt_prof[31], 1ull);
    }
#endif
}

}  // namespace dbgk
