// One-word count kernel for 64-bit stamps: the layout of k_wsk_count2 (dbg_wsk2.h) on the records of k_sk_count2 (dbg_sk2.h).
//
// k_sk_count2 keeps a 32-bit word per (slot, base) -- 16-bit count | 16-bit successor hint -- and with 64-bit stamps (sharded
// builds: global positions; reads of 2 GiB and more) the 160 KB of LDS then leave room for 320 staged records where a bucket
// holds ~530: most buckets pay the dedupe / quad list / insert barriers twice, and the kernel loses to k_sk_count (15.4 vs
// 13.4 ms on a 10 M-read shard).  Here a slot has four 16-bit counters (8 bytes) and ONE 16-bit hint (valid | base << 12 |
// slot of key + base): 24 KB less, 768 staged records beside the 64-bit stamps.  What the hint does not cover -- the last k-mer
// of a record, lane 63, a second successor base (3 % of the nodes), a successor in another hash sub-range -- is noticed in the node
// write, listed per wave and looked up right after it, one lane per edge (see dbg_wsk2.h for the scheme and its reasons:
// totals counted by the insert, reservation right after the insert barrier, queries leaving after the next pass's first
// barrier, keys cleared at the top of the next pass).
// Same descriptor, outputs and flags as k_sk_count2 (512: a 16-bit counter overflowed -> the host repeats with k_sk_count;
// 2048: the insert and the list phase disagree).  Instantiated for both stamp widths; the host uses it for 64-bit stamps.
#pragma once
#include "dbg_cnt_common.h"
#include "dbg_sk2.h"

namespace dbgk {

template <class ST>
struct Cnt3Cfg {
    static constexpr int CAP = 4096, NT = 1024;
#ifdef DBG_CNT_PROF
    static constexpr int STAGE = 704;
#else
    static constexpr int STAGE = 768;     // records staged per round
#endif
    static constexpr int QBUF = 256;      // cross-bucket queries staged per pass; more go out wave by wave
    static constexpr int DSEG = 64;       // deferred edges a wave can list per pass; more are looked up where they are found
    static constexpr int FLW = QBUF / 64; // waves that write the staged queries out
};

template <class ST>
struct Cnt3Lds {
    static constexpr int CAP = Cnt3Cfg<ST>::CAP, STAGE = Cnt3Cfg<ST>::STAGE, QBUF = Cnt3Cfg<ST>::QBUF;
    unsigned long long keys[CAP];
    uint32_t cnt2[CAP * 2];            // four 16-bit successor counters per slot: codes 0, 1 in dword 0; 2, 3 in dword 1
    ST stamp[CAP];
    uint16_t list[CAP];                // insert: quad list; afterwards: local node index -> slot
    uint16_t eoff[CAP];                // insert: dedupe set (uint32[CAP / 2]); afterwards: local node index -> first CSR edge
    uint16_t hint[CAP + 64];           // WHINT_VALID | base << 12 | slot of (key + base); [CAP + lane]: dummy words
    unsigned long long q_key[STAGE];   // staged w0
    unsigned long long q_meta[STAGE];  // staged w1 (bucket-hash field = multiplicity)
    ST st_stage[STAGE];
    uint16_t dseg[(Cnt3Cfg<ST>::NT / 64) * 2 * Cnt3Cfg<ST>::DSEG];  // per wave: (slot << 2 | base, CSR offset) of the edges without a hint
    unsigned long long qs_key[QBUF];   // queries of the pass being written; they leave after the NEXT pass's first barrier
    uint32_t qs_col[QBUF];
    unsigned long long dir_mask[CAP / 64];
    uint16_t dir_base[CAP / 64];
    uint32_t stk_mask[CNT_STACK], stk_val[CNT_STACK];
    uint32_t overflow, n_local, n_new, n_q, fail, n_flat;
    unsigned long long gbase, ebase, ri;
#ifdef DBG_CNT_PROF
    unsigned long long prof[64];
#endif
};
static_assert(sizeof(Cnt3Lds<uint64_t>) <= 160 * 1024 && sizeof(Cnt3Lds<uint32_t>) <= 160 * 1024, "LDS of k_sk_count3");

__device__ inline void cnt3_load(const uint32_t *cnt2, uint32_t slot, uint32_t c[4]) {
    const uint2 v = reinterpret_cast<const uint2 *>(cnt2)[slot];
    c[0] = v.x & 0xFFFFu; c[1] = v.x >> 16; c[2] = v.y & 0xFFFFu; c[3] = v.y >> 16;
}

template <class ST>
__global__ __launch_bounds__(1024) void k_sk_count3(const SkCount2Args *__restrict__ argp) {
    constexpr int CAP = Cnt3Cfg<ST>::CAP, NT = Cnt3Cfg<ST>::NT, QBUF = Cnt3Cfg<ST>::QBUF, DSEG = Cnt3Cfg<ST>::DSEG, FLW = Cnt3Cfg<ST>::FLW;
    constexpr int NPT = CAP / NT;
    constexpr uint32_t STAGE = Cnt3Cfg<ST>::STAGE;
    constexpr uint32_t HINT_OK = 0x8000u;
    static_assert(STAGE * 5 <= CAP && 2 * STAGE <= CAP / 2 && STAGE <= NT, "the quad list lives in list[], the dedupe set in eoff[]");
    extern __shared__ __attribute__((aligned(16))) unsigned char cnt_raw[];
    Cnt3Lds<ST> &s = *reinterpret_cast<Cnt3Lds<ST> *>(cnt_raw);
    constexpr uint64_t BH_FIELD = ((1ull << SK_BUCKET_BITS) - 1) << 6;
    uint16_t *flat = s.list;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = fresh_args2(argp)->k;
    const uint64_t n_buckets = fresh_args2(argp)->n_buckets;
    const uint32_t split_recs = fresh_args2(argp)->split_recs;
    const uint64_t kmask = (1ull << (2 * k)) - 1;
    bool clean = false;
    uint32_t prev_n = 0;
    if (threadIdx.x == 0) { s.fail = 0; s.n_q = 0; }
#ifdef DBG_CNT_PROF
    unsigned long long clast_ = clock64();
    if (threadIdx.x < 64) s.prof[threadIdx.x] = 0;
    __syncthreads();
#endif
    uint64_t pf_w0 = 0, pf_w1 = 0;
    ST pf_st = 0;
    uint64_t nx_beg = 0;
    uint32_t nx_n = 0;
    uint64_t r2_beg_v = 0, r2_n_v = 0;  // the range after the next one, in vector registers (k_sk_count2, finding 2)
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
    // staged queries of the previous pass: written out by the last FLW waves, 64 per wave (dbg_wsk2.h)
    uint32_t fl_cnt = 0;
    unsigned long long fl_got = 0;
    auto flush_issue = [&]() {
        const uint32_t nq = min(s.n_q, (uint32_t)QBUF);
        const uint32_t fw = wave - (uint32_t)(NT / 64 - FLW);
        fl_cnt = 0;
        if (fw < (uint32_t)FLW && fw * 64 < nq) {
            fl_cnt = min(64u, nq - fw * 64);
            if (lane == 0) fl_got = atomicAdd(&fresh_args2(argp)->out.scalars[SK2_QUERY_CURSOR], (unsigned long long)fl_cnt);
        }
    };
    auto flush_finish = [&]() {
        if (!fl_cnt) return;
        const auto &oq = fresh_args2(argp)->out;
        const unsigned long long qb = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(fl_got >> 32)) << 32) |
                                      __builtin_amdgcn_readfirstlane((uint32_t)fl_got);
        const uint32_t at = (wave - (uint32_t)(NT / 64 - FLW)) * 64 + lane;
        if (qb + fl_cnt > oq.q_cap || qb + fl_cnt > 0xFFFFFFF0ull) {
            if (lane == 0) { atomicOr(&oq.scalars[0], 64ull); s.fail = 1; }
        } else if (lane < fl_cnt) {
            oq.q_key[qb + lane] = s.qs_key[at];
            oq.q_col[qb + lane] = s.qs_col[at];
        }
        fl_cnt = 0;
    };
    load_range(blockIdx.x);
    prefetch(blockIdx.x);
    for (uint32_t bucket = blockIdx.x; bucket < n_buckets; bucket += gridDim.x) {
        const uint64_t r_beg = nx_beg;
        const uint32_t r_n = nx_n;
        if (r_n == 0) { prefetch(bucket + gridDim.x); continue; }
        bool have_pf = true;
        const bool check16 = r_n >= 65536u / 20u;  // can one edge of this bucket be seen 65 536 times?  (a record holds at most 19 k-mers)
        uint32_t stk_n = 1;
        bool root = true, failed = false;
        if (split_recs && r_n > split_recs) {
            uint32_t parts = 2;
            while (parts < 16 && (uint64_t)parts * split_recs < r_n) parts <<= 1;
            __syncthreads();
            if (threadIdx.x < parts) {
                uint32_t td = threadIdx.x;
                asm volatile("" : "+v"(td));
                s.stk_mask[td] = parts - 1;
                s.stk_val[td] = td;
            }
            stk_n = parts;
            root = false;
            __syncthreads();
        }
        while (stk_n) {
            uint32_t cur_mask = 0, cur_val = 0;
            --stk_n;
            if (!root) { cur_mask = s.stk_mask[stk_n]; cur_val = s.stk_val[stk_n]; }
            root = false;
            __syncthreads();
            CNT_TICK(0);
            flush_issue();
            if (!clean) {
                for (int i = threadIdx.x; i < CAP; i += NT) {
                    s.keys[i] = EMPTY_KEY;
                    s.stamp[i] = (ST)~(ST)0;
                    reinterpret_cast<uint2 *>(s.cnt2)[i] = make_uint2(0, 0);
                    s.hint[i] = 0;
                }
            } else {
#pragma unroll
                for (int u = 0; u < NPT; ++u) {
                    const uint32_t li = threadIdx.x + u * NT;
                    if (li < prev_n) s.keys[s.list[li]] = EMPTY_KEY;  // (list[] is the quad list from the next barrier on)
                }
            }
            clean = false;
            if (threadIdx.x == 0) { s.overflow = 0; s.n_local = 0; s.n_new = 0; }
            uint32_t my_new = 0;  // nodes | edges << 16 this lane saw first
            // ---- insert
            for (uint32_t c0 = 0; c0 < r_n; c0 += STAGE) {
                const uint32_t n_st = min(STAGE, r_n - c0);
                if (c0) __syncthreads();  // an earlier round is done with the staging (round 0: the top barrier; the clear above
                                          // reads list[] before the next barrier, the quad list is written after it)
                if (threadIdx.x == 0) s.n_flat = 0;
                if (c0 == 0 && have_pf) {
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
                {   // identical records collapse to one representative with a multiplicity and the smallest stamp (k_sk_count2)
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
                if (c0 == 0) flush_finish();  // (the waves without records: the previous pass's queries)
                __syncthreads();
                CNT_TICK(3);
                const uint32_t n_flat = s.n_flat;
                uint32_t p_old = 1, p_shf = 0, p_mult = 0;  // the previous iteration's counter add
                for (uint32_t f0 = 0; f0 < n_flat; f0 += NT / 4) {
                    if (f0 + wave * 16 >= n_flat) break;  // wave-uniform: nothing left for this wave
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
                    if (act) {
#pragma unroll 8
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
                    {   // the previous iteration's counter add has long returned: 0 in its field = first instance of that edge
                        const uint32_t was = (p_old >> p_shf) & 0xFFFFu;
                        my_new += won + (was == 0 ? 0x10000u : 0u);
                        if (check16 && was + p_mult > 0xFFFFu) atomicOr(&fresh_args2(argp)->out.scalars[0], 512ull);
                    }
                    const bool good = act && ok;
                    const uint32_t nxt = from_next_lane(good ? slot : 0xFFFFu, 0xFFFFu);
                    const bool edge = good && has_succ;
                    const bool in_wave = edge && (i < len - 1) && nxt != 0xFFFFu;
                    p_old = 1; p_shf = 0; p_mult = 0;
                    if (edge) {
                        p_shf = 16 * (b & 1);
                        p_mult = mult;
                        p_old = atomicAdd(&s.cnt2[slot * 2 + (b >> 1)], mult << p_shf);
                    }
                    if (good) atomicMin(&s.stamp[slot], stamp);
                    s.hint[in_wave ? slot : (uint32_t)CAP + lane] = (uint16_t)(HINT_OK | (b << 12) | (nxt & (CAP - 1)));
                }
                {   // the last iteration's add
                    const uint32_t was = (p_old >> p_shf) & 0xFFFFu;
                    my_new += was == 0 ? 0x10000u : 0u;
                    if (check16 && was + p_mult > 0xFFFFu) atomicOr(&fresh_args2(argp)->out.scalars[0], 512ull);
                }
            }
            {
                const uint32_t tot = wave_sum_dpp(my_new);
                if (lane == 63 && tot) atomicAdd(&s.n_new, tot);
            }
            CNT_TICK(4);
            __syncthreads();
            CNT_TICK(5);
            flush_finish();
            const uint32_t n_new = s.n_new;
            const bool over = s.overflow != 0;
            unsigned long long got = 0;
            if (threadIdx.x == 0) {
                s.n_q = 0;
                if (!over) got = atomicAdd(&fresh_args2(argp)->out.scalars[4], (unsigned long long)(n_new & 0xFFFFu) | ((unsigned long long)(n_new >> 16) << 32));
            }
            if (have_pf) {
                have_pf = false;
                prefetch(bucket + gridDim.x);
            }
            if (over) {  // split this hash sub-range in two and retry (nothing was written out)
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
            // ---- dense list of occupied slots + CSR edge offsets (list[] and eoff[] held the quad list and the dedupe set until
            //      the insert barrier)
            cnt_dense_list<CAP, NT>(s, s.keys);
            CNT_TICK(6);
            if (threadIdx.x == 0) cnt_take_reservation(s, fresh_args2(argp)->out, got, n_new, bucket, cur_mask, cur_val);  // the reservation is back
            CNT_TICK(9);
            __syncthreads();
            CNT_TICK(10);
            if (s.n_local != n_new) {  // uniform: the insert and the list phase disagree about this bucket
                if (threadIdx.x == 0) atomicOr(&fresh_args2(argp)->out.scalars[0], 2048ull);
                failed = true;
                break;
            }
            if (s.fail) break;
            const uint32_t n_local = n_new & 0xFFFFu;
            const uint64_t gbase = s.gbase, ebase = s.ebase;
            // ---- write nodes and their CSR rows; counters, stamp and hint of every slot read are cleared, the keys stay for the
            //      deferred lookups (cleared at the top of the next pass)
            const auto &ow = fresh_args2(argp)->out;
            cnt_write_directory<CAP>(s, ow, gbase);
            // successor of the k-mer in slot `sl` by base b, CSR position e: table lookup; a miss becomes a query (the rare
            // path of an edge that found no room on its wave's list)
            auto resolve = [&](uint32_t sl, uint32_t b, uint64_t e) {
                const uint64_t sk = ((s.keys[sl] << 2) | (uint64_t)b) & kmask;
                const int f = lds_find<CAP>(s.keys, sk);
                if (f >= 0) {
                    const uint32_t ix = cnt_local_index(s, (uint32_t)f);
                    ow.col[e] = (uint32_t)(gbase + ix) | ow.id_tag;
                    return;
                }
                const uint32_t qi = atomicAdd(&s.n_q, 1u);
                if (qi < (uint32_t)QBUF) {
                    s.qs_key[qi] = sk;
                    s.qs_col[qi] = (uint32_t)e;
                } else {
                    const unsigned long long g = atomicAdd(&ow.scalars[SK2_QUERY_CURSOR], 1ull);
                    if (g >= ow.q_cap || g >= 0xFFFFFFF0ull) { atomicOr(&ow.scalars[0], 64ull); return; }
                    ow.q_key[g] = sk;
                    ow.q_col[g] = (uint32_t)e;
                }
            };
            uint16_t *dseg = s.dseg + wave * (2 * DSEG);
            uint32_t dcur = 0;  // wave-uniform: deferred edges of this wave
#pragma unroll 1
            for (int u = 0; u < NPT; ++u) {
                if ((uint32_t)(u * NT) >= n_local) break;
                const uint32_t li = threadIdx.x + u * NT;
                uint32_t dmask = 0, i = 0, e_rel = 0, nzm = 0;
                if (li < n_local) {
                    i = s.list[li];
                    const unsigned long long key = s.keys[i];
                    const uint64_t node = gbase + li;
                    uint32_t c[4];
                    cnt3_load(s.cnt2, i, c);
                    const ST stamp = s.stamp[i];
                    const uint32_t hnt = s.hint[i];
                    s.stamp[i] = (ST)~(ST)0;
                    reinterpret_cast<uint2 *>(s.cnt2)[i] = make_uint2(0, 0);
                    s.hint[i] = 0;
                    nzm = (c[0] != 0) | ((c[1] != 0) << 1) | ((c[2] != 0) << 2) | ((c[3] != 0) << 3);
                    ow.keys[node] = key;
                    reinterpret_cast<ST *>(ow.stamps)[node] = stamp;
                    ow.flags[node] = (uint8_t)((uint32_t)(stamp & 1) | (nzm << 1));
                    e_rel = s.eoff[li];
                    uint64_t e = ebase + e_rel;
                    ow.rowptr[node] = (uint32_t)e;
                    const uint32_t hb = (hnt & HINT_OK) ? ((hnt >> 12) & 3u) : 4u;
                    dmask = nzm & ~(1u << hb);
                    uint32_t hcol = 0;
                    if (nzm & (1u << hb)) {
                        const uint32_t f = hnt & (CAP - 1);
                        const uint32_t ix = cnt_local_index(s, (uint32_t)f);
                        hcol = (uint32_t)(gbase + ix) | ow.id_tag;
                    }
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        if (c[b]) {
                            ow.col[e] = (uint32_t)b == hb ? hcol : NO_NODE;  // whole lines; the deferred lookups patch their hits
                            ow.ecnt[e] = c[b];
                            ++e;
                        }
                    }
                }
                if (__ballot(dmask != 0)) {
                    const uint32_t pc = (uint32_t)__popc(dmask);
                    uint32_t at = dcur, total = 0;
#pragma unroll
                    for (int q = 1; q <= 4; ++q) {
                        const unsigned long long mq = __ballot(pc >= (uint32_t)q);
                        at += lanes_below(mq);
                        total += (uint32_t)__popcll(mq);
                    }
                    dcur += total;
                    uint32_t m = dmask;
                    while (m) {
                        const uint32_t b = __ffs(m) - 1;
                        m &= m - 1;
                        const uint32_t er = e_rel + (uint32_t)__popc(nzm & ((1u << b) - 1u));
                        if (at < (uint32_t)DSEG) {
                            dseg[2 * at] = (uint16_t)((i << 2) | b);
                            dseg[2 * at + 1] = (uint16_t)er;
                        } else {
                            resolve(i, b, ebase + er);
                        }
                        ++at;
                    }
                }
            }
            CNT_TICK(11);
            __builtin_amdgcn_wave_barrier();
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            for (uint32_t j0 = 0; j0 < min(dcur, (uint32_t)DSEG); j0 += 64) {  // uniform per wave
                const uint32_t j = j0 + lane;
                bool miss = false;
                uint64_t sk = 0, e = 0;
                if (j < min(dcur, (uint32_t)DSEG)) {
                    const uint32_t sb = dseg[2 * j], er = dseg[2 * j + 1], sl = sb >> 2;
                    e = ebase + er;
                    sk = ((s.keys[sl] << 2) | (uint64_t)(sb & 3u)) & kmask;
                    const int f = lds_find<CAP>(s.keys, sk);
                    if (f >= 0) {
                        const uint32_t ix = cnt_local_index(s, (uint32_t)f);
                        ow.col[e] = (uint32_t)(gbase + ix) | ow.id_tag;
                    } else {
                        miss = true;
                    }
                }
                const unsigned long long mm = __ballot(miss);
                if (mm) {
                    uint32_t qb = 0;
                    if (lane == 0) qb = atomicAdd(&s.n_q, (uint32_t)__popcll(mm));
                    const uint32_t qi = __builtin_amdgcn_readfirstlane(qb) + lanes_below(mm);
                    const bool direct = miss && qi >= (uint32_t)QBUF;
                    if (miss && !direct) {
                        s.qs_key[qi] = sk;
                        s.qs_col[qi] = (uint32_t)e;
                    }
                    const unsigned long long md = __ballot(direct);
                    if (md) {
                        unsigned long long g0 = 0;
                        if (lane == 0) g0 = atomicAdd(&ow.scalars[SK2_QUERY_CURSOR], (unsigned long long)__popcll(md));
                        g0 = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(g0 >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)g0);
                        const unsigned long long g = g0 + lanes_below(md);
                        if (direct) {
                            if (g >= ow.q_cap || g >= 0xFFFFFFF0ull) {
                                atomicOr(&ow.scalars[0], 64ull);
                            } else {
                                ow.q_key[g] = sk;
                                ow.q_col[g] = (uint32_t)e;
                            }
                        }
                    }
                }
            }
            CNT_TICK(12);
            clean = true;
            prev_n = n_local;
        }
        if (failed || s.fail) return;
        if (have_pf) prefetch(bucket + gridDim.x);
    }
    __syncthreads();
    flush_issue();
    flush_finish();
#ifdef DBG_CNT_PROF
    if (threadIdx.x == 0) {
        for (int i = 0; i < 31; ++i) atomicAdd(&g_cnt_prof[i], s.prof[i]);
        atomicAdd(&g_cnt_prof[31], 1ull);
    }
#endif
}

}  // namespace dbgk
