// Two-word count kernel, second generation: k_wsk_count (dbg_wsk.h) with the successor hand-over of k_sk_count2 (dbg_sk2.h).
//
// In k_wsk_count a fifth of the time goes into successor lookups: after the insert every node hashes key + base (five
// multiplies on 128 bits) and probes the table again, in waves that run as long as their slowest lane.  But k-mer i + 1 of a
// record is inserted by the NEXT LANE of the same wave in the same instruction stream, so its slot is one DPP move away:
//   * hints.  The 128-bit keys leave no room for a hint per (slot, base) -- the LDS is full -- but 97 % of the nodes have ONE
//     successor base, so one 16-bit hint per slot does: valid | base << 12 | slot of key + base.  Every instance that knows
//     its neighbour's slot writes it (address select to a dummy word for the others, no branch); with several bases the last
//     writer wins and the other bases are looked up.
//   * what the hint does not cover (last k-mer of a record, lane 63, a second successor base, a successor filtered into
//     another hash sub-range: ~10 % of the edges) is found out in the node write, goes on a list private to the wave and is
//     looked up right after it, ONE LANE PER EDGE (dense waves instead of one lookup round per base per node); hits patch
//     their CSR column, misses become queries for the resolver.  The keys stay in the table during the write -- the write clears
//     counters, stamp and hint of the slots it reads, the high words are cleared at the top of the next pass -- so these
//     lookups need no barrier.
//   * node and edge totals are counted while inserting (compare-and-swap winners; counter adds that return 0, looked at one
//     iteration later) and summed per wave with DPP adds, so the global reservation goes out right after the insert barrier
//     and has the list phase to come back; the list phase's own count must agree (flag 2048 otherwise).
//   * queries staged by one pass leave after the next pass's first barrier, written by the two waves that have nothing to do
//     during the dedupe; two barriers per pass less than k_wsk_count.
// Same inputs, same outputs (node order within a bucket = slot order), same flags as k_wsk_count.  32-bit stamps only: with
// 64-bit stamps the hint array does not fit beside the table (those builds keep k_wsk_count).
#pragma once
#include "dbg_cnt_common.h"
#include "dbg_sk2.h"
#include "dbg_wsk.h"

namespace dbgk {

struct WCnt2Cfg {
#ifdef DBG_CNT_PROF
    static constexpr int STAGE = 224;   // (room for the clocks of the experiment build)
#else
    static constexpr int STAGE = 256;   // records staged per round (a bucket holds 95 +- 50 at the default geometry)
#endif
    static constexpr int QBUF = 128;    // cross-bucket successor queries staged per pass; more go out wave by wave
    static constexpr int DD = 512;      // dedupe set slots (>= 2 * STAGE)
    static constexpr int DSEG = STAGE * WCNT_QMAX / (WCNT_NT / 64) / 2;  // deferred edges per wave: (slot << 2 | base, CSR offset) pairs
    static constexpr int FLW = (QBUF + 63) / 64;                         // waves that write the staged queries out
};
constexpr uint32_t WHINT_VALID = 0x8000u;

struct WCnt2Lds {
    static constexpr int STAGE = WCnt2Cfg::STAGE, QBUF = WCnt2Cfg::QBUF, DD = WCnt2Cfg::DD;
    unsigned long long khi[WCAP];   // EMPTY_KEY / hi | W_PEND / hi   (hi < 2^62)
    unsigned long long klo[WCAP];
    uint32_t cnt2[WCAP * 2];        // four 16-bit successor counters per slot
    uint32_t stamp[WCAP];
    uint16_t list[WCAP];            // local node index -> slot
    uint16_t eoff[WCAP];            // local node index -> first CSR edge of the node, relative to the bucket
    uint16_t hint[WCAP + 64];       // WHINT_VALID | base << 12 | slot of (key + base); [WCAP + lane]: dummy words
    unsigned long long rb[STAGE][4];
    unsigned long long rmeta[STAGE];
    uint32_t rst[STAGE];
    uint32_t dd_tab[DD];
    uint32_t dd_mult[STAGE];
    uint16_t flat[STAGE * WCNT_QMAX];  // insert: quads of the representatives; node write: the waves' deferred edges
    unsigned long long q_lo[QBUF], q_hi[QBUF];
    uint32_t q_col[QBUF];
    uint32_t stk_mask[CNT_STACK], stk_val[CNT_STACK];
    uint32_t overflow, n_local, n_new, n_q, fail, n_flat;
    unsigned long long gbase, ebase, ri;
    unsigned long long dir_mask[WCAP / 64];
    uint16_t dir_base[WCAP / 64];
#ifdef DBG_CNT_PROF
    unsigned long long prof[64];
#endif
};
static_assert(sizeof(WCnt2Lds) <= 160 * 1024, "LDS of the two-word count kernel, second generation");

template <class ST>
__global__ __launch_bounds__(WCNT_NT) void k_wsk_count2(const uint64_t *__restrict__ b_start, const uint64_t *__restrict__ b_cnt,
                                                       const uint64_t *__restrict__ rec_w0, const uint64_t *__restrict__ rec_w1,
                                                       const ST *__restrict__ rec_st, const uint4 *__restrict__ rec_b /* k_wsk_gather */,
                                                       int k, uint64_t n_buckets, const WSkCountOut *__restrict__ outp /* in device memory */,
                                                       uint32_t split_recs) {
    static_assert(sizeof(ST) == 4, "k_wsk_count2 holds 32-bit stamps");
    extern __shared__ __attribute__((aligned(16))) unsigned char wcnt_raw[];
    WCnt2Lds &s = *reinterpret_cast<WCnt2Lds *>(wcnt_raw);
    constexpr int NPT = WCAP / WCNT_NT;
    constexpr int STAGE = WCnt2Cfg::STAGE, QBUF = WCnt2Cfg::QBUF, DD = WCnt2Cfg::DD, DSEG = WCnt2Cfg::DSEG, FLW = WCnt2Cfg::FLW;
    static_assert(DD >= 2 * STAGE && STAGE <= WCNT_NT && FLW <= WCNT_NT / 64 - STAGE / 64, "staging");
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool clean = false;       // uniform: the last pass wrote its nodes out (counters, stamps and hints of its slots are clear)
    uint32_t prev_n = 0;      // ... and these many nodes of it still have their high word in the table
    if (threadIdx.x == 0) { s.fail = 0; s.n_q = 0; }
#ifdef DBG_CNT_PROF
    unsigned long long clast_ = clock64();
    if (threadIdx.x < 64) s.prof[threadIdx.x] = 0;
    __syncthreads();
#endif
    (void)rec_w0;
    uint64_t pf_w1 = 0;
    uint4 pf_b0 = make_uint4(0, 0, 0, 0), pf_b1 = make_uint4(0, 0, 0, 0);
    ST pf_st = 0;
    uint64_t nx_beg = 0, nx_n = 0, r2_beg = 0, r2_n = 0;
    auto load_range = [&](uint64_t b, uint64_t &beg, uint64_t &n) {
        n = 0;
        if (b < n_buckets) { beg = b_start[b]; n = b_cnt[b]; }
    };
    auto prefetch_recs = [&](uint64_t b) {  // records of bucket b (its range is in r2_*), then the range after it
        nx_beg = r2_beg;
        nx_n = r2_n;
        if (threadIdx.x < min(nx_n, (uint64_t)STAGE)) {
            pf_w1 = rec_w1[nx_beg + threadIdx.x];
            pf_st = rec_st[nx_beg + threadIdx.x];
            pf_b0 = rec_b[2 * (nx_beg + threadIdx.x)];
            pf_b1 = rec_b[2 * (nx_beg + threadIdx.x) + 1];
        }
        load_range(b + gridDim.x, r2_beg, r2_n);
    };
    auto stage_regs = [&](uint32_t n_st) {  // see k_wsk_count
        if (threadIdx.x == 0) s.n_flat = 0;
        if (threadIdx.x < n_st) {
            s.rb[threadIdx.x][0] = ((unsigned long long)pf_b0.y << 32) | pf_b0.x;
            s.rb[threadIdx.x][1] = ((unsigned long long)pf_b0.w << 32) | pf_b0.z;
            s.rb[threadIdx.x][2] = ((unsigned long long)pf_b1.y << 32) | pf_b1.x;
            s.rb[threadIdx.x][3] = ((unsigned long long)pf_b1.w << 32) | pf_b1.z;
            s.rmeta[threadIdx.x] = pf_w1;
            s.rst[threadIdx.x] = pf_st;
            s.dd_mult[threadIdx.x] = 0;
        }
        for (uint32_t i = threadIdx.x; i < (uint32_t)DD; i += WCNT_NT) s.dd_tab[i] = 0xFFFFFFFFu;
    };
    // The queries a pass staged are written out by the last FLW waves, 64 per wave, after the next barrier: one global
    // atomic per wave, issued at once and looked at when the dedupe has the other waves busy.
    uint32_t fl_cnt = 0;
    unsigned long long fl_got = 0;
    auto flush_issue = [&]() {
        const uint32_t nq = min(s.n_q, (uint32_t)QBUF);
        const uint32_t fw = wave - (uint32_t)(WCNT_NT / 64 - FLW);  // wraps for the other waves
        fl_cnt = 0;
        if (fw < (uint32_t)FLW && fw * 64 < nq) {
            fl_cnt = min(64u, nq - fw * 64);
            if (lane == 0) fl_got = atomicAdd(&wfresh_args(outp)->scalars[5], (unsigned long long)fl_cnt);
        }
    };
    auto flush_finish = [&]() {
        if (!fl_cnt) return;
        const auto &oq = *wfresh_args(outp);
        const unsigned long long qb = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(fl_got >> 32)) << 32) |
                                      __builtin_amdgcn_readfirstlane((uint32_t)fl_got);
        const uint32_t at = (wave - (uint32_t)(WCNT_NT / 64 - FLW)) * 64 + lane;
        if (qb + fl_cnt > oq.q_cap || qb + fl_cnt > 0xFFFFFFF0ull) {
            if (lane == 0) { atomicOr(&oq.scalars[0], 64ull); s.fail = 1; }
        } else if (lane < fl_cnt) {
            oq.q_lo[qb + lane] = s.q_lo[at];
            oq.q_hi[qb + lane] = s.q_hi[at];
            oq.q_col[qb + lane] = s.q_col[at];
        }
        fl_cnt = 0;
    };
    bool staged = false;  // uniform: the staging arrays hold the first round of the bucket in nx_*
    load_range(blockIdx.x, r2_beg, r2_n);
    prefetch_recs(blockIdx.x);
    for (uint64_t bucket = blockIdx.x; bucket < n_buckets; bucket += gridDim.x) {
        const uint64_t r_beg = nx_beg, r_n = nx_n;
        bool lds_staged = staged;
        staged = false;
        if (r_n == 0) { prefetch_recs(bucket + gridDim.x); continue; }
        bool have_pf = true;
        const bool small_bucket = r_n * 51 < 0xFFFFull;  // a record holds at most 51 k-mers: no 16-bit counter can wrap
        uint32_t stk_n = 1;
        bool root = true, failed = false;
        if (split_recs && r_n > split_recs) {
            uint32_t parts = 2;
            while (parts < 16 && (uint64_t)parts * split_recs < r_n) parts <<= 1;
            __syncthreads();
            if (threadIdx.x < parts) {
                uint32_t td = threadIdx.x;
                asm volatile("" : "+v"(td));  // (see the directory write below)
                s.stk_mask[td] = parts - 1;
                s.stk_val[td] = td;
            }
            stk_n = parts;
            root = false;
            __syncthreads();
            CNT_TICK(0);
        }
        while (stk_n) {
            uint32_t cur_mask = 0, cur_val = 0;
            --stk_n;
            if (!root) { cur_mask = s.stk_mask[stk_n]; cur_val = s.stk_val[stk_n]; }
            root = false;
#if defined(DBG_CNT_PROF) && DBG_CNT_PROF == 3
            if (lane == 0) s.prof[32 + wave] = clock64();  // arrival at the top barrier: who is last, and by how much
#endif
            __syncthreads();
#if defined(DBG_CNT_PROF) && DBG_CNT_PROF == 3
            if (threadIdx.x == 0) {
                unsigned long long mx = 0, mine = s.prof[32];
                int who = 0;
                for (int w2 = 0; w2 < WCNT_NT / 64; ++w2) if (s.prof[32 + w2] > mx) { mx = s.prof[32 + w2]; who = w2; }
                s.prof[50] += mx - mine;
                s.prof[52 + who / 2] += 1;  // (pairs of waves)
                s.prof[51] += 1;
            }
#endif
            flush_issue();
            if (!clean) {
                for (int i = threadIdx.x; i < WCAP; i += WCNT_NT) {
                    s.khi[i] = EMPTY_KEY;
                    s.stamp[i] = 0xFFFFFFFFu;
                    reinterpret_cast<uint2 *>(s.cnt2)[i] = make_uint2(0, 0);
                    s.hint[i] = 0;
                }
            } else {
#pragma unroll
                for (int u = 0; u < NPT; ++u) {
                    const uint32_t li = threadIdx.x + u * WCNT_NT;
                    if (li < prev_n) s.khi[s.list[li]] = EMPTY_KEY;
                }
            }
            clean = false;
            if (threadIdx.x == 0) { s.overflow = 0; s.n_local = 0; s.n_new = 0; }
            CNT_TICK(13);
            // ---- insert
            uint32_t my_new = 0;  // nodes | edges << 16 this lane saw first
            for (uint64_t c0 = 0; c0 < r_n; c0 += STAGE) {
                const uint32_t n_st = (uint32_t)min((uint64_t)STAGE, r_n - c0);
                if (c0) __syncthreads();
                if (lds_staged && c0 == 0) {
                    CNT_TICK(14);
                    CNT_EVENT(21);
                } else {
                    CNT_EVENT(22);
                    if (have_pf && c0 == 0) {
                        stage_regs(n_st);
                    } else {
                        if (threadIdx.x == 0) s.n_flat = 0;
                        if (threadIdx.x < n_st) {  // one record per thread
                            const uint64_t ri = r_beg + c0 + threadIdx.x;
                            const uint4 b0 = rec_b[2 * ri], b1 = rec_b[2 * ri + 1];
                            s.rb[threadIdx.x][0] = ((unsigned long long)b0.y << 32) | b0.x;
                            s.rb[threadIdx.x][1] = ((unsigned long long)b0.w << 32) | b0.z;
                            s.rb[threadIdx.x][2] = ((unsigned long long)b1.y << 32) | b1.x;
                            s.rb[threadIdx.x][3] = ((unsigned long long)b1.w << 32) | b1.z;
                            s.rmeta[threadIdx.x] = rec_w1[ri];
                            s.rst[threadIdx.x] = rec_st[ri];
                            s.dd_mult[threadIdx.x] = 0;
                        }
                        for (uint32_t i = threadIdx.x; i < (uint32_t)DD; i += WCNT_NT) s.dd_tab[i] = 0xFFFFFFFFu;
                    }
                    CNT_TICK(15);
                    __syncthreads();
                    CNT_TICK(1);
                    if (c0 && s.overflow) break;  // an earlier round of this pass overflowed the table
                }
                lds_staged = false;
                // ---- identical records collapse to one representative (k_wsk_count); the representatives list their quads
                {
                    uint32_t nquad = 0;
                    const uint32_t r = threadIdx.x;
                    if (r < n_st) {
                        const unsigned long long b0 = s.rb[r][0], b1 = s.rb[r][1], b2 = s.rb[r][2], b3 = s.rb[r][3];
                        const unsigned long long mt = s.rmeta[r] & ~(((1ull << SK_BUCKET_BITS) - 1) << 6);  // length + flag
                        uint32_t hslot = (uint32_t)(mix64(b0 ^ mix64(b1 + 0x9E3779B97F4A7C15ull) ^ mix64(b2 ^ (b3 * 0xD6E8FEB86659FD93ull)) ^ mt) >> 40) &
                                         (DD - 1);
                        uint32_t rep = r;
                        for (uint32_t probe = 0; probe < (uint32_t)DD; ++probe) {
                            uint32_t cur = s.dd_tab[hslot];
                            if (cur == 0xFFFFFFFFu) {
                                cur = atomicCAS(&s.dd_tab[hslot], 0xFFFFFFFFu, r);
                                if (cur == 0xFFFFFFFFu) break;
                            }
                            if (s.rb[cur][0] == b0 && s.rb[cur][1] == b1 && s.rb[cur][2] == b2 && s.rb[cur][3] == b3 &&
                                (s.rmeta[cur] & ~(((1ull << SK_BUCKET_BITS) - 1) << 6)) == mt) { rep = cur; break; }
                            hslot = (hslot + 1) & (DD - 1);
                        }
                        atomicAdd(&s.dd_mult[rep], 1u);
                        if (rep != r) atomicMin(&s.rst[rep], s.rst[r]);
                        else nquad = ((uint32_t)wrec_len(mt) + 3) >> 2;
                    }
                    const uint32_t base = wave_alloc_n<WCNT_QMAX>(&s.n_flat, nquad);
                    for (uint32_t q = 0; q < nquad; ++q) s.flat[base + q] = (uint16_t)((r << 4) | q);
                }
                flush_finish();  // (the waves without records: the previous pass's queries)
                __syncthreads();
                CNT_TICK(3);
                const uint32_t n_flat = s.n_flat;
#ifdef DBG_CNT_PROF
                const unsigned long long wt0_ = clock64();
#endif
                // every lane stays in the loop (predicated): the wave hands successor slots from lane to lane.  The quads of a
                // record are consecutive in the list, so k-mer i + 1 of a record is in the next lane.
                uint32_t p_old = 1, p_shf = 0, p_mult = 0;  // the previous iteration's counter add
                for (uint32_t f0 = 0; f0 < n_flat; f0 += WCNT_NT / 4) {
                    if (f0 + wave * 16 >= n_flat) break;  // wave-uniform: nothing left for this wave (the list is dealt out in lane order)
                    const uint32_t f = f0 + (threadIdx.x >> 2);
                    bool act = f < n_flat;
                    const uint32_t e = act ? s.flat[f] : 0u;
                    const uint32_t r = e >> 4;
                    const int i = (int)((e & 15) * 4 + (threadIdx.x & 3));
                    const unsigned long long w1 = s.rmeta[r];
                    const int len = wrec_len(w1);
                    act = act && i < len;
                    const int wi = i >> 5, sh = (i & 31) * 2;
                    const unsigned long long a0 = s.rb[r][wi], a1 = s.rb[r][wi + 1], a2 = s.rb[r][wi + 2];
                    const uint64_t A = sh ? (a0 << sh) | (a1 >> (64 - sh)) : a0;
                    const uint64_t B = sh ? (a1 << sh) | (a2 >> (64 - sh)) : a1;
                    const K128 key = k128_from_windows(A, B, k);
                    if (cur_mask) act = act && (wsub_hash(key) & cur_mask) == cur_val;
                    const bool has_succ = (i < len - 1) || wrec_has_succ(w1);
                    const uint32_t b = base_after_kmer(B, k);
                    const ST st0 = s.rst[r];
                    const ST stamp = i ? (ST)((st0 | (ST)1) + (ST)(2 * i)) : st0;
                    const uint32_t mult = s.dd_mult[r];
                    uint32_t slot = wslot_of(key);
                    bool ok = false;
                    uint32_t won = 0;
                    if (act) {  // the claim protocol of k_wsk_count
                        for (int probe = 0, spins = 0; probe < CNT_PROBE_LIMIT && spins < (1 << 16); ) {
                            const unsigned long long cur = atomicCAS(&s.khi[slot], EMPTY_KEY, key.hi | W_PEND);
                            if (cur == EMPTY_KEY) {
                                wlds_store(&s.klo[slot], key.lo);
                                wlds_store(&s.khi[slot], key.hi);
                                ok = true;
                                won = 1;
                                break;
                            }
                            if ((cur & ~W_PEND) == key.hi) {  // this k-mer or one that shares its high word
                                if (cur & W_PEND) {
                                    if (++spins >= (1 << 16)) atomicOr(&wfresh_args(outp)->scalars[0], 1024ull);  // a claim that never completes: fail loudly
                                    continue;
                                }
                                if (wlds_load(&s.klo[slot]) == key.lo) { ok = true; break; }
                            }
                            slot = (slot + 1) & (WCAP - 1);
                            ++probe;
                        }
                        if (!ok) s.overflow = 1;
                    }
                    // the previous iteration's counter add has long returned: 0 in its field = first instance of that edge
                    {
                        const uint32_t was = (p_old >> p_shf) & 0xFFFFu;
                        my_new += won + (was == 0 ? 0x10000u : 0u);
                        if (!small_bucket && was + p_mult > 0xFFFFu) atomicOr(&wfresh_args(outp)->scalars[0], 512ull);  // 16-bit counter overflow
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
                    s.hint[in_wave ? slot : (uint32_t)WCAP + lane] = (uint16_t)(WHINT_VALID | (b << 12) | (nxt & (WCAP - 1)));
                }
                {   // the last iteration's add
                    const uint32_t was = (p_old >> p_shf) & 0xFFFFu;
                    my_new += was == 0 ? 0x10000u : 0u;
                    if (!small_bucket && was + p_mult > 0xFFFFu) atomicOr(&wfresh_args(outp)->scalars[0], 512ull);
                }
#ifdef DBG_CNT_PROF
#if DBG_CNT_PROF == 1
                if ((threadIdx.x & 63) == 0) s.prof[32 + (threadIdx.x >> 6)] += clock64() - wt0_;
#endif
                if (threadIdx.x == 0) { s.prof[48] += n_flat; s.prof[49] += n_st; }
#endif
            }
            {   // this wave's share of the bucket's node and edge totals
                const uint32_t tot = wave_sum_dpp(my_new);
                if (lane == 63 && tot) atomicAdd(&s.n_new, tot);
            }
            CNT_TICK(4);
            __syncthreads();
            CNT_TICK(5);
            flush_finish();  // (a bucket without rounds to run cannot get here, but an early `break` above can)
            const uint32_t n_new = s.n_new;
            const bool over = s.overflow != 0;
            unsigned long long got = 0;
            if (threadIdx.x == 0) {
                s.n_q = 0;  // the previous pass's queries are out; the node write below stages this pass's
                if (!over) got = atomicAdd(&wfresh_args(outp)->scalars[4], (unsigned long long)(n_new & 0xFFFFu) | ((unsigned long long)(n_new >> 16) << 32));
            }
            if (have_pf) {  // the registers are free: the next bucket's records, under the rest of this one
                have_pf = false;
                prefetch_recs(bucket + gridDim.x);
            }
            if (over) {  // split this hash sub-range in two and retry (nothing was written out)
                const uint32_t bit = cur_mask + 1;
                if (stk_n + 2 > CNT_STACK || bit >= (1u << 20)) {
                    if (threadIdx.x == 0) atomicOr(&wfresh_args(outp)->scalars[0], 8ull);
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
            // ---- dense list of occupied slots + CSR edge offsets
            cnt_dense_list<WCAP, WCNT_NT>(s, s.khi);
            CNT_TICK(6);
            if (threadIdx.x == 0) cnt_take_reservation(s, *wfresh_args(outp), got, n_new, bucket, cur_mask, cur_val);  // the reservation is back
            CNT_TICK(9);
            __syncthreads();
            CNT_TICK(10);
            if (s.n_local != n_new) {  // uniform: the insert and the list phase disagree about this bucket
                if (threadIdx.x == 0) atomicOr(&wfresh_args(outp)->scalars[0], 2048ull);
                failed = true;
                break;
            }
            if (s.fail) break;
            const uint32_t n_local = n_new & 0xFFFFu;
            const uint64_t gbase = s.gbase, ebase = s.ebase;
            if (stk_n == 0 && !have_pf) {  // uniform: last pass of this bucket, the registers hold the next one's first round
                stage_regs((uint32_t)min(nx_n, (uint64_t)STAGE));
                staged = true;
            }
            // ---- write nodes and their CSR rows.  Counters, stamp and hint of every slot read are cleared for the next pass;
            //      the keys stay for the deferred lookups (their high words are cleared at the top of the next pass).
#if defined(DBG_CNT_PROF) && DBG_CNT_PROF == 2
            const unsigned long long ww0_ = clock64();  // per-wave clocks of the node write + deferred lookups instead of the insert
#endif
            const auto &ow = *wfresh_args(outp);  // loaded here, not kept in SGPRs across the whole bucket loop
            cnt_write_directory<WCAP>(s, ow, gbase);
            // successor of the k-mer in slot `sl` by base b, CSR position e: table lookup; a miss becomes a query
            auto resolve = [&](uint32_t sl, uint32_t b, uint64_t e) {
                const K128 sk = k128_append(K128{s.khi[sl], s.klo[sl]}, b, k);
                const int f = wlds_find(s.khi, s.klo, sk);
                if (f >= 0) {
                    const uint32_t ix = cnt_local_index(s, (uint32_t)f);
                    ow.col[e] = (uint32_t)(gbase + ix) | ow.id_tag;
                    return;
                }
                const uint32_t qi = atomicAdd(&s.n_q, 1u);
                if (qi < (uint32_t)QBUF) {
                    s.q_lo[qi] = sk.lo;
                    s.q_hi[qi] = sk.hi;
                    s.q_col[qi] = (uint32_t)e;
                } else {  // rare: past the staging, one by one
                    const unsigned long long g = atomicAdd(&ow.scalars[5], 1ull);
                    if (g >= ow.q_cap || g >= 0xFFFFFFF0ull) { atomicOr(&ow.scalars[0], 64ull); return; }
                    ow.q_lo[g] = sk.lo;
                    ow.q_hi[g] = sk.hi;
                    ow.q_col[g] = (uint32_t)e;
                }
            };
            uint16_t *dseg = s.flat + wave * (2 * DSEG);
            uint32_t dcur = 0;  // wave-uniform: deferred edges of this wave
#pragma unroll 1
            for (int u = 0; u < NPT; ++u) {  // (one round at the default geometry: ~1000 nodes per bucket)
                if ((uint32_t)(u * WCNT_NT) >= n_local) break;
                const uint32_t li = threadIdx.x + u * WCNT_NT;
                uint32_t dmask = 0, i = 0, e_rel = 0, nzm = 0;
                if (li < n_local) {
                    i = s.list[li];
                    const unsigned long long khi = s.khi[i], klo = s.klo[i];
                    const uint64_t node = gbase + li;
                    uint32_t c[4];
                    wcnt_load(s.cnt2, i, c);
                    const ST stamp = s.stamp[i];
                    const uint32_t hnt = s.hint[i];
                    s.stamp[i] = 0xFFFFFFFFu;
                    reinterpret_cast<uint2 *>(s.cnt2)[i] = make_uint2(0, 0);
                    s.hint[i] = 0;
                    nzm = (c[0] != 0) | ((c[1] != 0) << 1) | ((c[2] != 0) << 2) | ((c[3] != 0) << 3);
                    ow.keys[node] = klo;
                    ow.keys_hi[node] = khi;
                    reinterpret_cast<ST *>(ow.stamps)[node] = stamp;
                    ow.flags[node] = (uint8_t)((uint32_t)(stamp & 1) | (nzm << 1));
                    e_rel = s.eoff[li];
                    uint64_t e = ebase + e_rel;
                    ow.rowptr[node] = (uint32_t)e;
                    const uint32_t hb = (hnt & WHINT_VALID) ? ((hnt >> 12) & 3u) : 4u;
                    dmask = nzm & ~(1u << hb);
                    uint32_t hcol = 0;
                    if (nzm & (1u << hb)) {
                        const uint32_t f = hnt & (WCAP - 1);
                        const uint32_t ix = cnt_local_index(s, (uint32_t)f);
                        hcol = (uint32_t)(gbase + ix) | ow.id_tag;
                    }
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        if (c[b]) {
                            ow.col[e] = (uint32_t)b == hb ? hcol : NO_NODE;  // whole lines now; the deferred lookups patch their hits
                            ow.ecnt[e] = c[b];
                            ++e;
                        }
                    }
                }
                // the edges without a hint: onto this wave's list (wave-uniform cursor + ballots), or looked up here when it is full
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
                K128 sk{0, 0};
                uint64_t e = 0;
                if (j < min(dcur, (uint32_t)DSEG)) {
                    const uint32_t sb = dseg[2 * j], er = dseg[2 * j + 1], sl = sb >> 2;
                    e = ebase + er;
                    sk = k128_append(K128{s.khi[sl], s.klo[sl]}, sb & 3u, k);
                    const int f = wlds_find(s.khi, s.klo, sk);
                    if (f >= 0) {
                        const uint32_t ix = cnt_local_index(s, (uint32_t)f);
                        ow.col[e] = (uint32_t)(gbase + ix) | ow.id_tag;
                    } else {
                        miss = true;
                    }
                }
                // the misses become queries: one LDS atomic per wave for their places in the staging, one global atomic per wave
                // for those that do not fit any more
                const unsigned long long mm = __ballot(miss);
                if (mm) {
                    uint32_t qb = 0;
                    if (lane == 0) qb = atomicAdd(&s.n_q, (uint32_t)__popcll(mm));
                    const uint32_t qi = __builtin_amdgcn_readfirstlane(qb) + lanes_below(mm);
                    const bool direct = miss && qi >= (uint32_t)QBUF;
                    if (miss && !direct) {
                        s.q_lo[qi] = sk.lo;
                        s.q_hi[qi] = sk.hi;
                        s.q_col[qi] = (uint32_t)e;
                    }
                    const unsigned long long md = __ballot(direct);
                    if (md) {
                        unsigned long long g0 = 0;
                        if (lane == 0) g0 = atomicAdd(&ow.scalars[5], (unsigned long long)__popcll(md));
                        g0 = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(g0 >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)g0);
                        const unsigned long long g = g0 + lanes_below(md);
                        if (direct) {
                            if (g >= ow.q_cap || g >= 0xFFFFFFF0ull) {
                                atomicOr(&ow.scalars[0], 64ull);
                            } else {
                                ow.q_lo[g] = sk.lo;
                                ow.q_hi[g] = sk.hi;
                                ow.q_col[g] = (uint32_t)e;
                            }
                        }
                    }
                }
            }
            CNT_TICK(12);
#if defined(DBG_CNT_PROF) && DBG_CNT_PROF == 2
            if ((threadIdx.x & 63) == 0) s.prof[32 + (threadIdx.x >> 6)] += clock64() - ww0_;
#endif
            clean = true;
            prev_n = n_local;
        }
        if (failed || s.fail) return;
        if (have_pf) prefetch_recs(bucket + gridDim.x);
    }
    __syncthreads();
    flush_issue();
    flush_finish();
#ifdef DBG_CNT_PROF
    if (threadIdx.x == 0) {
        for (int i = 0; i < 64; ++i) if (i != 31) atomicAdd(&g_cnt_prof[i], s.prof[i]);
        atomicAdd(&g_cnt_prof[31], 1ull);
    }
#endif
}

}  // namespace dbgk
