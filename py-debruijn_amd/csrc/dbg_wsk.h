// Super-k-mer engine for two-word k-mers: 32 <= k <= 63 over ACGT (BASELINE.json configs[4] key width).
//
// The one-word engine (dbg_sk.h) with three differences that follow from the key width:
//   * a super-k-mer holds up to k - 12 k-mers of k bases each -- up to 115 bases, four words -- so a record does not
//     carry bases: w0 = position of its first k-mer in the 2-bit packed reads (k_wpack), w1 = bucket hash | length |
//     flags, st = stamp.  Same three arrays as the one-word records: the multisplit kernels move them unchanged,
//     and there are only N_k / 26 of them at k = 63.  The count kernel reads each record's bases once from the packed
//     reads (40 contiguous bytes) into LDS, aligned.
//   * the LDS table holds 128-bit keys.  There is no 128-bit LDS atomic: a slot is claimed by a 64-bit
//     compare-and-swap on the HIGH word (at most 62 significant bits, so all-ones marks "empty" and bit 63 "low word
//     not written yet"), the winner stores the low word and then the high word without the pending bit; a probe
//     that meets a pending slot looks again.  After the insert phase nothing is pending and every later phase reads
//     plain keys.
//   * 16-byte keys: to keep 4096 slots per bucket in the 160 KB of LDS the four successor counters of a slot are
//     16-bit fields (two per dword); an edge seen more than 65 535 times raises a flag and the build falls back to the
//     global-table engine of dbg_wide.h (as that engine's own packed counters do).  4096 slots matter more here than for
//     one-word k-mers: one minimizer occurrence of the genome brings ~320 nodes at k = 63 (26 k-mers per record x 30x
//     coverage + the error k-mers), so a bucket holds only a handful of them and its size varies like a Poisson count.
// Everything else -- dedupe of identical records, quads of 4 k-mers on 4 lanes, dense node list + CSR edge offsets
// from ballots, in-bucket successor lookup, one packed global reservation per bucket, directory for the resolver,
// hash sub-ranges for oversized buckets -- is the design of k_sk_count; see there for the reasons.
//
//   k_wsk_extract   reads -> records                                           [debruijn.py:123-128 windows]
//   k_wsk_estimate  distinct / instances of one level-1 bucket (sizes the partition)
//   k_wsk_count     per-bucket counting in LDS -> keys (lo, hi), stamps, flags, CSR  [debruijn.py:129-143, :213-222]
//   k_wsucc_resolve successors that live in another bucket, through the directory
#pragma once
#include "dbg_sk.h"
#include "dbg_wide.h"

namespace dbgk {

constexpr int WSK_HALO = 128;
using TileLdsW = TileLdsT<WSK_HALO>;

// record meta word: bits 27..6 bucket hash (where the one-word records keep it: ms_child is shared), bits 33..28
// length - 1 (a super-k-mer holds at most k - 12 <= 51 k-mers), bit 0: the last k-mer has a successor
__host__ __device__ inline int wrec_len(uint64_t w1) { return (int)((w1 >> SK_META_BITS) & 63) + 1; }
__host__ __device__ inline uint32_t wrec_has_succ(uint64_t w1) { return (uint32_t)(w1 & 1); }

// 2m bits of a 2k-bit K128 value starting `off` bits above its least significant bit (off + 2m <= 2k)
__device__ inline uint32_t k128_bits(K128 v, int off, uint32_t mask) {
    if (off >= 64) return (uint32_t)(v.hi >> (off - 64)) & mask;
    const uint64_t lo = v.lo >> off;
    return (uint32_t)(off ? lo | (v.hi << (64 - off)) : lo) & mask;
}

// minimizer-hash bucket of a two-word k-mer: the same choice as the extraction (smallest 16-bit hash, leftmost on ties)
__device__ inline uint32_t wkmer_bucket22(K128 kmer, int k, int m) {
    const int w = k - m + 1;
    const uint32_t mmask = (uint32_t)((1ull << (2 * m)) - 1);
    uint32_t best = 0xFFFFFFFFu, best_mm = 0;
    for (int i = 0; i < w; ++i) {
        const uint32_t mm = k128_bits(kmer, 2 * (k - m - i), mmask);
        const uint32_t hv = mmer_hash16(mm);
        if (hv < best) { best = hv; best_mm = mm; }
    }
    return bucket_hash22(best_mm);
}

constexpr int WCAP = 4096;                                              // LDS table slots per bucket
// table slot and hash sub-range of a two-word k-mer: a multiplicative fold of the four 32-bit halves (k128_hash, two
// 64-bit mixes, cost ~40 instructions per insert and per successor lookup).  A cheaper fold -- the halves rotated and
// xor-ed, two multiplies instead of five -- was measured slower (28.5 vs 27.6 ms at k = 63): its probe runs are longer.
__device__ inline uint32_t wfold32(K128 a) {
    return (uint32_t)a.lo * 0x9E3779B1u ^ (uint32_t)(a.lo >> 32) * 0x85EBCA77u ^ (uint32_t)a.hi * 0xC2B2AE3Du ^
           (uint32_t)(a.hi >> 32) * 0x27D4EB2Fu;
}
__device__ inline uint32_t wslot_of(K128 key) {
    const uint32_t x = wfold32(key);
    return ((x ^ (x >> 15)) * 0x2C1B3C6Du) >> 20;  // 12 bits
}
__device__ inline uint32_t wsub_hash(K128 key) { return fmix32(wfold32(key) ^ 0x165667B1u); }

// ------------------------------------------------------------------------------------------------
// extraction: the generic kernel of dbg_sk.h (window minimum by a loop over the w m-mer hashes of the tile) with
// 64-bit read-start windows and positions instead of bases in the records
// ------------------------------------------------------------------------------------------------
constexpr int WSK_NT = 1024;                        // threads of the extraction workgroup (its LDS allows one per CU)
constexpr int WSK_NH = TILE + WSK_HALO - 32;        // m-mer positions of a tile whose 32-base window is in the tile's LDS image
struct WSkLds {
    TileLdsW t;
    uint32_t ma[WSK_NH], mb[WSK_NH];  // sliding-window minimum: (16-bit m-mer hash << 16 | tile position), doubled spans
    uint8_t minp[TILE];             // minimizer offset (0..w-1) from the k-mer position, 0xFF = no k-mer
    unsigned long long sbits[TILE / 64 + 1];
    unsigned long long vbits[TILE / 64 + 1];
    uint32_t wpre[TILE / 64 + 1];
    uint16_t list[TILE];
    uint32_t nrec;
};

// Window minimum over w = k - 12 (20..51) m-mer hashes per k-mer position: a doubling table in LDS -- M1 = the hashes,
// M2[j] = min(M1[j], M1[j + 1]), M4[j] = min(M2[j], M2[j + 2]), ... up to the largest power of two p <= w, then
// min over [j, j + w) = min(Mp[j], Mp[j + w - p]).  Four or five passes of two LDS reads instead of w reads per position
// (the plain loop took 37 ms of the first version's 117 at k = 63).  Ties go to the leftmost position: the position
// is the low half of the packed value.
template <class ST>
__global__ __launch_bounds__(WSK_NT) void k_wsk_extract(const char *__restrict__ bases, uint64_t n_bytes,
                                                        const uint32_t *__restrict__ startbits, int k, uint64_t n_tiles,
                                                        uint64_t *rec_w0, uint64_t *rec_w1, ST *rec_st, uint64_t seg_cap,
                                                        uint64_t *seg_cnt, uint64_t *seg_nk, uint64_t *seg_ne,
                                                        unsigned long long *scalars /* [0] err */,
                                                        uint64_t tile_first /* the n_tiles tiles from here on (dbg_shard_extract_part) */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wsk_raw[];
    WSkLds &s = *reinterpret_cast<WSkLds *>(wsk_raw);
    __shared__ uint64_t red[2 * (WSK_NT / 64)];
    constexpr int m = SK_MAX_M;
    const int w = k - m + 1;
    const uint64_t t_beg = tile_first + n_tiles * blockIdx.x / gridDim.x, t_end = tile_first + n_tiles * (blockIdx.x + 1) / gridDim.x;
    const uint64_t seg0 = (uint64_t)blockIdx.x * seg_cap;
    uint64_t cursor = 0;
    const uint64_t mid_mask = (1ull << (k - 1)) - 1ull;
    uint64_t n_k = 0, n_e = 0;
    bool overflow = false;
    for (uint64_t tile = t_beg; tile < t_end; ++tile) {
        const uint64_t tile0 = tile * TILE;
        __syncthreads();
        const uint32_t bad = load_tile(s.t, bases, n_bytes, startbits, tile0);
        if (bad) atomicOr(&scalars[0], 1ull);
        __syncthreads();
        for (int j = threadIdx.x; j < WSK_NH; j += WSK_NT)
            s.ma[j] = (mmer_hash16((uint32_t)(window32(s.t, j) >> (64 - 2 * m))) << 16) | (uint32_t)j;
        __syncthreads();
        uint32_t *A = s.ma, *B = s.mb;
        int span = 1;
        while (span * 2 <= w) {  // uniform
            for (int j = threadIdx.x; j < WSK_NH; j += WSK_NT) B[j] = (j + span < WSK_NH) ? min(A[j], A[j + span]) : A[j];
            __syncthreads();
            uint32_t *tmp = A; A = B; B = tmp;
            span *= 2;
        }
        for (int j0 = 0; j0 < TILE; j0 += WSK_NT) {
            const int j = j0 + threadIdx.x;
            const uint64_t p = tile0 + j;
            bool v = false;
            uint32_t mp = 0xFFu;
            if (p < n_bytes) {
                const uint64_t sw = startwin64(s.t, j);
                const uint32_t s0 = (uint32_t)(sw & 1ull), sk = (uint32_t)(sw >> k) & 1u;
                v = (((sw >> 1) & mid_mask) == 0) && !(sk && s0) && (p + (uint64_t)k <= n_bytes);
                if (v) {
                    n_k += 1;
                    n_e += sk ^ 1u;
                    mp = (min(A[j], A[j + w - span]) & 0xFFFFu) - (uint32_t)j;
                }
            }
            s.minp[j] = (uint8_t)mp;
            const unsigned long long vb = __ballot(v);
            if ((threadIdx.x & 63) == 0) s.vbits[j >> 6] = vb;
        }
        __syncthreads();
        for (int j0 = 0; j0 < TILE; j0 += WSK_NT) {
            const int j = j0 + threadIdx.x;
            const uint32_t mp = s.minp[j];
            const bool st = (mp != 0xFFu) && (j == 0 || (uint32_t)s.minp[j - 1] != mp + 1);
            const unsigned long long sb = __ballot(st);
            if ((threadIdx.x & 63) == 0) s.sbits[j >> 6] = sb;
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            uint32_t a = __popcll(s.sbits[threadIdx.x]), b = __popcll(s.sbits[threadIdx.x + 64]);
            uint32_t ia = a, ib = b;
            for (int d = 1; d < 64; d <<= 1) {
                uint32_t oa = __shfl_up(ia, d, 64), ob = __shfl_up(ib, d, 64);
                if ((int)threadIdx.x >= d) { ia += oa; ib += ob; }
            }
            const uint32_t tot_a = __shfl(ia, 63, 64);
            s.wpre[threadIdx.x] = ia - a;
            s.wpre[threadIdx.x + 64] = tot_a + ib - b;
            if (threadIdx.x == 63) s.nrec = tot_a + ib;
            if (threadIdx.x == 0) { s.sbits[TILE / 64] = 0; s.vbits[TILE / 64] = 0; }
        }
        __syncthreads();
        const uint32_t nrec = s.nrec;
        if (cursor + nrec > seg_cap) { overflow = true; break; }
        const uint64_t gbase = seg0 + cursor;
        cursor += nrec;
        for (int j0 = 0; j0 < TILE; j0 += WSK_NT) {
            const int j = j0 + threadIdx.x;
            const unsigned long long sb = s.sbits[j >> 6];
            if ((sb >> (j & 63)) & 1ull) s.list[s.wpre[j >> 6] + __popcll(sb & ((1ull << (j & 63)) - 1))] = (uint16_t)j;
        }
        __syncthreads();
        for (uint32_t r = threadIdx.x; r < nrec; r += WSK_NT) {
            const int j = s.list[r];
            const int wd = j >> 6, bt = j & 63;
            const unsigned long long sb = s.sbits[wd];
            const unsigned long long nxt_s = (bt == 63) ? 0ull : (sb >> (bt + 1));
            const unsigned long long nxt_i = (bt == 63) ? 0ull : ((~s.vbits[wd]) >> (bt + 1));
            const int room = 63 - bt;
            const unsigned long long stop = nxt_s | nxt_i;
            int len;
            if (stop) {
                len = 1 + (__ffsll((unsigned long long)stop) - 1);
            } else {
                const unsigned long long stop2 = s.sbits[wd + 1] | ~s.vbits[wd + 1];
                len = 1 + room + (__ffsll((unsigned long long)stop2) - 1);
            }
            if (j + len > TILE) len = TILE - j;
            const uint64_t p = tile0 + j;
            const uint32_t s0 = (uint32_t)(startwin64(s.t, j) & 1ull);
            const uint32_t sk_last = (uint32_t)(startwin64(s.t, j + len - 1) >> k) & 1u;
            const uint32_t has_succ = sk_last ^ 1u;
            const uint32_t mp = j + s.minp[j];
            const uint32_t bh = bucket_hash22((uint32_t)(window32(s.t, mp) >> (64 - 2 * m)));
            const uint64_t o = gbase + r;
            rec_w0[o] = p;
            rec_w1[o] = ((uint64_t)(len - 1) << SK_META_BITS) | ((uint64_t)bh << 6) | has_succ;
            rec_st[o] = (ST)((p << 1) | (s0 ^ 1u));
        }
    }
    if (overflow && threadIdx.x == 0) atomicOr(&scalars[0], 4ull);
    n_k = wave_sum_u64(n_k);
    n_e = wave_sum_u64(n_e);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = n_k; red[WSK_NT / 64 + (threadIdx.x >> 6)] = n_e; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t tk = 0, te = 0;
        for (int i = 0; i < WSK_NT / 64; ++i) { tk += red[i]; te += red[WSK_NT / 64 + i]; }
        seg_cnt[blockIdx.x] = cursor;
        seg_nk[blockIdx.x] = tk;
        seg_ne[blockIdx.x] = te;
    }
}

// distinct / instances of the records of one segment (a level-1 bucket): 64-bit hashes of the k-mers in a global set
__global__ __launch_bounds__(256) void k_wsk_estimate(const uint64_t *__restrict__ b_start, const uint64_t *__restrict__ b_cnt,
                                                      uint32_t bucket, const uint64_t *__restrict__ rec_w0,
                                                      const uint64_t *__restrict__ rec_w1, int k,
                                                      const uint64_t *__restrict__ pk, unsigned long long *set, uint64_t set_mask,
                                                      unsigned long long *out /* [0] instances [1] distinct */) {
    const uint64_t beg = b_start[bucket], n = b_cnt[bucket];
    uint64_t inst = 0, fresh = 0;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t p = rec_w0[beg + r];
        const int len = wrec_len(rec_w1[beg + r]);
        for (int i = 0; i < len; ++i) {
            unsigned long long hv = k128_hash(packed_kmer(pk, p + i, k));
            if (hv == EMPTY_KEY) hv = 0;
            ++inst;
            uint64_t slot = hv & set_mask;
            for (uint64_t probe = 0; probe <= set_mask; ++probe) {
                unsigned long long cur = set[slot];
                if (cur == EMPTY_KEY) {
                    cur = atomicCAS(&set[slot], EMPTY_KEY, hv);
                    if (cur == EMPTY_KEY) { ++fresh; break; }
                }
                if (cur == hv) break;
                slot = (slot + 1) & set_mask;
            }
        }
    }
    inst = wave_sum_u64(inst);
    fresh = wave_sum_u64(fresh);
    if ((threadIdx.x & 63) == 0) {
        if (inst) atomicAdd(&out[0], (unsigned long long)inst);
        if (fresh) atomicAdd(&out[1], (unsigned long long)fresh);
    }
}

// Fast extraction for a compile-time window W = k - 12 (k = 63 -> W = 51, the width BASELINE.json configs[4] names):
// the register scheme of k_sk_extract_w (dbg_sk.h).  Every lane owns 32 consecutive positions, reads 96 bases and 96
// read-start bits once, rolls the 32 + W - 1 m-mer hashes it needs in registers and takes the window minimum by the van
// Herk / Gil-Werman block scheme -- about 3 min operations per position and three barriers less per tile than the
// doubling table of k_wsk_extract, which stays for the other widths.  Same hash, same leftmost tie-break
// (packed = hash16 << 8 | offset), same records.
struct WSkLdsW {
    TileLdsW t;
    uint8_t minp[TILE];
    unsigned long long sbits[TILE / 64 + 1];
    unsigned long long vbits[TILE / 64 + 1];
    uint32_t wpre[TILE / 64 + 1];
    uint16_t list[TILE];
    uint8_t edge[256];
    uint32_t nrec;
};

template <class ST, int W>
__global__ __launch_bounds__(256, 2) void k_wsk_extract_w(const char *__restrict__ bases, uint64_t n_bytes,
                                                          const uint32_t *__restrict__ startbits, uint64_t n_tiles,
                                                          uint64_t *rec_w0, uint64_t *rec_w1, ST *rec_st, uint64_t seg_cap,
                                                          uint64_t *seg_cnt, uint64_t *seg_nk, uint64_t *seg_ne,
                                                          unsigned long long *scalars /* [0] err */, uint64_t tile_first) {
    constexpr int M = SK_MAX_M, K = W + M - 1, NV = 32 + W - 1;
    static_assert(TILE == 256 * 32, "one lane per 32 positions");
    static_assert(K >= 32 && K <= 63 && NV + M - 1 <= 96 && 32 + K <= 96 && WSK_HALO >= 96, "window must fit three 32-base registers");
    __shared__ WSkLdsW s;
    __shared__ uint64_t red[8];
    const uint64_t t_beg = tile_first + n_tiles * blockIdx.x / gridDim.x, t_end = tile_first + n_tiles * (blockIdx.x + 1) / gridDim.x;
    const uint64_t seg0 = (uint64_t)blockIdx.x * seg_cap;
    uint64_t cursor = 0;
    constexpr uint64_t mid_mask = (1ull << (K - 1)) - 1ull;
    uint64_t n_k = 0, n_e = 0;
    bool overflow = false;
    const int j0 = threadIdx.x * 32;
    for (uint64_t tile = t_beg; tile < t_end; ++tile) {
        const uint64_t tile0 = tile * TILE;
        __syncthreads();
        const uint32_t bad = load_tile(s.t, bases, n_bytes, startbits, tile0);
        if (bad) atomicOr(&scalars[0], 1ull);
        __syncthreads();
        // ---- register phase
        const uint64_t wv[4] = {window32(s.t, j0), window32(s.t, j0 + 32), window32(s.t, j0 + 64), 0};
        uint32_t pv[NV], P[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int wi = i >> 5, sh = 2 * (i & 31);
            const uint64_t win = (sh == 0) ? wv[wi] : (wi < 2) ? ((wv[wi] << sh) | (wv[wi + 1] >> (64 - sh))) : (wv[wi] << sh);
            pv[i] = (mmer_hash16((uint32_t)(win >> (64 - 2 * M))) << 8) | (uint32_t)i;
            P[i] = (i % W == 0) ? pv[i] : min(P[i - 1], pv[i]);
        }
        const uint64_t sb_lo = ((uint64_t)s.t.sb[(j0 >> 5) + 1] << 32) | s.t.sb[j0 >> 5];
        const uint64_t sb_hi = s.t.sb[(j0 >> 5) + 2];
        // validity of the lane's 32 positions at once, on the 96 start bits (see k_sk_extract_w): no read start at
        // q + 1 .. q + K - 1, not both at q and q + K, and the k-mer ends inside the reads
        const Bits128 sbits{sb_lo, sb_hi};
        const uint32_t skmask = (uint32_t)shr128(sbits, K).lo;
        const uint64_t pos0 = tile0 + (uint64_t)j0;
        const uint64_t fit = n_bytes >= pos0 + (uint64_t)K ? n_bytes - (pos0 + (uint64_t)K) + 1 : 0;  // positions q < fit end inside
        const uint32_t inside = fit >= 32 ? 0xFFFFFFFFu : ((1u << (uint32_t)fit) - 1u);
        const uint32_t vmask = ~(uint32_t)window_or128_lo<K - 1>(shr128(sbits, 1)) & ~(skmask & (uint32_t)sb_lo) & inside;
        (void)mid_mask;
        n_k += __popc(vmask);
        n_e += __popc(vmask & ~skmask);
        uint32_t off[32];
        {
            uint32_t S = 0xFFFFFFFFu;
#pragma unroll
            for (int i = NV - 1; i >= 0; --i) {
                S = (i % W == W - 1 || i == NV - 1) ? pv[i] : min(S, pv[i]);
                if (i < 32) {
                    const uint32_t mn = (i % W == 0) ? P[i + W - 1] : min(S, P[i + W - 1]);
                    off[i] = ((vmask >> i) & 1u) ? ((mn & 0xFFu) - (uint32_t)i) : 0xFFu;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 32; q += 4)
            reinterpret_cast<uint32_t *>(s.minp)[(j0 + q) >> 2] = off[q] | (off[q + 1] << 8) | (off[q + 2] << 16) | (off[q + 3] << 24);
        s.edge[threadIdx.x] = (uint8_t)off[31];
        reinterpret_cast<uint32_t *>(s.vbits)[threadIdx.x] = vmask;
        __syncthreads();
        uint32_t smask = 0;
        {
            uint32_t prev = threadIdx.x ? (uint32_t)s.edge[threadIdx.x - 1] : 0xFFu;
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                const bool st = (off[q] != 0xFFu) && (prev == 0xFFu || prev != off[q] + 1);
                smask |= (uint32_t)st << q;
                prev = off[q];
            }
            reinterpret_cast<uint32_t *>(s.sbits)[threadIdx.x] = smask;
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            uint32_t a = __popcll(s.sbits[threadIdx.x]), b = __popcll(s.sbits[threadIdx.x + 64]);
            uint32_t ia = a, ib = b;
            for (int d = 1; d < 64; d <<= 1) {
                uint32_t oa = __shfl_up(ia, d, 64), ob = __shfl_up(ib, d, 64);
                if ((int)threadIdx.x >= d) { ia += oa; ib += ob; }
            }
            const uint32_t tot_a = __shfl(ia, 63, 64);
            s.wpre[threadIdx.x] = ia - a;
            s.wpre[threadIdx.x + 64] = tot_a + ib - b;
            if (threadIdx.x == 63) s.nrec = tot_a + ib;
            if (threadIdx.x == 0) { s.sbits[TILE / 64] = 0; s.vbits[TILE / 64] = 0; }
        }
        __syncthreads();
        const uint32_t nrec = s.nrec;
        if (cursor + nrec > seg_cap) { overflow = true; break; }
        const uint64_t gbase = seg0 + cursor;
        cursor += nrec;
        {
            uint32_t sm = smask;
            uint32_t li = s.wpre[threadIdx.x >> 1] + ((threadIdx.x & 1) ? __popc((uint32_t)s.sbits[threadIdx.x >> 1]) : 0);
            while (sm) {
                const int q = __ffs(sm) - 1;
                sm &= sm - 1;
                s.list[li++] = (uint16_t)(j0 + q);
            }
        }
        __syncthreads();
        for (uint32_t r = threadIdx.x; r < nrec; r += 256) {
            const int j = s.list[r];
            const int wd = j >> 6, bt = j & 63;
            const unsigned long long sb = s.sbits[wd];
            const unsigned long long nxt_s = (bt == 63) ? 0ull : (sb >> (bt + 1));
            const unsigned long long nxt_i = (bt == 63) ? 0ull : ((~s.vbits[wd]) >> (bt + 1));
            const int room = 63 - bt;
            const unsigned long long stop = nxt_s | nxt_i;
            int len;
            if (stop) {
                len = 1 + (__ffsll((unsigned long long)stop) - 1);
            } else {
                const unsigned long long stop2 = s.sbits[wd + 1] | ~s.vbits[wd + 1];
                len = 1 + room + (__ffsll((unsigned long long)stop2) - 1);
            }
            if (j + len > TILE) len = TILE - j;
            const uint64_t p = tile0 + j;
            const uint32_t s0 = (uint32_t)(startwin64(s.t, j) & 1ull);
            const uint32_t sk_last = (uint32_t)(startwin64(s.t, j + len - 1) >> K) & 1u;
            const uint32_t has_succ = sk_last ^ 1u;
            const uint32_t mp = j + s.minp[j];
            const uint32_t bh = bucket_hash22((uint32_t)(window32(s.t, mp) >> (64 - 2 * M)));
            const uint64_t o = gbase + r;
            rec_w0[o] = p;
            rec_w1[o] = ((uint64_t)(len - 1) << SK_META_BITS) | ((uint64_t)bh << 6) | has_succ;
            rec_st[o] = (ST)((p << 1) | (s0 ^ 1u));
        }
    }
    if (overflow && threadIdx.x == 0) atomicOr(&scalars[0], 4ull);
    n_k = wave_sum_u64(n_k);
    n_e = wave_sum_u64(n_e);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = n_k; red[4 + (threadIdx.x >> 6)] = n_e; }
    __syncthreads();
    if (threadIdx.x == 0) {
        seg_cnt[blockIdx.x] = cursor;
        seg_nk[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
        seg_ne[blockIdx.x] = red[4] + red[5] + red[6] + red[7];
    }
}

// The bases of every record, aligned (first base in bits 63:62 of word 0) and zero beyond the record, in record
// (= bucket) order: one pass of independent 40-byte reads of the packed reads at full occupancy.  Read inside the count
// kernel, where a bucket's ~140 records are all a workgroup has in flight, the same reads were a dependent round trip
// through HBM at the start of every bucket: 8 ms of 40 at k = 63.
__global__ __launch_bounds__(256) void k_wsk_gather(const uint64_t *__restrict__ rec_w0, const uint64_t *__restrict__ rec_w1,
                                                    uint64_t n_rec, const uint64_t *__restrict__ pk, int k, uint4 *out /* [n_rec][2] */) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rec) return;
    const uint64_t p = rec_w0[r], w1 = rec_w1[r];
    const int nb = k + wrec_len(w1) - 1 + (int)wrec_has_succ(w1);  // bases the record covers (<= 115)
    const uint64_t wi = p >> 5;
    const int sh = (int)(p & 31) * 2;
    uint64_t W[5], v[4];
#pragma unroll
    for (int j = 0; j < 5; ++j) W[j] = pk[wi + j];  // (the packed array is padded)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[j] = sh ? (W[j] << sh) | (W[j + 1] >> (64 - sh)) : W[j];
        const int have = nb - 32 * j;  // bases of this word that belong to the record
        if (have <= 0) v[j] = 0;
        else if (have < 32) v[j] &= ~0ull << (64 - 2 * have);
    }
    out[2 * r] = make_uint4((uint32_t)v[0], (uint32_t)(v[0] >> 32), (uint32_t)v[1], (uint32_t)(v[1] >> 32));
    out[2 * r + 1] = make_uint4((uint32_t)v[2], (uint32_t)(v[2] >> 32), (uint32_t)v[3], (uint32_t)(v[3] >> 32));
}

// ------------------------------------------------------------------------------------------------
// per-bucket counting in LDS
// ------------------------------------------------------------------------------------------------
#ifndef DBG_WCNT_NT
#define DBG_WCNT_NT 1024
#endif
constexpr int WCNT_NT = DBG_WCNT_NT;
// Staging sizes by stamp type.  32-bit stamps (a single GPU's reads below 2 GiB): 288 records per round -- a bucket holds
// 140 +- 64 at the default geometry, and a second round pays the dedupe / quad list / insert barriers again.  64-bit
// stamps (sharded builds: global positions): the stamp array takes 16 KB more of the 160, so 128 records and fewer
// staged queries.
template <class ST>
struct WCntCfg {
    static constexpr int STAGE = sizeof(ST) == 8 ? 128 : 288;
#ifdef DBG_CNT_PROF
    static constexpr int QBUF = sizeof(ST) == 8 ? 128 : 256;  // (room for the clocks of the experiment build)
#else
    static constexpr int QBUF = sizeof(ST) == 8 ? 160 : 320;  // cross-bucket successor queries staged per bucket
#endif
    static constexpr int DD = sizeof(ST) == 8 ? 256 : 1024;    // dedupe set slots (>= 2 * STAGE)
};
constexpr int WCNT_QMAX = 13;        // quads of 4 k-mers per record: ceil(51 / 4)
constexpr unsigned long long W_PEND = 1ull << 63;

template <class ST>
struct WCntLds {
    static constexpr int WCNT_STAGE = WCntCfg<ST>::STAGE, WCNT_QBUF = WCntCfg<ST>::QBUF, WCNT_DD = WCntCfg<ST>::DD;
    unsigned long long khi[WCAP];   // EMPTY_KEY / hi | W_PEND / hi   (hi < 2^62)
    unsigned long long klo[WCAP];
    uint32_t cnt2[WCAP * 2];        // four 16-bit successor counters per slot: codes 0, 1 in dword 0; 2, 3 in dword 1
    ST stamp[WCAP];
    uint16_t list[WCAP];            // local node index -> slot
    uint16_t eoff[WCAP];            // local node index -> first CSR edge of the node, relative to the bucket
    // staged records (bases aligned: first base in bits 63:62 of word 0, zero beyond the record)
    unsigned long long rb[WCNT_STAGE][4];
    unsigned long long rmeta[WCNT_STAGE];
    ST rst[WCNT_STAGE];
    uint32_t dd_tab[WCNT_DD];
    uint32_t dd_mult[WCNT_STAGE];
    uint16_t flat[WCNT_STAGE * WCNT_QMAX];
    unsigned long long q_lo[WCNT_QBUF], q_hi[WCNT_QBUF];
    uint32_t q_off[WCNT_QBUF];
    uint32_t stk_mask[CNT_STACK], stk_val[CNT_STACK];
    uint32_t overflow, n_local, n_q, fail, n_flat;
    unsigned long long gbase, qbase, ebase, ri;
    unsigned long long dir_mask[WCAP / 64];
    uint16_t dir_base[WCAP / 64];
#ifdef DBG_CNT_PROF
    unsigned long long prof[64];
#endif
};

struct WSkCountOut {
    uint64_t *keys, *keys_hi;
    void *stamps;              // ST[node_cap]
    uint8_t *flags;            // (stamp & 1) | present-base mask << 1
    uint64_t node_cap;
    uint32_t *rowptr, *col, *ecnt;
    uint64_t edge_cap;
    uint64_t *q_lo, *q_hi;     // cross-bucket successors: the successor k-mer ...
    uint32_t *q_col;           // ... and the CSR position to patch
    uint64_t q_cap;
    SkRange *ranges;
    uint64_t n_buckets, range_cap;
    SkDirEnt *dirs;
    uint64_t own_lo, own_cnt;
    unsigned long long *scalars;  // [0] err [4] nodes | edges << 32 [5] queries [6] extra ranges
    uint32_t id_tag;           // OR-ed into every successor id written (sharded builds with tagged ids: owner << 29)
};

typedef const WSkCountOut __attribute__((address_space(4))) *WSkOutConstPtr;
__device__ inline WSkOutConstPtr wfresh_args(const WSkCountOut *p) {  // see fresh_args (dbg_sk.h)
    unsigned long long v = (unsigned long long)p;
    asm volatile("" : "+s"(v));
    return (WSkOutConstPtr)v;
}

__device__ inline void wcnt_load(const uint32_t *cnt2, uint32_t slot, uint32_t c[4]) {
    const uint2 v = reinterpret_cast<const uint2 *>(cnt2)[slot];
    c[0] = v.x & 0xFFFFu; c[1] = v.x >> 16; c[2] = v.y & 0xFFFFu; c[3] = v.y >> 16;
}

// LDS accesses of the claim protocol: relaxed workgroup-scope atomics (plain ds_read / ds_write, never cached in a
// register) with a compiler-only fence after each -- the hardware order is the issue order, see the insert loop of k_wsk_count
__device__ inline void wlds_store(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
}
__device__ inline unsigned long long wlds_load(const unsigned long long *p) {
    const unsigned long long v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    return v;
}

__device__ inline int wlds_find(const unsigned long long *khi, const unsigned long long *klo, K128 key) {
    uint32_t slot = wslot_of(key);
    for (int probe = 0; probe < WCAP; ++probe) {
        const unsigned long long cur = khi[slot];
        if (cur == EMPTY_KEY) return -1;
        if (cur == key.hi && klo[slot] == key.lo) return (int)slot;
        slot = (slot + 1) & (WCAP - 1);
    }
    return -1;
}

template <class ST>
__global__ __launch_bounds__(WCNT_NT) void k_wsk_count(const uint64_t *__restrict__ b_start, const uint64_t *__restrict__ b_cnt,
                                                      const uint64_t *__restrict__ rec_w0, const uint64_t *__restrict__ rec_w1,
                                                      const ST *__restrict__ rec_st, const uint4 *__restrict__ rec_b /* k_wsk_gather */,
                                                      int k, uint64_t n_buckets, const WSkCountOut *__restrict__ outp /* in device memory: see fresh_args, dbg_sk.h */,
                                                      uint32_t split_recs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wcnt_raw[];
    WCntLds<ST> &s = *reinterpret_cast<WCntLds<ST> *>(wcnt_raw);
    constexpr int NPT = WCAP / WCNT_NT;
    constexpr int WCNT_STAGE = WCntCfg<ST>::STAGE, WCNT_QBUF = WCntCfg<ST>::QBUF, WCNT_DD = WCntCfg<ST>::DD;
    static_assert(WCNT_DD >= 2 * WCNT_STAGE && WCNT_STAGE <= WCNT_NT, "staging");
    static_assert(sizeof(WCntLds<ST>) <= 160 * 1024, "LDS");
    bool clean = false;
    if (threadIdx.x == 0) s.fail = 0;
#ifdef DBG_CNT_PROF
    unsigned long long clast_ = clock64();
    if (threadIdx.x < 64) s.prof[threadIdx.x] = 0;
    __syncthreads();
#endif
    // The next bucket's records are fetched into registers while this bucket is in its later phases, its record range
    // one bucket earlier still (as in k_sk_count).
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
        if (threadIdx.x < min(nx_n, (uint64_t)WCNT_STAGE)) {
            pf_w1 = rec_w1[nx_beg + threadIdx.x];
            pf_st = rec_st[nx_beg + threadIdx.x];
            pf_b0 = rec_b[2 * (nx_beg + threadIdx.x)];
            pf_b1 = rec_b[2 * (nx_beg + threadIdx.x) + 1];
        }
        load_range(b + gridDim.x, r2_beg, r2_n);
    };
    // The first round of the NEXT bucket goes from the prefetch registers into the staging arrays at the start of this
    // bucket's node write (the arrays are idle after the insert): staged at the top of the next bucket, the wait for the
    // registers also waited for every global store of the node write before it (one counter for loads and stores).
    auto stage_regs = [&](uint32_t n_st) {
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
        for (uint32_t i = threadIdx.x; i < WCNT_DD; i += WCNT_NT) s.dd_tab[i] = 0xFFFFFFFFu;
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
        const bool small_bucket = r_n * 51 < 0xFFFFull;  // a record holds at most 51 k-mers
        uint32_t stk_n = 1;
        bool root = true, failed = false;
        if (split_recs && r_n > split_recs) {
            uint32_t parts = 2;
            while (parts < 16 && (uint64_t)parts * split_recs < r_n) parts <<= 1;
            __syncthreads();
            if (threadIdx.x < parts) { s.stk_mask[threadIdx.x] = parts - 1; s.stk_val[threadIdx.x] = threadIdx.x; }
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
            __syncthreads();
            if (!clean) {
                for (int i = threadIdx.x; i < WCAP; i += WCNT_NT) {
                    s.khi[i] = EMPTY_KEY;
                    s.stamp[i] = (ST)~(ST)0;
                    reinterpret_cast<uint2 *>(s.cnt2)[i] = make_uint2(0, 0);
                }
            }
            clean = false;
            if (threadIdx.x == 0) { s.overflow = 0; s.n_local = 0; s.n_q = 0; }
            CNT_TICK(13);
            // ---- insert
            for (uint64_t c0 = 0; c0 < r_n; c0 += WCNT_STAGE) {
                const uint32_t n_st = (uint32_t)min((uint64_t)WCNT_STAGE, r_n - c0);
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
                        for (uint32_t i = threadIdx.x; i < WCNT_DD; i += WCNT_NT) s.dd_tab[i] = 0xFFFFFFFFu;
                    }
                    CNT_TICK(15);
                    __syncthreads();
                    CNT_TICK(1);
                    if (c0 && s.overflow) break;  // an earlier round of this pass overflowed the table
                }
                lds_staged = false;
                // ---- identical records collapse to one representative with a multiplicity and the smallest stamp; the
                //      representatives list their quads of 4 k-mers in the same pass (a record knows that it is one when
                //      its own claim succeeds)
                {
                    uint32_t nquad = 0;
                    const uint32_t r = threadIdx.x;
                    if (r < n_st) {
                        const unsigned long long b0 = s.rb[r][0], b1 = s.rb[r][1], b2 = s.rb[r][2], b3 = s.rb[r][3];
                        const unsigned long long mt = s.rmeta[r] & ~(((1ull << SK_BUCKET_BITS) - 1) << 6);  // length + flag
                        uint32_t hslot = (uint32_t)(mix64(b0 ^ mix64(b1 + 0x9E3779B97F4A7C15ull) ^ mix64(b2 ^ (b3 * 0xD6E8FEB86659FD93ull)) ^ mt) >> 40) &
                                         (WCNT_DD - 1);
                        uint32_t rep = r;
                        for (uint32_t probe = 0; probe < WCNT_DD; ++probe) {
                            uint32_t cur = s.dd_tab[hslot];
                            if (cur == 0xFFFFFFFFu) {
                                cur = atomicCAS(&s.dd_tab[hslot], 0xFFFFFFFFu, r);
                                if (cur == 0xFFFFFFFFu) break;
                            }
                            if (s.rb[cur][0] == b0 && s.rb[cur][1] == b1 && s.rb[cur][2] == b2 && s.rb[cur][3] == b3 &&
                                (s.rmeta[cur] & ~(((1ull << SK_BUCKET_BITS) - 1) << 6)) == mt) { rep = cur; break; }
                            hslot = (hslot + 1) & (WCNT_DD - 1);
                        }
                        atomicAdd(&s.dd_mult[rep], 1u);
                        if (rep != r) atomicMin(&s.rst[rep], s.rst[r]);
                        else nquad = ((uint32_t)wrec_len(mt) + 3) >> 2;
                    }
                    const uint32_t base = wave_alloc_n<WCNT_QMAX>(&s.n_flat, nquad);
                    for (uint32_t q = 0; q < nquad; ++q) s.flat[base + q] = (uint16_t)((r << 4) | q);
                }
                __syncthreads();
                CNT_TICK(3);
                const uint32_t n_flat = s.n_flat;
#ifdef DBG_CNT_PROF
                const unsigned long long wt0_ = clock64();
#endif
                for (uint32_t f = threadIdx.x >> 2; f < n_flat; f += WCNT_NT / 4) {
                    const uint32_t e = s.flat[f];
                    const uint32_t r = e >> 4;
                    const int i = (int)((e & 15) * 4 + (threadIdx.x & 3));
                    const unsigned long long w1 = s.rmeta[r];
                    const int len = wrec_len(w1);
                    if (i >= len) continue;
                    const int wi = i >> 5, sh = (i & 31) * 2;
                    const unsigned long long a0 = s.rb[r][wi], a1 = s.rb[r][wi + 1], a2 = s.rb[r][wi + 2];
                    const uint64_t A = sh ? (a0 << sh) | (a1 >> (64 - sh)) : a0;
                    const uint64_t B = sh ? (a1 << sh) | (a2 >> (64 - sh)) : a1;
                    const K128 key = k128_from_windows(A, B, k);
                    if (cur_mask && (wsub_hash(key) & cur_mask) != cur_val) continue;
                    const bool has_succ = (i < len - 1) || wrec_has_succ(w1);
                    const uint32_t b = base_after_kmer(B, k);
                    const ST st0 = s.rst[r];
                    const ST stamp = i ? (ST)((st0 | (ST)1) + (ST)(2 * i)) : st0;
                    const uint32_t mult = s.dd_mult[r];
                    uint32_t slot = wslot_of(key);
                    bool ok = false;
                    // Claim first: most k-mers a bucket sees are new (identical records were collapsed), and one
                    // compare-and-swap answers all three cases -- empty (now ours), this k-mer's high word, another's.
                    // The LDS executes one wave's operations in order, so the low word is visible before the final
                    // high word without a wait between the two stores; a reader that sees the final high word reads
                    // the low word afterwards (its address depends on nothing, its issue on the compare).
                    // A pending high word equal to ours is looked at again in the NEXT iteration of this loop, never in
                    // a loop of its own: the claimant may be a lane of this wave, and it publishes in its own branch of
                    // the iteration.
                    for (int probe = 0, spins = 0; probe < CNT_PROBE_LIMIT && spins < (1 << 16); ) {
                        const unsigned long long cur = atomicCAS(&s.khi[slot], EMPTY_KEY, key.hi | W_PEND);
                        if (cur == EMPTY_KEY) {
                            wlds_store(&s.klo[slot], key.lo);
                            wlds_store(&s.khi[slot], key.hi);
                            ok = true;
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
                    if (!ok) { s.overflow = 1; continue; }
                    if (has_succ) {
                        const int shf = 16 * (int)(b & 1);
                        if (small_bucket) {  // uniform: fewer than 2^16 k-mer instances in the whole bucket -- no counter can wrap,
                            atomicAdd(&s.cnt2[slot * 2 + (b >> 1)], mult << shf);  // and nobody waits for the old value
                        } else {
                            const uint32_t old = atomicAdd(&s.cnt2[slot * 2 + (b >> 1)], mult << shf);
                            if (((old >> shf) & 0xFFFFu) + mult > 0xFFFFu) atomicOr(&wfresh_args(outp)->scalars[0], 512ull);  // 16-bit counter overflow
                        }
                    }
                    atomicMin(&s.stamp[slot], stamp);
                }
#ifdef DBG_CNT_PROF
                if ((threadIdx.x & 63) == 0) s.prof[32 + (threadIdx.x >> 6)] += clock64() - wt0_;
                if (threadIdx.x == 0) { s.prof[48] += n_flat; s.prof[49] += n_st; }
#endif
            }
            CNT_TICK(4);
            __syncthreads();
            CNT_TICK(5);
            if (have_pf) {  // the registers are free: the next bucket's records, under the rest of this one
                have_pf = false;
                prefetch_recs(bucket + gridDim.x);
            }
            if (s.overflow) {  // split this hash sub-range in two and retry (nothing was written out)
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
            // ---- dense list of occupied slots + CSR edge offsets: a wave takes its four 64-slot blocks together (reads
            //      back to back, ONE packed LDS atomic for all of them), as in k_sk_count
            {
                constexpr int NB = WCAP / WCNT_NT;
                unsigned long long kk[NB];
                uint2 cc[NB];
#pragma unroll
                for (int t = 0; t < NB; ++t) {
                    const int i = threadIdx.x + t * WCNT_NT;
                    kk[t] = s.khi[i];
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
                if (tot && (threadIdx.x & 63) == 0) base = atomicAdd(&s.n_local, tot);
                base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
                for (int t = 0; t < NB; ++t) {
                    const int i = threadIdx.x + t * WCNT_NT;
                    if ((threadIdx.x & 63) == 0) { s.dir_mask[i >> 6] = mask[t]; s.dir_base[i >> 6] = (uint16_t)base; }
                    if (kk[t] != EMPTY_KEY) {
                        const uint32_t li = (base & 0xFFFFu) + below[t];
                        s.list[li] = (uint16_t)i;
                        s.eoff[li] = (uint16_t)((base >> 16) + eexc[t]);
                    }
                    base += nn[t] | (ne[t] << 16);
                }
            }
            CNT_TICK(6);
            __syncthreads();
            CNT_TICK(7);
            const uint32_t n_local = s.n_local & 0xFFFFu, n_edges_local = s.n_local >> 16;
            unsigned long long got = 0;
            if (threadIdx.x == 0)
                got = atomicAdd(&wfresh_args(outp)->scalars[4], (unsigned long long)n_local | ((unsigned long long)n_edges_local << 32));
            // ---- successor lookups into registers; misses are staged as queries
            unsigned long long nsucc[NPT];
#pragma unroll
            for (int u = 0; u < NPT; ++u) {
                nsucc[u] = ~0ull;
                if ((uint32_t)(u * WCNT_NT) >= n_local) continue;
                const uint32_t li = threadIdx.x + u * WCNT_NT;
                K128 key{0, 0};
                uint32_t nz = 0;
                if (li < n_local) {
                    const uint32_t sl = s.list[li];
                    key.hi = s.khi[sl];
                    key.lo = s.klo[sl];
                    uint32_t c[4];
                    wcnt_load(s.cnt2, sl, c);
                    nz = (c[0] != 0) | ((c[1] != 0) << 1) | ((c[2] != 0) << 2) | ((c[3] != 0) << 3);
                }
                const uint32_t nz_all = nz;
                uint32_t missmask = 0;
                while (nz) {
                    const uint32_t b = __ffs(nz) - 1;
                    nz &= nz - 1;
                    const int f = wlds_find(s.khi, s.klo, k128_append(key, b, k));
                    if (f >= 0) {
                        const uint32_t ix = (uint32_t)s.dir_base[f >> 6] +
                                            (uint32_t)__popcll(s.dir_mask[f >> 6] & ((1ull << (f & 63)) - 1ull));
                        nsucc[u] = (nsucc[u] & ~(0xFFFFull << (16 * b))) | ((unsigned long long)ix << (16 * b));
                    } else {
                        missmask |= 1u << b;
                    }
                }
                uint32_t qi = wave_alloc_n<4>(&s.n_q, (uint32_t)__popc(missmask));
                while (missmask) {
                    const uint32_t b = __ffs(missmask) - 1;
                    missmask &= missmask - 1;
                    unsigned long long code;
                    if (qi < WCNT_QBUF) {
                        const K128 sk = k128_append(key, b, k);
                        s.q_lo[qi] = sk.lo;
                        s.q_hi[qi] = sk.hi;
                        s.q_off[qi] = s.eoff[li] + __popc(nz_all & ((1u << b) - 1u));
                        code = 0xFFFEull;
                    } else {
                        code = 0x8000ull | (qi - WCNT_QBUF);
                    }
                    nsucc[u] = (nsucc[u] & ~(0xFFFFull << (16 * b))) | (code << (16 * b));
                    ++qi;
                }
            }
            CNT_TICK(8);
            if (threadIdx.x == 0) {
                const auto &orr = *wfresh_args(outp);
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
            CNT_TICK(9);
            __syncthreads();
            CNT_TICK(10);
            const uint32_t nq = s.n_q;
            unsigned long long qgot = 0;
            if (threadIdx.x == 64 && nq) qgot = atomicAdd(&wfresh_args(outp)->scalars[5], (unsigned long long)nq);
            if (s.fail) break;
            const uint64_t gbase = s.gbase, ebase = s.ebase;
            if (stk_n == 0 && !have_pf) {  // uniform: last pass of this bucket, the registers hold the next one's first round
                stage_regs((uint32_t)min(nx_n, (uint64_t)WCNT_STAGE));
                staged = true;
            }
            // ---- write nodes and their CSR rows; every slot read is cleared for the next bucket
            const auto &ow = *wfresh_args(outp);  // loaded here, not kept in SGPRs across the whole bucket loop
            if (threadIdx.x < WCAP / 64) {
                SkDirEnt de;
                de.mask = s.dir_mask[threadIdx.x];
                de.base = (uint32_t)(gbase + s.dir_base[threadIdx.x]);
                de.pad = s.ri < ow.n_buckets ? 1u : 0u;
                const uint64_t di = s.ri < ow.n_buckets ? s.ri - ow.own_lo : ow.own_cnt + (s.ri - ow.n_buckets);
                ow.dirs[di * (WCAP / 64) + threadIdx.x] = de;
            }
#pragma unroll
            for (int u = 0; u < NPT; ++u) {
                if ((uint32_t)(u * WCNT_NT) >= n_local) break;
                const uint32_t li = threadIdx.x + u * WCNT_NT;
                if (li >= n_local) continue;
                const uint32_t i = s.list[li];
                const unsigned long long khi = s.khi[i], klo = s.klo[i];
                const uint64_t node = gbase + li;
                uint32_t c[4];
                wcnt_load(s.cnt2, i, c);
                const ST stamp = s.stamp[i];
                s.khi[i] = EMPTY_KEY;
                s.stamp[i] = (ST)~(ST)0;
                reinterpret_cast<uint2 *>(s.cnt2)[i] = make_uint2(0, 0);
                ow.keys[node] = klo;
                ow.keys_hi[node] = khi;
                reinterpret_cast<ST *>(ow.stamps)[node] = stamp;
                ow.flags[node] = (uint8_t)((uint32_t)(stamp & 1) | ((c[0] != 0) << 1) | ((c[1] != 0) << 2) | ((c[2] != 0) << 3) |
                                            ((c[3] != 0) << 4));
                uint64_t e = ebase + s.eoff[li];
                ow.rowptr[node] = (uint32_t)e;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (c[b]) {
                        const uint32_t v = (uint32_t)(nsucc[u] >> (16 * b)) & 0xFFFFu;
                        ow.col[e] = v < 0x8000u ? ((uint32_t)(gbase + v) | ow.id_tag) : NO_NODE;
                        ow.ecnt[e] = c[b];
                        ++e;
                    }
                }
            }
            CNT_TICK(11);
            const auto &oq = *wfresh_args(outp);
            if (threadIdx.x == 64 && nq) {
                s.qbase = qgot;
                if (qgot + nq > oq.q_cap || qgot + nq > 0xFFFFFFF0ull) { atomicOr(&oq.scalars[0], 64ull); s.fail = 1; }
            }
            __syncthreads();
            CNT_TICK(12);
            if (s.fail) break;
            if (nq) {
                const uint64_t qbase = s.qbase;
                for (uint32_t i = threadIdx.x; i < min(nq, (uint32_t)WCNT_QBUF); i += WCNT_NT) {
                    oq.q_lo[qbase + i] = s.q_lo[i];
                    oq.q_hi[qbase + i] = s.q_hi[i];
                    oq.q_col[qbase + i] = (uint32_t)(ebase + s.q_off[i]);
                }
                if (nq > (uint32_t)WCNT_QBUF) {  // rare: queries that did not fit the staging, straight from the registers
#pragma unroll
                    for (int u = 0; u < NPT; ++u) {
                        const uint32_t li = threadIdx.x + u * WCNT_NT;
                        if (li >= n_local) continue;
                        uint32_t rank = 0;
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            const uint32_t v = (uint32_t)(nsucc[u] >> (16 * b)) & 0xFFFFu;
                            if (v >= 0x8000u && v < 0xFFFEu) {
                                const uint64_t qi = (uint64_t)(v & 0x7FFFu) + WCNT_QBUF;
                                const uint64_t node = gbase + li;
                                const K128 sk = k128_append(K128{oq.keys_hi[node], oq.keys[node]}, (uint32_t)b, k);
                                oq.q_lo[qbase + qi] = sk.lo;
                                oq.q_hi[qbase + qi] = sk.hi;
                                oq.q_col[qbase + qi] = (uint32_t)(ebase + s.eoff[li] + rank);
                            }
                            if (v != 0xFFFFu) ++rank;
                        }
                    }
                }
            }
            clean = true;
        }
        if (failed || s.fail) return;
        if (have_pf) prefetch_recs(bucket + gridDim.x);
    }
#ifdef DBG_CNT_PROF
    if (threadIdx.x == 0) {
        for (int i = 0; i < 64; ++i) if (i != 31) atomicAdd(&g_cnt_prof[i], s.prof[i]);
        atomicAdd(&g_cnt_prof[31], 1ull);
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// successors that live in another bucket, through the directory (k_succ_resolve with two-word keys)
// ------------------------------------------------------------------------------------------------
__device__ inline uint32_t wdir_find(const SkDirEnt *__restrict__ dirs, uint64_t ri, const uint64_t *__restrict__ keys,
                                     const uint64_t *__restrict__ keys_hi, uint64_t n_nodes, K128 key, bool *whole = nullptr) {
    constexpr int NBLK = WCAP / 64;
    uint32_t slot = wslot_of(key);
    for (int blocks = 0; blocks <= NBLK; ++blocks) {
        const SkDirEnt de = dirs[ri * NBLK + (slot >> 6)];
        if (whole && blocks == 0 && de.pad != 1u) { *whole = false; return NO_NODE; }
        const int bit = (int)(slot & 63);
        const unsigned long long run_bits = de.mask >> bit;
        if (!(run_bits & 1ull)) return NO_NODE;
        const int avail = 64 - bit;
        const int run = (~run_bits) ? min(avail, __ffsll((unsigned long long)~run_bits) - 1) : avail;
        const uint64_t idx = (uint64_t)de.base + (uint64_t)__popcll(de.mask & ((1ull << bit) - 1ull));
        if (idx + run > n_nodes) return NO_NODE;
        for (int t = 0; t < run; ++t)
            if (keys[idx + t] == key.lo && keys_hi[idx + t] == key.hi) return (uint32_t)(idx + t);
        if (run < avail) return NO_NODE;
        slot = (slot + (uint32_t)run) & (WCAP - 1);
    }
    return NO_NODE;
}

// sort key of the owner split of a sharded build: bucket hash of every query in bits 40.. of q_meta, its index below
__global__ __launch_bounds__(256) void k_wq_bucket(const uint64_t *__restrict__ q_lo, const uint64_t *__restrict__ q_hi,
                                                   uint64_t *q_meta, uint64_t n, int k, int m) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) q_meta[i] = ((uint64_t)wkmer_bucket22(K128{q_hi[i], q_lo[i]}, k, m) << 40) | i;
}

// the queries grouped by owner, as they travel: (lo, hi) pairs; the high words come through the index the sort kept
__global__ __launch_bounds__(256) void k_wq_park(uint64_t n, const uint64_t *__restrict__ lo_sorted,
                                                 const uint64_t *__restrict__ meta_sorted, const uint64_t *__restrict__ hi_src,
                                                 uint64_t *pairs) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    pairs[2 * i] = lo_sorted[i];
    pairs[2 * i + 1] = hi_src[meta_sorted[i] & ((1ull << 40) - 1)];
}

// q_lo / q_hi: query i at [i * stride] (1: two arrays; 2: (lo, hi) pairs, q_hi = q_lo + 1).  id_tag is OR-ed into the ids.
__global__ __launch_bounds__(256) void k_wsucc_resolve(const uint64_t *__restrict__ q_lo, const uint64_t *__restrict__ q_hi,
                                                       int stride, const uint32_t *__restrict__ q_col, uint64_t n, SkGeom g,
                                                       const SkRange *__restrict__ ranges, uint64_t n_buckets, uint64_t n_ranges,
                                                       const SkDirEnt *__restrict__ dirs, const uint64_t *__restrict__ keys,
                                                       const uint64_t *__restrict__ keys_hi, uint64_t n_nodes, uint32_t *out,
                                                       uint32_t id_tag, unsigned long long *scalars) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const K128 key{q_hi[i * stride], q_lo[i * stride]};
    const uint64_t bucket = sk_bucket_of(wkmer_bucket22(key, g.k, g.m), g);
    uint32_t id = NO_NODE;
    if (bucket >= g.own_lo && bucket < g.own_lo + g.own_cnt) {
        bool whole = true;
        id = wdir_find(dirs, bucket - g.own_lo, keys, keys_hi, n_nodes, key, &whole);
        if (!whole) {
            const uint32_t sh = wsub_hash(key);
            uint32_t r = ranges[bucket].next;
            uint64_t ri = 0;
            bool have = false;
            for (int guard = 0; r && r < n_ranges && !have && guard < (1 << 20); ++guard) {
                const SkRange rg = ranges[r];
                if (rg.node_cnt && (sh & rg.mask) == rg.val) { ri = r; have = true; }
                r = rg.next;
            }
            if (have) id = wdir_find(dirs, g.own_cnt + (ri - n_buckets), keys, keys_hi, n_nodes, key);
        }
    }
    if (id == NO_NODE) { atomicOr(&scalars[0], 128ull); return; }  // every successor k-mer exists as a node of its owner
    out[q_col ? q_col[i] : i] = id | id_tag;
}

static_assert(sizeof(WCntLds<uint64_t>) <= 160 * 1024 && sizeof(WCntLds<uint32_t>) <= 160 * 1024, "LDS of the two-word count kernel");

}  // namespace dbgk
