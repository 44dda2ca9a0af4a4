// Super-k-mer engine: the HBM-lean build path (included by dbg_hip.hip; one translation unit).
//
//   k_sk_extract   reads -> fixed 16-byte super-k-mer records (+ stamp), one per run of
//                  consecutive k-mers that share a minimizer           [debruijn.py:123-128 windows]
//   k_ms_hist / k_ms_scatter   two-level LDS multisplit of the records by minimizer-hash bucket
//   k_sk_count     one workgroup per bucket: expand records, LDS open-address table with
//                  successor counters + first-occurrence stamp, compaction to the node arrays,
//                  in-bucket successor lookup                           [debruijn.py:129-143, :213-222]
//   k_q_bucket / k_q_answer    the successors that live in another bucket (about one per record)
//
// No global atomics per k-mer: they run at ~2-3e10/s chip-wide (memory-side), far below what
// this path needs; all per-k-mer atomics are LDS atomics.
#pragma once
#include <cstddef>
#include "dbg_device.h"

namespace dbgk {

// ---- record layout -----------------------------------------------------------------------------
// w0: bases 0..31 (first base in bits 63:62)
// w1: bits 63..28 bases 32..49 | bits 27..6 bucket hash (22 bits) | bits 5..1 len-1 | bit 0 last k-mer has a successor
// st: stamp of the first k-mer; k-mer i > 0 of the record has stamp (st | 1) + 2 i
constexpr int SK_MAX_M = 13;        // minimizer length for k >= 13 (else m = k)
constexpr int SK_BUCKET_BITS = 22;  // bucket hash bits carried in w1
constexpr int SK_META_BITS = 28;

__host__ __device__ inline int sk_m_for_k(int k) { return k < SK_MAX_M ? k : SK_MAX_M; }

__host__ __device__ inline uint32_t fmix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu;
    x ^= x >> 13; x *= 0xC2B2AE35u;
    x ^= x >> 16;
    return x;
}
// m <= 13: an m-mer is at most 26 bits.  Minimizer order = 16-bit hash, ties to the leftmost.
// one multiply: the minimizer order only has to look random, and this hash runs ~1.6x per base
__device__ inline uint32_t mmer_hash16(uint32_t mmer) { return ((mmer ^ (mmer >> 9) ^ 0x3C6EF372u) * 0x9E3779B1u) >> 16; }
__device__ inline uint32_t bucket_hash22(uint32_t mmer) { return fmix32(mmer * 0x9E3779B1u + 0x7F4A7C15u) >> (32 - SK_BUCKET_BITS); }

// minimizer-hash bucket of a single packed k-mer (pure function of the k-mer)
__device__ inline uint32_t kmer_bucket22(uint64_t kmer, int k, int m) {
    const int w = k - m + 1;
    const uint32_t mmask = (uint32_t)((1ull << (2 * m)) - 1);
    uint32_t best = 0xFFFFFFFFu, best_mm = 0;
    for (int i = 0; i < w; ++i) {
        const uint32_t mm = (uint32_t)(kmer >> (2 * (k - m - i))) & mmask;
        const uint32_t hv = mmer_hash16(mm);
        if (hv < best) { best = hv; best_mm = mm; }
    }
    return bucket_hash22(best_mm);
}

// ------------------------------------------------------------------------------------------------
// K1: extraction.  One tile of TILE positions per workgroup.
// ------------------------------------------------------------------------------------------------
struct SkLds {
    TileLds t;
    uint16_t hm[TILE + HALO];       // 16-bit m-mer hash per position
    uint8_t minp[TILE];             // minimizer offset (0..w-1) from the k-mer position, 0xFF = no k-mer
    unsigned long long sbits[TILE / 64 + 1];  // record starts
    unsigned long long vbits[TILE / 64 + 1];  // valid k-mer positions
    uint32_t wpre[TILE / 64 + 1];   // exclusive prefix of popcount(sbits)
    uint16_t list[TILE];            // record index -> start position (dense work list)
    uint32_t nrec;
};

// Persistent workgroups: workgroup g owns the tiles [g*T/G, (g+1)*T/G) and writes its records to its
// own segment [g*seg_cap, ...) of the record arrays -- no global atomics (a shared cursor would put
// one same-address atomic per tile on the critical path, ~12 ns each, serialised chip-wide).
template <class ST>
__global__ __launch_bounds__(256) void k_sk_extract(const char *__restrict__ bases, uint64_t n_bytes,
                                                    const uint32_t *__restrict__ startbits, int k, int m,
                                                    uint64_t n_tiles, uint64_t *rec_w0, uint64_t *rec_w1, ST *rec_st,
                                                    uint64_t seg_cap, uint64_t *seg_cnt, uint64_t *seg_nk,
                                                    uint64_t *seg_ne, unsigned long long *scalars /* [0] err */,
                                                    uint64_t tile_first /* the n_tiles tiles from here on (dbg_shard_extract_part) */) {
    __shared__ SkLds s;
    __shared__ uint64_t red[8];
    const uint64_t t_beg = tile_first + n_tiles * blockIdx.x / gridDim.x, t_end = tile_first + n_tiles * (blockIdx.x + 1) / gridDim.x;
    const uint64_t seg0 = (uint64_t)blockIdx.x * seg_cap;
    uint64_t cursor = 0;  // records written by this workgroup (uniform)
    const int w = k - m + 1;
    const uint32_t mid_mask = (k >= 2) ? ((1u << (k - 1)) - 1u) : 0u;
    uint64_t n_k = 0, n_e = 0;
    bool overflow = false;
    for (uint64_t tile = t_beg; tile < t_end; ++tile) {
        const uint64_t tile0 = tile * TILE;
        __syncthreads();  // LDS of the previous tile is free
        const uint32_t bad = load_tile(s.t, bases, n_bytes, startbits, tile0);
        if (bad) atomicOr(&scalars[0], 1ull);
        __syncthreads();
        for (int j = threadIdx.x; j < TILE + HALO - 32; j += 256)
            s.hm[j] = (uint16_t)mmer_hash16((uint32_t)(window32(s.t, j) >> (64 - 2 * m)));
        __syncthreads();
        for (int j0 = 0; j0 < TILE; j0 += 256) {
            const int j = j0 + threadIdx.x;
            const uint64_t p = tile0 + j;
            bool v = false;
            uint32_t mp = 0xFFu;
            if (p < n_bytes) {
                const uint32_t sw = startwin32(s.t, j);
                const uint32_t s0 = sw & 1u, sk = (sw >> k) & 1u;
                v = (((sw >> 1) & mid_mask) == 0) && !(sk && s0);
                if (v) {
                    n_k += 1;
                    n_e += sk ^ 1u;
                    uint32_t best = s.hm[j];
                    mp = 0;
                    for (int i = 1; i < w; ++i) {
                        const uint32_t hv = s.hm[j + i];
                        if (hv < best) { best = hv; mp = i; }
                    }
                }
            }
            s.minp[j] = (uint8_t)mp;
            const unsigned long long vb = __ballot(v);
            if ((threadIdx.x & 63) == 0) s.vbits[j >> 6] = vb;
        }
        __syncthreads();
        for (int j0 = 0; j0 < TILE; j0 += 256) {
            const int j = j0 + threadIdx.x;
            const uint32_t mp = s.minp[j];
            // same minimizer occurrence as the previous k-mer <=> offset one larger there
            const bool st = (mp != 0xFFu) && (j == 0 || (uint32_t)s.minp[j - 1] != mp + 1);
            const unsigned long long sb = __ballot(st);
            if ((threadIdx.x & 63) == 0) s.sbits[j >> 6] = sb;
        }
        __syncthreads();
        if (threadIdx.x < 64) {  // exclusive prefix of per-word record counts (128 words, one wave)
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
        if (cursor + nrec > seg_cap) { overflow = true; break; }  // uniform: caller retries with a larger segment
        const uint64_t gbase = seg0 + cursor;
        cursor += nrec;
        for (int j0 = 0; j0 < TILE; j0 += 256) {  // dense work list of record starts
            const int j = j0 + threadIdx.x;
            const unsigned long long sb = s.sbits[j >> 6];
            if ((sb >> (j & 63)) & 1ull) s.list[s.wpre[j >> 6] + __popcll(sb & ((1ull << (j & 63)) - 1))] = (uint16_t)j;
        }
        __syncthreads();
        for (uint32_t r = threadIdx.x; r < nrec; r += 256) {
            const int j = s.list[r];
            const int wd = j >> 6, bt = j & 63;
            const unsigned long long sb = s.sbits[wd];
            // run length: up to the next record start or the first position without a k-mer
            const unsigned long long nxt_s = (bt == 63) ? 0ull : (sb >> (bt + 1));
            const unsigned long long nxt_i = (bt == 63) ? 0ull : ((~s.vbits[wd]) >> (bt + 1));
            const int room = 63 - bt;  // positions after j inside this word
            const unsigned long long stop = nxt_s | nxt_i;
            int len;
            if (stop) {
                len = 1 + (__ffsll((unsigned long long)stop) - 1);
            } else {
                const unsigned long long stop2 = s.sbits[wd + 1] | ~s.vbits[wd + 1];  // sentinel word: all stop
                len = 1 + room + (__ffsll((unsigned long long)stop2) - 1);
            }
            if (j + len > TILE) len = TILE - j;
            const uint64_t p = tile0 + j;
            const uint32_t s0 = startwin32(s.t, j) & 1u;
            const uint32_t sk_last = (startwin32(s.t, j + len - 1) >> k) & 1u;
            const uint32_t has_succ = sk_last ^ 1u;
            const int nb = k + len - 1 + (int)has_succ;  // bases carried by the record
            uint64_t w0 = window32(s.t, j);
            uint64_t hi = window32(s.t, j + 32);
            if (nb < 32) { w0 &= ~0ull << (64 - 2 * nb); hi = 0; }
            else if (nb == 32) hi = 0;
            else hi &= ~0ull << (64 - 2 * (nb - 32));
            const uint32_t mp = j + s.minp[j];
            const uint32_t bh = bucket_hash22((uint32_t)(window32(s.t, mp) >> (64 - 2 * m)));
            const uint64_t w1 = (hi & (~0ull << SK_META_BITS)) | ((uint64_t)bh << 6) | ((uint64_t)(len - 1) << 1) | has_succ;
            const uint64_t o = gbase + r;
            rec_w0[o] = w0;
            rec_w1[o] = w1;
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

// Fast extraction for m = 13 and a compile-time window W = k - 12 (k = 31 -> W = 19).
// Every lane owns 32 consecutive positions: it reads 64 bases and 64 read-start bits once, rolls
// the 32 + W - 1 m-mer hashes it needs in registers and takes the sliding-window minimum with
// the van Herk / Gil-Werman block scheme (about 4 min ops per position instead of W LDS reads).
// Hash and leftmost tie-break are the same as in the generic kernel: packed = hash16 << 8 | offset.
struct SkLdsW {
    TileLds t;
    uint8_t minp[TILE];
    unsigned long long sbits[TILE / 64 + 1];
    unsigned long long vbits[TILE / 64 + 1];
    uint32_t wpre[TILE / 64 + 1];
    uint16_t list[TILE];            // record index -> start position (dense work list)
    uint8_t edge[256];
    uint32_t nrec;
};

#ifndef DBG_EXW_WAVES
#define DBG_EXW_WAVES 3
#endif
template <class ST, int W>
__global__ __launch_bounds__(256, DBG_EXW_WAVES) void k_sk_extract_w(const char *__restrict__ bases, uint64_t n_bytes,
                                                      const uint32_t *__restrict__ startbits, uint64_t n_tiles,
                                                      uint64_t *rec_w0, uint64_t *rec_w1, ST *rec_st, uint64_t seg_cap,
                                                      uint64_t *seg_cnt, uint64_t *seg_nk, uint64_t *seg_ne,
                                                      unsigned long long *scalars /* [0] err */, uint64_t tile_first) {
    constexpr int M = SK_MAX_M, K = W + M - 1, NV = 32 + W - 1;
    static_assert(TILE == 256 * 32, "one lane per 32 positions");
    static_assert(K <= 31 && NV + M - 1 <= 64, "window must fit the two 32-base registers");
    __shared__ SkLdsW s;
    __shared__ uint64_t red[8];
    const uint64_t t_beg = tile_first + n_tiles * blockIdx.x / gridDim.x, t_end = tile_first + n_tiles * (blockIdx.x + 1) / gridDim.x;
    const uint64_t seg0 = (uint64_t)blockIdx.x * seg_cap;
    uint64_t cursor = 0;
    constexpr uint32_t mid_mask = (1u << (K - 1)) - 1u;
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
        const uint64_t wa = window32(s.t, j0), wb = window32(s.t, j0 + 32);
        uint32_t pv[NV], P[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const uint64_t win = (i == 0) ? wa : (i < 32) ? ((wa << (2 * i)) | (wb >> (64 - 2 * i))) : (i == 32) ? wb : (wb << (2 * (i - 32)));
            pv[i] = (mmer_hash16((uint32_t)(win >> (64 - 2 * M))) << 8) | (uint32_t)i;
            P[i] = (i % W == 0) ? pv[i] : min(P[i - 1], pv[i]);
        }
        const uint64_t sb64 = ((uint64_t)s.t.sb[(j0 >> 5) + 1] << 32) | s.t.sb[j0 >> 5];
        // a k-mer at position q is valid iff no read starts at q + 1 .. q + K - 1, not both at q and q + K (a read of
        // exactly K bases has no successor but is a k-mer ... of a read with len == K: excluded, debruijn.py:123), and q
        // lies inside the reads -- all 32 positions of the lane at once, on the bit masks
        const uint32_t skmask = (uint32_t)(sb64 >> K);
        const uint64_t left = n_bytes > tile0 + (uint64_t)j0 ? n_bytes - (tile0 + (uint64_t)j0) : 0;
        const uint32_t inside = left >= 32 ? 0xFFFFFFFFu : ((1u << (uint32_t)left) - 1u);
        const uint32_t vmask = ~(uint32_t)window_or64<K - 1>(sb64 >> 1) & ~(skmask & (uint32_t)sb64) & inside;
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
                // new record unless the previous position holds a k-mer with the same minimizer occurrence
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
        {  // dense work list of record starts: this lane's 32 positions
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
            const uint32_t s0 = startwin32(s.t, j) & 1u;
            const uint32_t sk_last = (startwin32(s.t, j + len - 1) >> K) & 1u;
            const uint32_t has_succ = sk_last ^ 1u;
            const int nb = K + len - 1 + (int)has_succ;
            uint64_t w0 = window32(s.t, j);
            uint64_t hi = window32(s.t, j + 32);
            if (nb < 32) { w0 &= ~0ull << (64 - 2 * nb); hi = 0; }
            else if (nb == 32) hi = 0;
            else hi &= ~0ull << (64 - 2 * (nb - 32));
            const uint32_t mp = j + s.minp[j];
            const uint32_t bh = bucket_hash22((uint32_t)(window32(s.t, mp) >> (64 - 2 * M)));
            const uint64_t w1 = (hi & (~0ull << SK_META_BITS)) | ((uint64_t)bh << 6) | ((uint64_t)(len - 1) << 1) | has_succ;
            const uint64_t o = gbase + r;
            rec_w0[o] = w0;
            rec_w1[o] = w1;
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

// ------------------------------------------------------------------------------------------------
// K2-K4: multisplit.  Records of `parents` (contiguous ranges) are split by a bit field of w1 into
// nb children each; a workgroup owns one super-chunk (MS_SC records of one parent).
// counts matrix: cmat[sc * nb + b]; offsets come from an exclusive scan in (parent, b, sc) order.
// ------------------------------------------------------------------------------------------------
#ifndef DBG_MS_CH
#define DBG_MS_CH 4096
#endif
constexpr int MS_CH = DBG_MS_CH;      // records sorted per LDS round with 64-bit stamps (24 B per record)
// ... and with 32-bit stamps (20 B per record: 6144 fit): a child gets chunk / fan-out records per round in one piece --
// 12 instead of 8 at a fan-out of 512, 96-byte instead of 64-byte pieces of w0 / w1 (partition 4.82 -> 4.57 ms)
template <class ST> struct MsChunk { static constexpr int CH = sizeof(ST) == 8 ? MS_CH : MS_CH + MS_CH / 2; };
constexpr int MS_SC = 32768;          // records per super-chunk
constexpr int MS_MAX_NB = 1024;
#ifndef DBG_MS_NT
#define DBG_MS_NT 512
#endif
constexpr int MS_NT = DBG_MS_NT;      // threads of the scatter workgroup: its 90 KB of LDS allow one per CU, so the
                                      // workgroup itself has to bring the waves that hide the record loads

// Input of one multisplit level: `n_seg` contiguous segments of the record arrays, in groups of `spg`
// consecutive segments: all segments in ONE group (level 1: the per-workgroup output segments of
// k_sk_extract), every segment its own group (level 2: the children of level 1), or one group per
// level-1 bucket made of one segment per sender (level 2 of a sharded build: the senders split by
// level 1 before the exchange, so the receiver starts here).  A group is split into nb children that
// are laid out contiguously, groups in order.
struct MsParents {
    const uint64_t *start;    // [n_seg]
    const uint64_t *cnt;      // [n_seg]
    const uint64_t *sc_pre;   // [n_seg + 1] super-chunks before segment i
    uint32_t n_seg;
    uint32_t spg;             // segments per group (n_seg: one group; 1: every segment its own)
};

// which segment owns super-chunk g, and which super-chunk of that segment it is
__device__ inline void ms_locate(const MsParents &P, uint64_t g, uint32_t *seg, uint64_t *sidx) {
    uint32_t lo = 0, hi = P.n_seg;  // sc_pre[lo] <= g < sc_pre[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (P.sc_pre[mid] <= g) lo = mid; else hi = mid;
    }
    *seg = lo;
    *sidx = g - P.sc_pre[lo];
}

// child bucket of a record: the nb-way digit at `shift` of w1 (nb a power of two, fbits == 0), or -- so that the
// bucket count need not be a power of two -- the fbits-wide field at `shift` scaled into [0, nb)
__device__ inline uint32_t ms_child(uint64_t w1, int shift, int nb, int fbits) {
    const uint32_t v = (uint32_t)(w1 >> shift);
    return fbits ? (uint32_t)(((uint64_t)(v & ((1u << fbits) - 1u)) * (uint32_t)nb) >> fbits) : (v & (uint32_t)(nb - 1));
}

// SUMS: also adds up the k-mer and edge instances the records carry (sums[0], sums[1]): a sharded build learns the
// size of what it received in the pass it makes anyway
template <bool SUMS>
__global__ __launch_bounds__(256) void k_ms_hist(MsParents P, const uint64_t *__restrict__ w1, int shift, int nb, int fbits,
                                                 uint32_t *cmat, unsigned long long *sums) {
    __shared__ uint32_t hist[MS_MAX_NB];
    uint32_t seg;
    uint64_t sidx;
    ms_locate(P, blockIdx.x, &seg, &sidx);
    for (int b = threadIdx.x; b < nb; b += 256) hist[b] = 0;
    __syncthreads();
    const uint64_t beg = P.start[seg] + sidx * MS_SC;
    const uint64_t end = min(P.start[seg] + P.cnt[seg], beg + (uint64_t)MS_SC);
    uint64_t n_inst = 0, n_edge = 0;
    for (uint64_t i = beg + threadIdx.x; i < end; i += 256) {
        const uint64_t x = w1[i];
        atomicAdd(&hist[ms_child(x, shift, nb, fbits)], 1u);
        if (SUMS) { n_inst += ((x >> 1) & 31) + 1; n_edge += ((x >> 1) & 31) + (x & 1); }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nb; b += 256) cmat[(uint64_t)blockIdx.x * nb + b] = hist[b];
    if (SUMS) {
        n_inst = wave_sum_u64(n_inst);
        n_edge = wave_sum_u64(n_edge);
        if ((threadIdx.x & 63) == 0) {
            if (n_inst) atomicAdd(&sums[0], (unsigned long long)n_inst);
            if (n_edge) atomicAdd(&sums[1], (unsigned long long)n_edge);
        }
    }
}

// logical scan order: group-major, then child bucket, then super-chunk inside the group
struct MsLogical {
    MsParents P;
    const uint32_t *cmat;
    int nb;
    __device__ uint64_t operator()(uint64_t L) const {
        const uint32_t n_groups = P.n_seg / P.spg;
        uint32_t lo = 0, hi = n_groups;  // lbase[g] = nb * sc_pre[g * spg]
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((uint64_t)nb * P.sc_pre[mid * P.spg] <= L) lo = mid; else hi = mid;
        }
        const uint64_t g_lo = P.sc_pre[lo * P.spg];
        const uint64_t nsc = P.sc_pre[(lo + 1) * P.spg] - g_lo;
        const uint64_t r = L - (uint64_t)nb * g_lo;
        const uint64_t b = r / nsc, sidx = r - b * nsc;
        return cmat[(g_lo + sidx) * nb + b];
    }
};

// children descriptors: start/cnt of child (group g, bucket b) = index g * nb + b
__global__ __launch_bounds__(256) void k_ms_children(MsParents P, const uint64_t *__restrict__ offs, int nb,
                                                     uint64_t total, uint64_t *c_start, uint64_t *c_cnt) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_groups = P.n_seg / P.spg;
    const uint64_t n_child = (uint64_t)n_groups * nb;
    if (i >= n_child) return;
    const uint32_t g = (uint32_t)(i / nb), b = (uint32_t)(i % nb);
    const uint64_t g_lo = P.sc_pre[g * P.spg], g_hi = P.sc_pre[(g + 1) * P.spg];
    const uint64_t nsc = g_hi - g_lo;
    const uint64_t L = (uint64_t)nb * g_lo + (uint64_t)b * nsc;  // first logical element of this child
    const uint64_t Ln = L + nsc;                                   // ... of the next child
    const uint64_t total_L = (uint64_t)nb * P.sc_pre[P.n_seg];
    const uint64_t a = (L < total_L) ? offs[L] : total;
    const uint64_t e = (Ln < total_L) ? offs[Ln] : total;
    c_start[i] = a;
    c_cnt[i] = (nsc == 0) ? 0 : e - a;
}

template <class ST>
struct MsLds {
    uint64_t w0[MsChunk<ST>::CH];
    uint64_t w1[MsChunk<ST>::CH];
    ST st[MsChunk<ST>::CH];
    uint32_t hist[MS_MAX_NB];    // count in this chunk
    uint32_t start[MS_MAX_NB];   // exclusive prefix inside the chunk
    uint64_t run[MS_MAX_NB];     // global cursor of the super-chunk per child bucket
};

#ifdef DBG_MS_PROF
__device__ unsigned long long g_ms_prof[8];
#define MS_TICK(i) do { if (threadIdx.x == 0) { const unsigned long long now_ = clock64(); prof_[i] += now_ - last_; last_ = now_; } } while (0)
#else
#define MS_TICK(i) do {} while (0)
#endif

// STI: stamp type of the input (a sharded build receives rank-local 32-bit stamps and rebases them to global 64-bit
// ones here: seg_add[segment] = 2 x the byte offset of the sender's reads in the rank-major concatenation)
template <class ST, bool HAS_ST, class STI = ST>
__global__ __launch_bounds__(MS_NT) void k_ms_scatter(MsParents P, const uint64_t *__restrict__ in_w0,
                                                    const uint64_t *__restrict__ in_w1, const STI *__restrict__ in_st,
                                                    const uint64_t *__restrict__ seg_add,
                                                    int shift, int nb, int fbits, const uint64_t *__restrict__ offs,
                                                    uint64_t *out_w0, uint64_t *out_w1, ST *out_st) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ms_raw[];
    MsLds<ST> &s = *reinterpret_cast<MsLds<ST> *>(ms_raw);
    constexpr int CH = MsChunk<ST>::CH;
    static_assert(CH % MS_NT == 0 && sizeof(MsLds<ST>) <= 160 * 1024, "multisplit chunk");
#ifdef DBG_MS_PROF
    unsigned long long prof_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = clock64();
#endif
    uint32_t seg;
    uint64_t sidx;
    ms_locate(P, blockIdx.x, &seg, &sidx);
    const ST st_add = (HAS_ST && seg_add) ? (ST)seg_add[seg] : (ST)0;
    MS_TICK(0);
    // position of this super-chunk inside its group, and the group's extent in the logical order
    const uint32_t grp = seg / P.spg;
    const uint64_t g_lo = P.sc_pre[grp * P.spg], g_hi = P.sc_pre[(grp + 1) * P.spg];
    const uint64_t nsc = g_hi - g_lo, gidx = (uint64_t)blockIdx.x - g_lo;
    const uint64_t lbase = (uint64_t)nb * g_lo;
    for (int b = threadIdx.x; b < nb; b += MS_NT) s.run[b] = offs[lbase + (uint64_t)b * nsc + gidx];
    const uint64_t beg = P.start[seg] + sidx * MS_SC;
    const uint64_t end = min(P.start[seg] + P.cnt[seg], beg + (uint64_t)MS_SC);
    for (uint64_t c0 = beg; c0 < end; c0 += CH) {
        const int n = (int)min((uint64_t)CH, end - c0);
        for (int b = threadIdx.x; b < nb; b += MS_NT) s.hist[b] = 0;
        __syncthreads();
        MS_TICK(1);
        uint64_t r0[CH / MS_NT], r1[CH / MS_NT];
        ST rs[CH / MS_NT];
        uint32_t rk[CH / MS_NT];
#pragma unroll
        for (int i = 0; i < CH / MS_NT; ++i) {
            const int q = i * MS_NT + threadIdx.x;
            if (q < n) {
                r0[i] = in_w0[c0 + q];
                r1[i] = in_w1[c0 + q];
                if (HAS_ST) rs[i] = (ST)in_st[c0 + q] + st_add;
                rk[i] = atomicAdd(&s.hist[ms_child(r1[i], shift, nb, fbits)], 1u);
            }
        }
        __syncthreads();
        MS_TICK(2);
        {  // exclusive scan of hist (nb <= 1024: MS_MAX_NB / MS_NT entries per thread)
            constexpr int EPT = MS_MAX_NB / MS_NT;
            const int b0 = threadIdx.x * EPT;
            uint32_t h4[EPT];
            uint64_t sum = 0;
#pragma unroll
            for (int j = 0; j < EPT; ++j) { h4[j] = (b0 + j < nb) ? s.hist[b0 + j] : 0; sum += h4[j]; }
            uint64_t tot;
            uint32_t ex = (uint32_t)block_exscan<MS_NT / 64>(sum, &tot);
#pragma unroll
            for (int j = 0; j < EPT; ++j) { if (b0 + j < nb) s.start[b0 + j] = ex; ex += h4[j]; }
        }
        __syncthreads();
        MS_TICK(3);
#pragma unroll
        for (int i = 0; i < CH / MS_NT; ++i) {
            const int q = i * MS_NT + threadIdx.x;
            if (q < n) {
                const uint32_t b = ms_child(r1[i], shift, nb, fbits);
                const uint32_t d = s.start[b] + rk[i];
                s.w0[d] = r0[i];
                s.w1[d] = r1[i];
                if (HAS_ST) s.st[d] = rs[i];
            }
        }
        __syncthreads();
        MS_TICK(4);
        for (int q = threadIdx.x; q < n; q += MS_NT) {
            const uint64_t x1 = s.w1[q];
            const uint32_t b = ms_child(x1, shift, nb, fbits);
            const uint64_t g = s.run[b] + (q - s.start[b]);
            out_w0[g] = s.w0[q];
            out_w1[g] = x1;
            if (HAS_ST) out_st[g] = s.st[q];
        }
        __syncthreads();
        MS_TICK(5);
        for (int b = threadIdx.x; b < nb; b += MS_NT) s.run[b] += s.hist[b];
        __syncthreads();
        MS_TICK(6);
    }
#ifdef DBG_MS_PROF
    if (threadIdx.x == 0) {
        for (int i = 0; i < 7; ++i) atomicAdd(&g_ms_prof[i], prof_[i]);
        atomicAdd(&g_ms_prof[7], 1ull);
    }
#endif
}


// ------------------------------------------------------------------------------------------------
// Distinct-k-mer estimate: the k-mers of ONE level-1 bucket (1/nb1 of the data) go through a small
// global hash set; distinct / instances of that sample sizes the final bucket count.
// ------------------------------------------------------------------------------------------------
template <class ST>
__global__ __launch_bounds__(256) void k_estimate_distinct(const uint64_t *__restrict__ b_start,
                                                           const uint64_t *__restrict__ b_cnt, uint32_t bucket,
                                                           const uint64_t *__restrict__ rec_w0,
                                                           const uint64_t *__restrict__ rec_w1, int k,
                                                           unsigned long long *set, uint64_t set_mask,
                                                           unsigned long long *out /* [0] instances [1] distinct */) {
    const uint64_t beg = b_start[bucket], n = b_cnt[bucket];
    uint64_t inst = 0, fresh = 0;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t w0 = rec_w0[beg + r], w1 = rec_w1[beg + r];
        const uint64_t hi = w1 & (~0ull << SK_META_BITS);
        const int len = (int)((w1 >> 1) & 31) + 1;
        for (int i = 0; i < len; ++i) {
            const uint64_t kmer = (i ? (w0 << (2 * i)) | (hi >> (64 - 2 * i)) : w0) >> (64 - 2 * k);
            ++inst;
            uint64_t slot = mix64(kmer) & set_mask;
            for (uint64_t probe = 0; probe <= set_mask; ++probe) {
                unsigned long long cur = set[slot];
                if (cur == EMPTY_KEY) {
                    cur = atomicCAS(&set[slot], EMPTY_KEY, (unsigned long long)kmer);
                    if (cur == EMPTY_KEY) { ++fresh; break; }
                }
                if (cur == kmer) break;
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

// ------------------------------------------------------------------------------------------------
// K5: per-bucket counting in LDS.
// ------------------------------------------------------------------------------------------------
struct SkRange {            // one successfully counted (bucket, hash sub-range)
    uint32_t bucket;
    uint32_t mask, val;     // k-mers with (sub_hash & mask) == val
    uint32_t node_cnt;
    uint64_t node_base;
    uint32_t next;          // buckets counted in sub-ranges: chain of their ranges, headed by ranges[bucket].next (0 = end)
    uint32_t pad;
};

__device__ inline uint32_t fold32(uint64_t kmer) { return (uint32_t)kmer * 0x9E3779B1u ^ (uint32_t)(kmer >> 32) * 0x85EBCA77u; }
__device__ inline uint32_t sub_hash(uint64_t kmer) { return fmix32(fold32(kmer) ^ 0x27D4EB2Fu); }
// LDS table slot: Fibonacci hashing of the folded key, top bits (one multiply per probe sequence)
__device__ inline uint32_t slot_hash(uint64_t kmer) {
    const uint32_t lo = (uint32_t)kmer, hi = (uint32_t)(kmer >> 32);
    return (lo ^ (hi << 11) ^ (hi >> 7) ^ (lo >> 15)) * 0x9E3779B1u;
}
template <int CAP>
__device__ inline uint32_t slot_of(uint64_t kmer) {
    static_assert(CAP == 2048 || CAP == 4096, "table size");
    return slot_hash(kmer) >> (CAP == 4096 ? 20 : 21);
}

// cross-bucket successor queries staged per workgroup; also the record staging depth (a bucket holds ~530 records at
// the default geometry: a bucket that fits one staging round pays the dedupe / quad list / insert barriers once).
// With 64-bit stamps (sharded builds) the 4096-slot table leaves room for 640.
template <class ST, int CAP>
struct CntCfg {
    #ifdef DBG_CNT_PROF
    static constexpr int QBUF = CAP == 4096 ? (sizeof(ST) == 8 ? 624 : 768) : 400;  // room for the clocks of the experiment build
#else
    static constexpr int QBUF = CAP == 4096 ? (sizeof(ST) == 8 ? 640 : 768) : 400;
#endif
    // 4096 slots fill the LDS: one 1024-thread workgroup per CU.  2048 slots: two 512-thread workgroups per CU
    // that run out of step, so one's barriers and LDS stalls overlap the other's work.
    static constexpr int NT = CAP == 4096 ? 1024 : 512;
};
constexpr int CNT_STACK = 24;    // pending hash sub-ranges of one bucket
constexpr int CNT_PROBE_LIMIT = 1024;

// (Field order: measured.  With the small arrays and the keys below 64 KB most accesses become (index << shift) +
// 16-bit immediate -- 5 % fewer vector instructions in the ISA -- and the kernel gets SLOWER, 13.7 vs 13.5 ms, as does
// the two-word kernel, 28.2 vs 27.7: the tables' bank alignment against the staging arrays changes with it.)
template <class ST, int CAP>
struct CntLds {
    static constexpr int CNT_QBUF = CntCfg<ST, CAP>::QBUF;
    unsigned long long keys[CAP];
    uint32_t cnt[CAP * 4];
    ST stamp[CAP];
    uint16_t list[CAP];   // local node index -> slot (slot -> local node index: dir_base + rank in dir_mask)
    uint16_t eoff[CAP];   // local node index -> first CSR edge of the node, relative to the bucket
    unsigned long long q_key[CNT_QBUF];   // insert phase: staged record w0; afterwards: query keys
    unsigned long long q_meta[CNT_QBUF];  // insert phase: staged record w1; afterwards: query meta
    ST st_stage[CNT_QBUF];                // insert phase: staged record stamps
    uint32_t stk_mask[CNT_STACK], stk_val[CNT_STACK];
    int stk_n;
    uint32_t overflow, n_local /* nodes | edges << 16 while the list is built */, n_q, fail, n_flat;
    unsigned long long gbase, qbase, ebase, ri;
    unsigned long long dir_mask[CAP / 64];  // occupancy of every 64-slot block of the final table
    uint16_t dir_base[CAP / 64];            // local node index of the block's first node
#ifdef DBG_CNT_PROF
    unsigned long long prof[64];
#endif
};

// k-mer i of a record: 32-base window starting at base i (first base in bits 63:62)
__device__ inline uint64_t rec_window(uint64_t w0, uint64_t hi, int i) {
    return (w0 << (2 * i)) | ((hi >> 1) >> (63 - 2 * i));  // i = 0: the second term shifts out (no select for the shift by 64)
}

template <int CAP>
__device__ inline int lds_find(const unsigned long long *keys, uint64_t key) {
    uint32_t slot = slot_of<CAP>(key);
#pragma unroll 16
    for (int probe = 0; probe < CAP; ++probe) {
        const unsigned long long cur = keys[slot];
        if (cur == key) return (int)slot;
        if (cur == EMPTY_KEY) return -1;
        slot = (slot + 1) & (CAP - 1);
    }
    return -1;
}

// Directory of one counted range: per 64-slot block of its LDS table the occupancy mask and the node id of the
// block's first node.  The nodes of a block are written in slot order, so (mask, base) turn a slot into a node id and
// k_succ_resolve can repeat the table's linear probing against the node keys in HBM -- a successor that lives in
// another bucket costs the asker two dependent reads instead of a trip through a multisplit of all such queries.
struct SkDirEnt {
    unsigned long long mask;
    uint32_t base;
    uint32_t pad;
};
static_assert(sizeof(SkDirEnt) == 16, "directory entry");

// What the count kernel writes: the graph in ONE representation -- node keys, first-occurrence stamps, one byte per
// node (indegree flag | bases that occur << 1) and the CSR rows (rowptr, successor id, count).  The dense per-base views
// (counts[n][4], succ[n][4], rank bytes) that the traversal kernels read are derived from it on first use
// (k_dense_from_csr): writing both cost 2.7x the bytes and a third of the kernel's vector instructions.
struct SkCountOut {
    uint64_t *keys;
    void *stamps;              // ST[node_cap]
    uint8_t *flags;            // (stamp & 1) | present-base mask << 1
    uint64_t node_cap;
    uint32_t *rowptr;          // CSR: first edge of every node (edges < 2^32 - 16 is enforced)
    uint32_t *col, *ecnt;      // CSR: successor id (NO_NODE until k_succ_resolve fills a cross-bucket one), count
    uint64_t edge_cap;
    uint64_t *q_key;           // cross-bucket successors: the successor k-mer ...
    uint32_t *q_col;           // ... and the CSR position to patch
    uint64_t q_cap;
    SkRange *ranges;        // [0, n_buckets): the unsplit range of each bucket; beyond: ranges of split buckets
    uint64_t n_buckets;
    uint64_t range_cap;
    SkDirEnt *dirs;            // [own_cnt + extra ranges][CAP / 64]: only the buckets this build owns have a directory
    uint64_t own_lo, own_cnt;  // buckets [own_lo, own_lo + own_cnt) (everything for a single-GPU build)
    uint32_t id_tag;           // OR-ed into every successor id written (sharded builds: owner << 29)
    unsigned long long *scalars;  // [0] err [4] packed cursor: nodes (low 32) | edges (high 32) [5] queries [6] extra ranges
};

// The output descriptor lives in device memory and is re-read (scalar loads, a few per phase) where a phase
// needs it: passed by value its ~50 SGPRs stay live across the whole persistent loop, and the compiler
// spilled them to VGPR lanes -- a third of the node-write phase was v_readlane reloads.
// (constant address space: uniform scalar loads the stores of the phase cannot alias)
typedef const SkCountOut __attribute__((address_space(4))) *SkOutConstPtr;
__device__ inline SkOutConstPtr fresh_args(const SkCountOut *p) {
    unsigned long long v = (unsigned long long)p;
    asm volatile("" : "+s"(v));  // a new value for the optimiser: loads through it are not merged with earlier ones
    return (SkOutConstPtr)v;
}

// wave-aggregated LDS counter: returns this lane's index, adds popcount(active & pred) once per wave
__device__ inline uint32_t wave_alloc(uint32_t *counter, bool pred) {
    const unsigned long long mask = __ballot(pred);
    if (!mask) return 0;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((unsigned long long)mask) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, leader, 64);
    return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1));
}

// lanes below this one that are set in a ballot mask
__device__ inline uint32_t lanes_below(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// same for n <= NMAX items per lane: returns the index of the lane's first item.  The prefix sum is
// NMAX ballots + mbcnt (scalar/VALU only) instead of a six-step cross-lane scan through the LDS crossbar.
// Must be called with the wave converged.
template <int NMAX>
__device__ inline uint32_t wave_alloc_n(uint32_t *counter, uint32_t n) {
    uint32_t excl = 0, total = 0;
#pragma unroll
    for (int j = 1; j <= NMAX; ++j) {
        const unsigned long long mj = __ballot(n >= (uint32_t)j);
        if (j == 1 && !mj) return 0;
        excl += lanes_below(mj);
        total += (uint32_t)__popcll(mj);
    }
    uint32_t base = 0;
    if ((threadIdx.x & 63) == 0) base = atomicAdd(counter, total);
    base = __builtin_amdgcn_readfirstlane(base);
    return base + excl;
}

#ifdef DBG_CNT_PROF
__device__ unsigned long long g_cnt_prof[64];
#define CNT_TICK(i) do { if (threadIdx.x == 0) { const unsigned long long now_ = clock64(); s.prof[i] += now_ - clast_; clast_ = now_; } } while (0)
#define CNT_SUBTICK(i) do { if (threadIdx.x == 0) { const unsigned long long now_ = clock64(); s.prof[i] += now_ - csub_; csub_ = now_; } } while (0)
#define CNT_EVENT(i) do { if (threadIdx.x == 0) s.prof[i] += 1; } while (0)
#else
#define CNT_EVENT(i) do {} while (0)
#define CNT_TICK(i) do {} while (0)
#define CNT_SUBTICK(i) do {} while (0)
#endif

template <class ST, int CAP>
__global__ __launch_bounds__((CntCfg<ST, CAP>::NT)) void k_sk_count(const uint64_t *__restrict__ b_start, const uint64_t *__restrict__ b_cnt,
                                                     const uint64_t *__restrict__ rec_w0, const uint64_t *__restrict__ rec_w1,
                                                     const ST *__restrict__ rec_st, int k, int m, uint64_t n_buckets,
                                                     const SkCountOut *__restrict__ outp, uint32_t split_recs,
                                                     int phase_limit /* ablation only: 0 = run everything */) {
    // split_recs: a bucket with more records than this is counted in 2^j hash sub-ranges from the start (the host
    // knows the distinct k-mers per record): finding out by overflowing the table first costs a full insert pass
    // with probe runs through a full table -- per level of the split
    // Persistent: one workgroup per CU walks buckets blockIdx.x, blockIdx.x + gridDim.x, ...  The table
    // stays in LDS across buckets: the node write clears exactly the slots it reads (a third of the
    // table), and the next bucket's records are prefetched into registers while this bucket is in
    // its lookup / reservation / write phases.
    extern __shared__ __attribute__((aligned(16))) unsigned char cnt_raw[];
    CntLds<ST, CAP> &s = *reinterpret_cast<CntLds<ST, CAP> *>(cnt_raw);
    using LdsT = CntLds<ST, CAP>;
    constexpr int CNT_NT = CntCfg<ST, CAP>::NT;
    constexpr int NPT = CAP / CNT_NT;  // nodes per thread, upper bound
    constexpr int CNT_QBUF = CntCfg<ST, CAP>::QBUF;
    // scratch of the insert phase lives in arrays that are only needed once the table is final: the dedupe set and
    // then the quad list in eoff[], the multiplicities in list[].  A record holds at most w = k - m + 1 <= 19 k-mers:
    // five quads.
    static_assert(CNT_QBUF * 5 <= CAP, "quad list must fit eoff[]");
    constexpr uint32_t STAGE = CNT_QBUF;
    static_assert(STAGE <= CNT_NT, "one staged record per thread");
    static_assert(sizeof(LdsT) <= 160 * 1024, "LDS");
    const uint64_t kmask = (1ull << (2 * k)) - 1;
    uint16_t *flat = s.list;
    constexpr uint64_t BH_FIELD = ((1ull << SK_BUCKET_BITS) - 1) << 6;  // of a staged w1: the record's multiplicity (the bucket is known)

#ifdef DBG_CNT_PROF
    unsigned long long clast_ = clock64(), csub_ = clast_;
    if (threadIdx.x < 64) s.prof[threadIdx.x] = 0;
    __syncthreads();
#endif
    bool clean = false;  // uniform: every slot of the table is EMPTY / zero / max
    uint64_t pf_w0 = 0, pf_w1 = 0;
    ST pf_st = 0;
    uint64_t nx_beg = 0, nx_n = 0;  // record range of the prefetched bucket (uniform)
    uint64_t r2_beg = 0, r2_n = 0;  // ... and of the one after it: loaded a whole bucket ahead, so the record
                                    // prefetch never waits for the (scalar, L2-latency) load of its own address
    auto load_range = [&](uint64_t bucket, uint64_t &beg, uint64_t &n) {
        n = 0;
        if (bucket < n_buckets) { beg = b_start[bucket]; n = b_cnt[bucket]; }
    };
    auto prefetch = [&](uint64_t bucket) {  // records of `bucket` (its range is in r2_*), then the range after it
        nx_beg = r2_beg;
        nx_n = r2_n;
        if (threadIdx.x < min(nx_n, (uint64_t)STAGE)) {
            pf_w0 = rec_w0[nx_beg + threadIdx.x];
            pf_w1 = rec_w1[nx_beg + threadIdx.x];
            pf_st = rec_st[nx_beg + threadIdx.x];
        }
        load_range(bucket + gridDim.x, r2_beg, r2_n);
    };
    if (threadIdx.x == 0) s.fail = 0;
    load_range(blockIdx.x, r2_beg, r2_n);
    prefetch(blockIdx.x);

    for (uint64_t bucket = blockIdx.x; bucket < n_buckets; bucket += gridDim.x) {
        const uint64_t r_beg = nx_beg, r_n = nx_n;  // loaded with the prefetch: no exposed global latency here
        if (r_n == 0) { prefetch(bucket + gridDim.x); continue; }
        bool have_prefetch = true;  // registers hold the first STAGE records of this bucket
        bool skip_rest = false;     // ablation exit
        // The stack of hash sub-ranges still to count lives in registers (uniform) + LDS for its entries; the
        // common case (the whole bucket fits the table) never touches LDS for it.
        uint32_t stk_n = 1;
        bool root = true, failed = false;
        if (split_recs && r_n > split_recs) {  // uniform
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
            if (!clean) {
                for (int i = threadIdx.x; i < CAP; i += CNT_NT) {
                    s.keys[i] = EMPTY_KEY;
                    s.stamp[i] = (ST)~(ST)0;
                    reinterpret_cast<uint4 *>(s.cnt)[i] = make_uint4(0, 0, 0, 0);
                }
            }
            clean = false;
            if (threadIdx.x == 0) { s.overflow = 0; s.n_local = 0; s.n_q = 0; }
            // ---- insert: records are staged in LDS (first chunk: from the prefetch registers), work
            //      items are "quads" (record, 4 consecutive k-mers) listed densely, 4 lanes per quad
            for (uint64_t c0 = 0; c0 < r_n; c0 += STAGE) {
                const uint32_t n_st = (uint32_t)min((uint64_t)STAGE, r_n - c0);
                if (c0) __syncthreads();
                if (threadIdx.x == 0) s.n_flat = 0;
                if (c0 == 0 && have_prefetch) {
                    if (threadIdx.x < n_st) { s.q_key[threadIdx.x] = pf_w0; s.q_meta[threadIdx.x] = pf_w1 & ~BH_FIELD; s.st_stage[threadIdx.x] = pf_st; }
                } else {
                    for (uint32_t r = threadIdx.x; r < n_st; r += CNT_NT) {
                        s.q_key[r] = rec_w0[r_beg + c0 + r];
                        s.q_meta[r] = rec_w1[r_beg + c0 + r] & ~BH_FIELD;
                        s.st_stage[r] = rec_st[r_beg + c0 + r];
                    }
                }
                // dedupe scratch (eoff[] / list[] are free until the table is final): hash set of record indices in eoff[],
                // the quad list in list[]; the multiplicity of a representative counts up in the bucket-hash field of its
                // staged w1 (the insert reads that word anyway)
                uint32_t *dd_tab = reinterpret_cast<uint32_t *>(s.eoff);   // DD_SLOTS entries
                constexpr uint32_t DD_SLOTS = CAP / 2;                      // sizeof(eoff) / 4; >= 2 * STAGE
                static_assert(DD_SLOTS >= 2 * STAGE && STAGE < (1u << 20), "dedupe scratch");
                for (uint32_t i = threadIdx.x; i < DD_SLOTS; i += CNT_NT) dd_tab[i] = 0xFFFFFFFFu;
                __syncthreads();
                CNT_TICK(1);
                if (phase_limit == 1) { skip_rest = true; break; }  // clear + stage
                if (s.overflow) break;  // uniform: read after the barrier
                // ---- identical records (same window of the genome seen by several reads) collapse to one
                //      representative with a multiplicity and the smallest stamp: at 30x coverage this is
                //      most of the error-free data, so the per-k-mer table work drops by about that factor.
                //      A record knows that it is a representative when its own claim succeeds, and lists its quads
                //      of 4 k-mers in the same pass (STAGE <= CNT_NT: one record per thread).
                {
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
                                if (cur == 0xFFFFFFFFu) break;  // r is the representative
                            }
                            if (s.q_key[cur] == w0 && ((s.q_meta[cur] ^ w1) & ~BH_FIELD) == 0) { rep = cur; break; }
                            hslot = (hslot + 1) & (DD_SLOTS - 1);
                        }
                        atomicAdd(reinterpret_cast<uint32_t *>(&s.q_meta[rep]), 1u << 6);  // low dword: bits 27..6 = multiplicity
                        if (rep != r) atomicMin(&s.st_stage[rep], s.st_stage[r]);
                        else nquad = ((uint32_t)((w1 >> 1) & 31) + 4) >> 2;  // ceil(len / 4)
                    }
                    const uint32_t base = wave_alloc_n<5>(&s.n_flat, nquad);  // a record holds at most 19 k-mers: five quads
                    for (uint32_t q = 0; q < nquad; ++q) flat[base + q] = (uint16_t)((r << 3) | q);
                }
                __syncthreads();
                CNT_TICK(3);
                const uint32_t n_flat = s.n_flat;
                for (uint32_t f = threadIdx.x >> 2; f < n_flat; f += CNT_NT / 4) {
                    const uint32_t e = flat[f];
                    const uint32_t r = e >> 3;
                    const int i = (int)((e & 7) * 4 + (threadIdx.x & 3));
                    const uint64_t w0 = s.q_key[r], w1 = s.q_meta[r];
                    const int len = (int)((w1 >> 1) & 31) + 1;
                    if (i >= len) continue;
                    const ST st0 = s.st_stage[r];
                    const uint32_t mult = (uint32_t)(w1 >> 6) & ((1u << SK_BUCKET_BITS) - 1u);
                    const uint64_t hi = w1 & (~0ull << SK_META_BITS);
                    const uint32_t hs = (uint32_t)(w1 & 1);
                    const uint64_t win = rec_window(w0, hi, i);
                    const uint64_t kmer = win >> (64 - 2 * k);
                    if (cur_mask && (sub_hash(kmer) & cur_mask) != cur_val) continue;
                    const bool has_succ = (i < len - 1) || hs;
                    const uint32_t b = (uint32_t)(win >> (62 - 2 * k)) & 3u;
                    const ST stamp = i ? (ST)((st0 | (ST)1) + (ST)(2 * i)) : st0;
                    uint32_t slot = slot_of<CAP>(kmer);
                    bool ok = false;
                    // (unrolled 16-fold: the divergent exit of a rolled probe loop costs a scalar exec-mask round trip
                    //  per step -- rolled 14.4 ms, the compiler's own 4-fold 13.45, 16-fold 13.2, 32-fold 13.45)
#pragma unroll 16
                    for (int probe = 0; probe < CNT_PROBE_LIMIT; ++probe) {
                        unsigned long long cur = s.keys[slot];
                        if (cur == EMPTY_KEY) {
                            cur = atomicCAS(&s.keys[slot], EMPTY_KEY, (unsigned long long)kmer);
                            if (cur == EMPTY_KEY) cur = kmer;
                        }
                        if (cur == kmer) { ok = true; break; }
                        slot = (slot + 1) & (CAP - 1);
                    }
                    if (!ok) { s.overflow = 1; continue; }
                    if (has_succ) atomicAdd(&s.cnt[slot * 4 + b], mult);
                    atomicMin(&s.stamp[slot], stamp);
                }
            }
            if (skip_rest) break;
            CNT_TICK(4);
            __syncthreads();
            CNT_TICK(5);
            if (have_prefetch) {  // the registers are free: fetch the next bucket's records under the rest of this one
                have_prefetch = false;
                prefetch(bucket + gridDim.x);
            }
            if (phase_limit == 2) { skip_rest = true; break; }  // + insert
            if (s.overflow) {  // split this hash sub-range in two and retry (nothing was written out)
                const uint32_t bit = cur_mask + 1;  // masks are 2^j - 1
                if (stk_n + 2 > CNT_STACK || bit >= (1u << 20)) {
                    if (threadIdx.x == 0) atomicOr(&fresh_args(outp)->scalars[0], 8ull);  // bucket cannot be split further
                    failed = true;
                    break;
                }
                if (threadIdx.x == 0) {
                    s.stk_mask[stk_n] = cur_mask | bit; s.stk_val[stk_n] = cur_val;
                    s.stk_mask[stk_n + 1] = cur_mask | bit; s.stk_val[stk_n + 1] = cur_val | bit;
                }
                stk_n += 2;
                __syncthreads();  // entries visible to every thread's pop
                continue;
            }
            // ---- dense list of occupied slots + CSR edge offsets.  A wave looks at its CAP / CNT_NT blocks of 64 slots
            //      together: all their LDS reads go out back to back (one wait instead of one per block), and ONE packed LDS
            //      atomic hands the wave its node indices and the matching edge slots for all of them, so rows stay in
            //      node order (per block this was a chain of read -> ballots -> atomic round trip -> writes, four times)
            {
                constexpr int NB = CAP / CNT_NT;
                unsigned long long kk[NB];
                uint4 cc[NB];
#pragma unroll
                for (int t = 0; t < NB; ++t) {
                    const int i = threadIdx.x + t * CNT_NT;
                    kk[t] = s.keys[i];
                    cc[t] = reinterpret_cast<const uint4 *>(s.cnt)[i];
                }
                unsigned long long mask[NB];
                uint32_t below[NB], eexc[NB], nn[NB], ne[NB], tot = 0;
#pragma unroll
                for (int t = 0; t < NB; ++t) {
                    const bool occ = kk[t] != EMPTY_KEY;
                    const uint32_t deg = occ ? (cc[t].x != 0) + (cc[t].y != 0) + (cc[t].z != 0) + (cc[t].w != 0) : 0u;
                    mask[t] = __ballot(occ);
                    below[t] = lanes_below(mask[t]);
                    eexc[t] = 0;
                    ne[t] = 0;
#pragma unroll
                    for (int j = 1; j <= 4; ++j) {  // exclusive prefix of deg (0..4) over the wave: four ballots, no cross-lane scan
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
                    const int i = threadIdx.x + t * CNT_NT;
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
            if (phase_limit == 3) { skip_rest = true; break; }  // + dense list
            const uint32_t n_local = s.n_local & 0xFFFFu, n_edges_local = s.n_local >> 16;
            // ---- reservation of the node ids and CSR rows: one packed global atomic, issued now and consumed after
            //      the lookups (its ~1.5 us round trip hides behind them)
            unsigned long long got = 0;
            if (threadIdx.x == 0)
                got = atomicAdd(&fresh_args(outp)->scalars[4], (unsigned long long)n_local | ((unsigned long long)n_edges_local << 32));
            // ---- successor lookups into registers; misses are staged as queries (slot index local for now).
            //      Per node only the bases that occur are looked up (usually one): the wave loops
            //      max-popcount times instead of four.  Result per base, 16 bits: local node index,
            //      0xFFFF none, 0xFFFE miss staged, 0x8000 | (qi - CNT_QBUF) miss that did not fit the staging.
            unsigned long long nsucc[NPT];
#pragma unroll
            for (int u = 0; u < NPT; ++u) {
                nsucc[u] = ~0ull;
                if ((uint32_t)(u * CNT_NT) >= n_local) continue;  // uniform over the workgroup
                const uint32_t li = threadIdx.x + u * CNT_NT;
                unsigned long long key = 0;
                uint32_t nz = 0;
                if (li < n_local) {
                    const uint32_t sl = s.list[li];
                    key = s.keys[sl];
                    const uint4 c4 = reinterpret_cast<const uint4 *>(s.cnt)[sl];
                    nz = (c4.x != 0) | ((c4.y != 0) << 1) | ((c4.z != 0) << 2) | ((c4.w != 0) << 3);
                }
                const uint32_t nz_all = nz;
                uint32_t missmask = 0;
                if (u == 0) { CNT_TICK(8); CNT_SUBTICK(20); }
                while (nz) {
                    const uint32_t b = __ffs(nz) - 1;
                    nz &= nz - 1;
                    const int f = lds_find<CAP>(s.keys, ((key << 2) | (uint64_t)b) & kmask);
                    if (f >= 0) {  // slot -> local node index: rank of the slot among the occupied ones of its 64-slot block
                        const uint32_t ix = (uint32_t)s.dir_base[f >> 6] +
                                            (uint32_t)__popcll(s.dir_mask[f >> 6] & ((1ull << (f & 63)) - 1ull));
                        nsucc[u] = (nsucc[u] & ~(0xFFFFull << (16 * b))) | ((unsigned long long)ix << (16 * b));
                    }
                    else missmask |= 1u << b;
                }
                if (u == 0) CNT_SUBTICK(13);
                uint32_t qi = wave_alloc_n<4>(&s.n_q, (uint32_t)__popc(missmask));
                if (u == 0) CNT_SUBTICK(14);
                while (missmask) {
                    const uint32_t b = __ffs(missmask) - 1;
                    missmask &= missmask - 1;
                    unsigned long long code;
                    if (qi < CNT_QBUF) {
                        s.q_key[qi] = ((key << 2) | (uint64_t)b) & kmask;
                        s.q_meta[qi] = s.eoff[li] + __popc(nz_all & ((1u << b) - 1u));  // bucket-relative CSR column
                        code = 0xFFFEull;
                    } else {
                        code = 0x8000ull | (qi - CNT_QBUF);
                    }
                    nsucc[u] = (nsucc[u] & ~(0xFFFFull << (16 * b))) | (code << (16 * b));
                    ++qi;
                }
                if (u == 0) CNT_SUBTICK(15);
            }
            CNT_TICK(17);
            // ---- thread 0 turns the reservation into the bucket's bases before the barrier that ends the lookups: one
            //      barrier publishes both, and its serial work overlaps the other waves' lookups
            if (threadIdx.x == 0) {
                const auto &orr = *fresh_args(outp);
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
                    if (cur_mask) {  // link it into the bucket's chain (this workgroup is the only writer of the bucket's ranges)
                        rg.next = orr.ranges[bucket].next;
                        orr.ranges[bucket].next = (uint32_t)ri;
                    }
                    orr.ranges[ri] = rg;
                }
            }
            CNT_TICK(9);
            __syncthreads();
            CNT_TICK(10);
            if (phase_limit == 4) { skip_rest = true; break; }  // + successor lookups
            const uint32_t nq = s.n_q;
            unsigned long long qgot = 0;
            if (threadIdx.x == 64 && nq) qgot = atomicAdd(&fresh_args(outp)->scalars[5], (unsigned long long)nq);  // query cursor
            // (the minimizer bucket of every query is filled in by k_q_bucket afterwards: here it would occupy three
            //  waves for ~150 instructions per bucket while the other thirteen wait at the barrier)
            if (phase_limit == 5) { skip_rest = true; break; }  // + reservation
            if (s.fail) break;
            const uint64_t gbase = s.gbase, ebase = s.ebase;
            const auto &ow = *fresh_args(outp);  // loaded here, not kept in SGPRs across the whole bucket loop
            // ---- write nodes and their CSR rows: consecutive lanes -> consecutive nodes; every slot read is
            //      cleared for the next bucket
            if (threadIdx.x < CAP / 64) {  // the range's directory (k_succ_resolve)
                SkDirEnt de;
                de.mask = s.dir_mask[threadIdx.x];
                de.base = (uint32_t)(gbase + s.dir_base[threadIdx.x]);
                de.pad = s.ri < ow.n_buckets ? 1u : 0u;  // 1: the whole bucket is this one range (k_succ_resolve need not ask)
                const uint64_t di = s.ri < ow.n_buckets ? s.ri - ow.own_lo : ow.own_cnt + (s.ri - ow.n_buckets);
                ow.dirs[di * (CAP / 64) + threadIdx.x] = de;
            }
#pragma unroll
            for (int u = 0; u < NPT; ++u) {
                if ((uint32_t)(u * CNT_NT) >= n_local) break;
                const uint32_t li = threadIdx.x + u * CNT_NT;
                if (li >= n_local) continue;
                const uint32_t i = s.list[li];
                const unsigned long long key = s.keys[i];
                const uint64_t node = gbase + li;
                const uint4 c4 = reinterpret_cast<const uint4 *>(s.cnt)[i];
                const uint32_t c[4] = {c4.x, c4.y, c4.z, c4.w};
                const ST stamp = s.stamp[i];
                s.keys[i] = EMPTY_KEY;
                s.stamp[i] = (ST)~(ST)0;
                reinterpret_cast<uint4 *>(s.cnt)[i] = make_uint4(0, 0, 0, 0);
                ow.keys[node] = key;
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
            // ---- queries out (the cursor has had the whole node pass to come back)
            if (threadIdx.x == 64 && nq) {
                s.qbase = qgot;
                if (qgot + nq > (*fresh_args(outp)).q_cap || qgot + nq > 0xFFFFFFF0ull) { atomicOr(&(*fresh_args(outp)).scalars[0], 64ull); s.fail = 1; }
            }
            __syncthreads();
            CNT_TICK(12);
            if (s.fail) break;
            if (nq) {
                const auto &oq = *fresh_args(outp);
                const uint64_t qbase = s.qbase;
                for (uint32_t i = threadIdx.x; i < min(nq, (uint32_t)CNT_QBUF); i += CNT_NT) {
                    oq.q_key[qbase + i] = s.q_key[i];
                    oq.q_col[qbase + i] = (uint32_t)(ebase + s.q_meta[i]);
                }
                if (nq > (uint32_t)CNT_QBUF) {  // rare: queries that did not fit the staging, straight from the registers
#pragma unroll
                    for (int u = 0; u < NPT; ++u) {
                        const uint32_t li = threadIdx.x + u * CNT_NT;
                        if (li >= n_local) continue;
                        uint32_t rank = 0;
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            const uint32_t v = (uint32_t)(nsucc[u] >> (16 * b)) & 0xFFFFu;
                            if (v >= 0x8000u && v < 0xFFFEu) {
                                const uint64_t qi = (uint64_t)(v & 0x7FFFu) + CNT_QBUF;
                                const uint64_t node = gbase + li;
                                const uint64_t skey = ((oq.keys[node] << 2) | (uint64_t)b) & kmask;
                                oq.q_key[qbase + qi] = skey;
                                oq.q_col[qbase + qi] = (uint32_t)(ebase + s.eoff[li] + rank);
                            }
                            if (v != 0xFFFFu) ++rank;  // every base that occurs owns one CSR column
                        }
                    }
                }
            }
            clean = true;  // every occupied slot was reset above (uniform: all waves pass here)
        }
        if (failed || s.fail) return;  // s.fail was written before the last barrier every thread passed
        if (skip_rest) { clean = false; if (have_prefetch) prefetch(bucket + gridDim.x); }
    }
#ifdef DBG_CNT_PROF
    if (threadIdx.x == 0) {
        for (int i = 0; i < 31; ++i) atomicAdd(&g_cnt_prof[i], s.prof[i]);
        atomicAdd(&g_cnt_prof[31], 1ull);
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// K6: successors that live in another bucket, resolved by the asker.
// ------------------------------------------------------------------------------------------------
// How the 22-bit bucket hash of a minimizer maps to a bucket index (the two multisplit levels of the build).
struct SkGeom {
    int k, m;
    int l1;       // level 1: top l1 bits of the hash
    int nb2;      // children of level 2 (1: no second level)
    int fb2;      // != 0: level 2 scales the fb2 low bits of the hash into [0, nb2); 0: plain bit field of l2_pow bits
    int l2_pow;
    int nb3, fb3; // level 3 (nb3 > 1: level 2 is a plain l2_pow-bit field): the fb3 hash bits below it scaled into [0, nb3)
    int shard_bits, my_shard;  // sharded builds: owner = top shard_bits of the hash
    uint64_t own_lo, own_cnt;  // the buckets this build owns: [own_lo, own_lo + own_cnt) -- the directory holds only those
};

__host__ __device__ inline uint64_t sk_bucket_of(uint32_t bh, const SkGeom &g) {
    const uint32_t b1 = g.l1 ? bh >> (SK_BUCKET_BITS - g.l1) : 0u;
    if (g.nb2 <= 1 && g.nb3 <= 1) return b1;
    uint32_t b2 = 0;
    if (g.nb2 > 1) {
    if (g.fb2) b2 = (uint32_t)(((uint64_t)(bh & ((1u << g.fb2) - 1u)) * (uint32_t)g.nb2) >> g.fb2);
    else b2 = (bh >> (SK_BUCKET_BITS - g.l1 - g.l2_pow)) & (uint32_t)(g.nb2 - 1);
    }
    const uint64_t b12 = (uint64_t)b1 * (uint32_t)g.nb2 + b2;
    if (g.nb3 <= 1) return b12;
    const uint32_t b3 = (uint32_t)(((uint64_t)(bh & ((1u << g.fb3) - 1u)) * (uint32_t)g.nb3) >> g.fb3);
    return b12 * (uint32_t)g.nb3 + b3;
}

// sort key of the owner split of a sharded build: bucket hash of every query in bits 40.. of q_meta
__global__ __launch_bounds__(256) void k_q_bucket(const uint64_t *__restrict__ q_key, uint64_t *q_meta, uint64_t n, int k, int m) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) q_meta[i] = ((uint64_t)kmer_bucket22(q_key[i], k, m) << 40) | i;
}

// Node id of k-mer `key` in the counted range `ri`: the table's linear probing, replayed on the directory.  A key
// sits at or after its home slot with every slot in between occupied, and the nodes of a 64-slot block are stored
// in slot order from dir.base on -- so the probe is one directory entry (16 bytes) and a run of consecutive keys.
// `whole` (may be null): set to false when the first entry read is not marked as the directory of an unsplit bucket
// (the bucket was counted in hash sub-ranges, or is empty): the caller then goes through the ranges.
template <int CAP>
__device__ inline uint32_t dir_find(const SkDirEnt *__restrict__ dirs, uint64_t ri, const uint64_t *__restrict__ keys,
                                    uint64_t n_nodes, uint64_t key, bool *whole = nullptr) {
    constexpr int NBLK = CAP / 64;
    uint32_t slot = slot_of<CAP>(key);
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
            if (keys[idx + t] == key) return (uint32_t)(idx + t);
        if (run < avail) return NO_NODE;  // an empty slot ends the probe run
        slot = (slot + (uint32_t)run) & (CAP - 1);
    }
    return NO_NODE;
}

// out[q_col ? q_col[i] : i] = node id of q_key[i] (| id_tag).  One thread per query.
template <int CAP>
__global__ __launch_bounds__(256) void k_succ_resolve(const uint64_t *__restrict__ q_key, const uint32_t *__restrict__ q_col,
                                                      uint64_t n, SkGeom g, const SkRange *__restrict__ ranges,
                                                      uint64_t n_buckets, uint64_t n_ranges,
                                                      const SkDirEnt *__restrict__ dirs, const uint64_t *__restrict__ keys,
                                                      uint64_t n_nodes, uint32_t *out, uint32_t id_tag,
                                                      unsigned long long *scalars,
                                                      const uint64_t *__restrict__ q_meta /* NULL, or k_q_bucket's words: the bucket hash in bits 40.. */) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t key = q_key[i];
    const uint64_t bucket = sk_bucket_of(q_meta ? (uint32_t)(q_meta[i] >> 40) : kmer_bucket22(key, g.k, g.m), g);
    uint32_t id = NO_NODE;
    if (bucket >= g.own_lo && bucket < g.own_lo + g.own_cnt) {
        // common case: the bucket is one range and its directory says so -- two dependent reads (directory entry, key
        // run), each a 128-byte line of HBM; the ranges array is not touched (the resolver moves ~370 bytes per query
        // and runs at the HBM rate, so a line less per query is a quarter of its time)
        bool whole = true;
        id = dir_find<CAP>(dirs, bucket - g.own_lo, keys, n_nodes, key, &whole);
        uint64_t ri = bucket;
        bool have = false;
        if (!whole) {  // the bucket was counted in hash sub-ranges: walk its chain of ranges to the one of this key
            const uint32_t sh = sub_hash(key);
            uint32_t r = ranges[bucket].next;
            for (int guard = 0; r && r < n_ranges && !have && guard < (1 << 20); ++guard) {
                const SkRange rg = ranges[r];
                if (rg.node_cnt && (sh & rg.mask) == rg.val) { ri = r; have = true; }
                r = rg.next;
            }
        }
        if (have) id = dir_find<CAP>(dirs, g.own_cnt + (ri - n_buckets), keys, n_nodes, key);
    }
    if (id == NO_NODE) { atomicOr(&scalars[0], 128ull); return; }  // every successor k-mer exists as a node
    out[q_col ? q_col[i] : i] = id | id_tag;
}

// ------------------------------------------------------------------------------------------------
// Dense per-base views of the CSR the count kernel wrote (first use by a traversal kernel or an export)
// ------------------------------------------------------------------------------------------------
template <class ST>
__global__ __launch_bounds__(256) void k_dense_from_csr(uint64_t n_nodes, const ST *__restrict__ stamps_in,
                                                        const uint32_t *__restrict__ rowptr32, const uint32_t *__restrict__ col,
                                                        const uint32_t *__restrict__ ecnt, uint8_t *flags, uint64_t *stamps64,
                                                        uint64_t *rowptr64, uint32_t *cnt, uint32_t *succ, uint8_t *order) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const uint32_t f = flags[i];
    uint32_t e = rowptr32[i];
    uint32_t c[4], sc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        c[b] = 0;
        sc[b] = NO_NODE;
        if ((f >> (1 + b)) & 1u) { c[b] = ecnt[e]; sc[b] = col[e]; ++e; }
    }
    reinterpret_cast<uint4 *>(cnt)[i] = make_uint4(c[0], c[1], c[2], c[3]);
    reinterpret_cast<uint4 *>(succ)[i] = make_uint4(sc[0], sc[1], sc[2], sc[3]);
    flags[i] = (uint8_t)(f & 1u);
    if (stamps64) stamps64[i] = (uint64_t)stamps_in[i];
    rowptr64[i] = rowptr32[i];
    // rank of every code by (count descending, ASCII order A C G T = codes 0 1 3 2 ascending): six pair comparisons
    uint32_t r1 = 0, r2 = 0, r3 = 0;
    { const uint32_t q = c[0] >= c[1]; r1 += q; }
    { const uint32_t q = c[0] >= c[3]; r3 += q; }
    { const uint32_t q = c[0] >= c[2]; r2 += q; }
    { const uint32_t q = c[1] >= c[3]; r3 += q; r1 += 1u - q; }
    { const uint32_t q = c[1] >= c[2]; r2 += q; r1 += 1u - q; }
    { const uint32_t q = c[3] >= c[2]; r2 += q; r3 += 1u - q; }
    order[i] = (uint8_t)((1u << (2 * r1)) | (2u << (2 * r2)) | (3u << (2 * r3)));
}

}  // namespace dbgk
