// Device-side helpers shared by the gfx950 kernels of the de Bruijn hot path.
// Wavefront width is 64 everywhere in this file.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dbgk {

constexpr uint64_t EMPTY_KEY = ~0ull;
constexpr uint32_t NO_NODE = 0xFFFFFFFFu;

// One hash slot: 32 bytes, so the key probe, the stamp min and the successor
// counter add of one record touch a single 128-byte line.
struct __attribute__((aligned(32))) Slot {
    unsigned long long key;    // 2-bit packed k-mer, EMPTY_KEY when free
    unsigned long long stamp;  // min over occurrences of (byte offset << 1 | pos != 0); node id after compaction
    unsigned int cnt[4];       // occurrences of (k-mer, successor base code)
};
static_assert(sizeof(Slot) == 32, "slot layout");

__host__ __device__ inline uint64_t mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}

// Hash of a packed k-mer; slot = hash >> (64 - log2(capacity)).
__host__ __device__ inline uint64_t kmer_hash(uint64_t key) {
    return mix64(key * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull);
}

// 4 ASCII bases in one dword (lowest address = lowest byte) -> 8 bits, first base in bits 7:6.
__device__ inline uint32_t pack4(uint32_t d) {
    return (((d >> 1) & 0x03030303u) * 0x40100401u) >> 24;
}

// per-byte exact zero detector (no cross-byte carries): 0x80 in every zero byte of v
__device__ inline uint32_t zero_bytes(uint32_t v) {
    return ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) & 0x80808080u;
}

// 0x80 in every byte of d that is one of 'A','C','G','T'
__device__ inline uint32_t acgt_bytes(uint32_t d) {
    return zero_bytes(d ^ 0x41414141u) | zero_bytes(d ^ 0x43434343u) |
           zero_bytes(d ^ 0x47474747u) | zero_bytes(d ^ 0x54545454u);
}

__device__ inline char code_to_ascii(uint32_t code) {  // inverse of (ascii >> 1) & 3
    return (char)((0x47544341u >> (8 * code)) & 0xFF);  // 0:A 1:C 2:T 3:G
}

// rank of a base code in ASCII order (A < C < G < T) -- tie order of equal edge counts
__device__ inline uint32_t code_ascii_rank(uint32_t code) { return (0x2310u >> (4 * code)) & 3u; }  // 0->0,1->1,2->3,3->2

// Exclusive scan of one value per thread over a block of NW waves of 64.
// Returns the exclusive prefix; *block_total receives the block sum (all threads).
template <int NW>
__device__ inline uint64_t block_exscan(uint64_t v, uint64_t *block_total) {
    __shared__ uint64_t wave_sum[NW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wave_sum[wave] = inc;
    __syncthreads();
    uint64_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        uint64_t s = wave_sum[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *block_total = tot;
    return base + inc - v;
}
__device__ inline uint64_t block_exscan_256(uint64_t v, uint64_t *block_total) { return block_exscan<4>(v, block_total); }

__device__ inline uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
    return v;  // valid in lane 0
}

// ------------------------------------------------------------------------------------------
// tile loader shared by k_count and k_pull_reads: TILE positions + 64 halo, as a big-endian
// 2-bit stream in LDS (16 bases per dword) plus the matching slice of the read-start bitmap
// ------------------------------------------------------------------------------------------
constexpr int TILE = 8192;
constexpr int HALO = 64;
constexpr int PK_WORDS = (TILE + HALO) / 16 + 2;  // +2: window reads touch word+2
constexpr int SB_WORDS = (TILE + HALO) / 32 + 2;

// HALO_: positions readable beyond the tile (a window of 32 bases / 64 start bits at tile position j reaches j + 63;
// the wide super-k-mer extraction looks k - 13 minimizer windows further: dbg_wsk.h uses 128)
template <int HALO_>
struct TileLdsT {
    static constexpr int SPAN = TILE + HALO_;
    static constexpr int PK = SPAN / 16 + 2;  // +2: window reads touch word+2
    static constexpr int SB = SPAN / 32 + 2;
    uint32_t pk[PK];
    uint32_t sb[SB];
};
using TileLds = TileLdsT<HALO>;
static_assert(TileLds::PK == PK_WORDS && TileLds::SB == SB_WORDS, "tile layout");

// returns nonzero if a byte outside ACGT was seen among the bytes < n_bytes
template <class T>
__device__ inline uint32_t load_tile(T &t, const char *bases, uint64_t n_bytes, const uint32_t *startbits,
                                     uint64_t tile0) {
    uint32_t bad = 0;
    for (int v = threadIdx.x; v < T::PK; v += blockDim.x) {
        const uint64_t off = tile0 + (uint64_t)v * 16;
        uint4 q = make_uint4(0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u);  // 'A' padding
        if (v < T::SPAN / 16 && off < n_bytes) {
            if (off + 16 <= n_bytes) {
                q = *reinterpret_cast<const uint4 *>(bases + off);
            } else {
                uint32_t w[4] = {0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u};
                for (int b = 0; b < 16 && off + b < n_bytes; ++b) {
                    w[b >> 2] &= ~(0xFFu << (8 * (b & 3)));
                    w[b >> 2] |= (uint32_t)(uint8_t)bases[off + b] << (8 * (b & 3));
                }
                q = make_uint4(w[0], w[1], w[2], w[3]);
            }
            const uint32_t ok = acgt_bytes(q.x) & acgt_bytes(q.y) & acgt_bytes(q.z) & acgt_bytes(q.w);
            bad |= (ok != 0x80808080u);
        }
        t.pk[v] = (pack4(q.x) << 24) | (pack4(q.y) << 16) | (pack4(q.z) << 8) | pack4(q.w);
    }
    const uint64_t w0 = tile0 >> 5;
    for (int v = threadIdx.x; v < T::SB; v += blockDim.x) t.sb[v] = startbits[w0 + v];
    return bad;
}

// 32 bases starting at tile-relative position j, first base in bits 63:62
template <class T>
__device__ inline uint64_t window32(const T &t, int j) {
    const int w = j >> 4, sh = (j & 15) * 2;
    const uint64_t hi = ((uint64_t)t.pk[w] << 32) | t.pk[w + 1];
    const uint64_t lo = (uint64_t)t.pk[w + 2] << 32;
    return sh ? (hi << sh) | (lo >> (64 - sh)) : hi;
}

// read-start bits of positions j .. j+31 (bit 0 = position j)
template <class T>
__device__ inline uint32_t startwin32(const T &t, int j) {
    const int w = j >> 5, sh = j & 31;
    const uint64_t both = ((uint64_t)t.sb[w + 1] << 32) | t.sb[w];
    return (uint32_t)(both >> sh);
}


// Sliding-window OR over a bit mask: bit q of the result = OR of bits q .. q + WIDTH - 1 of x (bits beyond the word read
// as 0), by doubling -- about ten operations for all 64 positions.  The extraction kernels use it for "a read starts
// inside the k-mer at position q", which they used to test position by position (a third of their instructions).
template <int WIDTH>
__device__ inline uint64_t window_or64(uint64_t x) {
    static_assert(WIDTH >= 1 && WIDTH <= 64, "window width");
    uint64_t y = x;
    int have = 1;
    if (WIDTH >= 2) { y |= y >> 1; have = 2; }
    if (WIDTH >= 4) { y |= y >> 2; have = 4; }
    if (WIDTH >= 8) { y |= y >> 4; have = 8; }
    if (WIDTH >= 16) { y |= y >> 8; have = 16; }
    if (WIDTH >= 32) { y |= y >> 16; have = 32; }
    if (WIDTH > have) y |= y >> (WIDTH - have);
    return y;
}

// the same over a 128-bit mask (lo, hi): only the low word of the result is returned (positions 0..63), WIDTH <= 64
struct Bits128 { uint64_t lo, hi; };
__device__ inline Bits128 shr128(Bits128 a, int s) {  // 0 < s < 64
    return Bits128{(a.lo >> s) | (a.hi << (64 - s)), a.hi >> s};
}
template <int WIDTH>
__device__ inline uint64_t window_or128_lo(Bits128 x) {
    static_assert(WIDTH >= 1 && WIDTH <= 64, "window width");
    Bits128 y = x;
    int have = 1;
    auto step = [&](int s) { const Bits128 t = shr128(y, s); y.lo |= t.lo; y.hi |= t.hi; };
    if (WIDTH >= 2) { step(1); have = 2; }
    if (WIDTH >= 4) { step(2); have = 4; }
    if (WIDTH >= 8) { step(4); have = 8; }
    if (WIDTH >= 16) { step(8); have = 16; }
    if (WIDTH >= 32) { step(16); have = 32; }
    if (WIDTH >= 64) { step(32); have = 64; }
    if (WIDTH > have) step(WIDTH - have);
    return y.lo;
}

}  // namespace dbgk
