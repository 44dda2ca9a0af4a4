// How long would the SORT at the heart of a sorted-bucket count kernel take?  (tools/probes: an experiment, not product code)
//
// The count kernels (dbg_sk.h, dbg_sk2.h) find equal k-mers with an open-address LDS table.  The alternative the reviews
// asked to see measured: expand a bucket's records into (k-mer, payload) tuples, SORT them in LDS, and let duplicates,
// counts, the node list and in-bucket successors fall out of neighbour comparisons.  This probe times only what that
// variant cannot avoid -- the sort of one bucket's tuples plus the neighbour pass -- with rocPRIM's tuned block radix sort
// (the best case for the variant), on the BASELINE.json configs[1] geometry: 245 248 buckets, ~2 300 distinct-record
// k-mer instances per bucket after the record-level dedupe (the hash kernels insert exactly those), 62-bit keys.
// Compare with k_sk_count2's insert + list phases (3.8 + 0.9 ms of its 12.2 ms): profiles/r03_count_variants.md.
//
// Build: hipcc --offload-arch=gfx950 -O3 -o sort_probe sort_probe.hip ; run: ./sort_probe [items_per_thread]
#include <hip/hip_runtime.h>
#include <rocprim/block/block_radix_sort.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int NT, int IPT, int BITS>
__global__ __launch_bounds__(NT) void k_sort_buckets(const unsigned long long *__restrict__ keys, int per_bucket, int n_buckets,
                                                     unsigned int *nodes_out) {
    using Sort = rocprim::block_radix_sort<unsigned long long, NT, IPT, unsigned int>;
    __shared__ typename Sort::storage_type st;
    __shared__ unsigned long long edge[NT];
    unsigned int total = 0;
    for (int b = blockIdx.x; b < n_buckets; b += gridDim.x) {
        unsigned long long k[IPT];
        unsigned int v[IPT];
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const int j = threadIdx.x * IPT + i;
            k[i] = j < per_bucket ? keys[(size_t)b * per_bucket + j] : ~0ull;   // padding sorts last
            v[i] = (unsigned int)j;                                              // payload: instance index (stamp, base, multiplicity live there)
        }
        Sort().sort(k, v, st, 0, BITS);
        // neighbour pass: a node starts where the key differs from its predecessor (the variant's "insert")
        edge[threadIdx.x] = k[IPT - 1];
        __syncthreads();
        unsigned int heads = 0;
        unsigned long long prev = threadIdx.x ? edge[threadIdx.x - 1] : ~0ull;
#pragma unroll
        for (int i = 0; i < IPT; ++i) { heads += (k[i] != prev && k[i] != ~0ull); prev = k[i]; }
        total += heads + (v[0] & 1u);
        __syncthreads();
    }
    atomicAdd(nodes_out, total);
}

template <int NT, int IPT>
static void run(const unsigned long long *d_keys, int per_bucket, int n_buckets, unsigned int *d_out, int wg_per_cu, const char *what) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int grid = 256 * wg_per_cu;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(d_out, 0, 4);
        hipEventRecord(a);
        hipLaunchKernelGGL((k_sort_buckets<NT, IPT, 62>), dim3(grid), dim3(NT), 0, 0, d_keys, per_bucket, n_buckets, d_out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        unsigned int out = 0;
        hipMemcpy(&out, d_out, 4, hipMemcpyDeviceToHost);
        if (rep == 2) printf("%-44s %d threads x %d items, %d workgroup(s)/CU: %8.2f ms  (checksum %u)\n", what, NT, IPT, wg_per_cu, ms, out);
    }
}

int main(int argc, char **argv) {
    const int n_buckets = 245248, per_bucket = argc > 1 ? atoi(argv[1]) : 2304;
    // keys: 62-bit values; a third of the instances repeat an earlier key of their bucket (30x coverage after the record dedupe)
    std::vector<unsigned long long> h((size_t)n_buckets * per_bucket);
    unsigned long long x = 88172645463325252ull;
    for (size_t i = 0; i < h.size(); ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const size_t j = i % per_bucket;
        h[i] = (j > 8 && (x & 3) == 0) ? h[i - 1 - (x >> 40) % 8] : (x >> 2);
    }
    unsigned long long *d_keys; unsigned int *d_out;
    if (hipMalloc(&d_keys, h.size() * 8) != hipSuccess || hipMalloc(&d_out, 4) != hipSuccess) return 2;
    hipMemcpy(d_keys, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    printf("sort probe: %d buckets x %d tuples (u64 key, u32 payload), 62 key bits\n", n_buckets, per_bucket);
    run<1024, 3>(d_keys, per_bucket, n_buckets, d_out, 1, "block radix sort + neighbour pass");
    run<512, 5>(d_keys, per_bucket, n_buckets, d_out, 2, "block radix sort + neighbour pass");
    run<256, 9>(d_keys, per_bucket, n_buckets, d_out, 4, "block radix sort + neighbour pass");
    return 0;
}
