// sort.h -- stable ascending sort of token ids with their positions (round 6; VERDICT r5 item 8d: the last ATen arithmetic of the backward).
//
// The embedding gradient is a segmented sum: table[id] += rows of the tokens that carry `id` (reference: nn.Embedding's backward behind
// x_clip.py:320; the position-table gather of x_clip.py:382-385).  tokens.h scatter_add_sorted_kernel sums a run of equal ids in registers and
// needs the ids ASCENDING with the positions they came from -- and in a FIXED order inside a run, or the fp32 sums change bits from launch to
// launch.  This is the sort: least-significant-digit radix passes of 8 bits over (id, position) pairs, each pass stable:
//   sort_hist_kernel     a work-group counts the digits of its SORT_BLOCK pairs (integer LDS atomics: order-independent)   -> hist[block][digit]
//   sort_scan_kernel     one work-group: hist[block][digit] <- pairs with a smaller digit + same digit in earlier blocks
//   sort_scatter_kernel  a work-group ranks its pairs among the equal digits BEFORE them in the block and writes them to their places
// A pass moves 8 bytes per pair twice; 262 k ids below 2^16 are two passes, six launches of a few microseconds.
// The rank inside a block: the pairs are visited in index order, 64 at a time (one wave, one round); the lanes of a wave with the same digit find
// each other with 8 ballots (bit b of the digit: keep the lanes that agree), the lowest of them advances the digit's LDS counter by the size of the
// group -- one (round, wave) slice after the other, a barrier between them, which is what makes the pass stable.
#pragma once
#include "common.h"

namespace xc {

constexpr int SORT_BLOCK = 1024;                               // pairs per work-group: 256 threads x 4 rounds
constexpr int SORT_BINS = 256;

// pair = id (low 32 bits of the int64) << 32 | position
XC_DEV uint64_t sort_pair_first(const long long* ids, long i) { return ((uint64_t)(uint32_t)ids[i] << 32) | (uint64_t)(uint32_t)i; }

template <bool FIRST>
__global__ __launch_bounds__(256) void sort_hist_kernel(const long long* __restrict__ ids, const uint64_t* __restrict__ in, long n, int shift,
                                                        int* __restrict__ hist) {
    XC_LDS_DYNAMIC(lds);
    int* const cnt = reinterpret_cast<int*>(lds);              // [SORT_BINS]
    const int tid = threadIdx.x;
    cnt[tid] = 0;
    sync();
    const long base = (long)blockIdx.x * SORT_BLOCK;
#pragma unroll
    for (int r = 0; r < SORT_BLOCK / 256; ++r) {
        const long i = base + r * 256 + tid;
        if (i < n) {
            const uint64_t pr = FIRST ? sort_pair_first(ids, i) : in[i];
            lds_atomic_add(cnt + (int)((pr >> (32 + shift)) & (SORT_BINS - 1)), 1);
        }
    }
    sync();
    hist[(long)blockIdx.x * SORT_BINS + tid] = cnt[tid];
}

// one work-group of 4 x SORT_BINS threads: thread (q, d) owns digit d over the q-th quarter of the blocks.  hist[block][digit] in, offs[block][digit]
// out (another buffer: the loads of a batch are all in flight before its first store)
__global__ __launch_bounds__(1024) void sort_scan_kernel(const int* __restrict__ hist, int* __restrict__ offs, int nblk) {
    XC_LDS_DYNAMIC(lds);
    int* const part = reinterpret_cast<int*>(lds);             // [4][SORT_BINS] pairs of digit d in quarter q
    int* const tot = part + 4 * SORT_BINS;                     // [SORT_BINS]
    const int d = threadIdx.x & (SORT_BINS - 1), q = threadIdx.x >> 8;
    const int per = (nblk + 3) >> 2, b0 = q * per, b1 = b0 + per < nblk ? b0 + per : nblk;
    int sum = 0;
    for (int b = b0; b < b1; b += 8) {
        int t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = b + k < b1 ? hist[(long)(b + k) * SORT_BINS + d] : 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) sum += t[k];
    }
    part[q * SORT_BINS + d] = sum;
    sync();
    if (q == 0) tot[d] = part[d] + part[SORT_BINS + d] + part[2 * SORT_BINS + d] + part[3 * SORT_BINS + d];
    sync();
    int run = 0;                                               // pairs with a smaller digit, or the same digit in an earlier quarter
    for (int k = 0; k < d; ++k) run += tot[k];
    for (int k = 0; k < q; ++k) run += part[k * SORT_BINS + d];
    for (int b = b0; b < b1; b += 8) {
        int t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = b + k < b1 ? hist[(long)(b + k) * SORT_BINS + d] : 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (b + k < b1) offs[(long)(b + k) * SORT_BINS + d] = run;
            run += t[k];
        }
    }
}

template <bool FIRST, bool LAST>
__global__ __launch_bounds__(256) void sort_scatter_kernel(const long long* __restrict__ ids, const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
                                                           long long* __restrict__ out_ids, long long* __restrict__ out_pos, long n, int shift,
                                                           const int* __restrict__ offs) {
    XC_LDS_DYNAMIC(lds);
    int* const cnt = reinterpret_cast<int*>(lds);              // [SORT_BINS] pairs of the digit seen so far in this block
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
    cnt[tid] = 0;
    sync();
    const long base = (long)blockIdx.x * SORT_BLOCK;
    const int* const my_offs = offs + (long)blockIdx.x * SORT_BINS;
    for (int r = 0; r < SORT_BLOCK / 256; ++r) {
        const long i = base + r * 256 + tid;
        const bool valid = i < n;
        const uint64_t pr = valid ? (FIRST ? sort_pair_first(ids, i) : in[i]) : 0;
        const int d = (int)((pr >> (32 + shift)) & (SORT_BINS - 1));
        uint64_t m = wave_ballot64(valid);                     // the lanes of this wave with the same digit
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = ((d >> b) & 1) != 0;
            const uint64_t bb = wave_ballot64(valid && bit);
            m &= bit ? bb : ~bb;
        }
        const int before = popc64(m & ((1ull << lane) - 1)), group = popc64(m);
        int rank = 0;
        for (int w = 0; w < 4; ++w) {                          // one slice of 64 after the other: that is the stability
            if (wave == w) {
                const int seen = valid ? cnt[d] : 0;
                wave_sync();                                   // (every lane of the group has read the counter before its first lane moves it)
                if (valid && before == 0) cnt[d] = seen + group;
                rank = seen + before;
            }
            sync();
        }
        if (valid) {
            const long dst = (long)my_offs[d] + rank;
            if (LAST) {
                out_ids[dst] = (long long)(pr >> 32);
                out_pos[dst] = (long long)(pr & 0xffffffffull);
            } else {
                out[dst] = pr;
            }
        }
    }
}

}  // namespace xc
