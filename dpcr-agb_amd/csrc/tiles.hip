// Work-balanced tiles for the pair-compacted convolution kernel (k_spconv_cmp, spconv.hip).
// The kernel's time is set by its slowest tiles: with the fixed interleave (tile t = row blocks t, t + T, t + 2T, ...) the
// fullest of ~1000-2000 tiles holds 13-26 % more (row, offset) pairs than the mean on the NFI plots, and with one or two
// tiles per resident wave slot the kernel ends that much after a balanced schedule would (wave timelines: slots busy
// 86-89 % of the kernel).  Here the level's row blocks are ordered by pair count (counting sort: the count of a block of
// 2^il rows is at most K3 << il) and dealt to the tiles in serpentine order — rank j*T + t goes to tile t in even rounds,
// T-1-t in odd ones — so every tile gets the same number of blocks and, to ~1 %, the same number of pairs.
// Blocks of equal count land in an order that depends on atomics; no result depends on it: a row's sum is the same in
// whichever tile it is computed.  Reference: no counterpart (MinkowskiEngine schedules its gather-GEMM-scatter per offset).
#include "agb_common.h"

// thread per row: pairs of the row over all offsets, summed over the 2^il rows of its block by shuffles
__global__ __launch_bounds__(256) void k_tile_work(const int32_t* __restrict__ nbr, long long nbr_stride, int n_out, int K3,
                                                   int il, int32_t* __restrict__ work) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    int c = 0;
    if (row < n_out)
        for (int k = 0; k < K3; ++k) c += nbr[(long long)k * nbr_stride + row] >= 0;
    for (int d = 1; d < (1 << il); d <<= 1) c += __shfl_xor(c, d);
    if ((row & ((1 << il) - 1)) == 0 && row < n_out) work[row >> il] = c;
}

// one workgroup: histogram of the block counts (LDS), start[w] = blocks with a count above w (descending order), cursors
// cleared.  nbins <= 1024.
__global__ __launch_bounds__(1024) void k_tile_offsets(const int32_t* __restrict__ work, int nblk, int nbins,
                                                       int32_t* __restrict__ start, int32_t* __restrict__ cursor) {
    __shared__ int hist[1024];
    __shared__ int s[1024];
    const int t = threadIdx.x;
    hist[t] = 0;
    __syncthreads();
    for (int b = t; b < nblk; b += 1024) atomicAdd(&hist[work[b]], 1);
    __syncthreads();
    // thread t owns bin nbins-1-t (the fullest blocks first); inclusive scan, then shift
    const int v = t < nbins ? hist[nbins - 1 - t] : 0;
    s[t] = v;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int add = t >= d ? s[t - d] : 0;
        __syncthreads();
        s[t] += add;
        __syncthreads();
    }
    if (t < nbins) {
        start[nbins - 1 - t] = s[t] - v;
        cursor[nbins - 1 - t] = 0;
    }
}

// rank p -> tile: round j = p / T gives tile p % T its j-th block in even rounds, tile T-1 - p % T in odd ones; the
// ranks past the last block mark the empty slots.  Ranks inside a count bin: the workgroup counts its blocks per bin in
// LDS and claims one range per non-empty bin from the global cursor (a thread-per-block claim is ~26 k atomics on ~100
// addresses for the first level of the pyramid: 50 us).
__global__ __launch_bounds__(1024) void k_tile_deal(const int32_t* __restrict__ work, int nblk, int nbins,
                                                    const int32_t* __restrict__ start, int32_t* __restrict__ cursor, int ntiles,
                                                    int bpt, int32_t* __restrict__ tile_blocks) {
    __shared__ int s_cnt[1024];
    __shared__ int s_base[1024];
    const int t = threadIdx.x;
    const int b = blockIdx.x * 1024 + t;
    s_cnt[t] = 0;
    __syncthreads();
    int w = 0, local = 0;
    if (b < nblk) {
        w = work[b];
        local = atomicAdd(&s_cnt[w], 1);
    }
    __syncthreads();
    if (t < nbins && s_cnt[t] > 0) s_base[t] = start[t] + atomicAdd(&cursor[t], s_cnt[t]);
    __syncthreads();
    if (b >= ntiles * bpt) return;
    const int p = b < nblk ? s_base[w] + local : b;
    const int j = p / ntiles;
    int tl = p - j * ntiles;
    if (j & 1) tl = ntiles - 1 - tl;
    tile_blocks[(long long)tl * bpt + j] = b < nblk ? b : -1;
}

extern "C" {

// Scratch of agb_spconv_balance_tiles in bytes (int32: work[blocks], start / cursor[(K3 << il) + 1]).
size_t agb_spconv_balance_tiles_workspace_bytes(int n_out, int K3, int il) {
    if (n_out < 0 || K3 < 1 || il < 1 || il > 5) return 0;      // (the entry point's own contract: il 1..5)
    const size_t nblk = ((size_t)n_out + ((size_t)1 << il) - 1) >> il, nbins = ((size_t)K3 << il) + 1;
    if (nbins > 1024) return 0;                                  // more pairs per block than the counting sort has bins
    return 4 * (nblk + 2 * nbins + 16);
}

// tile_blocks int32[ntiles][bpt] (out): the row blocks (2^il rows each; -1 = none) of every tile, for agb_spconv_fwd_tiles.
// ntiles, bpt = rows per tile >> il and il come from agb_spconv_cmp_geometry for the calls that will use the table; one
// table serves every convolution on this map with that geometry, forward and (kflip) data gradient.
int agb_spconv_balance_tiles(const int32_t* nbr, long long nbr_stride, int n_out, int K3, int il, int ntiles, int bpt,
                             int32_t* tile_blocks, void* workspace, void* stream) {
    AGB_CHECK_ARG(nbr != nullptr && tile_blocks != nullptr && workspace != nullptr, "agb_spconv_balance_tiles: null argument");
    AGB_CHECK_ARG(n_out >= 1 && K3 >= 1 && il >= 1 && il <= 5 && nbr_stride >= n_out, "agb_spconv_balance_tiles: n %d, K3 %d, "
                  "block shift %d (1..5)", n_out, K3, il);
    const int nblk = (int)((n_out + (1LL << il) - 1) >> il), nbins = (K3 << il) + 1;
    AGB_CHECK_ARG(nbins <= 1024, "agb_spconv_balance_tiles: %d offsets x %d rows per block: more than 1023 pairs per block", K3,
                  1 << il);
    AGB_CHECK_ARG(ntiles >= 1 && bpt >= 1 && (long long)ntiles * bpt >= nblk, "agb_spconv_balance_tiles: %d tiles x %d blocks "
                  "cannot hold %d blocks", ntiles, bpt, nblk);
    hipStream_t s = (hipStream_t)stream;
    int32_t* work = (int32_t*)workspace;
    int32_t* start = work + nblk;
    int32_t* cursor = start + nbins;
    hipLaunchKernelGGL(k_tile_work, dim3(agb_cdiv(n_out, 256)), dim3(256), 0, s, nbr, nbr_stride, n_out, K3, il, work);
    hipLaunchKernelGGL(k_tile_offsets, dim3(1), dim3(1024), 0, s, work, nblk, nbins, start, cursor);
    hipLaunchKernelGGL(k_tile_deal, dim3(agb_cdiv((long long)ntiles * bpt, 1024)), dim3(1024), 0, s, work, nblk, nbins, start,
                       cursor, ntiles, bpt, tile_blocks);
    AGB_CHECK_LAUNCH("agb_spconv_balance_tiles");
    return AGB_OK;
}

}  // extern "C"
