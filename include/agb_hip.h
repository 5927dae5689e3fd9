/* agb_hip.h — C ABI of libagbhip.so: the MI355X (gfx950) drop-in kernels for the point-cloud encoder hot path
 * of StefOe/DPCR-AGB.  Plain pointers and sizes only; no torch types.
 *
 * Conventions
 *   - every function returns 0 on success or a negative code (AGB_E*); agb_last_error() gives the message
 *     (thread-local);
 *   - all pointers are DEVICE pointers unless a parameter is documented as host;
 *   - the caller owns every buffer; nothing is allocated, freed or synchronised inside — kernels are enqueued on
 *     `stream` (a hipStream_t passed as void*; NULL = the default stream);
 *   - variable-size results are two-phase: the count lands in a device int that the caller reads back;
 *   - feature matrices are row-major float32 with an explicit leading dimension (ld*, in floats).
 *
 * Each entry point cites the reference interface it replaces (paths relative to
 * /root/reference/torch-points3d/torch_points3d/).  MinkowskiEngine itself is an un-vendored dependency of
 * the reference: for those entry points the citation is the reference's CALL SITE of the ME operator.
 */
#ifndef AGB_HIP_H
#define AGB_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AGB_OK 0
#define AGB_EINVAL (-1)
#define AGB_ELAUNCH (-2)
#define AGB_ERANGE (-3)
#define AGB_EUNSUPPORTED (-4)

const char* agb_last_error(void);
/* Diagnostic: name of the compute kernel the last convolution / dense-product / weight-gradient entry point of the calling
 * thread launched (e.g. "k_spconv_cmp<128>", "k_spconv_pipe_b16<128, false, 64, true>", "k_dense_stream<2, false>"): the
 * bench tools label a timed launch with what actually ran (rocprof shows the same names) instead of mirroring dispatch rules. */
const char* agb_last_kernel(void);

/* ---------------------------------------------------------------------------------------------------------
 * Sparse-voxel coordinates (replaces ME's coordinate manager behind
 *   models/instance/minkowski.py:74   ME.SparseTensor(features=, coordinates=int32[N,1+3], device=)
 *   modules/MinkowskiEngine/SENet.py:47-53, resnet_block.py:48-55   strided convolution / pooling maps)
 * coords: int32 [n,4] = (batch, x, y, z), rows ordered by batch.  Hash table: keys uint64[cap],
 * vals int32[cap], cap = agb_hash_capacity(n) (power of two >= 2n).
 * n_dev (optional, may be NULL): device int with the live row count when n is only an upper bound.
 * --------------------------------------------------------------------------------------------------------- */
int agb_hash_capacity(int n);          /* host helper */
int agb_scan_scratch_elems(int n);     /* host helper: int32 scratch elements agb_coords_stride needs */
int agb_hash_clear(uint64_t* keys, int32_t* vals, int cap, void* stream);

/* Insert level-0 coordinates. slot_of_row int32[n] (scratch/out). status int32[4] (out):
 * [0] duplicate rows, [1] rows outside the packed 16-bit range, [2] rows breaking batch order. */
int agb_coords_insert(const int32_t* coords, int n, const int32_t* n_dev, uint64_t* keys, int32_t* vals, int cap,
                      int32_t* slot_of_row, int32_t* status, void* stream);

/* Strided level: out_coords = unique(floor(c / ts_out) * ts_out) in first-occurrence order (deterministic),
 * hash of the new level in keys/vals, *n_out_dev = number of rows.  slot_of_row, flags, excl: int32[n] scratch;
 * scratch: int32[agb_scan_scratch_elems(n)]; out_coords: int32[n,4] (upper bound); out_row_of_in: int32[n] or NULL. */
int agb_coords_stride(const int32_t* in_coords, int n, const int32_t* n_dev, int ts_out, uint64_t* keys,
                      int32_t* vals, int cap, int32_t* slot_of_row, int32_t* flags, int32_t* excl, int32_t* scratch,
                      int32_t* out_coords, int32_t* n_out_dev, int32_t* out_row_of_in, void* stream);

/* Kernel map as a dense neighbour table: nbr[k*nbr_stride + r] = row (in the hashed level) of
 * q_coords[r] + sign*offset_k*step, or -1.  offset_k, k = ix + K*(iy + K*iz): odd K -> (ix-K/2,..), even K -> (ix,..).
 * Forward map of a conv/pool in->out:  q = out coords, table = in level, sign=+1, require_multiple_of=0.
 * Transposed map (data gradients):     q = in coords, table = out level, sign=-1, require_multiple_of=ts_out.
 * pair_count (optional): device uint64[64*16], zero-filled by the caller: sharded counters (one per 128-B line) whose
 * SUM is the number of non-empty entries (ME's kernel-map size). */
int agb_kernel_map(const int32_t* q_coords, int n, const int32_t* n_dev, int K, int step, int sign,
                   int require_multiple_of, const uint64_t* keys, const int32_t* vals, int cap, int32_t* nbr,
                   long long nbr_stride, unsigned long long* pair_count, void* stream);

/* Dense-grid mode (small bounding volumes, e.g. LiDAR plots): every level keeps int32 grid[B][Z][Y][X]
 * (x fastest, INT_MAX = empty) described by desc = {ox, oy, oz, X, Y, Z, ts, B | halo << 16} (HOST int32[8]; origin a
 * multiple of ts; halo = margin of cells on every side that stays empty — rows falling there are reported as out of
 * range like rows outside the grid — so that kernels probing grid[cell + delta] directly never leave the plot's block).
 * Same results as the hash entry points above, one load per probe. */
int agb_coords_bbox(const int32_t* coords, int n, const int32_t* n_dev, int32_t* bbox /* dev int32[8]:
                    min x,y,z, max x,y,z, max batch */, void* stream);
int agb_grid_insert(const int32_t* coords, int n, const int32_t* n_dev, const int32_t* desc, int32_t* grid,
                    long long* cell_of_row /* [n] scratch */, int32_t* status, void* stream);
int agb_grid_stride(const int32_t* in_coords, int n, const int32_t* n_dev, const int32_t* desc /* OUTPUT level */,
                    int32_t* grid, long long* cell_of_row, int32_t* flags, int32_t* excl, int32_t* scratch,
                    int32_t* out_coords, int32_t* n_out_dev, int32_t* status, void* stream);
int agb_grid_kernel_map(const int32_t* q_coords, int n, const int32_t* n_dev, int K, int step, int sign,
                        const int32_t* desc /* PROBED level */, const int32_t* grid, int32_t* nbr,
                        long long nbr_stride, unsigned long long* pair_count, void* stream);

/* Row range of every batch element: ptr int32[B+1]. */
int agb_batch_ptr(const int32_t* coords, int n, const int32_t* n_dev, int B, int32_t* ptr, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Sparse convolution (replaces ME.MinkowskiConvolution forward/backward at
 *   modules/MinkowskiEngine/common.py:215-226, resnet_block.py:48-55,95-107, SENet.py:47-52,93-99)
 * W: [K3*Cin, Cout] = ME's kernel [K3, Cin, Cout] flattened.  Cin, Cout, ldx multiples of 4
 * (Cin == 4 or 8 selects the small-Cin/stem kernel; pad 3 -> 4 channels).
 *   Y[r,:]  = bias + sum_k X[nbr[k][r],:] @ W[k]
 * Data gradient = the same call with X=dY, W = W^T per offset ([K3*Cout, Cin]) and either the transposed map or,
 * for stride 1 / odd K, the forward map with kflip=1 (reads nbr[K3-1-k]).
 * --------------------------------------------------------------------------------------------------------- */
int agb_spconv_fwd(const float* X, int ldx, const float* W, const int32_t* nbr, long long nbr_stride, int kflip,
                   const float* bias, float* Y, int ldy, int n_out, int K3, int Cin, int Cout, void* stream);

/* Extended form.  perm/tile_cls/cls_tab/n_tiles (all or none): class-partitioned output rows from
 * agb_parity_partition — every 64-row tile holds rows of one lattice-parity class and visits only the kernel offsets
 * listed for it in cls_tab int32[classes][1+K3] (count, offsets...): the data gradient of a stride-s operator.
 * ksplit > 1: the offset/channel chunks are split over ksplit workgroups per tile, partials in
 * `partial` float[ksplit][n_out][Cout], folded in order (deterministic); agb_spconv_split_hint suggests ksplit. */
int agb_spconv_fwd_ex(const float* X, int ldx, const float* W, const int32_t* nbr, long long nbr_stride, int kflip,
                      const float* bias, float* Y, int ldy, int n_out, int K3, int Cin, int Cout,
                      const int32_t* perm, const int32_t* tile_cls, const int32_t* cls_tab, int n_tiles, int ksplit,
                      float* partial, void* stream);
int agb_spconv_split_hint(int n_out, int K3, int Cin, int Cout);  /* host helper (automatic kernel choice) */
/* the same for the identity map (nbr == NULL, a dense product): splits of the reduction dimension Cin */
int agb_dense_split_hint(int n_out, int Cin, int Cout);
/* Dense fp32 product Y = X W + bias that also leaves the partial BatchNorm statistics of its output — (count, mean, M2)
 * of every column per row tile, bn_part float[agb_dense_bn_chunks(...)][3][Cout] — for agb_bn_stats_fold: the
 * BatchNorm behind a Linear / 1x1 layer (PointNet.py:16-28, blocks.py:499-535) then never re-reads the layer output for
 * its statistics.  agb_dense_bn_chunks returns 0 for shapes that take a path without that epilogue. */
int agb_dense_bn_chunks(int n_out, int Cin, int Cout);
int agb_dense_fwd_bn(const float* X, int ldx, const float* W, const float* bias, float* Y, int ldy, int n_out, int Cin,
                     int Cout, float* bn_part, void* stream);
/* The same product with the kernel choice as per-call arguments (the library keeps no tuning state):
 * cmp_mode: 1 = automatic (the pair-compacted LDS-accumulating kernel for many-row layers with Cin % 64 == 0, the
 * register-accumulator kernels otherwise; what agb_spconv_fwd / _ex use), 0 = never the pair-compacted kernel,
 * 64 / 128 = always, with that many rows per wave (128-row tiles of maps with three or more offsets run the hand-scheduled
 * kernel k_spconv_cma, csrc/gen_cmp_asm.py), 129 = 128-row tiles on its C++ twin k_spconv_cmpt (the same sums bit for bit:
 * tests).  cmp_interleave_shift: tiles of the pair-compacted kernel made of
 * 2^shift-row blocks taken from regions ntiles blocks apart (0: contiguous row tiles; -1: chosen by the number of rows):
 * evens out the per-tile work where the pair density varies by region.  All choices compute the same sums; tile
 * interleaving is bit-identical, kernel choice changes only the fp32 summation order.
 * nbr == NULL (K3 == 1, Cin >= 12): the identity map — a dense [n, Cin] x [Cin, Cout] product, i.e. ME's 1x1 stride-1
 * convolution (kernel [Cin, Cout]; resnet_block.py:95-107 Bottleneck conv1 / conv3) and every nn.Linear of the path. */
int agb_spconv_fwd_opt(const float* X, int ldx, const float* W, const int32_t* nbr, long long nbr_stride, int kflip,
                       const float* bias, float* Y, int ldy, int n_out, int K3, int Cin, int Cout,
                       const int32_t* perm, const int32_t* tile_cls, const int32_t* cls_tab, int n_tiles, int ksplit,
                       float* partial, int cmp_mode, int cmp_interleave_shift, void* stream);
int agb_spconv_split_hint_opt(int n_out, int K3, int Cin, int Cout, int cmp_mode);  /* host helper */
/* WORK-BALANCED tiles for the pair-compacted kernel.  Its time is set by its fullest tiles (with one or two tiles per
 * resident wave the kernel ends 13-26 % after a balanced schedule on the NFI plots); agb_spconv_balance_tiles orders the
 * row blocks of a level by their pair count and deals them to the tiles so that all tiles hold the same number of
 * blocks and, to ~1 %, of pairs.  No counterpart in the reference (ME schedules gather-GEMM-scatter per kernel offset);
 * the sums are bit-identical to agb_spconv_fwd_opt's — a row's sum does not depend on the tile that computes it.
 *   agb_spconv_cmp_geometry  host helper: out[0] = tile height (0: another kernel takes this call), out[1] = rows per
 *                            tile, out[2] = tiles, out[3] = log2 rows per block (0: contiguous tiles, no table)
 *   agb_spconv_balance_tiles tile_blocks int32[ntiles][bpt] (bpt = rows per tile >> il; -1 = no block) from the kernel map;
 *                            workspace of agb_spconv_balance_tiles_workspace_bytes bytes.  One table serves every
 *                            stride-1 convolution on that map with that geometry, forward and data gradient (kflip).
 *                            Needs il in 1..5 and (K3 << il) + 1 <= 1024 (the counting sort's bins): the workspace
 *                            helper returns 0 and the entry point AGB_EINVAL otherwise (e.g. a 7^3 map with 4-row blocks) —
 *                            such calls run agb_spconv_fwd_opt with its fixed interleave.
 *   agb_spconv_fwd_tiles     agb_spconv_fwd_opt (no class partition) with that table; the geometry is checked.  When the
 *                            shape is one the pair-compacted kernel does not take (agb_spconv_cmp_geometry out[0] == 0)
 *                            the call runs the kernel agb_spconv_fwd_opt would and the table is not read. */
int agb_spconv_cmp_geometry(int n_out, int Cin, int Cout, int ldx, int ldy, int ksplit, int cmp_mode,
                            int cmp_interleave_shift, int32_t* out);
size_t agb_spconv_balance_tiles_workspace_bytes(int n_out, int K3, int il);
int agb_spconv_balance_tiles(const int32_t* nbr, long long nbr_stride, int n_out, int K3, int il, int ntiles, int bpt,
                             int32_t* tile_blocks, void* workspace, void* stream);
int agb_spconv_fwd_tiles(const float* X, int ldx, const float* W, const int32_t* nbr, long long nbr_stride, int kflip,
                         const float* bias, float* Y, int ldy, int n_out, int K3, int Cin, int Cout, int ksplit,
                         float* partial, int cmp_mode, int cmp_interleave_shift, const int32_t* tile_blocks, int tb_tiles,
                         int tb_blocks, void* stream);
/* Stride-1 odd-kernel convolution of a 3-channel input (the 7^3 stem; X rows 4 floats wide, W [K^3*3, Cout]) whose
 * neighbours are probed in the level's dense lookup grid (agb_grid_insert; halo >= K/2) instead of a pre-built [K^3][n]
 * kernel map.  nbr_out (optional) receives that map as a by-product — the values agb_grid_kernel_map would write — for
 * agb_spconv_bwd_weight.  Same ME convolution as agb_spconv_fwd (resnet.py conv1). */
int agb_spconv_fwd3_grid(const float* X, int ldx, const float* W, const int32_t* coords, const int32_t* grid,
                         const int32_t* desc, int K, const float* bias, float* Y, int ldy, int n_out, int Cout,
                         int32_t* nbr_out, long long nbr_out_stride, void* stream);
/* the same with bf16 operands (precision 1): 16 offsets x 4 padded channels per 64-deep K-chunk on
 * v_mfma_f32_32x32x16_bf16 — the absent neighbours of the dense-over-offsets product cost 1/16 of the fp32 MFMA;
 * precision 0 and 2 (split-bf16x3 measured slower than fp32 here) run the exact fp32 kernel */
int agb_spconv_fwd3_grid_lp(const float* X, int ldx, const float* W, const int32_t* coords, const int32_t* grid,
                            const int32_t* desc, int K, const float* bias, float* Y, int ldy, int n_out, int Cout,
                            int32_t* nbr_out, long long nbr_out_stride, int precision, void* stream);
/* The dense-over-offsets fp32 form of the stem (what agb_spconv_fwd3_grid ran until round 4; still taken for Cout != 64):
 * exported for A/B measurements and as the second implementation the parity tests compare with. */
int agb_spconv_fwd3_grid_dense(const float* X, int ldx, const float* W, const int32_t* coords, const int32_t* grid,
                               const int32_t* desc, int K, const float* bias, float* Y, int ldy, int n_out, int Cout,
                               int32_t* nbr_out, long long nbr_out_stride, void* stream);
/* The pair-sparse form of the same 3-channel stem (csrc/stem.hip; SENet.py:47-53): one wave per 64 output rows, the grid
 * probed per (row, offset), v_mfma_f32_4x4x1 on groups of four rows that have the offset.  Cout == 64, ldx == 4, fp32. */
int agb_stem_fwd_pairs(const float* X, int ldx, const float* W, const int32_t* coords, const int32_t* grid,
                       const int32_t* desc, int K, const float* bias, float* Y, int ldy, int n_out, int Cout,
                       int32_t* nbr_out, long long nbr_out_stride, void* stream);
/* Its weight gradient dW[k][c][o] += x[nbr(n, k)][c] dY[n][o] with the neighbours probed in the same grid — the stem's
 * 343 x N kernel map (578 MB at B = 32) is neither written nor read: one v_mfma_f32_4x4x1 per pair, dY staged in LDS per
 * 256-row chunk, row partitions folded in a fixed order through the caller's workspace (bitwise reproducible).
 * dW fp32 [K^3][4][Cout] (channel 3 of X is zero padding). */
size_t agb_stem_bwd_weight_grid_workspace_bytes(int n_out, int K);
int agb_stem_bwd_weight_grid(const float* X, int ldx, const float* dY, int ldy, const int32_t* coords, const int32_t* grid,
                             const int32_t* desc, int K, float* dW, int n_out, int Cout, void* workspace,
                             size_t workspace_bytes, void* stream);
/* The data gradient under its own name (SURVEY.md section 8(b) agb_spconv_bwd_data; ME's ConvolutionBackward behind
 * resnet_block.py:62-73 / senet_block.py:80-96):   dX[q] = [addend[q] +] sum_k dY[map[k][q]] @ Wt[k]
 * map: the TRANSPOSED kernel map (kflip 0; strided layers with the class partition perm / tile_cls / cls_tab / n_tiles of
 * agb_parity_partition, else NULL / 0) or the forward map of a stride-1 odd kernel read backwards (kflip 1).
 * Wt [K3][Cout][Cin] from agb_spconv_weight_transpose; Cin = channels of dX, Cout = channels of dY.  ksplit / partial as
 * agb_spconv_fwd_ex.  addend (optional, float [n_in][ld_add], may alias dX): the other gradient of a residual join, added in
 * the kernel's final store (after the sum: the value a separate addition would give) instead of by one more pass. */
int agb_spconv_bwd_data(const float* dY, int lddy, const float* Wt, const int32_t* map, long long map_stride, int kflip,
                        float* dX, int lddx, int n_in, int K3, int Cin, int Cout, const int32_t* perm,
                        const int32_t* tile_cls, const int32_t* cls_tab, int n_tiles, int ksplit, float* partial,
                        const float* addend, int ld_add, void* stream);
/* WT [K3][C][R] = per-offset transpose of W [K3][R][C] (R, C multiples of 4): the operand of the data gradient
 * dX = sum_k dY[nbrT[k]] @ W[k]^T, rebuilt once per layer per step (ME does the same inside its backward GEMMs with
 * a transposed-operand flag: MinkowskiEngine/src/convolution_kernel.cu ConvolutionBackwardKernelGPU). */
int agb_spconv_weight_transpose(const float* W, float* WT, int K3, int R, int C, void* stream);
/* every layer of a model in ONE launch: tab (DEVICE) int64 [n][6] = (W, WT, K3, R, C, first tile), first tile = running sum of
 * K3 * ceil(R / 64) * ceil(C / 64) over the layers before; total_tiles = the sum over all layers.  The caller keeps the
 * transposes until the weights change (one launch per optimiser step instead of one per layer and backward pass). */
int agb_spconv_weight_transpose_batched(const long long* tab, int n, long long total_tiles, void* stream);
/* the same, and `zero` (K3*R*C floats, or NULL) is cleared in the same pass: the weight-gradient buffer
 * agb_spconv_bwd_weight accumulates into, saving one fill launch per layer */
int agb_spconv_weight_transpose_z(const float* W, float* WT, float* zero, int K3, int R, int C, void* stream);
int agb_spconv_cmp_occupancy(int rows_per_wave);  /* resident workgroups per CU of the pair-compacted kernel (tuning aid) */
/* Low-precision MFMA operands, fp32 accumulate and I/O: precision 1 = bf16 (BASELINE config 5), 2 = split-bf16 x3
 * (a_hi*b_hi + a_hi*b_lo + a_lo*b_hi: fp32-level accuracy at 3/16 of the fp32 MFMA cost).  Same contract as
 * agb_spconv_fwd_ex but the weights are K-major: Wt float[K3][Cout][Cin] (forward: the transposed kernel; data
 * gradient: the kernel itself).  Cin >= 12. */
int agb_spconv_fwd_lp(const float* X, int ldx, const float* Wt, const int32_t* nbr, long long nbr_stride, int kflip,
                      const float* bias, float* Y, int ldy, int n_out, int K3, int Cin, int Cout,
                      const int32_t* perm, const int32_t* tile_cls, const int32_t* cls_tab, int n_tiles, int ksplit,
                      float* partial, int precision, void* stream);
/* The bf16 mode on bf16 STORAGE (BASELINE config 5: "bf16 with fp32 index kernels"): agb_to_bf16 makes the bf16 twin of a
 * row matrix (round to nearest even; C, ldx, ldy multiples of 4) — of an activation, of a gradient, of the K-major weights
 * — and agb_spconv_fwd_b16 is agb_spconv_fwd_lp(precision = 1) reading those twins: X16 uint16 [n_in][ldx16], Wt16 uint16
 * [K3][Cout][Cin]; 16-byte pieces of 8 channels go from global memory to LDS unconverted (half the bytes through the CU's
 * vector-memory path, no conversion instructions); fp32 accumulate, bias and output.  Cin, ldx16 multiples of 8.
 * Same ME convolution as agb_spconv_fwd (SENet.py:185, senet_block.py:99-147 under torch.cuda.amp in the reference). */
int agb_to_bf16(const float* X, long long ldx, long long n, int C, uint16_t* Y16, long long ldy, void* stream);
int agb_spconv_fwd_b16(const uint16_t* X16, int ldx16, const uint16_t* Wt16, const int32_t* nbr, long long nbr_stride,
                       int kflip, const float* bias, float* Y, int ldy, int n_out, int K3, int Cin, int Cout,
                       const int32_t* perm, const int32_t* tile_cls, const int32_t* cls_tab, int n_tiles, int ksplit,
                       float* partial, void* stream);
/* bf16 ROW STORAGE (KernelOptions.bf16_activations; BASELINE config 5): every activation / gradient row matrix [N, C] of
 * the sparse backbone is uint16 bf16 (round to nearest even when a row leaves a kernel; accumulators, statistics, per-plot
 * [B, C] matrices, parameters and their gradients stay fp32).  agb_spconv_fwd_h = agb_spconv_fwd_b16 with bf16 output rows
 * Y16 uint16 [n_out][ldy16] (ldy16 % 4 == 0); agb_spconv_fwd3_grid_h = the grid-probing 3-channel stem with bf16 operands
 * writing bf16 rows.  The weight gradient of this mode is agb_spconv_bwd_weight_b16 on the rows themselves.
 * Every entry point below that takes row matrices has a twin with the suffix _h and the same argument list where
 * `float*` row-matrix arguments are `uint16_t*` (leading dimensions in elements): same arithmetic in fp32, 8-byte instead
 * of 16-byte pieces (csrc/norm_rows.inc, csrc/pool_rows.inc). */
int agb_spconv_fwd_h(const uint16_t* X16, int ldx16, const uint16_t* Wt16, const int32_t* nbr, long long nbr_stride,
                     int kflip, const float* bias, uint16_t* Y16, int ldy16, int n_out, int K3, int Cin, int Cout,
                     const int32_t* perm, const int32_t* tile_cls, const int32_t* cls_tab, int n_tiles, int ksplit,
                     float* partial, void* stream);
/* The data gradient on bf16 rows under its own name (as agb_spconv_bwd_data is for fp32 rows):
 *   dX16[q] = bf16( [addend16[q] +] sum_k dY16[map[k][q]] W16[k] )     fp32 sum, ONE rounding.
 * W16 uint16 [K3][Cin][Cout] (Cin = channels of dX, Cout = channels of dY: the layer's kernel itself is the K-major operand
 * of its data gradient); map / kflip / perm / ksplit as agb_spconv_fwd_h.  addend16 (optional; bf16 [n_in][ld_add], ld_add
 * even, ksplit == 1): the other gradient of a residual join — a block input that feeds this layer and the shortcut
 * (resnet_block.py:93-133, senet_block.py:99-147) — added before the rounding instead of by a separate pass. */
int agb_spconv_bwd_data_h(const uint16_t* dY16, int lddy16, const uint16_t* W16, const int32_t* map, long long map_stride,
                          int kflip, uint16_t* dX16, int lddx16, int n_in, int K3, int Cin, int Cout, const int32_t* perm,
                          const int32_t* tile_cls, const int32_t* cls_tab, int n_tiles, int ksplit, float* partial,
                          const uint16_t* addend16, int ld_add, void* stream);
/* Both bf16 operand forms of one layer's weights W float[K3][R][C] in one launch: W16 uint16 [K3][R][C] (the data
 * gradient's K-major form) and Wt16 uint16 [K3][C][R] (the forward pass's), round to nearest even. */
int agb_weight_twins_bf16(const float* W, int K3, int R, int C, uint16_t* W16, uint16_t* Wt16, void* stream);
/* both bf16 forms of EVERY layer in one launch: tab (DEVICE) int64 [n][7] = (W, W16, Wt16, K3, R, C, first tile), first tile =
 * running sum of K3 * ceil(R / 64) * ceil(C / 64); kept by the caller until the weights change */
int agb_weight_twins_batched(const long long* tab, int n, long long total_tiles, void* stream);
int agb_spconv_fwd3_grid_h(const float* X, int ldx, const float* W, const int32_t* coords, const int32_t* grid,
                           const int32_t* desc, int K, const float* bias, uint16_t* Y16, int ldy16, int n_out, int Cout,
                           int32_t* nbr_out, long long nbr_out_stride, void* stream);
/* perm int32[n + stride^3*64] (rows grouped by class, -1 padding), tile_cls int32[max_tiles] with
 * max_tiles = n/64 + stride^3 + 1, scratch int32[256].  class = (c/ts_in mod stride) per axis, x fastest. */
int agb_parity_partition(const int32_t* coords, int n, int ts_in, int stride, int32_t* perm, int32_t* tile_cls,
                         int max_tiles, int32_t* scratch, void* stream);

/* dW[k] += sum_r X[nbr[k][r],:]^T @ dY[r,:]; dW ([K3*Cin, Cout]) must be zero-filled by the caller. */
int agb_spconv_bwd_weight(const float* X, int ldx, const float* dY, int ldy, const int32_t* nbr,
                          long long nbr_stride, float* dW, int n_out, int K3, int Cin, int Cout, void* stream);
/* The same with the operand precision as an argument: 0 = fp32 MFMA, 1 = bf16 operands, 2 = split-bf16 x3 (fp32
 * accumulate; gathered rows are transposed and packed to bf16 pairs while they are staged; Cin = 4 / 8 stays fp32).
 * nbr == NULL (K3 == 1): identity map, dW = X^T dY (1x1 stride-1 convolutions and nn.Linear weight gradients). */
int agb_spconv_bwd_weight_lp(const float* X, int ldx, const float* dY, int ldy, const int32_t* nbr,
                             long long nbr_stride, float* dW, int n_out, int K3, int Cin, int Cout, int precision,
                             void* stream);
/* The same with a caller-owned workspace = the REPRODUCIBLE form: the fp32 kernels then leave one partial tile per row chunk
 * and add the chunks in ascending order (bitwise reproducible, two-level sums) instead of accumulating with fp32 atomics —
 * Cin >= 12: csrc/dwreg.hip (both MFMA operands gathered straight into registers, four waves per (row chunk, offset, 64 x 64
 * tile) unit, no barrier while they multiply); Cin = 4 / 8 (the 7^3 stem): groups of four sub-chunks per workgroup.
 * workspace == NULL (= agb_spconv_bwd_weight_lp): the LDS-staged kernels with atomic accumulation, the faster form inside the
 * training step.  Since round 4 the workspace form covers EVERY operand precision and shape: bf16 / bf16x3 operands (the
 * LDS-staged kernel writes one partial tile per row chunk, folded in ascending order) and the HBM-bound dense shapes (which
 * take the register-operand kernel instead of the streaming kernel's cross-workgroup atomics).
 * agb_spconv_bwd_weight_workspace_bytes (host helper; dense = 1 for nbr == NULL) returns 0 only for empty products.
 * variant: 0 automatic, 1 = LDS-staged kernel, 2 = register-operand kernel whatever the workspace, 3 = the persistent-accumulator
 * kernel for every shape it can take, 4 = LDS-staged in its round-2..4 geometry (2048-row chunks, 64-pair steps; since round 5
 * the fp32 form walks 1280-row chunks in 32-pair steps: 23.5 KB of LDS, six workgroups per CU) (A/B measurements, tests). */
size_t agb_spconv_bwd_weight_workspace_bytes(int n_out, int K3, int Cin, int Cout, int dense, int precision);
/* fp32 maps (nbr != NULL) with Cin and Cout multiples of 64, 2 <= K3 <= 28 and >= 2048 rows can run on the
 * PERSISTENT-ACCUMULATOR kernel (csrc/dwa.hip, hand-scheduled: csrc/gen_dw_asm.py): one wave keeps the 64 x 64 tiles of up to
 * four offsets in its AGPRs for the whole launch, partial tiles leave once per wave, k_dwa_fold adds them in ascending order
 * (reproducible).  With a workspace, variant 0 takes it for the shapes it was measured faster on (agb_spconv_bwd_weight_persistent:
 * K3 <= 8, or >= 40000 rows with Cin >= 128), variant 3 for every shape it can take.  Inside the training step it is equal to
 * the staged kernel: the Python side passes the workspace only on request (AGB_PERSISTENT_WGRAD, KernelOptions.dw_variant = 3,
 * or the reproducible mode). */
/* ROW BOUND of the persistent kernel: it addresses X and dY rows with 32-bit byte offsets and this ABI carries no n_in, so
 * every row index of the map times ldx * 4 (ldy * 4) must stay below 2^32 (true for every stride <= 2 map of a level of
 * n_out rows the check accepts; a caller-built map that reaches further must pass variant 1 or 2). */
int agb_spconv_bwd_weight_persistent(int n_out, int K3, int Cin, int Cout, int ldx, int ldy);
int agb_spconv_bwd_weight_ws(const float* X, int ldx, const float* dY, int ldy, const int32_t* nbr,
                             long long nbr_stride, float* dW, int n_out, int K3, int Cin, int Cout, int precision,
                             int variant, void* workspace, size_t workspace_bytes, void* stream);
/* bf16 operands from bf16 STORAGE (the twins of agb_to_bf16; see agb_spconv_fwd_b16): the weight gradient of the bf16 mode
 * without the fp32 gathers and conversions; dW fp32, accumulated into.  Cin >= 12; ldx16, ldy16 multiples of 4. */
int agb_spconv_bwd_weight_b16(const uint16_t* X16, int ldx16, const uint16_t* dY16, int ldy16, const int32_t* nbr,
                              long long nbr_stride, float* dW, int n_out, int K3, int Cin, int Cout, void* stream);
/* The same with a caller-owned workspace of agb_spconv_bwd_weight_workspace_bytes(n_out, K3, Cin, Cout, nbr == NULL, 1)
 * bytes: the bf16 weight gradient as a fixed-order two-level sum — bitwise reproducible training in the bf16 modes
 * (the reference's eval.py:12-21 seeds everything to be repeatable; ME's atomics are not).  workspace == NULL: as above. */
int agb_spconv_bwd_weight_b16_ws(const uint16_t* X16, int ldx16, const uint16_t* dY16, int ldy16, const int32_t* nbr,
                                 long long nbr_stride, float* dW, int n_out, int K3, int Cin, int Cout, void* workspace,
                                 size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Pooling / broadcast (replaces ME.MinkowskiMaxPooling SENet.py:53; ME.MinkowskiGlobal{Sum,Avg,Max}Pooling
 * SENet.py:63, senet_block.py:43, PointNet.py:29; ME.MinkowskiBroadcastMultiplication senet_block.py:44-50)
 * --------------------------------------------------------------------------------------------------------- */
int agb_maxpool_fwd(const float* X, int ldx, const int32_t* nbr, long long nbr_stride, float* Y, int ldy,
                    int32_t* argmax /* [n_out, C] */, int n_out, int K3, int C, void* stream);
int agb_maxpool_bwd(const float* dY, int ldy, const int32_t* argmax, const int32_t* nbrT, long long nbrT_stride,
                    float* dX, int ldx, int n_in, int K3, int C, void* stream);
/* the same pair with the winner stored as its offset index (uint8 argk [n_out, C], 255 = no neighbour; K3 <= 255): the
 * gradient pass is bound by its gathers of (argmax, dY) rows and reads a quarter of the argmax bytes.  nbrT must be the
 * transpose of the forward map (same offset numbering: nbrT[k][q] = o <=> nbr[k][o] = q). */
int agb_maxpool_fwd_k(const float* X, int ldx, const int32_t* nbr, long long nbr_stride, float* Y, int ldy,
                      uint8_t* argk, int n_out, int K3, int C, void* stream);
int agb_maxpool_bwd_k(const float* dY, int ldy, const uint8_t* argk, const int32_t* nbrT, long long nbrT_stride,
                      float* dX, int ldx, int n_in, int K3, int C, void* stream);
/* Y[b,:] = reduce over rows ptr[b]..ptr[b+1] of A (optionally A*Bm); mode 0 sum, 1 average, 2 max (+argmax rows).
 * splits > 1 cuts every segment into that many row chunks (scratch: part float[B*splits*C], part_arg
 * int32[B*splits*C] for max) folded in a fixed order: deterministic, no float atomics. */
int agb_segment_reduce(const float* A, int lda, const float* Bm, int ldb, const int32_t* ptr, int B, int C, int mode,
                       int splits, float* part, int32_t* part_arg, float* Y, int32_t* argmax, void* stream);
/* out[r,:] = S[batch(r),:] (/ rows of the batch if average) (* M[r,:] if M) */
int agb_segment_broadcast(const float* S, const int32_t* coords, const int32_t* ptr, const float* M, int ldm,
                          float* out, int ldo, int n, int C, int average, void* stream);
/* out[r,:] = M[r,:] * S[batch(r),:] + T[batch(r),:] / rows(batch(r)): input gradient of the squeeze-excite layer
 * (SELayer, senet_block.py:33-50: global average pooling -> fc -> broadcast multiplication) in one pass. */
int agb_segment_scale_add(const float* S, const float* T, const int32_t* coords, const int32_t* ptr, const float* M,
                          int ldm, float* out, int ldo, int n, int C, void* stream);
/* dX[argmax[b,c], c] = dY[b,c]; dX zero-filled by the caller */
int agb_segment_max_bwd(const float* dY, const int32_t* argmax, float* dX, int ldx, int B, int C, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * BatchNorm over [n, C] rows fused with the following activation (0 none, 1 ReLU, 2 GELU-erf), and the residual
 * tail.  Replaces ME.MinkowskiBatchNorm (= nn.BatchNorm1d on .F) + activation: modules/MinkowskiEngine/common.py:
 * 215-226, resnet_block.py:62-73, senet_block.py:83-96, PointNet.py:16-39; KPConv blocks.py:460-535.
 * --------------------------------------------------------------------------------------------------------- */
int agb_bn_chunks(int n);  /* host helper: number of row chunks the statistics kernels use */
/* training != 0: batch statistics (Chan-combined), running stats updated when given; else running stats.
 * part: float[agb_bn_chunks(n)*3*C] scratch; mean, rstd: float[C] out. */
int agb_bn_stats(const float* X, int ldx, int n, int C, float eps, float momentum, int training, float* part,
                 float* mean, float* rstd, float* running_mean, float* running_var, void* stream);
/* the same, and in training mode the layer's int64 num_batches_tracked counter (device pointer or NULL) is incremented
 * by the fold kernel (nn.BatchNorm1d bookkeeping, torch/nn/modules/batchnorm.py) */
int agb_bn_stats_tracked(const float* X, int ldx, int n, int C, float eps, float momentum, int training, float* part,
                         float* mean, float* rstd, float* running_mean, float* running_var,
                         long long* num_batches_tracked, void* stream);
/* the fold half of agb_bn_stats_tracked on partials produced by agb_dense_fwd_bn */
int agb_bn_stats_fold(const float* part, int chunks, int C, float eps, float momentum, float* mean, float* rstd,
                      float* running_mean, float* running_var, long long* num_batches_tracked, void* stream);
int agb_bn_act_fwd(const float* X, int ldx, int n, int C, const float* mean, const float* rstd, const float* gamma,
                   const float* beta, int act, float* Y, int ldy, void* stream);
/* part: float[agb_bn_chunks(n)*2*C] scratch; dgamma, dbeta: float[C] out; dX may be NULL */
int agb_bn_act_bwd(const float* X, int ldx, const float* dY, int ldy, int n, int C, const float* mean,
                   const float* rstd, const float* gamma, const float* beta, int act, int training, float* part,
                   float* dX, int lddx, float* dgamma, float* dbeta, void* stream);
/* Same, plus colsum float[C] (optional): the column sums of dX — the bias gradient of the convolution feeding the
 * BatchNorm (common.py:215-226 ConvNormActivation, resnet_block.py:62-69) — in closed form from the folded sums:
 * exactly 0 with batch statistics (BatchNorm is blind to a constant added to its input; the reference's optimiser sees
 * fp32 rounding noise around that zero), gamma * rstd * dbeta with running statistics. */
int agb_bn_act_bwd_colsum(const float* X, int ldx, const float* dY, int ldy, int n, int C, const float* mean,
                          const float* rstd, const float* gamma, const float* beta, int act, int training, float* part,
                          float* dX, int lddx, float* dgamma, float* dbeta, float* colsum, void* stream);
/* The tail of a squeeze-excite residual block in four passes forward and eight backward instead of 9 + 14
 * (senet_block.py:83-96,126-147; resnet_block.py:70-73):  y = act(BatchNorm(z) * s[plot] * keep[plot] + r) with
 * s = excitation MLP of the plot means of BatchNorm(z).  coords int32[n][4] (batch index first), ptr int32[B+1].
 * Reductions run over plot-aligned row chunks and are folded in a fixed order: no atomics, bitwise reproducible.
 * Call order: _stats, _pool, agb_se_mlp_fwd, _fwd | _bwd_sums, _bwd_ds, agb_se_mlp_bwd, _bwd_fold, _bwd_apply. */
/* Without an excitation (s == NULL, keep == NULL, coords == NULL, ptr == NULL) the same entry points are the tail of a
 * plain residual block, y = act(BatchNorm(z) + r) (KPConv blocks.py:640-668): _fwd | _bwd_sums, agb_bn_bwd_fold, _bwd_apply. */
int agb_bn_bwd_fold(const float* part, int chunks, int C, float* dbeta, float* dgamma, void* stream);
int agb_se_tail_chunks(int n, int C, int B);   /* plot-aligned row chunks: sizes part (x 3 C floats) and spart (x 2 C) */
int agb_se_tail_stats(const float* Z, int ldz, const int32_t* ptr, int n, int C, int B, float eps, float momentum,
                      int training, float* part, float* mean, float* rstd, float* running_mean, float* running_var,
                      long long* num_batches_tracked, void* stream);
int agb_se_tail_pool(const float* part, const int32_t* ptr, int n, int B, int C, const float* mean, const float* rstd,
                     const float* gamma, const float* beta, float* zbar, float* pooled, void* stream);
int agb_se_tail_fwd(const float* Z, int ldz, const float* R, int ldr, const int32_t* coords, const float* mean,
                    const float* rstd, const float* gamma, const float* beta, const float* s, const float* keep, int act,
                    int n, int C, float* Y, int ldy, void* stream);
int agb_se_tail_bwd_sums(const float* Z, int ldz, const float* R, int ldr, const float* dY, int ldy, const int32_t* ptr,
                         int B, const float* mean, const float* rstd, const float* gamma, const float* beta,
                         const float* s, const float* keep, int act, int n, int C, float* spart, void* stream);
int agb_se_tail_bwd_ds(const float* spart, const int32_t* ptr, int n, const float* gamma, const float* beta,
                       const float* keep, int B, int C, float* S2, float* S3, float* ds, void* stream);
int agb_se_tail_bwd_fold(const float* S2, const float* S3, const float* zbar, const int32_t* ptr, const float* dp,
                         const float* s, const float* keep, const float* mean, const float* rstd, int B, int C, float* dte,
                         float* dbeta, float* dgamma, void* stream);
int agb_se_tail_bwd_apply(const float* Z, int ldz, const float* R, int ldr, const float* dY, int ldy, const int32_t* coords,
                          const float* mean, const float* rstd, const float* gamma, const float* beta, const float* s,
                          const float* keep, const float* dte, const float* dbeta, const float* dgamma, int act,
                          int training, int n, int C, float* dZ, int lddz, float* dR, int lddr, void* stream);
/* Y = act(A * scale[batch(row)] + R); scale float[B] (drop-path keep/(1-p)) and coords may be NULL */
int agb_add_act_fwd(const float* A, int lda, const float* R, int ldr, const float* scale, const int32_t* coords, int n,
                    int C, int act, float* Y, int ldy, void* stream);
int agb_add_act_bwd(const float* A, int lda, const float* R, int ldr, const float* scale, const int32_t* coords,
                    const float* dY, int ldy, int n, int C, int act, float* dA, float* dR, void* stream);
/* Channel-wise LayerNorm of every row (norm_type="ln": SENet.py:40-41 -> common.py:369-386 MinkowskiLayerNorm =
 * nn.LayerNorm(C, eps=1e-6) on .F).  stats float[n][2] = (mean, rstd) per row, written by _fwd and read by _bwd.
 * _bwd: dX may be NULL; part float[agb_layernorm_chunks(n)][2][C] scratch for the fixed-order column sums behind
 * dgamma / dbeta (either may be NULL).  C <= 2048. */
int agb_layernorm_chunks(int n);       /* host helper */
int agb_layernorm_fwd(const float* X, int ldx, int n, int C, const float* gamma, const float* beta, float eps, float* Y,
                      int ldy, float* stats, void* stream);
int agb_layernorm_bwd(const float* X, int ldx, const float* dY, int ldy, int n, int C, const float* gamma,
                      const float* stats, float* dX, int lddx, float* part, float* dgamma, float* dbeta, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * KPConv index path.  Replaces the CPython extensions behind modules/KPConv/common.py:
 *   radius_neighbors.batch_query   cpp_wrappers/cpp_neighbors/wrapper.cpp:58-238 -> neighbors/neighbors.cpp:211-333
 *   grid_subsampling.subsample_batch  cpp_wrappers/cpp_subsampling/wrapper.cpp:62-335 -> grid_subsampling.cpp:109-211
 * Clouds are stacked; ptr int32[B+1] gives the row range of every cloud, elem int32[n] the cloud of every row.
 * --------------------------------------------------------------------------------------------------------- */
int agb_elem_bbox(const float* pts, const int32_t* ptr, int B, int n, int32_t* bbox_ord /* int32[6B] scratch */,
                  float* bbox /* float[6B] out: min xyz, max xyz; may be NULL */, void* stream);
int agb_elem_of_row(const int32_t* ptr, int B, int n, int32_t* elem, void* stream);
/* out = pts @ R[elem] (transpose=0) or pts @ R[elem]^T (1), float32 products summed left to right (common.py:76-97) */
int agb_rotate_points(const float* pts, const int32_t* elem, const float* R /* [B,3,3] */, int n, int transpose,
                      float* out, void* stream);
/* Support cell grid: origin_cs = HOST float[4] {ox, oy, oz, cell size >= radius}; dims = HOST int32[4] {X, Y, Z, B}.
 * cell_start int32[cells+1] out; sorted float[ns*4] out (xyz + index bits, cell order); cell_of int32[ns],
 * cell_fill int32[cells+1], scan_scratch int32[agb_scan_scratch_elems(cells+1)], total_scratch int32[1]: scratch. */
int agb_ball_grid_build(const float* supports, int ns, const int32_t* s_ptr, const float* origin_cs,
                        const int32_t* dims, int32_t* cell_start, float* sorted, int32_t* cell_of,
                        int32_t* cell_fill, int32_t* scan_scratch, int32_t* total_scratch, void* stream);
/* Phase 1: counts[nq] and *max_count (device). Phase 2: out int32[nq,width] sorted by (d2, index), padded with ns. */
int agb_ball_query_count(const float* queries, int nq, const int32_t* q_elem, const float* origin_cs,
                         const int32_t* dims, const int32_t* cell_start, const float* sorted, float radius,
                         int32_t* counts, int32_t* max_count, void* stream);
int agb_ball_query_fill(const float* queries, int nq, const int32_t* q_elem, const float* origin_cs,
                        const int32_t* dims, const int32_t* cell_start, const float* sorted, float radius, int ns,
                        int width, int32_t* out, int32_t* status, void* stream);
/* RAGGED (CSR) radius search — SURVEY.md §8(d): the ball query then writes sum(counts) * 4 bytes instead of the padded
 * nq * max_count * 4 (16 k-point plots: 20 valid of 265 columns; the reference pads to the batch-wide maximum,
 * cpp_neighbors/neighbors.cpp:319-325, and gathers the padded matrix, modules/KPConv/blocks.py:304-310,383-386).
 * Order: agb_ball_query_count -> agb_ball_query_offsets (row_ptr int32[nq + 1] = exclusive scan of counts, row_ptr[nq] =
 * total; scratch int32[agb_scan_scratch_elems(nq)]) -> read row_ptr[nq] back -> agb_ball_query_fill_csr (indices
 * int32[capacity], capacity = that total; every row sorted by (d2, index) exactly like the padded rows).  agb_csr_to_padded rebuilds the reference's
 * matrix (pad = ns) for callers of batch_neighbors. */
int agb_ball_query_offsets(const int32_t* counts, int nq, int32_t* row_ptr, int32_t* scratch, void* stream);
int agb_ball_query_fill_csr(const float* queries, int nq, const int32_t* q_elem, const float* origin_cs,
                            const int32_t* dims, const int32_t* cell_start, const float* sorted, float radius, int ns,
                            const int32_t* row_ptr, int32_t* indices, int capacity, int32_t* status, void* stream);
/* The same with the longest list of the search stated (max_count: the value agb_ball_query_count left in *max_count, which the
 * caller reads back together with row_ptr[nq]): the kernel sizes its LDS slab from it (next power of two >= max(max_count,
 * 256) keys per wave instead of 1024) and two to four times the waves fit a CU.  Same rows. */
int agb_ball_query_fill_csr_m(const float* queries, int nq, const int32_t* q_elem, const float* origin_cs, const int32_t* dims,
                              const int32_t* cell_start, const float* sorted, float radius, int ns, const int32_t* row_ptr,
                              int32_t* indices, int capacity, int max_count, int32_t* status, void* stream);
int agb_csr_to_padded(const int32_t* row_ptr, const int32_t* indices, int nq, int width, int pad, int32_t* out,
                      void* stream);
/* Grid subsampling (barycentres, optional feature means), canonical order = cell key ascending per cloud.
 * cap = cells reserved per cloud (>= the cloud's NX*NY*NZ; status[0] counts violations).  workspace: one device buffer of
 * agb_grid_subsample_workspace_bytes(n, B, cap) bytes (all internal scratch is carved from it).
 * Out: out_pts float[n*3] (upper bound), out_feats float[n*fdim] or NULL, out_ptr int32[B+1], n_out_dev, status[4]. */
size_t agb_grid_subsample_workspace_bytes(int n, int B, int cap);
int agb_grid_subsample_ws(const float* pts, const float* feats, int fdim, int n, const int32_t* ptr, const int32_t* elem,
                          int B, float dl, int cap, void* workspace, float* out_pts, float* out_feats, int32_t* out_ptr,
                          int32_t* n_out_dev, int32_t* status, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * KPConv layer (replaces KPConv.forward modules/KPConv/blocks.py:264-400 and max_pool :98-114).
 * idx int32[N,H], entries >= Ns are shadow neighbours.  wf[n,k,:] = sum_h infl(n,h,k) * x[idx[n,h],:],
 * infl = max(0, 1 - |(s[idx]-q) - kp[k]| / extent); the layer output is wf[N,K*Cin] @ W[K*Cin,Cout] (plain GEMM).
 * --------------------------------------------------------------------------------------------------------- */
int agb_kpconv_gather_fwd(const float* q, const float* s, const int32_t* idx, int H, int Ns, const float* x, int ldx,
                          const float* kp, int K, float extent, float* wf, int N, int Cin, void* stream);
/* dx float[Ns,ldx] zero-filled by the caller (fp32 atomics) */
int agb_kpconv_gather_bwd(const float* q, const float* s, const int32_t* idx, int H, int Ns, const float* dwf,
                          const float* kp, int K, float extent, float* dx, int ldx, int N, int Cin, void* stream);
int agb_kp_maxpool_fwd(const float* x, int ldx, const int32_t* idx, int H, int Ns, float* y, int32_t* argmax, int N,
                       int C, void* stream);
int agb_kp_maxpool_bwd(const float* dy, const int32_t* argmax, float* dx, int ldx, int N, int C, void* stream);
/* The gather passes and the max-pooled shortcut on RAGGED rows (row_ptr / indices of agb_ball_query_fill_csr): row n =
 * indices[row_ptr[n] .. row_ptr[n + 1]) cut at `limit` entries (the reference's neighborhood_limits crop of the padded matrix;
 * INT_MAX: none).  max_count_dev (device int32, from agb_ball_query_count): the width the padded matrix would have — a row
 * shorter than min(limit, *max_count_dev) has shadow neighbours, whose zero feature row takes part in the max
 * (blocks.py:98-114).  Same results as the padded entry points. */
int agb_kpconv_gather_fwd_csr(const float* q, const float* s, const int32_t* row_ptr, const int32_t* indices, int limit,
                              int Ns, const float* x, int ldx, const float* kp, int K, float extent, float* wf, int N,
                              int Cin, void* stream);
int agb_kpconv_gather_bwd_csr(const float* q, const float* s, const int32_t* row_ptr, const int32_t* indices, int limit,
                              int Ns, const float* dwf, const float* kp, int K, float extent, float* dx, int ldx, int N,
                              int Cin, void* stream);
int agb_kp_maxpool_fwd_csr(const float* x, int ldx, const int32_t* row_ptr, const int32_t* indices, int limit,
                           const int32_t* max_count_dev, int Ns, float* y, int32_t* argmax, int N, int C, void* stream);

/* The whole rigid KPConv layer as ONE kernel per direction (csrc/kpfused.hip; replaces the expression
 * modules/KPConv/blocks.py:304-400: neighbours gathered, influences, matmul to [N, K, Cin], matmul with the kernel weights, sum
 * over K): a layer whose query and support sets are the SAME points `pts` [N][3] with ragged neighbour rows (row_ptr / indices of
 * agb_ball_query_fill_csr, rows cut at `limit` entries).  The weighted neighbourhood features wf[N, K, Cin] exist only as tiles
 * in LDS; fixed summation order in both directions (no atomics, nothing to zero-fill).
 *   agb_kpconv_fused_supported: 1 where the kernels cover the layer (K <= 16 kernel points, Cin == Cout in {16, 32}); N < 2^24.
 *   fwd: out [N][ldo] from x [N][ldx], kp [K][3], W [K][Cin][Cout].
 *   bwd: needs a SYMMETRIC neighbour relation (j in row n <=> n in row j: an uncropped radius search of a point set against
 *        itself).  dx [N][lddx] (NULL: not wanted) and dW [K][Cin][Cout] (NULL: not wanted; accumulate != 0: added to dW) from dy
 *        [N][lddy] and the layer input x [N][ldx]; workspace of agb_kpconv_fused_bwd_workspace_bytes bytes (used for dW only). */
int agb_kpconv_fused_supported(int K, int Cin, int Cout);
size_t agb_kpconv_fused_bwd_workspace_bytes(int N, int K, int Cin, int Cout);
int agb_kpconv_fused_fwd(const float* pts, const int32_t* row_ptr, const int32_t* indices, int limit, int N, const float* x,
                         int ldx, const float* kp, int K, float extent, const float* W, float* out, int ldo, int Cin, int Cout,
                         void* stream);
int agb_kpconv_fused_bwd(const float* pts, const int32_t* row_ptr, const int32_t* indices, int limit, int N, const float* dy,
                         int lddy, const float* kp, int K, float extent, const float* W, const float* x, int ldx, float* dx,
                         int lddx, float* dW, int accumulate, void* workspace, size_t workspace_bytes, int Cin, int Cout,
                         void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * GridSampling3D(size, quantize_coords=True, mode="last") for a batch of clouds
 * (replaces core/data_transform/grid_transform.py:112-128).  perm int64[n]: within-cloud shuffle.
 * Out: coords int32[n,3] (upper bound), keep int64[n] (rows of the ORIGINAL stacked order), out_ptr int32[B+1],
 * n_out_dev, bounds int32[6] (min/max of coords), status[4].  cap = cells reserved per cloud; workspace: one device
 * buffer of agb_voxelize_last_workspace_bytes(n, B, cap) bytes.
 * --------------------------------------------------------------------------------------------------------- */
size_t agb_voxelize_last_workspace_bytes(int n, int B, int cap);
int agb_voxelize_last_ws(const float* pos, const long long* perm, const int32_t* ptr, const int32_t* elem, int B, int n,
                         float size, int cap, void* workspace, int32_t* coords, long long* keep, int32_t* out_ptr,
                         int32_t* n_out_dev, int32_t* bounds, int32_t* status, void* stream);
/* the same with the shuffle drawn ON THE DEVICE from `seed` (no permutation tensor): per cloud a keyed pseudo-random
 * bijection of its rows (four Feistel rounds, cycle-walked; dpcr-agb_amd/csrc/voxelize.hip vox_perm) — a voxel's
 * representative is uniformly random over its points, which is all GridSampling3D(mode="last") asks of the shuffle
 * (grid_transform.py:118-121); reproducible for a seed */
int agb_voxelize_last_seeded_ws(const float* pos, unsigned long long seed, const int32_t* ptr, const int32_t* elem, int B, int n,
                                float size, int cap, void* workspace, int32_t* coords, long long* keep, int32_t* out_ptr,
                                int32_t* n_out_dev, int32_t* bounds, int32_t* status, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Fused multi-tensor AdaBelief step with clip_grad_value_ (replaces core/optimizer/adabelief.py:89-201 +
 * models/base_model.py:241-245).  descs: device array of {float* p; const float* g; float* m; float* v; int64 n}
 * per tensor; chunk_tensor / chunk_index: int32[n_chunks] (tensor of every agb_adabelief_chunk()-element block and
 * its chunk number).  mode 0 rectified-adaptive, 1 SGD-like, 2 no update, 3 non-rectified.  clip <= 0: no clip.
 * --------------------------------------------------------------------------------------------------------- */
int agb_adabelief_chunk(void);
int agb_adabelief_step(const void* descs, const int32_t* chunk_tensor, const int32_t* chunk_index, int n_chunks,
                       float decay, float beta1, float beta2, float one_minus_beta1, float one_minus_beta2, float eps,
                       float step, float inv_sqrt_bc2, int mode, float clip, void* stream);

/* ---- NFI transform chain on the device (dpcr-agb_amd/csrc/transform.hip) -------------------------------------------
 * Replaces, for a whole batch, the per-sample CPU transforms of conf/data/instance/NFI/transforms/sparse-xy.yaml:
 * ScalePos, MoveCenterPosPerSample, StartZFromZero (core/data_transform/transforms.py:590-598,722-739,766-769),
 * Polygon2dExtend (:1461-1496, matplotlib Path.contains_points restated in double) and the feature build
 * x = [1, pos.z, ||pos.xy - c + 1e-6||] (features.py:307-334,353-383).  xform: HOST float[8] = (sx, sy, sz, cx, cy, cz,
 * fcx, fcy); poly: device double[2*nv] (nv = 0: no crop).  workspace: one device buffer of agb_plot_workspace_bytes(n, B)
 * bytes.  Out (n rows reserved): pos_out, x_out float[n,3], src int64[n] (input row of every kept point, order
 * preserved), out_ptr int32[B+1], n_out_dev int32[1]. */
size_t agb_plot_workspace_bytes(int n, int B);
int agb_plot_prepare_ws(const float* pos, const int32_t* ptr, const int32_t* elem, int B, int n, const float* xform,
                        int scale_div, int z_from_zero, const double* poly, int nv, void* workspace, float* pos_out,
                        float* x_out, long long* src, int32_t* out_ptr, int32_t* n_out_dev, void* stream);
/* RandomCoordsFlip (core/data_transform/sparse_transforms.py:49-55) + ShiftVoxels (transforms.py:1046-1054) on voxel
 * coordinates int32[n,3], in place: flip int32[B,3] (0/1), shift int32[B,3], both drawn by the host; cmax int32[B,3]
 * scratch. */
int agb_coords_augment(int32_t* coords, const int32_t* elem, int B, int n, const int32_t* flip, const int32_t* shift,
                       int32_t* cmax, void* stream);

/* Train-time float augmentations of sparse-xy.yaml:4-69, applied to a whole batch from host-drawn parameters
 * (dpcr-agb_amd/train_transforms.py draws them in the reference's per-sample order):
 * agb_plot_augment: pos1 = ((raw[sel] - (0,0,zsub)) / scale + noise) @ M^T + shift + centre — RandomGroundRemoval's
 *   z shift, ScalePos, RandomNoise, Random3AxisRotation, RandomShiftPos, MoveCenterPosPerSample
 *   (transforms.py:1140-1150,590-598,482-505,747-759,722-739; features.py:44-60); sel int64[n] = rows kept by the ground
 *   removal / RandomDropout (:1060-1087); aug: B records of 24 floats (zsub, sx,sy,sz, M[9] row-major, tx,ty,tz, cx,cy,cz,
 *   5 pad); also writes mins float[3B] = per-plot minimum of pos1.
 * agb_plot_extend: StartZFromZero (:766-769) + AddRandomPoints (:775-815, as upstream every added point equals the
 *   per-axis minimum) + CopyJitterRandomPoints (:818-873): pos2 = per plot [pos1 with z - zmin] ++ [n_add x minimum] ++
 *   [pos_prev[cj_idx] + cj_noise]; ptr2 / elem2 = output layout (host-known counts).
 * agb_plot_crop_ws: RandomPolygon2dExtend (:1502-1552) with one transformed polygon per plot (polys double[B][2*nv]); a plot
 *   with no point inside is left whole; emits positions, features [1, z, xy distance], source rows like agb_plot_prepare. */
int agb_plot_augment(const float* raw, const long long* sel, const int32_t* elem, const int32_t* ptr, int B, int n,
                     const float* aug, const float* noise, float* pos1, float* mins, void* stream);
int agb_plot_extend(const float* pos1, const int32_t* ptr1, const float* mins, const int32_t* ptr2,
                    const int32_t* elem2, int B, int n2, const int32_t* n_add, const int32_t* cj_ptr,
                    const long long* cj_idx, const float* cj_noise, float* pos2, void* stream);
int agb_plot_crop_ws(const float* pos, const int32_t* ptr, const int32_t* elem, int B, int n, const double* polys, int nv,
                     float fcx, float fcy, void* workspace /* agb_plot_workspace_bytes(n, B) */, float* pos_out,
                     float* x_out, long long* src, int32_t* out_ptr, int32_t* n_out_dev, void* stream);

/* ---- squeeze-excite excitation MLP (dpcr-agb_amd/csrc/se.hip) -------------------------------------------------------
 * SELayer.fc of modules/MinkowskiEngine/senet_block.py:35-42 on the pooled features P [B,C]:
 * S = sigmoid(W2 act(W1 P + b1) + b2), W1 [H,C], W2 [C,H] (nn.Linear layout), H <= 256, act 0 none / 1 relu / 2 gelu.
 * fwd writes h_pre [B,H] (kept for bwd) and S [B,C]; bwd: scratch dz2 [ceil(C/512)][B,C], dh [B,H]; out dP [B,C], dW1, db1, dW2, db2
 * (biases may be NULL).  Sums over the batch run in batch order (deterministic). */
int agb_se_mlp_fwd(const float* P, const float* W1, const float* b1, const float* W2, const float* b2, int B, int C,
                   int H, int act, float* h_pre, float* S, void* stream);
int agb_se_mlp_bwd(const float* P, const float* W1, const float* W2, int B, int C, int H, int act, const float* h_pre,
                   const float* S, const float* dS, float* dz2, float* dh, float* dP, float* dW1, float* db1,
                   float* dW2, float* db2, void* stream);

/* ---- MinkowskiPointNet shared MLP (dpcr-agb_amd/csrc/pointnet.hip) --------------------------------------------------
 * modules/MinkowskiEngine/PointNet.py:16-29: three (Linear without bias -> BatchNorm -> activation) layers over the points
 * of a batch and a per-plot pooling of the last one.  The last layer's BatchNorm + activation is fused INTO the pooling
 * (the [n, C] activation is never written) and, in the backward pass, into the BatchNorm gradient (the broadcast pooled
 * gradient is never written).
 * agb_pointnet_pool_fwd: pooled[b,:] = reduce over rows ptr[b]..ptr[b+1] of act(gamma (Z - mean) rstd + beta); mode 0 sum,
 *   1 average, 2 max (+ argmax int32[B, C]: the winning row).  mean / rstd from agb_bn_stats on Z.  splits =
 *   agb_pointnet_pool_splits(n, B) row splits per plot, folded in order (part float[B*splits*C], part_arg int32[B*splits*C]
 *   for max; unused when splits == 1).
 * agb_pointnet_pool_bwd: dZ [n, C] (the operand of the following weight / data gradient products), dgamma, dbeta from
 *   dpooled [B, C]; coords int32[n][4] (plot index in column 0); part: float[agb_bn_chunks(n) * 2 * C].
 * agb_pointnet_mlp_fwd: the whole chain for C callers / inference: x [n, cin_pad] -> pooled [B, c3]; W_l [c_{l-1}, c_l]
 *   row-major (nn.Linear.weight transposed), widths multiples of 4 (cin_pad >= 12: zero-pad the 6 input columns);
 *   bn_l = HOST array of 4 device pointers {gamma, beta, running_mean, running_var}; workspace:
 *   agb_pointnet_mlp_workspace_bytes(n, B, c1, c2, c3) bytes.  The products run on the identity-map convolution
 *   kernels (agb_spconv_fwd_ex with nbr == NULL), no BLAS. */
int agb_pointnet_pool_splits(int n, int B);  /* host helper */
int agb_pointnet_pool_fwd(const float* Z, int ldz, int n, int C, const float* mean, const float* rstd, const float* gamma,
                          const float* beta, int act, const int32_t* ptr, int B, int mode, int splits, float* part,
                          int32_t* part_arg, float* pooled, int32_t* argmax, void* stream);
int agb_pointnet_pool_bwd(const float* Z, int ldz, int n, int C, const int32_t* coords, const int32_t* ptr, int B,
                          const float* dpooled, const int32_t* argmax, int mode, const float* mean, const float* rstd,
                          const float* gamma, const float* beta, int act, int training, float* part, float* dZ, int lddz,
                          float* dgamma, float* dbeta, void* stream);
/* The same two with the forward's per-plot sums of act'(.) and act'(.) * zhat (aux float[2][B][C], sum / avg pooling;
 * aux_part float[2][B * splits][C] scratch when splits > 1): the backward then takes dgamma / dbeta from B terms per channel
 * instead of a pass over Z (`part` may be NULL).  Max pooling needs no aux: its sums are gathered through argmax. */
int agb_pointnet_pool_fwd_aux(const float* Z, int ldz, int n, int C, const float* mean, const float* rstd,
                              const float* gamma, const float* beta, int act, const int32_t* ptr, int B, int mode,
                              int splits, float* part, int32_t* part_arg, float* pooled, int32_t* argmax, float* aux_part,
                              float* aux, void* stream);
int agb_pointnet_pool_bwd_aux(const float* Z, int ldz, int n, int C, const int32_t* coords, const int32_t* ptr, int B,
                              const float* dpooled, const int32_t* argmax, int mode, const float* mean, const float* rstd,
                              const float* gamma, const float* beta, int act, int training, float* part, const float* aux,
                              float* dZ, int lddz, float* dgamma, float* dbeta, void* stream);
size_t agb_pointnet_mlp_workspace_bytes(int n, int B, int c1, int c2, int c3);
int agb_pointnet_mlp_fwd(const float* x, int ldx, int n, int cin_pad, const float* W1, const float* const* bn1, int c1,
                         const float* W2, const float* const* bn2, int c2, const float* W3, const float* const* bn3,
                         int c3, int act, float eps, float momentum, int training, const int32_t* ptr, int B, int mode,
                         void* workspace, float* pooled, int32_t* argmax, void* stream);

/* ---- bf16-row twins (see agb_spconv_fwd_h): BatchNorm / residual / SE-tail kernels (norm.hip) and pooling / broadcast
 * kernels (pool.hip) on uint16 bf16 row matrices; per-plot matrices, statistics, partials and argmax buffers as in the fp32
 * forms.  Reference call sites: those of the fp32 entry points. */
int agb_bn_stats_tracked_h(const uint16_t* X, int ldx, int n, int C, float eps, float momentum, int training,
                           float* part, float* mean, float* rstd, float* running_mean, float* running_var,
                           long long* num_batches_tracked, void* stream);
int agb_bn_stats_h(const uint16_t* X, int ldx, int n, int C, float eps, float momentum, int training, float* part,
                   float* mean, float* rstd, float* running_mean, float* running_var, void* stream);
int agb_bn_act_fwd_h(const uint16_t* X, int ldx, int n, int C, const float* mean, const float* rstd,
                     const float* gamma, const float* beta, int act, uint16_t* Y, int ldy, void* stream);
int agb_bn_act_bwd_colsum_h(const uint16_t* X, int ldx, const uint16_t* dY, int ldy, int n, int C, const float* mean,
                            const float* rstd, const float* gamma, const float* beta, int act, int training,
                            float* part, uint16_t* dX, int lddx, float* dgamma, float* dbeta, float* colsum,
                            void* stream);
int agb_bn_act_bwd_h(const uint16_t* X, int ldx, const uint16_t* dY, int ldy, int n, int C, const float* mean,
                     const float* rstd, const float* gamma, const float* beta, int act, int training, float* part,
                     uint16_t* dX, int lddx, float* dgamma, float* dbeta, void* stream);
int agb_se_tail_stats_h(const uint16_t* Z, int ldz, const int32_t* ptr, int n, int C, int B, float eps,
                        float momentum, int training, float* part, float* mean, float* rstd, float* running_mean,
                        float* running_var, long long* num_batches_tracked, void* stream);
int agb_se_tail_fwd_h(const uint16_t* Z, int ldz, const uint16_t* R, int ldr, const int32_t* coords,
                      const float* mean, const float* rstd, const float* gamma, const float* beta, const float* s,
                      const float* keep, int act, int n, int C, uint16_t* Y, int ldy, void* stream);
int agb_se_tail_bwd_sums_h(const uint16_t* Z, int ldz, const uint16_t* R, int ldr, const uint16_t* dY, int ldy,
                           const int32_t* ptr, int B, const float* mean, const float* rstd, const float* gamma,
                           const float* beta, const float* s, const float* keep, int act, int n, int C, float* spart,
                           void* stream);
int agb_se_tail_bwd_apply_h(const uint16_t* Z, int ldz, const uint16_t* R, int ldr, const uint16_t* dY, int ldy,
                            const int32_t* coords, const float* mean, const float* rstd, const float* gamma,
                            const float* beta, const float* s, const float* keep, const float* dte,
                            const float* dbeta, const float* dgamma, int act, int training, int n, int C,
                            uint16_t* dZ, int lddz, uint16_t* dR, int lddr, void* stream);
int agb_add_act_fwd_h(const uint16_t* A, int lda, const uint16_t* R, int ldr, const float* scale,
                      const int32_t* coords, int n, int C, int act, uint16_t* Y, int ldy, void* stream);
int agb_add_act_bwd_h(const uint16_t* A, int lda, const uint16_t* R, int ldr, const float* scale,
                      const int32_t* coords, const uint16_t* dY, int ldy, int n, int C, int act, uint16_t* dA,
                      uint16_t* dR, void* stream);
int agb_maxpool_fwd_h(const uint16_t* X, int ldx, const int32_t* nbr, long long nbr_stride, uint16_t* Y, int ldy,
                      int32_t* argmax /* [n_out, C] */, int n_out, int K3, int C, void* stream);
int agb_maxpool_bwd_h(const uint16_t* dY, int ldy, const int32_t* argmax, const int32_t* nbrT, long long nbrT_stride,
                      uint16_t* dX, int ldx, int n_in, int K3, int C, void* stream);
int agb_maxpool_fwd_k_h(const uint16_t* X, int ldx, const int32_t* nbr, long long nbr_stride, uint16_t* Y, int ldy,
                        uint8_t* argk, int n_out, int K3, int C, void* stream);
int agb_maxpool_bwd_k_h(const uint16_t* dY, int ldy, const uint8_t* argk, const int32_t* nbrT, long long nbrT_stride,
                        uint16_t* dX, int ldx, int n_in, int K3, int C, void* stream);
int agb_segment_reduce_h(const uint16_t* A, int lda, const uint16_t* Bm, int ldb, const int32_t* ptr, int B, int C,
                         int mode, int splits, float* part, int32_t* part_arg, float* Y, int32_t* argmax,
                         void* stream);
int agb_segment_broadcast_h(const float* S, const int32_t* coords, const int32_t* ptr, const uint16_t* M, int ldm,
                            uint16_t* out, int ldo, int n, int C, int average, void* stream);
int agb_segment_scale_add_h(const float* S, const float* T, const int32_t* coords, const int32_t* ptr,
                            const uint16_t* M, int ldm, uint16_t* out, int ldo, int n, int C, void* stream);
int agb_segment_max_bwd_h(const float* dY, const int32_t* argmax, uint16_t* dX, int ldx, int B, int C, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * One call per network block (dpcr-agb_amd/csrc/net.hip).  Replaces, as ONE entry point per direction,
 *   modules/MinkowskiEngine/SENet.py:47-53      the stem: conv K^3 (3 -> 64) -> norm -> activation -> max pool 3^3 stride 2
 *   modules/MinkowskiEngine/senet_block.py:80-96 on resnet_block.py:62-73   SEBasicBlock.forward: conv1 -> norm1 -> act ->
 *       conv2 -> norm2 -> squeeze-excite -> drop path -> (+ downsample(x) | x) -> act
 * and their backward passes (ME + autograd in the reference).  The block is a fixed operator sequence; driving it launch by
 * launch from the host language costs the enqueuing thread more than the device needs for the small kernels.  The library
 * enqueues the SAME kernels, in the same order and geometry, as the per-operator entry points above: bit-identical results.
 *
 * f: HOST table of 64-bit fields (device pointers, sizes, options).  agb_net_fields() returns the comma-separated field
 * names in index order (agb_net_field_count() of them) so that a binding builds its index map from the library itself:
 *   x ldx n_in n_out B coords ptr y ldy | dy lddy dx lddx need_dx | training act stride has_down | cmp_mode cmp_il
 *   dw_variant det persistent | per convolution c1 / c2 / cd (downsample): _w _b _K3 _cin _cout, its BatchNorm _g _be _rm
 *   _rv _nbt _eps _mom (eps, mom: IEEE-754 double bit patterns), kernel maps _nbr _nbr_ld _nbrT _nbrT_ld, the class
 *   partition of a strided data gradient _perm _tile_cls _cls_tab _n_tiles, balanced tile tables _tf _tf_t _tf_b (forward)
 *   _tb _tb_t _tb_b (data gradient), _wt (W^T kept by the caller, or 0), gradients out _dw _db _dg _dbe | se_act se_H se_w1
 *   se_b1 se_w2 se_b2 keep (drop-path scale float[B] or 0) d_se_w1 d_se_b1 d_se_w2 d_se_b2 | stem only: feat ldf fdim
 *   grid desc (HOST int32[8], as agb_spconv_fwd3_grid) K pool_nbr pool_nbr_ld pool_nbrT pool_nbrT_ld pool_K3 n_pool.
 * saved: device arena the forward pass fills and the backward pass of the same call table reads (agb_net_*_bytes(f, 0));
 * scratch: device arena of temporaries (agb_net_*_bytes(f, 1) forward, (f, 2) backward); both 256-byte aligned.
 * The backward entry points take BatchNorm in batch-statistics mode (training != 0); the forward ones both modes. */
const char* agb_net_fields(void);
int agb_net_field_count(void);
size_t agb_net_stem_bytes(const int64_t* f, int which);
int agb_net_stem_fwd(const int64_t* f, void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, void* stream);
int agb_net_stem_bwd(const int64_t* f, void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, void* stream);
size_t agb_net_block_bytes(const int64_t* f, int which);
int agb_net_block_fwd(const int64_t* f, void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, void* stream);
int agb_net_block_bwd(const int64_t* f, void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, void* stream);

/* ---- regression head + loss (dpcr-agb_amd/csrc/head.hip) --------------------------------------------------------------
 * models/instance/minkowski.py:16-26 (SeparateLinear: one nn.Linear(C, 1) per regression target on the pooled features) and
 * models/instance/base.py:139-146,154-179 (targets standardised with the train statistics: d = out - (y - center) / scale;
 * smooth-L1 (beta 1) / L2 / L1 with mean reduction, summed over the configured functions; weighted by mean(task weights)).
 * One launch forward, one backward.  W, bias, dW, dbias: HOST arrays of T <= 8 DEVICE pointers (weight [C], bias [1] of
 * every target's layer; bias / gradient pointers may be NULL).  pooled [B][ldp]; y [B][T]; center, scale, weights [T].
 * loss_mask: 1 smooth-L1 | 2 L2 | 4 L1.  fwd out: out [B][T], dout [B][T] = dloss/dout (kept for bwd), loss_reg and loss
 * (device scalars).  bwd: gloss = device scalar arriving at `loss` (NULL: 1); dW_t [C], dbias_t [1], dpooled [B][lddp] or
 * NULL; the plots are summed in order (deterministic). */
int agb_reg_head_fwd(const float* pooled, int ldp, int B, int C, int T, const float* const* W, const float* const* bias,
                     const float* y, const float* center, const float* scale, const float* weights, int loss_mask, float* out,
                     float* dout, float* loss_reg, float* loss, void* stream);
int agb_reg_head_bwd(const float* pooled, int ldp, int B, int C, int T, const float* const* W, const float* dout,
                     const float* gloss, float* const* dW, float* const* dbias, float* dpooled, int lddp, void* stream);

/* ---- the names SURVEY.md section 8(b) gave this ABI (dpcr-agb_amd/csrc/aliases.hip) ------------------------------------
 * agb_hash_build = agb_coords_insert.  agb_kpconv_fwd / agb_kpconv_bwd: the whole rigid KPConv layer
 * (modules/KPConv/blocks.py:264-400) as one call each: y = wf @ W with wf[n,k,:] = sum_h infl(n,h,k) x[idx[n,h],:] kept in
 * wf [N][K*Cin] for the backward pass; backward: dW [K*Cin][Cout] += wf^T dy, dx [Ns][ldx] += gather^T(dy @ W^T) (both
 * zero-filled by the caller; either may be NULL), workspace of agb_kpconv_bwd_workspace_bytes bytes.
 * (agb_spconv_bwd_data is declared with the convolution entry points above.) */
int agb_hash_build(const int32_t* coords, int n, const int32_t* n_dev, uint64_t* keys, int32_t* vals, int cap,
                   int32_t* slot_of_row, int32_t* status, void* stream);
int agb_kpconv_fwd(const float* q, const float* s, const int32_t* idx, int H, int Ns, const float* x, int ldx, const float* kp,
                   int K, float extent, const float* W, float* wf, float* y, int ldy, int N, int Cin, int Cout, void* stream);
size_t agb_kpconv_bwd_workspace_bytes(int N, int K, int Cin, int Cout);
int agb_kpconv_bwd(const float* q, const float* s, const int32_t* idx, int H, int Ns, const float* wf, const float* dy, int lddy,
                   const float* kp, int K, float extent, const float* W, float* dx, int ldx, float* dW, int N, int Cin, int Cout,
                   void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AGB_HIP_H */
