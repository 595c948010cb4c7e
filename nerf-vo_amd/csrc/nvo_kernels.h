// Internal launcher declarations (C++ linkage) shared between the .hip translation units.
#pragma once
#include "nvo_common.h"

// ---- fill (adam.hip) -------------------------------------------------------------------------
// Zero `bytes` (multiple of 4) of device memory with a KERNEL.  hipMemsetAsync is avoided on every
// path that can be captured into a hipGraph: on ROCm 7.2 a captured 4-byte memset node replayed as
// byte value 0x01 (observed: the optimiser's skip flag read back 0x01010101 after graph replay).
int nvo_zero_async(void* ptr, size_t bytes, hipStream_t stream);

// ---- grid.hip -------------------------------------------------------------------------------
// format of the dL/d(encoded) pairs handed to the backward launchers (`dy_fmt`)
enum { NVO_DY_HALF = 0, NVO_DY_FLOAT = 1, NVO_DY_BF16 = 2 };
struct NvoGridSlices {
    uint32_t n_slices = 0;
    uint32_t* d_level = nullptr;   // device array of uint4 work items {level, first, chunk, n_chunks}
    uint32_t* d_first = nullptr;
    uint32_t zero_first = 0, zero_last = 0;  // entry range flushed with atomics (zeroed per launch)
    uint32_t acc_bits = 64;                  // 32: int32 accumulators with the L1-derived scale (set before create)
    uint32_t level_mask = 0xFFFFFFFFu;       // (set by create) levels that have items: the L1 pre-pass reads only those
    unsigned long long* d_l1 = nullptr;      // [levels][2] L1 norms of dy (inside the d_level allocation)
    uint32_t* d_live_n = nullptr;            // length of the live-sample list (inside the d_level allocation, before d_l1)
    // dynamic LDS the launch asks for: 160 KiB only when an item accumulates in fp32 (20K-entry slices), else 128 KiB
    // -- which leaves 32 KiB of a CU's LDS to a concurrently running kernel (the record scatter of mode 3)
    uint32_t lds_bytes = 160 * 1024;
    // The caller zeroes what the launch would zero (nvo_bwd_zero_ranges / option external_zero): a training step then
    // clears every accumulate-into buffer of all its networks with ONE launch instead of ~6 dependent 5-us launches.
    bool external_zero = false;
    // option grid_compact_live: the samples whose dL/dy is non-zero on ANY level are listed first (k_live_samples) and
    // the items scan that list instead of all N samples.  For the proposal networks of a nerfacto run 87-98 % of the
    // samples carry an exactly zero gradient from a few hundred steps on, and every one of them
    // was loaded and tested once per slice (25 slice scans for a proposal grid).
    bool compact_live = false;
    mutable NvoScratch live;              // [1 + N] uint32: count, then the live sample ids (grows with N; graph-safe)
    mutable NvoScratch codes;             // [coded levels][N] uint16: slice codes of the hashed levels (k_slice_codes)
    // option grid_bwd_runs (set before create): the items of DENSE levels scan with run merging -- a lane takes 8
    // consecutive samples, sums the corner contributions in registers while the cell stays the same and goes to the LDS
    // accumulators once per run (consecutive samples are neighbours on a ray, so a coarse cell holds a run of them)
    bool runs = false;
    // option grid_bwd_batch (set before create): the batch size the launches will see.  With it (and runs) every slice
    // gets the same number of chunks, chosen so that ALL items of the launch are resident at once (one item per CU)
    // and a chunk is a whole number of 8192-sample passes (1024 lanes x 8 consecutive samples): an item costs ~8 us of
    // dispatch, zeroing and flushing whatever it scans, so one round of long items beats several rounds of short ones.
    uint32_t batch_hint = 0;
    uint32_t dense_share_pct = 100;  // option grid_bwd_dense_share: chunks of a dense slice relative to the even split (percent)
    uint32_t fixed_cap = 0;  // (set before create) entries per 64-bit fixed-point slice of a dense level; 0 = 8192
    // (set before create) deterministic mode: every slice is ONE work item (no chunks meeting in float atomics), integer
    // accumulators on every level (LDS float atomics retire in no fixed order), no live-sample list (its append order
    // changes the run sums).  The gradient is then bitwise reproducible; the launch is several times slower.
    bool deterministic = false;
    // (nullable) device word OR-ed with 1 when an item meets a non-finite dL/dy (it also poisons its slice's first
    // gradient entry): the optimiser's overflow flag raised at the source -- the k_tl_accumulate passes of a stream
    // layout use their owner's word
    uint32_t* nf_flag = nullptr;
    // (set per launch by NetworkWithInputEncoding's backward, cleared afterwards) per-workgroup L1 sums of dL/d(encoded)
    // written by the fused-MLP backward that produced it: ext_l1[block][ext_l1_stride] floats, column 2 * level + feature;
    // ext_live[block] = samples with a non-zero dL/doutput (nullable).  See NvoMlpArgsT::dx_l1_partial.
    mutable const float* ext_l1 = nullptr;
    mutable const uint32_t* ext_live = nullptr;
    // (set per launch with ext_live) dL/doutput of that network when it is ONE 16-bit value per sample (compact output):
    // k_live_samples lists the samples from it instead of reading dL/d(encoded) of every level
    mutable const uint16_t* ext_dout = nullptr;
    mutable uint32_t ext_blocks = 0, ext_l1_stride = 0;
    // (set per launch by NetworkWithInputEncoding's backward -- module option "bwd_tile_live_ptr" -- and cleared afterwards;
    // streamed layouts, non-deterministic mode) the network's dL/doutput as rows of 16 16-bit values with one byte per
    // 16-sample tile that promises all-zero rows where (byte & ext_tile_bits) == 0: k_live_rows lists the samples with a
    // non-zero row reading only the live tiles, and both the slice-owner items and the record scatter walk that list.
    // ext_tile_count (nullable): 64 shards, 8 floats apart, whose sum is the number of live tiles -- while 3/4 of the
    // tiles or more are live nobody needs the list and the pass leaves at once.
    mutable const uint8_t* ext_tile_live = nullptr;
    mutable uint32_t ext_tile_bits = 0;
    mutable const void* ext_rows = nullptr;
    mutable const float* ext_tile_count = nullptr;
    mutable const uint32_t* ext_list = nullptr;  // (set by the stream launcher around its owner launch) the list to walk
    // (set per launch) the network's backward has written the list itself (NvoMlpArgsT::live_rows): `live` holds it, d_live_n
    // its length -- no k_live_rows pass
    mutable bool ext_list_given = false;
};
#include <utility>
#include <vector>
typedef std::vector<std::pair<void*, size_t>> NvoZeroRanges;
void nvo_grid_slices_zero_ranges(const NvoGridLevels& g, const NvoGridSlices* s, float* grad, NvoZeroRanges* out);
// level_mask: bit l set -> level l gets slice-owner work items (default: all levels); target_items: the
// chunk counts are scaled until the launch has about this many work items
int nvo_grid_slices_create(const NvoGridLevels& g, NvoGridSlices* s, uint32_t level_mask = 0xFFFFFFFFu,
                           uint32_t target_items = 1024, bool env_items = true);

// Streamed backward (mode 3): levels with many 4K-entry bins go through a scatter of self-contained 12-byte pair records,
// bin-sorted inside each tile, and a streaming accumulate into packed 2 x 32-bit LDS sums; the coarse levels keep
// slice-owner items.
// Adam fused into the tile-local accumulate pass (k_tl_accumulate_p): the single-item bins of the streamed HASHED levels
// hold their finished gradient in LDS when they flush, so the optimiser step of those entries happens right there --
// the gradient is neither written (4 B per parameter) nor read again by the optimiser launch (4 B), and that launch
// shrinks to the rest of the group.  Legal because the group's overflow verdict is final before the pass starts: every
// producer of its flag word (loss kernels, fused-MLP backwards, slice-owner items, the scatter pass that marks poisoned
// records) precedes it in stream order.  Pointers are to the ENCODING's first parameter in each flat buffer.
struct NvoGridAdam {
    float* params = nullptr;            // fp32 master weights (nullptr = off: gradients are stored as usual)
    void* params_half = nullptr;        // 16-bit working copy (fp16 tables)
    float* exp_avg = nullptr;
    float* exp_avg_sq = nullptr;
    const float* hyper_dev = nullptr;   // [0] = learning rate (nullable: lr)
    const float* bias_dev = nullptr;    // {1 - beta1^t, sqrt(1 - beta2^t)} of the group's next applied step
    const float* loss_scale_dev = nullptr;  // nullable: grad_scale
    const uint32_t* skip_flag = nullptr;    // the group's overflow flag word (non-zero: no step)
    float lr = 0.f, grad_scale = 1.f, beta1 = 0.9f, beta2 = 0.999f, eps = 1e-15f;
    float bias1 = 1.f, bias2_sqrt = 1.f;    // (bias_dev == nullptr) host-computed corrections
    // tcnn EmaOptimizer on the same entries (k_ema_update_dev's arithmetic): ema == nullptr = off
    float* ema = nullptr;
    void* ema_half = nullptr;
    float ema_decay = 0.f;
    const uint32_t* ema_step_dev = nullptr;
};

struct NvoGridStream {
    bool created = false;
    uint32_t n_levels = 0, n_bins = 0, max_slices = 0;
    uint32_t streamed_mask = 0;
    uint32_t owner_max_slices = 24;   // levels with at most this many 4K-entry bins stay slice-owner (measured optimum: levels 0-3 of the main grid)
    uint32_t tile = 512;              // samples per scatter workgroup (512 | 1024)
    uint32_t* d_meta = nullptr;       // one allocation holding the arrays below
    uint32_t* d_levels = nullptr;     // [n_levels] streamed level ids
    uint32_t* d_bin_first = nullptr;  // [n_levels + 1]
    uint32_t* d_bin_level = nullptr;  // [n_bins]
    uint32_t* d_bin_slice = nullptr;  // [n_bins]
    uint32_t* d_bin_chunks = nullptr; // [n_bins]
    NvoScratch work;                  // records | segment tables (sized for the largest batch seen; graph-safe growth)
    NvoGridSlices owner;
    // The slice-owner items of the coarse levels and the record pipeline of the streamed levels touch disjoint
    // gradient ranges, so they CAN run side by side (the former on this auxiliary stream, forked from / joined to the
    // caller's stream, also inside a graph capture).  Measured on the full step: 0.781 / 0.785 ms with the fork vs
    // 0.778 / 0.775 ms back to back -- the 512 slice-owner items already fill the CUs -- so the default is off
    // (option grid_stream_overlap / NVO_GRID_STREAM_OVERLAP=1).
    bool overlap = false;
    // Record layout: every (tile, level) keeps its bin-sorted 12-byte pair records in a fixed region + a [bin][tile]
    // segment table; accumulate items are static (grid.hip).  Accumulators of the record pass: two 32-bit fixed-point sums
    // in ONE 64-bit word (one LDS atomic per record), 8192-entry bins, overflow-proof scale from per-(tile, bin) L1 bounds
    // the scatter delivers with its rank atomics (k_tl_scatter_p).  (Rounds 2-3 also carried a globally bin-sorted layout
    // with count / scan passes and 64-bit accumulators over 4096-entry bins: 0.769 vs 0.744 ms per step, removed.)
    uint32_t bin_entries = 4096;      // (set by create)
    uint32_t dense_chunks = 8;        // tile-range chunks per bin of a streamed DENSE level (clustered samples)
    uint32_t* d_tl_items = nullptr;   // uint4 {bin, chunk | n_chunks << 16, streamed-level index | level << 8, slice}
    uint32_t n_tl_items = 0;
    uint32_t n_tl_slots = 0;          // (packed form) persistent workgroups the balanced item list was laid out for; 0 = dealt
    hipStream_t aux = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool external_zero = false;  // see NvoGridSlices::external_zero
    bool deterministic = false;  // (set before create) one accumulate item per bin on dense levels too; owner: see there
    NvoGridAdam adam;            // optimiser step inside the accumulate pass, see NvoGridAdam
};
// entries [first, first + n) of the table (in ENTRIES: two parameters each) whose Adam step NvoGridStream::adam takes over:
// the streamed hashed levels (one accumulate item per bin); n = 0 when the configuration has none
void nvo_grid_stream_adam_range(const NvoGridLevels& g, const NvoGridStream* st, uint64_t* first, uint64_t* n);
// false: this configuration zeroes data-dependent ranges (globally sorted layout) and cannot hand the zeroing over
bool nvo_grid_stream_zero_ranges(const NvoGridLevels& g, const NvoGridStream* st, float* grad, NvoZeroRanges* out);
int nvo_grid_stream_create(const NvoGridLevels& g, NvoGridStream* st);
void nvo_grid_stream_destroy(NvoGridStream* st);
int nvo_grid_bwd_stream_launch(const NvoGridLevels& g, NvoGridStream* st, hipStream_t stream, uint32_t N,
                               const float* x, const void* dy, int dy_fmt, bool soa, float* grad);
void nvo_grid_slices_destroy(NvoGridSlices* s);
// dydx_half (optional): [L][3][N] half2, d(out)/d(cell coordinate) for nvo_grid_bwd_input_dydx_launch
// out_bf16: the encoded features leave as bfloat16 pairs instead of fp16 pairs (bf16 MLP mode)
int nvo_grid_fwd_launch(const NvoGridLevels& g, hipStream_t stream, uint32_t N, const float* x,
                        const void* table_half, void* out_half, bool soa, uint32_t* indices,
                        void* dydx_half = nullptr, bool out_bf16 = false, const uint32_t* n_live = nullptr,
                        bool runs = false, int small_form = -1);
// small_form: form of the small-grid forward (5-level grids whose two coarsest levels fit the LDS): -1 = the default
// (NVO_GRID_FWD_SMALL, else the instruction-lean form), 0 = the generic kernel, 1 plain, 2 two samples per thread,
// 3 software-pipelined, 4 instruction-lean + pipelined.  All forms produce the same bits.
int nvo_grid_bwd_input_dydx_launch(const NvoGridLevels& g, hipStream_t stream, uint32_t N, const void* dydx_half,
                                   const void* dy, int dy_fmt, bool soa, float* dx, bool zero_dx);
int nvo_grid_bwd_launch(const NvoGridLevels& g, const NvoGridSlices* slices, hipStream_t stream,
                        uint32_t N, const float* x, const void* dy, int dy_fmt, bool soa,
                        float* grad, int mode);
// scratch (optional, module-owned): per-level partial gradients [L][N][3] of the two-stage input backward
typedef NvoScratch NvoGridInputScratch;
int nvo_grid_bwd_input_launch(const NvoGridLevels& g, hipStream_t stream, uint32_t N,
                              const float* x, const void* table_half, const void* dy,
                              int dy_fmt, bool soa, float* dx, bool zero_dx,
                              NvoGridInputScratch* scratch = nullptr);

// ---- sh.hip ---------------------------------------------------------------------------------
int nvo_sh_fwd_launch(hipStream_t stream, uint32_t N, uint32_t degree, const float* d01,
                      void* out_half, uint32_t out_stride, uint32_t out_width, bool out_bf16 = false);
int nvo_sh_bwd_input_launch(hipStream_t stream, uint32_t N, uint32_t degree, const float* d01,
                            const void* dy_half, uint32_t dy_stride, float* dd01);

// ---- mlp.hip --------------------------------------------------------------------------------
enum { NVO_ACT_NONE = 0, NVO_ACT_RELU = 1, NVO_ACT_SIGMOID = 2 };
enum {
    NVO_IO_F32_ROWS = 0,        // [B][n_in] float, columns >= n_in read as 1.0 (identity-encoding pad)
    NVO_IO_HALF2_SOA = 1,       // [n_in/2][B] half2 (grid encoding output), missing levels read as 0
    NVO_IO_HALF_ROWS = 2,       // [B][IN_PAD] half
    NVO_IO_NERFACTO_COLOR = 3,  // 64-wide row assembled on the fly: [SH16(ray) | geo15 | embed32(cam) | 1]
    NVO_IO_NGP_RGB = 4,         // 32-wide row: [density-net output 16 | SH16(ray of the packed sample)]
    NVO_IO_GRID_FUSED = 5,      // (forward only) the hash-grid encoding is evaluated inside the operand load: positions
                                // in, LDS-free register chain out; the encoded features never make the round trip
                                // through HBM between two kernels (they are still stored once when a backward follows)
};

// E: the 16-bit element type of everything the network streams (_Float16 | __bf16).  The layout does not depend
// on E (pointers only), so host code fills an NvoMlpArgs (= NvoMlpArgsT<_Float16>) whatever `bf16` says and the
// bf16 kernels reinterpret it.
template <typename E>
struct NvoMlpArgsT {
    uint32_t batch;       // multiple of 16
    uint32_t n_in;        // true input width (<= IN_PAD)
    int in_mode;
    const void* input;
    const E* weights;  // layer-major, each [out][in] row-major
    E* output;         // [B][OUT_PAD]   (compact_out: [B], column 0 only)
    int bf16;                 // 0: E = _Float16 (tcnn's precision), 1: E = __bf16 (mlp_bf16.hip)
    int compact_out;          // 1: output / doutput hold column 0 only (level-major half2 input, ReLU nets)
    int recompute_hidden;     // 1 (backward): `hidden` is not read, the single hidden layer is recomputed
    E* hidden;         // [N_HIDDEN][B][WIDTH] or nullptr (inference)
    int act, out_act;
    // backward only
    const E* doutput;  // [B][OUT_PAD], loss-scaled   (compact_out: [B])
    void* dinput;             // layout din_mode, nullable
    int din_mode;
    float* dweights;          // fp32, same order as weights, accumulated with atomics (pre-zeroed)
    // NVO_IO_NERFACTO_COLOR only (NerfactoField colour head input, never materialised in HBM)
    uint32_t samples_per_ray;
    const E* sh;         // [R][16]  SH of the ray direction
    const E* base_out;   // [B][16]  base MLP output: col 0 density pre-activation, 1..15 geo
    const E* embedding;  // [F][32]  appearance embedding (fp16 working copy)
    const int32_t* cam_idx;     // [R] or nullptr -> row 0 of `embedding` for every ray
    E* d_base_out;       // [B][16]  cols 1..15 written by the backward
    float* d_embedding;         // [F][32]  atomically accumulated, nullable
    float* d_sh;                // [R][16]  atomically accumulated, nullable
    // NVO_IO_NGP_RGB only (instant-ngp rgb head on packed samples)
    const int32_t* sample_ray;    // [B] ray index of each packed sample (< 0: empty slot)
    const float* d_extra_col0;    // [B] added to column 0 of d_base_out (dL/d density pre-activation)
    // NVO_IO_GRID_FUSED only: input = positions [B][3] float
    const NvoGridLevels* grid;    // DEVICE copy of the level table
    const void* grid_table;       // fp16 [entries][2]
    void* enc_out;                // [L][B] pairs of E (what a separate encoding kernel would have written), nullable
    // deterministic mode (backward; nullable): every workgroup STORES its dW block total to dw_partial[block][n_weights]
    // and a second launch sums the blocks in a fixed order (the default adds them with float atomics, whose order
    // varies); NVO_IO_NERFACTO_COLOR: the per-tile sums of the embedding / SH-direction gradients go to
    // tile_partial[tile][48] = {embedding 32 | d_sh 16} instead of float atomics (nvo_color_tile_reduce sums them)
    float* dw_partial;
    // (backward; nullable) dw_n_replicas further copies of the weight-gradient buffer ([r][n_weights] floats, ZERO on
    // entry): workgroup b adds its block total to copy b % (dw_n_replicas + 1) (0 = dweights itself).  Every workgroup
    // adding to the same few cache lines serialises at the L2's atomic units -- 7-10 us per launch, a quarter of the
    // kernels; nvo_fold_replicas sums the copies into dweights (and clears them) once per step.
    float* dw_replicas;
    uint32_t dw_n_replicas;
    float* tile_partial;
    // NVO_IO_HALF2_SOA backward with dinput (networks behind a hash grid; nullable): every workgroup STORES the L1 norm of
    // the dL/dinput values it wrote, per input column, to dx_l1_partial[block][IN_PAD] (and, compact_out, the number of
    // its samples with a non-zero dL/doutput to dx_live_partial[block]).  The grid backward's 32-bit accumulators take
    // their overflow-proof scale from these sums instead of a pass of their own over dL/d(encoded) (k_dy_l1 / the L1 half
    // of k_live_samples: ~14 us per 1 M-sample launch), and k_live_samples leaves at once while most samples are live.
    float* dx_l1_partial;
    uint32_t* dx_live_partial;
    // (backward; nullable) OR-ed with 1 when a workgroup's dW total is not finite: an overflow of the 16-bit chain INSIDE
    // the network (a hidden dZ = inf with finite roots and leaves) always lands in the weight gradient of the layer it
    // appears in (dW = dZ^T H: inf * h = inf or NaN for every h) -- and in dL/d(embedding), dL/d(SH) only together with it
    uint32_t* nf_flag;
    // (forward; nullable) device count of the rows in use: tiles past it are not evaluated (`batch` stays the stride of
    // the level-major input) -- the pass of the occupancy-grid back-end that finds where each ray ends
    const uint32_t* n_live;
    // (backward, chain / dW roles, level-major and colour-head layouts; nullable) one byte per 16-sample tile, written by the
    // kernel that produced dL/doutput: (tile_live[t] & tile_live_bits) == 0 promises that all 16 x out_pad values of tile t
    // are exactly zero.  Such a tile adds nothing to any dW and its dX is zero: the workgroup compacts the live ones of
    // its tiles into a list, walks only those, and stores zeros as the dead tiles' dX (k_mlp_bwd, "LIVE-TILE LIST").
    const uint8_t* tile_live;
    uint32_t tile_live_bits;
    // (nullable) 64 shards, 8 floats apart, whose sum is the number of tiles with a non-zero byte (an upper bound of the
    // live ones): while 3/4 of the tiles or more are live the list is not built at all
    const float* tile_live_count;
    // (backward, chain / dW roles, level-major layout; nullable -- the launcher has checked nvo_mlp_bwd_lists_rows) the
    // kernel LISTS the samples whose dL/doutput row is not all zero while it walks its live tiles: live_rows[k] = sample id,
    // *live_rows_n += their number (one atomic per workgroup; the order of the workgroups is not deterministic).  Without
    // a tile list (no bytes, or most tiles live) it adds `batch` to the word instead: "all samples".  What the hash grid's
    // backward behind this network walks (NvoGridSlices::ext_list) -- without a pass of its own over the rows.
    uint32_t* live_rows;
    uint32_t* live_rows_n;
};
typedef NvoMlpArgsT<_Float16> NvoMlpArgs;
bool nvo_mlp_shape_supported(int in_pad, int width, int n_hidden, int out_pad);
// per-element-type entry points (mlp.hip / mlp_bf16.hip); nvo_mlp_{fwd,bwd}_launch dispatch on a.bf16
int nvo_mlp_fwd_launch_f16(int in_pad, int width, int n_hidden, int out_pad, const NvoMlpArgs& a, hipStream_t stream);
int nvo_mlp_bwd_launch_f16(int in_pad, int width, int n_hidden, int out_pad, const NvoMlpArgs& a, hipStream_t stream);
int nvo_mlp_fwd_launch_bf16(int in_pad, int width, int n_hidden, int out_pad, const NvoMlpArgs& a, hipStream_t stream);
int nvo_mlp_bwd_launch_bf16(int in_pad, int width, int n_hidden, int out_pad, const NvoMlpArgs& a, hipStream_t stream);
int nvo_mlp_fwd_launch(int in_pad, int width, int n_hidden, int out_pad, const NvoMlpArgs& a,
                       hipStream_t stream);
int nvo_mlp_bwd_launch(int in_pad, int width, int n_hidden, int out_pad, const NvoMlpArgs& a,
                       hipStream_t stream);
// workgroups the backward of this shape launches for `batch` rows (rows of dw_partial) and its weight count
uint32_t nvo_mlp_bwd_blocks(int in_pad, int width, int n_hidden, uint32_t batch);
// true when the backward of this shape and batch can list the live rows itself (NvoMlpArgsT::live_rows): role-split form,
// and a workgroup's tiles fit its LDS row buffer
bool nvo_mlp_bwd_lists_rows(int in_pad, int width, int n_hidden, uint32_t batch);
uint64_t nvo_mlp_n_weights(int in_pad, int width, int n_hidden, int out_pad);

// ---- deterministic reductions (adam.hip) ---------------------------------------------------------
// dst[e] += sum_b partial[b][e], b ascending
int nvo_reduce_partials(hipStream_t stream, const float* partial, uint32_t n_blocks, uint64_t n, float* dst);
// out[c][k] += sum over rows r with cam(r) == c, r ascending within 8 interleaved sub-sequences that are combined in a
// fixed order.  cam: int32 [R] (cam_i64x3 = 0) or int64 [R][3] column 0 (= 1); rows [R][row_stride], K <= 32 columns
int nvo_reduce_by_camera(hipStream_t stream, uint32_t R, uint32_t K, const float* rows, uint32_t row_stride,
                         const void* cam, int cam_i64x3, uint32_t F, float* out);
// colour head: tile_partial [R * tiles_per_ray][48] -> per_ray [R][48] (tiles ascending); d_sh[r] = per_ray[r][32..48)
int nvo_color_tiles_to_rays(hipStream_t stream, uint32_t R, uint32_t tiles_per_ray, const float* tile_partial,
                            float* per_ray, float* d_sh);
