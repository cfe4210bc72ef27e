/*
 * lstep_hip.h -- C ABI of the MI355X (gfx950) implementation of the L-STEP hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch / C++ types.  Every pointer marked
 * "device" must be a HIP device pointer valid on the current device; `stream` is a hipStream_t passed
 * as void* (NULL = default stream).  All calls are asynchronous on `stream` and allocate nothing.
 *
 * The reference (kthrn22/L-STEP) is 100 % Python and has no FFI of its own; each entry point below
 * replaces the native work hidden behind the cited reference lines (paths relative to the reference
 * repository root).  The Python host layer that mirrors the reference classes on top of this ABI is
 * l-step_amd/{sampler,model,engine}.py; INTEGRATION.md shows the ctypes stub a reference maintainer adds.
 *
 * Conventions
 *   N+1 rows in node tables (row 0 = padding), E+1 rows in the edge table (row 0 = padding).
 *   F = node/edge feature width, P = positional-encoding width, D = time-encoding width;
 *   F, P multiples of 4 and <= 256, D <= 128 (the reference uses F = P = 172, D = 100).
 *   Feature rows are dense fp32, row-major, 16-byte aligned (row stride F*4 bytes).
 *   Return value: LSTEP_OK or a negative error code; lstep_last_error() gives the message (thread local).
 */
#ifndef LSTEP_HIP_H
#define LSTEP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSTEP_ABI_VERSION 40

#define LSTEP_OK 0
#define LSTEP_EINVAL (-1) /* bad argument (NULL pointer, unsupported width, num_neighbors <= 0 ...) */
#define LSTEP_EHIP (-2)   /* a HIP runtime call or kernel launch failed */

/* Time-sorted undirected temporal adjacency in CSR form, device resident.
 * Content = reference utils/utils.py:292-299 (every edge appended to both endpoints) after the stable
 * per-node sort by timestamp of utils/utils.py:99; row r owns entries [indptr[r], indptr[r+1]). */
typedef struct lstep_csr {
    const int64_t* indptr; /* device, [num_rows + 1] */
    const int32_t* nbr;    /* device, [nnz] neighbour node id */
    const int32_t* eid;    /* device, [nnz] edge id */
    const double* ts;      /* device, [nnz] interaction time, non-decreasing inside a row */
    int64_t num_rows;      /* max node id + 1 */
    int64_t nnz;           /* 2 * number of edges */
    int64_t max_degree;    /* longest adjacency row (0 = unknown).  lstep_gather_aggregate_fwd shares node-channel rows of more than 256
                            * slots among a workgroup's waves; when no row can be that long the kernel skips the workgroup barriers of that path */
} lstep_csr_t;

/* A slot of the device-resident PE history ring whose index lives ON THE DEVICE: slot = (*start + add) % slots.  Entry points that take a
 * `const lstep_ring_ref_t* ring` use it INSTEAD of their host-side slot / rotation argument when it is non-NULL, and then interpret their
 * slot-pointer argument (the slot's rows) as the base of the whole ring: rows of the slot = base + slot * slot_stride floats.  With it
 * the launch sequence of a training iteration no longer depends on how far the ring has rotated, so the iteration can be captured
 * once as a HIP graph and replayed; lstep_ring_tick advances *start by one at the end of an iteration. */
typedef struct {
    const int32_t* start;   /* device: physical slot of the window's oldest snapshot */
    int32_t add;            /* 0 = oldest snapshot, window length = the slot being built, ... */
    int32_t slots;          /* physical slots of the ring */
    int64_t slot_stride;    /* floats between two slots */
} lstep_ring_ref_t;
int lstep_ring_tick(int32_t* start, int32_t slots, void* stream);

/* Branch selection for lstep_gather_aggregate_fwd/bwd */
#define LSTEP_BRANCH_EDGE_NODE 1u /* A + N: models/LSTEP.py:147-158,177-211 */
#define LSTEP_BRANCH_PE 2u        /* C:     models/LSTEP.py:223-238 */
#define LSTEP_WEIGHTED_SUM 4u     /* with LSTEP_BRANCH_EDGE_NODE: the `weighted_sum` ablation of the node channel, models/LSTEP.py:190-206 */

int lstep_abi_version(void);
const char* lstep_last_error(void);

/* Streams of the caller's own.  The reference runs everything on the framework's default stream (SURVEY.md 8b "Threading"); the host
 * layer here overlaps update_pe (models/LSTEP.py:268-340), the ring's copies and the parameter-gradient products with the critical chain
 * on side streams.  Those must be DEDICATED queues: a framework that hands streams out round-robin from a small pool (PyTorch: 32 per
 * device) sooner or later gives two roles -- or a role and its own graph-capture stream -- the same queue; work a second host thread
 * enqueues on a stream that another thread is capturing is recorded into that graph instead of executed (round 4's memory access fault).
 * lstep_stream_create: a non-blocking hipStream_t, `priority` clamped to the device's range (0 = default, negative = higher);
 * lstep_stream_destroy: the stream must be idle and not capturing. */
int lstep_stream_create(void** out_stream, int32_t priority);
int lstep_stream_destroy(void* stream);

/* Checked build (round 5; no reference counterpart: the reference's index errors are Python IndexErrors, utils/utils.py:160-169,
 * models/LSTEP.py:152,181,233).  A library compiled with -DLSTEP_BOUNDS_CHECK=1 (tools/build_checked.py -> liblstep_hip_checked.so, loaded
 * with LSTEP_LIB=...) compares every id-indexed load of its kernels -- neighbour / edge / node ids in the gather stage, row ids of
 * update_pe, of the history filter, of the loss and of the row scatters -- with the row count of the table it indexes; an index out of
 * range reads the padding row 0 instead of faulting the GPU and is recorded in a sticky device record.
 * lstep_debug_bounds_check_enabled: 1 in a checked build, 0 otherwise (the other two calls are then no-ops that report nothing).
 * lstep_debug_set_limits: rows of the node-shaped tables (node features, PE tables: N + 1) and of the edge table (E + 1), for the loads
 *   whose entry point carries no row count of its own (0 = unknown: not checked).
 * lstep_debug_device_error: waits for the device, returns {tag of the first offending load (0 = none), its index, the limit, number of
 *   offenders since the last call} and clears the record.  Tags: enum CheckTag in csrc/lstep_common.h. */
int lstep_debug_bounds_check_enabled(void);
int lstep_debug_set_limits(int64_t node_rows, int64_t edge_rows);
int lstep_debug_device_error(int64_t out[4]);

/* S -- NeighborSampler.get_historical_neighbors, 'recent' strategy (utils/utils.py:148-213, search :129-146).
 * Row r < min(num_ids, num_times): the last min(c, K) of the c interactions of node_ids[r] strictly earlier
 * than times[r], right-aligned in a zero row; rows >= min(num_ids, num_times) stay zero (zip truncation, :169).
 * out_nbr/out_eid int64 [num_ids, K], out_nt float32 [num_ids, K] (float64 -> float32 cast of :166),
 * out_count int32 [num_ids] = c (may be NULL).  Bit-exact with the reference. */
int lstep_sample_recent(const lstep_csr_t* csr, const int64_t* node_ids, int64_t num_ids, const double* times,
                        int64_t num_times, int32_t num_neighbors, int64_t* out_nbr, int64_t* out_eid, float* out_nt,
                        int32_t* out_count, void* stream);

/* T -- TimeEncoder.forward (models/modules.py:27-39): out[i, d] = cos(dt[i] * w[d] + b[d]); rows with
 * zero_mask[i] != 0 are written as 0 (the "neighbour id == 0" masking of models/LSTEP.py:154,231,316).
 * zero_mask may be NULL.  dt float32 [n], w/b float32 [D], out float32 [n, D]. */
int lstep_time_encode(const float* dt, const uint8_t* zero_mask, int64_t n, const float* w, const float* b,
                      int32_t time_dim, float* out, void* stream);

/* A + N + C gather stage of combining_pe_raw_feat (models/LSTEP.py:147-158, 177-211, 223-238), fused:
 * one temporal search per row serves the K-neighbour and the time_gap-neighbour lookups.
 *   out_edge  [B, D+F]  sum_j a[j] * cat[time_feat_j, edge_raw[eid_j]] over the K right-aligned slots
 *                       (= edge_agg applied BEFORE edge_mlp_1; the two are linear)
 *   out_node  [B, F]    softmax-mask mean of node_raw over the time_gap neighbours, /time_gap, + node_raw[id]
 *   out_pe    [B, P+D]  sum_j cat[pe[nbr_j], time_feat_j] over the K slots (padding slots read pe[0])
 *   out_self  [B, P]    pe[id]
 *   out_count [B]       c = interactions strictly earlier than times[b] (saved for the backward)
 * edge_agg_w float32 [K]; pe may be NULL when LSTEP_BRANCH_PE is not requested.
 * ld_* are the row strides (in floats) of the four outputs: 0 = dense (the widths above); a larger multiple of 4 pads each
 * row: the columns from the width up to the next multiple of 16 are written as 0 (the dense layers then see 16-aligned K: 176
 * instead of 172), anything beyond belongs to the caller -- lstep_tail_fwd's concatenated operands take out_node / out_self
 * as their first column block this way (ld_node = 624, ld_self = 352). */
int lstep_gather_aggregate_fwd(const lstep_csr_t* csr, const float* node_raw, const float* edge_raw, const float* pe,
                               int32_t feat_dim, int32_t pe_dim, const float* time_w, const float* time_b,
                               int32_t time_dim, const float* edge_agg_w, const int64_t* node_ids, const double* times,
                               int64_t batch, int32_t num_neighbors, int32_t time_gap, uint32_t branches,
                               float* out_edge, float* out_node, float* out_pe, float* out_self, int32_t ld_edge,
                               int32_t ld_node, int32_t ld_pe, int32_t ld_self, int32_t* out_count, void* stream);

/* Backward of the gather stage.
 *   grad_edge [B, D+F], grad_pe_agg [B, P+D], grad_self [B, P]  (any may be NULL = zero); ld_* = their row strides
 *   in floats (0 = dense), matching the forward's padded outputs
 *   out_slot_dot [B, K]: <grad_edge[b], cat[time_feat, edge_row] of slot j>; its column sum is d(edge_agg.weight).
 *   PE gradient, accumulated with float atomics (buffer must be zeroed by the caller):
 *     slot_of == NULL : grad_pe_rows is dense [num_rows, P]; row nbr_j += grad_pe_agg[b, :P] for valid slots,
 *                       row id_b += grad_self[b].  Padding slots (row 0) are NOT added here: the caller adds
 *                       sum_b (K - min(c_b, K)) * grad_pe_agg[b, :P] to row 0 (one tiny reduction, no hot row).
 *     slot_of != NULL : int32 [num_rows] map node id -> row of the compact gradient [U, P] or -1 (no gradient);
 *                       grad_pe_rows is [U, P].  This is the training fast path: only the FFT-filtered rows of
 *                       the spliced PE carry gradient (train_LSTEP_link_prediction.py:229-230).
 *     out_hits != NULL: (needs slot_of) no atomics at all: int32 [B, K] receives slot_of[nbr_j] per slot (-1 for padding
 *                       slots / rows without gradient); the caller groups the hits by spliced row and reduces
 *                       grad_pe_agg rows with lstep_segment_rows_sum (contention-free on hub nodes, deterministic). */
int lstep_gather_aggregate_bwd(const lstep_csr_t* csr, const float* edge_raw, int32_t feat_dim, int32_t pe_dim,
                               const float* time_w, const float* time_b, int32_t time_dim, const int64_t* node_ids,
                               const double* times, const int32_t* count, int64_t batch, int32_t num_neighbors,
                               const float* grad_edge, const float* grad_pe_agg, const float* grad_self,
                               int32_t ld_edge, int32_t ld_pe, int32_t ld_self, const int32_t* slot_of,
                               float* out_slot_dot, float* grad_pe_rows, int32_t* out_hits, void* stream);

/* N for hub nodes (round 5) -- models/LSTEP.py:177-211 for a node that occurs many times in ONE batch (power-law graphs: one node collects a
 * fifth of a batch's endpoints).  The windows of consecutive occurrences overlap in all but a few slots, so the rows of a hub are served by
 * prefix differences over the union of their windows instead of time_gap row reads each (csrc/hub.hip):
 *   lstep_hub_worklist    from the grouping of the batch rows cat[src, dst] by node (lstep_group_by_key: seg / order int32 [n2]): rows of
 *                         nodes with >= min_occ (>= 2) occurrences are marked in served (uint8 [num_rows], zeroed here) and cut into work
 *                         items of <= 32 occurrences (work int32 [capacity, 2] = first sorted position, occurrences; *nwork their number);
 *                         seg_start int32 [n2 + 1] is scratch; capacity >= lstep_hub_capacity(n2, min_occ) cannot overflow.
 *   lstep_gather_aggregate_fwd_skip   lstep_gather_aggregate_fwd whose node channel skips the rows marked in skip_node (their out_node rows
 *                         are not written); needs the long-row form of the kernel (csr->max_degree and time_gap > 256, no weighted_sum).
 *   lstep_hub_node_sums   one workgroup per work item writes the out_node rows of its occurrences (feature width <= 176): same value as the
 *                         gather kernel's up to the summation order (a difference of two fixed-order prefixes). */
int64_t lstep_hub_capacity(int64_t n2, int32_t min_occ);
int lstep_hub_worklist(const int32_t* seg, const int32_t* order, int64_t n2, int32_t min_occ, int32_t* seg_start, uint8_t* served,
                       int64_t num_rows, int32_t* work, int32_t* nwork, int64_t capacity, void* stream);
int lstep_gather_aggregate_fwd_skip(const lstep_csr_t* csr, const float* node_raw, const float* edge_raw, const float* pe,
                                    int32_t feat_dim, int32_t pe_dim, const float* time_w, const float* time_b,
                                    int32_t time_dim, const float* edge_agg_w, const int64_t* node_ids, const double* times,
                                    int64_t batch, int32_t num_neighbors, int32_t time_gap, uint32_t branches,
                                    float* out_edge, float* out_node, float* out_pe, float* out_self, int32_t ld_edge,
                                    int32_t ld_node, int32_t ld_pe, int32_t ld_self, int32_t* out_count,
                                    const uint8_t* skip_node, void* stream);
int lstep_hub_node_sums(const lstep_csr_t* csr, const float* node_raw, int32_t feat_dim, const int64_t* node_ids, const double* times,
                        int32_t time_gap, const int32_t* order, const int32_t* work, const int32_t* nwork, int64_t capacity, float* out_node,
                        int32_t ld_node, void* stream);

/* The gather stage on EXPLICIT neighbourhoods, for the RNG-defined sampling strategies ('uniform', 'time_interval_aware':
 * utils/utils.py:175-198), whose draws are defined by numpy's RandomState call order and therefore made on the host
 * (lstep_amd.sampler.NeighborSampler._sample_random_host).  The reference draws three independent neighbourhoods per
 * combining_pe_raw_feat call (models/LSTEP.py:147 K slots, :177 time_gap slots, :223 K slots again), so each call serves ONE branch:
 *   LSTEP_BRANCH_EDGE_NODE  nbr / eid / nt [batch, K] for the edge channel, nbr_gap (and nt_gap with LSTEP_WEIGHTED_SUM) [batch, time_gap]
 *   LSTEP_BRANCH_PE         nbr / nt [batch, K]
 * lists exactly as get_historical_neighbors returned them (int64 ids, float32 times; padding slots are id 0 and gather row 0 like any
 * id).  num_rows = rows of the node / PE tables (bound of the self-row lookups).  Outputs as lstep_gather_aggregate_fwd.
 * lstep_gather_explicit_bwd: as lstep_gather_aggregate_bwd on the same lists, one channel per call. */
int lstep_gather_explicit_fwd(const float* node_raw, const float* edge_raw, const float* pe, int32_t feat_dim, int32_t pe_dim,
                              const float* time_w, const float* time_b, int32_t time_dim, const float* edge_agg_w, const int64_t* node_ids,
                              const double* times, int64_t batch, int32_t num_neighbors, int32_t time_gap, uint32_t branches,
                              const int64_t* nbr, const int64_t* eid, const float* nt, const int64_t* nbr_gap, const float* nt_gap,
                              int64_t num_rows, float* out_edge, float* out_node, float* out_pe, float* out_self, int32_t ld_edge,
                              int32_t ld_node, int32_t ld_pe, int32_t ld_self, void* stream);
int lstep_gather_explicit_bwd(const float* edge_raw, int32_t feat_dim, int32_t pe_dim, const float* time_w, const float* time_b,
                              int32_t time_dim, const int64_t* node_ids, const double* times, int64_t batch, int32_t num_neighbors,
                              const int64_t* nbr, const int64_t* eid, const float* nt, int64_t num_rows, const float* grad_edge,
                              const float* grad_pe_agg, const float* grad_self, int32_t ld_edge, int32_t ld_pe, int32_t ld_self,
                              const int32_t* slot_of, float* out_slot_dot, float* grad_pe_rows, int32_t* out_hits, void* stream);

/* F -- the linear core of fourier_transform_pe (models/LSTEP.py:104-137).  fft -> mask -> filter -> mask ->
 * ifft -> mask -> real part -> fft_agg is linear in the history, so for fixed weights it is a [T, P] real
 * coefficient table coef (built by the host layer from fft_filter / fft_agg / batch_idx):
 *   out[u, p] = sum_{s < t_len} coef[s, p] * hist[node_ids[u], s, p].
 * hist is addressed as base + node * node_stride + ((s + time_rot) % time_slots) * time_stride (element
 * strides), so both the reference's [N+1, t, P] tensor and a [T, N+1, P] device ring are accepted. */
int lstep_history_filter_fwd(const float* hist, int64_t node_stride, int64_t time_stride, int32_t time_slots,
                             int32_t time_rot, int32_t t_len, int32_t pe_dim, const int64_t* node_ids, int64_t num_ids,
                             const float* coef, float* out, void* stream);

/* d(coef)[s, p] = sum_u grad_out[u, p] * hist[node_ids[u], s, p], written as per-chunk partial sums
 * out_partial [num_chunks, t_len, P] with num_chunks = lstep_history_filter_bwd_chunks(num_ids); the caller
 * reduces over dim 0 (deterministic, no atomics). */
int64_t lstep_history_filter_bwd_chunks(int64_t num_ids);
int lstep_history_filter_bwd(const float* hist, int64_t node_stride, int64_t time_stride, int32_t time_slots,
                             int32_t time_rot, int32_t t_len, int32_t pe_dim, const int64_t* node_ids, int64_t num_ids,
                             const float* grad_out, float* out_partial, void* stream);

/* F over a history of table CLONES (train_link_prediction.py:229 clones the newest snapshot, :301 appends it): a node's row in
 * snapshot s equals its row in snapshot s-1 unless the batch in between wrote it.  A device ring records that in a change mask,
 * uint32 [num_rows, mask_words] over its PHYSICAL slots (bit ph of row n: "slot ph of node n differs from slot ph-1"; at most 128 slots),
 * and the *_runs_* kernels read one row per run of equal rows -- same sums as lstep_history_filter_fwd / _bwd on the same data,
 * different order of summation.  The mask is maintained with
 *   lstep_history_slot_bits  set (value 1) or clear (0) the bit of `slot` in every row: start of a new snapshot.  Row 0 (the padding row,
 *                            zeroed and rewritten by every update_pe, models/LSTEP.py:317) is always marked as changed.
 *   lstep_history_mark       set the bit of `slot` for the listed node ids; world > 1: only ids with id % world == rank, at row id / world
 *                            (owner-sharded rings); ids outside [0, num_rows) are ignored.
 * A bit that is set for an unchanged row costs one extra row read, a bit that is missing for a changed row gives wrong results. */
int lstep_history_slot_bits(uint32_t* mask, int32_t mask_words, int64_t num_rows, int32_t slot, int32_t value, const lstep_ring_ref_t* ring,
                            void* stream);
int lstep_history_mark(uint32_t* mask, int32_t mask_words, int64_t num_rows, int32_t slot, const int64_t* ids, int64_t num_ids,
                       int32_t world, int32_t rank, const lstep_ring_ref_t* ring, void* stream);
/* workspace: lstep_history_filter_runs_workspace(t_len, pe_dim) bytes, 16-byte aligned (float64 prefix sums of coef) */
int64_t lstep_history_filter_runs_workspace(int32_t t_len, int32_t pe_dim);
/* oldest (optional, [num_rows, node_stride]): the window's oldest snapshot.  With it the slots need not be full clones: a slot only has to hold
 * the rows its batch wrote (the marked ones), the kernels read nothing else -- the ring then appends lstep_copy_rows of the written rows per
 * batch instead of cloning the table, and moves `oldest` on with lstep_history_advance_oldest when the window slides.  A row whose bit
 * of the window's first slot is set is read from that slot, not from `oldest`: `oldest` may therefore still be one slide behind (the
 * rows it has yet to take over are exactly those), which lets the ring move it on concurrently with the next forward pass. */
int lstep_history_filter_runs_fwd(const float* hist, int64_t node_stride, int64_t time_stride, int32_t time_slots, int32_t time_rot,
                                  int32_t t_len, int32_t pe_dim, const uint32_t* mask, int32_t mask_words, const float* oldest,
                                  const int64_t* node_ids, int64_t num_ids, const float* coef, void* workspace, float* out,
                                  float* table_out, int32_t* slot_of, const int32_t* num_live, const lstep_ring_ref_t* ring, void* stream);
/* num_live (optional, device): node_ids is a capacity-sized list and only its first min(*num_live, num_ids) entries are batch nodes
 * (the batch-node count of train_LSTEP_link_prediction.py:221-222 stays on the device: no host round trip before the FFT splice). */
/* (table_out, slot_of: optional.  table_out [num_rows, node_stride] also receives out[u] at row node_ids[u] -- the splice
 * positional_encoding[:, -1][batch nodes] = filtered rows of train_LSTEP_link_prediction.py:230 -- and slot_of[node_ids[u]] = u.) */
/* out_partial [lstep_history_filter_bwd_chunks(num_ids), t_len, P] holds per-chunk DIFFERENCE sums; the caller adds them over dim 0 and
 * passes the [t_len, P] result to lstep_history_filter_runs_finish, which turns it into d(coef) (running sums in float64). */
/* rows of out_partial for lstep_history_filter_runs_bwd: [lstep_history_filter_runs_bwd_chunks(num_ids, t_len), t_len, P] */
int64_t lstep_history_filter_runs_bwd_chunks(int64_t num_ids, int32_t t_len);
int lstep_history_filter_runs_bwd(const float* hist, int64_t node_stride, int64_t time_stride, int32_t time_slots, int32_t time_rot,
                                  int32_t t_len, int32_t pe_dim, const uint32_t* mask, int32_t mask_words, const float* oldest,
                                  const int64_t* node_ids, int64_t num_ids, const float* grad_out, float* out_partial,
                                  const lstep_ring_ref_t* ring, void* stream);
/* dst[r, :width] = src[r, :width] for r in ids (row stride ld floats; rows outside [0, num_rows) ignored): the rows a batch wrote, copied
 * from the current PE table into the batch's history slot (train_link_prediction.py:301 appends the whole table). */
int lstep_copy_rows(float* dst, const float* src, int32_t width, int64_t ld, const int64_t* ids, int64_t num_ids, int64_t num_rows,
                    const lstep_ring_ref_t* ring, void* stream);
/* oldest[r] = slot_rows[r] for every row r whose bit of `slot` is set in the change mask: the window's oldest snapshot moves to that slot
 * (the trim of train_link_prediction.py:224-225). */
int lstep_history_advance_oldest(float* oldest, const float* slot_rows, int32_t width, int64_t ld, const uint32_t* mask, int32_t mask_words,
                                 int32_t slot, int64_t num_rows, const lstep_ring_ref_t* ring, void* stream);
int lstep_history_filter_runs_finish(const float* partial_sum, int32_t t_len, int32_t pe_dim, float* dcoef, void* stream);

/* Segmented row sums.  Entry e = 0..num_entries-1 belongs to output row ent_seg[e]; entries of one segment must be
 * adjacent (ent_seg grouped, e.g. sorted); for every segment s that occurs
 *   out[s, :W]    = sum_e table[ent_row[e], :W]                       (table row stride ld_table floats)
 *   out[s, W:W+D] = sum_e cos(ent_dt[e] * time_w + time_b)            (time_dim D may be 0: no time part, ent_dt unused)
 * `out` (row stride ld_out) MUST be zero-initialised by the caller (accumulate = 0): rows that own no entry stay zero.  The entry list is
 * cut into chunks of 16 entries (lists of up to 65 536 entries) or 64 entries (longer lists), one wave each.  A segment of at most 64
 * entries belongs WHOLLY to the wave of the chunk it starts in (that wave reads on past its chunk, the next one skips those entries): it is
 * summed in entry order and written with plain stores.  Only segments of more than 64 entries (hub nodes) are cut at the chunk boundaries:
 *   with `workspace` (lstep_segment_rows_sum_workspace bytes, 16-byte aligned): every chunk parks its partial sum there and further passes
 *     join a segment's partials in a fixed-shape tree (aligned groups of 16 whole-chunk partials first, then the groups and the remaining
 *     chunks in chunk order) -- the result is a function of the inputs alone (replicas of a table that run the same update on different
 *     GPUs stay bit-identical, two runs of one process too), and a hub with thousands of chunks is not a serial chain;
 *   with workspace = NULL: float atomics into the row, in arrival order (exact up to the order of three or more partial sums).
 * accumulate = 1: the sums are ADDED to what `out` already holds (a second reduction into the same rows, e.g. neighbour + self
 * gradients).  accumulate = 2: `out` is UNINITIALISED memory and (width + time_dim) a multiple of 4 -- every row that owns entries is
 * written whole (without a workspace a pre-pass zeroes the straddling segments' rows first), rows without entries stay undefined
 * (update_pe: every row owns entries; saves a 356 MB memset).  Uses:
 *   update_pe U1/U2 (models/LSTEP.py:282-290, 319-322) with table = pe, D = time dim: replaces both dense [N+1, P+D]
 *   torch_scatter targets;  gather backward: table = grad of the PE aggregate, D = 0, segment = spliced PE row. */
int64_t lstep_segment_rows_sum_workspace(int64_t num_entries, int32_t width, int32_t time_dim);
int lstep_segment_rows_sum(const float* table, int32_t width, int32_t ld_table, const float* time_w, const float* time_b,
                           int32_t time_dim, const int32_t* ent_seg, const int32_t* ent_row, const float* ent_dt,
                           int64_t num_entries, float* out, int32_t ld_out, int32_t accumulate, const int32_t* num_live, void* workspace,
                           int64_t workspace_bytes, void* stream);
/* num_live (optional, device): only the first min(*num_live, num_entries) entries count (a capacity-sized entry list whose length
 * lstep_group_by_key left on the device). */

/* In-place row write pe[ids[i], :] = rows[i, :] (models/LSTEP.py:303,339). ids must be unique. */
int lstep_scatter_rows(float* table, int32_t width, const int64_t* ids, int64_t num_ids, const float* rows,
                       void* stream);

/* Fused residual update + in-place write: table[ids[i], :] += tanh(z[i, :]) (models/LSTEP.py:299-303 with
 * z = self_update_pe(own) + pe_mlp_2(...); :335-339 with z = pe_mlp_2(...)). ids must be unique; ld_z = row stride of
 * z in floats (0 = width). */
int lstep_residual_tanh_rows(float* table, int32_t width, const int64_t* ids, int64_t num_ids, const float* z, int32_t ld_z,
                             void* stream);

/* Group int32 keys on the device (stable radix sort on the low key_bits bits + head flags + scan); the plumbing behind
 * the batch-node set (train_LSTEP_link_prediction.py:221-222), the update_pe segments (models/LSTEP.py:282-290, 319-324:
 * what torch_scatter + torch.unique do in the reference) and the gradient segments of the gather backward.
 *   sorted_keys[n], order[n] (original index of every sorted entry; equal keys keep their input order),
 *   seg[n] (rank of the entry's key among the distinct keys), uniq[<= n] (distinct keys ascending),
 *   summary[3] (device) = { number of distinct keys, number of entries with key < limit, number of distinct keys < limit }
 * Callers give entries they want dropped a key >= limit, so the wanted entries / segments are the leading ones.
 * workspace: lstep_group_by_key_workspace(n, key_bits) bytes of device scratch (nothing is allocated inside). */
int64_t lstep_group_by_key_workspace(int64_t n, int32_t key_bits);
int lstep_group_by_key(const int32_t* keys, int64_t n, int32_t key_bits, int32_t limit, void* workspace, int64_t workspace_bytes,
                       int32_t* sorted_keys, int32_t* order, int32_t* seg, int32_t* uniq, int32_t* summary, void* stream);

/* O -- weight / bias gradient of one dense layer of the tail (torch.nn.Linear y = x W^T + b as used by models/LSTEP.py:56-72,
 * models/modules.py:52-68), on the fp32 matrix cores:
 *   dw[n, k] = sum_r dy[r, n] * x[r, k]   (row stride ld_dw)        db[n] = sum_r dy[r, n]   (db may be NULL)
 * dy [m, n] row stride ldy, x [m, k] row stride ldx, fp32.  Exact fp32 products and sums (v_mfma_f32_16x16x4_f32), summed per
 * row slice and then over the slices in a fixed order: deterministic.  workspace: lstep_linear_wgrad_workspace(m, n, k) bytes of
 * 16-byte aligned device scratch. */
int64_t lstep_linear_wgrad_workspace(int64_t m, int32_t n, int32_t k);
int lstep_linear_wgrad(const float* dy, int32_t ldy, const float* x, int32_t ldx, int64_t m, int32_t n, int32_t k, float* dw,
                       int32_t ld_dw, float* db, void* workspace, int64_t workspace_bytes, void* stream);

/* The same for several layers at once: ONE partial launch and ONE reduction launch for up to 8 products (the four of the dense tail, the
 * two of the link predictor: models/LSTEP.py:56-72, models/modules.py:52-68), instead of a dependent launch pair per product.  Results are
 * those of lstep_linear_wgrad product by product (same tiling, same summation order); products whose operands do not allow the 16-byte
 * loads of the batched kernel, or that are too large for it (more than 16 384 rows), keep a partial launch of their own inside the call;
 * their reductions leave together as ONE launch at the end of the call (round 5: every product has its own slice of the workspace, and
 * nothing needs a single product's result earlier).  workspace: lstep_linear_wgrad_batch_workspace(count, descs) bytes, 16-byte aligned. */
typedef struct lstep_wgrad_desc {
    const float* dy;   /* [m, n], row stride ldy */
    const float* x;    /* [m, k], row stride ldx */
    float* dw;         /* [n, k], row stride ld_dw */
    float* db;         /* [n] or NULL */
    int64_t m;
    int32_t n, k, ldy, ldx, ld_dw, reserved;
} lstep_wgrad_desc_t;
int64_t lstep_linear_wgrad_batch_workspace(int32_t count, const lstep_wgrad_desc_t* descs);
int lstep_linear_wgrad_batch(int32_t count, const lstep_wgrad_desc_t* descs, void* workspace, int64_t workspace_bytes, void* stream);

/* A / N / C / O -- every dense layer after the gather stage in one launch (fp32 matrix cores), for the model's default widths
 * (feature / PE dim 172, time dim 100: 16-aligned channels 272 / 176 / 176).  Inputs: x_edge [m, ld_edge], x_pe [m, ld_pe] (gather
 * outputs), cat1 [m, 624] = [x_node | . | .] and cat2 [m, 352] = [own | .] with the first 176 columns filled by the gather stage
 * (lstep_gather_aggregate_fwd writes them there through its ld_node / ld_self arguments).  The 16-padded, composed weights
 *   w1 [272, 272], b1 [272]   edge_mlp_1 with edge_agg folded in                (models/LSTEP.py:161-166)
 *   wn1 [176, 272], bn1 [176] pe_neighbor_mlp_1                                 (:240-242)
 *   wq [176, 352], bq [176]   [self_update_neighbor_pe | pe_neighbor_mlp_2]     (:243-245)
 *   wall [176, 624], ball     out_node_emb . node_mlp . edge_mlp_2 pre-multiplied (:170,219,264)
 * are built by the host layer (DESIGN.md "dense tails").  Outputs: h1 = relu(w1 x_edge + b1) -> cat1[:, 176:448],
 * p1 = relu(wn1 x_pe + bn1) -> cat2[:, 176:352], q = own + tanh(wq [own; p1] + bq) -> cat1[:, 448:624],
 * out [m, 176] = wall [x_node; h1; q] + ball (columns >= 172 are padding and come out 0). */
int lstep_tail_fwd(const float* x_edge, int32_t ld_edge, const float* x_pe, int32_t ld_pe, float* cat1, float* cat2, float* out,
                   const float* w1, const float* b1, const float* wn1, const float* bn1, const float* wq, const float* bq,
                   const float* wall, const float* ball, int64_t m, void* stream);

/* Backward of lstep_tail_fwd with respect to its activations.  grad_out [m, 176]; cat1 / cat2 as lstep_tail_fwd left them; the
 * weights TRANSPOSED ([in, out] row-major): w1t [272, 272], wn1t [272, 176], wqt [352, 176], wallt [624, 176].  Writes
 *   d_xedge [m, 272], d_xpe [m, 272], d_own [m, ld_down] (first 176 columns)         -- inputs of lstep_gather_aggregate_bwd
 *   d_h1 [m, 272], d_p1 [m, 176], d_z [m, 176]  (pre-activation gradients)           -- dy operands of lstep_linear_wgrad:
 *   dW1 = d_h1^T x_edge, dWn1 = d_p1^T x_pe, dWq = d_z^T cat2, dWall = grad_out^T cat1. */
int lstep_tail_bwd(const float* grad_out, const float* cat1, const float* cat2, const float* w1t, const float* wn1t, const float* wqt,
                   const float* wallt, float* d_xedge, float* d_xpe, float* d_own, int32_t ld_down, float* d_h1, float* d_p1, float* d_z,
                   int64_t m, void* stream);

/* Stable sort of the LIVE entries of an int32 key array: entries with a negative key are dropped first (one select pass), the rest
 * are sorted on their low key_bits bits.  sorted_keys[num_live], order[num_live] (original index of every sorted entry; equal
 * keys keep their input order); *num_live is written on the HOST (the call waits for the select pass on `stream`).  Used for the
 * gradient hits of lstep_gather_aggregate_bwd (out_hits), where ~95 % of the entries are -1.
 * workspace: lstep_sort_live_workspace(n, key_bits) bytes of device scratch. */
int64_t lstep_sort_live_workspace(int64_t n, int32_t key_bits);
/* The same without the host round trip.  The sort runs on a FIXED number of items, `capacity` (the caller's estimate from earlier batches):
 * the live keys padded with `sentinel`, a value above every live key that still fits key_bits (it sorts last).  Outputs: sorted_keys /
 * order [capacity]; live_index [n]: indices of the live entries, ascending; *count (device int32): their number.  Consumers read *count on
 * the device (lstep_segment_rows_sum_live).  If *count > capacity, the live entries live_index[capacity .. *count) are NOT in the sorted
 * output: add them with lstep_scatter_add_overflow (float atomics; exact, only the order of summation differs). */
int64_t lstep_sort_live_bounded_workspace(int64_t n, int64_t capacity, int32_t key_bits);
int lstep_sort_live_bounded(const int32_t* keys, int64_t n, int32_t key_bits, int32_t sentinel, int64_t capacity, void* workspace,
                            int64_t workspace_bytes, int32_t* sorted_keys, int32_t* order, int32_t* live_index, int32_t* count, void* stream);
/* lstep_segment_rows_sum (no time part, accumulate 0 / 1) over such a padded list: only the first min(*num_live, num_entries) entries count. */
int lstep_segment_rows_sum_live(const float* table, int32_t width, int32_t ld_table, const int32_t* ent_seg, const int32_t* ent_row,
                                int32_t row_div, int64_t num_entries, const int32_t* num_live, float* out, int32_t ld_out, int32_t accumulate,
                                void* workspace, int64_t workspace_bytes, void* stream);      /* table row of entry e = ent_row[e] / row_div */
/* out[keys[e], :width] += table[e / div, :width] for e = live_index[i], i in [capacity, *count). */
int lstep_scatter_add_overflow(float* out, int32_t width, int32_t ld_out, const int32_t* keys, const int32_t* live_index, const int32_t* count,
                               int64_t capacity, int32_t div, const float* table, int32_t ld_table, void* stream);
int lstep_sort_live(const int32_t* keys, int64_t n, int32_t key_bits, void* workspace, int64_t workspace_bytes, int32_t* sorted_keys,
                    int32_t* order, int64_t* num_live, void* stream);

/* P -- the scalar end of a training iteration and its gradient (train_LSTEP_link_prediction.py:257-275):
 *   lp_loss = BCE(sigmoid(logits).clamp(0, 1), [1]*n + [0]*n),  pe_loss = MSE(e_src, e_dst) - neg_weight * MSE(e_src, e_neg),
 *   loss = (1 - pe_weight) * lp_loss + pe_weight * pe_loss,
 * e_x = rows[slot_of[x]] if slot_of[x] >= 0 (the spliced, differentiable row of batch node x) else table[x].
 * logits [2 n] (positive | negative edges), ids int64 [3 n] (src | dst | negative dst).  Outputs: predicts [2 n] (the clamped
 * probabilities), losses [3] = {lp_loss, pe_loss, loss}, and the gradient of `loss`: d_logits [2 n], and for the positional-encoding
 * rows one gradient row PER OCCURRENCE, g_rows [3 n, pe_dim] (16-byte aligned): row i = d loss / d e_src of edge i, row n + i = d e_dst,
 * row 2 n + i = d e_neg, with neg_slot [n] = slot_of[negative endpoint of edge i].  The gradient of spliced row u is the sum of the rows
 * whose endpoint is u: the caller reduces them with lstep_segment_rows_sum over its grouping of cat[src, dst] by batch node (fixed
 * summation order) and lstep_scatter_add_rows for the negatives that happen to be batch nodes -- no atomics on hub rows.
 * workspace: lstep_link_loss_workspace(n) bytes. */
int64_t lstep_link_loss_workspace(int64_t n);
int lstep_link_loss(const float* logits, const int64_t* ids, int64_t n, const float* table, const float* rows, const int32_t* slot_of,
                    int32_t pe_dim, float pe_weight, float neg_weight, float* predicts, float* d_logits, float* g_rows, int32_t* neg_slot,
                    float* losses, void* workspace, int64_t workspace_bytes, void* stream);

/* Link predictor (models/modules.py:42-68, MergeLayer) over the padded embeddings emb [rows, 176] of one batch, with no
 * concatenation materialised: for edge e < n the positive pair is (emb[pos_first + e], emb[pos_second + e]), the negative pair
 * (emb[neg_first + e], emb[neg_second + e]) -- training: blocks src | dst | negative dst = offsets (0, n, 0, 2n); evaluation:
 * src | dst | negative src | negative dst = (0, n, 2n, 3n).  w [176, 352] = fc1.weight re-laid as [first half | second half],
 * each half zero-padded from 172 to 176 (rows too); b1, w2 [176] zero-padded; b2 [1].
 * Outputs: h [2 n, 176] = relu(fc1) for the positive then the negative pairs, logits [2 n] = fc2(h). */
int lstep_head_fwd(const float* emb, int64_t n, int64_t pos_first, int64_t pos_second, int64_t neg_first, int64_t neg_second,
                   const float* w, const float* b1, const float* w2, const float* b2, float* h, float* logits, void* stream);

/* Backward of lstep_head_fwd in the training layout (0, n, 0, 2n).  wt [352, 176] = w transposed.  Outputs: d_emb [3 n, 176] (all
 * three row blocks: directly the grad_out of lstep_tail_bwd), d_h [2 n, 176] (pre-activation gradient) and d_hsum [n, 176] =
 * d_h[pos] + d_h[neg], the dy operands of the weight gradient: dw[:, :176] = d_hsum^T emb[0:n], dw[:, 176:] = d_h^T emb[n:3n],
 * db1 = column sums of d_h (lstep_linear_wgrad); dw2_partial [ceil(n / 16), 176]: per-slab partial sums of d_logit * h, whose column
 * sums are the gradient of fc2.weight (columns 0..171) and, in column 172, the sum of d_logit = the gradient of fc2.bias. */
int lstep_head_bwd(const float* d_logits, const float* h, int64_t n, const float* wt, const float* w2, float* d_emb, float* d_h,
                   float* d_hsum, float* dw2_partial, void* stream);

/* U1 / U2 -- the dense end of update_pe in one launch (models/LSTEP.py:292-303, 327-339), default widths (pe_dim <= 176, time dim
 * such that pe_dim + time_dim = 272):  z = w2 relu(w1 agg[r] + b1) + b2 (+ ws table[ids[r]] + bs when ws != NULL: phase 1;
 * phase 2 passes NULL, the reference discards that term), then IN PLACE table[ids[r], :] += tanh(z).
 * agg [>= n, ld_agg] = the segment sums of lstep_segment_rows_sum; ids int64 [n], unique; weights zero-padded to
 * w1 [176, 272], w2 / ws [176, 176], biases [176].  mirror (optional): a second [rows, pe_dim] table that receives the new rows as well
 * -- the batch's history slot, so that appending the snapshot (train_link_prediction.py:301) costs no separate copy. */
int lstep_update_rows(const float* agg, int32_t ld_agg, const int64_t* ids, int64_t n, const float* w1, const float* b1, const float* w2,
                      const float* b2, const float* ws, const float* bs, float* table, float* mirror, int32_t pe_dim, const int32_t* num_live,
                      const lstep_ring_ref_t* ring, int32_t mirror_world, int32_t mirror_rank, void* stream);
/* num_live (optional, device): only the first min(*num_live, n) rows are updated; n is then the capacity the launch covers.
 * mirror_world > 1: the mirror is a slot of an OWNER-SHARDED history ring (one process per GPU, node id owned by rank id % world): the new
 * row of node id is mirrored to row id / world, and only if id % world == mirror_rank.  (1, 0): the mirror has the table's shape. */

/* U2 with the first layer pre-multiplied into the messages.  pe_mlp_1 (models/LSTEP.py:327) is linear in the message sum and every
 * message's PE row is one of the U batch-node rows (:319-322), so W1 [sum pe ; sum time] = sum (W1[:, :pe_dim] pe) + W1[:, pe_dim:] sum time:
 * the caller forms y[u] = W1[:, :pe_dim] pe[batch node u] once ([U, 176], columns >= 172 zero), runs lstep_segment_rows_sum over y (entry rows
 * = positions in the batch-node list: lstep_update_entries_p2_dev with bn == NULL) and passes agg [>= n, ld_agg >= 176 + time_dim] =
 * [sum y (176) | sum time features (time_dim)] with w1b [176, 112] = W1[:, pe_dim:] zero-padded.  36 % fewer multiply-adds per touched
 * row than lstep_update_rows; the sums are re-associated (differences at the 1e-7 level). */
int lstep_update_rows_pre(const float* agg, int32_t ld_agg, const int64_t* ids, int64_t n, const float* w1b, const float* b1, const float* w2,
                          const float* b2, float* table, float* mirror, int32_t pe_dim, int32_t time_dim, const int32_t* num_live,
                          const lstep_ring_ref_t* ring, int32_t mirror_world, int32_t mirror_rank, void* stream);

/* Device-resident counts for the engine's update_pe (no host synchronisation anywhere in models/LSTEP.py:268-340):
 *   lstep_widen_ids             out[i] = i < *count ? ids32[i] : 0 -- lstep_group_by_key's distinct keys as an int64 id list of fixed
 *                               capacity whose dead tail is the padding node 0 (no history, no neighbours).
 *   lstep_update_entries_p2_dev lstep_update_entries_p2 with n_real / nseg read from lstep_group_by_key's device summary and the
 *                               row-0 decision (models/LSTEP.py:324: 0 is among the unique neighbour ids iff a slot of a live row is
 *                               padding) taken on the device: segment 0 is always reserved for row 0, touched = [0, uniq..., 0 ...],
 *                               counts_out = {nseg, row-0 flag}.  bn == NULL: ent_row = the row's position in the id list. */
int lstep_widen_ids(const int32_t* ids32, int64_t capacity, const int32_t* count, int64_t* out, void* stream);
int lstep_update_entries_p2_dev(const int32_t* order, const int32_t* seg, const int32_t* summary, const int32_t* live_rows, int64_t capacity,
                                int64_t touched_capacity, const int64_t* bn, const float* nt, const float* now32, int32_t num_neighbors,
                                const int32_t* uniq, int32_t* ent_row, float* ent_dt, int32_t* ent_seg, int64_t* touched, int32_t* counts_out,
                                void* stream);

/* out[slot[i], :width] += rows[i, :width] for every i with slot[i] >= 0 (float atomics).  The stragglers of the spliced-row gradient:
 * negative-sample rows whose own node happens to be a batch node (a few hundred per batch; everything else goes through the sorted,
 * deterministic lstep_segment_rows_sum path). */
int lstep_scatter_add_rows(float* out, int32_t width, int32_t ld_out, const int32_t* slot, int64_t n, const float* rows, int32_t ld_rows,
                           void* stream);

/* The gradient of the spliced PE rows of a SMALL batch in one launch, no sort and no atomics (backward of the PE channel of
 * models/LSTEP.py:222-249 and of the PE loss rows, train_LSTEP_link_prediction.py:257-275, with respect to the filtered rows of train:230):
 *   out[u, :width] = sum over the slots (b, j) with hits[b * K + j] == u of g_hit[b, :width]      (ascending b, j)
 *                  + sum over the rows b with slot_of[ids[b]] == u of g_self[b, :width]             (ascending b)
 * for u < num_rows; hits int32 [num_hits] (lstep_gather_aggregate_bwd's out_hits, -1 = no spliced row), slot_of / ids as everywhere.
 * Either part may be absent (g_hit or g_self NULL).  Every output row is written whole.  Cost O(num_rows * (num_hits + num_self) / 64)
 * scan steps: meant for the reference's own batch sizes (a few hundred batch nodes); the sorted path (lstep_sort_live_bounded +
 * lstep_segment_rows_sum_live) serves everything else. */
int lstep_spliced_grad_small(const int32_t* hits, int64_t num_hits, int32_t num_neighbors, const float* g_hit, int32_t ld_hit,
                             const int32_t* slot_of, const int64_t* ids, int64_t num_self, const float* g_self, int32_t ld_self,
                             int32_t width, float* out, int32_t ld_out, int64_t num_rows, void* stream);

/* U2, row 0 -- what the PADDED slots of update_pe's sampled neighbourhoods scatter into row 0 (models/LSTEP.py:317-322):
 *   sum_r (number of zero entries of nbr[r, :]) * table[ids[r], :width]
 * as per-block partial sums partial [lstep_padding_rows_sum_blocks(n), width] (the caller adds them: fixed order, no atomics).
 * nbr int64 [n, num_neighbors] (lstep_sample_recent's output), ids int64 [n] the source rows (NULL: row r of the table itself). */
int64_t lstep_padding_rows_sum_blocks(int64_t n);
int lstep_padding_rows_sum(const int64_t* nbr, int32_t num_neighbors, const int64_t* ids, int64_t n, const float* table, int32_t width,
                           int32_t ld_table, float* partial, void* stream);

/* F, parameter side -- the real [T, P] coefficient table the history filter kernels consume (models/LSTEP.py:104-137 is linear in the
 * history): coef[s, p] = Re(1/T sum_f m[f] W[f, p] e^{-2 pi i f s / T} sum_t a[t] m[t] e^{+2 pi i f t / T}), computed in complex128.
 * filter_weight = fft_filter.weight as interleaved (re, im) float32 [T, P, 2]; agg_weight = fft_agg.weight float32 [T]; mask float64 [T]
 * (1 on the valid leading snapshots while the window is not full, else all ones: models/LSTEP.py:108-113); c_out float64 [T, 2]
 * receives the per-frequency factor c[f] the backward needs.  T <= 256. */
int lstep_fft_coef_fwd(const float* filter_weight, const float* agg_weight, const double* mask, int32_t t_len, int32_t pe_dim, float* coef,
                       double* c_out, void* stream);
/* Gradients of the table: grad_filter (re, im) float32 [T, P, 2] (PyTorch's convention dL/dRe + i dL/dIm), grad_agg float32 [T];
 * scratch float64 [T, 2]. */
int lstep_fft_coef_bwd(const float* grad_coef, const float* filter_weight, const double* c, const double* mask, int32_t t_len, int32_t pe_dim,
                       float* grad_filter, float* grad_agg, double* scratch, void* stream);

/* c[i, j] = alpha * sum_k a[i, k] b[k, j] + beta * c[i, j] for weight-sized fp32 matrices (a few hundred rows / columns), with
 * ELEMENT strides for every operand (a[i, k] at a + i * sa_i + k * sa_k, ...), so transposed and sliced views need no copy.
 * One wave per 16 x 16 output tile on the fp32 matrix cores.  Used for the weight composition of the dense tail (out_node_emb .
 * node_mlp . edge_mlp_2, models/LSTEP.py:170,219,264) and its backward. */
int lstep_small_gemm(const float* a, int64_t sa_i, int64_t sa_k, const float* b, int64_t sb_k, int64_t sb_j, float* c, int64_t sc_i,
                     int64_t sc_j, int32_t m, int32_t n, int32_t k, float alpha, float beta, void* stream);

/* The weight composition of the dense tail around those products (lstep_amd/model.py `_combined_tail`: out_node_emb . node_mlp . edge_mlp_2
 * pre-multiplied, edge_agg folded into edge_mlp_1's bias, models/LSTEP.py:161-170,219,240-247,264), everything that is not a matrix product in
 * one launch per direction.  dims = {F, D+F, P, P+D, Ce, Fn, Cp, Pp, K} (the widths and their 16-aligned paddings, K = num_neighbors).
 * pack: params = the 16 parameter tensors {edge_mlp_1.weight, .bias, edge_agg.weight, .bias, edge_mlp_2.weight, .bias, node_mlp.weight, .bias,
 *   out_node_emb.weight, .bias, self_update_neighbor_pe.weight, .bias, pe_neighbor_mlp_1.weight, .bias, pe_neighbor_mlp_2.weight, .bias}
 *   (contiguous fp32), M [F, D+F] = Wo_a Wn_b.  flat = {W1p [Ce,Ce], b1p [Ce], Wn1p [Pp,Cp], bn1p [Pp], Wq [Pp,2Pp], bq [Pp], Wall [Fn,Fn+Ce+Pp],
 *   const [Fn]} back to back, with Wall[:F,:F] and Wall[:F,Fn:Fn+D+F] already holding the products Wo_a Wn_a and M W2; flat_t = the transposes
 *   of the four matrices back to back; a_sum [1] = sum of edge_agg.weight.
 * unpack (the hand-derived backward): grads_in = gradients of the 8 blocks of flat; fwd = {edge_mlp_1.bias, edge_mlp_2.bias, node_mlp.bias,
 *   out_node_emb.weight, M, a_sum}; dM [F, D+F] holds dA2 W2^T on entry and dA2 W2^T + dc b2^T on return; grads_out = dense gradients
 *   {edge_mlp_1.weight, .bias, edge_agg.weight, .bias, edge_mlp_2.bias, node_mlp.bias, out_node_emb.weight (= [dc bn^T | d Wo_b]: the caller
 *   adds dA1 Wn_a^T + dM Wn_b^T to the first block), .bias, self_update_neighbor_pe.weight, .bias, pe_neighbor_mlp_1.weight, .bias,
 *   pe_neighbor_mlp_2.weight, .bias}. */
int lstep_tail_weights_pack(const float* const* params, const float* M, const int32_t* dims, float* flat, float* flat_t, float* a_sum, void* stream);
int lstep_tail_weights_unpack(const float* const* grads_in, const float* const* fwd, float* const* grads_out, float* dM, const int32_t* dims,
                              void* stream);

/* U1 / U2 -- the message lists of update_pe, one kernel each (models/LSTEP.py:277-290, 305-324).
 * _p1: order int32 [num_entries <= 2 batch] = positions of cat[src, dst] grouped by receiving endpoint (lstep_group_by_key);
 *      ent_row[e] = the other endpoint, ent_dt[e] = float(double(now32) - times[.]) with now32 a float32 DEVICE scalar (the reference
 *      rounds the current time to float32 first, LSTEP.py:277).
 * _keys_p2: int32 keys of the sampled neighbour slots nbr int64 [n]: the neighbour id, or `sentinel` for padding slots (id 0) and, when
 *      world > 1, for neighbours owned by another rank (id % world != rank).
 * _p2: for the n_real grouped live slots (order / seg from lstep_group_by_key on those keys): ent_row = bn[slot / K], ent_dt = now32 -
 *      nt[slot] (float32 - float32, LSTEP.py:314), ent_seg = seg + shift; touched int64 [shift + nseg] = [0 if shift] + uniq[:nseg]
 *      (shift = 1 when padding slots exist: row 0 is updated too and takes segment 0, LSTEP.py:317-324). */
/* What an engine iteration derives from its batch first, in one launch: ids3 = cat[src, dst, neg] (neg may be NULL: two blocks), t3 = the times
 * repeated per block (train_LSTEP_link_prediction.py:233-251), keys = int32 cat[src, dst] (the grouping keys of train:221-222) and
 * now32[0] = float32(max times) (models/LSTEP.py:277).  All device pointers; ids3 / t3 hold 3 * batch entries. */
int lstep_batch_prepare(const int64_t* src, const int64_t* dst, const int64_t* neg, const double* times, int64_t batch, int64_t* ids3, double* t3,
                        int32_t* keys, float* now32, void* stream);
/* row[:width] = the block partials of lstep_padding_rows_sum added in block order, row[width:row_width] = 0 (update_pe phase 2: the padding
 * row's aggregate, models/LSTEP.py:316-322). */
int lstep_padding_rows_finish(const float* partial, int64_t blocks, int32_t width, float* row, int32_t row_width, void* stream);
/* Owner-sharded PE table (one process per GPU, node id owned by rank id % world): the request lists of one gather.  lstep_pull_keys: for
 * the ids {nbr[0..n_nbr), ids[0..n_ids), 0} -- the sampled neighbour slots of a batch's gather rows (models/LSTEP.py:233), the rows themselves
 * (:232) and the padding row -- keys = owner * num_rows + id, or world * num_rows for ids this rank owns; group them with lstep_group_by_key
 * (limit = world * num_rows).  lstep_pull_blocks: from the grouped unique keys, count[p] = distinct ids wanted from owner p and
 * req[p, 0..capacity) = those ids, -1 beyond.  New design: the reference is single-process. */
int lstep_pull_keys(const int64_t* nbr, int64_t n_nbr, const int64_t* ids, int64_t n_ids, int32_t world, int32_t rank, int64_t num_rows, int32_t* keys,
                    void* stream);
int lstep_pull_blocks(const int32_t* uniq, const int32_t* summary, int32_t world, int64_t num_rows, int64_t capacity, int32_t* req, int32_t* count,
                      void* stream);
/* Owner-sharded execution with every size on the device (round 4: a rank's iteration is a fixed launch sequence with fixed-capacity
 * collectives and can be replayed as one HIP graph; the reference is single-process, SURVEY.md 8e -- these replace host-sized
 * argsort / bincount / index_copy_ chains around its train_LSTEP_link_prediction.py:221-230 splice).
 * lstep_owner_partition: ids int64 [capacity] = the batch's sorted unique endpoints (lstep_widen_ids: *num_live of them, dead tail = node 0)
 *   split by owner rank id % world into world blocks of block_slots slots: ids_by_owner int64 [world * block_slots] (block p = the ids owned
 *   by rank p in ascending order, unused slots = 0), pos_by_owner int32 (the entry's position in `ids`, unused = 0), counts int32 [world]
 *   (clamped to block_slots).  *overflow is OR-ed with 1 when some owner holds more than block_slots entries (the surplus is dropped: the
 *   caller's capacity was too small and the step must not be trusted -- it never resets the word).  workspace:
 *   lstep_owner_partition_workspace(capacity, world) bytes.  world <= 16.
 * lstep_scatter_owner_rows: rows [world * block_slots, ld_rows] (the all-gathered blocks, same layout) -> table[id, :width] for the
 *   counts[p] leading slots of every block; slot_of (optional int32 [num_rows]) receives the slot number p * block_slots + i of each id
 *   (the row of the compact gradient buffer, lstep_gather_aggregate_bwd).
 * lstep_rows_by_id: ids int32 [n] with holes (-1): direction 0 copies table rows into buf [n, width] (an owner serving a pull request),
 *   direction 1 writes buf rows into the table (the requester storing what it received). */
int64_t lstep_owner_partition_workspace(int64_t capacity, int32_t world);
int lstep_owner_partition(const int64_t* ids, int64_t capacity, const int32_t* num_live, int32_t world, int64_t block_slots, void* workspace,
                          int64_t workspace_bytes, int64_t* ids_by_owner, int32_t* pos_by_owner, int32_t* counts, int32_t* overflow, void* stream);
int lstep_scatter_owner_rows(const float* rows, int32_t ld_rows, const int64_t* ids_by_owner, const int32_t* counts, int32_t world,
                             int64_t block_slots, float* table, int32_t width, int32_t* slot_of, void* stream);
int lstep_rows_by_id(const int32_t* ids, int64_t n, float* table, int32_t width, float* buf, int32_t direction, void* stream);
/* HOST functions (plain C++, no kernel; all pointers are host memory): the RNG-defined sampling strategies 'uniform' /
 * 'time_interval_aware' of NeighborSampler.get_historical_neighbors (utils/utils.py:175-198).  The reference draws
 * `random_state.choice(a=cnt, size=K, p=p)` once per row, in row order, from numpy's legacy MT19937 RandomState, so the result is defined
 * by that generator's stream: lstep_sample_random_host replays the two paths of RandomState.choice bit for bit (p == NULL: randint's masked
 * rejection on 32-bit outputs, nothing consumed when cnt == 1; p: float64 cumsum / last, two outputs per uniform double,
 * searchsorted side='right') over the time-sorted CSR (indptr int64 [num_rows + 1], nbr / eid int64, ts float64) for rows
 * r = 0 .. m - 1 (node_ids[r], times[r]; cnt = entries strictly before times[r], utils/utils.py:140) and writes the K picked slots of every
 * row with history into out_* [m, K] (int64, int64, float32(ts)) in DRAW order; rows without history are left untouched and consume
 * nothing.  mt_key [624] / mt_pos are numpy's RandomState.get_state()[1:3], advanced in place.  p_values / p_offsets [m + 1]: row r's
 * float32 probabilities (what torch.softmax produced, utils/utils.py:182), p_offsets[r + 1] - p_offsets[r] must equal cnt.
 * The caller re-sorts each row by sampled time with numpy (utils/utils.py:192-196: an unstable argsort whose tie order is numpy's).
 * lstep_count_before_host: cnt of every row (the sizes of the probability slices). */
int lstep_count_before_host(const int64_t* indptr, const double* ts, int64_t num_rows, const int64_t* node_ids, const double* times, int64_t m,
                            int64_t* out_count);
int lstep_sample_random_host(const int64_t* indptr, const int64_t* nbr, const int64_t* eid, const double* ts, int64_t num_rows,
                             const int64_t* node_ids, const double* times, int64_t m, int32_t num_neighbors, const float* p_values,
                             const int64_t* p_offsets, uint32_t* mt_key, int32_t* mt_pos, int64_t* out_nbr, int64_t* out_eid, float* out_t);
/* The same draws WITH the re-sort of utils/utils.py:192-196 (ABI 40): a node's history is time-sorted and float32 rounding is monotone, so
 * ordering a row's sampled slots by float32 time is ordering the drawn positions -- except among distinct positions with equal float32 times,
 * where the reference's order is numpy's unstable argsort's.  Phase A draws on the calling thread (the generator is sequential); phase B sorts
 * positions and gathers the triples on num_threads host threads (<= 0: hardware concurrency).  Rows WITHOUT such a tie are written in time
 * order, ambiguous[r] = 0; rows with one are written in draw order, ambiguous[r] = 1 (the caller sorts those with numpy); rows without history
 * are untouched, ambiguous[r] = 0.  ambiguous: uint8 [m]. */
int lstep_sample_random_sorted_host(const int64_t* indptr, const int64_t* nbr, const int64_t* eid, const double* ts, int64_t num_rows,
                                    const int64_t* node_ids, const double* times, int64_t m, int32_t num_neighbors, const float* p_values,
                                    const int64_t* p_offsets, uint32_t* mt_key, int32_t* mt_pos, int64_t* out_nbr, int64_t* out_eid,
                                    float* out_t, uint8_t* ambiguous, int32_t num_threads);
int lstep_update_entries_p1(const int32_t* order, int64_t num_entries, const int64_t* src, const int64_t* dst, const double* times,
                            const float* now32, int64_t batch, int32_t* ent_row, float* ent_dt, void* stream);
int lstep_update_keys_p2(const int64_t* nbr, int64_t n, int32_t sentinel, int32_t world, int32_t rank, int32_t* keys, void* stream);
int lstep_update_entries_p2(const int32_t* order, const int32_t* seg, int64_t n_real, const int64_t* bn, const float* nt, const float* now32,
                            int32_t num_neighbors, int32_t shift, const int32_t* uniq, int64_t nseg, int32_t* ent_row, float* ent_dt,
                            int32_t* ent_seg, int64_t* touched, void* stream);

/* Operands of lstep_head_fwd / _bwd from the reference-shaped parameters of MergeLayer (models/modules.py:52-53: fc1.weight
 * [hidden, 2 * half], fc1.bias [hidden], fc2.weight [1, hidden]; hidden, half <= 176) in one launch: w [176, 352] = [first half | second half]
 * zero-padded, wt [352, 176] its transpose, b1 [176], w2 [176]. */
int lstep_head_pack(const float* fc1_w, const float* fc1_b, const float* fc2_w, int32_t hidden, int32_t half, float* w, float* wt, float* b1,
                    float* w2, void* stream);

/* `optimizer.step()` of the training loop (train_LSTEP_link_prediction.py:283; utils/utils.py:49-67 creates a plain torch.optim.Adam: no
 * amsgrad, L2-style weight decay) for up to LSTEP_ADAM_MAX_TENSORS fp32 tensors in ONE launch: host arrays of device pointers (they travel
 * in the kernel arguments), numel per tensor, `steps` float32 on the device = step counts INCLUDING this step; tensor t uses
 * steps[step_index[t]] (host array; NULL = t).
 * Same arithmetic as torch._fused_adam_ (bias corrections in double, moments and update in float). */
#define LSTEP_ADAM_MAX_TENSORS 48
int lstep_adam_step(int32_t num_tensors, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                    const int64_t* numel, const float* steps, const int32_t* step_index, float lr, float beta1, float beta2, float eps, float weight_decay, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LSTEP_HIP_H */
