/*
 * ppp_mi355x.h -- C ABI of libppp_mi355x.so: the MI355X (gfx950) implementation of
 * PatchPerPix's vote_instances device path.
 *
 * This is the drop-in boundary.  In the reference the plug is the pycuda shim
 * PatchPerPix/vote_instances/cuda_code.py:5-59 (make_kernel / alloc_zero_array / sync)
 * plus the four JIT-templated kernels under PatchPerPix/vote_instances/cuda/.  Here the
 * kernels are compiled ahead of time for gfx950; shapes, thresholds and the
 * reference's -D build flags are run-time fields of ppp_params.
 *
 * Conventions
 *  - every pointer named d_* is a DEVICE pointer owned by the caller (the Python
 *    host code carries them as torch-ROCm tensors); the library never allocates,
 *    frees or retains them.  h_* pointers are host pointers.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls are
 *    asynchronous on that stream unless stated otherwise.
 *  - return value 0 = success, negative = error; ppp_last_error() gives the message
 *    (thread-local).  Nothing falls back to the CPU: without a HIP device every
 *    compute entry point fails with PPP_ERR_NO_DEVICE.
 *  - volumes are (Z, Y, X) C-contiguous; predictions are (C, Z, Y, X) with
 *    C = pz*py*px, channel r = (z*py + y)*px + x of the patch, float32 or float16
 *    (the zarr dtype written by experiments/flylight/setups/setup01/predict_no_gp.py:
 *    243-257; widening to f32 is exact, which is what the reference does on load,
 *    vote_instances.py:193-200).
 *
 * Consensus layout (ppp_params.cons_layout)
 *  PPP_CONS_COMPACT   [n_planes][bz][by][bx] float32 over the base-voxel box cons_box,
 *                     one plane per lexicographically positive pixel offset
 *                     d = (dz,dy,dx), |d_i| <= p_i-1: plane = L-1 with
 *                     L = (dz*(2py-1) + dy)*(2px-1) + dx  (L > 0 <=> d positive).
 *                     These are exactly the planes the reference ever writes.
 *  PPP_CONS_VOXEL_MAJOR [bz][by][bx][W] float32, W = (2pz-1)(2py-1)(2px-1): for every base
 *                     voxel v the consensus between v and v+q for ALL signed offsets q
 *                     (index Lc + (qz*(2py-1)+qy)*(2px-1)+qx, Lc = (W-1)/2); written directly by
 *                     ppp_consensus where ppp_consensus_writes_voxel_major() says so, else made from a
 *                     COMPACT array by ppp_cons_to_voxel_major; the layout ppp_patch_graph
 *                     is fastest on (every lane sweeps contiguous memory).
 *  PPP_CONS_REFERENCE [NSZ][NSY][NSX][Z][Y][X] float32, index o = d + p - 1,
 *                     NS = 2p (NSZ = 1 when pz == 1): the reference's array
 *                     (consensus_array.py:99-106); cons_box must be the whole volume.
 */
#ifndef PPP_MI355X_H
#define PPP_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PPP_ABI_VERSION 5

enum ppp_error {
    PPP_OK = 0,
    PPP_ERR_INVALID_ARG = -1,
    PPP_ERR_NO_DEVICE = -2,
    PPP_ERR_HIP = -3,
    PPP_ERR_UNSUPPORTED = -4,
    PPP_ERR_WORKSPACE = -5
};

enum ppp_dtype { PPP_F32 = 0, PPP_F16 = 1 };

/* background rule for the second pixel of a pair: -DUSE_INV_TH / -DUSE_HALF_TH /
 * -DUSE_LESS_THAN_TH (utilVoteInstances.py:389-406) */
enum ppp_bg_rule { PPP_BG_INV_TH = 0, PPP_BG_HALF_TH = 1, PPP_BG_LESS_THAN_TH = 2 };
/* vote value: (none = 1) / -DPROB_PRODUCT / -DNORM_PROB_PRODUCT (utilVoteInstances.py:412-427) */
enum ppp_value_rule { PPP_VAL_COUNT = 0, PPP_VAL_PROB_PRODUCT = 1, PPP_VAL_NORM_PROB_PRODUCT = 2 };
enum ppp_cons_layout { PPP_CONS_COMPACT = 0, PPP_CONS_REFERENCE = 1, PPP_CONS_VOXEL_MAJOR = 2 };

typedef struct ppp_box {
    int32_t z0, y0, x0; /* inclusive */
    int32_t z1, y1, x1; /* exclusive */
} ppp_box;

typedef struct ppp_params {
    int32_t abi_version;   /* PPP_ABI_VERSION */
    int32_t Z, Y, X;       /* DATAZSIZE, DATAYSIZE, DATAXSIZE                            */
    int32_t pz, py, px;    /* PSZ, PSY, PSX (odd)                                        */
    double th;             /* TH  = patch_threshold, compared in double like the         */
    double thi;            /* THI   reference's literal (utilVoteInstances.py:361-375)   */
    int32_t bg_rule;       /* enum ppp_bg_rule                                           */
    int32_t value_rule;    /* enum ppp_value_rule                                        */
    int32_t use_overlap;   /* -DOVERLAP: skip pixels whose overlap mask is non-zero      */
    int32_t normalise;     /* consensus_norm_aff: cons /= count (normConsensusArray.cu)  */
    int32_t norm_rank;     /* -DNORM_PATCH_RANK                                          */
    int32_t count_pos_neg; /* -DCOUNT_POS_NEG                                            */
    int32_t norm_aff;      /* -DNORM_PATCH_AFFINITY                                      */
    int32_t cons_layout;   /* enum ppp_cons_layout                                       */
    ppp_box cons_box;      /* base voxels held by the consensus buffer (tile)            */
    int32_t origin_z, origin_y, origin_x; /* global coordinate of local voxel (0,0,0) when the
                              buffers hold a sub-volume (slab of a larger volume); only the
                              per-pair LCG seed of ppp_patch_graph depends on absolute
                              coordinates (computePatchGraph.cu:24-27)                          */
    int32_t ring_z;        /* 0, or -- VOXEL_MAJOR rows only -- the number of z-slices of a RING the
                              row buffer holds: the row of voxel (z, y, x) lives in slice
                              (z + origin_z) mod ring_z of a [ring_z][by][bx][W] buffer, cons_box
                              (at most ring_z slices thick) says which slices are current.  A caller
                              that sweeps a column of tiles upwards computes every base voxel once
                              (ppp_consensus_part over the new slices) and keeps the rows its next
                              tile still needs; read by ppp_rank_patches_vm and
                              ppp_patch_graph_by_patch*.  (The reference holds the whole array in
                              managed memory: consensus_array.py:99-106.)                         */
    int32_t pred_clean;    /* 1: the caller KNOWS (ppp_pred_check said so for the buffer it passes as
                              d_pred) that every prediction value lies in [0, 1] and none falls
                              between the two class tests (with the shipped rule: none equals TH);
                              S1 then takes a shorter classification of its operands -- same bits.
                              0 (or anything else): not known; every kernel serves any input.      */
    int32_t rank_tile;     /* ppp_rank_patches_vm, cubic 5 / 7 / 9 patches: the tile of centres a workgroup
                              takes -- 0: the library's rule; 1: 8 x 8 x 16; 2: 8 x 16 x 16; 3: 16 x 8 x 16
                              (z, y, x).  Same scores whichever; larger tiles stage fewer rows per centre
                              (4.5 instead of 6 times) and win where the memory system is the slower part
                              (measured per box: profiles/r06_l_*), smaller ones where it is not -- a
                              caller with many launches times one of each and keeps the faster.            */
} ppp_params;

/* --- library / device ------------------------------------------------------------- */
int ppp_abi_version(void);
const char *ppp_last_error(void);
/* name of the kernel the last ppp_consensus call launched (which generation / specialisation
 * served the shape): "consensus_v3_kernel", "consensus_v2_kernel" or "consensus_gather_kernel" */
const char *ppp_consensus_kernel_name(void);
/* Development switches (PPP_* environment variables naming a kernel variant or a tile shape)
 * are read once per process, at their first use; ppp_reload_env() makes the next use read them
 * again (tests that compare variants within one process).                                   */
void ppp_reload_env(void);
/* number of HIP devices visible; 0 if none (never fails) */
int ppp_device_count(void);

/* number of consensus planes and floats of a consensus buffer for these params */
int64_t ppp_cons_planes(const ppp_params *p);
int64_t ppp_cons_elems(const ppp_params *p);

/* --- S1: consensus = scoring + vote + (count) + normalise ---------------------------
 * replaces  create_consensus_array_cuda (consensus_array.py:71-206) and the kernels
 * fillConsensusArray_allPatches (cuda/fillConsensusArray.cu:179-218, run once or twice)
 * + normConsensusArray (cuda/normConsensusArray.cu:32-43).
 * Deterministic: every consensus entry is the float sum of its votes in raster order of
 * the voting patch centres (a legal serialisation of the reference's atomicAdd).
 * d_cons : out, layout per p->cons_layout, fully overwritten (no pre-zeroing needed)
 * d_count: optional out (same layout), the vote counts (-DOUTPUT_CNT pass); may be NULL
 * d_overlap: uint8 (Z,Y,X) or NULL when !use_overlap                                   */
int ppp_consensus(const void *d_pred, int pred_dtype, const uint8_t *d_overlap, float *d_cons,
                  float *d_count, const ppp_params *p, void *stream);
/* 1 when ppp_consensus can write p->cons_layout = PPP_CONS_VOXEL_MAJOR directly for these
 * parameters (then no COMPACT array and no ppp_cons_to_voxel_major are needed); 0 otherwise
 * (write COMPACT, then convert).  d_count must be NULL for that layout. */
int ppp_consensus_writes_voxel_major(const ppp_params *p);
/* ppp_consensus for p->cons_layout = PPP_CONS_VOXEL_MAJOR when the rows are only read by
 * ppp_rank_patches_vm / ppp_patch_graph_by_patch: the entries S[w][-d] whose SOURCE voxel w - d
 * lies outside cons_box are left UNDEFINED instead of being zeroed.  Those consumers never read
 * them: a partner pixel of a patch that lies inside the volume is a voxel of the same window, and
 * the caller's box holds every window it asks about (the tile grown by the radius for the scores,
 * by radius + p - 1 for the pair rows); on a 144 x 152 x 152 box at 9^3 the zeroing pass is 8.7 GB
 * of scattered 4-byte stores, 40 ms next to the kernel's 220.  Same interface and errors as
 * ppp_consensus otherwise (fillConsensusArray.cu:5-218, normConsensusArray.cu:5-43). */
int ppp_consensus_rows(const void *d_pred, int pred_dtype, const uint8_t *d_overlap, float *d_cons,
                       const ppp_params *p, void *stream);

/* The same votes for the base voxels of `part` only (a non-empty sub-box of p->cons_box, which
 * keeps indexing the output): COMPACT planes or open VOXEL_MAJOR rows, nothing outside `part` is
 * touched except the mirrored entries S[u + d][-d] of voxel-major rows, which land wherever u + d
 * lies in cons_box.  What a caller builds with it: a consensus CACHE over a large box filled piece
 * by piece (every base voxel computed once: consensus_array.py:71-206 run over the whole block,
 * fillConsensusArray.cu:5-218), from which ppp_cons_planes_to_rows cuts the rows of a tile for the
 * ranking and, later, for the patch graph -- instead of computing them a second time.  Packed
 * kernel only (ppp_consensus_writes_voxel_major). */
int ppp_consensus_part(const void *d_pred, int pred_dtype, const uint8_t *d_overlap, float *d_cons,
                       const ppp_params *p, const ppp_box *part, void *stream);

/* --- S2: patch ranking ---------------------------------------------------------------
 * replaces rank_patches_cuda (ranked_patches.py:33-74) + kernel rankPatches
 * (cuda/rankPatches.cu:1-161).  Scores are written for the voxels of `score_box`
 * (NULL = whole volume) into d_score (Z,Y,X) float32; border voxels get -1 / -9999999,
 * interior non-foreground voxels 0.  The consensus buffer must cover score_box grown by
 * the patch radius (clipped to the volume).                                             */
int ppp_rank_patches(const void *d_pred, int pred_dtype, const float *d_cons,
                     const uint8_t *d_overlap, float *d_score, const ppp_box *score_box,
                     const ppp_params *p, void *stream);

/* The same scores from a VOXEL_MAJOR consensus (p->cons_layout), row-stationary: every
 * consensus row is staged in LDS once per tile of centres and serves all centres of the tile
 * whose window holds its voxel (csrc/ppp_rank_vm.hip) -- the gather form above re-fetches each
 * entry ~50 times from HBM.  Cubic patches of 3 / 5 / 7 / 9, count_pos_neg = 0; otherwise
 * PPP_ERR_UNSUPPORTED (ppp_rank_workspace_bytes returns 0).  d_work: ppp_rank_workspace_bytes.
 * Rows in a RING (p->ring_z) are read by the workgroup-per-tile kernel only (patches of 5 / 7 / 9):
 * ppp_rank_workspace_bytes with ring_z set returns 0 for any other shape -- ask before planning a ring. */
int64_t ppp_rank_workspace_bytes(const ppp_box *score_box, const ppp_params *p);
int ppp_rank_patches_vm(const void *d_pred, int pred_dtype, const float *d_cons_vm,
                        const uint8_t *d_overlap, float *d_score, const ppp_box *score_box,
                        void *d_work, const ppp_params *p, void *stream);

/* --- S5: patch graph (edge emission) --------------------------------------------------
 * replaces computePatchGraph_cuda (aff_patch_graph.py:113-187) + kernel computePatchGraph
 * (cuda/computePatchGraph.cu:3-136).  d_pairs u32[n_pairs][6] = (z,y,x) of patch A and B,
 * d_aff f32[n_pairs] out (d_aff[i] belongs to row i whatever the processing order).
 * The reference's 512-pairs-per-launch loop with its `offset` argument
 * (aff_patch_graph.py:137-159) is one launch here.
 * d_order (optional, may be NULL): a permutation of 0..n_pairs-1 giving the order in which
 * rows are assigned to lanes.  Results do not depend on it; speed does: rows with the same
 * patch offset (B - A) placed next to each other run with wave-uniform control flow.      */
int ppp_patch_graph(const void *d_pred, int pred_dtype, const float *d_cons,
                    const uint32_t *d_pairs, const uint32_t *d_order, uint64_t n_pairs,
                    float *d_aff, const ppp_params *p, void *stream);

/* Same result, different work decomposition: one workgroup per patch A stages the consensus
 * row of every pixel of A once in LDS and serves all pairs (A, *) from it (VOXEL_MAJOR layout
 * only; px in {3,5,7,9}, otherwise PPP_ERR_UNSUPPORTED).  The caller groups the rows by patch A:
 *   d_order            row ids, rows of one patch A contiguous (any order inside a group; rows
 *                      with similar B - A next to each other diverge least)
 *   d_group_start      int64 [n_groups + 1], positions in d_order where the groups start
 *   d_chunk_offsets    int64 [n_groups + 1], exclusive scan of ceil(group size / chunk) with
 *                      chunk = ppp_patch_graph_by_patch_chunk(p) (rows per workgroup, 0 if
 *                      the patch shape is unsupported); n_blocks = d_chunk_offsets[n_groups] */
int32_t ppp_patch_graph_by_patch_chunk(const ppp_params *p);
/* A second, smaller workgroup size for lists with few rows per patch (e.g. after set-cover
 * thinning): ppp_patch_graph_by_patch_chunked takes the chunk the caller built
 * d_chunk_offsets with -- either of the two values above.                                 */
int32_t ppp_patch_graph_by_patch_chunk_small(const ppp_params *p);
int ppp_patch_graph_by_patch_chunked(const void *d_pred, int pred_dtype, const float *d_cons_vm,
                                     const uint32_t *d_pairs, const uint32_t *d_order,
                                     const int64_t *d_group_start, const int64_t *d_chunk_offsets,
                                     int32_t n_groups, int64_t n_blocks, int32_t chunk, float *d_aff,
                                     const ppp_params *p, void *stream);
/* The thinning decisions of the patch intersection made beforehand (computePatchGraph.cu:75-86:
 * the pair's LCG advances on every combination of foreground pixels z1 of A, z2 of B that both
 * lie in the intersection of the two windows; the decisions depend on the two foreground sets and
 * the pair's seed only).  For pair rows whose windows intersect, ppp_patch_graph_lcg writes, at
 * uint64 word d_drop_off[pos] of d_drops (pos = position of the row in d_order), the masks of
 * DROPPED candidates [intersection pixel of A][intersection plane of B][chunk of candidate rows]:
 * ppp_patch_graph_lcg_words(dz, dy, dx) words for patch offset B - A (0 = no intersection).
 *   d_lcg_pos   int64 [n_lcg] positions in d_order of the rows to serve (d_drop_off[pos] >= 0;
 *               rows of similar offset next to each other run with uniform control flow)
 *   d_drop_off  int64 [rows in d_order], < 0 = no masks for this row
 * ppp_patch_graph_by_patch_lcg = ppp_patch_graph_by_patch_chunked that reads those masks and
 * runs the generator itself only for rows without (d_drop_off/d_drops NULL: for all).  Same bits.
 * ppp_patch_graph_lcg_words is 0 for every offset when the per-patch kernel of this patch width
 * does not read masks (the 25-wide 2-d kernel: it runs the generator itself). */
int64_t ppp_patch_graph_lcg_words(int32_t dz, int32_t dy, int32_t dx, const ppp_params *p);
int ppp_patch_graph_lcg(const void *d_pred, int pred_dtype, const uint32_t *d_pairs,
                        const uint32_t *d_order, const int64_t *d_lcg_pos, int64_t n_lcg,
                        const int64_t *d_drop_off, uint64_t *d_drops, const ppp_params *p, void *stream);
int ppp_patch_graph_by_patch_lcg(const void *d_pred, int pred_dtype, const float *d_cons_vm,
                                 const uint32_t *d_pairs, const uint32_t *d_order,
                                 const int64_t *d_group_start, const int64_t *d_chunk_offsets,
                                 int32_t n_groups, int64_t n_blocks, int32_t chunk, float *d_aff,
                                 const int64_t *d_drop_off, const uint64_t *d_drops,
                                 const ppp_params *p, void *stream);
int ppp_patch_graph_by_patch(const void *d_pred, int pred_dtype, const float *d_cons_vm,
                             const uint32_t *d_pairs, const uint32_t *d_order,
                             const int64_t *d_group_start, const int64_t *d_chunk_offsets,
                             int32_t n_groups, int64_t n_blocks, float *d_aff,
                             const ppp_params *p, void *stream);

/* --- S6: labelling -------------------------------------------------------------------
 * replaces setAffgraph (aff_patch_graph.py:31-40) + the connected-components branch of
 * affGraphToInstances (graph_to_labeling.py:50-54,61-86).
 *
 * ppp_label_components: union-find (global atomics) over the rows with aff > 0; nodes are
 * patch centres identified by their linear voxel index.  Output, for every node k of the
 * caller's node list d_nodes u32[n_nodes][3] (normally the selected patches):
 *   d_node_key[k] (uint32): the ORDER KEY of the component node k belongs to, or 0xFFFFFFFF
 *   if it is in no component (no positive edge).  The key of a component is the smallest
 *   position 2*row+side at which any of its members first appears among the rows with
 *   aff != 0 -- sorting the distinct keys ascending reproduces networkx's component
 *   enumeration order (graph_to_labeling.py:54-61).
 * d_work: workspace of ppp_label_workspace_bytes(p) bytes.                              */
size_t ppp_label_workspace_bytes(const ppp_params *p);
int ppp_label_components(const uint32_t *d_pairs, const float *d_aff, uint64_t n_pairs,
                         const uint32_t *d_nodes, uint64_t n_nodes, uint32_t *d_node_key,
                         void *d_work, const ppp_params *p, void *stream);

/* The same labelling in STREAMING form: the pair rows of a large volume need not exist at
 * the same time (tile by tile on one GPU), nor on one device (every rank labels the rows of
 * its own patches; the forests are merged afterwards -- the boundary-label merge).  Both
 * stitch_patch_graph.py's "one global graph from per-block pair lists" (:110-399) and
 * setAffgraph + connected_components (see above) are covered.
 *   ppp_label_begin        state of every node of d_nodes (parent = itself, no key)
 *   ppp_label_add          a batch of rows with their affinities; the row ids are GLOBAL
 *                          positions in the canonical pair list (d_row_ids int64[n], or
 *                          first_row_id + i when d_row_ids is NULL).  Order-free: atomicMin of
 *                          2*id+side on the nodes of rows with aff != 0, lock-free unions
 *                          (CAS hooking under the smaller root) for rows with aff > 0.
 *   ppp_label_union_edges  unions given as node pairs (linear voxel indices, int64): how
 *                          another rank's forest (node, parent[node]) is merged
 *   ppp_label_finish       d_node_key int64[n_nodes]: the component's order key (smallest
 *                          2*id+side of a member with a positive edge) or PPP_LABEL_NONE_KEY
 * Workspace (ppp_label_workspace_bytes): four volumes indexed by linear voxel index,
 * parent u32[V] | has_positive_edge u32[V] | firstpos u64[V] | key u64[V]; callers that merge
 * ranks read / write the node entries of these arrays directly.                           */
#define PPP_LABEL_NONE_KEY (1ull << 62)
int ppp_label_begin(const uint32_t *d_nodes, uint64_t n_nodes, void *d_work, const ppp_params *p,
                    void *stream);
int ppp_label_add(const uint32_t *d_pairs, const float *d_aff, const int64_t *d_row_ids,
                  int64_t first_row_id, uint64_t n_pairs, void *d_work, const ppp_params *p,
                  void *stream);
int ppp_label_union_edges(const int64_t *d_a, const int64_t *d_b, uint64_t n, void *d_work,
                          const ppp_params *p, void *stream);
int ppp_label_finish(const uint32_t *d_nodes, uint64_t n_nodes, int64_t *d_node_key, void *d_work,
                     const ppp_params *p, void *stream);

/* ppp_paint_instances: for every node k (d_nodes u32[n_nodes][3]) with label
 * d_labels[k] > 0, write the label into every voxel of its window whose patch value is
 * > TH, keeping the maximum label per voxel ("later components overwrite earlier ones",
 * graph_to_labeling.py:73-84, with labels = component rank + 1).  d_instances u32 (Z,Y,X)
 * is updated in place (caller zero-initialises).                                        */
int ppp_paint_instances(const void *d_pred, int pred_dtype, const uint32_t *d_nodes,
                        const uint32_t *d_labels, uint64_t n_nodes, uint32_t *d_instances,
                        const ppp_params *p, void *stream);

/* ppp_paint_patch_rows: the same painting with the patches given as a table instead of a dense
 * prediction block: d_rows [n_nodes][C] (float16 / float32), row k = pred[:, node k].  Serves
 * affGraphToInstances(sparse_labels=True) of the blockwise driver (graph_to_labeling.py:66-72,
 * stitch_patch_graph.py:388-396), where the patch of every node of the GLOBAL graph is read
 * from the prediction store -- here gathered chunk by chunk into the table.               */
int ppp_paint_patch_rows(const void *d_rows, int rows_dtype, const uint32_t *d_nodes,
                         const uint32_t *d_labels, uint64_t n_nodes, uint32_t *d_instances,
                         const ppp_params *p, void *stream);

/* --- the reference's NumPy-semantics stages (`cuda=False`) ------------------
 * A different function from the kernels above: integer votes.  Per interior foreground centre c
 * (utilVoteInstances.py:59-92, get_patch_sets.py:32-79; float32 compares):
 *     pf_c = { v in win(c) : pred[r_v][c] > th     and foreground[v] }
 *     pb_c = { v in win(c) : pred[r_v][c] < 1 - th and foreground[v] }
 * Votes: int16 [ppp_np_vote_planes(p)][Z][Y][X]; plane q >= 1 = the lexicographically positive
 * offset with linear signed index (dz (2py-1) + dy)(2px-1) + dx = q (the COMPACT plane order),
 * plane 0 = the zero offset (only ever non-zero for th < 0.5).  cons_box must be the whole volume.
 *
 * ppp_np_consensus    replaces create_consensus_array (consensus_array.py:18-68) + fillLookup
 *                     (utilVoteInstances.py:19-56): +1 per centre on the key of every unordered
 *                     pair of pf_c, -1 per centre on every DISTINCT key of a pair (pf_c, pb_c).
 * ppp_np_rank_patches replaces rank_patches (ranked_patches.py:76-105): d_score int32 (Z,Y,X),
 *                     #(ff votes > 0) - #(ff votes <= 0) + #(fb votes < 0) - #(fb votes >= 0) for
 *                     interior foreground centres, 0 elsewhere.
 * ppp_np_patch_graph  replaces computePatchGraph's NumPy branch (aff_patch_graph.py:209-282) for
 *                     the candidate rows d_rows int32 [n][6] = (A, B): d_weight[i] = sum of the
 *                     votes over all (p in pf_A, q in pf_B), |p - q| < patch shape on every axis,
 *                     p != q, with pf taken against d_mask (= mask_to_cover); d_count[i] = the
 *                     number of such pairs (the edge exists iff it is > 0).                      */
int64_t ppp_np_vote_planes(const ppp_params *p);
int ppp_np_consensus(const void *d_pred, int pred_dtype, const uint8_t *d_foreground, int16_t *d_votes,
                     const ppp_params *p, void *stream);
int ppp_np_rank_patches(const void *d_pred, int pred_dtype, const uint8_t *d_foreground, const int16_t *d_votes,
                        int32_t *d_score, const ppp_params *p, void *stream);
int ppp_np_patch_graph(const void *d_pred, int pred_dtype, const uint8_t *d_mask, const int16_t *d_votes,
                       const int32_t *d_rows, uint64_t n_rows, int64_t *d_weight, int32_t *d_count,
                       const ppp_params *p, void *stream);

/* --- layout helper ---------------------------------------------------------------------
 * expand a whole-volume COMPACT consensus into the reference's [NSZ][NSY][NSX][Z][Y][X]
 * array (what create_consensus_array_cuda returns / save_consensus writes).             */
int ppp_cons_to_reference(const float *d_cons_compact, float *d_cons_reference,
                          const ppp_params *p, void *stream);

/* re-layout a COMPACT consensus (p->cons_box, p->cons_layout ignored) as VOXEL_MAJOR       */
int ppp_cons_to_voxel_major(const float *d_cons_compact, float *d_cons_voxel_major,
                            const ppp_params *p, void *stream);

/* VOXEL_MAJOR rows of p->cons_box from COMPACT planes that are indexed by a box of their own,
 * `planes_box` (it must hold cons_box; same coordinate frame): S[v][+d] = planes[d][v],
 * S[v][-d] = planes[d][v - d], 0 where v - d lies outside planes_box (an entry nothing reads, as
 * with ppp_consensus_rows).  ppp_cons_to_voxel_major is the case planes_box == cons_box.
 * (aff_patch_graph.py:113-187 / ranked_patches.py:33-74 read the consensus array the
 * reference keeps whole in managed memory; this is how a tile's share of it reaches HBM rows.) */
int ppp_cons_planes_to_rows(const float *d_planes, const ppp_box *planes_box, float *d_rows,
                            const ppp_params *p, void *stream);

/* --- foreground / patch bit helpers used by the host stages --------------------------
 * ppp_patch_bits: for n centres (d_centres u32[n][3]) pack (pred[r][c] > thresh) for
 * r = 0..C-1 into ceil(C/32) uint32 words each (bit r%32 of word r/32), d_bits
 * u32[n][ceil(C/32)].  The compare is float32 against (float)thresh, which is how NumPy
 * evaluates `patch > fc_threshold` for a float32 patch (foreground_cover.py:156-158).      */
int ppp_patch_bits(const void *d_pred, int pred_dtype, const uint32_t *d_centres,
                   uint64_t n, double thresh, uint32_t *d_bits, const ppp_params *p,
                   void *stream);
/* the same bits for every voxel, d_bits_vol u32[Z*Y*X][ceil(C/32)] (the prediction is read once,
 * coalesced; callers with many centres gather their rows from it)                          */
int ppp_patch_bits_volume(const void *d_pred, int pred_dtype, double thresh, uint32_t *d_bits_vol,
                          const ppp_params *p, void *stream);

/* --- greedy foreground cover on the device ---------------------------------------------
 * ppp_cover_pass: one pass of computeForegroundCoverLoop (foreground_cover.py:111-180) as an
 * exact priority-parallel algorithm (csrc/ppp_cover.hip): every patch whose state is 0 takes
 * part; it ends up 1 (selected: more than pix_th voxels of the running mask were inside its
 * window where its prediction is > fc_threshold when its turn came) or 2 (not selected).
 *   d_mask    u8  [Z][Y][X]         running mask, cleared in place like mask_running
 *   d_bits    u32 [n][ceil(C/32)]   ppp_patch_bits(fc_threshold) of the ranked patches
 *   d_lin     i64 [n]               linear centre index of ranked patch k (interior centres)
 *   d_state   i32 [n]               in/out, see above (1 / 2 on entry: does not take part)
 *   d_cleared i32 [n]               out: interior voxels patch k cleared (0 if not selected)
 *   d_work    ppp_cover_workspace_bytes(n, p) bytes
 * The loop's stop rule ("interior of the mask is empty", checked before every patch) is not
 * applied here: the caller cuts the selected list, in rank order, after the patch at which
 * the running sum of d_cleared reaches the number of set interior voxels.  *rounds (may be
 * NULL) returns the number of parallel rounds.  Synchronises the stream.                   */
int64_t ppp_cover_workspace_bytes(int64_t n, const ppp_params *p);
int ppp_cover_pass(uint8_t *d_mask, const uint32_t *d_bits, const int64_t *d_lin, int64_t n,
                   int32_t pix_th, int32_t *d_state, int32_t *d_cleared, void *d_work,
                   const ppp_params *p, void *stream, int32_t *rounds);
/* The same pass with the patch bits in a table that has one row per VOXEL (row of patch k =
 * d_bits_by_voxel + (d_lin[k] - first_voxel) * words): the tiled assembly fills such a table
 * tile by tile while the prediction of a tile is resident (prediction provider: the bits cannot
 * be recomputed later) -- reordering it into rank order would need a second copy of it (92 bytes
 * per voxel at 9^3).  Same selections as ppp_cover_pass (foreground_cover.py:111-180).        */
int ppp_cover_pass_voxel_bits(uint8_t *d_mask, const uint32_t *d_bits_by_voxel, int64_t first_voxel,
                              const int64_t *d_lin, int64_t n, int32_t pix_th, int32_t *d_state,
                              int32_t *d_cleared, void *d_work, const ppp_params *p, void *stream,
                              int32_t *rounds);

/* --- S4: set-cover thinning on the device -------------------------------------------------
 * replaces thinOutForegroundCover (foreground_cover.py:183-256, sample == 1.0; a host loop in the
 * reference): while the interior of the running mask is not empty, keep the first patch that
 * covers the most still uncovered voxels and clear them.  Exact priority-parallel form of that
 * loop (csrc/ppp_cover.hip): rounds of count / 3-d neighbourhood minimum of the key
 * (count descending, index ascending) / keep-and-clear, then the stop rule from the kept
 * patches' keys and cleared-interior counts.
 *   d_mask  u8  [Z][Y][X]        mask_to_cover (not modified)
 *   d_bits  u32 [n][ceil(C/32)]  ppp_patch_bits(fc_threshold) of the selected patches, in the
 *                                order of the selected list (= rank order)
 *   d_lin   i64 [n]              linear centre index of selected patch k
 *   d_keep  u8  [n]              out: 1 = kept
 *   d_work  ppp_thin_workspace_bytes(n, p) bytes
 * *rounds (may be NULL): number of parallel rounds.  Synchronises the stream.               */
int64_t ppp_thin_workspace_bytes(int64_t n, const ppp_params *p);
int ppp_thin_cover(const uint8_t *d_mask, const uint32_t *d_bits, const int64_t *d_lin, int64_t n,
                   uint8_t *d_keep, void *d_work, const ppp_params *p, void *stream,
                   int32_t *rounds);

/* The same rounds one step at a time, for a cover SHARDED over ranks by z (every rank: its
 * own slices + a halo of p-1 slices, local coordinates, origin_z = first global slice).
 *   ppp_cover_open    d_lin int64[n]: LOCAL linear index of the rank's own ranked patches (rank
 *                     order), d_rank_id int32[n]: their GLOBAL rank (what neighbours compare
 *                     with); d_state / d_cleared as in ppp_cover_pass, but indexed locally
 *   ppp_cover_step    PPP_COVER_COUNT / _FILTER / _SELECT (d_bits u32[n][words] local table;
 *                     global_z: slices of the whole volume, for the interior test)
 *   ppp_cover_alive   *alive = some own patch was undecided at the last count (host sync)
 *   ppp_cover_zone    export (import = 0) / import (1) of the local slices [z_lo, z_hi) around
 *                     a slab boundary: d_rank int32 (own slices [own_lo, own_hi), INT32_MAX
 *                     elsewhere -> MIN over ranks), d_mask u8 0/1 and d_clean u8 (1 = not dirty)
 *                     (-> MIN over ranks); d_rank or the d_mask/d_clean pair may be NULL
 *   ppp_cover_close   the running mask back into d_mask (bytes whose bit was cleared -> 0)
 * Workspace: ppp_cover_workspace_bytes of the LOCAL geometry.                               */
#define PPP_COVER_COUNT 0
#define PPP_COVER_FILTER 1
#define PPP_COVER_SELECT 2
int ppp_cover_open(const uint8_t *d_mask, const int64_t *d_lin, const int32_t *d_rank_id, int64_t n,
                   const int32_t *d_state, int32_t *d_cleared, void *d_work, const ppp_params *p,
                   void *stream);
int ppp_cover_step(int32_t what, const uint32_t *d_bits, int32_t pix_th, int32_t *d_state,
                   int32_t *d_cleared, void *d_work, int32_t global_z, const ppp_params *p,
                   void *stream);
int ppp_cover_alive(void *d_work, const ppp_params *p, void *stream, int32_t *alive);
int ppp_cover_close(uint8_t *d_mask, void *d_work, const ppp_params *p, void *stream);
int ppp_cover_zone(int32_t import, void *d_work, int32_t z_lo, int32_t z_hi, int32_t own_lo,
                   int32_t own_hi, int32_t *d_rank, uint8_t *d_mask, uint8_t *d_clean,
                   const ppp_params *p, void *stream);

/* The set-cover thinning (ppp_thin_cover, foreground_cover.py:183-256) one round step at a time, SHARDED
 * over ranks by z like the cover above (round 6): every rank holds its own slices + p-1 halo slices
 * (local coordinates, origin_z = first global slice) and thins its OWN selected patches.
 *   ppp_thin_open    d_lin int64[n]: LOCAL linear index of the own selected patches, d_index int32[n]:
 *                    their positions in the GLOBAL selected list (the tie-break of the loop's argmax,
 *                    :210); d_state / d_count / d_cleared int32[n] are zeroed here and filled by the
 *                    steps: state 1 = kept in some round, 2 = retired with nothing left to cover;
 *                    count = voxels it covered when it was kept; cleared = interior voxels among them
 *   ppp_thin_step    PPP_COVER_COUNT / _FILTER / _SELECT (d_bits u32[n][words], the own patches' bits)
 *   ppp_thin_alive   *alive = some own patch was undecided at the last count (host sync)
 *   ppp_thin_zone    export / import of the local slices [z_lo, z_hi) around a slab boundary: d_key
 *                    int64 (own slices [own_lo, own_hi), INT64_MAX elsewhere -> MIN over ranks),
 *                    d_mask / d_clean as in ppp_cover_zone
 *   ppp_thin_close   the running mask back into d_mask
 * The loop's stop rule ("interior empty", tested before every pick) is the caller's: the kept patches
 * of all ranks in the order of their keys (count descending, index ascending), cut where the cumulative
 * cleared interior voxels reach the interior voxels the mask held at the start.
 * Workspace: ppp_thin_shard_workspace_bytes of the LOCAL geometry.                             */
int64_t ppp_thin_shard_workspace_bytes(const ppp_params *p);
int ppp_thin_open(const uint8_t *d_mask, const int64_t *d_lin, const int32_t *d_index, int64_t n, int32_t *d_state,
                  int32_t *d_count, int32_t *d_cleared, void *d_work, const ppp_params *p, void *stream);
int ppp_thin_step(int32_t what, const uint32_t *d_bits, int32_t *d_state, int32_t *d_count, int32_t *d_cleared,
                  void *d_work, int32_t global_z, const ppp_params *p, void *stream);
int ppp_thin_alive(void *d_work, const ppp_params *p, void *stream, int32_t *alive);
int ppp_thin_close(uint8_t *d_mask, void *d_work, const ppp_params *p, void *stream);
int ppp_thin_zone(int32_t import, void *d_work, int32_t z_lo, int32_t z_hi, int32_t own_lo, int32_t own_hi,
                  int64_t *d_key, uint8_t *d_mask, uint8_t *d_clean, const ppp_params *p, void *stream);

/* --- patch pairs on the device ---------------------------------------------------------
 * replaces computeAndStorePatchPairs (aff_patch_graph.py:43-110).  d_sorted_zyx int32[n][3]
 * is the selected list stably sorted by x (aff_patch_graph.py:45).  Two calls: count the
 * partners j > i of every patch i, then -- after an exclusive scan of the counts by the
 * caller -- write the rows (pts[i], pts[j]) in (i, j) order, followed by the n self pairs
 * when include_single (includeSinglePatchCCS).  Rows: d_rows u32[n_pair_rows (+ n)][6].   */
int ppp_patch_pairs_count(const int32_t *d_sorted_zyx, int64_t n, int32_t max_ps_dist,
                          int64_t *d_counts, const ppp_params *p, void *stream);
int ppp_patch_pairs_fill(const int32_t *d_sorted_zyx, int64_t n, int32_t max_ps_dist,
                         const int64_t *d_offsets, int64_t n_pair_rows, int32_t include_single,
                         uint32_t *d_rows, const ppp_params *p, void *stream);
/* The rows of a SUBSET of first patches (the patches of one tile / one rank; the blockwise
 * form of computeAndStorePatchPairs, stitch_patch_graph.py:209-248).  d_subset int64[m]:
 * indices into the x-sorted list.  count: d_counts[d_subset[k]] only.  fill: rows of subset
 * entry k start at local row d_local_offsets[k]; d_row_ids receives the GLOBAL row id of every
 * local row (d_global_offsets[i] + j; the self pair of patch i has id n_rows_total + i and
 * local position n_local_rows + k).                                                       */
int ppp_patch_pairs_count_subset(const int32_t *d_sorted_zyx, int64_t n, int32_t max_ps_dist,
                                 const int64_t *d_subset, int64_t m, int64_t *d_counts,
                                 const ppp_params *p, void *stream);
int ppp_patch_pairs_fill_subset(const int32_t *d_sorted_zyx, int64_t n, int32_t max_ps_dist,
                                const int64_t *d_subset, int64_t m, const int64_t *d_local_offsets,
                                const int64_t *d_global_offsets, int64_t n_local_rows,
                                int64_t n_rows_total, int32_t include_single, uint32_t *d_rows,
                                int64_t *d_row_ids, const ppp_params *p, void *stream);
/* sort keys (int64) that group pair rows by patch offset B - A, then by position of A: an
 * argsort of them is a good d_order for ppp_patch_graph                                    */
int ppp_pair_sort_keys(const uint32_t *d_rows, uint64_t n_rows, int64_t *d_keys,
                       const ppp_params *p, void *stream);
/* sort keys (int64) for ppp_patch_graph_by_patch: key >> 18 is the linear index of patch A (the
 * group), the low bits order a group's rows (intersecting windows first, then by B - A).
 * Rows whose patches are too far apart to share a stored consensus offset (|B - A|_i >
 * 2 (p_i - 1) on some axis) get PPP_PAIR_KEY_FAR: their affinity is exactly 0.0 whatever the
 * data, so a caller that pre-zeroes d_aff leaves them out of the groups.                    */
#define PPP_PAIR_KEY_FAR 0x7FFFFFFFFFFFFFFFll
int ppp_pair_group_keys(const uint32_t *d_rows, uint64_t n_rows, int64_t *d_keys,
                        const ppp_params *p, void *stream);

/* --- host stages (host pointers; they are host code in the reference as well) ---------
 * ppp_host_rank_order: all_patches + rank_patches_by_score (vote_instances.py:276,286-287,
 *   ranked_patches.py:21-30): interior foreground voxels in raster order, stably sorted by
 *   score descending.  out_lin holds linear voxel indices (capacity Z*Y*X); returns count. */
int64_t ppp_host_rank_order(const float *h_score, const uint8_t *h_foreground, const int32_t *vol,
                            const int32_t *patchshape, int64_t *out_lin);
/* ppp_host_cover_pass: one computeForegroundCoverLoop pass (foreground_cover.py:111-180) over
 *   n ranked patches; bits from ppp_patch_bits with fc_threshold; selected / mask / remaining
 *   are updated in place; score_threshold NaN = off, *stopped = 1 when it ended the pass.
 *   Returns #newly selected.                                                               */
int64_t ppp_host_cover_pass(uint8_t *h_mask_running, const uint8_t *h_overlap, const int32_t *vol,
                            const int32_t *patchshape, const int64_t *ranked_lin,
                            const float *ranked_score, const uint32_t *bits, int64_t n,
                            int32_t pix_th, double score_threshold, uint8_t *selected,
                            int64_t *remaining, int32_t *stopped);
/* ppp_host_cover_pass_marked: the same pass with `mark_close_neighboorhood`
 * (foreground_cover.py:141-143, 162-168): h_marked uint8 (Z,Y,X), in/out, shared by the passes of
 * one cover (NULL = the plain pass); a ranked patch with a marked centre is skipped, a selected
 * patch marks the box (0, +-3, +-3) around its centre with NumPy's slice semantics.         */
int64_t ppp_host_cover_pass_marked(uint8_t *h_mask_running, const uint8_t *h_overlap, const int32_t *vol,
                                   const int32_t *patchshape, const int64_t *ranked_lin,
                                   const float *ranked_score, const uint32_t *bits, int64_t n,
                                   int32_t pix_th, double score_threshold, uint8_t *selected,
                                   int64_t *remaining, int32_t *stopped, uint8_t *h_marked);
/* ppp_host_thin_cover: thinOutForegroundCover (foreground_cover.py:183-256), keep[n] out.   */
int64_t ppp_host_thin_cover(const uint8_t *h_mask, const int32_t *vol, const int32_t *patchshape,
                            const int64_t *sel_lin, const uint32_t *bits, int64_t n,
                            uint8_t *keep);

/* ppp_host_skeletonize_3d: the 3-d thinning behind `skeletonize_foreground` (vote_instances.py:
 * 219-224, stitch_patch_graph.py:756-759: skimage.morphology.skeletonize_3d = Lee / Kashyap / Chu
 * 1994, restated from the publication -- scikit-image is absent here, PARITY UNPINNED).
 * h_mask uint8 (Z,Y,X) 0 / non-zero, vol = {Z,Y,X}, h_out uint8 (Z,Y,X) 0 / 1; returns the number
 * of voxels kept or -1. */
int64_t ppp_host_skeletonize_3d(const uint8_t *h_mask, const int32_t *vol, uint8_t *h_out);
/* ppp_host_patch_pairs: computeAndStorePatchPairs (aff_patch_graph.py:43-110) with a grid
 *   hash instead of cKDTree; canonical row order (see file header of ppp_host.cpp).
 *   pairs == NULL returns the row count only.                                               */
int64_t ppp_host_patch_pairs(const int32_t *sel_zyx, int64_t n, const int32_t *patchshape,
                             int32_t max_ps_dist, int32_t include_single, int32_t *sorted_zyx,
                             uint32_t *pairs);

/* ppp_host_mws: mutex watershed on the patch graph (graph_mws.py:7-85 on the graph of
 *   setAffgraph, aff_patch_graph.py:31-40), host code like the reference.  pairs u32 [n][6],
 *   aff f32 [n]; vol = (Z,Y,X).  Writes every node of the graph (rows with aff != 0, first
 *   appearance order) to out_nodes int32 [cap][3] with its instance label in out_labels
 *   (1 + position of its component in the reference's output list; 0 = in no component).
 *   *n_labels = length of that list (emptied components included).  Returns the number of
 *   nodes, or -1 when cap is too small.                                                     */
int64_t ppp_host_mws(const uint32_t *pairs, const float *aff, int64_t n_rows, const int32_t *vol,
                     int32_t *out_nodes, int32_t *out_labels, int64_t cap, int64_t *n_labels);

/* ppp_host_mws_sorted: the loop of graph_mws.mws (:31-77) over an edge list that is already in
 *   the order the reference visits it (ppp_mws_edges makes it on the device): eu / ev int32
 *   [n_edges] node numbers, bit 31 of ev = attractive.  labels int32 [n_nodes] out (1 + position
 *   of the node's component in the reference's output list, 0 = none).  Returns the number of
 *   ids issued (emptied components included).                                               */
int64_t ppp_host_mws_sorted(const int32_t *eu, const int32_t *ev, int64_t n_edges, int64_t n_nodes,
                            int32_t *labels);

/* --- order-defining stages as device sorts (rocPRIM inside the library) ------------------
 * ppp_rank_order: all_patches + rank_patches_by_score (vote_instances.py:276,286-287,
 *   ranked_patches.py:21-30): the interior voxels with d_foreground != 0 in raster order, stably
 *   sorted by score descending.  d_lin int64 [capacity >= count], d_rank_score f32 (may be NULL);
 *   *count out.  Synchronises the stream.  Workspace: ppp_rank_order_workspace_bytes(p).
 * ppp_mws_edges: the edge list of the mutex watershed (setAffgraph + graph_mws.py:17-26): rows
 *   with aff != 0, in networkx's edge order (first appearance of the earlier endpoint, then row),
 *   stably sorted by |aff| descending.  Node numbers are positions in d_nodes u32 [n_nodes][3];
 *   d_eu / d_ev int32 [n_rows] out, bit 31 of d_ev = attractive (aff > 0); *n_edges out.  The
 *   rows must not repeat a node pair (the library's own pair lists never do).  Synchronises.  */
int64_t ppp_rank_order_workspace_bytes(const ppp_params *p);
int ppp_rank_order(const float *d_score, const uint8_t *d_foreground, int64_t *d_lin,
                   float *d_rank_score, int64_t *count, void *d_work, const ppp_params *p,
                   void *stream);
int64_t ppp_mws_edges_workspace_bytes(int64_t n_rows, int64_t n_nodes, const ppp_params *p);
int ppp_mws_edges(const uint32_t *d_pairs, const float *d_aff, int64_t n_rows, const uint32_t *d_nodes,
                  int64_t n_nodes, int32_t *d_eu, int32_t *d_ev, int64_t *n_edges, void *d_work,
                  const ppp_params *p, void *stream);

/* --- synthetic input (bench / tests only; same hash as patchperpix_amd/synth.py) ------
 * fills d_pred (C,Z,Y,X) from a label volume d_labels int32 (Z,Y,X).  voxel_offset is the
 * linear index of local voxel 0 in the global volume (0 unless the buffers are a slab).    */
int ppp_synth_pred(const int32_t *d_labels, void *d_pred, int pred_dtype, uint32_t seed,
                   float hi, float lo, float noise, uint64_t voxel_offset, const ppp_params *p,
                   void *stream);
/* Counter calibration (bench / profiles only; no counterpart in the reference): moves a KNOWN
 * number of bytes -- n_read elements of d_src read with one element per lane and load, n_write
 * floats written to d_dst with one float per lane and store -- so that a rocprofv3 --pmc pass
 * containing the call yields counter / true-bytes ratios for FETCH_SIZE and WRITE_SIZE
 * (kernels calib_read_kernel / calib_write_kernel).                                          */
int ppp_counter_calibration(const void *d_src, int src_dtype, int64_t n_read, float *d_dst, int64_t n_write,
                            void *stream);

/* Is a prediction buffer "clean" (ppp_params.pred_clean)?  One streaming read of n_values
 * contiguous values: *d_unclean (device int32, overwritten) = 0 when every value lies in [0, 1]
 * (as a bit pattern: no negative zero, inf or nan) and is either > TH or < BG of p (the two class
 * tests of fillConsensusArray.cu:44-47, 94-124; with the shipped rule the only value in between is
 * TH itself); bit 0 set: a value outside [0, 1]; bit 1: a value in the dead zone.  The caller reads
 * the flag and sets p->pred_clean = 1 for calls on THAT buffer while its contents stay unchanged.
 * The reference has no such notion: its kernels branch per operand (fillConsensusArray.cu:44-60). */
int ppp_pred_check(const void *d_pred, int pred_dtype, int64_t n_values, int32_t *d_unclean, const ppp_params *p,
                   void *stream);

/* The same generator for a BOX of a larger volume (tile-wise generation, BASELINE config [3]:
 * a rank of the 1024^3 workload holds one tile + halo of the prediction at a time).
 * p: Z / Y / X = extent of the prediction box, origin_* = its position in the volume;
 * d_labels: int32 labels over label_box = {z0, y0, x0, z1, y1, x1} (global), which must hold the
 * box grown by the patch radius (clipped to global_dims = {Z, Y, X}); the noise counter is the
 * global voxel's: bit-identical to ppp_synth_pred on the whole volume. */
int ppp_synth_pred_box(const int32_t *d_labels, const int32_t *label_box, void *d_pred, int pred_dtype,
                       uint32_t seed, float hi, float lo, float noise, const int32_t *global_dims,
                       const ppp_params *p, void *stream);

/* --- ppp+dec: tail of the patch decoder, fused with the scatter into the prediction block ------
 * replaces, for the shipped decoder (default_train_code.toml [model.autoencoder]: num_fmaps
 * [64, 128], kernel_size 3, num_repetitions 2, resize_conv, patchshape 7^3), the last stage of
 * Autoencoder.forward (torch_model.py:537-544: up[1] = nearest upsampling x2 + conv 64 -> 1 + ReLU,
 * up_conv[1] = two 1 -> 1 convolutions without activation, centre crop 8^3 -> 7^3) and the
 * per-voxel scatter of decode_sample (decode.py:61-65) with its float32 (C, Z, Y, X) array.
 * d_x   : float32 [n][fmaps = 64][side^3 = 4^3], the output of up_conv[0] for n foreground voxels
 * d_w1  : float32 [64][27] (= the conv weight [1][64][3][3][3]); d_w2, d_w3: float32 [27]; b*: biases
 * d_dst : int64 [n] linear voxel index of every decoded voxel in the prediction block
 * d_pred: the (C = 343, Z, Y, X) block described by p (float16 or float32): pred[r][dst] written
 * float arithmetic: f32 MFMA over the channels, f32 sums over the taps; checked against the torch
 * restatement of the same layers within a tolerance (the reference ships no decoder to pin to). */
int ppp_decode_tail(const float *d_x, int64_t n, int32_t fmaps, int32_t side, const float *d_w1, float b1,
                    const float *d_w2, float b2, const float *d_w3, float b3, const int64_t *d_dst,
                    void *d_pred, int pred_dtype, const ppp_params *p, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PPP_MI355X_H */
