// ppp_kernels.hpp -- launchers implemented by the .hip translation units.
#pragma once
#include <stdlib.h>
#include <string.h>
#include "ppp_common.hpp"

namespace ppp {

// Development switches (PPP_* environment variables that select a kernel variant or a tile
// shape) are read ONCE per process -- at their first use, and again only after ppp_reload_env()
// (what the tests call after changing one) -- never on every launch.
int env_epoch();
void env_reload();
struct EnvSwitch {
    const char *name;
    int seen = -1;
    bool is_set = false;
    char val[32] = {0};
    explicit EnvSwitch(const char *n) : name(n) {}
    // the variable's value, nullptr when unset
    const char *get() {
        if (seen != env_epoch()) {
            const char *e = getenv(name);
            is_set = e != nullptr;
            if (e) { strncpy(val, e, sizeof(val) - 1); val[sizeof(val) - 1] = 0; }
            seen = env_epoch();
        }
        return is_set ? val : nullptr;
    }
};


hipError_t launch_consensus(const void *pred, int dtype, const uint8_t *ov, float *cons,
                            float *cnt, const Geo &G, hipStream_t s);
hipError_t launch_consensus_v2(const void *pred, int dtype, const uint8_t *ov, float *cons,
                               float *cnt, const Geo &G, hipStream_t s);
const char *last_consensus_kernel();
bool consensus_v3_supported(const Geo &G);
hipError_t launch_consensus_v3(const void *pred, int dtype, const uint8_t *ov, float *cons,
                               float *cnt, const Geo &G, hipStream_t s);
hipError_t launch_consensus_part(const void *pred, int dtype, const uint8_t *ov, float *cons, const Geo &G,
                                 hipStream_t s);
bool consensus_v4_supported(const Geo &G);
hipError_t launch_consensus_v4(const void *pred, int dtype, const uint8_t *ov, float *cons,
                               float *cnt, const Geo &G, hipStream_t s);
hipError_t launch_rank(const void *pred, int dtype, const float *cons, const uint8_t *ov,
                       float *score, const ppp_box &sb, const Geo &G, hipStream_t s);
hipError_t launch_rank_v2(const void *pred, int dtype, const float *cons, const uint8_t *ov,
                          float *score, const ppp_box &sb, const Geo &G, hipStream_t s);
bool rank_vm_supported(const Geo &G);
size_t rank_vm_workspace_bytes(const ppp_box &sb, const Geo &G);
hipError_t launch_rank_vm(const void *pred, int dtype, const float *S, const uint8_t *ov, float *score,
                          const ppp_box &sb, void *work, const Geo &G, hipStream_t s);
bool rank_wg_supported(const Geo &G);
size_t rank_wg_workspace_bytes(const ppp_box &sb, const Geo &G);
hipError_t launch_rank_wg(const void *pred, int dtype, const float *S, const uint8_t *ov, float *score,
                          const ppp_box &sb, void *work, const Geo &G, hipStream_t s);
hipError_t launch_patch_graph(const void *pred, int dtype, const float *cons,
                              const uint32_t *pairs, const uint32_t *order, uint64_t n,
                              float *aff, const Geo &G, hipStream_t s);
int patch_graph_pa_chunk(const Geo &G, bool small = false);
hipError_t launch_patch_graph_pa(const void *pred, int dtype, const float *S, const uint32_t *rows,
                                 const uint32_t *order, const long long *group_start,
                                 const long long *chunk_offsets, int n_groups, long long n_blocks,
                                 int chunk, float *aff, const long long *drop_off,
                                 const unsigned long long *drops, const Geo &G, hipStream_t s);
long long patch_graph_lcg_words(const Geo &G, int dz, int dy, int dx);
hipError_t launch_patch_graph_lcg(const void *pred, int dtype, const uint32_t *rows, const uint32_t *order,
                                  const long long *lcg_pos, long long n, const long long *drop_off,
                                  unsigned long long *drops, const Geo &G, hipStream_t s);
hipError_t launch_label(const uint32_t *pairs, const float *aff, uint64_t n,
                        const uint32_t *nodes, uint64_t n_nodes, uint32_t *node_key, void *work,
                        const Geo &G, hipStream_t s);
hipError_t launch_pairs_count(const int32_t *pts, int64_t n, const int *box, int l1max,
                              int64_t *counts, hipStream_t s, const int64_t *subset = nullptr,
                              int64_t m = 0);
hipError_t launch_pairs_subset(const int32_t *pts, int64_t n, const int *box, int l1max,
                               const int64_t *subset, int64_t m, const int64_t *local_off,
                               const int64_t *gid_off, int64_t n_local_rows, int64_t n_rows_total,
                               int include_single, uint32_t *rows, long long *gid, hipStream_t s);
size_t label_workspace_bytes(const Geo &G);
hipError_t launch_label_begin(const uint32_t *nodes, uint64_t n_nodes, void *work, const Geo &G,
                              hipStream_t s);
hipError_t launch_label_add(const uint32_t *pairs, const float *aff, const long long *gid,
                            long long gid0, uint64_t n, void *work, const Geo &G, hipStream_t s);
hipError_t launch_label_union_edges(const long long *ea, const long long *eb, uint64_t n,
                                    void *work, const Geo &G, hipStream_t s);
hipError_t launch_label_finish(const uint32_t *nodes, uint64_t n_nodes, long long *key64,
                               uint32_t *key32, void *work, const Geo &G, hipStream_t s);
hipError_t launch_pairs_fill(const int32_t *pts, int64_t n, const int *box, int l1max,
                             const int64_t *offsets, int64_t n_pair_rows, int include_single,
                             uint32_t *rows, hipStream_t s);
hipError_t launch_pair_group_keys(const uint32_t *rows, uint64_t n, int64_t *keys, const Geo &G,
                                  hipStream_t s);
hipError_t launch_pair_keys(const uint32_t *rows, uint64_t n, int64_t *keys, const Geo &G,
                            hipStream_t s);
// the reference's NumPy-semantics stages (cuda=False), ppp_numpy_path.hip
hipError_t launch_np_consensus(const void *pred, int dtype, const uint8_t *fg, int16_t *cons, const Geo &G,
                               double th, hipStream_t s);
hipError_t launch_np_rank(const void *pred, int dtype, const uint8_t *fg, const int16_t *cons, int32_t *score,
                          const Geo &G, double th, hipStream_t s);
hipError_t launch_np_graph(const void *pred, int dtype, const uint8_t *mask, const int16_t *cons,
                           const int32_t *rows, uint64_t n, long long *weight, int32_t *count, const Geo &G,
                           double th, hipStream_t s);
hipError_t launch_paint_rows(const void *rows, int dtype, const uint32_t *nodes, const uint32_t *labels,
                             uint64_t n, uint32_t *inst, const Geo &G, hipStream_t s);
hipError_t launch_paint(const void *pred, int dtype, const uint32_t *nodes,
                        const uint32_t *labels, uint64_t n, uint32_t *inst, const Geo &G,
                        hipStream_t s);
hipError_t launch_cons_to_voxel_major(const float *compact, float *S, const Geo &G,
                                      hipStream_t s);
hipError_t launch_cons_planes_to_rows(const float *planes, const ppp_box &pb, float *S, const Geo &G,
                                      hipStream_t s);
hipError_t launch_cons_to_reference(const float *compact, float *ref, const Geo &G,
                                    hipStream_t s);
hipError_t launch_patch_bits(const void *pred, int dtype, const uint32_t *centres, uint64_t n,
                             float thresh, uint32_t *bits, const Geo &G, hipStream_t s);
hipError_t launch_patch_bits_volume(const void *pred, int dtype, float thresh, uint32_t *bits_vol,
                                    const Geo &G, hipStream_t s);
hipError_t launch_synth(const int32_t *labels, void *pred, int dtype, uint32_t seed, float hi,
                        float lo, float noise, unsigned long long voxel_offset, const Geo &G,
                        hipStream_t s);

hipError_t launch_pred_check(const void *pred, int dtype, long long n, const Geo &G, int *unclean, hipStream_t s);
hipError_t launch_counter_calibration(const void *src, int dtype, long long n_read, float *dst, long long n_write,
                                      hipStream_t s);
hipError_t launch_synth_box(const int32_t *labels, const int *lb, void *pred, int dtype, uint32_t seed,
                            float hi, float lo, float noise, const int *gdim, const Geo &G, hipStream_t s);

hipError_t launch_decode_tail(const float *X, long long B, int F, int S, const float *W1, float b1,
                              const float *W2, float b2, const float *W3, float b3, const long long *dst,
                              void *pred, int dtype, const Geo &G, hipStream_t s);

size_t cover_workspace_bytes(long long n, const Geo &G);
hipError_t run_cover_pass(uint8_t *mask, const uint32_t *bits, long long bits_vox, const long long *lin,
                          long long n, int pix_th, int32_t *state, int32_t *cleared, void *work,
                          const Geo &G, hipStream_t s, int *rounds);

size_t rank_order_workspace_bytes(const Geo &G);
hipError_t run_rank_order(const float *score, const uint8_t *fg, long long *lin, float *out_score,
                          long long *n_out, void *work, const Geo &G, hipStream_t s);
size_t mws_edges_workspace_bytes(long long n_rows, long long n_nodes, const Geo &G);
hipError_t run_mws_edges(const uint32_t *rows, const float *aff, long long n_rows, const uint32_t *nodes,
                         long long n_nodes, int32_t *out_u, int32_t *out_v, long long *n_edges, void *work,
                         const Geo &G, hipStream_t s);

size_t thin_workspace_bytes(long long n, const Geo &G);
hipError_t run_thin_cover(const uint8_t *mask, const uint32_t *bits, const long long *lin, long long n,
                          uint8_t *keep, void *work, const Geo &G, hipStream_t s, int *rounds);

hipError_t cover_open(const uint8_t *mask, const long long *lin, const int32_t *rankid, long long n,
                      const int32_t *state, int32_t *cleared, void *work, const Geo &G, hipStream_t s);
hipError_t cover_step_count(const uint32_t *bits, int pix_th, int32_t *state, void *work,
                            const Geo &G, hipStream_t s);
hipError_t cover_step_filter(void *work, const Geo &G, hipStream_t s);
hipError_t cover_step_select(const uint32_t *bits, int pix_th, int32_t *state, int32_t *cleared, void *work,
                             int gZ, const Geo &G, hipStream_t s);
hipError_t cover_alive(void *work, const Geo &G, int32_t *alive, hipStream_t s);
hipError_t cover_close(uint8_t *mask, void *work, const Geo &G, hipStream_t s);
hipError_t cover_zone_export(void *work, int z_lo, int z_hi, int own_lo, int own_hi,
                             int32_t *out_rank, uint8_t *out_mask, uint8_t *out_clean,
                             const Geo &G, hipStream_t s);
hipError_t cover_zone_import(void *work, int z_lo, int z_hi, const int32_t *in_rank,
                             const uint8_t *in_mask, const uint8_t *in_clean, const Geo &G,
                             hipStream_t s);
// set-cover thinning sharded over ranks by z (round 6)
size_t thin_shard_workspace_bytes(const Geo &G);
hipError_t thin_open(const uint8_t *mask, const long long *lin, const int32_t *gidx, long long n, int32_t *state,
                     int32_t *sel_count, int32_t *cleared, void *work, const Geo &G, hipStream_t s);
hipError_t thin_step_count(const uint32_t *bits, int32_t *state, void *work, const Geo &G, hipStream_t s);
hipError_t thin_step_filter(void *work, const Geo &G, hipStream_t s);
hipError_t thin_step_select(const uint32_t *bits, int32_t *state, int32_t *sel_count, int32_t *cleared, void *work,
                            int gZ, const Geo &G, hipStream_t s);
hipError_t thin_alive(void *work, const Geo &G, int32_t *alive, hipStream_t s);
hipError_t thin_close(uint8_t *mask, void *work, const Geo &G, hipStream_t s);
hipError_t thin_zone_export(void *work, int z_lo, int z_hi, int own_lo, int own_hi, long long *out_key,
                            uint8_t *out_mask, uint8_t *out_clean, const Geo &G, hipStream_t s);
hipError_t thin_zone_import(void *work, int z_lo, int z_hi, const long long *in_key, const uint8_t *in_mask,
                            const uint8_t *in_clean, const Geo &G, hipStream_t s);

}  // namespace ppp
